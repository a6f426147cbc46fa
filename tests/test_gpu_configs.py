"""BASELINE.json's other configurations as parity-test cases at sizes the oracle finishes in seconds:
P-shaped (1 view, dense 64-d features, faithful per-point loops in the oracle) and M-shaped
(Matterport mapper: c2w pose, per-view 3x3 K, depth_scale 4000, tau 0.02, 160 classes)."""
import dataclasses

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pipeline as o_pipe  # noqa: E402


def _run(cfg, seed, dense, K, T, hidden=128, vectorised=True):
    from geopurify_amd import pipeline as pl
    from geopurify_amd import synthetic as syn
    scene = syn.make_scene(cfg, seed)
    rigid = pl.scene_rigid_transform(cfg.voxel_size, seed)
    sd = pl.random_student_state_dict(cfg.feat_dim + pl.GEO_DIM, hidden=hidden, embed=128, num_blocks=1, seed=seed)
    if dense:
        feat = syn.make_dense_feature_maps(cfg, cfg.num_views, seed)
        text = np.random.default_rng(seed).normal(size=(cfg.num_classes, cfg.feat_dim)).astype(np.float32)
        vlm_np = {"text_embed": text, "logit_scale": np.float32(14.0)}
        vlm = pl.DenseFeatureVLM(feat, text, 14.0, "cuda")
        ref = o_pipe.evaluate_scene_oracle(scene, vlm_np, sd, rigid, K=K, num_iters=T, dense_feat=feat)
    else:
        vlm_np = syn.make_vlm_outputs(cfg, cfg.num_views, seed)
        vlm = pl.SyntheticVLM(vlm_np, "cuda")
        ref = o_pipe.evaluate_scene_oracle(scene, vlm_np, sd, rigid, K=K, num_iters=T, vectorised=vectorised)
    batch = pl.build_scene_batch(pl.upload_scene(scene, "cuda"), rigid, "cuda")
    hp = pl.HotPath(pl.StudentWeights(sd, "cuda"), cfg.mask_shape, K=K, num_iters=T, device="cuda")
    res = hp.evaluate_scene(batch, vlm)
    return scene, batch, hp, res, ref


def test_config_p_shape_dense_64d():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from geopurify_amd import synthetic as syn
    cfg = dataclasses.replace(syn.CONFIGS["P"], num_points=12000, image_dim=(324, 242), mask_shape=(242, 324), pitch=0.03)
    scene, batch, hp, res, ref = _run(cfg, 7, dense=True, K=48, T=3)
    assert len(batch.views) == 1 and res["scene_features"].shape == (12000, 64)
    assert torch.equal(batch.scene_inds_reconstruct.cpu(), ref["inv"])
    assert (res["scene_features"].cpu() - ref["scene_features"]).abs().max() < 1e-4


def test_config_m_shape_matterport_mapper():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from geopurify_amd import synthetic as syn
    cfg = dataclasses.replace(syn.CONFIGS["M"], num_points=9000, num_views=4, feat_dim=32, num_queries=24,
                              image_dim=(160, 128), mask_shape=(128, 160), pitch=0.035, min_visible=50)
    scene, batch, hp, res, ref = _run(cfg, 11, dense=False, K=32, T=3, vectorised=False)   # faithful Python loops
    assert cfg.dataset == "matterport" and len(batch.views) == len(ref["views"]) >= 2
    for v, r in zip(batch.views, ref["views"]):
        assert torch.equal(v.pt.cpu(), r["pt"]) and torch.equal(v.x.cpu(), r["x"]) and torch.equal(v.y.cpu(), r["y"])
    d = (res["scene_features"].cpu() - ref["scene_features"]).abs().max(dim=1).values
    assert (d < 1e-4).float().mean() > 0.995
    counts = torch.zeros((3, cfg.num_classes), dtype=torch.int64, device="cuda")
    hp.classify_and_count(res, batch.scene_label, cfg.num_classes, cfg.ignore_ids, counts)
    assert int(counts[2].sum()) == int((scene.labels < cfg.num_classes).sum())


def test_lseg_path_low_resolution_feature_maps():
    """the LSeg-style lift (bilinear align_corners=True resize of the network's [D,h,w] map, evaluated
    only at the visible pixels) through the whole scene path against the oracle (torch CPU F.interpolate)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from geopurify_amd import pipeline as pl
    from geopurify_amd import synthetic as syn
    cfg = dataclasses.replace(syn.CONFIGS["T"], feat_dim=32)
    seed = 23
    scene = syn.make_scene(cfg, seed)
    rigid = pl.scene_rigid_transform(cfg.voxel_size, seed)
    sd = pl.random_student_state_dict(cfg.feat_dim + pl.GEO_DIM, hidden=128, embed=128, num_blocks=1, seed=seed)
    rng = np.random.default_rng(seed)
    H, W = cfg.mask_shape
    feat_lo = rng.normal(size=(cfg.num_views, cfg.feat_dim, 30, 40)).astype(np.float32)      # 4x below the image size
    text = rng.normal(size=(cfg.num_classes, cfg.feat_dim)).astype(np.float32)
    ref = o_pipe.evaluate_scene_oracle(scene, {"text_embed": text, "logit_scale": np.float32(14.0)}, sd, rigid, K=32,
                                       num_iters=3, lseg_feat=(feat_lo, (H, W)))
    batch = pl.build_scene_batch(pl.upload_scene(scene, "cuda"), rigid, "cuda")
    hp = pl.HotPath(pl.StudentWeights(sd, "cuda"), cfg.mask_shape, K=32, num_iters=3, device="cuda")
    vlm = pl.LSegFeatureVLM(feat_lo, (H, W), text, 14.0, "cuda")
    F, _, _ = hp.lift_lseg(batch, vlm)
    assert (F.cpu() - ref["lifted"]).abs().max() <= 1e-6
    res = hp.evaluate_scene(batch, vlm)
    assert (res["scene_features"].cpu() - ref["scene_features"]).abs().max() < 1e-4
