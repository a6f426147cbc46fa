"""End-to-end parity of the device hot path (loader math -> lift -> student -> kNN -> pooling ->
classify/IoU) against the CPU oracle, through the reference-shaped host API."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pipeline as o_pipe  # noqa: E402


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from geopurify_amd import pipeline as pl
    from geopurify_amd import synthetic as syn
    cfg = syn.CONFIGS["T"]
    scene = syn.make_scene(cfg, 321)
    vlm_np = syn.make_vlm_outputs(cfg, cfg.num_views, 321)
    sd = pl.random_student_state_dict(cfg.feat_dim + pl.GEO_DIM, hidden=128, embed=128, num_blocks=2, seed=4)
    rigid = pl.scene_rigid_transform(cfg.voxel_size, 321)
    K, T = 32, 4
    ref = o_pipe.evaluate_scene_oracle(scene, vlm_np, sd, rigid, K=K, num_iters=T)
    batch = pl.build_scene_batch(pl.upload_scene(scene, "cuda"), rigid, "cuda")
    hp = pl.HotPath(pl.StudentWeights(sd, "cuda"), cfg.mask_shape, K=K, num_iters=T, device="cuda")
    return dict(pl=pl, syn=syn, cfg=cfg, scene=scene, vlm_np=vlm_np, sd=sd, rigid=rigid, ref=ref, batch=batch,
                hp=hp, K=K, T=T)


def test_loader_math_bit_exact(env):
    b, ref = env["batch"], env["ref"]
    assert np.array_equal(b.scene_coords_3d.cpu().numpy(), ref["coords_3d"].astype(np.float32))
    assert torch.equal(b.scene_inds_reconstruct.cpu(), ref["inv"])
    assert len(b.views) == len(ref["views"]) > 0
    for v, r in zip(b.views, ref["views"]):
        assert v.src_view == r["src_view"]
        assert torch.equal(v.pt.cpu(), r["pt"]) and torch.equal(v.x.cpu(), r["x"]) and torch.equal(v.y.cpu(), r["y"])


def test_lift_stage(env):
    hp, b, pl = env["hp"], env["batch"], env["pl"]
    F, text, scale = hp.lift_masks(b, pl.SyntheticVLM(env["vlm_np"], "cuda"))
    d = (F.cpu() - env["ref"]["lifted"]).abs().max(dim=1).values
    # fused features are convex combinations of unit vectors: 1e-5 absolute; the rare larger
    # differences are arg-max / top-3 decisions inside fp32 rounding noise of the oracle's own margins
    assert (d < 1e-5).float().mean() > 0.995, (d < 1e-5).float().mean()
    assert d.median() < 1e-6


def test_refine_stage_given_oracle_lift(env):
    """rows 8-12 in isolation: feed the ORACLE's lifted features, compare every downstream product."""
    hp, b, ref = env["hp"], env["batch"], env["ref"]
    out = hp.refine(b, ref["lifted"].cuda().contiguous())
    d = (out.cpu() - ref["scene_features"]).abs().max()
    assert d < 1e-4, d                                    # north_star: pooled features within 1e-4 fp32


def test_end_to_end_features_and_labels(env):
    hp, b, pl, cfg, ref = env["hp"], env["batch"], env["pl"], env["cfg"], env["ref"]
    res = hp.evaluate_scene(b, pl.SyntheticVLM(env["vlm_np"], "cuda"))
    d = (res["scene_features"].cpu() - ref["scene_features"]).abs().max(dim=1).values
    assert (d < 1e-4).float().mean() > 0.995
    counts = torch.zeros((3, cfg.num_classes), dtype=torch.int64, device="cuda")
    pred, zero = hp.classify_and_count(res, b.scene_label, cfg.num_classes, cfg.ignore_ids, counts)
    rpred, (ri, ru, rt) = o_pipe.classify_and_count(ref, env["scene"].labels, cfg.num_classes, cfg.ignore_ids)
    agree = (pred.cpu() == rpred).float().mean().item()
    assert agree > 0.995, agree
    assert np.array_equal(counts[2].cpu().numpy(), rt)     # target histogram is input-only: exact
    assert not zero.any()


def test_reference_api_tuple_path(env):
    """SonataXAffinityTrainer.evaluate_scene on the positional 20-tuple == SceneBatch fast path."""
    from geopurify_amd.affinity_module import SonataXAffinityTrainer
    pl, cfg = env["pl"], env["cfg"]
    model = SonataXAffinityTrainer({"mask_shape": list(cfg.mask_shape), "all_label": ["c"] * cfg.num_classes},
                                   None, None, device="cuda", use_lseg=False, vlm=pl.SyntheticVLM(env["vlm_np"], "cuda"),
                                   feature_dim=cfg.feat_dim, embed_dim=128, hidden_dim=128)
    sd = dict(env["sd"])
    # the trainer always builds 4 residual blocks (affinity_module.py:60-65): extend the 2-block test weights
    for i in (2, 3):
        for k in list(sd):
            if k.startswith("res_blocks.1."):
                sd[k.replace("res_blocks.1.", f"res_blocks.{i}.")] = sd[k].clone()
    model.affinity_student.load_state_dict(sd)
    model.K, model.num_pool_iters = env["K"], env["T"]
    groups = model.affinity_student.get_param_groups()
    assert set(groups) == {"input", "middle", "output"} and len(groups["output"]) == 1
    a = model.evaluate_scene(env["batch"])["scene_features"]
    tup = tuple(t.cpu() if torch.is_tensor(t) else t for t in env["batch"].as_tuple())
    assert len(tup) == 20
    bres = model.evaluate_scene(tup)
    assert torch.equal(a, bres["scene_features"])
    assert bres["text_features"].shape == (cfg.num_classes, cfg.feat_dim)


def test_dense_lift_config(env):
    """P-style config: dense feature maps (row 5) through the same refine path."""
    pl, syn = env["pl"], env["syn"]
    import dataclasses
    cfg = dataclasses.replace(syn.CONFIGS["T"], num_views=2, feat_dim=32)
    scene = syn.make_scene(cfg, 55)
    feat = syn.make_dense_feature_maps(cfg, cfg.num_views, 55)
    text = np.random.default_rng(1).normal(size=(cfg.num_classes, cfg.feat_dim)).astype(np.float32)
    sd = pl.random_student_state_dict(cfg.feat_dim + pl.GEO_DIM, hidden=128, embed=128, num_blocks=1, seed=9)
    rigid = pl.scene_rigid_transform(cfg.voxel_size, 55)
    ref = o_pipe.evaluate_scene_oracle(scene, {"text_embed": text, "logit_scale": 14.0}, sd, rigid, K=16, num_iters=2,
                                       dense_feat=feat)
    batch = pl.build_scene_batch(pl.upload_scene(scene, "cuda"), rigid, "cuda")
    hp = pl.HotPath(pl.StudentWeights(sd, "cuda"), cfg.mask_shape, K=16, num_iters=2, device="cuda")
    res = hp.evaluate_scene(batch, pl.DenseFeatureVLM(feat, text, 14.0, "cuda"))
    assert (res["scene_features"].cpu() - ref["scene_features"]).abs().max() < 1e-4
