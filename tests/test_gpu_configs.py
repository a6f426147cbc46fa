"""BASELINE.json's other configurations as parity-test cases at sizes the oracle finishes in seconds:
P-shaped (1 view, dense 64-d features, faithful per-point loops in the oracle) and M-shaped
(Matterport mapper: c2w pose, per-view 3x3 K, depth_scale 4000, tau 0.02, 160 classes)."""
import dataclasses

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import parity as o_parity  # noqa: E402
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
    hp.keep_lifted = True
    res = hp.evaluate_scene(batch, vlm)
    _run.ctx = {"vlm_np": vlm_np, "sd": sd, "rigid": rigid}                 # (what oracle/parity.py needs to re-derive a stage)
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
    ctx = _run.ctx
    # no blanket allowance (oracle/parity.py): lifted rows differ only at decisions inside fp32 noise, pooled features within 1e-4
    # at every point, class decisions differ only below a 1e-4 top-2 margin
    info = o_parity.check_scene(ref, res["scene_features"], hp.last_lifted, scene, ctx["vlm_np"], ctx["sd"], ctx["rigid"],
                                dict(K=32, num_iters=3, vectorised=False))
    counts = torch.zeros((3, cfg.num_classes), dtype=torch.int64, device="cuda")
    pred, _ = hp.classify_and_count(res, batch.scene_label, cfg.num_classes, cfg.ignore_ids, counts)
    mism, near = o_parity.check_labels(pred, info["target"])
    print(f"config M shape: lift mismatches {info['lift_mismatches']} (near ties {info['lift_near_ties']}), label mismatches {mism} ({near})")
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


def test_config_v_three_scannet_val_scene_sizes(golden_dir):
    """BASELINE configs[2] as a workload: scenes of the smallest, a typical and the largest ScanNet-val size (28k / ~115k /
    302k points, tests/golden/scannet_val_point_counts.txt) through the whole device path at the benchmark's shape
    (25 views, D=512, K=96, 19 applications, f16x3 student 518->512x9->128).  The oracle cannot run at these sizes in
    seconds: size-independent properties + exact IoU bookkeeping."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import os
    from geopurify_amd import pipeline as pl
    from geopurify_amd import synthetic as syn
    sizes = sorted(int(float(v)) for v in open(os.path.join(golden_dir, "scannet_val_point_counts.txt")).read().split())
    pick = [sizes[0], sizes[len(sizes) // 2 + 20], sizes[-1]]
    assert pick[0] == 28231 and pick[2] == 301855 and 100_000 < pick[1] < 130_000
    base = syn.CONFIGS["S"]
    vlm_np = syn.make_vlm_outputs(base, base.num_views, 1)
    vlm = pl.SyntheticVLM(vlm_np, "cuda")
    sd = pl.random_student_state_dict(base.feat_dim + pl.GEO_DIM, hidden=512, embed=128, num_blocks=4, seed=0)
    hp = pl.HotPath(pl.StudentWeights(sd, "cuda"), base.mask_shape, K=96, num_iters=19, device="cuda")
    counts = torch.zeros((3, base.num_classes), dtype=torch.int64, device="cuda")
    labelled = 0
    for i, n in enumerate(pick):
        cfg = dataclasses.replace(base, num_points=n)
        scene = syn.make_scene(cfg, 900 + i)
        batch = pl.build_scene_batch(pl.upload_scene(scene, "cuda"), pl.scene_rigid_transform(cfg.voxel_size, 900 + i), "cuda")
        assert batch.scene_coords.shape[0] == n and len(batch.views) >= 10
        F, text, scale = hp.lift_masks(batch, vlm)
        out = hp.refine(batch, F)
        assert hp.stats["pool_kernel"] == "cs_pool_kernel" and 0.6 * n < hp.stats["Nv"] <= n
        assert out.shape == (n, 512) and torch.isfinite(out).all()
        # pooling is a convex combination of voxel means of the lifted rows: every column stays inside the lifted range
        assert (out.amax(0) <= F.amax(0) + 1e-5).all() and (out.amin(0) >= F.amin(0) - 1e-5).all()
        # points of one voxel get the same pooled row
        inv = batch.scene_inds_reconstruct
        first = torch.zeros(hp.stats["Nv"], dtype=torch.int64, device="cuda").scatter_(0, inv, torch.arange(n, device="cuda"))
        assert torch.equal(out, out[first[inv]])
        before = counts.clone()
        pred, zero = hp.classify_and_count({"scene_features": out, "text_features": text, "logit_scale": scale}, batch.scene_label,
                                           base.num_classes, base.ignore_ids, counts)
        d = (counts - before).cpu()
        lab = torch.from_numpy(scene.labels)
        valid = lab < base.num_classes
        assert torch.equal(d[2], torch.bincount(lab[valid], minlength=base.num_classes))           # target histogram exact
        assert int(d[1].sum()) == int(valid.sum()) and (d[0] <= d[2]).all() and (d[0] <= d[1]).all()
        assert torch.equal(d[0], torch.bincount(lab[valid & (pred.cpu() == lab)], minlength=base.num_classes))
        labelled += int(valid.sum())
    assert int(counts[2].sum()) == labelled


def test_bench_two_ranks_share_one_gpu_counts_are_the_sum(tmp_path):
    """with --streams 2 every scene's histogram atomics run on side streams; the one collective must be ordered
    after them.  Two ranks (gloo, both on this box's single GPU) run bench.py; the all-reduced target counts must equal the
    labelled points of every scene both ranks evaluated."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import json
    import os
    import socket
    import subprocess
    import sys
    from geopurify_amd import synthetic as syn
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    steps, nscn = 6, 2
    env = dict(os.environ, GP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", "1",
           "--config", "T", "--scenes", str(nscn), "--streams", "2", "--no-cpu-baseline", "--no-train"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    cfg = syn.CONFIGS["T"]
    want = 0
    for rank in range(2):
        per_scene = [int((syn.make_scene(cfg, 5557 + 1000 * rank + s).labels < cfg.num_classes).sum()) for s in range(nscn)]
        want += sum(per_scene[i % nscn] for i in range(steps))
    assert rec["n_gpus"] == 2 and rec["iou_target_points"] == want, (rec["iou_target_points"], want)
    assert rec["value"] > 0 and rec["stages_ms_per_scene"]


def test_bench_config_v_two_ranks_shard_record(tmp_path):
    """BASELINE configs[2] with more than one rank: `bench.py --config V --gpus 2` under gloo on
    this box's one GPU.  The `shard` record must describe an LPT assignment of every scene exactly once, per-rank busy times,
    and the reduced target counts must be the labelled points of all scenes of both ranks."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, GP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    recs = {}
    for policy in ("lpt", "contiguous"):
        tail = [os.path.join(root, "bench.py"), "--gpus", "2", "--warmup", "1", "--config", "V",
                "--val-scenes", "2", "--shard-policy", policy, "--no-cpu-baseline", "--no-train", "--api", "device"]
        if policy == "lpt":
            # the driver's own command form: `python bench.py --gpus 2 ...` with NO launcher -- bench.py
            # starts its two ranks itself (a child torch.distributed.run, before any GPU call) and relays rank 0's line
            env_direct = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
            out = subprocess.run([sys.executable] + tail, env=env_direct, capture_output=True, text=True, timeout=900, cwd=root)
            assert out.returncode == 0, out.stderr[-3000:]
            assert "starting 2 ranks" in out.stderr
            lines = [l for l in out.stdout.splitlines() if l.strip()]
            assert len(lines) == 1 and lines[0].startswith("{"), lines          # ONE JSON line on stdout, nothing else
        else:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                   "--master-port", str(port)] + tail
            out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
            assert out.returncode == 0, out.stderr[-3000:]
        recs[policy] = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    for policy, rec in recs.items():
        sh = rec["shard"]
        assert rec["n_gpus"] == 2 and sh["policy"] == policy and sh["scenes_total"] == 4
        assert len(sh["busy_s_per_rank"]) == 2 and all(b > 0 for b in sh["busy_s_per_rank"])
        assert sh["imbalance_max_over_mean"] >= 1.0
        assert sum(sh["scenes_per_rank"]) == 4 and len(sh["points_per_rank"]) == 2
        assert rec["value"] > 0 and rec["iou_target_points"] > 0 and rec["scaling"] == "weak"
    assert max(recs["lpt"]["shard"]["points_per_rank"]) <= max(recs["contiguous"]["shard"]["points_per_rank"])
    # the same scenes under both policies: the reduced counts do not depend on the assignment
    assert recs["lpt"]["iou_target_points"] == recs["contiguous"]["iou_target_points"]
    assert recs["lpt"]["iou_intersection_points"] == recs["contiguous"]["iou_intersection_points"]
