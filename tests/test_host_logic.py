"""CPU tests of the host-side mirrors (no GPU compute): config loader, state_dict layout, voxelizer
matrices / RNG order, bicubic tap tables, synthetic scenes, scene sharding."""
import json
import os

import numpy as np
import pytest
import torch

from geopurify_amd import config as gp_config


def test_config_flatten_and_overrides(tmp_path):
    y = tmp_path / "c.yaml"
    y.write_text("DATA:\n  voxel_size: 0.02\n  mask_shape: [484, 648]\n  category_split:\n    base_category: [0, 1]\n"
                 "  save_path:\nTEST:\n  test_classes: 19\n  label: ['a', 'b']\n")
    c = gp_config.load_cfg_from_cfg_file(str(y))
    assert c.voxel_size == 0.02 and c.test_classes == 19 and c.category_split.base_category == [0, 1]
    c2 = gp_config.merge_cfg_from_list(c, ["voxel_size", "0.05", "save_path", "out/x", "label", "('x','y')"])
    assert c2.voxel_size == 0.05 and c2.save_path == "out/x" and c2.label == ["x", "y"] and c.voxel_size == 0.02
    with pytest.raises(ValueError):
        gp_config.merge_cfg_from_list(c, ["test_classes", "abc"])
    with pytest.raises(AssertionError):
        gp_config.merge_cfg_from_list(c, ["nope", "1"])


@pytest.mark.skipif(not os.path.isdir("/root/reference/config"), reason="reference yamls only exist in the build container")
def test_config_matches_reference_flattening(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "config_flat.json")))
    for fn, ref in g.items():
        if fn.startswith("__"):
            continue
        c = gp_config.load_cfg_from_cfg_file(os.path.join("/root/reference/config", fn))
        assert json.loads(json.dumps(dict(c), default=lambda o: dict(o))) == ref


def test_student_state_dict_layout():
    from geopurify_amd.affinity_module import AffinityPredictor
    m = AffinityPredictor(518, embed_dim=128, hidden_dim=512)
    sd = m.state_dict()
    assert tuple(sd["input_layer.0.kernel"].shape) == (27, 518, 512)
    assert tuple(sd["res_blocks.3.conv2.kernel"].shape) == (27, 512, 512)
    assert tuple(sd["output_layer.kernel"].shape) == (512, 128)
    for k in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
        assert f"input_layer.1.bn.{k}" in sd and f"res_blocks.0.norm2.bn.{k}" in sd
    g = m.get_param_groups()
    assert len(g["input"]) == 3 and len(g["middle"]) == 4 * 6 and len(g["output"]) == 1
    assert sum(p.numel() for p in m.parameters()) == 27 * 518 * 512 + 8 * 27 * 512 * 512 + 512 * 128 + 9 * 2 * 512


def test_voxelizer_matrices_follow_reference_rng(golden_dir):
    from geopurify_amd.voxelizer import default_voxelizer
    g = np.load(os.path.join(golden_dir, "voxelize.npz"))
    for case in (0, 1):
        p = f"c{case}_"
        np.random.seed(int(g[p + "seed"]))
        M_v, M_r = default_voxelizer(float(g[p + "voxel_size"])).get_transformation_matrix()
        assert np.array_equal(M_v, g[p + "M_v"]) and np.array_equal(M_r, g[p + "M_r"])


def test_bicubic_taps_match_torch():
    import torch.nn.functional as F
    from geopurify_amd.bicubic import aa_bicubic_taps
    for a, b in [(42, 162), (32, 121)]:
        x0, w = aa_bicubic_taps(a, b)
        eye = torch.eye(a).reshape(1, a, 1, a)
        ref = F.interpolate(eye, size=(1, b), mode="bicubic", align_corners=False, antialias=True)[0, :, 0, :].numpy()
        W = np.zeros((a, b), np.float32)
        for i in range(b):
            for j in range(4):
                if w[i, j] != 0:
                    W[x0[i] + j, i] = w[i, j]
        assert np.abs(W - ref).max() < 1e-6
    with pytest.raises(ValueError):
        aa_bicubic_taps(100, 40)


def test_synthetic_scene_shapes():
    from geopurify_amd import synthetic as syn
    cfg = syn.CONFIGS["T"]
    s = syn.make_scene(cfg, 1)
    assert s.coords.shape == (cfg.num_points, 3) and s.coords.dtype == np.float64
    assert len(s.views) == cfg.num_views and s.views[0].depth.shape == (cfg.image_dim[1], cfg.image_dim[0])
    s2 = syn.make_scene(cfg, 1)
    assert np.array_equal(s.coords, s2.coords) and np.array_equal(s.views[1].depth, s2.views[1].depth)
    v = syn.make_vlm_outputs(cfg, 2, 1)
    H, W = cfg.mask_shape
    assert v["pred_masks"].shape == (2, cfg.num_queries, ((H + 31) // 32) * 8, ((W + 31) // 32) * 8)
    assert v["pred_logits"].shape == (2, cfg.num_queries, cfg.num_classes + 1)


def test_validation_driver_cli_and_dataset_name(tmp_path):
    from geopurify_amd import validation
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = validation.get_parser(["--config", os.path.join(root, "config", "geopurify_synthetic_scannet.yaml"),
                                 "--split_idx", "1", "--split_total", "4", "save_path", str(tmp_path / "o"),
                                 "num_scenes", "5", "voxel_size", "0.05"])
    assert cfg.split_idx == 1 and cfg.split_total == 4 and cfg.num_scenes == 5 and cfg.voxel_size == 0.05
    assert cfg.test_classes == 19 and len(cfg.all_label) == 19 and cfg.category_split.ignore_category == [19, 20]
    assert os.path.isdir(tmp_path / "o" / "result" / "best")
    assert validation.get_dataset_name("data/ScanNet_3d") == "scannet"
    assert validation.get_dataset_name("/x/matterport_3d_160") == "matterport"
    with pytest.raises(ValueError):
        validation.get_dataset_name("data/nuscenes")


def test_lr_schedule_matches_torch_sequential_lr():
    """run/train.py:320-325: LinearLR(1e-6 -> 1) for warmup_iters, then CosineAnnealingLR(eta_min = base_lr*1e-3),
    three groups at 0.1x / 1x / 5x the base rate."""
    import warnings
    from torch.optim.lr_scheduler import CosineAnnealingLR, LinearLR, SequentialLR
    from geopurify_amd.training import lr_schedule
    base, warm, main = 1e-4, 6, 20
    ps = [torch.nn.Parameter(torch.zeros(1)) for _ in range(3)]
    opt = torch.optim.AdamW([{"params": [ps[0]], "lr": base * 0.1}, {"params": [ps[1]], "lr": base}, {"params": [ps[2]], "lr": base * 5.0}])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sched = SequentialLR(opt, schedulers=[LinearLR(opt, start_factor=1e-6, end_factor=1.0, total_iters=warm),
                                              CosineAnnealingLR(opt, T_max=main, eta_min=base * 1e-3)], milestones=[warm])
        for step in range(warm + main):
            got = [lr_schedule(step, base, g, warm, main) for g in ("input", "middle", "output")]
            want = [g["lr"] for g in opt.param_groups]
            assert np.allclose(got, want, rtol=1e-6, atol=1e-12), (step, got, want)
            opt.step()
            sched.step()


def test_oracle_adamw_is_torch_adamw():
    """the oracle's written-out AdamW (oracle/train.py) against torch.optim.AdamW, three steps."""
    from oracle import train as o_train
    torch.manual_seed(0)
    p0 = {"res_blocks.0.conv1.kernel": torch.randn(3, 4, 5), "output_layer.kernel": torch.randn(5, 2)}
    ref = {k: v.clone().requires_grad_(True) for k, v in p0.items()}
    opt = torch.optim.AdamW([{"params": [ref["res_blocks.0.conv1.kernel"]], "lr": 1e-3}, {"params": [ref["output_layer.kernel"]], "lr": 5e-3}],
                            weight_decay=1e-2)
    cur, state = {k: v.clone() for k, v in p0.items()}, {}
    for step in range(1, 4):
        grads = {k: torch.randn_like(v) for k, v in p0.items()}
        for k in ref:
            ref[k].grad = grads[k].clone()
        opt.step()
        cur = o_train.adamw_step(cur, grads, state, step, 1e-3, 1e-2)
    for k in ref:
        assert (cur[k] - ref[k].detach()).abs().max() < 1e-6


def test_train_driver_checkpoint_resume_and_scheduler(tmp_path):
    """run/train.py:215-263,320-334,371-391: checkpoint keys, resume (epoch from the dict or from the file name, bare
    state_dict accepted), optimizer state restored, scheduler fast-forwarded by start_epoch * len(loader) steps."""
    import warnings
    from geopurify_amd import train_driver as td
    from geopurify_amd.affinity_module import AffinityPredictor
    torch.manual_seed(0)
    st = AffinityPredictor(38, 16, 32)
    opt = td.build_optimizer(st, 1e-3, 1e-5)
    assert [g["name"] for g in opt.param_groups] == ["input_group", "middle_group", "output_group"]
    assert np.allclose([g["lr"] for g in opt.param_groups], [1e-4, 1e-3, 5e-3])
    for p in st.parameters():
        p.grad = torch.randn_like(p)
    opt.step()
    path = str(tmp_path / "affinity_predictor_epoch_3.pth")
    td.save_checkpoint(path, 3, st, opt, {"loss_train": {4: 1.25}})
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "tensorboard_scalars"}
    assert "input_layer.0.kernel" in ck["model_state_dict"] and "res_blocks.3.norm2.bn.running_var" in ck["model_state_dict"]
    st2 = AffinityPredictor(38, 16, 32)
    opt2 = td.build_optimizer(st2, 1e-3, 1e-5)
    start, scalars = td.load_resume(st2, opt2, path, "cpu")
    assert start == 4 and scalars == {"loss_train": {4: 1.25}}
    assert all(torch.equal(a, b) for a, b in zip(st.state_dict().values(), st2.state_dict().values()))
    assert opt2.state_dict()["state"][0]["step"] == opt.state_dict()["state"][0]["step"]
    bare = str(tmp_path / "weights_epoch_7.pth")
    torch.save(st.state_dict(), bare)                                   # bare state_dict: epoch parsed from the file name
    assert td.load_resume(AffinityPredictor(38, 16, 32), td.build_optimizer(AffinityPredictor(38, 16, 32), 1e-3, 1e-5), bare, "cpu")[0] == 8
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = td.build_scheduler(opt, 1e-3, 2, 10, 5)
        for _ in range(4 * 5):
            a.step()
        from geopurify_amd.training import lr_schedule
        assert np.allclose(a.get_last_lr(), [lr_schedule(20, 1e-3, g, 10, 40) for g in ("input", "middle", "output")], rtol=1e-6)


def test_compat_shims_resolve_reference_module_names():
    """compat/: the reference drivers' import names (`models.affinity_module`, `dataset.voxelizer`, `util.config`,
    `MinkowskiEngine` ...) resolve to the mirrors."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import models.affinity_module as m, dataset.voxelizer as v, dataset.voxelization_utils as vu, dataset.feature_loader as fl;"
        "import models.utils.fusion_util as fu, util.config as c, util.util as u, MinkowskiEngine as ME;"
        "import geopurify_amd.affinity_module as g;"
        "assert m.SonataXAffinityTrainer is g.SonataXAffinityTrainer and m.AffinityPredictor is g.AffinityPredictor;"
        "assert hasattr(v, 'Voxelizer') and hasattr(vu, 'sparse_quantize') and hasattr(fl, 'FusedFeatureLoader');"
        "assert hasattr(fu, 'PointCloudToImageMapper') and hasattr(fu, 'PointCloudToImageMappermatterport');"
        "assert hasattr(c, 'load_cfg_from_cfg_file') and hasattr(u, 'intersectionAndUnionGPU');"
        "mod = object(); assert ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm(mod) is mod; print('ok')")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(root, "compat"), root]))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr


def test_bench_line_contract_on_the_committed_run():
    """The bench.py JSON lines of the committed default runs (profiles/r01e_... and r02_bench_default_run.json, produced on
    the MI355X box by `python bench.py`) carry every key of the driver's contract with consistent values."""
    for name in ("r01e_bench_default_run.json", "r02_bench_default_run.json", "r04_bench_default_run.json"):
        _check_bench_line(name)
    # round 4: a roofline line per stage, the convolution layer's ceilings (next 4 ii), host enqueue time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(root, "profiles", "r04_bench_default_run.json")))
    st = d["roofline_stages"]
    for k in ("voxelize", "project+lists", "lift+fuse+fill", "scatter_mean", "kNN", "affinity", "pooling", "gather", "classify+iou"):
        assert st[k]["ms"] > 0 and abs(st[k]["frac"] - st[k]["achieved"] / 8000.0) < 1e-3, k
        assert abs(st[k]["achieved"] - st[k]["algorithmic_bytes"] / (st[k]["ms"] * 1e-3) / 1e9) < 0.002 * st[k]["achieved"] + 1.0, k   # (ms is rounded)
    assert st["student"]["ms"] > 10 * st["affinity"]["ms"]
    c = d["roofline_conv"]["ceilings"]
    assert c["mfma_floor_ms"] < c["no_mfma_ms"] < c["layer_ms"] and c["zero_operand_ms"] < c["layer_ms"]
    assert abs(c["mfma_floor_ms"] - 3 * 2.0 * c["pairs"] * 512 * 512 / 2.5e15 * 1e3) < 1e-3
    assert 0 < d["host_ms_per_scene"]["look_ahead_hook"] <= d["host_ms_per_scene"]["enqueue_total"] < d["ms_per_step"]
    assert d["roofline"]["gather_store_ceiling"]["frac"] > d["roofline"]["frac_isolated"]
    assert "r04_pool_pmc_summary" in d["roofline"]["traffic_source"]


def _check_bench_line(name):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(root, "profiles", name)))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - d["n_gpus"]) < 1e-3          # scenes/s x s/scene = GPUs
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1.0
    assert r["traffic"] is None or r["traffic"] > r["algorithmic_bytes_per_launch"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert d["value"] / c["value"] > 50                                           # BASELINE target: >= 50x the CPU path
    if name.startswith("r02"):
        assert c["sample"].startswith("oracle") and "ONE whole" in c["sample"]    # a measured scene, not an extrapolation
        assert d["config"]["schedule"] in ("split", "alternate") and d["iou_target_points"] > 0


def test_bench_helpers_traffic_and_schedule():
    """bench.py helpers that do not need a GPU: the PMC-derived traffic figure (2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes,
    scaled to the scene's voxel count) and the bounded host-thread count."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rec = json.load(open(os.path.join(root, "profiles", "pool_pmc.json")))["cs_pool_kernel"]
    t = bench.pmc_traffic("cs_pool_kernel", rec["nv"])
    assert t["traffic"] == int((2 * rec["fetch_kib"] + rec["write_kib"]) * 1024) and "pool_pmc.json" in t["traffic_source"]
    assert bench.pmc_traffic("cs_pool_kernel", rec["nv"] // 2)["traffic"] == int((2 * rec["fetch_kib"] + rec["write_kib"]) * 1024 * (rec["nv"] // 2) / rec["nv"])
    assert bench.pmc_traffic("no_such_kernel", 1000) == {"traffic": None}
    assert 1 <= bench.host_threads() <= 16


def test_bench_gpus_n_starts_n_ranks_itself():
    """`python bench.py --gpus 2` with no launcher environment must start TWO ranks by itself (a child
    torch.distributed.run, started before the parent's first GPU call) and fail loudly when they do.  On this GPU-less
    container every rank stops at bench.py's "needs a GPU" assertion: the parent must relay a non-zero exit code, print no
    JSON line, and the launcher's log must show both ranks (the GPU-side twin of this test runs the same command to the end:
    tests/test_gpu_configs.py::test_bench_config_v_two_ranks_shard_record)."""
    if torch.cuda.is_available():
        pytest.skip("GPU box: covered by tests/test_gpu_configs.py")
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "T", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode != 0
    assert "starting 2 ranks" in out.stderr and "--nproc-per-node=2" in out.stderr
    assert "bench.py needs a GPU" in out.stderr                      # the ranks ran bench.py's main() and said why they stopped
    assert "rank: 1" in out.stderr or "local_rank: 1" in out.stderr or "[rank1]" in out.stderr or "rank      : 1" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    # a launcher environment whose size contradicts --gpus is an error, not a silent one-rank run
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--config", "T"],
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120, cwd=root)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr


def test_bench_rank_watchdog_and_config_v_default():
    """(a) `bench.py --gpus N --config V` runs BASELINE configs[2]'s 312 scenes IN TOTAL unless --val-scenes is
    given; (b) a rank that never comes back must not hang `python bench.py --gpus 2`: after --rank-timeout seconds the parent kills
    the child launcher's process group, repeats the ranks' last stderr lines and exits non-zero (124), printing no JSON line."""
    import importlib.util
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.default_val_scenes(1) == 16
    for n in (2, 4, 8):
        per = bench.default_val_scenes(n)
        assert per == -(-312 // n) and len(bench.val_scene_sizes(per, n)) == 312      # the whole list, every scene once
    assert sorted(bench.val_scene_sizes(39, 8)) == sorted(bench.val_scene_sizes(312, 1))
    # (runs on GPU boxes too: the ranks sleep BEFORE their first GPU call -- the hidden --selftest-hang flag -- and the parent, which never
    # touches the GPU, kills the child's process group)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "T", "--no-cpu-baseline",
                          "--rank-timeout", "45", "--selftest-hang", "all"],     # every rank sleeps (a lone sleeper is reaped by the launcher)
                         env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 124, (out.returncode, out.stderr[-2000:])
    assert time.time() - t0 < 200
    assert "still running after 45 s" in out.stderr and "last stderr lines of the ranks" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_conv_partial_row_format_restatement():
    """oracle/conv_partial.py (the numpy statement of the 24-bit partial rows the GPU tests compare the kernels with, byte for byte): round
    trip within half a unit of the block exponent and <= 2^-22 of the quarter's maximum; an all-ones mantissa takes the next exponent; u stays
    inside 23 bits; a NaN / Inf quarter is marked 255; the layout puts lane f's first 16 bytes at 16 f and its last 8 at 256 + 8 f."""
    import numpy as np
    from oracle import conv_partial as cp
    rng = np.random.default_rng(0)
    P, cout = 37, 384
    v = (rng.standard_normal((P, cout)) * np.exp(rng.standard_normal((P, 1)) * 5.0)).astype(np.float32)
    v[3, :128] = 0.0
    v[3, 7] = np.float32(1.9999999)                       # mantissa all ones: exponent field 127 -> E = 128
    v[5, 130] = np.float32(-1024.0)
    rows, E = cp.encode(v)
    assert rows.size == P * cout * 3 and E.size == P * cout // 128
    Eq = E.reshape(P, -1).astype(np.int64)
    assert Eq[3, 0] == 128
    dec = cp.decode(rows, E, P, cout)
    m = np.abs(v.reshape(P, -1, 128)).max(axis=2)
    half = np.exp2(Eq - 149.0)
    assert (np.abs(dec - v).reshape(P, -1, 128) <= half[:, :, None]).all() and (half <= m * 2.0 ** -22 * (1 + 1e-6)).all() and (half > m * 2.0 ** -24).all()   # (an all-ones mantissa: 2^-22 (1 + 2^-24))
    # layout: element j of lane f of quarter q of row p
    p_, q_, f_, j_ = 11, 2, 9, 6
    u = int(np.rint(float(v[p_, q_ * 128 + f_ * 8 + j_]) * 2.0 ** (148 - int(Eq[p_, q_])))) + (1 << 22)
    rec = rows.reshape(P, cout // 128, 384)[p_, q_]
    lane = np.concatenate([rec[f_ * 16:f_ * 16 + 16], rec[256 + f_ * 8:256 + f_ * 8 + 8]])
    assert int(lane[3 * j_]) | int(lane[3 * j_ + 1]) << 8 | int(lane[3 * j_ + 2]) << 16 == u
    bad = v.copy()
    bad[2, 200] = np.nan
    bad[4, 10] = np.inf
    Eb = cp.exponents(bad)
    assert Eb[2, 1] == 255 and Eb[4, 0] == 255 and (Eb[2, 0] < 255) and (Eb[4, 1] < 255)
    assert cp.exponent_offset(P, cout) == ((P * cout * 3 + 15) // 16) * 16
