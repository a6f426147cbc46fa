#!/usr/bin/env python3
"""bench.py -- scenes/sec of the GeoPurify per-scene hot path on MI355X (BASELINE.json metric).

A "step" = ONE ScanNet-shaped synthetic scene (config S: ~150k points, 25 views, 512-d X-Decoder
features, K=96, 19 pooling applications as in the reference code) through the whole device path:
voxelizer + per-view mapping + mask-embedding lift + top-3 fusion + fills + point->voxel mean +
Student Affinity Network (9 sparse 3-D convs + linear) + exact kNN + affinity softmax + pooling +
gather + classify + IoU histogram.  Inputs (point cloud, depth maps, poses, synthetic VLM outputs,
random-init student weights) are resident in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

Scenes shard one per GPU with no data-path collective (weak scaling); the only collective is one
int64 all-reduce of the [3,C] IoU counts after the local loop, as the reference's (dead) metric
reduce would do (run/validation.py:441-450).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
FP32_MFMA_PEAK_TF = 157.3        # dense fp32-input MFMA peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 24; config V: every scene of this rank once)")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--config", default="S", choices=["P", "S", "M", "T", "V"],
                    help="S = BASELINE configs[1] (default); V = configs[2]: ScanNet-val scene SIZES (tests/golden/"
                         "scannet_val_point_counts.txt), a fixed subset of --val-scenes per GPU, sharded over the ranks")
    ap.add_argument("--val-scenes", type=int, default=None,
                    help="config V: scenes per GPU.  Default: ceil(312 / N) for N > 1 GPUs -- BASELINE configs[2]'s 312 ScanNet-val scenes "
                         "in total (run/validation.py:269-286 splits the whole list over the ranks) -- and a 16-scene stride subset at N = 1 "
                         "(generating 312 synthetic scenes on one rank's host cores takes tens of minutes; --val-scenes 312 runs them all)")
    ap.add_argument("--rank-timeout", type=float, default=float(os.environ.get("GP_BENCH_RANK_TIMEOUT", "2400")),
                    help="--gpus N > 1 without a launcher: wall-clock limit (s) of the child launcher; on expiry its process group is "
                         "killed, the ranks' last stderr lines are shown and bench.py exits non-zero")
    ap.add_argument("--shard-policy", default="lpt", choices=["lpt", "contiguous"], help="config V: scene -> rank assignment")
    ap.add_argument("--cpu-sample", default="full", choices=["full", "bounded"],
                    help="cpu_baseline: one whole scene of the workload through the oracle (measured, ~1 min at S) or the "
                         "bounded sub-sampled scene extrapolated per stage")
    ap.add_argument("--pool-iters", type=int, default=19, help="applications of A (reference code: 19; BASELINE wording: 3)")
    ap.add_argument("--pool-mode", default=os.environ.get("GP_POOL_MODE", "auto"), choices=["auto", "mfma_cs", "mfma_chain", "mfma_engine", "mfma", "mfma_persist", "tiles", "ell"])
    ap.add_argument("--pool-row-order", default="rcb", choices=["rcb", "morton"],
                    help="row order of the pooling operator: rcb (default: recursive coordinate bisection inside 1024-row Morton chunks into the "
                         "operator's 128-row blocks, smaller neighbour unions) or morton (the voxel order itself: rounds 1-5)")
    ap.add_argument("--api", default="both", choices=["device", "both"],
                    help="both: after the headline run also time the DROP-IN call SonataXAffinityTrainer.evaluate_scene(20-tuple of "
                         "CPU tensors) (run/validation.py:408), reported as the extra object `api_tuple` -- never as `value`")
    ap.add_argument("--scenes", type=int, default=2, help="distinct synthetic scenes rotated through the steps")
    ap.add_argument("--time-every", type=int, default=3, help="HIP-event pairs around the pooling launches and the convolution layers of every "
                    "N-th timed scene (1 = every scene: the pairs then cost 1.1 %% of the rate)")
    ap.add_argument("--streams", type=int, default=2, help="--schedule alternate: HIP streams that consecutive scenes alternate over "
                    "(1 = everything on one stream); --schedule split always uses two")
    ap.add_argument("--schedule", default="auto", choices=["auto", "split", "alternate"],
                    help="auto (default): split for the mask lift, alternate for the dense-feature lift of config P (a handful of "
                         "small kernels: 100 vs 92 scenes/s).  split: refine + classify of scene i on one stream, loader + lift of scene i+1 on a second "
                         "one beside scene i's convolutions and joined before its pooling; alternate: whole scenes alternate "
                         "over --streams streams (3 %% more scenes/s, but the pooling launches then share the chip with the "
                         "other scene's kernels: 0.40 instead of 0.26 ms per launch)")
    ap.add_argument("--selftest-hang", default=None, help=argparse.SUPPRESS)      # rank id or "all": the launcher watchdog's test
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the training-step rate (extra object `training_step`)")
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="budget of the bounded CPU-baseline sample")
    a = ap.parse_args()
    if a.steps is None:
        # 24 scenes = 0.6 s of timed region on S.  The schedule has edges: the first timed scene does its own loader + lift + prepare
        # in front of its student (the warm-up's last scene did not look ahead) and the last one has no look-ahead beside it -- with
        # 6 steps the edges cost 1 % of the rate (25.9 vs 25.65 ms per scene at 24 or 64 steps, profiles/r04_steps_sweep.log)
        a.steps = 0 if a.config == "V" else 24           # 0 = resolved to the rank's scene count below
    return a


def _lib_path():
    from geopurify_amd import _lib
    return _lib.LIB_PATH


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def host_threads():
    """Cores this process may actually use (cgroup / affinity share of the GPU box, capped at 16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


MFMA_F16_PEAK_TFLOPS = 2500.0      # dense f16/bf16 matrix-core peak, MI355X_MICROARCH.md


def conv_roofline(timer):
    """The student's 512->512 submanifold convolution layers (conv_phase1_dma_kernel + conv_phase2_kernel):
    every fp32-class product is three f16 MFMAs (hi*hi + hi*lo + lo*hi), so the issued rate is 3 x the
    algorithmic one; `frac` prices the ISSUED f16 flops against the dense f16 peak."""
    r = timer.summary()
    if r is None:
        return None
    ms, flop = r
    issued = 3.0 * flop / (ms * 1e-3) / 1e12
    return {"kernel": "conv_phase1_dma_kernel + conv_phase2_q24_kernel (one 512->512 layer)", "bound": "mfma",
            "achieved": round(issued, 1), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(issued / MFMA_F16_PEAK_TFLOPS, 4),
            "algorithmic_tflops": round(flop / (ms * 1e-3) / 1e12, 1), "algorithmic_flop_per_layer": flop,
            "avg_layer_ms": round(ms, 4)}


def conv_ceilings(hp, dev):
    """What bounds a 512->512 layer (VERDICT r3 next 4 ii): ONE layer of the last scene, alone on the GPU, on random pre-split rows:
      layer_ms             the product kernels (conv_phase1_dma_kernel per chunk + conv_phase2_q24_kernel per chunk)
      no_mfma_ms           the same launches with the MFMAs compiled out of the loop's body (conv_phase1_tuning_kernel, knob 3 = 2):
                           operand gathers into LDS, the partial-row round trip and phase 2 -- the layer's DATA-MOVEMENT ceiling
      zero_operand_ms      the product kernels on all-zero rows and weights: the same instruction stream, the same cycles, at the
                           clock the chip holds when the matrix pipes toggle nothing (MI355X_MICROARCH.md "DVFS give-back")
      mfma_floor_ms        3 x the layer's flops at the dense f16 peak (2.5 PFLOP/s).
    Algorithmic bytes per layer: gathered operand rows P x 2 KiB x 2 column tiles, weight tiles (P / 256) x 2 x 512 KiB, partial
    rows P x 1.5 KiB (24-bit block floating point, round 5; 2 KiB as fp32 before) written and read, output rows (fp32 where kept + hi/lo planes); the memory-side bytes per LAYER are in
    profiles/r04_conv_pmc_summary.json."""
    from geopurify_amd import _lib, ops
    st = hp.student
    pairs = getattr(st, "last_pairs", None)
    lay = [l for l in st.layers if l[0] == "f16x3" and ops.conv_weights_shape(l[1][0])[1] == ops.conv_weights_shape(l[1][0])[2]]
    if pairs is None or not lay:
        return None
    _, (hi, lo), scale, shift = lay[0]
    nv, c = pairs.nv, ops.conv_weights_shape(hi)[1]
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(3)

    def run_on(x, w_hi, w_lo, knob):
        xs = ops.split_f16(x, c, per_row=True)
        ys = (torch.empty((nv, c), dtype=torch.float16, device=dev), torch.empty((nv, c), dtype=torch.float16, device=dev),
              torch.empty(nv, dtype=torch.float32, device=dev))
        if getattr(st, "interleaved_rows", False) and getattr(st, "residual_from_planes", False):
            # the form the student's layers hand each other: interleaved rows in, interleaved rows out
            xs = (ops.interleave_planes(xs[0], xs[1]), None, xs[2])
            ys = (torch.empty((nv, 2 * c), dtype=torch.float16, device=dev), None, ys[2])
        f = lambda: ops.sparse_conv_f16x3(None, pairs, w_hi, w_lo, scale, shift, relu=True, x_split=xs[:2], x_row_inv=xs[2],
                                          out_split=ys[:2], out_row_inv=ys[2], want_f32=False)
        lib.gp_debug_set(3, knob)
        try:
            for _ in range(20):
                f()
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(8):
                    f()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 8)
        finally:
            lib.gp_debug_set(3, 0)
        return float(np.median(ts))
    x = torch.randn(nv, c, device=dev, generator=g)
    full = run_on(x, hi, lo, 0)
    nomfma = run_on(x, hi, lo, 2)
    zero = run_on(torch.zeros(nv, c, device=dev), torch.zeros_like(hi), torch.zeros_like(lo), 0)
    P = float(pairs.num_pairs)
    flop = 2.0 * P * c * c
    alg = {"gathered_rows": P * c * 4 * (c // 256), "weight_tiles": np.ceil(P / 256) * (c // 256) * 256 * c * 4,
           "partial_rows_written_and_read": 2 * P * (c * 3 + c // 128), "output_planes": nv * c * 4}   # 24-bit block floating point
    return {"layer_ms": round(full, 4), "no_mfma_ms": round(nomfma, 4), "zero_operand_ms": round(zero, 4),
            "mfma_floor_ms": round(3 * flop / (MFMA_F16_PEAK_TFLOPS * 1e12) * 1e3, 4), "pairs": int(P), "chunks": int(pairs.num_chunks),
            "algorithmic_bytes_per_layer": {k: int(v) for k, v in alg.items()},
            "note": "one 512->512 layer of the last scene alone on the GPU, random pre-split rows, median of 3 x 8 launches after 20 warm ones"}


def stage_rooflines(ms, n, nv, views, n_vis, cfg, pool_iters, d):
    """north_star: "scenes/sec ... as fraction of the HBM roofline".  One line per stage of the scene: the ALGORITHMIC bytes of
    SURVEY.md section 8(d) (what the stage has to move at least, N points, Nv voxels, V views, n_v visible points per view,
    K = 96) over the HIP-event time of the stage in the one-stream side pass, against 8 TB/s.  The student's stage is priced in
    issued f16 flops by `roofline_conv`; the pooling stage is `roofline`'s kernel x T plus its operator build."""
    H, W = cfg.mask_shape
    K = 96
    alg = {
        "voxelize": ("rows 1-2: N x 48 B (24 in, 8 key, 8 + 8 index out)", n * 48.0),
        "project+lists": ("row 3 + loader glue: V x N x (24 in + 24 out) + V x H x W x 8 (depth maps)", views * n * 48.0 + views * H * W * 8.0),
        "lift+fuse+fill": ("rows 5-7: N x D x 4 written + sum n_v x 8 (entry ids)", n * d * 4.0 + n_vis * 8.0),
        "scatter_mean": ("row 8: N x (D+6) x 4 read + Nv x (D+6) x 4 written + N x 8", n * (d + 6) * 4.0 + nv * (d + 6) * 4.0 + n * 8.0),
        "embed head": ("row 9, output layer + F.normalize: Nv x (512 x 4 split planes + 128 x 4)", nv * (512 * 4.0 + 128 * 4.0)),
        "kNN": ("row 10: Nv x (12 + K x 4)", nv * (12.0 + K * 4)),
        "affinity": ("row 11: Nv x (128 x 4 + K x 4 + K x 4)", nv * (128 * 4.0 + K * 8)),
        "pooling": (f"row 12: {pool_iters} x Nv x (2 x D x 4 + K x 8)", pool_iters * nv * (2.0 * d * 4 + K * 8)),
        "gather": ("row 12 tail + the class decision of row 13 in the same pass (gp_gather_rows_classify, round 6): Nv x D x 4 read + N x D x 4 "
                   "written + N x 9 (prediction, zero flag); wider class tables: the gather alone", (nv + n) * d * 4.0 + n * 9.0),
        "classify+iou": ("row 13: the IoU histograms over N x 16 (prediction, label); the N x D x 4 read of a separate classification pass "
                         "only where the fused gather does not apply (more than 32 classes)", n * 16.0 + (n * d * 4.0 if cfg.num_classes > 32 else 0.0)),
    }
    out = {}
    for name, (what, b) in alg.items():
        t = ms.get(name)
        if t is None or t <= 0:
            continue
        gbs = b / (t * 1e-3) / 1e9
        out[name] = {"ms": round(t, 4), "algorithmic_bytes": int(b), "achieved": round(gbs, 1), "unit": "GB/s",
                     "frac": round(gbs / HBM_PEAK_GBS, 4), "bytes": what}
    if "student convolutions" in ms:                          # the fine pass splits the student: its nine 3x3x3 layers | the head
        ms = dict(ms, student=ms["student convolutions"])
    for name in ("morton order", "grid+kernel_map", "student", "pool plan+split", "pool operator fill"):
        if name in ms:
            out[name] = {"ms": round(ms[name], 4), "note": {"morton order": "internal row order of the voxels (sort of Nv keys): index work, no SURVEY 8(d) figure",
                                                           "grid+kernel_map": "lattice grid + 27-offset kernel map: index work, no SURVEY 8(d) figure",
                                                           "student": "the nine 3x3x3 layers, matrix-core bound: see roofline_conv (the 1x1x1 output layer is the `embed head` line)",
                                                           "pool plan+split": "once per scene, needs the kNN lists only: the pooling operator's union sizes and structure (union rows, fragment masks, the element of every (row, neighbour) weight) + the f16 hi/lo splits of X",
                                                           "pool operator fill": "empty since round 4: the affinity kernel writes the weights straight into fragment order (HotPath(pool_structure_ahead=False): the separate fill pass)"}[name]}
    out["note"] = ("one-stream side pass after the timed region, HIP events at stage boundaries, mean over the side scenes; "
                   "achieved = SURVEY 8(d) algorithmic bytes / stage time; peak 8000 GB/s")
    return out


def training_step_rate(batch, dev, sd, steps=8):
    """SURVEY 8f-1 (BASELINE config 5 shape): optimisation steps per second of the student on the bench scene --
    4096 anchors x (1 + 63) samples, teacher features [N, 1088] synthetic, lifted features random unit rows
    (the lift itself is timed by the headline metric), sampler + forward + backward + AdamW inside the timed region."""
    from geopurify_amd import training
    N = batch.scene_coords.shape[0]
    g = torch.Generator(device=dev).manual_seed(1)
    F_lift = torch.nn.functional.normalize(torch.randn(N, 512, device=dev, generator=g), dim=1)
    F_teacher = torch.randn(N, 1088, device=dev, generator=g)
    tr = training.StudentTrainer(sd, dev, base_lr=1e-4, weight_decay=1e-5, warmup_iters=10, main_iters=1000)
    xyz = batch.scene_coords.float().contiguous()

    def one():
        anchors = torch.randperm(N, device=dev)[:4096]
        o = tr.scene_step(F_lift, batch.scene_gauss_features, batch.scene_inds_reconstruct, batch.scene_coords_3d, xyz, F_teacher,
                          anchors, num_negatives=63, K=96, optimize=True)
        return o
    torch.cuda.synchronize()
    torch.cuda.empty_cache()                           # the inference schedule's cached blocks have other sizes: without this the first steps
    for _ in range(5):                                 # free and re-allocate them one by one (device syncs inside the timed steps)
        o = one()                                      # warm-up: allocator, operator plans
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = []
    for _ in range(steps):
        o = one()
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append(e)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    log("training steps (ms, GPU clock between the steps' ends): " + " ".join(f"{a.elapsed_time(b):.1f}" for a, b in zip(marks[:-1], marks[1:])))
    # roofline of the step's matrix work: every fp32-class product is three f16 MFMAs, so the ISSUED rate is 3 x
    # the algorithmic one; priced over the WHOLE step (sampler, BatchNorm sweeps, top-k, AdamW included), i.e. a lower bound of what
    # the matrix kernels themselves reach -- their own durations are in profiles/r05_train_kernel_stats.csv
    P = int((o["nbr_map"] >= 0).sum().item())
    nvs, cin0, hid, emb, A, Dt = int(o["num_voxels"]), tr.cin_pad, 512, 128, 4096, 1088
    n_mid = sum(1 for k in sd if k.endswith(".conv1.kernel") or k.endswith(".conv2.kernel"))
    fwd = 2.0 * P * cin0 * hid + n_mid * 2.0 * P * hid * hid + 2.0 * nvs * hid * emb
    dgrad = n_mid * 2.0 * P * hid * hid + 2.0 * nvs * hid * emb              # (the input layer's data gradient is not needed)
    wgrad = fwd                                                              # dW[k] = X[in_k]^T dY[out_k]: the same products, reduced over the pairs
    sim = 2.0 * A * N * Dt                                                   # anchors x points teacher similarity (the sampler's GEMM)
    flop = fwd + dgrad + wgrad + sim
    issued = 3.0 * flop / dt / 1e12
    roof = {"bound": "mfma", "achieved": round(issued, 1), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(issued / MFMA_F16_PEAK_TFLOPS, 4),
            "algorithmic_tflops": round(flop / dt / 1e12, 1),
            "gflop_per_step": {"forward (gp_sparse_conv_f16x3 + head)": round(fwd / 1e9, 1), "data gradients (the same operator, transposed weights)": round(dgrad / 1e9, 1),
                               "weight gradients (wgrad_kernel)": round(wgrad / 1e9, 1), "teacher similarity (anchors x points)": round(sim / 1e9, 1)},
            "pairs": P, "note": "issued = 3 x algorithmic f16 flops (hi hi + hi lo + lo hi) / the whole step's time"}
    return {"value": round(1.0 / dt, 3), "unit": "optimizer steps/s (1 scene per step)", "ms_per_step": round(dt * 1e3, 2),
            "sampled_voxels": int(o["num_voxels"]), "loss": round(float(o["loss"]), 4), "data": "synthetic teacher + lifted features",
            "roofline_train": roof}


def api_tuple_rate(batches, vlms, sd, cfg, pool_iters, dev, steps=20, side_stream=None):
    """Throughput of the drop-in entry point itself (run/validation.py:408-411): the reference's DataLoader hands
    `evaluate_scene` the positional 20-tuple of CPU tensors (scene_based_collate_fn).  The same scenes as the headline run,
    as pinned-host tuples (points, colours + normals, labels, per-view lists, the [V*N,2] visibility table, the V RGB images:
    the host -> device copy is part of the call), through geopurify_amd.affinity_module.SonataXAffinityTrainer:
      serial      evaluate_scene(cpu tuple): copy, parse, lift, refine one after the other on one stream;
      prefetched  the next scene's tuple is copied on a second stream while this scene runs (what a DataLoader with
                  pin_memory + a device prefetcher gives), evaluate_scene receives the device tuple.
    Loader math (projection, voxelization) is NOT inside: the tuple already carries its results, as in the reference."""
    import types
    from geopurify_amd.affinity_module import SonataXAffinityTrainer
    ns = types.SimpleNamespace(all_label=[f"c{i}" for i in range(cfg.num_classes)], mask_shape=list(cfg.mask_shape), voxel_size=cfg.voxel_size)
    model = SonataXAffinityTrainer(ns, None, None, device="cuda", use_lseg=False, vlm=vlms[0], feature_dim=cfg.feat_dim).to(dev)
    model.affinity_student.load_state_dict({k: v for k, v in sd.items()}, strict=False)
    model.num_pool_iters = pool_iters
    model.eval()
    tuples, nbytes = [], 0
    H, W = cfg.mask_shape
    for b in batches:
        raw = list(b.as_tuple())
        if not (torch.is_tensor(raw[11]) and raw[11].numel()):           # slot 11: the V RGB images the 2D VLM is run on (:496)
            raw[11] = torch.stack([torch.full((H, W, 3), float(i)) for i in range(len(b.views))])
        t = tuple(x.cpu().pin_memory() if torch.is_tensor(x) else x for x in raw)
        nbytes = sum(x.numel() * x.element_size() for x in t if torch.is_tensor(x))
        tuples.append(t)
    n = len(tuples)

    def run(i, tup):
        model.vlm = vlms[i % n]
        return model.evaluate_scene(tup)
    for i in range(2):
        run(i, tuples[i % n])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        run(i, tuples[i % n])
    torch.cuda.synchronize()
    serial = (time.perf_counter() - t0) / steps
    copy_stream = torch.cuda.Stream()

    def upload(tup):
        with torch.cuda.stream(copy_stream):
            d = tuple(x.to(dev, non_blocking=True) if torch.is_tensor(x) else x for x in tup)
            ev = torch.cuda.Event()
            ev.record(copy_stream)
        return d, ev
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(copy_stream)
    d, ev = upload(tuples[0])
    e1.record(copy_stream)
    torch.cuda.synchronize()
    h2d_ms = e0.elapsed_time(e1)
    nxt = upload(tuples[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        d, ev = nxt
        torch.cuda.current_stream().wait_event(ev)
        if i + 1 < steps:
            nxt = upload(tuples[(i + 1) % n])
        for x in d:
            if torch.is_tensor(x):
                x.record_stream(torch.cuda.current_stream())
        run(i, d)
    torch.cuda.synchronize()
    pref = (time.perf_counter() - t0) / steps
    # look-ahead: geopurify_amd.data_loader.LookAheadLoader around the loader -- it copies the next pinned tuple on its own stream and
    # OFFERS the device tuple (SonataXAffinityTrainer.offer_next): parse, lift and prepare of the next scene run on the trainer's side
    # stream beside this scene's student
    from geopurify_amd.data_loader import LookAheadLoader
    model.side_stream = side_stream                       # (this process's streams share four hardware queues: no new ones here)

    class _Tuples:                                        # (stands in for the DataLoader: pinned 20-tuples, the scene's VLM set per scene)
        def __init__(self, count):
            self.count = count

        def __len__(self):
            return self.count

        def __iter__(self):
            for i in range(self.count):
                yield src[i % n]
    src = tuples
    if os.environ.get("GP_API_DEVICE_TUPLES") == "1":     # (experiment: the tuples already on the device -- what does the copy cost the look-ahead?)
        src = [tuple(x.to(dev) if torch.is_tensor(x) else x for x in t) for t in tuples]
    marks = []
    if os.environ.get("GP_API_TIMELINE") == "1":          # (tuning aid: where a scene's GPU time goes in the look-ahead form)
        hp_api = model._hot_path()
        orig_refine = hp_api.refine

        def refine_marked(batch, F, after_student=None, prepared=None):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()

            def hook():
                ev[1].record()
                if after_student is not None:
                    after_student()
                ev[2].record()
            r = orig_refine(batch, F, after_student=hook, prepared=prepared)
            ev[3].record()
            marks.append((ev, prepared is not None))
            return r
        hp_api.refine = refine_marked
    for rep in range(2):                                  # (the first pass warms the copy and side streams' allocator pools)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i, d in enumerate(LookAheadLoader(_Tuples(steps), model, dev, copy_stream=copy_stream)):
            off = getattr(model, "_offered", None)
            if off is not None:                               # the synthetic 2D outputs are per scene: the offered scene's own
                model._offered = (off[0], vlms[(i + 1) % n], off[2])
            run(i, d)
        torch.cuda.synchronize()
        ahead = (time.perf_counter() - t0) / steps
    for ev, pre in marks[-6:]:
        log(f"api look-ahead, scene on the GPU (lifted ahead: {pre}): student {ev[0].elapsed_time(ev[1]):6.2f} ms, wait for the look-ahead "
            f"{ev[1].elapsed_time(ev[2]):6.2f} ms, affinity + pooling + gather {ev[2].elapsed_time(ev[3]):6.2f} ms")
    return {"entry_point": "geopurify_amd.affinity_module.SonataXAffinityTrainer.evaluate_scene(20-tuple), run/validation.py:408",
            "serial": {"value": round(1.0 / serial, 3), "unit": "scenes/s", "ms_per_scene": round(serial * 1e3, 3)},
            "prefetched": {"value": round(1.0 / pref, 3), "unit": "scenes/s", "ms_per_scene": round(pref * 1e3, 3)},
            "look_ahead": {"value": round(1.0 / ahead, 3), "unit": "scenes/s", "ms_per_scene": round(ahead * 1e3, 3),
                           "what": "for batch_data in geopurify_amd.data_loader.LookAheadLoader(loader, model): evaluate_scene(batch_data) -- the next "
                                   "pinned tuple copied on the wrapper's stream and offered (offer_next): parse + lift + prepare of the next scene "
                                   "beside this scene's student; results bit-identical to the serial call"},
            "h2d_ms": round(h2d_ms, 3), "tuple_mbytes": round(nbytes / 1e6, 1), "steps": steps,
            "note": "pinned host tuples; loader math is not inside (the tuple carries its results); never the headline `value`"}


def pmc_traffic(kernel, nv):
    """HBM-side bytes per launch of the pooling kernel.  Hardware counters cannot be read from inside this process:
    they come from the committed rocprofv3 PMC passes over THIS script (scripts/pmc_pool.sh -> profiles/pool_pmc.json:
    FETCH_SIZE / WRITE_SIZE in KiB per launch, separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    16-B-per-lane reads on gfx950; the record names the voxel count and the commit it was collected at).  Used as is
    when the record's voxel count is this run's (same seeded scene), scaled by the voxel ratio otherwise (and said so);
    null when no counters were collected for this kernel."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pool_pmc.json")) as f:
            rec = json.load(f)[kernel]
        b = (2 * rec["fetch_kib"] + rec["write_kib"]) * 1024
        src = f"profiles/pool_pmc.json ({rec['profile']}, Nv={rec['nv']}, commit {rec.get('commit', '?')})"
        if abs(rec["nv"] - nv) > 0.005 * nv:
            b, src = b * nv / rec["nv"], src + f", scaled to Nv={nv}"
        return {"traffic": int(b), "traffic_source": src}
    except (OSError, KeyError, ValueError):
        return {"traffic": None}


class PoolTimer:
    """HIP events around every pooling launch, on the stream the kernels are launched on."""

    def __init__(self):
        self.events = []
        self.enabled = False
        self.kernel = "pool_ell"

    def wrap(self, ops):
        timer = self
        for name in ("pool_ell", "pool_tiles_apply", "pool_mfma_apply", "pool_mfma_apply_persistent", "pool_cs_apply", "pool_cs_apply_chain"):
            orig = getattr(ops, name)

            def timed(*a, _orig=orig, _name=name, **k):
                if not timer.enabled:
                    return _orig(*a, **k)
                s = torch.cuda.current_stream()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                r = _orig(*a, **k)
                e1.record(s)
                if _name == "pool_cs_apply_chain":            # (x_split, pong, op, d, applications, out): ONE launch = all applications
                    nv = int(a[2].nv) * int(a[4])
                else:
                    nv = int(a[1].nv) if hasattr(a[1], "nv") else int(a[1].shape[0])           # voxel rows of this launch
                timer.events.append((e0, e1, nv))
                timer.kernel = _name
                return r
            setattr(ops, name, timed)

    def mean_ms(self):
        return float(np.mean([a.elapsed_time(b) for a, b, _ in self.events])) if self.events else float("nan")

    def percentiles(self, qs=(10, 50, 90)):
        """launch-duration percentiles (ms) over the recorded launches: the pooling figure moves with clock and cache state,
        a single mean hides that"""
        if not self.events:
            return [float("nan")] * len(qs)
        return [float(v) for v in np.percentile([a.elapsed_time(b) for a, b, _ in self.events], qs)]

    def totals(self):
        """(sum of launch times in ms, sum of voxel rows) over the recorded launches: scenes of different sizes (config V)
        are priced by their own voxel counts."""
        return float(sum(a.elapsed_time(b) for a, b, _ in self.events)), int(sum(nv for _, _, nv in self.events))


class ConvTimer:
    """HIP events around every 3x3x3 convolution layer of the student (both kernels of a layer)."""

    def __init__(self):
        self.events = []                                  # (e0, e1, cin, cout, pairs)
        self.enabled = False

    def wrap(self, ops):
        timer, orig = self, ops.sparse_conv_f16x3

        def timed(x, pairs, w_hi, w_lo, *a, **k):
            if not timer.enabled:
                return orig(x, pairs, w_hi, w_lo, *a, **k)
            s = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            r = orig(x, pairs, w_hi, w_lo, *a, **k)
            e1.record(s)
            _, co_, ci_ = ops.conv_weights_shape(w_hi)
            timer.events.append((e0, e1, ci_, co_, int(pairs.num_pairs)))
            return r
        ops.sparse_conv_f16x3 = timed

    def summary(self):
        """Layers with cin == cout == 512 (8 of the 9): mean time, fp32-equivalent and issued (3 x f16) TFLOP/s."""
        ev = [(a.elapsed_time(b), ci, co, p) for a, b, ci, co, p in self.events if ci == co]
        if not ev:
            return None
        ms = float(np.mean([e[0] for e in ev]))
        flop = float(np.mean([2.0 * e[3] * e[1] * e[2] for e in ev]))
        return ms, flop


class StageTimer:
    """Per-stage HIP-event timing of the one-stream side pass (`stages_ms_per_scene`, `roofline_stages`).  `fine` is handed to
    build_scene_batch / HotPath.refine, which call it after each sub-stage's kernels are enqueued."""

    def __init__(self, fine=False):
        self.marks = []
        self.fine = self.mark if fine else None

    def mark(self, name):
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        self.marks.append((name, e))

    def table(self):
        out = {}
        for (n0, e0), (n1, e1) in zip(self.marks[:-1], self.marks[1:]):
            out[n1] = out.get(n1, 0.0) + e0.elapsed_time(e1)
        return out


def cpu_baseline(scene, vlm_np, sd, rigid, cfg, pool_iters, budget_s, mode="full"):
    """The oracle ("port": torch-CPU / numpy / sklearn restatement of the reference path, vectorised variant -- the
    reference's two per-point Python loops replaced by tensor ops, i.e. the STRONGER CPU baseline -- with the kNN by
    scipy cKDTree on all cores instead of the oracle's quadratic exact search) timed on the host cores.
    mode "full": ONE whole scene of the benchmarked workload, measured end to end (no extrapolation).
    mode "bounded": the same scene generator sub-sampled to 20 % of the points and 1/6 of the views (~budget_s),
    extrapolated per stage (linear in points x views for loader + lift, in points for the rest)."""
    import dataclasses
    from geopurify_amd import synthetic as syn
    from oracle import pipeline as o_pipe
    threads = host_threads()
    torch.set_num_threads(threads)
    kw = {}
    if cfg.dense_features:
        kw["dense_feat"] = vlm_np["dense"]
    if mode == "full":
        timings = {}
        t0 = time.perf_counter()
        o_pipe.evaluate_scene_oracle(scene, vlm_np, sd, rigid, K=96, num_iters=pool_iters, timings=timings, knn_impl="kdtree", **kw)
        wall = time.perf_counter() - t0
        return {"value": round(1.0 / wall, 6), "unit": "scenes/s", "cores": threads, "kind": "port",
                "sample": f"oracle (torch-CPU/numpy/sklearn, vectorised variant, kNN by scipy cKDTree), ONE whole {cfg.name} scene "
                          f"({scene.coords.shape[0]} pts x {len(scene.views)} views, T={pool_iters}) measured end to end: {wall:.1f} s",
                "stages_s": {k: round(v, 3) for k, v in timings.items()}}
    frac_pts, n_views = 0.2, max(2, cfg.num_views // 6)
    small_cfg = dataclasses.replace(cfg, num_points=int(cfg.num_points * frac_pts), num_views=n_views)
    small = syn.make_scene(small_cfg, 5557)
    vlm_small = {k: (v[:n_views] if isinstance(v, np.ndarray) and v.ndim >= 3 and v.shape[0] == cfg.num_views else v)
                 for k, v in vlm_np.items()}
    if cfg.dense_features:
        kw["dense_feat"] = vlm_small["dense"]
    timings = {}
    t0 = time.perf_counter()
    o_pipe.evaluate_scene_oracle(small, vlm_small, sd, rigid, K=96, num_iters=pool_iters, timings=timings, knn_impl="kdtree", **kw)
    wall = time.perf_counter() - t0
    pv = (cfg.num_points * cfg.num_views) / (small_cfg.num_points * small_cfg.num_views)
    pn = cfg.num_points / small_cfg.num_points
    est = sum(v * (pv if k in ("loader(project+voxelize)", "lift per view") else pn) for k, v in timings.items())
    return {"value": round(1.0 / est, 6), "unit": "scenes/s", "cores": threads, "kind": "port",
            "sample": f"oracle (vectorised variant, cKDTree kNN) on {small_cfg.num_points} pts x {n_views} views of the same scene "
                      f"generator, T={pool_iters}: {wall:.1f}s measured; EXTRAPOLATED per stage to {cfg.num_points} pts x "
                      f"{cfg.num_views} views = {est:.1f} s/scene",
            "stages_s_sample": {k: round(v, 3) for k, v in timings.items()}}


def val_scene_sizes(per_gpu, world):
    """BASELINE configs[2]: a fixed subset of the 312 ScanNet-val scene sizes (labelled-point counts of
    dataset/scannet_val_metrics.tsv, committed as tests/golden/scannet_val_point_counts.txt): per_gpu * world sizes taken at
    a constant stride through the list in file order, so the subset keeps the list's spread (28k .. 302k points)."""
    sizes = [int(float(v)) for v in open(os.path.join(ROOT, "tests", "golden", "scannet_val_point_counts.txt")).read().split()]
    n = min(per_gpu * world, len(sizes))
    return [sizes[(i * len(sizes)) // n] for i in range(n)]


VAL_SCENES_TOTAL = 312                                    # BASELINE configs[2]: the ScanNet-val list (dataset/scannet_val_metrics.tsv)


def default_val_scenes(n_gpus):
    """--val-scenes when not given: the whole 312-scene list split over N > 1 ranks, a 16-scene stride subset on one GPU."""
    return 16 if n_gpus <= 1 else -(-VAL_SCENES_TOTAL // n_gpus)


def launch_ranks(n, limit_s):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks HERE.  The parent never touches
    the GPU (no HIP call, no torch.cuda.is_available()): it starts `python -m torch.distributed.run --nproc-per-node N bench.py
    <same arguments>` as a CHILD process (never an exec of itself), relays rank 0's single JSON line and exits non-zero if the
    launcher does, if the line does not say n_gpus == N, or if the child is still running after limit_s seconds of wall clock
    (a hung rank: a collective that never completes, a dead peer) -- its whole process group is then killed and the last stderr
    lines of the ranks are repeated.  One process per GPU, scenes sharded by the split rule of run/validation.py:269-286, ONE
    all-reduce of the IoU counts (:441-450) over RCCL (backend "nccl")."""
    import collections
    import signal
    import socket
    import subprocess
    import threading
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"--gpus {n} without a launcher: starting {n} ranks (limit {limit_s:.0f} s): {' '.join(cmd)}")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_threads() // n)))
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True, start_new_session=True)
    tail = collections.deque(maxlen=60)
    out_lines = []

    def pump_err():
        for ln in p.stderr:
            tail.append(ln.rstrip("\n"))
            print(ln, end="", file=sys.stderr, flush=True)

    def pump_out():
        for ln in p.stdout:
            out_lines.append(ln.rstrip("\n"))
    threads = [threading.Thread(target=pump_err, daemon=True), threading.Thread(target=pump_out, daemon=True)]
    for t in threads:
        t.start()
    try:
        rc = p.wait(timeout=limit_s)
    except subprocess.TimeoutExpired:
        log(f"the {n} ranks are still running after {limit_s:.0f} s: killing the launcher's process group")
        try:
            os.killpg(p.pid, signal.SIGKILL)             # (the child leads its own session: launcher + every rank, nothing else)
        except ProcessLookupError:
            pass
        p.wait()
        for t in threads:
            t.join(5)
        log("last stderr lines of the ranks:")
        for ln in list(tail)[-30:]:
            print("    " + ln, file=sys.stderr)
        sys.exit(124)
    for t in threads:
        t.join(5)
    line = None
    for ln in out_lines:
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if rc != 0:
        log(f"the launcher exited with {rc}")
        sys.exit(rc)
    if line is None:
        log("no JSON line from rank 0")
        sys.exit(1)
    got = json.loads(line).get("n_gpus")
    if got != n:
        log(f"rank 0 reports n_gpus={got}, asked for {n}")
        sys.exit(1)
    print(line, flush=True)
    sys.exit(0)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus, args.rank_timeout)        # does not return
    rank = int(os.environ.get("RANK", "0"))
    if args.selftest_hang in (str(rank), "all"):      # tests/test_host_logic.py (hidden flag): ranks that never come back, before any GPU call
        time.sleep(3600)
    if args.val_scenes is None:
        args.val_scenes = default_val_scenes(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback of the product path)"
    backend = os.environ.get("GP_BENCH_BACKEND", "nccl")      # nccl = RCCL; "gloo" lets a test run two ranks on ONE GPU
    if world > torch.cuda.device_count() and backend == "nccl":
        raise SystemExit(f"bench.py: {world} ranks over RCCL need {world} GPUs, {torch.cuda.device_count()} visible")
    local = local % torch.cuda.device_count()         # more ranks than GPUs only happens in the gloo test (two ranks, one GPU)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    import dataclasses
    from geopurify_amd import ops, pipeline as pl, sharding, synthetic as syn
    val_mode = args.config == "V"
    cfg = syn.CONFIGS["S" if val_mode else args.config]
    # ---- synthetic inputs, resident in HBM before timing ------------------------------------------
    scenes, vlms, rigids, scene_cfgs = [], [], [], []
    vlm_np0 = None
    torch.set_num_threads(host_threads())
    shard = None
    if val_mode:
        sizes = val_scene_sizes(args.val_scenes, world)
        ids = list(range(len(sizes)))
        parts = sharding.assign_scenes_lpt(sizes, world) if args.shard_policy == "lpt" else \
            [sharding.get_batch_scenes(ids, r, world) for r in range(world)]
        mine = parts[rank]
        shard = {"policy": args.shard_policy, "scenes_total": len(sizes), "points_per_rank": [int(sum(sizes[i] for i in p)) for p in parts],
                 "scenes_per_rank": [len(p) for p in parts]}
        todo = [(5557 + gi, sizes[gi]) for gi in mine]
        args.scenes = len(todo)
    else:
        todo = [(5557 + 1000 * rank + s, cfg.num_points) for s in range(args.scenes)]
    shared_vlm = None
    for s, (seed, npts) in enumerate(todo):
        log(f"rank {rank}: generating synthetic scene {s + 1}/{len(todo)} ({cfg.name}, {npts} points)")
        c = dataclasses.replace(cfg, num_points=npts) if npts != cfg.num_points else cfg
        sc = syn.make_scene(c, seed)
        if cfg.dense_features:
            feat = syn.make_dense_feature_maps(cfg, cfg.num_views, seed)
            text = np.random.default_rng(seed).normal(size=(cfg.num_classes, cfg.feat_dim)).astype(np.float32)
            vlm_np = {"text_embed": text, "logit_scale": np.float32(1 / 0.07), "dense": feat}
            vlms.append(pl.DenseFeatureVLM(feat, text, 1 / 0.07, dev))
        elif val_mode and shared_vlm is not None:
            vlms.append(shared_vlm)                       # config V: one set of synthetic 2D outputs serves every scene (the
            vlm_np = vlm_np0                              # geometry, hence every kernel's work, differs per scene)
        else:
            # config V: the shared 2D outputs are seeded independently of the rank and of the assignment, so that the reduced
            # IoU counts of a scene set do not depend on the sharding policy (checked by the two-rank test)
            vlm_np = syn.make_vlm_outputs(cfg, cfg.num_views, 5557 if val_mode else seed)
            vlms.append(pl.SyntheticVLM(vlm_np, dev))
            shared_vlm = vlms[-1]
        if s == 0:
            vlm_np0 = vlm_np
        scenes.append(pl.upload_scene(sc, dev))
        scene_cfgs.append(c)
        rigids.append(pl.scene_rigid_transform(cfg.voxel_size, seed))
    if args.steps <= 0:
        args.steps = max(len(todo), 1)
    sd = pl.random_student_state_dict(cfg.feat_dim + pl.GEO_DIM, hidden=512, embed=128, num_blocks=4, seed=0)
    student = pl.StudentWeights(sd, dev)
    hp = pl.HotPath(student, cfg.mask_shape, K=96, sharpen=20.0, num_iters=args.pool_iters, device=dev, pool_mode=args.pool_mode,
                    pool_row_order=args.pool_row_order)
    counts = torch.zeros((3, cfg.num_classes), dtype=torch.int64, device=dev)
    pool_timer = PoolTimer()
    pool_timer.wrap(ops)
    conv_timer = ConvTimer()
    conv_timer.wrap(ops)

    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.streams))]

    if args.schedule == "auto":
        args.schedule = "alternate" if cfg.dense_features else "split"
    split = args.schedule == "split" and args.streams >= 2          # --streams 1: everything on one stream
    if split:                                           # [0] loader + lift of the NEXT scene, [1] refine + classify
        # the look-ahead stream has the HIGHER priority: its small kernels run beside the convolutions, whose workgroups hold a CU for
        # ~40 us each; the host's read-backs in the look-ahead (voxel count, per-view counts, pair count, union rows) wait for them
        side_prio = int(os.environ.get("GP_BENCH_SIDE_PRIORITY", "-1"))
        main_prio = int(os.environ.get("GP_BENCH_MAIN_PRIORITY", "0"))
        streams = [torch.cuda.Stream(device=dev, priority=side_prio), torch.cuda.Stream(device=dev, priority=main_prio)]
    pending = {}                                        # scene index -> (batch, F, text, scale, lift-done event), lifted ahead
    host_t = {"hook": 0.0, "step": 0.0, "blocked": 0.0, "readbacks": 0}      # host seconds inside the look-ahead hook / inside step() (timed region)

    def _tensors(obj):
        if torch.is_tensor(obj):
            yield obj
        elif isinstance(obj, (list, tuple)):
            for o in obj:
                yield from _tensors(o)
        elif isinstance(obj, dict):
            for o in obj.values():
                yield from _tensors(o)
        elif hasattr(obj, "__dataclass_fields__"):
            for f in obj.__dataclass_fields__:
                yield from _tensors(getattr(obj, f))
        elif type(obj).__module__.startswith("geopurify_amd") and hasattr(obj, "__dict__"):
            yield from _tensors(vars(obj))              # ConvPairs, PoolCs, Grid: plain holders of device arrays

    def step(i, stage=None, stream=None, prefetch=True):
        if split and stream is None:
            return _step_split(i, prefetch)
        with torch.cuda.stream(stream if stream is not None else streams[i % len(streams)]):
            return _step(i, stage)

    timeline = [] if os.environ.get("GP_BENCH_TIMELINE") == "1" else None
    student_end = {}

    def _lift_ahead(i, after=None):
        """Loader + lift + HotPath.prepare (voxel means, kernel map and pairs, kNN lists, pooling plan: everything of refine that
        does not need the student) of scene i on streams[0], after the event `after` (recorded on streams[1])."""
        j = i % max(args.scenes, 1)
        tl = timeline is not None and after is not None
        mk = (lambda: (lambda e: (e.record(streams[0]), e)[1])(torch.cuda.Event(enable_timing=True))) if tl else (lambda: None)
        h0 = time.perf_counter()
        with torch.cuda.stream(streams[0]):
            if after is not None:
                streams[0].wait_event(after)
            else:
                streams[0].wait_stream(streams[1])
            e_first = mk()
            batch = pl.build_scene_batch(scenes[j], rigids[j], dev)
            e_load, h1 = mk(), time.perf_counter()
            F, text, scale = hp.lift_dense(batch, vlms[j]) if cfg.dense_features else hp.lift_masks(batch, vlms[j])
            e_lift, h2 = mk(), time.perf_counter()
            prep = hp.prepare(batch, F)
            done = torch.cuda.Event(enable_timing=tl)
            done.record(streams[0])
        if tl:      # GP_BENCH_TIMELINE=1: where the look-ahead ran relative to the student it was meant to run beside
            timeline.append({"after": after, "gpu": (e_first, e_load, e_lift, done), "host": (h0, h1, h2, time.perf_counter())})
        # No record_stream on these tensors (round 6): the allocator answered each with an event record on the consumer stream when the block
        # was freed -- ~40 markers in a row at every scene boundary, 0.14 ms in which streams[1] ran nothing (rocprofv3 kernel trace,
        # profiles/r06_event_pairs_and_boundary.log).  The blocks are safe without: they belong to streams[0], and every LATER kernel of streams[0] sits
        # behind `after` -- the start of the NEXT refine on streams[1], i.e. behind every kernel of the scene that used them (or, for a scene
        # that was not lifted ahead, behind the whole of streams[1]: the wait_stream above).
        pending[i] = (batch, F, text, scale, done, prep)

    def _step_split(i, prefetch):
        """Scene i's refine + classify on streams[1]; the NEXT scene's loader + lift is enqueued on streams[0] from inside
        refine (HotPath.refine's after_student hook), gated on the start of this refine and joined before this scene's kNN /
        affinity / pooling: the lift's memory-bound kernels run beside the matrix-core-bound convolutions, the pooling
        kernel has the chip to itself.  One scene = one lift + one refine, as in the other schedules."""
        if i not in pending:
            _lift_ahead(i)
        batch, F, text, scale, done, prep = pending.pop(i)
        with torch.cuda.stream(streams[1]):
            streams[1].wait_event(done)
            started = torch.cuda.Event(enable_timing=timeline is not None)
            started.record(streams[1])
            t_started = time.perf_counter()

            def hook():
                t_h = time.perf_counter()
                if timeline is not None:
                    e_st = torch.cuda.Event(enable_timing=True)
                    e_st.record(streams[1])
                    student_end[started] = (e_st, t_h, t_started)
                if prefetch:
                    _lift_ahead(i + 1, after=started)
                    streams[1].wait_event(pending[i + 1][4])
                host_t["hook"] += time.perf_counter() - t_h
            feats = hp.refine(batch, F, after_student=hook, prepared=prep, classify_text=(text, scale))
            hp.classify_and_count({"scene_features": feats, "text_features": text, "logit_scale": scale},
                                  batch.scene_label, cfg.num_classes, cfg.ignore_ids, counts)
        return batch

    def _step(i, stage=None):
        j = i % max(args.scenes, 1)
        if stage:
            stage.mark("start")
        fine = getattr(stage, "fine", None) if stage else None         # the per-stage roofline pass: marks inside loader and refine
        hp.stage_mark = fine
        batch = pl.build_scene_batch(scenes[j], rigids[j], dev, mark=fine)
        if stage:
            stage.mark("loader: voxelize+project+lists")
        if cfg.dense_features:
            F, text, scale = hp.lift_dense(batch, vlms[j])
        else:
            F, text, scale = hp.lift_masks(batch, vlms[j])
        if stage:
            stage.mark("lift+fuse+fill")
        feats = hp.refine(batch, F, classify_text=(text, scale))
        if stage:
            stage.mark("refine: mean+student+knn+affinity+pool+gather")
        hp.classify_and_count({"scene_features": feats, "text_features": text, "logit_scale": scale},
                              batch.scene_label, cfg.num_classes, cfg.ignore_ids, counts)
        if stage:
            stage.mark("classify+iou")
        hp.stage_mark = None
        return batch

    def join_streams():
        """Order the default stream (on which the collective is issued) after every side stream, and back."""
        cur = torch.cuda.current_stream()
        for st in streams:
            cur.wait_stream(st)

    def fork_streams():
        cur = torch.cuda.current_stream()
        for st in streams:
            st.wait_stream(cur)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    log("inputs resident; warm-up")
    if args.scenes:
        for i in range(args.warmup):
            step(i, prefetch=i + 1 < args.warmup)
    barrier()
    log("timing")
    counts.zero_()
    fork_streams()                                    # the side streams start after the zeroing (default stream)
    # HIP-event pairs around every pooling launch and every convolution layer of EVERY --time-every-th scene of the timed region: a
    # pair costs ~8 us of stream time (the marker waits for the kernel in front of it and holds the one behind), 56 pairs per scene =
    # 0.24 ms = 1.1 % of the scene when every scene is timed (profiles/r06_event_pairs_and_boundary.log); the default, every 3rd scene (both rotated
    # scenes take their turn), still prices 152 pooling launches of a 24-scene timed region
    timers_on = os.environ.get("GP_BENCH_NO_TIMERS") != "1"
    t0 = time.perf_counter()
    last = None
    n_local = args.steps if args.scenes else 0
    host_t["hook"] = 0.0
    ops.READBACK["seconds"], ops.READBACK["calls"] = 0.0, 0
    for i in range(n_local):
        t_s = time.perf_counter()
        pool_timer.enabled = conv_timer.enabled = timers_on and i % max(args.time_every, 1) == 0
        last = step(i, prefetch=i + 1 < n_local)
        host_t["step"] += time.perf_counter() - t_s
    host_t["blocked"], host_t["readbacks"] = ops.READBACK["seconds"], ops.READBACK["calls"]
    join_streams()                                    # every scene's histogram atomics precede the collective
    hp.pool_chain_check()                             # (--pool-mode mfma_chain: the abort words of every timed scene's launch; else nothing pending)
    busy_ev = torch.cuda.Event(enable_timing=False)
    busy_ev.record()
    if world > 1:
        import torch.distributed as dist
        busy_ev.synchronize()
        busy = time.perf_counter() - t0               # this rank's own work, before it waits for the others
        dist.all_reduce(counts)                       # the one collective: int64 [3,C] IoU counts
    barrier()
    dt = time.perf_counter() - t0
    iou_target_points = int(counts[2].sum().item())   # labelled points counted over ALL ranks (before any untimed side pass)
    iou_intersection_points = int(counts[0].sum().item())
    if world == 1:
        busy = dt
    pool_timer.enabled = conv_timer.enabled = False
    if timeline:
        torch.cuda.synchronize()
        for rec in timeline[-6:]:
            st0 = rec["after"]
            e_st, t_hook, t_started = student_end[st0]
            g = [st0.elapsed_time(e) for e in rec["gpu"]]
            h = [(t - t_started) * 1e3 for t in rec["host"]]
            log(f"timeline (ms after this refine's start): student ends {st0.elapsed_time(e_st):6.2f} on the GPU (host done enqueuing it at {(t_hook - t_started) * 1e3:5.2f}); "
                f"look-ahead on the GPU: first {g[0]:6.2f}, loader done {g[1]:6.2f}, lift done {g[2]:6.2f}, prepare done {g[3]:6.2f}; "
                f"on the host: starts {h[0]:6.2f}, loader returned {h[1]:6.2f}, lift returned {h[2]:6.2f}, prepare returned {h[3]:6.2f}")
        timeline = None
    total_steps = n_local
    busy_all = [busy]
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        ns = torch.tensor([n_local], dtype=torch.int64, device=dev)
        dist.all_reduce(ns)
        total_steps = int(ns.item())
        bt = torch.zeros(world, dtype=torch.float64, device=dev)
        bt[rank] = busy
        dist.all_reduce(bt)
        busy_all = bt.cpu().tolist()

    if rank == 0:
        Nv = hp.stats["Nv"]
        D = cfg.feat_dim
        per_row = 2 * D * 4 + 96 * 8                      # algorithmic bytes per voxel row and application of A
        tot_ms, tot_rows = pool_timer.totals()
        tot_ms = tot_ms or float("nan")                 # (GP_BENCH_NO_TIMERS=1: nothing was timed)
        n_launch = max(len(pool_timer.events), 1)
        pool_ms = tot_ms / n_launch                       # mean launch duration and mean algorithmic bytes per launch over the
        pool_bytes_mean = tot_rows / n_launch * per_row   # timed launches (scenes differ in size): achieved = their ratio
        p10, p50, p90 = pool_timer.percentiles()
        # the same launches with nothing else on the GPU (with --streams 2 the timed region overlaps the pooling
        # of one scene with the loader/lift kernels of the next, which share its L2 and HBM bandwidth): one warm pass
        # (operator build, allocator, clocks), then three timed passes; the median pass is reported
        hp._pool(*hp._last_pool_inputs, plan=hp._last_pool_plan)
        hp.pool_chain_check()                    # (chained launches of these isolated passes: read and forget their operators)
        torch.cuda.synchronize()
        alone = []
        for _ in range(3):
            pool_timer.events, pool_timer.enabled = [], True
            hp._pool(*hp._last_pool_inputs, plan=hp._last_pool_plan)
            hp.pool_chain_check()                    # (chained launches of these isolated passes: read and forget their operators)
            torch.cuda.synchronize()
            pool_timer.enabled = False
            alone.append(pool_timer.mean_ms())
        pool_ms_alone = float(np.median(alone))
        pool_bytes = Nv * per_row * (args.pool_iters if hp.stats["pool_kernel"] == "cs_chain_kernel" else 1)   # per LAUNCH
        # the access pattern's own ceiling: the SAME launches with the LDS fragment reads, the MFMAs and the
        # weight-fragment DMA switched off (tuning bits 0 and 3 of the kernel): every union row is still gathered into LDS through
        # the same ring at the same occupancy and every output row is still stored.  Results of these passes are garbage; every
        # later pass recomputes its scene from the inputs.
        ceiling = None
        if hp.stats["pool_kernel"] == "cs_pool_kernel":
            from geopurify_amd import _lib
            lib = _lib.load()
            lib.gp_debug_set(4, 9)
            try:
                hp._pool(*hp._last_pool_inputs, plan=hp._last_pool_plan)
                hp.pool_chain_check()                    # (chained launches of these isolated passes: read and forget their operators)
                torch.cuda.synchronize()
                cl = []
                for _ in range(3):
                    pool_timer.events, pool_timer.enabled = [], True
                    hp._pool(*hp._last_pool_inputs, plan=hp._last_pool_plan)
                    hp.pool_chain_check()                    # (chained launches of these isolated passes: read and forget their operators)
                    torch.cuda.synchronize()
                    pool_timer.enabled = False
                    cl.append(pool_timer.mean_ms())
            finally:
                lib.gp_debug_set(4, 0)
            c_ms = float(np.median(cl))
            ceiling = {"avg_launch_ms": round(c_ms, 4), "passes_ms": [round(a, 4) for a in cl],
                       "frac": round(pool_bytes / (c_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                       "frac_isolated_over_ceiling": round(c_ms / pool_ms_alone, 4),
                       "what": "the isolated launches repeated with the fragment reads, the MFMAs and the weight-fragment DMA switched "
                               "off: union rows gathered into the LDS ring and output rows stored, nothing else (same grid, ring, occupancy)"}
        try:
            conv_ceil = conv_ceilings(hp, dev) if cfg.feat_dim == 512 else None
        except Exception as e:                            # an extra: never lose the headline line over it
            conv_ceil = {"error": repr(e)}
        achieved = tot_rows * per_row / (tot_ms * 1e-3) / 1e9      # all timed launches, each priced by its own voxel count
        # per-stage breakdown from a ONE-stream side pass (stage marks are meaningless while two scenes interleave)
        stage = StageTimer()
        side = min(2, max(args.scenes, 1))
        for i in range(side):
            step(i, stage, stream=streams[0])
        torch.cuda.synchronize()
        stages = stage.table()
        # per-stage roofline: the same pass once more with marks inside the loader and refine
        fine = StageTimer(fine=True)
        fb = [step(i, fine, stream=streams[0]) for i in range(side)]
        torch.cuda.synchronize()
        fine_ms = {k: v / side for k, v in fine.table().items()}
        n_pts = float(np.mean([int(b.scene_coords.shape[0]) for b in fb]))
        n_vis = float(np.mean([sum(int(v.pt.shape[0]) for v in b.views) for b in fb]))
        n_views = float(np.mean([len(b.views) for b in fb]))
        stage_roof = stage_rooflines(fine_ms, n_pts, hp.stats["Nv"], n_views, n_vis, cfg, args.pool_iters, D)
        pairs = int((hp.stats["nbr_map"] >= 0).sum().item())
        flops = student.flops(pairs, Nv)
        if val_mode:
            workload = (f"V: {shard['scenes_total']} ScanNet-val-SIZED synthetic scenes ({args.val_scenes} per GPU, sizes "
                        f"{min(val_scene_sizes(args.val_scenes, world))}..{max(val_scene_sizes(args.val_scenes, world))} points from "
                        f"scannet_val_point_counts.txt), 25 views, D={D}, K=96, pool_iters={args.pool_iters}, student 518->512x9->128")
        else:
            workload = (f"{cfg.name}: ScanNetV2-shaped scene, N={cfg.num_points} pts, {args.scenes} scenes rotated: Nv={Nv} voxels in the LAST scene "
                        f"(the one the isolated passes repeat), {int(round(tot_rows / n_launch))} on average over the timed pooling launches (what "
                        f"roofline.algorithmic_bytes_per_launch prices), {len(last.views)}/{cfg.num_views} views kept, "
                        f"D={D}, K=96, pool_iters={args.pool_iters}, student 518->512x9->128 random-init")
        out = {
            "metric": "scenes/sec (ScanNet-val shape) + pooled-feature GB/s vs HBM peak",
            "value": round(total_steps / dt, 4), "unit": "scenes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (student convolutions and pooling: f16 hi/lo split operands, 3 MFMAs per product, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": workload,
                       "sharding": (f"{shard['policy']} assignment of {shard['scenes_total']} scenes to {world} rank(s), " if val_mode else
                                    f"1 scene per GPU x {world}, ") + "one int64 all-reduce of IoU counts",
                       "streams": len(streams), "schedule": "split" if split else "alternate",
                       "pool_mode": args.pool_mode, "pool_row_order": args.pool_row_order, "library": os.path.relpath(_lib_path(), ROOT),
                       "env_switches": {k: v for k, v in sorted(os.environ.items()) if k.startswith("GP_")}},
            "roofline": {"kernel": hp.stats["pool_kernel"] + (f" (affinity pooling, all {args.pool_iters} applications of A in one launch)"
                                                              if hp.stats["pool_kernel"] == "cs_chain_kernel" else " (affinity pooling, one application of A)"),
                         "bound": "hbm",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), **pmc_traffic(hp.stats["pool_kernel"], int(round(tot_rows / n_launch))),
                         "algorithmic_bytes_per_launch": int(round(pool_bytes_mean)), "avg_launch_ms": round(pool_ms, 5),
                         "launches": n_launch, "timed_every_nth_scene": args.time_every,
                         "launch_ms_p10": round(p10, 5), "launch_ms_p50": round(p50, 5), "launch_ms_p90": round(p90, 5),
                         "avg_launch_ms_isolated": round(pool_ms_alone, 4),
                         "isolated_passes_ms": [round(a, 4) for a in alone],
                         "frac_isolated": round(pool_bytes / (pool_ms_alone * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "gather_store_ceiling": ceiling,
                         "note": "achieved = algorithmic_bytes_per_launch / avg_launch_ms, both means over the launches of the timed "
                                 "region (HIP events around every launch); _isolated: the last scene's launches (Nv in config.workload) "
                                 "repeated with nothing else on the GPU (one warm pass, then the median of three passes)"},
            "roofline_conv": dict(conv_roofline(conv_timer) or {}, ceilings=conv_ceil) or None,
            "roofline_stages": stage_roof,
            "stages_ms_per_scene": {k: round(v / side, 3) for k, v in stages.items()},
            "stages_note": f"one-stream side pass over {side} scene(s) after the timed region",
            "student": {"pairs": pairs, "gflop_per_scene": round(flops / 1e9, 1)},
            "host_ms_per_scene": {"enqueue_total": round(host_t["step"] / max(n_local, 1) * 1e3, 3),
                                  "look_ahead_hook": round(host_t["hook"] / max(n_local, 1) * 1e3, 3),
                                  "blocked_in_readbacks": round(host_t["blocked"] / max(n_local, 1) * 1e3, 3),
                                  "readbacks": round(host_t["readbacks"] / max(n_local, 1), 2),
                                  "host_work": round((host_t["step"] - host_t["blocked"]) / max(n_local, 1) * 1e3, 3),
                                  "note": "host wall time inside step() per scene = host_work (Python + ctypes + launches: what the host must "
                                          "do per scene) + blocked_in_readbacks (the look-ahead's device -> host read-backs, which wait for "
                                          "kernels that are gated behind the start of the previous scene's student: the GPU sets their "
                                          "length, not the host); the host paces the scenes only when host_work approaches ms_per_step"},
            "iou_target_points": iou_target_points,
            "iou_intersection_points": iou_intersection_points,
        }
        if val_mode:
            out["shard"] = dict(shard, busy_s_per_rank=[round(b, 4) for b in busy_all],
                                imbalance_max_over_mean=round(max(busy_all) / (sum(busy_all) / len(busy_all)), 4))
        log(f"gpu: {out['value']} scenes/s, {out['ms_per_step']} ms/scene; pooling {pool_ms:.3f} ms/launch")
        if world == 1 and args.pool_iters != 3 and not val_mode:
            # BASELINE.json words config 1 as "affinity pooling 3 iters"; the reference code applies A 19 times
            # (affinity_module.py:1584-1587), which is what `value` is measured on.  Same scenes with 3 applications:
            try:
                hp.num_iters = 3
                for i in range(2):
                    step(i, prefetch=i + 1 < 2)
                barrier()
                t3 = time.perf_counter()
                for i in range(4):
                    step(i, prefetch=i + 1 < 4)
                barrier()
                d3 = (time.perf_counter() - t3) / 4
                out["variant_pool_iters_3"] = {"value": round(1.0 / d3, 4), "unit": "scenes/s", "ms_per_step": round(d3 * 1e3, 3), "steps": 4}
            except Exception as e:
                out["variant_pool_iters_3"] = {"value": None, "error": repr(e)}
            finally:
                hp.num_iters = args.pool_iters
        if not args.no_train and world == 1 and cfg.feat_dim == 512 and not val_mode:
            try:                                  # an extra; never lose the headline line over it
                out["training_step"] = training_step_rate(last, dev, sd)
            except Exception as e:
                out["training_step"] = {"value": None, "error": repr(e)}
        if args.api == "both" and world == 1 and not val_mode and not cfg.dense_features and args.scenes:
            try:                                  # an extra; never lose the headline line over it
                log("drop-in entry point: evaluate_scene(20-tuple of CPU tensors)")
                with torch.cuda.stream(streams[0]):
                    bts = [pl.build_scene_batch(scenes[j], rigids[j], dev, batch_views=False) for j in range(min(args.scenes, 2))]
                torch.cuda.synchronize()
                out["api_tuple"] = api_tuple_rate(bts, vlms, sd, cfg, args.pool_iters, dev, side_stream=streams[0])
            except Exception as e:
                out["api_tuple"] = {"value": None, "error": repr(e)}
        if not args.no_cpu_baseline and world == 1:
            log("cpu baseline (the oracle on the host cores)")
            try:
                out["cpu_baseline"] = cpu_baseline(scenes[0], vlm_np0, sd, rigids[0], scene_cfgs[0], args.pool_iters, args.cpu_seconds,
                                                   mode=args.cpu_sample)
            except Exception as e:  # the baseline is a reported extra; never lose the GPU line
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
