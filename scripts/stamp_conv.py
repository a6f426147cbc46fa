#!/usr/bin/env python3
"""Where a conv_phase1 tile's time goes: in-kernel stamps of the stamped twin (gp_debug_ptr(1, buf, bytes)), the clock the
chip holds in the loop, and the same layer on all-zero operands (MI355X_MICROARCH.md "DVFS give-back": cycles are data-independent,
the clock is not).  usage: stamp_conv.py [knob-3 mask]"""
import os
import sys
import dataclasses

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import _lib, ops, pipeline as pl, synthetic as syn  # noqa: E402

ABL = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cfg = dataclasses.replace(syn.CONFIGS["S"], num_views=1)
sc = syn.make_scene(cfg, 5557)
rigid = pl.scene_rigid_transform(cfg.voxel_size, 5557)
vox = ops.voxelize(torch.from_numpy(sc.coords).cuda(), rigid)
coords = vox["coords_aug"].to(torch.int32).contiguous()
perm, rank = ops.morton_order(coords)
cs = coords[perm.long()].contiguous()
grid = ops.grid_build(cs)
nm = ops.kernel_map_build(grid, cs)
pairs = ops.conv_pairs_build(nm)
Nv = cs.shape[0]
print(f"Nv {Nv} pairs {pairs.num_pairs} chunks {pairs.num_chunks}", flush=True)
lib = _lib.load()
sc_ = torch.ones(512, device="cuda"); sh = torch.zeros(512, device="cuda")


def timeit(fn, n=8):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def stamped(label, X, W, pairs, abl):
    hi, lo = ops.conv_weights_split(W, 64.0)
    if os.environ.get("GP_STAMP_PLANES"):                 # the round 1-5a operand form: separate hi / lo planes
        xs = ops.split_f16(X)
        ys = tuple(torch.empty((Nv, 512), dtype=torch.float16, device="cuda") for _ in range(2))
    else:                                                 # what the student's layers hand each other now: interleaved rows
        xs = (ops.interleave_planes(*ops.split_f16(X)), None)
        ys = (torch.empty((Nv, 1024), dtype=torch.float16, device="cuda"), None)
    run = lambda: ops.sparse_conv_f16x3(None, pairs, hi, lo, sc_, sh, relu=True, x_split=xs, out_split=ys, want_f32=False)
    assert lib.gp_debug_set(3, abl) == 0, f"knob 3 mask {abl} is not in the table"
    for _ in range(40):                                   # ~0.1 s of back-to-back launches: the clock settles
        run()
    t = timeit(run, 16)
    print(f"{label:18s} knob3={abl:3d} chunks {pairs.num_chunks:2d}: layer {t:7.3f} ms", flush=True)
    nblk = 1 << 16
    buf = torch.zeros(nblk * 16, dtype=torch.int64, device="cuda")
    assert lib.gp_debug_ptr(1, buf.data_ptr(), buf.numel() * 8) == 0
    run(); torch.cuda.synchronize()
    buf.zero_()
    run(); torch.cuda.synchronize()
    lib.gp_debug_ptr(1, None, 0); lib.gp_debug_set(3, 0)
    s = buf.cpu().numpy().reshape(-1, 16).astype(np.float64)
    s = s[s[:, 6] > 0]                                     # workgroups of the LAST chunk launches that owned a tile
    real_us, pro, loop, iss, drain, tot = s[:, 1] / 100.0, s[:, 2], s[:, 3], s[:, 4], s[:, 5], s[:, 6]
    clk = tot / (s[:, 1] / 100.0) / 1e3                    # GHz: shader cycles per 100-MHz real-time tick
    cntv = (s[:, 7].astype(np.int64) >> 8)
    full = cntv == 256
    f = lambda a: f"{np.mean(a[full]):8.0f}"
    print(f"  stamped tiles {len(s)} ({int(full.sum())} full); in-kernel clock {np.median(clk):.2f} GHz; tile lifetime {np.median(real_us):.1f} us "
          f"(p10 {np.percentile(real_us, 10):.1f}, p90 {np.percentile(real_us, 90):.1f})")
    print(f"  cycles per full tile: prologue {f(pro)}  K loop {f(loop)} ({np.mean(loop[full]) / 16:.0f} per step; MFMA issue alone = 3072)  "
          f"store issue {f(iss)}  store drain {f(drain)}  total {f(tot)}")
    print(f"  inside the K loop, wave 0 (stamps cost ~10 %): DMA issue {np.mean(s[full, 8]) / 16:6.0f} cycles per step, end-of-step s_waitcnt "
          f"{np.mean(s[full, 9]) / 16:6.0f} + s_barrier {np.mean(s[full, 10]) / 16:6.0f}, reads + MFMA "
          f"{(np.mean(loop[full]) - np.mean(s[full, 8]) - np.mean(s[full, 9]) - np.mean(s[full, 10])) / 16:6.0f}; wave 4 (same SIMD): DMA issue "
          f"{np.mean(s[full, 12]) / 16:6.0f}, s_waitcnt {np.mean(s[full, 13]) / 16:6.0f}, s_barrier {np.mean(s[full, 14]) / 16:6.0f}", flush=True)
    return ys


Xr, Wr = torch.randn(Nv, 512, device="cuda"), torch.randn(27, 512, 512, device="cuda") * 0.01
ref = None
for chunk_rows in (8192, 16384):
    p = ops.conv_pairs_build(nm, chunk_rows)
    for abl in (ABL,):                                     # (the split-role loop, once knob 3 bit 64, was measured and removed: DESIGN 5.1)
        ys = stamped("random operands", Xr, Wr, p, abl)
        outs = [t for t in ys if t is not None]
        if ref is None:
            ref = [t.clone() for t in outs]
        print("  bits equal to the first run:", all(bool(torch.equal(a, b)) for a, b in zip(outs, ref)), flush=True)
stamped("all-zero operands", torch.zeros(Nv, 512, device="cuda"), torch.zeros(27, 512, 512, device="cuda"), ops.conv_pairs_build(nm, 8192), 0)
