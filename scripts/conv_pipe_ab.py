#!/usr/bin/env python3
"""The 512->512 layer on ONE box in the product operand form (interleaved row-scaled rows, step-blocked weights, balanced chunks) under
several values of knob 3, interleaved over several rounds so that box drift shows; bit equality of the outputs; then the stamped twin
(where a step's cycles go: DMA issue, `s_waitcnt`, `s_barrier` apart, for wave 0 and its SIMD partner wave 4).
knob 3: 0 product kernels, 512 the tuning twin with nothing switched, 256 the twin without the centre offset's partial rows (the price of
a centre-offset fold), 2 no MFMA, ...  (Round 6's PIPE 1-3 schedules -- knob values 64 / 128 / 192 -- exist in commit fc740de only:
profiles/r06_conv_schedules.log.)
usage: conv_pipe_ab.py [knob3 values ...]        default: 0 512 256"""
import os
import sys
import dataclasses

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import _lib, ops, pipeline as pl, synthetic as syn  # noqa: E402

KNOBS = [int(a) for a in sys.argv[1:]] or [0, 512, 256]
ROUNDS = int(os.environ.get("GP_AB_ROUNDS", "3"))
STAMPS = os.environ.get("GP_AB_STAMPS", "1") != "0"
cfg = dataclasses.replace(syn.CONFIGS["S"], num_views=1)
SEED = int(os.environ.get("GP_SCENE_SEED", "5557"))
sc = syn.make_scene(cfg, SEED)
rigid = pl.scene_rigid_transform(cfg.voxel_size, SEED)
vox = ops.voxelize(torch.from_numpy(sc.coords).cuda(), rigid)
coords = vox["coords_aug"].to(torch.int32).contiguous()
perm, rank = ops.morton_order(coords)
cs = coords[perm.long()].contiguous()
nm = ops.kernel_map_build(ops.grid_build(cs), cs)
Nv = cs.shape[0]
pairs = ops.conv_pairs_build(nm)
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(3)
X = torch.relu(torch.randn(Nv, 512, device="cuda", generator=g))
W = torch.randn(27, 512, 512, device="cuda", generator=g) * 0.01
hi, lo = ops.conv_weights_split(W, 64.0)
xs = ops.split_f16(X, per_row=True, interleaved=True)
sc_, sh = torch.ones(512, device="cuda"), torch.zeros(512, device="cuda")
print(f"seed {SEED}: Nv {Nv}, pairs {pairs.num_pairs}, {pairs.num_chunks} chunk launches", flush=True)


def layer(knob):
    ys = (torch.empty((Nv, 1024), dtype=torch.float16, device="cuda"), None, torch.empty(Nv, device="cuda"))
    assert lib.gp_debug_set(3, knob) == 0, f"knob 3 = {knob} is not in the table"
    run = lambda: ops.sparse_conv_f16x3(None, pairs, hi, lo, sc_, sh, relu=True, x_split=xs[:2], x_row_inv=xs[2], out_split=ys[:2],
                                        out_row_inv=ys[2], want_f32=False)
    return run, ys


def timeit(run, warm=30, reps=5, n=12):
    for _ in range(warm):
        run()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return float(np.median(ts)), min(ts), max(ts)


ref = None
res = {k: [] for k in KNOBS}
for r in range(ROUNDS):
    for k in KNOBS:
        run, ys = layer(k)
        med, lo_, hi_ = timeit(run)
        lib.gp_debug_set(3, 0)
        res[k].append(med)
        outs = [ys[0].clone(), ys[2].clone()]
        same = "-"
        if ref is None:
            ref = outs
        elif not (k & (2 | 8 | 256)):
            same = str(all(bool(torch.equal(a, b)) for a, b in zip(outs, ref)))
        print(f"round {r} knob3={k:4d}: layer {med:6.3f} ms (min {lo_:6.3f}, max {hi_:6.3f})  bits equal to the first run: {same}", flush=True)
print("medians over the rounds: " + "   ".join(f"knob3={k}: {np.median(v):.3f} ms" for k, v in res.items()), flush=True)

if STAMPS:
    for k in [0]:
        run, ys = layer(k)
        for _ in range(30):
            run()
        nblk = 1 << 15
        buf = torch.zeros(nblk * 16, dtype=torch.int64, device="cuda")
        assert lib.gp_debug_ptr(1, buf.data_ptr(), buf.numel() * 8) == 0
        for _ in range(10):
            run()
        t0 = timeit(run, warm=0, reps=1, n=8)[0]
        torch.cuda.synchronize()
        buf.zero_()
        run()
        torch.cuda.synchronize()
        lib.gp_debug_ptr(1, None, 0)
        lib.gp_debug_set(3, 0)
        s = buf.cpu().numpy().reshape(-1, 16)
        s = s[s[:, 6] > 0]
        f = s.astype(np.float64)
        clk = f[:, 6] / (f[:, 1] / 100.0) / 1e3
        full = (s[:, 7] >> 8) == 256
        m = lambda c: float(np.mean(f[full, c]))
        steps = 16.0
        hw0, hw4 = s[full, 11], s[full, 15]
        simd = lambda h: (h >> 4) & 3
        print(f"stamped twin, knob3={k}: layer {t0:.3f} ms; {len(s)} tiles ({int(full.sum())} full) of the last chunk; clock {np.median(clk):.2f} GHz; "
              f"tile {m(6):.0f} cycles = prologue {m(2):.0f} + K loop {m(3):.0f} ({m(3) / steps:.0f} per step) + store issue {m(4):.0f} + drain {m(5):.0f}")
        print(f"    per step, wave 0: DMA issue {m(8) / steps:5.0f}  s_waitcnt {m(9) / steps:5.0f}  s_barrier {m(10) / steps:5.0f}  rest (reads + MFMA) "
              f"{(m(3) - m(8) - m(9) - m(10)) / steps:5.0f}   |   wave 4: DMA issue {m(12) / steps:5.0f}  s_waitcnt {m(13) / steps:5.0f}  s_barrier {m(14) / steps:5.0f}")
        print(f"    waves 0 and 4 on the same SIMD (HW_ID bits 5:4) in {100.0 * float(np.mean(simd(hw0) == simd(hw4))):.1f} % of the tiles", flush=True)
