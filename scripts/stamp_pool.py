#!/usr/bin/env python3
"""Where does a workgroup of the matrix-core pooling kernel spend its time?  Runs the stamped instantiation
(s_memtime around the prologue, every step's work and hand-over wait, the epilogue) on one S-shaped voxel set and
prints the per-wave breakdown.  Tuning aid; the stamped kernel computes the same results.
usage: stamp_pool.py [block_rows=64] [ablate=0] [num_points]"""
import dataclasses
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import _lib, ops, pipeline as pl, synthetic as syn  # noqa: E402

BR = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ABL = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cfg = dataclasses.replace(syn.CONFIGS["S"], num_views=1, num_points=int(sys.argv[3]) if len(sys.argv) > 3 else 150_000)
sc = syn.make_scene(cfg, 5557)
rigid = pl.scene_rigid_transform(cfg.voxel_size, 5557)
vox = ops.voxelize(torch.from_numpy(sc.coords).cuda(), rigid)
coords = vox["coords_aug"].to(torch.int32).contiguous()
perm, rank = ops.morton_order(coords)
cs = coords[perm.long()].contiguous()
grid = ops.grid_build(cs)
K, D = 96, 512
nbr = ops.knn_lattice(grid, cs, perm, K)
Nv = cs.shape[0]
E = torch.nn.functional.normalize(torch.randn(Nv, 128, device="cuda"), dim=1)
w = ops.affinity_softmax(E, nbr, 20.0)
X = torch.randn(Nv, 544, device="cuda")
lib = _lib.load()
CS = BR == 0                                       # block_rows 0 selects the column-sliced kernel (128-row blocks)
if CS:
    BR = 128
op = ops.pool_cs_build(nbr, w) if CS else ops.pool_mfma_build(nbr, w, BR)
apply = ops.pool_cs_apply if CS else ops.pool_mfma_apply
xs = ops.split_f16(X, D)
ys = tuple(torch.empty((Nv, D), dtype=torch.float16, device="cuda") for _ in range(2))
NW = 4 if BR == 64 else 8
nq = 4 if BR == 64 else 2
nb = (Nv + BR - 1) // BR
nwg = ((nb * nq + 7) // 8) * 8 + 64          # (+ slack: the cs operator may use a smaller block height / the longest-first grid)
buf = torch.zeros(nwg * NW * 10, dtype=torch.int64, device="cuda")
for _ in range(3):
    apply(xs, op, D, out_split=ys)
torch.cuda.synchronize()
lib.gp_debug_set(4, ABL)
lib.gp_debug_ptr(0, buf.data_ptr(), buf.numel() * 8)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
apply(xs, op, D, out_split=ys)
torch.cuda.synchronize()
buf.zero_()
e0.record()
apply(xs, op, D, out_split=ys)
e1.record()
torch.cuda.synchronize()
lib.gp_debug_ptr(0, None, 0)
lib.gp_debug_set(4, 0)
ms = e0.elapsed_time(e1)
s = buf.cpu().numpy().reshape(-1, NW, 10)
live = s[:, 0, 8] > 0
s = s[live]
r0 = s[:, :, 0].astype(np.float64)
t_first = r0.min()
dur_us = s[:, :, 1] / 100.0
print(f"BR={BR} ablate={ABL} Nv={Nv} launch {ms * 1e3:.1f} us (stamped)  workgroups {len(s)}  waves/WG {NW}")
print(f"span of wave lifetimes: first start -> last end = {((r0 + s[:, :, 1]).max() - t_first) / 100:.1f} us;"
      f" wave lifetime mean {dur_us.mean():.2f} us (p10 {np.percentile(dur_us, 10):.2f}, p90 {np.percentile(dur_us, 90):.2f})")
tot = s[:, :, 7].astype(np.float64)
clk = tot / (s[:, :, 1] / 100.0) / 1e3
print(f"in-kernel clock (cycles / us of lifetime): {np.median(clk):.3f} GHz")
names = {2: "prologue", 3: "step work (reads+issue+MFMA)", 4: "hand-over wait", 5: "  of work: ids/weights read + DMA issue", 6: "epilogue"}
for k, nm in names.items():
    v = s[:, :, k].astype(np.float64)
    print(f"  {nm:42s} {100 * v.sum() / tot.sum():5.1f} % of wave cycles; mean per wave {v.mean():8.0f} cycles")
steps = s[:, :, 8].astype(np.float64)
print(f"  steps per workgroup mean {steps.mean():.2f};  per step: work {s[:, :, 3].sum() / steps.sum():.0f} cycles, wait {s[:, :, 4].sum() / steps.sum():.0f} cycles"
      f" (issue part {s[:, :, 5].sum() / steps.sum():.0f})")
# concurrency over time: how many workgroups are alive
st = (r0[:, 0] - t_first) / 100
en = st + dur_us.max(axis=1)
T = en.max()
ts = np.linspace(0, T, 41)
alive = [(int(((st <= t) & (en > t)).sum())) for t in ts]
print("  workgroups alive at 40 points of the launch:", alive)
xcc = s[:, 0, 9]
print("  workgroups per XCC:", np.bincount(xcc.astype(np.int64), minlength=8).tolist())
# per-wave asymmetry: which wave waits
for wv in range(NW):
    print(f"    wave {wv}: work {s[:, wv, 3].mean():8.0f}  wait {s[:, wv, 4].mean():8.0f}  prologue {s[:, wv, 2].mean():7.0f}  epilogue {s[:, wv, 6].mean():7.0f}")
