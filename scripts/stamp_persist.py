#!/usr/bin/env python3
"""Time stamps of cs_pool_persist_kernel (wave level): cycles in the step loop, in {next tile's first stages + epilogue}, at the
tile-entry hand-over; tiles per workgroup; end times.  usage: stamp_persist.py [knob 13 = static lists | 14 = claims] [ablate]"""
import dataclasses, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import _lib, ops, pipeline as pl, synthetic as syn  # noqa: E402

KNOB = int(sys.argv[1]) if len(sys.argv) > 1 else 14
ABL = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cfg = dataclasses.replace(syn.CONFIGS["S"], num_views=1)
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
w = ops.affinity_softmax(torch.nn.functional.normalize(torch.randn(Nv, 128, device="cuda"), dim=1), nbr, 20.0)
X = torch.randn(Nv, 544, device="cuda")
lib = _lib.load()
op = ops.pool_cs_build(nbr, w)
xs = ops.split_f16(X, D)
ys = tuple(torch.empty((Nv, D), dtype=torch.float16, device="cuda") for _ in range(2))
buf = torch.zeros(256 * 8 * 10, dtype=torch.int64, device="cuda")
lib.gp_debug_ptr(1, op.queue.data_ptr())
lib.gp_debug_set(11, KNOB)
for _ in range(3):
    ops.pool_cs_apply(xs, op, D, out_split=ys)
torch.cuda.synchronize()
lib.gp_debug_set(4, ABL)
lib.gp_debug_ptr(0, buf.data_ptr())
ops.pool_cs_apply(xs, op, D, out_split=ys)
torch.cuda.synchronize()
buf.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.pool_cs_apply(xs, op, D, out_split=ys); e1.record()
torch.cuda.synchronize()
lib.gp_debug_ptr(0, None); lib.gp_debug_set(4, 0); lib.gp_debug_set(11, 0)
s = buf.cpu().numpy().reshape(256, 8, 10).astype(np.float64)
loop, epi, entry, tot, tiles, steps, tend = (s[..., i] for i in range(7))
print(f"knob {KNOB} ablate {ABL}: launch {e0.elapsed_time(e1) * 1e3:.1f} us (stamped)")
print(f"  wave lifetime {tot.mean():.0f} cycles (min {tot.min():.0f}, max {tot.max():.0f}); tiles per workgroup {tiles[:, 0].mean():.2f} (min {tiles[:, 0].min():.0f}, max {tiles[:, 0].max():.0f})")
print(f"  loop {100 * loop.sum() / tot.sum():.1f} %  (per step {loop.sum() / steps.sum():.0f} cycles)   next-stages + epilogue {100 * epi.sum() / tot.sum():.1f} % ({epi.sum() / tiles.sum():.0f} per tile)"
      f"   tile-entry hand-over {100 * entry.sum() / tot.sum():.1f} % ({entry.sum() / tiles.sum():.0f} per tile)")
te = (tend[:, 0] - tend[:, 0].min()) / 100.0
print(f"  workgroup end times (us after the first to end): p50 {np.percentile(te, 50):.1f}  p90 {np.percentile(te, 90):.1f}  max {te.max():.1f}")
