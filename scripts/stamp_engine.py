#!/usr/bin/env python3
"""Time stamps of the producer/consumer pooling engine's consumer waves: cycles polling for a full slot, cycles in operand
reads + MFMAs, cycles in the epilogue.  usage: stamp_engine.py [ablate bits]"""
import dataclasses, os, sys
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
K, D = 96, 512
nbr = ops.knn_lattice(grid, cs, perm, K)
Nv = cs.shape[0]
w = ops.affinity_softmax(torch.nn.functional.normalize(torch.randn(Nv, 128, device="cuda"), dim=1), nbr, 20.0)
X = torch.randn(Nv, 544, device="cuda")
lib = _lib.load()
op = ops.pool_cs_build(nbr, w)
xs = ops.split_f16(X, D)
ys = tuple(torch.empty((Nv, D), dtype=torch.float16, device="cuda") for _ in range(2))
buf = torch.zeros(256 * 4 * 10, dtype=torch.int64, device="cuda")
for _ in range(3):
    ops.pool_cs_apply(xs, op, D, out_split=ys, engine=True)
torch.cuda.synchronize()
lib.gp_debug_set(4, ABL)
lib.gp_debug_ptr(0, buf.data_ptr(), buf.numel() * 8)
ops.pool_cs_apply(xs, op, D, out_split=ys, engine=True)
torch.cuda.synchronize()
buf.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.pool_cs_apply(xs, op, D, out_split=ys, engine=True); e1.record()
torch.cuda.synchronize()
lib.gp_debug_ptr(0, None, 0); lib.gp_debug_set(4, 0)
s = buf.cpu().numpy().reshape(-1, 4, 10).astype(np.float64)
tot, work, poll, epi, steps = s[..., 7], s[..., 3], s[..., 4], s[..., 6], s[..., 8]
print(f"ablate={ABL} launch {e0.elapsed_time(e1) * 1e3:.1f} us (stamped); consumer waves {tot.size}; steps per wave {steps.mean():.1f} (min {steps.min():.0f}, max {steps.max():.0f})")
print(f"  wave lifetime {tot.mean():.0f} cycles (min {tot.min():.0f}, max {tot.max():.0f});  per step: poll {poll.sum() / steps.sum():.0f}  reads+MFMA {work.sum() / steps.sum():.0f} cycles;"
      f"  epilogue {100 * epi.sum() / tot.sum():.1f} % of the lifetime, poll {100 * poll.sum() / tot.sum():.1f} %, work {100 * work.sum() / tot.sum():.1f} %")
