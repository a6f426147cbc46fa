#!/usr/bin/env python3
"""Timing of the lattice kNN (K = 96) on one S-shaped voxel set (tuning aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import ops, pipeline as pl, synthetic as syn  # noqa: E402

cfg = syn.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "S"]
sc = syn.make_scene(cfg, 5557)
vox = ops.voxelize(torch.from_numpy(sc.coords).cuda(), pl.scene_rigid_transform(cfg.voxel_size, 5557))
coords = vox["coords_aug"].to(torch.int32).contiguous()
perm, rank = ops.morton_order(coords)
cs = coords[perm.long()].contiguous()
grid = ops.grid_build(cs)
nbr = ops.knn_lattice(grid, cs, perm, 96)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.knn_lattice(grid, cs, perm, 96)
e1.record()
torch.cuda.synchronize()
print(f"Nv={cs.shape[0]}  knn_lattice K=96: {e0.elapsed_time(e1) / 10:.3f} ms  checksum {int(nbr.long().sum())}")
