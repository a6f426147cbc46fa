#!/usr/bin/env python3
"""Timing of the two affinity kernels (block form with LDS-staged distinct rows / one wave per row) on one S-shaped
voxel set in Morton order (tuning aid).  Prints the mean size of the 16-row neighbour unions as well."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import _lib, ops, pipeline as pl, synthetic as syn  # noqa: E402

cfg = syn.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "S"]
sc = syn.make_scene(cfg, 5557)
rigid = pl.scene_rigid_transform(cfg.voxel_size, 5557)
vox = ops.voxelize(torch.from_numpy(sc.coords).cuda(), rigid)
coords = vox["coords_aug"].to(torch.int32).contiguous()
perm, rank = ops.morton_order(coords)
cs = coords[perm.long()].contiguous()
grid = ops.grid_build(cs)
K = 96
nbr = ops.knn_lattice(grid, cs, perm, K)
Nv = cs.shape[0]
E = torch.nn.functional.normalize(torch.randn(Nv, 128, device="cuda"), dim=1)
lib = _lib.load()
n16 = nbr[: Nv // 16 * 16].reshape(-1, 16 * K).cpu().numpy()
u = np.array([len(np.unique(r)) for r in n16[::37]])
print(f"Nv={Nv}  distinct neighbour rows per 16-row block: mean {u.mean():.1f}  p99 {np.percentile(u, 99):.0f}  max {u.max()}", flush=True)


def timeit(fn, n=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, knob in (("block16", 0), ("block8", 2), ("wave", 1), ("block16", 0), ("block8", 2), ("wave", 1)):
    lib.gp_debug_set(15, knob)
    ms = timeit(lambda: ops.affinity_softmax(E, nbr, 20.0))
    alg = Nv * (128 * 4 + K * 8)
    print(f"{name:8s} {ms:.3f} ms   algorithmic {alg / ms / 1e6:.0f} GB/s", flush=True)
lib.gp_debug_set(15, 0)
