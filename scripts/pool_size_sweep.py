#!/usr/bin/env python3
"""Does the pooling kernel's time per row depend on whether its working set fits the 256 MiB Infinity Cache?
19 chained applications (the product's ping-pong of split planes) of cs_pool_kernel on scenes of growing size:
working set per application = X planes read (Nv x 2 KB) + planes written (Nv x 2 KB) + operator (~1.2 KB per row)."""
import dataclasses
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import ops, pipeline as pl, synthetic as syn  # noqa: E402

D, K, T = 512, 96, 19
for npts in [int(a) for a in sys.argv[1:]] or [30_000, 45_000, 60_000, 80_000, 110_000, 150_000, 220_000, 300_000]:
    cfg = dataclasses.replace(syn.CONFIGS["S"], num_views=1, num_points=npts)
    sc = syn.make_scene(cfg, 5557)
    rigid = pl.scene_rigid_transform(cfg.voxel_size, 5557)
    vox = ops.voxelize(torch.from_numpy(sc.coords).cuda(), rigid)
    coords = vox["coords_aug"].to(torch.int32).contiguous()
    perm, rank = ops.morton_order(coords)
    cs = coords[perm.long()].contiguous()
    grid = ops.grid_build(cs)
    nbr = ops.knn_lattice(grid, cs, perm, K)
    Nv = cs.shape[0]
    w = ops.affinity_softmax(torch.nn.functional.normalize(torch.randn(Nv, 128, device="cuda"), dim=1), nbr, 20.0)
    X = torch.randn(Nv, 544, device="cuda")
    op = ops.pool_cs_build(nbr, w)
    sp = [ops.split_f16(X, D), tuple(torch.empty((Nv, D), dtype=torch.float16, device="cuda") for _ in range(2))]

    def chain():
        src = sp[0]
        for t in range(T):
            dst = sp[(t + 1) % 2]
            ops.pool_cs_apply(src, op, D, out_split=dst)
            src = dst
    for _ in range(3):
        chain()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); chain(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / T)
    t = float(np.median(ts))
    ws = Nv * 4096 + op.total * (4 + 0.5 * 2 * 2 * 16)       # planes in + out, ids + the non-empty half of the hi/lo fragments (16 B per union row and group... approx.)
    print(f"N {npts:7d}  Nv {Nv:7d}  union rows/row {op.total / Nv:5.2f}  ms/application {t:7.4f}  ns/row {t * 1e6 / Nv:6.3f}  "
          f"algorithmic {Nv * 4864 / t / 1e6:7.1f} GB/s = {Nv * 4864 / t / 1e6 / 80:5.1f} % of 8 TB/s   planes in+out {Nv * 4096 / 1e6:6.1f} MB", flush=True)
