#!/usr/bin/env python3
"""cs_pool_kernel on the S scene: block height (rows_per_block).  19 chained applications (the product's ping-pong of split
planes), median of 5 chains.  (Round 4 also ran it with a longest-tile-first launch order and with a ten-group instantiation of
the kernel for blocks of up to 152 rows -- profiles/r04_pool_block_height.log; neither is in the tree: neither is faster.)"""
import dataclasses
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import _lib, ops, pipeline as pl, synthetic as syn  # noqa: E402

D, K, T = 512, 96, 19
lib = _lib.load()
for seed in (5557, 6557):
    cfg = dataclasses.replace(syn.CONFIGS["S"], num_views=1)
    sc = syn.make_scene(cfg, seed)
    rigid = pl.scene_rigid_transform(cfg.voxel_size, seed)
    vox = ops.voxelize(torch.from_numpy(sc.coords).cuda(), rigid)
    coords = vox["coords_aug"].to(torch.int32).contiguous()
    perm, rank = ops.morton_order(coords)
    cs = coords[perm.long()].contiguous()
    grid = ops.grid_build(cs)
    nbr = ops.knn_lattice(grid, cs, perm, K)
    Nv = cs.shape[0]
    w = ops.affinity_softmax(torch.nn.functional.normalize(torch.randn(Nv, 128, device="cuda"), dim=1), nbr, 20.0)
    X = torch.randn(Nv, 544, device="cuda")
    print(f"seed {seed}: Nv {Nv}", flush=True)
    ref = None
    for rpb in (128, 124, 120, 117, 112, 104, 96):
        for lf in (False,):
            op = ops.pool_cs_build(nbr, w, rows_per_block=rpb)
            steps = ((op.bu_off[1:] - op.bu_off[:-1]) // 32).float()
            sp = [ops.split_f16(X, D), tuple(torch.empty((Nv, D), dtype=torch.float16, device="cuda") for _ in range(2))]
            out = torch.empty((Nv, D), device="cuda")

            def chain():
                src = sp[0]
                for t in range(T):
                    last = t == T - 1
                    dst = None if last else sp[(t + 1) % 2]
                    ops.pool_cs_apply(src, op, D, out_split=dst, out_f32=out if last else None)
                    src = dst if dst is not None else src
            sp0 = (sp[0][0].clone(), sp[0][1].clone())
            for _ in range(3):
                sp[0][0].copy_(sp0[0]); sp[0][1].copy_(sp0[1])
                chain()
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                sp[0][0].copy_(sp0[0]); sp[0][1].copy_(sp0[1])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); chain(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / T)
            t = float(np.median(ts))
            if ref is None:
                ref = out.clone()
            nb = op.bu_off.shape[0] - 1
            print(f"  rows/block {rpb:3d} {'longest first' if lf else 'memory order '}: tiles {2 * nb:5d} = {2 * nb / 256:5.2f} rounds, steps/tile "
                  f"{steps.mean():5.2f} (min {steps.min():.0f} max {steps.max():.0f} sd {steps.std():4.2f}), union rows/row {op.total / Nv:4.2f}: "
                  f"{t:7.4f} ms/application = {Nv * 4864 / t / 1e6 / 80:5.2f} % of 8 TB/s   max |diff| vs first {float((out - ref).abs().max()):.2e}", flush=True)
