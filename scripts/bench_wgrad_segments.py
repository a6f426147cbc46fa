#!/usr/bin/env python3
"""The weight-gradient kernel under different segment lengths (steps of 32 pairs per workgroup): a layer's time follows the ROUNDS of
workgroups its launch needs (segments x 4 tiles over the CUs) times the segment length.  Training-shaped voxel set (S scene, 4096 anchors).
usage: bench_wgrad_segments.py [steps_per_segment ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import ops, pipeline as pl, synthetic as syn  # noqa: E402

SPS = [int(a) for a in sys.argv[1:]] or [64, 96, 112, 128, 144, 160, 176, 192, 224, 256]
cfg = syn.CONFIGS["S"]
scene = syn.make_scene(cfg, 5557)
rigid = pl.scene_rigid_transform(cfg.voxel_size, 5557)
batch = pl.build_scene_batch(pl.upload_scene(scene, "cuda"), rigid, "cuda")
N = batch.scene_coords.shape[0]
g = torch.Generator(device="cuda").manual_seed(1)
idx = torch.unique(torch.randint(0, N, (4096 * 65,), device="cuda", generator=g))           # ~ the sampler's point set
vox = torch.unique(batch.scene_inds_reconstruct[idx])
cs_ref = batch.scene_coords_3d[vox].floor().to(torch.int32).contiguous()
perm, rank = ops.morton_order(cs_ref)
cs = cs_ref[perm.long()].contiguous()
nm = ops.kernel_map_build(ops.grid_build(cs), cs)
Nv = cs.shape[0]
kk, rr, rin, counts = ops.kernel_map_pairs(nm)
print(f"Nv {Nv}, pairs {sum(counts)}, per offset: centre {counts[13]}, others {min(counts)}..{sorted(counts)[-2]}")
x = torch.randn(Nv, 512, device="cuda", generator=g)
dy = torch.zeros(Nv + 1, 512, device="cuda")
dy[:Nv] = torch.randn(Nv, 512, device="cuda", generator=g)
xs, ys = ops.split_f16(x), ops.split_f16(dy)
ncu = torch.cuda.get_device_properties(0).multi_processor_count
ref = None
for sps in SPS:
    plan = ops.wgrad_plan_from_pairs(kk, rr, rin, counts, Nv, steps_per_segment=sps)
    run = lambda: ops.conv_wgrad_f16x3(xs, ys, plan, 512, 512, 512)
    for _ in range(3):
        dw = run()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    wgs = plan.num_segments * 4
    if ref is None:
        ref = dw.clone()
    err = float((dw - ref).abs().max() / ref.abs().max())
    print(f"steps/segment {sps:4d}: {plan.num_segments:4d} segments x 4 tiles = {wgs:5d} workgroups = {wgs / ncu:5.2f} rounds of {ncu} CUs; "
          f"{np.median(ts):.3f} ms (min {min(ts):.3f}); max |dW - dW(first)| / max {err:.1e}")
