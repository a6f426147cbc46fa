#!/usr/bin/env python3
"""Timing of the sparse-conv kernels on one S-shaped voxel set, with ablations (tuning aid)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import _lib, ops, pipeline as pl, synthetic as syn  # noqa: E402
import dataclasses
cfg = dataclasses.replace(syn.CONFIGS["S"], num_views=1)
sc = syn.make_scene(cfg, 5557)
rigid = pl.scene_rigid_transform(cfg.voxel_size, 5557)
vox = ops.voxelize(torch.from_numpy(sc.coords).cuda(), rigid)
coords = vox["coords_aug"].to(torch.int32).contiguous()
perm, rank = ops.morton_order(coords)
cs = coords[perm.long()].contiguous()
grid = ops.grid_build(cs)
nm = ops.kernel_map_build(grid, cs)
import sys as _s
CHUNK = int(os.environ.get('GP_CHUNK', '0'))
pairs = ops.conv_pairs_build(nm, CHUNK if CHUNK > 0 else None)
print('chunks', pairs.num_chunks, 'max chunk pairs', pairs.max_chunk_pairs, flush=True)
Nv = cs.shape[0]
print("Nv", Nv, "pairs", pairs.num_pairs, flush=True)
X = torch.randn(Nv, 512, device="cuda")
W = torch.randn(27, 512, 512, device="cuda") * 0.01
hi, lo = ops.conv_weights_split(W, 64.0)
sc_ = torch.ones(512, device="cuda"); sh = torch.zeros(512, device="cuda")
lib = _lib.load()
flops = 2.0 * pairs.num_pairs * 512 * 512


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


ab = lambda v: lib.gp_debug_set(3, v.value)
for name, v in (("full", 0),):
    if ab is not None:
        ab(ctypes.c_int(v))
    t = timeit(lambda: ops.sparse_conv_f16x3(X, pairs, hi, lo, sc_, sh, relu=True))
    print(f"{name:36s} {t:7.3f} ms  ({flops / t / 1e9:7.1f} TFLOP/s effective)", flush=True)
if ab is not None:
    ab(ctypes.c_int(0))
xs = ops.split_f16(X)
ys = tuple(torch.empty((Nv, 512), dtype=torch.float16, device="cuda") for _ in range(2))
t = timeit(lambda: ops.sparse_conv_f16x3(None, pairs, hi, lo, sc_, sh, relu=True, x_split=xs, out_split=ys))
print(f"{'LDS-DMA path (pre-split in/out)':36s} {t:7.3f} ms  ({flops / t / 1e9:7.1f} TFLOP/s effective)", flush=True)
t = timeit(lambda: ops.sparse_conv_f16x3(None, pairs, hi, lo, sc_, sh, relu=True, x_split=xs))
print(f"{'LDS-DMA path (no split output)':36s} {t:7.3f} ms  ({flops / t / 1e9:7.1f} TFLOP/s effective)", flush=True)
for name, v in (("DMA: full", 0), ("DMA: no MFMA", 2), ("DMA: no MFMA, no epilogue", 10), ("DMA: no epilogue", 8)):
    lib.gp_debug_set(3, v)
    t = timeit(lambda: ops.sparse_conv_f16x3(None, pairs, hi, lo, sc_, sh, relu=True, x_split=xs))
    print(f"{name:36s} {t:7.3f} ms", flush=True)
lib.gp_debug_set(3, 0)
if os.environ.get("GP_SKIP_V1"):
    sys.exit(0)
t = timeit(lambda: ops.sparse_conv(X, nm, W, sc_, sh, relu=True), 2)
print(f"{'v1 fp32 MFMA kernel':36s} {t:7.3f} ms  ({flops / t / 1e9:7.1f} TFLOP/s)")
