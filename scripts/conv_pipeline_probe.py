#!/usr/bin/env python3
"""Experiment: phase 2 of chunk c on a helper stream beside phase 1 of chunk c + 1 (two partial slots, gp_debug_ptr(2, slot, bytes)).
One 512->512 layer of the S scene, per chunk height, serial against pipelined; the outputs must keep their bits.
NOTE: the library side of this experiment (the second slot honoured by gp_sparse_conv_f16x3) was measured and removed again
(profiles/r04_conv_pipelined_chunks.log: slower at every chunk height); on the current library both columns are the serial form."""
import os
import sys
import dataclasses

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import _lib, ops, pipeline as pl, synthetic as syn  # noqa: E402

cfg = dataclasses.replace(syn.CONFIGS["S"], num_views=1)
sc = syn.make_scene(cfg, 5557)
rigid = pl.scene_rigid_transform(cfg.voxel_size, 5557)
vox = ops.voxelize(torch.from_numpy(sc.coords).cuda(), rigid)
coords = vox["coords_aug"].to(torch.int32).contiguous()
perm, rank = ops.morton_order(coords)
cs = coords[perm.long()].contiguous()
grid = ops.grid_build(cs)
nm = ops.kernel_map_build(grid, cs)
Nv = cs.shape[0]
lib = _lib.load()
X, W = torch.randn(Nv, 512, device="cuda"), torch.randn(27, 512, 512, device="cuda") * 0.01
hi, lo = ops.conv_weights_split(W, 64.0)
xs = ops.split_f16(X, per_row=True)
sc_, sh = torch.ones(512, device="cuda"), torch.zeros(512, device="cuda")


def timeit(fn, n=12):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for chunk_rows in (8192, 6144, 4096, 16384):
    pairs = ops.conv_pairs_build(nm, chunk_rows)
    ys = tuple(torch.empty((Nv, 512), dtype=torch.float16, device="cuda") for _ in range(2)) + (torch.empty(Nv, device="cuda"),)
    run = lambda: ops.sparse_conv_f16x3(None, pairs, hi, lo, sc_, sh, relu=True, x_split=xs[:2], x_row_inv=xs[2], out_split=ys[:2],
                                        out_row_inv=ys[2], want_f32=False)
    t0 = timeit(run)
    ref = (ys[0].clone(), ys[1].clone(), ys[2].clone())
    slot = torch.empty((pairs.max_chunk_pairs, 512), dtype=torch.float32, device="cuda")
    assert lib.gp_debug_ptr(2, slot.data_ptr(), slot.numel() * 4) == 0
    try:
        t1 = timeit(run)
        same = all(torch.equal(a, b) for a, b in zip(ys, ref))
    finally:
        lib.gp_debug_ptr(2, None, 0)
    t2 = timeit(run)
    print(f"chunk {chunk_rows:6d} rows ({pairs.num_chunks:2d} chunks, {slot.numel() * 4 / 1e6:5.0f} MB per slot): serial {t0:6.3f} ms, pipelined {t1:6.3f} ms "
          f"(bits equal: {same}), serial again {t2:6.3f} ms", flush=True)
