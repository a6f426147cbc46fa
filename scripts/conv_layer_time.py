#!/usr/bin/env python3
"""One 512->512 layer of the S scene (pre-split operands in, split planes out), median of 5 x 12 launches after 30 warm ones.
usage: conv_layer_time.py [chunk_rows | balanced ...]     (environment: GP_CONV_TARGET_TILES sets ops.CONV_TARGET_TILES for this script, GP_SCENE_SEED the scene)"""
import os
import sys
import dataclasses

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import ops, pipeline as pl, synthetic as syn  # noqa: E402

if os.environ.get("GP_CONV_TARGET_TILES"):
    ops.CONV_TARGET_TILES = int(os.environ["GP_CONV_TARGET_TILES"])

cfg = dataclasses.replace(syn.CONFIGS["S"], num_views=1)
SEED = int(os.environ.get("GP_SCENE_SEED", "5557"))
sc = syn.make_scene(cfg, SEED)
rigid = pl.scene_rigid_transform(cfg.voxel_size, SEED)
vox = ops.voxelize(torch.from_numpy(sc.coords).cuda(), rigid)
coords = vox["coords_aug"].to(torch.int32).contiguous()
perm, rank = ops.morton_order(coords)
cs = coords[perm.long()].contiguous()
nm = ops.kernel_map_build(ops.grid_build(cs), cs)
Nv = cs.shape[0]
g = torch.Generator(device="cuda").manual_seed(3)
X, W = torch.randn(Nv, 512, device="cuda", generator=g), torch.randn(27, 512, 512, device="cuda", generator=g) * 0.01
hi, lo = ops.conv_weights_split(W, 64.0)
xs = ops.split_f16(X, per_row=True)
sc_, sh = torch.ones(512, device="cuda"), torch.zeros(512, device="cuda")
for chunk_rows in [a if a == "balanced" else int(a) for a in sys.argv[1:]] or ["balanced"]:
    pairs = ops.conv_pairs_build(nm, chunk_rows)
    ys = tuple(torch.empty((Nv, 512), dtype=torch.float16, device="cuda") for _ in range(2)) + (torch.empty(Nv, device="cuda"),)
    run = lambda: ops.sparse_conv_f16x3(None, pairs, hi, lo, sc_, sh, relu=True, x_split=xs[:2], x_row_inv=xs[2], out_split=ys[:2],
                                        out_row_inv=ys[2], want_f32=False)
    for _ in range(30):
        run()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(12):
            run()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 12)
    chk = float(ys[0].float().sum() + ys[1].float().sum() + ys[2].sum())
    to = np.array(list(pairs.chunk_tile_off))
    tiles = np.diff(to) * 2
    print(f"      seed {SEED}: Nv {Nv}, {pairs.num_chunks} launches, tiles per launch {tiles.min()}..{tiles.max()}, rounds of 256: {int(np.ceil(tiles / 256).sum())}", flush=True)
    print(f"chunk {str(chunk_rows):>8s} (target {ops.CONV_TARGET_TILES or 'auto'}): layer {np.median(ts):6.3f} ms "
          f"(min {min(ts):6.3f}, max {max(ts):6.3f}); checksum {chk:.6e}", flush=True)
