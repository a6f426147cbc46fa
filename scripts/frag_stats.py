#!/usr/bin/env python3
"""How sparse are the 16-row x 32-union-row weight fragments of the matrix-core pooling operator? (tuning aid)"""
import os, sys, dataclasses
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import ops, pipeline as pl, synthetic as syn

cfg = dataclasses.replace(syn.CONFIGS["S"], num_views=1)
sc = syn.make_scene(cfg, 5557)
rigid = pl.scene_rigid_transform(cfg.voxel_size, 5557)
vox = ops.voxelize(torch.from_numpy(sc.coords).cuda(), rigid)
coords = vox["coords_aug"].to(torch.int32).contiguous()
perm, rank = ops.morton_order(coords)
cs = coords[perm.long()].contiguous()
grid = ops.grid_build(cs)
nbr = ops.knn_lattice(grid, cs, perm, 96)
Nv = cs.shape[0]
E = torch.nn.functional.normalize(torch.randn(Nv, 128, device="cuda"), dim=1)
w = ops.affinity_softmax(E, nbr, 20.0)
for BR in (64, 128):
    op = ops.pool_mfma_build(nbr, w, BR)
    nw = BR // 16
    a = (op.wa_hi.view(-1, nw, 64, 8) != 0)
    frag = a.flatten(2).any(dim=2)                       # [steps, waves]
    print(f"BR={BR}: fragments {frag.numel()}, non-empty {frag.float().mean().item():.3f}; "
          f"density inside non-empty {a.flatten(2).float().mean(dim=2)[frag].mean().item():.3f}; "
          f"steps with all waves empty {(~frag.any(dim=1)).float().mean().item():.4f}")
    kg = a.view(-1, nw, 4, 16, 8).permute(0, 1, 2, 3, 4).flatten(3).any(dim=3)   # [steps, waves, kgroup(8 union rows)]
    print(f"   16x8 sub-fragments non-empty {kg.float().mean().item():.3f}")
    per_row = a.view(-1, nw, 4, 16, 8)                   # lane = kg*16 + m
    rows_any = per_row.permute(0, 1, 3, 2, 4).flatten(3).any(dim=3)            # [steps, waves, m]
    print(f"   (row, step) pairs non-empty {rows_any.float().mean().item():.3f}")
