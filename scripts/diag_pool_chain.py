#!/usr/bin/env python3
"""Diagnostics of gp_pool_cs_apply_chain: where do its planes differ from the T launches?  usage: diag_pool_chain.py [mode]"""
import dataclasses
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import _lib, ops, pipeline as pl, synthetic as syn  # noqa: E402

cfg = dataclasses.replace(syn.CONFIGS["S"], num_views=1, num_points=150000)
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
E = torch.nn.functional.normalize(torch.randn(Nv, 128, device="cuda"), dim=1)
w = ops.affinity_softmax(E, nbr, 20.0)
X = torch.randn(Nv, 544, device="cuda")
op = ops.pool_cs_build(nbr, w)
ops.pool_cs_deps(op)
scl = ops.pow2_scale(X, D)
x0 = ops.split_f16(X, D, scale=scl[0:1])
dep0 = op.dep.clone()


def planes():
    return tuple(t.clone() for t in x0), tuple(torch.zeros((Nv, D), dtype=torch.float16, device="cuda") for _ in range(2))


def launches(T):
    xs, pong = planes()
    out = torch.zeros(Nv, D, device="cuda")
    sp = [xs, pong]
    src = sp[0]
    for t in range(T):
        last = t == T - 1
        dst = None if last else sp[(t + 1) % 2]
        ops.pool_cs_apply(src, op, D, out_split=dst, out_f32=out if last else None, out_scale=scl[1:2] if last else None)
        src = dst
    torch.cuda.synchronize()
    return out, xs, pong


def chain(T):
    xs, pong = planes()
    out = torch.zeros(Nv, D, device="cuda")
    ops.pool_cs_apply_chain(xs, pong, op, D, T, out, out_scale=scl[1:2])
    torch.cuda.synchronize()
    return out, xs, pong


def report(tag, a, b):
    names = ["out f32", "A hi", "A lo", "B hi", "B lo"]
    ta = [a[0], *a[1], *a[2]]
    tb = [b[0], *b[1], *b[2]]
    for n, u, v in zip(names, ta, tb):
        ne = (u != v)
        rows = ne.any(1).nonzero().flatten()
        if rows.numel() == 0:
            print(f"  {tag} {n}: identical")
            continue
        cols = ne.any(0).nonzero().flatten()
        blocks = torch.unique(rows // 128)
        d = (u.float() - v.float()).abs().max().item()
        print(f"  {tag} {n}: {rows.numel()} rows differ in {blocks.numel()} row blocks (first {blocks[:8].tolist()}), columns {cols.min().item()}..{cols.max().item()} "
              f"({cols.numel()}), max |diff| {d:.3e}, scale {v.float().abs().max().item():.3e}")


for mode in ("lists", "wait for every block"):
    if mode != "lists":
        d = dep0.clone().view(-1, 64)
        d[:, 0] = 1000
        op.dep.copy_(d.view(-1))
    for T in (2, 3, 5, 19):
        ref = launches(T)
        for rep in range(2):
            got = chain(T)
            print(f"[{mode}] T = {T} run {rep}  abort {int(op.flags[0].item())}")
            report("", got, ref)
