#!/usr/bin/env python3
"""Interleaved timing of the pooling kernel variants on one S-shaped voxel set (tuning aid)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import _lib, ops, pipeline as pl, synthetic as syn  # noqa: E402

cfg = syn.CONFIGS["S"]
import dataclasses
cfg = dataclasses.replace(cfg, num_views=1, num_points=int(sys.argv[2]) if len(sys.argv) > 2 else cfg.num_points)
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
ORDER = os.environ.get("GP_ORDER", "")                  # tuning aid: re-number the rows (host-side) before the operators are built
if ORDER:
    cpu = cs.cpu().numpy().astype(np.int64)

    def rcb(idx, leaf=128):
        out, stack = [], [idx]
        while stack:
            ix = stack.pop()
            if len(ix) <= leaf:
                out.append(ix)
                continue
            pts = cpu[ix]
            ax = int(np.argmax(pts.max(0) - pts.min(0)))
            n = len(ix)
            half = ((n // 2 + leaf - 1) // leaf) * leaf
            if half >= n:
                half = n // 2
            o = np.argsort(pts[:, ax], kind="stable")
            stack.append(ix[o[half:]])
            stack.append(ix[o[:half]])
        return np.concatenate(out)

    chunk = int(ORDER[3:]) if len(ORDER) > 3 else Nv
    sigma = np.concatenate([rcb(np.arange(b, min(b + chunk, Nv))) for b in range(0, Nv, chunk)])
    rk = np.empty(Nv, np.int64)
    rk[sigma] = np.arange(Nv)
    sg = torch.from_numpy(sigma).cuda()
    nbr = torch.from_numpy(rk).cuda()[nbr.long()[sg]].to(torch.int32).contiguous()
    w = w[sg].contiguous()
    print(f"rows re-numbered: {ORDER}", flush=True)
X = torch.randn(Nv, 544, device="cuda")
Y = torch.empty(Nv, D, device="cuda")
lib = _lib.load()
bytes_alg = Nv * (2 * D * 4 + K * 8)


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


variants = [("ell", None)]
tiles = {}
for R in (4, 8, 16):
    tiles[R] = ops.pool_tiles_build(nbr, w, R)
    print(f"R={R}: union entries/row = {tiles[R].total / Nv:.2f}", flush=True)
    for nf4 in (1, 2):
        for un in (4, 8):
            if (R, nf4, un) in ((16, 2, 8), (4, 1, 4), (4, 1, 8)):
                continue
            variants.append((f"tiles R={R} nf4={nf4} unroll={un}", (R, nf4, un)))
variants = [v for v in variants if v[1] is None or v[1][0] == 8 and v[1][2] == 4]
mf = {}
for BR in (64, 128):
    mf[BR] = ops.pool_mfma_build(nbr, w, BR, min_steps=9); torch.cuda.synchronize()
    t0 = time.time(); mf[BR] = ops.pool_mfma_build(nbr, w, BR, min_steps=9); torch.cuda.synchronize()
    print(f"mfma BR={BR}: union rows/row (padded) {mf[BR].total / Nv:.2f}  build (2nd call) {1e3 * (time.time() - t0):.2f} ms", flush=True)
for R in (8,):
    t0 = time.time(); ops.pool_tiles_build(nbr, w, R); torch.cuda.synchronize()
    print(f"tiles R={R}: build (2nd call) {1e3 * (time.time() - t0):.2f} ms", flush=True)
xs = ops.split_f16(X, D)
ys = tuple(torch.empty((Nv, D), dtype=torch.float16, device="cuda") for _ in range(2))
cs = ops.pool_cs_build(nbr, w); torch.cuda.synchronize()
t0 = time.time(); cs = ops.pool_cs_build(nbr, w); torch.cuda.synchronize()
_bm = cs.bu_mask.cpu().numpy().astype(np.int64) & 0xFF
print(f"cs BR=128: union rows/row (padded) {cs.total / Nv:.2f}  non-empty fragments {np.unpackbits(_bm.astype(np.uint8)[:, None], axis=1).mean():.3f}"
      f"  build (2nd call) {1e3 * (time.time() - t0):.2f} ms", flush=True)
variants += [("cs128 column-sliced (split out)", ("cs", 0)), ("cs128 column-sliced (fp32 out)", ("cs32", 0)),
             ("engine cs128 x 128c (split out)", ("eng", 0)), ("engine cs128 x 128c (fp32 out)", ("eng32", 0)),
             ("mfma64 (split out)", ("mfma", 64, 0, 0)), ("mfma64 (fp32 out)", ("mfma32", 64, 0, 0)),
             ("mfma64 column-sliced waves (split out)", ("mfmacs", 64, 0, 0)),
             ("mfma128 8w x (32r x 128c) (split out)", ("mfma", 128, 0, 0)),
             ("persist64 (split out)", ("persist", 64, 0, 0)), ("persist64 (fp32 out)", ("persist32", 64, 0, 0)),
             ("persist64 static tile lists (split out)", ("persist", 64, 0, 0, 1))]
ysp = {br: tuple(torch.empty((mf[br].rows_padded, D), dtype=torch.float16, device="cuda") for _ in range(2)) for br in (64,)}
Yp = {br: torch.empty((mf[br].rows_padded, D), device="cuda") for br in (64,)}
print("min steps per row block:", {br: mf[br].min_steps for br in (64, 128)}, flush=True)
ABL = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # pool_mfma ablation bits (timing only, results invalid)
lib.gp_debug_set(4, ABL)
lib.gp_debug_set(9, int(sys.argv[4]) if len(sys.argv) > 4 else 0)
if len(sys.argv) > 1:                                   # e.g. "mfma": only variants whose name contains the word
    variants = [v for v in variants if sys.argv[1] in v[0]]
res = {}
for rnd in range(3):
    for name, v in variants:
        if v is None:
            t = timeit(lambda: ops.pool_ell(X, nbr, w, D, Y))
        elif v[0] == "cs":
            t = timeit(lambda: ops.pool_cs_apply(xs, cs, D, out_split=ys))
        elif v[0] == "eng":
            t = timeit(lambda: ops.pool_cs_apply(xs, cs, D, out_split=ys, engine=True))
        elif v[0] == "eng32":
            t = timeit(lambda: ops.pool_cs_apply(xs, cs, D, out_f32=Y, engine=True))
        elif v[0] == "cs32":
            t = timeit(lambda: ops.pool_cs_apply(xs, cs, D, out_f32=Y))
        elif v[0] == "mfma":
            t = timeit(lambda: ops.pool_mfma_apply(xs, mf[v[1]], D, out_split=ys))
        elif v[0] == "mfmacs":
            lib.gp_debug_set(11, 4)
            t = timeit(lambda: ops.pool_mfma_apply(xs, mf[v[1]], D, out_split=ys))
            lib.gp_debug_set(11, 0)
        elif v[0] == "persist":
            lib.gp_debug_set(12, v[4] if len(v) > 4 else 0)
            t = timeit(lambda: ops.pool_mfma_apply_persistent(xs, mf[v[1]], D, out_split=ysp[v[1]]))
            lib.gp_debug_set(12, 0)
        elif v[0] == "persist32":
            t = timeit(lambda: ops.pool_mfma_apply_persistent(xs, mf[v[1]], D, out_f32=Yp[v[1]]))
        elif v[0] == "mfma32":
            t = timeit(lambda: ops.pool_mfma_apply(xs, mf[v[1]], D, out_f32=Y))
        else:
            R, nf4, un = v
            lib.gp_debug_set(1, nf4); lib.gp_debug_set(2, un)
            t = timeit(lambda: ops.pool_tiles_apply(X, tiles[R], D, Y))
        res.setdefault(name, []).append(t)
for name, ts in res.items():
    t = min(ts)
    print(f"{name:32s} min {t:7.3f} ms  med {np.median(ts):7.3f} ms  -> {bytes_alg / t / 1e6:7.1f} GB/s algorithmic ({bytes_alg / t / 1e6 / 80:.1f}% of 8 TB/s)", flush=True)
yc = torch.empty((Nv, D), device="cuda"); ye = torch.empty((Nv, D), device="cuda")
ops.pool_cs_apply(xs, cs, D, out_f32=yc)
ops.pool_ell(X, nbr, w, D, ye)
print("cs128 vs ELL max |diff|:", float((yc - ye).abs().max()), flush=True)
yg = torch.empty((Nv, D), device="cuda")
lib.gp_debug_set(4, 0)                                   # the comparisons below run the product kernels
ops.pool_cs_apply(xs, cs, D, out_f32=yc)
ops.pool_cs_apply(xs, cs, D, out_f32=yg, engine=True)
torch.cuda.synchronize()
print("engine vs ELL max |diff|:", float((yg - ye).abs().max()), " engine == cs128 bitwise:", bool(torch.equal(yg, yc)), flush=True)
# the column-sliced wave mapping computes the same sums in the same order per element: identical outputs
ya = tuple(torch.empty((Nv, D), dtype=torch.float16, device="cuda") for _ in range(2))
yb = tuple(torch.empty((Nv, D), dtype=torch.float16, device="cuda") for _ in range(2))
ops.pool_mfma_apply(xs, mf[64], D, out_split=ya)
lib.gp_debug_set(11, 4)
ops.pool_mfma_apply(xs, mf[64], D, out_split=yb)
lib.gp_debug_set(11, 0)
torch.cuda.synchronize()
print("column-sliced == default:", bool(torch.equal(ya[0], yb[0]) and torch.equal(ya[1], yb[1])),
      float((ya[0].float() - yb[0].float()).abs().max()))
