#!/usr/bin/env python3
"""Round 5: the affinity weights computed on the matrix cores straight into the pooling operator's fragments
(gp_pool_cs_structure_valid + gp_affinity_cs_fragments) against affinity_block_kernel + the dst table (gp_pool_cs_structure +
gp_affinity_softmax_scatter): fragments, one pooled application, time.  usage: bench_affinity_cs.py [num_points]"""
import dataclasses
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import ops, pipeline as pl, synthetic as syn  # noqa: E402

NPTS = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
cfg = dataclasses.replace(syn.CONFIGS["S"], num_views=1, num_points=NPTS)
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
# embeddings with spatial structure + noise (cosines spread over [-1, 1] instead of all ~0)
base = torch.randn(64, 128, device="cuda")
E = torch.nn.functional.normalize(base[(cs[:, 0].long() // 8 + 3 * (cs[:, 1].long() // 8)) % 64] + 0.7 * torch.randn(Nv, 128, device="cuda"), dim=1).contiguous()


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


op_a = ops.pool_cs_plan(nbr, structure=True)
w = ops.affinity_softmax(E, nbr, 20.0, into=op_a)
op_b = ops.pool_cs_plan(nbr, structure="valid")
ops.affinity_cs_fragments(E, 20.0, op_b)
torch.cuda.synchronize()
assert torch.equal(op_a.bu_row, op_b.bu_row) and torch.equal(op_a.bu_mask, op_b.bu_mask) and torch.equal(op_a.bu_off, op_b.bu_off)
steps = op_a.total // 32
m = op_a.bu_mask.cpu().numpy().astype(np.int64)
live = ((m[:, None] >> np.arange(8)[None, :]) & 1).astype(bool)                      # [steps, 8]
fa = (op_a.wa_hi.float() + op_a.wa_lo.float()).view(steps, 8, 512).cpu().numpy() / 1024.0
fb = (op_b.wa_hi.float() + op_b.wa_lo.float()).view(steps, 8, 512).cpu().numpy() / 1024.0
d = np.abs(fa - fb)[live]
print(f"Nv {Nv}: {live.sum()} non-empty fragments of {live.size}; weights (hi + lo) / 2^10: max |new - old| {d.max():.3e}, "
      f"mean {d.mean():.3e}; row sums of the new operator in [{0:.0f}, ...]", flush=True)
# the oracle's fp64 weights for a sample of rows
idx = torch.randint(0, Nv, (2000,), device="cuda")
Ed = E.double()
cos = (Ed[idx][:, None, :] * Ed[nbr[idx].long()]).sum(-1)
wref = torch.softmax(20.0 * cos, dim=1)
print(f"affinity_block_kernel vs fp64 softmax (2000 rows): max |dw| {(w[idx].double() - wref).abs().max().item():.3e}", flush=True)
# the new fragments against the same fp64 weights: rebuild w[row, j] from op_b through op_a's dst table (element index of (row, j))
wb = ((op_b.wa_hi.float() + op_b.wa_lo.float()) / 1024.0)[op_a.dst[idx].long()]
print(f"affinity_cs_fragments  vs fp64 softmax (2000 rows): max |dw| {(wb.double() - wref).abs().max().item():.3e}", flush=True)
X = torch.randn(Nv, 544, device="cuda")
xs = ops.split_f16(X, D)
ya, yb = torch.empty(Nv, D, device="cuda"), torch.empty(Nv, D, device="cuda")
ops.pool_cs_apply(xs, op_a, D, out_f32=ya)
ops.pool_cs_apply(xs, op_b, D, out_f32=yb)
print(f"one pooled application: max |new - old| {(ya - yb).abs().max().item():.3e} (scale {ya.abs().max().item():.2f}); finite: {bool(torch.isfinite(yb).all())}", flush=True)
for rnd in range(2):
    t_old_s = timeit(lambda: ops.pool_cs_plan(nbr, structure=True))
    t_new_s = timeit(lambda: ops.pool_cs_plan(nbr, structure="valid"))
    t_old = timeit(lambda: ops.affinity_softmax(E, nbr, 20.0, into=op_a))
    t_new = timeit(lambda: ops.affinity_cs_fragments(E, 20.0, op_b))
    print(f"structure (count + fill, incl. one host sync): dst table {t_old_s:.3f} ms, validity words {t_new_s:.3f} ms;   "
          f"affinity: block kernel + scatter {t_old:.3f} ms, matrix cores -> fragments (incl. the split of E) {t_new:.3f} ms", flush=True)

from geopurify_amd import _lib
lib = _lib.load()
for bits, what in ((0, "everything"), (1, "no fragment reads / MFMA"), (2, "no list stores"), (4, "no softmax"), (8, "no fragment pass"), (16, "no LDS-DMA"),
                   (1 | 2 | 4 | 8, "ring only"), (1 | 2 | 4 | 8 | 16, "skeleton")):
    lib.gp_debug_set(8, bits)
    t = timeit(lambda: ops.affinity_cs_fragments(E, 20.0, op_b))
    print(f"tuning twin, {what:28s} {t:.3f} ms (incl. the split of E)", flush=True)
lib.gp_debug_set(8, 0)
t = timeit(lambda: ops.split_f16(E, 128, scale=torch.tensor([1024.0], device="cuda")))
print(f"the split of E alone (incl. a host->device scalar copy): {t:.3f} ms")
