#!/usr/bin/env python3
"""Round 5: all T applications of the pooling operator in ONE launch (gp_pool_cs_apply_chain) and the cheaper alternative (the two
column halves as two chains of launches on two streams) against the T launches of cs_pool_kernel -- bits, time, in-kernel stamps.
usage: bench_pool_chain.py [num_points] [T] [stamps]"""
import dataclasses
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import _lib, ops, pipeline as pl, synthetic as syn  # noqa: E402

NPTS = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 19
STAMPS = len(sys.argv) > 3 and sys.argv[3] == "stamps"
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
E = torch.nn.functional.normalize(torch.randn(Nv, 128, device="cuda"), dim=1)
w = ops.affinity_softmax(E, nbr, 20.0)
X = torch.randn(Nv, 544, device="cuda")
lib = _lib.load()
bytes_alg = Nv * (2 * D * 4 + K * 8)
op = ops.pool_cs_build(nbr, w)
ops.pool_cs_deps(op)
torch.cuda.synchronize()
dep = op.dep.view(-1, 64)[:, 0].cpu().numpy()
print(f"Nv {Nv}  row blocks {dep.size}  dependency lists: mean {dep.mean():.1f} p50 {np.median(dep):.0f} p99 {np.percentile(dep, 99):.0f} max {dep.max()}"
      f"  lists that wait for every block: {(dep > 63).sum()}", flush=True)
# the lists against numpy: list(b) = {b} + blocks of b's union rows, made symmetric
_off = op.bu_off.cpu().numpy(); _row = op.bu_row.cpu().numpy(); _nb = dep.size
_src = np.repeat(np.arange(_nb), np.diff(_off)); _dst = _row // 128
_e = np.unique(np.concatenate([_src * _nb + _dst, _dst * _nb + _src, np.arange(_nb) * (_nb + 1)]))
_want = np.split(_e % _nb, np.cumsum(np.bincount(_e // _nb, minlength=_nb))[:-1])
_got = op.dep.view(-1, 64).cpu().numpy()
_ok = all(g[0] == len(wl) and (g[0] > 63 or set(g[1:g[0] + 1].tolist()) == set(wl.tolist())) for g, wl in zip(_got, _want))
print("dependency lists == numpy's symmetric closure:", _ok, flush=True)
scl = ops.pow2_scale(X, D)
x0 = ops.split_f16(X, D, scale=scl[0:1])


def planes():
    return tuple(t.clone() for t in x0), tuple(torch.empty((Nv, D), dtype=torch.float16, device="cuda") for _ in range(2))


def run_launches(xs, pong, out):
    sp = [xs, pong]
    src = sp[0]
    for t in range(T):
        last = t == T - 1
        dst = None if last else sp[(t + 1) % 2]
        ops.pool_cs_apply(src, op, D, out_split=dst, out_f32=out if last else None, out_scale=scl[1:2] if last else None)
        src = dst


side = torch.cuda.Stream()


def run_halves(xs, pong, out):
    """the two 256-column halves as two chains: half 0 on the current stream, half 1 on `side`, started half an application apart"""
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    for half, st in ((0, main), (1, side)):
        with torch.cuda.stream(st):
            sp = [xs, pong]
            src = sp[0]
            for t in range(T):
                last = t == T - 1
                dst = None if last else sp[(t + 1) % 2]
                ops.pool_cs_apply_half(src, op, D, half, out_split=dst, out_f32=out if last else None, out_scale=scl[1:2] if last else None)
                src = dst
    main.wait_stream(side)


def run_chain(xs, pong, out):
    ops.pool_cs_apply_chain(xs, pong, op, D, T, out, out_scale=scl[1:2])


def timed(fn, reps=5):
    ts = []
    for _ in range(reps):
        xs, pong = planes()
        out = torch.empty(Nv, D, device="cuda")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(xs, pong, out)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return ts, (out, xs, pong)


ref_t, ref = timed(run_launches)
for name, fn in (("T launches of cs_pool_kernel", run_launches), ("two column-half chains on two streams", run_halves),
                 ("ONE chained launch", run_chain)):
    for rnd in range(2):
        ts, got = timed(fn)
        same = all(torch.equal(a, b) for a, b in zip([got[0], *got[1], *got[2]], [ref[0], *ref[1], *ref[2]]))
        t = float(np.median(ts))
        print(f"{name:40s} {t:7.3f} ms for {T} applications = {t / T:7.4f} ms each  (min {min(ts) / T:.4f})  "
              f"{bytes_alg * T / t / 1e6 / 80:5.1f} % of 8 TB/s   bits == the {T} launches: {same}", flush=True)
ops.pool_cs_chain_check(op)
print("abort word:", int(op.flags[0].item()), " epoch:", op.epoch, flush=True)

# ---- the hand-off under uneven load: a second stream streams 1 GiB copies beside the chained launch; every word is compared
big = torch.empty(256 << 20, dtype=torch.float32, device="cuda")
big2 = torch.empty_like(big)
bad = 0
for it in range(8):
    xs, pong = planes()
    out = torch.empty(Nv, D, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(3):
            big2.copy_(big)
    run_chain(xs, pong, out)
    torch.cuda.synchronize()
    ok = all(torch.equal(a, b) for a, b in zip([out, *xs, *pong], [ref[0], *ref[1], *ref[2]]))
    bad += not ok
print(f"chained launch beside 3 GiB of copies on a second stream, 8 runs: {8 - bad} bit-identical, abort word {int(op.flags[0].item())}", flush=True)

if STAMPS:
    nb = dep.size
    per_xcd = (2 * nb + 7) // 8
    grid_wg = per_xcd * 8 * T
    buf = torch.zeros(grid_wg * 8 * 10, dtype=torch.int64, device="cuda")
    lib.gp_debug_ptr(0, buf.data_ptr(), buf.numel() * 8)
    xs, pong = planes()
    out = torch.empty(Nv, D, device="cuda")
    run_chain(xs, pong, out)
    torch.cuda.synchronize()
    lib.gp_debug_ptr(0, None, 0)
    st = buf.cpu().numpy().reshape(grid_wg, 8, 10)[:, 0, :]
    live = st[:, 7] > 0
    st = st[live]
    app = st[:, 9] >> 8
    t0 = st[:, 0].min()
    start = (st[:, 0] - t0) / 100.0          # us (100 MHz)
    end = start + st[:, 1] / 100.0
    clk = np.median(st[:, 7] / (st[:, 1] / 100.0)) / 1e3
    print(f"stamps: {live.sum()} workgroups, whole launch {end.max():.1f} us = {end.max() / T:.2f} us per application; shader clock {clk:.2f} GHz")
    print("app   first start   last end   span   tiles   dependency wait (us): mean / p99 / max     life (us) mean")
    for a in range(T):
        m = app == a
        wait = st[m, 5] / (clk * 1e3)
        life = st[m, 1] / 100.0
        print(f"{a:3d}   {start[m].min():10.1f}  {end[m].max():9.1f}  {end[m].max() - start[m].min():6.1f}  {m.sum():5d}   "
              f"{wait.mean():6.2f} / {np.percentile(wait, 99):6.2f} / {wait.max():6.2f}        {life.mean():6.2f}")
    seg = {"prologue": 2, "step work": 3, "hand-over wait": 4, "epilogue": 6}
    tot = st[:, 7].sum()
    print("share of a workgroup's life: " + ", ".join(f"{k} {100.0 * st[:, v].sum() / tot:.1f} %" for k, v in seg.items())
          + f", dependency wait {100.0 * st[:, 5].sum() / tot:.1f} %")
