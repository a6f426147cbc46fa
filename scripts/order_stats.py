#!/usr/bin/env python3
"""CPU study (numpy / scipy, no GPU): how large are the neighbour unions of 64-row blocks of the pooling operator
under different voxel orders, and how full are the 16-row x 32-union-row weight fragments under different orders of
the union rows?  (tuning aid for pool_mfma.hip; run: python scripts/order_stats.py [num_points])"""
import dataclasses
import os
import sys
import time

import numpy as np
from scipy.spatial import cKDTree

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import pipeline as pl, synthetic as syn  # noqa: E402

K = 96


def voxels(npts, seed=5557):
    cfg = dataclasses.replace(syn.CONFIGS["S"], num_views=0, num_points=npts)
    sc = syn.make_scene(cfg, seed)
    M = pl.scene_rigid_transform(cfg.voxel_size, seed)
    h = np.concatenate([sc.coords, np.ones((len(sc.coords), 1))], 1) @ M.T
    c = np.floor(h[:, :3]).astype(np.int64)
    c = np.unique(c, axis=0)
    return c - c.min(0)


def part1by2(v):
    v = v.astype(np.uint64) & np.uint64(0x1FFFFF)
    v = (v | (v << np.uint64(32))) & np.uint64(0x1F00000000FFFF)
    v = (v | (v << np.uint64(16))) & np.uint64(0x1F0000FF0000FF)
    v = (v | (v << np.uint64(8))) & np.uint64(0x100F00F00F00F00F)
    v = (v | (v << np.uint64(4))) & np.uint64(0x10C30C30C30C30C3)
    v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
    return v


def morton(c):
    return part1by2(c[:, 0]) | (part1by2(c[:, 1]) << np.uint64(1)) | (part1by2(c[:, 2]) << np.uint64(2))


def hilbert3(c, bits=10):
    """Skilling's transpose algorithm, vectorised."""
    X = [c[:, i].astype(np.int64).copy() for i in range(3)]
    M = 1 << (bits - 1)
    Q = M
    while Q > 1:
        P = Q - 1
        for i in range(3):
            m = (X[i] & Q) != 0
            X[0] = np.where(m, X[0] ^ P, X[0])
            t = np.where(m, 0, (X[0] ^ X[i]) & P)
            X[0] ^= t
            X[i] ^= t
        Q >>= 1
    for i in range(1, 3):
        X[i] ^= X[i - 1]
    t = np.zeros_like(X[0])
    Q = M
    while Q > 1:
        t = np.where((X[2] & Q) != 0, t ^ (Q - 1), t)
        Q >>= 1
    for i in range(3):
        X[i] ^= t
    key = np.zeros(len(c), dtype=np.uint64)
    for b in range(bits - 1, -1, -1):
        for i in range(3):
            key = (key << np.uint64(1)) | ((X[i] >> b) & 1).astype(np.uint64)
    return key


def union_stats(nbr_o, br=64, name=""):
    """nbr_o: [Nv,K] ids in the SAME order as the rows.  padded union rows per row, fragment fill for sorted unions."""
    nv = len(nbr_o)
    nb = (nv + br - 1) // br
    tot = totp = 0
    frag = fragn = 0
    gfrag = 0
    for b in range(nb):
        rows = nbr_o[b * br:(b + 1) * br]
        u = np.unique(rows)
        tot += len(u)
        up = -(-len(u) // 32) * 32
        totp += up
        if b % 8 == 0:                                    # fragment statistics on a sample of blocks
            pos = np.searchsorted(u, rows)                # [rows, K] column in the sorted union
            step = pos // 32
            g = np.arange(len(rows))[:, None] // 16
            f = np.zeros((br // 16, up // 32), bool)
            f[np.broadcast_to(g, step.shape).ravel(), step.ravel()] = True
            frag += f.sum(); fragn += f.size
            # union rows re-ordered by the set of 16-row groups that use them (sorted by mask in Gray-ish order)
            use = np.zeros((len(u), br // 16), bool)
            use[pos.ravel(), np.broadcast_to(g, pos.shape).ravel()] = True
            mask = (use * (1 << np.arange(br // 16))).sum(1)
            order = np.lexsort((np.arange(len(u)), mask_rank(mask, br // 16)))
            newpos = np.empty(len(u), np.int64); newpos[order] = np.arange(len(u))
            step2 = newpos[pos] // 32
            f2 = np.zeros((br // 16, up // 32), bool)
            f2[np.broadcast_to(g, step2.shape).ravel(), step2.ravel()] = True
            gfrag += f2.sum()
    print(f"{name:28s} BR={br}: union/row {tot / nv:.3f}  padded {totp / nv:.3f}   16x32 fragments non-empty: sorted {frag / fragn:.3f}"
          f"  grouped-by-mask {gfrag / fragn:.3f}  (frags/row sorted {frag / fragn * totp / 32 * (br // 16) / nv:.3f}, grouped {gfrag / fragn * totp / 32 * (br // 16) / nv:.3f})",
          flush=True)
    return totp / nv


_rank_cache = {}


def mask_rank(mask, ng):
    """order of the group-set masks: by the lowest group, then by the number of groups, then by value (keeps rows used
    by the same leading group together)"""
    if ng not in _rank_cache:
        ms = list(range(1 << ng))
        low = np.array([((m & -m).bit_length() if m else 0) for m in ms])
        pc = np.array([bin(m).count("1") for m in ms])
        o = np.lexsort((np.array(ms), pc, low))
        r = np.empty(1 << ng, np.int64); r[o] = np.arange(1 << ng)
        _rank_cache[ng] = r
    return _rank_cache[ng][mask]


def reorder(nbr, perm):
    """rows in order perm (new row r = old row perm[r]); ids renamed"""
    inv = np.empty(len(perm), np.int64); inv[perm] = np.arange(len(perm))
    return inv[nbr[perm]]


def plane_order(c, cell=16, curve=hilbert3):
    """coarse cells in Hilbert order; inside a cell the voxels are projected along the cell's dominant normal axis
    (the axis of least variance) and ordered by a 2-D Morton key of the remaining two axes"""
    cc = c // cell
    ck = curve(cc)
    loc = c - cc * cell
    order0 = np.argsort(ck, kind="stable")
    keys = np.zeros(len(c), np.uint64)
    ck_s = ck[order0]
    starts = np.flatnonzero(np.concatenate([[True], ck_s[1:] != ck_s[:-1]]))
    ends = np.concatenate([starts[1:], [len(c)]])
    for s, e in zip(starts, ends):
        idx = order0[s:e]
        p = loc[idx].astype(np.float64)
        var = p.var(0) if len(idx) > 1 else np.zeros(3)
        ax = int(np.argmin(var))
        a, b = [i for i in range(3) if i != ax]
        u, v = loc[idx, a], loc[idx, b]
        k2 = np.zeros(len(idx), np.uint64)
        for bit in range(5):
            k2 |= ((u >> bit) & 1).astype(np.uint64) << np.uint64(2 * bit)
            k2 |= ((v >> bit) & 1).astype(np.uint64) << np.uint64(2 * bit + 1)
        keys[idx] = (k2 << np.uint64(8)) | loc[idx, ax].astype(np.uint64)
    return np.lexsort((keys, ck))


def greedy_patches(c, nbr, br=64):
    """region growing on the kNN graph: seeds in Hilbert order, a patch grows by the unassigned voxel nearest to its
    running centroid among the frontier (an upper bound on what a cheap device heuristic can reach)"""
    import heapq
    nv = len(c)
    assigned = np.zeros(nv, bool)
    horder = np.argsort(hilbert3(c), kind="stable")
    out = []
    cf = c.astype(np.float64)
    ptr = 0
    while ptr < nv:
        while ptr < nv and assigned[horder[ptr]]:
            ptr += 1
        if ptr >= nv:
            break
        seed = horder[ptr]
        patch = [seed]; assigned[seed] = True
        cen = cf[seed].copy()
        heap = []
        inheap = {seed}
        def push(i):
            for j in nbr[i, :24]:
                if not assigned[j] and j not in inheap:
                    inheap.add(j)
                    heapq.heappush(heap, (float(((cf[j] - cen) ** 2).sum()), int(j)))
        push(seed)
        while len(patch) < br and heap:
            _, j = heapq.heappop(heap)
            if assigned[j]:
                continue
            assigned[j] = True
            patch.append(j)
            push(j)
        out.extend(patch)
    return np.array(out)


def main():
    npts = int(sys.argv[1]) if len(sys.argv) > 1 else 150_000
    c = voxels(npts)
    nv = len(c)
    print(f"Nv = {nv}", flush=True)
    t0 = time.time()
    tree = cKDTree(c.astype(np.float64))
    _, nbr = tree.query(c.astype(np.float64), k=K, workers=8)
    print(f"kNN {time.time() - t0:.1f} s", flush=True)
    orders = {
        "morton": np.argsort(morton(c), kind="stable"),
        "hilbert": np.argsort(hilbert3(c), kind="stable"),
    }
    for cell in (8, 16, 32):
        orders[f"plane2d cell={cell}"] = plane_order(c, cell)
    for name, perm in orders.items():
        n2 = reorder(nbr, perm)
        for br in (64, 128):
            union_stats(n2, br, name)
    if "--greedy" in sys.argv:
        t0 = time.time()
        perm = greedy_patches(c, nbr)
        print(f"greedy {time.time() - t0:.1f} s", flush=True)
        n2 = reorder(nbr, perm)
        for br in (64, 128):
            union_stats(n2, br, "greedy patches")


if __name__ == "__main__":
    main()
