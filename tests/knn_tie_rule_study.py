#!/usr/bin/env python3
"""How much does the UNPINNED tie rule of the exact kNN matter?  (CPU only, oracle only.)

faiss.IndexFlatL2 (models/affinity_module.py:1551-1557) is absent from /root/reference, so the order in which it breaks ties
between voxels at the same squared distance cannot be pinned; the build adopts (d^2, id) ASCENDING.  Voxel coordinates are
integers, so ties at the rank-96/97 boundary are the rule, not the exception: the neighbour SET of most voxels depends on it.
This script measures, on the S-shaped synthetic scene, what the opposite rule -- (d^2, id DESCENDING) -- does downstream:
the affinity weights, the pooled features after 19 applications and the class decisions.  Lives under tests/ because it runs
the oracle.  usage: python tests/knn_tie_rule_study.py [num_points=150000] [embed source: random|smooth]"""
import dataclasses
import os
import sys
import time

import numpy as np
import torch
from scipy.spatial import cKDTree

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import pipeline as pl, synthetic as syn  # noqa: E402
from oracle import affinity as o_aff  # noqa: E402

K = 96


def knn_both_rules(c):
    """exact (K+1)-NN of integer voxels under the two tie rules: candidates from a KD-tree ball query that provably contains
    every voxel up to the K+1-th distance, then an exact lexicographic sort of (d2, +-id)."""
    tree = cKDTree(c.astype(np.float64))
    d, _ = tree.query(c.astype(np.float64), k=K + 1, workers=8)
    r = d[:, -1]                                            # distance of the (K+1)-th neighbour (self included)
    asc = np.empty((len(c), K), np.int64)
    desc = np.empty((len(c), K), np.int64)
    tied = np.zeros(len(c), bool)
    n_tie_members = np.zeros(len(c), np.int64)
    for s in range(0, len(c), 20000):
        cand = tree.query_ball_point(c[s:s + 20000].astype(np.float64), r[s:s + 20000] + 1e-9, workers=8)
        for i, ids in enumerate(cand):
            q = s + i
            ids = np.asarray(ids, dtype=np.int64)
            d2 = ((c[ids] - c[q]) ** 2).sum(1)
            # self first under both rules (the reference drops column 0 = the query itself at d2 = 0)
            oa = ids[np.lexsort((ids, d2))]
            od = ids[np.lexsort((-ids, d2))]
            oa = np.concatenate([[q], oa[oa != q]])[:K + 1]
            od = np.concatenate([[q], od[od != q]])[:K + 1]
            asc[q], desc[q] = oa[1:], od[1:]
            dk = np.sort(d2)[K]                             # squared distance at rank K (0-based, self included)
            m = int((d2 == dk).sum())
            inside = int((np.sort(d2)[:K + 1] == dk).sum())
            tied[q] = m > inside                            # more voxels at the boundary distance than slots left for them
            n_tie_members[q] = m
    return asc, desc, tied, n_tie_members


def main():
    npts = int(sys.argv[1]) if len(sys.argv) > 1 else 150_000
    mode = sys.argv[2] if len(sys.argv) > 2 else "smooth"
    cfg = dataclasses.replace(syn.CONFIGS["S"], num_views=0, num_points=npts)
    sc = syn.make_scene(cfg, 5557)
    M = pl.scene_rigid_transform(cfg.voxel_size, 5557)
    h = np.concatenate([sc.coords, np.ones((len(sc.coords), 1))], 1) @ M.T
    c = np.unique(np.floor(h[:, :3]).astype(np.int64), axis=0)
    c -= c.min(0)
    Nv = len(c)
    t0 = time.time()
    asc, desc, tied, members = knn_both_rules(c)
    same_set = np.array([set(a) == set(b) for a, b in zip(asc, desc)])
    print(f"Nv = {Nv}; kNN under both rules {time.time() - t0:.0f} s")
    print(f"voxels with a tie at the rank-{K}/{K + 1} boundary: {tied.mean() * 100:.1f} %   neighbour SET differs between the rules: "
          f"{(~same_set).mean() * 100:.1f} %   mean differing neighbours per affected voxel: "
          f"{np.mean([len(set(a) - set(b)) for a, b in zip(asc[~same_set], desc[~same_set])]):.2f} of {K}")
    g = torch.Generator().manual_seed(1)
    D, C = 512, 19
    if mode == "random":
        E = torch.nn.functional.normalize(torch.randn(Nv, 128, generator=g), dim=1)
        X = torch.nn.functional.normalize(torch.randn(Nv, D, generator=g), dim=1)
    else:
        # spatially smooth embeddings / features (what a trained student and a 2D VLM produce): low-frequency fields of the
        # voxel position plus noise
        cf = torch.from_numpy(c.astype(np.float32)) / 50.0
        B1, B2 = torch.randn(3, 128, generator=g), torch.randn(3, D, generator=g)
        E = torch.nn.functional.normalize(torch.sin(cf @ B1) + 0.3 * torch.randn(Nv, 128, generator=g), dim=1)
        X = torch.nn.functional.normalize(torch.sin(cf @ B2) + 0.3 * torch.randn(Nv, D, generator=g), dim=1)
    text = torch.nn.functional.normalize(torch.randn(C, D, generator=g), dim=1)
    torch.set_num_threads(8)
    res = {}
    for name, nbr in (("asc", asc), ("desc", desc)):
        nb = torch.from_numpy(nbr)
        w = o_aff.affinity_weights(E, nb, 20.0)
        Y = o_aff.pool_sparse(X, nb, w, 19)
        res[name] = (w, Y, (Y @ text.t()).argmax(1))
    dY = (res["asc"][1] - res["desc"][1]).abs()
    per_row = dY.max(1).values
    scale = res["asc"][1].abs().max().item()
    flips = (res["asc"][2] != res["desc"][2]).float().mean().item()
    print(f"embeddings / features: {mode}")
    print(f"pooled features after 19 applications, |asc - desc|:  max {dY.max().item():.3e}   p99 of the row maxima {np.percentile(per_row.numpy(), 99):.3e}"
          f"   median {per_row.median().item():.3e}   (feature scale {scale:.3f})")
    print(f"class decisions that flip (arg-max over {C} random text embeddings): {flips * 100:.3f} % of the voxels")
    print(f"rows within the north-star tolerance 1e-4 of each other: {(per_row <= 1e-4).float().mean().item() * 100:.1f} %")


if __name__ == "__main__":
    main()
