"""Pin the CPU oracle against golden vectors emitted by the reference's own numpy modules
(tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import metric, project, voxelize


def test_fnv_known_answers(golden_dir):
    g = np.load(os.path.join(golden_dir, "fnv_hash.npz"))
    h = voxelize.fnv_hash_vec(g["coords"])
    assert h.dtype == np.uint64
    assert np.array_equal(h, g["hash"])


@pytest.mark.parametrize("case", [0, 1, 2])
def test_voxelize_matches_reference(golden_dir, case):
    g = np.load(os.path.join(golden_dir, "voxelize.npz"))
    p = f"c{case}_"
    seed, vs, aug = int(g[p + "seed"]), float(g[p + "voxel_size"]), bool(g[p + "aug"])
    # RNG consumption order must match the reference so that seeds line up
    np.random.seed(seed)
    M_v, M_r = voxelize.get_transformation_matrix(vs, aug)
    assert np.array_equal(M_v, g[p + "M_v"]) and np.array_equal(M_r, g[p + "M_r"])
    np.random.seed(seed)
    feats = g[p + "feats"].copy()
    c, f, _, inv, inds = voxelize.voxelize(g[p + "points"], feats, None, vs, aug)
    assert np.array_equal(c, g[p + "coords_aug"])
    assert np.array_equal(inds, g[p + "inds"])
    assert np.array_equal(inv, g[p + "inds_reconstruct"])
    assert np.array_equal(f, g[p + "feats_out"])
    # structural properties
    assert (c >= 0).all() and np.array_equal(c, np.floor(c))
    h = voxelize.fnv_hash_vec(c)
    assert (np.diff(h.astype(np.float64)) >= 0).all() and len(np.unique(h)) == len(h)


def test_mapping_scannet(golden_dir):
    g = np.load(os.path.join(golden_dir, "mapping.npz"))
    dim = tuple(int(v) for v in g["sn_image_dim"])
    K = project.scannet_intrinsics(dim, g["sn_K_native"])
    assert np.array_equal(K, g["sn_K"])
    m, w = project.compute_mapping_scannet(g["sn_wvt"], g["sn_points"], g["sn_depth"], K, dim,
                                           int(g["sn_cut"]), float(g["sn_tau"]))
    assert np.array_equal(m, g["sn_mapping"])
    assert np.array_equal(w, g["sn_weight"])
    m2, _ = project.compute_mapping_scannet(g["sn_wvt"], g["sn_points"], None, K, dim,
                                            int(g["sn_cut"]), float(g["sn_tau"]))
    assert np.array_equal(m2, g["sn_mapping_nodepth"])
    assert m[:, 2].sum() > 500
    # depth passed as a str: z-buffer of the cloud itself (fusion_util.py:126-130)
    m3, _ = project.compute_mapping_scannet(g["sn_wvt"], g["sn_points_render"], "render", K, dim,
                                            int(g["sn_cut"]), float(g["sn_tau"]))
    assert np.array_equal(m3, g["sn_mapping_render"])
    n = len(g["sn_points"])
    assert m3[:n, 2].sum() > 500 and m3[n:, 2].sum() < m3[:n, 2].sum() // 10     # the far shell is occluded


def test_mapping_edge_cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "mapping.npz"))
    dim = tuple(int(v) for v in g["ex_image_dim"])
    m, w = project.compute_mapping_scannet(np.eye(4, dtype=np.float32), g["ex_points"], g["ex_depth"],
                                           g["ex_K"], dim, 10, 0.05)
    assert np.array_equal(m, g["ex_mapping"])
    assert np.array_equal(w, g["ex_weight"])


def test_mapping_matterport(golden_dir):
    g = np.load(os.path.join(golden_dir, "mapping.npz"))
    dim = tuple(int(v) for v in g["mp_image_dim"])
    m = project.compute_mapping_matterport(g["mp_c2w"], g["mp_points"], g["mp_depth"], g["mp_K"], dim,
                                           int(g["mp_cut"]), float(g["mp_tau"]))
    assert np.array_equal(m, g["mp_mapping"])
    assert m[:, 2].sum() > 500


@pytest.mark.parametrize("case", [0, 1, 2])
def test_iou_counts(golden_dir, case):
    g = np.load(os.path.join(golden_dir, "iou.npz"))
    p = f"c{case}_"
    i, u, t = metric.intersection_and_union(g[p + "pred"], g[p + "target"], int(g[p + "C"]),
                                            [int(g[p + "ignore"])])
    assert np.array_equal(i, g[p + "I"]) and np.array_equal(u, g[p + "U"]) and np.array_equal(t, g[p + "T"])


def test_golden_inventory(golden_dir):
    cfg = json.load(open(os.path.join(golden_dir, "config_flat.json")))
    assert cfg["geopurify_scannet.yaml"]["voxel_size"] == 0.02
    assert cfg["geopurify_scannet.yaml"]["mask_shape"] == [484, 648]
    sizes = np.loadtxt(os.path.join(golden_dir, "scannet_val_point_counts.txt"))
    assert len(sizes) == 312 and sizes.min() == 28231 and sizes.max() == 301855


def test_iou_histc_semantics():
    """util/util.py:160-177 uses torch.histc(bins=C, min=0, max=C-1): ids >= C are dropped and every
    ignore id overwrites the prediction first."""
    import torch
    rng = np.random.default_rng(4)
    C, ign = 19, [19, 20]
    tgt = rng.integers(0, 21, size=4000)
    pred = rng.integers(0, 19, size=4000)
    out = torch.from_numpy(pred.copy())
    t = torch.from_numpy(tgt)
    for ig in ign:
        out[t == ig] = ig
    inter = out[out == t]
    ai = torch.histc(inter.float(), bins=C, min=0, max=C - 1)
    ao = torch.histc(out.float(), bins=C, min=0, max=C - 1)
    at = torch.histc(t.float(), bins=C, min=0, max=C - 1)
    i, u, tt = metric.intersection_and_union(pred, tgt, C, ign)
    assert np.array_equal(i, ai.numpy().astype(np.int64))
    assert np.array_equal(u, (ao + at - ai).numpy().astype(np.int64))
    assert np.array_equal(tt, at.numpy().astype(np.int64))
