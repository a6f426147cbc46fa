#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference's own numpy modules (build container only).

Run:  python tests/golden/make_golden.py          (needs /root/reference; never runs on the GPU box)

Emits small .npz / .json fixtures next to this script.  Only inputs and expected outputs are
stored -- no reference source text.  Imported reference modules:
  dataset/voxelization_utils.py, dataset/voxelizer.py (after the collections.abc alias shim),
  models/utils/fusion_util.py, util/config.py, util/util.py (numpy intersectionAndUnion).
"""
import collections
import collections.abc
import json
import os
import sys

import numpy as np

collections.Sequence = collections.abc.Sequence
collections.Iterable = collections.abc.Iterable
REF = "/root/reference"
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))

from dataset.voxelization_utils import fnv_hash_vec  # noqa: E402
from dataset.voxelizer import Voxelizer  # noqa: E402
from models.utils.fusion_util import (PointCloudToImageMapper,  # noqa: E402
                                      PointCloudToImageMappermatterport, adjust_intrinsic,
                                      make_intrinsic)
from util import config as ref_config  # noqa: E402
from util.util import intersectionAndUnion  # noqa: E402

SCALE_B = (0.9, 1.1)
ROT_B = ((-np.pi / 64, np.pi / 64), (-np.pi / 64, np.pi / 64), (-np.pi, np.pi))
TRANS_B = ((-0.2, 0.2), (-0.2, 0.2), (0, 0))


def gold_fnv():
    rng = np.random.default_rng(1)
    a = rng.integers(0, 1200, size=(257, 3)).astype(np.float64)
    b = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1099, 1099, 1099],
                  [4294967295, 1, 2]], dtype=np.float64)
    arr = np.vstack([a, b])
    np.savez(os.path.join(HERE, "fnv_hash.npz"), coords=arr, hash=fnv_hash_vec(arr))


def make_points(rng, n):
    # two walls + floor, 2.2 cm pitch with jitter, duplicated points and negative coordinates
    u = rng.uniform(-2.0, 2.0, size=(n, 2))
    kind = rng.integers(0, 3, size=n)
    p = np.zeros((n, 3))
    p[kind == 0] = np.c_[u[kind == 0], np.full((kind == 0).sum(), -0.3)]
    p[kind == 1] = np.c_[u[kind == 1, 0], np.full((kind == 1).sum(), 1.7), u[kind == 1, 1]]
    p[kind == 2] = np.c_[np.full((kind == 2).sum(), -1.9), u[kind == 2]]
    p += rng.normal(0, 0.002, size=p.shape)
    p[: n // 20] = p[n // 2: n // 2 + n // 20]          # exact duplicates
    return p


def gold_voxelize():
    out = {}
    for case, (seed, n, vs, aug) in enumerate([(5557, 4000, 0.02, True), (7, 3000, 0.05, True),
                                               (11, 2500, 0.02, False)]):
        rng = np.random.default_rng(seed)
        pts = make_points(rng, n)
        feats = rng.uniform(0, 1, size=(n, 6))
        labels = rng.integers(0, 20, size=n).astype(np.float64)
        vox = Voxelizer(voxel_size=vs, clip_bound=None, use_augmentation=aug,
                        scale_augmentation_bound=SCALE_B, rotation_augmentation_bound=ROT_B,
                        translation_augmentation_ratio_bound=TRANS_B)
        np.random.seed(seed)
        M_v, M_r = vox.get_transformation_matrix()
        np.random.seed(seed)
        c, f, l, inv, inds = vox.voxelize(pts, feats.copy(), labels, return_ind=True)
        out[f"c{case}_seed"] = np.int64(seed)
        out[f"c{case}_voxel_size"] = np.float64(vs)
        out[f"c{case}_aug"] = np.bool_(aug)
        out[f"c{case}_points"] = pts
        out[f"c{case}_feats"] = feats
        out[f"c{case}_M_v"] = M_v
        out[f"c{case}_M_r"] = M_r
        out[f"c{case}_coords_aug"] = c
        out[f"c{case}_inds"] = np.asarray(inds, dtype=np.int64)
        out[f"c{case}_inds_reconstruct"] = np.asarray(inv, dtype=np.int64)
        out[f"c{case}_feats_out"] = f
    np.savez_compressed(os.path.join(HERE, "voxelize.npz"), **out)


def look_at_w2c(eye, target, up=(0, 0, 1)):
    """OpenCV-style camera (x right, y down, z forward)."""
    eye, target, up = map(lambda v: np.asarray(v, dtype=np.float64), (eye, target, up))
    z = target - eye
    z /= np.linalg.norm(z)
    x = np.cross(z, up)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z])                      # world -> cam rotation
    w2c = np.eye(4)
    w2c[:3, :3] = R
    w2c[:3, 3] = -R @ eye
    return w2c


def gold_mapping():
    out = {}
    rng = np.random.default_rng(3)
    # ---- ScanNet mapper: random scene + occluder depth
    n = 3000
    pts = make_points(rng, n) + np.array([0, 0, 0.0])
    K_native = make_intrinsic(1170.187988, 1170.187988, 647.75, 483.75)
    image_dim = (648, 484)
    mapper = PointCloudToImageMapper(image_dim, 0.05, 10, K_native)
    w2c = look_at_w2c([1.5, -1.5, 1.0], [-1.0, 1.5, 0.3])
    wvt = w2c.T.astype(np.float32)               # world_view_transform = W2C^T, float32
    # depth: true z of a plane-ish scene with noise + holes + occluder band
    homo = np.c_[pts, np.ones(n)]
    z = (wvt.T.astype(np.float64) @ homo.T)[2]
    yy, xx = np.mgrid[0:image_dim[1], 0:image_dim[0]]
    depth = 2.5 + (((xx // 8) * 37 + (yy // 8) * 91) % 64 - 32) / 50.0   # blocky, compressible
    # make most pixels that points land on depth-consistent up to a blocky +-8 % factor
    pc = wvt.T.astype(np.float64) @ homo.T
    uu = np.rint(pc[0] * mapper.intrinsics[0, 0] / pc[2] + mapper.intrinsics[0, 2]).astype(int)
    vv = np.rint(pc[1] * mapper.intrinsics[1, 1] / pc[2] + mapper.intrinsics[1, 2]).astype(int)
    ok = (uu >= 0) & (uu < image_dim[0]) & (vv >= 0) & (vv < image_dim[1]) & (pc[2] > 0)
    fac = 1.0 + (((uu // 8) * 37 + (vv // 8) * 91) % 64 - 32) / 400.0
    depth[vv[ok], uu[ok]] = np.round(pc[2][ok] * fac[ok], 3)
    depth[:, 300:340] = 0.4                      # occluder
    depth[100:120, :] = 0.0                      # holes
    m, wgt = mapper.compute_mapping(wvt, pts, depth)
    out.update(sn_points=pts, sn_wvt=wvt, sn_depth=depth, sn_K_native=K_native,
               sn_K=mapper.intrinsics, sn_image_dim=np.array(image_dim), sn_cut=np.int64(10),
               sn_tau=np.float64(0.05), sn_mapping=m, sn_weight=wgt, sn_z=z)
    m2, _ = mapper.compute_mapping(wvt, pts, None)
    out["sn_mapping_nodepth"] = m2
    # ---- "render" mode (depth passed as a str): the z-buffer of the cloud itself; a denser cloud with points behind
    #      each other along the rays (two shells) so that the buffer really occludes
    pts_r = np.concatenate([pts, pts * np.array([1.0, 1.0, 1.0]) + 0.35 * (pts - np.array([1.5, -1.5, 1.0]))])
    m3, _ = mapper.compute_mapping(wvt, pts_r, "render")
    out.update(sn_points_render=pts_r, sn_mapping_render=m3)
    # ---- exact-arithmetic edge cases: identity pose, power-of-two focal
    Ke = make_intrinsic(256.0, 256.0, 64.0, 48.0)
    mapper_e = PointCloudToImageMapper((128, 96), 0.05, 10, Ke)   # cx,cy already half size
    us = np.array([9.5, 10.0, 10.5, 11.5, 12.5, 117.0, 117.5, 118.0, 118.5, 64.0, 64.0, 64.0])
    vs = np.array([48.0, 48.0, 48.0, 48.0, 48.0, 48.0, 48.0, 48.0, 48.0, 9.5, 10.5, 85.5])
    zz = np.full(us.shape, 2.0)
    zz[3] = -2.0                                                   # behind the camera
    pe = np.c_[(us - 64.0) * zz / 256.0, (vs - 48.0) * zz / 256.0, zz]
    de = np.full((96, 128), 2.0)
    de[48, 12] = 2.0 / 0.95 + 1e-9                                 # just outside |d-z|<=tau*d ... inside
    de[48, 117] = 1.9                                              # |1.9-2|=0.1 > 0.095
    de[85, 64] = 2.1                                               # |2.1-2|=0.1 <= 0.105
    me, we = mapper_e.compute_mapping(np.eye(4, dtype=np.float32), pe, de)
    out.update(ex_points=pe, ex_depth=de, ex_K=mapper_e.intrinsics, ex_image_dim=np.array((128, 96)),
               ex_mapping=me, ex_weight=we)
    # ---- Matterport mapper: c2w float32, per-view 3x3 K
    Km = np.array([[1075.0, 0, 629.7], [0, 1076.2, 522.3], [0, 0, 1.0]])
    Km[0] *= 640 / 1280.0
    Km[1] *= 512 / 1024.0
    c2w = np.linalg.inv(look_at_w2c([1.2, -1.0, 0.9], [-1.5, 1.7, 0.2])).astype(np.float32)
    mm = PointCloudToImageMappermatterport((640, 512), 0.02, 10)
    yy, xx = np.mgrid[0:512, 0:640]
    dm = 2.6 + (((xx // 8) * 29 + (yy // 8) * 53) % 64 - 32) / 40.0
    pcm = np.linalg.inv(c2w) @ np.c_[pts, np.ones(n)].T
    uu = np.rint(pcm[0] * Km[0, 0] / pcm[2] + Km[0, 2]).astype(int)
    vv = np.rint(pcm[1] * Km[1, 1] / pcm[2] + Km[1, 2]).astype(int)
    ok = (uu >= 0) & (uu < 640) & (vv >= 0) & (vv < 512) & (pcm[2] > 0)
    fac = 1.0 + (((uu // 8) * 29 + (vv // 8) * 53) % 64 - 32) / 1000.0
    dm[vv[ok], uu[ok]] = np.round(pcm[2][ok] * fac[ok], 3)
    mp = mm.compute_mapping(c2w, pts, dm, Km)
    out.update(mp_points=pts, mp_c2w=c2w, mp_depth=dm, mp_K=Km, mp_image_dim=np.array((640, 512)),
               mp_cut=np.int64(10), mp_tau=np.float64(0.02), mp_mapping=mp)
    # ---- adjust_intrinsic
    Ka = make_intrinsic(1170.187988, 1170.187988, 647.75, 483.75)
    out["adj_in"] = Ka.copy()
    out["adj_out"] = adjust_intrinsic(Ka.copy(), [1296, 968], [648, 484])
    np.savez_compressed(os.path.join(HERE, "mapping.npz"), **out)


def gold_iou():
    rng = np.random.default_rng(9)
    out = {}
    for case, (C, ign) in enumerate([(19, 20), (160, 255), (21, 255)]):
        n = 5000
        tgt = rng.integers(0, C, size=n)      # no value == C: np.histogram's closed last bin would count it
        tgt[rng.random(n) < 0.05] = ign
        pred = rng.integers(0, C, size=n)
        agree = rng.random(n) < 0.6
        pred[agree] = np.minimum(tgt[agree], C - 1)
        i, u, t = intersectionAndUnion(pred.copy(), tgt.copy(), C, ignore_index=ign)
        out[f"c{case}_C"] = np.int64(C)
        out[f"c{case}_ignore"] = np.int64(ign)
        out[f"c{case}_pred"] = pred
        out[f"c{case}_target"] = tgt
        out[f"c{case}_I"], out[f"c{case}_U"], out[f"c{case}_T"] = i, u, t
    np.savez_compressed(os.path.join(HERE, "iou.npz"), **out)


def gold_config():
    res = {}
    cfgdir = os.path.join(REF, "config")
    for fn in sorted(os.listdir(cfgdir)):
        if not fn.startswith("geopurify_"):
            continue
        cfg = ref_config.load_cfg_from_cfg_file(os.path.join(cfgdir, fn))
        res[fn] = json.loads(json.dumps(dict(cfg), default=lambda o: dict(o)))
    # CLI override semantics
    cfg = ref_config.load_cfg_from_cfg_file(os.path.join(cfgdir, "geopurify_scannet.yaml"))
    cfg2 = ref_config.merge_cfg_from_list(cfg, ["voxel_size", "0.05", "test_classes", "21",
                                                "save_path", "out/x"])
    res["__override__"] = {"voxel_size": cfg2.voxel_size, "test_classes": cfg2.test_classes,
                           "save_path": cfg2.save_path}
    with open(os.path.join(HERE, "config_flat.json"), "w") as f:
        json.dump(res, f, indent=0, sort_keys=True)


def gold_scene_sizes():
    """Per-scene labelled-point totals of the 312 ScanNet-val scenes (shape statistics only:
    sum of the label histogram column of dataset/scannet_val_metrics.tsv)."""
    import csv
    sizes = []
    with open(os.path.join(REF, "dataset", "scannet_val_metrics.tsv"), newline="") as f:
        rd = csv.reader(f, delimiter="\t")
        next(rd)
        for row in rd:
            hist = [float(v) for v in row[3].replace("[", " ").replace("]", " ").split()]
            sizes.append(int(round(sum(hist))))
    with open(os.path.join(HERE, "scannet_val_point_counts.txt"), "w") as f:
        f.write("\n".join(str(s) for s in sizes) + "\n")
    return sizes


if __name__ == "__main__":
    gold_fnv()
    gold_voxelize()
    gold_mapping()
    gold_iou()
    gold_config()
    print(len(gold_scene_sizes()), 'scene sizes')
    print("golden fixtures written to", HERE)
