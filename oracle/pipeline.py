"""End-to-end CPU oracle of the per-scene hot path (test infrastructure; also the `cpu_baseline`
"port" that bench.py times).  Follows the reference call stack of SURVEY.md 3.1:

  loader      dataset/data_loader_ablation.py:242-264 (mapper), :280-288 (view-drop rule),
              :348-351 (x/y labels), :364-366 (scene voxelization)
  lift        models/affinity_module.py:455-714 (masks) / :348-453 (dense)
  refine      models/affinity_module.py:1491-1607 (evaluate_scene)
  tail        run/validation.py:413-439

`vectorised=True` replaces the reference's two per-point Python loops (:633-638, :664-670) by
tensor ops with identical math (the stronger CPU baseline); `vectorised=False` keeps them.
"""
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import affinity, lift, metric, project, student, voxelize


def loader_math(scene, rigid, val_keep=10_000_000):
    """Mapping per view, view-drop rule, visible lists, scene voxelization with a given rigid matrix."""
    cfg = scene.cfg
    views = []
    for vi, v in enumerate(scene.views):
        if cfg.dataset == "scannet":
            K = project.scannet_intrinsics(cfg.image_dim, v.K)
            m, _ = project.compute_mapping_scannet(v.pose, scene.coords, v.depth, K, cfg.image_dim, cfg.cut_bound,
                                                   cfg.vis_thres)
        else:
            m = project.compute_mapping_matterport(v.pose, scene.coords, v.depth, v.K, cfg.image_dim, cfg.cut_bound,
                                                   cfg.vis_thres)
        mask = m[:, 2]
        n_v = int(mask.sum())
        if n_v == 0 or n_v < cfg.min_visible or n_v > val_keep:
            continue
        pt = np.where(mask == 1)[0]
        views.append({"src_view": vi, "pt": torch.from_numpy(pt), "x": torch.from_numpy(m[pt, 0]),
                      "y": torch.from_numpy(m[pt, 1])})
    homo = np.hstack((scene.coords, np.ones((scene.coords.shape[0], 1))))
    c = np.floor(homo @ rigid.T[:, :3])
    c = np.floor(c - c.min(0))
    inds, inv = voxelize.sparse_quantize_index(c)
    return {"views": views, "coords_3d": c[inds], "inv": torch.from_numpy(np.asarray(inv).astype(np.int64)),
            "inds": inds}


def evaluate_scene_oracle(scene, vlm, sd, rigid, K=96, sharpen=20.0, num_iters=19, vectorised=True,
                          dense_feat=None, timings=None, num_blocks=None, knn_impl="exact", lseg_feat=None, lifted=None):
    """Returns dict(scene_features [N,D], text_features, logit_scale, + intermediates).
    lifted: per-point features [N,D] to use INSTEAD of the oracle's own lift (rows 8-12 from a given lift: oracle/parity.py
    re-derives the downstream stages from the product's lift when a decision inside fp32 rounding noise went the other way)."""
    cfg = scene.cfg
    t0 = time.perf_counter()

    def tick(name):
        nonlocal t0
        if timings is not None:
            t1 = time.perf_counter()
            timings[name] = timings.get(name, 0.0) + (t1 - t0)
            t0 = t1

    ld = loader_math(scene, rigid)
    tick("loader(project+voxelize)")
    N = scene.coords.shape[0]
    xyz32 = torch.from_numpy(scene.coords).float()
    text = torch.from_numpy(vlm["text_embed"])
    scale = float(vlm["logit_scale"])
    if lifted is not None:
        Fp = torch.as_tensor(lifted).float().cpu()
        if dense_feat is None and lseg_feat is None and len(ld["views"]):
            text = F.normalize(text, dim=-1)
    elif lseg_feat is not None:                            # LSeg path: (feat_lo [V,D,h,w], image_shape (H,W))
        feats = [torch.from_numpy(lseg_feat[0][v["src_view"]]) for v in ld["views"]]
        Fp, _ = lift.lift_lseg(feats, lseg_feat[1], [v["pt"] for v in ld["views"]], [v["x"] for v in ld["views"]],
                               [v["y"] for v in ld["views"]], xyz32)
    elif dense_feat is not None:
        feats = [torch.from_numpy(dense_feat[v["src_view"]]) for v in ld["views"]]
        Fp, _ = lift.lift_dense(feats, [v["pt"] for v in ld["views"]], [v["x"] for v in ld["views"]],
                                [v["y"] for v in ld["views"]], xyz32)
    else:
        fs, lgs = [], []
        for v in ld["views"]:
            s = v["src_view"]
            f, lg = lift.lift_masks_view(torch.from_numpy(vlm["pred_masks"][s]), torch.from_numpy(vlm["pred_logits"][s]),
                                         torch.from_numpy(vlm["mask_embed"][s]), text, scale, v["x"], v["y"],
                                         xyz32[v["pt"]], cfg.mask_shape)
            fs.append(f), lgs.append(lg)
        tick("lift per view")
        if len(ld["views"]):
            text = F.normalize(text, dim=-1)          # :628 rebinds text_features to the normalised copy that :711 returns
        Fp = lift.fuse_views_top3(N, [v["pt"] for v in ld["views"]], fs, lgs, xyz32, faithful_loops=not vectorised)
    tick("fuse+fill")
    inv = ld["inv"]
    Nv = ld["coords_3d"].shape[0]
    gauss = torch.from_numpy(np.concatenate([scene.colors, scene.normals], 1).astype(np.float32))
    X = torch.cat([affinity.scatter_mean(Fp, inv, Nv), affinity.scatter_mean(gauss[:, :6], inv, Nv)], dim=1)
    tick("scatter_mean")
    coords_i = np.floor(ld["coords_3d"]).astype(np.int64)
    if num_blocks is None:
        num_blocks = sum(1 for k in sd if k.endswith(".conv1.kernel"))
    E = student.student_forward(X, coords_i, sd, num_blocks=num_blocks)
    tick("student")
    nbr = affinity.knn_lattice(coords_i, K) if knn_impl == "exact" else affinity.knn_kdtree(coords_i, K)
    tick("knn")
    w = affinity.affinity_weights(E, nbr, sharpen)
    tick("affinity")
    Y = affinity.pool_sparse(X, nbr, w, num_iters)
    tick("pooling")
    D = Fp.shape[1]
    out = Y[inv][:, :D]
    tick("gather")
    return {"scene_features": out, "text_features": text, "logit_scale": scale, "lifted": Fp, "X": X, "E": E,
            "nbr": nbr, "w": w, "inv": inv, "coords_3d": ld["coords_3d"], "views": ld["views"]}


def classify_and_count(result, labels, num_classes, ignore_ids):
    pred, _ = metric.classify(result["scene_features"], result["text_features"], result["logit_scale"])
    return pred, metric.intersection_and_union(pred.numpy(), labels, num_classes, list(ignore_ids))
