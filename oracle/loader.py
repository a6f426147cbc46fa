"""Oracle of the per-view sample tuple of the reference datasets (test infrastructure).

Follows dataset/data_loader_ablation.py:242-394 (ScanNet: `ScannetLoaderFull.__getitem__` after the scene cache) and
dataset/data_loader_matterport.py:190-300 (the Matterport variant: camera-to-world = world_view_transform^T, intrinsics
given per view, label_2d all zero).  numpy only; np.random is consumed exactly as the reference does (two voxelizations
per view: the visible points first, then the whole scene -- dataset/voxelizer.py:32-58).
Pinned by tests/golden/ref_loader_*.npz (the reference's own __getitem__ run in the build container).
"""
import numpy as np

from . import project, voxelize


def scene_prepare(locs_in, feats_in, normals, labels_in, ignore_last):
    """The scene-cache branch (data_loader_ablation.py:154-224): colours in [-1, 1] go to [0, 1], point features are
    [colour | normal], labels -100 / 255 become the last ignore category."""
    feats = np.asarray(feats_in)
    if feats.min() >= -1.0 and feats.max() <= 1.0:
        feats = (feats.astype(np.float64) + 1.0) / 2.0
    point_features = np.concatenate([feats, normals], axis=1)
    labels = np.array(labels_in, copy=True)
    labels[labels == -100] = ignore_last
    labels[labels == 255] = ignore_last
    return point_features, labels


def view_sample(locs_in, labels_in, point_features, world_view_transform, intrinsics, depth, image_u8, label_img, *, dataset,
                img_dim, vis_thres, cut_bound, voxel_size, category_split, split, val_keep, label_2d_ids, input_color=False):
    """Returns the 20 slots as numpy arrays (slot 18 = None), or None for a dropped view."""
    N = locs_in.shape[0]
    mapping = np.ones([N, 4], dtype=int)
    if dataset == "scannet":
        K = project.scannet_intrinsics(img_dim, intrinsics)
        m, _ = project.compute_mapping_scannet(world_view_transform, locs_in, depth, K, img_dim, cut_bound, vis_thres)
    else:
        m = project.compute_mapping_matterport(np.asarray(world_view_transform).T, locs_in, depth, intrinsics, img_dim, cut_bound,
                                               vis_thres)
    mapping[:, 1:4] = m
    if mapping[:, 3].sum() == 0:
        return None
    mask = mapping[:, 3]
    label_3d = labels_in[mask == 1].copy()
    feature_3d = point_features[mask == 1].copy()
    locals_3d = locs_in[mask == 1].copy()
    binary = labels_in[mask == 1].copy()
    unique_map = mapping.copy()
    mapping = mapping[np.all(mapping != 0, axis=1)]
    # the reference writes into the array it tests (binary_label IS label_3d_clone, :265-274): a base class becomes 1 and is
    # then tested against the novel list as 1
    binary[np.isin(binary, category_split["base_category"])] = 1
    binary[np.isin(binary, category_split["novel_category"])] = 0
    n_vis = int(np.sum(mask))
    if split == "train":
        if n_vis < 400 or n_vis > 65000:
            return None
    elif n_vis < 400 or n_vis > val_keep:
        return None
    img = np.asarray(image_u8).astype(np.float32)            # cv2.resize to img_dim is the identity on img_dim-sized input
    if dataset == "scannet":
        lab2d = np.array(label_img, copy=True).astype(np.int32)
        ids = list(label_2d_ids) if split in ("val", "test") else [label_2d_ids[c] for c in category_split["base_category"]]
        remap = {v: i for i, v in enumerate(ids)}
        lab2d[~np.isin(lab2d, ids)] = 255
        lab2d = np.vectorize(lambda v: remap.get(v, v))(lab2d)
        if split not in ("val", "test"):
            lab2d[lab2d == 255] = len(category_split["base_category"])
    else:
        lab2d = np.zeros((img_dim[1], img_dim[0]), dtype=np.uint8)
    locs, feats, _, inds_reconstruct, _ = voxelize.voxelize(locals_3d, feature_3d, label_3d, voxel_size)
    feats = feats[:, :3]
    coords = np.concatenate([np.ones((locs.shape[0], 1), np.int32), locs.astype(np.int32)], axis=1)
    feats_out = (feats.astype(np.float32) / 255.0) if input_color else np.ones((coords.shape[0], 3), np.float32)
    x_label = mapping[:, 1][mapping[:, 1] != 0]
    y_label = mapping[:, 2][mapping[:, 2] != 0]
    locals_out = np.concatenate([np.ones((locals_3d.shape[0], 1), np.float32), locals_3d.astype(np.float32)], axis=1)
    scene_locs, _, _, scene_inv, _ = voxelize.voxelize(locs_in, point_features, labels_in, voxel_size)
    return (locs_in.astype(np.float32), scene_locs.astype(np.float32), scene_inv.astype(np.int64), labels_in.astype(np.int64),
            locals_out, coords, feats_out, feature_3d.astype(np.float32), label_3d.astype(np.int64), binary.astype(np.float32),
            lab2d.astype(np.int64), img, x_label.astype(np.int64), y_label.astype(np.int64), mask.astype(bool),
            inds_reconstruct.astype(np.int64), unique_map.astype(np.int64), mapping, None, point_features.astype(np.float32))


def fused_feature_item(locs_in, cols, labs, processed, *, split, voxel_size, n_occur=1, input_color=True, eval_all=True):
    """dataset/feature_loader.py:66-218 after the file reads (memcache off, aug off): `processed` is the dict of the chosen
    fused-feature file (picked by np.random.randint(n_occur) when n_occur > 1, consumed here in the reference's order).
    Returns (coords i32 [Nv,4], feats f32, labels i64, feat_3d, mask bool[, inds_reconstruct i64]) as numpy arrays."""
    labels_in = np.array(labs, copy=True)
    labels_in[labels_in == -100] = 255
    labels_in = labels_in.astype(np.uint8)
    feats_in = (np.asarray(cols) + 1.0) * 127.5
    if n_occur > 1:
        np.random.randint(n_occur)                           # (the caller has already resolved `processed` with the same draw)
    two_key = len(processed) == 2
    if two_key:
        feat_3d, mask_chunk = np.asarray(processed["feat"]), np.asarray(processed["mask_full"]).astype(bool)
        mask = mask_chunk.copy()
        if split != "train":
            full = np.zeros((locs_in.shape[0], feat_3d.shape[1]), dtype=feat_3d.dtype)
            full[mask] = feat_3d
            feat_3d, mask_chunk = full, np.ones_like(mask_chunk)
    else:
        feat_3d, mask_chunk = np.asarray(processed["feat"]), np.asarray(processed["mask_full"]).astype(bool).copy()
        mask = np.zeros(feat_3d.shape[0], dtype=bool)
        mask[np.asarray(processed["mask"])] = True
    if feat_3d.ndim > 2:
        feat_3d = feat_3d[..., 0]
    if split == "train":
        if not two_key:
            feat_3d = feat_3d[mask]
            sub = mask_chunk.copy()
            mask_chunk[sub] = mask                           # only the visible chunk points keep a feature row
        locs, feats, labels, inv, vox_ind = voxelize.voxelize(locs_in.astype(np.float64), feats_in, labels_in, voxel_size)
        mask = mask_chunk[vox_ind]
        rank = np.cumsum(mask_chunk.astype(np.int64)) - 1    # row in feat_3d of every chunk point
        feat_3d = feat_3d[rank[vox_ind[mask_chunk[vox_ind]]]]
    else:
        locs, feats, labels, inv, vox_ind = voxelize.voxelize(locs_in[mask_chunk].astype(np.float64), feats_in[mask_chunk],
                                                              labels_in[mask_chunk], voxel_size)
        feat_3d = feat_3d[vox_ind]
        mask = mask[vox_ind]
    if eval_all:
        labels = labels_in
    coords = np.concatenate([np.ones((locs.shape[0], 1), np.int32), locs.astype(np.int32)], axis=1)
    feats = (feats.astype(np.float32) / 127.5 - 1.0) if input_color else np.ones((coords.shape[0], 3), np.float32)
    out = (coords, feats.astype(np.float32), labels.astype(np.int64), feat_3d, mask)
    return out + (inv.astype(np.int64),) if eval_all else out
