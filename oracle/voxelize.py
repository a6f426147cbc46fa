"""Oracle rows 1-2: Voxelizer.voxelize + FNV-1 hash dedupe (test infrastructure).

Follows dataset/voxelizer.py:32-58,81-132 and dataset/voxelization_utils.py:6-18,38-102.
"""
import numpy as np
from scipy.linalg import expm, norm

FNV_OFFSET = np.uint64(14695981039346656037)
FNV_PRIME = np.uint64(1099511628211)

# dataset/point_loader.py:54-60
SCALE_AUGMENTATION_BOUND = (0.9, 1.1)
ROTATION_AUGMENTATION_BOUND = (
    (-np.pi / 64, np.pi / 64),
    (-np.pi / 64, np.pi / 64),
    (-np.pi, np.pi),
)


def fnv_hash_vec(arr):
    """dataset/voxelization_utils.py:6-18 -- multiply-then-xor (FNV-1) over columns, mod 2^64."""
    assert arr.ndim == 2
    a = np.asarray(arr).astype(np.uint64)
    h = np.full(a.shape[0], FNV_OFFSET, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for j in range(a.shape[1]):
            h = h * FNV_PRIME
            h = np.bitwise_xor(h, a[:, j])
    return h


def sparse_quantize_index(coords):
    """dataset/voxelization_utils.py:80,85,95-97 with return_index=True, hash_type='fnv',
    quantization_size=1: voxels in ascending-hash order, first-occurrence index, rank."""
    key = fnv_hash_vec(np.floor(coords / np.array([1, 1, 1])))
    _, inds, inv = np.unique(key, return_index=True, return_inverse=True)
    return inds, inv


def axis_rotation(axis_ind, theta):
    """dataset/voxelizer.py:7-8."""
    axis = np.zeros(3)
    axis[axis_ind] = 1
    return expm(np.cross(np.eye(3), axis / norm(axis) * theta))


def get_transformation_matrix(voxel_size, use_augmentation=True,
                              scale_bound=SCALE_AUGMENTATION_BOUND,
                              rot_bound=ROTATION_AUGMENTATION_BOUND):
    """dataset/voxelizer.py:32-58.  Consumes np.random in the reference's order:
    3x uniform (theta per axis), shuffle of the 3 matrices, 1x uniform (scale)."""
    M_v, M_r = np.eye(4), np.eye(4)
    rot = np.eye(3)
    if use_augmentation and rot_bound is not None:
        mats = []
        for axis_ind, b in enumerate(rot_bound):
            theta = 0
            if b is not None:
                theta = np.random.uniform(*b)
            mats.append(axis_rotation(axis_ind, theta))
        np.random.shuffle(mats)
        rot = mats[0] @ mats[1] @ mats[2]
    M_r[:3, :3] = rot
    scale = 1 / voxel_size
    if use_augmentation and scale_bound is not None:
        scale *= np.random.uniform(*scale_bound)
    np.fill_diagonal(M_v[:3, :3], scale)
    return M_v, M_r


def voxelize_with_matrices(coords, M_v, M_r, use_augmentation=True):
    """dataset/voxelizer.py:103-121 given the two matrices.
    Returns coords_aug f64 [Nv,3] (integer valued, >=0), inds i64 [Nv], inds_reconstruct i64 [N]."""
    rigid = M_v
    if use_augmentation:
        rigid = M_r @ rigid
    homo = np.hstack((coords, np.ones((coords.shape[0], 1), dtype=coords.dtype)))
    c = np.floor(homo @ rigid.T[:, :3])
    c = np.floor(c - c.min(0))
    inds, inv = sparse_quantize_index(c)
    return c[inds], inds.astype(np.int64), np.asarray(inv).astype(np.int64), rigid


def voxelize(coords, feats, labels, voxel_size, use_augmentation=True):
    """dataset/voxelizer.py:81-132 (clip_bound=None path).  feats' normal columns 3:6 are
    rotated only if feats has more than 6 columns (:124-125)."""
    M_v, M_r = get_transformation_matrix(voxel_size, use_augmentation)
    c, inds, inv, _ = voxelize_with_matrices(coords, M_v, M_r, use_augmentation)
    f = feats[inds]
    l = labels[inds] if labels is not None else None
    if f.shape[1] > 6:
        f[:, 3:6] = f[:, 3:6] @ (M_r[:3, :3].T)
    return c, f, l, inv, inds
