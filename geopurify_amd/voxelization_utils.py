"""Host mirror of dataset/voxelization_utils.py (reference API: fnv_hash_vec, sparse_quantize) on
the HIP voxelizer kernels.  numpy in, numpy out; the arithmetic runs on the GPU."""
import numpy as np
import torch

from . import ops


def _dev(a, dtype):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=dtype)).cuda()


def fnv_hash_vec(arr):
    """dataset/voxelization_utils.py:6-18 -- FNV-1 over the 3 columns of integer-valued coords -> uint64."""
    arr = np.asarray(arr)
    assert arr.ndim == 2
    if arr.shape[1] != 3:
        raise ValueError("the HIP FNV kernel hashes 3-column coordinates (the voxelizer's only use)")
    return ops.fnv_hash(_dev(arr, np.float64)).cpu().numpy().view(np.uint64)


def sparse_quantize(coords, feats=None, labels=None, ignore_label=255, set_ignore_label_when_collision=False,
                    return_index=False, hash_type="fnv", quantization_size=1):
    """dataset/voxelization_utils.py:38-102, the path the voxelizer uses (no feats/labels => index
    outputs, FNV hash): returns (inds, inds_reverse) in ascending-hash voxel order."""
    assert hash_type == "fnv", "only the FNV path (the one on the hot path) is implemented"
    assert feats is None and labels is None, "feature/label filtering is done by the caller from `inds`"
    coords = np.asarray(coords, dtype=np.float64)
    assert coords.ndim == 2 and coords.shape[1] == 3
    q = np.asarray(quantization_size, dtype=np.float64) * np.ones(3)
    c = np.floor(coords / q)
    if (c.min(0) != 0).any():
        # the reference hashes the coordinates as they are; the HIP voxelizer shifts by the minimum
        # first (as Voxelizer.voxelize does before calling this function), which only coincides for min == 0
        raise NotImplementedError("sparse_quantize on the HIP path expects min-shifted coordinates "
                                  "(the voxelizer's call site, dataset/voxelizer.py:119-121)")
    r = ops.voxelize(_dev(c, np.float64), np.eye(4))
    return r["inds"].cpu().numpy(), r["inds_reconstruct"].cpu().numpy()
