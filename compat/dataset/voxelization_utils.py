"""compat shim: `dataset.voxelization_utils`."""
from geopurify_amd.voxelization_utils import *  # noqa: F401,F403
from geopurify_amd.voxelization_utils import fnv_hash_vec, sparse_quantize  # noqa: F401
