"""compat shim: `dataset.voxelizer`."""
from geopurify_amd.voxelizer import *  # noqa: F401,F403
from geopurify_amd.voxelizer import Voxelizer  # noqa: F401
