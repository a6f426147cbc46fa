"""compat shim: `models.utils.fusion_util`."""
from geopurify_amd.fusion_util import *  # noqa: F401,F403
from geopurify_amd.fusion_util import (PointCloudToImageMapper, PointCloudToImageMappermatterport,  # noqa: F401
                                       adjust_intrinsic, make_intrinsic)
