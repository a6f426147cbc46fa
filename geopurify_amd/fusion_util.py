"""Host mirror of models/utils/fusion_util.py: the two point->pixel mappers with the reference's
constructors and compute_mapping() signatures, running on the HIP projection kernel
(gp_project_points_f64).  numpy in, numpy out."""
import math

import numpy as np
import torch

from . import ops


def make_intrinsic(fx, fy, mx, my):
    intrinsic = np.eye(4)
    intrinsic[0][0], intrinsic[1][1], intrinsic[0][2], intrinsic[1][2] = fx, fy, mx, my
    return intrinsic


def adjust_intrinsic(intrinsic, intrinsic_image_dim, image_dim):
    """models/utils/fusion_util.py:18-33."""
    if intrinsic_image_dim == image_dim:
        return intrinsic
    resize_width = int(math.floor(image_dim[1] * float(intrinsic_image_dim[0]) / float(intrinsic_image_dim[1])))
    intrinsic[0, 0] *= float(resize_width) / float(intrinsic_image_dim[0])
    intrinsic[1, 1] *= float(image_dim[1]) / float(intrinsic_image_dim[1])
    intrinsic[0, 2] *= float(image_dim[0] - 1) / float(intrinsic_image_dim[0] - 1)
    intrinsic[1, 2] *= float(image_dim[1] - 1) / float(intrinsic_image_dim[1] - 1)
    return intrinsic


def _run(w2c, coords, depth, K, image_dim, cut, tau, want_weight):
    c = torch.as_tensor(np.ascontiguousarray(coords, dtype=np.float64)).cuda()
    if isinstance(depth, str):
        if not want_weight:                              # only the ScanNet mapper has the render branch (:126-130)
            raise TypeError("the Matterport mapper takes a depth map or None (fusion_util.py:45-82)")
        d = ops.render_depth(c, w2c, K[0][0], K[1][1], K[0][2], K[1][2], image_dim[0], image_dim[1], cut)
    else:
        d = None if depth is None else torch.as_tensor(np.ascontiguousarray(depth, dtype=np.float64)).cuda()
    out = ops.project_points(c, w2c, K[0][0], K[1][1], K[0][2], K[1][2], d, image_dim[0], image_dim[1], cut, tau,
                             want_weight=want_weight)
    if want_weight:
        return out[0].cpu().numpy(), out[1].cpu().numpy()
    return out.cpu().numpy()


class PointCloudToImageMappermatterport(object):
    """fusion_util.py:36-82: argument is a camera_to_world matrix (inverted on the host, as the reference)."""

    def __init__(self, image_dim, visibility_threshold=0.25, cut_bound=0, intrinsics=None):
        self.image_dim = image_dim
        self.vis_thres = visibility_threshold
        self.cut_bound = cut_bound
        self.intrinsics = intrinsics

    def compute_mapping(self, camera_to_world, coords, depth=None, intrinsic=None):
        if self.intrinsics is not None:
            intrinsic = self.intrinsics
        w2c = np.linalg.inv(camera_to_world)
        return _run(w2c, coords, depth, intrinsic, self.image_dim, self.cut_bound, self.vis_thres, False)


class PointCloudToImageMapper(object):
    """fusion_util.py:85-147 (ScanNet): argument is world_view_transform = W2C^T; intrinsics are
    rescaled to image_dim in the constructor."""

    def __init__(self, image_dim, visibility_threshold=0.25, cut_bound=0, intrinsics=None):
        self.image_dim = image_dim
        self.vis_thres = visibility_threshold
        self.cut_bound = cut_bound
        self.intrinsics = np.array(intrinsics).copy()
        scale_x = self.image_dim[0] / (self.intrinsics[0, 2] * 2)
        scale_y = self.image_dim[1] / (self.intrinsics[1, 2] * 2)
        self.intrinsics[0, 0] *= scale_x
        self.intrinsics[1, 1] *= scale_y
        self.intrinsics[0, 2] = self.image_dim[0] / 2
        self.intrinsics[1, 2] = self.image_dim[1] / 2

    def compute_mapping(self, world_to_camera, coords, depth=None, intrinsic=None):
        if self.intrinsics is not None:
            intrinsic = self.intrinsics
        w2c = np.asarray(world_to_camera).T
        return _run(w2c, coords, depth, intrinsic, self.image_dim, self.cut_bound, self.vis_thres, True)
