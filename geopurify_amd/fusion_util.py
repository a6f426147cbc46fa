"""Host mirror of models/utils/fusion_util.py: the two point->pixel mappers with the reference's
constructors and compute_mapping() signatures, running on the HIP projection kernel
(gp_project_points_f64).  numpy in, numpy out."""
import math

import numpy as np
import torch

from . import ops


def make_intrinsic(fx, fy, mx, my):
    """4x4 pinhole matrix: focal lengths on the diagonal, principal point in the third column."""
    K = np.eye(4)
    K[(0, 1, 0, 1), (0, 1, 2, 2)] = (fx, fy, mx, my)
    return K


def adjust_intrinsic(intrinsic, intrinsic_image_dim, image_dim):
    """Rescale a pinhole matrix IN PLACE from images of intrinsic_image_dim (w, h) to image_dim (models/utils/fusion_util.py:18-33,
    same arithmetic): fy by the height ratio, fx by the aspect-preserving width over the old width, the principal point by the
    ratios of the last pixel indices."""
    if intrinsic_image_dim == image_dim:
        return intrinsic
    w_src, h_src = float(intrinsic_image_dim[0]), float(intrinsic_image_dim[1])
    w_keep_aspect = int(math.floor(image_dim[1] * w_src / h_src))
    for r, c, factor in ((0, 0, float(w_keep_aspect) / w_src),
                         (1, 1, float(image_dim[1]) / h_src),
                         (0, 2, float(image_dim[0] - 1) / float(intrinsic_image_dim[0] - 1)),
                         (1, 2, float(image_dim[1] - 1) / float(intrinsic_image_dim[1] - 1))):
        intrinsic[r, c] *= factor
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


class _Mapper(object):
    """What both mappers keep: target image size (w, h), relative depth tolerance, border margin, optional global intrinsics."""

    def __init__(self, image_dim, visibility_threshold=0.25, cut_bound=0, intrinsics=None):
        self.image_dim, self.vis_thres, self.cut_bound, self.intrinsics = image_dim, visibility_threshold, cut_bound, intrinsics

    def _K(self, intrinsic):
        return self.intrinsics if self.intrinsics is not None else intrinsic


class PointCloudToImageMappermatterport(_Mapper):
    """fusion_util.py:36-82: argument is a camera_to_world matrix (inverted on the host, as the reference)."""

    def compute_mapping(self, camera_to_world, coords, depth=None, intrinsic=None):
        return _run(np.linalg.inv(camera_to_world), coords, depth, self._K(intrinsic), self.image_dim, self.cut_bound, self.vis_thres, False)


class PointCloudToImageMapper(_Mapper):
    """fusion_util.py:85-147 (ScanNet): argument is world_view_transform = W2C^T; the constructor rescales the intrinsics to
    image_dim on the reference's assumption that the principal point sits at half the source size (:91-96)."""

    def __init__(self, image_dim, visibility_threshold=0.25, cut_bound=0, intrinsics=None):
        K = np.array(intrinsics).copy()
        for axis in (0, 1):
            K[axis, axis] *= image_dim[axis] / (K[axis, 2] * 2)
            K[axis, 2] = image_dim[axis] / 2
        super().__init__(image_dim, visibility_threshold, cut_bound, K)

    def compute_mapping(self, world_to_camera, coords, depth=None, intrinsic=None):
        return _run(np.asarray(world_to_camera).T, coords, depth, self._K(intrinsic), self.image_dim, self.cut_bound, self.vis_thres, True)
