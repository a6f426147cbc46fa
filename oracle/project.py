"""Oracle row 3: point -> pixel mapping with depth-consistency occlusion test (test infrastructure).

Follows models/utils/fusion_util.py:86-147 (ScanNet "gaussian" mapper, argument is the
world_view_transform = W2C^T) and :36-82 (Matterport mapper, argument is a camera-to-world
matrix that is inverted).  Output rows are (v=row, u=col, 1) for visible points, (0,0,0) otherwise.
"""
import numpy as np


def scannet_intrinsics(image_dim, intrinsics):
    """fusion_util.py:86-96 -- rescale K to image_dim assuming cx,cy are half the native size."""
    K = np.array(intrinsics, dtype=np.float64).copy()
    sx = image_dim[0] / (K[0, 2] * 2)
    sy = image_dim[1] / (K[1, 2] * 2)
    K[0, 0] *= sx
    K[1, 1] *= sy
    K[0, 2] = image_dim[0] / 2
    K[1, 2] = image_dim[1] / 2
    return K


def _project(w2c, coords, K):
    homo = np.concatenate([coords, np.ones([coords.shape[0], 1])], axis=1).T
    p = np.matmul(w2c, homo)
    p[0] = (p[0] * K[0][0]) / p[2] + K[0][2]
    p[1] = (p[1] * K[1][1]) / p[2] + K[1][2]
    with np.errstate(invalid="ignore"):
        pi = np.round(p).astype(int)
    return p, pi


def _finish(p, pi, image_dim, cut, depth, tau):
    n = p.shape[1]
    inside = ((pi[0] >= cut) * (pi[1] >= cut)
              * (pi[0] < image_dim[0] - cut) * (pi[1] < image_dim[1] - cut))
    if depth is not None:
        d = depth[pi[1][inside], pi[0][inside]]
        occ = np.abs(d - p[2][inside]) <= tau * d
        inside[inside == True] = occ  # noqa: E712
    else:
        inside = (p[2] > 0) * inside
    mapping = np.zeros((3, n), dtype=int)
    mapping[0][inside] = pi[1][inside]
    mapping[1][inside] = pi[0][inside]
    mapping[2][inside] = 1
    return mapping.T


def render_depth(p, pi, image_dim, cut):
    """fusion_util.py:126-130 (depth passed as a str): z-buffer of the cloud itself, 999999 where nothing lands."""
    inside = ((pi[0] >= cut) * (pi[1] >= cut) * (pi[0] < image_dim[0] - cut) * (pi[1] < image_dim[1] - cut))
    depth = np.ones((image_dim[1], image_dim[0])) * 999999
    ok = inside & (p[2] > 0.2)
    np.minimum.at(depth, (pi[1][ok], pi[0][ok]), p[2][ok])      # the reference's sequential loop keeps the minimum
    return depth


def compute_mapping_scannet(world_view_transform, coords, depth, K, image_dim, cut, tau):
    """fusion_util.py:99-147.  K is the already-rescaled intrinsics (scannet_intrinsics)."""
    p, pi = _project(np.asarray(world_view_transform).T, coords, K)
    if isinstance(depth, str):
        depth = render_depth(p, pi, image_dim, cut)
    mapping = _finish(p, pi, image_dim, cut, depth, tau)
    dist = np.sqrt((pi[0] - image_dim[0] / 2) ** 2 + (pi[1] - image_dim[1] / 2) ** 2)
    return mapping, np.exp(-dist / 10)


def compute_mapping_matterport(camera_to_world, coords, depth, K, image_dim, cut, tau):
    """fusion_util.py:45-82."""
    w2c = np.linalg.inv(camera_to_world)
    p, pi = _project(w2c, coords, K)
    return _finish(p, pi, image_dim, cut, depth, tau)
