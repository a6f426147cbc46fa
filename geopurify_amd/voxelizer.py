"""Voxelizer front end on the HIP kernels -- host mirror of dataset/voxelizer.py.

Same constructor and voxelize() contract as the reference class, but the per-point work (fp64
affine map, floor, min shift, FNV-1 hash, sort/unique) runs in gp_voxelize_f64.  Only the 4x4
matrices are produced on the host; they draw from np.random in the reference's order
(dataset/voxelizer.py:32-58: one angle per axis, a shuffle of the three rotations, one scale), so a
seeded run lands on the reference's voxel grid bit for bit.
"""
import numpy as np
import torch
from scipy.linalg import expm

from . import ops

# Point3DLoader's class constants (dataset/point_loader.py:54-60); its voxelizer always augments (:100-107)
SCALE_AUGMENTATION_BOUND = (0.9, 1.1)
ROTATION_AUGMENTATION_BOUND = ((-np.pi / 64, np.pi / 64), (-np.pi / 64, np.pi / 64), (-np.pi, np.pi))
TRANSLATION_AUGMENTATION_RATIO_BOUND = ((-0.2, 0.2), (-0.2, 0.2), (0, 0))


def _axis_rotation(axis_index, theta):
    """Rotation about a coordinate axis as the matrix exponential of the cross-product generator."""
    unit = np.zeros(3)
    unit[axis_index] = 1.0
    return expm(np.cross(np.eye(3), unit / np.linalg.norm(unit) * theta))


class Voxelizer:
    def __init__(self, voxel_size=1, clip_bound=None, use_augmentation=False, scale_augmentation_bound=None,
                 rotation_augmentation_bound=None, translation_augmentation_ratio_bound=None, ignore_label=255):
        self.voxel_size = voxel_size
        self.clip_bound = clip_bound
        self.use_augmentation = use_augmentation
        self.scale_augmentation_bound = scale_augmentation_bound
        self.rotation_augmentation_bound = rotation_augmentation_bound
        self.translation_augmentation_ratio_bound = translation_augmentation_ratio_bound
        self.ignore_label = ignore_label

    # -- matrices ------------------------------------------------------------------------------
    def get_transformation_matrix(self):
        """Returns (voxelization_matrix, rotation_matrix), both 4x4."""
        M_r = np.eye(4)
        bounds = self.rotation_augmentation_bound
        if self.use_augmentation and bounds is not None:
            if not hasattr(bounds, "__iter__"):
                raise ValueError("rotation_augmentation_bound must hold one (lo, hi) pair per axis")
            per_axis = [_axis_rotation(a, 0 if b is None else np.random.uniform(*b)) for a, b in enumerate(bounds)]
            np.random.shuffle(per_axis)
            M_r[:3, :3] = per_axis[0] @ per_axis[1] @ per_axis[2]
        s = 1 / self.voxel_size
        if self.use_augmentation and self.scale_augmentation_bound is not None:
            s *= np.random.uniform(*self.scale_augmentation_bound)
        M_v = np.eye(4)
        M_v[0, 0] = M_v[1, 1] = M_v[2, 2] = s
        return M_v, M_r

    # -- optional box clip (clip_bound is None everywhere on the hot path) -------------------------
    def clip(self, coords, center=None, trans_aug_ratio=None):
        lo, hi = coords.min(0).astype(float), coords.max(0).astype(float)
        size = hi - lo
        c = lo + 0.5 * size if center is None else center
        if trans_aug_ratio is not None:
            c = c + np.multiply(trans_aug_ratio, size)
        lim = np.asarray(self.clip_bound, dtype=float)
        return np.all((coords >= lim[:, 0] + c) & (coords < lim[:, 1] + c), axis=1)

    # -- np.random draws of one call, in the reference's order (voxelizer.py:86-104) -------------
    def _clip_box_and_matrices(self, lo, hi, center):
        """(clip box [2,3] or None, rigid 4x4, M_r): the translation ratios of the clip are drawn BEFORE the matrices, as in the
        reference; `rigid` is M_r @ M_v with augmentation and M_v alone without.  The host and the device form of voxelize share
        this, so a loader whose voxelizer is configured differently (no augmentation, a clip box) stays one code path."""
        box = None
        if self.clip_bound is not None:
            ratio = np.zeros(3)
            if self.use_augmentation and self.translation_augmentation_ratio_bound is not None:
                ratio = np.array([np.random.uniform(*b) for b in self.translation_augmentation_ratio_bound])
            size = hi - lo
            c = lo + 0.5 * size if center is None else center
            c = c + np.multiply(ratio, size)
            lim = np.asarray(self.clip_bound, dtype=float)
            box = np.stack([lim[:, 0] + c, lim[:, 1] + c])
        M_v, M_r = self.get_transformation_matrix()
        return box, (M_r @ M_v if self.use_augmentation else M_v), M_r

    def voxelize_device(self, coords, feats=None, labels=None, center=None):
        """The same call with device tensors in and out (coords f64 [N,3] on the GPU): dict(coords_aug f64 [Nv,3], inds i64 [Nv]
        -- indices into the CLIPPED cloud, as in the reference --, inds_reconstruct i64, feats[inds] with the normals of
        columns 3:6 rotated when there are more than six, labels[inds], M_r)."""
        if not (coords.dim() == 2 and coords.shape[1] == 3 and coords.shape[0]):
            raise AssertionError("voxelize: need N>0 points with 3 coordinates and one feature row each")
        lo, hi = (coords.amin(0).cpu().numpy().astype(float), coords.amax(0).cpu().numpy().astype(float)) \
            if self.clip_bound is not None else (None, None)
        box, rigid, M_r = self._clip_box_and_matrices(lo, hi, center)
        if box is not None:
            b = torch.from_numpy(box).to(coords.device)
            keep = ((coords >= b[0]) & (coords < b[1])).all(1)
            if bool(keep.any()):
                coords = coords[keep]
                feats = feats[keep] if feats is not None else None
                labels = labels[keep] if labels is not None else None
        r = ops.voxelize(coords.contiguous(), rigid)
        out = {"coords_aug": r["coords_aug"], "inds": r["inds"], "inds_reconstruct": r["inds_reconstruct"], "M_r": M_r,
               "feats": None, "labels": None}
        if feats is not None:
            f = feats[r["inds"]]
            if f.shape[1] > 6:                           # normals ride in columns 3:6 and rotate with the cloud
                rot = torch.from_numpy(M_r[:3, :3].T.copy()).to(f.device)
                f[:, 3:6] = (f[:, 3:6].double() @ rot).to(f.dtype)
            out["feats"] = f
        if labels is not None:
            out["labels"] = labels[r["inds"]]
        return out

    # -- the hot call ----------------------------------------------------------------------------
    def voxelize(self, coords, feats, labels, center=None, link=None, return_ind=False):
        """(coords_aug f64 [Nv,3], feats[inds], labels[inds], inds_reconstruct[, inds | link[inds]])."""
        if not (coords.shape[1] == 3 and coords.shape[0] == feats.shape[0] and coords.shape[0]):
            raise AssertionError("voxelize: need N>0 points with 3 coordinates and one feature row each")
        lo, hi = (coords.min(0).astype(float), coords.max(0).astype(float)) if self.clip_bound is not None else (None, None)
        box, rigid, M_r = self._clip_box_and_matrices(lo, hi, center)
        if box is not None:
            keep = np.all((coords >= box[0]) & (coords < box[1]), axis=1)
            if keep.sum():
                coords, feats = coords[keep], feats[keep]
                labels = labels[keep] if labels is not None else None
        dev_coords = torch.as_tensor(np.ascontiguousarray(coords, dtype=np.float64)).cuda()
        r = ops.voxelize(dev_coords, rigid)
        inds = r["inds"].cpu().numpy()
        inverse = r["inds_reconstruct"].cpu().numpy()
        out_feats, out_labels = feats[inds], labels[inds]
        if out_feats.shape[1] > 6:                       # normals ride in columns 3:6 and rotate with the cloud
            out_feats[:, 3:6] = out_feats[:, 3:6] @ M_r[:3, :3].T
        head = (r["coords_aug"].cpu().numpy(), out_feats, out_labels, inverse)
        if return_ind:
            return head + (inds,)
        if link is not None:
            return head + (link[inds],)
        return head


def default_voxelizer(voxel_size):
    """The instance Point3DLoader builds: augmentation on, no clipping."""
    return Voxelizer(voxel_size=voxel_size, clip_bound=None, use_augmentation=True,
                     scale_augmentation_bound=SCALE_AUGMENTATION_BOUND,
                     rotation_augmentation_bound=ROTATION_AUGMENTATION_BOUND,
                     translation_augmentation_ratio_bound=TRANSLATION_AUGMENTATION_RATIO_BOUND)
