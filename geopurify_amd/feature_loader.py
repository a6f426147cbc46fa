"""Host mirror of dataset/feature_loader.py (+ the parts of dataset/point_loader.py it needs):
OpenScene-style loader of pre-fused per-point features.

Same constructor and __getitem__ contract as the reference FusedFeatureLoader
(dataset/feature_loader.py:11-218) and the same collate functions (:221-255); the voxelization runs on
the HIP voxelizer (geopurify_amd.voxelizer.Voxelizer, augmentation always on as in Point3DLoader).
On-disk formats: scene `.pth` = (coords, colors in [-1,1], labels) (dataset/scripts/preprocess/
preprocess_3d_scannet.py:19-25); features `{scene}_{k}.pt` = {"feat": [n_seen,D], "mask_full": [N] bool}
or the three-key form with an extra "mask".  Out of scope: the SharedArray
/dev/shm cache (memcache_init) and the train-time colour/elastic augmentations (aug=True).

`device=` (not in the reference): the item is decoded ON the device -- the file contents are
uploaded once, the voxelizer's outputs stay there, the chunk-mask bookkeeping and the feature-row selection are
gp_fused_decode (csrc/fused_decode.hip) -- and returned as device tensors with the same values, shapes and dtypes.
"""
from glob import glob
from os.path import join

import numpy as np
import torch

from .voxelizer import default_voxelizer


class FusedFeatureLoader(torch.utils.data.Dataset):
    def __init__(self, datapath_prefix, datapath_prefix_feat, voxel_size=0.05, split="train", aug=False,
                 memcache_init=False, identifier=7791, loop=1, eval_all=False, input_color=False, device=None):
        super().__init__()
        self.device = torch.device(device) if device is not None else None
        if aug:
            raise NotImplementedError("train-time augmentations (dataset/augmentation.py) are out of scope")
        if memcache_init:
            raise NotImplementedError("the SharedArray /dev/shm cache is out of scope")
        self.split = split
        self.identifier = identifier
        self.data_paths = sorted(glob(join(datapath_prefix, split or "", "*.pth")))
        if len(self.data_paths) == 0:
            raise Exception("0 file is loaded in the point loader.")
        self.input_color = input_color
        self.voxel_size = voxel_size
        self.aug = aug
        self.loop = loop
        self.eval_all = eval_all
        self.dataset_name = datapath_prefix.split("/")[-1]
        self.use_shm = False
        self.voxelizer = default_voxelizer(voxel_size)
        self.datapath_feat = datapath_prefix_feat
        # scenes without fused features are dropped (feature_loader.py:41-61)
        if "nuscenes" in self.dataset_name:
            self.list_occur = None
        else:
            occ = [len(glob(join(self.datapath_feat, self._scene_name(p) + "_*.pt"))) for p in self.data_paths]
            keep = [i for i, n in enumerate(occ) if n != 0]
            self.data_paths = [self.data_paths[i] for i in keep]
            self.list_occur = [occ[i] for i in keep]
        if len(self.data_paths) == 0:
            raise Exception("0 file is loaded in the feature loader.")

    def _scene_name(self, path):
        return path[:-15].split("/")[-1] if "scannet" in self.dataset_name else path[:-4].split("/")[-1]

    def __len__(self):
        return len(self.data_paths) * self.loop

    def __getitem__(self, index_long):
        index = index_long % len(self.data_paths)
        locs_in, feats_in, labels_in = torch.load(self.data_paths[index], weights_only=False)
        labels_in[labels_in == -100] = 255
        labels_in = labels_in.astype(np.uint8)
        feats_in = np.zeros_like(locs_in) if (np.isscalar(feats_in) and feats_in == 0) else (feats_in + 1.0) * 127.5
        scene_name = self.data_paths[index][:-15].split("/")[-1] if self.dataset_name == "scannet_3d" \
            else self.data_paths[index][:-4].split("/")[-1]
        if "nuscenes" not in self.dataset_name:
            n_occur = self.list_occur[index]
            if n_occur < 1:
                raise NotImplementedError
            nn_occur = np.random.randint(n_occur) if n_occur > 1 else 0
            processed = torch.load(join(self.datapath_feat, scene_name + "_%d.pt" % nn_occur), weights_only=False)
        else:
            processed = torch.load(join(self.datapath_feat, scene_name + ".pt"), weights_only=False)
        return self._item(locs_in, feats_in, labels_in, processed)


def _as_bool(x):
    return torch.from_numpy(x).bool() if isinstance(x, np.ndarray) else x.bool()


def _decode_host(mask_chunk, feat, vox_ind, mode, row_keep=None):
    """Host form of gp_fused_decode (same contract: csrc/fused_decode.hip): rank(p) = row of point p in `feat`."""
    rank = torch.cumsum(mask_chunk.to(torch.int64), dim=0) - 1
    inside = mask_chunk[vox_ind]
    rows = rank[vox_ind]
    keep = inside if row_keep is None else inside & row_keep[rows.clamp(min=0)]
    if mode == 0:
        return feat[rows[keep]], keep
    out = torch.zeros((vox_ind.shape[0],) + tuple(feat.shape[1:]), dtype=feat.dtype)
    out[inside] = feat[rows[inside]]
    return out, keep


def _item(self, locs_in, feats_in, labels_in, processed):
    """__getitem__ past the file reads (dataset/feature_loader.py:113-218), on the host or -- with device= -- on the device.
    One formulation for the four cases (two-key / three-key file, training / evaluation): with rank(p) = row of point p in
    `feat`, a voxel keeps its representative's row if the representative is in the chunk (and, three-key form, was seen).
    np.random is drawn by the voxelizer's matrices only, as in the reference."""
    from . import ops
    dev = self.device
    two_key = len(processed.keys()) == 2
    train = self.split == "train"
    feat = processed["feat"]
    if feat.dim() > 2:
        feat = feat[..., 0]
    mask_chunk = _as_bool(processed["mask_full"])
    row_keep = None
    if not two_key:                                            # "mask": the chunk rows a camera saw (index list or bool mask)
        mv = processed["mask"]
        row_keep = torch.zeros(feat.shape[0], dtype=torch.bool)
        row_keep[torch.from_numpy(mv) if isinstance(mv, np.ndarray) else mv] = True
    chunk_only = not train and not two_key                   # :183-187: the evaluation form of a three-key file voxelizes the chunk only
    if dev is None:
        mc = mask_chunk.numpy().astype(bool)
        pts, cols, labs = (locs_in[mc], feats_in[mc], labels_in[mc]) if chunk_only else (locs_in, feats_in, labels_in)
        locs, feats, labels, inds_reconstruct, vox_ind = self.voxelizer.voxelize(pts, cols, labs, return_ind=True)
        vox_ind = torch.from_numpy(vox_ind)
        decode = _decode_host
        coords = torch.from_numpy(locs).int()
        feats = torch.from_numpy(feats).float() / 127.5 - 1.0 if self.input_color else torch.ones(coords.shape[0], 3)
        labels = torch.from_numpy(labels_in if self.eval_all else labels).long()
        inds_reconstruct = torch.from_numpy(inds_reconstruct).long()
        ones = torch.ones((coords.shape[0], 1), dtype=torch.int)
    else:
        feat, mask_chunk = feat.to(dev), mask_chunk.to(dev)
        row_keep = row_keep.to(dev) if row_keep is not None else None
        pts = torch.as_tensor(np.ascontiguousarray(locs_in, dtype=np.float64)).to(dev)
        cols = torch.as_tensor(np.ascontiguousarray(feats_in)).to(dev)
        labs = torch.as_tensor(np.ascontiguousarray(labels_in)).to(dev)
        labs_v = labs
        if chunk_only:
            pts, cols, labs_v = pts[mask_chunk], cols[mask_chunk], labs[mask_chunk]
        # the voxelizer's own device form: clip box, augmentation switch and normal rotation are those of the host call
        r = self.voxelizer.voxelize_device(pts.contiguous(), cols, labs_v)
        vox_ind = r["inds"]
        decode = ops.fused_decode
        coords = r["coords_aug"].to(torch.int32)
        if self.input_color:
            # tensor / tensor: an IEEE division, as on the host (torch's tensor / Python-scalar on the GPU multiplies by the reciprocal)
            c = r["feats"].float()
            feats = c / torch.full_like(c, 127.5) - 1.0
        else:
            feats = torch.ones((coords.shape[0], 3), device=dev)
        labels = (labs if self.eval_all else r["labels"]).long()
        inds_reconstruct = r["inds_reconstruct"].long()
        ones = torch.ones((coords.shape[0], 1), dtype=torch.int32, device=dev)
    if train:
        feat_3d, mask = decode(mask_chunk, feat, vox_ind, 0, row_keep)
    elif two_key:
        feat_3d, mask = decode(mask_chunk, feat, vox_ind, 1)
    else:
        feat_3d, mask = decode(torch.ones(feat.shape[0], dtype=torch.bool, device=feat.device), feat, vox_ind, 1, row_keep)
    coords = torch.cat((ones, coords), dim=1)
    if self.eval_all:
        return coords, feats, labels, feat_3d, mask, inds_reconstruct
    return coords, feats, labels, feat_3d, mask


FusedFeatureLoader._item = _item


def collation_fn(batch):
    """feature_loader.py:221-234 (note: column 0 is MULTIPLIED by the batch index, as in the reference)."""
    coords, feats, labels, feat_3d, mask_chunk = list(zip(*batch))
    for i in range(len(coords)):
        coords[i][:, 0] *= i
    return torch.cat(coords), torch.cat(feats), torch.cat(labels), torch.cat(feat_3d), torch.cat(mask_chunk)


def collation_fn_eval_all(batch):
    """feature_loader.py:237-255."""
    coords, feats, labels, feat_3d, mask, inds_recons = list(zip(*batch))
    inds_recons = list(inds_recons)
    acc = 0
    for i in range(len(coords)):
        coords[i][:, 0] *= i
        inds_recons[i] = acc + inds_recons[i]
        acc += coords[i].shape[0]
    return (torch.cat(coords), torch.cat(feats), torch.cat(labels), torch.cat(feat_3d), torch.cat(mask),
            torch.cat(inds_recons))
