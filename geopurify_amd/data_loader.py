"""Host mirror of dataset/data_loader_ablation.py / data_loader_matterport.py: the (scene, view) dataset surface, the
scene batch sampler and the collate that builds the positional 20-tuple evaluate_scene reads.

Reference: ScannetLoaderFull (data_loader_ablation.py:19-394), SceneBatchSampler (:401-421),
scene_based_collate_fn (:429-495).  Disk I/O (ScanNet `.pth` scenes, RGB-D frames, pose files) is out of scope
(SURVEY.md section 2 #4, #7): the dataset here serves the seeded synthetic scenes of geopurify_amd.synthetic, selected by
a `datapath_prefix` that starts with "synthetic" -- same constructor keywords, same `.samples` / `.data_paths`
attributes, same per-view sample tuple, with the mapper and both voxelizations running on the HIP kernels.
"""
import os
from collections import defaultdict
import random

import numpy as np
import torch
from torch.utils.data import Dataset, Sampler


class SceneBatchSampler(Sampler):
    """One batch = all views of one scene (data_loader_ablation.py:401-421)."""

    def __init__(self, samples_list, shuffle=True):
        self.samples_list, self.shuffle = samples_list, shuffle
        self.scene_to_indices = defaultdict(list)
        for idx, sample in enumerate(samples_list):
            self.scene_to_indices[sample["scene_name"]].append(idx)
        self.scenes = list(self.scene_to_indices.keys())

    def __iter__(self):
        if self.shuffle:
            random.shuffle(self.scenes)
        for name in self.scenes:
            yield self.scene_to_indices[name]

    def __len__(self):
        return len(self.scenes)


def scene_based_collate_fn(batch):
    """data_loader_ablation.py:429-495.  `batch`: the per-view sample tuples of ONE scene (None = dropped view).
    Scene-level slots (0-3, 19) come from the first surviving view; per-view slots are concatenated with the view
    index written into column 0 of the coordinate slots, `mask_2ds` becomes [V*N, 2] = (view index, visible) and
    `inds_reconstructs` is offset by the number of voxel rows of the preceding views."""
    views = [s for s in batch if s is not None]
    if not views:
        return None
    col = list(zip(*views))
    (locs_in, scene_locs, scene_inv, labels_in, locals_3d, coords, feats, feature_3d, labels, binary, label_2d, imgs,
     x_label, y_label, mask_2d, inds_rec, unique_map, mapping, captions, point_features) = col
    view_of_row = torch.cat([torch.full((m.shape[0],), i, dtype=torch.long, device=m.device) for i, m in enumerate(mask_2d)])
    batch_and_mask = torch.stack([view_of_row, torch.cat(list(mask_2d)).to(torch.int)], dim=1)
    coords, locals_3d, inds_rec = list(coords), list(locals_3d), list(inds_rec)
    voxel_rows_before = 0
    for i in range(len(views)):
        coords[i][:, 0] = i                       # in place, as the reference
        locals_3d[i][:, 0] = i
        inds_rec[i] += voxel_rows_before
        voxel_rows_before += coords[i].shape[0]
    return (locs_in[0], scene_locs[0], scene_inv[0], labels_in[0],
            torch.cat(locals_3d), torch.cat(coords), torch.cat(feats), torch.cat(feature_3d), torch.cat(labels), torch.cat(binary),
            torch.stack(label_2d, dim=0), torch.stack(imgs, dim=0), torch.cat(x_label), torch.cat(y_label),
            batch_and_mask, torch.cat(inds_rec), torch.cat(unique_map), torch.cat(mapping), captions, point_features[0])


def view_sample(locs_in, labels_in, point_features, world_view_transform, intrinsics, depth, image, label_img, *, dataset, img_dim,
                vis_thres, cut_bound, voxel_size, category_split, split, val_keep, label_2d_ids=None, input_color=False,
                min_visible=400):
    """The per-view sample tuple of the reference datasets after the scene cache and the image readers
    (data_loader_ablation.py:242-394; Matterport: data_loader_matterport.py:190-300) on arrays: the mapper and both
    voxelizations run on the HIP kernels, np.random is consumed in the reference's order (per-view voxelization, then the
    whole scene).  `image`: the view's RGB image already at img_dim ([H,W,3], uint8 or float); `label_img`: the 2D label
    image at img_dim (ScanNet) or None (zeros).  min_visible: the reference's 400; the synthetic T config lowers it.
    Returns None for a dropped view: no visible point, fewer than min_visible, or more than 65000 (split 'train') /
    val_keep (other splits) visible points (:254-255, :279-288)."""
    from .fusion_util import PointCloudToImageMapper, PointCloudToImageMappermatterport
    from .voxelizer import default_voxelizer
    mapping = np.ones([locs_in.shape[0], 4], dtype=int)
    if dataset == "scannet":
        mapper = PointCloudToImageMapper(img_dim, vis_thres, cut_bound, intrinsics)
        mapping[:, 1:4], _ = mapper.compute_mapping(world_view_transform, locs_in, depth)
    else:
        mapper = PointCloudToImageMappermatterport(img_dim, vis_thres, cut_bound)
        cam_to_world = np.asarray(world_view_transform).T                  # data_loader_matterport.py:213
        mapping[:, 1:4] = mapper.compute_mapping(cam_to_world, locs_in, depth, intrinsics)
    mask = mapping[:, 3]
    n_vis = int(mask.sum())
    if n_vis == 0:
        return None
    vis = mask == 1
    unique_map = mapping.copy()
    mapping = mapping[np.all(mapping != 0, axis=1)]
    label_3d, feature_3d, locals_3d = labels_in[vis].copy(), point_features[vis].copy(), locs_in[vis].copy()
    # binary_label IS the array the reference tests while writing into it (:265-274): a base class becomes 1 and is then
    # compared with the novel list as 1
    binary = labels_in[vis].copy()
    binary[np.isin(binary, category_split["base_category"])] = 1
    binary[np.isin(binary, category_split["novel_category"])] = 0
    upper = 65000 if split == "train" else val_keep
    if n_vis < min_visible or n_vis > upper:
        return None
    W, H = img_dim
    if label_img is not None and label_2d_ids is not None:               # ScanNet 2D labels (:297-322)
        ids = list(label_2d_ids) if split in ("val", "test") else [label_2d_ids[c] for c in category_split["base_category"]]
        lab2d = np.array(label_img, copy=True).astype(np.int32)
        lab2d[~np.isin(lab2d, ids)] = 255
        lut = np.arange(max(256, int(lab2d.max()) + 1))
        lut[np.asarray(ids, dtype=np.int64)] = np.arange(len(ids))
        lab2d = lut[lab2d]
        if split not in ("val", "test"):
            lab2d[lab2d == 255] = len(category_split["base_category"])
        label_2d = torch.from_numpy(lab2d).long()
    else:
        label_2d = torch.zeros((H, W), dtype=torch.long)
    vox = default_voxelizer(voxel_size)
    locs, feats, _, inds_reconstruct = vox.voxelize(locals_3d, feature_3d, label_3d)              # per-view voxelization (:324)
    coords = torch.cat((torch.ones(locs.shape[0], 1, dtype=torch.int), torch.from_numpy(locs).int()), dim=1)
    feats = torch.from_numpy(feats[:, :3]).float() / 255.0 if input_color else torch.ones(coords.shape[0], 3)
    scene_locs, _, _, scene_inv = vox.voxelize(locs_in, point_features, labels_in)                # whole scene (:364)
    locals_t = torch.from_numpy(locals_3d).float()
    return (torch.from_numpy(locs_in).float(), torch.from_numpy(scene_locs).float(), torch.from_numpy(scene_inv).long(),
            torch.from_numpy(labels_in).long(), torch.cat((torch.ones(locals_t.shape[0], 1), locals_t), dim=1), coords, feats,
            torch.from_numpy(feature_3d).float(), torch.from_numpy(label_3d).long(), torch.from_numpy(binary).float(),
            label_2d, torch.from_numpy(np.asarray(image)).float(),
            torch.from_numpy(mapping[:, 1][mapping[:, 1] != 0]).long(), torch.from_numpy(mapping[:, 2][mapping[:, 2] != 0]).long(),
            torch.from_numpy(mask).bool(), torch.from_numpy(inds_reconstruct).long(), torch.from_numpy(unique_map).long(),
            torch.from_numpy(mapping), None, torch.from_numpy(point_features).float())


class ScannetLoaderFull(Dataset):
    """(scene, view) samples of synthetic ScanNet-/Matterport-shaped scenes behind the reference's constructor.

    `datapath_prefix = "synthetic[:CONFIG[:NUM_SCENES[:SEED]]]"` (CONFIG in geopurify_amd.synthetic.CONFIGS, default "S").
    `specific_ids` filters by substring of the scene names, like the reference does on file names."""

    def __init__(self, datapath_prefix, datapath_prefix_2d=None, label_2d=None, category_split=None, scannet200=False,
                 val_keep=10000000, caption_path=None, entity_path=None, voxel_size=0.05, split="train", aug=False,
                 memcache_init=False, identifier=7791, loop=1, eval_all=False, input_color=False, specific_ids=None,
                 scene_config=None, device="cuda"):
        super().__init__()
        from . import synthetic as syn
        if not str(datapath_prefix).startswith("synthetic"):
            raise NotImplementedError(
                "ScannetLoaderFull: reading ScanNet/Matterport scenes from disk is out of scope (SURVEY.md section 2 #4); "
                "use datapath_prefix='synthetic[:CONFIG[:NUM_SCENES[:SEED]]]'")
        if aug or memcache_init:
            raise NotImplementedError("train-time augmentation and the SharedArray cache are out of scope")
        parts = str(datapath_prefix).split(":")
        self.cfg = syn.CONFIGS[parts[1] if len(parts) > 1 and parts[1] else "S"]
        n_scenes = int(parts[2]) if len(parts) > 2 else 2
        self.seed0 = int(parts[3]) if len(parts) > 3 else 5557
        self.category_split = category_split or {"base_category": [], "novel_category": [], "ignore_category": list(self.cfg.ignore_ids)}
        self.val_keep, self.voxel_size, self.split, self.loop = val_keep, voxel_size, split, loop
        self.input_color, self.scannet200, self.scene_config, self.device = input_color, scannet200, scene_config, device
        self.data_paths = [f"{datapath_prefix}/scene{i:04d}_00_vh_clean_2.pth" for i in range(n_scenes)]
        if specific_ids is not None:
            self.data_paths = [p for p in self.data_paths if any(s in p for s in specific_ids)]
        if len(self.data_paths) == 0:
            raise Exception("0 file is loaded in the feature loader.")
        self.samples = []
        for p in self.data_paths:
            name = os.path.basename(p).split("_vh_clean_2.pth")[0]
            for v in range(self.cfg.num_views):
                self.samples.append({"scene_data_path": p, "scene_name": name, "view": None, "view_idx": v, "intrinsics": None})
        self.scene_cache = {}

    def __len__(self):
        return len(self.samples) * self.loop

    def _scene(self, name):
        from . import synthetic as syn
        if name not in self.scene_cache:
            self.scene_cache.clear()                                   # one scene at a time, as the reference's cache is used
            idx = int(name[5:9])
            self.scene_cache[name] = syn.make_scene(self.cfg, self.seed0 + idx)
        return self.scene_cache[name]

    def __getitem__(self, index_long):
        """The per-view sample tuple of data_loader_ablation.py:373-394 (None when the view is dropped, :254-255,279-288)."""
        s = self.samples[index_long % len(self.samples)]
        scene, cfg = self._scene(s["scene_name"]), self.cfg
        view = scene.views[s["view_idx"]]
        point_features = np.concatenate([scene.colors, scene.normals], 1)        # rgb in [0,1] ++ normal (:163,214)
        W, H = cfg.image_dim
        img = np.full((H, W, 3), float(s["view_idx"]), dtype=np.float32)   # no RGB offline: the view index, for the VLM stand-in
        wvt = view.pose if cfg.dataset == "scannet" else np.ascontiguousarray(view.pose.T)   # Matterport poses are camera-to-world
        return view_sample(scene.coords, scene.labels.copy(), point_features, wvt, view.K, view.depth, img, None,
                           dataset=cfg.dataset, img_dim=cfg.image_dim, vis_thres=cfg.vis_thres, cut_bound=cfg.cut_bound,
                           voxel_size=self.voxel_size, category_split=self.category_split, split=self.split, val_keep=self.val_keep,
                           label_2d_ids=None, input_color=self.input_color, min_visible=cfg.min_visible)


class LookAheadLoader:
    """Wraps the evaluation DataLoader of run/validation.py:296-321 (any iterable of the positional 20-tuples) so that the drop-in
    call `model.evaluate_scene(batch_data)` (run/validation.py:408) runs at the rate of the device pipeline:

        for batch_data in LookAheadLoader(val_loader, model):      # instead of: for batch_data in val_loader:
            eval_results = model.evaluate_scene(batch_data)

    While scene i is being evaluated the wrapper has already (1) fetched tuple i + 1 from the loader, (2) copied its tensors to the
    device on its own copy stream (asynchronously when the loader pins its memory) and (3) OFFERED the device tuple to the trainer
    (SonataXAffinityTrainer.offer_next), which parses it, lifts it and runs HotPath.prepare on a side stream beside scene i's
    student.  The yielded tuple is the device tuple; every result equals the serial call's bit for bit.
    The copy of tuple i + 1 is ordered behind the end of scene i - 1 on the evaluating stream (`model.last_scene_done`): the blocks it
    is handed were last read there, so no record_stream markers are needed."""

    def __init__(self, loader, model, device=None, copy_stream=None):
        self.loader, self.model = loader, model
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        # (a process maps its streams onto a handful of hardware queues -- four by default -- and two streams on one queue run in order:
        # a program that already owns streams should hand one over instead of letting every helper create its own)
        self.copy_stream = copy_stream if copy_stream is not None else torch.cuda.Stream(device=self.device)

    def __len__(self):
        return len(self.loader)

    def _upload(self, batch_data):
        if batch_data is None:
            return None, None
        last = getattr(self.model, "last_scene_done", None)
        with torch.cuda.stream(self.copy_stream):
            if last is not None:
                self.copy_stream.wait_event(last)
            dev = tuple(x.to(self.device, non_blocking=True) if torch.is_tensor(x) and x.numel() else x for x in batch_data)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        return dev, ev

    def __iter__(self):
        it = iter(self.loader)
        cur, cur_ev = self._upload(next(it, None))
        while cur is not None:
            nxt, nxt_ev = self._upload(next(it, None))
            if cur_ev is not None:
                torch.cuda.current_stream(self.device).wait_event(cur_ev)          # (scenes that were not lifted ahead read it here)
            if nxt is not None:
                self.model.offer_next(nxt, ready=nxt_ev)
            yield cur
            cur, cur_ev = nxt, nxt_ev
