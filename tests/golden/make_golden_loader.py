#!/usr/bin/env python3
"""Golden vectors from the reference's OWN ScannetLoaderFull.__getitem__ (dataset/data_loader_ablation.py:128-394 and the
Matterport variant dataset/data_loader_matterport.py:144-300), build container only.
Run:  python tests/golden/make_golden_loader.py

The dataset modules import imageio, cv2, SharedArray, plyfile, models.scene (3DGS scene reader) and
models.utils.dataset_utils at module level; none of those exists offline.  Placeholder modules are registered for the names,
with the four calls the loader makes answered from in-memory arrays (the inputs stored in the fixture):
  Scene(cfg, ...).getTrainCameras(scale)  -> the synthetic views (pose, intrinsics, image)      [file readers: out of scope]
  load_point_ply(path, islabel=True)       -> (xyz, None, labels, normals) of the synthetic scene
  imageio.imread(path)                     -> the depth image in millimetres / depth units, or the 2D label image
  cv2.resize(img, img_dim[, NEAREST])      -> identity (the synthetic images already have img_dim)
Everything else -- the mapper, the view-drop rule, label handling, both voxelizations with the reference's np.random
stream, tensor conversions -- is the reference's code.  Only inputs and outputs are stored."""
import collections
import collections.abc
import importlib.machinery
import os
import sys
import tempfile
import types

import numpy as np
import torch

collections.Sequence = collections.abc.Sequence
collections.Iterable = collections.abc.Iterable
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)

IMAGES = {}                                                   # path -> array served by the imread placeholders


def _placeholder(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _imread(path):
    return IMAGES[str(path)]


def _resize(img, dsize, interpolation=None):
    assert (img.shape[1], img.shape[0]) == tuple(dsize), "the synthetic images already have img_dim"
    return img


class AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


class _Views(list):
    camera_info = None


class _Scene:
    registry = {}                                             # scene name -> _Views

    def __init__(self, cfg, *a, **k):
        self.name = os.path.basename(cfg.scene_path.rstrip("/"))
        if a and isinstance(a[0], str):                       # matterport: Scene(cfg, scene_data_path, ...)
            self.name = os.path.basename(a[0]).split(".pth")[0]

    def getTrainCameras(self, scale=1.0):
        return _Scene.registry[self.name]


def _load_point_ply(path, islabel=True):
    return PLY[os.path.basename(os.path.dirname(path))]


PLY = {}
for n in ("SharedArray", "plyfile", "open3d"):
    _placeholder(n)
_placeholder("imageio")
_placeholder("imageio.v2", imread=_imread)
_placeholder("imageio.v3", imread=_imread)
_placeholder("cv2", resize=_resize, INTER_NEAREST=0)
_placeholder("models.scene", Scene=_Scene)
_placeholder("models.utils.dataset_utils", load_point_ply=_load_point_ply)

from geopurify_amd import synthetic as syn  # noqa: E402  (input generation only)

# torch >= 2.6 defaults torch.load to weights_only=True, which refuses the numpy arrays of the reference's .pth scene files;
# the reference was written for the older default
_torch_load = torch.load
torch.load = lambda *a, **k: _torch_load(*a, **{**k, "weights_only": False})


def case_config(dataset, n_points, num_views):
    """the synthetic scene shape of a fixture case (also used by the tests to regenerate the train-rule scene)"""
    import dataclasses
    base = syn.CONFIGS["T"]
    if dataset == "scannet":
        return dataclasses.replace(base, num_points=n_points, num_views=num_views, dataset=dataset, depth_scale=1000.0)
    # Matterport intrinsics are given at image_dim and do not scale with it: a 640 x 512 image keeps a useful field of view
    return dataclasses.replace(base, num_points=n_points, num_views=num_views, dataset=dataset, depth_scale=4000.0,
                               image_dim=(640, 512), mask_shape=(512, 640), cut_bound=2)


def make_case(dataset, split, seed, n_points, num_views, val_keep, tmp, store_images=True):
    cfg = case_config(dataset, n_points, num_views)
    scene = syn.make_scene(cfg, seed)
    rng = np.random.default_rng(seed)
    W, H = cfg.image_dim
    name = "scene0000_00" if dataset == "scannet" else "region7"
    # the .pth the reference torch.load()s: colours in [-1, 1], labels with -100 / 255 sprinkled in
    feats_in = scene.colors * 2.0 - 1.0
    labels_in = scene.labels.astype(np.float64)
    labels_in[rng.random(n_points) < 0.02] = -100
    labels_in[rng.random(n_points) < 0.02] = 255
    root3d = os.path.join(tmp, "scannet_3d" if dataset == "scannet" else "matterport_3d")
    os.makedirs(os.path.join(root3d, split), exist_ok=True)
    if dataset == "scannet":
        pth = os.path.join(root3d, split, name + "_vh_clean_2.pth")
        torch.save((scene.coords, feats_in, labels_in.copy()), pth)
        PLY[name] = (scene.coords, None, scene.labels, scene.normals)
    else:
        pth = os.path.join(root3d, split, name + ".pth")
        torch.save((scene.coords, feats_in, scene.normals, labels_in.copy()), pth)
    views = _Views()
    infos = []
    inputs = {"locs_in": scene.coords, "feats_in": feats_in, "normals": scene.normals, "labels_in": labels_in}
    label_ids = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39]       # ScanNet-20 raw ids
    for i, v in enumerate(scene.views):
        img_path = os.path.join(tmp, "2d", name, "color", f"{i}.jpg") if dataset == "scannet" else \
            os.path.join(tmp, "2d", name, "color", f"uuid{i:04d}_i1_{i % 6}.jpg")
        image = torch.from_numpy(np.random.default_rng(7000 + i).random((3, H, W)).astype(np.float32))
        depth_units = np.round(v.depth * cfg.depth_scale).astype(np.uint16)
        if dataset == "scannet":
            IMAGES[img_path.replace("color", "depth").replace("jpg", "png")] = depth_units
            lab = rng.choice(np.array(label_ids + [0, 13, 40]), size=(H, W)).astype(np.uint8)
            IMAGES[os.path.join(tmp, "2d", name, "label", f"{i}.png")] = lab
            inputs[f"v{i}_label_img"] = lab
            wvt = torch.from_numpy(v.pose)                              # world_view_transform = W2C^T
            intr = v.K
        else:
            d = img_path.replace("color", "depth")
            _, img_type, yaw = img_path.split("/")[-1].split("_")
            IMAGES[d[:-8] + "d" + img_type[1] + "_" + yaw[0] + ".png"] = depth_units
            wvt = torch.from_numpy(np.ascontiguousarray(v.pose.T))      # the loader passes world_view_transform^T as camera-to-world
            intr = v.K
        views.append(types.SimpleNamespace(image_path=img_path, R=np.eye(3), T=np.zeros(3), original_image=image,
                                           world_view_transform=wvt))
        infos.append(types.SimpleNamespace(intrinsics=intr))
        inputs[f"v{i}_world_view_transform"] = wvt.numpy().copy()
        inputs[f"v{i}_intrinsics"] = np.asarray(intr, dtype=np.float64)
        inputs[f"v{i}_depth_units"] = depth_units                  # the loader divides by 1000 (ScanNet) / fusion.depth_scale
        inputs[f"v{i}_image_seed"] = np.int64(7000 + i)          # image = default_rng(seed).random((3, H, W)) as float32
        if store_images:
            inputs[f"v{i}_image_u8"] = (image.permute(1, 2, 0).numpy() * 255).astype(np.uint8)
    views.camera_info = infos
    _Scene.registry[name] = views
    split_cfg = AttrDict(base_category=[0, 2, 3, 5, 7, 8, 9, 11, 12, 13, 14, 15, 17, 18, 19], novel_category=[1, 4, 6, 10, 16],
                         ignore_category=[20, 21])
    scene_cfg = types.SimpleNamespace(scene=types.SimpleNamespace(scene_path=os.path.join(tmp, "gs")),
                                      fusion=types.SimpleNamespace(img_dim=cfg.image_dim, visibility_threshold=cfg.vis_thres,
                                                                   cut_boundary=cfg.cut_bound, depth_scale=cfg.depth_scale))
    if dataset == "scannet":
        from dataset.data_loader_ablation import ScannetLoaderFull
    else:
        from dataset.data_loader_matterport import ScannetLoaderFull
    ds = ScannetLoaderFull(datapath_prefix=root3d, datapath_prefix_2d=os.path.join(tmp, "2d"), label_2d=label_ids,
                           category_split=split_cfg, val_keep=val_keep, voxel_size=cfg.voxel_size, split=split, aug=False,
                           scene_config=scene_cfg)
    assert len(ds.samples) == num_views
    out = dict(inputs)
    out.update(dataset=dataset, split=split, depth_scale=np.float64(cfg.depth_scale), num_views=np.int64(num_views), val_keep=np.int64(val_keep), voxel_size=cfg.voxel_size,
               img_dim=np.array(cfg.image_dim), vis_thres=cfg.vis_thres, cut_bound=np.int64(cfg.cut_bound), label_2d_ids=np.array(label_ids),
               base_category=np.array(split_cfg.base_category), novel_category=np.array(split_cfg.novel_category),
               ignore_category=np.array(split_cfg.ignore_category))
    kept = []
    for i in range(num_views):
        np.random.seed(1000 + i)
        out[f"v{i}_np_seed"] = np.int64(1000 + i)
        r = ds[i]
        kept.append(r is not None)
        if r is not None:
            for j, x in enumerate(r):
                if x is not None and (store_images or j not in (10, 11)):
                    out[f"v{i}_out_{j}"] = x.numpy() if torch.is_tensor(x) else np.asarray(x)
    out["kept"] = np.array(kept)
    return out, kept


def main():
    with tempfile.TemporaryDirectory() as tmp:
        # ScanNet, val: one view below 400 visible points (camera looking at a corner), one above val_keep
        out, kept = make_case("scannet", "val", 31, 5000, 5, 470, tmp)
        print("scannet val kept:", kept, "visible:", [int(out[f"v{i}_out_14"].sum()) if k else None for i, k in enumerate(kept)])
        assert any(kept) and not all(kept)
        np.savez_compressed(os.path.join(HERE, "ref_loader_scannet.npz"), **out)
    for k in list(sys.modules):
        if k.startswith("dataset.data_loader"):
            del sys.modules[k]
    with tempfile.TemporaryDirectory() as tmp:
        # Matterport, val: view 1 below 400 visible points, view 3 above val_keep (images are regenerated from their seeds)
        out, kept = make_case("matterport", "val", 47, 8000, 5, 1500, tmp, store_images=False)
        print("matterport val kept:", kept, "visible:", [int(out[f"v{i}_out_14"].sum()) if k else None for i, k in enumerate(kept)])
        assert any(kept) and not all(kept)
        np.savez_compressed(os.path.join(HERE, "ref_loader_matterport.npz"), **out)
    with tempfile.TemporaryDirectory() as tmp:
        # train rule (400 .. 65000 visible points, :279-281): a 700k-point scene whose views see up to ~90k points; only the
        # keep / drop decisions are stored (the tests regenerate the scene from its seed)
        _Scene.registry.clear()
        from dataset.data_loader_ablation import ScannetLoaderFull  # noqa: F401
        out, kept = make_case("scannet", "train", 53, 700_000, 3, 10_000_000, tmp, store_images=False)
        assert any(kept) and not all(kept)
        small = {k: v for k, v in out.items() if not k.startswith("v") or k.endswith(("_world_view_transform", "_intrinsics", "_np_seed"))}
        small["kept"] = np.array(kept)
        print("scannet train kept:", kept)
        small["seed"], small["n_points"] = np.int64(53), np.int64(700_000)
        np.savez_compressed(os.path.join(HERE, "ref_loader_train_rule.npz"), **{k: v for k, v in small.items() if k not in ("locs_in", "feats_in", "normals", "labels_in")})
    print("ok")


if __name__ == "__main__":
    main()
