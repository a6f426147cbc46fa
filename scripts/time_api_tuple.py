#!/usr/bin/env python3
"""Where does the drop-in call evaluate_scene(20-tuple of CPU tensors) spend its time?  (tuning aid)"""
import os, sys, time, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import pipeline as pl, synthetic as syn
from geopurify_amd.affinity_module import SonataXAffinityTrainer

cfg = syn.CONFIGS["S"]
dev = "cuda"
scene = syn.make_scene(cfg, 5557)
vlm = pl.SyntheticVLM(syn.make_vlm_outputs(cfg, cfg.num_views, 5557), dev)
rigid = pl.scene_rigid_transform(cfg.voxel_size, 5557)
b = pl.build_scene_batch(pl.upload_scene(scene, dev), rigid, dev, batch_views=False)
raw = list(b.as_tuple())
H, W = cfg.mask_shape
raw[11] = torch.stack([torch.full((H, W, 3), float(i)) for i in range(len(b.views))])
tup = tuple(x.cpu().pin_memory() if torch.is_tensor(x) else x for x in raw)
ns = types.SimpleNamespace(all_label=[f"c{i}" for i in range(cfg.num_classes)], mask_shape=list(cfg.mask_shape), voxel_size=cfg.voxel_size)
model = SonataXAffinityTrainer(ns, None, None, device="cuda", use_lseg=False, vlm=vlm, feature_dim=cfg.feat_dim).to(dev).eval()


def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r


ms, _ = t(lambda: model.evaluate_scene(tup)); print(f"evaluate_scene(tuple)            {ms:7.2f} ms")
ms, hp = t(lambda: model._hot_path()); print(f"  _hot_path()                    {ms:7.2f} ms")
ms, batch = t(lambda: model._batch_from_tuple(tup, hp.device)); print(f"  _batch_from_tuple              {ms:7.2f} ms")
ms, lf = t(lambda: hp.lift_masks(batch, vlm)); print(f"  lift_masks                     {ms:7.2f} ms")
ms, _ = t(lambda: hp.refine(batch, lf[0])); print(f"  refine                         {ms:7.2f} ms")
ms, _ = t(lambda: hp.evaluate_scene(batch, vlm)); print(f"  hp.evaluate_scene(batch)       {ms:7.2f} ms")
bd = pl.build_scene_batch(pl.upload_scene(scene, dev), rigid, dev)
ms, _ = t(lambda: hp.evaluate_scene(bd, vlm)); print(f"  hp.evaluate_scene(device-built batch, one stream) {ms:7.2f} ms")
