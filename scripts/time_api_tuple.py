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

# ---- the look-ahead form (SonataXAffinityTrainer.offer_next): host tuple / device tuple offered one scene ahead
tup_b = tuple(x.clone().pin_memory() if torch.is_tensor(x) else x for x in tup)
dev_a = tuple(x.to(dev) if torch.is_tensor(x) else x for x in tup)
dev_b = tuple(x.to(dev) if torch.is_tensor(x) else x for x in tup)


def chain(seq, n=12, offer=True):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            if offer and i + 1 < n:
                model.offer_next(seq[(i + 1) % 2])
            model.evaluate_scene(seq[i % 2])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
    return ms


print(f"12 scenes, pinned host tuples, serial          {chain([tup, tup_b], offer=False):7.2f} ms per scene")
print(f"12 scenes, pinned host tuples, offer_next      {chain([tup, tup_b]):7.2f} ms per scene")
print(f"12 scenes, device tuples, serial               {chain([dev_a, dev_b], offer=False):7.2f} ms per scene")
print(f"12 scenes, device tuples, offer_next           {chain([dev_a, dev_b]):7.2f} ms per scene")
import geopurify_amd.ops as _ops
hp = model._hot_path()
orig_prepare = hp.prepare
acc = {"look": 0.0, "n": 0}
orig_look = model._look_ahead


def timed_look(*a, **k):
    t0 = time.perf_counter()
    r = orig_look(*a, **k)
    acc["look"] += time.perf_counter() - t0
    acc["n"] += 1
    return r


model._look_ahead = timed_look
_ops.READBACK["seconds"] = 0.0
ms = chain([dev_a, dev_b])
print(f"   host inside _look_ahead per scene: {acc['look'] / max(acc['n'], 1) * 1e3:.2f} ms ({acc['n']} calls), of which blocked in ops read-backs "
      f"{_ops.READBACK['seconds'] / max(acc['n'], 1) * 1e3:.2f} ms")

# ---- the loader wrapper (geopurify_amd.data_loader.LookAheadLoader) on pinned host tuples and on device tuples; GPU timeline of one scene
from geopurify_amd.data_loader import LookAheadLoader


def wrapped(seq, n=12):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for d in LookAheadLoader([seq[i % 2] for i in range(n)], model):
            model.evaluate_scene(d)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
    return ms


print(f"12 scenes, device tuples, LookAheadLoader      {wrapped([dev_a, dev_b]):7.2f} ms per scene")
print(f"12 scenes, pinned host tuples, LookAheadLoader {wrapped([tup, tup_b]):7.2f} ms per scene")
model._look_ahead = orig_look
marks = []
orig_refine = hp.refine


def refine_marked(batch, F, after_student=None, prepared=None):
    e0 = torch.cuda.Event(enable_timing=True); e0.record()

    def hook():
        e1 = torch.cuda.Event(enable_timing=True); e1.record()         # student end (main stream)
        if after_student is not None:
            after_student()
        e2 = torch.cuda.Event(enable_timing=True); e2.record()         # after the wait for the look-ahead
        marks.append((e0, e1, e2))
    r = orig_refine(batch, F, after_student=hook, prepared=prepared)
    e3 = torch.cuda.Event(enable_timing=True); e3.record()
    marks[-1] = marks[-1] + (e3,)
    return r


hp.refine = refine_marked
wrapped([dev_a, dev_b], 6)
torch.cuda.synchronize()
for e0, e1, e2, e3 in marks[-4:]:
    print(f"   scene on the GPU: student {e0.elapsed_time(e1):6.2f} ms, wait for the look-ahead {e1.elapsed_time(e2):6.2f} ms, affinity + pooling + gather {e2.elapsed_time(e3):6.2f} ms")
