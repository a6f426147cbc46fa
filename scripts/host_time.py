#!/usr/bin/env python3
"""Host-side time per stage of one S scene (launch + blocking syncs) against the wall time: is the single host thread the
bottleneck?  (Measured: 32.7 ms host vs 33.8 ms wall on one stream; with two streams the GPU is the limit.)"""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from geopurify_amd import ops, pipeline as pl, synthetic as syn
cfg = syn.CONFIGS["S"]
dev = torch.device("cuda", 0)
scene = pl.upload_scene(syn.make_scene(cfg, 5557), dev)
vlm = pl.SyntheticVLM(syn.make_vlm_outputs(cfg, cfg.num_views, 5557), dev)
rigid = pl.scene_rigid_transform(cfg.voxel_size, 5557)
sd = pl.random_student_state_dict(cfg.feat_dim + pl.GEO_DIM, hidden=512, embed=128, num_blocks=4, seed=0)
hp = pl.HotPath(pl.StudentWeights(sd, dev), cfg.mask_shape, device=dev)
counts = torch.zeros((3, cfg.num_classes), dtype=torch.int64, device=dev)
def step():
    t = [time.perf_counter()]
    b = pl.build_scene_batch(scene, rigid, dev); t.append(time.perf_counter())
    F, text, scale = hp.lift_masks(b, vlm); t.append(time.perf_counter())
    f = hp.refine(b, F); t.append(time.perf_counter())
    hp.classify_and_count({"scene_features": f, "text_features": text, "logit_scale": scale}, b.scene_label, cfg.num_classes, cfg.ignore_ids, counts); t.append(time.perf_counter())
    return np.diff(t)
for _ in range(2): step()
torch.cuda.synchronize()
acc = []
t0 = time.perf_counter()
for _ in range(5):
    acc.append(step())
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host ms per stage (loader, lift, refine, classify):", (np.mean(acc, 0) * 1e3).round(2), "host total", round((t1 - t0) / 5 * 1e3, 2), "ms; wall incl final drain", round((t2 - t0) / 5 * 1e3, 2))
