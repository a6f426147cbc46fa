#!/usr/bin/env python3
"""Scene-level nearest-seen fill: grid resolutions vs brute force (tuning aid)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import _lib, ops, pipeline as pl, synthetic as syn
cfgname = sys.argv[1] if len(sys.argv) > 1 else "S"
cfg = syn.CONFIGS[cfgname]
sc = syn.make_scene(cfg, 5557)
rigid = pl.scene_rigid_transform(cfg.voxel_size, 5557)
b = pl.build_scene_batch(pl.upload_scene(sc, "cuda"), rigid, "cuda")
N = b.scene_coords.shape[0]
seen = torch.zeros(N, dtype=torch.uint8, device="cuda")
for v in b.views:
    seen[v.pt] = 1
print("N", N, "seen", int(seen.sum()), flush=True)
lib = _lib.load()
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ref = None
for ng in (0, 48, 64, 96, 128):
    if ng == 0:
        lib.gp_debug_set(5, 1)
    else:
        lib.gp_debug_set(5, 0); lib.gp_debug_set(6, ng)
    ms = t(lambda: ops.nn1_masked(b.scene_coords, seen, 1 - seen))
    out = ops.nn1_masked(b.scene_coords, seen, 1 - seen)
    if ref is None: ref = out
    print(f"{'brute force' if ng == 0 else 'grid ' + str(ng):14s} {ms:8.3f} ms  equal={bool(torch.equal(out, ref))}", flush=True)
