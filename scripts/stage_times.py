#!/usr/bin/env python3
"""Per-op wall times (synchronised) of one scene through the device hot path -- a debugging aid."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import ops, pipeline as pl, synthetic as syn  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="S")
ap.add_argument("--views", type=int, default=None)
ap.add_argument("--iters", type=int, default=19)
ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()


def log(*s):
    print(*s, file=sys.stderr, flush=True)


# wrap every op with a synchronised timer
acc = {}
for name in dir(ops):
    fn = getattr(ops, name)
    if callable(fn) and not name.startswith("_") and fn.__module__ == ops.__name__ and name not in ("Grid",):
        def mk(fn, name):
            def w(*x, **k):
                torch.cuda.synchronize()
                t = time.perf_counter()
                r = fn(*x, **k)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t
                acc.setdefault(name, []).append(dt)
                if dt > 0.5:
                    log(f"   slow op {name}: {dt:.2f}s")
                return r
            return w
        setattr(ops, name, mk(fn, name))

import dataclasses
cfg = syn.CONFIGS[a.config]
if a.views:
    cfg = dataclasses.replace(cfg, num_views=a.views)
t = time.perf_counter()
sc = syn.make_scene(cfg, 5557)
vlm = pl.SyntheticVLM(syn.make_vlm_outputs(cfg, cfg.num_views, 5557), "cuda")
log(f"inputs generated in {time.perf_counter() - t:.1f}s")
sd = pl.random_student_state_dict(cfg.feat_dim + 6, seed=0)
st = pl.StudentWeights(sd, "cuda")
hp = pl.HotPath(st, cfg.mask_shape, num_iters=a.iters, device="cuda")
pl.upload_scene(sc, "cuda")
rigid = pl.scene_rigid_transform(cfg.voxel_size, 5557)
for rep in range(a.reps):
    acc.clear()
    t0 = time.perf_counter()
    b = pl.build_scene_batch(sc, rigid, "cuda")
    torch.cuda.synchronize(); t1 = time.perf_counter()
    log(f"rep {rep}: loader {t1 - t0:.3f}s views kept {len(b.views)} n_v {[len(v.pt) for v in b.views][:6]}")
    F, text, scale = hp.lift_masks(b, vlm)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    log(f"rep {rep}: lift {t2 - t1:.3f}s")
    out = hp.refine(b, F)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    log(f"rep {rep}: refine {t3 - t2:.3f}s  Nv={hp.stats['Nv']}")
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        log(f"   {k:24s} n={len(v):4d} total={sum(v)*1e3:9.2f} ms  mean={np.mean(v)*1e3:8.3f} ms")
