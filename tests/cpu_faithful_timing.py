#!/usr/bin/env python3
"""CPU oracle, both variants of the consensus fusion: "faithful" keeps the reference's two per-point
Python loops (affinity_module.py:633-638, 664-670), "vectorised" replaces them by tensor ops (the stronger baseline that
bench.py reports).  Timed on a 50k-point mask-lift scene (the size of the plumbing config); CPU only.
Lives under tests/ because it runs the oracle, which only tests, smoke() and bench.py's CPU-baseline leg may do."""
import dataclasses
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import pipeline as pl, synthetic as syn  # noqa: E402
from oracle import pipeline as o_pipe  # noqa: E402

torch.set_num_threads(min(os.cpu_count() or 1, 16))
cfg = dataclasses.replace(syn.CONFIGS["T"], num_points=50000, num_views=4)
scene = syn.make_scene(cfg, 11)
vlm = syn.make_vlm_outputs(cfg, cfg.num_views, 11)
sd = pl.random_student_state_dict(cfg.feat_dim + pl.GEO_DIM, hidden=32, embed=32, num_blocks=1, seed=0)
rigid = pl.scene_rigid_transform(cfg.voxel_size, 11)
res = {}
for name, vec in (("vectorised", True), ("faithful", False)):
    tm = {}
    t0 = time.perf_counter()
    out = o_pipe.evaluate_scene_oracle(scene, vlm, sd, rigid, K=16, num_iters=1, vectorised=vec, timings=tm, knn_impl="kdtree")
    res[name] = (time.perf_counter() - t0, tm, out["lifted"])
    print(f"{name:11s} whole scene {res[name][0]:7.2f} s   fuse+fill {tm.get('fuse+fill', float('nan')):7.2f} s   lift per view {tm.get('lift per view', float('nan')):6.2f} s",
          flush=True)
d = (res["vectorised"][2] - res["faithful"][2]).abs().max().item()
print(f"max |lifted(vectorised) - lifted(faithful)| = {d:.2e}   threads {torch.get_num_threads()}")
