#!/usr/bin/env python3
"""Host -> device time of one S scene's raw inputs (what the reference's DataLoader hands over as CPU tensors: points,
colours + normals, labels, per-view depth maps and images), pageable and pinned -- the PCIe share that bench.py's `value`
excludes by contract (inputs resident in HBM).  The 2D VLM's outputs are produced on the device in the reference and are
not part of it."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import synthetic as syn  # noqa: E402

cfg = syn.CONFIGS["S"]
sc = syn.make_scene(cfg, 5557)
W, H = cfg.image_dim
host = [sc.coords, np.concatenate([sc.colors, sc.normals], 1).astype(np.float32), sc.labels, np.stack([v.depth for v in sc.views]),
        np.zeros((cfg.num_views, H, W, 3), np.float32)]
nbytes = sum(a.nbytes for a in host)
torch.cuda.init()
torch.zeros(1, device="cuda")
for name, pin in (("pageable", False), ("pinned", True)):
    ts = [torch.from_numpy(a) for a in host]
    if pin:
        ts = [t.pin_memory() for t in ts]
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dev = [t.to("cuda", non_blocking=pin) for t in ts]
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"{name:9s} {nbytes / 1e6:.1f} MB in {best * 1e3:.2f} ms = {nbytes / best / 1e9:.1f} GB/s")
