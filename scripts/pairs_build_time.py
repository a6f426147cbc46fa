#!/usr/bin/env python3
"""Wall time of ops.conv_pairs_build (kernels + its host synchronisations) on the S scene: equal chunk heights against the balanced plan."""
import os, sys, time, dataclasses
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import _lib, ops, pipeline as pl, synthetic as syn  # noqa: E402

cfg = dataclasses.replace(syn.CONFIGS["S"], num_views=1)
sc = syn.make_scene(cfg, 5557)
rigid = pl.scene_rigid_transform(cfg.voxel_size, 5557)
vox = ops.voxelize(torch.from_numpy(sc.coords).cuda(), rigid)
coords = vox["coords_aug"].to(torch.int32).contiguous()
perm, rank = ops.morton_order(coords)
cs = coords[perm.long()].contiguous()
nm = ops.kernel_map_build(ops.grid_build(cs), cs)
lib = _lib.load()
for label, arg in (("8192 rows", 8192), ("balanced", "balanced"), ("8192 rows", 8192), ("balanced", "balanced")):
    for _ in range(3):
        ops.conv_pairs_build(nm, arg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        p = ops.conv_pairs_build(nm, arg)
    torch.cuda.synchronize()
    print(f"{label:10s}: {1e3 * (time.perf_counter() - t0) / 20:6.3f} ms per call ({p.num_chunks} chunks)", flush=True)
# the plan alone: kernels, then the read-back
nv = nm.shape[1]
mc = (nv + 255) // 256
plan = torch.empty(mc + 2, dtype=torch.int32, device="cuda")
ws = torch.empty(lib.gp_conv_chunk_plan_workspace_bytes(nv, 256), dtype=torch.uint8, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        _lib.check(lib.gp_conv_chunk_plan(nm.data_ptr(), nv, 27, 256, 2, 512, mc, plan.data_ptr(), plan[mc + 1:].data_ptr(), ws.data_ptr(), ws.numel(), None), "plan")
    e1.record()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        h = plan.cpu().tolist()
    t1 = time.perf_counter()
    print(f"plan kernels {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us per call; read-back of {plan.numel()} ints {1e6 * (t1 - t0) / 20:7.1f} us", flush=True)
