#!/usr/bin/env python3
"""Does a second stream's work run BESIDE a long backlog of convolution launches, or after it?  The student of one S scene is enqueued
on stream A; then (host order) a chain of small kernels on stream B, gated on an event recorded on A before the student.  Printed: when
B's chain starts and ends relative to the student's start and end, for several forms of B's work and stream priorities."""
import os
import sys
import dataclasses

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import ops, pipeline as pl, synthetic as syn  # noqa: E402

dev = "cuda"
cfg = dataclasses.replace(syn.CONFIGS["S"], num_views=1)
sc = syn.make_scene(cfg, 5557)
rigid = pl.scene_rigid_transform(cfg.voxel_size, 5557)
vox = ops.voxelize(torch.from_numpy(sc.coords).cuda(), rigid)
coords = vox["coords_aug"].to(torch.int32).contiguous()
perm, rank = ops.morton_order(coords)
cs = coords[perm.long()].contiguous()
grid = ops.grid_build(cs)
nm = ops.kernel_map_build(grid, cs)
pairs = ops.conv_pairs_build(nm)
Nv = cs.shape[0]
st = pl.StudentWeights(pl.random_student_state_dict(518, 512, 128, 4, seed=1), dev)
X = torch.randn(Nv, st.cin_pad, device=dev)
xs = st.split_input(X)
buf = torch.zeros(1 << 20, device=dev)
big = torch.zeros(64 << 20, device=dev)


def small_chain(n=60):
    for _ in range(n):
        buf.add_(1.0)                                       # 4 MB elementwise: a few microseconds alone


def knn_chain(n=4):
    for _ in range(n):
        ops.knn_lattice(grid, cs, perm, 96)                 # 0.43 ms alone, <= 70 VGPRs


def stream_chain(n=8):
    for _ in range(n):
        big.add_(1.0)                                       # 512 MB of HBM traffic each: ~0.1 ms alone


def run(label, chain, prio_b, gate=True, b_first=False):
    A = torch.cuda.Stream(priority=0)
    B = torch.cuda.Stream(priority=prio_b)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    a0, a1, b0, b1 = ev(), ev(), ev(), ev()
    torch.cuda.synchronize()
    with torch.cuda.stream(A):
        a0.record(A)
        if not b_first:
            st.forward(X, nm, pairs, x_split=xs)
            a1.record(A)
    with torch.cuda.stream(B):
        if gate:
            B.wait_event(a0)
        b0.record(B)
        chain()
        b1.record(B)
    if b_first:
        with torch.cuda.stream(A):
            st.forward(X, nm, pairs, x_split=xs)
            a1.record(A)
    torch.cuda.synchronize()
    print(f"{label:58s} student {a0.elapsed_time(a1):6.2f} ms; chain starts {a0.elapsed_time(b0):+7.2f}, ends {a0.elapsed_time(b1):+7.2f} ms after the student's start", flush=True)


for _ in range(2):
    st.forward(X, nm, pairs, x_split=xs); small_chain(); knn_chain(1); stream_chain(1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, ch in (("60 small kernels", small_chain), ("4 x kNN", knn_chain), ("8 x 512 MB add", stream_chain)):
    e0.record(); ch(); e1.record(); torch.cuda.synchronize()
    print(f"alone: {name:20s} {e0.elapsed_time(e1):6.2f} ms")
for name, ch in (("60 small kernels", small_chain), ("4 x kNN", knn_chain), ("8 x 512 MB add", stream_chain)):
    for prio in (0, -1):
        run(f"{name}, enqueued after the student, priority {prio}", ch, prio)
    run(f"{name}, enqueued BEFORE the student (host order)", ch, 0, gate=True, b_first=True)
    run(f"{name}, after, no gate event", ch, 0, gate=False)
