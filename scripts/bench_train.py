#!/usr/bin/env python3
"""Training step of the student at the reference shape on one synthetic S-shaped scene (BASELINE config 5):
N = 150k points, teacher features [N, 1088], 4096 anchors x (1 positive + 63 negatives), student 518 -> 512 x 9 -> 128.
Prints per-stage HIP-event times and steps/s.  Usage: bench_train.py [steps] [teacher_dim]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import ops, pipeline as pl, synthetic as syn, training  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
Dt = int(sys.argv[2]) if len(sys.argv) > 2 else 1088
cfg = syn.CONFIGS["S"]
scene = syn.make_scene(cfg, 5557)
rigid = pl.scene_rigid_transform(cfg.voxel_size, 5557)
batch = pl.build_scene_batch(pl.upload_scene(scene, "cuda"), rigid, "cuda")
N = batch.scene_coords.shape[0]
g = torch.Generator(device="cuda").manual_seed(1)
F_lift = torch.nn.functional.normalize(torch.randn(N, 512, device="cuda", generator=g), dim=1)
F_teacher = torch.randn(N, Dt, device="cuda", generator=g)
sd = pl.random_student_state_dict(512 + pl.GEO_DIM, hidden=512, embed=128, num_blocks=4, seed=0)
tr = training.StudentTrainer(sd, "cuda", base_lr=1e-4, weight_decay=1e-5, warmup_iters=10, main_iters=1000)
xyz = batch.scene_coords.float().contiguous()


class Ev:
    def __init__(self):
        self.marks = []

    def mark(self, name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.marks.append((name, e))

    def table(self):
        torch.cuda.synchronize()
        return {b[0]: a[1].elapsed_time(b[1]) for a, b in zip(self.marks[:-1], self.marks[1:])}


def one_step(ev=None):
    anchors = torch.randperm(N, device="cuda")[:4096]
    if ev:
        ev.mark("start")
    out = tr.scene_step(F_lift, batch.scene_gauss_features, batch.scene_inds_reconstruct, batch.scene_coords_3d, xyz, F_teacher,
                        anchors, num_negatives=63, K=96, optimize=False)
    if ev:
        ev.mark("sample + student forward/backward")
    tr.optimizer_step(out["grads"])
    if ev:
        ev.mark("AdamW (64 M parameters)")
    return out


out = one_step()
torch.cuda.synchronize()
print(f"N={N} sampled points={out['num_samples']} sampled voxels={out['num_voxels']} loss={float(out['loss']):.4f}", flush=True)
for _ in range(2):                                   # the anchors -- and with them the voxel count and every buffer size -- change per step:
    one_step()                                       # two more steps let the caching allocator see the sizes before the clock starts
torch.cuda.synchronize()
t0 = time.perf_counter()
ev = Ev()
ends = []
for _ in range(steps):
    out = one_step(ev if _ == steps - 1 else None)
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    ends.append(e)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"{dt * 1e3:.1f} ms/step = {1 / dt:.2f} steps/s; last loss {float(out['loss']):.4f}")
print("   per step (GPU clock between the steps' ends): " + " ".join(f"{a.elapsed_time(b):.1f}" for a, b in zip(ends[:-1], ends[1:])))
for k, v in ev.table().items():
    print(f"   {k:40s} {v:8.2f} ms")
