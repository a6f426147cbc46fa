#!/usr/bin/env python3
"""Where a training step's wall time goes outside the kernels: the phases of StudentTrainer.scene_step timed sync-to-sync on the host
(wall) beside the GPU time of the same phase (HIP events), S-shaped synthetic scene.  Diagnostic; usage: time_train_phases.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import ops, pipeline as pl, synthetic as syn, training  # noqa: E402
from geopurify_amd.pipeline import GEO_DIM  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = syn.CONFIGS["S"]
scene = syn.make_scene(cfg, 5557)
rigid = pl.scene_rigid_transform(cfg.voxel_size, 5557)
batch = pl.build_scene_batch(pl.upload_scene(scene, "cuda"), rigid, "cuda")
N = batch.scene_coords.shape[0]
g = torch.Generator(device="cuda").manual_seed(1)
F_lift = torch.nn.functional.normalize(torch.randn(N, 512, device="cuda", generator=g), dim=1)
F_teacher = torch.randn(N, 1088, device="cuda", generator=g)
sd = pl.random_student_state_dict(512 + GEO_DIM, hidden=512, embed=128, num_blocks=4, seed=0)
tr = training.StudentTrainer(sd, "cuda")
xyz = batch.scene_coords.float().contiguous()
acc = {}


class Phase:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        torch.cuda.synchronize()
        self.t = time.perf_counter()

    def __exit__(self, *a):
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        d = acc.setdefault(self.name, [0.0, 0.0])
        d[0] += (t1 - self.t) * 1e3
        d[1] += (t2 - self.t) * 1e3


def step():
    dev = "cuda"
    anchors = torch.randperm(N, device="cuda")[:4096]
    with Phase("1 point kNN + sampler"):
        nbrs, flag = ops.knn_points(xyz, anchors, 96)
        positive, negative = training.sample_contrastive_pairs_hybrid(F_teacher, nbrs, anchors, 63)
        assert not int(flag.item())
    with Phase("2 unique points / voxels"):
        all_idx, point_to_batch = torch.unique(torch.cat([anchors, positive, negative.flatten()]), return_inverse=True)
        vox = batch.scene_inds_reconstruct[all_idx]
        uniq_vox, sample_to_voxel = torch.unique(vox, return_inverse=True)
    with Phase("3 morton order, segments, voxel inputs"):
        cs_ref = batch.scene_coords_3d[uniq_vox].floor().to(torch.int32).contiguous()
        perm, rank = ops.morton_order(cs_ref)
        cs = cs_ref[perm.long()].contiguous()
        s2v = rank.long()[sample_to_voxel].contiguous()
        order = torch.sort(s2v, stable=True).indices
        Nvs = cs.shape[0]
        seg = torch.zeros(Nvs + 1, dtype=torch.int64, device=dev)
        seg[1:] = torch.bincount(s2v, minlength=Nvs).cumsum(0)
        X = torch.zeros((Nvs, tr.cin_pad), dtype=torch.float32, device=dev)
        ops.scatter_mean_csr(F_lift[all_idx].contiguous(), 512, order, seg, Nvs, X, col0=0)
        ops.scatter_mean_csr(batch.scene_gauss_features[all_idx].contiguous(), GEO_DIM, order, seg, Nvs, X, col0=512)
    with Phase("4 grid + kernel map"):
        grid = ops.grid_build(cs)
        nbr_map = ops.kernel_map_build(grid, cs)
    with Phase("5 forward_backward (pairs, wgrad plan, student, loss, gradients)"):
        loss, grads, E = tr.forward_backward(X, nbr_map, s2v, point_to_batch.contiguous(), 4096, 63)
    with Phase("6 AdamW"):
        tr.optimizer_step(grads)


step()
acc.clear()
for _ in range(steps):
    step()
print(f"{'phase':70s} {'host enqueue':>13s} {'until done':>11s}   (ms per step, {steps} steps)")
for k, v in acc.items():
    print(f"{k:70s} {v[0] / steps:13.2f} {v[1] / steps:11.2f}")
print(f"{'sum':70s} {sum(v[0] for v in acc.values()) / steps:13.2f} {sum(v[1] for v in acc.values()) / steps:11.2f}")
