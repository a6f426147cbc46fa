"""Worker of tests/test_gpu_training.py::test_two_rank_sync_batchnorm_training_step: two ranks (gloo, one GPU) train on
different synthetic scenes with SyncBatchNorm; rank 0 also runs the single-process reference on the concatenated scene
(block-diagonal kernel map, plain BatchNorm over all rows) and compares loss, gradients and running statistics."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import ops, pipeline as pl, sharding, training  # noqa: E402


def surface_voxels(rng, n):
    pts = []
    while len(pts) < n:
        a, b = rng.integers(0, 60, 2)
        pts.append((a, b, 5) if rng.random() < 0.5 else (a, 7, b))
    return np.unique(np.array(pts, dtype=np.int64), axis=0)[:n]


def scene(seed, n, cin, A, Nn):
    rng = np.random.default_rng(seed)
    c = surface_voxels(rng, n)
    ct = torch.from_numpy(c).to(torch.int32).cuda()
    perm, rank = ops.morton_order(ct)
    cs = ct[perm.long()].contiguous()
    nm = ops.kernel_map_build(ops.grid_build(cs), cs)
    g = torch.Generator(device="cuda").manual_seed(seed)
    X = torch.zeros((len(c), pl._pad_to(cin, pl.CONV_PAD)), device="cuda")
    X[:, :cin] = torch.randn(len(c), cin, device="cuda", generator=g)
    S = 3 * A
    s2v = torch.randint(0, len(c), (S,), device="cuda", generator=g)
    anchors = torch.randint(0, S, (A,), device="cuda", generator=g)
    pos = torch.randint(0, S, (A,), device="cuda", generator=g)
    neg = torch.randint(0, S, (A * Nn,), device="cuda", generator=g)
    return X, nm, s2v, anchors, pos, neg


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    cin, hidden, A, Nn = 38, 256, 64, 15
    sd = pl.random_student_state_dict(cin, hidden=hidden, embed=128, num_blocks=1, seed=5)
    sizes = (1800, 1400)
    X, nm, s2v, an, po, ne = scene(100 + rank, sizes[rank], cin, A, Nn)
    tr = training.StudentTrainer(sd, "cuda", sync_bn=True)
    loss, grads, E = tr.forward_backward(X, nm, s2v, torch.cat([an, po, ne]), A, Nn)
    sharding.allreduce_mean_gradients(grads)
    # the same step with the all-reduces launched inside the backward pass (sharding.GradientBuckets): the same averaged gradients
    tr2 = training.StudentTrainer(sd, "cuda", sync_bn=True)
    sink = sharding.GradientBuckets(tr2.gradient_order(), "cuda", bucket_bytes=1 << 20)
    loss2, local2, _ = tr2.forward_backward(X, nm, s2v, torch.cat([an, po, ne]), A, Nn, grad_sink=sink)
    grads2 = sink.finish()
    torch.cuda.synchronize()
    # (two runs of the step differ in the last bits: InfoNCE's scatter adds with fp32 atomics)
    same = len(sink.buckets) >= 2 and abs(float(loss2) - float(loss)) <= 1e-6 * abs(float(loss)) and set(grads2) == set(grads) and \
        all(float((grads2[k] - grads[k]).abs().max()) <= 1e-5 * float(grads[k].abs().max()) + 1e-12 for k in grads)
    print(f"rank {rank}: overlapped buckets ({len(sink.buckets)}) == one all-reduce after backward: {same}")
    st = torch.tensor([1.0 if same else 0.0], dtype=torch.float64)
    dist.all_reduce(st, op=dist.ReduceOp.MIN)
    same = bool(st.item())
    lt = torch.tensor([float(loss)], dtype=torch.float64)
    dist.all_reduce(lt)
    torch.cuda.synchronize()
    ok = bool(same)
    if rank == 0:
        parts = [scene(100 + r, sizes[r], cin, A, Nn) for r in range(world)]
        nv0, S0 = parts[0][0].shape[0], parts[0][2].shape[0]
        Xc = torch.cat([p[0] for p in parts])
        nm1 = torch.where(parts[1][1] >= 0, parts[1][1] + nv0, parts[1][1])
        nmc = torch.cat([parts[0][1], nm1], dim=1).contiguous()
        s2vc = torch.cat([parts[0][2], parts[1][2] + nv0])
        p2b = torch.cat([parts[0][3], parts[1][3] + S0, parts[0][4], parts[1][4] + S0, parts[0][5], parts[1][5] + S0])
        ref = training.StudentTrainer(sd, "cuda", sync_bn=False)
        loss_r, grads_r, _ = ref.forward_backward(Xc, nmc, s2vc, p2b, 2 * A, Nn)
        torch.cuda.synchronize()
        dl = abs(float(lt) / world - float(loss_r)) / max(abs(float(loss_r)), 1e-9)
        print(f"loss mean-of-ranks {float(lt) / world:.6f} single-process {float(loss_r):.6f} rel {dl:.2e}")
        ok &= dl < 2e-5
        for k in sorted(grads_r):
            a, b = grads[k].double(), grads_r[k].double()
            err = float((a - b).abs().max() / (b.abs().max() + 1e-30))
            print(f"  grad {k:40s} max rel err {err:.2e}")
            ok &= err < 5e-4                      # fp32 sums in different orders (row blocks differ), as in test_gpu_training
        for k in sorted(ref.buffers):
            err = float((tr.buffers[k] - ref.buffers[k]).abs().max())
            print(f"  running {k:40s} max abs err {err:.2e}")
            ok &= err < 1e-5
    flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64)
    dist.broadcast(flag, 0)
    dist.destroy_process_group()
    sys.exit(0 if flag.item() == 1.0 else 1)


if __name__ == "__main__":
    main()
