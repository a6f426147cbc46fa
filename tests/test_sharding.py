"""Scene sharding + the single metric all-reduce: split rule, LPT balance, and a world_size-2 gloo
run (CPU) whose reduced counts equal the single-process sum."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from geopurify_amd import sharding


def test_contiguous_split_rule_matches_reference(golden_dir):
    ids = [f"scene{i:04d}_00" for i in range(312)]
    for t in (1, 2, 4, 8, 5):
        parts = [sharding.get_batch_scenes(ids, i, t) for i in range(t)]
        assert sum(parts, []) == ids
        sizes = [len(p) for p in parts]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    # known answers of the rule start = i*(n//t) + min(i, n%t)
    assert sharding.get_batch_scenes(list(range(10)), 0, 4) == [0, 1, 2]
    assert sharding.get_batch_scenes(list(range(10)), 1, 4) == [3, 4, 5]
    assert sharding.get_batch_scenes(list(range(10)), 2, 4) == [6, 7]
    assert sharding.get_batch_scenes(list(range(10)), 3, 4) == [8, 9]
    assert sharding.get_batch_scenes(list(range(405)), 7, 8)[0] == 355


def test_lpt_balances_scannet_val_sizes(golden_dir):
    sizes = np.loadtxt(os.path.join(golden_dir, "scannet_val_point_counts.txt"))
    parts = sharding.assign_scenes_lpt(list(sizes), 8)
    assert sorted(sum(parts, [])) == list(range(312))
    loads = np.array([sizes[p].sum() for p in parts])
    assert loads.max() / loads.mean() < 1.01                  # 8-way imbalance below 1 %
    contiguous = np.array([sizes[sharding.get_batch_scenes(list(range(312)), i, 8)].sum() for i in range(8)])
    assert loads.max() <= contiguous.max()


def _scene_counts(i, counts, C=19):
    rng = np.random.default_rng(1000 + i)
    n = 2000 + 37 * i
    tgt = rng.integers(0, 21, n)
    pred = np.where(rng.random(n) < 0.6, np.minimum(tgt, C - 1), rng.integers(0, C, n))
    from oracle.metric import intersection_and_union        # test infrastructure: stands in for the GPU scene
    it, un, tg = intersection_and_union(pred, tgt, C, [19, 20])
    counts += torch.from_numpy(np.stack([it, un + it - tg, tg]))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    costs = [2000 + 37 * i for i in range(11)]
    c, mine = sharding.evaluate_sharded(11, _scene_counts, 19, "cpu", rank, world, costs=costs, policy="lpt")
    q.put((rank, c.numpy().copy(), mine))
    dist.destroy_process_group()


def test_two_rank_gloo_allreduce_equals_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    single, mine = sharding.evaluate_sharded(11, _scene_counts, 19, "cpu")
    assert mine == list(range(11))
    by_rank = {r: (c, m) for r, c, m in res}
    assert np.array_equal(by_rank[0][0], single.numpy()) and np.array_equal(by_rank[1][0], single.numpy())
    assert sorted(by_rank[0][1] + by_rank[1][1]) == list(range(11))
    s = sharding.summarize(single, {"base_category": [0, 1, 2, 3, 4, 5, 6, 7, 8, 13],
                                    "novel_category": [9, 10, 11, 12, 14, 15, 16, 17, 18]})
    lines = sharding.log_lines(s)
    assert lines[3].startswith("Val 2d result: mIoU_Base/mAcc_Base/allAcc_Base ") and len(lines) == 9
    assert 0 < s["All"]["mIoU"] < 1


def _grad_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    grads = {"b.kernel": torch.randn(27, 6, 8, generator=g), "a.bn.weight": torch.randn(8, generator=g), "c.kernel": torch.randn(8, 4, generator=g)}
    sharding.allreduce_mean_gradients(grads)
    q.put((rank, {k: v.numpy().copy() for k, v in grads.items()}))
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_is_the_mean():
    """Training step, N > 1: one bucketed all-reduce of the student gradients, averaged (DDP semantics)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    want = {}
    for r in range(2):
        g = torch.Generator().manual_seed(100 + r)
        for k, shape in (("b.kernel", (27, 6, 8)), ("a.bn.weight", (8,)), ("c.kernel", (8, 4))):
            want[k] = want.get(k, 0) + torch.randn(*shape, generator=g).numpy() / 2
    for r in range(2):
        for k in want:
            assert np.allclose(res[r][k], want[k], atol=1e-6)


def _bucket_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    order = [("out.kernel", (8, 4)), ("b2.bn.weight", (8,)), ("b2.kernel", (27, 6, 8)), ("b1.bn.bias", (8,)), ("b1.kernel", (27, 6, 8)), ("in.kernel", (27, 5, 8))]
    sink = sharding.GradientBuckets(order, "cpu", bucket_bytes=6000)          # 27*6*8*4 = 5184 bytes: several buckets
    g = torch.Generator().manual_seed(200 + rank)
    launched = []
    for name, shape in order:
        t = torch.randn(*shape, generator=g)
        if name.endswith(".kernel") and name != "out.kernel":
            sink.view(name).copy_(t)                                          # "the kernel wrote into the slice"
            sink.ready(name)
        else:
            sink.put(name, t)
        launched.append(sum(b["work"] is not None for b in sink.buckets))
    out = sink.finish()
    q.put((rank, len(sink.buckets), launched, {k: v.numpy().copy() for k, v in out.items()}))
    dist.destroy_process_group()


def test_two_rank_gradient_buckets_overlap_protocol():
    """sharding.GradientBuckets: slices in backward order, a bucket's all-reduce launched the moment its last gradient is ready (before
    the later gradients exist), finish() = the mean over the ranks; a gradient never announced is an error, one process is a pass-through."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    order = [("out.kernel", (8, 4)), ("b2.bn.weight", (8,)), ("b2.kernel", (27, 6, 8)), ("b1.bn.bias", (8,)), ("b1.kernel", (27, 6, 8)), ("in.kernel", (27, 5, 8))]
    want = {}
    for r in range(2):
        g = torch.Generator().manual_seed(200 + r)
        for k, shape in order:
            want[k] = want.get(k, 0) + torch.randn(*shape, generator=g).numpy() / 2
    for r in range(2):
        assert res[r][1] >= 3                                                  # several buckets ...
        assert res[r][2][-1] == res[r][1] and res[r][2][2] >= 1                 # ... the first launched while later gradients were still to come
        for k in want:
            assert res[r][3][k].shape == want[k].shape and np.allclose(res[r][3][k], want[k], atol=1e-6)
    # one process: nothing to reduce, the slices come back as they were written; a forgotten gradient is an error
    sink = sharding.GradientBuckets(order, "cpu")
    for k, shape in order[:-1]:
        sink.put(k, torch.full(shape, 2.0))
    with pytest.raises(RuntimeError, match="never marked ready"):
        sink.finish()
    sink.put(order[-1][0], torch.full(order[-1][1], 3.0))
    out = sink.finish()
    assert float(out["b1.kernel"].sum()) == 2.0 * 27 * 6 * 8 and float(out["in.kernel"][0, 0, 0]) == 3.0


def _syncbn_worker(rank, world, port, q):
    """SyncBatchNorm host logic with torch stand-ins for the kernels' fp64 column sums."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(7 + rank)
    n, C = (300, 250)[rank], 24                                           # different row counts per rank
    y = torch.randn(n, C, generator=g) * (1.0 + rank) + 0.3 * rank
    dout = torch.randn(n, C, generator=g)
    eps = 1e-5

    def col_sums(mean):
        return y.double().sum(0) if mean is None else ((y - mean).double() ** 2).sum(0)
    mean, var, n_tot = sharding.sync_batch_stats(col_sums, n, C, "cpu")
    # the per-step form: the row count is all-reduced ONCE and handed to every layer -- same statistics
    n_all = sharding.sync_row_count(n, "cpu")
    mean2, var2, n_tot2 = sharding.sync_batch_stats(col_sums, n, C, "cpu", n_total=n_all)
    assert n_all == n_tot == n_tot2 and torch.equal(mean, mean2) and torch.equal(var, var2)
    rm, rv = torch.zeros(C), torch.ones(C)
    sharding.sync_running_stats(rm, rv, mean, var, n_tot, 0.1)
    xhat = (y - mean) / torch.sqrt(var + eps)
    local = torch.cat([dout.double().sum(0), (dout.double() * xhat.double()).sum(0)])
    g_sums, l_sums = sharding.sync_bwd_sums(local)
    gamma = torch.linspace(0.5, 1.5, C)
    dy = gamma / torch.sqrt(var + eps) * (dout - g_sums[:C] / n_tot - xhat * g_sums[C:] / n_tot)
    grads = {"bn.weight": l_sums[C:].clone(), "bn.bias": l_sums[:C].clone()}
    sharding.allreduce_mean_gradients(grads)
    q.put((rank, {k: v.numpy().copy() for k, v in dict(mean=mean, var=var, rm=rm, rv=rv, dy=dy, dg=grads["bn.weight"], db=grads["bn.bias"]).items()}, n_tot))
    dist.destroy_process_group()


def test_two_rank_sync_batchnorm_equals_single_process_over_all_rows():
    """run/train.py:212-213 (SyncBatchNorm): two ranks with different scenes reproduce single-process BatchNorm over the
    concatenated rows -- statistics, running statistics (1e-6), dy and the DDP-averaged affine gradients (1e-5)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_syncbn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r, d, n_tot = q.get(timeout=120)
        got[r] = d
        assert n_tot == 550
    for p in procs:
        p.join(timeout=60)
    ys, douts = [], []
    for rank in range(2):
        g = torch.Generator().manual_seed(7 + rank)
        n, C = (300, 250)[rank], 24
        ys.append(torch.randn(n, C, generator=g) * (1.0 + rank) + 0.3 * rank)
        douts.append(torch.randn(n, C, generator=g))
    Y = torch.cat(ys).double().requires_grad_(True)
    gamma = torch.linspace(0.5, 1.5, 24).double().requires_grad_(True)
    beta = torch.zeros(24, dtype=torch.float64, requires_grad=True)
    rm, rv = torch.zeros(24, dtype=torch.float64), torch.ones(24, dtype=torch.float64)
    out = torch.nn.functional.batch_norm(Y, rm, rv, gamma, beta, training=True, momentum=0.1, eps=1e-5)
    # DDP averages the per-rank losses' gradients: loss = (1/2) sum_r <dout_r, out_r>
    (out * torch.cat(douts).double()).sum().mul(0.5).backward()
    for r in range(2):
        assert np.allclose(got[r]["mean"], Y.detach().mean(0).numpy(), atol=1e-6)
        assert np.allclose(got[r]["var"], Y.detach().var(0, unbiased=False).numpy(), atol=1e-6)
        assert np.allclose(got[r]["rm"], rm.numpy(), atol=1e-6) and np.allclose(got[r]["rv"], rv.numpy(), atol=1e-6)
        assert np.allclose(got[r]["dg"], gamma.grad.numpy(), atol=1e-5) and np.allclose(got[r]["db"], beta.grad.numpy(), atol=1e-5)
    dy_ref = Y.grad.numpy() * 2.0                                          # d/dY of sum_r <dout_r, out_r>
    assert np.allclose(np.concatenate([got[0]["dy"], got[1]["dy"]]), dy_ref, atol=1e-5)


def test_summary_and_log_lines_match_reference_validate(golden_dir):
    """The running Base/Novel/All numbers and the log strings of the product (sharding.summarize / log_lines on exact
    int64 counts) equal what the reference's validate() logged (tests/golden/ref_validate.npz), line for line."""
    import os
    from oracle import metric
    g = np.load(os.path.join(golden_dir, "ref_validate.npz"))
    C = int(g["test_classes"])
    ign = [int(v) for v in g["test_ignore_label"]]
    split = {"base_category": g["base_category"].tolist(), "novel_category": g["novel_category"].tolist()}
    counts = torch.zeros((3, C), dtype=torch.int64)
    ref = [str(s) for s in g["log_lines"]]
    n = int(g["num_scenes"])
    per = len(ref) // n
    for i in range(n):
        I, U, T = metric.intersection_and_union(g[f"s{i}_pred"], g[f"s{i}_label"], C, ign)
        counts += torch.from_numpy(np.stack([I, U - T + I, T]))
        lines = ["Process: [{}/{}]".format(i, n)] + sharding.log_lines(sharding.summarize(counts, split))
        assert lines == ref[i * per:(i + 1) * per]
    s = sharding.summarize(counts, split)
    assert np.float32(s["Base"]["mIoU"]) == np.float32(g["result"][0])
    assert np.float32(s["Novel"]["mIoU"]) == np.float32(g["result"][1])


def test_equal_steps_scene_ids():
    for n, w in ((4, 8), (5, 2), (7, 3), (8, 8), (1, 4), (312, 8)):
        parts = [sharding.equal_steps_scene_ids(n, r, w) for r in range(w)]
        assert len({len(p) for p in parts}) == 1 and len(parts[0]) == -(-n // w)
        assert set(sum(parts, [])) == set(range(n))              # every scene is used; surplus slots wrap around
    assert sharding.equal_steps_scene_ids(0, 0, 2) == []


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _train_worker(rank, world, port, tmp, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from geopurify_amd import train_driver as td
        torch.manual_seed(0)

        class Student(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.a, self.b, self.c = (torch.nn.Linear(3, 3) for _ in range(3))

            def get_param_groups(self):
                return {"input": list(self.a.parameters()), "middle": list(self.b.parameters()), "output": list(self.c.parameters())}

        class Model(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.affinity_student = Student()

            def forward(self, batch):
                s = self.affinity_student
                return s.c(s.b(s.a(batch))).pow(2).mean()

        model = Model()
        ids = sharding.equal_steps_scene_ids(3, rank, world)       # odd scene count on two ranks
        loader = [torch.full((4, 3), float(i + 1)) for i in ids]
        opt = td.build_optimizer(model.affinity_student, 1e-2, 0.0)
        sch = td.build_scheduler(opt, 1e-2, 0, 2, len(loader))
        args = td.gp_config.CfgNode({"epochs": 2, "print_freq": 1})
        td.train(model, opt, sch, loader, args, rank=rank, world=world)
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        q.put((rank, len(loader), sch.get_last_lr(), flat.numpy()))
    finally:
        dist.destroy_process_group()


def test_train_loop_two_ranks_odd_scene_count(tmp_path):
    """with num_scenes % world != 0 every rank must still run the same number of steps (one gradient
    all-reduce each), end with identical weights and identical learning rates -- and not hang."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == 2
    assert res[0][2] == res[1][2]
    assert np.array_equal(res[0][3], res[1][3])                   # same averaged gradients -> same weights on both ranks
