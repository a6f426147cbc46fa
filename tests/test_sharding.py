"""Scene sharding + the single metric all-reduce: split rule, LPT balance, and a world_size-2 gloo
run (CPU) whose reduced counts equal the single-process sum."""
import os
import socket

import numpy as np
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
