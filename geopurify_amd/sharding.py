"""Scene sharding across GPUs (one process per GPU) and the single metric all-reduce.

The reference shards evaluation by hand: `--split_idx/--split_total` pick a contiguous slice of the
scene list (run/validation.py:269-286) and run/val.sh runs the slices one after the other, never
merging their metrics; its per-scene dist.all_reduce calls are dead code (run/validation.py:441-450).
Here: scenes are independent, so each rank evaluates its own slice with no data-path collective and
ONE int64 all-reduce of the [3,C] (intersection, output, target) counts follows the local loop
(RCCL over xGMI on the GPU box, gloo in the CPU tests).  int64 keeps the counts exact (the reference
accumulates them in fp32).
"""
import numpy as np
import torch


def get_batch_scenes(scene_ids, batch_idx, total_batches=4):
    """Contiguous split rule of run/validation.py:269-280: the first (n mod t) slices get one extra scene."""
    n = len(scene_ids)
    size, rem = divmod(n, total_batches)
    start = batch_idx * size + min(batch_idx, rem)
    return scene_ids[start:start + size + (1 if batch_idx < rem else 0)]


def equal_steps_scene_ids(num_scenes, rank, world_size):
    """Training shards: every rank gets the SAME number of steps, ceil(n / world) -- one gradient all-reduce per step
    needs every rank in every collective.  The id list is padded by wrapping around (what torch's DistributedSampler does,
    the sampler the reference's DDP path would use), then dealt contiguously.  Deterministic, rank-independent length."""
    if num_scenes <= 0:
        return []
    per = -(-num_scenes // world_size)
    ids = [i % num_scenes for i in range(per * world_size)]
    return ids[rank * per:(rank + 1) * per]


def assign_scenes_lpt(costs, world_size):
    """Greedy longest-processing-time assignment (scene sizes span 28k..302k points):
    scenes sorted by cost descending, each to the currently lightest rank.  Deterministic.
    Returns a list of index lists, one per rank."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world_size
    out = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += costs[i]
    return out


def reduce_counts(counts, group=None):
    """The one collective: SUM all-reduce of the int64 [3,C] counts (no-op without a process group)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    return counts


def allreduce_mean_gradients(grads, group=None):
    """Data-parallel training (run/train.py:206 wraps the student in DDP): ONE bucketed all-reduce of all student
    gradients (63.9 M fp32 = 256 MB at the reference shape) over RCCL/xGMI, averaged over the ranks.
    grads: dict name -> tensor, updated in place; no-op without a process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return grads
    names = sorted(grads)
    flat = torch.cat([grads[n].reshape(-1).float() for n in names])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= dist.get_world_size(group)
    off = 0
    for n in names:
        k = grads[n].numel()
        grads[n].copy_(flat[off:off + k].view_as(grads[n]))
        off += k
    return grads


class GradientBuckets:
    """Data-parallel gradient averaging OVERLAPPED with the backward pass (run/train.py:206 wraps the student in DistributedDataParallel,
    whose reducer does this with autograd hooks; here the backward pass is the library's own, so it says itself when a gradient is done).
    The gradients of a step live in a few flat buckets laid out in the ORDER THE BACKWARD PASS PRODUCES THEM (output layer first); the
    weight-gradient kernels write straight into their slice (`view(name)` is the kernel's output buffer: no concatenation, no copy back),
    and as soon as the last gradient of a bucket is `ready` its SUM all-reduce is launched asynchronously (RCCL's own stream, ordered
    behind the producing kernels by the process group; on xGMI a 60-MB bucket is well under a layer's 4-5 ms of backward work), while the
    next layers' data and weight gradients run.  `finish()` waits for the handles, scales by 1 / world size and returns name -> tensor.
    order: list of (name, shape) in backward order.  Without a process group (or world size 1) nothing is launched."""

    def __init__(self, order, device, group=None, bucket_bytes=64 << 20, dtype=torch.float32):
        self.group, self.world = group, _world(group)
        self.layout, self.buckets = {}, []
        cur, cur_names, off = [], [], 0
        item = torch.empty(0, dtype=dtype).element_size()

        def close():
            nonlocal cur, cur_names, off
            if cur_names:
                self.buckets.append({"flat": torch.empty(off, dtype=dtype, device=device), "names": cur_names, "pending": len(cur_names),
                                     "work": None})
            cur, cur_names, off = [], [], 0
        for name, shape in order:
            n = 1
            for d in shape:
                n *= int(d)
            n_al = (n + 63) // 64 * 64                                  # slices start on 256-byte boundaries (16-byte vector stores)
            if cur_names and (off + n_al) * item > bucket_bytes:
                close()
            self.layout[name] = (len(self.buckets), off, n, tuple(int(d) for d in shape))
            cur_names.append(name)
            off += n_al
        close()

    def view(self, name):
        """the slice of `name` (its shape): the buffer the gradient is to be written into"""
        b, off, n, shape = self.layout[name]
        return self.buckets[b]["flat"][off:off + n].view(shape)

    def put(self, name, tensor):
        """a gradient that was produced elsewhere: copied into its slice, then ready"""
        self.view(name).copy_(tensor.reshape(self.layout[name][3]))
        self.ready(name)

    def ready(self, name):
        """the kernels writing `name` are enqueued on the current stream; launches the bucket's all-reduce when it was its last"""
        bk = self.buckets[self.layout[name][0]]
        bk["pending"] -= 1
        if bk["pending"] == 0 and self.world > 1:
            import torch.distributed as dist
            bk["work"] = dist.all_reduce(bk["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        """name -> averaged gradient (views of the buckets); every gradient must have been marked ready"""
        for bk in self.buckets:
            if bk["pending"] != 0:
                raise RuntimeError(f"GradientBuckets.finish: {bk['pending']} gradient(s) of {bk['names']} were never marked ready")
            if bk["work"] is not None:
                bk["work"].wait()
                bk["flat"].mul_(1.0 / self.world)
        return {name: self.view(name) for name in self.layout}


def _world(group=None):
    import torch.distributed as dist
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def sync_row_count(n_local, device, group=None):
    """Rows of all ranks, once per training step: submanifold convolutions keep the row set, so every BatchNorm layer of a step
    normalises over the same count (it used to ride in every layer's first all-reduce and cost a host sync each)."""
    import torch.distributed as dist
    if _world(group) == 1:
        return int(n_local)
    t = torch.tensor([float(n_local)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(round(float(t.item())))


def sync_batch_stats(col_sums_fn, n_local, channels, device, group=None, n_total=None):
    """SyncBatchNorm statistics (run/train.py:212-213: the reference converts the student to MinkowskiSyncBatchNorm, so a
    BatchNorm layer normalises with the mean / biased variance of the rows of ALL ranks).  Two passes like the single-process
    kernel (gp_col_stats), each followed by ONE small all-reduce of fp64 [C (+1)]:
        col_sums_fn(None)  -> fp64 [C] sum of this rank's rows;   all-reduce with the row count  -> global mean
        col_sums_fn(mean)  -> fp64 [C] sum of squared deviations from the GLOBAL mean; all-reduce -> global biased variance
    Returns (mean fp32 [C], var fp32 [C], n_total).  Without a process group (or world size 1) the all-reduces are skipped.
    n_total: the all-rank row count when the caller already has it (sync_row_count, once per step): the count then neither rides
    in the first all-reduce nor costs a host synchronisation here."""
    import torch.distributed as dist
    multi = _world(group) > 1
    if n_total is not None:
        s = col_sums_fn(None).clone()
        if multi:
            dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
        mean = (s / n_total).float()
    else:
        t = torch.empty(channels + 1, dtype=torch.float64, device=device)
        t[:channels] = col_sums_fn(None)
        t[channels] = float(n_local)
        if multi:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        n_total = int(round(float(t[channels].item())))
        mean = (t[:channels] / n_total).float()
    sq = col_sums_fn(mean).clone()
    if multi:
        dist.all_reduce(sq, op=dist.ReduceOp.SUM, group=group)
    var = (sq / n_total).float()
    return mean, var, n_total


def sync_running_stats(running_mean, running_var, mean, var, n_total, momentum):
    """running <- (1 - m) running + m batch, the batch variance unbiased over the rows of all ranks (torch semantics)."""
    unbiased = var * (n_total / (n_total - 1.0)) if n_total > 1 else var
    running_mean.mul_(1.0 - momentum).add_(mean, alpha=momentum)
    running_var.mul_(1.0 - momentum).add_(unbiased, alpha=momentum)


def sync_bwd_sums(local_sums_f64, group=None):
    """The two reduction vectors of the BatchNorm backward pass (sum dz | sum dz * xhat, fp64 [2C]) over all ranks.
    Returns (global sums fp32 [2C] for the dy formula, local sums fp32 [2C]): the affine gradients dgamma / dbeta stay LOCAL
    sums -- they are averaged with every other gradient by allreduce_mean_gradients, as DDP does for SyncBatchNorm."""
    import torch.distributed as dist
    local = local_sums_f64.float()
    g = local_sums_f64.clone()
    if _world(group) > 1:
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
    return g.float(), local


def evaluate_sharded(num_scenes, scene_counts_fn, num_classes, device, rank=0, world_size=1, costs=None,
                     policy="contiguous"):
    """Run scene_counts_fn(scene_index, counts) for this rank's scenes, then all-reduce.
    scene_counts_fn adds the scene's (I, O, T) histograms into `counts` (int64 [3,C] on `device`)."""
    ids = list(range(num_scenes))
    if policy == "lpt" and costs is not None:
        mine = assign_scenes_lpt(costs, world_size)[rank]
    else:
        mine = get_batch_scenes(ids, rank, world_size)
    counts = torch.zeros((3, num_classes), dtype=torch.int64, device=device)
    for i in mine:
        scene_counts_fn(i, counts)
    return reduce_counts(counts), mine


def summarize(counts, category_split=None):
    """mIoU / mAcc / allAcc for Base / Novel / All exactly as run/validation.py:490-523: the reference keeps the
    per-class counts as fp32 numpy vectors (histc output summed by AverageMeter) and evaluates I/(U+1e-10) on them, so the
    exact int64 counts are converted to fp32 here -- identical numbers (and identical log strings) while a class holds
    fewer than 2^24 points, where the reference's own fp32 sums stop being exact."""
    c = counts.detach().cpu().numpy().astype(np.float32)
    inter, out, tgt = c[0], c[1], c[2]
    union = out + tgt - inter

    def block(idx):
        i, u, t = inter[idx], union[idx], tgt[idx]
        iou, acc = i / (u + 1e-10), i / (t + 1e-10)
        return {"mIoU": np.mean(iou), "mAcc": np.mean(acc), "allAcc": sum(i) / (sum(t) + 1e-10),
                "iou_class": iou, "intersection": i, "union": u, "target": t}

    res = {"All": block(np.arange(len(inter)))}
    if category_split is not None:
        res["Base"] = block(np.asarray(category_split["base_category"], dtype=np.int64))
        res["Novel"] = block(np.asarray(category_split["novel_category"], dtype=np.int64))
    return res


def log_lines(summary):
    """The reference's log strings (run/validation.py:524-553), one list entry per logger.info call."""
    lines = []
    for name in ("Base", "Novel", "All"):
        if name in summary:
            s = summary[name]
            lines.append("Raw stats {}: intersection {}, union {}, target {}".format(name, s["intersection"], s["union"],
                                                                                    s["target"]))
    for name in ("Base", "Novel", "All"):
        if name in summary:
            s = summary[name]
            lines.append("Val 2d result: mIoU_{0}/mAcc_{0}/allAcc_{0} {1:.4f}/{2:.4f}/{3:.4f}.".format(
                name, s["mIoU"], s["mAcc"], s["allAcc"]))
    for name in ("Base", "Novel", "All"):
        if name in summary:
            lines.append("iou_class_{} '{}'".format(name, summary[name]["iou_class"]))
    return lines
