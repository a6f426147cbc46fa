"""Host mirror of the pieces of util/util.py the evaluation driver uses (run/validation.py:22-30):
AverageMeter, intersectionAndUnionGPU (on the HIP histogram kernel), LR helpers."""
import math

import torch

from . import ops


class AverageMeter:
    """util/util.py:108-124."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def poly_learning_rate(base_lr, curr_iter, max_iter, power=0.9):
    return base_lr * (1 - float(curr_iter) / max_iter) ** power


def cosine_learning_rate(base_lr, curr_iter, max_iter):
    return base_lr * 0.5 * (1 + math.cos(math.pi * curr_iter / max_iter))


def intersection_union_counts(output, target, K, ignore_indexs=(255,), counts=None):
    """Exact int64 (intersection, output, target) histograms [3,K] on the device, accumulated into
    `counts` if given (the form the one RCCL all-reduce sums)."""
    output = output.reshape(-1).to(torch.int64).contiguous()
    target = target.reshape(-1).to(torch.int64).contiguous()
    if counts is None:
        counts = torch.zeros((3, K), dtype=torch.int64, device=output.device)
    ops.iou_hist(output, target, K, list(ignore_indexs), counts)
    return counts


def intersectionAndUnionGPU(output, target, K, ignore_indexs=[255]):
    """util/util.py:160-177.  Same return convention (three fp32 [K] tensors on the GPU); like the
    reference it overwrites `output` in place where the target is an ignore id."""
    assert output.dim() in [1, 2, 3, 4]
    assert output.shape == target.shape
    output = output.view(-1)
    target = target.view(-1).to(output.device)
    for ig in ignore_indexs:
        output[target == ig] = ig
    c = intersection_union_counts(output, target, K, ignore_indexs)
    inter, out, tgt = c[0].float(), c[1].float(), c[2].float()
    return inter, out + tgt - inter, tgt
