"""Oracle row 13: classify + IoU counts (test infrastructure).

Follows util/util.py:145-177 (intersectionAndUnion / intersectionAndUnionGPU) and the per-scene
tail of run/validation.py:413-439.
"""
import numpy as np
import torch
import torch.nn.functional as F


def intersection_and_union(pred, target, num_classes, ignore_indexs):
    """util/util.py:160-177 on CPU: pred[target==ig] = ig for every ignore id (in place on a copy),
    then three histc(bins=C, min=0, max=C-1).  Values >= C fall outside and are dropped.
    Returned as exact int64 counts (the reference stores them in fp32)."""
    pred = np.asarray(pred).reshape(-1).astype(np.int64).copy()
    target = np.asarray(target).reshape(-1).astype(np.int64)
    for ig in ignore_indexs:
        pred[target == ig] = ig
    inter = pred[pred == target]

    def hist(v):
        v = v[(v >= 0) & (v <= num_classes - 1)]
        return np.bincount(v, minlength=num_classes).astype(np.int64)

    ai, ao, at = hist(inter), hist(pred), hist(target)
    return ai, ao + at - ai, at


def classify(scene_features, text_features, logit_scale):
    """run/validation.py:413-416: normalise both, logits = scale * F @ T^T, argmax."""
    f = F.normalize(scene_features, dim=-1)
    t = F.normalize(text_features, dim=-1)
    logits = logit_scale * (f @ t.t())
    return torch.max(logits, 1)[1], logits


def mean_iou(inter, union, index_list):
    """run/validation.py:490-523 style: mean over a category index list of I/(U+1e-10)."""
    iou = inter / (union + 1e-10)
    return float(np.mean(iou[index_list]))
