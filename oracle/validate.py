"""Oracle for the per-scene tail of validate() and its running meters -- TEST INFRASTRUCTURE ONLY.

Follows run/validation.py:413-439 (normalise, classify, arg-max; points whose feature row is all zero take the
prediction of the nearest point with a non-zero row, the distance measured on `scene_coords[:, 1:4]` of an [N,3]
tensor, i.e. on (y, z) only; intersectionAndUnionGPU) and :452-553 (AverageMeter sums of the fp32 count vectors for
Base / Novel / All, I/(U+1e-10) means, the log strings).  Pinned by tests/golden/ref_validate.npz, which the
reference's own validate() produced (tests/golden/make_golden_validate.py).
"""
import numpy as np
import torch
import torch.nn.functional as F
from sklearn.neighbors import KDTree

from . import metric


def scene_tail(scene_features, text_features, logit_scale, scene_coords, scene_label, num_classes, ignore_ids):
    """-> (pred int64 [N] after the zero-row fill, (I, U, T) int64 counts)."""
    f = F.normalize(scene_features, dim=-1)
    t = F.normalize(text_features, dim=-1)
    logits = logit_scale * (f @ t.t())
    pred = torch.max(logits, 1)[1]
    unseen = torch.sum(f.abs(), dim=1) == 0
    if unseen.any():
        seen = ~unseen
        seen_c = scene_coords[seen][:, 1:4]                    # [N,3] sliced 1:4 -> columns (y, z)   (:422-423)
        unseen_c = scene_coords[unseen][:, 1:4]
        if seen_c.shape[0] > 0:
            _, idx = KDTree(seen_c.numpy()).query(unseen_c.numpy(), k=1)
            src = torch.where(seen)[0][torch.from_numpy(idx.flatten())]
            pred[torch.where(unseen)[0]] = pred[src]
    return pred, metric.intersection_and_union(pred.numpy(), np.asarray(scene_label), num_classes, list(ignore_ids))


class Meters:
    """The nine AverageMeters of validate() reduced to their .sum vectors (fp32, like the reference's numpy arrays)."""

    def __init__(self, num_classes, base, novel):
        self.base, self.novel = np.asarray(base), np.asarray(novel)
        self.sum = np.zeros((3, num_classes), dtype=np.float32)

    def update(self, inter, union, target):
        self.sum += np.stack([inter, union, target]).astype(np.float32)

    def block(self, idx):
        i, u, t = self.sum[0][idx], self.sum[1][idx], self.sum[2][idx]
        iou, acc = i / (u + 1e-10), i / (t + 1e-10)
        return {"intersection": i, "union": u, "target": t, "iou_class": iou, "mIoU": np.mean(iou), "mAcc": np.mean(acc),
                "allAcc": sum(i) / (sum(t) + 1e-10)}

    def summary(self):
        return {"Base": self.block(self.base), "Novel": self.block(self.novel), "All": self.block(np.arange(self.sum.shape[1]))}


def log_lines(i, n, summary):
    """run/validation.py:486-553: the logger.info calls of one scene, in order."""
    out = ["Process: [{}/{}]".format(i, n)]
    for name in ("Base", "Novel", "All"):
        s = summary[name]
        out.append("Raw stats {}: intersection {}, union {}, target {}".format(name, s["intersection"], s["union"], s["target"]))
    for name in ("Base", "Novel", "All"):
        s = summary[name]
        out.append("Val 2d result: mIoU_{0}/mAcc_{0}/allAcc_{0} {1:.4f}/{2:.4f}/{3:.4f}.".format(name, s["mIoU"], s["mAcc"], s["allAcc"]))
    for name in ("Base", "Novel", "All"):
        out.append("iou_class_{} '{}'".format(name, summary[name]["iou_class"]))
    return out
