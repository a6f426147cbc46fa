"""Host mirror of the pieces of util/util.py the evaluation driver uses (run/validation.py:22-30):
AverageMeter, intersectionAndUnionGPU (on the HIP histogram kernel), LR helpers."""
import math

import torch

from . import ops


class AverageMeter:
    """Last value, weighted sum, count and mean of a stream of values (util/util.py:108-124): update(val, n) weighs val by n."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val, self.sum, self.count = val, self.sum + val * n, self.count + n
        self.avg = self.sum / self.count


def poly_learning_rate(base_lr, curr_iter, max_iter, power=0.9):
    """base_lr * (1 - t)^power, t = curr_iter / max_iter."""
    t = float(curr_iter) / max_iter
    return base_lr * (1 - t) ** power


def cosine_learning_rate(base_lr, curr_iter, max_iter):
    """Half a cosine period from base_lr down to 0."""
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
    if not (1 <= output.dim() <= 4 and output.shape == target.shape):
        raise AssertionError("intersectionAndUnionGPU: prediction and target need the same shape (1 to 4 dimensions)")
    flat_out, flat_tgt = output.view(-1), target.view(-1).to(output.device)
    for ig in ignore_indexs:                                  # (the reference's in-place edit of the caller's predictions)
        flat_out[flat_tgt == ig] = ig
    inter, n_out, n_tgt = (h.float() for h in intersection_union_counts(flat_out, flat_tgt, K, ignore_indexs))
    return inter, n_out + n_tgt - inter, n_tgt


def save_checkpoint(state, is_best, sav_path, filename="model_last.pth.tar"):
    """util/util.py:38-42."""
    import os
    import shutil
    path = os.path.join(sav_path, filename)
    torch.save(state, path)
    if is_best:
        shutil.copyfile(path, os.path.join(sav_path, "model_best.pth.tar"))


def export_pointcloud(name, points, colors=None, normals=None):
    """util/util.py:190-206 writes a point cloud through open3d; here a plain ASCII PLY with the same content (first batch
    element of [B,N,3] inputs, tensors moved to the host), so the driver's debug dumps work without open3d."""
    import numpy as np
    if len(points.shape) > 2:
        points = points[0]
        if normals is not None:
            normals = normals[0]
    if isinstance(points, torch.Tensor):
        points = points.detach().cpu().numpy()
        if normals is not None:
            normals = normals.detach().cpu().numpy()
    cols = [np.asarray(points, dtype=np.float64)]
    props = ["property double x", "property double y", "property double z"]
    if normals is not None:
        cols.append(np.asarray(normals, dtype=np.float64))
        props += ["property double nx", "property double ny", "property double nz"]
    if colors is not None:
        c = np.asarray(colors.detach().cpu().numpy() if isinstance(colors, torch.Tensor) else colors, dtype=np.float64)
        cols.append(np.clip(np.rint(c * 255.0), 0, 255))
        props += ["property uchar red", "property uchar green", "property uchar blue"]
    data = np.concatenate(cols, axis=1)
    with open(name, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\n%s\nend_header\n" % (data.shape[0], "\n".join(props)))
        ncol = data.shape[1]
        ncolor = 3 if colors is not None else 0
        for row in data:
            f.write(" ".join(["%.9g" % v for v in row[:ncol - ncolor]] + ["%d" % int(v) for v in row[ncol - ncolor:]]) + "\n")


def get_palette(num_cls=21, colormap="scannet"):
    """util/util.py:253-281 returns a flat [3*num_cls] list of 0..255 colour components (for PIL putpalette) read from the
    reference's per-dataset colour tables; plotting is out of scope, so the palette here is generated (golden-angle hues,
    class 0 light grey, last entry black for "unlabelled"): same shape and type, not the same colours."""
    import colorsys
    n = {"scannet": 21, "scannet_200": 201, "matterport": 21, "matterport_160": 161}.get(colormap, num_cls)
    out = []
    for i in range(n):
        if i == n - 1:
            rgb = (0, 0, 0)
        elif i == 0:
            rgb = (174, 199, 232)
        else:
            r, g, b = colorsys.hsv_to_rgb((i * 0.61803398875) % 1.0, 0.55 + 0.3 * ((i * 7) % 3) / 2.0, 0.95 - 0.25 * ((i * 5) % 4) / 3.0)
            rgb = (int(r * 255), int(g * 255), int(b * 255))
        out.extend(rgb)
    return out
