#!/usr/bin/env python3
"""Golden vectors from the reference's OWN scene_based_collate_fn (dataset/data_loader_ablation.py:429-495), build
container only.   Run:  python tests/golden/make_golden_collate.py

dataset/data_loader_ablation.py imports imageio, cv2, SharedArray (via dataset.point_loader), models.scene and plyfile
at module level; placeholder modules are registered for those names (the collate function touches none of them).
Inputs: per-view sample tuples with the reference's slot layout (:373-394), random small tensors, one dropped view.
Only inputs and outputs are stored."""
import collections
import collections.abc
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

collections.Sequence = collections.abc.Sequence
collections.Iterable = collections.abc.Iterable
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)


def _placeholder(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m


class _Unused:
    def __init__(self, *a, **k):
        raise RuntimeError("placeholder executed")


for n in ("imageio", "cv2", "SharedArray", "plyfile", "open3d"):
    _placeholder(n)
_placeholder("imageio.v2")
_placeholder("models.scene", Scene=_Unused)
_placeholder("models.utils.dataset_utils", load_point_ply=_Unused)

from dataset.data_loader_ablation import SceneBatchSampler, scene_based_collate_fn  # noqa: E402


def sample(rng, N, Nv, n_vis, nvv, H, W, view):
    t = torch.from_numpy
    mask = np.zeros(N, bool)
    mask[rng.choice(N, n_vis, replace=False)] = True
    return (t(rng.normal(size=(N, 3)).astype(np.float32)), t(rng.integers(0, 50, size=(Nv, 3)).astype(np.float32)),
            t(rng.integers(0, Nv, size=N)), t(rng.integers(0, 21, size=N)),
            t(np.c_[np.ones(n_vis), rng.normal(size=(n_vis, 3))].astype(np.float32)),
            t(np.c_[np.ones(nvv), rng.integers(0, 50, size=(nvv, 3))].astype(np.int32)), torch.ones(nvv, 3),
            t(rng.uniform(size=(n_vis, 6)).astype(np.float32)), t(rng.integers(0, 21, size=n_vis)),
            t(rng.integers(0, 2, size=n_vis).astype(np.float32)), t(rng.integers(0, 20, size=(H, W))),
            torch.full((H, W, 3), float(view)), t(rng.integers(10, H - 10, size=n_vis)), t(rng.integers(10, W - 10, size=n_vis)),
            t(mask), t(rng.integers(0, nvv, size=n_vis)), t(rng.integers(0, 9, size=(N, 4))),
            t(rng.integers(1, 9, size=(n_vis, 4))), None, t(rng.uniform(size=(N, 6)).astype(np.float32)))


def main():
    rng = np.random.default_rng(3)
    N, Nv, H, W = 60, 41, 24, 32
    views = [sample(rng, N, Nv, 17, 13, H, W, 0), None, sample(rng, N, Nv, 9, 8, H, W, 2), sample(rng, N, Nv, 22, 19, H, W, 3)]
    out = {"num_views": np.int64(len(views))}
    for i, v in enumerate(views):
        out[f"in{i}_none"] = np.bool_(v is None)
        if v is not None:
            for j, x in enumerate(v):
                if x is not None:
                    out[f"in{i}_{j}"] = x.clone().numpy()
    res = scene_based_collate_fn(views)
    for j, x in enumerate(res):
        if torch.is_tensor(x):
            out[f"out_{j}"] = x.numpy()
    assert res[18] == (None, None, None)
    assert scene_based_collate_fn([None, None]) is None
    s = SceneBatchSampler([{"scene_name": n} for n in ["a", "a", "b", "a", "c", "b"]], shuffle=False)
    out["sampler_batches"] = np.array([str(b) for b in s])
    np.savez_compressed(os.path.join(HERE, "ref_collate.npz"), **out)
    print("collate ok:", [tuple(x.shape) if torch.is_tensor(x) else x for x in res][:20], list(s))


if __name__ == "__main__":
    main()
