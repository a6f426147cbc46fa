#!/usr/bin/env python3
"""Golden vectors from the reference's OWN FusedFeatureLoader (dataset/feature_loader.py:11-218) and its two collate
functions (:221-255), build container only.   Run:  python tests/golden/make_golden_feature_loader.py

dataset/point_loader.py imports SharedArray at module level (absent offline): an EMPTY placeholder module is registered
(memcache_init=False never touches it).  The loader reads real temp files written here: a ScanNet-style scene `.pth`
(coords, colours in [-1, 1], labels with -100) and fused-feature `.pt` files in the 2-key form {"feat", "mask_full"} and
the 3-key form {"feat", "mask", "mask_full"}; np.random is seeded before every __getitem__.  Cases: split val / train x
2-key / 3-key, eval_all on, input_color on; two occurrence files for one case (np.random.randint picks one).
Only inputs and outputs are stored."""
import collections
import collections.abc
import importlib.machinery
import os
import sys
import tempfile
import types

import numpy as np
import torch

collections.Sequence = collections.abc.Sequence
collections.Iterable = collections.abc.Iterable
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
m = types.ModuleType("SharedArray")
m.__spec__ = importlib.machinery.ModuleSpec("SharedArray", None)
sys.modules["SharedArray"] = m
# torch >= 2.6 defaults torch.load to weights_only=True, which refuses the numpy arrays of the scene files
_torch_load = torch.load
torch.load = lambda *a, **k: _torch_load(*a, **{**k, "weights_only": False})

from dataset.feature_loader import FusedFeatureLoader, collation_fn, collation_fn_eval_all  # noqa: E402


def main():
    rng = np.random.default_rng(77)
    N, D = 3000, 12
    locs = rng.uniform(0, 2.5, size=(N, 3)).astype(np.float32)
    locs[:, 2] *= 0.05
    cols = rng.uniform(-1, 1, size=(N, 3)).astype(np.float32)
    labs = rng.integers(0, 20, size=N).astype(np.float64)
    labs[rng.random(N) < 0.03] = -100
    mask_full = rng.random(N) < 0.6                                   # points inside the feature chunk
    n_in = int(mask_full.sum())
    feat2 = [rng.normal(size=(n_in, D)).astype(np.float32) for _ in range(2)]     # 2-key: one row per chunk point
    vis = np.sort(rng.choice(n_in, size=int(0.7 * n_in), replace=False))          # 3-key: rows of the chunk seen by a view
    feat3 = rng.normal(size=(n_in, D)).astype(np.float32)
    out = {"locs": locs, "cols": cols, "labs": labs, "mask_full": mask_full, "feat2_0": feat2[0], "feat2_1": feat2[1],
           "feat3": feat3, "mask_visible": vis, "voxel_size": np.float64(0.05)}
    cases = []
    for split in ("val", "train"):
        for form in ("2key", "3key"):
            with tempfile.TemporaryDirectory() as tmp:
                d3 = os.path.join(tmp, "scannet_3d")                  # dataset_name must be exactly "scannet_3d" (:92)
                os.makedirs(os.path.join(d3, split))
                fdir = os.path.join(tmp, "feat")
                os.makedirs(fdir)
                torch.save((locs.copy(), cols.copy(), labs.copy()), os.path.join(d3, split, "scene0001_00_vh_clean_2.pth"))
                if form == "2key":
                    for k in range(2):
                        torch.save({"feat": torch.from_numpy(feat2[k]), "mask_full": torch.from_numpy(mask_full)},
                                   os.path.join(fdir, f"scene0001_00_{k}.pt"))
                else:
                    torch.save({"feat": torch.from_numpy(feat3), "mask": torch.from_numpy(vis), "mask_full": torch.from_numpy(mask_full)},
                               os.path.join(fdir, "scene0001_00_0.pt"))
                ds = FusedFeatureLoader(d3, fdir, voxel_size=0.05, split=split, eval_all=True, input_color=True)
                name = f"{split}_{form}"
                np.random.seed(11)
                r = ds[0]
                for j, x in enumerate(r):
                    out[f"{name}_out_{j}"] = x.numpy()
                cases.append(name)
                if name == "val_2key":                                # both collates on two copies of the item
                    b = collation_fn_eval_all([tuple(t.clone() for t in r), tuple(t.clone() for t in r)])
                    for j, x in enumerate(b):
                        out[f"collate_eval_all_{j}"] = x.numpy()
                    b = collation_fn([tuple(t.clone() for t in r[:5]), tuple(t.clone() for t in r[:5])])
                    for j, x in enumerate(b):
                        out[f"collate_{j}"] = x.numpy()
    out["cases"] = np.array(cases)
    out["np_seed"] = np.int64(11)
    np.savez_compressed(os.path.join(HERE, "ref_feature_loader.npz"), **out)
    print("ok", cases, {k: out[f"{k}_out_3"].shape for k in cases})


if __name__ == "__main__":
    main()
