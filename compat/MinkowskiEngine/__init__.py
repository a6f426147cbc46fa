"""Kernel-free stand-in for the parts of MinkowskiEngine the reference DRIVERS touch: `from MinkowskiEngine import
SparseTensor` (run/validation.py:17, run/train.py:17 -- imported, never used by the drivers) and
`ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm(model)` (run/validation.py:201-202, run/train.py:212-213).
Everything numeric that the reference does through ME (SparseTensor maths, MinkowskiConvolution, MinkowskiBatchNorm) lives
in models/affinity_module.py, which geopurify_amd.affinity_module replaces with HIP kernels."""
from types import SimpleNamespace

import torch

__version__ = "0.0-geopurify-amd-stub"


class SparseTensor:
    """Data holder only: `.F` features [N,C], `.C` int32 coordinates [N,4] (batch index in column 0).  No kernels."""

    def __init__(self, features=None, coordinates=None, device=None, **kwargs):
        self.F = features if device is None or features is None else features.to(device)
        self.C = coordinates

    @property
    def features(self):
        return self.F

    @property
    def coordinates(self):
        return self.C


def _batched_coordinates(coords, dtype=torch.int32, device=None):
    """ME.utils.batched_coordinates: floor to int32 and prepend the batch index (models/affinity_module.py:1543)."""
    out = []
    for b, c in enumerate(coords):
        ci = torch.floor(torch.as_tensor(c).float()).to(dtype)
        out.append(torch.cat([torch.full((ci.shape[0], 1), b, dtype=dtype, device=ci.device), ci], 1))
    res = torch.cat(out)
    return res if device is None else res.to(device)


utils = SimpleNamespace(batched_coordinates=_batched_coordinates)


class MinkowskiSyncBatchNorm:
    @classmethod
    def convert_sync_batchnorm(cls, module, process_group=None):
        """Returns the module unchanged: nothing needs converting.  BatchNorm in training mode is computed by the HIP training step
        (geopurify_amd/training.py), which synchronises the batch statistics over the ranks by itself whenever torch.distributed is
        initialised with more than one rank (StudentTrainer(sync_bn=True): fp64 column sums all-reduced, sharding.sync_batch_stats) --
        what the reference obtains from this call (run/train.py:212-213)."""
        return module
