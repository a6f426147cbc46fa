"""Kernel-free stand-in for the parts of MinkowskiEngine the reference DRIVERS touch (run/validation.py:17,219,
run/train.py:211): `ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm(model)`.  Everything numeric that the reference
does through ME (SparseTensor, MinkowskiConvolution, MinkowskiBatchNorm) lives in models/affinity_module.py, which
geopurify_amd.affinity_module replaces with HIP kernels."""

__version__ = "0.0-geopurify-amd-stub"


class MinkowskiSyncBatchNorm:
    @classmethod
    def convert_sync_batchnorm(cls, module, process_group=None):
        return module
