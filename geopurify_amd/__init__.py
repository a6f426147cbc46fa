"""geopurify_amd -- MI355X-native (gfx950) implementation of GeoPurify's per-scene hot path:
2D->3D feature lift + geometry-guided affinity pooling, behind the reference's own
models.affinity_module / dataset.* Python surface.  HIP kernels live in csrc/ and are reached
through the C-ABI library libgeopurify_hip.so (include/geopurify_hip.h)."""
__version__ = "0.1.0"
