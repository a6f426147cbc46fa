"""compat shim: `dataset.data_loader_ablation` (run/validation.py:143-149, run/train.py:30-34)."""
from geopurify_amd.data_loader import ScannetLoaderFull, SceneBatchSampler, scene_based_collate_fn  # noqa: F401
