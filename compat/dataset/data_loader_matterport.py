"""compat shim: `dataset.data_loader_matterport` (run/validation.py:139-142): same three names; the Matterport variant is
selected by the synthetic config ("synthetic:M")."""
from geopurify_amd.data_loader import ScannetLoaderFull, SceneBatchSampler, scene_based_collate_fn  # noqa: F401
