"""compat shim: `dataset.feature_loader`."""
from geopurify_amd.feature_loader import *  # noqa: F401,F403
from geopurify_amd.feature_loader import FusedFeatureLoader, collation_fn, collation_fn_eval_all  # noqa: F401
