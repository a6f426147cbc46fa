"""compat shim: `util.config`."""
from geopurify_amd.config import *  # noqa: F401,F403
from geopurify_amd.config import CfgNode, load_cfg_from_cfg_file, merge_cfg_from_list  # noqa: F401
