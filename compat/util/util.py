"""compat shim: `util.util`."""
from geopurify_amd.util import *  # noqa: F401,F403
