"""compat shim: the reference's `models.affinity_module` names, served by geopurify_amd."""
from geopurify_amd.affinity_module import *  # noqa: F401,F403
from geopurify_amd.affinity_module import AffinityPredictor, MinkowskiResBlock, SonataXAffinityTrainer  # noqa: F401
