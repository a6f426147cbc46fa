"""compat shim: `models.utils.visualization` -- the names the reference drivers import (run/validation.py:35:
visualize_2d_semantic, get_color_palette, save_3d_point_cloud; models/affinity_module.py:22: get_pca_color).
Debug plotting / PLY dumps are out of scope (matplotlib / open3d visualisers): the functions exist
so that the drivers import, accept the reference's arguments and do nothing."""


def visualize_2d_semantic(*args, **kwargs):
    return None


def get_color_palette(*args, **kwargs):
    from geopurify_amd.util import get_palette
    return get_palette(*args, **kwargs)


def save_3d_point_cloud(*args, **kwargs):
    return None


def get_pca_color(*args, **kwargs):
    raise NotImplementedError("PCA colouring of features is a debug visualiser (out of scope)")
