"""The C-ABI library loads on a CPU-only host and exports every symbol include/*.h declares."""
import ctypes
import os
import re

from geopurify_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "geopurify_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gp_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert len(names) >= 30
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_table_matches_header():
    assert sorted(_lib.SIGNATURES.keys()) == _declared()
    lib = _lib.load()
    assert lib.gp_version() >= 100


def test_size_queries_without_gpu():
    lib = _lib.load()
    assert lib.gp_voxelize_workspace_bytes(150000) > 150000 * 8 * 5
    assert lib.gp_knn_workspace_bytes(1000) >= 8000
    ext = (ctypes.c_int32 * 3)(550, 550, 140)
    assert lib.gp_grid_bytes(125000, ext) > 0
    bad = (ctypes.c_int32 * 3)(0, 5, 5)
    assert lib.gp_grid_bytes(10, bad) == 0
    assert lib.gp_nn1_workspace_bytes(1000, 100) > 0


def test_debug_knobs_reject_what_is_not_in_the_table():
    """gp_debug_set is a declared export; a key or value outside the table of csrc/error.hip is GP_EINVAL
    (host-side check, no GPU needed).  The engine is no longer a knob (11 = 8): it is gp_pool_cs_apply_engine."""
    lib = _lib.load()
    EINVAL = -22
    for key, value in ((0, 0), (16, 0), (-1, 0),                 # no such knob
                       (8, 32), (8, -1),                         # matrix-core affinity mask (round 5): bits 0-4 only
                       (4, 1024), (4, 2048), (4, -1),            # pooling mask: bits >= 10 are undefined
                       (3, 64), (11, 8), (11, 1), (7, 1), (5, 2), (15, 3), (1, 5), (10, 65)):
        assert lib.gp_debug_set(key, value) == EINVAL, (key, value)
        assert b"gp_debug_set" in lib.gp_last_error()
    for key, value in ((4, 9), (4, 128), (4, 0), (3, 2), (3, 32), (3, 0), (11, 4), (11, 0), (15, 2), (15, 0), (7, 2), (7, 0), (8, 31), (8, 0)):
        assert lib.gp_debug_set(key, value) == 0, (key, value)
    # a stamp buffer comes with its size
    assert lib.gp_debug_ptr(0, None, 64) == EINVAL and lib.gp_debug_ptr(4, None, 0) == EINVAL
    assert lib.gp_debug_ptr(0, None, 0) == 0


def test_lift_views_workspace_is_sized_by_fill_capacity():
    """ADVICE r2 / VERDICT r3 next 8: the in-view fill's partial arrays (192 B per unit) are sized by a capacity in fill queries,
    not by the entry count.  At config-M entry counts (80 views x 45k visible points) the default capacity of ops.lift_masks_views
    (a quarter of the entries) takes 0.5 GB off the query; capacity 0 means `total` (the round-3 size)."""
    lib = _lib.load()
    total, n = 80 * 45_000, 500_000
    args = (80, 200, 128, 160)
    full = lib.gp_lift_masks_views_workspace_bytes(*args, total, n, 0)
    assert full == lib.gp_lift_masks_views_workspace_bytes(*args, total, n, total)
    quarter = lib.gp_lift_masks_views_workspace_bytes(*args, total, n, total // 4)
    assert full - quarter == 16 * (8 + 4) * (total - total // 4)
    assert full - quarter > 500e6
    assert lib.gp_lift_masks_views_workspace_bytes(*args, total, n, total + 1) == 0       # a capacity beyond the entries is an error
