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
