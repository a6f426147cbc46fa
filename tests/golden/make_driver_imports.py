#!/usr/bin/env python3
"""List the names the reference DRIVERS import (run/validation.py, run/train.py) -- names only, via ast; build container
only.  Output: tests/golden/reference_driver_imports.txt, one `driver module name scope` per line.  scope:
  in   = a module of the reference's own tree on the hot path's boundary: must resolve under compat/
  ext  = third-party package outside this library (tensorboardX, omegaconf, cv2, open3d, imageio, detectron2, xdecoder ...)
  std  = standard library / torch / numpy / sklearn (present in the image)"""
import ast
import os

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
IN_SCOPE = ("MinkowskiEngine", "util", "models", "dataset")
EXT = ("tensorboardX", "imageio", "cv2", "open3d", "omegaconf", "xdecoder", "detectron2")


def scope(mod):
    top = mod.split(".")[0]
    return "in" if top in IN_SCOPE else ("ext" if top in EXT else "std")


rows = set()
for drv in ("run/validation.py", "run/train.py"):
    tree = ast.parse(open(os.path.join(REF, drv)).read())
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            for a in node.names:
                rows.add((drv, a.name, "-", scope(a.name)))
        elif isinstance(node, ast.ImportFrom) and node.module:
            for a in node.names:
                rows.add((drv, node.module, a.name, scope(node.module)))
with open(os.path.join(HERE, "reference_driver_imports.txt"), "w") as f:
    for r in sorted(rows):
        f.write(" ".join(r) + "\n")
print(len(rows), "names;", sum(r[3] == "in" for r in rows), "in scope")
