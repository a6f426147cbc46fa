"""CPU oracle for the GeoPurify per-scene hot path  --  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy / torch-CPU / sklearn) of the reference
algorithm rows listed in SURVEY.md section 8(a).  It exists to CHECK the HIP path.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it; nothing under ``geopurify_amd/`` does, and the product
path raises when the HIP library is missing instead of falling back to this code.

Pinning status (see DESIGN.md "Oracle"):
  * rows 1-3, 13 (voxelizer, FNV hash, both point->pixel mappers, IoU counts) and the
    config loader are PINNED against golden vectors emitted by importing the
    reference's own numpy modules in the build container
    (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).
  * rows 5-12 sit on third-party engines that are absent from /root/reference
    (MinkowskiEngine, faiss, torch_scatter, sklearn KDTree ties, torch.sparse):
    "parity unpinned" for those boundaries; each is restated from the cited
    reference lines and cross-checked against an independent dense formulation
    (dense conv3d, brute-force (d2,id) kNN, dense A@X, bincount means).

All file:line citations are relative to /root/reference/.
"""
