import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from geopurify_amd import ops, pipeline as pl, synthetic as syn
cfg = syn.CONFIGS["S"]
scene = pl.upload_scene(syn.make_scene(cfg, 5557), "cuda")
rigid = pl.scene_rigid_transform(cfg.voxel_size, 5557)
vlm = pl.SyntheticVLM(syn.make_vlm_outputs(cfg, cfg.num_views, 5557), "cuda")
st = pl.StudentWeights(pl.random_student_state_dict(cfg.feat_dim + pl.GEO_DIM, hidden=128, embed=128, num_blocks=1, seed=1), "cuda")
hp = pl.HotPath(st, cfg.mask_shape, K=16, num_iters=1, device="cuda")
batch = pl.build_scene_batch(scene, rigid, "cuda")
cap = {}
orig = ops.nn1_masked
def spy(xyz, rm, qm, workspace=None):
    if xyz.shape[0] == batch.scene_coords.shape[0]:
        cap["args"] = (xyz, rm.clone(), qm.clone())
    return orig(xyz, rm, qm, workspace=workspace)
ops.nn1_masked = spy
hp.lift_masks(batch, vlm)
xyz, rm, qm = cap["args"]
N = xyz.shape[0]
nq, nr = int(qm.sum()), int(rm.sum())
print("N", N, "refs", nr, "queries", nq)
ext = (xyz.amax(0) - xyz.amin(0)).max().item()
h, h2 = ext / 128, ext / 32
q = xyz[qm.bool()].double(); r = xyz[rm.bool()].double()
d = torch.empty(nq, dtype=torch.float64, device="cuda")
for i in range(0, nq, 4096):
    d[i:i+4096] = torch.cdist(q[i:i+4096], r).min(1).values
print("extent", ext, "fine h", h, "coarse h", h2)
for k in (0.5, 1, 2, 3, 4, 6, 8, 16):
    print(f"  NN dist <= {k} fine cells: {(d <= k*h).float().mean().item():.3f}")
for k in (0.5, 1, 2, 3):
    print(f"  NN dist <= {k} coarse cells: {(d <= k*h2).float().mean().item():.3f}")
import time
for _ in range(2):
    torch.cuda.synchronize(); t=time.time(); orig(xyz, rm, qm); torch.cuda.synchronize(); print("nn1_masked ms", (time.time()-t)*1e3)
from geopurify_amd import _lib
lib = _lib.load()
ref = orig(xyz, rm, qm)
lib.gp_debug_set(7, 2)
for ng in (128, 64):
    lib.gp_debug_set(6, ng)
    out = orig(xyz, rm, qm); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t = time.time(); orig(xyz, rm, qm); torch.cuda.synchronize(); ts.append((time.time() - t) * 1e3)
    print(f"NG={ng}: {min(ts):.3f} ms  same={bool(torch.equal(out, ref))}")
lib.gp_debug_set(6, 0)
for ng2 in (32, 48):
    lib.gp_debug_set(13, ng2)
    out = orig(xyz, rm, qm); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t = time.time(); orig(xyz, rm, qm); torch.cuda.synchronize(); ts.append((time.time() - t) * 1e3)
    print(f"NG2={ng2}: {min(ts):.3f} ms  same={bool(torch.equal(out, ref))}")
lib.gp_debug_set(13, 0)
lib.gp_debug_set(7, 0)
for ng2 in (32, 40, 48, 56, 64):
    lib.gp_debug_set(13, ng2)
    out = orig(xyz, rm, qm); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t = time.time(); orig(xyz, rm, qm); torch.cuda.synchronize(); ts.append((time.time() - t) * 1e3)
    print(f"wave-only (default) NG2={ng2}: {min(ts):.3f} ms  same={bool(torch.equal(out, ref))}")
lib.gp_debug_set(13, 0); lib.gp_debug_set(7, 0)
