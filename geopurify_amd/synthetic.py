"""Seeded synthetic ScanNet/Matterport-shaped scenes.

There are no datasets or checkpoints offline, so the hot path is driven by synthetic inputs of the
reference's shapes: a room (floor + 4 walls) with axis-aligned boxes sampled on a jittered lattice,
pinhole views with analytically ray-cast depth maps (so occlusion is real), and synthetic X-Decoder
outputs (pred_masks / pred_logits / mask_embed / text_embed / logit_scale) per view.

Everything here is input generation (numpy, host side) -- it is outside every timed region.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

SCANNET_K_NATIVE = np.array([[1170.187988, 0, 647.75, 0], [0, 1170.187988, 483.75, 0],
                             [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float64)


@dataclass
class SceneConfig:
    name: str = "S"
    num_points: int = 150_000
    num_views: int = 25
    feat_dim: int = 512
    num_classes: int = 19            # len(all_label)
    ignore_ids: tuple = (19, 20)     # test_ignore_label
    num_queries: int = 200
    image_dim: tuple = (648, 484)    # (W, H) = fusion.img_dim
    mask_shape: tuple = (484, 648)   # (H, W) = cfg.mask_shape
    voxel_size: float = 0.02
    pitch: float = 0.022
    cut_bound: int = 10
    vis_thres: float = 0.05
    dataset: str = "scannet"         # or "matterport"
    depth_scale: float = 1000.0
    dense_features: bool = False     # P config: dense 64-d feature maps instead of masks
    min_visible: int = 400


CONFIGS = {
    "P": SceneConfig(name="P", num_points=50_000, num_views=1, feat_dim=64, dense_features=True),
    "S": SceneConfig(name="S"),
    "M": SceneConfig(name="M", num_points=500_000, num_views=80, num_classes=160, ignore_ids=(255,),
                     image_dim=(640, 512), mask_shape=(512, 640), vis_thres=0.02,
                     dataset="matterport", depth_scale=4000.0),
    # small shapes for CPU-oracle parity tests
    "T": SceneConfig(name="T", num_points=6000, num_views=3, feat_dim=32, num_queries=24,
                     image_dim=(160, 120), mask_shape=(120, 160), pitch=0.035, min_visible=50),
}


@dataclass
class View:
    pose: np.ndarray                 # scannet: world_view_transform = W2C^T float32 [4,4]; matterport: c2w float32
    K: np.ndarray                    # scannet: native 4x4 colour intrinsics; matterport: 3x3 at image_dim
    depth: np.ndarray                # float64 [H,W] metres (0 = hole)


@dataclass
class Scene:
    cfg: SceneConfig
    coords: np.ndarray               # float64 [N,3]
    colors: np.ndarray               # float64 [N,3] in [0,1]
    normals: np.ndarray              # float64 [N,3]
    labels: np.ndarray               # int64 [N]
    views: List[View] = field(default_factory=list)
    boxes: Optional[np.ndarray] = None   # [B,2,3] lo/hi incl. the room as box 0


# ------------------------------------------------------------------------------------------
def _sample_rect(rng, origin, eu, ev, lu, lv, pitch, normal, label):
    nu, nv = max(int(lu / pitch), 1), max(int(lv / pitch), 1)
    gu, gv = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    u = (gu.reshape(-1) + 0.5 + rng.uniform(-0.3, 0.3, nu * nv)) * pitch
    v = (gv.reshape(-1) + 0.5 + rng.uniform(-0.3, 0.3, nu * nv)) * pitch
    p = origin[None] + u[:, None] * eu[None] + v[:, None] * ev[None]
    p = p + normal[None] * rng.normal(0, 0.0015, size=(p.shape[0], 1))
    return p, np.tile(normal, (p.shape[0], 1)), np.full(p.shape[0], label, np.int64)


def _box_faces(lo, hi, inward):
    ex, ey, ez = np.eye(3)
    d = hi - lo
    s = -1.0 if inward else 1.0
    faces = [
        (lo, ex, ey, d[0], d[1], -s * ez),                                   # bottom
        (np.array([lo[0], lo[1], hi[2]]), ex, ey, d[0], d[1], s * ez),       # top
        (lo, ex, ez, d[0], d[2], -s * ey),                                   # y = lo
        (np.array([lo[0], hi[1], lo[2]]), ex, ez, d[0], d[2], s * ey),       # y = hi
        (lo, ey, ez, d[1], d[2], -s * ex),                                   # x = lo
        (np.array([hi[0], lo[1], lo[2]]), ey, ez, d[1], d[2], s * ex),       # x = hi
    ]
    return faces


def _geometry(rng, cfg):
    """Room + boxes whose total surface area gives ~num_points at the lattice pitch."""
    target_area = cfg.num_points * cfg.pitch ** 2 * 1.04
    nbox = int(rng.integers(8, 17))
    # unit layout, scaled afterwards
    room = np.array([7.0, 5.0, 2.6]) * rng.uniform(0.9, 1.1, 3)
    boxes = []
    for _ in range(nbox):
        sz = rng.uniform([0.4, 0.4, 0.3], [1.6, 1.2, 1.4])
        lo = np.array([rng.uniform(0.2, room[0] - sz[0] - 0.2), rng.uniform(0.2, room[1] - sz[1] - 0.2), 0.0])
        boxes.append((lo, lo + sz))

    def area(scale):
        zs = min(1.0, scale * 1.6)
        r = room * np.array([scale, scale, zs])
        a = r[0] * r[1] + 2 * (r[0] + r[1]) * r[2]          # floor + walls (no ceiling)
        for lo, hi in boxes:
            d = (hi - lo) * np.array([scale, scale, zs])
            a += d[0] * d[1] + 2 * (d[0] + d[1]) * d[2]     # top + sides
        return a

    lo_s, hi_s = 0.05, 20.0
    for _ in range(60):
        mid = 0.5 * (lo_s + hi_s)
        if area(mid) < target_area:
            lo_s = mid
        else:
            hi_s = mid
    sc = np.array([hi_s, hi_s, min(1.0, hi_s * 1.6)])
    room = room * sc
    boxes = [(lo * sc, hi * sc) for lo, hi in boxes]
    return room, boxes


def make_scene(cfg: SceneConfig, seed: int = 5557) -> Scene:
    rng = np.random.default_rng(seed)
    room, boxes = _geometry(rng, cfg)
    C = cfg.num_classes
    pts, nrm, lab = [], [], []
    zero = np.zeros(3)
    faces = _box_faces(zero, room, inward=True)
    for fi, (o, eu, ev, lu, lv, n) in enumerate(faces):
        if fi == 1:
            continue                                         # no ceiling
        p, nn, ll = _sample_rect(rng, o, eu, ev, lu, lv, cfg.pitch, n, (0 if fi >= 2 else 1) % C)
        pts.append(p), nrm.append(nn), lab.append(ll)
    for bi, (lo, hi) in enumerate(boxes):
        for fi, (o, eu, ev, lu, lv, n) in enumerate(_box_faces(lo, hi, inward=False)):
            if fi == 0:
                continue                                     # bottom face sits on the floor
            p, nn, ll = _sample_rect(rng, o, eu, ev, lu, lv, cfg.pitch, n, (2 + bi) % C)
            pts.append(p), nrm.append(nn), lab.append(ll)
    pts, nrm, lab = np.concatenate(pts), np.concatenate(nrm), np.concatenate(lab)
    # remove points inside other boxes (hidden floor / box overlap), then trim/pad to exactly N
    keep = np.ones(len(pts), bool)
    for lo, hi in boxes:
        inside = np.all((pts > lo + 0.004) & (pts < hi - 0.004), axis=1)
        keep &= ~inside
    pts, nrm, lab = pts[keep], nrm[keep], lab[keep]
    N = cfg.num_points
    if len(pts) >= N:
        sel = np.sort(rng.choice(len(pts), N, replace=False))
    else:                                                    # pad with jittered copies
        extra = rng.choice(len(pts), N - len(pts), replace=True)
        sel = np.concatenate([np.arange(len(pts)), extra])
    pts, nrm, lab = pts[sel].copy(), nrm[sel].copy(), lab[sel].copy()
    if len(sel) > len(np.unique(sel)):
        dup = np.concatenate([[False], np.diff(np.sort(sel)) == 0])
        pts[np.argsort(sel)[dup]] += rng.normal(0, 0.004, size=(dup.sum(), 3))
    perm = rng.permutation(N)                                # scans are not surface-ordered
    pts, nrm, lab = pts[perm], nrm[perm], lab[perm]
    ign = rng.random(N) < 0.05
    lab[ign] = cfg.ignore_ids[-1]
    colors = rng.uniform(0, 1, size=(N, 3))
    scene = Scene(cfg, pts, colors, nrm, lab)
    scene.boxes = np.array([[zero, room]] + [[lo, hi] for lo, hi in boxes])
    for _ in range(cfg.num_views):
        scene.views.append(_make_view(rng, cfg, room, scene.boxes))
    return scene


# ------------------------------------------------------------------------------------------
def look_at_w2c(eye, target, up=(0.0, 0.0, 1.0)):
    eye, target, up = (np.asarray(v, dtype=np.float64) for v in (eye, target, up))
    z = target - eye
    z = z / np.linalg.norm(z)
    x = np.cross(z, up)
    x = x / np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z])
    w2c = np.eye(4)
    w2c[:3, :3] = R
    w2c[:3, 3] = -R @ eye
    return w2c


def mapper_intrinsics(cfg: SceneConfig, K):
    """Pinhole K actually used for projection at image_dim (mirrors fusion_util.py:86-96 for
    ScanNet; Matterport passes per-view 3x3 K as is)."""
    if cfg.dataset == "scannet":
        Kc = np.array(K, dtype=np.float64).copy()
        sx = cfg.image_dim[0] / (Kc[0, 2] * 2)
        sy = cfg.image_dim[1] / (Kc[1, 2] * 2)
        Kc[0, 0] *= sx
        Kc[1, 1] *= sy
        Kc[0, 2] = cfg.image_dim[0] / 2
        Kc[1, 2] = cfg.image_dim[1] / 2
        return Kc
    return np.asarray(K, dtype=np.float64)


def _raycast_depth(w2c, K, image_dim, boxes):
    """z-depth of the first surface hit per pixel: the room (from inside) and the boxes (slab test)."""
    W, H = image_dim
    R, t = w2c[:3, :3], w2c[:3, 3]
    eye = -R.T @ t
    u, v = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    dc = np.stack([(u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], np.ones_like(u)], -1)  # z_cam = 1
    dw = dc @ R                                              # world direction per unit camera depth
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / dw
        best = np.full((H, W), np.inf)
        for bi, (lo, hi) in enumerate(boxes):
            t0 = (lo - eye) * inv
            t1 = (hi - eye) * inv
            tn = np.minimum(t0, t1).max(-1)
            tf = np.maximum(t0, t1).min(-1)
            if bi == 0:                                      # inside the room: exit point, but not the ceiling
                hit = tf
                zc = eye[2] + dw[..., 2] * tf
                ok = (tf > 0) & ~(np.isclose(zc, hi[2]) & (dw[..., 2] > 0))
            else:
                hit = tn
                ok = (tn <= tf) & (tn > 0)
            best = np.where(ok & (hit < best), hit, best)
    best[~np.isfinite(best)] = 0.0
    return best


def _make_view(rng, cfg, room, boxes):
    W, H = cfg.image_dim
    for _ in range(50):
        # stand near a corner, look across the room (long sight lines, like a hand-held scan)
        cx, cy = rng.integers(0, 2, 2)
        fx, fy = rng.uniform(0.08, 0.3, 2)
        eye = np.array([(fx if cx == 0 else 1 - fx) * room[0], (fy if cy == 0 else 1 - fy) * room[1],
                        rng.uniform(0.45, 0.7) * room[2]])
        if any(np.all((eye > lo - 0.05) & (eye < hi + 0.05)) for lo, hi in boxes[1:]):
            continue
        tgt = np.array([(rng.uniform(0.5, 1.0) if cx == 0 else rng.uniform(0.0, 0.5)) * room[0],
                        (rng.uniform(0.5, 1.0) if cy == 0 else rng.uniform(0.0, 0.5)) * room[1],
                        rng.uniform(0.05, 0.4) * room[2]])
        break
    w2c = look_at_w2c(eye, tgt)
    if cfg.dataset == "scannet":
        K = SCANNET_K_NATIVE.copy()
        pose = w2c.T.astype(np.float32)                      # world_view_transform
        Kp = mapper_intrinsics(cfg, K)
        w2c_used = pose.T.astype(np.float64)
    else:
        f = rng.uniform(520, 560)
        K = np.array([[f, 0, W / 2 - 0.5 + rng.uniform(-4, 4)], [0, f, H / 2 - 0.5 + rng.uniform(-4, 4)],
                      [0, 0, 1.0]])
        pose = np.linalg.inv(w2c).astype(np.float32)         # camera_to_world
        Kp = K
        w2c_used = np.linalg.inv(pose)
    depth = _raycast_depth(w2c_used, Kp, cfg.image_dim, boxes)
    depth = depth + (depth > 0) * rng.normal(0, 0.005, size=depth.shape)
    q = cfg.depth_scale
    depth = np.round(np.maximum(depth, 0) * q) / q           # PNG integer depth / depth_scale
    return View(pose, K, depth)


# ------------------------------------------------------------------------------------------
def make_vlm_outputs(cfg: SceneConfig, num_views: int, seed: int, hw=None):
    """Synthetic X-Decoder outputs for `num_views` views (numpy, fp32).

    pred_masks [V,Q,h,w]: gaussian-blob logits at stride 4 of the image padded to a multiple of 32
    (SURVEY 8a'); pred_logits [V,Q,C+1]; mask_embed [V,Q,D]; text_embed [C,D]; logit_scale."""
    rng = np.random.default_rng(seed + 7919)
    H, W = cfg.mask_shape
    if hw is None:
        hw = (((H + 31) // 32) * 32 // 4, ((W + 31) // 32) * 32 // 4)
    h, w = hw
    Q, C, D = cfg.num_queries, cfg.num_classes, cfg.feat_dim
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    masks = np.empty((num_views, Q, h, w), np.float32)
    for v in range(num_views):
        cx = rng.uniform(0, w, Q).astype(np.float32)
        cy = rng.uniform(0, h, Q).astype(np.float32)
        sg = rng.uniform(0.04, 0.16, Q).astype(np.float32) * w
        amp = rng.uniform(6, 14, Q).astype(np.float32)
        off = rng.uniform(2, 5, Q).astype(np.float32)
        for q in range(Q):
            r2 = (xx - cx[q]) ** 2 + (yy - cy[q]) ** 2
            masks[v, q] = amp[q] * np.exp(-r2 / (2 * sg[q] ** 2)) - off[q]
        masks[v] += rng.normal(0, 0.05, size=(Q, h, w)).astype(np.float32)
    logits = (rng.normal(0, 1, size=(num_views, Q, C + 1)) * 4).astype(np.float32)
    embed = rng.normal(0, 1, size=(num_views, Q, D)).astype(np.float32)
    text = rng.normal(0, 1, size=(C, D)).astype(np.float32)
    return {"pred_masks": masks, "pred_logits": logits, "mask_embed": embed, "text_embed": text,
            "logit_scale": np.float32(np.exp(np.log(1 / 0.07)))}


def make_dense_feature_maps(cfg: SceneConfig, num_views: int, seed: int):
    """P config: dense per-pixel feature maps U(-1,1) [V,D,H,W] (lift row 5)."""
    rng = np.random.default_rng(seed + 104729)
    H, W = cfg.mask_shape
    return rng.uniform(-1, 1, size=(num_views, cfg.feat_dim, H, W)).astype(np.float32)
