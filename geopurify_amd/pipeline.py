"""Device-resident per-scene hot path: loader math -> 2D->3D lift -> student -> kNN affinity -> pooling.

Host orchestration only: every numeric step is a HIP kernel reached through the C-ABI
(geopurify_amd.ops).  Mirrors, stage by stage, the reference call stack of SURVEY.md 3.1:
  dataset/data_loader_ablation.py:242-264,348-372  (mapper, visible lists, scene voxelization)
  models/affinity_module.py:455-714               (lift_xdecoder_features)   / :348-453 (dense lift)
  models/affinity_module.py:1491-1607             (evaluate_scene)
  run/validation.py:413-439                       (classify + IoU counts)
Voxel-level arrays live in an internal Morton order (perm/rank); outputs are per point, so the
order never leaks.  Host synchronisations per scene: voxel count + extent (one readback) and the
per-view visible counts (one readback).
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

from . import ops
from .bicubic import aa_bicubic_taps

GEO_DIM = 6
CONV_PAD = 32                     # gp_sparse_conv needs cin % 32 == 0


def _pad_to(v, m):
    return (v + m - 1) // m * m


# --------------------------------------------------------------------------------------------------
class StudentWeights:
    """Device copy of an AffinityPredictor state_dict (ME layout) with the input
    kernel zero-padded to a multiple of 32 channels and BatchNorm(eval) folded to scale/shift.

    mode "f16x3" (default): 3x3x3 layers run on the f16 matrix cores with fp32-class accuracy
    (gp_sparse_conv_f16x3: operands split hi+lo, weights pre-scaled by a power of two whose inverse
    is folded into the BN scale).  mode "f32": every layer on the exact-fp32 MFMA kernel."""

    def __init__(self, state_dict, device, eps=1e-5, mode="f16x3", residual_from_planes=True, interleaved_rows=True, blocked_weights=True,
                 head="f16x3"):
        """The four options are explicit arguments since round 6 (rounds 4-5: environment switches, retired with their last same-box A/B --
        DESIGN.md section 0): residual_from_planes (a residual block's input added back from the split planes its first convolution reads
        instead of fp32 rows the producer writes as well: 43.05 -> 43.47 scenes/s), interleaved_rows (the layers hand each other
        [K step][hi 32 | lo 32] rows, one full line per row and step: 44.0 -> 45.1), blocked_weights (step-blocked weight tiles:
        43.72 -> 44.68), head ("f16x3": the 1x1x1 output layer on the matrix cores fused with F.normalize, 276 -> 98 us; "f32")."""
        sd = {k: v.detach() for k, v in state_dict.items()}
        self.mode = mode
        self.residual_from_planes = bool(residual_from_planes)
        self.interleaved_rows = bool(interleaved_rows)
        self.blocked_weights = bool(blocked_weights)
        w0 = sd["input_layer.0.kernel"].float()
        self.cin = w0.shape[1]
        self.cin_pad = _pad_to(self.cin, CONV_PAD)
        self.hidden = w0.shape[2]
        w0p = torch.zeros((27, self.cin_pad, self.hidden), dtype=torch.float32)
        w0p[:, :self.cin] = w0
        self.layers = []          # (weights, (scale, shift)) per 3x3x3 conv, in execution order
        self._add_layer(w0p, self._fold(sd, "input_layer.1", eps), device)
        self.num_blocks = 0
        while f"res_blocks.{self.num_blocks}.conv1.kernel" in sd:
            i = self.num_blocks
            self._add_layer(sd[f"res_blocks.{i}.conv1.kernel"].float(), self._fold(sd, f"res_blocks.{i}.norm1", eps), device)
            self._add_layer(sd[f"res_blocks.{i}.conv2.kernel"].float(), self._fold(sd, f"res_blocks.{i}.norm2", eps), device)
            self.num_blocks += 1
        self.w_out = sd["output_layer.kernel"].float().to(device).contiguous()
        self.embed = self.w_out.shape[1]
        # the 1x1x1 output layer on the same f16 hi/lo operator as the 3x3x3 layers, fused with F.normalize (gp_embed_head_f16x3)
        self.head = None
        # (head="f32": the exact-fp32 kernel + l2norm_rows_ of rounds 1-3)
        if (mode == "f16x3" and self.embed == 128 and self.hidden % 64 == 0 and all(l[0] == "f16x3" for l in self.layers)
                and head != "f32"):
            amax = float(self.w_out.abs().max().item())
            p2 = 2.0 ** int(np.floor(np.log2(16384.0 / amax))) if amax > 0 else 1.0
            hi, lo = ops.conv_weights_split(self.w_out.reshape(1, self.hidden, self.embed), p2, blocked=False)   # the dense head's operand
            self.head = (hi, lo, 1.0 / p2)

    @staticmethod
    def _fold(sd, prefix, eps):
        s = sd[prefix + ".bn.weight"].float() / torch.sqrt(sd[prefix + ".bn.running_var"].float() + eps)
        b = sd[prefix + ".bn.bias"].float() - sd[prefix + ".bn.running_mean"].float() * s
        return s, b

    def _add_layer(self, w, bn, device):
        w = w.to(device).contiguous()
        scale, shift = bn[0].to(device).contiguous(), bn[1].to(device).contiguous()
        use_fast = self.mode == "f16x3" and w.shape[2] % 256 == 0
        if use_fast:
            # weights x 2^k with the largest magnitude in [2^13, 2^14): every weight within 2^-18 of it keeps a normal f16 lo
            # half (the inverse goes into the BatchNorm scale; powers of two, exact)
            amax = float(w.abs().max().item())
            p2 = 2.0 ** int(np.floor(np.log2(16384.0 / amax))) if amax > 0 else 1.0
            hi, lo = ops.conv_weights_split(w, p2, blocked=None if self.blocked_weights else False)
            self.layers.append(("f16x3", (hi, lo), (scale / p2).contiguous(), shift))
        else:
            self.layers.append(("f32", w, scale, shift))

    def _conv(self, li, x, ctx, residual=None, x_split=None, want_split=False, want_f32=True, il=False):
        """x_split / the returned split: (hi, lo, row_inv_scale) -- operands pre-scaled by a power of two per row -- or, il=True,
        (rows, None, row_inv_scale) with the hi / lo halves interleaved per 32-channel step (what the next convolution stages in full
        lines).  want_f32=False: the fp32 copy of the output is not written (only the next convolution reads this layer)."""
        kind, w, scale, shift = self.layers[li]
        if kind == "f16x3":
            out_split = None
            if want_split:
                nv, cout = ctx["pairs"].nv, ops.conv_weights_shape(w[0])[1]
                dev = w[0].device
                if il:
                    out_split = (torch.empty((nv, 2 * cout), dtype=torch.float16, device=dev), None, torch.empty(nv, dtype=torch.float32, device=dev))
                else:
                    out_split = (torch.empty((nv, cout), dtype=torch.float16, device=dev), torch.empty((nv, cout), dtype=torch.float16, device=dev),
                                 torch.empty(nv, dtype=torch.float32, device=dev))
            y = ops.sparse_conv_f16x3(x, ctx["pairs"], w[0], w[1], scale, shift, residual=residual, relu=True,
                                      x_split=x_split[:2] if x_split is not None else None,
                                      out_split=out_split[:2] if out_split is not None else None,
                                      x_row_inv=x_split[2] if x_split is not None else None,
                                      out_row_inv=out_split[2] if out_split is not None else None, want_f32=want_f32)
            return y, out_split
        return ops.sparse_conv(x, ctx["nbr_map"], w, scale, shift, residual=residual, relu=True), None

    @property
    def fast(self):
        """every 3x3x3 layer runs on the f16 hi/lo operator (operands pre-split, LDS-DMA staging)"""
        return all(l[0] == "f16x3" for l in self.layers)

    def split_input(self, x):
        """The first layer's operand in the form the fast path stages it: (hi, lo, row_inv_scale) of x[:, :cin_pad].  Needs the voxel
        means only, so a scheduler can run it ahead (HotPath.prepare)."""
        if not self.fast:
            return None
        return ops.split_f16(x, self.cin_pad, per_row=True, interleaved=self.residual_from_planes and self.interleaved_rows)

    def forward(self, x, nbr_map, pairs=None, x_split=None, mark=None, planes=False, plane_rows=None):
        """x fp32 [Nv, >=cin_pad] (internal order).  Returns L2-normalised embeddings [Nv, embed].
        On the f16x3 path every layer also emits its output pre-split (hi/lo f16) so that the next
        layer stages both operands by LDS-DMA.  x_split: split_input(x) when it was made ahead.
        planes=True (only with the fused output layer): returns the embeddings x 2^10 as f16 (hi, lo) planes INSTEAD -- the operand
        of the matrix-core affinity kernel, written by the output layer's epilogue (no fp32 rows, no split pass); plane_rows (i32 [Nv]):
        the plane row of voxel row r (the pooling operator's own row order, ops.rcb_order)."""
        ctx = {"nbr_map": nbr_map, "pairs": pairs}
        fast = self.fast
        if pairs is None and any(l[0] == "f16x3" for l in self.layers):
            ctx["pairs"] = ops.conv_pairs_build(nbr_map, col_tiles=max(1, self.hidden // 256))
        xs = (x_split if x_split is not None else self.split_input(x)) if fast else None
        # On the fast path a block's input is added back from the SPLIT PLANES its first convolution reads (hi + lo) * row scale -- the
        # value that convolution multiplies with -- so no layer writes fp32 rows unless the dense output layer needs them (no fused head).
        planes_res = fast and self.residual_from_planes
        # ... and every layer but the last writes its output as INTERLEAVED rows (the dense head reads separate planes)
        il = planes_res and self.interleaved_rows
        h, hs = self._conv(0, x, ctx, x_split=xs, want_split=fast, want_f32=not planes_res, il=il)
        for b in range(self.num_blocks):
            t, ts = self._conv(1 + 2 * b, h, ctx, x_split=hs, want_split=fast, want_f32=not fast, il=il)   # conv1 output: next conv only
            last = b == self.num_blocks - 1
            head = last and fast and self.head is not None          # the output layer reads the split planes only
            h, hs = self._conv(2 + 2 * b, t, ctx, residual=hs if planes_res else h, x_split=ts, want_split=fast and (not last or head),
                               want_f32=not (head or (planes_res and not last)), il=il and not last)
        self.last_pairs = ctx["pairs"]
        if mark is not None:
            mark("student convolutions")                  # (stage marks of bench.py's per-stage pass)
        if fast and self.head is not None and hs is not None:
            if planes:
                return ops.embed_head_f16x3(hs[:2], self.head[0], self.head[1], self.head[2], x_row_inv=hs[2], normalize=True,
                                            planes=True, want_f32=False, plane_rows=plane_rows)[1]
            return ops.embed_head_f16x3(hs[:2], self.head[0], self.head[1], self.head[2], x_row_inv=hs[2], normalize=True)
        e = ops.l2norm_rows_(ops.sparse_conv(h, None, self.w_out))
        if planes:
            return ops.split_f16(e, self.embed, scale=torch.tensor([ops.AFFINITY_PLANE_SCALE], dtype=torch.float32, device=e.device),
                                 dst_row=plane_rows)
        return e

    def flops(self, num_pairs, nv):
        per_pair = self.cin * self.hidden + 2 * self.num_blocks * self.hidden * self.hidden
        return 2.0 * num_pairs * per_pair + 2.0 * nv * self.hidden * self.embed


# --------------------------------------------------------------------------------------------------
@dataclass
class ViewLists:
    pt: torch.Tensor              # i64 [n_v] ascending ids of the visible points
    x: torch.Tensor               # i64 [n_v] pixel row  (x_label)
    y: torch.Tensor               # i64 [n_v] pixel col  (y_label)
    src_view: int                 # index of the view before dropping (for the VLM outputs)


@dataclass
class SceneBatch:
    """Device-side equivalent of the 20-tuple of scene_based_collate_fn
    (dataset/data_loader_ablation.py:429-495), holding what evaluate_scene reads."""
    scene_coords: torch.Tensor            # f32 [N,3]
    scene_coords_3d: torch.Tensor         # f32 [Nv,3] integer-valued voxel coords (hash order)
    scene_inds_reconstruct: torch.Tensor  # i64 [N]
    scene_label: torch.Tensor             # i64 [N]
    scene_gauss_features: torch.Tensor    # f32 [N,6]
    views: List[ViewLists] = field(default_factory=list)
    order: Optional[torch.Tensor] = None       # CSR of each voxel's points (from the voxelizer sort)
    seg_start: Optional[torch.Tensor] = None
    extent: Optional[list] = None              # host ints: max voxel coord + 1 per axis (min is 0)
    imgs: Optional[torch.Tensor] = None        # f32 [V,H,W,3] slot 11 of the tuple: what the 2D VLM is run on (:496)
    ent: Optional[dict] = None                 # all views' entries in one set of arrays (ops.views_visible_lists) + "total", "max_nv"

    def as_tuple(self):
        """The reference's positional 20-tuple (unused per-view voxelization slots are empty)."""
        dev = self.scene_coords.device
        N = self.scene_coords.shape[0]
        e = torch.empty(0, device=dev)
        V = len(self.views)
        ori = torch.cat([torch.cat([torch.full((len(v.pt), 1), float(i), device=dev),
                                    self.scene_coords[v.pt]], 1) for i, v in enumerate(self.views)]) if V else e
        mask = torch.zeros((V, N), dtype=torch.long, device=dev)
        for i, v in enumerate(self.views):
            mask[i, v.pt] = 1
        vid = torch.arange(V, device=dev).repeat_interleave(N)
        mask_2ds = torch.stack([vid, mask.reshape(-1)], 1)
        xl = torch.cat([v.x for v in self.views]) if V else e.long()
        yl = torch.cat([v.y for v in self.views]) if V else e.long()
        imgs = self.imgs if self.imgs is not None else e
        return (self.scene_coords, self.scene_coords_3d, self.scene_inds_reconstruct, self.scene_label, ori,
                e, e, e, e, e, e, imgs, xl, yl, mask_2ds, e, e, e, (None,) * V, self.scene_gauss_features)


def _view_matrices(cfg, views):
    """Per view: world->camera matrix and intrinsics at image_dim, as the mappers build them."""
    from .synthetic import mapper_intrinsics
    out = []
    for v in views:
        K = mapper_intrinsics(cfg, v.K)
        if cfg.dataset == "scannet":
            w2c = np.asarray(v.pose).T.astype(np.float64)                 # fusion_util.py:114 (W2C^T float32 -> .T)
        else:
            w2c = np.linalg.inv(np.asarray(v.pose))                       # fusion_util.py:60 (inverse of the fp32 c2w)
        out.append((w2c, K))
    return out


def _voxel_extent(vox):
    """max voxel coordinate + 1 per axis (the voxelizer shifts the minimum to 0), i64 [3] on the device."""
    ci = vox["coords_aug"].to(torch.int32)
    return ops.minmax_i32(ci.contiguous())[3:6].to(torch.int64) + 1


def build_scene_batch(scene, rigid, device="cuda", val_keep=10_000_000, batch_views=True, mark=None):
    """Loader math on the device for one synthetic scene (geopurify_amd.synthetic.Scene):
    scene voxelization (rows 1-2), per-view mapping (row 3), visible lists and the view-drop rule
    (data_loader_ablation.py:254-255,280-288).  batch_views: all views in one set of launches (needs the depth maps
    stacked on the device, upload_scene) -- the same lists as the view-by-view path, bit for bit.
    mark: optional callable(name) called on the host after each sub-stage's kernels are enqueued (bench.py's per-stage events)."""
    from .synthetic import mapper_intrinsics
    cfg = scene.cfg
    dev = torch.device(device)
    coords64 = scene.coords_dev if hasattr(scene, "coords_dev") else torch.from_numpy(scene.coords).to(dev)
    N = coords64.shape[0]
    vox = ops.voxelize(coords64, rigid)                                   # sync #1 (nv)
    if mark is not None:
        mark("voxelize")
    W, H = cfg.image_dim
    V = len(scene.views)
    gauss = scene.gauss_dev if hasattr(scene, "gauss_dev") else torch.from_numpy(
        np.concatenate([scene.colors, scene.normals], 1).astype(np.float32)).to(dev)
    labels = scene.labels_dev if hasattr(scene, "labels_dev") else torch.from_numpy(scene.labels).to(dev)
    if batch_views and V and getattr(scene, "depth_all_dev", None) is not None and V * N < 2 ** 31:
        mats = _view_matrices(cfg, scene.views)
        params = np.stack([np.concatenate([m.reshape(16), [K[0, 0], K[1, 1], K[0, 2], K[1, 2]]]) for m, K in mats]).astype(np.float64)
        ent = ops.views_visible_lists(coords64, torch.from_numpy(params).to(dev), scene.depth_all_dev, W, H, cfg.cut_bound,
                                      cfg.vis_thres, cfg.min_visible, val_keep)
        tail = torch.cat([ent["view_off"], _voxel_extent(vox)])
        host = ops.readback(tail)                                         # sync #2 (entries per view + extent)
        views = []
        for i in range(V):
            o, n_v = host[i], host[i + 1] - host[i]
            if n_v == 0 or n_v < cfg.min_visible or n_v > val_keep:
                continue
            views.append(ViewLists(ent["pt"][o:o + n_v], ent["x"][o:o + n_v], ent["y"][o:o + n_v], i))
        ent["total"] = host[V]
        ent["max_nv"] = max([host[i + 1] - host[i] for i in range(V)] + [0])
        ent["sum_nv2"] = float(sum((host[i + 1] - host[i]) ** 2 for i in range(V)))
        ent["num_views"] = V
        if mark is not None:
            mark("project+lists")
        return SceneBatch(coords64.float(), vox["coords_aug"].float(), vox["inds_reconstruct"], labels, gauss, views,
                          vox["order"], vox["seg_start"], host[V + 1:V + 4], ent=ent)
    depth_dev = scene.depth_dev if hasattr(scene, "depth_dev") else [torch.from_numpy(v.depth).to(dev) for v in scene.views]
    pt = torch.empty((V, N), dtype=torch.int64, device=dev)
    xs = torch.empty((V, N), dtype=torch.int64, device=dev)
    ys = torch.empty((V, N), dtype=torch.int64, device=dev)
    counts = torch.zeros(V + 3, dtype=torch.int64, device=dev)
    ws = None
    for i, (w2c, K) in enumerate(_view_matrices(cfg, scene.views)):
        mapping = ops.project_points(coords64, w2c, K[0, 0], K[1, 1], K[0, 2], K[1, 2], depth_dev[i], W, H,
                                     cfg.cut_bound, cfg.vis_thres)
        if ws is None:
            ws = torch.empty(ops._lib.load().gp_visible_lists_workspace_bytes(N), dtype=torch.uint8, device=dev)
        ops.visible_lists(mapping, pt[i], xs[i], ys[i], counts[i:i + 1], ws)
    counts[V:V + 3] = _voxel_extent(vox)
    host = counts.cpu().tolist()                                          # sync #2 (n_v per view + extent)
    views = []
    for i in range(V):
        n_v = host[i]
        if n_v == 0 or n_v < cfg.min_visible or n_v > val_keep:
            continue
        views.append(ViewLists(pt[i, :n_v], xs[i, :n_v], ys[i, :n_v], i))
    return SceneBatch(coords64.float(), vox["coords_aug"].float(), vox["inds_reconstruct"], labels, gauss, views,
                      vox["order"], vox["seg_start"], host[V:V + 3])


def upload_scene(scene, device="cuda"):
    """Make the raw inputs resident in HBM (outside any timed region)."""
    dev = torch.device(device)
    scene.coords_dev = torch.from_numpy(scene.coords).to(dev)
    shapes = {v.depth.shape for v in scene.views}
    if len(shapes) == 1:                                           # one stacked array: what the all-views loader kernels read
        scene.depth_all_dev = torch.from_numpy(np.stack([v.depth for v in scene.views])).to(dev)
        scene.depth_dev = list(scene.depth_all_dev.unbind(0))
    else:
        scene.depth_all_dev = None
        scene.depth_dev = [torch.from_numpy(v.depth).to(dev) for v in scene.views]
    scene.gauss_dev = torch.from_numpy(np.concatenate([scene.colors, scene.normals], 1).astype(np.float32)).to(dev)
    scene.labels_dev = torch.from_numpy(scene.labels).to(dev)
    return scene


# --------------------------------------------------------------------------------------------------
class SyntheticVLM:
    """Stand-in for the frozen X-Decoder (out of scope, weights unavailable offline): returns the five
    outputs the lift consumes (SURVEY.md 8a'), pre-generated per view and resident on the device."""

    def __init__(self, outputs, device="cuda", index_from_image=False):
        dev = torch.device(device)
        self.index_from_image = index_from_image       # the generator's view index is read from pixel (0,0) of the image
        self.ignores_image = not index_from_image      # outputs precomputed for every view: the lift may take them stacked
        self.pred_masks = torch.as_tensor(outputs["pred_masks"]).to(dev)
        self.pred_logits = torch.as_tensor(outputs["pred_logits"]).to(dev)
        self.mask_embed = torch.as_tensor(outputs["mask_embed"]).to(dev)
        self.text_embed = torch.as_tensor(outputs["text_embed"]).to(dev)
        self.logit_scale = float(outputs["logit_scale"])

    def __call__(self, view_index, image=None):
        if self.index_from_image and image is not None:
            view_index = int(image[0, 0, 0].item())
        return {"pred_masks": self.pred_masks[view_index], "pred_logits": self.pred_logits[view_index],
                "mask_embed": self.mask_embed[view_index], "text_embed": self.text_embed,
                "logit_scale": self.logit_scale}


class DenseFeatureVLM:
    """P config: dense per-pixel feature maps [V,D,H,W] (LSeg-style lift, affinity_module.py:416-449)."""

    def __init__(self, feat_maps, text_embed, logit_scale, device="cuda"):
        dev = torch.device(device)
        self.feat = torch.as_tensor(feat_maps).to(dev)
        self.text_embed = torch.as_tensor(text_embed).to(dev)
        self.logit_scale = float(logit_scale)


class LSegFeatureVLM:
    """LSeg-style dense features at the network's own resolution: feat_lo [V,D,h,w] (the reference runs LSeg on a
    320x240 resize and interpolates the output back to the image size, affinity_module.py:384-414)."""

    def __init__(self, feat_lo, image_shape, text_embed, logit_scale, device="cuda"):
        dev = torch.device(device)
        self.feat_lo = torch.as_tensor(feat_lo).to(dev).contiguous()
        self.image_shape = tuple(int(v) for v in image_shape)          # (H, W) of the images the pixels index
        self.text_embed = torch.as_tensor(text_embed).to(dev)
        self.logit_scale = float(logit_scale)


# --------------------------------------------------------------------------------------------------
class HotPath:
    """evaluate_scene on the device.  K, sharpen and num_iters are the reference's hard-coded
    constants (affinity_module.py:1492-1493,1584-1587) exposed as options."""

    def __init__(self, student: StudentWeights, mask_shape, K=96, sharpen=20.0, num_iters=19, device="cuda",
                 pool_mode="auto", pool_tile_rows=8, pool_block_rows=64, batch_views=True, pool_structure_ahead=True, affinity_mfma=True,
                 all_views_max_pairs=2e11, pool_row_order="rcb"):
        self.student = student
        self.batch_views = batch_views                 # lift all views of a scene in one set of launches when the inputs allow it
        # the all-views in-view fill is a brute-force search per view: sum over views of (queries x references) <= sum n_v^2 / 4
        # pair tests in fp64.  The limit is that COST (round 3 had a fixed 131 072 visible points per view): config M
        # (80 views of 30-60k visible points) is 4e10 pair tests and takes the lift from 23.6 to 14.4 ms against the view-by-view
        # grid search; beyond 2e11 (a scene with views of several hundred thousand visible points) the view-by-view path is taken.
        self.all_views_max_pairs = float(all_views_max_pairs)
        self.pool_mode, self.pool_tile_rows, self.pool_block_rows = pool_mode, pool_tile_rows, pool_block_rows
        # `prepare` also builds the pooling operator's structure (gp_pool_cs_structure) so that the affinity kernel writes the
        # weights in fragment order and no fill pass sits between the student and the pooling (pool_structure_ahead=False: the
        # two-pass form of rounds 3-4; last A/B 25.61 = 25.61 ms per scene: the pass left the critical path, the work stayed)
        self.pool_structure_ahead = bool(pool_structure_ahead)
        # rows 11 + operator fill as one matrix-core kernel (affinity_mfma=False: affinity_block_kernel + dst table; last A/B 26.08 -> 25.72 ms)
        self.affinity_mfma = bool(affinity_mfma)
        # row order of the POOLING operator (round 6): "rcb" = recursive coordinate bisection inside 1024-row Morton chunks into the
        # operator's 128-row blocks (ops.rcb_order: block unions 4.75 -> 4.2 rows per row, a launch -4 %; only where the matrix-core
        # affinity kernel fills the operator -- it reads the embedding planes by pooling row), "morton" = the voxel order itself
        self.pool_row_order = pool_row_order
        self.mask_shape = tuple(mask_shape)
        self.K, self.sharpen, self.num_iters = K, sharpen, num_iters
        self.device = torch.device(device)
        self._taps = {}
        self.stats = {}
        self.stage_mark = None                          # optional callable(name): bench.py's per-stage HIP events (side pass only)
        self._chain_ops = []                              # operators of chained pooling launches whose abort word is still unread
        self.keep_lifted, self.last_lifted = False, None  # parity tests: evaluate_scene keeps the lifted features [N, D] it refined

    def _tap_tables(self, h, w):
        key = (h, w)
        if key not in self._taps:
            H, W = self.mask_shape
            tx0, twx = aa_bicubic_taps(w, W)
            ty0, twy = aa_bicubic_taps(h, H)
            d = self.device
            self._taps[key] = tuple(torch.from_numpy(a).to(d) for a in (tx0, twx, ty0, twy))
        return self._taps[key]

    # ---- rows 6-7 -------------------------------------------------------------------------------
    def lift_masks(self, batch: SceneBatch, vlm):
        """lift_xdecoder_features (affinity_module.py:455-714).  `vlm(view_index, image=...)` -> dict(pred_masks [Q,h,w],
        pred_logits [Q,C+1], mask_embed [Q,D], text_embed [C,D], logit_scale): the synthetic stand-in, or any adapter of a
        real 2D model (affinity_module.ForwardSegAllVLM)."""
        dev = self.device
        N = batch.scene_coords.shape[0]
        V = len(batch.views)
        # per-(view, segment) tables for ALL source views in one launch when the VLM outputs are batched and indexed by
        # src_view (the synthetic stand-in on device-built batches); a batch that carries images (the reference's tuple,
        # slot 11) runs the VLM on imgs[view_idx] view by view, as affinity_module.py:496,518-519 does
        has_img = batch.imgs is not None and batch.imgs.numel() > 0
        batched = ((not has_img or getattr(vlm, "ignores_image", False)) and torch.is_tensor(getattr(vlm, "mask_embed", None))
                   and vlm.mask_embed.dim() == 3 and vlm.mask_embed.is_contiguous())
        outs = None
        if batched:
            text_embed, logit_scale = vlm.text_embed, float(vlm.logit_scale)
            Q, D = vlm.mask_embed.shape[1:]
        else:
            outs = [vlm(v.src_view, image=batch.imgs[v.src_view]) if has_img else vlm(v.src_view) for v in batch.views]
            if V:
                text_embed, logit_scale = outs[0]["text_embed"].to(dev), float(outs[0]["logit_scale"])
                Q, D = outs[0]["mask_embed"].shape
            else:
                text_embed, logit_scale = vlm.text_embed, float(vlm.logit_scale)
                Q, D = 1, text_embed.shape[1]
        text_norm = torch.nn.functional.normalize(text_embed, dim=-1).contiguous()
        C = text_norm.shape[0]
        n_tab = vlm.mask_embed.shape[0] if batched else max(V, 1)
        f_seg = torch.empty((n_tab, Q, D), dtype=torch.float32, device=dev)
        l_seg = torch.empty((n_tab, Q, C), dtype=torch.float32, device=dev)
        if batched:
            ops.segment_tables(vlm.mask_embed.view(n_tab * Q, D), text_norm, logit_scale,
                               f_seg.view(n_tab * Q, D), l_seg.view(n_tab * Q, C))
        # segment scores (:544) of all source views in one softmax + max when the logits are stacked (rows are independent)
        scores_all = None
        if batched and getattr(vlm, "pred_logits", None) is not None and vlm.pred_logits.dim() == 3:
            scores_all = torch.softmax(vlm.pred_logits, dim=-1)[..., :-1].max(-1).values.contiguous()
        ent = batch.ent
        pm_all = getattr(vlm, "pred_masks", None)
        all_views = (self.batch_views and batched and scores_all is not None and ent is not None and ent["total"] > 0
                     and ent["num_views"] <= 128 and ent.get("sum_nv2", float(ent["max_nv"]) ** 2 * ent["num_views"]) / 4 <= self.all_views_max_pairs
                     and torch.is_tensor(pm_all) and pm_all.dim() == 4 and pm_all.is_contiguous()
                     and pm_all.shape[0] >= ent["num_views"] and pm_all.shape[1] <= 1024)        # (Q <= 1024: the score sort's LDS table; else view by view)
        if all_views:
            # every view of the scene in one set of launches: segments, in-view fill and the point -> (view, segment) lists
            _, start, pvv, pvs = ops.lift_masks_views(pm_all, scores_all, self._tap_tables(pm_all.shape[2], pm_all.shape[3]),
                                                      self.mask_shape, batch.scene_coords, ent, ent["total"], ent["num_views"])
            return self._fuse_and_fill(batch, start, pvv, pvs, f_seg, l_seg, D, text_norm, logit_scale)
        cnt = torch.zeros(N + 1, dtype=torch.int64, device=dev)
        segs = []
        ws_m = ws_n = None
        for i, v in enumerate(batch.views):
            out = vlm(v.src_view) if batched else outs[i]
            pm = out["pred_masks"].to(dev).contiguous()
            scores = scores_all[v.src_view] if scores_all is not None else \
                torch.softmax(out["pred_logits"].to(dev), dim=-1)[..., :-1].max(-1).values.contiguous()
            if ws_m is None:
                lib = ops._lib.load()
                ws_m = torch.empty(lib.gp_lift_masks_workspace_bytes(*pm.shape), dtype=torch.uint8, device=dev)
                ws_n = torch.empty(lib.gp_nn1_masked_workspace_bytes(N), dtype=torch.uint8, device=dev)
            seg = ops.lift_masks_view(pm, scores, self._tap_tables(pm.shape[1], pm.shape[2]), self.mask_shape,
                                      v.x, v.y, workspace=ws_m)
            # in-view fill of uncovered points from the nearest covered point of the same view (:604-625)
            xyz = batch.scene_coords[v.pt].contiguous()
            covered = (seg >= 0).to(torch.uint8)
            nn = ops.nn1_masked(xyz, covered, 1 - covered, workspace=ws_n)
            seg = torch.where(nn >= 0, seg[nn.clamp(min=0)], seg)
            if not batched:
                ops.segment_tables(out["mask_embed"].to(dev).contiguous(), text_norm, logit_scale, f_seg[i], l_seg[i])
            ops.pv_count(v.pt, cnt)
            segs.append(seg)
        start = ops.exclusive_scan_i64(cnt)
        total = sum(len(v.pt) for v in batch.views)
        cursor = torch.zeros(N, dtype=torch.int32, device=dev)
        pvv = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
        pvs = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
        for i, v in enumerate(batch.views):
            ops.pv_fill(v.pt, segs[i], v.src_view if batched else i, start, cursor, pvv, pvs)
        return self._fuse_and_fill(batch, start, pvv, pvs, f_seg, l_seg, D, text_norm if V else text_embed, logit_scale)

    def _fuse_and_fill(self, batch, start, pvv, pvs, f_seg, l_seg, D, text, logit_scale):
        """Consensus fusion over a point's (view, segment) list (:647-686) and the scene-level fill of never-seen points."""
        dev = self.device
        N = batch.scene_coords.shape[0]
        F = torch.empty((N, D), dtype=torch.float32, device=dev)
        seen = ops.fuse_views_top3(start, pvv, pvs, N, f_seg, l_seg, F)
        # scene-level fill of never-seen points (:687-696)
        nn = ops.nn1_masked(batch.scene_coords, seen, 1 - seen)
        src = torch.where(nn >= 0, nn, torch.arange(N, device=dev))
        F = ops.gather_rows(F, D, src)
        # the reference returns the NORMALISED text embeddings (affinity_module.py:628 rebinds the name returned at :711)
        return F, text, logit_scale

    # ---- row 5 ----------------------------------------------------------------------------------
    def lift_dense(self, batch: SceneBatch, vlm: DenseFeatureVLM):
        dev = self.device
        N = batch.scene_coords.shape[0]
        D = vlm.feat.shape[1]
        s = torch.zeros((N, D), dtype=torch.float32, device=dev)
        cnt = torch.zeros(N, dtype=torch.float32, device=dev)
        for v in batch.views:
            ops.lift_dense_accum(vlm.feat[v.src_view], v.pt, v.x, v.y, s, cnt)
        seen = ops.lift_dense_finish(s, D, cnt)
        nn = ops.nn1_masked(batch.scene_coords, seen, 1 - seen)
        src = torch.where(nn >= 0, nn, torch.arange(N, device=dev))
        return ops.gather_rows(s, D, src), vlm.text_embed, vlm.logit_scale

    def lift_lseg(self, batch: SceneBatch, vlm: LSegFeatureVLM):
        """lift_lseg_features (affinity_module.py:348-452): bilinear(align_corners=True) resize of each view's
        low-resolution map, sampled at the visible pixels, mean over views, unseen points <- nearest seen point."""
        dev = self.device
        N = batch.scene_coords.shape[0]
        D = vlm.feat_lo.shape[1]
        H, W = vlm.image_shape
        s = torch.zeros((N, D), dtype=torch.float32, device=dev)
        cnt = torch.zeros(N, dtype=torch.float32, device=dev)
        for v in batch.views:
            ops.lift_dense_bilinear_accum(vlm.feat_lo[v.src_view], H, W, v.pt, v.x, v.y, s, cnt)
        seen = ops.lift_dense_finish(s, D, cnt)
        nn = ops.nn1_masked(batch.scene_coords, seen, 1 - seen)
        src = torch.where(nn >= 0, nn, torch.arange(N, device=dev))
        return ops.gather_rows(s, D, src), vlm.text_embed, vlm.logit_scale

    # ---- rows 8-12 ------------------------------------------------------------------------------
    def prepare(self, batch: SceneBatch, F):
        """Everything of `refine` that does not depend on the student's output: the internal row order, the voxel means (row 8),
        the lattice grid, the 27-offset kernel map and its compacted pairs, the exact kNN lists (row 10), and -- on the default
        pooling path -- the first half of the pooling operator (which needs the lists only) and the split planes of the features.
        Three of the scene's host synchronisations live here (pair count, padded union rows, extent when the batch has none).
        A scheduler may run it ahead on another stream (bench.py: beside the previous scene's convolutions, with the lift)."""
        dev = self.device
        N, D = F.shape
        st = self.student
        assert st.cin == D + GEO_DIM, f"student expects {st.cin} input channels, lift gives {D}+{GEO_DIM}"
        coords = batch.scene_coords_3d.floor().to(torch.int32).contiguous()      # batched_coordinates floors (:1543)
        Nv = coords.shape[0]
        perm, rank = ops.morton_order(coords)
        cs = coords[perm.long()].contiguous()
        mark = self.stage_mark if self.stage_mark is not None else (lambda name: None)
        mark("morton order")
        if batch.order is None:                                               # generic tuple input: CSR from the index
            order = torch.sort(batch.scene_inds_reconstruct, stable=True).indices
            segc = torch.zeros(Nv + 1, dtype=torch.int64, device=dev)
            segc[1:] = torch.bincount(batch.scene_inds_reconstruct, minlength=Nv).cumsum(0)
            batch.order, batch.seg_start = order, segc
        X = torch.zeros((Nv, st.cin_pad), dtype=torch.float32, device=dev)
        ops.scatter_mean_csr(F, D, batch.order, batch.seg_start, Nv, X, col0=0, row_map=rank)
        ops.scatter_mean_csr(batch.scene_gauss_features, GEO_DIM, batch.order, batch.seg_start, Nv, X, col0=D,
                             row_map=rank)
        mark("scatter_mean")
        if batch.extent is not None:
            grid = ops.grid_build(cs, [0, 0, 0], batch.extent)
        else:
            grid = ops.grid_build(cs)
        nbr_map = ops.kernel_map_build(grid, cs)
        pairs = ops.conv_pairs_build(nbr_map, col_tiles=max(1, st.hidden // 256)) if any(l[0] == "f16x3" for l in st.layers) else None
        mark("grid+kernel_map")
        nbr = ops.knn_lattice(grid, cs, perm, self.K)
        mark("kNN")
        state = {"X": X, "rank": rank, "nbr_map": nbr_map, "pairs": pairs, "nbr": nbr, "Nv": Nv, "D": D, "pool": None,
                 "xs": st.split_input(X)}                   # the first layer's pre-split operand (per-row scales)
        mode = self._pool_mode(D)
        if mode in ("mfma_cs", "mfma_engine", "mfma_chain"):
            sc = ops.pow2_scale(X, D)
            # "valid": the structure for the matrix-core affinity kernel (gp_affinity_cs_fragments: needs 128-wide embeddings and
            # K <= 96); True: the dst table of rounds 3-4 (affinity_block_kernel scatters its weights); False: the two-pass build
            how = self.pool_structure_ahead
            if how and self.affinity_mfma and self.K <= 96 and self.student is not None and self.student.embed == 128:
                how = "valid"
            rho = None
            nbr_op = nbr
            if how == "valid" and self.pool_row_order == "rcb" and Nv > 1024:
                # the operator's own row order: compact 128-row blocks (smaller neighbour unions).  Everything that enters the pooling by row
                # is written through `rho` (the feature planes here, the embedding planes by the output layer), the lists are renumbered,
                # and the final voxel -> point gather composes rho with the Morton rank; nothing else sees the order
                sigma, rho = ops.rcb_order(cs, 1024, 128)
                nbr_op = ops.rows_renumber(nbr, sigma, rho)
                state["rank_pool"] = rho[rank.long()].contiguous()
            state["pool"] = {"op": ops.pool_cs_plan(nbr_op, structure=how), "sc": sc, "rho": rho, "nbr": nbr_op,
                             "x_split": ops.split_f16(X, D, scale=sc[0:1], dst_row=rho),
                             "pong": tuple(torch.empty((Nv, D), dtype=torch.float16, device=dev) for _ in range(2))}
            if mode == "mfma_chain" and (state["pool"]["op"].dst is not None or state["pool"]["op"].valid is not None):   # (the lists need bu_row only)
                ops.pool_cs_deps(state["pool"]["op"])
            mark("pool plan+split")
        return state

    def refine(self, batch: SceneBatch, F, after_student=None, prepared=None, classify_text=None):
        """evaluate_scene after the lift (affinity_module.py:1524-1589). F fp32 [N,D] -> [N,D].
        after_student: optional callable run on the host once the student's kernels are enqueued and before the affinity /
        pooling kernels are -- a scheduler's hook (bench.py enqueues the next scene's loader, lift and `prepare` on a second
        stream there, so that they run beside the matrix-core-bound convolutions and never beside the HBM-bound pooling).
        prepared: the result of `prepare(batch, F)` when the scheduler ran it ahead.
        classify_text: (text_features [C, D], logit_scale) when the caller will classify the returned features next
        (classify_and_count): the final voxel -> point gather then computes the predictions in the same pass
        (gp_gather_rows_classify) and classify_and_count(result with THESE features) picks them up -- same rows, same labels."""
        p = prepared if prepared is not None else self.prepare(batch, F)
        X, rank, nbr, Nv, D = p["X"], p["rank"], p["nbr"], p["Nv"], p["D"]
        mark = self.stage_mark if self.stage_mark is not None else (lambda name: None)
        # the operator's structure was built ahead: the affinity kernel writes its weights straight into fragment order
        op = p["pool"]["op"] if p["pool"] is not None else None
        mfma_aff = op is not None and op.valid is not None
        rho = p["pool"]["rho"] if (p["pool"] is not None and mfma_aff) else None
        E = self.student.forward(X, p["nbr_map"], p["pairs"], x_split=p.get("xs"), mark=self.stage_mark, planes=mfma_aff, plane_rows=rho)
        mark("embed head" if self.stage_mark is not None else "student")
        if after_student is not None:
            after_student()
        if mfma_aff:
            # rows 11 + the operator fill on the matrix cores: similarities of every non-empty fragment, the valid ones through the
            # row's softmax, fragments written whole (no [Nv, K] weight matrix exists on this path); E = the embeddings as planes x 2^10
            ops.affinity_cs_fragments(None, self.sharpen, op, planes=E)
            w = None
        else:
            w = ops.affinity_softmax(E, nbr, self.sharpen, into=op if op is not None and op.dst is not None else None)
        mark("affinity")
        self._last_E, self._last_E_rho = E, rho
        self._last_pool_inputs = (X, nbr, w, Nv, D)       # kept for bench.py's isolated timing of row 12 ...
        self._last_pool_plan = p["pool"]                  # ... which re-applies THIS scene's operator (its row order included)
        out = self._pool(X, nbr, w, Nv, D, plan=p["pool"])
        mark("pooling")
        self._fused_pred = None
        if classify_text is not None and ops.can_gather_rows_classify(D, classify_text[0].shape[0]) and classify_text[0].shape[0] <= 32:
            text_norm = torch.nn.functional.normalize(classify_text[0], dim=-1).contiguous()
            out, pred, zero = ops.gather_rows_classify(out, D, batch.scene_inds_reconstruct, text_norm, classify_text[1],
                                                       row_map=p.get("rank_pool", rank))
            self._fused_pred = (out, classify_text[0], pred, zero)
        else:
            out = ops.gather_rows(out, D, batch.scene_inds_reconstruct, row_map=p.get("rank_pool", rank))
        mark("gather")
        self.stats = {"Nv": Nv, "nbr_map": p["nbr_map"], "pool_bytes_per_iter": Nv * (2 * D * 4 + self.K * 8),
                      "pool_kernel": self._pool_kernel}
        return out

    def _pool_mode(self, D):
        mode = self.pool_mode
        mfma_ok = D == 512 and self.pool_block_rows * self.K <= 16384 and self.num_iters >= 1
        cs_ok = D == 512 and 128 * self.K <= 12288 and self.num_iters >= 1
        tiles_ok = self.num_iters > 1 and self.pool_tile_rows * self.K <= 1536 and (D % 512 == 0 or D == 64)     # (D = 64: pool_tiles64_kernel, config P)
        if mode == "auto":
            mode = ("mfma_cs" if cs_ok else "mfma") if ((cs_ok or mfma_ok) and self.num_iters >= 3) else ("tiles" if tiles_ok else "ell")
        if mode in ("mfma", "mfma_persist") and not mfma_ok:
            raise ValueError(f"pool_mode='mfma' needs D == 512 and block_rows*K <= 16384 (D={D}, K={self.K})")
        if mode in ("mfma_cs", "mfma_engine", "mfma_chain") and not cs_ok:
            raise ValueError(f"pool_mode='{mode}' needs D == 512 and K <= 96 (D={D}, K={self.K})")
        return mode

    def _pool(self, X, nbr, w, Nv, D, plan=None):
        """Row 12: num_iters applications of the row-stochastic affinity operator (affinity_module.py:1575-1587).
        pool_mode: "auto" (matrix cores when the shape allows, else tiles, else ELL), "mfma_cs", "mfma_engine", "mfma",
        "mfma_persist", "tiles", "ell".  plan: what `prepare` built ahead for the default path (operator plan, split planes)."""
        dev = X.device
        mode = self._pool_mode(D)
        R = self.pool_tile_rows
        tiles_ok = self.num_iters > 1 and R * self.K <= 1536 and (D % 512 == 0 or D == 64)
        if w is None and (plan is None or not plan["op"].filled):
            # (isolated calls after a scene whose weights went straight into fragments: the [Nv, K] weights from the embedding planes)
            E = self._last_E
            if isinstance(E, tuple):
                E = (E[0].float() + E[1].float()) / ops.AFFINITY_PLANE_SCALE
                if getattr(self, "_last_E_rho", None) is not None:        # the planes were written in the operator's row order: back to voxel rows
                    E = E[self._last_E_rho.long()]
            w = ops.affinity_softmax(E.contiguous(), nbr, self.sharpen)
        if self.num_iters == 0:
            out = torch.empty((Nv, D), dtype=torch.float32, device=dev)
            out.copy_(X[:, :D])
            self._pool_kernel = "none"
            return out
        if mode in ("mfma_cs", "mfma_engine", "mfma_chain"):
            # column-sliced matrix-core pooling (default): 128-row blocks x 256-column halves, union rows grouped by the 16-row
            # groups that use them, empty weight fragments skipped (pool_mfma_cs.hip).  "mfma_engine": the producer / consumer
            # form of the same operator (persistent, 128-column tiles) -- same bits, same speed on MI355X.
            if plan is None:                            # (isolated calls: bench.py's pooling-only passes, tests)
                sc = ops.pow2_scale(X, D)
                plan = {"op": ops.pool_cs_plan(nbr), "sc": sc, "x_split": ops.split_f16(X, D, scale=sc[0:1]),
                        "pong": tuple(torch.empty((Nv, D), dtype=torch.float16, device=dev) for _ in range(2))}
            op = plan["op"] if plan["op"].filled else ops.pool_cs_fill(plan["op"], nbr, w)
            out = torch.empty((Nv, D), dtype=torch.float32, device=dev)
            sc = plan["sc"]
            sp = [plan["x_split"], plan["pong"]]
            if self.stage_mark is not None:
                self.stage_mark("pool operator fill")
            if mode == "mfma_chain" and self.num_iters >= 2:
                # all applications in ONE launch (gp_pool_cs_apply_chain: per-block flags instead of kernel boundaries; same planes,
                # same bits).  The abort word of its contract is read at the next host synchronisation (pool_chain_check).
                if op.dep is None:
                    ops.pool_cs_deps(op)
                ops.pool_cs_apply_chain(sp[0], sp[1], op, D, self.num_iters, out, out_scale=sc[1:2])
                if op not in self._chain_ops:            # (an operator re-applied by isolated timing passes is pending once)
                    self._chain_ops.append(op)
                self._pool_kernel = "cs_chain_kernel"
                return out
            src = sp[0]
            for t in range(self.num_iters):
                last = t == self.num_iters - 1
                dst = None if last else sp[(t + 1) % 2]
                ops.pool_cs_apply(src, op, D, out_split=dst, out_f32=out if last else None, out_scale=sc[1:2] if last else None,
                                  engine=mode == "mfma_engine")
                src = dst
            self._pool_kernel = "cs_engine_kernel" if mode == "mfma_engine" else "cs_pool_kernel"
            return out
        if mode in ("mfma", "mfma_persist"):
            # matrix-core pooling: operands stay split (hi, lo) f16 between applications, fp32 only at the end.
            # "mfma" (default): one (64 rows x 128 columns) tile per workgroup, two workgroups per CU.
            # "mfma_persist": the persistent kernel (one workgroup per CU, 256 columns per workgroup, weight fragments shared
            # by both column groups, rings kept full across row blocks) -- same speed within run-to-run noise on MI355X
            # (DESIGN.md section 6); needs >= 4 steps per row block and 64-row blocks, else falls back to "mfma"
            op = ops.pool_mfma_build(nbr, w, self.pool_block_rows, min_steps=9 if mode == "mfma_persist" else 0)
            persistent = mode == "mfma_persist" and op.min_steps >= 9 and self.pool_block_rows == 64
            rows = op.rows_padded if persistent else Nv
            out = torch.empty((rows, D), dtype=torch.float32, device=dev)
            # one power of two for the whole operand (pooling is a convex combination: magnitudes never grow), so that the lo
            # halves of small features stay normal f16 numbers; the fp32 output of the last application is scaled back
            sc = ops.pow2_scale(X, D)
            sp = [ops.split_f16(X, D, scale=sc[0:1]), tuple(torch.empty((rows, D), dtype=torch.float16, device=dev) for _ in range(2))]
            if persistent and self.num_iters > 1:          # the ping-pong partner of the input planes also needs whole row blocks
                sp.append(tuple(torch.empty((rows, D), dtype=torch.float16, device=dev) for _ in range(2)))
            apply = ops.pool_mfma_apply_persistent if persistent else ops.pool_mfma_apply
            src = sp[0]
            for t in range(self.num_iters):
                last = t == self.num_iters - 1
                dst = None if last else sp[1 + (t % 2)] if persistent else sp[(t + 1) % 2]
                apply(src, op, D, out_split=dst, out_f32=out if last else None, out_scale=sc[1:2] if last else None)
                src = dst
            self._pool_kernel = "pool_mfma_persist_kernel" if persistent else "pool_mfma_kernel"
            return out[:Nv]
        use_tiles = mode == "tiles" and tiles_ok
        tiles = ops.pool_tiles_build(nbr, w, R) if use_tiles else None
        bufs = [torch.empty((Nv, D), dtype=torch.float32, device=dev) for _ in range(2)]
        cur = X
        for t in range(self.num_iters):
            if use_tiles:
                ops.pool_tiles_apply(cur, tiles, D, bufs[t % 2])
            else:
                ops.pool_ell(cur, nbr, w, D, bufs[t % 2])
            cur = bufs[t % 2]
        self._pool_kernel = ("pool_tiles64_kernel" if D == 64 else "pool_tiles_kernel") if use_tiles else "pool_ell_kernel"
        return cur

    def evaluate_scene(self, batch: SceneBatch, vlm):
        if isinstance(vlm, LSegFeatureVLM):
            F, text, scale = self.lift_lseg(batch, vlm)
        elif isinstance(vlm, DenseFeatureVLM):
            F, text, scale = self.lift_dense(batch, vlm)
        else:
            F, text, scale = self.lift_masks(batch, vlm)
        if self.keep_lifted:                 # (parity tests compare the lift stage too; off by default: the tensor is 300 MB at S)
            self.last_lifted = F
        feats = self.refine(batch, F)
        if self._chain_ops:
            self.pool_chain_check()                        # (chained pooling only: never hand out features of a launch that gave up)
        return {"scene_features": feats, "text_features": text, "logit_scale": scale}

    def pool_chain_check(self, block=True):
        """Host side of gp_pool_cs_apply_chain's contract: raise if a chained pooling launch gave up.  Waits for each pending operator's
        launch (the event recorded on the launching stream) before it reads the abort word, and forgets an operator only after the read.
        block=False reads only the operators whose launch has already finished (no stall: classify_and_count of the next scenes);
        the blocking form runs where the consumer synchronises anyway: HotPath.evaluate_scene / the trainer's evaluate_scene before they
        return features of a chained launch, the end of bench.py's timed region, the tests."""
        keep = []
        while self._chain_ops:
            op = self._chain_ops[0]
            ev = getattr(op, "chain_done", None)
            if not block and ev is not None and not ev.query():        # still running: a later call (or the blocking one) reads it
                keep.append(self._chain_ops.pop(0))
                continue
            ops.pool_cs_chain_check(op)
            self._chain_ops.pop(0)
        self._chain_ops = keep

    # ---- row 13 + caller tail --------------------------------------------------------------------
    def classify_and_count(self, result, labels, num_classes, ignore_ids, counts):
        feats = result["scene_features"]
        fp = getattr(self, "_fused_pred", None)
        if fp is not None and fp[0] is feats and fp[1] is result["text_features"]:
            pred, zero = fp[2], fp[3]                     # computed by refine's final gather (classify_text=...): the same labels
            self._fused_pred = None
            ops.iou_hist(pred, labels, num_classes, ignore_ids, counts)
            if self._chain_ops:
                self.pool_chain_check(block=False)
            return pred, zero
        text_norm = torch.nn.functional.normalize(result["text_features"], dim=-1).contiguous()
        if text_norm.shape[0] > 32 and feats.shape[1] % 32 == 0:
            pred, zero = ops.classify_argmax_gemm(feats, text_norm)
        else:
            pred, zero = ops.classify_argmax(feats, text_norm, result["logit_scale"])
        ops.iou_hist(pred, labels, num_classes, ignore_ids, counts)
        if self._chain_ops:
            self.pool_chain_check(block=False)             # (chained pooling only: abort words of launches that have finished)
        return pred, zero


# --------------------------------------------------------------------------------------------------
def random_student_state_dict(input_dim, hidden=512, embed=128, num_blocks=4, seed=0):
    """Random-init weights of the AffinityPredictor architecture in the ME state_dict layout
    (He-normal kernels; BN statistics near identity).  No checkpoint is available offline."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def kern(ci, co, kv=27):
        std = (2.0 / (kv * ci)) ** 0.5
        return torch.randn((kv, ci, co) if kv > 1 else (ci, co), generator=g) * std

    def bn(prefix, c):
        sd[prefix + ".bn.weight"] = 1.0 + 0.1 * torch.randn(c, generator=g)
        sd[prefix + ".bn.bias"] = 0.1 * torch.randn(c, generator=g)
        sd[prefix + ".bn.running_mean"] = 0.1 * torch.randn(c, generator=g)
        sd[prefix + ".bn.running_var"] = 1.0 + 0.2 * torch.rand(c, generator=g)
        sd[prefix + ".bn.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    sd["input_layer.0.kernel"] = kern(input_dim, hidden)
    bn("input_layer.1", hidden)
    for i in range(num_blocks):
        sd[f"res_blocks.{i}.conv1.kernel"] = kern(hidden, hidden)
        bn(f"res_blocks.{i}.norm1", hidden)
        sd[f"res_blocks.{i}.conv2.kernel"] = kern(hidden, hidden)
        bn(f"res_blocks.{i}.norm2", hidden)
    sd["output_layer.kernel"] = kern(hidden, embed, kv=1)
    return sd


def scene_rigid_transform(voxel_size, seed):
    """Host mirror of Voxelizer.get_transformation_matrix (dataset/voxelizer.py:32-58) with the
    always-on augmentation of Point3DLoader (dataset/point_loader.py:54-60,100-107): consumes
    np.random in the reference's order.  Returns M_r @ M_v."""
    from .voxelizer import Voxelizer, default_voxelizer
    np.random.seed(seed)
    vox = default_voxelizer(voxel_size)
    M_v, M_r = vox.get_transformation_matrix()
    return M_r @ M_v
