"""Host mirror of models/affinity_module.py: AffinityPredictor + SonataXAffinityTrainer.

Same names, constructor arguments and call surface as the reference classes the drivers touch
(run/validation.py:166-219,408-411), with the numeric work on the HIP path
(geopurify_amd.pipeline.HotPath).  What is NOT here, on purpose (out of scope):
the frozen X-Decoder / LSeg / Sonata models.  The 2D VLM is injected as a callable that returns the
five X-Decoder outputs the lift consumes; `SyntheticVLM` is the offline stand-in.
"""
import torch
import torch.nn as nn

from . import pipeline
from .pipeline import GEO_DIM, HotPath, SceneBatch, StudentWeights, ViewLists


class _SparseConvParams(nn.Module):
    """Parameter holder with MinkowskiConvolution's state_dict layout: `kernel` [kv,cin,cout] (2-D for kv=1)."""

    def __init__(self, cin, cout, kernel_size):
        super().__init__()
        kv = kernel_size ** 3
        shape = (kv, cin, cout) if kv > 1 else (cin, cout)
        self.kernel = nn.Parameter(torch.randn(shape) * (2.0 / (kv * cin)) ** 0.5)


class _SparseBatchNorm(nn.Module):
    """MinkowskiBatchNorm wraps a BatchNorm1d as `.bn` (keys `<name>.bn.weight`, ...)."""

    def __init__(self, c):
        super().__init__()
        self.bn = nn.BatchNorm1d(c)


class _ReLU(nn.Module):
    pass


class MinkowskiResBlock(nn.Module):
    """models/affinity_module.py:33-49."""

    def __init__(self, channels):
        super().__init__()
        self.conv1 = _SparseConvParams(channels, channels, 3)
        self.norm1 = _SparseBatchNorm(channels)
        self.conv2 = _SparseConvParams(channels, channels, 3)
        self.norm2 = _SparseBatchNorm(channels)


class AffinityPredictor(nn.Module):
    """models/affinity_module.py:51-85.  Holds parameters in the reference's state_dict layout;
    the forward pass is executed by the HIP sparse-conv kernels on a (features, nbr_map) pair."""

    def __init__(self, input_dim, embed_dim=128, hidden_dim=256):
        super().__init__()
        self.embed_dim = embed_dim
        self.input_layer = nn.Sequential(_SparseConvParams(input_dim, hidden_dim, 3), _SparseBatchNorm(hidden_dim), _ReLU())
        self.res_blocks = nn.Sequential(*[MinkowskiResBlock(hidden_dim) for _ in range(4)])
        self.output_layer = _SparseConvParams(hidden_dim, embed_dim, 1)
        self._dev_weights = None
        self._dev_key = None

    def get_param_groups(self):
        return {"input": list(self.input_layer.parameters()),
                "middle": list(self.res_blocks.parameters()),
                "output": list(self.output_layer.parameters())}

    def load_state_dict(self, state_dict, strict=True):
        out = super().load_state_dict(state_dict, strict=strict)
        self._dev_weights = None
        return out

    def train(self, mode=True):
        if mode:
            self._dev_weights = None             # parameters are about to change: drop the folded / split device copy
        return super().train(mode)

    def _weights_key(self, device):
        """Identity + in-place version of every parameter and buffer: optimizer.step(), FusedAdamW, .to(), load_state_dict
        and manual edits all change it, so a stale folded copy can never be served."""
        return (str(device),) + tuple((id(t), t._version, t.device.type) for t in list(self.parameters()) + list(self.buffers()))

    def device_weights(self, device):
        key = self._weights_key(device)
        if self._dev_weights is None or self.training or self._dev_key != key:
            self._dev_weights = StudentWeights(self.state_dict(), device)
            self._dev_key = key
        return self._dev_weights

    def forward(self, features, nbr_map):
        """features fp32 [Nv, cin_pad] in Morton order + its 27-offset kernel map -> unnormalised is
        not exposed: returns the L2-normalised embeddings the caller applies next (:1546-1547)."""
        return self.device_weights(features.device).forward(features, nbr_map)


_VLM_FACTORY = None


def register_vlm_factory(factory):
    """Install the builder of the 2D VLM that the reference constructs inside SonataXAffinityTrainer.__init__
    (models/affinity_module.py:224-249: build X-Decoder from `xdecoder_cfg`, load its weights, set the text prompts).
    `factory(cfg, xdecoder_cfg, device, use_lseg)` -> a callable `vlm(view_index, image=img [H,W,3] 0..255)` returning
    dict(pred_masks [Q,h,w], pred_logits [Q,C+1], mask_embed [Q,D], text_embed [C,D], logit_scale), e.g. a
    ForwardSegAllVLM around the real model.  With a factory registered the reference's own constructor call
    `SonataXAffinityTrainer(args, xdecoder_cfg, scene_config, device, False)` (run/validation.py:166) works unchanged.
    Pass None to clear.  Returns the previous factory."""
    global _VLM_FACTORY
    old, _VLM_FACTORY = _VLM_FACTORY, factory
    return old


class ForwardSegAllVLM:
    """Adapter of an X-Decoder-style model to the lift's VLM hook: what the reference does with each view's image
    (models/affinity_module.py:496,518-519): img [H,W,3] -> [1,3,H,W] on the device ->
    `model.forward_seg_all([{'image': img, 'height': mask_shape[0], 'width': mask_shape[1]}])` -> (_, outputs)."""

    def __init__(self, model, mask_shape, device="cuda"):
        self.model, self.mask_shape, self.device = model, tuple(mask_shape), device

    def __call__(self, view_index, image=None):
        if image is None:
            raise ValueError("ForwardSegAllVLM needs the view's image (slot 11 of the batch tuple)")
        img = image.unsqueeze(0).permute(0, 3, 1, 2).contiguous().to(self.device)
        _, out = self.model.forward_seg_all([{"image": img, "height": self.mask_shape[0], "width": self.mask_shape[1]}])
        return {"pred_masks": out["pred_masks"][0], "pred_logits": out["pred_logits"][0], "mask_embed": out["mask_embed"][0],
                "text_embed": out["text_embed"], "logit_scale": out["logit_scale"]}


class SonataXAffinityTrainer(nn.Module):
    """models/affinity_module.py:129-1607, inference surface.

    cfg needs: all_label, mask_shape, voxel_size (util/config flat namespace).  `xdecoder_cfg` and
    `scene_config` are accepted for signature compatibility.  `vlm`: object with __call__(view_index)
    -> dict(pred_masks, pred_logits, mask_embed, text_embed, logit_scale) and attributes
    text_embed / logit_scale (e.g. pipeline.SyntheticVLM), or a pipeline.DenseFeatureVLM when
    use_lseg=True (dense-feature lift)."""

    def __init__(self, cfg, xdecoder_cfg=None, scene_config=None, device="cuda", use_lseg=True, vlm=None,
                 feature_dim=512, embed_dim=128, hidden_dim=512, teacher=None, allow_deferred_vlm=False):
        super().__init__()
        self.cfg = cfg
        self.device = device
        self.use_lseg = use_lseg
        self.scene_config = scene_config
        self.use_ape = bool(getattr(cfg, "use_ape", False)) if not isinstance(cfg, dict) else bool(cfg.get("use_ape", False))
        if self.use_ape:
            raise NotImplementedError("lift_ape_features needs the absent xdecoder_test package (SURVEY.md section 2 #1)")
        if vlm is None and _VLM_FACTORY is not None:
            vlm = _VLM_FACTORY(cfg, xdecoder_cfg, device, use_lseg)          # the reference builds its VLM here (:224-249)
        if vlm is None and not allow_deferred_vlm:
            raise RuntimeError(
                "SonataXAffinityTrainer: no 2D VLM.  The X-Decoder / LSeg model itself is outside this library: register a "
                "builder with geopurify_amd.affinity_module.register_vlm_factory(factory) (then the reference's constructor "
                "call works unchanged), pass vlm=..., or pass allow_deferred_vlm=True and set .vlm before the first scene")
        self.vlm = vlm
        self.feature_dim = feature_dim
        self.affinity_student = AffinityPredictor(input_dim=feature_dim + GEO_DIM, embed_dim=embed_dim,
                                                  hidden_dim=hidden_dim)
        self.K = 96                               # :1492
        self.affinity_sharpen_factor = 20         # :1493
        self.num_pool_iters = 19                  # :1584-1587 (1 + 18)
        self.teacher = teacher                    # stand-in for get_sonata_features (:995-1063)
        self.num_anchors_per_scene = 4096         # :277
        self.num_negatives_per_anchor = 63        # :278
        self.info_nce_temperature = 0.07          # :279

    def _hot_path(self):
        """The device pipeline of this trainer, kept across scenes (its resize tap tables, 15 ms of host work per mask shape,
        and the student's device weights are built once) and rebuilt when the weights or the options change."""
        dev = torch.device(self.device if self.device != "cuda" else f"cuda:{torch.cuda.current_device()}")
        st = self.affinity_student.device_weights(dev)
        shape = tuple(self.cfg["mask_shape"] if isinstance(self.cfg, dict) else self.cfg.mask_shape)
        key = (id(st), shape, self.K, float(self.affinity_sharpen_factor), self.num_pool_iters, str(dev))
        if getattr(self, "_hp_key", None) != key:
            self._hp = HotPath(st, shape, K=self.K, sharpen=float(self.affinity_sharpen_factor), num_iters=self.num_pool_iters, device=dev)
            self._hp_key = key
        return self._hp

    @staticmethod
    def _batch_from_tuple(batch_data, device):
        """Parse the reference's positional 20-tuple (names at affinity_module.py:1501-1522)."""
        (scene_coords, scene_coords_3d, scene_inds_reconstruct, scene_label, ori_coords_3ds, _c, _f, _g, _l, _b,
         _l2, imgs, x_labels, y_labels, mask_2ds, _ir, _um, _mp, _cap, scene_gauss_features) = batch_data
        dev = torch.device(device)
        N = scene_coords.shape[0]
        nb = True                                                # (pinned host tensors copy asynchronously)
        mask_2ds = mask_2ds.to(dev, non_blocking=nb)
        V = int(mask_2ds.shape[0] // N)
        imgs = imgs if torch.is_tensor(imgs) and imgs.dim() == 4 and imgs.shape[0] == V else None   # slot 11: what the VLM sees
        x_labels, y_labels = x_labels.to(dev, non_blocking=nb), y_labels.to(dev, non_blocking=nb)
        # The collate concatenates the views in order, so the rows of view i are one contiguous range of x_labels / y_labels /
        # ori_coords_3ds and the visible flags of mask_2ds in view-major order name their points in the same order: ONE
        # nonzero + ONE bincount + ONE host read-back instead of three launches and a sync per view.
        flat = torch.nonzero(mask_2ds[:, 1]).squeeze(1)            # view * N + point, ascending
        view_of = ori_coords_3ds[:, 0].to(dev, non_blocking=nb).long()
        counts = torch.bincount(view_of, minlength=V)
        host = torch.cat([counts, torch.bincount(torch.div(flat, N, rounding_mode="floor"), minlength=V)]).cpu().tolist()
        if host[:V] != host[V:2 * V] or not bool((view_of[1:] >= view_of[:-1]).all()):
            raise ValueError("batch tuple: the rows of ori_coords_3ds / x_labels do not follow the visible flags of mask_2ds view by view")
        pts = flat - torch.div(flat, N, rounding_mode="floor") * N
        views, o = [], 0
        for i in range(V):
            n_i = host[i]
            views.append(ViewLists(pts[o:o + n_i], x_labels[o:o + n_i], y_labels[o:o + n_i], i))
            o += n_i
        # the same lists as ONE set of view-major arrays: what gp_views_visible_lists produces on device-built batches, so that
        # the lift takes its all-views kernels (gp_lift_masks_views) on the reference's tuple as well
        ent = None
        if V and o > 0:
            ent = {"pt": pts.contiguous(), "x": x_labels.long().contiguous(), "y": y_labels.long().contiguous(),
                   "view": view_of.to(torch.int32).contiguous(), "view_off": torch.cat([counts.new_zeros(1), counts.cumsum(0)]),
                   "keep": torch.ones(V, dtype=torch.uint8, device=dev), "total": o, "max_nv": max(host[:V]),
                   "sum_nv2": float(sum(h * h for h in host[:V])), "num_views": V}
        return SceneBatch(scene_coords.to(dev, non_blocking=nb).float().contiguous(), scene_coords_3d.to(dev, non_blocking=nb).float().contiguous(),
                          scene_inds_reconstruct.to(dev, non_blocking=nb).long().contiguous(), scene_label.to(dev, non_blocking=nb).long(),
                          scene_gauss_features[:, :6].to(dev, non_blocking=nb).float().contiguous(), views, imgs=imgs, ent=ent)

    def offer_next(self, batch_data, vlm=None, ready=None):
        """Look-ahead for callers that can see one scene ahead (a loader wrapper: geopurify_amd.validation.validate, bench.py's
        `api_tuple`): hand the NEXT scene's 20-tuple (or SceneBatch) here before calling evaluate_scene on the current one.
        evaluate_scene then enqueues the offered scene's copy / tuple parse / lift / `HotPath.prepare` on a second stream right
        after the current scene's student -- beside its matrix-core-bound convolutions, joined before its pooling -- and the
        offered scene's own evaluate_scene call (same object) starts at its student.  Results are those of the serial call.
        vlm: the 2D VLM of the offered scene when it differs from `self.vlm` (the synthetic stand-in is per scene).
        ready: an event behind which the offered tensors are valid (a device tuple still being copied on the loader's stream:
        geopurify_amd.data_loader.LookAheadLoader)."""
        self._offered = (batch_data, vlm, ready)

    def _lift(self, hp, batch, vlm):
        if isinstance(vlm, pipeline.LSegFeatureVLM):
            return hp.lift_lseg(batch, vlm)
        if isinstance(vlm, pipeline.DenseFeatureVLM):
            return hp.lift_dense(batch, vlm)
        return hp.lift_masks(batch, vlm)

    def _look_ahead(self, hp, batch_data, vlm, after, ready=None):
        """copy + parse + lift + prepare of an offered scene on the side stream, behind the event `after` (the start of the current
        scene's refine on the caller's stream: every kernel that used the blocks this look-ahead may be handed has run by then)."""
        if getattr(self, "_side", None) is None:            # (`side_stream`: a stream the program already owns -- see LookAheadLoader)
            self._side = getattr(self, "side_stream", None) or torch.cuda.Stream(device=hp.device)
        with torch.cuda.stream(self._side):
            self._side.wait_event(after)
            if ready is not None:
                self._side.wait_event(ready)
            batch = batch_data if isinstance(batch_data, SceneBatch) else self._batch_from_tuple(batch_data, hp.device)
            F, text, scale = self._lift(hp, batch, vlm)
            prep = hp.prepare(batch, F)
            done = torch.cuda.Event()
            done.record(self._side)
        return batch, F, text, scale, prep, done

    @torch.no_grad()
    def evaluate_scene(self, batch_data, vis_prefix="scene0695_00"):
        """-> {"scene_features" [N,D], "text_features" [C,D], "logit_scale"} (:1604-1607)."""
        self.affinity_student.eval()
        if self.vlm is None:
            raise RuntimeError("no 2D VLM attached: pass vlm=... (the X-Decoder itself is out of scope)")
        hp = self._hot_path()
        cur = torch.cuda.current_stream(hp.device)
        ahead = getattr(self, "_ahead", None)
        self._ahead = None
        if ahead is not None and ahead[0] is batch_data:             # this scene was offered and lifted ahead
            _, batch, F, text, scale, prep, done = ahead
            cur.wait_event(done)
        else:
            batch = batch_data if isinstance(batch_data, SceneBatch) else self._batch_from_tuple(batch_data, hp.device)
            F, text, scale = self._lift(hp, batch, self.vlm)
            prep = None
        if hp.keep_lifted:                                           # (parity tests compare the lift stage too)
            hp.last_lifted = F
        offered, self._offered = getattr(self, "_offered", None), None
        hook = None
        if offered is not None:
            started = torch.cuda.Event()
            started.record(cur)

            def hook():
                nxt, nvlm, ready = offered
                self._ahead = (nxt,) + self._look_ahead(hp, nxt, nvlm if nvlm is not None else self.vlm, started, ready)
                cur.wait_event(self._ahead[-1])                  # the pooling below has the chip to itself
        feats = hp.refine(batch, F, after_student=hook, prepared=prep)
        if hp._chain_ops:
            hp.pool_chain_check()                                    # (chained pooling only: its abort word, before the features leave)
        self.last_scene_done = torch.cuda.Event()                    # (a loader that copies ahead orders its copies behind this)
        self.last_scene_done.record(cur)
        return {"scene_features": feats, "text_features": text, "logit_scale": scale}

    def forward(self, batch_data):
        """Training forward (affinity_module.py:1138-1237): lift (no grad) -> teacher features -> contrastive sampling
        -> student on the sampled voxels (BatchNorm in training mode) -> InfoNCE.  Returns the loss; `.backward()`
        fills the student's gradients (HIP backward pass), so run/train.py's loop applies unchanged.
        The Sonata teacher is not available offline: pass `teacher=callable(batch) -> [N, Dt]` to the constructor."""
        from . import training
        if self.vlm is None or self.teacher is None:
            raise RuntimeError("training needs a 2D VLM (vlm=...) and a teacher (teacher=callable(batch) -> [N,Dt]); "
                               "X-Decoder and Sonata themselves are out of scope")
        dev = torch.device(self.device if self.device != "cuda" else f"cuda:{torch.cuda.current_device()}")
        hp = HotPath(None, tuple(self.cfg["mask_shape"] if isinstance(self.cfg, dict) else self.cfg.mask_shape), device=dev)   # lift only
        batch = batch_data if isinstance(batch_data, SceneBatch) else self._batch_from_tuple(batch_data, hp.device)
        with torch.no_grad():
            if isinstance(self.vlm, pipeline.LSegFeatureVLM):
                F_lift, _, _ = hp.lift_lseg(batch, self.vlm)
            elif isinstance(self.vlm, pipeline.DenseFeatureVLM):
                F_lift, _, _ = hp.lift_dense(batch, self.vlm)
            else:
                F_lift, _, _ = hp.lift_masks(batch, self.vlm)
            F_teacher = self.teacher(batch).to(F_lift.device).float().contiguous()
        return training.training_forward(self.affinity_student, F_lift, batch.scene_gauss_features, batch.scene_inds_reconstruct,
                                         batch.scene_coords_3d, batch.scene_coords.float().contiguous(), F_teacher,
                                         num_anchors=self.num_anchors_per_scene, num_negatives=self.num_negatives_per_anchor,
                                         temperature=self.info_nce_temperature, K=self.K)
