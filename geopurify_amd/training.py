"""Training step of the Student Affinity Network on the device.

Mirrors models/affinity_module.py:1099-1237 (sample_contrastive_pairs_hybrid + SonataXAffinityTrainer.forward) and
run/train.py:188-198,320-325,346-353 (AdamW with three parameter groups, LinearLR -> CosineAnnealingLR).

What runs where:
  * 3x3x3 convolutions, forward and data-gradient: gp_sparse_conv_f16x3 (dgrad = the same operator with weights
    V[k] = W[26-k]^T; gradients are scaled by a power of two before the f16 hi/lo split so that they stay normal);
  * BatchNorm(training), ReLU masks, InfoNCE forward+backward, AdamW, anchors' K nearest points: train.hip;
  * weight gradients dW[k] = X[in_k]^T dY[out_k]: gp_conv_wgrad_f16x3 (wgrad.hip); the anchors x points similarity:
    the convolution operator with one offset (_anchor_similarities); the 512->128 output layer is the convolution operator
    with one offset and the identity map (forward and data gradient: gp_sparse_conv, true-fp32 MFMA; weight gradient:
    gp_conv_wgrad_f16x3 over identity pairs with the gradient padded to 256 columns); shapes those kernels do not take
    (hidden width < 256, CPU tensors): torch.matmul;
  * set logic (unique / argmax / topk of the sampler): torch device ops, as in the reference.
The teacher (Sonata) is not available offline: its per-point features are an input tensor.
Deviation: the voxel input is [mean lifted feature | mean geometry] (518 channels) as in
evaluate_scene; the released forward feeds 512 channels into the 518-channel layer and cannot run.
"""
import math

import torch

from . import ops, sharding
from .pipeline import CONV_PAD, GEO_DIM, _pad_to

N_MACRO = 48                                   # affinity_module.py:1122
GROUP_LR = {"input": 0.1, "middle": 1.0, "output": 5.0}          # run/train.py:193-195
W_POW2 = 16.0                                  # weights (|w| << 1) are split as 16*w; undone by the scale vector


def param_group(name):
    return "input" if name.startswith("input_layer") else ("output" if name.startswith("output_layer") else "middle")


def lr_schedule(step_index, base_lr, group, warmup_iters, main_iters):
    """SequentialLR([LinearLR(1e-6 -> 1, warmup_iters), CosineAnnealingLR(T_max=main_iters, eta_min=base_lr*1e-3)])
    evaluated in closed form for optimizer step number `step_index` (0-based) (run/train.py:320-325)."""
    lr0 = base_lr * GROUP_LR[group]
    if step_index < warmup_iters:
        return lr0 * (1e-6 + (1.0 - 1e-6) * step_index / max(warmup_iters, 1))
    t = min(step_index - warmup_iters, main_iters)
    eta_min = base_lr * 1e-3
    return eta_min + (lr0 - eta_min) * 0.5 * (1.0 + math.cos(math.pi * t / max(main_iters, 1)))


def _anchor_similarities(F_teacher, anchor_indices):
    """sim[a, p] = <Fn[anchor_a], Fn[p]> with Fn = F.normalize(F_teacher) (affinity_module.py:1114-1115).  On the device with a feature
    width that is a multiple of 32 this is the gather-GEMM of the convolution operator with ONE offset: the anchors are the gathered
    rows, the points play the output channels (weights are stored [cout][cin] = [point][feature]), f16 hi/lo operands with fp32
    accumulation (fp32-class, 2.5x the rate of the library fp32 GEMM); the normalisation, the split and the padding of the point rows
    to a multiple of 256 are one sweep (gp_normalize_split_f16).  Otherwise torch.  Returns sim [A, N] (a view of [A, Np])."""
    N, Dt = F_teacher.shape
    if not (F_teacher.is_cuda and Dt % 32 == 0 and N >= 256):
        Fn = torch.nn.functional.normalize(F_teacher, p=2, dim=1)
        return Fn[anchor_indices] @ Fn.t()
    Np = (N + 255) // 256 * 256
    hi, lo = ops.normalize_split_f16(F_teacher.contiguous(), Np)
    pairs = ops.conv_pairs_build(anchor_indices.to(torch.int32).view(1, -1).contiguous(), chunk_rows=None)      # one gather-GEMM, one chunk
    sim = ops.sparse_conv_f16x3(None, pairs, hi.view(1, Np, Dt), lo.view(1, Np, Dt), None, None, relu=False, x_split=(hi, lo),
                                dense_single_offset=True)                      # phase 1 writes the fp32 matrix itself: no 2 x 2.4-GB second pass
    return sim[:, :N]


# --------------------------------------------------------------------------------------------------
def sample_contrastive_pairs_hybrid(F_teacher, neighbor_indices, anchor_indices, num_negatives):
    """affinity_module.py:1113-1136 after the randperm (anchors are the caller's draw).  neighbor_indices i64 [A,K]
    are the anchors' rows of the point kNN.  Returns positive [A], negative [A, num_negatives].
    On the device the arg-max and the 48 global negatives of a row come from one kernel (gp_sampler_select: two sweeps of the row instead
    of torch's arg-max sweep + 4-pass radix select + gather); the 15 local ones are torch.topk over the 96 neighbours."""
    sim = _anchor_similarities(F_teacher, anchor_indices)
    A, N = sim.shape
    rows = torch.arange(A, device=sim.device)
    if sim.is_cuda and N >= N_MACRO + 2:
        positive, macro = ops.sampler_select(sim, anchor_indices.contiguous(), N_MACRO, n=N)
        sim[rows, anchor_indices] = float("inf")
        sim[rows, positive] = float("inf")
    else:
        sim[rows, anchor_indices] = float("-inf")         # the reference clones the [A,N] matrix for this; marking in place is the same
        positive = torch.argmax(sim, dim=1)
        sim[rows, anchor_indices] = float("inf")          # in place, as the reference (sim_matrix_neg aliases the matrix)
        sim[rows, positive] = float("inf")
        _, macro = torch.topk(sim, k=N_MACRO, largest=False, dim=1)
    sims_local = torch.gather(sim, 1, neighbor_indices)
    _, hardest = torch.topk(sims_local, k=num_negatives - N_MACRO, largest=False, dim=1)
    micro = torch.gather(neighbor_indices, 1, hardest)
    return positive, torch.cat([macro, micro], dim=1)


# --------------------------------------------------------------------------------------------------
class StudentTrainer:
    """fp32 master weights of an AffinityPredictor (ME state_dict layout) + BatchNorm running statistics + AdamW
    state, and the forward/backward of one scene."""

    def __init__(self, state_dict, device="cuda", base_lr=1e-4, weight_decay=1e-5, temperature=0.07, bn_momentum=0.1,
                 bn_eps=1e-5, warmup_iters=0, main_iters=1, sync_bn=False, group=None):
        dev = torch.device(device)
        self.device = dev
        sd = {k: v.detach().clone() for k, v in state_dict.items()}
        w0 = sd["input_layer.0.kernel"].float()
        self.cin, self.hidden = w0.shape[1], w0.shape[2]
        self.cin_pad = _pad_to(self.cin, CONV_PAD)
        w0p = torch.zeros((27, self.cin_pad, self.hidden), dtype=torch.float32)
        w0p[:, :self.cin] = w0
        sd["input_layer.0.kernel"] = w0p
        self.num_blocks = 0
        while f"res_blocks.{self.num_blocks}.conv1.kernel" in sd:
            self.num_blocks += 1
        self.params, self.buffers = {}, {}
        for k, v in sd.items():
            if k.endswith("kernel") or k.endswith(".bn.weight") or k.endswith(".bn.bias"):
                self.params[k] = v.float().to(dev).contiguous()
            elif k.endswith("running_mean") or k.endswith("running_var"):
                self.buffers[k] = v.float().to(dev).contiguous()
        self.embed = self.params["output_layer.kernel"].shape[1]
        self.base_lr, self.weight_decay, self.temperature = base_lr, weight_decay, temperature
        self.bn_momentum, self.bn_eps = bn_momentum, bn_eps
        self.warmup_iters, self.main_iters = warmup_iters, main_iters
        self.opt_state = {}                          # name -> (exp_avg, exp_avg_sq), allocated on the first optimizer step
        self.steps_done = 0
        self.fast = self.hidden % 256 == 0           # f16x3 matrix-core path; else the exact fp32 MFMA kernel
        # SyncBatchNorm (run/train.py:212-213): statistics and backward reductions over the rows of ALL ranks -- four small
        # all-reduces per BatchNorm layer and step (sharding.sync_*); without a process group identical to the local path
        self.sync_bn, self.group = sync_bn, group
        self._ident_plans = {}

    # ---- state_dict in the reference layout (input kernel un-padded) -----------------------------------
    def state_dict(self):
        out = {k: v.clone() for k, v in self.params.items()}
        out["input_layer.0.kernel"] = out["input_layer.0.kernel"][:, :self.cin].contiguous()
        out.update({k: v.clone() for k, v in self.buffers.items()})
        return out

    # ---- convolutions ---------------------------------------------------------------------------------
    def _conv(self, x, x_split, w, ctx):
        """raw submanifold convolution y = sum_k x[nbr_k] @ w[k] (no BN, no ReLU)."""
        if self.fast and w.shape[2] % 256 == 0:
            hi, lo = ops.conv_weights_split(w, W_POW2)
            scale = ctx["inv_pow2"][w.shape[2]]
            return ops.sparse_conv_f16x3(x, ctx["pairs"], hi, lo, scale, None, relu=False, x_split=x_split)
        return ops.sparse_conv(x, ctx["nbr_map"], w)

    def _grad_split(self, dy, scale2=None):
        """dY * s split into f16 hi/lo with one extra all-zero row (the target of the padded pairs of the weight gradient); s = the
        power of two of gp_pow2_scale (max |dY| * s in [2^13, 2^14): 1e-6-sized gradients become normal f16 numbers), a device scalar --
        no host sync.  scale2 = [s, 1/s] when the producer of dY took it in its own sweep (bn_train_backward(dy_scale2=)), else one
        amax pass here.  Returns ((hi, lo) [nv+1, c], 1/s as a 1-element device tensor)."""
        if isinstance(dy, tuple):                        # bn_train_backward(split=True) wrote the planes already
            return dy, scale2[1:2]
        if scale2 is None:
            scale2 = ops.pow2_scale(dy)
        return ops.split_f16(dy, scale=scale2[0:1], extra_zero_rows=1), scale2[1:2]

    def _dgrad(self, dy, w, ctx, gs=None, residual=None):
        """dx = sum_k dy[nbr_k] @ w[26-k]^T (+ residual: the identity branch's gradient, added in the operator's output pass)."""
        if self.fast and w.shape[1] % 256 == 0:
            (hi_y, lo_y), inv_s = gs if gs is not None else self._grad_split(dy)
            nv = hi_y.shape[0] - 1
            hi, lo = ops.conv_weights_split(w, W_POW2, transpose_flip=True)
            scale = (ctx["inv_pow2"][w.shape[1]] * inv_s).contiguous()
            return ops.sparse_conv_f16x3(None, ctx["pairs"], hi, lo, scale, None, residual=residual, relu=False, x_split=(hi_y[:nv], lo_y[:nv]))
        dx = ops.sparse_conv(dy, ctx["nbr_map"], w.flip(0).transpose(1, 2).contiguous())
        return dx if residual is None else dx + residual

    def _wgrad(self, x, x_split, dy, ctx, cin, gs=None, out=None):
        """dW[k] = x[in_k]^T @ dy[out_k]: matrix-core kernel (gp_conv_wgrad_f16x3) when the shapes allow (cin >= 256,
        cout a multiple of 256), else library GEMMs on gathered rows.  out: the buffer to write it into (a gradient bucket's slice)."""
        cout = (dy[0] if isinstance(dy, tuple) else dy).shape[1]
        if self.fast and x_split is not None and cin >= 256 and cout % 256 == 0:
            ysplit, inv_s = gs if gs is not None else self._grad_split(dy)
            return ops.conv_wgrad_f16x3(x_split, ysplit, ctx["wgrad_plan"], cin, cin, cout, inv_scale=inv_s, out=out)
        dw = torch.zeros((27, cin, cout), dtype=torch.float32, device=dy.device)
        for k, (out_rows, in_rows) in enumerate(ctx["offset_pairs"]):
            if out_rows.numel():
                dw[k] = x[in_rows, :cin].t() @ dy[out_rows]
        return dw

    def gradient_order(self):
        """(name, shape) of every gradient in the order forward_backward produces them (the layout of sharding.GradientBuckets)"""
        P = self.params
        names = ["output_layer.kernel"]
        for i in reversed(range(self.num_blocks)):
            names += [f"res_blocks.{i}.norm2.bn.weight", f"res_blocks.{i}.norm2.bn.bias", f"res_blocks.{i}.conv2.kernel",
                      f"res_blocks.{i}.norm1.bn.weight", f"res_blocks.{i}.norm1.bn.bias", f"res_blocks.{i}.conv1.kernel"]
        names += ["input_layer.1.bn.weight", "input_layer.1.bn.bias", "input_layer.0.kernel"]
        return [(n, tuple(P[n].shape)) for n in names]

    # ---- one scene: loss and gradients --------------------------------------------------------------
    def forward_backward(self, X, nbr_map, sample_to_voxel, point_to_batch, num_anchors, num_negatives, update_running=True, grad_sink=None):
        """X fp32 [Nv, cin_pad] voxel inputs (rows in the order of nbr_map i32 [27,Nv]); sample_to_voxel i64 [S];
        point_to_batch i64 [A*(2+Nn)].  Returns (loss 0-d device tensor, grads dict, embeddings [Nv, embed]).
        grad_sink (sharding.GradientBuckets over gradient_order()): the weight gradients are written into its slices and every gradient is
        announced the moment its kernels are enqueued, so that the buckets' all-reduces run beside the rest of the backward pass; the
        returned dict then holds the slices -- LOCAL sums until grad_sink.finish() has averaged them."""
        P, B = self.params, self.buffers
        dev = X.device
        Nv = X.shape[0]
        ctx = {"nbr_map": nbr_map, "pairs": ops.conv_pairs_build(nbr_map, col_tiles=max(1, self.hidden // 256)) if self.fast else None,
               "inv_pow2": {c: torch.full((c,), 1.0 / W_POW2, dtype=torch.float32, device=dev) for c in {self.hidden}}}
        if nbr_map.is_cuda:
            kk, rr, rin, counts = ops.kernel_map_pairs(nbr_map)
            offs = [0]
            for n in counts:
                offs.append(offs[-1] + n)
            ctx["offset_pairs"] = [(rr[offs[k]:offs[k + 1]], rin[offs[k]:offs[k + 1]]) for k in range(len(counts))]
            ctx["wgrad_plan"] = ops.wgrad_plan_from_pairs(kk, rr, rin, counts, Nv) if self.fast else None
        else:
            ctx["offset_pairs"] = []
            for k in range(27):
                m = nbr_map[k]
                out_rows = torch.nonzero(m >= 0).squeeze(1)
                ctx["offset_pairs"].append((out_rows, m[out_rows].long()))
            ctx["wgrad_plan"] = None
        mom = self.bn_momentum
        n_all = sharding.sync_row_count(Nv, dev, self.group) if self.sync_bn else Nv       # one count per step, not one per layer

        def bn_fwd(y, prefix, residual=None, want_split=True, want_f32=True):
            """(out fp32 | None, planes, statistics).  want_f32=False: a layer without a residual whose fp32 output nothing reads -- the next
            convolution and the weight gradient take the planes, the backward pass recomputes the ReLU mask from y (bn_bwd(beta=))"""
            want_f32 = want_f32 or not (want_split and self.fast)
            rm, rv = (B[prefix + ".bn.running_mean"], B[prefix + ".bn.running_var"]) if update_running else (None, None)
            if self.sync_bn:
                c = y.shape[1]
                mean, var, n_tot = sharding.sync_batch_stats(lambda m: ops.col_sums_f64(y, c, m), y.shape[0], c, dev, self.group,
                                                             n_total=n_all)
                out, sp = ops.bn_train_apply(y, mean, var, P[prefix + ".bn.weight"], P[prefix + ".bn.bias"], self.bn_eps, residual=residual,
                                             relu=True, want_split=want_split and self.fast, momentum=mom, want_f32=want_f32)
                if rm is not None:
                    sharding.sync_running_stats(rm, rv, mean, var, n_tot, mom)
                return out, sp, (mean, var, n_tot)
            mean, var = ops.col_stats(y)
            out, sp = ops.bn_train_apply(y, mean, var, P[prefix + ".bn.weight"], P[prefix + ".bn.bias"], self.bn_eps, residual=residual,
                                         relu=True, want_split=want_split and self.fast, momentum=mom, running_mean=rm, running_var=rv,
                                         want_f32=want_f32)
            return out, sp, (mean, var)

        def bn_bwd(dout, act, y, st, gamma, want_dz=False, beta=None, allow_split=True):
            """(dy, dgamma, dbeta, dz | None, scale2 of dy | None); with SyncBatchNorm the dy formula uses the reductions over all ranks.
            beta (a layer without a residual): the ReLU mask comes from y, act is not read"""
            if beta is not None:
                act = None
            sc2 = torch.empty(2, dtype=torch.float32, device=dev) if self.fast else None
            if self.sync_bn:
                c = st[0].shape[0]
                g_sums, l_sums = sharding.sync_bwd_sums(ops.bn_bwd_sums_f64(dout, act, y, st[0], st[1], self.bn_eps,
                                                                            mask_affine=(gamma, beta) if beta is not None else None), self.group)
                r = ops.bn_bwd_apply(dout, act, y, st[0], st[1], self.bn_eps, gamma, g_sums, st[2], want_dz=want_dz, dy_scale2=sc2, beta_mask=beta)
                dy, dz = r if want_dz else (r, None)
                return dy, l_sums[c:].clone(), l_sums[:c].clone(), dz, sc2
            # one process: the sweep writes the gradient's split planes itself (scale from a bound taken in the reduction pass)
            r = ops.bn_train_backward(dout, act, y, st[0], st[1], self.bn_eps, gamma, want_dz=want_dz, dy_scale2=sc2, beta_mask=beta,
                                      split=allow_split and self.fast and y.shape[1] % 256 == 0)
            return r[0], r[1], r[2], (r[3] if want_dz else None), sc2

        # ---------------- forward (activations kept for the backward pass)
        saved = []
        xs = ops.split_f16(X, self.cin_pad) if self.fast else None
        y0 = self._conv(X, xs, P["input_layer.0.kernel"], ctx)
        h, hs, st0 = bn_fwd(y0, "input_layer.1")
        blocks = []
        for i in range(self.num_blocks):
            y1 = self._conv(h, hs, P[f"res_blocks.{i}.conv1.kernel"], ctx)
            a1, a1s, st1 = bn_fwd(y1, f"res_blocks.{i}.norm1", want_f32=False)
            y2 = self._conv(a1, a1s, P[f"res_blocks.{i}.conv2.kernel"], ctx)
            h2, h2s, st2 = bn_fwd(y2, f"res_blocks.{i}.norm2", residual=h)
            blocks.append((h, y1, a1, st1, y2, st2, h2, hs, a1s))
            h, hs = h2, h2s
        Wo = P["output_layer.kernel"]
        dense_hip = self.fast and hs is not None and Wo.shape[0] % 32 == 0 and Wo.shape[0] >= 256 and Wo.shape[1] % 128 == 0 and Wo.shape[1] <= 256
        E = ops.sparse_conv(h, None, Wo.unsqueeze(0).contiguous()) if dense_hip else h @ Wo
        loss, dE = ops.infonce_fwd_bwd(E, sample_to_voxel, point_to_batch, num_anchors, num_negatives, self.temperature)

        # ---------------- backward
        class _Grads(dict):                               # g[name] = tensor: into the sink's slice (copied unless it was written there)
            def __setitem__(d, name, t):
                if grad_sink is not None:
                    v = grad_sink.view(name)
                    if t.data_ptr() == v.data_ptr():
                        grad_sink.ready(name)
                    else:
                        grad_sink.put(name, t)
                    t = v
                dict.__setitem__(d, name, t)
        g = _Grads()
        buf = (lambda name: grad_sink.view(name)) if grad_sink is not None else (lambda name: None)
        if dense_hip:
            # dW = h^T dE on the weight-gradient kernel: one "offset" whose pairs are the identity; the gradient rides in a
            # 256-column operand (columns >= embed are zero) because the kernel's tiles are 256 x 256
            plan = self._ident_plans.get((Nv, str(dev)))              # the identity plan depends on the row count only
            if plan is None:
                ident = torch.arange(Nv, device=dev)
                plan = self._ident_plans[(Nv, str(dev))] = ops.wgrad_plan_build([(ident, ident)], Nv)
                if len(self._ident_plans) > 8:
                    self._ident_plans.pop(next(iter(self._ident_plans)))
            dEp = torch.zeros((Nv, 256), dtype=torch.float32, device=dev)
            dEp[:, :Wo.shape[1]] = dE
            ysplit, inv_s = self._grad_split(dEp)
            g["output_layer.kernel"] = ops.conv_wgrad_f16x3(hs, ysplit, plan, Wo.shape[0], Wo.shape[0], 256, inv_scale=inv_s)[0, :, :Wo.shape[1]].contiguous()
            dh = ops.sparse_conv(dE.contiguous(), None, Wo.t().contiguous().unsqueeze(0))
        else:
            g["output_layer.kernel"] = h.t() @ dE
            dh = dE @ Wo.t()
        for i in reversed(range(self.num_blocks)):
            h_in, y1, a1, st1, y2, st2, h_out, h_in_s, a1_s = blocks[i]
            dy2, dg2, db2, dz, sc2 = bn_bwd(dh, h_out, y2, st2, P[f"res_blocks.{i}.norm2.bn.weight"], want_dz=True)
            g[f"res_blocks.{i}.norm2.bn.weight"], g[f"res_blocks.{i}.norm2.bn.bias"] = dg2, db2
            gs2 = self._grad_split(dy2, sc2) if self.fast else None
            g[f"res_blocks.{i}.conv2.kernel"] = self._wgrad(a1, a1_s, dy2, ctx, self.hidden, gs2, out=buf(f"res_blocks.{i}.conv2.kernel"))
            da1 = self._dgrad(dy2, P[f"res_blocks.{i}.conv2.kernel"], ctx, gs2)
            dy1, dg1, db1, _, sc1 = bn_bwd(da1, a1, y1, st1, P[f"res_blocks.{i}.norm1.bn.weight"], beta=P[f"res_blocks.{i}.norm1.bn.bias"])
            g[f"res_blocks.{i}.norm1.bn.weight"], g[f"res_blocks.{i}.norm1.bn.bias"] = dg1, db1
            gs1 = self._grad_split(dy1, sc1) if self.fast else None
            g[f"res_blocks.{i}.conv1.kernel"] = self._wgrad(h_in, h_in_s, dy1, ctx, self.hidden, gs1, out=buf(f"res_blocks.{i}.conv1.kernel"))
            dh = self._dgrad(dy1, P[f"res_blocks.{i}.conv1.kernel"], ctx, gs1, residual=dz)
        h0 = blocks[0][0] if self.num_blocks else h
        dy0, dg0, db0, _, sc0 = bn_bwd(dh, h0, y0, st0, P["input_layer.1.bn.weight"], beta=P["input_layer.1.bn.bias"],
                                       allow_split=self.cin_pad >= 256)          # (a narrow input layer's weight gradient takes fp32 rows)
        g["input_layer.1.bn.weight"], g["input_layer.1.bn.bias"] = dg0, db0
        g["input_layer.0.kernel"] = self._wgrad(X, xs, dy0, ctx, self.cin_pad, self._grad_split(dy0, sc0) if self.fast else None,
                                                out=buf("input_layer.0.kernel"))
        return loss, dict(g), E

    # ---- optimizer ---------------------------------------------------------------------------------------
    def optimizer_step(self, grads):
        """AdamW (run/train.py:198) with the three learning-rate groups and the warm-up/cosine schedule."""
        self.steps_done += 1
        for name, p in self.params.items():
            lr = lr_schedule(self.steps_done - 1, self.base_lr, param_group(name), self.warmup_iters, self.main_iters) \
                if (self.warmup_iters or self.main_iters > 1) else self.base_lr * GROUP_LR[param_group(name)]
            if name not in self.opt_state:
                self.opt_state[name] = (torch.zeros_like(p), torch.zeros_like(p))
            m, v = self.opt_state[name]
            ops.adamw_step_(p, grads[name].contiguous(), m, v, lr, self.steps_done, weight_decay=self.weight_decay)

    # ---- one scene of the reference's forward (everything after the lift) -------------------------------------
    def scene_step(self, F_lift, gauss, inds_reconstruct, coords_3d, xyz, F_teacher, anchor_indices, num_negatives=63, K=96,
                   optimize=True, grad_sink=None):
        """F_lift fp32 [N,D] lifted 2D features, gauss fp32 [N,6], inds_reconstruct i64 [N] point -> voxel row,
        coords_3d [Nv,3] integer voxel coordinates (float or int), xyz fp32 [N,3], F_teacher fp32 [N,Dt],
        anchor_indices i64 [A] (the reference's randperm draw).  Returns dict(loss, ...)."""
        dev = self.device
        nbrs, flag = ops.knn_points(xyz.contiguous(), anchor_indices, K)
        positive, negative = sample_contrastive_pairs_hybrid(F_teacher, nbrs, anchor_indices, num_negatives)
        if int(flag.item()):
            raise RuntimeError("knn_points: degenerate duplicate points around an anchor")
        A = anchor_indices.shape[0]
        all_idx, point_to_batch = torch.unique(torch.cat([anchor_indices, positive, negative.flatten()]), return_inverse=True)
        vox = inds_reconstruct[all_idx]
        uniq_vox, sample_to_voxel = torch.unique(vox, return_inverse=True)
        # voxel subset in Morton order (kernel map, conv tiles and BN are order-independent up to fp32 summation order)
        cs_ref = coords_3d[uniq_vox].floor().to(torch.int32).contiguous()
        perm, rank = ops.morton_order(cs_ref)
        cs = cs_ref[perm.long()].contiguous()
        s2v = rank.long()[sample_to_voxel].contiguous()
        order = torch.sort(s2v, stable=True).indices
        Nvs = cs.shape[0]
        seg = torch.zeros(Nvs + 1, dtype=torch.int64, device=dev)
        seg[1:] = torch.bincount(s2v, minlength=Nvs).cumsum(0)
        D = F_lift.shape[1]
        X = torch.zeros((Nvs, self.cin_pad), dtype=torch.float32, device=dev)
        ops.scatter_mean_csr(F_lift[all_idx].contiguous(), D, order, seg, Nvs, X, col0=0)
        ops.scatter_mean_csr(gauss[all_idx].contiguous(), GEO_DIM, order, seg, Nvs, X, col0=D)
        grid = ops.grid_build(cs)
        nbr_map = ops.kernel_map_build(grid, cs)
        loss, grads, E = self.forward_backward(X, nbr_map, s2v, point_to_batch.contiguous(), A, num_negatives, grad_sink=grad_sink)
        if grad_sink is not None:
            grads = grad_sink.finish()                    # (averaged over the ranks; the all-reduces ran beside the backward pass)
        if optimize:
            self.optimizer_step(grads)
        return {"loss": loss, "grads": grads, "num_voxels": Nvs, "num_samples": int(all_idx.shape[0]), "positive": positive,
                "negative": negative, "neighbors": nbrs, "perm": perm, "uniq_vox": uniq_vox, "embeddings": E, "nbr_map": nbr_map}


class _LossWithGradients(torch.autograd.Function):
    """Hands the gradients computed by the HIP backward pass to torch.autograd, so that the reference's training loop
    (`loss = model(batch); loss.backward(); optimizer.step()`, run/train.py:346-353) runs unchanged."""

    @staticmethod
    def forward(ctx, loss, grads, *params):
        ctx.grads = grads
        return loss.detach().clone()

    @staticmethod
    def backward(ctx, g):
        return (None, None) + tuple(g * gr for gr in ctx.grads)


def training_forward(student_module, F_lift, gauss, inds_reconstruct, coords_3d, xyz, F_teacher, num_anchors=4096,
                     num_negatives=63, temperature=0.07, K=96, anchor_indices=None):
    """SonataXAffinityTrainer.forward after the lift and the teacher (affinity_module.py:1157-1233) for an
    AffinityPredictor-shaped nn.Module whose parameters live on the device.  Returns the loss as a tensor whose
    .backward() fills the parameters' .grad; BatchNorm running statistics are updated in place."""
    dev = F_lift.device
    N = F_lift.shape[0]
    A = min(num_anchors, N // 3)
    if anchor_indices is None:
        anchor_indices = torch.randperm(N, device=dev)[:A]               # affinity_module.py:1112
    named = dict(student_module.named_parameters())
    # SyncBatchNorm whenever a process group is up (run/train.py:212-213 converts the student before wrapping it in DDP)
    tr = StudentTrainer(student_module.state_dict(), dev, temperature=temperature,
                        bn_momentum=student_module.input_layer[1].bn.momentum, bn_eps=student_module.input_layer[1].bn.eps,
                        sync_bn=sharding._world() > 1)
    # data-parallel ranks (run/train.py:206 wraps the student in DDP): the gradients are averaged over the ranks by bucketed all-reduces
    # launched INSIDE the backward pass (sharding.GradientBuckets); the loss tensor says so (`gradients_averaged`), train_driver.train
    # then does not all-reduce again
    sink = sharding.GradientBuckets(tr.gradient_order(), dev) if sharding._world() > 1 else None
    out = tr.scene_step(F_lift, gauss, inds_reconstruct, coords_3d, xyz, F_teacher, anchor_indices, num_negatives, K, optimize=False,
                        grad_sink=sink)
    names = [n for n in named if n in out["grads"]]
    grads = []
    for n in names:
        g = out["grads"][n]
        if n == "input_layer.0.kernel":
            g = g[:, :tr.cin].contiguous()
        grads.append(g.view_as(named[n]))
    with torch.no_grad():
        for k, b in student_module.named_buffers():
            if k in tr.buffers:
                b.copy_(tr.buffers[k])
            elif k.endswith("num_batches_tracked"):
                b += 1
    loss = _LossWithGradients.apply(out["loss"], grads, *[named[n] for n in names])
    loss.gradients_averaged = sink is not None
    return loss


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW with the update done by gp_adamw_step (one HIP launch per parameter tensor).  Same constructor
    arguments, param_groups and state_dict layout ('step', 'exp_avg', 'exp_avg_sq'), so it can replace the optimizer of
    run/train.py:198 and load / save its checkpoints.  amsgrad / maximize are not supported."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                if not (p.is_cuda and p.is_contiguous() and p.dtype == torch.float32):
                    raise ValueError("FusedAdamW needs contiguous fp32 parameters on the device")
                ops.adamw_step_(p.data, p.grad.contiguous(), st["exp_avg"], st["exp_avg_sq"], group["lr"], int(st["step"]),
                                weight_decay=group["weight_decay"], betas=group["betas"], eps=group["eps"])
        return loss
