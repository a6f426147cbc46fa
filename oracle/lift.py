"""Oracle rows 5-7: 2D->3D feature lift (test infrastructure).

Follows models/affinity_module.py:416-449 (dense-feature lift), :495-646 (per-view mask-embedding
lift) and :647-696 (multi-view consensus top-3 fusion + scene-level nearest-seen fill).
Third-party pieces: sklearn KDTree (exact 1-NN, fp64) and torch F.interpolate(bicubic, antialias).
"""
import numpy as np
import torch
import torch.nn.functional as F
from sklearn.neighbors import KDTree

_f = np.float32
_d = np.float64


# --------------------------------------------------------------------------------------------
# bicubic antialias resize, restated explicitly (F.interpolate(mode="bicubic",
# align_corners=False, antialias=True), affinity_module.py:527-533).  Verified bit-exact against
# torch 2.10 CPU in the build container: Keys a=-0.5 kernel evaluated in fp32 with fused
# multiply-adds, weights renormalised, separable horizontal-then-vertical passes whose
# accumulation is  t = s0*w0 ; t = fma(s_j, w_j, t).
# --------------------------------------------------------------------------------------------
def _sfma(a, b, c):
    return _f(_d(a) * _d(b) + _d(c))


def _cubic_aa(x):
    A = _f(-0.5)
    x = _f(abs(x))
    if x < 1:
        t = _sfma(_f(A + _f(2)), x, -_f(A + _f(3)))
        t = _f(t * x)
        return _sfma(t, x, _f(1))
    if x < 2:
        t = _sfma(A, x, -_f(_f(5) * A))
        t = _sfma(t, x, _f(_f(8) * A))
        return _sfma(t, x, -_f(_f(4) * A))
    return _f(0)


def aa_bicubic_weights(in_size, out_size):
    """Per output index: (first input index, fp32 weights[<=4 when upsampling])."""
    scale = _f(_f(in_size) / _f(out_size))
    support = _f(_f(2.0) * scale) if scale >= 1 else _f(2.0)
    invscale = _f(_f(1.0) / scale) if scale >= 1 else _f(1.0)
    out = []
    for i in range(out_size):
        center = _f(_d(scale) * (i + 0.5))
        xmin = max(0, int(_d(_f(center - support)) + 0.5))
        xmax = min(in_size, int(_d(_f(center + support)) + 0.5))
        w = [_cubic_aa(_f((_d(_f(_f(j + xmin) - center)) + 0.5) * _d(invscale)))
             for j in range(xmax - xmin)]
        tot = _f(0)
        for v in w:
            tot = _f(tot + v)
        out.append((xmin, np.array([_f(v / tot) for v in w], dtype=_f)))
    return out


def _fma_arr(a, w, t):
    return (a.astype(_d) * _d(w) + t.astype(_d)).astype(_f)


def bicubic_aa_resize_explicit(x, out_hw):
    """x: fp32 [Q,h,w] numpy -> [Q,H,W].  Bit-exact restatement of the torch CPU kernel."""
    x = np.asarray(x, dtype=_f)
    Q, h, w = x.shape
    H, W = out_hw
    wh, wv = aa_bicubic_weights(w, W), aa_bicubic_weights(h, H)
    tmp = np.empty((Q, h, W), _f)
    for i, (x0, ws) in enumerate(wh):
        t = (x[:, :, x0] * ws[0]).astype(_f)
        for j in range(1, len(ws)):
            t = _fma_arr(x[:, :, x0 + j], ws[j], t)
        tmp[:, :, i] = t
    out = np.empty((Q, H, W), _f)
    for i, (y0, ws) in enumerate(wv):
        t = (tmp[:, y0, :] * ws[0]).astype(_f)
        for j in range(1, len(ws)):
            t = _fma_arr(tmp[:, y0 + j, :], ws[j], t)
        out[:, i, :] = t
    return out


# --------------------------------------------------------------------------------------------
# exact 1-NN fill (sklearn KDTree, as the reference)
# --------------------------------------------------------------------------------------------
def nn1_indices(ref_xyz, query_xyz):
    """Index into ref of the nearest reference point per query (KDTree, fp64; :619-621,:693-694)."""
    tree = KDTree(np.asarray(ref_xyz))
    _, idx = tree.query(np.asarray(query_xyz), k=1)
    return idx.reshape(-1)


def nn1_indices_bruteforce(ref_xyz, query_xyz, chunk=2048):
    """Independent check: brute force fp64 squared distances, lowest index on ties."""
    r = np.asarray(ref_xyz, dtype=_d)
    q = np.asarray(query_xyz, dtype=_d)
    out = np.empty(q.shape[0], np.int64)
    for s in range(0, q.shape[0], chunk):
        dd = ((q[s:s + chunk, None, :] - r[None, :, :]) ** 2).sum(-1)
        out[s:s + chunk] = dd.argmin(1)
    return out


# --------------------------------------------------------------------------------------------
# row 5: dense-feature lift
# --------------------------------------------------------------------------------------------
def lift_dense(feat2d_views, point_idx_views, x_views, y_views, scene_coords):
    """affinity_module.py:416-449.  feat2d_views[v]: fp32 [D,H,W]; point_idx_views[v]: visible
    point ids (ascending); x = pixel row, y = pixel col.  Mean over views, unseen points take the
    feature of the nearest seen point (xyz, k=1)."""
    N = scene_coords.shape[0]
    D = feat2d_views[0].shape[0]
    s = torch.zeros((N, D), dtype=torch.float32)
    cnt = torch.zeros((N, 1), dtype=torch.float32)
    for f2, pi, x, y in zip(feat2d_views, point_idx_views, x_views, y_views):
        lifted = f2[:, x, y].permute(1, 0)
        s.index_add_(0, pi, lifted)
        cnt.index_add_(0, pi, torch.ones((len(pi), 1), dtype=torch.float32))
    cnt[cnt == 0] = 1e-6
    out = s / cnt
    seen = cnt.squeeze() > 1e-5
    if (~seen).any() and seen.any():
        idx = nn1_indices(scene_coords[seen].numpy(), scene_coords[~seen].numpy())
        out[~seen] = out[seen][torch.from_numpy(idx)]
    return out, seen


def lift_lseg(feat_lo_views, image_shape, point_idx_views, x_views, y_views, scene_coords):
    """affinity_module.py:404-452: each view's low-resolution map [D,h,w] is resized to the image size with
    F.interpolate(bilinear, align_corners=True) (torch CPU here), then lifted exactly as lift_dense."""
    full = [F.interpolate(f.unsqueeze(0), size=tuple(image_shape), mode="bilinear", align_corners=True).squeeze(0)
            for f in feat_lo_views]
    return lift_dense(full, point_idx_views, x_views, y_views, scene_coords)


# --------------------------------------------------------------------------------------------
# row 6: mask-embedding lift for one view
# --------------------------------------------------------------------------------------------
def segment_scores(pred_logits):
    """affinity_module.py:544: scores, labels = softmax(logits)[..., :-1].max(-1)."""
    return F.softmax(pred_logits, dim=-1)[..., :-1].max(-1)


def lift_masks_view(pred_masks, pred_logits, mask_embed, text_embed, logit_scale,
                    x_label, y_label, coords_view, mask_shape, explicit_resize=False,
                    return_debug=False):
    """affinity_module.py:526-630 for one view.
    pred_masks fp32 [Q,h,w]; pred_logits [Q,C+1]; mask_embed [Q,D]; x_label=row, y_label=col of the
    n_v visible points; coords_view fp32 [n_v,3].  Returns (f [n_v,D] normalised, logits [n_v,C])."""
    if explicit_resize:
        resized = torch.from_numpy(bicubic_aa_resize_explicit(pred_masks.numpy(), tuple(mask_shape)))
    else:
        resized = F.interpolate(pred_masks[None], size=tuple(mask_shape), mode="bicubic",
                                align_corners=False, antialias=True)[0]
    scores, _labels = segment_scores(pred_logits)
    keep = scores > 0.0
    cur_scores = scores[keep]
    cur_masks = resized[keep].sigmoid()
    cur_embed = mask_embed[keep]
    n_v = x_label.shape[0]
    D = mask_embed.shape[-1]
    dbg = {}
    if cur_masks.shape[0] == 0:
        feat = torch.zeros((n_v, D))
    else:
        cur_prob = cur_scores.view(-1, 1, 1) * cur_masks
        ids = cur_prob.argmax(0)
        # :560-570 segment validity (a no-op for sampled points, kept for fidelity)
        kept, masks = [], []
        for k in range(cur_embed.shape[0]):
            area = (ids == k).sum().item()
            orig = (cur_masks[k] >= 0.5).sum().item()
            m = (ids == k) & (cur_masks[k] >= 0.5)
            if area > 0 and orig > 0 and m.sum().item() > 0:
                kept.append(k)
                masks.append(m)
        if not kept:
            feat = torch.zeros((n_v, D))
        else:
            emb = cur_embed[kept]
            stack = torch.stack(masks, 0)
            m3 = stack[:, x_label, y_label]
            feat = torch.zeros((n_v, D))
            cnt = torch.zeros((n_v, 1))
            for sm, e in zip(m3, emb):
                feat[sm] += e
                cnt[sm] += 1
            cnt[cnt == 0] = 1e-5
            feat = feat / cnt
        if return_debug:
            keep_idx = torch.where(keep)[0]
            prob_at = cur_prob[:, x_label, y_label]            # [Qk, n_v]
            top2 = prob_at.topk(min(2, prob_at.shape[0]), dim=0)
            dbg["seg"] = keep_idx[top2.indices[0]]
            dbg["margin"] = (top2.values[0] - top2.values[1]) if prob_at.shape[0] > 1 else None
            dbg["logit_at"] = resized[keep_idx[top2.indices[0]], x_label, y_label]
            dbg["resized"] = resized
    # :604-625 in-view fill of zero-sum rows
    zero = torch.sum(feat, dim=1) == 0
    dbg["zero_before_fill"] = zero.clone()
    if zero.any():
        true_idx = torch.where(~zero)[0]
        idx = nn1_indices(coords_view[~zero].numpy(), coords_view[zero].numpy())
        feat[zero] = feat[true_idx[torch.from_numpy(idx)]]
        dbg["fill_src"] = true_idx[torch.from_numpy(idx)]
    f = F.normalize(feat, dim=-1)
    t = F.normalize(text_embed, dim=-1)
    logits = logit_scale * (f @ t.t())
    if return_debug:
        return f, logits, dbg
    return f, logits


# --------------------------------------------------------------------------------------------
# row 7: consensus top-3 fusion
# --------------------------------------------------------------------------------------------
def fuse_views_top3(N, point_idx_views, f_views, logits_views, scene_coords, faithful_loops=False,
                    chunk_size=50000, return_debug=False):
    """affinity_module.py:633-696.  point_idx_views[v] ascending ids of the points visible in
    surviving view v; f_views[v] [n_v,D], logits_views[v] [n_v,C].
    faithful_loops=True keeps the reference's per-point Python loops (:633-638,:664-670)."""
    D = f_views[0].shape[1]
    C = logits_views[0].shape[1]
    out = torch.zeros((N, D), dtype=torch.float32)
    counter = torch.zeros(N, dtype=torch.long)
    dbg = {}
    if faithful_loops:
        from collections import defaultdict
        info = defaultdict(list)
        for pi, f, lg in zip(point_idx_views, f_views, logits_views):
            for i, g in enumerate(pi):
                info[g.item()].append((f[i], lg[i]))
            counter[pi] += 1
        keys = torch.tensor(list(info.keys()))
        for s in range(0, len(keys), chunk_size):
            ck = keys[s:s + chunk_size]
            M = max(len(info[i.item()]) for i in ck)
            pf = torch.zeros(len(ck), M, D)
            pl = torch.zeros(len(ck), M, C)
            valid = torch.zeros(len(ck), M, dtype=torch.bool)
            for j, g in enumerate(ck):
                vd = info[g.item()]
                valid[j, :len(vd)] = True
                for k, (a, b) in enumerate(vd):
                    pf[j, k] = a
                    pl[j, k] = b
            if return_debug:
                out[ck], mc, mk = _fuse_chunk(pf, pl, valid, margins=True)
                dbg.setdefault("class_margin", torch.full((N,), float("inf")))[ck] = mc
                dbg.setdefault("cut_margin", torch.full((N,), float("inf")))[ck] = mk
            else:
                out[ck] = _fuse_chunk(pf, pl, valid)
    else:
        # vectorised: same math; slot k of point p = its k-th surviving view in ascending view order
        for pi in point_idx_views:
            counter[pi] += 1
        seen_idx = torch.where(counter > 0)[0]
        slot_of = torch.zeros(N, dtype=torch.long)
        M = int(counter.max().item()) if len(seen_idx) else 0
        row_of = torch.full((N,), -1, dtype=torch.long)
        row_of[seen_idx] = torch.arange(len(seen_idx))
        for s in range(0, len(seen_idx), chunk_size):
            ck = seen_idx[s:s + chunk_size]
            lo, hi = s, s + len(ck)
            Mc = int(counter[ck].max().item())
            pf = torch.zeros(len(ck), Mc, D)
            pl = torch.zeros(len(ck), Mc, C)
            valid = torch.zeros(len(ck), Mc, dtype=torch.bool)
            slot = torch.zeros(N, dtype=torch.long)
            for pi, f, lg in zip(point_idx_views, f_views, logits_views):
                r = row_of[pi]
                sel = (r >= lo) & (r < hi)
                rr = r[sel] - lo
                ss = slot[pi[sel]]
                pf[rr, ss] = f[sel]
                pl[rr, ss] = lg[sel]
                valid[rr, ss] = True
                slot[pi[sel]] += 1
            if return_debug:
                out[ck], mc, mk = _fuse_chunk(pf, pl, valid, margins=True)
                dbg.setdefault("class_margin", torch.full((N,), float("inf")))[ck] = mc
                dbg.setdefault("cut_margin", torch.full((N,), float("inf")))[ck] = mk
            else:
                out[ck] = _fuse_chunk(pf, pl, valid)
        del slot_of, M
    seen = counter != 0
    dbg["seen"] = seen
    if seen.any() and (~seen).any():
        true_idx = torch.where(seen)[0]
        idx = nn1_indices(scene_coords[seen].numpy(), scene_coords[~seen].numpy())
        out[~seen] = out[true_idx[torch.from_numpy(idx)]]
        dbg["fill_src"] = true_idx[torch.from_numpy(idx)]
    if return_debug:
        return out, dbg
    return out


def _fuse_chunk(pf, pl, valid, margins=False):
    """affinity_module.py:672-683.  margins=True also returns the two decision margins of every point (parity bookkeeping, not
    part of the reference): consensus class (best minus second-best mean logit) and the top-3 cut (third minus fourth agreement
    among the valid views; +inf with three views or fewer)."""
    Mc = pf.shape[1]
    K = min(Mc, 3)
    avg = pl.sum(1) / valid.sum(1, keepdim=True).clamp(min=1)
    cstar = torch.argmax(avg, dim=1)
    agree = torch.gather(pl, 2, cstar.view(-1, 1, 1).expand(-1, Mc, -1)).squeeze(-1)
    agree.masked_fill_(~valid, -torch.inf)
    top_s, top_i = torch.topk(agree, k=K, dim=1)
    top_f = torch.gather(pf, 1, top_i.unsqueeze(-1).expand(-1, -1, pf.shape[2]))
    w = F.softmax(top_s, dim=1)
    out = (top_f * w.unsqueeze(-1)).sum(1)
    if not margins:
        return out
    if avg.shape[1] > 1:
        a2 = avg.topk(2, dim=1).values
        m_class = a2[:, 0] - a2[:, 1]
    else:
        m_class = torch.full((pf.shape[0],), float("inf"))
    m_cut = torch.full((pf.shape[0],), float("inf"))
    if Mc > 3:
        s4 = agree.topk(4, dim=1).values
        has4 = torch.isfinite(s4[:, 3])
        m_cut[has4] = (s4[:, 2] - s4[:, 3])[has4]
    return out, m_class, m_cut
