"""Near-tie bookkeeping of the end-to-end parity tests (TEST INFRASTRUCTURE, like the rest of oracle/: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline import it).

The hot path contains discrete decisions taken on fp32 quantities that the device and the CPU oracle round differently (device
`expf` / sigmoid, a 512-term dot product summed in another order):

  D1  the segment of a visible pixel       arg-max_q score_q * sigmoid(mask_q)        models/affinity_module.py:556-557
  D2  "is the pixel inside its segment"    sigmoid(mask) >= 0.5                       :566-570
  D3  the consensus class of a point       arg-max_c mean_v logits[v, c]              :672-674
  D4  the three views that vote            top-3 of logits[v, c*]                     :676-678
  D5  the class of a point                 arg-max_c logit_scale * <f, t_c>           run/validation.py:413-416

Everything else on the path is either exact index work or continuous.  A device result may differ from the oracle's by more than
the continuous tolerance ONLY at a point whose decision margin -- measured on the ORACLE's own numbers -- is below the rounding
noise of the quantity decided (eps_* below).  `lift_near_ties` returns those points for D1-D4 (propagated through the two
nearest-neighbour fills, which copy another point's row); `class_near_ties` for D5.  `check_scene` is the rule the tests apply:

  * every point whose LIFTED feature differs from the oracle's by more than `tol_lift` must be a D1-D4 near tie;
  * if there is no such point, the pooled features must be within `tol` of the oracle's at EVERY point;
    otherwise (a near tie went the other way: the flipped point changes its voxel's mean, and with it the student's input over its
    receptive field) rows 8-12 of the oracle are re-run FROM THE DEVICE'S LIFT and the pooled features must be within `tol` of that
    at every point -- no point is ever excused from the continuous tolerance.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import lift, metric
from . import pipeline as o_pipe

EPS_PROB = 1e-6       # D1: probabilities <= 1, fp32 sigmoid / softmax: a few ulp
EPS_LOGIT = 1e-5      # D2: the resized mask logit at the winning segment, against 0
EPS_FUSE = 1e-4       # D3, D4: logits of magnitude <= logit_scale (14..100), 512-term dot products in another order
EPS_CLASS = 1e-4      # D5: the same quantity


def lift_near_ties(views, vlm, xyz32, mask_shape, N, eps_prob=EPS_PROB, eps_logit=EPS_LOGIT, eps_fuse=EPS_FUSE):
    """views: oracle.pipeline.loader_math(...)["views"]; vlm: the arrays of synthetic.make_vlm_outputs.
    -> bool [N]: the lifted feature of the point hangs on a decision D1-D4 inside the rounding noise."""
    text = torch.from_numpy(vlm["text_embed"])
    scale = float(vlm["logit_scale"])
    flag = torch.zeros(N, dtype=torch.bool)
    fs, lgs = [], []
    for v in views:
        s = v["src_view"]
        f, lg, dbg = lift.lift_masks_view(torch.from_numpy(vlm["pred_masks"][s]), torch.from_numpy(vlm["pred_logits"][s]),
                                          torch.from_numpy(vlm["mask_embed"][s]), text, scale, v["x"], v["y"], xyz32[v["pt"]],
                                          mask_shape, return_debug=True)
        fs.append(f), lgs.append(lg)
        fl = torch.zeros(len(v["pt"]), dtype=torch.bool)
        if dbg.get("margin") is not None:
            fl |= dbg["margin"] < eps_prob
        if "logit_at" in dbg:
            fl |= dbg["logit_at"].abs() < eps_logit
        zero = dbg["zero_before_fill"]
        if zero.any():
            if fl.any():
                # a flipped D1 / D2 may move a pixel into or out of the set the in-view fill copies from: every filled point of
                # this view is then open (conservative)
                fl = fl | zero
            elif "fill_src" in dbg:
                fl[zero] |= fl[dbg["fill_src"]]
        flag[v["pt"][fl]] = True
    if not views:
        return flag
    _, dbg = lift.fuse_views_top3(N, [v["pt"] for v in views], fs, lgs, xyz32, return_debug=True)
    if "class_margin" in dbg:
        flag |= dbg["class_margin"] < eps_fuse
        flag |= dbg["cut_margin"] < eps_fuse
    seen = dbg["seen"]
    if "fill_src" in dbg:
        flag[~seen] |= flag[dbg["fill_src"]]
    return flag


def class_near_ties(features, text, scale, eps=EPS_CLASS):
    """D5 on the oracle's pooled features: bool [N], top-2 logit margin below eps (fp64 logits of the normalised rows)."""
    f = F.normalize(torch.as_tensor(features).double(), dim=-1)
    t = F.normalize(torch.as_tensor(text).double(), dim=-1)
    lg = float(scale) * f @ t.t()
    if lg.shape[1] < 2:
        return torch.zeros(lg.shape[0], dtype=torch.bool)
    top2 = lg.topk(2, dim=1).values
    return (top2[:, 0] - top2[:, 1]) < eps


def check_scene(ref, got_features, got_lifted, scene, vlm, sd, rigid, oracle_kwargs, tol=1e-4, tol_lift=1e-5, masks=True):
    """The rule of the module docstring.  ref: evaluate_scene_oracle(...) of the same scene; got_*: the device's tensors.
    -> dict(lift_mismatches, lift_near_ties, rerun, max_diff) (the tests print it).  Raises AssertionError."""
    got_features = got_features.detach().cpu().float()
    got_lifted = got_lifted.detach().cpu().float()
    N = got_lifted.shape[0]
    d_l = (got_lifted - ref["lifted"]).abs().max(dim=1).values
    bad = d_l > tol_lift
    info = {"lift_mismatches": int(bad.sum()), "lift_near_ties": 0, "rerun": False}
    if masks:
        xyz32 = torch.from_numpy(scene.coords).float()
        near = lift_near_ties(ref["views"], vlm, xyz32, scene.cfg.mask_shape, N)
        info["lift_near_ties"] = int(near.sum())
        stray = bad & ~near
        assert not stray.any(), (f"{int(stray.sum())} lifted rows differ from the oracle by more than {tol_lift} without a decision inside "
                                 f"fp32 rounding noise (first: point {int(stray.nonzero()[0])}, diff {float(d_l[stray].max()):.3e})")
    else:
        assert not bad.any(), f"{int(bad.sum())} lifted rows differ by more than {tol_lift} (max {float(d_l.max()):.3e}); this lift has no decisions"
    target = ref
    if bad.any():
        target = o_pipe.evaluate_scene_oracle(scene, vlm, sd, rigid, lifted=got_lifted, **oracle_kwargs)
        info["rerun"] = True
    d = (got_features - target["scene_features"]).abs().max(dim=1).values
    info["max_diff"] = float(d.max())
    assert float(d.max()) < tol, (f"pooled features: {int((d >= tol).sum())} of {N} points outside {tol} (max {float(d.max()):.3e}); "
                                  f"lift mismatches {info['lift_mismatches']}, oracle re-run from the device's lift: {info['rerun']}")
    info["target"] = target
    return info


def check_labels(pred, target, eps=EPS_CLASS):
    """D5: device class decisions against the oracle's (target: the dict check_scene compared the features with).
    Every disagreement must have an oracle top-2 margin below eps.  -> (mismatches, near ties)"""
    ref_pred, _ = metric.classify(target["scene_features"], target["text_features"], target["logit_scale"])
    near = class_near_ties(target["scene_features"], target["text_features"], target["logit_scale"], eps)
    mism = torch.as_tensor(pred).cpu().long() != ref_pred
    stray = mism & ~near
    assert not stray.any(), f"{int(stray.sum())} class decisions differ from the oracle's with a top-2 margin above {eps}"
    return int(mism.sum()), int(near.sum())
