#!/usr/bin/env python3
"""Golden vectors from the reference's OWN validate() (run/validation.py:343-559), build container only.

Run:  python tests/golden/make_golden_validate.py     (needs /root/reference; never runs on the GPU box)

run/validation.py imports tensorboardX, imageio, MinkowskiEngine, cv2, open3d, omegaconf, xdecoder, detectron2 and
models.affinity_module at module level (:15-40); placeholder modules are registered for those names (none of them is
touched by validate()).  validate() is then called as is with
  * a stand-in loader (an iterable of 20-tuples with .dataset.data_paths),
  * a stand-in model whose evaluate_scene returns given (scene_features, text_features, logit_scale) -- the hot
    path's OUTPUT is the input of this row,
  * the module globals it reads (args, logger) set the way main_worker sets them.
What runs is the reference's text: normalise / classify / arg-max (:413-416), the zero-row nearest fill with the
[:, 1:4] slice of the [N,3] coordinates (:417-432, sklearn KDTree), intersectionAndUnionGPU (util/util.py:160-177, with
Tensor.cuda() made a no-op on this GPU-less box), the Base/Novel/All meters and the log strings (:452-553).
Only inputs and outputs are stored.
"""
import importlib.machinery
import importlib.util
import logging
import os
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)


def _placeholder(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Unused:
    def __init__(self, *a, **k):
        raise RuntimeError("placeholder executed")


_placeholder("tensorboardX", SummaryWriter=_Unused)
_placeholder("imageio")
_placeholder("MinkowskiEngine", SparseTensor=_Unused)
_placeholder("cv2")
_placeholder("open3d")
_placeholder("omegaconf", OmegaConf=_Unused)
_placeholder("models.utils.visualization", visualize_2d_semantic=_Unused, get_color_palette=_Unused, save_3d_point_cloud=_Unused)
for _n in ("xdecoder", "xdecoder.utils", "xdecoder.modeling", "detectron2", "detectron2.utils"):
    _placeholder(_n)
_placeholder("xdecoder.utils.arguments", load_opt_from_config_files=_Unused)
_placeholder("detectron2.utils.memory", retry_if_cuda_oom=_Unused)
_placeholder("xdecoder.modeling.modules", sem_seg_postprocess=_Unused)
_placeholder("models.affinity_module", SonataXAffinityTrainer=_Unused)

spec = importlib.util.spec_from_file_location("ref_run_validation", os.path.join(REF, "run", "validation.py"))
ref_val = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_val)

from util import config as ref_config  # noqa: E402  (the reference's config loader)


class _Capture(logging.Handler):
    def __init__(self):
        super().__init__()
        self.lines = []

    def emit(self, record):
        self.lines.append(record.getMessage())


def make_scene(rng, N, D, C, zero_frac):
    xyz = rng.uniform(0, 6, size=(N, 3)).astype(np.float32)
    text = rng.normal(0, 1, size=(C, D)).astype(np.float32)
    labels = rng.integers(0, C + 2, size=N).astype(np.int64)            # includes the ignore ids C, C+1
    # features correlated with the label's text embedding so that the counts are not trivial
    lab = np.minimum(labels, C - 1)
    f = (0.15 * text[lab] + rng.normal(0, 1, size=(N, D))).astype(np.float32) * 0.05
    zero = rng.random(N) < zero_frac
    f[zero] = 0.0
    return xyz, labels, f, text, zero


def main():
    rng = np.random.default_rng(77)
    args = ref_config.load_cfg_from_cfg_file(os.path.join(REF, "config", "geopurify_scannet.yaml"))
    args.multiprocessing_distributed = False
    C = args.test_classes
    scenes = [make_scene(rng, 2600, 64, C, 0.04), make_scene(rng, 1900, 64, C, 0.015), make_scene(rng, 1500, 64, C, 0.0)]
    logit_scale = torch.tensor(float(np.exp(np.log(1 / 0.07))))

    def tup(s):
        xyz, labels, f, text, zero = s
        e = torch.zeros(0)
        return (torch.from_numpy(xyz), e, e, torch.from_numpy(labels), e, e, e, e, e, e, e, e, e, e, e, e, e, e, (None,), e)

    class _Loader(list):
        dataset = types.SimpleNamespace(data_paths=[f"/data/scene{i:04d}_00_vh_clean_2.pth" for i in range(len(scenes))])

    class _Model:
        def __init__(self):
            self.i = 0

        def eval(self):
            return self

        def evaluate_scene(self, batch_data, vis_prefix=None):
            s = scenes[self.i]
            self.i += 1
            return {"scene_features": torch.from_numpy(s[2]), "text_features": torch.from_numpy(s[3]), "logit_scale": logit_scale}

    preds = []
    real_iou = ref_val.intersectionAndUnionGPU

    def spy(output, target, K, ignore):
        preds.append(output.clone().numpy())
        return real_iou(output, target, K, ignore)

    cap = _Capture()
    lg = logging.getLogger("golden-validate")
    lg.setLevel(logging.INFO)
    lg.addHandler(cap)
    lg.propagate = False
    ref_val.args, ref_val.logger = args, lg
    ref_val.intersectionAndUnionGPU = spy
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self                        # no GPU in the build container
    cwd = os.getcwd()
    try:
        with tempfile.TemporaryDirectory() as tmp:
            os.chdir(tmp)                                                # validate() creates xdecoder_test/paper under cwd
            result = ref_val.validate(_Loader([tup(s) for s in scenes]), _Model(), None)
    finally:
        os.chdir(cwd)
        torch.Tensor.cuda = real_cuda
    out = {"num_scenes": np.int64(len(scenes)), "logit_scale": np.float32(logit_scale), "test_classes": np.int64(C),
           "test_ignore_label": np.array(args.test_ignore_label), "base_category": np.array(args.category_split["base_category"]),
           "novel_category": np.array(args.category_split["novel_category"]),
           "result": np.array(result, dtype=np.float64), "log_lines": np.array(cap.lines)}
    for i, s in enumerate(scenes):
        out[f"s{i}_coords"], out[f"s{i}_label"], out[f"s{i}_features"], out[f"s{i}_text"] = s[0], s[1], s[2], s[3]
        out[f"s{i}_zero"] = s[4]
        out[f"s{i}_pred"] = preds[i]
    np.savez_compressed(os.path.join(HERE, "ref_validate.npz"), **out)
    print("validate():", result, "zero rows per scene:", [int(s[4].sum()) for s in scenes], "log lines:", len(cap.lines))
    for ln in cap.lines[-9:]:
        print("  ", ln[:150])
    print("ref_validate.npz", os.path.getsize(os.path.join(HERE, "ref_validate.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
