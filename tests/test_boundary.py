"""CPU tests of the drop-in boundary: every in-scope name the reference drivers import resolves under compat/,
the collate mirror reproduces the reference's own scene_based_collate_fn, the VLM hook receives the view's image, and the
student's device-weight cache can never go stale."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_in_scope_driver_import_resolves_under_compat(golden_dir):
    """tests/golden/reference_driver_imports.txt = names imported by run/validation.py and run/train.py (ast, names only)."""
    rows = [l.split() for l in open(os.path.join(golden_dir, "reference_driver_imports.txt")) if l.strip()]
    need = sorted({(m, n) for _, m, n, scope in rows if scope == "in"})
    assert len(need) >= 20
    code = ["import importlib"]
    for m, n in need:
        code.append(f"mod = importlib.import_module({m!r})")
        if n != "-":                                        # `from m import n`: attribute, else submodule
            code.append(f"_ = getattr(mod, {n!r}) if hasattr(mod, {n!r}) else importlib.import_module({m!r} + '.' + {n!r})")
    code.append("print('ok')")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "compat"), ROOT]))
    out = subprocess.run([sys.executable, "-c", "\n".join(code)], env=env, capture_output=True, text=True, timeout=180)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr[-2000:]
    ext = sorted({m.split(".")[0] for _, m, _, scope in rows if scope == "ext"})
    assert ext == ["cv2", "detectron2", "imageio", "omegaconf", "open3d", "tensorboardX", "xdecoder"]     # documented as outside the library


def test_collate_matches_reference(golden_dir):
    """scene_based_collate_fn against the reference's own output (tests/golden/ref_collate.npz, data_loader_ablation.py:429-495)."""
    from geopurify_amd.data_loader import SceneBatchSampler, scene_based_collate_fn
    g = np.load(os.path.join(golden_dir, "ref_collate.npz"))
    views = []
    for i in range(int(g["num_views"])):
        if bool(g[f"in{i}_none"]):
            views.append(None)
            continue
        views.append(tuple(torch.from_numpy(g[f"in{i}_{j}"].copy()) if f"in{i}_{j}" in g.files else None for j in range(20)))
    out = scene_based_collate_fn(views)
    assert len(out) == 20 and out[18] == (None, None, None)
    for j in range(20):
        if j == 18:
            continue
        ref = g[f"out_{j}"]
        assert tuple(out[j].shape) == ref.shape and str(out[j].numpy().dtype) == str(ref.dtype), j
        assert np.array_equal(out[j].numpy(), ref), j
    assert scene_based_collate_fn([None, None]) is None
    s = SceneBatchSampler([{"scene_name": n} for n in ["a", "a", "b", "a", "c", "b"]], shuffle=False)
    assert [str(b) for b in s] == [str(x) for x in g["sampler_batches"]] and len(s) == 3


def test_tuple_keeps_images_and_vlm_sees_them():
    """The reference runs the 2D model on imgs[view_idx] (affinity_module.py:496,518-519): slot 11 must reach the hook."""
    from geopurify_amd.affinity_module import ForwardSegAllVLM, SonataXAffinityTrainer
    N, V, H, W = 50, 3, 12, 16
    e = torch.zeros(0)
    mask = torch.zeros((V, N), dtype=torch.long)
    mask[:, :10] = 1
    tup = (torch.randn(N, 3), torch.zeros(7, 3), torch.zeros(N, dtype=torch.long), torch.zeros(N, dtype=torch.long),
           torch.cat([torch.cat([torch.full((10, 1), float(i)), torch.randn(10, 3)], 1) for i in range(V)]), e, e, e, e, e, e,
           torch.stack([torch.full((H, W, 3), float(10 + i)) for i in range(V)]), torch.zeros(30, dtype=torch.long),
           torch.zeros(30, dtype=torch.long), torch.stack([torch.arange(V).repeat_interleave(N), mask.reshape(-1)], 1), e, e, e,
           (None,) * V, torch.rand(N, 6))
    b = SonataXAffinityTrainer._batch_from_tuple(tup, "cpu")
    assert b.imgs is not None and tuple(b.imgs.shape) == (V, H, W, 3) and [v.src_view for v in b.views] == [0, 1, 2]
    assert torch.equal(b.as_tuple()[11], tup[11])

    calls = []

    class Model:
        def forward_seg_all(self, inputs):
            calls.append((tuple(inputs[0]["image"].shape), float(inputs[0]["image"][0, 0, 0, 0]), inputs[0]["height"], inputs[0]["width"]))
            return None, {"pred_masks": torch.zeros(1, 4, 3, 4), "pred_logits": torch.zeros(1, 4, 6), "mask_embed": torch.ones(1, 4, 8),
                          "text_embed": torch.ones(5, 8), "logit_scale": torch.tensor(14.0)}

    vlm = ForwardSegAllVLM(Model(), (H, W), device="cpu")
    out = vlm(1, image=b.imgs[1])
    assert calls == [((1, 3, H, W), 11.0, H, W)]
    assert tuple(out["pred_masks"].shape) == (4, 3, 4) and tuple(out["mask_embed"].shape) == (4, 8)
    with pytest.raises(ValueError):
        vlm(0)


def test_vlm_factory_and_constructor_contract():
    """run/validation.py:166 calls SonataXAffinityTrainer(args, xdecoder_cfg, scene_config, device, False): with a registered
    factory that call attaches the VLM; without one it fails AT CONSTRUCTION with an actionable message."""
    from geopurify_amd import affinity_module as am
    cfg = {"mask_shape": [12, 16], "all_label": ["a", "b"]}
    old = am.register_vlm_factory(None)
    try:
        with pytest.raises(RuntimeError, match="register_vlm_factory"):
            am.SonataXAffinityTrainer(cfg, {"x": 1}, None, "cuda", False)
        seen = {}

        def factory(c, xcfg, device, use_lseg):
            seen.update(cfg=c, xcfg=xcfg, device=device, use_lseg=use_lseg)
            return "the-vlm"

        am.register_vlm_factory(factory)
        m = am.SonataXAffinityTrainer(cfg, {"x": 1}, None, "cuda", False)
        assert m.vlm == "the-vlm" and seen == {"cfg": cfg, "xcfg": {"x": 1}, "device": "cuda", "use_lseg": False}
        assert isinstance(m, torch.nn.Module) and set(m.affinity_student.get_param_groups()) == {"input", "middle", "output"}
    finally:
        am.register_vlm_factory(old)
    m2 = am.SonataXAffinityTrainer(cfg, None, None, "cuda", False, allow_deferred_vlm=True)
    assert m2.vlm is None


def test_device_weight_cache_key_follows_every_parameter_change():
    """evaluate -> optimizer.step() -> evaluate must not reuse folded weights.  The cache key is the tuple of
    parameter / buffer versions, which every in-place update bumps."""
    from geopurify_amd.affinity_module import AffinityPredictor
    m = AffinityPredictor(38, embed_dim=8, hidden_dim=16)
    k0 = m._weights_key("cuda:0")
    assert m._weights_key("cuda:0") == k0 and m._weights_key("cuda:1") != k0
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    k1 = m._weights_key("cuda:0")
    assert k1 != k0
    with torch.no_grad():
        m.input_layer[1].bn.running_mean.add_(1.0)               # buffers count too (BatchNorm statistics are folded)
    assert m._weights_key("cuda:0") != k1
    m._dev_weights = object()
    m.train()
    assert m._dev_weights is None                              # entering training mode drops the copy
    m._dev_weights = object()
    m.load_state_dict(m.state_dict())
    assert m._dev_weights is None


def test_util_helpers(tmp_path):
    from geopurify_amd import util
    util.save_checkpoint({"a": torch.ones(2)}, True, str(tmp_path))
    assert os.path.isfile(tmp_path / "model_last.pth.tar") and os.path.isfile(tmp_path / "model_best.pth.tar")
    util.export_pointcloud(str(tmp_path / "p.ply"), torch.rand(1, 5, 3), colors=np.random.rand(5, 3), normals=torch.rand(1, 5, 3))
    txt = open(tmp_path / "p.ply").read().splitlines()
    assert txt[0] == "ply" and "element vertex 5" in txt and len(txt) == txt.index("end_header") + 6
    pal = util.get_palette()
    assert len(pal) == 63 and all(isinstance(v, int) and 0 <= v <= 255 for v in pal)
    assert len(util.get_palette(colormap="scannet_200")) == 603
