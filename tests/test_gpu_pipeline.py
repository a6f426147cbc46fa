"""End-to-end parity of the device hot path (loader math -> lift -> student -> kNN -> pooling ->
classify/IoU) against the CPU oracle, through the reference-shaped host API."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import metric as o_metric  # noqa: E402
from oracle import parity as o_parity  # noqa: E402
from oracle import pipeline as o_pipe  # noqa: E402


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from geopurify_amd import pipeline as pl
    from geopurify_amd import synthetic as syn
    cfg = syn.CONFIGS["T"]
    scene = syn.make_scene(cfg, 321)
    vlm_np = syn.make_vlm_outputs(cfg, cfg.num_views, 321)
    sd = pl.random_student_state_dict(cfg.feat_dim + pl.GEO_DIM, hidden=128, embed=128, num_blocks=2, seed=4)
    rigid = pl.scene_rigid_transform(cfg.voxel_size, 321)
    K, T = 32, 4
    ref = o_pipe.evaluate_scene_oracle(scene, vlm_np, sd, rigid, K=K, num_iters=T)
    batch = pl.build_scene_batch(pl.upload_scene(scene, "cuda"), rigid, "cuda")
    hp = pl.HotPath(pl.StudentWeights(sd, "cuda"), cfg.mask_shape, K=K, num_iters=T, device="cuda")
    return dict(pl=pl, syn=syn, cfg=cfg, scene=scene, vlm_np=vlm_np, sd=sd, rigid=rigid, ref=ref, batch=batch,
                hp=hp, K=K, T=T)


def test_loader_math_bit_exact(env):
    b, ref = env["batch"], env["ref"]
    assert np.array_equal(b.scene_coords_3d.cpu().numpy(), ref["coords_3d"].astype(np.float32))
    assert torch.equal(b.scene_inds_reconstruct.cpu(), ref["inv"])
    assert len(b.views) == len(ref["views"]) > 0
    for v, r in zip(b.views, ref["views"]):
        assert v.src_view == r["src_view"]
        assert torch.equal(v.pt.cpu(), r["pt"]) and torch.equal(v.x.cpu(), r["x"]) and torch.equal(v.y.cpu(), r["y"])


def test_lift_stage(env):
    """Rows 5-7: every lifted row within 1e-5 of the oracle's, except rows that hang on a decision (segment arg-max, the 0.5 test,
    consensus class, top-3 cut) whose ORACLE margin is inside fp32 rounding noise (oracle/parity.py) -- expected: none."""
    hp, b, pl = env["hp"], env["batch"], env["pl"]
    F, text, scale = hp.lift_masks(b, pl.SyntheticVLM(env["vlm_np"], "cuda"))
    d = (F.cpu() - env["ref"]["lifted"]).abs().max(dim=1).values
    near = o_parity.lift_near_ties(env["ref"]["views"], env["vlm_np"], torch.from_numpy(env["scene"].coords).float(),
                                   env["cfg"].mask_shape, F.shape[0])
    bad = d > 1e-5
    print(f"lift: {int(bad.sum())} rows outside 1e-5, {int(near.sum())} near-tie points")
    assert not (bad & ~near).any(), (int((bad & ~near).sum()), float(d.max()))
    assert d.median() < 1e-6


def test_refine_stage_given_oracle_lift(env):
    """rows 8-12 in isolation: feed the ORACLE's lifted features, compare every downstream product."""
    hp, b, ref = env["hp"], env["batch"], env["ref"]
    out = hp.refine(b, ref["lifted"].cuda().contiguous())
    d = (out.cpu() - ref["scene_features"]).abs().max()
    assert d < 1e-4, d                                    # north_star: pooled features within 1e-4 fp32


@pytest.fixture(scope="module")
def env512(env):
    """The same tiny scene with 512-d features (the shape the tiled and matrix-core pooling kernels serve)."""
    import dataclasses
    pl, syn = env["pl"], env["syn"]
    cfg = dataclasses.replace(env["cfg"], feat_dim=512)
    vlm_np = syn.make_vlm_outputs(cfg, cfg.num_views, 322)
    sd = pl.random_student_state_dict(cfg.feat_dim + pl.GEO_DIM, hidden=128, embed=128, num_blocks=1, seed=5)
    K, T = 32, 4
    ref = o_pipe.evaluate_scene_oracle(env["scene"], vlm_np, sd, env["rigid"], K=K, num_iters=T)
    return dict(cfg=cfg, sd=sd, ref=ref, K=K, T=T)


@pytest.mark.parametrize("pool_mode,block_rows", [("auto", 64), ("mfma_cs", 64), ("mfma_chain", 64), ("mfma_engine", 64), ("mfma", 64), ("mfma", 128), ("mfma_persist", 64),
                                                   ("tiles", 64), ("ell", 64)])
def test_refine_every_pooling_kernel_d512(env, env512, pool_mode, block_rows):
    """rows 8-12 at D = 512 through each pooling kernel (matrix-core, tiled, ELL): same oracle, same tolerance."""
    pl, b, ref = env["pl"], env["batch"], env512["ref"]
    hp = pl.HotPath(pl.StudentWeights(env512["sd"], "cuda"), env512["cfg"].mask_shape, K=env512["K"],
                    num_iters=env512["T"], device="cuda", pool_mode=pool_mode, pool_block_rows=block_rows)
    out = hp.refine(b, ref["lifted"].cuda().contiguous())
    d = (out.cpu() - ref["scene_features"]).abs().max()
    assert d < 1e-4, d                                    # north_star: pooled features within 1e-4 fp32
    # K = 32 on this tiny scene: row blocks may have fewer than 4 steps, in which case "mfma_persist" falls back to the
    # one-tile-per-workgroup kernel (the persistent kernel is covered at K = 96 by test_gpu_reference_golden / fullsize)
    hp.pool_chain_check()
    want = {"auto": ("cs_pool_kernel",), "mfma_cs": ("cs_pool_kernel",), "mfma_chain": ("cs_chain_kernel",), "mfma_engine": ("cs_engine_kernel",), "mfma": ("pool_mfma_kernel",), "mfma_persist": ("pool_mfma_persist_kernel", "pool_mfma_kernel"),
            "tiles": ("pool_tiles_kernel",), "ell": ("pool_ell_kernel",)}
    assert hp.stats["pool_kernel"] in want[pool_mode]


def test_refine_with_a_state_prepared_ahead_on_another_stream(env, env512):
    """HotPath.prepare run ahead on a side stream (bench.py's look-ahead) and handed to refine: the same bits as refine alone,
    and the after_student hook is called exactly once between the student and the affinity kernels."""
    pl, b, ref = env["pl"], env["batch"], env512["ref"]
    hp = pl.HotPath(pl.StudentWeights(env512["sd"], "cuda"), env512["cfg"].mask_shape, K=env512["K"],
                    num_iters=env512["T"], device="cuda")
    F = ref["lifted"].cuda().contiguous()
    plain = hp.refine(b, F).clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        state = hp.prepare(b, F)
        done = torch.cuda.Event()
        done.record(side)
    assert set(state) >= {"X", "rank", "nbr_map", "pairs", "nbr", "Nv", "D", "pool"} and state["pool"] is not None
    torch.cuda.current_stream().wait_event(done)
    calls = []
    ahead = hp.refine(b, F, after_student=lambda: calls.append(hp.stats.get("pool_kernel")), prepared=state)
    assert calls and len(calls) == 1
    assert torch.equal(ahead, plain)
    assert (ahead.cpu() - ref["scene_features"]).abs().max() < 1e-4


def _oracle_kwargs(env):
    return dict(K=env["K"], num_iters=env["T"])


def test_end_to_end_features_and_labels(env):
    """The whole path against the oracle with NO blanket allowance (oracle/parity.py): lifted rows may differ only at decisions
    inside fp32 rounding noise; pooled features within 1e-4 at EVERY point (of the oracle re-run from the device's lift if such a
    decision went the other way); every class decision that differs must have an oracle top-2 margin below 1e-4."""
    hp, b, pl, cfg, ref = env["hp"], env["batch"], env["pl"], env["cfg"], env["ref"]
    hp.keep_lifted = True
    try:
        res = hp.evaluate_scene(b, pl.SyntheticVLM(env["vlm_np"], "cuda"))
    finally:
        hp.keep_lifted = False
    info = o_parity.check_scene(ref, res["scene_features"], hp.last_lifted, env["scene"], env["vlm_np"], env["sd"], env["rigid"],
                                _oracle_kwargs(env))
    counts = torch.zeros((3, cfg.num_classes), dtype=torch.int64, device="cuda")
    pred, zero = hp.classify_and_count(res, b.scene_label, cfg.num_classes, cfg.ignore_ids, counts)
    mism, near = o_parity.check_labels(pred, info["target"])
    print(f"end to end: lift mismatches {info['lift_mismatches']} (near ties {info['lift_near_ties']}), max |d| {info['max_diff']:.2e}, "
          f"label mismatches {mism} (near ties {near})")
    rpred, (ri, ru, rt) = o_pipe.classify_and_count(info["target"], env["scene"].labels, cfg.num_classes, cfg.ignore_ids)
    assert np.array_equal(counts[2].cpu().numpy(), rt)     # target histogram is input-only: exact
    if mism == 0:                                          # no class near tie flipped: the IoU bookkeeping is exact
        cn = counts.cpu().numpy()                          # rows: intersection, predicted area, target area
        assert np.array_equal(cn[0], ri) and np.array_equal(cn[1] + cn[2] - cn[0], ru)
    assert not zero.any()


def test_reference_api_tuple_path(env):
    """SonataXAffinityTrainer.evaluate_scene on the reference's positional 20-tuple (CPU tensors, as the DataLoader hands
    them over) against the ORACLE, and bit-identical to the device-built SceneBatch path."""
    from geopurify_amd.affinity_module import SonataXAffinityTrainer
    pl, cfg, ref = env["pl"], env["cfg"], env["ref"]
    model = SonataXAffinityTrainer({"mask_shape": list(cfg.mask_shape), "all_label": ["c"] * cfg.num_classes},
                                   None, None, device="cuda", use_lseg=False, vlm=pl.SyntheticVLM(env["vlm_np"], "cuda"),
                                   feature_dim=cfg.feat_dim, embed_dim=128, hidden_dim=128)
    sd = dict(env["sd"])
    # the trainer always builds 4 residual blocks (affinity_module.py:60-65): blocks 2-3 get weights that make them the
    # identity on non-negative inputs (zero kernels, BN = identity), so the 2-block oracle result applies
    for i in (2, 3):
        for k in list(sd):
            if k.startswith("res_blocks.1."):
                v = sd[k].clone()
                nk = k.replace("res_blocks.1.", f"res_blocks.{i}.")
                if k.endswith("kernel") or k.endswith("bn.bias") or k.endswith("running_mean"):
                    v = torch.zeros_like(v)
                elif k.endswith("bn.weight") or k.endswith("running_var"):
                    v = torch.ones_like(v)
                sd[nk] = v
    model.affinity_student.load_state_dict(sd)
    model.K, model.num_pool_iters = env["K"], env["T"]
    groups = model.affinity_student.get_param_groups()
    assert set(groups) == {"input", "middle", "output"} and len(groups["output"]) == 1
    tup = tuple(t.cpu() if torch.is_tensor(t) else t for t in env["batch"].as_tuple())
    assert len(tup) == 20
    model._hot_path().keep_lifted = True
    bres = model.evaluate_scene(tup)
    # vs the oracle, north-star tolerance at every point (oracle/parity.py: no blanket allowance)
    o_parity.check_scene(ref, bres["scene_features"], model._hot_path().last_lifted, env["scene"], env["vlm_np"], env["sd"],
                         env["rigid"], _oracle_kwargs(env))
    assert (bres["text_features"].cpu() - ref["text_features"]).abs().max() <= 1e-6   # normalised, as the reference returns them
    a = model.evaluate_scene(env["batch"])["scene_features"]
    assert torch.equal(a, bres["scene_features"])                                  # tuple path == SceneBatch path
    assert bres["text_features"].shape == (cfg.num_classes, cfg.feat_dim)
    # the look-ahead form (offer_next: a loader wrapper hands over the NEXT scene's tuple; its copy / parse / lift / prepare run on the
    # trainer's side stream beside this scene's student): three scenes in a row, every result bit-identical to the serial call
    tup2 = tuple(t.clone().pin_memory() if torch.is_tensor(t) else t for t in tup)
    seq = [tup, tup2, tup]
    outs = []
    for i, t in enumerate(seq):
        if i + 1 < len(seq):
            model.offer_next(seq[i + 1])
        outs.append(model.evaluate_scene(t)["scene_features"].clone())
    torch.cuda.synchronize()
    assert model._ahead is None and model._offered is None
    for o in outs:
        assert torch.equal(o, bres["scene_features"])
    # ... and through the loader wrapper a user puts around the reference's DataLoader (copies ahead on its own stream, offers the device tuple)
    from geopurify_amd.data_loader import LookAheadLoader
    n_seen = 0
    for d in LookAheadLoader([tup2, tup, tup2, tup], model):
        assert all(x.is_cuda for x in d if torch.is_tensor(x) and x.numel())
        assert torch.equal(model.evaluate_scene(d)["scene_features"], bres["scene_features"])
        n_seen += 1
    assert n_seen == 4 and model._ahead is None and model._offered is None
    # an offered scene that is NOT the next one evaluated is dropped, not confused with it
    model.offer_next(tup2)
    assert torch.equal(model.evaluate_scene(tup)["scene_features"], bres["scene_features"])      # (lifts tup2 ahead ...)
    assert torch.equal(model.evaluate_scene(tup)["scene_features"], bres["scene_features"])      # (... which this call ignores)
    assert model._ahead is None


def test_dataset_sampler_collate_drive_evaluate_scene():
    """The reference driver's data path (run/validation.py:296-321,408): ScannetLoaderFull -> SceneBatchSampler ->
    DataLoader(collate_fn=scene_based_collate_fn) -> model.evaluate_scene(batch_data), here over a synthetic scene, with
    the 2D stand-in reading the view from the IMAGE it is handed (slot 11).  Checked against the oracle run on the same
    scene with the scene-level voxel grid the loader drew."""
    import dataclasses
    from geopurify_amd import pipeline as pl, synthetic as syn
    from geopurify_amd.affinity_module import SonataXAffinityTrainer
    from geopurify_amd.data_loader import ScannetLoaderFull, SceneBatchSampler, scene_based_collate_fn
    from oracle import affinity as o_aff, lift as o_lift, student as o_student
    cfg = syn.CONFIGS["T"]
    np.random.seed(77)
    ds = ScannetLoaderFull("synthetic:T:2:400", None, label_2d=list(range(20)), category_split=None, voxel_size=cfg.voxel_size,
                           split="val", eval_all=True, specific_ids=["scene0001"])
    assert len(ds.data_paths) == 1 and len(ds.samples) == cfg.num_views and ds.samples[0]["scene_name"] == "scene0001_00"
    loader = torch.utils.data.DataLoader(ds, num_workers=0, collate_fn=scene_based_collate_fn,
                                         batch_sampler=SceneBatchSampler(ds.samples, shuffle=False))
    batches = list(loader)
    assert len(batches) == 1 and len(batches[0]) == 20
    tup = batches[0]
    N = cfg.num_points
    V = tup[11].shape[0]
    assert tup[0].shape == (N, 3) and tup[14].shape == (V * N, 2) and tup[11].shape[1:] == (cfg.mask_shape[0], cfg.mask_shape[1], 3)
    assert int(tup[4][:, 0].max()) == V - 1 and tup[12].shape == tup[13].shape == (int(tup[14][:, 1].sum()),)
    scene = ds._scene("scene0001_00")
    vlm_np = syn.make_vlm_outputs(cfg, cfg.num_views, 401)
    sd = pl.random_student_state_dict(cfg.feat_dim + pl.GEO_DIM, hidden=128, embed=128, num_blocks=4, seed=6)
    model = SonataXAffinityTrainer({"mask_shape": list(cfg.mask_shape), "all_label": ["c"] * cfg.num_classes}, None, None,
                                   device="cuda", use_lseg=False, vlm=pl.SyntheticVLM(vlm_np, "cuda", index_from_image=True),
                                   feature_dim=cfg.feat_dim, embed_dim=128, hidden_dim=128)
    model.affinity_student.load_state_dict(sd)
    model.K, model.num_pool_iters = 24, 3
    model._hot_path().keep_lifted = True
    res = model.evaluate_scene(tup)
    # oracle on the tuple's own lists (the loader math itself is pinned elsewhere): lift -> mean -> student -> kNN -> pool
    xyz = tup[0]
    text = torch.from_numpy(vlm_np["text_embed"])
    fs, lgs, pts = [], [], []
    view_of = tup[4][:, 0].long()
    for i in range(V):
        sel = view_of == i
        src = int(tup[11][i, 0, 0, 0])                      # the generator's view index, carried by the image
        pt = torch.where(tup[14][i * N:(i + 1) * N, 1] == 1)[0]
        f, lg = o_lift.lift_masks_view(torch.from_numpy(vlm_np["pred_masks"][src]), torch.from_numpy(vlm_np["pred_logits"][src]),
                                       torch.from_numpy(vlm_np["mask_embed"][src]), text, float(vlm_np["logit_scale"]),
                                       tup[12][sel], tup[13][sel], xyz[pt], cfg.mask_shape)
        fs.append(f), lgs.append(lg), pts.append(pt)
    Fp = o_lift.fuse_views_top3(N, pts, fs, lgs, xyz)
    inv, coords = tup[2], tup[1].numpy().astype(np.int64)
    X = torch.cat([o_aff.scatter_mean(Fp, inv, coords.shape[0]), o_aff.scatter_mean(tup[19][:, :6], inv, coords.shape[0])], 1)
    E = o_student.student_forward(X, coords, sd, num_blocks=4)
    nbr = o_aff.knn_lattice(coords, 24)
    Y = o_aff.pool_sparse(X, nbr, o_aff.affinity_weights(E, nbr, 20.0), 3)[inv][:, :cfg.feat_dim]
    d = (res["scene_features"].cpu() - Y).abs().max(dim=1).values
    # lifted rows: only decisions inside fp32 rounding noise may differ (oracle/parity.py); with none flipped, every point within 1e-4
    got_lift = model._hot_path().last_lifted.cpu()
    bad = (got_lift - Fp).abs().max(dim=1).values > 1e-5
    views = [{"src_view": int(tup[11][i, 0, 0, 0]), "pt": pts[i], "x": tup[12][view_of == i], "y": tup[13][view_of == i]} for i in range(V)]
    near = o_parity.lift_near_ties(views, vlm_np, xyz, cfg.mask_shape, N)
    assert not (bad & ~near).any(), int((bad & ~near).sum())
    if bad.any():                                          # a near tie went the other way: the downstream oracle from the device's lift
        X = torch.cat([o_aff.scatter_mean(got_lift, inv, coords.shape[0]), o_aff.scatter_mean(tup[19][:, :6], inv, coords.shape[0])], 1)
        E = o_student.student_forward(X, coords, sd, num_blocks=4)
        Y = o_aff.pool_sparse(X, nbr, o_aff.affinity_weights(E, nbr, 20.0), 3)[inv][:, :cfg.feat_dim]
        d = (res["scene_features"].cpu() - Y).abs().max(dim=1).values
    assert d.max() < 1e-4, (float(d.max()), int(bad.sum()))


def test_dense_lift_config(env):
    """P-style config: dense feature maps (row 5) through the same refine path."""
    pl, syn = env["pl"], env["syn"]
    import dataclasses
    cfg = dataclasses.replace(syn.CONFIGS["T"], num_views=2, feat_dim=32)
    scene = syn.make_scene(cfg, 55)
    feat = syn.make_dense_feature_maps(cfg, cfg.num_views, 55)
    text = np.random.default_rng(1).normal(size=(cfg.num_classes, cfg.feat_dim)).astype(np.float32)
    sd = pl.random_student_state_dict(cfg.feat_dim + pl.GEO_DIM, hidden=128, embed=128, num_blocks=1, seed=9)
    rigid = pl.scene_rigid_transform(cfg.voxel_size, 55)
    ref = o_pipe.evaluate_scene_oracle(scene, {"text_embed": text, "logit_scale": 14.0}, sd, rigid, K=16, num_iters=2,
                                       dense_feat=feat)
    batch = pl.build_scene_batch(pl.upload_scene(scene, "cuda"), rigid, "cuda")
    hp = pl.HotPath(pl.StudentWeights(sd, "cuda"), cfg.mask_shape, K=16, num_iters=2, device="cuda")
    res = hp.evaluate_scene(batch, pl.DenseFeatureVLM(feat, text, 14.0, "cuda"))
    assert (res["scene_features"].cpu() - ref["scene_features"]).abs().max() < 1e-4


def test_validation_driver_two_scenes(tmp_path):
    """the evaluation driver end to end on two tiny synthetic scenes; counts vs the oracle per scene"""
    import os
    from geopurify_amd import validation, pipeline as pl, synthetic as syn
    from geopurify_amd.affinity_module import SonataXAffinityTrainer
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = validation.get_parser(["--config", os.path.join(root, "config", "geopurify_synthetic_scannet.yaml"),
                                  "save_path", str(tmp_path / "o"), "synthetic_config", "T", "num_scenes", "2",
                                  "mask_shape", "[120, 160]", "pool_iters", "3"])
    cfg = syn.CONFIGS["T"]
    model = SonataXAffinityTrainer(args, None, None, device="cuda", use_lseg=False, feature_dim=cfg.feat_dim,
                                   embed_dim=128, hidden_dim=128, allow_deferred_vlm=True)
    model.K, model.num_pool_iters = 24, 3
    sd = model.affinity_student.state_dict()
    hp = model._hot_path()
    state, total = {}, np.zeros((3, 19), np.int64)
    scenes, refs = [], []
    for sid in (0, 1):
        seed = 5557 + sid
        scene = syn.make_scene(cfg, seed)
        vlm_np = syn.make_vlm_outputs(cfg, cfg.num_views, seed)
        rigid = pl.scene_rigid_transform(cfg.voxel_size, seed)
        ref = o_pipe.evaluate_scene_oracle(scene, vlm_np, sd, rigid, K=24, num_iters=3)
        refs.append(ref)
        _, (ri, ru, rt) = o_pipe.classify_and_count(ref, scene.labels, 19, [19, 20])
        total += np.stack([ri, ru + ri - rt, rt])

        def provider(scene=scene, vlm_np=vlm_np, rigid=rigid):
            state["vlm"] = pl.SyntheticVLM(vlm_np, "cuda")
            return pl.build_scene_batch(pl.upload_scene(scene, "cuda"), rigid, "cuda")
        scenes.append((f"scannet_synthetic_{sid:04d}", provider))

    def evaluate(batch, sid):
        model.vlm = state["vlm"]
        return model.evaluate_scene(batch, vis_prefix=sid)

    (base, novel), counts = validation.validate(scenes, evaluate, args, hp)
    c = counts.cpu().numpy()
    assert np.array_equal(c[2], total[2])                              # target histogram exact
    # predictions: exact wherever the oracle's own arg-max margin exceeds what a 1e-4 feature difference can move a logit;
    # the counts may then differ from the oracle's by at most the points inside that margin
    unsafe = 0
    for (sid, provider), r in zip(scenes, refs):
        batch = provider()
        res = evaluate(batch, sid)
        tmp = torch.zeros((3, 19), dtype=torch.int64, device="cuda")
        pred = validation.scene_tail(hp, res, batch.scene_coords, batch.scene_label, 19, [19, 20], tmp).cpu()
        rp, logits = o_metric.classify(r["scene_features"], r["text_features"], r["logit_scale"])
        top2 = logits.topk(2, dim=1).values
        feat_ok = (res["scene_features"].cpu() - r["scene_features"]).abs().max(dim=1).values < 1e-4
        safe = ((top2[:, 0] - top2[:, 1]) > 5e-3) & feat_ok
        assert torch.equal(pred[safe], rp[safe])
        unsafe += int((pred != rp).sum())
        assert (~safe).float().mean() < 0.02
    assert np.abs(c[0] - total[0]).sum() <= unsafe and np.abs(c[1] - total[1]).sum() <= 2 * unsafe
    assert 0.0 <= base <= 1.0 and 0.0 <= novel <= 1.0


def test_fused_feature_loader_on_disk_formats(tmp_path):
    """dataset.feature_loader surface: .pth scene + fused-feature .pt (2-key and 3-key forms), eval and train
    splits, checked against the oracle voxelizer run with the same seed."""
    import os
    from geopurify_amd.feature_loader import FusedFeatureLoader, collation_fn_eval_all
    from oracle import voxelize as o_vox
    rng = np.random.default_rng(21)
    for split in ("val", "train"):
        d3 = tmp_path / f"root_{split}" / "scannet_3d"      # dataset_name must be exactly "scannet_3d" (feature_loader.py:92)
        os.makedirs(d3 / split)
        os.makedirs(tmp_path / f"feat_{split}")
        N, D = 5000, 16
        locs = rng.uniform(0, 2.5, size=(N, 3)).astype(np.float32)
        locs[:, 2] *= 0.05
        cols = rng.uniform(-1, 1, size=(N, 3)).astype(np.float32)
        labs = rng.integers(0, 20, size=N).astype(np.float64)
        labs[:50] = -100
        torch.save((locs, cols, labs), d3 / split / "scene0001_00_vh_clean_2.pth")
        mask_full = torch.from_numpy(rng.random(N) < 0.6)
        feat = torch.randn(int(mask_full.sum()), D)
        torch.save({"feat": feat, "mask_full": mask_full}, tmp_path / f"feat_{split}" / "scene0001_00_0.pt")
        ds = FusedFeatureLoader(str(d3), str(tmp_path / f"feat_{split}"), voxel_size=0.05, split=split, eval_all=True,
                                input_color=True)
        np.random.seed(3)
        coords, feats, labels, feat_3d, mask, inv = ds[0]
        # oracle with the same RNG stream
        np.random.seed(3)
        M_v, M_r = o_vox.get_transformation_matrix(0.05, True)
        c, inds, oinv, _ = o_vox.voxelize_with_matrices(locs.astype(np.float64), M_v, M_r, True)
        assert torch.equal(coords[:, 1:], torch.from_numpy(c).int()) and (coords[:, 0] == 1).all()
        assert torch.equal(inv, torch.from_numpy(oinv))
        assert torch.allclose(feats, torch.from_numpy(((cols + 1.0) * 127.5)[inds]).float() / 127.5 - 1.0)
        full = torch.zeros(N, D)
        full[mask_full] = feat
        if split == "val":
            assert torch.equal(feat_3d, full[torch.from_numpy(inds)]) and torch.equal(mask, mask_full[torch.from_numpy(inds)])
        else:
            rep = torch.from_numpy(inds)
            assert torch.equal(mask, mask_full[rep]) and torch.equal(feat_3d, full[rep[mask_full[rep]]])
        assert labels.shape[0] == N and int((labels == 255).sum()) == 50       # eval_all returns the per-point labels
        b = collation_fn_eval_all([(coords.clone(), feats, labels, feat_3d, mask, inv.clone()),
                                   (coords.clone(), feats, labels, feat_3d, mask, inv.clone())])
        assert (b[0][:coords.shape[0], 0] == 0).all() and (b[0][coords.shape[0]:, 0] == 1).all()
        assert int(b[5][N:].min()) == coords.shape[0]


def test_all_views_loader_and_lift_equal_the_view_by_view_path(env):
    """The all-views launches (gp_views_visible_lists, gp_lift_masks_views) reproduce the view-by-view path bit for bit:
    entry lists incl. the view-drop rule, segments after the in-view fill (through the fused features, which read the
    point -> (view, segment) lists in view order), on a scene where some views are dropped and one with 70 views
    (second word of the per-point view mask)."""
    import dataclasses
    pl, syn = env["pl"], env["syn"]
    cases = [(dataclasses.replace(syn.CONFIGS["T"], num_points=9000, num_views=7, min_visible=1), 77),
             (dataclasses.replace(syn.CONFIGS["T"], num_points=3000, num_views=70), 78),
             (syn.CONFIGS["T"], 321)]
    for cfg, seed in cases:
        scene = pl.upload_scene(syn.make_scene(cfg, seed), "cuda")
        rigid = pl.scene_rigid_transform(cfg.voxel_size, seed)
        if cfg.min_visible == 1:                                        # drop the views below the median visible count
            sizes = sorted(len(v.pt) for v in pl.build_scene_batch(scene, rigid, "cuda", batch_views=False).views)
            cfg = dataclasses.replace(cfg, min_visible=sizes[len(sizes) // 2])
            scene.cfg = cfg
        b_all = pl.build_scene_batch(scene, rigid, "cuda")
        b_one = pl.build_scene_batch(scene, rigid, "cuda", batch_views=False)
        assert b_all.ent is not None and b_one.ent is None
        assert [v.src_view for v in b_all.views] == [v.src_view for v in b_one.views] and len(b_all.views) > 0
        if cfg.num_views == 7:
            assert 0 < len(b_all.views) < cfg.num_views                 # the drop rule was exercised
        for va, vo in zip(b_all.views, b_one.views):
            assert torch.equal(va.pt, vo.pt) and torch.equal(va.x, vo.x) and torch.equal(va.y, vo.y)
        assert b_all.extent == b_one.extent
        vlm = pl.SyntheticVLM(syn.make_vlm_outputs(cfg, cfg.num_views, seed), "cuda")
        sd = pl.random_student_state_dict(cfg.feat_dim + pl.GEO_DIM, hidden=128, embed=128, num_blocks=1, seed=1)
        st = pl.StudentWeights(sd, "cuda")
        F_all, t_all, _ = pl.HotPath(st, cfg.mask_shape, K=16, num_iters=1, device="cuda").lift_masks(b_all, vlm)
        F_one, t_one, _ = pl.HotPath(st, cfg.mask_shape, K=16, num_iters=1, device="cuda", batch_views=False).lift_masks(b_all, vlm)
        F_pv, _, _ = pl.HotPath(st, cfg.mask_shape, K=16, num_iters=1, device="cuda").lift_masks(b_one, vlm)
        assert torch.equal(F_all, F_one) and torch.equal(F_all, F_pv) and torch.equal(t_all, t_one)
        assert F_all.abs().sum() > 0
