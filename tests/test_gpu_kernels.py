"""GPU parity tests: every HIP kernel, called through the C-ABI, against the CPU oracle on the same
seeded inputs (sizes the oracle finishes in seconds) and against the committed golden fixtures.
Bar: bit-exact for integer / index outputs, stated tolerances for fp32."""
import os

import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import affinity as o_aff  # noqa: E402
from oracle import lift as o_lift  # noqa: E402
from oracle import metric as o_metric  # noqa: E402
from oracle import project as o_proj  # noqa: E402
from oracle import student as o_student  # noqa: E402
from oracle import voxelize as o_vox  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from geopurify_amd import ops as _ops
    from geopurify_amd import _lib
    _lib.load()                      # fails loudly if the HIP library is missing
    return _ops


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a)) if not torch.is_tensor(a) else a
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


def surface_voxels(rng, n=4000, ext=60):
    a = np.c_[rng.integers(0, ext, n), rng.integers(0, ext, n), rng.integers(3, 5, n)]
    b = np.c_[rng.integers(0, ext, n // 2), np.full(n // 2, 17), rng.integers(0, 40, n // 2)]
    c = np.c_[rng.integers(0, ext, n // 2), (rng.integers(0, ext, n // 2) * 0.6).astype(int), np.zeros(n // 2, int)]
    c[:, 2] = (c[:, 0] * 0.5).astype(int) + 6                     # oblique sheet
    iso = np.array([[ext + 200, 5, 5], [ext + 330, 90, 41], [ext + 331, 90, 41]])   # isolated voxels
    v = np.unique(np.vstack([a, b, c, iso]), axis=0)
    return v[rng.permutation(len(v))].astype(np.int32)


# ------------------------------------------------------------------------------------------ rows 1-2
@pytest.mark.parametrize("case", [0, 1, 2])
def test_voxelize_golden(ops, golden_dir, case):
    g = np.load(os.path.join(golden_dir, "voxelize.npz"))
    p = f"c{case}_"
    rigid = g[p + "M_r"] @ g[p + "M_v"] if bool(g[p + "aug"]) else g[p + "M_v"]
    r = ops.voxelize(dev(g[p + "points"]), rigid)
    assert r["nv"] == len(g[p + "inds"])
    assert np.array_equal(r["coords_aug"].cpu().numpy(), g[p + "coords_aug"])
    assert np.array_equal(r["inds"].cpu().numpy(), g[p + "inds"])
    assert np.array_equal(r["inds_reconstruct"].cpu().numpy(), g[p + "inds_reconstruct"])
    # CSR consistency: every point of segment v maps to voxel v, ids ascending inside a segment
    order, seg = r["order"].cpu().numpy(), r["seg_start"].cpu().numpy()
    inv = g[p + "inds_reconstruct"]
    assert np.array_equal(inv[order], np.repeat(np.arange(r["nv"]), np.diff(seg)))
    for v in (0, r["nv"] // 2, r["nv"] - 1):
        s = order[seg[v]:seg[v + 1]]
        assert (np.diff(s) > 0).all() and s[0] == g[p + "inds"][v]


def test_fnv_golden(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, "fnv_hash.npz"))
    h = ops.fnv_hash(dev(g["coords"])).cpu().numpy().view(np.uint64)
    assert np.array_equal(h, g["hash"])


def test_voxelize_vs_oracle_large(ops):
    from geopurify_amd import synthetic as syn
    sc = syn.make_scene(syn.CONFIGS["T"], 77)
    np.random.seed(5)
    M_v, M_r = o_vox.get_transformation_matrix(0.02, True)
    c, inds, inv, rigid = o_vox.voxelize_with_matrices(sc.coords, M_v, M_r, True)
    r = ops.voxelize(dev(sc.coords), rigid)
    assert np.array_equal(r["coords_aug"].cpu().numpy(), c)
    assert np.array_equal(r["inds"].cpu().numpy(), inds)
    assert np.array_equal(r["inds_reconstruct"].cpu().numpy(), inv)


# ------------------------------------------------------------------------------------------ row 3
def test_project_golden_scannet(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, "mapping.npz"))
    W, H = (int(v) for v in g["sn_image_dim"])
    K = g["sn_K"]
    w2c = g["sn_wvt"].T.astype(np.float64)
    m, wgt = ops.project_points(dev(g["sn_points"]), w2c, K[0, 0], K[1, 1], K[0, 2], K[1, 2], dev(g["sn_depth"]),
                                W, H, int(g["sn_cut"]), float(g["sn_tau"]), want_weight=True)
    assert np.array_equal(m.cpu().numpy(), g["sn_mapping"])
    assert np.allclose(wgt.cpu().numpy(), g["sn_weight"], rtol=1e-12, atol=0)
    m2 = ops.project_points(dev(g["sn_points"]), w2c, K[0, 0], K[1, 1], K[0, 2], K[1, 2], None, W, H,
                            int(g["sn_cut"]), float(g["sn_tau"]))
    assert np.array_equal(m2.cpu().numpy(), g["sn_mapping_nodepth"])
    # "render" mode: z-buffer of the cloud (atomicMin on the fp64 bit pattern), then the same occlusion test
    from geopurify_amd.fusion_util import PointCloudToImageMapper
    mapper = PointCloudToImageMapper((W, H), float(g["sn_tau"]), int(g["sn_cut"]), g["sn_K_native"])
    m3, w3 = mapper.compute_mapping(g["sn_wvt"], g["sn_points_render"], "render")
    assert np.array_equal(m3, g["sn_mapping_render"])
    pr = dev(g["sn_points_render"])
    d = ops.render_depth(pr, w2c, K[0, 0], K[1, 1], K[0, 2], K[1, 2], W, H, int(g["sn_cut"]))
    from oracle import project as o_proj
    p_, pi_ = o_proj._project(w2c, g["sn_points_render"], K)
    assert np.array_equal(d.cpu().numpy(), o_proj.render_depth(p_, pi_, (W, H), int(g["sn_cut"])))


def test_project_golden_edges_and_matterport(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, "mapping.npz"))
    W, H = (int(v) for v in g["ex_image_dim"])
    K = g["ex_K"]
    m = ops.project_points(dev(g["ex_points"]), np.eye(4), K[0, 0], K[1, 1], K[0, 2], K[1, 2], dev(g["ex_depth"]),
                           W, H, 10, 0.05)
    assert np.array_equal(m.cpu().numpy(), g["ex_mapping"])
    W, H = (int(v) for v in g["mp_image_dim"])
    K = g["mp_K"]
    w2c = np.linalg.inv(g["mp_c2w"])            # host mirror of fusion_util.py:60 (inverse of the fp32 matrix)
    m = ops.project_points(dev(g["mp_points"]), w2c, K[0, 0], K[1, 1], K[0, 2], K[1, 2], dev(g["mp_depth"]), W, H,
                           int(g["mp_cut"]), float(g["mp_tau"]))
    assert np.array_equal(m.cpu().numpy(), g["mp_mapping"])


# ------------------------------------------------------------------------------------------ order / grid / kernel map
def _sorted_voxels(ops, c):
    ct = dev(c)
    perm, rank = ops.morton_order(ct)
    cs = ct[perm.long()].contiguous()
    grid = ops.grid_build(cs)
    assert grid.status() == 0
    return ct, perm, rank, cs, grid


def test_morton_grid_kernel_map(ops):
    rng = np.random.default_rng(0)
    c = surface_voxels(rng)
    ct, perm, rank, cs, grid = _sorted_voxels(ops, c)
    p = perm.cpu().numpy()
    assert np.array_equal(np.sort(p), np.arange(len(c)))
    assert np.array_equal(rank.cpu().numpy()[p], np.arange(len(c)))
    nm = ops.kernel_map_build(grid, cs).cpu().numpy()
    ref = o_student.build_kernel_map(cs.cpu().numpy())
    assert np.array_equal(nm, ref)
    # unsorted input must be flagged, not silently accepted
    bad = ops.grid_build(ct)
    assert bad.status() != 0


# ------------------------------------------------------------------------------------------ row 10
@pytest.mark.parametrize("K", [96, 7])
def test_knn_exact_with_ties(ops, K):
    rng = np.random.default_rng(1)
    c = surface_voxels(rng, 3000)
    ct, perm, rank, cs, grid = _sorted_voxels(ops, c)
    nbr_int = ops.knn_lattice(grid, cs, perm, K)                     # rows of the sorted arrays, ids = reference rows
    torch.cuda.synchronize()
    # back to reference order: nbr_ref[perm[i], j] = perm[nbr_int[i, j]]
    nbr_ref = torch.empty_like(nbr_int)
    nbr_ref[perm.long()] = perm[nbr_int.long()]
    ref = o_aff.knn_lattice(c, K).numpy()
    assert np.array_equal(nbr_ref.cpu().numpy(), ref)                # bit exact incl. (d2, id) order


def test_knn_sparse_fallback(ops):
    """far-apart clusters force the ring-3 and exhaustive paths"""
    rng = np.random.default_rng(2)
    pts = []
    for cx in range(0, 900, 60):
        pts.append(np.c_[rng.integers(0, 3, 9) + cx, rng.integers(0, 3, 9), rng.integers(0, 3, 9) + (cx // 7)])
    c = np.unique(np.vstack(pts), axis=0).astype(np.int32)
    c = c[rng.permutation(len(c))]
    K = 20
    ct, perm, rank, cs, grid = _sorted_voxels(ops, c)
    nbr_int = ops.knn_lattice(grid, cs, perm, K)
    nbr_ref = torch.empty_like(nbr_int)
    nbr_ref[perm.long()] = perm[nbr_int.long()]
    assert np.array_equal(nbr_ref.cpu().numpy(), o_aff.knn_lattice(c, K).numpy())


# ------------------------------------------------------------------------------------------ rows 8, 11, 12
def test_scatter_mean_gather_bit_exact(ops):
    rng = np.random.default_rng(3)
    N, D = 5000, 64
    pts = rng.uniform(0, 1.2, size=(N, 3))
    r = ops.voxelize(dev(pts), np.diag([50.0, 50.0, 50.0, 1.0]))
    Fp = torch.randn(N, D)
    nv = r["nv"]
    out = torch.zeros((nv, D + 8), dtype=torch.float32, device="cuda")
    ops.scatter_mean_csr(dev(Fp), D, r["order"], r["seg_start"], nv, out, col0=0)
    geo = torch.rand(N, 6)
    ops.scatter_mean_csr(dev(geo), 6, r["order"], r["seg_start"], nv, out, col0=D)
    inv = r["inds_reconstruct"].cpu()
    ref = o_aff.scatter_mean(Fp, inv, nv)
    assert torch.equal(out[:, :D].cpu(), ref)                        # same summation order -> bit exact
    assert torch.equal(out[:, D:D + 6].cpu(), o_aff.scatter_mean(geo, inv, nv))
    g = ops.gather_rows(out, D, r["inds_reconstruct"])
    assert torch.equal(g.cpu(), ref[inv])
    # with a row map (internal order)
    perm = torch.randperm(nv)
    rank = torch.empty_like(perm)
    rank[perm] = torch.arange(nv)
    out2 = torch.zeros((nv, D), dtype=torch.float32, device="cuda")
    ops.scatter_mean_csr(dev(Fp), D, r["order"], r["seg_start"], nv, out2, row_map=dev(rank, torch.int32))
    assert torch.equal(out2.cpu()[rank], ref)
    g2 = ops.gather_rows(out2, D, r["inds_reconstruct"], row_map=dev(rank, torch.int32))
    assert torch.equal(g2.cpu(), ref[inv])


def test_affinity_and_pooling(ops):
    rng = np.random.default_rng(4)
    c = surface_voxels(rng, 2500)
    Nv, K, D = len(c), 96, 512
    nbr = o_aff.knn_lattice(c, K)
    E = F.normalize(torch.randn(Nv, 128), dim=1)
    w_ref = o_aff.affinity_weights(E, nbr, 20.0)
    w = ops.affinity_softmax(dev(E), dev(nbr, torch.int32), 20.0)
    assert (w.cpu() - w_ref).abs().max() < 2e-6                      # fp32 softmax, tolerance 2e-6 absolute
    X = torch.randn(Nv, D + 32)
    Xd = dev(X)
    bufs = [torch.empty((Nv, D), device="cuda"), torch.empty((Nv, D), device="cuda")]
    cur = Xd
    T = 5
    for t in range(T):
        ops.pool_ell(cur, dev(nbr, torch.int32), dev(w_ref), D, bufs[t % 2])
        cur = bufs[t % 2]
    ref64 = o_aff.pool_gather(X[:, :D], nbr, w_ref, T)
    ref32 = o_aff.pool_sparse(X[:, :D].contiguous(), nbr, w_ref, T)
    err = (cur.cpu().double() - ref64).abs().max().item()
    assert err < 1e-4, err                                           # north_star tolerance: 1e-4 fp32
    assert (cur.cpu() - ref32).abs().max() < 1e-4


def test_affinity_block_form_is_bit_identical_to_the_wave_form(ops):
    """The block kernel (distinct neighbour rows of 16 rows staged once in LDS) keeps the per-row arithmetic order of the
    one-wave-per-row kernel: same bits.  Covers the LDS path (Morton-adjacent rows share neighbours), the global fallback
    (random neighbours: the union of 16 rows exceeds the LDS capacity), K = 20 / 96 / 128, a ragged last block, duplicate
    neighbour ids inside a row and padded embedding rows (ld_e > 128)."""
    from geopurify_amd._lib import load
    lib = load()
    rng = np.random.default_rng(41)
    c = surface_voxels(rng, 3001)
    ct, perm, rank, cs, grid = _sorted_voxels(ops, c)
    cases = []
    for K in (20, 96, 128):
        nbr = torch.as_tensor(o_aff.knn_lattice(cs.cpu().numpy(), K)).to(torch.int32)
        cases.append((K, nbr, "morton"))
    Nv = len(c)
    cases.append((96, torch.from_numpy(rng.integers(0, Nv, (Nv, 96))).to(torch.int32), "random"))
    dup = torch.from_numpy(rng.integers(0, 7, (Nv, 33))).to(torch.int32)
    cases.append((33, dup, "duplicates"))
    g = torch.Generator().manual_seed(3)
    Epad = torch.randn(Nv, 160, generator=g)
    Epad[:, :128] = F.normalize(Epad[:, :128], dim=1)
    E = dev(Epad)[:, :128]                                            # row stride 160 floats
    for K, nbr, name in cases:
        n = dev(nbr)
        w_block = ops.affinity_softmax(E, n, 20.0)                    # default: 16 rows per workgroup
        try:
            lib.gp_debug_set(15, 1)
            w_wave = ops.affinity_softmax(E, n, 20.0)
            lib.gp_debug_set(15, 2)                                   # 8 rows per workgroup
            w_block8 = ops.affinity_softmax(E, n, 20.0)
        finally:
            lib.gp_debug_set(15, 0)
        assert torch.equal(w_block, w_wave) and torch.equal(w_block8, w_wave), (name, K)
        w_ref = o_aff.affinity_weights(Epad[:, :128].contiguous(), nbr.long(), 20.0)
        assert (w_block.cpu() - w_ref).abs().max() < 2e-6, (name, K)


# ------------------------------------------------------------------------------------------ row 9
def _bn_fold(sd, prefix):
    s = sd[prefix + ".bn.weight"] / torch.sqrt(sd[prefix + ".bn.running_var"] + 1e-5)
    return s, sd[prefix + ".bn.bias"] - sd[prefix + ".bn.running_mean"] * s


def test_sparse_conv_single_layer(ops):
    rng = np.random.default_rng(5)
    c = surface_voxels(rng, 1500)
    ct, perm, rank, cs, grid = _sorted_voxels(ops, c)
    nm = ops.kernel_map_build(grid, cs)
    Nv = len(c)
    X = torch.randn(Nv, 64)
    W = torch.randn(27, 64, 128) * 0.05
    y = ops.sparse_conv(dev(X), nm, dev(W))
    ref = o_student.sparse_conv3(X.double(), nm.cpu().numpy().astype(np.int64), W.double())
    assert (y.cpu().double() - ref).abs().max() < 2e-5               # fp32 accumulation, K<=27*64
    # epilogue: scale/shift + residual + relu, kv=1 path
    sc, sh = torch.rand(128) + 0.5, torch.randn(128)
    res = torch.randn(Nv, 128)
    W1 = torch.randn(64, 128) * 0.1
    y2 = ops.sparse_conv(dev(X), None, dev(W1), dev(sc), dev(sh), dev(res), relu=True)
    ref2 = torch.relu((X.double() @ W1.double()) * sc.double() + sh.double() + res.double())
    assert (y2.cpu().double() - ref2).abs().max() < 2e-5


def test_student_forward_vs_oracle(ops):
    """whole student (input conv + 2 res blocks + linear + l2norm) at reduced width"""
    rng = np.random.default_rng(6)
    c = surface_voxels(rng, 2000)
    ct, perm, rank, cs, grid = _sorted_voxels(ops, c)
    nm = ops.kernel_map_build(grid, cs)
    Nv = len(c)
    cin, hid, emb, nb = 96, 128, 128, 2
    sd = o_student.random_student_state_dict(cin, hidden=hid, embed=emb, num_blocks=nb, seed=1)
    X = torch.randn(Nv, cin)
    Xs = X[perm.cpu().long()]                                         # internal (Morton) order
    ref = o_student.student_forward(Xs, cs.cpu().numpy(), sd, num_blocks=nb, dtype=torch.float64)
    g = {k: dev(v) for k, v in sd.items() if v.is_floating_point()}
    s, b = _bn_fold(sd, "input_layer.1")
    h = ops.sparse_conv(dev(Xs), nm, g["input_layer.0.kernel"], dev(s), dev(b), relu=True)
    for i in range(nb):
        s1, b1 = _bn_fold(sd, f"res_blocks.{i}.norm1")
        s2, b2 = _bn_fold(sd, f"res_blocks.{i}.norm2")
        t = ops.sparse_conv(h, nm, g[f"res_blocks.{i}.conv1.kernel"], dev(s1), dev(b1), relu=True)
        h = ops.sparse_conv(t, nm, g[f"res_blocks.{i}.conv2.kernel"], dev(s2), dev(b2), residual=h, relu=True)
    e = ops.sparse_conv(h, None, g["output_layer.kernel"])
    ops.l2norm_rows_(e)
    assert (e.cpu().double() - ref).abs().max() < 1e-5               # unit-norm embeddings, fp32 vs fp64 oracle


# ------------------------------------------------------------------------------------------ rows 5-7
def test_lift_dense(ops):
    rng = np.random.default_rng(7)
    N, D, H, W, V = 3000, 64, 40, 56, 3
    xyz = torch.from_numpy(rng.normal(size=(N, 3)).astype(np.float32))
    feats, pis, xs, ys = [], [], [], []
    for v in range(V):
        pi = np.sort(rng.choice(N - 100, 900, replace=False))
        feats.append(torch.from_numpy(rng.uniform(-1, 1, size=(D, H, W)).astype(np.float32)))
        pis.append(torch.from_numpy(pi)), xs.append(torch.from_numpy(rng.integers(0, H, 900)))
        ys.append(torch.from_numpy(rng.integers(0, W, 900)))
    ref, seen_ref = o_lift.lift_dense(feats, pis, xs, ys, xyz)
    s = torch.zeros((N, D), device="cuda")
    cnt = torch.zeros(N, device="cuda")
    for v in range(V):
        ops.lift_dense_accum(dev(feats[v]), dev(pis[v]), dev(xs[v]), dev(ys[v]), s, cnt)
    seen = ops.lift_dense_finish(s, D, cnt).bool()
    assert torch.equal(seen.cpu(), seen_ref)
    nn = ops.nn1(dev(xyz)[seen].contiguous(), dev(xyz)[~seen].contiguous())
    s[~seen] = s[seen][nn]
    assert torch.equal(s.cpu(), ref)                                 # same order of fp32 adds -> bit exact


def test_lift_lseg_bilinear_sampling(ops):
    """LSeg path: the bilinear(align_corners=True) resize evaluated only at the sampled pixels equals
    torch's CPU resize + lift to the bit (same fp32 operation order, fma form pinned in the kernel)."""
    rng = np.random.default_rng(17)
    N, D, h, w, H, W, V = 3000, 64, 24, 32, 61, 83, 3                  # odd output size: non-trivial fractional taps
    xyz = torch.from_numpy(rng.normal(size=(N, 3)).astype(np.float32))
    feats, pis, xs, ys = [], [], [], []
    for v in range(V):
        pi = np.sort(rng.choice(N - 100, 900, replace=False))
        feats.append(torch.from_numpy(rng.normal(size=(D, h, w)).astype(np.float32)))
        pis.append(torch.from_numpy(pi))
        x = rng.integers(0, H, 900); y = rng.integers(0, W, 900)
        x[:4] = [0, H - 1, 0, H - 1]; y[:4] = [0, 0, W - 1, W - 1]      # the four corners (clamped taps)
        xs.append(torch.from_numpy(x)), ys.append(torch.from_numpy(y))
    ref, seen_ref = o_lift.lift_lseg(feats, (H, W), pis, xs, ys, xyz)
    s = torch.zeros((N, D), device="cuda")
    cnt = torch.zeros(N, device="cuda")
    for v in range(V):
        ops.lift_dense_bilinear_accum(dev(feats[v]), H, W, dev(pis[v]), dev(xs[v]), dev(ys[v]), s, cnt)
    seen = ops.lift_dense_finish(s, D, cnt).bool()
    assert torch.equal(seen.cpu(), seen_ref)
    nn = ops.nn1(dev(xyz)[seen].contiguous(), dev(xyz)[~seen].contiguous())
    s[~seen] = s[seen][nn]
    d = (s.cpu() - ref).abs().max().item()
    assert d <= 1e-6, d                                               # tolerance 1e-6 (fp32); bit-exact where the host
    assert (s.cpu() == ref).float().mean() > 0.999                    # libm/fma conventions agree (they do on x86-64 torch CPU)


def test_nn1_exact(ops):
    rng = np.random.default_rng(8)
    ref = rng.normal(size=(20000, 3)).astype(np.float32)
    q = rng.normal(size=(3000, 3)).astype(np.float32)
    q[:10] = ref[100:110]                                            # exact hits
    nn = ops.nn1(dev(ref), dev(q)).cpu().numpy()
    assert np.array_equal(nn, o_lift.nn1_indices(ref, q))
    assert np.array_equal(nn, o_lift.nn1_indices_bruteforce(ref, q))


def test_lift_masks_view_and_fuse(ops):
    from geopurify_amd import synthetic as syn
    from geopurify_amd.bicubic import aa_bicubic_taps
    cfg = syn.CONFIGS["T"]
    V = 3
    vlm = syn.make_vlm_outputs(cfg, V, 11)
    rng = np.random.default_rng(9)
    N = 2500
    H, W = cfg.mask_shape
    Q, h, w = vlm["pred_masks"].shape[1:]
    xyz = torch.from_numpy(rng.normal(size=(N, 3)).astype(np.float32))
    tx0, twx = aa_bicubic_taps(w, W)
    ty0, twy = aa_bicubic_taps(h, H)
    taps = (dev(tx0), dev(twx), dev(ty0), dev(twy))
    text = torch.from_numpy(vlm["text_embed"])
    tn = F.normalize(text, dim=-1)
    ls = float(vlm["logit_scale"])
    C, D = text.shape
    f_seg = torch.empty((V, Q, D), device="cuda")
    l_seg = torch.empty((V, Q, C), device="cuda")
    pis, fs, lgs = [], [], []
    cnt = torch.zeros(N, dtype=torch.int64, device="cuda")
    segs = []
    n_margin = 0
    for v in range(V):
        pi = torch.from_numpy(np.sort(rng.choice(N - 60, 800, replace=False)))
        x = torch.from_numpy(rng.integers(10, H - 10, 800))
        y = torch.from_numpy(rng.integers(10, W - 10, 800))
        pm, pl, me = (torch.from_numpy(vlm[k][v]) for k in ("pred_masks", "pred_logits", "mask_embed"))
        f, lg, dbg = o_lift.lift_masks_view(pm, pl, me, text, ls, x, y, xyz[pi], cfg.mask_shape,
                                            explicit_resize=True, return_debug=True)
        scores = o_lift.segment_scores(pl)[0]
        seg, seg_logit = ops.lift_masks_view(dev(pm), dev(scores), taps, (H, W), dev(x), dev(y), want_logit=True)
        seg = seg.cpu()
        # resized logit at the winning segment: bit exact with the explicit (torch-CPU-order) resize
        same = seg >= 0
        zero_ref = dbg["zero_before_fill"]
        exp_seg = torch.where(zero_ref, torch.full_like(dbg["seg"], -1), dbg["seg"])
        mism = (seg.long() != exp_seg)
        # index side: exact wherever the oracle's own arg-max margin exceeds fp32 rounding noise
        safe = dbg["margin"] > 1e-6
        assert not (mism & safe).any()
        n_margin += int(mism.sum())
        ok = ~mism & same
        assert torch.equal(seg_logit.cpu()[ok], dbg["logit_at"][ok])
        # in-view fill + tables
        z = seg < 0
        segd = seg.cuda()
        if z.any():
            xv = dev(xyz[pi])
            nn = ops.nn1(xv[~z.cuda()].contiguous(), xv[z.cuda()].contiguous())
            src = torch.where(~z)[0].cuda()[nn]
            assert torch.equal(src.cpu()[~mism[z]], dbg["fill_src"][~mism[z]]) or mism.any()
            segd[z.cuda()] = segd[src]
        ops.segment_tables(dev(me), dev(tn), ls, f_seg[v], l_seg[v])
        if not mism.any():
            assert (f_seg[v][segd.long()].cpu() - f).abs().max() < 1e-6
            assert (l_seg[v][segd.long()].cpu() - lg).abs().max() < 2e-4      # logits ~ +-14, fp32 dot order
        ops.pv_count(dev(pi), cnt)
        pis.append(pi), fs.append(f), lgs.append(lg), segs.append(segd)
    assert n_margin <= 2
    start = ops.exclusive_scan_i64(torch.cat([cnt, cnt.new_zeros(1)]))
    total = int(start[-1].item())
    assert total == sum(len(p) for p in pis)
    cursor = torch.zeros(N, dtype=torch.int32, device="cuda")
    pvv = torch.empty(total, dtype=torch.int32, device="cuda")
    pvs = torch.empty(total, dtype=torch.int32, device="cuda")
    for v in range(V):
        ops.pv_fill(dev(pis[v]), segs[v], v, start, cursor, pvv, pvs)
    out = torch.empty((N, D), device="cuda")
    seen = ops.fuse_views_top3(start, pvv, pvs, N, f_seg, l_seg, out).bool()
    ref, dbg = o_lift.fuse_views_top3(N, pis, fs, lgs, xyz, return_debug=True)
    assert torch.equal(seen.cpu(), dbg["seen"])
    if n_margin == 0:
        d = (out.cpu()[dbg["seen"]] - ref[dbg["seen"]]).abs().max(dim=1).values
        # fused features are convex combinations of unit vectors; a row outside 1e-5 must hang on a consensus-class or top-3-cut
        # decision whose ORACLE margin is inside the rounding noise of the logits (1e-4 on +-14): no blanket allowance
        near = ((dbg["class_margin"] < 1e-4) | (dbg["cut_margin"] < 1e-4))[dbg["seen"]]
        assert not ((d >= 1e-5) & ~near).any(), (int(((d >= 1e-5) & ~near).sum()), float(d.max()))
        assert d.median() < 1e-6
    xyzd = dev(xyz)
    nn = ops.nn1(xyzd[seen].contiguous(), xyzd[~seen].contiguous())
    assert torch.equal(torch.where(seen)[0][nn].cpu(), dbg["fill_src"])


# ------------------------------------------------------------------------------------------ row 13
def test_classify_and_iou(ops):
    rng = np.random.default_rng(10)
    N, D, C = 6000, 512, 19
    Fp = torch.randn(N, D)
    Fp[:50] = 0
    text = torch.randn(C, D)
    tn = F.normalize(text, dim=-1)
    pred, zero = ops.classify_argmax(dev(Fp), dev(tn), 14.285)
    ref_pred, logits = o_metric.classify(Fp, text, 14.285)
    top2 = logits.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 1e-4
    assert torch.equal(pred.cpu()[safe], ref_pred[safe]) and safe.float().mean() > 0.99
    assert torch.equal(zero.cpu().bool(), Fp.abs().sum(1) == 0)
    # the kernel with the text matrix staged in LDS (default) keeps the arithmetic of the plain one: identical labels,
    # also on a ragged point count and a padded feature stride
    from geopurify_amd._lib import load
    lib = load()
    for n_pts, pad in ((N, 0), (N - 37, 32)):
        Fd = dev(torch.cat([Fp[:n_pts], torch.zeros(n_pts, pad)], 1)) if pad else dev(Fp[:n_pts])
        a = ops.classify_argmax(Fd, dev(tn), 14.285, d=D)
        lib.gp_debug_set(14, 1)
        try:
            b = ops.classify_argmax(Fd, dev(tn), 14.285, d=D)
        finally:
            lib.gp_debug_set(14, 0)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    # large class count: logits by the exact-fp32 MFMA GEMM, same decisions
    text160 = torch.randn(160, D)
    tn160 = F.normalize(text160, dim=-1)
    p160, z160 = ops.classify_argmax_gemm(dev(Fp), dev(tn160))
    r160, lg160 = o_metric.classify(Fp, text160, 14.285)
    t2 = lg160.topk(2, dim=1).values
    safe160 = (t2[:, 0] - t2[:, 1]) > 1e-4
    assert torch.equal(p160.cpu()[safe160], r160[safe160]) and safe160.float().mean() > 0.99
    assert torch.equal(z160.cpu().bool(), Fp.abs().sum(1) == 0)
    tgt = torch.from_numpy(rng.integers(0, 21, N))
    counts = torch.zeros((3, C), dtype=torch.int64, device="cuda")
    ops.iou_hist(dev(ref_pred), dev(tgt), C, [19, 20], counts)
    i, u, t = o_metric.intersection_and_union(ref_pred.numpy(), tgt.numpy(), C, [19, 20])
    cn = counts.cpu().numpy()
    assert np.array_equal(cn[0], i) and np.array_equal(cn[1] + cn[2] - cn[0], u) and np.array_equal(cn[2], t)


def test_iou_golden(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, "iou.npz"))
    for case in range(3):
        p = f"c{case}_"
        C = int(g[p + "C"])
        counts = torch.zeros((3, C), dtype=torch.int64, device="cuda")
        ops.iou_hist(dev(g[p + "pred"]), dev(g[p + "target"]), C, [int(g[p + "ignore"])], counts)
        cn = counts.cpu().numpy()
        assert np.array_equal(cn[0], g[p + "I"]) and np.array_equal(cn[2], g[p + "T"])
        assert np.array_equal(cn[1] + cn[2] - cn[0], g[p + "U"])


# ------------------------------------------------------------------------------------------ error behaviour
def test_error_paths(ops):
    from geopurify_amd._lib import GeoPurifyHipError
    x = torch.zeros((10, 512), device="cuda")
    nbr = torch.zeros((10, 96), dtype=torch.int32, device="cuda")
    w = torch.zeros((10, 96), device="cuda")
    with pytest.raises(GeoPurifyHipError):
        ops.pool_ell(x, nbr, w, 512, x)                              # aliasing is rejected
    with pytest.raises(GeoPurifyHipError):
        ops.sparse_conv(x, None, torch.zeros((100, 128), device="cuda"))   # cin not a multiple of 32
    # half-specified chunking (a chunk count without the host chunk tables) is rejected instead of falling back to the
    # upper-bound tile count (the branch a C-ABI caller could still reach)
    rng = np.random.default_rng(3)
    c = surface_voxels(rng, 600)
    ct, perm, rank, cs, grid = _sorted_voxels(ops, c)
    pairs = ops.conv_pairs_build(ops.kernel_map_build(grid, cs), 256)
    hi, lo = ops.conv_weights_split(torch.randn(27, 32, 256, device="cuda") * 0.05, 1.0)
    xin = torch.randn(len(c), 32, device="cuda")
    good = ops.sparse_conv_f16x3(xin, pairs, hi, lo)
    saved = pairs.chunk_tile_off
    pairs.chunk_tile_off = None
    with pytest.raises(GeoPurifyHipError, match="chunk tables"):
        ops.sparse_conv_f16x3(xin, pairs, hi, lo)
    pairs.chunk_tile_off = saved
    saved_rows = pairs.chunk_row_off
    bad = list(saved_rows)
    bad[-1] -= 1                                                       # chunks that do not cover the rows
    pairs.chunk_row_off = (ctypes.c_int32 * len(bad))(*bad)
    with pytest.raises(GeoPurifyHipError, match="do not cover"):
        ops.sparse_conv_f16x3(xin, pairs, hi, lo)
    bad = list(saved_rows)
    bad[1] = bad[2]                                                    # an empty chunk: not ascending
    pairs.chunk_row_off = (ctypes.c_int32 * len(bad))(*bad)
    with pytest.raises(GeoPurifyHipError, match="do not cover"):
        ops.sparse_conv_f16x3(xin, pairs, hi, lo)
    pairs.chunk_row_off = saved_rows
    assert torch.equal(ops.sparse_conv_f16x3(xin, pairs, hi, lo), good)
    # the matrix-core pooling builder pads row blocks only on request: 0 by default, 9 for the persistent kernel
    nb2 = torch.randint(0, 200, (200, 16), dtype=torch.int32, device="cuda")
    w2 = torch.rand(200, 16, device="cuda")
    assert ops.pool_mfma_build(nb2, w2, 64).min_steps < 9 <= ops.pool_mfma_build(nb2, w2, 64, min_steps=9).min_steps
    with pytest.raises(GeoPurifyHipError, match="min_steps"):
        ops.pool_mfma_build(nb2, w2, 64, min_steps=65)
    # tuning knobs: an undefined mask is rejected, a stamp buffer that is too small is rejected at the launch
    from geopurify_amd import _lib
    lib = _lib.load()
    assert lib.gp_debug_set(4, 1024) == -22 and lib.gp_debug_set(11, 8) == -22 and lib.gp_debug_set(8, 32) == -22 and lib.gp_debug_set(16, 0) == -22
    opm = ops.pool_mfma_build(nb2, w2, 64)
    xs2 = ops.split_f16(torch.randn(200, 512, device="cuda"), 512)
    ys2 = tuple(torch.empty((200, 512), dtype=torch.float16, device="cuda") for _ in range(2))
    small = torch.zeros(8, dtype=torch.int64, device="cuda")
    assert lib.gp_debug_ptr(0, small.data_ptr(), small.numel() * 8) == 0
    try:
        with pytest.raises(GeoPurifyHipError, match="stamp buffer"):
            ops.pool_mfma_apply(xs2, opm, 512, out_split=ys2)
    finally:
        assert lib.gp_debug_ptr(0, None, 0) == 0
    # the convolution's stamped twin: too small a buffer is rejected, a fitting one is written, the result keeps its bits
    hi5, lo5 = ops.conv_weights_split(torch.randn(27, 64, 256, device="cuda") * 0.05, 1.0)
    xs5 = ops.split_f16(torch.randn(len(c), 64, device="cuda"))
    good5 = ops.sparse_conv_f16x3(None, pairs, hi5, lo5, x_split=xs5)
    stamps = torch.zeros(4096 * 16, dtype=torch.int64, device="cuda")          # 16 x uint64 per workgroup (waves 0 and 4)
    try:
        assert lib.gp_debug_ptr(1, small.data_ptr(), small.numel() * 8) == 0
        with pytest.raises(GeoPurifyHipError, match="stamp buffer"):
            ops.sparse_conv_f16x3(None, pairs, hi5, lo5, x_split=xs5)
        assert lib.gp_debug_ptr(1, stamps.data_ptr(), stamps.numel() * 8) == 0
        y5 = ops.sparse_conv_f16x3(None, pairs, hi5, lo5, x_split=xs5)
        torch.cuda.synchronize()
    finally:
        assert lib.gp_debug_ptr(1, None, 0) == 0
    assert torch.equal(y5, good5)
    st = stamps.view(-1, 16).cpu()
    st = st[st[:, 6] > 0]
    assert len(st) > 0 and bool((st[:, 2] + st[:, 3] <= st[:, 6]).all())   # prologue + K loop inside the tile's total cycles
    assert bool((st[:, 8] + st[:, 9] + st[:, 10] <= st[:, 3]).all())       # DMA issue + s_waitcnt + s_barrier inside the K loop (wave 0)
    assert bool((((st[:, 11] >> 4) & 3) == ((st[:, 15] >> 4) & 3)).all())  # waves 0 and 4 of a workgroup share a SIMD (HW_ID bits 5:4)


def test_pooling_tuning_masks_keep_the_ring_discipline(ops):
    """The round-3 faults (scripts/bench_pool.py, pooling knob 4 = 8 / 9: memory access fault in pool_mfma_kernel): tuning bit 3
    dropped two of a stage's seven LDS-DMA instructions while the hand-over kept waiting vmcnt(7), so the older stage's row-id
    load could still be in flight when the ids were read back -- garbage row numbers went into the gather.  Now (a) the product
    kernels compile the bits out and (b) the *_tuning_kernel twins replace a switched-off fetch by one hot piece, keeping the
    instruction count.  Every mask that faulted (8, 9) and every other combination of the fetch bits must run to completion on
    every matrix-core kernel, and with the bits cleared the results must be the product kernels' bits."""
    from geopurify_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(17)
    c = surface_voxels(rng, 3000)
    ct, perm, rank, cs, grid = _sorted_voxels(ops, c)
    nbr = ops.knn_lattice(grid, cs, perm, 96)
    Nv = cs.shape[0]
    w = ops.affinity_softmax(torch.nn.functional.normalize(torch.randn(Nv, 128, device="cuda"), dim=1), nbr, 20.0)
    X = torch.randn(Nv, 512, device="cuda")
    xs = ops.split_f16(X, 512)
    op_cs = ops.pool_cs_build(nbr, w)
    op64, op128 = ops.pool_mfma_build(nbr, w, 64), ops.pool_mfma_build(nbr, w, 128)
    new = lambda: tuple(torch.empty((Nv, 512), dtype=torch.float16, device="cuda") for _ in range(2))
    runs = {"cs": lambda y: ops.pool_cs_apply(xs, op_cs, 512, out_split=y),
            "engine": lambda y: ops.pool_cs_apply(xs, op_cs, 512, out_split=y, engine=True),
            "mfma64": lambda y: ops.pool_mfma_apply(xs, op64, 512, out_split=y),
            "mfma128": lambda y: ops.pool_mfma_apply(xs, op128, 512, out_split=y)}
    ref = {}
    for name, f in runs.items():
        ref[name] = new()
        f(ref[name])
    torch.cuda.synchronize()
    try:
        for mask in (8, 9, 2, 10, 11, 1, 3, 16, 24, 27):
            assert lib.gp_debug_set(4, mask) == 0
            for name, f in runs.items():
                f(new())                                                  # results are meaningless; the launch must complete
            torch.cuda.synchronize()
    finally:
        assert lib.gp_debug_set(4, 0) == 0
    for name, f in runs.items():
        y = new()
        f(y)
        torch.cuda.synchronize()
        assert torch.equal(y[0], ref[name][0]) and torch.equal(y[1], ref[name][1]), name
    assert torch.equal(ref["cs"][0], ref["engine"][0]) and torch.equal(ref["cs"][1], ref["engine"][1])


# ------------------------------------------------------------------------------------------ row 9 fast path
def test_sparse_conv_f16x3_matches_fp32_accuracy(ops):
    """f16-split (hi+lo, 3 MFMAs) convolution: fp32-class accuracy against the fp64 oracle."""
    rng = np.random.default_rng(11)
    c = surface_voxels(rng, 2500)
    ct, perm, rank, cs, grid = _sorted_voxels(ops, c)
    nm = ops.kernel_map_build(grid, cs)
    pairs = ops.conv_pairs_build(nm, 2048)
    nmc = nm.cpu().numpy()
    assert pairs.num_pairs == int((nmc >= 0).sum())
    pos = pairs.pair_pos.cpu().numpy()
    assert np.array_equal(pos >= 0, nmc >= 0)
    assert np.array_equal(pairs.pair_in.cpu().numpy()[pos[pos >= 0]], nmc[nmc >= 0])
    off = pairs.pair_off.cpu().numpy()                                  # (chunk, k) segments, chunk-major
    CH = 2048
    seg_counts = np.concatenate([(nmc[:, c0:c0 + CH] >= 0).sum(1) for c0 in range(0, len(c), CH)])
    assert np.array_equal(np.diff(off), seg_counts) and off[-1] == pairs.num_pairs
    ts = pairs.tile_start.cpu().numpy()
    assert np.array_equal(np.diff(ts), (seg_counts + 255) // 256)
    Nv = len(c)
    X = torch.randn(Nv, 96) * 3.0
    X[:, :8] *= 1e-3                                                   # small-magnitude channels too
    W = torch.randn(27, 96, 256) * 0.05
    sc, sh = torch.rand(256) + 0.5, torch.randn(256)
    res = torch.randn(Nv, 256)
    p2 = 2.0 ** int(np.floor(np.log2(2.0 / float(W.abs().max()))))
    hi, lo = ops.conv_weights_split(dev(W), p2)
    y = ops.sparse_conv_f16x3(dev(X), pairs, hi, lo, dev(sc / p2), dev(sh), residual=dev(res), relu=True)
    ref = o_student.sparse_conv3(X.double(), nmc.astype(np.int64), W.double())
    ref = torch.relu(ref * sc.double() + sh.double() + res.double())
    err = (y.cpu().double() - ref).abs().max().item()
    y32 = ops.sparse_conv(dev(X), nm, dev(W), dev(sc), dev(sh), residual=dev(res), relu=True)
    err32 = (y32.cpu().double() - ref).abs().max().item()
    assert err < 5e-5 and err < 4 * err32 + 1e-6, (err, err32)        # same class as the exact-fp32 MFMA kernel
    # LDS-DMA path: pre-split operand in, split output out; identical arithmetic => identical result
    xs = ops.split_f16(dev(X))
    assert torch.equal(xs[0].float() + xs[1].float(), dev(X)) or (xs[0].float() + xs[1].float() - dev(X)).abs().max() < 1e-6
    ys = tuple(torch.empty((Nv, 256), dtype=torch.float16, device="cuda") for _ in range(2))
    y2 = ops.sparse_conv_f16x3(None, pairs, hi, lo, dev(sc / p2), dev(sh), residual=dev(res), relu=True, x_split=xs,
                               out_split=ys)
    # ... with fp32 partial rows (fp32_partials=True); the product kernel stores them as 24-bit block floating point: a
    # partial row is rounded within 2^-22 of the largest magnitude of its 128-column quarter, an output row sums at most 27 of them
    from geopurify_amd._lib import load
    lib = load()
    y2f = ops.sparse_conv_f16x3(None, pairs, hi, lo, dev(sc / p2), dev(sh), residual=dev(res), relu=True, x_split=xs, fp32_partials=True)
    assert torch.equal(y2f, y)
    part_max = float((ref - sh.double() - res.double()).abs().max() / sc.min())      # bound on any partial sum's magnitude
    assert (y2 - y).abs().max().item() <= 27 * 2.0 ** -22 * part_max * float(sc.max())
    err2 = (y2.cpu().double() - ref).abs().max().item()
    assert err2 < 5e-5 and err2 < 4 * err32 + 1e-6, (err2, err32)
    # the residual as the split planes of an earlier layer (hi + lo) * row scale instead of fp32 rows: the same sum up to the planes' 2^-22
    # of the row's largest magnitude; fp32 rows AND planes together are an error
    rh, rl, rinv = ops.split_f16(dev(res), per_row=True)
    y3 = ops.sparse_conv_f16x3(None, pairs, hi, lo, dev(sc / p2), dev(sh), residual=(rh, rl, rinv), relu=True, x_split=xs)
    res_back = (rh.float() + rl.float()) * rinv[:, None]
    assert (res_back - dev(res)).abs().max().item() <= 2.0 ** -22 * float(res.abs().max())
    assert (y3 - y2).abs().max().item() <= 2.0 ** -21 * float(res.abs().max()) + 1e-7
    # INTERLEAVED rows ([K step][hi 32 | lo 32]: operand, output and residual as ONE tensor each; the LDS-DMA kernel stages full lines):
    # the same arithmetic on the same values -- bit for bit the separate planes' result, and the output rows are the planes' halves
    ys3 = tuple(torch.empty((Nv, 256), dtype=torch.float16, device="cuda") for _ in range(2))
    y3b = ops.sparse_conv_f16x3(None, pairs, hi, lo, dev(sc / p2), dev(sh), residual=(rh, rl, rinv), relu=True, x_split=xs, out_split=ys3)
    assert torch.equal(y3b, y3)
    y_il = torch.full((Nv, 512), float("nan"), dtype=torch.float16, device="cuda")
    y4 = ops.sparse_conv_f16x3(None, pairs, hi, lo, dev(sc / p2), dev(sh), residual=(ops.interleave_planes(rh, rl), None, rinv), relu=True,
                               x_split=(ops.interleave_planes(*xs), None), out_split=(y_il, None))
    assert torch.equal(y4, y3)
    h4, l4 = ops.deinterleave_planes(y_il)
    assert torch.equal(h4, ys3[0]) and torch.equal(l4, ys3[1])
    # ... and through the fp32-partial twin (fp32_partials=True: the round 1-4 phase-2 kernel, four columns per lane) the same equality
    ys5 = tuple(torch.empty((Nv, 256), dtype=torch.float16, device="cuda") for _ in range(2))
    y5 = ops.sparse_conv_f16x3(None, pairs, hi, lo, dev(sc / p2), dev(sh), residual=(rh, rl, rinv), relu=True, x_split=xs, out_split=ys5, fp32_partials=True)
    y_il5 = torch.full((Nv, 512), float("nan"), dtype=torch.float16, device="cuda")
    y6 = ops.sparse_conv_f16x3(None, pairs, hi, lo, dev(sc / p2), dev(sh), residual=(ops.interleave_planes(rh, rl), None, rinv), relu=True,
                               x_split=(ops.interleave_planes(*xs), None), out_split=(y_il5, None), fp32_partials=True)
    h6, l6 = ops.deinterleave_planes(y_il5)
    assert torch.equal(y6, y5) and torch.equal(h6, ys5[0]) and torch.equal(l6, ys5[1])
    # the split kernel writes that form itself (lo = NULL): the same halves as its planes
    ph, pl, pinv = ops.split_f16(dev(X), per_row=True)
    rows, none_, rinv2 = ops.split_f16(dev(X), per_row=True, interleaved=True)
    assert none_ is None and torch.equal(rinv2, pinv) and torch.equal(rows, ops.interleave_planes(ph, pl))
    from geopurify_amd._lib import GeoPurifyHipError
    lib_c = load()
    with pytest.raises(GeoPurifyHipError):
        from geopurify_amd.ops import _ptr, check, _stream
        check(lib_c.gp_sparse_conv_f16x3(None, 0, _ptr(xs[0]), _ptr(xs[1]), xs[0].stride(0), _ptr(pairs.pair_in), _ptr(pairs.pair_pos), _ptr(pairs.pair_off),
                                         _ptr(pairs.tile_start), _ptr(pairs.tile_desc), pairs.nseg, pairs.num_pairs, Nv, 27, _ptr(hi), _ptr(lo), 96, 256,
                                         _ptr(pairs.partial), None, None, _ptr(dev(res)), 256, 1, _ptr(y3), 256, None, None, 0, int(pairs.num_chunks),
                                         pairs.chunk_row_off, pairs.chunk_tile_off, pairs.chunk_pair_off, None, None, _ptr(rh), _ptr(rl), rh.stride(0), _ptr(rinv),
                                         int(hi.dim() == 5), 0, _stream()), "gp_sparse_conv_f16x3")
    assert (ys[0].float() + ys[1].float() - y2).abs().max() <= 2e-6 * max(1.0, float(y2.abs().max()))
    # chunk heights chosen from the kernel map (gp_conv_chunk_plan): every launch within the tile target, the chunk tables consistent
    # with the map, and -- a row's sum does not depend on which rows share its tiles -- the same bits
    for target in (32, 64):
        old_target, ops.CONV_TARGET_TILES = ops.CONV_TARGET_TILES, target
        try:
            bal = ops.conv_pairs_build(nm, "balanced", col_tiles=1)
        finally:
            ops.CONV_TARGET_TILES = old_target
        rows_b, tiles_b, pairs_b = list(bal.chunk_row_off), list(bal.chunk_tile_off), list(bal.chunk_pair_off)
        assert rows_b[0] == 0 and rows_b[-1] == Nv and all(b > a and (b % 256 == 0 or b == Nv) for a, b in zip(rows_b[:-1], rows_b[1:]))
        assert bal.num_chunks > 1 and bal.num_pairs == pairs.num_pairs
        for ci in range(bal.num_chunks):
            cnt_k = (nmc[:, rows_b[ci]:rows_b[ci + 1]] >= 0).sum(1)
            assert pairs_b[ci + 1] - pairs_b[ci] == cnt_k.sum() and tiles_b[ci + 1] - tiles_b[ci] == ((cnt_k + 255) // 256).sum()
            assert tiles_b[ci + 1] - tiles_b[ci] <= target or rows_b[ci + 1] - rows_b[ci] <= 256
            if ci + 1 < bal.num_chunks:                                 # greedy: one more granule would have passed the target
                more = (nmc[:, rows_b[ci]:rows_b[ci + 1] + 256] >= 0).sum(1)
                assert ((more + 255) // 256).sum() > target
        yb = ops.sparse_conv_f16x3(None, bal, hi, lo, dev(sc / p2), dev(sh), residual=dev(res), relu=True, x_split=xs)
        assert torch.equal(yb, y2), target                              # (the 24-bit partial rows are per (row, quarter): no tile-mate enters)
    # edge shapes of the plan: fewer rows than one granule (one chunk), a target no granule fits (one granule per chunk), bad arguments
    from geopurify_amd import _lib as _l
    lib_ = _l.load()
    tiny = ops.conv_pairs_build(nm[:, :100].clamp(max=99).contiguous(), "balanced", col_tiles=1)
    assert tiny.num_chunks == 1 and list(tiny.chunk_row_off) == [0, 100]
    old_target, ops.CONV_TARGET_TILES = ops.CONV_TARGET_TILES, 1
    try:
        one = ops.conv_pairs_build(nm, "balanced", col_tiles=1)
    finally:
        ops.CONV_TARGET_TILES = old_target
    assert list(one.chunk_row_off) == list(range(0, Nv, 256)) + [Nv]
    assert torch.equal(ops.sparse_conv_f16x3(None, one, hi, lo, dev(sc / p2), dev(sh), residual=dev(res), relu=True, x_split=xs), y2)
    buf = torch.zeros(64, dtype=torch.int32, device="cuda")
    wsb = torch.zeros(lib_.gp_conv_chunk_plan_workspace_bytes(Nv, 256), dtype=torch.uint8, device="cuda")
    assert lib_.gp_conv_chunk_plan(nm.data_ptr(), Nv, 27, 32, 1, 64, 16, buf.data_ptr(), buf[40:].data_ptr(), wsb.data_ptr(), wsb.numel(), None) == -22   # granule < 64
    assert lib_.gp_conv_chunk_plan(nm.data_ptr(), Nv, 27, 256, 1, 64, 16, buf.data_ptr(), buf[40:].data_ptr(), wsb.data_ptr(), 8, None) == -22            # workspace
    assert lib_.gp_conv_chunk_plan(nm.data_ptr(), Nv, 27, 256, 1, 64, 2, buf.data_ptr(), buf[40:].data_ptr(), wsb.data_ptr(), wsb.numel(), None) == 0        # max_chunks caps the plan
    assert int(buf[40]) == 2 and buf[:3].tolist()[0] == 0 and buf[:3].tolist()[2] == Nv
    # launch grouping (ConvPairs.regroup): the same pairs in the same order, g chunks per launch => the same bits
    for g in (2, 5):
        grouped = pairs.regroup(g)
        assert grouped.num_chunks == -(-pairs.num_chunks // min(g, pairs.num_chunks)) and grouped.num_pairs == pairs.num_pairs
        assert grouped.max_chunk_pairs >= pairs.max_chunk_pairs
        yg = ops.sparse_conv_f16x3(None, grouped, hi, lo, dev(sc / p2), dev(sh), residual=dev(res), relu=True, x_split=xs)
        assert torch.equal(yg, y2), g


def test_conv_partial_rows_24bit_encoding_byte_for_byte(ops):
    """The partial rows between the two phases are BYTES with a stated format (include/geopurify_hip.h): per pair
    row and 128-column quarter an exponent byte E = the exponent field of (largest |v| + 1 ulp) and per element u = rint(v 2^(148 - E)) +
    2^22 in three little-endian bytes; a quarter's 384 bytes = its 16 lanes' first 16 bytes, then their last 8; the exponent bytes follow
    the rows.  Restated in numpy (oracle/conv_partial.py) from the fp32 partial rows the tuning twin leaves in the same buffer (fp32_partials=True: plane_flags bit 3) and compared
    byte for byte; the decoded values are within half a unit 2^(E - 149) <= 2^-22 of the quarter's maximum of the fp32 rows."""
    from geopurify_amd._lib import load
    lib = load()
    rng = np.random.default_rng(9)
    c = surface_voxels(rng, 900)
    ct, perm, rank, cs, grid = _sorted_voxels(ops, c)
    nm = ops.kernel_map_build(grid, cs)
    pairs = ops.conv_pairs_build(nm, None)                                           # one chunk: the buffer holds every pair row
    assert pairs.num_chunks == 1
    Nv, cin, cout = len(c), 64, 256
    g = torch.Generator().manual_seed(4)
    X = torch.relu(torch.randn(Nv, cin, generator=g)) * torch.exp(torch.randn(Nv, 1, generator=g) * 2.0)
    W = torch.randn(27, cin, cout, generator=g) * 0.05
    hi, lo = ops.conv_weights_split(dev(W), 16.0)
    xh, xl, inv = ops.split_f16(dev(X), per_row=True)
    sc = torch.full((cout,), 1.0 / 16.0)
    ops.sparse_conv_f16x3(None, pairs, hi, lo, dev(sc), None, x_split=(xh, xl), x_row_inv=inv, fp32_partials=True)
    P32 = pairs.partial[:pairs.num_pairs, :cout].clone().cpu().numpy()
    ops.sparse_conv_f16x3(None, pairs, hi, lo, dev(sc), None, x_split=(xh, xl), x_row_inv=inv)
    raw = pairs.partial.view(torch.uint8).flatten().cpu().numpy()
    P = pairs.num_pairs
    from oracle import conv_partial as o_cp
    want_rows, want_E = o_cp.encode(P32)
    e_off = o_cp.exponent_offset(P, cout)
    assert np.array_equal(raw[e_off:e_off + want_E.size], want_E)
    assert np.array_equal(raw[:want_rows.size], want_rows)
    dec = o_cp.decode(raw[:want_rows.size], raw[e_off:e_off + want_E.size], P, cout)
    half_unit = np.exp2(want_E.reshape(P, -1).astype(np.float64) - 149.0)
    m = np.abs(P32.reshape(P, -1, 128)).max(axis=2)
    assert (np.abs(dec - P32).reshape(P, -1, 128) <= half_unit[:, :, None]).all() and (half_unit <= m * 2.0 ** -22 * (1 + 1e-6)).all()


def test_sparse_conv_f16x3_wide_rows_row_scales_and_non_finite_rows(ops):
    """The 24-bit partial rows at their edges: 1024 output columns (the second register set of phase 2), the row-scaled split output,
    rows of very different magnitudes (the block exponent is per pair row and 128 columns: a 1e-6 row next to a 1e+3 row keeps ITS
    relative accuracy), an all-zero row, and non-finite activations: an Inf or NaN input row makes exactly the output rows that
    gather it non-finite (fp32 partial rows: the same rows) and leaves every other row's bits alone."""
    rng = np.random.default_rng(5)
    c = surface_voxels(rng, 1500)
    ct, perm, rank, cs, grid = _sorted_voxels(ops, c)
    nm = ops.kernel_map_build(grid, cs)
    pairs = ops.conv_pairs_build(nm, 512)
    nmc = nm.cpu().numpy()
    Nv = len(c)
    g = torch.Generator().manual_seed(3)
    X = torch.randn(Nv, 64, generator=g) * torch.exp(torch.randn(Nv, 1, generator=g) * 4.0)      # row magnitudes over ~7 decades
    X[11] = 0.0
    W = torch.randn(27, 64, 1024, generator=g) * 0.05
    p2 = 2.0 ** int(np.floor(np.log2(2.0 / float(W.abs().max()))))
    hi, lo = ops.conv_weights_split(dev(W), p2)
    sc = torch.full((1024,), 1.0 / p2)
    xh, xl, inv = ops.split_f16(dev(X), per_row=True)
    ys = tuple(torch.empty((Nv, 1024), dtype=torch.float16, device="cuda") for _ in range(2))
    yinv = torch.empty(Nv, device="cuda")
    y = ops.sparse_conv_f16x3(None, pairs, hi, lo, dev(sc), None, x_split=(xh, xl), x_row_inv=inv, out_split=ys, out_row_inv=yinv)
    ref = o_student.sparse_conv3(X.double(), nmc.astype(np.int64), W.double())
    # per output row: error against the row's own scale (the largest partial sum that enters it is bounded by the row's gathered inputs)
    gathered = torch.zeros(Nv, dtype=torch.float64)
    Xmax = X.abs().max(dim=1).values.double()
    for k in range(27):
        u = np.where(nmc[k] >= 0)[0]
        gathered[u] = torch.maximum(gathered[u], Xmax[nmc[k][u]])
    err = (y.cpu().double() - ref).abs().max(dim=1).values
    bound = 2e-5 * gathered * float(W.abs().max()) * 64 ** 0.5 + 1e-30
    assert bool((err <= bound).all()), float((err / bound).max())
    back = (ys[0].float() + ys[1].float()) * yinv[:, None]
    assert (back - y).abs().max(dim=1).values.le(2e-6 * y.abs().max(dim=1).values + 1e-30).all()
    # non-finite rows
    for bad in (float("inf"), float("nan")):
        Xb = X.clone()
        Xb[700, 5] = bad
        bh, bl, invb = ops.split_f16(dev(Xb), per_row=True)
        yb = ops.sparse_conv_f16x3(None, pairs, hi, lo, dev(sc), None, x_split=(bh, bl), x_row_inv=invb)
        hit = np.zeros(Nv, dtype=bool)
        for k in range(27):
            hit |= nmc[k] == 700
        fin = torch.isfinite(yb).all(dim=1).cpu().numpy()
        assert not fin[hit].any() and fin[~hit].all(), (bad, int(hit.sum()), int((~fin).sum()))
        assert torch.equal(yb[torch.from_numpy(~hit).cuda()], y[torch.from_numpy(~hit).cuda()])


def test_embed_head_f16x3_matches_fp64_and_the_fp32_kernel(ops):
    """The student's 1x1x1 output layer + F.normalize in one kernel on pre-split rows (affinity_module.py:66,71,1547): fp32-class
    accuracy against fp64, the error class of the exact-fp32 kernel + l2norm_rows_ it replaces; ragged row count, an all-zero row
    (F.normalize: 0 / max(0, 1e-12) = 0), wide dynamic range across rows (the per-row scales)."""
    from geopurify_amd._lib import GeoPurifyHipError
    g = torch.Generator().manual_seed(21)
    nv, cin, cout = 2500 + 37, 512, 128
    X = torch.relu(torch.randn(nv, cin, generator=g)) * torch.exp(torch.randn(nv, 1, generator=g) * 3.0)
    X[17] = 0.0
    W = torch.randn(cin, cout, generator=g) * 0.04
    p2 = 2.0 ** int(np.floor(np.log2(16384.0 / float(W.abs().max()))))
    hi, lo = ops.conv_weights_split(dev(W.reshape(1, cin, cout)), p2)
    xs = ops.split_f16(dev(X), cin, per_row=True)
    e = ops.embed_head_f16x3(xs[:2], hi, lo, 1.0 / p2, x_row_inv=xs[2])
    ref = X.double() @ W.double()
    refn = ref / ref.norm(dim=1, keepdim=True).clamp_min(1e-12)
    err = (e.cpu().double() - refn).abs().max().item()
    e32 = ops.l2norm_rows_(ops.sparse_conv(dev(X), None, dev(W)))
    err32 = (e32.cpu().double() - refn).abs().max().item()
    assert err < 2e-6 and err < 4 * err32 + 5e-7, (err, err32)
    assert torch.equal(e[17], torch.zeros(cout, device="cuda"))
    assert (e.norm(dim=1).cpu() - 1.0).abs()[torch.arange(nv) != 17].max() < 1e-6
    # without the normalisation: the plain product, relative to the row's magnitude
    y = ops.embed_head_f16x3(xs[:2], hi, lo, 1.0 / p2, x_row_inv=xs[2], normalize=False)
    rel = ((y.cpu().double() - ref).abs().max(dim=1).values / ref.abs().max(dim=1).values.clamp_min(1e-30))
    assert rel[torch.arange(nv) != 17].max() < 4e-6, rel.max()
    # bitwise repeatable, and independent of how many rows share a launch (rows 0..1279 alone = the same bits)
    assert torch.equal(ops.embed_head_f16x3(xs[:2], hi, lo, 1.0 / p2, x_row_inv=xs[2]), e)
    # round 5: the same rows x 2^10 as f16 hi / lo planes from the same epilogue (the matrix-core affinity kernel's operand): the fp32
    # rows beside them are unchanged, the planes are the split of exactly those rows, with or without the fp32 output
    e2, (ph, plo) = ops.embed_head_f16x3(xs[:2], hi, lo, 1.0 / p2, x_row_inv=xs[2], planes=True)
    assert torch.equal(e2, e)
    sv = e * 1024.0
    assert torch.equal(ph, sv.half()) and torch.equal(plo, (sv - sv.half().float()).half())
    none, (ph2, plo2) = ops.embed_head_f16x3(xs[:2], hi, lo, 1.0 / p2, x_row_inv=xs[2], planes=True, want_f32=False)
    assert none is None and torch.equal(ph2, ph) and torch.equal(plo2, plo)
    part = ops.embed_head_f16x3((xs[0][:1280], xs[1][:1280]), hi, lo, 1.0 / p2, x_row_inv=xs[2][:1280])
    assert torch.equal(part, e[:1280])
    with pytest.raises(GeoPurifyHipError, match="embedding channels"):
        ops.embed_head_f16x3(xs[:2], hi[:, :64].contiguous(), lo[:, :64].contiguous(), 1.0 / p2)
    with pytest.raises(ValueError):
        ops.embed_head_f16x3((xs[0][:, :256].contiguous(), xs[1][:, :256].contiguous()), hi, lo, 1.0 / p2)


# ------------------------------------------------------------------------------------------ row 12 fast path
@pytest.mark.parametrize("R", [4, 8, 16])
def test_pool_tiles_matches_ell_and_oracle(ops, R):
    rng = np.random.default_rng(12)
    c = surface_voxels(rng, 2500)
    ct, perm, rank, cs, grid = _sorted_voxels(ops, c)
    Nv, K, D = len(c), 96, 512
    nbr = ops.knn_lattice(grid, cs, perm, K)                            # internal (Morton) rows
    E = F.normalize(torch.randn(Nv, 128), dim=1)
    w = ops.affinity_softmax(dev(E), nbr, 20.0)
    tiles = ops.pool_tiles_build(nbr, w, R)
    # structure: every (row, neighbour, weight) appears exactly once in its tile's dense block
    off, urow, uw = tiles.tile_off.cpu().numpy(), tiles.u_row.cpu().numpy(), tiles.u_w.cpu().numpy()
    nbc, wc = nbr.cpu().numpy(), w.cpu().numpy()
    for t in (0, len(off) // 2, len(off) - 2):
        rows = range(t * R, min((t + 1) * R, Nv))
        u = urow[off[t]:off[t + 1]]
        assert len(np.unique(u)) == len(u) and set(u) == set(nbc[list(rows)].reshape(-1))
        dense = np.zeros((Nv if False else len(u), R), np.float32)
        pos = {v: i for i, v in enumerate(u)}
        for r_i, row in enumerate(rows):
            for j in range(K):
                dense[pos[nbc[row, j]], r_i] = wc[row, j]
        assert np.array_equal(uw[off[t]:off[t + 1]], dense)
    X = torch.randn(Nv, D + 32)
    Xd = dev(X)
    a = [torch.empty((Nv, D), device="cuda") for _ in range(2)]
    b = [torch.empty((Nv, D), device="cuda") for _ in range(2)]
    ca, cb = Xd, Xd
    T = 5
    for t in range(T):
        ops.pool_tiles_apply(ca, tiles, D, a[t % 2]); ca = a[t % 2]
        ops.pool_ell(cb, nbr, w, D, b[t % 2]); cb = b[t % 2]
    assert (ca - cb).abs().max() < 1e-5                                # same operator, different fp32 order
    ref = o_aff.pool_gather(X[:, :D], nbr.cpu().long(), w.cpu(), T)
    assert (ca.cpu().double() - ref).abs().max() < 1e-5
    if R in (4, 8):
        # the same operator at D = 64 (config P's width): one tile per wave, its union dealt over the four 16-lane quarters (pool_tiles64_kernel); rows of a
        # wider matrix (ld 96), unions that are no multiple of 16
        X6 = dev(torch.randn(Nv, 96))
        a6 = [torch.empty((Nv, 64), device="cuda") for _ in range(2)]
        b6 = [torch.empty((Nv, 64), device="cuda") for _ in range(2)]
        ca, cb = X6, X6
        for t in range(T):
            ops.pool_tiles_apply(ca, tiles, 64, a6[t % 2]); ca = a6[t % 2]
            ops.pool_ell(cb, nbr, w, 64, b6[t % 2]); cb = b6[t % 2]
        assert (ca - cb).abs().max() < 1e-5
        ref6 = o_aff.pool_gather(X6[:, :64].cpu(), nbr.cpu().long(), w.cpu(), T)
        assert (ca.cpu().double() - ref6).abs().max() < 1e-5
    else:
        with pytest.raises(Exception, match="d=64"):
            ops.pool_tiles_apply(dev(torch.randn(Nv, 64)), tiles, 64, torch.empty((Nv, 64), device="cuda"))


def test_gather_rows_classify_equals_the_two_calls(ops):
    """gp_gather_rows_classify (the final voxel -> point gather with the class decision computed in the same pass): the rows of
    gp_gather_rows and the labels / zero flags of gp_classify_argmax on them, bit for bit -- with and without a row map, a row of
    zeros, a point count that leaves the last group partly empty; shapes the fused kernel does not take are an error."""
    g = torch.Generator().manual_seed(8)
    Nv, N, D, C = 3000, 7013, 512, 19
    X = torch.randn(Nv, D + 32, generator=g)
    X[17, :D] = 0.0
    idx = torch.randint(0, Nv, (N,), generator=g)
    idx[5] = 17
    rmap = torch.randperm(Nv, generator=g).to(torch.int32)
    text = F.normalize(torch.randn(C, D, generator=g), dim=1)
    Xd, idd, rd, td = dev(X), dev(idx), dev(rmap), dev(text)
    for row_map in (None, rd):
        ref = ops.gather_rows(Xd, D, idd, row_map=row_map)
        pred_ref, zero_ref = ops.classify_argmax(ref, td, 14.2)
        out, pred, zero = ops.gather_rows_classify(Xd, D, idd, td, 14.2, row_map=row_map)
        assert torch.equal(out, ref) and torch.equal(pred, pred_ref) and torch.equal(zero, zero_ref)
        assert int(zero.sum()) >= 1 or row_map is not None
    assert ops.can_gather_rows_classify(512, 19) and not ops.can_gather_rows_classify(512, 160) and not ops.can_gather_rows_classify(96, 19)
    with pytest.raises(Exception, match="gp_gather_rows_classify"):
        ops.gather_rows_classify(Xd, 96, idd, dev(F.normalize(torch.randn(C, 96), dim=1)), 1.0)


# ------------------------------------------------------------------------------------------ row 12: the operator's own row order
@pytest.mark.parametrize("chunk", [1024, 2048])
def test_rcb_order_properties_and_the_maps_it_feeds(ops, chunk):
    """gp_rcb_order (recursive coordinate bisection inside Morton chunks into 128-row leaves): sigma / rho are inverse permutations
    that keep every chunk in place; a leaf's bounding box is tighter than the Morton block's it replaces and the neighbour union of
    the operator's 128-row blocks shrinks; gp_rows_renumber_i32, the row maps of gp_split_f16_scaled and of gp_embed_head_f16x3's
    planes are the index arithmetic they claim to be; the re-ordered operator pools to the same features."""
    rng = np.random.default_rng(21)
    c = surface_voxels(rng, 9000, ext=110)
    ct, perm, rank, cs, grid = _sorted_voxels(ops, c)
    Nv, K, D = len(c), 96, 512
    sigma, rho = ops.rcb_order(cs, chunk, 128)
    sg, rh = sigma.cpu().numpy().astype(np.int64), rho.cpu().numpy().astype(np.int64)
    assert np.array_equal(np.sort(sg), np.arange(Nv)) and np.array_equal(rh[sg], np.arange(Nv)) and np.array_equal(sg[rh], np.arange(Nv))
    assert np.array_equal(sg // chunk, np.arange(Nv) // chunk)                       # a row never leaves its chunk
    xyz = cs.cpu().numpy().astype(np.int64)
    def box_volume(order):
        vols = []
        for b in range(0, Nv - 127, 128):
            p = xyz[order[b:b + 128]]
            vols.append(np.prod(p.max(0) - p.min(0) + 1))
        return float(np.mean(vols))
    assert box_volume(sg) < 0.8 * box_volume(np.arange(Nv))                            # compact leaves
    nbr = ops.knn_lattice(grid, cs, perm, K)
    nbr2 = ops.rows_renumber(nbr, sigma, rho)
    assert torch.equal(nbr2.cpu(), torch.from_numpy(rh)[nbr.cpu().long()[torch.from_numpy(sg)]].to(torch.int32))
    union = lambda nb: float(np.mean([len(np.unique(nb[b:b + 128])) for b in range(0, Nv - 127, 128)])) / 128.0
    u_m, u_r = union(nbr.cpu().numpy()), union(nbr2.cpu().numpy())
    assert u_r < 0.98 * u_m, (u_m, u_r)                                                # smaller block unions: what the order is for (S scene: 4.75 -> 4.2)
    # the row maps: split planes and embedding planes written through rho
    X = torch.randn(Nv, D, device="cuda")
    sc = ops.pow2_scale(X, D)
    h0, l0 = ops.split_f16(X, D, scale=sc[0:1])
    h1, l1 = ops.split_f16(X, D, scale=sc[0:1], dst_row=rho)
    assert torch.equal(h1[rho.long()], h0) and torch.equal(l1[rho.long()], l0)
    hh, ll, inv0 = ops.split_f16(X, D, per_row=True)
    hp_, lp_, inv1 = ops.split_f16(X, D, per_row=True, dst_row=rho)
    assert torch.equal(hp_[rho.long()], hh) and torch.equal(lp_[rho.long()], ll) and torch.equal(inv1[rho.long()], inv0)
    with pytest.raises(ValueError):
        ops.split_f16(X, D, dst_row=rho)
    Wo = torch.randn(1, D, 128, device="cuda") * 0.05
    whi, wlo = ops.conv_weights_split(Wo, 64.0, blocked=False)
    y0, (e0h, e0l) = ops.embed_head_f16x3((hh, ll), whi, wlo, 1.0 / 64.0, x_row_inv=inv0, planes=True)
    y1, (e1h, e1l) = ops.embed_head_f16x3((hh, ll), whi, wlo, 1.0 / 64.0, x_row_inv=inv0, planes=True, plane_rows=rho)
    assert torch.equal(y1, y0) and torch.equal(e1h[rho.long()], e0h) and torch.equal(e1l[rho.long()], e0l)
    # the operator in its own order: same pooled features (another fp32 summation order), rows back through sigma
    E = F.normalize(torch.randn(Nv, 128), dim=1)
    w = ops.affinity_softmax(dev(E), nbr, 20.0)
    op_m, op_r = ops.pool_cs_build(nbr, w), ops.pool_cs_build(nbr2, w[sigma.long()].contiguous())
    assert op_r.total < op_m.total                                                     # fewer padded union rows to gather
    ym, yr = torch.empty(Nv, D, device="cuda"), torch.empty(Nv, D, device="cuda")
    ops.pool_cs_apply((h0, l0), op_m, D, out_f32=ym, out_scale=sc[1:2])
    ops.pool_cs_apply((h1, l1), op_r, D, out_f32=yr, out_scale=sc[1:2])
    assert (yr[rho.long()] - ym).abs().max() < 2e-6 * float(X.abs().max())
    with pytest.raises(Exception, match="chunk_rows"):
        ops.rcb_order(cs, 512, 128)


@pytest.mark.parametrize("n_vox,K,BR", [(130, 96, 64), (70, 32, 64), (130, 32, 128), (65, 8, 64)])
def test_pool_mfma_tiny_voxel_sets(ops, n_vox, K, BR):
    """edge shapes: a single (partial) row block, unions of exactly K+1..Nv rows, K far from 96, 1-step applications."""
    rng = np.random.default_rng(15)
    c = surface_voxels(rng, n_vox)
    ct, perm, rank, cs, grid = _sorted_voxels(ops, c)
    Nv, D = len(c), 512
    nbr = ops.knn_lattice(grid, cs, perm, K)
    w = ops.affinity_softmax(dev(F.normalize(torch.randn(Nv, 128), dim=1)), nbr, 20.0)
    op = ops.pool_mfma_build(nbr, w, BR)
    X = torch.randn(Nv, D)
    out = torch.empty((Nv, D), device="cuda")
    ops.pool_mfma_apply(ops.split_f16(dev(X)), op, D, out_f32=out)
    ref = o_aff.pool_gather(X, nbr.cpu().long(), w.cpu(), 1)
    assert (out.cpu().double() - ref).abs().max() < 1e-5



def test_pool_mfma_operator_with_non_local_neighbours(ops):
    """Neighbour lists that are NOT local (random ids): a 64-row block's union has thousands of distinct ids, which takes the
    operator builder's full-size hash table instead of the small one; the operator still equals the ELL gather."""
    rng = np.random.default_rng(77)
    Nv, K, D = 5000, 96, 512
    nbr = dev(torch.from_numpy(np.stack([rng.choice(Nv, K, replace=False) for _ in range(Nv)])).to(torch.int32))
    w = torch.softmax(torch.randn(Nv, K), dim=1)
    op = ops.pool_mfma_build(nbr, dev(w), 64)
    bn, bo, br = op.bu_n.cpu().numpy(), op.bu_off.cpu().numpy(), op.bu_row.cpu().numpy()
    assert bn[:-1].min() > 1024 and bn[-1] < 1024                     # full blocks: beyond the small table; the ragged last one: inside
    for b in (0, len(bn) - 1):
        u = br[bo[b]:bo[b] + bn[b]]
        rows = np.arange(b * 64, min(b * 64 + 64, Nv))
        assert (np.diff(u) > 0).all() and set(u) == set(nbr.cpu().numpy()[rows].reshape(-1))
    X = torch.randn(Nv, D)
    out = torch.empty((Nv, D), device="cuda")
    ops.pool_mfma_apply(ops.split_f16(dev(X), D), op, D, out_f32=out)
    ref = torch.empty((Nv, D), device="cuda")
    ops.pool_ell(dev(X), nbr, dev(w), D, ref)
    assert (out - ref).abs().max() < 1e-5

@pytest.mark.parametrize("n_vox,BR", [(2500, 64), (2531, 64), (2500, 128), (2531, 128)])
def test_pool_mfma_matches_ell_and_oracle(ops, n_vox, BR):
    """Matrix-core pooling (split f16 operands, fp32 accumulation) against the ELL gather and the oracle
    (models/affinity_module.py:1575-1587: torch.sparse.mm repeated num_iters times)."""
    rng = np.random.default_rng(14)
    c = surface_voxels(rng, n_vox)
    ct, perm, rank, cs, grid = _sorted_voxels(ops, c)
    Nv, K, D = len(c), 96, 512
    nbr = ops.knn_lattice(grid, cs, perm, K)
    E = F.normalize(torch.randn(Nv, 128), dim=1)
    w = ops.affinity_softmax(dev(E), nbr, 20.0)
    op = ops.pool_mfma_build(nbr, w, BR)
    NW = BR // 16
    # structure: padded sorted unions, every (row, neighbour, weight) exactly once in A-fragment order
    bo, bn, br = op.bu_off.cpu().numpy(), op.bu_n.cpu().numpy(), op.bu_row.cpu().numpy()
    wa = (op.wa_hi.float() + op.wa_lo.float()).cpu().numpy().reshape(-1, NW, 64, 8) / 1024.0
    nbc, wc = nbr.cpu().numpy(), w.cpu().numpy()
    assert (np.diff(bo) % 32 == 0).all() and bo[0] == 0
    for b in (0, len(bn) // 2, len(bn) - 1):
        rows = np.arange(b * BR, min(b * BR + BR, Nv))
        u = br[bo[b]:bo[b] + bn[b]]
        assert (np.diff(u) > 0).all() and set(u) == set(nbc[rows].reshape(-1))
        assert (br[bo[b] + bn[b]:bo[b + 1]] == u[0]).all()
        dense = np.zeros((BR, bo[b + 1] - bo[b]), np.float64)
        pos = {v: i for i, v in enumerate(u)}
        for r_i, row in enumerate(rows):
            for j in range(K):
                dense[r_i, pos[nbc[row, j]]] = wc[row, j]
        blk = wa[bo[b] // 32:bo[b + 1] // 32]                           # [steps, wave, lane, j]
        got = np.zeros_like(dense)
        for s_ in range(blk.shape[0]):
            for wv in range(NW):
                for lane in range(64):
                    got[wv * 16 + lane % 16, s_ * 32 + (lane // 16) * 8:s_ * 32 + (lane // 16) * 8 + 8] = blk[s_, wv, lane]
        assert np.abs(got - dense).max() < 1e-7
    X = torch.randn(Nv, 544)
    Xd = dev(X)
    T = 5
    sp = [ops.split_f16(Xd, D), tuple(torch.empty((Nv, D), dtype=torch.float16, device="cuda") for _ in range(2))]
    out = torch.empty((Nv, D), device="cuda")
    b_ = [torch.empty((Nv, D), device="cuda") for _ in range(2)]
    cb = Xd
    for t in range(T):
        last = t == T - 1
        ops.pool_mfma_apply(sp[t % 2], op, D, out_split=None if last else sp[(t + 1) % 2], out_f32=out if last else None)
        ops.pool_ell(cb, nbr, w, D, b_[t % 2]); cb = b_[t % 2]
    assert (out - cb).abs().max() < 2e-5                               # tolerance: fp32-class split arithmetic
    ref = o_aff.pool_gather(X[:, :D], nbr.cpu().long(), w.cpu(), T)
    assert (out.cpu().double() - ref).abs().max() < 2e-5
    # both outputs at once agree with each other
    y2 = tuple(torch.empty((Nv, D), dtype=torch.float16, device="cuda") for _ in range(2))
    o2 = torch.empty((Nv, D), device="cuda")
    ops.pool_mfma_apply(sp[0], op, D, out_split=y2, out_f32=o2)
    assert ((y2[0].float() + y2[1].float()) - o2).abs().max() < 1e-6


def _cs_dense_block(op, b, K, nbc, wc, Nv):
    """Dense [128, padded union] weight block of row block b rebuilt from the fragment arrays (non-empty fragments only) and
    from the ELL lists; also checks the union rows and the (step, group) masks."""
    bo, bn, br, bm = (t.cpu().numpy() for t in (op.bu_off, op.bu_n, op.bu_row, op.bu_mask))
    wa = (op.wa_hi.float() + op.wa_lo.float()).cpu().numpy().reshape(-1, 8, 64, 8) / 1024.0
    rpb = op.block_rows
    rows = np.arange(b * rpb, min(b * rpb + rpb, Nv))
    u = br[bo[b]:bo[b] + bn[b]]
    assert len(set(u)) == len(u) and set(u) == set(nbc[rows].reshape(-1))
    assert (br[bo[b] + bn[b]:bo[b + 1]] == u[0]).all()
    dense = np.zeros((128, bo[b + 1] - bo[b]), np.float64)
    pos = {v: i for i, v in enumerate(u)}
    for r_i, row in enumerate(rows):
        for j in range(K):
            dense[r_i, pos[nbc[row, j]]] = wc[row, j]
    steps = (bo[b + 1] - bo[b]) // 32
    got = np.zeros_like(dense)
    for s_ in range(steps):
        m = int(bm[bo[b] // 32 + s_]) & 0xFFFFFFFF
        for gq in range(8):
            nz = np.abs(dense[gq * 16:gq * 16 + 16, s_ * 32:s_ * 32 + 32]).max() > 0
            assert bool((m >> gq) & 1) == bool(nz), (b, s_, gq, m)
            if not nz:
                continue                                               # an empty fragment is undefined memory: never read
            for lane in range(64):
                got[gq * 16 + lane % 16, s_ * 32 + (lane // 16) * 8:s_ * 32 + (lane // 16) * 8 + 8] = wa[bo[b] // 32 + s_, gq, lane]
    assert np.abs(got - dense).max() < 1e-7
    # order of the union rows: (first group, last group, group set, id) ascending
    use = (dense[:, :len(u)] != 0).reshape(8, 16, -1).any(1)           # [group, union row]
    keys = []
    for i in range(len(u)):
        gs = np.flatnonzero(use[:, i])
        keys.append((gs[0], gs[-1], int((use[:, i] * (1 << np.arange(8))).sum()), u[i]))
    assert keys == sorted(keys)


def test_affinity_cs_fragments_on_non_local_neighbour_lists(ops):
    """The same kernels on neighbour lists that are NOT local (random distinct ids: a block's union is ~10 000 rows = hundreds of steps):
    the builder's validity words go through global atomics (they do not fit its LDS buffer), the affinity kernel's fragment pass
    re-loads the words of the steps beyond the 24 it keeps in LDS, and the operator builder runs its second, full-size count pass."""
    rng = np.random.default_rng(77)
    Nv, K = 2200, 96
    nbr = torch.from_numpy(np.stack([rng.choice(Nv, K, replace=False) for _ in range(Nv)]).astype(np.int32)).cuda()
    E = F.normalize(torch.randn(Nv, 128), dim=1).cuda().contiguous()
    op = ops.pool_cs_plan(nbr, structure="valid")
    assert op.max_union > 1024 and int((op.bu_off[1:] - op.bu_off[:-1]).max()) // 32 > 24
    ops.affinity_cs_fragments(E, 20.0, op)
    ref_op = ops.pool_cs_plan(nbr, structure=True)
    w_blk = ops.affinity_softmax(E, nbr, 20.0, into=ref_op)
    assert torch.equal(ref_op.bu_row, op.bu_row) and torch.equal(ref_op.bu_mask, op.bu_mask)
    frag = (op.wa_hi.float() + op.wa_lo.float()) / 1024.0
    w = frag[ref_op.dst.long()]
    Ed = E.double()
    wref = torch.softmax(20.0 * (Ed[:, None, :] * Ed[nbr.long()]).sum(-1), dim=1)
    assert (w.double() - wref).abs().max().item() < 2e-6 and (w - w_blk).abs().max().item() < 4e-6
    X = torch.randn(Nv, 544, device="cuda")
    y_ell, y_cs = torch.empty(Nv, 512, device="cuda"), torch.empty(Nv, 512, device="cuda")
    ops.pool_ell(X, nbr, w.contiguous(), 512, y_ell)
    ops.pool_cs_apply(ops.split_f16(X, 512), op, 512, out_f32=y_cs)
    assert (y_cs - y_ell).abs().max().item() < 1e-5


@pytest.mark.parametrize("n_vox,rpb,K", [(2531, 128, 96), (2500, 100, 96), (1900, 128, 20), (300, 128, 96), (4000, 117, 64)])
def test_affinity_cs_fragments_vs_fp64_and_the_block_kernel(ops, n_vox, rpb, K):
    """gp_pool_cs_structure_valid + gp_affinity_cs_fragments (rows 11 + operator fill on the matrix cores) on small operators: ragged
    block heights, K below 96, a single block; weights read back from the fragments against an fp64 softmax (2e-6) and against
    affinity_block_kernel's (4e-6: both carry fp32 rounding); the validity words against the neighbour lists; an application of the
    operator equal to the ELL kernel's with those weights (1e-5)."""
    rng = np.random.default_rng(21)
    c = surface_voxels(rng, n_vox)
    ct, perm, rank, cs, grid = _sorted_voxels(ops, c)
    Nv = len(c)
    K = min(K, Nv)
    nbr = ops.knn_lattice(grid, cs, perm, K)
    E = F.normalize(torch.randn(8, 128)[torch.from_numpy((c[:, 0] // 6) % 8)] + 0.6 * torch.randn(Nv, 128), dim=1).cuda().contiguous()
    op = ops.pool_cs_plan(nbr, rows_per_block=rpb, structure="valid")
    ops.affinity_cs_fragments(E, 20.0, op)
    ref_op = ops.pool_cs_plan(nbr, rows_per_block=rpb, structure=True)
    w_blk = ops.affinity_softmax(E, nbr, 20.0, into=ref_op)
    assert torch.equal(ref_op.bu_row, op.bu_row) and torch.equal(ref_op.bu_mask, op.bu_mask)
    # validity words == the dst table's (step, row, bit) set
    steps = op.total // 32
    el = ref_op.dst.long()                                                  # element = ((step * 8 + group) * 64 + lane) * 8 + e
    st, grp, ln, e8 = el // 4096, (el // 512) % 8, (el // 8) % 64, el % 8
    krow = (ln // 16) * 8 + e8
    rl = grp * 16 + ln % 16
    want = torch.zeros(steps * 128, dtype=torch.int64, device="cuda")
    want.index_put_(((st * 128 + rl).flatten(),), torch.bitwise_left_shift(torch.ones_like(krow), krow).flatten(), accumulate=True)
    got = op.valid[:steps * 128].long() & 0xFFFFFFFF
    assert torch.equal(got, want)
    frag = (op.wa_hi.float() + op.wa_lo.float()) / 1024.0
    w = frag[el]
    Ed = E.double()
    wref = torch.softmax(20.0 * (Ed[:, None, :] * Ed[nbr.long()]).sum(-1), dim=1)
    assert (w.double() - wref).abs().max().item() < 2e-6
    assert (w - w_blk).abs().max().item() < 4e-6
    X = torch.randn(Nv, 544, device="cuda")
    y_ell = torch.empty(Nv, 512, device="cuda")
    ops.pool_ell(X, nbr, w.contiguous(), 512, y_ell)
    y_cs = torch.empty(Nv, 512, device="cuda")
    ops.pool_cs_apply(ops.split_f16(X, 512), op, 512, out_f32=y_cs)
    assert (y_cs - y_ell).abs().max().item() < 1e-5


@pytest.mark.parametrize("kind,rpb,T", [("lattice", 128, 4), ("lattice", 100, 3), ("random", 128, 4), ("random", 117, 2)])
def test_pool_cs_chained_launch_small_and_overflowing_lists(ops, kind, rpb, T):
    """gp_pool_cs_apply_chain on small operators: bit-identical to T launches of gp_pool_cs_apply for ragged block heights, and for
    neighbour lists that are NOT local (random ids: every row block depends on more than 63 others, so every tile takes the
    "wait for every block" path of its dependency list); dependency lists against numpy; epochs across repeated calls."""
    rng = np.random.default_rng(5)
    K, D = 96, 512
    if kind == "lattice":
        c = surface_voxels(rng, 9000)
        ct, perm, rank, cs, grid = _sorted_voxels(ops, c)
        Nv = len(c)
        nbr = ops.knn_lattice(grid, cs, perm, K)
    else:
        Nv = 9100
        nbr = torch.from_numpy(np.stack([rng.choice(Nv, K, replace=False) for _ in range(Nv)]).astype(np.int32)).cuda()
    w = torch.softmax(torch.randn(Nv, K, device="cuda"), dim=1).contiguous()
    op = ops.pool_cs_build(nbr, w, rows_per_block=rpb)
    ops.pool_cs_deps(op)
    off, row = op.bu_off.cpu().numpy(), op.bu_row.cpu().numpy()
    nb = off.size - 1
    src, dst = np.repeat(np.arange(nb), np.diff(off)), row // rpb
    e = np.unique(np.concatenate([src * nb + dst, dst * nb + src, np.arange(nb) * (nb + 1)]))
    want = np.split(e % nb, np.cumsum(np.bincount(e // nb, minlength=nb))[:-1])
    got = op.dep.view(-1, 64).cpu().numpy()
    for g, wl in zip(got, want):
        assert (g[0] > 63 and len(wl) > 63) or (g[0] == len(wl) and set(g[1:g[0] + 1].tolist()) == set(wl.tolist()))
    assert (got[:, 0] > 63).all() == (kind == "random")
    X = torch.randn(Nv, 544, device="cuda")
    x0 = ops.split_f16(X, D)
    for rep in range(3):
        sp = [tuple(t.clone() for t in x0), tuple(torch.full((Nv, D), float("nan"), dtype=torch.float16, device="cuda") for _ in range(2))]
        ref = torch.full((Nv, D), float("nan"), device="cuda")
        src_ = sp[0]
        for t in range(T):
            last = t == T - 1
            dst_ = None if last else sp[(t + 1) % 2]
            ops.pool_cs_apply(src_, op, D, out_split=dst_, out_f32=ref if last else None)
            src_ = dst_
        xs = tuple(t.clone() for t in x0)
        pong = tuple(torch.full((Nv, D), float("nan"), dtype=torch.float16, device="cuda") for _ in range(2))
        out = torch.full((Nv, D), float("nan"), device="cuda")
        ops.pool_cs_apply_chain(xs, pong, op, D, T, out)
        torch.cuda.synchronize()
        for a, b in zip([out, *xs, *pong], [ref, *sp[0], *sp[1]]):
            assert torch.equal(a, b)
    ops.pool_cs_chain_check(op)
    assert op.epoch == 3 * T
    with pytest.raises(Exception, match="applications"):
        ops.pool_cs_apply_chain(xs, pong, op, D, 1, out)


def test_pool_cs_chained_launch_gives_up_instead_of_hanging(ops):
    """gp_pool_cs_apply_chain's contract for a dependency that never comes: the wait is bounded (2 s of the constant clock), the workgroup
    sets the abort word and leaves, every later poll sees it -- the grid drains -- and the host side of the contract
    (pool_cs_chain_check) raises.  Staged by pointing one row block's list at a flag nobody publishes (the header word in front of
    the flags: index -1)."""
    import time
    rng = np.random.default_rng(9)
    c = surface_voxels(rng, 3000)
    ct, perm, rank, cs, grid = _sorted_voxels(ops, c)
    Nv, K, D = len(c), 32, 512
    nbr = ops.knn_lattice(grid, cs, perm, K)
    w = torch.softmax(torch.randn(Nv, K, device="cuda"), dim=1).contiguous()
    op = ops.pool_cs_build(nbr, w)
    ops.pool_cs_deps(op)
    dep = op.dep.view(-1, 64)
    n0 = int(dep[3, 0])
    assert 1 <= n0 < 63
    dep[3, 1] = -1                                                   # row block 3 waits for a flag that stays 0
    xs = ops.split_f16(torch.randn(Nv, 544, device="cuda"), D)
    pong = tuple(torch.empty((Nv, D), dtype=torch.float16, device="cuda") for _ in range(2))
    out = torch.empty(Nv, D, device="cuda")
    t0 = time.time()
    ops.pool_cs_apply_chain(xs, pong, op, D, 3, out)
    torch.cuda.synchronize()
    took = time.time() - t0
    assert 1.5 < took < 30, took                                     # one bounded wait, not one per dependent tile
    with pytest.raises(Exception, match="waited 2 s"):
        ops.pool_cs_chain_check(op)
    # the operator is usable again once the caller has dealt with it: lists repaired, abort word cleared
    ops.pool_cs_deps(op)
    ref = torch.empty(Nv, D, device="cuda")
    sp = [tuple(t.clone() for t in xs), tuple(torch.empty_like(t) for t in pong)]
    x_keep = tuple(t.clone() for t in xs)
    src = sp[0]
    for t in range(3):
        last = t == 2
        dst = None if last else sp[(t + 1) % 2]
        ops.pool_cs_apply(src, op, D, out_split=dst, out_f32=ref if last else None)
        src = dst
    ops.pool_cs_apply_chain(tuple(t.clone() for t in x_keep), pong, op, D, 3, out)
    torch.cuda.synchronize()
    ops.pool_cs_chain_check(op)
    assert torch.equal(out, ref)


@pytest.mark.parametrize("n_vox,rpb", [(2500, 128), (2531, 128), (2531, 100), (2500, 117)])
def test_pool_cs_matches_ell_and_oracle(ops, n_vox, rpb):
    """Column-sliced matrix-core pooling (blocks of rpb <= 128 rows, union rows grouped by the 16-row groups that use them,
    empty weight fragments skipped) against the ELL gather and the oracle (models/affinity_module.py:1575-1587)."""
    rng = np.random.default_rng(14)
    c = surface_voxels(rng, n_vox)
    ct, perm, rank, cs, grid = _sorted_voxels(ops, c)
    Nv, K, D = len(c), 96, 512
    nbr = ops.knn_lattice(grid, cs, perm, K)
    E = F.normalize(torch.randn(Nv, 128), dim=1)
    w = ops.affinity_softmax(dev(E), nbr, 20.0)
    op = ops.pool_cs_build(nbr, w, rows_per_block=rpb)
    nbc, wc = nbr.cpu().numpy(), w.cpu().numpy()
    bo = op.bu_off.cpu().numpy()
    assert (np.diff(bo) % 32 == 0).all() and bo[0] == 0 and (np.diff(bo) >= 32).all() and len(bo) - 1 == -(-len(c) // rpb)
    nb = len(bo) - 1
    for b in (0, nb // 2, nb - 1):
        _cs_dense_block(op, b, K, nbc, wc, Nv)
    bm = op.bu_mask.cpu().numpy().astype(np.int64) & 0xFF
    fill = np.unpackbits(bm.astype(np.uint8)[:, None], axis=1).mean()
    assert 0.2 < fill < 0.9, fill                                      # the grouping leaves a good part of the fragments empty
    X = torch.randn(Nv, 544)
    Xd = dev(X)
    T = 5
    sp = [ops.split_f16(Xd, D), tuple(torch.empty((Nv, D), dtype=torch.float16, device="cuda") for _ in range(2))]
    out = torch.empty((Nv, D), device="cuda")
    b_ = [torch.empty((Nv, D), device="cuda") for _ in range(2)]
    cb = Xd
    for t in range(T):
        last = t == T - 1
        ops.pool_cs_apply(sp[t % 2], op, D, out_split=None if last else sp[(t + 1) % 2], out_f32=out if last else None)
        ops.pool_ell(cb, nbr, w, D, b_[t % 2]); cb = b_[t % 2]
    assert (out - cb).abs().max() < 2e-5                               # tolerance: fp32-class split arithmetic
    ref = o_aff.pool_gather(X[:, :D], nbr.cpu().long(), w.cpu(), T)
    assert (out.cpu().double() - ref).abs().max() < 2e-5
    y2 = tuple(torch.empty((Nv, D), dtype=torch.float16, device="cuda") for _ in range(2))
    o2 = torch.empty((Nv, D), device="cuda")
    ops.pool_cs_apply(sp[0], op, D, out_split=y2, out_f32=o2)
    assert ((y2[0].float() + y2[1].float()) - o2).abs().max() < 1e-6
    # bitwise reproducible: the same operator applied twice gives the same bits
    o3 = torch.empty((Nv, D), device="cuda")
    ops.pool_cs_apply(sp[0], op, D, out_f32=o3)
    assert torch.equal(o2, o3)
    # the operator built in the other order -- structure first (lists only), weights written in place by the affinity kernel --
    # is the same operator: same weights, same union rows and masks, same bits out of an application
    op2 = ops.pool_cs_plan(nbr, rows_per_block=rpb, structure=True)
    assert op2.dst is not None and not op2.filled
    w2 = ops.affinity_softmax(dev(E), nbr, 20.0, into=op2)
    assert op2.filled and torch.equal(w2, w)
    assert torch.equal(op2.bu_row, op.bu_row) and torch.equal(op2.bu_mask, op.bu_mask) and torch.equal(op2.bu_off, op.bu_off)
    dstc = op2.dst.cpu().long()
    assert dstc.min() >= 0 and dstc.max() < op2.wa_hi.numel() and dstc.unique().numel() == dstc.numel()   # one element per (row, j)
    assert torch.equal(op2.wa_hi[op2.dst.long()], op.wa_hi[op2.dst.long()]) and torch.equal(op2.wa_lo[op2.dst.long()], op.wa_lo[op2.dst.long()])
    o4 = torch.empty((Nv, D), device="cuda")
    ops.pool_cs_apply(sp[0], op2, D, out_f32=o4)
    assert torch.equal(o4, o3)
    with pytest.raises(ValueError):
        ops.affinity_softmax(dev(E), nbr[:-1].contiguous(), 20.0, into=op2)


@pytest.mark.parametrize("n_vox,K", [(130, 96), (70, 32), (257, 32), (65, 8), (129, 1)])
def test_pool_cs_tiny_voxel_sets(ops, n_vox, K):
    """edge shapes: a single (partial) row block, one-step blocks, K far from 96."""
    rng = np.random.default_rng(15)
    c = surface_voxels(rng, n_vox)
    ct, perm, rank, cs, grid = _sorted_voxels(ops, c)
    Nv, D = len(c), 512
    nbr = ops.knn_lattice(grid, cs, perm, K)
    w = ops.affinity_softmax(dev(F.normalize(torch.randn(Nv, 128), dim=1)), nbr, 20.0)
    op = ops.pool_cs_build(nbr, w)
    X = torch.randn(Nv, D)
    out = torch.empty((Nv, D), device="cuda")
    ops.pool_cs_apply(ops.split_f16(dev(X)), op, D, out_f32=out)
    ref = o_aff.pool_gather(X, nbr.cpu().long(), w.cpu(), 1)
    assert (out.cpu().double() - ref).abs().max() < 1e-5


def test_pool_cs_operator_with_non_local_neighbours(ops):
    """Neighbour lists that are NOT local (random ids): a 128-row block's union has thousands of distinct ids (the builder's
    full-size hash table, sorts of 8192 keys); the operator still equals the ELL gather."""
    rng = np.random.default_rng(77)
    Nv, K, D = 5000, 96, 512
    nbr = dev(torch.from_numpy(np.stack([rng.choice(Nv, K, replace=False) for _ in range(Nv)])).to(torch.int32))
    w = torch.softmax(torch.randn(Nv, K), dim=1)
    op = ops.pool_cs_build(nbr, dev(w))
    bn = op.bu_n.cpu().numpy()
    assert bn[:-1].min() > 2048
    _cs_dense_block(op, len(bn) - 1, K, nbr.cpu().numpy(), w.numpy(), Nv)
    X = torch.randn(Nv, D)
    out = torch.empty((Nv, D), device="cuda")
    ops.pool_cs_apply(ops.split_f16(dev(X), D), op, D, out_f32=out)
    ref = torch.empty((Nv, D), device="cuda")
    ops.pool_ell(dev(X), nbr, dev(w), D, ref)
    assert (out - ref).abs().max() < 1e-5


def test_nn1_masked_grid_path_equals_bruteforce(ops):
    """>= 32768 points take the grid-accelerated search; must equal the brute force (incl. far queries)."""
    from geopurify_amd import _lib
    rng = np.random.default_rng(14)
    n = 60000
    xyz = rng.uniform(0, 4, size=(n, 3)).astype(np.float32)
    xyz[:, 2] *= 0.02                                                  # a thin slab: surface-like
    xyz[:200] += np.array([30.0, 0, 0], np.float32)                    # a far-away cluster of queries
    xyz[200:260] = xyz[300:360]                                        # exact duplicates (ties -> lowest index)
    ref = rng.random(n) < 0.7
    ref[:200] = False
    qry = ~ref
    xd = dev(xyz)
    rm, qm = dev(ref.astype(np.uint8)), dev(qry.astype(np.uint8))
    lib = _lib.load()
    nn_grid = ops.nn1_masked(xd, rm, qm).cpu().numpy()
    lib.gp_debug_set(5, 1)                                             # force the brute-force kernel
    nn_bf = ops.nn1_masked(xd, rm, qm).cpu().numpy()
    lib.gp_debug_set(5, 0)
    assert np.array_equal(nn_grid, nn_bf)
    assert (nn_grid[ref] == -1).all() and (nn_grid[qry] >= 0).all() and ref[nn_grid[qry]].all()
    sub = np.where(qry)[0][:400]
    exp = np.where(ref)[0][o_lift.nn1_indices_bruteforce(xyz[ref], xyz[sub])]
    assert np.array_equal(nn_grid[sub], exp)


def test_c_abi_example_runs_without_python_or_torch(ops, tmp_path):
    """examples/c_abi_pooling.c: a plain C99 program (gcc, no torch, no C++) builds the matrix-core pooling operator
    and applies it through include/geopurify_hip.h, checks itself against a double-precision loop and the error path."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "c_abi_pooling")
    libdir = os.path.join(root, "geopurify_amd")
    cc = subprocess.run(["gcc", "-std=c99", "-O2", "-D__HIP_PLATFORM_AMD__", os.path.join(root, "examples", "c_abi_pooling.c"), "-I" + os.path.join(root, "include"),
                         "-I/opt/rocm/include", "-L" + libdir, "-lgeopurify_hip", "-L/opt/rocm/lib", "-lamdhip64",
                         "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", exe], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and run.stdout.strip().endswith("OK"), run.stdout + run.stderr


# ------------------------------------------------------------------------------------------ SURVEY 8(f)-3
@pytest.mark.gpu
@pytest.mark.parametrize("dtype,d", [(torch.float16, 768), (torch.float32, 512), (torch.float16, 3)])
def test_fused_decode_vs_index_arithmetic(ops, dtype, d):
    """gp_fused_decode against the reference's own index arithmetic (dataset/feature_loader.py:141-190: nonzero / cumsum /
    fancy indexing) on random masks: both modes, with and without the three-key form's row mask, 16-byte and 2-byte rows."""
    g = torch.Generator().manual_seed(5)
    n, nv = 5000, 1700
    mask_chunk = torch.rand(n, generator=g) < 0.6
    rows = int(mask_chunk.sum())
    feat = torch.randn(rows, d, generator=g).to(dtype)
    row_keep = torch.rand(rows, generator=g) < 0.7
    vox_ind = torch.randperm(n, generator=g)[:nv].sort().values
    rank = torch.cumsum(mask_chunk.long(), 0) - 1
    for rk in (None, row_keep):
        # training forms: feat' = feat[rk], mask_chunk' = chunk & seen, rows of the representatives inside mask_chunk'
        mc2 = mask_chunk.clone()
        if rk is not None:
            mc2[mask_chunk.clone()] = rk
        want_mask = mc2[vox_ind]
        want_rows = feat[rank[vox_ind[want_mask]]]
        got, gm = ops.fused_decode(mask_chunk.cuda(), feat.cuda(), vox_ind.cuda(), 0, None if rk is None else rk.cuda())
        assert got.dtype == dtype and torch.equal(gm.cpu(), want_mask) and torch.equal(got.cpu(), want_rows)
        # evaluation form (two keys): scatter to all points, then one row per representative
        full = torch.zeros((n, d), dtype=dtype)
        full[mask_chunk] = feat
        got, gm = ops.fused_decode(mask_chunk.cuda(), feat.cuda(), vox_ind.cuda(), 1, None if rk is None else rk.cuda())
        assert torch.equal(got.cpu(), full[vox_ind]) and torch.equal(gm.cpu(), want_mask)
    from geopurify_amd._lib import GeoPurifyHipError
    with pytest.raises(GeoPurifyHipError, match="mode"):
        ops.fused_decode(mask_chunk.cuda(), feat.cuda(), vox_ind.cuda(), 2)


def test_voxelizer_device_form_equals_host_form(ops):
    """FusedFeatureLoader(device=) used to re-implement the voxelizer call inline (always M_r @ M_v, no clip box, no
    normal rotation).  Voxelizer.voxelize_device shares the clip / matrix logic with the host call: the same np.random stream
    gives the same voxels, representatives, features (normals rotated) and labels -- without augmentation, with a clip box and
    with nine feature columns."""
    from geopurify_amd.voxelizer import Voxelizer
    rng = np.random.default_rng(5)
    N = 6000
    pts = rng.uniform(-2.0, 2.0, size=(N, 3))
    feats = rng.normal(size=(N, 9))
    labels = rng.integers(0, 20, size=N)
    for kw in (dict(use_augmentation=False),
               dict(use_augmentation=True, scale_augmentation_bound=(0.9, 1.1),
                    rotation_augmentation_bound=((-0.05, 0.05), (-0.05, 0.05), (-3.14, 3.14))),
               dict(use_augmentation=True, clip_bound=((-1.0, 1.0), (-1.5, 1.5), (-0.5, 2.5)),
                    scale_augmentation_bound=(0.9, 1.1), rotation_augmentation_bound=((-0.05, 0.05), (-0.05, 0.05), (-3.14, 3.14)),
                    translation_augmentation_ratio_bound=((-0.2, 0.2), (-0.2, 0.2), (0, 0)))):
        vz = Voxelizer(voxel_size=0.05, **kw)
        np.random.seed(11)
        c_h, f_h, l_h, inv_h, ind_h = vz.voxelize(pts, feats.copy(), labels, return_ind=True)
        np.random.seed(11)
        r = vz.voxelize_device(torch.from_numpy(pts).cuda(), torch.from_numpy(feats).cuda(), torch.from_numpy(labels).cuda())
        assert np.array_equal(r["coords_aug"].cpu().numpy(), c_h) and np.array_equal(r["inds"].cpu().numpy(), ind_h)
        assert np.array_equal(r["inds_reconstruct"].cpu().numpy(), inv_h) and np.array_equal(r["labels"].cpu().numpy(), l_h)
        assert np.abs(r["feats"].cpu().numpy() - f_h).max() < 1e-12
        if "clip_bound" in kw:
            assert len(inv_h) < N                                       # the clip box removed points on both paths
