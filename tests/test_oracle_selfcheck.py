"""Cross-check the oracle rows that sit on absent third-party engines ("parity unpinned") against
independent dense / brute-force formulations.  CPU only, small sizes."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import affinity, lift, student


def _surface_voxels(rng, n=900):
    """unique integer voxels on two perpendicular sheets"""
    a = np.c_[rng.integers(0, 24, n), rng.integers(0, 24, n), rng.integers(3, 5, n)]
    b = np.c_[rng.integers(0, 24, n // 2), np.full(n // 2, 7), rng.integers(0, 16, n // 2)]
    return np.unique(np.vstack([a, b]), axis=0)


def test_sparse_conv_vs_dense_conv3d():
    rng = np.random.default_rng(0)
    c = _surface_voxels(rng)
    c = c[rng.permutation(len(c))]
    X = torch.randn(len(c), 7, dtype=torch.float64)
    W = torch.randn(27, 7, 5, dtype=torch.float64)
    nm = student.build_kernel_map(c)
    y1 = student.sparse_conv3(X, nm, W)
    y2 = student.sparse_conv3_dense_check(X, c, W)
    assert torch.allclose(y1, y2, atol=1e-10)
    assert (nm[13] == np.arange(len(c))).all()                      # centre offset = identity
    # symmetry of the map: nbr_k(u) = v  <=>  nbr_{26-k}(v) = u
    for k in (0, 5, 22):
        u = np.where(nm[k] >= 0)[0]
        assert (nm[26 - k][nm[k][u]] == u).all()


def test_student_forward_shapes_and_norm():
    rng = np.random.default_rng(1)
    c = _surface_voxels(rng, 300)
    sd = student.random_student_state_dict(10, hidden=16, embed=8, num_blocks=2, seed=3)
    X = torch.randn(len(c), 10)
    E = student.student_forward(X, c, sd, num_blocks=2)
    assert E.shape == (len(c), 8)
    assert torch.allclose(E.norm(dim=1), torch.ones(len(c)), atol=1e-5)
    E64 = student.student_forward(X, c, sd, num_blocks=2, dtype=torch.float64)
    assert (E - E64.float()).abs().max() < 1e-5


def test_knn_canonical_order_with_ties():
    rng = np.random.default_rng(2)
    c = _surface_voxels(rng, 500)
    c = c[rng.permutation(len(c))]
    K = 12
    nbr = affinity.knn_lattice(c, K, chunk=97).numpy()
    ci = c.astype(np.int64)
    for i in rng.integers(0, len(c), 40):
        d2 = ((ci - ci[i]) ** 2).sum(1)
        order = np.lexsort((np.arange(len(c)), d2))                 # (d2, id)
        assert order[0] == i
        assert np.array_equal(nbr[i], order[1:K + 1])
    assert (nbr != np.arange(len(c))[:, None]).all()


def test_scatter_mean_and_pooling_formulations():
    rng = np.random.default_rng(3)
    N, Nv, D, K = 4000, 700, 9, 8
    inv = torch.from_numpy(np.r_[np.arange(Nv), rng.integers(0, Nv, N - Nv)])
    Fp = torch.randn(N, D)
    m32 = affinity.scatter_mean(Fp, inv, Nv)
    cnt = np.bincount(inv.numpy(), minlength=Nv)
    ref = np.zeros((Nv, D))
    np.add.at(ref, inv.numpy(), Fp.double().numpy())
    ref /= cnt[:, None]
    assert np.abs(m32.numpy() - ref).max() < 1e-5
    nbr = torch.from_numpy(np.stack([rng.choice(Nv, K, replace=False) for _ in range(Nv)]))
    E = F.normalize(torch.randn(Nv, 6), dim=1)
    w = affinity.affinity_weights(E, nbr)
    assert torch.allclose(w.sum(1), torch.ones(Nv), atol=1e-5)
    y_sp = affinity.pool_sparse(m32, nbr, w, 5)
    y_g = affinity.pool_gather(m32, nbr, w, 5)
    y_d = affinity.pool_dense(m32, nbr, w, 5)
    assert (y_g - y_d).abs().max() < 1e-12
    assert (y_sp.double() - y_d).abs().max() < 1e-5


def test_bicubic_aa_explicit_matches_torch():
    torch.manual_seed(0)
    x = torch.randn(3, 32, 42) * 5
    ref = F.interpolate(x[None], size=(121, 162), mode="bicubic", align_corners=False, antialias=True)[0]
    mine = lift.bicubic_aa_resize_explicit(x.numpy(), (121, 162))
    # bit-exact in the build container; 2e-5 leaves room for a different CPU dispatch of the torch kernel
    assert np.abs(mine - ref.numpy()).max() < 2e-5
    # exact 4x case
    x = torch.randn(2, 30, 40)
    ref = F.interpolate(x[None], size=(120, 160), mode="bicubic", align_corners=False, antialias=True)[0]
    assert np.abs(lift.bicubic_aa_resize_explicit(x.numpy(), (120, 160)) - ref.numpy()).max() < 2e-5


def test_nn1_kdtree_vs_bruteforce():
    rng = np.random.default_rng(5)
    ref = rng.normal(size=(700, 3)).astype(np.float32)
    q = rng.normal(size=(300, 3)).astype(np.float32)
    assert np.array_equal(lift.nn1_indices(ref, q), lift.nn1_indices_bruteforce(ref, q))


def test_fuse_faithful_loops_equal_vectorised():
    rng = np.random.default_rng(6)
    N, D, C, V = 500, 16, 7, 5
    pis, fs, lgs = [], [], []
    for v in range(V):
        pi = torch.from_numpy(np.sort(rng.choice(N - 40, rng.integers(100, 300), replace=False)))
        f = F.normalize(torch.randn(len(pi), D), dim=1)
        pis.append(pi), fs.append(f), lgs.append(14.0 * (f @ F.normalize(torch.randn(C, D), dim=1).t()))
    xyz = torch.from_numpy(rng.normal(size=(N, 3)).astype(np.float32))
    a = lift.fuse_views_top3(N, pis, fs, lgs, xyz, faithful_loops=True, chunk_size=128)
    b = lift.fuse_views_top3(N, pis, fs, lgs, xyz, faithful_loops=False, chunk_size=200)
    assert (a - b).abs().max() < 1e-6
    assert (a[-40:].abs().sum(1) > 0).all()                          # never-seen points were filled


def test_lift_masks_view_full_vs_explicit_resize():
    from geopurify_amd import synthetic as syn
    cfg = syn.CONFIGS["T"]
    vlm = syn.make_vlm_outputs(cfg, 1, 11)
    rng = np.random.default_rng(8)
    n = 500
    H, W = cfg.mask_shape
    x = torch.from_numpy(rng.integers(10, H - 10, n))
    y = torch.from_numpy(rng.integers(10, W - 10, n))
    xyz = torch.from_numpy(rng.normal(size=(n, 3)).astype(np.float32))
    args = (torch.from_numpy(vlm["pred_masks"][0]), torch.from_numpy(vlm["pred_logits"][0]),
            torch.from_numpy(vlm["mask_embed"][0]), torch.from_numpy(vlm["text_embed"]),
            float(vlm["logit_scale"]), x, y, xyz, cfg.mask_shape)
    f1, l1, d1 = lift.lift_masks_view(*args, explicit_resize=False, return_debug=True)
    f2, l2, d2 = lift.lift_masks_view(*args, explicit_resize=True, return_debug=True)
    assert (f1 - f2).abs().max() < 1e-5 and (l1 - l2).abs().max() < 1e-3
    z = d1["zero_before_fill"]
    assert 0 < z.sum() < n                                           # both covered and uncovered pixels occur
    assert torch.allclose(f1.norm(dim=1), torch.ones(n), atol=1e-5)
