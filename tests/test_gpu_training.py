"""training step of the student (BatchNorm in training mode, InfoNCE, conv dgrad/wgrad, AdamW, the
anchors' point kNN and the contrastive sampler) against the torch-autograd CPU oracle (oracle/train.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import student as o_student  # noqa: E402
from oracle import train as o_train  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from geopurify_amd import ops as _ops
    return _ops


def dev(x):
    return (torch.from_numpy(x) if isinstance(x, np.ndarray) else x).cuda().contiguous()


def surface_voxels(rng, n):
    """unique integer voxels on a few planes (27-neighbourhood occupancy like a scanned room)."""
    pts = []
    while len(pts) < n:
        a, b = rng.integers(0, 60, 2)
        pts.append((a, b, 5) if rng.random() < 0.5 else (a, 7, b))
    return np.unique(np.array(pts, dtype=np.int64), axis=0)[:n]


# ------------------------------------------------------------------------------------------ kernels
def test_batchnorm_training_forward_backward(ops):
    torch.manual_seed(0)
    nv, c = 3001, 192
    y = torch.randn(nv, c) * 2 + 0.5
    res = torch.randn(nv, c)
    gamma, beta = torch.rand(c) + 0.5, torch.randn(c) * 0.1
    rm, rv = torch.zeros(c), torch.ones(c)
    yr = y.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    out_ref = F.relu(F.batch_norm(yr, rm_ref, rv_ref, gr, br, training=True, momentum=0.1, eps=1e-5) + res)
    dout = torch.randn(nv, c)
    out_ref.backward(dout)
    mean, var = ops.col_stats(dev(y))
    assert (mean.cpu() - y.mean(0)).abs().max() < 1e-6 and (var.cpu() - y.var(0, unbiased=False)).abs().max() < 1e-5
    rmd, rvd = dev(rm), dev(rv)
    out, sp = ops.bn_train_apply(dev(y), mean, var, dev(gamma), dev(beta), 1e-5, residual=dev(res), relu=True, want_split=True,
                                 momentum=0.1, running_mean=rmd, running_var=rvd)
    assert (out.cpu() - out_ref.detach()).abs().max() < 1e-5
    assert ((sp[0].float() + sp[1].float()) - out).abs().max() < 1e-6
    assert (rmd.cpu() - rm_ref).abs().max() < 1e-6 and (rvd.cpu() - rv_ref).abs().max() < 1e-5
    dy, dg, db, dz = ops.bn_train_backward(dev(dout), out, dev(y), mean, var, 1e-5, dev(gamma), want_dz=True)
    assert (dy.cpu() - yr.grad).abs().max() < 1e-5
    assert (dg.cpu() - gr.grad).abs().max() < 2e-3 and (db.cpu() - br.grad).abs().max() < 2e-3      # sums of 3001 terms
    assert torch.equal(dz.cpu(), dout * (out_ref.detach() > 0))
    # the gradient's split scale from the same sweep == the separate amax pass over dy; the unaligned (scalar) form writes the same dy
    sc2 = torch.empty(2, device="cuda")
    dy2 = ops.bn_train_backward(dev(dout), out, dev(y), mean, var, 1e-5, dev(gamma), dy_scale2=sc2)[0]
    assert torch.equal(dy2, dy) and torch.equal(sc2, ops.pow2_scale(dy)) and float(sc2[0] * sc2[1]) == 1.0
    assert 2.0 ** 13 <= float(dy.abs().max() * sc2[0]) < 2.0 ** 14
    yo = torch.zeros(nv * c + 1, device="cuda")[1:].view(nv, c)                 # 4-byte aligned only
    yo.copy_(dev(y))
    dy3 = ops.bn_train_backward(dev(dout), out, yo, mean, var, 1e-5, dev(gamma), dy_scale2=sc2)[0]
    assert torch.equal(dy3, dy) and torch.equal(sc2, ops.pow2_scale(dy))
    hi, lo = ops.split_f16(dy, scale=sc2[0:1], extra_zero_rows=1)
    assert hi.shape[0] == nv + 1 and not hi[nv].any() and not lo[nv].any()
    assert ((hi[:nv].float() + lo[:nv].float()) * sc2[1] - dy).abs().max() <= dy.abs().max() * 2.0 ** -21
    # a layer WITHOUT a residual: the ReLU mask recomputed from y (beta_mask) is the mask of its activation, bit for bit -- same dy, sums and dz;
    # the forward pass then needs no fp32 output (want_f32=False writes the same planes)
    out_n, sp_n = ops.bn_train_apply(dev(y), mean, var, dev(gamma), dev(beta), 1e-5, relu=True, want_split=True)
    none, sp_p = ops.bn_train_apply(dev(y), mean, var, dev(gamma), dev(beta), 1e-5, relu=True, want_split=True, want_f32=False)
    assert none is None and torch.equal(sp_p[0], sp_n[0]) and torch.equal(sp_p[1], sp_n[1])
    ra = ops.bn_train_backward(dev(dout), out_n, dev(y), mean, var, 1e-5, dev(gamma), want_dz=True)
    rb = ops.bn_train_backward(dev(dout), None, dev(y), mean, var, 1e-5, dev(gamma), want_dz=True, beta_mask=dev(beta))
    assert all(torch.equal(a, b) for a, b in zip(ra, rb)) and 0.2 < float((ra[3] == 0).float().mean()) < 0.8
    sa = ops.bn_bwd_sums_f64(dev(dout), out_n, dev(y), mean, var, 1e-5)
    sb = ops.bn_bwd_sums_f64(dev(dout), None, dev(y), mean, var, 1e-5, mask_affine=(dev(gamma), dev(beta)))
    assert torch.equal(sa, sb)
    # SPLIT FORM: the sweep writes hi + lo = dy * s with s from a BOUND of max |dy| taken in the reduction pass (no fp32 dy): the bound holds,
    # stays within a small factor of the true maximum, the planes carry dy to 2^-21 of that maximum, row nv is zero, sums and dz are the same
    for a_, b_ in ((out, None), (None, dev(beta))):
        yy, oo = (dev(y), a_) if a_ is not None else (dev(y), None)
        ref_r = ops.bn_train_backward(dev(dout), oo if a_ is not None else None, yy, mean, var, 1e-5, dev(gamma), want_dz=True,
                                      beta_mask=b_) if a_ is None else (dy, dg, db, dz)
        sc3 = torch.empty(2, device="cuda")
        (sh, sl), dg3, db3, dz3 = ops.bn_train_backward(dev(dout), oo, yy, mean, var, 1e-5, dev(gamma), want_dz=True, dy_scale2=sc3,
                                                        beta_mask=b_, split=True)
        dyr = ref_r[0]
        top = float(dyr.abs().max() * sc3[0])
        assert 2.0 ** 10 <= top < 2.0 ** 14, top                                  # bound >= true max, looser by less than 2^3
        assert float(sc3[0] * sc3[1]) == 1.0 and sh.shape == (nv + 1, c) and not sh[nv].any() and not sl[nv].any()
        assert ((sh[:nv].float() + sl[:nv].float()) * sc3[1] - dyr).abs().max() <= dyr.abs().max() * 2.0 ** -18
        assert torch.equal(dg3, ref_r[1]) and torch.equal(db3, ref_r[2]) and torch.equal(dz3, ref_r[3])
    da = ops.bn_bwd_apply(dev(dout), out_n, dev(y), mean, var, 1e-5, dev(gamma), sa.float(), nv)
    db_ = ops.bn_bwd_apply(dev(dout), None, dev(y), mean, var, 1e-5, dev(gamma), sa.float(), nv, beta_mask=dev(beta))
    assert torch.equal(da, db_)


def test_weight_split_transpose_flip_is_the_split_of_the_mirrored_transposed_weights(ops):
    """the data-gradient operand V[k] = W[26-k]^T taken straight from W: the same halves as the split of the flipped, transposed copy"""
    torch.manual_seed(5)
    for cin, cout in ((512, 256), (256, 544 - 32), (96, 256)):                    # the last shape has no blocked form: the fallback
        w = torch.randn(27, cin, cout, device="cuda") * 0.02
        got = ops.conv_weights_split(w, 16.0, transpose_flip=True)
        ref = ops.conv_weights_split(w.flip(0).transpose(1, 2).contiguous(), 16.0)
        assert got[0].shape == ref[0].shape and torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])


def _select_reference(sim, anchors, k):
    """arg-max over the points other than the anchor (lowest index on ties); the k lowest other than anchor and positive by (value, index)"""
    A, n = sim.shape
    pos = np.empty(A, dtype=np.int64)
    macro = np.empty((A, k), dtype=np.int64)
    for a in range(A):
        row = sim[a].astype(np.float64) + 0.0                        # -0 -> +0
        m = row.copy()
        m[anchors[a]] = -np.inf
        pos[a] = int(np.argmax(m))
        m = row.copy()
        m[anchors[a]] = np.inf
        m[pos[a]] = np.inf
        macro[a] = np.lexsort((np.arange(n), m))[:k]
    return pos, macro


@pytest.mark.parametrize("case", ["random", "ties", "one_stride", "equal_block", "small", "unaligned"])
def test_sampler_select_is_argmax_and_k_lowest(ops, case):
    """gp_sampler_select against numpy (exact: indices ordered by (value, index)); the cases drive the common path (a few candidates under
    the bound of the thread minima), the radix path (more than 4096 elements at or below the bound: all low values in ONE thread's stride;
    thousands of exactly equal lowest values, picked by index) and rows that cannot be read 16 bytes at a time."""
    rng = np.random.default_rng({"random": 0, "ties": 1, "one_stride": 2, "equal_block": 3, "small": 4, "unaligned": 5}[case])
    A, n, k, ld = 24, 50001, 48, 50004
    if case == "random":
        sim = (rng.standard_normal((A, n)) * 0.03).astype(np.float32)
        sim[0, :100] = -0.0
    elif case == "ties":
        sim = np.round(rng.standard_normal((A, n)) * 4).astype(np.float32) / 64        # ~40 distinct values
    elif case == "one_stride":
        A, n, ld = 6, 600001, 600004                                  # 586 elements per thread: 20 threads hold 11.7k low values, all at or
        sim = rng.random((A, n), dtype=np.float32) + 1.0              # below the bound (the 49th lowest thread minimum is an ordinary value)
        j = np.arange(0, n // 4, 1024)
        idx = (4 * (np.arange(20)[:, None] + j[None, :]))[:, :, None] + np.arange(4)[None, None, :]
        idx = idx.reshape(-1)
        idx = idx[idx < n]
        sim[:, idx] = -rng.random((A, len(idx)), dtype=np.float32)
    elif case == "equal_block":
        sim = rng.random((A, n), dtype=np.float32)
        sim[:, 1000:1000 + 9000] = -1.0                                                   # 9000 equal lowest values: the first 48 by index
        sim[:, 20000:20005] = -1.5
    elif case == "small":
        n, ld = 50, 52
        sim = rng.standard_normal((A, n)).astype(np.float32)
    else:
        n, ld = 4999, 5001
        sim = rng.standard_normal((A, n)).astype(np.float32)
    anchors = rng.integers(0, n, A)
    buf = torch.full((A, ld), float("nan"), device="cuda")
    buf[:, :n] = dev(sim)
    before = buf.clone()
    pos, macro = ops.sampler_select(buf, dev(anchors.astype(np.int64)), k, n=n)
    torch.cuda.synchronize()
    assert torch.equal(torch.nan_to_num(buf), torch.nan_to_num(before))
    rp, rm = _select_reference(sim, anchors, k)
    assert np.array_equal(pos.cpu().numpy(), rp)
    assert np.array_equal(macro.cpu().numpy(), rm)


def test_normalize_split_is_f_normalize(ops):
    torch.manual_seed(3)
    n, d, n_pad = 1003, 1088, 1280
    x = torch.randn(n, d) * torch.rand(n, 1) * 5
    x[7] = 0
    hi, lo = ops.normalize_split_f16(dev(x), n_pad)
    ref = F.normalize(x.double(), dim=1)
    got = hi.double().cpu() + lo.double().cpu()
    assert got.shape == (n_pad, d) and not got[n:].any() and not got[7].any()
    assert (got[:n] - ref).abs().max() < 3e-7                         # fp32 norm and quotient + the split's 2^-22 relative


def test_infonce_forward_backward(ops):
    torch.manual_seed(1)
    nv, d, S, A, Nn = 500, 128, 700, 96, 63
    E = torch.randn(nv, d)
    s2v = torch.randint(0, nv, (S,))
    p2b = torch.randint(0, S, (A * (2 + Nn),))
    Er = E.clone().requires_grad_(True)
    loss_ref = o_train.info_nce(Er[s2v], p2b, A, Nn, 0.07)
    loss_ref.backward()
    loss, dE = ops.infonce_fwd_bwd(dev(E), dev(s2v), dev(p2b), A, Nn, 0.07)
    assert abs(float(loss) - float(loss_ref.detach())) < 1e-5 * max(1.0, abs(float(loss_ref.detach())))
    assert (dE.cpu() - Er.grad).abs().max() < 1e-6 + 1e-4 * Er.grad.abs().max()


def test_adamw_matches_torch(ops):
    torch.manual_seed(2)
    p0 = torch.randn(10007)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([p_ref], lr=3e-4, weight_decay=1e-5)
    p, m, v = dev(p0.clone()), torch.zeros(10007, device="cuda"), torch.zeros(10007, device="cuda")
    for step in range(1, 4):
        g = torch.randn(10007)
        p_ref.grad = g.clone()
        opt.step()
        ops.adamw_step_(p, dev(g), m, v, 3e-4, step, weight_decay=1e-5)
    assert (p.cpu() - p_ref.detach()).abs().max() < 1e-6


def test_fused_adamw_optimizer_is_torch_adamw(ops):
    from geopurify_amd.training import FusedAdamW
    torch.manual_seed(3)
    shapes = [(27, 6, 8), (8,), (8, 4)]
    ref = [torch.randn(*s, device="cuda").requires_grad_(True) for s in shapes]
    mine = [r.detach().clone().requires_grad_(True) for r in ref]
    o_ref = torch.optim.AdamW([{"params": ref[:1], "lr": 1e-3}, {"params": ref[1:], "lr": 5e-3}], weight_decay=1e-2)
    o_mine = FusedAdamW([{"params": mine[:1], "lr": 1e-3}, {"params": mine[1:], "lr": 5e-3}], weight_decay=1e-2)
    for _ in range(4):
        for a, b in zip(ref, mine):
            g = torch.randn_like(a)
            a.grad, b.grad = g.clone(), g.clone()
        o_ref.step()
        o_mine.step()
    for a, b in zip(ref, mine):
        assert (a - b).abs().max() < 1e-6
    sd = o_mine.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 4.0


def test_knn_points_exact(ops):
    rng = np.random.default_rng(3)
    xyz = (rng.random((20000, 3)) * np.array([7, 5, 2.6])).astype(np.float32)
    xyz[100] = xyz[50]                                               # an exact duplicate pair (distance 0 tie by id)
    q = np.concatenate([[50, 100], rng.choice(20000, 200, replace=False)]).astype(np.int64)
    out, flag = ops.knn_points(dev(xyz), dev(q), 96)
    assert int(flag.item()) == 0
    assert np.array_equal(out.cpu().numpy(), o_train.knn_points_bruteforce(xyz, q, 96))
    # k + 1 > 256: the 1024-thread form
    out, flag = ops.knn_points(dev(xyz), dev(q[:20]), 300)
    assert int(flag.item()) == 0 and np.array_equal(out.cpu().numpy(), o_train.knn_points_bruteforce(xyz, q[:20], 300))


def test_knn_points_order_with_the_stride_period_takes_the_histogram_bound(ops):
    """points ordered so that a query's neighbourhood lies in a few threads' strides (index = thread + 256 j): the bound from the thread minima
    admits more than 2048 points and the kernel falls back to the distance histogram of rounds 3-5 -- same exact result"""
    rng = np.random.default_rng(8)
    n = 256 * 200
    xyz = (rng.random((n, 3)) * 10).astype(np.float32)
    near = np.concatenate([np.arange(t, n, 256) for t in range(40)])           # 8000 points in 40 strides ...
    xyz[near] = (np.array([5.0, 5.0, 5.0]) + rng.standard_normal((len(near), 3)) * 0.05).astype(np.float32)      # ... in a 5-cm cluster
    far = np.setdiff1d(np.arange(n), near)
    xyz[far] += np.where(np.abs(xyz[far] - 5.0).max(1, keepdims=True) < 1.0, 3.0, 0.0).astype(np.float32)       # nothing else near it
    q = near[:6].astype(np.int64)
    out, flag = ops.knn_points(dev(xyz), dev(q), 96)
    assert int(flag.item()) == 0
    assert np.array_equal(out.cpu().numpy(), o_train.knn_points_bruteforce(xyz, q, 96))


def test_gather_gemm_single_dense_offset_equals_the_two_phase_operator(ops):
    """plane_flags bit 4 (the sampler's anchors x points similarity): phase 1 writes the fp32 rows itself; against the two-phase call with
    fp32 partial rows (bit-identical: the same stores, then a pass that adds nothing) and against fp64"""
    torch.manual_seed(9)
    n, d, A = 5120, 160, 700
    Fn = F.normalize(torch.randn(n, d), dim=1)
    anchors = torch.randperm(n)[:A]
    hi, lo = ops.split_f16(dev(Fn))
    pairs = ops.conv_pairs_build(dev(anchors.to(torch.int32)).view(1, -1).contiguous(), chunk_rows=None)
    w = (hi.view(1, n, d), lo.view(1, n, d))
    direct = ops.sparse_conv_f16x3(None, pairs, w[0], w[1], x_split=(hi, lo), dense_single_offset=True)
    two = ops.sparse_conv_f16x3(None, pairs, w[0], w[1], x_split=(hi, lo), fp32_partials=True)
    assert direct.shape == (A, n) and torch.equal(direct, two)
    ref = Fn[anchors].double() @ Fn.double().t()
    assert (direct.double().cpu() - ref).abs().max() < 2e-6


@pytest.mark.parametrize("cin_pad", [256, 544])
def test_conv_weight_gradient_kernel(ops, cin_pad):
    """dW[k] = X[in_k]^T dY[out_k] on the matrix cores against an fp64 gather-GEMM; cin_pad = 544 exercises the
    slid-back last row tile (rows 288..543 computed, 512..543 stored), several segments per offset and padded pairs."""
    rng = np.random.default_rng(21)
    coords = surface_voxels(rng, 3000)
    nv, cout = len(coords), 256
    cs_ref = dev(coords.astype(np.int32))
    perm, rank = ops.morton_order(cs_ref)
    cs = cs_ref[perm.long()].contiguous()
    nbr_map = ops.kernel_map_build(ops.grid_build(cs), cs)
    X = torch.randn(nv, cin_pad, device="cuda")
    dY = torch.randn(nv, cout, device="cuda") * 3e-5                     # gradient-sized values (need the power-of-two scaling)
    pairs = []
    for k in range(27):
        m = nbr_map[k]
        out_rows = torch.nonzero(m >= 0).squeeze(1)
        pairs.append((out_rows, m[out_rows].long()))
    plan = ops.wgrad_plan_build(pairs, nv, steps_per_segment=8)          # many segments
    s = 2.0 ** 14
    dys = torch.zeros((nv + 1, cout), device="cuda")
    dys[:nv] = dY * s
    inv_s = torch.tensor([1.0 / s], device="cuda")
    dw = ops.conv_wgrad_f16x3(ops.split_f16(X), ops.split_f16(dys), plan, cin_pad, cin_pad, cout, inv_scale=inv_s)
    Xd, Yd = X.double(), dY.double()
    worst = 0.0
    for k, (o, i) in enumerate(pairs):
        ref = Xd[i].t() @ Yd[o] if o.numel() else torch.zeros(cin_pad, cout, dtype=torch.float64, device="cuda")
        err = float((dw[k].double() - ref).abs().max() / (ref.abs().max() + 1e-30))
        worst = max(worst, err)
    assert worst < 2e-6, worst                                           # fp32-class: 2^-22 split error, fp32 accumulation


# ------------------------------------------------------------------------------------------ student forward/backward
def _setup_student(hidden, num_blocks, seed, cin=38, nvox=1500, S=1200, A=64, Nn=63):
    from geopurify_amd import pipeline as pl
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    coords = surface_voxels(rng, nvox)
    nv = len(coords)
    sd = pl.random_student_state_dict(cin, hidden=hidden, embed=128, num_blocks=num_blocks, seed=seed)
    X = torch.randn(nv, cin) * 0.5
    s2v = torch.randint(0, nv, (S,))
    p2b = torch.randint(0, S, (A * (2 + Nn),))
    return coords, sd, X, s2v, p2b, A, Nn


@pytest.mark.parametrize("hidden,num_blocks,cin", [(128, 1, 38), (256, 2, 38), (256, 1, 518)])
def test_student_training_step_matches_autograd(ops, hidden, num_blocks, cin):
    """loss, every gradient, the AdamW-updated weights and the BatchNorm running statistics of one step.
    hidden=128: exact fp32 MFMA convolutions; hidden=256: the f16x3 matrix-core path (forward, dgrad, weight gradient);
    cin=518 (= 512 + 6, padded to 544): the input layer's weight gradient through the slid-back last row tile."""
    from geopurify_amd.training import StudentTrainer
    coords, sd, X, s2v, p2b, A, Nn = _setup_student(hidden, num_blocks, seed=5, cin=cin)
    ref = o_train.train_step_oracle(sd, X, coords, s2v, p2b, A, Nn, 0.07, num_blocks, base_lr=1e-3, weight_decay=1e-2)
    tr = StudentTrainer(sd, "cuda", base_lr=1e-3, weight_decay=1e-2)
    cs_ref = dev(coords.astype(np.int32))
    perm, rank = ops.morton_order(cs_ref)
    cs = cs_ref[perm.long()].contiguous()
    Xd = torch.zeros((len(coords), tr.cin_pad), device="cuda")
    Xd[:, :X.shape[1]] = dev(X)[perm.long()]
    nbr_map = ops.kernel_map_build(ops.grid_build(cs), cs)
    loss, grads, E = tr.forward_backward(Xd, nbr_map, rank.long()[dev(s2v)].contiguous(), dev(p2b), A, Nn)
    assert abs(float(loss) - ref["loss"]) < 2e-4 * max(1.0, abs(ref["loss"]))
    E_ref = ref["embeddings"]
    assert (E.cpu()[rank.long().cpu()] - E_ref).abs().max() < 1e-3 * E_ref.abs().max()
    for name, g_ref in ref["grads"].items():
        g = grads[name].cpu()
        if name == "input_layer.0.kernel":
            assert float(g[:, X.shape[1]:].abs().max()) == 0.0        # padded input channels never receive gradient
            g = g[:, :X.shape[1]]
        err = (g - g_ref).abs().max() / (g_ref.abs().max() + 1e-12)
        assert err < 5e-3, (name, float(err))                         # tolerance: fp32 sums in different orders, 2-5 layers deep
    tr.optimizer_step(grads)
    new = tr.state_dict()
    for name, p_ref in ref["params"].items():
        d = (new[name].cpu() - p_ref).abs()
        # the first AdamW step moves a weight by lr * g / (|g| + 1e-8): where |g| is far above the 1e-8 epsilon the
        # update is +-lr whatever the gradient's rounding; near zero it is ill-conditioned and only bounded by lr
        g_ref = ref["grads"][name]
        well = g_ref.abs() > 1e-5
        assert d[well].max() < 2e-6 if well.any() else True, (name, float(d[well].max()))
        assert d.max() <= 2.02e-3 * o_train.PARAM_GROUP_LR[o_train.param_group(name)], (name, float(d.max()))   # opposite signs at most
    for prefix, (rm, rv) in ref["bn"].items():
        assert (new[prefix + ".bn.running_mean"].cpu() - rm).abs().max() < 1e-4
        assert (new[prefix + ".bn.running_var"].cpu() - rv).abs().max() < 1e-4


def test_scene_training_step_end_to_end(ops):
    """sampler (point kNN + teacher similarities) + voxel subset + student step on a tiny scene; the oracle is fed the
    device's own anchors and replays everything else."""
    from geopurify_amd import pipeline as pl
    from geopurify_amd import synthetic as syn
    from geopurify_amd.training import StudentTrainer
    cfg = syn.CONFIGS["T"]
    scene = syn.make_scene(cfg, 77)
    rigid = pl.scene_rigid_transform(cfg.voxel_size, 77)
    batch = pl.build_scene_batch(pl.upload_scene(scene, "cuda"), rigid, "cuda")
    N = batch.scene_coords.shape[0]
    g = torch.Generator().manual_seed(9)
    D, Dt, A, Nn, K = 32, 64, 128, 63, 96                 # Dt % 32 == 0: anchor similarities on the matrix cores
    F_lift = torch.randn(N, D, generator=g)
    F_teacher = torch.randn(N, Dt, generator=g)
    anchors = torch.randperm(N, generator=g)[:A]
    sd = pl.random_student_state_dict(D + pl.GEO_DIM, hidden=128, embed=128, num_blocks=1, seed=3)
    tr = StudentTrainer(sd, "cuda", base_lr=1e-3)
    xyz = batch.scene_coords.float().contiguous()
    out = tr.scene_step(dev(F_lift), batch.scene_gauss_features, batch.scene_inds_reconstruct, batch.scene_coords_3d, xyz,
                        dev(F_teacher), dev(anchors), num_negatives=Nn, K=K, optimize=False)
    # ---- sampler parity
    nbr_ref = o_train.knn_points_bruteforce(xyz.cpu().numpy(), anchors.numpy(), K)
    assert np.array_equal(out["neighbors"].cpu().numpy(), nbr_ref)
    pos_ref, neg_ref, _ = o_train.sample_pairs(F_teacher, torch.from_numpy(nbr_ref), anchors, Nn)
    assert (out["positive"].cpu() == pos_ref).float().mean() > 0.98          # argmax / top-k near-ties inside GEMM rounding
    same = [len(set(a.tolist()) & set(b.tolist())) for a, b in zip(out["negative"].cpu(), neg_ref)]
    assert np.mean(same) > 0.98 * Nn
    # ---- everything downstream of the sampler, replayed by the oracle on the device's own samples
    pos, neg = out["positive"].cpu(), out["negative"].cpu()
    inv = batch.scene_inds_reconstruct.cpu()
    all_idx, p2b, uniq_vox, s2v = o_train.build_sample_sets(anchors, pos, neg, inv)
    gauss = batch.scene_gauss_features.cpu()
    Xv = torch.cat([o_train.scatter_mean_rows(F_lift[all_idx], s2v, len(uniq_vox)),
                    o_train.scatter_mean_rows(gauss[all_idx], s2v, len(uniq_vox))], dim=1)
    coords_v = batch.scene_coords_3d.cpu()[uniq_vox].floor().long().numpy()
    ref = o_train.train_step_oracle(sd, Xv, coords_v, s2v, p2b, A, Nn, 0.07, 1)
    assert out["num_voxels"] == len(uniq_vox) and out["num_samples"] == len(all_idx)
    assert abs(float(out["loss"]) - ref["loss"]) < 2e-4 * max(1.0, abs(ref["loss"]))
    for name in ("output_layer.kernel", "res_blocks.0.conv1.kernel", "input_layer.1.bn.weight"):
        g_ref = ref["grads"][name]
        gd = out["grads"][name].cpu()
        assert (gd - g_ref).abs().max() / (g_ref.abs().max() + 1e-12) < 5e-3, name


def test_reference_training_loop_surface(ops):
    """run/train.py:188-198,346-353 unchanged: optimizer from get_param_groups(), `loss = model(batch)`,
    `loss.backward()`, `optimizer.step()` -- the gradients come from the HIP backward pass."""
    from geopurify_amd import pipeline as pl
    from geopurify_amd import synthetic as syn
    from geopurify_amd.affinity_module import SonataXAffinityTrainer
    cfg = syn.CONFIGS["T"]
    scene = syn.make_scene(cfg, 78)
    rigid = pl.scene_rigid_transform(cfg.voxel_size, 78)
    batch = pl.build_scene_batch(pl.upload_scene(scene, "cuda"), rigid, "cuda")
    vlm = pl.SyntheticVLM(syn.make_vlm_outputs(cfg, cfg.num_views, 78), "cuda")
    N = batch.scene_coords.shape[0]
    torch.manual_seed(11)
    teacher_feats = torch.randn(N, 40, device="cuda")
    model = SonataXAffinityTrainer({"mask_shape": cfg.mask_shape, "all_label": ["c%d" % i for i in range(cfg.num_classes)]},
                                   device="cuda", use_lseg=False, vlm=vlm, feature_dim=cfg.feat_dim, hidden_dim=128,
                                   teacher=lambda b: teacher_feats).to("cuda")
    model.num_anchors_per_scene = 256
    groups = model.affinity_student.get_param_groups()
    base_lr = 1e-3
    opt = torch.optim.AdamW([{"params": groups["input"], "lr": base_lr * 0.1}, {"params": groups["middle"], "lr": base_lr},
                             {"params": groups["output"], "lr": base_lr * 5.0}], weight_decay=1e-5)
    model.train()
    w_before = model.affinity_student.output_layer.kernel.detach().clone()
    rm_before = model.affinity_student.input_layer[1].bn.running_mean.clone()
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = model(batch)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and 2.0 < losses[0] < 6.0            # ~log(64) = 4.16 at random init
    for p in model.affinity_student.parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all()
    assert not torch.equal(model.affinity_student.output_layer.kernel.detach(), w_before)
    assert not torch.equal(model.affinity_student.input_layer[1].bn.running_mean, rm_before)
    assert int(model.affinity_student.input_layer[1].bn.num_batches_tracked) == 3


def test_train_driver_loop_and_checkpoints(ops, tmp_path):
    """geopurify_amd.train_driver.train: two epochs over two tiny scenes, log scalars, checkpoints, resume."""
    from geopurify_amd import config as gp_config
    from geopurify_amd import pipeline as pl
    from geopurify_amd import synthetic as syn
    from geopurify_amd import train_driver as td
    from geopurify_amd.affinity_module import SonataXAffinityTrainer
    cfg = syn.CONFIGS["T"]
    args = gp_config.CfgNode({"mask_shape": list(cfg.mask_shape), "epochs": 2, "save_path": str(tmp_path), "save_freq": 1, "print_freq": 1})
    (tmp_path / "model").mkdir()
    model = SonataXAffinityTrainer(args, device="cuda", use_lseg=False, feature_dim=cfg.feat_dim, hidden_dim=128,
                                   allow_deferred_vlm=True).to("cuda")
    model.num_anchors_per_scene = 128
    batches = []
    for i in range(2):
        scene = syn.make_scene(cfg, 90 + i)
        batches.append((pl.build_scene_batch(pl.upload_scene(scene, "cuda"), pl.scene_rigid_transform(cfg.voxel_size, 90 + i), "cuda"),
                        pl.SyntheticVLM(syn.make_vlm_outputs(cfg, cfg.num_views, 90 + i), "cuda"),
                        torch.randn(cfg.num_points, 24, device="cuda")))

    class Loader:
        def __len__(self):
            return len(batches)

        def __iter__(self):
            for b, vlm, feats in batches:
                model.vlm, model.teacher = vlm, (lambda _b, f=feats: f)
                yield b

    opt = td.build_optimizer(model.affinity_student, 1e-3, 1e-5)
    sched = td.build_scheduler(opt, 1e-3, 1, 2, 2)
    scalars = td.train(model, opt, sched, Loader(), args)
    assert set(scalars) == {"lr", "loss_train"} and set(scalars["loss_train"]) == {1, 2}
    assert all(np.isfinite(v) for v in scalars["loss_train"].values())
    assert (tmp_path / "model" / "affinity_predictor_last.pth").exists() and (tmp_path / "model" / "affinity_predictor_epoch_1.pth").exists()
    opt2 = td.build_optimizer(model.affinity_student, 1e-3, 1e-5)
    start, sc = td.load_resume(model.affinity_student, opt2, str(tmp_path / "model" / "affinity_predictor_epoch_1.pth"), "cuda")
    assert start == 2 and sc["loss_train"] == scalars["loss_train"]


def test_two_rank_sync_batchnorm_training_step(ops):
    """run/train.py:206-213 (DDP + SyncBatchNorm): two ranks on this box's one GPU (gloo) train on different scenes; the
    mean loss, the all-reduced gradients and the running statistics equal a single process on the concatenated scene."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "syncbn_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-3000:])
