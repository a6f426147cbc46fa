"""Parity of the BENCHMARKED student path: StudentWeights(mode="f16x3") at hidden 256 and 512 -- nine chained 3x3x3
layers on the f16 matrix cores (hi/lo split operands, pre-split hand-off between layers, residuals, chunked pair
lists) -- against the fp64 oracle (oracle/student.py, models/affinity_module.py:33-72,1541-1547), then the affinity
weights and the pooled features computed from those embeddings (affinity_module.py:1559-1589)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import affinity as o_aff  # noqa: E402
from oracle import student as o_student  # noqa: E402


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from geopurify_amd import _lib, ops, pipeline
    _lib.load()
    return ops, pipeline


def _surface(rng, n):
    """~n voxels on two sheets and a wall (27-neighbour occupancy ~8, like a scanned room)."""
    from test_gpu_kernels import surface_voxels
    return surface_voxels(rng, n)


@pytest.mark.parametrize("hidden,chunk_rows", [(256, 1024), (512, 1024), (512, None)])
def test_f16x3_student_chain_vs_fp64_oracle(env, hidden, chunk_rows):
    ops, pl = env
    rng = np.random.default_rng(100 + hidden)
    coords = _surface(rng, 2200)
    Nv, D, K, T = len(coords), 512, 96, 19
    assert 3000 <= Nv <= 5500, Nv
    # input rows shaped like evaluate_scene's: voxel means of unit-norm 512-d features (|x| ~ 0.04) and rgb/normal columns
    sem = torch.from_numpy(rng.normal(0, 1, size=(Nv, D)).astype(np.float32))
    sem = torch.nn.functional.normalize(sem, dim=1) * torch.from_numpy(rng.uniform(0.3, 1.0, size=(Nv, 1)).astype(np.float32))
    geo = torch.from_numpy(np.c_[rng.uniform(0, 1, size=(Nv, 3)), rng.normal(0, 0.6, size=(Nv, 3))].astype(np.float32))
    X = torch.cat([sem, geo], 1)
    sd = pl.random_student_state_dict(D + 6, hidden=hidden, embed=128, num_blocks=4, seed=hidden)
    E64 = o_student.student_forward(X, coords.astype(np.int64), sd, num_blocks=4, dtype=torch.float64)

    dev = "cuda"
    st = pl.StudentWeights(sd, dev, mode="f16x3")
    assert st.cin_pad == 544 and all(l[0] == "f16x3" for l in st.layers) and len(st.layers) == 9
    c = torch.from_numpy(coords).to(dev).contiguous()
    perm, rank = ops.morton_order(c)
    cs = c[perm.long()].contiguous()
    grid = ops.grid_build(cs)
    nbr_map = ops.kernel_map_build(grid, cs)
    pairs = ops.conv_pairs_build(nbr_map, chunk_rows=chunk_rows)
    assert (pairs.num_chunks > 1) == (chunk_rows is not None)
    Xd = torch.zeros((Nv, st.cin_pad), device=dev)
    Xd[:, :D + 6] = X.to(dev)[perm.long()]
    E = st.forward(Xd, nbr_map, pairs=pairs)
    E_ref = E64[perm.long().cpu()]
    e_err = (E.cpu().double() - E_ref).abs().max().item()
    assert e_err <= 1e-5, e_err                                       # unit-norm embeddings: tolerance 1e-5 absolute vs fp64
    # the exact-fp32 kernels on the same weights: the f16x3 chain must be in the same accuracy class
    E32 = pl.StudentWeights(sd, dev, mode="f32").forward(Xd, nbr_map)
    e32_err = (E32.cpu().double() - E_ref).abs().max().item()
    assert e_err <= 4 * max(e32_err, 5e-7), (e_err, e32_err)

    # downstream: affinity weights from those embeddings (2e-6) and 19 pooling applications (1e-4) in the oracle's row order
    nbr_o = o_aff.knn_lattice(coords.astype(np.int64), K)
    w_ref = o_aff.affinity_weights(E64.float(), nbr_o, 20.0)
    nbr = ops.knn_lattice(grid, cs, perm, K)                          # ids in the caller's (hash-order) numbering? -> Morton rows
    w = ops.affinity_softmax(E, nbr, 20.0)
    # nbr holds Morton rows; map the oracle's lists into Morton rows and compare as sets per row, weights by neighbour
    rk = rank.long().cpu()
    nbr_o_m = rk[nbr_o][perm.long().cpu()]                           # [Morton row, K] Morton ids of the oracle's neighbours
    got = torch.sort(nbr.long().cpu(), dim=1)
    exp = torch.sort(nbr_o_m, dim=1)
    assert torch.equal(got.values, exp.values)                        # neighbour sets bit-exact
    w_got = torch.gather(w.cpu(), 1, got.indices)
    w_exp = torch.gather(w_ref[perm.long().cpu()], 1, exp.indices)
    assert (w_got - w_exp).abs().max().item() <= 2e-6
    hp = pl.HotPath(st, (8, 8), K=K, num_iters=T, device=dev, pool_mode="mfma")
    Y = hp._pool(Xd, nbr, w, Nv, D)
    Y_ref = o_aff.pool_gather(X[:, :D], nbr_o, w_ref, T)[perm.long().cpu()]
    p_err = (Y.cpu().double() - Y_ref).abs().max().item()
    assert p_err <= 1e-4, p_err                                       # north-star tolerance: pooled features 1e-4
