"""Oracle rows 8, 10, 11, 12: point->voxel mean, exact kNN, cosine affinity, pooling (test infra).

Follows models/affinity_module.py:1524-1536 (torch_scatter.scatter_mean), :1551-1557
(faiss.IndexFlatL2.search, K+1 then drop column 0), :1559-1572 (cosine affinity, softmax x20),
:1575-1589 (sparse COO A, 1+18 torch.sparse.mm, gather to points, first D columns).
torch_scatter and faiss are absent from /root/reference: "parity unpinned" at those boundaries.
"""
import numpy as np
import torch
import torch.nn.functional as F


def scatter_mean(src, index, dim_size=None, dtype=None):
    """torch_scatter.scatter_mean(src, index, dim=0): sum / clamp(count, min=1)."""
    if dim_size is None:
        dim_size = int(index.max().item()) + 1
    s = src if dtype is None else src.to(dtype)
    out = torch.zeros((dim_size, s.shape[1]), dtype=s.dtype)
    out.index_add_(0, index, s)
    cnt = torch.zeros(dim_size, dtype=s.dtype)
    cnt.index_add_(0, index, torch.ones(index.shape[0], dtype=s.dtype))
    cnt.clamp_(min=1)
    return out / cnt[:, None]


def knn_lattice(voxel_coords, K, chunk=1024):
    """Exact (K+1)-NN on integer voxel coordinates, canonical rule: the K+1 smallest by
    (squared distance, id) ascending; column 0 (self) dropped (affinity_module.py:1551-1557 +
    SURVEY 8a' tie rule).  Returns int64 [Nv,K]."""
    c = torch.as_tensor(np.asarray(voxel_coords)).to(torch.int64)
    Nv = c.shape[0]
    assert Nv > K, "need more than K voxels"
    ids = torch.arange(Nv, dtype=torch.int64)
    out = torch.empty((Nv, K), dtype=torch.int64)
    for s in range(0, Nv, chunk):
        q = c[s:s + chunk]
        d2 = ((q[:, None, :] - c[None, :, :]) ** 2).sum(-1)           # exact int64
        key = d2 * Nv + ids[None, :]                                   # lexicographic (d2, id)
        top = torch.topk(key, K + 1, dim=1, largest=False, sorted=True).indices
        out[s:s + chunk] = top[:, 1:]
    return out


def knn_kdtree(voxel_coords, K):
    """Fast CPU kNN for the timing baseline only (scipy cKDTree, all cores): exact distances, but ties
    are broken in tree order, not by (d2, id) -- never used as a parity reference."""
    from scipy.spatial import cKDTree
    c = np.asarray(voxel_coords, dtype=np.float64)
    _, idx = cKDTree(c).query(c, k=K + 1, workers=-1)
    return torch.from_numpy(idx[:, 1:].astype(np.int64))


def affinity_weights(E, nbr, sharpen=20.0):
    """affinity_module.py:1559-1572: a_ij = <E_i, E_nbr(i,j)>, w = softmax_j(sharpen * a_ij)."""
    Nv, K = nbr.shape
    center = E.repeat_interleave(K, dim=0)
    neigh = E[nbr.flatten()]
    a = torch.einsum("bd,bd->b", center, neigh)
    return F.softmax(a.view(Nv, K) * sharpen, dim=1)


def pool_sparse(X, nbr, w, num_iters=19):
    """affinity_module.py:1575-1587: A = sparse_coo(rows, nbr, w); X <- A @ X, num_iters times."""
    Nv, K = nbr.shape
    rows = torch.arange(Nv).repeat_interleave(K)
    A = torch.sparse_coo_tensor(torch.stack([rows, nbr.flatten()]), w.flatten(), size=(Nv, Nv))
    Y = torch.sparse.mm(A, X)
    for _ in range(num_iters - 1):
        Y = torch.sparse.mm(A, Y)
    return Y


def pool_gather(X, nbr, w, num_iters=19, dtype=torch.float64, chunk=8192):
    """Independent formulation: Y[i] = sum_j w[i,j] * X[nbr[i,j]] by explicit gather, in `dtype`."""
    Y = X.to(dtype)
    wd = w.to(dtype)
    for _ in range(num_iters):
        Z = torch.empty_like(Y)
        for s in range(0, Y.shape[0], chunk):
            g = Y[nbr[s:s + chunk]]                    # [c,K,D]
            Z[s:s + chunk] = (wd[s:s + chunk, :, None] * g).sum(1)
        Y = Z
    return Y


def pool_dense(X, nbr, w, num_iters=19, dtype=torch.float64):
    """Second independent check (small Nv only): dense A @ X."""
    Nv = X.shape[0]
    A = torch.zeros((Nv, Nv), dtype=dtype)
    A.scatter_(1, nbr, w.to(dtype))
    Y = X.to(dtype)
    for _ in range(num_iters):
        Y = A @ Y
    return Y
