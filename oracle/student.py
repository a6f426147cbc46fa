"""Oracle row 9: Student Affinity Network forward (submanifold sparse 3D conv) (test infra).

Follows models/affinity_module.py:33-72 (MinkowskiResBlock / AffinityPredictor) and :1541-1547
(SparseTensor creation, forward, F.normalize).  MinkowskiEngine is absent from /root/reference:
"parity unpinned" for the kernel-offset <-> kernel[k] order; adopted order (upstream ME, first
spatial axis fastest):  k = (dx+1) + 3(dy+1) + 9(dz+1),  Y[u] = sum_k X[u + o_k] @ W[k].

state_dict keys (ME layout): input_layer.0.kernel [27,Cin,H]; input_layer.1.bn.{weight,bias,
running_mean,running_var}; res_blocks.{0-3}.{conv1,conv2}.kernel [27,H,H];
res_blocks.{0-3}.{norm1,norm2}.bn.*; output_layer.kernel [H,E].
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def kernel_offsets():
    """[27,3] int offsets in ME order (x fastest)."""
    offs = []
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                offs.append((dx, dy, dz))
    return np.array(offs, dtype=np.int64)


def build_kernel_map(coords):
    """coords int [Nv,3] unique.  Returns nbr_map int64 [27,Nv]: row of voxel u+o_k or -1."""
    c = np.asarray(coords).astype(np.int64)
    lo = c.min(0) - 1
    ext = c.max(0) - lo + 2
    def key(a):
        a = a - lo
        return (a[:, 0] * ext[1] + a[:, 1]) * ext[2] + a[:, 2]
    k0 = key(c)
    order = np.argsort(k0, kind="stable")
    ks = k0[order]
    out = np.full((27, c.shape[0]), -1, dtype=np.int64)
    for k, o in enumerate(kernel_offsets()):
        q = key(c + o)
        pos = np.searchsorted(ks, q)
        pos = np.minimum(pos, len(ks) - 1)
        hit = ks[pos] == q
        out[k, hit] = order[pos[hit]]
    return out


def sparse_conv3(X, nbr_map, W):
    """Gather-GEMM-scatter per offset, fp32 (what ME does): Y[u] += X[nbr_k(u)] @ W[k]."""
    Y = torch.zeros((X.shape[0], W.shape[2]), dtype=X.dtype)
    for k in range(27):
        m = torch.from_numpy(nbr_map[k])
        out_idx = torch.where(m >= 0)[0]
        if len(out_idx) == 0:
            continue
        Y.index_add_(0, out_idx, X[m[out_idx]] @ W[k])
    return Y


def bn_eval(x, p, prefix):
    w, b = p[prefix + ".bn.weight"], p[prefix + ".bn.bias"]
    mu, var = p[prefix + ".bn.running_mean"], p[prefix + ".bn.running_var"]
    return F.batch_norm(x, mu, var, w, b, training=False, eps=BN_EPS)


def student_forward(X, coords, sd, num_blocks=4, dtype=torch.float32):
    """AffinityPredictor.forward + F.normalize(p=2, dim=1).  X [Nv,Cin], coords int [Nv,3]."""
    p = {k: v.to(dtype) for k, v in sd.items() if v.is_floating_point()}
    X = X.to(dtype)
    nm = build_kernel_map(coords)
    out = F.relu(bn_eval(sparse_conv3(X, nm, p["input_layer.0.kernel"]), p, "input_layer.1"))
    for i in range(num_blocks):
        idt = out
        o = F.relu(bn_eval(sparse_conv3(out, nm, p[f"res_blocks.{i}.conv1.kernel"]), p,
                           f"res_blocks.{i}.norm1"))
        o = bn_eval(sparse_conv3(o, nm, p[f"res_blocks.{i}.conv2.kernel"]), p, f"res_blocks.{i}.norm2")
        out = F.relu(o + idt)
    Y = out @ p["output_layer.kernel"]
    return F.normalize(Y, p=2, dim=1)


def sparse_conv3_dense_check(X, coords, W):
    """Independent dense check (small grids only): densify, conv3d(padding=1, no bias) with
    Wt[o,i,az,ay,ax] = W[ax + 3 ay + 9 az, i, o] (cross-correlation taps = +offset), read back."""
    c = torch.as_tensor(np.asarray(coords)).long()
    c = c - c.min(0).values
    sx, sy, sz = (c.max(0).values + 1).tolist()
    Cin, Cout = W.shape[1], W.shape[2]
    vol = torch.zeros((1, Cin, sz, sy, sx), dtype=X.dtype)
    vol[0, :, c[:, 2], c[:, 1], c[:, 0]] = X.t()
    Wt = W.reshape(3, 3, 3, Cin, Cout).permute(4, 3, 0, 1, 2).contiguous()   # [o,i,dz,dy,dx]
    out = F.conv3d(vol, Wt, padding=1)
    return out[0, :, c[:, 2], c[:, 1], c[:, 0]].t().contiguous()


def random_student_state_dict(input_dim, hidden=512, embed=128, num_blocks=4, seed=0):
    """He-normal kernels, BN gamma=1 beta=0 with mildly random running stats."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def kern(ci, co, kv=27):
        std = (2.0 / (kv * ci)) ** 0.5
        shape = (kv, ci, co) if kv > 1 else (ci, co)
        return torch.randn(shape, generator=g) * std

    def bn(prefix, c):
        sd[prefix + ".bn.weight"] = 1.0 + 0.1 * torch.randn(c, generator=g)
        sd[prefix + ".bn.bias"] = 0.1 * torch.randn(c, generator=g)
        sd[prefix + ".bn.running_mean"] = 0.1 * torch.randn(c, generator=g)
        sd[prefix + ".bn.running_var"] = 1.0 + 0.2 * torch.rand(c, generator=g)
        sd[prefix + ".bn.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    sd["input_layer.0.kernel"] = kern(input_dim, hidden)
    bn("input_layer.1", hidden)
    for i in range(num_blocks):
        sd[f"res_blocks.{i}.conv1.kernel"] = kern(hidden, hidden)
        bn(f"res_blocks.{i}.norm1", hidden)
        sd[f"res_blocks.{i}.conv2.kernel"] = kern(hidden, hidden)
        bn(f"res_blocks.{i}.norm2", hidden)
    sd["output_layer.kernel"] = kern(hidden, embed, kv=1)
    return sd
