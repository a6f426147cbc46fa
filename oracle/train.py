"""Oracle for the training step of the Student Affinity Network -- TEST INFRASTRUCTURE ONLY.

CPU restatement with torch autograd of
  * models/affinity_module.py:1099-1136  sample_contrastive_pairs_hybrid (anchors given: randperm is the
    caller's RNG draw),
  * models/affinity_module.py:1138-1237  SonataXAffinityTrainer.forward: sampled points -> voxel subset ->
    student (BatchNorm in TRAINING mode: batch statistics, running-stat update) -> InfoNCE,
  * run/train.py:188-198,346-353         AdamW with three parameter groups (0.1x / 1x / 5x base lr).
Deviations recorded in SURVEY section 3.3: the released forward feeds 512 channels into the 518-channel input layer
and cannot run; like evaluate_scene, the voxel input here is [mean lifted feature (D) | mean geometry (6)].
MinkowskiEngine, faiss and torch_scatter are absent from /root/reference: parity unpinned at those
boundaries (ME: submanifold conv = oracle.student.sparse_conv3; MinkowskiBatchNorm = BatchNorm1d over the rows).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import student as o_student

N_MACRO = 48                      # affinity_module.py:1122


def knn_points_bruteforce(xyz, queries, K):
    """faiss.IndexFlatL2.search(xyz[queries], K+1)[:, 1:] restated exactly: (d^2, id) order in fp64 of the fp32
    coordinates, the query point itself (distance 0, its own id first among ties at 0) dropped as the reference
    drops column 0.  xyz fp32 [N,3] numpy, queries int64 [A]."""
    x = np.asarray(xyz, dtype=np.float64)
    out = np.empty((len(queries), K), dtype=np.int64)
    ids = np.arange(len(x))
    for i, q in enumerate(np.asarray(queries)):
        d = ((x - x[q]) ** 2).sum(1)
        order = np.lexsort((ids, d))[: K + 1]
        out[i] = order[1:]
    return out


def sample_pairs(F_teacher, neighbor_indices, anchor_indices, num_negatives):
    """affinity_module.py:1113-1136 after the randperm.  neighbor_indices [A,K] are the anchors' rows."""
    Fn = F.normalize(F_teacher, p=2, dim=1)
    sim = Fn[anchor_indices] @ Fn.t()                                  # einsum('ad,pd->ap')
    A, N = sim.shape
    pos_sim = sim.clone()
    pos_sim.scatter_(1, anchor_indices.unsqueeze(1), float("-inf"))
    positive = torch.argmax(pos_sim, dim=1)
    n_micro = num_negatives - N_MACRO
    ar = torch.arange(N).unsqueeze(0)
    excl = (ar == anchor_indices.unsqueeze(1)) | (ar == positive.unsqueeze(1))
    neg_sim = sim.clone()
    neg_sim[excl] = float("inf")
    _, macro = torch.topk(neg_sim, k=N_MACRO, largest=False, dim=1)
    # the reference gathers the local similarities from the matrix AFTER the in-place +inf marking (:1125,1129)
    sims_local = torch.gather(neg_sim, 1, neighbor_indices)
    _, hardest = torch.topk(sims_local, k=n_micro, largest=False, dim=1)
    micro = torch.gather(neighbor_indices, 1, hardest)
    return positive, torch.cat([macro, micro], dim=1), sim


def student_train_forward(X, nbr_map, params, bn_state, num_blocks, momentum=0.1):
    """AffinityPredictor.forward with BatchNorm in training mode.  params: dict of leaf tensors (requires_grad);
    bn_state: dict prefix -> (running_mean, running_var) updated in place like nn.BatchNorm1d."""
    def bn(x, prefix):
        rm, rv = bn_state[prefix]
        return F.batch_norm(x, rm, rv, params[prefix + ".bn.weight"], params[prefix + ".bn.bias"], training=True,
                            momentum=momentum, eps=o_student.BN_EPS)

    def conv(x, name):
        W = params[name]
        out = torch.zeros((x.shape[0], W.shape[2]), dtype=x.dtype)
        for k in range(27):
            m = torch.from_numpy(nbr_map[k])
            rows = torch.where(m >= 0)[0]
            if len(rows):
                out = out.index_add(0, rows, x[m[rows]] @ W[k])
        return out

    out = F.relu(bn(conv(X, "input_layer.0.kernel"), "input_layer.1"))
    for i in range(num_blocks):
        idt = out
        o = F.relu(bn(conv(out, f"res_blocks.{i}.conv1.kernel"), f"res_blocks.{i}.norm1"))
        o = bn(conv(o, f"res_blocks.{i}.conv2.kernel"), f"res_blocks.{i}.norm2")
        out = F.relu(o + idt)
    return out @ params["output_layer.kernel"]


def info_nce(E_samples, point_to_batch_map, num_anchors, num_negatives, temperature):
    """affinity_module.py:1219-1233."""
    En = F.normalize(E_samples, p=2, dim=1)
    a = En[point_to_batch_map[:num_anchors]]
    p = En[point_to_batch_map[num_anchors:2 * num_anchors]]
    n = En[point_to_batch_map[2 * num_anchors:]].reshape(num_anchors, num_negatives, -1)
    l_pos = torch.einsum("bd,bd->b", a, p).unsqueeze(-1)
    l_neg = torch.einsum("bd,bnd->bn", a, n)
    logits = torch.cat([l_pos, l_neg], dim=1) / temperature
    return F.cross_entropy(logits, torch.zeros(num_anchors, dtype=torch.long))


def build_sample_sets(anchor, positive, negative, inds_reconstruct):
    """affinity_module.py:1196-1203: unique sampled points, their voxels, the two inverse maps."""
    all_idx, point_to_batch = torch.unique(torch.cat([anchor, positive, negative.flatten()]), return_inverse=True)
    vox = inds_reconstruct[all_idx]
    uniq_vox, sample_to_voxel = torch.unique(vox, return_inverse=True)
    return all_idx, point_to_batch, uniq_vox, sample_to_voxel


def scatter_mean_rows(x, index, n):
    out = torch.zeros((n, x.shape[1]), dtype=x.dtype).index_add_(0, index, x)
    cnt = torch.zeros(n, dtype=x.dtype).index_add_(0, index, torch.ones(len(index), dtype=x.dtype))
    return out / cnt.clamp(min=1).unsqueeze(1)


PARAM_GROUP_LR = {"input": 0.1, "middle": 1.0, "output": 5.0}            # run/train.py:193-195


def param_group(name):
    return "input" if name.startswith("input_layer") else ("output" if name.startswith("output_layer") else "middle")


def adamw_step(params, grads, state, step, base_lr, weight_decay, lr_factor=1.0, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.AdamW (defaults of run/train.py:198) written out; state: name -> (exp_avg, exp_avg_sq)."""
    new = {}
    for name, p in params.items():
        g = grads[name]
        lr = base_lr * PARAM_GROUP_LR[param_group(name)] * lr_factor
        m, v = state.get(name, (torch.zeros_like(p), torch.zeros_like(p)))
        p2 = p * (1 - lr * weight_decay)
        m = betas[0] * m + (1 - betas[0]) * g
        v = betas[1] * v + (1 - betas[1]) * g * g
        bc1, bc2 = 1 - betas[0] ** step, 1 - betas[1] ** step
        denom = (v.sqrt() / (bc2 ** 0.5)) + eps
        new[name] = p2 - (lr / bc1) * (m / denom)
        state[name] = (m, v)
    return new


def lr_factor(step_index, warmup_iters, main_iters, eta_min_ratio=1e-3):
    """SequentialLR(LinearLR(1e-6 -> 1 over warmup_iters), CosineAnnealingLR(T_max=main_iters, eta_min=1e-3 base))
    as a multiplier of each group's lr at optimizer step number `step_index` (0-based) (run/train.py:320-325).
    eta_min is base_lr*1e-3 for EVERY group (absolute), so the caller applies it per group: returns (kind, value)."""
    if step_index < warmup_iters:
        return "scale", 1e-6 + (1.0 - 1e-6) * step_index / warmup_iters
    t = step_index - warmup_iters
    return "cosine", 0.5 * (1 + np.cos(np.pi * t / main_iters))


def train_step_oracle(sd, X_vox, coords_vox, sample_to_voxel, point_to_batch, num_anchors, num_negatives, temperature,
                      num_blocks, base_lr=1e-4, weight_decay=1e-5, step=1, opt_state=None, lr_factor_value=1.0):
    """One optimisation step on one scene.  sd: ME-layout state_dict (fp32); X_vox [Nv_s, Cin]; coords_vox int [Nv_s,3].
    Returns dict(loss, grads, params (updated), bn (updated running stats), embeddings)."""
    names = [k for k in sd if k.endswith("kernel") or k.endswith(".bn.weight") or k.endswith(".bn.bias")]
    params = {k: sd[k].clone().float().requires_grad_(True) for k in names}
    bn_state = {k[:-len(".bn.running_mean")]: (sd[k].clone().float(), sd[k.replace("running_mean", "running_var")].clone().float())
                for k in sd if k.endswith(".bn.running_mean")}
    nbr_map = o_student.build_kernel_map(np.asarray(coords_vox))
    E = student_train_forward(X_vox.float(), nbr_map, params, bn_state, num_blocks)
    loss = info_nce(E[sample_to_voxel], point_to_batch, num_anchors, num_negatives, temperature)
    grads_list = torch.autograd.grad(loss, [params[k] for k in names])
    grads = dict(zip(names, grads_list))
    state = {} if opt_state is None else opt_state
    new = adamw_step({k: v.detach() for k, v in params.items()}, grads, state, step, base_lr, weight_decay, lr_factor_value)
    return {"loss": float(loss.detach()), "grads": grads, "params": new, "bn": bn_state, "embeddings": E.detach(), "opt_state": state}
