// kernels of the student's TRAINING step that are not convolutions
// (models/affinity_module.py:1138-1237 SonataXAffinityTrainer.forward, run/train.py:346-353).
//   * BatchNorm over the voxel rows in training mode (ME.MinkowskiBatchNorm = BatchNorm1d on the feature rows):
//     column statistics, normalise + affine (+ residual) + ReLU, and the backward pass;
//   * InfoNCE over (anchor, positive, 63 negatives) rows of the embedding, forward and backward fused;
//   * AdamW update (torch.optim.AdamW semantics);
//   * exact K nearest points of the anchor points (the reference: faiss.IndexFlatL2 on the CPU over all points;
//     only the anchors' rows are ever used, affinity_module.py:1127).
// All of them are HBM-bound row/column sweeps; the convolutions (forward, dgrad = the same kernel with mirrored,
// transposed weights) stay in sparse_conv_v2.hip.
#include "gp_common.h"

namespace {

constexpr int CS_ROWS = 256;      // rows per workgroup of the column reductions

// ------------------------------------------------------------------------------------------------ column sums
// partial[chunk][q][c] = sum over the chunk's rows of f_q(row, c), q < NQ; fp64 accumulation, fixed order.
template <int NQ, typename F>
__device__ __forceinline__ void col_partial(int64_t nv, int c, double *__restrict__ partial, F f) {
    __shared__ double red[4][NQ][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = blockIdx.y * 64 + lane;
    const int64_t r0 = (int64_t)blockIdx.x * CS_ROWS;
    double acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = 0.0;
    if (col < c)
        for (int64_t r = r0 + wv; r < r0 + CS_ROWS && r < nv; r += 4) {
            double v[NQ];
            f(r, col, v);
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] += v[q];
        }
#pragma unroll
    for (int q = 0; q < NQ; ++q) red[wv][q][lane] = acc[q];
    __syncthreads();
    if (wv == 0 && col < c)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            partial[((int64_t)blockIdx.x * NQ + q) * c + col] = red[0][q][lane] + red[1][q][lane] + red[2][q][lane] + red[3][q][lane];
}

__global__ void __launch_bounds__(256) cs_sum_kernel(const float *__restrict__ y, int64_t ld, int64_t nv, int c, double *__restrict__ partial) {
    col_partial<1>(nv, c, partial, [&](int64_t r, int col, double *v) { v[0] = (double)y[r * ld + col]; });
}
__global__ void __launch_bounds__(256) cs_var_kernel(const float *__restrict__ y, int64_t ld, int64_t nv, int c,
                                                     const float *__restrict__ mean, double *__restrict__ partial) {
    col_partial<1>(nv, c, partial, [&](int64_t r, int col, double *v) { double d = (double)y[r * ld + col] - (double)mean[col]; v[0] = d * d; });
}
// out[q][c] = scale * sum over chunks: one workgroup per 64 columns, the chunks strided over 4 waves (fixed order)
__global__ void __launch_bounds__(256) cs_final_kernel(const double *__restrict__ partial, int64_t nchunks, int nq, int c, double scale,
                                                       float *__restrict__ out) {
    __shared__ double red[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;                 // flat (q, col)
    double s = 0.0;
    if (i < nq * c) {
        const int q = i / c, col = i % c;
        for (int64_t k = wv; k < nchunks; k += 4) s += partial[(k * nq + q) * c + col];
    }
    red[wv][lane] = s;
    __syncthreads();
    if (wv == 0 && i < nq * c) out[i] = (float)((red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) * scale);
}

// ------------------------------------------------------------------------------------------------ BN forward / backward
__global__ void bn_apply_kernel(const float *__restrict__ y, int64_t ld, int64_t nv, int c, const float *__restrict__ mean,
                                const float *__restrict__ var, const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                const float *__restrict__ residual, int64_t ld_res, int relu, float *__restrict__ out, int64_t ld_out,
                                _Float16 *__restrict__ out_hi, _Float16 *__restrict__ out_lo, int64_t ld_sp) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= nv * c) return;
    int64_t r = i / c;
    int col = (int)(i % c);
    float invstd = 1.0f / sqrtf(var[col] + eps);
    float v = (y[r * ld + col] - mean[col]) * invstd * gamma[col] + beta[col];
    if (residual) v += residual[r * ld_res + col];
    if (relu) v = v > 0.f ? v : 0.f;
    out[r * ld_out + col] = v;
    if (out_hi) {
        _Float16 h = (_Float16)v;
        out_hi[r * ld_sp + col] = h;
        out_lo[r * ld_sp + col] = (_Float16)(v - (float)h);
    }
}
__global__ void bn_running_kernel(const float *__restrict__ mean, const float *__restrict__ var, int c, int64_t nv, float momentum,
                                  float *__restrict__ running_mean, float *__restrict__ running_var) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    float unbiased = nv > 1 ? var[i] * ((float)nv / (float)(nv - 1)) : var[i];
    running_mean[i] = (1.f - momentum) * running_mean[i] + momentum * mean[i];
    running_var[i] = (1.f - momentum) * running_var[i] + momentum * unbiased;
}
// dz = dout * (act > 0) (act == nullptr: no mask); sums of dz and dz * xhat
__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(const float *__restrict__ dout, int64_t ld_d, const float *__restrict__ act, int64_t ld_a,
                                                            const float *__restrict__ y, int64_t ld_y, const float *__restrict__ mean,
                                                            const float *__restrict__ var, float eps, int64_t nv, int c, double *__restrict__ partial) {
    col_partial<2>(nv, c, partial, [&](int64_t r, int col, double *v) {
        float dz = dout[r * ld_d + col];
        if (act && !(act[r * ld_a + col] > 0.f)) dz = 0.f;
        float xhat = (y[r * ld_y + col] - mean[col]) * (1.0f / sqrtf(var[col] + eps));
        v[0] = (double)dz;
        v[1] = (double)dz * (double)xhat;
    });
}
__global__ void bn_bwd_apply_kernel(const float *__restrict__ dout, int64_t ld_d, const float *__restrict__ act, int64_t ld_a,
                                    const float *__restrict__ y, int64_t ld_y, const float *__restrict__ mean, const float *__restrict__ var,
                                    float eps, const float *__restrict__ gamma, const float *__restrict__ sums, int64_t n_total, int64_t nv, int c,
                                    float *__restrict__ dy, int64_t ld_dy, float *__restrict__ dz_out, int64_t ld_dz) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= nv * c) return;
    int64_t r = i / c;
    int col = (int)(i % c);
    float dz = dout[r * ld_d + col];
    if (act && !(act[r * ld_a + col] > 0.f)) dz = 0.f;
    float invstd = 1.0f / sqrtf(var[col] + eps);
    float xhat = (y[r * ld_y + col] - mean[col]) * invstd;
    float inv_n = 1.0f / (float)n_total;
    dy[r * ld_dy + col] = gamma[col] * invstd * (dz - sums[col] * inv_n - xhat * sums[c + col] * inv_n);
    if (dz_out) dz_out[r * ld_dz + col] = dz;
}

// ------------------------------------------------------------------------------------------------ InfoNCE
// En[s] = E[s2v[s]] / max(|E[s2v[s]]|, 1e-12)   (F.normalize), one wave per sample, d <= 256
__global__ void nce_normalize_kernel(const float *__restrict__ e, int64_t ld, const int64_t *__restrict__ s2v, int64_t ns, int d,
                                     float *__restrict__ en, float *__restrict__ norm) {
    int64_t s = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    if (s >= ns) return;
    int lane = gp_lane();
    const float *row = e + s2v[s] * ld;
    float v[4], ss = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) { int k = lane + 64 * t; v[t] = k < d ? row[k] : 0.f; ss += v[t] * v[t]; }
    ss = gp_wave_sum(ss);
    float nrm = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
    for (int t = 0; t < 4; ++t) { int k = lane + 64 * t; if (k < d) en[s * d + k] = v[t] / nrm; }
    if (lane == 0) norm[s] = nrm;
}
// one wave per anchor: logits over (positive, negatives), cross entropy with target 0, gradient w.r.t. the
// normalised rows (atomics: a sampled point serves many anchors).  nl = 1 + negatives <= 64.
__global__ void nce_anchor_kernel(const float *__restrict__ en, int d, const int64_t *__restrict__ p2b, int64_t na, int nneg,
                                  float inv_t, float *__restrict__ loss, float *__restrict__ den) {
    int64_t a = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    if (a >= na) return;
    int lane = gp_lane();
    const int nl = 1 + nneg;
    const int64_t ia = p2b[a];
    int64_t mine = lane == 0 ? p2b[na + a] : (lane < nl ? p2b[2 * na + a * nneg + (lane - 1)] : 0);
    float av[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { int k = lane + 64 * t; av[t] = k < d ? en[ia * d + k] : 0.f; }
    float logit = -INFINITY;
    for (int j = 0; j < nl; ++j) {
        int64_t ij = __shfl(mine, j, 64);
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) { int k = lane + 64 * t; if (k < d) s += av[t] * en[ij * d + k]; }
        s = gp_wave_sum(s);
        if (lane == j) logit = s * inv_t;
    }
    float mx = gp_wave_max(logit);
    float ex = lane < nl ? expf(logit - mx) : 0.f;
    float den_s = gp_wave_sum(ex);
    float p = ex / den_s;
    float l0 = __shfl(logit, 0, 64);
    if (lane == 0) atomicAdd(loss, (mx + logf(den_s) - l0) / (float)na);
    float g = lane < nl ? (p - (lane == 0 ? 1.f : 0.f)) * inv_t / (float)na : 0.f;       // d loss / d (a . e_j)
    float ga[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < nl; ++j) {
        int64_t ij = __shfl(mine, j, 64);
        float gj = __shfl(g, j, 64);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            int k = lane + 64 * t;
            if (k < d) {
                ga[t] += gj * en[ij * d + k];
                atomicAdd(&den[ij * d + k], gj * av[t]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) { int k = lane + 64 * t; if (k < d) atomicAdd(&den[ia * d + k], ga[t]); }
}
// normalize backward, then add into the voxel row: dE[s2v[s]] += (dEn - En (En . dEn)) / norm
__global__ void nce_scatter_kernel(const float *__restrict__ en, const float *__restrict__ den, const float *__restrict__ norm,
                                   const int64_t *__restrict__ s2v, int64_t ns, int d, float *__restrict__ de, int64_t ld) {
    int64_t s = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    if (s >= ns) return;
    int lane = gp_lane();
    float e[4], g[4], dot = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int k = lane + 64 * t;
        e[t] = k < d ? en[s * d + k] : 0.f;
        g[t] = k < d ? den[s * d + k] : 0.f;
        dot += e[t] * g[t];
    }
    dot = gp_wave_sum(dot);
    float inv = 1.f / norm[s];
    float *row = de + s2v[s] * ld;
#pragma unroll
    for (int t = 0; t < 4; ++t) { int k = lane + 64 * t; if (k < d) atomicAdd(&row[k], (g[t] - e[t] * dot) * inv); }
}

// ------------------------------------------------------------------------------------------------ AdamW
__global__ void adamw_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v, int64_t n,
                             float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float gi = g[i];
    float pi = p[i] * (1.f - lr * wd);
    float mi = b1 * m[i] + (1.f - b1) * gi;
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
    m[i] = mi;
    v[i] = vi;
}

// ------------------------------------------------------------------------------------------------ K nearest points of a query point
constexpr int KP_BINS = 2048, KP_CAP = 2048;
__device__ __forceinline__ int kp_bin(double d2) { return (int)(__float_as_uint((float)d2) >> 20); }   // sign 0, 8 exponent + 3 mantissa bits
__global__ void __launch_bounds__(256)
knn_points_kernel(const float *__restrict__ xyz, int64_t n, const int64_t *__restrict__ queries, int k, int64_t *__restrict__ out,
                  int32_t *__restrict__ flag) {
    __shared__ unsigned hist[KP_BINS];
    __shared__ double cd[KP_CAP];
    __shared__ int ci[KP_CAP];
    __shared__ int s_thr, s_cnt;
    const int tid = threadIdx.x;
    const int64_t q = queries[blockIdx.x];
    const double qx = xyz[q * 3], qy = xyz[q * 3 + 1], qz = xyz[q * 3 + 2];
    for (int i = tid; i < KP_BINS; i += 256) hist[i] = 0;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    for (int64_t i = tid; i < n; i += 256) {
        double dx = xyz[i * 3] - qx, dy = xyz[i * 3 + 1] - qy, dz = xyz[i * 3 + 2] - qz;
        atomicAdd(&hist[kp_bin(dx * dx + dy * dy + dz * dz)], 1u);
    }
    __syncthreads();
    if (tid == 0) {                                       // first bin at which the running count reaches k+1
        unsigned run = 0;
        int t = 0;
        for (; t < KP_BINS; ++t) { run += hist[t]; if (run >= (unsigned)(k + 1)) break; }
        s_thr = t < KP_BINS ? t : KP_BINS - 1;
        if (run > (unsigned)KP_CAP) { *flag = 1; s_thr = -1; }
    }
    __syncthreads();
    const int thr = s_thr;
    if (thr < 0) return;
    for (int64_t i = tid; i < n; i += 256) {
        double dx = xyz[i * 3] - qx, dy = xyz[i * 3 + 1] - qy, dz = xyz[i * 3 + 2] - qz;
        double d2 = dx * dx + dy * dy + dz * dz;
        if (kp_bin(d2) <= thr) { int pos = atomicAdd(&s_cnt, 1); cd[pos] = d2; ci[pos] = (int)i; }
    }
    __syncthreads();
    const int cnt = s_cnt;
    int np2 = 1;
    while (np2 < cnt) np2 <<= 1;
    for (int i = cnt + tid; i < np2; i += 256) { cd[i] = INFINITY; ci[i] = INT32_MAX; }
    __syncthreads();
    for (int kk = 2; kk <= np2; kk <<= 1)                 // bitonic sort by (d2, id)
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += 256) {
                int ixj = i ^ j;
                if (ixj > i) {
                    bool up = (i & kk) == 0;
                    bool gt = cd[i] > cd[ixj] || (cd[i] == cd[ixj] && ci[i] > ci[ixj]);
                    if (gt == up) { double td = cd[i]; cd[i] = cd[ixj]; cd[ixj] = td; int ti = ci[i]; ci[i] = ci[ixj]; ci[ixj] = ti; }
                }
            }
            __syncthreads();
        }
    for (int j = tid; j < k; j += 256) out[(int64_t)blockIdx.x * k + j] = ci[j + 1];     // column 0 (the point itself) dropped
}

}  // namespace

extern "C" size_t gp_col_stats_workspace_bytes(int64_t nv, int32_t c) {
    int64_t nch = (nv + CS_ROWS - 1) / CS_ROWS;
    return gp_align_up((size_t)nch * 2 * c * sizeof(double), 256);
}

// mean[c], var[c] (biased) of the rows of y: two passes (mean, then squared deviations), fp64 accumulation
extern "C" int gp_col_stats(const float *y, int64_t ld, int64_t nv, int32_t c, float *mean, float *var, void *workspace,
                            size_t workspace_bytes, void *stream_) {
    GP_CHECK_ARG(y && mean && var && workspace && nv > 0 && c > 0, "gp_col_stats: null/empty argument");
    if (workspace_bytes < gp_col_stats_workspace_bytes(nv, c)) { gp_set_error("gp_col_stats: workspace too small"); return GP_ENOMEM; }
    hipStream_t s = gp_stream(stream_);
    double *partial = static_cast<double *>(workspace);
    int64_t nch = (nv + CS_ROWS - 1) / CS_ROWS;
    dim3 grid((unsigned)nch, (unsigned)((c + 63) / 64));
    cs_sum_kernel<<<grid, 256, 0, s>>>(y, ld, nv, c, partial);
    cs_final_kernel<<<(c + 63) / 64, 256, 0, s>>>(partial, nch, 1, c, 1.0 / (double)nv, mean);
    cs_var_kernel<<<grid, 256, 0, s>>>(y, ld, nv, c, mean, partial);
    cs_final_kernel<<<(c + 63) / 64, 256, 0, s>>>(partial, nch, 1, c, 1.0 / (double)nv, var);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

// fp64 column sums for SyncBatchNorm (run/train.py:212-213 converts the student to MinkowskiSyncBatchNorm): the caller
// all-reduces them over the ranks.  mean == NULL: out[col] = sum_r y[r][col]; else out[col] = sum_r (y[r][col] - mean[col])^2.
__global__ void __launch_bounds__(256) cs_final_f64_kernel(const double *__restrict__ partial, int64_t nchunks, int nq, int c,
                                                           double *__restrict__ out) {
    __shared__ double red[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    double s = 0.0;
    if (i < nq * c) {
        const int q = i / c, col = i % c;
        for (int64_t k = wv; k < nchunks; k += 4) s += partial[(k * nq + q) * c + col];
    }
    red[wv][lane] = s;
    __syncthreads();
    if (wv == 0 && i < nq * c) out[i] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}
extern "C" int gp_col_sums_f64(const float *y, int64_t ld, int64_t nv, int32_t c, const float *mean, double *out, void *workspace,
                               size_t workspace_bytes, void *stream_) {
    GP_CHECK_ARG(y && out && workspace && nv > 0 && c > 0, "gp_col_sums_f64: null/empty argument");
    if (workspace_bytes < gp_col_stats_workspace_bytes(nv, c)) { gp_set_error("gp_col_sums_f64: workspace too small"); return GP_ENOMEM; }
    hipStream_t s = gp_stream(stream_);
    double *partial = static_cast<double *>(workspace);
    int64_t nch = (nv + CS_ROWS - 1) / CS_ROWS;
    dim3 grid((unsigned)nch, (unsigned)((c + 63) / 64));
    if (mean) cs_var_kernel<<<grid, 256, 0, s>>>(y, ld, nv, c, mean, partial);
    else cs_sum_kernel<<<grid, 256, 0, s>>>(y, ld, nv, c, partial);
    cs_final_f64_kernel<<<(c + 63) / 64, 256, 0, s>>>(partial, nch, 1, c, out);
    GP_CHECK_LAUNCH();
    return GP_OK;
}
// the two reduction vectors of the BatchNorm backward pass as fp64 sums: sums[0:c] = sum dz, sums[c:2c] = sum dz * xhat
extern "C" int gp_bn_bwd_sums_f64(const float *dout, int64_t ld_dout, const float *act, int64_t ld_act, const float *y, int64_t ld_y,
                                  const float *mean, const float *var, float eps, int64_t nv, int32_t c, double *sums,
                                  void *workspace, size_t workspace_bytes, void *stream_) {
    GP_CHECK_ARG(dout && y && mean && var && sums && workspace && nv > 0 && c > 0, "gp_bn_bwd_sums_f64: null/empty argument");
    if (workspace_bytes < gp_col_stats_workspace_bytes(nv, c)) { gp_set_error("gp_bn_bwd_sums_f64: workspace too small"); return GP_ENOMEM; }
    hipStream_t s = gp_stream(stream_);
    double *partial = static_cast<double *>(workspace);
    int64_t nch = (nv + CS_ROWS - 1) / CS_ROWS;
    dim3 grid((unsigned)nch, (unsigned)((c + 63) / 64));
    bn_bwd_reduce_kernel<<<grid, 256, 0, s>>>(dout, ld_dout, act, ld_act, y, ld_y, mean, var, eps, nv, c, partial);
    cs_final_f64_kernel<<<(2 * c + 63) / 64, 256, 0, s>>>(partial, nch, 2, c, sums);
    GP_CHECK_LAUNCH();
    return GP_OK;
}
// dy = gamma/sqrt(var+eps) * (dz - sums[col]/n_total - xhat * sums[c+col]/n_total) with caller-supplied (all-reduced) sums
extern "C" int gp_bn_bwd_apply(const float *dout, int64_t ld_dout, const float *act, int64_t ld_act, const float *y, int64_t ld_y,
                               const float *mean, const float *var, float eps, const float *gamma, const float *sums, int64_t n_total,
                               int64_t nv, int32_t c, float *dy, int64_t ld_dy, float *dz_out, int64_t ld_dz, void *stream_) {
    GP_CHECK_ARG(dout && y && mean && var && gamma && sums && dy && nv > 0 && c > 0 && n_total >= nv, "gp_bn_bwd_apply: bad argument");
    int64_t n = nv * c;
    bn_bwd_apply_kernel<<<(unsigned)((n + 255) / 256), 256, 0, gp_stream(stream_)>>>(dout, ld_dout, act, ld_act, y, ld_y, mean, var, eps, gamma,
                                                                                      sums, n_total, nv, c, dy, ld_dy, dz_out, ld_dz);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

// out = [relu]((y - mean) / sqrt(var + eps) * gamma + beta [+ residual]); optional split copy; optional running-stat update
extern "C" int gp_bn_train_apply(const float *y, int64_t ld, int64_t nv, int32_t c, const float *mean, const float *var,
                                 const float *gamma, const float *beta, float eps, const float *residual, int64_t ld_res,
                                 int32_t relu, float *out, int64_t ld_out, void *out_hi, void *out_lo, int64_t ld_split,
                                 float momentum, float *running_mean, float *running_var, void *stream_) {
    GP_CHECK_ARG(y && mean && var && gamma && beta && out && nv > 0 && c > 0, "gp_bn_train_apply: null/empty argument");
    GP_CHECK_ARG((out_hi == nullptr) == (out_lo == nullptr), "gp_bn_train_apply: split outputs come as a pair");
    hipStream_t s = gp_stream(stream_);
    int64_t n = nv * c;
    bn_apply_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(y, ld, nv, c, mean, var, gamma, beta, eps, residual, ld_res, relu, out, ld_out,
                                                                 static_cast<_Float16 *>(out_hi), static_cast<_Float16 *>(out_lo), ld_split);
    if (running_mean && running_var)
        bn_running_kernel<<<(c + 255) / 256, 256, 0, s>>>(mean, var, c, nv, momentum, running_mean, running_var);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

// dz = dout * (act > 0) (act NULL: dz = dout); dgamma = sum dz*xhat, dbeta = sum dz;
// dy = gamma/sqrt(var+eps) * (dz - dbeta/nv - xhat*dgamma/nv); dz_out (nullable) receives dz (identity branch)
extern "C" int gp_bn_train_backward(const float *dout, int64_t ld_dout, const float *act, int64_t ld_act, const float *y, int64_t ld_y,
                                    const float *mean, const float *var, float eps, const float *gamma, int64_t nv, int32_t c,
                                    float *dy, int64_t ld_dy, float *dz_out, int64_t ld_dz, float *dgamma, float *dbeta,
                                    void *workspace, size_t workspace_bytes, void *stream_) {
    GP_CHECK_ARG(dout && y && mean && var && gamma && dy && dgamma && dbeta && workspace && nv > 0 && c > 0,
                 "gp_bn_train_backward: null/empty argument");
    size_t need = gp_col_stats_workspace_bytes(nv, c) + gp_align_up((size_t)2 * c * sizeof(float), 256);
    if (workspace_bytes < need) { gp_set_error("gp_bn_train_backward: workspace too small"); return GP_ENOMEM; }
    hipStream_t s = gp_stream(stream_);
    double *partial = static_cast<double *>(workspace);
    float *sums = reinterpret_cast<float *>(static_cast<char *>(workspace) + gp_col_stats_workspace_bytes(nv, c));
    int64_t nch = (nv + CS_ROWS - 1) / CS_ROWS;
    dim3 grid((unsigned)nch, (unsigned)((c + 63) / 64));
    bn_bwd_reduce_kernel<<<grid, 256, 0, s>>>(dout, ld_dout, act, ld_act, y, ld_y, mean, var, eps, nv, c, partial);
    cs_final_kernel<<<(2 * c + 63) / 64, 256, 0, s>>>(partial, nch, 2, c, 1.0, sums);
    int64_t n = nv * c;
    bn_bwd_apply_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(dout, ld_dout, act, ld_act, y, ld_y, mean, var, eps, gamma, sums, nv, nv, c,
                                                                     dy, ld_dy, dz_out, ld_dz);
    GP_CHECK_HIP(hipMemcpyAsync(dbeta, sums, (size_t)c * sizeof(float), hipMemcpyDeviceToDevice, s));
    GP_CHECK_HIP(hipMemcpyAsync(dgamma, sums + c, (size_t)c * sizeof(float), hipMemcpyDeviceToDevice, s));
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" size_t gp_infonce_workspace_bytes(int64_t num_samples, int32_t d) {
    return gp_align_up((size_t)num_samples * d * sizeof(float), 256) * 2 + gp_align_up((size_t)num_samples * sizeof(float), 256);
}

// loss (device scalar, overwritten) and dE [nv, d] (overwritten) of the InfoNCE of affinity_module.py:1219-1233
extern "C" int gp_infonce_fwd_bwd(const float *e, int64_t ld_e, int64_t nv, int32_t d, const int64_t *sample_to_voxel, int64_t num_samples,
                                  const int64_t *point_to_batch, int64_t num_anchors, int32_t num_negatives, float temperature,
                                  float *loss, float *de, int64_t ld_de, void *workspace, size_t workspace_bytes, void *stream_) {
    GP_CHECK_ARG(e && sample_to_voxel && point_to_batch && loss && de && workspace, "gp_infonce_fwd_bwd: null argument");
    GP_CHECK_ARG(nv > 0 && num_samples > 0 && num_anchors > 0 && d > 0 && d <= 256, "gp_infonce_fwd_bwd: bad shape (d <= 256)");
    GP_CHECK_ARG(num_negatives >= 0 && num_negatives < 64, "gp_infonce_fwd_bwd: 1 + negatives must fit one wave (<= 64)");
    GP_CHECK_ARG(temperature > 0.f, "gp_infonce_fwd_bwd: temperature must be positive");
    if (workspace_bytes < gp_infonce_workspace_bytes(num_samples, d)) { gp_set_error("gp_infonce_fwd_bwd: workspace too small"); return GP_ENOMEM; }
    hipStream_t s = gp_stream(stream_);
    GpCarver cv(workspace, workspace_bytes);
    float *en = cv.take<float>(num_samples * d);
    float *den = cv.take<float>(num_samples * d);
    float *norm = cv.take<float>(num_samples);
    GP_CHECK_HIP(hipMemsetAsync(den, 0, (size_t)num_samples * d * sizeof(float), s));
    GP_CHECK_HIP(hipMemsetAsync(loss, 0, sizeof(float), s));
    GP_CHECK_HIP(hipMemset2DAsync(de, (size_t)ld_de * sizeof(float), 0, (size_t)d * sizeof(float), (size_t)nv, s));
    nce_normalize_kernel<<<(unsigned)((num_samples * 64 + 255) / 256), 256, 0, s>>>(e, ld_e, sample_to_voxel, num_samples, d, en, norm);
    nce_anchor_kernel<<<(unsigned)((num_anchors * 64 + 255) / 256), 256, 0, s>>>(en, d, point_to_batch, num_anchors, num_negatives,
                                                                                 1.0f / temperature, loss, den);
    nce_scatter_kernel<<<(unsigned)((num_samples * 64 + 255) / 256), 256, 0, s>>>(en, den, norm, sample_to_voxel, num_samples, d, de, ld_de);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

// torch.optim.AdamW step on one flat fp32 tensor; step >= 1
extern "C" int gp_adamw_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int64_t step, void *stream_) {
    GP_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "gp_adamw_step: null/empty argument");
    double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    adamw_kernel<<<(unsigned)((n + 255) / 256), 256, 0, gp_stream(stream_)>>>(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps,
                                                                             weight_decay, (float)bc1, (float)sqrt(bc2));
    GP_CHECK_LAUNCH();
    return GP_OK;
}

// out i64 [num_queries, k]: the k nearest OTHER rows of xyz for each query row (faiss IndexFlatL2.search(k+1)[:, 1:]),
// ordered by (squared distance in fp64 of the fp32 coordinates, row id).  *flag_dev != 0: a query had more than 2048
// candidates inside its (k+1)-th distance bin (massively duplicated points): result invalid.
extern "C" int gp_knn_points_f32(const float *xyz, int64_t n, const int64_t *queries, int64_t num_queries, int32_t k, int64_t *out,
                                 int32_t *flag_dev, void *stream_) {
    GP_CHECK_ARG(xyz && queries && out && flag_dev && n > 0 && num_queries > 0, "gp_knn_points_f32: null/empty argument");
    GP_CHECK_ARG(k >= 1 && k + 1 <= n && k + 1 <= KP_CAP / 2, "gp_knn_points_f32: k=%d out of range", k);
    GP_CHECK_ARG(n < INT32_MAX, "gp_knn_points_f32: too many points");
    hipStream_t s = gp_stream(stream_);
    GP_CHECK_HIP(hipMemsetAsync(flag_dev, 0, sizeof(int32_t), s));
    knn_points_kernel<<<(unsigned)num_queries, 256, 0, s>>>(xyz, n, queries, k, out, flag_dev);
    GP_CHECK_LAUNCH();
    return GP_OK;
}
