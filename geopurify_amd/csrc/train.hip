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
// partial[chunk][q][c] = sum over the chunk's rows of f_q(row, c), q < NQ; fp64 accumulation, fixed order (a column's rows wv, wv + 4, ...
// of the chunk in its wave wv, then the four waves in order).  W = 4: a lane owns four adjacent columns and reads them with one 16-byte
// load per operand (a workgroup = 256 rows x 256 columns); W = 1: one column per lane (any c, any alignment).  Same sums either way.
template <int W> __device__ __forceinline__ void cs_ld(const float *p, float *o) {
    if (W == 4) *reinterpret_cast<float4 *>(o) = *reinterpret_cast<const float4 *>(p);
    else o[0] = p[0];
}
template <int NQ, int W, typename F>
__device__ __forceinline__ void col_partial(int64_t nv, int c, double *__restrict__ partial, F f) {
    __shared__ double red[4][NQ][64 * W];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = (blockIdx.y * 64 + lane) * W;
    const int64_t r0 = (int64_t)blockIdx.x * CS_ROWS;
    double acc[W][NQ];
#pragma unroll
    for (int k = 0; k < W; ++k)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[k][q] = 0.0;
    if (col < c)
        for (int64_t r = r0 + wv; r < r0 + CS_ROWS && r < nv; r += 4) {
            double v[W][NQ];
            f(r, col, v);
#pragma unroll
            for (int k = 0; k < W; ++k)
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[k][q] += v[k][q];
        }
#pragma unroll
    for (int k = 0; k < W; ++k)
#pragma unroll
        for (int q = 0; q < NQ; ++q) red[wv][q][lane * W + k] = acc[k][q];
    __syncthreads();
    if (wv == 0 && col < c)
#pragma unroll
        for (int k = 0; k < W; ++k)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int l = lane * W + k;
                partial[((int64_t)blockIdx.x * NQ + q) * c + col + k] = red[0][q][l] + red[1][q][l] + red[2][q][l] + red[3][q][l];
            }
}
// grid of a column reduction and whether the 16-byte form applies
static inline bool cs_vec(int c, std::initializer_list<int64_t> lds, std::initializer_list<const void *> ptrs) {
    bool ok = c % 4 == 0;
    for (int64_t l : lds) ok = ok && l % 4 == 0;
    for (const void *p : ptrs) ok = ok && (reinterpret_cast<uintptr_t>(p) & 15) == 0;
    return ok;
}
static inline dim3 cs_grid(int64_t nv, int c, bool vec) {
    return dim3((unsigned)((nv + CS_ROWS - 1) / CS_ROWS), (unsigned)((c + (vec ? 255 : 63)) / (vec ? 256 : 64)));
}

template <int W>
__global__ void __launch_bounds__(256) cs_sum_kernel(const float *__restrict__ y, int64_t ld, int64_t nv, int c, double *__restrict__ partial) {
    col_partial<1, W>(nv, c, partial, [&](int64_t r, int col, double (*v)[1]) {
        float x[W];
        cs_ld<W>(y + r * ld + col, x);
#pragma unroll
        for (int k = 0; k < W; ++k) v[k][0] = (double)x[k];
    });
}
// sums of x and x^2 in one sweep (gp_col_stats)
template <int W>
__global__ void __launch_bounds__(256) cs_sum2_kernel(const float *__restrict__ y, int64_t ld, int64_t nv, int c, double *__restrict__ partial) {
    col_partial<2, W>(nv, c, partial, [&](int64_t r, int col, double (*v)[2]) {
        float x[W];
        cs_ld<W>(y + r * ld + col, x);
#pragma unroll
        for (int k = 0; k < W; ++k) { const double d = (double)x[k]; v[k][0] = d; v[k][1] = d * d; }
    });
}
template <int W>
__global__ void __launch_bounds__(256) cs_var_kernel(const float *__restrict__ y, int64_t ld, int64_t nv, int c,
                                                     const float *__restrict__ mean, double *__restrict__ partial) {
    col_partial<1, W>(nv, c, partial, [&](int64_t r, int col, double (*v)[1]) {
        float x[W];
        cs_ld<W>(y + r * ld + col, x);
#pragma unroll
        for (int k = 0; k < W; ++k) { const double d = (double)x[k] - (double)mean[col + k]; v[k][0] = d * d; }
    });
}
// out[q][c] = scale * sum over chunks: one workgroup per 64 columns, the chunks strided over CF_WAVES waves (fixed order).  16 waves:
// with 4 the 8-16 workgroups of a 512-column layer each walked 110 dependent loads (40 us per call, 27 calls per training step).
constexpr int CF_WAVES = 16;
template <typename OUT>
__device__ __forceinline__ void cs_final_body(const double *__restrict__ partial, int64_t nchunks, int nq, int c, double scale, OUT *__restrict__ out) {
    __shared__ double red[CF_WAVES][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;                 // flat (q, col)
    double s = 0.0;
    if (i < nq * c) {
        const int q = i / c, col = i % c;
        for (int64_t k = wv; k < nchunks; k += CF_WAVES) s += partial[(k * nq + q) * c + col];
    }
    red[wv][lane] = s;
    __syncthreads();
    if (wv == 0 && i < nq * c) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < CF_WAVES; ++w) t += red[w][lane];
        out[i] = (OUT)(t * scale);
    }
}
// mean[c], var[c] from the partial sums of x (q = 0) and x^2 (q = 1)
__global__ void __launch_bounds__(CF_WAVES * 64) cs_meanvar_final_kernel(const double *__restrict__ partial, int64_t nchunks, int c, double inv_n,
                                                                         float *__restrict__ mean, float *__restrict__ var) {
    __shared__ double red[2][CF_WAVES][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    double s0 = 0.0, s1 = 0.0;
    if (col < c)
        for (int64_t k = wv; k < nchunks; k += CF_WAVES) { s0 += partial[(k * 2) * c + col]; s1 += partial[(k * 2 + 1) * c + col]; }
    red[0][wv][lane] = s0;
    red[1][wv][lane] = s1;
    __syncthreads();
    if (wv == 0 && col < c) {
        double t0 = 0.0, t1 = 0.0;
#pragma unroll
        for (int w = 0; w < CF_WAVES; ++w) { t0 += red[0][w][lane]; t1 += red[1][w][lane]; }
        const double m = t0 * inv_n, v = t1 * inv_n - m * m;
        mean[col] = (float)m;
        var[col] = (float)(v > 0.0 ? v : 0.0);
    }
}
__global__ void __launch_bounds__(CF_WAVES * 64) cs_final_kernel(const double *__restrict__ partial, int64_t nchunks, int nq, int c, double scale,
                                                                 float *__restrict__ out) {
    cs_final_body(partial, nchunks, nq, c, scale, out);
}

// ------------------------------------------------------------------------------------------------ BN forward / backward
// Row sweeps of the BatchNorm kernels: the launch has a multiple of c / W threads, so a thread keeps ITS W columns for every row it visits --
// the per-column constants (mean, 1/std, gamma, beta, the backward sums) are loaded once, the loop is loads, a few multiply-adds and stores.
// (Per-element index arithmetic and per-element loads of the column vectors made bn_bwd_apply 390 us for 1.2 GB; this form: see DESIGN 10.)
__host__ __device__ inline unsigned bn_sweep_blocks(int64_t nv, int cw) {
    int g = 256, x = cw;
    while (x) { const int t = g % x; g = x; x = t; }      // gcd(256, cw)
    const int64_t m = cw / g;                             // blocks must be a multiple of m for 256 * blocks % cw == 0
    int64_t want = (nv * cw + 255) / 256;
    want = want < 4096 ? want : 4096;
    int64_t blocks = want / m * m;
    return (unsigned)(blocks < m ? m : blocks);
}
template <int W>
__global__ void __launch_bounds__(256)
bn_apply_kernel(const float *__restrict__ y, int64_t ld, int64_t nv, int c, const float *__restrict__ mean,
                const float *__restrict__ var, const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                const float *__restrict__ residual, int64_t ld_res, int relu, float *__restrict__ out, int64_t ld_out,
                _Float16 *__restrict__ out_hi, _Float16 *__restrict__ out_lo, int64_t ld_sp) {
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    const int cw = c / W;
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, nthreads = (int64_t)gridDim.x * blockDim.x;
    const int col = (int)(t % cw) * W;
    const int64_t rstep = nthreads / cw;
    float mu[W], is[W], ga[W], be[W];
#pragma unroll
    for (int k = 0; k < W; ++k) { mu[k] = mean[col + k]; is[k] = 1.0f / sqrtf(var[col + k] + eps); ga[k] = gamma[col + k]; be[k] = beta[col + k]; }
    for (int64_t r = t / cw; r < nv; r += rstep) {
        float yv[W], rv[W], v[W];
        if (W == 4) {
            *reinterpret_cast<float4 *>(yv) = *reinterpret_cast<const float4 *>(y + r * ld + col);
            if (residual) *reinterpret_cast<float4 *>(rv) = *reinterpret_cast<const float4 *>(residual + r * ld_res + col);
        } else {
            yv[0] = y[r * ld + col];
            if (residual) rv[0] = residual[r * ld_res + col];
        }
#pragma unroll
        for (int k = 0; k < W; ++k) {
            v[k] = (yv[k] - mu[k]) * is[k] * ga[k] + be[k];
            if (residual) v[k] += rv[k];
            if (relu) v[k] = v[k] > 0.f ? v[k] : 0.f;
        }
        if (out) {                                          // (nullptr: only the split planes -- a layer whose backward mask comes from y)
            if (W == 4) *reinterpret_cast<float4 *>(out + r * ld_out + col) = *reinterpret_cast<const float4 *>(v);
            else out[r * ld_out + col] = v[0];
        }
        if (out_hi) {
            if (W == 4) {
                f16x4 h, l;
#pragma unroll
                for (int k = 0; k < 4; ++k) { h[k] = (_Float16)v[k]; l[k] = (_Float16)(v[k] - (float)h[k]); }
                *reinterpret_cast<f16x4 *>(out_hi + r * ld_sp + col) = h;
                *reinterpret_cast<f16x4 *>(out_lo + r * ld_sp + col) = l;
            } else {
                const _Float16 h = (_Float16)v[0];
                out_hi[r * ld_sp + col] = h;
                out_lo[r * ld_sp + col] = (_Float16)(v[0] - (float)h);
            }
        }
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
// dz = dout * mask; sums of dz and dz * xhat.  The mask of the ReLU behind the layer: act > 0 (act = the layer's fp32 output), or -- act ==
// nullptr and beta_m given, a layer WITHOUT a residual -- recomputed from y: (y - mean) * invstd * gamma + beta > 0, the expression
// bn_apply_kernel evaluated (same operations in the same order: the same float), which saves reading the 232-MB activation here and in
// bn_bwd_apply_kernel and lets the forward pass skip writing it; both nullptr: no mask.
template <int W>
__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(const float *__restrict__ dout, int64_t ld_d, const float *__restrict__ act, int64_t ld_a,
                                                            const float *__restrict__ y, int64_t ld_y, const float *__restrict__ mean,
                                                            const float *__restrict__ var, float eps, const float *__restrict__ gamma_m,
                                                            const float *__restrict__ beta_m, int64_t nv, int c, double *__restrict__ partial,
                                                            float *__restrict__ pmax /*nullable: [chunk][2][c] max |dz|, max |xhat|*/) {
    const int col0 = (blockIdx.y * 64 + (threadIdx.x & 63)) * W;
    float mu[W], is[W], ga[W], be[W], mdz[W], mxh[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const bool in = col0 + k < c;
        mu[k] = in ? mean[col0 + k] : 0.f; is[k] = in ? 1.0f / sqrtf(var[col0 + k] + eps) : 0.f;
        ga[k] = (in && beta_m) ? gamma_m[col0 + k] : 0.f; be[k] = (in && beta_m) ? beta_m[col0 + k] : 0.f;
        mdz[k] = mxh[k] = 0.f;
    }
    col_partial<2, W>(nv, c, partial, [&](int64_t r, int col, double (*v)[2]) {
        float dz[W], yv[W], av[W];
        cs_ld<W>(dout + r * ld_d + col, dz);
        cs_ld<W>(y + r * ld_y + col, yv);
        if (act) cs_ld<W>(act + r * ld_a + col, av);
#pragma unroll
        for (int k = 0; k < W; ++k) {
            if (act) { if (!(av[k] > 0.f)) dz[k] = 0.f; }
            else if (beta_m) { if (!((yv[k] - mu[k]) * is[k] * ga[k] + be[k] > 0.f)) dz[k] = 0.f; }
            const float xhat = (yv[k] - mu[k]) * is[k];
            v[k][0] = (double)dz[k];
            v[k][1] = (double)dz[k] * (double)xhat;
            mdz[k] = fmaxf(mdz[k], fabsf(dz[k]));
            mxh[k] = fmaxf(mxh[k], fabsf(xhat));
        }
    });
    if (pmax) {                                           // (uniform) the chunk's column maxima: what bounds |dy| before dy exists
        __shared__ float rmax[4][2][64 * W];
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < W; ++k) { rmax[wv][0][lane * W + k] = mdz[k]; rmax[wv][1][lane * W + k] = mxh[k]; }
        __syncthreads();
        if (wv == 0 && col0 < c)
#pragma unroll
            for (int k = 0; k < W; ++k)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int l = lane * W + k;
                    pmax[((int64_t)blockIdx.x * 2 + q) * c + col0 + k] = fmaxf(fmaxf(rmax[0][q][l], rmax[1][q][l]), fmaxf(rmax[2][q][l], rmax[3][q][l]));
                }
    }
}
// sums[0:c] = sum dz, sums[c:2c] = sum dz * xhat (fp64, the chunks in cs_final_kernel's order) and, from the chunks' column maxima, a bound
// of max |dy|: |dy| = |gamma| invstd |dz - s1 / n - xhat s2 / n| <= |gamma| invstd (max |dz| + |s1| / n + max |xhat| |s2| / n) per column; the
// largest goes to amax_bits (the uint image of a non-negative float, atomicMax) -- the scale of the gradient's f16 split, known BEFORE the sweep
// that computes dy, which can therefore write the split planes itself (no fp32 dy, no separate split pass).  The bound is within a small factor
// of the true maximum (it is attained when the largest |dz| meets opposite-signed means): the split loses a fraction of a bit of head room.
__global__ void __launch_bounds__(CF_WAVES * 64)
bn_bwd_final_bound_kernel(const double *__restrict__ partial, const float *__restrict__ pmax, int64_t nchunks, int c, const float *__restrict__ var,
                          float eps, const float *__restrict__ gamma, int64_t n_total, float *__restrict__ sums, unsigned *__restrict__ amax_bits) {
    __shared__ double red[2][CF_WAVES][64];
    __shared__ float rmx[2][CF_WAVES][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    double s0 = 0.0, s1 = 0.0;
    float m0 = 0.f, m1 = 0.f;
    if (col < c)
        for (int64_t k = wv; k < nchunks; k += CF_WAVES) {
            s0 += partial[(k * 2) * c + col]; s1 += partial[(k * 2 + 1) * c + col];
            m0 = fmaxf(m0, pmax[(k * 2) * c + col]); m1 = fmaxf(m1, pmax[(k * 2 + 1) * c + col]);
        }
    red[0][wv][lane] = s0; red[1][wv][lane] = s1; rmx[0][wv][lane] = m0; rmx[1][wv][lane] = m1;
    __syncthreads();
    if (wv == 0) {
        float b = 0.f;
        if (col < c) {
            double t0 = 0.0, t1 = 0.0;
            float x0 = 0.f, x1 = 0.f;
#pragma unroll
            for (int w = 0; w < CF_WAVES; ++w) { t0 += red[0][w][lane]; t1 += red[1][w][lane]; x0 = fmaxf(x0, rmx[0][w][lane]); x1 = fmaxf(x1, rmx[1][w][lane]); }
            const float f0 = (float)t0, f1 = (float)t1, inv_n = 1.0f / (float)n_total;
            sums[col] = f0;
            sums[c + col] = f1;
            b = fabsf(gamma[col]) * (1.0f / sqrtf(var[col] + eps)) * (x0 + fabsf(f0) * inv_n + x1 * fabsf(f1) * inv_n);
        }
        b = gp_wave_max(b);
        if (lane == 0) atomicMax(amax_bits, __float_as_uint(b));
    }
}
// W = 4: c and every leading dimension are multiples of 4 (16-byte accesses); amax_bits (nullable): atomicMax of |dy| as the uint image of a
// non-negative float, one atomic per workgroup -- the power-of-two scale of the gradient's f16 split without another sweep over dy.
template <int W>
__global__ void __launch_bounds__(256)
bn_bwd_apply_kernel(const float *__restrict__ dout, int64_t ld_d, const float *__restrict__ act, int64_t ld_a,
                    const float *__restrict__ y, int64_t ld_y, const float *__restrict__ mean, const float *__restrict__ var,
                    float eps, const float *__restrict__ gamma, const float *__restrict__ beta_m, const float *__restrict__ sums, int64_t n_total,
                    int64_t nv, int c, float *__restrict__ dy, int64_t ld_dy, float *__restrict__ dz_out, int64_t ld_dz,
                    unsigned *__restrict__ amax_bits) {
    const int cw = c / W;
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, nthreads = (int64_t)gridDim.x * blockDim.x;
    const int col = (int)(t % cw) * W;
    const int64_t rstep = nthreads / cw;
    const float inv_n = 1.0f / (float)n_total;
    float mu[W], is[W], ga[W], be[W], s1[W], s2[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
        mu[k] = mean[col + k]; is[k] = 1.0f / sqrtf(var[col + k] + eps); ga[k] = gamma[col + k]; be[k] = beta_m ? beta_m[col + k] : 0.f;
        s1[k] = sums[col + k] * inv_n; s2[k] = sums[c + col + k] * inv_n;
    }
    float m = 0.f;
    for (int64_t r = t / cw; r < nv; r += rstep) {
        float dz[W], yv[W], av[W], o[W];
        if (W == 4) {
            *reinterpret_cast<float4 *>(dz) = *reinterpret_cast<const float4 *>(dout + r * ld_d + col);
            *reinterpret_cast<float4 *>(yv) = *reinterpret_cast<const float4 *>(y + r * ld_y + col);
            if (act) *reinterpret_cast<float4 *>(av) = *reinterpret_cast<const float4 *>(act + r * ld_a + col);
        } else {
            dz[0] = dout[r * ld_d + col];
            yv[0] = y[r * ld_y + col];
            if (act) av[0] = act[r * ld_a + col];
        }
#pragma unroll
        for (int k = 0; k < W; ++k) {
            if (act) { if (!(av[k] > 0.f)) dz[k] = 0.f; }
            else if (beta_m) { if (!((yv[k] - mu[k]) * is[k] * ga[k] + be[k] > 0.f)) dz[k] = 0.f; }
            const float xhat = (yv[k] - mu[k]) * is[k];
            o[k] = ga[k] * is[k] * (dz[k] - s1[k] - xhat * s2[k]);
            m = fmaxf(m, fabsf(o[k]));
        }
        if (W == 4) {
            *reinterpret_cast<float4 *>(dy + r * ld_dy + col) = *reinterpret_cast<const float4 *>(o);
            if (dz_out) *reinterpret_cast<float4 *>(dz_out + r * ld_dz + col) = *reinterpret_cast<const float4 *>(dz);
        } else {
            dy[r * ld_dy + col] = o[0];
            if (dz_out) dz_out[r * ld_dz + col] = dz[0];
        }
    }
    if (amax_bits) {
        __shared__ float s_m[4];
        m = gp_wave_max(m);
        if (gp_lane() == 0) s_m[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
            atomicMax(amax_bits, __float_as_uint(m));
        }
    }
}
// the same sweep writing dy * s as f16 hi / lo planes (s = scale2[0], from bn_bwd_final_bound_kernel) instead of fp32 rows; row nv of the planes
// (the weight gradient's padded pairs point there) is zeroed by the first threads.  c % 4 == 0, 16-byte aligned rows.
__global__ void __launch_bounds__(256)
bn_bwd_apply_split_kernel(const float *__restrict__ dout, int64_t ld_d, const float *__restrict__ act, int64_t ld_a,
                          const float *__restrict__ y, int64_t ld_y, const float *__restrict__ mean, const float *__restrict__ var,
                          float eps, const float *__restrict__ gamma, const float *__restrict__ beta_m, const float *__restrict__ sums, int64_t n_total,
                          int64_t nv, int c, const float *__restrict__ scale2, _Float16 *__restrict__ dy_hi, _Float16 *__restrict__ dy_lo, int64_t ld_h,
                          float *__restrict__ dz_out, int64_t ld_dz) {
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    constexpr int W = 4;
    const int cw = c / W;
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, nthreads = (int64_t)gridDim.x * blockDim.x;
    const int col = (int)(t % cw) * W;
    const int64_t rstep = nthreads / cw;
    const float inv_n = 1.0f / (float)n_total, sc = scale2[0];
    float mu[W], is[W], ga[W], be[W], s1[W], s2[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
        mu[k] = mean[col + k]; is[k] = 1.0f / sqrtf(var[col + k] + eps); ga[k] = gamma[col + k]; be[k] = beta_m ? beta_m[col + k] : 0.f;
        s1[k] = sums[col + k] * inv_n; s2[k] = sums[c + col + k] * inv_n;
    }
    if (t < cw) {
        f16x4 z = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        *reinterpret_cast<f16x4 *>(dy_hi + nv * ld_h + col) = z;
        *reinterpret_cast<f16x4 *>(dy_lo + nv * ld_h + col) = z;
    }
    for (int64_t r = t / cw; r < nv; r += rstep) {
        float dz[W], yv[W], av[W];
        *reinterpret_cast<float4 *>(dz) = *reinterpret_cast<const float4 *>(dout + r * ld_d + col);
        *reinterpret_cast<float4 *>(yv) = *reinterpret_cast<const float4 *>(y + r * ld_y + col);
        if (act) *reinterpret_cast<float4 *>(av) = *reinterpret_cast<const float4 *>(act + r * ld_a + col);
        f16x4 h, l;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            if (act) { if (!(av[k] > 0.f)) dz[k] = 0.f; }
            else if (beta_m) { if (!((yv[k] - mu[k]) * is[k] * ga[k] + be[k] > 0.f)) dz[k] = 0.f; }
            const float xhat = (yv[k] - mu[k]) * is[k];
            const float o = ga[k] * is[k] * (dz[k] - s1[k] - xhat * s2[k]) * sc;
            h[k] = (_Float16)o;
            l[k] = (_Float16)(o - (float)h[k]);
        }
        *reinterpret_cast<f16x4 *>(dy_hi + r * ld_h + col) = h;
        *reinterpret_cast<f16x4 *>(dy_lo + r * ld_h + col) = l;
        if (dz_out) *reinterpret_cast<float4 *>(dz_out + r * ld_dz + col) = *reinterpret_cast<const float4 *>(dz);
    }
}
// scale2[0] holds the amax bits on entry; [s, 1/s] on exit
__global__ void bn_scale2_kernel(float *__restrict__ scale2) {
    const float s = gp_pow2_for(__uint_as_float(reinterpret_cast<const unsigned *>(scale2)[0]));
    scale2[0] = s;
    scale2[1] = 1.f / s;
}
static int bn_bwd_apply_launch(const float *dout, int64_t ld_dout, const float *act, int64_t ld_act, const float *y, int64_t ld_y, const float *mean,
                               const float *var, float eps, const float *gamma, const float *beta_m, const float *sums, int64_t n_total, int64_t nv, int c, float *dy,
                               int64_t ld_dy, float *dz_out, int64_t ld_dz, float *dy_scale2, hipStream_t s) {
    if (dy_scale2) GP_CHECK_HIP(hipMemsetAsync(dy_scale2, 0, 8, s));
    unsigned *bits = reinterpret_cast<unsigned *>(dy_scale2);
    auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool vec = c % 4 == 0 && ld_dout % 4 == 0 && ld_y % 4 == 0 && ld_dy % 4 == 0 && (!act || ld_act % 4 == 0) && (!dz_out || ld_dz % 4 == 0) &&
                     al16(dout) && al16(y) && al16(dy) && al16(act) && al16(dz_out);
    const unsigned blocks = bn_sweep_blocks(nv, vec ? c / 4 : c);
    if (vec) bn_bwd_apply_kernel<4><<<blocks, 256, 0, s>>>(dout, ld_dout, act, ld_act, y, ld_y, mean, var, eps, gamma, beta_m, sums, n_total, nv, c, dy, ld_dy, dz_out, ld_dz, bits);
    else bn_bwd_apply_kernel<1><<<blocks, 256, 0, s>>>(dout, ld_dout, act, ld_act, y, ld_y, mean, var, eps, gamma, beta_m, sums, n_total, nv, c, dy, ld_dy, dz_out, ld_dz, bits);
    if (dy_scale2) bn_scale2_kernel<<<1, 1, 0, s>>>(dy_scale2);
    return GP_OK;
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
// One workgroup per query.  Sweep 1: every thread keeps the lowest d^2 of the points it visits; the (k+1)-th lowest of the NT thread minima
// bounds the (k+1)-th lowest distance from above (k+1 distinct points lie at or below it) and, points being in no particular order along a
// thread's stride, only ~1.3 (k+1) points do.  Sweep 2 (the coordinates again, from L2) collects them; a bitonic sort by (d^2, id) orders them.
// (Rounds 3-5 found the bound with a 2048-bin LDS histogram of all n distances: 150k same-address-heavy LDS atomics per query, 1.34 ms for
// 4096 queries; it remains as the fallback when more than KP_CAP points lie under the thread-minimum bound -- a point order with the stride's
// period -- and flags the query only if its own bound does not fit either: massively duplicated points.)
template <int NT>
__global__ void __launch_bounds__(NT)
knn_points_kernel(const float *__restrict__ xyz, int64_t n, const int64_t *__restrict__ queries, int k, int64_t *__restrict__ out,
                  int32_t *__restrict__ flag, int marked_only) {
    __shared__ unsigned hist[KP_BINS];
    __shared__ double cd[KP_CAP];
    __shared__ int ci[KP_CAP];
    __shared__ double s_tm[NT];
    __shared__ double s_bound;
    __shared__ int s_thr, s_cnt;
    const int tid = threadIdx.x;
    if (marked_only && out[(int64_t)blockIdx.x * k] != -1) return;       // (uniform) only the queries knn_points_multi_kernel handed back
    const int64_t q = queries[blockIdx.x];
    const double qx = xyz[q * 3], qy = xyz[q * 3 + 1], qz = xyz[q * 3 + 2];
    auto dist2 = [&](int64_t i) { const double dx = xyz[i * 3] - qx, dy = xyz[i * 3 + 1] - qy, dz = xyz[i * 3 + 2] - qz; return dx * dx + dy * dy + dz * dz; };
    double tmin = INFINITY;
    for (int64_t i = tid; i < n; i += NT) { const double d2 = dist2(i); tmin = d2 < tmin ? d2 : tmin; }
    s_tm[tid] = tmin;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    {
        int rank = 0;
        for (int u = 0; u < NT; ++u) { const double m = s_tm[u]; rank += (m < tmin) || (m == tmin && u < tid); }
        if (rank == k) s_bound = tmin;
    }
    __syncthreads();
    const double bound = s_bound;
    for (int64_t i = tid; i < n; i += NT) {
        const double d2 = dist2(i);
        if (d2 <= bound) { const int pos = atomicAdd(&s_cnt, 1); if (pos < KP_CAP) { cd[pos] = d2; ci[pos] = (int)i; } }
    }
    __syncthreads();
    if (s_cnt > KP_CAP) {                                  // (uniform) the histogram bound of rounds 3-5
        __syncthreads();
        for (int i = tid; i < KP_BINS; i += NT) hist[i] = 0;
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        for (int64_t i = tid; i < n; i += NT) atomicAdd(&hist[kp_bin(dist2(i))], 1u);
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
        for (int64_t i = tid; i < n; i += NT) {
            const double d2 = dist2(i);
            if (kp_bin(d2) <= thr) { int pos = atomicAdd(&s_cnt, 1); cd[pos] = d2; ci[pos] = (int)i; }
        }
        __syncthreads();
    }
    const int cnt = s_cnt;
    int np2 = 1;
    while (np2 < cnt) np2 <<= 1;
    for (int i = cnt + tid; i < np2; i += NT) { cd[i] = INFINITY; ci[i] = INT32_MAX; }
    __syncthreads();
    for (int kk = 2; kk <= np2; kk <<= 1)                 // bitonic sort by (d2, id)
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += NT) {
                int ixj = i ^ j;
                if (ixj > i) {
                    bool up = (i & kk) == 0;
                    bool gt = cd[i] > cd[ixj] || (cd[i] == cd[ixj] && ci[i] > ci[ixj]);
                    if (gt == up) { double td = cd[i]; cd[i] = cd[ixj]; cd[ixj] = td; int ti = ci[i]; ci[i] = ci[ixj]; ci[ixj] = ti; }
                }
            }
            __syncthreads();
        }
    for (int j = tid; j < k; j += NT) out[(int64_t)blockIdx.x * k + j] = ci[j + 1];     // column 0 (the point itself) dropped
}

// KQ queries per workgroup: the coordinates of a point are loaded once for KQ distances (one query per workgroup reads the 1.8 MB of
// coordinates 2 x 4096 times from L2: 14.7 GB, which is what bounded it at 1.18 ms), the same thread-minimum bounds, KQ_CAP candidates per query.
// The two sweeps work in FP32 and only the candidates' distances are taken in fp64 (the order the result is defined in): with s32 the fp32
// evaluation of a squared distance, |s32 - d^2| <= eps d^2, eps = 2^-21 (three subtractions, three squares, two additions of non-negative
// terms: 5.1 x 2^-24).  B32 = the (k+1)-th lowest thread minimum of s32: k + 1 points have s32 <= B32, hence d^2 <= B32 (1 + 2 eps) =: B64 --
// an upper bound of the (k+1)-th lowest true distance; a point with d^2 <= B64 has s32 <= B32 (1 + 4 eps).  Sweep 2 therefore tests
// s32 <= B32 (1 + 2^-18) in fp32 and, for the few that pass, d^2 <= B32 (1 + 2^-19) in fp64 -- every true neighbour is collected, with its fp64
// distance.  (A bound below 1e-30 -- k + 1 points within 1e-15 of the query, where fp32 squares underflow -- hands the query back.)
// A query with more candidates than KQ_CAP under its bound gets -1 in its first output column and is redone by knn_points_kernel (marked_only).
constexpr int KQ = 4, KQ_CAP = 512;
__global__ void __launch_bounds__(256)
knn_points_multi_kernel(const float *__restrict__ xyz, int64_t n, const int64_t *__restrict__ queries, int64_t num_queries, int k,
                        int64_t *__restrict__ out) {
    __shared__ double cd[KQ][KQ_CAP];
    __shared__ int ci[KQ][KQ_CAP];
    __shared__ float s_tm[KQ][256];
    __shared__ float s_bound[KQ];
    __shared__ int s_cnt[KQ];
    const int tid = threadIdx.x;
    const int64_t q0 = (int64_t)blockIdx.x * KQ;
    const int nq = (int)(num_queries - q0 < KQ ? num_queries - q0 : KQ);
    float qx[KQ], qy[KQ], qz[KQ], tmin[KQ];
#pragma unroll
    for (int j = 0; j < KQ; ++j) {
        const int64_t q = queries[q0 + (j < nq ? j : 0)];
        qx[j] = xyz[q * 3]; qy[j] = xyz[q * 3 + 1]; qz[j] = xyz[q * 3 + 2];
        tmin[j] = INFINITY;
    }
    for (int64_t i = tid; i < n; i += 256) {
        const float px = xyz[i * 3], py = xyz[i * 3 + 1], pz = xyz[i * 3 + 2];
#pragma unroll
        for (int j = 0; j < KQ; ++j) {
            const float dx = px - qx[j], dy = py - qy[j], dz = pz - qz[j];
            const float s32 = dx * dx + dy * dy + dz * dz;
            tmin[j] = s32 < tmin[j] ? s32 : tmin[j];
        }
    }
    if (tid < KQ) s_cnt[tid] = 0;
    {   // the (k+1)-th lowest of the 256 thread minima of every query: bitonic sorts (shuffles inside a wave, LDS across waves)
        float v[KQ];
#pragma unroll
        for (int j = 0; j < KQ; ++j) v[j] = tmin[j];
        for (int kk = 2; kk <= 256; kk <<= 1)
            for (int jj = kk >> 1; jj > 0; jj >>= 1) {
                const bool keep_low = ((tid & jj) == 0) == ((tid & kk) == 0);
                if (jj >= 64) {
                    __syncthreads();
#pragma unroll
                    for (int j = 0; j < KQ; ++j) s_tm[j][tid] = v[j];
                    __syncthreads();
                }
#pragma unroll
                for (int j = 0; j < KQ; ++j) {
                    const float o = jj >= 64 ? s_tm[j][tid ^ jj] : __shfl_xor(v[j], jj, 64);
                    v[j] = keep_low ? (o < v[j] ? o : v[j]) : (o > v[j] ? o : v[j]);
                }
            }
        if (tid == k)
#pragma unroll
            for (int j = 0; j < KQ; ++j) s_bound[j] = v[j];
    }
    __syncthreads();
    float f32[KQ];
    double b64[KQ], qxd[KQ], qyd[KQ], qzd[KQ];
#pragma unroll
    for (int j = 0; j < KQ; ++j) {
        f32[j] = s_bound[j] * (1.0f + 0x1p-18f);
        b64[j] = (double)s_bound[j] * (1.0 + 0x1p-19);
        qxd[j] = qx[j]; qyd[j] = qy[j]; qzd[j] = qz[j];
    }
    for (int64_t i = tid; i < n; i += 256) {
        const float px = xyz[i * 3], py = xyz[i * 3 + 1], pz = xyz[i * 3 + 2];
#pragma unroll
        for (int j = 0; j < KQ; ++j) {
            const float dx = px - qx[j], dy = py - qy[j], dz = pz - qz[j];
            const float s32 = dx * dx + dy * dy + dz * dz;
            if (s32 <= f32[j]) {
                const double ex = (double)px - qxd[j], ey = (double)py - qyd[j], ez = (double)pz - qzd[j];
                const double d2 = ex * ex + ey * ey + ez * ez;
                if (d2 <= b64[j]) { const int pos = atomicAdd(&s_cnt[j], 1); if (pos < KQ_CAP) { cd[j][pos] = d2; ci[j][pos] = (int)i; } }
            }
        }
    }
    __syncthreads();
    int cnt[KQ], np2 = 1;
#pragma unroll
    for (int j = 0; j < KQ; ++j) { cnt[j] = s_cnt[j] < KQ_CAP ? s_cnt[j] : KQ_CAP; while (np2 < cnt[j]) np2 <<= 1; }
#pragma unroll
    for (int j = 0; j < KQ; ++j)
        for (int i = cnt[j] + tid; i < np2; i += 256) { cd[j][i] = INFINITY; ci[j][i] = INT32_MAX; }
    __syncthreads();
    for (int kk = 2; kk <= np2; kk <<= 1)                 // KQ bitonic sorts by (d2, id), side by side
        for (int jj = kk >> 1; jj > 0; jj >>= 1) {
            for (int t = tid; t < KQ * np2; t += 256) {
                const int j = t / np2, i = t - j * np2, ixj = i ^ jj;
                if (ixj > i) {
                    const bool up = (i & kk) == 0;
                    const bool gt = cd[j][i] > cd[j][ixj] || (cd[j][i] == cd[j][ixj] && ci[j][i] > ci[j][ixj]);
                    if (gt == up) { const double td = cd[j][i]; cd[j][i] = cd[j][ixj]; cd[j][ixj] = td; const int ti = ci[j][i]; ci[j][i] = ci[j][ixj]; ci[j][ixj] = ti; }
                }
            }
            __syncthreads();
        }
    for (int j = 0; j < nq; ++j) {
        const bool over = s_cnt[j] > KQ_CAP || !(s_bound[j] >= 1e-30f);
        for (int t = tid; t < k; t += 256) out[(q0 + j) * k + t] = over ? (int64_t)-1 : (int64_t)ci[j][t + 1];      // column 0 (the point itself) dropped
    }
}

// ------------------------------------------------------------------------------------------------ the sampler's selections
// sample_contrastive_pairs_hybrid (affinity_module.py:1116-1124) on one row of the anchors x points similarity per workgroup:
//   positive = arg-max over the points other than the anchor (ties: the lowest index);
//   macro    = the k points of lowest similarity other than the anchor and the positive, ascending by (value, index)
// -- torch.argmax + torch.topk(largest=False) of the reference, which on the device are a 4-pass radix select over the whole 2.4-GB
// matrix plus a gather pass (4.7 ms at 4096 x 150k) and an arg-max sweep (0.5 ms).  Here the row is read from memory ONCE:
//   sweep (HBM, 16 bytes per lane, fully coalesced): every thread keeps the lowest key and the highest (key, -index) of the elements it
//     reads, and the lowest key of every GROUP of 4 << lg consecutive elements (1 << lg neighbouring lanes, joined by shuffles) goes to LDS.
//     The (k+1)-th lowest of the 1024 thread minima is an upper bound B of the k-th lowest selectable element (k+1 distinct elements lie
//     at or below it, at most one of them the positive), and on anything but adversarial data only ~2k elements of the row lie at or below it;
//   collect: the groups whose minimum is <= B (a walk over LDS) are read again -- a few dozen 64-byte pieces -- their elements <= B go to
//     LDS, are ranked by counting, the k lowest written in order.
//   If more than SR_CAP elements lie at or below B (massive ties, or the low values all in a few threads' strides) the candidates are
//     narrowed by a radix select over the 64-bit key (value | index) 12 bits at a time -- keys are unique, so it terminates with <= SR_CAP --
//     one walk over the flagged groups per level, then collected and ranked the same way.
// Keys: the order-preserving uint image of a float (-0 counts as +0; a NaN with a clear sign bit orders above +inf, as in torch).
// LDS: 48 KiB of group minima + 16 KiB of candidates (the rare path's histogram shares them) + 4 KiB: two workgroups per CU.
constexpr int SR_NT = 1024, SR_CAP = 2048, SR_BINS = 4096, SR_GROUPS = 12288;
__device__ __forceinline__ unsigned sr_key(float v) {
    unsigned u = __float_as_uint(v);
    if (u == 0x80000000u) u = 0u;
    return (u >> 31) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ unsigned long long sr_shfl_xor(unsigned long long v, int o) {
    unsigned lo = __shfl_xor((unsigned)v, o, 64), hi = __shfl_xor((unsigned)(v >> 32), o, 64);
    return ((unsigned long long)hi << 32) | lo;
}
// f(index, key) over the elements of the groups whose minimum is at or below `bound`; a group's float4s are loaded together
template <int LG, typename F>
__device__ __forceinline__ void sr_walk(const float *__restrict__ row, int n, bool vec, const unsigned *gmin, int ngroups, unsigned bound, F f) {
    constexpr int gsz = 4 << LG, NB = (1 << LG) < 4 ? (1 << LG) : 4;             // float4s per batch
    for (int g = threadIdx.x; g < ngroups; g += SR_NT)
        if (gmin[g] <= bound) {
            const int e0 = g * gsz;
            if (vec && e0 + gsz <= n) {
                for (int b = 0; b < (1 << LG); b += NB) {
                    float4 a[NB];
#pragma unroll
                    for (int u = 0; u < NB; ++u) a[u] = *reinterpret_cast<const float4 *>(row + e0 + 4 * (b + u));
#pragma unroll
                    for (int u = 0; u < NB; ++u) {
                        const int i = e0 + 4 * (b + u);
                        f(i, sr_key(a[u].x)); f(i + 1, sr_key(a[u].y)); f(i + 2, sr_key(a[u].z)); f(i + 3, sr_key(a[u].w));
                    }
                }
            } else {
                const int e1 = e0 + gsz < n ? e0 + gsz : n;
                for (int i = e0; i < e1; ++i) f(i, sr_key(row[i]));
            }
        }
}
template <int LG>
__global__ void __launch_bounds__(SR_NT, 8)
sampler_select_kernel(const float *__restrict__ sim, int64_t ld, int n, const int64_t *__restrict__ anchor, int k,
                      int64_t *__restrict__ positive, int64_t *__restrict__ macro) {
    constexpr int lg = LG;
    __shared__ __align__(16) unsigned long long cand[SR_CAP];
    __shared__ unsigned gmin[SR_GROUPS];
    __shared__ __align__(16) unsigned s_min[SR_NT];
    __shared__ unsigned long long s_red[SR_NT / 64];
    __shared__ unsigned s_bound;
    __shared__ int s_cnt, s_pos, s_bin, s_acc, s_done;
    unsigned *hist = reinterpret_cast<unsigned *>(cand);                 // [SR_BINS]: the rare path's, before the candidates are collected
    const int tid = threadIdx.x, lane = tid & 63;
    const float *row = sim + (int64_t)blockIdx.x * ld;
    const bool vec = (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(sim) & 15) == 0);
    const int anc = (int)anchor[blockIdx.x];
    const int n4 = (n + 3) >> 2, ngroups = (n4 + (1 << lg) - 1) >> lg;
    // ---- the sweep
    unsigned tmin = 0xffffffffu, tmaxk = 0u;
    int tmaxi = 0x7fffffff;                               // (a thread meets its elements in ascending index order: `>` keeps the lowest index)
    auto slot = [&](int f, const float *v, bool full) {   // one float4 slot: elements 4 f .. 4 f + 3
        const int i0 = 4 * f;
        unsigned gk = 0xffffffffu;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + q;
            if ((full || i < n) && i != anc) {
                const unsigned kx = sr_key(v[q]);
                gk = kx < gk ? kx : gk;
                if (kx > tmaxk) { tmaxk = kx; tmaxi = i; }
            }
        }
        tmin = gk < tmin ? gk : tmin;
#pragma unroll
        for (int o = 1; o < (1 << LG); o <<= 1) { const unsigned t = __shfl_xor(gk, o, 64); gk = t < gk ? t : gk; }
        if ((lane & ((1 << LG) - 1)) == 0 && (full || f < n4)) gmin[f >> LG] = gk;
    };
    int f0 = tid - lane;
    if (vec) {
        // four slots per thread and iteration, their loads issued together (one load in flight per wave left the sweep latency-bound:
        // 2.1 TB/s); every slot of the iteration lies inside the row's complete float4s, so no element test
        const int nfull = n >> 2;
        for (; f0 + 3 * SR_NT + 64 <= nfull; f0 += 4 * SR_NT) {
            float4 a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const float4 *>(row + 4 * (int64_t)(f0 + u * SR_NT + lane));
#pragma unroll
            for (int u = 0; u < 4; ++u) slot(f0 + u * SR_NT + lane, reinterpret_cast<const float *>(&a[u]), true);
        }
    }
    for (; f0 < n4; f0 += SR_NT) {                        // the row's end (and rows that cannot be read 16 bytes at a time)
        const int f = f0 + lane, i0 = 4 * f;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (vec && i0 + 3 < n) {
            *reinterpret_cast<float4 *>(v) = *reinterpret_cast<const float4 *>(row + i0);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (i0 + q < n) v[q] = row[i0 + q];
        }
        slot(f, v, false);
    }
    s_min[tid] = tmin;
    unsigned long long tmax = ((unsigned long long)tmaxk << 32) | (unsigned)(0x7fffffff - tmaxi);     // highest key, then lowest index
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = sr_shfl_xor(tmax, o); tmax = t > tmax ? t : tmax; }
    if (lane == 0) s_red[tid >> 6] = tmax;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    if (tid == 0) {
        unsigned long long m = s_red[0];
        for (int w = 1; w < SR_NT / 64; ++w) m = s_red[w] > m ? s_red[w] : m;
        s_pos = 0x7fffffff - (int)(unsigned)(m & 0xffffffffu);
    }
    {   // the (k+1)-th lowest of the 1024 thread minima: a bitonic sort, exchanges inside a wave by shuffles, across waves through LDS
        // (ranking every minimum against all 1024 by counting -- the first form -- was 1 M compares per row: 65 of a row's ~110 us)
        unsigned v = tmin;
        for (int kk = 2; kk <= SR_NT; kk <<= 1)
            for (int j = kk >> 1; j > 0; j >>= 1) {
                unsigned o;
                if (j >= 64) {
                    __syncthreads();
                    s_min[tid] = v;
                    __syncthreads();
                    o = s_min[tid ^ j];
                } else {
                    o = __shfl_xor(v, j, 64);
                }
                const bool keep_low = ((tid & j) == 0) == ((tid & kk) == 0);
                v = keep_low ? (o < v ? o : v) : (o > v ? o : v);
            }
        if (tid == k) s_bound = v;                        // ascending order: position k holds the (k+1)-th lowest
    }
    __syncthreads();
    const unsigned bound = s_bound;
    const int pos = s_pos;
    if (tid == 0) positive[blockIdx.x] = pos;
    // ---- collect the elements at or below the bound
    sr_walk<LG>(row, n, vec, gmin, ngroups, bound, [&](int i, unsigned kx) {
        if (kx <= bound && i != anc && i != pos) {
            const int p = atomicAdd(&s_cnt, 1);
            if (p < SR_CAP) cand[p] = ((unsigned long long)kx << 32) | (unsigned)i;
        }
    });
    __syncthreads();
    int cnt = s_cnt;
    if (cnt > SR_CAP) {
        // ---- the rare path: radix select on the 64-bit keys of the elements at or below the bound, 12 bits per walk
        unsigned long long prefix = 0ull;
        int acc = 0, shift = 64;
        for (int level = 0;; ++level) {
            const int width = level < 5 ? 12 : 4;
            shift -= width;
            for (int i = tid; i < SR_BINS; i += SR_NT) hist[i] = 0u;
            __syncthreads();
            sr_walk<LG>(row, n, vec, gmin, ngroups, bound, [&](int i, unsigned kx) {
                if (kx <= bound && i != anc && i != pos) {
                    const unsigned long long key = ((unsigned long long)kx << 32) | (unsigned)i;
                    if (level == 0 || (key >> (shift + width)) == prefix) atomicAdd(&hist[(unsigned)(key >> shift) & ((1u << width) - 1u)], 1u);
                }
            });
            __syncthreads();
            if (tid == 0) {
                const int need = k - acc;
                int run = 0, t = 0;
                for (; t < (1 << width) - 1; ++t) { if (run + (int)hist[t] >= need) break; run += (int)hist[t]; }
                s_bin = t;
                s_acc = acc + run;
                s_done = (acc + run + (int)hist[t] <= SR_CAP);
                s_cnt = 0;
            }
            __syncthreads();
            prefix = (prefix << width) | (unsigned)s_bin;
            acc = s_acc;
            const int done = s_done;
            __syncthreads();                              // (the histogram shares the candidates' memory: everyone has read s_* before it is reused)
            if (done) break;                              // the last level's bins hold one key each: acc + 1 <= k
        }
        sr_walk<LG>(row, n, vec, gmin, ngroups, bound, [&](int i, unsigned kx) {
            if (kx <= bound && i != anc && i != pos) {
                const unsigned long long key = ((unsigned long long)kx << 32) | (unsigned)i;
                if ((key >> shift) <= prefix) { const int p = atomicAdd(&s_cnt, 1); if (p < SR_CAP) cand[p] = key; }
            }
        });
        __syncthreads();
        cnt = s_cnt < SR_CAP ? s_cnt : SR_CAP;
    }
    // ---- rank by counting (keys are unique); the k lowest in order
    for (int t = tid; t < cnt; t += SR_NT) {
        const unsigned long long me = cand[t];
        int rank = 0;
        int u = 0;
        for (; u + 2 <= cnt; u += 2) {
            const ulonglong2 c2 = *reinterpret_cast<const ulonglong2 *>(&cand[u]);
            rank += (c2.x < me) + (c2.y < me);
        }
        if (u < cnt) rank += cand[u] < me;
        if (rank < k) macro[(int64_t)blockIdx.x * k + rank] = (int64_t)(unsigned)(me & 0xffffffffu);
    }
}

// rows of x / max(|row|, eps) (F.normalize, affinity_module.py:1114) written as f16 hi/lo planes; rows n .. n_pad of the planes are zero
// (the similarity GEMM's weight operand wants a multiple of 256 rows).  One wave per row, two sweeps of the row (the second from L1/L2).
__global__ void __launch_bounds__(256)
normalize_split_kernel(const float *__restrict__ x, int64_t ld_x, int d, int64_t n, int64_t n_pad, float eps, _Float16 *__restrict__ hi,
                       _Float16 *__restrict__ lo, int64_t ld_h) {
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    const int lane = gp_lane();
    for (int64_t r = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6; r < n_pad; r += ((int64_t)gridDim.x * blockDim.x) >> 6) {
        float inv = 0.f;
        if (r < n) {
            float ss = 0.f;
            for (int c = lane * 4; c < d; c += 256) {
                const float4 a = *reinterpret_cast<const float4 *>(x + r * ld_x + c);
                ss += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
            }
            inv = fmaxf(sqrtf(gp_wave_sum(ss)), eps);
        }
        for (int c = lane * 4; c < d; c += 256) {
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (r < n) {
                const float4 a = *reinterpret_cast<const float4 *>(x + r * ld_x + c);
                v[0] = a.x / inv; v[1] = a.y / inv; v[2] = a.z / inv; v[3] = a.w / inv;
            }
            f16x4 h, l;
#pragma unroll
            for (int q = 0; q < 4; ++q) { h[q] = (_Float16)v[q]; l[q] = (_Float16)(v[q] - (float)h[q]); }
            *reinterpret_cast<f16x4 *>(hi + r * ld_h + c) = h;
            *reinterpret_cast<f16x4 *>(lo + r * ld_h + c) = l;
        }
    }
}

}  // namespace

extern "C" size_t gp_col_stats_workspace_bytes(int64_t nv, int32_t c) {
    int64_t nch = (nv + CS_ROWS - 1) / CS_ROWS;
    return gp_align_up((size_t)nch * 2 * c * sizeof(double), 256);
}

// mean[c], var[c] (biased) of the rows of y: one sweep, fp64 sums of x and x^2 in a fixed order, var = E[x^2] - mean^2 in fp64
extern "C" int gp_col_stats(const float *y, int64_t ld, int64_t nv, int32_t c, float *mean, float *var, void *workspace,
                            size_t workspace_bytes, void *stream_) {
    GP_CHECK_ARG(y && mean && var && workspace && nv > 0 && c > 0, "gp_col_stats: null/empty argument");
    if (workspace_bytes < gp_col_stats_workspace_bytes(nv, c)) { gp_set_error("gp_col_stats: workspace too small"); return GP_ENOMEM; }
    hipStream_t s = gp_stream(stream_);
    double *partial = static_cast<double *>(workspace);
    int64_t nch = (nv + CS_ROWS - 1) / CS_ROWS;
    const bool vec = cs_vec(c, {ld}, {y});
    if (vec) cs_sum2_kernel<4><<<cs_grid(nv, c, true), 256, 0, s>>>(y, ld, nv, c, partial);
    else cs_sum2_kernel<1><<<cs_grid(nv, c, false), 256, 0, s>>>(y, ld, nv, c, partial);
    cs_meanvar_final_kernel<<<(c + 63) / 64, CF_WAVES * 64, 0, s>>>(partial, nch, c, 1.0 / (double)nv, mean, var);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

// fp64 column sums for SyncBatchNorm (run/train.py:212-213 converts the student to MinkowskiSyncBatchNorm): the caller
// all-reduces them over the ranks.  mean == NULL: out[col] = sum_r y[r][col]; else out[col] = sum_r (y[r][col] - mean[col])^2.
__global__ void __launch_bounds__(CF_WAVES * 64) cs_final_f64_kernel(const double *__restrict__ partial, int64_t nchunks, int nq, int c,
                                                                     double *__restrict__ out) {
    cs_final_body(partial, nchunks, nq, c, 1.0, out);
}
extern "C" int gp_col_sums_f64(const float *y, int64_t ld, int64_t nv, int32_t c, const float *mean, double *out, void *workspace,
                               size_t workspace_bytes, void *stream_) {
    GP_CHECK_ARG(y && out && workspace && nv > 0 && c > 0, "gp_col_sums_f64: null/empty argument");
    if (workspace_bytes < gp_col_stats_workspace_bytes(nv, c)) { gp_set_error("gp_col_sums_f64: workspace too small"); return GP_ENOMEM; }
    hipStream_t s = gp_stream(stream_);
    double *partial = static_cast<double *>(workspace);
    int64_t nch = (nv + CS_ROWS - 1) / CS_ROWS;
    const bool vec = cs_vec(c, {ld}, {y});
    if (mean) {
        if (vec) cs_var_kernel<4><<<cs_grid(nv, c, true), 256, 0, s>>>(y, ld, nv, c, mean, partial);
        else cs_var_kernel<1><<<cs_grid(nv, c, false), 256, 0, s>>>(y, ld, nv, c, mean, partial);
    } else {
        if (vec) cs_sum_kernel<4><<<cs_grid(nv, c, true), 256, 0, s>>>(y, ld, nv, c, partial);
        else cs_sum_kernel<1><<<cs_grid(nv, c, false), 256, 0, s>>>(y, ld, nv, c, partial);
    }
    cs_final_f64_kernel<<<(c + 63) / 64, CF_WAVES * 64, 0, s>>>(partial, nch, 1, c, out);
    GP_CHECK_LAUNCH();
    return GP_OK;
}
// the two reduction vectors of the BatchNorm backward pass as fp64 sums: sums[0:c] = sum dz, sums[c:2c] = sum dz * xhat
extern "C" int gp_bn_bwd_sums_f64(const float *dout, int64_t ld_dout, const float *act, int64_t ld_act, const float *y, int64_t ld_y,
                                  const float *mean, const float *var, float eps, const float *gamma_mask, const float *beta_mask, int64_t nv,
                                  int32_t c, double *sums, void *workspace, size_t workspace_bytes, void *stream_) {
    GP_CHECK_ARG(dout && y && mean && var && sums && workspace && nv > 0 && c > 0, "gp_bn_bwd_sums_f64: null/empty argument");
    GP_CHECK_ARG(!beta_mask || (gamma_mask && !act), "gp_bn_bwd_sums_f64: the mask comes from act OR from y (gamma_mask + beta_mask)");
    if (workspace_bytes < gp_col_stats_workspace_bytes(nv, c)) { gp_set_error("gp_bn_bwd_sums_f64: workspace too small"); return GP_ENOMEM; }
    hipStream_t s = gp_stream(stream_);
    double *partial = static_cast<double *>(workspace);
    int64_t nch = (nv + CS_ROWS - 1) / CS_ROWS;
    const bool vec = cs_vec(c, {ld_dout, ld_y, act ? ld_act : 0}, {dout, y, act});
    if (vec) bn_bwd_reduce_kernel<4><<<cs_grid(nv, c, true), 256, 0, s>>>(dout, ld_dout, act, ld_act, y, ld_y, mean, var, eps, gamma_mask, beta_mask, nv, c, partial, nullptr);
    else bn_bwd_reduce_kernel<1><<<cs_grid(nv, c, false), 256, 0, s>>>(dout, ld_dout, act, ld_act, y, ld_y, mean, var, eps, gamma_mask, beta_mask, nv, c, partial, nullptr);
    cs_final_f64_kernel<<<(2 * c + 63) / 64, CF_WAVES * 64, 0, s>>>(partial, nch, 2, c, sums);
    GP_CHECK_LAUNCH();
    return GP_OK;
}
// dy = gamma/sqrt(var+eps) * (dz - sums[col]/n_total - xhat * sums[c+col]/n_total) with caller-supplied (all-reduced) sums
extern "C" int gp_bn_bwd_apply(const float *dout, int64_t ld_dout, const float *act, int64_t ld_act, const float *y, int64_t ld_y,
                               const float *mean, const float *var, float eps, const float *gamma, const float *beta_mask, const float *sums,
                               int64_t n_total, int64_t nv, int32_t c, float *dy, int64_t ld_dy, float *dz_out, int64_t ld_dz, float *dy_scale2,
                               void *stream_) {
    GP_CHECK_ARG(dout && y && mean && var && gamma && sums && dy && nv > 0 && c > 0 && n_total >= nv, "gp_bn_bwd_apply: bad argument");
    GP_CHECK_ARG(!beta_mask || !act, "gp_bn_bwd_apply: the mask comes from act OR from y (beta_mask)");
    int rc = bn_bwd_apply_launch(dout, ld_dout, act, ld_act, y, ld_y, mean, var, eps, gamma, beta_mask, sums, n_total, nv, c, dy, ld_dy, dz_out, ld_dz, dy_scale2,
                                 gp_stream(stream_));
    if (rc != GP_OK) return rc;
    GP_CHECK_LAUNCH();
    return GP_OK;
}

// out = [relu]((y - mean) / sqrt(var + eps) * gamma + beta [+ residual]); optional split copy; optional running-stat update
extern "C" int gp_bn_train_apply(const float *y, int64_t ld, int64_t nv, int32_t c, const float *mean, const float *var,
                                 const float *gamma, const float *beta, float eps, const float *residual, int64_t ld_res,
                                 int32_t relu, float *out, int64_t ld_out, void *out_hi, void *out_lo, int64_t ld_split,
                                 float momentum, float *running_mean, float *running_var, void *stream_) {
    GP_CHECK_ARG(y && mean && var && gamma && beta && (out || out_hi) && nv > 0 && c > 0, "gp_bn_train_apply: null/empty argument");
    GP_CHECK_ARG((out_hi == nullptr) == (out_lo == nullptr), "gp_bn_train_apply: split outputs come as a pair");
    hipStream_t s = gp_stream(stream_);
    auto al = [](const void *p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
    const bool vec = c % 4 == 0 && ld % 4 == 0 && (!out || ld_out % 4 == 0) && (!residual || ld_res % 4 == 0) && (!out_hi || ld_split % 4 == 0) && al(y, 16) &&
                     al(out, 16) && al(residual, 16) && al(out_hi, 8) && al(out_lo, 8);
    const unsigned blocks = bn_sweep_blocks(nv, vec ? c / 4 : c);
    if (vec) bn_apply_kernel<4><<<blocks, 256, 0, s>>>(y, ld, nv, c, mean, var, gamma, beta, eps, residual, ld_res, relu, out, ld_out,
                                                     static_cast<_Float16 *>(out_hi), static_cast<_Float16 *>(out_lo), ld_split);
    else bn_apply_kernel<1><<<blocks, 256, 0, s>>>(y, ld, nv, c, mean, var, gamma, beta, eps, residual, ld_res, relu, out, ld_out,
                                                 static_cast<_Float16 *>(out_hi), static_cast<_Float16 *>(out_lo), ld_split);
    if (running_mean && running_var)
        bn_running_kernel<<<(c + 255) / 256, 256, 0, s>>>(mean, var, c, nv, momentum, running_mean, running_var);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

// dz = dout * (act > 0) (act NULL: dz = dout); dgamma = sum dz*xhat, dbeta = sum dz;
// dy = gamma/sqrt(var+eps) * (dz - dbeta/nv - xhat*dgamma/nv); dz_out (nullable) receives dz (identity branch)
extern "C" size_t gp_bn_train_backward_workspace_bytes(int64_t nv, int32_t c) {
    const int64_t nch = (nv + CS_ROWS - 1) / CS_ROWS;
    return gp_col_stats_workspace_bytes(nv, c) + gp_align_up((size_t)2 * c * sizeof(float), 256) + gp_align_up((size_t)nch * 2 * c * sizeof(float), 256);
}

extern "C" int gp_bn_train_backward(const float *dout, int64_t ld_dout, const float *act, int64_t ld_act, const float *y, int64_t ld_y,
                                    const float *mean, const float *var, float eps, const float *gamma, const float *beta_mask, int64_t nv, int32_t c,
                                    float *dy, int64_t ld_dy, float *dz_out, int64_t ld_dz, float *dgamma, float *dbeta, float *dy_scale2,
                                    void *dy_hi, void *dy_lo, int64_t ld_h, void *workspace, size_t workspace_bytes, void *stream_) {
    GP_CHECK_ARG(dout && y && mean && var && gamma && (dy || dy_hi) && dgamma && dbeta && workspace && nv > 0 && c > 0,
                 "gp_bn_train_backward: null/empty argument");
    GP_CHECK_ARG(!beta_mask || !act, "gp_bn_train_backward: the mask comes from act OR from y (beta_mask)");
    if (workspace_bytes < gp_bn_train_backward_workspace_bytes(nv, c)) { gp_set_error("gp_bn_train_backward: workspace too small"); return GP_ENOMEM; }
    hipStream_t s = gp_stream(stream_);
    double *partial = static_cast<double *>(workspace);
    float *sums = reinterpret_cast<float *>(static_cast<char *>(workspace) + gp_col_stats_workspace_bytes(nv, c));
    float *pmax = reinterpret_cast<float *>(reinterpret_cast<char *>(sums) + gp_align_up((size_t)2 * c * sizeof(float), 256));
    int64_t nch = (nv + CS_ROWS - 1) / CS_ROWS;
    const bool vec = cs_vec(c, {ld_dout, ld_y, act ? ld_act : 0}, {dout, y, act});
    if (dy_hi) {
        // the split planes of dy * s straight from the sweep (s from a bound of max |dy| taken in the reduction pass): no fp32 dy, no split pass
        GP_CHECK_ARG(!dy && dy_lo && dy_scale2 && vec && ld_h % 4 == 0 && ld_h >= c && (!dz_out || ld_dz % 4 == 0) &&
                         ((uintptr_t)dy_hi & 7) == 0 && ((uintptr_t)dy_lo & 7) == 0 && ((uintptr_t)dz_out & 15) == 0,
                     "gp_bn_train_backward: the split form (dy_hi / dy_lo [nv + 1 rows] + dy_scale2, dy = NULL) needs c %% 4 == 0 and 16-byte aligned rows");
        GP_CHECK_HIP(hipMemsetAsync(dy_scale2, 0, 8, s));
        bn_bwd_reduce_kernel<4><<<cs_grid(nv, c, true), 256, 0, s>>>(dout, ld_dout, act, ld_act, y, ld_y, mean, var, eps, gamma, beta_mask, nv, c, partial, pmax);
        bn_bwd_final_bound_kernel<<<(c + 63) / 64, CF_WAVES * 64, 0, s>>>(partial, pmax, nch, c, var, eps, gamma, nv, sums, reinterpret_cast<unsigned *>(dy_scale2));
        bn_scale2_kernel<<<1, 1, 0, s>>>(dy_scale2);
        bn_bwd_apply_split_kernel<<<bn_sweep_blocks(nv, c / 4), 256, 0, s>>>(dout, ld_dout, act, ld_act, y, ld_y, mean, var, eps, gamma, beta_mask, sums, nv, nv, c,
                                                                             dy_scale2, static_cast<_Float16 *>(dy_hi), static_cast<_Float16 *>(dy_lo), ld_h, dz_out, ld_dz);
    } else {
        if (vec) bn_bwd_reduce_kernel<4><<<cs_grid(nv, c, true), 256, 0, s>>>(dout, ld_dout, act, ld_act, y, ld_y, mean, var, eps, gamma, beta_mask, nv, c, partial, nullptr);
        else bn_bwd_reduce_kernel<1><<<cs_grid(nv, c, false), 256, 0, s>>>(dout, ld_dout, act, ld_act, y, ld_y, mean, var, eps, gamma, beta_mask, nv, c, partial, nullptr);
        cs_final_kernel<<<(2 * c + 63) / 64, CF_WAVES * 64, 0, s>>>(partial, nch, 2, c, 1.0, sums);
        int rc = bn_bwd_apply_launch(dout, ld_dout, act, ld_act, y, ld_y, mean, var, eps, gamma, beta_mask, sums, nv, nv, c, dy, ld_dy, dz_out, ld_dz, dy_scale2, s);
        if (rc != GP_OK) return rc;
    }
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
    if (k + 1 <= 256) {
        knn_points_multi_kernel<<<(unsigned)((num_queries + KQ - 1) / KQ), 256, 0, s>>>(xyz, n, queries, num_queries, k, out);
        knn_points_kernel<256><<<(unsigned)num_queries, 256, 0, s>>>(xyz, n, queries, k, out, flag_dev, 1);      // the queries handed back (none, as a rule)
    } else {
        knn_points_kernel<1024><<<(unsigned)num_queries, 1024, 0, s>>>(xyz, n, queries, k, out, flag_dev, 0);
    }
    GP_CHECK_LAUNCH();
    return GP_OK;
}

// positive i64 [A], macro i64 [A, k] of the rows of sim fp32 [A, >= n] (leading dimension ld): see sampler_select_kernel
extern "C" int gp_sampler_select(const float *sim, int64_t ld, int64_t num_anchors, int64_t n, const int64_t *anchor_idx, int32_t k,
                                 int64_t *positive, int64_t *macro, void *stream_) {
    GP_CHECK_ARG(sim && anchor_idx && positive && macro && num_anchors > 0 && ld >= n, "gp_sampler_select: null/empty argument");
    GP_CHECK_ARG(k >= 1 && k < SR_NT && k <= SR_CAP && n >= (int64_t)k + 2 && n <= (int64_t)SR_GROUPS * 256,
                 "gp_sampler_select: k=%d, n=%lld out of range (1 <= k < %d, k + 2 <= n <= %d)", k, (long long)n, SR_NT, SR_GROUPS * 256);
    int lg = 0;                                             // 4 << lg elements per group: the fewest that fit the row's groups into LDS
    while ((((n + 3) >> 2) + (1 << lg) - 1) >> lg > SR_GROUPS) ++lg;
    hipStream_t s = gp_stream(stream_);
#define SR_LAUNCH(L) case L: sampler_select_kernel<L><<<(unsigned)num_anchors, SR_NT, 0, s>>>(sim, ld, (int)n, anchor_idx, k, positive, macro); break;
    switch (lg) { SR_LAUNCH(0) SR_LAUNCH(1) SR_LAUNCH(2) SR_LAUNCH(3) SR_LAUNCH(4) SR_LAUNCH(5) SR_LAUNCH(6) default: break; }
#undef SR_LAUNCH
    GP_CHECK_LAUNCH();
    return GP_OK;
}

// hi + lo = x[r] / max(|x[r]|_2, eps) for r < n, zero rows for n <= r < n_pad
extern "C" int gp_normalize_split_f16(const float *x, int64_t ld_x, int32_t d, int64_t n, int64_t n_pad, float eps, void *hi, void *lo,
                                      int64_t ld_h, void *stream_) {
    GP_CHECK_ARG(x && hi && lo && n > 0 && n_pad >= n && d > 0 && d % 4 == 0 && ld_x % 4 == 0 && ld_h % 4 == 0 && ld_h >= d,
                 "gp_normalize_split_f16: bad argument");
    GP_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 15) == 0, "gp_normalize_split_f16: x must be 16-byte aligned");
    const int64_t waves = n_pad < 16384 ? n_pad : 16384;
    normalize_split_kernel<<<(unsigned)((waves + 3) / 4), 256, 0, gp_stream(stream_)>>>(x, ld_x, d, n, n_pad, eps, static_cast<_Float16 *>(hi),
                                                                                       static_cast<_Float16 *>(lo), ld_h);
    GP_CHECK_LAUNCH();
    return GP_OK;
}
