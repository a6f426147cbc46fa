// Rows 5-7: 2D->3D feature lift.  Dense-map lift, per-view mask-embedding lift (bicubic-antialias
// resize evaluated only at the sampled pixels), point->view CSR, consensus top-3 fusion.
#include <cstring>
#include <rocprim/device/device_scan.hpp>

#include "gp_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// row 5: one wave per visible point; lanes stride the channels.  A point occurs once per view, views
// are issued in order on one stream, so plain read-modify-write reproduces the sequential index_add_.
__global__ void lift_dense_accum_kernel(const float *__restrict__ feat2d, int d, int H, int W,
                                        const int64_t *__restrict__ pt, const int64_t *__restrict__ x,
                                        const int64_t *__restrict__ y, int64_t n_v, float *__restrict__ sum,
                                        int64_t ld_sum, float *__restrict__ cnt) {
    int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    if (i >= n_v) return;
    int lane = gp_lane();
    int64_t p = pt[i];
    int64_t pix = x[i] * W + y[i];
    int64_t plane = (int64_t)H * W;
    for (int c = lane; c < d; c += 64) sum[p * ld_sum + c] += feat2d[c * plane + pix];
    if (lane == 0) cnt[p] += 1.0f;
}

// LSeg-style lift (affinity_module.py:404-433): the reference resizes the [D,h,w] feature map to the image size with
// F.interpolate(bilinear, align_corners=True) and then samples it at the visible pixels; here the resize is evaluated
// only at those pixels.  Arithmetic follows torch's CPU kernel to the bit: source = scale*i in fp32, lambda clipped to
// [0,1], value = fma(w0, a, w1*b) per axis (rows of the two source lines first, then the two lines).
__device__ __forceinline__ void bilinear_tap(float scale, int64_t i, int in_size, int &i0, int &i1, float &w0, float &w1) {
    float real = scale * (float)i;
    int f = (int)floorf(real);
    i0 = f < in_size - 1 ? f : in_size - 1;
    float lam = real - (float)i0;
    lam = lam < 0.f ? 0.f : (lam > 1.f ? 1.f : lam);
    i1 = i0 + 1 < in_size - 1 ? i0 + 1 : in_size - 1;
    w0 = 1.f - lam;
    w1 = lam;
}
__global__ void lift_dense_bilinear_accum_kernel(const float *__restrict__ feat, int d, int h, int w, float scale_h,
                                                 float scale_w, const int64_t *__restrict__ pt,
                                                 const int64_t *__restrict__ x, const int64_t *__restrict__ y, int64_t n_v,
                                                 float *__restrict__ sum, int64_t ld_sum, float *__restrict__ cnt) {
    int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    if (i >= n_v) return;
    int lane = gp_lane();
    int64_t p = pt[i];
    int r0, r1, c0, c1;
    float wr0, wr1, wc0, wc1;
    bilinear_tap(scale_h, x[i], h, r0, r1, wr0, wr1);
    bilinear_tap(scale_w, y[i], w, c0, c1, wc0, wc1);
    int64_t plane = (int64_t)h * w;
    for (int c = lane; c < d; c += 64) {
        const float *f = feat + c * plane;
        float top = __builtin_fmaf(wc0, f[(int64_t)r0 * w + c0], wc1 * f[(int64_t)r0 * w + c1]);
        float bot = __builtin_fmaf(wc0, f[(int64_t)r1 * w + c0], wc1 * f[(int64_t)r1 * w + c1]);
        sum[p * ld_sum + c] += __builtin_fmaf(wr0, top, wr1 * bot);
    }
    if (lane == 0) cnt[p] += 1.0f;
}

__global__ void lift_dense_finish_kernel(float *__restrict__ sum, int64_t ld_sum, int d, const float *__restrict__ cnt,
                                         int64_t n, uint8_t *__restrict__ seen) {
    int64_t p = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    if (p >= n) return;
    int lane = gp_lane();
    float c = cnt[p];
    if (c == 0.f) c = 1e-6f;
    for (int k = lane; k < d; k += 64) sum[p * ld_sum + k] = sum[p * ld_sum + k] / c;
    if (lane == 0) seen[p] = c > 1e-5f ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// [Q, hw] -> [hw, Q] so that the Q logits of one low-res pixel are contiguous
__global__ void transpose_kernel(const float *__restrict__ src, int rows, int cols, float *__restrict__ dst) {
    __shared__ float tile[64][65];
    int bx = blockIdx.x * 64, by = blockIdx.y * 64;
    int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;          // 256 threads: 4 rows per pass
    for (int r = ty; r < 64; r += 4) {
        int row = by + r, col = bx + tx;
        tile[r][tx] = (row < rows && col < cols) ? src[(int64_t)row * cols + col] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        int orow = bx + r, ocol = by + tx;                      // dst[col][row]
        if (orow < cols && ocol < rows) dst[(int64_t)orow * rows + ocol] = tile[tx][r];
    }
}

// row 6: one wave per visible point, lanes over queries.  Separable resize in the torch CPU order:
// horizontal taps first (t = s0*w0; t = fma(s_i, w_i, t)), then vertical over the row results.
__device__ __forceinline__ void lift_masks_point(const float *__restrict__ mt /*[h*w, Q]*/, int Q, int h, int w,
                                                 const float *__restrict__ scores, const int32_t *__restrict__ tx0,
                                                 const float *__restrict__ twx, const int32_t *__restrict__ ty0,
                                                 const float *__restrict__ twy, int out_h, int out_w, int row, int col, int lane,
                                                 int &seg_out, float &logit_out) {
    bool inb = (unsigned)row < (unsigned)out_h && (unsigned)col < (unsigned)out_w;
    float best = -1.f, best_logit = 0.f;
    int best_q = -1;
    if (inb) {
        int x0 = tx0[col], y0 = ty0[row];
        float wx[4], wy[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { wx[t] = twx[col * 4 + t]; wy[t] = twy[row * 4 + t]; }
        for (int q = lane; q < Q; q += 64) {
            float sc = scores[q];
            float hr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int yy = min(y0 + j, h - 1);
                const float *r = mt + ((int64_t)yy * w) * Q + q;
                float t = __fmul_rn(r[(int64_t)min(x0, w - 1) * Q], wx[0]);
#pragma unroll
                for (int a = 1; a < 4; ++a) t = fmaf(r[(int64_t)min(x0 + a, w - 1) * Q], wx[a], t);
                hr[j] = t;
            }
            float v = __fmul_rn(hr[0], wy[0]);
#pragma unroll
            for (int j = 1; j < 4; ++j) v = fmaf(hr[j], wy[j], v);
            if (sc > 0.f) {
                float sg = 1.f / (1.f + expf(-v));
                float pr = __fmul_rn(sc, sg);
                if (pr > best) { best = pr; best_q = q; best_logit = v; }   // ascending q: first max kept
            }
        }
    }
    // wave arg-max: larger value, then smaller query index
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ob = __shfl_xor(best, o, 64);
        int oq = __shfl_xor(best_q, o, 64);
        float ol = __shfl_xor(best_logit, o, 64);
        bool take = (oq >= 0) && (best_q < 0 || ob > best || (ob == best && oq < best_q));
        if (take) { best = ob; best_q = oq; best_logit = ol; }
    }
    float sg = 1.f / (1.f + expf(-best_logit));
    seg_out = (best_q >= 0 && sg >= 0.5f) ? best_q : -1;
    logit_out = best_logit;
}
__global__ void lift_masks_kernel(const float *__restrict__ mt /*[h*w, Q]*/, int Q, int h, int w,
                                  const float *__restrict__ scores, const int32_t *__restrict__ tx0,
                                  const float *__restrict__ twx, const int32_t *__restrict__ ty0,
                                  const float *__restrict__ twy, int out_h, int out_w,
                                  const int64_t *__restrict__ px, const int64_t *__restrict__ py, int64_t n_v,
                                  int32_t *__restrict__ seg, float *__restrict__ seg_logit) {
    int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    if (i >= n_v) return;
    int lane = gp_lane();
    int sg;
    float lg;
    lift_masks_point(mt, Q, h, w, scores, tx0, twx, ty0, twy, out_h, out_w, (int)px[i], (int)py[i], lane, sg, lg);   // x_label = pixel row
    if (lane == 0) {
        seg[i] = sg;
        if (seg_logit) seg_logit[i] = lg;
    }
}

// ---- the same for ALL views of a scene: entries e = (view, point, pixel) in view-major order (gp_views_visible_lists); the masks
// of view v are mt + v * h*w*Q, its scores scores + v * Q.  Entries of dropped views get -1.
// (order: the queries of view blockIdx.z by descending score, lv_sort_scores_kernel -- row r of the transposed copy is query order[r])
__global__ void transpose_views_kernel(const float *__restrict__ src, int rows, int cols, float *__restrict__ dst,
                                       const int32_t *__restrict__ order) {
    __shared__ float tile[64][65];
    const int64_t vo = (int64_t)blockIdx.z * rows * cols;
    const int32_t *ord = order + (int64_t)blockIdx.z * rows;
    int bx = blockIdx.x * 64, by = blockIdx.y * 64;
    int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    __shared__ int s_ord[64];                                  // (one load per tile row instead of a dependent load in front of every read)
    if (threadIdx.x < 64) s_ord[threadIdx.x] = by + threadIdx.x < rows ? ord[by + threadIdx.x] : 0;
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        int row = by + r, col = bx + tx;
        tile[r][tx] = (row < rows && col < cols) ? src[vo + (int64_t)s_ord[r] * cols + col] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        int orow = bx + r, ocol = by + tx;
        if (orow < cols && ocol < rows) dst[vo + (int64_t)orow * rows + ocol] = tile[tx][r];
    }
}
// the queries of every view by (score descending, index ascending): order i32 [V, Q] (rank -> query), sscore f32 [V, Q].  One
// workgroup per view, bitonic sort of 64-bit keys in LDS (Q <= 1024).
__global__ void __launch_bounds__(256) lv_sort_scores_kernel(const float *__restrict__ scores, int Q, int32_t *__restrict__ order,
                                                             float *__restrict__ sscore) {
    __shared__ unsigned long long key[1024];
    const int v = blockIdx.x;
    int np2 = 1;
    while (np2 < Q) np2 <<= 1;
    for (int i = threadIdx.x; i < np2; i += 256) {
        unsigned long long k = ~0ull;
        if (i < Q) {
            unsigned u = __float_as_uint(scores[(int64_t)v * Q + i]);
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);                     // monotone in the float's value
            k = ((unsigned long long)(~u) << 32) | (unsigned)i;                // ascending key = descending score, then ascending index
        }
        key[i] = k;
    }
    __syncthreads();
    for (int kk = 2; kk <= np2; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < np2; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long x = key[i], y = key[ixj];
                    if ((x > y) == ((i & kk) == 0)) { key[i] = y; key[ixj] = x; }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < Q; i += 256) {
        const int qo = (int)(key[i] & 0xFFFFFFFFull);
        order[(int64_t)v * Q + i] = qo;
        sscore[(int64_t)v * Q + i] = scores[(int64_t)v * Q + qo];
    }
}
// One wave per entry on the score-ordered copy.  The candidates of a pixel are the Q products score_q x sigmoid(logit_q) <= score_q:
// with the queries in descending score order, a pass of 64 can be skipped -- with every pass behind it -- as soon as its LARGEST
// score is strictly below the best product found so far (nothing in it can win or tie).  The decision is the one of
// lift_masks_point (largest product, then smallest query index; the same arithmetic per candidate), reached after 1.1 passes
// instead of 4 on the S scene's synthetic masks (92 % of the pixels stop after the first 64 queries).
__device__ __forceinline__ void lift_masks_point_sorted(const float *__restrict__ mt /*[h*w, Q] in rank order*/, int Q, int h, int w,
                                                        const float *__restrict__ sscore, const int32_t *__restrict__ order,
                                                        const int32_t *__restrict__ tx0, const float *__restrict__ twx,
                                                        const int32_t *__restrict__ ty0, const float *__restrict__ twy, int out_h, int out_w,
                                                        int row, int col, int lane, int &seg_out) {
    bool inb = (unsigned)row < (unsigned)out_h && (unsigned)col < (unsigned)out_w;
    float best = -1.f, best_logit = 0.f;
    int best_q = -1;
    if (inb) {
        int x0 = tx0[col], y0 = ty0[row];
        float wx[4], wy[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { wx[t] = twx[col * 4 + t]; wy[t] = twy[row * 4 + t]; }
        for (int q0 = 0; q0 < Q; q0 += 64) {
            if (q0 > 0) {                                                     // wave-uniform: can the rest still matter?
                float wb = best;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) wb = fmaxf(wb, __shfl_xor(wb, o, 64));
                if (sscore[q0] < wb) break;
            }
            const int q = q0 + lane;
            if (q < Q) {
                float sc = sscore[q];
                float hr[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int yy = min(y0 + j, h - 1);
                    const float *r = mt + ((int64_t)yy * w) * Q + q;
                    float t = __fmul_rn(r[(int64_t)min(x0, w - 1) * Q], wx[0]);
#pragma unroll
                    for (int a = 1; a < 4; ++a) t = fmaf(r[(int64_t)min(x0 + a, w - 1) * Q], wx[a], t);
                    hr[j] = t;
                }
                float v = __fmul_rn(hr[0], wy[0]);
#pragma unroll
                for (int j = 1; j < 4; ++j) v = fmaf(hr[j], wy[j], v);
                if (sc > 0.f) {
                    float sg = 1.f / (1.f + expf(-v));
                    float pr = __fmul_rn(sc, sg);
                    const int qo = order[q];
                    if (pr > best || (pr == best && qo < best_q)) { best = pr; best_q = qo; best_logit = v; }
                }
            }
        }
    }
    // wave arg-max: larger value, then smaller (original) query index
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ob = __shfl_xor(best, o, 64);
        int oq = __shfl_xor(best_q, o, 64);
        float ol = __shfl_xor(best_logit, o, 64);
        bool take = (oq >= 0) && (best_q < 0 || ob > best || (ob == best && oq < best_q));
        if (take) { best = ob; best_q = oq; best_logit = ol; }
    }
    float sg = 1.f / (1.f + expf(-best_logit));
    seg_out = (best_q >= 0 && sg >= 0.5f) ? best_q : -1;
}
// Measured and left out (round 4): the entries of a view come in POINT order, scattered over the image, and the kernel fetches 4.6 GB
// from beyond L2 per S scene for 0.39 GB of transposed logits (PMC: 53 % L2 hits); with the entries radix-sorted by (view, 8 x 8 pixel
// tile) it takes 524 instead of 545 us and the sort costs 80 us: the wave's 16 taps x Q multiply-adds and their dependent loads
// bound it, not the bytes.  Four consecutive queries per lane with 16-byte loads (Q = 200: one pass of 16 loads on 50 lanes instead of
// four passes of 16 dword loads, the last with 8 live lanes; same bits) is SLOWER: 733 us -- the four passes keep four times as many
// loads in flight.  Round 5: a tile-major form (bins of 16 low-resolution cells staged in LDS from the [Q, h, w] layout): 1.9 ms
// (profiles/r05_lift_tile_major.log); what pays is doing less of the work: the score order above.
__global__ void lift_masks_views_kernel(const float *__restrict__ mt /*[V, h*w, Q] in rank order*/, int Q, int h, int w,
                                        const float *__restrict__ sscore /*[V,Q]*/, const int32_t *__restrict__ order /*[V,Q]*/,
                                        const int32_t *__restrict__ tx0, const float *__restrict__ twx, const int32_t *__restrict__ ty0,
                                        const float *__restrict__ twy, int out_h, int out_w,
                                        const int64_t *__restrict__ ent_x, const int64_t *__restrict__ ent_y,
                                        const int32_t *__restrict__ ent_view, const uint8_t *__restrict__ keep, int64_t total,
                                        int32_t *__restrict__ seg) {
    int64_t e = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    if (e >= total) return;
    int lane = gp_lane();
    // one wave per entry: its view and pixel are wave-uniform -- said so to the compiler (readfirstlane), the tap tables of the pixel
    // (2 + 8 values) then come through the scalar cache instead of ten 64-lane loads of one address each
    const int v = __builtin_amdgcn_readfirstlane(ent_view[e]);
    const int ex = __builtin_amdgcn_readfirstlane((int)ent_x[e]), ey = __builtin_amdgcn_readfirstlane((int)ent_y[e]);
    int sg = -1;
    if (keep[v])                                                 // wave-uniform
        lift_masks_point_sorted(mt + (int64_t)v * h * w * Q, Q, h, w, sscore + (int64_t)v * Q, order + (int64_t)v * Q, tx0, twx, ty0, twy,
                                out_h, out_w, ex, ey, lane, sg);
    if (lane == 0) seg[e] = sg;
}

// ---- in-view fill for all views (affinity_module.py:604-625): every entry without a segment takes the segment of the
// nearest entry WITH one in the same view -- lexicographic minimum of (squared distance in fp64, entry index), the rule of
// gp_nn1_masked_f64.  Covered entries are compacted (exclusive scan of the flags: view-major order is kept, so a view's
// references are one contiguous range), queries likewise; blocks of 256 queries of one view stream that view's references
// through LDS tiles, the reference range cut into FV_CHUNKS parts for parallelism, then the parts are reduced in order.
constexpr int FV_CHUNKS = 16, FV_TILE = 1024;
__global__ void fill_flags_kernel(const int32_t *__restrict__ seg, int64_t total, int32_t *__restrict__ cov /*[total+1]*/) {
    int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e < total) cov[e] = seg[e] >= 0 ? 1 : 0;
    else if (e == total) cov[e] = 0;
}
__global__ void fill_compact_kernel(const float *__restrict__ xyz, const int64_t *__restrict__ ent_pt, const int32_t *__restrict__ seg,
                                    const int32_t *__restrict__ rs /*[total+1]*/, int64_t total, float *__restrict__ rxyz,
                                    int32_t *__restrict__ rent, int32_t *__restrict__ qent) {
    int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int r = rs[e];
    if (seg[e] >= 0) {
        const int64_t p = ent_pt[e];
        rxyz[(int64_t)r * 3] = xyz[p * 3]; rxyz[(int64_t)r * 3 + 1] = xyz[p * 3 + 1]; rxyz[(int64_t)r * 3 + 2] = xyz[p * 3 + 2];
        rent[r] = (int32_t)e;
    } else {
        qent[e - r] = (int32_t)e;
    }
}
// per-view table for the fill kernels, built once per scene by one small launch: tab[v] = {first query, first reference,
// first 256-query block} (+ a sentinel row): a block finds its view by a binary search over the block column instead of
// walking the view offsets (four dependent global loads per view and block: 0.19 ms per scene in 62k mostly idle blocks)
__global__ void fill_table_kernel(const int64_t *__restrict__ view_off, const int32_t *__restrict__ rs, int nviews, int4 *__restrict__ tab) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int blk = 0;
    for (int v = 0; v <= nviews; ++v) {
        const int64_t o = view_off[v];
        const int r = rs[o], q = (int)(o - r);
        tab[v] = make_int4(q, r, blk, 0);
        if (v < nviews) {
            const int64_t o1 = view_off[v + 1];
            const int q1 = (int)(o1 - rs[o1]);
            blk += (q1 - q + 255) / 256;
        }
    }
}
// block -> (view, block of 256 queries of that view); blocks beyond the last view's exit
__device__ __forceinline__ bool fill_locate(const int4 *__restrict__ tab, int nviews, int bx, int &q0, int &q1, int &r0, int &r1) {
    if (bx >= tab[nviews].z) return false;
    int lo = 0, hi = nviews - 1;
    while (lo < hi) {                                            // last view whose first block is <= bx (empty views share a start)
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].z <= bx) lo = mid; else hi = mid - 1;
    }
    const int4 a = tab[lo], b = tab[lo + 1];
    q0 = a.x + (bx - a.z) * 256;
    q1 = b.x;
    r0 = a.y;
    r1 = b.y;
    return true;
}
// The partial results of the FV_CHUNKS reference parts live in part_d / part_i [FV_CHUNKS][stride], indexed by the QUERY's compact
// index -- so `stride` only has to cover the fill queries (uncovered entries), not every entry (192 B per entry,
// 0.7 GB at config M).  The number of queries is known on the device only; the caller passes a capacity (fill_cap) and a block of
// queries that does not fit below it takes the overflow path: ONE block (blockIdx.y == 0) sweeps the view's whole reference
// range and writes the result itself -- same (d2, index) rule, same result, no partials, just less parallel.
__global__ void __launch_bounds__(256)
fill_part_kernel(const float *__restrict__ xyz, const int64_t *__restrict__ ent_pt, const int4 *__restrict__ tab, int nviews,
                 const float *__restrict__ rxyz, const int32_t *__restrict__ qent,
                 double *__restrict__ part_d, int32_t *__restrict__ part_i, int64_t stride,
                 const int32_t *__restrict__ rent, int32_t *__restrict__ seg) {
    __shared__ double sx[FV_TILE], sy[FV_TILE], sz[FV_TILE];
    int q0, q1, vr0, vr1;
    if (!fill_locate(tab, nviews, blockIdx.x, q0, q1, vr0, vr1)) return;                  // block-uniform
    const int n_ref = vr1 - vr0;
    if (n_ref == 0) return;
    const bool overflow = (int64_t)q0 + 256 > stride;                                     // block-uniform
    if (overflow && blockIdx.y != 0) return;
    const int per = overflow ? n_ref : (n_ref + FV_CHUNKS - 1) / FV_CHUNKS;
    const int r0 = vr0 + blockIdx.y * per, r1 = r0 + per < vr1 ? r0 + per : vr1;
    const int qi = q0 + threadIdx.x;
    const bool live = qi < q1;
    double qx = 0, qy = 0, qz = 0;
    if (live) {
        const int64_t p = ent_pt[qent[qi]];
        qx = xyz[p * 3]; qy = xyz[p * 3 + 1]; qz = xyz[p * 3 + 2];
    }
    double best = INFINITY;
    int bi = -1;
    for (int t0 = r0; t0 < r1; t0 += FV_TILE) {
        int cnt = r1 - t0 < FV_TILE ? r1 - t0 : FV_TILE;
        __syncthreads();
        for (int j = threadIdx.x; j < cnt; j += 256) {
            sx[j] = rxyz[(int64_t)(t0 + j) * 3]; sy[j] = rxyz[(int64_t)(t0 + j) * 3 + 1]; sz[j] = rxyz[(int64_t)(t0 + j) * 3 + 2];
        }
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {
            double dx = qx - sx[j], dy = qy - sy[j], dz = qz - sz[j];
            double d2 = (dx * dx + dy * dy) + dz * dz;
            if (d2 < best) { best = d2; bi = t0 + j; }
        }
    }
    if (!live) return;
    if (overflow) {
        if (bi >= 0) seg[qent[qi]] = seg[rent[bi]];              // references keep their segment: no read/write overlap
        return;
    }
    part_d[(int64_t)blockIdx.y * stride + qi] = best;
    part_i[(int64_t)blockIdx.y * stride + qi] = bi;
}
__global__ void fill_reduce_kernel(const int4 *__restrict__ tab, int nviews,
                                   const double *__restrict__ part_d, const int32_t *__restrict__ part_i, int64_t stride,
                                   const int32_t *__restrict__ rent, const int32_t *__restrict__ qent, int32_t *__restrict__ seg) {
    int q0, q1, vr0, vr1;
    if (!fill_locate(tab, nviews, blockIdx.x, q0, q1, vr0, vr1)) return;
    if (vr1 == vr0) return;
    if ((int64_t)q0 + 256 > stride) return;                      // an overflow block wrote its results itself
    const int qi = q0 + threadIdx.x;
    if (qi >= q1) return;
    const int per = (vr1 - vr0 + FV_CHUNKS - 1) / FV_CHUNKS;
    double best = INFINITY;
    int bi = -1;
    for (int c = 0; c < FV_CHUNKS; ++c) {
        if (vr0 + c * per >= vr1) break;                         // chunks beyond the range wrote nothing
        double d = part_d[(int64_t)c * stride + qi];
        int i = part_i[(int64_t)c * stride + qi];
        if (i >= 0 && d < best) { best = d; bi = i; }
    }
    if (bi >= 0) seg[qent[qi]] = seg[rent[bi]];                  // references keep their segment: no read/write overlap
}

// ---- point -> (view, segment) CSR for all views at once.  vmask[p] = bit set of the kept views that see p (<= 128 views),
// a point's entries are ordered by view like the sequential per-view fill: position = start[p] + popcount(lower views).
__global__ void pv_mask_kernel(const int64_t *__restrict__ ent_pt, const int32_t *__restrict__ ent_view, const uint8_t *__restrict__ keep,
                               int64_t total, unsigned long long *__restrict__ vmask) {
    int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int v = ent_view[e];
    if (keep[v]) atomicOr(&vmask[ent_pt[e] * 2 + (v >> 6)], 1ull << (v & 63));
}
__global__ void pv_cnt_kernel(const unsigned long long *__restrict__ vmask, int64_t n, int64_t *__restrict__ cnt /*[n+1]*/) {
    int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p < n) cnt[p] = __popcll(vmask[2 * p]) + __popcll(vmask[2 * p + 1]);
    else if (p == n) cnt[p] = 0;
}
__global__ void pv_fill_views_kernel(const int64_t *__restrict__ ent_pt, const int32_t *__restrict__ ent_view,
                                     const uint8_t *__restrict__ keep, const int32_t *__restrict__ seg, int64_t total,
                                     const unsigned long long *__restrict__ vmask, const int64_t *__restrict__ start,
                                     int32_t *__restrict__ pv_view, int32_t *__restrict__ pv_seg) {
    int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int v = ent_view[e];
    if (!keep[v]) return;
    const int64_t p = ent_pt[e];
    const unsigned long long m0 = vmask[2 * p], m1 = vmask[2 * p + 1];
    const int rank = v < 64 ? __popcll(m0 & ((1ull << v) - 1ull)) : __popcll(m0) + __popcll(m1 & ((1ull << (v - 64)) - 1ull));
    const int64_t s_ = start[p] + rank;
    pv_view[s_] = v;
    pv_seg[s_] = seg[e];
}

// per-segment tables: f_seg = normalize(embed), logit_seg = scale * f_seg . text_norm
__global__ void segment_tables_kernel(const float *__restrict__ emb, int Q, int d, const float *__restrict__ text,
                                      int C, float scale, float *__restrict__ fseg, float *__restrict__ lseg) {
    int q = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (q >= Q) return;
    int lane = gp_lane();
    float ss = 0.f;
    for (int c = lane; c < d; c += 64) { float v = emb[(int64_t)q * d + c]; ss += v * v; }
    ss = gp_wave_sum(ss);
    float nrm = fmaxf(sqrtf(ss), 1e-12f);
    for (int c = lane; c < d; c += 64) fseg[(int64_t)q * d + c] = emb[(int64_t)q * d + c] / nrm;
    for (int k = 0; k < C; ++k) {
        float dot = 0.f;
        for (int c = lane; c < d; c += 64) dot += (emb[(int64_t)q * d + c] / nrm) * text[(int64_t)k * d + c];
        dot = gp_wave_sum(dot);
        if (lane == 0) lseg[(int64_t)q * C + k] = scale * dot;
    }
}

// ------------------------------------------------------------------------------------------------
// point -> (view, segment) CSR
__global__ void pv_count_kernel(const int64_t *__restrict__ pt, int64_t n_v, int64_t *__restrict__ cnt) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n_v) cnt[pt[i]] += 1;                                // a point occurs once per view
}
__global__ void pv_fill_kernel(const int64_t *__restrict__ pt, const int32_t *__restrict__ seg, int64_t n_v, int view,
                               const int64_t *__restrict__ start, int32_t *__restrict__ cursor,
                               int32_t *__restrict__ pv_view, int32_t *__restrict__ pv_seg) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n_v) return;
    int64_t p = pt[i];
    int64_t s = start[p] + cursor[p];
    cursor[p] += 1;
    pv_view[s] = view;
    pv_seg[s] = seg[i];
}

// row 7: one wave per point
__global__ void fuse_top3_kernel(const int64_t *__restrict__ start, const int32_t *__restrict__ pv_view,
                                 const int32_t *__restrict__ pv_seg, int64_t n, const float *__restrict__ fseg,
                                 const float *__restrict__ lseg, int Q, int d, int C, float *__restrict__ out,
                                 int64_t ld_out, uint8_t *__restrict__ seen) {
    int64_t p = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    if (p >= n) return;
    int lane = gp_lane();
    int64_t b = start[p], e = start[p + 1];
    int M = (int)(e - b);
    if (lane == 0 && seen) seen[p] = M > 0 ? 1 : 0;
    if (M == 0) {
        for (int c = lane * 4; c < d; c += 256) *reinterpret_cast<float4 *>(out + p * ld_out + c) = make_float4(0, 0, 0, 0);
        return;
    }
    // a point's (view, segment) entries, one per lane (M <= 64: one round of loads; the loops below read them by shuffle
    // instead of re-loading pv_view / pv_seg in every iteration -- the dependent loads were this kernel's time).  Points seen
    // by more than 64 views take the plain loops.
    const bool in_regs = M <= 64;
    int my_sg = -1, my_v = 0;
    if (in_regs && lane < M) { my_sg = pv_seg[b + lane]; my_v = pv_view[b + lane]; }
    // consensus class: argmax_c of (sum_v logits_v[c]) / M, first maximum (sum in ascending entry order)
    float bestv = -INFINITY;
    int bestc = 0x7fffffff;
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + lane;
        float sum = 0.f;
        if (in_regs) {
            for (int j0 = 0; j0 < M; j0 += 4) {                      // four independent logit loads in flight, added in order
                float t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = j0 + u;
                    const int sg = __shfl(my_sg, j < M ? j : 0, 64), vw = __shfl(my_v, j < M ? j : 0, 64);
                    t[u] = (j < M && sg >= 0 && c < C) ? lseg[((int64_t)vw * Q + sg) * C + c] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (j0 + u < M) sum += t[u];
            }
        } else if (c < C) {
            for (int64_t j = b; j < e; ++j) { int sg = pv_seg[j]; sum += sg >= 0 ? lseg[((int64_t)pv_view[j] * Q + sg) * C + c] : 0.f; }
        }
        if (c < C) {
            sum = sum / (float)M;
            if (sum > bestv) { bestv = sum; bestc = c; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(bestv, o, 64);
        int oc = __shfl_xor(bestc, o, 64);
        if (ov > bestv || (ov == bestv && oc < bestc)) { bestv = ov; bestc = oc; }
    }
    // top-3 views by agreement score (stable: earlier view wins ties)
    float s0 = -INFINITY, s1 = -INFINITY, s2 = -INFINITY;
    int64_t j0 = -1, j1 = -1, j2 = -1;
    if (in_regs) {
        const float my_s = (lane < M && my_sg >= 0) ? lseg[((int64_t)my_v * Q + my_sg) * C + bestc] : 0.f;   // all scores in one round
        for (int jj = 0; jj < M; ++jj) {
            const float sc = __shfl(my_s, jj, 64);
            const int64_t j = b + jj;
            if (sc > s0) { s2 = s1; j2 = j1; s1 = s0; j1 = j0; s0 = sc; j0 = j; }
            else if (sc > s1) { s2 = s1; j2 = j1; s1 = sc; j1 = j; }
            else if (sc > s2) { s2 = sc; j2 = j; }
        }
    } else {
        for (int64_t j = b; j < e; ++j) {
            int sg = pv_seg[j];
            float sc = sg >= 0 ? lseg[((int64_t)pv_view[j] * Q + sg) * C + bestc] : 0.f;
            if (sc > s0) { s2 = s1; j2 = j1; s1 = s0; j1 = j0; s0 = sc; j0 = j; }
            else if (sc > s1) { s2 = s1; j2 = j1; s1 = sc; j1 = j; }
            else if (sc > s2) { s2 = sc; j2 = j; }
        }
    }
    // softmax over the top-min(M,3) scores (missing slots are -inf -> weight 0)
    float e0 = 1.f, e1 = (j1 >= 0) ? expf(s1 - s0) : 0.f, e2 = (j2 >= 0) ? expf(s2 - s0) : 0.f;
    float den = e0 + e1 + e2;
    float w0 = e0 / den, w1 = e1 / den, w2 = e2 / den;
    // a slot whose segment is -1 carries a zero feature: drop its term (weight still in the softmax)
    if (pv_seg[j0] < 0) w0 = 0.f;
    if (j1 >= 0 && pv_seg[j1] < 0) { w1 = 0.f; }
    if (j2 >= 0 && pv_seg[j2] < 0) { w2 = 0.f; }
    auto frow = [&](int64_t j) { int sg = pv_seg[j] < 0 ? 0 : pv_seg[j]; return fseg + ((int64_t)pv_view[j] * Q + sg) * d; };
    const float *f0 = frow(j0);
    const float *f1 = j1 >= 0 ? frow(j1) : f0;
    const float *f2 = j2 >= 0 ? frow(j2) : f0;
    for (int c = lane * 4; c < d; c += 256) {
        float4 a = *reinterpret_cast<const float4 *>(f0 + c);
        float4 r = make_float4(a.x * w0, a.y * w0, a.z * w0, a.w * w0);
        if (j1 >= 0) {
            float4 t = *reinterpret_cast<const float4 *>(f1 + c);
            r.x += t.x * w1; r.y += t.y * w1; r.z += t.z * w1; r.w += t.w * w1;
        }
        if (j2 >= 0) {
            float4 t = *reinterpret_cast<const float4 *>(f2 + c);
            r.x += t.x * w2; r.y += t.y * w2; r.z += t.z * w2; r.w += t.w * w2;
        }
        *reinterpret_cast<float4 *>(out + p * ld_out + c) = r;
    }
}

}  // namespace

// ================================================================================================
extern "C" int gp_lift_dense_accum(const float *feat2d, int32_t d, int32_t height, int32_t width, const int64_t *pt,
                                   const int64_t *x, const int64_t *y, int64_t n_v, float *sum, int64_t ld_sum,
                                   float *cnt, void *stream_) {
    GP_CHECK_ARG(feat2d && pt && x && y && sum && cnt, "gp_lift_dense_accum: null argument");
    if (n_v == 0) return GP_OK;
    lift_dense_accum_kernel<<<(int)((n_v * 64 + 255) / 256), 256, 0, gp_stream(stream_)>>>(feat2d, d, height, width, pt,
                                                                                            x, y, n_v, sum, ld_sum, cnt);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" int gp_lift_dense_bilinear_accum(const float *feat, int32_t d, int32_t h, int32_t w, int32_t out_h, int32_t out_w,
                                            const int64_t *pt, const int64_t *x, const int64_t *y, int64_t n_v, float *sum,
                                            int64_t ld_sum, float *cnt, void *stream_) {
    GP_CHECK_ARG(feat && pt && x && y && sum && cnt, "gp_lift_dense_bilinear_accum: null argument");
    GP_CHECK_ARG(d > 0 && h > 0 && w > 0 && out_h > 0 && out_w > 0, "gp_lift_dense_bilinear_accum: bad shape");
    if (n_v == 0) return GP_OK;
    // area_pixel_compute_scale<float>(in, out, align_corners=true): (in-1)/(out-1) in fp32, 0 for a single output pixel
    const float sh = out_h > 1 ? (float)(h - 1) / (float)(out_h - 1) : 0.f;
    const float sw = out_w > 1 ? (float)(w - 1) / (float)(out_w - 1) : 0.f;
    lift_dense_bilinear_accum_kernel<<<(int)((n_v * 64 + 255) / 256), 256, 0, gp_stream(stream_)>>>(feat, d, h, w, sh, sw, pt, x, y,
                                                                                                     n_v, sum, ld_sum, cnt);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" int gp_lift_dense_finish(float *sum, int64_t ld_sum, int32_t d, const float *cnt, int64_t n, uint8_t *seen,
                                    void *stream_) {
    GP_CHECK_ARG(sum && cnt && seen && n > 0, "gp_lift_dense_finish: null/empty argument");
    lift_dense_finish_kernel<<<(int)((n * 64 + 255) / 256), 256, 0, gp_stream(stream_)>>>(sum, ld_sum, d, cnt, n, seen);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" size_t gp_lift_masks_workspace_bytes(int32_t q, int32_t h, int32_t w) {
    return gp_align_up((size_t)q * h * w * sizeof(float), 256);
}

extern "C" int gp_lift_masks_view(const float *pred_masks, int32_t q, int32_t h, int32_t w, const float *scores,
                                  const int32_t *tap_x0, const float *tap_wx, const int32_t *tap_y0,
                                  const float *tap_wy, int32_t out_h, int32_t out_w, const int64_t *x,
                                  const int64_t *y, int64_t n_v, int32_t *seg, float *seg_logit, void *workspace,
                                  size_t workspace_bytes, void *stream_) {
    GP_CHECK_ARG(pred_masks && scores && tap_x0 && tap_wx && tap_y0 && tap_wy && x && y && seg && workspace,
                 "gp_lift_masks_view: null argument");
    GP_CHECK_ARG(q > 0 && h > 0 && w > 0, "gp_lift_masks_view: bad mask shape");
    if (workspace_bytes < gp_lift_masks_workspace_bytes(q, h, w)) { gp_set_error("gp_lift_masks_view: workspace too small"); return GP_ENOMEM; }
    if (n_v == 0) return GP_OK;
    hipStream_t s = gp_stream(stream_);
    float *mt = static_cast<float *>(workspace);
    int hw = h * w;
    dim3 tg((hw + 63) / 64, (q + 63) / 64);
    transpose_kernel<<<tg, 256, 0, s>>>(pred_masks, q, hw, mt);
    lift_masks_kernel<<<(int)((n_v * 64 + 255) / 256), 256, 0, s>>>(mt, q, h, w, scores, tap_x0, tap_wx, tap_y0, tap_wy,
                                                                    out_h, out_w, x, y, n_v, seg, seg_logit);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

// ---- all views of a scene at once (entries from gp_views_visible_lists)
static size_t lv_scan32_tmp(int64_t n) {
    size_t t = 0;
    (void)rocprim::exclusive_scan(nullptr, t, (int32_t *)nullptr, (int32_t *)nullptr, (int32_t)0, (size_t)n, rocprim::plus<int32_t>(), 0);
    return t;
}
static size_t lv_scan64_tmp(int64_t n) {
    size_t t = 0;
    (void)rocprim::exclusive_scan(nullptr, t, (int64_t *)nullptr, (int64_t *)nullptr, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), 0);
    return t;
}
struct LvWork {
    float *mt; int32_t *cov, *rs, *rent, *qent; float *rxyz; double *part_d; int32_t *part_i; unsigned long long *vmask; int4 *tab;
    int64_t *cnt; char *tmp; size_t tmp_bytes;
    int32_t *order; float *sscore;                       // the queries of every view by descending score
};
static size_t lv_carve(void *ws, size_t bytes, int32_t nsrc, int32_t q, int32_t h, int32_t w, int64_t total, int64_t n, int64_t fill_cap,
                       LvWork &k) {
    GpCarver cv(ws, bytes);
    k.mt = cv.take<float>((size_t)nsrc * q * h * w);
    k.order = cv.take<int32_t>((size_t)nsrc * q);
    k.sscore = cv.take<float>((size_t)nsrc * q);
    k.cov = cv.take<int32_t>(total + 1);
    k.rs = cv.take<int32_t>(total + 1);
    k.rent = cv.take<int32_t>(total);
    k.qent = cv.take<int32_t>(total);
    k.rxyz = cv.take<float>(total * 3);
    k.part_d = cv.take<double>((size_t)FV_CHUNKS * fill_cap);
    k.part_i = cv.take<int32_t>((size_t)FV_CHUNKS * fill_cap);
    k.vmask = cv.take<unsigned long long>(n * 2);
    k.cnt = cv.take<int64_t>(n + 1);
    k.tab = cv.take<int4>(130);
    size_t a = lv_scan32_tmp(total + 1), b = lv_scan64_tmp(n + 1);
    k.tmp_bytes = a > b ? a : b;
    k.tmp = cv.take<char>(k.tmp_bytes);
    return cv.off;
}
// fill_cap: capacity (in fill queries = uncovered entries) of the partial-result arrays of the in-view fill, 1 .. total; 0 = total
// (every entry could be a query).  Queries beyond the capacity are still answered (one block per 256 of them sweeps the whole
// reference range), so any value is CORRECT; a capacity of the expected query count keeps the fill fully parallel.
extern "C" size_t gp_lift_masks_views_workspace_bytes(int32_t nsrc, int32_t q, int32_t h, int32_t w, int64_t total, int64_t n,
                                                      int64_t fill_cap) {
    if (nsrc <= 0 || q <= 0 || h <= 0 || w <= 0 || total <= 0 || n <= 0 || fill_cap < 0 || fill_cap > total) return 0;
    LvWork k;
    return lv_carve(nullptr, 0, nsrc, q, h, w, total, n, fill_cap ? fill_cap : total, k);
}
// seg i32 [total] (out: segment of every entry after the in-view fill, -1 = none); pv_start i64 [n+1], pv_view / pv_seg i32 [total]
// (out: the point -> (view, segment) CSR that gp_fuse_views_top3 reads).  Replaces, for all views together, the per-view
// sequence gp_lift_masks_view -> gp_nn1_masked_f64 + gather -> gp_pv_count -> scan -> gp_pv_fill.
extern "C" int gp_lift_masks_views(const float *pred_masks, int32_t nsrc, int32_t q, int32_t h, int32_t w, const float *scores,
                                   const int32_t *tap_x0, const float *tap_wx, const int32_t *tap_y0, const float *tap_wy,
                                   int32_t out_h, int32_t out_w, const float *xyz, int64_t n, const int64_t *ent_pt,
                                   const int64_t *ent_x, const int64_t *ent_y, const int32_t *ent_view, const int64_t *view_off,
                                   const uint8_t *keep, int32_t nviews, int64_t total, int64_t fill_cap, int32_t *seg,
                                   int64_t *pv_start, int32_t *pv_view, int32_t *pv_seg, void *workspace, size_t workspace_bytes,
                                   void *stream_) {
    GP_CHECK_ARG(pred_masks && scores && tap_x0 && tap_wx && tap_y0 && tap_wy && xyz && ent_pt && ent_x && ent_y && ent_view && view_off &&
                     keep && seg && pv_start && pv_view && pv_seg && workspace,
                 "gp_lift_masks_views: null argument");
    GP_CHECK_ARG(q > 0 && h > 0 && w > 0 && n > 0 && total > 0, "gp_lift_masks_views: empty shape");
    GP_CHECK_ARG(q <= 1024, "gp_lift_masks_views: %d queries (the score sort holds 1024)", q);
    GP_CHECK_ARG(nviews > 0 && nviews <= 128 && nviews <= nsrc, "gp_lift_masks_views: %d views (1..128, <= %d mask sets)", nviews, nsrc);
    GP_CHECK_ARG(total < (int64_t)INT32_MAX, "gp_lift_masks_views: too many entries");
    GP_CHECK_ARG(fill_cap >= 0 && fill_cap <= total, "gp_lift_masks_views: fill_cap=%lld outside 0..total", (long long)fill_cap);
    if (fill_cap == 0) fill_cap = total;
    LvWork k;
    if (lv_carve(workspace, workspace_bytes, nsrc, q, h, w, total, n, fill_cap, k) > workspace_bytes) {
        gp_set_error("gp_lift_masks_views: workspace too small");
        return GP_ENOMEM;
    }
    hipStream_t s = gp_stream(stream_);
    const int hw = h * w;
    lv_sort_scores_kernel<<<nviews, 256, 0, s>>>(scores, q, k.order, k.sscore);
    transpose_views_kernel<<<dim3((hw + 63) / 64, (q + 63) / 64, nviews), 256, 0, s>>>(pred_masks, q, hw, k.mt, k.order);
    lift_masks_views_kernel<<<(unsigned)((total * 64 + 255) / 256), 256, 0, s>>>(k.mt, q, h, w, k.sscore, k.order, tap_x0, tap_wx, tap_y0,
                                                                               tap_wy, out_h, out_w, ent_x, ent_y, ent_view, keep, total, seg);
    // in-view fill
    const unsigned eb = (unsigned)((total + 1 + 255) / 256);
    fill_flags_kernel<<<eb, 256, 0, s>>>(seg, total, k.cov);
    size_t tb = k.tmp_bytes;
    GP_CHECK_HIP(rocprim::exclusive_scan(k.tmp, tb, k.cov, k.rs, (int32_t)0, (size_t)(total + 1), rocprim::plus<int32_t>(), s));
    fill_compact_kernel<<<eb, 256, 0, s>>>(xyz, ent_pt, seg, k.rs, total, k.rxyz, k.rent, k.qent);
    const unsigned qb = (unsigned)(total / 256 + nviews + 1);                           // >= sum over views of ceil(queries / 256)
    fill_table_kernel<<<1, 64, 0, s>>>(view_off, k.rs, nviews, k.tab);
    fill_part_kernel<<<dim3(qb, FV_CHUNKS), 256, 0, s>>>(xyz, ent_pt, k.tab, nviews, k.rxyz, k.qent, k.part_d, k.part_i, fill_cap,
                                                         k.rent, seg);
    fill_reduce_kernel<<<qb, 256, 0, s>>>(k.tab, nviews, k.part_d, k.part_i, fill_cap, k.rent, k.qent, seg);
    // point -> (view, segment) lists
    GP_CHECK_HIP(hipMemsetAsync(k.vmask, 0, (size_t)n * 2 * sizeof(unsigned long long), s));
    pv_mask_kernel<<<eb, 256, 0, s>>>(ent_pt, ent_view, keep, total, k.vmask);
    pv_cnt_kernel<<<(unsigned)((n + 1 + 255) / 256), 256, 0, s>>>(k.vmask, n, k.cnt);
    tb = k.tmp_bytes;
    GP_CHECK_HIP(rocprim::exclusive_scan(k.tmp, tb, k.cnt, pv_start, (int64_t)0, (size_t)(n + 1), rocprim::plus<int64_t>(), s));
    pv_fill_views_kernel<<<eb, 256, 0, s>>>(ent_pt, ent_view, keep, seg, total, k.vmask, pv_start, pv_view, pv_seg);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" int gp_segment_tables(const float *mask_embed, int32_t q, int32_t d, const float *text_norm, int32_t c,
                                 float logit_scale, float *f_seg, float *logit_seg, void *stream_) {
    GP_CHECK_ARG(mask_embed && text_norm && f_seg && logit_seg && q > 0 && d > 0 && c > 0, "gp_segment_tables: null/empty argument");
    segment_tables_kernel<<<(q * 64 + 255) / 256, 256, 0, gp_stream(stream_)>>>(mask_embed, q, d, text_norm, c, logit_scale,
                                                                              f_seg, logit_seg);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" int gp_pv_count(const int64_t *pt, int64_t n_v, int64_t *cnt, void *stream_) {
    GP_CHECK_ARG(pt && cnt, "gp_pv_count: null argument");
    if (n_v == 0) return GP_OK;
    pv_count_kernel<<<(int)((n_v + 255) / 256), 256, 0, gp_stream(stream_)>>>(pt, n_v, cnt);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" size_t gp_scan_workspace_bytes(int64_t n) {
    size_t t = 0;
    (void)rocprim::exclusive_scan(nullptr, t, (int64_t *)nullptr, (int64_t *)nullptr, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), 0);
    return gp_align_up(t, 256);
}

extern "C" int gp_exclusive_scan_i64(const int64_t *in, int64_t n, int64_t *out, void *workspace, size_t workspace_bytes,
                                     void *stream_) {
    GP_CHECK_ARG(in && out && n > 0, "gp_exclusive_scan_i64: null/empty argument");
    size_t t = workspace_bytes;
    GP_CHECK_HIP(rocprim::exclusive_scan(workspace, t, in, out, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), gp_stream(stream_)));
    return GP_OK;
}

extern "C" int gp_pv_fill(const int64_t *pt, const int32_t *seg, int64_t n_v, int32_t view, const int64_t *pv_start,
                          int32_t *cursor, int32_t *pv_view, int32_t *pv_seg, void *stream_) {
    GP_CHECK_ARG(pt && seg && pv_start && cursor && pv_view && pv_seg, "gp_pv_fill: null argument");
    if (n_v == 0) return GP_OK;
    pv_fill_kernel<<<(int)((n_v + 255) / 256), 256, 0, gp_stream(stream_)>>>(pt, seg, n_v, view, pv_start, cursor, pv_view, pv_seg);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" int gp_fuse_views_top3(const int64_t *pv_start, const int32_t *pv_view, const int32_t *pv_seg, int64_t n,
                                  const float *f_seg, const float *logit_seg, int32_t q, int32_t d, int32_t c,
                                  float *out, int64_t ld_out, uint8_t *seen, void *stream_) {
    GP_CHECK_ARG(pv_start && pv_view && pv_seg && f_seg && logit_seg && out && n > 0, "gp_fuse_views_top3: null/empty argument");
    GP_CHECK_ARG(d % 4 == 0 && ld_out % 4 == 0, "gp_fuse_views_top3: d and ld_out must be multiples of 4");
    fuse_top3_kernel<<<(int)((n * 64 + 255) / 256), 256, 0, gp_stream(stream_)>>>(pv_start, pv_view, pv_seg, n, f_seg,
                                                                                logit_seg, q, d, c, out, ld_out, seen);
    GP_CHECK_LAUNCH();
    return GP_OK;
}
