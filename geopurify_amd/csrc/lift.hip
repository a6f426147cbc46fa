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
__global__ void lift_masks_kernel(const float *__restrict__ mt /*[h*w, Q]*/, int Q, int h, int w,
                                  const float *__restrict__ scores, const int32_t *__restrict__ tx0,
                                  const float *__restrict__ twx, const int32_t *__restrict__ ty0,
                                  const float *__restrict__ twy, int out_h, int out_w,
                                  const int64_t *__restrict__ px, const int64_t *__restrict__ py, int64_t n_v,
                                  int32_t *__restrict__ seg, float *__restrict__ seg_logit) {
    int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    if (i >= n_v) return;
    int lane = gp_lane();
    int row = (int)px[i], col = (int)py[i];                     // x_label = pixel row, y_label = pixel col
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
    if (lane == 0) {
        float sg = 1.f / (1.f + expf(-best_logit));
        seg[i] = (best_q >= 0 && sg >= 0.5f) ? best_q : -1;
        if (seg_logit) seg_logit[i] = best_logit;
    }
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
    // consensus class: argmax_c of (sum_v logits_v[c]) / M, first maximum
    float bestv = -INFINITY;
    int bestc = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
        float s = 0.f;
        for (int64_t j = b; j < e; ++j) { int sg = pv_seg[j]; s += sg >= 0 ? lseg[((int64_t)pv_view[j] * Q + sg) * C + c] : 0.f; }
        s = s / (float)M;
        if (s > bestv) { bestv = s; bestc = c; }
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
    for (int64_t j = b; j < e; ++j) {
        int sg = pv_seg[j];
        float s = sg >= 0 ? lseg[((int64_t)pv_view[j] * Q + sg) * C + bestc] : 0.f;
        if (s > s0) { s2 = s1; j2 = j1; s1 = s0; j1 = j0; s0 = s; j0 = j; }
        else if (s > s1) { s2 = s1; j2 = j1; s1 = s; j1 = j; }
        else if (s > s2) { s2 = s; j2 = j; }
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
