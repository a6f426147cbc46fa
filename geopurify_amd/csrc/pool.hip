// Rows 8, 11, 12 (+ final gather, row L2 norm): point->voxel mean, cosine-affinity softmax,
// affinity pooling in ELL form.  All HBM/L2-bandwidth-bound gathers: one wave per destination row,
// 16 B per lane coalesced row segments, wave-uniform neighbour indices and weights in SGPRs.
#include "gp_common.h"

extern int g_gp_knobs[16];

namespace {

// ------------------------------------------------------------------------------------------------
// scatter_mean in CSR form.  One wave per voxel; lanes stride the feature columns in float4 (or
// scalar for the tail).  Sum order = ascending point id (the order of `order` inside a segment),
// i.e. the order of a sequential index_add_, then one correctly rounded division by the count.
__global__ void scatter_mean_csr_kernel(const float *__restrict__ src, int64_t ld_src, int d,
                                        const int64_t *__restrict__ order, const int64_t *__restrict__ seg,
                                        int64_t nv, const int32_t *__restrict__ row_map,
                                        float *__restrict__ out, int64_t ld_out, int col0) {
    int64_t v = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    if (v >= nv) return;
    v = __builtin_amdgcn_readfirstlane((int)v);
    int lane = gp_lane();
    int64_t b = seg[v], e = seg[v + 1];
    int64_t orow = row_map ? row_map[v] : v;
    float inv_cnt_den = (float)(e - b < 1 ? 1 : e - b);
    for (int c = lane; c < d; c += 64) {
        float acc = 0.f;
        for (int64_t j = b; j < e; ++j) acc += src[order[j] * ld_src + c];
        out[orow * ld_out + col0 + c] = acc / inv_cnt_den;
    }
}

// vectorised variant for d % 4 == 0, 16-byte aligned rows
__global__ void scatter_mean_csr_v4_kernel(const float *__restrict__ src, int64_t ld_src, int d,
                                           const int64_t *__restrict__ order, const int64_t *__restrict__ seg,
                                           int64_t nv, const int32_t *__restrict__ row_map,
                                           float *__restrict__ out, int64_t ld_out, int col0) {
    int64_t v = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    if (v >= nv) return;
    v = __builtin_amdgcn_readfirstlane((int)v);
    int lane = gp_lane();
    int64_t b = seg[v], e = seg[v + 1];
    int64_t orow = row_map ? row_map[v] : v;
    float den = (float)(e - b < 1 ? 1 : e - b);
    for (int c = lane * 4; c < d; c += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int64_t j = b; j < e; ++j) {
            float4 t = *reinterpret_cast<const float4 *>(src + order[j] * ld_src + c);
            acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
        }
        acc.x /= den; acc.y /= den; acc.z /= den; acc.w /= den;
        *reinterpret_cast<float4 *>(out + orow * ld_out + col0 + c) = acc;
    }
}

// ------------------------------------------------------------------------------------------------
__global__ void gather_rows_kernel(const float *__restrict__ src, int64_t ld_src, int d,
                                   const int64_t *__restrict__ index, int64_t n,
                                   const int32_t *__restrict__ row_map, float *__restrict__ out, int64_t ld_out) {
    int64_t p = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    if (p >= n) return;
    int lane = gp_lane();
    int64_t r = index[p];
    if (row_map) r = row_map[r];
    if ((d & 3) == 0 && (ld_src & 3) == 0 && (ld_out & 3) == 0) {
        for (int c = lane * 4; c < d; c += 256)
            *reinterpret_cast<float4 *>(out + p * ld_out + c) = *reinterpret_cast<const float4 *>(src + r * ld_src + c);
    } else {
        for (int c = lane; c < d; c += 64) out[p * ld_out + c] = src[r * ld_src + c];
    }
}

// ------------------------------------------------------------------------------------------------
__global__ void l2norm_rows_kernel(float *__restrict__ x, int64_t ld, int d, int64_t n) {
    int64_t r = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    if (r >= n) return;
    int lane = gp_lane();
    float ss = 0.f;
    for (int c = lane; c < d; c += 64) { float v = x[r * ld + c]; ss += v * v; }
    ss = gp_wave_sum(ss);
    float nrm = fmaxf(sqrtf(ss), 1e-12f);
    for (int c = lane; c < d; c += 64) x[r * ld + c] = x[r * ld + c] / nrm;
}

// ------------------------------------------------------------------------------------------------
// affinity: one wave per voxel row.  4 lanes share a neighbour (each reads d/4 contiguous floats),
// 16 neighbours per pass; dot products reduced with two shuffles; softmax over K with wave reductions.
// SCATTER: each weight also goes, x GP_POOL_CS_WSCALE and split hi + lo, to element dst[i * k + j] of the pooling operator's
// fragment arrays (gp_pool_cs_structure) -- the same value gp_pool_cs_fill would read back from w, so the operator keeps its bits.
__device__ __forceinline__ void affinity_emit_fragment(float wgt, int32_t idx, _Float16 *__restrict__ wa_hi, _Float16 *__restrict__ wa_lo) {
    const float v = wgt * GP_POOL_CS_WSCALE;
    const _Float16 h = (_Float16)v;
    wa_hi[idx] = h;
    wa_lo[idx] = (_Float16)(v - (float)h);
}
template <int D, bool SCATTER>
__global__ void affinity_softmax_kernel(const float *__restrict__ e, int64_t ld_e, const int32_t *__restrict__ nbr,
                                        int k, int64_t nv, float sharpen, float *__restrict__ w, const int32_t *__restrict__ dst,
                                        _Float16 *__restrict__ wa_hi, _Float16 *__restrict__ wa_lo) {
    int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    if (i >= nv) return;
    int lane = gp_lane();
    int sub = lane & 3, grp = lane >> 2;          // 16 groups of 4 lanes
    constexpr int SEG = D / 4;                     // floats per lane
    float ei[SEG];
    const float *erow = e + i * ld_e + sub * SEG;
#pragma unroll
    for (int c = 0; c < SEG; c += 4) {
        float4 t = *reinterpret_cast<const float4 *>(erow + c);
        ei[c] = t.x; ei[c + 1] = t.y; ei[c + 2] = t.z; ei[c + 3] = t.w;
    }
    // logits for neighbours j = pass*16 + grp, kept by lane `sub==0` of each group; K <= 128 -> 8 passes
    float logit[8];
    float mx = -INFINITY;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        int j = p * 16 + grp;
        float dot = 0.f;
        if (j < k) {
            int64_t r = nbr[i * k + j];
            const float *nrow = e + r * ld_e + sub * SEG;
#pragma unroll
            for (int c = 0; c < SEG; c += 4) {
                float4 t = *reinterpret_cast<const float4 *>(nrow + c);
                dot += ei[c] * t.x + ei[c + 1] * t.y + ei[c + 2] * t.z + ei[c + 3] * t.w;
            }
        }
        dot += __shfl_xor(dot, 1, 64);
        dot += __shfl_xor(dot, 2, 64);
        logit[p] = (j < k) ? dot * sharpen : -INFINITY;
        mx = fmaxf(mx, logit[p]);
    }
    mx = gp_wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        logit[p] = (logit[p] == -INFINITY) ? 0.f : expf(logit[p] - mx);
        if (sub == 0) sum += logit[p];
    }
    sum = gp_wave_sum(sum);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        int j = p * 16 + grp;
        if (sub == 0 && j < k) {
            const float wgt = logit[p] / sum;
            w[i * k + j] = wgt;
            if constexpr (SCATTER) affinity_emit_fragment(wgt, dst[i * k + j], wa_hi, wa_lo);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// pooling v1: one wave per (row, 256-column slab).  The neighbour row index and weight are wave
// uniform (scalar loads); each lane moves 16 B of every neighbour row -> 1 KiB coalesced per request.
template <int UNROLL>
__global__ void __launch_bounds__(256) pool_ell_kernel(const float *__restrict__ x, int64_t ld_x,
                                                       const int32_t *__restrict__ nbr, const float *__restrict__ w,
                                                       int k, int64_t nv, int d, float *__restrict__ y, int64_t ld_y,
                                                       int slabs) {
    int64_t wid = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    int64_t row = wid / slabs;
    int slab = (int)(wid - row * slabs);
    if (row >= nv) return;
    row = __builtin_amdgcn_readfirstlane((int)row);
    slab = __builtin_amdgcn_readfirstlane(slab);
    int lane = gp_lane();
    int c = slab * 256 + lane * 4;
    bool act = c < d;
    const int32_t *nb = nbr + row * k;
    const float *wr = w + row * k;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int j = 0;
    for (; j + UNROLL <= k; j += UNROLL) {
        float4 t[UNROLL];
        float ww[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            int64_t r = nb[j + u];
            ww[u] = wr[j + u];
            t[u] = act ? *reinterpret_cast<const float4 *>(x + r * ld_x + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            acc.x = fmaf(ww[u], t[u].x, acc.x);
            acc.y = fmaf(ww[u], t[u].y, acc.y);
            acc.z = fmaf(ww[u], t[u].z, acc.z);
            acc.w = fmaf(ww[u], t[u].w, acc.w);
        }
    }
    for (; j < k; ++j) {
        int64_t r = nb[j];
        float wj = wr[j];
        if (act) {
            float4 t = *reinterpret_cast<const float4 *>(x + r * ld_x + c);
            acc.x = fmaf(wj, t.x, acc.x); acc.y = fmaf(wj, t.y, acc.y);
            acc.z = fmaf(wj, t.z, acc.z); acc.w = fmaf(wj, t.w, acc.w);
        }
    }
    if (act) *reinterpret_cast<float4 *>(y + row * ld_y + c) = acc;
}

// affinity, block form (D = 128, K <= 128): one 1024-thread workgroup owns 16 consecutive rows (Morton-adjacent when the
// caller orders the voxels so), one wave per row.  The 16 x K neighbour ids are de-duplicated through an LDS hash table
// (~13 distinct rows per row instead of 96), the distinct embedding rows are loaded ONCE into LDS and every dot product
// reads its neighbour row from there: 7x less traffic through L2 than the one-wave-per-row kernel above, same arithmetic
// order per row (same 4-lane split of the 128 floats, same shuffles, same softmax) => bit-identical weights.
// Rows of the union beyond the LDS capacity are read from global memory as before.
// LDS row layout: the 32-float segment of lane `sub` starts at float sub * 36 and a row is 144 floats, so the four lanes of a
// neighbour group read four different bank quads (offsets 0, 36, 8, 44 mod 64) and rows idx, idx+1, idx+2, idx+3 fill the rest.
constexpr int AB_D = 128, AB_SEG_LD = 36, AB_LD = 144;
template <int R> struct AbGeo {                               // R rows per workgroup (R waves): 16 -> one workgroup per CU, 8 -> two
    static constexpr int HS = R * 128, CAP = R == 16 ? 240 : 120;
    static constexpr size_t SMEM = (size_t)CAP * AB_LD * 4 + HS * 4 + HS * 2 + HS * 4;
};
template <int R, bool SCATTER>
__global__ void __launch_bounds__(R * 64)
affinity_block_kernel(const float *__restrict__ e, int64_t ld_e, const int32_t *__restrict__ nbr, int k, int64_t nv, float sharpen,
                      float *__restrict__ w, const int32_t *__restrict__ dst, _Float16 *__restrict__ wa_hi, _Float16 *__restrict__ wa_lo) {
    constexpr int AB_ROWS = R, AB_HS = AbGeo<R>::HS, AB_CAP = AbGeo<R>::CAP, NT = R * 64;
    extern __shared__ __align__(16) unsigned char ab_smem[];
    float *rows = reinterpret_cast<float *>(ab_smem);                                   // [AB_CAP][AB_LD]
    int *keys = reinterpret_cast<int *>(ab_smem + (size_t)AB_CAP * AB_LD * 4);           // [AB_HS] voxel id or -1
    short *index = reinterpret_cast<short *>(keys + AB_HS);                            // [AB_HS] compact row index of a slot
    int *ulist = reinterpret_cast<int *>(index + AB_HS);                               // [AB_HS] voxel id of compact index
    __shared__ int s_wcnt[R];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t i = blockIdx.x * (int64_t)AB_ROWS + wv;
    const bool live = i < nv;
    int id0 = -1, id1 = -1;                                             // this row's neighbour ids: lane j and j + 64
    if (live) {
        if (lane < k) id0 = nbr[i * k + lane];
        if (lane + 64 < k) id1 = nbr[i * k + lane + 64];
    }
    for (int t = tid; t < AB_HS; t += NT) keys[t] = -1;
    __syncthreads();
    // ---- phase 1: insert this row's neighbour ids (lane j and j + 64), remember their slots
    int slot0 = 0, slot1 = 0;
    auto insert = [&](int id) {
        unsigned h = ((unsigned)id * 2654435761u) >> (R == 16 ? 21 : 22);    // log2(AB_HS) bits
        while (true) {
            const int old = atomicCAS(&keys[h], -1, id);
            if (old == -1 || old == id) break;
            h = (h + 1) & (AB_HS - 1);
        }
        return (int)h;
    };
    if (id0 >= 0) slot0 = insert(id0);
    if (id1 >= 0) slot1 = insert(id1);
    __syncthreads();
    // ---- phase 2: compact the occupied slots (two per thread, in slot order)
    {
        const int k0 = keys[2 * tid], k1 = keys[2 * tid + 1];
        const int c = (k0 >= 0) + (k1 >= 0);
        int incl = c;                                                    // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        if (lane == 63) s_wcnt[wv] = incl;
        __syncthreads();
        int base = 0;
        for (int q = 0; q < wv; ++q) base += s_wcnt[q];
        int idx = base + incl - c;
        index[2 * tid] = (short)(k0 >= 0 ? idx : -1);
        if (k0 >= 0) { ulist[idx] = k0; ++idx; }
        index[2 * tid + 1] = (short)(k1 >= 0 ? idx : -1);
        if (k1 >= 0) ulist[idx] = k1;
    }
    __syncthreads();
    int U = 0;
    for (int q = 0; q < R; ++q) U += s_wcnt[q];
    const int Uc = U < AB_CAP ? U : AB_CAP;
    // ---- phase 3: the distinct rows, once (a wave moves two 512-byte rows per instruction; all loads in flight together)
    {
        constexpr int NI = (AB_CAP + 2 * R - 1) / (2 * R);
        const int q4 = lane & 31;
        float4 t[NI];
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const int u = wv * 2 + (lane >> 5) + q * 2 * R;
            t[q] = *reinterpret_cast<const float4 *>(e + (int64_t)ulist[u < Uc ? u : 0] * ld_e + q4 * 4);   // U >= 1
        }
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const int u = wv * 2 + (lane >> 5) + q * 2 * R;
            if (u < Uc) *reinterpret_cast<float4 *>(rows + (size_t)u * AB_LD + (q4 >> 3) * AB_SEG_LD + (q4 & 7) * 4) = t[q];
        }
    }
    __syncthreads();
    if (!live) return;
    // ---- phase 4: the row's K dot products and its softmax, exactly as affinity_softmax_kernel<128>
    const int sub = lane & 3, grp = lane >> 2;
    constexpr int SEG = AB_D / 4;
    float ei[SEG];
    const float *erow = e + i * ld_e + sub * SEG;
#pragma unroll
    for (int c = 0; c < SEG; c += 4) {
        float4 t = *reinterpret_cast<const float4 *>(erow + c);
        ei[c] = t.x; ei[c + 1] = t.y; ei[c + 2] = t.z; ei[c + 3] = t.w;
    }
    float logit[8];
    float mx = -INFINITY;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int j = p * 16 + grp;
        const int sl = __shfl(j < 64 ? slot0 : slot1, j & 63, 64);
        float dot = 0.f;
        if (j < k) {
            const int idx = index[sl];
            if (idx < AB_CAP) {                                          // two loops: LDS and global address spaces
                const float *nrow = rows + (size_t)idx * AB_LD + sub * AB_SEG_LD;
#pragma unroll
                for (int c = 0; c < SEG; c += 4) {
                    float4 t = *reinterpret_cast<const float4 *>(nrow + c);
                    dot += ei[c] * t.x + ei[c + 1] * t.y + ei[c + 2] * t.z + ei[c + 3] * t.w;
                }
            } else {
                const float *nrow = e + (int64_t)keys[sl] * ld_e + sub * SEG;
#pragma unroll
                for (int c = 0; c < SEG; c += 4) {
                    float4 t = *reinterpret_cast<const float4 *>(nrow + c);
                    dot += ei[c] * t.x + ei[c + 1] * t.y + ei[c + 2] * t.z + ei[c + 3] * t.w;
                }
            }
        }
        dot += __shfl_xor(dot, 1, 64);
        dot += __shfl_xor(dot, 2, 64);
        logit[p] = (j < k) ? dot * sharpen : -INFINITY;
        mx = fmaxf(mx, logit[p]);
    }
    mx = gp_wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        logit[p] = (logit[p] == -INFINITY) ? 0.f : expf(logit[p] - mx);
        if (sub == 0) sum += logit[p];
    }
    sum = gp_wave_sum(sum);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int j = p * 16 + grp;
        if (sub == 0 && j < k) {
            const float wgt = logit[p] / sum;
            w[i * k + j] = wgt;
            if constexpr (SCATTER) affinity_emit_fragment(wgt, dst[i * k + j], wa_hi, wa_lo);
        }
    }
}

}  // namespace

// ================================================================================================
extern "C" int gp_scatter_mean_csr(const float *src, int64_t ld_src, int32_t d, const int64_t *order,
                                   const int64_t *seg_start, int64_t nv, const int32_t *row_map, float *out,
                                   int64_t ld_out, int32_t col0, void *stream_) {
    GP_CHECK_ARG(src && order && seg_start && out && nv > 0 && d > 0, "gp_scatter_mean_csr: null/empty argument");
    hipStream_t s = gp_stream(stream_);
    int64_t threads = nv * 64;
    int blocks = (int)((threads + 255) / 256);
    bool v4 = (d % 4 == 0) && (ld_src % 4 == 0) && (ld_out % 4 == 0) && (col0 % 4 == 0) &&
              ((uintptr_t)src % 16 == 0) && ((uintptr_t)out % 16 == 0);
    if (v4)
        scatter_mean_csr_v4_kernel<<<blocks, 256, 0, s>>>(src, ld_src, d, order, seg_start, nv, row_map, out, ld_out, col0);
    else
        scatter_mean_csr_kernel<<<blocks, 256, 0, s>>>(src, ld_src, d, order, seg_start, nv, row_map, out, ld_out, col0);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" int gp_gather_rows(const float *src, int64_t ld_src, int32_t d, const int64_t *index, int64_t n,
                              const int32_t *row_map, float *out, int64_t ld_out, void *stream_) {
    GP_CHECK_ARG(src && index && out && n > 0 && d > 0, "gp_gather_rows: null/empty argument");
    int blocks = (int)((n * 64 + 255) / 256);
    gather_rows_kernel<<<blocks, 256, 0, gp_stream(stream_)>>>(src, ld_src, d, index, n, row_map, out, ld_out);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" int gp_l2norm_rows(float *x, int64_t ld, int32_t d, int64_t n, void *stream_) {
    GP_CHECK_ARG(x && n > 0 && d > 0, "gp_l2norm_rows: null/empty argument");
    int blocks = (int)((n * 64 + 255) / 256);
    l2norm_rows_kernel<<<blocks, 256, 0, gp_stream(stream_)>>>(x, ld, d, n);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

// dst / wa_hi / wa_lo all NULL: the weights only.  All set: every weight also lands in the pooling operator's fragment arrays.
template <bool SCATTER>
static int affinity_launch(const float *e, int64_t ld_e, int32_t d, const int32_t *nbr, int32_t k, int64_t nv, float sharpen, float *w,
                           const int32_t *dst, _Float16 *wa_hi, _Float16 *wa_lo, hipStream_t s) {
    int blocks = (int)((nv * 64 + 255) / 256);
    if (d == 128 && g_gp_knobs[15] != 1) {              // block form: distinct neighbour rows of R rows staged once in LDS
        GP_SMEM_ATTR((affinity_block_kernel<16, SCATTER>), AbGeo<16>::SMEM);
        GP_SMEM_ATTR((affinity_block_kernel<8, SCATTER>), AbGeo<8>::SMEM);
        if (g_gp_knobs[15] == 2)                        // measured on config S: 16 rows 0.49 ms, 8 rows 0.55 ms, wave form 0.81 ms
            affinity_block_kernel<8, SCATTER><<<(unsigned)((nv + 7) / 8), 512, AbGeo<8>::SMEM, s>>>(e, ld_e, nbr, k, nv, sharpen, w, dst, wa_hi, wa_lo);
        else
            affinity_block_kernel<16, SCATTER><<<(unsigned)((nv + 15) / 16), 1024, AbGeo<16>::SMEM, s>>>(e, ld_e, nbr, k, nv, sharpen, w, dst, wa_hi, wa_lo);
        GP_CHECK_LAUNCH();
        return GP_OK;
    }
    switch (d) {
        case 128: affinity_softmax_kernel<128, SCATTER><<<blocks, 256, 0, s>>>(e, ld_e, nbr, k, nv, sharpen, w, dst, wa_hi, wa_lo); break;
        case 64: affinity_softmax_kernel<64, SCATTER><<<blocks, 256, 0, s>>>(e, ld_e, nbr, k, nv, sharpen, w, dst, wa_hi, wa_lo); break;
        case 32: affinity_softmax_kernel<32, SCATTER><<<blocks, 256, 0, s>>>(e, ld_e, nbr, k, nv, sharpen, w, dst, wa_hi, wa_lo); break;
        case 16: affinity_softmax_kernel<16, SCATTER><<<blocks, 256, 0, s>>>(e, ld_e, nbr, k, nv, sharpen, w, dst, wa_hi, wa_lo); break;
        default: gp_set_error("gp_affinity_softmax: embedding dim %d not in {16,32,64,128}", d); return GP_EINVAL;
    }
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" int gp_affinity_softmax(const float *e, int64_t ld_e, int32_t d, const int32_t *nbr, int32_t k,
                                   int64_t nv, float sharpen, float *w, void *stream_) {
    GP_CHECK_ARG(e && nbr && w && nv > 0, "gp_affinity_softmax: null/empty argument");
    GP_CHECK_ARG(k > 0 && k <= 128, "gp_affinity_softmax: k=%d not in 1..128", k);
    GP_CHECK_ARG(ld_e % 4 == 0 && (uintptr_t)e % 16 == 0, "gp_affinity_softmax: rows must be 16-byte aligned");
    return affinity_launch<false>(e, ld_e, d, nbr, k, nv, sharpen, w, nullptr, nullptr, nullptr, gp_stream(stream_));
}

// The same weights, written to w AND -- x 2^10, split hi + lo -- to element dst[row * k + j] of the pooling operator's fragment arrays
// (dst, wa_hi, wa_lo from gp_pool_cs_structure): the operator is complete when this returns, no gp_pool_cs_fill pass.
extern "C" int gp_affinity_softmax_scatter(const float *e, int64_t ld_e, int32_t d, const int32_t *nbr, int32_t k, int64_t nv,
                                           float sharpen, float *w, const int32_t *dst, void *wa_hi, void *wa_lo, void *stream_) {
    GP_CHECK_ARG(e && nbr && w && dst && wa_hi && wa_lo && nv > 0, "gp_affinity_softmax_scatter: null/empty argument");
    GP_CHECK_ARG(k > 0 && k <= 128, "gp_affinity_softmax_scatter: k=%d not in 1..128", k);
    GP_CHECK_ARG(ld_e % 4 == 0 && (uintptr_t)e % 16 == 0, "gp_affinity_softmax_scatter: rows must be 16-byte aligned");
    return affinity_launch<true>(e, ld_e, d, nbr, k, nv, sharpen, w, dst, static_cast<_Float16 *>(wa_hi), static_cast<_Float16 *>(wa_lo),
                                 gp_stream(stream_));
}

extern "C" int gp_pool_ell(const float *x, int64_t ld_x, const int32_t *nbr, const float *w, int32_t k, int64_t nv,
                           int32_t d, float *y, int64_t ld_y, void *stream_) {
    GP_CHECK_ARG(x && nbr && w && y && nv > 0 && k > 0, "gp_pool_ell: null/empty argument");
    GP_CHECK_ARG(d > 0 && d % 4 == 0 && ld_x % 4 == 0 && ld_y % 4 == 0, "gp_pool_ell: d/ld must be multiples of 4");
    GP_CHECK_ARG((uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0, "gp_pool_ell: x/y must be 16-byte aligned");
    GP_CHECK_ARG(x != y, "gp_pool_ell: x and y must not alias");
    int slabs = (d + 255) / 256;
    int64_t waves = nv * slabs;
    int blocks = (int)((waves * 64 + 255) / 256);
    pool_ell_kernel<8><<<blocks, 256, 0, gp_stream(stream_)>>>(x, ld_x, nbr, w, k, nv, d, y, ld_y, slabs);
    GP_CHECK_LAUNCH();
    return GP_OK;
}
