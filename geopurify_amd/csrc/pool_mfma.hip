// Row 12, matrix-core variant of the affinity pooling (the metric's roofline stage).
//
// The tiled VALU kernels (pool_tiles.hip) are co-limited by L2->CU gather bandwidth and by the fp32 FMA
// rate (zero-padded tiles cost 1.8x the useful FMAs).  Here a 256-thread workgroup owns 64 Morton-adjacent
// rows (4 waves x 16 rows) and sweeps the block's neighbour union (6.4 union rows per output row instead of
// 96 neighbour rows) in steps of 32 union rows:
//   * the 32 rows x 512 columns of a step are staged into LDS by global_load_lds from PRE-SPLIT operands
//     (x = hi + lo, two f16 planes written by the previous application's epilogue), whole 1-KiB row
//     segments, double buffered;
//   * each wave multiplies its dense 16 x 32 weight block (pre-split f16, stored in MFMA A-fragment
//     order, one 16-byte load per lane) with the staged rows on v_mfma_f32_16x16x32_f16, the B
//     fragments read column-major from the row-major image by ds_read_b64_tr_b16 (hardware transpose;
//     the image is XOR-swizzled through the DMA source addresses so that the reads are conflict-free);
//   * hi*hi + hi*lo + lo*hi with fp32 accumulation = fp32-class accuracy (same scheme as the sparse
//     convolution; the dropped lo*lo term is 2^-22 relative).
// The matrix cores make the zero padding free; what remains is the block-union traffic: ~1.8 GB through
// L2 and ~1 GB to/from HBM per application at Nv = 134k.
#include <cstring>
#include <rocprim/device/device_scan.hpp>

#include "gp_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((vector_size(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int PM_ROWS = 64;        // output rows per workgroup
constexpr int PM_W = 4;            // waves per workgroup (16 rows each)
constexpr int PM_KS = 32;          // union rows per step (MFMA K)
constexpr int PM_D = 512;          // columns
constexpr int PM_MAXID = 8192;     // ids sorted per block in the builder (64 rows x K <= 8192)
constexpr int PM_EP = PM_D + 4;    // epilogue staging pitch (floats)
constexpr float PM_WSCALE = 1024.f;   // weights (<= 1) are stored x 2^10 so that their f16 lo parts stay normal

__device__ __forceinline__ void glds16(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}
// 16-byte chunk swizzle of a staged row: physical chunk = logical chunk ^ swz(row)
__device__ __forceinline__ int pm_swz(int r) { return 2 * ((r & 3) | (((r >> 3) & 1) << 2)); }

// ------------------------------------------------------------------------------------------------ builder
__device__ __forceinline__ void bitonic_sort_lds(int *a, int n_pow2, int tid, int nthreads) {
    for (int k = 2; k <= n_pow2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n_pow2; i += nthreads) {
                int ixj = i ^ j;
                if (ixj > i) {
                    int x = a[i], y = a[ixj];
                    bool up = (i & k) == 0;
                    if ((x > y) == up) { a[i] = y; a[ixj] = x; }
                }
            }
            __syncthreads();
        }
}

// sorted unique union of the neighbour ids of rows [b*64, b*64+64).  count pass: bu_n / padded count;
// fill pass: bu_row (padding repeats the first id; its weights stay zero).
__global__ void __launch_bounds__(512)
pm_union_kernel(const int32_t *__restrict__ nbr, int64_t nv, int k, int64_t *__restrict__ padded_cnt,
                int32_t *__restrict__ bu_n, const int64_t *__restrict__ bu_off, int32_t *__restrict__ bu_row) {
    __shared__ int s_ids[PM_MAXID];
    __shared__ int s_wcnt[8];
    __shared__ int s_base;
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t r0 = b * PM_ROWS;
    const int rows = (int)((nv - r0) < PM_ROWS ? (nv - r0) : PM_ROWS);
    const int n = rows * k;
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (int i = tid; i < np2; i += 512) s_ids[i] = i < n ? nbr[r0 * k + i] : INT32_MAX;
    __syncthreads();
    bitonic_sort_lds(s_ids, np2, tid, 512);
    if (tid == 0) s_base = 0;
    __syncthreads();
    const int64_t o = bu_row ? bu_off[b] : 0;
    for (int i0 = 0; i0 < n; i0 += 512) {
        int i = i0 + tid;
        int head = (i < n) && (i == 0 || s_ids[i] != s_ids[i - 1]);
        unsigned long long m = __ballot(head);
        if (lane == 0) s_wcnt[wv] = __popcll(m);
        __syncthreads();
        int before = s_base;
        for (int w = 0; w < wv; ++w) before += s_wcnt[w];
        int r = before + __popcll(m & ((1ull << lane) - 1ull));
        if (head && bu_row) bu_row[o + r] = s_ids[i];
        __syncthreads();
        if (tid == 0) { int tot = 0; for (int w = 0; w < 8; ++w) tot += s_wcnt[w]; s_base += tot; }
        __syncthreads();
    }
    const int U = s_base, Up = (U + PM_KS - 1) / PM_KS * PM_KS;
    if (!bu_row) {
        if (tid == 0) { padded_cnt[b] = Up; bu_n[b] = U; }
    } else {
        for (int i = U + tid; i < Up; i += 512) bu_row[o + i] = s_ids[0];
    }
}

// scatter the ELL weights into MFMA A-fragment order: wa[(kstep*4 + wave)*64 + lane][8], lane = (k>>3)*16 + m
__global__ void pm_weights_kernel(const int32_t *__restrict__ nbr, const float *__restrict__ w, int64_t nv, int k,
                                  const int64_t *__restrict__ bu_off, const int32_t *__restrict__ bu_n,
                                  const int32_t *__restrict__ bu_row, _Float16 *__restrict__ wa_hi, _Float16 *__restrict__ wa_lo) {
    int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= nv * k) return;
    int64_t r = e / k;
    int64_t b = r / PM_ROWS;
    int wv = (int)((r % PM_ROWS) / 16), m = (int)(r % 16);
    int id = nbr[e];
    const int32_t *u = bu_row + bu_off[b];
    int lo = 0, hi = bu_n[b] - 1;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (u[mid] < id) lo = mid + 1; else hi = mid; }
    int64_t ks = bu_off[b] / PM_KS + lo / PM_KS;
    int kk = lo % PM_KS;
    int64_t idx = ((ks * PM_W + wv) * 64 + (kk >> 3) * 16 + m) * 8 + (kk & 7);
    float v = w[e] * PM_WSCALE;
    _Float16 h = (_Float16)v;
    wa_hi[idx] = h;
    wa_lo[idx] = (_Float16)(v - (float)h);
}

// ------------------------------------------------------------------------------------------------ apply
struct PmSmem {
    _Float16 xh[2][PM_KS][PM_D];     // 2 x 32 KiB
    _Float16 xl[2][PM_KS][PM_D];     // 2 x 32 KiB
};

constexpr size_t PM_EPI_BYTES = (size_t)PM_W * 16 * PM_EP * sizeof(float);
constexpr size_t PM_SMEM_BYTES = sizeof(PmSmem) > PM_EPI_BYTES ? sizeof(PmSmem) : PM_EPI_BYTES;

__device__ __forceinline__ f16x8 pm_tr8(const _Float16 *p0, const _Float16 *p1) {
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p0);
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p1);
    f16x8 r;
    __builtin_memcpy(&r, &a, 8);
    __builtin_memcpy(reinterpret_cast<char *>(&r) + 8, &b, 8);
    return r;
}

__global__ void __launch_bounds__(256)
pool_mfma_kernel(const _Float16 *__restrict__ x_hi, const _Float16 *__restrict__ x_lo, int64_t ld_x,
                 const int64_t *__restrict__ bu_off, const int32_t *__restrict__ bu_row,
                 const _Float16 *__restrict__ wa_hi, const _Float16 *__restrict__ wa_lo, int64_t nv, int64_t nblocks,
                 _Float16 *__restrict__ y_hi, _Float16 *__restrict__ y_lo, int64_t ld_y, float *__restrict__ y_f32,
                 int64_t ld_yf, int64_t per_xcd) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    PmSmem &sm = *reinterpret_cast<PmSmem *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t b = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);     // XCD-contiguous block order
    if (b >= nblocks) return;
    const int64_t ub0 = bu_off[b];
    const int nsteps = (int)((bu_off[b + 1] - ub0) / PM_KS);
    const int64_t ks0 = ub0 / PM_KS;

    // DMA staging: wave wv moves rows 8wv .. 8wv+7 of the step, both planes, one 1-KiB piece per row;
    // lane l writes physical chunk l, i.e. fetches logical chunk l ^ swz(row)
    auto issue = [&](int s, int buf) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = wv * 8 + i;
            const int row = __builtin_amdgcn_readfirstlane(bu_row[ub0 + (int64_t)s * PM_KS + r]);
            const int64_t src = (int64_t)row * ld_x + ((lane ^ pm_swz(r)) * 8);
            glds16(x_hi + src, &sm.xh[buf][r][0]);
            glds16(x_lo + src, &sm.xl[buf][r][0]);
        }
    };
    const _Float16 *wah = wa_hi + ((ks0 * PM_W + wv) * 64 + lane) * 8;
    const _Float16 *wal = wa_lo + ((ks0 * PM_W + wv) * 64 + lane) * 8;
    constexpr int64_t WSTEP = (int64_t)PM_W * 64 * 8;                 // halfs per k-step of weights

    f32x4 acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposed-read addressing: group g = lane>>4 owns k rows 8g..8g+7; lane 4q+p of the group supplies
    // row 8g+q (second read: +4), logical columns 4p..4p+3 of the 16-column block
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int r_a = 8 * g + q, r_b = r_a + 4;
    const int sw_a = pm_swz(r_a), sw_b = pm_swz(r_b);
    const int half_off = (p & 1) * 4;                                   // halfs inside the 16-byte chunk

    f16x8 ah, al;
    if (nsteps > 0) {
        issue(0, 0);
        ah = *reinterpret_cast<const f16x8 *>(wah);
        al = *reinterpret_cast<const f16x8 *>(wal);
    }
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        f16x8 ah_n = ah, al_n = al;
        if (s + 1 < nsteps) {
            issue(s + 1, buf ^ 1);
            ah_n = *reinterpret_cast<const f16x8 *>(wah + (s + 1) * WSTEP);
            al_n = *reinterpret_cast<const f16x8 *>(wal + (s + 1) * WSTEP);
        }
        const _Float16 *xh_a = &sm.xh[buf][r_a][0], *xh_b = &sm.xh[buf][r_b][0];
        const _Float16 *xl_a = &sm.xl[buf][r_a][0], *xl_b = &sm.xl[buf][r_b][0];
#pragma unroll
        for (int cb4 = 0; cb4 < 32; cb4 += 4) {
            f16x8 bh[4], bl[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = 2 * (cb4 + u) + (p >> 1);                 // logical 16-byte chunk
                const int oa = ((c ^ sw_a) * 8) + half_off, ob = ((c ^ sw_b) * 8) + half_off;
                bh[u] = pm_tr8(xh_a + oa, xh_b + ob);
                bl[u] = pm_tr8(xl_a + oa, xl_b + ob);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[cb4 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[u], acc[cb4 + u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[cb4 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[u], acc[cb4 + u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[cb4 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[u], acc[cb4 + u], 0, 0, 0);
        }
        ah = ah_n;
        al = al_n;
        __syncthreads();
    }
    // ---- epilogue through LDS: wave's 16 x 512 fp32 tile -> row-major, then coalesced split/fp32 stores
    constexpr int EP = PM_EP;                                           // floats per staged row (bank-skewed)
    float *st = reinterpret_cast<float *>(smem_raw) + wv * (16 * EP);
    const int fl = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int cb = 0; cb < 32; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) st[(fq * 4 + r) * EP + cb * 16 + fl] = acc[cb][r];
    gp_wave_sync();
    const int64_t row0 = b * PM_ROWS + wv * 16;
    for (int t = 0; t < 32; ++t) {                                      // 16 rows x 128 float4 = 2048 float4 / 64 lanes
        int idx = t * 64 + lane;
        int row = idx >> 7, c4 = idx & 127;
        int64_t grow = row0 + row;
        if (grow < nv) {
            float4 v = *reinterpret_cast<const float4 *>(st + row * EP + c4 * 4);
            constexpr float inv = 1.f / PM_WSCALE;
            v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
            float xv[4] = {v.x, v.y, v.z, v.w};
            f16x4 h, l;
#pragma unroll
            for (int i = 0; i < 4; ++i) { h[i] = (_Float16)xv[i]; l[i] = (_Float16)(xv[i] - (float)h[i]); }
            if (y_hi) {
                *reinterpret_cast<f16x4 *>(y_hi + grow * ld_y + c4 * 4) = h;
                *reinterpret_cast<f16x4 *>(y_lo + grow * ld_y + c4 * 4) = l;
            }
            if (y_f32) *reinterpret_cast<float4 *>(y_f32 + grow * ld_yf + c4 * 4) = v;
        }
    }
}

size_t pm_scan_tmp(int64_t n) {
    size_t t = 0;
    (void)rocprim::exclusive_scan(nullptr, t, (int64_t *)nullptr, (int64_t *)nullptr, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), 0);
    return t;
}

}  // namespace

extern "C" size_t gp_pool_mfma_workspace_bytes(int64_t nv) {
    if (nv <= 0) return 0;
    int64_t nb = (nv + PM_ROWS - 1) / PM_ROWS;
    GpCarver cv(nullptr, 0);
    cv.take<int64_t>(nb + 1);
    cv.take<char>(pm_scan_tmp(nb + 1));
    return cv.off;
}

// pass 1: bu_off i64 [nblocks+1] (padded union rows before each block; multiple of 32), bu_n i32 [nblocks]
extern "C" int gp_pool_mfma_count(const int32_t *nbr, int64_t nv, int32_t k, int64_t *bu_off, int32_t *bu_n,
                                  void *workspace, size_t workspace_bytes, void *stream_) {
    GP_CHECK_ARG(nbr && bu_off && bu_n && workspace && nv > 0 && k > 0, "gp_pool_mfma_count: null/empty argument");
    GP_CHECK_ARG((int64_t)PM_ROWS * k <= PM_MAXID, "gp_pool_mfma_count: k=%d too large (64*k <= %d)", k, PM_MAXID);
    int64_t nb = (nv + PM_ROWS - 1) / PM_ROWS;
    GpCarver cv(workspace, workspace_bytes);
    int64_t *cnt = cv.take<int64_t>(nb + 1);
    size_t tb = pm_scan_tmp(nb + 1);
    char *tmp = cv.take<char>(tb);
    if (!cv.ok()) { gp_set_error("gp_pool_mfma_count: workspace too small"); return GP_ENOMEM; }
    hipStream_t s = gp_stream(stream_);
    GP_CHECK_HIP(hipMemsetAsync(cnt + nb, 0, sizeof(int64_t), s));
    pm_union_kernel<<<(unsigned)nb, 512, 0, s>>>(nbr, nv, k, cnt, bu_n, nullptr, nullptr);
    GP_CHECK_HIP(rocprim::exclusive_scan(tmp, tb, cnt, bu_off, (int64_t)0, (size_t)(nb + 1), rocprim::plus<int64_t>(), s));
    GP_CHECK_LAUNCH();
    return GP_OK;
}

// pass 2: bu_row i32 [total], wa_hi / wa_lo f16 [total/32 * 4 * 64 * 8] (zeroed here, then scattered)
extern "C" int gp_pool_mfma_fill(const int32_t *nbr, const float *w, int64_t nv, int32_t k, const int64_t *bu_off,
                                 const int32_t *bu_n, int64_t total_rows, int32_t *bu_row, void *wa_hi, void *wa_lo,
                                 void *stream_) {
    GP_CHECK_ARG(nbr && w && bu_off && bu_n && bu_row && wa_hi && wa_lo && nv > 0 && total_rows > 0 && total_rows % PM_KS == 0,
                 "gp_pool_mfma_fill: bad argument");
    int64_t nb = (nv + PM_ROWS - 1) / PM_ROWS;
    hipStream_t s = gp_stream(stream_);
    size_t wbytes = (size_t)(total_rows / PM_KS) * PM_W * 64 * 8 * sizeof(_Float16);
    GP_CHECK_HIP(hipMemsetAsync(wa_hi, 0, wbytes, s));
    GP_CHECK_HIP(hipMemsetAsync(wa_lo, 0, wbytes, s));
    pm_union_kernel<<<(unsigned)nb, 512, 0, s>>>(nbr, nv, k, nullptr, nullptr, bu_off, bu_row);
    int64_t ne = nv * k;
    pm_weights_kernel<<<(unsigned)((ne + 255) / 256), 256, 0, s>>>(nbr, w, nv, k, bu_off, bu_n, bu_row,
                                                                   static_cast<_Float16 *>(wa_hi), static_cast<_Float16 *>(wa_lo));
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" int gp_pool_mfma_apply(const void *x_hi, const void *x_lo, int64_t ld_x, const int64_t *bu_off,
                                  const int32_t *bu_row, const void *wa_hi, const void *wa_lo, int64_t nv, int32_t d,
                                  void *y_hi, void *y_lo, int64_t ld_y, float *y_f32, int64_t ld_yf, void *stream_) {
    GP_CHECK_ARG(x_hi && x_lo && bu_off && bu_row && wa_hi && wa_lo && nv > 0, "gp_pool_mfma_apply: null/empty argument");
    GP_CHECK_ARG(d == PM_D, "gp_pool_mfma_apply: d=%d (kernel specialised for %d columns)", d, PM_D);
    GP_CHECK_ARG((y_hi && y_lo) || y_f32, "gp_pool_mfma_apply: no output requested");
    GP_CHECK_ARG(ld_x % 8 == 0 && (uintptr_t)x_hi % 16 == 0 && (uintptr_t)x_lo % 16 == 0, "gp_pool_mfma_apply: x rows must be 16-byte aligned");
    GP_CHECK_ARG(!y_hi || (ld_y % 4 == 0 && y_hi != x_hi && y_lo != x_lo), "gp_pool_mfma_apply: y must not alias x");
    GP_CHECK_ARG(!y_f32 || ld_yf % 4 == 0, "gp_pool_mfma_apply: fp32 output rows must be 16-byte aligned");
    static bool attr_set = false;
    if (!attr_set) {
        GP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pool_mfma_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)PM_SMEM_BYTES));
        attr_set = true;
    }
    int64_t nb = (nv + PM_ROWS - 1) / PM_ROWS;
    int64_t per_xcd = (nb + 7) / 8;
    pool_mfma_kernel<<<(unsigned)(per_xcd * 8), 256, PM_SMEM_BYTES, gp_stream(stream_)>>>(
        static_cast<const _Float16 *>(x_hi), static_cast<const _Float16 *>(x_lo), ld_x, bu_off, bu_row,
        static_cast<const _Float16 *>(wa_hi), static_cast<const _Float16 *>(wa_lo), nv, nb, static_cast<_Float16 *>(y_hi),
        static_cast<_Float16 *>(y_lo), ld_y, y_f32, ld_yf, per_xcd);
    GP_CHECK_LAUNCH();
    return GP_OK;
}
