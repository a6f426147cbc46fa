// Row 12, block-shared variant of the tiled pooling kernel.
//
// pool_tiles_kernel fetches every union row of an 8-row tile from L2 (21.5 rows per output row) and is
// bound by L2->CU gather bandwidth.  Here 8 Morton-adjacent tiles (64 rows) form a block whose waves
// share ONE copy of the block's union rows (6.4 rows per output row): the 512-thread workgroup sweeps
// the block union in chunks of PB_CH rows staged into LDS by global_load_lds (whole 2-KiB rows, perfectly
// coalesced, three chunks in flight); each wave walks its own tile entries -- sorted by their position in
// the block union -- and accumulates its 8 x 512 output tile from LDS (conflict-free 1-KiB reads), with
// the 8 weights of an entry read back from a per-wave LDS batch as broadcasts (VGPR operands: an SGPR
// weight halves the v_fmac_f32 rate on gfx950).  FMAs are the useful ones of the R=8 tiling; traffic
// through L2 drops 3.3x.
#include <cstring>
#include <rocprim/device/device_scan.hpp>

#include "gp_common.h"

namespace {

constexpr int PB_R = 8;            // rows per wave tile
constexpr int PB_W = 8;            // waves (tiles) per block
constexpr int PB_ROWS = PB_R * PB_W;
constexpr int PB_CH = 16;          // union rows per LDS chunk (16 x 2 KiB = 32 KiB)
constexpr int PB_NBUF = 3;
constexpr int PB_BATCH = 32;       // tile entries per weight batch
constexpr int PB_MAXU = 4096;      // max union rows of a block handled by the builder (64 rows x 96 = 6144 worst case)

__device__ __forceinline__ void glds16(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// builder: per block, the sorted union of its tiles' union rows; per tile entry its position in it;
// entries (position, 8 weights) re-ordered by position.
__device__ __forceinline__ void bitonic_sort_lds(int *a, int n_pow2, int tid, int nthreads) {
    for (int k = 2; k <= n_pow2; k <<= 1) {
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
}

// pass A: union size per block (0 = overflow -> the caller falls back to the tile kernel)
__global__ void __launch_bounds__(512)
block_union_kernel(const int64_t *__restrict__ tile_off, const int32_t *__restrict__ u_row, int64_t ntiles, int64_t nblocks,
                   int64_t *__restrict__ bu_cnt, int32_t *__restrict__ bu_row, const int64_t *__restrict__ bu_off,
                   int32_t *__restrict__ flag) {
    __shared__ int s_ids[2 * PB_MAXU];
    __shared__ int s_n;
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x;
    int64_t t0 = b * PB_W, t1 = t0 + PB_W < ntiles ? t0 + PB_W : ntiles;
    int64_t e0 = tile_off[t0], e1 = tile_off[t1];
    int n = (int)(e1 - e0);
    if (n > 2 * PB_MAXU) {                       // cannot even sort the candidates
        if (tid == 0) { if (bu_cnt) bu_cnt[b] = 0; atomicOr(flag, 1); }
        return;
    }
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (int i = tid; i < np2; i += 512) s_ids[i] = i < n ? u_row[e0 + i] : INT32_MAX;
    __syncthreads();
    bitonic_sort_lds(s_ids, np2, tid, 512);
    // unique count / compaction (ids are distinct inside a tile but repeat across tiles)
    if (tid == 0) s_n = 0;
    __syncthreads();
    if (bu_cnt) {                                // counting pass
        int c = 0;
        for (int i = tid; i < n; i += 512) c += (i == 0 || s_ids[i] != s_ids[i - 1]);
        atomicAdd(&s_n, c);
        __syncthreads();
        if (tid == 0) {
            bu_cnt[b] = s_n;
            if (s_n > PB_MAXU) atomicOr(flag, 1);
        }
    } else {                                     // fill pass: ordered compaction by a serial-free rank: rank = #heads before i
        // heads flagged, ranks by block-wide prefix over chunks of 512
        __shared__ int s_base;
        if (tid == 0) s_base = 0;
        __syncthreads();
        const int64_t o = bu_off[b];
        for (int i0 = 0; i0 < n; i0 += 512) {
            int i = i0 + tid;
            int head = (i < n) && (i == 0 || s_ids[i] != s_ids[i - 1]);
            // wave prefix + cross-wave offsets
            unsigned long long m = __ballot(head);
            int lane = tid & 63, wv = tid >> 6;
            __shared__ int s_wcnt[8];
            if (lane == 0) s_wcnt[wv] = __popcll(m);
            __syncthreads();
            int before = s_base;
            for (int w = 0; w < wv; ++w) before += s_wcnt[w];
            int r = before + __popcll(m & ((1ull << lane) - 1ull));
            if (head) bu_row[o + r] = s_ids[i];
            __syncthreads();
            if (tid == 0) { int tot = 0; for (int w = 0; w < 8; ++w) tot += s_wcnt[w]; s_base += tot; }
            __syncthreads();
        }
    }
}

// pass B: per tile (one wave): position of every entry in the block union, entries sorted by position
__global__ void __launch_bounds__(256)
tile_positions_kernel(const int64_t *__restrict__ tile_off, const int32_t *__restrict__ u_row, const float *__restrict__ u_w,
                      int64_t ntiles, const int64_t *__restrict__ bu_off, const int32_t *__restrict__ bu_row,
                      int32_t *__restrict__ we_pos, float *__restrict__ we_w) {
    __shared__ int s_key[4][1024];               // (pos << 10 | local index), R*K <= 768 entries per tile
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int64_t t = (int64_t)blockIdx.x * 4 + wv;
    if (t >= ntiles) return;
    const int64_t e0 = tile_off[t];
    const int n = (int)(tile_off[t + 1] - e0);
    const int64_t b = t / PB_W;
    const int64_t o = bu_off[b];
    const int U = (int)(bu_off[b + 1] - o);
    int *key = s_key[wv];
    for (int i = lane; i < 1024; i += 64) {
        int k = INT32_MAX;
        if (i < n) {
            int id = u_row[e0 + i];
            int lo = 0, hi = U - 1;                                  // binary search in the sorted block union
            while (lo < hi) { int mid = (lo + hi) >> 1; if (bu_row[o + mid] < id) lo = mid + 1; else hi = mid; }
            k = (lo << 10) | i;
        }
        key[i] = k;
    }
    gp_wave_sync();
    // wave-level bitonic sort of 1024 keys
    for (int k = 2; k <= 1024; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = lane; i < 1024; i += 64) {
                int ixj = i ^ j;
                if (ixj > i) {
                    int x = key[i], y = key[ixj];
                    bool up = (i & k) == 0;
                    if ((x > y) == up) { key[i] = y; key[ixj] = x; }
                }
            }
            gp_wave_sync();
        }
    for (int i = lane; i < n; i += 64) {
        int kk = key[i];
        int src = kk & 1023;
        we_pos[e0 + i] = kk >> 10;
        const float4 *ws = reinterpret_cast<const float4 *>(u_w + (e0 + src) * PB_R);
        float4 *wd = reinterpret_cast<float4 *>(we_w + (e0 + i) * PB_R);
        wd[0] = ws[0];
        wd[1] = ws[1];
    }
}

// ------------------------------------------------------------------------------------------------
struct PbSmem {
    float xs[PB_NBUF][PB_CH][512];               // 3 x 32 KiB staged union rows
    float ww[PB_W][2][PB_BATCH * PB_R];          // per-wave weight batches (2 x 1 KiB)
    int wp[PB_W][2][PB_BATCH];                   // per-wave position batches
};

__global__ void __launch_bounds__(512)
pool_block_kernel(const float *__restrict__ x, int64_t ld_x, const int64_t *__restrict__ bu_off,
                  const int32_t *__restrict__ bu_row, const int64_t *__restrict__ tile_off,
                  const int32_t *__restrict__ we_pos, const float *__restrict__ we_w, int64_t nv, int64_t ntiles,
                  int64_t nblocks, float *__restrict__ y, int64_t ld_y, int64_t per_xcd) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    PbSmem &sm = *reinterpret_cast<PbSmem *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t b = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);     // XCD-contiguous block order
    if (b >= nblocks) return;
    const int64_t t = b * PB_W + wv;
    const int64_t ub0 = bu_off[b];
    const int UB = (int)(bu_off[b + 1] - ub0);
    const int nch = (UB + PB_CH - 1) / PB_CH;
    int64_t e0 = 0, e1 = 0;
    if (t < ntiles) { e0 = tile_off[t]; e1 = tile_off[t + 1]; }

    // x chunk staging: 16 rows x 2 KiB = 32 DMA pieces of 1 KiB; wave wv moves rows 2wv, 2wv+1 (both halves)
    auto issue_x = [&](int c, int buf) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            int lr = wv * 2 + r;
            int ui = c * PB_CH + lr;
            int row = ui < UB ? bu_row[ub0 + ui] : bu_row[ub0];       // padding rows re-fetch row 0 (never referenced)
            row = __builtin_amdgcn_readfirstlane(row);
            const float *src = x + (int64_t)row * ld_x + lane * 4;
            glds16(src, &sm.xs[buf][lr][0]);
            glds16(src + 256, &sm.xs[buf][lr][256]);
        }
    };
    // weight batch staging: 32 entries x 8 weights = 1 KiB (one DMA piece) + 32 positions
    auto issue_batch = [&](int64_t eb, int bb) {
        int64_t last = e1 - 1;
        int64_t wi = eb * PB_R + lane * 4;                            // floats; clamp inside the tile's entries
        int64_t wmax = (last + 1) * PB_R - 4;
        glds16(we_w + (wi < wmax ? wi : wmax), &sm.ww[wv][bb][0]);
        if (lane < PB_BATCH) sm.wp[wv][bb][lane] = (eb + lane <= last) ? we_pos[eb + lane] : 0x7fffffff;
    };

    float4 acc[PB_R][2];
#pragma unroll
    for (int r = 0; r < PB_R; ++r) { acc[r][0] = make_float4(0.f, 0.f, 0.f, 0.f); acc[r][1] = make_float4(0.f, 0.f, 0.f, 0.f); }

    // prologue: chunks 0 and 1, weight batch 0
    if (nch > 0) issue_x(0, 0);
    if (nch > 1) issue_x(1, 1);
    int64_t bstart = e0;
    int bb = 0;
    if (e0 < e1) issue_batch(e0, 0);
    __syncthreads();                                                   // drains the DMAs (vmcnt(0)) and publishes LDS
    int64_t ecur = e0;
    for (int c = 0; c < nch; ++c) {
        const int buf = c % PB_NBUF;
        if (c + 2 < nch) issue_x(c + 2, (c + 2) % PB_NBUF);
        const int cbeg = c * PB_CH, cend = cbeg + PB_CH;
        while (ecur < e1) {
            int li = (int)(ecur - bstart);
            if (li == PB_BATCH) {                                      // next weight batch (wave-local)
                bstart += PB_BATCH;
                bb ^= 1;
                issue_batch(bstart, bb);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                gp_wave_sync();
                li = 0;
            }
            const int pos = __builtin_amdgcn_readfirstlane(sm.wp[wv][bb][li]);
            if (pos >= cend) break;
            const float *xr = &sm.xs[buf][pos - cbeg][lane * 4];
            const float4 x0 = *reinterpret_cast<const float4 *>(xr);
            const float4 x1 = *reinterpret_cast<const float4 *>(xr + 256);
            const float4 wa = *reinterpret_cast<const float4 *>(&sm.ww[wv][bb][li * PB_R]);
            const float4 wb = *reinterpret_cast<const float4 *>(&sm.ww[wv][bb][li * PB_R + 4]);
            const float wr[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
#pragma unroll
            for (int r = 0; r < PB_R; ++r) {
                acc[r][0].x = fmaf(wr[r], x0.x, acc[r][0].x); acc[r][0].y = fmaf(wr[r], x0.y, acc[r][0].y);
                acc[r][0].z = fmaf(wr[r], x0.z, acc[r][0].z); acc[r][0].w = fmaf(wr[r], x0.w, acc[r][0].w);
                acc[r][1].x = fmaf(wr[r], x1.x, acc[r][1].x); acc[r][1].y = fmaf(wr[r], x1.y, acc[r][1].y);
                acc[r][1].z = fmaf(wr[r], x1.z, acc[r][1].z); acc[r][1].w = fmaf(wr[r], x1.w, acc[r][1].w);
            }
            ++ecur;
        }
        // chunk c+1 must have landed and every wave must be done with chunk c; chunk c+2 (this wave's 4
        // newest DMA pieces) stays in flight across the barrier
        if (c + 2 < nch) asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (t < ntiles) {
#pragma unroll
        for (int r = 0; r < PB_R; ++r) {
            int64_t row = t * PB_R + r;
            if (row < nv) {
                *reinterpret_cast<float4 *>(y + row * ld_y + lane * 4) = acc[r][0];
                *reinterpret_cast<float4 *>(y + row * ld_y + 256 + lane * 4) = acc[r][1];
            }
        }
    }
}

size_t scan64_tmp_b(int64_t n) {
    size_t t = 0;
    (void)rocprim::exclusive_scan(nullptr, t, (int64_t *)nullptr, (int64_t *)nullptr, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), 0);
    return t;
}

}  // namespace

extern "C" size_t gp_pool_blocks_workspace_bytes(int64_t nv) {
    if (nv <= 0) return 0;
    int64_t nb = (nv + PB_ROWS - 1) / PB_ROWS;
    GpCarver cv(nullptr, 0);
    cv.take<int64_t>(nb + 1);
    cv.take<int32_t>(64);
    cv.take<char>(scan64_tmp_b(nb + 1));
    return cv.off;
}

// pass 1: bu_off i64 [nblocks+1] from the R=8 tile arrays; *flag_dev != 0 if a block union exceeds the builder's capacity
extern "C" int gp_pool_blocks_count(const int64_t *tile_off, const int32_t *u_row, int64_t nv, int64_t *bu_off,
                                    int32_t *flag_dev, void *workspace, size_t workspace_bytes, void *stream_) {
    GP_CHECK_ARG(tile_off && u_row && bu_off && flag_dev && workspace && nv > 0, "gp_pool_blocks_count: null/empty argument");
    int64_t nt = (nv + PB_R - 1) / PB_R, nb = (nv + PB_ROWS - 1) / PB_ROWS;
    GpCarver cv(workspace, workspace_bytes);
    int64_t *cnt = cv.take<int64_t>(nb + 1);
    cv.take<int32_t>(64);
    size_t tb = scan64_tmp_b(nb + 1);
    char *tmp = cv.take<char>(tb);
    if (!cv.ok()) { gp_set_error("gp_pool_blocks_count: workspace too small"); return GP_ENOMEM; }
    hipStream_t s = gp_stream(stream_);
    GP_CHECK_HIP(hipMemsetAsync(cnt + nb, 0, sizeof(int64_t), s));
    GP_CHECK_HIP(hipMemsetAsync(flag_dev, 0, sizeof(int32_t), s));
    block_union_kernel<<<(unsigned)nb, 512, 0, s>>>(tile_off, u_row, nt, nb, cnt, nullptr, nullptr, flag_dev);
    GP_CHECK_HIP(rocprim::exclusive_scan(tmp, tb, cnt, bu_off, (int64_t)0, (size_t)(nb + 1), rocprim::plus<int64_t>(), s));
    GP_CHECK_LAUNCH();
    return GP_OK;
}

// pass 2: bu_row i32 [total union], we_pos i32 [entries], we_w f32 [entries, 8] (entries re-ordered by position)
extern "C" int gp_pool_blocks_fill(const int64_t *tile_off, const int32_t *u_row, const float *u_w, int64_t nv,
                                   const int64_t *bu_off, int32_t *bu_row, int32_t *we_pos, float *we_w,
                                   int32_t *flag_dev, void *stream_) {
    GP_CHECK_ARG(tile_off && u_row && u_w && bu_off && bu_row && we_pos && we_w && flag_dev && nv > 0, "gp_pool_blocks_fill: null/empty argument");
    int64_t nt = (nv + PB_R - 1) / PB_R, nb = (nv + PB_ROWS - 1) / PB_ROWS;
    hipStream_t s = gp_stream(stream_);
    block_union_kernel<<<(unsigned)nb, 512, 0, s>>>(tile_off, u_row, nt, nb, nullptr, bu_row, bu_off, flag_dev);
    tile_positions_kernel<<<(unsigned)((nt + 3) / 4), 256, 0, s>>>(tile_off, u_row, u_w, nt, bu_off, bu_row, we_pos, we_w);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" int gp_pool_blocks_apply(const float *x, int64_t ld_x, const int64_t *bu_off, const int32_t *bu_row,
                                    const int64_t *tile_off, const int32_t *we_pos, const float *we_w, int64_t nv, int32_t d,
                                    float *y, int64_t ld_y, void *stream_) {
    GP_CHECK_ARG(x && bu_off && bu_row && tile_off && we_pos && we_w && y && nv > 0, "gp_pool_blocks_apply: null/empty argument");
    GP_CHECK_ARG(d == 512, "gp_pool_blocks_apply: d=%d (this kernel is specialised for 512 columns; use gp_pool_tiles/ell)", d);
    GP_CHECK_ARG(ld_x % 4 == 0 && ld_y % 4 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0 && x != y,
                 "gp_pool_blocks_apply: x/y 16-byte aligned rows, no alias");
    static bool attr_set = false;
    if (!attr_set) {
        GP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pool_block_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PbSmem)));
        attr_set = true;
    }
    int64_t nt = (nv + PB_R - 1) / PB_R, nb = (nv + PB_ROWS - 1) / PB_ROWS;
    int64_t per_xcd = (nb + 7) / 8;
    pool_block_kernel<<<(unsigned)(per_xcd * 8), 512, sizeof(PbSmem), gp_stream(stream_)>>>(
        x, ld_x, bu_off, bu_row, tile_off, we_pos, we_w, nv, nt, nb, y, ld_y, per_xcd);
    GP_CHECK_LAUNCH();
    return GP_OK;
}
