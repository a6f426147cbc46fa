// Row 12, fast path: affinity pooling with register tiling over R Morton-adjacent rows.
//
// The ELL kernel (pool.hip) moves K=96 neighbour rows per output row through L2 and is L2-bound.
// Rows that are adjacent in Morton order share most of their neighbours, so the operator is
// re-blocked ONCE per scene (it is applied 19 times): for every tile of R consecutive rows
//     union(tile)  = sorted-free set of the distinct neighbour rows of its R rows      (~14 rows per
//     Wd[u][r]     = weight of neighbour u for tile row r, 0 if u is not a neighbour     output row at R=16)
// and one wave then computes the R x D output tile as  sum_u Wd[u][:] (x) X[u, :]  with the R weights of
// a union row in SGPRs (scalar loads) and each lane holding 8 columns of every tile row in registers.
// Each union row is fetched once per tile instead of once per (row, neighbour): ~7x less L2 traffic;
// the zero entries of Wd cost FMAs, which the otherwise idle VALU absorbs.
// Summation order differs from the ELL kernel (union order instead of neighbour order): results agree
// to fp32 rounding (tests: 1e-5 against the fp64 oracle after 5 applications).
#include <cstring>
#include <rocprim/device/device_scan.hpp>

#include "gp_common.h"

namespace {

constexpr int HS = 2048;              // hash slots per tile (>= 2 * R * K for R=8,K=128 / R=16,K=64 ... checked on host)
constexpr int PT_WAVES = 4;

__device__ __forceinline__ unsigned hash_id(int id) { return ((unsigned)id * 2654435761u) >> 21; }   // 11 bits

// insert every neighbour id of the tile into the wave's LDS table (keys only)
__device__ __forceinline__ void tile_insert(int *tab, const int32_t *__restrict__ nbr, int64_t nv, int k, int R,
                                            int64_t row0, int lane) {
    for (int i = lane; i < HS; i += 64) tab[i] = -1;
    gp_wave_sync();
    const int total = R * k;
    for (int e = lane; e < total; e += 64) {
        int r = e / k, j = e - r * k;
        int64_t row = row0 + r;
        if (row >= nv) continue;
        int id = nbr[row * k + j];
        unsigned h = hash_id(id);
        while (true) {
            int old = atomicCAS(&tab[h], -1, id);
            if (old == -1 || old == id) break;
            h = (h + 1) & (HS - 1);
        }
    }
    gp_wave_sync();
}

__global__ void __launch_bounds__(PT_WAVES * 64)
tiles_count_kernel(const int32_t *__restrict__ nbr, int64_t nv, int k, int R, int64_t ntiles, int64_t *__restrict__ cnt) {
    __shared__ int s_tab[PT_WAVES][HS];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int64_t t = (int64_t)blockIdx.x * PT_WAVES + wv;
    if (t >= ntiles) return;
    int *tab = s_tab[wv];
    tile_insert(tab, nbr, nv, k, R, t * R, lane);
    int c = 0;
    for (int i = lane; i < HS; i += 64) c += tab[i] >= 0;
    c = gp_wave_sum_i(c);
    if (lane == 0) cnt[t] = c;
}

__global__ void __launch_bounds__(PT_WAVES * 64)
tiles_fill_kernel(const int32_t *__restrict__ nbr, const float *__restrict__ w, int64_t nv, int k, int R, int64_t ntiles,
                  const int64_t *__restrict__ off, int32_t *__restrict__ u_row, float *__restrict__ u_w) {
    __shared__ int s_tab[PT_WAVES][HS];
    __shared__ int s_slot[PT_WAVES][HS];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int64_t t = (int64_t)blockIdx.x * PT_WAVES + wv;
    if (t >= ntiles) return;
    int *tab = s_tab[wv], *slot = s_slot[wv];
    const int64_t row0 = t * R;
    tile_insert(tab, nbr, nv, k, R, row0, lane);
    const int64_t base = off[t];
    // slot numbering in table order (deterministic), union row ids out
    int running = 0;
    for (int i0 = 0; i0 < HS; i0 += 64) {
        int id = tab[i0 + lane];
        unsigned long long m = __ballot(id >= 0);
        int s = running + __popcll(m & ((1ull << lane) - 1ull));
        if (id >= 0) { slot[i0 + lane] = s; u_row[base + s] = id; }
        running += __popcll(m);
    }
    const int U = running;
    for (int i = lane; i < U * R; i += 64) u_w[base * R + i] = 0.f;
    gp_wave_sync();
    __threadfence_block();
    const int total = R * k;
    for (int e = lane; e < total; e += 64) {
        int r = e / k, j = e - r * k;
        int64_t row = row0 + r;
        if (row >= nv) continue;
        int id = nbr[row * k + j];
        unsigned h = hash_id(id);
        while (tab[h] != id) h = (h + 1) & (HS - 1);
        u_w[(base + slot[h]) * R + r] = w[row * k + j];
    }
}

// ------------------------------------------------------------------------------------------------
constexpr int PCH = 64;    // union entries staged per chunk

// The R weights of a union row are wave-uniform.  Streaming them through the scalar cache
// (s_load per row) measured latency-bound (37 % VALU), so each wave stages chunks of PCH union rows
// {row id, R weights} into its own LDS slice with coalesced vector loads (double-buffered: the next
// chunk is in flight while the current one is accumulated) and reads them back as broadcasts.
template <int R, int NF4, int UNR>   // NF4 float4 per lane: 2 -> 512 columns per wave, 1 -> 256 columns (slabs)
__global__ void __launch_bounds__(256)
pool_tiles_kernel(const float *__restrict__ x, int64_t ld_x, const int64_t *__restrict__ off,
                  const int32_t *__restrict__ u_row, const float *__restrict__ u_w, int64_t nv, int64_t ntiles, int d,
                  float *__restrict__ y, int64_t ld_y, int slabs, int64_t chunk) {
    __shared__ __align__(16) float s_w[4][2][PCH * R];
    __shared__ __align__(16) int s_r[4][2][PCH];
    // XCD-contiguous order: blocks b, b+8, ... share an XCD -> give each XCD a contiguous range of tiles
    int64_t b = blockIdx.x;
    int64_t lb = (b & 7) * chunk + (b >> 3);
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int64_t wid = lb * 4 + wv;
    int64_t t = wid / slabs;
    const int slab = (int)(wid - t * slabs);
    if (t >= ntiles) return;
    t = __builtin_amdgcn_readfirstlane((int)t);
    const int c0 = slab * (NF4 * 256) + lane * 4;
    const int64_t beg = off[t], end = off[t + 1];
    float4 acc[R][NF4];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int f = 0; f < NF4; ++f) acc[r][f] = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int WF4 = PCH * R / 4 / 64;                 // float4 of weights per lane per chunk (R=16: 4, 8: 2, 4: 1)
    float4 pw[WF4];
    int pr = 0;
    auto fetch_chunk = [&](int64_t s0) {                   // global -> registers (coalesced); zero padded
        int n = (int)((end - s0) < PCH ? (end - s0) : PCH);
        pr = lane < n ? u_row[s0 + lane] : 0;
        const float4 *src = reinterpret_cast<const float4 *>(u_w + s0 * R);
#pragma unroll
        for (int i = 0; i < WF4; ++i) {
            int idx = i * 64 + lane;
            pw[i] = (idx * 4 < n * R) ? src[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stash_chunk = [&](int buf) {                      // registers -> LDS
        s_r[wv][buf][lane] = pr;
#pragma unroll
        for (int i = 0; i < WF4; ++i) reinterpret_cast<float4 *>(s_w[wv][buf])[i * 64 + lane] = pw[i];
    };
    const float *xbase = x + c0;
    // group of 4 union rows: row ids by one broadcast ds_read_b128, then 4*NF4 independent 16-byte loads
    auto load_group = [&](float4 (&xv)[4][NF4], const int *rbuf, int e) {
        const int4 rr = *reinterpret_cast<const int4 *>(rbuf + e);
        const int rows[4] = {__builtin_amdgcn_readfirstlane(rr.x), __builtin_amdgcn_readfirstlane(rr.y),
                             __builtin_amdgcn_readfirstlane(rr.z), __builtin_amdgcn_readfirstlane(rr.w)};
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int f = 0; f < NF4; ++f)
                xv[u][f] = *reinterpret_cast<const float4 *>(xbase + (int64_t)rows[u] * ld_x + f * 256);
    };
    auto fma_group = [&](const float4 (&xv)[4][NF4], const float *wbuf, int e) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int r4 = 0; r4 < R / 4; ++r4) {
                const float4 w4 = *reinterpret_cast<const float4 *>(wbuf + (e + u) * R + r4 * 4);
                const float ww[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int f = 0; f < NF4; ++f) {
                        float4 &a = acc[r4 * 4 + q][f];
                        a.x = fmaf(ww[q], xv[u][f].x, a.x);
                        a.y = fmaf(ww[q], xv[u][f].y, a.y);
                        a.z = fmaf(ww[q], xv[u][f].z, a.z);
                        a.w = fmaf(ww[q], xv[u][f].w, a.w);
                    }
                }
            }
        }
    };

    if (beg < end) {
        fetch_chunk(beg);
        stash_chunk(0);
    }
    gp_wave_sync();
    int buf = 0;
    float4 xa[4][NF4], xb[4][NF4];
    for (int64_t s0 = beg; s0 < end; s0 += PCH, buf ^= 1) {
        const int n = (int)((end - s0) < PCH ? (end - s0) : PCH);
        const int ng = (n + 3) >> 2;                       // padded entries carry zero weights and row 0
        const bool more = s0 + PCH < end;
        if (more) fetch_chunk(s0 + PCH);
        const float *wbuf = s_w[wv][buf];
        const int *rbuf = s_r[wv][buf];
        load_group(xa, rbuf, 0);
        int g = 0;
        while (true) {
            if (g + 1 < ng) load_group(xb, rbuf, (g + 1) * 4);
            fma_group(xa, wbuf, g * 4);
            if (++g >= ng) break;
            if (g + 1 < ng) load_group(xa, rbuf, (g + 1) * 4);
            fma_group(xb, wbuf, g * 4);
            if (++g >= ng) break;
        }
        if (more) stash_chunk(buf ^ 1);
        gp_wave_sync();
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t row = t * R + r;
        if (row < nv) {
#pragma unroll
            for (int f = 0; f < NF4; ++f) *reinterpret_cast<float4 *>(y + row * ld_y + c0 + f * 256) = acc[r][f];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// D = 64 (BASELINE configs[0], the dense-feature lift): a row is 256 bytes = 16 lanes x float4.  One wave per tile of R rows; its four
// 16-lane quarters take every fourth union entry (four in flight per quarter: 16 union rows per wave and round trip, the next round's row
// ids prefetched under this round's rows), row id and the R weights of an entry are one address per quarter, the union row is one
// coalesced 256-byte read; the quarters' partial sums meet through two DPP-free shuffles per accumulator at the end (a fixed order: the
// result is reproducible).  The generic ELL kernel moves K x 256 B per output row through L2 (1.1 GB per application at 45k voxels: 0.045
// of 8 TB/s on config P); a tile of R = 8 rows moves its ~150 union rows once.  (First form, a tile per QUARTER: 1.4 waves per SIMD, each
// walking ~40 dependent round trips -- 0.137 ms per application, no better than ELL.)
template <int R>
__global__ void __launch_bounds__(256)
pool_tiles64_kernel(const float *__restrict__ x, int64_t ld_x, const int64_t *__restrict__ off, const int32_t *__restrict__ u_row,
                    const float *__restrict__ u_w, int64_t nv, int64_t ntiles, float *__restrict__ y, int64_t ld_y) {
    const int lane = threadIdx.x & 63, q = lane >> 4, c = (lane & 15) * 4;
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= ntiles) return;                                   // (wave-uniform)
    const int64_t beg = off[t], end = off[t + 1];
    const int len = (int)(end - beg);
    float4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int UN = 8;                                      // entries in flight per quarter; a round covers 4 * UN entries (UN = 4: 0.045 ms on config P)
    int rows[UN], rows_next[UN];
    auto load_ids = [&](int e0, int (&dst)[UN]) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int e = e0 + u * 4 + q;
            dst[u] = e < len ? u_row[beg + e] : -1;
        }
    };
    load_ids(0, rows);
    for (int e0 = 0; e0 < len; e0 += 4 * UN) {
        float4 xv[UN], w0[UN], w1[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const bool ok = rows[u] >= 0;
            const int64_t e = beg + e0 + u * 4 + q;
            xv[u] = ok ? *reinterpret_cast<const float4 *>(x + (int64_t)rows[u] * ld_x + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 *wp = reinterpret_cast<const float4 *>(u_w + e * R);
            w0[u] = ok ? wp[0] : make_float4(0.f, 0.f, 0.f, 0.f);
            w1[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (R == 8) { if (ok) w1[u] = wp[1]; }
        }
        if (e0 + 4 * UN < len) load_ids(e0 + 4 * UN, rows_next);
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const float ww[8] = {w0[u].x, w0[u].y, w0[u].z, w0[u].w, w1[u].x, w1[u].y, w1[u].z, w1[u].w};
#pragma unroll
            for (int r = 0; r < R; ++r) {
                acc[r].x = fmaf(ww[r], xv[u].x, acc[r].x);
                acc[r].y = fmaf(ww[r], xv[u].y, acc[r].y);
                acc[r].z = fmaf(ww[r], xv[u].z, acc[r].z);
                acc[r].w = fmaf(ww[r], xv[u].w, acc[r].w);
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) rows[u] = rows_next[u];
    }
    // quarters 0 + 1 and 2 + 3, then the two halves: every lane ends with the tile's sums of its four columns
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            acc[r].x += __shfl_xor(acc[r].x, o, 64);
            acc[r].y += __shfl_xor(acc[r].y, o, 64);
            acc[r].z += __shfl_xor(acc[r].z, o, 64);
            acc[r].w += __shfl_xor(acc[r].w, o, 64);
        }
    }
    // quarter q stores rows q, q + 4 (R = 8) -- 256-byte runs
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = t * R + r;
        if ((r & 3) == q && row < nv) *reinterpret_cast<float4 *>(y + row * ld_y + c) = acc[r];
    }
}

size_t scan64_tmp(int64_t n) {
    size_t t = 0;
    (void)rocprim::exclusive_scan(nullptr, t, (int64_t *)nullptr, (int64_t *)nullptr, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), 0);
    return t;
}

}  // namespace

extern int g_gp_knobs[16];
#define g_pool_nf4 g_gp_knobs[1]
#define g_pool_unroll g_gp_knobs[2]

extern "C" size_t gp_pool_tiles_workspace_bytes(int64_t nv, int32_t r) {
    if (nv <= 0 || r <= 0) return 0;
    int64_t nt = (nv + r - 1) / r;
    GpCarver cv(nullptr, 0);
    cv.take<int64_t>(nt + 1);
    cv.take<char>(scan64_tmp(nt + 1));
    return cv.off;
}

// pass 1: tile_off i64 [ntiles+1] (exclusive scan of the union sizes; tile_off[ntiles] = total entries)
extern "C" int gp_pool_tiles_count(const int32_t *nbr, int64_t nv, int32_t k, int32_t r, int64_t *tile_off,
                                   void *workspace, size_t workspace_bytes, void *stream_) {
    GP_CHECK_ARG(nbr && tile_off && workspace && nv > 0, "gp_pool_tiles_count: null/empty argument");
    GP_CHECK_ARG(r == 4 || r == 8 || r == 16, "gp_pool_tiles_count: r=%d (4, 8 or 16)", r);
    GP_CHECK_ARG(k > 0 && r * k <= HS - HS / 4, "gp_pool_tiles_count: r*k=%d too large for the %d-slot tile table", r * k, HS);
    int64_t nt = (nv + r - 1) / r;
    GpCarver cv(workspace, workspace_bytes);
    int64_t *cnt = cv.take<int64_t>(nt + 1);
    size_t tb = scan64_tmp(nt + 1);
    char *tmp = cv.take<char>(tb);
    if (!cv.ok()) { gp_set_error("gp_pool_tiles_count: workspace too small"); return GP_ENOMEM; }
    hipStream_t s = gp_stream(stream_);
    GP_CHECK_HIP(hipMemsetAsync(cnt + nt, 0, sizeof(int64_t), s));
    tiles_count_kernel<<<(unsigned)((nt + PT_WAVES - 1) / PT_WAVES), PT_WAVES * 64, 0, s>>>(nbr, nv, k, r, nt, cnt);
    GP_CHECK_HIP(rocprim::exclusive_scan(tmp, tb, cnt, tile_off, (int64_t)0, (size_t)(nt + 1), rocprim::plus<int64_t>(), s));
    GP_CHECK_LAUNCH();
    return GP_OK;
}

// pass 2: u_row i32 [total], u_w f32 [total, r]
extern "C" int gp_pool_tiles_fill(const int32_t *nbr, const float *w, int64_t nv, int32_t k, int32_t r,
                                  const int64_t *tile_off, int32_t *u_row, float *u_w, void *stream_) {
    GP_CHECK_ARG(nbr && w && tile_off && u_row && u_w && nv > 0, "gp_pool_tiles_fill: null/empty argument");
    GP_CHECK_ARG(r == 4 || r == 8 || r == 16, "gp_pool_tiles_fill: r=%d (4, 8 or 16)", r);
    int64_t nt = (nv + r - 1) / r;
    tiles_fill_kernel<<<(unsigned)((nt + PT_WAVES - 1) / PT_WAVES), PT_WAVES * 64, 0, gp_stream(stream_)>>>(nbr, w, nv, k, r, nt,
                                                                                                       tile_off, u_row, u_w);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" int gp_pool_tiles_apply(const float *x, int64_t ld_x, const int64_t *tile_off, const int32_t *u_row,
                                   const float *u_w, int32_t r, int64_t nv, int32_t d, float *y, int64_t ld_y,
                                   void *stream_) {
    GP_CHECK_ARG(x && tile_off && u_row && u_w && y && nv > 0, "gp_pool_tiles_apply: null/empty argument");
    GP_CHECK_ARG(d > 0 && d % 4 == 0 && ld_x % 4 == 0 && ld_y % 4 == 0, "gp_pool_tiles_apply: d/ld must be multiples of 4");
    GP_CHECK_ARG((uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0 && x != y, "gp_pool_tiles_apply: x/y 16-byte aligned, no alias");
    int64_t nt = (nv + r - 1) / r;
    hipStream_t s = gp_stream(stream_);
    // R=16 keeps 256 columns per wave (64 accumulator registers); R<=8 keeps 512 (tunable: gp_debug_set)
    if (d == 64) {
        // config P's width: one tile per wave, its union dealt over the four 16-lane quarters (pool_tiles64_kernel)
        GP_CHECK_ARG(r == 8 || r == 4, "gp_pool_tiles_apply: d=64 takes r=4 or r=8 tiles (r=%d)", r);
        const unsigned g64 = (unsigned)((nt + 3) / 4);
        if (r == 8) pool_tiles64_kernel<8><<<g64, 256, 0, s>>>(x, ld_x, tile_off, u_row, u_w, nv, nt, y, ld_y);
        else pool_tiles64_kernel<4><<<g64, 256, 0, s>>>(x, ld_x, tile_off, u_row, u_w, nv, nt, y, ld_y);
        GP_CHECK_LAUNCH();
        return GP_OK;
    }
    int nf4 = g_pool_nf4 ? g_pool_nf4 : ((r == 4) ? 2 : 1);   // measured best: 256 columns per wave for r=8,16
    GP_CHECK_ARG(d % (nf4 * 256) == 0, "gp_pool_tiles_apply: d=%d must be a multiple of %d (use gp_pool_ell otherwise)", d, nf4 * 256);
    int slabs = d / (nf4 * 256);
    int64_t waves = nt * slabs;
    int64_t blocks = (waves + 3) / 4;
    int64_t chunk = (blocks + 7) / 8;
    unsigned grid = (unsigned)(chunk * 8);
#define GP_PT_LAUNCH(RR, NF, UN) pool_tiles_kernel<RR, NF, UN><<<grid, 256, 0, s>>>(x, ld_x, tile_off, u_row, u_w, nv, nt, d, y, ld_y, slabs, chunk)
    switch (r * 100 + nf4 * 10 + g_pool_unroll) {
        case 1614: GP_PT_LAUNCH(16, 1, 4); break;
        case 1618: GP_PT_LAUNCH(16, 1, 8); break;
        case 1624: GP_PT_LAUNCH(16, 2, 4); break;
        case 814: GP_PT_LAUNCH(8, 1, 4); break;
        case 818: GP_PT_LAUNCH(8, 1, 8); break;
        case 824: GP_PT_LAUNCH(8, 2, 4); break;
        case 828: GP_PT_LAUNCH(8, 2, 8); break;
        case 424: GP_PT_LAUNCH(4, 2, 4); break;
        case 428: GP_PT_LAUNCH(4, 2, 8); break;
        default: gp_set_error("gp_pool_tiles_apply: unsupported variant r=%d nf4=%d unroll=%d", r, nf4, g_pool_unroll); return GP_EINVAL;
    }
#undef GP_PT_LAUNCH
    GP_CHECK_LAUNCH();
    return GP_OK;
}
