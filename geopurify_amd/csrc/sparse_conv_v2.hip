// Row 9, fast path: submanifold sparse convolution as a per-offset gather-GEMM on the f16 matrix
// cores with fp32-class accuracy, followed by a deterministic per-voxel sum.
//
//   pairs     For every kernel offset k the (input row, output row) pairs are compacted once per
//             scene (flags -> scan -> scatter over the [27,Nv] kernel map); pair_pos[k][u] is the
//             global pair index of (k,u) or -1.
//   phase 1   P[pair, :] = X[in(pair), :] @ W[k]      (256 pairs x 256 channels per workgroup,
//             accumulators in registers, Cin reduced in 32-channel steps through a double-buffered
//             LDS ring).  fp32 operands are split on the fly into two f16 halves x = hi + lo
//             (|x - hi - lo| <= 2^-22 |x|); the product keeps hi*hi + hi*lo + lo*hi, three
//             v_mfma_f32_16x16x32_f16 per tile with fp32 accumulation -- every f16 product is exact in
//             fp32, the dropped lo*lo term is 2^-22 relative.  Weights are pre-split, pre-scaled by a
//             power of two and stored [k][cout][cin] so that the B fragment is a straight 16-byte read.
//   phase 2   Y[u, :] = epilogue( sum_k P[pair_pos[k][u], :] )  in ascending k (bitwise reproducible),
//             BatchNorm(eval) scale/shift, residual, ReLU fused.
//
// Weight re-use is per offset (all pairs of k stream through W[k]) instead of per output tile, which
// cuts the weight traffic from L2 by ~8x against the output-stationary v1 kernel.
#include <cstring>
#include <type_traits>
#include <rocprim/device/device_scan.hpp>

#include <hip/hip_fp16.h>

#include <stdlib.h>

#include "gp_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 256, TN = 256, TK = 32;
constexpr int NT2 = 512;
constexpr int APITCH = 32;   // halfs per LDS row: 64-byte rows = four 16-byte slots
// XOR swizzle of the 16-byte slot inside a row so that the ds_read_b128 lane groups of a 16-row
// fragment read hit 16 distinct slot positions of the 256-byte bank row (checked for all four groups)
__device__ __forceinline__ int sw_slot(int row, int slot) {
    const int h = (0x78 >> (((row >> 2) & 3) * 2)) & 3;      // h = {0,2,3,1}[(row>>2)&3] ... packed 2 bits each
    return slot ^ h;
}

// ------------------------------------------------------------------------------------------------
// Pairs are ordered by (chunk of CH Morton-consecutive output rows, offset k, output row): a chunk's
// 27 offset segments gather from the same few thousand input rows, so the A operand stays in the
// XCD's L2 while the chunk is processed (k-major order re-read every input row ~7x from beyond L2).
// (the chunks are given by their row offsets R[0 .. nchunks], R[0] = 0, R[nchunks] = nv: equal heights, or heights chosen so that
// every chunk is a whole number of rounds of phase-1 tiles -- gp_conv_chunk_plan)
__device__ __forceinline__ void pair_decode(int64_t i, int kv, const int32_t *__restrict__ R, int nchunks, int &c, int &k, int64_t &u) {
    const int row_q = (int)(i / kv);                          // chunk c starts at pair slot kv * R[c]
    int lo = 0, hi = nchunks - 1;
    while (lo < hi) {                                         // last c with R[c] <= row_q
        const int mid = (lo + hi + 1) >> 1;
        if (R[mid] <= row_q) lo = mid; else hi = mid - 1;
    }
    c = lo;
    // (the table lives on the device and is not validated there: a malformed one -- not ascending, not ending at nv -- must give
    // wrong pairs, never an access outside the map; gp_sparse_conv_f16x3 checks the host copy)
    const int64_t r0 = R[c], rows_raw = R[c + 1] - r0, rows_c = rows_raw > 0 ? rows_raw : 1;
    const int64_t rem = i - (int64_t)kv * r0;
    int64_t kk = rem / rows_c;
    kk = kk < 0 ? 0 : (kk >= kv ? kv - 1 : kk);
    k = (int)kk;
    u = r0 + (rem - kk * rows_c);
}
__global__ void pair_flags_kernel(const int32_t *__restrict__ nm, int64_t nv, int kv, const int32_t *__restrict__ R, int nchunks,
                                  int32_t *__restrict__ f) {
    int64_t total = (int64_t)kv * nv;
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= total) return;
    int c, k; int64_t u;
    pair_decode(i, kv, R, nchunks, c, k, u);
    u = u < 0 ? 0 : (u >= nv ? nv - 1 : u);
    f[i] = nm[(int64_t)k * nv + u] >= 0 ? 1 : 0;
}
__global__ void pair_emit_kernel(const int32_t *__restrict__ nm, const int32_t *__restrict__ sc, int64_t nv, int kv,
                                 const int32_t *__restrict__ R, int nchunks, int32_t *__restrict__ pair_in, int32_t *__restrict__ pair_pos,
                                 int32_t *__restrict__ seg_off /*[nseg+1]*/) {
    int64_t total = (int64_t)kv * nv;
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= total) return;
    int c, k; int64_t u;
    pair_decode(i, kv, R, nchunks, c, k, u);
    u = u < 0 ? 0 : (u >= nv ? nv - 1 : u);
    int in = nm[(int64_t)k * nv + u];
    int s = sc[i];
    pair_pos[(int64_t)k * nv + u] = in >= 0 ? s : -1;
    if (in >= 0) pair_in[s] = in;
    if (u == (int64_t)R[c]) seg_off[c * kv + k] = s;             // first row of the (chunk, offset) segment
    if (i == total - 1) seg_off[(c + 1) * kv] = s + (in >= 0 ? 1 : 0);
}
// ---- chunk heights chosen for the phase-1 launch: per granule of rows and offset, the number of pairs; then ONE wave walks the
// granules (lane = offset) and closes a chunk when its tile count -- sum over offsets of ceil(pairs / TM), times the column tiles --
// would pass the target (a whole number of rounds of one-tile workgroups on the chip)
__global__ void chunk_count_kernel(const int32_t *__restrict__ nm, int64_t nv, int kv, int granule, int32_t *__restrict__ cnt /*[ngran][32]*/) {
    const int g = blockIdx.x, k = blockIdx.y;
    const int64_t r0 = (int64_t)g * granule, r1 = r0 + granule < nv ? r0 + granule : nv;
    int c = 0;
    for (int64_t u = r0 + threadIdx.x; u < r1; u += blockDim.x) c += nm[(int64_t)k * nv + u] >= 0;
    c = gp_wave_sum(c);
    __shared__ int s_c[4];
    if (gp_lane() == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) cnt[g * 32 + k] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
}
// (256 threads stage windows of 512 granules' counts in LDS as u16 -- a count is at most the granule's rows -- and wave 0 walks them:
// a dependent global load per granule made this one-wave loop 0.23 ms on the S scene)
constexpr int CP_WIN = 512;
// sum over lanes 0..31 (wave-uniform result) with DPP adds inside the 16-lane rows and two v_readlane: the shuffle form (six
// ds_bpermute round trips) made the one-wave walk below 0.28 us per granule
__device__ __forceinline__ int cp_sum32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);      // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);     // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);     // row_mirror
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16);
}
__global__ void __launch_bounds__(256)
chunk_plan_kernel(const int32_t *__restrict__ cnt, int ngran, int kv, int granule, int64_t nv, int col_tiles, int target,
                  int max_chunks, int32_t *__restrict__ R, int32_t *__restrict__ n_chunks) {
    __shared__ unsigned short s_cnt[CP_WIN * 32];
    const int tid = threadIdx.x, lane = tid & 63;
    int sum = 0, nc = 0, start_g = 0;
    if (tid == 0) R[0] = 0;
    for (int g0 = 0; g0 < ngran; g0 += CP_WIN) {
        const int n = ngran - g0 < CP_WIN ? ngran - g0 : CP_WIN;
        __syncthreads();
        {   // 16-byte loads, four in flight per thread (one dword load per iteration was one memory round trip per 256 counts)
            const int4 *c4 = reinterpret_cast<const int4 *>(cnt + (int64_t)g0 * 32);
            const int n4 = n * 8;
            for (int i0 = tid; i0 < n4; i0 += 1024) {
                int4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = (i0 + q * 256 < n4) ? c4[i0 + q * 256] : make_int4(0, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = i0 + q * 256;
                    if (i < n4) {
                        const int k0 = (i * 4) & 31;                      // columns kv .. 31 of the workspace are not written
                        s_cnt[i * 4 + 0] = (unsigned short)(k0 + 0 < kv ? v[q].x : 0);
                        s_cnt[i * 4 + 1] = (unsigned short)(k0 + 1 < kv ? v[q].y : 0);
                        s_cnt[i * 4 + 2] = (unsigned short)(k0 + 2 < kv ? v[q].z : 0);
                        s_cnt[i * 4 + 3] = (unsigned short)(k0 + 3 < kv ? v[q].w : 0);
                    }
                }
            }
        }
        __syncthreads();
        if (tid < 64) {
            // four granules per round: their running sums and tile counts are computed as if no chunk closed among them (four
            // independent reductions in flight instead of one dependent chain per granule); a close -- 16 in 524 granules --
            // invalidates what follows it in the round, which is then redone one granule at a time
            auto step = [&](int g, int add, int tiles_with) {
                if (tiles_with > target && g != start_g && nc + 1 < max_chunks) {      // close the chunk in front of this granule
                    ++nc;
                    if (lane == 0) R[nc] = g * granule;
                    sum = add;
                    start_g = g;
                    return true;
                }
                sum += add;
                return false;
            };
            for (int gl = 0; gl < n; gl += 4) {
                int a[4], run[4], t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] = (lane < 32 && gl + j < n) ? s_cnt[(gl + j) * 32 + lane] : 0;
                run[0] = sum + a[0];
#pragma unroll
                for (int j = 1; j < 4; ++j) run[j] = run[j - 1] + a[j];
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = cp_sum32(lane < 32 ? (run[j] + TM - 1) / TM : 0) * col_tiles;
                bool redo = false;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (gl + j >= n) break;
                    const int tiles_with = redo ? cp_sum32(lane < 32 ? (sum + a[j] + TM - 1) / TM : 0) * col_tiles : t[j];
                    redo = step(g0 + gl + j, a[j], tiles_with) || redo;
                }
            }
        }
    }
    if (tid == 0) { R[nc + 1] = (int32_t)nv; n_chunks[0] = nc + 1; }
}
// one descriptor per 256-pair tile: {offset k, first pair, pair count} -- a single 16-byte load in the
// GEMM prologue instead of a dependent binary search over tile_start
__global__ void tile_desc_kernel(const int32_t *__restrict__ seg_off, const int32_t *__restrict__ tile_start, int nseg, int kv,
                                 int4 *__restrict__ desc) {
    int sgi = blockIdx.x * blockDim.x + threadIdx.x;
    if (sgi >= nseg) return;
    int t0 = tile_start[sgi], t1 = tile_start[sgi + 1];
    int p0 = seg_off[sgi], p1 = seg_off[sgi + 1];
    for (int t = t0; t < t1; ++t) {
        int base = p0 + (t - t0) * TM;
        desc[t] = make_int4(sgi % kv, base, min(TM, p1 - base), 0);
    }
}

// tile_start[s] = number of TM-row tiles before segment s (serial scan over a few thousand segments)
__global__ void tile_start_kernel(const int32_t *__restrict__ seg_off, int nseg, int32_t *__restrict__ tile_start) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    int acc = 0;
    for (int s = 0; s < nseg; ++s) {
        tile_start[s] = acc;
        acc += (seg_off[s + 1] - seg_off[s] + TM - 1) / TM;
    }
    tile_start[nseg] = acc;
}

// ------------------------------------------------------------------------------------------------
struct V2Smem {
    _Float16 a_hi[2][TM][APITCH];
    _Float16 a_lo[2][TM][APITCH];
    _Float16 b_hi[2][TN][APITCH];
    _Float16 b_lo[2][TN][APITCH];
};

__device__ __forceinline__ void split8(const float4 &u, const float4 &v, f16x8 &hi, f16x8 &lo) {
    float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        _Float16 h = (_Float16)x[i];
        hi[i] = h;
        lo[i] = (_Float16)(x[i] - (float)h);
    }
}


// one Cin step of a wave's 64x128 tile with a compile-time number of live 16-row tiles (the last
// m-tile of an offset is partial): no per-MFMA control flow, term-major order so that consecutive
// MFMAs hit different accumulators.
template <int NRT>
__device__ __forceinline__ void mma_step_f16x3(const V2Smem &sm, int buf, int wm, int wn, int fl, int fsw,
                                               f32x4 (&acc)[4][8]) {
    f16x8 ah[NRT > 0 ? NRT : 1], al[NRT > 0 ? NRT : 1];
#pragma unroll
    for (int i = 0; i < NRT; ++i) {
        ah[i] = *reinterpret_cast<const f16x8 *>(&sm.a_hi[buf][wm * 64 + i * 16 + fl][fsw]);
        al[i] = *reinterpret_cast<const f16x8 *>(&sm.a_lo[buf][wm * 64 + i * 16 + fl][fsw]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        f16x8 bh = *reinterpret_cast<const f16x8 *>(&sm.b_hi[buf][wn * 128 + j * 16 + fl][fsw]);
        f16x8 bl = *reinterpret_cast<const f16x8 *>(&sm.b_lo[buf][wn * 128 + j * 16 + fl][fsw]);
#pragma unroll
        for (int i = 0; i < NRT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NRT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NRT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh, acc[i][j], 0, 0, 0);
    }
}
// the same step with the A fragments at explicit byte offsets of the dynamic LDS (the LDS-DMA kernel has two A images: separate
// hi / lo planes of 64-byte rows, or ONE image of 128-byte rows [hi 32 | lo 32] filled by full-line pieces)
template <int NRT>
__device__ __forceinline__ void mma_step_f16x3_off(const unsigned char *smem, const uint32_t (&fa_hi)[4], const uint32_t (&fa_lo)[4],
                                                   uint32_t a_off, const V2Smem &sm, int buf, int wn, int fl, int fsw, f32x4 (&acc)[4][8]) {
    f16x8 ah[NRT > 0 ? NRT : 1], al[NRT > 0 ? NRT : 1];
#pragma unroll
    for (int i = 0; i < NRT; ++i) {
        ah[i] = *reinterpret_cast<const f16x8 *>(smem + fa_hi[i] + a_off);
        al[i] = *reinterpret_cast<const f16x8 *>(smem + fa_lo[i] + a_off);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        f16x8 bh = *reinterpret_cast<const f16x8 *>(&sm.b_hi[buf][wn * 128 + j * 16 + fl][fsw]);
        f16x8 bl = *reinterpret_cast<const f16x8 *>(&sm.b_lo[buf][wn * 128 + j * 16 + fl][fsw]);
#pragma unroll
        for (int i = 0; i < NRT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NRT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NRT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh, acc[i][j], 0, 0, 0);
    }
}
// grid.x = (#m-tiles upper bound) * n_tiles ; tile -> offset k by a search in tile_off (device)
// (TUNE: the tuning bits of `ablate_` -- knob 3 -- are honoured; the product instantiation compiles them out)
template <bool TUNE>
__global__ void __launch_bounds__(NT2)
conv_phase1_kernel(const float *__restrict__ x, int64_t ld_x, const int32_t *__restrict__ pair_in,
                   const int32_t *__restrict__ off, const int32_t *__restrict__ tile_start, const int4 *__restrict__ tile_desc,
                   int nseg, int kv, const _Float16 *__restrict__ w_hi, const _Float16 *__restrict__ w_lo, int cin, int cout,
                   float *__restrict__ P, int n_tiles, int ablate_, int tile_begin, int tile_count, int pair_base, int w_blocked) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int ablate = TUNE ? ablate_ : 0;
    V2Smem &sm = *reinterpret_cast<V2Smem *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // ---- which (offset, m-tile, n-tile)?
    // XCD-contiguous tile order: blocks b, b+8, ... share an XCD; each XCD walks a contiguous range of
    // (m-tile, n-tile) pairs, both channel tiles of an m-tile back to back (shared A rows hit its L2)
    const int64_t nb = gridDim.x, per_xcd = nb >> 3;
    const int64_t lb = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int nt = (int)(lb % n_tiles);
    const int mt_local = (int)(lb / n_tiles);
    if (mt_local >= tile_count) return;
    const int mt = tile_begin + mt_local;
    if (mt >= tile_start[nseg]) return;
    const int4 td = tile_desc[mt];
    const int k = td.x, base = td.y, cnt = td.z;
    const int n0 = nt * TN;
    // staging roles: A row = tid/2 (pair), half = tid%2 (16 channels); B col = tid/2, half = tid%2
    const int s_row = tid >> 1, s_half = tid & 1;
    const bool a_ok = s_row < cnt;
    const int in_row = a_ok ? pair_in[base + s_row] : 0;
    const float *xa = x + (int64_t)in_row * ld_x + s_half * 16;
    const _Float16 *wbh = w_hi + ((int64_t)k * cout + n0 + s_row) * cin + s_half * 16;
    const _Float16 *wbl = w_lo + ((int64_t)k * cout + n0 + s_row) * cin + s_half * 16;
    if (w_blocked) {
        // step-blocked weights: column c of the tile sits in layout row (c & 128) | (c & 7) << 4 | (c >> 3 & 15) (the LDS-DMA kernel's row)
        const int rho = (s_row & 128) | ((s_row & 7) << 4) | ((s_row >> 3) & 15);
        const int64_t o = ((((int64_t)k * n_tiles + nt) * (cin / TK)) * TN + rho) * TK + s_half * 16;
        wbh = w_hi + o;
        wbl = w_lo + o;
    }
    const int bmul = w_blocked ? TN : 1;

    float4 ra[4];
    f16x8 rbh[2], rbl[2];
    auto load_step = [&](int c0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            ra[i] = a_ok ? *reinterpret_cast<const float4 *>(xa + c0 + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            rbh[i] = *reinterpret_cast<const f16x8 *>(wbh + c0 * bmul + i * 8);
            rbl[i] = *reinterpret_cast<const f16x8 *>(wbl + c0 * bmul + i * 8);
        }
    };
    auto store_step = [&](int buf) {
        f16x8 h0, l0, h1, l1;
        split8(ra[0], ra[1], h0, l0);
        split8(ra[2], ra[3], h1, l1);
        const int q0 = sw_slot(s_row, s_half * 2) * 8, q1 = sw_slot(s_row, s_half * 2 + 1) * 8;
        *reinterpret_cast<f16x8 *>(&sm.a_hi[buf][s_row][q0]) = h0;
        *reinterpret_cast<f16x8 *>(&sm.a_hi[buf][s_row][q1]) = h1;
        *reinterpret_cast<f16x8 *>(&sm.a_lo[buf][s_row][q0]) = l0;
        *reinterpret_cast<f16x8 *>(&sm.a_lo[buf][s_row][q1]) = l1;
        *reinterpret_cast<f16x8 *>(&sm.b_hi[buf][s_row][q0]) = rbh[0];
        *reinterpret_cast<f16x8 *>(&sm.b_hi[buf][s_row][q1]) = rbh[1];
        *reinterpret_cast<f16x8 *>(&sm.b_lo[buf][s_row][q0]) = rbl[0];
        *reinterpret_cast<f16x8 *>(&sm.b_lo[buf][s_row][q1]) = rbl[1];
    };

    // wave tile: rows wm*64 .. +64 (4 row tiles), cols wn*128 .. +128 (8 col tiles)
    const int wm = wv >> 1, wn = wv & 1;
    const int fl = lane & 15, fq = lane >> 4;
    const int fsw = sw_slot(fl, fq) * 8;                       // row bits 2..3 come from fl in every 16-row tile
    // skip row tiles that are entirely padding
    const int rows_here = cnt - wm * 64;
    const int nrt = __builtin_amdgcn_readfirstlane(rows_here <= 0 ? 0 : (rows_here >= 64 ? 4 : (rows_here + 15) >> 4));

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int steps = cin / TK;
    load_step(0);
    store_step(0);
    __syncthreads();
    for (int s = 0; s < steps; ++s) {
        const int buf = s & 1;
        if (s + 1 < steps && !(ablate & 1)) load_step((s + 1) * TK);
        if (!(ablate & 2)) mma_step_f16x3<4>(sm, buf, wm, wn, fl, fsw, acc);   // padded rows are computed and discarded
        if (s + 1 < steps && !(ablate & 4)) store_step(buf ^ 1);
        __syncthreads();
    }
    if (ablate & 8) return;
    // ---- store the partial rows (fp32).  C layout: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < nrt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int row = wm * 64 + i * 16 + fq * 4 + r;
                if (row < cnt) {
                    float *dst = P + (int64_t)(base - pair_base + row) * cout + n0 + wn * 128 + fl;
#pragma unroll
                    for (int j = 0; j < 8; ++j) dst[j * 16] = acc[i][j][r];
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// phase 1, LDS-DMA variant: activations arrive pre-split (hi/lo f16 rows written by the previous
// layer's epilogue), so both operands are staged global -> LDS by global_load_lds_dwordx4 with no
// VGPR round trip, no conversion and no ds_write in the loop.  The LDS image is lane-linear per
// instruction (16 rows x 64 B); the XOR swizzle is applied on the SOURCE address (slot q = p ^ h(row))
// and again on the fragment reads.  Epilogue goes through LDS so that rows are stored in 512-byte runs.
__device__ __forceinline__ void glds16(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

__device__ __forceinline__ uint64_t cv_now() {
    uint64_t t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ uint64_t cv_real() {
    uint64_t t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// STAMP (tuning twin only): waves 0 and 4 of every workgroup write 16 x uint64 into `stamp` (wave 0: [0..11], wave 4: [12..15] = its DMA-issue / wait / barrier cycles and HW_ID): {real-time start, real-time length, prologue
// (descriptor + row ids + first stage landed), K loop, partial-store ISSUE, store drain (vmcnt(0)), whole tile -- shader cycles --,
// XCC id | pairs << 8, cycles of the loop spent issuing LDS-DMA, in the hand-over's `s_waitcnt`, in its `s_barrier`, HW_ID}
// Round-4 stamps of this loop (scripts/stamp_conv.py, profiles/r04_conv_stamps.log): per step 550-820 cycles of DMA issue during
// which neither wave of a SIMD feeds the matrix pipe, 2 150 of reads + MFMA, 1 850 waiting at the barrier for the SIMD partner's
// MFMAs -- 4 550-4 800 cycles for 3 072 cycles of matrix work.  Measured and left out: waves 0-3 issue ALL of a stage's LDS-DMA (64 rows of each operand), waves 4-7 none, so that a SIMD's second wave
// multiplies while the first absorbs the memory pipeline's back-pressure: the step falls to 4 150 cycles (-8.6 %), the in-kernel
// clock from 2.04 to 1.99 GHz, and the layer takes the same 1.929 ms -- on all-zero operands the same cycle count runs at
// 2.37 GHz and 1.664 ms: the layer is bound by the clock the chip holds under this load, not by the schedule.
// Round 6 (profiles/r06_conv_schedules.log; source: commit fc740de, `PIPE` 1-3 of this body, all bit-identical to it):
//  * the hand-over's two waits stamped apart: `s_waitcnt vmcnt(0)` costs 48 of a step's 4 545 cycles -- the next stage HAS landed when a wave
//    gets there; a deeper ring (weights two steps ahead, five half-stages in 160 KiB) has nothing to hide.  The `s_barrier` is where wave 0
//    sits, 1 780 cycles per step, waiting for wave 4, its SIMD partner (HW_ID says so in 100 % of the tiles): the arbiter serves the older
//    wave's MFMAs and LDS-DMA first (DMA issue: 470 cycles in wave 0, 1 180 in wave 4), so the younger one runs the end of every step alone,
//    every LDS round trip exposed.
//  * PIPE 1: hand-over between a step's LDS reads (head: hi*hi, hi*lo) and its register-only tail (lo*hi), the DMA of stage s + 2 issued
//    right behind barrier s -- a full step in flight;  PIPE 2: the same with waves 4-7 multiplying their tail BEFORE they issue;  PIPE 3:
//    waves 0-3 stage everything (16 instructions each), waves 4-7 nothing.  Stamped twins: 4 545 -> 4 404 / 3 941 / 4 065 cycles per step,
//    the in-kernel clock 2.38 -> 2.38 / 2.28 / 2.30 GHz; PRODUCT kernels, layer alone on one box: 1.573-1.579 (this loop) vs 1.639 / 1.587-1.592 /
//    1.592 ms.  The compiler's schedule of this loop already sinks half of a step's MFMAs below the barrier, which is most of what the
//    hand-written orders buy; what they save in cycles beyond that the clock gives back (13 % fewer cycles, 4 % less clock, 0 % less time).
//  * the centre-offset fold priced (tuning bit 8): with the centre offset's partial rows neither written nor read the twin's layer takes
//    1.597 instead of 1.635 ms -- 2.3 % is the MOST a fold could return, before its own costs (a 128-row x 512-column tile per workgroup or a
//    row-maximum exchange between the two column tiles): not built.
//  * ONE wave per SIMD (source: commit 501a632, `conv_phase1_w4_kernel`): four waves x 128 x 128 with the 256 accumulator registers in AGPRs,
//    every LDS byte read by two waves instead of four, no SIMD partner to wait for at the barrier; inline-asm MFMAs and LDS-DMA, the K loop
//    scheduled by hand (B fragments one column tile ahead, next step's A fragments and the DMA of stage s + 2 between the lo*hi MFMAs).
//    Compiler-scheduled around asm MFMAs: 1.869 ms, bit-identical; hand-scheduled, no spills in the loop: 1.613 ms vs 1.546 ms for this body
//    on the same box, and not bit-identical (nothing pads the MFMA -> `v_accvgpr` / store hazards the compiler cannot see behind asm; the
//    waits that would fix it only add time).  With one wave per SIMD nothing hides an LDS round trip or a DMA issue stall.  Not kept.
template <bool TUNE, bool STAMP>
__device__ __forceinline__ void
conv_phase1_dma_body(const _Float16 *__restrict__ x_hi, const _Float16 *__restrict__ x_lo, int64_t ld_xh,
                     const int32_t *__restrict__ pair_in, const int32_t *__restrict__ off,
                     const int32_t *__restrict__ tile_start, const int4 *__restrict__ tile_desc, int nseg, int kv,
                     const _Float16 *__restrict__ w_hi, const _Float16 *__restrict__ w_lo, int cin, int cout,
                     float *__restrict__ P, int n_tiles, int ablate_, int tile_begin, int tile_count, int pair_base,
                     const float *__restrict__ x_inv_scale, uint64_t *__restrict__ stamp, int64_t q_e_off, int w_blocked, int x_il) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int ablate = TUNE ? ablate_ : 0;
    uint64_t st_t0 = 0, st_r0 = 0, st_pro = 0, st_loop = 0, st_iss = 0, st_dma = 0, st_wait = 0, st_bar = 0;
    if constexpr (STAMP) { st_t0 = cv_now(); st_r0 = cv_real(); }
    V2Smem &sm = *reinterpret_cast<V2Smem *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-contiguous tile order: blocks b, b+8, ... share an XCD; each XCD walks a contiguous range of
    // (m-tile, n-tile) pairs, both channel tiles of an m-tile back to back (shared A rows hit its L2)
    const int64_t nb = gridDim.x, per_xcd = nb >> 3;
    const int64_t lb = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int nt = (int)(lb % n_tiles);
    const int mt_local = (int)(lb / n_tiles);
    if (mt_local >= tile_count) return;
    const int mt = tile_begin + mt_local;
    if (mt >= tile_start[nseg]) return;
    const int4 td = tile_desc[mt];
    const int k = td.x, base = td.y, cnt = td.z;
    const int n0 = nt * TN;
    // DMA roles: every wave stages RPW = 32 rows of each array, NI = 2 instructions of 16 rows per plane.  The lo planes sit at a
    // uniform distance from the hi planes.
    constexpr int NI = 2, RPW = 16 * NI;
    constexpr bool issuer = true;
    const int lrow = lane >> 2, lp = lane & 3;
    const int q = (lp ^ ((0x78 >> (((lane >> 4) & 3) * 2)) & 3)) * 8;         // logical 8-half slot this lane fetches
    const int64_t da = x_lo - x_hi, db = w_lo - w_hi;                         // (in halfs)
    // A operand, two forms.  Planes (x_hi, x_lo separate, rows of 512 halfs): an instruction stages 16 rows x 64 bytes of one plane --
    // 16 half lines.  INTERLEAVED rows (x_il: [K step][hi 32 | lo 32], what phase 2 writes for the next layer): an instruction stages
    // 8 rows x 128 bytes -- FULL lines, hi and lo of a row and step in one request (MI355X guide: fragment-shaped 16 x 64-byte loads
    // cost 2 x the address-path time of full-line pieces at the same traffic) -- into one image of 128-byte rows whose 16-byte slot p
    // of row r holds logical slot p ^ (r >> 1 & 7) (0-3: hi k-groups, 4-7: lo): the four ds_read_b128 lane groups of a fragment
    // read then touch 16 different slots of the 256-byte bank row.
    const _Float16 *ga[4], *gb_hi[NI];
    uint32_t la[4];                                                            // LDS byte offset of the instruction's 1 KiB, ring slot 0
    int in_rows[4];
    const uint32_t a_bufstride = x_il ? 2u * TM * APITCH * 2u : (uint32_t)TM * APITCH * 2u;     // 32 KiB | 16 KiB
    const int amul = x_il ? 2 : 1;
    if (x_il) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wv * RPW + i * 8 + (lane >> 3);
            const int in_row = pair_in[base + (row < cnt ? row : cnt - 1)];
            in_rows[i] = in_row;
            const int L = (lane & 7) ^ ((row >> 1) & 7);
            ga[i] = x_hi + (int64_t)in_row * ld_xh + L * 8;
            la[i] = (uint32_t)((wv * RPW + i * 8) * 128);
        }
    } else {
#pragma unroll
        for (int t = 0; t < NI; ++t) {
            const int row = wv * RPW + t * 16 + lrow;
            const int in_row = pair_in[base + (row < cnt ? row : cnt - 1)];       // clamped, unconditional: both loads overlap
            in_rows[2 * t] = in_rows[2 * t + 1] = in_row;
            ga[2 * t] = x_hi + (int64_t)in_row * ld_xh + q;
            ga[2 * t + 1] = ga[2 * t] + da;
            la[2 * t] = (uint32_t)((wv * RPW + t * 16) * APITCH * 2);
            la[2 * t + 1] = la[2 * t] + 2u * TM * APITCH * 2u;                  // a_lo follows the two slots of a_hi
        }
    }
#pragma unroll
    for (int t = 0; t < NI; ++t) {
        int row = (issuer ? wv : 0) * RPW + t * 16 + lrow;
        // LDS row j * 16 + f of a wave's 128 weight rows (column tile j, MFMA column f) holds OUTPUT COLUMN f * 8 + j: a lane's eight
        // accumulators of a row are then eight consecutive columns of the partial row (one 16-byte and one 8-byte store per row in
        // the 24-bit format below), and the permutation costs nothing: it is the source address of the LDS-DMA
        const int wcol = (row & 128) | ((row & 15) << 3) | ((row >> 4) & 7);
        gb_hi[t] = w_hi + ((int64_t)k * cout + n0 + wcol) * cin + q;
        // step-blocked weights (gp_conv_weights_split_blocked: [offset][column tile][K step][256 LDS rows][32]): a step's 16 KiB are
        // contiguous and every LDS-DMA instruction of the weight operand is ONE 1-KiB run instead of 16 half lines (a quarter fewer L2
        // requests in the K loop); the row permutation above is part of that layout
        if (w_blocked) gb_hi[t] = w_hi + ((((int64_t)k * n_tiles + nt) * (cin / TK)) * TN + row) * TK + q;
    }
    const int bmul = w_blocked ? TN : 1;                      // halfs of the weight operand between two K steps, over TK
    auto issue = [&](int c0, int buf) {
        if (!issuer) return;
        const uint32_t ab = (uint32_t)buf * a_bufstride;
#pragma unroll
        for (int t = 0; t < NI; ++t) {
            const int r0 = wv * RPW + t * 16;
            glds16(ga[2 * t] + c0 * amul, smem_raw + la[2 * t] + ab);
            glds16(ga[2 * t + 1] + c0 * amul, smem_raw + la[2 * t + 1] + ab);
            glds16(gb_hi[t] + c0 * bmul, &sm.b_hi[buf][r0][0]);
            glds16(gb_hi[t] + db + c0 * bmul, &sm.b_lo[buf][r0][0]);
        }
    };

    const int wm = wv >> 1, wn = wv & 1;
    const int fl = lane & 15, fq = lane >> 4;
    const int fsw = sw_slot(fl, fq) * 8;
    uint32_t fa_hi[4], fa_lo[4];                               // this lane's A fragment reads (byte offsets, ring slot 0)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wm * 64 + i * 16 + fl;
        if (x_il) {
            const int hsw = (row >> 1) & 7;
            fa_hi[i] = (uint32_t)(row * 128 + ((fq ^ hsw) << 4));
            fa_lo[i] = (uint32_t)(row * 128 + (((4 + fq) ^ hsw) << 4));
        } else {
            fa_hi[i] = (uint32_t)((row * APITCH + fsw) * 2);
            fa_lo[i] = fa_hi[i] + 2u * TM * APITCH * 2u;
        }
    }
    const int rows_here = cnt - wm * 64;
    const int nrt = __builtin_amdgcn_readfirstlane(rows_here <= 0 ? 0 : (rows_here >= 64 ? 4 : (rows_here + 15) >> 4));

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int steps = cin / TK;
    issue(0, 0);
    // per-row power-of-two scales of the pre-split operand (x_hi + x_lo = x * 2^e(row), gp_split_f16_scaled): the partial
    // row of a pair is multiplied back by 2^-e(input row).  Used in the epilogue only, and there as 16 values per lane: the lanes
    // that already hold a staged row's input id (four per row: one stores) fetch its scale -- ONE dependent load per lane, issued
    // behind the first stage's DMA -- and park it in LDS behind the ring; the epilogue reads its 16 from there.  (Rounds 4-5 gathered
    // row id + scale per lane and (i, r): 32 loads per lane in front of the second stage's DMA, 16 registers live through the loop.)
    float *s_rinv = reinterpret_cast<float *>(smem_raw + sizeof(V2Smem));
    if (x_il) {
        float rv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) rv[i] = x_inv_scale ? x_inv_scale[in_rows[i]] : 1.f;
        if ((lane & 7) == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) s_rinv[wv * RPW + i * 8 + (lane >> 3)] = rv[i];
        }
    } else {
        float rv[NI];
#pragma unroll
        for (int t = 0; t < NI; ++t) rv[t] = x_inv_scale ? x_inv_scale[in_rows[2 * t]] : 1.f;
        if (lp == 0) {
#pragma unroll
            for (int t = 0; t < NI; ++t) s_rinv[wv * RPW + t * 16 + lrow] = rv[t];
        }
    }
    __syncthreads();
    if constexpr (STAMP) st_pro = cv_now();
    // the last tile of a (chunk, offset) segment is partly filled (8192-row chunks: 1 tile in 9, a third full on average): a
    // wave multiplies only the 16-row tiles that hold pairs -- matrix work the chip's power budget does not have to pay for.
    // nrt is wave-uniform and fixed for the tile: one copy of the loop per value (a switch INSIDE the loop costs 98 spills).
    auto k_loop = [&](auto nrt_c) {
        constexpr int NRT = decltype(nrt_c)::value;
        for (int s = 0; s < steps; ++s) {
            const int buf = s & 1;
            uint64_t st_a = 0;
            if constexpr (STAMP) st_a = cv_now();
            if (s + 1 < steps) issue((s + 1) * TK, buf ^ 1);
            if constexpr (STAMP) st_dma += cv_now() - st_a;
            if constexpr (NRT > 0)
                if (!(ablate & 2)) mma_step_f16x3_off<NRT>(smem_raw, fa_hi, fa_lo, (uint32_t)buf * a_bufstride, sm, buf, wn, fl, fsw, acc);
            if constexpr (STAMP) {                            // the hand-over's two waits timed apart
                st_a = cv_now();
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                const uint64_t st_b = cv_now();
                asm volatile("s_barrier" ::: "memory");
                st_wait += st_b - st_a;
                st_bar += cv_now() - st_b;
            } else {
                __syncthreads();
            }
        }
    };
    if (nrt == 4) k_loop(std::integral_constant<int, 4>{});
    else if (nrt == 3) k_loop(std::integral_constant<int, 3>{});
    else if (nrt == 2) k_loop(std::integral_constant<int, 2>{});
    else if (nrt == 1) k_loop(std::integral_constant<int, 1>{});
    else k_loop(std::integral_constant<int, 0>{});
    if constexpr (STAMP) st_loop = cv_now();
    auto stamp_out = [&]() {
        if constexpr (STAMP) {
            st_iss = cv_now();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint64_t t3 = cv_now(), r3 = cv_real();
            if ((tid == 0 || tid == 256) && stamp) {
                uint64_t *o = stamp + (int64_t)blockIdx.x * 16;
                unsigned xcc, hwid;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
                if (tid == 0) {
                    o[0] = st_r0; o[1] = r3 - st_r0; o[2] = st_pro - st_t0; o[3] = st_loop - st_pro; o[4] = st_iss - st_loop;
                    o[5] = t3 - st_iss; o[6] = t3 - st_t0; o[7] = (uint64_t)(xcc & 0xff) | ((uint64_t)cnt << 8);
                    o[8] = st_dma; o[9] = st_wait; o[10] = st_bar; o[11] = hwid;
                } else {                                  // wave 4: the SIMD partner of wave 0 (its HW_ID says so)
                    o[12] = st_dma; o[13] = st_wait; o[14] = st_bar; o[15] = hwid;
                }
            }
        }
    };
    if (ablate & 8) { stamp_out(); return; }
    // (tuning bit 8 of knob 3, 256: the tiles of the CENTRE offset -- the identity map, 13.5 % of the pairs -- store nothing, and phase 2's
    // twin does not read their rows: the price of those partial rows' round trip, i.e. the most a fold of the centre offset into
    // phase 2 could return)
    if ((ablate & 256) && k == (kv >> 1)) { stamp_out(); return; }
    // ---- epilogue: accumulators straight to the partial buffer (no LDS staging: that made every slice's LDS reads wait for the
    //      previous slice's stores -- one vector-memory counter -- 4 store round trips per tile, a third of the kernel's time).
    // PARTIAL ROWS ARE STORED AS 24-BIT BLOCK FLOATING POINT (round 5): per pair row and 128-column quarter (= what one wave owns) an
    // exponent byte E and per element u = rint(v * 2^(148 - E)) + 2^22 in three bytes, |v * 2^(148 - E)| < 2^22 for every element of
    // the quarter -- 3 + 1/128 bytes instead of 4 per element of the round trip phase 1 -> Infinity Cache -> phase 2 (2 x 2.0 GB per
    // 512 -> 512 layer in fp32).  Encoding costs ONE fused multiply-add per element: t = fma(acc, s, 1.5 * 2^23) lies in [2^23, 2^24),
    // where floats are the integers, so the hardware's round-to-nearest-even IS the quantisation and the three low bytes of t's bit
    // pattern ARE u; eight elements = 24 bytes per lane, packed by six byte permutes.  The rounding is half a unit of 2^(E - 148): between
    // 2^-23 and 2^-22 of the quarter's largest magnitude per partial row -- no more than the NEXT rounding on the path (phase 2 splits
    // the summed row into f16 hi + lo at 2^-22 of the row's maximum).  E = 255 marks a quarter with an Inf or a NaN activation row (phase 2 writes NaN).
    unsigned char *pb = reinterpret_cast<unsigned char *>(P);
    const bool as_f32 = TUNE && (ablate & 32);               // tuning twin: the fp32 rows of rounds 1-4 (host pairs them with conv_phase2_kernel)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < nrt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = wm * 64 + i * 16 + fq * 4 + r;
                const float sc = s_rinv[grow];
                const int64_t prow = base - pair_base + grow;
                const int col = n0 + wn * 128 + fl * 8;
                if (as_f32) {
                    if (grow < cnt) {
                        float *dst = P + prow * cout + col;
                        *reinterpret_cast<float4 *>(dst) = make_float4(acc[i][0][r] * sc, acc[i][1][r] * sc, acc[i][2][r] * sc, acc[i][3][r] * sc);
                        *reinterpret_cast<float4 *>(dst + 4) = make_float4(acc[i][4][r] * sc, acc[i][5][r] * sc, acc[i][6][r] * sc, acc[i][7][r] * sc);
                    }
                    continue;
                }
                // the quarter's largest magnitude: 8 accumulators, then the 16 lanes that share fq -- as BIT PATTERNS (non-negative floats
                // order like unsigned integers), so that each step is one v_max_u32 with a DPP operand (two quad permutes, two row
                // rotations; no LDS round trip, no canonicalising moves); a NaN row (NaN x anything: every column of the row is NaN)
                // enters as +Inf
                float m = fmaxf(fmaxf(fabsf(acc[i][0][r]), fabsf(acc[i][1][r])), fabsf(acc[i][2][r]));
                m = fmaxf(fmaxf(m, fabsf(acc[i][3][r])), fabsf(acc[i][4][r]));
                m = fmaxf(fmaxf(m, fabsf(acc[i][5][r])), fabsf(acc[i][6][r]));
                m = fmaxf(m, fabsf(acc[i][7][r]));
                unsigned mi = (acc[i][0][r] != acc[i][0][r]) ? 0x7f800000u : __float_as_uint(m);
                mi = max(mi, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mi, 0xb1, 0xf, 0xf, true));     // quad_perm [1,0,3,2]
                mi = max(mi, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mi, 0x4e, 0xf, 0xf, true));     // quad_perm [2,3,0,1]
                mi = max(mi, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mi, 0x124, 0xf, 0xf, true));    // row_ror:4
                mi = max(mi, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mi, 0x128, 0xf, 0xf, true));    // row_ror:8
                // exponent field of the raw maximum; + 1 in the last place first: a maximum with an all-ones mantissa would round up
                // to 2^22 -- it takes the next exponent.  The accumulators are multiplied by 2^(148 - Em) (field 275 - Em, kept a
                // finite float: below 2^-106 everything rounds to 0 anyway); the STORED exponent carries the row's 2^e(sc) as well:
                // decoded value = (u - 2^22) 2^(E - 148) = acc * sc.  E outside [22, 254]: the value is below 2^-104 (kept, scaled
                // wrongly by a power of two: it is nothing) or beyond fp32 (E = 255: phase 2 writes NaN where fp32 had Inf)
                const unsigned Em = (mi + 1u) >> 23;
                const unsigned fld = 275u - Em;
                const float s = __uint_as_float((fld > 254u ? 254u : fld) << 23);
                int Es = (int)Em + ((int)(__float_as_uint(sc) >> 23) - 127);
                Es = Es < 22 ? 22 : Es;
                const unsigned E = (Em == 255u || Es > 254) ? 255u : (unsigned)Es;
                unsigned t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = __float_as_uint(fmaf(acc[i][j][r], s, 12582912.f));
                if (grow < cnt) {
                    u32x4 h;
                    u32x2 l;
                    h[0] = __builtin_amdgcn_perm(t[1], t[0], 0x04020100u);
                    h[1] = __builtin_amdgcn_perm(t[2], t[1], 0x05040201u);
                    h[2] = __builtin_amdgcn_perm(t[3], t[2], 0x06050402u);
                    h[3] = __builtin_amdgcn_perm(t[5], t[4], 0x04020100u);
                    l[0] = __builtin_amdgcn_perm(t[6], t[5], 0x05040201u);
                    l[1] = __builtin_amdgcn_perm(t[7], t[6], 0x06050402u);
                    // a quarter's 384 bytes: the 16 lanes' first 16 bytes (elements 0-4 and a third of 5), then their last 8 --
                    // every store naturally aligned, 256- and 128-byte runs (below 4 GiB: checked on the host)
                    unsigned char *dst = pb + ((unsigned)prow * (unsigned)cout + (unsigned)(col & ~127)) * 3u;
                    *reinterpret_cast<u32x4 *>(dst + fl * 16) = h;
                    *reinterpret_cast<u32x2 *>(dst + 256 + fl * 8) = l;
                    if (fl == 0) (pb + q_e_off)[(unsigned)prow * (unsigned)(cout >> 7) + (unsigned)(col >> 7)] = (unsigned char)E;
                }
            }
        }
    }
    stamp_out();
}

#define P1_PARAMS const _Float16 *__restrict__ x_hi, const _Float16 *__restrict__ x_lo, int64_t ld_xh, const int32_t *__restrict__ pair_in, \
                  const int32_t *__restrict__ off, const int32_t *__restrict__ tile_start, const int4 *__restrict__ tile_desc, int nseg,  \
                  int kv, const _Float16 *__restrict__ w_hi, const _Float16 *__restrict__ w_lo, int cin, int cout, float *__restrict__ P, \
                  int n_tiles, int ablate, int tile_begin, int tile_count, int pair_base, const float *__restrict__ x_inv_scale,           \
                  uint64_t *__restrict__ stamp, int64_t q_e_off, int w_blocked, int x_il
#define P1_FWD x_hi, x_lo, ld_xh, pair_in, off, tile_start, tile_desc, nseg, kv, w_hi, w_lo, cin, cout, P, n_tiles, ablate, tile_begin, tile_count, pair_base, x_inv_scale, stamp, q_e_off, w_blocked, x_il
// the product kernel (tuning bits compiled out) and its twin with the bits of knob 3 live, under its own name in a trace
// (bench.py's data-movement ceiling of the convolution and scripts/bench_conv.py's ablations launch the twin)
__global__ void __launch_bounds__(NT2) conv_phase1_dma_kernel(P1_PARAMS) { conv_phase1_dma_body<false, false>(P1_FWD); }
__global__ void __launch_bounds__(NT2) conv_phase1_tuning_kernel(P1_PARAMS) { conv_phase1_dma_body<true, false>(P1_FWD); }
__global__ void __launch_bounds__(NT2) conv_phase1_stamp_kernel(P1_PARAMS) { conv_phase1_dma_body<true, true>(P1_FWD); }

#undef P1_PARAMS
#undef P1_FWD

// sum of a voxel's partial rows in ASCENDING offset order (bitwise reproducible), NL independent loads in flight per round
// (a voxel has 7.3 partial rows on average: one round of 8 for most voxels)
template <int NL>
__device__ __forceinline__ float4 conv_gather_sum(const float *__restrict__ P, int mypos, int kv, int cout, int c, int pair_base) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned long long m = __ballot(mypos >= 0) & ((kv >= 64) ? ~0ull : ((1ull << kv) - 1ull));
    while (m) {
        int kk[NL];
        float4 t[NL];
        int cntv = 0;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            if (m) { kk[i] = __builtin_ctzll(m); m &= m - 1; ++cntv; } else kk[i] = -1;
        }
#pragma unroll
        for (int i = 0; i < NL; ++i)
            if (i < cntv) {
                const int pos = __shfl(mypos, kk[i], 64);
                t[i] = *reinterpret_cast<const float4 *>(P + (int64_t)(pos - pair_base) * cout + c);
            }
#pragma unroll
        for (int i = 0; i < NL; ++i)
            if (i < cntv) { a.x += t[i].x; a.y += t[i].y; a.z += t[i].z; a.w += t[i].w; }
    }
    return a;
}

constexpr int GS_NL = 8;
// the same sum for two column blocks (c and c + 256) of the same partial rows at once: 2 NL loads in flight per round
template <int NL>
__device__ __forceinline__ void conv_gather_sum2(const float *__restrict__ P, int mypos, int kv, int cout, int c, int pair_base,
                                                 float4 &a0, float4 &a1) {
    a0 = make_float4(0.f, 0.f, 0.f, 0.f);
    a1 = a0;
    unsigned long long m = __ballot(mypos >= 0) & ((kv >= 64) ? ~0ull : ((1ull << kv) - 1ull));
    while (m) {
        int kk[NL];
        float4 t0[NL], t1[NL];
        int cntv = 0;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            if (m) { kk[i] = __builtin_ctzll(m); m &= m - 1; ++cntv; } else kk[i] = -1;
        }
#pragma unroll
        for (int i = 0; i < NL; ++i)
            if (i < cntv) {
                const int pos = __shfl(mypos, kk[i], 64);
                const float *row = P + (int64_t)(pos - pair_base) * cout + c;
                t0[i] = *reinterpret_cast<const float4 *>(row);
                t1[i] = *reinterpret_cast<const float4 *>(row + 256);
            }
#pragma unroll
        for (int i = 0; i < NL; ++i)
            if (i < cntv) {
                a0.x += t0[i].x; a0.y += t0[i].y; a0.z += t0[i].z; a0.w += t0[i].w;
                a1.x += t1[i].x; a1.y += t1[i].y; a1.z += t1[i].z; a1.w += t1[i].w;
            }
    }
}
// phase 2: one wave per output voxel; lanes hold 2 x float4 of the 512 (or cout) channels
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) conv_phase2_kernel(const float *__restrict__ P, const int32_t *__restrict__ pair_pos, int64_t nv, int kv,
                                   int cout, const float *__restrict__ scale, const float *__restrict__ shift,
                                   const float *__restrict__ residual, int64_t ld_res, int relu,
                                   float *__restrict__ y, int64_t ld_y, _Float16 *__restrict__ y_hi,
                                   _Float16 *__restrict__ y_lo, int64_t ld_yh, int64_t row_begin, int64_t row_count,
                                   int pair_base, float *__restrict__ y_inv_scale, const _Float16 *__restrict__ res_hi,
                                   const _Float16 *__restrict__ res_lo, int64_t ld_rh, const float *__restrict__ res_inv, int plane_flags) {
    const bool y_il = (plane_flags & 2) != 0, r_il = (plane_flags & 4) != 0;      // (see conv_phase2_q24_kernel)
    // (res_hi / res_lo / res_inv: the residual as the split planes an earlier layer wrote -- (hi + lo) * res_inv[row] -- instead of fp32 rows)
    // Waves walk the chunk's rows with a stride of the whole grid (row w, w + W, ...), the NEXT row's 27 positions loaded while this
    // row's partial rows are gathered: the host sizes the grid to what is resident at once (6 waves per SIMD at 80 registers), so
    // that a chunk is not one full round of waves plus a third of one, and the position -> rows dependency is paid once per wave.
    // (Measured and left out: the positions row-major, [nv][32], one 128-byte line per row instead of 27 sectors -- 1.890 vs 1.893 ms
    // per layer: neighbouring waves share the sectors and the prefetch hides their latency.)
    const int lane = gp_lane();
    const int64_t wave0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t row_end = (row_begin + row_count < nv) ? row_begin + row_count : nv;
    int64_t u = row_begin + wave0;
    if (u >= row_end) return;
    int mypos_next = (lane < kv) ? pair_pos[(int64_t)lane * nv + u] : -1;
    for (; u < row_end; u += n_waves) {
    const int mypos = mypos_next;
    if (u + n_waves < row_end) mypos_next = (lane < kv) ? pair_pos[(int64_t)lane * nv + u + n_waves] : -1;
    if (y_inv_scale && y_hi && cout <= 1024) {
        // pre-split output with a per-row power of two: hi + lo = y * 2^e, 2^e chosen so that the row's largest magnitude
        // lands in [2^13, 2^14) -- every element within 2^-18 of it keeps a NORMAL f16 lo half, i.e. the full 2^-22 relative
        // accuracy of the split; the next layer's phase 1 multiplies its partial rows by y_inv_scale[row] = 2^-e (exact)
        float4 av[4];
        float amax = 0.f;
        const bool pair512 = cout == 512;                    // the usual width: both column blocks of a row gathered together
        if (pair512) conv_gather_sum2<GS_NL>(P, mypos, kv, cout, lane * 4, pair_base, av[0], av[1]);
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            const int c = lane * 4 + ch * 256;
            if (c < cout) {
                float4 a = pair512 ? av[ch & 1] : conv_gather_sum<GS_NL>(P, mypos, kv, cout, c, pair_base);
                float4 sc = scale ? *reinterpret_cast<const float4 *>(scale + c) : make_float4(1.f, 1.f, 1.f, 1.f);
                float4 sh = shift ? *reinterpret_cast<const float4 *>(shift + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                a.x = a.x * sc.x + sh.x; a.y = a.y * sc.y + sh.y; a.z = a.z * sc.z + sh.z; a.w = a.w * sc.w + sh.w;
                if (residual) {
                    float4 r = *reinterpret_cast<const float4 *>(residual + u * ld_res + c);
                    a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w;
                }
                if (res_hi) {
                    typedef _Float16 f16x4r __attribute__((ext_vector_type(4)));
                    const _Float16 *rph = r_il ? res_hi + u * ld_rh + ((c >> 5) << 6) + (c & 31) : res_hi + u * ld_rh + c;
                    const _Float16 *rpl = r_il ? rph + 32 : res_lo + u * ld_rh + c;
                    const f16x4r rh = *reinterpret_cast<const f16x4r *>(rph), rl = *reinterpret_cast<const f16x4r *>(rpl);
                    const float ri = res_inv ? res_inv[u] : 1.f;
                    a.x += ((float)rh[0] + (float)rl[0]) * ri; a.y += ((float)rh[1] + (float)rl[1]) * ri;
                    a.z += ((float)rh[2] + (float)rl[2]) * ri; a.w += ((float)rh[3] + (float)rl[3]) * ri;
                }
                if (relu) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
                if (y) *reinterpret_cast<float4 *>(y + u * ld_y + c) = a;      // fp32 copy only where a later layer reads it (residual, linear)
                av[ch] = a;
                amax = fmaxf(amax, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
            }
        }
        amax = gp_wave_max(amax);
        const float s = gp_pow2_for(amax);
        if (lane == 0) y_inv_scale[u] = 1.f / s;
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            const int c = lane * 4 + ch * 256;
            if (c < cout) {
                float v[4] = {av[ch].x * s, av[ch].y * s, av[ch].z * s, av[ch].w * s};
                f16x4 h, l;
#pragma unroll
                for (int i = 0; i < 4; ++i) { h[i] = (_Float16)v[i]; l[i] = (_Float16)(v[i] - (float)h[i]); }
                _Float16 *ph = y_il ? y_hi + u * ld_yh + ((c >> 5) << 6) + (c & 31) : y_hi + u * ld_yh + c;
                _Float16 *pl = y_il ? ph + 32 : y_lo + u * ld_yh + c;
                *reinterpret_cast<f16x4 *>(ph) = h;
                *reinterpret_cast<f16x4 *>(pl) = l;
            }
        }
        continue;
    }
    for (int c = lane * 4; c < cout; c += 256) {
        float4 a = conv_gather_sum<GS_NL>(P, mypos, kv, cout, c, pair_base);
        float4 sc = scale ? *reinterpret_cast<const float4 *>(scale + c) : make_float4(1.f, 1.f, 1.f, 1.f);
        float4 sh = shift ? *reinterpret_cast<const float4 *>(shift + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        a.x = a.x * sc.x + sh.x; a.y = a.y * sc.y + sh.y; a.z = a.z * sc.z + sh.z; a.w = a.w * sc.w + sh.w;
        if (residual) {
            float4 r = *reinterpret_cast<const float4 *>(residual + u * ld_res + c);
            a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w;
        }
        if (res_hi) {
            typedef _Float16 f16x4r __attribute__((ext_vector_type(4)));
            const _Float16 *rph = r_il ? res_hi + u * ld_rh + ((c >> 5) << 6) + (c & 31) : res_hi + u * ld_rh + c;
                    const _Float16 *rpl = r_il ? rph + 32 : res_lo + u * ld_rh + c;
                    const f16x4r rh = *reinterpret_cast<const f16x4r *>(rph), rl = *reinterpret_cast<const f16x4r *>(rpl);
            const float ri = res_inv ? res_inv[u] : 1.f;
            a.x += ((float)rh[0] + (float)rl[0]) * ri; a.y += ((float)rh[1] + (float)rl[1]) * ri;
            a.z += ((float)rh[2] + (float)rl[2]) * ri; a.w += ((float)rh[3] + (float)rl[3]) * ri;
        }
        if (relu) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
        if (y) *reinterpret_cast<float4 *>(y + u * ld_y + c) = a;
        if (y_hi) {                                           // pre-split operand of the next layer
            typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
            float v[4] = {a.x, a.y, a.z, a.w};
            f16x4 h, l;
#pragma unroll
            for (int i = 0; i < 4; ++i) { h[i] = (_Float16)v[i]; l[i] = (_Float16)(v[i] - (float)h[i]); }
            _Float16 *ph = y_il ? y_hi + u * ld_yh + ((c >> 5) << 6) + (c & 31) : y_hi + u * ld_yh + c;
            _Float16 *pl = y_il ? ph + 32 : y_lo + u * ld_yh + c;
            *reinterpret_cast<f16x4 *>(ph) = h;
            *reinterpret_cast<f16x4 *>(pl) = l;
        }
    }
    }   // rows of this wave
}

// ------------------------------------------------------------------------------------------------
// phase 2 over 24-bit block-floating partial rows (the format of conv_phase1_dma_body's epilogue): a lane owns EIGHT consecutive
// columns (c = lane * 8 + 512 it) = 24 bytes of a partial row (a quarter's 384 bytes hold its 16 lanes' first 16 bytes, then their
// last 8: one aligned 16-byte and one 8-byte load) plus the quarter's exponent byte (one address per 16 lanes); an element is (u - 2^22) * 2^(E - 148), exact in fp32, and the sum runs in ASCENDING offset order as
// before (bitwise reproducible).
template <int NL>
__device__ __forceinline__ void conv_gather_sum_q24(const unsigned char *__restrict__ pb, int64_t e_off, int mypos, int kv,
                                                    int cout, int c, bool act, int pair_base, float (&a)[8]) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.f;
    unsigned long long m = __ballot(mypos >= 0) & ((kv >= 64) ? ~0ull : ((1ull << kv) - 1ull));
    const int nq = cout >> 7;
    const unsigned char *pb_e = pb + e_off;
    while (m) {
        int kk[NL];
        u32x4 th[NL];
        u32x2 tl[NL];
        unsigned te[NL];
        int cntv = 0;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            if (m) { kk[i] = __builtin_ctzll(m); m &= m - 1; ++cntv; } else kk[i] = -1;
        }
#pragma unroll
        for (int i = 0; i < NL; ++i)
            if (i < cntv) {
                // 32-bit byte offsets from uniform bases (the host checks that a chunk's rows stay below 4 GiB): one address register
                // per load instead of two
                const unsigned pos = (unsigned)(__shfl(mypos, kk[i], 64) - pair_base);
                if (act) {
                    const unsigned o = (pos * (unsigned)cout + ((unsigned)c & ~127u)) * 3u, f = ((unsigned)c >> 3) & 15u;   // quarter, lane in it
                    th[i] = *reinterpret_cast<const u32x4 *>(pb + o + f * 16u);
                    tl[i] = *reinterpret_cast<const u32x2 *>(pb + o + 256u + f * 8u);
                    te[i] = pb_e[pos * (unsigned)nq + ((unsigned)c >> 7)];
                }
            }
#pragma unroll
        for (int i = 0; i < NL; ++i)
            if (i < cntv && act) {
                const float s = te[i] == 255u ? __uint_as_float(0x7fc00000u) : __uint_as_float((te[i] - 21u) << 23);   // 2^(E - 148)
                const unsigned d[6] = {th[i][0], th[i][1], th[i][2], th[i][3], tl[i][0], tl[i][1]};
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const unsigned u0 = __builtin_amdgcn_perm(0u, d[g * 3], 0x0c020100u);
                    const unsigned u1 = __builtin_amdgcn_perm(d[g * 3 + 1], d[g * 3], 0x0c050403u);
                    const unsigned u2 = __builtin_amdgcn_perm(d[g * 3 + 2], d[g * 3 + 1], 0x0c040302u);
                    const unsigned u3 = __builtin_amdgcn_perm(0u, d[g * 3 + 2], 0x0c030201u);
                    a[g * 4 + 0] = fmaf((float)u0 - 4194304.f, s, a[g * 4 + 0]);
                    a[g * 4 + 1] = fmaf((float)u1 - 4194304.f, s, a[g * 4 + 1]);
                    a[g * 4 + 2] = fmaf((float)u2 - 4194304.f, s, a[g * 4 + 2]);
                    a[g * 4 + 3] = fmaf((float)u3 - 4194304.f, s, a[g * 4 + 3]);
                }
            }
    }
}

template <bool WIDE /* cout > 512: a second register set for columns 512 .. */, bool NO_CENTRE = false /* tuning twin: see knob 3, 256 */>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
conv_phase2_q24_kernel(const unsigned char *__restrict__ pb, int64_t e_off, const int32_t *__restrict__ pair_pos, int64_t nv,
                       int kv, int cout, const float *__restrict__ scale, const float *__restrict__ shift,
                       const float *__restrict__ residual, int64_t ld_res, int relu, float *__restrict__ y, int64_t ld_y,
                       _Float16 *__restrict__ y_hi, _Float16 *__restrict__ y_lo, int64_t ld_yh, int64_t row_begin, int64_t row_count,
                       int pair_base, float *__restrict__ y_inv_scale, const _Float16 *__restrict__ res_hi,
                       const _Float16 *__restrict__ res_lo, int64_t ld_rh, const float *__restrict__ res_inv, int plane_flags) {
    // plane_flags bit 1: y_hi is ONE tensor of interleaved rows [K step][hi 32 | lo 32] (the next layer's full-line operand; y_lo
    // unused); bit 2: the residual planes come in that form.  A lane's eight columns lie inside one 32-column step.
    const bool y_il = (plane_flags & 2) != 0, r_il = (plane_flags & 4) != 0;
    // (res_hi / res_lo / res_inv: the residual read from the split planes an earlier layer wrote for ITS consumer -- (hi + lo) * res_inv[row],
    //  the value that layer's successor multiplied with -- so that no fp32 copy of a block's input is written only to be added once)
    // the row walk of conv_phase2_kernel: rows w, w + W, ... per wave, the next row's 27 positions loaded under this row's gathers
    const int lane = gp_lane();
    const int64_t wave0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t row_end = (row_begin + row_count < nv) ? row_begin + row_count : nv;
    int64_t u = row_begin + wave0;
    if (u >= row_end) return;
    const bool rowscale = y_inv_scale && y_hi && cout <= 1024;
    const int kv_read = NO_CENTRE ? (kv >> 1) : -1;          // (twin: the lane of the centre offset reads nothing)
    int mypos_next = (lane < kv && lane != kv_read) ? pair_pos[(int64_t)lane * nv + u] : -1;
    // the affine epilogue's scale / shift of this lane's first eight columns: the same for every row of the walk -- loaded once
    // (in the loop they were 4 KiB of L1 requests per output row beside the row's 11 KiB of partial rows)
    float sc0[8], sh0[8];
    {
        const int c = lane * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 a = (scale && c < cout) ? *reinterpret_cast<const float4 *>(scale + c + h * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
            const float4 b = (shift && c < cout) ? *reinterpret_cast<const float4 *>(shift + c + h * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            sc0[h * 4] = a.x; sc0[h * 4 + 1] = a.y; sc0[h * 4 + 2] = a.z; sc0[h * 4 + 3] = a.w;
            sh0[h * 4] = b.x; sh0[h * 4 + 1] = b.y; sh0[h * 4 + 2] = b.z; sh0[h * 4 + 3] = b.w;
        }
    }
    for (; u < row_end; u += n_waves) {
        const int mypos = mypos_next;
        if (u + n_waves < row_end) mypos_next = (lane < kv && lane != kv_read) ? pair_pos[(int64_t)lane * nv + u + n_waves] : -1;
        float av[WIDE ? 2 : 1][8];
        float amax = 0.f;
        // columns [0, 512) and [512, 1024) with static register sets (the row-scaled split needs the whole row before its first store);
        // wider rows (no row scale: checked on the host) continue in the loop below
        auto block = [&](int c0, float (&a)[8]) {
            const int c = c0 + lane * 8;
            const bool act = c < cout;
            conv_gather_sum_q24<WIDE ? 4 : GS_NL>(pb, e_off, mypos, kv, cout, c, act, pair_base, a);   // (the wide form: fewer loads in flight, no spills)
            if (!act) return;
            if (c0 == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = a[j] * sc0[j] + sh0[j];
            } else {
                float scv[8], shv[8];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float4 sc = scale ? *reinterpret_cast<const float4 *>(scale + c + h * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
                    const float4 sh = shift ? *reinterpret_cast<const float4 *>(shift + c + h * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                    scv[h * 4] = sc.x; scv[h * 4 + 1] = sc.y; scv[h * 4 + 2] = sc.z; scv[h * 4 + 3] = sc.w;
                    shv[h * 4] = sh.x; shv[h * 4 + 1] = sh.y; shv[h * 4 + 2] = sh.z; shv[h * 4 + 3] = sh.w;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = a[j] * scv[j] + shv[j];
            }
            if (residual) {
                const float4 r0 = *reinterpret_cast<const float4 *>(residual + u * ld_res + c);
                const float4 r1 = *reinterpret_cast<const float4 *>(residual + u * ld_res + c + 4);
                a[0] += r0.x; a[1] += r0.y; a[2] += r0.z; a[3] += r0.w; a[4] += r1.x; a[5] += r1.y; a[6] += r1.z; a[7] += r1.w;
            }
            if (res_hi) {
                const _Float16 *rph = r_il ? res_hi + u * ld_rh + ((c >> 5) << 6) + (c & 31) : res_hi + u * ld_rh + c;
                const _Float16 *rpl = r_il ? rph + 32 : res_lo + u * ld_rh + c;
                const f16x8 rh = *reinterpret_cast<const f16x8 *>(rph), rl = *reinterpret_cast<const f16x8 *>(rpl);
                const float ri = res_inv ? res_inv[u] : 1.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] += ((float)rh[j] + (float)rl[j]) * ri;
            }
            if (relu) {
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = fmaxf(a[j], 0.f);
            }
            if (y) {                                         // fp32 copy only where a later layer reads it (residual, linear)
                *reinterpret_cast<float4 *>(y + u * ld_y + c) = make_float4(a[0], a[1], a[2], a[3]);
                *reinterpret_cast<float4 *>(y + u * ld_y + c + 4) = make_float4(a[4], a[5], a[6], a[7]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(a[j]));
        };
        auto split_store = [&](int c0, const float (&a)[8], float s) {
            const int c = c0 + lane * 8;
            if (c >= cout) return;
            f16x8 h, l;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = a[j] * s;
                h[j] = (_Float16)v;
                l[j] = (_Float16)(v - (float)h[j]);
            }
            _Float16 *ph = y_il ? y_hi + u * ld_yh + ((c >> 5) << 6) + (c & 31) : y_hi + u * ld_yh + c;
            _Float16 *pl = y_il ? ph + 32 : y_lo + u * ld_yh + c;
            *reinterpret_cast<f16x8 *>(ph) = h;
            *reinterpret_cast<f16x8 *>(pl) = l;
        };
        block(0, av[0]);
        if (!rowscale && y_hi) split_store(0, av[0], 1.f);
        if constexpr (WIDE) {
            block(512, av[1]);
            if (!rowscale && y_hi) split_store(512, av[1], 1.f);
        }
        if (rowscale) {
            // pre-split output with a per-row power of two: hi + lo = y * 2^e, the row's largest magnitude in [2^13, 2^14) (see
            // conv_phase2_kernel); the next layer's phase 1 multiplies its partial rows by y_inv_scale[row] = 2^-e (exact)
            amax = gp_wave_max(amax);
            const float s = gp_pow2_for(amax);
            if (lane == 0) y_inv_scale[u] = 1.f / s;
            split_store(0, av[0], s);
            if constexpr (WIDE) split_store(512, av[1], s);
        } else if constexpr (WIDE) {
            for (int c0 = 1024; c0 < cout; c0 += 512) {
                block(c0, av[0]);
                if (y_hi) split_store(c0, av[0], 1.f);
            }
        }
    }
}

// fp32 rows -> hi/lo f16 rows with a power-of-two pre-scale: global (device scalar `scale`, from gp_pow2_scale) or per row
// (row_inv receives 2^-e(row)).  One wave per row in the per-row mode (d <= 1024... any d: two sweeps).
__global__ void split_rows_scaled_kernel(const float *__restrict__ x, int64_t ld_x, int d, int64_t n, _Float16 *__restrict__ hi,
                                         _Float16 *__restrict__ lo, int64_t ld_h, const float *__restrict__ scale,
                                         float *__restrict__ row_inv, const int32_t *__restrict__ dst_row) {
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    const int lane = gp_lane();
    for (int64_t r = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6; r < n; r += ((int64_t)gridDim.x * blockDim.x) >> 6) {
        float s = scale ? scale[0] : 1.f;
        const int64_t ro = dst_row ? (int64_t)dst_row[r] : r;            // (the row the halves are written to: gp_rcb_order's map)
        if (row_inv) {
            float amax = 0.f;
            for (int c = lane * 4; c < d; c += 256) {
                float4 a = *reinterpret_cast<const float4 *>(x + r * ld_x + c);
                amax = fmaxf(amax, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
            }
            s = gp_pow2_for(gp_wave_max(amax));
            if (lane == 0) row_inv[ro] = 1.f / s;
        }
        for (int c = lane * 4; c < d; c += 256) {
            float4 a = *reinterpret_cast<const float4 *>(x + r * ld_x + c);
            float v[4] = {a.x * s, a.y * s, a.z * s, a.w * s};
            f16x4 h, l;
#pragma unroll
            for (int k = 0; k < 4; ++k) { h[k] = (_Float16)v[k]; l[k] = (_Float16)(v[k] - (float)h[k]); }
            if (lo) {
                *reinterpret_cast<f16x4 *>(hi + ro * ld_h + c) = h;
                *reinterpret_cast<f16x4 *>(lo + ro * ld_h + c) = l;
            } else {                                                   // interleaved rows: [32-column step][hi 32 | lo 32] in ONE tensor
                _Float16 *ph = hi + ro * ld_h + ((c >> 5) << 6) + (c & 31);
                *reinterpret_cast<f16x4 *>(ph) = h;
                *reinterpret_cast<f16x4 *>(ph + 32) = l;
            }
        }
    }
}

// amax over an [n, d] block: non-negative floats order like their bit patterns -> atomicMax on the uint image
__global__ void amax_kernel(const float *__restrict__ x, int64_t ld_x, int d, int64_t n, unsigned *__restrict__ out) {
    float m = 0.f;
    const int64_t total = n * (d / 4);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i / (d / 4);
        int c = (int)(i - r * (d / 4)) * 4;
        float4 a = *reinterpret_cast<const float4 *>(x + r * ld_x + c);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
    }
    // one atomic per block (same-address atomics serialise at the memory side)
    __shared__ float s_m[4];
    m = gp_wave_max(m);
    if (gp_lane() == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, __float_as_uint(fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]))));
}
__global__ void pow2_scale_kernel(const unsigned *__restrict__ amax_bits, float *__restrict__ out2) {
    const float s = gp_pow2_for(__uint_as_float(amax_bits[0]));
    out2[0] = s;
    out2[1] = 1.f / s;
}

// fp32 rows -> hi/lo f16 rows (input of the first layer)
__global__ void split_rows_kernel(const float *__restrict__ x, int64_t ld_x, int d, int64_t n, _Float16 *__restrict__ hi,
                                  _Float16 *__restrict__ lo, int64_t ld_h) {
    int64_t total = n * (d / 4);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i / (d / 4);
        int c = (int)(i - r * (d / 4)) * 4;
        float4 a = *reinterpret_cast<const float4 *>(x + r * ld_x + c);
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        float v[4] = {a.x, a.y, a.z, a.w};
        f16x4 h, l;
#pragma unroll
        for (int k = 0; k < 4; ++k) { h[k] = (_Float16)v[k]; l[k] = (_Float16)(v[k] - (float)h[k]); }
        *reinterpret_cast<f16x4 *>(hi + r * ld_h + c) = h;
        *reinterpret_cast<f16x4 *>(lo + r * ld_h + c) = l;
    }
}

// weight prepare: w fp32 [kv][cin][cout] -> hi/lo f16 [kv][cout][cin], scaled by `s`
__global__ void weight_split_kernel(const float *__restrict__ w, int kv, int cin, int cout, float s,
                                    _Float16 *__restrict__ hi, _Float16 *__restrict__ lo) {
    int64_t total = (int64_t)kv * cin * cout;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t k = i / ((int64_t)cin * cout);
        int64_t rem = i - k * (int64_t)cin * cout;
        int n = (int)(rem / cin), c = (int)(rem % cin);               // output index order [k][n][c]
        float v = w[(k * cin + c) * cout + n] * s;
        _Float16 h = (_Float16)v;
        hi[i] = h;
        lo[i] = (_Float16)(v - (float)h);
    }
}

// the same split into the step-blocked layout of the two-phase kernels: [kv][cout / 256][cin / 32][256 rows][32 halfs], row rho of a
// column tile = its column (rho & 128) | (rho & 15) << 3 | (rho >> 4 & 7) (the LDS row the LDS-DMA kernel stages it in)
// transpose_flip: the operand of the DATA-GRADIENT convolution V[k] = W[kv - 1 - k]^T taken straight from w ([kv][cout][cin] of V's
// dimensions: the forward layer's [kv][its cin][its cout]) -- no flipped / transposed fp32 copy in between, and reads along w's rows.
// One workgroup per (offset, column tile, K step) = one 16-KiB block of each output; thread c owns column c of the tile: it reads the step's
// 32 values of its column (forward: one coalesced 1-KiB row of w per value across the workgroup; transposed: 128 contiguous bytes of its own
// row) and writes them as ONE 64-byte run per half into row rho(c) = (c & 128) | (c >> 3 & 15) | (c & 7) << 4 of the block.  (The element-
// per-thread form this replaces read w with a 2-KiB stride between neighbouring lanes: 34 us per 28-MB layer, 17 layers per training step.)
__global__ void __launch_bounds__(256)
weight_split_blocked_kernel(const float *__restrict__ w, int kv, int cin, int cout, float s,
                            _Float16 *__restrict__ hi, _Float16 *__restrict__ lo, int transpose_flip) {
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    const int steps = cin / TK, nt = cout / TN;
    const int st = blockIdx.x % steps, t = (blockIdx.x / steps) % nt, k = blockIdx.x / (steps * nt);
    const int c = threadIdx.x;
    const int rho = (c & 128) | ((c >> 3) & 15) | ((c & 7) << 4);
    float v[TK];
    if (transpose_flip) {
        const float *src = w + ((int64_t)(kv - 1 - k) * cout + t * TN + c) * cin + st * TK;
#pragma unroll
        for (int q = 0; q < TK / 4; ++q) *reinterpret_cast<float4 *>(v + 4 * q) = *reinterpret_cast<const float4 *>(src + 4 * q);
    } else {
        const float *src = w + ((int64_t)k * cin + st * TK) * cout + t * TN + c;
#pragma unroll
        for (int kk = 0; kk < TK; ++kk) v[kk] = src[(int64_t)kk * cout];
    }
    const int64_t o = ((((int64_t)k * nt + t) * steps + st) * TN + rho) * TK;
#pragma unroll
    for (int q = 0; q < TK / 8; ++q) {
        f16x8 h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float x = v[q * 8 + j] * s; h[j] = (_Float16)x; l[j] = (_Float16)(x - (float)h[j]); }
        *reinterpret_cast<f16x8 *>(hi + o + q * 8) = h;
        *reinterpret_cast<f16x8 *>(lo + o + q * 8) = l;
    }
}

size_t scan_tmp32(int64_t n) {
    size_t t = 0;
    (void)rocprim::exclusive_scan(nullptr, t, (int32_t *)nullptr, (int32_t *)nullptr, (int32_t)0, (size_t)n, rocprim::plus<int32_t>(), 0);
    return t;
}

}  // namespace

extern int g_gp_knobs[16];
extern void *g_gp_debug_ptr[4];
extern size_t g_gp_debug_bytes[4];
#define g_conv_ablate g_gp_knobs[3]


extern "C" size_t gp_conv_pairs_workspace_bytes(int64_t nv, int32_t kv) {
    if (nv <= 0 || kv <= 0) return 0;
    GpCarver cv(nullptr, 0);
    cv.take<int32_t>((size_t)kv * nv);
    cv.take<int32_t>((size_t)kv * nv);
    cv.take<char>(scan_tmp32((int64_t)kv * nv));
    return cv.off;
}

extern "C" int gp_conv_pairs_build(const int32_t *nbr_map, int64_t nv, int32_t kv, int32_t num_chunks, const int32_t *chunk_row_off,
                                   int32_t *pair_in, int32_t *pair_pos, int32_t *seg_off, int32_t *tile_start, int32_t *tile_desc,
                                   void *workspace, size_t workspace_bytes, void *stream_) {
    GP_CHECK_ARG(nbr_map && pair_in && pair_pos && seg_off && tile_start && tile_desc && workspace && nv > 0 && kv > 0, "gp_conv_pairs_build: null/empty argument");
    GP_CHECK_ARG((int64_t)kv * nv < (1ll << 31), "gp_conv_pairs_build: kernel map too large");
    GP_CHECK_ARG(num_chunks >= 1 && chunk_row_off, "gp_conv_pairs_build: num_chunks=%d needs the chunk row offsets (device, [num_chunks + 1], 0 .. nv ascending)", num_chunks);
    int64_t total = (int64_t)kv * nv;
    GpCarver cv(workspace, workspace_bytes);
    int32_t *f = cv.take<int32_t>(total), *sc = cv.take<int32_t>(total);
    size_t tb = scan_tmp32(total);
    char *tmp = cv.take<char>(tb);
    if (!cv.ok()) { gp_set_error("gp_conv_pairs_build: workspace too small (%zu < %zu)", workspace_bytes, cv.off); return GP_ENOMEM; }
    hipStream_t s = gp_stream(stream_);
    int blocks = (int)((total + 255) / 256);
    int nseg = num_chunks * kv;
    pair_flags_kernel<<<blocks, 256, 0, s>>>(nbr_map, nv, kv, chunk_row_off, num_chunks, f);
    GP_CHECK_HIP(rocprim::exclusive_scan(tmp, tb, f, sc, (int32_t)0, (size_t)total, rocprim::plus<int32_t>(), s));
    pair_emit_kernel<<<blocks, 256, 0, s>>>(nbr_map, sc, nv, kv, chunk_row_off, num_chunks, pair_in, pair_pos, seg_off);
    tile_start_kernel<<<1, 64, 0, s>>>(seg_off, nseg, tile_start);
    tile_desc_kernel<<<(nseg + 255) / 256, 256, 0, s>>>(seg_off, tile_start, nseg, kv, reinterpret_cast<int4 *>(tile_desc));
    GP_CHECK_LAUNCH();
    return GP_OK;
}

// Chunk heights for gp_conv_pairs_build chosen from the kernel map: chunks are closed at multiples of granule_rows so that a chunk's
// phase-1 launch -- sum over the offsets of ceil(pairs / 256) row tiles, times col_tiles column tiles -- stays within target_tiles
// (2 x the CU count: two rounds of one-tile workgroups; equal heights leave 4-8 % of the tile slots of their rounds empty).
// chunk_row_off i32 [max_chunks + 1] and n_chunks i32 [1] are device outputs (read back by the caller to size the pair arrays).
extern "C" size_t gp_conv_chunk_plan_workspace_bytes(int64_t nv, int32_t granule_rows) {
    if (nv <= 0 || granule_rows <= 0) return 0;
    return (size_t)((nv + granule_rows - 1) / granule_rows) * 32 * sizeof(int32_t);
}
extern "C" int gp_conv_chunk_plan(const int32_t *nbr_map, int64_t nv, int32_t kv, int32_t granule_rows, int32_t col_tiles,
                                  int32_t target_tiles, int32_t max_chunks, int32_t *chunk_row_off, int32_t *n_chunks, void *workspace,
                                  size_t workspace_bytes, void *stream_) {
    GP_CHECK_ARG(nbr_map && chunk_row_off && n_chunks && workspace && nv > 0 && kv > 0 && kv <= 32, "gp_conv_chunk_plan: null/empty argument");
    GP_CHECK_ARG(granule_rows >= 64 && granule_rows <= 65535 && col_tiles >= 1 && target_tiles >= col_tiles && max_chunks >= 1, "gp_conv_chunk_plan: bad sizes");
    const int64_t ngran = (nv + granule_rows - 1) / granule_rows;
    GP_CHECK_ARG(workspace_bytes >= gp_conv_chunk_plan_workspace_bytes(nv, granule_rows), "gp_conv_chunk_plan: workspace too small");
    hipStream_t s = gp_stream(stream_);
    int32_t *cnt = static_cast<int32_t *>(workspace);
    chunk_count_kernel<<<dim3((unsigned)ngran, (unsigned)kv), 256, 0, s>>>(nbr_map, nv, kv, granule_rows, cnt);
    chunk_plan_kernel<<<1, 256, 0, s>>>(cnt, (int)ngran, kv, granule_rows, nv, col_tiles, target_tiles, max_chunks, chunk_row_off, n_chunks);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" int gp_conv_weights_split(const float *w, int32_t kv, int32_t cin, int32_t cout, float scale_pow2,
                                     void *w_hi, void *w_lo, void *stream_) {
    GP_CHECK_ARG(w && w_hi && w_lo && kv > 0 && cin > 0 && cout > 0, "gp_conv_weights_split: null/empty argument");
    weight_split_kernel<<<2048, 256, 0, gp_stream(stream_)>>>(w, kv, cin, cout, scale_pow2, static_cast<_Float16 *>(w_hi),
                                                              static_cast<_Float16 *>(w_lo));
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" int gp_conv_weights_split_blocked(const float *w, int32_t kv, int32_t cin, int32_t cout, float scale_pow2, void *w_hi,
                                             void *w_lo, int32_t transpose_flip, void *stream_) {
    GP_CHECK_ARG(w && w_hi && w_lo && kv > 0 && cin > 0 && cout > 0, "gp_conv_weights_split_blocked: null/empty argument");
    GP_CHECK_ARG(cin % TK == 0 && cout % TN == 0, "gp_conv_weights_split_blocked: cin=%d must be a multiple of %d and cout=%d of %d", cin, TK, cout, TN);
    GP_CHECK_ARG(!transpose_flip || ((reinterpret_cast<uintptr_t>(w) & 15) == 0), "gp_conv_weights_split_blocked: w must be 16-byte aligned");
    weight_split_blocked_kernel<<<(unsigned)((int64_t)kv * (cout / TN) * (cin / TK)), 256, 0, gp_stream(stream_)>>>(
        w, kv, cin, cout, scale_pow2, static_cast<_Float16 *>(w_hi), static_cast<_Float16 *>(w_lo), transpose_flip);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" int gp_split_f16(const float *x, int64_t ld_x, int32_t d, int64_t n, void *hi, void *lo, int64_t ld_h,
                            void *stream_) {
    GP_CHECK_ARG(x && hi && lo && n > 0 && d > 0 && d % 4 == 0 && ld_x % 4 == 0 && ld_h % 4 == 0, "gp_split_f16: bad argument");
    split_rows_kernel<<<2048, 256, 0, gp_stream(stream_)>>>(x, ld_x, d, n, static_cast<_Float16 *>(hi), static_cast<_Float16 *>(lo), ld_h);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" int gp_pow2_scale(const float *x, int64_t ld_x, int32_t d, int64_t n, float *scale2, void *workspace,
                             size_t workspace_bytes, void *stream_) {
    GP_CHECK_ARG(x && scale2 && workspace && workspace_bytes >= 4 && n > 0 && d > 0 && d % 4 == 0 && ld_x % 4 == 0,
                 "gp_pow2_scale: bad argument");
    hipStream_t s = gp_stream(stream_);
    GP_CHECK_HIP(hipMemsetAsync(workspace, 0, 4, s));
    amax_kernel<<<1024, 256, 0, s>>>(x, ld_x, d, n, static_cast<unsigned *>(workspace));      // 256 threads: 4-wave block reduce
    pow2_scale_kernel<<<1, 1, 0, s>>>(static_cast<const unsigned *>(workspace), scale2);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" int gp_split_f16_scaled(const float *x, int64_t ld_x, int32_t d, int64_t n, void *hi, void *lo, int64_t ld_h,
                                   const float *scale, float *row_inv_scale, const int32_t *dst_row, void *stream_) {
    GP_CHECK_ARG(x && hi && n > 0 && d > 0 && d % 4 == 0 && ld_x % 4 == 0 && ld_h % 4 == 0, "gp_split_f16_scaled: bad argument");
    GP_CHECK_ARG(lo || (d % 32 == 0 && ld_h >= 2 * (int64_t)d), "gp_split_f16_scaled: lo = NULL asks for interleaved rows in hi: d %% 32 == 0 and ld_h >= 2 d");
    GP_CHECK_ARG(!(scale && row_inv_scale), "gp_split_f16_scaled: one global scale OR per-row scales");
    int64_t waves = n < 16384 ? n : 16384;
    split_rows_scaled_kernel<<<(unsigned)((waves * 64 + 255) / 256), 256, 0, gp_stream(stream_)>>>(
        x, ld_x, d, n, static_cast<_Float16 *>(hi), static_cast<_Float16 *>(lo), ld_h, scale, row_inv_scale, dst_row);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" int gp_sparse_conv_f16x3(const float *x, int64_t ld_x, const void *x_hi, const void *x_lo, int64_t ld_xh,
                                    const int32_t *pair_in, const int32_t *pair_pos,
                                    const int32_t *pair_off, const int32_t *tile_start, const int32_t *tile_desc,
                                    int32_t nseg, int64_t num_pairs, int64_t nv, int32_t kv,
                                    const void *w_hi, const void *w_lo, int32_t cin, int32_t cout, float *partial,
                                    const float *scale, const float *shift, const float *residual, int64_t ld_res,
                                    int32_t relu, float *y, int64_t ld_y, void *y_hi, void *y_lo, int64_t ld_yh,
                                    int32_t num_chunks, const int32_t *chunk_row_off_host, const int32_t *chunk_tile_off_host,
                                    const int32_t *chunk_pair_off_host, const float *x_row_inv_scale, float *y_row_inv_scale,
                                    const void *res_hi, const void *res_lo, int64_t ld_rh, const float *res_row_inv_scale, int32_t w_blocked,
                                    int32_t plane_flags, void *stream_) {
    GP_CHECK_ARG(plane_flags >= 0 && plane_flags < 32, "gp_sparse_conv_f16x3: plane_flags is a mask of 1 (x interleaved), 2 (y interleaved), 4 (residual interleaved), 8 (fp32 partial rows), 16 (one dense offset: phase 1 writes y)");
    // bit 4: ONE offset whose map holds every output row (a gather-GEMM: the training sampler's anchors x points similarity): pair p IS
    // output row p, nothing is summed, so phase 1 stores its fp32 rows straight into y and phase 2 (a 2 x 2.4 GB round trip there) is not run
    const bool direct = (plane_flags & 16) != 0;
    GP_CHECK_ARG(!direct || (kv == 1 && num_pairs == nv && x_hi && y && ld_y == cout && !scale && !shift && !residual && !res_hi && !relu && !y_hi &&
                             !y_row_inv_scale && !(g_gp_knobs[3] & 16)),
                 "gp_sparse_conv_f16x3: plane_flags bit 4 is for kv = 1 with a pair for every row, pre-split x, contiguous fp32 y and no epilogue");
    GP_CHECK_ARG(!(plane_flags & 1) || (x_hi && ld_xh % 64 == 0 && cin % 32 == 0), "gp_sparse_conv_f16x3: interleaved x rows come as ONE tensor (x_hi) of 2 x cin halfs per row");
    GP_CHECK_ARG(!(plane_flags & 2) || (y_hi && ld_yh % 64 == 0 && (uintptr_t)y_hi % 16 == 0), "gp_sparse_conv_f16x3: interleaved y rows go to ONE tensor (y_hi) of 2 x cout halfs per row");
    GP_CHECK_ARG(!(plane_flags & 4) || (res_hi && ld_rh % 64 == 0), "gp_sparse_conv_f16x3: interleaved residual rows come as ONE tensor (res_hi)");
    GP_CHECK_ARG(w_blocked == 0 || w_blocked == 1, "gp_sparse_conv_f16x3: w_blocked is 0 (row-major weights) or 1 (gp_conv_weights_split_blocked)");
    GP_CHECK_ARG(!res_hi || ((res_lo || (plane_flags & 4)) && !residual && ld_rh % 8 == 0 && (uintptr_t)res_hi % 16 == 0 && (uintptr_t)res_lo % 16 == 0),
                 "gp_sparse_conv_f16x3: the residual comes as fp32 rows OR as 16-byte aligned split planes (res_hi + res_lo), not both");
    GP_CHECK_ARG(res_hi || (!res_lo && !res_row_inv_scale), "gp_sparse_conv_f16x3: res_lo / res_row_inv_scale belong to res_hi");
    GP_CHECK_ARG((x || (x_hi && (x_lo || (plane_flags & 1)))) && pair_in && pair_pos && pair_off && tile_start && tile_desc && nseg > 0 && w_hi && w_lo && partial && (y || y_hi), "gp_sparse_conv_f16x3: null argument");
    GP_CHECK_ARG(!x_hi || (ld_xh % 8 == 0 && (uintptr_t)x_hi % 16 == 0 && (uintptr_t)x_lo % 16 == 0), "gp_sparse_conv_f16x3: pre-split rows must be 16-byte aligned");
    GP_CHECK_ARG(!(plane_flags & 1) || !(g_gp_knobs[3] & 16), "gp_sparse_conv_f16x3: interleaved x rows need the LDS-DMA path");
    GP_CHECK_ARG(!y_hi || ((y_lo || (plane_flags & 2)) && ld_yh % 4 == 0), "gp_sparse_conv_f16x3: y_hi/y_lo come as a pair");
    GP_CHECK_ARG(!x_row_inv_scale || x_hi, "gp_sparse_conv_f16x3: x_row_inv_scale belongs to pre-split operands (x_hi/x_lo)");
    GP_CHECK_ARG(!y_row_inv_scale || (y_hi && cout <= 1024), "gp_sparse_conv_f16x3: y_row_inv_scale needs y_hi/y_lo and cout <= 1024");
    GP_CHECK_ARG(nv > 0 && num_pairs > 0 && (kv == 27 || kv == 1), "gp_sparse_conv_f16x3: bad sizes");
    GP_CHECK_ARG(cin % TK == 0, "gp_sparse_conv_f16x3: cin=%d must be a multiple of %d", cin, TK);
    GP_CHECK_ARG(x_hi || (ld_x % 4 == 0 && (uintptr_t)x % 16 == 0), "gp_sparse_conv_f16x3: x rows must be 16-byte aligned");
    GP_SMEM_ATTR(conv_phase1_kernel<false>, sizeof(V2Smem));
    GP_SMEM_ATTR(conv_phase1_kernel<true>, sizeof(V2Smem));
    constexpr size_t P1_DMA_SMEM = sizeof(V2Smem) + TM * sizeof(float);     // the ring + the tile's 256 row scales
    GP_SMEM_ATTR(conv_phase1_dma_kernel, P1_DMA_SMEM);
    GP_SMEM_ATTR(conv_phase1_tuning_kernel, P1_DMA_SMEM);
    GP_SMEM_ATTR(conv_phase1_stamp_kernel, P1_DMA_SMEM);

    // tuning aid: gp_debug_ptr(1, buf, bytes) selects the stamped twin; every chunk launch writes its workgroups' stamps at
    // blockIdx * 16 uint64 (a chunk overwrites the previous one's: the last chunk of the last call stays)
    uint64_t *stamp = static_cast<uint64_t *>(g_gp_debug_ptr[1]);
    GP_CHECK_ARG(cout % TN == 0, "gp_sparse_conv_f16x3: cout=%d must be a multiple of %d on this path", cout, TN);
    hipStream_t s = gp_stream(stream_);
    int n_tiles = cout / TN;
    // Chunked execution: phase 1 and phase 2 alternate over chunks of output rows so that the partial
    // rows of a chunk (written by phase 1, read once by phase 2) stay in the 256 MiB Infinity Cache instead
    // of making an HBM round trip; `partial` then only needs room for the largest chunk.
    // Measured and left out: pipelining the chunks over helper streams (phase 2 of chunk c beside phase 1 of chunk c+1, two
    // partial slots) -- 2.13 vs 2.07 ms per 512->512 layer: phase 1 is co-limited by its own L2 traffic (A gather + weight
    // tiles + partial stores), so a memory-bound neighbour only takes bandwidth from it.  Round 4, again with both slots inside the
    // Infinity Cache (scripts/conv_pipeline_probe.py, profiles/r04_conv_pipelined_chunks.log): 8192-row chunks 2.08 vs 1.89 ms,
    // 6144 rows 2.26 vs 2.23, 4096 rows 2.82 vs 2.72 -- slower at every height, same bits; the 2 x 17 cross-stream event waits cost
    // more than phase 2's 25 us launches hide.
    // chunk tables (host copies of the per-chunk tile / pair offsets) give EXACT tile counts; without them tile_count is an
    // upper bound and only the one-tile-per-workgroup kernels (which test tile_start[nseg] on the device) may run
    // (The round-2 fault -- a memory access fault in scripts/bench_conv.py and an abort in the unchunked case of
    // test_f16x3_student_chain_vs_fp64_oracle -- was an experimental persistent phase 1 that walked tile_count tiles without
    // that device-side test; it was removed in 27fdcb6.  Every kernel launched here tests `mt >= tile_start[nseg]`.)
    // Half-specified chunking is an error, not a silent fall-back to the upper bound: the caller sized `partial` for chunks.
    GP_CHECK_ARG((num_chunks >= 1) == (chunk_tile_off_host != nullptr) && (num_chunks >= 1) == (chunk_pair_off_host != nullptr) &&
                     (num_chunks >= 1) == (chunk_row_off_host != nullptr),
                 "gp_sparse_conv_f16x3: num_chunks=%d needs ALL THREE host chunk tables (row, tile and pair offsets), num_chunks=0 needs none",
                 num_chunks);
    if (num_chunks >= 1) {
        bool rows_ok = chunk_row_off_host[0] == 0 && chunk_row_off_host[num_chunks] == nv;
        for (int c = 0; c < num_chunks && rows_ok; ++c) rows_ok = chunk_row_off_host[c + 1] > chunk_row_off_host[c];
        GP_CHECK_ARG(rows_ok, "gp_sparse_conv_f16x3: the %d chunk row offsets do not cover rows 0 .. %lld in ascending order", num_chunks, (long long)nv);
    }
    const bool chunked = num_chunks >= 1;
    const int nchunk = chunked ? num_chunks : 1;
    constexpr int p2_wg_per_cu = 6;                       // (rounds 4-5: GP_CONV_P2_WG_PER_CU for the sweep -- 4 / 5 / 6 / 8: 1.954 / 1.935 / 1.946 / 1.967 ms per layer)
    // Measured and left out (round 2): a persistent phase 1 (one workgroup per CU, 3-deep ring for the gathered rows issued two
    // steps ahead, weight tiles one step ahead, split staging roles, rings and epilogue stores running through tile boundaries,
    // swapped MFMA operands for 16-byte partial stores): bit-identical results, 1.99 vs 1.96 ms per 512->512 layer.  Its
    // ablations say why: matrix work alone 0.86 ms, loads alone 0.50 ms, loads + partial stores 0.97 ms, all three 1.45 ms,
    // loop skeleton 0.08 ms -- phase 1 is co-limited by its 9.7 GB of L2 / Infinity-Cache traffic per layer (3.9 GB gathered
    // rows + 3.8 GB weight tiles + 2 GB partial rows), which scheduling does not change.
    // The 24-bit partial rows are addressed with 32-bit byte offsets: every chunk is checked BEFORE the first launch, and a call with an
    // oversized chunk (an unchunked call with > ~2.7 M pairs at 512 columns) runs on the fp32 partial rows of rounds 1-4 instead -- 64-bit
    // offsets, the tuning twin's stores + conv_phase2_kernel -- rather than failing half way with y partly written.
    bool q24_fits = !(plane_flags & 8);                   // (bit 3 of plane_flags ASKS for the fp32 rows: the reference form of the 24-bit format)
    for (int c = 0; c < nchunk; ++c) {
        const int64_t cp = chunked ? (int64_t)chunk_pair_off_host[c + 1] - chunk_pair_off_host[c] : num_pairs;
        if (cp * cout * 3 + 32 >= ((int64_t)1 << 32)) q24_fits = false;
    }
    for (int c = 0; c < nchunk; ++c) {
        int tile_begin = chunked ? chunk_tile_off_host[c] : 0;
        int tile_count = chunked ? chunk_tile_off_host[c + 1] - tile_begin : (int)(num_pairs / TM + nseg);
        int pair_base = chunked ? chunk_pair_off_host[c] : 0;
        int64_t row_begin = chunked ? chunk_row_off_host[c] : 0;
        int64_t row_count = chunked ? chunk_row_off_host[c + 1] - row_begin : nv;
        // the LDS-DMA path keeps the chunk's partial rows as 24-bit block floating point inside `partial` (sized for fp32 rows by
        // the caller): three bytes per element, then one exponent byte per (row, 128 columns)
        const int64_t chunk_pairs = chunked ? (int64_t)chunk_pair_off_host[c + 1] - pair_base : num_pairs;
        const int64_t q_e_off = (chunk_pairs * cout * 3 + 15) & ~(int64_t)15;
        const bool dma_path = x_hi && !(g_conv_ablate & 16);
        const bool q24 = dma_path && !(g_conv_ablate & 32) && q24_fits;   // tuning bit 5: the fp32 partial rows of rounds 1-4, same kernels otherwise
        if (tile_count > 0) {
            int64_t nblocks = (((int64_t)tile_count * n_tiles + 7) / 8) * 8;
            const int tune = (g_conv_ablate & ~16) | ((dma_path && (!q24_fits || direct)) ? 32 : 0);   // bit 4 picks the register-staged path on the host; the rest are kernel tuning bits
#define P1_ARGS static_cast<const _Float16 *>(x_hi), static_cast<const _Float16 *>(x_lo), ld_xh, pair_in, pair_off, tile_start,              \
                reinterpret_cast<const int4 *>(tile_desc), nseg, kv, static_cast<const _Float16 *>(w_hi), static_cast<const _Float16 *>(w_lo), \
                cin, cout, direct ? y + (int64_t)pair_base * cout : partial, n_tiles, tune, tile_begin, tile_count, pair_base, x_row_inv_scale, stamp, \
                q_e_off, w_blocked, plane_flags & 1
            if (dma_path) {
                GP_CHECK_ARG(!stamp || g_gp_debug_bytes[1] >= (size_t)nblocks * 16 * sizeof(uint64_t),
                             "gp_sparse_conv_f16x3: the stamp buffer of gp_debug_ptr(1) holds %zu bytes, this launch writes %zu",
                             g_gp_debug_bytes[1], (size_t)nblocks * 16 * sizeof(uint64_t));
                if (stamp) conv_phase1_stamp_kernel<<<(unsigned)nblocks, NT2, P1_DMA_SMEM, s>>>(P1_ARGS);
                else if (tune) conv_phase1_tuning_kernel<<<(unsigned)nblocks, NT2, P1_DMA_SMEM, s>>>(P1_ARGS);
                else conv_phase1_dma_kernel<<<(unsigned)nblocks, NT2, P1_DMA_SMEM, s>>>(P1_ARGS);
            } else {
                GP_CHECK_ARG(x, "gp_sparse_conv_f16x3: fp32 x required for the register-staged path");
                GP_CHECK_ARG(!x_row_inv_scale, "gp_sparse_conv_f16x3: the register-staged path splits unscaled fp32 rows");
#define P1R_ARGS x, ld_x, pair_in, pair_off, tile_start, reinterpret_cast<const int4 *>(tile_desc), nseg, kv, static_cast<const _Float16 *>(w_hi), \
                 static_cast<const _Float16 *>(w_lo), cin, cout, partial, n_tiles, tune, tile_begin, tile_count, pair_base, w_blocked
                if (tune) conv_phase1_kernel<true><<<(unsigned)nblocks, NT2, sizeof(V2Smem), s>>>(P1R_ARGS);
                else conv_phase1_kernel<false><<<(unsigned)nblocks, NT2, sizeof(V2Smem), s>>>(P1R_ARGS);
#undef P1R_ARGS
            }
#undef P1_ARGS
        }
        if (direct) continue;
        // one resident round: 6 workgroups of 4 waves per CU (80 registers per lane)
        const int64_t p2_full = (row_count * 64 + 255) / 256;
        int64_t p2_res = (int64_t)gp_cu_count() * p2_wg_per_cu;
        // (the 24-bit kernel: 4 waves per SIMD at 128 registers -- 4 workgroups per CU are one resident round)
        if (q24 && p2_wg_per_cu == 6) p2_res = (int64_t)gp_cu_count() * 4;
        const unsigned p2_grid = (unsigned)((p2_res > 0 && p2_res < p2_full) ? p2_res : p2_full);
#define P2Q_ARGS reinterpret_cast<const unsigned char *>(partial), q_e_off, pair_pos, nv, kv, cout, scale, shift, residual, ld_res, \
                 relu, y, ld_y, static_cast<_Float16 *>(y_hi), static_cast<_Float16 *>(y_lo), ld_yh, row_begin, row_count, pair_base, y_row_inv_scale, \
                 static_cast<const _Float16 *>(res_hi), static_cast<const _Float16 *>(res_lo), ld_rh, res_row_inv_scale, plane_flags
        if (q24 && cout <= 512 && (g_conv_ablate & 256)) conv_phase2_q24_kernel<false, true><<<p2_grid, 256, 0, s>>>(P2Q_ARGS);
        else if (q24 && cout > 512) conv_phase2_q24_kernel<true><<<p2_grid, 256, 0, s>>>(P2Q_ARGS);
        else if (q24) conv_phase2_q24_kernel<false><<<p2_grid, 256, 0, s>>>(P2Q_ARGS);
#undef P2Q_ARGS
        else
            conv_phase2_kernel<<<p2_grid, 256, 0, s>>>(
                partial, pair_pos, nv, kv, cout, scale, shift, residual, ld_res, relu, y, ld_y, static_cast<_Float16 *>(y_hi),
                static_cast<_Float16 *>(y_lo), ld_yh, row_begin, row_count, pair_base, y_row_inv_scale,
                static_cast<const _Float16 *>(res_hi), static_cast<const _Float16 *>(res_lo), ld_rh, res_row_inv_scale, plane_flags);
    }
    GP_CHECK_LAUNCH();
    return GP_OK;
}
