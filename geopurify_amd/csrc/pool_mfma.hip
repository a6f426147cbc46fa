// Row 12, matrix-core variant of the affinity pooling (the metric's roofline stage).
//
// The tiled VALU kernels (pool_tiles.hip) are co-limited by L2->CU gather bandwidth and by the fp32 FMA
// rate (zero-padded tiles cost 1.8x the useful FMAs).  Here a workgroup owns a block of 16*NW
// Morton-adjacent rows (NW waves x 16 rows) and NC columns, and sweeps the block's neighbour union (a few
// union rows per output row instead of 96 neighbour rows) in steps of 32 union rows:
//   * a step's 32 rows x NC columns are staged into LDS by global_load_lds from PRE-SPLIT operands
//     (x = hi + lo, two f16 planes written by the previous application's epilogue);
//   * each wave multiplies its dense 16 x 32 weight block (pre-split f16, stored in MFMA A-fragment
//     order) with the staged rows on v_mfma_f32_16x16x32_f16, the B fragments read column-major from the
//     row-major image by ds_read_b64_tr_b16 (hardware transpose; the image is XOR-swizzled through the
//     DMA source addresses so that the reads are conflict-free: SQ_LDS_BANK_CONFLICT = 0);
//   * hi*hi + hi*lo + lo*hi with fp32 accumulation = fp32-class accuracy (same scheme as the sparse
//     convolution; the dropped lo*lo term is 2^-22 relative).
// The matrix cores make the zero padding free; what remains is the block-union traffic through L2.
#include <cstring>
#include <rocprim/device/device_scan.hpp>

#include "gp_common.h"

extern int g_gp_knobs[16];
extern void *g_gp_debug_ptr[4];
extern size_t g_gp_debug_bytes[4];

namespace {

// in-kernel time stamps (tuning aid, STAMP instantiations only: gp_debug_ptr(0, buffer) selects them)
__device__ __forceinline__ uint64_t pq_now() {
    uint64_t t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ uint64_t pq_real() {
    uint64_t t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((vector_size(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int PM_KS = 32;             // union rows per step (MFMA K)
constexpr int PM_MIN_STEPS = 9;       // every row block is padded to at least this many steps
constexpr int PM_D = 512;             // columns
constexpr int PM_MAXID = 16384;       // ids sorted per block in the builder (block_rows x K <= 16384)
constexpr float PM_WSCALE = 1024.f;   // weights (<= 1) are stored x 2^10 so that their f16 lo parts stay normal

__device__ __forceinline__ void glds16(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

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

// sorted unique union of the neighbour ids of rows [b*br, b*br+br): de-duplicated through an LDS hash table
// (the union is ~7 % of the br*k ids), then a bitonic sort of the unique ids only.
// count pass: bu_n / padded count; fill pass: bu_row (padding repeats the first id; its weights stay zero).
constexpr int PM_HS = 16384;          // hash slots (>= the largest possible union, so insertion always ends)
__global__ void __launch_bounds__(1024)
pm_union_kernel(const int32_t *__restrict__ nbr, int64_t nv, int k, int br, int min_steps, int64_t *__restrict__ padded_cnt,
                int32_t *__restrict__ bu_n, const int64_t *__restrict__ bu_off, int32_t *__restrict__ bu_row) {
    extern __shared__ int s_mem[];                       // keys[PM_HS] | dense[np2(br*k)]
    int *keys = s_mem, *dense = s_mem + PM_HS;
    __shared__ int s_wcnt[16];
    __shared__ int s_base;
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t r0 = b * br;
    const int rows = (int)((nv - r0) < br ? (nv - r0) : br);
    const int n = rows * k;
    // a small table first (unions of lattice neighbourhoods are a few hundred ids: 2048 slots at < 25 % load; clearing and
    // compacting 16384 slots was most of this kernel's time), the full-size one if more than 1024 distinct ids show up or a
    // probe sequence gets long
    __shared__ int s_new, s_over;
    int hs = 2048, shift = 21;
    for (;;) {
        for (int i = tid; i < hs; i += 1024) keys[i] = -1;
        if (tid == 0) { s_base = 0; s_new = 0; s_over = 0; }
        __syncthreads();
        for (int i = tid; i < n; i += 1024) {
            const int id = nbr[r0 * k + i];
            unsigned h = ((unsigned)id * 2654435761u) >> shift;
            int probes = 0;
            while (true) {
                const int old = atomicCAS(&keys[h], -1, id);
                if (old == -1) { atomicAdd(&s_new, 1); break; }
                if (old == id) break;
                h = (h + 1) & (hs - 1);
                if (++probes > 256 && hs < PM_HS) { s_over = 1; break; }   // (the full table always has a free slot)
            }
        }
        __syncthreads();
        const bool redo = hs < PM_HS && (s_over || s_new > 1024);          // block-uniform
        __syncthreads();
        if (!redo) break;
        hs = PM_HS;
        shift = 18;
    }
    for (int i0 = 0; i0 < hs; i0 += 1024) {                             // compact the occupied slots
        const int key = keys[i0 + tid];
        const unsigned long long m = __ballot(key >= 0);
        if (lane == 0) s_wcnt[wv] = __popcll(m);
        __syncthreads();
        int before = s_base;
        for (int w = 0; w < wv; ++w) before += s_wcnt[w];
        if (key >= 0) dense[before + __popcll(m & ((1ull << lane) - 1ull))] = key;
        __syncthreads();
        if (tid == 0) { int tot = 0; for (int w = 0; w < 16; ++w) tot += s_wcnt[w]; s_base += tot; }
        __syncthreads();
    }
    // padded to whole steps and to at least min_steps steps (pad rows repeat the first id with zero weights; the persistent
    // kernel learns its next tile at step 5 and needs it from step n - 4 on: PM_MIN_STEPS); pass 2 takes the padded size of pass 1
    const int U = s_base;
    if (!bu_row) {
        if (tid == 0) { padded_cnt[b] = max((U + PM_KS - 1) / PM_KS, min_steps) * PM_KS; bu_n[b] = U; }
        return;
    }
    const int Up = (int)(bu_off[b + 1] - bu_off[b]);
    int np2 = 1;
    while (np2 < U) np2 <<= 1;
    for (int i = U + tid; i < np2; i += 1024) dense[i] = INT32_MAX;
    __syncthreads();
    bitonic_sort_lds(dense, np2, tid, 1024);
    const int64_t o = bu_off[b];
    for (int i = tid; i < Up; i += 1024) bu_row[o + i] = i < U ? dense[i] : dense[0];
}

// scatter the ELL weights into MFMA A-fragment order: wa[(kstep*nw + wave)*64 + lane][8], lane = (k>>3)*16 + m.
// One workgroup per row block: the block's sorted union (typically ~430 ids) is staged in LDS and every (row, neighbour)
// element finds its column there by binary search (in global memory the nine dependent loads per element were the cost:
// 0.17 ms per scene); unions beyond the LDS buffer fall back to the search in global memory.
constexpr int PW_LDS = 2048;
__global__ void __launch_bounds__(256)
pm_weights_kernel(const int32_t *__restrict__ nbr, const float *__restrict__ w, int64_t nv, int k, int br,
                  const int64_t *__restrict__ bu_off, const int32_t *__restrict__ bu_n,
                  const int32_t *__restrict__ bu_row, _Float16 *__restrict__ wa_hi, _Float16 *__restrict__ wa_lo) {
    __shared__ int s_u[PW_LDS];
    const int64_t b = blockIdx.x;
    const int nw = br / 16;
    const int64_t off = bu_off[b];
    const int un = bu_n[b];
    const int32_t *ug = bu_row + off;
    const bool in_lds = un <= PW_LDS;
    if (in_lds)
        for (int i = threadIdx.x; i < un; i += 256) s_u[i] = ug[i];
    __syncthreads();
    const int64_t r0 = b * br;
    const int rows = (int)((nv - r0) < br ? (nv - r0) : br);
    const int64_t ks0 = off / PM_KS;
    for (int t = threadIdx.x; t < rows * k; t += 256) {
        const int rl = t / k;
        const int64_t e = (r0 + rl) * k + (t - rl * k);
        const int id = nbr[e];
        int lo = 0, hi = un - 1;
        if (in_lds) {
            while (lo < hi) { int mid = (lo + hi) >> 1; if (s_u[mid] < id) lo = mid + 1; else hi = mid; }
        } else {
            while (lo < hi) { int mid = (lo + hi) >> 1; if (ug[mid] < id) lo = mid + 1; else hi = mid; }
        }
        const int wv = rl / 16, m = rl % 16;
        const int64_t ks = ks0 + lo / PM_KS;
        const int kk = lo % PM_KS;
        const int64_t idx = ((ks * nw + wv) * 64 + (kk >> 3) * 16 + m) * 8 + (kk & 7);
        const float v = w[e] * PM_WSCALE;
        const _Float16 h = (_Float16)v;
        wa_hi[idx] = h;
        wa_lo[idx] = (_Float16)(v - (float)h);
    }
}

// ------------------------------------------------------------------------------------------------ apply
// One workgroup = 16*NW rows x NC columns (grid = row blocks x 512/NC column parts, the parts of a row block
// adjacent on one XCD).  A 3-deep ring of stages is filled by LDS-DMA two steps ahead (~100 KiB in flight per
// CU: below that the gather is latency-bound).  A stage carries everything step k needs: 32 union rows x NC
// columns x {hi, lo}, the waves' weight fragments, and the row ids of stage k+2 (read back when that stage
// is issued), so the stage hand-over is the only synchronisation in the loop.
//
// Synchronisation is hand-counted: the LDS reads are inline asm with explicit lgkmcnt waits (a
// compiler-visible LDS read would wait for EVERY outstanding LDS-DMA, since the compiler cannot see that
// they target other ring slots, and serialise the ring), and the hand-over is `s_waitcnt vmcnt(N);
// s_barrier` with N = the DMA instructions issued for the later stage (loads return in order).
//
// Staging geometry (RB = 2*NC bytes per staged row and plane; one DMA instruction = 1 KiB = RPI rows):
//   wave wv stages union rows RPW*wv .. RPW*wv+RPW-1 (RPW = 32/NW) with two instructions per plane;
//   instruction i (0,1), lane part u = lane / (64/RPI) -> row RPW*wv + 2u + i, stored in LDS row slot
//   RPW*wv + RPI*i + u; lane chunk c = lane % (RB/16) lands in physical 16-byte chunk c and fetches logical
//   chunk c ^ 2t(row), t(row) = (row & 3) | ((row >> 3) & 1) << 2.
// Transposed reads: 16-lane group g owns k rows 8g..8g+7; lane 4q+p supplies row 8g+q (second read: row
// 8g+q+4), logical columns 4p..4p+3 of a 16-column block; with the swizzle the 32 lanes of a half-wave hit
// 32 distinct 8-byte slots of the 256-byte bank window.
template <int NW, int NC, int MT, int CGN>
struct PqGeo {
    // NW waves = RGN row groups x CGN column groups; a wave owns MT*16 rows x WC columns of the NC-column part
    static constexpr int RGN = NW / CGN;
    static constexpr int BR = 16 * MT * RGN;             // rows per workgroup
    static constexpr int WC = NC / CGN;                  // columns per wave
    static constexpr int RB = NC * 2;                    // bytes per staged row (one plane)
    static constexpr int RPW = PM_KS / NW;               // rows staged per wave
    static constexpr int RPI = 1024 / RB;                // rows per DMA instruction
    static constexpr int PLANE = PM_KS * RB;
    static constexpr int OFF_W = 2 * PLANE;              // weights: hi NW x 1 KiB, lo NW x 1 KiB (one per 16-row group)
    static constexpr int OFF_ID = OFF_W + NW * 2048;     // row ids: NW x 256 B (64 lanes x 4 B per DMA, RPW ids used)
    static constexpr int STAGE = OFF_ID + NW * 256;
    static constexpr int NST = 3;
    static constexpr int DMA_PER_STAGE = 7;              // per wave: 4 x rows, 2 x weights, 1 x ids
    static constexpr int NCB = WC / 16;                  // 16-column blocks per wave
    static constexpr int EP = WC + 4;                    // epilogue staging pitch (floats)
    static constexpr int ROWB = (RPW == 8 ? 2 : 4) * RB; // LDS distance row 8g+q -> row 8g+q+4
    static constexpr size_t SMEM = (size_t)NST * STAGE;
    static_assert(RPW / RPI == 2, "two row instructions per plane and wave");
    static_assert((size_t)NW * MT * 16 * EP * sizeof(float) <= SMEM, "epilogue staging must fit in the ring");
    static_assert(OFF_W + 2 * NW * 1024 < 65536, "intra-stage LDS offsets are 16-bit immediates");
};

template <int OFF>
__device__ __forceinline__ void pq_tr(s16x4 &d, uint32_t addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void pq_rd128(f16x8 &d, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void pq_rd64(int2 &d, uint32_t addr) {
    asm volatile("ds_read_b64 %0, %1" : "=v"(d) : "v"(addr));
}
template <int N>
__device__ __forceinline__ void pq_wait_lgkm(s16x4 (&f)[2][2][2]) {
    asm volatile("s_waitcnt lgkmcnt(%[n])"
                 : "+v"(f[0][0][0]), "+v"(f[0][0][1]), "+v"(f[0][1][0]), "+v"(f[0][1][1]), "+v"(f[1][0][0]), "+v"(f[1][0][1]),
                   "+v"(f[1][1][0]), "+v"(f[1][1][1])
                 : [n] "n"(N));
}
template <int N>
__device__ __forceinline__ void pq_wait_lgkm3(int2 &id, f16x8 (&a)[1], f16x8 (&b)[1]) {
    asm volatile("s_waitcnt lgkmcnt(%[n])" : "+v"(id), "+v"(a[0]), "+v"(b[0]) : [n] "n"(N));
}
template <int N>
__device__ __forceinline__ void pq_wait_lgkm3(int2 &id, f16x8 (&a)[2], f16x8 (&b)[2]) {
    asm volatile("s_waitcnt lgkmcnt(%[n])" : "+v"(id), "+v"(a[0]), "+v"(b[0]), "+v"(a[1]), "+v"(b[1]) : [n] "n"(N));
}
template <int N>
__device__ __forceinline__ void pq_wait_lgkm3(int2 &id, f16x8 (&a)[4], f16x8 (&b)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%[n])"
                 : "+v"(id), "+v"(a[0]), "+v"(b[0]), "+v"(a[1]), "+v"(b[1]), "+v"(a[2]), "+v"(b[2]), "+v"(a[3]), "+v"(b[3])
                 : [n] "n"(N));
}
template <int N>
__device__ __forceinline__ void pq_handover() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}
__device__ __forceinline__ f16x8 pq_cat(s16x4 a, s16x4 b) {
    typedef short s16x8 __attribute__((vector_size(16)));
    s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(f16x8, v);
}
// transposed reads of 2 column blocks (CB0, CB0+1) from the stage based at the address registers:
// f[u][plane][half]   (lgkmcnt is a 4-bit counter: groups of 8 reads, at most 11 LDS operations outstanding)
template <typename G, int CB0>
__device__ __forceinline__ void pq_read_group(s16x4 (&f)[2][2][2], const uint32_t (&addr)[8]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        constexpr int HI = ((CB0 + 0) >> 3) * 256;       // CB0 even: both blocks share the 128-column half
        pq_tr<HI>(f[u][0][0], addr[(CB0 + u) & 7]);
        pq_tr<HI + G::ROWB>(f[u][0][1], addr[(CB0 + u) & 7]);
        pq_tr<HI + G::PLANE>(f[u][1][0], addr[(CB0 + u) & 7]);
        pq_tr<HI + G::PLANE + G::ROWB>(f[u][1][1], addr[(CB0 + u) & 7]);
    }
}
template <typename G, int MT>
__device__ __forceinline__ void pq_mma_group(f32x4 *acc, const s16x4 (&f)[2][2][2], const f16x8 (&ah)[MT], const f16x8 (&al)[MT]) {
    f16x8 bh[2], bl[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) { bh[u] = pq_cat(f[u][0][0], f[u][0][1]); bl[u] = pq_cat(f[u][1][0], f[u][1][1]); }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int u = 0; u < 2; ++u)
            acc[mt * G::NCB + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bh[u], acc[mt * G::NCB + u], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int u = 0; u < 2; ++u)
            acc[mt * G::NCB + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bl[u], acc[mt * G::NCB + u], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int u = 0; u < 2; ++u)
            acc[mt * G::NCB + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt], bh[u], acc[mt * G::NCB + u], 0, 0, 0);
}
// all column blocks of one step, reads of group i+1 in flight under the MFMAs of group i
template <typename G, int MT, int CB>
__device__ __forceinline__ void pq_sweep(f32x4 *acc, s16x4 (&fa)[2][2][2], s16x4 (&fb)[2][2][2], const uint32_t (&addr)[8],
                                         const f16x8 (&ah)[MT], const f16x8 (&al)[MT]) {
    // entry: group CB is in flight in fa
    if constexpr (CB + 2 < G::NCB) {
        pq_read_group<G, CB + 2>(fb, addr);
        pq_wait_lgkm<8>(fa);
        pq_mma_group<G, MT>(acc + CB, fa, ah, al);
        pq_sweep<G, MT, CB + 2>(acc, fb, fa, addr, ah, al);
    } else {
        pq_wait_lgkm<0>(fa);
        pq_mma_group<G, MT>(acc + CB, fa, ah, al);
    }
}

// ---- persistent variant: operands swapped (A = staged X columns, B = weights), so that the accumulator holds
//      4 CONSECUTIVE OUTPUT COLUMNS per lane (D[m = column 4(l>>4)+r][n = row l&15]) and the epilogue stores straight
//      from registers -- no LDS staging, the ring is never drained
template <typename G, int MT>
__device__ __forceinline__ void pg_mma_group(f32x4 *acc, const s16x4 (&f)[2][2][2], const f16x8 (&ah)[MT], const f16x8 (&al)[MT]) {
    f16x8 bh[2], bl[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) { bh[u] = pq_cat(f[u][0][0], f[u][0][1]); bl[u] = pq_cat(f[u][1][0], f[u][1][1]); }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int u = 0; u < 2; ++u)
            acc[mt * G::NCB + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[u], ah[mt], acc[mt * G::NCB + u], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int u = 0; u < 2; ++u)
            acc[mt * G::NCB + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[u], ah[mt], acc[mt * G::NCB + u], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int u = 0; u < 2; ++u)
            acc[mt * G::NCB + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[u], al[mt], acc[mt * G::NCB + u], 0, 0, 0);
}
template <typename G, int MT, int CB>
__device__ __forceinline__ void pg_sweep(f32x4 *acc, s16x4 (&fa)[2][2][2], s16x4 (&fb)[2][2][2], const uint32_t (&addr)[8],
                                         const f16x8 (&ah)[MT], const f16x8 (&al)[MT]) {
    if constexpr (CB + 2 < G::NCB) {
        pq_read_group<G, CB + 2>(fb, addr);
        pq_wait_lgkm<8>(fa);
        pg_mma_group<G, MT>(acc + CB, fa, ah, al);
        pg_sweep<G, MT, CB + 2>(acc, fb, fa, addr, ah, al);
    } else {
        pq_wait_lgkm<0>(fa);
        pg_mma_group<G, MT>(acc + CB, fa, ah, al);
    }
}

// Persistent matrix-core pooling: ONE 512-thread workgroup per CU owns 16*MT*4 rows x 256 columns at a time (4 row
// groups x 2 column groups of waves: the two column groups share every staged weight fragment, which halves the
// weight traffic of the 128-column kernel above) and walks a strided list of (row block, column half) tiles: workgroup
// i of XCD label q = blockIdx & 7 takes tiles lo_q + i, lo_q + i + 32, ... of the label's contiguous tile range, so
// the 32 workgroups of an XCD work on 32 adjacent tiles at any time (the halo rows they share meet in that XCD's L2).
//
// The gather is latency-bound by the bytes a CU keeps in flight (Little: ~2 us issue->landed under load), and LDS is
// both the buffer and the in-flight space, so the two operand streams get rings of their own depth:
//   X ring  XD stages of 32 union rows x 256 columns x {hi, lo} (32 KiB), issued XD-1 steps ahead;
//   W ring  3 stages of the step's weight fragments (+ row ids), issued 2 steps ahead; W-stage g carries the row ids
//           of X-stage g + (XD-1), read back from LDS at step g when that X stage is issued.
// Both rings run THROUGH tile boundaries (the next tile's first stages are issued during the last steps of the
// current one) and the epilogue stores go out with the rings full: the accumulators hold 4 consecutive output columns
// per lane (operands swapped, see pg_mma_group), so they are stored straight from registers.  All waits are
// hand-counted (vector-memory operations complete in issue order): per step every wave issues [W (MT), ids (1),
// X (4)] = DPS operations, and at the end of step g it needs W(g+1), its ids and X(g+1) => at most
// 4 (XD-3) + DPS (+ the stores of an epilogue issued since) younger operations may be outstanding.
// Needs >= XD + 1 steps per tile (host-checked) and output buffers padded to whole row blocks (no store predicate =>
// a fixed number of vector-memory operations per wave).  Waves 4-7 issue their DMA after the sweep, so that the two
// waves of a SIMD alternate DMA issue and matrix work (knob bit 5 turns that off: 4-8 % slower).  Measured and left out:
// non-temporal weight loads, sc1 / nt output stores (all within run-to-run noise), a 4-deep X ring (no gain while the
// weight ring stays 2 steps ahead), G consecutive tiles per workgroup handed out by the dispatcher (slower: the 32
// workgroups of an XCD then work 32 G tiles apart and the halo rows no longer meet in L2).
// The epilogue goes through a wave-private LDS area (inline-asm LDS operations with their own lgkmcnt waits: a
// compiler-visible LDS access would wait for every outstanding LDS-DMA) so that each store instruction writes 4 rows x
// 256 contiguous bytes: the 32-byte pieces of a store straight from the accumulators cost 4x the L2 write requests,
// and the kernel runs at the L2 request rate.
template <int MT, int XD>
struct PgGeo {
    using Q = PqGeo<8, 256, MT, 2>;
    static constexpr int NWF = 4 * MT;                       // 16-row weight fragments per stage and plane
    static constexpr int XSTAGE = 2 * Q::PLANE;              // 32 KiB
    static constexpr int WD = 3;
    static constexpr int OFF_ID = 2 * NWF * 1024;            // inside a W stage
    static constexpr int WSTAGE = OFF_ID + 8 * 256;
    static constexpr int WRING = XD * XSTAGE;
    static constexpr int AHX = XD - 1;
    static constexpr int DPS = 5 + MT;                       // DMA instructions per wave and step: MT weights, 1 ids, 4 rows
    static constexpr int BASE_WAIT = 4 * (AHX - 2) + DPS;
    // epilogue staging: per wave 16 rows x 256 B (+16 B pitch skew) -- one f16 plane of its 16 x 128 tile, or half of the
    // fp32 tile -- so that the tile leaves in 256-byte row runs (2 full lines per request instead of 32-byte pieces)
    static constexpr int STG_PITCH = 272, STG_WAVE = 16 * STG_PITCH;
    static constexpr int OFF_STG = XD * XSTAGE + WD * WSTAGE;
    static constexpr size_t SMEM = (size_t)OFF_STG + 8 * STG_WAVE;
    static_assert(WSTAGE < 65536, "intra-stage LDS offsets are 16-bit immediates");
    static_assert(SMEM <= 160 * 1024, "rings + epilogue staging must fit the CU's LDS");
};

struct PgTile { int64_t b; int col0; int64_t ub0; int n; };
typedef int pg_i32x4 __attribute__((ext_vector_type(4)));

template <int OFF>
__device__ __forceinline__ void pg_wr64(uint32_t addr, f16x4 v) {
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void pg_wr128(uint32_t addr, f32x4 v) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void pg_rd128(f32x4 &d, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void pg_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void pg_lgkm0(f32x4 (&r)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3])::"memory");
}

template <int MT, int XD, bool F32OUT>
__global__ void __launch_bounds__(512)
pool_mfma_persist_kernel(const _Float16 *__restrict__ x_hi, const _Float16 *__restrict__ x_lo, int64_t ld_x,
                         const int64_t *__restrict__ bu_off, const int32_t *__restrict__ bu_row,
                         const _Float16 *__restrict__ wa_hi, const _Float16 *__restrict__ wa_lo, int64_t nblocks,
                         _Float16 *__restrict__ y_hi, _Float16 *__restrict__ y_lo, int64_t ld_y, float *__restrict__ y_f32,
                         int64_t ld_yf, const float *__restrict__ out_scale, unsigned *__restrict__ queue) {
    using P = PgGeo<MT, XD>;
    using G = typename P::Q;
    static_assert(MT == 1, "the staged epilogue is written for 16 rows x 128 columns per wave");
    constexpr int NWF = P::NWF, AHX = P::AHX, NSTORE = 8;      // per wave and tile: 2 passes x 4 row-run stores
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // ---- this workgroup's tiles.  XCD label q = blockIdx & 7 owns the contiguous tile range [lo, hi); workgroup wi of the label
    // starts with tile lo + wi.  queue != nullptr: every further tile is CLAIMED from the label's counter (queue[q], zero at
    // launch; claim c -> tile lo + W + c) one tile ahead, so the label's W workgroups always work on the W lowest unfinished
    // tiles -- what the hardware dispatcher does for a one-tile-per-workgroup grid, and what keeps the halo rows of
    // neighbouring tiles meeting in the XCD's L2.  queue == nullptr: the static list lo + wi, lo + wi + W, ... (workgroups
    // drift apart by tens of microseconds over the ~60 rounds of a launch; X-row L2 hits 54 % instead of 70 %).
    // The claim is asynchronous so that the rings never drain: wave 0 issues the atomic at step 0 of a tile (complete
    // after the hand-over of step 1), the load of the claimed tile's two union offsets at step 2 (complete after the
    // hand-over of step 3), writes {tile, steps, first union row} into an LDS slot at step 4 and every wave reads it at
    // step 5 -- the first step that can need the next tile (its row ids ride in W stages from step n - 4 on, n >= 9).
    // Extra vector-memory operations in wave 0's queue only make its hand-counted waits stricter.
    const int64_t T = nblocks * 2;
    const int label = blockIdx.x & 7, wi = blockIdx.x >> 3, W = (int)(gridDim.x >> 3);
    const int64_t lo = label * T / 8, hi = (label + 1) * T / 8;
    const bool dynamic = queue != nullptr;
    auto leave = [&]() {                                     // the last workgroup out re-arms the counters for the next launch
        if (dynamic && tid == 0) {
            const unsigned done = atomicAdd(queue + 8, 1u);
            if (done == gridDim.x - 1) {
#pragma unroll
                for (int q = 0; q < 9; ++q) queue[q] = 0u;
            }
        }
    };
    int64_t t = lo + wi;
    if (t >= hi) { leave(); return; }
    auto load_tile = [&](int64_t tt) {
        PgTile r;
        r.b = tt >> 1;
        r.col0 = (int)(tt & 1) * 256;
        r.ub0 = bu_off[r.b];
        r.n = (int)((bu_off[r.b + 1] - r.ub0) / PM_KS);
        return r;
    };
    PgTile cur = load_tile(t);
    PgTile nx1 = cur;
    bool has1 = false;
    int64_t claimed = 0;                                     // wave 0: the tile its last claim returned
    const bool late_issue = wv >= 4;

    // ---- DMA roles (same staging geometry as PqGeo<8,256,..>: 4 rows per wave, two 1-KiB instructions per plane)
    const int du = lane / (64 / G::RPI), dc = lane % (G::RB / 16);
    const int row_i0 = G::RPW * wv + 2 * du;
    const int t0 = (row_i0 & 3) | (((row_i0 >> 3) & 1) << 2), t1 = ((row_i0 + 1) & 3) | ((((row_i0 + 1) >> 3) & 1) << 2);
    const int dcol0 = (dc ^ (2 * t0)) * 8, dcol1 = (dc ^ (2 * t1)) * 8;
    auto issue_x = [&](const PgTile &TT, int2 id, int xslot) {
        unsigned char *dst = smem_raw + xslot * P::XSTAGE;
        const int64_t s0 = (int64_t)id.x * ld_x + TT.col0 + dcol0, s1 = (int64_t)id.y * ld_x + TT.col0 + dcol1;
        glds16(x_hi + s0, dst + (G::RPW * wv) * G::RB);
        glds16(x_lo + s0, dst + G::PLANE + (G::RPW * wv) * G::RB);
        glds16(x_hi + s1, dst + (G::RPW * wv) * G::RB + 1024);
        glds16(x_lo + s1, dst + G::PLANE + (G::RPW * wv) * G::RB + 1024);
    };
    // W stage (TT, k) -> W-ring slot, with the row ids of (IT, kid)
    auto issue_w = [&](const PgTile &TT, int k, const PgTile &IT, int kid, int wslot) {
        unsigned char *dst = smem_raw + P::WRING + wslot * P::WSTAGE;
        const int64_t ks = TT.ub0 / PM_KS + k;
        const int32_t *ids = bu_row + IT.ub0 + G::RPW * wv + (int64_t)kid * PM_KS + (lane & (G::RPW - 1));
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int L = wv + 8 * i, plane = L / NWF, frag = L % NWF;
            const _Float16 *src = (plane ? wa_lo : wa_hi) + ((ks * NWF + frag) * 64 + lane) * 8;
            glds16(src, dst + L * 1024);
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)ids,
                                         (__attribute__((address_space(3))) void *)(dst + P::OFF_ID + wv * 256), 4, 0, 0);
    };

    // ---- read roles
    const int rg = wv % G::RGN, cg = wv / G::RGN;
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem_raw;
    uint32_t addr[8];
    {
        const int r = 8 * g + q, r_w = r % G::RPW;
        const int slot = (r / G::RPW) * G::RPW + G::RPI * (r_w & 1) + (r_w >> 1);
        const uint32_t rowb = (uint32_t)slot * G::RB + (uint32_t)(cg * G::WC * 2) + (uint32_t)((p >> 1) * 16 + (p & 1) * 8);
        const uint32_t tt = (uint32_t)(q | ((g & 1) << 2));
#pragma unroll
        for (int k = 0; k < 8; ++k) addr[k] = lds0 + ((rowb + 32u * k) ^ (tt << 5));
    }
    const uint32_t addr_w = lds0 + P::WRING + (MT * rg) * 1024 + lane * 16;
    const uint32_t addr_id = lds0 + P::WRING + P::OFF_ID + wv * 256 + du * 8;
    const uint32_t stg = lds0 + P::OFF_STG + wv * P::STG_WAVE;                          // epilogue staging, wave-private
    const uint32_t rd = stg + (lane >> 4) * P::STG_PITCH + (lane & 15) * 16;
    const uint32_t slot = lds0 + P::OFF_STG + 256;           // 16 bytes of pitch padding in wave 0's staging row 0: the next-tile slot

    f32x4 acc[MT * G::NCB];
#pragma unroll
    for (int i = 0; i < MT * G::NCB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: X stages 0 .. AHX-1 and W stages 0, 1 of the first tile (every tile has >= AHX + 2 steps)
    {
        const int32_t *idg = bu_row + cur.ub0 + G::RPW * wv;
        int2 ip[AHX];
#pragma unroll
        for (int j = 0; j < AHX; ++j) ip[j] = *reinterpret_cast<const int2 *>(idg + j * PM_KS + 2 * du);
#pragma unroll
        for (int j = 0; j < AHX; ++j) asm volatile("" ::"v"(ip[j].x), "v"(ip[j].y));   // the id loads land before the first DMA
        issue_w(cur, 0, cur, AHX, 0);
        issue_x(cur, ip[0], 0);
        issue_w(cur, 1, cur, AHX + 1, 1);
        issue_x(cur, ip[1], 1);
        if constexpr (AHX == 3) issue_x(cur, ip[2], 2);
        pq_handover<P::BASE_WAIT>();
    }
    // split outputs stay in the pre-scaled domain of the planes (pooling is linear); only the fp32 output is scaled back
    const float inv = (1.f / PM_WSCALE) * ((F32OUT && out_scale) ? out_scale[0] : 1.f);
    const int fl = lane & 15, fq = lane >> 4;
    s16x4 f0[2][2][2], f1[2][2][2];
    f16x8 ah[MT], al[MT];
    int2 idn;
    int xs = 0, ws = 0;
    bool first_tile = true;
    for (;;) {
        const int n = cur.n;
        for (int s = 0; s < n; ++s) {
            const uint32_t xoff = (uint32_t)xs * P::XSTAGE, woff = (uint32_t)ws * P::WSTAGE;
            if (s <= 5) {                                    // the next tile (see the head of the kernel)
                if (dynamic) {
                    if (wv == 0) {
                        // The two asynchronous results land in v200 / v[202:203]: registers named here and nowhere else, far
                        // above what the compiler allocates for this kernel (~122; a compiler-chosen destination would be
                        // copied or reused between the issue and the arrival two steps later).
                        if (s == 0) {
                            const unsigned *qa = queue + label;
                            const unsigned one = 1u;
                            if (lane == 0)                                // ONE atomic per claim: same-address atomics serialise
                                asm volatile("global_atomic_add v200, %0, %1, off sc0" ::"v"(qa), "v"(one) : "memory", "v200");
                        } else if (s == 2) {
                            uint32_t c;
                            asm volatile("v_readfirstlane_b32 %0, v200" : "=s"(c)::"memory");
                            claimed = lo + W + (int64_t)c;
                            int64_t cb = claimed >> 1;
                            cb = cb < nblocks ? cb : nblocks - 1;
                            const int64_t *da = bu_off + cb + (lane & 1);
                            asm volatile("global_load_dwordx2 v[202:203], %0, off" ::"v"(da) : "memory", "v202", "v203");
                        } else if (s == 4) {
                            uint32_t d0l, d0h, d1l, d1h;
                            asm volatile("v_readlane_b32 %0, v202, 0\n\tv_readlane_b32 %1, v203, 0\n\t"
                                         "v_readlane_b32 %2, v202, 1\n\tv_readlane_b32 %3, v203, 1"
                                         : "=s"(d0l), "=s"(d0h), "=s"(d1l), "=s"(d1h)::"memory");
                            const int64_t u0 = (int64_t)(((uint64_t)d0h << 32) | d0l), u1 = (int64_t)(((uint64_t)d1h << 32) | d1l);
                            pg_i32x4 sv;                   // (an int vector: bit casts of float-vector elements are miscompiled)
                            sv[0] = claimed < hi ? (int32_t)(claimed - lo) : (int32_t)-1;
                            sv[1] = (int32_t)((u1 - u0) / PM_KS);
                            sv[2] = (int32_t)d0l;
                            sv[3] = (int32_t)d0h;
                            asm volatile("ds_write_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(slot), "v"(sv) : "memory");
                        }
                    }
                    if (s == 5) {
                        pg_i32x4 sv;
                        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(sv) : "v"(slot) : "memory");
                        const int32_t rel = __builtin_amdgcn_readfirstlane(sv[0]);
                        has1 = rel >= 0;
                        if (has1) {
                            const int64_t tt = lo + rel;
                            nx1.b = tt >> 1;
                            nx1.col0 = (int)(tt & 1) * 256;
                            nx1.n = __builtin_amdgcn_readfirstlane(sv[1]);
                            nx1.ub0 = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(sv[3]) << 32) |
                                                (uint32_t)__builtin_amdgcn_readfirstlane(sv[2]));
                        }
                    }
                } else if (s == 5) {
                    has1 = t + W < hi;
                    if (has1) nx1 = load_tile(t + W);
                }
            }
            uint32_t a[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = addr[k] + xoff;
            pq_rd64(idn, addr_id + woff);
            pq_rd128<0>(ah[0], addr_w + woff);
            pq_rd128<NWF * 1024>(al[0], addr_w + woff);
            if constexpr (MT == 2) {
                pq_rd128<1024>(ah[1], addr_w + woff);
                pq_rd128<NWF * 1024 + 1024>(al[1], addr_w + woff);
            }
            pq_read_group<G, 0>(f0, a);
            pq_wait_lgkm3<8>(idn, ah, al);
            const int xs2 = xs + AHX >= XD ? xs + AHX - XD : xs + AHX;
            const int ws2 = ws + 2 >= P::WD ? ws + 2 - P::WD : ws + 2;
            const int kx = s + AHX, kw = s + 2;
            const bool x_any = kx < n || has1, w_any = kw < n || has1;
            auto do_issue = [&]() {
                // W stage two steps ahead (this tile's, or step 0 / 1 of the next tile) with the row ids of the X stage that
                // will be issued when it is read; X stage AHX steps ahead.  Wave-uniform selects instead of code copies.
                const bool wc = kw < n;
                const PgTile &TW = wc ? cur : nx1;
                const int kwl = wc ? kw : kw - n;
                int ki = kwl + AHX;
                const bool id_next = wc && ki >= n && has1;            // ids of the next tile's first stages ride in this tile's last ones
                const PgTile &TI = id_next ? nx1 : TW;
                ki = id_next ? ki - n : (ki < TW.n ? ki : TW.n - 1);
                if (wc || has1) issue_w(TW, kwl, TI, ki, ws2);
                const bool xc = kx < n;
                if (xc || has1) issue_x(xc ? cur : nx1, idn, xs2);
            };
            if (!late_issue) do_issue();
            pg_sweep<G, MT, 0>(acc, f0, f1, a, ah, al);
            if (late_issue) do_issue();
            const bool last = s == n - 1;
            if (last) {
                // epilogue: lane (fl, fq) holds columns 16 cb + 4 fq .. +3 of row fl.  The wave's 16 x 128 tile is transposed
                // through its private LDS area one plane (or half an fp32 tile) at a time and leaves in 256-byte row runs:
                // lane l stores 16 bytes of row (l >> 4) + 4 pass, 16-byte chunk l & 15.
                const int64_t row0 = cur.b * G::BR + rg * 16 + (lane >> 4);
                const int colw = cur.col0 + cg * G::WC;
                f32x4 r[4];
                if constexpr (F32OUT) {
                    const uint32_t wr = stg + fl * P::STG_PITCH + fq * 16;
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        pg_wr128<0>(wr, acc[half * 4 + 0] * inv);
                        pg_wr128<64>(wr, acc[half * 4 + 1] * inv);
                        pg_wr128<128>(wr, acc[half * 4 + 2] * inv);
                        pg_wr128<192>(wr, acc[half * 4 + 3] * inv);
                        pg_lgkm0();
                        pg_rd128<0>(r[0], rd);
                        pg_rd128<4 * P::STG_PITCH>(r[1], rd);
                        pg_rd128<8 * P::STG_PITCH>(r[2], rd);
                        pg_rd128<12 * P::STG_PITCH>(r[3], rd);
                        pg_lgkm0(r);
#pragma unroll
                        for (int ps = 0; ps < 4; ++ps)
                            *reinterpret_cast<f32x4 *>(y_f32 + (row0 + 4 * ps) * ld_yf + colw + half * 64 + (lane & 15) * 4) = r[ps];
                    }
                } else {
                    const uint32_t wr = stg + fl * P::STG_PITCH + fq * 8;
                    f16x4 h[G::NCB], l[G::NCB];
#pragma unroll
                    for (int cb = 0; cb < G::NCB; ++cb) {
                        const f32x4 v = acc[cb] * inv;
#pragma unroll
                        for (int i = 0; i < 4; ++i) { h[cb][i] = (_Float16)v[i]; l[cb][i] = (_Float16)(v[i] - (float)h[cb][i]); }
                    }
#pragma unroll
                    for (int plane = 0; plane < 2; ++plane) {
                        const f16x4 *src = plane ? l : h;
                        pg_wr64<0>(wr, src[0]);   pg_wr64<32>(wr, src[1]);  pg_wr64<64>(wr, src[2]);  pg_wr64<96>(wr, src[3]);
                        pg_wr64<128>(wr, src[4]); pg_wr64<160>(wr, src[5]); pg_wr64<192>(wr, src[6]); pg_wr64<224>(wr, src[7]);
                        pg_lgkm0();
                        pg_rd128<0>(r[0], rd);
                        pg_rd128<4 * P::STG_PITCH>(r[1], rd);
                        pg_rd128<8 * P::STG_PITCH>(r[2], rd);
                        pg_rd128<12 * P::STG_PITCH>(r[3], rd);
                        pg_lgkm0(r);
                        _Float16 *yp = plane ? y_lo : y_hi;
#pragma unroll
                        for (int ps = 0; ps < 4; ++ps)
                            *reinterpret_cast<f32x4 *>(yp + (row0 + 4 * ps) * ld_y + colw + (lane & 15) * 8) = r[ps];
                    }
                }
#pragma unroll
                for (int cb = 0; cb < G::NCB; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            // hand-over.  Steady state: [W, ids, X] of this step are the youngest DPS operations, before them the X stage of
            // the previous step (AHX = 3) and the stores of an epilogue issued since the stage being waited for.
            if (x_any && w_any) {
                const bool stores_since = last || (s == 0 && !first_tile);
                if (stores_since) pq_handover<P::BASE_WAIT + NSTORE>(); else pq_handover<P::BASE_WAIT>();
            } else {
                pq_handover<0>();                  // the last steps of the workgroup's last tile: nothing left to overlap
            }
            xs = xs + 1 >= XD ? 0 : xs + 1;
            ws = ws + 1 >= P::WD ? 0 : ws + 1;
        }
        if (!has1) break;
        first_tile = false;
        t = nx1.b * 2 + (nx1.col0 >> 8);
        cur = nx1;
        has1 = false;                                          // known again from step 5 of the new tile
    }
    leave();
}

// TUNE: the tuning bits of `ablate_` are honoured; the PRODUCT instantiation (TUNE = false) compiles them out.
// Round-3 finding (three memory access faults in scripts/bench_pool.py, masks 8 / 8 / 9): the bits used to be live in the product
// kernel and bits 1 / 3 DROPPED LDS-DMA instructions from a stage while the hand-over kept waiting `vmcnt(DMA_PER_STAGE)`.  With
// bit 3 a stage had 5 instructions instead of 7, so the wait released the barrier with the two youngest instructions of the OLDER
// stage still in flight -- one of them the stage's row-id load (issued last).  pq_rd64 then read stale ids from the ring slot
// and the next issue() gathered from garbage row numbers.  Now a tuning bit never changes the NUMBER of DMA instructions: it
// replaces the instruction's source by one hot 16-byte piece (all lanes the same address), so every hand-counted wait keeps
// its meaning in every mask.
template <int NW, int NC, int MT, int CGN, bool STAMP, bool TUNE>
__device__ __forceinline__ void
pq_body(const _Float16 *__restrict__ x_hi, const _Float16 *__restrict__ x_lo, int64_t ld_x,
        const int64_t *__restrict__ bu_off, const int32_t *__restrict__ bu_row,
        const _Float16 *__restrict__ wa_hi, const _Float16 *__restrict__ wa_lo, int64_t nv, int64_t nblocks,
        _Float16 *__restrict__ y_hi, _Float16 *__restrict__ y_lo, int64_t ld_y, float *__restrict__ y_f32,
        int64_t ld_yf, int64_t per_xcd, int ablate_, const float *__restrict__ out_scale, uint64_t *__restrict__ stamp) {
    using G = PqGeo<NW, NC, MT, CGN>;
    const int ablate = TUNE ? ablate_ : 0;
    static_assert(G::BR / 16 == NW, "one weight fragment group per wave to stage");
    uint64_t st_t0 = 0, st_r0 = 0, st_pro = 0, st_work = 0, st_wait = 0, st_issue = 0;
    if constexpr (STAMP) { st_t0 = pq_now(); st_r0 = pq_real(); }
    constexpr int NQ = PM_D / NC;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t lb = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);    // XCD-contiguous order
    const int64_t b = lb / NQ;
    const int col0 = (int)(lb % NQ) * NC;
    if (b >= nblocks) return;
    const int64_t ub0 = bu_off[b];
    const int n = (int)((bu_off[b + 1] - ub0) / PM_KS);                            // steps (>= 1)
    const int64_t ks0 = ub0 / PM_KS;

    // ---- DMA roles
    const int du = lane / (64 / G::RPI), dc = lane % (G::RB / 16);
    const int row_i0 = G::RPW * wv + 2 * du;                                         // instruction 0; instruction 1: +1
    const int t0 = (row_i0 & 3) | (((row_i0 >> 3) & 1) << 2), t1 = ((row_i0 + 1) & 3) | ((((row_i0 + 1) >> 3) & 1) << 2);
    const int64_t dsrc0 = col0 + ((dc ^ (2 * t0)) * 8);
    const int64_t dsrc1 = col0 + ((dc ^ (2 * t1)) * 8);
    const int32_t *idg = bu_row + ub0 + G::RPW * wv;                               // this wave's row ids, step 0
    const _Float16 *wah = wa_hi + ((ks0 * NW + wv) * 64 + lane) * 8;
    const _Float16 *wal = wa_lo + ((ks0 * NW + wv) * 64 + lane) * 8;
    constexpr int64_t WSTEP = (int64_t)NW * 64 * 8;
    // stage k -> ring slot: rows(k), weights(k), ids(min(k+2, n-1))
    auto issue = [&](int2 id, int k, int slot) {
        unsigned char *dst = smem_raw + slot * G::STAGE;
        // tuning aids (TUNE only): bit 1 fetches one hot piece instead of the gathered rows, bit 3 instead of the weight
        // fragments -- the same 7 instructions per stage either way (see the note above the kernel)
        const bool hot_x = (ablate & 2) != 0, hot_w = (ablate & 8) != 0;
        const int64_t s0 = hot_x ? 0 : (int64_t)id.x * ld_x + dsrc0, s1 = hot_x ? 0 : (int64_t)id.y * ld_x + dsrc1;
        glds16(x_hi + s0, dst + (G::RPW * wv) * G::RB);
        glds16(x_lo + s0, dst + G::PLANE + (G::RPW * wv) * G::RB);
        glds16(x_hi + s1, dst + (G::RPW * wv) * G::RB + 1024);
        glds16(x_lo + s1, dst + G::PLANE + (G::RPW * wv) * G::RB + 1024);
        glds16(hot_w ? wa_hi : wah + (int64_t)k * WSTEP, dst + G::OFF_W + wv * 1024);
        glds16(hot_w ? wa_lo : wal + (int64_t)k * WSTEP, dst + G::OFF_W + NW * 1024 + wv * 1024);
        const int kid = k + 2 < n ? k + 2 : n - 1;
        // 64 lanes x 4 B: the wave's RPW ids, repeated (lane & (RPW-1)): always inside the block's padded union
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void *)(idg + (int64_t)kid * PM_KS + (lane & (G::RPW - 1))),
            (__attribute__((address_space(3))) void *)(dst + G::OFF_ID + wv * 256), 4, 0, 0);
    };

    // ---- read roles
    const int rg = wv % G::RGN, cg = wv / G::RGN;                                   // this wave's rows / columns
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem_raw;
    uint32_t addr[8];
    {
        const int r = 8 * g + q, r_w = r % G::RPW;
        const int slot = (r / G::RPW) * G::RPW + G::RPI * (r_w & 1) + (r_w >> 1);
        const uint32_t rowb = (uint32_t)slot * G::RB + (uint32_t)(cg * G::WC * 2) + (uint32_t)((p >> 1) * 16 + (p & 1) * 8);
        const uint32_t t = (uint32_t)(q | ((g & 1) << 2));
#pragma unroll
        for (int k = 0; k < 8; ++k) addr[k] = lds0 + ((rowb + 32u * k) ^ (t << 5));
    }
    const uint32_t addr_w = lds0 + G::OFF_W + (MT * rg) * 1024 + lane * 16;
    const uint32_t addr_id = lds0 + G::OFF_ID + wv * 256 + du * 8;

    f32x4 acc[MT * G::NCB];
#pragma unroll
    for (int i = 0; i < MT * G::NCB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: stages 0 and 1 in flight (a one-step block stages its only step twice: no branches here,
    //      so that the compiler's own wait for the two id loads sits before the first DMA and nowhere else)
    {
        const int k1 = n > 1 ? 1 : 0;
        const int2 i0 = *reinterpret_cast<const int2 *>(idg + 2 * du);
        const int2 i1 = *reinterpret_cast<const int2 *>(idg + k1 * PM_KS + 2 * du);
        asm volatile("" ::"v"(i0.x), "v"(i0.y), "v"(i1.x), "v"(i1.y));    // both id loads land before the first DMA
        issue(i0, 0, 0);
        issue(i1, k1, 1);
        pq_handover<G::DMA_PER_STAGE>();
    }
    if constexpr (STAMP) st_pro = pq_now();
    s16x4 f0[2][2][2], f1[2][2][2];
    f16x8 ah[MT], al[MT];
    int2 idn;
    for (int s0 = 0; s0 < n; s0 += G::NST) {
#pragma unroll
        for (int J = 0; J < G::NST; ++J) {
            const int s = s0 + J;
            if (s < n) {
                uint32_t a[8];
                uint64_t st_a = 0, st_b = 0;
                if constexpr (STAMP) st_a = pq_now();
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] = addr[k] + J * G::STAGE;
                pq_rd64(idn, addr_id + J * G::STAGE);
                pq_rd128<0>(ah[0], addr_w + J * G::STAGE);
                pq_rd128<NW * 1024>(al[0], addr_w + J * G::STAGE);
                if constexpr (MT >= 2) {
                    pq_rd128<1024>(ah[1], addr_w + J * G::STAGE);
                    pq_rd128<NW * 1024 + 1024>(al[1], addr_w + J * G::STAGE);
                }
                if constexpr (MT == 4) {                   // a wave that owns all 64 rows reads all four weight fragments
                    pq_rd128<2048>(ah[2], addr_w + J * G::STAGE);
                    pq_rd128<NW * 1024 + 2048>(al[2], addr_w + J * G::STAGE);
                    pq_rd128<3072>(ah[3], addr_w + J * G::STAGE);
                    pq_rd128<NW * 1024 + 3072>(al[3], addr_w + J * G::STAGE);
                    pq_wait_lgkm3<0>(idn, ah, al);         // 9 + 8 reads would overflow the 4-bit LDS counter
                    pq_read_group<G, 0>(f0, a);
                } else {
                    pq_read_group<G, 0>(f0, a);
                    pq_wait_lgkm3<8>(idn, ah, al);
                }
                if (s + 2 < n) issue(idn, s + 2, (J + 2) % G::NST);
                if constexpr (STAMP) if (ablate & 64) { pq_wait_lgkm<0>(f0); st_issue += pq_now() - st_a; }
                if (!(ablate & 1)) pq_sweep<G, MT, 0>(acc, f0, f1, a, ah, al);   // tuning aid: bit 0 skips reads + MFMAs
                else pq_wait_lgkm<0>(f0);
                if constexpr (STAMP) { st_b = pq_now(); st_work += st_b - st_a; }
                if (s + 2 < n) pq_handover<G::DMA_PER_STAGE>(); else pq_handover<0>();
                if constexpr (STAMP) st_wait += pq_now() - st_b;
            }
        }
    }
    if (ablate & 4) return;                                // tuning aid: bit 2 skips the epilogue
    uint64_t st_e0 = 0;
    if constexpr (STAMP) st_e0 = pq_now();
    // ---- epilogue through LDS (the ring is drained: the last hand-over waited for vmcnt(0))
    // the split planes carry x * s (s = the power of two of gp_pow2_scale, so that the lo halves stay normal f16 numbers);
    // pooling is linear, so the planes written for the next application stay in that domain and only the fp32 output is
    // multiplied by out_scale = 1/s
    const float inv = 1.f / PM_WSCALE;
    float *st = reinterpret_cast<float *>(smem_raw) + wv * (MT * 16 * G::EP);
    const int fl = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int cb = 0; cb < G::NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[(mt * 16 + fq * 4 + r) * G::EP + cb * 16 + fl] = acc[mt * G::NCB + cb][r] * inv;
    gp_wave_sync();
    const int64_t row0 = b * G::BR + rg * (16 * MT);
    const int colw = col0 + cg * G::WC;
    constexpr int C4 = G::WC / 4;                                       // float4 per row
    constexpr int NIT = MT * 16 * C4 / 64;
    // all LDS reads first, then the stores: the compiler orders every LDS read after ALL outstanding vector
    // memory operations (it cannot tell stores from LDS-DMA), so interleaving would serialise the stores
    float4 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = it * 64 + lane;
        v[it] = *reinterpret_cast<const float4 *>(st + (idx / C4) * G::EP + (idx % C4) * 4);
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = it * 64 + lane;
        const int row = idx / C4, c4 = idx % C4;
        const int64_t grow = row0 + row;
        if (grow < nv && !(ablate & 16)) {      // tuning aid: bit 4 skips the output stores
            float xv[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
            f16x4 h, l;
#pragma unroll
            for (int i = 0; i < 4; ++i) { h[i] = (_Float16)xv[i]; l[i] = (_Float16)(xv[i] - (float)h[i]); }
            if (y_hi) {
                *reinterpret_cast<f16x4 *>(y_hi + grow * ld_y + colw + c4 * 4) = h;
                *reinterpret_cast<f16x4 *>(y_lo + grow * ld_y + colw + c4 * 4) = l;
            }
            if (y_f32) {
                const float so = out_scale ? out_scale[0] : 1.f;
                *reinterpret_cast<float4 *>(y_f32 + grow * ld_yf + colw + c4 * 4) = make_float4(v[it].x * so, v[it].y * so, v[it].z * so, v[it].w * so);
            }
        }
    }
    if constexpr (STAMP) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint64_t t3 = pq_now(), r3 = pq_real();
        if (lane == 0 && stamp) {
            uint64_t *o = stamp + ((int64_t)blockIdx.x * NW + wv) * 10;
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            o[0] = st_r0; o[1] = r3 - st_r0; o[2] = st_pro - st_t0; o[3] = st_work; o[4] = st_wait; o[5] = st_issue;
            o[6] = t3 - st_e0; o[7] = t3 - st_t0; o[8] = (uint64_t)n; o[9] = xcc;
        }
    }
}

#define PQ_PARAMS const _Float16 *__restrict__ x_hi, const _Float16 *__restrict__ x_lo, int64_t ld_x, const int64_t *__restrict__ bu_off, \
                  const int32_t *__restrict__ bu_row, const _Float16 *__restrict__ wa_hi, const _Float16 *__restrict__ wa_lo, int64_t nv, \
                  int64_t nblocks, _Float16 *__restrict__ y_hi, _Float16 *__restrict__ y_lo, int64_t ld_y, float *__restrict__ y_f32,   \
                  int64_t ld_yf, int64_t per_xcd, int ablate, const float *__restrict__ out_scale, uint64_t *__restrict__ stamp
#define PQ_FWD x_hi, x_lo, ld_x, bu_off, bu_row, wa_hi, wa_lo, nv, nblocks, y_hi, y_lo, ld_y, y_f32, ld_yf, per_xcd, ablate, out_scale, stamp
// the product kernel: no tuning bits, no stamps
template <int NW, int NC, int MT, int CGN>
__global__ void __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) pool_mfma_kernel(PQ_PARAMS) { pq_body<NW, NC, MT, CGN, false, false>(PQ_FWD); }
// the same body with the tuning bits live (and, STAMP, the in-kernel time stamps), under its own name in a kernel trace
template <int NW, int NC, int MT, int CGN, bool STAMP>
__global__ void __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) pool_mfma_tuning_kernel(PQ_PARAMS) { pq_body<NW, NC, MT, CGN, STAMP, true>(PQ_FWD); }
#undef PQ_PARAMS
#undef PQ_FWD

size_t pm_scan_tmp(int64_t n) {
    size_t t = 0;
    (void)rocprim::exclusive_scan(nullptr, t, (int64_t *)nullptr, (int64_t *)nullptr, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), 0);
    return t;
}

int pm_np2(int64_t n) { int p = 1; while (p < n) p <<= 1; return p; }

template <int NW, int NC, int MT, int CGN>
int pm_launch(const void *x_hi, const void *x_lo, int64_t ld_x, const int64_t *bu_off, const int32_t *bu_row, const void *wa_hi,
              const void *wa_lo, int64_t nv, void *y_hi, void *y_lo, int64_t ld_y, float *y_f32, int64_t ld_yf, const float *out_scale,
              hipStream_t s) {
    using G = PqGeo<NW, NC, MT, CGN>;
    int64_t nb = (nv + G::BR - 1) / G::BR;
    int64_t per_xcd = (nb * (PM_D / NC) + 7) / 8;
    // tuning aid (knob 9): one workgroup per CU by asking for more LDS than two workgroups could share
    const size_t smem = (g_gp_knobs[9] > 0 && G::SMEM < 90 * 1024) ? 90 * 1024 : G::SMEM;
    uint64_t *stamp = static_cast<uint64_t *>(g_gp_debug_ptr[0]);
    GP_CHECK_ARG(!stamp || g_gp_debug_bytes[0] >= (size_t)(per_xcd * 8) * NW * 10 * sizeof(uint64_t),
                 "gp_pool_mfma_apply: the stamp buffer of gp_debug_ptr(0) holds %zu bytes, this launch writes %zu",
                 g_gp_debug_bytes[0], (size_t)(per_xcd * 8) * NW * 10 * sizeof(uint64_t));
#define PM_ARGS static_cast<const _Float16 *>(x_hi), static_cast<const _Float16 *>(x_lo), ld_x, bu_off, bu_row,                        \
                static_cast<const _Float16 *>(wa_hi), static_cast<const _Float16 *>(wa_lo), nv, nb, static_cast<_Float16 *>(y_hi),     \
                static_cast<_Float16 *>(y_lo), ld_y, y_f32, ld_yf, per_xcd, g_gp_knobs[4], out_scale, stamp
    if (stamp) {                                          // tuning aid: the instantiation with in-kernel time stamps
        GP_SMEM_ATTR((pool_mfma_tuning_kernel<NW, NC, MT, CGN, true>), 90 * 1024 > G::SMEM ? 90 * 1024 : G::SMEM);
        pool_mfma_tuning_kernel<NW, NC, MT, CGN, true><<<(unsigned)(per_xcd * 8), NW * 64, smem, s>>>(PM_ARGS);
    } else if (g_gp_knobs[4] != 0) {                      // tuning aid: parts of the kernel switched off (never the product kernel)
        GP_SMEM_ATTR((pool_mfma_tuning_kernel<NW, NC, MT, CGN, false>), 90 * 1024 > G::SMEM ? 90 * 1024 : G::SMEM);
        pool_mfma_tuning_kernel<NW, NC, MT, CGN, false><<<(unsigned)(per_xcd * 8), NW * 64, smem, s>>>(PM_ARGS);
    } else {
        GP_SMEM_ATTR((pool_mfma_kernel<NW, NC, MT, CGN>), 90 * 1024 > G::SMEM ? 90 * 1024 : G::SMEM);
        pool_mfma_kernel<NW, NC, MT, CGN><<<(unsigned)(per_xcd * 8), NW * 64, smem, s>>>(PM_ARGS);
    }
#undef PM_ARGS
    GP_CHECK_LAUNCH();
    return GP_OK;
}

template <int MT, int XD, bool F32OUT>
int pg_launch(const void *x_hi, const void *x_lo, int64_t ld_x, const int64_t *bu_off, const int32_t *bu_row, const void *wa_hi,
              const void *wa_lo, int64_t nblocks, void *y_hi, void *y_lo, int64_t ld_y, float *y_f32, int64_t ld_yf,
              const float *out_scale, unsigned *queue, hipStream_t s) {
    using P = PgGeo<MT, XD>;
    GP_SMEM_ATTR((pool_mfma_persist_kernel<MT, XD, F32OUT>), P::SMEM);
    const int n_cu = gp_cu_count();
    GP_CHECK_ARG(n_cu > 0, "gp_pool_mfma_apply_persistent: cannot read the device's compute-unit count");
    int per_label = g_gp_knobs[10] > 0 ? g_gp_knobs[10] : (n_cu >= 8 ? n_cu / 8 : 1);      // workgroups per XCD label
    if (g_gp_knobs[12] == 1) queue = nullptr;                                              // tuning aid: the static tile lists
    pool_mfma_persist_kernel<MT, XD, F32OUT><<<(unsigned)(per_label * 8), 512, P::SMEM, s>>>(
        static_cast<const _Float16 *>(x_hi), static_cast<const _Float16 *>(x_lo), ld_x, bu_off, bu_row,
        static_cast<const _Float16 *>(wa_hi), static_cast<const _Float16 *>(wa_lo), nblocks, static_cast<_Float16 *>(y_hi),
        static_cast<_Float16 *>(y_lo), ld_y, y_f32, ld_yf, out_scale, queue);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

}  // namespace

// Persistent variant (one 512-thread workgroup per CU, 256 columns per workgroup, ring kept full across row blocks).
// Requirements beyond gp_pool_mfma_apply: every row block has at least 9 steps (min_steps, from the builder's bu_off:
// min over blocks of (bu_off[b+1]-bu_off[b])/32; a block of 64 rows with 96 neighbours each has 13 on scenes), and the
// OUTPUT buffers hold y_rows >= ceil(nv / block_rows) * block_rows rows (rows >= nv receive zeros).  Exactly one of
// (y_hi, y_lo) / y_f32.  queue: 9 x uint32 of device memory, ZERO at the first launch and left zero by every launch (the
// tile counters of the 8 XCD labels + a finished-workgroup counter); launches that share a queue must be stream-ordered.
// NULL selects static tile lists (slower: see the kernel's comment).
// out_scale: optional device scalar multiplied into the fp32 output (power-of-two pre-scaling of the split operands).
extern "C" int gp_pool_mfma_apply_persistent(const void *x_hi, const void *x_lo, int64_t ld_x, const int64_t *bu_off,
                                             const int32_t *bu_row, const void *wa_hi, const void *wa_lo, int64_t nv, int32_t d,
                                             int32_t block_rows, int32_t min_steps, void *y_hi, void *y_lo, int64_t ld_y,
                                             float *y_f32, int64_t ld_yf, int64_t y_rows, const float *out_scale, uint32_t *queue,
                                             void *stream_) {
    GP_CHECK_ARG(x_hi && x_lo && bu_off && bu_row && wa_hi && wa_lo && nv > 0, "gp_pool_mfma_apply_persistent: null/empty argument");
    GP_CHECK_ARG(d == PM_D, "gp_pool_mfma_apply_persistent: d=%d (kernel specialised for %d columns)", d, PM_D);
    GP_CHECK_ARG(block_rows == 64, "gp_pool_mfma_apply_persistent: block_rows=%d (64; 128-row blocks: gp_pool_mfma_apply)", block_rows);
    GP_CHECK_ARG(min_steps >= 9, "gp_pool_mfma_apply_persistent: a row block with %d < 9 steps (use gp_pool_mfma_apply)", min_steps);
    GP_CHECK_ARG(((y_hi && y_lo) != 0) != (y_f32 != nullptr), "gp_pool_mfma_apply_persistent: exactly one output form");
    GP_CHECK_ARG(ld_x % 8 == 0 && (uintptr_t)x_hi % 16 == 0 && (uintptr_t)x_lo % 16 == 0, "gp_pool_mfma_apply_persistent: x rows must be 16-byte aligned");
    GP_CHECK_ARG(!y_hi || (ld_y % 8 == 0 && (uintptr_t)y_hi % 16 == 0 && (uintptr_t)y_lo % 16 == 0 && y_hi != x_hi && y_lo != x_lo),
                 "gp_pool_mfma_apply_persistent: y rows must be 16-byte aligned and must not alias x");
    GP_CHECK_ARG(!y_f32 || (ld_yf % 4 == 0 && (uintptr_t)y_f32 % 16 == 0), "gp_pool_mfma_apply_persistent: fp32 output rows must be 16-byte aligned");
    int64_t nb = (nv + block_rows - 1) / block_rows;
    GP_CHECK_ARG(y_rows >= nb * block_rows, "gp_pool_mfma_apply_persistent: output needs %lld rows (whole row blocks), has %lld",
                 (long long)(nb * block_rows), (long long)y_rows);
    hipStream_t s = gp_stream(stream_);
#define PG_ARGS x_hi, x_lo, ld_x, bu_off, bu_row, wa_hi, wa_lo, nb, y_hi, y_lo, ld_y, y_f32, ld_yf, out_scale, queue, s
    return y_f32 ? pg_launch<1, 3, true>(PG_ARGS) : pg_launch<1, 3, false>(PG_ARGS);
#undef PG_ARGS
}

extern "C" size_t gp_pool_mfma_workspace_bytes(int64_t nv, int32_t block_rows) {
    if (nv <= 0 || block_rows <= 0) return 0;
    int64_t nb = (nv + block_rows - 1) / block_rows;
    GpCarver cv(nullptr, 0);
    cv.take<int64_t>(nb + 1);
    cv.take<char>(pm_scan_tmp(nb + 1));
    return cv.off;
}

// pass 1: bu_off i64 [nblocks+1] (padded union rows before each block; multiple of 32), bu_n i32 [nblocks]
extern "C" int gp_pool_mfma_count(const int32_t *nbr, int64_t nv, int32_t k, int32_t block_rows, int32_t min_steps, int64_t *bu_off,
                                  int32_t *bu_n, void *workspace, size_t workspace_bytes, void *stream_) {
    GP_CHECK_ARG(nbr && bu_off && bu_n && workspace && nv > 0 && k > 0, "gp_pool_mfma_count: null/empty argument");
    GP_CHECK_ARG(min_steps >= 0 && min_steps <= 64, "gp_pool_mfma_count: min_steps=%d (0..64; %d for the persistent kernel)", min_steps, PM_MIN_STEPS);
    GP_CHECK_ARG(block_rows == 64 || block_rows == 128, "gp_pool_mfma_count: block_rows=%d (64 or 128)", block_rows);
    GP_CHECK_ARG((int64_t)block_rows * k <= PM_MAXID, "gp_pool_mfma_count: k=%d too large (block_rows*k <= %d)", k, PM_MAXID);
    int64_t nb = (nv + block_rows - 1) / block_rows;
    GpCarver cv(workspace, workspace_bytes);
    int64_t *cnt = cv.take<int64_t>(nb + 1);
    size_t tb = pm_scan_tmp(nb + 1);
    char *tmp = cv.take<char>(tb);
    if (!cv.ok()) { gp_set_error("gp_pool_mfma_count: workspace too small"); return GP_ENOMEM; }
    hipStream_t s = gp_stream(stream_);
    GP_CHECK_HIP(hipMemsetAsync(cnt + nb, 0, sizeof(int64_t), s));
    size_t sm = (size_t)(PM_HS + pm_np2((int64_t)block_rows * k)) * sizeof(int);
    GP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pm_union_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (PM_HS + PM_MAXID) * (int)sizeof(int)));
    pm_union_kernel<<<(unsigned)nb, 1024, sm, s>>>(nbr, nv, k, block_rows, min_steps, cnt, bu_n, nullptr, nullptr);
    GP_CHECK_HIP(rocprim::exclusive_scan(tmp, tb, cnt, bu_off, (int64_t)0, (size_t)(nb + 1), rocprim::plus<int64_t>(), s));
    GP_CHECK_LAUNCH();
    return GP_OK;
}

// pass 2: bu_row i32 [total], wa_hi / wa_lo f16 [total/32 * (block_rows/16) * 64 * 8] (zeroed here, then scattered)
extern "C" int gp_pool_mfma_fill(const int32_t *nbr, const float *w, int64_t nv, int32_t k, int32_t block_rows,
                                 const int64_t *bu_off, const int32_t *bu_n, int64_t total_rows, int32_t *bu_row, void *wa_hi,
                                 void *wa_lo, void *stream_) {
    GP_CHECK_ARG(nbr && w && bu_off && bu_n && bu_row && wa_hi && wa_lo && nv > 0 && total_rows > 0 && total_rows % PM_KS == 0,
                 "gp_pool_mfma_fill: bad argument");
    GP_CHECK_ARG(block_rows == 64 || block_rows == 128, "gp_pool_mfma_fill: block_rows=%d (64 or 128)", block_rows);
    GP_CHECK_ARG((int64_t)block_rows * k <= PM_MAXID, "gp_pool_mfma_fill: k=%d too large", k);
    int64_t nb = (nv + block_rows - 1) / block_rows;
    hipStream_t s = gp_stream(stream_);
    size_t wbytes = (size_t)(total_rows / PM_KS) * (block_rows / 16) * 64 * 8 * sizeof(_Float16);
    GP_CHECK_HIP(hipMemsetAsync(wa_hi, 0, wbytes, s));
    GP_CHECK_HIP(hipMemsetAsync(wa_lo, 0, wbytes, s));
    size_t sm = (size_t)(PM_HS + pm_np2((int64_t)block_rows * k)) * sizeof(int);
    GP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pm_union_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (PM_HS + PM_MAXID) * (int)sizeof(int)));
    pm_union_kernel<<<(unsigned)nb, 1024, sm, s>>>(nbr, nv, k, block_rows, 0, nullptr, nullptr, bu_off, bu_row);
    pm_weights_kernel<<<(unsigned)nb, 256, 0, s>>>(nbr, w, nv, k, block_rows, bu_off, bu_n, bu_row, static_cast<_Float16 *>(wa_hi),
                                                   static_cast<_Float16 *>(wa_lo));
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" int gp_pool_mfma_apply(const void *x_hi, const void *x_lo, int64_t ld_x, const int64_t *bu_off,
                                  const int32_t *bu_row, const void *wa_hi, const void *wa_lo, int64_t nv, int32_t d,
                                  int32_t block_rows, void *y_hi, void *y_lo, int64_t ld_y, float *y_f32, int64_t ld_yf,
                                  const float *out_scale, void *stream_) {
    GP_CHECK_ARG(x_hi && x_lo && bu_off && bu_row && wa_hi && wa_lo && nv > 0, "gp_pool_mfma_apply: null/empty argument");
    GP_CHECK_ARG(d == PM_D, "gp_pool_mfma_apply: d=%d (kernel specialised for %d columns)", d, PM_D);
    GP_CHECK_ARG(block_rows == 64 || block_rows == 128, "gp_pool_mfma_apply: block_rows=%d (64 or 128)", block_rows);
    GP_CHECK_ARG((y_hi && y_lo) || y_f32, "gp_pool_mfma_apply: no output requested");
    GP_CHECK_ARG(ld_x % 8 == 0 && (uintptr_t)x_hi % 16 == 0 && (uintptr_t)x_lo % 16 == 0, "gp_pool_mfma_apply: x rows must be 16-byte aligned");
    GP_CHECK_ARG(!y_hi || (ld_y % 4 == 0 && y_hi != x_hi && y_lo != x_lo), "gp_pool_mfma_apply: y must not alias x");
    GP_CHECK_ARG(!y_f32 || ld_yf % 4 == 0, "gp_pool_mfma_apply: fp32 output rows must be 16-byte aligned");
    hipStream_t s = gp_stream(stream_);
    if (block_rows == 64 && g_gp_knobs[11] == 4)           // tuning aid: every wave owns all 64 rows x 32 columns (X read once from LDS)
        return pm_launch<4, 128, 4, 4>(x_hi, x_lo, ld_x, bu_off, bu_row, wa_hi, wa_lo, nv, y_hi, y_lo, ld_y, y_f32, ld_yf, out_scale, s);
    if (block_rows == 64)
        return pm_launch<4, 128, 1, 1>(x_hi, x_lo, ld_x, bu_off, bu_row, wa_hi, wa_lo, nv, y_hi, y_lo, ld_y, y_f32, ld_yf, out_scale, s);
    return pm_launch<8, 256, 2, 2>(x_hi, x_lo, ld_x, bu_off, bu_row, wa_hi, wa_lo, nv, y_hi, y_lo, ld_y, y_f32, ld_yf, out_scale, s);
}
