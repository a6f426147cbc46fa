// Row 12, column-sliced matrix-core pooling ("cs"; the default pooling kernel from round 3 on).
//
// What the counters and in-kernel time stamps of pool_mfma.hip said (profiles/r03_pool_stamps_*.log): in the steady
// state of its loop a CU already takes in 65 GB/s of gathered rows and weight fragments -- the measured ceiling of the
// L2 -> LDS gather path (MI355X_MICROARCH.md, "Indexed rows: gather into LDS") -- and waits only 190 of 1 560 cycles
// per step at the hand-over; a quarter of every workgroup's life is prologue and epilogue.  The kernel is bound by the
// BYTES each CU pulls through L2, so this kernel pulls fewer:
//   * a workgroup owns 128 Morton-adjacent rows x 256 columns (pool_mfma.hip: 64 x 128): the union of a 128-row block
//     has 4.75 rows per output row instead of 6.66, and the weight fragments are read by 2 column halves, not 4
//     quarters:  X 9.7 KB + weights 2.4 KB per output row and application instead of 13.6 + 6.8 KB;
//   * every wave owns ALL 128 rows x 32 columns (column-sliced), so the eight 16-row groups see the same staged union
//     rows and a (group, step) weight fragment that is entirely zero can be skipped by the whole workgroup with no
//     imbalance: the builder orders a block's union rows by (first group, last group) that use them, which leaves 60 %
//     of the 16 x 32 fragments non-empty (sorted by id: 79 %), and stores one bit per (step, group);
//     empty fragments are neither fetched (their LDS-DMA reads one hot line) nor read from LDS nor multiplied;
//   * row ids and fragment masks come through the scalar cache (s_load one step ahead) instead of an LDS-DMA + LDS
//     read-back per step.
// Numerics are those of pool_mfma.hip: pre-split f16 (hi, lo) operands, hi*hi + hi*lo + lo*hi on
// v_mfma_f32_16x16x32_f16 with fp32 accumulation, the union swept in the order the builder fixed (bitwise reproducible).
#include <cstring>
#include <type_traits>
#include <rocprim/device/device_scan.hpp>

#include "gp_common.h"

extern int g_gp_knobs[16];
extern void *g_gp_debug_ptr[4];

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((vector_size(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int CS_KS = 32;              // union rows per step (MFMA K)
constexpr int CS_D = 512;              // feature columns
constexpr int CS_BR = 128;             // rows per block
constexpr int CS_NG = CS_BR / 16;      // 16-row groups per block = weight fragments per step
constexpr int CS_NC = 256;             // columns per workgroup
constexpr int CS_NW = 8;               // waves per workgroup
constexpr int CS_WC = CS_NC / CS_NW;   // columns per wave
constexpr int CS_MAXID = 16384;        // sort buffers of the builder (a power of two)
constexpr int CS_MAXNK = 12288;        // block_rows x K ids per block (K <= 96): with the position table the builder's LDS is full
constexpr int CS_HS = 16384;           // hash slots of the builder
constexpr float CS_WSCALE = 1024.f;    // weights (<= 1) are stored x 2^10 so that their f16 lo parts stay normal

// LDS stage: X hi [32 rows][512 B] | X lo | weights hi [8 groups][1 KiB] | weights lo
constexpr int CS_RB = CS_NC * 2;                   // bytes per staged row and plane
constexpr int CS_PLANE = CS_KS * CS_RB;            // 16 KiB
constexpr int CS_OFF_W = 2 * CS_PLANE;             // 32 KiB
constexpr int CS_WPL = CS_NG * 1024;               // one weight plane: 8 KiB
constexpr int CS_STAGE = CS_OFF_W + 2 * CS_WPL;    // 48 KiB
constexpr int CS_NST = 3;
constexpr int CS_DMA = 6;                          // LDS-DMA instructions per wave and stage: 4 x rows, 2 x weights
constexpr int CS_EP = CS_WC + 4;                   // epilogue staging pitch (floats)
constexpr size_t CS_SMEM = (size_t)CS_NST * CS_STAGE;
static_assert((size_t)CS_NW * CS_BR * CS_EP * sizeof(float) <= CS_SMEM, "epilogue staging must fit in the ring");

__device__ __forceinline__ uint64_t cs_now() {
    uint64_t t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ uint64_t cs_real() {
    uint64_t t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ void cs_glds16(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}
template <int OFF>
__device__ __forceinline__ void cs_tr(s16x4 &d, uint32_t addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void cs_rd128(f16x8 &d, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
// a weight fragment's {hi, lo} reads under a wave-uniform condition (bit BIT of the scalar m), as ONE in-place update of the
// two destinations: the branch sits inside the statement, so the compiler sees plain read-modify-write data flow
template <int BIT, int OFF_HI, int OFF_LO>
__device__ __forceinline__ void cs_rd128_if(f16x8 &h, f16x8 &l, uint32_t addr, unsigned m) {
    asm volatile("s_bitcmp1_b32 %3, %4\n\t"
                 "s_cbranch_scc0 1f\n\t"
                 "ds_read_b128 %0, %2 offset:%5\n\t"
                 "ds_read_b128 %1, %2 offset:%6\n"
                 "1:"
                 : "+v"(h), "+v"(l)
                 : "v"(addr), "s"(m), "n"(BIT), "n"(OFF_HI), "n"(OFF_LO)
                 : "scc");
}
// every LDS read issued so far has landed; ties the fragment registers to the wait so that no use moves above it
__device__ __forceinline__ void cs_wait_b(s16x4 (&f)[2][2][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f[0][0][0]), "+v"(f[0][0][1]), "+v"(f[0][1][0]), "+v"(f[0][1][1]), "+v"(f[1][0][0]), "+v"(f[1][0][1]),
                   "+v"(f[1][1][0]), "+v"(f[1][1][1]));
}
__device__ __forceinline__ void cs_wait_a(f16x8 (&h)[4], f16x8 (&l)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3]));
}
template <int N>
__device__ __forceinline__ void cs_handover() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}
__device__ __forceinline__ f16x8 cs_cat(s16x4 a, s16x4 b) {
    typedef short s16x8 __attribute__((vector_size(16)));
    s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(f16x8, v);
}

// ------------------------------------------------------------------------------------------------ builder
__device__ __forceinline__ void cs_bitonic(int *a, int n_pow2, int tid, int nthreads) {
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

// distinct neighbour ids of the rows of block b -> dense[0 .. U) (unsorted), through an LDS hash table: a small table
// first (unions of lattice neighbourhoods are a few hundred ids), the full-size one if it overflows
__device__ int cs_union(const int32_t *__restrict__ nbr, int n, int *keys, int *dense) {
    __shared__ int s_wcnt[16];
    __shared__ int s_base, s_new, s_over;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int hs = 2048, shift = 21;
    for (;;) {
        for (int i = tid; i < hs; i += 1024) keys[i] = -1;
        if (tid == 0) { s_base = 0; s_new = 0; s_over = 0; }
        __syncthreads();
        for (int i = tid; i < n; i += 1024) {
            const int id = nbr[i];                                       // (nbr points at the block's first entry)
            unsigned h = ((unsigned)id * 2654435761u) >> shift;
            int probes = 0;
            while (true) {
                const int old = atomicCAS(&keys[h], -1, id);
                if (old == -1) { atomicAdd(&s_new, 1); break; }
                if (old == id) break;
                h = (h + 1) & (hs - 1);
                if (++probes > 256 && hs < CS_HS) { s_over = 1; break; }   // (the full table always has a free slot)
            }
        }
        __syncthreads();
        const bool redo = hs < CS_HS && (s_over || s_new > 1024);          // block-uniform
        __syncthreads();
        if (!redo) break;
        hs = CS_HS;
        shift = 18;
    }
    for (int i0 = 0; i0 < hs; i0 += 1024) {                                // compact the occupied slots
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
    return s_base;
}

// pass 1: padded union size of every block (a multiple of 32 union rows, at least one step)
__global__ void __launch_bounds__(1024)
cs_count_kernel(const int32_t *__restrict__ nbr, int64_t nv, int k, int64_t *__restrict__ padded_cnt, int32_t *__restrict__ bu_n) {
    extern __shared__ int s_mem[];
    int *keys = s_mem, *dense = s_mem + CS_HS;
    const int64_t b = blockIdx.x, r0 = b * CS_BR;
    const int rows = (int)((nv - r0) < CS_BR ? (nv - r0) : CS_BR);
    const int U = cs_union(nbr + r0 * k, rows * k, keys, dense);
    if (threadIdx.x == 0) { padded_cnt[b] = (int64_t)((U + CS_KS - 1) / CS_KS) * CS_KS; bu_n[b] = U; }
}

// pass 2: the block's union rows in (first group, last group, group set, id) order, one bit per (step, group) that says
// whether the 16 x 32 weight fragment holds a non-zero, and the ELL weights scattered into MFMA fragment order:
// wa[(step * 8 + group) * 64 + lane][8], lane = (k >> 3) * 16 + m  (k = union row within the step, m = row within the group).
__global__ void __launch_bounds__(1024)
cs_fill_kernel(const int32_t *__restrict__ nbr, const float *__restrict__ w, int64_t nv, int k, const int64_t *__restrict__ bu_off,
               int32_t *__restrict__ bu_row, uint32_t *__restrict__ bu_mask, _Float16 *__restrict__ wa_hi, _Float16 *__restrict__ wa_lo) {
    extern __shared__ int s_mem[];                           // A[16384] | B[16384] | npos u16 [br*k]
    int *A = s_mem, *B = s_mem + CS_HS;
    unsigned short *npos = reinterpret_cast<unsigned short *>(s_mem + CS_HS + CS_MAXID);
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x, r0 = b * CS_BR;
    const int rows = (int)((nv - r0) < CS_BR ? (nv - r0) : CS_BR);
    const int n = rows * k;
    const int32_t *nb = nbr + r0 * k;
    const int U = cs_union(nb, n, A, B);
    int np2 = 1;
    while (np2 < U) np2 <<= 1;
    for (int i = U + tid; i < np2; i += 1024) B[i] = INT32_MAX;
    __syncthreads();
    cs_bitonic(B, np2, tid, 1024);                            // B[0 .. U): the ids, ascending
    // group set of every union row
    for (int i = tid; i < U; i += 1024) A[i] = 0;
    __syncthreads();
    for (int t = tid; t < n; t += 1024) {
        const int id = nb[t];
        int lo = 0, hi = U - 1;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (B[mid] < id) lo = mid + 1; else hi = mid; }
        atomicOr(&A[lo], 1 << ((t / k) >> 4));
    }
    __syncthreads();
    // order key: first group | last group | group set | index among the sorted ids (deterministic)
    for (int i = tid; i < np2; i += 1024) {
        int key = INT32_MAX;
        if (i < U) {
            const int m = A[i];
            const int first = __ffs(m) - 1, last = 31 - __clz(m);
            key = (first << 28) | (last << 25) | (m << 14) | i;      // 3 + 3 + 8 + 14 bits (bit 31 stays clear)
        }
        A[i] = key;
    }
    __syncthreads();
    cs_bitonic(A, np2, tid, 1024);
    const int64_t o = bu_off[b];
    const int Up = (int)(bu_off[b + 1] - o);
    for (int p = tid; p < Up; p += 1024) {
        const int src = p < U ? (A[p] & 0x3FFF) : (A[0] & 0x3FFF);   // padding repeats the first row (its weights stay zero)
        bu_row[o + p] = B[src];
        if (p < U) npos[src] = (unsigned short)p;
    }
    const int64_t ks0 = o / CS_KS;
    __shared__ unsigned s_mask[CS_MAXNK / CS_KS];
    for (int t = tid; t < Up / CS_KS; t += 1024) {
        unsigned m = 0;
        for (int p = t * CS_KS; p < (t + 1) * CS_KS && p < U; ++p) m |= (unsigned)(A[p] >> 14) & 0xFFu;
        bu_mask[ks0 + t] = m;
        s_mask[t] = m;
    }
    __syncthreads();
    // zero the block's non-empty fragments (the only ones the apply kernel fetches), then scatter into them: the
    // barrier orders this workgroup's zero stores before its element stores (both through the same L2)
    for (int i = tid; i < (Up / CS_KS) * CS_NG * 64; i += 1024) {
        const int f = i >> 6;
        if ((s_mask[f >> 3] >> (f & 7)) & 1u) {
            const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            *reinterpret_cast<f16x8 *>(wa_hi + (ks0 * CS_NG * 64 + i) * 8) = z;
            *reinterpret_cast<f16x8 *>(wa_lo + (ks0 * CS_NG * 64 + i) * 8) = z;
        }
    }
    __syncthreads();
    for (int t = tid; t < n; t += 1024) {
        const int rl = t / k;
        const int id = nb[t];
        int lo = 0, hi = U - 1;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (B[mid] < id) lo = mid + 1; else hi = mid; }
        const int p = npos[lo];
        const int64_t ks = ks0 + p / CS_KS;
        const int kk = p % CS_KS;
        const int64_t idx = ((ks * CS_NG + (rl >> 4)) * 64 + (kk >> 3) * 16 + (rl & 15)) * 8 + (kk & 7);
        const float v = w[r0 * k + t] * CS_WSCALE;
        const _Float16 h = (_Float16)v;
        wa_hi[idx] = h;
        wa_lo[idx] = (_Float16)(v - (float)h);
    }
}

// ------------------------------------------------------------------------------------------------ apply
// One 512-thread workgroup = 128 rows x 256 columns (grid = row blocks x 2 column halves, the halves of a row block
// adjacent on one XCD); wave wv owns columns 32 wv .. 32 wv + 31 of the half for all 128 rows (16 accumulator tiles).
// A 3-deep ring of 48-KiB stages is filled by LDS-DMA two steps ahead (96 KiB in flight per CU); wave wv stages union
// rows 4 wv .. 4 wv + 3 of a step (two 1-KiB instructions per plane, two rows each) and the weight fragment of group wv.
// The image is XOR-swizzled through the DMA source addresses exactly as in pool_mfma.hip (physical 16-byte chunk c of
// row r holds logical chunk c ^ 2 t(r), t(r) = (r & 3) | ((r >> 3) & 1) << 2), which makes the transposed fragment
// reads (ds_read_b64_tr_b16) conflict-free.  Synchronisation is hand-counted: LDS reads are inline asm with their own
// lgkmcnt waits (a compiler-visible LDS read would wait for every outstanding LDS-DMA), the hand-over is
// `s_waitcnt vmcnt(6); s_barrier` (6 = the DMA instructions of the younger stage; vector memory operations complete in
// issue order and the loop issues no other).  Row ids and fragment masks are scalar loads issued one step ahead.
// Tile geometry of the kernel: NC columns per workgroup (256: the halves above; 128: quarters -- the XCD's 32 workgroups then
// cover 8 row blocks instead of 16, a 1 024-row window whose union rows fit the 4-MiB L2), NST ring slots.
template <int NC, int NST>
struct CsGeo {
    static constexpr int WC = NC / CS_NW;              // columns per wave
    static constexpr int NU = WC / 16;                 // column tiles per wave
    static constexpr int RB = NC * 2;                  // bytes per staged row and plane
    static constexpr int PLANE = CS_KS * RB;
    static constexpr int OFF_W = 2 * PLANE;
    static constexpr int STAGE = OFF_W + 2 * CS_WPL;
    static constexpr int DMA = (2 * PLANE + 2 * CS_WPL) / 1024 / CS_NW;   // LDS-DMA instructions per wave and stage
    static constexpr int EP = WC + 4;                  // epilogue staging pitch (floats)
    static constexpr int D = NST - 1;                  // stages issued ahead
    static constexpr size_t SMEM = (size_t)NST * STAGE;
    static constexpr int WPS = SMEM <= 80 * 1024 ? 4 : 2;   // waves per SIMD the register budget is set for (two workgroups per CU if the ring allows)
    static_assert((size_t)CS_NW * (CS_BR / 2) * EP * sizeof(float) <= SMEM, "epilogue staging (64 rows at a time) must fit in the ring");
    static_assert(SMEM <= 160 * 1024, "ring");
    static_assert(NC == 256 || NC == 128, "column tile");
};

template <int NC, int NST, bool STAMP>
__global__ void __launch_bounds__(512, (CsGeo<NC, NST>::WPS))
cs_pool_kernel(const _Float16 *__restrict__ x_hi, const _Float16 *__restrict__ x_lo, int64_t ld_x,
               const int64_t *__restrict__ bu_off, const int32_t *__restrict__ bu_row, const uint32_t *__restrict__ bu_mask,
               const _Float16 *__restrict__ wa_hi, const _Float16 *__restrict__ wa_lo, int64_t nv, int64_t nblocks,
               _Float16 *__restrict__ y_hi, _Float16 *__restrict__ y_lo, int64_t ld_y, float *__restrict__ y_f32, int64_t ld_yf,
               int64_t per_xcd, int ablate, const float *__restrict__ out_scale, uint64_t *__restrict__ stamp) {
    using G = CsGeo<NC, NST>;
    constexpr int NU = G::NU, D = G::D, NSPLIT = CS_D / NC;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    uint64_t st_t0 = 0, st_r0 = 0, st_pro = 0, st_work = 0, st_wait = 0, st_issue = 0;
    if constexpr (STAMP) { st_t0 = cs_now(); st_r0 = cs_real(); }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t lb = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);    // XCD-contiguous order
    const int64_t b = lb / NSPLIT;
    const int col0 = (int)(lb % NSPLIT) * NC;
    if (b >= nblocks) return;
    const int64_t ub0 = bu_off[b];
    const int n = (int)((bu_off[b + 1] - ub0) / CS_KS);                            // steps (>= 1)
    const int64_t ks0 = ub0 / CS_KS;

    // ---- DMA roles: wave wv stages union rows 4 wv .. 4 wv + 3 of a step and the weight fragment of group wv
    //      NC = 256: a 1-KiB instruction carries two rows (lanes 0-31 | 32-63), two instructions per plane
    //      NC = 128: a 1-KiB instruction carries the four rows (16 lanes each), one instruction per plane
    const int du = NC == 256 ? lane >> 5 : lane >> 4, dc = NC == 256 ? lane & 31 : lane & 15;
    const int swz = (wv >> 1) & 1;
    const int t0 = du | (swz << 2), t1 = (2 + du) | (swz << 2);                     // t(row) of rows 4 wv + du (, 4 wv + 2 + du)
    const int64_t dsrc0 = col0 + ((dc ^ (2 * t0)) * 8);
    const int64_t dsrc1 = col0 + ((dc ^ (2 * t1)) * 8);
    const int32_t *idg = bu_row + ub0 + 4 * wv;                                    // this wave's row ids, step 0
    const uint32_t *mkg = bu_mask + ks0;
    const _Float16 *wah = wa_hi + (ks0 * CS_NG + wv) * 512;
    const _Float16 *wal = wa_lo + (ks0 * CS_NG + wv) * 512;
    auto issue = [&](i32x4 id, unsigned mk, int k, int slot) {
        unsigned char *dst = smem_raw + slot * G::STAGE;
        if (!(ablate & 2)) {                               // tuning aid: bit 1 skips the row gather
            if constexpr (NC == 256) {
                const int ida = du ? id.y : id.x, idb = du ? id.w : id.z;
                const int64_t s0 = (int64_t)ida * ld_x + dsrc0, s1 = (int64_t)idb * ld_x + dsrc1;
                cs_glds16(x_hi + s0, dst + (4 * wv) * G::RB);
                cs_glds16(x_lo + s0, dst + G::PLANE + (4 * wv) * G::RB);
                cs_glds16(x_hi + s1, dst + (4 * wv) * G::RB + 1024);
                cs_glds16(x_lo + s1, dst + G::PLANE + (4 * wv) * G::RB + 1024);
            } else {
                const int idr = du == 0 ? id.x : du == 1 ? id.y : du == 2 ? id.z : id.w;
                const int64_t s0 = (int64_t)idr * ld_x + dsrc0;
                cs_glds16(x_hi + s0, dst + (4 * wv) * G::RB);
                cs_glds16(x_lo + s0, dst + G::PLANE + (4 * wv) * G::RB);
            }
        }
        if (!(ablate & 8)) {                               // tuning aid: bit 3 skips the weight fragments
            // an empty fragment is never read: all lanes fetch its first 16 bytes (one hot line) to keep the DMA count fixed
            const int lo = ((mk >> wv) & 1u) ? lane * 8 : 0;
            cs_glds16(wah + (int64_t)k * (CS_NG * 512) + lo, dst + G::OFF_W + wv * 1024);
            cs_glds16(wal + (int64_t)k * (CS_NG * 512) + lo, dst + G::OFF_W + CS_WPL + wv * 1024);
        }
    };
    auto load_ids = [&](int k) { return *reinterpret_cast<const i32x4 *>(idg + (int64_t)k * CS_KS); };

    // ---- read roles
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem_raw;
    uint32_t addr[NU];
    {
        const uint32_t rowb = (uint32_t)(8 * g + q) * G::RB + (uint32_t)(wv * G::WC * 2) + (uint32_t)(p * 8);
        const uint32_t t = (uint32_t)(q | ((g & 1) << 2));
#pragma unroll
        for (int u = 0; u < NU; ++u) addr[u] = lds0 + ((rowb + 32u * u) ^ (t << 5));
    }
    const uint32_t addr_w = lds0 + G::OFF_W + lane * 16;

    f32x4 acc[CS_NG * NU];
#pragma unroll
    for (int i = 0; i < CS_NG * NU; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: stages 0 .. D - 1 in flight (a short block stages its last step again); mq[j] = mask of stage s + j
    unsigned mq[NST];
    i32x4 idv;
    {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int kj = j < n ? j : n - 1;
            const i32x4 ij = load_ids(kj);
            mq[j] = mkg[kj];
            issue(ij, mq[j], kj, j);
        }
        const int kd = D < n ? D : n - 1;
        idv = load_ids(kd);
        mq[D] = mkg[kd];
        asm volatile("" ::"s"(idv.x), "s"(idv.y), "s"(idv.z), "s"(idv.w), "s"(mq[D]));   // (waited for here, not inside the loop)
        cs_handover<(D - 1) * G::DMA>();
    }
    if constexpr (STAMP) st_pro = cs_now();
    // Software pipeline (the fragment reads of all eight waves leave the barrier together and take ~500 cycles to come back;
    // an MFMA batch in front of each wait hides part of that):
    //   step s:  reads {staged rows, weight fragments of groups 0-3} of stage s      | waves 0-3: DMA of stage s + D
    //            MFMA batch "groups 4-7" of stage s - 1 (fragments read in step s - 1, rows kept in bhp / blp)
    //            wait; reads {weight fragments of groups 4-7} of stage s; next step's scalars (s_load)
    //            MFMA batch "groups 0-3" of stage s                                  | waves 4-7: DMA of stage s + D
    //            wait (every LDS read of stage s has landed in registers); hand-over
    // Waves 0-3 issue their DMA first and waves 4-7 last, so that the two waves of a SIMD alternate between DMA issue
    // (which stalls on the memory pipeline's back-pressure) and matrix work.
    // Measured and left out (profiles/r03_pool_cs_variants.log): releasing a slot as soon as its operands are in registers
    // (a second barrier per step, the stage THREE steps ahead issued into it: 0.259 instead of 0.232 ms -- the launch is bound by
    // the bytes that reach HBM, 1.28 GB at 5.5 TB/s, not by the bytes in flight).
    s16x4 fb[NU][2][2];
    f16x8 ah0[4], al0[4], ah1[4], al1[4], bhp[NU], blp[NU];
#pragma unroll
    for (int i = 0; i < 4; ++i) { ah0[i] = al0[i] = ah1[i] = al1[i] = f16x8{0, 0, 0, 0, 0, 0, 0, 0}; }
#pragma unroll
    for (int u = 0; u < NU; ++u) { bhp[u] = blp[u] = f16x8{0, 0, 0, 0, 0, 0, 0, 0}; }
    unsigned mP = 0;                                         // fragment mask of the previous step (its groups 4-7 are pending)
    auto mfma_hi = [&](unsigned m) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
            if (__builtin_expect((m >> (4 + mt)) & 1u, 1)) {
#pragma unroll
                for (int u = 0; u < NU; ++u) acc[(4 + mt) * NU + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1[mt], bhp[u], acc[(4 + mt) * NU + u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < NU; ++u) acc[(4 + mt) * NU + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1[mt], blp[u], acc[(4 + mt) * NU + u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < NU; ++u) acc[(4 + mt) * NU + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1[mt], bhp[u], acc[(4 + mt) * NU + u], 0, 0, 0);
            }
    };
    const bool late = !(ablate & 32) && wv >= 4;            // tuning aid: bit 5 makes every wave issue first
    const bool do_reads = !(ablate & 1);                     // tuning aid: bit 0 skips reads + MFMAs
    for (int s0 = 0; s0 < n; s0 += NST) {
#pragma unroll
        for (int J = 0; J < NST; ++J) {
            const int s = s0 + J;
            if (s < n) {
                uint64_t st_a = 0, st_b = 0;
                if constexpr (STAMP) st_a = cs_now();
                const uint32_t aw = addr_w + J * G::STAGE;
                const unsigned m = mq[0];
                if (do_reads) {
                    // staged rows: fb[col block][plane][rows 8g+q | 8g+q+4]; weight fragments of groups 0-3
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        const uint32_t au = addr[u] + J * G::STAGE;
                        cs_tr<0>(fb[u][0][0], au);
                        cs_tr<4 * G::RB>(fb[u][0][1], au);
                        cs_tr<G::PLANE>(fb[u][1][0], au);
                        cs_tr<G::PLANE + 4 * G::RB>(fb[u][1][1], au);
                    }
                    if (m & 1u) { cs_rd128<0 * 1024>(ah0[0], aw); cs_rd128<CS_WPL + 0 * 1024>(al0[0], aw); }
                    if (m & 2u) { cs_rd128<1 * 1024>(ah0[1], aw); cs_rd128<CS_WPL + 1 * 1024>(al0[1], aw); }
                    if (m & 4u) { cs_rd128<2 * 1024>(ah0[2], aw); cs_rd128<CS_WPL + 2 * 1024>(al0[2], aw); }
                    if (m & 8u) { cs_rd128<3 * 1024>(ah0[3], aw); cs_rd128<CS_WPL + 3 * 1024>(al0[3], aw); }
                }
                if (!late && s + D < n) issue(idv, mq[D], s + D, (J + D) % NST);
                if constexpr (STAMP) if (ablate & 64) st_issue += cs_now() - st_a;
                if (do_reads) {
                    mfma_hi(mP);                            // groups 4-7 of the previous step
                    if constexpr (NU == 2)
                        asm volatile("s_waitcnt lgkmcnt(0)"
                                     : "+v"(fb[0][0][0]), "+v"(fb[0][0][1]), "+v"(fb[0][1][0]), "+v"(fb[0][1][1]), "+v"(fb[NU - 1][0][0]),
                                       "+v"(fb[NU - 1][0][1]), "+v"(fb[NU - 1][1][0]), "+v"(fb[NU - 1][1][1]));
                    else
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[0][0][0]), "+v"(fb[0][0][1]), "+v"(fb[0][1][0]), "+v"(fb[0][1][1]));
                    cs_wait_a(ah0, al0);
                }
                // scalars of the stage issued in the next step (clamped: never past the block's padded union); they are
                // waited for right before the barrier, a whole MFMA batch later
                const int kn = s + D + 1 < n ? s + D + 1 : n - 1;
                const i32x4 idn = load_ids(kn);
                const unsigned mN = mkg[kn];
                if (do_reads) {
                    if (m & 16u) { cs_rd128<4 * 1024>(ah1[0], aw); cs_rd128<CS_WPL + 4 * 1024>(al1[0], aw); }
                    if (m & 32u) { cs_rd128<5 * 1024>(ah1[1], aw); cs_rd128<CS_WPL + 5 * 1024>(al1[1], aw); }
                    if (m & 64u) { cs_rd128<6 * 1024>(ah1[2], aw); cs_rd128<CS_WPL + 6 * 1024>(al1[2], aw); }
                    if (m & 128u) { cs_rd128<7 * 1024>(ah1[3], aw); cs_rd128<CS_WPL + 7 * 1024>(al1[3], aw); }
#pragma unroll
                    for (int u = 0; u < NU; ++u) { bhp[u] = cs_cat(fb[u][0][0], fb[u][0][1]); blp[u] = cs_cat(fb[u][1][0], fb[u][1][1]); }
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        if (__builtin_expect((m >> mt) & 1u, 1)) {
#pragma unroll
                            for (int u = 0; u < NU; ++u) acc[mt * NU + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0[mt], bhp[u], acc[mt * NU + u], 0, 0, 0);
#pragma unroll
                            for (int u = 0; u < NU; ++u) acc[mt * NU + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0[mt], blp[u], acc[mt * NU + u], 0, 0, 0);
#pragma unroll
                            for (int u = 0; u < NU; ++u) acc[mt * NU + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0[mt], bhp[u], acc[mt * NU + u], 0, 0, 0);
                        }
                }
                if (late && s + D < n) issue(idv, mq[D], s + D, (J + D) % NST);
                // every LDS read of this stage is in registers before the barrier lets its slot be refilled, and the scalar
                // loads are waited for HERE, so that no compiler-placed lgkmcnt(0) sits inside the next step
                cs_wait_a(ah1, al1);
                asm volatile("" ::"s"(idn.x), "s"(idn.y), "s"(idn.z), "s"(idn.w), "s"(mN));
                if constexpr (STAMP) { st_b = cs_now(); st_work += st_b - st_a; }
                // stage s + 1 has landed: everything younger stays in flight (min(D - 1, n - 2 - s) stages)
                if (s + D < n) cs_handover<(D - 1) * G::DMA>();
                else if (D >= 3 && s + D - 1 < n) cs_handover<(D >= 3 ? D - 2 : 0) * G::DMA>();
                else if (D >= 4 && s + D - 2 < n) cs_handover<(D >= 4 ? D - 3 : 0) * G::DMA>();
                else cs_handover<0>();
                if constexpr (STAMP) st_wait += cs_now() - st_b;
                mP = m;
#pragma unroll
                for (int j = 0; j < D; ++j) mq[j] = mq[j + 1];
                mq[D] = mN;
                idv = idn;
            }
        }
    }
    if (do_reads) mfma_hi(mP);                               // groups 4-7 of the last step
    if (ablate & 4) return;                                // tuning aid: bit 2 skips the epilogue
    uint64_t st_e0 = 0;
    if constexpr (STAMP) st_e0 = cs_now();
    // ---- epilogue through LDS (the ring is drained: the last hand-over waited for vmcnt(0)).  The split planes carry
    // x * s (s = the power of two of gp_pow2_scale); pooling is linear, so the planes written for the next application stay
    // in that domain and only the fp32 output is multiplied by out_scale = 1/s.
    const float inv = 1.f / CS_WSCALE;
    float *stg = reinterpret_cast<float *>(smem_raw) + wv * ((CS_BR / 2) * G::EP);   // 64 rows at a time (bounds staging and live registers)
    const int fl = lane & 15, fq = lane >> 4;
    const int64_t row0 = b * CS_BR;
    const int colw = col0 + wv * G::WC;
    // lane -> 8 consecutive columns of a row: every store instruction writes RPI rows x (WC * 2) bytes
    constexpr int LPR = G::WC / 8, RPI = 64 / LPR;          // lanes per row, rows per instruction
    const int er = lane / LPR, ec = (lane % LPR) * 8;
    const float so = (y_f32 && out_scale) ? out_scale[0] : 1.f;
#pragma unroll
    for (int h8 = 0; h8 < 2; ++h8) {
        if (h8) gp_wave_sync();
#pragma unroll
        for (int mt = 0; mt < CS_NG / 2; ++mt)
#pragma unroll
            for (int cb = 0; cb < NU; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) stg[(mt * 16 + fq * 4 + r) * G::EP + cb * 16 + fl] = acc[(h8 * (CS_NG / 2) + mt) * NU + cb][r] * inv;
        gp_wave_sync();
        float4 v[64 / RPI][2];
#pragma unroll
        for (int it = 0; it < 64 / RPI; ++it) {
            const float *sp = stg + (it * RPI + er) * G::EP + ec;
            v[it][0] = *reinterpret_cast<const float4 *>(sp);
            v[it][1] = *reinterpret_cast<const float4 *>(sp + 4);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int it = 0; it < 64 / RPI; ++it) {
            const int64_t grow = row0 + h8 * 64 + it * RPI + er;
            if (grow < nv && !(ablate & 16)) {             // tuning aid: bit 4 skips the output stores
                const float xv[8] = {v[it][0].x, v[it][0].y, v[it][0].z, v[it][0].w, v[it][1].x, v[it][1].y, v[it][1].z, v[it][1].w};
                if (y_hi) {
                    f16x8 h, l;
#pragma unroll
                    for (int i = 0; i < 8; ++i) { h[i] = (_Float16)xv[i]; l[i] = (_Float16)(xv[i] - (float)h[i]); }
                    *reinterpret_cast<f16x8 *>(y_hi + grow * ld_y + colw + ec) = h;
                    *reinterpret_cast<f16x8 *>(y_lo + grow * ld_y + colw + ec) = l;
                }
                if (y_f32) {
                    float *yp = y_f32 + grow * ld_yf + colw + ec;
                    *reinterpret_cast<float4 *>(yp) = make_float4(xv[0] * so, xv[1] * so, xv[2] * so, xv[3] * so);
                    *reinterpret_cast<float4 *>(yp + 4) = make_float4(xv[4] * so, xv[5] * so, xv[6] * so, xv[7] * so);
                }
            }
        }
    }
    if constexpr (STAMP) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint64_t t3 = cs_now(), r3 = cs_real();
        if (lane == 0 && stamp) {
            uint64_t *o = stamp + ((int64_t)blockIdx.x * CS_NW + wv) * 10;
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            o[0] = st_r0; o[1] = r3 - st_r0; o[2] = st_pro - st_t0; o[3] = st_work; o[4] = st_wait; o[5] = st_issue;
            o[6] = t3 - st_e0; o[7] = t3 - st_t0; o[8] = (uint64_t)n; o[9] = xcc;
        }
    }
}

// ------------------------------------------------------------------------------------------------ persistent form
// The same loop, one workgroup per CU walking the tiles lo + wi, lo + wi + W, ... of its XCD label, with the tile boundary
// pipelined: the next tile's descriptor and first row ids / masks are loaded during steps 0 and 1 of the current tile (their
// latency under the MFMA batches, like every step's scalars), its first two stages are issued BEFORE the current tile's
// epilogue (which stages through ring slot 2 only, 32 rows at a time), and the first hand-overs of the next tile count the
// epilogue's stores as younger operations instead of draining them.  What a one-tile workgroup pays per tile -- dispatch,
// two dependent scalar round trips, the first DMA round trip (11 % of its life), the gap until the CU's next workgroup
// starts (4-5 % of the CUs idle at any time) -- overlaps the epilogue here.
__device__ __forceinline__ void cs_handover_rt(int n) {   // n in {0, 6, 7, 22, 23, 38, 39}: never MORE than asked for
    if (n >= 39) cs_handover<39>();
    else if (n >= 38) cs_handover<38>();
    else if (n >= 23) cs_handover<23>();
    else if (n >= 22) cs_handover<22>();
    else if (n >= 7) cs_handover<7>();
    else if (n >= 6) cs_handover<6>();
    else cs_handover<0>();
}

// scalar loads the compiler cannot turn into vector loads (a loop that also stores makes every load in it "clobberable", and
// a vector load would have to be waited for with vmcnt(0), draining the ring); the results are tied to the next lgkmcnt(0)
__device__ __forceinline__ const void *cs_uniform(const void *p) {   // (a wave-uniform pointer the compiler keeps in VGPRs)
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const void *)(uintptr_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ i32x4 cs_sload4(const void *p) {
    i32x4 r;
    p = cs_uniform(p);
    asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(r) : "s"(p));
    return r;
}
__device__ __forceinline__ unsigned cs_sload1(const void *p) {
    unsigned r;
    p = cs_uniform(p);
    asm volatile("s_load_dword %0, %1, 0x0" : "=s"(r) : "s"(p));
    return r;
}
__device__ __forceinline__ int64_t cs_sload2(const void *p) {
    int64_t r;
    p = cs_uniform(p);
    asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=s"(r) : "s"(p));
    return r;
}

__global__ void __launch_bounds__(512, 2)
cs_pool_persist_kernel(const _Float16 *__restrict__ x_hi, const _Float16 *__restrict__ x_lo, int64_t ld_x,
                       const int64_t *__restrict__ bu_off, const int32_t *__restrict__ bu_row, const uint32_t *__restrict__ bu_mask,
                       const _Float16 *__restrict__ wa_hi, const _Float16 *__restrict__ wa_lo, int64_t nv, int64_t nblocks,
                       _Float16 *__restrict__ y_hi, _Float16 *__restrict__ y_lo, int64_t ld_y, float *__restrict__ y_f32, int64_t ld_yf,
                       int64_t per_xcd, int ablate, const float *__restrict__ out_scale, unsigned *__restrict__ queue,
                       uint64_t *__restrict__ stamp) {
    using G = CsGeo<256, 3>;
    constexpr int NC = 256, NST = 3, D = 2, NU = 2;
    static_assert(G::DMA == 6, "hand-over counts below");
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = (int)(gridDim.x >> 3);
    const int64_t t_end = ((int64_t)(blockIdx.x & 7) + 1) * per_xcd < nblocks * 2 ? ((int64_t)(blockIdx.x & 7) + 1) * per_xcd : nblocks * 2;
    // ---- tiles: the first two of a workgroup are lo + wi and lo + wi + W; every further one is CLAIMED from the label's counter
    //      (queue[label], zero at launch; claim c -> tile lo + 2 W + c), so that the label's W workgroups always work on the W
    //      lowest unfinished tiles (what the hardware dispatcher does for one-tile workgroups).  Wave 0 issues the atomic in step
    //      0 of tile i for tile i + 2 -- its result lands in v250, a register named here and nowhere else (the build checks that
    //      the compiler stays below it), and is complete at the latest with the tile's last hand-over -- and publishes the tile
    //      in LDS word (i & 1) after the loop; every wave reads word ((i + 1) & 1) at the top of tile i + 1.
    const int64_t t_lo = (int64_t)(blockIdx.x & 7) * per_xcd;
    int64_t lb = t_lo + (blockIdx.x >> 3);
    auto leave = [&]() {                                     // the last workgroup out re-arms the counters for the next launch
        if (queue && tid == 0) {
            const unsigned done = atomicAdd(queue + 8, 1u);
            if (done == gridDim.x - 1) {
#pragma unroll
                for (int qq = 0; qq < 9; ++qq) queue[qq] = 0u;
            }
        }
    };
    if (lb >= t_end) { leave(); return; }
    // ---- DMA roles (see cs_pool_kernel)
    const int du = lane >> 5, dc = lane & 31;
    const int swz = (wv >> 1) & 1;
    const int t0 = du | (swz << 2), t1 = (2 + du) | (swz << 2);
    const int dsw0 = (dc ^ (2 * t0)) * 8, dsw1 = (dc ^ (2 * t1)) * 8;
    const _Float16 *wah, *wal;                               // the tile being STAGED (the next tile from its first issue on)
    int64_t dsrc0, dsrc1;
    auto issue = [&](i32x4 id, unsigned mk, int k, int slot) {
        unsigned char *dst = smem_raw + slot * G::STAGE;
        if (!(ablate & 2)) {
            const int ida = du ? id.y : id.x, idb = du ? id.w : id.z;
            const int64_t s0 = (int64_t)ida * ld_x + dsrc0, s1 = (int64_t)idb * ld_x + dsrc1;
            cs_glds16(x_hi + s0, dst + (4 * wv) * G::RB);
            cs_glds16(x_lo + s0, dst + G::PLANE + (4 * wv) * G::RB);
            cs_glds16(x_hi + s1, dst + (4 * wv) * G::RB + 1024);
            cs_glds16(x_lo + s1, dst + G::PLANE + (4 * wv) * G::RB + 1024);
        }
        if (!(ablate & 8)) {
            const int lo = ((mk >> wv) & 1u) ? lane * 8 : 0;
            cs_glds16(wah + (int64_t)k * (CS_NG * 512) + lo, dst + G::OFF_W + wv * 1024);
            cs_glds16(wal + (int64_t)k * (CS_NG * 512) + lo, dst + G::OFF_W + CS_WPL + wv * 1024);
        }
    };
    // ---- read roles
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem_raw;
    uint32_t addr[NU];
    {
        const uint32_t rowb = (uint32_t)(8 * g + q) * G::RB + (uint32_t)(wv * G::WC * 2) + (uint32_t)(p * 8);
        const uint32_t t = (uint32_t)(q | ((g & 1) << 2));
#pragma unroll
        for (int u = 0; u < NU; ++u) addr[u] = lds0 + ((rowb + 32u * u) ^ (t << 5));
    }
    const uint32_t addr_w = lds0 + G::OFF_W + lane * 16;
    const bool late = !(ablate & 32) && wv >= 4;
    const bool do_reads = !(ablate & 1);
    const float inv = 1.f / CS_WSCALE;
    const float so = (y_f32 && out_scale) ? out_scale[0] : 1.f;
    const int stores_per_tile = (y_hi ? 16 : 0) + (y_f32 ? 16 : 0);      // per wave, a full block (counted below)
    static_assert((size_t)CS_NW * 32 * G::EP * sizeof(float) <= (size_t)G::STAGE, "epilogue staging must fit in one slot");
    const int fl = lane & 15, fq = lane >> 4;
    const int er = lane >> 2, ec = (lane & 3) * 8;

    // ---- first tile: descriptor, stages 0 and 1
    int64_t ub0 = bu_off[lb >> 1];
    int n = (int)((bu_off[(lb >> 1) + 1] - ub0) / CS_KS);
    const int32_t *idg = bu_row + ub0 + 4 * wv;
    const uint32_t *mkg = bu_mask + ub0 / CS_KS;
    wah = wa_hi + ((ub0 / CS_KS) * CS_NG + wv) * 512;
    wal = wa_lo + ((ub0 / CS_KS) * CS_NG + wv) * 512;
    dsrc0 = (int)(lb & 1) * NC + dsw0;
    dsrc1 = (int)(lb & 1) * NC + dsw1;
    auto load_ids = [&](const int32_t *base, int k) { return *reinterpret_cast<const i32x4 *>(base + (int64_t)k * CS_KS); };   // (before the loop)
    auto sload_ids = [&](const int32_t *base, int k) { return cs_sload4(base + (int64_t)k * CS_KS); };
    unsigned mq[NST];
    i32x4 idv;
    {
        const int k1 = n > 1 ? 1 : 0, k2 = n > 2 ? 2 : n - 1;
        const i32x4 i0 = load_ids(idg, 0), i1 = load_ids(idg, k1);
        mq[0] = mkg[0];
        mq[1] = mkg[k1];
        issue(i0, mq[0], 0, 0);
        issue(i1, mq[1], k1, 1);
        idv = sload_ids(idg, k2);
        mq[2] = cs_sload1(mkg + k2);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(idv), "+s"(mq[2]));
        if (wv == 0 && lane == 0) {                          // tile 1 (static)
            const int t1s = lb + W < t_end ? (int)(lb + W) : -1;
            asm volatile("ds_write_b32 %0, %1" ::"v"(lds0 + (uint32_t)G::SMEM + 4u), "v"(t1s) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        cs_handover<6>();
    }
    int extra = 0;                                           // the previous tile's epilogue stores that may still be in flight
    uint64_t st_t0 = 0, st_loop = 0, st_epi = 0, st_entry = 0, st_tiles = 0, st_steps = 0;   // tuning aid (stamp != nullptr)
    if (stamp) st_t0 = cs_now();
    for (int ti = 0;; ++ti) {
        uint64_t st_a = 0, st_b = 0, st_c = 0;
        if (stamp) st_a = cs_now();
        const int64_t b = lb >> 1;
        const int col0 = (int)(lb & 1) * NC;
        int64_t lbn;
        {
            int nx;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(nx) : "v"(lds0 + (uint32_t)G::SMEM + 4u * ((ti + 1) & 1)) : "memory");
            lbn = __builtin_amdgcn_readfirstlane(nx);
        }
        const bool has_next = lbn >= 0;
        const bool claims = has_next && wv == 0 && queue != nullptr;     // this wave has one more vector-memory operation in steps 0 and 1
        f32x4 acc[CS_NG * NU];
#pragma unroll
        for (int i = 0; i < CS_NG * NU; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        s16x4 fb[NU][2][2];
        f16x8 ah0[4], al0[4], ah1[4], al1[4], bhp[NU], blp[NU];
#pragma unroll
        for (int i = 0; i < 4; ++i) { ah0[i] = al0[i] = ah1[i] = al1[i] = f16x8{0, 0, 0, 0, 0, 0, 0, 0}; }
#pragma unroll
        for (int u = 0; u < NU; ++u) { bhp[u] = blp[u] = f16x8{0, 0, 0, 0, 0, 0, 0, 0}; }
        unsigned mP = 0;
        auto mfma_hi = [&](unsigned m) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                if (__builtin_expect((m >> (4 + mt)) & 1u, 1)) {
#pragma unroll
                    for (int u = 0; u < NU; ++u) acc[(4 + mt) * NU + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1[mt], bhp[u], acc[(4 + mt) * NU + u], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < NU; ++u) acc[(4 + mt) * NU + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1[mt], blp[u], acc[(4 + mt) * NU + u], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < NU; ++u) acc[(4 + mt) * NU + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1[mt], bhp[u], acc[(4 + mt) * NU + u], 0, 0, 0);
                }
        };
        // the next tile: descriptor (loaded in step 0), first ids / masks (loaded in step 1)
        int64_t ub0n = 0, ub1n = 0;
        i32x4 ip0 = {0, 0, 0, 0}, ip1 = ip0, ip2 = ip0;
        unsigned mp0 = 0, mp1 = 0, mp2 = 0;
        auto next_heads = [&]() {
            const int nn = (int)((ub1n - ub0n) / CS_KS);
            const int32_t *ign = bu_row + ub0n + 4 * wv;
            const uint32_t *mgn = bu_mask + ub0n / CS_KS;
            const int k1 = nn > 1 ? 1 : 0, k2 = nn > 2 ? 2 : nn - 1;
            ip0 = sload_ids(ign, 0); ip1 = sload_ids(ign, k1); ip2 = sload_ids(ign, k2);
            mp0 = cs_sload1(mgn); mp1 = cs_sload1(mgn + k1); mp2 = cs_sload1(mgn + k2);
        };
        for (int s0 = 0; s0 < n; s0 += NST) {
#pragma unroll
            for (int J = 0; J < NST; ++J) {
                const int s = s0 + J;
                if (s < n) {
                    const uint32_t aw = addr_w + J * G::STAGE;
                    const unsigned m = mq[0];
                    if (do_reads) {
#pragma unroll
                        for (int u = 0; u < NU; ++u) {
                            const uint32_t au = addr[u] + J * G::STAGE;
                            cs_tr<0>(fb[u][0][0], au);
                            cs_tr<4 * G::RB>(fb[u][0][1], au);
                            cs_tr<G::PLANE>(fb[u][1][0], au);
                            cs_tr<G::PLANE + 4 * G::RB>(fb[u][1][1], au);
                        }
                        if (m & 1u) { cs_rd128<0 * 1024>(ah0[0], aw); cs_rd128<CS_WPL + 0 * 1024>(al0[0], aw); }
                        if (m & 2u) { cs_rd128<1 * 1024>(ah0[1], aw); cs_rd128<CS_WPL + 1 * 1024>(al0[1], aw); }
                        if (m & 4u) { cs_rd128<2 * 1024>(ah0[2], aw); cs_rd128<CS_WPL + 2 * 1024>(al0[2], aw); }
                        if (m & 8u) { cs_rd128<3 * 1024>(ah0[3], aw); cs_rd128<CS_WPL + 3 * 1024>(al0[3], aw); }
                    }
                    if (!late && s + D < n) issue(idv, mq[D], s + D, (J + D) % NST);
                    if (s == 0 && claims && lane == 0) {
                        const unsigned *qa = queue + (blockIdx.x & 7);
                        const unsigned one = 1u;
                        asm volatile("global_atomic_add v250, %0, %1, off sc0" ::"v"(qa), "v"(one) : "memory", "v250");
                    }
                    if (do_reads) {
                        mfma_hi(mP);
                        asm volatile("s_waitcnt lgkmcnt(0)"
                                     : "+v"(fb[0][0][0]), "+v"(fb[0][0][1]), "+v"(fb[0][1][0]), "+v"(fb[0][1][1]), "+v"(fb[1][0][0]),
                                       "+v"(fb[1][0][1]), "+v"(fb[1][1][0]), "+v"(fb[1][1][1]));
                        cs_wait_a(ah0, al0);
                    }
                    const int kn = s + D + 1 < n ? s + D + 1 : n - 1;
                    i32x4 idn = sload_ids(idg, kn);
                    unsigned mN = cs_sload1(mkg + kn);
                    if (has_next) {                          // (scalar loads: waited for before the barrier, like idn / mN)
                        if (s == 0) { ub0n = cs_sload2(bu_off + (lbn >> 1)); ub1n = cs_sload2(bu_off + (lbn >> 1) + 1); }
                        if (s == 1) next_heads();
                    }
                    if (do_reads) {
                        if (m & 16u) { cs_rd128<4 * 1024>(ah1[0], aw); cs_rd128<CS_WPL + 4 * 1024>(al1[0], aw); }
                        if (m & 32u) { cs_rd128<5 * 1024>(ah1[1], aw); cs_rd128<CS_WPL + 5 * 1024>(al1[1], aw); }
                        if (m & 64u) { cs_rd128<6 * 1024>(ah1[2], aw); cs_rd128<CS_WPL + 6 * 1024>(al1[2], aw); }
                        if (m & 128u) { cs_rd128<7 * 1024>(ah1[3], aw); cs_rd128<CS_WPL + 7 * 1024>(al1[3], aw); }
#pragma unroll
                        for (int u = 0; u < NU; ++u) { bhp[u] = cs_cat(fb[u][0][0], fb[u][0][1]); blp[u] = cs_cat(fb[u][1][0], fb[u][1][1]); }
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
                            if (__builtin_expect((m >> mt) & 1u, 1)) {
#pragma unroll
                                for (int u = 0; u < NU; ++u) acc[mt * NU + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0[mt], bhp[u], acc[mt * NU + u], 0, 0, 0);
#pragma unroll
                                for (int u = 0; u < NU; ++u) acc[mt * NU + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0[mt], blp[u], acc[mt * NU + u], 0, 0, 0);
#pragma unroll
                                for (int u = 0; u < NU; ++u) acc[mt * NU + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0[mt], bhp[u], acc[mt * NU + u], 0, 0, 0);
                            }
                    }
                    if (late && s + D < n) issue(idv, mq[D], s + D, (J + D) % NST);
                    cs_wait_a(ah1, al1);
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(idn), "+s"(mN), "+s"(ub0n), "+s"(ub1n), "+s"(ip0), "+s"(ip1), "+s"(ip2), "+s"(mp0), "+s"(mp1), "+s"(mp2));
                    // stage s + 1 has landed; in step 0 the previous tile's epilogue stores (issued between stages 1 and 2) may still
                    // be in flight and count as younger operations
                    if (s + D < n) cs_handover_rt(6 + (s == 0 ? extra : 0) + ((claims && s <= 1) ? 1 : 0));
                    else cs_handover<0>();
                    mP = m;
                    mq[0] = mq[1]; mq[1] = mq[2]; mq[2] = mN;
                    idv = idn;
                }
            }
        }
        if (stamp) { st_b = cs_now(); st_loop += st_b - st_a; ++st_tiles; st_steps += n; }
        if (do_reads) mfma_hi(mP);                           // groups 4-7 of the last step
        if (has_next && wv == 0) {                           // tile ti + 2 (the atomic is complete: the last hand-over waited for vmcnt(0))
            int t2 = -1;
            if (queue) {
                unsigned c;
                asm volatile("v_readfirstlane_b32 %0, v250" : "=s"(c)::"memory");
                const int64_t cl = t_lo + 2 * (int64_t)W + c;
                t2 = cl < t_end ? (int)cl : -1;
            } else if (lbn + W < t_end) {
                t2 = (int)(lbn + W);
            }
            if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(lds0 + (uint32_t)G::SMEM + 4u * (ti & 1)), "v"(t2) : "memory");
        }
        if (has_next && n < 2) {                             // (a one-step tile never reached the look-ahead loads)
            ub0n = cs_sload2(bu_off + (lbn >> 1));
            ub1n = cs_sload2(bu_off + (lbn >> 1) + 1);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ub0n), "+s"(ub1n));
            next_heads();
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ip0), "+s"(ip1), "+s"(ip2), "+s"(mp0), "+s"(mp1), "+s"(mp2));
        }
        // ---- the next tile's first two stages go out before the epilogue (the ring is drained: the last hand-over waited for
        //      vmcnt(0), and every wave is past its LDS reads)
        int nn = 0;
        if (has_next) {
            nn = (int)((ub1n - ub0n) / CS_KS);
            wah = wa_hi + ((ub0n / CS_KS) * CS_NG + wv) * 512;
            wal = wa_lo + ((ub0n / CS_KS) * CS_NG + wv) * 512;
            dsrc0 = (int)(lbn & 1) * NC + dsw0;
            dsrc1 = (int)(lbn & 1) * NC + dsw1;
            issue(ip0, mp0, 0, 0);
            issue(ip1, mp1, nn > 1 ? 1 : 0, 1);
        }
        // ---- epilogue through ring slot 2, 32 rows at a time (see cs_pool_kernel for the scaling conventions)
        const int64_t row0 = b * CS_BR;
        if (!(ablate & 4)) {
            const int colw = col0 + wv * G::WC;
            // (LDS traffic in inline asm: a compiler-visible LDS access would first wait for the LDS-DMA in flight, vmcnt(0); a wave's
            //  LDS operations execute in order, so its reads see its writes and the next pass's writes follow this pass's reads)
            const uint32_t sw = lds0 + 2 * G::STAGE + (uint32_t)wv * (32 * G::EP * 4) + (uint32_t)((fq * 4 * G::EP + fl) * 4);
            const uint32_t sr = lds0 + 2 * G::STAGE + (uint32_t)wv * (32 * G::EP * 4) + (uint32_t)((er * G::EP + ec) * 4);
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int cb = 0; cb < NU; ++cb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float val = acc[(ch * 2 + mt) * NU + cb][r] * inv;
                            asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(sw), "v"(val), "n"(((mt * 16 + r) * G::EP + cb * 16) * 4) : "memory");
                        }
                f32x4 v[2][2];
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[it][0]) : "v"(sr), "n"(it * 16 * G::EP * 4) : "memory");
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[it][1]) : "v"(sr), "n"(it * 16 * G::EP * 4 + 16) : "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[1][0]), "+v"(v[1][1])::"memory");
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int64_t grow = row0 + ch * 32 + it * 16 + er;
                    if (grow < nv && !(ablate & 16)) {
                        const float xv[8] = {v[it][0].x, v[it][0].y, v[it][0].z, v[it][0].w, v[it][1].x, v[it][1].y, v[it][1].z, v[it][1].w};
                        if (y_hi) {
                            f16x8 h, l;
#pragma unroll
                            for (int i = 0; i < 8; ++i) { h[i] = (_Float16)xv[i]; l[i] = (_Float16)(xv[i] - (float)h[i]); }
                            *reinterpret_cast<f16x8 *>(y_hi + grow * ld_y + colw + ec) = h;
                            *reinterpret_cast<f16x8 *>(y_lo + grow * ld_y + colw + ec) = l;
                        }
                        if (y_f32) {
                            float *yp = y_f32 + grow * ld_yf + colw + ec;
                            *reinterpret_cast<float4 *>(yp) = make_float4(xv[0] * so, xv[1] * so, xv[2] * so, xv[3] * so);
                            *reinterpret_cast<float4 *>(yp + 4) = make_float4(xv[4] * so, xv[5] * so, xv[6] * so, xv[7] * so);
                        }
                    }
                }
            }
        }
        if (stamp) { st_c = cs_now(); st_epi += st_c - st_b; }
        if (!has_next) break;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (wave 0's publication is in LDS before the barrier below)
        // ---- on to the next tile: stage 0 must have landed; stage 1 and this epilogue's stores may stay in flight.  The store
        //      count is only known for a full block written with every store (no tuning bits): otherwise drain them.
        extra = (row0 + CS_BR <= nv && !(ablate & (4 | 16))) ? stores_per_tile : 0;
        const bool counted = extra != 0 || (stores_per_tile == 0);
        lb = lbn;
        ub0 = ub0n;
        n = nn;
        idg = bu_row + ub0 + 4 * wv;
        mkg = bu_mask + ub0 / CS_KS;
        mq[0] = mp0; mq[1] = mp1; mq[2] = mp2;
        idv = ip2;
        if (counted) cs_handover_rt(6 + extra);
        else { cs_handover<6>(); }
        if (stamp) st_entry += cs_now() - st_c;
    }
    if (stamp && lane == 0) {
        uint64_t *o = stamp + ((int64_t)blockIdx.x * CS_NW + wv) * 10;
        o[0] = st_loop; o[1] = st_epi; o[2] = st_entry; o[3] = cs_now() - st_t0; o[4] = st_tiles; o[5] = st_steps; o[6] = cs_real();
    }
    leave();
}

// ------------------------------------------------------------------------------------------------ engine
// Producer / consumer form of the same operator ("engine"): ONE persistent 512-thread workgroup per CU.
//   waves 4-7 = loaders (one per SIMD): nothing but LDS-DMA.  They fill a ring of four 32-KiB slots (32 union rows x 128
//               columns x {hi, lo} + the step's 8 x {hi, lo} weight fragments) as fast as slots come free -- up to three
//               stages (96 KiB) in flight per CU, across tile boundaries -- and absorb the memory pipeline's back-pressure;
//   waves 0-3 = consumers (one per SIMD): wave cw owns all 128 rows x 32 columns of the tile (16 accumulator tiles) and
//               keeps TWO stages' operands in registers: the reads of stage g + 1 run under the MFMAs of stage g, and a
//               slot is released as soon as its operands have landed in registers.
// A tile is (128-row block, 128-column quarter).  The tiles of an XCD label are CLAIMED in order from the label's counter
// (queue[label], one returning atomic per tile, issued by loader 0 one tile ahead and collected a stage later, when its
// queue is drained anyway), so that the label's 32 workgroups always work on the 32 lowest unfinished tiles = 8
// neighbouring row blocks: a 1 024-row window whose union rows fit the XCD's 4-MiB L2 (static strided lists let the
// workgroups drift apart: 1.30 GB from memory per application instead of 1.08).  Loader 0 publishes {tile, steps, first
// union row} in an LDS ring that the other eleven waves follow.
// There is no s_barrier: slot hand-over goes through two monotonic LDS counters per slot (full: +1 per loader once its
// DMA has landed, s_waitcnt vmcnt; free: +1 per consumer once its operand reads have landed, s_waitcnt lgkmcnt), which the
// other side polls.  Consumers issue no LDS-DMA, so their epilogue (wave-private LDS staging, stores) is plain code and
// overlaps the loaders' work on the next tile.
constexpr int EG_NC = 128;                          // columns per tile
constexpr int EG_RB = EG_NC * 2;                    // bytes per staged row and plane
constexpr int EG_PLANE = CS_KS * EG_RB;             // 8 KiB
constexpr int EG_OFF_W = 2 * EG_PLANE;              // 16 KiB
constexpr int EG_SLOT = EG_OFF_W + 2 * CS_WPL;      // 32 KiB
constexpr int EG_NSLOT = 4;
constexpr int EG_NCW = 4;                           // consumer waves (32 columns each, one per SIMD)
constexpr int EG_NLW = 4;                           // loader waves
constexpr int EG_THREADS = 64 * (EG_NCW + EG_NLW);
constexpr int EG_OFF_FLAG = EG_NSLOT * EG_SLOT;     // {full, fragment mask} x 4 | free[4] (+32) | published tiles (+48)
constexpr int EG_TQ = 16;                           // ring of published tiles: {tile, steps, first union row / 32, -}
constexpr int EG_OFF_TQ = EG_OFF_FLAG + 64;
constexpr int EG_OFF_STG = EG_OFF_TQ + EG_TQ * 16;  // epilogue staging: 4 waves x 32 rows x CS_EP floats
constexpr int EG_STG_WAVE = 32 * CS_EP * 4;
constexpr size_t EG_SMEM = (size_t)EG_OFF_STG + EG_NCW * EG_STG_WAVE;
static_assert(EG_SMEM <= 160 * 1024, "engine LDS");

__device__ __forceinline__ void eg_wait_ge(uint32_t flag_addr, uint32_t target) {
    for (;;) {
        uint32_t v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(flag_addr) : "memory");
        if ((int32_t)(__builtin_amdgcn_readfirstlane(v) - target) >= 0) break;
        __builtin_amdgcn_s_sleep(1);
    }
}
__device__ __forceinline__ void eg_signal(uint32_t flag_addr) {
    if ((threadIdx.x & 63) == 0) {
        const uint32_t one = 1;
        asm volatile("ds_add_u32 %0, %1" ::"v"(flag_addr), "v"(one) : "memory");
    }
}
// entry i of the workgroup's tile sequence (published by loader 0): tile < 0 = no more tiles
__device__ __forceinline__ void eg_tile(uint32_t lds0, int i, int &t, int &n, int64_t &ub0) {
    eg_wait_ge(lds0 + EG_OFF_FLAG + 48, (uint32_t)(i + 1));
    i32x4 e;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e) : "v"(lds0 + EG_OFF_TQ + (uint32_t)(i & (EG_TQ - 1)) * 16u) : "memory");
    t = __builtin_amdgcn_readfirstlane(e.x);
    n = __builtin_amdgcn_readfirstlane(e.y);
    ub0 = (int64_t)__builtin_amdgcn_readfirstlane(e.z) * CS_KS;
}

template <bool STAMP>
__global__ void __launch_bounds__(EG_THREADS, 2)
cs_engine_kernel(const _Float16 *__restrict__ x_hi, const _Float16 *__restrict__ x_lo, int64_t ld_x,
                 const int64_t *__restrict__ bu_off, const int32_t *__restrict__ bu_row, const uint32_t *__restrict__ bu_mask,
                 const _Float16 *__restrict__ wa_hi, const _Float16 *__restrict__ wa_lo, int64_t nv, int64_t nblocks,
                 _Float16 *__restrict__ y_hi, _Float16 *__restrict__ y_lo, int64_t ld_y, float *__restrict__ y_f32, int64_t ld_yf,
                 int ablate, const float *__restrict__ out_scale, unsigned *__restrict__ queue, uint64_t *__restrict__ stamp) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem_raw;
    if (tid < 16) reinterpret_cast<uint32_t *>(smem_raw + EG_OFF_FLAG)[tid] = 0u;
    __syncthreads();
    // ---- the tiles: label q = blockIdx & 7 owns the contiguous tile range [t_lo, t_hi); workgroup wi of the label starts with
    //      tile t_lo + wi and claims t_lo + W + c, c = 0, 1, ... from queue[label]   (tile = 4 * row block + column quarter)
    const int64_t T = nblocks * (CS_D / EG_NC);
    const int label = blockIdx.x & 7, wi = blockIdx.x >> 3, W = (int)(gridDim.x >> 3);
    const int64_t t_lo = label * T / 8, t_hi = (label + 1) * T / 8;
    uint64_t st_t0 = 0, st_poll = 0, st_work = 0, st_epi = 0, st_steps = 0;
    if constexpr (STAMP) st_t0 = cs_now();

    if (wv >= EG_NCW) {
        // ================================================================ loaders
        // Loader l owns ring slot l and stages l, l + 4, l + 8, ... of the workgroup's stage sequence (all steps of all its tiles
        // in order): it waits until the consumers have released the slot, issues the WHOLE stage (16 x 1 KiB of rows, the
        // non-empty weight fragments), writes the stage's fragment mask next to the slot's counter, waits for its own DMA
        // (vmcnt(0): nothing else is in its queue) and signals.
        const int l = wv - EG_NCW;
        const int du = lane >> 4, dc = lane & 15;
        const int dsw0 = (dc ^ (2 * du)) * 8, dsw1 = (dc ^ (2 * (du | 4))) * 8;  // source column of this lane's chunk: rows 0-7 / 8-15 (mod 16)
        // ---- loader 0 also feeds the tile ring: entry i + 1 is claimed when it enters tile i, in three phases a stage apart
        //      (atomic issued | claimed tile's two union offsets loaded | published), or at once if a short tile needs it earlier
        int pub = 0, phase = 0;
        bool ended = false;
        unsigned cv = 0;
        int64_t tn = 0, o0 = 0, o1 = 0;
        auto publish = [&](int t_, int n_, int u_) {
            if (lane == 0) {
                const i32x4 e = {t_, n_, u_, 0};
                const uint32_t ea = lds0 + EG_OFF_TQ + (uint32_t)(pub & (EG_TQ - 1)) * 16u, ca = lds0 + EG_OFF_FLAG + 48;
                const uint32_t one = 1;
                asm volatile("ds_write_b128 %0, %1\n\tds_add_u32 %2, %3" ::"v"(ea), "v"(e), "v"(ca), "v"(one) : "memory");
            }
            ++pub;
        };
        auto advance = [&]() {
            if (phase == 0) {
                if (lane == 0) cv = atomicAdd(queue + label, 1u);
                phase = 1;
            } else if (phase == 1) {
                tn = t_lo + W + (int64_t)__builtin_amdgcn_readfirstlane(cv);
                if (tn >= t_hi) {
                    publish(-1, 0, 0);
                    ended = true;
                    phase = 0;
                } else {
                    o0 = bu_off[tn >> 2];
                    o1 = bu_off[(tn >> 2) + 1];
                    phase = 2;
                }
            } else {
                publish((int)tn, (int)((o1 - o0) / CS_KS), (int)(o0 / CS_KS));
                phase = 0;
            }
        };
        int ti = 0, t, n;
        int64_t ub0;
        if (l == 0) {
            const int64_t t0 = t_lo + wi;
            if (t0 < t_hi) {
                const int64_t a0 = bu_off[t0 >> 2], a1 = bu_off[(t0 >> 2) + 1];
                publish((int)t0, (int)((a1 - a0) / CS_KS), (int)(a0 / CS_KS));
                advance();                                                    // entry 1: the atomic goes out now
            } else {
                publish(-1, 0, 0);
                ended = true;
            }
        }
        eg_tile(lds0, 0, t, n, ub0);
        int k = l;                                                            // this loader's next stage = step k of tile ti (k may run past n)
        uint32_t j = 0;                                                       // uses of the slot so far
        while (t >= 0) {
            while (k >= n) {                                                  // on to the tile that holds the stage
                k -= n;
                ++ti;
                if (l == 0) {
                    while (!ended && pub < ti + 1) advance();                 // (a tile shorter than three of this loader's stages)
                    if (!ended && phase == 0 && pub < ti + 2) advance();      // entering tile ti: claim entry ti + 1
                }
                eg_tile(lds0, ti, t, n, ub0);
                if (t < 0) break;
            }
            if (t < 0) break;
            const int col0 = (t & 3) * EG_NC;
            const int32_t *idg = bu_row + ub0 + (int64_t)k * CS_KS;
            i32x4 id[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) id[i] = *reinterpret_cast<const i32x4 *>(idg + 4 * i);
            const unsigned mk = bu_mask[ub0 / CS_KS + k];
            eg_wait_ge(lds0 + EG_OFF_FLAG + 32 + l * 4, (uint32_t)EG_NCW * j);  // free[l]: the slot's previous stage is consumed
            unsigned char *dst = smem_raw + l * EG_SLOT;
            if (!(ablate & 2)) {                                              // tuning aid: bit 1 skips the row gather
#pragma unroll
                for (int i = 0; i < 8; ++i) {                                 // rows 4 i .. 4 i + 3
                    const int idr = du == 0 ? id[i].x : du == 1 ? id[i].y : du == 2 ? id[i].z : id[i].w;
                    const int64_t so = (int64_t)idr * ld_x + col0 + (((i >> 1) & 1) ? dsw1 : dsw0);
                    cs_glds16(x_hi + so, dst + (4 * i) * EG_RB);
                    cs_glds16(x_lo + so, dst + EG_PLANE + (4 * i) * EG_RB);
                }
            }
            const int64_t wk = (ub0 / CS_KS + k) * (CS_NG * 512) + lane * 8;
#pragma unroll
            for (int gq = 0; gq < CS_NG; ++gq)
                if ((mk >> gq) & 1u) {                                        // empty fragments are neither fetched nor read
                    cs_glds16(wa_hi + wk + gq * 512, dst + EG_OFF_W + gq * 1024);
                    cs_glds16(wa_lo + wk + gq * 512, dst + EG_OFF_W + CS_WPL + gq * 1024);
                }
            if (lane == 0) {
                const uint32_t ma = lds0 + EG_OFF_FLAG + l * 8 + 4;
                asm volatile("ds_write_b32 %0, %1" ::"v"(ma), "v"(mk) : "memory");
            }
            k += 4;
            ++j;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            eg_signal(lds0 + EG_OFF_FLAG + l * 8);                              // full[l]
            if (l == 0 && !ended && phase != 0) advance();                    // the claim in flight moves one phase per stage
        }
        if (l == 0 && lane == 0) {                                            // the last workgroup out re-arms the counters for the next launch
            const unsigned done = atomicAdd(queue + 8, 1u);
            if (done == gridDim.x - 1) {
#pragma unroll
                for (int q = 0; q < 9; ++q) queue[q] = 0u;
            }
        }
        return;
    }
    // ==================================================================== consumers
    // Software pipeline over the workgroup's whole stage sequence (across tile boundaries), half a stage deep: the MFMAs of
    // groups 0-3 of stage g run under the reads of its groups 4-7 (after which the slot is released), the MFMAs of groups 4-7
    // under the reads of stage g + 1's rows (second register set) and groups 0-3; the {full, mask} word of stage g + 1 is
    // sampled along with the first batch, so that in the steady state (loaders ahead) no poll round trip is exposed.
    const int cw = wv;
    const int gq = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    uint32_t addr[2];
    {
        const uint32_t rowb = (uint32_t)(8 * gq + q) * EG_RB + (uint32_t)(cw * CS_WC * 2) + (uint32_t)(p * 8);
        const uint32_t t5 = (uint32_t)(q | ((gq & 1) << 2)) << 5;
        addr[0] = lds0 + (rowb ^ t5);
        addr[1] = lds0 + ((rowb + 32u) ^ t5);
    }
    const uint32_t addr_w = lds0 + EG_OFF_W + lane * 16;
    const float inv = 1.f / CS_WSCALE;
    const float so = (y_f32 && out_scale) ? out_scale[0] : 1.f;
    float *stg = reinterpret_cast<float *>(smem_raw + EG_OFF_STG + cw * EG_STG_WAVE);
    const int fl = lane & 15, fq = lane >> 4;
    const int er = lane >> 2, ec = (lane & 3) * 8;
    const bool do_mma = !(ablate & 1);
    struct Rows { s16x4 f[2][2][2]; };                       // one stage's staged rows as B fragments: [column tile][hi, lo][k half]
    Rows P, Q;
    f16x8 ah[8], al[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ah[i] = al[i] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    int2 fs = {0, 0};                                        // the sampled {full, mask} word of the next stage
    auto poll = [&](uint32_t gg) -> unsigned {               // blocking: stage gg is full; returns its fragment mask
        unsigned m;
        for (;;) {
            int2 fm;
            asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(fm) : "v"(lds0 + EG_OFF_FLAG + (gg & 3u) * 8) : "memory");
            m = (unsigned)__builtin_amdgcn_readfirstlane(fm.y);
            if ((int32_t)((uint32_t)__builtin_amdgcn_readfirstlane(fm.x) - ((gg >> 2) + 1u)) >= 0) break;
            __builtin_amdgcn_s_sleep(1);
        }
        return m;
    };
    auto issue_rows = [&](Rows &o, uint32_t gg) {
        if (ablate & 256) return;                            // tuning aid: bit 8 skips the staged-row reads
        const uint32_t so_ = (gg & 3u) * EG_SLOT;
        const uint32_t a0 = addr[0] + so_, a1 = addr[1] + so_;
        cs_tr<0>(o.f[0][0][0], a0);
        cs_tr<4 * EG_RB>(o.f[0][0][1], a0);
        cs_tr<EG_PLANE>(o.f[0][1][0], a0);
        cs_tr<EG_PLANE + 4 * EG_RB>(o.f[0][1][1], a0);
        cs_tr<0>(o.f[1][0][0], a1);
        cs_tr<4 * EG_RB>(o.f[1][0][1], a1);
        cs_tr<EG_PLANE>(o.f[1][1][0], a1);
        cs_tr<EG_PLANE + 4 * EG_RB>(o.f[1][1][1], a1);
    };
    auto issue_w03 = [&](uint32_t gg, unsigned m) {
        const uint32_t aw = addr_w + (gg & 3u) * EG_SLOT;
        cs_rd128_if<0, 0 * 1024, CS_WPL + 0 * 1024>(ah[0], al[0], aw, m);
        cs_rd128_if<1, 1 * 1024, CS_WPL + 1 * 1024>(ah[1], al[1], aw, m);
        cs_rd128_if<2, 2 * 1024, CS_WPL + 2 * 1024>(ah[2], al[2], aw, m);
        cs_rd128_if<3, 3 * 1024, CS_WPL + 3 * 1024>(ah[3], al[3], aw, m);
    };
    auto issue_w47 = [&](uint32_t gg, unsigned m) {          // ... and the sample of stage gg + 1's word behind them
        const uint32_t aw = addr_w + (gg & 3u) * EG_SLOT;
        cs_rd128_if<4, 4 * 1024, CS_WPL + 4 * 1024>(ah[4], al[4], aw, m);
        cs_rd128_if<5, 5 * 1024, CS_WPL + 5 * 1024>(ah[5], al[5], aw, m);
        cs_rd128_if<6, 6 * 1024, CS_WPL + 6 * 1024>(ah[6], al[6], aw, m);
        cs_rd128_if<7, 7 * 1024, CS_WPL + 7 * 1024>(ah[7], al[7], aw, m);
        asm volatile("ds_read_b64 %0, %1" : "=v"(fs) : "v"(lds0 + EG_OFF_FLAG + ((gg + 1u) & 3u) * 8) : "memory");
    };
    auto landed_first = [&](Rows &o) {                       // rows + groups 0-3
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(ah[0]), "+v"(ah[1]), "+v"(ah[2]), "+v"(ah[3]), "+v"(al[0]), "+v"(al[1]), "+v"(al[2]), "+v"(al[3]),
                       "+v"(o.f[0][0][0]), "+v"(o.f[0][0][1]), "+v"(o.f[0][1][0]), "+v"(o.f[0][1][1]), "+v"(o.f[1][0][0]),
                       "+v"(o.f[1][0][1]), "+v"(o.f[1][1][0]), "+v"(o.f[1][1][1]));
    };
    auto landed_second = [&](uint32_t gg) {                  // groups 4-7 + the sample: stage gg's slot is free
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(ah[4]), "+v"(ah[5]), "+v"(ah[6]), "+v"(ah[7]), "+v"(al[4]), "+v"(al[5]), "+v"(al[6]), "+v"(al[7]), "+v"(fs));
        eg_signal(lds0 + EG_OFF_FLAG + 32 + (gg & 3u) * 4);
    };
    int t, n;
    int64_t ub0;
    eg_tile(lds0, 0, t, n, ub0);
    uint32_t g = 0;
    unsigned m_cur = 0;
    if (t >= 0) {                                            // prologue: rows and groups 0-3 of stage 0
        m_cur = poll(0);
        if (ablate & 512) m_cur = 0;
        issue_rows(P, 0);
        issue_w03(0, m_cur);
        landed_first(P);
    }
    for (int ti = 0; t >= 0; ++ti) {
        const int64_t b = t >> 2;
        const int col0 = (t & 3) * EG_NC;
        f32x4 acc[CS_NG * 2];
#pragma unroll
        for (int i = 0; i < CS_NG * 2; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        // groups MT0 .. MT0 + 3 of the current stage: per accumulator + hi*hi, + hi*lo, + lo*hi (the order of cs_pool_kernel)
        auto mma = [&](const Rows &cur, auto mt0) {
            if (!do_mma) return;
            constexpr int MT0 = decltype(mt0)::value;
            f16x8 bh[2], bl[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                bh[u] = cs_cat(cur.f[u][0][0], cur.f[u][0][1]);
                bl[u] = cs_cat(cur.f[u][1][0], cur.f[u][1][1]);
            }
#pragma unroll
            for (int mt = MT0; mt < MT0 + 4; ++mt)
                if (__builtin_expect((m_cur >> mt) & 1u, 1)) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[mt * 2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bh[u], acc[mt * 2 + u], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[mt * 2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bl[u], acc[mt * 2 + u], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[mt * 2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt], bh[u], acc[mt * 2 + u], 0, 0, 0);
                }
        };
        int t_n = 0, n_n = 0;
        int64_t ub0_n = 0;
        for (int k = 0; k < n; ++k, ++g) {
            uint64_t st_a = 0;
            if constexpr (STAMP) st_a = cs_now();
            unsigned m_nxt = 0;
            bool more = true;
#define EG_STEP(CUR, NXT)                                                                                                          \
            {                                                                                                                      \
                issue_w47(g, m_cur);                                                                                               \
                mma(CUR, std::integral_constant<int, 0>{});                                                                        \
                landed_second(g);                                                                                                  \
                if (k == n - 1) {                            /* the last step of a tile looks up the next tile */                  \
                    eg_tile(lds0, ti + 1, t_n, n_n, ub0_n);                                                                        \
                    more = t_n >= 0;                                                                                               \
                }                                                                                                                  \
                if (more) {                                                                                                        \
                    uint64_t st_b = 0;                                                                                             \
                    if constexpr (STAMP) st_b = cs_now();                                                                          \
                    if ((int32_t)((uint32_t)__builtin_amdgcn_readfirstlane(fs.x) - (((g + 1u) >> 2) + 1u)) >= 0)                   \
                        m_nxt = (unsigned)__builtin_amdgcn_readfirstlane(fs.y);     /* the sample already saw it full */          \
                    else                                                                                                           \
                        m_nxt = poll(g + 1u);                                                                                      \
                    if constexpr (STAMP) st_poll += cs_now() - st_b;                                                               \
                    if (ablate & 512) m_nxt = 0;             /* tuning aid: bit 9 skips the weight-fragment reads (and MFMAs) */   \
                    issue_rows(NXT, g + 1u);                                                                                       \
                    issue_w03(g + 1u, m_nxt);                                                                                      \
                }                                                                                                                  \
                mma(CUR, std::integral_constant<int, 4>{});                                                                        \
                if (more) landed_first(NXT);                                                                                       \
            }
            EG_STEP(P, Q)
            if (more) P = Q;                                 // 16 register moves per step; no second copy of the loop body
#undef EG_STEP
            m_cur = m_nxt;
            if constexpr (STAMP) { st_work += cs_now() - st_a; ++st_steps; }
        }
        t = t_n; n = n_n; ub0 = ub0_n;
        if (ablate & 4) continue;                            // tuning aid: bit 2 skips the epilogue
        // ---- epilogue: 32 rows at a time through the wave's private staging area; every store instruction writes 16 rows x 64 bytes
        uint64_t st_e = 0;
        if constexpr (STAMP) st_e = cs_now();
        const int64_t row0 = b * CS_BR;
        const int colw = col0 + cw * CS_WC;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) stg[(mt * 16 + fq * 4 + r) * CS_EP + cb * 16 + fl] = acc[(ch * 2 + mt) * 2 + cb][r] * inv;
            gp_wave_sync();
            float4 v[2][2];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const float *sp = stg + (it * 16 + er) * CS_EP + ec;
                v[it][0] = *reinterpret_cast<const float4 *>(sp);
                v[it][1] = *reinterpret_cast<const float4 *>(sp + 4);
            }
            gp_wave_sync();
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int64_t grow = row0 + ch * 32 + it * 16 + er;
                if (grow < nv && !(ablate & 16)) {
                    const float xv[8] = {v[it][0].x, v[it][0].y, v[it][0].z, v[it][0].w, v[it][1].x, v[it][1].y, v[it][1].z, v[it][1].w};
                    if (y_hi) {
                        f16x8 h, lo8;
#pragma unroll
                        for (int i = 0; i < 8; ++i) { h[i] = (_Float16)xv[i]; lo8[i] = (_Float16)(xv[i] - (float)h[i]); }
                        *reinterpret_cast<f16x8 *>(y_hi + grow * ld_y + colw + ec) = h;
                        *reinterpret_cast<f16x8 *>(y_lo + grow * ld_y + colw + ec) = lo8;
                    }
                    if (y_f32) {
                        float *yp = y_f32 + grow * ld_yf + colw + ec;
                        *reinterpret_cast<float4 *>(yp) = make_float4(xv[0] * so, xv[1] * so, xv[2] * so, xv[3] * so);
                        *reinterpret_cast<float4 *>(yp + 4) = make_float4(xv[4] * so, xv[5] * so, xv[6] * so, xv[7] * so);
                    }
                }
            }
        }
        if constexpr (STAMP) st_epi += cs_now() - st_e;
    }
    if constexpr (STAMP) {
        const uint64_t t3 = cs_now();
        if (lane == 0 && stamp) {
            uint64_t *o = stamp + ((int64_t)blockIdx.x * EG_NCW + cw) * 10;
            o[0] = 0; o[1] = 0; o[2] = 0; o[3] = st_work; o[4] = st_poll; o[5] = 0; o[6] = st_epi; o[7] = t3 - st_t0; o[8] = st_steps; o[9] = 0;
        }
    }
}

size_t cs_scan_tmp(int64_t n) {
    size_t t = 0;
    (void)rocprim::exclusive_scan(nullptr, t, (int64_t *)nullptr, (int64_t *)nullptr, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), 0);
    return t;
}
int cs_np2(int64_t n) { int p = 1; while (p < n) p <<= 1; return p; }

}  // namespace

extern "C" size_t gp_pool_cs_workspace_bytes(int64_t nv) {
    if (nv <= 0) return 0;
    int64_t nb = (nv + CS_BR - 1) / CS_BR;
    GpCarver cv(nullptr, 0);
    cv.take<int64_t>(nb + 1);
    cv.take<char>(cs_scan_tmp(nb + 1));
    return cv.off;
}

// pass 1: bu_off i64 [nblocks+1] (padded union rows before each 128-row block; multiples of 32), bu_n i32 [nblocks]
extern "C" int gp_pool_cs_count(const int32_t *nbr, int64_t nv, int32_t k, int64_t *bu_off, int32_t *bu_n, void *workspace,
                                size_t workspace_bytes, void *stream_) {
    GP_CHECK_ARG(nbr && bu_off && bu_n && workspace && nv > 0 && k > 0, "gp_pool_cs_count: null/empty argument");
    GP_CHECK_ARG((int64_t)CS_BR * k <= CS_MAXNK, "gp_pool_cs_count: k=%d too large (128*k <= %d)", k, CS_MAXNK);
    int64_t nb = (nv + CS_BR - 1) / CS_BR;
    GpCarver cv(workspace, workspace_bytes);
    int64_t *cnt = cv.take<int64_t>(nb + 1);
    size_t tb = cs_scan_tmp(nb + 1);
    char *tmp = cv.take<char>(tb);
    if (!cv.ok()) { gp_set_error("gp_pool_cs_count: workspace too small"); return GP_ENOMEM; }
    hipStream_t s = gp_stream(stream_);
    GP_CHECK_HIP(hipMemsetAsync(cnt + nb, 0, sizeof(int64_t), s));
    size_t sm = (size_t)(CS_HS + cs_np2((int64_t)CS_BR * k)) * sizeof(int);
    GP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(cs_count_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (CS_HS + CS_MAXID) * (int)sizeof(int)));
    cs_count_kernel<<<(unsigned)nb, 1024, sm, s>>>(nbr, nv, k, cnt, bu_n);
    GP_CHECK_HIP(rocprim::exclusive_scan(tmp, tb, cnt, bu_off, (int64_t)0, (size_t)(nb + 1), rocprim::plus<int64_t>(), s));
    GP_CHECK_LAUNCH();
    return GP_OK;
}

// pass 2: bu_row i32 [total], bu_mask u32 [total/32], wa_hi / wa_lo f16 [total/32 * 8 * 512] (only the fragments whose
// mask bit is set are defined -- and read)
extern "C" int gp_pool_cs_fill(const int32_t *nbr, const float *w, int64_t nv, int32_t k, const int64_t *bu_off, int64_t total_rows,
                               int32_t *bu_row, uint32_t *bu_mask, void *wa_hi, void *wa_lo, void *stream_) {
    GP_CHECK_ARG(nbr && w && bu_off && bu_row && bu_mask && wa_hi && wa_lo && nv > 0 && total_rows > 0 && total_rows % CS_KS == 0,
                 "gp_pool_cs_fill: bad argument");
    GP_CHECK_ARG((int64_t)CS_BR * k <= CS_MAXNK, "gp_pool_cs_fill: k=%d too large (128*k <= %d)", k, CS_MAXNK);
    int64_t nb = (nv + CS_BR - 1) / CS_BR;
    hipStream_t s = gp_stream(stream_);
    const size_t sm_max = (size_t)(CS_HS + CS_MAXID) * sizeof(int) + (size_t)CS_MAXNK * sizeof(unsigned short);
    size_t sm = (size_t)(CS_HS + CS_MAXID) * sizeof(int) + (size_t)CS_BR * k * sizeof(unsigned short);
    GP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(cs_fill_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_max));
    cs_fill_kernel<<<(unsigned)nb, 1024, sm, s>>>(nbr, w, nv, k, bu_off, bu_row, bu_mask, static_cast<_Float16 *>(wa_hi),
                                                  static_cast<_Float16 *>(wa_lo));
    GP_CHECK_LAUNCH();
    return GP_OK;
}

static int cs_launch_persist(const void *x_hi, const void *x_lo, int64_t ld_x, const int64_t *bu_off, const int32_t *bu_row,
                             const uint32_t *bu_mask, const void *wa_hi, const void *wa_lo, int64_t nv, void *y_hi, void *y_lo,
                             int64_t ld_y, float *y_f32, int64_t ld_yf, const float *out_scale, unsigned *queue, hipStream_t s) {
    using G = CsGeo<256, 3>;
    static bool attr_set = false;
    static int n_cu = 0;
    if (!attr_set) {
        GP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(cs_pool_persist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::SMEM + 16));
        int dev = 0;
        GP_CHECK_HIP(hipGetDevice(&dev));
        GP_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        attr_set = true;
    }
    const int64_t nb = (nv + CS_BR - 1) / CS_BR;
    const int64_t per_xcd = (nb * 2 + 7) / 8;
    const unsigned grid = (unsigned)((g_gp_knobs[10] > 0 ? g_gp_knobs[10] : (n_cu >= 8 ? n_cu / 8 : 1)) * 8);
    cs_pool_persist_kernel<<<grid, 512, G::SMEM + 16, s>>>(static_cast<const _Float16 *>(x_hi), static_cast<const _Float16 *>(x_lo), ld_x, bu_off, bu_row, bu_mask,
            static_cast<const _Float16 *>(wa_hi), static_cast<const _Float16 *>(wa_lo), nv, nb, static_cast<_Float16 *>(y_hi),
            static_cast<_Float16 *>(y_lo), ld_y, y_f32, ld_yf, per_xcd, g_gp_knobs[4], out_scale, queue, static_cast<uint64_t *>(g_gp_debug_ptr[0]));
    GP_CHECK_LAUNCH();
    return GP_OK;
}

// One application y = A x on pre-split operands (see gp_pool_mfma_apply for the operand conventions).  d must be 512.
extern "C" int gp_pool_cs_apply(const void *x_hi, const void *x_lo, int64_t ld_x, const int64_t *bu_off, const int32_t *bu_row,
                                const uint32_t *bu_mask, const void *wa_hi, const void *wa_lo, int64_t nv, int32_t d, void *y_hi,
                                void *y_lo, int64_t ld_y, float *y_f32, int64_t ld_yf, const float *out_scale, void *stream_) {
    GP_CHECK_ARG(x_hi && x_lo && bu_off && bu_row && bu_mask && wa_hi && wa_lo && nv > 0, "gp_pool_cs_apply: null/empty argument");
    GP_CHECK_ARG(d == CS_D, "gp_pool_cs_apply: d=%d (kernel specialised for %d columns)", d, CS_D);
    GP_CHECK_ARG((y_hi && y_lo) || y_f32, "gp_pool_cs_apply: no output requested");
    GP_CHECK_ARG(ld_x % 8 == 0 && (uintptr_t)x_hi % 16 == 0 && (uintptr_t)x_lo % 16 == 0, "gp_pool_cs_apply: x rows must be 16-byte aligned");
    GP_CHECK_ARG(!y_hi || (ld_y % 8 == 0 && (uintptr_t)y_hi % 16 == 0 && (uintptr_t)y_lo % 16 == 0 && y_hi != x_hi && y_lo != x_lo),
                 "gp_pool_cs_apply: y rows must be 16-byte aligned and must not alias x");
    GP_CHECK_ARG(!y_f32 || (ld_yf % 4 == 0 && (uintptr_t)y_f32 % 16 == 0), "gp_pool_cs_apply: fp32 output rows must be 16-byte aligned");
    hipStream_t s = gp_stream(stream_);
    const int64_t nb = (nv + CS_BR - 1) / CS_BR;
    uint64_t *stamp = static_cast<uint64_t *>(g_gp_debug_ptr[0]);
#define CS_ARGS static_cast<const _Float16 *>(x_hi), static_cast<const _Float16 *>(x_lo), ld_x, bu_off, bu_row, bu_mask,              \
                static_cast<const _Float16 *>(wa_hi), static_cast<const _Float16 *>(wa_lo), nv, nb, static_cast<_Float16 *>(y_hi),     \
                static_cast<_Float16 *>(y_lo), ld_y, y_f32, ld_yf, per_xcd, g_gp_knobs[4], out_scale, stamp
#define CS_LAUNCH(NC_, NST_)                                                                                                           \
    {                                                                                                                                  \
        using G = CsGeo<NC_, NST_>;                                                                                                    \
        static bool attr_set = false;                                                                                                  \
        if (!attr_set) {                                                                                                               \
            GP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(cs_pool_kernel<NC_, NST_, false>),                         \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::SMEM));                               \
            GP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(cs_pool_kernel<NC_, NST_, true>),                          \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::SMEM));                               \
            attr_set = true;                                                                                                           \
        }                                                                                                                              \
        const int64_t per_xcd = (nb * (CS_D / NC_) + 7) / 8;                                                                           \
        if (stamp) cs_pool_kernel<NC_, NST_, true><<<(unsigned)(per_xcd * 8), 512, G::SMEM, s>>>(CS_ARGS);                             \
        else cs_pool_kernel<NC_, NST_, false><<<(unsigned)(per_xcd * 8), 512, G::SMEM, s>>>(CS_ARGS);                                  \
    }
    if (g_gp_knobs[11] == 9) CS_LAUNCH(128, 4)              // tuning aid: column quarters, 4-slot ring
    else if (g_gp_knobs[11] == 10) CS_LAUNCH(128, 3)
    else if (g_gp_knobs[11] == 11) CS_LAUNCH(128, 5)
    else if (g_gp_knobs[11] == 12) CS_LAUNCH(128, 2)        // two workgroups per CU
    else if (g_gp_knobs[11] == 13 || g_gp_knobs[11] == 14) {   // tuning aid: the persistent form with static tile lists (14: claims from a debug queue)
        const int rc = cs_launch_persist(x_hi, x_lo, ld_x, bu_off, bu_row, bu_mask, wa_hi, wa_lo, nv, y_hi, y_lo, ld_y, y_f32, ld_yf, out_scale,
                                         g_gp_knobs[11] == 14 ? static_cast<unsigned *>(g_gp_debug_ptr[1]) : nullptr, s);
        if (rc != GP_OK) return rc;
    }
    else CS_LAUNCH(256, 3)
#undef CS_LAUNCH
#undef CS_ARGS
    GP_CHECK_LAUNCH();
    return GP_OK;
}

// The same application through the producer / consumer engine (cs_engine_kernel: one persistent workgroup per CU, tiles
// claimed in order per XCD).  queue: 9 x uint32 of device memory, ZERO at the first launch and left zero by every launch
// (the tile counters of the 8 XCD labels + a finished-workgroup counter); launches that share a queue must be stream-ordered.
// Results are bit-identical to gp_pool_cs_apply.
extern "C" int gp_pool_cs_apply_engine(const void *x_hi, const void *x_lo, int64_t ld_x, const int64_t *bu_off, const int32_t *bu_row,
                                       const uint32_t *bu_mask, const void *wa_hi, const void *wa_lo, int64_t nv, int32_t d,
                                       void *y_hi, void *y_lo, int64_t ld_y, float *y_f32, int64_t ld_yf, const float *out_scale,
                                       uint32_t *queue, void *stream_) {
    GP_CHECK_ARG(x_hi && x_lo && bu_off && bu_row && bu_mask && wa_hi && wa_lo && queue && nv > 0, "gp_pool_cs_apply_engine: null/empty argument");
    GP_CHECK_ARG(d == CS_D, "gp_pool_cs_apply_engine: d=%d (kernel specialised for %d columns)", d, CS_D);
    GP_CHECK_ARG((y_hi && y_lo) || y_f32, "gp_pool_cs_apply_engine: no output requested");
    GP_CHECK_ARG(ld_x % 8 == 0 && (uintptr_t)x_hi % 16 == 0 && (uintptr_t)x_lo % 16 == 0, "gp_pool_cs_apply_engine: x rows must be 16-byte aligned");
    GP_CHECK_ARG(!y_hi || (ld_y % 8 == 0 && (uintptr_t)y_hi % 16 == 0 && (uintptr_t)y_lo % 16 == 0 && y_hi != x_hi && y_lo != x_lo),
                 "gp_pool_cs_apply_engine: y rows must be 16-byte aligned and must not alias x");
    GP_CHECK_ARG(!y_f32 || (ld_yf % 4 == 0 && (uintptr_t)y_f32 % 16 == 0), "gp_pool_cs_apply_engine: fp32 output rows must be 16-byte aligned");
    hipStream_t s = gp_stream(stream_);
    static bool eattr = false;
    static int n_cu = 0;
    if (!eattr) {
        GP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(cs_engine_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)EG_SMEM));
        GP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(cs_engine_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)EG_SMEM));
        int dev = 0;
        GP_CHECK_HIP(hipGetDevice(&dev));
        GP_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        eattr = true;
    }
    const int64_t nb = (nv + CS_BR - 1) / CS_BR;
    GP_CHECK_ARG(nb * (CS_D / EG_NC) < (int64_t)1 << 31, "gp_pool_cs_apply_engine: too many tiles");
    uint64_t *stamp = static_cast<uint64_t *>(g_gp_debug_ptr[0]);
    const unsigned grid = (unsigned)((g_gp_knobs[10] > 0 ? g_gp_knobs[10] : (n_cu >= 8 ? n_cu / 8 : 1)) * 8);   // knob 10: workgroups per XCD label (tuning aid)
#define EG_ARGS static_cast<const _Float16 *>(x_hi), static_cast<const _Float16 *>(x_lo), ld_x, bu_off, bu_row, bu_mask,              \
                static_cast<const _Float16 *>(wa_hi), static_cast<const _Float16 *>(wa_lo), nv, nb, static_cast<_Float16 *>(y_hi),     \
                static_cast<_Float16 *>(y_lo), ld_y, y_f32, ld_yf, g_gp_knobs[4], out_scale, queue, stamp
    if (stamp) cs_engine_kernel<true><<<grid, EG_THREADS, EG_SMEM, s>>>(EG_ARGS);
    else cs_engine_kernel<false><<<grid, EG_THREADS, EG_SMEM, s>>>(EG_ARGS);
#undef EG_ARGS
    GP_CHECK_LAUNCH();
    return GP_OK;
}
