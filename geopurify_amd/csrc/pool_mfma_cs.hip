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
#include <rocprim/device/device_scan.hpp>

#include "gp_common.h"

extern int g_gp_knobs[16];
extern void *g_gp_debug_ptr[4];
extern size_t g_gp_debug_bytes[4];

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
constexpr float CS_WSCALE = GP_POOL_CS_WSCALE;    // weights (<= 1) are stored x 2^10 so that their f16 lo parts stay normal

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
// AUX: the cache-policy bits of the instruction (0 = default; 16 = sc1: served by L2, never by this CU's L1)
template <int AUX = 0>
__device__ __forceinline__ void cs_glds16(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)l, 16, 0, AUX);
}
// agent-visible accesses of the chained launch (cs_chain_kernel): sc1 stores are written through to memory, sc1 loads are never
// served by a CU's L1 (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility")
__device__ __forceinline__ uint32_t cs_ld_sc1(const uint32_t *p) {           // global_load_dword ... sc1 (the compiler counts it in vmcnt)
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void cs_st_sc1(uint32_t *p, uint32_t v) {         // global_store_dword ... sc1
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 16-byte written-through store.  There is no compiler-visible form (agent-scope atomic stores stop at 8 bytes, which move at
// 0.54-0.70 x the 16-byte rate), so it is inline asm -- and the hazard recogniser does not look inside inline asm: a store of more
// than 64 bits reads its data registers AFTER issue, gfx950 needs two wait states before a VALU may overwrite them.  The first
// build of this kernel had the next address computed into the first store's data registers in the very next instruction (28 % of
// the hi rows came out as garbage, the lo rows -- second store, nothing behind it -- fine): the s_nop belongs to the store.
template <typename V>
__device__ __forceinline__ void cs_st16_sc1(void *p, V v) {
    static_assert(sizeof(V) == 16, "16-byte store");
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

template <int OFF>
__device__ __forceinline__ void cs_tr(s16x4 &d, uint32_t addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void cs_rd128(f16x8 &d, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
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

// distinct neighbour ids of the rows of block b -> dense[0 .. U) (unsorted), through an LDS hash table of `cap` slots (a power of two
// >= 2048; `dense` holds cap entries too): a 2048-slot table first (unions of lattice neighbourhoods are a few hundred ids), all
// `cap` slots if that overflows.  Returns -1 if the union does not fit cap / 2 ids (the caller sized cap from the largest union, or
// runs the block again with the full-size table).
__device__ int cs_union(const int32_t *__restrict__ nbr, int n, int *keys, int *dense, int cap) {
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
                if (++probes > (hs < cap ? 256 : hs)) { s_over = 1; break; }   // (a table with a free slot ends the probe sequence)
            }
        }
        __syncthreads();
        const bool over = s_over || (hs < cap && s_new > hs / 2);          // block-uniform (the last table may fill up to its last slot)
        __syncthreads();
        if (!over) break;
        if (hs >= cap) return -1;
        hs = cap;
        shift = 32 - (31 - __clz(cap));
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

// pass 1: padded union size of every block (a multiple of 32 union rows, at least one step).  Launched twice: with a 2048-slot table
// (16 KiB of LDS: several workgroups per CU, and room beside a convolution tile) for every block -- a union above 1024 ids marks the
// block (bu_n = -1) -- then with the full-size table (128 KiB) for the marked blocks only.  stats[0] = the largest union (atomicMax).
__global__ void __launch_bounds__(1024)
cs_count_kernel(const int32_t *__restrict__ nbr, int64_t nv, int k, int rpb, int64_t *__restrict__ padded_cnt, int32_t *__restrict__ bu_n,
                int cap, int second, unsigned long long *__restrict__ stats) {
    extern __shared__ int s_mem[];
    int *keys = s_mem, *dense = s_mem + cap;
    const int64_t b = blockIdx.x, r0 = b * rpb;
    if (second && bu_n[b] >= 0) return;
    const int rows = (int)((nv - r0) < rpb ? (nv - r0) : rpb);
    const int U = cs_union(nbr + r0 * k, rows * k, keys, dense, cap);
    if (threadIdx.x == 0) {
        padded_cnt[b] = U < 0 ? 0 : (int64_t)((U + CS_KS - 1) / CS_KS) * CS_KS;
        bu_n[b] = U;
        if (U > 0 && stats) atomicMax(stats, (unsigned long long)U);
    }
}

// pass 2: the block's union rows in (first group, last group, group set, id) order, one bit per (step, group) that says
// whether the 16 x 32 weight fragment holds a non-zero, and the ELL weights scattered into MFMA fragment order:
// wa[(step * 8 + group) * 64 + lane][8], lane = (k >> 3) * 16 + m  (k = union row within the step, m = row within the group).
// MODE 1 (gp_pool_cs_structure): everything that needs the neighbour lists only -- union rows, masks, zeroed fragments --
// and, instead of the weights, dst[row * k + j] = the element index of (row, neighbour j) in the fragment arrays: whoever produces the
// weights later (gp_affinity_softmax_scatter) stores them straight into fragment order.
// MODE 2 (gp_pool_cs_structure_valid): union rows and masks as before; no fragment is touched; instead valid u32 [steps][128]
// (bit p of valid[step][row] = union row 32 step + p is a neighbour of that row): gp_affinity_cs_fragments computes every
// (row, union row) similarity of a fragment on the matrix cores, keeps the valid ones and writes WHOLE fragments.
constexpr int CS_FILL_WEIGHTS = 0, CS_FILL_DST = 1, CS_FILL_VALID = 2;
template <int MODE>
__global__ void __launch_bounds__(1024)
cs_fill_kernel(const int32_t *__restrict__ nbr, const float *__restrict__ w, int64_t nv, int k, int rpb, const int64_t *__restrict__ bu_off,
               int32_t *__restrict__ bu_row, uint32_t *__restrict__ bu_mask, _Float16 *__restrict__ wa_hi, _Float16 *__restrict__ wa_lo,
               int32_t *__restrict__ dst, uint32_t *__restrict__ valid, int cap) {
    constexpr bool WEIGHTS = MODE == CS_FILL_WEIGHTS;
    extern __shared__ int s_mem[];                           // A[cap] | B[cap] | npos u16 [br*k]   (cap: a power of two >= 2 x the largest union)
    int *A = s_mem, *B = s_mem + cap;
    unsigned short *npos = reinterpret_cast<unsigned short *>(s_mem + 2 * cap);
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x, r0 = b * rpb;
    const int rows = (int)((nv - r0) < rpb ? (nv - r0) : rpb);
    const int n = rows * k;
    const int32_t *nb = nbr + r0 * k;
    const int U = cs_union(nb, n, A, B, cap);                  // (>= 0: the host sized cap from the largest union of pass 1)
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
    if constexpr (MODE == CS_FILL_VALID) {
        // validity words: in LDS (A is free now: the order keys have been consumed) while the block's steps x 128 words fit it,
        // else in global memory (zeroed by this workgroup first: the barrier orders its zero stores before its atomics, both
        // through the same L2)
        const int nsteps = Up / CS_KS;
        const bool in_lds = nsteps * CS_BR <= cap;
        unsigned *sv = reinterpret_cast<unsigned *>(A);
        uint32_t *gv = valid + ks0 * CS_BR;
        __syncthreads();                                      // (every read of A above is done)
        for (int i = tid; i < nsteps * CS_BR; i += 1024) { if (in_lds) sv[i] = 0u; else gv[i] = 0u; }
        __syncthreads();
        for (int t = tid; t < n; t += 1024) {
            const int rl = t / k;
            const int id = nb[t];
            int lo = 0, hi = U - 1;
            while (lo < hi) { int mid = (lo + hi) >> 1; if (B[mid] < id) lo = mid + 1; else hi = mid; }
            const int p = npos[lo];
            if (in_lds) atomicOr(&sv[(p / CS_KS) * CS_BR + rl], 1u << (p % CS_KS));
            else atomicOr(&gv[(p / CS_KS) * CS_BR + rl], 1u << (p % CS_KS));
        }
        if (in_lds) {
            __syncthreads();
            for (int i = tid; i < nsteps * CS_BR; i += 1024) gv[i] = sv[i];
        }
        return;
    }
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
        if constexpr (WEIGHTS) {
            const float v = w[r0 * k + t] * GP_POOL_CS_WSCALE;
            const _Float16 h = (_Float16)v;
            wa_hi[idx] = h;
            wa_lo[idx] = (_Float16)(v - (float)h);
        } else {
            dst[r0 * k + t] = (int32_t)idx;                              // (the host checks total_rows * 128 < 2^31)
        }
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
// (TUNE: the ablation bits of `ablate_` are honoured; the production instantiation compiles them out)
// The chained launch (CHAIN, cs_chain_kernel): ALL T applications of a scene in one grid of T x tiles workgroups.  Workgroup
// blockIdx = 8 j + label is tile j mod per_xcd of application t = j / per_xcd in its label's range, so the hardware dispatcher
// is the work queue (in-order, compact front: what kept the XCD's L2 window together in every persistent experiment of rounds 2-3)
// and a workgroup only ever waits for workgroups with a smaller blockIdx.  Application t reads plane set (t even ? A : B) and writes
// the other one (the last one writes fp32); a row block's tile of application t may start once every row block in its dependency
// list has PUBLISHED application t - 1: flags[half][block] = base + applications done, one sc1 store by one lane after every wave
// of the workgroup has drained its (sc1, written-through) output stores behind a barrier; the consumer polls the flags with sc1
// loads and gathers the rows with sc1 LDS-DMA (the XCDs' L2s are not coherent with each other, a CU's L1 is never refreshed).
// The lists are symmetric (d in list(b) <=> b in list(d), built by gp_pool_cs_deps), so "my inputs are published" also means
// "nobody still reads the rows I am about to overwrite" and two plane sets are enough.  Odd labels walk their range backwards:
// the two ends of neighbouring ranges then finish an application together and no range's first tile waits for another range's
// last one (CPU study on the S scene: 10 of 16 191 edges with less than a tenth of a sweep of slack, 136 walking all forwards).
// A wait is bounded (2 s of the constant 100 MHz clock): on expiry the workgroup sets abort and returns, every later poll sees
// abort and returns -- the grid always drains; the host reads the word (gp_pool_cs_apply_chain's contract).
struct CsChain {
    const _Float16 *a_hi, *a_lo;       // plane set A (application 0's input; rewritten by applications 1, 3, ...)
    _Float16 *b_hi, *b_lo;             // plane set B
    int64_t ld;                        // row pitch of both sets (elements)
    uint32_t *flags;                   // [32 header words: word 0 = abort] [2 halves][nblocks]
    const int32_t *dep;                // [nblocks][64]: word 0 = n, words 1 .. min(n, 63) = row blocks (n > 63: wait for every block)
    int32_t T;                         // applications (>= 2)
    uint32_t base;                     // flags epoch: a published application t reads base + t + 1
    int32_t half_sel = -1;             // (not chained) 0 / 1: the grid covers this 256-column half only (cs_pool_half_kernel)
};
constexpr int CS_DEP_CAP = 64;
constexpr int CS_FLAG_HDR = 32;

template <bool STAMP, bool TUNE, bool CHAIN = false>
__device__ __forceinline__ void
cs_pool_body(const _Float16 *__restrict__ x_hi_, const _Float16 *__restrict__ x_lo_, int64_t ld_x_,
               const int64_t *__restrict__ bu_off, const int32_t *__restrict__ bu_row, const uint32_t *__restrict__ bu_mask,
               const _Float16 *__restrict__ wa_hi, const _Float16 *__restrict__ wa_lo, int64_t nv, int64_t nblocks,
               _Float16 *__restrict__ y_hi_, _Float16 *__restrict__ y_lo_, int64_t ld_y_, float *__restrict__ y_f32_, int64_t ld_yf,
               int64_t per_xcd, int ablate_, const float *__restrict__ out_scale, uint64_t *__restrict__ stamp, int rpb,
               const CsChain &ch) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    constexpr int AUX = CHAIN ? 16 : 0;
    const int ablate = TUNE ? ablate_ : 0;
    uint64_t st_t0 = 0, st_r0 = 0, st_pro = 0, st_work = 0, st_wait = 0, st_issue = 0;
    if constexpr (STAMP) { st_t0 = cs_now(); st_r0 = cs_real(); }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-contiguous order: blocks b, b + 8, ... share an XCD (observed round-robin placement); each XCD walks a contiguous range of
    // tiles, the two column halves of a row block back to back (they share the weight fragments in its L2) and Morton-adjacent row
    // blocks side by side (their halo rows meet in its L2).  Measured and left out (round 4, profiles/r04_pool_block_height.log):
    // starting the row blocks longest-first, dealt round-robin over the XCDs, 0.279 instead of 0.223 ms per application -- the
    // drain of the last round is worth less than the L2 sharing between neighbouring tiles.
    int64_t lb;
    int app = 0;
    bool app_last = true;
    const _Float16 *x_hi = x_hi_, *x_lo = x_lo_;
    _Float16 *y_hi = y_hi_, *y_lo = y_lo_;
    float *y_f32 = y_f32_;
    int64_t ld_x = ld_x_, ld_y = ld_y_;
    if constexpr (CHAIN) {
        const int64_t j = blockIdx.x >> 3;
        const int label = blockIdx.x & 7;
        app = (int)(j / per_xcd);
        const int64_t r = j - (int64_t)app * per_xcd;
        lb = (int64_t)label * per_xcd + ((label & 1) ? per_xcd - 1 - r : r);
        app_last = app == ch.T - 1;
        const bool even = (app & 1) == 0;
        x_hi = even ? ch.a_hi : ch.b_hi;
        x_lo = even ? ch.a_lo : ch.b_lo;
        y_hi = app_last ? nullptr : even ? ch.b_hi : const_cast<_Float16 *>(ch.a_hi);
        y_lo = app_last ? nullptr : even ? ch.b_lo : const_cast<_Float16 *>(ch.a_lo);
        y_f32 = app_last ? y_f32_ : nullptr;
        ld_x = ld_y = ch.ld;
    } else {
        lb = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
        if (ch.half_sel >= 0) lb = 2 * lb + ch.half_sel;         // (per_xcd then counts row blocks, not tiles)
    }
    const int64_t b = lb >> 1;
    const int col0 = (int)(lb & 1) * CS_NC;
    if (b >= nblocks) return;
    uint32_t *my_flag = nullptr;
    if constexpr (CHAIN) {
        uint32_t *fl = ch.flags + CS_FLAG_HDR + (lb & 1) * nblocks;
        my_flag = fl + b;
        if (app > 0) {
            // every wave polls for itself (no barrier in front of the first DMA): lane 0 watches abort, lanes 1 .. n the row blocks
            const uint32_t want = ch.base + (uint32_t)app;
            const int32_t dv = ch.dep[b * CS_DEP_CAP + lane];
            const int n = __builtin_amdgcn_readfirstlane(dv);
            const uint64_t w0 = cs_real();
            for (int64_t i0 = 0;;) {
                uint32_t v = want, ab = 0;
                if (n < CS_DEP_CAP) {
                    if (lane >= 1 && lane <= n) v = cs_ld_sc1(fl + dv);
                } else if (i0 + lane < nblocks) {
                    v = cs_ld_sc1(fl + i0 + lane);
                }
                if (lane == 0) ab = cs_ld_sc1(ch.flags);
                if (__builtin_amdgcn_readfirstlane(ab) != 0) return;                       // (all eight waves see it: the workgroup leaves)
                if (__ballot((int32_t)(v - want) < 0) == 0ull) {
                    if (n < CS_DEP_CAP || (i0 += 64) >= nblocks) break;
                    continue;
                }
                if (cs_real() - w0 > 200000000ull) {                                       // 2 s: a dependency that never comes
                    if (lane == 0) cs_st_sc1(ch.flags, 1u);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    return;
                }
                __builtin_amdgcn_s_sleep(8);
            }
            if constexpr (STAMP) st_issue = cs_now() - st_t0;                              // (slot 5 of the stamps: the dependency wait)
        }
    }
    const int64_t ub0 = bu_off[b];
    const int n = (int)((bu_off[b + 1] - ub0) / CS_KS);                            // steps (>= 1)
    const int64_t ks0 = ub0 / CS_KS;

    // ---- DMA roles
    const int du = lane >> 5, dc = lane & 31;
    const int swz = (wv >> 1) & 1;
    const int t0 = du | (swz << 2), t1 = (2 + du) | (swz << 2);                     // t(row) of rows 4 wv + du, 4 wv + 2 + du
    const int64_t dsrc0 = col0 + ((dc ^ (2 * t0)) * 8);
    const int64_t dsrc1 = col0 + ((dc ^ (2 * t1)) * 8);
    const int32_t *idg = bu_row + ub0 + 4 * wv;                                    // this wave's row ids, step 0
    const uint32_t *mkg = bu_mask + ks0;
    const _Float16 *wah = wa_hi + (ks0 * CS_NG + wv) * 512;
    const _Float16 *wal = wa_lo + (ks0 * CS_NG + wv) * 512;
    auto issue = [&](i32x4 id, unsigned mk, int k, int slot) {
        unsigned char *dst = smem_raw + slot * CS_STAGE;
        const int ida = du ? id.y : id.x, idb = du ? id.w : id.z;
        // Every stage is 4 row pieces + 2 fragment pieces IF the wave's group has a fragment in it, in every tuning mask: the hand-over's
        // hand-counted `vmcnt` (handover_behind) means "the older stage has landed" only then.  Tuning bit 1 (no row gather) and bit 3 (no weight
        // fragments) therefore do not drop instructions, they make all lanes fetch ONE hot 16-byte piece instead (round 3
        // dropped them: the ceiling launches of bench.py handed a slot over with half of the older stage in flight, and the
        // 64-row kernel of pool_mfma.hip, whose row ids ride in the ring, read stale ids and faulted).
        const bool hot_x = (ablate & 2) != 0, hot_w = (ablate & 8) != 0;
        const int64_t s0 = hot_x ? 0 : (int64_t)ida * ld_x + dsrc0, s1 = hot_x ? 0 : (int64_t)idb * ld_x + dsrc1;
        cs_glds16<AUX>(x_hi + s0, dst + (4 * wv) * CS_RB);
        cs_glds16<AUX>(x_lo + s0, dst + CS_PLANE + (4 * wv) * CS_RB);
        cs_glds16<AUX>(x_hi + s1, dst + (4 * wv) * CS_RB + 1024);
        // tuning bit 7: what would an 8-BIT lo plane buy?  Half of the lo rows come from the hot piece (the bytes of the gather as they
        // would be; one instruction more than the real thing would issue), and the epilogue stores half of its lo bytes -- a price, not
        // a result
        cs_glds16<AUX>(x_lo + ((ablate & 128) ? 0 : s1), dst + CS_PLANE + (4 * wv) * CS_RB + 1024);
        // an empty fragment (half of them on the S scene) is never read and -- round 5 -- never fetched: the wave issues 4 instead of 6
        // instructions for that stage, and the hand-over that lets this stage stay in flight counts accordingly (cs_stage_dma below).
        // (Rounds 3-5a fetched one hot piece instead, to keep the count fixed: a third of the loop's LDS-DMA instructions were dummies.)
        if ((mk >> wv) & 1u) {
            const int lo = hot_w ? 0 : lane * 8;
            cs_glds16(wah + (int64_t)k * (CS_NG * 512) + lo, dst + CS_OFF_W + wv * 1024);
            cs_glds16(wal + (int64_t)k * (CS_NG * 512) + lo, dst + CS_OFF_W + CS_WPL + wv * 1024);
        }
    };
    // hand-over with the stage of mask `mk` (the youngest, issued by THIS wave) allowed to stay in flight: 4 row pieces + its fragment's 2
    auto handover_behind = [&](unsigned mk) {
        if ((mk >> wv) & 1u) cs_handover<CS_DMA>(); else cs_handover<CS_DMA - 2>();
    };
    auto load_ids = [&](int k) { return *reinterpret_cast<const i32x4 *>(idg + (int64_t)k * CS_KS); };

    // ---- read roles
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem_raw;
    uint32_t addr[2];
    {
        const uint32_t rowb = (uint32_t)(8 * g + q) * CS_RB + (uint32_t)(wv * CS_WC * 2) + (uint32_t)(p * 8);
        const uint32_t t = (uint32_t)(q | ((g & 1) << 2));
        addr[0] = lds0 + (rowb ^ (t << 5));
        addr[1] = lds0 + ((rowb + 32u) ^ (t << 5));
    }
    const uint32_t addr_w = lds0 + CS_OFF_W + lane * 16;

    f32x4 acc[CS_NG * 2];
#pragma unroll
    for (int i = 0; i < CS_NG * 2; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: stages 0 and 1 in flight (a one-step block stages its only step twice)
    unsigned mA, mB, mC;
    i32x4 idv;
    {
        const int k1 = n > 1 ? 1 : 0, k2 = n > 2 ? 2 : n - 1;
        const i32x4 i0 = load_ids(0), i1 = load_ids(k1);
        mA = mkg[0];
        mB = mkg[k1];
        issue(i0, mA, 0, 0);
        issue(i1, mB, k1, 1);
        idv = load_ids(k2);
        mC = mkg[k2];
        asm volatile("" ::"s"(idv.x), "s"(idv.y), "s"(idv.z), "s"(idv.w), "s"(mC));   // (waited for here, not inside the loop)
        handover_behind(mB);                                   // stage 0 has landed; stage 1 may be in flight
    }
    if constexpr (STAMP) st_pro = cs_now();
    // Software pipeline (the fragment reads of all eight waves leave the barrier together and take ~500 cycles to come back;
    // an MFMA batch in front of each wait hides part of that):
    //   step s:  reads {staged rows, weight fragments of groups 0-3} of stage s      | waves 0-3: DMA of stage s + 2
    //            MFMA batch "groups 4-7" of stage s - 1 (fragments read in step s - 1, rows kept in bhp / blp)
    //            wait; reads {weight fragments of groups 4-7} of stage s; next step's scalars (s_load)
    //            MFMA batch "groups 0-3" of stage s                                  | waves 4-7: DMA of stage s + 2
    //            wait (every LDS read of stage s has landed in registers); hand-over
    // Waves 0-3 issue their DMA first and waves 4-7 last, so that the two waves of a SIMD alternate between DMA issue
    // (which stalls on the memory pipeline's back-pressure) and matrix work.
    // Measured and left out (profiles/r03_pool_cs_variants.log): releasing a slot as soon as its operands are in registers
    // (a second barrier per step, the stage THREE steps ahead issued into it: 0.259 instead of 0.232 ms -- the launch is bound by
    // the bytes that reach HBM, 1.28 GB at 5.5 TB/s, not by the bytes in flight).
    s16x4 fb[2][2][2];
    f16x8 ah0[4], al0[4], ah1[4], al1[4], bhp[2], blp[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { ah0[i] = al0[i] = ah1[i] = al1[i] = f16x8{0, 0, 0, 0, 0, 0, 0, 0}; }
#pragma unroll
    for (int u = 0; u < 2; ++u) { bhp[u] = blp[u] = f16x8{0, 0, 0, 0, 0, 0, 0, 0}; }
    unsigned mP = 0;                                         // fragment mask of the previous step (its groups 4-7 are pending)
    auto mfma_hi = [&](unsigned m) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
            if (__builtin_expect((m >> (4 + mt)) & 1u, 1)) {
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[(4 + mt) * 2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1[mt], bhp[u], acc[(4 + mt) * 2 + u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[(4 + mt) * 2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1[mt], blp[u], acc[(4 + mt) * 2 + u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[(4 + mt) * 2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1[mt], bhp[u], acc[(4 + mt) * 2 + u], 0, 0, 0);
            }
    };
    const bool late = !(ablate & 32) && wv >= 4;            // tuning aid: bit 5 makes every wave issue first
    const bool do_reads = !(ablate & 1);                     // tuning aid: bit 0 skips reads + MFMAs
    for (int s0 = 0; s0 < n; s0 += CS_NST) {
#pragma unroll
        for (int J = 0; J < CS_NST; ++J) {
            const int s = s0 + J;
            if (s < n) {
                uint64_t st_a = 0, st_b = 0;
                if constexpr (STAMP) st_a = cs_now();
                const uint32_t a0 = addr[0] + J * CS_STAGE, a1 = addr[1] + J * CS_STAGE, aw = addr_w + J * CS_STAGE;
                const unsigned m = mA;
                if (do_reads) {
                    // staged rows: fb[col block][plane][rows 8g+q | 8g+q+4]; weight fragments of groups 0-3
                    cs_tr<0>(fb[0][0][0], a0);
                    cs_tr<4 * CS_RB>(fb[0][0][1], a0);
                    cs_tr<CS_PLANE>(fb[0][1][0], a0);
                    cs_tr<CS_PLANE + 4 * CS_RB>(fb[0][1][1], a0);
                    cs_tr<0>(fb[1][0][0], a1);
                    cs_tr<4 * CS_RB>(fb[1][0][1], a1);
                    cs_tr<CS_PLANE>(fb[1][1][0], a1);
                    cs_tr<CS_PLANE + 4 * CS_RB>(fb[1][1][1], a1);
                    if (m & 1u) { cs_rd128<0 * 1024>(ah0[0], aw); cs_rd128<CS_WPL + 0 * 1024>(al0[0], aw); }
                    if (m & 2u) { cs_rd128<1 * 1024>(ah0[1], aw); cs_rd128<CS_WPL + 1 * 1024>(al0[1], aw); }
                    if (m & 4u) { cs_rd128<2 * 1024>(ah0[2], aw); cs_rd128<CS_WPL + 2 * 1024>(al0[2], aw); }
                    if (m & 8u) { cs_rd128<3 * 1024>(ah0[3], aw); cs_rd128<CS_WPL + 3 * 1024>(al0[3], aw); }
                }
                if (!late && s + 2 < n) issue(idv, mC, s + 2, (J + 2) % CS_NST);
                if constexpr (STAMP) if (ablate & 64) st_issue += cs_now() - st_a;
                if (do_reads) {
                    mfma_hi(mP);                            // groups 4-7 of the previous step
                    cs_wait_b(fb);
                    cs_wait_a(ah0, al0);
                }
                // scalars of the stage issued in the next step (clamped: never past the block's padded union); they are
                // waited for right before the barrier, a whole MFMA batch later
                const int kn = s + 3 < n ? s + 3 : n - 1;
                const i32x4 idn = load_ids(kn);
                const unsigned mN = mkg[kn];
                if (do_reads) {
                    if (m & 16u) { cs_rd128<4 * 1024>(ah1[0], aw); cs_rd128<CS_WPL + 4 * 1024>(al1[0], aw); }
                    if (m & 32u) { cs_rd128<5 * 1024>(ah1[1], aw); cs_rd128<CS_WPL + 5 * 1024>(al1[1], aw); }
                    if (m & 64u) { cs_rd128<6 * 1024>(ah1[2], aw); cs_rd128<CS_WPL + 6 * 1024>(al1[2], aw); }
                    if (m & 128u) { cs_rd128<7 * 1024>(ah1[3], aw); cs_rd128<CS_WPL + 7 * 1024>(al1[3], aw); }
#pragma unroll
                    for (int u = 0; u < 2; ++u) { bhp[u] = cs_cat(fb[u][0][0], fb[u][0][1]); blp[u] = cs_cat(fb[u][1][0], fb[u][1][1]); }
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        if (__builtin_expect((m >> mt) & 1u, 1)) {
#pragma unroll
                            for (int u = 0; u < 2; ++u) acc[mt * 2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0[mt], bhp[u], acc[mt * 2 + u], 0, 0, 0);
#pragma unroll
                            for (int u = 0; u < 2; ++u) acc[mt * 2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0[mt], blp[u], acc[mt * 2 + u], 0, 0, 0);
#pragma unroll
                            for (int u = 0; u < 2; ++u) acc[mt * 2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0[mt], bhp[u], acc[mt * 2 + u], 0, 0, 0);
                        }
                }
                if (late && s + 2 < n) issue(idv, mC, s + 2, (J + 2) % CS_NST);
                // every LDS read of this stage is in registers before the barrier lets its slot be refilled, and the scalar
                // loads are waited for HERE, so that no compiler-placed lgkmcnt(0) sits inside the next step
                cs_wait_a(ah1, al1);
                asm volatile("" ::"s"(idn.x), "s"(idn.y), "s"(idn.z), "s"(idn.w), "s"(mN));
                if constexpr (STAMP) { st_b = cs_now(); st_work += st_b - st_a; }
                if (s + 2 < n) handover_behind(mC); else cs_handover<0>();      // (mC: the stage issued in this step)
                if constexpr (STAMP) st_wait += cs_now() - st_b;
                mP = m; mA = mB; mB = mC; mC = mN; idv = idn;
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
    float *stg = reinterpret_cast<float *>(smem_raw) + wv * (CS_BR * CS_EP);
    const int fl = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < CS_NG; ++mt)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) stg[(mt * 16 + fq * 4 + r) * CS_EP + cb * 16 + fl] = acc[mt * 2 + cb][r] * inv;
    gp_wave_sync();
    const int64_t row0 = b * rpb;
    const int colw = col0 + wv * CS_WC;
    // lane -> 8 consecutive columns of row it * 16 + (lane >> 2): every store instruction writes 16 rows x 64 bytes
    const int er = lane >> 2, ec = (lane & 3) * 8;
    const float so = (y_f32 && out_scale) ? out_scale[0] : 1.f;
#pragma unroll
    for (int h8 = 0; h8 < 2; ++h8) {                        // two halves of 64 rows: bounds the live registers
        float4 v[4][2];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const float *sp = stg + (h8 * 64 + it * 16 + er) * CS_EP + ec;
            v[it][0] = *reinterpret_cast<const float4 *>(sp);
            v[it][1] = *reinterpret_cast<const float4 *>(sp + 4);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int lrow = h8 * 64 + it * 16 + er;       // (rows rpb .. 127 of a block do not exist: their weights are zero)
            const int64_t grow = row0 + lrow;
            if (lrow < rpb && grow < nv && !(ablate & 16)) {             // tuning aid: bit 4 skips the output stores
                const float xv[8] = {v[it][0].x, v[it][0].y, v[it][0].z, v[it][0].w, v[it][1].x, v[it][1].y, v[it][1].z, v[it][1].w};
                if (y_hi) {
                    f16x8 h, l;
#pragma unroll
                    for (int i = 0; i < 8; ++i) { h[i] = (_Float16)xv[i]; l[i] = (_Float16)(xv[i] - (float)h[i]); }
                    if constexpr (CHAIN) {                  // written through: another XCD gathers these rows in the same launch
                        cs_st16_sc1(y_hi + grow * ld_y + colw + ec, h);
                        cs_st16_sc1(y_lo + grow * ld_y + colw + ec, l);
                    } else {
                        *reinterpret_cast<f16x8 *>(y_hi + grow * ld_y + colw + ec) = h;
                        if (ablate & 128) {
                            typedef _Float16 f16x4_ __attribute__((ext_vector_type(4)));
                            *reinterpret_cast<f16x4_ *>(y_lo + grow * ld_y + colw + (ec >> 1)) = f16x4_{l[0], l[1], l[2], l[3]};   // 8 of the 16 bytes, packed
                        } else {
                            *reinterpret_cast<f16x8 *>(y_lo + grow * ld_y + colw + ec) = l;
                        }
                    }
                }
                if (y_f32) {
                    float *yp = y_f32 + grow * ld_yf + colw + ec;
                    *reinterpret_cast<float4 *>(yp) = make_float4(xv[0] * so, xv[1] * so, xv[2] * so, xv[3] * so);
                    *reinterpret_cast<float4 *>(yp + 4) = make_float4(xv[4] * so, xv[5] * so, xv[6] * so, xv[7] * so);
                }
            }
        }
    }
    if constexpr (CHAIN) {
        // publish: every wave's stores have left (vmcnt(0)), the workgroup meets, ONE lane stores the flag
        if (!app_last) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) cs_st_sc1(my_flag, ch.base + (uint32_t)app + 1u);
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
            o[6] = t3 - st_e0; o[7] = t3 - st_t0; o[8] = (uint64_t)n; o[9] = xcc | ((uint64_t)app << 8);
        }
    }
}

#define CS_POOL_PARAMS const _Float16 *__restrict__ x_hi, const _Float16 *__restrict__ x_lo, int64_t ld_x, const int64_t *__restrict__ bu_off,          \
                       const int32_t *__restrict__ bu_row, const uint32_t *__restrict__ bu_mask, const _Float16 *__restrict__ wa_hi,                    \
                       const _Float16 *__restrict__ wa_lo, int64_t nv, int64_t nblocks, _Float16 *__restrict__ y_hi, _Float16 *__restrict__ y_lo,       \
                       int64_t ld_y, float *__restrict__ y_f32, int64_t ld_yf, int64_t per_xcd, int ablate, const float *__restrict__ out_scale,       \
                       uint64_t *__restrict__ stamp, int rpb
#define CS_POOL_FWD x_hi, x_lo, ld_x, bu_off, bu_row, bu_mask, wa_hi, wa_lo, nv, nblocks, y_hi, y_lo, ld_y, y_f32, ld_yf, per_xcd, ablate, out_scale, stamp, rpb
// the product kernel (STAMP = false: no tuning bits, no stamps) and its stamped instantiation
template <bool STAMP>
__global__ void __launch_bounds__(512, 2) cs_pool_kernel(CS_POOL_PARAMS) { cs_pool_body<STAMP, STAMP>(CS_POOL_FWD, CsChain{}); }
// the same body with the tuning bits live, under its own name: launches with parts of the kernel switched off (bench.py's
// gather + store ceiling, scripts/bench_pool.py ablations) do not mix into the product kernel's rows of a kernel trace
__global__ void __launch_bounds__(512, 2) cs_pool_tuning_kernel(CS_POOL_PARAMS) { cs_pool_body<false, true>(CS_POOL_FWD, CsChain{}); }
// one column half only (gp_pool_cs_apply_half: two independent chains of launches, one per half, on two streams)
__global__ void __launch_bounds__(512, 2) cs_pool_half_kernel(CS_POOL_PARAMS, int half) {
    CsChain ch{};
    ch.half_sel = half;
    cs_pool_body<false, false>(CS_POOL_FWD, ch);
}
#undef CS_POOL_PARAMS
#undef CS_POOL_FWD
// all T applications in one launch (see CsChain above)
template <bool STAMP>
__global__ void __launch_bounds__(512, 2)
cs_chain_kernel(CsChain ch, const int64_t *__restrict__ bu_off, const int32_t *__restrict__ bu_row, const uint32_t *__restrict__ bu_mask,
                const _Float16 *__restrict__ wa_hi, const _Float16 *__restrict__ wa_lo, int64_t nv, int64_t nblocks, float *__restrict__ y_f32,
                int64_t ld_yf, int64_t per_xcd, const float *__restrict__ out_scale, uint64_t *__restrict__ stamp, int rpb) {
    cs_pool_body<STAMP, false, true>(nullptr, nullptr, 0, bu_off, bu_row, bu_mask, wa_hi, wa_lo, nv, nblocks, nullptr, nullptr, 0, y_f32, ld_yf,
                                     per_xcd, 0, out_scale, stamp, rpb, ch);
}

// ------------------------------------------------------------------------------------------------ affinity -> fragments
// Row 11 on the matrix cores, fused with the operator fill (models/affinity_module.py:1559-1572: cosine similarity of the unit
// embeddings, x sharpen, softmax over a row's K neighbours).  One 512-thread workgroup per row block; wave g owns the block's 16-row
// group g.  The block's union rows are staged 32 at a time (the pooling kernel's steps) as f16 hi / lo planes of the embeddings x 2^10;
// for a (step, group) fragment that is not empty the wave computes ALL 32 x 16 similarities of the step's union rows with its rows
// (2 tiles x 4 K steps x {hi hi, hi lo, lo hi} v_mfma_f32_16x16x32_f16, its own rows' fragments held in registers), keeps the
// entries whose validity bit is set (gp_pool_cs_structure_valid) in a per-row list in LDS (slot = entries of that row seen so far),
// runs the softmax of each row over its list (the wave form's arithmetic: true row maximum, expf, fp32 sum) and writes every
// non-empty fragment WHOLE -- zeros where there is no neighbour -- in the order the pooling kernel reads (weights x 2^10, hi + lo).
// Against affinity_block_kernel: 0.33 GB gathered instead of 0.95, no dst table, no zeroing pass, fragments in 8-byte pieces instead
// of 2-byte scatters, and the 12.3 M dot products are 15 us of matrix work.
constexpr int AF_D = 128;                         // embedding width
constexpr int AF_RB = AF_D * 2;                   // bytes per staged row and plane
constexpr int AF_PLANE = CS_KS * AF_RB;           // 8 KiB
constexpr int AF_STAGE = 2 * AF_PLANE;            // 16 KiB
constexpr int AF_NST = 4;                         // ring depth (3, 4 and 6 stages measure the same: the ring is not what a step waits for)
constexpr int AF_KEEP = 24;                       // steps whose validity words stay in LDS for the fragment pass (longer blocks re-load them)
constexpr int AF_KMAX = 96;                       // neighbours per row (the list pitch is AF_KMAX + 1 floats)
constexpr int AF_PITCH = AF_KMAX + 1;
constexpr int AF_OFF_V = AF_NST * AF_STAGE;       // the steps' validity words: [stage][wave][64] u32
constexpr int AF_OFF_K = AF_OFF_V + AF_NST * CS_NW * 256;                  // kept validity words: [AF_KEEP][128] u32
constexpr int AF_OFF_P = AF_OFF_K + AF_KEEP * CS_BR * 4;
constexpr size_t AF_SMEM = (size_t)AF_OFF_P + (size_t)CS_BR * AF_PITCH * sizeof(float);
constexpr float AF_ESCALE = 1024.f;               // the embedding planes carry e x 2^10 (lo halves stay normal numbers)
constexpr int AF_DMA = 3;                         // LDS-DMA instructions per wave and stage: rows hi, rows lo, validity words

// (TUNE: the bits of knob 8 are honoured -- 1 no fragment reads / MFMA, 2 no list stores, 4 no softmax, 8 no fragment pass, 16 no
// LDS-DMA; the product instantiation compiles them out)
template <bool TUNE>
__global__ void __launch_bounds__(512, 2)
affinity_cs_kernel(const _Float16 *__restrict__ e_hi, const _Float16 *__restrict__ e_lo, int64_t nv, float sharpen,
                   const int64_t *__restrict__ bu_off, const int32_t *__restrict__ bu_row, const uint32_t *__restrict__ bu_mask,
                   const uint32_t *__restrict__ bu_valid, int64_t nblocks, int rpb, int64_t per_xcd,
                   _Float16 *__restrict__ wa_hi, _Float16 *__restrict__ wa_lo, int tune_) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tune = TUNE ? tune_ : 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);                     // wave = 16-row group
    const int64_t b = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);  // XCD-contiguous: neighbouring blocks share halo rows in L2
    if (b >= nblocks) return;
    const int64_t ub0 = bu_off[b];
    const int n = (int)((bu_off[b + 1] - ub0) / CS_KS);
    const int64_t ks0 = ub0 / CS_KS;
    const int m = lane & 15, q = lane >> 4;
    float *plist = reinterpret_cast<float *>(smem_raw + AF_OFF_P) + (size_t)(g * 16 + m) * AF_PITCH;   // this lane's row's list
    const uint32_t plist_a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem_raw + AF_OFF_P +
                             (uint32_t)((g * 16 + m) * AF_PITCH * 4);
    // ---- this wave's own rows as the B operand (row m of the group, 8 consecutive channels per lane and K step)
    f16x8 bh[4], bl[4];
    {
        int64_t r = b * rpb + g * 16 + m;
        r = r < nv ? r : nv - 1;                                               // (rows past the end have no valid bit)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bh[ks] = *reinterpret_cast<const f16x8 *>(e_hi + r * AF_D + ks * 32 + q * 8);
            bl[ks] = *reinterpret_cast<const f16x8 *>(e_lo + r * AF_D + ks * 32 + q * 8);
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(bh[0]), "+v"(bh[1]), "+v"(bh[2]), "+v"(bh[3]), "+v"(bl[0]), "+v"(bl[1]), "+v"(bl[2]), "+v"(bl[3]));
    }
    // ---- staging: wave g stages union rows 4 g .. 4 g + 3 of a step: one 1-KiB instruction per plane (lane -> row lane / 16, 16-byte
    // piece lane % 16) and the step's validity words of ITS rows (64 words from its group's first: the 48 behind its 16 are other
    // groups' -- fetched, never read; the array is padded by 64 words).  Physical piece p of row r holds logical piece p ^ (r & 15):
    // the fragment reads of 16 consecutive rows at one logical piece then touch 16 different piece positions (a row is 256 bytes =
    // all 64 banks).  The loop issues nothing but these LDS-DMA instructions, so `s_waitcnt vmcnt(AF_DMA)` means "the older stage
    // has landed"; row ids and fragment masks come through the scalar cache.
    // MFMA tile t takes the step's union rows 8 a + 4 t + c (a, c = 0 .. 3) as its rows 4 a + c: a lane's C registers of the two tiles
    // are then 8 CONSECUTIVE union rows (8 q .. 8 q + 7) for its row m -- exactly one 16-byte piece of the weight fragment the pooling
    // kernel reads (fragment lane q * 16 + m = this lane), so the fragments leave in 1-KiB stores.  Swizzle key of a staged row:
    // 4 (row / 8) + row % 4 -- the 16 rows of either tile have 16 different keys.
    const int srow_l = lane >> 4;
    const int srow_k = (((4 * g + srow_l) >> 3) << 2) | ((4 * g + srow_l) & 3);
    const int spiece = (lane & 15) ^ srow_k;
    const int32_t *idg = bu_row + ub0 + 4 * g;
    const uint32_t *vgl = bu_valid + ks0 * CS_BR + g * 16 + lane;
    auto load_ids = [&](int k) { return *reinterpret_cast<const i32x4 *>(idg + (int64_t)k * CS_KS); };   // (wave-uniform address: a scalar load)
    auto issue = [&](i32x4 id4, int k, int slot) {
        if (tune & 16) return;
        const int64_t id = srow_l == 0 ? id4.x : srow_l == 1 ? id4.y : srow_l == 2 ? id4.z : id4.w;
        unsigned char *dst = smem_raw + slot * AF_STAGE + (4 * g) * AF_RB;
        cs_glds16<0>(e_hi + id * AF_D + spiece * 8, dst);
        cs_glds16<0>(e_lo + id * AF_D + spiece * 8, dst + AF_PLANE);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vgl + (int64_t)k * CS_BR),
                                         (__attribute__((address_space(3))) void *)(smem_raw + AF_OFF_V + (slot * CS_NW + g) * 256), 4, 0, 0);
    };
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem_raw;
    // fragment read of tile t, K step ks: staged row 8 (m / 4) + 4 t + m % 4 (key m), logical piece 4 ks + q
    uint32_t rd[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int i = 8 * (m >> 2) + 4 * t + (m & 3);
            rd[t][ks] = lds0 + (uint32_t)(i * AF_RB + (((4 * ks + q) ^ m) << 4));
        }
    const uint32_t rdv = lds0 + AF_OFF_V + (uint32_t)(g * 256 + m * 4);
    const uint32_t *mk = bu_mask + ks0;
#pragma unroll
    for (int j = 0; j < AF_NST - 1; ++j)
        if (j < n) issue(load_ids(j), j, j);
    // the scalars of a step (row ids of the stage it issues, its fragment mask) are loaded one step ahead and waited for at the END
    // of the step before: a scalar load's round trip right behind the barrier was most of a step (round 5, first version: 50 us per block)
    i32x4 idn = load_ids(AF_NST - 1 < n ? AF_NST - 1 : n - 1);
    unsigned fm = mk[0];
    int base = 0;
    unsigned live = 0;                                                         // bit s: step s < AF_KEEP has a fragment of this group
    const uint32_t keep_a = lds0 + AF_OFF_K + (uint32_t)((g * 16 + m) * 4);
    for (int s = 0; s < n; ++s) {
        const int slot = s % AF_NST;
        // the stage of step s has landed when at most the YOUNGER stages' instructions are outstanding (this wave's; AF_NST - 2 stages
        // while the block lasts, fewer at its end), and the barrier makes that true for every wave's rows; it also says every wave
        // is done reading the slot that step s + AF_NST - 1 refills
        {
            const int younger = (tune & 16) ? 0 : n - 1 - s < AF_NST - 2 ? n - 1 - s : AF_NST - 2;
            static_assert(AF_NST == 4, "the hand-over below lists the waits of a four-stage ring");
            if (younger == 2) cs_handover<2 * AF_DMA>();
            else if (younger == 1) cs_handover<1 * AF_DMA>();
            else cs_handover<0>();
        }
        if (s + AF_NST - 1 < n) issue(idn, s + AF_NST - 1, (s + AF_NST - 1) % AF_NST);
        const int kn = s + AF_NST < n ? s + AF_NST : n - 1;
        const i32x4 idn2 = load_ids(kn);
        const unsigned fmn = mk[s + 1 < n ? s + 1 : n - 1];
        if ((fm >> g) & 1u) {                                                  // wave-uniform: this group has a neighbour in the step
            uint32_t v;
            asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(rdv + (uint32_t)(slot * CS_NW * 256)) : "memory");
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            const uint32_t so = (uint32_t)(slot * AF_STAGE);
            if (!(tune & 1)) {
                f16x8 ah0[4], al0[4], ah1[4], al1[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    cs_rd128<0>(ah0[ks], rd[0][ks] + so);
                    cs_rd128<AF_PLANE>(al0[ks], rd[0][ks] + so);
                    cs_rd128<0>(ah1[ks], rd[1][ks] + so);
                    cs_rd128<AF_PLANE>(al1[ks], rd[1][ks] + so);
                }
                cs_wait_a(ah0, al0);
                cs_wait_a(ah1, al1);
                // four independent chains (two tiles x {hi hi, cross terms}): no MFMA waits for the one before it, and the cross terms
                // (2^-11 of the main term) are summed among themselves -- added one by one to the 2^20-sized main sum, each of the
                // twelve additions rounded at THAT size (cosine error 2e-7, weights 1.2e-6 from the fp64 softmax); now four do
                f32x4 cr[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0[ks], bh[ks], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1[ks], bh[ks], acc[1], 0, 0, 0);
                    cr[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0[ks], bl[ks], cr[0], 0, 0, 0);
                    cr[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1[ks], bl[ks], cr[1], 0, 0, 0);
                    cr[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0[ks], bh[ks], cr[0], 0, 0, 0);
                    cr[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1[ks], bh[ks], cr[1], 0, 0, 0);
                }
                acc[0] += cr[0];
                acc[1] += cr[1];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v));                    // (the validity word: read before the fragments)
            if (tune & 16) v = 0xFFFFFFFFu;
            if (s < AF_KEEP) {                                                 // kept for the fragment pass (q == 0 lanes: one word per row)
                if (q == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(keep_a + (uint32_t)(s * CS_BR * 4)), "v"(v) : "memory");
                live |= 1u << s;
            }
            // C layout: this lane holds union rows 8 q + 4 t + r of the step for its row m
            if (!(tune & 2)) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kb = 8 * q + 4 * t + r;
                        if ((v >> kb) & 1u) {
                            const int sl = base + __popc(v & ((1u << kb) - 1u));
                            // (inline asm: a compiler-visible LDS access inside the loop would wait for every LDS-DMA in flight)
                            if (sl < AF_KMAX) asm volatile("ds_write_b32 %0, %1" ::"v"(plist_a + (uint32_t)(sl * 4)), "v"(acc[t][r]) : "memory");
                        }
                    }
            }
            base += __popc(v);
        }
        asm volatile("" ::"s"(idn2.x), "s"(idn2.y), "s"(idn2.z), "s"(idn2.w), "s"(fmn));    // (the scalar loads are waited for HERE)
        idn = idn2;
        fm = fmn;
    }
    // ---- softmax of every row over its list (4 lanes q of a row share it: entries q, q + 4, ...)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                         // the list stores above
    gp_wave_sync();
    const int cnt = (tune & 4) ? 0 : base < AF_KMAX ? base : AF_KMAX;
    const float lscale = sharpen / (AF_ESCALE * AF_ESCALE);
    float mx = -INFINITY;
    for (int j = q; j < cnt; j += 4) mx = fmaxf(mx, plist[j] * lscale);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
    for (int j = q; j < cnt; j += 4) {
        const float ex = expf(plist[j] * lscale - mx);
        plist[j] = ex;
        sum += ex;
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float rs = cnt > 0 ? GP_POOL_CS_WSCALE / sum : 0.f;
    gp_wave_sync();
    // ---- the fragments
    base = 0;
    if (tune & 8) return;
    const uint32_t *keep = reinterpret_cast<const uint32_t *>(smem_raw + AF_OFF_K) + g * 16 + m;
    auto emit = [&](int s, uint32_t v) {
        _Float16 *fh = wa_hi + ((ks0 + s) * CS_NG + g) * 512, *fl = wa_lo + ((ks0 + s) * CS_NG + g) * 512;
        f16x8 h, l;                                              // union rows 8 q .. 8 q + 7 of row m = fragment lane q * 16 + m = this lane
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kb = 8 * q + e;
            float wv_ = 0.f;
            if ((v >> kb) & 1u) {
                const int sl = base + __popc(v & ((1u << kb) - 1u));
                if (sl < AF_KMAX) wv_ = plist[sl] * rs;
            }
            h[e] = (_Float16)wv_;
            l[e] = (_Float16)(wv_ - (float)h[e]);
        }
        *reinterpret_cast<f16x8 *>(fh + lane * 8) = h;
        *reinterpret_cast<f16x8 *>(fl + lane * 8) = l;
        base += __popc(v);
    };
    // steps below AF_KEEP: the fragment's step from the bit set, its validity word from LDS -- no memory round trip in this loop
    while (live) {
        const int s = __builtin_ctz(live);
        live &= live - 1;
        emit(s, keep[s * CS_BR]);
    }
    // longer blocks: the rest with the words re-loaded, eight steps at a time
    const uint32_t *vg = bu_valid + ks0 * CS_BR + g * 16 + m;
    for (int s0 = AF_KEEP; s0 < n; s0 += 8) {
        uint32_t v8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = (s0 + j < n && ((mk[s0 + j] >> g) & 1u)) ? vg[(int64_t)(s0 + j) * CS_BR] : 0u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int s = s0 + j;
            if (s >= n || !((mk[s] >> g) & 1u)) continue;
            emit(s, v8[j]);
        }
    }
}

// ---- dependency lists of the chained launch: row block d is in list(b) iff a union row of b lies in d, or one of d lies in b
__device__ __forceinline__ void cs_dep_bitmap(const int64_t *__restrict__ bu_off, const int32_t *__restrict__ bu_row, int64_t b, int rpb,
                                              int words, unsigned *bm) {
    for (int i = threadIdx.x; i < words; i += blockDim.x) bm[i] = 0u;
    __syncthreads();
    const int64_t o = bu_off[b];
    const int up = (int)(bu_off[b + 1] - o);
    for (int u = threadIdx.x; u < up; u += blockDim.x) {
        const int d = bu_row[o + u] / rpb;
        atomicOr(&bm[d >> 5], 1u << (d & 31));
    }
    if (threadIdx.x == 0) atomicOr(&bm[b >> 5], 1u << (b & 31));     // a block always waits for itself (its own rows are rewritten)
    __syncthreads();
}
// pass 1: the forward list, ascending (word 0 = its length n, complete only while n < CS_DEP_CAP), nfwd[b] = n
__global__ void __launch_bounds__(256)
cs_deps_fwd_kernel(const int64_t *__restrict__ bu_off, const int32_t *__restrict__ bu_row, int64_t nblocks, int rpb, int words,
                   int32_t *__restrict__ dep, int32_t *__restrict__ nfwd) {
    extern __shared__ unsigned s_bm[];
    __shared__ int s_cnt[256];
    const int64_t b = blockIdx.x;
    cs_dep_bitmap(bu_off, bu_row, b, rpb, words, s_bm);
    const int tid = threadIdx.x;
    const int per = (words + 255) / 256, w0 = tid * per, w1 = min(words, w0 + per);
    int c = 0;
    for (int w = w0; w < w1; ++w) c += __popc(s_bm[w]);
    s_cnt[tid] = c;
    __syncthreads();
    int before = 0, total = 0;
    for (int i = 0; i < 256; ++i) { const int v = s_cnt[i]; if (i < tid) before += v; total += v; }
    for (int w = w0; w < w1; ++w) {
        unsigned m = s_bm[w];
        while (m) {
            const int bit = __ffs(m) - 1;
            m &= m - 1;
            ++before;
            if (before < CS_DEP_CAP) dep[b * CS_DEP_CAP + before] = w * 32 + bit;
        }
    }
    if (tid == 0) { dep[b * CS_DEP_CAP] = total; nfwd[b] = total; }
}
// pass 2: b joins list(d) for every d in list(b) that does not name b itself (a list that overflows waits for every block)
__global__ void __launch_bounds__(256)
cs_deps_sym_kernel(const int64_t *__restrict__ bu_off, const int32_t *__restrict__ bu_row, int64_t nblocks, int rpb, int words,
                   int32_t *__restrict__ dep, const int32_t *__restrict__ nfwd) {
    extern __shared__ unsigned s_bm[];
    const int64_t b = blockIdx.x;
    cs_dep_bitmap(bu_off, bu_row, b, rpb, words, s_bm);
    for (int w = threadIdx.x; w < words; w += blockDim.x) {
        unsigned m = s_bm[w];
        while (m) {
            const int bit = __ffs(m) - 1;
            m &= m - 1;
            const int64_t d = (int64_t)w * 32 + bit;
            if (d == b) continue;
            const int nd = nfwd[d];
            if (nd >= CS_DEP_CAP) continue;                           // d waits for every block anyway
            bool found = false;
            for (int i = 1; i <= nd; ++i) found |= dep[d * CS_DEP_CAP + i] == (int32_t)b;   // (entries 1 .. nfwd are pass 1's: immutable here)
            if (found) continue;
            const int slot = atomicAdd(&dep[d * CS_DEP_CAP], 1) + 1;
            if (slot < CS_DEP_CAP) dep[d * CS_DEP_CAP + slot] = (int32_t)b;
        }
    }
}

// ------------------------------------------------------------------------------------------------ engine
// Producer / consumer form of the same operator ("engine"): ONE persistent 512-thread workgroup per CU.
//   waves 4-7 = loaders: nothing but LDS-DMA.  They fill a ring of four 32-KiB slots (32 union rows x 128 columns x
//               {hi, lo} + the step's 8 x {hi, lo} weight fragments) as fast as slots come free -- up to three stages
//               (96 KiB) in flight per CU, across tile boundaries -- and absorb the memory pipeline's back-pressure;
//   waves 0-3 = consumers (one per SIMD): wave cw owns all 128 rows x 32 columns of the tile (16 accumulator tiles),
//               copies a landed stage's operands into registers, releases the slot AT ONCE and only then multiplies.
// A tile is (128-row block, 128-column quarter); tiles of an XCD label are handed out in order, so that the label's
// 32 workgroups work on 8 neighbouring row blocks (a 1 024-row window: its union rows fit the XCD's 4-MiB L2, which the
// 2 048-row window of cs_pool_kernel does not -- 41 % instead of 68 % L2 hits, 1.28 GB instead of 1.03 GB from memory).
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
constexpr int EG_OFF_FLAG = EG_NSLOT * EG_SLOT;     // {full, fragment mask} x 4 | free[4] (uint32)
constexpr int EG_OFF_STG = EG_OFF_FLAG + 256;       // epilogue staging: 4 waves x 32 rows x CS_EP floats
constexpr int EG_STG_WAVE = 32 * CS_EP * 4;
constexpr size_t EG_SMEM = (size_t)EG_OFF_STG + 4 * EG_STG_WAVE;
constexpr int EG_DMA = 8;                           // LDS-DMA instructions per loader and stage: 4 x rows, 4 x weights

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

// (TUNE: the tuning bits of `ablate_` are honoured; the product instantiation compiles them out.  The loaders wait vmcnt(0)
// for their own stage, so a dropped instruction cannot mis-time a hand-over here.)
template <bool STAMP, bool TUNE>
__device__ __forceinline__ void
cs_engine_body(const _Float16 *__restrict__ x_hi, const _Float16 *__restrict__ x_lo, int64_t ld_x,
               const int64_t *__restrict__ bu_off, const int32_t *__restrict__ bu_row, const uint32_t *__restrict__ bu_mask,
               const _Float16 *__restrict__ wa_hi, const _Float16 *__restrict__ wa_lo, int64_t nv, int64_t nblocks,
               _Float16 *__restrict__ y_hi, _Float16 *__restrict__ y_lo, int64_t ld_y, float *__restrict__ y_f32, int64_t ld_yf,
               int ablate_, const float *__restrict__ out_scale, uint64_t *__restrict__ stamp, int rpb) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int ablate = TUNE ? ablate_ : 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem_raw;
    if (tid < 12) reinterpret_cast<uint32_t *>(smem_raw + EG_OFF_FLAG)[tid] = 0u;
    __syncthreads();
    // ---- this workgroup's tiles: label q = blockIdx & 7 owns the contiguous tile range [lo, hi); workgroup wi of the label
    //      takes tiles lo + wi, lo + wi + W, ...   (tile = 4 * row block + column quarter)
    const int64_t T = nblocks * (CS_D / EG_NC);
    const int label = blockIdx.x & 7, wi = blockIdx.x >> 3, W = (int)(gridDim.x >> 3);
    const int64_t t_lo = label * T / 8, t_hi = (label + 1) * T / 8;
    uint64_t st_t0 = 0, st_poll = 0, st_work = 0, st_epi = 0, st_steps = 0;
    if constexpr (STAMP) st_t0 = cs_now();

    if (wv >= 4) {
        // ================================================================ loaders
        // Loader l owns ring slot l and stages l, l + 4, l + 8, ... of the workgroup's stage sequence (all steps of all its tiles
        // in order): it waits until the consumers have released the slot, issues the WHOLE stage (16 x 1 KiB of rows, the
        // non-empty weight fragments), writes the stage's fragment mask next to the slot's counter, loads the scalars of its
        // next stage while the DMA is in flight, waits for its own DMA (vmcnt(0): nothing else is in its queue) and signals.
        // Four loaders = up to four stages between issue and release, and the per-stage latency chain (scalar loads, poll,
        // issue, landing) runs four stages wide instead of once per stage.
        const int l = wv - 4;
        const int du = lane >> 4, dc = lane & 15;
        const int dsw0 = (dc ^ (2 * du)) * 8, dsw1 = (dc ^ (2 * (du | 4))) * 8;  // source column of this lane's chunk: rows 0-7 / 8-15 (mod 16)
        int64_t t = t_lo + wi;
        if (t >= t_hi) return;
        int64_t ub0 = bu_off[t >> 2];
        int n = (int)((bu_off[(t >> 2) + 1] - ub0) / CS_KS);
        int k = l;                                                            // this loader's next stage = step k of tile t (k may run past n)
        uint32_t j = 0;                                                       // uses of the slot so far
        for (;;) {
            while (k >= n) {                                                  // on to the tile that holds the stage
                k -= n;
                t += W;
                if (t >= t_hi) break;
                ub0 = bu_off[t >> 2];
                n = (int)((bu_off[(t >> 2) + 1] - ub0) / CS_KS);
            }
            if (t >= t_hi) break;
            const int col0 = (int)(t & 3) * EG_NC;
            const int32_t *idg = bu_row + ub0 + (int64_t)k * CS_KS;
            i32x4 id[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) id[i] = *reinterpret_cast<const i32x4 *>(idg + 4 * i);
            const unsigned mk = bu_mask[ub0 / CS_KS + k];
            eg_wait_ge(lds0 + EG_OFF_FLAG + 32 + l * 4, 4u * j);                // free[l]: the slot's previous stage is consumed
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
        }
        return;
    }
    // ==================================================================== consumers
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
    uint32_t g = 0;
    s16x4 fb[2][2][2];
    f16x8 ah[8], al[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ah[i] = al[i] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    int64_t ub0_n = 0;
    int n_n = 0;
    {
        const int64_t t = t_lo + wi;
        if (t < t_hi) {
            ub0_n = bu_off[t >> 2];
            n_n = (int)((bu_off[(t >> 2) + 1] - ub0_n) / CS_KS);
        }
    }
    for (int64_t t = t_lo + wi; t < t_hi; t += W) {
        const int64_t b = t >> 2;
        const int col0 = (int)(t & 3) * EG_NC;
        const int64_t ub0 = ub0_n;
        const int n = n_n;
        f32x4 acc[CS_NG * 2];
#pragma unroll
        for (int i = 0; i < CS_NG * 2; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t + W < t_hi) {                                  // next tile's descriptor and first mask (scalar loads, used a tile later)
            const int64_t bn = (t + W) >> 2;
            ub0_n = bu_off[bn];
            n_n = (int)((bu_off[bn + 1] - ub0_n) / CS_KS);
        }
        for (int k = 0; k < n; ++k, ++g) {
            const uint32_t slot = g & 3u;
            uint64_t st_a = 0, st_b = 0;
            if constexpr (STAMP) st_a = cs_now();
            unsigned m;                                                                    // the stage's fragment mask rides next to the counter
            for (;;) {                                                                     // full[slot]: the slot's loader has signalled this stage
                int2 fm;
                asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(fm) : "v"(lds0 + EG_OFF_FLAG + slot * 8) : "memory");
                m = (unsigned)__builtin_amdgcn_readfirstlane(fm.y);
                if ((int32_t)((uint32_t)__builtin_amdgcn_readfirstlane(fm.x) - ((g >> 2) + 1u)) >= 0) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if constexpr (STAMP) { st_b = cs_now(); st_poll += st_b - st_a; }
            const uint32_t so_ = slot * EG_SLOT;
            const uint32_t a0 = addr[0] + so_, a1 = addr[1] + so_, aw = addr_w + so_;
            if (!(ablate & 256)) {                           // tuning aid: bit 8 skips the staged-row reads
            cs_tr<0>(fb[0][0][0], a0);
            cs_tr<4 * EG_RB>(fb[0][0][1], a0);
            cs_tr<EG_PLANE>(fb[0][1][0], a0);
            cs_tr<EG_PLANE + 4 * EG_RB>(fb[0][1][1], a0);
            cs_tr<0>(fb[1][0][0], a1);
            cs_tr<4 * EG_RB>(fb[1][0][1], a1);
            cs_tr<EG_PLANE>(fb[1][1][0], a1);
            cs_tr<EG_PLANE + 4 * EG_RB>(fb[1][1][1], a1);
            }
            if (ablate & 512) m = 0;                         // tuning aid: bit 9 skips the weight-fragment reads (and their MFMAs)
            if (m & 1u) { cs_rd128<0 * 1024>(ah[0], aw); cs_rd128<CS_WPL + 0 * 1024>(al[0], aw); }
            if (m & 2u) { cs_rd128<1 * 1024>(ah[1], aw); cs_rd128<CS_WPL + 1 * 1024>(al[1], aw); }
            if (m & 4u) { cs_rd128<2 * 1024>(ah[2], aw); cs_rd128<CS_WPL + 2 * 1024>(al[2], aw); }
            if (m & 8u) { cs_rd128<3 * 1024>(ah[3], aw); cs_rd128<CS_WPL + 3 * 1024>(al[3], aw); }
            if (m & 16u) { cs_rd128<4 * 1024>(ah[4], aw); cs_rd128<CS_WPL + 4 * 1024>(al[4], aw); }
            if (m & 32u) { cs_rd128<5 * 1024>(ah[5], aw); cs_rd128<CS_WPL + 5 * 1024>(al[5], aw); }
            if (m & 64u) { cs_rd128<6 * 1024>(ah[6], aw); cs_rd128<CS_WPL + 6 * 1024>(al[6], aw); }
            if (m & 128u) { cs_rd128<7 * 1024>(ah[7], aw); cs_rd128<CS_WPL + 7 * 1024>(al[7], aw); }
            cs_wait_b(fb);
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(ah[0]), "+v"(ah[1]), "+v"(ah[2]), "+v"(ah[3]), "+v"(ah[4]), "+v"(ah[5]), "+v"(ah[6]), "+v"(ah[7]),
                           "+v"(al[0]), "+v"(al[1]), "+v"(al[2]), "+v"(al[3]), "+v"(al[4]), "+v"(al[5]), "+v"(al[6]), "+v"(al[7]));
            eg_signal(lds0 + EG_OFF_FLAG + 32 + slot * 4);                                  // free[slot]: the operands are in registers
            if (do_mma) {
                f16x8 bh[2], bl[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) { bh[u] = cs_cat(fb[u][0][0], fb[u][0][1]); bl[u] = cs_cat(fb[u][1][0], fb[u][1][1]); }
#pragma unroll
                for (int mt = 0; mt < CS_NG; ++mt)
                    if (__builtin_expect((m >> mt) & 1u, 1)) {
#pragma unroll
                        for (int u = 0; u < 2; ++u) acc[mt * 2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bh[u], acc[mt * 2 + u], 0, 0, 0);
#pragma unroll
                        for (int u = 0; u < 2; ++u) acc[mt * 2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bl[u], acc[mt * 2 + u], 0, 0, 0);
#pragma unroll
                        for (int u = 0; u < 2; ++u) acc[mt * 2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt], bh[u], acc[mt * 2 + u], 0, 0, 0);
                    }
            }
            if constexpr (STAMP) { st_work += cs_now() - st_b; ++st_steps; }
        }
        if (ablate & 4) continue;                            // tuning aid: bit 2 skips the epilogue
        // ---- epilogue: 32 rows at a time through the wave's private staging area; every store instruction writes 16 rows x 64 bytes
        uint64_t st_e = 0;
        if constexpr (STAMP) st_e = cs_now();
        const int64_t row0 = b * rpb;
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
                const int lrow = ch * 32 + it * 16 + er;
                const int64_t grow = row0 + lrow;
                if (lrow < rpb && grow < nv && !(ablate & 16)) {
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
            uint64_t *o = stamp + ((int64_t)blockIdx.x * 4 + cw) * 10;
            o[0] = 0; o[1] = 0; o[2] = 0; o[3] = st_work; o[4] = st_poll; o[5] = 0; o[6] = st_epi; o[7] = t3 - st_t0; o[8] = st_steps; o[9] = 0;
        }
    }
}

#define EG_PARAMS const _Float16 *__restrict__ x_hi, const _Float16 *__restrict__ x_lo, int64_t ld_x, const int64_t *__restrict__ bu_off,            \
                  const int32_t *__restrict__ bu_row, const uint32_t *__restrict__ bu_mask, const _Float16 *__restrict__ wa_hi,                      \
                  const _Float16 *__restrict__ wa_lo, int64_t nv, int64_t nblocks, _Float16 *__restrict__ y_hi, _Float16 *__restrict__ y_lo,         \
                  int64_t ld_y, float *__restrict__ y_f32, int64_t ld_yf, int ablate, const float *__restrict__ out_scale, uint64_t *__restrict__ stamp, \
                  int rpb
#define EG_FWD x_hi, x_lo, ld_x, bu_off, bu_row, bu_mask, wa_hi, wa_lo, nv, nblocks, y_hi, y_lo, ld_y, y_f32, ld_yf, ablate, out_scale, stamp, rpb
// the product engine (no tuning bits, no stamps) and the same body with both live, under its own name in a kernel trace
__global__ void __launch_bounds__(512, 2) cs_engine_kernel(EG_PARAMS) { cs_engine_body<false, false>(EG_FWD); }
template <bool STAMP>
__global__ void __launch_bounds__(512, 2) cs_engine_tuning_kernel(EG_PARAMS) { cs_engine_body<STAMP, true>(EG_FWD); }
#undef EG_PARAMS
#undef EG_FWD

size_t cs_scan_tmp(int64_t n) {
    size_t t = 0;
    (void)rocprim::exclusive_scan(nullptr, t, (int64_t *)nullptr, (int64_t *)nullptr, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), 0);
    return t;
}
int cs_np2(int64_t n) { int p = 1; while (p < n) p <<= 1; return p; }

}  // namespace

// rows per block: 16 .. 128 (the kernels' blocks hold eight 16-row groups; a block of fewer rows leaves the last groups partly
// empty, their weight fragments are never fetched).  128 is the default and the fastest: round 4 measured every height from 96
// to 128 -- and, with a ten-group instantiation of the kernel, from 131 to 152 -- within 4 % of each other on the S scene
// (profiles/r04_pool_block_height.log: 6.9 to 10.9 "rounds" of tiles, 0.222 - 0.232 ms): the launch is not bound by whole rounds.
static bool cs_rpb_ok(int32_t rpb) { return rpb >= 16 && rpb <= CS_BR; }

extern "C" size_t gp_pool_cs_workspace_bytes(int64_t nv, int32_t rows_per_block) {
    if (nv <= 0 || !cs_rpb_ok(rows_per_block)) return 0;
    int64_t nb = (nv + rows_per_block - 1) / rows_per_block;
    GpCarver cv(nullptr, 0);
    cv.take<int64_t>(nb + 1);
    cv.take<char>(cs_scan_tmp(nb + 1));
    return cv.off;
}

// pass 1: bu_off i64 [nblocks+1] (padded union rows before each block; multiples of 32), bu_n i32 [nblocks];
// nblocks = ceil(nv / rows_per_block)
extern "C" int gp_pool_cs_count(const int32_t *nbr, int64_t nv, int32_t k, int32_t rows_per_block, int64_t *bu_off, int32_t *bu_n,
                                int64_t *max_union, void *workspace, size_t workspace_bytes, void *stream_) {
    GP_CHECK_ARG(nbr && bu_off && bu_n && workspace && nv > 0 && k > 0, "gp_pool_cs_count: null/empty argument");
    GP_CHECK_ARG(cs_rpb_ok(rows_per_block), "gp_pool_cs_count: rows_per_block=%d (16..%d)", rows_per_block, CS_BR);
    GP_CHECK_ARG((int64_t)CS_BR * k <= CS_MAXNK, "gp_pool_cs_count: k=%d too large (128*k <= %d)", k, CS_MAXNK);
    int64_t nb = (nv + rows_per_block - 1) / rows_per_block;
    GpCarver cv(workspace, workspace_bytes);
    int64_t *cnt = cv.take<int64_t>(nb + 1);
    size_t tb = cs_scan_tmp(nb + 1);
    char *tmp = cv.take<char>(tb);
    if (!cv.ok()) { gp_set_error("gp_pool_cs_count: workspace too small"); return GP_ENOMEM; }
    hipStream_t s = gp_stream(stream_);
    GP_CHECK_HIP(hipMemsetAsync(cnt + nb, 0, sizeof(int64_t), s));
    if (max_union) GP_CHECK_HIP(hipMemsetAsync(max_union, 0, sizeof(int64_t), s));
    // every block with the 2048-slot table (16 KiB), then the blocks it marked (a union above 1024 ids) with the full-size one
    const int cap_big = CS_HS > cs_np2((int64_t)CS_BR * k) ? CS_HS : cs_np2((int64_t)CS_BR * k);
    GP_SMEM_ATTR(cs_count_kernel, (size_t)2 * CS_HS * sizeof(int));
    cs_count_kernel<<<(unsigned)nb, 1024, (size_t)2 * 2048 * sizeof(int), s>>>(nbr, nv, k, rows_per_block, cnt, bu_n, 2048, 0,
                                                                              reinterpret_cast<unsigned long long *>(max_union));
    cs_count_kernel<<<(unsigned)nb, 1024, (size_t)2 * cap_big * sizeof(int), s>>>(nbr, nv, k, rows_per_block, cnt, bu_n, cap_big, 1,
                                                                                 reinterpret_cast<unsigned long long *>(max_union));
    GP_CHECK_HIP(rocprim::exclusive_scan(tmp, tb, cnt, bu_off, (int64_t)0, (size_t)(nb + 1), rocprim::plus<int64_t>(), s));
    GP_CHECK_LAUNCH();
    return GP_OK;
}

// LDS of the fill kernels: two tables of `cap` ints (cap = a power of two >= 2 x the largest union, 2048 .. 16384; max_union = 0:
// not known, the full size) + the position table
static int cs_fill_cap(int32_t max_union) {
    if (max_union <= 0) return CS_HS;
    int cap = cs_np2(2 * (int64_t)max_union);
    return cap < 2048 ? 2048 : cap > CS_HS ? CS_HS : cap;
}
static size_t cs_fill_smem(int cap, int32_t k) { return (size_t)2 * cap * sizeof(int) + (size_t)CS_BR * k * sizeof(unsigned short); }

// pass 2: bu_row i32 [total], bu_mask u32 [total/32], wa_hi / wa_lo f16 [total/32 * 8 * 512] (only the fragments whose
// mask bit is set are defined -- and read)
extern "C" int gp_pool_cs_fill(const int32_t *nbr, const float *w, int64_t nv, int32_t k, int32_t rows_per_block, const int64_t *bu_off,
                               int64_t total_rows, int32_t max_union, int32_t *bu_row, uint32_t *bu_mask, void *wa_hi, void *wa_lo,
                               void *stream_) {
    GP_CHECK_ARG(nbr && w && bu_off && bu_row && bu_mask && wa_hi && wa_lo && nv > 0 && total_rows > 0 && total_rows % CS_KS == 0,
                 "gp_pool_cs_fill: bad argument");
    GP_CHECK_ARG(cs_rpb_ok(rows_per_block), "gp_pool_cs_fill: rows_per_block=%d (16..%d)", rows_per_block, CS_BR);
    GP_CHECK_ARG((int64_t)CS_BR * k <= CS_MAXNK, "gp_pool_cs_fill: k=%d too large (128*k <= %d)", k, CS_MAXNK);
    int64_t nb = (nv + rows_per_block - 1) / rows_per_block;
    hipStream_t s = gp_stream(stream_);
    const size_t sm_max = (size_t)(CS_HS + CS_MAXID) * sizeof(int) + (size_t)CS_MAXNK * sizeof(unsigned short);
    const int cap = cs_fill_cap(max_union);
    GP_SMEM_ATTR(cs_fill_kernel<CS_FILL_WEIGHTS>, sm_max);
    cs_fill_kernel<CS_FILL_WEIGHTS><<<(unsigned)nb, 1024, cs_fill_smem(cap, k), s>>>(nbr, w, nv, k, rows_per_block, bu_off, bu_row, bu_mask,
                                                                                    static_cast<_Float16 *>(wa_hi), static_cast<_Float16 *>(wa_lo),
                                                                                    nullptr, nullptr, cap);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

// pass 2 without the weights: the operator's structure (bu_row, bu_mask, zeroed fragments) and dst i32 [nv, k], the element of
// wa_hi / wa_lo that (row, neighbour j) owns.  gp_affinity_softmax_scatter then writes the weights in place; a scheduler runs this
// before the student's embeddings exist (it needs the kNN lists only).
extern "C" int gp_pool_cs_structure(const int32_t *nbr, int64_t nv, int32_t k, int32_t rows_per_block, const int64_t *bu_off,
                                    int64_t total_rows, int32_t max_union, int32_t *bu_row, uint32_t *bu_mask, void *wa_hi, void *wa_lo,
                                    int32_t *dst, void *stream_) {
    GP_CHECK_ARG(nbr && dst && bu_off && bu_row && bu_mask && wa_hi && wa_lo && nv > 0 && total_rows > 0 && total_rows % CS_KS == 0,
                 "gp_pool_cs_structure: bad argument");
    GP_CHECK_ARG(cs_rpb_ok(rows_per_block), "gp_pool_cs_structure: rows_per_block=%d (16..%d)", rows_per_block, CS_BR);
    GP_CHECK_ARG((int64_t)CS_BR * k <= CS_MAXNK, "gp_pool_cs_structure: k=%d too large (128*k <= %d)", k, CS_MAXNK);
    GP_CHECK_ARG(total_rows * (CS_NG * 64 * 8 / CS_KS) < (int64_t)INT32_MAX,
                 "gp_pool_cs_structure: %lld union rows: fragment element indices do not fit 32 bits (use gp_pool_cs_fill)", (long long)total_rows);
    int64_t nb = (nv + rows_per_block - 1) / rows_per_block;
    hipStream_t s = gp_stream(stream_);
    const size_t sm_max = (size_t)(CS_HS + CS_MAXID) * sizeof(int) + (size_t)CS_MAXNK * sizeof(unsigned short);
    const int cap = cs_fill_cap(max_union);
    GP_SMEM_ATTR(cs_fill_kernel<CS_FILL_DST>, sm_max);
    cs_fill_kernel<CS_FILL_DST><<<(unsigned)nb, 1024, cs_fill_smem(cap, k), s>>>(nbr, nullptr, nv, k, rows_per_block, bu_off, bu_row, bu_mask,
                                                                                static_cast<_Float16 *>(wa_hi), static_cast<_Float16 *>(wa_lo), dst,
                                                                                nullptr, cap);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

// One application y = A x on pre-split operands (see gp_pool_mfma_apply for the operand conventions).  d must be 512.
// engine = false: cs_pool_kernel (one tile per workgroup, the default); engine = true: cs_engine_kernel (persistent producer /
// consumer form, one workgroup per CU).  The choice is an ARGUMENT of the call (round 3 selected the engine through the
// process-global debug knob 11: a raised error left every later launch on the engine, and two host threads raced on it).
// rows_per_block: the builder's (gp_pool_cs_count / _fill).
static int cs_apply(const void *x_hi, const void *x_lo, int64_t ld_x, const int64_t *bu_off, const int32_t *bu_row,
                    const uint32_t *bu_mask, const void *wa_hi, const void *wa_lo, int64_t nv, int32_t d, int32_t rpb,
                    void *y_hi, void *y_lo, int64_t ld_y, float *y_f32, int64_t ld_yf,
                    const float *out_scale, bool engine, void *stream_, int half = -1) {
    GP_CHECK_ARG(x_hi && x_lo && bu_off && bu_row && bu_mask && wa_hi && wa_lo && nv > 0, "gp_pool_cs_apply: null/empty argument");
    GP_CHECK_ARG(half >= -1 && half <= 1 && !(engine && half >= 0), "gp_pool_cs_apply_half: half=%d (0 or 1)", half);
    GP_CHECK_ARG(d == CS_D, "gp_pool_cs_apply: d=%d (kernel specialised for %d columns)", d, CS_D);
    GP_CHECK_ARG(cs_rpb_ok(rpb), "gp_pool_cs_apply: rows_per_block=%d (16..%d)", rpb, CS_BR);
    GP_CHECK_ARG((y_hi && y_lo) || y_f32, "gp_pool_cs_apply: no output requested");
    GP_CHECK_ARG(ld_x % 8 == 0 && (uintptr_t)x_hi % 16 == 0 && (uintptr_t)x_lo % 16 == 0, "gp_pool_cs_apply: x rows must be 16-byte aligned");
    GP_CHECK_ARG(!y_hi || (ld_y % 8 == 0 && (uintptr_t)y_hi % 16 == 0 && (uintptr_t)y_lo % 16 == 0 && y_hi != x_hi && y_lo != x_lo),
                 "gp_pool_cs_apply: y rows must be 16-byte aligned and must not alias x");
    GP_CHECK_ARG(!y_f32 || (ld_yf % 4 == 0 && (uintptr_t)y_f32 % 16 == 0), "gp_pool_cs_apply: fp32 output rows must be 16-byte aligned");
    hipStream_t s = gp_stream(stream_);
    const int64_t nb = (nv + rpb - 1) / rpb;
    const int64_t per_xcd = half >= 0 ? (nb + 7) / 8 : (nb * (CS_D / CS_NC) + 7) / 8;
    uint64_t *stamp = static_cast<uint64_t *>(g_gp_debug_ptr[0]);
    const int tune = g_gp_knobs[4];                       // tuning bits: only ever handed to the *_tuning_kernel twins
    if (engine) {
        const int n_cu = gp_cu_count();
        GP_CHECK_ARG(n_cu > 0, "gp_pool_cs_apply_engine: cannot read the device's compute-unit count");
        const unsigned grid = (unsigned)((n_cu >= 8 ? n_cu / 8 : 1) * 8);
        GP_CHECK_ARG(!stamp || g_gp_debug_bytes[0] >= (size_t)grid * 4 * 10 * sizeof(uint64_t),
                     "gp_pool_cs_apply_engine: the stamp buffer of gp_debug_ptr(0) holds %zu bytes, this launch writes %zu",
                     g_gp_debug_bytes[0], (size_t)grid * 4 * 10 * sizeof(uint64_t));
#define EG_ARGS static_cast<const _Float16 *>(x_hi), static_cast<const _Float16 *>(x_lo), ld_x, bu_off, bu_row, bu_mask,              \
                static_cast<const _Float16 *>(wa_hi), static_cast<const _Float16 *>(wa_lo), nv, nb, static_cast<_Float16 *>(y_hi),     \
                static_cast<_Float16 *>(y_lo), ld_y, y_f32, ld_yf, tune, out_scale, stamp, rpb
        if (stamp) {
            GP_SMEM_ATTR(cs_engine_tuning_kernel<true>, EG_SMEM);
            cs_engine_tuning_kernel<true><<<grid, 512, EG_SMEM, s>>>(EG_ARGS);
        } else if (tune != 0) {
            GP_SMEM_ATTR(cs_engine_tuning_kernel<false>, EG_SMEM);
            cs_engine_tuning_kernel<false><<<grid, 512, EG_SMEM, s>>>(EG_ARGS);
        } else {
            GP_SMEM_ATTR(cs_engine_kernel, EG_SMEM);
            cs_engine_kernel<<<grid, 512, EG_SMEM, s>>>(EG_ARGS);
        }
#undef EG_ARGS
        GP_CHECK_LAUNCH();
        return GP_OK;
    }
    GP_CHECK_ARG(!stamp || g_gp_debug_bytes[0] >= (size_t)(per_xcd * 8) * CS_NW * 10 * sizeof(uint64_t),
                 "gp_pool_cs_apply: the stamp buffer of gp_debug_ptr(0) holds %zu bytes, this launch writes %zu",
                 g_gp_debug_bytes[0], (size_t)(per_xcd * 8) * CS_NW * 10 * sizeof(uint64_t));
#define CS_ARGS static_cast<const _Float16 *>(x_hi), static_cast<const _Float16 *>(x_lo), ld_x, bu_off, bu_row, bu_mask,              \
                static_cast<const _Float16 *>(wa_hi), static_cast<const _Float16 *>(wa_lo), nv, nb, static_cast<_Float16 *>(y_hi),     \
                static_cast<_Float16 *>(y_lo), ld_y, y_f32, ld_yf, per_xcd, tune, out_scale, stamp, rpb
    if (half >= 0) {
        GP_SMEM_ATTR(cs_pool_half_kernel, CS_SMEM);
        cs_pool_half_kernel<<<(unsigned)(per_xcd * 8), 512, CS_SMEM, s>>>(CS_ARGS, half);
    } else if (stamp) {
        GP_SMEM_ATTR(cs_pool_kernel<true>, CS_SMEM);
        cs_pool_kernel<true><<<(unsigned)(per_xcd * 8), 512, CS_SMEM, s>>>(CS_ARGS);
    } else if (tune != 0) {
        GP_SMEM_ATTR(cs_pool_tuning_kernel, CS_SMEM);
        cs_pool_tuning_kernel<<<(unsigned)(per_xcd * 8), 512, CS_SMEM, s>>>(CS_ARGS);
    } else {
        GP_SMEM_ATTR(cs_pool_kernel<false>, CS_SMEM);
        cs_pool_kernel<false><<<(unsigned)(per_xcd * 8), 512, CS_SMEM, s>>>(CS_ARGS);
    }
#undef CS_ARGS
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" int gp_pool_cs_apply(const void *x_hi, const void *x_lo, int64_t ld_x, const int64_t *bu_off, const int32_t *bu_row,
                                const uint32_t *bu_mask, const void *wa_hi, const void *wa_lo, int64_t nv, int32_t d,
                                int32_t rows_per_block, void *y_hi, void *y_lo, int64_t ld_y, float *y_f32,
                                int64_t ld_yf, const float *out_scale, void *stream_) {
    return cs_apply(x_hi, x_lo, ld_x, bu_off, bu_row, bu_mask, wa_hi, wa_lo, nv, d, rows_per_block, y_hi, y_lo, ld_y, y_f32,
                    ld_yf, out_scale, false, stream_);
}

// One 256-column half (0 or 1) of the same application: columns 256 half .. 256 half + 255 of every row.  The halves are
// independent, so two streams can each carry one half's chain of applications (measured in round 5).
extern "C" int gp_pool_cs_apply_half(const void *x_hi, const void *x_lo, int64_t ld_x, const int64_t *bu_off, const int32_t *bu_row,
                                     const uint32_t *bu_mask, const void *wa_hi, const void *wa_lo, int64_t nv, int32_t d,
                                     int32_t rows_per_block, int32_t half, void *y_hi, void *y_lo, int64_t ld_y, float *y_f32,
                                     int64_t ld_yf, const float *out_scale, void *stream_) {
    GP_CHECK_ARG(half == 0 || half == 1, "gp_pool_cs_apply_half: half=%d (0 or 1)", half);
    return cs_apply(x_hi, x_lo, ld_x, bu_off, bu_row, bu_mask, wa_hi, wa_lo, nv, d, rows_per_block, y_hi, y_lo, ld_y, y_f32,
                    ld_yf, out_scale, false, stream_, half);
}

// The same application through the persistent producer / consumer engine (cs_engine_kernel); bit-identical results.
extern "C" int gp_pool_cs_apply_engine(const void *x_hi, const void *x_lo, int64_t ld_x, const int64_t *bu_off, const int32_t *bu_row,
                                       const uint32_t *bu_mask, const void *wa_hi, const void *wa_lo, int64_t nv, int32_t d,
                                       int32_t rows_per_block, void *y_hi, void *y_lo, int64_t ld_y, float *y_f32, int64_t ld_yf,
                                       const float *out_scale, void *stream_) {
    return cs_apply(x_hi, x_lo, ld_x, bu_off, bu_row, bu_mask, wa_hi, wa_lo, nv, d, rows_per_block, y_hi, y_lo, ld_y, y_f32, ld_yf,
                    out_scale, true, stream_);
}

// gp_pool_cs_structure for gp_affinity_cs_fragments: union rows, fragment masks and valid u32 [total_rows / 32 * 128 + 64] (bit p of
// valid[step * 128 + row] = union row 32 step + p of the row's block is one of its neighbours; the last 64 words are padding that the
// affinity kernel's 64-word fetches may touch).  No fragment is written: the affinity kernel writes every non-empty fragment whole.
// The neighbour ids of a row must be distinct (a k-NN list is).
extern "C" int gp_pool_cs_structure_valid(const int32_t *nbr, int64_t nv, int32_t k, int32_t rows_per_block, const int64_t *bu_off,
                                          int64_t total_rows, int32_t max_union, int32_t *bu_row, uint32_t *bu_mask, uint32_t *bu_valid,
                                          void *stream_) {
    GP_CHECK_ARG(nbr && bu_off && bu_row && bu_mask && bu_valid && nv > 0 && total_rows > 0 && total_rows % CS_KS == 0,
                 "gp_pool_cs_structure_valid: bad argument");
    GP_CHECK_ARG(cs_rpb_ok(rows_per_block), "gp_pool_cs_structure_valid: rows_per_block=%d (16..%d)", rows_per_block, CS_BR);
    GP_CHECK_ARG((int64_t)CS_BR * k <= CS_MAXNK, "gp_pool_cs_structure_valid: k=%d too large (128*k <= %d)", k, CS_MAXNK);
    int64_t nb = (nv + rows_per_block - 1) / rows_per_block;
    hipStream_t s = gp_stream(stream_);
    const size_t sm_max = (size_t)(CS_HS + CS_MAXID) * sizeof(int) + (size_t)CS_MAXNK * sizeof(unsigned short);
    const int cap = cs_fill_cap(max_union);
    GP_SMEM_ATTR(cs_fill_kernel<CS_FILL_VALID>, sm_max);
    GP_CHECK_HIP(hipMemsetAsync(bu_valid + total_rows / CS_KS * CS_BR, 0, 64 * sizeof(uint32_t), s));
    cs_fill_kernel<CS_FILL_VALID><<<(unsigned)nb, 1024, cs_fill_smem(cap, k), s>>>(nbr, nullptr, nv, k, rows_per_block, bu_off, bu_row, bu_mask, nullptr,
                                                                                  nullptr, nullptr, bu_valid, cap);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

// Row 11 fused with the operator fill, on the matrix cores: from the unit embeddings as f16 hi / lo planes of e x 2^10
// (gp_split_f16_scaled with scale 1024; d = 128, rows of 128 halves) and the operator's structure (gp_pool_cs_structure_valid) to the
// weight fragments wa_hi / wa_lo = softmax_j(sharpen * <e_i, e_nbr(i,j)>) x 2^10 in the order gp_pool_cs_apply reads -- the cosine /
// softmax / COO build of models/affinity_module.py:1559-1572 and the fill pass in one kernel.  k <= 96.  No [nv, k] weight matrix
// is produced (gp_affinity_softmax does that).
extern "C" int gp_affinity_cs_fragments(const void *e_hi, const void *e_lo, int64_t nv, int32_t d, int32_t k, float sharpen,
                                        const int64_t *bu_off, const int32_t *bu_row, const uint32_t *bu_mask, const uint32_t *bu_valid,
                                        int32_t rows_per_block, void *wa_hi, void *wa_lo, void *stream_) {
    GP_CHECK_ARG(e_hi && e_lo && bu_off && bu_row && bu_mask && bu_valid && wa_hi && wa_lo && nv > 0, "gp_affinity_cs_fragments: null/empty argument");
    GP_CHECK_ARG(d == AF_D, "gp_affinity_cs_fragments: d=%d (kernel specialised for %d-wide embeddings)", d, AF_D);
    GP_CHECK_ARG(k > 0 && k <= AF_KMAX, "gp_affinity_cs_fragments: k=%d (1..%d)", k, AF_KMAX);
    GP_CHECK_ARG(cs_rpb_ok(rows_per_block), "gp_affinity_cs_fragments: rows_per_block=%d (16..%d)", rows_per_block, CS_BR);
    GP_CHECK_ARG((uintptr_t)e_hi % 16 == 0 && (uintptr_t)e_lo % 16 == 0, "gp_affinity_cs_fragments: embedding planes must be 16-byte aligned");
    hipStream_t s = gp_stream(stream_);
    const int64_t nb = (nv + rows_per_block - 1) / rows_per_block;
    const int64_t per_xcd = (nb + 7) / 8;
    const int tune = g_gp_knobs[8];                       // tuning bits: only ever handed to the tuning twin
    if (tune != 0) {
        GP_SMEM_ATTR(affinity_cs_kernel<true>, AF_SMEM);
        affinity_cs_kernel<true><<<(unsigned)(per_xcd * 8), 512, AF_SMEM, s>>>(static_cast<const _Float16 *>(e_hi), static_cast<const _Float16 *>(e_lo),
                                                                               nv, sharpen, bu_off, bu_row, bu_mask, bu_valid, nb, rows_per_block,
                                                                               per_xcd, static_cast<_Float16 *>(wa_hi), static_cast<_Float16 *>(wa_lo), tune);
    } else {
        GP_SMEM_ATTR(affinity_cs_kernel<false>, AF_SMEM);
        affinity_cs_kernel<false><<<(unsigned)(per_xcd * 8), 512, AF_SMEM, s>>>(static_cast<const _Float16 *>(e_hi), static_cast<const _Float16 *>(e_lo),
                                                                                nv, sharpen, bu_off, bu_row, bu_mask, bu_valid, nb, rows_per_block,
                                                                                per_xcd, static_cast<_Float16 *>(wa_hi), static_cast<_Float16 *>(wa_lo), 0);
    }
    GP_CHECK_LAUNCH();
    return GP_OK;
}

// ------------------------------------------------------------------------------------------------ the chained launch
// Dependency lists for gp_pool_cs_apply_chain: dep i32 [nblocks * 64] (word 0 of a row block = n, words 1 .. min(n, 63) = the row
// blocks whose previous application it waits for; n > 63: every block), scratch i32 [nblocks].  Needs the operator's structure
// only (bu_off, bu_row: gp_pool_cs_fill or gp_pool_cs_structure), so a scheduler runs it ahead with them.
extern "C" int gp_pool_cs_deps(const int64_t *bu_off, const int32_t *bu_row, int64_t nv, int32_t rows_per_block, int32_t *dep,
                               int32_t *scratch, void *stream_) {
    GP_CHECK_ARG(bu_off && bu_row && dep && scratch && nv > 0, "gp_pool_cs_deps: null/empty argument");
    GP_CHECK_ARG(cs_rpb_ok(rows_per_block), "gp_pool_cs_deps: rows_per_block=%d (16..%d)", rows_per_block, CS_BR);
    const int64_t nb = (nv + rows_per_block - 1) / rows_per_block;
    const int64_t words = (nb + 31) / 32;
    GP_CHECK_ARG(words * 4 <= 64 * 1024, "gp_pool_cs_deps: %lld row blocks (the block bitmap holds 524288)", (long long)nb);
    hipStream_t s = gp_stream(stream_);
    cs_deps_fwd_kernel<<<(unsigned)nb, 256, (size_t)words * 4, s>>>(bu_off, bu_row, nb, rows_per_block, (int)words, dep, scratch);
    cs_deps_sym_kernel<<<(unsigned)nb, 256, (size_t)words * 4, s>>>(bu_off, bu_row, nb, rows_per_block, (int)words, dep, scratch);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

// words of the flags array of gp_pool_cs_apply_chain: 32 header words (word 0 = abort) + 2 column halves x nblocks
extern "C" size_t gp_pool_cs_chain_flag_words(int64_t nv, int32_t rows_per_block) {
    if (nv <= 0 || !cs_rpb_ok(rows_per_block)) return 0;
    return (size_t)CS_FLAG_HDR + 2 * (size_t)((nv + rows_per_block - 1) / rows_per_block);
}

// ALL `applications` (>= 2) of y = A x in ONE launch: application t reads plane set (t even ? x : p) and writes the other one, the
// last one writes y_f32 (x out_scale[0]) only -- the same sequence, planes and bits as `applications` calls of gp_pool_cs_apply
// that ping-pong between x and p.  x_hi / x_lo are REWRITTEN (from application 1 on), as in that sequence.
//   dep    gp_pool_cs_deps' lists.
//   flags  u32 [gp_pool_cs_chain_flag_words]: zeroed ONCE by the caller when allocated, never again; word 0 is the abort word: the
//          kernel sets it to 1 if a workgroup waited 2 s for a dependency (the launch then drains without computing; the outputs
//          are invalid) -- the caller reads it at its next synchronisation point and must treat non-zero as an error.
//   epoch  a counter the caller keeps per flags array: every call passes a value at least `applications` above the previous call's
//          (the published values are epoch + 1 .. epoch + applications - 1; 32-bit wrap-around is handled by signed differences).
// One flags array serves one launch at a time (launches on one stream are ordered; do not share it between streams).
// Workgroups wait only for workgroups with a smaller index, so the grid needs no co-residency guarantee beyond the in-order
// dispatch of workgroups; should that order ever not hold, the bounded wait turns a hang into the abort word.
extern "C" int gp_pool_cs_apply_chain(void *x_hi, void *x_lo, void *p_hi, void *p_lo, int64_t ld, const int64_t *bu_off,
                                      const int32_t *bu_row, const uint32_t *bu_mask, const void *wa_hi, const void *wa_lo, int64_t nv,
                                      int32_t d, int32_t rows_per_block, int32_t applications, float *y_f32, int64_t ld_yf,
                                      const float *out_scale, const int32_t *dep, uint32_t *flags, uint32_t epoch, void *stream_) {
    GP_CHECK_ARG(x_hi && x_lo && p_hi && p_lo && bu_off && bu_row && bu_mask && wa_hi && wa_lo && y_f32 && dep && flags && nv > 0,
                 "gp_pool_cs_apply_chain: null/empty argument");
    GP_CHECK_ARG(d == CS_D, "gp_pool_cs_apply_chain: d=%d (kernel specialised for %d columns)", d, CS_D);
    GP_CHECK_ARG(cs_rpb_ok(rows_per_block), "gp_pool_cs_apply_chain: rows_per_block=%d (16..%d)", rows_per_block, CS_BR);
    GP_CHECK_ARG(applications >= 2 && applications < 65536, "gp_pool_cs_apply_chain: applications=%d (2..65535; one: gp_pool_cs_apply)", applications);
    GP_CHECK_ARG(ld % 8 == 0 && (uintptr_t)x_hi % 16 == 0 && (uintptr_t)x_lo % 16 == 0 && (uintptr_t)p_hi % 16 == 0 && (uintptr_t)p_lo % 16 == 0,
                 "gp_pool_cs_apply_chain: plane rows must be 16-byte aligned");
    GP_CHECK_ARG(x_hi != p_hi && x_lo != p_lo && x_hi != x_lo && p_hi != p_lo, "gp_pool_cs_apply_chain: the four planes must not alias");
    GP_CHECK_ARG(ld_yf % 4 == 0 && (uintptr_t)y_f32 % 16 == 0, "gp_pool_cs_apply_chain: fp32 output rows must be 16-byte aligned");
    hipStream_t s = gp_stream(stream_);
    const int64_t nb = (nv + rows_per_block - 1) / rows_per_block;
    const int64_t per_xcd = (nb * (CS_D / CS_NC) + 7) / 8;
    const int64_t grid = per_xcd * 8 * applications;
    GP_CHECK_ARG(grid < (int64_t)INT32_MAX, "gp_pool_cs_apply_chain: %lld workgroups", (long long)grid);
    CsChain ch;
    ch.a_hi = static_cast<const _Float16 *>(x_hi);
    ch.a_lo = static_cast<const _Float16 *>(x_lo);
    ch.b_hi = static_cast<_Float16 *>(p_hi);
    ch.b_lo = static_cast<_Float16 *>(p_lo);
    ch.ld = ld;
    ch.flags = flags;
    ch.dep = dep;
    ch.T = applications;
    ch.base = epoch;
    uint64_t *stamp = static_cast<uint64_t *>(g_gp_debug_ptr[0]);
    GP_CHECK_ARG(!stamp || g_gp_debug_bytes[0] >= (size_t)grid * CS_NW * 10 * sizeof(uint64_t),
                 "gp_pool_cs_apply_chain: the stamp buffer of gp_debug_ptr(0) holds %zu bytes, this launch writes %zu",
                 g_gp_debug_bytes[0], (size_t)grid * CS_NW * 10 * sizeof(uint64_t));
    if (stamp) {
        GP_SMEM_ATTR(cs_chain_kernel<true>, CS_SMEM);
        cs_chain_kernel<true><<<(unsigned)grid, 512, CS_SMEM, s>>>(ch, bu_off, bu_row, bu_mask, static_cast<const _Float16 *>(wa_hi),
                                                                    static_cast<const _Float16 *>(wa_lo), nv, nb, y_f32, ld_yf, per_xcd,
                                                                    out_scale, stamp, rows_per_block);
    } else {
        GP_SMEM_ATTR(cs_chain_kernel<false>, CS_SMEM);
        cs_chain_kernel<false><<<(unsigned)grid, 512, CS_SMEM, s>>>(ch, bu_off, bu_row, bu_mask, static_cast<const _Float16 *>(wa_hi),
                                                                     static_cast<const _Float16 *>(wa_lo), nv, nb, y_f32, ld_yf, per_xcd,
                                                                     out_scale, nullptr, rows_per_block);
    }
    GP_CHECK_LAUNCH();
    return GP_OK;
}
