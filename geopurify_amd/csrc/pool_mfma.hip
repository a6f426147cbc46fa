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
// One workgroup = 64 rows x 128 columns (grid = row blocks x 4 column quarters, the 4 quarters of a row
// block adjacent on one XCD); 2 workgroups per CU.  A 3-deep ring of stages is filled by LDS-DMA two steps
// ahead (~100 KiB in flight per CU: below that the gather is latency-bound, Little's law at ~14 TB/s of L2
// bandwidth).  A stage carries everything step k needs: 32 union rows x 128 columns x {hi, lo}, the four
// waves' weight fragments, and the row ids of stage k+2 (read back when that stage is issued), so the stage
// hand-over is the only synchronisation in the loop.
//
// Synchronisation is hand-counted: the LDS reads are inline asm with explicit lgkmcnt waits (a
// compiler-visible LDS read would wait for EVERY outstanding LDS-DMA, since the compiler cannot see that
// they target other ring slots, and serialise the ring), and the hand-over is `s_waitcnt vmcnt(N);
// s_barrier` with N = the DMA instructions issued for the later stage (loads return in order).
constexpr int PQ_NC = 128;                       // columns per workgroup
constexpr int PQ_NST = 3;                        // ring stages
constexpr int PQ_PLANE = PM_KS * PQ_NC * 2;      // 8 KiB: 32 rows x 256 B
constexpr int PQ_OFF_W = 2 * PQ_PLANE;           // weights: hi 4 x 1 KiB, lo 4 x 1 KiB
constexpr int PQ_OFF_ID = PQ_OFF_W + 8192;       // row ids: 4 waves x 256 B (64 lanes x 4 B per DMA, 8 ids used)
constexpr int PQ_STAGE = PQ_OFF_ID + 1024;       // 25600 B
constexpr int PQ_DMA_PER_STAGE = 7;              // per wave: 4 x rows, 2 x weights, 1 x ids
constexpr int PQ_EP = PQ_NC + 4;                 // epilogue staging pitch (floats)
constexpr size_t PQ_SMEM_BYTES = (size_t)PQ_NST * PQ_STAGE;
static_assert((size_t)PM_W * 16 * PQ_EP * sizeof(float) <= PQ_SMEM_BYTES, "epilogue staging must fit in the ring");
static_assert(2 * PQ_STAGE < 65536 && PQ_NST == 3, "LDS offsets are 16-bit immediates: slots 0,1 from one base, slot 2 from a second");

template <int OFF>
__device__ __forceinline__ void pq_tr(s16x4 &d, uint32_t addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void pq_rd128(f16x8 &d, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void pq_rd64(int2 &d, uint32_t addr) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void pq_wait_lgkm(s16x4 (&f)[2][2][2]) {
    asm volatile("s_waitcnt lgkmcnt(%[n])"
                 : "+v"(f[0][0][0]), "+v"(f[0][0][1]), "+v"(f[0][1][0]), "+v"(f[0][1][1]), "+v"(f[1][0][0]), "+v"(f[1][0][1]),
                   "+v"(f[1][1][0]), "+v"(f[1][1][1])
                 : [n] "n"(N));
}
template <int N>
__device__ __forceinline__ void pq_wait_lgkm3(int2 &id, f16x8 &a, f16x8 &b) {
    asm volatile("s_waitcnt lgkmcnt(%[n])" : "+v"(id), "+v"(a), "+v"(b) : [n] "n"(N));
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
// transposed reads of 2 column blocks (CB0, CB0+1): f[u][plane][half].  Ring slots 0,1 are addressed from
// the slot-0 base registers (J = 0,1), slot 2 from a second set of base registers with J = 0
// (lgkmcnt is a 4-bit counter: groups of 8 reads, at most 11 LDS operations outstanding)
template <int J, int CB0>
__device__ __forceinline__ void pq_read_group(s16x4 (&f)[2][2][2], const uint32_t (&addr)[8]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        pq_tr<(J & 1) * PQ_STAGE>(f[u][0][0], addr[CB0 + u]);
        pq_tr<(J & 1) * PQ_STAGE + 512>(f[u][0][1], addr[CB0 + u]);
        pq_tr<(J & 1) * PQ_STAGE + PQ_PLANE>(f[u][1][0], addr[CB0 + u]);
        pq_tr<(J & 1) * PQ_STAGE + PQ_PLANE + 512>(f[u][1][1], addr[CB0 + u]);
    }
}
__device__ __forceinline__ void pq_mma_group(f32x4 *acc, const s16x4 (&f)[2][2][2], f16x8 ah, f16x8 al) {
    f16x8 bh[2], bl[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) { bh[u] = pq_cat(f[u][0][0], f[u][0][1]); bl[u] = pq_cat(f[u][1][0], f[u][1][1]); }
#pragma unroll
    for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[u], acc[u], 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[u], acc[u], 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[u], acc[u], 0, 0, 0);
}

__global__ void __launch_bounds__(256, 2)
pool_mfma_kernel(const _Float16 *__restrict__ x_hi, const _Float16 *__restrict__ x_lo, int64_t ld_x,
                 const int64_t *__restrict__ bu_off, const int32_t *__restrict__ bu_row,
                 const _Float16 *__restrict__ wa_hi, const _Float16 *__restrict__ wa_lo, int64_t nv, int64_t nblocks,
                 _Float16 *__restrict__ y_hi, _Float16 *__restrict__ y_lo, int64_t ld_y, float *__restrict__ y_f32,
                 int64_t ld_yf, int64_t per_xcd) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t lb = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);    // XCD-contiguous order
    const int64_t b = lb >> 2;
    const int col0 = (int)(lb & 3) * PQ_NC;
    if (b >= nblocks) return;
    const int64_t ub0 = bu_off[b];
    const int n = (int)((bu_off[b + 1] - ub0) / PM_KS);                            // steps (>= 1)
    const int64_t ks0 = ub0 / PM_KS;

    // ---- DMA roles: wave wv stages union rows 8wv..8wv+7 of a step; one instruction = 4 rows x 256 B.
    //      instruction i (0,1), lane quarter u = lane>>4 -> row 8wv + 2u + i, stored in LDS row slot
    //      8wv + 4i + u; lane chunk c = lane&15 lands in physical 16-B chunk c and fetches logical chunk
    //      c ^ 2t(row), t(row) = (row & 3) | ((row >> 3) & 1) << 2   (conflict-free transposed reads)
    const int du = lane >> 4, dc = lane & 15;
    const int64_t dsrc0 = col0 + ((dc ^ (2 * ((2 * (du & 1) + 0) | ((wv & 1) << 2)))) * 8);
    const int64_t dsrc1 = col0 + ((dc ^ (2 * ((2 * (du & 1) + 1) | ((wv & 1) << 2)))) * 8);
    const int32_t *idg = bu_row + ub0 + 8 * wv;                                    // this wave's row ids, step 0
    const _Float16 *wah = wa_hi + ((ks0 * PM_W + wv) * 64 + lane) * 8;
    const _Float16 *wal = wa_lo + ((ks0 * PM_W + wv) * 64 + lane) * 8;
    constexpr int64_t WSTEP = (int64_t)PM_W * 64 * 8;
    // stage k -> ring slot: rows(k), weights(k), ids(min(k+2, n-1))
    auto issue = [&](int2 id, int k, int slot) {
        unsigned char *dst = smem_raw + slot * PQ_STAGE;
        const int64_t s0 = (int64_t)id.x * ld_x + dsrc0, s1 = (int64_t)id.y * ld_x + dsrc1;
        glds16(x_hi + s0, dst + (8 * wv) * 256);
        glds16(x_lo + s0, dst + PQ_PLANE + (8 * wv) * 256);
        glds16(x_hi + s1, dst + (8 * wv) * 256 + 1024);
        glds16(x_lo + s1, dst + PQ_PLANE + (8 * wv) * 256 + 1024);
        glds16(wah + (int64_t)k * WSTEP, dst + PQ_OFF_W + wv * 1024);
        glds16(wal + (int64_t)k * WSTEP, dst + PQ_OFF_W + 4096 + wv * 1024);
        const int kid = k + 2 < n ? k + 2 : n - 1;
        // 16 lanes x 4 B: the wave's 8 ids (+ 8 more, always inside the block's padded union)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(idg + (int64_t)kid * PM_KS + (lane & 7)),
                                         (__attribute__((address_space(3))) void *)(dst + PQ_OFF_ID + wv * 256), 4, 0, 0);
    };

    // ---- read roles: 16-lane group g owns k rows 8g..8g+7; lane 4q+p supplies row 8g+q (LDS slot
    //      8g + 4(q&1) + (q>>1); the second read, row 8g+q+4, is 2 slots = 512 B further), logical
    //      columns 4p..4p+3 of the 16-column block
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem_raw;
    uint32_t addr[8];
    {
        const uint32_t rowb = (uint32_t)(8 * g + 4 * (q & 1) + (q >> 1)) * 256u + (uint32_t)((p >> 1) * 16 + (p & 1) * 8);
        const uint32_t t = (uint32_t)(q | ((g & 1) << 2));
#pragma unroll
        for (int k = 0; k < 8; ++k) addr[k] = lds0 + ((rowb + 32u * k) ^ (t << 5));
    }
    const uint32_t addr_w = lds0 + PQ_OFF_W + wv * 1024 + lane * 16;
    const uint32_t addr_id = lds0 + PQ_OFF_ID + wv * 256 + du * 8;
    uint32_t addr2[8];                                                  // the same, based at ring slot 2
#pragma unroll
    for (int k = 0; k < 8; ++k) addr2[k] = addr[k] + 2 * PQ_STAGE;
    const uint32_t addr_w2 = addr_w + 2 * PQ_STAGE, addr_id2 = addr_id + 2 * PQ_STAGE;

    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: stages 0 and 1 in flight (a one-step block stages its only step twice: no branches here,
    //      so that the compiler's own wait for the two id loads sits before the first DMA and nowhere else)
    {
        const int k1 = n > 1 ? 1 : 0;
        const int2 i0 = *reinterpret_cast<const int2 *>(idg + 2 * du);
        const int2 i1 = *reinterpret_cast<const int2 *>(idg + k1 * PM_KS + 2 * du);
        asm volatile("" ::"v"(i0.x), "v"(i0.y), "v"(i1.x), "v"(i1.y));    // both id loads land before the first DMA
        issue(i0, 0, 0);
        issue(i1, k1, 1);
        pq_handover<PQ_DMA_PER_STAGE>();
    }
    s16x4 f0[2][2][2], f1[2][2][2];
    f16x8 ah, al;
    int2 idn;
    for (int s0 = 0; s0 < n; s0 += PQ_NST) {
#define PQ_STEP(J, JO, A, AW, AID)                                                                              \
        if (s0 + J < n) {                                                                                       \
            const int s = s0 + J;                                                                               \
            pq_rd64<JO * PQ_STAGE>(idn, AID);                                                                   \
            pq_rd128<JO * PQ_STAGE>(ah, AW);                                                                    \
            pq_rd128<JO * PQ_STAGE + 4096>(al, AW);                                                             \
            pq_read_group<JO, 0>(f0, A);                                                                        \
            pq_wait_lgkm3<8>(idn, ah, al);                                                                      \
            if (s + 2 < n) issue(idn, s + 2, (J + 2) % PQ_NST);                                                 \
            pq_read_group<JO, 2>(f1, A);                                                                        \
            pq_wait_lgkm<8>(f0);                                                                                \
            pq_mma_group(acc, f0, ah, al);                                                                      \
            pq_read_group<JO, 4>(f0, A);                                                                        \
            pq_wait_lgkm<8>(f1);                                                                                \
            pq_mma_group(acc + 2, f1, ah, al);                                                                  \
            pq_read_group<JO, 6>(f1, A);                                                                        \
            pq_wait_lgkm<8>(f0);                                                                                \
            pq_mma_group(acc + 4, f0, ah, al);                                                                  \
            pq_wait_lgkm<0>(f1);                                                                                \
            pq_mma_group(acc + 6, f1, ah, al);                                                                  \
            if (s + 2 < n) pq_handover<PQ_DMA_PER_STAGE>(); else pq_handover<0>();                              \
        }
        PQ_STEP(0, 0, addr, addr_w, addr_id)
        PQ_STEP(1, 1, addr, addr_w, addr_id)
        PQ_STEP(2, 0, addr2, addr_w2, addr_id2)
#undef PQ_STEP
    }
    // ---- epilogue through LDS (the ring is drained: the last hand-over waited for vmcnt(0))
    constexpr float inv = 1.f / PM_WSCALE;
    float *st = reinterpret_cast<float *>(smem_raw) + wv * (16 * PQ_EP);
    const int fl = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int cb = 0; cb < 8; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) st[(fq * 4 + r) * PQ_EP + cb * 16 + fl] = acc[cb][r] * inv;
    gp_wave_sync();
    const int64_t row0 = b * PM_ROWS + wv * 16;
#pragma unroll
    for (int it = 0; it < 8; ++it) {                                    // 16 rows x 32 float4 per wave
        const int idx = it * 64 + lane;
        const int row = idx >> 5, c4 = idx & 31;
        const int64_t grow = row0 + row;
        if (grow < nv) {
            float4 v = *reinterpret_cast<const float4 *>(st + row * PQ_EP + c4 * 4);
            float xv[4] = {v.x, v.y, v.z, v.w};
            f16x4 h, l;
#pragma unroll
            for (int i = 0; i < 4; ++i) { h[i] = (_Float16)xv[i]; l[i] = (_Float16)(xv[i] - (float)h[i]); }
            if (y_hi) {
                *reinterpret_cast<f16x4 *>(y_hi + grow * ld_y + col0 + c4 * 4) = h;
                *reinterpret_cast<f16x4 *>(y_lo + grow * ld_y + col0 + c4 * 4) = l;
            }
            if (y_f32) *reinterpret_cast<float4 *>(y_f32 + grow * ld_yf + col0 + c4 * 4) = v;
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
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)PQ_SMEM_BYTES));
        attr_set = true;
    }
    int64_t nb = (nv + PM_ROWS - 1) / PM_ROWS;
    int64_t per_xcd = (nb * (PM_D / PQ_NC) + 7) / 8;
    pool_mfma_kernel<<<(unsigned)(per_xcd * 8), 256, PQ_SMEM_BYTES, gp_stream(stream_)>>>(
        static_cast<const _Float16 *>(x_hi), static_cast<const _Float16 *>(x_lo), ld_x, bu_off, bu_row,
        static_cast<const _Float16 *>(wa_hi), static_cast<const _Float16 *>(wa_lo), nv, nb, static_cast<_Float16 *>(y_hi),
        static_cast<_Float16 *>(y_lo), ld_y, y_f32, ld_yf, per_xcd);
    GP_CHECK_LAUNCH();
    return GP_OK;
}
