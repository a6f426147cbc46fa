// Round 5, pricing a structure ON THE GPU before building it: an OUTPUT-STATIONARY sparse convolution (a workgroup owns 256 output rows
// x 256 output columns, keeps their sums in registers and walks the 27 offsets; per offset the rows that have that neighbour are
// compacted, so no partial rows ever leave the CU) moves, per 512 -> 512 layer of the S scene, 14.5 GB of weight tiles and 4 GB of
// gathered rows through L2 -> LDS (the two-phase kernel: 3.9 + 4.0 GB, plus 2 x 2 GB of fp32 partial rows through the fabric).
// This program runs ONLY that data movement -- the LDS-DMA of every (offset, 32-channel step) stage, double-buffered, one barrier per
// step, no fragment reads -- optionally with the step's MFMAs issued on register operands beside it, on a synthetic kernel map with
// the scene's statistics (27 % of the (row, offset) pairs present, centre offset full, neighbours a few hundred rows away in a
// Morton-like order).  What it answers: does the load side of that structure fit under the 1.9 ms the two-phase layer takes?
// build: hipcc -O3 --offload-arch=gfx950 -o conv_os_loads conv_os_loads.hip ; run: ./conv_os_loads [nv] [mfma 0/1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int C = 512, KV = 27, TR = 256, TN = 256, TK = 32;
constexpr int A_PLANE = TR * 64, B_PLANE = TN * 64;            // bytes per stage and plane
constexpr int STAGE = 2 * A_PLANE + 2 * B_PLANE;               // 64 KiB

__device__ __forceinline__ void glds16(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void handover() { asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory"); }

template <bool MFMA>
__global__ void __launch_bounds__(512)
os_loads_kernel(const _Float16 *__restrict__ x_hi, const _Float16 *__restrict__ x_lo, const _Float16 *__restrict__ w_hi,
                const _Float16 *__restrict__ w_lo, const int *__restrict__ tk_off /*[tiles][28]*/, const int *__restrict__ in_rows,
                int nblocks, float *__restrict__ sink) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long nb = gridDim.x, per_xcd = nb >> 3;
    const long lb = (long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int b = (int)(lb >> 1), n0 = (int)(lb & 1) * TN;
    if (b >= nblocks) return;
    const int lrow = lane >> 2, q = (lane & 3) * 8;
    f32x4 acc[16];
    f16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(0.001f * lane + i); fb[i] = (_Float16)(0.5f * i); }
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    const int *off = tk_off + (long)b * 28;
    // stage (k, s): B = W_k[n0 .. n0+255][32 s .. +32] (every wave: 2 x 16 rows per plane), A = the offset's compacted rows (wave wv:
    // row tiles wv, wv + 8 of the ceil(n_k / 16))
    // (the row ids of an offset are loaded once per offset, into registers: a compiler-visible load inside the loop waits for every
    //  outstanding LDS-DMA -- here that drain happens once per 16 steps)
    int in0 = 0, in1 = 0;
    auto load_ids = [&](int p0, int mk) {
        in0 = wv < mk ? in_rows[p0 + wv * 16 + lrow] : 0;
        in1 = wv + 8 < mk ? in_rows[p0 + (wv + 8) * 16 + lrow] : 0;
    };
    auto issue = [&](int k, int s, int buf, int mk) -> int {
        unsigned char *st = smem + buf * STAGE;
        int n = 0;
        for (int t = 0; t < 2; ++t) {
            const int row = wv * 32 + t * 16 + lrow;
            const long wo = ((long)k * C + n0 + row) * C + s * TK + q;
            glds16(w_hi + wo, st + 2 * A_PLANE + (wv * 32 + t * 16) * 64);
            glds16(w_lo + wo, st + 2 * A_PLANE + B_PLANE + (wv * 32 + t * 16) * 64);
            n += 2;
        }
        if (wv < mk) {
            const long xo = (long)in0 * C + s * TK + q;
            glds16(x_hi + xo, st + wv * 16 * 64);
            glds16(x_lo + xo, st + A_PLANE + wv * 16 * 64);
            n += 2;
        }
        if (wv + 8 < mk) {
            const long xo = (long)in1 * C + s * TK + q;
            glds16(x_hi + xo, st + (wv + 8) * 16 * 64);
            glds16(x_lo + xo, st + A_PLANE + (wv + 8) * 16 * 64);
            n += 2;
        }
        return n;
    };
    int k = 0, s = 0, buf = 0;
    int mk = (off[1] - off[0]) >> 4;
    load_ids(off[0], mk);
    issue(k, s, buf, mk);
    const int total = KV * (C / TK);
    for (int it = 0; it < total; ++it) {
        int k2 = k, s2 = s + 1;
        if (s2 == C / TK) { s2 = 0; ++k2; }
        int m2 = mk, n2 = 0;
        if (it + 1 < total) {
            if (k2 != k) { const int p2 = off[k2]; m2 = (off[k2 + 1] - p2) >> 4; load_ids(p2, m2); }
            n2 = issue(k2, s2, buf ^ 1, m2);
        }
        if (MFMA) {                                   // the step's matrix work on register operands: 16 column tiles x mk row tiles x 3 / 8 waves
            const int nm = (mk * 16 * 3 + 7) / 8;
            for (int i = 0; i < nm; i += 4) {        // (static accumulator indices: a run-time index turns into register-indexed moves)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[j], 0, 0, 0);
            }
        }
        // the older stage has landed when at most n2 operations are outstanding (in-order completion)
        if (n2 == 0) handover<0>(); else if (n2 == 4) handover<4>(); else if (n2 == 6) handover<6>(); else handover<8>();
        k = k2; s = s2; mk = m2; buf ^= 1;
    }
    float r = 0;
    for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (r == 12345.678f) sink[blockIdx.x] = r + smem[tid];
}

int main(int argc, char **argv) {
    const int nv = argc > 1 ? atoi(argv[1]) : 133933;
    const int nblocks = (nv + TR - 1) / TR;
    std::mt19937 rng(7);
    std::vector<int> tk_off((size_t)nblocks * 28), rows;
    std::uniform_real_distribution<float> u01(0.f, 1.f);
    long pairs = 0, padded = 0;
    for (int b = 0; b < nblocks; ++b) {
        for (int k = 0; k < KV; ++k) {
            tk_off[(size_t)b * 28 + k] = (int)rows.size();
            const int shift = (k - 13) * 37 + ((k * 7919) % 211) - 105;              // a few hundred rows away, offset-dependent
            int n = 0;
            for (int r = 0; r < TR && b * TR + r < nv; ++r) {
                const bool have = k == 13 || u01(rng) < 0.2425f;                        // 6.31 / 26 of the non-centre offsets
                if (!have) continue;
                int in = b * TR + r + shift + (int)(u01(rng) * 64) - 32;
                in = std::min(std::max(in, 0), nv - 1);
                rows.push_back(in);
                ++n;
            }
            pairs += n;
            while (n % 16) { rows.push_back(rows.back()); ++n; }
            padded += n;
        }
        tk_off[(size_t)b * 28 + KV] = (int)rows.size();
    }
    printf("nv %d, row blocks %d, pairs %ld (%.2f per row), padded to 16-row tiles %ld (x %.3f)\n", nv, nblocks, pairs, (double)pairs / nv, padded,
           (double)padded / pairs);
    _Float16 *xh, *xl, *wh, *wl;
    int *d_off, *d_rows;
    float *sink;
    hipMalloc(&xh, (size_t)nv * C * 2); hipMalloc(&xl, (size_t)nv * C * 2);
    hipMalloc(&wh, (size_t)KV * C * C * 2); hipMalloc(&wl, (size_t)KV * C * C * 2);
    hipMemset(xh, 0x11, (size_t)nv * C * 2); hipMemset(xl, 0x12, (size_t)nv * C * 2);
    hipMemset(wh, 0x13, (size_t)KV * C * C * 2); hipMemset(wl, 0x14, (size_t)KV * C * C * 2);
    hipMalloc(&d_off, tk_off.size() * 4); hipMalloc(&d_rows, rows.size() * 4); hipMalloc(&sink, 1 << 20);
    hipMemcpy(d_off, tk_off.data(), tk_off.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice);
    const long per_xcd = ((long)nblocks * 2 + 7) / 8;
    const unsigned grid = (unsigned)(per_xcd * 8);
    hipFuncSetAttribute((const void *)os_loads_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
    hipFuncSetAttribute((const void *)os_loads_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const double gb_w = (double)nblocks * 2 * KV * (C / TK) * 2 * B_PLANE / 1e9, gb_a = (double)padded * 2 * C * 2 * 2 / 1e9;
    printf("per layer: weight tiles %.2f GB + gathered rows %.2f GB through L2 -> LDS, %u workgroups of 512 threads, %d KiB of LDS\n", gb_w, gb_a, grid,
           2 * STAGE / 1024);
    for (int mf = 0; mf < 2; ++mf)
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            if (mf) os_loads_kernel<true><<<grid, 512, 2 * STAGE>>>(xh, xl, wh, wl, d_off, d_rows, nblocks, sink);
            else os_loads_kernel<false><<<grid, 512, 2 * STAGE>>>(xh, xl, wh, wl, d_off, d_rows, nblocks, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%s  %.3f ms per layer  (%.1f TB/s into LDS)%s\n", mf ? "loads + the steps' MFMAs on register operands" : "loads only                                   ",
                   ms, (gb_w + gb_a) / ms, hipGetLastError() == hipSuccess ? "" : "  LAUNCH ERROR");
        }
    return 0;
}
