// Row 9: submanifold sparse 3D convolution (MinkowskiConvolution, stride 1, no bias) as an
// output-tile-stationary gather-GEMM with exact fp32 MFMA (v_mfma_f32_16x16x4_f32).
//
// One 512-thread workgroup owns a tile of BM=128 Morton-contiguous output voxels x BN=128 output
// channels.  For every kernel offset k the (input row, output row) pairs of the tile are compacted
// in LDS (groups of 16 pairs = one MFMA row block), so MFMA work is spent only on existing
// neighbours.  Per (k, 32-channel slab) step the gathered input rows (A, k-major in LDS) and the
// weight slab W[k][c0:c0+32][n0:n0+128] (B) are staged through registers into a double-buffered
// LDS ring while the previous step's MFMAs run; each wave owns 16 output channels and keeps one
// 16x16 accumulator per pair group.  After the Cin reduction of an offset, the group accumulators
// are added into the fp32 tile accumulator in LDS at their output rows (waves own disjoint
// columns: no atomics, no barrier).  Epilogue fuses BatchNorm(eval) scale/shift, residual and ReLU.
#include "gp_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int NTHREADS = 512, NWAVES = 8;
constexpr int MAXG = BM / 16;          // pair groups per offset
constexpr int PITCH = BM + 16;         // LDS row pitch (floats) for A (k-major) and B: == 16 mod 32 banks
constexpr int KV_MAX = 27;

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvSmem {
    float acc[BM][BN];                 // 64 KiB tile accumulator
    float a[2][BK][PITCH];             // 2 x 18 KiB gathered inputs, k-major
    float b[2][BK][PITCH];             // 2 x 18 KiB weight slab
    int pin[KV_MAX][BM];               // compacted input rows per offset (-1 = padding)
    unsigned char pout[KV_MAX][BM];    // matching local output rows
    int pcount[KV_MAX + 1];
};

template <int NG>
__device__ __forceinline__ void mma_step(const float (*__restrict__ a)[PITCH], const float (*__restrict__ b)[PITCH],
                                         f32x4 (&acc)[MAXG], int wv, int fl, int fq) {
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
        float bf = b[kk * 4 + fq][wv * 16 + fl];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float af = a[kk * 4 + fq][g * 16 + fl];
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[g], 0, 0, 0);
        }
    }
}

__global__ void __launch_bounds__(NTHREADS)
sparse_conv_kernel(const float *__restrict__ x, int64_t ld_x, const int32_t *__restrict__ nbr_map, int64_t nv,
                   const float *__restrict__ w, int kv, int cin, int cout, const float *__restrict__ scale,
                   const float *__restrict__ shift, const float *__restrict__ residual, int64_t ld_res, int relu,
                   float *__restrict__ y, int64_t ld_y, int n_tiles, int m_tiles) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    ConvSmem &sm = *reinterpret_cast<ConvSmem *>(smem_raw);

    // XCD-aware tile order: blocks b and b+8 share an XCD; give every XCD one channel tile so that
    // its workgroups stream the same weight slabs through their L2 at about the same time.
    const int b = blockIdx.x;
    int xcd = b & 7, within = b >> 3;
    int nt, mt;
    if (n_tiles <= 8 && (8 % n_tiles) == 0) {
        int per = 8 / n_tiles;                       // XCDs per channel tile
        nt = xcd % n_tiles;
        mt = within * per + xcd / n_tiles;
    } else {
        nt = b % n_tiles;
        mt = b / n_tiles;
    }
    if (mt >= m_tiles) return;
    const int64_t row0 = (int64_t)mt * BM;
    const int n0 = nt * BN;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    // ---- zero the tile accumulator, build the compacted pair lists
    for (int i = tid; i < BM * BN / 4; i += NTHREADS) reinterpret_cast<float4 *>(&sm.acc[0][0])[i] = make_float4(0, 0, 0, 0);
    for (int k = wv; k < kv; k += NWAVES) {
        int base = 0;
        for (int half = 0; half < BM / 64; ++half) {
            int r = half * 64 + lane;
            int64_t row = row0 + r;
            int in = -1;
            if (row < nv) in = nbr_map ? nbr_map[(int64_t)k * nv + row] : (int)row;
            unsigned long long m = __ballot(in >= 0);
            int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            if (in >= 0) { sm.pin[k][pos] = in; sm.pout[k][pos] = (unsigned char)r; }
            base += __popcll(m);
        }
        int padded = (base + 15) & ~15;
        for (int p = base + lane; p < padded; p += 64) { sm.pin[k][p] = -1; sm.pout[k][p] = 255; }
        if (lane == 0) sm.pcount[k] = base;
    }
    __syncthreads();

    const int csteps = cin / BK;
    // staging registers: A = 2 float4 (pair = tid%128, k-quads tid/128 and tid/128+4), B = 2 float4
    float4 ra0, ra1, rb0, rb1;
    const int a_pair = tid & (BM - 1), a_kq = tid >> 7;          // 0..3
    const int b_row = tid >> 5, b_c4 = tid & 31;                 // rows b_row, b_row+16

    auto next_k = [&](int k) { while (k < kv && sm.pcount[k] == 0) ++k; return k; };

    auto load_step = [&](int k, int cs) {
        int np = (sm.pcount[k] + 15) & ~15;
        int in = (a_pair < np) ? sm.pin[k][a_pair] : -1;
        const float *xr = x + (int64_t)(in < 0 ? 0 : in) * ld_x + cs * BK;
        bool ok = in >= 0;
        ra0 = ok ? *reinterpret_cast<const float4 *>(xr + a_kq * 4) : make_float4(0, 0, 0, 0);
        ra1 = ok ? *reinterpret_cast<const float4 *>(xr + (a_kq + 4) * 4) : make_float4(0, 0, 0, 0);
        const float *wr = w + ((int64_t)k * cin + cs * BK) * cout + n0;
        rb0 = *reinterpret_cast<const float4 *>(wr + (int64_t)b_row * cout + b_c4 * 4);
        rb1 = *reinterpret_cast<const float4 *>(wr + (int64_t)(b_row + 16) * cout + b_c4 * 4);
    };
    auto store_step = [&](int buf) {
        int kq = a_kq * 4;
        sm.a[buf][kq + 0][a_pair] = ra0.x;
        sm.a[buf][kq + 1][a_pair] = ra0.y;
        sm.a[buf][kq + 2][a_pair] = ra0.z;
        sm.a[buf][kq + 3][a_pair] = ra0.w;
        sm.a[buf][kq + 16][a_pair] = ra1.x;
        sm.a[buf][kq + 17][a_pair] = ra1.y;
        sm.a[buf][kq + 18][a_pair] = ra1.z;
        sm.a[buf][kq + 19][a_pair] = ra1.w;
        *reinterpret_cast<float4 *>(&sm.b[buf][b_row][b_c4 * 4]) = rb0;
        *reinterpret_cast<float4 *>(&sm.b[buf][b_row + 16][b_c4 * 4]) = rb1;
    };

    f32x4 acc[MAXG];
#pragma unroll
    for (int g = 0; g < MAXG; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};

    int k = next_k(0);
    int cs = 0, buf = 0;
    if (k < kv) {
        load_step(k, 0);
        store_step(0);
    }
    __syncthreads();
    const int fl = lane & 15, fq = lane >> 4;
    while (k < kv) {
        // where is the next step?
        int nk = k, ncs = cs + 1;
        if (ncs == csteps) { ncs = 0; nk = next_k(k + 1); }
        if (nk < kv) load_step(nk, ncs);                      // global loads in flight during the MFMAs
        const int ng = (sm.pcount[k] + 15) >> 4;
        switch (ng) {
            case 1: mma_step<1>(sm.a[buf], sm.b[buf], acc, wv, fl, fq); break;
            case 2: mma_step<2>(sm.a[buf], sm.b[buf], acc, wv, fl, fq); break;
            case 3: mma_step<3>(sm.a[buf], sm.b[buf], acc, wv, fl, fq); break;
            case 4: mma_step<4>(sm.a[buf], sm.b[buf], acc, wv, fl, fq); break;
            case 5: mma_step<5>(sm.a[buf], sm.b[buf], acc, wv, fl, fq); break;
            case 6: mma_step<6>(sm.a[buf], sm.b[buf], acc, wv, fl, fq); break;
            case 7: mma_step<7>(sm.a[buf], sm.b[buf], acc, wv, fl, fq); break;
            default: mma_step<8>(sm.a[buf], sm.b[buf], acc, wv, fl, fq); break;
        }
        if (cs == csteps - 1) {
            // offset finished: add the group accumulators into the tile accumulator at their output rows
#pragma unroll
            for (int g = 0; g < MAXG; ++g) {
                if (g < ng) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int o = sm.pout[k][g * 16 + fq * 4 + r];
                        if (o != 255) sm.acc[o][wv * 16 + fl] += acc[g][r];
                    }
                    acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        if (nk < kv) store_step(buf ^ 1);
        __syncthreads();
        buf ^= 1;
        k = nk;
        cs = ncs;
    }

    // ---- epilogue: BN(eval) scale/shift, residual, ReLU.  Each wave owns 16 columns.
    const int col = n0 + wv * 16 + fl;
    const float sc = scale ? scale[col] : 1.f, sh = shift ? shift[col] : 0.f;
    for (int r = fq; r < BM; r += 4) {
        int64_t row = row0 + r;
        if (row >= nv) break;
        float v = sm.acc[r][wv * 16 + fl];
        v = v * sc + sh;
        if (residual) v += residual[row * ld_res + col];
        if (relu) v = fmaxf(v, 0.f);
        y[row * ld_y + col] = v;
    }
}

}  // namespace

extern "C" int gp_sparse_conv(const float *x, int64_t ld_x, const int32_t *nbr_map, int64_t nv, const float *w,
                              int32_t kv, int32_t cin, int32_t cout, const float *scale, const float *shift,
                              const float *residual, int64_t ld_res, int32_t relu, float *y, int64_t ld_y,
                              void *stream_) {
    GP_CHECK_ARG(x && w && y && nv > 0, "gp_sparse_conv: null/empty argument");
    GP_CHECK_ARG(kv == 27 || kv == 1, "gp_sparse_conv: kv=%d (27 or 1)", kv);
    GP_CHECK_ARG(kv == 1 || nbr_map, "gp_sparse_conv: nbr_map required for kv=27");
    GP_CHECK_ARG(cin > 0 && cin % BK == 0, "gp_sparse_conv: cin=%d must be a multiple of %d (pad with zero channels)", cin, BK);
    GP_CHECK_ARG(cout > 0 && cout % BN == 0, "gp_sparse_conv: cout=%d must be a multiple of %d", cout, BN);
    GP_CHECK_ARG(ld_x % 4 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)w % 16 == 0, "gp_sparse_conv: x/w rows must be 16-byte aligned");
    GP_CHECK_ARG(x != y, "gp_sparse_conv: x and y must not alias");
    GP_SMEM_ATTR(sparse_conv_kernel, sizeof(ConvSmem));
    int m_tiles = (int)((nv + BM - 1) / BM), n_tiles = cout / BN;
    int blocks;
    if (n_tiles <= 8 && (8 % n_tiles) == 0) {
        int per = 8 / n_tiles;
        blocks = ((m_tiles + per - 1) / per) * 8;
    } else {
        blocks = m_tiles * n_tiles;
    }
    sparse_conv_kernel<<<blocks, NTHREADS, sizeof(ConvSmem), gp_stream(stream_)>>>(
        x, ld_x, kv == 1 ? nullptr : nbr_map, nv, w, kv, cin, cout, scale, shift, residual, ld_res, relu, y, ld_y,
        n_tiles, m_tiles);
    GP_CHECK_LAUNCH();
    return GP_OK;
}
