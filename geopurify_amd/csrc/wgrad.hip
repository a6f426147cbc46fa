// weight gradient of the submanifold 3x3x3 convolution on the matrix cores.
//   dW[k][ci][co] = sum over the pairs p of offset k of  X[in_p][ci] * dY[out_p][co]
// A GEMM whose reduction dimension is the pair list: both operands are GATHERED ROWS (ci / co contiguous), i.e. both
// are K-major for the MFMA, so both fragments come out of row-major LDS images through ds_read_b64_tr_b16.
// One 512-thread workgroup owns a 256 x 256 tile of dW[k] and a segment of the pair list of offset k; per step it
// stages 32 pairs (32 rows x 256 columns x {hi, lo} of X and of dY = 64 KiB) by LDS-DMA into a two-stage ring while
// the 8 waves (4 x 2, 64 x 128 outputs each) run hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16 (96 MFMAs per
// wave and step: long enough that plain double buffering hides the gather).  Segment tiles go to a partial buffer
// and a second kernel sums them in fixed order (deterministic) and undoes the power-of-two scaling of dY.
// Same staging tricks as pool_mfma.hip: XOR-swizzled rows through the DMA source addresses (conflict-free
// transposed reads), inline-asm LDS reads with explicit lgkmcnt waits (compiler-visible reads would wait for the
// DMA in flight), `s_waitcnt vmcnt(0); s_barrier` hand-over.
#include "gp_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((vector_size(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WG_T = 256;                      // tile edge (rows of dW = ci, columns = co)
constexpr int WG_KS = 32;                      // pairs per step
constexpr int WG_RB = WG_T * 2;                // bytes per staged row and plane
constexpr int WG_PLANE = WG_KS * WG_RB;        // 16 KiB
constexpr int WG_STAGE = 4 * WG_PLANE;         // X_hi | X_lo | Y_hi | Y_lo
constexpr size_t WG_SMEM = 2 * (size_t)WG_STAGE;

__device__ __forceinline__ void wg_glds16(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}
template <int OFF>
__device__ __forceinline__ void wg_tr(s16x4 &d, uint32_t addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
// fragment = 2 transposed reads (k rows 8g+q and 8g+q+4 -> 2048 bytes apart)
struct WgFrag { s16x4 a, b; };
template <int OFF>
__device__ __forceinline__ void wg_read(WgFrag &f, uint32_t addr) {
    wg_tr<OFF>(f.a, addr);
    wg_tr<OFF + 4 * WG_RB>(f.b, addr);
}
template <int N>
__device__ __forceinline__ void wg_wait4(WgFrag &f0, WgFrag &f1, WgFrag &f2, WgFrag &f3) {
    asm volatile("s_waitcnt lgkmcnt(%[n])"
                 : "+v"(f0.a), "+v"(f0.b), "+v"(f1.a), "+v"(f1.b), "+v"(f2.a), "+v"(f2.b), "+v"(f3.a), "+v"(f3.b)
                 : [n] "n"(N));
}
__device__ __forceinline__ void wg_handover() { asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ f16x8 wg_cat(const WgFrag &f) {
    typedef short s16x8 __attribute__((vector_size(16)));
    s16x8 v = __builtin_shufflevector(f.a, f.b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(f16x8, v);
}

// segs[s] = {offset k, first step (units of 32 pairs in the padded pair arrays), number of steps, 0}
__global__ void __launch_bounds__(512, 1)
wgrad_kernel(const _Float16 *__restrict__ x_hi, const _Float16 *__restrict__ x_lo, int64_t ld_x, const _Float16 *__restrict__ y_hi,
             const _Float16 *__restrict__ y_lo, int64_t ld_y, const int32_t *__restrict__ pin, const int32_t *__restrict__ pout,
             const int4 *__restrict__ segs, int ntiles, int ntn, int cin_pad, float *__restrict__ part) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int seg = blockIdx.x / ntiles, tile = blockIdx.x % ntiles;
    const int mt = tile / ntn, nt = tile % ntn;
    const int m0 = mt * WG_T < cin_pad - WG_T ? mt * WG_T : cin_pad - WG_T;     // last row tile slides back (overlap, masked at the store)
    const int n0 = nt * WG_T;
    const int4 sd = segs[seg];
    const int64_t g0 = sd.y;
    const int n = sd.z;

    // ---- DMA roles: wave wv stages pairs 4wv..4wv+3 of the step; instruction i (0,1) = pairs 4wv+2i (+1 for the
    //      upper half of the lanes); lane chunk c lands in physical 16-byte chunk c, fetches logical chunk c ^ 2t(row)
    const int dh = lane >> 5, dc = lane & 31;
    int64_t xoff[2], yoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 4 * wv + 2 * i + dh;
        const int t = (r & 3) | (((r >> 3) & 1) << 2);
        xoff[i] = m0 + ((dc ^ (2 * t)) * 8);
        yoff[i] = n0 + ((dc ^ (2 * t)) * 8);
    }
    const int32_t *pin_l = pin + g0 * WG_KS + 4 * wv + dh;
    const int32_t *pout_l = pout + g0 * WG_KS + 4 * wv + dh;
    auto load_ids = [&](int s, int (&ix)[2], int (&iy)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) { ix[i] = pin_l[(int64_t)s * WG_KS + 2 * i]; iy[i] = pout_l[(int64_t)s * WG_KS + 2 * i]; }
    };
    auto issue = [&](const int (&ix)[2], const int (&iy)[2], int slot) {
        unsigned char *dst = smem_raw + slot * WG_STAGE + (4 * wv) * WG_RB;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int64_t sx = (int64_t)ix[i] * ld_x + xoff[i], sy = (int64_t)iy[i] * ld_y + yoff[i];
            wg_glds16(x_hi + sx, dst + i * 2 * WG_RB);
            wg_glds16(x_lo + sx, dst + WG_PLANE + i * 2 * WG_RB);
            wg_glds16(y_hi + sy, dst + 2 * WG_PLANE + i * 2 * WG_RB);
            wg_glds16(y_lo + sy, dst + 3 * WG_PLANE + i * 2 * WG_RB);
        }
    };

    // ---- read roles: 16-lane group g owns k rows 8g..8g+7; lane 4q+p supplies row 8g+q (second read: +4 rows),
    //      logical columns 4p..4p+3 of a 16-column block
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem_raw;
    const int wm = wv >> 1, wn = wv & 1;                  // wave tile: rows 64 wm.., columns 128 wn..
    uint32_t addr[8];
    {
        const uint32_t rowb = (uint32_t)(8 * g + q) * WG_RB + (uint32_t)((p >> 1) * 16 + (p & 1) * 8);
        const uint32_t t = (uint32_t)(q | ((g & 1) << 2));
#pragma unroll
        for (int k = 0; k < 8; ++k) addr[k] = lds0 + ((rowb + 32u * k) ^ (t << 5));
    }
    // A (X image) column blocks 4 wm + i, i < 4: blocks 4wm..4wm+3 share the 128-column half (4wm)>>3
    const uint32_t a_half = (uint32_t)((4 * wm) >> 3) * 256u;
    const int a_k0 = (4 * wm) & 7;                        // 0 or 4
    const uint32_t b_half = (uint32_t)wn * 256u;          // B (dY image) column blocks 8 wn + j, j < 8

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int ix[2], iy[2], ixn[2] = {0, 0}, iyn[2] = {0, 0};
    load_ids(0, ix, iy);
    issue(ix, iy, 0);
    if (n > 1) load_ids(1, ixn, iyn);
    wg_handover();
    for (int s0 = 0; s0 < n; s0 += 2) {
#pragma unroll
        for (int J = 0; J < 2; ++J) {
            const int s = s0 + J;
            if (s < n) {
                if (s + 1 < n) {
                    issue(ixn, iyn, J ^ 1);
                    if (s + 2 < n) load_ids(s + 2, ixn, iyn);
                }
                const uint32_t so = J * WG_STAGE;
                WgFrag ah[4], al[4], b0h[2], b0l[2], b1h[2], b1l[2];
                // A fragments (held for the whole step): two groups of 8 reads
                wg_read<0>(ah[0], addr[a_k0 + 0] + a_half + so); wg_read<WG_PLANE>(al[0], addr[a_k0 + 0] + a_half + so);
                wg_read<0>(ah[1], addr[a_k0 + 1] + a_half + so); wg_read<WG_PLANE>(al[1], addr[a_k0 + 1] + a_half + so);
                wg_read<0>(ah[2], addr[a_k0 + 2] + a_half + so); wg_read<WG_PLANE>(al[2], addr[a_k0 + 2] + a_half + so);
                wg_read<0>(ah[3], addr[a_k0 + 3] + a_half + so); wg_read<WG_PLANE>(al[3], addr[a_k0 + 3] + a_half + so);
                wg_wait4<8>(ah[0], al[0], ah[1], al[1]);
                // B fragments in groups of 2 column blocks (8 reads), the next group in flight under the MFMAs
#define WG_READ_B(BH, BL, J0)                                                                                          \
                wg_read<2 * WG_PLANE>(BH[0], addr[(J0) & 7] + b_half + so); wg_read<3 * WG_PLANE>(BL[0], addr[(J0) & 7] + b_half + so); \
                wg_read<2 * WG_PLANE>(BH[1], addr[((J0) + 1) & 7] + b_half + so); wg_read<3 * WG_PLANE>(BL[1], addr[((J0) + 1) & 7] + b_half + so);
#define WG_MMA(BH, BL, J0)                                                                                             \
                {                                                                                                      \
                    f16x8 bh0 = wg_cat(BH[0]), bh1 = wg_cat(BH[1]), bl0 = wg_cat(BL[0]), bl1 = wg_cat(BL[1]);        \
                    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                    \
                        f16x8 ahv = wg_cat(ah[i]), alv = wg_cat(al[i]);                                                \
                        acc[i][J0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahv, bh0, acc[i][J0], 0, 0, 0);            \
                        acc[i][(J0) + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahv, bh1, acc[i][(J0) + 1], 0, 0, 0); \
                        acc[i][J0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahv, bl0, acc[i][J0], 0, 0, 0);            \
                        acc[i][(J0) + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahv, bl1, acc[i][(J0) + 1], 0, 0, 0); \
                        acc[i][J0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alv, bh0, acc[i][J0], 0, 0, 0);            \
                        acc[i][(J0) + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alv, bh1, acc[i][(J0) + 1], 0, 0, 0); \
                    }                                                                                                  \
                }
                WG_READ_B(b0h, b0l, 0)
                wg_wait4<8>(ah[2], al[2], ah[3], al[3]);
                WG_READ_B(b1h, b1l, 2)
                wg_wait4<8>(b0h[0], b0l[0], b0h[1], b0l[1]);
                WG_MMA(b0h, b0l, 0)
                WG_READ_B(b0h, b0l, 4)
                wg_wait4<8>(b1h[0], b1l[0], b1h[1], b1l[1]);
                WG_MMA(b1h, b1l, 2)
                WG_READ_B(b1h, b1l, 6)
                wg_wait4<8>(b0h[0], b0l[0], b0h[1], b0l[1]);
                WG_MMA(b0h, b0l, 4)
                wg_wait4<0>(b1h[0], b1l[0], b1h[1], b1l[1]);
                WG_MMA(b1h, b1l, 6)
#undef WG_READ_B
#undef WG_MMA
                wg_handover();
            }
        }
    }
    // ---- store the segment tile (rows below mt*256 of a slid-back last tile belong to the previous tile)
    float *out = part + (int64_t)blockIdx.x * WG_T * WG_T;
    const int fl = lane & 15, fq = lane >> 4;
    const int row_lo = mt * WG_T - m0;                    // first valid local row
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 64 * wm + 16 * i + 4 * fq + r;
            if (row >= row_lo) {
#pragma unroll
                for (int j = 0; j < 8; ++j) out[(int64_t)row * WG_T + 128 * wn + 16 * j + fl] = acc[i][j][r];
            }
        }
}

// dW[k][r][c] = inv_scale[0] * sum over the segments of offset k of their tile values (fixed order)
__global__ void wgrad_reduce_kernel(const float *__restrict__ part, const int32_t *__restrict__ seg_off, int kv, int ntiles, int ntn,
                                    int cin_pad, int cin_out, int cout, const float *__restrict__ inv_scale, float *__restrict__ dw) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    int64_t total = (int64_t)kv * cin_out * cout;
    if (i >= total) return;
    int c = (int)(i % cout);
    int r = (int)((i / cout) % cin_out);
    int k = (int)(i / ((int64_t)cout * cin_out));
    int mt = r / WG_T;
    int m0 = mt * WG_T < cin_pad - WG_T ? mt * WG_T : cin_pad - WG_T;
    int nt = c / WG_T;
    int64_t loc = (int64_t)(r - m0) * WG_T + (c - nt * WG_T);
    float s = 0.f;
    for (int sg = seg_off[k]; sg < seg_off[k + 1]; ++sg) s += part[((int64_t)sg * ntiles + mt * ntn + nt) * WG_T * WG_T + loc];
    dw[i] = s * (inv_scale ? inv_scale[0] : 1.f);
}

}  // namespace

extern "C" size_t gp_conv_wgrad_workspace_bytes(int64_t num_segments, int32_t cin_pad, int32_t cout) {
    int64_t ntiles = (int64_t)((cin_pad + WG_T - 1) / WG_T) * (cout / WG_T);
    return gp_align_up((size_t)(num_segments * ntiles) * WG_T * WG_T * sizeof(float), 256);
}

// x_hi/x_lo f16 [nv, ld_x >= cin_pad], y_hi/y_lo f16 [nv+1, ld_y >= cout] (row nv all zero: target of padded pairs);
// pair_in / pair_out i32: per offset the (input row, output row) pairs, each offset's list padded to a multiple of
// 32 with (0, nv); segs i32 [num_segments,4] = {offset, first step, steps, 0}, ordered by offset;
// seg_off i32 [kv+1] = first segment of each offset.  dw f32 [kv, cin_out, cout] (cin_out <= cin_pad rows written),
// multiplied by inv_scale[0] (device scalar, nullable).
extern "C" int gp_conv_wgrad_f16x3(const void *x_hi, const void *x_lo, int64_t ld_x, const void *y_hi, const void *y_lo, int64_t ld_y,
                                   const int32_t *pair_in, const int32_t *pair_out, const int32_t *segs, int64_t num_segments,
                                   const int32_t *seg_off, int32_t kv, int32_t cin_pad, int32_t cin_out, int32_t cout,
                                   const float *inv_scale, float *dw, void *workspace, size_t workspace_bytes, void *stream_) {
    GP_CHECK_ARG(x_hi && x_lo && y_hi && y_lo && pair_in && pair_out && segs && seg_off && dw && workspace,
                 "gp_conv_wgrad_f16x3: null argument");
    GP_CHECK_ARG(num_segments > 0 && kv > 0, "gp_conv_wgrad_f16x3: empty");
    GP_CHECK_ARG(cin_pad >= WG_T && cin_pad % 8 == 0 && cout % WG_T == 0 && cin_out <= cin_pad && cin_out > 0,
                 "gp_conv_wgrad_f16x3: cin_pad=%d (>= 256, multiple of 8), cout=%d (multiple of 256)", cin_pad, cout);
    GP_CHECK_ARG(ld_x % 8 == 0 && ld_y % 8 == 0 && ld_x >= cin_pad && ld_y >= cout, "gp_conv_wgrad_f16x3: rows must be 16-byte aligned");
    if (workspace_bytes < gp_conv_wgrad_workspace_bytes(num_segments, cin_pad, cout)) { gp_set_error("gp_conv_wgrad_f16x3: workspace too small"); return GP_ENOMEM; }
    hipStream_t s = gp_stream(stream_);
    GP_SMEM_ATTR(wgrad_kernel, WG_SMEM);
    const int ntn = cout / WG_T, ntm = (cin_pad + WG_T - 1) / WG_T, ntiles = ntn * ntm;
    float *part = static_cast<float *>(workspace);
    wgrad_kernel<<<(unsigned)(num_segments * ntiles), 512, WG_SMEM, s>>>(
        static_cast<const _Float16 *>(x_hi), static_cast<const _Float16 *>(x_lo), ld_x, static_cast<const _Float16 *>(y_hi),
        static_cast<const _Float16 *>(y_lo), ld_y, pair_in, pair_out, reinterpret_cast<const int4 *>(segs), ntiles, ntn, cin_pad, part);
    int64_t total = (int64_t)kv * cin_out * cout;
    wgrad_reduce_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(part, seg_off, kv, ntiles, ntn, cin_pad, cin_out, cout, inv_scale, dw);
    GP_CHECK_LAUNCH();
    return GP_OK;
}
