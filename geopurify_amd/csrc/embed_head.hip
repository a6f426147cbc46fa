// Row 9, last layer + the normalisation of row 11: the student's 1x1x1 output convolution (hidden -> 128 embedding channels,
// /root/reference/models/affinity_module.py:66,71) fused with F.normalize(., p=2, dim=1) (:1547).
//
// The layer is a dense [nv, cin] x [cin, 128] product read once: 4 B per input element (the f16 hi + lo planes the last 3x3x3
// layer's epilogue already writes, with their per-row power-of-two scale) and 512 B per output row -- HBM-bound, 2.6 KB per voxel row
// at cin = 512.  Same arithmetic as the 3x3x3 layers (gp_sparse_conv_f16x3): every product of two f16 halves is exact in the
// fp32 accumulator of v_mfma_f32_16x16x32_f16, three products per element (hi*hi + hi*lo + lo*hi), summed in a fixed order.
//
// One 256-thread workgroup owns 128 rows x 128 columns; wave w owns rows 32 w .. 32 w + 31 (two 16-row tiles x eight 16-column
// tiles = 64 accumulator registers).  The A fragments go from global memory straight into registers (a lane reads 32 contiguous
// bytes of its row per plane and 64-channel chunk: the k index of an MFMA is a free permutation as long as both operands use the
// same one, so lane (row fl, slot fq) takes channels 16 fq .. 16 fq + 15 of the chunk -- the first eight for the chunk's first
// MFMA step, the others for its second -- and a row's 128-byte line is consumed whole by four lanes); the weight chunk (128 columns x
// 64 channels x two planes = 32 KB) is shared by the four waves through a double-buffered LDS image with a 144-byte row pitch
// (16 lanes x 16-byte reads fall on 64 distinct banks).  Chunk c + 1's loads are issued before chunk c's 96 MFMAs.
#include "gp_common.h"

namespace {
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int EH_ROWS = 128, EH_COLS = 128, EH_KC = 64, EH_PITCH = EH_KC + 8, EH_NT = 256;
constexpr int EH_EP = EH_COLS + 8;                  // halfs per row of the epilogue's plane image (272 bytes)
struct EhSmem {
    _Float16 b_hi[2][EH_COLS][EH_PITCH];
    _Float16 b_lo[2][EH_COLS][EH_PITCH];
};
static_assert(sizeof(EhSmem) >= 4 * 2 * 32 * EH_EP * sizeof(_Float16), "the epilogue's plane images of the four waves reuse the weight buffers");

__global__ void __launch_bounds__(EH_NT)
embed_head_kernel(const _Float16 *__restrict__ x_hi, const _Float16 *__restrict__ x_lo, int64_t ld_x, const float *__restrict__ x_row_inv,
                  const _Float16 *__restrict__ w_hi /*[128][cin]*/, const _Float16 *__restrict__ w_lo, int64_t nv, int cin, float out_scale,
                  int normalize, float *__restrict__ y, int64_t ld_y, _Float16 *__restrict__ e_hi, _Float16 *__restrict__ e_lo,
                  float plane_scale, const int32_t *__restrict__ e_dst_row) {
    __shared__ EhSmem sm;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & 15, fq = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * EH_ROWS + wv * 32;
    // A rows of this lane (rows past the end repeat the last row; their results are not stored)
    const _Float16 *ah[2], *al[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int64_t r = r0 + i * 16 + fl;
        r = r < nv ? r : nv - 1;
        ah[i] = x_hi + r * ld_x + fq * 16;
        al[i] = x_lo + r * ld_x + fq * 16;
    }
    // weight staging role: column tid / 2, channels 32 (tid & 1) .. + 31 of the chunk (four 16-byte pieces per plane)
    const int s_col = tid >> 1, s_half = tid & 1;
    const _Float16 *wh = w_hi + (int64_t)s_col * cin + s_half * 32, *wl = w_lo + (int64_t)s_col * cin + s_half * 32;

    f16x8 a_h[2][2], a_l[2][2], n_h[2][2], n_l[2][2], t_h[4], t_l[4];
    auto load_a = [&](int c0, f16x8 (&h)[2][2], f16x8 (&l)[2][2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                h[i][s] = *reinterpret_cast<const f16x8 *>(ah[i] + c0 + s * 8);
                l[i][s] = *reinterpret_cast<const f16x8 *>(al[i] + c0 + s * 8);
            }
    };
    auto load_b = [&](int c0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            t_h[q] = *reinterpret_cast<const f16x8 *>(wh + c0 + q * 8);
            t_l[q] = *reinterpret_cast<const f16x8 *>(wl + c0 + q * 8);
        }
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *reinterpret_cast<f16x8 *>(&sm.b_hi[buf][s_col][s_half * 32 + q * 8]) = t_h[q];
            *reinterpret_cast<f16x8 *>(&sm.b_lo[buf][s_col][s_half * 32 + q * 8]) = t_l[q];
        }
    };
    f32x4 acc[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int chunks = cin / EH_KC;
    load_a(0, a_h, a_l);
    load_b(0);
    store_b(0);
    __syncthreads();
    for (int c = 0; c < chunks; ++c) {
        const int buf = c & 1;
        const bool more = c + 1 < chunks;
        if (more) {
            load_a((c + 1) * EH_KC, n_h, n_l);
            load_b((c + 1) * EH_KC);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f16x8 bh = *reinterpret_cast<const f16x8 *>(&sm.b_hi[buf][j * 16 + fl][fq * 16 + s * 8]);
                const f16x8 bl = *reinterpret_cast<const f16x8 *>(&sm.b_lo[buf][j * 16 + fl][fq * 16 + s * 8]);
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h[i][s], bh, acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h[i][s], bl, acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_l[i][s], bh, acc[i][j], 0, 0, 0);
            }
        }
        if (more) {
            store_b(buf ^ 1);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int s = 0; s < 2; ++s) { a_h[i][s] = n_h[i][s]; a_l[i][s] = n_l[i][s]; }
        }
        __syncthreads();
    }
    // epilogue.  C layout: column = lane & 15 (+ 16 j), row = (lane >> 4) * 4 + reg.  The scales are powers of two (exact).
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = r0 + i * 16 + fq * 4 + r;
            const float sc = out_scale * (x_row_inv ? x_row_inv[row < nv ? row : nv - 1] : 1.f);
            float v[8], ss = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[j] = acc[i][j][r] * sc; ss += v[j] * v[j]; }
            if (normalize) {
                // the 16 lanes that share `fq` hold the row's other columns
                ss += __shfl_xor(ss, 1, 64);
                ss += __shfl_xor(ss, 2, 64);
                ss += __shfl_xor(ss, 4, 64);
                ss += __shfl_xor(ss, 8, 64);
                const float nrm = fmaxf(sqrtf(ss), 1e-12f);             // F.normalize: x / max(||x||, eps)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = v[j] / nrm;
            }
            if (row < nv && y) {
                float *dst = y + row * ld_y + fl;
#pragma unroll
                for (int j = 0; j < 8; ++j) dst[j * 16] = v[j];
            }
            if (e_hi) {
                // the row as f16 hi / lo planes of v x plane_scale (what gp_affinity_cs_fragments stages by LDS-DMA): through a wave-private
                // LDS image (the weight buffers are free: the loop's last barrier is behind every read of them), 272-byte row pitch so that
                // the four row groups of a store instruction fall on different banks, then 16-byte pieces to global memory
                _Float16 *st = reinterpret_cast<_Float16 *>(&sm) + wv * (2 * 32 * EH_EP);
                const int lrow = i * 16 + fq * 4 + r;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float sv = v[j] * plane_scale;
                    const _Float16 h = (_Float16)sv;
                    st[lrow * EH_EP + j * 16 + fl] = h;
                    st[32 * EH_EP + lrow * EH_EP + j * 16 + fl] = (_Float16)(sv - (float)h);
                }
            }
        }
    if (e_hi) {
        gp_wave_sync();
        const _Float16 *st = reinterpret_cast<const _Float16 *>(&sm) + wv * (2 * 32 * EH_EP);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int id = lane + 64 * t, lrow = id >> 4, piece = id & 15;
            const int64_t row = r0 + lrow;
            if (row < nv) {
                const int64_t ro = e_dst_row ? (int64_t)e_dst_row[row] : row;          // (the plane row: gp_rcb_order's map when the operator is re-ordered)
                *reinterpret_cast<f16x8 *>(e_hi + ro * EH_COLS + piece * 8) = *reinterpret_cast<const f16x8 *>(st + lrow * EH_EP + piece * 8);
                *reinterpret_cast<f16x8 *>(e_lo + ro * EH_COLS + piece * 8) = *reinterpret_cast<const f16x8 *>(st + 32 * EH_EP + lrow * EH_EP + piece * 8);
            }
        }
    }
}
}  // namespace

extern "C" int gp_embed_head_f16x3(const void *x_hi, const void *x_lo, int64_t ld_x, const float *x_row_inv_scale, const void *w_hi,
                                   const void *w_lo, int64_t nv, int32_t cin, int32_t cout, float out_scale, int32_t l2_normalize,
                                   float *y, int64_t ld_y, void *e_hi, void *e_lo, float plane_scale, const int32_t *e_dst_row, void *stream_) {
    GP_CHECK_ARG(x_hi && x_lo && w_hi && w_lo && (y || e_hi) && nv > 0, "gp_embed_head_f16x3: null/empty argument");
    GP_CHECK_ARG(!e_dst_row || e_hi, "gp_embed_head_f16x3: e_dst_row maps the rows of the output planes (e_hi / e_lo)");
    GP_CHECK_ARG((e_hi != nullptr) == (e_lo != nullptr) && (!e_hi || ((uintptr_t)e_hi % 16 == 0 && (uintptr_t)e_lo % 16 == 0 && plane_scale > 0.f)),
                 "gp_embed_head_f16x3: the output planes come as a 16-byte aligned pair with a positive scale");
    GP_CHECK_ARG(cout == EH_COLS, "gp_embed_head_f16x3: cout=%d, this kernel writes %d embedding channels", cout, EH_COLS);
    GP_CHECK_ARG(cin > 0 && cin % EH_KC == 0, "gp_embed_head_f16x3: cin=%d must be a multiple of %d", cin, EH_KC);
    GP_CHECK_ARG(ld_x % 8 == 0 && ld_x >= cin && (uintptr_t)x_hi % 16 == 0 && (uintptr_t)x_lo % 16 == 0,
                 "gp_embed_head_f16x3: pre-split rows must be 16-byte aligned and hold cin channels");
    GP_CHECK_ARG((uintptr_t)w_hi % 16 == 0 && (uintptr_t)w_lo % 16 == 0 && (!y || ld_y >= cout), "gp_embed_head_f16x3: bad weight / output layout");
    GP_CHECK_ARG(out_scale > 0.f, "gp_embed_head_f16x3: out_scale must be the positive inverse of the weights' pre-scale");
    const int64_t blocks = (nv + EH_ROWS - 1) / EH_ROWS;
    embed_head_kernel<<<(unsigned)blocks, EH_NT, 0, gp_stream(stream_)>>>(
        static_cast<const _Float16 *>(x_hi), static_cast<const _Float16 *>(x_lo), ld_x, x_row_inv_scale, static_cast<const _Float16 *>(w_hi),
        static_cast<const _Float16 *>(w_lo), nv, cin, out_scale, l2_normalize, y, ld_y, static_cast<_Float16 *>(e_hi), static_cast<_Float16 *>(e_lo),
        plane_scale, e_dst_row);
    GP_CHECK_LAUNCH();
    return GP_OK;
}
