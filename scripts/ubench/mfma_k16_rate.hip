// MFMA issue rate on gfx950: v_mfma_f32_16x16x32_f16 (K = 32, new on CDNA4) vs the legacy v_mfma_f32_16x16x16_f16 (K = 16).
// One wave per SIMD (256 threads per block, one block per CU), independent accumulators, operands in registers.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K32>
__global__ void k(float *out, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    f16x8 a8, b8;
    f16x4 a4, b4;
    for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(threadIdx.x * 0.001f + i); b8[i] = (_Float16)(i * 0.5f); }
    for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (K32) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[j], 0, 0, 0);
            else acc[j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[j], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float *d;
    hipMalloc(&d, 256 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200000;
    for (int v = 0; v < 2; ++v) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (v) k<1><<<256, 256>>>(d, iters); else k<0><<<256, 256>>>(d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double mf = 8.0 * iters;                                  // MFMAs per wave
            double flop = mf * 1024 * (v ? 16384.0 : 8192.0);         // 1024 waves
            printf("%s: %.2f ms, %.1f ns per MFMA per wave, %.1f TFLOP/s\n", v ? "16x16x32_f16" : "16x16x16_f16", ms, ms * 1e6 / mf, flop / ms / 1e9);
        }
    }
    return 0;
}
