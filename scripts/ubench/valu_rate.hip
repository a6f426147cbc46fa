// micro-benchmark: sustained fp32 FMA rate of plain v_fmac_f32 vs v_pk_fma_f32 (calibration aid)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float w) {
    float a[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) a[i] = threadIdx.x * 0.001f + i;
    float x = out[threadIdx.x & 7];
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(w), "v"(x));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(w), "v"(x));
        } else {
#pragma unroll
            for (int i = 0; i < 32; i += 2) {
                f2 acc = {a[i], a[i + 1]};
                f2 xv = {x, x}, wv = {w, w};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(wv), "v"(xv));
                a[i] = acc[0]; a[i + 1] = acc[1];
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float *d;
    hipMalloc(&d, 1 << 24);
    hipMemset(d, 0, 1 << 24);
    int iters = 20000;
    for (int mode = 0; mode < 3; ++mode) {
        for (int bpc : {2, 4, 8}) {
            int blocks = 256 * bpc;
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) k<0><<<blocks, 256>>>(d, iters, 1.0001f);
                if (mode == 1) k<1><<<blocks, 256>>>(d, iters, 1.0001f);
                if (mode == 2) k<2><<<blocks, 256>>>(d, iters, 1.0001f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            double fma = (double)blocks * 256 * iters * 32;
            printf("mode %d (%s) blocks/CU %d: %.3f ms  %.1f TFMA/s = %.1f TFLOP/s\n", mode,
                   mode == 0 ? "v_fmac vgpr" : mode == 1 ? "v_fmac sgpr" : "v_pk_fma", bpc, ms, fma / ms / 1e9, 2 * fma / ms / 1e9);
        }
    }
    return 0;
}
