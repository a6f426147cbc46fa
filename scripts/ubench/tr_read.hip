// check of ds_read_b64_tr_b16 semantics: LDS image [32 rows][64 cols] f16 with value row*64+col;
// group g=lane>>4 reads rows 8g+q (q=0..3), lane i=4q+p supplies &img[8g+q][c0+4p]; expect lane i to
// receive column c0+i of rows 8g..8g+3 in elements 0..3.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((vector_size(8)));
__global__ void k(int *o, int c0, int pitch) {
    __shared__ __align__(16) _Float16 lds[32 * 80];
    for (int i = threadIdx.x; i < 32 * 80; i += 64) lds[i] = (_Float16)0;
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * 64; i += 64) lds[(i / 64) * pitch + (i % 64)] = (_Float16)(float)i;   // exact for i < 2048
    __syncthreads();
    int lane = threadIdx.x, g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const _Float16 *a = &lds[(8 * g + q) * pitch + c0 + 4 * p];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)a);
    for (int e = 0; e < 4; ++e) {
        _Float16 h;
        short sv = v[e];
        __builtin_memcpy(&h, &sv, 2);
        o[lane * 4 + e] = (int)(float)h;
    }
}
int main() {
    int *d; hipMalloc(&d, 64 * 4 * 4);
    int h[256];
    for (int pitch : {64, 72}) {
        k<<<1, 64>>>(d, 16, pitch);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 4; ++e) {
                int g = lane >> 4, i = lane & 15;
                int expect = (8 * g + e) * 64 + 16 + i;
                if (h[lane * 4 + e] != expect) ++bad;
            }
        printf("pitch %d: mismatches %d  lane0=%d %d %d %d  lane17=%d %d %d %d\n", pitch, bad, h[0], h[1], h[2], h[3], h[68], h[69], h[70], h[71]);
    }
    return 0;
}
