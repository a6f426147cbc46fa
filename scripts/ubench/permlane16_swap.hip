#include <hip/hip_runtime.h>
__global__ void k(unsigned *o) {
    unsigned a = threadIdx.x, b = threadIdx.x + 1000;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned *d; hipMalloc(&d, 512); k<<<1, 64>>>(d); unsigned h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int i = 0; i < 128; ++i) printf("%u%c", h[i], (i % 16 == 15) ? '\n' : ' ');
}
