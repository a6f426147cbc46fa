/* The drop-in boundary without Python or PyTorch: a plain C program that drives the affinity pooling
 * (models/affinity_module.py:1575-1587, `torch.sparse.mm(A, X)`) through the C-ABI of include/geopurify_hip.h --
 * raw device pointers, explicit sizes, caller-owned workspaces, int status + gp_last_error().
 *
 *   gcc -std=c99 -O2 -D__HIP_PLATFORM_AMD__ examples/c_abi_pooling.c -Iinclude -I/opt/rocm/include -Lgeopurify_amd -lgeopurify_hip \\
 *       -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/geopurify_amd -Wl,-rpath,/opt/rocm/lib -lm -o /tmp/c_abi_pooling
 *   /tmp/c_abi_pooling            # prints the max deviation from a double-precision host loop and "OK"
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "geopurify_hip.h"

#define CHECK_HIP(e)                                                                       \
    do {                                                                                   \
        hipError_t err_ = (e);                                                             \
        if (err_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(err_)); return 2; } \
    } while (0)
#define CHECK_GP(e)                                                                        \
    do {                                                                                   \
        int rc_ = (e);                                                                     \
        if (rc_ != GP_OK) { fprintf(stderr, "%s failed (%d): %s\n", #e, rc_, gp_last_error()); return 3; } \
    } while (0)

static uint64_t rng_state = 88172645463325252ull;
static double rnd(void) {                               /* xorshift64*, uniform in [0,1) */
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return (double)((rng_state * 2685821657736338717ull) >> 11) / 9007199254740992.0;
}

int main(void) {
    const int64_t nv = 3000;
    const int32_t k = 96, d = 512, br = 64;
    /* a row-stochastic operator with local structure: neighbours of row i lie in a window around i, distinct, no self */
    int32_t *nbr = (int32_t *)malloc(sizeof(int32_t) * nv * k);
    float *w = (float *)malloc(sizeof(float) * nv * k);
    float *x = (float *)malloc(sizeof(float) * nv * d);
    for (int64_t i = 0; i < nv; ++i) {
        double s = 0;
        for (int j = 0; j < k; ++j) {
            int64_t c = i - k / 2 + j + (j >= k / 2);   /* skips i itself */
            c = (c % nv + nv) % nv;
            nbr[i * k + j] = (int32_t)c;
            w[i * k + j] = (float)(0.05 + rnd());
            s += w[i * k + j];
        }
        for (int j = 0; j < k; ++j) w[i * k + j] = (float)(w[i * k + j] / s);
    }
    for (int64_t i = 0; i < nv * d; ++i) x[i] = (float)(2.0 * rnd() - 1.0);

    int32_t *d_nbr, *d_bu_n, *d_bu_row;
    float *d_w, *d_x, *d_y;
    int64_t *d_bu_off;
    void *d_xh, *d_xl, *d_wah, *d_wal, *d_ws;
    const int64_t nb = (nv + br - 1) / br;
    CHECK_HIP(hipMalloc((void **)&d_nbr, sizeof(int32_t) * nv * k));
    CHECK_HIP(hipMalloc((void **)&d_w, sizeof(float) * nv * k));
    CHECK_HIP(hipMalloc((void **)&d_x, sizeof(float) * nv * d));
    CHECK_HIP(hipMalloc((void **)&d_y, sizeof(float) * nv * d));
    CHECK_HIP(hipMalloc(&d_xh, 2 * nv * d));
    CHECK_HIP(hipMalloc(&d_xl, 2 * nv * d));
    CHECK_HIP(hipMalloc((void **)&d_bu_off, sizeof(int64_t) * (nb + 1)));
    CHECK_HIP(hipMalloc((void **)&d_bu_n, sizeof(int32_t) * nb));
    CHECK_HIP(hipMemcpy(d_nbr, nbr, sizeof(int32_t) * nv * k, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_w, w, sizeof(float) * nv * k, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_x, x, sizeof(float) * nv * d, hipMemcpyHostToDevice));

    size_t ws_bytes = gp_pool_mfma_workspace_bytes(nv, br);
    CHECK_HIP(hipMalloc(&d_ws, ws_bytes));
    CHECK_GP(gp_pool_mfma_count(d_nbr, nv, k, br, 0, d_bu_off, d_bu_n, d_ws, ws_bytes, NULL));
    int64_t total = 0;
    CHECK_HIP(hipMemcpy(&total, d_bu_off + nb, sizeof(int64_t), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMalloc((void **)&d_bu_row, sizeof(int32_t) * total));
    size_t wa_bytes = (size_t)(total / 32) * (br / 16) * 64 * 8 * 2;
    CHECK_HIP(hipMalloc(&d_wah, wa_bytes));
    CHECK_HIP(hipMalloc(&d_wal, wa_bytes));
    CHECK_GP(gp_pool_mfma_fill(d_nbr, d_w, nv, k, br, d_bu_off, d_bu_n, total, d_bu_row, d_wah, d_wal, NULL));
    CHECK_GP(gp_split_f16(d_x, d, d, nv, d_xh, d_xl, d, NULL));
    CHECK_GP(gp_pool_mfma_apply(d_xh, d_xl, d, d_bu_off, d_bu_row, d_wah, d_wal, nv, d, br, NULL, NULL, 0, d_y, d, NULL, NULL));
    CHECK_HIP(hipDeviceSynchronize());

    float *y = (float *)malloc(sizeof(float) * nv * d);
    CHECK_HIP(hipMemcpy(y, d_y, sizeof(float) * nv * d, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int64_t i = 0; i < nv; i += 7)                  /* every 7th row against a double-precision loop */
        for (int c = 0; c < d; ++c) {
            double ref = 0;
            for (int j = 0; j < k; ++j) ref += (double)w[i * k + j] * (double)x[(int64_t)nbr[i * k + j] * d + c];
            double e = fabs(ref - (double)y[i * d + c]);
            if (e > worst) worst = e;
        }
    printf("padded union rows per output row: %.2f, max |y - ref| = %.3g\n", (double)total / (double)nv, worst);
    /* error path: the library reports, it does not abort */
    int rc = gp_pool_mfma_apply(d_xh, d_xl, d, d_bu_off, d_bu_row, d_wah, d_wal, nv, 300, br, NULL, NULL, 0, d_y, d, NULL, NULL);
    printf("d = 300 is rejected: rc = %d, \"%s\"\n", rc, gp_last_error());
    if (worst > 1e-5 || rc != GP_EINVAL) { printf("FAILED\n"); return 1; }
    printf("OK\n");
    return 0;
}
