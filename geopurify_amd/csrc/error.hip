// Error reporting + version for libgeopurify_hip.so
#include <stdarg.h>

#include "gp_common.h"

static thread_local char g_err[512] = "";

extern "C" void gp_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char *gp_last_error(void) { return g_err; }
extern "C" int gp_version(void) { return 100; }

// tuning knobs for experiments (gp_debug_set): 1 = tiled pooling float4 per lane (0 auto), 2 = tiled pooling unroll,
// 3 = conv phase-1 ablation mask (1 no loads after step 0, 2 no MFMA, 8 no partial stores, 16 register-staged path,
// 32 LDS-staged epilogue), 4 = matrix-core pooling ablation mask (1 no reads/MFMA, 2 no row gather, 4 no epilogue,
// 8 no weight fragments, 16 no output stores), 5 = force brute-force 1-NN, 6 = 1-NN grid cells per axis,
// 8 no weight fragments, 16 no output stores; persistent kernel: 32 waves 4-7 issue DMA after the sweep, 64 nt weight loads),
// 9 = matrix-core pooling: force one workgroup per CU, 10 = persistent pooling: workgroups per XCD label (0 = CUs/8),
// 11 = matrix-core pooling: 4 = column-sliced waves (64 rows x 32 columns each), 12 = persistent pooling: 1 forces the static tile lists (no queue),
// 7 = 1-NN: 2 brings back the fine-grid pass for near queries, 13 = 1-NN coarse grid cells per axis (<= 64),
// 14 = classify: 1 forces the kernel without the LDS-staged text matrix,
// 15 = affinity: 1 forces the one-wave-per-row kernel (no LDS staging of the distinct neighbour rows), 2 = 8 rows per workgroup instead of 16
int g_gp_knobs[16] = {0, 0, 4, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
extern "C" int gp_debug_set(int32_t key, int32_t value) {
    if (key < 1 || key > 15) return GP_EINVAL;
    g_gp_knobs[key] = value;
    return GP_OK;
}

// device buffers for experiments (gp_debug_ptr): 0 = matrix-core pooling: in-kernel time stamps, 10 x uint64 per wave
// (selects the stamped instantiation while non-null)
void *g_gp_debug_ptr[4] = {nullptr, nullptr, nullptr, nullptr};
extern "C" int gp_debug_ptr(int32_t key, void *p) {
    if (key < 0 || key > 3) return GP_EINVAL;
    g_gp_debug_ptr[key] = p;
    return GP_OK;
}
