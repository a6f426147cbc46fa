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

// tuning knobs for experiments: 1 = pooling float4 per lane (0 auto), 2 = pooling unroll, 3 = conv phase-1 ablation mask
int g_gp_knobs[16] = {0, 0, 4, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
extern "C" int gp_debug_set(int32_t key, int32_t value) {
    if (key < 1 || key > 15) return GP_EINVAL;
    g_gp_knobs[key] = value;
    return GP_OK;
}
