// Error reporting + version for libgeopurify_hip.so
#include <stdarg.h>

#include "gp_common.h"

static thread_local char g_err[512] = "";

extern "C" __attribute__((visibility("hidden"))) void gp_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char *gp_last_error(void) { return g_err; }
extern "C" int gp_version(void) { return 100; }

// Tuning knobs for experiments (gp_debug_set).  NONE of them reaches a product kernel: pooling and convolution honour knobs 3 / 4
// only in their *_tuning_kernel twins (the product instantiations compile the bits out), the others pick a launch shape or a
// slower, equally tested kernel.  A key or value outside this table is GP_EINVAL.
//   1  tiled pooling: float4 per lane (0 = auto, 1..4)             2  tiled pooling: unroll (0..8; knob default 4)
//   3  convolution phase 1 (mask): 1 no loads after step 0 and 4 no LDS staging stores (register-staged path only), 2 no MFMA,
//      8 no partial stores, 16 register-staged path, 32 fp32 partial rows (rounds 1-4) instead of 24-bit block floating point,
//      256 no partial rows of the centre offset
//      (written or read: the price of a centre-offset fold), 512 nothing (selects the twin)
//   4  matrix-core pooling (mask): 1 no reads / MFMA, 2 hot piece instead of the row gather, 4 no epilogue, 8 hot piece instead of
//      the weight fragments, 16 no output stores, 32 every wave issues its DMA first, 64 stamp the issue segment, 128 the bytes of an
//      8-bit lo plane (a price: half of the lo rows from the hot piece, half of the lo output bytes),
//      256 / 512 engine: no staged-row reads / no weight-fragment reads
//   5  1-NN: 1 forces brute force          6  1-NN fine grid cells per axis (0 = 128, <= 256)
//   7  1-NN: 2 brings back the fine-grid pass for near queries            9  64-row pooling: 1 forces one workgroup per CU
//   8  matrix-core affinity (mask, tuning twin): 1 no fragment reads / MFMA, 2 no list stores, 4 no softmax, 8 no fragment pass, 16 no LDS-DMA
//  10  persistent pooling: workgroups per XCD label (0 = CUs / 8, <= 64)   11  64-row pooling: 4 = column-sliced waves
//  12  persistent pooling: 1 forces the static tile lists                  13  1-NN coarse grid cells per axis (0 = 32, <= 64)
//  14  classify: 1 forces the kernel without the LDS-staged text matrix
//  15  affinity: 1 forces the one-wave-per-row kernel, 2 = 8 rows per workgroup instead of 16
int g_gp_knobs[16] = {0, 0, 4, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
namespace {
struct KnobRule { int lo, hi; unsigned mask; };          // mask != 0: value must be a subset of the mask; else lo <= value <= hi
const KnobRule k_rules[16] = {
    {0, -1, 0},          // 0: no such knob
    {0, 4, 0},           // 1
    {0, 8, 0},           // 2
    {0, 0, 1u | 2u | 4u | 8u | 16u | 32u | 256u | 512u},               // 3
    {0, 0, 1u | 2u | 4u | 8u | 16u | 32u | 64u | 128u | 256u | 512u},     // 4
    {0, 1, 0},           // 5
    {0, 256, 0},         // 6
    {0, 2, 0},           // 7
    {0, 0, 1u | 2u | 4u | 8u | 16u},                               // 8
    {0, 1, 0},           // 9
    {0, 64, 0},          // 10
    {0, 4, 0},           // 11 (0 or 4; the engine is gp_pool_cs_apply_engine, not a knob)
    {0, 1, 0},           // 12
    {0, 64, 0},          // 13
    {0, 1, 0},           // 14
    {0, 2, 0},           // 15
};
}  // namespace
extern "C" int gp_debug_set(int32_t key, int32_t value) {
    if (key < 1 || key > 15) { gp_set_error("gp_debug_set: no knob %d", key); return GP_EINVAL; }
    const KnobRule &r = k_rules[key];
    const bool ok = r.mask ? (value >= 0 && ((unsigned)value & ~r.mask) == 0u) : (value >= r.lo && value <= r.hi);
    if (!ok || (key == 11 && value != 0 && value != 4) || (key == 7 && value == 1)) {
        gp_set_error("gp_debug_set: value %d is not defined for knob %d", value, key);
        return GP_EINVAL;
    }
    g_gp_knobs[key] = value;
    return GP_OK;
}

// device buffers for experiments (gp_debug_ptr): 0 = matrix-core pooling: in-kernel time stamps, 10 x uint64 per wave
// (selects the stamped instantiation while non-null); `bytes` is the buffer's size, checked by every launch that writes stamps
void *g_gp_debug_ptr[4] = {nullptr, nullptr, nullptr, nullptr};
size_t g_gp_debug_bytes[4] = {0, 0, 0, 0};
extern "C" int gp_debug_ptr(int32_t key, void *p, size_t bytes) {
    if (key < 0 || key > 3) { gp_set_error("gp_debug_ptr: no buffer %d", key); return GP_EINVAL; }
    if ((p == nullptr) != (bytes == 0)) { gp_set_error("gp_debug_ptr: a buffer comes with its size (and NULL with 0)"); return GP_EINVAL; }
    g_gp_debug_ptr[key] = p;
    g_gp_debug_bytes[key] = bytes;
    return GP_OK;
}
