// Shared helpers for the gfx950 kernels behind include/geopurify_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>

#include "../../include/geopurify_hip.h"

#define GP_WAVE 64

// gp_pool_cs_*: the affinity weights (<= 1) are stored x 2^10 in the operator's f16 fragments so that their lo parts stay normal
// (shared by the builder in pool_mfma_cs.hip and the affinity kernels that write fragments directly, pool.hip)
#define GP_POOL_CS_WSCALE 1024.f

// (internal to the library: hidden, not part of the C-ABI of include/geopurify_hip.h -- the message is read with gp_last_error)
extern "C" __attribute__((visibility("hidden"))) void gp_set_error(const char *fmt, ...);

#define GP_CHECK_ARG(cond, ...)                \
    do {                                       \
        if (!(cond)) {                         \
            gp_set_error(__VA_ARGS__);         \
            return GP_EINVAL;                  \
        }                                      \
    } while (0)

#define GP_CHECK_HIP(expr)                                                              \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            gp_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                         __LINE__);                                                     \
            return GP_EHIP;                                                             \
        }                                                                               \
    } while (0)

#define GP_CHECK_LAUNCH() GP_CHECK_HIP(hipGetLastError())

static inline hipStream_t gp_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// hipFuncAttributeMaxDynamicSharedMemorySize once per (kernel, DEVICE): the attribute belongs to the device's code object, so a
// process that drives several GPUs (or first calls from two threads) needs it on each of them -- a function-local `static bool`
// is neither.  One bit per device id in an atomic word per call site; a lost race sets the attribute twice (harmless).
#define GP_SMEM_ATTR(func, bytes)                                                                                          \
    do {                                                                                                                   \
        static std::atomic<uint64_t> gp_attr_done_{0};                                                                     \
        int gp_dev_ = 0;                                                                                                   \
        GP_CHECK_HIP(hipGetDevice(&gp_dev_));                                                                              \
        const uint64_t gp_bit_ = 1ull << (gp_dev_ & 63);                                                                   \
        if (!(gp_attr_done_.load(std::memory_order_acquire) & gp_bit_)) {                                                  \
            GP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(func), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                             (int)(bytes)));                                                               \
            gp_attr_done_.fetch_or(gp_bit_, std::memory_order_release);                                                    \
        }                                                                                                                  \
    } while (0)

// compute units of the CURRENT device (cached per device id; 0 on error)
static inline int gp_cu_count() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    int n = cache[dev & 63].load(std::memory_order_relaxed);
    if (n > 0) return n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    cache[dev & 63].store(n, std::memory_order_relaxed);
    return n;
}

static inline size_t gp_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// carve a sub-buffer out of a caller workspace (256-byte aligned)
struct GpCarver {
    char *base;
    size_t size, off;
    GpCarver(void *p, size_t n) : base(static_cast<char *>(p)), size(n), off(0) {}
    template <typename T>
    T *take(size_t count) {
        size_t bytes = gp_align_up(count * sizeof(T), 256);
        char *r = base ? base + off : nullptr;
        off += bytes;
        return reinterpret_cast<T *>(r);
    }
    bool ok() const { return off <= size; }
};

__device__ __forceinline__ int gp_lane() { return threadIdx.x & 63; }

// order LDS/global accesses of the lanes of ONE wave (no cross-wave meaning)
__device__ __forceinline__ void gp_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float gp_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float gp_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// power of two s with amax * s in [2^13, 2^14): the pre-scale of an fp32 -> (hi, lo) f16 split.  Every element within
// 2^-18 of amax then has a NORMAL f16 lo half (|lo| >= 2^-14), i.e. x = hi + lo to 2^-22 relative; f16 max is 65504.
// amax == 0 (or not finite) -> 1.  Exponent clamped so that s and 1/s are normal fp32 numbers.
__device__ __forceinline__ float gp_pow2_for(float amax) {
    if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.f;
    int e;
    frexpf(amax, &e);                      // amax = m * 2^e, m in [0.5, 1)
    int k = 14 - e;                        // amax * 2^k = m * 2^14 in [2^13, 2^14)
    k = k > 100 ? 100 : (k < -100 ? -100 : k);
    return ldexpf(1.f, k);
}
__device__ __forceinline__ int gp_wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
