// Rows 1-3: voxelizer (fp64 affine + floor + FNV-1 hash + sort/unique) and point->pixel mapping.
// Integer/index work is bit-exact with the numpy reference; fp64 products follow the BLAS
// accumulation order (fma chain over k) of the reference's [N,4]@[4,3] and [4,4]@[4,N] matmuls.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "gp_common.h"

namespace {

struct Mat34 { double m[3][4]; };
struct Mat44 { double m[4][4]; };

__device__ __forceinline__ double dot4_blas(const double *r, double x, double y, double z) {
    // acc = r0*x ; acc = fma(r1,y,acc) ; acc = fma(r2,z,acc) ; acc = fma(r3,1,acc)
    double acc = r[0] * x;
    acc = fma(r[1], y, acc);
    acc = fma(r[2], z, acc);
    acc = fma(r[3], 1.0, acc);
    return acc;
}

__global__ void vox_affine_kernel(const double *__restrict__ c, int64_t n, Mat34 R, double *__restrict__ out,
                                  long long *__restrict__ mn) {
    long long lo[3] = {LLONG_MAX, LLONG_MAX, LLONG_MAX};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double x = c[i * 3], y = c[i * 3 + 1], z = c[i * 3 + 2];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            double v = floor(dot4_blas(R.m[a], x, y, z));
            out[i * 3 + a] = v;
            long long iv = (long long)v;
            lo[a] = iv < lo[a] ? iv : lo[a];
        }
    }
    // wave reduce -> block reduce -> one atomic per block and axis (same-address atomics serialise at the memory side)
    __shared__ long long s_lo[4][3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int o = 32; o > 0; o >>= 1) {
            long long t = __shfl_xor(lo[a], o, 64);
            lo[a] = t < lo[a] ? t : lo[a];
        }
        if (gp_lane() == 0) s_lo[threadIdx.x >> 6][a] = lo[a];
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        long long v = s_lo[0][a];
#pragma unroll
        for (int w = 1; w < 4; ++w) v = s_lo[w][a] < v ? s_lo[w][a] : v;
        atomicMin(&mn[a], v);
    }
}

__global__ void init_min_kernel(long long *mn) {
    if (threadIdx.x < 3) mn[threadIdx.x] = LLONG_MAX;
}

__device__ __forceinline__ uint64_t fnv3(double a, double b, double c) {
    uint64_t h = 14695981039346656037ull;
    h *= 1099511628211ull; h ^= (uint64_t)a;
    h *= 1099511628211ull; h ^= (uint64_t)b;
    h *= 1099511628211ull; h ^= (uint64_t)c;
    return h;
}

__global__ void vox_hash_kernel(double *__restrict__ c, int64_t n, const long long *__restrict__ mn,
                                uint64_t *__restrict__ keys, int64_t *__restrict__ vals) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        v[a] = floor(c[i * 3 + a] - (double)mn[a]);
        c[i * 3 + a] = v[a];
    }
    keys[i] = fnv3(v[0], v[1], v[2]);
    vals[i] = i;
}

__global__ void fnv_kernel(const double *__restrict__ c, int64_t n, uint64_t *__restrict__ h) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) h[i] = fnv3(c[i * 3], c[i * 3 + 1], c[i * 3 + 2]);
}

__global__ void head_flags_kernel(const uint64_t *__restrict__ ks, int64_t n, int32_t *__restrict__ flag) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || ks[i] != ks[i - 1]) ? 1 : 0;
}

__global__ void vox_emit_kernel(const double *__restrict__ c, const int64_t *__restrict__ order,
                                const int32_t *__restrict__ flag, const int32_t *__restrict__ scan, int64_t n,
                                double *__restrict__ coords_aug, int64_t *__restrict__ inds,
                                int64_t *__restrict__ inv, int64_t *__restrict__ nv_dev,
                                int64_t *__restrict__ order_out, int64_t *__restrict__ seg_start) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t p = order[i];
    int64_t v = (int64_t)scan[i] - 1;
    inv[p] = v;
    if (order_out) order_out[i] = p;
    if (flag[i]) {
        inds[v] = p;
        coords_aug[v * 3] = c[p * 3];
        coords_aug[v * 3 + 1] = c[p * 3 + 1];
        coords_aug[v * 3 + 2] = c[p * 3 + 2];
        if (seg_start) seg_start[v] = i;
    }
    if (i == n - 1) {
        *nv_dev = v + 1;
        if (seg_start) seg_start[v + 1] = n;
    }
}

// ------------------------------------------------------------------------------------------------
// one point through one view: pixel (vi = row, ui = column) and the visibility decision of fusion_util.py:99-147 / :45-82
__device__ __forceinline__ bool project_point(double x, double y, double z, const double *m0, const double *m1, const double *m2,
                                              double fx, double fy, double cx, double cy, const double *__restrict__ depth,
                                              int W, int H, int cut, double tau, long long &ui, long long &vi) {
    double p0 = dot4_blas(m0, x, y, z), p1 = dot4_blas(m1, x, y, z), p2 = dot4_blas(m2, x, y, z);
    double u = (p0 * fx) / p2 + cx;
    double v = (p1 * fy) / p2 + cy;
    double ur = rint(u), vr = rint(v);
    bool finite = (fabs(ur) < 9.0e15) && (fabs(vr) < 9.0e15);      // also false for NaN
    ui = finite ? (long long)ur : -1;
    vi = finite ? (long long)vr : -1;
    bool inside = finite && ui >= cut && vi >= cut && ui < (long long)W - cut && vi < (long long)H - cut;
    if (depth) {
        if (inside) {
            double d = depth[vi * W + ui];
            inside = fabs(d - p2) <= tau * d;
        }
    } else {
        inside = inside && (p2 > 0.0);
    }
    return inside;
}

// ---- all views of a scene in one launch per step (instead of V x {project, flags, scan, compact}):
// params f64 [V,20] = row-major world->camera (16) | fx fy cx cy; depth f64 [V,H,W] or NULL.  Entries come out view-major,
// ascending point id inside a view -- the concatenation of the per-view lists of gp_visible_lists.
__global__ void views_flags_kernel(const double *__restrict__ c, int64_t n, const double *__restrict__ params,
                                   const double *__restrict__ depth, int W, int H, int cut, double tau, int32_t *__restrict__ flags) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int v = blockIdx.y;
    if (i >= n) return;
    const double *P = params + v * 20;
    long long ui, vi;
    flags[(int64_t)v * n + i] = project_point(c[i * 3], c[i * 3 + 1], c[i * 3 + 2], P, P + 4, P + 8, P[16], P[17], P[18], P[19],
                                              depth ? depth + (int64_t)v * W * H : nullptr, W, H, cut, tau, ui, vi) ? 1 : 0;
}
__global__ void views_compact_kernel(const double *__restrict__ c, int64_t n, int nviews, const double *__restrict__ params,
                                     const double *__restrict__ depth, int W, int H, int cut, double tau,
                                     const int32_t *__restrict__ sc /* exclusive scan of the flags */, int64_t *__restrict__ ent_pt,
                                     int64_t *__restrict__ ent_x, int64_t *__restrict__ ent_y, int32_t *__restrict__ ent_view,
                                     int64_t *__restrict__ view_off) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int v = blockIdx.y;
    if (i >= n) return;
    const double *P = params + v * 20;
    long long ui, vi;
    const bool inside = project_point(c[i * 3], c[i * 3 + 1], c[i * 3 + 2], P, P + 4, P + 8, P[16], P[17], P[18], P[19],
                                      depth ? depth + (int64_t)v * W * H : nullptr, W, H, cut, tau, ui, vi);
    const int32_t pos = sc[(int64_t)v * n + i];
    if (inside) { ent_pt[pos] = i; ent_x[pos] = vi; ent_y[pos] = ui; ent_view[pos] = v; }
    if (i == 0) view_off[v] = pos;
    if (v == nviews - 1 && i == n - 1) view_off[nviews] = pos + (inside ? 1 : 0);
}
// the view-drop rule of the loader (data_loader_ablation.py:254-255, 280-288) on the device
__global__ void views_keep_kernel(const int64_t *__restrict__ view_off, int nviews, int64_t min_visible, int64_t val_keep,
                                  uint8_t *__restrict__ keep) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nviews) return;
    const int64_t nv = view_off[v + 1] - view_off[v];
    keep[v] = (nv != 0 && nv >= min_visible && nv <= val_keep) ? 1 : 0;
}

__global__ void project_kernel(const double *__restrict__ c, int64_t n, Mat44 M, double fx, double fy, double cx,
                               double cy, const double *__restrict__ depth, int W, int H, int cut, double tau,
                               int64_t *__restrict__ mapping, double *__restrict__ weight) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    long long ui, vi;
    const bool inside = project_point(c[i * 3], c[i * 3 + 1], c[i * 3 + 2], M.m[0], M.m[1], M.m[2], fx, fy, cx, cy, depth, W, H,
                                      cut, tau, ui, vi);
    mapping[i * 3] = inside ? vi : 0;
    mapping[i * 3 + 1] = inside ? ui : 0;
    mapping[i * 3 + 2] = inside ? 1 : 0;
    if (weight) {
        double a = (double)ui - (double)W / 2, b = (double)vi - (double)H / 2;
        weight[i] = exp(-sqrt(a * a + b * b) / 10.0);
    }
}

// z-buffer of the cloud itself (fusion_util.py:126-130, depth given as a str): depth[v,u] = min z over the
// points with z > 0.2 that project inside the cut bound; pixels nobody hits keep 999999.  Positive doubles
// order like their bit patterns, so the minimum is an integer atomicMin (order-independent => exact).
__global__ void depth_fill_kernel(double *__restrict__ depth, int64_t n) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) depth[i] = 999999.0;
}
__global__ void render_depth_kernel(const double *__restrict__ c, int64_t n, Mat44 M, double fx, double fy, double cx,
                                    double cy, int W, int H, int cut, double *__restrict__ depth) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = c[i * 3], y = c[i * 3 + 1], z = c[i * 3 + 2];
    double p0 = dot4_blas(M.m[0], x, y, z), p1 = dot4_blas(M.m[1], x, y, z), p2 = dot4_blas(M.m[2], x, y, z);
    double u = (p0 * fx) / p2 + cx;
    double v = (p1 * fy) / p2 + cy;
    double ur = rint(u), vr = rint(v);
    bool finite = (fabs(ur) < 9.0e15) && (fabs(vr) < 9.0e15);
    long long ui = finite ? (long long)ur : -1, vi = finite ? (long long)vr : -1;
    bool inside = finite && ui >= cut && vi >= cut && ui < (long long)W - cut && vi < (long long)H - cut;
    if (inside && p2 > 0.2)
        atomicMin(reinterpret_cast<unsigned long long *>(depth + vi * W + ui), (unsigned long long)__double_as_longlong(p2));
}

size_t sort_tmp_bytes(int64_t n) {
    size_t t = 0;
    (void)rocprim::radix_sort_pairs(nullptr, t, (uint64_t *)nullptr, (uint64_t *)nullptr, (int64_t *)nullptr,
                                    (int64_t *)nullptr, (size_t)n, 0, 64, 0);
    size_t t2 = 0;
    (void)rocprim::inclusive_scan(nullptr, t2, (int32_t *)nullptr, (int32_t *)nullptr, (size_t)n, rocprim::plus<int32_t>(), 0);
    return t > t2 ? t : t2;
}

struct VoxWs {
    long long *mn; double *ctmp; uint64_t *k0, *k1; int64_t *v0, *v1; int32_t *flag, *scan; char *tmp; size_t tmp_bytes;
};
VoxWs carve_vox(GpCarver &cv, int64_t n) {
    VoxWs w;
    w.tmp_bytes = sort_tmp_bytes(n);
    w.mn = cv.take<long long>(4);
    w.ctmp = cv.take<double>(3 * n);
    w.k0 = cv.take<uint64_t>(n);
    w.k1 = cv.take<uint64_t>(n);
    w.v0 = cv.take<int64_t>(n);
    w.v1 = cv.take<int64_t>(n);
    w.flag = cv.take<int32_t>(n);
    w.scan = cv.take<int32_t>(n);
    w.tmp = cv.take<char>(w.tmp_bytes);
    return w;
}

}  // namespace

extern "C" size_t gp_voxelize_workspace_bytes(int64_t n) {
    if (n <= 0) return 0;
    GpCarver cv(nullptr, 0);
    carve_vox(cv, n);
    return cv.off;
}

extern "C" int gp_voxelize_f64(const double *coords, int64_t n, const double *rigid_host, double *coords_aug,
                               int64_t *inds, int64_t *inds_reconstruct, int64_t *nv_dev, int64_t *order,
                               int64_t *seg_start, void *workspace, size_t workspace_bytes, void *stream_) {
    GP_CHECK_ARG(coords && rigid_host && coords_aug && inds && inds_reconstruct && nv_dev && workspace,
                 "gp_voxelize_f64: null argument");
    GP_CHECK_ARG(n > 0 && n < (1ll << 31), "gp_voxelize_f64: n=%lld out of range (empty clouds are rejected like the reference's assert)", (long long)n);
    GpCarver cv(workspace, workspace_bytes);
    VoxWs w = carve_vox(cv, n);
    if (!cv.ok()) { gp_set_error("gp_voxelize_f64: workspace too small (%zu < %zu)", workspace_bytes, cv.off); return GP_ENOMEM; }
    hipStream_t s = gp_stream(stream_);
    Mat34 R;
    for (int a = 0; a < 3; ++a)
        for (int k = 0; k < 4; ++k) R.m[a][k] = rigid_host[a * 4 + k];
    int blocks = (int)((n + 255) / 256);
    init_min_kernel<<<1, 64, 0, s>>>(w.mn);
    vox_affine_kernel<<<blocks < 256 ? blocks : 256, 256, 0, s>>>(coords, n, R, w.ctmp, w.mn);       // 256 threads: 4-wave block reduce
    vox_hash_kernel<<<blocks, 256, 0, s>>>(w.ctmp, n, w.mn, w.k0, w.v0);
    GP_CHECK_LAUNCH();
    size_t tb = w.tmp_bytes;
    GP_CHECK_HIP(rocprim::radix_sort_pairs(w.tmp, tb, w.k0, w.k1, w.v0, w.v1, (size_t)n, 0, 64, s));
    head_flags_kernel<<<blocks, 256, 0, s>>>(w.k1, n, w.flag);
    tb = w.tmp_bytes;
    GP_CHECK_HIP(rocprim::inclusive_scan(w.tmp, tb, w.flag, w.scan, (size_t)n, rocprim::plus<int32_t>(), s));
    vox_emit_kernel<<<blocks, 256, 0, s>>>(w.ctmp, w.v1, w.flag, w.scan, n, coords_aug, inds, inds_reconstruct, nv_dev,
                                           order, seg_start);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" int gp_fnv_hash_f64(const double *coords, int64_t n, uint64_t *hash, void *stream_) {
    GP_CHECK_ARG(coords && hash && n > 0, "gp_fnv_hash_f64: null/empty argument");
    fnv_kernel<<<(int)((n + 255) / 256), 256, 0, gp_stream(stream_)>>>(coords, n, hash);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" int gp_project_points_f64(const double *coords, int64_t n, const double *w2c_host, double fx, double fy,
                                     double cx, double cy, const double *depth, int32_t width, int32_t height,
                                     int32_t cut_bound, double vis_thres, int64_t *mapping, double *weight,
                                     void *stream_) {
    GP_CHECK_ARG(coords && w2c_host && mapping && n > 0, "gp_project_points_f64: null/empty argument");
    GP_CHECK_ARG(width > 0 && height > 0, "gp_project_points_f64: bad image size %dx%d", width, height);
    Mat44 M;
    for (int a = 0; a < 4; ++a)
        for (int k = 0; k < 4; ++k) M.m[a][k] = w2c_host[a * 4 + k];
    project_kernel<<<(int)((n + 255) / 256), 256, 0, gp_stream(stream_)>>>(coords, n, M, fx, fy, cx, cy, depth, width,
                                                                          height, cut_bound, vis_thres, mapping, weight);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

static size_t views_scan_tmp(int64_t n) {
    size_t t = 0;
    (void)rocprim::exclusive_scan(nullptr, t, (int32_t *)nullptr, (int32_t *)nullptr, (int32_t)0, (size_t)n, rocprim::plus<int32_t>(), 0);
    return t;
}
extern "C" size_t gp_views_visible_lists_workspace_bytes(int64_t n, int32_t nviews) {
    if (n <= 0 || nviews <= 0) return 0;
    GpCarver cv(nullptr, 0);
    cv.take<int32_t>(n * nviews); cv.take<int32_t>(n * nviews); cv.take<char>(views_scan_tmp(n * nviews));
    return cv.off;
}
extern "C" int gp_views_visible_lists(const double *coords, int64_t n, const double *params, const double *depth, int32_t nviews,
                                      int32_t width, int32_t height, int32_t cut_bound, double vis_thres, int64_t min_visible,
                                      int64_t val_keep, int64_t *ent_pt, int64_t *ent_x, int64_t *ent_y, int32_t *ent_view,
                                      int64_t *view_off, uint8_t *keep, void *workspace, size_t workspace_bytes, void *stream_) {
    GP_CHECK_ARG(coords && params && ent_pt && ent_x && ent_y && ent_view && view_off && keep && workspace && n > 0,
                 "gp_views_visible_lists: null/empty argument");
    GP_CHECK_ARG(nviews > 0 && nviews <= 65535 && width > 0 && height > 0, "gp_views_visible_lists: bad view count / image size");
    GP_CHECK_ARG(n * (int64_t)nviews < (int64_t)INT32_MAX, "gp_views_visible_lists: n * views must stay below 2^31");
    GpCarver cv(workspace, workspace_bytes);
    int32_t *flags = cv.take<int32_t>(n * nviews), *sc = cv.take<int32_t>(n * nviews);
    size_t tb = views_scan_tmp(n * nviews);
    char *tmp = cv.take<char>(tb);
    if (!cv.ok()) { gp_set_error("gp_views_visible_lists: workspace too small"); return GP_ENOMEM; }
    hipStream_t s = gp_stream(stream_);
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)nviews);
    views_flags_kernel<<<grid, 256, 0, s>>>(coords, n, params, depth, width, height, cut_bound, vis_thres, flags);
    GP_CHECK_HIP(rocprim::exclusive_scan(tmp, tb, flags, sc, (int32_t)0, (size_t)(n * nviews), rocprim::plus<int32_t>(), s));
    views_compact_kernel<<<grid, 256, 0, s>>>(coords, n, nviews, params, depth, width, height, cut_bound, vis_thres, sc, ent_pt, ent_x,
                                              ent_y, ent_view, view_off);
    views_keep_kernel<<<(nviews + 255) / 256, 256, 0, s>>>(view_off, nviews, min_visible, val_keep, keep);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" int gp_render_depth_f64(const double *coords, int64_t n, const double *w2c_host, double fx, double fy, double cx,
                                   double cy, int32_t width, int32_t height, int32_t cut_bound, double *depth,
                                   void *stream_) {
    GP_CHECK_ARG(coords && w2c_host && depth && n > 0, "gp_render_depth_f64: null/empty argument");
    GP_CHECK_ARG(width > 0 && height > 0, "gp_render_depth_f64: bad image size %dx%d", width, height);
    Mat44 M;
    for (int a = 0; a < 4; ++a)
        for (int k = 0; k < 4; ++k) M.m[a][k] = w2c_host[a * 4 + k];
    hipStream_t s = gp_stream(stream_);
    int64_t px = (int64_t)width * height;
    depth_fill_kernel<<<(int)((px + 255) / 256), 256, 0, s>>>(depth, px);
    render_depth_kernel<<<(int)((n + 255) / 256), 256, 0, s>>>(coords, n, M, fx, fy, cx, cy, width, height, cut_bound, depth);
    GP_CHECK_LAUNCH();
    return GP_OK;
}
