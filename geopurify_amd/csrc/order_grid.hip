// Morton ordering of voxels + lattice grid build + 27-offset kernel map (SURVEY 8a rows 9/10 support).
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "gp_grid.h"

namespace {

__global__ void minmax_kernel(const int32_t *__restrict__ c, int64_t nv, int32_t *__restrict__ mm) {
    // mm[0..2] = min, mm[3..5] = max (pre-initialised)
    int lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            int v = c[i * 3 + a];
            lo[a] = min(lo[a], v);
            hi[a] = max(hi[a], v);
        }
    }
    // wave reduce -> block reduce in LDS -> ONE atomic per block and bound (same-address atomics serialise at the memory
    // side: one per wave of a 586-block grid cost 140 us, 64 blocks x 6 cost 1 us)
    __shared__ int s_lo[4][3], s_hi[4][3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = min(lo[a], __shfl_xor(lo[a], o, 64));
            hi[a] = max(hi[a], __shfl_xor(hi[a], o, 64));
        }
        if (gp_lane() == 0) { s_lo[threadIdx.x >> 6][a] = lo[a]; s_hi[threadIdx.x >> 6][a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        atomicMin(&mm[a], min(min(s_lo[0][a], s_lo[1][a]), min(s_lo[2][a], s_lo[3][a])));
        atomicMax(&mm[3 + a], max(max(s_hi[0][a], s_hi[1][a]), max(s_hi[2][a], s_hi[3][a])));
    }
}

__global__ void init_minmax_kernel(int32_t *mm) {
    if (threadIdx.x < 3) mm[threadIdx.x] = INT32_MAX;
    else if (threadIdx.x < 6) mm[threadIdx.x] = INT32_MIN;
}

__global__ void morton_keys_kernel(const int32_t *__restrict__ c, int64_t nv, const int32_t *__restrict__ mm,
                                   uint64_t *__restrict__ keys, int32_t *__restrict__ vals) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= nv) return;
    uint32_t x = (uint32_t)(c[i * 3 + 0] - mm[0]);
    uint32_t y = (uint32_t)(c[i * 3 + 1] - mm[1]);
    uint32_t z = (uint32_t)(c[i * 3 + 2] - mm[2]);
    keys[i] = gp_morton3(x, y, z);
    vals[i] = (int32_t)i;
}

__global__ void invert_perm_kernel(const int32_t *__restrict__ perm, int64_t nv, int32_t *__restrict__ rank) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < nv) rank[perm[i]] = (int32_t)i;
}

// ---------------------------------------------------------------------------------------- grid
__global__ void grid_init_kernel(void *grid, int3 origin, int3 extent, int3 cdim, int64_t nv, int32_t max_cells,
                                 int64_t cell_index_off, int64_t records_off) {
    GpGridHeader *h = reinterpret_cast<GpGridHeader *>(grid);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        h->origin[0] = origin.x; h->origin[1] = origin.y; h->origin[2] = origin.z;
        h->extent[0] = extent.x; h->extent[1] = extent.y; h->extent[2] = extent.z;
        h->cdim[0] = cdim.x; h->cdim[1] = cdim.y; h->cdim[2] = cdim.z;
        h->status = 0; h->ncells = 0; h->max_cells = max_cells; h->nv = nv;
        h->cell_index_off = cell_index_off; h->records_off = records_off;
    }
    int32_t *ci = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(grid) + cell_index_off);
    int64_t ncell = (int64_t)cdim.x * cdim.y * cdim.z;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < ncell; i += (int64_t)gridDim.x * blockDim.x)
        ci[i] = -1;
}

__device__ __forceinline__ int64_t cell_raster(const GpGridHeader *h, int rx, int ry, int rz) {
    return ((int64_t)(rz >> 3) * h->cdim[1] + (ry >> 3)) * h->cdim[0] + (rx >> 3);
}

__global__ void grid_heads_kernel(void *grid, const int32_t *__restrict__ c, int64_t nv) {
    GpGridHeader *h = reinterpret_cast<GpGridHeader *>(grid);
    int32_t *ci = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(grid) + h->cell_index_off);
    GpCellRec *recs = reinterpret_cast<GpCellRec *>(reinterpret_cast<char *>(grid) + h->records_off);
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= nv) return;
    int rx = c[i * 3] - h->origin[0], ry = c[i * 3 + 1] - h->origin[1], rz = c[i * 3 + 2] - h->origin[2];
    bool bad = (unsigned)rx >= (unsigned)h->extent[0] || (unsigned)ry >= (unsigned)h->extent[1] ||
               (unsigned)rz >= (unsigned)h->extent[2];
    uint64_t key = gp_morton3(rx, ry, rz);
    bool head = true;
    if (i > 0) {
        int px = c[i * 3 - 3] - h->origin[0], py = c[i * 3 - 2] - h->origin[1], pz = c[i * 3 - 1] - h->origin[2];
        uint64_t pk = gp_morton3(px, py, pz);
        if (pk >= key) bad = true;
        head = (pk >> 9) != (key >> 9);
    }
    if (bad) { atomicOr(&h->status, 1); return; }
    if (head) {
        int slot = atomicAdd(&h->ncells, 1);
        GpCellRec &r = recs[slot];
        r.start = (int32_t)i;
        r.count = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) r.bits[j] = 0u;
        ci[cell_raster(h, rx, ry, rz)] = slot;
    }
}

__global__ void grid_fill_kernel(void *grid, const int32_t *__restrict__ c, int64_t nv) {
    GpGridHeader *h = reinterpret_cast<GpGridHeader *>(grid);
    if (h->status) return;
    int32_t *ci = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(grid) + h->cell_index_off);
    GpCellRec *recs = reinterpret_cast<GpCellRec *>(reinterpret_cast<char *>(grid) + h->records_off);
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= nv) return;
    int rx = c[i * 3] - h->origin[0], ry = c[i * 3 + 1] - h->origin[1], rz = c[i * 3 + 2] - h->origin[2];
    int slot = ci[cell_raster(h, rx, ry, rz)];
    uint32_t l = gp_local9(rx, ry, rz);
    atomicOr(&recs[slot].bits[l >> 5], 1u << (l & 31));
    atomicAdd(&recs[slot].count, 1);
}

// ---------------------------------------------------------------------------------------- kernel map
__global__ void kernel_map_kernel(const void *grid, const int32_t *__restrict__ c, int64_t nv,
                                  int32_t *__restrict__ nbr_map) {
    GpGridView g(grid);
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= nv) return;
    int rx = c[i * 3] - g.h->origin[0], ry = c[i * 3 + 1] - g.h->origin[1], rz = c[i * 3 + 2] - g.h->origin[2];
    int k = 0;
    for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx, ++k)
                nbr_map[(int64_t)k * nv + i] = (k == 13) ? (int32_t)i : g.lookup_rel(rx + dx, ry + dy, rz + dz);
}

}  // namespace

// ============================================================================================
// per-axis minimum and maximum of integer coordinates [nv,3] -> mm i32 [6] = (min xyz, max xyz), on the device, no sync
// (the loader needs the voxel extent; torch's column reduction of an [n,3] array took 0.115 ms)
extern "C" int gp_minmax_i32(const int32_t *coords, int64_t nv, int32_t *mm, void *stream_) {
    GP_CHECK_ARG(coords && mm && nv > 0, "gp_minmax_i32: null/empty argument");
    hipStream_t s = gp_stream(stream_);
    int blocks = (int)((nv + 255) / 256);
    init_minmax_kernel<<<1, 64, 0, s>>>(mm);
    minmax_kernel<<<min(blocks, 64), 256, 0, s>>>(coords, nv, mm);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" size_t gp_morton_order_workspace_bytes(int64_t nv) {
    size_t tmp = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tmp, (uint64_t *)nullptr, (uint64_t *)nullptr, (int32_t *)nullptr,
                              (int32_t *)nullptr, (size_t)nv, 0, 64, 0);
    GpCarver cv(nullptr, 0);
    cv.take<int32_t>(8);
    cv.take<uint64_t>(nv);
    cv.take<uint64_t>(nv);
    cv.take<int32_t>(nv);
    cv.take<char>(tmp);
    return cv.off;
}

extern "C" int gp_morton_order(const int32_t *coords, int64_t nv, int32_t *perm, int32_t *rank, void *workspace,
                               size_t workspace_bytes, void *stream_) {
    GP_CHECK_ARG(nv > 0 && nv < (1ll << 31), "gp_morton_order: nv=%lld out of range", (long long)nv);
    hipStream_t s = gp_stream(stream_);
    size_t tmp = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tmp, (uint64_t *)nullptr, (uint64_t *)nullptr, (int32_t *)nullptr,
                              (int32_t *)nullptr, (size_t)nv, 0, 64, 0);
    GpCarver cv(workspace, workspace_bytes);
    int32_t *mm = cv.take<int32_t>(8);
    uint64_t *k0 = cv.take<uint64_t>(nv);
    uint64_t *k1 = cv.take<uint64_t>(nv);
    int32_t *v0 = cv.take<int32_t>(nv);
    char *t = cv.take<char>(tmp);
    if (!cv.ok()) { gp_set_error("gp_morton_order: workspace too small (%zu < %zu)", workspace_bytes, cv.off); return GP_ENOMEM; }
    int blocks = (int)((nv + 255) / 256);
    init_minmax_kernel<<<1, 64, 0, s>>>(mm);
    minmax_kernel<<<min(blocks, 64), 256, 0, s>>>(coords, nv, mm);          // 256 threads: the kernel's LDS reduce assumes 4 waves
    morton_keys_kernel<<<blocks, 256, 0, s>>>(coords, nv, mm, k0, v0);
    GP_CHECK_LAUNCH();
    GP_CHECK_HIP(rocprim::radix_sort_pairs(t, tmp, k0, k1, v0, perm, (size_t)nv, 0, 64, s));
    invert_perm_kernel<<<blocks, 256, 0, s>>>(perm, nv, rank);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

static void grid_layout(int64_t nv, const int32_t *extent, int3 &cdim, int32_t &max_cells, int64_t &ci_off,
                        int64_t &rec_off, size_t &total) {
    cdim = make_int3(((extent[0] - 1) >> 3) + 1, ((extent[1] - 1) >> 3) + 1, ((extent[2] - 1) >> 3) + 1);
    int64_t ncell = (int64_t)cdim.x * cdim.y * cdim.z;
    max_cells = (int32_t)(nv < ncell ? nv : ncell);
    ci_off = 256;
    rec_off = ci_off + (int64_t)gp_align_up((size_t)ncell * 4, 256);
    total = (size_t)rec_off + gp_align_up((size_t)max_cells * sizeof(GpCellRec), 256);
}

extern "C" size_t gp_grid_bytes(int64_t nv, const int32_t *extent_host) {
    if (!extent_host || nv <= 0 || extent_host[0] <= 0 || extent_host[1] <= 0 || extent_host[2] <= 0) return 0;
    int3 cdim; int32_t mc; int64_t a, b; size_t total;
    grid_layout(nv, extent_host, cdim, mc, a, b, total);
    return total;
}

extern "C" int gp_grid_build(const int32_t *coords, int64_t nv, const int32_t *origin_host, const int32_t *extent_host,
                             void *grid, size_t grid_bytes, void *stream_) {
    GP_CHECK_ARG(coords && grid && origin_host && extent_host && nv > 0, "gp_grid_build: null/empty argument");
    for (int a = 0; a < 3; ++a)
        if (extent_host[a] <= 0 || extent_host[a] > (1 << 15)) { gp_set_error("gp_grid_build: extent[%d]=%d not in 1..32768", a, extent_host[a]); return GP_ERANGE; }
    int3 cdim; int32_t mc; int64_t ci_off, rec_off; size_t total;
    grid_layout(nv, extent_host, cdim, mc, ci_off, rec_off, total);
    if (grid_bytes < total) { gp_set_error("gp_grid_build: grid buffer too small (%zu < %zu)", grid_bytes, total); return GP_ENOMEM; }
    hipStream_t s = gp_stream(stream_);
    int blocks = (int)((nv + 255) / 256);
    grid_init_kernel<<<1024, 256, 0, s>>>(grid, make_int3(origin_host[0], origin_host[1], origin_host[2]),
                                         make_int3(extent_host[0], extent_host[1], extent_host[2]), cdim, nv, mc, ci_off, rec_off);
    grid_heads_kernel<<<blocks, 256, 0, s>>>(grid, coords, nv);
    grid_fill_kernel<<<blocks, 256, 0, s>>>(grid, coords, nv);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" int gp_kernel_map_build(const void *grid, const int32_t *coords, int64_t nv, int32_t *nbr_map, void *stream_) {
    GP_CHECK_ARG(grid && coords && nbr_map && nv > 0, "gp_kernel_map_build: null/empty argument");
    kernel_map_kernel<<<(int)((nv + 255) / 256), 256, 0, gp_stream(stream_)>>>(grid, coords, nv, nbr_map);
    GP_CHECK_LAUNCH();
    return GP_OK;
}
