// device-side decode of a fused-feature file (dataset/feature_loader.py:113-192).
//
// A `{scene}_{k}.pt` file holds `feat` [rows, D] -- one row per point of the scene's CHUNK, in point order -- and
// `mask_full` [N] (which points are in the chunk); the three-key form adds `mask` (which chunk rows were seen by a
// camera).  After voxelization the loader keeps one row per voxel REPRESENTATIVE point vox_ind[v].  The reference does
// this with nonzero / cumsum / fancy indexing on the host, twice per item; here it is one exclusive scan of the chunk
// mask (rank(p) = row of point p in `feat`), one pass over the voxels, one scan of the kept voxels (compact form only)
// and one row copy:
//     p = vox_ind[v];  in = mask_chunk[p];  r = rank(p);  keep = in && (row_keep ? row_keep[r] : 1)
//     mode 0 (training forms, :141-181):   mask_out[v] = keep;  out[pos(v)] = feat[r] for the kept voxels, in voxel order
//     mode 1 (evaluation forms, :119-126, :183-190):   mask_out[v] = keep;  out[v] = in ? feat[r] : 0
// Rows are copied as bytes (fp16 and fp32 files alike).  Memory-bound: nv * row_bytes read + written.
#include <rocprim/device/device_scan.hpp>

#include "gp_common.h"

namespace {

__global__ void fd_flags_kernel(const uint8_t *__restrict__ mask_chunk, int64_t n, const int32_t *__restrict__ rank,
                                const uint8_t *__restrict__ row_keep, int64_t feat_rows, const int64_t *__restrict__ vox_ind,
                                int64_t nv, int32_t *__restrict__ keep, int32_t *__restrict__ row, uint8_t *__restrict__ mask_out) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    const int64_t p = vox_ind[v];
    int r = -1, k = 0;
    if (p >= 0 && p < n && mask_chunk[p]) {
        const int rr = rank[p];
        if (rr < feat_rows) {                                // (a file whose mask and row count disagree: treated as outside)
            r = rr;
            k = row_keep ? (row_keep[rr] != 0) : 1;
        }
    }
    keep[v] = k;
    row[v] = r;
    mask_out[v] = (uint8_t)k;
}

// one wave per voxel row; 16-byte pieces when the rows allow it
template <int VEC>
__global__ void __launch_bounds__(256)
fd_copy_kernel(const unsigned char *__restrict__ feat, int64_t row_bytes, const int32_t *__restrict__ keep,
               const int32_t *__restrict__ pos, const int32_t *__restrict__ row, int64_t nv, int mode,
               unsigned char *__restrict__ out) {
    const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= nv) return;
    const int lane = threadIdx.x & 63;
    const int r = row[v];
    int64_t dst;
    if (mode == 0) {
        if (!keep[v]) return;
        dst = pos[v];
    } else {
        dst = v;
    }
    typedef unsigned char piece __attribute__((ext_vector_type(VEC)));
    const piece *s = reinterpret_cast<const piece *>(feat + (int64_t)(r < 0 ? 0 : r) * row_bytes);
    piece *d = reinterpret_cast<piece *>(out + dst * row_bytes);
    const int64_t np = row_bytes / VEC;
    piece z;
#pragma unroll
    for (int i = 0; i < VEC; ++i) z[i] = 0;
    for (int64_t i = lane; i < np; i += 64) d[i] = r < 0 ? z : s[i];
}

size_t fd_scan_tmp(int64_t n) {
    size_t t = 0;
    (void)rocprim::exclusive_scan(nullptr, t, (int32_t *)nullptr, (int32_t *)nullptr, (int32_t)0, (size_t)n, rocprim::plus<int32_t>(), 0);
    return t;
}

struct FdWork {
    int32_t *ones, *rank, *keep, *pos, *row;
    char *tmp;
    size_t tmp_bytes;
};
size_t fd_carve(void *ws, size_t bytes, int64_t n, int64_t nv, FdWork &k) {
    GpCarver cv(ws, bytes);
    k.ones = cv.take<int32_t>(n + 1);
    k.rank = cv.take<int32_t>(n + 1);
    k.keep = cv.take<int32_t>(nv + 1);
    k.pos = cv.take<int32_t>(nv + 1);
    k.row = cv.take<int32_t>(nv + 1);
    const size_t a = fd_scan_tmp(n + 1), b = fd_scan_tmp(nv + 1);
    k.tmp_bytes = a > b ? a : b;
    k.tmp = cv.take<char>(k.tmp_bytes);
    return cv.off;
}

__global__ void fd_widen_kernel(const uint8_t *__restrict__ m, int64_t n, int32_t *__restrict__ o) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n) o[i] = (i < n && m[i]) ? 1 : 0;
}
// n_sel[0] = kept voxels; n_sel[1] = points in the chunk = the number of rows a well-formed file holds (the caller compares it
// with feat_rows: the host formulation of the reference raises on a file whose mask and row count disagree, fd_flags_kernel
// only masks the rows beyond the file's end)
__global__ void fd_count_kernel(const int32_t *__restrict__ pos, int64_t nv, const int32_t *__restrict__ rank, int64_t n,
                                int64_t *__restrict__ n_sel) {
    n_sel[0] = pos[nv];
    n_sel[1] = rank[n];
}

}  // namespace

extern "C" size_t gp_fused_decode_workspace_bytes(int64_t n, int64_t nv) {
    if (n <= 0 || nv <= 0) return 0;
    FdWork k;
    return fd_carve(nullptr, 0, n, nv, k);
}

extern "C" int gp_fused_decode(const uint8_t *mask_chunk, int64_t n, const uint8_t *row_keep, const void *feat, int64_t feat_rows,
                               int64_t row_bytes, const int64_t *vox_ind, int64_t nv, int32_t mode, void *out, uint8_t *mask_out,
                               int64_t *n_sel, void *workspace, size_t workspace_bytes, void *stream_) {
    GP_CHECK_ARG(mask_chunk && feat && vox_ind && out && mask_out && n_sel && workspace, "gp_fused_decode: null argument");
    GP_CHECK_ARG(n > 0 && nv > 0 && feat_rows > 0 && row_bytes > 0, "gp_fused_decode: empty argument");
    GP_CHECK_ARG(n < ((int64_t)1 << 31) - 1 && nv < ((int64_t)1 << 31) - 1, "gp_fused_decode: more than 2^31 points");
    GP_CHECK_ARG(mode == 0 || mode == 1, "gp_fused_decode: mode=%d (0 compact / training, 1 dense / evaluation)", mode);
    GP_CHECK_ARG(row_bytes % 2 == 0, "gp_fused_decode: row_bytes=%lld (rows of fp16 or fp32 elements)", (long long)row_bytes);
    FdWork k;
    const size_t need = fd_carve(workspace, workspace_bytes, n, nv, k);
    if (need > workspace_bytes) { gp_set_error("gp_fused_decode: workspace too small (%zu < %zu)", workspace_bytes, need); return GP_ENOMEM; }
    hipStream_t s = gp_stream(stream_);
    fd_widen_kernel<<<(unsigned)((n + 1 + 255) / 256), 256, 0, s>>>(mask_chunk, n, k.ones);
    GP_CHECK_HIP(rocprim::exclusive_scan(k.tmp, k.tmp_bytes, k.ones, k.rank, (int32_t)0, (size_t)(n + 1), rocprim::plus<int32_t>(), s));
    fd_flags_kernel<<<(unsigned)((nv + 255) / 256), 256, 0, s>>>(mask_chunk, n, k.rank, row_keep, feat_rows, vox_ind, nv, k.keep, k.row, mask_out);
    GP_CHECK_HIP(hipMemsetAsync(k.keep + nv, 0, sizeof(int32_t), s));
    GP_CHECK_HIP(rocprim::exclusive_scan(k.tmp, k.tmp_bytes, k.keep, k.pos, (int32_t)0, (size_t)(nv + 1), rocprim::plus<int32_t>(), s));
    fd_count_kernel<<<1, 1, 0, s>>>(k.pos, nv, k.rank, n, n_sel);
    const bool v16 = row_bytes % 16 == 0 && (uintptr_t)feat % 16 == 0 && (uintptr_t)out % 16 == 0;
    const unsigned grid = (unsigned)((nv + 3) / 4);
    if (v16)
        fd_copy_kernel<16><<<grid, 256, 0, s>>>(static_cast<const unsigned char *>(feat), row_bytes, k.keep, k.pos, k.row, nv, mode,
                                                static_cast<unsigned char *>(out));
    else
        fd_copy_kernel<2><<<grid, 256, 0, s>>>(static_cast<const unsigned char *>(feat), row_bytes, k.keep, k.pos, k.row, nv, mode,
                                               static_cast<unsigned char *>(out));
    GP_CHECK_LAUNCH();
    return GP_OK;
}
