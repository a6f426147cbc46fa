// Exact 1-NN fill (sklearn KDTree k=1 semantics), classify/argmax, IoU histograms (row 13).
#include <cstring>
#include <rocprim/device/device_scan.hpp>

#include "gp_common.h"

extern int g_gp_knobs[16];

namespace {

constexpr int NN_TILE = 1024;

// brute force in fp64: d2 = ((dx*dx + dy*dy) + dz*dz), strict '<' keeps the lowest reference index.
// grid = (query blocks, reference splits); partial results are combined by nn1_reduce_kernel.
__global__ void __launch_bounds__(256)
nn1_partial_kernel(const float *__restrict__ ref, int64_t n_ref, const float *__restrict__ qry, int64_t n_q,
                   int splits, double *__restrict__ pd, int64_t *__restrict__ pi) {
    __shared__ double sx[NN_TILE], sy[NN_TILE], sz[NN_TILE];
    int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    int sp = blockIdx.y;
    int64_t per = (n_ref + splits - 1) / splits;
    int64_t r0 = sp * per, r1 = r0 + per < n_ref ? r0 + per : n_ref;
    double qx = 0, qy = 0, qz = 0;
    if (q < n_q) { qx = qry[q * 3]; qy = qry[q * 3 + 1]; qz = qry[q * 3 + 2]; }
    double best = INFINITY;
    int64_t bi = -1;
    for (int64_t t0 = r0; t0 < r1; t0 += NN_TILE) {
        int64_t cnt = r1 - t0 < NN_TILE ? r1 - t0 : NN_TILE;
        __syncthreads();
        for (int j = threadIdx.x; j < cnt; j += 256) {
            sx[j] = ref[(t0 + j) * 3]; sy[j] = ref[(t0 + j) * 3 + 1]; sz[j] = ref[(t0 + j) * 3 + 2];
        }
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {
            double dx = qx - sx[j], dy = qy - sy[j], dz = qz - sz[j];
            double d2 = (dx * dx + dy * dy) + dz * dz;
            if (d2 < best) { best = d2; bi = t0 + j; }
        }
    }
    if (q < n_q) { pd[(int64_t)sp * n_q + q] = best; pi[(int64_t)sp * n_q + q] = bi; }
}

__global__ void nn1_reduce_kernel(const double *__restrict__ pd, const int64_t *__restrict__ pi, int64_t n_q, int splits,
                                  int64_t *__restrict__ nn) {
    int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (q >= n_q) return;
    double best = INFINITY;
    int64_t bi = -1;
    for (int s = 0; s < splits; ++s) {                            // ascending splits = ascending index ranges
        double d = pd[(int64_t)s * n_q + q];
        if (d < best) { best = d; bi = pi[(int64_t)s * n_q + q]; }
    }
    nn[q] = bi;
}


// ---- masked variant: references / queries are subsets of one point array, selected by byte masks.
// Ordered compaction (exclusive scans) keeps ascending point ids, so "lowest index wins ties" holds.
__global__ void mask_flags_kernel(const uint8_t *__restrict__ rm, const uint8_t *__restrict__ qm, int64_t n,
                                  int32_t *__restrict__ rf, int32_t *__restrict__ qf) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) { rf[i] = rm[i] ? 1 : 0; qf[i] = qm[i] ? 1 : 0; }
}
__global__ void mask_compact_kernel(const float *__restrict__ xyz, const uint8_t *__restrict__ rm,
                                    const uint8_t *__restrict__ qm, const int32_t *__restrict__ rs,
                                    const int32_t *__restrict__ qs, int64_t n, float *__restrict__ rxyz,
                                    int64_t *__restrict__ ridx, int64_t *__restrict__ qidx, int32_t *__restrict__ counts,
                                    int64_t *__restrict__ nn) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    nn[i] = -1;
    if (rm[i]) {
        int p = rs[i];
        rxyz[p * 3] = xyz[i * 3]; rxyz[p * 3 + 1] = xyz[i * 3 + 1]; rxyz[p * 3 + 2] = xyz[i * 3 + 2];
        ridx[p] = i;
    }
    if (qm[i]) qidx[qs[i]] = i;
    if (i == n - 1) { counts[0] = rs[i] + (rm[i] ? 1 : 0); counts[1] = qs[i] + (qm[i] ? 1 : 0); }
}
// brute force for small sets: one query per thread; the compacted reference set is cut into NN_CHUNKS ranges that
// different workgroups stream through LDS tiles (a view has a few thousand queries = a handful of workgroups: one
// range per workgroup would leave the chip idle), then a second kernel takes the lexicographic (d2, index) minimum
// over the ranges in ascending order -- the same winner as a single sequential scan.
constexpr int NN_CHUNKS = 16;
__global__ void __launch_bounds__(256)
nn1_masked_part_kernel(const float *__restrict__ xyz, const float *__restrict__ rxyz, const int64_t *__restrict__ qidx,
                       const int32_t *__restrict__ counts, double *__restrict__ part_d, int32_t *__restrict__ part_i, int64_t stride) {
    __shared__ double sx[NN_TILE], sy[NN_TILE], sz[NN_TILE];
    const int n_ref = counts[0], n_q = counts[1];
    if ((int64_t)blockIdx.x * 256 >= n_q || n_ref == 0) return;
    const int per = (n_ref + NN_CHUNKS - 1) / NN_CHUNKS;
    const int r0 = blockIdx.y * per, r1 = r0 + per < n_ref ? r0 + per : n_ref;
    int64_t qi = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    int64_t q = qi < n_q ? qidx[qi] : -1;
    double qx = 0, qy = 0, qz = 0;
    if (q >= 0) { qx = xyz[q * 3]; qy = xyz[q * 3 + 1]; qz = xyz[q * 3 + 2]; }
    double best = INFINITY;
    int bi = -1;
    for (int t0 = r0; t0 < r1; t0 += NN_TILE) {
        int cnt = r1 - t0 < NN_TILE ? r1 - t0 : NN_TILE;
        __syncthreads();
        for (int j = threadIdx.x; j < cnt; j += 256) {
            sx[j] = rxyz[(int64_t)(t0 + j) * 3]; sy[j] = rxyz[(int64_t)(t0 + j) * 3 + 1]; sz[j] = rxyz[(int64_t)(t0 + j) * 3 + 2];
        }
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {
            double dx = qx - sx[j], dy = qy - sy[j], dz = qz - sz[j];
            double d2 = (dx * dx + dy * dy) + dz * dz;
            if (d2 < best) { best = d2; bi = t0 + j; }
        }
    }
    if (qi < n_q) { part_d[(int64_t)blockIdx.y * stride + qi] = best; part_i[(int64_t)blockIdx.y * stride + qi] = bi; }
}
__global__ void nn1_masked_reduce_kernel(const double *__restrict__ part_d, const int32_t *__restrict__ part_i, int64_t stride,
                                         const int64_t *__restrict__ ridx, const int64_t *__restrict__ qidx,
                                         const int32_t *__restrict__ counts, int64_t *__restrict__ nn) {
    const int n_ref = counts[0], n_q = counts[1];
    int64_t qi = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (qi >= n_q || n_ref == 0) return;
    double best = INFINITY;
    int bi = -1;
    for (int c = 0; c < NN_CHUNKS; ++c) {
        double d = part_d[(int64_t)c * stride + qi];
        int i = part_i[(int64_t)c * stride + qi];
        if (i >= 0 && d < best) { best = d; bi = i; }
    }
    nn[qidx[qi]] = ridx[bi];
}


// ---- grid-accelerated exact 1-NN (large point sets).  A uniform grid over the bounding box of ALL
// points (so every query lies inside its cell); references are bucketed by cell (counting sort).  A query
// scans the cubic shells of cells around its own cell; after shell r every unscanned point is at least
// r*h away, so the search stops once best_d <= r*h (with a 1e-9 relative slack for the cell rounding).
// Winner = lexicographic minimum of (d2 in fp64, original index): identical to the brute-force kernel.
constexpr int NGMAX = 128, NG2MAX = 64;                                 // buffers are sized for this many cells per axis

struct NnGrid { double lo[3]; double inv_h; double h; int ng; };

__global__ void nn_bbox_kernel(const float *__restrict__ xyz, int64_t n, float *__restrict__ bb /*[6] min,max*/) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
#pragma unroll
        for (int a = 0; a < 3; ++a) { float v = xyz[i * 3 + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
    // wave reduce -> block reduce in LDS -> one atomic per block and bound (same-address atomics serialise: one per wave
    // of a 586-block grid cost 160 us)
    __shared__ float s_lo[4][3], s_hi[4][3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int o = 32; o > 0; o >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64)); }
        if (gp_lane() == 0) { s_lo[threadIdx.x >> 6][a] = lo[a]; s_hi[threadIdx.x >> 6][a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        const float l = fminf(fminf(s_lo[0][a], s_lo[1][a]), fminf(s_lo[2][a], s_lo[3][a]));
        const float h = fmaxf(fmaxf(s_hi[0][a], s_hi[1][a]), fmaxf(s_hi[2][a], s_hi[3][a]));
        // float atomic min/max through the ordered-int trick (values are finite; an empty block contributes +-inf: no-ops)
        int *pl = reinterpret_cast<int *>(bb + a), *ph = reinterpret_cast<int *>(bb + 3 + a);
        int il = __float_as_int(l), ih = __float_as_int(h);
        if (il >= 0) atomicMin(pl, il); else atomicMax(reinterpret_cast<unsigned *>(pl), (unsigned)il);
        if (ih >= 0) atomicMax(ph, ih); else atomicMin(reinterpret_cast<unsigned *>(ph), (unsigned)ih);
    }
}
__global__ void nn_bbox_init_kernel(float *bb) {
    if (threadIdx.x < 3) bb[threadIdx.x] = INFINITY;
    else if (threadIdx.x < 6) bb[threadIdx.x] = -INFINITY;
}
__device__ __forceinline__ void nn_grid_of(const float *bb, NnGrid &g, int NG) {
    g.ng = NG;
    double ext = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) { g.lo[a] = bb[a]; double e = (double)bb[3 + a] - (double)bb[a]; ext = e > ext ? e : ext; }
    g.h = ext > 0 ? ext * (1.0 + 1e-6) / NG : 1.0;
    g.inv_h = 1.0 / g.h;
}
__device__ __forceinline__ int nn_cell(const NnGrid &g, double v, int a) {
    int c = (int)floor((v - g.lo[a]) * g.inv_h);
    return c < 0 ? 0 : (c >= g.ng ? g.ng - 1 : c);
}
__global__ void nn_cell_count_kernel(const float *__restrict__ rxyz, const int32_t *__restrict__ counts, const float *__restrict__ bb,
                                     int32_t *__restrict__ cell_cnt, int32_t *__restrict__ rcell, int NG) {
    int n_ref = counts[0];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_ref) return;
    NnGrid g; nn_grid_of(bb, g, NG);
    int c = (nn_cell(g, rxyz[i * 3 + 2], 2) * NG + nn_cell(g, rxyz[i * 3 + 1], 1)) * NG + nn_cell(g, rxyz[i * 3], 0);
    rcell[i] = c;
    atomicAdd(&cell_cnt[c], 1);
}
__global__ void nn_cell_fill_kernel(const float *__restrict__ rxyz, const int64_t *__restrict__ ridx, const int32_t *__restrict__ counts,
                                    const int32_t *__restrict__ rcell, const int32_t *__restrict__ cell_start,
                                    int32_t *__restrict__ cursor, float *__restrict__ sxyz, int64_t *__restrict__ sidx) {
    int n_ref = counts[0];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_ref) return;
    int c = rcell[i];
    int p = cell_start[c] + atomicAdd(&cursor[c], 1);
    sxyz[p * 3] = rxyz[i * 3]; sxyz[p * 3 + 1] = rxyz[i * 3 + 1]; sxyz[p * 3 + 2] = rxyz[i * 3 + 2];
    sidx[p] = ridx[i];
}
__global__ void __launch_bounds__(256)
nn_grid_query_kernel(const float *__restrict__ xyz, const float *__restrict__ sxyz, const int64_t *__restrict__ sidx,
                     const int32_t *__restrict__ cell_start, const int64_t *__restrict__ qidx, const int32_t *__restrict__ counts,
                     const float *__restrict__ bb, int64_t *__restrict__ nn, int NG, int r_max, int pending_only) {
    const int n_ref = counts[0], n_q = counts[1];
    int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= n_q || n_ref == 0) return;
    NnGrid g; nn_grid_of(bb, g, NG);
    const int64_t q = qidx[qi];
    if (pending_only && nn[q] != -2) return;
    const double qx = xyz[q * 3], qy = xyz[q * 3 + 1], qz = xyz[q * 3 + 2];
    const int cx = nn_cell(g, qx, 0), cy = nn_cell(g, qy, 1), cz = nn_cell(g, qz, 2);
    double best = INFINITY;
    int64_t bi = INT64_MAX;
    bool done = false;
    for (int r = 0; r < NG && r <= r_max; ++r) {
        const int z0 = cz - r, z1 = cz + r, y0 = cy - r, y1 = cy + r, x0 = cx - r, x1 = cx + r;
        for (int z = max(z0, 0); z <= min(z1, NG - 1); ++z)
            for (int y = max(y0, 0); y <= min(y1, NG - 1); ++y) {
                const bool face = (z == z0 || z == z1 || y == y0 || y == y1);
                const int xs = face ? 1 : (x1 - x0 > 0 ? x1 - x0 : 1);           // interior rows: only the two x faces
                for (int x = x0; x <= x1; x += xs) {
                    if (x < 0 || x >= NG) continue;
                    const int c = (z * NG + y) * NG + x;
                    for (int j = cell_start[c]; j < cell_start[c + 1]; ++j) {
                        double dx = qx - sxyz[j * 3], dy = qy - sxyz[j * 3 + 1], dz = qz - sxyz[j * 3 + 2];
                        double d2 = (dx * dx + dy * dy) + dz * dz;
                        int64_t id = sidx[j];
                        if (d2 < best || (d2 == best && id < bi)) { best = d2; bi = id; }
                    }
                }
            }
        const double bound = r * g.h * (1.0 - 1e-9);
        if (best <= bound * bound) { done = true; break; }
        if (x0 <= 0 && y0 <= 0 && z0 <= 0 && x1 >= NG - 1 && y1 >= NG - 1 && z1 >= NG - 1) { done = true; break; }   // whole grid scanned
    }
    nn[q] = done ? bi : -2;                                   // -2: not settled within r_max shells (next pass)
}

__global__ void nn_mark_pending_kernel(const int64_t *__restrict__ qidx, const int32_t *__restrict__ counts, int64_t *__restrict__ nn) {
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi < counts[1] && counts[0] > 0) nn[qidx[qi]] = -2;
}
// far queries (not settled within 2 fine shells): one WAVE per query on the coarse grid; the 64 lanes
// split the (z,y) cell rows of each shell and scan their points, then a wave reduction of (d2, id)
__global__ void __launch_bounds__(256)
nn_grid_query_wave_kernel(const float *__restrict__ xyz, const float *__restrict__ sxyz, const int64_t *__restrict__ sidx,
                          const int32_t *__restrict__ cell_start, const int64_t *__restrict__ qidx,
                          const int32_t *__restrict__ counts, const float *__restrict__ bb, int64_t *__restrict__ nn, int NG) {
    const int n_ref = counts[0], n_q = counts[1];
    const int qi = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (qi >= n_q || n_ref == 0) return;
    const int64_t q = qidx[qi];
    if (nn[q] != -2) return;                                   // wave-uniform
    NnGrid g; nn_grid_of(bb, g, NG);
    const double qx = xyz[q * 3], qy = xyz[q * 3 + 1], qz = xyz[q * 3 + 2];
    const int cx = nn_cell(g, qx, 0), cy = nn_cell(g, qy, 1), cz = nn_cell(g, qz, 2);
    double best = INFINITY;
    long long bi = INT64_MAX;
    __shared__ int s_cs[4][64], s_off[4][65];
    int *cs_ = s_cs[threadIdx.x >> 6], *off_ = s_off[threadIdx.x >> 6];
    for (int r = 0; r < NG; ++r) {
        const int z0 = cz - r, y0 = cy - r, x0 = cx - r, x1 = cx + r, side = 2 * r + 1;
        // the shell's cells, 64 at a time: every lane fetches one cell's point range (one round of loads), the ranges are
        // numbered flat by a wave prefix sum and the lanes then share the POINTS evenly (the former loop gave a lane a whole
        // row of cells: tens of dependent loads in sequence).  Same candidate set, same (d2, id) minimum.
        // shell cells in a fixed order: the two z faces (side^2 each), then per middle slab its ring of 4 side - 4 cells
        const int face = side * side, ring = 4 * side - 4;
        const int nshell = r == 0 ? 1 : 2 * face + (side - 2) * ring;
        for (int t0 = 0; t0 < nshell; t0 += 64) {
            const int t = t0 + lane;
            int c_start = 0, c_cnt = 0;
            if (t < nshell) {
                int dz, dy, dx;
                if (t < face) { dz = 0; dy = t / side; dx = t % side; }
                else if (t < 2 * face) { const int u = t - face; dz = side - 1; dy = u / side; dx = u % side; }
                else {
                    const int u = t - 2 * face, v = u % ring;
                    dz = 1 + u / ring;
                    if (v < side) { dy = 0; dx = v; }
                    else if (v < 2 * side) { dy = side - 1; dx = v - side; }
                    else { const int w = v - 2 * side; dy = 1 + (w >> 1); dx = (w & 1) ? side - 1 : 0; }
                }
                const int z = z0 + dz, y = y0 + dy, x = x0 + dx;
                if (z >= 0 && z < NG && y >= 0 && y < NG && x >= 0 && x < NG) {
                    const int c = (z * NG + y) * NG + x;
                    c_start = cell_start[c];
                    c_cnt = cell_start[c + 1] - c_start;
                }
            }
            int incl = c_cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                int v = __shfl_up(incl, o, 64);
                if (lane >= o) incl += v;
            }
            const int total_p = __shfl(incl, 63, 64);
            if (total_p == 0) continue;                              // wave-uniform
            cs_[lane] = c_start;
            off_[lane] = incl - c_cnt;
            if (lane == 63) off_[64] = total_p;
            gp_wave_sync();
            for (int j = lane; j < total_p; j += 64) {
                int lo = 0, hi = 63;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (off_[mid] <= j) lo = mid; else hi = mid - 1;
                }
                const int pj = cs_[lo] + (j - off_[lo]);
                double dx = qx - sxyz[pj * 3], dy = qy - sxyz[pj * 3 + 1], dz = qz - sxyz[pj * 3 + 2];
                double d2 = (dx * dx + dy * dy) + dz * dz;
                long long id = sidx[pj];
                if (d2 < best || (d2 == best && id < bi)) { best = d2; bi = id; }
            }
            gp_wave_sync();
        }
        // wave-wide lexicographic minimum (all lanes end up with the same value)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            double ob = __shfl_xor(best, o, 64);
            long long oi = __shfl_xor(bi, o, 64);
            if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        const double bound = r * g.h * (1.0 - 1e-9);
        if (best <= bound * bound) break;
        if (x0 <= 0 && y0 <= 0 && z0 <= 0 && x1 >= NG - 1 && cy + r >= NG - 1 && cz + r >= NG - 1) break;
    }
    if (lane == 0) nn[q] = bi;
}

// ---- ordered compaction of the visible points of one view: (point id, pixel row, pixel col)
__global__ void vis_flags_kernel(const int64_t *__restrict__ mapping, int64_t n, int32_t *__restrict__ f) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) f[i] = mapping[i * 3 + 2] != 0 ? 1 : 0;
}
__global__ void vis_compact_kernel(const int64_t *__restrict__ mapping, const int32_t *__restrict__ sc, int64_t n,
                                   int64_t *__restrict__ pt, int64_t *__restrict__ x, int64_t *__restrict__ y,
                                   int64_t *__restrict__ count) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool v = mapping[i * 3 + 2] != 0;
    if (v) { int p = sc[i]; pt[p] = i; x[p] = mapping[i * 3]; y[p] = mapping[i * 3 + 1]; }
    if (i == n - 1) *count = sc[i] + (v ? 1 : 0);
}

size_t scan32_tmp(int64_t n) {
    size_t t = 0;
    (void)rocprim::exclusive_scan(nullptr, t, (int32_t *)nullptr, (int32_t *)nullptr, (int32_t)0, (size_t)n, rocprim::plus<int32_t>(), 0);
    return t;
}

int nn1_splits(int64_t n_ref, int64_t n_q) {
    int64_t qblocks = (n_q + 255) / 256;
    int s = (int)(2048 / (qblocks > 0 ? qblocks : 1));
    if (s < 1) s = 1;
    int64_t maxs = (n_ref + NN_TILE - 1) / NN_TILE;
    if (s > maxs) s = (int)maxs;
    if (s > 64) s = 64;
    return s < 1 ? 1 : s;
}

// ------------------------------------------------------------------------------------------------
// classify: 16 lanes per point (4 points per wave); a lane keeps its share of the row (every 16th float4) in
// registers, normalised once.  logits_c = scale * <f/|f|, t_c>; first maximum wins.  Rows wider than 16*4*CL_MAXJ
// floats or not a multiple of 4 take the generic one-wave-per-point kernel below.
constexpr int CL_MAXJ = 8;                    // float4 per lane -> d <= 512
__device__ __forceinline__ float sum16(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__global__ void classify16_kernel(const float *__restrict__ feat, int64_t ld, int d, int64_t n, const float *__restrict__ text, int C,
                                  float scale, int64_t *__restrict__ pred, uint8_t *__restrict__ zero_row) {
    const int64_t p = ((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 4);
    const int l = threadIdx.x & 15;
    const bool live = p < n;
    const int nj = d / 64;                                 // float4 per lane (d % 64 == 0)
    float4 v[CL_MAXJ];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < CL_MAXJ; ++j) {
        v[j] = (live && j < nj) ? *reinterpret_cast<const float4 *>(feat + p * ld + (j * 16 + l) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        ss += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
    }
    ss = sum16(ss);
    const float nrm = fmaxf(sqrtf(ss), 1e-12f);
    float sa = 0.f;
#pragma unroll
    for (int j = 0; j < CL_MAXJ; ++j) {
        v[j].x /= nrm; v[j].y /= nrm; v[j].z /= nrm; v[j].w /= nrm;
        sa += fabsf(v[j].x) + fabsf(v[j].y) + fabsf(v[j].z) + fabsf(v[j].w);
    }
    sa = sum16(sa);
    float bestv = -INFINITY;
    int bestc = 0;
    for (int k = 0; k < C; ++k) {
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < CL_MAXJ; ++j)
            if (j < nj) {
                const float4 t = *reinterpret_cast<const float4 *>(text + (int64_t)k * d + (j * 16 + l) * 4);
                dot += v[j].x * t.x + v[j].y * t.y + v[j].z * t.z + v[j].w * t.w;
            }
        dot = sum16(dot) * scale;
        if (dot > bestv) { bestv = dot; bestc = k; }
    }
    if (live && l == 0) {
        pred[p] = bestc;
        if (zero_row) zero_row[p] = (sa == 0.f) ? 1 : 0;
    }
}

// the same arithmetic (same 16-lane split of the row, same sum16 order, same divisions => the same bits and labels), but the
// text matrix staged ONCE per workgroup in LDS (19 x 512 floats = 38 KiB do not fit the 32-KiB vector L1: the kernel above
// re-fetches them from L2 for every group of points) and two points per 16-lane group and class sweep
// GATHER (round 6: gp_gather_rows_classify): point p's row is row row_map[index[p]] of `feat` -- the final voxel -> point gather of
// affinity_module.py:1589 -- and is ALSO written to out[p]: the per-point feature matrix (307 MB at S) is written once and never read
// back by a separate classification pass.  Same loads per value, same arithmetic: the same rows, bits and labels as gather + classify.
constexpr int CL_Q = 2;
template <bool GATHER>
__global__ void __launch_bounds__(256)
classify16_lds_kernel(const float *__restrict__ feat, int64_t ld, int d, int64_t n, const float *__restrict__ text, int C,
                      float scale, int64_t *__restrict__ pred, uint8_t *__restrict__ zero_row, const int64_t *__restrict__ index,
                      const int32_t *__restrict__ row_map, float *__restrict__ out, int64_t ld_out) {
    extern __shared__ __align__(16) float cl_text[];        // [C][d]
    for (int i = threadIdx.x * 4; i < C * d; i += 256 * 4) *reinterpret_cast<float4 *>(cl_text + i) = *reinterpret_cast<const float4 *>(text + i);
    __syncthreads();
    const int l = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int nj = d / 64;
    for (int64_t base = blockIdx.x * (int64_t)(16 * CL_Q); base < n; base += (int64_t)gridDim.x * (16 * CL_Q)) {
        float4 v[CL_Q][CL_MAXJ];
        float sa[CL_Q], bestv[CL_Q];
        int bestc[CL_Q];
        bool live[CL_Q];
#pragma unroll
        for (int q = 0; q < CL_Q; ++q) {
            const int64_t p = base + g * CL_Q + q;
            live[q] = p < n;
            int64_t src_row = p;
            if constexpr (GATHER) {
                if (live[q]) { src_row = index[p]; if (row_map) src_row = row_map[src_row]; }
            }
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < CL_MAXJ; ++j) {
                v[q][j] = (live[q] && j < nj) ? *reinterpret_cast<const float4 *>(feat + src_row * ld + (j * 16 + l) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (GATHER) {
                    if (live[q] && j < nj) *reinterpret_cast<float4 *>(out + p * ld_out + (j * 16 + l) * 4) = v[q][j];
                }
                ss += v[q][j].x * v[q][j].x + v[q][j].y * v[q][j].y + v[q][j].z * v[q][j].z + v[q][j].w * v[q][j].w;
            }
            ss = sum16(ss);
            const float nrm = fmaxf(sqrtf(ss), 1e-12f);
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < CL_MAXJ; ++j) {
                v[q][j].x /= nrm; v[q][j].y /= nrm; v[q][j].z /= nrm; v[q][j].w /= nrm;
                a += fabsf(v[q][j].x) + fabsf(v[q][j].y) + fabsf(v[q][j].z) + fabsf(v[q][j].w);
            }
            sa[q] = sum16(a);
            bestv[q] = -INFINITY;
            bestc[q] = 0;
        }
        for (int k = 0; k < C; ++k) {
            float dot[CL_Q];
#pragma unroll
            for (int q = 0; q < CL_Q; ++q) dot[q] = 0.f;
#pragma unroll
            for (int j = 0; j < CL_MAXJ; ++j)
                if (j < nj) {
                    const float4 t = *reinterpret_cast<const float4 *>(cl_text + k * d + (j * 16 + l) * 4);
#pragma unroll
                    for (int q = 0; q < CL_Q; ++q) dot[q] += v[q][j].x * t.x + v[q][j].y * t.y + v[q][j].z * t.z + v[q][j].w * t.w;
                }
#pragma unroll
            for (int q = 0; q < CL_Q; ++q) {
                const float dv = sum16(dot[q]) * scale;
                if (dv > bestv[q]) { bestv[q] = dv; bestc[q] = k; }
            }
        }
#pragma unroll
        for (int q = 0; q < CL_Q; ++q)
            if (live[q] && l == 0) {
                const int64_t p = base + g * CL_Q + q;
                pred[p] = bestc[q];
                if (zero_row) zero_row[p] = (sa[q] == 0.f) ? 1 : 0;
            }
    }
}

// generic shape: one wave per point
__global__ void classify_kernel(const float *__restrict__ feat, int64_t ld, int d, int64_t n,
                                const float *__restrict__ text, int C, float scale, int64_t *__restrict__ pred,
                                uint8_t *__restrict__ zero_row) {
    int64_t p = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    if (p >= n) return;
    int lane = gp_lane();
    float ss = 0.f, sa = 0.f;
    for (int c = lane; c < d; c += 64) { float v = feat[p * ld + c]; ss += v * v; }
    ss = gp_wave_sum(ss);
    float nrm = fmaxf(sqrtf(ss), 1e-12f);
    for (int c = lane; c < d; c += 64) sa += fabsf(feat[p * ld + c] / nrm);
    sa = gp_wave_sum(sa);
    float bestv = -INFINITY;
    int bestc = 0;
    for (int k = 0; k < C; ++k) {
        float dot = 0.f;
        for (int c = lane; c < d; c += 64) dot += (feat[p * ld + c] / nrm) * text[(int64_t)k * d + c];
        dot = gp_wave_sum(dot) * scale;
        if (dot > bestv) { bestv = dot; bestc = k; }
    }
    if (lane == 0) {
        pred[p] = bestc;
        if (zero_row) zero_row[p] = (sa == 0.f) ? 1 : 0;
    }
}

// arg-max over the first C columns of precomputed logits rows (first maximum wins); optional zero-row
// flag of the feature rows (sum |f| == 0).  Used with the MFMA GEMM logits for large class counts.
__global__ void rows_argmax_kernel(const float *__restrict__ logits, int64_t ld, int C, int64_t n,
                                   const float *__restrict__ feat, int64_t ld_f, int d, int64_t *__restrict__ pred,
                                   uint8_t *__restrict__ zero_row) {
    int64_t p = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    if (p >= n) return;
    int lane = gp_lane();
    float bv = -INFINITY;
    int bc = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
        float v = logits[p * ld + c];
        if (v > bv) { bv = v; bc = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(bv, o, 64);
        int oc = __shfl_xor(bc, o, 64);
        if (ov > bv || (ov == bv && oc < bc)) { bv = ov; bc = oc; }
    }
    float sa = 0.f;
    if (zero_row && feat) {
        for (int c = lane; c < d; c += 64) sa += fabsf(feat[p * ld_f + c]);
        sa = gp_wave_sum(sa);
    }
    if (lane == 0) {
        pred[p] = bc == 0x7fffffff ? 0 : bc;
        if (zero_row && feat) zero_row[p] = (sa == 0.f) ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------------
__global__ void iou_hist_kernel(const int64_t *__restrict__ pred, const int64_t *__restrict__ target, int64_t n, int C,
                                const int64_t ig0, const int64_t ig1, const int64_t ig2, const int64_t ig3, int nig,
                                unsigned long long *__restrict__ counts) {
    extern __shared__ unsigned int sh[];                           // [3*C]
    for (int i = threadIdx.x; i < 3 * C; i += blockDim.x) sh[i] = 0;
    __syncthreads();
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t p = pred[i], t = target[i];
        if ((nig > 0 && t == ig0) || (nig > 1 && t == ig1) || (nig > 2 && t == ig2) || (nig > 3 && t == ig3)) p = t;
        if (p == t && p >= 0 && p < C) atomicAdd(&sh[p], 1u);
        if (p >= 0 && p < C) atomicAdd(&sh[C + p], 1u);
        if (t >= 0 && t < C) atomicAdd(&sh[2 * C + t], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * C; i += blockDim.x)
        if (sh[i]) atomicAdd(&counts[i], (unsigned long long)sh[i]);
}

}  // namespace

extern "C" size_t gp_nn1_workspace_bytes(int64_t n_ref, int64_t n_query) {
    if (n_ref <= 0 || n_query <= 0) return 0;
    int s = nn1_splits(n_ref, n_query);
    GpCarver cv(nullptr, 0);
    cv.take<double>((size_t)s * n_query);
    cv.take<int64_t>((size_t)s * n_query);
    return cv.off;
}

extern "C" int gp_nn1_f64(const float *ref_xyz, int64_t n_ref, const float *query_xyz, int64_t n_query, int64_t *nn,
                          void *workspace, size_t workspace_bytes, void *stream_) {
    GP_CHECK_ARG(ref_xyz && query_xyz && nn && workspace, "gp_nn1_f64: null argument");
    GP_CHECK_ARG(n_ref > 0, "gp_nn1_f64: empty reference set (the reference's KDTree raises here too)");
    if (n_query == 0) return GP_OK;
    int s = nn1_splits(n_ref, n_query);
    GpCarver cv(workspace, workspace_bytes);
    double *pd = cv.take<double>((size_t)s * n_query);
    int64_t *pi = cv.take<int64_t>((size_t)s * n_query);
    if (!cv.ok()) { gp_set_error("gp_nn1_f64: workspace too small (%zu < %zu)", workspace_bytes, cv.off); return GP_ENOMEM; }
    hipStream_t st = gp_stream(stream_);
    dim3 grid((unsigned)((n_query + 255) / 256), (unsigned)s);
    nn1_partial_kernel<<<grid, 256, 0, st>>>(ref_xyz, n_ref, query_xyz, n_query, s, pd, pi);
    nn1_reduce_kernel<<<(int)((n_query + 255) / 256), 256, 0, st>>>(pd, pi, n_query, s, nn);
    GP_CHECK_LAUNCH();
    return GP_OK;
}


extern "C" size_t gp_nn1_masked_workspace_bytes(int64_t n) {
    if (n <= 0) return 0;
    GpCarver cv(nullptr, 0);
    cv.take<int32_t>(n); cv.take<int32_t>(n); cv.take<int32_t>(n); cv.take<int32_t>(n);
    cv.take<float>(3 * n); cv.take<int64_t>(n); cv.take<int64_t>(n); cv.take<int32_t>(64);
    cv.take<char>(scan32_tmp(n > (int64_t)NGMAX * NGMAX * NGMAX + 1 ? n : (int64_t)NGMAX * NGMAX * NGMAX + 1));
    // grid path
    cv.take<float>(8); cv.take<int32_t>(n); cv.take<int32_t>((size_t)NGMAX * NGMAX * NGMAX + 1); cv.take<int32_t>((size_t)NGMAX * NGMAX * NGMAX + 1);
    cv.take<int32_t>((size_t)NGMAX * NGMAX * NGMAX + 1); cv.take<float>(3 * n); cv.take<int64_t>(n);
    cv.take<float>(3 * n); cv.take<int64_t>(n); cv.take<int32_t>(3 * (NG2MAX * NG2MAX * NG2MAX + 1));
    return cv.off;
}

extern "C" int gp_nn1_masked_f64(const float *xyz, int64_t n, const uint8_t *ref_mask, const uint8_t *query_mask,
                                 int64_t *nn, void *workspace, size_t workspace_bytes, void *stream_) {
    GP_CHECK_ARG(xyz && ref_mask && query_mask && nn && workspace, "gp_nn1_masked_f64: null argument");
    GP_CHECK_ARG(n > 0 && n < (1ll << 31), "gp_nn1_masked_f64: n=%lld out of range", (long long)n);
    GpCarver cv(workspace, workspace_bytes);
    int32_t *rf = cv.take<int32_t>(n), *qf = cv.take<int32_t>(n), *rs = cv.take<int32_t>(n), *qs = cv.take<int32_t>(n);
    float *rxyz = cv.take<float>(3 * n);
    int64_t *ridx = cv.take<int64_t>(n), *qidx = cv.take<int64_t>(n);
    int32_t *counts = cv.take<int32_t>(64);
    const int64_t ncell_max = (int64_t)NGMAX * NGMAX * NGMAX;
    int NG = g_gp_knobs[6] > 0 ? g_gp_knobs[6] : 128;
    if (NG > NGMAX) NG = NGMAX;
    const int64_t ncell = (int64_t)NG * NG * NG;
    size_t tb = scan32_tmp(n > ncell_max + 1 ? n : ncell_max + 1);
    char *tmp = cv.take<char>(tb);
    float *bb = cv.take<float>(8);
    int32_t *rcell = cv.take<int32_t>(n);
    int32_t *cell_cnt = cv.take<int32_t>(ncell_max + 1), *cell_start = cv.take<int32_t>(ncell_max + 1), *cursor = cv.take<int32_t>(ncell_max + 1);
    float *sxyz = cv.take<float>(3 * n);
    int64_t *sidx = cv.take<int64_t>(n);
    float *sxyz2 = cv.take<float>(3 * n);
    int64_t *sidx2 = cv.take<int64_t>(n);
    int32_t *cells2 = cv.take<int32_t>(3 * (NG2MAX * NG2MAX * NG2MAX + 1));
    if (!cv.ok()) { gp_set_error("gp_nn1_masked_f64: workspace too small (%zu < %zu)", workspace_bytes, cv.off); return GP_ENOMEM; }
    hipStream_t st = gp_stream(stream_);
    int blocks = (int)((n + 255) / 256);
    mask_flags_kernel<<<blocks, 256, 0, st>>>(ref_mask, query_mask, n, rf, qf);
    size_t t = tb;
    GP_CHECK_HIP(rocprim::exclusive_scan(tmp, t, rf, rs, (int32_t)0, (size_t)n, rocprim::plus<int32_t>(), st));
    t = tb;
    GP_CHECK_HIP(rocprim::exclusive_scan(tmp, t, qf, qs, (int32_t)0, (size_t)n, rocprim::plus<int32_t>(), st));
    mask_compact_kernel<<<blocks, 256, 0, st>>>(xyz, ref_mask, query_mask, rs, qs, n, rxyz, ridx, qidx, counts, nn);
    if (n >= 32768 && !g_gp_knobs[5]) {
        // grid path: bbox of all points; references bucketed by cell on a coarse 32^3 grid, one wave per query scanning cubic
        // shells of cells (optionally near queries first on a fine grid, one thread each: knob 7)
        nn_bbox_init_kernel<<<1, 64, 0, st>>>(bb);
        nn_bbox_kernel<<<blocks < 64 ? blocks : 64, 256, 0, st>>>(xyz, n, bb);       // 256 threads: the LDS reduce assumes 4 waves
        auto bucket = [&](int ng, int32_t *cnt, int32_t *start, int32_t *cur, float *sx, int64_t *si) -> int {
            int64_t nc = (int64_t)ng * ng * ng;
            GP_CHECK_HIP(hipMemsetAsync(cnt, 0, (nc + 1) * sizeof(int32_t), st));
            GP_CHECK_HIP(hipMemsetAsync(cur, 0, (nc + 1) * sizeof(int32_t), st));
            nn_cell_count_kernel<<<blocks, 256, 0, st>>>(rxyz, counts, bb, cnt, rcell, ng);
            size_t tt = tb;
            GP_CHECK_HIP(rocprim::exclusive_scan(tmp, tt, cnt, start, (int32_t)0, (size_t)(nc + 1), rocprim::plus<int32_t>(), st));
            nn_cell_fill_kernel<<<blocks, 256, 0, st>>>(rxyz, ridx, counts, rcell, start, cur, sx, si);
            return GP_OK;
        };
        // every query on the coarse grid, one wave each (default); knob 7 = 2 brings back the two-level search (near queries by
        // one thread each on a fine grid first): 0.39 vs 0.26 ms on the S scene since the wave kernel enumerates points flat
        const bool wave_only = g_gp_knobs[7] != 2;
        int rc = wave_only ? GP_OK : bucket(NG, cell_cnt, cell_start, cursor, sxyz, sidx);
        if (rc) return rc;
        int NG2 = g_gp_knobs[13] > 0 ? g_gp_knobs[13] : 32;           // cells per axis of the coarse grid (tuning aid: knob 13)
        if (NG2 > NG2MAX) NG2 = NG2MAX;
        const int nc2 = NG2 * NG2 * NG2 + 1;
        rc = bucket(NG2, cells2, cells2 + nc2, cells2 + 2 * nc2, sxyz2, sidx2);
        if (rc) return rc;
        if (wave_only) nn_mark_pending_kernel<<<blocks, 256, 0, st>>>(qidx, counts, nn);
        else nn_grid_query_kernel<<<blocks, 256, 0, st>>>(xyz, sxyz, sidx, cell_start, qidx, counts, bb, nn, NG, 2, 0);
        nn_grid_query_wave_kernel<<<(unsigned)((n * 64 + 255) / 256), 256, 0, st>>>(xyz, sxyz2, sidx2, cells2 + nc2, qidx, counts, bb, nn, NG2);
    } else {
        // the cell arrays of the grid path are idle here: partial results of the reference ranges live in them
        double *part_d = reinterpret_cast<double *>(cell_cnt);
        int32_t *part_i = reinterpret_cast<int32_t *>(cell_cnt) + 2 * (int64_t)NN_CHUNKS * n;
        static_assert((size_t)NN_CHUNKS * 32768 * 12 <= ((size_t)NGMAX * NGMAX * NGMAX + 1) * 4 * 2, "partials must fit the cell arrays");
        nn1_masked_part_kernel<<<dim3((unsigned)blocks, NN_CHUNKS), 256, 0, st>>>(xyz, rxyz, qidx, counts, part_d, part_i, n);
        nn1_masked_reduce_kernel<<<blocks, 256, 0, st>>>(part_d, part_i, n, ridx, qidx, counts, nn);
    }
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" size_t gp_visible_lists_workspace_bytes(int64_t n) {
    if (n <= 0) return 0;
    GpCarver cv(nullptr, 0);
    cv.take<int32_t>(n); cv.take<int32_t>(n); cv.take<char>(scan32_tmp(n));
    return cv.off;
}

extern "C" int gp_visible_lists(const int64_t *mapping, int64_t n, int64_t *pt, int64_t *x, int64_t *y,
                                int64_t *count_dev, void *workspace, size_t workspace_bytes, void *stream_) {
    GP_CHECK_ARG(mapping && pt && x && y && count_dev && workspace && n > 0, "gp_visible_lists: null/empty argument");
    GpCarver cv(workspace, workspace_bytes);
    int32_t *f = cv.take<int32_t>(n), *sc = cv.take<int32_t>(n);
    size_t tb = scan32_tmp(n);
    char *tmp = cv.take<char>(tb);
    if (!cv.ok()) { gp_set_error("gp_visible_lists: workspace too small"); return GP_ENOMEM; }
    hipStream_t st = gp_stream(stream_);
    int blocks = (int)((n + 255) / 256);
    vis_flags_kernel<<<blocks, 256, 0, st>>>(mapping, n, f);
    GP_CHECK_HIP(rocprim::exclusive_scan(tmp, tb, f, sc, (int32_t)0, (size_t)n, rocprim::plus<int32_t>(), st));
    vis_compact_kernel<<<blocks, 256, 0, st>>>(mapping, sc, n, pt, x, y, count_dev);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" int gp_classify_argmax(const float *feat, int64_t ld, int32_t d, int64_t n, const float *text_norm, int32_t c,
                                  float logit_scale, int64_t *pred, uint8_t *zero_row, void *stream_) {
    GP_CHECK_ARG(feat && text_norm && pred && n > 0 && d > 0 && c > 0, "gp_classify_argmax: null/empty argument");
    const bool fast = d % 64 == 0 && d <= 64 * CL_MAXJ && ld % 4 == 0 && (uintptr_t)feat % 16 == 0 && (uintptr_t)text_norm % 16 == 0;
    const size_t text_bytes = (size_t)c * d * sizeof(float);
    if (fast && text_bytes <= 64 * 1024 && !g_gp_knobs[14]) {
        GP_SMEM_ATTR(classify16_lds_kernel<false>, 64 * 1024);
        const int64_t groups = (n + 16 * CL_Q - 1) / (16 * CL_Q);
        classify16_lds_kernel<false><<<(int)(groups < 2048 ? groups : 2048), 256, text_bytes, gp_stream(stream_)>>>(
            feat, ld, d, n, text_norm, c, logit_scale, pred, zero_row, nullptr, nullptr, nullptr, 0);
    } else if (fast)
        classify16_kernel<<<(int)((n * 16 + 255) / 256), 256, 0, gp_stream(stream_)>>>(feat, ld, d, n, text_norm, c, logit_scale, pred, zero_row);
    else
        classify_kernel<<<(int)((n * 64 + 255) / 256), 256, 0, gp_stream(stream_)>>>(feat, ld, d, n, text_norm, c, logit_scale,
                                                                                   pred, zero_row);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

// out[p, 0:d] = src[row_map[index[p]], 0:d] (gp_gather_rows) AND pred / zero_row of gp_classify_argmax on those rows, in one pass
extern "C" int gp_gather_rows_classify(const float *src, int64_t ld_src, int32_t d, const int64_t *index, int64_t n, const int32_t *row_map,
                                       float *out, int64_t ld_out, const float *text_norm, int32_t c, float logit_scale, int64_t *pred,
                                       uint8_t *zero_row, void *stream_) {
    GP_CHECK_ARG(src && index && out && text_norm && pred && n > 0 && d > 0 && c > 0, "gp_gather_rows_classify: null/empty argument");
    GP_CHECK_ARG(d % 64 == 0 && d <= 64 * CL_MAXJ && ld_src % 4 == 0 && ld_out % 4 == 0 && ld_out >= d && (uintptr_t)src % 16 == 0 &&
                     (uintptr_t)out % 16 == 0 && (uintptr_t)text_norm % 16 == 0 && (size_t)c * d * sizeof(float) <= 64 * 1024,
                 "gp_gather_rows_classify: d=%d must be a multiple of 64 up to %d, rows 16-byte aligned, c * d * 4 <= 64 KiB (use gp_gather_rows + "
                 "gp_classify_argmax otherwise)", d, 64 * CL_MAXJ);
    GP_CHECK_ARG(out != src, "gp_gather_rows_classify: out must not alias src");
    GP_SMEM_ATTR(classify16_lds_kernel<true>, 64 * 1024);
    const int64_t groups = (n + 16 * CL_Q - 1) / (16 * CL_Q);
    classify16_lds_kernel<true><<<(int)(groups < 2048 ? groups : 2048), 256, (size_t)c * d * sizeof(float), gp_stream(stream_)>>>(
        src, ld_src, d, n, text_norm, c, logit_scale, pred, zero_row, index, row_map, out, ld_out);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" int gp_rows_argmax(const float *logits, int64_t ld, int32_t c, int64_t n, const float *feat, int64_t ld_f,
                              int32_t d, int64_t *pred, uint8_t *zero_row, void *stream_) {
    GP_CHECK_ARG(logits && pred && n > 0 && c > 0, "gp_rows_argmax: null/empty argument");
    rows_argmax_kernel<<<(int)((n * 64 + 255) / 256), 256, 0, gp_stream(stream_)>>>(logits, ld, c, n, feat, ld_f, d, pred, zero_row);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" int gp_iou_hist_i64(const int64_t *pred, const int64_t *target, int64_t n, int32_t num_classes,
                               const int64_t *ignore_ids_host, int32_t num_ignore, int64_t *counts, void *stream_) {
    GP_CHECK_ARG(pred && target && counts && n > 0, "gp_iou_hist_i64: null/empty argument");
    GP_CHECK_ARG(num_classes > 0 && num_classes <= 4096, "gp_iou_hist_i64: num_classes=%d out of range", num_classes);
    GP_CHECK_ARG(num_ignore >= 0 && num_ignore <= 4, "gp_iou_hist_i64: at most 4 ignore ids");
    int64_t ig[4] = {0, 0, 0, 0};
    for (int i = 0; i < num_ignore; ++i) ig[i] = ignore_ids_host[i];
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    iou_hist_kernel<<<blocks, 256, 3 * num_classes * sizeof(unsigned int), gp_stream(stream_)>>>(
        pred, target, n, num_classes, ig[0], ig[1], ig[2], ig[3], num_ignore,
        reinterpret_cast<unsigned long long *>(counts));
    GP_CHECK_LAUNCH();
    return GP_OK;
}
