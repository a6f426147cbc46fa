// Row 12, operator build: a row order for the POOLING operator only (round 6).
//
// The column-sliced pooling kernel (pool_mfma_cs.hip) gathers, per block of 128 consecutive rows, the union of the rows' K nearest
// neighbours.  On a surface scan a block of 128 Morton-consecutive voxels is an irregular, often elongated patch: its union holds 4.75
// rows per output row.  A compact patch of 128 voxels (about 11 x 11 on the surface, neighbour radius ~5.5) has (11 + 11)^2 / 128 = 3.9:
// fewer union rows to gather per block, fewer steps per tile (profiles/r06_pool_combined_price.log: -4 .. -6 % of a launch).
//
// gp_rcb_order re-numbers the rows INSIDE chunks of `chunk_rows` Morton-consecutive rows by recursive coordinate bisection: a segment
// longer than `leaf_rows` is sorted along the axis of its largest extent (ties: the lower axis; equal coordinates keep their order) and
// cut at ceil(len / 2 / leaf) * leaf rows (len / 2 if that is the whole segment), so that every leaf but a chunk's last holds exactly
// `leaf_rows` rows and leaves coincide with the operator's row blocks.  One workgroup per chunk, everything in LDS: coordinates by row,
// the current order, one 32-bit key per position -- segment (5 bits) | coordinate along the segment's axis (15 bits) | position (12 bits) --
// sorted by a bitonic network once per level.  Nothing else on the path sees this order: the voxel arrays stay in Morton order (the lattice
// grid, the kernel map and the kNN search need its 8^3 cells contiguous); the pooling operator, its feature planes and the embedding
// planes of the affinity kernel are written through the map, and the final voxel -> point gather composes it with the Morton rank.
#include "gp_common.h"

namespace {

constexpr int RCB_MAX_SEG = 32;

template <int CH, int NT>
__global__ void __launch_bounds__(NT)
rcb_chunk_kernel(const int32_t *__restrict__ coords /*[nv, 3]*/, int64_t nv, int leaf, int32_t *__restrict__ sigma, int32_t *__restrict__ rho) {
    __shared__ int s_c[3][CH];                 // coordinates by row of the chunk
    __shared__ unsigned s_key[CH];
    __shared__ unsigned short s_perm[2][CH];   // position -> row of the chunk
    __shared__ unsigned char s_seg[CH];        // position -> segment
    __shared__ int s_start[RCB_MAX_SEG + 1], s_len[RCB_MAX_SEG], s_axis[RCB_MAX_SEG], s_lo[RCB_MAX_SEG][3], s_hi[RCB_MAX_SEG][3];
    __shared__ int s_nseg, s_more;
    const int tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * CH;
    const int n = (int)((nv - base) < CH ? (nv - base) : CH);
    for (int i = tid; i < CH; i += NT) {
        const int64_t r = base + (i < n ? i : n - 1);
        s_c[0][i] = coords[r * 3];
        s_c[1][i] = coords[r * 3 + 1];
        s_c[2][i] = coords[r * 3 + 2];
        s_perm[0][i] = (unsigned short)i;
    }
    if (tid == 0) { s_nseg = 1; s_start[0] = 0; s_len[0] = n; s_more = n > leaf; }
    __syncthreads();
    int cur = 0, level = 0;
    while (s_more) {
        const int nseg = s_nseg;
        // ---- per segment: bounding box -> axis
        for (int i = tid; i < nseg * 3; i += NT) { s_lo[i / 3][i % 3] = 0x7fffffff; s_hi[i / 3][i % 3] = -0x7fffffff; }
        for (int i = tid; i < CH; i += NT) {
            int sg = 0;
            if (i < n) {
                int lo = 0, hi = nseg - 1;                       // last segment with start <= i
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_start[mid] <= i) lo = mid; else hi = mid - 1; }
                sg = lo;
            }
            s_seg[i] = (unsigned char)sg;
        }
        __syncthreads();
        for (int i = tid; i < n; i += NT) {
            const int sg = s_seg[i];
            if (s_len[sg] > leaf) {
                const int r = s_perm[cur][i];
#pragma unroll
                for (int a = 0; a < 3; ++a) { atomicMin(&s_lo[sg][a], s_c[a][r]); atomicMax(&s_hi[sg][a], s_c[a][r]); }
            }
        }
        __syncthreads();
        if (tid < nseg) {
            int ax = 0, best = -1;
            if (s_len[tid] > leaf) {
#pragma unroll
                for (int a = 0; a < 3; ++a) { const int e = s_hi[tid][a] - s_lo[tid][a]; if (e > best) { best = e; ax = a; } }
            } else {
                ax = -1;                                          // a leaf: keeps its order
            }
            s_axis[tid] = ax;
        }
        __syncthreads();
        // ---- keys: segment | coordinate along the segment's axis | position; padding sorts behind everything
        for (int i = tid; i < CH; i += NT) {
            unsigned k = 0xffffffffu;
            if (i < n) {
                const int sg = s_seg[i], ax = s_axis[sg];
                unsigned c = 0;
                if (ax >= 0) c = (unsigned)(s_c[ax][s_perm[cur][i]] - s_lo[sg][ax]) & 0x7fffu;
                k = ((unsigned)sg << 27) | (c << 12) | (unsigned)i;
            }
            s_key[i] = k;
        }
        __syncthreads();
        // In a FULL chunk the segments of level L are the aligned blocks of CH >> L positions (every cut is at half a power-of-two
        // segment): the network only has to sort inside those blocks -- merges up to `limit`, the last one ascending in every block.
        // (A chunk's tail -- the scene's last chunk -- has irregular segments: the whole array is sorted, the segment id leads the key.)
        const int limit = (n == CH) ? (CH >> level) : CH;
        for (int kk = 2; kk <= limit; kk <<= 1)
            for (int j = kk >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < CH; i += NT) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const unsigned x = s_key[i], y = s_key[ixj];
                        const bool up = (kk == limit) || ((i & kk) == 0);
                        if ((x > y) == up) { s_key[i] = y; s_key[ixj] = x; }
                    }
                }
                __syncthreads();
            }
        ++level;
        for (int i = tid; i < n; i += NT) s_perm[cur ^ 1][i] = s_perm[cur][s_key[i] & 0xfffu];
        cur ^= 1;
        // ---- cut the segments that were sorted
        if (tid == 0) {
            int ns = 0, more = 0;
            int st[RCB_MAX_SEG], ln[RCB_MAX_SEG];
            for (int sg = 0; sg < nseg; ++sg) {
                const int len = s_len[sg], start = s_start[sg];
                if (len > leaf && ns + 2 <= RCB_MAX_SEG) {
                    int half = ((len / 2 + leaf - 1) / leaf) * leaf;
                    if (half >= len) half = len / 2;
                    st[ns] = start; ln[ns] = half; ++ns;
                    st[ns] = start + half; ln[ns] = len - half; ++ns;
                    more |= (half > leaf) | (len - half > leaf);
                } else {
                    st[ns] = start; ln[ns] = len; ++ns;
                }
            }
            for (int sg = 0; sg < ns; ++sg) { s_start[sg] = st[sg]; s_len[sg] = ln[sg]; }
            s_nseg = ns;
            s_more = more && ns < RCB_MAX_SEG;
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += NT) {
        const int r = s_perm[cur][i];
        sigma[base + i] = (int32_t)(base + r);
        rho[base + r] = (int32_t)(base + i);
    }
}

// out[p, j] = rho[nbr[sigma[p], j]]: the neighbour lists in the new numbering, rows in the new order
__global__ void rows_renumber_kernel(const int32_t *__restrict__ nbr, int64_t nv, int k, const int32_t *__restrict__ sigma,
                                     const int32_t *__restrict__ rho, int32_t *__restrict__ out) {
    const int64_t total = nv * k;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / k;
        const int j = (int)(i - p * k);
        out[i] = rho[nbr[(int64_t)sigma[p] * k + j]];
    }
}

}  // namespace

extern "C" int gp_rcb_order(const int32_t *coords, int64_t nv, int32_t chunk_rows, int32_t leaf_rows, int32_t *sigma, int32_t *rho,
                            void *stream_) {
    GP_CHECK_ARG(coords && sigma && rho && nv > 0, "gp_rcb_order: null/empty argument");
    GP_CHECK_ARG(chunk_rows == 1024 || chunk_rows == 2048, "gp_rcb_order: chunk_rows=%d (1024 or 2048: a chunk lives in one workgroup's LDS)", chunk_rows);
    GP_CHECK_ARG(leaf_rows >= 16 && chunk_rows % leaf_rows == 0 && chunk_rows / leaf_rows <= RCB_MAX_SEG,
                 "gp_rcb_order: leaf_rows=%d must divide the chunk into at most %d leaves", leaf_rows, RCB_MAX_SEG);
    GP_CHECK_ARG(nv < ((int64_t)1 << 31), "gp_rcb_order: row ids are 32-bit");
    hipStream_t s = gp_stream(stream_);
    const unsigned blocks = (unsigned)((nv + chunk_rows - 1) / chunk_rows);
    if (chunk_rows == 1024) rcb_chunk_kernel<1024, 256><<<blocks, 256, 0, s>>>(coords, nv, leaf_rows, sigma, rho);
    else rcb_chunk_kernel<2048, 512><<<blocks, 512, 0, s>>>(coords, nv, leaf_rows, sigma, rho);
    GP_CHECK_LAUNCH();
    return GP_OK;
}

extern "C" int gp_rows_renumber_i32(const int32_t *nbr, int64_t nv, int32_t k, const int32_t *sigma, const int32_t *rho, int32_t *out,
                                    void *stream_) {
    GP_CHECK_ARG(nbr && sigma && rho && out && nv > 0 && k > 0 && out != nbr, "gp_rows_renumber_i32: null/empty argument (out must not alias nbr)");
    rows_renumber_kernel<<<2048, 256, 0, gp_stream(stream_)>>>(nbr, nv, k, sigma, rho, out);
    GP_CHECK_LAUNCH();
    return GP_OK;
}
