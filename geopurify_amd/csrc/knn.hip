// Row 10: exact (K+1)-nearest voxels on the integer lattice, canonical (d^2, id) order, self dropped.
//
// One wave per query.  Candidates come from the (2R+1)^3 block of 8^3-voxel cells around the query
// (contiguous runs in the Morton-ordered voxel array), which contains every voxel with
// d^2 < B = (8R+1)^2.  Squared distances are small integers, so selection is a counting problem:
//   pass 1  LDS histogram of d^2 (< B)           -> threshold T = d^2 of the (K+1)-th smallest
//   pass 2  emit d^2 < T; collect the ties d^2 == T and keep the smallest ids among them
//   pass 3  rank-sort the K+1 winners by (d^2, id), drop rank 0 (the query itself)
// Queries whose (K+1)-th neighbour is not provably inside the block (or with too many ties) are
// appended to a retry list for the next, larger ring; the last resort is an exhaustive scan with a
// bisection on the (d^2, id) key, so the result is exact for every input.
#include "gp_grid.h"

namespace {

constexpr int KNN_MAXSEL = 128;   // K+1 <= 128
constexpr int KNN_MAXTIE = 256;    // ties at the threshold distance kept in LDS (more: the query retries on the next ring / exhaustively)

template <int R>
struct KnnCfg {
    static constexpr int B = (8 * R + 1) * (8 * R + 1);
    static constexpr int HB = (B + 63) / 64 * 64;
};

template <int R, int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
knn_ring_kernel(const void *grid, const int32_t *__restrict__ coords, const int32_t *__restrict__ ids, int64_t nv,
                int k, int32_t *__restrict__ nbr, const int32_t *__restrict__ qlist,
                const int32_t *__restrict__ qcount, int32_t *__restrict__ fail_list, int32_t *__restrict__ fail_count) {
    constexpr int B = KnnCfg<R>::B, HB = KnnCfg<R>::HB;
    __shared__ int s_hist[WAVES][HB];
    __shared__ unsigned long long s_selkey[WAVES][KNN_MAXSEL];
    __shared__ int s_selrow[WAVES][KNN_MAXSEL];
    __shared__ int s_tieid[WAVES][KNN_MAXTIE];
    __shared__ int s_tierow[WAVES][KNN_MAXTIE];
    __shared__ int s_cnt[WAVES][2];
    __shared__ int s_cstart[WAVES][(2 * R + 1) * (2 * R + 1) * (2 * R + 1)], s_coff[WAVES][(2 * R + 1) * (2 * R + 1) * (2 * R + 1) + 1];

    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int64_t widx = (int64_t)blockIdx.x * WAVES + wv;
    int64_t total = qlist ? (int64_t)*qcount : nv;
    if (widx >= total) return;
    const int qi = __builtin_amdgcn_readfirstlane(qlist ? qlist[widx] : (int)widx);
    GpGridView g(grid);
    const int qx = coords[(int64_t)qi * 3], qy = coords[(int64_t)qi * 3 + 1], qz = coords[(int64_t)qi * 3 + 2];
    const int cx0 = (qx - g.h->origin[0]) >> 3, cy0 = (qy - g.h->origin[1]) >> 3, cz0 = (qz - g.h->origin[2]) >> 3;
    int *hist = s_hist[wv];
    for (int b = lane; b < HB; b += 64) hist[b] = 0;
    if (lane < 2) s_cnt[wv][lane] = 0;
    // ---- candidate table: the (2R+1)^3 cells' (first row, rows before) in LDS, fetched with all lanes at once -- one round
    // of dependent loads (cell index -> record) instead of one per cell -- and a flat candidate numbering 0 .. total-1
    constexpr int SIDE = 2 * R + 1, NC = SIDE * SIDE * SIDE;
    int *cstart = s_cstart[wv], *coff = s_coff[wv];
    int total_c = 0;
    for (int c0 = 0; c0 < NC; c0 += 64) {
        const int c = c0 + lane;
        int start = 0, cnt = 0;
        if (c < NC) {
            const int dx = c % SIDE - R, dy = (c / SIDE) % SIDE - R, dz = c / (SIDE * SIDE) - R;
            const int slot = g.cell_slot(cx0 + dx, cy0 + dy, cz0 + dz);
            if (slot >= 0) { start = g.recs[slot].start; cnt = g.recs[slot].count; }
        }
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (c < NC) { cstart[c] = start; coff[c] = total_c + incl - cnt; }
        total_c += __shfl(incl, 63, 64);
    }
    if (lane == 0) coff[NC] = total_c;
    gp_wave_sync();
    // candidate j -> row: the cell whose range holds j (binary search over the NC + 1 offsets in LDS)
    auto row_of = [&](int j) {
        int lo = 0, hi = NC - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (coff[mid] <= j) lo = mid; else hi = mid - 1;
        }
        return cstart[lo] + (j - coff[lo]);
    };

    // ---- pass 1: histogram of d^2 over the candidates (two per lane and round: both coordinate loads in flight together)
    for (int j0 = 0; j0 < total_c; j0 += 128) {
        const int ja = j0 + lane, jb = j0 + 64 + lane;
        const bool va = ja < total_c, vb = jb < total_c;
        const int64_t ra = va ? row_of(ja) : 0, rb = vb ? row_of(jb) : 0;
        const int ax = coords[ra * 3], ay = coords[ra * 3 + 1], az = coords[ra * 3 + 2];
        const int bx = coords[rb * 3], by = coords[rb * 3 + 1], bz = coords[rb * 3 + 2];
        const int da = (ax - qx) * (ax - qx) + (ay - qy) * (ay - qy) + (az - qz) * (az - qz);
        const int db = (bx - qx) * (bx - qx) + (by - qy) * (by - qy) + (bz - qz) * (bz - qz);
        if (va && da < B) atomicAdd(&hist[da], 1);
        if (vb && db < B) atomicAdd(&hist[db], 1);
    }
    gp_wave_sync();

    // ---- threshold: smallest T with cum(<=T) >= K+1
    constexpr int CH = HB / 64;
    int loc = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) loc += hist[lane * CH + c];
    int incl = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    int tot = __shfl(incl, 63, 64);
    const int need = k + 1;
    bool fail = tot < need;
    int T = 0, c_lt = 0;
    if (!fail) {
        int excl = incl - loc;
        unsigned long long m = __ballot(incl >= need);
        int owner = __ffsll((long long)m) - 1;
        int myT = 0, mylt = 0;
        if (lane == owner) {
            int run = excl;
            for (int c = 0; c < CH; ++c) {
                int h = hist[lane * CH + c];
                if (run + h >= need) { myT = lane * CH + c; mylt = run; break; }
                run += h;
            }
        }
        T = __shfl(myT, owner, 64);
        c_lt = __shfl(mylt, owner, 64);
        if (hist[T] > KNN_MAXTIE) fail = true;
    }
    if (fail) {
        if (lane == 0) {
            int p = atomicAdd(fail_count, 1);
            fail_list[p] = qi;
        }
        return;
    }

    // ---- pass 2: emit winners below T, collect ties at T
    for (int j0 = 0; j0 < total_c; j0 += 64) {
        const int j = j0 + lane;
        if (j < total_c) {
            const int64_t r = row_of(j);
            int ex = coords[r * 3] - qx, ey = coords[r * 3 + 1] - qy, ez = coords[r * 3 + 2] - qz;
            int d2 = ex * ex + ey * ey + ez * ez;
            if (d2 < T) {
                int id = ids ? ids[r] : (int)r;
                int p = atomicAdd(&s_cnt[wv][0], 1);
                s_selkey[wv][p] = ((unsigned long long)(unsigned)d2 << 32) | (unsigned)id;
                s_selrow[wv][p] = (int)r;
            } else if (d2 == T) {
                int id = ids ? ids[r] : (int)r;
                int p = atomicAdd(&s_cnt[wv][1], 1);
                s_tieid[wv][p] = id;
                s_tierow[wv][p] = (int)r;
            }
        }
    }
    gp_wave_sync();
    const int m_tie = s_cnt[wv][1];
    const int take = need - c_lt;                  // ties to keep: the `take` smallest ids
    for (int t = lane; t < m_tie; t += 64) {
        int id = s_tieid[wv][t];
        int rank = 0;
        for (int u = 0; u < m_tie; ++u) rank += (s_tieid[wv][u] < id);
        if (rank < take) {
            s_selkey[wv][c_lt + rank] = ((unsigned long long)(unsigned)T << 32) | (unsigned)id;
            s_selrow[wv][c_lt + rank] = s_tierow[wv][t];
        }
    }
    gp_wave_sync();

    // ---- pass 3: rank sort of the K+1 winners, drop rank 0 (self)
    for (int t = lane; t < need; t += 64) {
        unsigned long long key = s_selkey[wv][t];
        int rank = 0;
        for (int u = 0; u < need; ++u) rank += (s_selkey[wv][u] < key);
        if (rank > 0) nbr[(int64_t)qi * k + rank - 1] = s_selrow[wv][t];
    }
}

// Measured and left out (round 6; `knn_cell_kernel`, not in the tree): the first ring as one WORKGROUP PER 8^3 CELL -- the 27-cell block staged
// once in LDS (packed coordinates + row ids, <= 2048 candidates) and the cell's ~25-60 queries answered from there by the same histogram /
// threshold / tie / rank steps, the same results (tests) -- to stop re-reading ~19 KB of candidates from L2 twice per query (5 GB per S
// scene: this kernel runs at the L2 ceiling).  0.88 ms with four waves and 48 KiB of LDS per workgroup, 0.54 ms with eight waves and
// 36 KiB (32 waves per CU) against 0.45 ms here: per query the LDS atomics of the histogram, the rank sort and the staging of sparse
// cells' blocks cost more than the L2 reads they replace.
// exhaustive fallback: one 256-thread block per failed query, bisection on the 64-bit (d2,id) key
__global__ void __launch_bounds__(256)
knn_exhaustive_kernel(const int32_t *__restrict__ coords, const int32_t *__restrict__ ids, int64_t nv, int k,
                      int32_t *__restrict__ nbr, const int32_t *__restrict__ qlist, const int32_t *__restrict__ qcount) {
    __shared__ int s_red[4];
    __shared__ unsigned long long s_selkey[KNN_MAXSEL];
    __shared__ int s_selrow[KNN_MAXSEL];
    __shared__ int s_n;
    const int total = *qcount;
    for (int w = blockIdx.x; w < total; w += gridDim.x) {
        const int qi = qlist[w];
        const int qx = coords[(int64_t)qi * 3], qy = coords[(int64_t)qi * 3 + 1], qz = coords[(int64_t)qi * 3 + 2];
        const int need = k + 1;
        auto keyof = [&](int64_t r) {
            long long ex = coords[r * 3] - qx, ey = coords[r * 3 + 1] - qy, ez = coords[r * 3 + 2] - qz;
            unsigned long long d2 = (unsigned long long)(ex * ex + ey * ey + ez * ez);
            unsigned id = (unsigned)(ids ? ids[r] : (int)r);
            return (d2 << 32) | id;      // extents are < 2^15 (gp_grid_build), so d2 < 2^32
        };
        // smallest key value t such that count(key <= t) >= need, by bisection over 64 bits
        unsigned long long lo = 0, hi = ~0ull;
        while (lo < hi) {
            unsigned long long mid = lo + ((hi - lo) >> 1);
            int c = 0;
            for (int64_t r = threadIdx.x; r < nv; r += 256) c += (keyof(r) <= mid);
            c = gp_wave_sum_i(c);
            __syncthreads();
            if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = c;
            __syncthreads();
            int tot = s_red[0] + s_red[1] + s_red[2] + s_red[3];
            if (tot >= need) hi = mid; else lo = mid + 1;
        }
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
        for (int64_t r = threadIdx.x; r < nv; r += 256) {
            unsigned long long key = keyof(r);
            if (key <= lo) {
                int p = atomicAdd(&s_n, 1);
                if (p < KNN_MAXSEL) { s_selkey[p] = key; s_selrow[p] = (int)r; }
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < need; t += 256) {
            unsigned long long key = s_selkey[t];
            int rank = 0;
            for (int u = 0; u < need; ++u) rank += (s_selkey[u] < key);
            if (rank > 0) nbr[(int64_t)qi * k + rank - 1] = s_selrow[t];
        }
        __syncthreads();
    }
}

__global__ void zero2_kernel(int32_t *a, int32_t *b) {
    if (threadIdx.x == 0) { *a = 0; *b = 0; }
}

}  // namespace

extern "C" size_t gp_knn_workspace_bytes(int64_t nv) {
    GpCarver cv(nullptr, 0);
    cv.take<int32_t>(64);
    cv.take<int32_t>(nv);
    cv.take<int32_t>(nv);
    return cv.off;
}

extern "C" int gp_knn_lattice(const void *grid, const int32_t *coords, const int32_t *ids, int64_t nv, int32_t k,
                              int32_t *nbr, void *workspace, size_t workspace_bytes, void *stream_) {
    GP_CHECK_ARG(grid && coords && nbr && workspace, "gp_knn_lattice: null argument");
    GP_CHECK_ARG(k >= 1 && k <= GP_KNN_MAX_K, "gp_knn_lattice: k=%d not in 1..%d", k, GP_KNN_MAX_K);
    GP_CHECK_ARG(nv > k, "gp_knn_lattice: need more than k voxels (nv=%lld, k=%d)", (long long)nv, k);
    GpCarver cv(workspace, workspace_bytes);
    int32_t *counts = cv.take<int32_t>(64);
    int32_t *list_a = cv.take<int32_t>(nv);
    int32_t *list_b = cv.take<int32_t>(nv);
    if (!cv.ok()) { gp_set_error("gp_knn_lattice: workspace too small (%zu < %zu)", workspace_bytes, cv.off); return GP_ENOMEM; }
    hipStream_t s = gp_stream(stream_);
    zero2_kernel<<<1, 64, 0, s>>>(counts, counts + 1);
    constexpr int W1 = 4, W3 = 2;
    // ring 1: all queries
    knn_ring_kernel<1, W1><<<(int)((nv + W1 - 1) / W1), W1 * 64, 0, s>>>(grid, coords, ids, nv, k, nbr, nullptr, nullptr,
                                                                      list_a, counts);
    // ring 3: failures of ring 1 (grid sized for the worst case; surplus waves exit immediately)
    knn_ring_kernel<3, W3><<<(int)((nv + W3 - 1) / W3), W3 * 64, 0, s>>>(grid, coords, ids, nv, k, nbr, list_a, counts,
                                                                      list_b, counts + 1);
    knn_exhaustive_kernel<<<1024, 256, 0, s>>>(coords, ids, nv, k, nbr, list_b, counts + 1);
    GP_CHECK_LAUNCH();
    return GP_OK;
}
