// Lattice grid over Morton-ordered voxels: 8x8x8-voxel cells, each occupied cell has a record
// {start row, count, 512-bit occupancy bitmap in local-Morton order}.  Because the voxels are
// sorted by Morton code, a cell's voxels are contiguous and ordered by their 9-bit local code, so
// the row of voxel (x,y,z) is  start + popcount(bitmap below its local code): an exact O(1) lookup
// that serves both the 27-offset kernel map and kNN candidate enumeration (row 10).
#pragma once
#include "gp_common.h"

struct GpGridHeader {
    int32_t origin[3];
    int32_t extent[3];
    int32_t cdim[3];
    int32_t status;      // bit0: input not strictly Morton-increasing (unsorted or duplicate voxels)
    int32_t ncells;      // occupied cells (atomic counter during build)
    int32_t max_cells;
    int64_t nv;
    int64_t cell_index_off;   // byte offsets from the grid base
    int64_t records_off;
};

struct GpCellRec {
    int32_t start;
    int32_t count;
    uint32_t bits[16];
};

__host__ __device__ __forceinline__ uint32_t gp_spread3_10(uint32_t v) {
    // spread the low 10 bits of v so that there are two zero bits between each
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__host__ __device__ __forceinline__ uint64_t gp_morton3(uint32_t x, uint32_t y, uint32_t z) {
    // 21 bits per axis -> 63-bit code (x lowest)
    uint64_t lo = gp_spread3_10(x) | (gp_spread3_10(y) << 1) | (gp_spread3_10(z) << 2);
    uint64_t mid = gp_spread3_10(x >> 10) | (gp_spread3_10(y >> 10) << 1) | (gp_spread3_10(z >> 10) << 2);
    uint64_t hi = gp_spread3_10(x >> 20) | (gp_spread3_10(y >> 20) << 1) | (gp_spread3_10(z >> 20) << 2);
    return lo | (mid << 30) | (hi << 60);
}
__device__ __forceinline__ uint32_t gp_local9(int x, int y, int z) {
    return (uint32_t)gp_morton3(x & 7, y & 7, z & 7);
}

struct GpGridView {
    const GpGridHeader *h;
    const int32_t *cell_index;
    const GpCellRec *recs;
    __device__ __forceinline__ GpGridView(const void *grid) {
        h = reinterpret_cast<const GpGridHeader *>(grid);
        const char *b = reinterpret_cast<const char *>(grid);
        cell_index = reinterpret_cast<const int32_t *>(b + h->cell_index_off);
        recs = reinterpret_cast<const GpCellRec *>(b + h->records_off);
    }
    // record slot of cell (cx,cy,cz) or -1
    __device__ __forceinline__ int cell_slot(int cx, int cy, int cz) const {
        if ((unsigned)cx >= (unsigned)h->cdim[0] || (unsigned)cy >= (unsigned)h->cdim[1] ||
            (unsigned)cz >= (unsigned)h->cdim[2])
            return -1;
        return cell_index[((int64_t)cz * h->cdim[1] + cy) * h->cdim[0] + cx];
    }
    // row of the voxel at relative coordinates (rx,ry,rz) or -1
    __device__ __forceinline__ int lookup_rel(int rx, int ry, int rz) const {
        if ((unsigned)rx >= (unsigned)h->extent[0] || (unsigned)ry >= (unsigned)h->extent[1] ||
            (unsigned)rz >= (unsigned)h->extent[2])
            return -1;
        int slot = cell_slot(rx >> 3, ry >> 3, rz >> 3);
        if (slot < 0) return -1;
        const GpCellRec &r = recs[slot];
        uint32_t l = gp_local9(rx, ry, rz);
        uint32_t wi = l >> 5, bi = l & 31;
        uint32_t word = r.bits[wi];
        if (!((word >> bi) & 1u)) return -1;
        int idx = r.start + __popc(word & ((1u << bi) - 1u));
        for (uint32_t j = 0; j < wi; ++j) idx += __popc(r.bits[j]);
        return idx;
    }
};
