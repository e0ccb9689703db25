// msda_cache.h -- "coarse levels live in LDS".
//
// Measured on MI355X: the row gathers of this op run against the
// vector-memory path (64 B/clk/CU), not against HBM -- the forward moves 24x its
// compulsory bytes through it.  LDS reads are 4x faster (ds_read_b128: 256 B/clk/CU) and
// CDNA4 has 160 KiB of it per CU.  A (batch, head) slice of the small levels is tiny
// (16x16 + 8x8 pixels x 256 B = 80 KiB at the north-star shape) yet receives a full
// 1/L share of the taps each.  So a workgroup that works through many queries of one
// (b, h) first copies every level that fits a byte budget into LDS, once, and serves
// those levels' taps with ds_read_b128; only the big levels go to L1/L2.
//
// Which levels are cached is decided on the device (the level table is device memory in
// the reference API), identically by every workgroup: all levels up to the largest
// pixel count whose running total fits, then as many levels of the next size as still
// fit, in index order.
#pragma once
#include "msda_device.h"

namespace mmfs {

constexpr int kMaxCacheLevels = 128;          // more levels than this: plain kernels
constexpr uint32_t kCachedBit = 0x40000000u;  // tap offset refers to the LDS cache

struct LevelInfo {
    int Hl, Wl, start;
    int lds_off;          // byte offset of the level's first pixel row in the cache, or -1
};

// Fills lvl[0..L) (LDS).  cache byte 0..row_bytes-1 is reserved for an all-zero row that
// stands in for corners outside the map.  Returns the bytes used (uniform).
template <int THREADS>
__device__ int plan_level_cache(const int64_t *__restrict__ shapes, const int64_t *__restrict__ start,
                                int L, int row_bytes, int budget_bytes, LevelInfo *lvl, int *scratch2)
{
    const int tid = threadIdx.x;
    for (int l = tid; l < L; l += THREADS) {
        LevelInfo e;
        e.Hl = (int)shapes[2 * l]; e.Wl = (int)shapes[2 * l + 1]; e.start = (int)start[l]; e.lds_off = -1;
        lvl[l] = e;
    }
    if (tid == 0) { scratch2[0] = 0; scratch2[1] = 0x7fffffff; }
    __syncthreads();
    const int avail = budget_bytes - row_bytes;
    // largest pixel-count threshold T such that all levels with <= T pixels fit together
    for (int c = tid; c < L; c += THREADS) {
        const int T = lvl[c].Hl * lvl[c].Wl;
        long long sum = 0;
        for (int j = 0; j < L; ++j) {
            const int px = lvl[j].Hl * lvl[j].Wl;
            if (px <= T) sum += (long long)px * row_bytes;
        }
        if (T > 0 && sum <= avail) atomicMax(&scratch2[0], T);
    }
    __syncthreads();
    const int T1 = scratch2[0];
    for (int c = tid; c < L; c += THREADS) {           // the next size up, for leftovers
        const int px = lvl[c].Hl * lvl[c].Wl;
        if (px > T1) atomicMin(&scratch2[1], px);
    }
    __syncthreads();
    const int T2 = scratch2[1];
    __syncthreads();
    if (tid == 0) {
        int off = row_bytes;
        for (int l = 0; l < L; ++l) {
            const int px = lvl[l].Hl * lvl[l].Wl;
            if (px > 0 && px <= T1) { lvl[l].lds_off = off; off += px * row_bytes; }
        }
        for (int l = 0; l < L; ++l) {
            const int px = lvl[l].Hl * lvl[l].Wl;
            if (px == T2 && (long long)off + (long long)px * row_bytes <= budget_bytes) {
                lvl[l].lds_off = off; off += px * row_bytes;
            }
        }
        scratch2[0] = off;
    }
    __syncthreads();
    return scratch2[0];
}

// Copies the cached levels' rows of one (b, h) slab into the cache (16-byte vectors).
template <int THREADS>
__device__ void fill_level_cache(const char *__restrict__ slab, int64_t pixel_stride_bytes, int row_bytes,
                                 int L, const LevelInfo *lvl, uint4 *cache)
{
    const int tid = threadIdx.x;
    const int vec_per_row = row_bytes / 16;
    for (int i = tid; i < vec_per_row; i += THREADS) cache[i] = make_uint4(0u, 0u, 0u, 0u);
    for (int l = 0; l < L; ++l) {
        const int off = lvl[l].lds_off;
        if (off < 0) continue;
        const int n = lvl[l].Hl * lvl[l].Wl * vec_per_row;
        const char *src = slab + (int64_t)lvl[l].start * pixel_stride_bytes;
        for (int i = tid; i < n; i += THREADS) {
            const int px = i / vec_per_row, v = i % vec_per_row;
            cache[off / 16 + i] = *reinterpret_cast<const uint4 *>(src + (int64_t)px * pixel_stride_bytes + v * 16);
        }
    }
}

}  // namespace mmfs
