// msda_bwd_block.h -- structures shared by the two generations of the block-stationary grad_value
// kernels: the cell sort + 2x2-block reduce on the vector ALUs (msda_bwd_block.hip) and the 4x4-block
// reduce on the matrix cores (msda_bwd_tile.hip).
#pragma once
#include "msda_device.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mmfs {
namespace blk {

constexpr int kThreads = 1024;          // sort: 16 waves per workgroup
constexpr int kWaves = kThreads / 64;
constexpr int kMaxTileCells = 5120;     // cells per sort tile (two counter arrays of 20 KiB): a 64x64 level (65x65 cells) is ONE tile
constexpr int kScanUnroll = 4;
constexpr int kMaxLevels = 128;         // levels the block path takes
constexpr int kLdsLevels = 32;          // level rows the reduce keeps in LDS (beyond: read from the table)
#ifndef MMFS_BLK_ROUND
#define MMFS_BLK_ROUND 2
#endif
constexpr int kRoundPx = MMFS_BLK_ROUND; // pixels per LDS round when split blocks add up their parts

struct CTile {
    int level, Hl, Wl, cbase;           // cbase: the level's first entry in the cell table
    int ya, yb, xa, xb;                 // cell coordinates [ya, yb) x [xa, xb)
};

struct LevelRow {
    int Hl, Wl, lstart, cbase;
    int bbase, nbx, nby, split;         // first (virtual) block index, blocks per row / column, lane groups per block
    int cap;                            // records of a block's list walked in place before the rest is queued
    int bbase4, nbx4, nby4;             // matrix-core reduce: first 4x4 block of the level, blocks per row / column
};
constexpr int kMaxSplit = 8;            // <= lane groups per reduce workgroup for every head width
#ifndef MMFS_BLK_H
#define MMFS_BLK_H 2
#endif
#ifndef MMFS_BLK_W
#define MMFS_BLK_W 2
#endif
constexpr int kBH = MMFS_BLK_H, kBW = MMFS_BLK_W;      // pixels of a block (rows x columns)
constexpr int kNC = (kBH + 1) * (kBW + 1);              // cells whose footprints touch a block
constexpr int kNPX = kBH * kBW;
static_assert(kNPX % 4 == 0, "block weights travel as 16-byte vectors");

struct CellHeader {
    int n_tiles, n_blocks, n_cells, L;
    int n_blocks4, pad[3];              // 4x4 blocks of all levels (matrix-core reduce); pad[0]: every row has exactly one owner level
};

// workspace table: CellHeader | LevelRow[L] | CTile[cap]
__device__ __host__ inline LevelRow *level_rows(CellHeader *h) { return reinterpret_cast<LevelRow *>(h + 1); }
__device__ __host__ inline const LevelRow *level_rows(const CellHeader *h) { return reinterpret_cast<const LevelRow *>(h + 1); }
__device__ __host__ inline CTile *tiles_of(CellHeader *h, int L) { return reinterpret_cast<CTile *>(level_rows(h) + L); }
__device__ __host__ inline const CTile *tiles_of(const CellHeader *h, int L) { return reinterpret_cast<const CTile *>(level_rows(h) + L); }



// ---------------------------------------------------------------- matrix-core reduce (msda_bwd_tile.hip)
// Blocks of kTB x kTB pixels; their index space lives in LevelRow::bbase4 / nbx4 / nby4 and
// CellHeader::n_blocks4.  A block's list of records is cut into work items of kTileChunk records;
// a block of more than one item leaves fp32 partial tiles that a last small kernel adds up.
constexpr int kTB = 4;
#ifndef MMFS_TILE_CHUNK
#define MMFS_TILE_CHUNK 1024
#endif
constexpr int kTileChunk = MMFS_TILE_CHUNK;
constexpr int kTileLanes = 8;             // queue lanes of the extra work items (one per XCD, keyed by h % 8)

struct TileHeader {
    uint32_t n_extra[kTileLanes];         // queued extra items per lane
    uint32_t n_partials, cap_extra, cap_partials, n_multi;
    uint4 null_rec;                       // (spare)
    uint4 zero_row[32];                   // (spare)
};
// What a work item needs to know about its block, written by the plan kernel (one 64-byte scalar load):
// the five record runs (one per cell row), the level's extent and first pixel, the block's position.
struct TileDesc {
    int first[5];                         // first record of each run (index into the record list)
    int cnt[5];                           // records per run
    uint32_t hw;                          // Hl << 16 | Wl
    int lstart;
    uint32_t byx;                         // block row << 16 | block column
    uint32_t parts, pbase;                // work items of the block; first partial tile when parts > 1
    uint32_t arrived;                     // items of a block of several that have left their partial tile (the last one adds them up)
};
static_assert(sizeof(TileDesc) == 64, "one 64-byte line per block");
struct TileItem { uint32_t bh, blk, part, pidx; };          // one extra work item (part >= 1)

struct TileReduceArgs {
    const uint4 *records;                 // cell-sorted {query, y, x, attention}
    const uint2 *celltab;                 // [B, H, cell_stride] {first record, count}
    const CellHeader *hdr;                // level rows (device)
    int cell_stride;
    TileHeader *th;
    TileDesc *tdesc;                      // [B, H, blocks_bound]
    TileItem *titems;                     // [kTileLanes, cap_extra]
    uint32_t *slice_done;                 // [B, H] sort workgroups of the slice that have finished (zeroed with the cursors)
    float *tpartials;                     // [cap_partials, kTB*kTB, D]
    int blocks_bound;
};
constexpr uint32_t kVoidPart = 0xffffffffu;

#ifdef __HIPCC__
// The blocks of one (b, h) slice: descriptor (record runs, position) and, from the length of the list,
// the number of work items; the extra ones are queued (one queue lane per XCD, keyed by h like every
// other kernel's head -> XCD affinity).  Run by the LAST sort workgroup of the slice (the cell table
// it reads was written by the slice's sort workgroups, which share an XCD and so an L2): thread `tid`
// of `nthreads` takes blocks tid, tid + nthreads, ...
__device__ inline void plan_slice_blocks(const TileReduceArgs &a, const Dims &d, int64_t bh, int tid, int nthreads)
{
    const LevelRow *lv = level_rows(a.hdr);
    const int nblk = a.hdr->n_blocks4;
    for (int blk = tid; blk < nblk; blk += nthreads) {
        int level = 0;
        while (level + 1 < d.L && blk >= lv[level + 1].bbase4) ++level;
        while (level < d.L && lv[level].nbx4 * lv[level].nby4 == 0) ++level;        // (empty levels own no block)
        TileDesc td;
#pragma unroll
        for (int r = 0; r < 5; ++r) { td.first[r] = 0; td.cnt[r] = 0; }
        td.hw = 0; td.lstart = 0; td.byx = 0; td.parts = 1; td.pbase = 0; td.arrived = 0;
        if (level < d.L) {
            const LevelRow lr = lv[level];
            const int rel = blk - lr.bbase4, by = rel / lr.nbx4, bx = rel - by * lr.nbx4;
            const uint2 *tab = a.celltab + bh * a.cell_stride + lr.cbase;
            uint2 ent[kTB + 1][kTB + 1];
#pragma unroll
            for (int dy = 0; dy <= kTB; ++dy)
#pragma unroll
                for (int dx = 0; dx <= kTB; ++dx) {
                    const int cy = kTB * by + dy, cx = kTB * bx + dx;
                    ent[dy][dx] = make_uint2(0u, 0u);
                    if (cy <= lr.Hl && cx <= lr.Wl) {
                        const unsigned long long e = __hip_atomic_load(
                            reinterpret_cast<const unsigned long long *>(&tab[cy * (lr.Wl + 1) + cx]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ent[dy][dx] = make_uint2((uint32_t)e, (uint32_t)(e >> 32));
                    }
                }
            int64_t n = 0;
#pragma unroll
            for (int dy = 0; dy <= kTB; ++dy) {
                td.first[dy] = (int)ent[dy][0].x;               // (the cells of a row are one contiguous run)
                int c = 0;
#pragma unroll
                for (int dx = 0; dx <= kTB; ++dx) c += (int)ent[dy][dx].y;
                td.cnt[dy] = c;
                n += c;
            }
            td.hw = ((uint32_t)lr.Hl << 16) | (uint32_t)lr.Wl;
            td.lstart = lr.lstart;
            td.byx = ((uint32_t)by << 16) | (uint32_t)bx;
            const uint32_t parts = (uint32_t)((n + kTileChunk - 1) / kTileChunk);
            if (parts > 1) {
                const int ql = (int)(bh % d.H) % kTileLanes;
                const uint32_t pb = atomicAdd(&a.th->n_partials, parts);
                const uint32_t eb = atomicAdd(&a.th->n_extra[ql], parts - 1);
                if (pb + parts <= a.th->cap_partials && eb + parts - 1 <= a.th->cap_extra) {
                    td.parts = parts; td.pbase = pb;
                    for (uint32_t p = 1; p < parts; ++p) {
                        TileItem ti;
                        ti.bh = (uint32_t)bh; ti.blk = (uint32_t)blk; ti.part = p; ti.pidx = pb + p;
                        a.titems[(size_t)ql * a.th->cap_extra + eb + p - 1] = ti;
                    }
                } else if (eb < a.th->cap_extra) {
                    // reserved queue entries that cannot be used must read as "nothing to do"
                    for (uint32_t p = 1; p < parts && eb + p - 1 < a.th->cap_extra; ++p) {
                        TileItem ti;
                        ti.bh = (uint32_t)bh; ti.blk = (uint32_t)blk; ti.part = kVoidPart; ti.pidx = 0;
                        a.titems[(size_t)ql * a.th->cap_extra + eb + p - 1] = ti;
                    }
                }
            }
        }
        a.tdesc[bh * a.blocks_bound + blk] = td;
    }
}
#endif

// 16-bit storage, D in {32, 64, 128}; MMFS_VALUE_ALGO=block keeps the vector-ALU reduce
bool tile_reduce_supported(int dtype, const Dims &d);
hipError_t tile_reduce(int dtype, const void *grad_out, void *grad_value, const TileReduceArgs &a, const Dims &d,
                       hipStream_t st);

}  // namespace blk
}  // namespace mmfs
