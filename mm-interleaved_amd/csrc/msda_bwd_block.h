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
constexpr int kMaxTileCells = 4096;     // cells per sort tile (two counter arrays of 16 KiB)
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
    int n_blocks4, pad[3];              // 4x4 blocks of all levels (matrix-core reduce)
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
constexpr int kTileChunk = 1024;
constexpr int kTileLanes = 8;             // queue lanes of the extra work items (one per XCD, keyed by h % 8)

struct TileHeader {
    uint32_t n_extra[kTileLanes];         // queued extra items per lane
    uint32_t n_partials, cap_extra, cap_partials, pad;
    uint4 null_rec;                       // what a K-step reads past the end of a list: weight 0, far outside
    uint4 zero_row[32];                   // 512 zero bytes: the grad_out row of such a record
};
struct TileInfo { uint32_t parts, pbase; };                 // per (b, h, block)
struct TileItem { uint32_t bh, blk, part, pidx; };          // one extra work item (part >= 1)

struct TileReduceArgs {
    const uint4 *records;                 // cell-sorted {query, y, x, attention}
    const uint2 *celltab;                 // [B, H, cell_stride] {first record, count}
    const CellHeader *hdr;                // level rows (device)
    int cell_stride;
    TileHeader *th;
    TileInfo *tinfo;                      // [B, H, blocks_bound]
    TileItem *titems;                     // [kTileLanes, cap_extra]
    float *tpartials;                     // [cap_partials, kTB*kTB, D]
    int blocks_bound;
};
// 16-bit storage, D in {32, 64, 128}; MMFS_VALUE_ALGO=block keeps the vector-ALU reduce
bool tile_reduce_supported(int dtype, const Dims &d);
hipError_t tile_reduce(int dtype, const void *grad_out, void *grad_value, const TileReduceArgs &a, const Dims &d,
                       hipStream_t st);

}  // namespace blk
}  // namespace mmfs
