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
    int band;                           // cell rows per sort tile when the level's tiles are whole rows, else 0
};
constexpr int kMaxSplit = 8;            // <= lane groups per reduce workgroup for every head width

// How a level's (H+1) x (W+1) cells are cut into sort tiles: R rows x C columns of cells per tile, n tiles
// (n = 0: the level owns no tile -- empty, or an extent the sorted backward refuses).  Whole cell rows
// whenever one fits a tile's counters: the cells of a row that touch a block are then one contiguous run of
// the record list (the matrix-core reduce relies on it).  At least nt_min tiles (few (b, h, level) slices).
struct LevelTiling { int R, C, n; };
__device__ __host__ inline LevelTiling level_tiling(int64_t Hl, int64_t Wl, int nt_min)
{
    LevelTiling t;
    t.R = 0; t.C = 0; t.n = 0;
    if (Hl <= 0 || Wl <= 0 || Hl >= 65536 || Wl >= 65536) return t;
    const int Hc = (int)Hl + 1, Wc = (int)Wl + 1;
    const int64_t cells = (int64_t)Hc * Wc;
    int64_t nt = (cells + kMaxTileCells - 1) / kMaxTileCells;
    if (nt < nt_min) nt = nt_min;
    if (nt > cells) nt = cells;
    const int64_t tc = (cells + nt - 1) / nt;
    if (Wc <= kMaxTileCells) {
        int64_t r = tc / Wc;
        if (r < 1) r = 1;
        if (r > kMaxTileCells / Wc) r = kMaxTileCells / Wc;
        t.R = (int)r; t.C = Wc;
    } else {
        t.R = 1; t.C = (int)(tc < kMaxTileCells ? tc : kMaxTileCells);
    }
    const int64_t n = (int64_t)((Hc + t.R - 1) / t.R) * ((Wc + t.C - 1) / t.C);
    t.n = (int)(n < 0x3fffffff ? n : 0x3fffffff);
    return t;
}
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
    int n_blocks4, pad[3];              // 4x4 blocks of all levels (matrix-core reduce); pad[0]: every row has exactly one owner level;
                                        // pad[1]: levels cut into more than one sort tile (their seams' blocks are planned by the slice's last workgroup)
    uint32_t stamp, reserved;           // what the plan was made for (header_stamp): a sort / reduce handed another call's workspace finds nothing to do
    const void *loc_src, *attn_src;     // the op's own loc / attn when the sort reads them in place (no re-pack), else null
};

// The dimensions a plan belongs to, folded into 32 bits (never 0).  The staged entry points let a caller run the sort
// or the reduce on a workspace that was prepared by another call: the header then carries another
// stamp -- or none -- and the kernels return without touching anything instead of following stale pointers.
__device__ __host__ inline uint32_t header_stamp(const Dims &d)
{
    uint32_t h = 2166136261u;
    const int v[7] = {d.B, d.S, d.H, d.D, d.L, d.Nq, d.P};
    for (int i = 0; i < 7; ++i) { h ^= (uint32_t)v[i]; h *= 16777619u; }
    return h ? h : 1u;
}

// workspace table: CellHeader | LevelRow[L] | CTile[cap]
__device__ __host__ inline LevelRow *level_rows(CellHeader *h) { return reinterpret_cast<LevelRow *>(h + 1); }
__device__ __host__ inline const LevelRow *level_rows(const CellHeader *h) { return reinterpret_cast<const LevelRow *>(h + 1); }
__device__ __host__ inline CTile *tiles_of(CellHeader *h, int L) { return reinterpret_cast<CTile *>(level_rows(h) + L); }
__device__ __host__ inline const CTile *tiles_of(const CellHeader *h, int L) { return reinterpret_cast<const CTile *>(level_rows(h) + L); }



// ---------------------------------------------------------------- matrix-core reduce (msda_bwd_tile.hip)
// Blocks of kTB x kTB pixels; their index space lives in LevelRow::bbase4 / nbx4 / nby4 and
// CellHeader::n_blocks4.  A block's list of records is cut into work items of tile_chunk(D) records;
// a block of more than one item leaves fp32 partial tiles that a last small kernel adds up.
constexpr int kTB = 4;
#ifndef MMFS_TILE_CHUNK
#define MMFS_TILE_CHUNK 768      // (r04zu: 640 / 768 / 896 / 1024 / 1280 / 1536 -> north star's reduce 136 / 134 / 135 / 137 / 138 / 137 us)
#endif
// (records per work item: MMFS_TILE_CHUNK for heads of 128 channels; narrower heads move half / a quarter of the bytes
// per record, so their items hold twice / four times the records -- the same 192 KB of grad_out rows per item.  Measured,
// r03r: 2048 at D = 128 costs the north star 18 us of reduce, 1024 at D = 64 costs the LLM shape 45 MB of fp32 partial
// tiles each way -- and 20 us of the NEXT forward, whose inputs they push out of the memory-side cache)
// ... and a launch with FEW records gets shorter items: with the decoders' one-image geometry (4.9 M record visits, 7 000
// items for 3 584 wave slots) the items of 2048 records are the long poles of the last round -- reduce 75 -> 62 us at half
// the length (r04zt); from ~8 M visits on (4 images: 19.6 M, the image decoder's block: 26 M) the full length stands.
__host__ __device__ inline int tile_chunk(const Dims &d)
{
    const int full = MMFS_TILE_CHUNK * (d.D >= 128 ? 1 : d.D >= 64 ? 2 : 4);
    const long long visits = (long long)d.B * d.H * d.Nq * d.K * 25 / 16;
    return (d.D == 64 && visits < 8000000LL) ? full / 2 : full;          // (heads of 32 channels: no difference either way)
}

struct TileHeader {
    uint32_t spare0[8];
    uint32_t n_partials, cap_extra, cap_partials, n_multi;       // cap_extra: queue places per (b, h) slice
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
// grad_loc / grad_attn from the same cell-sorted records (msda_bwd_taps_sorted.hip): every cell -- every sample -- has ONE
// owner block there: the block whose pixels hold the cell's own pixel (cy, cx) -> block (cy / 4, cx / 4), the last block
// row / column also taking the border cells cy = Hl / cx = Wl.  The owned cells of a run are its first ones (a run is
// the cells 4 bx .. 4 bx + 4 of one cell row, in that order), so a block's owned records are five prefixes of its runs.
struct TapsDesc {
    int ocnt[5];                          // owned records of each run (run 4: only in the last block row)
    int level;
    int pad[2];
};
static_assert(sizeof(TapsDesc) == 32, "two per 64-byte line");

struct TileReduceArgs {
    const uint4 *records;                 // cell-sorted {query, y, x, attention}
    const uint2 *celltab;                 // [B, H, cell_stride] {first record, count}
    const CellHeader *hdr;                // level rows (device)
    int cell_stride;
    TileHeader *th;
    TileDesc *tdesc;                      // [B, H, blocks_bound]
    TileItem *titems;                     // [B, H, cap_extra]: a slice's extra items are walked next to its blocks (one XCD, one L2: the
                                          // slice's grad_out rows; a queue shared by the whole batch read 8 slices at once per XCD -- r03q)
    uint32_t *slice_done;                 // [B, H] sort workgroups of the slice that have finished (zeroed with the cursors)
    uint32_t *n_extra;                    // [B, H] queued extra items of the slice (zeroed with the cursors)
    float *tpartials;                     // [cap_partials, kTB*kTB, D]
    int blocks_bound;
    // grad_loc / grad_attn on the sorted records (Dims::taps_sorted; else null / 0)
    TapsDesc *xdesc;                      // [B, H, blocks_bound]
    void *g_loc, *g_attn;                 // the op's grad_loc / grad_attn: the sort writes the zeros of samples that get no record
    int qshift;                           // a record's low 16 bits are query * P + point: query = bits >> qshift (P = 1 << qshift)
};
constexpr uint32_t kVoidPart = 0xffffffffu;

#ifdef __HIPCC__
// A block's descriptor is complete but for its work items: from the length n of its list, the number of
// items; the extra ones are queued, slice by slice.
__device__ inline void queue_block(const TileReduceArgs &a, const Dims &d, int64_t bh, int blk, TileDesc &td, int64_t n)
{
    const uint32_t parts = (uint32_t)((n + tile_chunk(d) - 1) / tile_chunk(d));
    if (parts > 1) {
        const uint32_t pb = atomicAdd(&a.th->n_partials, parts);
        const uint32_t eb = atomicAdd(&a.n_extra[bh], parts - 1);
        if (pb + parts <= a.th->cap_partials && eb + parts - 1 <= a.th->cap_extra) {
            td.parts = parts; td.pbase = pb;
            for (uint32_t p = 1; p < parts; ++p) {
                TileItem ti;
                ti.bh = (uint32_t)bh; ti.blk = (uint32_t)blk; ti.part = p; ti.pidx = pb + p;
                a.titems[(size_t)bh * a.th->cap_extra + eb + p - 1] = ti;
            }
        } else if (eb < a.th->cap_extra) {
            // reserved queue entries that cannot be used must read as "nothing to do"
            for (uint32_t p = 1; p < parts && eb + p - 1 < a.th->cap_extra; ++p) {
                TileItem ti;
                ti.bh = (uint32_t)bh; ti.blk = (uint32_t)blk; ti.part = kVoidPart; ti.pidx = 0;
                a.titems[(size_t)bh * a.th->cap_extra + eb + p - 1] = ti;
            }
        }
    }
    a.tdesc[bh * a.blocks_bound + blk] = td;
}

// A block whose five cell rows lie inside ONE sort tile of whole rows is planned by that tile's workgroup,
// straight from its prefix sums in LDS (plan_tile_blocks); the others -- the seams between the bands of a
// level cut into several tiles, and every block of a level whose tiles are not whole rows -- by the slice's
// last workgroup from the cell table (plan_slice_blocks).
__device__ __host__ inline bool block_is_tile_local(const LevelRow &lr, int by)
{
    const int last = kTB * by + kTB < lr.Hl ? kTB * by + kTB : lr.Hl;      // the block's last cell row
    return lr.band > 0 && (kTB * by) / lr.band == last / lr.band;
}

// The blocks of tile `tl` (whole cell rows [ya, yb) of level row `lr`) that are local to it.  off[]: the tile's
// exclusive prefix sums (off[ncell] = total), base: the tile's first record.  The cells of a row that touch a
// block are one contiguous run, so a run's length is a difference of two prefix sums.
// In two halves: plan_tile_begin stores the descriptors and ASKS for the queue places of the blocks that need
// several work items (two returning device atomics, microseconds under the sort's store traffic);
// plan_tile_finish, called after the workgroup has moved its records, takes the answers.
struct PendingBlock { int blk; uint32_t parts, pb, eb; };          // blk < 0: nothing pending
__device__ __host__ inline int tile_local_blocks(const CTile &tl, const LevelRow &lr, int *by_lo)
{
    *by_lo = (tl.ya + kTB - 1) / kTB;
    int by_hi = *by_lo;                                             // (exclusive)
    while (by_hi < lr.nby4 && (kTB * by_hi + kTB < lr.Hl ? kTB * by_hi + kTB : lr.Hl) < tl.yb) ++by_hi;
    return (by_hi - *by_lo) * lr.nbx4;
}
// TS (Dims::taps_sorted, a compile-time property of the sort kernel that plans: the sort's two-vector variants have no
// register to spare for a run-time flag): also the owned prefix of each run, for msda_bwd_taps_sorted.hip
template <bool TS>
__device__ __forceinline__ int64_t describe_local_block(TileDesc &td, const CTile &tl, const LevelRow &lr, int by, int bx,
                                                        const uint32_t *off, int64_t base, TapsDesc &xd)
{
    const int tw = tl.Wl + 1;
    int64_t n = 0;
#pragma unroll
    for (int r = 0; r <= kTB; ++r) {
        const int cy = kTB * by + r;
        td.first[r] = 0; td.cnt[r] = 0;
        if (TS) xd.ocnt[r] = 0;
        if (cy <= lr.Hl) {
            const int p0 = (cy - tl.ya) * tw + kTB * bx, p1 = (cy - tl.ya) * tw + min(kTB * bx + kTB + 1, tw);
            td.first[r] = (int)(uint32_t)(base + off[p0]);
            td.cnt[r] = (int)(off[p1] - off[p0]);
            n += td.cnt[r];
            if (TS && (r < kTB || by == lr.nby4 - 1)) {
                const int q1 = bx == lr.nbx4 - 1 ? p1 : (cy - tl.ya) * tw + kTB * bx + kTB;
                xd.ocnt[r] = (int)(off[q1] - off[p0]);
            }
        }
    }
    if (TS) { xd.level = tl.level; xd.pad[0] = xd.pad[1] = 0; }
    td.hw = ((uint32_t)lr.Hl << 16) | (uint32_t)lr.Wl;
    td.lstart = lr.lstart;
    td.byx = ((uint32_t)by << 16) | (uint32_t)bx;
    td.parts = 1; td.pbase = 0; td.arrived = 0;
    return n;
}
template <bool TS>
__device__ inline PendingBlock plan_tile_begin(const TileReduceArgs &a, const Dims &d, int64_t bh, const CTile &tl, const LevelRow &lr,
                                               const uint32_t *off, int64_t base, int tid, int nthreads)
{
    PendingBlock pd;
    pd.blk = -1; pd.parts = 0; pd.pb = 0; pd.eb = 0;
    int by_lo;
    const int nloc = tile_local_blocks(tl, lr, &by_lo);
    // (a tile of <= kMaxTileCells cells has fewer local blocks than the workgroup has threads; were it otherwise,
    // the rest is planned in one piece)
    for (int i = tid + nthreads; i < nloc; i += nthreads) {
        const int by = by_lo + i / lr.nbx4, bx = i % lr.nbx4;
        TileDesc td;
        TapsDesc xd;
        const int64_t n = describe_local_block<TS>(td, tl, lr, by, bx, off, base, xd);
        if (TS) a.xdesc[bh * a.blocks_bound + lr.bbase4 + by * lr.nbx4 + bx] = xd;
        queue_block(a, d, bh, lr.bbase4 + by * lr.nbx4 + bx, td, n);
    }
    if (tid >= nloc) return pd;
    const int by = by_lo + tid / lr.nbx4, bx = tid % lr.nbx4;
    const int blk = lr.bbase4 + by * lr.nbx4 + bx;
    TileDesc td;
    TapsDesc xd;
    const int64_t n = describe_local_block<TS>(td, tl, lr, by, bx, off, base, xd);
    if (TS) a.xdesc[bh * a.blocks_bound + blk] = xd;
    a.tdesc[bh * a.blocks_bound + blk] = td;
    const uint32_t parts = (uint32_t)((n + tile_chunk(d) - 1) / tile_chunk(d));
    if (parts > 1) {
        pd.blk = blk; pd.parts = parts;
        pd.pb = atomicAdd(&a.th->n_partials, parts);
        pd.eb = atomicAdd(&a.n_extra[bh], parts - 1);
    }
    return pd;
}
__device__ inline void plan_tile_finish(const TileReduceArgs &a, const Dims &d, int64_t bh, const PendingBlock &pd)
{
    if (pd.blk < 0) return;
    const uint32_t parts = pd.parts, pb = pd.pb, eb = pd.eb;
    if (pb + parts <= a.th->cap_partials && eb + parts - 1 <= a.th->cap_extra) {
        TileDesc *td = &a.tdesc[bh * a.blocks_bound + pd.blk];
        td->parts = parts; td->pbase = pb;
        for (uint32_t p = 1; p < parts; ++p) {
            TileItem ti;
            ti.bh = (uint32_t)bh; ti.blk = (uint32_t)pd.blk; ti.part = p; ti.pidx = pb + p;
            a.titems[(size_t)bh * a.th->cap_extra + eb + p - 1] = ti;
        }
    } else if (eb < a.th->cap_extra) {
        // reserved queue entries that cannot be used must read as "nothing to do"
        for (uint32_t p = 1; p < parts && eb + p - 1 < a.th->cap_extra; ++p) {
            TileItem ti;
            ti.bh = (uint32_t)bh; ti.blk = (uint32_t)pd.blk; ti.part = kVoidPart; ti.pidx = 0;
            a.titems[(size_t)bh * a.th->cap_extra + eb + p - 1] = ti;
        }
    }
}

// The blocks of one (b, h) slice that no tile could plan by itself.  Run by the LAST sort workgroup of the
// slice (the cell table it reads was written by the slice's sort workgroups, which share an XCD and so an
// L2): thread `tid` of `nthreads` takes blocks tid, tid + nthreads, ...
template <bool TS>
__device__ inline void plan_slice_blocks(const TileReduceArgs &a, const Dims &d, int64_t bh, int tid, int nthreads)
{
    const LevelRow *lv = level_rows(a.hdr);
    const int nblk = a.hdr->n_blocks4;
    for (int blk = tid; blk < nblk; blk += nthreads) {
        int level = 0;
        while (level + 1 < d.L && blk >= lv[level + 1].bbase4) ++level;
        while (level < d.L && lv[level].nbx4 * lv[level].nby4 == 0) ++level;        // (empty levels own no block)
        TileDesc td;
        TapsDesc xd;
#pragma unroll
        for (int r = 0; r < 5; ++r) { td.first[r] = 0; td.cnt[r] = 0; xd.ocnt[r] = 0; }
        td.hw = 0; td.lstart = 0; td.byx = 0; td.parts = 1; td.pbase = 0; td.arrived = 0;
        xd.level = level < d.L ? level : 0; xd.pad[0] = xd.pad[1] = 0;
        int64_t n = 0;
        if (level < d.L) {
            const LevelRow lr = lv[level];
            const int rel = blk - lr.bbase4, by = rel / lr.nbx4, bx = rel - by * lr.nbx4;
            if (block_is_tile_local(lr, by)) continue;
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
#pragma unroll
            for (int dy = 0; dy <= kTB; ++dy) {
                td.first[dy] = (int)ent[dy][0].x;               // (the cells of a row are one contiguous run)
                int c = 0, oc = 0;
#pragma unroll
                for (int dx = 0; dx <= kTB; ++dx) {
                    c += (int)ent[dy][dx].y;
                    if (TS && (dx < kTB || bx == lr.nbx4 - 1)) oc += (int)ent[dy][dx].y;
                }
                td.cnt[dy] = c;
                if (dy < kTB || by == lr.nby4 - 1) xd.ocnt[dy] = oc;
                n += c;
            }
            td.hw = ((uint32_t)lr.Hl << 16) | (uint32_t)lr.Wl;
            td.lstart = lr.lstart;
            td.byx = ((uint32_t)by << 16) | (uint32_t)bx;
        }
        if (TS) a.xdesc[bh * a.blocks_bound + blk] = xd;
        queue_block(a, d, bh, blk, td, n);
    }
}
#endif

// 16-bit storage, D in {32, 64, 128}; MMFS_VALUE_ALGO=block keeps the vector-ALU reduce
bool tile_reduce_supported(int dtype, const Dims &d);
// the planned workspace as msda_bwd_taps_sorted.hip reads it (descriptors, items, records), *cap_extra: queue places per slice
TileReduceArgs taps_sorted_args(void *workspace, int dtype, const Dims &d, uint32_t *cap_extra);
hipError_t tile_reduce(int dtype, const void *grad_out, void *grad_value, const TileReduceArgs &a, const Dims &d,
                       uint32_t cap_extra, hipStream_t st);      // cap_extra: TileHeader::cap_extra (the host's copy)

}  // namespace blk
}  // namespace mmfs
