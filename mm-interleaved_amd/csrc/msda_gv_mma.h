// msda_gv_mma.h -- grad_value of the SMALL levels, query-chunk stationary (msda_gv_mma.hip): what the host plan and
// the kernel share.
//
// The sorted backward (msda_bwd_block.hip + msda_bwd_tile.hip) sorts every sample by cell through global memory
// and lets a wave own a 4x4 pixel block: per 16 records it moves 4 KB of grad_out rows global -> LDS, reads them
// back transposed and multiplies.  For the small levels of the pyramid -- every level receives the same number of
// samples whatever its size -- the sort can stay inside a workgroup instead:
//
//   * a 1024-lane workgroup owns a GROUP of levels of one (batch, head) slab, all of their 4x4 pixel blocks, and a
//     range of queries; it walks the range in chunks of QC queries whose grad_out rows (QC x D x 2 bytes) it
//     copies into LDS ONCE -- every sample of the chunk reads its row from there, 1.56 times on average;
//   * the chunk's samples of the group's levels are binned by block in LDS (count, prefix, place: 4-byte records
//     {sample, query}), each list ordered so that eight consecutive records belong to queries of eight different
//     residues modulo 8 while such queries last -- the rows a transposing LDS read touches together then sit in
//     different banks; a dense block's samples are dealt over 2^k "virtual blocks" so that the 16 waves carry
//     equal loads, and every virtual block is owned by one wave which keeps its 16 x D fp32 tile in registers
//     across all chunks: no partial sums inside the workgroup until the end;
//   * the product is the tile reduce's: grad_value[pixel, :] += W[pixel, record] * grad_out[query(record), :] on
//     v_mfma_f32_16x16x32, rows fetched with the transposing LDS read, weights as hi + lo 16-bit parts.
//
// Which levels, how they are grouped and how the queries are cut is decided on the HOST from a host copy of the
// level table (the hybrid entry point has one) and travels as a kernel argument (Table, < 2 KB); the sort's plan
// kernel is told to leave those levels alone (Dims::gv_skip).
#pragma once
#include "msda_device.h"
#include <stdint.h>

namespace mmfs {
namespace gv {

constexpr int kWaves = 16;
constexpr int kThreads = kWaves * 64;
constexpr int kMaxLevels = 16;            // levels one launch can own
constexpr int kMaxGroups = 32;
constexpr int kMaxSegs = 4;               // levels per group
constexpr int kMaxVb = 64;                // virtual blocks per group (16 waves x slots)
constexpr int kTB = 4;                    // pixels per block side (= msda_bwd_tile.hip)
constexpr int kLds = 160 * 1024;
constexpr int kCtrl = 6144;               // counters (two sets of 64 x 8), list bases, segment and virtual-block tables
constexpr int kATile = 2048;              // per wave: hi tile + lo tile, 16 pixels x 32 records x 2 bytes each
constexpr int kRows0 = kCtrl + kWaves * kATile;
constexpr int kMaxSamples = 2048;         // samples of a chunk (2 per lane kept between the two binning passes)
constexpr int kMaxQc = 256;               // queries of a chunk (8 bits in a record)

struct Level { int32_t level, Hl, Wl, lstart, nbx, nby; };
struct Seg { uint16_t lslot, log2s, v0, rb0; };           // level slot; 2^log2s virtual blocks per block; first virtual block; first block of the group
struct Group {
    Seg seg[kMaxSegs];
    uint16_t nseg, nvb, qc, nrb;          // segments, virtual blocks, queries per chunk, blocks
    uint16_t qparts, wg0;                 // workgroups (query ranges) of the group; first workgroup of the group inside a slab
    uint32_t pbase;                       // first partial tile of the group inside a slab (qparts > 1)
};
struct Table {
    int32_t n_levels, n_groups, wgs_per_slab, ptiles_per_slab;
    uint64_t skip[2];                     // bit l: level l is served here (the sorted backward leaves it alone)
    Level lv[kMaxLevels];
    Group g[kMaxGroups];
};
static_assert(sizeof(Table) <= 2048, "travels as a kernel argument");

template <int D> struct Geom {
    static constexpr int RB = D * 2;                  // bytes of a grad_out row of one head
    static constexpr int LPR = RB / 16;               // 16-byte chunks per row
    static constexpr int NT = D / 16;                 // 16-channel column tiles
    static constexpr int SLOTS = D >= 128 ? 2 : 4;    // virtual blocks per wave: 64 accumulator registers either way
    static constexpr int VB = kWaves * SLOTS;
};

// rows the chunk's DMA rounds cover (whole rounds of 1024 lanes x 16 bytes); the row of zeros sits behind them
__host__ __device__ inline int rows_alloc(int qc, int lpr) { return ((qc * lpr + kThreads - 1) / kThreads * kThreads) / lpr; }
// dynamic LDS of a workgroup whose chunk holds qc queries and ns samples
__host__ __device__ inline int lds_bytes(int qc, int ns, int rb)
{
    const int rows = (rows_alloc(qc, rb / 16) + 1) * rb;
    return kRows0 + rows + ((ns * 8 + 15) & ~15) + 4 * ns * 4;
}

// host: the plan for these dimensions and this level table (n_groups == 0: the kernel takes nothing)
Table make_plan(int dtype, const Dims &d, const int64_t *host_shapes, const int64_t *host_start);
int64_t workspace_bytes(const Table &t, const Dims &d);           // fp32 partial tiles + arrival counters
hipError_t backward_value(int dtype, const void *grad_out, const void *loc, const void *attn, void *grad_value,
                          void *workspace, const Dims &d, const Table &t, hipStream_t st);

}  // namespace gv
}  // namespace mmfs
