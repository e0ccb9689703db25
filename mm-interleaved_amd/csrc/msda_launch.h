// msda_launch.h -- host-side entry points of the kernel translation units
// (consumed by msda_capi.hip, which owns the extern "C" ABI of include/mmfs_msda.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "msda_device.h"

namespace mmfs {

// dtype codes are enum mmfs_dtype of include/mmfs_msda.h
// algo: 0 the library chooses, 1 row gather (msda_fwd.hip), 2 LDS-resident levels (msda_fwd_mma.hip; the
// caller has checked fwd_mma_supported), 3 slices of 32 channels (msda_fwd_q8.hip; fwd_q8_supported), 4 a wave per
// query (msda_fwd_wq.hip; fwd_wq_supported)
hipError_t forward(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                   const void *loc, const void *attn, void *out, const Dims &d, hipStream_t st, int algo = 0);

// Second formulation of the forward for 16-bit storage, D in {64, 128}: small levels resident in LDS and sampled by
// the matrix cores, large levels by row gather.                                                   [msda_fwd_mma.hip]
bool fwd_mma_supported(int dtype, const Dims &d);        // the shape allows it
bool fwd_mma_applies(int dtype, const Dims &d);          // ... and it is expected to pay (the default routing)
hipError_t forward_mma(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                       const void *loc, const void *attn, void *out, const Dims &d, hipStream_t st);

// Third formulation of the forward for 16-bit storage, head widths that are multiples of 32 channels: slices of 32
// channels, every level whose slice fits in LDS sampled by the matrix cores in tiles of 8 queries.  [msda_fwd_q8.hip]
bool fwd_q8_supported(int dtype, const Dims &d);
bool fwd_q8_applies(int dtype, const Dims &d);
hipError_t forward_q8(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                      const void *loc, const void *attn, void *out, const Dims &d, hipStream_t st);

// Fourth formulation of the forward for 16-bit storage, D = 128: a wave per query, the bilinear weights on the diagonal
// of the matrix cores' A operand, the pixel rows as loaded (memory or LDS) as the B operand.      [msda_fwd_wq.hip]
bool fwd_wq_supported(int dtype, const Dims &d);
bool fwd_wq_applies(int dtype, const Dims &d);
hipError_t forward_wq(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                      const void *loc, const void *attn, void *out, const Dims &d, hipStream_t st);

// Location / attention-weight gradients (always) and, when scatter is true, grad_value
// accumulated with global float atomics into the fp32 (fp64 for dtype 3) buffer gv_acc,
// which the caller must have zero-filled.            [msda_bwd.hip]
hipError_t backward_taps(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                         const void *loc, const void *attn, const void *grad_out,
                         void *gv_acc, void *grad_loc, void *grad_attn, const Dims &d, bool scatter,
                         hipStream_t st, const LevelSel *sel = nullptr);
bool bwd_has_vector_path(int dtype, const Dims &d);
// One kernel for all levels, 16-bit storage, D = 128: the levels that fit in LDS contracted on the matrix cores, the
// others by row gather (replaces msda_bwd_vec + msda_taps_coarse where it applies).             [msda_taps_mma.hip]
bool taps_mma_supported(int dtype, const Dims &d);
bool taps_mma_applies(int dtype, const Dims &d);        // supported, expected to pay, and d.taps_algo does not say otherwise
// job (msda_plan.h; may be null): the grad_value half's opening launch -- clear the sort's cursors, plan -- done by this
// kernel's first workgroup on the side; value_prepare_job says whether that is all the opening launch would do.
namespace blk { struct PrepareJob; }
hipError_t backward_taps_mma(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                             const void *loc, const void *attn, const void *grad_out, void *grad_loc, void *grad_attn,
                             const Dims &d, hipStream_t st, const blk::PrepareJob *job = nullptr);
// Fills *job and returns true when backward_value_prepare for these arguments would neither re-pack loc / attn nor do
// anything but clear cursors and plan (block generation, the sort reads loc / attn where they are); else false.
bool value_prepare_job(int dtype, const void *loc, const void *attn, const int64_t *shapes, const int64_t *start,
                       void *workspace, const Dims &d, blk::PrepareJob *job);

// grad_value by pixel-stationary tiles: no atomics, no fp32 buffer, every element of
// grad_value (storage dtype) written exactly once.     [msda_bwd_value.hip]
bool bwd_value_tiled_supported(int dtype, const Dims &d);
int64_t bwd_value_tiled_workspace_bytes(int dtype, const Dims &d);   // re-packed loc/attn copies
// shapes / start given to _prepare: the block generation plans in the same launch and *planned says so
// (pass it on to _sort); all_rows_owned: the HOST vouches that every grad_value row has an owner level
// (canonical table), so no zero-fill pass is launched.
hipError_t backward_value_prepare(int dtype, const void *loc, const void *attn, void *workspace,
                                  const Dims &d, hipStream_t st,
                                  const int64_t *shapes = nullptr, const int64_t *start = nullptr, bool *planned = nullptr);
hipError_t backward_value_sort(int dtype, const int64_t *shapes, const int64_t *start, void *workspace,
                               const Dims &d, hipStream_t st, bool planned = false);
hipError_t backward_value_reduce(int dtype, const void *grad_out, void *grad_value, void *workspace,
                                 const Dims &d, hipStream_t st, bool all_rows_owned = false);

// Second generation: samples sorted by the cell of their top-left corner, 2x2 pixel blocks as
// owners (2.25 instead of 4 grad_out row reads per sample).   [msda_bwd_block.hip]
// backward_value_sort / _reduce route here when it applies (and no level is skipped).
bool bwd_value_block_supported(int dtype, const Dims &d);
int64_t bwd_value_block_workspace_bytes(int dtype, const Dims &d);
hipError_t backward_value_block_prepare(int dtype, const void *loc, const void *attn, const int64_t *shapes,
                                        const int64_t *start, void *workspace, const Dims &d, hipStream_t st);
hipError_t backward_value_block_sort(int dtype, const int64_t *shapes, const int64_t *start, void *workspace,
                                     const Dims &d, bool planned, hipStream_t st, void *g_loc = nullptr, void *g_attn = nullptr);
// grad_loc / grad_attn from the cell-sorted records: a wave per 4x4 block of cells, the block's 5x5 pixel rows of value as one
// matrix-core operand, the records' grad_out rows as the other (d.taps_sorted: the sort writes what this needs).
// Order of a backward on this route: prepare (plan), sort, taps_sorted, reduce.                 [msda_bwd_taps_sorted.hip]
int sort_tiles_exact(int dtype, const Dims &d, const int64_t *host_shapes);      // 0: unknown (no host table / not the block generation)
bool taps_sorted_supported(int dtype, const Dims &d);
hipError_t backward_taps_sorted(int dtype, const void *value, const void *grad_out, void *grad_loc, void *grad_attn,
                                void *workspace, const Dims &d, hipStream_t st);
hipError_t backward_value_block_reduce(int dtype, const void *grad_out, void *grad_value, void *workspace,
                                       const Dims &d, bool all_rows_owned, hipStream_t st);
hipError_t backward_value_run(int dtype, const int64_t *shapes, const int64_t *start,
                              const void *grad_out, void *grad_value, void *workspace, const Dims &d,
                              hipStream_t st, bool planned = false, bool all_rows_owned = false);
hipError_t backward_value_tiled(int dtype, const int64_t *shapes, const int64_t *start,
                                const void *loc, const void *attn, const void *grad_out,
                                void *grad_value, void *workspace, const Dims &d, hipStream_t st,
                                bool all_rows_owned = false);

// Hybrid routing of grad_loc / grad_attn: levels of <= 256 pixels as dense MFMA dot products, the rest
// through the gather kernel restricted to plan.fine_taps.                   [msda_dense.hip]
struct HybridPlan {
    bool active;              // some level is dense
    bool dots_active;
    DotPlan dots;             // dense levels, chunked by pixel rows
    LevelSel fine_taps;       // levels left to msda_bwd_vec
};
HybridPlan make_hybrid_plan(int dtype, const Dims &d, const int64_t *host_shapes, const int64_t *host_start);
hipError_t backward_taps_coarse(int dtype, const void *value, const void *loc, const void *attn,
                                const void *grad_out, void *grad_loc, void *grad_attn, const Dims &d,
                                const HybridPlan &p, hipStream_t st, const blk::PrepareJob *job = nullptr);   // job: as backward_taps_mma

hipError_t cast_from_f32(int dtype, const float *src, void *dst, int64_t n, hipStream_t st);
// ``bytes`` zero bytes at ``p`` as a KERNEL in stream order.  Not hipMemsetAsync: recorded into a HIP graph that becomes
// a memset node, and on replay the counters a memset node was to clear were seen uncleared by the kernel node behind
// it (round 5, r05g36-37: msda_bwd_value_sort found 0xc4281000 in its level cursor and wrote its records 26 GB past
// the workspace -- one replay in a few dozen, only the first run on a fresh device).  A kernel orders like a kernel.
hipError_t zero_fill(void *p, size_t bytes, hipStream_t st);

// grad_value for a level table the device-side check of the sorted backward refused (its plan's verdict is read on the
// device from the workspace; nothing happens for a table it served): the reference's float-atomic scatter.
// acc: refused_table_scratch_bytes of fp32 image for 16-bit storage (ignored for fp32).          [msda_bwd_refused.hip]
int64_t refused_table_scratch_bytes(int dtype, const Dims &d);
hipError_t backward_value_refused_table(int dtype, const int64_t *shapes, const int64_t *start, const void *loc,
                                        const void *attn, const void *grad_out, void *grad_value, void *workspace,
                                        float *acc, const Dims &d, hipStream_t st);

}  // namespace mmfs
