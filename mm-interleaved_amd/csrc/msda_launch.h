// msda_launch.h -- host-side entry points of the kernel translation units
// (consumed by msda_capi.hip, which owns the extern "C" ABI of include/mmfs_msda.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "msda_device.h"

namespace mmfs {

// dtype codes are enum mmfs_dtype of include/mmfs_msda.h
hipError_t forward(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                   const void *loc, const void *attn, void *out, const Dims &d, hipStream_t st,
                   const LevelSel *sel = nullptr, const float *init = nullptr);

// Forward with the coarse levels of the (b, h) slice held in LDS   [msda_fwd_cached.hip]
bool forward_cached_applicable(int dtype, const Dims &d, int *q_per_block);
hipError_t forward_cached(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                          const void *loc, const void *attn, void *out, const Dims &d, int q_per_block,
                          hipStream_t st);

// Location / attention-weight gradients (always) and, when scatter is true, grad_value
// accumulated with global float atomics into the fp32 (fp64 for dtype 3) buffer gv_acc,
// which the caller must have zero-filled.            [msda_bwd.hip]
hipError_t backward_taps(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                         const void *loc, const void *attn, const void *grad_out,
                         void *gv_acc, void *grad_loc, void *grad_attn, const Dims &d, bool scatter,
                         hipStream_t st, const LevelSel *sel = nullptr);
bool bwd_has_vector_path(int dtype, const Dims &d);

// grad_value by pixel-stationary tiles: no atomics, no fp32 buffer, every element of
// grad_value (storage dtype) written exactly once.     [msda_bwd_value.hip]
bool bwd_value_tiled_supported(int dtype, const Dims &d);
int64_t bwd_value_tiled_workspace_bytes(int dtype, const Dims &d);   // re-packed loc/attn copies
hipError_t backward_value_prepare(int dtype, const void *loc, const void *attn, void *workspace,
                                  const Dims &d, hipStream_t st);
hipError_t backward_value_sort(int dtype, const int64_t *shapes, const int64_t *start, void *workspace,
                               const Dims &d, hipStream_t st, uint64_t skip_levels = 0);
hipError_t backward_value_reduce(int dtype, const void *grad_out, void *grad_value, void *workspace,
                                 const Dims &d, hipStream_t st, uint64_t skip_levels = 0);

// Second generation: samples sorted by the cell of their top-left corner, 2x2 pixel blocks as
// owners (2.25 instead of 4 grad_out row reads per sample).   [msda_bwd_block.hip]
// backward_value_sort / _reduce route here when it applies (and no level is skipped).
bool bwd_value_block_supported(int dtype, const Dims &d);
int64_t bwd_value_block_workspace_bytes(int dtype, const Dims &d);
hipError_t backward_value_block_sort(int dtype, const int64_t *shapes, const int64_t *start, void *workspace,
                                     const Dims &d, hipStream_t st);
hipError_t backward_value_block_reduce(int dtype, const void *grad_out, void *grad_value, void *workspace,
                                       const Dims &d, hipStream_t st);
hipError_t backward_value_run(int dtype, const int64_t *shapes, const int64_t *start,
                              const void *grad_out, void *grad_value, void *workspace, const Dims &d,
                              hipStream_t st);
hipError_t backward_value_tiled(int dtype, const int64_t *shapes, const int64_t *start,
                                const void *loc, const void *attn, const void *grad_out,
                                void *grad_value, void *workspace, const Dims &d, hipStream_t st);

// Hybrid path: levels of <= 256 pixels as dense MFMA products, the rest through the gather
// kernels above restricted to plan.fine.                   [msda_dense.hip]
struct HybridPlan {
    bool active;              // some part has a dense level
    // grad_loc / grad_attn: dense dot products, levels chunked by pixel rows
    bool dots_active;
    DotPlan dots;
    LevelSel fine_taps;       // levels left to msda_bwd_vec
    // forward / grad_value (experimental, whole levels of <= 256 pixels)
    bool coarse_active;
    LevelSel fine;            // levels left to the gather / sort+reduce kernels
    CoarsePlan coarse;        // dense levels
    uint64_t coarse_mask;     // bit l set: level l is dense
};
HybridPlan make_hybrid_plan(int dtype, const Dims &d, const int64_t *host_shapes, const int64_t *host_start);
int64_t hybrid_fwd_workspace_bytes(int dtype, const Dims &d, const HybridPlan &p);   // packed value + fp32 partial output
const float *hybrid_fwd_init(void *workspace, const Dims &d, const HybridPlan &p);
hipError_t forward_coarse(int dtype, const void *value, const void *loc, const void *attn, void *workspace,
                          const Dims &d, const HybridPlan &p, hipStream_t st);
hipError_t backward_taps_coarse(int dtype, const void *value, const void *loc, const void *attn,
                                const void *grad_out, void *grad_loc, void *grad_attn, const Dims &d,
                                const HybridPlan &p, hipStream_t st);
int64_t hybrid_bwd_partial_bytes(const Dims &d, const HybridPlan &p);
hipError_t backward_value_coarse(int dtype, const void *loc, const void *attn, const void *grad_out,
                                 void *grad_value, void *partial, const Dims &d, const HybridPlan &p,
                                 hipStream_t st);

hipError_t cast_from_f32(int dtype, const float *src, void *dst, int64_t n, hipStream_t st);

}  // namespace mmfs
