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
                   const void *loc, const void *attn, void *out, const Dims &d, hipStream_t st);

hipError_t backward(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                    const void *loc, const void *attn, const void *grad_out,
                    void *grad_value_acc, void *grad_loc, void *grad_attn, const Dims &d, hipStream_t st);

hipError_t cast_from_f32(int dtype, const float *src, void *dst, int64_t n, hipStream_t st);

}  // namespace mmfs
