// msda_dots.h -- channel dot products and lane-group sums of the grad_loc / grad_attn kernels
// (msda_bwd.hip: row gather; msda_taps_mma.hip: row gather for the large levels next to the matrix cores).
#pragma once
#include "msda_device.h"
#include <type_traits>

namespace mmfs {

// Sum over the LPI lanes of a query's lane group; every lane gets the total.
// Within a 16-lane DPP row this is pure VALU (v_add_f32 with a DPP operand: quad swaps,
// half-row and row mirrors) -- no LDS crossbar traffic; wider groups finish with wave shuffles.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true);
    return v + __int_as_float(moved);
}

template <int LPI>
__device__ __forceinline__ float group_sum(float v)
{
    if (LPI >= 2) v = dpp_add<0xB1>(v);      // quad_perm [1,0,3,2]  : lane ^ 1
    if (LPI >= 4) v = dpp_add<0x4E>(v);      // quad_perm [2,3,0,1]  : lane ^ 2
    if (LPI >= 8) v = dpp_add<0x141>(v);     // row_half_mirror      : 7 - lane within 8
    if (LPI >= 16) v = dpp_add<0x140>(v);    // row_mirror           : 15 - lane within 16
#pragma unroll
    for (int off = 16; off < LPI; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Four values summed over a 16-lane group TOGETHER; the totals are valid in the group's lane 0 only
// (the one lane that finishes the sample).  Instead of four butterflies of four steps each, the
// value count halves with every step: the row mirror leaves each half-row with two of the four
// sums, the half-row mirror each quad with one, two quad swaps finish it, and lane 0 collects the
// other quads' totals -- 5 adds, 6 selects, 3 moves instead of 16 adds (+ their moves): the taps
// kernel is bound by its vector instruction count.
__device__ __forceinline__ void group_sum4_row(float (&d)[4], int lig)
{
    const bool hi = (lig & 8) != 0, odd = (lig & 4) != 0;
    auto mv = [](float v, auto ctrl) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    // lanes 0-7 keep (d0, d1), lanes 8-15 keep (d2, d3); partner = 15 - lane
    const float k0 = hi ? d[2] : d[0], k1 = hi ? d[3] : d[1];
    const float s0 = hi ? d[0] : d[2], s1 = hi ? d[1] : d[3];
    const float r0 = k0 + mv(s0, std::integral_constant<int, 0x140>());
    const float r1 = k1 + mv(s1, std::integral_constant<int, 0x140>());
    // lanes with bit 2 clear keep the first, the others the second; partner = 7 - lane within the half
    const float k = odd ? r1 : r0, s = odd ? r0 : r1;
    float t = k + mv(s, std::integral_constant<int, 0x141>());
    t = dpp_add<0xB1>(t);                            // lane ^ 1
    t = dpp_add<0x4E>(t);                            // lane ^ 2: quad q now holds the total of d[q]
    d[0] = t;
    d[1] = mv(t, std::integral_constant<int, 0x104>());   // row_shl:4  -> lane 0 reads lane 4
    d[2] = mv(t, std::integral_constant<int, 0x108>());   // row_shl:8  -> lane 8
    d[3] = mv(t, std::integral_constant<int, 0x10C>());   // row_shl:12 -> lane 12
}

// The same for 8-lane groups (heads of 64 channels: two groups per DPP row): half-row mirror, a quad swap, a lane swap; the
// totals in the group's lane 0 (lanes 0 and 8 of the row).
__device__ __forceinline__ void group_sum4_half(float (&d)[4], int lig)
{
    const bool hi = (lig & 4) != 0, odd = (lig & 2) != 0;
    auto mv = [](float v, auto ctrl) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    // lanes 0-3 keep (d0, d1), lanes 4-7 keep (d2, d3); partner = 7 - lane within the group
    const float k0 = hi ? d[2] : d[0], k1 = hi ? d[3] : d[1];
    const float s0 = hi ? d[0] : d[2], s1 = hi ? d[1] : d[3];
    const float r0 = k0 + mv(s0, std::integral_constant<int, 0x141>());
    const float r1 = k1 + mv(s1, std::integral_constant<int, 0x141>());
    // lanes with bit 1 clear keep the first, the others the second; partner = lane ^ 2
    const float k = odd ? r1 : r0, s = odd ? r0 : r1;
    float t = k + mv(s, std::integral_constant<int, 0x4E>());
    t = dpp_add<0xB1>(t);                            // lane ^ 1: lane pair p of the group now holds the total of d[p]
    d[0] = t;
    d[1] = mv(t, std::integral_constant<int, 0x102>());   // row_shl:2 -> lane 0 reads lane 2
    d[2] = mv(t, std::integral_constant<int, 0x104>());   // row_shl:4 -> lane 4
    d[3] = mv(t, std::integral_constant<int, 0x106>());   // row_shl:6 -> lane 6
}

// Dot product of two 16-byte channel vectors, fp32 result.  The taps kernel is VALU-bound (94 % busy
// at the north-star shape, rocprofv3 SQ_ACTIVE_INST_VALU): unpacking 16-bit channels costs one
// instruction per element and the multiply-adds another 0.5-1.  For 16-bit storage the packed dot
// product instructions (v_dot2c_f32_bf16 / v_dot2c_f32_f16: two exact products + fp32 accumulate)
// take the vectors as they are -- 4 instructions per row instead of ~13.
template <typename T> struct RowDot {
    static __device__ __forceinline__ float run(const uint4 &a, const uint4 &b) {
        float x[Vec16<T>::N], y[Vec16<T>::N];
        Vec16<T>::unpack(a, x); Vec16<T>::unpack(b, y);
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < Vec16<T>::N; ++i) acc = fmaf(x[i], y[i], acc);
        return acc;
    }
};
template <> struct RowDot<bf16_t> {
    typedef __bf16 v2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ float run(const uint4 &a, const uint4 &b) {
        float acc = 0.f;
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2, a.x), __builtin_bit_cast(v2, b.x), acc, false);
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2, a.y), __builtin_bit_cast(v2, b.y), acc, false);
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2, a.z), __builtin_bit_cast(v2, b.z), acc, false);
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2, a.w), __builtin_bit_cast(v2, b.w), acc, false);
        return acc;
    }
};
template <> struct RowDot<half_t> {
    typedef _Float16 v2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ float run(const uint4 &a, const uint4 &b) {
        float acc = 0.f;
        acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(v2, a.x), __builtin_bit_cast(v2, b.x), acc, false);
        acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(v2, a.y), __builtin_bit_cast(v2, b.y), acc, false);
        acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(v2, a.z), __builtin_bit_cast(v2, b.z), acc, false);
        acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(v2, a.w), __builtin_bit_cast(v2, b.w), acc, false);
        return acc;
    }
};

}  // namespace mmfs
