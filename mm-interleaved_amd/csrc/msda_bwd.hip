// msda_bwd.hip -- backward kernels of multi-scale deformable attention for gfx950.
//
// Replaces the reference backward family
//   mm_interleaved/models/utils/ops/src/cuda/ms_deform_im2col_cuda.cuh:304-923
//   (block = D threads per (b,q,h); per sample two __syncthreads, a shared-memory
//   reduction by thread 0 or a tree, and 4 scalar atomics per thread)
// with the forward's organisation (msda_fwd.hip): tap records staged in LDS per
// query tile, LPI lanes x 16-byte channel vectors per query, head -> XCD affinity.
//
// Per sample the reference needs, for the location / weight gradients, sums over the
// D channels of  g*v1, g*v2, g*v3, g*v4  (g = grad_out row, v1..v4 = the four corner
// rows): everything else is per-sample scalar algebra (cuh:119-161):
//     grad_attn  = w1*d1 + w2*d2 + w3*d3 + w4*d4
//     grad_loc.x = Wl * attn * ( (1-fy)*(d2-d1) + fy*(d4-d3) )
//     grad_loc.y = Hl * attn * ( (1-fx)*(d3-d1) + fx*(d4-d2) )
// so each lane forms 4 partial dot products over its own channels and the LPI lanes
// of the query combine them with wavefront shuffles -- no barrier, no shared-memory
// reduction inside the sample loop.  Every (query, sample) has exactly one writer for
// grad_loc / grad_attn; the results go back through the LDS record so the global
// stores are as coalesced as the loads were.
//
// grad_value is accumulated in fp32 with hardware global_atomic_add_f32, matching the
// reference's "accumulate in fp32, cast at the end" (ms_deform_attn_cuda.cu:122-165).
// For the atomics the lanes of a query switch to an interleaved channel map
// (channel = j*LPI + lane) so one atomic instruction covers LPI consecutive floats.
#include <algorithm>
#include "msda_device.h"
#include "msda_dots.h"
#include "msda_launch.h"
#include <type_traits>

namespace mmfs {

constexpr int kThreads = 256;
constexpr int kRecsPerBlock = 512;
#ifndef MMFS_TAPS_UNROLL
#define MMFS_TAPS_UNROLL 2
#endif
constexpr int kUnroll = MMFS_TAPS_UNROLL;

template <typename A>
__device__ __forceinline__ void atomic_add(A *p, A v)
{
    // relaxed, agent scope: the slice may be touched from any XCD.  Measured on MI355X:
    // ~330 G float adds/s for the whole chip whatever the footprint or scope
    // (tools/ubench/atomics.hip), i.e. one lane-add per L2 channel per clock -- which
    // is why this path is only the fallback (see msda_bwd_value.hip).
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// SCATTER = true : also accumulates grad_value with global atomics (fallback path)
// SCATTER = false: location / weight gradients only; grad_value comes from the
//                  pixel-stationary kernel in msda_bwd_value.hip
// BUF: value rows through a buffer descriptor (see msda_device.h); only without SCATTER
template <typename T, int LPI, bool SCATTER, bool BUF>
__global__ void __launch_bounds__(kThreads)
msda_bwd_vec(const T *__restrict__ value, const int64_t *__restrict__ shapes,
             const int64_t *__restrict__ start, const T *__restrict__ loc,
             const T *__restrict__ attn, const T *__restrict__ grad_out,
             float *__restrict__ grad_value, T *__restrict__ grad_loc, T *__restrict__ grad_attn,
             const Dims d, const LevelSel sel)
{
    typedef Vec16<T> V;
    constexpr int VEC = V::N;
    constexpr int QPB = kThreads / LPI;
    constexpr int KC = (kRecsPerBlock / QPB) > kUnroll ? (kRecsPerBlock / QPB) : kUnroll;
    constexpr int STRIDE = 2 * KC + 1;
    constexpr bool PRE = BUF && !SCATTER;           // records carry byte offsets, not pixel rows
    __shared__ uint4 lds[QPB * STRIDE];
    __shared__ uint8_t sel_idx[kMaxSelLevels];
    __shared__ LevelLds levels;

    const BlockCoord bc = block_coord(d, QPB);
    const int tid = threadIdx.x;
    const int qi = tid / LPI, lig = tid % LPI;
    const int q = bc.q0 + qi;
    const bool q_ok = q < d.Nq;
    // hybrid routing: the levels not in sel get their grad_loc / grad_attn from msda_taps_coarse
    const bool all_levels = sel.n < 0;
    const int Ksel = all_levels ? d.K : sel.n * d.P;
    if (!all_levels && tid < kMaxSelLevels) sel_idx[tid] = sel.idx[tid];
    levels.load(shapes, start, d.L, tid, kThreads);
    const bool pair_ok = (((uintptr_t)loc | (uintptr_t)grad_loc) & (2 * sizeof(T) - 1)) == 0;
    __syncthreads();

    const int64_t HD = (int64_t)d.H * d.D;
    const int64_t slice = ((int64_t)bc.b * d.S) * HD + (int64_t)bc.h * d.D;
    const T *vbase = value + slice + lig * VEC;
    float *gvbase = grad_value + slice + lig;
    const uint32_t row_bytes = (uint32_t)(HD * sizeof(T));
    const uint32_t lane_off = (uint32_t)(lig * 16);
    __amdgpu_buffer_rsrc_t rsrc;
    if (BUF) rsrc = make_slab_rsrc(value + slice, ((int64_t)d.S * HD - (int64_t)bc.h * d.D) * (int64_t)sizeof(T));

    // upstream gradient of this query: contiguous map for the dots, interleaved for atomics
    float gat[VEC];
    uint4 graw = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int j = 0; j < VEC; ++j) gat[j] = 0.f;
    if (q_ok) {
        const T *gp = grad_out + (((int64_t)bc.b * d.Nq + q) * d.H + bc.h) * d.D;
        graw = *reinterpret_cast<const uint4 *>(gp + lig * VEC);
        if (SCATTER) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) gat[j] = to_f32(gp[j * LPI + lig]);
        }
    }

    for (int k0 = 0; k0 < Ksel; k0 += KC) {
        const int kc = min(KC, Ksel - k0);
        const int kc_pad = (kc + kUnroll - 1) / kUnroll * kUnroll;
        if (k0 > 0) __syncthreads();
        // ---- stage
        for (int r = tid; r < QPB * kc_pad; r += kThreads) {
            const int rq = r / kc_pad, kk = r - rq * kc_pad;
            int row[4] = {-1, -1, -1, -1};
            float fx = 0.f, fy = 0.f, a = 0.f;
            uint32_t lv = 0;                              // the sample's level: its extents are looked up when the sample is finished
            const int sq = bc.q0 + rq;
            if (kk < kc && sq < d.Nq) {
                const int ks = k0 + kk;
                const int l = all_levels ? ks / d.P : (int)sel_idx[ks / d.P];
                const int k = l * d.P + ks % d.P;
                const int64_t s = (((int64_t)bc.b * d.Nq + sq) * d.H + bc.h) * d.K + k;
                int Hl, Wl, lstart;
                levels.get(shapes, start, l, Hl, Wl, lstart);
                float lx, ly;
                load_xy(loc, s, pair_ok, lx, ly);
                const Tap<float> t = locate<float>(lx, ly, Hl, Wl, lstart);
                fx = t.fx; fy = t.fy; a = to_f32(attn[s]);
                // lazy_attn: nobody reads the gradients of a zero-weight sample -> no rows, zeros out
                if (!(d.lazy_attn && !SCATTER && a == 0.f)) {
                    row[0] = t.row[0]; row[1] = t.row[1]; row[2] = t.row[2]; row[3] = t.row[3];
                }
                lv = (uint32_t)l;
            }
            if (!BUF) {
                // flat addresses have no descriptor to stop a row that a malformed level table puts past S
#pragma unroll
                for (int c = 0; c < 4; ++c) row[c] = row[c] < d.S ? row[c] : -1;
            }
            if (PRE) {
#pragma unroll
                for (int c = 0; c < 4; ++c)      // pixel row -> byte offset in the slab, or "outside" (reads as zeros)
                    row[c] = row[c] >= 0 ? (int)((uint32_t)row[c] * row_bytes) : (int)kOobOffset;
            }
            uint4 *dst = &lds[rq * STRIDE + 2 * kk];
            dst[0] = make_uint4(row[0], row[1], row[2], row[3]);
            dst[1] = make_uint4(__float_as_uint(fx), __float_as_uint(fy), __float_as_uint(a), lv);
        }
        __syncthreads();
        // ---- gather + reduce + scatter
        if (q_ok) {
            uint4 *recs = &lds[qi * STRIDE];
            for (int kk = 0; kk < kc_pad; kk += kUnroll) {
                uint4 raw[kUnroll][4];
                int rows[kUnroll][4];
                uint4 meta[kUnroll];
                if (!SCATTER && d.lazy_attn) {
                    // whole wave on zero-weight samples (consecutive tokens are blind to the same
                    // images): skip the loads -- an "outside" row moves no data but still costs the
                    // address path its cycles -- and leave zeros for the store pass
                    bool live = false;
#pragma unroll
                    for (int u = 0; u < kUnroll; ++u) live |= (recs[2 * (kk + u) + 1].z << 1) != 0u;
                    if (__builtin_amdgcn_ballot_w64(live) == 0ull) {
                        if (lig == 0) {
#pragma unroll
                            for (int u = 0; u < kUnroll; ++u) recs[2 * (kk + u)] = make_uint4(0u, 0u, 0u, 0u);
                        }
                        continue;
                    }
                }
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    const uint4 rr = recs[2 * (kk + u)];
                    meta[u] = recs[2 * (kk + u) + 1];
                    rows[u][0] = (int)rr.x; rows[u][1] = (int)rr.y; rows[u][2] = (int)rr.z; rows[u][3] = (int)rr.w;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (PRE)      // offsets made when the record was staged; a corner outside the map reads as zeros
                            raw[u][c] = buffer_load16(rsrc, (uint32_t)rows[u][c] + lane_off);
                        else if (BUF)
                            raw[u][c] = buffer_load16(rsrc, rows[u][c] >= 0 ? (uint32_t)rows[u][c] * row_bytes + lane_off
                                                                           : kOobOffset);
                        else
                            raw[u][c] = *reinterpret_cast<const uint4 *>(vbase + (int64_t)max(rows[u][c], 0) * HD);
                    }
                }
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    float dot[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float acc = RowDot<T>::run(graw, raw[u][c]);
                        dot[c] = (BUF || rows[u][c] >= 0) ? acc : 0.f;
                    }
                    if (LPI == 16 && !SCATTER) group_sum4_row(dot, lig);      // totals in the group's lane 0
                    else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) dot[c] = group_sum<LPI>(dot[c]);
                    }
                    // the four dots go back through the record; the per-sample algebra is done in the store
                    // pass below, one lane per SAMPLE (here only one lane in LPI would do it, per instruction)
                    if (lig == 0)
                        recs[2 * (kk + u)] = make_uint4(__float_as_uint(dot[0]), __float_as_uint(dot[1]),
                                                        __float_as_uint(dot[2]), __float_as_uint(dot[3]));
                    if (SCATTER) {
                        const float fx = __uint_as_float(meta[u].x), fy = __uint_as_float(meta[u].y);
                        const float a = __uint_as_float(meta[u].z);
                        const float gy = 1.f - fy, gx = 1.f - fx;
                        const float w[4] = {gy * gx, gy * fx, fy * gx, fy * fx};
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            if (rows[u][c] >= 0) {
                                const float coef = w[c] * a;
                                float *p = gvbase + (int64_t)rows[u][c] * HD;
#pragma unroll
                                for (int j = 0; j < VEC; ++j) atomic_add(p + j * LPI, coef * gat[j]);
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
        // ---- coalesced store of this chunk's grad_attn / grad_loc
        for (int r = tid; r < QPB * kc; r += kThreads) {
            const int rq = r / kc, kk = r - rq * kc;
            const int sq = bc.q0 + rq;
            if (sq >= d.Nq) continue;
            const uint4 res = lds[rq * STRIDE + 2 * kk], meta = lds[rq * STRIDE + 2 * kk + 1];
            const int ks = k0 + kk;
            const int k = all_levels ? ks : (int)sel_idx[ks / d.P] * d.P + ks % d.P;
            const int64_t s = (((int64_t)bc.b * d.Nq + sq) * d.H + bc.h) * d.K + k;
            // per-sample algebra of the reference (cuh:119-161) on the four corner dots
            const float d0 = __uint_as_float(res.x), d1 = __uint_as_float(res.y), d2 = __uint_as_float(res.z), d3 = __uint_as_float(res.w);
            const float fx = __uint_as_float(meta.x), fy = __uint_as_float(meta.y), a = __uint_as_float(meta.z);
            int Hli, Wli, lst;                          // (exact for any extent: nothing is packed into 16 bits here)
            levels.get(shapes, start, (int)meta.w, Hli, Wli, lst);
            const float gy = 1.f - fy, gx = 1.f - fx;
            const float ga = (gy * gx) * d0 + (gy * fx) * d1 + (fy * gx) * d2 + (fy * fx) * d3;
            const float dw = gy * (d1 - d0) + fy * (d3 - d2);
            const float dh = gx * (d2 - d0) + fx * (d3 - d1);
            store_stream(grad_attn + s, (T)ga);              // (final results of the step: non-temporal, msda_device.h)
            store_xy(grad_loc, s, pair_ok, (float)Wli * dw * a, (float)Hli * dh * a);
        }
    }
}

// Scalar fallback (any D, fp64): one thread per (b, q, h), serial over samples and
// channels; single writer for grad_loc / grad_attn, atomics for grad_value.
template <typename T>
__global__ void __launch_bounds__(kThreads)
msda_bwd_scalar(const T *__restrict__ value, const int64_t *__restrict__ shapes,
                const int64_t *__restrict__ start, const T *__restrict__ loc,
                const T *__restrict__ attn, const T *__restrict__ grad_out,
                typename Acc<T>::type *__restrict__ grad_value, T *__restrict__ grad_loc,
                T *__restrict__ grad_attn, const Dims d, const int64_t items)
{
    typedef typename Acc<T>::type A;
    const int64_t HD = (int64_t)d.H * d.D;
    for (int64_t item = (int64_t)blockIdx.x * kThreads + threadIdx.x; item < items;
         item += (int64_t)gridDim.x * kThreads) {
        const int h = (int)(item % d.H);
        const int64_t b = item / d.H / d.Nq;
        const int64_t slice = b * d.S * HD + (int64_t)h * d.D;
        const T *g = grad_out + item * d.D;
        for (int k = 0; k < d.K; ++k) {
            const int l = k / d.P;
            const int64_t s = item * d.K + k;
            const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
            const Tap<A> t = locate<A>((A)loc[2 * s], (A)loc[2 * s + 1], Hl, Wl, (int)start[l]);
            const A gy = 1 - t.fy, gx = 1 - t.fx, a = (A)attn[s];
            const A w[4] = {gy * gx, gy * t.fx, t.fy * gx, t.fy * t.fx};
            A dot[4] = {0, 0, 0, 0};
            for (int c = 0; c < d.D; ++c) {
                const A gc = (A)g[c];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (t.row[i] >= 0) {
                        const int64_t off = slice + (int64_t)t.row[i] * HD + c;
                        dot[i] += gc * (A)value[off];
                        atomic_add(grad_value + off, w[i] * a * gc);
                    }
                }
            }
            grad_attn[s] = (T)(w[0] * dot[0] + w[1] * dot[1] + w[2] * dot[2] + w[3] * dot[3]);
            grad_loc[2 * s] = (T)((A)Wl * a * (gy * (dot[1] - dot[0]) + t.fy * (dot[3] - dot[2])));
            grad_loc[2 * s + 1] = (T)((A)Hl * a * (gx * (dot[2] - dot[0]) + t.fx * (dot[3] - dot[1])));
        }
    }
}

// dst = src, one float per thread and trip (the fp32 "cast": a copy -- as a kernel, not hipMemcpyAsync: recorded into a HIP
// graph that is a memcpy node, and nodes of that family do not order like kernels there, see zero_fill)
__global__ void __launch_bounds__(kThreads)
copy_f32_kernel(const float *__restrict__ src, float *__restrict__ dst, const int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

// dst = (T) src, 4 elements per thread
template <typename T>
__global__ void __launch_bounds__(kThreads)
cast_kernel(const float *__restrict__ src, T *__restrict__ dst, const int64_t n)
{
    const int64_t n4 = n / 4;
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4 *>(src)[i];
        T o[4] = {(T)v.x, (T)v.y, (T)v.z, (T)v.w};
        reinterpret_cast<uint2 *>(dst)[i] = *reinterpret_cast<const uint2 *>(o);
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride)
        dst[i] = (T)src[i];
}

// ---------------------------------------------------------------- launchers
template <typename T, int LPI>
static hipError_t launch_vec(const void *value, const int64_t *shapes, const int64_t *start,
                             const void *loc, const void *attn, const void *go, void *gv, void *gl,
                             void *ga, Dims d, bool scatter, hipStream_t st, const LevelSel &sel)
{
    constexpr int QPB = kThreads / LPI;
    d.q_tiles = (d.Nq + QPB - 1) / QPB;
    const int64_t blocks = (int64_t)d.B * d.q_tiles * d.H;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    const bool buf = (int64_t)d.S * d.H * d.D * (int64_t)sizeof(T) <= kMaxSlabBytes;
    if (scatter)
        hipLaunchKernelGGL((msda_bwd_vec<T, LPI, true, false>), dim3((unsigned)blocks), dim3(kThreads), 0, st,
                           (const T *)value, shapes, start, (const T *)loc, (const T *)attn, (const T *)go,
                           (float *)gv, (T *)gl, (T *)ga, d, sel);
    else if (buf)
        hipLaunchKernelGGL((msda_bwd_vec<T, LPI, false, true>), dim3((unsigned)blocks), dim3(kThreads), 0, st,
                           (const T *)value, shapes, start, (const T *)loc, (const T *)attn, (const T *)go,
                           (float *)nullptr, (T *)gl, (T *)ga, d, sel);
    else
        hipLaunchKernelGGL((msda_bwd_vec<T, LPI, false, false>), dim3((unsigned)blocks), dim3(kThreads), 0, st,
                           (const T *)value, shapes, start, (const T *)loc, (const T *)attn, (const T *)go,
                           (float *)nullptr, (T *)gl, (T *)ga, d, sel);
    return hipGetLastError();
}

template <typename T>
static hipError_t launch_scalar(const void *value, const int64_t *shapes, const int64_t *start,
                                const void *loc, const void *attn, const void *go, void *gv, void *gl,
                                void *ga, Dims d, hipStream_t st)
{
    const int64_t items = (int64_t)d.B * d.Nq * d.H;
    const int64_t blocks = std::min<int64_t>((items + kThreads - 1) / kThreads, 256 * 32);
    hipLaunchKernelGGL((msda_bwd_scalar<T>), dim3((unsigned)blocks), dim3(kThreads), 0, st,
                       (const T *)value, shapes, start, (const T *)loc, (const T *)attn, (const T *)go,
                       (typename Acc<T>::type *)gv, (T *)gl, (T *)ga, d, items);
    return hipGetLastError();
}

template <typename T>
static hipError_t dispatch_bwd(const void *value, const int64_t *shapes, const int64_t *start,
                               const void *loc, const void *attn, const void *go, void *gv, void *gl,
                               void *ga, const Dims &d, bool scatter, hipStream_t st, const LevelSel &sel)
{
    constexpr int VEC = 16 / (int)sizeof(T);
    if (d.D % VEC == 0) {
        switch (d.D / VEC) {
#define MMFS_CASE(n) case n: return launch_vec<T, n>(value, shapes, start, loc, attn, go, gv, gl, ga, d, scatter, st, sel);
            MMFS_CASE(1) MMFS_CASE(2) MMFS_CASE(4) MMFS_CASE(8) MMFS_CASE(16) MMFS_CASE(32) MMFS_CASE(64)
#undef MMFS_CASE
            default: break;
        }
    }
    if (sel.n >= 0) return hipErrorInvalidValue;
    return launch_scalar<T>(value, shapes, start, loc, attn, go, gv, gl, ga, d, st);
}

bool bwd_has_vector_path(int dtype, const Dims &d)
{
    if (dtype == 3) return false;
    const int vec = dtype == 0 ? 4 : 8;
    if (d.D % vec) return false;
    const int lpi = d.D / vec;
    return lpi >= 1 && lpi <= 64 && (lpi & (lpi - 1)) == 0;
}

// scatter = true : grad_value accumulated here with atomics into the fp32/fp64 buffer gv
// scatter = false: gv ignored (requires bwd_has_vector_path)
hipError_t backward_taps(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                         const void *loc, const void *attn, const void *grad_out,
                         void *gv, void *gl, void *ga, const Dims &d, bool scatter, hipStream_t st,
                         const LevelSel *sel)
{
    if (!scatter && !bwd_has_vector_path(dtype, d)) return hipErrorInvalidValue;
    LevelSel all;
    all.n = -1;
    const bool routed = sel != nullptr && sel->n >= 0;
    // every level in one kernel, the small ones resident in LDS and contracted on the matrix cores
    if (!scatter && !routed && taps_mma_applies(dtype, d))
        return backward_taps_mma(dtype, value, shapes, start, loc, attn, grad_out, gl, ga, d, st);
    if (routed && scatter) return hipErrorInvalidValue;
    const LevelSel &s = routed ? *sel : all;
    switch (dtype) {
        case 0: return dispatch_bwd<float>(value, shapes, start, loc, attn, grad_out, gv, gl, ga, d, scatter, st, s);
        case 1: return dispatch_bwd<half_t>(value, shapes, start, loc, attn, grad_out, gv, gl, ga, d, scatter, st, s);
        case 2: return dispatch_bwd<bf16_t>(value, shapes, start, loc, attn, grad_out, gv, gl, ga, d, scatter, st, s);
        case 3: if (routed) return hipErrorInvalidValue;
                return launch_scalar<double>(value, shapes, start, loc, attn, grad_out, gv, gl, ga, d, st);
        default: return hipErrorInvalidValue;
    }
}

namespace {
__global__ void __launch_bounds__(256) zero_fill_kernel(unsigned char *__restrict__ p, size_t head, size_t vecs, size_t tail)
{
    // [head bytes][vecs x 16 bytes][tail bytes]: the body starts on a 16-byte boundary
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n = (size_t)gridDim.x * blockDim.x;
    if (i < head) p[i] = 0;
    uint4 *body = reinterpret_cast<uint4 *>(p + head);
    for (size_t k = i; k < vecs; k += n) body[k] = make_uint4(0u, 0u, 0u, 0u);
    if (i < tail) p[head + vecs * 16 + i] = 0;
}
}  // namespace

hipError_t zero_fill(void *ptr, size_t bytes, hipStream_t st)
{
    if (bytes == 0) return hipSuccess;
    unsigned char *p = static_cast<unsigned char *>(ptr);
    const size_t head = std::min<size_t>(bytes, (16 - ((uintptr_t)p & 15)) & 15);
    const size_t vecs = (bytes - head) / 16, tail = bytes - head - vecs * 16;
    const size_t blocks = std::min<size_t>(std::max<size_t>(1, (vecs + 255) / 256), 256 * 16);
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, head, vecs, tail);
    return hipGetLastError();
}

hipError_t cast_from_f32(int dtype, const float *src, void *dst, int64_t n, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    const int64_t blocks = std::min<int64_t>((n / 4 + kThreads) / kThreads, 256 * 16);
    switch (dtype) {
        case 0: hipLaunchKernelGGL(copy_f32_kernel, dim3((unsigned)std::min<int64_t>((n + kThreads - 1) / kThreads, 256 * 16)), dim3(kThreads), 0, st,
                                   src, (float *)dst, n); break;
        case 1: hipLaunchKernelGGL((cast_kernel<half_t>), dim3((unsigned)blocks), dim3(kThreads), 0, st, src, (half_t *)dst, n); break;
        case 2: hipLaunchKernelGGL((cast_kernel<bf16_t>), dim3((unsigned)blocks), dim3(kThreads), 0, st, src, (bf16_t *)dst, n); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace mmfs
