// mmfs_norm.hip -- the RMS normalisation of the LLM-side synchronizer block as ONE pass each way.
//
// Replaces, for the two norms of ``LlamaMMFSAttention`` (mm_interleaved/models/decoders/modeling_llama_mmfs.py:
// 53-70 ``LlamaRMSNorm``; used at :352-353 on the token stream [B, Lq, 4096] and on the feature bank), the chain
// of framework kernels its forward is made of -- to(float32), pow, mean, add, rsqrt, mul, to(16 bit), mul: seven
// launches and ~0.8 GB of traffic for a 67 MB token tensor -- with one read and one write of the tensor.
// Same arithmetic and the same roundings as the reference: statistics in fp32, x * rsqrt(mean(x^2) + eps) in
// fp32, rounded to the storage type when that is 16 bits wide, THEN multiplied by the gain (a 16-bit product,
// rounded again).  The module-level profile that asked for it: profiles/r02_module_bench_cfg3_cfg4.jsonl -- the
// "other" (non-GEMM, non-op) kernels were the largest cost of the LLM path, 3.35 of 6.8 ms at 2048 tokens.
//
// A wave owns a row (kept in registers between the two passes over it), a 256-lane workgroup four rows at a time;
// bytes-bound, every access a 16-byte vector.  Backward: dx = rstd * (g - xn * mean(g * xn)) with g = dy * gain;
// the gain's gradient is summed per lane over the wave's rows, over the workgroup's waves in LDS, and leaves as
// one fp32 atomic per column and workgroup.
#include "../../include/mmfs_msda.h"
#include "msda_env.h"
#include "msda_device.h"
#include <cstdlib>
#include <algorithm>

namespace mmfs {
namespace {

constexpr int kNormThreads = 256;
constexpr int kNormMaxVec = 16;               // 16-byte vectors of a row a lane keeps: rows up to 64 * 16 vectors

template <typename T> struct NormIO {
    static constexpr int N = 16 / (int)sizeof(T);
    static __device__ __forceinline__ void unpack(const uint4 &r, float (&o)[N]) { Vec16<T>::unpack(r, o); }
    static __device__ __forceinline__ uint4 pack(const float (&v)[N]) { return Vec16<T>::pack(v); }
    // round to the storage type and back: what ``.to(weight.dtype)`` does between the two products
    static __device__ __forceinline__ float round(float v) { return (float)(T)v; }
};

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <typename T, int NV>
__global__ void __launch_bounds__(kNormThreads)
rmsnorm_fwd(const T *__restrict__ x, const T *__restrict__ w, T *__restrict__ y, float *__restrict__ rstd_out,
            const int64_t rows, const int C, const float eps)
{
    typedef NormIO<T> IO;
    constexpr int N = IO::N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = C / N;
    const int64_t waves = (int64_t)gridDim.x * (kNormThreads / 64);
    for (int64_t r = (int64_t)blockIdx.x * (kNormThreads / 64) + wave; r < rows; r += waves) {
        const uint4 *xr = reinterpret_cast<const uint4 *>(x + r * C);
        uint4 raw[NV];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = lane + 64 * i;
            raw[i] = make_uint4(0u, 0u, 0u, 0u);
            if (v < nvec) raw[i] = xr[v];
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (lane + 64 * i < nvec) {
                float f[N];
                IO::unpack(raw[i], f);
#pragma unroll
                for (int j = 0; j < N; ++j) ss = fmaf(f[j], f[j], ss);
            }
        }
        ss = wave_sum(ss);
        const float rstd = rsqrtf(ss / (float)C + eps);
        if (lane == 0 && rstd_out != nullptr) rstd_out[r] = rstd;
        uint4 *yr = reinterpret_cast<uint4 *>(y + r * C);
        const uint4 *wr = reinterpret_cast<const uint4 *>(w);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = lane + 64 * i;
            if (v < nvec) {
                float f[N], g[N], o[N];
                IO::unpack(raw[i], f);
                IO::unpack(wr[v], g);
#pragma unroll
                for (int j = 0; j < N; ++j) o[j] = g[j] * (sizeof(T) == 2 ? IO::round(f[j] * rstd) : f[j] * rstd);
                yr[v] = IO::pack(o);
            }
        }
    }
}

template <typename T, int NV>
__global__ void __launch_bounds__(kNormThreads)
rmsnorm_bwd(const T *__restrict__ dy, const T *__restrict__ x, const T *__restrict__ w, const float *__restrict__ rstd_in,
            T *__restrict__ dx, float *__restrict__ dw, const int64_t rows, const int C, const int partials)
{
    typedef NormIO<T> IO;
    constexpr int N = IO::N;
    __shared__ float red[(kNormThreads / 64 - 1) * 64 * 4];       // the other waves' partial gain gradients, one vector slot at a time
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = C / N;
    const int64_t waves = (int64_t)gridDim.x * (kNormThreads / 64);
    float dwa[NV][N];
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) dwa[i][j] = 0.f;
    const uint4 *wr = reinterpret_cast<const uint4 *>(w);
    for (int64_t r = (int64_t)blockIdx.x * (kNormThreads / 64) + wave; r < rows; r += waves) {
        const uint4 *xr = reinterpret_cast<const uint4 *>(x + r * C);
        const uint4 *gr = reinterpret_cast<const uint4 *>(dy + r * C);
        const float rstd = rstd_in[r];
        uint4 xraw[NV], graw[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = lane + 64 * i;
            xraw[i] = graw[i] = make_uint4(0u, 0u, 0u, 0u);
            if (v < nvec) { xraw[i] = xr[v]; graw[i] = gr[v]; }
        }
        float dot = 0.f;                                           // sum_c g * xn
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = lane + 64 * i;
            if (v < nvec) {
                float xf[N], gf[N], wf[N];
                IO::unpack(xraw[i], xf); IO::unpack(graw[i], gf); IO::unpack(wr[v], wf);
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const float xn = xf[j] * rstd;
                    dot = fmaf(gf[j] * wf[j], xn, dot);
                    dwa[i][j] = fmaf(gf[j], sizeof(T) == 2 ? IO::round(xn) : xn, dwa[i][j]);
                }
            }
        }
        dot = wave_sum(dot) / (float)C;
        uint4 *dr = reinterpret_cast<uint4 *>(dx + r * C);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = lane + 64 * i;
            if (v < nvec) {
                float xf[N], gf[N], wf[N], o[N];
                IO::unpack(xraw[i], xf); IO::unpack(graw[i], gf); IO::unpack(wr[v], wf);
#pragma unroll
                for (int j = 0; j < N; ++j) o[j] = rstd * (gf[j] * wf[j] - xf[j] * rstd * dot);
                dr[v] = IO::pack(o);
            }
        }
    }
    // gain gradient: waves 1.. hand their sums to wave 0 through LDS (one vector slot of all lanes at a time),
    // wave 0 adds them up and issues the atomics
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (64 * i >= nvec) continue;                              // (uniform: every thread skips the same slots)
#pragma unroll
        for (int j0 = 0; j0 < N; j0 += 4) {
            __syncthreads();
            if (wave > 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) red[((wave - 1) * 64 + lane) * 4 + j] = dwa[i][j0 + j];
            }
            __syncthreads();
            if (wave == 0) {
                const int v = lane + 64 * i;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float s = dwa[i][j0 + j];
#pragma unroll
                    for (int ow = 0; ow < kNormThreads / 64 - 1; ++ow) s += red[(ow * 64 + lane) * 4 + j];
                    if (v >= nvec) continue;
                    // partials: this workgroup's row of a [gridDim.x, C] fp32 matrix the caller adds up (plain stores: 2 M
                    // atomics on 4096 addresses from 512 workgroups on 8 XCDs took 150 of this kernel's 180 us); else dw += s
                    if (partials) dw[(int64_t)blockIdx.x * C + (int64_t)v * N + j0 + j] = s;
                    else if (s != 0.f)
                        __hip_atomic_fetch_add(dw + (int64_t)v * N + j0 + j, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}

// vectors of a row a lane keeps: the power of two that covers C
int norm_nv(int dtype, int64_t C)
{
    const int64_t per_lane = (C / (16 / (dtype == MMFS_F32 ? 4 : 2)) + 63) / 64;
    int nv = 1;
    while (nv < per_lane) nv *= 2;
    return nv;
}

#define MMFS_NORM_DISPATCH(KERNEL, T, ...)                                                                        \
    switch (nv) {                                                                                                 \
        case 1: hipLaunchKernelGGL((KERNEL<T, 1>), dim3(grid), dim3(kNormThreads), 0, st, __VA_ARGS__); break;    \
        case 2: hipLaunchKernelGGL((KERNEL<T, 2>), dim3(grid), dim3(kNormThreads), 0, st, __VA_ARGS__); break;    \
        case 4: hipLaunchKernelGGL((KERNEL<T, 4>), dim3(grid), dim3(kNormThreads), 0, st, __VA_ARGS__); break;    \
        case 8: hipLaunchKernelGGL((KERNEL<T, 8>), dim3(grid), dim3(kNormThreads), 0, st, __VA_ARGS__); break;    \
        default: hipLaunchKernelGGL((KERNEL<T, 16>), dim3(grid), dim3(kNormThreads), 0, st, __VA_ARGS__); break;  \
    }

int norm_grid(int64_t rows)
{
    const int64_t wgs = (rows + kNormThreads / 64 - 1) / (kNormThreads / 64);
    return (int)(wgs < 1 ? 1 : (wgs > 2048 ? 2048 : wgs));
}

}  // namespace
}  // namespace mmfs

extern "C" {

int mmfs_rmsnorm_supported(int dtype, int64_t C)
{
    const int es = dtype == MMFS_F32 ? 4 : (dtype == MMFS_F16 || dtype == MMFS_BF16) ? 2 : 0;
    if (!es || C <= 0) return 0;
    const int64_t n = 16 / es;
    return C % n == 0 && C / n <= 64 * mmfs::kNormMaxVec;
}

int mmfs_rmsnorm_forward(int dtype, const void *x, const void *weight, void *y, float *rstd,
                         int64_t rows, int64_t C, float eps, void *stream)
{
    if (dtype != MMFS_F32 && dtype != MMFS_F16 && dtype != MMFS_BF16) return MMFS_E_DTYPE;
    if (rows < 0 || C < 0) return MMFS_E_DIMS;
    if (rows == 0 || C == 0) return MMFS_OK;
    if (!mmfs_rmsnorm_supported(dtype, C)) return MMFS_E_UNSUPPORTED;
    if (!x || !weight || !y) return MMFS_E_NULLPTR;
    if (((uintptr_t)x | (uintptr_t)weight | (uintptr_t)y) % 16) return MMFS_E_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int grid = mmfs::norm_grid(rows);
    using namespace mmfs;
    const int nv = norm_nv(dtype, C);
    switch (dtype) {
        case MMFS_F32: MMFS_NORM_DISPATCH(rmsnorm_fwd, float, (const float *)x, (const float *)weight, (float *)y, rstd, rows, (int)C, eps); break;
        case MMFS_F16: MMFS_NORM_DISPATCH(rmsnorm_fwd, half_t, (const half_t *)x, (const half_t *)weight, (half_t *)y, rstd, rows, (int)C, eps); break;
        default: MMFS_NORM_DISPATCH(rmsnorm_fwd, bf16_t, (const bf16_t *)x, (const bf16_t *)weight, (bf16_t *)y, rstd, rows, (int)C, eps); break;
    }
    return (int)hipGetLastError();
}

static int rmsnorm_backward_impl(int dtype, const void *grad_y, const void *x, const void *weight, const float *rstd,
                                 void *grad_x, float *gw, int64_t rows, int64_t C, int grid, int partials, void *stream)
{
    if (dtype != MMFS_F32 && dtype != MMFS_F16 && dtype != MMFS_BF16) return MMFS_E_DTYPE;
    if (rows < 0 || C < 0) return MMFS_E_DIMS;
    if (rows == 0 || C == 0) return MMFS_OK;
    if (!mmfs_rmsnorm_supported(dtype, C)) return MMFS_E_UNSUPPORTED;
    if (!grad_y || !x || !weight || !rstd || !grad_x || !gw) return MMFS_E_NULLPTR;
    if (((uintptr_t)grad_y | (uintptr_t)x | (uintptr_t)weight | (uintptr_t)grad_x) % 16) return MMFS_E_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    using namespace mmfs;
    const int nv = norm_nv(dtype, C);
    switch (dtype) {
        case MMFS_F32: MMFS_NORM_DISPATCH(rmsnorm_bwd, float, (const float *)grad_y, (const float *)x, (const float *)weight, rstd, (float *)grad_x, gw, rows, (int)C, partials); break;
        case MMFS_F16: MMFS_NORM_DISPATCH(rmsnorm_bwd, half_t, (const half_t *)grad_y, (const half_t *)x, (const half_t *)weight, rstd, (half_t *)grad_x, gw, rows, (int)C, partials); break;
        default: MMFS_NORM_DISPATCH(rmsnorm_bwd, bf16_t, (const bf16_t *)grad_y, (const bf16_t *)x, (const bf16_t *)weight, rstd, (bf16_t *)grad_x, gw, rows, (int)C, partials); break;
    }
    return (int)hipGetLastError();
}

int mmfs_rmsnorm_backward(int dtype, const void *grad_y, const void *x, const void *weight, const float *rstd,
                          void *grad_x, float *grad_weight_f32, int64_t rows, int64_t C, void *stream)
{
    const int env_grid = mmfs::knob_int(mmfs::K_NORM_BWD_GRID, 0);      // tuning
    // (few workgroups: the atomics on the gain gradient are what this entry point spends its time on)
    const int grid = std::min(mmfs::norm_grid(rows), env_grid > 0 ? env_grid : 128);
    return rmsnorm_backward_impl(dtype, grad_y, x, weight, rstd, grad_x, grad_weight_f32, rows, C, grid, 0, stream);
}

int mmfs_rmsnorm_backward_partials_rows(int64_t rows)
{
    const int env_grid = mmfs::knob_int(mmfs::K_NORM_BWD_GRID, 0);      // tuning
    return rows <= 0 ? 0 : std::min(mmfs::norm_grid(rows), env_grid > 0 ? env_grid : 512);
}

int mmfs_rmsnorm_backward_partials(int dtype, const void *grad_y, const void *x, const void *weight, const float *rstd,
                                   void *grad_x, float *grad_weight_partials, int64_t rows, int64_t C, void *stream)
{
    return rmsnorm_backward_impl(dtype, grad_y, x, weight, rstd, grad_x, grad_weight_partials, rows, C,
                                 mmfs_rmsnorm_backward_partials_rows(rows), 1, stream);
}

}  // extern "C"
