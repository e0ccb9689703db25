// mmfs_query.hip -- the two layout changes around the image decoder's synchronizer block, each as ONE pass.
//
// ``MMFSBlock`` (mm_interleaved/models/decoders/sd_mmfs.py:121-146) receives a UNet residual in the UNet's layout
// [B, C, H, W] and works on tokens [B, H*W, C]:
//     query = LayerNorm(rearrange(sample, "b c h w -> b (h w) c")) + pos_embed          (:124-131)
//     ...
//     return conv(rearrange(out, "b (h w) c -> b c h w"))  -- the caller adds the residual (:262-270)
// With framework kernels that is a transposing copy, the normalisation, an add, and at the other end a strided
// add: four passes over the residual, 0.96 of the 3.7 ms of kernels of a sampling step of the 13 blocks at 512 px
// (profiles/r03at_sample_kernels.log).  Here:
//   * ``query_prep``: a workgroup takes TP consecutive pixels x all C channels of one sample, transposes them
//     through LDS ([TP][C + pad] in the storage type; the global reads are 16-byte vectors along the pixels, the
//     LDS reads 16-byte vectors along the channels), a wave per pixel computes mean and variance in fp32 (two
//     passes over its registers), applies the affine, rounds to the storage type, adds the position row and
//     rounds again -- the roundings of the framework's two kernels;
//   * ``tokens_add``: the way back, y[b, c, p] = round(tok[b, p, c] + res[b, c, p]) through the same tile.
// Bytes-bound: each reads its input once and writes its output once.
#include "../../include/mmfs_msda.h"
#include "msda_env.h"
#include "msda_device.h"
#include <cstdlib>

namespace mmfs {
namespace {

constexpr int kQueryThreads = 256;
constexpr int kQueryPad = 8;                  // elements: rows of the tile stay 16-byte aligned

__device__ __forceinline__ float qwave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// The tile: pixel p's channels as 16-byte vectors, vector v at slot v ^ ((p >> 3) & 7) of row p.  A lane that moves
// 8 pixels of one channel touches 8 rows of one column; the lanes of a wave differ in the pixel octet (p >> 3) and
// in the channel -- the swizzle puts the octets on different banks (row pitch = 32 k + 4 words).
__device__ __forceinline__ int tile_pitch(int C) { return (C + 63) / 64 * 64 + kQueryPad; }
__device__ __forceinline__ int tile_at(int p, int c, int pitch) { return p * pitch + ((((c >> 3) ^ (p >> 3)) & 7) | ((c >> 3) & ~7)) * 8 + (c & 7); }

constexpr int kQueryBatch = 4;                // global loads a lane has in flight

// x [B, C, HW] -> q [B, HW, C]; gridDim = (tiles of TP pixels, B); TP = 8 << pv_shift
template <typename T, int NV>
__global__ void __launch_bounds__(kQueryThreads)
query_prep(const T *__restrict__ x, const T *__restrict__ gamma, const T *__restrict__ beta, const T *__restrict__ pos,
           T *__restrict__ q, float *__restrict__ mean_out, float *__restrict__ rstd_out,
           const int C, const int HW, const int pv_shift, const float eps)
{
    typedef Vec16<T> V;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint16_t *tile = reinterpret_cast<uint16_t *>(smem_raw);
    const int pitch = tile_pitch(C);
    const int TP = 8 << pv_shift;
    const int b = blockIdx.y, p0 = blockIdx.x * TP;
    const int npx = min(TP, HW - p0);
    const int total = C << pv_shift;
    const uint16_t *xb = reinterpret_cast<const uint16_t *>(x) + (int64_t)b * C * HW + p0;
    // ---- in: 8 pixels of one channel per lane, scattered down a column of the tile
    for (int u0 = threadIdx.x; u0 < total; u0 += kQueryBatch * kQueryThreads) {
        uint4 r[kQueryBatch];
#pragma unroll
        for (int k = 0; k < kQueryBatch; ++k) {
            const int u = u0 + k * kQueryThreads;
            const int c = u >> pv_shift, pv = u & ((1 << pv_shift) - 1);
            if (u < total && pv * 8 < npx) r[k] = *reinterpret_cast<const uint4 *>(xb + (int64_t)c * HW + pv * 8);
        }
#pragma unroll
        for (int k = 0; k < kQueryBatch; ++k) {
            const int u = u0 + k * kQueryThreads;
            const int c = u >> pv_shift, pv = u & ((1 << pv_shift) - 1);
            if (u < total && pv * 8 < npx) {
                uint16_t *col = tile + tile_at(pv * 8, c, pitch);
                col[0 * pitch] = (uint16_t)(r[k].x & 0xffffu); col[1 * pitch] = (uint16_t)(r[k].x >> 16);
                col[2 * pitch] = (uint16_t)(r[k].y & 0xffffu); col[3 * pitch] = (uint16_t)(r[k].y >> 16);
                col[4 * pitch] = (uint16_t)(r[k].z & 0xffffu); col[5 * pitch] = (uint16_t)(r[k].z >> 16);
                col[6 * pitch] = (uint16_t)(r[k].w & 0xffffu); col[7 * pitch] = (uint16_t)(r[k].w >> 16);
            }
        }
    }
    // ---- out: a wave per pixel; a lane's channels are the same for every pixel, so is its share of the affine
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = C / 8;
    float g[NV][8], bt[NV][8];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = lane + 64 * i;
        if (v < nvec) {
            V::unpack(reinterpret_cast<const uint4 *>(gamma)[v], g[i]);
            V::unpack(reinterpret_cast<const uint4 *>(beta)[v], bt[i]);
        }
    }
    __syncthreads();
    const float inv_c = 1.f / (float)C;
    for (int p = wave; p < npx; p += kQueryThreads / 64) {
        const int64_t tok = (int64_t)b * HW + p0 + p;
        uint4 praw[NV];
        if (pos != nullptr) {                                   // (requested before the statistics: their latency hides it)
            const uint4 *pr = reinterpret_cast<const uint4 *>(pos + (int64_t)(p0 + p) * C);
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (lane + 64 * i < nvec) praw[i] = pr[lane + 64 * i];
        }
        const uint4 *row = reinterpret_cast<const uint4 *>(tile + p * pitch);
        const int sw = (p >> 3) & 7;
        float f[NV][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = lane + 64 * i;
            if (v < nvec) {
                V::unpack(row[v ^ sw], f[i]);
#pragma unroll
                for (int j = 0; j < 8; ++j) s += f[i][j];
            }
        }
        const float mean = qwave_sum(s) * inv_c;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (lane + 64 * i < nvec) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = f[i][j] - mean; ss = fmaf(d, d, ss); }
            }
        }
        const float rstd = rsqrtf(qwave_sum(ss) * inv_c + eps);
        if (lane == 0 && mean_out != nullptr) { mean_out[tok] = mean; rstd_out[tok] = rstd; }
        uint4 *qr = reinterpret_cast<uint4 *>(q + tok * C);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = lane + 64 * i;
            if (v < nvec) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = fmaf((f[i][j] - mean) * rstd, g[i][j], bt[i][j]);
                if (pos != nullptr) {
                    float pe[8];
                    V::unpack(praw[i], pe);
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (float)(T)o[j] + pe[j];       // (rounded between the two, as two kernels)
                }
                qr[v] = V::pack(o);
            }
        }
    }
}

// tok [B, HW, C] + res [B, C, HW] -> y [B, C, HW]; a workgroup moves kAddPx pixels x kAddCh channels (nothing here
// spans the channels): gridDim = (pixel tiles, channel tiles, B); 256-byte pieces of token rows in, 128-byte pieces of
// channel rows out, one batch of loads per lane each way
constexpr int kAddPx = 64, kAddCh = 128;
constexpr int kAddPitch = kAddCh + kQueryPad;
constexpr int kAddUnits = kAddPx * kAddCh / 8 / kQueryThreads;          // 16-byte vectors per lane: 4

template <typename T>
__global__ void __launch_bounds__(kQueryThreads)
tokens_add(const T *__restrict__ tok, const T *__restrict__ res, T *__restrict__ y, const int C, const int HW)
{
    typedef Vec16<T> V;
    __shared__ __attribute__((aligned(16))) uint16_t tile[kAddPx * kAddPitch];
    const int b = blockIdx.z, p0 = blockIdx.x * kAddPx, c0 = blockIdx.y * kAddCh;
    const int npx = min(kAddPx, HW - p0), nch = min(kAddCh, C - c0);
    const T *tb = tok + ((int64_t)b * HW + p0) * C + c0;
    const int64_t base = ((int64_t)b * C + c0) * HW + p0;
    uint4 rt[kAddUnits], rr[kAddUnits];
#pragma unroll
    for (int k = 0; k < kAddUnits; ++k) {                                // token rows: 16 vectors per pixel
        const int u = threadIdx.x + k * kQueryThreads;
        const int p = u >> 4, v = u & 15;
        if (p < npx && v * 8 < nch) rt[k] = *reinterpret_cast<const uint4 *>(tb + (int64_t)p * C + v * 8);
    }
#pragma unroll
    for (int k = 0; k < kAddUnits; ++k) {                                // residual: 8 pixel octets per channel
        const int u = threadIdx.x + k * kQueryThreads;
        const int c = u >> 3, pv = u & 7;
        if (c < nch && pv * 8 < npx) rr[k] = *reinterpret_cast<const uint4 *>(res + base + (int64_t)c * HW + pv * 8);
    }
#pragma unroll
    for (int k = 0; k < kAddUnits; ++k) {
        const int u = threadIdx.x + k * kQueryThreads;
        const int p = u >> 4, v = u & 15;
        if (p < npx && v * 8 < nch) *reinterpret_cast<uint4 *>(tile + tile_at(p, v * 8, kAddPitch)) = rt[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kAddUnits; ++k) {
        const int u = threadIdx.x + k * kQueryThreads;
        const int c = u >> 3, pv = u & 7;
        if (c < nch && pv * 8 < npx) {
            const uint16_t *col = tile + tile_at(pv * 8, c, kAddPitch);
            uint4 t;
            t.x = (uint32_t)col[0 * kAddPitch] | ((uint32_t)col[1 * kAddPitch] << 16);
            t.y = (uint32_t)col[2 * kAddPitch] | ((uint32_t)col[3 * kAddPitch] << 16);
            t.z = (uint32_t)col[4 * kAddPitch] | ((uint32_t)col[5 * kAddPitch] << 16);
            t.w = (uint32_t)col[6 * kAddPitch] | ((uint32_t)col[7 * kAddPitch] << 16);
            float a[8], r[8];
            V::unpack(t, a);
            V::unpack(rr[k], r);
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += r[j];
            *reinterpret_cast<uint4 *>(y + base + (int64_t)c * HW + pv * 8) = V::pack(a);
        }
    }
}

// query_prep's pixels per tile, as 8 << shift: the most that keeps the tile within 25 KB (six workgroups of four waves
// per CU: the loops are latency-bound per workgroup; 32 pixels at 320 channels, 16 at 640, 8 at 1280); MMFS_QUERY_LDS_KB: tuning
int query_tile_shift(int64_t C, int64_t HW)
{
    const int env_kb = mmfs::knob_int(mmfs::K_QUERY_LDS_KB, 0);
    const int64_t budget = (int64_t)(env_kb > 0 ? env_kb : 25) * 1024;
    const int64_t pitch = (C + 63) / 64 * 64 + kQueryPad;
    int sh = 3;
    while (sh > 0 && (int64_t)(8 << sh) * pitch * 2 > budget) --sh;
    while (sh > 0 && (8 << (sh - 1)) >= HW) --sh;
    return sh;
}

bool query_ok(int dtype, int64_t C, int64_t HW)
{
    return (dtype == MMFS_F16 || dtype == MMFS_BF16) && C > 0 && HW > 0 && C % 8 == 0 && HW % 8 == 0 && C <= 2048;
}

}  // namespace
}  // namespace mmfs

extern "C" {

int mmfs_query_prep_supported(int dtype, int64_t C, int64_t HW) { return mmfs::query_ok(dtype, C, HW) ? 1 : 0; }

int mmfs_query_prep(int dtype, const void *x, const void *gamma, const void *beta, const void *pos, void *q,
                    float *mean, float *rstd, int64_t B, int64_t C, int64_t HW, float eps, void *stream)
{
    using namespace mmfs;
    if (dtype != MMFS_F32 && dtype != MMFS_F16 && dtype != MMFS_BF16) return MMFS_E_DTYPE;
    if (B < 0 || C < 0 || HW < 0) return MMFS_E_DIMS;
    if (B == 0 || C == 0 || HW == 0) return MMFS_OK;
    if (!query_ok(dtype, C, HW) || B > 65535) return MMFS_E_UNSUPPORTED;
    if (!x || !gamma || !beta || !q || ((mean == nullptr) != (rstd == nullptr))) return MMFS_E_NULLPTR;
    if (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)pos | (uintptr_t)q) % 16) return MMFS_E_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int sh = query_tile_shift(C, HW), tp = 8 << sh;
    const dim3 grid((unsigned)((HW + tp - 1) / tp), (unsigned)B);
    const size_t lds = (size_t)tp * ((C + 63) / 64 * 64 + kQueryPad) * 2;
    const int nv = (int)((C / 8 + 63) / 64);
#define MMFS_QP(T, NV)                                                                                              \
    hipLaunchKernelGGL((query_prep<T, NV>), grid, dim3(kQueryThreads), lds, st, (const T *)x, (const T *)gamma,     \
                       (const T *)beta, (const T *)pos, (T *)q, mean, rstd, (int)C, (int)HW, sh, eps)
#define MMFS_QP_NV(T)                                                                                               \
    do { if (nv <= 1) MMFS_QP(T, 1); else if (nv == 2) MMFS_QP(T, 2); else MMFS_QP(T, 4); } while (0)
    if (dtype == MMFS_F16) MMFS_QP_NV(half_t); else MMFS_QP_NV(bf16_t);
#undef MMFS_QP_NV
#undef MMFS_QP
    return (int)hipGetLastError();
}

int mmfs_tokens_add(int dtype, const void *tok, const void *res, void *y, int64_t B, int64_t C, int64_t HW, void *stream)
{
    using namespace mmfs;
    if (dtype != MMFS_F32 && dtype != MMFS_F16 && dtype != MMFS_BF16) return MMFS_E_DTYPE;
    if (B < 0 || C < 0 || HW < 0) return MMFS_E_DIMS;
    if (B == 0 || C == 0 || HW == 0) return MMFS_OK;
    if (!query_ok(dtype, C, HW) || B > 65535) return MMFS_E_UNSUPPORTED;
    if (!tok || !res || !y) return MMFS_E_NULLPTR;
    if (((uintptr_t)tok | (uintptr_t)res | (uintptr_t)y) % 16) return MMFS_E_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((HW + kAddPx - 1) / kAddPx), (unsigned)((C + kAddCh - 1) / kAddCh), (unsigned)B);
    if (dtype == MMFS_F16)
        hipLaunchKernelGGL((tokens_add<half_t>), grid, dim3(kQueryThreads), 0, st, (const half_t *)tok, (const half_t *)res,
                           (half_t *)y, (int)C, (int)HW);
    else
        hipLaunchKernelGGL((tokens_add<bf16_t>), grid, dim3(kQueryThreads), 0, st, (const bf16_t *)tok, (const bf16_t *)res,
                           (bf16_t *)y, (int)C, (int)HW);
    return (int)hipGetLastError();
}

}  // extern "C"
