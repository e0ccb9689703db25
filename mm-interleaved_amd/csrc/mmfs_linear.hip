// mmfs_linear.hip -- y = x W^T + b for a HANDFUL of tokens (a decode step: batch x 1 token): a weight-streaming kernel.
//
// At 4 tokens the Linear layers of an MMFS layer (mm_interleaved/models/utils/ops/modules/mmfs.py:174-176, 274: the
// query heads, the output projection) are not matrix products but 4 dot products per weight row: all that happens is
// that the weights stream past once -- 5 + 8 MB per layer at the LLM's width.  The BLAS library's kernels for that shape
// take 8-12 us each (profiles/r03br_decode_kernels.log: 156 of a decode step's 350 us of kernels); this one is bound
// by the stream: a wave owns a weight row, a lane 16-byte pieces of it 1 KB apart (eight requests in flight, the first batch issued
// before the activations are staged),
// the tokens' activations sit in LDS (read as 16-byte vectors, conflict-free), products by the packed dot-product
// instructions (two exact 16-bit products + fp32 accumulate per lane and instruction: no unpacking), one butterfly per
// (row, token) at the end, bias added in fp32, ONE rounding to the storage type -- what the library's epilogue does.
#include "../../include/mmfs_msda.h"
#include "msda_env.h"
#include "msda_device.h"
#include "msda_dots.h"
#include <cstdlib>

namespace mmfs {
namespace {

constexpr int kLinThreads = 256;
// (weight rows per wave R, 16-byte pieces of a row a lane has in flight U: template parameters; MMFS_LIN_ROWS /
// MMFS_LIN_UNROLL: tuning)

// x [M, K] (rows ldx elements apart), W [N, K] packed, bias [N] or null, res [M, N] (rows ldr apart) or null -> y [M, N]
// (rows ldy apart); M <= MT
template <typename T, int MT, int kLinRows, int kLinUnroll>
__global__ void __launch_bounds__(kLinThreads)
linear_small(const T *__restrict__ x, const T *__restrict__ W, const T *__restrict__ bias, const T *__restrict__ res,
             T *__restrict__ y, const int M, const int N, const int K, const int64_t ldx, const int64_t ldy,
             const int64_t ldr, const int early)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4 *xs = reinterpret_cast<uint4 *>(smem_raw);                       // [MT][K / 8]
    const int nvec = K / 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = (blockIdx.x * (kLinThreads / 64) + wave) * kLinRows;
    const uint4 *wr[kLinRows];
#pragma unroll
    for (int c = 0; c < kLinRows; ++c) wr[c] = reinterpret_cast<const uint4 *>(W + (int64_t)min(max(n0 + c, 0), N - 1) * K);
    // the first batch of the rows' pieces is requested BEFORE the activations are staged: two round trips become one
    uint4 w[kLinUnroll][kLinRows];
    auto request = [&](int v0) {
#pragma unroll
        for (int u = 0; u < kLinUnroll; ++u) {
            const int v = v0 + 64 * u;
#pragma unroll
            for (int c = 0; c < kLinRows; ++c) w[u][c] = (v < nvec && n0 < N) ? wr[c][v] : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    if (early) request(lane);
    for (int i = threadIdx.x; i < MT * nvec; i += kLinThreads) {
        const int m = i / nvec, v = i - m * nvec;
        xs[i] = m < M ? reinterpret_cast<const uint4 *>(x + (int64_t)m * ldx)[v] : make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    if (n0 >= N) return;
    float acc[kLinRows][MT];
#pragma unroll
    for (int c = 0; c < kLinRows; ++c)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[c][m] = 0.f;
    for (int v0 = lane; v0 < nvec; v0 += 64 * kLinUnroll) {
        if (v0 != lane || !early) request(v0);
#pragma unroll
        for (int u = 0; u < kLinUnroll; ++u) {
            const int v = v0 + 64 * u;
            if (v >= nvec) break;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const uint4 xv = xs[m * nvec + v];
#pragma unroll
                for (int c = 0; c < kLinRows; ++c) acc[c][m] += RowDot<T>::run(w[u][c], xv);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < kLinRows; ++c)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float s = acc[c][m];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            acc[c][m] = s;
        }
    // lane (c, m) writes y[m][n0 + c]
    if (lane < kLinRows * MT) {
        const int c = lane / MT, m = lane % MT;
        if (m < M && n0 + c < N) {
            float s = 0.f;
#pragma unroll
            for (int cc = 0; cc < kLinRows; ++cc)
#pragma unroll
                for (int mm = 0; mm < MT; ++mm) s = (cc == c && mm == m) ? acc[cc][mm] : s;
            if (bias != nullptr) s += to_f32(bias[n0 + c]);
            T r = (T)s;
            // (+ the caller's residual, as the framework's add that would follow: a second rounding, of the sum of two stored values)
            if (res != nullptr) r = (T)(to_f32(r) + to_f32(res[(int64_t)m * ldr + n0 + c]));
            y[(int64_t)m * ldy + n0 + c] = r;
        }
    }
}

}  // namespace
}  // namespace mmfs

extern "C" {

int mmfs_linear_small_supported(int dtype, int64_t M, int64_t N, int64_t K)
{
    if (dtype != MMFS_F16 && dtype != MMFS_BF16) return 0;
    if (M < 1 || M > 8 || N < 1 || K < 8 || K % 8) return 0;
    const int64_t mt = M <= 4 ? 4 : 8;
    return mt * K * 2 <= 64 * 1024 && N <= 0x3fffffffLL;
}

int mmfs_linear_small_add(int dtype, const void *x, const void *weight, const void *bias, const void *residual, void *y,
                          int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldy, int64_t ldr, void *stream)
{
    using namespace mmfs;
    if (dtype != MMFS_F32 && dtype != MMFS_F16 && dtype != MMFS_BF16) return MMFS_E_DTYPE;
    if (M < 0 || N < 0 || K < 0) return MMFS_E_DIMS;
    if (M == 0 || N == 0) return MMFS_OK;
    if (!mmfs_linear_small_supported(dtype, M, N, K)) return MMFS_E_UNSUPPORTED;
    if (!x || !weight || !y) return MMFS_E_NULLPTR;
    if (ldx < K || ldy < N || (residual && ldr < N)) return MMFS_E_DIMS;
    if (((uintptr_t)x | (uintptr_t)weight) % 16 || (ldx * 2) % 16) return MMFS_E_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int env_rows = mmfs::knob_int(mmfs::K_LIN_ROWS, 0);
    const int env_unroll = mmfs::knob_int(mmfs::K_LIN_UNROLL, 0);
    const int rows = env_rows == 1 || env_rows == 2 ? env_rows : 1;             // (r03bv: 1 row per wave, 8 pieces in flight)
    const int unroll = env_unroll == 4 || env_unroll == 8 ? env_unroll : 8;
    const int early = mmfs::knob_int(mmfs::K_LIN_EARLY, 1);               // tuning
    const int per_wg = (kLinThreads / 64) * rows;
    const dim3 grid((unsigned)((N + per_wg - 1) / per_wg));
    const int mt = M <= 4 ? 4 : 8;
    const size_t lds = (size_t)mt * K * 2;
#define MMFS_LIN(T, MT, R, U)                                                                                         \
    hipLaunchKernelGGL((linear_small<T, MT, R, U>), grid, dim3(kLinThreads), lds, st, (const T *)x, (const T *)weight, \
                       (const T *)bias, (const T *)residual, (T *)y, (int)M, (int)N, (int)K, ldx, ldy, ldr, early)
#define MMFS_LIN_RU(T, MT)                                                                                            \
    do {                                                                                                              \
        if (rows == 1) { if (unroll == 8) MMFS_LIN(T, MT, 1, 8); else MMFS_LIN(T, MT, 1, 4); }                        \
        else { if (unroll == 8) MMFS_LIN(T, MT, 2, 8); else MMFS_LIN(T, MT, 2, 4); }                                  \
    } while (0)
    if (dtype == MMFS_F16) { if (mt == 4) MMFS_LIN_RU(half_t, 4); else MMFS_LIN_RU(half_t, 8); }
    else { if (mt == 4) MMFS_LIN_RU(bf16_t, 4); else MMFS_LIN_RU(bf16_t, 8); }
#undef MMFS_LIN_RU
#undef MMFS_LIN
    return (int)hipGetLastError();
}

int mmfs_linear_small(int dtype, const void *x, const void *weight, const void *bias, void *y,
                      int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldy, void *stream)
{
    return mmfs_linear_small_add(dtype, x, weight, bias, nullptr, y, M, N, K, ldx, ldy, 0, stream);
}

}  // extern "C"
