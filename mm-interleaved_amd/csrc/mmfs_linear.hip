// mmfs_linear.hip -- y = x W^T + b for a HANDFUL of tokens (a decode step: batch x 1 token): a weight-streaming kernel.
//
// At 4 tokens the Linear layers of an MMFS layer (mm_interleaved/models/utils/ops/modules/mmfs.py:174-176, 274: the
// query heads, the output projection) are not matrix products but 4 dot products per weight row: all that happens is
// that the weights stream past once -- 5 + 8 MB per layer at the LLM's width.  The BLAS library's kernels for that shape
// take 8-12 us each (profiles/r03br_decode_kernels.log: 156 of a decode step's 350 us of kernels); this one is bound
// by the stream: a wave owns two weight rows, a lane 16-byte pieces of them 1 KB apart (four requests per row in flight),
// the tokens' activations sit in LDS (read as 16-byte vectors, conflict-free), products by the packed dot-product
// instructions (two exact 16-bit products + fp32 accumulate per lane and instruction: no unpacking), one butterfly per
// (row, token) at the end, bias added in fp32, ONE rounding to the storage type -- what the library's epilogue does.
#include "../../include/mmfs_msda.h"
#include "msda_device.h"
#include "msda_dots.h"

namespace mmfs {
namespace {

constexpr int kLinThreads = 256;
constexpr int kLinRows = 2;                    // weight rows per wave
constexpr int kLinUnroll = 4;                  // 16-byte pieces of a row a lane has in flight

// x [M, K] (rows ldx elements apart), W [N, K] packed, bias [N] or null -> y [M, N] (rows ldy apart); M <= MT
template <typename T, int MT>
__global__ void __launch_bounds__(kLinThreads)
linear_small(const T *__restrict__ x, const T *__restrict__ W, const T *__restrict__ bias, T *__restrict__ y,
             const int M, const int N, const int K, const int64_t ldx, const int64_t ldy)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4 *xs = reinterpret_cast<uint4 *>(smem_raw);                       // [MT][K / 8]
    const int nvec = K / 8;
    for (int i = threadIdx.x; i < MT * nvec; i += kLinThreads) {
        const int m = i / nvec, v = i - m * nvec;
        xs[i] = m < M ? reinterpret_cast<const uint4 *>(x + (int64_t)m * ldx)[v] : make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = (blockIdx.x * (kLinThreads / 64) + wave) * kLinRows;
    if (n0 >= N) return;
    const uint4 *wr[kLinRows];
#pragma unroll
    for (int c = 0; c < kLinRows; ++c) wr[c] = reinterpret_cast<const uint4 *>(W + (int64_t)min(n0 + c, N - 1) * K);
    float acc[kLinRows][MT];
#pragma unroll
    for (int c = 0; c < kLinRows; ++c)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[c][m] = 0.f;
    for (int v0 = lane; v0 < nvec; v0 += 64 * kLinUnroll) {
        uint4 w[kLinUnroll][kLinRows];
#pragma unroll
        for (int u = 0; u < kLinUnroll; ++u) {
            const int v = v0 + 64 * u;
#pragma unroll
            for (int c = 0; c < kLinRows; ++c) w[u][c] = v < nvec ? wr[c][v] : make_uint4(0u, 0u, 0u, 0u);
        }
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
            y[(int64_t)m * ldy + n0 + c] = (T)s;
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

int mmfs_linear_small(int dtype, const void *x, const void *weight, const void *bias, void *y,
                      int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldy, void *stream)
{
    using namespace mmfs;
    if (dtype != MMFS_F32 && dtype != MMFS_F16 && dtype != MMFS_BF16) return MMFS_E_DTYPE;
    if (M < 0 || N < 0 || K < 0) return MMFS_E_DIMS;
    if (M == 0 || N == 0) return MMFS_OK;
    if (!mmfs_linear_small_supported(dtype, M, N, K)) return MMFS_E_UNSUPPORTED;
    if (!x || !weight || !y) return MMFS_E_NULLPTR;
    if (ldx < K || ldy < N) return MMFS_E_DIMS;
    if (((uintptr_t)x | (uintptr_t)weight) % 16 || (ldx * 2) % 16) return MMFS_E_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int per_wg = (kLinThreads / 64) * kLinRows;
    const dim3 grid((unsigned)((N + per_wg - 1) / per_wg));
    const int mt = M <= 4 ? 4 : 8;
    const size_t lds = (size_t)mt * K * 2;
#define MMFS_LIN(T, MT)                                                                                               \
    hipLaunchKernelGGL((linear_small<T, MT>), grid, dim3(kLinThreads), lds, st, (const T *)x, (const T *)weight,      \
                       (const T *)bias, (T *)y, (int)M, (int)N, (int)K, ldx, ldy)
    if (dtype == MMFS_F16) { if (mt == 4) MMFS_LIN(half_t, 4); else MMFS_LIN(half_t, 8); }
    else { if (mt == 4) MMFS_LIN(bf16_t, 4); else MMFS_LIN(bf16_t, 8); }
#undef MMFS_LIN
    return (int)hipGetLastError();
}

}  // extern "C"
