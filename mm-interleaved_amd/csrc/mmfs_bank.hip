// mmfs_bank.hip -- the multi-image feature bank, built on the device in one pass.
//
// What it replaces: the reference assembles MMFS's ``input_flatten`` with Python loops over the
// batch -- zero-filled [B, n, C, h, w] buffers per level, slice copies per sequence, a
// "b n c h w -> b n (h w) c" rearrange per level and a concatenation over levels
// (mm_interleaved/models/mm_interleaved.py:223-250; the UNet side does the rearrange + concatenation
// in sd_mmfs.py:241-245).  That is three passes over the bank, one of them a transposition read
// with a 2-byte stride.  Here: per-level feature maps [N_img, C, h_l, w_l] (channel-major, as the
// encoder emits them) + an index per bank slot -> the token-major bank [n_slots, sum_l h_l*w_l, C],
// every element read once and written once; slots whose index is negative are zero rows.
//
// The kernel is a tiled transposition: a 256-lane workgroup moves a 64-token x 64-channel tile
// through LDS -- 16-byte reads along the token axis of the source, 16-byte writes along the channel
// axis of the bank (scalar accesses when a level's h*w or C is not a multiple of the vector) -- so
// both sides are full cache lines.  HBM-bound: bytes = 2 * n_slots * S * C * e.
//
// The backward pass (an encoder that is trained needs d bank / d features) is the same tile walked
// the other way: for image i, the tiles of every slot that shows image i are summed in fp32
// registers, transposed through LDS and stored channel-major; an image no slot shows gets zeros.
// Deterministic (no atomics): the slots that show an image are found by ballot, walked in order.
#include "msda_device.h"
#include "../../include/mmfs_msda.h"

namespace mmfs {
namespace {

constexpr int kBankThreads = 256;
constexpr int kTile = 64;                 // tokens x channels per workgroup
constexpr int kMaxBankLevels = 8;

struct BankLevel {
    void *ptr;        // [N_img, C, hw]
    int hw;           // tokens of the level
    int tok0;         // first token of the level in the bank row
    int tile0;        // first token tile of the level
};
struct BankLevels {
    int n, tiles, S;  // levels, token tiles over all levels, tokens per image
    BankLevel lv[kMaxBankLevels];
};

template <int BYTES> struct Word;
template <> struct Word<2> { typedef uint16_t type; };
template <> struct Word<4> { typedef uint32_t type; };

__device__ __forceinline__ int level_of(const BankLevels &L, int tile)
{
    int l = 0;
    while (l + 1 < L.n && tile >= L.lv[l + 1].tile0) ++l;
    return l;
}

// ---------------------------------------------------------------- forward: gather + transpose
template <int BYTES>
__global__ void __launch_bounds__(kBankThreads)
bank_gather_kernel(const BankLevels L, const int64_t *__restrict__ src_index, void *__restrict__ bank_,
                   const int C, const int64_t n_img, const int c_tiles)
{
    typedef typename Word<BYTES>::type E;
    constexpr int VEC = 16 / BYTES;                     // elements per 16-byte access
    constexpr int PITCH = kTile + 4 / BYTES;            // +1 dword: transposed reads hit distinct banks
    __shared__ __attribute__((aligned(16))) E tile[kTile * PITCH];

    const int tid = threadIdx.x;
    const int ttile = blockIdx.x / c_tiles, ctile = blockIdx.x - ttile * c_tiles;
    const int slot = blockIdx.y;
    const int l = level_of(L, ttile);
    const int hw = L.lv[l].hw;
    const int t0 = (ttile - L.lv[l].tile0) * kTile, c0 = ctile * kTile;
    const int nt = min(kTile, hw - t0), nc = min(kTile, C - c0);
    const int64_t img = src_index[slot];
    const bool live = img >= 0 && img < n_img;
    E *bank = (E *)bank_ + ((int64_t)slot * L.S + L.lv[l].tok0 + t0) * C + c0;

    if (live) {
        const E *src = (const E *)L.lv[l].ptr + ((int64_t)img * C + c0) * hw + t0;
        if (hw % VEC == 0 && ((uintptr_t)L.lv[l].ptr & 15) == 0) {
            constexpr int VPR = kTile / VEC;            // vectors per channel row
            for (int i = tid; i < kTile * VPR; i += kBankThreads) {
                const int c = i / VPR, v = i - c * VPR;
                if (c < nc && v * VEC < nt) {           // (hw % VEC == 0: a vector is all inside or all outside)
                    const uint4 x = *reinterpret_cast<const uint4 *>(src + (int64_t)c * hw + v * VEC);
                    uint32_t *dst = reinterpret_cast<uint32_t *>(tile + c * PITCH + v * VEC);
                    dst[0] = x.x; dst[1] = x.y; dst[2] = x.z; dst[3] = x.w;
                }
            }
        } else {
            for (int i = tid; i < kTile * kTile; i += kBankThreads) {
                const int c = i / kTile, t = i - c * kTile;
                if (c < nc && t < nt) tile[c * PITCH + t] = src[(int64_t)c * hw + t];
            }
        }
        __syncthreads();
    }
    if (C % VEC == 0 && ((uintptr_t)bank_ & 15) == 0) {
        constexpr int VPT = kTile / VEC;                // vectors per token
        for (int i = tid; i < kTile * VPT; i += kBankThreads) {
            const int t = i / VPT, v = i - t * VPT;
            if (t < nt && v * VEC < nc) {
                E x[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) x[j] = live ? tile[(v * VEC + j) * PITCH + t] : (E)0;
                *reinterpret_cast<uint4 *>(bank + (int64_t)t * C + v * VEC) = *reinterpret_cast<const uint4 *>(x);
            }
        }
    } else {
        for (int i = tid; i < kTile * kTile; i += kBankThreads) {
            const int t = i / kTile, c = i - t * kTile;
            if (t < nt && c < nc) bank[(int64_t)t * C + c] = live ? tile[c * PITCH + t] : (E)0;
        }
    }
}

// ---------------------------------------------------------------- backward: sum over slots + transpose back
template <typename T>
__global__ void __launch_bounds__(kBankThreads)
bank_scatter_kernel(const BankLevels L, const int64_t *__restrict__ src_index, const T *__restrict__ grad_bank,
                    const int C, const int n_slots, const int c_tiles)
{
    typedef Vec16<T> V;
    constexpr int VEC = V::N;                           // elements per 16-byte access
    constexpr int VPT = kTile / VEC;                    // vectors per token (and per channel row)
    constexpr int TPP = kBankThreads / VPT;             // tokens per pass of the workgroup
    constexpr int PASSES = kTile / TPP;
    constexpr int PITCH = kTile + 1;
    __shared__ float tile[kTile * PITCH];
    __shared__ unsigned long long masks[kBankThreads / 64];

    const int tid = threadIdx.x;
    const int ttile = blockIdx.x / c_tiles, ctile = blockIdx.x - ttile * c_tiles;
    const int64_t img = blockIdx.y;
    const int l = level_of(L, ttile);
    const int hw = L.lv[l].hw;
    const int t0 = (ttile - L.lv[l].tile0) * kTile, c0 = ctile * kTile;
    const int nt = min(kTile, hw - t0), nc = min(kTile, C - c0);
    const bool vec_in = C % VEC == 0 && ((uintptr_t)grad_bank & 15) == 0;
    const bool vec_out = hw % VEC == 0 && ((uintptr_t)L.lv[l].ptr & 15) == 0;

    // lane -> (token, vector of channels): the bank rows are read as full lines
    float acc[PASSES][VEC];
#pragma unroll
    for (int k = 0; k < PASSES; ++k)
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[k][j] = 0.f;
    const int v = tid % VPT, tq = tid / VPT;
    // which slots show this image: every lane tests one slot of a 256-slot step, the four wave
    // ballots go through LDS, and all lanes then walk the set bits in ascending order (the order of
    // the sum is fixed; the walk costs one iteration per MATCH, not per slot)
    for (int s0 = 0; s0 < n_slots; s0 += kBankThreads) {
        const bool mine = s0 + tid < n_slots && src_index[s0 + tid] == img;
        const unsigned long long bal = __ballot(mine);
        __syncthreads();                                // (the previous step's masks are all read)
        if ((tid & 63) == 0) masks[tid >> 6] = bal;
        __syncthreads();
        for (int w = 0; w < kBankThreads / 64; ++w) {
            unsigned long long m = masks[w];
            while (m) {
                const int s = s0 + w * 64 + __builtin_ctzll(m);
                m &= m - 1;
                const T *g = grad_bank + ((int64_t)s * L.S + L.lv[l].tok0 + t0) * C + c0 + v * VEC;
#pragma unroll
                for (int k = 0; k < PASSES; ++k) {
                    const int t = tq + k * TPP;
                    if (t >= nt || v * VEC >= nc) continue;
                    if (vec_in) {                       // (C % VEC == 0: a vector is all inside or all outside)
                        float x[VEC];
                        V::unpack(*reinterpret_cast<const uint4 *>(g + (int64_t)t * C), x);
#pragma unroll
                        for (int j = 0; j < VEC; ++j) acc[k][j] += x[j];
                    } else {
#pragma unroll
                        for (int j = 0; j < VEC; ++j)
                            if (v * VEC + j < nc) acc[k][j] += to_f32(g[(int64_t)t * C + j]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < PASSES; ++k)
#pragma unroll
        for (int j = 0; j < VEC; ++j) tile[(v * VEC + j) * PITCH + tq + k * TPP] = acc[k][j];
    __syncthreads();
    // tokens fastest: the maps are written as full lines
    T *dst = (T *)L.lv[l].ptr + ((int64_t)img * C + c0) * hw + t0;
    if (vec_out) {
        for (int i = tid; i < kTile * VPT; i += kBankThreads) {
            const int cc = i / VPT, vv = i - cc * VPT;
            if (cc < nc && vv * VEC < nt) {
                float x[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) x[j] = tile[cc * PITCH + vv * VEC + j];
                *reinterpret_cast<uint4 *>(dst + (int64_t)cc * hw + vv * VEC) = V::pack(x);
            }
        }
    } else {
        for (int i = tid; i < kTile * kTile; i += kBankThreads) {
            const int cc = i / kTile, t = i - cc * kTile;
            if (cc < nc && t < nt) dst[(int64_t)cc * hw + t] = (T)tile[cc * PITCH + t];
        }
    }
}

int make_levels(int n_levels, void *const *ptrs, const int64_t *hw, BankLevels *out)
{
    if (n_levels < 1 || n_levels > kMaxBankLevels) return MMFS_E_UNSUPPORTED;
    if (!ptrs || !hw) return MMFS_E_NULLPTR;
    int64_t tok = 0, tiles = 0;
    out->n = n_levels;
    for (int l = 0; l < n_levels; ++l) {
        if (hw[l] <= 0 || hw[l] > (1 << 28)) return MMFS_E_DIMS;
        if (!ptrs[l]) return MMFS_E_NULLPTR;
        out->lv[l].ptr = ptrs[l];
        out->lv[l].hw = (int)hw[l];
        out->lv[l].tok0 = (int)tok;
        out->lv[l].tile0 = (int)tiles;
        tok += hw[l];
        tiles += (hw[l] + kTile - 1) / kTile;
        if (tok > (1 << 28)) return MMFS_E_DIMS;
    }
    out->S = (int)tok;
    out->tiles = (int)tiles;
    return MMFS_OK;
}

}  // namespace
}  // namespace mmfs

extern "C" {

int mmfs_bank_gather(int dtype, int n_levels, const void *const *level_ptrs, const int64_t *level_hw,
                     const int64_t *src_index, void *bank, int64_t n_img, int64_t C, int64_t n_slots,
                     void *stream)
{
    using namespace mmfs;
    if (dtype != MMFS_F32 && dtype != MMFS_F16 && dtype != MMFS_BF16) return MMFS_E_DTYPE;
    if (n_img < 0 || C < 0 || n_slots < 0 || C > (1 << 24) || n_slots > 65535) return MMFS_E_DIMS;
    BankLevels L;
    const int rc = make_levels(n_levels, const_cast<void *const *>(level_ptrs), level_hw, &L);
    if (rc) return rc;
    if (n_slots == 0 || C == 0) return MMFS_OK;
    if (!src_index || !bank) return MMFS_E_NULLPTR;
    const int es = dtype == MMFS_F32 ? 4 : 2;
    if (((uintptr_t)bank % es)) return MMFS_E_ALIGN;
    const int c_tiles = (int)((C + kTile - 1) / kTile);
    if ((int64_t)L.tiles * c_tiles > 0x7fffffffLL) return MMFS_E_DIMS;
    const dim3 grid((unsigned)(L.tiles * c_tiles), (unsigned)n_slots);
    hipStream_t st = (hipStream_t)stream;
    if (es == 4)
        hipLaunchKernelGGL((bank_gather_kernel<4>), grid, dim3(kBankThreads), 0, st, L, src_index, bank, (int)C, n_img, c_tiles);
    else
        hipLaunchKernelGGL((bank_gather_kernel<2>), grid, dim3(kBankThreads), 0, st, L, src_index, bank, (int)C, n_img, c_tiles);
    return (int)hipGetLastError();
}

int mmfs_bank_scatter(int dtype, int n_levels, void *const *grad_level_ptrs, const int64_t *level_hw,
                      const int64_t *src_index, const void *grad_bank, int64_t n_img, int64_t C,
                      int64_t n_slots, void *stream)
{
    using namespace mmfs;
    if (dtype != MMFS_F32 && dtype != MMFS_F16 && dtype != MMFS_BF16) return MMFS_E_DTYPE;
    if (n_img < 0 || C < 0 || n_slots < 0 || C > (1 << 24) || n_img > 65535 || n_slots > 0x7fffffffLL) return MMFS_E_DIMS;
    BankLevels L;
    const int rc = make_levels(n_levels, grad_level_ptrs, level_hw, &L);
    if (rc) return rc;
    if (n_img == 0 || C == 0) return MMFS_OK;
    if (n_slots > 0 && (!src_index || !grad_bank)) return MMFS_E_NULLPTR;
    const int c_tiles = (int)((C + kTile - 1) / kTile);
    if ((int64_t)L.tiles * c_tiles > 0x7fffffffLL) return MMFS_E_DIMS;
    const dim3 grid((unsigned)(L.tiles * c_tiles), (unsigned)n_img);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MMFS_F32)
        hipLaunchKernelGGL((bank_scatter_kernel<float>), grid, dim3(kBankThreads), 0, st, L, src_index,
                           (const float *)grad_bank, (int)C, (int)n_slots, c_tiles);
    else if (dtype == MMFS_F16)
        hipLaunchKernelGGL((bank_scatter_kernel<half_t>), grid, dim3(kBankThreads), 0, st, L, src_index,
                           (const half_t *)grad_bank, (int)C, (int)n_slots, c_tiles);
    else
        hipLaunchKernelGGL((bank_scatter_kernel<bf16_t>), grid, dim3(kBankThreads), 0, st, L, src_index,
                           (const bf16_t *)grad_bank, (int)C, (int)n_slots, c_tiles);
    return (int)hipGetLastError();
}

}  // extern "C"
