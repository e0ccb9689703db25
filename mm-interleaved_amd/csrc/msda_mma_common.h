// msda_mma_common.h -- what the two "LDS-resident levels" kernels share (msda_fwd_mma.hip: forward;
// msda_taps_mma.hip: grad_loc / grad_attn): the small levels of a (batch, head) slab copied into LDS once per
// 1024-lane workgroup and sampled by v_mfma_f32_16x16x32 from there, the large levels by row gather.
//   * geometry of the LDS image (row pitch D*e + 32 bytes, line pitch padded so that a sample's four corners
//     sit in four different 32-byte bank slots), the level table kept in LDS and the rule that decides on
//     the device which levels are resident (smallest first, while they fit);
//   * the image fill (channel-permuted for the forward, whose products' columns must land in the row-gather
//     accumulators; natural order for the taps kernel, whose products contract over the channels);
//   * wave-level synchronisation for wave-private LDS records.
#pragma once
#include "msda_device.h"
#include "msda_env.h"
#include <cstdlib>
#include <algorithm>

namespace mmfs {
namespace mma {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#ifndef MMFS_MMA_WAVES
#define MMFS_MMA_WAVES 16
#endif
constexpr int kMmaWaves = MMFS_MMA_WAVES;     // waves per workgroup (one workgroup per CU: the LDS image is shared)
constexpr int kMmaThreads = kMmaWaves * 64;
constexpr int kMmaMaxLevels = 64;             // level table kept in LDS
constexpr int kChunk = 16;                    // samples of a query staged at a time (one per lane of a 16-lane group)
constexpr int kLdsTotal = 160 * 1024;         // LDS of a CU (MI355X_MICROARCH.md)
constexpr int kTabInts = 6;                   // per level: H, W, start, image base (-1: not resident), line pitch, bytes

template <typename T> struct FwdMma;
template <> struct FwdMma<bf16_t> {
    static __device__ __forceinline__ f32x4 run(const s16x8 &a, const s16x8 &b, const f32x4 &c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    // fp32 weight -> leading 16 bits, rounded remainder (the difference is exact)
    static __device__ __forceinline__ void split(float w, uint32_t &hi, uint32_t &lo) {
        const uint32_t h = __float_as_uint(w) & 0xffff0000u;
        hi = h >> 16;
        lo = (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)(w - __uint_as_float(h)));
    }
    // the same two parts, each in BOTH halves of a word (msda_fwd_wq.hip: the lane's mask picks the half)
    static __device__ __forceinline__ void split_dup(float w, uint32_t &hi2, uint32_t &lo2) {
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        const uint32_t b = __float_as_uint(w);
        hi2 = __builtin_amdgcn_perm(b, b, 0x03020302u);                  // (bytes 2, 3 of w twice)
        const float r = w - __uint_as_float(b & 0xffff0000u);
        bf2 p; p[0] = (__bf16)r; p[1] = (__bf16)r;
        lo2 = __builtin_bit_cast(uint32_t, p);
    }
};
template <> struct FwdMma<half_t> {
    static __device__ __forceinline__ f32x4 run(const s16x8 &a, const s16x8 &b, const f32x4 &c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ void split(float w, uint32_t &hi, uint32_t &lo) {
        const float c = w != w ? w : fminf(fmaxf(w, -65504.f), 65504.f);
        const _Float16 h = (_Float16)c;
        const _Float16 l = (_Float16)(c - (float)h);
        hi = (uint32_t)__builtin_bit_cast(uint16_t, h);
        lo = (uint32_t)__builtin_bit_cast(uint16_t, l);
    }
    static __device__ __forceinline__ void split_dup(float w, uint32_t &hi2, uint32_t &lo2) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const float c = w != w ? w : fminf(fmaxf(w, -65504.f), 65504.f);
        const _Float16 h = (_Float16)c;
        const _Float16 l = (_Float16)(c - (float)h);
        h2 ph; ph[0] = h; ph[1] = h;
        h2 pl; pl[0] = l; pl[1] = l;
        hi2 = __builtin_bit_cast(uint32_t, ph);
        lo2 = __builtin_bit_cast(uint32_t, pl);
    }
};

// Geometry for a head of D channels of 2 bytes; EXTRA: wave-private scratch bytes besides the sample records.
template <int D, int EXTRA = 0> struct MmaGeom {
    static constexpr int RB = D * 2;                  // bytes of a pixel row of one head
    static constexpr int LPI = RB / 16;               // lanes per query in the row-gather layout (16-byte vectors)
    static constexpr int QPW = 64 / LPI;              // queries a wave works on at a time
    static constexpr int NG = D / 16;                 // 16-channel column groups of a row
    static constexpr int RP = RB + 32;                // pitch of a pixel row in the LDS image
    static constexpr int QSTRIDE = (2 * kChunk + 1) * 16;       // bytes between the record rows of two queries (+16: banks)
    static constexpr int REC_BYTES = QPW * QSTRIDE;   // wave-private sample records
    static constexpr int WSCR = REC_BYTES + EXTRA;    // wave-private bytes
    static constexpr int TAB_BYTES = ((kMmaMaxLevels * kTabInts * 4 + 64) + 255) & ~255;
    static constexpr int IMG0 = (TAB_BYTES + kMmaWaves * WSCR + 255) & ~255;    // image offset in the dynamic LDS (256-aligned)
    // position (in halfwords) of channel 8 * lig + i inside the channel-PERMUTED image of a pixel row
    static __device__ __forceinline__ int img_pos(int lig, int i) {
        if (D == 128) return i * 16 + lig;                                        // [g = i][n = lig]
        return i < 4 ? i * 16 + lig : (i - 4) * 16 + lig + 8;                     // D == 64: [g = i % 4][n = lig + 8 * (i / 4)]
    }
};

// bytes between two lines of a W-pixel-wide level: y-neighbours 64 bytes apart modulo the 256-byte bank row
template <int D> __host__ __device__ __forceinline__ int line_pitch(int W)
{
    const int raw = W * MmaGeom<D>::RP;
    return raw + ((64 - raw % 256) & 255);
}

__device__ __forceinline__ void wave_sync()
{
    // same wave writes and reads: LDS keeps program order; the fences keep the compiler from moving
    // the accesses of OTHER lanes' data across
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int CTRL> __device__ __forceinline__ float dpp_move(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// Eight channels (one 16-byte piece, piece_off bytes into the head's row) of one query with the reference's per-channel
// arithmetic (cuh:275-299): fp32 multiply-add of every element, corners outside the map skipped, a sample that fails the
// range test or carries a zero attention weight reads nothing (as in every formulation of this library).
// What the matrix-core forwards redo a result with that came out non-finite: their products carry the weights as
// hi + lo 16-bit parts (Inf x hi + Inf x lo is NaN when lo is zero or negative) and multiply zero weights with the rows
// they are handed (0 x Inf), the reference does neither -- recomputed, the result is the reference's element for element.
// tab: the level table in LDS (kTabInts ints per level: H, W, first pixel, ...).  Not a tuned path.
template <typename T>
__device__ __forceinline__ void exact_lane8(const int *tab, __amdgpu_buffer_rsrc_t rsrc, uint32_t row_bytes,
                                            const uint16_t *loc_q, const uint16_t *attn_q, int K, int P, uint32_t piece_off,
                                            float (&acc)[8])
{
    typedef Vec16<T> V;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int k = 0; k < K; ++k) {
        const int l = k / P;
        const int Hl = tab[kTabInts * l], Wl = tab[kTabInts * l + 1], lstart = tab[kTabInts * l + 2];
        const float lx = to_f32(__builtin_bit_cast(T, loc_q[2 * k])), ly = to_f32(__builtin_bit_cast(T, loc_q[2 * k + 1]));
        const float a = to_f32(__builtin_bit_cast(T, attn_q[k]));
        const Tap<float> t = locate<float>(lx, ly, Hl, Wl, lstart);
        if (a == 0.f) continue;
        const float gy = 1.f - t.fy, gx = 1.f - t.fx;
        const float w[4] = {gy * gx * a, gy * t.fx * a, t.fy * gx * a, t.fy * t.fx * a};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (t.row[c] < 0) continue;
            const uint4 raw = buffer_load16(rsrc, (uint32_t)t.row[c] * row_bytes + piece_off);
            float v[8];
            V::unpack(raw, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fmaf(w[c], v[i], acc[i]);
        }
    }
}

// Workgroups of a launch.  Few runs of queries (< 2 per CU): one workgroup per run, dealt by the dispatcher.  Else the
// kernels are PERSISTENT: one workgroup per CU, runs dealt w, w + n, ... -- n a multiple of H, so that run -> head -> XCD
// stays what the head's slab in the XCD's L2 expects, and at any moment an XCD's workgroups cover 32 consecutive runs of
// its head = two (b, h) slabs, as the dispatcher's deal did.  A CU holds one of these workgroups at a time; relaunching a
// 1024-lane, 160 KB workgroup per run left it idle for a quarter of the kernel (r03c).  Round 3's first measurement of
// this (r03e) had it slower -- every wave waiting at the barrier between two runs for the slowest wave of the previous
// one; on the closing build (image fill level by level, hosted plan, ...) it is faster: north star forward 135.6 ->
// 132.3 us, grad_loc / grad_attn 137.3 -> 131.3, step -1.7 % (r03be / r03bf).  A slab-major deal (a workgroup's runs in
// ONE slab, its image filled once) loses far more than the fills it saves: all B slabs of a head are then in flight in
// an XCD at once, 11 MB against 4 MB of L2 -- forward 183 us (r03bf; = one run of 1024 queries).
// MMFS_MMA_GRID=n: n workgroups whatever the shape (tuning); MMFS_MMA_PERSIST=0: always one workgroup per run.
inline int64_t persistent_grid(int64_t runs, int H)
{
    // (the tests compare the deals inside one process: mmfs_env_reload)
    const int env_grid = knob_int(K_MMA_GRID, 0), env_persist = knob_int(K_MMA_PERSIST, 1);
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        return n;
    }();
    if (env_grid > 0) return std::min<int64_t>(runs, env_grid);
    if (!env_persist || cus <= 0 || H <= 0 || cus % H || runs < 2 * (int64_t)cus) return runs;
    return cus;
}

// Queries per run (= per image fill) of the LDS-resident kernels.  256 where that still leaves two runs per CU (the north star:
// 1024 runs); fewer queries than that per (b, h) slab and the runs are shortened -- 128, then 64 -- so that the launch keeps
// filling the chip: at B * H = 64 slabs and 1024 queries, runs of 256 are ONE workgroup per CU (forward 48.6 us, the row
// gather 44.7), runs of 128 two (39.1 us); at 256 queries runs of 256 left three quarters of the CUs idle (32 us against the
// row gather's 17: r05ac).  enough_runs: even the shortest runs do not give every CU two -- such launches stay on the row gather.
inline int device_cus()
{
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        return n > 0 ? n : 256;
    }();
    return cus;
}
inline int pick_queries_per_run(const Dims &d, int unit, int env_q)
{
    int q = 256;
    if (env_q > 0) q = env_q;
    else {
        const int64_t slabs = (int64_t)d.B * d.H, want = 2 * (int64_t)device_cus();
        while (q > 64 && slabs * ((d.Nq + q - 1) / q) < want) q >>= 1;
    }
    return std::max(unit, (q + unit - 1) / unit * unit);
}
inline bool enough_runs(const Dims &d)
{
    return (int64_t)d.B * d.H * ((d.Nq + 63) / 64) >= 2 * (int64_t)device_cus();
}

// Level table -> LDS, and which levels live in the image: smallest first (ties: lower index), while they fit
// behind the zero row.  Call from every thread of the workgroup; ends with a barrier.
template <int D>
__device__ __forceinline__ void build_level_table(int *tab, unsigned char *img, const int64_t *__restrict__ shapes,
                                                  const int64_t *__restrict__ start, int L, int tid, int img_budget)
{
    constexpr int RP = MmaGeom<D>::RP;
    for (int l = tid; l < L; l += kMmaThreads) {
        const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
        tab[kTabInts * l] = Hl; tab[kTabInts * l + 1] = Wl; tab[kTabInts * l + 2] = (int)start[l];
        int bytes = (Hl > 0 && Wl > 0) ? 1 << 24 : 0;                     // "never fits"; an empty level takes no room
        int lp = 0;
        if (Hl > 0 && Wl > 0 && Hl <= 1024 && Wl <= 1024) {
            lp = line_pitch<D>(Wl);
            const int64_t bb = (int64_t)Hl * lp;
            if (bb < (1 << 24)) bytes = (int)bb;
        }
        tab[kTabInts * l + 4] = lp; tab[kTabInts * l + 5] = bytes;
    }
    __syncthreads();
    for (int l = tid; l < L; l += kMmaThreads) {
        const int px = tab[kTabInts * l] * tab[kTabInts * l + 1], bytes = tab[kTabInts * l + 5];
        int cum = 0;
        for (int l2 = 0; l2 < L; ++l2) {
            const int px2 = tab[kTabInts * l2] * tab[kTabInts * l2 + 1];
            if (px2 < px || (px2 == px && l2 <= l)) cum += tab[kTabInts * l2 + 5];
        }
        // (cum includes this level; the zero row sits in front of the first level)
        tab[kTabInts * l + 3] = (cum + RP <= img_budget && px > 0) ? RP + cum - bytes : -1;
    }
    if (tid < RP / 4) reinterpret_cast<uint32_t *>(img)[tid] = 0u;        // the zero row
    __syncthreads();
}

// Resident levels global -> LDS (once per run of queries).  PERMUTE: channel-permuted (16-bit writes); else natural
// order (one 16-byte write per lane).  Level after level, a lane's pieces of a level four at a time: requested
// together, then written (the plain loop -- request, wait, write, next -- spent 9 k clocks on five dependent trips;
// a version that batched ALL pieces of all levels behind a per-piece level search cost more in index arithmetic
// than it saved: profiles/r03_experiments.md, r03g).  Ends with a barrier.
template <int D, bool PERMUTE>
__device__ __forceinline__ void fill_image(const int *tab, unsigned char *img, __amdgpu_buffer_rsrc_t rsrc,
                                           uint32_t row_bytes, int L, int S, int tid)
{
    typedef MmaGeom<D> G;
    constexpr int NB = 4;
    for (int l = 0; l < L; ++l) {
        const int base = tab[kTabInts * l + 3];
        if (base < 0) continue;
        const int Hl = tab[kTabInts * l], Wl = tab[kTabInts * l + 1], st = tab[kTabInts * l + 2], lp = tab[kTabInts * l + 4];
        const int units = Hl * Wl * G::LPI;
        const float inv_w = 1.0f / (float)Wl;
        for (int u0 = tid; u0 < units; u0 += NB * kMmaThreads) {
            uint4 raw[NB];
            int dst[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int u = u0 + i * kMmaThreads;
                dst[i] = -1;
                raw[i] = make_uint4(0u, 0u, 0u, 0u);
                if (u < units) {
                    const int p = u / G::LPI, lig = u % G::LPI;           // (LPI is a power of two)
                    int y = (int)((float)p * inv_w);                       // p / Wl for p < 2^20: a float guess, corrected
                    y -= (y * Wl > p); y += ((y + 1) * Wl <= p);
                    const int x = p - y * Wl;
                    const uint32_t goff = (uint32_t)(st + p) < (uint32_t)S ? (uint32_t)(st + p) * row_bytes + (uint32_t)lig * 16u : kOobOffset;
                    raw[i] = buffer_load16(rsrc, goff);
                    dst[i] = base + y * lp + x * G::RP + (PERMUTE ? 2 * lig : 16 * lig);
                }
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                if (dst[i] < 0) continue;
                if (PERMUTE) {
                    // channel 8 * lig + j -> halfword img_pos(lig, j): the lane's own column of the row's 32-byte groups
                    uint16_t *row = reinterpret_cast<uint16_t *>(img + dst[i]);
                    const uint32_t w[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        row[G::img_pos(0, j)] = (uint16_t)(w[j >> 1] >> (16 * (j & 1)));
                } else {
                    *reinterpret_cast<uint4 *>(img + dst[i]) = raw[i];
                }
            }
        }
    }
    __syncthreads();
}

}  // namespace mma
}  // namespace mmfs
