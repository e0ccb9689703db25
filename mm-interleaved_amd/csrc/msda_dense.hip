// msda_dense.hip -- the small levels of multi-scale deformable attention as dense matrix
// products on the matrix cores ("hybrid" path).
//
// Why.  Measured on MI355X the row-gather kernels are bound by the
// vector-memory path: every tap corner is one D*sizeof(T)-byte row through L1 (64 B/clk/CU),
// 24x the op's compulsory bytes, and every level receives the same number of taps whatever its
// size.  A level of <= 256 pixels is a tiny dense matrix, though: for a tile of queries
//
//      out[q, :]      +=  A_l[q, :] . V_l               A_l[q, pix] = sum of the bilinear*attention
//      grad_value_l   +=  A_l^T . grad_out                            weights of q's taps on pix
//      dot[q, pix]     =  grad_out[q, :] . V_l[pix, :]  (then grad_attn / grad_loc are 4 look-ups)
//
// are GEMMs with K_l = H_l*W_l <= 256, and v_mfma_f32_32x32x16_{bf16,f16} does 1017 FLOP/clk/SIMD.
// For the 16x16 + 8x8 levels of the north-star shape that is 2560 MFMA clocks per 64 queries
// against 8192 clocks of row reads -- with the matrix pipe otherwise idle.  The big levels stay
// with the gather kernels (LevelSel routing, msda_fwd.hip / msda_bwd.hip / msda_bwd_value.hip).
//
// Precision.  The MFMA takes 16-bit operands.  V and grad_out ARE 16-bit (this path exists for
// f16 / bf16 storage only), products are exact and accumulate in fp32 like the fmaf chains of
// the gather kernels.  The fp32 weights A are split into hi + lo 16-bit halves (hi = leading
// bits, lo = rounded remainder): both halves sit side by side in ONE 32-bit word of the weight
// tile and multiply the same (duplicated) V element, so one MFMA K-slot pair evaluates
// (hi + lo) * v -- 16-17 significant bits of weight, well inside the storage type's rounding.
// The weight tile is built in LDS with plain read-add-write: the 4 lanes of a query own the
// pixels with (pixel & 3) == lane, so no two lanes ever touch one word (LDS float atomics run
// at 0.33 lane-adds/clk/CU).
//
// Semantics that differ from the gather kernels, by construction of a dense product: a
// non-finite value / grad_out element in a dense level reaches every query of its (b, h) as
// 0 * Inf = NaN instead of only the queries that sample it.  (Non-finite *locations* are
// handled identically: they produce zero weights.)
//
// The reference has no counterpart (its kernels are one thread per output scalar,
// ms_deform_im2col_cuda.cuh:240-923); results are checked against the same oracle.
#include "msda_device.h"
#include "msda_launch.h"
#include <cstdlib>

namespace mmfs {

namespace {

constexpr int kDT = 256;                         // 4 waves
constexpr int kBT = 1024;                        // 16 waves: the workgroups that own a CU
constexpr int kTileQ = 64;                       // queries per tile
constexpr int kAStride = kCoarseMaxPx + 4;       // words per query row ([q][pixel] tiles)
constexpr int kQStride = kTileQ + 4;             // words per pixel row ([pixel][q] tile)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <typename T> struct Mma;

template <> struct Mma<bf16_t> {
    static __device__ __forceinline__ f32x16 run(const uint4 &a, const uint4 &b, const f32x16 &c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                       __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    // fp32 weight -> {hi (low half), lo (high half)}, hi + lo == a to ~2^-17 relative
    static __device__ __forceinline__ uint32_t split(float a) {
        uint32_t hi = __float_as_uint(a) & 0xffff0000u;
        if (a != a) hi = 0x7fc00000u;                                  // keep NaN a NaN
        const bool inf = (__float_as_uint(a) & 0x7fffffffu) == 0x7f800000u;
        const float r = inf ? 0.f : a - __uint_as_float(hi);          // exact
        const __bf16 lo = (__bf16)r;                                   // RNE
        return (hi >> 16) | ((uint32_t)__builtin_bit_cast(uint16_t, lo) << 16);
    }
};

template <> struct Mma<half_t> {
    static __device__ __forceinline__ f32x16 run(const uint4 &a, const uint4 &b, const f32x16 &c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a),
                                                      __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t split(float a) {
        // weights beyond the f16 range saturate (they would need |attention| > 65504)
        const float c = a != a ? a : fminf(fmaxf(a, -65504.f), 65504.f);
        const _Float16 hi = (_Float16)c;
        const _Float16 lo = (_Float16)(c - (float)hi);
        return (uint32_t)__builtin_bit_cast(uint16_t, hi) | ((uint32_t)__builtin_bit_cast(uint16_t, lo) << 16);
    }
};

// C/D layout of the 32x32 MFMAs: column = lane & 31, row = this (cdna_hip_programming.md section 3)
__device__ __forceinline__ int mfma_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// Development aid (tools/exp_build.sh dense_prof "-DMMFS_PROFILE_DENSE"): shader clocks per phase,
// summed over workgroups (thread 0 of each), read back with mmfs_debug_dense_profile().
#ifdef MMFS_PROFILE_DENSE
}  // namespace
__device__ unsigned long long g_dense_prof[32];
namespace {
#define PROF_DECL unsigned long long prof_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long prof_c = __builtin_readcyclecounter()
#define PROF(i) do { const unsigned long long prof_n = __builtin_readcyclecounter(); prof_t[i] += prof_n - prof_c; prof_c = prof_n; } while (0)
#define PROF_END(base) do { if (threadIdx.x == 0) for (int i = 0; i < 8; ++i) atomicAdd(&g_dense_prof[(base) + i], prof_t[i]); } while (0)
#else
#define PROF_DECL do {} while (0)
#define PROF(i) do {} while (0)
#define PROF_END(base) do {} while (0)
#endif

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ f32x16 zero16()
{
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// The P samples (locations + attention weights) of one (query, level), requested early so the
// memory latency hides behind the previous phase.  NV = P / 4 (16-byte location vectors, 8-byte
// weight vectors; P = 4, 8); NV = 0: any P, read when needed.
template <typename T, int NV>
struct Samples {
    uint4 l[NV > 0 ? NV : 1];
    uint2 a[NV > 0 ? NV : 1];
    __device__ __forceinline__ void request(const T *__restrict__ loc, const T *__restrict__ attn, int64_t s0) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            l[v] = reinterpret_cast<const uint4 *>(loc + 2 * s0)[v];
            a[v] = reinterpret_cast<const uint2 *>(attn + s0)[v];
        }
    }
};

// The weight tile of one level for the 64 queries q0..q0+63 of (b, h), accumulated in fp32 with
// plain read-add-write.  Thread t serves query t/4 and only the pixels with (pixel & 3) == t%4.
//   BY_PIXEL = false: A[q][pixel]  (row stride kAStride)      -- forward
//   BY_PIXEL = true : A[pixel][q]  (row stride kQStride)      -- grad_value
template <bool BY_PIXEL>
__device__ __forceinline__ void add_sample(float *__restrict__ A, float lx, float ly, float a, int Hl, int Wl,
                                           int qi, int j)
{
    const Tap<float> t = locate<float>(lx, ly, Hl, Wl, 0);
    const float gy = 1.f - t.fy, gx = 1.f - t.fx;
    const float w[4] = {gy * gx * a, gy * t.fx * a, t.fy * gx * a, t.fy * t.fx * a};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (t.row[c] >= 0 && (t.row[c] & 3) == j) {
            float *e = BY_PIXEL ? A + t.row[c] * kQStride + qi : A + qi * kAStride + t.row[c];
            *e += w[c];
        }
    }
}

template <typename T, int NV, bool BY_PIXEL>
__device__ __forceinline__ void build_weight_tile(float *__restrict__ A, const Samples<T, NV> &sm,
                                                  const T *__restrict__ loc, const T *__restrict__ attn,
                                                  const Dims &d, int b, int h, int q0, int level, int Hl, int Wl)
{
    const int tid = threadIdx.x, qi = tid >> 2, j = tid & 3;
    const int q = q0 + qi;
    if (q >= d.Nq) return;
    if (NV > 0) {
        typedef Vec16<T> V;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            float l[8], a[8];
            V::unpack(sm.l[v], l);
            V::unpack(make_uint4(sm.a[v].x, sm.a[v].y, 0u, 0u), a);      // first 4 valid
#pragma unroll
            for (int i = 0; i < 4; ++i) add_sample<BY_PIXEL>(A, l[2 * i], l[2 * i + 1], a[i], Hl, Wl, qi, j);
        }
    } else {
        const int64_t s0 = ((((int64_t)b * d.Nq + q) * d.H + h) * d.L + level) * d.P;
        for (int p = 0; p < d.P; ++p)
            add_sample<BY_PIXEL>(A, to_f32(loc[2 * (s0 + p)]), to_f32(loc[2 * (s0 + p) + 1]), to_f32(attn[s0 + p]),
                                 Hl, Wl, qi, j);
    }
}

// rows x cols4 16-byte vectors of a tile: fill with zeros / split every fp32 weight in place
template <int STRIDE4>
__device__ __forceinline__ void tile_zero(float *A, int rows, int cols4)
{
    uint4 *v = reinterpret_cast<uint4 *>(A);
    for (int i = threadIdx.x; i < rows * cols4; i += kDT) {
        const int r = i / cols4, c = i - r * cols4;
        v[r * STRIDE4 + c] = make_uint4(0u, 0u, 0u, 0u);
    }
}

template <typename T, int STRIDE4>
__device__ __forceinline__ void tile_split(float *A, int rows, int cols4)
{
    uint4 *v = reinterpret_cast<uint4 *>(A);
    for (int i = threadIdx.x; i < rows * cols4; i += kDT) {
        const int r = i / cols4, c = i - r * cols4;
        uint4 x = v[r * STRIDE4 + c];
        x.x = Mma<T>::split(__uint_as_float(x.x)); x.y = Mma<T>::split(__uint_as_float(x.y));
        x.z = Mma<T>::split(__uint_as_float(x.z)); x.w = Mma<T>::split(__uint_as_float(x.w));
        v[r * STRIDE4 + c] = x;
    }
}

// 4 consecutive 16-bit elements -> the MFMA operand that meets the {hi, lo} weight words: every
// element twice
__device__ __forceinline__ uint4 dup4(uint32_t lo, uint32_t hi)
{
    return make_uint4(__builtin_amdgcn_perm(lo, lo, 0x01000100u), __builtin_amdgcn_perm(lo, lo, 0x03020302u),
                      __builtin_amdgcn_perm(hi, hi, 0x01000100u), __builtin_amdgcn_perm(hi, hi, 0x03020302u));
}

// ---------------------------------------------------------------- value of the dense levels, packed
// vt[(b*H + h)][g][ch][4]: the dense levels' pixels of one (b, h), levels back to back (each padded
// with zero pixels to a multiple of 8), in groups of 4 pixels per channel -- the 8 bytes one lane
// of the forward's B operand needs.
template <typename T>
__global__ void __launch_bounds__(256)
coarse_pack_kernel(const T *__restrict__ value, T *__restrict__ vt, const Dims d, const CoarsePlan cp)
{
    __shared__ CoarseLevel lv[kMaxCoarse];
    if ((int)threadIdx.x < cp.n) lv[threadIdx.x] = cp.lv[threadIdx.x];
    __syncthreads();
    const int groups = cp.ktot / 4;
    const int64_t total = (int64_t)d.B * d.H * groups * d.D;
    const uint16_t *src = reinterpret_cast<const uint16_t *>(value);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ch = (int)(i % d.D);
        const int64_t r = i / d.D;
        const int g = (int)(r % groups);
        const int64_t bh = r / groups;
        const int b = (int)(bh / d.H), h = (int)(bh % d.H);
        const int kk = g * 4;
        int ci = 0;
        while (ci + 1 < cp.n && kk >= lv[ci + 1].coff) ++ci;
        const int pix0 = kk - lv[ci].coff, px = lv[ci].Hl * lv[ci].Wl;
        uint32_t v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pix = pix0 + j;
            v[j] = pix < px ? (uint32_t)src[(((int64_t)b * d.S + lv[ci].start + pix) * d.H + h) * d.D + ch] : 0u;
        }
        reinterpret_cast<uint2 *>(vt)[i] = make_uint2(v[0] | (v[1] << 16), v[2] | (v[3] << 16));
    }
}

// ---------------------------------------------------------------- forward, dense levels
// One workgroup = 64 queries of one (b, h).  Per dense level: zero the weight tile, build it,
// split it, then every wave multiplies it with its 32-channel slice of the packed value.
// NS = D / 32 channel slices; the 2 * NS (32-query block, slice) jobs are dealt to the 4 waves.
// Everything that comes from global memory is requested a phase ahead: the next level's samples
// before this level is built, this level's value fragments before its tile is zeroed.
// Result: cinit[b, q, h, :] (fp32), the starting value of the gather kernel's accumulators.
template <typename T, int NS, int NV>
__global__ void __launch_bounds__(kDT, 2)
msda_fwd_coarse(const T *__restrict__ vt, const T *__restrict__ loc, const T *__restrict__ attn,
                float *__restrict__ cinit, const Dims d, const CoarsePlan cp)
{
    constexpr int JOBS = (2 * NS + 3) / 4;
    constexpr int PF = 8;                           // pixel octets requested ahead (one group of 64 pixels)
    __shared__ __attribute__((aligned(16))) float A[kTileQ * kAStride];

    const BlockCoord bc = block_coord(d, kTileQ);
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);            // provably wave-uniform
    const int ns = wave % NS;
    const int64_t bh = (int64_t)bc.b * d.H + bc.h;
    const int64_t sq = (((int64_t)bc.b * d.Nq + min(bc.q0 + (tid >> 2), d.Nq - 1)) * d.H + bc.h) * d.L;

    f32x16 acc[JOBS];
#pragma unroll
    for (int j = 0; j < JOBS; ++j) acc[j] = zero16();

    Samples<T, NV> nxt;
    nxt.request(loc, attn, (sq + uni(cp.lv[0].level)) * d.P);
    for (int ci = 0; ci < cp.n; ++ci) {
        const int level = uni(cp.lv[ci].level), Hl = uni(cp.lv[ci].Hl), Wl = uni(cp.lv[ci].Wl);
        const int coff = uni(cp.lv[ci].coff), K = uni(cp.lv[ci].kpad);
        // the product runs over whole groups of 64 pixels: the tile's extra columns are zero and
        // meet (finite) elements of the level's last pixels
        const int ngrp = (K + 63) / 64, kz = ngrp * 64, last_g4 = K / 4 - 1;
        const Samples<T, NV> cur = nxt;
        if (ci + 1 < cp.n) nxt.request(loc, attn, (sq + uni(cp.lv[ci + 1].level)) * d.P);
        const uint2 *bsrc = reinterpret_cast<const uint2 *>(vt) + (bh * cp.ktot + coff) / 4 * d.D + ns * 32 + l32;
        uint2 x[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) x[u] = bsrc[(int64_t)min(u * 2 + kg, last_g4) * d.D];

        if (ci > 0) __syncthreads();                // the previous level's fragments are all read
        tile_zero<kAStride / 4>(A, kTileQ, kz / 4);
        __syncthreads();
        build_weight_tile<T, NV, false>(A, cur, loc, attn, d, bc.b, bc.h, bc.q0, level, Hl, Wl);
        __syncthreads();
        tile_split<T, kAStride / 4>(A, kTileQ, kz / 4);
        __syncthreads();

        for (int g = 0; g < ngrp; ++g) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int kb = g * PF + u;
                const uint4 bfrag = dup4(x[u].x, x[u].y);
                x[u] = bsrc[(int64_t)min((kb + PF) * 2 + kg, last_g4) * d.D];
#pragma unroll
                for (int j = 0; j < JOBS; ++j) {
                    const int jj = wave + 4 * j;
                    if (4 * JOBS <= 2 * NS || jj < 2 * NS) {           // constant-true unless NS == 1
                        const int mb = jj / NS;
                        const uint4 afrag = *reinterpret_cast<const uint4 *>(
                            &A[(mb * 32 + l32) * kAStride + kb * 8 + kg * 4]);
                        acc[j] = Mma<T>::run(afrag, bfrag, acc[j]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < JOBS; ++j) {
        const int jj = wave + 4 * j;
        if (jj < 2 * NS) {
            const int mb = jj / NS;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = bc.q0 + mb * 32 + mfma_row(r, lane);
                if (q < d.Nq)
                    cinit[(((int64_t)bc.b * d.Nq + q) * d.H + bc.h) * d.D + ns * 32 + l32] = acc[j][r];
            }
        }
    }
}

// ---------------------------------------------------------------- grad_loc / grad_attn, dense levels
// dot[q, pix] = grad_out[q, :] . value[pix, :] for 64 queries x all pixels of a dense level by
// MFMA, parked in LDS; then one thread per (query, point) looks its four corners up and finishes
// the per-sample algebra of msda_bwd_vec (cuh:119-161).
// One 1024-lane workgroup per CU owns a run of query tiles of one (b, h) and walks level by level,
// tile by tile.  16 waves = 2 query blocks x 8 pixel blocks, one 32x32 product per step each:
//   * a wave's value fragments (its 32 pixels x D channels) do not change along the tiles of a
//     level: loaded once per level, straight from global memory;
//   * the tile's grad_out rows are read coalesced (one 16-byte vector per thread, requested a step
//     ahead) and go through a double-buffered LDS tile into the MFMA operand layout.  Fetching
//     MFMA operands lane-by-lane from global memory (every lane its own row) ran at ~1 lane/clk
//     through the vector-memory path: 154 us for this kernel instead of ~40;
//   * the look-up threads' samples are requested a step ahead as well.
// MB = 32-query blocks per tile (workgroup = MB * 512 lanes)
template <typename T, int NS, int MB>
__global__ void __launch_bounds__(MB * 512)
msda_taps_coarse(const T *__restrict__ value, const T *__restrict__ loc, const T *__restrict__ attn,
                 const T *__restrict__ grad_out, T *__restrict__ grad_loc, T *__restrict__ grad_attn,
                 const Dims d, const DotPlan cp, const int chunks, const int tiles_per_chunk)
{
    constexpr int KB = 2 * NS;                      // 16-channel steps
    constexpr int VPR = 4 * NS;                     // 16-byte vectors per grad_out row (D / 8)
    constexpr int GS = VPR + 1;                     // row stride of the staged tile, in vectors
    constexpr int TQ = 32 * MB;                     // queries per tile
    __shared__ __attribute__((aligned(16))) float G[2][TQ * kAStride];     // products of this / the previous step
    __shared__ uint4 gtile[TQ * GS];

    int bid = blockIdx.x;
    const int h = bid % d.H; bid /= d.H;
    const int chunk = bid % chunks;
    const int b = bid / chunks;
    const int q_tiles = (d.Nq + TQ - 1) / TQ;
    const int t_begin = chunk * tiles_per_chunk, t_end = min(q_tiles, t_begin + tiles_per_chunk);
    if (t_begin >= t_end) return;

    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, kg = lane >> 5;
    const int wave = uni(tid >> 6);
    const int mb = wave % MB, nbw = wave / MB;       // this wave's 32 queries / 32 pixels
    const int64_t HD = (int64_t)d.H * d.D;
    const uint32_t *loc2 = reinterpret_cast<const uint32_t *>(loc);       // (x, y) pairs of 16-bit scalars
    const uint16_t *loc1 = reinterpret_cast<const uint16_t *>(loc);
    const uint16_t *attn1 = reinterpret_cast<const uint16_t *>(attn);
    const bool pair_ok = (reinterpret_cast<uintptr_t>(loc) & 3) == 0;
    const int items = TQ * d.P;                 // look-ups per step; thread i < items does one (P <= 16)
    const int iq = tid / d.P, ip = tid - iq * d.P;
    const int vq = tid / VPR, vc = tid - vq * VPR;  // staging: (row, vector) of the grad_out tile
    const bool stager = tid < TQ * VPR;

    auto go_vector = [&](int t) {
        const int q = min(t * TQ + vq, d.Nq - 1);
        return *reinterpret_cast<const uint4 *>(grad_out + (((int64_t)b * d.Nq + q) * d.H + h) * d.D + vc * 8);
    };
    auto sample_index = [&](int t, int level) {
        const int q = min(t * TQ + iq, d.Nq - 1);
        return ((((int64_t)b * d.Nq + q) * d.H + h) * d.L + level) * d.P + ip;
    };
    auto sample_load = [&](int t, int level, uint32_t &xy, uint32_t &a) {
        if (tid < items) {
            const int64_t s = sample_index(t, level);
            if (pair_ok) xy = loc2[s];
            else xy = (uint32_t)loc1[2 * s] | ((uint32_t)loc1[2 * s + 1] << 16);
            a = attn1[s];
        }
    };

    // One thread per (query, point) of a step's tile: four look-ups in the step's products and the
    // per-sample algebra.  A step's look-ups run during the NEXT step, between the issue of that
    // step's MFMAs and the parking of their results, so the matrix pipe and the VALU / LDS work of
    // the look-ups overlap (serialised, the look-ups and their barrier were 2/3 of a step).
    struct Pending { int t, level, Hl, Wl, own0, own1, pix0; };
    auto lookups = [&](const Pending &p, uint32_t xy, uint32_t aw, const float *__restrict__ Gp) {
        if (tid < items && p.t * TQ + iq < d.Nq) {
            const int64_t s = sample_index(p.t, p.level);
            float l[Vec16<T>::N];
            Vec16<T>::unpack(make_uint4(xy, aw, 0u, 0u), l);               // {x, y, a, -}
            const float a = l[2];
            const Tap<float> tp = locate<float>(l[0], l[1], p.Hl, p.Wl, 0);
            // the chunk that owns the sample's top row finishes it (same arithmetic as locate)
            const float yy = l[1] * (float)p.Hl - 0.5f, xx = l[0] * (float)p.Wl - 0.5f;
            const bool inside = (yy > -1.f) && (xx > -1.f) && (yy < (float)p.Hl) && (xx < (float)p.Wl);
            const int y0 = inside ? (int)floorf(yy) : -1;
            if (y0 >= p.own0 && y0 < p.own1) {
                const float *g = Gp + iq * kAStride - p.pix0;
                float dot[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) dot[c] = tp.row[c] >= 0 ? g[tp.row[c]] : 0.f;
                const float fx = tp.fx, fy = tp.fy, gy = 1.f - fy, gx = 1.f - fx;
                const float w[4] = {gy * gx, gy * fx, fy * gx, fy * fx};
                const float ga = w[0] * dot[0] + w[1] * dot[1] + w[2] * dot[2] + w[3] * dot[3];
                const float dw = gy * (dot[1] - dot[0]) + fy * (dot[3] - dot[2]);
                const float dh = gx * (dot[2] - dot[0]) + fx * (dot[3] - dot[1]);
                grad_attn[s] = (T)ga;
                grad_loc[2 * s] = (T)((float)p.Wl * dw * a);
                grad_loc[2 * s + 1] = (T)((float)p.Hl * dh * a);
            }
        }
    };

    PROF_DECL;
    uint32_t nxy = 0, na = 0, cxy = 0, ca = 0, pxy = 0, pa = 0;     // samples: being fetched / this step's / pending
    if (stager) gtile[vq * GS + vc] = go_vector(t_begin);
    sample_load(t_begin, uni(cp.c[0].level), nxy, na);
    __syncthreads();
    int buf = 0;
    bool have_pending = false;
    Pending pend = {0, 0, 1, 1, 0, 0, 0};
    for (int ci = 0; ci < cp.n; ++ci) {
        const int level = uni(cp.c[ci].level), Hl = uni(cp.c[ci].Hl), Wl = uni(cp.c[ci].Wl);
        const int row0 = uni(cp.c[ci].row0), own0 = uni(cp.c[ci].own0), own1 = uni(cp.c[ci].own1);
        const int px = uni(cp.c[ci].nrows) * Wl, NB = (px + 31) / 32;      // this chunk's pixels
        const int first = uni(cp.c[ci].start) + row0 * Wl, pix0 = row0 * Wl;
        const int next_level = uni(cp.c[min(ci + 1, cp.n - 1)].level);
        const bool prod = nbw < NB;                 // this wave has a pixel block in this chunk
        uint4 bf[KB];
        if (prod) {
            const int pix = min(nbw * 32 + l32, px - 1);
            const T *vp = value + ((int64_t)b * d.S + first + pix) * HD + (int64_t)h * d.D + kg * 8;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) bf[kb] = *reinterpret_cast<const uint4 *>(vp + kb * 16);
        }
        for (int t = t_begin; t < t_end; ++t) {
            const bool last_tile = t + 1 == t_end;
            const bool more = !(last_tile && ci + 1 == cp.n);
            const int nt = last_tile ? t_begin : t + 1;
            cxy = nxy; ca = na;
            uint4 nv = make_uint4(0u, 0u, 0u, 0u);
            if (more) {
                if (stager) nv = go_vector(nt);
                sample_load(nt, last_tile ? next_level : level, nxy, na);
            }
            f32x16 acc = zero16();
            if (prod) {
                const uint4 *arow = &gtile[(mb * 32 + l32) * GS + kg];
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) acc = Mma<T>::run(arow[kb * 2], bf[kb], acc);
            }
            PROF(0);
            if (have_pending) lookups(pend, pxy, pa, G[buf ^ 1]);
            PROF(1);
            if (prod) {
#pragma unroll
                for (int r = 0; r < 16; ++r) G[buf][(mb * 32 + mfma_row(r, lane)) * kAStride + nbw * 32 + l32] = acc[r];
            }
            PROF(2);
            __syncthreads();                        // tile read, products parked, previous look-ups done
            if (more && stager) gtile[vq * GS + vc] = nv;
            pend.t = t; pend.level = level; pend.Hl = Hl; pend.Wl = Wl; pend.own0 = own0; pend.own1 = own1; pend.pix0 = pix0;
            pxy = cxy; pa = ca;
            have_pending = true;
            buf ^= 1;
            PROF(3);
            __syncthreads();                        // next tile staged
            PROF(4);
        }
    }
    if (have_pending) lookups(pend, pxy, pa, G[buf ^ 1]);
    PROF_END(0);
}

// ---------------------------------------------------------------- grad_value, dense levels
// One 1024-lane workgroup per CU = one (b, h, chunk of query tiles), all dense levels in turn.
// A level's whole grad_value [pixels x D] lives in MFMA accumulators across the chunk.  Per tile:
//   stage   the tile's grad_out rows, read coalesced a step ahead, are stored TRANSPOSED in LDS
//           ([channel][query]) so that a lane's B operand (4 queries of its channel) is one 8-byte read;
//   build   256 threads turn one sample each (requested a step ahead) into a {4 pixels, 4 weights}
//           record, then all 1024 apply the records to the pixel-major fp32 weight tile: the 16
//           lanes of a query own the pixels with (pixel & 15) == lane -- plain read-add-write, no
//           two lanes on one word (LDS float atomics: 0.33 lane-adds/clk/CU);
//   split   every row becomes [hi of 64 queries | lo of 64 queries], in place (read all, barrier,
//           write): the MFMA K-slots are {hi q0..q3, lo q0..q3} against {g q0..q3, g q0..q3};
//   product 16 waves = channel slices x pixel blocks.
// Partial sums per chunk go to the workspace; coarse_value_epilogue adds the chunks and stores the
// rows in the storage type.
struct alignas(16) SampleRec { int pix[4]; float w[4]; };

template <typename T> __device__ __forceinline__ void split_hi_lo(float a, uint32_t &hi, uint32_t &lo)
{
    const uint32_t w = Mma<T>::split(a);
    hi = w & 0xffffu; lo = w >> 16;
}

template <typename T, int NS>
__global__ void __launch_bounds__(kBT)
msda_value_coarse(const T *__restrict__ loc, const T *__restrict__ attn, const T *__restrict__ grad_out,
                  float *__restrict__ partial, const Dims d, const CoarsePlan cp, const int chunks,
                  const int tiles_per_chunk)
{
    constexpr int PBSTEP = 16 / NS;                 // pixel blocks between a wave's jobs
    constexpr int NJ = NS >= 4 ? NS / 2 : 1;        // jobs per wave (8 pixel blocks * NS slices / 16 waves)
    constexpr int QB = kTileQ / 8;                  // query octets per tile
    constexpr int RS = 66;                          // words per pixel row of the weight tile (8-byte accesses: no bank conflicts)
    constexpr int VPR = 4 * NS;                     // 16-byte vectors per grad_out row
    constexpr int GTS = kTileQ + 4;                 // halfwords per channel row of the transposed grad_out tile
    constexpr int MAXPASS = 4;                      // P <= 16
    __shared__ __attribute__((aligned(16))) float At[kCoarseMaxPx * RS];
    __shared__ SampleRec rec[kTileQ * 4];
    __shared__ __attribute__((aligned(16))) uint16_t gT[32 * NS * GTS];

    int bid = blockIdx.x;
    const int h = bid % d.H; bid /= d.H;
    const int chunk = bid % chunks;
    const int b = bid / chunks;
    const int q_tiles = (d.Nq + kTileQ - 1) / kTileQ;
    const int t_begin = chunk * tiles_per_chunk, t_end = min(q_tiles, t_begin + tiles_per_chunk);
    if (t_begin >= t_end) return;

    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, kg = lane >> 5;
    const int wave = uni(tid >> 6);
    const int ns = wave % NS, pb0 = wave / NS;
    const uint16_t *loc1 = reinterpret_cast<const uint16_t *>(loc);
    const uint16_t *attn1 = reinterpret_cast<const uint16_t *>(attn);
    const int sq = tid >> 2, sp = tid & 3;          // record makers: threads 0..255 -> (query, point of the pass)
    const int aq = tid >> 4, aj = tid & 15;         // record appliers: (query, owned pixel class)
    const int vq = tid / VPR, vc = tid - vq * VPR;  // stagers: (row, vector) of the grad_out tile
    const bool stager = tid < kTileQ * VPR, maker = tid < kTileQ * 4;
    const int passes = (d.P + 3) / 4;

    auto go_vector = [&](int t) {
        const int q = t * kTileQ + vq;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);       // rows past the last query contribute nothing
        if (stager && q < d.Nq)
            v = *reinterpret_cast<const uint4 *>(grad_out + (((int64_t)b * d.Nq + q) * d.H + h) * d.D + vc * 8);
        return v;
    };
    auto sample_load = [&](int t, int level, uint32_t (&xy)[MAXPASS], uint32_t (&a)[MAXPASS]) {
        if (maker) {
            const int q = min(t * kTileQ + sq, d.Nq - 1);
#pragma unroll
            for (int pass = 0; pass < MAXPASS; ++pass) {
                if (pass < passes) {
                    const int64_t s = ((((int64_t)b * d.Nq + q) * d.H + h) * d.L + level) * d.P + min(pass * 4 + sp, d.P - 1);
                    xy[pass] = (uint32_t)loc1[2 * s] | ((uint32_t)loc1[2 * s + 1] << 16);
                    a[pass] = attn1[s];
                }
            }
        }
    };

    uint4 nv = go_vector(t_begin);
    uint32_t nxy[MAXPASS] = {0, 0, 0, 0}, na[MAXPASS] = {0, 0, 0, 0};
    sample_load(t_begin, uni(cp.lv[0].level), nxy, na);

    for (int ci = 0; ci < cp.n; ++ci) {
        const int level = uni(cp.lv[ci].level), Hl = uni(cp.lv[ci].Hl), Wl = uni(cp.lv[ci].Wl);
        const int coff = uni(cp.lv[ci].coff), kpad = uni(cp.lv[ci].kpad);
        const int next_level = uni(cp.lv[min(ci + 1, cp.n - 1)].level);
        const int PB = (Hl * Wl + 31) / 32;
        const int rows = min(kCoarseMaxPx, (PB + PBSTEP - 1) / PBSTEP * PBSTEP * 32);   // rows some job reads
        const bool mm = pb0 < PB;                   // this wave has a pixel block in this level

        f32x16 acc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = zero16();

        for (int t = t_begin; t < t_end; ++t) {
            const int q0 = t * kTileQ;
            const bool last_tile = t + 1 == t_end;
            const bool more = !(last_tile && ci + 1 == cp.n);
            const int nt = last_tile ? t_begin : t + 1;
            PROF_DECL;
            const uint4 cv = nv;
            uint32_t sxy[MAXPASS], sa[MAXPASS];
#pragma unroll
            for (int i = 0; i < MAXPASS; ++i) { sxy[i] = nxy[i]; sa[i] = na[i]; }
            __syncthreads();                        // the previous tile's fragments are all read
            if (more) {                             // next step's global reads: a whole step to arrive
                nv = go_vector(nt);
                sample_load(nt, last_tile ? next_level : level, nxy, na);
            }
            PROF(0);
            {   // zero the weight tile; stage the grad_out tile transposed
                uint2 *v2 = reinterpret_cast<uint2 *>(At);
                for (int r = tid >> 5; r < rows; r += kBT / 32) v2[r * (RS / 2) + (tid & 31)] = make_uint2(0u, 0u);
                if (stager) {
                    const uint32_t w[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        gT[(vc * 8 + 2 * i) * GTS + vq] = (uint16_t)(w[i] & 0xffffu);
                        gT[(vc * 8 + 2 * i + 1) * GTS + vq] = (uint16_t)(w[i] >> 16);
                    }
                }
            }
            __syncthreads();
            PROF(1);
#pragma unroll
            for (int pass = 0; pass < MAXPASS; ++pass) {
                if (pass < passes) {
                    if (maker) {
                        SampleRec rc;
                        rc.pix[0] = rc.pix[1] = rc.pix[2] = rc.pix[3] = -1;
                        rc.w[0] = rc.w[1] = rc.w[2] = rc.w[3] = 0.f;
                        if (q0 + sq < d.Nq && pass * 4 + sp < d.P) {
                            float l[Vec16<T>::N];
                            Vec16<T>::unpack(make_uint4(sxy[pass], sa[pass], 0u, 0u), l);     // {x, y, a, -}
                            const Tap<float> tp = locate<float>(l[0], l[1], Hl, Wl, 0);
                            const float gy = 1.f - tp.fy, gx = 1.f - tp.fx, a = l[2];
                            rc.pix[0] = tp.row[0]; rc.pix[1] = tp.row[1]; rc.pix[2] = tp.row[2]; rc.pix[3] = tp.row[3];
                            rc.w[0] = gy * gx * a; rc.w[1] = gy * tp.fx * a; rc.w[2] = tp.fy * gx * a; rc.w[3] = tp.fy * tp.fx * a;
                        }
                        rec[tid] = rc;
                    }
                    __syncthreads();
                    PROF(2);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const SampleRec rc = rec[aq * 4 + k];
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (rc.pix[c] >= 0 && (rc.pix[c] & 15) == aj) At[rc.pix[c] * RS + aq] += rc.w[c];
                    }
                    __syncthreads();
                    PROF(3);
                }
            }
            {   // split: fp32 row -> [hi x 64 | lo x 64], in place (every thread reads its words first)
                uint2 *v2 = reinterpret_cast<uint2 *>(At);
                const int c = tid & 15;             // 4 queries
                uint2 hi[4], lo[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int r = (tid >> 4) + it * (kBT / 16);
                    hi[it] = lo[it] = make_uint2(0u, 0u);
                    if (r < rows) {
                        const uint2 x0 = v2[r * (RS / 2) + 2 * c], x1 = v2[r * (RS / 2) + 2 * c + 1];
                        uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
                        split_hi_lo<T>(__uint_as_float(x0.x), h0, l0); split_hi_lo<T>(__uint_as_float(x0.y), h1, l1);
                        split_hi_lo<T>(__uint_as_float(x1.x), h2, l2); split_hi_lo<T>(__uint_as_float(x1.y), h3, l3);
                        hi[it] = make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));
                        lo[it] = make_uint2(l0 | (l1 << 16), l2 | (l3 << 16));
                    }
                }
                __syncthreads();
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int r = (tid >> 4) + it * (kBT / 16);
                    if (r < rows) {
                        v2[r * (RS / 2) + c] = hi[it];
                        v2[r * (RS / 2) + 16 + c] = lo[it];
                    }
                }
            }
            __syncthreads();
            PROF(4);
            // (waves without a pixel block in this level multiply zero / foreign rows: never stored)
            {
                const uint2 *v2 = reinterpret_cast<const uint2 *>(At);
                const uint2 *g2 = reinterpret_cast<const uint2 *>(gT + (ns * 32 + l32) * GTS);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    const uint2 g = g2[qb * 2 + kg];
                    const uint4 bfrag = make_uint4(g.x, g.y, g.x, g.y);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int pb = min(pb0 + j * PBSTEP, kCoarseMaxPx / 32 - 1);
                        const uint2 ah = v2[(pb * 32 + l32) * (RS / 2) + qb * 2 + kg];
                        const uint2 al = v2[(pb * 32 + l32) * (RS / 2) + 16 + qb * 2 + kg];
                        acc[j] = Mma<T>::run(make_uint4(ah.x, ah.y, al.x, al.y), bfrag, acc[j]);
                    }
                    __builtin_amdgcn_sched_barrier(0);      // keep the fragment reads next to their products
                }
            }
            PROF(5);
            PROF_END(8 + (ci == 0 ? 0 : 8));
        }
        if (mm) {
            // through a buffer descriptor over the level's kpad rows of this chunk: a lane offset +
            // a uniform offset per element, and rows past kpad are dropped by the hardware
            const __amdgpu_buffer_rsrc_t prs = make_slab_rsrc(
                partial + ((((int64_t)chunk * d.B + b) * d.H + h) * cp.ktot + coff) * d.D, (int64_t)kpad * d.D * 4);
            const uint32_t prow = (uint32_t)d.D * 4u;
            const uint32_t plane = (uint32_t)(pb0 * 32 + 4 * kg) * prow + (uint32_t)(ns * 32 + l32) * 4u;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[j][r]), prs, (int)plane,
                                                          (int)((uint32_t)(j * PBSTEP * 32 + (r & 3) + 8 * (r >> 2)) * prow), 0);
            }
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
coarse_value_epilogue(const float *__restrict__ partial, T *__restrict__ grad_value, const Dims d,
                      const CoarsePlan cp, const int chunks)
{
    __shared__ CoarseLevel lv[kMaxCoarse];
    if ((int)threadIdx.x < cp.n) lv[threadIdx.x] = cp.lv[threadIdx.x];
    __syncthreads();
    const int d4 = d.D / 4;
    const int64_t total = (int64_t)d.B * d.H * cp.ktot * d4;
    const int64_t chunk_stride = (int64_t)d.B * d.H * cp.ktot * d.D;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % d4);
        const int64_t r = i / d4;
        const int kk = (int)(r % cp.ktot);
        const int64_t bh = r / cp.ktot;
        int ci = 0;
        while (ci + 1 < cp.n && kk >= lv[ci + 1].coff) ++ci;
        const int pix = kk - lv[ci].coff;
        if (pix >= lv[ci].Hl * lv[ci].Wl) continue;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        const float *src = partial + r * d.D + c4 * 4;
        for (int c = 0; c < chunks; ++c) {
            const float4 v = *reinterpret_cast<const float4 *>(src + c * chunk_stride);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const int b = (int)(bh / d.H), h = (int)(bh % d.H);
        T o[4] = {(T)s.x, (T)s.y, (T)s.z, (T)s.w};
        T *dst = grad_value + (((int64_t)b * d.S + lv[ci].start + pix) * d.H + h) * d.D + c4 * 4;
        *reinterpret_cast<uint2 *>(dst) = *reinterpret_cast<const uint2 *>(o);
    }
}

int64_t up256(int64_t v) { return (v + 255) / 256 * 256; }

int tile_chunks(const Dims &d, int tile_q, int target);
int value_chunks(const Dims &d, const CoarsePlan &) { return tile_chunks(d, kTileQ, 256); }

// samples per (query, level) as whole vectors?  (P = 4 or 8, 16-byte aligned tensors)
inline int sample_vectors(const void *loc, const void *attn, const Dims &d)
{
    if ((d.P != 4 && d.P != 8) || ((uintptr_t)loc % 16) || ((uintptr_t)attn % 8)) return 0;
    return d.P / 4;
}

template <typename T, int NS>
hipError_t launch_fwd_coarse(const void *value, const void *loc, const void *attn, void *workspace,
                             const Dims &din, const CoarsePlan &cp, hipStream_t st)
{
    Dims d = din;
    d.q_tiles = (d.Nq + kTileQ - 1) / kTileQ;
    const int64_t blocks = (int64_t)d.B * d.q_tiles * d.H;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    T *vt = (T *)workspace;
    float *cinit = (float *)((char *)workspace + up256((int64_t)d.B * d.H * cp.ktot * d.D * (int64_t)sizeof(T)));
    const int64_t pk = (int64_t)d.B * d.H * (cp.ktot / 4) * d.D;
    hipLaunchKernelGGL((coarse_pack_kernel<T>), dim3((unsigned)std::min<int64_t>((pk + 255) / 256, 256 * 32)),
                       dim3(256), 0, st, (const T *)value, vt, d, cp);
#define MMFS_L(NV) hipLaunchKernelGGL((msda_fwd_coarse<T, NS, NV>), dim3((unsigned)blocks), dim3(kDT), 0, st, \
                                      (const T *)vt, (const T *)loc, (const T *)attn, cinit, d, cp)
    switch (sample_vectors(loc, attn, d)) { case 1: MMFS_L(1); break; case 2: MMFS_L(2); break; default: MMFS_L(0); }
#undef MMFS_L
    return hipGetLastError();
}

// chunks of query tiles per (b, h) so that about ``target`` workgroups exist
int tile_chunks(const Dims &d, int tile_q = kTileQ, int target = 256)
{
    const int q_tiles = (d.Nq + tile_q - 1) / tile_q;
    const int64_t slices = (int64_t)d.B * d.H;
    const int64_t want = (target + slices - 1) / slices;
    return (int)std::max<int64_t>(1, std::min<int64_t>(want, q_tiles));
}

#ifndef MMFS_TAPS_MB
#define MMFS_TAPS_MB 2
#endif
template <typename T, int NS>
hipError_t launch_taps_coarse(const void *value, const void *loc, const void *attn, const void *go,
                              void *gl, void *ga, const Dims &d, const DotPlan &cp, hipStream_t st)
{
    constexpr int MB = MMFS_TAPS_MB;                // 2: 64-query tiles, 1024 lanes, one workgroup per CU (59 us at cfg2); 1: 68 us
    if (d.P > 16) return hipErrorInvalidValue;
    const int chunks = tile_chunks(d, 32 * MB, MB == 1 ? 768 : 256);
    const int q_tiles = (d.Nq + 32 * MB - 1) / (32 * MB);
    const int tpc = (q_tiles + chunks - 1) / chunks;
    const int64_t blocks = (int64_t)d.B * d.H * chunks;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((msda_taps_coarse<T, NS, MB>), dim3((unsigned)blocks), dim3(512 * MB), 0, st,
                       (const T *)value, (const T *)loc, (const T *)attn, (const T *)go, (T *)gl, (T *)ga, d, cp,
                       chunks, tpc);
    return hipGetLastError();
}

template <typename T, int NS>
hipError_t launch_value_coarse(const void *loc, const void *attn, const void *go, void *gv, void *partial,
                               const Dims &d, const CoarsePlan &cp, hipStream_t st)
{
    const int q_tiles = (d.Nq + kTileQ - 1) / kTileQ;
    const int tpc = (q_tiles + value_chunks(d, cp) - 1) / value_chunks(d, cp);
    // only chunks that own a tile: the epilogue adds every chunk's partial rows, and a chunk
    // without tiles would never write its own (e.g. 65 tiles over 32 chunks of 3 -> 22 chunks)
    const int chunks = (q_tiles + tpc - 1) / tpc;
    const int64_t blocks = (int64_t)d.B * d.H * chunks;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((msda_value_coarse<T, NS>), dim3((unsigned)blocks), dim3(kBT), 0, st,
                       (const T *)loc, (const T *)attn, (const T *)go, (float *)partial, d, cp, chunks, tpc);
    const int64_t items = (int64_t)d.B * d.H * cp.ktot * (d.D / 4);
    hipLaunchKernelGGL((coarse_value_epilogue<T>), dim3((unsigned)std::min<int64_t>((items + 255) / 256, 256 * 32)),
                       dim3(256), 0, st, (const float *)partial, (T *)gv, d, cp, chunks);
    return hipGetLastError();
}

#define MMFS_DENSE_DISPATCH(FN, ...)                                                            \
    do {                                                                                        \
        const int ns = d.D / 32;                                                                \
        if (dtype == 1) {                                                                       \
            if (ns == 1) return FN<half_t, 1>(__VA_ARGS__);                                     \
            if (ns == 2) return FN<half_t, 2>(__VA_ARGS__);                                     \
            if (ns == 4) return FN<half_t, 4>(__VA_ARGS__);                                     \
        } else if (dtype == 2) {                                                                \
            if (ns == 1) return FN<bf16_t, 1>(__VA_ARGS__);                                     \
            if (ns == 2) return FN<bf16_t, 2>(__VA_ARGS__);                                     \
            if (ns == 4) return FN<bf16_t, 4>(__VA_ARGS__);                                     \
        }                                                                                       \
        return hipErrorInvalidValue;                                                            \
    } while (0)

}  // namespace

// Which levels go dense.  Needs a HOST copy of the level table (the reference API keeps it in
// device memory); without one, or for fp32 / fp64 storage or other head widths, the plan is
// inactive and the plain kernels run.
HybridPlan make_hybrid_plan(int dtype, const Dims &d, const int64_t *host_shapes, const int64_t *host_start)
{
    HybridPlan p;
    p.active = p.dots_active = p.coarse_active = false;
    p.fine.n = p.fine_taps.n = -1;
    p.dots.n = 0;
    p.coarse.n = 0;
    p.coarse.ktot = 0;
    p.coarse_mask = 0;
    if (!host_shapes || !host_start) return p;
    if (dtype != 1 && dtype != 2) return p;
    if (d.D != 32 && d.D != 64 && d.D != 128) return p;
    if (d.L <= 0 || d.L > kMaxSelLevels || d.P <= 0 || d.P > 16 || d.Nq < 32) return p;
    if ((int64_t)d.Nq * d.H * d.D * 2 > kMaxSlabBytes) return p;           // grad_out rows by 32-bit offsets
    if (const char *e = getenv("MMFS_HYBRID")) if (atoi(e) == 0) return p;

    // ---- grad_loc / grad_attn by dense dot products.  Per 64 queries a level costs px * D / 31.8
    // MFMA clocks (1017 FLOP/clk/SIMD) against 8 * P * D clocks of row reads (64 B/clk/CU): on
    // paper dense pays below ~254 * P pixels (190 * P with room for the look-ups).  Levels above 256
    // pixels can be walked in chunks of whole rows that share one row (see DotChunk).
    int nft = 0;
    for (int l = 0; l < d.L; ++l) {
        const int64_t Hl = host_shapes[2 * l], Wl = host_shapes[2 * l + 1], st = host_start[l];
        const int64_t px = Hl * Wl;
        bool dense = Hl > 0 && Wl > 0 && Wl <= kCoarseMaxPx / 2 && Hl < 32768 && st >= 0 && st + px <= d.S;
        int R = 0, step = 0, chunks = 0;
        if (dense) {
            R = (int)std::min<int64_t>(Hl, kCoarseMaxPx / Wl);            // pixel rows per chunk
            step = R >= Hl ? (int)Hl : R - 1;                               // rows a chunk owns
            chunks = R >= Hl ? 1 : (int)((Hl - 1 + step - 1) / step);
            const int64_t px_eff = (int64_t)chunks * R * Wl;                // pixels multiplied, overlap included
            dense = px_eff <= 190LL * d.P && p.dots.n + chunks <= kMaxDotChunks;
            // Measured (MI355X): a chunk step costs ~4k clocks per 64 queries, 4x its
            // MFMA time, so only single-chunk levels (<= 256 pixels) beat their row reads; 32x32 levels
            // in 5 chunks ran 1.6x slower than gathering them.  MMFS_DOT_CHUNKS=1 re-enables them.
            static const bool multi = getenv("MMFS_DOT_CHUNKS") && atoi(getenv("MMFS_DOT_CHUNKS")) > 0;
            if (chunks > 1 && !multi) dense = false;
        }
        if (!dense) { p.fine_taps.idx[nft++] = (uint8_t)l; continue; }
        for (int c = 0; c < chunks; ++c) {
            DotChunk &k = p.dots.c[p.dots.n++];
            k.level = l; k.Hl = (int)Hl; k.Wl = (int)Wl; k.start = (int)st;
            k.row0 = c * step;
            k.nrows = (int)std::min<int64_t>(R, Hl - k.row0);
            k.own0 = c == 0 ? -1 : k.row0;
            k.own1 = c + 1 == chunks ? (int)Hl : k.row0 + step;
        }
    }
    if (p.dots.n > 0) {
        for (int i = nft; i < kMaxSelLevels; ++i) p.fine_taps.idx[i] = 0;
        p.fine_taps.n = nft;
        p.dots_active = true;
    }

    // ---- forward / grad_value (experimental): whole levels of <= min(256, 64 * P) pixels
    const int64_t max_px = std::min<int64_t>(kCoarseMaxPx, 64LL * d.P);
    int nf = 0;
    for (int l = 0; l < d.L; ++l) {
        const int64_t Hl = host_shapes[2 * l], Wl = host_shapes[2 * l + 1], st = host_start[l];
        const int64_t px = Hl * Wl;
        const bool dense = Hl > 0 && Wl > 0 && px <= max_px && p.coarse.n < kMaxCoarse && st >= 0 && st + px <= d.S;
        if (dense) {
            CoarseLevel &c = p.coarse.lv[p.coarse.n++];
            c.level = l; c.Hl = (int)Hl; c.Wl = (int)Wl; c.start = (int)st;
            c.coff = p.coarse.ktot;
            c.kpad = (int)((px + 7) / 8 * 8);
            p.coarse.ktot += c.kpad;
            p.coarse_mask |= 1ull << l;
        } else {
            p.fine.idx[nf++] = (uint8_t)l;
        }
    }
    if (p.coarse.n > 0) {
        for (int i = nf; i < kMaxSelLevels; ++i) p.fine.idx[i] = 0;
        p.fine.n = nf;
        p.coarse_active = true;
    }
    p.active = p.dots_active || p.coarse_active;
    return p;
}

int64_t hybrid_fwd_workspace_bytes(int dtype, const Dims &d, const HybridPlan &p)
{
    if (!p.coarse_active) return 0;
    const int64_t es = 2;
    return up256((int64_t)d.B * d.H * p.coarse.ktot * d.D * es) + up256((int64_t)d.B * d.Nq * d.H * d.D * 4);
}

const float *hybrid_fwd_init(void *workspace, const Dims &d, const HybridPlan &p)
{
    return (const float *)((char *)workspace + up256((int64_t)d.B * d.H * p.coarse.ktot * d.D * 2));
}

hipError_t forward_coarse(int dtype, const void *value, const void *loc, const void *attn, void *workspace,
                          const Dims &d, const HybridPlan &p, hipStream_t st)
{
    if (!p.coarse_active) return hipErrorInvalidValue;
    MMFS_DENSE_DISPATCH(launch_fwd_coarse, value, loc, attn, workspace, d, p.coarse, st);
}

hipError_t backward_taps_coarse(int dtype, const void *value, const void *loc, const void *attn,
                                const void *grad_out, void *grad_loc, void *grad_attn, const Dims &d,
                                const HybridPlan &p, hipStream_t st)
{
    if (!p.dots_active) return hipErrorInvalidValue;
    MMFS_DENSE_DISPATCH(launch_taps_coarse, value, loc, attn, grad_out, grad_loc, grad_attn, d, p.dots, st);
}

int64_t hybrid_bwd_partial_bytes(const Dims &d, const HybridPlan &p)
{
    if (!p.coarse_active) return 0;
    return up256((int64_t)value_chunks(d, p.coarse) * d.B * d.H * p.coarse.ktot * d.D * 4);
}

hipError_t backward_value_coarse(int dtype, const void *loc, const void *attn, const void *grad_out,
                                 void *grad_value, void *partial, const Dims &d, const HybridPlan &p,
                                 hipStream_t st)
{
    if (!p.coarse_active) return hipErrorInvalidValue;
    MMFS_DENSE_DISPATCH(launch_value_coarse, loc, attn, grad_out, grad_value, partial, d, p.coarse, st);
}

}  // namespace mmfs

#ifdef MMFS_PROFILE_DENSE
extern "C" int mmfs_debug_dense_profile(unsigned long long *out, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(mmfs::g_dense_prof), sizeof(unsigned long long) * 32);
    if (e == hipSuccess && reset) {
        unsigned long long z[32] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(mmfs::g_dense_prof), z, sizeof(z));
    }
    return (int)e;
}
#endif
