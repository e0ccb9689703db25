// msda_dense.hip -- grad_loc / grad_attn of the small levels as dense dot products on the matrix
// cores ("hybrid" routing of the backward's location / weight gradients).
//
// Why.  The row-gather kernel (msda_bwd.hip) reads four value rows per sample whatever the level's
// size, and every level receives the same number of samples.  A level of <= 256 pixels is a tiny
// dense matrix, though: for a tile of 64 queries of one (b, h)
//
//      dot[q, pix] = grad_out[q, :] . V_l[pix, :]       64 x K_l x D product, both operands ARE 16-bit storage
//
// and then grad_attn / grad_loc of a sample are the reference's scalar algebra
// (ms_deform_im2col_cuda.cuh:119-161) on 4 look-ups in dot.  v_mfma_f32_32x32x16_{bf16,f16} does
// 1017 FLOP/clk/SIMD; for the 16x16 + 8x8 levels of the north-star shape that is 2560 MFMA clocks
// per 64 queries against 8192 clocks of row reads.  The big levels stay with the gather kernel
// (LevelSel routing).  Which levels go dense is decided on the host (make_hybrid_plan), so the
// caller must know the level table there.
//
// Non-finite inputs propagate exactly as in the gather kernel (a dot product only involves its own
// two rows).  The reference has no counterpart; results are checked against the same oracle.
//
// (Round 1 also had a dense forward and a dense grad_value built on a [query, pixel] weight tile in
// LDS; both were slower than the kernels they replaced -- the tile moves 64 KiB through LDS three
// times to place 1024 non-zeros -- and were removed in round 2, when grad_value moved to the
// matrix cores by another route: csrc/msda_bwd_tile.hip.)
#include "msda_device.h"
#include "msda_env.h"
#include "msda_launch.h"
#include "msda_plan.h"
#include <cstring>
#include <cstdlib>

namespace mmfs {

namespace {

constexpr int kTileQ = 64;                       // queries per tile
constexpr int kAStride = kCoarseMaxPx + 4;       // words per query row ([q][pixel] tiles)

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
                 const Dims d, const DotPlan cp, const int chunks, const int tiles_per_chunk, const blk::PrepareJob job)
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
                store_stream(grad_attn + s, (T)ga);
                store_stream(grad_loc + 2 * s, (T)((float)p.Wl * dw * a));
                store_stream(grad_loc + 2 * s + 1, (T)((float)p.Hl * dh * a));
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
    // ---- the grad_value half's opening launch, hosted here when one call runs both halves (msda_plan.h): the first
    // workgroup clears the sort's cursors and plans; the product tiles serve as its scratch
    if (job.cursor_words > 0 && blockIdx.x == 0) {
        __syncthreads();
        blk::prepare_tail(job, reinterpret_cast<unsigned char *>(&G[0][0]));
    }
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
                              void *gl, void *ga, const Dims &d, const DotPlan &cp, hipStream_t st, const blk::PrepareJob *job)
{
    blk::PrepareJob jb;
    if (job != nullptr) jb = *job;
    else memset(&jb, 0, sizeof(jb));
    constexpr int MB = MMFS_TAPS_MB;                // 2: 64-query tiles, 1024 lanes, one workgroup per CU (59 us at cfg2); 1: 68 us
    if (d.P > 16) return hipErrorInvalidValue;
    const int chunks = tile_chunks(d, 32 * MB, MB == 1 ? 768 : 256);
    const int q_tiles = (d.Nq + 32 * MB - 1) / (32 * MB);
    const int tpc = (q_tiles + chunks - 1) / chunks;
    const int64_t blocks = (int64_t)d.B * d.H * chunks;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((msda_taps_coarse<T, NS, MB>), dim3((unsigned)blocks), dim3(512 * MB), 0, st,
                       (const T *)value, (const T *)loc, (const T *)attn, (const T *)go, (T *)gl, (T *)ga, d, cp,
                       chunks, tpc, jb);
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
    p.active = p.dots_active = false;
    p.fine_taps.n = -1;
    p.dots.n = 0;
    if (!host_shapes || !host_start) return p;
    if (dtype != 1 && dtype != 2) return p;
    if (d.D != 32 && d.D != 64 && d.D != 128) return p;
    if (d.L <= 0 || d.L > kMaxSelLevels || d.P <= 0 || d.P > 16 || d.Nq < 32) return p;
    if ((int64_t)d.Nq * d.H * d.D * 2 > kMaxSlabBytes) return p;           // grad_out rows by 32-bit offsets
    if (const char *e = knob_str(K_HYBRID)) if (atoi(e) == 0) return p;

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
            const bool multi = knob_int(K_DOT_CHUNKS, 0) > 0;
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

    p.active = p.dots_active;
    return p;
}

hipError_t backward_taps_coarse(int dtype, const void *value, const void *loc, const void *attn,
                                const void *grad_out, void *grad_loc, void *grad_attn, const Dims &d,
                                const HybridPlan &p, hipStream_t st, const blk::PrepareJob *job)
{
    if (!p.dots_active) return hipErrorInvalidValue;
    MMFS_DENSE_DISPATCH(launch_taps_coarse, value, loc, attn, grad_out, grad_loc, grad_attn, d, p.dots, st, job);
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
