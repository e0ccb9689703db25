// msda_device.h -- device-side building blocks shared by the forward and backward
// kernels of the multi-scale deformable attention op (gfx950 / CDNA4 only).
//
// Semantics follow the reference kernels
//   mm_interleaved/models/utils/ops/src/cuda/ms_deform_im2col_cuda.cuh:36-87 (bilinear tap),
//   :288-291 (pixel coordinates and the strict range test);
// the organisation (one 16-byte channel vector per lane, tap records staged in LDS
// by the whole workgroup, head -> XCD affinity) is this repository's own.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mmfs {

struct Dims {
    int B, S, H, D, L, Nq, P;
    int K;          // L * P samples per (b, q, h)
    int q_tiles;    // ceil(Nq / queries-per-block), filled by the launcher
    int lazy_attn;  // backward: grad_attn / grad_loc of samples whose attention is exactly 0 may be written as 0
    int blocks4;    // backward: 4x4 pixel blocks of all levels when the caller knows the level table on the host (else 0)
    int32_t *table_status;   // backward, level table checked on the device: where the plan reports a table it cannot serve (or null)
    int taps_algo;  // backward, grad_loc / grad_attn: 0 the library chooses, 1 row gather (+ dense small levels), 2 LDS-resident levels
    int tiles_hint;          // backward: sort tiles of all levels when the caller knows the level table on the host (else 0: a bound is launched)
    int taps_sorted;         // backward: grad_loc / grad_attn come from the cell-sorted records too (msda_bwd_taps_sorted.hip): the sort keeps
                             // zero-weight samples (unless lazy_attn), its records carry query * P + point, it zeroes the gradients of samples without a record
};

// ---------------------------------------------------------------- storage types
typedef _Float16 half_t;
typedef __bf16 bf16_t;

template <typename T> struct Acc { typedef float type; };
template <> struct Acc<double> { typedef double type; };

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }

// 16 bytes of T -> VEC floats
template <typename T> struct Vec16;

template <> struct Vec16<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void unpack(const uint4 &r, float (&o)[4]) {
        o[0] = __uint_as_float(r.x); o[1] = __uint_as_float(r.y);
        o[2] = __uint_as_float(r.z); o[3] = __uint_as_float(r.w);
    }
    static __device__ __forceinline__ uint4 pack(const float (&v)[4]) {
        return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]),
                          __float_as_uint(v[2]), __float_as_uint(v[3]));
    }
};

template <> struct Vec16<bf16_t> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void unpack(const uint4 &r, float (&o)[8]) {
        // bf16 -> f32 is a 16-bit shift: low half via shl, high half via mask
        o[0] = __uint_as_float(r.x << 16); o[1] = __uint_as_float(r.x & 0xffff0000u);
        o[2] = __uint_as_float(r.y << 16); o[3] = __uint_as_float(r.y & 0xffff0000u);
        o[4] = __uint_as_float(r.z << 16); o[5] = __uint_as_float(r.z & 0xffff0000u);
        o[6] = __uint_as_float(r.w << 16); o[7] = __uint_as_float(r.w & 0xffff0000u);
    }
    static __device__ __forceinline__ uint32_t pk(float a, float b) {
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        bf2 p; p[0] = (__bf16)a; p[1] = (__bf16)b;          // RNE; v_cvt_pk_bf16_f32 on gfx950
        return __builtin_bit_cast(uint32_t, p);
    }
    static __device__ __forceinline__ uint4 pack(const float (&v)[8]) {
        return make_uint4(pk(v[0], v[1]), pk(v[2], v[3]), pk(v[4], v[5]), pk(v[6], v[7]));
    }
};

template <> struct Vec16<half_t> {
    static constexpr int N = 8;
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ void up(uint32_t w, float &a, float &b) {
        h2 p = __builtin_bit_cast(h2, w); a = (float)p[0]; b = (float)p[1];
    }
    static __device__ __forceinline__ void unpack(const uint4 &r, float (&o)[8]) {
        up(r.x, o[0], o[1]); up(r.y, o[2], o[3]); up(r.z, o[4], o[5]); up(r.w, o[6], o[7]);
    }
    static __device__ __forceinline__ uint32_t pk(float a, float b) {
        h2 p; p[0] = (_Float16)a; p[1] = (_Float16)b;        // RNE
        return __builtin_bit_cast(uint32_t, p);
    }
    static __device__ __forceinline__ uint4 pack(const float (&v)[8]) {
        return make_uint4(pk(v[0], v[1]), pk(v[2], v[3]), pk(v[4], v[5]), pk(v[6], v[7]));
    }
};

// ---------------------------------------------------------------- bilinear tap
// One sample's 2x2 footprint.  row[i] is the pixel index inside the whole value
// tensor's S axis (level start included), or -1 when the corner is outside the map
// or the sample fails the range test (then it contributes nothing, cuh:291).
template <typename A>
struct Tap {
    int row[4];     // (y0,x0) (y0,x1) (y1,x0) (y1,x1)
    A fx, fy;       // fractional parts ("lw", "lh" in the reference)
    int Hl, Wl;
};

template <typename A>
__device__ __forceinline__ Tap<A> locate(A lx, A ly, int Hl, int Wl, int level_start)
{
    Tap<A> t;
    t.Hl = Hl; t.Wl = Wl;
    const A y = ly * (A)Hl - (A)0.5;
    const A x = lx * (A)Wl - (A)0.5;
    // strict comparisons: NaN fails, exactly -1 / Hl / Wl fail (cuh:291)
    const bool inside = (y > (A)-1) && (x > (A)-1) && (y < (A)Hl) && (x < (A)Wl);
    const A yf = floor(y), xf = floor(x);
    const int y0 = inside ? (int)yf : 0, x0 = inside ? (int)xf : 0;
    t.fy = inside ? y - yf : (A)0;
    t.fx = inside ? x - xf : (A)0;
    const bool top = y0 >= 0, left = x0 >= 0, bottom = y0 + 1 <= Hl - 1, right = x0 + 1 <= Wl - 1;
    const int base = level_start + y0 * Wl + x0;
    t.row[0] = (inside && top && left) ? base : -1;
    t.row[1] = (inside && top && right) ? base + 1 : -1;
    t.row[2] = (inside && bottom && left) ? base + Wl : -1;
    t.row[3] = (inside && bottom && right) ? base + Wl + 1 : -1;
    return t;
}

// ---------------------------------------------------------------- staging helpers
// The texture path spends the same 16 clocks on a wave-wide load whether a lane asks for 2 bytes or
// for 16, and the row-gather kernels are bound by exactly that path: what a workgroup reads besides
// its rows has to be FEW instructions.  So the level table is fetched once per workgroup into LDS
// (instead of three 8-byte loads per sample), a sample's (x, y) pair is one load, and the staging
// loops run over the live samples only (no lanes parked on padding).
constexpr int kStageLevels = 128;

struct LevelLds {
    int tab[kStageLevels * 3];           // Hl, Wl, start of levels [0, min(L, kStageLevels))
    // call from every thread of the workgroup, then __syncthreads() before the first get()
    __device__ __forceinline__ void load(const int64_t *__restrict__ shapes, const int64_t *__restrict__ start,
                                         int L, int tid, int nthreads) {
        for (int l = tid; l < L && l < kStageLevels; l += nthreads) {
            tab[3 * l] = (int)shapes[2 * l]; tab[3 * l + 1] = (int)shapes[2 * l + 1]; tab[3 * l + 2] = (int)start[l];
        }
    }
    __device__ __forceinline__ void get(const int64_t *__restrict__ shapes, const int64_t *__restrict__ start,
                                        int l, int &Hl, int &Wl, int &st) const {
        if (l < kStageLevels) { Hl = tab[3 * l]; Wl = tab[3 * l + 1]; st = tab[3 * l + 2]; }
        else { Hl = (int)shapes[2 * l]; Wl = (int)shapes[2 * l + 1]; st = (int)start[l]; }
    }
};

// (x, y) of sample s as one load when the pair is naturally aligned (pair_ok: the tensor base is)
template <typename T>
__device__ __forceinline__ void load_xy(const T *__restrict__ loc, int64_t s, bool pair_ok, float &x, float &y)
{
    if (sizeof(T) == 2 && pair_ok) {
        const uint32_t w = *reinterpret_cast<const uint32_t *>(loc + 2 * s);
        T p[2];
        __builtin_memcpy(p, &w, 4);
        x = to_f32(p[0]); y = to_f32(p[1]);
    } else if (sizeof(T) == 4 && pair_ok) {
        const uint2 w = *reinterpret_cast<const uint2 *>(loc + 2 * s);
        x = __uint_as_float(w.x); y = __uint_as_float(w.y);
    } else {
        x = to_f32(loc[2 * s]); y = to_f32(loc[2 * s + 1]);
    }
}

// (grad_loc is a final result of the step: non-temporal, see store16_stream)
template <typename T>
__device__ __forceinline__ void store_xy(T *__restrict__ dst, int64_t s, bool pair_ok, float x, float y)
{
    T p[2] = {(T)x, (T)y};
    if (sizeof(T) == 2 && pair_ok) {
        uint32_t w;
        __builtin_memcpy(&w, p, 4);
        __builtin_nontemporal_store(w, reinterpret_cast<uint32_t *>(dst + 2 * s));
    } else if (sizeof(T) == 4 && pair_ok) {
        uint2 w;
        __builtin_memcpy(&w, p, 8);
        __builtin_nontemporal_store(w.x, reinterpret_cast<uint32_t *>(dst + 2 * s));
        __builtin_nontemporal_store(w.y, reinterpret_cast<uint32_t *>(dst + 2 * s) + 1);
    } else {
        dst[2 * s] = p[0]; dst[2 * s + 1] = p[1];
    }
}

// ---------------------------------------------------------------- streamed outputs
// Results nobody reads again within the step (the forward's output, grad_value, grad_loc / grad_attn) leave with
// the non-temporal hint: written normally they displaced the NEXT kernel's inputs from the memory-side cache --
// with the hint on grad_value alone the forward that follows the backward ran 8.6 us faster
// (profiles/r03_experiments.md, r03i).
__device__ __forceinline__ void store16_stream(void *dst, const uint4 &v)
{
    uint32_t *o = reinterpret_cast<uint32_t *>(dst);
    __builtin_nontemporal_store(v.x, o); __builtin_nontemporal_store(v.y, o + 1);
    __builtin_nontemporal_store(v.z, o + 2); __builtin_nontemporal_store(v.w, o + 3);
}

// one element / an (x, y) pair of a streamed result (grad_attn, grad_loc)
template <typename T>
__device__ __forceinline__ void store_stream(T *dst, T v)
{
    if constexpr (sizeof(T) == 2) __builtin_nontemporal_store(__builtin_bit_cast(uint16_t, v), reinterpret_cast<uint16_t *>(dst));
    else if constexpr (sizeof(T) == 4) __builtin_nontemporal_store(__builtin_bit_cast(uint32_t, v), reinterpret_cast<uint32_t *>(dst));
    else *dst = v;
}

// ---------------------------------------------------------------- buffer addressing
// Row gathers go through a buffer descriptor (SRD) whose base is the workgroup's
// (batch, head) slab: the per-lane address is a 32-bit byte offset, and an offset at or
// beyond num_records makes the hardware return zeros WITHOUT touching memory -- exactly
// the reference's "corner outside the map reads 0" (cuh:58-81), with no select and no
// clamped dummy read.  Needs the slab to be < 2 GiB; the launchers fall back to the
// flat-address kernels otherwise.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kOobOffset = 0x80000000u;
constexpr int64_t kMaxSlabBytes = 0x7fffffffLL;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_slab_rsrc(const void *base, int64_t bytes)
{
    // make wave-uniformity provable (cdna_hip_programming.md T20): rebuild the pointer
    // from readfirstlane'd halves
    const uint64_t a = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    const uint32_t n = __builtin_amdgcn_readfirstlane((uint32_t)bytes);
    void *p = (void *)(((uint64_t)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(p, (short)0, (int)n, 0x00020000);
}

__device__ __forceinline__ uint4 buffer_load16(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_offset)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_offset, 0, 0);
    return make_uint4(v[0], v[1], v[2], v[3]);
}

// ---------------------------------------------------------------- level routing (hybrid path)
// When the host knows the level table (a host copy handed to mmfs_msda_backward_hybrid), levels of at
// most kCoarseMaxPx pixels are taken out of the row-gather taps kernel and their grad_loc / grad_attn
// evaluated as dense dot products on the matrix cores (msda_dense.hip); the gather kernel then visits
// only the levels listed in a LevelSel.
constexpr int kMaxSelLevels = 64;      // hybrid routing only for L <= 64
constexpr int kCoarseMaxPx = 256;

struct LevelSel {
    int n;                             // levels to visit; < 0: all L levels in order
    uint8_t idx[kMaxSelLevels];
};

// Dense dot products for grad_loc / grad_attn (msda_taps_coarse): a level is walked in chunks of
// whole pixel rows, at most 256 pixels each, consecutive chunks sharing one row, so that every
// sample's 2x2 footprint lies inside the chunk that owns its top row.
constexpr int kMaxDotChunks = 48;

struct DotChunk {
    int level, Hl, Wl, start;          // the level (index in the table, extent, first pixel on the S axis)
    int row0, nrows;                   // pixel rows [row0, row0 + nrows) of the level; nrows * Wl <= 256
    int own0, own1;                    // finishes the samples whose top row y0 is in [own0, own1); own0 = -1
                                       // also takes the samples outside the map (all-zero gradients)
};

struct DotPlan {
    int n;
    DotChunk c[kMaxDotChunks];
};

// Workgroup -> (b, h, first query).  Blocks are dealt to XCDs round-robin
// (block i -> XCD i % 8, observed, MI355X_MICROARCH.md "Workgroup dispatch"), so
// taking h = block % H pins every head's value slice [S, D] of a sample to one
// XCD's 4 MiB L2 when H is a multiple of 8 (and to H of the 8 L2s otherwise).
// This is a speed choice only: any placement gives the same results.
struct BlockCoord { int b, h, q0; };
__device__ __forceinline__ BlockCoord block_coord(const Dims &d, int queries_per_block)
{
    BlockCoord c;
    const int bid = blockIdx.x;
#ifdef MMFS_LINEAR_MAP                 // experiment: (b, h, q-tile) order, q-tile fastest
    c.q0 = (bid % d.q_tiles) * queries_per_block;
    const int t = bid / d.q_tiles;
    c.h = t % d.H;
    c.b = t / d.H;
#else
    c.h = bid % d.H;
    const int t = bid / d.H;
    c.q0 = (t % d.q_tiles) * queries_per_block;
    c.b = t / d.q_tiles;
#endif
    return c;
}

}  // namespace mmfs
