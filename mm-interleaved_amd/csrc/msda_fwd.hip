// msda_fwd.hip -- forward kernels of multi-scale deformable attention for gfx950.
//
// Replaces the reference forward
//   mm_interleaved/models/utils/ops/src/cuda/ms_deform_im2col_cuda.cuh:240-302
//   (one thread per output scalar, every thread re-deriving every tap)
// with a design built around the 64-wide wavefront and the LDS:
//
//   * a workgroup owns QPB consecutive queries of ONE (batch, head) pair; the head is
//     chosen from the block index so a head's value slice stays in one XCD's L2;
//   * the workgroup first turns the tile's sampling locations / attention weights
//     (read once, coalesced) into tap records in LDS: 4 pixel-row indices and 4
//     weights already multiplied by the attention weight;
//   * then LPI = D*sizeof(T)/16 lanes cooperate on one query: every lane owns one
//     16-byte channel vector, so each corner of a tap is a single fully coalesced
//     D*sizeof(T)-byte row read, and the record is an LDS broadcast read;
//   * accumulation is fp32 in registers (the reference's opmath), no cross-lane
//     traffic at all in the forward; the result is stored as one 16-byte vector.
//
// A scalar kernel (one thread per output element) covers head widths that are not
// 16-byte * power-of-two, and fp64.
#include "msda_device.h"
#include "msda_launch.h"

namespace mmfs {

constexpr int kThreads = 256;
constexpr int kRecsPerBlock = 512;      // tap records staged per chunk (16 KiB)
#ifndef MMFS_FWD_UNROLL
#define MMFS_FWD_UNROLL 2      // measured on MI355X: 2 -> 183 us, 1 -> 198, 4 -> 235, 8 -> 231 (cfg2, bf16)
#endif
constexpr int kUnroll = MMFS_FWD_UNROLL;   // taps in flight per lane (4x row reads each)

struct alignas(16) FwdRec {
    int row[4];
    float w[4];
};

// BUF = true : row reads through a buffer descriptor (32-bit offsets, hardware zero for
//              corners outside the map)               -- slabs < 2 GiB, the normal case
// BUF = false: 64-bit flat addresses, clamped read + select
template <typename T, int LPI, bool BUF>
__global__ void __launch_bounds__(kThreads)
msda_fwd_vec(const T *__restrict__ value, const int64_t *__restrict__ shapes,
             const int64_t *__restrict__ start, const T *__restrict__ loc,
             const T *__restrict__ attn, T *__restrict__ out, const Dims d)
{
    typedef Vec16<T> V;
    constexpr int VEC = V::N;
    constexpr int QPB = kThreads / LPI;           // queries per block
    constexpr int KC = (kRecsPerBlock / QPB) > kUnroll ? (kRecsPerBlock / QPB) : kUnroll;  // samples per query per chunk (pow2)
    constexpr int STRIDE = 2 * KC + 1;            // uint4 units; +1 breaks the bank alignment
    __shared__ uint4 lds[QPB * STRIDE];
    __shared__ LevelLds levels;
    // taps of the chunk that weigh something for at least one query of a wave: bit kk of the wave's word
    // (set while staging; the gather walks the set bits, so a tap nobody needs is never issued and the walk
    // has a counted trip -- the shape the compiler's wait-count pass pipelines cleanly)
    constexpr bool kPipe = BUF && KC <= 64;
    constexpr int QPW = 64 / LPI > 0 ? 64 / LPI : 1;       // queries per wave
    __shared__ unsigned long long live[kThreads / 64];

    const BlockCoord bc = block_coord(d, QPB);
    const int tid = threadIdx.x;
    if (kPipe && tid < kThreads / 64) live[tid] = 0ull;
    levels.load(shapes, start, d.L, tid, kThreads);
    const bool pair_ok = ((uintptr_t)loc & (2 * sizeof(T) - 1)) == 0;
    __syncthreads();
    const int qi = tid / LPI, lig = tid % LPI;
    const int q = bc.q0 + qi;
    const bool q_ok = q < d.Nq;
    const int Ksel = d.K;

    const int64_t HD = (int64_t)d.H * d.D;
    const T *slab = value + ((int64_t)bc.b * d.S) * HD + (int64_t)bc.h * d.D;   // this (b, h)
    const T *vbase = slab + lig * VEC;
    const uint32_t row_bytes = (uint32_t)(HD * sizeof(T));
    const uint32_t lane_off = (uint32_t)(lig * 16);
    __amdgpu_buffer_rsrc_t rsrc;
    if (BUF) rsrc = make_slab_rsrc(slab, ((int64_t)d.S * HD - (int64_t)bc.h * d.D) * (int64_t)sizeof(T));

    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;

    for (int k0 = 0; k0 < Ksel; k0 += KC) {
        const int kc = min(KC, Ksel - k0);
        const int kc_pad = (kc + kUnroll - 1) / kUnroll * kUnroll;
        if (k0 > 0) __syncthreads();              // previous chunk fully consumed
        // ---- stage: locations + weights -> tap records (coalesced over samples)
        for (int r = tid; r < QPB * kc_pad; r += kThreads) {
            const int rq = r / kc_pad, kk = r - rq * kc_pad;
            FwdRec rec;
            rec.row[0] = rec.row[1] = rec.row[2] = rec.row[3] = -1;
            rec.w[0] = rec.w[1] = rec.w[2] = rec.w[3] = 0.f;
            const int sq = bc.q0 + rq;
            if (kk < kc && sq < d.Nq) {
                const int k = k0 + kk;
                const int l = k / d.P;
                const int64_t s = (((int64_t)bc.b * d.Nq + sq) * d.H + bc.h) * d.K + k;
                float lx, ly;
                load_xy(loc, s, pair_ok, lx, ly);
                const float a = to_f32(attn[s]);
                int Hl, Wl, lstart;
                levels.get(shapes, start, l, Hl, Wl, lstart);
                const Tap<float> t = locate<float>(lx, ly, Hl, Wl, lstart);
                const float gy = 1.f - t.fy, gx = 1.f - t.fx;
                // a zero attention weight (an image the token cannot see: the masked softmax gives
                // exactly 0, mmfs.py:203-231) reads no rows at all -- it is marked "outside"
                if (a != 0.f) {
                    rec.row[0] = t.row[0]; rec.row[1] = t.row[1]; rec.row[2] = t.row[2]; rec.row[3] = t.row[3];
                }
                rec.w[0] = gy * gx * a; rec.w[1] = gy * t.fx * a;
                rec.w[2] = t.fy * gx * a; rec.w[3] = t.fy * t.fx * a;
            }
            uint4 *dst = &lds[rq * STRIDE + 2 * kk];
            if (!BUF) {
                // flat addresses have no descriptor to stop a row that a malformed level table puts past S
#pragma unroll
                for (int c = 0; c < 4; ++c) rec.row[c] = rec.row[c] < d.S ? rec.row[c] : -1;
            }
            if (BUF) {
#pragma unroll
                for (int c = 0; c < 4; ++c)      // pixel row -> byte offset in the slab, or "outside"
                    rec.row[c] = rec.row[c] >= 0 ? (int)((uint32_t)rec.row[c] * row_bytes) : (int)kOobOffset;
            }
            dst[0] = make_uint4(rec.row[0], rec.row[1], rec.row[2], rec.row[3]);
            dst[1] = make_uint4(__float_as_uint(rec.w[0]), __float_as_uint(rec.w[1]),
                                __float_as_uint(rec.w[2]), __float_as_uint(rec.w[3]));
            if (kPipe) {
                const uint32_t any_w = (__float_as_uint(rec.w[0]) | __float_as_uint(rec.w[1]) |
                                        __float_as_uint(rec.w[2]) | __float_as_uint(rec.w[3])) << 1;   // (-0 is zero too)
                if (any_w != 0u) atomicOr(&live[rq / QPW], 1ull << kk);
            }
        }
        __syncthreads();
        if (kPipe) {
            // ---- gather, software-pipelined: the 4 row reads of the next live tap are in flight while the
            // current one is multiplied (4..8 reads in flight per lane)
            const int wv = tid >> 6;
            const unsigned long long mraw = live[wv];
            unsigned long long m = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(mraw >> 32)) << 32) |
                                   (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)mraw);
            if ((tid & 63) == 0) live[wv] = 0ull;                 // (mine; the next chunk's staging sets it after the barrier)
            const uint4 *recs = &lds[qi * STRIDE];
            uint4 rawA[4], rawB[4], wA, wB;
            auto issue = [&](uint4 (&raw)[4], uint4 &ww) {
                const int kk = __builtin_ctzll(m);
                m &= m - 1ull;
                const uint4 rr = recs[2 * kk];
                ww = recs[2 * kk + 1];
                raw[0] = buffer_load16(rsrc, rr.x + lane_off);
                raw[1] = buffer_load16(rsrc, rr.y + lane_off);
                raw[2] = buffer_load16(rsrc, rr.z + lane_off);
                raw[3] = buffer_load16(rsrc, rr.w + lane_off);
                __builtin_amdgcn_sched_barrier(0);
            };
            auto consume = [&](const uint4 (&raw)[4], const uint4 &ww) {
                const float w4[4] = {__uint_as_float(ww.x), __uint_as_float(ww.y), __uint_as_float(ww.z), __uint_as_float(ww.w)};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float v[VEC];
                    V::unpack(raw[c], v);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) acc[i] = fmaf(w4[c], v[i], acc[i]);
                }
#pragma unroll
                for (int i = 0; i < VEC; ++i) asm volatile("" : "+v"(acc[i]));     // the sums are due HERE, not after the next reads
                __builtin_amdgcn_sched_barrier(0);
            };
            const int n_live = __builtin_popcountll(m);
            if (n_live & 1) { issue(rawA, wA); consume(rawA, wA); }
            if (n_live >= 2) {
                issue(rawA, wA);
                for (int i = 2; i < n_live - 1; i += 2) {
                    issue(rawB, wB);
                    consume(rawA, wA);
                    issue(rawA, wA);
                    consume(rawB, wB);
                }
                issue(rawB, wB);
                consume(rawA, wA);
                consume(rawB, wB);
            }
        } else
        // ---- gather: kUnroll taps (4*kUnroll row reads) in flight per lane
        if (q_ok) {
            const uint4 *recs = &lds[qi * STRIDE];
            for (int kk = 0; kk < kc_pad; kk += kUnroll) {
                uint4 raw[kUnroll][4];
                float w[kUnroll][4];
                bool ok[kUnroll][4];
                uint4 rrs[kUnroll];
                uint32_t any_w = 0u;
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    rrs[u] = recs[2 * (kk + u)];
                    const uint4 ww = recs[2 * (kk + u) + 1];
                    any_w |= (ww.x | ww.y | ww.z | ww.w) << 1;          // (sign bit aside: -0 is zero too)
                    w[u][0] = __uint_as_float(ww.x); w[u][1] = __uint_as_float(ww.y);
                    w[u][2] = __uint_as_float(ww.z); w[u][3] = __uint_as_float(ww.w);
                }
                // every tap of every query of this wave weighs zero (consecutive tokens share what
                // they can see, so whole waves are blind to an image): nothing to read, nothing to add.
                // An "outside" row costs no data but its load still costs the address path its cycles.
                if (__builtin_amdgcn_ballot_w64(any_w != 0u) == 0ull) continue;
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    const uint4 rr = rrs[u];
                    const int rows[4] = {(int)rr.x, (int)rr.y, (int)rr.z, (int)rr.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (BUF) {
                            ok[u][c] = true;
                            raw[u][c] = buffer_load16(rsrc, (uint32_t)rows[c] + lane_off);
                        } else {
                            ok[u][c] = rows[c] >= 0;
                            const int64_t off = (int64_t)max(rows[c], 0) * HD;
                            raw[u][c] = *reinterpret_cast<const uint4 *>(vbase + off);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float v[VEC];
                        V::unpack(raw[u][c], v);
                        const float wc = w[u][c];
#pragma unroll
                        for (int i = 0; i < VEC; ++i)
                            acc[i] = fmaf(wc, (BUF || ok[u][c]) ? v[i] : 0.f, acc[i]);
                    }
                }
            }
        }
    }
    if (q_ok) {
        T *o = out + (((int64_t)bc.b * d.Nq + q) * d.H + bc.h) * d.D + lig * VEC;
        store16_stream(o, V::pack(acc));                 // (the output is not read again in the step: profiles/r03_experiments.md r03j)
    }
}

// Scalar fallback: any D, any storage type (incl. fp64).  One thread per output
// element, same arithmetic.  Not a tuned path: the real head widths (32, 64, 128)
// all take msda_fwd_vec.
template <typename T>
__global__ void __launch_bounds__(kThreads)
msda_fwd_scalar(const T *__restrict__ value, const int64_t *__restrict__ shapes,
                const int64_t *__restrict__ start, const T *__restrict__ loc,
                const T *__restrict__ attn, T *__restrict__ out, const Dims d, const int64_t total)
{
    typedef typename Acc<T>::type A;
    const int64_t HD = (int64_t)d.H * d.D;
    for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * kThreads) {
        const int c = (int)(idx % d.D);
        const int64_t item = idx / d.D;                 // (b*Nq + q)*H + h
        const int h = (int)(item % d.H);
        const int64_t b = item / d.H / d.Nq;
        const T *vb = value + b * d.S * HD + (int64_t)h * d.D + c;
        A acc = 0;
        for (int k = 0; k < d.K; ++k) {
            const int l = k / d.P;
            const int64_t s = item * d.K + k;
            const Tap<A> t = locate<A>((A)loc[2 * s], (A)loc[2 * s + 1], (int)shapes[2 * l],
                                       (int)shapes[2 * l + 1], (int)start[l]);
            const A gy = 1 - t.fy, gx = 1 - t.fx;
            const A v1 = t.row[0] >= 0 ? (A)vb[(int64_t)t.row[0] * HD] : (A)0;
            const A v2 = t.row[1] >= 0 ? (A)vb[(int64_t)t.row[1] * HD] : (A)0;
            const A v3 = t.row[2] >= 0 ? (A)vb[(int64_t)t.row[2] * HD] : (A)0;
            const A v4 = t.row[3] >= 0 ? (A)vb[(int64_t)t.row[3] * HD] : (A)0;
            acc += (gy * gx * v1 + gy * t.fx * v2 + t.fy * gx * v3 + t.fy * t.fx * v4) * (A)attn[s];
        }
        out[idx] = (T)acc;
    }
}

// ---------------------------------------------------------------- launchers
template <typename T, int LPI>
static hipError_t launch_vec(const void *value, const int64_t *shapes, const int64_t *start,
                             const void *loc, const void *attn, void *out, Dims d, hipStream_t st)
{
    constexpr int QPB = kThreads / LPI;
    d.q_tiles = (d.Nq + QPB - 1) / QPB;
    const int64_t blocks = (int64_t)d.B * d.q_tiles * d.H;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    // one (batch) slab of value must be addressable with 31-bit byte offsets for the buffer path
    const bool buf = (int64_t)d.S * d.H * d.D * (int64_t)sizeof(T) <= kMaxSlabBytes;
    if (buf)
        hipLaunchKernelGGL((msda_fwd_vec<T, LPI, true>), dim3((unsigned)blocks), dim3(kThreads), 0, st,
                           (const T *)value, shapes, start, (const T *)loc, (const T *)attn, (T *)out, d);
    else
        hipLaunchKernelGGL((msda_fwd_vec<T, LPI, false>), dim3((unsigned)blocks), dim3(kThreads), 0, st,
                           (const T *)value, shapes, start, (const T *)loc, (const T *)attn, (T *)out, d);
    return hipGetLastError();
}

template <typename T>
static hipError_t launch_scalar(const void *value, const int64_t *shapes, const int64_t *start,
                                const void *loc, const void *attn, void *out, Dims d, hipStream_t st)
{
    const int64_t total = (int64_t)d.B * d.Nq * d.H * d.D;
    const int64_t blocks = std::min<int64_t>((total + kThreads - 1) / kThreads, 256 * 32);
    hipLaunchKernelGGL((msda_fwd_scalar<T>), dim3((unsigned)blocks), dim3(kThreads), 0, st,
                       (const T *)value, shapes, start, (const T *)loc, (const T *)attn, (T *)out, d, total);
    return hipGetLastError();
}

template <typename T>
static hipError_t dispatch_fwd(const void *value, const int64_t *shapes, const int64_t *start,
                               const void *loc, const void *attn, void *out, const Dims &d, hipStream_t st)
{
    constexpr int VEC = 16 / (int)sizeof(T);
    if (d.D % VEC == 0) {
        switch (d.D / VEC) {
#define MMFS_CASE(n) case n: return launch_vec<T, n>(value, shapes, start, loc, attn, out, d, st);
            MMFS_CASE(1) MMFS_CASE(2) MMFS_CASE(4) MMFS_CASE(8) MMFS_CASE(16) MMFS_CASE(32) MMFS_CASE(64)
#undef MMFS_CASE
            default: break;
        }
    }
    return launch_scalar<T>(value, shapes, start, loc, attn, out, d, st);
}

hipError_t forward(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                   const void *loc, const void *attn, void *out, const Dims &d, hipStream_t st, int algo)
{
    if (algo == 4 || (algo == 0 && fwd_wq_applies(dtype, d)))
        return forward_wq(dtype, value, shapes, start, loc, attn, out, d, st);
    if (algo == 3 || (algo == 0 && fwd_q8_applies(dtype, d)))
        return forward_q8(dtype, value, shapes, start, loc, attn, out, d, st);
    if (algo == 2 || (algo == 0 && fwd_mma_applies(dtype, d)))
        return forward_mma(dtype, value, shapes, start, loc, attn, out, d, st);
    switch (dtype) {
        case 0: return dispatch_fwd<float>(value, shapes, start, loc, attn, out, d, st);
        case 1: return dispatch_fwd<half_t>(value, shapes, start, loc, attn, out, d, st);
        case 2: return dispatch_fwd<bf16_t>(value, shapes, start, loc, attn, out, d, st);
        case 3: return launch_scalar<double>(value, shapes, start, loc, attn, out, d, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mmfs
