// msda_fwd_cached.hip -- forward with the coarse levels of the (batch, head) slice in LDS.
//
// Same arithmetic and tap-record scheme as msda_fwd.hip (reference semantics:
// ms_deform_im2col_cuda.cuh:240-302), different work decomposition: a 1024-lane workgroup
// owns a long run of queries of ONE (b, h), copies every level that fits its LDS budget
// once (msda_cache.h), and then serves those levels' taps with ds_read_b128 while the big
// levels still go through the buffer-load path.  See msda_cache.h for why.
#include "msda_cache.h"
#include "msda_launch.h"
#include <cstdlib>

namespace mmfs {

namespace {

constexpr int kThreads = 1024;
constexpr int kUnroll = 2;
constexpr int kCacheBytes = 112 * 1024; // value cache; + 16 waves x ~2 KiB records + 2 KiB tables <= 160 KiB

template <int LPI> struct WaveTile {
    static constexpr int QPW = 64 / LPI;                                   // queries per wave pass
    static constexpr int KC = LPI > kUnroll ? LPI : kUnroll;               // samples per chunk: QPW*KC = 64 records
    static constexpr int STRIDE = 2 * KC + 1;                              // uint4 units per query (+1: bank skew)
    static constexpr int WAVE_RECS = QPW * STRIDE;                         // uint4 units per wave
};

template <typename T, int LPI>
__global__ void __launch_bounds__(kThreads)
msda_fwd_cached(const T *__restrict__ value, const int64_t *__restrict__ shapes,
                const int64_t *__restrict__ start, const T *__restrict__ loc,
                const T *__restrict__ attn, T *__restrict__ out, const Dims d, const int q_per_block,
                const int cache_budget)
{
    typedef Vec16<T> V;
    typedef WaveTile<LPI> W;
    constexpr int VEC = V::N;
    constexpr int QPW = W::QPW, KC = W::KC, STRIDE = W::STRIDE;
    constexpr int RPL = (QPW * KC + 63) / 64;                                 // records per lane per chunk
    extern __shared__ uint4 smem[];
    LevelInfo *lvl = reinterpret_cast<LevelInfo *>(smem);                     // [kMaxCacheLevels]
    int *scratch2 = reinterpret_cast<int *>(smem + kMaxCacheLevels);          // 16 bytes
    uint4 *cache = smem + kMaxCacheLevels + 1;                                // [kCacheBytes / 16]
    uint4 *recs_all = cache + kCacheBytes / 16;                               // kWaves x [QPW * STRIDE]

    const BlockCoord bc = block_coord(d, q_per_block);
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int qi = lane / LPI, lig = lane % LPI;
    const int64_t HD = (int64_t)d.H * d.D;
    const T *slab = value + ((int64_t)bc.b * d.S) * HD + (int64_t)bc.h * d.D;
    const uint32_t row_bytes = (uint32_t)(HD * sizeof(T));    // pixel stride in the slab
    const uint32_t head_bytes = (uint32_t)(d.D * sizeof(T));  // pixel stride in the cache
    const uint32_t lane_off = (uint32_t)(lig * 16);
    const __amdgpu_buffer_rsrc_t rsrc =
        make_slab_rsrc(slab, ((int64_t)d.S * HD - (int64_t)bc.h * d.D) * (int64_t)sizeof(T));

    plan_level_cache<kThreads>(shapes, start, d.L, (int)head_bytes, cache_budget, lvl, scratch2);
    fill_level_cache<kThreads>(reinterpret_cast<const char *>(slab), row_bytes, (int)head_bytes, d.L, lvl, cache);
    __syncthreads();
    // From here on the 16 waves never synchronise again: each owns its queries and its own
    // record area, so they drift apart and one wave's staging loads overlap another's gathers
    // (a workgroup-wide barrier per chunk measured 20 % slower than not caching at all).
    uint4 *wrecs = recs_all + wave * W::WAVE_RECS;
    const int q_end = min(d.Nq, bc.q0 + q_per_block);
    for (int q0 = bc.q0 + wave * QPW; q0 < q_end; q0 += (kThreads / 64) * QPW) {
        const int q = q0 + qi;
        const bool q_ok = q < q_end;
        float acc[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
        for (int k0 = 0; k0 < d.K; k0 += KC) {
            const int kc = min(KC, d.K - k0);
            const int kc_pad = (kc + kUnroll - 1) / kUnroll * kUnroll;
            // ---- stage this wave's QPW x kc taps (one or two per lane)
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                const int r = lane + i * 64;
                const int rq = r / KC, kk = r % KC;
                if (rq >= QPW || kk >= kc_pad) continue;
                uint32_t off[4] = {kOobOffset, kOobOffset, kOobOffset, kOobOffset};
                float w[4] = {0.f, 0.f, 0.f, 0.f};
                const int sq = q0 + rq;
                const int l = min((k0 + kk) / d.P, d.L - 1);
                const LevelInfo lv = lvl[l];
                if (lv.lds_off >= 0) off[0] = off[1] = off[2] = off[3] = kCachedBit;   // zero row
                if (kk < kc && sq < q_end) {
                    const int64_t s = (((int64_t)bc.b * d.Nq + sq) * d.H + bc.h) * d.K + (k0 + kk);
                    const float a = to_f32(attn[s]);
                    const Tap<float> t = locate<float>(to_f32(loc[2 * s]), to_f32(loc[2 * s + 1]), lv.Hl, lv.Wl, 0);
                    const float gy = 1.f - t.fy, gx = 1.f - t.fx;
                    w[0] = gy * gx * a; w[1] = gy * t.fx * a; w[2] = t.fy * gx * a; w[3] = t.fy * t.fx * a;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (lv.lds_off >= 0)     // level-local pixel -> cache byte offset (zero row if outside)
                            off[c] = kCachedBit | (t.row[c] >= 0 ? (uint32_t)lv.lds_off + (uint32_t)t.row[c] * head_bytes : 0u);
                        else
                            off[c] = t.row[c] >= 0 ? (uint32_t)(lv.start + t.row[c]) * row_bytes : kOobOffset;
                    }
                }
                uint4 *dst = &wrecs[rq * STRIDE + 2 * kk];
                dst[0] = make_uint4(off[0], off[1], off[2], off[3]);
                dst[1] = make_uint4(__float_as_uint(w[0]), __float_as_uint(w[1]), __float_as_uint(w[2]),
                                    __float_as_uint(w[3]));
            }
            __builtin_amdgcn_wave_barrier();       // same wave wrote and reads: LDS keeps program order
            // ---- gather
            const uint4 *recs = &wrecs[qi * STRIDE];
            for (int kk = 0; kk < kc_pad; kk += kUnroll) {
                uint4 raw[kUnroll][4];
                float w[kUnroll][4];
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    const uint4 rr = recs[2 * (kk + u)];
                    const uint4 ww = recs[2 * (kk + u) + 1];
                    const uint32_t o[4] = {rr.x, rr.y, rr.z, rr.w};
                    w[u][0] = __uint_as_float(ww.x); w[u][1] = __uint_as_float(ww.y);
                    w[u][2] = __uint_as_float(ww.z); w[u][3] = __uint_as_float(ww.w);
                    // every lane group of the wave is at the same sample index -> same level
                    const bool cached = (__builtin_amdgcn_readfirstlane(o[0]) & kCachedBit) != 0;
                    if (cached) {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            raw[u][c] = cache[((o[c] & (kCachedBit - 1)) + lane_off) >> 4];
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) raw[u][c] = buffer_load16(rsrc, o[c] + lane_off);
                    }
                }
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float v[VEC];
                        V::unpack(raw[u][c], v);
#pragma unroll
                        for (int i = 0; i < VEC; ++i) acc[i] = fmaf(w[u][c], v[i], acc[i]);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();       // records are overwritten by the next chunk
        }
        if (q_ok) {
            T *o = out + (((int64_t)bc.b * d.Nq + q) * d.H + bc.h) * d.D + lig * VEC;
            *reinterpret_cast<uint4 *>(o) = V::pack(acc);
        }
    }
}

template <typename T, int LPI>
hipError_t launch(const void *value, const int64_t *shapes, const int64_t *start, const void *loc,
                  const void *attn, void *out, Dims d, int q_per_block, hipStream_t st)
{
    constexpr size_t lds = ((size_t)kMaxCacheLevels + 1 + kCacheBytes / 16 +
                            (size_t)(kThreads / 64) * WaveTile<LPI>::WAVE_RECS) * 16;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static bool configured = false;          // one-time attribute of the kernel object, not library state
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fwd_cached<T, LPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured = true;
    }
    d.q_tiles = (d.Nq + q_per_block - 1) / q_per_block;
    const int64_t blocks = (int64_t)d.B * d.q_tiles * d.H;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((msda_fwd_cached<T, LPI>), dim3((unsigned)blocks), dim3(kThreads), lds, st,
                       (const T *)value, shapes, start, (const T *)loc, (const T *)attn, (T *)out, d, q_per_block,
                       getenv("MMFS_FWD_CACHE_BYTES") ? atoi(getenv("MMFS_FWD_CACHE_BYTES")) : kCacheBytes);
    return hipGetLastError();
}

template <typename T>
hipError_t dispatch(const void *value, const int64_t *shapes, const int64_t *start, const void *loc,
                    const void *attn, void *out, const Dims &d, int qpb, hipStream_t st)
{
    constexpr int VEC = 16 / (int)sizeof(T);
    switch (d.D / VEC) {
#define MMFS_CASE(n) case n: return launch<T, n>(value, shapes, start, loc, attn, out, d, qpb, st);
        MMFS_CASE(4) MMFS_CASE(8) MMFS_CASE(16) MMFS_CASE(32)
#undef MMFS_CASE
        default: return hipErrorNotSupported;
    }
}

}  // namespace

// Is the LDS-cached forward applicable and worth it?  Host-side knowledge only: dims.
bool forward_cached_applicable(int dtype, const Dims &d, int *q_per_block)
{
    if (dtype < 0 || dtype > 2) return false;
    const int es = dtype == 0 ? 4 : 2;
    const int vec = 16 / es;
    if (d.D % vec) return false;
    const int lpi = d.D / vec;
    if (lpi != 4 && lpi != 8 && lpi != 16 && lpi != 32) return false;
    if (d.L < 2 || d.L > kMaxCacheLevels) return false;            // one level: nothing "coarse"
    if ((int64_t)d.S * d.H * d.D * es >= (int64_t)kCachedBit) return false;   // bit 30 marks cache offsets
    if (d.D * es > kCacheBytes / 8) return false;
    // the fill (<= 88 KiB per workgroup) must be amortised over enough queries, and there must be
    // enough workgroups to fill 256 CUs
    const int qpp = kThreads / lpi;                 // queries per sweep of the 16 waves
    int qpb = qpp * 4;
    if (const char *e = getenv("MMFS_FWD_QPB")) qpb = std::max(qpp, atoi(e) / qpp * qpp);   // tuning knob
    if (d.Nq < qpb) return false;
    if ((int64_t)d.B * d.H * ((d.Nq + qpb - 1) / qpb) < 256) return false;
    // Measured on MI355X at the north-star shape: 173 us with levels 16x16
    // and 8x8 in LDS vs 166 us for the plain kernel (202 us for this structure with an empty
    // cache) -- the forward is not purely load-path bound, and the long-lived 16-wave workgroup
    // costs more than the LDS reads save.  Kept as an opt-in (MMFS_FWD_CACHE=1) until it wins.
    const char *on = getenv("MMFS_FWD_CACHE");
    if (!on || atoi(on) == 0) return false;
    *q_per_block = qpb;
    return true;
}

hipError_t forward_cached(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                          const void *loc, const void *attn, void *out, const Dims &d, int q_per_block,
                          hipStream_t st)
{
    switch (dtype) {
        case 0: return dispatch<float>(value, shapes, start, loc, attn, out, d, q_per_block, st);
        case 1: return dispatch<half_t>(value, shapes, start, loc, attn, out, d, q_per_block, st);
        case 2: return dispatch<bf16_t>(value, shapes, start, loc, attn, out, d, q_per_block, st);
        default: return hipErrorNotSupported;
    }
}

}  // namespace mmfs
