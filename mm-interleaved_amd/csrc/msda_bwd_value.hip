// msda_bwd_value.hip -- grad_value of multi-scale deformable attention, pixel-stationary.
//
// The reference scatters grad_value with one float atomicAdd per (sample, corner,
// channel) (mm_interleaved/models/utils/ops/src/cuda/ms_deform_im2col_cuda.cuh:128-155).
// On MI355X global float atomics retire at ~330 G adds/s for the whole chip -- one
// lane-add per L2 channel per clock, independent of footprint and scope -- and LDS
// float atomics (ds_add_f32) at 0.33 lane-adds/clk/CU, while LDS *integer* atomics run
// at ~4.5 lane-ops/clk/CU (tools/ubench/, gpurun_out logs quoted in DESIGN.md).  The
// 2.1e9 adds of the north-star shape cost 6.1 ms that way.  So this kernel turns the
// scatter into a gather:
//
//   * a workgroup owns a rectangular tile of one level's pixels for one (batch, head):
//     every pixel of grad_value has exactly ONE owner, so it is written once, with a
//     plain store, directly in the storage dtype (fp32 accumulation in registers,
//     rounded once at the end == the reference's "accumulate in fp32, cast at the end",
//     ms_deform_attn_cuda.cu:122-165).  No fp32 buffer, no memset, no cast pass;
//   * the workgroup scans the level's sampling locations of its (b, h) (coalesced
//     16-byte reads), re-derives the taps and counting-sorts the contributions that land
//     in its tile by pixel, in LDS, with integer atomics: count -> prefix -> scatter of
//     {query, weight} records;
//   * then the lanes of a wave own the D channels of one pixel: walk the pixel's record
//     run (LDS broadcast reads), gather the grad_out rows (coalesced D*sizeof(T) bytes,
//     L2-resident: the head's rows of one sample), FMA into registers, store the row.
//
// Tiles are planned on the device from the level table (it lives in device memory, as
// in the reference API), identically by every workgroup; the host only supplies an upper
// bound on the tile count.  Records that do not fit the LDS list are handled in rounds
// over pixel ranges (re-scan), and a single pixel that alone overflows it in rounds
// over query ranges, so any distribution of sampling locations is handled.
#include "msda_device.h"
#include "msda_launch.h"

namespace mmfs {

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kMaxTilePx = 1024;        // pixels per tile (counter arrays: 2 x 4 KiB)
constexpr int kListCap = 6144;          // {q, weight} records per round (48 KiB)
constexpr int kUnroll = 8;              // records in flight per lane group

struct TileParams {
    int tiles_bound;   // host upper bound on tiles per (b, h) slice
    int nt_min;        // minimum tiles per level (load balance)
};

struct Tile {
    int level, Hl, Wl, lstart;
    int ya, yb, xa, xb;
    bool valid;
};

// Every workgroup derives the same plan from the level table.
__device__ Tile plan_tile(const int64_t *__restrict__ shapes, const int64_t *__restrict__ start,
                          int L, int t, int nt_min)
{
    Tile r;
    r.valid = false;
    r.level = r.Hl = r.Wl = r.lstart = r.ya = r.yb = r.xa = r.xb = 0;
    for (int l = 0; l < L; ++l) {
        const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
        const int px = Hl * Wl;
        if (px <= 0) continue;
        int nt = max(nt_min, (px + kMaxTilePx - 1) / kMaxTilePx);
        nt = min(nt, px);
        const int tpx = (px + nt - 1) / nt;                     // <= kMaxTilePx
        int R, C;
        if (Wl <= tpx) { R = tpx / Wl; C = Wl; } else { R = 1; C = tpx; }
        const int ny = (Hl + R - 1) / R, nx = (Wl + C - 1) / C;
        const int n = ny * nx;
        if (t < n) {
            const int ty = t / nx, tx = t % nx;
            r.level = l; r.Hl = Hl; r.Wl = Wl; r.lstart = (int)start[l];
            r.ya = ty * R; r.yb = min(Hl, r.ya + R);
            r.xa = tx * C; r.xb = min(Wl, r.xa + C);
            r.valid = true;
            return r;
        }
        t -= n;
    }
    return r;
}

// Exclusive prefix sum over a[0..n) (n <= kMaxTilePx), total left in a[n].
__device__ void block_exclusive_scan(uint32_t *a, int n, uint32_t *wave_tot)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int PER = kMaxTilePx / kThreads;                   // 4 consecutive entries per thread
    uint32_t v[PER], sum = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int idx = tid * PER + i;
        v[i] = idx < n ? a[idx] : 0u;
        sum += v[i];
    }
    uint32_t inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w) base += wave_tot[w];
    uint32_t run = base + inc - sum;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int idx = tid * PER + i;
        if (idx < n) a[idx] = run;
        run += v[i];
    }
    if (tid == kThreads - 1) a[n] = run;
    __syncthreads();
}

enum ScanMode { kCount = 0, kScatter = 1 };

// Scan the sampling locations of queries [q_lo, q_hi) at the tile's level and, for every
// tap corner that lands on a tile pixel in [p_lo, p_hi):
//   kCount  : off[pixel] += 1
//   kScatter: list[off[pixel] - base + cur[pixel]++] = {q, bilinear weight * attention}
template <typename T, int MODE>
__device__ __forceinline__ void scan_samples(const T *__restrict__ loc, const T *__restrict__ attn,
                                             const Dims &d, const Tile &tl, int b, int h,
                                             int q_lo, int q_hi, int p_lo, int p_hi, uint32_t base,
                                             uint32_t *off, uint32_t *cur, uint2 *list)
{
    const int tw = tl.xb - tl.xa;
    for (int q = q_lo + (int)threadIdx.x; q < q_hi; q += kThreads) {
        const int64_t s0 = ((((int64_t)b * d.Nq + q) * d.H + h) * d.L + tl.level) * d.P;
        for (int p = 0; p < d.P; ++p) {
            const float lx = to_f32(loc[2 * (s0 + p)]), ly = to_f32(loc[2 * (s0 + p) + 1]);
            const float y = ly * (float)tl.Hl - 0.5f, x = lx * (float)tl.Wl - 0.5f;
            const bool inside = (y > -1.f) && (x > -1.f) && (y < (float)tl.Hl) && (x < (float)tl.Wl);
            if (!inside) continue;
            const float yf = floorf(y), xf = floorf(x);
            const int y0 = (int)yf, x0 = (int)xf;
            // quick reject: the 2x2 footprint misses the tile
            if (y0 + 1 < tl.ya || y0 >= tl.yb || x0 + 1 < tl.xa || x0 >= tl.xb) continue;
            const float fy = y - yf, fx = x - xf;
            float a = 0.f;
            if (MODE == kScatter) a = to_f32(attn[s0 + p]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int yy = y0 + (c >> 1), xx = x0 + (c & 1);
                // a corner outside the map is also outside every tile
                if (yy < tl.ya || yy >= tl.yb || xx < tl.xa || xx >= tl.xb) continue;
                const int pl = (yy - tl.ya) * tw + (xx - tl.xa);
                if (pl < p_lo || pl >= p_hi) continue;
                if (MODE == kCount) {
                    atomicAdd(&off[pl], 1u);
                } else {
                    const float wy = (c >> 1) ? fy : 1.f - fy, wx = (c & 1) ? fx : 1.f - fx;
                    const uint32_t slot = off[pl] - base + atomicAdd(&cur[pl], 1u);
                    list[slot] = make_uint2((uint32_t)q, __float_as_uint(wy * wx * a));
                }
            }
        }
    }
}

template <typename T, int CPL> struct ChanVec;          // CPL channels of T <-> floats
template <typename T, int CPL> struct ChanVec {
    static __device__ __forceinline__ void load(const T *p, float (&o)[CPL]) {
        T tmp[CPL];
        __builtin_memcpy(tmp, __builtin_assume_aligned(p, sizeof(T) * CPL), sizeof(T) * CPL);
#pragma unroll
        for (int i = 0; i < CPL; ++i) o[i] = to_f32(tmp[i]);
    }
    static __device__ __forceinline__ void store(T *p, const float (&v)[CPL]) {
        T tmp[CPL];
#pragma unroll
        for (int i = 0; i < CPL; ++i) tmp[i] = (T)v[i];
        __builtin_memcpy(__builtin_assume_aligned(p, sizeof(T) * CPL), tmp, sizeof(T) * CPL);
    }
};

// acc += sum over records first, first+step, ... (< end) of weight * grad_out[q, h, my channels]
template <typename T, int CPL>
__device__ __forceinline__ void reduce_run(const uint2 *__restrict__ list, int first, int end, int step,
                                           const T *__restrict__ gslice, int64_t HD, float (&acc)[CPL])
{
    for (int e = first; e < end; e += kUnroll * step) {
        float g[kUnroll][CPL], w[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int ee = e + u * step;
            const bool ok = ee < end;
            const uint2 rec = list[ok ? ee : 0];
            w[u] = ok ? __uint_as_float(rec.y) : 0.f;
            ChanVec<T, CPL>::load(gslice + (int64_t)(ok ? rec.x : 0u) * HD, g[u]);
            if (!ok) {
#pragma unroll
                for (int i = 0; i < CPL; ++i) g[u][i] = 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u)
#pragma unroll
            for (int i = 0; i < CPL; ++i) acc[i] = fmaf(w[u], g[u][i], acc[i]);
    }
}

// LPS lanes own the D = LPS*CPL channels of one pixel; a wave works on 64/LPS pixels.
template <typename T, int LPS, int CPL>
__global__ void __launch_bounds__(kThreads)
msda_bwd_value_tiled(const int64_t *__restrict__ shapes, const int64_t *__restrict__ start,
                     const T *__restrict__ loc, const T *__restrict__ attn,
                     const T *__restrict__ grad_out, T *__restrict__ grad_value,
                     const Dims d, const TileParams tp)
{
    constexpr int GPW = 64 / LPS;                   // pixel groups per wave
    constexpr int GROUPS = kWaves * GPW;            // pixel groups per workgroup
    constexpr int D = LPS * CPL;
    static_assert(GROUPS * D * 4 <= kListCap * 8, "combine scratch must fit the record list");
    __shared__ uint32_t off[kMaxTilePx + 1];
    __shared__ uint32_t cur[kMaxTilePx];
    __shared__ uint2 list[kListCap];
    __shared__ uint32_t wave_tot[kWaves];

    const int bid = blockIdx.x;
    const int h = bid % d.H;
    const int t = (bid / d.H) % tp.tiles_bound;
    const int b = (bid / d.H) / tp.tiles_bound;
    const Tile tl = plan_tile(shapes, start, d.L, t, tp.nt_min);
    if (!tl.valid) return;

    const int tid = threadIdx.x;
    const int tw = tl.xb - tl.xa;
    const int npx = (tl.yb - tl.ya) * tw;
    const int gid = tid / LPS, lig = tid % LPS;     // pixel group of this lane, lane in group
    const int64_t HD = (int64_t)d.H * d.D;
    const T *gslice = grad_out + ((int64_t)b * d.Nq * d.H + h) * d.D + lig * CPL;
    T *vslice = grad_value + ((int64_t)b * d.S * d.H + h) * d.D + lig * CPL;

    for (int i = tid; i < npx; i += kThreads) off[i] = 0u;
    __syncthreads();
    scan_samples<T, kCount>(loc, attn, d, tl, b, h, 0, d.Nq, 0, npx, 0u, off, cur, list);
    __syncthreads();
    block_exclusive_scan(off, npx, wave_tot);

    int p_lo = 0;
    while (p_lo < npx) {
        const uint32_t base = off[p_lo];
        // largest p_hi in (p_lo, npx] whose records fit the list
        int lo = p_lo, hi = npx;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (off[mid] - base <= (uint32_t)kListCap) lo = mid; else hi = mid - 1;
        }
        const int p_hi = lo;
        if (p_hi > p_lo) {
            // ---- a range of pixels whose records fit: sort them, one pixel per lane group
            const uint32_t nrec = off[p_hi] - base;
            if (nrec) {
                for (int i = p_lo + tid; i < p_hi; i += kThreads) cur[i] = 0u;
                __syncthreads();
                scan_samples<T, kScatter>(loc, attn, d, tl, b, h, 0, d.Nq, p_lo, p_hi, base, off, cur, list);
                __syncthreads();
            }
            for (int p0 = p_lo; p0 < p_hi; p0 += GROUPS) {
                const int p = p0 + gid;
                const bool act = p < p_hi;
                const int first = act ? (int)(off[p] - base) : 0;
                int n = act ? (int)(off[p + 1] - off[p]) : 0;
                // groups of one wave iterate together: run to the longest list in the wave
                int nmax = n;
                if (GPW > 1) {
#pragma unroll
                    for (int o = LPS; o < 64; o <<= 1) nmax = max(nmax, __shfl_xor(nmax, o, 64));
                }
                float acc[CPL];
#pragma unroll
                for (int i = 0; i < CPL; ++i) acc[i] = 0.f;
                if (GPW > 1) {
                    // pad the shorter lists with zero-weight reads of their first record
                    for (int e = 0; e < nmax; e += kUnroll) {
                        float g[kUnroll][CPL], w[kUnroll];
#pragma unroll
                        for (int u = 0; u < kUnroll; ++u) {
                            const bool ok = e + u < n;
                            const uint2 rec = list[ok ? first + e + u : 0];
                            w[u] = ok ? __uint_as_float(rec.y) : 0.f;
                            ChanVec<T, CPL>::load(gslice + (int64_t)(ok ? rec.x : 0u) * HD, g[u]);
                            if (!ok) {
#pragma unroll
                                for (int i = 0; i < CPL; ++i) g[u][i] = 0.f;
                            }
                        }
#pragma unroll
                        for (int u = 0; u < kUnroll; ++u)
#pragma unroll
                            for (int i = 0; i < CPL; ++i) acc[i] = fmaf(w[u], g[u][i], acc[i]);
                    }
                } else {
                    reduce_run<T, CPL>(list, first, first + n, 1, gslice, HD, acc);
                }
                if (act) {
                    const int pg = tl.lstart + (tl.ya + p / tw) * tl.Wl + tl.xa + p % tw;
                    ChanVec<T, CPL>::store(vslice + (int64_t)pg * HD, acc);
                }
            }
            __syncthreads();                        // list and cur are reused by the next round
            p_lo = p_hi;
        } else {
            // ---- one pixel with more records than the list holds: rounds over query ranges.
            // A sample puts at most one corner on a given pixel, so qw queries give <= qw*P records.
            const int qw = max(1, kListCap / max(1, d.P));
            float acc[CPL];
#pragma unroll
            for (int i = 0; i < CPL; ++i) acc[i] = 0.f;
            for (int q0 = 0; q0 < d.Nq; q0 += qw) {
                if (tid == 0) cur[p_lo] = 0u;
                __syncthreads();
                // records are placed at cur[] alone: pass base = off[p_lo] so slot = cur
                scan_samples<T, kScatter>(loc, attn, d, tl, b, h, q0, min(d.Nq, q0 + qw), p_lo, p_lo + 1,
                                          off[p_lo], off, cur, list);
                __syncthreads();
                const int n = (int)cur[p_lo];
                // all lane groups share this pixel's records: group g takes g, g+GROUPS, ...
                reduce_run<T, CPL>(list, gid, n, GROUPS, gslice, HD, acc);
                __syncthreads();
            }
            // combine the GROUPS partial rows through LDS (the list is free now)
            float *scratch = reinterpret_cast<float *>(list);
#pragma unroll
            for (int i = 0; i < CPL; ++i) scratch[gid * D + lig * CPL + i] = acc[i];
            __syncthreads();
            if (gid == 0) {
                float tot[CPL];
#pragma unroll
                for (int i = 0; i < CPL; ++i) {
                    tot[i] = 0.f;
                    for (int g2 = 0; g2 < GROUPS; ++g2) tot[i] += scratch[g2 * D + lig * CPL + i];
                }
                const int p = p_lo;
                const int pg = tl.lstart + (tl.ya + p / tw) * tl.Wl + tl.xa + p % tw;
                ChanVec<T, CPL>::store(vslice + (int64_t)pg * HD, tot);
            }
            __syncthreads();
            p_lo += 1;
        }
    }
}

// lanes per pixel / channels per lane for a head width, or false
bool lane_map(int dtype, int D, int *lps, int *cpl)
{
    const int es = dtype == 0 ? 4 : 2;
    if (dtype != 0 && dtype != 1 && dtype != 2) return false;
    if ((D * es) % 4) return false;
    const int words = D * es / 4;                      // 4-byte words per pixel row
    int l = words >= 64 ? 64 : words;
    if (l < 4 || (l & (l - 1))) return false;
    if (D % l) return false;
    const int c = D / l;
    if (c * es > 16 || (c & (c - 1))) return false;
    *lps = l; *cpl = c;
    return true;
}

TileParams make_params(const Dims &d)
{
    TileParams tp;
    // aim at >= 8 tiles per CU over the whole launch (256 CUs), at least 1 per level
    const int64_t slices = (int64_t)d.B * d.H * std::max(1, d.L);
    int nt = (int)std::min<int64_t>(64, std::max<int64_t>(1, (2048 + slices - 1) / slices));
    tp.nt_min = nt;
    // tiles per level <= 2*nt_l + 1 with nt_l <= nt_min + px_l/kMaxTilePx + 1 (see plan_tile)
    const int64_t bound = 2LL * d.L * (nt + 1) + 2LL * ((d.S + kMaxTilePx - 1) / kMaxTilePx) + d.L;
    tp.tiles_bound = (int)std::min<int64_t>(bound, 0x3fffffff);
    return tp;
}

template <typename T, int LPS, int CPL>
hipError_t launch(const int64_t *shapes, const int64_t *start, const void *loc, const void *attn,
                  const void *go, void *gv, const Dims &d, hipStream_t st)
{
    const TileParams tp = make_params(d);
    const int64_t blocks = (int64_t)d.B * d.H * tp.tiles_bound;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((msda_bwd_value_tiled<T, LPS, CPL>), dim3((unsigned)blocks), dim3(kThreads), 0, st,
                       shapes, start, (const T *)loc, (const T *)attn, (const T *)go, (T *)gv, d, tp);
    return hipGetLastError();
}

template <typename T>
hipError_t dispatch(int lps, int cpl, const int64_t *shapes, const int64_t *start, const void *loc,
                    const void *attn, const void *go, void *gv, const Dims &d, hipStream_t st)
{
    // 16-bit storage: 2 channels per 4-byte word, fp32: 1; wider per-lane vectors once D > 64 words
    constexpr int C0 = 4 / (int)sizeof(T);
#define MMFS_CASE(L_, C_) if (lps == L_ && cpl == C_) return launch<T, L_, C_>(shapes, start, loc, attn, go, gv, d, st);
    MMFS_CASE(64, C0) MMFS_CASE(64, 2 * C0) MMFS_CASE(64, 4 * C0)
    MMFS_CASE(32, C0) MMFS_CASE(16, C0) MMFS_CASE(8, C0) MMFS_CASE(4, C0)
#undef MMFS_CASE
    return hipErrorInvalidValue;
}

}  // namespace

bool bwd_value_tiled_supported(int dtype, const Dims &d)
{
    int lps, cpl;
    if (!lane_map(dtype, d.D, &lps, &cpl)) return false;
    if (d.P > kListCap) return false;
    const TileParams tp = make_params(d);
    return (int64_t)d.B * d.H * tp.tiles_bound <= 0x7fffffffLL;
}

hipError_t backward_value_tiled(int dtype, const int64_t *shapes, const int64_t *start,
                                const void *loc, const void *attn, const void *grad_out,
                                void *grad_value, const Dims &d, hipStream_t st)
{
    int lps, cpl;
    if (!lane_map(dtype, d.D, &lps, &cpl)) return hipErrorInvalidValue;
    switch (dtype) {
        case 0: return dispatch<float>(lps, cpl, shapes, start, loc, attn, grad_out, grad_value, d, st);
        case 1: return dispatch<half_t>(lps, cpl, shapes, start, loc, attn, grad_out, grad_value, d, st);
        case 2: return dispatch<bf16_t>(lps, cpl, shapes, start, loc, attn, grad_out, grad_value, d, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mmfs
