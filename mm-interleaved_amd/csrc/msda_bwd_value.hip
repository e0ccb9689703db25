// msda_bwd_value.hip -- grad_value of multi-scale deformable attention, pixel-stationary.
//
// The reference scatters grad_value with one float atomicAdd per (sample, corner,
// channel) (mm_interleaved/models/utils/ops/src/cuda/ms_deform_im2col_cuda.cuh:128-155).
// On MI355X global float atomics retire at ~330 G adds/s for the whole chip -- one
// lane-add per L2 channel per clock, independent of footprint and scope -- and LDS
// float atomics (ds_add_f32) at 0.33 lane-adds/clk/CU, while LDS *integer* atomics run
// at ~4.5 lane-ops/clk/CU (tools/ubench/, logs quoted in DESIGN.md).  The 2.1e9 adds of
// the north-star shape cost 6.1 ms that way.  So this kernel turns the scatter into a
// gather:
//
//   * a workgroup owns a rectangular tile of one level's pixels for one (batch, head):
//     every pixel of grad_value has exactly ONE owner, so it is written once, with a
//     plain store, directly in the storage dtype (fp32 accumulation in registers,
//     rounded once at the end == the reference's "accumulate in fp32, cast at the end",
//     ms_deform_attn_cuda.cu:122-165).  No fp32 image, no memset, no cast pass;
//   * the workgroup scans the level's sampling locations of its (b, h) twice: the first
//     scan counts, per tile pixel, the tap corners that land on it (LDS integer atomics),
//     a prefix sum turns counts into offsets, the second scan writes each contribution's
//     {query, weight} record to its sorted position in a scratch list in global memory
//     (exactly Nq*P*4 records per (b, h, level): no capacity limit, no overflow rounds);
//   * a second kernel of many small workgroups then streams over the pixels in (b, pixel, h)
//     order: the lanes of a group own the D channels of one pixel, read the pixel's run of
//     records (coalesced, one record per lane, handed round with wave shuffles), gather the
//     grad_out rows through a buffer descriptor (D*sizeof(T) contiguous bytes, L2-resident:
//     one head's rows of one sample), FMA into registers and store the row.  Runs longer than
//     16 batches (coarse levels, hot spots) are finished by all groups of the workgroup
//     together, combined through LDS.
//
// The sampling locations arrive as [B, Nq, H, L, P, 2]: for one (b, h, level) the P*2
// scalars of consecutive queries are H*L*P*2 elements apart, so scanning them straight
// from there wastes 7/8 of every cache line.  A small pre-pass re-packs loc/attn into
// [B, H, L, Nq, P(,2)] in the caller-provided workspace; the scans then read 16
// contiguous bytes per lane.
//
// Tiles are planned on the device from the level table (it lives in device memory, as in the
// reference API) by a one-thread kernel into a table; the host only supplies an upper bound on
// the tile count.
#include "msda_device.h"
#include "msda_env.h"
#include "msda_launch.h"
#include <type_traits>
#include <cstdlib>

namespace mmfs {

namespace {

constexpr int kThreads = 1024;          // 16 waves per workgroup
constexpr int kWaves = kThreads / 64;
constexpr int kMaxTilePx = 4096;        // pixels per tile (two counter arrays of 16 KiB)
#ifndef MMFS_VAL_UNROLL
#define MMFS_VAL_UNROLL 16
#endif
constexpr int kUnroll = MMFS_VAL_UNROLL;   // grad_out rows in flight per lane group
constexpr int kScanUnroll = 4;          // queries in flight per thread while scanning

struct TileParams {
    int tiles_bound;   // host upper bound on tiles per (b, h) slice
    int nt_min;        // minimum tiles per level (load balance)
};

struct Tile {
    int level, Hl, Wl, lstart;
    int ya, yb, xa, xb;
    bool valid;
};

// The plan is computed ONCE per launch by a single thread into a table in the workspace (the
// level table is device memory, so the host cannot do it): the sort workgroups then fetch their
// tile with one load and surplus workgroups exit at once.  Re-deriving the plan in every
// workgroup cost ~3 us per workgroup at L = 4 and grows with L (L = 90 for 30 images x 3 levels).
struct TileTable {
    int n_tiles;
    int pad[7];
    Tile tile[1];          // [tiles_bound]
};

__global__ void plan_tiles_kernel(const int64_t *__restrict__ shapes, const int64_t *__restrict__ start,
                                  int L, int nt_min, int cap,
                                  TileTable *__restrict__ table)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int n = 0;
    for (int l = 0; l < L && n < cap; ++l) {
        const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
        const int px = Hl * Wl;
        if (px <= 0) continue;
        int nt = max(nt_min, (px + kMaxTilePx - 1) / kMaxTilePx);
        nt = min(nt, px);
        const int tpx = (px + nt - 1) / nt;
        int R, C;
        if (Wl <= tpx) { R = tpx / Wl; C = Wl; } else { R = 1; C = tpx; }
        const int lstart = (int)start[l];
        for (int ya = 0; ya < Hl && n < cap; ya += R)
            for (int xa = 0; xa < Wl && n < cap; xa += C) {
                Tile t;
                t.level = l; t.Hl = Hl; t.Wl = Wl; t.lstart = lstart;
                t.ya = ya; t.yb = min(Hl, ya + R); t.xa = xa; t.xb = min(Wl, xa + C);
                t.valid = true;
                table->tile[n++] = t;
            }
    }
    table->n_tiles = n;
}

// Exclusive prefix sum over a[0..n) (n <= kMaxTilePx), total left in a[n].
__device__ void block_exclusive_scan(uint32_t *a, int n, uint32_t *wave_tot)
{
    constexpr int PER = kMaxTilePx / kThreads;                   // consecutive counters per thread
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t c[PER], v = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        c[i] = tid * PER + i < n ? a[tid * PER + i] : 0u;
        v += c[i];
    }
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w) base += wave_tot[w];
    uint32_t run = base + inc - v;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        if (tid * PER + i < n) a[tid * PER + i] = run;
        run += c[i];
    }
    if (tid == kThreads - 1) a[n] = run;
    __syncthreads();
}

enum ScanMode { kCount = 0, kScatter = 1 };

// One sample against the tile: for every tap corner on a tile pixel
//   kCount  : off[pixel] += 1
//   kScatter: list[off[pixel] + cur[pixel]++] = {q, bilinear weight * attention}
template <int MODE>
__device__ __forceinline__ void visit_sample(float lx, float ly, float a, int q, const Tile &tl, int tw,
                                             uint32_t *off, uint32_t *cur, uint2 *__restrict__ list)
{
    const float y = ly * (float)tl.Hl - 0.5f, x = lx * (float)tl.Wl - 0.5f;
    const bool inside = (y > -1.f) && (x > -1.f) && (y < (float)tl.Hl) && (x < (float)tl.Wl);
    if (!inside) return;
    const float yf = floorf(y), xf = floorf(x);
    const int y0 = (int)yf, x0 = (int)xf;
    // quick reject: the 2x2 footprint misses the tile
    if (y0 + 1 < tl.ya || y0 >= tl.yb || x0 + 1 < tl.xa || x0 >= tl.xb) return;
    const float fy = y - yf, fx = x - xf;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int yy = y0 + (c >> 1), xx = x0 + (c & 1);
        // a corner outside the map is also outside every tile
        if (yy < tl.ya || yy >= tl.yb || xx < tl.xa || xx >= tl.xb) continue;
        const int pl = (yy - tl.ya) * tw + (xx - tl.xa);
        if (MODE == kCount) {
            atomicAdd(&off[pl], 1u);
        } else {
            const float wy = (c >> 1) ? fy : 1.f - fy, wx = (c & 1) ? fx : 1.f - fx;
            const uint32_t slot = off[pl] + atomicAdd(&cur[pl], 1u);
            // plain store: the 8-byte records merge in this XCD's L2 and are read back by this CU
            // (non-temporal stores measured 1.8x slower for the whole kernel: no write combining)
            list[slot] = make_uint2((uint32_t)q, __float_as_uint(wy * wx * a));
        }
    }
}

// Scan the sampling locations of all queries at the tile's level (re-packed copies).
// NV > 0: the P samples of one (b,h,level,q) are NV 16-byte vectors of locations (and NV
//         8-byte vectors of weights); kScanUnroll queries are loaded before any is used.
// NV = 0: any P, scalar loads.
template <typename T, int MODE, int NV>
__device__ __forceinline__ void scan_samples(const T *__restrict__ loc, const T *__restrict__ attn,
                                             const Dims &d, const Tile &tl, int b, int h,
                                             uint32_t *off, uint32_t *cur, uint2 *__restrict__ list)
{
    const int tw = tl.xb - tl.xa;
    const int64_t qstride = d.P;                               // samples between consecutive queries
    const int64_t s_first = ((((int64_t)b * d.H + h) * d.L + tl.level) * d.Nq) * d.P;
    if (NV == 0) {
        for (int q = (int)threadIdx.x; q < d.Nq; q += kThreads) {
            const int64_t s0 = s_first + q * qstride;
            for (int p = 0; p < d.P; ++p)
                visit_sample<MODE>(to_f32(loc[2 * (s0 + p)]), to_f32(loc[2 * (s0 + p) + 1]),
                                   MODE == kScatter ? to_f32(attn[s0 + p]) : 0.f, q, tl, tw, off, cur, list);
        }
        return;
    }
    typedef Vec16<T> V;
    constexpr int VEC = V::N;                                  // location scalars per 16 bytes
    constexpr int NVV = NV > 0 ? NV : 1;
    for (int q0 = (int)threadIdx.x; q0 < d.Nq; q0 += kThreads * kScanUnroll) {
        uint4 lraw[kScanUnroll][NVV];
        uint2 araw[kScanUnroll][NVV];
#pragma unroll
        for (int u = 0; u < kScanUnroll; ++u) {
            const int q = q0 + u * kThreads;
            const int64_t s0 = s_first + (int64_t)min(q, d.Nq - 1) * qstride;
#pragma unroll
            for (int v = 0; v < NVV; ++v) {
                lraw[u][v] = reinterpret_cast<const uint4 *>(loc + 2 * s0)[v];
                if (MODE == kScatter) araw[u][v] = reinterpret_cast<const uint2 *>(attn + s0)[v];
                else araw[u][v] = make_uint2(0u, 0u);
            }
        }
#pragma unroll
        for (int u = 0; u < kScanUnroll; ++u) {
            const int q = q0 + u * kThreads;
            if (q >= d.Nq) break;
#pragma unroll
            for (int v = 0; v < NVV; ++v) {
                float l[VEC], a[VEC];
                V::unpack(lraw[u][v], l);
                V::unpack(make_uint4(araw[u][v].x, araw[u][v].y, 0u, 0u), a);   // first VEC/2 valid
#pragma unroll
                for (int i = 0; i < VEC / 2; ++i)
                    visit_sample<MODE>(l[2 * i], l[2 * i + 1], a[i], q, tl, tw, off, cur, list);
            }
        }
    }
}

// One lane's view of a batch of LPS records: the record this lane fetched (coalesced
// LPS*8-byte read) turned into a row offset ("outside" past the end of the run) and a weight.
struct BatchRec { uint32_t off; float w; };

__device__ __forceinline__ BatchRec fetch_batch(const uint2 *__restrict__ list, int e, int end, uint32_t row_bytes)
{
    BatchRec r;
    r.off = kOobOffset; r.w = 0.f;
    if (e < end) {
        const uint2 rec = list[e];
        r.off = rec.x * row_bytes;
        r.w = __uint_as_float(rec.y);
    }
    return r;
}

// acc += sum over one batch of LPS records.  Every lane fetched one record of the batch; the
// group needs each record on ALL its lanes (they read one row together).  The records go through
// a small per-group LDS slot: one ds_write_b64 per lane, then a broadcast ds_read_b64 per record.
// (Measured alternatives on MI355X at the north-star shape, this form = 350 us: ds_bpermute
// hand-off 343 us; DPP rotation, every lane reading a different row at a time, 772 us; every lane
// loading the batch's records itself 446 us; one wave per pixel with the records on the scalar
// side (s_load + SGPR offsets) 493 us -- too few bytes in flight per wave.)
// All LPS row reads are requested before the first is used.
template <typename T, int LPS, bool BUF>
__device__ __forceinline__ void consume_batch(const BatchRec &mine, int lig, uint2 *__restrict__ slot,
                                              const T *__restrict__ gslice, int64_t HD,
                                              __amdgpu_buffer_rsrc_t rsrc, uint32_t row_bytes,
                                              uint32_t lane_off, float (&acc)[Vec16<T>::N])
{
    typedef Vec16<T> V;
    constexpr int U = LPS < kUnroll ? LPS : kUnroll;
    slot[lig] = make_uint2(mine.off, __float_as_uint(mine.w));
    __builtin_amdgcn_wave_barrier();            // same wave writes and reads: LDS keeps program order
#pragma unroll
    for (int u0 = 0; u0 < LPS; u0 += U) {
        uint4 raw[U];
        float w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint2 rec = slot[u0 + u];
            w[u] = __uint_as_float(rec.y);
            if (BUF) {
                raw[u] = buffer_load16(rsrc, rec.x + lane_off);
            } else {
                const bool ok = rec.x != kOobOffset;
                raw[u] = *reinterpret_cast<const uint4 *>(gslice + (int64_t)(ok ? rec.x / row_bytes : 0u) * HD);
                if (!ok) raw[u] = make_uint4(0u, 0u, 0u, 0u);   // 0 * Inf must not leak
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float g[V::N];
            V::unpack(raw[u], g);
#pragma unroll
            for (int i = 0; i < V::N; ++i) acc[i] = fmaf(w[u], g[i], acc[i]);
        }
    }
    __builtin_amdgcn_wave_barrier();            // the slot is rewritten by the next batch
}

// ---------------------------------------------------------------- kernel A: sort
// One 1024-lane workgroup per tile: count -> prefix -> scatter, then publish for every tile
// pixel where its run of records lives: pixtab[(b*H + h)*S + pixel] = {first record, count}.
template <typename T, int NV>
__global__ void __launch_bounds__(kThreads)
msda_bwd_value_sort(const int64_t *__restrict__ shapes, const int64_t *__restrict__ start,
                    const T *__restrict__ loc, const T *__restrict__ attn,
                    uint2 *__restrict__ records, uint32_t *__restrict__ level_cursor,
                    uint2 *__restrict__ pixtab, const TileTable *__restrict__ table,
                    const Dims d, const TileParams tp)
{
    __shared__ uint32_t off[kMaxTilePx + 1];
    __shared__ uint32_t cur[kMaxTilePx];
    __shared__ uint32_t wave_tot[kWaves];
    __shared__ uint32_t region;

    const int bid = blockIdx.x;
    const int h = bid % d.H;
    const int t = (bid / d.H) % tp.tiles_bound;
    const int b = (bid / d.H) / tp.tiles_bound;
    if (t >= table->n_tiles) return;
    const Tile tl = table->tile[t];

    const int tid = threadIdx.x;
    const int tw = tl.xb - tl.xa;
    const int npx = (tl.yb - tl.ya) * tw;

    for (int i = tid; i < npx; i += kThreads) { off[i] = 0u; cur[i] = 0u; }
    __syncthreads();
    scan_samples<T, kCount, NV>(loc, attn, d, tl, b, h, off, cur, nullptr);
    __syncthreads();
    block_exclusive_scan(off, npx, wave_tot);
    const uint32_t total = off[npx];
    // this tile's slice of the (b, h, level) record area: the level's tiles share Nq*P*4 slots
    // (every tap corner lands in exactly one tile)
    const int64_t slot = ((int64_t)b * d.H + h) * d.L + tl.level;
    if (tid == 0) region = total ? atomicAdd(&level_cursor[slot], total) : 0u;
    __syncthreads();
    const int64_t base = slot * ((int64_t)d.Nq * d.P * 4) + region;      // absolute record index
    if (total) scan_samples<T, kScatter, NV>(loc, attn, d, tl, b, h, off, cur, records + base);
    uint2 *tab = pixtab + ((int64_t)b * d.H + h) * d.S;
    for (int p = tid; p < npx; p += kThreads) {
        const int pg = tl.lstart + (tl.ya + p / tw) * tl.Wl + tl.xa + p % tw;
        tab[pg] = make_uint2((uint32_t)(base + off[p]), off[p + 1] - off[p]);
    }
}

// ---------------------------------------------------------------- kernel B: reduce
// Many small workgroups stream over the pixels in (b, pixel, h) order, like the forward streams
// over queries -- the grad_out rows of few (b, h) slices are live in an XCD's L2 at a time.
// A lane group (LPS lanes x 16 B) owns one pixel: it walks the pixel's run of records in
// batches of LPS (one coalesced record per lane, handed round with wave shuffles, all LPS
// grad_out rows requested before the first is used) and stores the row once.
// Runs longer than kOwnBatches batches (coarse levels, hot spots) are finished co-operatively:
// all groups of the workgroup split the remainder and combine through LDS.
constexpr int kRThreads = 256;
#ifndef MMFS_VAL_OWN
#define MMFS_VAL_OWN 16     // measured: 4 -> 473 us, 16 -> 449 us (cfg2); unroll 4/8/16 makes no difference
#endif
constexpr int kOwnBatches = MMFS_VAL_OWN;

template <typename T, int LPS, bool BUF>
__global__ void __launch_bounds__(kRThreads)
msda_bwd_value_reduce(const T *__restrict__ grad_out, T *__restrict__ grad_value,
                      const uint2 *__restrict__ records, const uint2 *__restrict__ pixtab,
                      const Dims d, const int chunks)
{
    typedef Vec16<T> V;
    constexpr int VEC = V::N;
    constexpr int GROUPS = kRThreads / LPS;          // pixels per workgroup
    constexpr int D = LPS * VEC;
    __shared__ uint2 rest[GROUPS];                   // {first record, count} left after phase 1
    __shared__ float scratch[GROUPS * D];
    __shared__ uint2 slots[GROUPS * (LPS + 1)];      // per-group record hand-off (+1: bank skew)

    const int bid = blockIdx.x;
    const int h = bid % d.H;
    const int chunk = (bid / d.H) % chunks;
    const int b = (bid / d.H) / chunks;
    const int tid = threadIdx.x;
    const int gid = tid / LPS, lig = tid % LPS;
    uint2 *slot = slots + gid * (LPS + 1);
    const int pg = chunk * GROUPS + gid;             // pixel on the S axis
    const bool act = pg < d.S;

    const int64_t HD = (int64_t)d.H * d.D;
    const T *gslice = grad_out + ((int64_t)b * d.Nq * d.H + h) * d.D + lig * VEC;
    const uint32_t row_bytes = (uint32_t)(HD * sizeof(T));
    const uint32_t lane_off = (uint32_t)(lig * 16);
    __amdgpu_buffer_rsrc_t rsrc;
    if (BUF) rsrc = make_slab_rsrc(grad_out + ((int64_t)b * d.Nq * d.H + h) * d.D,
                                   ((int64_t)d.Nq * HD - (int64_t)h * d.D) * (int64_t)sizeof(T));

    uint2 run = make_uint2(0u, 0u);
    if (act) run = pixtab[((int64_t)b * d.H + h) * d.S + pg];
    const uint2 *list = records + run.x;
    const int n = (int)run.y;
    const int own = min(n, kOwnBatches * LPS);

    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    // ---- phase 1: the group's own pixel, up to kOwnBatches batches
    int nb = (own + LPS - 1) / LPS;
#pragma unroll
    for (int o = LPS; o < 64; o <<= 1) nb = max(nb, __shfl_xor(nb, o, 64));    // wave-uniform trip count
    BatchRec pre = fetch_batch(list, lig, own, row_bytes);
    for (int j = 0; j < nb; ++j) {
        const BatchRec cur_rec = pre;
        if (j + 1 < nb) pre = fetch_batch(list, (j + 1) * LPS + lig, own, row_bytes);
        consume_batch<T, LPS, BUF>(cur_rec, lig, slot, gslice, HD, rsrc, row_bytes, lane_off, acc);
    }
    // ---- phase 2: long runs, pixel by pixel, all groups together
    if (lig == 0) rest[gid] = make_uint2(run.x + (uint32_t)own, (uint32_t)(n - own));
    __syncthreads();
    for (int g = 0; g < GROUPS; ++g) {
        const uint2 r = rest[g];                      // uniform over the workgroup
        if (r.y == 0u) continue;
        const uint2 *rl = records + r.x;
        const int rn = (int)r.y;
        const int batches = (rn + LPS - 1) / LPS;
        float part[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) part[i] = 0.f;
        for (int j = gid; j < batches; j += GROUPS) {
            const BatchRec br = fetch_batch(rl, j * LPS + lig, rn, row_bytes);
            consume_batch<T, LPS, BUF>(br, lig, slot, gslice, HD, rsrc, row_bytes, lane_off, part);
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) scratch[gid * D + lig * VEC + i] = part[i];
        __syncthreads();
        if (gid == g) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float t2 = 0.f;
                for (int g2 = 0; g2 < GROUPS; ++g2) t2 += scratch[g2 * D + lig * VEC + i];
                acc[i] += t2;
            }
        }
        __syncthreads();
    }
    if (act) {
        T *o = grad_value + (((int64_t)b * d.S + pg) * d.H + h) * d.D + lig * VEC;
        store16_stream(o, V::pack(acc));
    }
}

// [B, Nq, H, L, chunk] -> [B, H, L, Nq, chunk], chunk = P*2 (loc) or P (attn) elements,
// moved as VB-byte vectors (VB = 16, 8, 4 or 2, the widest that divides the chunk).
template <int VB>
__global__ void __launch_bounds__(256)
repack_kernel(const char *__restrict__ src, char *__restrict__ dst, int B, int Nq, int HL,
              int vec_per_chunk, int64_t total)
{
    typedef typename std::conditional<VB == 16, uint4, typename std::conditional<VB == 8, uint2,
            typename std::conditional<VB == 4, uint32_t, uint16_t>::type>::type>::type V;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        // i enumerates the destination: (((b*HL + hl)*Nq + q)*vec_per_chunk + v)
        const int v = (int)(i % vec_per_chunk);
        int64_t r = i / vec_per_chunk;
        const int q = (int)(r % Nq); r /= Nq;
        const int hl = (int)(r % HL);
        const int64_t b = r / HL;
        const int64_t s = ((b * Nq + q) * HL + hl) * vec_per_chunk + v;
        reinterpret_cast<V *>(dst)[i] = reinterpret_cast<const V *>(src)[s];
    }
}

static hipError_t repack(const void *src, void *dst, const Dims &d, int chunk_bytes, hipStream_t st)
{
    const bool al = (((uintptr_t)src | (uintptr_t)dst) % 16) == 0;
    int vb = 2;
    if (al && chunk_bytes % 16 == 0) vb = 16;
    else if (al && chunk_bytes % 8 == 0) vb = 8;
    else if (al && chunk_bytes % 4 == 0) vb = 4;
    else if (chunk_bytes % 2) return hipErrorInvalidValue;
    const int vpc = chunk_bytes / vb;
    const int64_t total = (int64_t)d.B * d.Nq * d.H * d.L * vpc;
    const int64_t blocks = std::min<int64_t>((total + 255) / 256, 256 * 64);
    const int HL = d.H * d.L;
#define MMFS_RP(n) hipLaunchKernelGGL((repack_kernel<n>), dim3((unsigned)blocks), dim3(256), 0, st, \
                                      (const char *)src, (char *)dst, d.B, d.Nq, HL, vpc, total)
    if (vb == 16) MMFS_RP(16); else if (vb == 8) MMFS_RP(8); else if (vb == 4) MMFS_RP(4); else MMFS_RP(2);
#undef MMFS_RP
    return hipGetLastError();
}

TileParams make_params(const Dims &d)
{
    TileParams tp;
    // Tiles are big (a whole level when it has <= kMaxTilePx pixels): the per-tile fixed costs
    // (launch, plan, two scans, barriers) are paid few times.  nt_min only spreads the work
    // when there are few (b, h, level) slices: aim at ~2 workgroups per CU (256 CUs; measured best at the north-star shape).
    const int64_t slices = (int64_t)d.B * d.H * std::max(1, d.L);
    int64_t nt = std::max<int64_t>(1, (512 + slices - 1) / slices);
    if (const char *e = knob_str(K_NT_MIN)) nt = std::max(1, atoi(e));         // tuning knob
    tp.nt_min = (int)std::min<int64_t>(nt, 256);
    // tiles per level <= 2*nt_l + 1 with nt_l <= nt_min + px_l/kMaxTilePx + 1 (see plan_tile)
    const int64_t bound = 2LL * d.L * (tp.nt_min + 1) + 2LL * ((d.S + kMaxTilePx - 1) / kMaxTilePx) + d.L;
    tp.tiles_bound = (int)std::min<int64_t>(bound, 0x3fffffff);
    return tp;
}

struct Scratch {           // carved from the caller's workspace, 16-byte aligned pieces
    char *loc_t, *attn_t;
    uint32_t *cursor;
    TileTable *table;      // the tile plan (one per launch)
    int64_t table_bytes;
    uint2 *pixtab;         // [B, H, S] {first record, count}
    uint2 *records;        // [B, H, L, Nq*P*4] {query, weight}, pixel-sorted inside each tile
    int64_t cursor_bytes, total;
};

Scratch carve(void *workspace, int dtype, const Dims &d)
{
    const int64_t es = dtype == 0 ? 4 : 2;
    const int64_t pts = (int64_t)d.B * d.Nq * d.H * d.L * d.P;
    auto up = [](int64_t v) { return (v + 15) / 16 * 16; };
    Scratch s;
    char *p = (char *)workspace;
    s.loc_t = p;                 p += up(pts * 2 * es);
    s.attn_t = p;                p += up(pts * es);
    s.cursor = (uint32_t *)p;    s.cursor_bytes = up(((int64_t)d.B * d.H * d.L + (int64_t)d.B * d.H) * 4);  p += s.cursor_bytes;   // (+ the block path's arrival counters: same layout)
    s.table = (TileTable *)p;    s.table_bytes = up((int64_t)sizeof(TileTable) + (int64_t)make_params(d).tiles_bound * sizeof(Tile));
    p += s.table_bytes;
    s.pixtab = (uint2 *)p;       p += up((int64_t)d.B * d.H * d.S * 8);
    s.records = (uint2 *)p;      p += up(pts * 4 * 8);
    s.total = p - (char *)workspace;
    return s;
}

template <typename T, int NV>
hipError_t launch_sort(const int64_t *shapes, const int64_t *start, const Scratch &sc, const Dims &d,
                       hipStream_t st)
{
    const TileParams tp = make_params(d);
    const int64_t blocks = (int64_t)d.B * d.H * tp.tiles_bound;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(plan_tiles_kernel, dim3(1), dim3(64), 0, st, shapes, start, d.L, tp.nt_min,
                       tp.tiles_bound, sc.table);
    hipLaunchKernelGGL((msda_bwd_value_sort<T, NV>), dim3((unsigned)blocks), dim3(kThreads), 0, st,
                       shapes, start, (const T *)sc.loc_t, (const T *)sc.attn_t, sc.records, sc.cursor,
                       sc.pixtab, sc.table, d, tp);
    return hipGetLastError();
}

template <typename T, int LPS>
hipError_t launch_reduce(const Scratch &sc, const void *go, void *gv, const Dims &d, hipStream_t st)
{
    constexpr int GROUPS = kRThreads / LPS;
    const int chunks = (d.S + GROUPS - 1) / GROUPS;
    const int64_t blocks = (int64_t)d.B * d.H * chunks;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    if ((int64_t)d.Nq * d.H * d.D * (int64_t)sizeof(T) <= kMaxSlabBytes)
        hipLaunchKernelGGL((msda_bwd_value_reduce<T, LPS, true>), dim3((unsigned)blocks), dim3(kRThreads), 0, st,
                           (const T *)go, (T *)gv, sc.records, sc.pixtab, d, chunks);
    else
        hipLaunchKernelGGL((msda_bwd_value_reduce<T, LPS, false>), dim3((unsigned)blocks), dim3(kRThreads), 0, st,
                           (const T *)go, (T *)gv, sc.records, sc.pixtab, d, chunks);
    return hipGetLastError();
}

template <typename T>
hipError_t dispatch_sort(const int64_t *shapes, const int64_t *start, const Scratch &sc, const Dims &d,
                         hipStream_t st)
{
    // vectorised scan when the P locations of a (b,h,level,q) are whole, aligned 16-byte vectors
    const int loc_bytes = d.P * 2 * (int)sizeof(T);
    int nv = 0;
    if (loc_bytes % 16 == 0 && loc_bytes / 16 <= 2) nv = loc_bytes / 16;
    switch (nv) {
        case 1: return launch_sort<T, 1>(shapes, start, sc, d, st);
        case 2: return launch_sort<T, 2>(shapes, start, sc, d, st);
        default: return launch_sort<T, 0>(shapes, start, sc, d, st);
    }
}

template <typename T>
hipError_t dispatch_reduce(const Scratch &sc, const void *go, void *gv, const Dims &d, hipStream_t st)
{
    constexpr int VEC = 16 / (int)sizeof(T);
    switch (d.D / VEC) {
#define MMFS_CASE(n) case n: return launch_reduce<T, n>(sc, go, gv, d, st);
        MMFS_CASE(1) MMFS_CASE(2) MMFS_CASE(4) MMFS_CASE(8) MMFS_CASE(16) MMFS_CASE(32) MMFS_CASE(64)
#undef MMFS_CASE
        default: return hipErrorInvalidValue;
    }
}

}  // namespace

bool bwd_value_tiled_supported(int dtype, const Dims &d)
{
    if (!bwd_has_vector_path(dtype, d)) return false;
    if ((int64_t)d.B * d.H * d.L * d.Nq * d.P * 4 > 0xffffffffLL) return false;   // 32-bit record indices
    const TileParams tp = make_params(d);
    return (int64_t)d.B * d.H * tp.tiles_bound <= 0x7fffffffLL;
}

int64_t bwd_value_tiled_workspace_bytes(int dtype, const Dims &d)
{
    // room for either generation (they share the leading pieces: re-packed loc / attn, cursors)
    return std::max<int64_t>(carve(nullptr, dtype, d).total, bwd_value_block_workspace_bytes(dtype, d));
}

// Stage 2a: re-pack loc/attn into the workspace and clear the per-level cursors.
hipError_t backward_value_prepare(int dtype, const void *loc, const void *attn, void *workspace,
                                  const Dims &d, hipStream_t st,
                                  const int64_t *shapes, const int64_t *start, bool *planned)
{
    if (planned) *planned = false;
    if (!bwd_value_tiled_supported(dtype, d)) return hipErrorInvalidValue;
    if (bwd_value_block_supported(dtype, d)) {       // one launch, plan included when the table is at hand
        if (planned) *planned = shapes != nullptr && start != nullptr;
        return backward_value_block_prepare(dtype, loc, attn, planned ? shapes : nullptr, planned ? start : nullptr,
                                            workspace, d, st);
    }
    const int es = dtype == 0 ? 4 : 2;
    const Scratch sc = carve(workspace, dtype, d);
    hipError_t e = mmfs::zero_fill(sc.cursor, (size_t)sc.cursor_bytes, st);
    if (e != hipSuccess) return e;
    e = repack(loc, sc.loc_t, d, d.P * 2 * es, st);
    if (e != hipSuccess) return e;
    return repack(attn, sc.attn_t, d, d.P * es, st);
}

// Stage 2b: sort the tap contributions by pixel (prepared workspace -> records + run table).
hipError_t backward_value_sort(int dtype, const int64_t *shapes, const int64_t *start, void *workspace,
                               const Dims &d, hipStream_t st, bool planned)
{
    if (!bwd_value_tiled_supported(dtype, d)) return hipErrorInvalidValue;
    if (bwd_value_block_supported(dtype, d))
        return backward_value_block_sort(dtype, shapes, start, workspace, d, planned, st);
    const Scratch sc = carve(workspace, dtype, d);
    switch (dtype) {
        case 0: return dispatch_sort<float>(shapes, start, sc, d, st);
        case 1: return dispatch_sort<half_t>(shapes, start, sc, d, st);
        case 2: return dispatch_sort<bf16_t>(shapes, start, sc, d, st);
        default: return hipErrorInvalidValue;
    }
}

// Stage 2c: reduce every pixel's run into its grad_value row.
hipError_t backward_value_reduce(int dtype, const void *grad_out, void *grad_value, void *workspace,
                                 const Dims &d, hipStream_t st, bool all_rows_owned)
{
    if (!bwd_value_tiled_supported(dtype, d)) return hipErrorInvalidValue;
    if (bwd_value_block_supported(dtype, d))      // must mirror backward_value_sort
        return backward_value_block_reduce(dtype, grad_out, grad_value, workspace, d, all_rows_owned, st);
    const Scratch sc = carve(workspace, dtype, d);
    switch (dtype) {
        case 0: return dispatch_reduce<float>(sc, grad_out, grad_value, d, st);
        case 1: return dispatch_reduce<half_t>(sc, grad_out, grad_value, d, st);
        case 2: return dispatch_reduce<bf16_t>(sc, grad_out, grad_value, d, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t backward_value_run(int dtype, const int64_t *shapes, const int64_t *start,
                              const void *grad_out, void *grad_value, void *workspace, const Dims &d,
                              hipStream_t st, bool planned, bool all_rows_owned)
{
    const hipError_t e = backward_value_sort(dtype, shapes, start, workspace, d, st, planned);
    if (e != hipSuccess) return e;
    return backward_value_reduce(dtype, grad_out, grad_value, workspace, d, st, all_rows_owned);
}

hipError_t backward_value_tiled(int dtype, const int64_t *shapes, const int64_t *start,
                                const void *loc, const void *attn, const void *grad_out,
                                void *grad_value, void *workspace, const Dims &d, hipStream_t st,
                                bool all_rows_owned)
{
    bool planned = false;
    const hipError_t e = backward_value_prepare(dtype, loc, attn, workspace, d, st, shapes, start, &planned);
    if (e != hipSuccess) return e;
    return backward_value_run(dtype, shapes, start, grad_out, grad_value, workspace, d, st, planned, all_rows_owned);
}

}  // namespace mmfs
