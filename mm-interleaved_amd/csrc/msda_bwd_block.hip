// msda_bwd_block.hip -- grad_value of multi-scale deformable attention, block-stationary and
// cell-sorted: the second generation of the pixel-stationary kernels in msda_bwd_value.hip.
//
// What bounds the pixel-stationary reduce is not HBM and not the 64 B/clk/CU L1 path but rows that
// miss L1: random 256-byte rows out of an L2-resident region arrive at ~9.6 TB/s chip-wide
// (tools/ubench/gather.hip), and every sample makes FOUR pixel owners read the same grad_out row.
// Here the unit of the sort is the SAMPLE, keyed by the cell of its top-left corner, and the unit of
// the reduce is a 2x2 pixel block:
//
//   * one 16-byte record {query, y, x, attention} (pixel coordinates) per sample instead of four 8-byte {query,
//     weight} records: a quarter of the LDS atomics and of the scattered stores of the sort
//     (those stores, ~1.3 lanes/clk/CU, bound it), half the record bytes;
//   * a lane group owns a 2x2 block of one level's pixels and walks the runs of the 9 cells whose
//     footprints touch it: one grad_out row read per (sample, block) serves up to 4 pixels --
//     on average 2.25 row reads per sample instead of 4;
//   * the weights are the same products (wy * wx * attention) in the same order as before; only the
//     order of the fp32 sums changes.  Every grad_value row still has exactly one owner and is
//     written once in the storage type.
//
// Cells are indexed (y0 + 1, x0 + 1) with y0 in [-1, H-1], x0 in [-1, W-1]: (H+1)(W+1) per level.
// Reference semantics as in msda_bwd_value.hip (ms_deform_im2col_cuda.cuh:128-155, cast at the end:
// ms_deform_attn_cuda.cu:122-165).
#include "msda_device.h"
#include "msda_env.h"
#include "msda_launch.h"
#include "msda_bwd_block.h"
#include "msda_plan.h"
#include <vector>
#include <cstdlib>

namespace mmfs {

using namespace blk;

namespace {

// The 9 runs of a block, seen as one list: run k holds [pre[k], pre[k+1]) of it.  Kept in LDS
// (one per lane group; every group reads the others' in phase 2).
struct BlockRuns {
    uint32_t first[kNC];
    int pre[kNC + 1];
};

// The queue has one lane per XCD: workgroup w of the chunk kernel runs on XCD w % 8 (round-robin
// dispatch) and takes its chunks from lane w % 8, which holds the heads h with h % 8 == w % 8 -- the
// same head -> XCD affinity as every other kernel here, so a (b, h) slice of grad_out is pulled into
// ONE L2 instead of all eight.
struct OvfHeader { uint32_t n_slots, cap_slots, cap_entries, n_partials, cap_partials, pad[3]; uint32_t n_entries[kOvfLanes]; };

// Long lists.  The plan sizes `split` for uniformly spread samples; real MMFS inputs are not: every
// text token of the LLM path samples around the SAME reference point (the image centre), so a few
// blocks own most records of a level.  A block whose list is longer than twice what uniformly spread
// samples would give it (LevelRow::cap, at least kCapRecords) is not walked in place -- its owner
// would be one of the few busy lane groups of the chip -- but queued whole, in chunks of kOvfChunk
// records.  A second kernel gives every chunk a whole workgroup (its lane groups interleave the
// chunk's batches and meet in LDS): a block of one chunk gets its rows stored right there, a longer
// one leaves an fp32 partial per chunk for a third kernel to add up and round.  (Float atomics into
// one accumulator per block were tried first: 3x slower, the few hot addresses serialise.)  With
// uniformly spread samples the queue stays empty and the two extra launches return at once.
#ifndef MMFS_OVF_CHUNK
#define MMFS_OVF_CHUNK 2048
#endif
constexpr int kOvfChunk = MMFS_OVF_CHUNK;   // records per queued chunk = per WORKGROUP of the chunk kernel


struct OvfSlot {
    BlockRuns runs;                     // where the block's records are
    int b, h, by, bx;
    int Hl, Wl, lstart;
    uint32_t pbase, n_partials;         // its partial sums: [pbase + c] is chunk c's; 0 when the block is one chunk
    int pad;
};
struct OvfEntry { uint32_t slot, start, count, pidx; };      // pidx = kNoPartial: store the rows directly
constexpr uint32_t kNoPartial = 0xffffffffu;

__global__ void plan_cells_kernel(const PlanArgs pa)
{
    __shared__ __attribute__((aligned(16))) unsigned char plan_lds[kPlanLdsBytes];
    if (blockIdx.x == 0) plan_cells_body(pa, plan_lds);
}

// Stage 2a and the plan in ONE launch (small shapes are bound by launches, not by bytes: the backward
// used to open with a memset, two re-pack kernels and the plan kernel): every workgroup copies its
// share of loc / attn into the [b, h, level, query, point] order the sort scans and clears its share
// of the level cursors; the LAST workgroup also plans.  vb_*: bytes per copy (16 / 8 / 4 / 2).
template <typename V>
__device__ __forceinline__ void repack_part(const char *__restrict__ src, char *__restrict__ dst, int Nq, int HL,
                                            int vpc, int64_t total)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        // i enumerates the destination: (((b*HL + hl)*Nq + q)*vpc + v)
        const int v = (int)(i % vpc);
        int64_t r = i / vpc;
        const int q = (int)(r % Nq); r /= Nq;
        const int hl = (int)(r % HL);
        const int64_t b = r / HL;
        reinterpret_cast<V *>(dst)[i] = reinterpret_cast<const V *>(src)[((b * Nq + q) * HL + hl) * vpc + v];
    }
}

__device__ __forceinline__ void repack_any(const char *src, char *dst, int Nq, int HL, int vb, int vpc, int64_t total)
{
    if (vb == 16) repack_part<uint4>(src, dst, Nq, HL, vpc, total);
    else if (vb == 8) repack_part<uint2>(src, dst, Nq, HL, vpc, total);
    else if (vb == 4) repack_part<uint32_t>(src, dst, Nq, HL, vpc, total);
    else repack_part<uint16_t>(src, dst, Nq, HL, vpc, total);
}

__global__ void __launch_bounds__(256)
msda_bwd_prepare(const char *__restrict__ loc, char *__restrict__ loc_t, int vb_l, int vpc_l, int64_t total_l,
                 const char *__restrict__ attn, char *__restrict__ attn_t, int vb_a, int vpc_a, int64_t total_a,
                 int Nq, int HL, uint32_t *__restrict__ cursor, int64_t cursor_words, const PlanArgs pa)
{
    __shared__ __attribute__((aligned(16))) unsigned char plan_lds[kPlanLdsBytes];
    if (pa.shapes != nullptr && blockIdx.x == gridDim.x - 1) plan_cells_body(pa, plan_lds);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < cursor_words; i += (int64_t)gridDim.x * 256) cursor[i] = 0u;
    repack_any(loc, loc_t, Nq, HL, vb_l, vpc_l, total_l);
    repack_any(attn, attn_t, Nq, HL, vb_a, vpc_a, total_a);
}

// Exclusive prefix sum over a[0..n) (n <= kMaxTileCells), total left in a[n].
template <int THREADS>
__device__ void block_exclusive_scan(uint32_t *a, int n, uint32_t *wave_tot)
{
    constexpr int PER = kMaxTileCells / THREADS;
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
    if (tid == THREADS - 1) a[n] = run;
    __syncthreads();
}

// The sample's cell inside the tile, or -1: outside the level, zero weight, or another tile's.
// keep_zero: a zero weight is no reason to drop the sample (its grad_attn is not zero: Dims::taps_sorted without lazy_attn);
// *dead: the sample gets no record in ANY tile of its level.
__device__ __forceinline__ int cell_in_tile(float lx, float ly, float a, const CTile &tl, int tw, bool keep_zero = false, bool *dead = nullptr)
{
    const float y = ly * (float)tl.Hl - 0.5f, x = lx * (float)tl.Wl - 0.5f;
    const bool inside = (y > -1.f) && (x > -1.f) && (y < (float)tl.Hl) && (x < (float)tl.Wl);
    const bool none = !inside || (a == 0.f && !keep_zero);
    if (dead) *dead = none;
    if (none) return -1;
    const int cy = (int)floorf(y) + 1, cx = (int)floorf(x) + 1;
    if (cy < tl.ya || cy >= tl.yb || cx < tl.xa || cx >= tl.xb) return -1;
    return (cy - tl.ya) * tw + (cx - tl.xa);
}

enum ScanMode { kCount = 0, kScatter = 1, kScatterLds = 2 };

// One sample against the tile: if its top-left cell is in the tile
//   kCount  : off[cell] += 1
//   kScatter: list[off[cell] + cur[cell]++] = {q, y, x, attention}
// COMPACT (16-bit storage feeding the matrix-core reduce, Nq <= 65536): the record is the sample's INPUT
// words, 8 bytes instead of 16 -- {query | weight bits << 16, x bits | y bits << 16} -- and the reduce
// redoes y = ly * H - 0.5 itself: half the record traffic of a step (67 MB written + read at the north star).
template <int MODE, bool COMPACT>
__device__ __forceinline__ void visit_sample(float lx, float ly, float a, uint32_t xy_bits, uint32_t a_bits, int q,
                                             const CTile &tl, int tw, uint32_t *off, uint32_t *cur, void *__restrict__ list)
{
    // a sample whose attention weight is exactly zero adds nothing to grad_value (images a token
    // cannot see get exactly 0 from the masked softmax, mmfs.py:203-231): no record for it
    const int pl = cell_in_tile(lx, ly, a, tl, tw);
    if (pl < 0) return;
    const float y = ly * (float)tl.Hl - 0.5f, x = lx * (float)tl.Wl - 0.5f;
    if (MODE == kCount) {
        atomicAdd(&off[pl], 1u);
    } else {
        // kScatterLds: the whole tile's records fit the LDS window; off[] itself is the cursor (the cell table
        // has been written from it already) and ``list`` is the window, slot = index inside the tile
        const uint32_t slot = MODE == kScatterLds ? atomicAdd(&off[pl], 1u) : off[pl] + atomicAdd(&cur[pl], 1u);
        if (COMPACT) reinterpret_cast<uint2 *>(list)[slot] = make_uint2((uint32_t)q | (a_bits << 16), xy_bits);
        else reinterpret_cast<uint4 *>(list)[slot] = make_uint4((uint32_t)q, __float_as_uint(y), __float_as_uint(x), __float_as_uint(a));
    }
}

template <typename T> __device__ __forceinline__ uint32_t raw_bits(T v)
{
    if (sizeof(T) == 2) return (uint32_t)__builtin_bit_cast(uint16_t, v);
    return 0u;                                   // (fp32 storage never takes the compact format)
}
template <> __device__ __forceinline__ uint32_t raw_bits<float>(float) { return 0u; }

template <typename T, int MODE, int NV, bool COMPACT, int THREADS>
__device__ __forceinline__ void scan_samples(const T *__restrict__ loc, const T *__restrict__ attn,
                                             const Dims &d, const CTile &tl, int b, int h,
                                             uint32_t *off, uint32_t *cur, void *__restrict__ list)
{
    const int tw = tl.xb - tl.xa;
    const int64_t s_first = ((((int64_t)b * d.H + h) * d.L + tl.level) * d.Nq) * d.P;
    if (NV == 0) {
        for (int q = (int)threadIdx.x; q < d.Nq; q += THREADS) {
            const int64_t s0 = s_first + (int64_t)q * d.P;
            for (int p = 0; p < d.P; ++p) {
                const T lx = loc[2 * (s0 + p)], ly = loc[2 * (s0 + p) + 1], a = attn[s0 + p];
                visit_sample<MODE, COMPACT>(to_f32(lx), to_f32(ly), to_f32(a), raw_bits(lx) | (raw_bits(ly) << 16), raw_bits(a),
                                            q, tl, tw, off, cur, list);
            }
        }
        return;
    }
    typedef Vec16<T> V;
    constexpr int VEC = V::N;
    constexpr int NVV = NV > 0 ? NV : 1;
    for (int q0 = (int)threadIdx.x; q0 < d.Nq; q0 += THREADS * kScanUnroll) {
        uint4 lraw[kScanUnroll][NVV];
        uint2 araw[kScanUnroll][NVV];
#pragma unroll
        for (int u = 0; u < kScanUnroll; ++u) {
            const int q = q0 + u * THREADS;
            const int64_t s0 = s_first + (int64_t)min(q, d.Nq - 1) * d.P;
#pragma unroll
            for (int v = 0; v < NVV; ++v) {
                lraw[u][v] = reinterpret_cast<const uint4 *>(loc + 2 * s0)[v];
                araw[u][v] = reinterpret_cast<const uint2 *>(attn + s0)[v];     // (both passes: zero weights are skipped)
            }
        }
#pragma unroll
        for (int u = 0; u < kScanUnroll; ++u) {
            const int q = q0 + u * THREADS;
            if (q >= d.Nq) break;
#pragma unroll
            for (int v = 0; v < NVV; ++v) {
                float l[VEC], a[VEC];
                V::unpack(lraw[u][v], l);
                V::unpack(make_uint4(araw[u][v].x, araw[u][v].y, 0u, 0u), a);
                const uint32_t lw[4] = {lraw[u][v].x, lraw[u][v].y, lraw[u][v].z, lraw[u][v].w};
                const uint32_t aw[2] = {araw[u][v].x, araw[u][v].y};
#pragma unroll
                for (int i = 0; i < VEC / 2; ++i)
                    // (16-bit storage: sample i's (x, y) pair is word i of the vector, its weight half-word i)
                    visit_sample<MODE, COMPACT>(l[2 * i], l[2 * i + 1], a[i], lw[i & 3], (aw[(i >> 1) & 1] >> (16 * (i & 1))) & 0xffffu,
                                                q, tl, tw, off, cur, list);
            }
        }
    }
}

// Nq <= THREADS * kScanUnroll (one trip of the scan): a thread's samples stay in its registers between the
// counting pass and the placing pass, with the cell and the RANK inside the cell the counting atomic returned
// -- the placing pass then needs no atomic and no second read of loc / attn: slot = off[cell] + rank.
// (The second scan with its returning LDS atomics was 12 of the workgroup's 36 kclk at the north star.)
#ifndef MMFS_SORT_WINDOW_ROUNDS
#define MMFS_SORT_WINDOW_ROUNDS 1
#endif
constexpr bool kWindowRounds = MMFS_SORT_WINDOW_ROUNDS != 0;       // a kept scan may place its records window by window
constexpr uint32_t kNoCell = 0xffffffffu;
constexpr int kCellBits = 13;
static_assert(kMaxTileCells <= (1 << kCellBits), "a cell index and a rank share one word");
template <typename T, int NV, bool COMPACT, int THREADS, bool TS, int UNROLL = kScanUnroll>
struct KeptScan {
    typedef Vec16<T> V;
    static constexpr int VEC = V::N, SPV = V::N / 2;
    uint4 lraw[UNROLL][NV];
    uint2 araw[UNROLL][NV];
    uint32_t key[UNROLL][NV][SPV];
    // Two vectors per query (P = 8 of 16-bit storage): the samples' words AND their keys do not fit the 128
    // registers a 1024-thread workgroup leaves a thread; the placing pass then reads the words again (L2 hits) -- unless
    // the workgroup is 512 threads with twice the samples each (UNROLL = 8: 256 registers a thread; round 6, r06v)
    static constexpr bool kKeepRaw = NV == 1 || THREADS <= 512;
    // G > 1: a query's samples span G * NV vectors (many points per query: the reference's own speed test has 64);
    // "virtual query" tid + u * THREADS then stands for vectors (qv % G) * NV .. + NV of query qv / G
    int G = 1;

    // in_place: loc / attn are the op's own [B, Nq, H, L, P] arrays (a query's P samples of this (h, level) are
    // contiguous there too: the same vectors, one per 2 * H * L * P elements instead of back to back -- the
    // (b, h) slice's other levels, sorted on the same XCD at the same time, use the rest of the lines), else
    // the [B, H, L, Nq, P] copies of msda_bwd_prepare.
    __device__ __forceinline__ void load(const T *__restrict__ loc, const T *__restrict__ attn, const Dims &d,
                                         const CTile &tl, int b, int h, bool in_place)
    {
        const int64_t s_first = ((((int64_t)b * d.H + h) * d.L + tl.level) * d.Nq) * d.P;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int qv = min((int)threadIdx.x + u * THREADS, d.Nq * G - 1);
            const int qq = G == 1 ? qv : qv / G;
            const int v0 = (qv - qq * G) * NV;
            const int64_t s0 = in_place ? ((((int64_t)b * d.Nq + qq) * d.H + h) * d.L + tl.level) * d.P
                                        : s_first + (int64_t)qq * d.P;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                lraw[u][v] = reinterpret_cast<const uint4 *>(loc + 2 * s0)[v0 + v];
                araw[u][v] = reinterpret_cast<const uint2 *>(attn + s0)[v0 + v];
            }
        }
    }

    // Dims::taps_sorted: zero weights keep their record unless the caller never reads their gradients (lazy_attn), and the
    // level's first tile writes the zero gradients of the samples that get no record (g_loc / g_attn: the op's grad_loc /
    // grad_attn, [B, Nq, H, L, P(, 2)])
    __device__ __forceinline__ void count(const Dims &d, const CTile &tl, uint32_t *off, int b = 0, int h = 0,
                                          T *__restrict__ g_loc = nullptr, T *__restrict__ g_attn = nullptr)      // (after load())
    {
        const int tw = tl.xb - tl.xa;
        const bool keep_zero = TS && !d.lazy_attn;
        const bool zero_writer = TS && g_loc != nullptr && tl.ya == 0 && tl.xa == 0;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int q = (int)threadIdx.x + u * THREADS;                // (virtual query)
            const int qq = G == 1 ? q : q / G;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                float l[VEC], a[VEC];
                V::unpack(lraw[u][v], l);
                V::unpack(make_uint4(araw[u][v].x, araw[u][v].y, 0u, 0u), a);
#pragma unroll
                for (int i = 0; i < SPV; ++i) {
                    bool dead = false;
                    const int pl = q < d.Nq * G ? cell_in_tile(l[2 * i], l[2 * i + 1], a[i], tl, tw, keep_zero, &dead) : -1;
                    key[u][v][i] = pl < 0 ? kNoCell : ((uint32_t)pl | (atomicAdd(&off[pl], 1u) << kCellBits));
                    if (TS && zero_writer && dead && q < d.Nq * G) {
                        const int64_t s = ((((int64_t)b * d.Nq + qq) * d.H + h) * d.L + tl.level) * d.P + ((q - qq * G) * NV + v) * SPV + i;
                        g_attn[s] = (T)0.f;
                        g_loc[2 * s] = (T)0.f; g_loc[2 * s + 1] = (T)0.f;
                    }
                }
            }
        }
    }

    // list: the tile's records (the LDS window, or its place in the record area)
    // slots [s0, s0 + cap) only (a tile of more records than the window holds is placed window by window: ``list`` is
    // the window, slot s lands at s - s0); reload: read the words again first (they are not kept when NV == 2)
    __device__ __forceinline__ void place(const T *__restrict__ loc, const T *__restrict__ attn, const Dims &d,
                                          const CTile &tl, int b, int h, bool in_place, const uint32_t *off, void *__restrict__ list,
                                          uint32_t s0 = 0u, uint32_t cap = 0xffffffffu, bool reload = true)
    {
        if (!kKeepRaw && reload) load(loc, attn, d, tl, b, h, in_place);
        constexpr bool ts = COMPACT && TS;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint32_t qv = threadIdx.x + u * THREADS;
            const uint32_t qq = G == 1 ? qv : qv / (uint32_t)G;
            // the record carries the query -- or (taps_sorted) query * P + point: the sample's place in grad_loc / grad_attn
            const uint32_t q = ts ? qq * (uint32_t)d.P + (qv - qq * (uint32_t)G) * (uint32_t)(NV * SPV) : qq;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const uint32_t lw[4] = {lraw[u][v].x, lraw[u][v].y, lraw[u][v].z, lraw[u][v].w};
                const uint32_t aw[2] = {araw[u][v].x, araw[u][v].y};
                float l[VEC], a[VEC];
                if (!COMPACT) {
                    V::unpack(lraw[u][v], l);
                    V::unpack(make_uint4(araw[u][v].x, araw[u][v].y, 0u, 0u), a);
                }
#pragma unroll
                for (int i = 0; i < SPV; ++i) {
                    const uint32_t k = key[u][v][i];
                    if (k == kNoCell) continue;
                    uint32_t slot = off[k & ((1u << kCellBits) - 1u)] + (k >> kCellBits);
                    if (slot - s0 >= cap) continue;                   // (another window's; unsigned: slot < s0 too)
                    slot -= s0;
                    if (COMPACT)
                        reinterpret_cast<uint2 *>(list)[slot] =
                            make_uint2((ts ? q + (uint32_t)(v * SPV + i) : q) | (((aw[(i >> 1) & 1] >> (16 * (i & 1))) & 0xffffu) << 16), lw[i & 3]);
                    else
                        reinterpret_cast<uint4 *>(list)[slot] =
                            make_uint4(q, __float_as_uint(l[2 * i + 1] * (float)tl.Hl - 0.5f),
                                       __float_as_uint(l[2 * i] * (float)tl.Wl - 0.5f), __float_as_uint(a[i]));
                }
            }
        }
    }
};

struct TileParams {
    int tiles_bound;
    int nt_min;
    int vgroups;      // kept scan: groups of NV vectors per query (1 unless a query's samples of a level span more than NV vectors)
    int hgroup;       // sort: heads whose workgroups share an XCD (their loc / attn words share 128-byte lines); <= 1: head h on XCD h % 8
};

// Development aid (tools/exp_build.sh sprof "-DMMFS_PROFILE_SORT"; tools/sort_prof.py): shader clocks per phase of
// a sort workgroup, summed per level (thread 0 of each workgroup).
#ifdef MMFS_PROFILE_SORT
}  // namespace
__device__ unsigned long long g_sort_prof[8 * 8];       // [level % 8][phase]
__device__ unsigned long long g_sort_wg[4096 * 3];      // [workgroup] start, own work done, end (100 MHz ticks)
namespace {
#define SPROF_WG(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_sort_wg[blockIdx.x * 3 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define SPROF_DECL SPROF_WG(0); unsigned long long sp_c = __builtin_readcyclecounter()
#define SPROF(i) do { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long n_ = __builtin_readcyclecounter(); \
                      atomicAdd(&g_sort_prof[(tl.level & 7) * 8 + (i)], n_ - sp_c); sp_c = n_; } } while (0)
#else
#define SPROF_DECL do {} while (0)
#define SPROF(i) do {} while (0)
#define SPROF_WG(k) do {} while (0)
#endif

// ---------------------------------------------------------------- kernel A: sort by cell
// THREADS: 1024 lanes, or 256 for tiles of few samples (launch_sort): four workgroups per CU instead of one, so that
// the chain of dependent round trips a tile is (header, samples, counters, the level's cursor, records) overlaps with
// three other tiles' -- at the ViT-Adapter injector's shape (512 slices x 3 levels of ~1000 samples) a 1024-lane
// workgroup spends 7 us on a tile whatever its size, six rounds of them (tools/sort_prof.py injector, r04zw)
template <typename T, int NV, bool COMPACT, int THREADS, bool TS, int UNROLL = kScanUnroll>
__global__ void __launch_bounds__(THREADS)
msda_bwd_cell_sort(const T *loc, const T *attn, uint4 *__restrict__ records,
                   uint32_t *__restrict__ level_cursor, uint2 *__restrict__ celltab,
                   const CellHeader *__restrict__ hdr, const Dims d, const TileParams tp, const int cell_stride,
                   const uint32_t win_bytes, const TileReduceArgs ta)
{
    // A tile's sorted records leave through an LDS window when they fit it: scattered straight to memory, every
    // wave-store hits 64 different lines and the scatter pass took 40 of the workgroup's 60-78 kclk
    // (tools/sort_prof.py); out of the window they go as whole runs.  The window is the launch's dynamic LDS
    // (sort_window_bytes: the level's samples when they fit, else only the cursors of the direct path -- a
    // window that is never used would halve the workgroups a CU holds).
    __shared__ uint32_t off[kMaxTileCells + 1];
    extern __shared__ __attribute__((aligned(16))) unsigned char win[];
    uint32_t *cur = reinterpret_cast<uint32_t *>(win);          // (the direct path's cursors: it does not use the window)
    __shared__ uint32_t wave_tot[THREADS / 64];
    __shared__ uint32_t region;

    const int bid = blockIdx.x;
    int h = bid % d.H;
    int t = (bid / d.H) % tp.tiles_bound;
    int b = (bid / d.H) / tp.tiles_bound;
    if (tp.hgroup > 1) {
        // workgroups go to the XCDs round-robin: the 4 x tiles workgroups that read the same lines of loc / attn -- the
        // levels of ``hgroup`` neighbouring heads of one batch slice -- take consecutive slots of ONE XCD
        const int per = tp.hgroup * tp.tiles_bound, hg = d.H / tp.hgroup;
        const int x = bid & 7, k = bid >> 3;
        const int group = (k / per) * 8 + x, within = k % per;
        if (group >= d.B * hg) return;
        b = group / hg;
        h = (group % hg) * tp.hgroup + within % tp.hgroup;
        t = within / tp.hgroup;
    }
    if (hdr->stamp != header_stamp(d) || t >= hdr->n_tiles) return;      // (a plan made for other dimensions: not ours)
    const CTile tl = tiles_of(hdr, d.L)[t];
    // (matrix-core reduce: the level's row, for the blocks this tile plans itself; asked for early)
    LevelRow lr = {};
    const int seamed = hdr->pad[1];
    if (ta.th != nullptr) lr = level_rows(hdr)[tl.level];

    const int tid = threadIdx.x;
    const int tw = tl.xb - tl.xa;
    const int ncell = (tl.yb - tl.ya) * tw;

    // one trip of the scan: the samples stay in registers between the two passes (KeptScan)
    constexpr int KNV = NV > 0 ? NV : 1;
    // (only where the launch expects windows: into memory, the two-scan path's stores are the faster -- SD 512 px
    // geometry, 32768 samples per level: 151 us against 196)
    const bool kept = NV > 0 && (int64_t)d.Nq * tp.vgroups <= THREADS * UNROLL && win_bytes > kMaxTileCells * 4u;
    KeptScan<T, KNV, COMPACT, THREADS, TS, UNROLL> ks;
    ks.G = tp.vgroups;

    // (the sort reads the op's own loc / attn when the opening launch said so: nothing was re-packed then)
    const bool in_place = hdr->loc_src != nullptr;
    if (in_place) {
        if (!kept) return;          // (a header that does not belong to this launch: nothing is sorted, nothing trapped)
        loc = reinterpret_cast<const T *>(hdr->loc_src); attn = reinterpret_cast<const T *>(hdr->attn_src);
    }

    SPROF_DECL;
    if (kept) ks.load(loc, attn, d, tl, b, h, in_place);          // (in flight while the counters are cleared)
    for (int i = tid; i < ncell; i += THREADS) { off[i] = 0u; if (!kept) cur[i] = 0u; }
    __syncthreads();
    SPROF(0);
    if (kept) ks.count(d, tl, off, b, h, reinterpret_cast<T *>(ta.g_loc), reinterpret_cast<T *>(ta.g_attn));
    else scan_samples<T, kCount, NV, COMPACT, THREADS>(loc, attn, d, tl, b, h, off, cur, nullptr);
    __syncthreads();
    SPROF(1);
    block_exclusive_scan<THREADS>(off, ncell, wave_tot);
    SPROF(2);
    const uint32_t total = off[ncell];
    // this tile's slice of the (b, h, level) record area: the level's tiles share Nq*P slots
    const int64_t slot = ((int64_t)b * d.H + h) * d.L + tl.level;
    if (tid == 0) region = total ? atomicAdd(&level_cursor[slot], total) : 0u;
    __syncthreads();
    const int64_t base = slot * ((int64_t)d.Nq * d.P) + region;
    void *const area = COMPACT ? (void *)(reinterpret_cast<uint2 *>(records) + base) : (void *)(records + base);
    const bool windowed = (uint64_t)total * (COMPACT ? 8u : 16u) <= win_bytes;
    // The cell table: read by the vector-ALU reduce and, for the matrix-core one, by the slice's last workgroup
    // for the blocks on the seams between tiles -- with no seam, by nobody.
    uint2 *tab = celltab + ((int64_t)b * d.H + h) * cell_stride + tl.cbase;
    for (int p = tid; p < (ta.th != nullptr && !seamed ? 0 : ncell); p += THREADS) {
        const int cg = (tl.ya + p / tw) * (tl.Wl + 1) + tl.xa + p % tw;
        // (written through, agent scope: the slice's last workgroup reads the table within this launch)
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(&tab[cg]),
                           ((unsigned long long)(off[p + 1] - off[p]) << 32) | (unsigned long long)(uint32_t)(base + off[p]),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    PendingBlock pending;
    pending.blk = -1;
    if (ta.th != nullptr && lr.band > 0) pending = plan_tile_begin<TS>(ta, d, (int64_t)b * d.H + h, tl, lr, off, base, tid, THREADS);
    SPROF(3);
    // (a kept scan whose records exceed the window places them window by window: the records of a window are
    // consecutive slots, so each window leaves as one coalesced copy)
    const uint32_t win_cap = win_bytes / (COMPACT ? 8u : 16u);
    // the window's first n records -> slots s0 .. of the tile's place in the record area, coalesced
    // (-DMMFS_SORT_NT_RECS: with the non-temporal hint -- experiment, profiles/r03_experiments.md r03p)
    auto copy_out = [&](uint32_t s0, uint32_t n) {
        if (COMPACT) {
            const uint2 *src = reinterpret_cast<const uint2 *>(win);
            uint2 *dst = reinterpret_cast<uint2 *>(area) + s0;
            for (uint32_t i = tid; i < n; i += THREADS) {
#ifdef MMFS_SORT_NT_RECS
                __builtin_nontemporal_store(*reinterpret_cast<const unsigned long long *>(&src[i]), reinterpret_cast<unsigned long long *>(&dst[i]));
#else
                dst[i] = src[i];
#endif
            }
        } else {
            const uint4 *src = reinterpret_cast<const uint4 *>(win);
            uint4 *dst = reinterpret_cast<uint4 *>(area) + s0;
            for (uint32_t i = tid; i < n; i += THREADS) dst[i] = src[i];
        }
    };
    const bool rounds = kept && !windowed && kWindowRounds;
    if (total) {
        if (rounds) {
            for (uint32_t s0 = 0; s0 < total; s0 += win_cap) {
                const uint32_t n = min(win_cap, total - s0);
                ks.place(loc, attn, d, tl, b, h, in_place, off, win, s0, win_cap, true);
                __syncthreads();
                copy_out(s0, n);
                __syncthreads();
            }
        } else if (kept) {
            ks.place(loc, attn, d, tl, b, h, in_place, off, windowed ? (void *)win : area);
        } else if (windowed) {
            __syncthreads();                                      // the table and the plan have read off[]; now off[] becomes the cursors
            scan_samples<T, kScatterLds, NV, COMPACT, THREADS>(loc, attn, d, tl, b, h, off, cur, win);
        } else {
            scan_samples<T, kScatter, NV, COMPACT, THREADS>(loc, attn, d, tl, b, h, off, cur, area);
        }
        if (windowed && !rounds) {
            __syncthreads();
            copy_out(0u, total);
        }
    }
    if (ta.th != nullptr) plan_tile_finish(ta, d, (int64_t)b * d.H + h, pending);
    // Matrix-core reduce, levels cut into several tiles: the LAST workgroup of the (b, h) slice turns the slice's
    // cell table into the descriptors and work items of the blocks on the seams.  Hand-off inside the launch without the L2 write-back of a release
    // fence (with the record scatter in flight that write-back costs more than the sort): the table is
    // stored and read with 8-byte agent-scope accesses, the stores are drained before the arrival
    // counter moves (cdna_hip_programming.md guideline 16, "8-B agent atomics both sides").
    SPROF(4);
    SPROF_WG(1);
    if (ta.th != nullptr && seamed) {
        __shared__ uint32_t arrived;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0)
            arrived = __hip_atomic_fetch_add(&ta.slice_done[(int64_t)b * d.H + h], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        SPROF(5);
        if (arrived == (uint32_t)hdr->n_tiles - 1u)
            plan_slice_blocks<TS>(ta, d, (int64_t)b * d.H + h, tid, THREADS);
        SPROF(6);
    }
    SPROF_WG(2);
}

// ---------------------------------------------------------------- kernel B: reduce by 2x2 block
constexpr int kRThreads = 256;
#ifndef MMFS_BLK_WAVES
#define MMFS_BLK_WAVES 3          // waves per SIMD the register allocation must leave room for
#endif
#ifndef MMFS_BLK_UNROLL
#define MMFS_BLK_UNROLL 4
#endif
constexpr int kUnroll = MMFS_BLK_UNROLL;


struct BlkRec { uint32_t off; float w[kNPX]; };   // row offset ("outside" past the end), weights of the block's pixels


// Record e of the block's list -> its grad_out row offset and the weights it adds to the block's
// four pixels (zero where the corner is another block's).
__device__ __forceinline__ BlkRec fetch_record(const uint4 *__restrict__ records, const BlockRuns *__restrict__ brp,
                                               int e, int end, uint32_t row_bytes)
{
    BlkRec r;
    r.off = kOobOffset;
#pragma unroll
    for (int i = 0; i < kNPX; ++i) r.w[i] = 0.f;
    if (e < end) {
        int k = 0;
#pragma unroll
        for (int j = 1; j < kNC; ++j) k += e >= brp->pre[j] ? 1 : 0;   // pre[] is non-decreasing
        const int p = brp->pre[k];
        const uint32_t f = brp->first[k];
        const uint4 rec = records[f + (uint32_t)(e - p)];
        // (records carry the pixel coordinates; the fractions are the same subtraction the sort would do)
        const float py = __uint_as_float(rec.y), px_ = __uint_as_float(rec.z), a = __uint_as_float(rec.w);
        const float fy = py - floorf(py), fx = px_ - floorf(px_);
        // the cell is (BH*by + dy, BW*bx + dx), i.e. the sample's top row is y0 = BH*by + dy - 1: block
        // pixel row ry takes its corner cy = ry - (dy - 1), weight 1-fy for cy = 0, fy for cy = 1
        const int dy = k / (kBW + 1), dx = k - dy * (kBW + 1);
        float wy[kBH], wx[kBW];
#pragma unroll
        for (int ry = 0; ry < kBH; ++ry) wy[ry] = ry == dy - 1 ? 1.f - fy : (ry == dy ? fy : 0.f);
#pragma unroll
        for (int rx = 0; rx < kBW; ++rx) wx[rx] = rx == dx - 1 ? 1.f - fx : (rx == dx ? fx : 0.f);
        r.off = rec.x * row_bytes;
#pragma unroll
        for (int ry = 0; ry < kBH; ++ry)
#pragma unroll
            for (int rx = 0; rx < kBW; ++rx) r.w[ry * kBW + rx] = wy[ry] * wx[rx] * a;
    }
    return r;
}

// acc[px][:] += sum over one batch of LPS records (every lane fetched one; all lanes of the group
// read each row together: records go through the group's LDS slots -- the row offsets are read
// when the rows are requested, the weights only when they are used).  Rows are requested kUnroll
// at a time, one group of kUnroll ahead of the FMAs, in a ROLLED loop: unrolled over the whole
// batch the compiler turns the sums into one chain per accumulator across all the rows and keeps
// every row unpacked at once (208 VGPRs).
template <typename T, int LPS, bool BUF>
__device__ __forceinline__ void consume_batch(const BlkRec &mine, int lig, uint32_t *__restrict__ slot_off,
                                              uint4 *__restrict__ slot_w, const T *__restrict__ gslice, int64_t HD,
                                              __amdgpu_buffer_rsrc_t rsrc, uint32_t row_bytes, uint32_t lane_off,
                                              float (&acc)[kNPX][Vec16<T>::N])
{
    typedef Vec16<T> V;
    constexpr int U = LPS < kUnroll ? LPS : kUnroll;
    constexpr int WV = kNPX / 4;                     // weight vectors per record
    slot_off[lig] = mine.off;
#pragma unroll
    for (int v = 0; v < WV; ++v)
        slot_w[lig * WV + v] = make_uint4(__float_as_uint(mine.w[4 * v]), __float_as_uint(mine.w[4 * v + 1]),
                                          __float_as_uint(mine.w[4 * v + 2]), __float_as_uint(mine.w[4 * v + 3]));
    __builtin_amdgcn_wave_barrier();
    auto request = [&](int u) {
        const uint32_t off = slot_off[u];
        if (BUF) return buffer_load16(rsrc, off + lane_off);
        // flat addresses: the records were fetched with a row pitch of 1, so off is the row index
        const bool ok = off != kOobOffset;
        uint4 r = *reinterpret_cast<const uint4 *>(gslice + (int64_t)(ok ? off : 0u) * HD);
        if (!ok) r = make_uint4(0u, 0u, 0u, 0u);
        return r;
    };
    uint4 cur[U], nxt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { cur[u] = request(u); nxt[u] = cur[u]; }
#pragma unroll 1
    for (int u0 = 0; u0 < LPS; u0 += U) {
        if (u0 + U < LPS) {
#pragma unroll
            for (int u = 0; u < U; ++u) nxt[u] = request(u0 + U + u);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float g[V::N];
            V::unpack(cur[u], g);
#pragma unroll
            for (int v = 0; v < WV; ++v) {
                const uint4 ww = slot_w[(u0 + u) * WV + v];
                const float w[4] = {__uint_as_float(ww.x), __uint_as_float(ww.y), __uint_as_float(ww.z), __uint_as_float(ww.w)};
#pragma unroll
                for (int px = 0; px < 4; ++px) {
#pragma unroll
                    for (int i = 0; i < V::N; ++i) acc[4 * v + px][i] = fmaf(w[px], g[i], acc[4 * v + px][i]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
    __builtin_amdgcn_wave_barrier();
}

template <typename T, int LPS, bool BUF>
__global__ void __launch_bounds__(kRThreads, MMFS_BLK_WAVES)
msda_bwd_block_reduce(const T *__restrict__ grad_out, T *__restrict__ grad_value,
                      const uint4 *__restrict__ records, const uint2 *__restrict__ celltab,
                      const CellHeader *__restrict__ hdr, const Dims d, const int chunks, const int cell_stride,
                      OvfHeader *__restrict__ ovf, OvfSlot *__restrict__ oslots, OvfEntry *__restrict__ oentries,
                      float *__restrict__ oacc)
{
    typedef Vec16<T> V;
    constexpr int VEC = V::N;
    constexpr int GROUPS = kRThreads / LPS;          // lane groups per workgroup
    constexpr int D = LPS * VEC;
    static_assert(GROUPS % kMaxSplit == 0, "a block's lane groups must share a workgroup");
    __shared__ LevelRow lvs[kLdsLevels];
    __shared__ BlockRuns runs[GROUPS];
    __shared__ float scratch[GROUPS * kRoundPx * D];  // partial sums of split blocks, kRoundPx pixels per round
    __shared__ uint32_t slots_off[GROUPS * (LPS + 1)];
    __shared__ uint4 slots_w[GROUPS * (LPS + 1) * (kNPX / 4)];

    const int bid = blockIdx.x;
    const int h = bid % d.H;
    const int chunk = (bid / d.H) % chunks;
    const int b = (bid / d.H) / chunks;
    const int tid = threadIdx.x;
    const int gid = tid / LPS, lig = tid % LPS;
    const int n_blocks = hdr->stamp == header_stamp(d) ? hdr->n_blocks : 0;              // virtual blocks: block x split
    if (chunk * GROUPS >= n_blocks) return;          // whole workgroup beyond the last block
    const LevelRow *grows = level_rows(hdr);
    for (int i = tid; i < min(d.L, kLdsLevels); i += kRThreads) lvs[i] = grows[i];
    __syncthreads();

    uint32_t *slot = slots_off + gid * (LPS + 1);
    uint4 *slot4 = slots_w + gid * (LPS + 1) * (kNPX / 4);
    const int vb = chunk * GROUPS + gid;

    // virtual block -> (level, block, part)
    int l = 0;
    while (l + 1 < d.L && vb >= (l + 1 < kLdsLevels ? lvs[l + 1].bbase : grows[l + 1].bbase)) ++l;
    const LevelRow lr = l < kLdsLevels ? lvs[l] : grows[l];
    const int rel = vb - lr.bbase;
    const int split = lr.split;
    const int blk = rel / split, part = rel - blk * split;
    const bool act = vb < n_blocks && lr.nbx > 0 && blk < lr.nbx * lr.nby;      // (padding between levels: idle)
    const int by = act ? blk / lr.nbx : 0, bx = act ? blk - by * lr.nbx : 0;

    const int64_t HD = (int64_t)d.H * d.D;
    const T *gslice = grad_out + ((int64_t)b * d.Nq * d.H + h) * d.D + lig * VEC;
    const uint32_t row_bytes = BUF ? (uint32_t)(HD * sizeof(T)) : 1u;     // pitch of a record's row offset
    const uint32_t lane_off = (uint32_t)(lig * 16);
    __amdgpu_buffer_rsrc_t rsrc;
    if (BUF) rsrc = make_slab_rsrc(grad_out + ((int64_t)b * d.Nq * d.H + h) * d.D,
                                   ((int64_t)d.Nq * HD - (int64_t)h * d.D) * (int64_t)sizeof(T));

    // the cell runs of the block, as one list (lane 0 of the group builds the prefix)
    const uint2 *tab = celltab + ((int64_t)b * d.H + h) * cell_stride + lr.cbase;
    if (lig == 0) {
        int run = 0;
        runs[gid].pre[0] = 0;
#pragma unroll
        for (int k = 0; k < kNC; ++k) {
            uint2 r = make_uint2(0u, 0u);
            const int cy = kBH * by + k / (kBW + 1), cx = kBW * bx + k % (kBW + 1);
            if (act && cy <= lr.Hl && cx <= lr.Wl) r = tab[cy * (lr.Wl + 1) + cx];
            runs[gid].first[k] = r.x;
            run += (int)r.y;
            runs[gid].pre[k + 1] = run;
        }
    }
    __builtin_amdgcn_wave_barrier();
    const BlockRuns *br = &runs[gid];
    const int n = br->pre[kNC];

    float acc[kNPX][VEC];
#pragma unroll
    for (int px = 0; px < kNPX; ++px)
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[px][i] = 0.f;

    // this group's batches of the list: part, part + split, ... -- at most kCapRecords records of it;
    // what lies beyond the covered prefix of a long list is queued below
    // A list longer than the level's cap (hot spot) is not walked here at all: its few owners would
    // be the only busy lane groups of the chip.  It is queued whole, in chunks (below).
    const bool hot = n > lr.cap;
    const int covered = hot ? 0 : n;
    const int batches = (covered + LPS - 1) / LPS;
    int nb = batches > part ? (batches - part + split - 1) / split : 0;
#pragma unroll
    for (int o = LPS; o < 64; o <<= 1) nb = max(nb, __shfl_xor(nb, o, 64));    // wave-uniform trip count
    BlkRec pre = fetch_record(records, br, part * LPS + lig, covered, row_bytes);
    for (int j = 0; j < nb; ++j) {
        const BlkRec cur_rec = pre;
        if (j + 1 < nb) pre = fetch_record(records, br, (part + (j + 1) * split) * LPS + lig, covered, row_bytes);
        consume_batch<T, LPS, BUF>(cur_rec, lig, slot, slot4, gslice, HD, rsrc, row_bytes, lane_off, acc);
    }
    // split blocks: the parts meet in LDS, part 0 adds them up (uniform decision per workgroup is not
    // possible -- levels may change inside a workgroup -- so every workgroup passes the two barriers)
#pragma unroll
    for (int r0 = 0; r0 < kNPX; r0 += kRoundPx) {
        if (r0 > 0) __syncthreads();
#pragma unroll
        for (int px = 0; px < kRoundPx; ++px)
#pragma unroll
            for (int i = 0; i < VEC; ++i) scratch[(gid * kRoundPx + px) * D + lig * VEC + i] = acc[r0 + px][i];
        __syncthreads();
        if (act && part == 0) {
            for (int s2 = 1; s2 < split; ++s2) {
#pragma unroll
                for (int px = 0; px < kRoundPx; ++px)
#pragma unroll
                    for (int i = 0; i < VEC; ++i) acc[r0 + px][i] += scratch[((gid + s2) * kRoundPx + px) * D + lig * VEC + i];
            }
        }
    }
    if (act && part == 0) {
        bool queued = false;
        if (n > covered) {
            // queue the rest of the list in chunks; the block's sums so far go to an accumulator slot
            const int rest = n - covered, nchunks = (rest + kOvfChunk - 1) / kOvfChunk;
            uint32_t sl = 0, base = 0, pb = 0;
            const int qlane = h % kOvfLanes;
            const uint32_t npart = nchunks > 1 ? (uint32_t)nchunks : 0u;     // one chunk: rows stored by the chunk kernel
            if (lig == 0) {
                sl = atomicAdd(&ovf->n_slots, 1u);
                base = atomicAdd(&ovf->n_entries[qlane], (uint32_t)nchunks);
                if (npart) pb = atomicAdd(&ovf->n_partials, npart);
            }
            sl = __shfl(sl, 0, LPS); base = __shfl(base, 0, LPS); pb = __shfl(pb, 0, LPS);
            if (sl < ovf->cap_slots && base + (uint32_t)nchunks <= ovf->cap_entries &&
                pb + npart <= ovf->cap_partials) {
                queued = true;
                if (lig == 0) {
                    OvfSlot os;
                    os.runs = *br;
                    os.b = b; os.h = h; os.by = by; os.bx = bx;
                    os.Hl = lr.Hl; os.Wl = lr.Wl; os.lstart = lr.lstart; os.pad = 0;
                    os.pbase = pb; os.n_partials = npart;
                    oslots[sl] = os;
                }
                for (int c = lig; c < nchunks; c += LPS) {
                    OvfEntry e;
                    e.slot = sl; e.start = (uint32_t)(covered + c * kOvfChunk);
                    e.count = (uint32_t)min(kOvfChunk, rest - c * kOvfChunk);
                    e.pidx = npart ? pb + (uint32_t)c : kNoPartial;
                    oentries[(size_t)qlane * ovf->cap_entries + base + c] = e;
                }
            } else {
                // Queue full (more hot blocks than the host provided for): whatever this block did
                // reserve must still read as empty to the two kernels that walk the queue, then the
                // list is finished here, slowly.
                if (lig == 0 && sl < ovf->cap_slots) {
                    OvfSlot os;
                    os.runs = *br;
                    os.b = os.h = os.by = os.bx = 0;
                    os.Hl = os.Wl = 0; os.lstart = 0; os.pbase = 0; os.n_partials = 0; os.pad = 0;
                    oslots[sl] = os;
                }
                for (int c = lig; c < nchunks; c += LPS) {
                    if (base + (uint32_t)c < ovf->cap_entries) {
                        OvfEntry e;
                        e.slot = 0; e.start = 0; e.count = 0; e.pidx = 0;
                        oentries[(size_t)qlane * ovf->cap_entries + base + c] = e;
                    }
                }
                for (int e0 = covered; e0 < n; e0 += LPS) {
                    const BlkRec r = fetch_record(records, br, e0 + lig, n, row_bytes);
                    consume_batch<T, LPS, BUF>(r, lig, slot, slot4, gslice, HD, rsrc, row_bytes, lane_off, acc);
                }
            }
        }
        if (!queued) {
#pragma unroll
            for (int px = 0; px < kNPX; ++px) {
                const int y = kBH * by + px / kBW, x = kBW * bx + px % kBW;
                if (y < lr.Hl && x < lr.Wl) {
                    T *o = grad_value + (((int64_t)b * d.S + lr.lstart + y * lr.Wl + x) * d.H + h) * d.D + lig * VEC;
                    store16_stream(o, V::pack(acc[px]));
                }
            }
        }
    }
}

// ---------------------------------------------------------------- kernel C: queued chunks of long lists
// One WORKGROUP per queued chunk (a fixed grid walks its queue lane with a stride): the lane groups
// interleave the chunk's batches, meet in LDS, and group 0 either stores the block's rows (a block
// of one chunk) or leaves the chunk's fp32 partial.
template <typename T, int LPS, bool BUF>
__global__ void __launch_bounds__(kRThreads, MMFS_BLK_WAVES)
msda_bwd_block_overflow(const T *__restrict__ grad_out, T *__restrict__ grad_value,
                        const uint4 *__restrict__ records, const OvfHeader *__restrict__ ovf,
                        const OvfSlot *__restrict__ oslots, const OvfEntry *__restrict__ oentries,
                        float *__restrict__ oacc, const Dims d)
{
    typedef Vec16<T> V;
    constexpr int VEC = V::N;
    constexpr int GROUPS = kRThreads / LPS;
    constexpr int D = LPS * VEC;
    __shared__ BlockRuns runs;
    __shared__ float scratch[GROUPS * 4 * D];
    __shared__ uint32_t slots_off[GROUPS * (LPS + 1)];
    __shared__ uint4 slots_w[GROUPS * (LPS + 1) * (kNPX / 4)];

    const int tid = threadIdx.x, gid = tid / LPS, lig = tid % LPS;
    const int qlane = blockIdx.x % kOvfLanes;
    const uint32_t n_entries = min(ovf->n_entries[qlane], ovf->cap_entries);
    const uint32_t e_step = (uint32_t)(gridDim.x / kOvfLanes);
    uint32_t *slot = slots_off + gid * (LPS + 1);
    uint4 *slot4 = slots_w + gid * (LPS + 1) * (kNPX / 4);
    const int64_t HD = (int64_t)d.H * d.D;
    const uint32_t row_bytes = BUF ? (uint32_t)(HD * sizeof(T)) : 1u;
    const uint32_t lane_off = (uint32_t)(lig * 16);

    for (uint32_t ei = (uint32_t)(blockIdx.x / kOvfLanes); ei < n_entries; ei += e_step) {
        const OvfEntry en = oentries[(size_t)qlane * ovf->cap_entries + ei];     // uniform over the workgroup
        if (en.count == 0) continue;                  // a reservation its owner could not use
        const OvfSlot *os = oslots + en.slot;
        __syncthreads();                              // the previous chunk's LDS pieces are free
        if (tid == 0) runs = os->runs;
        __syncthreads();
        const int b = os->b, h = os->h;
        const T *gslice = grad_out + ((int64_t)b * d.Nq * d.H + h) * d.D + lig * VEC;
        __amdgpu_buffer_rsrc_t rsrc;
        if (BUF) rsrc = make_slab_rsrc(grad_out + ((int64_t)b * d.Nq * d.H + h) * d.D,
                                       ((int64_t)d.Nq * HD - (int64_t)h * d.D) * (int64_t)sizeof(T));
        else rsrc = make_slab_rsrc(grad_out, 0);

        float acc[kNPX][VEC];
#pragma unroll
        for (int px = 0; px < kNPX; ++px)
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[px][i] = 0.f;

        const int first = (int)en.start, end = (int)(en.start + en.count);
        const int batches = ((int)en.count + LPS - 1) / LPS;
        const int nb = (batches + GROUPS - 1) / GROUPS;                     // uniform: every group runs nb rounds
        BlkRec pre = fetch_record(records, &runs, first + gid * LPS + lig, end, row_bytes);
        for (int j = 0; j < nb; ++j) {
            const BlkRec cur_rec = pre;
            if (j + 1 < nb) pre = fetch_record(records, &runs, first + (gid + (j + 1) * GROUPS) * LPS + lig, end, row_bytes);
            consume_batch<T, LPS, BUF>(cur_rec, lig, slot, slot4, gslice, HD, rsrc, row_bytes, lane_off, acc);
        }
        // the groups meet in LDS, 4 pixels per round; group 0 adds them up
#pragma unroll
        for (int r0 = 0; r0 < kNPX; r0 += 4) {
            if (r0 > 0) __syncthreads();
#pragma unroll
            for (int px = 0; px < 4; ++px)
#pragma unroll
                for (int i = 0; i < VEC; ++i) scratch[(gid * 4 + px) * D + lig * VEC + i] = acc[r0 + px][i];
            __syncthreads();
            if (gid == 0) {
#pragma unroll 1
                for (int g2 = 1; g2 < GROUPS; ++g2) {
#pragma unroll
                    for (int px = 0; px < 4; ++px)
#pragma unroll
                        for (int i = 0; i < VEC; ++i) acc[r0 + px][i] += scratch[(g2 * 4 + px) * D + lig * VEC + i];
                }
            }
        }
        if (gid == 0) {
            if (en.pidx == kNoPartial) {
#pragma unroll
                for (int px = 0; px < kNPX; ++px) {
                    const int y = kBH * os->by + px / kBW, x = kBW * os->bx + px % kBW;
                    if (y < os->Hl && x < os->Wl) {
                        T *o = grad_value + (((int64_t)b * d.S + os->lstart + y * os->Wl + x) * d.H + h) * d.D + lig * VEC;
                        store16_stream(o, V::pack(acc[px]));
                    }
                }
            } else {
                float *oa = oacc + (size_t)en.pidx * (kNPX * D) + lig * VEC;
#pragma unroll
                for (int px = 0; px < kNPX; ++px)
#pragma unroll
                    for (int i = 0; i < VEC; ++i) oa[px * D + i] = acc[px][i];
            }
        }
    }
}

// ---------------------------------------------------------------- kernel D: accumulator slots -> grad_value rows
template <typename T>
__global__ void __launch_bounds__(256)
msda_bwd_block_ovf_store(T *__restrict__ grad_value, const OvfHeader *__restrict__ ovf,
                         const OvfSlot *__restrict__ oslots, const float *__restrict__ oacc, const Dims d)
{
    const uint32_t n_slots = min(ovf->n_slots, ovf->cap_slots);
    for (uint32_t sl = blockIdx.x; sl < n_slots; sl += gridDim.x) {
        const OvfSlot os = oslots[sl];
        if (os.n_partials == 0) continue;             // one chunk: the chunk kernel stored the rows itself
        const float *oa = oacc + (size_t)os.pbase * (kNPX * d.D);
        for (int i = threadIdx.x; i < kNPX * d.D; i += 256) {
            const int px = i / d.D, ch = i - px * d.D;
            const int y = kBH * os.by + px / kBW, x = kBW * os.bx + px % kBW;
            if (y < os.Hl && x < os.Wl) {
                float sum = 0.f;
                for (uint32_t c = 0; c < os.n_partials; ++c) sum += oa[(size_t)c * (kNPX * d.D) + i];
                grad_value[(((int64_t)os.b * d.S + os.lstart + y * os.Wl + x) * d.H + os.h) * d.D + ch] = (T)sum;
            }
        }
    }
}

TileParams make_params(const Dims &d)
{
    TileParams tp;
    const int64_t slices = (int64_t)d.B * d.H * std::max(1, d.L);
    // every tile of a level scans all of the level's samples (twice), so tiles are only added when there
    // are too few (b, h, level) slices to give half the CUs a workgroup (measured at the north-star shape,
    // 256 slices: 1 tile per level 65 us, 2: 71, 3: 77, 4: 91, 6: 140)
    int64_t nt = std::max<int64_t>(1, (128 + slices - 1) / slices);
    if (const char *e = knob_str(K_NT_MIN)) nt = std::max(1, atoi(e));
    tp.nt_min = (int)std::min<int64_t>(nt, 256);
    // cells per level <= 2 * pixels + 2; tiles per level <= 2 * nt_l + 1
    const int64_t cells = 2LL * d.S + 2LL * d.L;
    const int64_t bound = 2LL * d.L * (tp.nt_min + 1) + 2LL * ((cells + kMaxTileCells - 1) / kMaxTileCells) + d.L;
    tp.tiles_bound = (int)std::min<int64_t>(bound, 0x3fffffff);
    tp.vgroups = 1;
    tp.hgroup = 1;
    return tp;
}

// Dynamic LDS of a sort workgroup: the record window (msda_bwd_cell_sort) when a tile's records are expected
// to fit one -- a tile never holds more than the level's Nq*P samples, about 1/nt of them when the level is
// cut into nt tiles -- else just the direct path's cursors (two workgroups per CU instead of one).
constexpr uint32_t kMaxSortWindow = 128 * 1024;
uint32_t sort_window_bytes(const Dims &d, const TileParams &tp, bool compact)
{
    // (Dims::taps_sorted needs the kept scan -- the only one that writes a sample's place into its record and the zeros of the
    // samples that get none -- and the kernel reads "a window is expected" off the size: a little more than the floor then)
    const uint32_t floor_bytes = kMaxTileCells * 4 + (d.taps_sorted ? 16u : 0u);
    int64_t want = (int64_t)d.Nq * d.P * (compact ? 8 : 16);
    if (tp.nt_min > 1) want = want / tp.nt_min + want / tp.nt_min / 4;
    if (const char *e = knob_str(K_SORT_WINDOW_KB)) want = std::min<int64_t>(atoll(e) * 1024, kMaxSortWindow);
    // more records than the largest window: placed window by window while that takes few rounds (the SD block's
    // 32768 samples per level: two), else the direct path (MMFS_SORT_ROUNDS=0: always the direct path)
    const int max_rounds = knob_int(K_SORT_ROUNDS, 4);
    if (want > kMaxSortWindow) return (kWindowRounds && want <= (int64_t)max_rounds * kMaxSortWindow) ? kMaxSortWindow : floor_bytes;
    return (uint32_t)std::max<int64_t>(floor_bytes, (want + 15) / 16 * 16);
}

// One trip of the scan with a window to sort into: the sort keeps its samples in registers (KeptScan) -- and can
// then read them from the op's own arrays, so that nothing has to be re-packed.  (MMFS_SORT_REPACK=1 re-packs anyway.)
// How the sort reads a query's samples of a level: NV 16-byte vectors per (virtual) query, G virtual queries per query.
// One or two vectors per query (P <= 8 of 16-bit storage): G = 1, every path of the sort takes them.  More (the reference's
// own speed test has 64 points per level): only the kept scan can, as G groups of NV vectors -- when all of them fit one
// trip and a window exists; else NV = 0, the scalar scan.  (MMFS_SORT_MANY_POINTS=0: always the scalar scan.)
// Tiles of few samples -- a level of a slice has Nq * P of them -- go to 256- or 512-lane workgroups (msda_bwd_cell_sort):
// every query still has a lane of the kept scan (Nq <= lanes * kScanUnroll), one group of vectors per query, and MANY
// slices (otherwise there is nothing to overlap with).  MMFS_SORT_SMALL=0: always 1024 lanes.
constexpr int kSmallThreads = 256, kMidThreads = 512;
int sort_lanes(const Dims &d, int vgroups, int nv)
{
    const char *e = knob_str(K_SORT_SMALL);                    // (the tests hold the variants to the oracle)
    if ((e && e[0] == '0') || nv <= 0 || vgroups != 1 || (int64_t)d.B * d.H * d.L < 512) return kThreads;
    const int64_t samples = (int64_t)d.Nq * d.P;
    if (samples <= 2048 && d.Nq <= kSmallThreads * kScanUnroll) return kSmallThreads;
    if (samples <= 8192 && d.Nq <= kMidThreads * kScanUnroll) return kMidThreads;      // (two per CU: the ViT-Adapter extractor)
    return kThreads;
}

struct KeptCfg { int nv, g; };
KeptCfg kept_config(int es, const Dims &d, const TileParams &tp, bool compact)
{
    KeptCfg c = {0, 1};
    const int loc_bytes = d.P * 2 * es;
    if (loc_bytes % 16 != 0) return c;
    const int vpq = loc_bytes / 16;
    if (vpq <= 2) { c.nv = vpq; return c; }
    const char *e = knob_str(K_SORT_MANY_POINTS);
    if ((e && e[0] == '0') || sort_window_bytes(d, tp, compact) <= kMaxTileCells * 4u) return c;
    if ((int64_t)d.Nq * vpq <= kThreads * kScanUnroll) { c.nv = 1; c.g = vpq; }
    else if (vpq % 2 == 0 && (int64_t)d.Nq * (vpq / 2) <= kThreads * kScanUnroll) { c.nv = 2; c.g = vpq / 2; }
    return c;
}

bool sort_keeps_samples(int dtype, const Dims &d, const TileParams &tp)
{
    const int es = dtype == 0 ? 4 : 2;
    const bool compact = es == 2 && tile_reduce_supported(dtype, d);
    const KeptCfg c = kept_config(es, d, tp, compact);
    return c.nv > 0 && (int64_t)d.Nq * c.g <= kThreads * kScanUnroll && sort_window_bytes(d, tp, compact) > kMaxTileCells * 4u;
}

int cell_stride_of(const Dims &d) { return 2 * d.S + 2 * d.L; }      // >= sum (H+1)(W+1)
// >= sum over levels of blocks * split, each level padded to kMaxSplit:
// blocks <= pixels + 1, and split * blocks <= blocks + 2 * (2.25 * Nq * P) / 128 + kMaxSplit
int64_t block_bound_of(const Dims &d)
{
    return (int64_t)d.S + (int64_t)d.L * (2 * kMaxSplit + 2 + ((int64_t)d.Nq * d.P * (kBH + 1) * (kBW + 1) / (kBH * kBW)) / 64);
}

struct Scratch {
    char *loc_t, *attn_t;
    uint32_t *cursor;
    CellHeader *hdr;
    uint2 *celltab;            // [B, H, cell_stride] {first record, count}
    uint4 *records;            // [B, H, L, Nq*P] {query, y, x, attention}, cell-sorted inside each tile
    OvfHeader *ovf;            // long lists: counters, slots, queued chunks, fp32 accumulators
    OvfSlot *oslots;
    OvfEntry *oentries;
    float *oacc;
    uint32_t cap_slots, cap_entries, cap_partials;
    // matrix-core reduce (msda_bwd_tile.hip): header, per-block info, queued extra items, fp32 partial tiles
    TileHeader *th;
    TileDesc *tdesc;
    TileItem *titems;
    uint32_t *slice_done, *n_extra;
    float *tpartials;
    uint32_t tile_cap_extra, tile_cap_partials;
    int tile_blocks_bound;     // >= 4x4 blocks of one (b, h)
    TapsDesc *xdesc;           // [B, H, tile_blocks_bound] (msda_bwd_taps_sorted.hip)
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
    s.cursor = (uint32_t *)p;    s.cursor_bytes = up(((int64_t)d.B * d.H * d.L + 2 * (int64_t)d.B * d.H) * 4);  p += s.cursor_bytes;   // + per slice: an arrival counter, the length of its queue of extra items
    s.hdr = (CellHeader *)p;
    p += up((int64_t)sizeof(CellHeader) + (int64_t)d.L * sizeof(LevelRow) + (int64_t)make_params(d).tiles_bound * sizeof(CTile));
    s.celltab = (uint2 *)p;      p += up((int64_t)d.B * d.H * cell_stride_of(d) * 8);
    s.records = (uint4 *)p;      p += up(pts * 16);
    // every sample is visited (kBH+1)(kBW+1)/(kBH kBW) times; a block is queued only after kCapRecords
    // of its records were walked in place, in chunks of kOvfChunk
    // a block is queued only after >= kCapRecords of its records were walked in place, in chunks of kOvfChunk
    const int64_t visits = pts * (kBH + 1) * (kBW + 1) / (kBH * kBW);
    // a queued block holds > kCapRecords records and is cut into ceil(n / kOvfChunk) chunks; only blocks
    // of >= 2 chunks need partials.  These bounds cannot be exceeded (the kernels nevertheless survive it:
    // the owner then finishes in place)
    s.cap_slots = (uint32_t)std::min<int64_t>(visits / kCapRecords + 64, 0x3fffffff);
    s.cap_entries = (uint32_t)std::min<int64_t>(visits / kOvfChunk + s.cap_slots, 0x3fffffff);
    s.cap_partials = (uint32_t)std::min<int64_t>(2 * (visits / kOvfChunk) + 64, 0x3fffffff);
    s.ovf = (OvfHeader *)p;      p += up(sizeof(OvfHeader));
    s.oslots = (OvfSlot *)p;     p += up((int64_t)s.cap_slots * sizeof(OvfSlot));
    s.oentries = (OvfEntry *)p;  p += up((int64_t)kOvfLanes * s.cap_entries * sizeof(OvfEntry));   // any lane may take all
    s.oacc = (float *)p;         p += up((int64_t)s.cap_partials * kNPX * d.D * 4);                       // partial sums
    // matrix-core reduce: blocks <= sum ceil(H/4) ceil(W/4) <= S/4 + L; a block of n > tile_chunk(D) records is cut into
    // ceil(n / tile_chunk(D)) items, so items and partial tiles are bounded by twice the visits / tile_chunk(D).  (The
    // bounds assume the AVERAGE 25/16 visits per sample; a block whose items or tiles do not fit is walked by one
    // wave alone -- slow, still correct.)
    s.th = nullptr; s.tdesc = nullptr; s.titems = nullptr; s.slice_done = nullptr; s.n_extra = nullptr; s.tpartials = nullptr; s.xdesc = nullptr;
    s.tile_cap_extra = s.tile_cap_partials = 0; s.tile_blocks_bound = 0;
    if (tile_reduce_supported(dtype, d)) {
        const int64_t tvisits = pts * (kTB + 1) * (kTB + 1) / (kTB * kTB);
        s.tile_blocks_bound = (int)std::min<int64_t>((int64_t)d.S / 4 + d.L + 1, 0x3fffffff);
        s.tile_cap_partials = (uint32_t)std::min<int64_t>(2 * (tvisits / tile_chunk(d)) + 64, 0x3fffffff);
        // queue places per (b, h) slice: twice what the slice's average share of the visits needs
        s.tile_cap_extra = (uint32_t)std::min<int64_t>(2 * ((int64_t)d.Nq * d.L * d.P * (kTB + 1) * (kTB + 1) / (kTB * kTB) / tile_chunk(d)) + 16, 0x3fffffff);
        s.th = (TileHeader *)p;      p += up(sizeof(TileHeader));
        s.tdesc = (TileDesc *)p;     p += up((int64_t)d.B * d.H * s.tile_blocks_bound * sizeof(TileDesc));
        s.titems = (TileItem *)p;    p += up((int64_t)d.B * d.H * s.tile_cap_extra * sizeof(TileItem));
        s.slice_done = s.cursor + (int64_t)d.B * d.H * d.L;      // (zeroed with the cursors by backward_value_prepare)
        s.n_extra = s.slice_done + (int64_t)d.B * d.H;
        s.tpartials = (float *)p;    p += up((int64_t)s.tile_cap_partials * kTB * kTB * d.D * 4);
        s.xdesc = (TapsDesc *)p;     p += up((int64_t)d.B * d.H * s.tile_blocks_bound * sizeof(TapsDesc));
    }
    s.total = p - (char *)workspace;
    return s;
}

// Rows of grad_value no level owns (a level table with gaps): zero, as the reference's zero-filled
// output leaves them (ms_deform_attn_cuda.cu:127).  Canonical tables have none: the kernel returns at once.
template <typename T>
__global__ void __launch_bounds__(256)
zero_uncovered_rows(T *__restrict__ grad_value, const CellHeader *__restrict__ hdr, const Dims d)
{
    if (hdr->pad[0]) return;
    const LevelRow *lv = level_rows(hdr);
    const int64_t row_elems = (int64_t)d.H * d.D;
    for (int64_t r = blockIdx.x; r < (int64_t)d.B * d.S; r += gridDim.x) {
        const int pix = (int)(r % d.S);
        bool owned = false;
        for (int l = 0; l < d.L; ++l) owned |= lv[l].Hl > 0 && lv[l].Wl > 0 && pix >= lv[l].lstart && pix < lv[l].lstart + lv[l].Hl * lv[l].Wl;
        if (owned) continue;
        for (int64_t i = threadIdx.x; i < row_elems; i += 256) grad_value[r * row_elems + i] = (T)0.f;
    }
}

template <typename T>
void launch_zero_uncovered(void *gv, const Scratch &sc, const Dims &d, hipStream_t st)
{
    hipLaunchKernelGGL((zero_uncovered_rows<T>), dim3(1024), dim3(256), 0, st, (T *)gv, sc.hdr, d);
}

TileReduceArgs tile_args(const Scratch &sc, const Dims &d)
{
    TileReduceArgs a;
    a.records = sc.records; a.celltab = sc.celltab; a.hdr = sc.hdr; a.cell_stride = cell_stride_of(d);
    a.th = sc.th; a.tdesc = sc.tdesc; a.titems = sc.titems; a.slice_done = sc.slice_done; a.n_extra = sc.n_extra; a.tpartials = sc.tpartials;
    a.blocks_bound = sc.tile_blocks_bound;
    a.xdesc = d.taps_sorted ? sc.xdesc : nullptr; a.g_loc = a.g_attn = nullptr;
    a.qshift = 0;
    if (d.taps_sorted) while ((1 << a.qshift) < d.P) ++a.qshift;
    return a;
}

static PlanArgs plan_args(const int64_t *shapes, const int64_t *start, const Scratch &sc, const Dims &d, const TileParams &tp)
{
    PlanArgs pa;
    pa.shapes = shapes; pa.start = start; pa.L = d.L; pa.S = d.S; pa.nt_min = tp.nt_min; pa.cap = tp.tiles_bound;
    pa.samples_per_level = (int64_t)d.Nq * d.P; pa.hdr = sc.hdr; pa.ovf_header = reinterpret_cast<uint32_t *>(sc.ovf);
    pa.cap_slots = sc.cap_slots; pa.cap_entries = sc.cap_entries; pa.cap_partials = sc.cap_partials;
    pa.th = sc.th; pa.tile_cap_extra = sc.tile_cap_extra; pa.tile_cap_partials = sc.tile_cap_partials;
    pa.loc_src = pa.attn_src = nullptr;
    pa.status = d.table_status;
    pa.stamp = header_stamp(d);
    return pa;
}

template <typename T, int NV>
hipError_t launch_sort(const int64_t *shapes, const int64_t *start, const Scratch &sc, const Dims &d, bool planned,
                       hipStream_t st, int vgroups, void *g_loc, void *g_attn)
{
    TileParams tp = make_params(d);
    tp.vgroups = vgroups;
    // Four heads whose weights share a 128-byte line of attn (L * P * element size <= 32 bytes per head; their loc words
    // share lines in pairs): their workgroups on one XCD -- north star: the sort fetched 66 MB for 25 MB of loc + attn,
    // 37.9 -> 34.6 us (profiles/r03_experiments.md r03ay).  Pairs of heads (SD / LLM geometry, 64 bytes of attn per
    // head) measured slower than head h on XCD h % 8: 147 -> 154 us.  MMFS_SORT_HGROUP: tuning
    const int env_hg = knob_int(K_SORT_HGROUP, -1);
    int hgroup = env_hg >= 0 ? std::min(env_hg, 4) : ((int64_t)d.L * d.P * (int64_t)sizeof(T) <= 32 ? 4 : 1);
    while (hgroup > 1 && d.H % hgroup) --hgroup;
    tp.hgroup = std::max(hgroup, 1);
    // (the grid's stride: the exact tile count when the caller knows the level table on the host -- the workgroups of a
    // bound-sized grid past the plan's tile count return at once, but 1024-lane workgroups are not free to start: north
    // star, 26 places per slice for 4 tiles: sort 35.7 -> 31.1 us, r06j)
    if (d.tiles_hint > 0 && d.tiles_hint <= tp.tiles_bound) tp.tiles_bound = d.tiles_hint;
    int64_t blocks = (int64_t)d.B * d.H * tp.tiles_bound;
    if (tp.hgroup > 1) blocks = ((int64_t)d.B * (d.H / tp.hgroup) + 7) / 8 * 8 * tp.hgroup * tp.tiles_bound;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    // (every cell of every level lies in exactly one tile, so the sort writes the whole cell table)
    if (!planned)
        hipLaunchKernelGGL(plan_cells_kernel, dim3(1), dim3(64), 0, st, plan_args(shapes, start, sc, d, tp));
    // the matrix-core reduce takes 8-byte records (its support test bounds Nq by 65536), the others 16-byte ones
    const bool compact = sizeof(T) == 2 && sc.th != nullptr;
    const uint32_t win = sort_window_bytes(d, tp, compact);
    TileReduceArgs ta = tile_args(sc, d);
    ta.g_loc = g_loc; ta.g_attn = g_attn;
    auto go_ts = [&](auto tag_compact, auto tag_threads, auto tag_ts) {
        constexpr bool C = decltype(tag_compact)::value;
        constexpr int THREADS = decltype(tag_threads)::value;
        constexpr bool TS = decltype(tag_ts)::value && C && NV > 0;       // (only the kept scan of the compact records serves it)
        static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_bwd_cell_sort<T, NV, C, THREADS, TS>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, kMaxSortWindow);
        (void)once;
        hipLaunchKernelGGL((msda_bwd_cell_sort<T, NV, C, THREADS, TS>), dim3((unsigned)blocks), dim3(THREADS), win, st,
                           (const T *)sc.loc_t, (const T *)sc.attn_t, sc.records, sc.cursor, sc.celltab, sc.hdr, d, tp,
                           cell_stride_of(d), win, ta);
    };
    // Two vectors of locations per query (P = 8 of 16-bit storage: the decoders' geometry) in a 1024-lane workgroup: the
    // samples' words do not fit next to their keys, so every placing pass reads loc / attn AGAIN -- and a tile of more
    // records than the window holds (the SD block: 32 768 samples per level) places in two passes: three reads of the
    // level's words in all (641 MB of traffic for 101 MB of input).  512 lanes with eight samples each keep the
    // words in registers (256 a lane): one read.  MMFS_SORT_WIDE=0: the 1024-lane kernel
    auto go_wide = [&](auto tag_ts) {
        constexpr bool TS = decltype(tag_ts)::value && NV > 0;
        constexpr int THREADS = kMidThreads, UNROLL = 2 * kScanUnroll;
        static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_bwd_cell_sort<T, NV, true, THREADS, TS, UNROLL>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, kMaxSortWindow);
        (void)once;
        hipLaunchKernelGGL((msda_bwd_cell_sort<T, NV, true, THREADS, TS, UNROLL>), dim3((unsigned)blocks), dim3(THREADS), win, st,
                           (const T *)sc.loc_t, (const T *)sc.attn_t, sc.records, sc.cursor, sc.celltab, sc.hdr, d, tp,
                           cell_stride_of(d), win, ta);
    };
    auto go = [&](auto tag_compact, auto tag_threads) {
        if (d.taps_sorted) go_ts(tag_compact, tag_threads, std::true_type());
        else go_ts(tag_compact, tag_threads, std::false_type());
    };
    const int lanes = sort_lanes(d, vgroups, NV);
    if constexpr ((NV == 2 || NV == 1) && sizeof(T) == 2) {
        const char *e = knob_str(K_SORT_WIDE);
        // (one-vector queries -- P = 4: the north star -- only when asked, "2": measured in r06x)
        if (compact && lanes == kThreads && !(e && e[0] == '0') && (NV == 2 || (e && e[0] == '2'))) {
            if (d.taps_sorted) go_wide(std::true_type()); else go_wide(std::false_type());
            return hipGetLastError();
        }
    }
    typedef std::integral_constant<bool, sizeof(T) == 2> Compact;
    typedef std::integral_constant<bool, false> Wide;
    typedef std::integral_constant<int, kThreads> Big;
    typedef std::integral_constant<int, kMidThreads> Mid;
    typedef std::integral_constant<int, kSmallThreads> Small;
    if (compact) { if (lanes == kSmallThreads) go(Compact(), Small()); else if (lanes == kMidThreads) go(Compact(), Mid()); else go(Compact(), Big()); }
    else { if (lanes == kSmallThreads) go(Wide(), Small()); else if (lanes == kMidThreads) go(Wide(), Mid()); else go(Wide(), Big()); }
    return hipGetLastError();
}

template <typename T>
hipError_t dispatch_sort(const int64_t *shapes, const int64_t *start, const Scratch &sc, const Dims &d, bool planned,
                         hipStream_t st, void *g_loc, void *g_attn)
{
    const KeptCfg c = kept_config((int)sizeof(T), d, make_params(d), sizeof(T) == 2 && sc.th != nullptr);
    switch (c.nv) {
        case 1: return launch_sort<T, 1>(shapes, start, sc, d, planned, st, c.g, g_loc, g_attn);
        case 2: return launch_sort<T, 2>(shapes, start, sc, d, planned, st, c.g, g_loc, g_attn);
        default: return launch_sort<T, 0>(shapes, start, sc, d, planned, st, 1, g_loc, g_attn);
    }
}

template <typename T, int LPS>
hipError_t launch_reduce(const Scratch &sc, const void *go, void *gv, const Dims &d, hipStream_t st)
{
    constexpr int GROUPS = kRThreads / LPS;
    const int64_t chunks64 = (block_bound_of(d) + GROUPS - 1) / GROUPS;
    if (chunks64 > 0x7fffffffLL) return hipErrorInvalidValue;
    const int chunks = (int)chunks64;
    const int64_t blocks = (int64_t)d.B * d.H * chunks;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    if ((int64_t)d.Nq * d.H * d.D * (int64_t)sizeof(T) <= kMaxSlabBytes)
        hipLaunchKernelGGL((msda_bwd_block_reduce<T, LPS, true>), dim3((unsigned)blocks), dim3(kRThreads), 0, st,
                           (const T *)go, (T *)gv, sc.records, sc.celltab, sc.hdr, d, chunks, cell_stride_of(d),
                           sc.ovf, sc.oslots, sc.oentries, sc.oacc);
    else
        hipLaunchKernelGGL((msda_bwd_block_reduce<T, LPS, false>), dim3((unsigned)blocks), dim3(kRThreads), 0, st,
                           (const T *)go, (T *)gv, sc.records, sc.celltab, sc.hdr, d, chunks, cell_stride_of(d),
                           sc.ovf, sc.oslots, sc.oentries, sc.oacc);
    // long lists (normally none: both kernels read the queue length on the device and return)
    const unsigned oblocks = std::min<unsigned>(sc.cap_entries, 512u) * kOvfLanes;
    if ((int64_t)d.Nq * d.H * d.D * (int64_t)sizeof(T) <= kMaxSlabBytes)
        hipLaunchKernelGGL((msda_bwd_block_overflow<T, LPS, true>), dim3(oblocks), dim3(kRThreads), 0, st,
                           (const T *)go, (T *)gv, sc.records, sc.ovf, sc.oslots, sc.oentries, sc.oacc, d);
    else
        hipLaunchKernelGGL((msda_bwd_block_overflow<T, LPS, false>), dim3(oblocks), dim3(kRThreads), 0, st,
                           (const T *)go, (T *)gv, sc.records, sc.ovf, sc.oslots, sc.oentries, sc.oacc, d);
    hipLaunchKernelGGL((msda_bwd_block_ovf_store<T>), dim3(std::min<unsigned>(sc.cap_slots, 1024u)), dim3(256), 0, st,
                       (T *)gv, sc.ovf, sc.oslots, sc.oacc, d);
    return hipGetLastError();
}

template <typename T>
hipError_t dispatch_reduce(const Scratch &sc, const void *go, void *gv, const Dims &d, hipStream_t st)
{
    constexpr int VEC = 16 / (int)sizeof(T);
    switch (d.D / VEC) {
#define MMFS_CASE(n) case n: return launch_reduce<T, n>(sc, go, gv, d, st);
        MMFS_CASE(4) MMFS_CASE(8) MMFS_CASE(16) MMFS_CASE(32)
#undef MMFS_CASE
        default: return hipErrorInvalidValue;
    }
}

}  // namespace

// Block-stationary grad_value: same preconditions as the pixel-stationary kernels plus a level
// count that fits the reduce's LDS table and lane groups of 4..32 lanes (D * sizeof(T) = 64..512 B).
bool bwd_value_block_supported(int dtype, const Dims &d)
{
    if (!bwd_value_tiled_supported(dtype, d)) return false;
    if (d.L > kMaxLevels) return false;
    const int vec = dtype == 0 ? 4 : 8;
    const int lps = d.D / vec;
    if (lps < 4 || lps > 32) return false;
    if (2LL * d.S + 2LL * d.L > 0x3fffffffLL) return false;
    if (const char *e = knob_str(K_VALUE_ALGO)) if (e[0] == 'p') return false;       // "pixel"
    const TileParams tp = make_params(d);
    return (int64_t)d.B * d.H * tp.tiles_bound <= 0x7fffffffLL;
}

int64_t bwd_value_block_workspace_bytes(int dtype, const Dims &d)
{
    return carve(nullptr, dtype, d).total;
}

// Where the plan's verdict on the level table lives (CellHeader::pad[2]: the device-side check refused it), for the
// float-atomic fallback of msda_bwd_refused.hip.
const int *value_table_refused_flag(void *workspace, int dtype, const Dims &d)
{
    return &carve(workspace, dtype, d).hdr->pad[2];
}

// (the workspace starts with the same three pieces as the pixel-stationary layout -- re-packed
// loc, re-packed attn, level cursors -- so backward_value_prepare serves both)

// Stage 2a for this generation: re-pack + cursors (+ the plan when the level table is at hand), one launch.
hipError_t backward_value_block_prepare(int dtype, const void *loc, const void *attn, const int64_t *shapes,
                                        const int64_t *start, void *workspace, const Dims &d, hipStream_t st)
{
    if (!bwd_value_block_supported(dtype, d)) return hipErrorInvalidValue;
    const int es = dtype == 0 ? 4 : 2;
    const Scratch sc = carve(workspace, dtype, d);
    auto gran = [](const void *a, const void *b, int chunk_bytes) {
        const bool al = (((uintptr_t)a | (uintptr_t)b) % 16) == 0;
        if (al && chunk_bytes % 16 == 0) return 16;
        if (al && chunk_bytes % 8 == 0) return 8;
        if (al && chunk_bytes % 4 == 0) return 4;
        return 2;
    };
    const int cl = d.P * 2 * es, ca = d.P * es;
    if (ca % 2) return hipErrorInvalidValue;
    const int vb_l = gran(loc, sc.loc_t, cl), vb_a = gran(attn, sc.attn_t, ca);
    const int vpc_l = cl / vb_l, vpc_a = ca / vb_a;
    const int64_t groups = (int64_t)d.B * d.Nq * d.H * d.L;
    int64_t total_l = groups * vpc_l, total_a = groups * vpc_a;
    PlanArgs pa;
    pa.shapes = nullptr;
    if (shapes != nullptr && start != nullptr) {
        const TileParams tp = make_params(d);
        pa = plan_args(shapes, start, sc, d, tp);
        // the sort can read loc / attn where they are (the header tells it): no re-pack
        const bool aligned = (uintptr_t)loc % 16 == 0 && (uintptr_t)attn % 8 == 0;
        const char *e = knob_str(K_SORT_REPACK);
        if (aligned && sort_keeps_samples(dtype, d, tp) && !(e && e[0] == '1')) {
            pa.loc_src = loc; pa.attn_src = attn;
            total_l = total_a = 0;
        }
    }
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((total_l + 255) / 256, 256 * 64));
    hipLaunchKernelGGL(msda_bwd_prepare, dim3((unsigned)blocks), dim3(256), 0, st,
                       (const char *)loc, (char *)sc.loc_t, vb_l, vpc_l, total_l,
                       (const char *)attn, (char *)sc.attn_t, vb_a, vpc_a, total_a,
                       d.Nq, d.H * d.L, sc.cursor, sc.cursor_bytes / 4, pa);
    return hipGetLastError();
}

bool value_prepare_job(int dtype, const void *loc, const void *attn, const int64_t *shapes, const int64_t *start,
                       void *workspace, const Dims &d, blk::PrepareJob *job)
{
    const char *off = knob_str(K_PREPARE_IN_TAPS);                   // "0": always its own launch
    if (off && off[0] == '0') return false;
    if (!bwd_value_tiled_supported(dtype, d) || !bwd_value_block_supported(dtype, d) || !shapes || !start) return false;
    const TileParams tp = make_params(d);
    const Scratch sc = carve(workspace, dtype, d);
    job->pa = plan_args(shapes, start, sc, d, tp);
    job->cursor = sc.cursor; job->cursor_words = sc.cursor_bytes / 4;
    const bool aligned = (uintptr_t)loc % 16 == 0 && (uintptr_t)attn % 8 == 0;
    const char *e = knob_str(K_SORT_REPACK);
    if (!aligned || !sort_keeps_samples(dtype, d, tp) || (e && e[0] == '1')) return false;
    job->pa.loc_src = loc; job->pa.attn_src = attn;
    return true;
}

hipError_t backward_value_block_sort(int dtype, const int64_t *shapes, const int64_t *start, void *workspace,
                                     const Dims &d, bool planned, hipStream_t st, void *g_loc, void *g_attn)
{
    if (!bwd_value_block_supported(dtype, d)) return hipErrorInvalidValue;
    if (d.taps_sorted && !(taps_sorted_supported(dtype, d) && g_loc && g_attn)) return hipErrorInvalidValue;
    const Scratch sc = carve(workspace, dtype, d);
    switch (dtype) {
        case 0: return dispatch_sort<float>(shapes, start, sc, d, planned, st, g_loc, g_attn);
        case 1: return dispatch_sort<half_t>(shapes, start, sc, d, planned, st, g_loc, g_attn);
        case 2: return dispatch_sort<bf16_t>(shapes, start, sc, d, planned, st, g_loc, g_attn);
        default: return hipErrorInvalidValue;
    }
}

// grad_loc / grad_attn from the cell-sorted records (msda_bwd_taps_sorted.hip) need: the matrix-core reduce's 8-byte records
// (16-bit storage, D in {32, 64, 128}), the sample's place query * P + point in the record's 16 query bits (P a power of two),
// the kept scan of the sort (the only one that writes that place and the zeros of samples without a record), no level
// taken away from the sort.  MMFS_TAPS_ALGO = vec | mma: never.
bool taps_sorted_supported(int dtype, const Dims &d)
{
    if (!bwd_value_block_supported(dtype, d) || !tile_reduce_supported(dtype, d)) return false;
    if (d.P <= 0 || (d.P & (d.P - 1)) || (int64_t)d.Nq * d.P > 65536) return false;
    if (const char *e = knob_str(K_TAPS_ALGO)) if (e[0] == 'v' || e[0] == 'm' || e[0] == 'g') return false;     // vec / mma / gather
    return sort_keeps_samples(dtype, d, make_params(d));
}

// Tiles the plan will make for a level table the HOST knows (Dims::tiles_hint): the plan's own rule per level.
int sort_tiles_exact(int dtype, const Dims &d, const int64_t *host_shapes)
{
    if (!host_shapes || !bwd_value_block_supported(dtype, d)) return 0;
    const TileParams tp = make_params(d);
    int64_t n = 0;
    for (int l = 0; l < d.L; ++l) n += level_tiling(host_shapes[2 * l], host_shapes[2 * l + 1], tp.nt_min).n;
    return n > 0 && n <= tp.tiles_bound ? (int)n : 0;
}

// the pieces of the workspace msda_bwd_taps_sorted.hip reads
blk::TileReduceArgs blk::taps_sorted_args(void *workspace, int dtype, const Dims &d, uint32_t *cap_extra)
{
    const Scratch sc = carve(workspace, dtype, d);
    *cap_extra = sc.tile_cap_extra;
    return tile_args(sc, d);
}

hipError_t backward_value_block_reduce(int dtype, const void *grad_out, void *grad_value, void *workspace,
                                       const Dims &d, bool all_rows_owned, hipStream_t st)
{
    if (!bwd_value_block_supported(dtype, d)) return hipErrorInvalidValue;
    const Scratch sc = carve(workspace, dtype, d);
    // rows no level owns (only a table the HOST has not vouched for can have them)
    if (!all_rows_owned) switch (dtype) {
        case 0: launch_zero_uncovered<float>(grad_value, sc, d, st); break;
        case 1: launch_zero_uncovered<half_t>(grad_value, sc, d, st); break;
        case 2: launch_zero_uncovered<bf16_t>(grad_value, sc, d, st); break;
        default: break;
    }
    if (sc.th != nullptr) {
        const TileReduceArgs a = tile_args(sc, d);
        return tile_reduce(dtype, grad_out, grad_value, a, d, sc.tile_cap_extra, st);
    }
    switch (dtype) {
        case 0: return dispatch_reduce<float>(sc, grad_out, grad_value, d, st);
        case 1: return dispatch_reduce<half_t>(sc, grad_out, grad_value, d, st);
        case 2: return dispatch_reduce<bf16_t>(sc, grad_out, grad_value, d, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mmfs

// Host-only self-check of the sort's tiling rules for ONE level (tests/test_plan_cpu.py sweeps it): the very
// functions the device plan uses (level_tiling, tile_local_blocks, block_is_tile_local).  Returns 0 when
// the tiles cover every cell exactly once within the counters' capacity AND every 4x4 block is planned exactly
// once -- by the one tile that holds its five cell rows, or, on a seam, by the slice's last workgroup;
// else a positive count of what is wrong.  -1: the level owns no tile (empty / refused extents).
extern "C" int mmfs_msda_plan_selfcheck(int64_t Hl, int64_t Wl, int nt_min)
{
    using namespace mmfs::blk;
    const LevelTiling lt = level_tiling(Hl, Wl, nt_min);
    if (lt.n == 0) return -1;
    const int Hc = (int)Hl + 1, Wc = (int)Wl + 1;
    LevelRow lr = {};
    lr.Hl = (int)Hl; lr.Wl = (int)Wl;
    lr.nbx4 = ((int)Wl + kTB - 1) / kTB; lr.nby4 = ((int)Hl + kTB - 1) / kTB;
    lr.band = lt.C == Wc ? lt.R : 0;
    int wrong = 0;
    int64_t covered = 0, n = 0;
    std::vector<int> planned((size_t)lr.nby4, 0);                  // per block row: tiles that plan it
    for (int ya = 0; ya < Hc; ya += lt.R)
        for (int xa = 0; xa < Wc; xa += lt.C, ++n) {
            CTile t;
            t.level = 0; t.Hl = lr.Hl; t.Wl = lr.Wl; t.cbase = 0;
            t.ya = ya; t.yb = std::min(Hc, ya + lt.R); t.xa = xa; t.xb = std::min(Wc, xa + lt.C);
            const int64_t cells = (int64_t)(t.yb - t.ya) * (t.xb - t.xa);
            if (cells > kMaxTileCells) ++wrong;
            covered += cells;
            if (lr.band == 0) continue;                             // (tiles that are not whole rows plan nothing)
            int by_lo;
            const int nloc = tile_local_blocks(t, lr, &by_lo);
            if (nloc % lr.nbx4 != 0 || nloc > 1024) ++wrong;        // whole block rows; one block per sort lane
            for (int by = by_lo; by < by_lo + nloc / lr.nbx4; ++by) {
                if (!block_is_tile_local(lr, by)) ++wrong;          // the two rules must agree
                ++planned[(size_t)by];
            }
        }
    if (n != lt.n || covered != (int64_t)Hc * Wc) ++wrong;
    for (int by = 0; by < lr.nby4; ++by)
        if (planned[(size_t)by] + (block_is_tile_local(lr, by) ? 0 : 1) != 1) ++wrong;
    return wrong;
}

#ifdef MMFS_PROFILE_SORT
extern "C" int mmfs_debug_sort_profile(unsigned long long *out, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(mmfs::g_sort_prof), 64 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        unsigned long long z[64] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(mmfs::g_sort_prof), z, sizeof z);
    }
    return (int)e;
}
extern "C" int mmfs_debug_sort_timeline(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mmfs::g_sort_wg), 4096 * 3 * sizeof(unsigned long long));
}
#endif
