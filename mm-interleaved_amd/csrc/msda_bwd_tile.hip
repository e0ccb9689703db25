// msda_bwd_tile.hip -- grad_value of multi-scale deformable attention on the matrix cores.
//
// Third generation of the owner-computes grad_value (reference: atomicAdd per sample, corner and
// channel, ms_deform_im2col_cuda.cuh:128-155; fp32 accumulation, cast at the end,
// ms_deform_attn_cuda.cu:122-165).  The 2x2-block reduce of msda_bwd_block.hip turned out to be
// bound by its vector instructions, not by the row gather: the rocprof counters put it at 77 % VALU
// busy while its grad_out rows arrive at 10 TB/s, and tools/ubench/gather2.hip shows random 256-byte
// rows out of an L2-resident slice moving at 34 TB/s (profiles/r02a_ubench_gather2.log).  Per record
// it unpacks a 16-bit row and runs 4 pixels x 8 FMAs per lane, 56 % of them against a zero weight.
//
// Here the sum over a block's records IS a matrix product and runs as one:
//
//      grad_value[pixel, :] = sum_records  W[pixel, record] * grad_out[query(record), :]
//
//   * a wave owns a 4x4 block of one level's pixels (25/16 = 1.56 row reads per sample instead of
//     2.25) and walks the cell-sorted records of the 5x5 cells whose footprints touch it: five
//     contiguous runs of the record list (the sort's tiles are whole cell rows), seen as one list;
//   * 16 records per step.  Their grad_out rows travel global -> LDS by DMA (global_load_lds, no
//     registers, no vector instructions; 3 steps in flight), their records too (5 slots);
//   * the rows are the B operand of v_mfma_f32_32x32x16_{bf16,f16} (K = record, N = channel), read
//     out of LDS with the transposing ds_read_b64_tr_b16; the 16-byte chunks of a row are stored
//     XOR-swizzled (the swizzle is applied to the DMA's SOURCE address, the LDS image of a DMA is
//     lane-linear) so that the four rows a 16-lane group reads together sit in different banks;
//   * the A operand is the 32 x 16 weight tile: rows 0..15 the leading 16 bits of the fp32 weight of
//     (pixel, record), rows 16..31 the rounded remainder -- hi + lo carries >= 16 significant bits,
//     well inside the storage type's rounding, and costs nothing: M = 32 has room for both.  The
//     tile is built in LDS by 64 lanes = 16 records x 4 corners (one weight each, two 2-byte writes);
//   * accumulators: D/32 x 16 fp32 registers per lane.  Epilogue: hi + lo rows added, lane pairs
//     exchange one value so that every lane stores whole dwords.
//   * A list longer than kTileChunk records (small levels, hot spots -- every text token of the LLM
//     path samples around the same reference point) is cut into work items planned on the device; a
//     block of several items leaves fp32 partial tiles that a last small kernel adds up and rounds.
//
// Semantics that differ from the gather kernels, by construction of a matrix product: a non-finite
// grad_out element reaches all 16 pixels of the blocks its sample touches (0 * Inf = NaN), not only
// the sample's four corners.  Finite inputs: same products, fp32 sums in a different order.
#include "msda_bwd_block.h"
#include "msda_launch.h"
#include <cstdlib>

namespace mmfs {
namespace blk {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int kKS = 16;                 // records per MFMA step (K of 32x32x16)
constexpr int kStages = 3;              // row slots: one being multiplied, two in flight
constexpr int kRecSlots = 5;            // record slots: fetched four steps ahead of their weights
constexpr int kQueueWgs = 256 * kTileLanes;   // workgroups that walk the queue of extra items
constexpr uint32_t kVoidPart = 0xffffffffu;

template <typename T> struct TileMma;
template <> struct TileMma<bf16_t> {
    static __device__ __forceinline__ f32x16 run(const s16x8 &a, const s16x8 &b, const f32x16 &c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    // fp32 weight -> leading 16 bits, rounded remainder (weights are finite: products of fractions
    // and a finite attention weight... an infinite attention weight stays infinite in hi, lo = 0)
    static __device__ __forceinline__ void split(float w, uint16_t &hi, uint16_t &lo) {
        const uint32_t h = __float_as_uint(w) & 0xffff0000u;
        const bool special = (__float_as_uint(w) & 0x7f800000u) == 0x7f800000u;      // Inf / NaN
        const float r = special ? 0.f : w - __uint_as_float(h);                       // exact
        hi = (uint16_t)((w != w ? 0x7fc00000u : h) >> 16);
        lo = __builtin_bit_cast(uint16_t, (__bf16)r);
    }
    static __device__ __forceinline__ uint32_t pack2(float a, float b) { return Vec16<bf16_t>::pk(a, b); }
};
template <> struct TileMma<half_t> {
    static __device__ __forceinline__ f32x16 run(const s16x8 &a, const s16x8 &b, const f32x16 &c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ void split(float w, uint16_t &hi, uint16_t &lo) {
        const float c = w != w ? w : fminf(fmaxf(w, -65504.f), 65504.f);
        const _Float16 h = (_Float16)c;
        const _Float16 l = (_Float16)(c - (float)h);
        hi = __builtin_bit_cast(uint16_t, h);
        lo = __builtin_bit_cast(uint16_t, l);
    }
    static __device__ __forceinline__ uint32_t pack2(float a, float b) { return Vec16<half_t>::pk(a, b); }
};

template <int D> struct TileGeom {
    static constexpr int RB = D * 2;               // bytes of a grad_out row (one head)
    static constexpr int LPR = RB / 16;            // 16-byte chunks (= DMA lanes) per row
    static constexpr int RPI = 64 / LPR;           // rows per DMA instruction
    static constexpr int NR = kKS / RPI;           // DMA instructions per step
    static constexpr int NB = D / 32;              // 32-channel column blocks = MFMAs per step
    static constexpr int RP = 256 / RB;            // rows per 256 bytes (one pass over the 64 banks)
    static constexpr int SLOT = kKS * RB;          // bytes of a row slot
    static constexpr int LDS_BYTES = kStages * SLOT + kRecSlots * 256 + 1024;
    // chunk swizzle of row r (row index inside its slot): the 4 rows a 16-lane group of a transposing
    // read touches together must land in different banks
    static __device__ __forceinline__ int swz(int r) { return 4 * ((r / RP) % (4 / RP)); }
};

#define MMFS_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n) : "memory")

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

__device__ __forceinline__ void dma16(const void *src, void *lds_dst)
{
    __builtin_amdgcn_global_load_lds((glb_void_t *)src, (lds_void_t *)lds_dst, 16, 0, 0);
}

__device__ __forceinline__ int mfma_row32(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// level of a 4x4 block (uniform over the wave)
__device__ __forceinline__ int level_of_block4(const LevelRow *__restrict__ lv, int L, int blk, int lane)
{
    for (int l0 = 0; l0 < L; l0 += 64) {
        const int l = l0 + lane;
        bool hit = false;
        if (l < L) {
            const int bb = lv[l].bbase4, cnt = lv[l].nbx4 * lv[l].nby4;
            hit = blk >= bb && blk < bb + cnt;
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
        if (m) return l0 + (int)__builtin_ctzll(m);
    }
    return -1;
}

// the five record runs of a block (one per cell row) as one list: run r holds [pre[r], pre[r+1])
struct Runs5 { int first[5]; int pre[6]; };

__device__ __forceinline__ Runs5 block_runs(const uint2 *__restrict__ tab, const LevelRow &lr, int by, int bx, int lane)
{
    uint2 ent = make_uint2(0u, 0u);
    if (lane < 25) {
        const int cy = kTB * by + lane / 5, cx = kTB * bx + lane % 5;
        if (cy <= lr.Hl && cx <= lr.Wl) ent = tab[cy * (lr.Wl + 1) + cx];
    }
    Runs5 r;
    r.pre[0] = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        r.first[j] = __builtin_amdgcn_readlane((int)ent.x, 5 * j);
        int c = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) c += __builtin_amdgcn_readlane((int)ent.y, 5 * j + i);
        r.pre[j + 1] = r.pre[j] + c;
    }
    return r;
}

struct ItemArgs { int b, h, blk, part; bool whole, partial_out; uint32_t pidx; };

template <typename T, int D>
__device__ __forceinline__ void tile_item(const T *__restrict__ grad_out, T *__restrict__ grad_value,
                                          const TileReduceArgs &a, const Dims &d, const ItemArgs &it,
                                          unsigned char *__restrict__ rows, unsigned char *__restrict__ recs,
                                          unsigned char *__restrict__ atile)
{
    typedef TileGeom<D> G;
    typedef TileMma<T> M;
    const int lane = threadIdx.x;
    const LevelRow *lv = level_rows(a.hdr);
    const int level = level_of_block4(lv, d.L, it.blk, lane);
    if (level < 0) return;
    const LevelRow lr = lv[level];
    const int rel = it.blk - lr.bbase4;
    const int by = rel / lr.nbx4, bx = rel - by * lr.nbx4;
    const int64_t bh = (int64_t)it.b * d.H + it.h;
    const Runs5 ru = block_runs(a.celltab + bh * a.cell_stride + lr.cbase, lr, by, bx, lane);
    const int n = ru.pre[5];
    const int e0 = it.part * kTileChunk;
    const int e1 = it.whole ? n : min(n, e0 + kTileChunk);
    const int nks = e1 > e0 ? (e1 - e0 + kKS - 1) / kKS : 0;

    const int64_t HDB = (int64_t)d.H * d.D * (int64_t)sizeof(T);                  // bytes between consecutive queries
    const char *gslice = (const char *)(grad_out + ((int64_t)it.b * d.Nq * d.H + it.h) * d.D);
    const int64_t zero_off = (const char *)a.th->zero_row - gslice;
    const uint4 *null_rec = &a.th->null_rec;

    f32x16 acc[G::NB];
#pragma unroll
    for (int nb = 0; nb < G::NB; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;

    // records of step s -> record slot s % kRecSlots (16 lanes, one record each)
    auto issue_records = [&](int s) {
        if (lane < kKS) {
            const int e = e0 + kKS * s + lane;
            const uint4 *src = null_rec;
            if (e < e1) {
                int j = 0, del = ru.first[0] - ru.pre[0];
#pragma unroll
                for (int k = 1; k < 5; ++k)
                    if (e >= ru.pre[k]) { j = k; del = ru.first[k] - ru.pre[k]; }
                (void)j;
                src = a.records + (uint32_t)(e + del);
            }
            dma16(src, recs + (s % kRecSlots) * 256);
        }
    };
    // grad_out rows of step s -> row slot s % kStages (needs the step's records in LDS)
    auto issue_rows = [&](int s) {
        const uint32_t *rq = reinterpret_cast<const uint32_t *>(recs + (s % kRecSlots) * 256);
        unsigned char *slot = rows + (s % kStages) * G::SLOT;
        uint32_t q[G::NR];
#pragma unroll
        for (int u = 0; u < G::NR; ++u) q[u] = rq[(u * G::RPI + lane / G::LPR) * 4];      // (a void record reads q = 0)
#pragma unroll
        for (int u = 0; u < G::NR; ++u) {
            const int rr = u * G::RPI + lane / G::LPR;
            const int chunk = (lane % G::LPR) ^ G::swz(rr);
            const bool valid = e0 + kKS * s + rr < e1;
            // branch-free: a row past the end of the list is the zero row of the header
            const int64_t off = valid ? (int64_t)q[u] * HDB : zero_off;
            dma16(gslice + off + chunk * 16, slot + u * 1024);
        }
    };

    if (nks > 0) issue_records(0);
    if (nks > 1) issue_records(1);
    for (int i = -2; i < nks; ++i) {
        // wait for records(i+2) and rows(i): everything issued before them, i.e. all but
        // rows(i+1) [NR ops, issued in the previous round] and records(i+3) [1 op, likewise]
        const bool rows_next = i >= -1 && i + 1 < nks, rec_next = i + 3 < nks;
        if (rows_next && rec_next) MMFS_WAIT_VM(G::NR + 1);
        else if (rows_next) MMFS_WAIT_VM(G::NR);
        else if (rec_next) MMFS_WAIT_VM(1);
        else MMFS_WAIT_VM(0);
        if (i + 2 < nks) issue_rows(i + 2);
        if (i + 4 < nks) issue_records(i + 4);
        if (i < 0) continue;

        // ---- weight tile of step i: 16 records x 4 corners, one weight per lane
        reinterpret_cast<uint4 *>(atile)[lane] = make_uint4(0u, 0u, 0u, 0u);
        {
            const int r = lane >> 2, c = lane & 3;
            const uint4 rec = reinterpret_cast<const uint4 *>(recs + (i % kRecSlots) * 256)[r];
            const float y = __uint_as_float(rec.y), x = __uint_as_float(rec.z), av = __uint_as_float(rec.w);
            const float yf = floorf(y), xf = floorf(x);
            const float fy = y - yf, fx = x - xf;
            const int iy = (int)yf + (c >> 1) - kTB * by, ix = (int)xf + (c & 1) - kTB * bx;
            const float wy = (c >> 1) ? fy : 1.f - fy, wx = (c & 1) ? fx : 1.f - fx;
            const float wgt = wy * wx * av;
            if ((unsigned)iy < (unsigned)kTB && (unsigned)ix < (unsigned)kTB) {
                uint16_t hi, lo;
                M::split(wgt, hi, lo);
                const int m = iy * kTB + ix;
                reinterpret_cast<uint16_t *>(atile)[m * kKS + r] = hi;
                reinterpret_cast<uint16_t *>(atile)[(16 + m) * kKS + r] = lo;
            }
        }
        const s16x8 A = *reinterpret_cast<const s16x8 *>(atile + (lane & 31) * 32 + (lane >> 5) * 16);

        // ---- rows of step i as B operands, one MFMA per 32 channels
        const unsigned char *slot = rows + (i % kStages) * G::SLOT;
        const int g4 = lane >> 4, j16 = lane & 15;
#pragma unroll
        for (int nb = 0; nb < G::NB; ++nb) {
            s16x8 B;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int krow = 8 * (g4 >> 1) + 4 * t + (j16 >> 2);
                const int cb = nb * 64 + (g4 & 1) * 32 + (j16 & 3) * 8;             // byte offset inside the row
                const int off = krow * G::RB + (((cb >> 4) ^ G::swz(krow)) << 4) + (cb & 15);
                const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4 *)(slot + off));
                B[4 * t] = v[0]; B[4 * t + 1] = v[1]; B[4 * t + 2] = v[2]; B[4 * t + 3] = v[3];
            }
            acc[nb] = M::run(A, B, acc[nb]);
        }
    }

    // ---- epilogue: hi + lo rows, lane pairs exchange so that every lane stores whole dwords
    const bool odd = lane & 1;
    const int ch0 = (lane & 31) & ~1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int pa = mfma_row32(r, lane), pb = mfma_row32(r + 4, lane);      // pixels 0..15 of the block
        const int p = odd ? pb : pa;
        const int y = kTB * by + (p >> 2), x = kTB * bx + (p & 3);
#pragma unroll
        for (int nb = 0; nb < G::NB; ++nb) {
            const float va = acc[nb][r] + acc[nb][r + 8], vb = acc[nb][r + 4] + acc[nb][r + 12];
            const float recv = __shfl_xor(odd ? va : vb, 1, 64);
            const float lo_ch = odd ? recv : va, hi_ch = odd ? vb : recv;       // channels ch0, ch0 + 1 of pixel p
            const int ch = nb * 32 + ch0;
            if (it.partial_out) {
                float2 *o = reinterpret_cast<float2 *>(a.tpartials + ((int64_t)it.pidx * (kTB * kTB) + p) * D + ch);
                *o = make_float2(lo_ch, hi_ch);
            } else if (y < lr.Hl && x < lr.Wl) {
                T *o = grad_value + (((int64_t)it.b * d.S + lr.lstart + y * lr.Wl + x) * d.H + it.h) * d.D + ch;
                *reinterpret_cast<uint32_t *>(o) = M::pack2(lo_ch, hi_ch);
            }
        }
    }
}

// One wave per work item.  The first kQueueWgs workgroups walk the queue of extra items (the later
// parts of long lists: started first, they are the long poles); every other workgroup is one block.
template <typename T, int D>
__global__ void __launch_bounds__(64, 3)
msda_bwd_tile_reduce(const T *__restrict__ grad_out, T *__restrict__ grad_value, const TileReduceArgs a,
                     const Dims d, const int blocks_grid)
{
    typedef TileGeom<D> G;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[G::LDS_BYTES];
    unsigned char *rows = lds, *recs = lds + kStages * G::SLOT, *atile = recs + kRecSlots * 256;
    int w = blockIdx.x;
    if (w < kQueueWgs) {
        const int ql = w % kTileLanes;
        const uint32_t n = min(a.th->n_extra[ql], a.th->cap_extra);
        for (uint32_t i = (uint32_t)(w / kTileLanes); i < n; i += kQueueWgs / kTileLanes) {
            const TileItem ti = a.titems[(size_t)ql * a.th->cap_extra + i];
            if (ti.part == kVoidPart) continue;            // a reservation its block could not use
            ItemArgs it;
            it.b = (int)(ti.bh / (uint32_t)d.H); it.h = (int)(ti.bh % (uint32_t)d.H);
            it.blk = (int)ti.blk; it.part = (int)ti.part; it.whole = false; it.partial_out = true; it.pidx = ti.pidx;
            tile_item<T, D>(grad_out, grad_value, a, d, it, rows, recs, atile);
        }
        return;
    }
    w -= kQueueWgs;
    const int nblk = a.hdr->n_blocks4;
    ItemArgs it;
    it.h = w % d.H;
    const int t = w / d.H;
    const int j = t % blocks_grid;
    it.b = t / blocks_grid;
    if (j >= nblk) return;
    it.blk = nblk - 1 - j;                      // coarse levels (long lists) first
    const TileInfo info = a.tinfo[((int64_t)it.b * d.H + it.h) * a.blocks_bound + it.blk];
    it.part = 0; it.whole = info.parts <= 1; it.partial_out = info.parts > 1; it.pidx = info.pbase;
    tile_item<T, D>(grad_out, grad_value, a, d, it, rows, recs, atile);
}

// One thread per (b, h, block): length of the block's list -> number of work items; the extra ones
// are queued (one queue lane per XCD, keyed by h like every other kernel's head -> XCD affinity).
__global__ void __launch_bounds__(256)
msda_bwd_tile_plan(const TileReduceArgs a, const Dims d, const int blocks_grid)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int nblk = a.hdr->n_blocks4;
    const int blk = (int)(idx % blocks_grid);
    const int64_t bh = idx / blocks_grid;
    if (bh >= (int64_t)d.B * d.H || blk >= nblk) return;
    const LevelRow *lv = level_rows(a.hdr);
    int level = 0;
    while (level + 1 < d.L && blk >= lv[level + 1].bbase4) ++level;
    while (level < d.L && lv[level].nbx4 * lv[level].nby4 == 0) ++level;         // (empty levels own no block)
    TileInfo info;
    info.parts = 1; info.pbase = 0;
    if (level < d.L) {
        const LevelRow lr = lv[level];
        const int rel = blk - lr.bbase4, by = rel / lr.nbx4, bx = rel - by * lr.nbx4;
        const uint2 *tab = a.celltab + bh * a.cell_stride + lr.cbase;
        int64_t n = 0;
        for (int dy = 0; dy <= kTB; ++dy)
            for (int dx = 0; dx <= kTB; ++dx) {
                const int cy = kTB * by + dy, cx = kTB * bx + dx;
                if (cy <= lr.Hl && cx <= lr.Wl) n += tab[cy * (lr.Wl + 1) + cx].y;
            }
        const uint32_t parts = (uint32_t)((n + kTileChunk - 1) / kTileChunk);
        if (parts > 1) {
            const int ql = (int)(bh % d.H) % kTileLanes;
            const uint32_t pb = atomicAdd(&a.th->n_partials, parts);
            const uint32_t eb = atomicAdd(&a.th->n_extra[ql], parts - 1);
            if (pb + parts <= a.th->cap_partials && eb + parts - 1 <= a.th->cap_extra) {
                info.parts = parts; info.pbase = pb;
                for (uint32_t p = 1; p < parts; ++p) {
                    TileItem ti;
                    ti.bh = (uint32_t)bh; ti.blk = (uint32_t)blk; ti.part = p; ti.pidx = pb + p;
                    a.titems[(size_t)ql * a.th->cap_extra + eb + p - 1] = ti;
                }
            } else if (eb < a.th->cap_extra) {
                // reserved queue entries that cannot be used must read as "nothing to do"
                for (uint32_t p = 1; p < parts && eb + p - 1 < a.th->cap_extra; ++p) {
                    TileItem ti;
                    ti.bh = (uint32_t)bh; ti.blk = (uint32_t)blk; ti.part = kVoidPart; ti.pidx = 0;
                    a.titems[(size_t)ql * a.th->cap_extra + eb + p - 1] = ti;
                }
            }
        }
    }
    a.tinfo[bh * a.blocks_bound + blk] = info;
}

// Blocks of several items: partial tiles -> grad_value rows.
template <typename T, int D>
__global__ void __launch_bounds__(64)
msda_bwd_tile_finalize(T *__restrict__ grad_value, const TileReduceArgs a, const Dims d, const int blocks_grid)
{
    typedef TileMma<T> M;
    const int lane = threadIdx.x;
    const int w = blockIdx.x;
    const int h = w % d.H, t = w / d.H, blk = t % blocks_grid, b = t / blocks_grid;
    if (blk >= a.hdr->n_blocks4) return;
    const TileInfo info = a.tinfo[((int64_t)b * d.H + h) * a.blocks_bound + blk];
    if (info.parts <= 1) return;
    const LevelRow *lv = level_rows(a.hdr);
    const int level = level_of_block4(lv, d.L, blk, lane);
    if (level < 0) return;
    const LevelRow lr = lv[level];
    const int rel = blk - lr.bbase4, by = rel / lr.nbx4, bx = rel - by * lr.nbx4;
    for (int p = 0; p < kTB * kTB; ++p) {
        const int y = kTB * by + (p >> 2), x = kTB * bx + (p & 3);
        if (y >= lr.Hl || x >= lr.Wl) continue;
        for (int ch = lane * 2; ch < D; ch += 128) {
            float s0 = 0.f, s1 = 0.f;
            for (uint32_t c = 0; c < info.parts; ++c) {
                const float2 v = *reinterpret_cast<const float2 *>(
                    a.tpartials + ((int64_t)(info.pbase + c) * (kTB * kTB) + p) * D + ch);
                s0 += v.x; s1 += v.y;
            }
            T *o = grad_value + (((int64_t)b * d.S + lr.lstart + y * lr.Wl + x) * d.H + h) * d.D + ch;
            *reinterpret_cast<uint32_t *>(o) = M::pack2(s0, s1);
        }
    }
}

template <typename T, int D>
hipError_t launch_tile(const void *go, void *gv, const TileReduceArgs &a, const Dims &d, hipStream_t st)
{
    // the exact block count is only known on the device; the grid takes the caller's hint (host copy
    // of the level table) or the bound, surplus workgroups return at once
    const int blocks_grid = d.blocks4 > 0 ? std::min(d.blocks4, a.blocks_bound) : a.blocks_bound;
    const int64_t items = (int64_t)d.B * d.H * blocks_grid;
    if (items + kQueueWgs > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(msda_bwd_tile_plan, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, a, d, blocks_grid);
    hipLaunchKernelGGL((msda_bwd_tile_reduce<T, D>), dim3((unsigned)(items + kQueueWgs)), dim3(64), 0, st,
                       (const T *)go, (T *)gv, a, d, blocks_grid);
    hipLaunchKernelGGL((msda_bwd_tile_finalize<T, D>), dim3((unsigned)items), dim3(64), 0, st, (T *)gv, a, d, blocks_grid);
    return hipGetLastError();
}

template <typename T>
hipError_t dispatch_tile(const void *go, void *gv, const TileReduceArgs &a, const Dims &d, hipStream_t st)
{
    switch (d.D) {
        case 32: return launch_tile<T, 32>(go, gv, a, d, st);
        case 64: return launch_tile<T, 64>(go, gv, a, d, st);
        case 128: return launch_tile<T, 128>(go, gv, a, d, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace

bool tile_reduce_supported(int dtype, const Dims &d)
{
    if (dtype != 1 && dtype != 2) return false;
    if (d.D != 32 && d.D != 64 && d.D != 128) return false;
    if (d.L > kMaxLevels) return false;
    if (const char *e = getenv("MMFS_VALUE_ALGO")) if (e[0] == 'b' || e[0] == 'p') return false;     // "block", "pixel"
    // queue entries carry (b, h) and the block in 32 bits each; grid = B*H*blocks (+ queue) workgroups
    if ((int64_t)d.B * d.H * ((int64_t)d.S / 4 + d.L + 1) + kQueueWgs > 0x7fffffffLL) return false;
    return true;
}

hipError_t tile_reduce(int dtype, const void *grad_out, void *grad_value, const TileReduceArgs &a, const Dims &d,
                       hipStream_t st)
{
    switch (dtype) {
        case 1: return dispatch_tile<half_t>(grad_out, grad_value, a, d, st);
        case 2: return dispatch_tile<bf16_t>(grad_out, grad_value, a, d, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace blk
}  // namespace mmfs
