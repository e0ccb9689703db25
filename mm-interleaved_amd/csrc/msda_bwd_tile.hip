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
//     registers, no vector instructions), their records too.  A record is 8 bytes: the sample's INPUT
//     words {query | weight bits << 16, x bits | y bits << 16} as the sort found them (16-bit storage,
//     Nq <= 65536); the pixel coordinates are recomputed here with the sort's own expression;
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
//   * A list longer than tile_chunk(D) records (small levels, hot spots -- every text token of the LLM
//     path samples around the same reference point) is cut into work items planned on the device; a
//     block of several items leaves fp32 partial tiles; the item that finishes last adds them up and rounds.
//     The planning itself rides on the sort kernel: a sort tile plans the blocks it holds whole, the slice's last
//     workgroup those on the seams between tiles (plan_tile_begin / plan_slice_blocks, msda_bwd_block.h).
//
// Semantics that differ from the gather kernels, by construction of a matrix product: a non-finite
// grad_out element reaches all 16 pixels of the blocks its sample touches (0 * Inf = NaN), not only
// the sample's four corners.  Finite inputs: same products, fp32 sums in a different order.
#include "msda_bwd_block.h"
#include "msda_env.h"
#include "msda_launch.h"
#include <cstdlib>
#include <type_traits>

namespace mmfs {
namespace blk {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int kKS = 16;                 // records per MFMA step (K of 32x32x16)
constexpr int kStages = 2;              // row slots: one being multiplied, one in flight (more waves per CU beat a deeper pipe)
constexpr int kRecBatch = 4 * kKS;      // records fetched at a time (one DMA, 64 lanes): four steps' worth
constexpr int kRecSlots = 2;            // record batches in LDS

template <typename T> struct TileMma;
template <> struct TileMma<bf16_t> {
    static __device__ __forceinline__ f32x16 run(const s16x8 &a, const s16x8 &b, const f32x16 &c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    // fp32 weight -> leading 16 bits, rounded remainder.  (A non-finite weight -- a non-finite attention
    // weight -- ends as NaN in the sums either way: Inf - Inf in the remainder.)
    static __device__ __forceinline__ void split(float w, uint16_t &hi, uint16_t &lo) {
        const uint32_t h = __float_as_uint(w) & 0xffff0000u;
        hi = (uint16_t)(h >> 16);
        lo = __builtin_bit_cast(uint16_t, (__bf16)(w - __uint_as_float(h)));           // the difference is exact
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
    static constexpr int ROWS_BYTES = kStages * SLOT > 16 * (RB + 16) ? kStages * SLOT : 16 * (RB + 16);   // (the epilogue stages the block's rows here, pitch RB + 16)
    static constexpr int REC_BYTES = kRecSlots * kRecBatch * 8;       // per batch: 64 first words, then 64 second words
#ifndef MMFS_TILE_EXTRA_LDS
#define MMFS_TILE_EXTRA_LDS 0          // (experiment: fewer waves per CU, profiles/r06_experiments.md)
#endif
    static constexpr int LDS_BYTES = ROWS_BYTES + REC_BYTES + 1024 + MMFS_TILE_EXTRA_LDS;
    // chunk swizzle of row r (row index inside its slot): the 4 rows a 16-lane group of a transposing
    // read touches together must land in different banks
    static __device__ __forceinline__ int swz(int r) { return 4 * ((r / RP) % (4 / RP)); }
};

#define MMFS_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n) : "memory")

// Development aid (tools/exp_build.sh tprof "-DMMFS_PROFILE_TILE"): shader clocks per phase of a work
// item, summed over items, read back with mmfs_debug_tile_profile().
#ifdef MMFS_PROFILE_TILE
}  // namespace
constexpr int kProfSlots = 32768;
__device__ unsigned long long g_tile_prof[kProfSlots * 12];     // one row per workgroup (mod kProfSlots): no contended atomics
namespace {
// (the deltas are kept in registers and flushed once, at the end of the item: an atomic per phase would sit
// in the very vmcnt queue the phases wait on)
#define TPROF_DECL unsigned long long tprof_c = __builtin_readcyclecounter(), tprof_t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define TPROF(i) do { const unsigned long long tn = __builtin_readcyclecounter(); tprof_t[i] += tn - tprof_c; tprof_c = tn; } while (0)
#define TPROF_COUNT(i, v) do { tprof_t[i] += (unsigned long long)(v); } while (0)
#define TPROF_FLUSH() do { if (threadIdx.x == 0) for (int i_ = 0; i_ < 12; ++i_) atomicAdd(&g_tile_prof[(blockIdx.x % kProfSlots) * 12 + i_], tprof_t[i_]); } while (0)
#else
#define TPROF_FLUSH() do {} while (0)
#define TPROF_DECL do {} while (0)
#define TPROF(i) do {} while (0)
#define TPROF_COUNT(i, v) do {} while (0)
#endif

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

// (experiments of round 3, profiles/r03_experiments.md r03i: -DMMFS_TILE_NT_RECS streams the records past the L2's
// replacement order, -DMMFS_TILE_NT_STORE the finished grad_value rows: do the one-touch streams push the (b, h) slice
// of grad_out -- read 26 times over -- out of the XCD's L2?)
#ifdef MMFS_TILE_NT_RECS
constexpr int kRecAux = 2;
#else
constexpr int kRecAux = 0;
#endif
__device__ __forceinline__ void dma4(const void *src, void *lds_dst)
{
    __builtin_amdgcn_global_load_lds((glb_void_t *)src, (lds_void_t *)lds_dst, 4, 0, kRecAux);
}
// 16 bytes per lane from rsrc + voffset to lds_dst + lane * 16; an offset past the descriptor's
// extent moves nothing (or zeros): what a record past the end of a list asks for
__device__ __forceinline__ void dma16_buf(__amdgpu_buffer_rsrc_t rsrc, uint32_t voffset, void *lds_dst)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t *)lds_dst, 16, (int)voffset, 0, 0, 0);
}

__device__ __forceinline__ int mfma_row32(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

struct ItemArgs { int b, h, part; bool whole, partial_out; uint32_t pidx; };

// One work item = up to tile_chunk(D) records of one block's list (all of it when `whole`).
//
// Software pipeline, four steps per round so that every LDS address is an immediate (two row slots,
// a batch of 64 records per round):
//   step k (slot k % 2):   wait for rows(k) -> request rows(k + 1) -> multiply step k
//   first step of round j: also request record batch j + 1 (its slot was last read in round j - 1)
// Every request is issued whether or not the list reaches that far (a request past the end moves no
// data), so the number of requests in flight at each wait is a compile-time constant:
//   ... | rows(4j+1) recs(j+1) | rows(4j+2) | rows(4j+3) | rows(4j+4) | rows(4j+5) recs(j+2) | ...
//   step 4j+1 needs rows(4j+1): issued after it: recs(j+1), two requests -> vmcnt(2);  step 4j+3 needs
//   rows(4j+3) and recs(j+1) (for rows(4j+4)) -> vmcnt(0);  the others: vmcnt(0).
template <typename T, int D>
__device__ __forceinline__ void tile_item(const T *__restrict__ grad_out, T *__restrict__ grad_value,
                                          const TileReduceArgs &a, const Dims &d, const TileDesc &td, const ItemArgs &it,
                                          uint32_t *arrived_ctr, unsigned char *__restrict__ lds)
{
    typedef TileGeom<D> G;
    typedef TileMma<T> M;
    unsigned char *rows = lds, *recs = lds + G::ROWS_BYTES, *atile = recs + G::REC_BYTES;
    const int lane = threadIdx.x;
    TPROF_DECL;
    const int Hl = (int)(td.hw >> 16), Wl = (int)(td.hw & 0xffffu);
    const int by = (int)(td.byx >> 16), bx = (int)(td.byx & 0xffffu);
    int pre[6];
    pre[0] = 0;
#pragma unroll
    for (int r = 0; r < 5; ++r) pre[r + 1] = pre[r] + td.cnt[r];
    const int n = pre[5];
    const int e0 = it.part * tile_chunk(d);
    const int e1 = it.whole ? n : min(n, e0 + tile_chunk(d));
    const int cnt = e1 > e0 ? e1 - e0 : 0;                        // records of this item
    const int nks = (cnt + kKS - 1) / kKS;
    const int rounds = (nks + 3) / 4;
    if (__builtin_amdgcn_readfirstlane(n) >= 0) TPROF(0);          // descriptor arrived
    TPROF_COUNT(8, 1); TPROF_COUNT(9, rounds); TPROF_COUNT(10, nks);

    const uint32_t HDB = (uint32_t)(d.H * d.D) * (uint32_t)sizeof(T);            // bytes between consecutive queries
    const T *gslice = grad_out + ((int64_t)it.b * d.Nq * d.H + it.h) * d.D;
    const __amdgpu_buffer_rsrc_t rsrc =
        make_slab_rsrc(gslice, ((int64_t)d.Nq * d.H * d.D - (int64_t)it.h * d.D) * (int64_t)sizeof(T));

    f32x16 acc[G::NB];
#pragma unroll
    for (int nb = 0; nb < G::NB; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;

    if (rounds > 0) {
        // rows past the end of the list are never written: they must not hold stale non-finite bits
        for (int i = lane; i < kStages * G::SLOT / 16; i += 64) reinterpret_cast<uint4 *>(rows)[i] = make_uint4(0u, 0u, 0u, 0u);

        // lane constants
        const int rsel = lane / G::LPR;                                     // row of a DMA instruction this lane serves
        const int g4 = lane >> 4, j16 = lane & 15;
        int troff[G::NB];                                                   // transposing-read offsets inside a slot
#pragma unroll
        for (int nb = 0; nb < G::NB; ++nb) {
            const int krow = 8 * (g4 >> 1) + (j16 >> 2);                    // (+ 4 for the second read: same swizzle)
            const int cb = nb * 64 + (g4 & 1) * 32 + (j16 & 3) * 8;
            troff[nb] = krow * G::RB + (((cb >> 4) ^ G::swz(krow)) << 4) + (cb & 15);
        }
        const int a_rd = (lane & 31) * 32 + (lane >> 5) * 16;               // this lane's 16 bytes of the A operand
        const int wr = lane >> 2, wc = lane & 3;                            // weight tile: record, corner
        const int iy0 = (wc >> 1) - kTB * by, ix0 = (wc & 1) - kTB * bx;

        // record batch j -> slot j % 2 (48 lanes, one record each)
        auto issue_records = [&](int j) {
            if (lane < kRecBatch) {
                const int e = e0 + kRecBatch * j + lane;
                uint32_t idx = 0u;                                           // (any valid address: never looked at)
                if (e < e1) {
                    int del = td.first[0] - pre[0];
#pragma unroll
                    for (int k = 1; k < 5; ++k)
                        if (e >= pre[k]) del = td.first[k] - pre[k];
                    idx = (uint32_t)(e + del);
                }
                const uint32_t *rw = reinterpret_cast<const uint32_t *>(a.records) + 2 * (size_t)idx;
                unsigned char *slot = recs + (j & 1) * (kRecBatch * 8);
                dma4(rw, slot);                                              // {query | weight << 16} of the batch's 64 records
                dma4(rw + 1, slot + kRecBatch * 4);                          // {x | y << 16}
            }
        };
        // grad_out rows of step k = 4j + S -> row slot S % 2
        auto issue_rows = [&](int j, auto stage) {
            constexpr int S = decltype(stage)::value;
            const uint32_t *rq = reinterpret_cast<const uint32_t *>(recs + (j & 1) * (kRecBatch * 8) + S * (kKS * 4));
            const int rel0 = kKS * (4 * j + S);
            uint32_t q[G::NR];
#pragma unroll
            for (int u = 0; u < G::NR; ++u) q[u] = (rq[u * G::RPI + rsel] & 0xffffu) >> a.qshift;     // (taps_sorted: the record carries query * P + point)
#pragma unroll
            for (int u = 0; u < G::NR; ++u) {
                const int rr = u * G::RPI + rsel;
                const uint32_t chunk = (uint32_t)((lane % G::LPR) ^ G::swz(rr));
                const uint32_t off = rel0 + rr < cnt ? __umul24(q[u], HDB) + chunk * 16u : kOobOffset;     // (Nq, H*D*e < 2^24: checked on the host)
                dma16_buf(rsrc, off, rows + (S % 2) * G::SLOT + u * 1024);
            }
        };
        auto multiply = [&](int j, auto stage) {
            constexpr int S = decltype(stage)::value;
            // ---- weight tile: 16 records x 4 corners, one weight per lane
            reinterpret_cast<uint4 *>(atile)[lane] = make_uint4(0u, 0u, 0u, 0u);
            {
                const uint32_t *rb = reinterpret_cast<const uint32_t *>(recs + (j & 1) * (kRecBatch * 8)) + S * kKS + wr;
                const uint32_t w0 = rb[0], w1 = rb[kRecBatch];
                const float av = to_f32(__builtin_bit_cast(T, (uint16_t)(w0 >> 16)));
                const float lx = to_f32(__builtin_bit_cast(T, (uint16_t)(w1 & 0xffffu))), ly = to_f32(__builtin_bit_cast(T, (uint16_t)(w1 >> 16)));
                const float y = ly * (float)Hl - 0.5f, x = lx * (float)Wl - 0.5f;          // (the sort's expression, bit for bit)
                const float yf = floorf(y), xf = floorf(x);
                const float fy = y - yf, fx = x - xf;
                const int iy = (int)yf + iy0, ix = (int)xf + ix0;
                const float wy = (wc >> 1) ? fy : 1.f - fy, wx = (wc & 1) ? fx : 1.f - fx;
                const float wgt = wy * wx * av;
                if (kKS * (4 * j + S) + wr < cnt && (unsigned)iy < (unsigned)kTB && (unsigned)ix < (unsigned)kTB) {
                    uint16_t hi, lo;
                    M::split(wgt, hi, lo);
                    const int m = iy * kTB + ix;
                    reinterpret_cast<uint16_t *>(atile)[m * kKS + wr] = hi;
                    reinterpret_cast<uint16_t *>(atile)[(16 + m) * kKS + wr] = lo;
                }
            }
            const s16x8 A = *reinterpret_cast<const s16x8 *>(atile + a_rd);
            // ---- the step's rows as B operands, one MFMA per 32 channels
            const unsigned char *slot = rows + (S % 2) * G::SLOT;
#pragma unroll
            for (int nb = 0; nb < G::NB; ++nb) {
                s16x8 B;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4 *)(slot + troff[nb] + t * 4 * G::RB));
                    B[4 * t] = v[0]; B[4 * t + 1] = v[1]; B[4 * t + 2] = v[2]; B[4 * t + 3] = v[3];
                }
                acc[nb] = M::run(A, B, acc[nb]);
            }
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        using S2 = std::integral_constant<int, 2>;
        using S3 = std::integral_constant<int, 3>;

        // prologue: recs(0) | rows(0)
        issue_records(0);
        MMFS_WAIT_VM(0);
        TPROF(1);                                                 // LDS cleared, first records landed
        issue_rows(0, S0{});
        for (int j = 0; j < rounds; ++j) {
            MMFS_WAIT_VM(0);
            if (j == 0) TPROF(2);                                 // first rows landed
            issue_rows(j, S1{});                                  // rows(4j + 1)
            issue_records(j + 1);
            if (4 * j < nks) multiply(j, S0{});
            MMFS_WAIT_VM(2);
            issue_rows(j, S2{});
            if (4 * j + 1 < nks) multiply(j, S1{});
            MMFS_WAIT_VM(0);
            issue_rows(j, S3{});
            if (4 * j + 2 < nks) multiply(j, S2{});
            MMFS_WAIT_VM(0);
            issue_rows(j + 1, S0{});                              // rows(4j + 4): first step of batch j + 1
            if (4 * j + 3 < nks) multiply(j, S3{});
        }
        TPROF(3);                                                 // rounds
        MMFS_WAIT_VM(0);                                          // (requests past the end of the list)
        TPROF(4);
    }

    // ---- epilogue: hi + lo rows; lane pairs exchange one value so that every lane holds whole dwords
    const bool odd = lane & 1;
    const int ch0 = (lane & 31) & ~1;
    constexpr int PITCH = G::RB + 16;                             // staging pitch: rows 8 apart in different banks
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int pa = mfma_row32(r, lane), pb = mfma_row32(r + 4, lane);      // pixels 0..15 of the block
        const int p = odd ? pb : pa;
#pragma unroll
        for (int nb = 0; nb < G::NB; ++nb) {
            const float va = acc[nb][r] + acc[nb][r + 8], vb = acc[nb][r + 4] + acc[nb][r + 12];
            const float recv = __shfl_xor(odd ? va : vb, 1, 64);
            const float lo_ch = odd ? recv : va, hi_ch = odd ? vb : recv;       // channels ch0, ch0 + 1 of pixel p
            const int ch = nb * 32 + ch0;
            if (it.partial_out) {
                // (written through, agent scope: another item of the block adds the tiles up within this launch)
                float *o = a.tpartials + ((int64_t)it.pidx * (kTB * kTB) + p) * D + ch;
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(o),
                                   ((unsigned long long)__float_as_uint(hi_ch) << 32) | (unsigned long long)__float_as_uint(lo_ch),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                *reinterpret_cast<uint32_t *>(rows + p * PITCH + ch * 2) = M::pack2(lo_ch, hi_ch);
            }
        }
    }
    if (it.partial_out) {
        // A block of several items: the item that arrives last adds the partial tiles up (in tile order,
        // whoever is last) and rounds.  The tiles are stored and read with agent-scope accesses and the
        // stores are drained before the counter moves: no L2 write-back (a release fence here writes
        // back every dirty grad_value line of the XCD, measured: the kernel twice as slow).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t prev = 0u;
        if (lane == 0) prev = __hip_atomic_fetch_add(arrived_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        prev = (uint32_t)__builtin_amdgcn_readfirstlane((int)prev);
        if (prev == td.parts - 1u) {
            constexpr int CPL = D >= 128 ? 2 : 1;                  // channels per lane
            const int ch = lane * CPL;
            float s[kTB * kTB][CPL];
#pragma unroll
            for (int p = 0; p < kTB * kTB; ++p)
#pragma unroll
                for (int k = 0; k < CPL; ++k) s[p][k] = 0.f;
            if (ch < D) {
                for (uint32_t c = 0; c < td.parts; ++c) {
                    const float *src = a.tpartials + (int64_t)(td.pbase + c) * (kTB * kTB) * D + ch;
#pragma unroll
                    for (int p = 0; p < kTB * kTB; ++p) {
                        if (CPL == 2) {
                            const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(src + p * D),
                                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            s[p][0] += __uint_as_float((uint32_t)v); s[p][CPL - 1] += __uint_as_float((uint32_t)(v >> 32));
                        } else {
                            s[p][0] += __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t *>(src + p * D),
                                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                        }
                    }
                }
#pragma unroll
                for (int p = 0; p < kTB * kTB; ++p) {
                    const int y = kTB * by + (p >> 2), x = kTB * bx + (p & 3);
                    if (y < Hl && x < Wl) {
                        T *o = grad_value + (((int64_t)it.b * d.S + td.lstart + y * Wl + x) * d.H + it.h) * d.D + ch;
                        if (CPL == 2) *reinterpret_cast<uint32_t *>(o) = M::pack2(s[p][0], s[p][CPL - 1]);
                        else o[0] = (T)s[p][0];
                    }
                }
            }
        }
    } else {
        // the block's 16 rows leave as whole 16-byte chunks, LPR consecutive lanes per row
#pragma unroll
        for (int i = 0; i < kTB * kTB * G::LPR / 64; ++i) {
            const int c = i * 64 + lane, p = c / G::LPR, chunk = c % G::LPR;
            const uint4 v = *reinterpret_cast<const uint4 *>(rows + p * PITCH + chunk * 16);
            const int y = kTB * by + (p >> 2), x = kTB * bx + (p & 3);
            if (y < Hl && x < Wl) {
                T *o = grad_value + (((int64_t)it.b * d.S + td.lstart + y * Wl + x) * d.H + it.h) * d.D + chunk * 8;
#ifndef MMFS_TILE_PLAIN_STORE
                store16_stream(o, v);              // (grad_value is not read again in the step: r03i, 0.4604 -> 0.4431 ms)
#else
                *reinterpret_cast<uint4 *>(o) = v;
#endif
            }
        }
    }
#ifdef MMFS_PROFILE_TILE
    {   // epilogue clocks and counts by kind of item: whole blocks / parts that leave a partial tile
        const unsigned long long tn = __builtin_readcyclecounter();
        if (it.partial_out) { tprof_t[6] += tn - tprof_c; tprof_t[7] += 1; } else { tprof_t[11] += tn - tprof_c; }
    }
#endif
    TPROF(5);                                                     // epilogue issued
    TPROF_FLUSH();
}

// One wave per work item.  A (b, h) slice's workgroups are consecutive in the dispatch order of their XCD (h is the
// fastest index): first the slice's queue of extra items -- the later parts of long lists, the long poles -- then
// its blocks, coarse levels first.  The XCD's L2 then holds one or two slices' grad_out rows at a time (a queue
// shared by the whole batch, walked first, had every XCD read rows of all B slices of its head at once: 544 MB of
// traffic for 156 MB of algorithmic bytes at the north star, r03q).
template <typename T, int D>
__global__ void __launch_bounds__(64, 4)
msda_bwd_tile_reduce(const T *__restrict__ grad_out, T *__restrict__ grad_value, const TileReduceArgs a,
                     const Dims d, const int blocks_grid, const int extra_grid)
{
    typedef TileGeom<D> G;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[G::LDS_BYTES];
    const int w = blockIdx.x;
    ItemArgs it;
    it.h = w % d.H;
    const int t = w / d.H;
    const int per_slice = extra_grid + blocks_grid;
    int j = t % per_slice;
    it.b = t / per_slice;
    if (a.hdr->stamp != header_stamp(d)) return;                 // (a plan made for other dimensions: nothing to do)
    const int64_t bh = (int64_t)it.b * d.H + it.h;
    if (j < extra_grid) {
        const uint32_t n = min(a.n_extra[bh], a.th->cap_extra);
        if ((uint32_t)j >= n) return;
        const TileItem ti = a.titems[(size_t)bh * a.th->cap_extra + j];
        if (ti.part == kVoidPart) return;                  // a reservation its block could not use
        it.part = (int)ti.part; it.whole = false; it.partial_out = true; it.pidx = ti.pidx;
        TileDesc *tdp = a.tdesc + (bh * a.blocks_bound + ti.blk);
        const TileDesc td = *tdp;
        tile_item<T, D>(grad_out, grad_value, a, d, td, it, &tdp->arrived, lds);
        return;
    }
    j -= extra_grid;
    const int nblk = a.hdr->n_blocks4;
    if (j >= nblk) return;
    const int blk = nblk - 1 - j;                          // coarse levels (long lists) first
    TileDesc *tdp = a.tdesc + (bh * a.blocks_bound + blk);
    const TileDesc td = *tdp;
    it.part = 0; it.whole = td.parts <= 1; it.partial_out = td.parts > 1; it.pidx = td.pbase;
    tile_item<T, D>(grad_out, grad_value, a, d, td, it, &tdp->arrived, lds);
}

template <typename T, int D>
hipError_t launch_tile(const void *go, void *gv, const TileReduceArgs &a, const Dims &d, uint32_t cap_extra, hipStream_t st)
{
    // the exact block count is only known on the device; the grid takes the caller's hint (host copy
    // of the level table) or the bound, surplus workgroups return at once.  Queue places: the slice's capacity
    // (how many are taken is only known on the device)
    const int blocks_grid = d.blocks4 > 0 ? std::min(d.blocks4, a.blocks_bound) : a.blocks_bound;
    const int64_t items = (int64_t)d.B * d.H * ((int64_t)blocks_grid + cap_extra);
    if (items > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((msda_bwd_tile_reduce<T, D>), dim3((unsigned)items), dim3(64), 0, st,
                       (const T *)go, (T *)gv, a, d, blocks_grid, (int)cap_extra);
    return hipGetLastError();
}
template <typename T>
hipError_t dispatch_tile(const void *go, void *gv, const TileReduceArgs &a, const Dims &d, uint32_t cap_extra, hipStream_t st)
{
    switch (d.D) {
        case 32: return launch_tile<T, 32>(go, gv, a, d, cap_extra, st);
        case 64: return launch_tile<T, 64>(go, gv, a, d, cap_extra, st);
        case 128: return launch_tile<T, 128>(go, gv, a, d, cap_extra, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace

#ifdef MMFS_PROFILE_TILE
extern "C" int mmfs_debug_tile_profile(unsigned long long *out, int reset)
{
    static unsigned long long host[mmfs::blk::kProfSlots * 12];
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(mmfs::blk::g_tile_prof), sizeof(host));
    for (int i = 0; i < 16; ++i) out[i] = 0;
    for (int s = 0; s < mmfs::blk::kProfSlots; ++s)
        for (int i = 0; i < 12; ++i) out[i] += host[s * 12 + i];
    if (e == hipSuccess && reset) {
        for (auto &v : host) v = 0;
        e = hipMemcpyToSymbol(HIP_SYMBOL(mmfs::blk::g_tile_prof), host, sizeof(host));
    }
    return (int)e;
}
#endif

bool tile_reduce_supported(int dtype, const Dims &d)
{
    if (dtype != 1 && dtype != 2) return false;
    if (d.D != 32 && d.D != 64 && d.D != 128) return false;
    if (d.L > kMaxLevels) return false;
    if (const char *e = knob_str(K_VALUE_ALGO)) if (e[0] == 'b' || e[0] == 'p') return false;       // "block", "pixel"
    // queue entries carry (b, h) and the block in 32 bits each; grid = B*H*blocks (+ queue) workgroups
    if ((int64_t)d.B * d.H * ((int64_t)d.S / 4 + d.L + 1 + 2 * ((int64_t)d.Nq * d.L * d.P * 25 / 16 / tile_chunk(d)) + 16) > 0x7fffffffLL) return false;
    // the rows are fetched through a buffer descriptor over one (b, h) slice: 31-bit byte offsets
    const int64_t es = 2;
    if ((int64_t)d.Nq * d.H * d.D * es > kMaxSlabBytes) return false;
    if (d.Nq > 65536 || (int64_t)d.H * d.D * es >= (1 << 24)) return false;            // 16-bit query index in the records; 24-bit multiply for the row offset
    return true;
}

hipError_t tile_reduce(int dtype, const void *grad_out, void *grad_value, const TileReduceArgs &a, const Dims &d,
                       uint32_t cap_extra, hipStream_t st)
{
    switch (dtype) {
        case 1: return dispatch_tile<half_t>(grad_out, grad_value, a, d, cap_extra, st);
        case 2: return dispatch_tile<bf16_t>(grad_out, grad_value, a, d, cap_extra, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace blk
}  // namespace mmfs
