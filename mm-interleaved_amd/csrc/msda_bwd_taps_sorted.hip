// msda_bwd_taps_sorted.hip -- grad_loc / grad_attn of multi-scale deformable attention from the CELL-SORTED records.
//
// Reference: ms_deform_attn_col2im_bilinear, ms_deform_im2col_cuda.cuh:90-162 (per sample: the four corner values dotted
// with grad_out, then  grad_attn = sum_c w_c d_c,  grad_loc = (W a (hh (d2 - d1) + lh (d4 - d3)),  H a (hw (d3 - d1) + lw (d4 - d2)))).
//
// The gather kernels (msda_bwd.hip, msda_taps_mma.hip) own QUERIES and fetch, per sample, four value rows through the
// vector-memory path: 4.3 GB at the north star, the 64 B/clk/CU of that path being what they run against.  The grad_value
// half already sorts the samples by the cell of their top-left corner (msda_bwd_block.hip) -- and a sample's four corners
// are the four pixels around its cell.  So here the unit of work is a 4x4 block of CELLS:
//
//   * a wave owns the cells (cy, cx) with cy / 4 == by, cx / 4 == bx of one level of one (b, h) slice (the last block row /
//     column also the border cells cy = H / cx = W): every sample has exactly one owner.  The corners of all its samples lie
//     in the 5x5 pixels (4 by - 1 .. 4 by + 3) x (4 bx - 1 .. 4 bx + 3): 25 value rows, loaded ONCE per work item, straight
//     into the registers of a matrix-core operand (A of v_mfma_f32_16x16x32: row = pixel, 16 bytes of "its" row per lane);
//   * the block's owned records are five prefixes of the runs the grad_value plan describes (TapsDesc, msda_bwd_block.h); 16
//     records per step, their grad_out rows global -> LDS by DMA (as in msda_bwd_tile.hip; one row visit per sample, not
//     25/16), read back as the B operand (column = record): one product chain gives D[pixel, record] = value[pixel, :] .
//     grad_out[query(record), :] for all 25 pixels -- a sample needs four of them;
//   * which four: the accumulator layout puts pixel row j of the 5x5 into the lanes 16 j .. 16 j + 15 (j < 4; the fifth row
//     and column ride in the second product's rows), so lane (record n, j) holds the five dots of pixel row j with ITS
//     record's gradient.  It selects the two columns its sample straddles (conditional moves by the sample's column, no
//     multiplication by zero: a non-finite value row reaches the samples that touch it and no other), weights them along x,
//     and contributes as the sample's top or bottom row -- or not at all; three partial sums per lane meet across the four
//     lane rows (v_permlane16_swap / v_permlane32_swap);
//   * results leave by sample index (query * P + point rides in the record: Dims::taps_sorted), 2 + 4 bytes per sample.
//
// No workgroup barrier, no cross-item state: a block's records may be cut anywhere, so a long list is shared by the work
// items the grad_value plan made for it (same grid, same item -> XCD order: a slice's grad_out rows in one L2).
// Samples without a record (outside the level, NaN, zero weight under lazy_attn) were zeroed by the sort.
#include "msda_bwd_block.h"
#include "msda_env.h"
#include "msda_launch.h"
#include <type_traits>

namespace mmfs {
namespace blk {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int kKS = 16;                 // records per step (N of the product)
constexpr int kRecBatch = 4 * kKS;      // records fetched at a time (one DMA per record word, 64 lanes): four steps' worth

template <typename T> struct DotMma;
template <> struct DotMma<bf16_t> {
    static __device__ __forceinline__ f32x4 run(const s16x8 &a, const s16x8 &b, const f32x4 &c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t pack2(float a, float b) { return Vec16<bf16_t>::pk(a, b); }
};
template <> struct DotMma<half_t> {
    static __device__ __forceinline__ f32x4 run(const s16x8 &a, const s16x8 &b, const f32x4 &c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t pack2(float a, float b) { return Vec16<half_t>::pk(a, b); }
};

template <int D> struct TsGeom {
    static constexpr int RB = D * 2;               // bytes of a row (one head, 16-bit storage)
    static constexpr int LPR = RB / 16;            // 16-byte chunks (= DMA lanes) per row
    static constexpr int RPI = 64 / LPR;           // rows per DMA instruction
    static constexpr int NR = kKS / RPI;           // DMA instructions per step
    static constexpr int KT = D / 32;              // products of a chain (K = 32 channels each)
    static constexpr int SLOT = kKS * RB;          // bytes of a row slot
    static constexpr int ROWS_BYTES = 2 * SLOT;    // one slot being multiplied, one in flight
    static constexpr int REC_BYTES = 2 * kRecBatch * 8;       // per batch: 64 first words, then 64 second words
    // the step's dots D[record][pixel] (fp32) on their way from the accumulator layout (a lane = a record x a pixel ROW) to
    // the lane that needs them (a lane = a record x a CORNER): 16 records at a pitch of 36 words (the two 16-byte stores of
    // the 8 lanes of a store phase fall into 8 x 4 different banks)
    static constexpr int DOT_PITCH = 36 * 4;
    static constexpr int DOT_BYTES = kKS * DOT_PITCH;
    static constexpr int LDS_BYTES = ROWS_BYTES + REC_BYTES + DOT_BYTES;
    // Channel chunk of lane group g in product t of the chain: 4 t + g (both operands).  Row r of a slot stores its logical
    // chunk c at position c ^ swz(r) (applied to the DMA's SOURCE address: the LDS image of a DMA is lane-linear), which
    // makes the ds_read_b128 of the B operand -- 16 lanes = 8 + 8 rows at two neighbouring chunks per LDS cycle -- free
    // of bank conflicts for every head width (checked by brute force over the instruction's lane groups:
    // tests/test_taps_sorted_model.py).
    static __device__ __forceinline__ int swz(int r) { return D == 128 ? r : D == 64 ? (r >> 1) & 7 : ((r >> 3) & 1) * 3; }
};

#define MMFS_TS_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n) : "memory")

// Development aid (tools/exp_build1.sh tsprof msda_bwd_taps_sorted "-DMMFS_PROFILE_TS"; tools/ts_prof.py): shader clocks per
// phase of a work item, summed over items, read back with mmfs_debug_ts_profile().
#ifdef MMFS_PROFILE_TS
}  // namespace
constexpr int kTsProfSlots = 32768;
__device__ unsigned long long g_ts_prof[kTsProfSlots * 12];      // one row per workgroup (mod kTsProfSlots): no contended atomics
namespace {
#define TSPROF_DECL unsigned long long tsp_c = __builtin_readcyclecounter(), tsp_t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define TSPROF(i) do { const unsigned long long tn = __builtin_readcyclecounter(); tsp_t[i] += tn - tsp_c; tsp_c = tn; } while (0)
#define TSPROF_COUNT(i, v) do { tsp_t[i] += (unsigned long long)(v); } while (0)
#define TSPROF_FLUSH() do { if (threadIdx.x == 0) for (int i_ = 0; i_ < 12; ++i_) atomicAdd(&g_ts_prof[(blockIdx.x % kTsProfSlots) * 12 + i_], tsp_t[i_]); } while (0)
#else
#define TSPROF_DECL do {} while (0)
#define TSPROF(i) do {} while (0)
#define TSPROF_COUNT(i, v) do {} while (0)
#define TSPROF_FLUSH() do {} while (0)
#endif

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

__device__ __forceinline__ void ts_dma4(const void *src, void *lds_dst)
{
    __builtin_amdgcn_global_load_lds((glb_void_t *)src, (lds_void_t *)lds_dst, 4, 0, 0);
}
__device__ __forceinline__ void ts_dma16_buf(__amdgpu_buffer_rsrc_t rsrc, uint32_t voffset, void *lds_dst)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t *)lds_dst, 16, (int)voffset, 0, 0, 0);
}

// x + (the same register of the lanes 16 / 32 away): after both, every lane holds the sum over its four lane rows
__device__ __forceinline__ float add_rows16(float x)
{
    const uint32_t u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float add_rows32(float x)
{
    const uint32_t u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// c ? a : b on values that are already there: a v_cndmask, never a branch (a conditional expression whose arms are
// expressions becomes control flow under exec masks -- the first build of this kernel spent its time there)
__device__ __forceinline__ float sel(bool c, float a, float b) { return c ? a : b; }
__device__ __forceinline__ int sel_i(bool c, int a, int b) { return c ? a : b; }

struct TsItem { int b, h, part; };

// One work item: part `part` of the td.parts equal parts of a block's owned records.
//
// Software pipeline as in msda_bwd_tile.hip, four steps per round so that every LDS address is an immediate:
//   step k (slot k % 2):   wait for rows(k) -> request rows(k + 1) -> multiply step k
//   first step of round j: also request record batch j + 1
// Every request is issued whether or not the list reaches that far (a request past the end moves no data).
template <typename T, int D>
__device__ __forceinline__ void taps_item(const T *__restrict__ value, const T *__restrict__ grad_out,
                                          T *__restrict__ grad_loc, T *__restrict__ grad_attn,
                                          const TileReduceArgs &a, const Dims &d, const TileDesc &td, const TapsDesc &xd,
                                          const TsItem &it, unsigned char *__restrict__ lds)
{
    typedef TsGeom<D> G;
    typedef DotMma<T> M;
    unsigned char *rows = lds, *recs = lds + G::ROWS_BYTES, *dots = recs + G::REC_BYTES;
    const int lane = threadIdx.x;
    TSPROF_DECL;
    const int Hl = (int)(td.hw >> 16), Wl = (int)(td.hw & 0xffffu);
    const int by = (int)(td.byx >> 16), bx = (int)(td.byx & 0xffffu);
    int pre[6];
    pre[0] = 0;
#pragma unroll
    for (int r = 0; r < 5; ++r) pre[r + 1] = pre[r] + xd.ocnt[r];
    if (__builtin_amdgcn_readfirstlane(pre[5]) >= 0) TSPROF(0);     // descriptors arrived
    const int n = pre[5];
    const int parts = (int)td.parts > 0 ? (int)td.parts : 1;
    const int e0 = (int)((int64_t)n * it.part / parts), e1 = (int)((int64_t)n * (it.part + 1) / parts);
    const int cnt = e1 - e0;                                       // records of this item
    if (cnt <= 0) return;
    const int nks = (cnt + kKS - 1) / kKS;
    const int rounds = (nks + 3) / 4;
    TSPROF_COUNT(8, 1); TSPROF_COUNT(9, rounds); TSPROF_COUNT(10, nks); TSPROF_COUNT(11, cnt);

    const int n16 = lane & 15, j = lane >> 4;                      // this lane's record of a step / its pixel row of the 5x5
    const int y00 = kTB * by - 1, x00 = kTB * bx - 1;              // the 5x5's first pixel

    // ---- the block's 25 value rows as A operands: group 0 = rows 0..3 x columns 0..3 (m = 4 py + px), group 1 holds, for
    // m = 4 jj + s: s = 0 pixel (jj, 4), s = 1 pixel (4, jj), s = 2 pixel (4, 4) (jj = 0 only), else nothing.
    // They travel like the grad_out rows: global -> LDS by DMA, whole rows per 16 / 8 / 4 lanes (loaded straight into the
    // operand layout -- a lane = 16 bytes of "its" row -- four consecutive lanes ask for four different rows and the texture
    // path serves 16 bytes per clock instead of 64: 512 clocks per work item, as much as eight steps' row requests), into the
    // row slots (free until the first step's rows are asked for), operand row r = 16 g + m at row r, chunks swizzled as the
    // slots' rows are; a pixel that is not there (outside the map, an unused operand row) moves nothing and is masked at the read.
    auto pixel_of_row = [&](int r, int &y, int &x) -> bool {
        const int g = r >> 4, jj = (r >> 2) & 3, sx = r & 3;
        const int py = g == 0 ? jj : (sx == 0 ? jj : 4), px = g == 0 ? sx : (sx == 0 ? 4 : sx == 1 ? jj : 4);
        y = y00 + py; x = x00 + px;
        return (g == 0 || sx <= 1 || (sx == 2 && jj == 0)) && y >= 0 && y < Hl && x >= 0 && x < Wl;
    };
    {
        const T *vslice = value + (((int64_t)it.b * d.S + td.lstart) * d.H + it.h) * d.D;
        const __amdgpu_buffer_rsrc_t vrsrc =
            make_slab_rsrc(vslice, ((int64_t)Hl * Wl * d.H * d.D - (int64_t)it.h * d.D) * (int64_t)sizeof(T));
        const uint32_t PXB = (uint32_t)(d.H * d.D) * (uint32_t)sizeof(T);        // bytes between consecutive pixels
#pragma unroll
        for (int u = 0; u < 32 / G::RPI; ++u) {
            const int r = u * G::RPI + lane / G::LPR;
            int y, x;
            const bool ok = pixel_of_row(r, y, x);
            const uint32_t chunk = (uint32_t)((lane % G::LPR) ^ G::swz(r & 15));
            const uint32_t off = ok ? (uint32_t)(y * Wl + x) * PXB + chunk * 16u : kOobOffset;
            ts_dma16_buf(vrsrc, off, rows + u * 1024);
        }
    }
    s16x8 va[2][G::KT];                                                 // (filled in the prologue, once the rows have landed)

    const uint32_t HDB = (uint32_t)(d.H * d.D) * (uint32_t)sizeof(T);            // bytes between consecutive queries
    const T *gslice = grad_out + ((int64_t)it.b * d.Nq * d.H + it.h) * d.D;
    const __amdgpu_buffer_rsrc_t rsrc =
        make_slab_rsrc(gslice, ((int64_t)d.Nq * d.H * d.D - (int64_t)it.h * d.D) * (int64_t)sizeof(T));
    const int rsel = lane / G::LPR;                                     // row of a DMA instruction this lane serves
    int b_rd[G::KT];                                                    // this lane's B fragments inside a slot
#pragma unroll
    for (int t = 0; t < G::KT; ++t) b_rd[t] = n16 * G::RB + (((4 * t + j) ^ G::swz(n16)) << 4);
    const int64_t out_base = (((int64_t)it.b * d.Nq) * d.H + it.h) * d.L * d.P + (int64_t)xd.level * d.P;
    const int64_t q_stride = (int64_t)d.H * d.L * d.P;
    const uint32_t pmask = (1u << a.qshift) - 1u;
    const float fH = (float)Hl, fW = (float)Wl;

    // record batch jb -> slot jb % 2
    auto issue_records = [&](int jb) {
        const int e = e0 + kRecBatch * jb + lane;
        uint32_t idx = 0u;                                               // (any valid address: never looked at)
        if (e < e1) {
            int del = td.first[0] - pre[0];
#pragma unroll
            for (int k = 1; k < 5; ++k)
                if (e >= pre[k]) del = td.first[k] - pre[k];
            idx = (uint32_t)(e + del);
        }
        const uint32_t *rw = reinterpret_cast<const uint32_t *>(a.records) + 2 * (size_t)idx;
        unsigned char *slot = recs + (jb & 1) * (kRecBatch * 8);
        ts_dma4(rw, slot);                                               // {query * P + point | weight << 16}
        ts_dma4(rw + 1, slot + kRecBatch * 4);                           // {x | y << 16}
    };
    // grad_out rows of step k = 4 jb + S -> row slot S % 2
    auto issue_rows = [&](int jb, auto stage) {
        constexpr int S = decltype(stage)::value;
        const uint32_t *rq = reinterpret_cast<const uint32_t *>(recs + (jb & 1) * (kRecBatch * 8) + S * (kKS * 4));
        const int rel0 = kKS * (4 * jb + S);
        uint32_t q[G::NR];
#pragma unroll
        for (int u = 0; u < G::NR; ++u) q[u] = (rq[u * G::RPI + rsel] & 0xffffu) >> a.qshift;
#pragma unroll
        for (int u = 0; u < G::NR; ++u) {
            const int rr = u * G::RPI + rsel;
            const uint32_t chunk = (uint32_t)((lane % G::LPR) ^ G::swz(rr));
            const uint32_t off = rel0 + rr < cnt ? __umul24(q[u], HDB) + chunk * 16u : kOobOffset;
            ts_dma16_buf(rsrc, off, rows + (S % 2) * G::SLOT + u * 1024);
        }
    };
    // what a step leaves for the next one to store: the stores of step k are issued after the row requests of step k + 1, so
    // that the wait for those rows (vmcnt counts stores too) finds them done instead of exposing their round trip
    int64_t pend_s = 0;
    uint32_t pend_gl = 0u;
    float pend_ga = 0.f;
    bool pend_ok = false;
    auto flush = [&]() {
#ifdef TS_EXP_NOSTORE
        pend_ok = pend_ok && pend_ga == 12345.678f;
#endif
        if (pend_ok) {
            if (j == 0) grad_attn[pend_s] = (T)pend_ga;
            if (j == 1) *reinterpret_cast<uint32_t *>(grad_loc + 2 * pend_s) = pend_gl;
        }
        pend_ok = false;
    };
    // The step's dots leave the accumulators through LDS: lane (n, j) holds pixel row j's five dots and the fifth row's
    // pixel (4, j) (+ (4, 4) in lane row 0) of record n -- as 8 consecutive words [8 j .. 8 j + 7] of the record's 36-word
    // line: pixel (py < 4, px < 4) at 8 py + px, (py < 4, 4) at 8 py + 4, (4, px < 4) at 8 px + 5, (4, 4) at 6 -- and reads
    // back ONE: the dot of corner c = j of its record's sample, wherever in the 5x5 that is.  (The first build selected in
    // registers: 190 vector instructions per step, the kernel's bound -- profiles/r06_experiments.md.)
    const int dot_wr = n16 * G::DOT_PITCH + j * 32;
    const int cy = j >> 1, cx = j & 1;                              // this lane's corner of its record's sample
    auto multiply = [&](int jb, auto stage) {
        constexpr int S = decltype(stage)::value;
        const unsigned char *slot = rows + (S % 2) * G::SLOT;
        // ---- this lane's record (the four lanes n16, n16 + 16, ... share it)
        const uint32_t *rb = reinterpret_cast<const uint32_t *>(recs + (jb & 1) * (kRecBatch * 8)) + S * kKS + n16;
        const uint32_t w0 = rb[0], w1 = rb[kRecBatch];
        // ---- D[pixel, record] = value row . grad_out row
        f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < G::KT; ++t) {
            const s16x8 bf = *reinterpret_cast<const s16x8 *>(slot + b_rd[t]);
            d0 = M::run(va[0][t], bf, d0);
            d1 = M::run(va[1][t], bf, d1);
        }
        *reinterpret_cast<f32x4 *>(dots + dot_wr) = d0;
        *reinterpret_cast<f32x4 *>(dots + dot_wr + 16) = d1;
        const float av = to_f32(__builtin_bit_cast(T, (uint16_t)(w0 >> 16)));
        const float lx = to_f32(__builtin_bit_cast(T, (uint16_t)(w1 & 0xffffu))), ly = to_f32(__builtin_bit_cast(T, (uint16_t)(w1 >> 16)));
        const float y = ly * fH - 0.5f, x = lx * fW - 0.5f;              // (the sort's expression, bit for bit)
        const float yf = floorf(y), xf = floorf(x);
        const float fy = y - yf, fx = x - xf;
        const int y0 = (int)yf + cy, x0 = (int)xf + cx;                  // this corner's pixel in the map ...
        const int py = y0 - y00, px = x0 - x00;                          // ... and in the 5x5: 0 .. 5 (5: past the map's last row / column)
        // a corner outside the map reads 0 (cuh:58-81); inside the map it is inside the 5x5 (the block owns the sample)
        const bool in_map = (y0 >= 0) & (y0 < Hl) & (x0 >= 0) & (x0 < Wl);
        const int w_in = py * 8 + px, w_row4 = px * 8 + 5;
        int word = sel_i(py < 4, sel_i(px < 4, w_in, py * 8 + 4), sel_i(px < 4, w_row4, 6));
        word = sel_i(in_map, word, 0);
        const float dot_raw = *reinterpret_cast<const float *>(dots + n16 * G::DOT_PITCH + word * 4);
        const float dot = sel(in_map, dot_raw, 0.f);
        // ---- this corner's share (cuh:119-161): w = wy wx;  d/dx: wy (+-1);  d/dy: (+-1) wx
        const float wy = sel(cy != 0, fy, 1.f - fy), wx = sel(cx != 0, fx, 1.f - fx);
        const float wyd = wy * dot, wxd = wx * dot;
        float ga = wx * wyd;
        float gx = sel(cx != 0, wyd, -wyd);
        float gy = sel(cy != 0, wxd, -wxd);
        ga = add_rows32(add_rows16(ga));
        gx = add_rows32(add_rows16(gx));
        gy = add_rows32(add_rows16(gy));
        // ---- out by sample index (stored by the next step)
        const uint32_t qp = w0 & 0xffffu;
        pend_s = out_base + (int64_t)(qp >> a.qshift) * q_stride + (qp & pmask);
        pend_ga = ga;
        pend_gl = M::pack2(fW * av * gx, fH * av * gy);
        pend_ok = kKS * (4 * jb + S) + n16 < cnt;
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    using S3 = std::integral_constant<int, 3>;

    // prologue: value rows + recs(0) | operands out of LDS | rows(0)
    issue_records(0);
    MMFS_TS_WAIT_VM(0);
    {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            int y, x;
            const uint32_t keep = pixel_of_row(16 * g + n16, y, x) ? 0xffffffffu : 0u;
#pragma unroll
            for (int t = 0; t < G::KT; ++t) {
                uint4 v = *reinterpret_cast<const uint4 *>(rows + g * 16 * G::RB + b_rd[t]);
                v.x &= keep; v.y &= keep; v.z &= keep; v.w &= keep;
                va[g][t] = __builtin_bit_cast(s16x8, v);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (the operands are in registers before the slots are written again)
    }
    issue_rows(0, S0{});
    // (vmcnt counts the stores too, in issue order: a wait that lets the n newest requests stand assumes the stores of the
    // step before are among them -- when a step had none, the wait is merely stricter than needed)
    for (int jb = 0; jb < rounds; ++jb) {
        MMFS_TS_WAIT_VM(0);
        issue_rows(jb, S1{});                                  // rows(4 jb + 1)
        issue_records(jb + 1);
        flush();
        if (4 * jb < nks) multiply(jb, S0{});
        MMFS_TS_WAIT_VM(2);                                    // rows(4 jb + 1) are needed; the records / the stores behind them may stand
        issue_rows(jb, S2{});
        flush();
        if (4 * jb + 1 < nks) multiply(jb, S1{});
        MMFS_TS_WAIT_VM(0);
        issue_rows(jb, S3{});
        flush();
        if (4 * jb + 2 < nks) multiply(jb, S2{});
        MMFS_TS_WAIT_VM(0);
        issue_rows(jb + 1, S0{});                              // rows(4 jb + 4): first step of batch jb + 1
        flush();
        if (4 * jb + 3 < nks) multiply(jb, S3{});
    }
    flush();
    TSPROF(3);                                                 // rounds
    MMFS_TS_WAIT_VM(0);                                        // (requests past the end of the list: the LDS is another item's soon)
    TSPROF(4);
    TSPROF_FLUSH();
}

// One wave per work item, the grid and the item -> workgroup order of msda_bwd_tile_reduce: a (b, h) slice's workgroups are
// consecutive in the dispatch order of their XCD (h fastest), first the slice's queue of extra items, then its blocks.
template <typename T, int D>
__global__ void __launch_bounds__(64)
msda_bwd_taps_sorted(const T *__restrict__ value, const T *__restrict__ grad_out, T *__restrict__ grad_loc,
                     T *__restrict__ grad_attn, const TileReduceArgs a, const Dims d, const int blocks_grid, const int extra_grid)
{
    typedef TsGeom<D> G;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[G::LDS_BYTES];
    const int w = blockIdx.x;
    TsItem it;
    it.h = w % d.H;
    const int t = w / d.H;
    const int per_slice = extra_grid + blocks_grid;
    int jx = t % per_slice;
    it.b = t / per_slice;
    if (a.hdr->stamp != header_stamp(d)) return;                 // (a plan made for other dimensions: nothing to do)
    const int64_t bh = (int64_t)it.b * d.H + it.h;
    if (jx < extra_grid) {
        const uint32_t n = min(a.n_extra[bh], a.th->cap_extra);
        if ((uint32_t)jx >= n) return;
        const TileItem ti = a.titems[(size_t)bh * a.th->cap_extra + jx];
        if (ti.part == kVoidPart) return;                  // a reservation its block could not use
        it.part = (int)ti.part;
        const TileDesc td = a.tdesc[bh * a.blocks_bound + ti.blk];
        if ((uint32_t)it.part >= td.parts) return;
        const TapsDesc xd = a.xdesc[bh * a.blocks_bound + ti.blk];
        taps_item<T, D>(value, grad_out, grad_loc, grad_attn, a, d, td, xd, it, lds);
        return;
    }
    jx -= extra_grid;
    const int nblk = a.hdr->n_blocks4;
    if (jx >= nblk) return;
    const int blk = nblk - 1 - jx;                         // coarse levels (long lists) first
    const TileDesc td = a.tdesc[bh * a.blocks_bound + blk];
    const TapsDesc xd = a.xdesc[bh * a.blocks_bound + blk];
    it.part = 0;
    taps_item<T, D>(value, grad_out, grad_loc, grad_attn, a, d, td, xd, it, lds);
}

template <typename T, int D>
hipError_t launch_taps(const void *value, const void *go, void *gl, void *ga, const TileReduceArgs &a, const Dims &d,
                       uint32_t cap_extra, hipStream_t st)
{
    const int blocks_grid = d.blocks4 > 0 ? std::min(d.blocks4, a.blocks_bound) : a.blocks_bound;
    const int64_t items = (int64_t)d.B * d.H * ((int64_t)blocks_grid + cap_extra);
    if (items > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((msda_bwd_taps_sorted<T, D>), dim3((unsigned)items), dim3(64), 0, st,
                       (const T *)value, (const T *)go, (T *)gl, (T *)ga, a, d, blocks_grid, (int)cap_extra);
    return hipGetLastError();
}
template <typename T>
hipError_t dispatch_taps(const void *value, const void *go, void *gl, void *ga, const TileReduceArgs &a, const Dims &d,
                         uint32_t cap_extra, hipStream_t st)
{
    switch (d.D) {
        case 32: return launch_taps<T, 32>(value, go, gl, ga, a, d, cap_extra, st);
        case 64: return launch_taps<T, 64>(value, go, gl, ga, a, d, cap_extra, st);
        case 128: return launch_taps<T, 128>(value, go, gl, ga, a, d, cap_extra, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace

#ifdef MMFS_PROFILE_TS
extern "C" int mmfs_debug_ts_profile(unsigned long long *out, int reset)
{
    static unsigned long long host[kTsProfSlots * 12];
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ts_prof), sizeof(host));
    for (int i = 0; i < 16; ++i) out[i] = 0;
    for (int s = 0; s < kTsProfSlots; ++s)
        for (int i = 0; i < 12; ++i) out[i] += host[s * 12 + i];
    if (e == hipSuccess && reset) {
        for (auto &v : host) v = 0;
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_ts_prof), host, sizeof(host));
    }
    return (int)e;
}
#endif
}  // namespace blk

hipError_t backward_taps_sorted(int dtype, const void *value, const void *grad_out, void *grad_loc, void *grad_attn,
                                void *workspace, const Dims &d, hipStream_t st)
{
    if (!d.taps_sorted || !taps_sorted_supported(dtype, d)) return hipErrorInvalidValue;
    uint32_t cap_extra = 0;
    const blk::TileReduceArgs a = blk::taps_sorted_args(workspace, dtype, d, &cap_extra);
    if (!a.xdesc) return hipErrorInvalidValue;
    switch (dtype) {
        case 1: return blk::dispatch_taps<half_t>(value, grad_out, grad_loc, grad_attn, a, d, cap_extra, st);
        case 2: return blk::dispatch_taps<bf16_t>(value, grad_out, grad_loc, grad_attn, a, d, cap_extra, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mmfs
