// msda_fwd_wq.hip -- forward of multi-scale deformable attention for gfx950, fourth formulation:
// A WAVE PER QUERY, the bilinear weights on the DIAGONAL of the matrix cores' A operand, the pixel rows -- as they
// arrive from memory or from LDS, 16 bytes per lane -- as the B operand.
//
// Replaces the reference forward
//   mm_interleaved/models/utils/ops/src/cuda/ms_deform_im2col_cuda.cuh:240-302
// for 16-bit storage and heads of 128 channels (the north-star shape, BASELINE config 2).
//
// Why (round 4's counters, VERDICT r4 "what's weak" 1): msda_fwd_mma is bound by its vector ALU -- 78 % busy, and 60 % of
// its vector instructions are the unpack + fp32 multiply-add of every 16-bit element of the large levels' rows (16
// instructions per wave-wide row load), because a row as loaded (a lane = 8 consecutive channels of ONE row) is not an
// operand of a product that contracts over ROWS: the matrix cores want 8 consecutive K per lane.  But it IS the B operand
// of a product that contracts over (corner, channel-in-lane):
//
//   * a wave works on ONE query at a time.  Lane l = (j = l / 16, n = l % 16) loads 16 bytes of corner j of the current
//     sample: channels 8n .. 8n + 7 of that corner's pixel row.  One wave-wide load = the sample's whole 2x2 footprint
//     (4 rows x 256 bytes: what a load instruction of the row-gather kernels moves, too);
//   * as the B operand of v_mfma_f32_16x16x32 that register quad is B[k = 8j + i][n] = value[row_j, 8n + i];
//   * the A operand is A[m][8j + i] = (i == m % 8) ? part_{m / 8}(w_j) : 0 -- the sample's four corner weights
//     (bilinear x attention) on a diagonal, rows 0 .. 7 their leading 16 bits, rows 8 .. 15 the rounded remainder
//     (hi + lo >= 16 significant bits, as in every matrix-core kernel of this library).  A lane's fragment has ONE
//     non-zero halfword at a position that depends on the lane only: four v_and of the weight word with lane-constant
//     masks build it;
//   * D[m][n] += sum_j part(w_j) * value[row_j, 8n + m % 8]: ONE product per sample accumulates all four corners of all
//     128 channels, in place, chained over every sample of the query -- 4 vector instructions + 1 product per sample
//     where the row gather spends 64 per sample (16 per row);
//   * LDS-resident levels (the small ones: which, is decided on the device as in msda_fwd_mma) are the same product
//     with the B operand read by ds_read_b128: the image is in NATURAL channel order at a row pitch of exactly 256
//     bytes -- the fill is a linear copy of 16-byte pieces, and the four rows of a footprint fall into the read's four
//     16-lane groups without a bank conflict whatever their addresses (MI355X_MICROARCH.md, LDS: the groups of a
//     ds_read_b128 interleave two rows at complementary 64 / 128-byte pieces of the 256-byte bank row);
//   * epilogue per query: hi rows (lanes 0 .. 31) + lo rows (lanes 32 .. 63) by v_permlane32_swap, 4 channels per lane,
//     one 8-byte store.
//
// Non-finite values.  The zero entries of A multiply the OTHER channels of the same rows, so an Inf / NaN in one
// channel of a sampled row would turn its neighbours in the 8-channel group into NaN (0 x Inf) -- the reference
// (cuh:58-81, :275-299) keeps it to its own channel.  A contaminated accumulator is always non-finite, a clean one
// of finite inputs always finite: the epilogue tests the query's sums, and a query with a non-finite sum is
// recomputed by wq_exact_query -- the per-channel fp32 arithmetic of msda_fwd_vec, corners outside the map skipped.
// Results are the reference's for every input; the slow path runs only where it produces Inf / NaN.
//
// fp32 storage, other head widths, L > 64: the other formulations (msda_fwd.hip routes).
#include "msda_mma_common.h"
#include "msda_env.h"
#include "msda_launch.h"
#include <cstdlib>
#include <type_traits>

namespace mmfs {

using namespace mma;

// Development aid (tools/exp_build.sh wqprof "-DMMFS_PROFILE_WQ"; tools/wq_prof.py): shader clocks per phase of a wave
#ifdef MMFS_PROFILE_WQ
constexpr int kWProfSlots = 4096;
__device__ unsigned long long g_wq_prof[kWProfSlots * 8];
#define WPROF_DECL unsigned long long wprof_c = __builtin_readcyclecounter(), wprof_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define WPROF(i) do { const unsigned long long tn = __builtin_readcyclecounter(); wprof_t[i] += tn - wprof_c; wprof_c = tn; } while (0)
#define WPROF_COUNT(i, v) do { wprof_t[i] += (unsigned long long)(v); } while (0)
#define WPROF_FLUSH() do { if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_wq_prof[(blockIdx.x % kWProfSlots) * 8 + i_], wprof_t[i_]); } while (0)
#else
#define WPROF_DECL do {} while (0)
#define WPROF(i) do {} while (0)
#define WPROF_COUNT(i, v) do {} while (0)
#define WPROF_FLUSH() do {} while (0)
#endif

namespace wq {

constexpr int kGroup = 4;                         // queries staged together: 64 lanes = kGroup x kChunk samples
// Records of a query's chunk, in BATCHES of four samples (a lane reads one 16-byte vector of a batch's offsets and one of
// its weight words): [4 corners][4 samples] offsets, then [2 parts x 4 corners][4 samples] weight words (the 16-bit part
// in both halves) = 192 bytes; strides chosen so that the staging stores of a 32-lane group fall into 32 banks.
constexpr int kBatch = 208;                       // bytes between two batches
constexpr int kBatchW = 64;                       // where the weight words start inside a batch
constexpr int kQStride = (kChunk / 4) * kBatch;   // 832: records of one query
constexpr int kWaveRec = kGroup * kQStride;       // wave-private records

template <int D> struct Geom {
    static constexpr int RB = D * 2;              // bytes of a pixel row of one head = row pitch of the LDS image
    static constexpr int TAB_BYTES = ((kMmaMaxLevels * kTabInts * 4 + 64) + 255) & ~255;
    static constexpr int IMG0 = (TAB_BYTES + kMmaWaves * kWaveRec + 255) & ~255;
};

// Level table -> LDS and which levels live in the image (smallest first, ties by index, while they fit behind the row
// of zeros) -- msda_mma_common.h's rule at this kernel's pitch: a level is a linear run of Hl * Wl rows of RB bytes.
template <int RB>
__device__ __forceinline__ void build_table(int *tab, unsigned char *img, const int64_t *__restrict__ shapes,
                                            const int64_t *__restrict__ start, int L, int tid, int img_budget)
{
    for (int l = tid; l < L; l += kMmaThreads) {
        const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
        tab[kTabInts * l] = Hl; tab[kTabInts * l + 1] = Wl; tab[kTabInts * l + 2] = (int)start[l];
        int bytes = (Hl > 0 && Wl > 0) ? 1 << 24 : 0;                     // "never fits"; an empty level takes no room
        if (Hl > 0 && Wl > 0 && Hl <= 1024 && Wl <= 1024) {
            const int64_t bb = ((int64_t)Hl * Wl * RB + 1023) & ~1023LL;  // (whole 1 KiB blocks: what a wave's DMA request fills)
            if (bb < (1 << 24)) bytes = (int)bb;
        }
        tab[kTabInts * l + 4] = Wl * RB; tab[kTabInts * l + 5] = bytes;
    }
    __syncthreads();
    for (int l = tid; l < L; l += kMmaThreads) {
        const int px = tab[kTabInts * l] * tab[kTabInts * l + 1], bytes = tab[kTabInts * l + 5];
        int cum = 0;
        for (int l2 = 0; l2 < L; ++l2) {
            const int px2 = tab[kTabInts * l2] * tab[kTabInts * l2 + 1];
            if (px2 < px || (px2 == px && l2 <= l)) cum += tab[kTabInts * l2 + 5];
        }
        tab[kTabInts * l + 3] = (cum + 1024 <= img_budget && px > 0) ? 1024 + cum - bytes : -1;     // (the row of zeros comes first)
    }
    if (tid < RB / 4) reinterpret_cast<uint32_t *>(img)[tid] = 0u;
    __syncthreads();
}

// Resident levels global -> LDS, once per run of queries: a linear copy by DMA (buffer_load ... lds: 64 pieces of 16 bytes
// per wave request land in 1 KiB of consecutive LDS, no registers in between), every request of every level issued before
// the one wait (fill_wait: the caller stages its first samples in between) -- one round trip per run, not one per level.  A level's room in the image is a whole number of
// 1 KiB blocks (build_table), so the lanes past the end of its last block write into padding.
template <int RB>
__device__ __forceinline__ void fill_image(const int *tab, uint32_t img_lds, const u32x4 rsrc_words,
                                           uint32_t row_bytes, int L, int S, int wave, int lane)
{
    constexpr int LPR = RB / 16;
    for (int l = 0; l < L; ++l) {
        const int base = tab[kTabInts * l + 3];
        if (base < 0) continue;
        const int st = tab[kTabInts * l + 2];
        const int units = tab[kTabInts * l] * tab[kTabInts * l + 1] * LPR;
        for (int blk = wave; blk * 64 < units; blk += kMmaWaves) {
            const int u = blk * 64 + lane;
            const uint32_t p = (uint32_t)(st + u / LPR);
            const uint32_t goff = (u < units && p < (uint32_t)S) ? p * row_bytes + (uint32_t)(u % LPR) * 16u : kOobOffset;
            // (as assembly: told that LDS is being written, the compiler makes every later LDS access -- the level table, the
            //  records of the group staged meanwhile -- wait for ALL outstanding memory requests first.  Its own count of
            //  requests in flight stays safe without these: a wait for a younger load also waits for every older one.)
            const uint32_t lds_dst = img_lds + (uint32_t)(base + blk * 1024);
            asm volatile("buffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(goff), "s"(rsrc_words), "{m0}"(lds_dst) : "memory");
        }
    }
}

// ... and the wait for it: this wave's requests have landed, then every wave's
__device__ __forceinline__ void fill_wait()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// A query whose sums came out non-finite, recomputed channel by channel (exact_lane8, msda_mma_common.h): 16 lanes x 8
// channels; the other lanes of the wave compute the same and do not store.
template <typename T, int D>
__device__ __forceinline__ void exact_query(const int *tab, __amdgpu_buffer_rsrc_t rsrc, uint32_t row_bytes,
                                            const uint16_t *loc_q, const uint16_t *attn_q, int K, int P, T *out_row, int lane)
{
    static_assert(D == 128, "16 lanes x 8 channels");
    float acc[8];
    exact_lane8<T>(tab, rsrc, row_bytes, loc_q, attn_q, K, P, (uint32_t)(lane & 15) * 16u, acc);
    if (lane < 16) store16_stream(out_row + (lane & 15) * 8, Vec16<T>::pack(acc));
}

}  // namespace wq

template <typename T, int D, bool MULTI>
__global__ void __launch_bounds__(kMmaThreads)
msda_fwd_wq(const T *__restrict__ value, const int64_t *__restrict__ shapes,
            const int64_t *__restrict__ start, const T *__restrict__ loc,
            const T *__restrict__ attn, T *__restrict__ out, const Dims d, const int q_per_wg, const int img_budget,
            const int n_runs)
{
    typedef wq::Geom<D> G;
    typedef FwdMma<T> M;
    typedef Vec16<T> V;
    static_assert(D == 128, "a wave-wide load is one sample's four rows of 256 bytes");
    constexpr int QG = wq::kGroup, BS = wq::kBatch, BW = wq::kBatchW, QS = wq::kQStride;
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    int *tab = reinterpret_cast<int *>(smem);
    unsigned char *img = smem + G::IMG0;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    WPROF_DECL;
    const int L = d.L;
    const int64_t HD = (int64_t)d.H * d.D;
    const uint32_t row_bytes = (uint32_t)(HD * sizeof(T));
    wq::build_table<G::RB>(tab, img, shapes, start, L, tid, img_budget);

    // ---- lane roles
    const int cj = lane >> 4;                                             // corner whose row this lane loads / whose weight it holds
    const int cn = lane & 15;                                             // 16-byte piece of the row = B column = A row m
    const uint32_t lane_off = (uint32_t)cn * 16u;
    const uint32_t slot_off = (uint32_t)(cj * 16);                        // a batch's offsets of this lane's corner ...
    const uint32_t slot_w = (uint32_t)(BW + ((cn >> 3) * 4 + cj) * 16);   // ... and the weight words of its (part, corner)
    // A fragment: the weight word (part of w_cj in both halves) lands in halfword cn % 8 of the lane's 8 K positions
    uint32_t amask[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) amask[r] = (r == ((cn & 7) >> 1)) ? ((cn & 1) ? 0xffff0000u : 0x0000ffffu) : 0u;
    const int out_dword = 4 * cn + 2 * (cj & 1) + (lane >> 5);           // epilogue: the lane's two channels of the output row
    const bool pair_ok = ((uintptr_t)loc & (2 * sizeof(T) - 1)) == 0;
    const int kk = lane & 15, sq = lane >> 4;                             // staging role: sample of the chunk, query of the group
    int lane_lvl[4];
    {
        const int l = kk < d.K ? kk / d.P : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) lane_lvl[i] = tab[kTabInts * l + i];
    }

    for (int run = blockIdx.x; run < n_runs; run += gridDim.x) {
    const int h = run % d.H;
    const int tq = run / d.H;
    const int q_wg0 = (tq % d.q_tiles) * q_per_wg;
    const int b = tq / d.q_tiles;
    const T *slab = value + ((int64_t)b * d.S) * HD + (int64_t)h * d.D;
    const int64_t slab_bytes = ((int64_t)d.S * HD - (int64_t)h * d.D) * (int64_t)sizeof(T);
    const __amdgpu_buffer_rsrc_t rsrc = make_slab_rsrc(slab, slab_bytes);
    // (the same descriptor as four words, for the image fill's assembly: base, stride 0, extent, raw 32-bit data format)
    const u32x4 rsrc_words = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)slab),
                              (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uintptr_t)slab >> 32)) & 0xffffu,
                              (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)slab_bytes), 0x00020000u};

    unsigned char *wrec = smem + G::TAB_BYTES + wave * wq::kWaveRec;
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(lds_u8 *)smem;        // (LDS addresses are 32-bit)
    const uint32_t wrec_lds = smem_lds + (uint32_t)(G::TAB_BYTES + wave * wq::kWaveRec);
    const uint32_t img_lane = smem_lds + (uint32_t)G::IMG0 + lane_off;

    const int q_wg1 = min(d.Nq, q_wg0 + q_per_wg);
    const int n_chunks = (d.K + kChunk - 1) / kChunk;
    const int q_first = q_wg0 + wave * QG;
    const int n_groups = q_first < q_wg1 ? (q_wg1 - q_first + kMmaWaves * QG - 1) / (kMmaWaves * QG) : 0;
    const int n_steps = n_groups * n_chunks;
    uint32_t pf_w0, pf_w1, pf_a;                                          // the next step's sample words, raw
    const uint16_t *loc_wg = reinterpret_cast<const uint16_t *>(loc) + 2 * (((int64_t)b * d.Nq * d.H + h) * d.K);
    const uint16_t *attn_wg = reinterpret_cast<const uint16_t *>(attn) + (((int64_t)b * d.Nq * d.H + h) * d.K);
    const uint32_t q_stride = (uint32_t)d.H * (uint32_t)d.K;
    auto prefetch = [&](int step) {
        const int q = q_first + (step / n_chunks) * (kMmaWaves * QG) + sq;
        const int k = (step % n_chunks) * kChunk + kk;
        pf_w0 = pf_w1 = pf_a = 0u;
        if (step < n_steps && k < d.K && q < q_wg1) {
            const uint32_t s = (uint32_t)q * q_stride + (uint32_t)k;
            const uint16_t *lw = loc_wg + 2 * (size_t)s;
            if (pair_ok) pf_w0 = *reinterpret_cast<const uint32_t *>(lw);
            else { pf_w0 = lw[0]; pf_w1 = lw[1]; }
            pf_a = attn_wg[s];
        }
    };
    prefetch(0);
    if (run != (int)blockIdx.x) __syncthreads();                          // every wave is done with the previous image
    WPROF(0);
    // (the first group's sample words are waited for HERE -- they were requested before the barrier -- so that its staging,
    //  which runs while the image is on its way, meets no memory wait of its own)
    asm volatile("" : "+v"(pf_w0), "+v"(pf_w1), "+v"(pf_a));
    wq::fill_image<G::RB>(tab, smem_lds + (uint32_t)G::IMG0, rsrc_words, row_bytes, L, d.S, wave, lane);    // (requests only)
    if (n_steps == 0) wq::fill_wait();

    // (n_chunks == 1, K <= 16 -- the north-star shape: a query is finished when its products are, one accumulator lives;
    //  else the group's four accumulators are carried across the chunks)
    f32x4 acc[MULTI ? QG : 1];
#pragma unroll
    for (int i = 0; i < (MULTI ? QG : 1); ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int step = 0; step < n_steps; ++step) {
        // A wave that is behind the others of its SIMD -- more steps left -- issues first.  Left to the arbiter (oldest wave
        // first) the waves of a workgroup finished a run up to a quarter of its length apart, and the barrier in front of the
        // next image fill waited for the last: 11.3 k -> 6.1 k clocks per wave and run, forward 120.9 -> 116.4 us (r05f).
#ifndef WQ_NO_SETPRIO
        { const int left = n_steps - 1 - step; if (left >= 3) __builtin_amdgcn_s_setprio(3); else if (left == 2) __builtin_amdgcn_s_setprio(2); else if (left == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#endif
        const int q0 = q_first + (step / n_chunks) * (kMmaWaves * QG);
        const int chunk = step % n_chunks;
        const int k0 = chunk * kChunk;
        unsigned long long live_g, live_l;                                // by lane (16 * query + sample): contributes something
        {
            // ---- stage: one sample per lane (its words arrived during the previous step)
            const int k = k0 + kk;
            const int q = q0 + sq;
            const bool k_ok = k < d.K && q < q_wg1;
            // (one chunk per query: a lane stages the same sample index, hence the same level, all along -- lane_lvl)
            int Hl = lane_lvl[0], Wl = lane_lvl[1], lstart = lane_lvl[2], ibase = lane_lvl[3];
            if constexpr (MULTI) {
                const int l = k < d.K ? k / d.P : 0;
                Hl = tab[kTabInts * l]; Wl = tab[kTabInts * l + 1]; lstart = tab[kTabInts * l + 2]; ibase = tab[kTabInts * l + 3];
            }
            const bool in_lds = ibase >= 0;
            // (straight-line: a lane without a sample decodes zeros and writes nothing)
            uint32_t off[4], whi[4], wlo[4];
            bool weighs;
            {
                asm volatile("" : "+v"(pf_w0), "+v"(pf_w1), "+v"(pf_a));  // (the decode stays below the loads' wait)
                const uint32_t xb = pair_ok ? (pf_w0 & 0xffffu) : pf_w0, yb = pair_ok ? (pf_w0 >> 16) : pf_w1;
                const float lx = to_f32(__builtin_bit_cast(T, (uint16_t)xb)), ly = to_f32(__builtin_bit_cast(T, (uint16_t)yb));
                const float a = to_f32(__builtin_bit_cast(T, (uint16_t)pf_a));
                const float y = ly * (float)Hl - 0.5f, x = lx * (float)Wl - 0.5f;
                // strict comparisons: NaN fails, exactly -1 / Hl / Wl fail (cuh:291)
                const bool inside = k_ok && (y > -1.f) && (x > -1.f) && (y < (float)Hl) && (x < (float)Wl);
                const float yf = floorf(y), xf = floorf(x);
                const int y0 = inside ? (int)yf : 0, x0 = inside ? (int)xf : 0;
                const float fy = y - yf, fx = x - xf;
                const float gy = 1.f - fy, gx = 1.f - fx;
                const bool on = inside && a != 0.f;                        // a zero attention weight reads nothing
                const bool top = y0 >= 0, left = x0 >= 0, bottom = y0 + 1 <= Hl - 1, right = x0 + 1 <= Wl - 1;
                const bool ok[4] = {on && top && left, on && top && right, on && bottom && left, on && bottom && right};
                const float ga = gy * a, fa = fy * a;
                const float w[4] = {ok[0] ? ga * gx : 0.f, ok[1] ? ga * fx : 0.f, ok[2] ? fa * gx : 0.f, ok[3] ? fa * fx : 0.f};
                // offsets: the image (rows of RB bytes behind the level's base; 0 = the row of zeros) or the slab in memory
                // (rows of row_bytes; "outside" = past the descriptor's extent)
                const uint32_t unit = in_lds ? (uint32_t)G::RB : row_bytes;
                const uint32_t o00 = (in_lds ? (uint32_t)ibase : (uint32_t)lstart * row_bytes) + (uint32_t)(y0 * Wl + x0) * unit;
                const uint32_t none = in_lds ? 0u : kOobOffset;
                const uint32_t orow[4] = {o00, o00 + unit, o00 + (uint32_t)Wl * unit, o00 + (uint32_t)Wl * unit + unit};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    M::split_dup(w[c], whi[c], wlo[c]);
                    off[c] = ok[c] ? orow[c] : none;
#ifdef WQ_NO_GLOBAL                                                      // knock-out: every row-gather load is "outside" (no data moves)
                    if (!in_lds) off[c] = kOobOffset;
#endif
#ifdef WQ_SAME_GLOBAL                                                    // knock-out: every row-gather load hits the same four rows
                    if (!in_lds) off[c] = (uint32_t)c * row_bytes;
#endif
#ifdef WQ_SAME_LDS                                                       // knock-out: every LDS sample reads the row of zeros
                    if (in_lds) off[c] = 0u;
#endif
                }
                // (a valid corner of weight 0 still multiplies its row, as in the reference: 0 x Inf is NaN there too)
                weighs = ok[0] || ok[1] || ok[2] || ok[3];
            }
            // Which samples get a record.  Samples that contribute nothing (outside the map, a zero attention weight) are
            // left out and a query's records COMPACTED -- but only when that leaves the group regular: the same samples live
            // in all four queries (an image no token of the group can see: whole levels drop out, half the work).  A ragged
            // pattern (samples over the border here and there) would send the group down the generic, unpipelined path for a
            // handful of skipped samples (clustered locations: 180 us against 122, r05v): then every sample keeps its place,
            // the dead ones as records of zero weight that point at the row of zeros / past the slab's end.
            const unsigned long long st_g = __builtin_amdgcn_ballot_w64(k_ok && !in_lds), st_l = __builtin_amdgcn_ballot_w64(k_ok && in_lds);
            unsigned long long bl_g = __builtin_amdgcn_ballot_w64(weighs && !in_lds);
            unsigned long long bl_l = __builtin_amdgcn_ballot_w64(weighs && in_lds);
            const bool same4 = bl_g == (bl_g & 0xffffull) * 0x0001000100010001ull && bl_l == (bl_l & 0xffffull) * 0x0001000100010001ull;
            const bool compact = same4 || (bl_g == st_g && bl_l == st_l);
            if (!compact) { bl_g = st_g; bl_l = st_l; }
            const bool writes = compact ? weighs : k_ok;
            live_g = bl_g; live_l = bl_l;
            // records of a query: the live row-gather samples from the bottom, the live LDS samples from the top
            const uint32_t mine_g = (uint32_t)(bl_g >> (16 * sq)) & 0xffffu, mine_l = (uint32_t)(bl_l >> (16 * sq)) & 0xffffu;
            const uint32_t below = (1u << kk) - 1u;
            const int ridx = in_lds ? kChunk - 1 - __builtin_popcount(mine_l & below) : __builtin_popcount(mine_g & below);
            wave_sync();                                                  // the previous step's records are consumed
            if (writes) {
                // [corner][sample]: load u of a batch carries the four corners of its sample u.  (The other way round -- lane
                // group j = corner u of sample j, four rows of four different samples per load, three 16-byte stores here --
                // measured the same on uniform locations, r05w, and costs the generic path a select per lane.)
                uint32_t *dst = reinterpret_cast<uint32_t *>(wrec + sq * QS + (ridx >> 2) * BS + (ridx & 3) * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    dst[4 * c] = off[c];
                    dst[BW / 4 + 4 * c] = whi[c];
                    dst[BW / 4 + 16 + 4 * c] = wlo[c];
                }
            }
            wave_sync();
        }
        if (step == 0) {
            WPROF(1);
            wq::fill_wait();
            WPROF(4);
        }
        prefetch(step + 1);
        __builtin_amdgcn_sched_barrier(0);
        WPROF(1);

        // ---- the group's queries, one after the other: every sample one product
        auto lds_u4 = [](uint32_t addr) -> uint4 {
            const u32x4 v = *(const __attribute__((address_space(3))) u32x4 *)(uintptr_t)addr;
            return make_uint4(v[0], v[1], v[2], v[3]);
        };
        auto product = [&](f32x4 &a4, const uint4 &rows, uint32_t ww) {
#ifdef WQ_NO_MFMA                                                        // knock-out experiment: the rows arrive, nothing multiplies
            a4[0] += __uint_as_float((ww & amask[0]) ^ rows.x ^ rows.y ^ rows.z ^ rows.w);
#else
            const uint4 aw = make_uint4(ww & amask[0], ww & amask[1], ww & amask[2], ww & amask[3]);
            a4 = M::run(__builtin_bit_cast(s16x8, aw), __builtin_bit_cast(s16x8, rows), a4);
#endif
        };
        auto at = [](const uint4 &v, int u) -> uint32_t { return u == 0 ? v.x : u == 1 ? v.y : u == 2 ? v.z : v.w; };
        // hi rows (lanes 0 .. 31: g = 0, 1) + lo rows (lanes 32 .. 63) of register i are channel 8n + 4g + i.  One
        // v_permlane32_swap of registers (0, 2) puts the two halves of register 0 side by side in lanes 0 .. 31 and those of
        // register 2 in lanes 32 .. 63: one add sums both; the same for (1, 3).  A lane then holds two consecutive channels
        // -- 8n + 4g + {0, 1} below lane 32, {2, 3} above -- one 4-byte store each.  Returns whether the query's sums are all
        // finite (and stored); else nothing is stored and the caller recomputes the query.
        auto finish = [&](f32x4 &a4, int q) -> bool {
            const auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a4[0]), __float_as_uint(a4[2]), false, false);
            const auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a4[1]), __float_as_uint(a4[3]), false, false);
            const float e = __uint_as_float(s02[0]) + __uint_as_float(s02[1]);
            const float f = __uint_as_float(s13[0]) + __uint_as_float(s13[1]);
            a4 = f32x4{0.f, 0.f, 0.f, 0.f};
            const float t = fmaf(f, 0.f, e * 0.f);
            if (__builtin_amdgcn_ballot_w64(t != t) != 0ull) return false;
            T *orow = out + (((int64_t)b * d.Nq + q) * d.H + h) * d.D;
            __builtin_nontemporal_store(V::pk(e, f), reinterpret_cast<uint32_t *>(orow) + out_dword);
            return true;
        };
        // Any chunk of any query: batch after batch, a sample from memory (position < n_g), from the image (position >= 16 -
        // n_l) or nobody's; four rows in flight.
        auto any_query = [&](const uint32_t rq, const int n_g, const int n_l, f32x4 &a4) {
#pragma unroll 1
            for (int bt = 0; bt < kChunk / 4; ++bt) {
                if (4 * bt >= n_g && 4 * bt + 3 < kChunk - n_l) continue;
                const uint4 o4 = lds_u4(rq + (uint32_t)(bt * BS) + slot_off), w4 = lds_u4(rq + (uint32_t)(bt * BS) + slot_w);
                uint4 rows[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int pos = 4 * bt + u;
                    if (pos < n_g) rows[u] = buffer_load16(rsrc, at(o4, u) + lane_off);
                    else if (pos >= kChunk - n_l) rows[u] = lds_u4(img_lane + at(o4, u));
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int pos = 4 * bt + u;
                    if (pos < n_g || pos >= kChunk - n_l) product(a4, rows[u], at(w4, u));
                }
            }
        };
        auto n_live = [&](unsigned long long m, int qi) { return __builtin_popcount((uint32_t)(m >> (16 * qi)) & 0xffffu); };
        uint32_t redo = 0u;                                               // queries of the group whose sums were not finite
        if constexpr (!MULTI) {
            // The regular group -- four queries, each with NG4 full batches of row-gather samples and NL4 of LDS samples, no
            // tails (the north-star pyramid: 2 + 2) -- as straight-line code with the next query's rows in flight:
            //   issue(0) | lds(0) issue(1) gather(0) | lds(1) issue(2) gather(1) | lds(2) issue(3) gather(2) | lds(3) gather(3)
            auto regular = [&](auto NG4c, auto NL4c) {
                constexpr int NG4 = decltype(NG4c)::value, NL4 = decltype(NL4c)::value;
                uint4 B[2][NG4 ? NG4 : 1][4];
                uint4 W[2][NG4 ? NG4 : 1];
                auto issue_q = [&](int qi, int set) {
#pragma unroll
                    for (int bt = 0; bt < NG4; ++bt) {
                        const uint32_t ra = wrec_lds + (uint32_t)(qi * QS + bt * BS);
                        const uint4 o4 = lds_u4(ra + slot_off);
                        W[set][bt] = lds_u4(ra + slot_w);
#pragma unroll
                        for (int u = 0; u < 4; ++u) B[set][bt][u] = buffer_load16(rsrc, at(o4, u) + lane_off);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                issue_q(0, 0);
#pragma unroll
                for (int qi = 0; qi < QG; ++qi) {
                    // (the LDS samples' two round trips -- records, then rows -- once per query, not once per batch: while they
                    //  run only ONE row-gather set is live, so the registers are there; 117.0 -> 114.0 us, r05n)
                    {
                        uint4 o4[NL4 ? NL4 : 1], w4[NL4 ? NL4 : 1];
                        uint4 rows[NL4 ? NL4 : 1][4];
#pragma unroll
                        for (int bt = 0; bt < NL4; ++bt) {
                            const uint32_t ra = wrec_lds + (uint32_t)(qi * QS + (kChunk / 4 - 1 - bt) * BS);
                            o4[bt] = lds_u4(ra + slot_off); w4[bt] = lds_u4(ra + slot_w);
                        }
#pragma unroll
                        for (int bt = 0; bt < NL4; ++bt) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) rows[bt][u] = lds_u4(img_lane + at(o4[bt], u));
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int bt = 0; bt < NL4; ++bt) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) product(acc[0], rows[bt][u], at(w4[bt], u));
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (qi + 1 < QG) issue_q(qi + 1, (qi + 1) & 1);
#pragma unroll
                    for (int bt = 0; bt < NG4; ++bt) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) product(acc[0], B[qi & 1][bt][u], at(W[qi & 1][bt], u));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (!finish(acc[0], q0 + qi)) redo |= 1u << qi;
                }
            };
            // (wave-uniform: every query of the group has all its samples live, the same split for all four)
            const bool all4 = q0 + QG <= q_wg1;
            const uint32_t g0 = (uint32_t)live_g & 0xffffu, l0 = (uint32_t)live_l & 0xffffu;
            const bool same = all4 && live_g == g0 * 0x0001000100010001ull && live_l == l0 * 0x0001000100010001ull;
            const int ng = __builtin_popcount(g0), nl = __builtin_popcount(l0);
            if (same && ng == 8 && nl == 8) regular(std::integral_constant<int, 2>(), std::integral_constant<int, 2>());
            else if (same && ng == 0 && nl == 16) regular(std::integral_constant<int, 0>(), std::integral_constant<int, 4>());
            else {
#pragma unroll 1
                for (int qi = 0; qi < QG; ++qi) {
                    if (q0 + qi >= q_wg1) break;
                    any_query(wrec_lds + (uint32_t)(qi * QS), n_live(live_g, qi), n_live(live_l, qi), acc[0]);
                    if (!finish(acc[0], q0 + qi)) redo |= 1u << qi;
                }
            }
            WPROF(2);
        } else {
            // (a rolled loop over the queries: the accumulator of query qi moves through a4 by wave-uniform selects)
#pragma unroll 1
            for (int qi = 0; qi < QG; ++qi) {
                if (q0 + qi >= q_wg1) break;
                f32x4 a4 = qi == 0 ? acc[0] : qi == 1 ? acc[1 % (MULTI ? QG : 1)] : qi == 2 ? acc[2 % (MULTI ? QG : 1)] : acc[3 % (MULTI ? QG : 1)];
                any_query(wrec_lds + (uint32_t)(qi * QS), n_live(live_g, qi), n_live(live_l, qi), a4);
                if (chunk == n_chunks - 1) {
                    if (!finish(a4, q0 + qi)) redo |= 1u << qi;
                }
#pragma unroll
                for (int j = 0; j < (MULTI ? QG : 1); ++j) acc[j] = qi == j ? a4 : acc[j];
            }
            WPROF(2);
        }
        // (the next step's sample words have arrived long since: waiting for them HERE leaves the top of the loop without a
        //  pending request of the compiler's own, so that the staging of a run's first group does not wait for the image)
        asm volatile("" : "+v"(pf_w0), "+v"(pf_w1), "+v"(pf_a));
        // ---- the rare query with a non-finite sum, channel by channel (see the header)
        while (redo) {
            const int qi = __builtin_ctz(redo);
            redo &= redo - 1u;
            const int q = q0 + qi;
            const uint16_t *lq = loc_wg + 2 * (size_t)((uint32_t)q * q_stride);
            const uint16_t *aq = attn_wg + (size_t)((uint32_t)q * q_stride);
            wq::exact_query<T, D>(tab, rsrc, row_bytes, lq, aq, d.K, d.P, out + (((int64_t)b * d.Nq + q) * d.H + h) * d.D, lane);
        }
    }
    WPROF_COUNT(7, n_steps);
    }   // runs
    WPROF_FLUSH();
}

// ---------------------------------------------------------------- launcher
template <typename T, int D>
static hipError_t launch_wq(const void *value, const int64_t *shapes, const int64_t *start,
                            const void *loc, const void *attn, void *out, Dims d, hipStream_t st)
{
    typedef wq::Geom<D> G;
    static const hipError_t once1 = hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fwd_wq<T, D, false>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal);
    static const hipError_t once2 = hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fwd_wq<T, D, true>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal);
    if (once1 != hipSuccess) return once1;
    if (once2 != hipSuccess) return once2;
    const int env_kb = knob_int(K_FWD_WQ_LDS_KB, 0);       // tuning / tests
    const int lds_total = env_kb > 0 ? std::min(kLdsTotal, std::max(G::IMG0 + 1024, env_kb * 1024)) : kLdsTotal;
    const int env_q = knob_int(K_FWD_WQ_QPW, 0);
    const int unit = kMmaWaves * wq::kGroup;
    const int q_per_wg = pick_queries_per_run(d, unit, env_q);       // (256, or shorter runs for few queries: msda_mma_common.h)
    d.q_tiles = (d.Nq + q_per_wg - 1) / q_per_wg;
    const int64_t runs = (int64_t)d.B * d.q_tiles * d.H;
    if (runs > 0x7fffffffLL) return hipErrorInvalidValue;
    const int grid = (int)persistent_grid(runs, d.H);
    if (d.K <= kChunk)
        hipLaunchKernelGGL((msda_fwd_wq<T, D, false>), dim3((unsigned)grid), dim3(kMmaThreads), lds_total, st,
                           (const T *)value, shapes, start, (const T *)loc, (const T *)attn, (T *)out, d, q_per_wg,
                           lds_total - G::IMG0, (int)runs);
    else
        hipLaunchKernelGGL((msda_fwd_wq<T, D, true>), dim3((unsigned)grid), dim3(kMmaThreads), lds_total, st,
                           (const T *)value, shapes, start, (const T *)loc, (const T *)attn, (T *)out, d, q_per_wg,
                           lds_total - G::IMG0, (int)runs);
    return hipGetLastError();
}

bool fwd_wq_supported(int dtype, const Dims &d)
{
    if (dtype != 1 && dtype != 2) return false;
    if (d.D != 128) return false;
    if (d.L > kMmaMaxLevels || d.K <= 0) return false;
    if ((int64_t)d.Nq * d.H * d.K >= (1LL << 30)) return false;            // 32-bit sample offsets inside a (b, h) slab
    return (int64_t)d.S * d.H * d.D * 2 <= kMaxSlabBytes;
}

bool fwd_wq_applies(int dtype, const Dims &d)
{
    const char *algo = knob_str(K_FWD_ALGO);                           // "wq": whenever the shape allows
    if (algo && algo[0] != 'w') return false;
    if (!fwd_wq_supported(dtype, d)) return false;
    if (algo && algo[0] == 'w') return true;
    // one chunk of samples per query (K <= 16: the north-star shape): the straight-line pipeline.  Longer sample lists run
    // chunk by chunk through the generic path and measured slower than msda_fwd_mma (the reference's speed-test shape,
    // K = 128: 156 vs 101 us, r05v): they keep the LDS-resident formulation
    // ... and only launches that give every CU two runs of queries (enough_runs, msda_mma_common.h: r05ac)
    return d.K <= kChunk && d.Nq >= 64 && (int64_t)d.Nq * d.K >= 4096 && enough_runs(d);
}

hipError_t forward_wq(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                      const void *loc, const void *attn, void *out, const Dims &d, hipStream_t st)
{
    if (dtype == 1) return launch_wq<half_t, 128>(value, shapes, start, loc, attn, out, d, st);
    return launch_wq<bf16_t, 128>(value, shapes, start, loc, attn, out, d, st);
}

}  // namespace mmfs

#ifdef MMFS_PROFILE_WQ
extern "C" int mmfs_debug_wq_profile(unsigned long long *out, int reset)
{
    static unsigned long long host[mmfs::kWProfSlots * 8];
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(mmfs::g_wq_prof), sizeof(host));
    for (int i = 0; i < 8; ++i) out[i] = 0;
    for (int s = 0; s < mmfs::kWProfSlots; ++s)
        for (int i = 0; i < 8; ++i) out[i] += host[s * 8 + i];
    if (e == hipSuccess && reset) {
        for (auto &v : host) v = 0;
        e = hipMemcpyToSymbol(HIP_SYMBOL(mmfs::g_wq_prof), host, sizeof(host));
    }
    return (int)e;
}
#endif
