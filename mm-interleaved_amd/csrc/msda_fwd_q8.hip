// msda_fwd_q8.hip -- forward of multi-scale deformable attention for gfx950, third formulation:
// the head is cut into SLICES of 32 channels, a workgroup owns one slice of one (batch, head), and every level whose
// slice fits in the CU's LDS is sampled by the matrix cores in tiles of EIGHT queries.
//
// Replaces the reference forward
//   mm_interleaved/models/utils/ops/src/cuda/ms_deform_im2col_cuda.cuh:240-302
// for 16-bit storage and head widths that are multiples of 32 channels -- first of all the geometry the reference
// actually runs: H = 16, P = 8, D = 64 (decoders/modeling_llama_mmfs.py:326-339, decoders/sd_mmfs.py:50-53), which
// round 3 left on the row-gather kernel at 7-8 % of the HBM roofline (missing #2).
//
// Why slices.  A level is worth keeping in LDS when its rows fit: at full head width (msda_fwd_mma.hip) that is
// 16x16 + 8x8 for a head of 128 channels and nothing larger; of a 64-byte slice of every pixel row, 32x32 + 16x16 + 8x8
// fit (101 KB with their zero borders) -- three of the four levels of the image decoder's pyramid, ALL levels of the
// LLM's (one image).  What a slice costs is the per-sample arithmetic (locations, bilinear weights), repeated per
// slice: twice for heads of 64 channels, four times for 128.
//
// Organisation:
//   * a 1024-lane workgroup per run of queries of one (batch, head, slice), persistent (msda_mma_common.h); the
//     resident levels' slices are copied into an LDS image once per run: natural channel order, a border of zero
//     pixels around every level (line -1, line H, and >= 2 zero pixels between the lines), so that a sample's four
//     corners are ALWAYS four addresses of the image -- corners outside the map read zeros (cuh:58-81) without a
//     test -- and one 4-byte offset per sample says where;
//   * the 16 waves work on their own: a wave takes tiles of 8 queries; a PASS stages 8 samples of each (one per
//     lane: location, bilinear weights x attention weight, as leading 16 bits + rounded remainder: hi + lo >= 16
//     significant bits, inside the storage type's rounding) into wave-private LDS records;
//   * a K-BLOCK = sample k of all 8 queries = 32 pixel rows = ONE v_mfma_f32_16x16x32 per 16 channels: the B operand
//     are the rows, fetched from the image with the transposing ds_read_b64_tr_b16 (every lane supplies the address
//     of 8 bytes of "its" row); the A operand holds the tile's weights block-diagonally -- row 2j (hi) and 2j + 1 (lo)
//     of the product are query j, non-zero only in the four K positions of query j's corners -- so the 16 rows of ONE
//     accumulator tile are the 8 queries, and a K-block's products chain into it across all samples (no adds in the
//     vector ALU, no per-query product).  Round 3's kernel spent one product per QUERY and batch of 8 samples and
//     added its two useful rows by hand;
//   * bank conflicts: the lines of a level are padded to a pixel count = 2 (mod 4), which puts the four corners of
//     any sample into the four residues of the row index mod 4 = four different 64-byte halves of the bank rows;
//     and of the two samples whose rows a 32-lane group of a transposing read touches, the second reads the OTHER
//     32-byte piece of its rows -- legal because A is block-diagonal: those K rows only feed their own queries'
//     output rows, whose two accumulators simply swap roles (undone in the epilogue).  Eight rows, eight bank slots;
//   * levels that do not fit (64x64 of the image decoder) take a row gather inside the same kernel: 16 rows of 64
//     bytes per load instruction, fp32 multiply-add, four K-blocks in flight while the resident ones multiply;
//   * epilogue: hi + lo rows and the row-gather partial sums meet in LDS, one 16-byte store per (query, 8 channels).
//
// fp32 storage, head widths that are not multiples of 32, L > 64: msda_fwd_vec (msda_fwd.hip).
#include "msda_mma_common.h"
#include "msda_env.h"
#include "msda_launch.h"
#include <cstdlib>
#include <type_traits>

namespace mmfs {

using namespace mma;

// Development aid (tools/exp_build.sh q8prof "-DMMFS_PROFILE_Q8"; tools/q8_prof.py): shader clocks per phase of a wave,
// summed over waves per workgroup slot, read back with mmfs_debug_q8_profile().
#ifdef MMFS_PROFILE_Q8
constexpr int kQProfSlots = 4096;
__device__ unsigned long long g_q8_prof[kQProfSlots * 8];
#define QPROF_DECL unsigned long long qprof_c = __builtin_readcyclecounter(), qprof_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define QPROF(i) do { const unsigned long long tn = __builtin_readcyclecounter(); qprof_t[i] += tn - qprof_c; qprof_c = tn; } while (0)
#define QPROF_COUNT(i, v) do { qprof_t[i] += (unsigned long long)(v); } while (0)
#define QPROF_FLUSH() do { if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_q8_prof[(blockIdx.x % kQProfSlots) * 8 + i_], qprof_t[i_]); } while (0)
#else
#define QPROF_DECL do {} while (0)
#define QPROF(i) do {} while (0)
#define QPROF_COUNT(i, v) do {} while (0)
#define QPROF_FLUSH() do {} while (0)
#endif


// Development aid (tools/debug/q8_probe.py; -DQ8_PROBE): at the end of a pass's staging every lane reads back what it stored for
// corner 2 of its sample and compares it with the weight recomputed from copies of its factors; mismatches are logged.
#ifdef Q8_PROBE
__device__ unsigned int g_q8_probe_n;
__device__ unsigned int g_q8_probe[4096 * 8];
#endif

// Round 4 left four "guards" in this kernel (products as inline assembly with wait states, idle cycles behind the staging
// stores, no ds_read2, registers kept alive) against a fault it could only describe: queries 6 and 7 of a tile (the lanes
// 48..63) wrong now and then, fp16 only, under load only.  Round 6 decoded it (profiles/r06_experiments.md, "the sliced
// forward's heisenbug, decoded"; tools/ubench/pk_opsel_mfma.hip): MI355X computes  v_pk_mul_f32 ... op_sel:[0,1]  -- which
// hipcc makes of the fp16 kernel's bilinear weights {fy * gx, fx * gy} -- with src1's high register read as ZERO in the
// lanes 48..63 whenever another wave of the SIMD is running a matrix product.  The guards only moved the compiler away
// from that instruction by accident; the library is now built with such instructions' operands exchanged
// (tools/fix_pk_opsel.py, csrc/Makefile; tests/test_isa_lint.py), the guards are gone, and -DQ8_PROBE below is how the
// fault was caught in the act (tools/debug/q8_probe.py; RAW=1 tools/exp_build1.sh builds the unrepaired kernel).

namespace q8 {


typedef uint32_t u32v4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32v2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32v4 lds_u32v4;       // (plain vector types: the HIP structs cannot be copied out of an address space)
typedef __attribute__((address_space(3))) u32v2 lds_u32v2;
constexpr int kRB = 64;                 // bytes of a slice's pixel row
constexpr int kCS = 32;                 // channels of a slice
constexpr int kQT = 8;                  // queries of a wave's tile
constexpr int kTab = 8;                 // ints per level in the LDS table: H, W, start, image base (-1), line pixels, bytes
constexpr int kTabBytes = kMmaMaxLevels * kTab * 4;
constexpr int kRec = 304;               // bytes between the records of two K-blocks: [0, 288) a resident K-block's offsets (32) +
                                        // hi (128) + lo (128) fragments, or a row-gather K-block's offsets + weights (256);
                                        // [288, 304) stays zero.  (256 + 48: the staging writes of a pass's 8 K-blocks
                                        // fall into different banks)
constexpr int kScratch = 3072;          // wave-private LDS: 8 records; the epilogue's 1 KB + 2 KB of fp32 sums
constexpr int kMaxK = 128;              // samples per query (L * P) the per-sample-index table has room for
constexpr int kKtabBytes = kMaxK * 32;  // per sample index: {H, W as floats, image base (-1), bytes per line} {first pixel, W, H, -}
constexpr int kImg0 = (kTabBytes + kKtabBytes + kMmaWaves * kScratch + 255) & ~255;
static_assert(8 * kRec <= kScratch, "records of a pass");

// pixels per line of a W-pixel-wide level in the image: W + at least two zero pixels, = 2 (mod 4)
__host__ __device__ __forceinline__ int line_pixels(int W) { return W + 2 + ((4 - (W & 3)) & 3); }

// Level table -> LDS; which levels live in the image (smallest first, ties: lower index, while they fit); the image
// cleared once (its borders stay zero: the fills only write the pixels).  Every thread; ends with a barrier.
__device__ __forceinline__ void build_table(int *tab, unsigned char *smem, const int64_t *__restrict__ shapes,
                                            const int64_t *__restrict__ start, int L, int K, int P, int tid, int lds_total)
{
    for (int l = tid; l < L; l += kMmaThreads) {
        const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
        int bytes = (Hl > 0 && Wl > 0) ? 1 << 24 : 0;                   // "never fits"; an empty level takes no room
        int lw = 0;
        if (Hl > 0 && Wl > 0 && Hl <= 2048 && Wl <= 2048) {
            lw = line_pixels(Wl);
            const int64_t bb = (int64_t)(Hl + 2) * lw * kRB;
            if (bb < (1 << 24)) bytes = (int)bb;
        }
        tab[kTab * l] = Hl; tab[kTab * l + 1] = Wl; tab[kTab * l + 2] = (int)start[l];
        tab[kTab * l + 4] = lw; tab[kTab * l + 5] = bytes;
    }
    for (int i = tid; i < (lds_total - kImg0) / 16; i += kMmaThreads)
        reinterpret_cast<uint4 *>(smem + kImg0)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    for (int l = tid; l < L; l += kMmaThreads) {
        const int px = tab[kTab * l] * tab[kTab * l + 1], bytes = tab[kTab * l + 5];
        int cum = 0;
        for (int l2 = 0; l2 < L; ++l2) {
            const int px2 = tab[kTab * l2] * tab[kTab * l2 + 1];
            if (px2 < px || (px2 == px && l2 <= l)) cum += tab[kTab * l2 + 5];
        }
        tab[kTab * l + 3] = (px > 0 && kImg0 + cum <= lds_total) ? kImg0 + cum - bytes : -1;      // (byte offset in the workgroup's LDS)
    }
    __syncthreads();
    // what the staging needs of sample index k's level, as it needs it
    uint4 *ktab = reinterpret_cast<uint4 *>(smem + kTabBytes);
    for (int k = tid; k < K && k < kMaxK; k += kMmaThreads) {
        const int l = k / P;
        const int Hl = tab[kTab * l], Wl = tab[kTab * l + 1];
        ktab[2 * k] = make_uint4(__float_as_uint((float)Hl), __float_as_uint((float)Wl), (uint32_t)tab[kTab * l + 3],
                                 (uint32_t)(tab[kTab * l + 4] * kRB));
        ktab[2 * k + 1] = make_uint4((uint32_t)tab[kTab * l + 2], (uint32_t)Wl, (uint32_t)Hl, 0u);
    }
    __syncthreads();
}

// Resident levels' slices global -> LDS, natural channel order, four 16-byte pieces in flight per lane.  Ends with a barrier.
__device__ __forceinline__ void fill_image(const int *tab, unsigned char *smem, __amdgpu_buffer_rsrc_t rsrc,
                                           uint32_t row_bytes, int L, int S, int tid)
{
    constexpr int NB = 4;
    for (int l = 0; l < L; ++l) {
        const int base = tab[kTab * l + 3];
        if (base < 0) continue;
        const int Hl = tab[kTab * l], Wl = tab[kTab * l + 1], st = tab[kTab * l + 2], lw = tab[kTab * l + 4];
        const int units = Hl * Wl * (kRB / 16);
        const float inv_w = 1.0f / (float)Wl;
        for (int u0 = tid; u0 < units; u0 += NB * kMmaThreads) {
            uint4 raw[NB];
            int dst[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int u = u0 + i * kMmaThreads;
                dst[i] = -1;
                raw[i] = make_uint4(0u, 0u, 0u, 0u);
                if (u < units) {
                    const int p = u >> 2, c = u & 3;
                    int y = (int)((float)p * inv_w);                       // p / Wl for p < 2^22: a float guess, corrected
                    y -= (y * Wl > p); y += ((y + 1) * Wl <= p);
                    const int x = p - y * Wl;
                    const uint32_t goff = (uint32_t)(st + p) < (uint32_t)S ? (uint32_t)(st + p) * row_bytes + (uint32_t)c * 16u : kOobOffset;
                    raw[i] = buffer_load16(rsrc, goff);
                    dst[i] = base + ((y + 1) * lw + x + 1) * kRB + 16 * c;
                }
            }
#pragma unroll
            for (int i = 0; i < NB; ++i)
                if (dst[i] >= 0) *reinterpret_cast<uint4 *>(smem + dst[i]) = raw[i];
        }
    }
    __syncthreads();
}

// Eight channels of one query, recomputed with the reference's per-channel arithmetic (cuh:275-299): fp32 multiply-add of
// every element, corners outside the map skipped, a sample that fails the range test or carries a zero attention weight
// reads nothing.  What a tile with a non-finite sum is redone with (round 5): a product multiplies the rows of EIGHT
// queries with a block-diagonal weight tile, so a non-finite value turns the zeros of the other queries' weights into NaN
// -- the reference keeps it with the queries that sample it.  A tile's sums are finite unless something non-finite was
// multiplied, and then every query of the tile is recomputed here: element for element the reference's result.
template <typename T>
__device__ __forceinline__ void exact8(const uint4 *ktab, __amdgpu_buffer_rsrc_t rsrc, uint32_t row_bytes,
                                       const uint16_t *loc_q, const uint16_t *attn_q, int K, uint32_t piece_off, float (&acc)[8])
{
    typedef Vec16<T> V;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int k = 0; k < K; ++k) {
        const uint4 ku = ktab[2 * k + 1];
        const int lstart = (int)ku.x, Wl = (int)ku.y, Hl = (int)ku.z;
        const float lx = to_f32(__builtin_bit_cast(T, loc_q[2 * k])), ly = to_f32(__builtin_bit_cast(T, loc_q[2 * k + 1]));
        const float a = to_f32(__builtin_bit_cast(T, attn_q[k]));
        const Tap<float> t = locate<float>(lx, ly, Hl, Wl, lstart);
        if (a == 0.f) continue;
        const float gy = 1.f - t.fy, gx = 1.f - t.fx;
        const float w[4] = {gy * gx * a, gy * t.fx * a, t.fy * gx * a, t.fy * t.fx * a};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (t.row[c] < 0) continue;
            const uint4 raw = buffer_load16(rsrc, (uint32_t)t.row[c] * row_bytes + piece_off);
            float v[8];
            V::unpack(raw, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fmaf(w[c], v[i], acc[i]);
        }
    }
}

}  // namespace q8

template <typename T>
__global__ void __launch_bounds__(kMmaThreads)
msda_fwd_q8(const T *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ start,
            const T *__restrict__ loc, const T *__restrict__ attn, T *__restrict__ out, const Dims d,
            const int q_per_run, const int lds_total, const int n_runs, const int n_slices)
{
    using namespace q8;
    typedef FwdMma<T> M;
    typedef Vec16<T> V;
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    int *tab = reinterpret_cast<int *>(smem);
    const uint4 *ktab = reinterpret_cast<const uint4 *>(smem + kTabBytes);

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int L = d.L, K = d.K;
    const int64_t HD = (int64_t)d.H * d.D;
    const uint32_t row_bytes = (uint32_t)(HD * sizeof(T));
    QPROF_DECL;
    build_table(tab, smem, shapes, start, L, K, d.P, tid, lds_total);

    // (LDS addresses as plain 32-bit numbers: "smem + x" costs an add per use, the dynamic LDS starts where it starts)
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_byte *)smem;
    const int wrec = kTabBytes + kKtabBytes + wave * kScratch;            // (byte offset of the wave's records in the workgroup's LDS)
    // the 16 zero bytes behind every record (A-operand lanes that hold nothing read them)
    if (lane < 8) *reinterpret_cast<uint4 *>(smem + wrec + lane * kRec + 288) = make_uint4(0u, 0u, 0u, 0u);
    // ---- lane roles
    const int sj = lane >> 3, si = lane & 7;                              // staging: query of the tile, sample of the pass
    // A operand: row m = 2 * query + (hi / lo), K block kb.  Query j's four corners sit in K block j / 2, positions
    // 4 * (j & 1) ..: the staging lane writes query j's 16-byte fragment as the lane with kb == m / 4 needs it (weights in
    // the lower or the upper half), every other lane of the product reads the record's 16 zero bytes: one read, no select
    const int am = lane & 15, akb = lane >> 4;
    const uint32_t a_rd = lds0 + wrec + ((akb == (am >> 2)) ? 32 + 128 * (am & 1) + 16 * (am >> 1) : 288);
    const int bkb = lane >> 4, br = (lane >> 2) & 3, bc = lane & 3;       // B operand: K block, corner, 8-byte piece
    const uint32_t b_rd = lds0 + wrec + 8 * bkb;                          // offsets of queries 2 kb, 2 kb + 1
    // (odd K blocks read the OTHER 32-byte piece of their rows: header)
    const uint32_t b_c0 = lds0 + (br & 1) * kRB + 8 * bc + ((bkb & 1) ? 32 : 0);
    const uint32_t b_hi = br >> 1;
    const int gj = lane >> 3, ghalf = (lane >> 2) & 1, gc = lane & 3;     // row gather: query, corner pair, 16-byte piece
    const uint32_t g_rd = lds0 + wrec + 16 * gj + 8 * ghalf;
    const int dn = lane & 15, drq = lane >> 4;                            // product: column, row quad (queries 2 drq, 2 drq + 1)
    const bool pair_ok = ((uintptr_t)loc & (2 * sizeof(T) - 1)) == 0;
    const int n_pass = (K + 7) / 8;
    const uint32_t q_stride = (uint32_t)d.H * (uint32_t)K;                // samples between consecutive queries of this head

    for (int run = blockIdx.x; run < n_runs; run += gridDim.x) {
    // run -> (b, query tile, slice, h), h fastest: a head's slab stays in one XCD's L2, and the slices of a (b, h, tile)
    // run next to each other (they read the same locations and weights)
    const int h = run % d.H;
    int t = run / d.H;
    const int sl = t % n_slices; t /= n_slices;
    const int q_run0 = (t % d.q_tiles) * q_per_run;
    const int b = t / d.q_tiles;
    const T *slab = value + ((int64_t)b * d.S) * HD + (int64_t)h * d.D + sl * kCS;
    const __amdgpu_buffer_rsrc_t rsrc =
        make_slab_rsrc(slab, ((int64_t)d.S * HD - (int64_t)h * d.D - sl * kCS) * (int64_t)sizeof(T));
    const int q_run1 = min(d.Nq, q_run0 + q_per_run);
    const int q_first = q_run0 + wave * kQT;
    const int n_tiles = q_first < q_run1 ? (q_run1 - q_first + kMmaWaves * kQT - 1) / (kMmaWaves * kQT) : 0;
    const uint16_t *loc_wg = reinterpret_cast<const uint16_t *>(loc) + 2 * (((int64_t)b * d.Nq * d.H + h) * K);
    const uint16_t *attn_wg = reinterpret_cast<const uint16_t *>(attn) + (((int64_t)b * d.Nq * d.H + h) * K);

    // (the next pass's sample words are requested while this pass multiplies; kept RAW until they are used)
    uint32_t pf_w0 = 0u, pf_w1 = 0u, pf_a = 0u;
    auto prefetch = [&](int tile, int pass) {
        const int q = q_first + tile * (kMmaWaves * kQT) + sj;
        const int k = pass * 8 + si;
        pf_w0 = pf_w1 = pf_a = 0u;
        if (tile < n_tiles && k < K && q < q_run1) {
            const uint32_t s = (uint32_t)q * q_stride + (uint32_t)k;
            const uint16_t *lw = loc_wg + 2 * (size_t)s;
            if (pair_ok) pf_w0 = *reinterpret_cast<const uint32_t *>(lw);
            else { pf_w0 = lw[0]; pf_w1 = lw[1]; }
            pf_a = attn_wg[s];
        }
    };
    prefetch(0, 0);
    if (run != (int)blockIdx.x) __syncthreads();                          // every wave is done with the previous image
    fill_image(tab, smem, rsrc, row_bytes, L, d.S, tid);
    QPROF(0);                                                             // table / barrier + image fill (incl. waiting for the slowest wave)

    for (int tile = 0; tile < n_tiles; ++tile) {
#ifndef Q8_NO_SETPRIO
        // (as msda_fwd_wq.hip, r05f: the wave that is behind the others of its SIMD -- more tiles left -- issues first, so that
        // the workgroup's waves reach the barrier in front of the next image fill together)
        { const int left = n_tiles - 1 - tile; if (left >= 3) __builtin_amdgcn_s_setprio(3); else if (left == 2) __builtin_amdgcn_s_setprio(2); else if (left == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#endif
        const int q0 = q_first + tile * (kMmaWaves * kQT);
        f32x4 acc[2];
        float accv[8];
        acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) accv[i] = 0.f;
        bool any_gather = false;
        for (int pass = 0; pass < n_pass; ++pass) {
#ifdef Q8_PROBE
        float pr_fy = 0.f, pr_gx = 0.f, pr_aa = 0.f, pr_w2 = 0.f;
#endif
        uint4 st_a, st_b;
        // ---- stage: one sample per lane (its words arrived during the previous pass); what the sample's level is
        // comes out of a per-sample-index table (made once per workgroup: no division, one 16-byte read)
        const int k = pass * 8 + si;
        const bool k_ok = k < K;
        const uint4 kt = ktab[2 * (k_ok ? k : 0)];
        const float Hf = __uint_as_float(kt.x), Wf = __uint_as_float(kt.y);
        const int ibase = (int)kt.z, lw_bytes = (int)kt.w;
        const bool resident = ibase >= 0;
        // K-blocks of the pass by kind (the same for every query: the level decides); lanes 0..7 are query 0's samples
        const uint32_t rmask = (uint32_t)__builtin_amdgcn_ballot_w64(k_ok && resident) & 0xffu;
        const uint32_t gmask = (uint32_t)__builtin_amdgcn_ballot_w64(k_ok && !resident) & 0xffu;
        wave_sync();                                                      // the previous pass's records are consumed
        {
            const int q = q0 + sj;
            asm volatile("" : "+v"(pf_w0), "+v"(pf_w1), "+v"(pf_a));      // (opaque HERE: the decode must not move up to the loads)
            const uint32_t xb = pair_ok ? (pf_w0 & 0xffffu) : pf_w0, yb = pair_ok ? (pf_w0 >> 16) : pf_w1;
            const float lx = to_f32(__builtin_bit_cast(T, (uint16_t)xb)), ly = to_f32(__builtin_bit_cast(T, (uint16_t)yb));
            const float a = to_f32(__builtin_bit_cast(T, (uint16_t)pf_a));
            const float y = ly * Hf - 0.5f, x = lx * Wf - 0.5f;
            // strict comparisons: NaN fails, exactly -1 / Hl / Wl fail (cuh:291)
            const bool inside = (y > -1.f) && (x > -1.f) && (y < Hf) && (x < Wf);
            const float yf = floorf(y), xf = floorf(x);
            // a zero attention weight (an image the token cannot see) reads nothing; nor does a query past the run's end
            const bool on = k_ok && q < q_run1 && inside && a != 0.f;
            const float fy = on ? y - yf : 0.f, fx = on ? x - xf : 0.f;
            const float gy = 1.f - fy, gx = 1.f - fx;
            const float aa = on ? a : 0.f;
            const float w[4] = {gy * gx * aa, gy * fx * aa, fy * gx * aa, fy * fx * aa};
            unsigned char *rec = smem + wrec + si * kRec;
            st_a = st_b = make_uint4(0u, 0u, 0u, 0u);                     // what this lane's 16-byte stores hand to the LDS
#ifdef Q8_PROBE
            pr_fy = fy; pr_gx = gx; pr_aa = aa;
            asm volatile("" : "+v"(pr_fy), "+v"(pr_gx), "+v"(pr_aa));
            pr_w2 = w[2];
#endif
            if (k_ok && resident) {
                // "off": the last pixel of line -1 -- its four "corners" are that zero pixel, the border pixel (0, -1)
                // behind it, the last (padding) pixel of line 0 and the border pixel (1, -1): zeros all
                const int rel = on ? ((int)yf + 1) * lw_bytes + ((int)xf + 1) * kRB : lw_bytes - kRB;
                uint32_t hi[4], lo[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) M::split(w[c], hi[c], lo[c]);
                *reinterpret_cast<uint32_t *>(rec + 4 * sj) = (uint32_t)(ibase + rel);
                const uint32_t h01 = hi[0] | (hi[1] << 16), h23 = hi[2] | (hi[3] << 16);
                const uint32_t l01 = lo[0] | (lo[1] << 16), l23 = lo[2] | (lo[3] << 16);
                const bool odd = sj & 1;                                  // K positions 4 .. 7 of the query pair's K block
                st_a = make_uint4(odd ? 0u : h01, odd ? 0u : h23, odd ? h01 : 0u, odd ? h23 : 0u);
                st_b = make_uint4(odd ? 0u : l01, odd ? 0u : l23, odd ? l01 : 0u, odd ? l23 : 0u);
                *reinterpret_cast<uint4 *>(rec + 32 + 16 * sj) = st_a;
                *reinterpret_cast<uint4 *>(rec + 160 + 16 * sj) = st_b;
            } else if (k_ok) {
                const uint4 ku = ktab[2 * k + 1];
                const int lstart = (int)ku.x, Wl = (int)ku.y, Hl = (int)ku.z;
                const int y0 = (int)yf, x0 = (int)xf;
                const bool top = y0 >= 0, left = x0 >= 0, bottom = y0 + 1 <= Hl - 1, right = x0 + 1 <= Wl - 1;
                const bool ok[4] = {on && top && left, on && top && right, on && bottom && left, on && bottom && right};
                const int p00 = lstart + y0 * Wl + x0;
                const int row[4] = {p00, p00 + 1, p00 + Wl, p00 + Wl + 1};
                uint32_t o[4], wb[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    o[c] = ok[c] ? (uint32_t)row[c] * row_bytes : kOobOffset;
                    wb[c] = ok[c] ? __float_as_uint(w[c]) : 0u;
                }
                // (stored by corner PAIR: a row-gather lane takes corners {half, 2 + half} as one 8-byte read)
                st_a = make_uint4(o[0], o[2], o[1], o[3]);
                st_b = make_uint4(wb[0], wb[2], wb[1], wb[3]);
                *reinterpret_cast<uint4 *>(rec + 16 * sj) = st_a;
                *reinterpret_cast<uint4 *>(rec + 128 + 16 * sj) = st_b;
            }
        }
#ifdef Q8_PROBE
        if (k_ok) {
            asm volatile("" : "+v"(pr_w2));
            const float chk = pr_fy * pr_gx * pr_aa;
            const unsigned char *rec = smem + wrec + si * kRec;
            uint32_t got, want;
            if (resident) {
                const uint4 r = *reinterpret_cast<const uint4 *>(rec + 32 + 16 * sj);
                got = ((sj & 1) ? r.w : r.y) & 0xffffu;
                uint32_t hi, lo;
                M::split(chk, hi, lo);
                want = hi;
            } else {
                const uint4 r = *reinterpret_cast<const uint4 *>(rec + 128 + 16 * sj);
                got = r.y;
                want = __float_as_uint(chk);
            }
            if (got != want) {
                const unsigned int i = atomicAdd(&g_q8_probe_n, 1u);
                if (i < 4096u) {
                    unsigned int *o = g_q8_probe + 8 * i;
                    o[0] = lane; o[1] = pass | (tile << 8) | (wave << 16) | ((resident ? 1u : 0u) << 24); o[2] = want; o[3] = got;
                    o[4] = __float_as_uint(pr_w2); o[5] = __float_as_uint(pr_fy); o[6] = __float_as_uint(pr_gx); o[7] = __float_as_uint(pr_aa);
                }
            }
        }
#endif
        wave_sync();
        if (pass + 1 < n_pass) prefetch(tile, pass + 1); else prefetch(tile + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        QPROF(1);                                                         // stage

        // ---- resident K-blocks: 32 rows out of the image and two products (2 x 16 channels) each.  N K-blocks as ONE
        // straight-line sequence -- all record reads, then all transposing reads, then the products (two chains) -- so
        // that their LDS round trips overlap (a loop over single K-blocks was two dependent round trips + a product
        // each: r04d); every record address is a lane constant + an immediate
        auto products = [&](auto n_tag, auto ib_tag) {
            constexpr int N = decltype(n_tag)::value, IB = decltype(ib_tag)::value;
            u32v4 a4[N];
            u32v2 o2[N];
#pragma unroll
            for (int u = 0; u < N; ++u) {
                a4[u] = *reinterpret_cast<const lds_u32v4 *>((uintptr_t)(a_rd + (IB + u) * kRec));
                o2[u] = *reinterpret_cast<const lds_u32v2 *>((uintptr_t)(b_rd + (IB + u) * kRec));     // queries 2 kb, 2 kb + 1
            }
            s16x8 Bv[N][2];
#pragma unroll
            for (int u = 0; u < N; ++u) {
                const uint32_t bd = b_c0 + __umul24(b_hi, (uint32_t)__builtin_amdgcn_readlane(lw_bytes, IB + u));
                const uint32_t p0 = o2[u].x + bd, p1 = o2[u].y + bd;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(uintptr_t)(p0 ^ (32u * g)));
                    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(uintptr_t)(p1 ^ (32u * g)));
                    Bv[u][g][0] = v0[0]; Bv[u][g][1] = v0[1]; Bv[u][g][2] = v0[2]; Bv[u][g][3] = v0[3];
                    Bv[u][g][4] = v1[0]; Bv[u][g][5] = v1[1]; Bv[u][g][6] = v1[2]; Bv[u][g][7] = v1[3];
                }
            }
#pragma unroll
            for (int u = 0; u < N; ++u) {
                const s16x8 A = __builtin_bit_cast(s16x8, a4[u]);
#pragma unroll
                for (int g = 0; g < 2; ++g) acc[g] = M::run(A, Bv[u][g], acc[g]);
            }
        };
        using N1 = std::integral_constant<int, 1>;
        using N2 = std::integral_constant<int, 2>;
        // the K-blocks of a nibble of the pass: all four resident (the usual case: P is a multiple of 4) -> one sequence
        auto nibble = [&](uint32_t rm, auto ib_tag) {
            constexpr int IB = decltype(ib_tag)::value;
            const uint32_t m4 = (rm >> IB) & 0xfu;
            if (m4 == 0xfu) {
                products(N2{}, std::integral_constant<int, IB>{});
                products(N2{}, std::integral_constant<int, IB + 2>{});
            } else
            if (m4) {
                if (m4 & 1u) products(N1{}, std::integral_constant<int, IB>{});
                if (m4 & 2u) products(N1{}, std::integral_constant<int, IB + 1>{});
                if (m4 & 4u) products(N1{}, std::integral_constant<int, IB + 2>{});
                if (m4 & 8u) products(N1{}, std::integral_constant<int, IB + 3>{});
            }
        };
        // ---- a row-gather K-block: 2 x 16 rows of 64 bytes
        auto issue = [&](int i, uint4 (&raw)[2]) {
            const u32v2 go = *reinterpret_cast<const lds_u32v2 *>((uintptr_t)(g_rd + i * kRec));
            raw[0] = buffer_load16(rsrc, go.x + (uint32_t)(16 * gc));
            raw[1] = buffer_load16(rsrc, go.y + (uint32_t)(16 * gc));
        };
        auto consume = [&](int i, const uint4 (&raw)[2]) {
            const u32v2 gw = *reinterpret_cast<const lds_u32v2 *>((uintptr_t)(g_rd + 128 + i * kRec));
            const float w2[2] = {__uint_as_float(gw.x), __uint_as_float(gw.y)};
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float v[8];
                V::unpack(raw[c], v);
#pragma unroll
                for (int e = 0; e < 8; ++e) accv[e] = fmaf(w2[c], v[e], accv[e]);
            }
        };
        {
            uint32_t gm = (uint32_t)__builtin_amdgcn_readfirstlane((int)gmask);
            const uint32_t rm = (uint32_t)__builtin_amdgcn_readfirstlane((int)rmask);
            any_gather = any_gather || gm != 0u;
            // two row-gather K-blocks in flight while the resident ones multiply (four: the kernel spills at its 128 registers)
            uint4 r0[2], r1[2];
            int i0 = -1, i1 = -1;
            auto take = [&]() { const int i = __builtin_ctz(gm); gm &= gm - 1u; return i; };
            if (gm) { i0 = take(); issue(i0, r0); }
            if (gm) { i1 = take(); issue(i1, r1); }
            __builtin_amdgcn_sched_barrier(0);
            QPROF(2);                                                     // first row requests
            nibble(rm, std::integral_constant<int, 0>{});
            nibble(rm, std::integral_constant<int, 4>{});
            __builtin_amdgcn_sched_barrier(0);
            QPROF(3);                                                     // resident K-blocks
            QPROF_COUNT(6, __builtin_popcount(rm));
            while (i0 >= 0) {
                consume(i0, r0); i0 = -1;
                if (gm) { i0 = take(); issue(i0, r0); }
                if (i1 >= 0) { consume(i1, r1); i1 = -1; if (gm) { i1 = take(); issue(i1, r1); } }
            }
            QPROF(4);                                                     // row gather
        }
        }   // passes

        // ---- epilogue: hi + lo rows (rows 4 rq .. 4 rq + 3 = queries 2 rq, 2 rq + 1); odd row quads multiplied the
        // pieces in the other order
        wave_sync();                                                      // the records are consumed
        float *es = reinterpret_cast<float *>(smem + wrec);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int ch = 16 * ((drq & 1) ? (g ^ 1) : g) + dn;
            es[(2 * drq) * kCS + ch] = acc[g][0] + acc[g][1];
            es[(2 * drq + 1) * kCS + ch] = acc[g][2] + acc[g][3];
        }
        if (any_gather) {
            float4 *pv = reinterpret_cast<float4 *>(smem + wrec + 1024 + ghalf * 1024 + gj * 128 + gc * 32);
            pv[0] = make_float4(accv[0], accv[1], accv[2], accv[3]);
            pv[1] = make_float4(accv[4], accv[5], accv[6], accv[7]);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 15" ::: "memory");                        // (16-byte LDS stores and their registers: see the staging)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 8; ++e) asm volatile("" :: "v"(accv[e]));
        }
        wave_sync();
        const int q = q0 + (lane >> 2);
        if (lane < 32 && q < q_run1) {
            const float4 *ps = reinterpret_cast<const float4 *>(smem + wrec + (lane >> 2) * 128 + (lane & 3) * 32);
            float4 s0 = ps[0], s1 = ps[1];
            if (any_gather) {
                const float4 a0 = ps[64], a1 = ps[65], b0 = ps[128], b1 = ps[129];       // (+ 1024, + 2048 bytes)
                s0.x += a0.x + b0.x; s0.y += a0.y + b0.y; s0.z += a0.z + b0.z; s0.w += a0.w + b0.w;
                s1.x += a1.x + b1.x; s1.y += a1.y + b1.y; s1.z += a1.z + b1.z; s1.w += a1.w + b1.w;
            }
            float v[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            // (a non-finite sum anywhere in the tile: every query of it channel by channel, see q8::exact8)
            float nf = v[0] * 0.f;
#pragma unroll
            for (int e = 1; e < 8; ++e) nf = fmaf(v[e], 0.f, nf);
            if (__builtin_amdgcn_ballot_w64(nf != nf) != 0ull) {
                const uint32_t s0q = (uint32_t)q * q_stride;
                exact8<T>(ktab, rsrc, row_bytes, loc_wg + 2 * (size_t)s0q, attn_wg + (size_t)s0q, K, (uint32_t)(lane & 3) * 16u, v);
            }
            T *o = out + (((int64_t)b * d.Nq + q) * d.H + h) * d.D + sl * kCS + (lane & 3) * 8;
            store16_stream(o, V::pack(v));
        }
        wave_sync();
        // (the sums lay over the records' zero bytes)
        if (lane < 8) *reinterpret_cast<uint4 *>(smem + wrec + lane * kRec + 288) = make_uint4(0u, 0u, 0u, 0u);
        QPROF(5);                                                         // epilogue
        QPROF_COUNT(7, n_pass);
    }   // tiles
    }   // runs
    QPROF_FLUSH();
}

// The host sees no level table (device pointers only): "every level's slice fits in the image" is judged from S -- a
// pyramid of S pixels takes S * 64 bytes plus its zero borders (32x32 + 16x16 + 8x8: 86 KB of pixels, 101 KB with them).
static bool q8_all_resident_likely(const Dims &d)
{
    return (int64_t)d.S * q8::kRB * 5 / 4 <= (int64_t)(kLdsTotal - q8::kImg0);
}

// ---------------------------------------------------------------- launcher
template <typename T>
static hipError_t launch_q8(const void *value, const int64_t *shapes, const int64_t *start,
                            const void *loc, const void *attn, void *out, Dims d, hipStream_t st)
{
    using namespace q8;
    static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fwd_q8<T>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal);
    if (once != hipSuccess) return once;
    const int env_kb = knob_int(K_FWD_Q8_LDS_KB, 0);      // tuning / tests
    const int lds_total = env_kb > 0 ? std::min(kLdsTotal, std::max(kImg0 + 1024, env_kb * 1024)) : kLdsTotal;
    // Queries per run: the image fill (one pass over the resident levels' slices + the barrier around it: 14-19 k clocks
    // per wave, a quarter of a 512-query run at the LLM geometry, r04f) is paid per run, so runs are as long as the shape
    // allows while every CU still gets one: up to 1024 queries (longer runs put more (b, h) slabs of the row-gather
    // levels into an XCD's L2 at a time: r03bf).
    const int unit = kMmaWaves * kQT;
    const int n_slices = d.D / kCS;
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        return n > 0 ? n : 256;
    }();
    // (shapes whose levels do not all fit keep 512: at the LLM geometry with 4 images 1024 cost 258 us against 234)
    int q_per_run = q8_all_resident_likely(d) ? 1024 : 512;
    const int64_t units = (int64_t)d.B * d.H * n_slices;
    while (q_per_run > unit && units * ((d.Nq + q_per_run - 1) / q_per_run) < cus) q_per_run -= unit;
    const int env_q = knob_int(K_FWD_Q8_QPR, 0);
    if (env_q > 0) q_per_run = env_q;
    q_per_run = std::max(unit, (q_per_run + unit - 1) / unit * unit);
    d.q_tiles = (d.Nq + q_per_run - 1) / q_per_run;
    const int64_t runs = (int64_t)d.B * d.q_tiles * n_slices * d.H;
    if (runs > 0x7fffffffLL) return hipErrorInvalidValue;
    const int grid = (int)persistent_grid(runs, d.H);
    hipLaunchKernelGGL((msda_fwd_q8<T>), dim3((unsigned)grid), dim3(kMmaThreads), lds_total, st,
                       (const T *)value, shapes, start, (const T *)loc, (const T *)attn, (T *)out, d, q_per_run,
                       lds_total, (int)runs, n_slices);
    return hipGetLastError();
}

bool fwd_q8_supported(int dtype, const Dims &d)
{
    if (dtype != 1 && dtype != 2) return false;
    if (d.D <= 0 || d.D % q8::kCS) return false;
    if (d.L > kMmaMaxLevels || d.K <= 0 || d.K > q8::kMaxK || d.P <= 0) return false;
    if ((int64_t)d.Nq * d.H * d.K >= (1LL << 30)) return false;            // 32-bit sample offsets inside a (b, h) slab
    return (int64_t)d.S * d.H * d.D * 2 <= kMaxSlabBytes;
}

bool fwd_q8_applies(int dtype, const Dims &d)
{
    const char *algo = knob_str(K_FWD_ALGO);                           // "q8": whenever the shape allows
    if (!fwd_q8_supported(dtype, d)) return false;
    if (algo && algo[0] == 'q') return true;
    if (algo) return false;
    // Default where it measured faster than the row gather (profiles/r04_experiments.md r04n): heads of 32 / 64 channels
    // whose whole pyramid is resident -- the LLM layer with one image per sequence (61 -> 53 us at cfg3's op), the
    // ViT-Adapter's injector (42 -> 35 us).  With a row-gather level left (the image decoder's 64x64: 335 vs 339 us; the
    // LLM's 4 images: 234 vs 222) it does not pay, and heads of 128 channels pay the per-sample arithmetic four times
    // (north star: 196 us against msda_fwd_mma's 128).
    // Round 5 (r05ae, tools/fwd_nq.py): and only launches with enough queries per CU -- a run is one 1024-lane workgroup and an
    // image fill.  At the LLM layer's geometry (64 slabs x 2 slices) 2048 queries: 49.6 us against the row gather's 63.2, but
    // 1024 queries: 41.5 against 31.2, 512: 23.0 against 18.5 (BASELINE config 3 runs 128 / 512 / 2048 tokens); the injector
    // (512 slabs of 256 queries, one slice): 33 against 39, at 128 queries 26.7 against 25.3.
    const int64_t per_cu = (int64_t)d.B * d.H * (d.D / q8::kCS) * d.Nq / device_cus();
    if (per_cu < (d.D <= 32 ? 384 : 768)) return false;
    return d.D <= 64 && q8_all_resident_likely(d) && d.S >= 512 && d.Nq >= 128 && (int64_t)d.Nq * d.K >= 3072;      // (the injector: 256 queries x 12 samples per image fill)
}

hipError_t forward_q8(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                      const void *loc, const void *attn, void *out, const Dims &d, hipStream_t st)
{
    if (dtype == 1) return launch_q8<half_t>(value, shapes, start, loc, attn, out, d, st);
    return launch_q8<bf16_t>(value, shapes, start, loc, attn, out, d, st);
}

}  // namespace mmfs

#ifdef Q8_PROBE
extern "C" int mmfs_debug_q8_probe(unsigned int *out, unsigned int *n, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(n, HIP_SYMBOL(mmfs::g_q8_probe_n), sizeof(unsigned int));
    if (e == hipSuccess) e = hipMemcpyFromSymbol(out, HIP_SYMBOL(mmfs::g_q8_probe), 4096 * 8 * sizeof(unsigned int));
    if (e == hipSuccess && reset) { const unsigned int z = 0; e = hipMemcpyToSymbol(HIP_SYMBOL(mmfs::g_q8_probe_n), &z, sizeof(z)); }
    return (int)e;
}
#endif

#ifdef MMFS_PROFILE_Q8
extern "C" int mmfs_debug_q8_profile(unsigned long long *out, int reset)
{
    static unsigned long long host[mmfs::kQProfSlots * 8];
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(mmfs::g_q8_prof), sizeof(host));
    for (int i = 0; i < 8; ++i) out[i] = 0;
    for (int s = 0; s < mmfs::kQProfSlots; ++s)
        for (int i = 0; i < 8; ++i) out[i] += host[s * 8 + i];
    if (e == hipSuccess && reset) {
        for (auto &v : host) v = 0;
        e = hipMemcpyToSymbol(HIP_SYMBOL(mmfs::g_q8_prof), host, sizeof(host));
    }
    return (int)e;
}
#endif

