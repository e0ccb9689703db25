// msda_gv_mma.hip -- grad_value of the small levels of the pyramid, sorted and reduced INSIDE a workgroup.
//
// Replaces, for the levels the host plan selects, the reference's per-sample float atomics
//   mm_interleaved/models/utils/ops/src/cuda/ms_deform_im2col_cuda.cuh:128-155 (fp32 accumulation, one rounding at
//   the end: src/cuda/ms_deform_attn_cuda.cu:122-165)
// and this repository's own cell sort + 4x4-block reduce (msda_bwd_block.hip, msda_bwd_tile.hip), which stay in
// charge of the large levels.  The organisation is described in msda_gv_mma.h; the arithmetic is the tile
// reduce's (same products, fp32 sums in another order):
//
//   per chunk of QC queries of one (b, h):
//     rows    grad_out rows global -> LDS by DMA, 16-byte chunks XOR-swizzled by the row index (applied to the
//             DMA's SOURCE address) so that a transposing read of rows with different low index bits is conflict-free;
//     bin     every lane decodes up to 2 samples of the group's levels, finds the 1, 2 or 4 blocks whose pixels the
//             sample's corners touch, and takes a rank in each block's list (LDS atomics, one counter per list and
//             CLASS of query = index modulo 8); prefix over the <= 64 lists; the lanes then write their 4-byte
//             records {sample | query << 16}, a list's classes merged round-robin: eight consecutive records have
//             rows that a transposing read can fetch together without a bank conflict;
//     product every wave walks the lists of the virtual blocks it owns, 32 records per step: the A operand
//             (16 pixels x 32 records, hi and lo parts) is built in a wave-private LDS tile by 64 lanes = 16 records
//             x 4 corners, the B operand (32 records x 16 channels) is read transposed from the records' rows;
//   at the end the 2^k virtual blocks of a block are added up through LDS; a group cut into several query ranges
//   leaves fp32 partial tiles and the range that finishes last adds them (written through, read past the caches, a
//   drained arrival counter).
//
// Semantics inherited from the matrix-core formulation (as msda_bwd_tile.hip): a non-finite grad_out element
// reaches all 16 pixels of the blocks its sample touches; samples of zero attention weight are not visited.
#include "msda_gv_mma.h"
#include "msda_env.h"
#include "msda_mma_common.h"
#include "msda_launch.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace mmfs {
namespace gv {

using mma::f32x4;
using mma::s16x4;
using mma::s16x8;
using mma::lds_s16x4;
using mma::wave_sync;

namespace {

// Development aid (tools/exp_build.sh gprof "-DMMFS_PROFILE_GV"; tools/gv_prof.py): shader clocks per phase,
// thread 0 of each workgroup, summed per slot.
#ifdef MMFS_PROFILE_GV
}  // namespace
constexpr int kGProfSlots = 2048;
__device__ unsigned long long g_gv_prof[kGProfSlots * 8];
namespace {
#define GPROF_DECL unsigned long long gprof_c = __builtin_readcyclecounter(), gprof_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define GPROF(i) do { const unsigned long long tn = __builtin_readcyclecounter(); gprof_t[i] += tn - gprof_c; gprof_c = tn; } while (0)
#define GPROF_COUNT(i, v) do { gprof_t[i] += (unsigned long long)(v); } while (0)
#define GPROF_FLUSH() do { if (threadIdx.x == 0) for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_gv_prof[(blockIdx.x % kGProfSlots) * 8 + i_], gprof_t[i_]); } while (0)
#else
#define GPROF_DECL do {} while (0)
#define GPROF(i) do {} while (0)
#define GPROF_COUNT(i, v) do {} while (0)
#define GPROF_FLUSH() do {} while (0)
#endif

typedef __attribute__((address_space(3))) void lds_void_t;

// 32-byte column tile n of row r sits at tile position n ^ swz(r): eight rows with different r % 8 read together
// (one phase of a transposing read) touch eight different 32-byte bank slots
template <int D> __device__ __host__ __forceinline__ int swz(int r) { return D >= 128 ? (r & 7) : ((r >> 1) & 3); }

// a sample's geometry on its level: the reference's expressions (cuh:288-291; strict comparisons, NaN fails)
struct Geo { int y0, x0; float fy, fx, a; bool live; };
template <typename T> __device__ __forceinline__ Geo decode(uint32_t locw, uint32_t aw, int Hl, int Wl)
{
    Geo g;
    const float lx = to_f32(__builtin_bit_cast(T, (uint16_t)(locw & 0xffffu))), ly = to_f32(__builtin_bit_cast(T, (uint16_t)(locw >> 16)));
    g.a = to_f32(__builtin_bit_cast(T, (uint16_t)(aw & 0xffffu)));
    const float y = ly * (float)Hl - 0.5f, x = lx * (float)Wl - 0.5f;
    const bool inside = (y > -1.f) && (x > -1.f) && (y < (float)Hl) && (x < (float)Wl);
    const float yf = floorf(y), xf = floorf(x);
    g.y0 = inside ? (int)yf : 0; g.x0 = inside ? (int)xf : 0;
    g.fy = y - yf; g.fx = x - xf;
    g.live = inside && g.a != 0.f;                 // (a zero weight adds nothing: msda_bwd_block.hip, cell_in_tile)
    return g;
}

// Place p of a step's 32 records <-> column k of the product: bits 2 and 3 swapped (an involution).  One phase of a
// transposing read serves the lanes of two 16-lane groups, columns 8 (2j + u) + 4 t + e for u = 0, 1, e = 0..3:
// with the swap those are the EIGHT CONSECUTIVE places 16 j + 8 t .. + 7 of the list -- eight different classes.
__device__ __host__ __forceinline__ int perm(int p) { return (p & 0x13) | ((p & 4) << 1) | ((p & 8) >> 1); }

// segment table in LDS: 8 ints per segment
enum { kSegH = 0, kSegW = 1, kSegNbx = 2, kSegV0 = 3, kSegLog2s = 4, kSegLevel = 5, kSegStart = 6, kSegRb0 = 7 };

template <typename T, int D>
__global__ void __launch_bounds__(kThreads)
msda_gv_mma(const T *__restrict__ grad_out, const T *__restrict__ loc, const T *__restrict__ attn,
            T *__restrict__ grad_value, float *__restrict__ partials, uint32_t *__restrict__ arrive,
            const Dims d, const Table tab)
{
    typedef Geom<D> G;
    typedef mma::FwdMma<T> M;
    typedef Vec16<T> V;
    constexpr int RB = G::RB, LPR = G::LPR, NT = G::NT, SLOTS = G::SLOTS;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    GPROF_DECL;

    // ---- which slab, group and query range (h fastest: a head's slab stays in one XCD's L2)
    const int h = blockIdx.x % d.H;
    const int tw = blockIdx.x / d.H;
    const int u = tw % tab.wgs_per_slab;
    const int b = tw / tab.wgs_per_slab;
    int g = 0;
    while (g + 1 < tab.n_groups && (int)tab.g[g + 1].wg0 <= u) ++g;
    const int part = u - (int)tab.g[g].wg0, qparts = tab.g[g].qparts;
    const int QC = tab.g[g].qc, nvb = tab.g[g].nvb, nseg = tab.g[g].nseg, nrb = tab.g[g].nrb;
    const int NLP = nseg * d.P;                       // samples of the group's levels per query
    const int NS = QC * NLP;                          // samples of a chunk
    const int nchunks = (d.Nq + QC - 1) / QC;
    const int c0 = (int)((int64_t)part * nchunks / qparts), c1 = (int)((int64_t)(part + 1) * nchunks / qparts);

    // ---- LDS
    uint32_t *cnt2 = reinterpret_cast<uint32_t *>(smem);          // [2][64][8] records per (virtual block, query class): one set per chunk parity
    uint32_t *lbase = cnt2 + 2 * kMaxVb * 8;                       // [64] first record of each list
    uint32_t *nrec = lbase + kMaxVb;                               // [64] records of each list
    int *segtab = reinterpret_cast<int *>(nrec + kMaxVb);          // [kMaxSegs][8]
    int *vbd = segtab + kMaxSegs * 8;                              // [64][4] segment, block row, block column, part of the block
    int *flag = vbd + kMaxVb * 4;                                  // [1]
    unsigned char *atile = smem + kCtrl + wave * kATile;
    unsigned char *rows = smem + kRows0;
    const int QZ = rows_alloc(QC, LPR);                            // the row of zeros
    unsigned char *samp = rows + (QZ + 1) * RB;                    // [NS] {loc word, attention bits}
    uint32_t *recs = reinterpret_cast<uint32_t *>(samp + ((NS * 8 + 15) & ~15));      // [<= 4 * NS] {sample | query << 16}

    if (tid < nseg) {
        const Seg sg = tab.g[g].seg[tid];
        const Level lv = tab.lv[sg.lslot];
        int *st = segtab + tid * 8;
        st[kSegH] = lv.Hl; st[kSegW] = lv.Wl; st[kSegNbx] = lv.nbx; st[kSegV0] = sg.v0; st[kSegLog2s] = sg.log2s;
        st[kSegLevel] = lv.level; st[kSegStart] = lv.lstart; st[kSegRb0] = sg.rb0;
    }
    cnt2[tid] = 0u;                                                // (2 x 64 x 8 = one word per lane)
    if (tid < RB / 4) reinterpret_cast<uint32_t *>(rows + QZ * RB)[tid] = 0u;
    __syncthreads();
    if (tid < nvb) {
        int s = 0;
        while (s + 1 < nseg && segtab[(s + 1) * 8 + kSegV0] <= tid) ++s;
        const int rel = tid - segtab[s * 8 + kSegV0], l2 = segtab[s * 8 + kSegLog2s], nbx = segtab[s * 8 + kSegNbx];
        const int rb = rel >> l2, by = rb / nbx;
        vbd[tid * 4] = s; vbd[tid * 4 + 1] = by; vbd[tid * 4 + 2] = rb - by * nbx; vbd[tid * 4 + 3] = rel & ((1 << l2) - 1);
    }

    // ---- accumulators: SLOTS virtual blocks per wave, 16 pixels x D channels each
    f32x4 acc[SLOTS][NT];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[s][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const uint32_t HDB = (uint32_t)(d.H * d.D) * (uint32_t)sizeof(T);                // bytes between consecutive queries
    const T *gslice = grad_out + ((int64_t)b * d.Nq * d.H + h) * d.D;
    const __amdgpu_buffer_rsrc_t rsrc = make_slab_rsrc(gslice, ((int64_t)d.Nq * d.H * d.D - (int64_t)h * d.D) * (int64_t)sizeof(T));
    const uint16_t *loc_wg = reinterpret_cast<const uint16_t *>(loc) + 2 * (((int64_t)b * d.Nq * d.H + h) * d.K);
    const uint16_t *attn_wg = reinterpret_cast<const uint16_t *>(attn) + (((int64_t)b * d.Nq * d.H + h) * d.K);
    const uint32_t q_stride = (uint32_t)d.H * (uint32_t)d.K;
    const bool pair_ok = ((uintptr_t)loc & 3) == 0;
    const float inv_nlp = 1.0f / (float)NLP, inv_p = 1.0f / (float)d.P;
    constexpr int KS = kMaxSamples / kThreads;                   // samples per lane
    // product roles
    const int am = lane & 15, akc = lane >> 4;                   // A operand: pixel, block of 8 records
    const int a_off = am * 64 + (((akc ^ (am >> 2)) & 3) << 4);
    const int bG = lane >> 4, be = (lane >> 2) & 3, bc = lane & 3;      // B operand: block of 8 records, row of the read, 8-byte piece
    const int wr = lane >> 2, wcy = (lane >> 1) & 1, wcx = lane & 1;     // weight tile: record of the pass, corner
    GPROF(0);

    for (int c = c0; c < c1; ++c) {
        const int q0 = c * QC;
        __syncthreads();                                         // the previous chunk's rows, samples and records are consumed
        // ---- rows: DMA, LDS image lane-linear, the swizzle on the source side
        for (int it = 0; it * kThreads < QC * LPR; ++it) {
            const int i = it * kThreads + tid;
            const int r = i / LPR, cpos = i % LPR;
            const uint32_t off = (r < QC && q0 + r < d.Nq)
                ? (uint32_t)(q0 + r) * HDB + (uint32_t)((cpos ^ (2 * swz<D>(r))) * 16) : kOobOffset;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t *)(rows + (it * kThreads + wave * 64) * 16), 16, (int)off, 0, 0, 0);
        }
        // ---- samples: decode, count.  A list is kept by CLASS of query (index modulo 8: the swizzle of its row)
        uint32_t *cnt = cnt2 + (c & 1) * (kMaxVb * 8);
        uint32_t key[KS][4];
#pragma unroll
        for (int k = 0; k < KS; ++k) {
#pragma unroll
            for (int j = 0; j < 4; ++j) key[k][j] = 0xffffffffu;
            const int sidx = k * kThreads + tid;
            if (sidx >= NS) continue;
            const int ql = (int)(((float)sidx + 0.5f) * inv_nlp);
            const int rem = sidx - ql * NLP;
            const int ls = (int)(((float)rem + 0.5f) * inv_p);
            const int p = rem - ls * d.P;
            const int *st = segtab + ls * 8;
            uint32_t locw = 0u, aw = 0u;
            if (q0 + ql < d.Nq) {
                const uint32_t s = (uint32_t)(q0 + ql) * q_stride + (uint32_t)(st[kSegLevel] * d.P + p);
                const uint16_t *lw = loc_wg + 2 * (size_t)s;
                locw = pair_ok ? *reinterpret_cast<const uint32_t *>(lw) : ((uint32_t)lw[0] | ((uint32_t)lw[1] << 16));
                aw = attn_wg[s];
            }
            reinterpret_cast<uint2 *>(samp)[sidx] = make_uint2(locw, aw);
            const int Hl = st[kSegH], Wl = st[kSegW];
            const Geo ge = decode<T>(locw, aw, Hl, Wl);
            if (!ge.live) continue;
            // the blocks whose pixels the footprint's corners inside the map touch
            const int ya = max(ge.y0, 0) >> 2, yb = min(ge.y0 + 1, Hl - 1) >> 2;
            const int xa = max(ge.x0, 0) >> 2, xb = min(ge.x0 + 1, Wl - 1) >> 2;
            const int nbx = st[kSegNbx], l2 = st[kSegLog2s];
            const int v00 = st[kSegV0] + (sidx & ((1 << l2) - 1));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int by = (j >> 1) ? yb : ya, bx = (j & 1) ? xb : xa;
                if (((j >> 1) && yb == ya) || ((j & 1) && xb == xa)) continue;
                const uint32_t v = (uint32_t)(v00 + ((by * nbx + bx) << l2));
                const uint32_t cl = (uint32_t)ql & 7u;
                key[k][j] = v | (cl << 6) | (atomicAdd(&cnt[v * 8 + cl], 1u) << 9);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the rows have landed)
        __syncthreads();
        GPROF(1);
        // ---- prefix over the lists
        if (wave == 0) {
            uint32_t n = 0u;
            if (lane < nvb) {
                const uint4 a0 = reinterpret_cast<const uint4 *>(cnt)[lane * 2], a1 = reinterpret_cast<const uint4 *>(cnt)[lane * 2 + 1];
                n = a0.x + a0.y + a0.z + a0.w + a1.x + a1.y + a1.z + a1.w;
            }
            uint32_t incl = n;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_up(incl, o, 64);
                if (lane >= o) incl += t;
            }
            lbase[lane] = incl - n; nrec[lane] = n;
        }
        __syncthreads();
        // ---- place: the classes of a list are merged round-robin -- record r of class c goes behind the records
        // (r', c') with r' < r, or r' == r and c' < c -- so that eight consecutive records are of eight different
        // classes for as long as every class has records left
        cnt2[((c + 1) & 1) * (kMaxVb * 8) + (tid & (kMaxVb * 8 - 1))] = 0u;      // (the other parity's counters: last read a chunk ago)
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int sidx = k * kThreads + tid;
            const int ql = (int)(((float)sidx + 0.5f) * inv_nlp);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (key[k][j] == 0xffffffffu) continue;
                const uint32_t v = key[k][j] & 63u, cl = (key[k][j] >> 6) & 7u, r = key[k][j] >> 9;
                const uint4 a0 = reinterpret_cast<const uint4 *>(cnt)[v * 2], a1 = reinterpret_cast<const uint4 *>(cnt)[v * 2 + 1];
                const uint32_t cc[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                uint32_t pos = lbase[v];
#pragma unroll
                for (uint32_t c2 = 0; c2 < 8; ++c2) pos += min(cc[c2], r + (c2 < cl ? 1u : 0u));
                recs[pos] = (uint32_t)sidx | ((uint32_t)ql << 16);
            }
        }
        __syncthreads();
        GPROF(2);

        // ---- products: every wave on its own
        auto walk = [&](auto slot_c) {
            constexpr int SL = decltype(slot_c)::value;
            const int v = wave + kWaves * SL;
            if (v >= nvb) return;
            const int n = __builtin_amdgcn_readfirstlane((int)nrec[v]);
            if (n == 0) return;
            const uint32_t *list = recs + __builtin_amdgcn_readfirstlane((int)lbase[v]);
            const int ls = __builtin_amdgcn_readfirstlane(vbd[v * 4]);
            const int by4 = kTB * __builtin_amdgcn_readfirstlane(vbd[v * 4 + 1]), bx4 = kTB * __builtin_amdgcn_readfirstlane(vbd[v * 4 + 2]);
            const int Hl = __builtin_amdgcn_readfirstlane(segtab[ls * 8 + kSegH]), Wl = __builtin_amdgcn_readfirstlane(segtab[ls * 8 + kSegW]);
            for (int p0 = 0; p0 < n; p0 += 32) {
                const int cs = min(32, n - p0);
                // weight tile: zero, then one weight per lane and pass
                reinterpret_cast<uint4 *>(atile)[lane] = make_uint4(0u, 0u, 0u, 0u);
                reinterpret_cast<uint4 *>(atile)[64 + lane] = make_uint4(0u, 0u, 0u, 0u);
                wave_sync();
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int pl = 16 * t + wr;                       // place in the step's 32 records
                    const int k = perm(pl);                            // its column of the product
                    if (pl < cs) {
                        const uint32_t rec = list[p0 + pl];
                        const uint2 sw = reinterpret_cast<const uint2 *>(samp)[rec & 0xffffu];
                        const Geo ge = decode<T>(sw.x, sw.y, Hl, Wl);
                        const int yy = ge.y0 + wcy, xx = ge.x0 + wcx;
                        const int py = yy - by4, px = xx - bx4;
                        if ((unsigned)py < (unsigned)kTB && (unsigned)px < (unsigned)kTB && yy < Hl && xx < Wl) {
                            const float wgt = (wcy ? ge.fy : 1.f - ge.fy) * (wcx ? ge.fx : 1.f - ge.fx) * ge.a;
                            uint32_t hi, lo;
                            M::split(wgt, hi, lo);
                            const int m = py * kTB + px;
                            const int o = m * 64 + ((((k >> 3) ^ (m >> 2)) & 3) << 4) + ((k & 7) << 1);
                            *reinterpret_cast<uint16_t *>(atile + o) = (uint16_t)hi;
                            *reinterpret_cast<uint16_t *>(atile + 1024 + o) = (uint16_t)lo;
                        }
                    }
                }
                wave_sync();
                const s16x8 Ah = *reinterpret_cast<const s16x8 *>(atile + a_off);
                const s16x8 Al = *reinterpret_cast<const s16x8 *>(atile + 1024 + a_off);
                // the records' rows: this lane supplies 8 bytes of rows 8 bG + be and 8 bG + 4 + be
                const unsigned char *ba[2];
                int xs[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int pl = perm(8 * bG + 4 * t + be);
                    const int q = pl < cs ? (int)(list[p0 + pl] >> 16) : QZ;
                    ba[t] = rows + q * RB + 8 * bc;
                    xs[t] = swz<D>(q) << 5;
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    s16x8 Bv;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const s16x4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(ba[t] + ((nt << 5) ^ xs[t])));
                        Bv[4 * t] = x[0]; Bv[4 * t + 1] = x[1]; Bv[4 * t + 2] = x[2]; Bv[4 * t + 3] = x[3];
                    }
                    acc[SL][nt] = M::run(Ah, Bv, acc[SL][nt]);
                    acc[SL][nt] = M::run(Al, Bv, acc[SL][nt]);
                }
                wave_sync();
            }
        };
        walk(std::integral_constant<int, 0>{});
        walk(std::integral_constant<int, 1>{});
        if constexpr (SLOTS > 2) {
            walk(std::integral_constant<int, 2>{});
            walk(std::integral_constant<int, 3>{});
        }
        GPROF(3);
    }

    // ---- the virtual blocks of a block added up through LDS, slot by slot; rows stored (or partial tiles left)
    float *tl = reinterpret_cast<float *>(smem + kCtrl);          // [16 waves][16 pixels][D]
    const int64_t slab = (int64_t)b * d.H + h;
    // partial tiles of the group's query ranges: [qparts][nrb][16][D] fp32 behind one descriptor; written through and read
    // past the caches (sc0 sc1): the ranges of a group may run on different XCDs
    constexpr int kCoherent = 1 | 16;
    const __amdgpu_buffer_rsrc_t prs = make_slab_rsrc(partials + (slab * tab.ptiles_per_slab + tab.g[g].pbase) * (int64_t)(kTB * kTB * D),
                                                      (int64_t)qparts * nrb * (kTB * kTB * D) * 4);
    auto put_rows = [&](int s, int rbl, int px, int c8, const float (&sum)[8]) {
        // block rbl of segment s, pixel px of the block, channels 8 c8 ..
        const int nbx = segtab[s * 8 + kSegNbx], Hl = segtab[s * 8 + kSegH], Wl = segtab[s * 8 + kSegW];
        const int by = rbl / nbx, bx = rbl - by * nbx;
        const int y = kTB * by + (px >> 2), x = kTB * bx + (px & 3);
        if (y < Hl && x < Wl) {
            T *o = grad_value + (((int64_t)b * d.S + segtab[s * 8 + kSegStart] + y * Wl + x) * d.H + h) * d.D + c8 * 8;
            store16_stream(o, V::pack(sum));
        }
    };
    __syncthreads();
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
        const int v = wave + kWaves * sl;
        if (v < nvb) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    tl[(wave * 16 + 4 * (lane >> 4) + i) * D + 16 * nt + (lane & 15)] = acc[sl][nt][i];
        }
        __syncthreads();
        for (int e = tid; e < kWaves * 16 * LPR; e += kThreads) {
            const int w = e / (16 * LPR), px = (e / LPR) & 15, c8 = e % LPR;
            const int vv = kWaves * sl + w;
            if (vv >= nvb || vbd[vv * 4 + 3] != 0) continue;      // (the first virtual block of a block adds them up)
            const int s = vbd[vv * 4];
            const int ns = 1 << segtab[s * 8 + kSegLog2s];
            float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int j = 0; j < ns; ++j) {
                const float4 *src = reinterpret_cast<const float4 *>(tl + ((w + j) * 16 + px) * D + c8 * 8);
                const float4 a0 = src[0], a1 = src[1];
                sum[0] += a0.x; sum[1] += a0.y; sum[2] += a0.z; sum[3] += a0.w;
                sum[4] += a1.x; sum[5] += a1.y; sum[6] += a1.z; sum[7] += a1.w;
            }
            const int rbl = (vv - segtab[s * 8 + kSegV0]) >> segtab[s * 8 + kSegLog2s];
            if (qparts == 1) {
                put_rows(s, rbl, px, c8, sum);
            } else {
                const uint32_t o = (uint32_t)((((part * nrb + segtab[s * 8 + kSegRb0] + rbl) * (kTB * kTB) + px) * D + c8 * 8) * 4);
                const u32x4 w0 = {__float_as_uint(sum[0]), __float_as_uint(sum[1]), __float_as_uint(sum[2]), __float_as_uint(sum[3])};
                const u32x4 w1 = {__float_as_uint(sum[4]), __float_as_uint(sum[5]), __float_as_uint(sum[6]), __float_as_uint(sum[7])};
                __builtin_amdgcn_raw_buffer_store_b128(w0, prs, (int)o, 0, kCoherent);
                __builtin_amdgcn_raw_buffer_store_b128(w1, prs, (int)o + 16, 0, kCoherent);
            }
        }
        __syncthreads();
    }
    GPROF(4);
    if (qparts > 1) {
        // the range that arrives last adds the partial tiles up (in range order, whoever is last) and rounds
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            uint32_t *ctr = arrive + slab * kMaxGroups + g;
            const uint32_t prev = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = prev == (uint32_t)(qparts - 1) ? 1 : 0;
        }
        __syncthreads();
        if (*flag) {
            for (int e = tid; e < nrb * 16 * LPR; e += kThreads) {
                const int rbg = e / (16 * LPR), px = (e / LPR) & 15, c8 = e % LPR;
                float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                for (int pp = 0; pp < qparts; ++pp) {
                    const uint32_t o = (uint32_t)((((pp * nrb + rbg) * (kTB * kTB) + px) * D + c8 * 8) * 4);
                    const u32x4 w0 = __builtin_amdgcn_raw_buffer_load_b128(prs, (int)o, 0, kCoherent);
                    const u32x4 w1 = __builtin_amdgcn_raw_buffer_load_b128(prs, (int)o + 16, 0, kCoherent);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { sum[i] += __uint_as_float(w0[i]); sum[4 + i] += __uint_as_float(w1[i]); }
                }
                int s = 0;
                while (s + 1 < nseg && segtab[(s + 1) * 8 + kSegRb0] <= rbg) ++s;
                put_rows(s, rbg - segtab[s * 8 + kSegRb0], px, c8, sum);
            }
        }
    }
    GPROF(5);
    GPROF_COUNT(6, c1 - c0);
    GPROF_FLUSH();
}


bool shape_supported(int dtype, const Dims &d)
{
    if (dtype != 1 && dtype != 2) return false;
    if (d.D != 128 && d.D != 64) return false;
    if (d.L > 128 || d.L < 1 || d.P < 1 || d.K <= 0) return false;
    if ((int64_t)d.Nq * d.H * d.K >= (1LL << 30)) return false;              // 32-bit sample offsets inside a (b, h) slab
    if ((int64_t)d.Nq * d.H * d.D * 2 > kMaxSlabBytes) return false;          // the rows go through a buffer descriptor
    if ((int64_t)d.B * d.H * kMaxGroups * 64 > 0x3fffffffLL) return false;
    return true;
}

template <typename T, int D>
hipError_t launch(const void *go, const void *loc, const void *attn, void *gv, float *partials, uint32_t *arrive,
                  const Dims &d, const Table &t, hipStream_t st)
{
    static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_gv_mma<T, D>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    if (once != hipSuccess) return once;
    int lds = 0;
    bool parts = false;
    for (int g = 0; g < t.n_groups; ++g) {
        lds = std::max(lds, lds_bytes(t.g[g].qc, t.g[g].qc * t.g[g].nseg * d.P, Geom<D>::RB));
        parts |= t.g[g].qparts > 1;
    }
    lds = std::max(lds, kCtrl + kWaves * 16 * D * 4);                          // the epilogue's tiles
    if (lds > kLds) return hipErrorInvalidValue;
    if (parts) {
        const hipError_t e = mmfs::zero_fill(arrive, (size_t)d.B * d.H * kMaxGroups * sizeof(uint32_t), st);
        if (e != hipSuccess) return e;
    }
    const int64_t grid = (int64_t)d.B * d.H * t.wgs_per_slab;
    if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((msda_gv_mma<T, D>), dim3((unsigned)grid), dim3(kThreads), lds, st,
                       (const T *)go, (const T *)loc, (const T *)attn, (T *)gv, partials, arrive, d, t);
    return hipGetLastError();
}

}  // namespace

// ---------------------------------------------------------------- the host plan
Table make_plan(int dtype, const Dims &d, const int64_t *hs, const int64_t *hst)
{
    Table t;
    memset(&t, 0, sizeof(t));
    if (!hs || !hst || !shape_supported(dtype, d)) return t;
    if (d.Nq < knob_int(K_GV_MIN_NQ, 256)) return t;
    const int VB = d.D >= 128 ? Geom<128>::VB : Geom<64>::VB;
    const int RB = d.D * 2;
    struct Cand { int l, nb; };
    std::vector<Cand> cand;
    for (int l = 0; l < d.L; ++l) {
        const int64_t Hl = hs[2 * l], Wl = hs[2 * l + 1];
        if (Hl <= 0 || Wl <= 0 || Hl >= 32768 || Wl >= 32768) continue;
        const int64_t nb = ((Hl + kTB - 1) / kTB) * ((Wl + kTB - 1) / kTB);
        if (nb > VB) continue;
        // a block should see a couple of dozen records per chunk, or the 32-record steps run mostly empty
        if (nb * 24 * 16 > (int64_t)kMaxQc * d.P * 25) continue;
        if (hst[l] < 0 || hst[l] + Hl * Wl > d.S) continue;
        cand.push_back(Cand{l, (int)nb});
    }
    std::stable_sort(cand.begin(), cand.end(), [](const Cand &a, const Cand &b) { return a.nb < b.nb; });
    if ((int)cand.size() > kMaxLevels) cand.resize(kMaxLevels);
    if (cand.empty()) return t;
    const int64_t slabs = (int64_t)d.B * d.H;
    const double unit = (double)cand.size() * (double)slabs / (double)knob_int(K_GV_TARGET_WGS, 512);   // levels per workgroup
    // groups of levels, smallest first: as many levels as the virtual blocks allow while a chunk still holds 64 queries
    // (the chunk's rows, barriers and prefix are shared by the levels of a group); MMFS_GV_MAX_SEGS=1: a level per group
    const int max_segs = std::max(1, std::min(kMaxSegs, knob_int(K_GV_MAX_SEGS, kMaxSegs)));
    struct Tmp { std::vector<int> ci; int nb; };
    std::vector<Tmp> groups;
    for (int i = 0; i < (int)cand.size(); ++i) {
        if (groups.empty() || (int)groups.back().ci.size() >= max_segs || groups.back().nb + cand[i].nb > VB ||
            ((int)groups.back().ci.size() + 1) * d.P * 64 > kMaxSamples)
            groups.push_back(Tmp{{}, 0});
        groups.back().ci.push_back(i);
        groups.back().nb += cand[i].nb;
    }
    if ((int)groups.size() > kMaxGroups) groups.resize(kMaxGroups);
    struct Built { Group g; double work; };
    std::vector<Built> built;
    int n_levels = 0;
    for (const Tmp &tg : groups) {
        const int nseg = (int)tg.ci.size();
        // chunk size: the LDS budget and the 4096 samples the lanes keep between the two binning passes
        const int nlp = nseg * d.P;
        int qc = std::min(kMaxQc, kMaxSamples / nlp) & ~15;
        while (qc >= 16 && lds_bytes(qc, qc * nlp, RB) > kLds) qc -= 16;
        if (qc < 16) continue;                                      // (too many points per query: the sorted backward keeps these levels)
        qc = std::min(qc, (d.Nq + 15) / 16 * 16);
        Group g;
        memset(&g, 0, sizeof(g));
        // virtual blocks: every level gets about VB / nseg of them
        int l2[kMaxSegs], total = 0;
        for (int i = 0; i < nseg; ++i) {
            int s = 0;
            while (s < 4 && (cand[tg.ci[i]].nb << (s + 1)) * nseg <= VB) ++s;
            l2[i] = s;
            total += cand[tg.ci[i]].nb << s;
        }
        while (total > VB) {
            int worst = -1;
            for (int i = 0; i < nseg; ++i) if (l2[i] > 0 && (worst < 0 || l2[i] > l2[worst])) worst = i;
            if (worst < 0) break;
            total -= cand[tg.ci[worst]].nb << (l2[worst] - 1);
            --l2[worst];
        }
        if (total > VB) continue;
        // segments by split, descending: a block's virtual blocks then sit side by side inside one slot of 16 waves
        int order[kMaxSegs];
        for (int i = 0; i < nseg; ++i) order[i] = i;
        std::stable_sort(order, order + nseg, [&](int a, int b) { return l2[a] > l2[b]; });
        int v0 = 0, rb0 = 0;
        for (int k = 0; k < nseg; ++k) {
            const Cand &c = cand[tg.ci[order[k]]];
            Level &lv = t.lv[n_levels];
            lv.level = c.l; lv.Hl = (int)hs[2 * c.l]; lv.Wl = (int)hs[2 * c.l + 1]; lv.lstart = (int)hst[c.l];
            lv.nbx = (lv.Wl + kTB - 1) / kTB; lv.nby = (lv.Hl + kTB - 1) / kTB;
            g.seg[k].lslot = (uint16_t)n_levels; g.seg[k].log2s = (uint16_t)l2[order[k]];
            g.seg[k].v0 = (uint16_t)v0; g.seg[k].rb0 = (uint16_t)rb0;
            v0 += c.nb << l2[order[k]]; rb0 += c.nb;
            t.skip[c.l >> 6] |= 1ull << (c.l & 63);
            ++n_levels;
        }
        g.nseg = (uint16_t)nseg; g.nvb = (uint16_t)v0; g.nrb = (uint16_t)rb0; g.qc = (uint16_t)qc;
        const int nchunks = (d.Nq + qc - 1) / qc;
        int qp = (int)((double)nseg / std::max(unit, 1e-9) + 0.5);
        qp = std::max(1, std::min(qp, nchunks));
        g.qparts = (uint16_t)qp;
        built.push_back(Built{g, (double)nseg / qp});
    }
    if (built.empty()) { memset(&t, 0, sizeof(t)); return t; }
    // heavy workgroups first
    std::stable_sort(built.begin(), built.end(), [](const Built &a, const Built &b) { return a.work > b.work; });
    int wg0 = 0;
    uint32_t pbase = 0;
    for (int i = 0; i < (int)built.size(); ++i) {
        Group &g = built[i].g;
        g.wg0 = (uint16_t)wg0; wg0 += g.qparts;
        g.pbase = pbase;
        if (g.qparts > 1) pbase += (uint32_t)g.qparts * g.nrb;
        t.g[i] = g;
    }
    t.n_levels = n_levels; t.n_groups = (int)built.size(); t.wgs_per_slab = wg0; t.ptiles_per_slab = (int)pbase;
    return t;
}

int64_t workspace_bytes(const Table &t, const Dims &d)
{
    if (t.n_groups == 0) return 0;
    const int64_t slabs = (int64_t)d.B * d.H;
    return (slabs * kMaxGroups * 4 + 255) / 256 * 256 + slabs * t.ptiles_per_slab * (int64_t)(kTB * kTB) * d.D * 4;
}

hipError_t backward_value(int dtype, const void *grad_out, const void *loc, const void *attn, void *grad_value,
                          void *workspace, const Dims &d, const Table &t, hipStream_t st)
{
    if (t.n_groups == 0) return hipSuccess;
    uint32_t *arrive = reinterpret_cast<uint32_t *>(workspace);
    float *partials = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + ((int64_t)d.B * d.H * kMaxGroups * 4 + 255) / 256 * 256);
    if (dtype == 1) {
        if (d.D == 128) return launch<half_t, 128>(grad_out, loc, attn, grad_value, partials, arrive, d, t, st);
        return launch<half_t, 64>(grad_out, loc, attn, grad_value, partials, arrive, d, t, st);
    }
    if (d.D == 128) return launch<bf16_t, 128>(grad_out, loc, attn, grad_value, partials, arrive, d, t, st);
    return launch<bf16_t, 64>(grad_out, loc, attn, grad_value, partials, arrive, d, t, st);
}

}  // namespace gv
}  // namespace mmfs

#ifdef MMFS_PROFILE_GV
extern "C" int mmfs_debug_gv_profile(unsigned long long *out, int reset)
{
    static unsigned long long host[mmfs::gv::kGProfSlots * 8];
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(mmfs::gv::g_gv_prof), sizeof(host));
    for (int i = 0; i < 8; ++i) out[i] = 0;
    for (int s = 0; s < mmfs::gv::kGProfSlots; ++s)
        for (int i = 0; i < 8; ++i) out[i] += host[s * 8 + i];
    if (e == hipSuccess && reset) {
        for (auto &v : host) v = 0;
        e = hipMemcpyToSymbol(HIP_SYMBOL(mmfs::gv::g_gv_prof), host, sizeof(host));
    }
    return (int)e;
}
#endif
