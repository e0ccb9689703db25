// msda_taps_mma.hip -- grad_loc / grad_attn of multi-scale deformable attention for gfx950, second
// formulation: ONE kernel for all levels, the small ones resident in LDS and contracted on the matrix cores.
//
// Replaces the location / weight half of the reference backward
//   mm_interleaved/models/utils/ops/src/cuda/ms_deform_im2col_cuda.cuh:304-404 (per sample: two barriers and a
//   shared-memory reduction over the D channel threads), algebra :119-161
// for 16-bit storage and heads of 128 channels, and with it this repository's own pair of kernels -- the
// row-gather taps of the large levels (msda_bwd.hip) next to a DENSE matrix-core product over every pixel of the
// small levels (msda_dense.hip: 64 queries x 256 pixels x D per step, of which a query needs 4 pixels per sample).
//
// Per sample the gradients need the four dots  d_c = grad_out[q, :] . value[corner_c, :]  and scalar algebra:
//   * organisation and LDS image as in msda_fwd_mma.hip (a 1024-lane workgroup per run of queries of one
//     (b, h); the levels that fit live in LDS -- in NATURAL channel order here; 16 waves on their own);
//   * large levels: row gather, v_dot2c on the rows as loaded, joint DPP reduction (msda_dots.h);
//   * LDS-resident levels: v_mfma_f32_16x16x32 with the GATHERED value rows as the A operand (M = 16 rows =
//     4 samples x 4 corners of ONE query; a lane reads 16 bytes of "its" row straight out of the image) and the
//     queries' grad_out rows as the B operand (K = 32 channels per step, chained over D / 32 steps; column n is
//     query n mod QPW).  D[4s + c, j] is dot c of sample s of query j: the lane that holds it holds all four
//     corners of its sample and writes them into the sample's record.  A product's other columns pair a
//     query's rows with another query's gradient: computed, never read -- a non-finite row or gradient stays
//     with its own sample, as in the row gather;
//   * the per-sample algebra and the (coalesced) stores are the store pass of msda_bwd_vec, one lane per sample.
//
// fp32 storage, other head widths, L > 64, small query counts: msda_bwd_vec (+ msda_taps_coarse when the caller
// brings a host copy of the level table).
#include "msda_dots.h"
#include "msda_env.h"
#include "msda_mma_common.h"
#include "msda_plan.h"
#include <cstring>
#include "msda_launch.h"
#include <cstdlib>

namespace mmfs {

using namespace mma;

// Development aid (tools/exp_build.sh tapsprof "-DMMFS_PROFILE_TAPS"; tools/taps_prof.py): shader clocks per phase of a wave
#ifdef MMFS_PROFILE_TAPS
constexpr int kTProfSlots = 4096;
__device__ unsigned long long g_taps_prof[kTProfSlots * 8];
#define APROF_DECL unsigned long long aprof_c = __builtin_readcyclecounter(), aprof_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define APROF(i) do { const unsigned long long tn = __builtin_readcyclecounter(); aprof_t[i] += tn - aprof_c; aprof_c = tn; } while (0)
#define APROF_COUNT(i, v) do { aprof_t[i] += (unsigned long long)(v); } while (0)
#define APROF_FLUSH() do { if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_taps_prof[(blockIdx.x % kTProfSlots) * 8 + i_], aprof_t[i_]); } while (0)
#else
#define APROF_DECL do {} while (0)
#define APROF(i) do {} while (0)
#define APROF_COUNT(i, v) do {} while (0)
#define APROF_FLUSH() do {} while (0)
#endif

#ifndef MMFS_TAPS_CHAINS
#define MMFS_TAPS_CHAINS 1      // measured (r03h): 1: 144.7 us, 2: 152.4 us (the second chain's registers spill)
#endif
constexpr int kChains = MMFS_TAPS_CHAINS;       // product chains in flight in the matrix-core phase (1, 2 or 4)

template <typename T, int D>
__global__ void __launch_bounds__(kMmaThreads)
msda_taps_mma(const T *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ start,
              const T *__restrict__ loc, const T *__restrict__ attn, const T *__restrict__ grad_out,
              T *__restrict__ grad_loc, T *__restrict__ grad_attn, const Dims d, const int q_per_wg, const int img_budget,
              const int n_runs, const blk::PrepareJob job)
{
    constexpr int LPI = D * 2 / 16, QPW = 64 / LPI;
    constexpr int GSH = QPW * D * 2;                                      // the wave's grad_out rows (B operand source)
    typedef MmaGeom<D, GSH> G;
    typedef FwdMma<T> M;
    constexpr int NKS = D / 32;                                           // K steps of a product (32 channels each)
    static_assert(D == 128 || D == 64, "taps on the matrix cores: heads of 128 / 64 channels (four / eight queries per wave)");
    constexpr int PS = QPW / 4;                                           // staging / store passes (64 lanes = 4 queries x 16 samples)
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    int *tab = reinterpret_cast<int *>(smem);
    unsigned char *img = smem + G::IMG0;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int L = d.L;
    const int64_t HD = (int64_t)d.H * d.D;
    const uint32_t row_bytes = (uint32_t)(HD * sizeof(T));
    APROF_DECL;
    build_level_table<D>(tab, img, shapes, start, L, tid, img_budget);
    // one run per workgroup, or persistent workgroups (persistent_grid, msda_mma_common.h)
    for (int run = blockIdx.x; run < n_runs; run += gridDim.x) {
    const int h = run % d.H;
    const int tq = run / d.H;
    const int q_wg0 = (tq % d.q_tiles) * q_per_wg;
    const int b = tq / d.q_tiles;
    const T *slab = value + ((int64_t)b * d.S) * HD + (int64_t)h * d.D;
    const __amdgpu_buffer_rsrc_t rsrc = make_slab_rsrc(slab, ((int64_t)d.S * HD - (int64_t)h * d.D) * (int64_t)sizeof(T));

    // ---- from here on every wave works on its own
    unsigned char *wrec = smem + G::TAB_BYTES + wave * G::WSCR;         // records: [QPW][kChunk] x 32 bytes
    unsigned char *gsh = wrec + G::REC_BYTES;                            // grad_out rows of the wave's queries
    const int qi = lane / LPI, lig = lane % LPI;                          // row-gather role
    const int kk = lane & 15;                                             // staging / store role: sample of the chunk
    const uint32_t lane_off = (uint32_t)(lig * 16);
    const bool pair_ok = (((uintptr_t)loc | (uintptr_t)grad_loc) & (2 * sizeof(T) - 1)) == 0;
    // product roles
    const int am = lane & 15, akb = lane >> 4;                            // A: row m = 4 * sample + corner, K block
    const int bn = lane & 15, bkb = lane >> 4;                            // B: column (query bn mod QPW), K block

    const int q_wg1 = min(d.Nq, q_wg0 + q_per_wg);
    const int n_chunks = (d.K + kChunk - 1) / kChunk;
    const int q_first = q_wg0 + wave * QPW;
    const int n_groups = q_first < q_wg1 ? (q_wg1 - q_first + kMmaWaves * QPW - 1) / (kMmaWaves * QPW) : 0;
    const int n_steps = n_groups * n_chunks;
    const uint16_t *loc_wg = reinterpret_cast<const uint16_t *>(loc) + 2 * (((int64_t)b * d.Nq * d.H + h) * d.K);
    const uint16_t *attn_wg = reinterpret_cast<const uint16_t *>(attn) + (((int64_t)b * d.Nq * d.H + h) * d.K);
    T *gl_wg = grad_loc + 2 * (((int64_t)b * d.Nq * d.H + h) * d.K);
    T *ga_wg = grad_attn + (((int64_t)b * d.Nq * d.H + h) * d.K);
    const uint32_t q_stride = (uint32_t)d.H * (uint32_t)d.K;
    // (the next step's sample words are requested while this step gathers; kept RAW until they are used)
    uint32_t pf_w0[PS], pf_w1[PS], pf_a[PS];
#pragma unroll
    for (int ps = 0; ps < PS; ++ps) pf_w0[ps] = pf_w1[ps] = pf_a[ps] = 0u;
    auto prefetch = [&](int step) {
        const int k = (step % n_chunks) * kChunk + kk;
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) {
            const int q = q_first + (step / n_chunks) * (kMmaWaves * QPW) + ps * 4 + (lane >> 4);
            pf_w0[ps] = pf_w1[ps] = pf_a[ps] = 0u;
            if (step < n_steps && k < d.K && q < d.Nq) {
                const uint32_t s = (uint32_t)q * q_stride + (uint32_t)k;
                const uint16_t *lw = loc_wg + 2 * (size_t)s;
                if (pair_ok) pf_w0[ps] = *reinterpret_cast<const uint32_t *>(lw);
                else { pf_w0[ps] = lw[0]; pf_w1[ps] = lw[1]; }
                pf_a[ps] = attn_wg[s];
            }
        }
    };
    prefetch(0);
    // ---- the run's image
    if (run != (int)blockIdx.x) __syncthreads();                          // every wave is done with the previous image
    APROF(0);                                                             // run setup + barrier
    fill_image<D, false>(tab, img, rsrc, row_bytes, L, d.S, tid);         // natural channel order
    APROF(1);                                                             // image fill
    uint4 graw = make_uint4(0u, 0u, 0u, 0u);                              // this lane's 16 bytes of its query's grad_out row
    s16x8 Bf[NKS];                                                        // B operand: the wave's queries, all of D

    for (int step = 0; step < n_steps; ++step) {
#ifndef TAPS_NO_SETPRIO
        // (as msda_fwd_wq.hip: the wave that is behind the others of its SIMD issues first, so that the waves of the workgroup
        // reach the barrier in front of the next image fill together: 126.8 -> 120.7 us at the north star, r05h)
        { const int left = n_steps - 1 - step; if (left >= 3) __builtin_amdgcn_s_setprio(3); else if (left == 2) __builtin_amdgcn_s_setprio(2); else if (left == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#endif
        const int q0 = q_first + (step / n_chunks) * (kMmaWaves * QPW);
        const int chunk = step % n_chunks;
        const int k0 = chunk * kChunk;
        const int k = k0 + kk;
        const bool k_ok = k < d.K;
        const int l = k_ok ? k / d.P : 0;
        const int Hl = tab[kTabInts * l], Wl = tab[kTabInts * l + 1], lstart = tab[kTabInts * l + 2];
        const int ibase = tab[kTabInts * l + 3], lp = tab[kTabInts * l + 4];
        const bool in_lds = k_ok && ibase >= 0;
        const uint32_t lmask = (uint32_t)__builtin_amdgcn_ballot_w64(in_lds) & 0xffffu;
        const uint32_t gmask = (uint32_t)__builtin_amdgcn_ballot_w64(k_ok && !in_lds) & 0xffffu;
        const int n_l = __builtin_popcount(lmask);
        const uint32_t below = (1u << kk) - 1u;
        const int ridx = in_lds ? kChunk - 1 - __builtin_popcount(lmask & below) : __builtin_popcount(gmask & below);

        wave_sync();                                                      // the previous step's records are stored
        if (chunk == 0) {
            // ---- a new group of queries: their grad_out rows, once in the row-gather layout (registers), once in
            // LDS from where every lane takes its B fragments
            const int q = q0 + qi;
            graw = make_uint4(0u, 0u, 0u, 0u);
            if (q < d.Nq)
                graw = *reinterpret_cast<const uint4 *>(grad_out + (((int64_t)b * d.Nq + q) * d.H + h) * d.D + lig * 8);
            *reinterpret_cast<uint4 *>(gsh + qi * (D * 2) + lig * 16) = graw;
        }
        // ---- stage: one sample per lane
        uint32_t live = 0u;
        uint4 st_r0[PS];
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) {
            const int sq = ps * 4 + (lane >> 4);
            const int q = q0 + sq;
            // (a row-gather sample that reads nothing must not read row 0 either when the wave walks its tap for
            // another query: "outside" offsets, whose loads return zeros -> zero dots)
            uint4 r0 = in_lds ? make_uint4(0u, 0u, 0u, 0u) : make_uint4(kOobOffset, kOobOffset, kOobOffset, kOobOffset);
            uint4 r1 = make_uint4(0u, 0u, 0u, 0u);
            bool reads = false;
            if (k_ok && q < d.Nq) {
                asm volatile("" : "+v"(pf_w0[ps]), "+v"(pf_w1[ps]), "+v"(pf_a[ps]));
                const uint32_t xb = pair_ok ? (pf_w0[ps] & 0xffffu) : pf_w0[ps], yb = pair_ok ? (pf_w0[ps] >> 16) : pf_w1[ps];
                const float lx = to_f32(__builtin_bit_cast(T, (uint16_t)xb)), ly = to_f32(__builtin_bit_cast(T, (uint16_t)yb));
                const float a = to_f32(__builtin_bit_cast(T, (uint16_t)pf_a[ps]));
                const float y = ly * (float)Hl - 0.5f, x = lx * (float)Wl - 0.5f;
                const bool inside = (y > -1.f) && (x > -1.f) && (y < (float)Hl) && (x < (float)Wl);
                const float yf = floorf(y), xf = floorf(x);
                const int y0 = inside ? (int)yf : 0, x0 = inside ? (int)xf : 0;
                const float fy = inside ? y - yf : 0.f, fx = inside ? x - xf : 0.f;
                // lazy_attn: nobody reads the gradients of a zero-weight sample -> no rows, zeros out
                const bool on = inside && !(d.lazy_attn && a == 0.f);
                const bool top = y0 >= 0, left = x0 >= 0, bottom = y0 + 1 <= Hl - 1, right = x0 + 1 <= Wl - 1;
                const bool ok[4] = {on && top && left, on && top && right, on && bottom && left, on && bottom && right};
                r1 = make_uint4(__float_as_uint(fx), __float_as_uint(fy), __float_as_uint(a), (uint32_t)l);
                if (in_lds) {
                    const int o00 = ibase + y0 * lp + x0 * G::RP;
                    r0 = make_uint4(ok[0] ? o00 : 0, ok[1] ? o00 + G::RP : 0, ok[2] ? o00 + lp : 0, ok[3] ? o00 + lp + G::RP : 0);
                } else {
                    reads = ok[0] || ok[1] || ok[2] || ok[3];
                    if (reads) {
                        const int p00 = lstart + y0 * Wl + x0;
                        const int row[4] = {p00, p00 + 1, p00 + Wl, p00 + Wl + 1};
                        uint32_t o[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[c] = ok[c] ? (uint32_t)row[c] * row_bytes : kOobOffset;
                        r0 = make_uint4(o[0], o[1], o[2], o[3]);
                    }
                }
            }
            const unsigned long long bl = __builtin_amdgcn_ballot_w64(reads);
            live |= (uint32_t)(bl | (bl >> 16) | (bl >> 32) | (bl >> 48)) & 0xffffu;
            st_r0[ps] = r0;
            if (k_ok) {
                uint4 *dst = reinterpret_cast<uint4 *>(wrec + sq * G::QSTRIDE + ridx * 32);
                dst[1] = r1;
            }
        }
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) {
            // a tap no query of the wave reads is never walked: its records keep what is written here -- zero dots
            uint4 r0 = st_r0[ps];
            if (!in_lds && !((live >> kk) & 1u)) r0 = make_uint4(0u, 0u, 0u, 0u);
            if (k_ok) *reinterpret_cast<uint4 *>(wrec + (ps * 4 + (lane >> 4)) * G::QSTRIDE + ridx * 32) = r0;
        }
        wave_sync();
        APROF(2);                                                         // stage
        if (chunk == 0) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                Bf[ks] = __builtin_bit_cast(s16x8, *reinterpret_cast<const uint4 *>(gsh + (bn & (QPW - 1)) * (D * 2) + (4 * ks + bkb) * 16));
        }

        // ---- LDS-resident levels on the matrix cores: tiles of 4 samples x 4 corners of one query
        auto mma_phase = [&]() {
            for (int t4 = 0; 4 * t4 < n_l; ++t4) {
                const int r = 4 * t4 + (am >> 2);                         // rank of this row's sample among the chunk's LDS samples
                const int rs = 4 * t4 + (lane >> 4);                      // ... and of the sample whose dots this lane's row quad holds
#pragma unroll 1
                for (int j = 0; j < QPW; j += kChains) {                  // kChains queries at a time: independent product chains
                    const unsigned char *ap[kChains];
#pragma unroll
                    for (int u = 0; u < kChains; ++u) {
                        uint32_t off = 0u;                                // the zero row
                        if (r < n_l) off = *reinterpret_cast<const uint32_t *>(wrec + (j + u) * G::QSTRIDE + (kChunk - 1 - r) * 32 + 4 * (am & 3));
                        ap[u] = img + off + 16 * akb;
                    }
                    f32x4 acc[kChains];
#pragma unroll
                    for (int u = 0; u < kChains; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
                        for (int u = 0; u < kChains; ++u) {
                            const s16x8 A = __builtin_bit_cast(s16x8, *reinterpret_cast<const uint4 *>(ap[u] + 64 * ks));
                            acc[u] = M::run(A, Bf[ks], acc[u]);
                        }
                    }
                    // column j, row quad s: the four corner dots of sample 4 * t4 + s of query j
#pragma unroll
                    for (int u = 0; u < kChains; ++u)
                        if (bn == j + u && rs < n_l)
                            *reinterpret_cast<uint4 *>(wrec + (j + u) * G::QSTRIDE + (kChunk - 1 - rs) * 32) =
                                make_uint4(__float_as_uint(acc[u][0]), __float_as_uint(acc[u][1]), __float_as_uint(acc[u][2]), __float_as_uint(acc[u][3]));
                }
            }
        };

        // ---- row gather of the large levels, software-pipelined over the samples that read something
        {
            unsigned m = (unsigned)__builtin_amdgcn_readfirstlane((int)live);
            const unsigned gm = (unsigned)__builtin_amdgcn_readfirstlane((int)gmask);
            uint4 *recs = reinterpret_cast<uint4 *>(wrec + qi * G::QSTRIDE);
            auto issue = [&](uint4 (&raw)[4], int &gi) {
                const int kq = __builtin_ctz(m);
                m &= m - 1u;
                gi = __builtin_popcount(gm & ((1u << kq) - 1u));
                const uint4 rr = recs[2 * gi];
                raw[0] = buffer_load16(rsrc, rr.x + lane_off);
                raw[1] = buffer_load16(rsrc, rr.y + lane_off);
                raw[2] = buffer_load16(rsrc, rr.z + lane_off);
                raw[3] = buffer_load16(rsrc, rr.w + lane_off);
                __builtin_amdgcn_sched_barrier(0);
            };
            auto consume = [&](const uint4 (&raw)[4], int gi) {
                float dot[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) dot[c] = RowDot<T>::run(graw, raw[c]);
                if (LPI == 16) group_sum4_row(dot, lig); else group_sum4_half(dot, lig);     // totals in the group's lane 0
                if (lig == 0)
                    recs[2 * gi] = make_uint4(__float_as_uint(dot[0]), __float_as_uint(dot[1]), __float_as_uint(dot[2]), __float_as_uint(dot[3]));
                __builtin_amdgcn_sched_barrier(0);
            };
            const int n_live = __builtin_popcount(m);
            if (n_live >= 4 && (n_live & 3) == 0) {
                uint4 r0[4], r1[4], r2[4], r3[4];
                int g0, g1, g2, g3;
                issue(r0, g0); issue(r1, g1); issue(r2, g2); issue(r3, g3);
                prefetch(step + 1);
                __builtin_amdgcn_sched_barrier(0);
                APROF(3);                                                 // first issues + next requests
                mma_phase();
                __builtin_amdgcn_sched_barrier(0);
                APROF(4);                                                 // matrix-core phase
                for (int i = 4; i < n_live; i += 4) {
                    consume(r0, g0); issue(r0, g0);
                    consume(r1, g1); issue(r1, g1);
                    consume(r2, g2); issue(r2, g2);
                    consume(r3, g3); issue(r3, g3);
                }
                consume(r0, g0); consume(r1, g1); consume(r2, g2); consume(r3, g3);
                APROF(5);                                                 // gather loop
            } else {
                prefetch(step + 1);
                __builtin_amdgcn_sched_barrier(0);
                mma_phase();
                __builtin_amdgcn_sched_barrier(0);
                uint4 rawA[4], rawB[4];
                int gA, gB;
                if (n_live & 1) { issue(rawA, gA); consume(rawA, gA); }
                if (n_live >= 2) {
                    issue(rawA, gA);
                    for (int i = 2; i < n_live - 1; i += 2) {
                        issue(rawB, gB);
                        consume(rawA, gA);
                        issue(rawA, gA);
                        consume(rawB, gB);
                    }
                    issue(rawB, gB);
                    consume(rawA, gA);
                    consume(rawB, gB);
                }
            }
        }
        wave_sync();

        // ---- per-sample algebra of the reference (cuh:119-161) on the four corner dots, one lane per sample;
        // coalesced stores of this chunk's grad_attn / grad_loc
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) {
            const int sq = ps * 4 + (lane >> 4);
            const int q = q0 + sq;
            if (k_ok && q < d.Nq) {
                const uint4 *rec = reinterpret_cast<const uint4 *>(wrec + sq * G::QSTRIDE + ridx * 32);
                const uint4 res = rec[0], meta = rec[1];
                const float d0 = __uint_as_float(res.x), d1 = __uint_as_float(res.y), d2 = __uint_as_float(res.z), d3 = __uint_as_float(res.w);
                const float fx = __uint_as_float(meta.x), fy = __uint_as_float(meta.y), a = __uint_as_float(meta.z);
                const float gy = 1.f - fy, gx = 1.f - fx;
                const float ga = (gy * gx) * d0 + (gy * fx) * d1 + (fy * gx) * d2 + (fy * fx) * d3;
                const float dw = gy * (d1 - d0) + fy * (d3 - d2);
                const float dh = gx * (d2 - d0) + fx * (d3 - d1);
                const uint32_t s = (uint32_t)q * q_stride + (uint32_t)k;
                store_stream(ga_wg + s, (T)ga);              // (final results of the step: non-temporal, msda_device.h)
                store_xy(gl_wg, (int64_t)s, pair_ok, (float)Wl * dw * a, (float)Hl * dh * a);
            }
        }
        APROF(6);                                                         // per-sample algebra + stores
    }
    APROF_COUNT(7, n_steps);
    }   // runs
    APROF_FLUSH();
    // ---- the opening launch of the grad_value half, hosted here (one launch less per backward): the FIRST workgroup
    // clears the sort's cursors and plans it; its LDS is free by now.  (One workgroup per run: it is done long before the
    // kernel is.  Persistent workgroups all end together and the plan is 4 us of one of them: a workgroup of its own
    // for it measured no different, profiles/r03_experiments.md r03bm.)
    if (job.cursor_words > 0 && blockIdx.x == 0) {
        __syncthreads();
        blk::prepare_tail(job, smem);
    }
}

// ---------------------------------------------------------------- launcher
template <typename T, int D>
static hipError_t launch_taps_mma(const void *value, const int64_t *shapes, const int64_t *start, const void *loc,
                                  const void *attn, const void *go, void *gl, void *ga, Dims d, hipStream_t st,
                                  const blk::PrepareJob *job)
{
    blk::PrepareJob jb;
    if (job != nullptr) jb = *job;
    else { memset(&jb, 0, sizeof(jb)); }
    typedef MmaGeom<D, (64 / (D * 2 / 16)) * D * 2> G;
    static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_taps_mma<T, D>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal);
    if (once != hipSuccess) return once;
    const int env_q = knob_int(K_TAPS_MMA_QPW, 0);
    const int unit = kMmaWaves * G::QPW;
    const int q_per_wg = pick_queries_per_run(d, unit, env_q);       // (256, or shorter runs for few queries: msda_mma_common.h)
    d.q_tiles = (d.Nq + q_per_wg - 1) / q_per_wg;
    const int64_t runs = (int64_t)d.B * d.q_tiles * d.H;
    if (runs > 0x7fffffffLL) return hipErrorInvalidValue;
    const int grid = (int)persistent_grid(runs, d.H);                       // (one workgroup per CU when there are many runs, else = runs)
    hipLaunchKernelGGL((msda_taps_mma<T, D>), dim3((unsigned)grid), dim3(kMmaThreads), kLdsTotal, st,
                       (const T *)value, shapes, start, (const T *)loc, (const T *)attn, (const T *)go, (T *)gl, (T *)ga,
                       d, q_per_wg, kLdsTotal - G::IMG0, (int)runs, jb);
    return hipGetLastError();
}

bool taps_mma_supported(int dtype, const Dims &d)
{
    if (dtype != 1 && dtype != 2) return false;
    if (d.D != 128 && d.D != 64) return false;
    if (d.L > kMmaMaxLevels || d.K <= 0) return false;
    if ((int64_t)d.Nq * d.H * d.K >= (1LL << 30)) return false;            // 32-bit sample offsets inside a (b, h) slab
    return (int64_t)d.S * d.H * d.D * 2 <= kMaxSlabBytes;
}

bool taps_mma_applies(int dtype, const Dims &d)
{
    const char *algo = knob_str(K_TAPS_ALGO);                          // "vec": never; "mma": whenever the shape allows
    if (d.taps_algo == 1 || (d.taps_algo == 0 && algo && algo[0] == 'v')) return false;
    if (!taps_mma_supported(dtype, d)) return false;
    if (d.taps_algo == 2 || (algo && algo[0] == 'm')) return true;
    if (d.D != 128) return false;                              // (heads of 64 channels: round 5, opt-in until measured -- below)
    return d.Nq >= 64 && (int64_t)d.Nq * d.K >= 4096;          // (as fwd_mma_applies: samples per image fill; speed-test shape 139 -> 102 us)
}

hipError_t backward_taps_mma(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                             const void *loc, const void *attn, const void *grad_out, void *grad_loc, void *grad_attn,
                             const Dims &d, hipStream_t st, const blk::PrepareJob *job)
{
    if (d.D == 64) {
        if (dtype == 1) return launch_taps_mma<half_t, 64>(value, shapes, start, loc, attn, grad_out, grad_loc, grad_attn, d, st, job);
        return launch_taps_mma<bf16_t, 64>(value, shapes, start, loc, attn, grad_out, grad_loc, grad_attn, d, st, job);
    }
    if (dtype == 1) return launch_taps_mma<half_t, 128>(value, shapes, start, loc, attn, grad_out, grad_loc, grad_attn, d, st, job);
    return launch_taps_mma<bf16_t, 128>(value, shapes, start, loc, attn, grad_out, grad_loc, grad_attn, d, st, job);
}

}  // namespace mmfs

#ifdef MMFS_PROFILE_TAPS
extern "C" int mmfs_debug_taps_profile(unsigned long long *out, int reset)
{
    static unsigned long long host[mmfs::kTProfSlots * 8];
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(mmfs::g_taps_prof), sizeof(host));
    for (int i = 0; i < 8; ++i) out[i] = 0;
    for (int s = 0; s < mmfs::kTProfSlots; ++s)
        for (int i = 0; i < 8; ++i) out[i] += host[s * 8 + i];
    if (e == hipSuccess && reset) {
        for (auto &v : host) v = 0;
        e = hipMemcpyToSymbol(HIP_SYMBOL(mmfs::g_taps_prof), host, sizeof(host));
    }
    return (int)e;
}
#endif
