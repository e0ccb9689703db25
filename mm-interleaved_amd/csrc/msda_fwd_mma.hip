// msda_fwd_mma.hip -- forward of multi-scale deformable attention for gfx950, second formulation:
// the SMALL levels of the pyramid live in LDS and are sampled by the matrix cores.
//
// Replaces the reference forward
//   mm_interleaved/models/utils/ops/src/cuda/ms_deform_im2col_cuda.cuh:240-302
// for 16-bit storage and head widths of 64 / 128 channels.  Why a second formulation (round 2's
// counters): msda_fwd_vec IS its row stream -- 4 pixel rows per sample through the texture
// path, 4.29 GB at the north-star shape, and that path returns 64 B per clock per CU whatever level of
// the cache answers; the unpack + multiply-add of every 16-bit row element keeps the vector ALU busy
// next to it.  But every level receives the same number of samples whatever its size, and the small
// levels of the pyramid (16x16 and 8x8: half of all samples at the north star) are 80 KiB per (b, h):
//
//   * a 1024-lane workgroup owns a run of queries of ONE (batch, head) and first copies that slab's
//     small levels into LDS (which levels fit is decided on the device from the level table: smallest
//     first, until the budget is full), channel-permuted and with a row pitch of D*e + 32 bytes;
//   * its 16 waves then work on their own: a wave stages the samples of QPW = 1024/D queries, one
//     sample per lane (locations, bilinear weights), into wave-private LDS records -- no workgroup
//     barrier after the fill;
//   * samples of the LARGE levels take the row-gather path of msda_fwd_vec (buffer loads of whole
//     D*e-byte rows, fp32 multiply-add in the vector ALU, counted software pipeline over the live taps);
//   * samples of the LDS-resident levels never touch the texture path or the vector ALU's
//     multiply-adds: the 32 pixel rows of 8 samples x 4 corners of a query are the B operand of
//     v_mfma_f32_16x16x32_{bf16,f16} (K = row, N = 16 channels), fetched straight out of the LDS image
//     with the transposing ds_read_b64_tr_b16 (every lane supplies the address of 8 bytes of "its"
//     row, so the rows of a product are gathered, not contiguous); the A operand holds the queries'
//     weights -- row 4j the leading 16 bits of the fp32 weight (corner weight x attention weight) of
//     query j's samples, row 4j+1 the rounded remainder, as in the grad_value product of
//     msda_bwd_tile.hip: hi + lo carries >= 16 significant bits, inside the storage type's rounding.
//     The product's rows 4j, 4j+1 land in the 16 lanes that own query j in the row-gather layout, and
//     the LDS image is channel-permuted so that column n of column group g is channel 8n + g: exactly
//     accumulator g of lane n.  One product serves ONE query (its B rows are its own), the other
//     queries' rows of the result are discarded: a non-finite value row reaches the queries that sample
//     it and no other (cuh:58-81).  Corners outside the map, samples that fail the range test and
//     samples of zero attention weight point at a row of zeros kept in the image.
//
// Bank conflicts of the transposing read (32-lane groups, 8 rows x 32 bytes each): the pitch D*e + 32
// puts a footprint's x-neighbours 32 bytes apart (mod 256), and every level's line pitch is padded so
// that its y-neighbours are 64 bytes apart: the four corners of a sample never collide.
//
// fp32 storage, other head widths, L > 64 and small query counts stay on msda_fwd_vec (msda_fwd.hip).
#include "msda_mma_common.h"
#include "msda_env.h"
#include "msda_launch.h"
#include <cstdlib>

namespace mmfs {

using namespace mma;

// Development aid (tools/exp_build.sh fprof "-DMMFS_PROFILE_FWD"; tools/fwd_prof.py): shader clocks per phase of a
// wave, summed over the waves of a workgroup slot, read back with mmfs_debug_fwd_profile().
#ifdef MMFS_PROFILE_FWD
constexpr int kFProfSlots = 4096;
__device__ unsigned long long g_fwd_prof[kFProfSlots * 8];
#define FPROF_DECL unsigned long long fprof_c = __builtin_readcyclecounter(), fprof_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define FPROF(i) do { const unsigned long long tn = __builtin_readcyclecounter(); fprof_t[i] += tn - fprof_c; fprof_c = tn; } while (0)
#define FPROF_COUNT(i, v) do { fprof_t[i] += (unsigned long long)(v); } while (0)
#define FPROF_FLUSH() do { if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_fwd_prof[(blockIdx.x % kFProfSlots) * 8 + i_], fprof_t[i_]); } while (0)
#else
#define FPROF_DECL do {} while (0)
#define FPROF(i) do {} while (0)
#define FPROF_COUNT(i, v) do {} while (0)
#define FPROF_FLUSH() do {} while (0)
#endif

template <typename T, int D>
__global__ void __launch_bounds__(kMmaThreads)
msda_fwd_mma(const T *__restrict__ value, const int64_t *__restrict__ shapes,
             const int64_t *__restrict__ start, const T *__restrict__ loc,
             const T *__restrict__ attn, T *__restrict__ out, const Dims d, const int q_per_wg, const int img_budget,
             const int n_runs)
{
    typedef MmaGeom<D> G;
    typedef FwdMma<T> M;
    typedef Vec16<T> V;
    constexpr int VEC = 8;
    constexpr int LPI = G::LPI, QPW = G::QPW, NG = G::NG;
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    int *tab = reinterpret_cast<int *>(smem);
    unsigned char *img = smem + G::IMG0;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    FPROF_DECL;
    const int L = d.L;
    const int64_t HD = (int64_t)d.H * d.D;
    const uint32_t row_bytes = (uint32_t)(HD * sizeof(T));
    build_level_table<D>(tab, img, shapes, start, L, tid, img_budget);
    // One run of queries per workgroup, dealt by the dispatcher in the order run -> (b, h, queries) with h = run mod H
    // (a head's slab stays in one XCD's L2).  The loop serves persistent workgroups (persistent_grid, msda_mma_common.h).
    for (int run = blockIdx.x; run < n_runs; run += gridDim.x) {
    const int h = run % d.H;
    const int tq = run / d.H;
    const int q_wg0 = (tq % d.q_tiles) * q_per_wg;
    const int b = tq / d.q_tiles;
    const T *slab = value + ((int64_t)b * d.S) * HD + (int64_t)h * d.D;
    const __amdgpu_buffer_rsrc_t rsrc = make_slab_rsrc(slab, ((int64_t)d.S * HD - (int64_t)h * d.D) * (int64_t)sizeof(T));

    // ---- from here on every wave works on its own
    unsigned char *wrec = smem + G::TAB_BYTES + wave * G::WSCR;         // records: [QPW][kChunk] x 32 bytes
    const int qi = lane / LPI, lig = lane % LPI;                          // row-gather role: query of the wave, 16-byte vector
    const int kk = lane & 15;                                             // staging role: sample of the chunk
    const uint32_t lane_off = (uint32_t)(lig * 16);
    const bool pair_ok = ((uintptr_t)loc & (2 * sizeof(T) - 1)) == 0;
    // product roles
    const int am = lane & 15, akb = lane >> 4;                            // A operand: row m, K block
    const int a_q = (D == 128) ? (am >> 2) : (2 * (am >> 2) + ((am >> 1) & 1));       // whose weights this row holds
    const int a_part = (D == 128) ? (am & 3) : (am & 1);                  // 0: hi, 1: lo, (D == 128) 2, 3: zero rows
    const int bG = lane >> 4, be = (lane >> 2) & 3, bc = lane & 3;        // B operand: K block, corner (row of the read), 8-byte piece

    // ---- a wave's work is a sequence of steps: (group of QPW queries) x (chunk of kChunk samples per query).
    // The locations / weights of step s + 1 are requested while step s gathers, so a step opens with
    // arithmetic, not with a global round trip.
    const int q_wg1 = min(d.Nq, q_wg0 + q_per_wg);
    const int n_chunks = (d.K + kChunk - 1) / kChunk;
    const int q_first = q_wg0 + wave * QPW;
    const int n_groups = q_first < q_wg1 ? (q_wg1 - q_first + kMmaWaves * QPW - 1) / (kMmaWaves * QPW) : 0;
    const int n_steps = n_groups * n_chunks;
    constexpr int PS = QPW / 4;                                           // staging passes (one sample per lane each)
    // (kept as the RAW words: converting them would make the wave wait for the loads on the spot)
    uint32_t pf_w0[PS], pf_w1[PS], pf_a[PS];
    // (addresses: one 64-bit base per workgroup, 32-bit sample offsets per lane -- fwd_mma_supported bounds them)
    const uint16_t *loc_wg = reinterpret_cast<const uint16_t *>(loc) + 2 * (((int64_t)b * d.Nq * d.H + h) * d.K);
    const uint16_t *attn_wg = reinterpret_cast<const uint16_t *>(attn) + (((int64_t)b * d.Nq * d.H + h) * d.K);
    const uint32_t q_stride = (uint32_t)d.H * (uint32_t)d.K;              // samples between consecutive queries of this head
    auto prefetch = [&](int step) {
        const int q0 = q_first + (step / n_chunks) * (kMmaWaves * QPW);
        const int k = (step % n_chunks) * kChunk + kk;
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) {
            const int q = q0 + ps * 4 + (lane >> 4);
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
    fill_image<D, true>(tab, img, rsrc, row_bytes, L, d.S, tid);          // channel-permuted (header)
    FPROF(0);                                                             // barrier + image writes (incl. waiting for the slowest wave)
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;

    for (int step = 0; step < n_steps; ++step) {
        // (as msda_fwd_wq.hip, r05f: the wave that is behind the others of its SIMD issues first)
        { const int left = n_steps - 1 - step; if (left >= 3) __builtin_amdgcn_s_setprio(3); else if (left == 2) __builtin_amdgcn_s_setprio(2); else if (left == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
        const int q0 = q_first + (step / n_chunks) * (kMmaWaves * QPW);
        const int chunk = step % n_chunks;
        const int k0 = chunk * kChunk;
        {
            // ---- the chunk's samples by kind (the same for every query: the level decides)
            const int k = k0 + kk;
            const bool k_ok = k < d.K;
            const int l = k_ok ? k / d.P : 0;
            const int Hl = tab[kTabInts * l], Wl = tab[kTabInts * l + 1], lstart = tab[kTabInts * l + 2];
            const int ibase = tab[kTabInts * l + 3], lp = tab[kTabInts * l + 4];
            const bool in_lds = k_ok && ibase >= 0;
            const uint32_t lmask = (uint32_t)__builtin_amdgcn_ballot_w64(in_lds) & 0xffffu;            // by kk (lanes 0..15)
            const uint32_t gmask = (uint32_t)__builtin_amdgcn_ballot_w64(k_ok && !in_lds) & 0xffffu;
            const int n_l = __builtin_popcount(lmask);
            const uint32_t below = (1u << kk) - 1u;
            // records: row-gather samples from the bottom (index = rank among them), LDS samples from the top
            const int ridx = in_lds ? kChunk - 1 - __builtin_popcount(lmask & below) : __builtin_popcount(gmask & below);

            wave_sync();                                                  // the previous step's records are consumed
            // ---- stage: one sample per lane, QPW / 4 passes (the sample's words arrived during the previous step)
            uint32_t live = 0u;                                           // row-gather samples (by kk) that weigh something for some query
#pragma unroll
            for (int ps = 0; ps < PS; ++ps) {
                const int sq = ps * 4 + (lane >> 4);                      // query of the wave this lane stages
                const int q = q0 + sq;
                uint4 r0 = make_uint4(0u, 0u, 0u, 0u), r1 = make_uint4(0u, 0u, 0u, 0u);
                bool weighs = false;
                if (k_ok && q < d.Nq) {
                    // (opaque to the compiler HERE, so that the decode below cannot move up to the loads)
                    asm volatile("" : "+v"(pf_w0[ps]), "+v"(pf_w1[ps]), "+v"(pf_a[ps]));
                    const uint32_t xb = pair_ok ? (pf_w0[ps] & 0xffffu) : pf_w0[ps], yb = pair_ok ? (pf_w0[ps] >> 16) : pf_w1[ps];
                    const float lx = to_f32(__builtin_bit_cast(T, (uint16_t)xb)), ly = to_f32(__builtin_bit_cast(T, (uint16_t)yb));
                    const float a = to_f32(__builtin_bit_cast(T, (uint16_t)pf_a[ps]));
                    const float y = ly * (float)Hl - 0.5f, x = lx * (float)Wl - 0.5f;
                    // strict comparisons: NaN fails, exactly -1 / Hl / Wl fail (cuh:291)
                    const bool inside = (y > -1.f) && (x > -1.f) && (y < (float)Hl) && (x < (float)Wl);
                    const float yf = floorf(y), xf = floorf(x);
                    const int y0 = inside ? (int)yf : 0, x0 = inside ? (int)xf : 0;
                    const float fy = inside ? y - yf : 0.f, fx = inside ? x - xf : 0.f;
                    const float gy = 1.f - fy, gx = 1.f - fx;
                    // a zero attention weight (an image the token cannot see) reads nothing
                    const bool on = inside && a != 0.f;
                    const bool top = y0 >= 0, left = x0 >= 0, bottom = y0 + 1 <= Hl - 1, right = x0 + 1 <= Wl - 1;
                    const bool ok[4] = {on && top && left, on && top && right, on && bottom && left, on && bottom && right};
                    // (a corner that contributes nothing weighs nothing: a sample outside the map is not "live")
                    const float w[4] = {ok[0] ? gy * gx * a : 0.f, ok[1] ? gy * fx * a : 0.f, ok[2] ? fy * gx * a : 0.f, ok[3] ? fy * fx * a : 0.f};
                    if (in_lds) {
                        const int o00 = ibase + y0 * lp + x0 * G::RP;
                        const int off[4] = {o00, o00 + G::RP, o00 + lp, o00 + lp + G::RP};
                        uint32_t hi[4], lo[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            M::split(w[c], hi[c], lo[c]);
                        }
                        r0 = make_uint4(ok[0] ? off[0] : 0, ok[1] ? off[1] : 0, ok[2] ? off[2] : 0, ok[3] ? off[3] : 0);
                        r1 = make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
                    } else {
                        const int p00 = lstart + y0 * Wl + x0;
                        const int row[4] = {p00, p00 + 1, p00 + Wl, p00 + Wl + 1};
                        uint32_t o[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            o[c] = ok[c] ? (uint32_t)row[c] * row_bytes : kOobOffset;
                        r0 = make_uint4(o[0], o[1], o[2], o[3]);
                        r1 = make_uint4(__float_as_uint(w[0]), __float_as_uint(w[1]), __float_as_uint(w[2]), __float_as_uint(w[3]));
                        weighs = ((r1.x | r1.y | r1.z | r1.w) << 1) != 0u;                   // (-0 is zero too)
                    }
                } else if (!in_lds) {
                    r0 = make_uint4(kOobOffset, kOobOffset, kOobOffset, kOobOffset);
                }
                if (k_ok) {
                    uint4 *dst = reinterpret_cast<uint4 *>(wrec + sq * G::QSTRIDE + ridx * 32);
                    dst[0] = r0; dst[1] = r1;
                }
                const unsigned long long bl = __builtin_amdgcn_ballot_w64(weighs);
                live |= (uint32_t)(bl | (bl >> 16) | (bl >> 32) | (bl >> 48)) & 0xffffu;
            }
            wave_sync();
            FPROF(1);                                                     // stage

            // ---- LDS-resident levels on the matrix cores: batches of 8 samples per query
            auto mma_phase = [&]() {
                for (int b8 = 0; 8 * b8 < n_l; ++b8) {
                    // A: this lane's 8 weights = samples r0, r0 + 1 (rank among the chunk's LDS samples) x 4 corners
                    s16x8 A;
                    {
                        const int r0 = 8 * b8 + 2 * akb;
                        const bool on = (D == 64) || a_part < 2;
                        uint2 a0 = make_uint2(0u, 0u), a1 = make_uint2(0u, 0u);
                        const unsigned char *qrec = wrec + a_q * G::QSTRIDE + 16 + 8 * (a_part & 1);
                        if (on && r0 < n_l) a0 = *reinterpret_cast<const uint2 *>(qrec + (kChunk - 1 - r0) * 32);
                        if (on && r0 + 1 < n_l) a1 = *reinterpret_cast<const uint2 *>(qrec + (kChunk - 2 - r0) * 32);
                        const uint4 aw = make_uint4(a0.x, a0.y, a1.x, a1.y);
                        A = __builtin_bit_cast(s16x8, aw);
                    }
#pragma unroll 1
                    for (int j = 0; j < QPW; ++j) {
                        // B rows of query j: this lane supplies 8 bytes of corner `be` of samples 2 * bG, 2 * bG + 1
                        const unsigned char *ad[2];
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const int r = 8 * b8 + 2 * bG + t;
                            uint32_t off = 0u;                                               // the zero row
                            if (r < n_l) off = *reinterpret_cast<const uint32_t *>(wrec + j * G::QSTRIDE + (kChunk - 1 - r) * 32 + 4 * be);
                            ad[t] = img + off + 8 * bc;
                        }
                        const bool mine = qi == j;
                        // four products at a time (independent: they pipeline), then their sums (in the lanes of query j only)
#pragma unroll
                        for (int g0 = 0; g0 < NG; g0 += 4) {
                            f32x4 Tv[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                s16x8 Bv;
#pragma unroll
                                for (int t = 0; t < 2; ++t) {
                                    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(ad[t] + 32 * (g0 + u)));
                                    Bv[4 * t] = v[0]; Bv[4 * t + 1] = v[1]; Bv[4 * t + 2] = v[2]; Bv[4 * t + 3] = v[3];
                                }
                                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                                Tv[u] = M::run(A, Bv, zero);
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int g = g0 + u;
                                if (D == 128) {
                                    // rows 4j (hi), 4j + 1 (lo) sit in the lanes of query j; column n = lane & 15 is channel 8n + g
                                    if (mine) acc[g] += Tv[u][0] + Tv[u][1];
                                    // (rows 2, 3 of the quad are unused, but their registers must stay the product's own:
                                    // overlapped with the next product's result, the products wait for each other)
                                    asm volatile("" :: "v"(Tv[u][2]), "v"(Tv[u][3]));
                                } else {
                                    // D == 64: row quad r = j / 2 spans the lanes of queries 2r and 2r + 1; columns 0..7 are
                                    // accumulator g of lig = n, columns 8..15 accumulator g + 4 of lig = n - 8
                                    const bool odd = j & 1;
                                    const float v = odd ? Tv[u][2] + Tv[u][3] : Tv[u][0] + Tv[u][1];
                                    const float vr = dpp_move<0x128>(v);                     // row_ror:8 : lane n <- lane n ^ 8
                                    acc[g] += mine ? (odd ? vr : v) : 0.f;
                                    acc[g + 4] += mine ? (odd ? v : vr) : 0.f;
                                }
                            }
                        }
                    }
                }
            };

            // ---- row gather of the large levels: counted, software-pipelined walk over the live samples, with the
            // next step's sample words requested and the matrix-core phase run while the first rows are in flight
            {
                unsigned m = (unsigned)__builtin_amdgcn_readfirstlane((int)live);
                const unsigned gm = (unsigned)__builtin_amdgcn_readfirstlane((int)gmask);
                const uint4 *recs = reinterpret_cast<const uint4 *>(wrec + qi * G::QSTRIDE);
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
                    const uint4 ww = recs[2 * gi + 1];                    // (the weights wait in LDS, not in registers)
                    const float w4[4] = {__uint_as_float(ww.x), __uint_as_float(ww.y), __uint_as_float(ww.z), __uint_as_float(ww.w)};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float v[VEC];
                        V::unpack(raw[c], v);
#pragma unroll
                        for (int i = 0; i < VEC; ++i) acc[i] = fmaf(w4[c], v[i], acc[i]);
                    }
#pragma unroll
                    for (int i = 0; i < VEC; ++i) asm volatile("" : "+v"(acc[i]));
                    __builtin_amdgcn_sched_barrier(0);
                };
                const int n_live = __builtin_popcount(m);
                if (n_live >= 4 && (n_live & 3) == 0) {
                    uint4 r0[4], r1[4], r2[4], r3[4];
                    int g0, g1, g2, g3;
                    issue(r0, g0); issue(r1, g1); issue(r2, g2); issue(r3, g3);
                    prefetch(step + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    FPROF(2);                                             // first issues + next step's requests
                    mma_phase();
                    __builtin_amdgcn_sched_barrier(0);
                    FPROF(3);                                             // matrix-core phase
                    for (int i = 4; i < n_live; i += 4) {
                        consume(r0, g0); issue(r0, g0);
                        consume(r1, g1); issue(r1, g1);
                        consume(r2, g2); issue(r2, g2);
                        consume(r3, g3); issue(r3, g3);
                    }
                    consume(r0, g0); consume(r1, g1); consume(r2, g2); consume(r3, g3);
                    FPROF(4);                                             // gather loop
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
                    FPROF(5);                                             // the general path, whole
                }
            }
        }
        if (chunk == n_chunks - 1) {
            const int q = q0 + qi;
            if (q < d.Nq) {
                // (a non-finite sum: the lane's 8 channels again with the reference's arithmetic -- the products carry the
                //  weights as hi + lo parts, and Inf x hi + Inf x lo is NaN where the reference has Inf; round 5)
                float nf = acc[0] * 0.f;
#pragma unroll
                for (int i = 1; i < VEC; ++i) nf = fmaf(acc[i], 0.f, nf);
                if (nf != nf) {
                    const uint32_t s0q = (uint32_t)q * q_stride;
                    exact_lane8<T>(tab, rsrc, row_bytes, loc_wg + 2 * (size_t)s0q, attn_wg + (size_t)s0q, d.K, d.P, lane_off, acc);
                }
                T *o = out + (((int64_t)b * d.Nq + q) * d.H + h) * d.D + lig * VEC;
                store16_stream(o, V::pack(acc));             // (the output is not read again in the step: r03j)
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
            FPROF(6);                                                     // store
        }
    }
    FPROF_COUNT(7, n_steps);
    }   // runs
    FPROF_FLUSH();
}

// ---------------------------------------------------------------- launcher
template <typename T, int D>
static hipError_t launch_mma(const void *value, const int64_t *shapes, const int64_t *start,
                             const void *loc, const void *attn, void *out, Dims d, hipStream_t st)
{
    typedef MmaGeom<D> G;
    static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fwd_mma<T, D>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal);
    if (once != hipSuccess) return once;
    const int env_kb = knob_int(K_FWD_MMA_LDS_KB, 0);      // tuning / debugging
    const int lds_total = env_kb > 0 ? std::min(kLdsTotal, std::max(G::IMG0 + 1024, env_kb * 1024)) : kLdsTotal;
    // queries per workgroup: the image fill (up to IMG_BUDGET bytes through the texture path) is paid per workgroup,
    // so runs are long; but the grid should still be a few workgroups per CU for the tail
    const int env_q = knob_int(K_FWD_MMA_QPW, 0);
    const int unit = kMmaWaves * G::QPW;
    const int q_per_wg = pick_queries_per_run(d, unit, env_q);       // (256, or shorter runs for few queries: msda_mma_common.h)
    d.q_tiles = (d.Nq + q_per_wg - 1) / q_per_wg;
    const int64_t runs = (int64_t)d.B * d.q_tiles * d.H;
    if (runs > 0x7fffffffLL) return hipErrorInvalidValue;
    const int grid = (int)persistent_grid(runs, d.H);                       // (one workgroup per CU when there are many runs, else = runs)
    hipLaunchKernelGGL((msda_fwd_mma<T, D>), dim3((unsigned)grid), dim3(kMmaThreads), lds_total, st,
                       (const T *)value, shapes, start, (const T *)loc, (const T *)attn, (T *)out, d, q_per_wg,
                       lds_total - G::IMG0, (int)runs);
    return hipGetLastError();
}

bool fwd_mma_supported(int dtype, const Dims &d)
{
    if (dtype != 1 && dtype != 2) return false;
    if (d.D != 128 && d.D != 64) return false;
    if (d.L > kMmaMaxLevels || d.K <= 0) return false;
    if ((int64_t)d.Nq * d.H * d.K >= (1LL << 30)) return false;            // 32-bit sample offsets inside a (b, h) slab
    return (int64_t)d.S * d.H * d.D * 2 <= kMaxSlabBytes;
}

bool fwd_mma_applies(int dtype, const Dims &d)
{
    const char *algo = knob_str(K_FWD_ALGO);                           // "vec": never; "mma": whenever the shape allows
    if (algo && algo[0] == 'v') return false;
    if (!fwd_mma_supported(dtype, d)) return false;
    if (algo && algo[0] == 'm') return true;
    // the image is filled once per workgroup: worth it from a few hundred queries per (b, h) on.  Heads of 64
    // channels (the decoders' real geometry: 128-byte rows, 8 queries per wave) measured no faster than the
    // row-gather kernel (profiles/r03_experiments.md, r03c: SD block 364 vs 345 us, LLM 4 images 250 vs 248):
    // they stay on msda_fwd_vec unless asked for (MMFS_FWD_LDS_LEVELS / MMFS_FWD_ALGO=mma)
    // (r03ac: what matters is the samples a workgroup sees per image fill -- the reference's speed-test shape, 128 queries
    // of 128 samples, runs 132 -> 91 us this way; 128 queries of 16 samples do not pay)
    return d.D == 128 && d.Nq >= 64 && (int64_t)d.Nq * d.K >= 4096 && enough_runs(d);       // (round 5: + two runs per CU, r05ac)
}

hipError_t forward_mma(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                       const void *loc, const void *attn, void *out, const Dims &d, hipStream_t st)
{
    if (dtype == 1) {
        if (d.D == 128) return launch_mma<half_t, 128>(value, shapes, start, loc, attn, out, d, st);
        return launch_mma<half_t, 64>(value, shapes, start, loc, attn, out, d, st);
    }
    if (d.D == 128) return launch_mma<bf16_t, 128>(value, shapes, start, loc, attn, out, d, st);
    return launch_mma<bf16_t, 64>(value, shapes, start, loc, attn, out, d, st);
}

}  // namespace mmfs

#ifdef MMFS_PROFILE_FWD
extern "C" int mmfs_debug_fwd_profile(unsigned long long *out, int reset)
{
    static unsigned long long host[mmfs::kFProfSlots * 8];
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(mmfs::g_fwd_prof), sizeof(host));
    for (int i = 0; i < 8; ++i) out[i] = 0;
    for (int s = 0; s < mmfs::kFProfSlots; ++s)
        for (int i = 0; i < 8; ++i) out[i] += host[s * 8 + i];
    if (e == hipSuccess && reset) {
        for (auto &v : host) v = 0;
        e = hipMemcpyToSymbol(HIP_SYMBOL(mmfs::g_fwd_prof), host, sizeof(host));
    }
    return (int)e;
}
#endif
