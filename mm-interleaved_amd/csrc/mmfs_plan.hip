// mmfs_plan.hip -- the MMFS "sampling plan" fused into one kernel each way.
//
// Between its Linear layers and the deformable-attention op the reference runs ~20 elementwise /
// reduction kernels over [N, Lq, H, n*L, P(+1)] tensors
// (mm_interleaved/models/utils/ops/modules/mmfs.py:181-265): add the per-image relative-position
// term, scale offsets per level, add the visibility penalty, overwrite the sink logit, softmax
// over all n*L*(P+1) logits, split points / sinks, divide offsets by the level extent, add the
// reference point, cast.  With the two linear heads rewritten as  head(Wq) + table[relpos]
// (mmfs_amd/modules/mmfs.py) everything after the GEMMs is a per-(sample, query, head) function
// of two small vectors and two table rows; this file evaluates it in one pass, in fp32, from
//     off_q  [N, Lq, H, P, 2]      = sampling_offsets(q')          (bias included)
//     att_q  [N, Lq, H, L, P]      = attention_weights(q'), point columns only (the sink column
//                                    is a constant, mmfs.py:225, and is never computed)
//     off_tab[M, H, P, 2], att_tab[M, H, L, P]     = head.weight @ query_relpos rows
//     relpos [N, Lr, n] int64      (0 = image not visible; Lr = 1 or Lq)
// to
//     loc  [N, Lq, H, n*L, P, 2],  attn [N, Lq, H, n*L, P],  sink [N, Lq, H] (sum of sink weights).
//
// One lane owns one (image, level) row of P logits / P locations (a 16-byte vector for bf16 P=8);
// the 2^k >= n*L lanes of an item reduce with wave shuffles.  The backward walks runs of consecutive
// queries and keeps the gradient of the lane's table row in registers (the relative position of
// an image changes at most once per image along the sequence), flushing it with a few float atomics.
#include "../../include/mmfs_msda.h"
#include "msda_env.h"
#include "msda_device.h"
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <math.h>
#include <type_traits>

namespace mmfs {

namespace {

constexpr int kThreads = 256;
constexpr int kQueryRun = 8;            // backward: consecutive queries a lane group walks (table grads in registers)

struct PlanDims {
    int N, Lq, H, L, P, n, M, Lr, Nr;
    int ld_off, ld_att;     // elements between the rows of two tokens in off_q / att_q (H*2P / H*L*P when packed)
    int ld_toff, ld_tatt;   // the same for the two tables' rows (plan kernels; the sampler's tables are packed)
};

// lane group: G = 2^k lanes, one per (image, level) row of P points; reductions stay inside it
template <int G> __device__ __forceinline__ float group_max(float v)
{
#pragma unroll
    for (int o = G / 2; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
template <int G> __device__ __forceinline__ float group_add(float v)
{
#pragma unroll
    for (int o = G / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// P elements of T <-> floats, as one aligned vector access (P * sizeof(T) is 8, 16 or 32 bytes)
template <typename T, int P> __device__ __forceinline__ void load_row(const T *p, float (&o)[P])
{
    T tmp[P];
    __builtin_memcpy(tmp, __builtin_assume_aligned(p, P * sizeof(T) >= 16 ? 16 : P * sizeof(T)), P * sizeof(T));
#pragma unroll
    for (int i = 0; i < P; ++i) o[i] = to_f32(tmp[i]);
}
template <typename T, int P> __device__ __forceinline__ void store_row(T *p, const float (&v)[P])
{
    T tmp[P];
#pragma unroll
    for (int i = 0; i < P; ++i) tmp[i] = (T)v[i];
    __builtin_memcpy(__builtin_assume_aligned(p, P * sizeof(T) >= 16 ? 16 : P * sizeof(T)), tmp, P * sizeof(T));
}

// Forward: lane <-> one (item, image, level) row of P logits; G lanes (>= n*L) form an item.
//   att_q [N, Lq, H, L, P]  att_tab [M, H, L, P]   (the dead sink column is not computed at all)
template <typename T, int P, int G>
__global__ void __launch_bounds__(kThreads)
plan_forward_kernel(const T *__restrict__ off_q, const T *__restrict__ att_q,
                    const T *__restrict__ off_tab, const T *__restrict__ att_tab,
                    const int64_t *__restrict__ relpos, const float *__restrict__ ref,
                    const int64_t *__restrict__ shapes, const float *__restrict__ ratios,
                    T *__restrict__ loc, T *__restrict__ attn, float *__restrict__ sink, const PlanDims d)
{
    constexpr int IPB = kThreads / G;                       // items per workgroup
    const int gl = threadIdx.x % G;                         // row of the item: kl = k*L + l
    const int64_t item = (int64_t)blockIdx.x * IPB + threadIdx.x / G;     // (nb*Lq + q)*H + h
    const int nL = d.n * d.L;
    const bool item_ok = item < (int64_t)d.N * d.Lq * d.H;
    const bool act = item_ok && gl < nL;
    const int64_t it = item_ok ? item : 0;
    const int h = (int)(it % d.H);
    const int64_t nq = it / d.H;
    const int q = (int)(nq % d.Lq);
    const int nb = (int)(nq / d.Lq);
    const int k = act ? gl / d.L : 0, l = act ? gl % d.L : 0;
    const int64_t r = relpos[((int64_t)nb * d.Lr + (d.Lr == 1 ? 0 : q)) * d.n + k];
    const float sink_logit = -logf((float)nL);

    float lg[P];
    float m = sink_logit;
    if (act) {
        float a[P], t[P];
        load_row<T, P>(att_q + nq * d.ld_att + (h * d.L + l) * P, a);
        load_row<T, P>(att_tab + r * d.ld_tatt + (h * d.L + l) * P, t);
        const float pen = r == 0 ? -10000.f : 0.f;          // image not visible (mmfs.py:203-218)
#pragma unroll
        for (int p = 0; p < P; ++p) { lg[p] = a[p] + t[p] + pen; m = fmaxf(m, lg[p]); }
    } else {
#pragma unroll
        for (int p = 0; p < P; ++p) lg[p] = -INFINITY;
    }
    m = group_max<G>(m);
    float z = act ? __expf(sink_logit - m) : 0.f;           // this row's sink slot (mmfs.py:225)
    const float my_sink = z;
#pragma unroll
    for (int p = 0; p < P; ++p) { lg[p] = __expf(lg[p] - m); z += lg[p]; }
    z = group_add<G>(z);
    const float inv = 1.f / z;
    const float sink_sum = group_add<G>(my_sink) * inv;
    if (item_ok && gl == 0) sink[item] = sink_sum;
    if (!act) return;
#pragma unroll
    for (int p = 0; p < P; ++p) { lg[p] *= inv; asm volatile("" : "+v"(lg[p])); }       // (fp32 product, then rounded: see below)
    const int64_t row = item * nL + gl;
    store_row<T, P>(attn + row * P, lg);
    // locations: ref + (offset_q + offset_table) * ratio_l / (W, H)        (mmfs.py:193-198, 243-250)
    float oq[2 * P], ot[2 * P], xy[2 * P];
    load_row<T, 2 * P>(off_q + nq * d.ld_off + h * 2 * P, oq);
    load_row<T, 2 * P>(off_tab + r * d.ld_toff + h * 2 * P, ot);
    const float rx = ref[((int64_t)(d.Nr == 1 ? 0 : nb) * d.Lq + q) * 2];
    const float ry = ref[((int64_t)(d.Nr == 1 ? 0 : nb) * d.Lq + q) * 2 + 1];
    const float sx = ratios[l] / (float)shapes[2 * gl + 1], sy = ratios[l] / (float)shapes[2 * gl];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        // An fp32 result, THEN rounded to the storage type -- in both kernels that evaluate the plan (here and
        // mmfs_sample_fwd).  For fp16 storage the compiler folds ``(half)fma(a, b, c)`` into v_fma_mixlo_f16, which rounds
        // the exact sum ONCE; with offsets that are halves and scales that are powers of two the fp32 result often sits
        // exactly on a half-precision tie, where the two roundings part (a golden: one location in 4608 -> two outputs one
        // unit apart between the fused sampler and plan + op).  The opaque register keeps the fp32 rounding.
        xy[2 * p] = fmaf(oq[2 * p] + ot[2 * p], sx, rx);
        xy[2 * p + 1] = fmaf(oq[2 * p + 1] + ot[2 * p + 1], sy, ry);
        asm volatile("" : "+v"(xy[2 * p]), "+v"(xy[2 * p + 1]));
    }
    store_row<T, 2 * P>(loc + row * 2 * P, xy);
}

// Backward: same lane <-> row map; a lane group walks kQueryRun consecutive queries of one
// (sample, head) and keeps the gradients of ITS table row (image k's relative position) in
// registers, flushing them with P + 2P float atomics when the relative position changes or the
// run ends.  d_att_q / d_off_q need the sum over the images of a level / over the levels of an
// image: done through a per-group LDS slab.
// (OT: the type the query-side gradients are stored in -- fp32, or the storage type for a caller that casts them anyway)
template <typename T, typename OT, int P, int G>
__global__ void __launch_bounds__(kThreads)
plan_backward_kernel(const T *__restrict__ grad_loc, const T *__restrict__ grad_attn,
                     const float *__restrict__ grad_sink, const T *__restrict__ attn,
                     const float *__restrict__ sink, const int64_t *__restrict__ relpos,
                     const int64_t *__restrict__ shapes, const float *__restrict__ ratios,
                     OT *__restrict__ d_off_q, OT *__restrict__ d_att_q,
                     float *__restrict__ d_off_tab, float *__restrict__ d_att_tab, const PlanDims d)
{
    constexpr int GPB = kThreads / G;                       // lane groups per workgroup
    __shared__ float slab[GPB][G][3 * P];                   // per row: dlogit[P] | doff[2P]
    __shared__ long long seg_key[GPB][G];                   // (sample*H + head, relpos) of the row's table grads
    const int grp = threadIdx.x / G, gl = threadIdx.x % G;
    const int q_runs = (d.Lq + kQueryRun - 1) / kQueryRun;
    const int64_t unit = (int64_t)blockIdx.x * GPB + grp;   // (nb*H + h)*q_runs + run
    const int nL = d.n * d.L;
    const bool unit_ok = unit < (int64_t)d.N * d.H * q_runs;
    const int64_t u = unit_ok ? unit : 0;
    const int run = (int)(u % q_runs);
    const int h = (int)((u / q_runs) % d.H);
    const int nb = (int)(u / q_runs / d.H);
    const bool act = unit_ok && gl < nL;
    const int k = act ? gl / d.L : 0, l = act ? gl % d.L : 0;
    const float sx = act ? ratios[l] / (float)shapes[2 * gl + 1] : 0.f;
    const float sy = act ? ratios[l] / (float)shapes[2 * gl] : 0.f;

    float t_att[P], t_off[2 * P];
    int64_t t_rp = -1;
#pragma unroll
    for (int p = 0; p < P; ++p) { t_att[p] = 0.f; t_off[2 * p] = 0.f; t_off[2 * p + 1] = 0.f; }
    auto flush = [&]() {
        if (t_rp < 0) return;
        float *ta = d_att_tab + t_rp * d.ld_tatt + (h * d.L + l) * P;
        float *to = d_off_tab + t_rp * d.ld_toff + h * 2 * P;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            __hip_atomic_fetch_add(ta + p, t_att[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(to + 2 * p, t_off[2 * p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(to + 2 * p + 1, t_off[2 * p + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            t_att[p] = 0.f; t_off[2 * p] = 0.f; t_off[2 * p + 1] = 0.f;
        }
    };

    const int q_lo = run * kQueryRun, q_hi = min(d.Lq, q_lo + kQueryRun);
    for (int q = q_lo; q < q_hi; ++q) {
        const int64_t item = ((int64_t)nb * d.Lq + q) * d.H + h;
        float w[P], g[P], dl[P], dof[2 * P];
        float part = 0.f;
        int64_t r = 0;
        if (act) {
            r = relpos[((int64_t)nb * d.Lr + (d.Lr == 1 ? 0 : q)) * d.n + k];
            if (r != t_rp) { flush(); t_rp = r; }
            const int64_t row = item * nL + gl;
            load_row<T, P>(attn + row * P, w);
            load_row<T, P>(grad_attn + row * P, g);
#pragma unroll
            for (int p = 0; p < P; ++p) part += w[p] * g[p];
        }
        // softmax backward: dlogit_i = w_i (g_i - sum_j w_j g_j), the sinks' share included
        float dot = group_add<G>(part);
        if (unit_ok && grad_sink) dot += grad_sink[item] * sink[item];
        if (act) {
            float gxy[2 * P];
            load_row<T, 2 * P>(grad_loc + (item * nL + gl) * 2 * P, gxy);
#pragma unroll
            for (int p = 0; p < P; ++p) {
                dl[p] = w[p] * (g[p] - dot);
                dof[2 * p] = gxy[2 * p] * sx;
                dof[2 * p + 1] = gxy[2 * p + 1] * sy;
                t_att[p] += dl[p];
                t_off[2 * p] += dof[2 * p];
                t_off[2 * p + 1] += dof[2 * p + 1];
                slab[grp][gl][p] = dl[p];
                slab[grp][gl][P + 2 * p] = dof[2 * p];
                slab[grp][gl][P + 2 * p + 1] = dof[2 * p + 1];
            }
        }
        __builtin_amdgcn_wave_barrier();            // a group never spans waves (G <= 64)
        // d_att_q[item, l, :] = sum over images k of dlogit[k*L + l, :]  (rows l < L do it)
        if (unit_ok && gl < d.L) {
            float acc[P];
#pragma unroll
            for (int p = 0; p < P; ++p) acc[p] = 0.f;
            for (int kk = 0; kk < d.n; ++kk)
#pragma unroll
                for (int p = 0; p < P; ++p) acc[p] += slab[grp][kk * d.L + gl][p];
            store_row<OT, P>(d_att_q + (item / d.H) * d.ld_att + (h * d.L + gl) * P, acc);
        }
        // d_off_q[item, :, :] = sum over all rows (images and levels) of doff  (row 0 does it)
        if (unit_ok && gl == 0) {
            float acc[2 * P];
#pragma unroll
            for (int p = 0; p < 2 * P; ++p) acc[p] = 0.f;
            for (int rr = 0; rr < nL; ++rr)
#pragma unroll
                for (int p = 0; p < 2 * P; ++p) acc[p] += slab[grp][rr][P + p];
            store_row<OT, 2 * P>(d_off_q + (item / d.H) * d.ld_off + h * 2 * P, acc);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // ---- the table rows are shared by every query of a (sample, head): same-address atomics
    //      serialise in L2 (measured: 6.6 ms when every lane group flushed its own), so the groups
    //      of the workgroup -- consecutive query runs, normally the same table row -- are summed
    //      through LDS first and only the first group of each equal-key segment issues atomics.
    __syncthreads();
#pragma unroll
    for (int p = 0; p < P; ++p) {
        slab[grp][gl][p] = t_att[p];
        slab[grp][gl][P + 2 * p] = t_off[2 * p];
        slab[grp][gl][P + 2 * p + 1] = t_off[2 * p + 1];
    }
    seg_key[grp][gl] = (act && t_rp >= 0) ? (((long long)nb * d.H + h) << 20 | (long long)t_rp) : -1LL - grp;
    __syncthreads();
    if (act && t_rp >= 0 && (grp == 0 || seg_key[grp - 1][gl] != seg_key[grp][gl])) {
        const long long key = seg_key[grp][gl];
        for (int g2 = grp + 1; g2 < GPB && seg_key[g2][gl] == key; ++g2) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                t_att[p] += slab[g2][gl][p];
                t_off[2 * p] += slab[g2][gl][P + 2 * p];
                t_off[2 * p + 1] += slab[g2][gl][P + 2 * p + 1];
            }
        }
        flush();
    }
}


// ---------------------------------------------------------------- plan -> sampler, one kernel (forward / inference)
// SURVEY.md 8f N1, second half: the sampling plan feeds the sampler directly -- the locations and weights
// [N, Lq, H, n*L, P(, 2)] never exist in HBM.  This is msda_fwd_vec (csrc/msda_fwd.hip) with its staging
// replaced: instead of reading loc / attn it evaluates the plan for its 256 / LPI queries of one (sample,
// head) from the heads' outputs and table rows, in the SAME arithmetic as plan_forward_kernel (same lane
// groups, same reduction order, results rounded to the storage type before use), so the output is bit-
// identical to plan_forward + ms_deform_attn_forward.  Forward only: the backward needs loc / attn as
// tensors (the sort scans them, the plan backward reads the weights), so training keeps the two kernels.
constexpr int kSampleRecs = 512;        // tap records staged per chunk (as msda_fwd_vec)
constexpr int kSampleUnroll = 2;

__device__ __forceinline__ float rgroup_max(float v, int G)
{
    for (int o = G / 2; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float rgroup_add(float v, int G)
{
    for (int o = G / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename T, int LPI, int P>
__global__ void __launch_bounds__(kThreads)
mmfs_sample_fwd(const T *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ start,
                const T *__restrict__ off_q, const T *__restrict__ att_q,
                const T *__restrict__ off_tab, const T *__restrict__ att_tab,
                const int64_t *__restrict__ relpos, const float *__restrict__ ref, const float *__restrict__ ratios,
                T *__restrict__ out, float *__restrict__ sink, const Dims d, const PlanDims pd, const int G,
                const T *__restrict__ token)
{
    typedef Vec16<T> V;
    constexpr int VEC = V::N;
    constexpr int QPB = kThreads / LPI;
    constexpr int KC = (kSampleRecs / QPB) > P ? (kSampleRecs / QPB) : P;      // samples per query per chunk: whole rows of P
    __shared__ float ssum[QPB];                           // the queries' summed sink weights (for the ignore-token term)
    static_assert(KC % P == 0 && KC % kSampleUnroll == 0, "chunks hold whole rows of P points");
    constexpr int STRIDE = 2 * KC + 1;
    __shared__ uint4 lds[QPB * STRIDE];
    __shared__ LevelLds levels;
    __shared__ float2 stat[QPB];                          // softmax max and 1 / sum per query of the tile
    __shared__ float4 plan[QPB * KC];                     // {x, y, weight} of the chunk's samples
    constexpr bool kPipe = KC <= 64;                      // live-tap mask + pipelined walk, as in msda_fwd_vec
    constexpr int QPW = 64 / LPI > 0 ? 64 / LPI : 1;
    __shared__ unsigned long long live[kThreads / 64];

    const BlockCoord bc = block_coord(d, QPB);
    const int tid = threadIdx.x;
    const int qi = tid / LPI, lig = tid % LPI;
    const int q = bc.q0 + qi;
    const bool q_ok = q < d.Nq;
    const int nL = d.L;                                   // = n * levels per image
    const int64_t HD = (int64_t)d.H * d.D;
    const T *slab = value + ((int64_t)bc.b * d.S) * HD + (int64_t)bc.h * d.D;
    const uint32_t row_bytes = (uint32_t)(HD * sizeof(T));
    const uint32_t lane_off = (uint32_t)(lig * 16);
    const __amdgpu_buffer_rsrc_t rsrc = make_slab_rsrc(slab, ((int64_t)d.S * HD - (int64_t)bc.h * d.D) * (int64_t)sizeof(T));
    levels.load(shapes, start, nL, tid, kThreads);
    if (kPipe && tid < kThreads / 64) live[tid] = 0ull;
    const float sink_logit = -logf((float)nL);

    // logits of row gl (= image k, level l) of the item (bc.b, sq, bc.h), as plan_forward_kernel forms them
    auto row_logits = [&](int sq, int gl, float (&lg)[P]) {
        const int64_t tk = (int64_t)bc.b * pd.Lq + sq;
        const int k = gl / pd.L, l = gl % pd.L;
        const int64_t r = relpos[((int64_t)bc.b * pd.Lr + (pd.Lr == 1 ? 0 : sq)) * pd.n + k];
        float a[P], t[P];
        load_row<T, P>(att_q + tk * pd.ld_att + (bc.h * pd.L + l) * P, a);
        load_row<T, P>(att_tab + ((r * pd.H + bc.h) * pd.L + l) * P, t);
        const float pen = r == 0 ? -10000.f : 0.f;
#pragma unroll
        for (int p = 0; p < P; ++p) lg[p] = a[p] + t[p] + pen;
        return r;
    };

    // ---- softmax statistics of the tile's queries (lane group of G per query, like the plan kernel)
    for (int base = 0; base < QPB * G; base += kThreads) {
        if (base + (tid & ~63) >= QPB * G) continue;      // (whole waves only: the groups shuffle)
        const int s = base + tid, rq = s / G, gl = s % G;
        const int sq = bc.q0 + rq;
        const bool item_ok = rq < QPB && sq < d.Nq;
        const bool act = item_ok && gl < nL;
        float lg[P];
        float m = sink_logit;
        if (act) {
            row_logits(sq, gl, lg);
#pragma unroll
            for (int p = 0; p < P; ++p) m = fmaxf(m, lg[p]);
        } else {
#pragma unroll
            for (int p = 0; p < P; ++p) lg[p] = -INFINITY;
        }
        m = rgroup_max(m, G);
        float z = act ? __expf(sink_logit - m) : 0.f;
        const float my_sink = z;
#pragma unroll
        for (int p = 0; p < P; ++p) { lg[p] = __expf(lg[p] - m); z += lg[p]; }
        z = rgroup_add(z, G);
        const float inv = 1.f / z;
        const float sink_sum = rgroup_add(my_sink, G) * inv;
        if (item_ok && gl == 0) {
            stat[rq] = make_float2(m, inv);
            ssum[rq] = sink_sum;
            if (sink != nullptr) sink[((int64_t)bc.b * pd.Lq + sq) * pd.H + bc.h] = sink_sum;
        }
    }
    __syncthreads();

    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;

    for (int k0 = 0; k0 < d.K; k0 += KC) {
        const int kc = min(KC, d.K - k0);                 // a multiple of P (and of the unroll)
        const int rows = kc / P;
        if (k0 > 0) __syncthreads();
        // ---- stage, step 1: one lane per (query, row): the row's P weights and locations, rounded to the
        // storage type like the tensors of the two-kernel path, parked in LDS
        for (int s = tid; s < QPB * rows; s += kThreads) {
            const int rq = s / rows, rr = s - rq * rows, gl = k0 / P + rr;
            const int sq = bc.q0 + rq;
            float4 *dst = &plan[rq * KC + rr * P];
            if (sq >= d.Nq) {
#pragma unroll
                for (int p = 0; p < P; ++p) dst[p] = make_float4(-8.f, -8.f, 0.f, 0.f);      // outside every map, weight 0
                continue;
            }
            float lg[P];
            const int64_t r = row_logits(sq, gl, lg);
            const float2 st2 = stat[rq];
            const int l = gl % pd.L;
            float oq[2 * P], ot[2 * P];
            load_row<T, 2 * P>(off_q + ((int64_t)bc.b * pd.Lq + sq) * pd.ld_off + bc.h * 2 * P, oq);
            load_row<T, 2 * P>(off_tab + (r * pd.H + bc.h) * 2 * P, ot);
            const float rx = ref[((int64_t)(pd.Nr == 1 ? 0 : bc.b) * pd.Lq + sq) * 2];
            const float ry = ref[((int64_t)(pd.Nr == 1 ? 0 : bc.b) * pd.Lq + sq) * 2 + 1];
            int Hl, Wl, lstart;
            levels.get(shapes, start, gl, Hl, Wl, lstart);
            const float sx = ratios[l] / (float)Wl, sy = ratios[l] / (float)Hl;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                float wgt = __expf(lg[p] - st2.x);
                wgt *= st2.y;
                // (as plan_forward_kernel: fp32 results behind an opaque register, then rounded to the storage type)
                float lx = fmaf(oq[2 * p] + ot[2 * p], sx, rx), ly = fmaf(oq[2 * p + 1] + ot[2 * p + 1], sy, ry);
                asm volatile("" : "+v"(lx), "+v"(ly), "+v"(wgt));
                dst[p] = make_float4(to_f32((T)lx), to_f32((T)ly), to_f32((T)wgt), 0.f);
            }
        }
        __syncthreads();
        // ---- stage, step 2: one lane per sample: location -> tap record (as msda_fwd_vec does from its tensors)
        for (int r = tid; r < QPB * kc; r += kThreads) {
            const int rq = r / kc, kk = r - rq * kc, gl = (k0 + kk) / P;
            const float4 pl = plan[rq * KC + kk];
            const float a = pl.z;
            int Hl, Wl, lstart;
            levels.get(shapes, start, gl, Hl, Wl, lstart);
            const Tap<float> t = locate<float>(pl.x, pl.y, Hl, Wl, lstart);
            const float gy = 1.f - t.fy, gx = 1.f - t.fx;
            uint32_t off[4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                off[c] = (a != 0.f && t.row[c] >= 0) ? (uint32_t)t.row[c] * row_bytes : kOobOffset;
            uint4 *dst = &lds[rq * STRIDE + 2 * kk];
            const uint4 ww = make_uint4(__float_as_uint(gy * gx * a), __float_as_uint(gy * t.fx * a),
                                        __float_as_uint(t.fy * gx * a), __float_as_uint(t.fy * t.fx * a));
            dst[0] = make_uint4(off[0], off[1], off[2], off[3]);
            dst[1] = ww;
            if (kPipe && ((ww.x | ww.y | ww.z | ww.w) << 1) != 0u) atomicOr(&live[rq / QPW], 1ull << kk);
        }
        __syncthreads();
        if (kPipe) {
            const int wv = tid >> 6;
            const unsigned long long mraw = live[wv];
            unsigned long long m = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(mraw >> 32)) << 32) |
                                   (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)mraw);
            if ((tid & 63) == 0) live[wv] = 0ull;
            const uint4 *recs = &lds[qi * STRIDE];
            uint4 rawA[4], rawB[4], wA, wB;
            auto issue = [&](uint4 (&raw)[4], uint4 &wq) {
                const int kk = __builtin_ctzll(m);
                m &= m - 1ull;
                const uint4 rr = recs[2 * kk];
                wq = recs[2 * kk + 1];
                raw[0] = buffer_load16(rsrc, rr.x + lane_off);
                raw[1] = buffer_load16(rsrc, rr.y + lane_off);
                raw[2] = buffer_load16(rsrc, rr.z + lane_off);
                raw[3] = buffer_load16(rsrc, rr.w + lane_off);
                __builtin_amdgcn_sched_barrier(0);
            };
            auto consume = [&](const uint4 (&raw)[4], const uint4 &wq) {
                const float w4[4] = {__uint_as_float(wq.x), __uint_as_float(wq.y), __uint_as_float(wq.z), __uint_as_float(wq.w)};
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
            const int n_live = __builtin_popcountll(m);
            if (n_live & 1) { issue(rawA, wA); consume(rawA, wA); }
            if (n_live >= 2) {
                issue(rawA, wA);
                for (int i = 2; i < n_live - 1; i += 2) {
                    issue(rawB, wB);
                    consume(rawA, wA);
                    issue(rawA, wA);
                    consume(rawB, wB);
                }
                issue(rawB, wB);
                consume(rawA, wA);
                consume(rawB, wB);
            }
        } else
        // ---- gather (as msda_fwd_vec: 2 taps = 8 row reads in flight per lane, whole-wave skip of zero weights)
        if (q_ok) {
            const uint4 *recs = &lds[qi * STRIDE];
            for (int kk = 0; kk < kc; kk += kSampleUnroll) {
                uint4 raw[kSampleUnroll][4];
                float w[kSampleUnroll][4];
                uint4 rrs[kSampleUnroll];
                uint32_t any_w = 0u;
#pragma unroll
                for (int u = 0; u < kSampleUnroll; ++u) {
                    rrs[u] = recs[2 * (kk + u)];
                    const uint4 ww = recs[2 * (kk + u) + 1];
                    any_w |= (ww.x | ww.y | ww.z | ww.w) << 1;
                    w[u][0] = __uint_as_float(ww.x); w[u][1] = __uint_as_float(ww.y);
                    w[u][2] = __uint_as_float(ww.z); w[u][3] = __uint_as_float(ww.w);
                }
                if (__builtin_amdgcn_ballot_w64(any_w != 0u) == 0ull) continue;
#pragma unroll
                for (int u = 0; u < kSampleUnroll; ++u) {
                    raw[u][0] = buffer_load16(rsrc, rrs[u].x + lane_off);
                    raw[u][1] = buffer_load16(rsrc, rrs[u].y + lane_off);
                    raw[u][2] = buffer_load16(rsrc, rrs[u].z + lane_off);
                    raw[u][3] = buffer_load16(rsrc, rrs[u].w + lane_off);
                }
#pragma unroll
                for (int u = 0; u < kSampleUnroll; ++u)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float v[VEC];
                        V::unpack(raw[u][c], v);
#pragma unroll
                        for (int i = 0; i < VEC; ++i) acc[i] = fmaf(w[u][c], v[i], acc[i]);
                    }
            }
        }
    }
    if (q_ok) {
        T *o = out + (((int64_t)bc.b * d.Nq + q) * d.H + bc.h) * d.D + lig * VEC;
        if (token != nullptr) {
            // the sinks' share goes to the ignore token (mmfs.py:236-241, 274): out + token * sink, with the framework
            // statement's roundings -- sampled output, sink weight and product each rounded to the storage type first
            const float sw = to_f32((T)ssum[qi]);
            float tk[VEC];
            V::unpack(*reinterpret_cast<const uint4 *>(token + (int64_t)bc.h * d.D + lig * VEC), tk);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                // (two roundings, as the framework's two kernels: the product must not be contracted into the sum --
                // __fmul_rn / __fadd_rn do not stop this compiler, an opaque register does)
                float prod = tk[i] * sw;
                asm volatile("" : "+v"(prod));
                acc[i] = to_f32((T)acc[i]) + to_f32((T)prod);
            }
        }
        *reinterpret_cast<uint4 *>(o) = V::pack(acc);
    }
}

// ---------------------------------------------------------------- the same for a handful of queries (a decode step)
// One new token per sequence is ONE query per (sample, head), and mmfs_sample_fwd serves it with 8 of a workgroup's
// 256 lanes: softmax statistics (index -> table row: two dependent round trips), barrier, the same rows again for the
// plan, barrier, tap records, barrier, and then the query's 24 samples two at a time -- a dozen dependent round trips,
// 13.9 us per call for 4 tokens (profiles/r03by_decode_kernel_stats.csv; 64-lane workgroups of the
// same kernel: 13.6, r04w).  Here a 64-lane workgroup (one wave: its barriers are free) owns ONE query:
//   A  lane gl < n*L evaluates row gl of the plan ONCE (logits, softmax over the lane group, locations) -- the
//      arithmetic, lane groups and reduction order of plan_forward_kernel, so locations and weights are the same bits;
//   B  lane s evaluates sample s's tap record from A's row (LDS);
//   C  the 64 / LPI lane groups take the samples round-robin, EVERY row load of the query in flight at once, and add
//      their partial sums up in a fixed tree.
// Three dependent global round trips (index, table rows, value rows).  Against the two-kernel path (and mmfs_sample_fwd)
// the same products in fp32, summed in another order: equal within one rounding of the storage type, not bit for bit
// (mmfs_sample_forward_groups tells a caller which kernel a shape takes).
constexpr int kDecodeMaxQ = 8;          // queries per (sample, head) up to which a workgroup per query pays
constexpr int kDecodeMaxK = 256;        // samples per query held as tap records
constexpr int kDecodeBatch = 4;         // samples in flight per lane group (16 row loads per lane)

template <typename T, int LPI, int P>
__global__ void __launch_bounds__(64)
mmfs_sample_decode(const T *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ start,
                   const T *__restrict__ off_q, const T *__restrict__ att_q,
                   const T *__restrict__ off_tab, const T *__restrict__ att_tab,
                   const int64_t *__restrict__ relpos, const float *__restrict__ ref, const float *__restrict__ ratios,
                   T *__restrict__ out, float *__restrict__ sink, const Dims d, const PlanDims pd, const int G,
                   const T *__restrict__ token)
{
    typedef Vec16<T> V;
    constexpr int VEC = V::N;
    constexpr int NSUB = 64 / LPI;
    __shared__ float4 plan[kDecodeMaxK];                  // {x, y, weight} per sample, rounded to the storage type
    __shared__ int rowinfo[64 * 3];                       // Hl, Wl, start of the rows' levels
    __shared__ uint4 recs[2 * kDecodeMaxK];               // per sample: four row offsets, four corner weights
    const int tid = threadIdx.x;
    const int h = blockIdx.x % d.H;
    const int q = (blockIdx.x / d.H) % d.Nq, b = blockIdx.x / d.H / d.Nq;
    const int nL = d.L;
    const int64_t HD = (int64_t)d.H * d.D;
    const T *slab = value + ((int64_t)b * d.S) * HD + (int64_t)h * d.D;
    const uint32_t row_bytes = (uint32_t)(HD * sizeof(T));
    const __amdgpu_buffer_rsrc_t rsrc = make_slab_rsrc(slab, ((int64_t)d.S * HD - (int64_t)h * d.D) * (int64_t)sizeof(T));
    const float sink_logit = -logf((float)nL);
    const int64_t tk = (int64_t)b * pd.Lq + q;

    // ---- A: row gl of the plan (lanes of the first lane group; the others idle through the shuffles)
    float sink_sum;
    {
        const int gl = tid;
        const bool act = gl < nL;
        float lg[P], oq[2 * P], ot[2 * P];
        float m = sink_logit, rx = 0.f, ry = 0.f, sx = 0.f, sy = 0.f;
        if (act) {
            const int k = gl / pd.L, l = gl % pd.L;
            const int64_t r = relpos[((int64_t)b * pd.Lr + (pd.Lr == 1 ? 0 : q)) * pd.n + k];
            float a[P], t[P];
            load_row<T, P>(att_q + tk * pd.ld_att + (h * pd.L + l) * P, a);
            load_row<T, 2 * P>(off_q + tk * pd.ld_off + h * 2 * P, oq);
            rx = ref[((int64_t)(pd.Nr == 1 ? 0 : b) * pd.Lq + q) * 2];
            ry = ref[((int64_t)(pd.Nr == 1 ? 0 : b) * pd.Lq + q) * 2 + 1];
            const int Hl = (int)shapes[2 * gl], Wl = (int)shapes[2 * gl + 1];
            rowinfo[3 * gl] = Hl; rowinfo[3 * gl + 1] = Wl; rowinfo[3 * gl + 2] = (int)start[gl];
            sx = ratios[l] / (float)Wl; sy = ratios[l] / (float)Hl;
            load_row<T, P>(att_tab + ((r * pd.H + h) * pd.L + l) * P, t);
            load_row<T, 2 * P>(off_tab + (r * pd.H + h) * 2 * P, ot);
            const float pen = r == 0 ? -10000.f : 0.f;
#pragma unroll
            for (int p = 0; p < P; ++p) { lg[p] = a[p] + t[p] + pen; m = fmaxf(m, lg[p]); }
        } else {
#pragma unroll
            for (int p = 0; p < P; ++p) lg[p] = -INFINITY;
#pragma unroll
            for (int p = 0; p < 2 * P; ++p) { oq[p] = 0.f; ot[p] = 0.f; }
        }
        m = rgroup_max(m, G);
        float z = act ? __expf(sink_logit - m) : 0.f;
        const float my_sink = z;
#pragma unroll
        for (int p = 0; p < P; ++p) { lg[p] = __expf(lg[p] - m); z += lg[p]; }
        z = rgroup_add(z, G);
        const float inv = 1.f / z;
        sink_sum = rgroup_add(my_sink, G) * inv;
        if (tid == 0 && sink != nullptr) sink[tk * pd.H + h] = sink_sum;
        if (act) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                float wgt = lg[p] * inv;
                // (as plan_forward_kernel: fp32 results behind an opaque register, then rounded to the storage type)
                float lx = fmaf(oq[2 * p] + ot[2 * p], sx, rx), ly = fmaf(oq[2 * p + 1] + ot[2 * p + 1], sy, ry);
                asm volatile("" : "+v"(lx), "+v"(ly), "+v"(wgt));
                plan[gl * P + p] = make_float4(to_f32((T)lx), to_f32((T)ly), to_f32((T)wgt), 0.f);
            }
        }
    }
    sink_sum = __shfl(sink_sum, 0, 64);
    __syncthreads();
    // ---- B: one lane per sample: location -> tap record (as msda_fwd_vec does from its tensors)
    for (int s = tid; s < d.K; s += 64) {
        const int gl = s / P;
        const float4 pl = plan[s];
        const float a = pl.z;
        const Tap<float> t = locate<float>(pl.x, pl.y, rowinfo[3 * gl], rowinfo[3 * gl + 1], rowinfo[3 * gl + 2]);
        const float gy = 1.f - t.fy, gx = 1.f - t.fx;
        uint32_t off[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
            off[c] = (a != 0.f && t.row[c] >= 0) ? (uint32_t)t.row[c] * row_bytes : kOobOffset;
        recs[2 * s] = make_uint4(off[0], off[1], off[2], off[3]);
        recs[2 * s + 1] = make_uint4(__float_as_uint(gy * gx * a), __float_as_uint(gy * t.fx * a),
                                     __float_as_uint(t.fy * gx * a), __float_as_uint(t.fy * t.fx * a));
    }
    __syncthreads();
    // ---- C: lane group `sub` takes samples sub, sub + NSUB, ...; a batch's row loads all leave before the first is used
    const int sub = tid / LPI, lig = tid % LPI;
    const uint32_t lane_off = (uint32_t)(lig * 16);
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    for (int k0 = 0; k0 < d.K; k0 += NSUB * kDecodeBatch) {
        uint4 raw[kDecodeBatch][4], wq[kDecodeBatch];
#pragma unroll
        for (int u = 0; u < kDecodeBatch; ++u) {
            const int kk = k0 + u * NSUB + sub;
            uint4 rr = make_uint4(kOobOffset, kOobOffset, kOobOffset, kOobOffset);
            wq[u] = make_uint4(0u, 0u, 0u, 0u);
            if (kk < d.K) { rr = recs[2 * kk]; wq[u] = recs[2 * kk + 1]; }
            raw[u][0] = buffer_load16(rsrc, rr.x + lane_off);
            raw[u][1] = buffer_load16(rsrc, rr.y + lane_off);
            raw[u][2] = buffer_load16(rsrc, rr.z + lane_off);
            raw[u][3] = buffer_load16(rsrc, rr.w + lane_off);
        }
#pragma unroll
        for (int u = 0; u < kDecodeBatch; ++u) {
            const float w4[4] = {__uint_as_float(wq[u].x), __uint_as_float(wq[u].y), __uint_as_float(wq[u].z), __uint_as_float(wq[u].w)};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float v[VEC];
                V::unpack(raw[u][c], v);
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[i] = fmaf(w4[c], v[i], acc[i]);
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= LPI; o >>= 1)
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] += __shfl_xor(acc[i], o, 64);
    if (sub == 0) {
        T *o = out + (((int64_t)b * d.Nq + q) * d.H + h) * d.D + lig * VEC;
        if (token != nullptr) {
            // (mmfs_sample_fwd's statement of the ignore-token term: sampled output, sink weight and product each rounded first)
            const float sw = to_f32((T)sink_sum);
            float tkn[VEC];
            V::unpack(*reinterpret_cast<const uint4 *>(token + (int64_t)h * d.D + lig * VEC), tkn);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float prod = tkn[i] * sw;
                asm volatile("" : "+v"(prod));
                acc[i] = to_f32((T)acc[i]) + to_f32((T)prod);
            }
        }
        *reinterpret_cast<uint4 *>(o) = V::pack(acc);
    }
}

// lane groups that share a query's samples: 1 = mmfs_sample_fwd (sums in sample order, bit-identical to plan + op)
int sample_groups(int64_t Lq, int64_t K, int64_t nL, int lpi)
{
    const char *e = mmfs::knob_str(mmfs::K_SAMPLE_DECODE);   // (the tests hold both kernels to the same goldens)
    if ((e && e[0] == '0') || Lq > kDecodeMaxQ || K > kDecodeMaxK || nL > 64 || lpi > 32) return 1;
    return 64 / lpi;
}

int esize(int dtype) { return dtype == MMFS_F32 ? 4 : (dtype == MMFS_F16 || dtype == MMFS_BF16) ? 2 : 0; }

int check_dims(int64_t N, int64_t Lq, int64_t H, int64_t L, int64_t P, int64_t n, int64_t M, int64_t Lr,
               int64_t Nr, PlanDims *d)
{
    const int64_t lim = 0x7fffffffLL;
    if (N < 0 || Lq < 0 || H <= 0 || L <= 0 || P <= 0 || n <= 0 || M <= 0) return MMFS_E_DIMS;
    if (N > lim || Lq > lim || H > lim || n > lim || M > lim) return MMFS_E_DIMS;
    if ((Lr != 1 && Lr != Lq) || (Nr != 1 && Nr != N)) return MMFS_E_DIMS;
    if ((P != 4 && P != 8 && P != 16) || n * L > 64) return MMFS_E_UNSUPPORTED;
    if (N * Lq * H > lim) return MMFS_E_DIMS;
    d->N = (int)N; d->Lq = (int)Lq; d->H = (int)H; d->L = (int)L; d->P = (int)P; d->n = (int)n;
    d->M = (int)M; d->Lr = (int)Lr; d->Nr = (int)Nr;
    d->ld_off = d->ld_toff = (int)(H * 2 * P);
    d->ld_att = d->ld_tatt = (int)(H * L * P);
    return MMFS_OK;
}

}  // namespace
}  // namespace mmfs

// dispatch over the points per row P (4, 8, 16) and the lane-group width G (pow2 >= n*L)
template <typename T, int P, typename F>
static int for_group(int nL, F &&f)
{
    if (nL <= 4) return f(std::integral_constant<int, 4>());
    if (nL <= 8) return f(std::integral_constant<int, 8>());
    if (nL <= 16) return f(std::integral_constant<int, 16>());
    if (nL <= 32) return f(std::integral_constant<int, 32>());
    return f(std::integral_constant<int, 64>());
}

extern "C" {

// rows `ld` elements apart (0 = packed, `cols`), vector accesses of `vec` elements of `es` bytes: dimensions and alignment
static int heads_rows(int64_t &ld, int64_t cols, const void *p, int64_t vec, int64_t es)
{
    if (ld == 0) ld = cols;
    if (ld < cols || ld > 0x7fffffffLL) return MMFS_E_DIMS;
    const int64_t a = vec * es < 16 ? vec * es : 16;                   // (load_row / store_row: accesses of at most 16 bytes)
    if ((ld * es) % a || (uintptr_t)p % (uintptr_t)a) return MMFS_E_ALIGN;
    return MMFS_OK;
}

int mmfs_plan_forward(int dtype, const void *off_q, const void *att_q, const void *off_tab,
                      const void *att_tab, const int64_t *relpos, const float *ref,
                      const int64_t *shapes, const float *ratios, void *loc, void *attn, float *sink,
                      int64_t N, int64_t Lq, int64_t H, int64_t L, int64_t P, int64_t n, int64_t M,
                      int64_t Lr, int64_t Nr, void *stream)
{
    return mmfs_plan_forward_heads(dtype, off_q, att_q, 0, 0, off_tab, att_tab, 0, 0, relpos, ref, shapes, ratios, loc, attn,
                                   sink, N, Lq, H, L, P, n, M, Lr, Nr, stream);
}

int mmfs_plan_forward_heads(int dtype, const void *off_q, const void *att_q, int64_t ld_off, int64_t ld_att,
                            const void *off_tab, const void *att_tab, int64_t ld_toff, int64_t ld_tatt,
                            const int64_t *relpos, const float *ref,
                            const int64_t *shapes, const float *ratios, void *loc, void *attn, float *sink,
                            int64_t N, int64_t Lq, int64_t H, int64_t L, int64_t P, int64_t n, int64_t M,
                            int64_t Lr, int64_t Nr, void *stream)
{
    using namespace mmfs;
    const int es = esize(dtype);
    if (!es) return MMFS_E_DTYPE;
    PlanDims d;
    const int rc = check_dims(N, Lq, H, L, P, n, M, Lr, Nr, &d);
    if (rc) return rc;
    const int64_t items = N * Lq * H;
    if (items == 0) return MMFS_OK;
    if (!off_q || !att_q || !off_tab || !att_tab || !relpos || !ref || !shapes || !ratios || !loc || !attn || !sink)
        return MMFS_E_NULLPTR;
    int rr;
    if ((rr = heads_rows(ld_off, H * 2 * P, off_q, 2 * P, es)) || (rr = heads_rows(ld_att, H * L * P, att_q, P, es)) ||
        (rr = heads_rows(ld_toff, H * 2 * P, off_tab, 2 * P, es)) || (rr = heads_rows(ld_tatt, H * L * P, att_tab, P, es)))
        return rr;
    d.ld_off = (int)ld_off; d.ld_att = (int)ld_att; d.ld_toff = (int)ld_toff; d.ld_tatt = (int)ld_tatt;
    hipStream_t st = (hipStream_t)stream;
    auto go = [&](auto tag_t, auto tag_p) {
        typedef decltype(tag_t) T;
        constexpr int PP = decltype(tag_p)::value;
        return for_group<T, PP>(d.n * d.L, [&](auto tag_g) {
            constexpr int G = decltype(tag_g)::value;
            const unsigned blocks = (unsigned)((items + kThreads / G - 1) / (kThreads / G));
            hipLaunchKernelGGL((plan_forward_kernel<T, PP, G>), dim3(blocks), dim3(kThreads), 0, st,
                               (const T *)off_q, (const T *)att_q, (const T *)off_tab, (const T *)att_tab,
                               relpos, ref, shapes, ratios, (T *)loc, (T *)attn, sink, d);
            return (int)hipGetLastError();
        });
    };
    auto by_p = [&](auto tag_t) {
        if (P == 4) return go(tag_t, std::integral_constant<int, 4>());
        if (P == 8) return go(tag_t, std::integral_constant<int, 8>());
        return go(tag_t, std::integral_constant<int, 16>());
    };
    if (dtype == MMFS_F32) return by_p(float());
    if (dtype == MMFS_F16) return by_p(half_t());
    return by_p(bf16_t());
}

int mmfs_plan_backward(int dtype, const void *grad_loc, const void *grad_attn, const float *grad_sink,
                       const void *attn, const float *sink, const int64_t *relpos,
                       const int64_t *shapes, const float *ratios,
                       float *d_off_q, float *d_att_q, float *d_off_tab, float *d_att_tab,
                       int64_t N, int64_t Lq, int64_t H, int64_t L, int64_t P, int64_t n, int64_t M,
                       int64_t Lr, int64_t Nr, void *stream)
{
    return mmfs_plan_backward_heads(dtype, grad_loc, grad_attn, grad_sink, attn, sink, relpos, shapes, ratios, d_off_q, d_att_q,
                                    0, 0, 0, d_off_tab, d_att_tab, 0, 0, N, Lq, H, L, P, n, M, Lr, Nr, stream);
}

int mmfs_plan_backward_heads(int dtype, const void *grad_loc, const void *grad_attn, const float *grad_sink,
                             const void *attn, const float *sink, const int64_t *relpos,
                             const int64_t *shapes, const float *ratios,
                             void *d_off_q, void *d_att_q, int64_t ld_off, int64_t ld_att, int q_grads_in_storage_type,
                             float *d_off_tab, float *d_att_tab, int64_t ld_toff, int64_t ld_tatt,
                             int64_t N, int64_t Lq, int64_t H, int64_t L, int64_t P, int64_t n, int64_t M,
                             int64_t Lr, int64_t Nr, void *stream)
{
    using namespace mmfs;
    const int es = esize(dtype);
    if (!es) return MMFS_E_DTYPE;
    PlanDims d;
    const int rc = check_dims(N, Lq, H, L, P, n, M, Lr, Nr, &d);
    if (rc) return rc;
    if (N * Lq * H == 0) return MMFS_OK;
    if (!grad_loc || !grad_attn || !attn || !sink || !relpos || !shapes || !ratios || !d_off_q || !d_att_q ||
        !d_off_tab || !d_att_tab)
        return MMFS_E_NULLPTR;
    const int qes = q_grads_in_storage_type ? es : 4;
    int rr;
    if ((rr = heads_rows(ld_off, H * 2 * P, d_off_q, 2 * P, qes)) || (rr = heads_rows(ld_att, H * L * P, d_att_q, P, qes)) ||
        (rr = heads_rows(ld_toff, H * 2 * P, d_off_tab, 1, 4)) || (rr = heads_rows(ld_tatt, H * L * P, d_att_tab, 1, 4)))
        return rr;
    d.ld_off = (int)ld_off; d.ld_att = (int)ld_att; d.ld_toff = (int)ld_toff; d.ld_tatt = (int)ld_tatt;
    const int64_t units = N * H * ((Lq + kQueryRun - 1) / kQueryRun);
    hipStream_t st = (hipStream_t)stream;
    auto go = [&](auto tag_t, auto tag_p) {
        typedef decltype(tag_t) T;
        constexpr int PP = decltype(tag_p)::value;
        return for_group<T, PP>(d.n * d.L, [&](auto tag_g) {
            constexpr int G = decltype(tag_g)::value;
            const unsigned blocks = (unsigned)((units + kThreads / G - 1) / (kThreads / G));
            if (q_grads_in_storage_type && !std::is_same<T, float>::value)
                hipLaunchKernelGGL((plan_backward_kernel<T, T, PP, G>), dim3(blocks), dim3(kThreads), 0, st,
                                   (const T *)grad_loc, (const T *)grad_attn, grad_sink, (const T *)attn, sink,
                                   relpos, shapes, ratios, (T *)d_off_q, (T *)d_att_q, d_off_tab, d_att_tab, d);
            else
                hipLaunchKernelGGL((plan_backward_kernel<T, float, PP, G>), dim3(blocks), dim3(kThreads), 0, st,
                                   (const T *)grad_loc, (const T *)grad_attn, grad_sink, (const T *)attn, sink,
                                   relpos, shapes, ratios, (float *)d_off_q, (float *)d_att_q, d_off_tab, d_att_tab, d);
            return (int)hipGetLastError();
        });
    };
    auto by_p = [&](auto tag_t) {
        if (P == 4) return go(tag_t, std::integral_constant<int, 4>());
        if (P == 8) return go(tag_t, std::integral_constant<int, 8>());
        return go(tag_t, std::integral_constant<int, 16>());
    };
    if (dtype == MMFS_F32) return by_p(float());
    if (dtype == MMFS_F16) return by_p(half_t());
    return by_p(bf16_t());
}

int mmfs_sample_forward(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                        const void *off_q, const void *att_q, const void *off_tab, const void *att_tab,
                        const int64_t *relpos, const float *ref, const float *ratios, void *out, float *sink,
                        int64_t N, int64_t S, int64_t Lq, int64_t H, int64_t D, int64_t L, int64_t P, int64_t n,
                        int64_t M, int64_t Lr, int64_t Nr, void *stream)
{
    return mmfs_sample_forward_token(dtype, value, shapes, start, off_q, att_q, off_tab, att_tab, relpos, ref, ratios,
                                     nullptr, out, sink, N, S, Lq, H, D, L, P, n, M, Lr, Nr, stream);
}

int mmfs_sample_forward_token(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                              const void *off_q, const void *att_q, const void *off_tab, const void *att_tab,
                              const int64_t *relpos, const float *ref, const float *ratios, const void *token,
                              void *out, float *sink,
                              int64_t N, int64_t S, int64_t Lq, int64_t H, int64_t D, int64_t L, int64_t P, int64_t n,
                              int64_t M, int64_t Lr, int64_t Nr, void *stream)
{
    return mmfs_sample_forward_heads(dtype, value, shapes, start, off_q, att_q, 0, 0, off_tab, att_tab, relpos, ref, ratios,
                                     token, out, sink, N, S, Lq, H, D, L, P, n, M, Lr, Nr, stream);
}

int mmfs_sample_forward_heads(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                              const void *off_q, const void *att_q, int64_t ld_off, int64_t ld_att,
                              const void *off_tab, const void *att_tab,
                              const int64_t *relpos, const float *ref, const float *ratios, const void *token,
                              void *out, float *sink,
                              int64_t N, int64_t S, int64_t Lq, int64_t H, int64_t D, int64_t L, int64_t P, int64_t n,
                              int64_t M, int64_t Lr, int64_t Nr, void *stream)
{
    using namespace mmfs;
    const int es = esize(dtype);
    if (!es) return MMFS_E_DTYPE;
    PlanDims pd;
    const int rc = check_dims(N, Lq, H, L, P, n, M, Lr, Nr, &pd);
    if (rc) return rc;
    if (S < 0 || D <= 0 || S > 0x7ffffffdLL || H * D > 0x7fffffffLL) return MMFS_E_DIMS;
    if (N * Lq * H == 0) return MMFS_OK;
    if (P == 16 || S == 0) return MMFS_E_UNSUPPORTED;                       // (P = 16: the two-kernel path)
    const int vec = 16 / es;
    if (D % vec) return MMFS_E_UNSUPPORTED;
    const int lpi = (int)(D / vec);
    if (lpi < 1 || lpi > 64 || (lpi & (lpi - 1))) return MMFS_E_UNSUPPORTED;
    if (S * H * D * (int64_t)es > kMaxSlabBytes) return MMFS_E_UNSUPPORTED;        // buffer-descriptor rows only
    if (!value || !shapes || !start || !off_q || !att_q || !off_tab || !att_tab || !relpos || !ref || !ratios || !out)
        return MMFS_E_NULLPTR;
    if (((uintptr_t)value | (uintptr_t)out | (uintptr_t)token) % 16) return MMFS_E_ALIGN;
    // token rows of off_q / att_q: packed, or ld elements apart (columns of one wider matrix); vector loads of P
    // (2 P) elements need the rows aligned like the packed ones
    if (ld_off == 0) ld_off = H * 2 * P;
    if (ld_att == 0) ld_att = H * L * P;
    if (ld_off < H * 2 * P || ld_att < H * L * P || ld_off > 0x7fffffffLL || ld_att > 0x7fffffffLL) return MMFS_E_DIMS;
    if ((ld_off * es) % (2 * P * es) || (ld_att * es) % (P * es) || (uintptr_t)off_q % (2 * P * es) || (uintptr_t)att_q % (P * es))
        return MMFS_E_ALIGN;
    pd.ld_off = (int)ld_off; pd.ld_att = (int)ld_att;
    Dims d;
    d.B = (int)N; d.S = (int)S; d.H = (int)H; d.D = (int)D; d.L = (int)(n * L); d.Nq = (int)Lq; d.P = (int)P;
    d.K = d.L * d.P; d.lazy_attn = 0; d.blocks4 = 0;
    int G = 4;
    while (G < d.L) G *= 2;                                                 // the plan kernel's lane-group width
    hipStream_t st = (hipStream_t)stream;
    auto go = [&](auto tag_t, auto tag_lpi, auto tag_p) {
        typedef decltype(tag_t) T;
        constexpr int LPI = decltype(tag_lpi)::value, PP = decltype(tag_p)::value;
        constexpr int QPB = kThreads / LPI;
        Dims dd = d;
        dd.q_tiles = (d.Nq + QPB - 1) / QPB;
        const int64_t blocks = (int64_t)d.B * dd.q_tiles * d.H;
        if (blocks > 0x7fffffffLL) return (int)MMFS_E_DIMS;
        hipLaunchKernelGGL((mmfs_sample_fwd<T, LPI, PP>), dim3((unsigned)blocks), dim3(kThreads), 0, st,
                           (const T *)value, shapes, start, (const T *)off_q, (const T *)att_q,
                           (const T *)off_tab, (const T *)att_tab, relpos, ref, ratios, (T *)out, sink, dd, pd, G,
                           (const T *)token);
        return (int)hipGetLastError();
    };
    auto go_decode = [&](auto tag_t, auto tag_lpi, auto tag_p) {
        typedef decltype(tag_t) T;
        constexpr int LPI = decltype(tag_lpi)::value, PP = decltype(tag_p)::value;
        if constexpr (LPI <= 32) {
            hipLaunchKernelGGL((mmfs_sample_decode<T, LPI, PP>), dim3((unsigned)(d.B * d.Nq * d.H)), dim3(64), 0, st,
                               (const T *)value, shapes, start, (const T *)off_q, (const T *)att_q,
                               (const T *)off_tab, (const T *)att_tab, relpos, ref, ratios, (T *)out, sink, d, pd, G,
                               (const T *)token);
            return (int)hipGetLastError();
        } else {
            return (int)MMFS_E_UNSUPPORTED;
        }
    };
    const bool decode = sample_groups(Lq, d.K, d.L, lpi) > 1;
    auto by_p = [&](auto tag_t, auto tag_lpi) {
        if (decode) {
            if (P == 4) return go_decode(tag_t, tag_lpi, std::integral_constant<int, 4>());
            return go_decode(tag_t, tag_lpi, std::integral_constant<int, 8>());
        }
        if (P == 4) return go(tag_t, tag_lpi, std::integral_constant<int, 4>());
        return go(tag_t, tag_lpi, std::integral_constant<int, 8>());
    };
    auto by_lpi = [&](auto tag_t) {
        switch (lpi) {
            case 1: return by_p(tag_t, std::integral_constant<int, 1>());
            case 2: return by_p(tag_t, std::integral_constant<int, 2>());
            case 4: return by_p(tag_t, std::integral_constant<int, 4>());
            case 8: return by_p(tag_t, std::integral_constant<int, 8>());
            case 16: return by_p(tag_t, std::integral_constant<int, 16>());
            case 32: return by_p(tag_t, std::integral_constant<int, 32>());
            default: return by_p(tag_t, std::integral_constant<int, 64>());
        }
    };
    if (dtype == MMFS_F32) return by_lpi(float());
    if (dtype == MMFS_F16) return by_lpi(half_t());
    return by_lpi(bf16_t());
}

int mmfs_sample_forward_groups(int dtype, int64_t Lq, int64_t D, int64_t nL, int64_t P)
{
    const int es = mmfs::esize(dtype);
    if (!es || D <= 0 || D % (16 / es) || (P != 4 && P != 8) || nL <= 0 || Lq <= 0) return 0;
    const int64_t lpi = D / (16 / es);
    if (lpi > 64 || (lpi & (lpi - 1))) return 0;
    return mmfs::sample_groups(Lq, nL * P, nL, (int)lpi);
}

}  // extern "C"
