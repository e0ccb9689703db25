// msda_plan.h -- the plan of the sorted grad_value backward (cell tiles, 2x2 / 4x4 blocks, queue capacities) as a
// device function that any kernel can host: the prepare kernel of msda_bwd_block.hip, the stand-alone plan kernel,
// and -- one launch less per backward -- the LAST workgroup of the grad_loc / grad_attn kernel (msda_taps_mma.hip),
// which has nothing else to do with its LDS by then (PrepareJob).
#pragma once
#include "msda_bwd_block.h"

namespace mmfs {
namespace blk {

#ifndef MMFS_BLK_CAP
#define MMFS_BLK_CAP 128
#endif
constexpr int kCapRecords = MMFS_BLK_CAP;   // smallest LevelRow::cap: lists up to here are always walked in place
constexpr int kOvfLanes = 8;                // queue lanes of the vector-ALU reduce's long lists (msda_bwd_block.hip)

#ifdef __HIPCC__
// Lane groups per block of a level: aim at <= 128 records of the block's list per lane group.
__device__ __host__ inline int64_t expected_list(int64_t samples, int64_t blocks)
{
    // every sample is visited by (1 + 1/BH)(1 + 1/BW) blocks on average
    const int64_t visits = samples * (kBH + 1) * (kBW + 1) / (kBH * kBW);
    return blocks > 0 ? (visits + blocks - 1) / blocks : 0;
}
__device__ __host__ inline int split_of(int64_t samples, int64_t blocks)
{
    const int64_t per_block = expected_list(samples, blocks);
    int s = 1;
    while (s < kMaxSplit && per_block > 128LL * s) s <<= 1;
    return s;
}
// Records of a block's list that are walked in place; a list longer than twice what uniformly spread
// samples would give (hot spots: the LLM path's common reference point) has its rest queued.
__device__ __host__ inline int cap_of(int64_t samples, int64_t blocks)
{
    const int64_t c = 2 * expected_list(samples, blocks);
    return (int)(c < kCapRecords ? kCapRecords : (c > 0x3fffffff ? 0x3fffffff : c));
}

struct PlanArgs {
    const int64_t *shapes, *start;     // shapes == nullptr: no plan in this launch
    int L, S, nt_min, cap;
    int64_t samples_per_level;
    CellHeader *hdr;
    uint32_t *ovf_header;
    uint32_t cap_slots, cap_entries, cap_partials;
    TileHeader *th;
    uint32_t tile_cap_extra, tile_cap_partials;
    const void *loc_src, *attn_src;    // handed to the sort through the header (null: it reads the re-packed copies)
    int32_t *status;                   // where a table the sorted backward cannot serve is reported (device-accessible; may be null)
    uint32_t stamp;                    // header_stamp of the call's dimensions
};

// The plan: one workgroup's job.  A lane per level for everything that divides (the 64-bit divisions of one
// level cost a single lane ~2 us: serial, the plan was 7.5 us at 4 levels and the opening launch of the backward
// is nothing but the plan since the sort reads loc / attn in place); lane 0 only adds up the levels' bases.
__device__ inline void plan_cells_body(const PlanArgs &pa, unsigned char *lds)
{
    const int L = pa.L, S = pa.S, nt_min = pa.nt_min, cap = pa.cap;
    const int64_t samples_per_level = pa.samples_per_level;
    CellHeader *__restrict__ hdr = pa.hdr;
    uint32_t *__restrict__ ovf_header = pa.ovf_header;
    TileHeader *__restrict__ th = pa.th;
    const int tid = threadIdx.x, nthr = blockDim.x;
    if (th != nullptr && tid < 32) th->zero_row[tid] = make_uint4(0u, 0u, 0u, 0u);
    if (tid == nthr - 1) {
        if (th != nullptr) {
            th->n_partials = 0u; th->cap_extra = pa.tile_cap_extra; th->cap_partials = pa.tile_cap_partials; th->n_multi = 0u;
            th->null_rec = make_uint4(0u, __float_as_uint(-8.f), __float_as_uint(-8.f), 0u);
        }
        ovf_header[0] = 0u; ovf_header[1] = pa.cap_slots; ovf_header[2] = pa.cap_entries;               // OvfHeader
        ovf_header[3] = 0u; ovf_header[4] = pa.cap_partials; ovf_header[5] = ovf_header[6] = ovf_header[7] = 0u;
        for (int i = 0; i < kOvfLanes; ++i) ovf_header[8 + i] = 0u;
    }
    // scratch (kPlanLdsBytes of LDS, 16-byte aligned, handed in by the kernel that hosts the plan)
    int64_t *ltab = reinterpret_cast<int64_t *>(lds);                                    // [3 * kMaxLevels] the level table as given: H, W, first row
    LevelRow *rows = reinterpret_cast<LevelRow *>(ltab + 3 * kMaxLevels);                // [kMaxLevels] bases filled in by lane 0
    int *tile_r = reinterpret_cast<int *>(rows + kMaxLevels), *tile_c = tile_r + kMaxLevels, *tile_n = tile_c + kMaxLevels,
        *tile_base = tile_n + kMaxLevels;
    int &bad_s = tile_base[kMaxLevels], &covered_all = tile_base[kMaxLevels + 1];
    __syncthreads();                                    // (the scratch may have been somebody's data a moment ago)
    if (tid == 0) bad_s = 0;
    for (int l = tid; l < L && l < kMaxLevels; l += nthr) {
        const int64_t Hl64 = pa.shapes[2 * l], Wl64 = pa.shapes[2 * l + 1], a0 = pa.start[l];
        ltab[3 * l] = Hl64; ltab[3 * l + 1] = Wl64; ltab[3 * l + 2] = a0;
        const int Hl = (int)Hl64, Wl = (int)Wl64;
        LevelRow r;
        r.Hl = Hl; r.Wl = Wl; r.lstart = (int)a0; r.cbase = 0; r.bbase = 0; r.bbase4 = 0;
        r.nbx = (Wl + kBW - 1) / kBW; r.nby = (Hl + kBH - 1) / kBH; r.split = 1; r.cap = kCapRecords;
        r.nbx4 = (Wl + kTB - 1) / kTB; r.nby4 = (Hl + kTB - 1) / kTB;
        r.band = 0;
        LevelTiling lt = level_tiling(Hl64, Wl64, nt_min);
        const int R = lt.R, C = lt.C, n = lt.n;
        if (n == 0) {
            r.nbx = r.nby = r.nbx4 = r.nby4 = 0;                   // (empty; or refused below: no tiles)
        } else {
            r.split = split_of(samples_per_level, (int64_t)r.nbx * r.nby);
            r.cap = cap_of(samples_per_level, (int64_t)r.nbx * r.nby);
            r.band = C == Wl + 1 ? R : 0;
        }
        rows[l] = r; tile_r[l] = R; tile_c[l] = C; tile_n[l] = n;
    }
    __syncthreads();
    if (tid == 0) {
        int cbase = 0, bbase = 0, bbase4 = 0, seamed = 0;
        int64_t covered = 0, n = 0;
        for (int l = 0; l < L; ++l) {
            LevelRow &r = rows[l];
            r.cbase = cbase; r.bbase = bbase; r.bbase4 = bbase4;
            tile_base[l] = (int)min(n, (int64_t)cap);
            if (tile_n[l] == 0) continue;                                // (empty, or refused below)
            bbase4 += r.nbx4 * r.nby4;
            n += tile_n[l];
            if (tile_n[l] > 1) ++seamed;
            cbase += (r.Hl + 1) * (r.Wl + 1);
            bbase += r.nbx * r.nby * r.split;
            bbase = (bbase + kMaxSplit - 1) / kMaxSplit * kMaxSplit;     // a block's groups never straddle workgroups
            covered += (int64_t)r.Hl * r.Wl;
        }
        hdr->n_tiles = (int)min(n, (int64_t)cap); hdr->n_blocks = bbase; hdr->n_cells = cbase; hdr->L = L;
        hdr->n_blocks4 = bbase4; hdr->pad[1] = seamed; hdr->pad[2] = 0;
        hdr->stamp = pa.stamp; hdr->reserved = 0u;
        hdr->loc_src = pa.loc_src; hdr->attn_src = pa.attn_src;
        covered_all = covered == (int64_t)S;
    }
    __syncthreads();
    LevelRow *lv = level_rows(hdr);
    CTile *tile = tiles_of(hdr, L);
    // The level table lives in device memory (reference API) and the caller may not have looked at it
    // (MMFS_BWD_DEVICE_CHECKED_LEVELS: no device->host copy per call).  Owner-computes needs every
    // grad_value row to belong to at most one level: checked here.  Rows that belong to NO level (a
    // table with gaps; canonical tables have none) are zero-filled by zero_uncovered_rows.  Overlapping
    // levels -- the reference would add both levels' gradients into the shared rows -- cannot be served
    // by this path.  No trap (that would take the whole HIP context down, asynchronously): the plan is emptied --
    // no tiles, no blocks, no level owns a row, so the sort and the reduce find nothing to do and the zero-fill
    // pass clears every grad_value row -- and the fact is reported through PlanArgs::status (a word the caller
    // owns; the Python shim raises at its next call) and CellHeader::pad[2].
    for (int l = tid; l < L && l < kMaxLevels; l += nthr) {
        const LevelRow r = rows[l];
        lv[l] = r;
        int n = tile_base[l];
        const int Hc = r.Hl + 1, Wc = r.Wl + 1, R = tile_r[l], C = tile_c[l];
        if (tile_n[l] > 0)
            for (int ya = 0; ya < Hc && n < cap; ya += R)
                for (int xa = 0; xa < Wc && n < cap; xa += C, ++n) {
                    CTile t;
                    t.level = l; t.Hl = r.Hl; t.Wl = r.Wl; t.cbase = r.cbase;
                    t.ya = ya; t.yb = min(Hc, ya + R); t.xa = xa; t.xb = min(Wc, xa + C);
                    tile[n] = t;
                }
        const int64_t Hl = ltab[3 * l], Wl = ltab[3 * l + 1], a0 = ltab[3 * l + 2];
        bool bad = false;
        if (Hl < 0 || Wl < 0 || Hl >= 65536 || Wl >= 65536) bad = true;
        else if (Hl > 0 && Wl > 0) {
            const int64_t a1 = a0 + Hl * Wl;
            if (a0 < 0 || a1 > S) bad = true;
            for (int k = 0; k < l; ++k) {
                const int64_t b0 = ltab[3 * k + 2], b1 = b0 + ltab[3 * k] * ltab[3 * k + 1];
                if (ltab[3 * k] > 0 && ltab[3 * k + 1] > 0 && a0 < b1 && b0 < a1) bad = true;
            }
        }
        if (bad) atomicOr(&bad_s, 1);
    }
    __syncthreads();
    const bool bad = bad_s != 0;
    if (bad)
        for (int l = tid; l < L && l < kMaxLevels; l += nthr) {         // no level owns a row: every row is zero-filled
            lv[l].Hl = lv[l].Wl = 0; lv[l].nbx = lv[l].nby = lv[l].nbx4 = lv[l].nby4 = 0;
        }
    if (tid == 0) {
        hdr->pad[0] = (!bad && covered_all) ? 1 : 0;         // canonical in the sense that matters: every row has exactly one owner
        hdr->pad[2] = bad ? 1 : 0;
        if (bad) {
            hdr->n_tiles = 0; hdr->n_blocks = 0; hdr->n_cells = 0; hdr->n_blocks4 = 0; hdr->pad[1] = 0;
            hdr->loc_src = hdr->attn_src = nullptr;
            if (pa.status != nullptr)            // (system scope: the word may be mapped host memory)
                __hip_atomic_store(pa.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __threadfence_system();
        }
    }
}


constexpr int kPlanLdsBytes = 3 * kMaxLevels * 8 + kMaxLevels * (int)sizeof(LevelRow) + (4 * kMaxLevels + 2) * 4 + 8;

// What msda_bwd_prepare does when nothing has to be re-packed (the sort reads loc / attn where they are): clear the
// level cursors / arrival counters / queue lengths and plan.  One workgroup's job; ``lds``: kPlanLdsBytes of scratch.
struct PrepareJob {
    PlanArgs pa;
    uint32_t *cursor;
    int64_t cursor_words;              // 0: no job
};
__device__ inline void prepare_tail(const PrepareJob &job, unsigned char *lds)
{
    for (int64_t i = threadIdx.x; i < job.cursor_words; i += blockDim.x) job.cursor[i] = 0u;
    plan_cells_body(job.pa, lds);
}

#endif

}  // namespace blk
}  // namespace mmfs
