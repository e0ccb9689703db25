/*
 * include/mmfs_msda.h -- C ABI of libmmfs_msda.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for the one hot path this repository replaces in
 * OpenGVLab/MM-Interleaved: the native extension ``MultiScaleDeformableAttention``
 * behind the Multi-modal Feature Synchronizer (MMFS).  The reference binds that
 * extension with pybind11 (two functions); each entry point below names the
 * reference interface it replaces.  The Python shim that adapts torch tensors to
 * these raw pointers is mm-interleaved_amd/MultiScaleDeformableAttention.py; the
 * binding a reference maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HIP) unless stated otherwise;
 *   - all tensors are contiguous, row-major, in the layouts of the reference op:
 *       value   [B, S, H, D]          S = sum_l Hl*Wl over all L levels
 *       shapes  [L, 2]  int64 (Hl, Wl)        (device memory, as in the reference,
 *       start   [L]     int64                  ms_deform_attn_cuda.cu:68-69)
 *       loc     [B, Nq, H, L, P, 2]   (x, y) normalised to [0, 1]
 *       attn    [B, Nq, H, L, P]
 *       out     [B, Nq, H*D]
 *   - ``dtype`` selects the storage type of value/loc/attn/out/grad_out; arithmetic
 *     is fp32 for 16-bit storage (the reference's opmath, ms_deform_im2col_cuda.cuh:32);
 *   - ``stream`` is a hipStream_t (NULL = the null stream).  Launches are
 *     asynchronous; the library keeps no state, allocates and frees nothing, and is
 *     re-entrant from any host thread (autograd engine threads, checkpoint re-runs);
 *   - return value: 0 on success; < 0 argument error (MMFS_E_*); > 0 a hipError_t
 *     from the launch.  Unlike the reference, launch errors are returned, not
 *     printed (ms_deform_im2col_cuda.cuh:951-955).
 */
#ifndef MMFS_MSDA_H_
#define MMFS_MSDA_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMFS_MSDA_ABI_VERSION 13  /* 13: + mmfs_msda_backward_sorted (+ _workspace_bytes): the whole backward on the cell-sorted records -- grad_loc / grad_attn
                                   *     from the records the grad_value sort makes (csrc/msda_bwd_taps_sorted.hip), no per-sample value-row gather;
                                   *     GONE: the workgroup-local grad_value of the small levels (csrc/msda_gv_mma.hip: parity-green, slower on every
                                   *     shipped geometry for three rounds) with stage MMFS_HYB_BWD_VALUE_BLOCKS, MMFS_BWD_VALUE_SORTED_ONLY /
                                   *     MMFS_BWD_VALUE_LDS_BLOCKS, mmfs_msda_backward_value_lds_levels, mmfs_msda_debug_value_plan
                                   * 12: + mmfs_env_reload / mmfs_env_knob (the environment knobs are one table, read once); 11: + MMFS_FWD_QUERY_WAVES (the forward's fourth formulation, csrc/msda_fwd_wq.hip: the default for 16-bit
                                   *     heads of 128 channels with one chunk of samples per query -- the north-star shape); every
                                   *     matrix-core forward returns the reference's Inf / NaN element for element (a non-finite sum
                                   *     is recomputed channel by channel)
                                   * 10: + mmfs_sample_forward_groups, mmfs_linear_small_add, mmfs_plan_forward_heads / mmfs_plan_backward_heads; mmfs_sample_forward* serve at most 8 queries per (sample,
                                   *     head) -- a decode step -- with a workgroup per query (same products, another order of
                                   *     the fp32 sums: see there)
                                   * 9: + MMFS_FWD_SLICES (the forward's third formulation, csrc/msda_fwd_q8.hip);
                                   *    mmfs_msda_backward_checked serves a table its device-side check refuses IN THE
                                   *    SAME CALL (float-atomic fallback, csrc/msda_bwd_refused.hip; round 3 returned a
                                   *    zero grad_value for it): the workspace of MMFS_BWD_DEVICE_CHECKED_LEVELS grows by
                                   *    an fp32 image for 16-bit storage
                                   * 8: + stage MMFS_HYB_BWD_VALUE_BLOCKS of mmfs_msda_backward_hybrid (grad_value of the small levels sorted
                                   *      and reduced inside a workgroup, csrc/msda_gv_mma.hip), MMFS_BWD_VALUE_SORTED_ONLY /
                                   *      MMFS_BWD_VALUE_LDS_BLOCKS, mmfs_msda_backward_value_lds_levels
                                   * 7: + mmfs_msda_forward_flags (the forward's two formulations, selectable);
                                   *    + mmfs_msda_backward_checked: a level table the device-side check cannot serve is
                                   *      REPORTED through a status word, no device trap (MMFS_BWD_DEVICE_CHECKED_LEVELS
                                   *      is refused by mmfs_msda_backward, which has nowhere to report)
                                   * 6: + mmfs_sample_forward (plan -> sampler in one kernel)
                                   * 5: + MMFS_BWD_DEVICE_CHECKED_LEVELS; the dense forward / grad_value products
                                   *    (mmfs_msda_forward_hybrid*, MMFS_BWD_DENSE_VALUE) are gone: measured slower than
                                   *    the row-gather forward and the matrix-core tile reduce on every shipped geometry */

enum mmfs_dtype {
    MMFS_F32  = 0,
    MMFS_F16  = 1,
    MMFS_BF16 = 2,   /* new capability: the reference has no bf16 (ms_deform_attn_cuda.cu:65) */
    MMFS_F64  = 3
};

enum mmfs_status {
    MMFS_OK          = 0,
    MMFS_E_DTYPE     = -1,   /* unknown dtype code */
    MMFS_E_DIMS      = -2,   /* negative dim or a product that overflows the kernel's index range */
    MMFS_E_NULLPTR   = -3,   /* required pointer is NULL while the tensor is non-empty */
    MMFS_E_ALIGN     = -4,   /* tensor base not aligned to its element size */
    MMFS_E_UNSUPPORTED = -5
};

/* ABI version of the loaded library (== MMFS_MSDA_ABI_VERSION at build time). */
int mmfs_msda_abi_version(void);

/* Static description of the build, e.g. "gfx950 hipcc <ver>".  Host string. */
const char *mmfs_msda_build_info(void);

/* Text for a status returned by any entry point.  Host string, never NULL. */
const char *mmfs_msda_status_string(int status);

/* Host-only self-check of the sorted backward's tiling rules for one H x W level cut into at least
 * ``nt_min`` sort tiles (the functions the device plan uses, run on the host): 0 when the tiles cover every
 * cell exactly once and every 4x4 block of the level is planned exactly once, -1 when the level owns no
 * tile (empty, or an extent >= 65536), else a count of violations.  No reference counterpart: test hook. */
int mmfs_msda_plan_selfcheck(int64_t H, int64_t W, int nt_min);

/*
 * Forward.  Replaces ``ms_deform_attn_forward`` of the reference extension
 *   mm_interleaved/models/utils/ops/src/vision.cpp:14
 *   -> src/ms_deform_attn.h:20-39 -> src/cuda/ms_deform_attn_cuda.cu:21-81
 *   -> kernel src/cuda/ms_deform_im2col_cuda.cuh:240-302.
 * Writes every element of ``out`` (no pre-zeroing needed).  One launch covers the
 * whole batch; ``im2col_step`` of the reference has no effect on results and is
 * not part of this ABI.
 */
int mmfs_msda_forward(int dtype,
                      const void *value, const int64_t *shapes, const int64_t *start,
                      const void *loc, const void *attn, void *out,
                      int64_t B, int64_t S, int64_t H, int64_t D,
                      int64_t L, int64_t Nq, int64_t P, void *stream);

/*
 * The same forward with the formulation chosen by the caller (tests, measurements; mmfs_msda_forward is
 * flags = 0).  Two kernels implement it:
 *   row gather   (csrc/msda_fwd.hip)      every sample's four pixel rows through the vector-memory path;
 *                any storage type, any head width;
 *   LDS levels   (csrc/msda_fwd_mma.hip)  16-bit storage, D in {64, 128}, L <= 64: the levels of the pyramid
 *                that fit in the CU's LDS (smallest first, decided on the device from the table) are copied
 *                there once per workgroup and sampled by the matrix cores, the others by row gather.
 * flags = 0 (rounds 3-4) picked LDS levels for heads of 128 channels when a (b, h) slab has at least 64 queries and 4096 samples;
 * round 5: such shapes with at most 16 samples per query take "query waves" (below), longer sample lists keep LDS levels -- both only
 * in launches that give every CU two runs of (256, 128 or 64) queries, else the row gather, which fills the chip better -- (heads of
 * 64 channels measured no faster that way and stay on the row gather / the slices unless MMFS_FWD_LDS_LEVELS asks).
 * MMFS_FWD_LDS_LEVELS on a shape that does not allow it returns MMFS_E_UNSUPPORTED.
 */
#define MMFS_FWD_ROW_GATHER 1u
#define MMFS_FWD_LDS_LEVELS 2u
/*   slices       csrc/msda_fwd_q8.hip (round 4): 16-bit storage, head widths that are multiples of 32 channels, L <= 64.
 *                A workgroup owns a 32-channel SLICE of one (batch, head); every level whose slice fits in LDS (32x32 +
 *                16x16 + 8x8: 101 KB) is sampled by the matrix cores in tiles of 8 queries, the rest by row gather.
 * MMFS_FWD_SLICES on a shape that does not allow it returns MMFS_E_UNSUPPORTED. */
#define MMFS_FWD_SLICES 4u
/*   query waves  csrc/msda_fwd_wq.hip (round 5): 16-bit storage, D = 128, L <= 64.  A wave works on one query at a time;
 *                a sample's four pixel rows -- one wave-wide load, from memory or from the LDS image of the levels that
 *                fit -- are the B operand of ONE matrix-core product whose A operand carries the four bilinear weights
 *                on a diagonal (no unpack, no multiply-add in the vector ALU).  The default for that shape.
 * MMFS_FWD_QUERY_WAVES on a shape that does not allow it returns MMFS_E_UNSUPPORTED. */
#define MMFS_FWD_QUERY_WAVES 8u
int mmfs_msda_forward_flags(int dtype,
                            const void *value, const int64_t *shapes, const int64_t *start,
                            const void *loc, const void *attn, void *out,
                            int64_t B, int64_t S, int64_t H, int64_t D,
                            int64_t L, int64_t Nq, int64_t P, unsigned flags, void *stream);

/* Flags of mmfs_msda_backward(). */
#define MMFS_BWD_CANONICAL_LEVELS 1u
/*   The caller guarantees the level table is the canonical packing the reference's
 *   callers build (modeling_llama_mmfs.py:303-305, sd_mmfs.py:35-37):
 *       start[l] == sum_{k<l} H_k*W_k   and   sum_l H_l*W_l == S.
 *   (The table lives in device memory, so the library cannot check it without a
 *   device->host sync.)  With the guarantee every grad_value pixel has one owner level
 *   and the pixel-stationary kernel applies; without it the library falls back to
 *   float-atomic accumulation, which is also correct for gapped or overlapping levels. */
#define MMFS_BWD_FORCE_ATOMIC 2u      /* testing/measurement: always take the atomic path */
#define MMFS_BWD_DENSE_TAPS   4u      /* mmfs_msda_backward_hybrid: small levels' grad_loc / grad_attn by MFMA */
/* The caller does not read grad_attn (nor grad_loc, which is 0 anyway) of samples whose attention
 * weight is exactly 0: they may be written as 0 without reading the value rows.  True for MMFS: the
 * weights come out of its masked softmax (mmfs.py:203-231), whose backward multiplies every
 * incoming gradient by the weight itself, and images a token cannot see have weight exactly 0.
 * Honoured by mmfs_msda_backward and mmfs_msda_backward_hybrid (row-gather kernel); a hint. */
#define MMFS_BWD_LAZY_ZERO_ATTN 16u
/* The caller has NOT looked at the level table (it lives in device memory, and the reference's callers
 * rebuild it on every call: modeling_llama_mmfs.py:298-308, sd_mmfs.py:31-41) and does not want to pay a
 * device->host copy per backward to find out whether MMFS_BWD_CANONICAL_LEVELS holds.  The sorted
 * backward is taken and the table is checked ON THE DEVICE: any table whose levels do not overlap is
 * served (rows that belong to no level are zero-filled, like the reference's zero-initialised output).
 * Overlapping, out-of-range or >= 65536-wide levels cannot be served by the sorted backward: the SAME call then
 * computes grad_value as the reference does, with float atomics into a zero-filled fp32 image and a cast
 * (cuh:128-155, .cu:122-165; csrc/msda_bwd_refused.hip -- three launches that read the plan's verdict on the
 * device and return at once for any table the sorted backward served), and ORs 1 into the caller's status word
 * (mmfs_msda_backward_checked) so that the caller can learn it took the slow path.  The workspace of such a call
 * is larger by that image for 16-bit storage (mmfs_msda_backward_workspace_bytes with this flag says how much).
 * Only mmfs_msda_backward_checked takes this flag. */
#define MMFS_BWD_DEVICE_CHECKED_LEVELS 32u
/* grad_loc / grad_attn have two formulations (tests, measurements; default: the library chooses):
 *   row gather   csrc/msda_bwd.hip, plus -- mmfs_msda_backward_hybrid only -- levels of <= 256 pixels as a DENSE
 *                matrix-core product over all their pixels (csrc/msda_dense.hip);
 *   LDS levels   csrc/msda_taps_mma.hip: one kernel for all levels, 16-bit storage, D = 128, L <= 64; the levels that
 *                fit in the CU's LDS (decided on the device) are contracted on the matrix cores from GATHERED rows,
 *                the others by row gather.  Default from 64 queries and 4096 samples per (b, h) slab on.
 * MMFS_BWD_TAPS_LDS_LEVELS on a shape that does not allow it returns MMFS_E_UNSUPPORTED. */
#define MMFS_BWD_TAPS_ROW_GATHER 64u
#define MMFS_BWD_TAPS_LDS_LEVELS 128u
/* 1 when, for these arguments, grad_loc / grad_attn of ALL levels come from the one LDS-levels kernel (the staged
 * hybrid backward's MMFS_HYB_BWD_TAPS_COARSE stage then has nothing to launch), else 0.  Host-only. */
int mmfs_msda_backward_taps_fused(int dtype, int64_t B, int64_t S, int64_t H, int64_t D,
                                  int64_t L, int64_t Nq, int64_t P, unsigned flags);

/*
 * Scratch the backward needs for these arguments (0 when none).  Host-only computation.
 * Pixel-stationary path: re-packed copies of loc/attn (3*pts elements, pts = B*Nq*H*L*P), one
 * cursor per (b, h, level), a {first, count} run table per (b, h, pixel) and the pixel-sorted
 * {query, weight} records (4*pts * 8 bytes);
 * atomic path with 16-bit storage: an fp32 image of grad_value.
 */
int64_t mmfs_msda_backward_workspace_bytes(int dtype, int64_t B, int64_t S, int64_t H, int64_t D,
                                           int64_t L, int64_t Nq, int64_t P, unsigned flags);

/*
 * Backward.  Replaces ``ms_deform_attn_backward`` of the reference extension
 *   src/vision.cpp:15 -> src/ms_deform_attn.h:42-61
 *   -> src/cuda/ms_deform_attn_cuda.cu:84-166
 *   -> kernels src/cuda/ms_deform_im2col_cuda.cuh:304-923.
 *
 *   grad_out    [B, Nq, H*D]          storage dtype
 *   grad_value  [B, S, H, D]          storage dtype, fully overwritten (no pre-zeroing)
 *   grad_loc    [B, Nq, H, L, P, 2]   storage dtype, fully overwritten
 *   grad_attn   [B, Nq, H, L, P]      storage dtype, fully overwritten
 *   workspace   device scratch of at least mmfs_msda_backward_workspace_bytes(...)
 *               bytes (may be NULL when that is 0); contents irrelevant on entry
 *
 * grad_value is accumulated in fp32 (fp64 for MMFS_F64) and rounded once at the end,
 * the reference's semantics (ms_deform_attn_cuda.cu:122-129, 156-165).  Two
 * implementations, chosen by the library (see flags):
 *   pixel-stationary (no atomics; msda_bwd_value.hip) and float-atomic scatter
 *   (msda_bwd.hip; zero-fills and casts through ``workspace`` itself).
 */
int mmfs_msda_backward(int dtype,
                       const void *value, const int64_t *shapes, const int64_t *start,
                       const void *loc, const void *attn, const void *grad_out,
                       void *grad_value, void *grad_loc, void *grad_attn,
                       void *workspace, int64_t workspace_bytes,
                       int64_t B, int64_t S, int64_t H, int64_t D,
                       int64_t L, int64_t Nq, int64_t P, unsigned flags, void *stream);

/*
 * The same backward for callers that pass MMFS_BWD_DEVICE_CHECKED_LEVELS: ``table_status`` points at an int32
 * the device can write (device memory or mapped host memory), owned and zero-initialised by the caller.  When
 * the device-side check finds a table the sorted backward cannot serve, the sorted launches find nothing to do,
 * the float-atomic fallback at the end of the sequence computes grad_value as the reference does, and bit 0 of
 * *table_status is set (information only: every output of the call is correct; MMFS_BWD_FORCE_ATOMIC, or no flag,
 * takes the float-atomic path directly and faster).
 * Without the flag ``table_status`` is ignored (may be NULL) and the call is mmfs_msda_backward.
 */
int mmfs_msda_backward_checked(int dtype,
                               const void *value, const int64_t *shapes, const int64_t *start,
                               const void *loc, const void *attn, const void *grad_out,
                               void *grad_value, void *grad_loc, void *grad_attn,
                               void *workspace, int64_t workspace_bytes,
                               int64_t B, int64_t S, int64_t H, int64_t D,
                               int64_t L, int64_t Nq, int64_t P, unsigned flags,
                               int32_t *table_status, void *stream);

/*
 * The two stages of the pixel-stationary backward as separate launches (what
 * mmfs_msda_backward runs back to back when MMFS_BWD_CANONICAL_LEVELS applies); exported so
 * a caller can time or overlap them.  Same tensors and conventions as above.
 *   _taps  : grad_loc and grad_attn only (reads value)            -- kernel msda_bwd_vec<.., false>
 *   _value : grad_value only, canonical level table REQUIRED      -- kernel msda_bwd_value_tiled
 * Both return MMFS_E_UNSUPPORTED when the head width has no vector path
 * (D*sizeof(dtype) must be 16 bytes * 2^k, k <= 6; MMFS_F64 never): use mmfs_msda_backward.
 */
int mmfs_msda_backward_taps(int dtype,
                            const void *value, const int64_t *shapes, const int64_t *start,
                            const void *loc, const void *attn, const void *grad_out,
                            void *grad_loc, void *grad_attn,
                            int64_t B, int64_t S, int64_t H, int64_t D,
                            int64_t L, int64_t Nq, int64_t P, void *stream);
int mmfs_msda_backward_value(int dtype,
                             const int64_t *shapes, const int64_t *start,
                             const void *loc, const void *attn, const void *grad_out,
                             void *grad_value,
                             void *workspace, int64_t workspace_bytes,   /* as for mmfs_msda_backward */
                             int64_t B, int64_t S, int64_t H, int64_t D,
                             int64_t L, int64_t Nq, int64_t P, void *stream);

/* mmfs_msda_backward_value == _prepare (re-pack loc/attn into the workspace, clear the level
 * cursors), then _sort, then _reduce (_run == _sort + _reduce); exported separately so each kernel
 * can be timed / profiled on its own.  Two generations answer to _sort / _reduce:
 *   block-stationary (csrc/msda_bwd_block.hip; default for L <= 128 and head rows of 64..512 bytes):
 *     _sort keys every SAMPLE by the cell of its top-left corner (kernel msda_bwd_cell_sort, one
 *     16-byte record per sample); _reduce lets a lane group own a 2x2 pixel block and walk the 9
 *     cell runs that touch it (kernel msda_bwd_block_reduce: 2.25 instead of 4 grad_out row reads
 *     per sample), queues the lists of hot blocks whole and finishes them with
 *     msda_bwd_block_overflow / msda_bwd_block_ovf_store;
 *   pixel-stationary (csrc/msda_bwd_value.hip; the fallback, or MMFS_VALUE_ALGO=pixel): _sort writes
 *     one record per tap corner sorted by pixel (kernel msda_bwd_value_sort), _reduce lets a lane
 *     group own one pixel (kernel msda_bwd_value_reduce). */
int mmfs_msda_backward_value_prepare(int dtype, const void *loc, const void *attn,
                                     void *workspace, int64_t workspace_bytes,
                                     int64_t B, int64_t S, int64_t H, int64_t D,
                                     int64_t L, int64_t Nq, int64_t P, void *stream);
int mmfs_msda_backward_value_sort(int dtype, const int64_t *shapes, const int64_t *start,
                                  void *workspace, int64_t workspace_bytes,
                                  int64_t B, int64_t S, int64_t H, int64_t D,
                                  int64_t L, int64_t Nq, int64_t P, void *stream);
int mmfs_msda_backward_value_reduce(int dtype, const void *grad_out, void *grad_value,
                                    void *workspace, int64_t workspace_bytes,
                                    int64_t B, int64_t S, int64_t H, int64_t D,
                                    int64_t L, int64_t Nq, int64_t P, void *stream);
int mmfs_msda_backward_value_run(int dtype, const int64_t *shapes, const int64_t *start,
                                 const void *grad_out, void *grad_value,
                                 void *workspace, int64_t workspace_bytes,
                                 int64_t B, int64_t S, int64_t H, int64_t D,
                                 int64_t L, int64_t Nq, int64_t P, void *stream);

/* ------------------------------------------------------------------------------------------
 * Hybrid backward (no counterpart in the reference; same results within the storage type's
 * rounding).  On MI355X the row gathers of this op are bound by the vector-memory path, not by
 * HBM, and every level receives the same number of taps whatever its size.  When the caller can
 * hand over a HOST copy of the level table (``host_shapes`` [L, 2], ``host_start`` [L]: int64 in
 * host memory, equal to the device tables), grad_loc / grad_attn of levels of at most
 * min(256, 64*P) pixels are evaluated as a dense product on the matrix cores
 * (csrc/msda_dense.hip, kernel msda_taps_coarse):
 *     dot = grad_out . V_l^T    [queries x pixels], then 4 look-ups per sample
 * and the other levels run through the row-gather kernel of the plain entry point, restricted to
 * those levels.  grad_value always takes the sorted path of mmfs_msda_backward.
 * Applies to MMFS_F16 / MMFS_BF16, D in {32, 64, 128}, L <= 64, Nq >= 32, at least one such level,
 * MMFS_BWD_CANONICAL_LEVELS and MMFS_BWD_DENSE_TAPS in ``flags``; otherwise the *_workspace_bytes
 * query returns 0 and the call MMFS_E_UNSUPPORTED: use mmfs_msda_backward.
 * One behavioural difference, inherent to a dense product: a NON-FINITE value element inside a
 * dense level propagates (0 * Inf = NaN) to grad_loc / grad_attn of every query of its (b, h),
 * not only to the queries that sample it.  Environment MMFS_HYBRID=0 disables the routing.
 *
 * ``stages`` selects which launches a call issues (OR of the bits; all of them = the whole
 * pass, in this order), so each kernel can be timed on its own.  The stages of one pass share the
 * workspace and run in this order: VALUE_PREPARE also plans the sort (level rows, tiles) and may tell
 * the sort to read ``loc`` / ``attn`` where they are instead of re-packing them, so VALUE_SORT needs
 * the VALUE_PREPARE of THIS entry point on the same workspace, and ``loc`` / ``attn`` unchanged until
 * the sort has run.
 */
#define MMFS_HYB_BWD_TAPS_FINE      1u
#define MMFS_HYB_BWD_TAPS_COARSE    2u
#define MMFS_HYB_BWD_VALUE_PREPARE  4u
#define MMFS_HYB_BWD_VALUE_SORT     8u
#define MMFS_HYB_BWD_VALUE_REDUCE  16u
#define MMFS_HYB_BWD_ALL           31u
int64_t mmfs_msda_backward_hybrid_workspace_bytes(int dtype,
                                                  const int64_t *host_shapes, const int64_t *host_start,
                                                  int64_t B, int64_t S, int64_t H, int64_t D,
                                                  int64_t L, int64_t Nq, int64_t P, unsigned flags);
int mmfs_msda_backward_hybrid(int dtype,
                              const void *value, const int64_t *shapes, const int64_t *start,
                              const int64_t *host_shapes, const int64_t *host_start,
                              const void *loc, const void *attn, const void *grad_out,
                              void *grad_value, void *grad_loc, void *grad_attn,
                              void *workspace, int64_t workspace_bytes,
                              int64_t B, int64_t S, int64_t H, int64_t D,
                              int64_t L, int64_t Nq, int64_t P,
                              unsigned flags, unsigned stages, void *stream);

/* ------------------------------------------------------------------------------------------
 * The whole backward on the CELL-SORTED records (round 6; replaces ms_deform_attn_backward, vision.cpp:15,
 * ms_deform_attn_cuda.cu:84-166, ms_deform_im2col_cuda.cuh:90-162 + 304-923, like mmfs_msda_backward).
 * The grad_value half sorts the samples by the cell of their top-left corner anyway; a sample's four corners are the four
 * pixels around that cell.  Here grad_loc / grad_attn are computed from the same records (csrc/msda_bwd_taps_sorted.hip):
 * a wave per 4x4 block of cells holds the block's 5x5 value rows as one matrix-core operand and multiplies them with the
 * records' grad_out rows -- no value row is gathered per sample (the gather kernels: 4 rows per sample through the
 * vector-memory path).  Same results within the storage type's rounding (same products, fp32 sums in another order).
 * Applies to MMFS_F16 / MMFS_BF16, D in {32, 64, 128}, P a power of two with Nq * P <= 65536, a level table the HOST has
 * verified (MMFS_BWD_CANONICAL_LEVELS in ``flags``) and shapes whose sort keeps its samples in registers (Nq <= 4096 per
 * vector group); otherwise *_workspace_bytes returns 0 and the call MMFS_E_UNSUPPORTED: use mmfs_msda_backward[_hybrid].
 * MMFS_BWD_LAZY_ZERO_ATTN as there.  ``stages``: OR of the bits, all of them = the whole pass, in this order (each stage
 * needs the ones before it on the same workspace, and ``loc`` / ``attn`` unchanged until the sort has run).
 */
#define MMFS_SRT_BWD_PREPARE  1u   /* clear the cursors, plan (level rows, tiles) */
#define MMFS_SRT_BWD_SORT     2u   /* cell sort; also writes the zero grad_loc / grad_attn of samples that get no record */
#define MMFS_SRT_BWD_TAPS     4u   /* grad_loc / grad_attn from the records */
#define MMFS_SRT_BWD_REDUCE   8u   /* grad_value from the records (the matrix-core tile reduce) */
#define MMFS_SRT_BWD_ALL     15u
int64_t mmfs_msda_backward_sorted_workspace_bytes(int dtype, int64_t B, int64_t S, int64_t H, int64_t D,
                                                  int64_t L, int64_t Nq, int64_t P, unsigned flags);
/* blocks4: the number of 4x4 pixel blocks of all levels when the caller knows the level table on the host (exact grids),
 * else 0 (a bound is launched). */
int mmfs_msda_backward_sorted(int dtype,
                              const void *value, const int64_t *shapes, const int64_t *start,
                              const void *loc, const void *attn, const void *grad_out,
                              void *grad_value, void *grad_loc, void *grad_attn,
                              void *workspace, int64_t workspace_bytes,
                              int64_t B, int64_t S, int64_t H, int64_t D,
                              int64_t L, int64_t Nq, int64_t P, int64_t blocks4,
                              unsigned flags, unsigned stages, void *stream);

/*
 * dst[i] = (dtype) src[i], round-to-nearest-even.  Replaces the trailing
 * ``grad_value.to(torch::kHalf)`` of the reference backward (ms_deform_attn_cuda.cu:156-165);
 * used internally by the atomic path and exported for callers that keep fp32 buffers.
 * ``dtype`` must be MMFS_F16 or MMFS_BF16 (MMFS_F32 copies).
 */
int mmfs_msda_cast_from_f32(int dtype, const float *src, void *dst, int64_t n, void *stream);

/* ------------------------------------------------------------------------------------------
 * MMFS sampling plan: everything the reference's MMFS.forward does
 * between its Linear layers and the op, mm_interleaved/models/utils/ops/modules/mmfs.py:181-265
 * (relative-position term, per-level offset scaling, visibility penalty, constant sink logit,
 * softmax over all n*L*(P+1) logits, offsets -> normalised locations), in ONE kernel each way.
 * The linear heads are evaluated by the caller as head(W q) (off_q, att_q) plus table rows
 * head.weight @ query_relpos[m] (off_tab, att_tab); see mmfs_amd/modules/mmfs.py.
 *
 *   off_q   [N, Lq, H, P, 2]       att_q   [N, Lq, H, L, P]            (storage dtype)
 *   off_tab [M, H, P, 2]           att_tab [M, H, L, P]                (storage dtype)
 *           (point columns only: the head's (P+1)-th "sink" column is a constant, mmfs.py:225)
 *   relpos  [N, Lr, n] int64   image's rank among the visible ones, 0 = not visible; Lr = 1 or Lq
 *   ref     [Nr, Lq, 2] fp32   reference point (x, y) per query; Nr = 1 or N
 *   shapes  [n*L, 2] int64 (H_l, W_l);   ratios [L] fp32 (spatial_shape / base_spatial_shape)
 *   loc     [N, Lq, H, n*L, P, 2]  attn [N, Lq, H, n*L, P]  (storage dtype)   sink [N, Lq, H] fp32
 * Limits: P in {4, 8, 16}, n*L <= 64 (else MMFS_E_UNSUPPORTED: use the framework ops).
 * Backward: d_off_q / d_att_q are fully written (fp32, shaped like off_q / att_q); d_off_tab /
 * d_att_tab (fp32, shaped like the tables) are ACCUMULATED into and must be zero-filled by the
 * caller; grad_sink may be NULL (the sink weights feed nothing trainable in the reference).
 */
int mmfs_plan_forward(int dtype, const void *off_q, const void *att_q, const void *off_tab,
                      const void *att_tab, const int64_t *relpos, const float *ref,
                      const int64_t *shapes, const float *ratios, void *loc, void *attn, float *sink,
                      int64_t N, int64_t Lq, int64_t H, int64_t L, int64_t P, int64_t n, int64_t M,
                      int64_t Lr, int64_t Nr, void *stream);
int mmfs_plan_backward(int dtype, const void *grad_loc, const void *grad_attn, const float *grad_sink,
                       const void *attn, const float *sink, const int64_t *relpos,
                       const int64_t *shapes, const float *ratios,
                       float *d_off_q, float *d_att_q, float *d_off_tab, float *d_att_tab,
                       int64_t N, int64_t Lq, int64_t H, int64_t L, int64_t P, int64_t n, int64_t M,
                       int64_t Lr, int64_t Nr, void *stream);

/* The two calls for a caller that evaluates both query heads as ONE GEMM (their weights stacked: both read the same
 * activations, mmfs.py:174-176) and both tables as one: ``off_q`` / ``att_q`` are two column ranges of one
 * [N*Lq, H*2P + H*L*P] matrix -- token rows ``ld_off`` / ``ld_att`` elements apart (0 = packed) --, ``off_tab`` / ``att_tab``
 * likewise of one [M, ...] matrix (``ld_toff`` / ``ld_tatt``); the backward writes the gradients in the same layout, the
 * query-side ones in the storage type when ``q_grads_in_storage_type`` (one rounding of the fp32 sums, what a caller's cast
 * of the fp32 tensors does) -- so the gradient of the stacked GEMM's result is ONE tensor, made by this kernel. */
int mmfs_plan_forward_heads(int dtype, const void *off_q, const void *att_q, int64_t ld_off, int64_t ld_att,
                            const void *off_tab, const void *att_tab, int64_t ld_toff, int64_t ld_tatt,
                            const int64_t *relpos, const float *ref,
                            const int64_t *shapes, const float *ratios, void *loc, void *attn, float *sink,
                            int64_t N, int64_t Lq, int64_t H, int64_t L, int64_t P, int64_t n, int64_t M,
                            int64_t Lr, int64_t Nr, void *stream);
int mmfs_plan_backward_heads(int dtype, const void *grad_loc, const void *grad_attn, const float *grad_sink,
                             const void *attn, const float *sink, const int64_t *relpos,
                             const int64_t *shapes, const float *ratios,
                             void *d_off_q, void *d_att_q, int64_t ld_off, int64_t ld_att, int q_grads_in_storage_type,
                             float *d_off_tab, float *d_att_tab, int64_t ld_toff, int64_t ld_tatt,
                             int64_t N, int64_t Lq, int64_t H, int64_t L, int64_t P, int64_t n, int64_t M,
                             int64_t Lr, int64_t Nr, void *stream);

/*
 * The plan feeding the sampler (second half): mmfs_plan_forward followed by
 * mmfs_msda_forward as ONE kernel -- the locations and weights [N, Lq, H, n*L, P(, 2)] never exist in
 * memory.  Same inputs as mmfs_plan_forward plus the op's value / level tables (``shapes`` / ``start``
 * have n*L rows), same arithmetic (the plan's numbers are rounded to the storage type before use), so
 * ``out`` [N, Lq, H*D] is bit-identical to the two calls (more than 8 queries per (sample, head); fewer: see
 * mmfs_sample_forward_groups).  ``sink`` [N, Lq, H] fp32 may be NULL.
 * Forward only -- the backward needs loc / attn as tensors (a training step keeps the two calls).
 * MMFS_E_UNSUPPORTED (use the two calls) for P = 16, head rows that are not 16 bytes x 2^k (k <= 6), value
 * slabs of 2 GiB or more.
 */
int mmfs_sample_forward(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                        const void *off_q, const void *att_q, const void *off_tab, const void *att_tab,
                        const int64_t *relpos, const float *ref, const float *ratios, void *out, float *sink,
                        int64_t N, int64_t S, int64_t Lq, int64_t H, int64_t D, int64_t L, int64_t P, int64_t n,
                        int64_t M, int64_t Lr, int64_t Nr, void *stream);
/* The same with MMFS's ignore-token term folded into the store (mmfs.py:236-241, 274: ``out + ignore_token * sink``;
 * three framework kernels and a full-size temporary per block otherwise): ``token`` [H, D] of the storage type, 16-byte
 * aligned, or NULL (= mmfs_sample_forward).  Roundings as the framework statement's: the sampled output, the sink
 * weight and their product are each rounded to the storage type before the sum -- bit-identical to the three kernels. */
int mmfs_sample_forward_token(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                              const void *off_q, const void *att_q, const void *off_tab, const void *att_tab,
                              const int64_t *relpos, const float *ref, const float *ratios, const void *token,
                              void *out, float *sink,
                              int64_t N, int64_t S, int64_t Lq, int64_t H, int64_t D, int64_t L, int64_t P, int64_t n,
                              int64_t M, int64_t Lr, int64_t Nr, void *stream);
/* The same with the two query heads as COLUMNS of wider matrices: token row t of ``off_q`` starts ``ld_off`` elements
 * after row t - 1 (``ld_att`` for ``att_q``; 0 = packed, H*2P / H*L*P).  What a caller that evaluates
 * sampling_offsets and attention_weights as ONE GEMM (their weights stacked: both read the same activations,
 * mmfs.py:174-176) hands over: two column ranges of its [N*Lq, H*2P + H*L*P] result, no copies. */
int mmfs_sample_forward_heads(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                              const void *off_q, const void *att_q, int64_t ld_off, int64_t ld_att,
                              const void *off_tab, const void *att_tab,
                              const int64_t *relpos, const float *ref, const float *ratios, const void *token,
                              void *out, float *sink,
                              int64_t N, int64_t S, int64_t Lq, int64_t H, int64_t D, int64_t L, int64_t P, int64_t n,
                              int64_t M, int64_t Lr, int64_t Nr, void *stream);

/* How many lane groups share ONE query's samples in mmfs_sample_forward* for this shape (``nL`` = n * L level rows):
 * 1 = the samples are summed in their order, ``out`` bit-identical to mmfs_plan_forward + mmfs_msda_forward;
 * > 1 (at most 8 queries per (sample, head): a decode step; csrc/mmfs_plan.hip ``mmfs_sample_decode``) = every row of
 * a query is requested at once and the groups' fp32 partial sums are added in a fixed tree: the same products, equal to
 * the two calls within one rounding of the storage type, deterministic.  0 for shapes mmfs_sample_forward refuses. */
int mmfs_sample_forward_groups(int dtype, int64_t Lq, int64_t D, int64_t nL, int64_t P);

/* ---------------------------------------------------------------------------------------------
 * Multi-image feature bank: MMFS's ``input_flatten`` built in one pass.
 * Replaces the Python loops of mm_interleaved/models/mm_interleaved.py:223-250 (zero-filled
 * [B, n, C, h, w] buffers per level, per-sequence slice copies, "b n c h w -> b n (h w) c", concatenation
 * over levels) and the rearrange + concatenation of decoders/sd_mmfs.py:241-245.
 *   level_ptrs [n_levels]  HOST array of DEVICE pointers: level l is [n_img, C, hw_l] (channel-major,
 *                          contiguous), storage dtype;   level_hw [n_levels]  HOST array: hw_l = h_l * w_l
 *   src_index  [n_slots]   device int64: the image a bank slot shows; < 0 or >= n_img -> a zero slot
 *   bank       [n_slots, sum_l hw_l, C]   token-major, levels in list order, fully written
 * mmfs_bank_scatter is the adjoint (needed when the image encoder is trained):
 *   grad_level_ptrs[l] [n_img, C, hw_l] = sum over the slots s with src_index[s] == image of
 *   grad_bank[s] (fp32 accumulation, slots in ascending order, rounded once); fully written, images
 *   no slot shows get zeros.  No atomics: bit-reproducible.
 * Limits: n_levels <= 8, n_slots <= 65535 (gather) / n_img <= 65535 (scatter); dtypes f32, f16, bf16.
 */
int mmfs_bank_gather(int dtype, int n_levels, const void *const *level_ptrs, const int64_t *level_hw,
                     const int64_t *src_index, void *bank, int64_t n_img, int64_t C, int64_t n_slots,
                     void *stream);
int mmfs_bank_scatter(int dtype, int n_levels, void *const *grad_level_ptrs, const int64_t *level_hw,
                      const int64_t *src_index, const void *grad_bank, int64_t n_img, int64_t C,
                      int64_t n_slots, void *stream);

/*
 * RMS normalisation of the LLM-side block (``LlamaRMSNorm``, mm_interleaved/models/decoders/modeling_llama_mmfs.py:53-70,
 * applied at :352-353 to the token stream and to the feature bank), one pass each way instead of the reference's
 * chain of seven framework kernels.  x, y, grad_y, grad_x: [rows, C] contiguous, ``weight`` [C], all of storage type
 * ``dtype`` (MMFS_F32 / MMFS_F16 / MMFS_BF16), 16-byte aligned; ``rstd`` [rows] fp32 (forward: written when not
 * NULL; backward: read).  y = weight * round(x * rsqrt(mean(x^2) + eps)), the rounding to the storage type between
 * the two products exactly where the reference has its ``.to(weight.dtype)``.  Backward: ``grad_weight_f32`` [C] fp32
 * is ACCUMULATED into (atomics): zero it first; cast it to the storage type afterwards.
 * mmfs_rmsnorm_supported: C a multiple of the 16-byte vector and at most 1024 vectors (8192 channels of 16 bits).
 */
int mmfs_rmsnorm_supported(int dtype, int64_t C);
int mmfs_rmsnorm_forward(int dtype, const void *x, const void *weight, void *y, float *rstd,
                         int64_t rows, int64_t C, float eps, void *stream);
int mmfs_rmsnorm_backward(int dtype, const void *grad_y, const void *x, const void *weight, const float *rstd,
                          void *grad_x, float *grad_weight_f32, int64_t rows, int64_t C, void *stream);
/* The same backward without atomics: workgroup g leaves ITS sum of the gain gradient in row g of
 * ``grad_weight_partials`` [mmfs_rmsnorm_backward_partials_rows(rows), C] fp32 (every element written: no zeroing), and
 * the caller adds the rows up (one framework reduction).  2 M float atomics on C addresses from hundreds of workgroups on
 * 8 XCDs were 150 of mmfs_rmsnorm_backward's 180 us at 8192 x 4096; what the Python operator package calls. */
int mmfs_rmsnorm_backward_partials_rows(int64_t rows);
int mmfs_rmsnorm_backward_partials(int dtype, const void *grad_y, const void *x, const void *weight, const float *rstd,
                                   void *grad_x, float *grad_weight_partials, int64_t rows, int64_t C, void *stream);

/* ---- the two layout changes around the image decoder's synchronizer block (csrc/mmfs_query.hip) ----------------------
 * MMFSBlock (mm_interleaved/models/decoders/sd_mmfs.py:121-146) gets a UNet residual [B, C, H, W] and works on tokens:
 *   mmfs_query_prep:  q[b, p, :] = round(LayerNorm_C(x[b, :, p]) * gamma + beta) + pos[p, :]      (sd_mmfs.py:124-131)
 *                     x [B, C, HW] -> q [B, HW, C]; statistics in fp32; ``pos`` [HW, C] may be NULL (no second term);
 *                     ``mean`` / ``rstd`` [B * HW] fp32, both or neither: what a LayerNorm backward needs.
 *   mmfs_tokens_add:  y[b, c, p] = round(tok[b, p, c] + res[b, c, p])     (the rearrange back + the caller's residual add,
 *                     sd_mmfs.py:146, 262-270); tok [B, HW, C], res and y [B, C, HW].
 * One read of the input and one write of the output each (a transposing tile in LDS) instead of a copy, a normalisation
 * and two adds.  16-bit storage types, C % 8 == 0, C <= 2048, HW % 8 == 0 (mmfs_query_prep_supported); else
 * MMFS_E_UNSUPPORTED and the caller keeps the framework's kernels. */
int mmfs_query_prep_supported(int dtype, int64_t C, int64_t HW);
int mmfs_query_prep(int dtype, const void *x, const void *gamma, const void *beta, const void *pos, void *q,
                    float *mean, float *rstd, int64_t B, int64_t C, int64_t HW, float eps, void *stream);
int mmfs_tokens_add(int dtype, const void *tok, const void *res, void *y, int64_t B, int64_t C, int64_t HW, void *stream);

/* ---- a Linear layer for a handful of tokens (csrc/mmfs_linear.hip) -----------------------------------------------------
 * y[m, :] = x[m, :] W^T + bias for M <= 8 token rows -- what an MMFS layer's Linear layers (mmfs.py:174-176, 274) are in a
 * decode step: each weight row is read once and meets M activations.  x [M, K] with rows ``ldx`` elements apart, W [N, K]
 * packed (nn.Linear's layout), bias [N] or NULL, y [M, N] with rows ``ldy`` apart; fp32 accumulation, bias added in fp32,
 * one rounding to the storage type.  16-bit storage types, K % 8 == 0, M * K * 2 bytes <= 64 KB (rounded up to 4 or 8
 * rows); else MMFS_E_UNSUPPORTED (mmfs_linear_small_supported) and the caller keeps its BLAS call. */
int mmfs_linear_small_supported(int dtype, int64_t M, int64_t N, int64_t K);
int mmfs_linear_small(int dtype, const void *x, const void *weight, const void *bias, void *y,
                      int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldy, void *stream);
/* The same followed by ``+ residual`` [M, N] (rows ``ldr`` apart; NULL = mmfs_linear_small): the decoder layer's
 * ``hidden_states = residual + hidden_states`` (modeling_llama_mmfs.py:700-717) in the projection's store -- the result is
 * rounded to the storage type, THEN added and rounded again: the bits of the framework's two kernels. */
int mmfs_linear_small_add(int dtype, const void *x, const void *weight, const void *bias, const void *residual, void *y,
                          int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldy, int64_t ldr, void *stream);

/* ---- the library's environment knobs (csrc/msda_env.h) --------------------------------------------------------------------
 * Every MMFS_* variable the library reads is a tuning / test hook (which formulation runs, how a launch is cut; none changes a
 * result).  They are ONE table, read once at the first use; no launch path calls getenv().  (The reference has no counterpart:
 * its extension reads no environment, ops/src/vision.cpp:14-15.)
 *   mmfs_env_reload: read the environment again (a test that flips a knob inside one process says so).
 *   mmfs_env_knob:   entry ``index`` of the table (0 ... until MMFS_E_DIMS): its name, one line of documentation, and the value
 *                    the library holds for it (NULL: unset).  Any of the three pointers may be NULL. */
void mmfs_env_reload(void);
int mmfs_env_knob(int index, const char **name, const char **doc, const char **value);

#ifdef __cplusplus
}
#endif
#endif /* MMFS_MSDA_H_ */
