// msda_capi.hip -- the extern "C" surface declared in include/mmfs_msda.h.
// Validates arguments, narrows the dims to the kernels' 32-bit index range and
// forwards to the kernel launchers.  No state, no allocation.
#include "../../include/mmfs_msda.h"
#include "msda_env.h"
#include "msda_launch.h"
#include "msda_plan.h"
#include <cstring>
#include <cstdlib>

namespace {

int elem_size(int dtype)
{
    switch (dtype) {
        case MMFS_F32: return 4;
        case MMFS_F16: case MMFS_BF16: return 2;
        case MMFS_F64: return 8;
        default: return 0;
    }
}

// Fills d; returns MMFS_OK, or an error.  *empty is set when there is nothing to launch.
int make_dims(int64_t B, int64_t S, int64_t H, int64_t D, int64_t L, int64_t Nq, int64_t P,
              mmfs::Dims *d)
{
    const int64_t lim = 0x7fffffffLL;
    if (B < 0 || S < 0 || H < 0 || D < 0 || L < 0 || Nq < 0 || P < 0) return MMFS_E_DIMS;
    if (B > lim || S > lim || H > lim || D > lim || L > lim || Nq > lim || P > lim) return MMFS_E_DIMS;
    if (L * P > lim || H * D > lim) return MMFS_E_DIMS;
    // pixel-row indices (level start + y*W + x) are kept as int32 in the tap records
    if (S > lim - 2) return MMFS_E_DIMS;
    // level extents are packed into 16 bits each in the backward's records; checked
    // on the device tables by the host shim (they live in device memory here).
    d->B = (int)B; d->S = (int)S; d->H = (int)H; d->D = (int)D;
    d->L = (int)L; d->Nq = (int)Nq; d->P = (int)P;
    d->K = (int)(L * P);
    d->q_tiles = 0;
    d->lazy_attn = 0;
    d->blocks4 = 0;
    d->table_status = nullptr;
    d->taps_algo = 0;
    d->taps_sorted = 0;
    d->tiles_hint = 0;
    return MMFS_OK;
}

bool misaligned(const void *p, int esize) { return ((uintptr_t)p % (uintptr_t)esize) != 0; }

}  // namespace

extern "C" {

int mmfs_msda_abi_version(void) { return MMFS_MSDA_ABI_VERSION; }

const char *mmfs_msda_build_info(void)
{
    return "libmmfs_msda gfx950 (CDNA4) hip " __VERSION__;
}

const char *mmfs_msda_status_string(int status)
{
    switch (status) {
        case MMFS_OK: return "ok";
        case MMFS_E_DTYPE: return "mmfs_msda: unknown dtype code";
        case MMFS_E_DIMS: return "mmfs_msda: negative or out-of-range dimension";
        case MMFS_E_NULLPTR: return "mmfs_msda: NULL pointer for a non-empty tensor";
        case MMFS_E_ALIGN: return "mmfs_msda: tensor base pointer not aligned to 16 bytes / element size";
        case MMFS_E_UNSUPPORTED: return "mmfs_msda: unsupported configuration";
        default: break;
    }
    if (status > 0) return hipGetErrorString((hipError_t)status);
    return "mmfs_msda: unknown status";
}

int mmfs_msda_forward(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                      const void *loc, const void *attn, void *out,
                      int64_t B, int64_t S, int64_t H, int64_t D, int64_t L, int64_t Nq, int64_t P,
                      void *stream)
{
    return mmfs_msda_forward_flags(dtype, value, shapes, start, loc, attn, out, B, S, H, D, L, Nq, P, 0u, stream);
}

int mmfs_msda_forward_flags(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                            const void *loc, const void *attn, void *out,
                            int64_t B, int64_t S, int64_t H, int64_t D, int64_t L, int64_t Nq, int64_t P,
                            unsigned flags, void *stream)
{
    const int es = elem_size(dtype);
    if (!es) return MMFS_E_DTYPE;
    mmfs::Dims d;
    const int rc = make_dims(B, S, H, D, L, Nq, P, &d);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n_out = B * Nq * H * D;
    if (n_out == 0) return MMFS_OK;
    if (!out) return MMFS_E_NULLPTR;
    if (d.K == 0 || S == 0)                          // nothing to sample: the op's value is 0
        return (int)mmfs::zero_fill(out, (size_t)n_out * es, st);
    if (!value || !shapes || !start || !loc || !attn) return MMFS_E_NULLPTR;
    // the vector kernels move 16-byte channel vectors; rows are D*es apart, so the
    // bases must be 16-byte aligned whenever D*es is a multiple of 16
    const int al = ((D * es) % 16 == 0) ? 16 : es;
    if (misaligned(value, al) || misaligned(out, al) || misaligned(loc, es) || misaligned(attn, es))
        return MMFS_E_ALIGN;
    if ((flags & MMFS_FWD_LDS_LEVELS) && !mmfs::fwd_mma_supported(dtype, d)) return MMFS_E_UNSUPPORTED;
    if ((flags & MMFS_FWD_SLICES) && !mmfs::fwd_q8_supported(dtype, d)) return MMFS_E_UNSUPPORTED;
    if ((flags & MMFS_FWD_QUERY_WAVES) && !mmfs::fwd_wq_supported(dtype, d)) return MMFS_E_UNSUPPORTED;
    const int algo = (flags & MMFS_FWD_QUERY_WAVES) ? 4 : (flags & MMFS_FWD_SLICES) ? 3 : (flags & MMFS_FWD_LDS_LEVELS) ? 2 : (flags & MMFS_FWD_ROW_GATHER) ? 1 : 0;
    return (int)mmfs::forward(dtype, value, shapes, start, loc, attn, out, d, st, algo);
}

static bool use_tiled(int dtype, const mmfs::Dims &d, unsigned flags)
{
    if (flags & MMFS_BWD_FORCE_ATOMIC) return false;
    if (!mmfs::bwd_has_vector_path(dtype, d) || !mmfs::bwd_value_tiled_supported(dtype, d)) return false;
    if (flags & MMFS_BWD_CANONICAL_LEVELS) return true;
    // nobody has looked at the table: the block-stationary generation checks it on the device
    return (flags & MMFS_BWD_DEVICE_CHECKED_LEVELS) && mmfs::bwd_value_block_supported(dtype, d);
}

int mmfs_msda_backward_taps_fused(int dtype, int64_t B, int64_t S, int64_t H, int64_t D,
                                  int64_t L, int64_t Nq, int64_t P, unsigned flags)
{
    mmfs::Dims d;
    if (!elem_size(dtype) || make_dims(B, S, H, D, L, Nq, P, &d)) return 0;
    d.taps_algo = (flags & MMFS_BWD_TAPS_LDS_LEVELS) ? 2 : (flags & MMFS_BWD_TAPS_ROW_GATHER) ? 1 : 0;
    return mmfs::taps_mma_applies(dtype, d) ? 1 : 0;
}

int64_t mmfs_msda_backward_workspace_bytes(int dtype, int64_t B, int64_t S, int64_t H, int64_t D,
                                           int64_t L, int64_t Nq, int64_t P, unsigned flags)
{
    mmfs::Dims d;
    if (!elem_size(dtype) || make_dims(B, S, H, D, L, Nq, P, &d)) return 0;
    if (use_tiled(dtype, d, flags)) {
        const int64_t tiled = mmfs::bwd_value_tiled_workspace_bytes(dtype, d);
        // a table nobody has looked at may be one the sorted backward refuses: room for the float-atomic fallback's
        // fp32 image behind the sorted backward's pieces (msda_bwd_refused.hip)
        if (!(flags & MMFS_BWD_CANONICAL_LEVELS)) return (tiled + 255) / 256 * 256 + mmfs::refused_table_scratch_bytes(dtype, d);
        return tiled;
    }
    // atomic path: 16-bit storage accumulates into an fp32 image of grad_value
    if (dtype == MMFS_F16 || dtype == MMFS_BF16) return B * S * H * D * 4;
    return 0;
}

int mmfs_msda_backward(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                       const void *loc, const void *attn, const void *grad_out,
                       void *grad_value, void *grad_loc, void *grad_attn,
                       void *workspace, int64_t workspace_bytes,
                       int64_t B, int64_t S, int64_t H, int64_t D, int64_t L, int64_t Nq, int64_t P,
                       unsigned flags, void *stream)
{
    // a table nobody has looked at needs somewhere to report what the device finds
    if (flags & MMFS_BWD_DEVICE_CHECKED_LEVELS) return MMFS_E_UNSUPPORTED;
    return mmfs_msda_backward_checked(dtype, value, shapes, start, loc, attn, grad_out, grad_value, grad_loc, grad_attn,
                                      workspace, workspace_bytes, B, S, H, D, L, Nq, P, flags, nullptr, stream);
}

int mmfs_msda_backward_checked(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                               const void *loc, const void *attn, const void *grad_out,
                               void *grad_value, void *grad_loc, void *grad_attn,
                               void *workspace, int64_t workspace_bytes,
                               int64_t B, int64_t S, int64_t H, int64_t D, int64_t L, int64_t Nq, int64_t P,
                               unsigned flags, int32_t *table_status, void *stream)
{
    const int es = elem_size(dtype);
    if (!es) return MMFS_E_DTYPE;
    mmfs::Dims d;
    const int rc = make_dims(B, S, H, D, L, Nq, P, &d);
    if (rc) return rc;
    if ((flags & MMFS_BWD_DEVICE_CHECKED_LEVELS) && !table_status) return MMFS_E_NULLPTR;
    d.table_status = (flags & MMFS_BWD_DEVICE_CHECKED_LEVELS) ? table_status : nullptr;
    d.lazy_attn = (flags & MMFS_BWD_LAZY_ZERO_ATTN) ? 1 : 0;
    d.taps_algo = (flags & MMFS_BWD_TAPS_LDS_LEVELS) ? 2 : (flags & MMFS_BWD_TAPS_ROW_GATHER) ? 1 : 0;
    if (d.taps_algo == 2 && !(mmfs::taps_mma_supported(dtype, d) && use_tiled(dtype, d, flags))) return MMFS_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n_samples = B * Nq * H * L * P;
    const int64_t n_value = B * S * H * D;
    if (n_value && !grad_value) return MMFS_E_NULLPTR;
    if (n_samples == 0 || S == 0 || D == 0) {           // nothing flows: all gradients are zero
        hipError_t e = hipSuccess;
        if (n_value) e = mmfs::zero_fill(grad_value, (size_t)n_value * es, st);
        if (e == hipSuccess && n_samples) {
            if (!grad_loc || !grad_attn) return MMFS_E_NULLPTR;
            e = mmfs::zero_fill(grad_loc, (size_t)n_samples * 2 * es, st);
            if (e == hipSuccess) e = mmfs::zero_fill(grad_attn, (size_t)n_samples * es, st);
        }
        return (int)e;
    }
    if (!value || !shapes || !start || !loc || !attn || !grad_out || !grad_loc || !grad_attn)
        return MMFS_E_NULLPTR;
    const int al = ((D * es) % 16 == 0) ? 16 : es;
    if (misaligned(value, al) || misaligned(grad_out, al) || misaligned(grad_value, al) ||
        misaligned(loc, es) || misaligned(attn, es) || misaligned(grad_loc, es) || misaligned(grad_attn, es))
        return MMFS_E_ALIGN;

    if (use_tiled(dtype, d, flags)) {
        const bool canonical = (flags & MMFS_BWD_CANONICAL_LEVELS) != 0;
        const int64_t tiled_bytes = mmfs::bwd_value_tiled_workspace_bytes(dtype, d);
        const int64_t acc_off = (tiled_bytes + 255) / 256 * 256;
        if (!workspace || workspace_bytes < (canonical ? tiled_bytes : acc_off + mmfs::refused_table_scratch_bytes(dtype, d)))
            return MMFS_E_NULLPTR;
        if (misaligned(workspace, 16)) return MMFS_E_ALIGN;
        // a table the device-side check refuses gets the reference's float-atomic scatter IN THIS CALL (three launches
        // that return at once otherwise): grad_value is right either way, the status word only says which path ran
        auto finish = [&](hipError_t e) {
            if (e != hipSuccess || canonical) return (int)e;
            return (int)mmfs::backward_value_refused_table(dtype, shapes, start, loc, attn, grad_out, grad_value, workspace,
                                                           reinterpret_cast<float *>((char *)workspace + acc_off), d, st);
        };
        // (the grad_value half's opening launch hosted by the LDS-levels taps kernel where that one runs: msda_plan.h)
        mmfs::blk::PrepareJob job;
        if (mmfs::taps_mma_applies(dtype, d) && mmfs::value_prepare_job(dtype, loc, attn, shapes, start, workspace, d, &job)) {
            hipError_t e = mmfs::backward_taps_mma(dtype, value, shapes, start, loc, attn, grad_out, grad_loc, grad_attn, d, st, &job);
            if (e != hipSuccess) return (int)e;
            return finish(mmfs::backward_value_run(dtype, shapes, start, grad_out, grad_value, workspace, d, st, true, canonical));
        }
        hipError_t e = mmfs::backward_taps(dtype, value, shapes, start, loc, attn, grad_out,
                                           nullptr, grad_loc, grad_attn, d, false, st);
        if (e != hipSuccess) return (int)e;
        return finish(mmfs::backward_value_tiled(dtype, shapes, start, loc, attn, grad_out, grad_value,
                                                 workspace, d, st, canonical));
    }
    // ---- float-atomic path
    const bool narrow = (dtype == MMFS_F16 || dtype == MMFS_BF16);
    void *acc = grad_value;
    size_t acc_bytes = (size_t)n_value * es;
    if (narrow) {
        if (workspace_bytes < n_value * 4 || !workspace) return MMFS_E_NULLPTR;
        if (misaligned(workspace, 16)) return MMFS_E_ALIGN;
        acc = workspace;
        acc_bytes = (size_t)n_value * 4;
    }
    hipError_t e = mmfs::zero_fill(acc, acc_bytes, st);              // reference: at::zeros, .cu:127
    if (e != hipSuccess) return (int)e;
    e = mmfs::backward_taps(dtype, value, shapes, start, loc, attn, grad_out, acc, grad_loc, grad_attn,
                            d, true, st);
    if (e != hipSuccess) return (int)e;
    if (narrow) e = mmfs::cast_from_f32(dtype, (const float *)acc, grad_value, n_value, st);   // .cu:156-165
    return (int)e;
}

int mmfs_msda_backward_taps(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                            const void *loc, const void *attn, const void *grad_out,
                            void *grad_loc, void *grad_attn,
                            int64_t B, int64_t S, int64_t H, int64_t D, int64_t L, int64_t Nq, int64_t P,
                            void *stream)
{
    const int es = elem_size(dtype);
    if (!es) return MMFS_E_DTYPE;
    mmfs::Dims d;
    const int rc = make_dims(B, S, H, D, L, Nq, P, &d);
    if (rc) return rc;
    if (B * Nq * H * L * P == 0) return MMFS_OK;
    if (S == 0 || D == 0 || !mmfs::bwd_has_vector_path(dtype, d)) return MMFS_E_UNSUPPORTED;
    if (!value || !shapes || !start || !loc || !attn || !grad_out || !grad_loc || !grad_attn)
        return MMFS_E_NULLPTR;
    if (misaligned(value, 16) || misaligned(grad_out, 16) || misaligned(loc, es) || misaligned(attn, es) ||
        misaligned(grad_loc, es) || misaligned(grad_attn, es))
        return MMFS_E_ALIGN;
    return (int)mmfs::backward_taps(dtype, value, shapes, start, loc, attn, grad_out, nullptr, grad_loc,
                                    grad_attn, d, false, (hipStream_t)stream);
}

int mmfs_msda_backward_value(int dtype, const int64_t *shapes, const int64_t *start,
                             const void *loc, const void *attn, const void *grad_out, void *grad_value,
                             void *workspace, int64_t workspace_bytes,
                             int64_t B, int64_t S, int64_t H, int64_t D, int64_t L, int64_t Nq, int64_t P,
                             void *stream)
{
    const int es = elem_size(dtype);
    if (!es) return MMFS_E_DTYPE;
    mmfs::Dims d;
    const int rc = make_dims(B, S, H, D, L, Nq, P, &d);
    if (rc) return rc;
    const int64_t n_value = B * S * H * D;
    if (n_value == 0) return MMFS_OK;
    if (!grad_value) return MMFS_E_NULLPTR;
    if (B * Nq * H * L * P == 0)
        return (int)mmfs::zero_fill(grad_value, (size_t)n_value * es, (hipStream_t)stream);
    if (!mmfs::bwd_value_tiled_supported(dtype, d)) return MMFS_E_UNSUPPORTED;
    if (!shapes || !start || !loc || !attn || !grad_out) return MMFS_E_NULLPTR;
    if (misaligned(grad_out, 16) || misaligned(grad_value, 16) || misaligned(loc, es) || misaligned(attn, es))
        return MMFS_E_ALIGN;
    if (!workspace || workspace_bytes < mmfs::bwd_value_tiled_workspace_bytes(dtype, d)) return MMFS_E_NULLPTR;
    if (misaligned(workspace, 16)) return MMFS_E_ALIGN;
    return (int)mmfs::backward_value_tiled(dtype, shapes, start, loc, attn, grad_out, grad_value, workspace, d,
                                           (hipStream_t)stream);
}

// The two halves of mmfs_msda_backward_value (so the kernel proper can be timed on its own).
int mmfs_msda_backward_value_prepare(int dtype, const void *loc, const void *attn,
                                     void *workspace, int64_t workspace_bytes,
                                     int64_t B, int64_t S, int64_t H, int64_t D, int64_t L, int64_t Nq, int64_t P,
                                     void *stream)
{
    const int es = elem_size(dtype);
    if (!es) return MMFS_E_DTYPE;
    mmfs::Dims d;
    const int rc = make_dims(B, S, H, D, L, Nq, P, &d);
    if (rc) return rc;
    if (B * Nq * H * L * P == 0 || B * S * H * D == 0) return MMFS_OK;
    if (!mmfs::bwd_value_tiled_supported(dtype, d)) return MMFS_E_UNSUPPORTED;
    if (!loc || !attn) return MMFS_E_NULLPTR;
    if (misaligned(loc, es) || misaligned(attn, es)) return MMFS_E_ALIGN;
    if (!workspace || workspace_bytes < mmfs::bwd_value_tiled_workspace_bytes(dtype, d)) return MMFS_E_NULLPTR;
    if (misaligned(workspace, 16)) return MMFS_E_ALIGN;
    return (int)mmfs::backward_value_prepare(dtype, loc, attn, workspace, d, (hipStream_t)stream);
}

// common argument checks of the sort / reduce stages; returns > 0 when there is work to launch
static int value_stage_args(int dtype, int64_t B, int64_t S, int64_t H, int64_t D, int64_t L, int64_t Nq,
                            int64_t P, void *workspace, int64_t workspace_bytes, mmfs::Dims *d)
{
    if (!elem_size(dtype)) return MMFS_E_DTYPE;
    const int rc = make_dims(B, S, H, D, L, Nq, P, d);
    if (rc) return rc;
    if (B * S * H * D == 0 || B * Nq * H * L * P == 0) return 0;
    if (!mmfs::bwd_value_tiled_supported(dtype, *d)) return MMFS_E_UNSUPPORTED;
    if (!workspace || workspace_bytes < mmfs::bwd_value_tiled_workspace_bytes(dtype, *d)) return MMFS_E_NULLPTR;
    if (misaligned(workspace, 16)) return MMFS_E_ALIGN;
    return 1;
}

int mmfs_msda_backward_value_sort(int dtype, const int64_t *shapes, const int64_t *start,
                                  void *workspace, int64_t workspace_bytes,
                                  int64_t B, int64_t S, int64_t H, int64_t D, int64_t L, int64_t Nq, int64_t P,
                                  void *stream)
{
    mmfs::Dims d;
    const int rc = value_stage_args(dtype, B, S, H, D, L, Nq, P, workspace, workspace_bytes, &d);
    if (rc <= 0) return rc;
    if (!shapes || !start) return MMFS_E_NULLPTR;
    return (int)mmfs::backward_value_sort(dtype, shapes, start, workspace, d, (hipStream_t)stream);
}

int mmfs_msda_backward_value_reduce(int dtype, const void *grad_out, void *grad_value,
                                    void *workspace, int64_t workspace_bytes,
                                    int64_t B, int64_t S, int64_t H, int64_t D, int64_t L, int64_t Nq, int64_t P,
                                    void *stream)
{
    const int es = elem_size(dtype);
    if (!es) return MMFS_E_DTYPE;
    const int64_t n_value = B * S * H * D;
    if (n_value > 0 && !grad_value) return MMFS_E_NULLPTR;
    if (n_value > 0 && B * Nq * H * L * P == 0)
        return (int)mmfs::zero_fill(grad_value, (size_t)n_value * es, (hipStream_t)stream);
    mmfs::Dims d;
    const int rc = value_stage_args(dtype, B, S, H, D, L, Nq, P, workspace, workspace_bytes, &d);
    if (rc <= 0) return rc;
    if (!grad_out) return MMFS_E_NULLPTR;
    if (misaligned(grad_out, 16) || misaligned(grad_value, 16)) return MMFS_E_ALIGN;
    return (int)mmfs::backward_value_reduce(dtype, grad_out, grad_value, workspace, d, (hipStream_t)stream);
}

int mmfs_msda_backward_value_run(int dtype, const int64_t *shapes, const int64_t *start,
                                 const void *grad_out, void *grad_value,
                                 void *workspace, int64_t workspace_bytes,
                                 int64_t B, int64_t S, int64_t H, int64_t D, int64_t L, int64_t Nq, int64_t P,
                                 void *stream)
{
    const int rc = mmfs_msda_backward_value_sort(dtype, shapes, start, workspace, workspace_bytes,
                                                 B, S, H, D, L, Nq, P, stream);
    if (rc) return rc;
    return mmfs_msda_backward_value_reduce(dtype, grad_out, grad_value, workspace, workspace_bytes,
                                           B, S, H, D, L, Nq, P, stream);
}

// ---------------------------------------------------------------- hybrid entry points
static mmfs::HybridPlan hybrid_plan(int dtype, const mmfs::Dims &d, const int64_t *host_shapes,
                                    const int64_t *host_start)
{
    return mmfs::make_hybrid_plan(dtype, d, host_shapes, host_start);
}

int64_t mmfs_msda_backward_hybrid_workspace_bytes(int dtype, const int64_t *host_shapes, const int64_t *host_start,
                                                  int64_t B, int64_t S, int64_t H, int64_t D,
                                                  int64_t L, int64_t Nq, int64_t P, unsigned flags)
{
    mmfs::Dims d;
    if (!elem_size(dtype) || make_dims(B, S, H, D, L, Nq, P, &d)) return 0;
    if (B * Nq * H * L * P == 0 || S == 0 || D == 0 || !use_tiled(dtype, d, flags)) return 0;
    const mmfs::HybridPlan plan = hybrid_plan(dtype, d, host_shapes, host_start);
    // (something for this entry point to do: a level for the dense taps product)
    if (!((flags & MMFS_BWD_DENSE_TAPS) && plan.dots_active)) return 0;
    return (mmfs::bwd_value_tiled_workspace_bytes(dtype, d) + 255) / 256 * 256;
}

int mmfs_msda_backward_hybrid(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                              const int64_t *host_shapes, const int64_t *host_start,
                              const void *loc, const void *attn, const void *grad_out,
                              void *grad_value, void *grad_loc, void *grad_attn,
                              void *workspace, int64_t workspace_bytes,
                              int64_t B, int64_t S, int64_t H, int64_t D, int64_t L, int64_t Nq, int64_t P,
                              unsigned flags, unsigned stages, void *stream)
{
    const int es = elem_size(dtype);
    if (!es) return MMFS_E_DTYPE;
    mmfs::Dims d;
    const int rc = make_dims(B, S, H, D, L, Nq, P, &d);
    if (rc) return rc;
    if (B * Nq * H * L * P == 0 || S == 0 || D == 0 || !use_tiled(dtype, d, flags)) return MMFS_E_UNSUPPORTED;
    d.lazy_attn = (flags & MMFS_BWD_LAZY_ZERO_ATTN) ? 1 : 0;
    d.taps_algo = (flags & MMFS_BWD_TAPS_LDS_LEVELS) ? 2 : (flags & MMFS_BWD_TAPS_ROW_GATHER) ? 1 : 0;
    if (d.taps_algo == 2 && !mmfs::taps_mma_supported(dtype, d)) return MMFS_E_UNSUPPORTED;
    bool sorted_levels = !host_shapes;          // some level owns pixels: the sort + tile reduce have something to do
    if (host_shapes) {          // exact grid for the matrix-core grad_value reduce (else a bound is launched)
        int64_t nb4 = 0;
        for (int64_t l = 0; l < L; ++l)
            if (host_shapes[2 * l] > 0 && host_shapes[2 * l + 1] > 0) {
                nb4 += ((host_shapes[2 * l] + 3) / 4) * ((host_shapes[2 * l + 1] + 3) / 4);
                sorted_levels = true;
            }
        d.blocks4 = (int)std::min<int64_t>(nb4, 0x3fffffff);
        d.tiles_hint = mmfs::sort_tiles_exact(dtype, d, host_shapes);
    }
    const mmfs::HybridPlan plan = hybrid_plan(dtype, d, host_shapes, host_start);
    const bool dense_taps = (flags & MMFS_BWD_DENSE_TAPS) && plan.dots_active;
    if (!dense_taps) return MMFS_E_UNSUPPORTED;
    if (!value || !shapes || !start || !loc || !attn || !grad_out || !grad_value || !grad_loc || !grad_attn)
        return MMFS_E_NULLPTR;
    if (misaligned(value, 16) || misaligned(grad_out, 16) || misaligned(grad_value, 16) ||
        misaligned(loc, es) || misaligned(attn, es) || misaligned(grad_loc, es) || misaligned(grad_attn, es))
        return MMFS_E_ALIGN;
    const int64_t base = (mmfs::bwd_value_tiled_workspace_bytes(dtype, d) + 255) / 256 * 256;
    if (!workspace || workspace_bytes < base) return MMFS_E_NULLPTR;
    if (misaligned(workspace, 16)) return MMFS_E_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    // (every level dense -- e.g. the ViT-Adapter extractor's single 16x16 map -- leaves the row-gather
    // kernel nothing to write: not launched)
    // (one kernel for every level where the LDS-resident formulation applies: the "fine" stage then does all the
    // levels and the "coarse" stage has nothing left)
    const bool fused_taps = mmfs::taps_mma_applies(dtype, d);
    // (when this call runs both halves and the grad_value half would open with nothing but "clear the cursors, plan",
    // the first workgroup of the grad_loc / grad_attn kernel does that on the side: one launch less)
    mmfs::blk::PrepareJob job;
    // (... or, with heads the LDS-levels kernel does not take, the first workgroup of the dense-levels kernel)
    const bool host_mma = fused_taps && (stages & MMFS_HYB_BWD_TAPS_FINE);
    const bool host_dense = !fused_taps && dense_taps && (stages & MMFS_HYB_BWD_TAPS_COARSE);
    const bool folded = (host_mma || host_dense) && (stages & MMFS_HYB_BWD_VALUE_PREPARE) && sorted_levels &&
                        mmfs::value_prepare_job(dtype, loc, attn, shapes, start, workspace, d, &job);
    if (e == hipSuccess && (stages & MMFS_HYB_BWD_TAPS_FINE) && fused_taps)
        e = mmfs::backward_taps_mma(dtype, value, shapes, start, loc, attn, grad_out, grad_loc, grad_attn, d, st, folded ? &job : nullptr);
    if (e == hipSuccess && (stages & MMFS_HYB_BWD_TAPS_FINE) && !fused_taps && !(dense_taps && plan.fine_taps.n == 0))
        e = mmfs::backward_taps(dtype, value, shapes, start, loc, attn, grad_out, nullptr, grad_loc, grad_attn,
                                d, false, st, dense_taps ? &plan.fine_taps : nullptr);
    if (e == hipSuccess && dense_taps && !fused_taps && (stages & MMFS_HYB_BWD_TAPS_COARSE))
        e = mmfs::backward_taps_coarse(dtype, value, loc, attn, grad_out, grad_loc, grad_attn, d, plan, st,
                                       folded && host_dense ? &job : nullptr);
    // (the plan rides in the prepare launch only when this very call also sorts; the hybrid path needs
    // MMFS_BWD_CANONICAL_LEVELS, so every grad_value row has an owner and no zero-fill pass is due)
    // (the prepare stage always plans: a staged pass then runs the very kernels of the one-call pass)
    bool planned = false;
    if (e == hipSuccess && (stages & MMFS_HYB_BWD_VALUE_PREPARE) && sorted_levels && !folded)
        e = mmfs::backward_value_prepare(dtype, loc, attn, workspace, d, st, shapes, start, &planned);
    if (e == hipSuccess && (stages & MMFS_HYB_BWD_VALUE_SORT) && sorted_levels)
        e = mmfs::backward_value_sort(dtype, shapes, start, workspace, d, st, true);
    if (e == hipSuccess && (stages & MMFS_HYB_BWD_VALUE_REDUCE) && sorted_levels)
        e = mmfs::backward_value_reduce(dtype, grad_out, grad_value, workspace, d, st, true);
    return (int)e;
}

// ---------------------------------------------------------------- the backward on the cell-sorted records
static bool sorted_route(int dtype, mmfs::Dims &d, unsigned flags)
{
    if (!(flags & MMFS_BWD_CANONICAL_LEVELS) || (flags & (MMFS_BWD_FORCE_ATOMIC | MMFS_BWD_TAPS_ROW_GATHER | MMFS_BWD_TAPS_LDS_LEVELS))) return false;
    if (!use_tiled(dtype, d, flags)) return false;
    d.lazy_attn = (flags & MMFS_BWD_LAZY_ZERO_ATTN) ? 1 : 0;
    d.taps_sorted = 1;
    return mmfs::taps_sorted_supported(dtype, d);
}

int64_t mmfs_msda_backward_sorted_workspace_bytes(int dtype, int64_t B, int64_t S, int64_t H, int64_t D,
                                                  int64_t L, int64_t Nq, int64_t P, unsigned flags)
{
    mmfs::Dims d;
    if (!elem_size(dtype) || make_dims(B, S, H, D, L, Nq, P, &d)) return 0;
    if (B * Nq * H * L * P == 0 || S == 0 || D == 0 || !sorted_route(dtype, d, flags)) return 0;
    return mmfs::bwd_value_block_workspace_bytes(dtype, d);
}

int mmfs_msda_backward_sorted(int dtype, const void *value, const int64_t *shapes, const int64_t *start,
                              const void *loc, const void *attn, const void *grad_out,
                              void *grad_value, void *grad_loc, void *grad_attn,
                              void *workspace, int64_t workspace_bytes,
                              int64_t B, int64_t S, int64_t H, int64_t D, int64_t L, int64_t Nq, int64_t P, int64_t blocks4,
                              unsigned flags, unsigned stages, void *stream)
{
    const int es = elem_size(dtype);
    if (!es) return MMFS_E_DTYPE;
    mmfs::Dims d;
    const int rc = make_dims(B, S, H, D, L, Nq, P, &d);
    if (rc) return rc;
    if (B * Nq * H * L * P == 0 || S == 0 || D == 0 || !sorted_route(dtype, d, flags)) return MMFS_E_UNSUPPORTED;
    if (blocks4 < 0) return MMFS_E_DIMS;
    d.blocks4 = (int)std::min<int64_t>(blocks4, 0x3fffffff);
    if (!value || !shapes || !start || !loc || !attn || !grad_out || !grad_value || !grad_loc || !grad_attn)
        return MMFS_E_NULLPTR;
    if (misaligned(value, 16) || misaligned(grad_out, 16) || misaligned(grad_value, 16) ||
        misaligned(loc, es) || misaligned(attn, es) || misaligned(grad_loc, 2 * es) || misaligned(grad_attn, es))
        return MMFS_E_ALIGN;
    if (!workspace || workspace_bytes < mmfs::bwd_value_block_workspace_bytes(dtype, d)) return MMFS_E_NULLPTR;
    if (misaligned(workspace, 16)) return MMFS_E_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    if (stages & MMFS_SRT_BWD_PREPARE)
        e = mmfs::backward_value_block_prepare(dtype, loc, attn, shapes, start, workspace, d, st);
    if (e == hipSuccess && (stages & MMFS_SRT_BWD_SORT))
        e = mmfs::backward_value_block_sort(dtype, shapes, start, workspace, d, true, st, grad_loc, grad_attn);
    if (e == hipSuccess && (stages & MMFS_SRT_BWD_TAPS))
        e = mmfs::backward_taps_sorted(dtype, value, grad_out, grad_loc, grad_attn, workspace, d, st);
    if (e == hipSuccess && (stages & MMFS_SRT_BWD_REDUCE))
        e = mmfs::backward_value_block_reduce(dtype, grad_out, grad_value, workspace, d, true, st);
    return (int)e;
}

int mmfs_msda_cast_from_f32(int dtype, const float *src, void *dst, int64_t n, void *stream)
{
    if (dtype != MMFS_F32 && dtype != MMFS_F16 && dtype != MMFS_BF16) return MMFS_E_DTYPE;
    if (n < 0) return MMFS_E_DIMS;
    if (n == 0) return MMFS_OK;
    if (!src || !dst) return MMFS_E_NULLPTR;
    if (misaligned(src, 16) || misaligned(dst, 8)) return MMFS_E_ALIGN;
    return (int)mmfs::cast_from_f32(dtype, src, dst, n, (hipStream_t)stream);
}

}  // extern "C"
