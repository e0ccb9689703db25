// msda_bwd_refused.hip -- grad_value for a level table the device-side check REFUSED, in the same call.
//
// The reference scatters grad_value with one float atomic per (sample, corner, channel) into a zero-filled
// fp32 image and casts at the end (ms_deform_im2col_cuda.cuh:128-155, ms_deform_attn_cuda.cu:122-129, 156-165):
// that serves ANY level table -- overlapping levels add up in the rows they share.  The sorted backward of this
// library needs every grad_value row to have one owner level, so for a table nobody has looked at
// (MMFS_BWD_DEVICE_CHECKED_LEVELS: the reference's callers build fresh level tensors per call and the shim never
// copies them to the host) the plan checks the table on the device.  Until round 3 a refused table left
// grad_value all zeros behind a call that reported success (VERDICT r3 "a call that returns OK with a wrong
// gradient").  Now the checked route ends with three launches that read the plan's verdict ON THE DEVICE and
// return at once when the table was fine (every real caller); for a refused table they are the reference's
// scatter: zero the fp32 image, float atomics, cast.  Slow (one thread per (sample, channel)), never taken by
// the reference's own callers, and correct.
//
// Deviation, on purpose: a corner whose row index falls outside [0, S) is skipped (an out-of-range table makes
// the reference write past the end of grad_value).
#include "msda_launch.h"
#include "msda_bwd_block.h"

namespace mmfs {

const int *value_table_refused_flag(void *workspace, int dtype, const Dims &d);       // msda_bwd_block.hip

namespace {

__global__ void __launch_bounds__(256)
refused_zero(const int *__restrict__ refused, float *__restrict__ acc, int64_t n4)
{
    if (!*refused) return;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
        reinterpret_cast<float4 *>(acc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// one thread per (sample, channel): consecutive threads walk the channels of one sample (coalesced grad_out reads,
// atomics on consecutive floats)
template <typename T>
__global__ void __launch_bounds__(256)
refused_scatter(const int *__restrict__ refused, const int64_t *__restrict__ shapes, const int64_t *__restrict__ start,
                const T *__restrict__ loc, const T *__restrict__ attn, const T *__restrict__ grad_out,
                float *__restrict__ acc, const Dims d)
{
    if (!*refused) return;
    const int64_t n = (int64_t)d.B * d.Nq * d.H * d.K * d.D;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % d.D);
        const int64_t s = i / d.D;                                     // sample: ((b * Nq + q) * H + h) * K + k
        const int k = (int)(s % d.K);
        const int64_t t = s / d.K;
        const int h = (int)(t % d.H);
        const int64_t bq = t / d.H;
        const int64_t b = bq / d.Nq;
        const int l = k / d.P;
        const int64_t Hl = shapes[2 * l], Wl = shapes[2 * l + 1], st = start[l];
        if (Hl <= 0 || Wl <= 0) continue;
        const float lx = to_f32(loc[2 * s]), ly = to_f32(loc[2 * s + 1]), a = to_f32(attn[s]);
        const float y = ly * (float)Hl - 0.5f, x = lx * (float)Wl - 0.5f;
        if (!((y > -1.f) && (x > -1.f) && (y < (float)Hl) && (x < (float)Wl))) continue;     // cuh:291, NaN fails
        const float yf = floorf(y), xf = floorf(x);
        const int64_t y0 = (int64_t)yf, x0 = (int64_t)xf;
        const float fy = y - yf, fx = x - xf, gy = 1.f - fy, gx = 1.f - fx;
        const float g = to_f32(grad_out[(bq * d.H + h) * d.D + c]) * a;         // "top_grad_value", cuh:113
        const float w[4] = {gy * gx, gy * fx, fy * gx, fy * fx};
#pragma unroll
        for (int cn = 0; cn < 4; ++cn) {
            const int64_t yy = y0 + (cn >> 1), xx = x0 + (cn & 1);
            if (yy < 0 || xx < 0 || yy >= Hl || xx >= Wl) continue;
            const int64_t pix = st + yy * Wl + xx;
            if (pix < 0 || pix >= d.S) continue;                                // (a table that points outside value)
            atomicAdd(acc + ((b * d.S + pix) * d.H + h) * d.D + c, w[cn] * g);
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
refused_cast(const int *__restrict__ refused, const float *__restrict__ acc, T *__restrict__ dst, int64_t n)
{
    if (!*refused) return;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = (T)acc[i];
}

template <typename T>
hipError_t run(const int *flag, const int64_t *shapes, const int64_t *start, const void *loc, const void *attn,
               const void *go, void *gv, float *acc, const Dims &d, hipStream_t st)
{
    const int64_t n_value = (int64_t)d.B * d.S * d.H * d.D;
    const int64_t n = (int64_t)d.B * d.Nq * d.H * d.K * d.D;
    // (small grids, grid-stride loops: for every table the sorted backward served -- every real call -- these launches
    // do nothing, and a million workgroups that only read the verdict took ~100 us each: profiles/r04_experiments.md r04a)
    auto grid = [](int64_t work) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>((work + 255) / 256, 2048)); };
    float *image = acc;
    if (sizeof(T) == 4) image = reinterpret_cast<float *>(gv);        // fp32 storage: grad_value itself (the checked
                                                                      // route has zero-filled every row of a refused table)
    else hipLaunchKernelGGL(refused_zero, dim3(grid((n_value + 3) / 4)), dim3(256), 0, st, flag, acc, (n_value + 3) / 4);
    hipLaunchKernelGGL((refused_scatter<T>), dim3(grid(n)), dim3(256), 0, st, flag, shapes, start, (const T *)loc,
                       (const T *)attn, (const T *)go, image, d);
    if (sizeof(T) != 4)
        hipLaunchKernelGGL((refused_cast<T>), dim3(grid(n_value)), dim3(256), 0, st, flag, (const float *)acc, (T *)gv, n_value);
    return hipGetLastError();
}

}  // namespace

int64_t refused_table_scratch_bytes(int dtype, const Dims &d)
{
    if (dtype != 1 && dtype != 2) return 0;
    const int64_t n_value = (int64_t)d.B * d.S * d.H * d.D;
    return ((n_value + 3) / 4 * 16 + 255) / 256 * 256;
}

hipError_t backward_value_refused_table(int dtype, const int64_t *shapes, const int64_t *start, const void *loc,
                                        const void *attn, const void *grad_out, void *grad_value, void *workspace,
                                        float *acc, const Dims &d, hipStream_t st)
{
    const int *flag = value_table_refused_flag(workspace, dtype, d);
    switch (dtype) {
        case 0: return run<float>(flag, shapes, start, loc, attn, grad_out, grad_value, nullptr, d, st);
        case 1: return run<half_t>(flag, shapes, start, loc, attn, grad_out, grad_value, acc, d, st);
        case 2: return run<bf16_t>(flag, shapes, start, loc, attn, grad_out, grad_value, acc, d, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mmfs
