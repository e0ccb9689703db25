// Micro-benchmarks that settle design questions for the backward (MI355X):
//   1. LDS float atomic rate (ds_add_f32), conflict-free lanes
//   2. LDS read-modify-write by plain ds_read/ds_write
//   3. global float atomic rate, distinct addresses, L2-resident footprint
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomics.hip -o atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(256) lds_atomic(float *out, int iters, int px)
{
    extern __shared__ float acc[];
    for (int i = threadIdx.x; i < px * 128; i += 256) acc[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned s = blockIdx.x * 7919u + wave * 104729u + 1u;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const int p = (s >> 8) % px;                 // wave-uniform random pixel
        float *row = acc + p * 128;
        const float v = (float)(it & 3);
#pragma unroll
        for (int c = 0; c < 4; ++c) {                // 4 "corners" x 2 channel halves
            atomicAdd(row + lane, v);
            atomicAdd(row + 64 + lane, v);
            row = acc + ((p + c + 1) % px) * 128;
        }
    }
    __syncthreads();
    float t = 0.f;
    for (int i = threadIdx.x; i < px * 128; i += 256) t += acc[i];
    if (t == 12345.f) out[0] = t;
}

__global__ void __launch_bounds__(256) lds_rmw(float *out, int iters, int px)
{
    extern __shared__ float acc[];
    for (int i = threadIdx.x; i < px * 128; i += 256) acc[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned s = blockIdx.x * 7919u + wave * 104729u + 1u;
    // each wave owns px/4 pixels: plain read-add-write, no atomics needed
    const int own = px / 4;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const int p = wave * own + (s >> 8) % own;
        float2 *row = reinterpret_cast<float2 *>(acc + p * 128);
        const float v = (float)(it & 3);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float2 a = row[lane];
            a.x += v; a.y += v;
            row[lane] = a;
            row = reinterpret_cast<float2 *>(acc + (wave * own + (p + c + 1) % own) * 128);
        }
    }
    __syncthreads();
    float t = 0.f;
    for (int i = threadIdx.x; i < px * 128; i += 256) t += acc[i];
    if (t == 12345.f) out[0] = t;
}

template <int SCOPE>
__global__ void __launch_bounds__(256) glob_atomic(float *buf, size_t n_rows, int iters)
{
    const int lane = threadIdx.x & 63;
    unsigned s = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2654435761u + 1u;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        float *row = buf + (size_t)((s >> 4) % n_rows) * 128;
        __hip_atomic_fetch_add(row + lane, 1.f, __ATOMIC_RELAXED, SCOPE);
        __hip_atomic_fetch_add(row + 64 + lane, 1.f, __ATOMIC_RELAXED, SCOPE);
    }
}

int main()
{
    float *out; CK(hipMalloc(&out, 1 << 20));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms;
    const int blocks = 512, iters = 4096;
    for (int px : {64, 128, 256}) {
        const size_t lds = (size_t)px * 128 * 4;
        CK(hipFuncSetAttribute((const void *)lds_atomic, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CK(hipFuncSetAttribute((const void *)lds_rmw, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(lds_atomic, dim3(blocks), dim3(256), lds, 0, out, iters, px);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        }
        double lane_ops = (double)blocks * 4 * iters * 8 * 64;
        printf("lds_atomic px=%3d (%3zu KB): %.3f ms  %.1f G lane-adds/s  (%.2f lane-adds/clk/CU @2.4GHz, 256 CUs)\n",
               px, lds >> 10, ms, lane_ops / ms / 1e6, lane_ops / (ms * 1e-3) / 256 / 2.4e9);
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(lds_rmw, dim3(blocks), dim3(256), lds, 0, out, iters, px);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        }
        lane_ops = (double)blocks * 4 * iters * 4 * 128;
        printf("lds_rmw    px=%3d (%3zu KB): %.3f ms  %.1f G elem-adds/s (%.2f elem-adds/clk/CU)\n",
               px, lds >> 10, ms, lane_ops / ms / 1e6, lane_ops / (ms * 1e-3) / 256 / 2.4e9);
    }
    for (size_t rows : {(size_t)4096, (size_t)65536, (size_t)1 << 20}) {
        float *buf; CK(hipMalloc(&buf, rows * 512)); CK(hipMemset(buf, 0, rows * 512));
        const int git = 512, gblocks = 2048;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL((glob_atomic<__HIP_MEMORY_SCOPE_AGENT>), dim3(gblocks), dim3(256), 0, 0, buf, rows, git);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        }
        double ops = (double)gblocks * 4 * git * 128;
        printf("global_atomic agent rows=%8zu (%6zu KB): %.3f ms  %.1f G adds/s\n", rows, rows * 512 >> 10, ms, ops / ms / 1e6);
        CK(hipFree(buf));
    }
    return 0;
}
