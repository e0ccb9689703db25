// LDS integer atomic rates (ds_add_u32 / ds_add_rtn_u32) with per-lane random addresses,
// the primitive of an in-LDS counting sort.  Build: hipcc --offload-arch=gfx950 -O3 lds_int.hip -o lds_int
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool RTN>
__global__ void __launch_bounds__(256) k(unsigned *out, int iters, int nctr)
{
    extern __shared__ unsigned ctr[];
    for (int i = threadIdx.x; i < nctr; i += 256) ctr[i] = 0;
    __syncthreads();
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u, acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s = s * 1664525u + 1013904223u;
            const unsigned a = (s >> 10) % nctr;
            if (RTN) acc += atomicAdd(&ctr[a], 1u); else atomicAdd(&ctr[a], 1u);
        }
    }
    __syncthreads();
    unsigned t = acc;
    for (int i = threadIdx.x; i < nctr; i += 256) t += ctr[i];
    if (t == 0xdeadbeef) out[0] = t;
}

int main()
{
    unsigned *out; CK(hipMalloc(&out, 4096));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms;
    const int blocks = 2048, iters = 512;
    for (int nctr : {64, 512, 4096}) {
        for (int rtn = 0; rtn < 2; ++rtn) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(a));
                if (rtn) hipLaunchKernelGGL(k<true>, dim3(blocks), dim3(256), nctr * 4, 0, out, iters, nctr);
                else hipLaunchKernelGGL(k<false>, dim3(blocks), dim3(256), nctr * 4, 0, out, iters, nctr);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
            }
            const double ops = (double)blocks * 256 * iters * 8;
            printf("ds_add%s_u32 counters=%4d: %.3f ms  %.1f G lane-ops/s  (%.2f lane-ops/clk/CU)\n",
                   rtn ? "_rtn" : "    ", nctr, ms, ops / ms / 1e6, ops / (ms * 1e-3) / 256 / 2.4e9);
        }
    }
    return 0;
}
