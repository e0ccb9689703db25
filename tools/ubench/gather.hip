// Row-gather micro-benchmark: random 256-byte rows (16 lanes x 16 B) out of N rows spaced STRIDE
// bytes apart -- the access pattern of every hot kernel here (one head's rows inside [.., H, D]).
// Question: does the power-of-two 2 KiB row stride (H*D*2 B) limit L2 channel parallelism?
// Build: hipcc --offload-arch=gfx950 -O3 gather.hip -o gather
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(256) gather(const char *base, size_t region_stride, int n_regions, int nrows,
                                             int stride, int iters, float *out)
{
    const int grp = threadIdx.x >> 4, lig = threadIdx.x & 15;
    // blocks of one XCD (block % 8) share a region, like the head -> XCD affinity of the kernels
    const char *reg = base + (size_t)(blockIdx.x % n_regions) * region_stride;
    unsigned s = (blockIdx.x * 16 + grp) * 2654435761u + 7u;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s = s * 1664525u + 1013904223u;
            const unsigned row = (s >> 8) % (unsigned)nrows;
            v[u] = *reinterpret_cast<const uint4 *>(reg + (size_t)row * stride + lig * 16);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += __uint_as_float(v[u].x) + __uint_as_float(v[u].w);
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main()
{
    const size_t bytes = 512ull << 20;
    char *buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
    float *out; CK(hipMalloc(&out, 64));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int blocks = 256 * 8, iters = 256, nrows = 5440;
    struct Cfg { const char *name; int stride; size_t region_stride; int n_regions; } cfgs[] = {
        {"dense 256B rows, 1 region/XCD (1.4 MB each)", 256, 4u << 20, 8},
        {"stride 2048 (H=8,D=128 bf16), region = head offset 256B", 2048, 256, 8},
        {"stride 2048, 8 separate slabs (16 MB apart)", 2048, 16u << 20, 8},
        {"stride 2304 (non power of two)", 2304, 16u << 20, 8},
        {"stride 2048 + all blocks all heads (n_regions=1)", 2048, 0, 1},
        {"stride 4096 (H=16,D=128)", 4096, 256, 8},
        {"stride 1024 (H=8,D=64 bf16 -> 128B rows x2)", 1024, 256, 8},
    };
    for (auto &c : cfgs) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(gather, dim3(blocks), dim3(256), 0, 0, buf, c.region_stride, c.n_regions, nrows,
                               c.stride, iters, out);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        }
        const double gb = (double)blocks * 16 * iters * 8 * 256 / 1e9;
        printf("%-62s %8.3f ms  %7.1f TB/s  (%.1f B/clk/CU @2.1GHz)\n", c.name, ms, gb / ms, gb / ms * 1e12 / 256 / 2.1e9 / 1e3 * 1e0);
    }
    return 0;
}
