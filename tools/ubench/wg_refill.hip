// wg_refill.hip -- what does it cost to replace a finished one-wave workgroup by the next one?
// The matrix-core grad_value reduce (csrc/msda_bwd_tile.hip) is ~25 000 workgroups of one wave, 10 KB of LDS each, 16 resident per
// CU, ~35 000 clocks of work per workgroup; its counters show 74 % of the wave slots occupied on average (profiles/r06ad_pmc_summary.txt:
// SQ_WAVE_CYCLES against the kernel's duration).  Is the rest the refill of a slot?  Here: the same launch shape, every workgroup spins for
// a fixed number of clocks (optionally after a chain of dependent scalar loads, as the reduce's descriptor fetch is), and the wall time
// against rounds x spin says what a refill costs.  PERSIST: the same work as resident workgroups that loop.
//   hipcc --offload-arch=gfx950 -O3 wg_refill.hip -o /tmp/wg_refill && /tmp/wg_refill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <bool PERSIST, int LDS_BYTES>
__global__ void __launch_bounds__(64, 4) spin(const int *chain, int hops, long long clocks, int total, unsigned long long *sink)
{
    __shared__ unsigned char lds[LDS_BYTES];
    for (int w = blockIdx.x; w < total; w += PERSIST ? (int)gridDim.x : total) {
        int idx = w & 1023;
        for (int h = 0; h < hops; ++h) idx = __builtin_nontemporal_load(chain + ((idx * 64 + h * 4096) & 0xfffff));     // dependent loads, different lines
        lds[threadIdx.x] = (unsigned char)idx;
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < clocks) __builtin_amdgcn_s_sleep(2);
        if (lds[threadIdx.x ^ 1] == 255 && idx == 77) atomicAdd(sink, 1ull);
    }
}

template <bool PERSIST, int LDS_BYTES>
static void run(const char *what, int hops, long long clocks, int rounds, const int *chain, unsigned long long *sink)
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int slots = p.multiProcessorCount * 16;
    const int total = slots * rounds;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((spin<PERSIST, LDS_BYTES>), dim3(PERSIST ? slots : total), dim3(64), 0, 0, chain, hops, clocks, total, sink);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // wall_clock64 (s_memrealtime) ticks at 100 MHz: `clocks` are those ticks
    const double ideal_us = rounds * clocks / 100.0;
    printf("  %-44s %2d rounds x %5.1f us of spin, %d dependent loads first: %7.1f us wall, %6.1f us over rounds x spin = %5.2f us per round\n",
           what, rounds, clocks / 100.0, hops, best * 1e3, best * 1e3 - ideal_us, (best * 1e3 - ideal_us) / rounds);
}

int main()
{
    int *chain = nullptr;
    unsigned long long *sink = nullptr;
    if (hipMalloc(&chain, (1 << 20) * sizeof(int)) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) { printf("no device\n"); return 1; }
    std::vector<int> h(1 << 20);
    for (int i = 0; i < (1 << 20); ++i) h[i] = (i * 2654435761u >> 12) & 1023;
    (void)hipMemcpy(chain, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice);
    (void)hipMemset(sink, 0, 8);
    printf("one-wave workgroups, 16 per CU resident (10 KB of LDS each)\n");
    for (int hops : {0, 3}) {
        run<false, 10240>("a workgroup per item", hops, 1470, 6, chain, sink);       // 14.7 us: a work item of the reduce
        run<true, 10240>("resident workgroups", hops, 1470, 6, chain, sink);
        run<false, 10240>("a workgroup per item", hops, 400, 24, chain, sink);       // 4 us items
        run<true, 10240>("resident workgroups", hops, 400, 24, chain, sink);
        run<false, 10240>("a workgroup per item", hops, 100, 96, chain, sink);       // 1 us items
        run<true, 10240>("resident workgroups", hops, 100, 96, chain, sink);
    }
    return 0;
}
