// (RESOLVED in round 6, after this reproducer came back clean: the fault is a packed-fp32 erratum -- tools/ubench/pk_opsel_mfma.hip,
// profiles/r06_experiments.md r06aa.  This program stays as the record that the hazard it tests does NOT exist on MI355X.)
// lds_store_war.hip -- does a vector instruction that OVERWRITES the data registers of a 16-byte LDS store right behind it
// change what the store writes?  (the sliced forward's round-4 heisenbug "has no root cause".)
//
// Background.  csrc/msda_fwd_q8.hip stages its records with 16-byte LDS stores and, in round 4, produced wrong rows for the
// LAST lanes of a tile (queries 6 and 7) in 0.2-3 % of runs, only with >= 5 busy waves per CU and never under an in-kernel
// check (profiles/r04_experiments.md r04f-m).  The fix that held was timing: 16 idle cycles after the stores, before the
// wave's next vector instruction (its registers were the stores' data registers: the lo fragment's v_and wrote into what
// the hi fragment's store was still reading).  MI355X_MICROARCH.md (LDS): "a store also moves its address and data VGPRs
// to the LDS, at 2 cycles per source dword ... ds_write_b128 13 cycles".  LLVM's hazard recogniser knows the documented
// form of this (a store of more than 64 bits followed by a VALU write of its data registers: 2 wait states on gfx940+),
// and the compiler's code had those.  The question a reproducer can answer: with the LDS queue of the CU busy, how many
// wait states does the hardware ACTUALLY need?
//
// The program: every wave owns 1 KiB of LDS.  Per iteration a lane makes four fresh words, stores them with ONE
// ds_write_b128 and -- inside the same assembly statement, so that nothing of the compiler's comes between -- overwrites
// the four data registers with v_not after GAP wait states (0, 1, 2, 3, 4, 6, 8, 12, 16); then it waits for the store,
// reads the 16 bytes back and compares them with a copy of the words kept elsewhere.  Around it the other waves of the
// CU do the same (the LDS is busy with 16-byte stores and reads), at 1 ... 8 waves per SIMD.  Controls: the same with an
// 8-byte store (ds_write_b64: no hazard documented) and a 4-byte one; and the 16-byte store QUEUED behind eight 16-byte reads
// of the same wave that nobody has waited for (the store's data registers overwritten while it still stands in the queue).
//
//   hipcc --offload-arch=gfx950 -O3 lds_store_war.hip -o /tmp/lds_store_war && /tmp/lds_store_war
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define NOPS_0 ""
#define NOPS_1 "s_nop 0\n\t"
#define NOPS_2 "s_nop 1\n\t"
#define NOPS_3 "s_nop 2\n\t"
#define NOPS_4 "s_nop 3\n\t"
#define NOPS_6 "s_nop 5\n\t"
#define NOPS_8 "s_nop 7\n\t"
#define NOPS_12 "s_nop 11\n\t"
#define NOPS_16 "s_nop 15\n\t"

// Store 16 bytes out of v[100:103], then GAP wait states, then overwrite the four data registers (fixed registers: an
// assembly operand of 128 bits cannot be addressed dword by dword).  TAIL_LAST: only the LAST data register is overwritten
// (the LDS takes the dwords in order: the one it reads last is the one a following instruction can still beat).
#define STORE128(GAPSTR, TAIL)                                                                                      \
    asm volatile("v_mov_b32 v100, %1\n\tv_mov_b32 v101, %2\n\tv_mov_b32 v102, %3\n\tv_mov_b32 v103, %4\n\ts_nop 4\n\t"  \
                 "ds_write_b128 %0, v[100:103]\n\t" GAPSTR TAIL                                                     \
                 :: "v"(addr), "v"(w0), "v"(w1), "v"(w2), "v"(w3) : "v100", "v101", "v102", "v103", "memory")
// the same store QUEUED behind eight 16-byte reads of the wave's own (issued, not waited for): if the data registers are
// read when the store reaches the LDS rather than when it issues, the window is as long as the queue
#define STORE128_QUEUED(GAPSTR, TAIL)                                                                               \
    asm volatile("v_mov_b32 v100, %1\n\tv_mov_b32 v101, %2\n\tv_mov_b32 v102, %3\n\tv_mov_b32 v103, %4\n\ts_nop 4\n\t"  \
                 "ds_read_b128 v[108:111], %5\n\tds_read_b128 v[112:115], %5 offset:16\n\tds_read_b128 v[116:119], %5 offset:32\n\t" \
                 "ds_read_b128 v[120:123], %5 offset:48\n\tds_read_b128 v[108:111], %5 offset:64\n\tds_read_b128 v[112:115], %5 offset:80\n\t" \
                 "ds_read_b128 v[116:119], %5 offset:96\n\tds_read_b128 v[120:123], %5 offset:112\n\t"               \
                 "ds_write_b128 %0, v[100:103]\n\t" GAPSTR TAIL "\n\ts_waitcnt lgkmcnt(0)"                           \
                 :: "v"(addr), "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(other)                                          \
                 : "v100", "v101", "v102", "v103", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", \
                   "v118", "v119", "v120", "v121", "v122", "v123", "memory")
#define TAIL_ALL "v_not_b32 v100, v100\n\tv_not_b32 v101, v101\n\tv_not_b32 v102, v102\n\tv_not_b32 v103, v103"
#define TAIL_LAST "v_not_b32 v103, v103"

template <int GAP>
__device__ __forceinline__ void store128_queued(uint32_t addr, uint32_t other, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3)
{
#define CASE(G, STR) if (GAP == G) STORE128_QUEUED(STR, TAIL_ALL);
    CASE(0, NOPS_0) CASE(1, NOPS_1) CASE(2, NOPS_2) CASE(4, NOPS_4) CASE(8, NOPS_8) CASE(16, NOPS_16)
#undef CASE
}

template <int GAP, bool LAST_ONLY>
__device__ __forceinline__ void store128(uint32_t addr, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3)
{
#define CASE(G, STR) if (GAP == G) { if (LAST_ONLY) STORE128(STR, TAIL_LAST); else STORE128(STR, TAIL_ALL); }
    CASE(0, NOPS_0) CASE(1, NOPS_1) CASE(2, NOPS_2) CASE(3, NOPS_3) CASE(4, NOPS_4) CASE(6, NOPS_6) CASE(8, NOPS_8)
    CASE(12, NOPS_12) CASE(16, NOPS_16)
#undef CASE
}

// KIND 0: ds_write_b128, all four data registers overwritten after GAP states; 3: only the LAST one (the dword the LDS takes
// last); 1: ds_write_b64 (control); 2: ds_write_b32
template <int KIND, int GAP>
__global__ void __launch_bounds__(1024) war_kernel(unsigned long long *bad, int iters, uint32_t seed)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    const uint32_t base = (uint32_t)(uintptr_t)(lds_u8 *)lds + wave * 2048u;           // 2 KiB per wave: two slots of 1 KiB
    uint32_t s = mix(gid * 2654435761u + seed), sink = 0;
    unsigned long long n_bad = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t w0 = s = mix(s + 1u), w1 = s = mix(s + 2u), w2 = s = mix(s + 3u), w3 = s = mix(s + 4u);
        const uint32_t addr = base + (uint32_t)(it & 1) * 1024u + lane * 16u;
        if (KIND == 0) store128<GAP, false>(addr, w0, w1, w2, w3);
        if (KIND == 3) store128<GAP, true>(addr, w0, w1, w2, w3);
        if (KIND == 4) store128_queued<GAP>(addr, base + (uint32_t)((it + 1) & 1) * 1024u + (lane & 56u) * 16u, w0, w1, w2, w3);
        if (KIND == 1) {
            if (GAP == 0)
                asm volatile("v_mov_b32 v104, %1\n\tv_mov_b32 v105, %2\n\ts_nop 4\n\tds_write_b64 %0, v[104:105]\n\t"
                             "v_not_b32 v104, v104\n\tv_not_b32 v105, v105" :: "v"(addr), "v"(w0), "v"(w1) : "v104", "v105", "memory");
            else
                asm volatile("v_mov_b32 v104, %1\n\tv_mov_b32 v105, %2\n\ts_nop 4\n\tds_write_b64 %0, v[104:105]\n\ts_nop 1\n\t"
                             "v_not_b32 v104, v104\n\tv_not_b32 v105, v105" :: "v"(addr), "v"(w0), "v"(w1) : "v104", "v105", "memory");
        }
        if (KIND == 2) {
            uint32_t v = w0;
            if (GAP == 0) asm volatile("ds_write_b32 %1, %0\n\tv_not_b32 %0, %0" : "+v"(v) : "v"(addr) : "memory");
            else asm volatile("ds_write_b32 %1, %0\n\ts_nop 1\n\tv_not_b32 %0, %0" : "+v"(v) : "v"(addr) : "memory");
            sink ^= v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        u32x4 r;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
        bool ok = r[0] == w0;
        if (KIND <= 1) ok = ok && r[1] == w1;
        if (KIND == 0 || KIND == 3 || KIND == 4) ok = ok && r[1] == w1 && r[2] == w2 && r[3] == w3;
        n_bad += ok ? 0 : 1;
    }
    if (n_bad) atomicAdd(bad, n_bad);
    if (sink == 0x12345u) atomicAdd(bad + 1, 1ull);                                    // (keeps the overwritten registers alive)
}

template <int KIND, int GAP>
static void run(const char *what, int waves_per_simd, int iters, unsigned long long *dbad)
{
    const int threads = 64 * 4 * waves_per_simd > 1024 ? 1024 : 64 * 4 * waves_per_simd;
    const int wg_per_cu = (64 * 4 * waves_per_simd + threads - 1) / threads;
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int grid = p.multiProcessorCount * wg_per_cu;
    const size_t lds = (size_t)(threads / 64) * 2048;
    (void)hipMemset(dbad, 0, 16);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((war_kernel<KIND, GAP>), dim3(grid), dim3(threads), lds, 0, dbad, iters, 12345u + GAP);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    unsigned long long h[2] = {0, 0};
    (void)hipMemcpy(h, dbad, 16, hipMemcpyDeviceToHost);
    const double stores = (double)grid * threads * iters;
    printf("  %-14s gap %2d states, %d waves/SIMD: %.3g lane-stores in %.1f ms; read back WRONG: %llu (%.3g per million)\n",
           what, GAP, waves_per_simd, stores, ms, h[0], stores > 0 ? 1e6 * (double)h[0] / stores : 0.0);
}

int main()
{
    unsigned long long *dbad = nullptr;
    if (hipMalloc(&dbad, 16) != hipSuccess) { printf("no device\n"); return 1; }
    const int iters = 20000;
    printf("a 16-byte LDS store followed, GAP wait states later, by a vector instruction that overwrites its first data register\n");
    for (int w : {1, 2, 4, 8}) {
        run<0, 0>("ds_write_b128", w, iters, dbad);
        run<0, 1>("ds_write_b128", w, iters, dbad);
        run<0, 2>("ds_write_b128", w, iters, dbad);
        run<0, 3>("ds_write_b128", w, iters, dbad);
        run<0, 4>("ds_write_b128", w, iters, dbad);
        run<0, 6>("ds_write_b128", w, iters, dbad);
        run<0, 8>("ds_write_b128", w, iters, dbad);
        run<0, 12>("ds_write_b128", w, iters, dbad);
        run<0, 16>("ds_write_b128", w, iters, dbad);
        run<4, 0>("b128 queued", w, iters, dbad);
        run<4, 1>("b128 queued", w, iters, dbad);
        run<4, 2>("b128 queued", w, iters, dbad);
        run<4, 4>("b128 queued", w, iters, dbad);
        run<4, 8>("b128 queued", w, iters, dbad);
        run<4, 16>("b128 queued", w, iters, dbad);
        run<3, 0>("b128, last dword", w, iters, dbad);
        run<3, 2>("b128, last dword", w, iters, dbad);
        run<3, 4>("b128, last dword", w, iters, dbad);
        run<3, 8>("b128, last dword", w, iters, dbad);
        run<1, 0>("ds_write_b64", w, iters, dbad);
        run<1, 2>("ds_write_b64", w, iters, dbad);
        run<2, 0>("ds_write_b32", w, iters, dbad);
    }
    (void)hipFree(dbad);
    return 0;
}
