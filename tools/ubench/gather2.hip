// Row-gather roofline of MI355X, second generation.
//
// The first ubench (gather.hip) spent an LCG and a runtime `% nrows` per load and printed a broken
// B/clk/CU column; its 9.6 TB/s "L2 row ceiling" contradicted the guide's 34.5 TB/s L2 figure.
// Here the address generation is a handful of full-rate instructions (or a precomputed table), and
// the sweep covers what a gather kernel can actually choose:
//   * row length (lanes per row 16 / 32 / 64 = 256 B / 512 B / 1 KiB contiguous per row),
//   * loads in flight per lane (1..16), waves per CU (4..32),
//   * where the rows live: 16 KiB (L1), 1 MiB per XCD (L2), 16 MiB per XCD (MALL), 2 GiB (HBM),
//   * how they travel: global_load_dwordx4 to VGPRs, the same with nt / sc1 policy, LDS-DMA,
//   * coalesced streaming of an L2-resident region (what a tile-streaming kernel would do),
//   * random 256-B rows out of LDS (ds_read_b128),
//   * vector-ALU issue rates of the instructions the gather kernels are made of.
// Reported against the guide's figures: L2 34.5 TB/s, L1 64 B/clk/CU, LDS 256 B/clk/CU, clock 2.4 GHz
// nominal (the effective clock of each run is measured with s_memtime).
// Build: hipcc --offload-arch=gfx950 -O3 gather2.hip -o gather2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void keep(const u32x4 &v) { asm volatile("" :: "v"(v)); }

// row index of load number x: two multiplicative rounds (5 vector instructions per load); the TABLE
// variant takes the rows from a host-side mt19937 instead, to show the hash is random enough
__device__ __forceinline__ unsigned hrow(unsigned x, int log2rows)
{
    unsigned h = x * 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    return h >> (32 - log2rows);
}

enum { POL_PLAIN = 0, POL_NT = 1, POL_SC1 = 2 };

template <int POL>
__device__ __forceinline__ u32x4 ld16(const char *p)
{
    u32x4 v;
    if (POL == POL_PLAIN) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p));
    if (POL == POL_NT)    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p));
    if (POL == POL_SC1)   asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p));
    return v;
}

// ------------------------------------------------------------------ random rows -> VGPRs
// LPR lanes share one row (LPR*16 contiguous bytes); U loads in flight per lane; TABLE: row offsets
// come from a precomputed table (one extra coalesced dword load per gather load) instead of the hash.
template <int LPR, int U, int POL, bool TABLE>
__global__ void __launch_bounds__(256) gather_rows(const char *base, size_t region_stride, int n_regions,
                                                  int log2rows, int stride, int iters,
                                                  const unsigned *table, unsigned long long *clk)
{
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    const unsigned grp = tid / LPR, lig = tid % LPR;
    const unsigned ngrp = gridDim.x * 256 / LPR;
    const char *reg = base + (size_t)(blockIdx.x % n_regions) * region_stride + lig * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        u32x4 v[U];
        unsigned row[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned x = (unsigned)(it * U + u) * ngrp + grp;
            row[u] = TABLE ? table[(size_t)(it * U + u) * ngrp + grp] : hrow(x, log2rows);     // one entry per row: the 16 lanes read the same word
        }
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld16<POL>(reg + (size_t)row[u] * (unsigned)stride);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < U; ++u) keep(v[u]);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

// ------------------------------------------------------------------ random rows -> LDS (DMA)
// every wave owns U KiB of LDS; a wave-instruction lands 64 x 16 B = 4 rows of 256 B (LPR = 16)
template <int LPR, int U>
__global__ void __launch_bounds__(256) gather_rows_lds(const char *base, size_t region_stride, int n_regions,
                                                      int log2rows, int stride, int iters,
                                                      unsigned long long *clk)
{
    __shared__ __attribute__((aligned(1024))) char lds[4 * U * 1024];
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    const unsigned grp = tid / LPR, lig = tid % LPR;
    const unsigned ngrp = gridDim.x * 256 / LPR;
    const char *reg = base + (size_t)(blockIdx.x % n_regions) * region_stride + lig * 16;
    char *mine = lds + (threadIdx.x / 64) * (U * 1024);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned x = (unsigned)(it * U + u) * ngrp + grp;
            const char *src = reg + (size_t)hrow(x, log2rows) * (unsigned)stride;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(mine + u * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
    if (iters < 0) clk[1] = lds[threadIdx.x];
}

// ------------------------------------------------------------------ coalesced streaming of a region
// a wave reads consecutive KiB of its XCD's region round and round (L2-resident after the first pass)
template <int U, bool TO_LDS>
__global__ void __launch_bounds__(256) stream_region(const char *base, size_t region_stride, int n_regions,
                                                    unsigned region_bytes, int iters, unsigned long long *clk)
{
    __shared__ __attribute__((aligned(1024))) char lds[4 * U * 1024];
    const unsigned wave = (blockIdx.x / n_regions) * 4 + threadIdx.x / 64;      // wave index inside its region
    const unsigned nwave = (gridDim.x / n_regions) * 4;
    const char *reg = base + (size_t)(blockIdx.x % n_regions) * region_stride + (threadIdx.x % 64) * 16;
    char *mine = lds + (threadIdx.x / 64) * (U * 1024);
    unsigned off = (wave * 1024u) % region_bytes;
    const unsigned step = (nwave * 1024u) % region_bytes;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (TO_LDS)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reg + off),
                                                 (__attribute__((address_space(3))) void *)(mine + u * 1024), 16, 0, 0);
            else
                v[u] = ld16<POL_PLAIN>(reg + off);
            off += step;
            if (off >= region_bytes) off -= region_bytes;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!TO_LDS) {
#pragma unroll
            for (int u = 0; u < U; ++u) keep(v[u]);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
    if (iters < 0) clk[1] = lds[threadIdx.x];
}

// ------------------------------------------------------------------ random 256-B rows out of LDS
template <int U, bool TR>
__global__ void __launch_bounds__(1024) lds_rows(int log2rows, int iters, unsigned long long *clk, float *out)
{
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    for (int i = threadIdx.x; i < (256 << log2rows) / 4; i += blockDim.x) ((unsigned *)lds)[i] = i;
    __syncthreads();
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned grp = tid / 16, lig = tid % 16, ngrp = gridDim.x * blockDim.x / 16;
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned x = (unsigned)(it * U + u) * ngrp + grp;
            const unsigned a = hrow(x, log2rows) * 256u + lig * 16u;
            if (TR) {   // two transposing 8-byte reads: the MFMA-operand form of the same 16 bytes
                typedef short s16x4 __attribute__((ext_vector_type(4)));
                s16x4 p, q;
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(p) : "v"(a & ~8u));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8" : "=v"(q) : "v"(a & ~8u));
                v[u][0] = __builtin_bit_cast(unsigned long long, p) & 0xffffffffu;
                v[u][1] = __builtin_bit_cast(unsigned long long, p) >> 32;
                v[u][2] = __builtin_bit_cast(unsigned long long, q) & 0xffffffffu;
                v[u][3] = __builtin_bit_cast(unsigned long long, q) >> 32;
            } else {
                asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(a));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u][0];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
    if (acc == 0x12345u) out[0] = 1.f;
}

// ------------------------------------------------------------------ vector-ALU issue rates
// KIND 0: v_fma_f32   1: v_pk_fma_f32   2: v_dot2c_f32_bf16   3: shl+and unpack pair   4: v_mul_lo_u32
//      5: v_cvt_pk_bf16_f32   6: v_perm_b32
template <int KIND>
__global__ void __launch_bounds__(256) valu_rate(int iters, unsigned long long *clk, float *out)
{
    float a[8];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8];
    unsigned w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = f2{a[i], a[i] + 1}; w[i] = threadIdx.x * 2654435761u + i; }
    const float m = 1.0001f, c = 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
                if (KIND == 2) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a[i]) : "v"(w[i]), "v"(w[(i + 1) & 7]));
                if (KIND == 3) { unsigned lo, hi;
                                 asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(lo) : "v"(w[i]));
                                 asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(hi) : "v"(w[i]));
                                 asm volatile("" :: "v"(lo), "v"(hi)); }
                if (KIND == 4) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(w[i]) : "v"(w[(i + 1) & 7]));
                if (KIND == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
                if (KIND == 6) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(w[i]) : "v"(w[(i + 1) & 7]), "v"(w[(i + 2) & 7]), "v"(0x07060302u));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1] + (float)w[i];
    if (s == 1.2345f) out[0] = s;
}

// ------------------------------------------------------------------ host
static hipEvent_t ev0, ev1;
static unsigned long long *d_clk;
static std::vector<unsigned long long> h_clk;

struct Res { double ms, mclk; };     // wall time of the launch, mean s_memtime ticks per workgroup

template <typename F> static Res timed(int blocks, F launch)
{
    float ms = 0, best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(ev0)); launch(); CK(hipEventRecord(ev1)); CK(hipEventSynchronize(ev1));
        CK(hipGetLastError());
        CK(hipEventElapsedTime(&ms, ev0, ev1));
        if (ms < best) best = ms;
    }
    h_clk.resize(blocks);
    CK(hipMemcpy(h_clk.data(), d_clk, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double s = 0; for (int i = 0; i < blocks; ++i) s += (double)h_clk[i];
    return Res{best, s / blocks};
}

static void report(const char *name, double bytes, Res r)
{
    // s_memtime runs at a fixed 100 MHz on this part (checked below against the FMA chain), so the
    // per-clock figure uses the nominal 2.4 GHz shader clock and the fma-calibrated one side by side
    const double tbs = bytes / (r.ms * 1e-3) / 1e12;
    printf("%-86s %8.3f ms %7.2f TB/s %6.1f B/clk/CU@2.4\n", name, r.ms, tbs, bytes / (r.ms * 1e-3) / 256 / 2.4e9);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    const bool table_only = argc > 1 && !strcmp(argv[1], "table");     // only the hash-vs-table comparison
    const size_t bytes = 2ull << 30;
    char *buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 1, bytes));
    float *out; CK(hipMalloc(&out, 64));
    CK(hipMalloc(&d_clk, 65536 * sizeof(unsigned long long)));
    CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
    char name[256];

    // ---- 0. vector-ALU issue rates (also calibrates the clock: v_fma_f32 is 1 per 2 clk per SIMD... or not)
    printf("== VALU issue: wave-instructions per SIMD per clock @2.4 GHz nominal (256 CUs x 4 SIMDs)\n");
    if (!table_only) {
        const char *kn[] = {"v_fma_f32", "v_pk_fma_f32", "v_dot2c_f32_bf16", "v_lshlrev+v_and (bf16 unpack pair, 2 instr)",
                            "v_mul_lo_u32", "v_cvt_pk_bf16_f32", "v_perm_b32"};
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = 256 * wps, iters = 4096;
            for (int kind = 0; kind < 7; ++kind) {
                Res r = timed(blocks, [&] {
                    switch (kind) {
                    case 0: hipLaunchKernelGGL(valu_rate<0>, dim3(blocks), dim3(256), 0, 0, iters, d_clk, out); break;
                    case 1: hipLaunchKernelGGL(valu_rate<1>, dim3(blocks), dim3(256), 0, 0, iters, d_clk, out); break;
                    case 2: hipLaunchKernelGGL(valu_rate<2>, dim3(blocks), dim3(256), 0, 0, iters, d_clk, out); break;
                    case 3: hipLaunchKernelGGL(valu_rate<3>, dim3(blocks), dim3(256), 0, 0, iters, d_clk, out); break;
                    case 4: hipLaunchKernelGGL(valu_rate<4>, dim3(blocks), dim3(256), 0, 0, iters, d_clk, out); break;
                    case 5: hipLaunchKernelGGL(valu_rate<5>, dim3(blocks), dim3(256), 0, 0, iters, d_clk, out); break;
                    case 6: hipLaunchKernelGGL(valu_rate<6>, dim3(blocks), dim3(256), 0, 0, iters, d_clk, out); break;
                    } });
                const double instr = (double)iters * 32 * (kind == 3 ? 2 : 1) * wps;       // per SIMD
                printf("  %d wave/SIMD %-46s %7.3f ms  %.3f instr/clk/SIMD (%.2f clk each)  memtime ticks %.0f\n", wps, kn[kind], r.ms,
                       instr / (r.ms * 1e-3 * 2.4e9), (r.ms * 1e-3 * 2.4e9) / instr, r.mclk);
            }
            if (quick) break;
        }
    }

    // ---- 1. random rows -> VGPRs
    printf("== random rows, global_load_dwordx4 -> VGPRs. guide: L1 64 B/clk/CU = 39.3 TB/s, L2 34.5 TB/s, HBM 8 (6.3 copy)\n");
    struct Where { const char *name; int log2rows; int stride; size_t region_stride; int n_regions; } wheres[] = {
        {"L1: 64 rows (16 KiB) per XCD region", 6, 256, 4u << 20, 8},
        {"L2: 4096 rows dense (1 MiB per XCD)", 12, 256, 4u << 20, 8},
        {"L2: 4096 rows stride 2048 (value layout, H=8 D=128)", 12, 2048, 256, 8},
        {"MALL: 65536 rows dense (16 MiB per XCD, 128 MiB total)", 16, 256, 16u << 20, 8},
        {"HBM: 2^23 rows dense (2 GiB, one region)", 23, 256, 0, 1},
    };
#define RUN_ROWS(LPR, U, POL, TABLE, w, wpc)                                                                         \
    do {                                                                                                             \
        const int blocks = 256 * (wpc) / 4;                                                                          \
        const int iters = (quick ? 64 : 256) * 8 / U;                                                                \
        Res r = timed(blocks, [&] { hipLaunchKernelGGL((gather_rows<LPR, U, POL, TABLE>), dim3(blocks), dim3(256), 0, 0, \
                                       buf, w.region_stride, w.n_regions, w.log2rows, w.stride * (LPR / 16), iters,   \
                                       (const unsigned *)tab, d_clk); });                                            \
        snprintf(name, sizeof name, "%-56s row %4dB U=%2d %2dw/CU %s%s", w.name, LPR * 16, U, wpc,                   \
                 POL == POL_NT ? "nt " : POL == POL_SC1 ? "sc1 " : "", TABLE ? "table" : "");                        \
        report(name, (double)blocks * 256 * iters * U * 16, r);                                                      \
    } while (0)
    unsigned *tab = nullptr;
    if (table_only) { RUN_ROWS(16, 8, POL_PLAIN, false, wheres[1], 32); RUN_ROWS(16, 8, POL_PLAIN, false, wheres[2], 32); }
    for (auto &w : wheres) {
        if (table_only) break;
        // loads in flight at 32 waves/CU
        RUN_ROWS(16, 1, POL_PLAIN, false, w, 32);
        RUN_ROWS(16, 2, POL_PLAIN, false, w, 32);
        RUN_ROWS(16, 4, POL_PLAIN, false, w, 32);
        RUN_ROWS(16, 8, POL_PLAIN, false, w, 32);
        RUN_ROWS(16, 16, POL_PLAIN, false, w, 32);
        // waves per CU at 8 in flight
        RUN_ROWS(16, 8, POL_PLAIN, false, w, 4);
        RUN_ROWS(16, 8, POL_PLAIN, false, w, 8);
        RUN_ROWS(16, 8, POL_PLAIN, false, w, 16);
        // longer rows
        if (w.log2rows >= 8) {
            Where w2 = w; w2.log2rows -= 1; RUN_ROWS(32, 8, POL_PLAIN, false, w2, 32);
            Where w4 = w; w4.log2rows -= 2; RUN_ROWS(64, 8, POL_PLAIN, false, w4, 32);
        }
        // cache policy
        RUN_ROWS(16, 8, POL_NT, false, w, 32);
        RUN_ROWS(16, 8, POL_SC1, false, w, 32);
    }
    {   // precomputed table instead of the hash (same rows), L2 case
        const Where &w = wheres[1];
        const int blocks = 256 * 8, iters = (quick ? 64 : 256);
        const size_t n = (size_t)iters * 8 * blocks * 256 / 16;      // one entry per gathered row
        std::vector<unsigned> h(n);
        std::mt19937 rng(1234);
        for (size_t i = 0; i < n; ++i) h[i] = rng() >> (32 - w.log2rows);
        CK(hipMalloc(&tab, n * 4)); CK(hipMemcpy(tab, h.data(), n * 4, hipMemcpyHostToDevice));
        RUN_ROWS(16, 8, POL_PLAIN, true, w, 32);
        RUN_ROWS(16, 8, POL_PLAIN, true, wheres[2], 32);
        CK(hipFree(tab)); tab = nullptr;
    }
    if (table_only) { printf("done\n"); return 0; }

    // ---- 2. random rows -> LDS by DMA
    printf("== random 256-B rows, global_load_lds_dwordx4 (LDS-DMA)\n");
#define RUN_DMA(U, w, wpc)                                                                                           \
    do {                                                                                                             \
        const int blocks = 256 * (wpc) / 4;                                                                          \
        const int iters = (quick ? 64 : 256) * 8 / U;                                                                \
        Res r = timed(blocks, [&] { hipLaunchKernelGGL((gather_rows_lds<16, U>), dim3(blocks), dim3(256), 0, 0,      \
                                       buf, w.region_stride, w.n_regions, w.log2rows, w.stride, iters, d_clk); });   \
        snprintf(name, sizeof name, "%-56s row  256B U=%2d %2dw/CU lds-dma", w.name, U, wpc);                        \
        report(name, (double)blocks * 256 * iters * U * 16, r);                                                      \
    } while (0)
    for (int wi : {0, 1, 3}) {
        const Where &w = wheres[wi];
        RUN_DMA(2, w, 32); RUN_DMA(4, w, 32); RUN_DMA(8, w, 32); RUN_DMA(8, w, 16); RUN_DMA(16, w, 16); RUN_DMA(8, w, 8);
    }

    // ---- 3. coalesced streaming of an L2-resident region
    printf("== coalesced streaming of a 1.375 MiB region per XCD (L2-resident), 1 KiB per wave-instruction\n");
#define RUN_STREAM(U, LDS, wpc, rb)                                                                                  \
    do {                                                                                                             \
        const int blocks = 256 * (wpc) / 4;                                                                          \
        const int iters = (quick ? 64 : 256) * 8 / U;                                                                \
        Res r = timed(blocks, [&] { hipLaunchKernelGGL((stream_region<U, LDS>), dim3(blocks), dim3(256), 0, 0,       \
                                       buf, (size_t)(4u << 20), 8, (unsigned)(rb), iters, d_clk); });                \
        snprintf(name, sizeof name, "stream %7d B region/XCD U=%2d %2dw/CU %s", (int)(rb), U, wpc, LDS ? "lds-dma" : "vgpr"); \
        report(name, (double)blocks * 256 * iters * U * 16, r);                                                      \
    } while (0)
    for (unsigned rb : {1441792u, 65536u}) {
        RUN_STREAM(4, false, 32, rb); RUN_STREAM(8, false, 32, rb); RUN_STREAM(8, false, 16, rb); RUN_STREAM(16, false, 8, rb);
        RUN_STREAM(4, true, 32, rb); RUN_STREAM(8, true, 32, rb); RUN_STREAM(8, true, 16, rb); RUN_STREAM(16, true, 8, rb);
        RUN_STREAM(16, true, 4, rb);
    }

    // ---- 4. random 256-B rows out of LDS
    printf("== random 256-B rows out of a 64 KiB LDS region (guide: ds_read_b128 256 B/clk/CU = 157 TB/s)\n");
#define RUN_LDS(U, TR, wpc)                                                                                          \
    do {                                                                                                             \
        const int blocks = 256, iters = quick ? 512 : 2048;                                                          \
        Res r = timed(blocks, [&] { hipLaunchKernelGGL((lds_rows<U, TR>), dim3(blocks), dim3((wpc) * 64), 65536, 0,  \
                                       8, iters * 8 / U, d_clk, out); });                                            \
        snprintf(name, sizeof name, "lds rows U=%2d %2dw/CU %s", U, wpc, TR ? "2x ds_read_b64_tr_b16" : "ds_read_b128"); \
        report(name, (double)blocks * (wpc) * 64 * (iters * 8 / U) * U * 16, r);                                     \
    } while (0)
    RUN_LDS(4, false, 16); RUN_LDS(8, false, 16); RUN_LDS(8, false, 8); RUN_LDS(8, false, 4); RUN_LDS(16, false, 4);
    RUN_LDS(8, true, 16); RUN_LDS(8, true, 8);
    printf("done\n");
    return 0;
}
