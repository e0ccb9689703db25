// tile_lds_mix.hip -- what does the PER-STEP INSTRUCTION MIX of the grad_value tile reduce cost on a CU, with everything
// that is not that mix taken away?   ("... or the ubench that says why not".)
//
// msda_bwd_tile_reduce (csrc/msda_bwd_tile.hip) walks cell-sorted records 16 at a time.  A step of one wave, heads of 128
// channels, 16-bit storage:
//     rows in      4 x buffer_load_dwordx4 ... lds   16 grad_out rows of 256 B, global (L2) -> LDS, no registers      4 KB
//     row index    4 x ds_read_b32                   the records' query indices for those requests                      1 KB
//     weights      ds_write_b128 (zero the 32 x 16 tile), 2 x ds_read_b32 (record words), ~20 vector instructions,
//                  2 x ds_write_b16 (hi / lo of one weight), ds_read_b128 (this lane's 16 bytes of the A operand)       ~3 KB
//     rows out     8 x ds_read_b64_tr_b16            the rows as B operands (transposing read)                          4 KB
//     products     4 x v_mfma_f32_32x32x16_bf16
// at 14 resident waves per CU (64-lane workgroups, 128 registers, 10 KB of LDS each).  tools/tile_prof.py measures 1.6 k
// clocks per step and wave in the kernel = one step per ~114 clocks per CU.
//
// This program issues exactly that mix -- same builtins, same LDS footprint per workgroup, same residency -- from a
// 1 MB slab that never leaves the L2 (one (b, h) slice of grad_out), with every list 64 steps long and no prologue,
// epilogue, partial tile, descriptor or queue: the ceiling of the FORMULATION on this part.  Variants knock one
// ingredient out at a time to show what the clocks are spent on.
//
//   hipcc --offload-arch=gfx950 -O3 tile_lds_mix.hip -o /tmp/tile_lds_mix && /tmp/tile_lds_mix
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int kRows = 4096;             // rows of the slab (queries of one slice)
constexpr int kRB = 256;                // bytes of a row (D = 128, 16-bit)
constexpr int kSlot = 16 * kRB;         // a step's rows
constexpr int kLdsOf(int stages) { return stages * kSlot + 1024 + 1024; }      // row slots, records, weight tile: the kernel's 10 KB at two slots

enum : unsigned { DMA = 1, IDX = 2, WEIGHTS = 4, TR = 8, MFMA = 16, ALL = 31, VIA_REGS = 32 };       // VIA_REGS: the rows through registers + ds_write_b128 instead of the DMA

#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n) : "memory")

template <unsigned WHAT, int STAGES = 2>
__global__ void __launch_bounds__(64, 4) mix_kernel(const uint16_t *__restrict__ slab, float *__restrict__ sink, int steps, uint32_t seed)
{
    __shared__ __attribute__((aligned(1024))) unsigned char lds[kLdsOf(STAGES)];
    unsigned char *rows = lds, *recs = lds + STAGES * kSlot, *atile = recs + 1024;
    const int lane = threadIdx.x;
    // records: 128 "query | weight" words and 128 "x | y" words, as a sorted list would hold them (random rows)
    uint32_t s = (blockIdx.x * 64u + lane) * 2654435761u + seed;
    for (int i = lane; i < 256; i += 64) {
        s = s * 1664525u + 1013904223u;
        reinterpret_cast<uint32_t *>(recs)[i] = i < 128 ? ((s >> 8) % kRows) | (0x3f00u << 16) : (s & 0x3fff3fffu) | 0x3c003c00u;
    }
    for (int i = lane; i < STAGES * kSlot / 16; i += 64) reinterpret_cast<uint4 *>(rows)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    const uint64_t base = (uint64_t)slab;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)base)),
        (short)0, kRows * kRB, 0x00020000);
    const int rsel = lane >> 4, g4 = lane >> 4, j16 = lane & 15;
    int troff[4];
    for (int nb = 0; nb < 4; ++nb) {
        const int krow = 8 * (g4 >> 1) + (j16 >> 2);
        const int cb = nb * 64 + (g4 & 1) * 32 + (j16 & 3) * 8;
        troff[nb] = krow * kRB + (((cb >> 4) ^ (4 * ((krow / 1) % 4))) << 4) + (cb & 15);
    }
    const int a_rd = (lane & 31) * 32 + (lane >> 5) * 16;
    const int wr = lane >> 2, wc = lane & 3;
    f32x16 acc[4];
    for (int nb = 0; nb < 4; ++nb)
        for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;

    uint4 held[4];                                      // VIA_REGS: the step's rows on their way (16 B per lane and request)
    auto issue_rows = [&](int k) {
        if (!(WHAT & DMA)) return;
        if (WHAT & VIA_REGS) {
            const uint32_t *rq = reinterpret_cast<const uint32_t *>(recs) + (k & 7) * 16;
            uint32_t q[4];
            for (int u = 0; u < 4; ++u) q[u] = (WHAT & IDX) ? (rq[u * 4 + rsel] & 0xffffu) : (uint32_t)((k * 16 + u * 4 + rsel) * 37) % kRows;
            for (int u = 0; u < 4; ++u) {
                const int rr = u * 4 + rsel;
                const uint32_t chunk = (uint32_t)((lane & 15) ^ (4 * (rr % 4)));
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(q[u] * kRB + chunk * 16u), 0, 0);
                held[u] = make_uint4(v[0], v[1], v[2], v[3]);
            }
            return;
        }
        const uint32_t *rq = reinterpret_cast<const uint32_t *>(recs) + (k & 7) * 16;
        uint32_t q[4];
        for (int u = 0; u < 4; ++u) q[u] = (WHAT & IDX) ? (rq[u * 4 + rsel] & 0xffffu) : (uint32_t)((k * 16 + u * 4 + rsel) * 37) % kRows;
        for (int u = 0; u < 4; ++u) {
            const int rr = u * 4 + rsel;
            const uint32_t chunk = (uint32_t)((lane & 15) ^ (4 * (rr % 4)));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t *)(rows + (k % STAGES) * kSlot + u * 1024), 16,
                                                     (int)(q[u] * kRB + chunk * 16u), 0, 0, 0);
        }
    };
    auto multiply = [&](int k) {
        s16x8 A;
        if (WHAT & WEIGHTS) {
            reinterpret_cast<uint4 *>(atile)[lane] = make_uint4(0u, 0u, 0u, 0u);
            const uint32_t *rb = reinterpret_cast<const uint32_t *>(recs) + (k & 7) * 16 + wr;
            const uint32_t w0 = rb[0], w1 = rb[128];
            const float av = __uint_as_float(w0 & 0xffff0000u);
            const float lx = __uint_as_float(w1 << 16), ly = __uint_as_float(w1 & 0xffff0000u);
            const float y = ly * 16.f - 0.5f, x = lx * 16.f - 0.5f;
            const float yf = floorf(y), xf = floorf(x);
            const float fy = y - yf, fx = x - xf;
            const int iy = ((int)yf + (wc >> 1)) & 3, ix = ((int)xf + (wc & 1)) & 3;
            const float wgt = ((wc >> 1) ? fy : 1.f - fy) * ((wc & 1) ? fx : 1.f - fx) * av;
            const uint32_t h = __float_as_uint(wgt) & 0xffff0000u;
            const uint16_t hi = (uint16_t)(h >> 16), lo = __builtin_bit_cast(uint16_t, (__bf16)(wgt - __uint_as_float(h)));
            const int m = iy * 4 + ix;
            reinterpret_cast<uint16_t *>(atile)[m * 16 + wr] = hi;
            reinterpret_cast<uint16_t *>(atile)[(16 + m) * 16 + wr] = lo;
            A = *reinterpret_cast<const s16x8 *>(atile + a_rd);
        } else {
            for (int i = 0; i < 8; ++i) A[i] = (short)(0x3c00 + lane + k);
        }
        const unsigned char *slot = rows + (k % STAGES) * kSlot;
        for (int nb = 0; nb < 4; ++nb) {
            s16x8 B;
            if (WHAT & TR) {
                for (int t = 0; t < 2; ++t) {
                    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4 *)(slot + troff[nb] + t * 4 * kRB));
                    B[4 * t] = v[0]; B[4 * t + 1] = v[1]; B[4 * t + 2] = v[2]; B[4 * t + 3] = v[3];
                }
            } else {
                for (int i = 0; i < 8; ++i) B[i] = (short)(0x3c00 + nb + i);
            }
            if (WHAT & MFMA)
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), acc[nb], 0, 0, 0);
            else
                acc[nb][0] += (float)A[0] + (float)B[nb];
        }
    };
    for (int k = 0; k < STAGES - 1; ++k) issue_rows(k);
    for (int k = 0; k < steps; ++k) {
        if (STAGES > 2 && (WHAT & DMA) && !(WHAT & VIA_REGS)) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (STAGES - 2)) : "memory");      // (the later steps' requests stay in flight)
        else WAIT_VM(0);
        if (WHAT & VIA_REGS) {                          // the rows that arrived go to their slot at the LDS's full write rate
            for (int u = 0; u < 4; ++u) reinterpret_cast<uint4 *>(rows + (k % STAGES) * kSlot + u * 1024)[lane] = held[u];
        }
        issue_rows(k + STAGES - 1);
        multiply(k);
    }
    WAIT_VM(0);
    float t = 0.f;
    for (int nb = 0; nb < 4; ++nb)
        for (int i = 0; i < 16; ++i) t += acc[nb][i];
    if (t == 12345.678f) sink[blockIdx.x * 64 + lane] = t;           // (never: keeps the work alive)
}

template <unsigned WHAT, int STAGES = 2>
static double run(const char *label, const uint16_t *slab, float *sink, int wgs, int steps, double clk_mhz, int cus)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((mix_kernel<WHAT, STAGES>), dim3(wgs), dim3(64), 0, 0, slab, sink, steps, 7u);
    hipEventRecord(e0, 0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((mix_kernel<WHAT, STAGES>), dim3(wgs), dim3(64), 0, 0, slab, sink, steps, 11u + r);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    const double steps_per_cu = (double)wgs * steps / cus;
    const double clk_per_step_cu = us * clk_mhz / steps_per_cu;
    printf("%-58s %9.1f us   %7.1f clk per step and CU   (%.2f TB/s of rows through LDS)\n", label, us, clk_per_step_cu,
           (double)wgs * steps * 4096.0 / (us * 1e-6) / 1e12);
    return clk_per_step_cu;
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double clk_mhz = p.clockRate / 1e3;
    uint16_t *slab; float *sink;
    hipMalloc(&slab, (size_t)kRows * kRB);
    std::vector<uint16_t> h((size_t)kRows * kRB / 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint16_t)(0x3c00 + (i * 2654435761u >> 25));
    hipMemcpy(slab, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int steps = 64;
    // the north star: 64 slices x 6.4 k steps of 16 record visits = 1 600 steps per CU; here 14 waves per CU x 8 rounds of 64 steps
    const int wgs = cus * 14 * 8;
    hipMalloc(&sink, (size_t)wgs * 64 * 4);
    printf("%s: %d CUs, clock %.0f MHz (the counter the kernel's phase clocks use runs at 100 MHz: tools/tile_prof.py converts)\n", p.name, cus, clk_mhz);
    printf("%d workgroups of one wave (14 resident per CU: 10 KB of LDS, 128 registers), %d steps of 16 rows each\n\n", wgs, steps);
    const double all = run<ALL>("the whole mix", slab, sink, wgs, steps, clk_mhz, cus);
    run<ALL, 3>("the whole mix, THREE row slots (two steps' rows in flight; 14 KB: 11 waves per CU)", slab, sink, wgs, steps, clk_mhz, cus);
    run<ALL, 4>("the whole mix, FOUR row slots (18 KB: 8 waves per CU)", slab, sink, wgs, steps, clk_mhz, cus);
    run<DMA | IDX, 3>("only the row requests, three slots", slab, sink, wgs, steps, clk_mhz, cus);
    run<ALL | VIA_REGS>("the whole mix, rows through registers + ds_write_b128", slab, sink, wgs, steps, clk_mhz, cus);
    run<DMA | IDX | VIA_REGS>("only the row requests, through registers + ds_write_b128", slab, sink, wgs, steps, clk_mhz, cus);
    run<ALL & ~MFMA>("  without the products", slab, sink, wgs, steps, clk_mhz, cus);
    run<ALL & ~TR>("  without the transposing reads", slab, sink, wgs, steps, clk_mhz, cus);
    run<ALL & ~WEIGHTS>("  without the weight tile", slab, sink, wgs, steps, clk_mhz, cus);
    run<ALL & ~IDX>("  without the row-index reads", slab, sink, wgs, steps, clk_mhz, cus);
    run<ALL & ~(DMA | IDX)>("  without the row requests", slab, sink, wgs, steps, clk_mhz, cus);
    run<DMA | IDX>("only the row requests (global -> LDS)", slab, sink, wgs, steps, clk_mhz, cus);
    run<TR | MFMA>("only transposing reads + products", slab, sink, wgs, steps, clk_mhz, cus);
    run<WEIGHTS>("only the weight tile", slab, sink, wgs, steps, clk_mhz, cus);
    // (scaled by TIME, not by the nominal clock: under this load the part does not run at its nominal clock)
    printf("\nnorth star: 6.55 M record visits = 409 600 steps: at this rate (%.1f nominal clk per step and CU) its rounds alone take %.1f us\n",
           all, 409600.0 / cus * all / clk_mhz);
    return 0;
}
