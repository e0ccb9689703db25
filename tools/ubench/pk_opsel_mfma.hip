// pk_opsel_mfma.hip -- does v_pk_mul_f32 with op_sel:[0,1] (the LOW result takes src1's HIGH register) return a wrong low
// result when matrix products are running on the same SIMD?
//
// Where this comes from (profiles/r06_experiments.md, "the sliced forward's heisenbug, decoded"): in the -DQ8_BUILTIN_MFMA
// build of msda_fwd_q8 the bilinear weight of corner 2 (fy * gx * a) of the lanes 48..63 is ZERO now and then.  A probe in
// the kernel shows the instruction:  v_pk_mul_f32 t, {fy, fx}, {gy, gx} op_sel:[0,1] op_sel_hi:[1,0]  -- its high result
// (fx * gy) is right, its low result (fy * gx) is 0 in the last 16 lanes, with both operands intact before and after, wait
// states in front / behind / between making no difference.  The bf16 kernel has no op_sel:[0,1] multiply (and never fails).
//
// This program: every wave alternates a burst of matrix products (form F; forms 5 .. 11: other things a neighbour might run)
// with one checked packed instruction (SEL: which, with which selection).
//   hipcc --offload-arch=gfx950 -O3 pk_opsel_mfma.hip -o /tmp/pk_opsel_mfma && /tmp/pk_opsel_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ float unit(uint32_t r) { return __uint_as_float((r & 0x007fffffu) | 0x3f000000u); }      // [0.5, 1)

// FORM 0: products in place (D = C).  1: D != C, ping-pong.  2: D = the B operand's registers.  3: no products at all.
template <bool F16, int FORM, int SEL, int GAP>
__global__ void __launch_bounds__(1024) k(unsigned long long *bad, int iters, int burst, uint32_t seed)
{
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int wave = threadIdx.x >> 6;
    uint32_t s = mix(gid * 2654435761u + seed);
    u32x4 A, B;
    for (int j = 0; j < 4; ++j) { s = mix(s + j); A[j] = F16 ? 0x2c002c00u | (s & 0x03ff03ffu) : 0x3d803d80u | (s & 0x007f007fu); s = mix(s); B[j] = F16 ? 0x2c002c00u | (s & 0x03ff03ffu) : 0x3d803d80u | (s & 0x007f007fu); }
    f32x4 C = {0.f, 0.f, 0.f, 0.f}, D2 = {0.f, 0.f, 0.f, 0.f};
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    typedef double f64x4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    f32x16 C16;
    for (int j = 0; j < 16; ++j) C16[j] = 0.f;
    f64x4 C8 = {0., 0., 0., 0.};
    u32x2 A2 = {A[0], A[1]}, B2 = {B[0], B[1]};
    f32x2 P2 = {0.f, 0.f}, Q2 = {0.5f, 0.25f};
    u32x4 L4 = {0u, 0u, 0u, 0u};
    __shared__ __attribute__((aligned(16))) unsigned int lds_buf[4 * 1024];
    for (int i = threadIdx.x; i < 4 * 1024; i += blockDim.x) lds_buf[i] = i;
    __syncthreads();
    typedef __attribute__((address_space(3))) unsigned int lds_u32;
    const uint32_t lds_addr = (uint32_t)(uintptr_t)(lds_u32 *)lds_buf + (threadIdx.x & 1023) * 16u;
    unsigned long long n_bad = 0, rows[4] = {0, 0, 0, 0}, zero = 0;
    for (int it = 0; it < iters; ++it) {
        const int nb = burst + ((wave + it) & 3);                                     // (the waves of a SIMD drift apart)
        for (int m = 0; m < nb; ++m) {
            // what ELSE on the SIMD triggers it (forms 5 ..): other matrix shapes, dot products, plain and packed fp32, LDS traffic
            if (FORM == 5) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(C16) : "v"(A), "v"(B));
            if (FORM == 6) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %0" : "+v"(C) : "v"(A2), "v"(B2));
            if (FORM == 7) asm volatile("v_dot2c_f32_f16 %0, %1, %2\n\tv_dot2c_f32_f16 %0, %1, %2\n\tv_dot2c_f32_f16 %0, %1, %2\n\tv_dot2c_f32_f16 %0, %1, %2" : "+v"(C[0]) : "v"(A[0]), "v"(B[0]));
            if (FORM == 8) asm volatile("v_fma_f32 %0, %1, %2, %0\n\tv_fma_f32 %0, %1, %2, %0\n\tv_fma_f32 %0, %1, %2, %0\n\tv_fma_f32 %0, %1, %2, %0" : "+v"(C[0]) : "v"(C[1]), "v"(C[2]));
            if (FORM == 9) asm volatile("v_pk_fma_f32 %0, %1, %1, %0\n\tv_pk_fma_f32 %0, %1, %1, %0\n\tv_pk_fma_f32 %0, %1, %1, %0\n\tv_pk_fma_f32 %0, %1, %1, %0" : "+v"(P2) : "v"(Q2));
            if (FORM == 10) { asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(L4) : "v"(lds_addr)); C[3] += __uint_as_float(L4[0] & 0x3f800000u); }
            if (FORM == 11) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(C8) : "v"(A2), "v"(B2));
            if (FORM == 4) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop 7" : "+v"(C) : "v"(A), "v"(B));   // (as the shipped kernel has them)
            if (FORM == 0) {
                if (F16 && FORM == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(C) : "v"(A), "v"(B));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(C) : "v"(A), "v"(B));
            } else if (FORM == 1) {
                if (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, %1\n\tv_mfma_f32_16x16x32_f16 %1, %2, %3, %0" : "=&v"(D2), "+v"(C) : "v"(A), "v"(B));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %1\n\tv_mfma_f32_16x16x32_bf16 %1, %2, %3, %0" : "=&v"(D2), "+v"(C) : "v"(A), "v"(B));
            } else if (FORM == 2) {
                f32x4 Bd = __builtin_bit_cast(f32x4, B);
                if (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %0, %1\n\ts_nop 7" : "+v"(Bd), "+v"(C) : "v"(A));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %0, %1\n\ts_nop 7" : "+v"(Bd), "+v"(C) : "v"(A));
                C = Bd * 0.001f;
            }
        }
        // the checked multiply
        s = mix(s + it);
        f32x2 fr = {unit(s), unit(mix(s ^ 0x1234567u))}, gr = {unit(mix(s ^ 0x89abcdeu)), unit(mix(s ^ 0x3141592u))}, t;
        asm volatile("" : "+v"(fr), "+v"(gr));
        for (int g = 0; g < GAP; ++g) asm volatile("s_nop 15");
        float a0 = fr[0], a1 = fr[1], b0 = gr[0], b1 = gr[1];                             // what the low / high result multiply (add)
        bool add = false;
        if (SEL == 0) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(t) : "v"(fr), "v"(gr)); b0 = gr[1]; b1 = gr[0]; }
        if (SEL == 1) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=&v"(t) : "v"(fr), "v"(gr)); b1 = gr[0]; }
        if (SEL == 2) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=&v"(t) : "v"(fr), "v"(gr)); a0 = fr[1]; }
        if (SEL == 3) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1]" : "=&v"(t) : "v"(fr), "v"(gr)); a0 = fr[1]; b0 = gr[1]; }
        if (SEL == 4) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(t) : "v"(fr), "v"(gr)); b0 = gr[1]; add = true; }
        if (SEL == 5) { asm volatile("v_pk_fma_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=&v"(t) : "v"(fr), "v"(gr)); a0 = fr[1]; }
        if (SEL == 6) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,0]" : "=&v"(t) : "v"(fr), "v"(gr)); a1 = fr[0]; b1 = gr[0]; }
        // three sources: t = a * b + c with c = {fr.hi, gr.lo} (any two more numbers)
        f32x2 cr = {fr[1], gr[0]};
        float c0 = cr[0], c1 = cr[1];
        bool fma = false;
        if (SEL == 7) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=&v"(t) : "v"(fr), "v"(gr), "v"(cr)); b0 = gr[1]; fma = true; }
        if (SEL == 8) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=&v"(t) : "v"(fr), "v"(gr), "v"(cr)); c0 = cr[1]; fma = true; }
        if (SEL == 9) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=&v"(t) : "v"(fr), "v"(gr), "v"(cr)); a0 = fr[1]; fma = true; }
        if (SEL == 10) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1]" : "=&v"(t) : "v"(fr), "v"(gr), "v"(cr)); b0 = gr[1]; c0 = cr[1]; fma = true; }
        if (SEL == 11) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,1]" : "=&v"(t) : "v"(fr), "v"(gr), "v"(cr)); a0 = fr[1]; b0 = gr[1]; c0 = cr[1]; fma = true; }
        if (SEL == 12) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,0,0]" : "=&v"(t) : "v"(fr), "v"(gr), "v"(cr)); a1 = fr[0]; b1 = gr[0]; c1 = cr[0]; fma = true; }
        if (SEL == 13) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0]" : "=&v"(t) : "v"(fr), "v"(gr)); a0 = fr[1]; add = true; }
        if (SEL == 14) { asm volatile("v_pk_mul_f32 %0, %2, %1 op_sel:[1,0]" : "=&v"(t) : "v"(fr), "v"(gr)); b0 = gr[1]; }     // (the operands swapped: gr.hi * fr.lo -- what a fix would emit)
        bool mov = false;
        if (SEL == 15) { asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=&v"(t) : "v"(fr), "v"(gr)); a0 = fr[0]; a1 = gr[1]; mov = true; }     // lo = src0.lo, hi = src1.hi
        if (SEL == 16) { asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=&v"(t) : "v"(fr), "v"(gr)); a0 = fr[1]; a1 = gr[0]; mov = true; }     // lo = src0.hi, hi = src1.lo
        if (SEL == 17) {                                                                   // 16-bit packed: the selects stay inside one register
            float r;
            asm volatile("v_pk_mul_f16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(r) : "v"(fr[0]), "v"(gr[0]));
            float r2;
            asm volatile("v_pk_mul_f16 %0, %2, %1 op_sel:[1,0] op_sel_hi:[0,1]" : "=&v"(r2) : "v"(fr[0]), "v"(gr[0]));
            t[0] = r; t[1] = 0.f; a0 = r2; a1 = 0.f; mov = true;
        }
        float lo_ref, hi_ref;
        if (mov) { lo_ref = a0; hi_ref = a1; asm volatile("" : "+v"(lo_ref), "+v"(hi_ref)); }
        else if (fma) asm volatile("v_fma_f32 %0, %2, %3, %4\n\tv_fma_f32 %1, %5, %6, %7" : "=&v"(lo_ref), "=&v"(hi_ref) : "v"(a0), "v"(b0), "v"(c0), "v"(a1), "v"(b1), "v"(c1));
        else if (add) asm volatile("v_add_f32 %0, %2, %3\n\tv_add_f32 %1, %4, %5" : "=&v"(lo_ref), "=&v"(hi_ref) : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
        else asm volatile("v_mul_f32 %0, %2, %3\n\tv_mul_f32 %1, %4, %5" : "=&v"(lo_ref), "=&v"(hi_ref) : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
        const bool b = __float_as_uint(t[0]) != __float_as_uint(lo_ref) || __float_as_uint(t[1]) != __float_as_uint(hi_ref);
        if (b) { ++n_bad; ++rows[(threadIdx.x & 63) >> 4]; if (t[0] == 0.f) ++zero; }
    }
    if (n_bad) { atomicAdd(bad, n_bad); for (int j = 0; j < 4; ++j) if (rows[j]) atomicAdd(bad + 1 + j, rows[j]); atomicAdd(bad + 5, zero); }
    if (C[0] + C[1] + C[2] + C[3] + D2[0] + C16[0] + C16[7] + (float)C8[0] + P2[0] + P2[1] == 1.2345f) atomicAdd(bad + 6, 1ull);
}

template <bool F16, int FORM, int SEL, int GAP>
static void run(int lanes, int wgs_per_cu, int iters, int burst, unsigned long long *dbad)
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int grid = p.multiProcessorCount * wgs_per_cu;
    (void)hipMemset(dbad, 0, 64);
    hipLaunchKernelGGL((k<F16, FORM, SEL, GAP>), dim3(grid), dim3(lanes), 0, 0, dbad, iters, burst, 777u + FORM);
    (void)hipDeviceSynchronize();
    unsigned long long h[8] = {0};
    (void)hipMemcpy(h, dbad, 64, hipMemcpyDeviceToHost);
    static const char *names[] = {"pk_mul op_sel:[0,1]", "pk_mul op_sel_hi:[1,0]", "pk_mul op_sel:[1,0]", "pk_mul op_sel:[1,1]", "pk_add op_sel:[0,1]", "pk_fma op_sel:[1,0,0]", "pk_mul op_sel_hi:[0,0]", "pk_fma op_sel:[0,1,0]", "pk_fma op_sel:[0,0,1]", "pk_fma op_sel:[1,0,0] + c", "pk_fma op_sel:[0,1,1]", "pk_fma op_sel:[1,1,1]", "pk_fma op_sel_hi:[0,0,0]", "pk_add op_sel:[1,0]", "pk_mul swapped [1,0]", "pk_mov op_sel:[0,1]", "pk_mov op_sel:[1,0]", "pk_mul_f16 op_sel:[0,1]"};
    printf("  %-4s form %d  %-24s gap %3d  %4d-lane WGs x %d per CU, bursts of %d..%d: wrong %llu of %.3g lane-results "
           "(lanes 0-15 / 16-31 / 32-47 / 48-63: %llu / %llu / %llu / %llu; low result == 0: %llu)\n",
           F16 ? "f16" : "bf16", FORM, names[SEL], GAP * 16, lanes, wgs_per_cu, burst, burst + 3, h[0], (double)grid * lanes * iters, h[1], h[2], h[3], h[4], h[5]);
}

int main()
{
    unsigned long long *dbad = nullptr;
    if (hipMalloc(&dbad, 64) != hipSuccess) { printf("no device\n"); return 1; }
    const int iters = 20000;
    const int burst = 4;
    printf("the instruction, products in place (form 0), no gap, 16 waves per CU\n");
    run<true, 0, 0, 0>(1024, 1, iters, burst, dbad);
    run<true, 0, 1, 0>(1024, 1, iters, burst, dbad);
    run<true, 0, 2, 0>(1024, 1, iters, burst, dbad);
    run<true, 0, 3, 0>(1024, 1, iters, burst, dbad);
    run<true, 0, 4, 0>(1024, 1, iters, burst, dbad);
    run<true, 0, 5, 0>(1024, 1, iters, burst, dbad);
    run<true, 0, 6, 0>(1024, 1, iters, burst, dbad);
    run<true, 0, 7, 0>(1024, 1, iters, burst, dbad);
    run<true, 0, 8, 0>(1024, 1, iters, burst, dbad);
    run<true, 0, 9, 0>(1024, 1, iters, burst, dbad);
    run<true, 0, 10, 0>(1024, 1, iters, burst, dbad);
    run<true, 0, 11, 0>(1024, 1, iters, burst, dbad);
    run<true, 0, 12, 0>(1024, 1, iters, burst, dbad);
    run<true, 0, 13, 0>(1024, 1, iters, burst, dbad);
    run<true, 0, 14, 0>(1024, 1, iters, burst, dbad);
    run<true, 0, 15, 0>(1024, 1, iters, burst, dbad);
    run<true, 0, 16, 0>(1024, 1, iters, burst, dbad);
    run<true, 0, 17, 0>(1024, 1, iters, burst, dbad);
    printf("whose products: the wave's own (one wave per SIMD) or the neighbours' (wait states between the burst and the instruction)\n");
    run<true, 0, 0, 0>(256, 1, iters, burst, dbad);
    run<true, 0, 0, 1>(256, 1, iters, burst, dbad);
    run<true, 0, 0, 4>(256, 1, iters, burst, dbad);
    run<true, 0, 0, 1>(1024, 1, iters, burst, dbad);
    run<true, 0, 0, 4>(1024, 1, iters, burst, dbad);
    run<true, 0, 0, 16>(1024, 1, iters, burst, dbad);
    printf("what the neighbours run (op_sel:[0,1] multiply, no gap, 16 waves per CU): 5 = v_mfma_f32_32x32x16_bf16, 6 = v_mfma_f32_4x4x4_16b_f16,\n"
           "  7 = v_dot2c_f32_f16 x 4, 8 = v_fma_f32 x 4, 9 = v_pk_fma_f32 x 4, 10 = ds_read_b128 + wait, 11 = v_mfma_f64_16x16x4_f64\n");
    run<true, 5, 0, 0>(1024, 1, iters, burst, dbad);
    run<true, 6, 0, 0>(1024, 1, iters, burst, dbad);
    run<true, 7, 0, 0>(1024, 1, iters, burst, dbad);
    run<true, 8, 0, 0>(1024, 1, iters, burst, dbad);
    run<true, 9, 0, 0>(1024, 1, iters, burst, dbad);
    run<true, 10, 0, 0>(1024, 1, iters, burst, dbad);
    run<true, 11, 0, 0>(1024, 1, iters, burst, dbad);
    printf("the forms of the products\n");
    run<true, 3, 0, 0>(1024, 1, iters, burst, dbad);
    run<true, 4, 0, 0>(1024, 1, iters, burst, dbad);
    run<true, 4, 0, 4>(1024, 1, iters, burst, dbad);
    run<true, 1, 0, 4>(1024, 1, iters, burst, dbad);
    run<true, 2, 0, 4>(1024, 1, iters, burst, dbad);
    run<false, 0, 0, 4>(1024, 1, iters, burst, dbad);
    (void)hipFree(dbad);
    return 0;
}
