// (RESOLVED in round 6, after this reproducer came back clean: the fault is a packed-fp32 erratum -- tools/ubench/pk_opsel_mfma.hip,
// profiles/r06_experiments.md r06aa.  This program stays as the record that the hazard it tests does NOT exist on MI355X.)
// mfma_war.hip -- is an LDS load that OVERWRITES the A operand registers of a matrix product right behind it ordered after
// the product's read of them?  (the root cause of the sliced forward's round-4 heisenbug.)
//
// What round 6 found on the way here (profiles/r06_experiments.md): of msda_fwd_q8's four guards only ONE matters today -- the
// products as inline assembly with wait states behind them.  The build with the compiler's own products (-DQ8_BUILTIN_MFMA)
// is wrong at full size in fp16 -- queries 6 and 7 of a tile (accumulator rows 12..15: the lanes 48..63), by up to 0.15, in
// every run -- and right in bf16 from instruction streams that are identical but for the opcode.  In that stream a product's
// A operand is reloaded right behind it:
//
//        v_mfma_f32_16x16x32_f16 v[50:53], v[54:57], v[120:123], v[50:53]      ; A = v[54:57]
//        ds_read_b64_tr_b16      v[54:55], v47                                 ; the NEXT K-block's A, into the same registers
//        ds_read_b64_tr_b16      v[56:57], v46
//
// An LDS load takes >= ~60 cycles to return, a product 32 to run: by timing alone this is safe -- unless the product has
// to QUEUE for the matrix pipe behind other waves' products (4 waves per SIMD share one) and reads its operands when it
// starts, not when it issues.  Then the load can land first, and the lanes the pipe reads last get the next block's
// weights.  That is the symptom's signature: only with many busy waves, only lanes 48..63.
//
// This program: per iteration a wave issues PRE products on other registers (the pipe's queue), then the product under
// test, then -- GAP wait states later -- one ds_read_b128 that replaces its A operand with other data (from LDS), and
// compares the result with the same product computed with A left alone.  1 ... 8 waves per SIMD, both opcodes.
//
//   hipcc --offload-arch=gfx950 -O3 mfma_war.hip -o /tmp/mfma_war && /tmp/mfma_war
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// two 16-bit floats in a word, finite, of moderate size: bf16 exponents 120..135, fp16 exponents 8..23
template <bool F16> __device__ __forceinline__ uint32_t two(uint32_t r)
{
    if (F16) {
        const uint32_t a = (r & 0x83ffu) | ((8u + ((r >> 10) & 15u)) << 10);
        const uint32_t b = ((r >> 16) & 0x83ffu) | ((8u + ((r >> 26) & 15u)) << 10);
        return a | (b << 16);
    }
    const uint32_t a = (r & 0x807fu) | ((120u + ((r >> 7) & 15u)) << 7);
    const uint32_t b = ((r >> 16) & 0x807fu) | ((120u + ((r >> 23) & 15u)) << 7);
    return a | (b << 16);
}

#define MFMA_BF "v_mfma_f32_16x16x32_bf16"
#define MFMA_F16 "v_mfma_f32_16x16x32_f16"

// PRE products on scratch registers, the product under test, GAP states, the load over A, the wait, a drain
#define TEST(OP, PRESTR, GAPSTR)                                                                                        \
    asm volatile("s_nop 4\n\t" PRESTR OP " %0, %1, %2, %3\n\t" GAPSTR                                                    \
                 "ds_read_b128 %1, %4\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15"                                   \
                 : "=&v"(d), "+v"(A) : "v"(B), "v"(C), "v"(lds_addr)                                                    \
                 : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "memory")
#define PRE0 ""
#define PRE2(OP) OP " v[100:103], %2, %2, 0\n\t" OP " v[104:107], %2, %2, 0\n\t"
#define PRE4(OP) PRE2(OP) OP " v[100:103], %2, %2, v[100:103]\n\t" OP " v[104:107], %2, %2, v[104:107]\n\t"

template <bool F16, int PRE, int GAP>
__global__ void __launch_bounds__(256) war_kernel(unsigned long long *bad, int iters, uint32_t seed)
{
    __shared__ __attribute__((aligned(16))) uint32_t other[4 * 256];
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = mix(gid * 2654435761u + seed);
    for (int i = threadIdx.x; i < 4 * 256; i += blockDim.x) other[i] = two<F16>(mix(i * 977u + seed));     // what overwrites A
    __syncthreads();
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    const uint32_t lds_addr = (uint32_t)(uintptr_t)(lds_u32 *)other + threadIdx.x * 16u;
    unsigned long long n_bad = 0, n_rows[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        u32x4 A, B;
        f32x4 C;
        for (int j = 0; j < 4; ++j) { s = mix(s + j); A[j] = two<F16>(s); s = mix(s ^ 0x9e3779b9u); B[j] = two<F16>(s); s = mix(s + 77u); C[j] = __uint_as_float((s & 0x007fffffu) | 0x3f000000u); }
        f32x4 ref, d;
        if (F16) asm volatile("s_nop 4\n\t" MFMA_F16 " %0, %1, %2, %3\n\ts_nop 15\n\ts_nop 15" : "=&v"(ref) : "v"(A), "v"(B), "v"(C));
        else asm volatile("s_nop 4\n\t" MFMA_BF " %0, %1, %2, %3\n\ts_nop 15\n\ts_nop 15" : "=&v"(ref) : "v"(A), "v"(B), "v"(C));
#define GO(OP)                                                                            \
        if (PRE == 0 && GAP == 0) TEST(OP, PRE0, "");                                       \
        if (PRE == 0 && GAP == 4) TEST(OP, PRE0, "s_nop 3\n\t");                            \
        if (PRE == 0 && GAP == 16) TEST(OP, PRE0, "s_nop 15\n\t");                          \
        if (PRE == 2 && GAP == 0) TEST(OP, PRE2(OP), "");                                   \
        if (PRE == 2 && GAP == 4) TEST(OP, PRE2(OP), "s_nop 3\n\t");                        \
        if (PRE == 2 && GAP == 16) TEST(OP, PRE2(OP), "s_nop 15\n\t");                      \
        if (PRE == 4 && GAP == 0) TEST(OP, PRE4(OP), "");                                   \
        if (PRE == 4 && GAP == 4) TEST(OP, PRE4(OP), "s_nop 3\n\t");                        \
        if (PRE == 4 && GAP == 8) TEST(OP, PRE4(OP), "s_nop 7\n\t");                        \
        if (PRE == 4 && GAP == 16) TEST(OP, PRE4(OP), "s_nop 15\n\t");
        if (F16) { GO(MFMA_F16) } else { GO(MFMA_BF) }
#undef GO
        bool any = false;
        for (int j = 0; j < 4; ++j) any |= __float_as_uint(d[j]) != __float_as_uint(ref[j]);
        if (any) { ++n_bad; ++n_rows[(threadIdx.x & 63) >> 4]; }
        s ^= A[0];                                                                         // (A now holds the loaded data: keep it alive)
    }
    if (n_bad) {
        atomicAdd(bad, n_bad);
        for (int j = 0; j < 4; ++j) if (n_rows[j]) atomicAdd(bad + 1 + j, n_rows[j]);
    }
    if (s == 0x12345u) atomicAdd(bad + 5, 1ull);
}

template <bool F16, int PRE, int GAP>
static void run(int waves_per_simd, int iters, unsigned long long *dbad)
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int grid = p.multiProcessorCount * waves_per_simd;              // 256-thread blocks: 1 wave per SIMD each
    (void)hipMemset(dbad, 0, 64);
    hipLaunchKernelGGL((war_kernel<F16, PRE, GAP>), dim3(grid), dim3(256), 0, 0, dbad, iters, 4242u + PRE * 17 + GAP);
    (void)hipDeviceSynchronize();
    unsigned long long h[8] = {0};
    (void)hipMemcpy(h, dbad, 64, hipMemcpyDeviceToHost);
    const double n = (double)grid * 4 * iters;                            // wave-products
    printf("  %-5s %d products queued ahead, load %2d states behind, %d waves/SIMD: lane-results that differ %llu of %.3g "
           "(lanes 0-15 / 16-31 / 32-47 / 48-63: %llu / %llu / %llu / %llu)\n",
           F16 ? "f16" : "bf16", PRE, GAP, waves_per_simd, h[0], n * 64, h[1], h[2], h[3], h[4]);
}

int main()
{
    unsigned long long *dbad = nullptr;
    if (hipMalloc(&dbad, 64) != hipSuccess) { printf("no device\n"); return 1; }
    const int iters = 20000;
    printf("v_mfma_f32_16x16x32: an LDS load that overwrites the A operand right behind the product\n");
    for (int w : {1, 2, 4, 8}) {
        run<true, 0, 0>(w, iters, dbad);  run<false, 0, 0>(w, iters, dbad);
        run<true, 0, 4>(w, iters, dbad);  run<true, 0, 16>(w, iters, dbad);
        run<true, 2, 0>(w, iters, dbad);  run<false, 2, 0>(w, iters, dbad);
        run<true, 2, 4>(w, iters, dbad);  run<true, 2, 16>(w, iters, dbad);
        run<true, 4, 0>(w, iters, dbad);  run<false, 4, 0>(w, iters, dbad);
        run<true, 4, 4>(w, iters, dbad);  run<true, 4, 8>(w, iters, dbad);  run<true, 4, 16>(w, iters, dbad);
        run<false, 4, 4>(w, iters, dbad); run<false, 4, 16>(w, iters, dbad);
    }
    (void)hipFree(dbad);
    return 0;
}
