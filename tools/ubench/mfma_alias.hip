// mfma_alias.hip -- does v_mfma_f32_16x16x32_bf16 compute the same when its DESTINATION registers are its own A or B
// operand registers?  
//
// Background.  Round 4's sliced forward produced wrong rows now and then under load (profiles/r04_experiments.md r04f-m).
// One suspect was what LLVM does with the builtin: for a product with a 4-register destination it does not mark the
// destination early-clobber, so the register allocator may place the result ON the A or B operand
// (v_mfma_f32_16x16x32_bf16 v[78:81], v[70:73], v[78:81], v[58:61]); tools/mfma_overlap.py counts 43 such products in
// msda_fwd_mma, 4 in msda_taps_mma, several in msda_fwd_wq.  The 4-pass product reads its operands over its passes.  If
// the hardware wrote the first result rows before the last operand rows were read, those kernels would be wrong under the
// same conditions.  This program runs the three encodings side by side -- destination on fresh registers, on A, on B --
// on the same operands, millions of times per wave, at 1 ... 8 waves per SIMD, with and without LDS traffic between
// the products, and counts results that differ bit for bit.
//
//   hipcc --offload-arch=gfx950 -O3 mfma_alias.hip -o /tmp/mfma_alias && /tmp/mfma_alias
//
// The products are inline assembly (the only way to force the register overlap), fenced by wait states on both sides
// because the hazard recogniser does not see into the assembly.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// two bf16 in a word, finite, of moderate size (exponents 120 .. 135)
__device__ __forceinline__ uint32_t bf2(uint32_t r) {
    const uint32_t a = (r & 0x807fu) | ((120u + ((r >> 7) & 15u)) << 7);
    const uint32_t b = ((r >> 16) & 0x807fu) | ((120u + ((r >> 23) & 15u)) << 7);
    return a | (b << 16);
}

template <bool LDS_TRAFFIC>
__global__ void __launch_bounds__(256) alias_kernel(unsigned long long *bad, int iters, uint32_t seed)
{
    __shared__ uint32_t sh[4096];
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = mix(gid * 2654435761u + seed);
    unsigned long long n_a = 0, n_b = 0, n_chain = 0;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sh[i] = mix(i + seed);
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        u32x4 A, B;
        f32x4 C;
        for (int j = 0; j < 4; ++j) { s = mix(s + j); A[j] = bf2(s); s = mix(s ^ 0x9e3779b9u); B[j] = bf2(s); s = mix(s + 77u); C[j] = __uint_as_float((s & 0x007fffffu) | 0x3f000000u); }
        if (LDS_TRAFFIC) {                                              // keep the LDS queue of the CU busy around the products
            const uint32_t a = sh[(s >> 8) & 4095], b = sh[(s >> 20) & 4095];
            A[0] ^= (a & 0x007f007fu); B[1] ^= (b & 0x007f007fu);
        }
        f32x4 d0, d1, d2, d3, d4;
        // reference: destination in registers of its own (early clobber)
        asm volatile("s_nop 4\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %3\n\ts_nop 15" : "=&v"(d0) : "v"(A), "v"(B), "v"(C));
        // destination ON the A operand
        { u32x4 t = A; asm volatile("s_nop 4\n\tv_mfma_f32_16x16x32_bf16 %0, %0, %1, %2\n\ts_nop 15" : "+v"(t) : "v"(B), "v"(C)); d1 = __builtin_bit_cast(f32x4, t); }
        // destination ON the B operand
        { u32x4 t = B; asm volatile("s_nop 4\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %0, %2\n\ts_nop 15" : "+v"(t) : "v"(A), "v"(C)); d2 = __builtin_bit_cast(f32x4, t); }
        // a chain as the kernels have it: product 1 lands on its A operand, product 2 accumulates onto it at once
        asm volatile("s_nop 4\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %3\n\tv_mfma_f32_16x16x32_bf16 %0, %2, %1, %0\n\ts_nop 15"
                     : "=&v"(d3) : "v"(A), "v"(B), "v"(C));
        { u32x4 t = A; u32x4 a2 = A;
          asm volatile("s_nop 4\n\tv_mfma_f32_16x16x32_bf16 %0, %0, %2, %3\n\tv_mfma_f32_16x16x32_bf16 %0, %2, %1, %0\n\ts_nop 15"
                       : "+&v"(t) : "v"(a2), "v"(B), "v"(C)); d4 = __builtin_bit_cast(f32x4, t); }
        for (int j = 0; j < 4; ++j) {
            n_a += __float_as_uint(d0[j]) != __float_as_uint(d1[j]);
            n_b += __float_as_uint(d0[j]) != __float_as_uint(d2[j]);
            n_chain += __float_as_uint(d3[j]) != __float_as_uint(d4[j]);
        }
        if (LDS_TRAFFIC) sh[(s >> 4) & 4095] = __float_as_uint(d0[0]) ^ s;
    }
    if (n_a) atomicAdd(&bad[0], n_a);
    if (n_b) atomicAdd(&bad[1], n_b);
    if (n_chain) atomicAdd(&bad[2], n_chain);
}

int main()
{
    unsigned long long *bad;
    hipMalloc(&bad, 3 * sizeof(unsigned long long));
    int dev = 0, cus = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int iters = 20000;
    printf("v_mfma_f32_16x16x32_bf16: destination on fresh registers vs ON the A / B operand, %d CUs, %d iterations per wave\n", cus, iters);
    for (int lds = 0; lds < 2; ++lds)
        for (int wps : {1, 2, 4, 5, 6, 8}) {                            // waves per SIMD (256-lane workgroups = one wave per SIMD each)
            hipMemset(bad, 0, 3 * sizeof(unsigned long long));
            const int grid = cus * wps;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (lds) hipLaunchKernelGGL(alias_kernel<true>, dim3(grid), dim3(256), 0, 0, bad, iters, 12345u + wps);
            else hipLaunchKernelGGL(alias_kernel<false>, dim3(grid), dim3(256), 0, 0, bad, iters, 12345u + wps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[3];
            hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
            const double prods = (double)grid * 4 * iters;
            printf("  %d waves/SIMD, LDS traffic %s: %.3g products per encoding in %.1f ms; result words that differ: dst on A %llu, dst on B %llu, chained %llu\n",
                   wps, lds ? "yes" : "no ", prods, ms, h[0], h[1], h[2]);
        }
    return 0;
}
