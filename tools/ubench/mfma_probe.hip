// Probes of the two gfx950 primitives the matrix-core grad_value reduce (csrc/msda_bwd_tile.hip) is
// built on, dumped as tables so that a wrong assumption can be read off without another GPU run:
//   P1  ds_read_b64_tr_b16 with per-lane addresses: which 16-bit element lands in which lane / slot
//   P2  global_load_lds_dwordx4: where in LDS the 16 bytes of each lane land
//   P3  v_mfma_f32_32x32x16_bf16 with B read by two transposing reads from rows at arbitrary
//       (swizzled) places: D = A . B against a host product
// Build: hipcc --offload-arch=gfx950 -O3 mfma_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// P1: LDS holds its own halfword index; lane l reads 8 bytes at addr[l]
__global__ void p1(const int *addr, short *out)
{
    __shared__ short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4 *)((char *)lds + addr[threadIdx.x]));
    for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = v[i];
}

// P2: lane l fetches 16 bytes from src + off[l]; LDS destination base 1024
__global__ void p2(const unsigned *src, const int *off, unsigned *out)
{
    __shared__ __attribute__((aligned(1024))) unsigned lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const char *)src + off[threadIdx.x]),
                                     (__attribute__((address_space(3))) void *)(lds + 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}

// P3: A [32 x 16] bf16 row-major at lds_a; 16 rows of 256 bytes (128 bf16 channels) swizzled like the
// kernel does; D[32 x 32] for column block nb
__global__ void p3(const unsigned short *A, const unsigned short *rows, float *D, int nb)
{
    __shared__ __attribute__((aligned(1024))) unsigned char lds[4096 + 1024];
    const int lane = threadIdx.x;
    // rows through the same DMA + swizzle as the kernel: 4 instructions of 4 rows
    for (int u = 0; u < 4; ++u) {
        const int rr = u * 4 + lane / 16;
        const int chunk = (lane % 16) ^ (4 * (rr % 4));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const char *)rows + rr * 256 + chunk * 16),
                                         (__attribute__((address_space(3))) void *)(lds + u * 1024), 16, 0, 0);
    }
    for (int i = lane; i < 512; i += 64) ((unsigned short *)(lds + 4096))[i] = A[i];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const s16x8 a = *reinterpret_cast<const s16x8 *>(lds + 4096 + (lane & 31) * 32 + (lane >> 5) * 16);
    const int g4 = lane >> 4, j16 = lane & 15;
    s16x8 b;
    for (int t = 0; t < 2; ++t) {
        const int krow = 8 * (g4 >> 1) + 4 * t + (j16 >> 2);
        const int cb = nb * 64 + (g4 & 1) * 32 + (j16 & 3) * 8;
        const int off = krow * 256 + (((cb >> 4) ^ (4 * (krow % 4))) << 4) + (cb & 15);
        const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(lds + off));
        b[4 * t] = v[0]; b[4 * t + 1] = v[1]; b[4 * t + 2] = v[2]; b[4 * t + 3] = v[3];
    }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        D[row * 32 + (lane & 31)] = c[r];
    }
}

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main()
{
    // ---- P1a: canonical addresses lane * 8
    int h_addr[64]; short h_out[256];
    int *d_addr; short *d_out;
    CK(hipMalloc(&d_addr, sizeof h_addr)); CK(hipMalloc(&d_out, sizeof h_out));
    for (int variant = 0; variant < 2; ++variant) {
        // variant 1: the 4 rows of each 16-lane group 512 bytes apart (row = lane%16/4), chunk = lane%4
        for (int l = 0; l < 64; ++l)
            h_addr[l] = variant == 0 ? l * 8 : (l / 16) * 2048 + ((l % 16) / 4) * 512 + (l % 4) * 8;
        CK(hipMemcpy(d_addr, h_addr, sizeof h_addr, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(p1, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        CK(hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost));
        printf("P1 variant %d (halfword index read by lane: e0 e1 e2 e3), expected by the kernel: e = [row e of the group][column lane%%16]\n", variant);
        int bad = 0;
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d addr %5d :", l, h_addr[l]);
            for (int e = 0; e < 4; ++e) {
                printf(" %5d", h_out[l * 4 + e]);
                // expectation: element e comes from the address given by lane (group*16 + 4e + (l%16)/4), halfword (l%16)%4
                const int src_lane = (l / 16) * 16 + 4 * e + (l % 16) / 4;
                const int want = h_addr[src_lane] / 2 + (l % 16) % 4;
                if (h_out[l * 4 + e] != (short)want) ++bad;
            }
            printf("\n");
        }
        printf("P1 variant %d: %s (%d mismatches against the kernel's assumption)\n", variant, bad ? "MISMATCH" : "OK", bad);
    }
    // ---- P2
    {
        std::vector<unsigned> src(4096);
        for (int i = 0; i < 4096; ++i) src[i] = i;
        int h_off[64];
        for (int l = 0; l < 64; ++l) h_off[l] = ((l * 37) % 200) * 16;
        unsigned *d_src, *d_o; int *d_off;
        CK(hipMalloc(&d_src, 4096 * 4)); CK(hipMalloc(&d_o, 4096)); CK(hipMalloc(&d_off, 256));
        CK(hipMemcpy(d_src, src.data(), 4096 * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_off, h_off, 256, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(p2, dim3(1), dim3(64), 0, 0, d_src, d_off, d_o);
        std::vector<unsigned> o(1024);
        CK(hipMemcpy(o.data(), d_o, 4096, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int k = 0; k < 4; ++k)
                if (o[256 + l * 4 + k] != (unsigned)(h_off[l] / 4 + k)) ++bad;
        for (int i = 0; i < 256; ++i) if (o[i] != 0xdeadbeefu) ++bad;
        for (int i = 512; i < 1024; ++i) if (o[i] != 0xdeadbeefu) ++bad;
        printf("P2 LDS-DMA placement (base + lane * 16): %s (%d mismatches)\n", bad ? "MISMATCH" : "OK", bad);
        if (bad) for (int i = 240; i < 530; i += 1) printf("  lds[%d] = %u\n", i, o[i]);
    }
    // ---- P3
    {
        std::vector<unsigned short> A(512), R(16 * 128);
        srand(1);
        for (auto &a : A) a = f2bf((rand() % 17 - 8) / 8.f);
        for (auto &r : R) r = f2bf((rand() % 33 - 16) / 4.f);
        unsigned short *dA, *dR; float *dD;
        CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dR, 4096)); CK(hipMalloc(&dD, 4096));
        CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice));
        CK(hipMemcpy(dR, R.data(), 4096, hipMemcpyHostToDevice));
        for (int nb = 0; nb < 4; ++nb) {
            hipLaunchKernelGGL(p3, dim3(1), dim3(64), 0, 0, dA, dR, dD, nb);
            std::vector<float> D(1024);
            CK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
            int bad = 0; double worst = 0;
            for (int m = 0; m < 32; ++m)
                for (int n = 0; n < 32; ++n) {
                    double want = 0;
                    for (int k = 0; k < 16; ++k) want += (double)bf2f(A[m * 16 + k]) * bf2f(R[k * 128 + nb * 32 + n]);
                    const double e = fabs(want - D[m * 32 + n]);
                    if (e > 1e-3) ++bad;
                    if (e > worst) worst = e;
                }
            printf("P3 MFMA with transposing reads, column block %d: %s (%d of 1024 wrong, worst %.3g)\n", nb, bad ? "MISMATCH" : "OK", bad, worst);
        }
    }
    return 0;
}
