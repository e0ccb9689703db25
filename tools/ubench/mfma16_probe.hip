// Probes of the gfx950 behaviours the LDS-resident forward (csrc/msda_fwd_mma.hip) relies on, dumped so
// that a wrong assumption can be read off without another GPU run:
//   Q1  v_mfma_f32_16x16x32_{bf16,f16}: A lane l = row l%16, k = 8*(l/16)+i; B lane l = column l%16,
//       k = 8*(l/16)+i; D lane l = column l%16, rows 4*(l/16)+i  -- D = A . B against a host product
//   Q2  the same product with B fetched by two ds_read_b64_tr_b16 from GATHERED rows (every K row at its
//       own address, 8-byte pieces supplied by lanes 4e+c of each 16-lane group)
//   Q3  DPP row_ror:8 moves lane n ^ 8 -> lane n inside a row of 16
//   Q4  a 1024-lane workgroup with 160 KiB of dynamic LDS launches and sees all of it
// Build: hipcc --offload-arch=gfx950 -O3 mfma16_probe.hip -o mfma16_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A [16 x 32] row-major, B [32 x 16] row-major (K rows of 16 columns), D [16 x 16]
template <bool BF>
__global__ void q1(const unsigned short *A, const unsigned short *B, float *D)
{
    const int l = threadIdx.x;
    s16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (short)A[(l % 16) * 32 + 8 * (l / 16) + i];
        b[i] = (short)B[(8 * (l / 16) + i) * 16 + l % 16];
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    if (BF) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(4 * (l / 16) + i) * 16 + l % 16] = c[i];
}

// Q2: 32 K rows of 16 halfwords each live at rowaddr[k] (bytes, 8-aligned) in LDS
__global__ void q2(const unsigned short *A, const unsigned short *B, const int *rowaddr, float *D)
{
    __shared__ __attribute__((aligned(256))) unsigned char lds[32768];
    const int l = threadIdx.x;
    for (int i = l; i < 32 * 16; i += 64) {
        const int k = i / 16, n = i % 16;
        *reinterpret_cast<unsigned short *>(lds + rowaddr[k] + 2 * n) = B[k * 16 + n];
    }
    __syncthreads();
    s16x8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (short)A[(l % 16) * 32 + 8 * (l / 16) + i];
    const int G = l >> 4, e = (l >> 2) & 3, c4 = l & 3;
    for (int t = 0; t < 2; ++t) {
        const int k = 8 * G + 4 * t + e;
        const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(lds + rowaddr[k] + 8 * c4));
        b[4 * t] = v[0]; b[4 * t + 1] = v[1]; b[4 * t + 2] = v[2]; b[4 * t + 3] = v[3];
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(4 * (l / 16) + i) * 16 + l % 16] = c[i];
}

__global__ void q3(float *out)
{
    const float v = (float)threadIdx.x;
    out[threadIdx.x] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, true));
}

__global__ void __launch_bounds__(1024) q4(unsigned *out, int bytes)
{
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    unsigned *w = reinterpret_cast<unsigned *>(smem);
    for (int i = threadIdx.x; i < bytes / 4; i += 1024) w[i] = (unsigned)i * 2654435761u;
    __syncthreads();
    unsigned acc = 0;
    for (int i = threadIdx.x; i < bytes / 4; i += 1024) acc += w[(i + 4097) % (bytes / 4)] ^ (unsigned)(((i + 4097) % (bytes / 4)) * 2654435761u);
    atomicAdd(out, acc);
}

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short s; memcpy(&s, &h, 2); return s; }
static float h2f(unsigned short s) { _Float16 h; memcpy(&h, &s, 2); return (float)h; }

int main()
{
    srand(3);
    unsigned short *dA, *dB; float *dD; int *dR;
    CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dD, 1024)); CK(hipMalloc(&dR, 128));
    for (int bf = 1; bf >= 0; --bf) {
        std::vector<unsigned short> A(512), B(512);
        std::vector<float> Af(512), Bf(512);
        for (int i = 0; i < 512; ++i) {
            const float a = (rand() % 17 - 8) / 8.f, b = (rand() % 33 - 16) / 4.f;       // asymmetric on purpose
            A[i] = bf ? f2bf(a) : f2h(a); B[i] = bf ? f2bf(b) : f2h(b);
            Af[i] = bf ? bf2f(A[i]) : h2f(A[i]); Bf[i] = bf ? bf2f(B[i]) : h2f(B[i]);
        }
        CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
        std::vector<double> want(256, 0.0);
        for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) for (int k = 0; k < 32; ++k) want[m * 16 + n] += (double)Af[m * 32 + k] * Bf[k * 16 + n];
        std::vector<float> D(256);
        if (bf) hipLaunchKernelGGL(q1<true>, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        else hipLaunchKernelGGL(q1<false>, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 256; ++i) if (fabs(want[i] - D[i]) > 1e-3) ++bad;
        printf("Q1 mfma_f32_16x16x32_%s operand / result layout: %s (%d of 256 wrong)\n", bf ? "bf16" : "f16", bad ? "MISMATCH" : "OK", bad);
        if (bad) {
            // which (m', n') of the host product does each D slot hold?
            for (int i = 0; i < 256; ++i) {
                int hit = -1;
                for (int j = 0; j < 256; ++j) if (fabs(want[j] - D[i]) < 1e-4) { hit = hit < 0 ? j : -2; }
                printf("  D[%2d][%2d] = %9.4f  want %9.4f  (matches host element %d)\n", i / 16, i % 16, D[i], want[i], hit);
            }
        }
        if (bf) {
            int rowaddr[32];
            for (int k = 0; k < 32; ++k) rowaddr[k] = ((k * 37 + 11) % 97) * 288 + ((k * 5) % 8) * 32;      // gathered, 8-aligned, distinct
            CK(hipMemcpy(dR, rowaddr, 128, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(q2, dim3(1), dim3(64), 0, 0, dA, dB, dR, dD);
            CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
            bad = 0;
            for (int i = 0; i < 256; ++i) if (fabs(want[i] - D[i]) > 1e-3) ++bad;
            printf("Q2 the same product, B by transposing reads of gathered rows: %s (%d of 256 wrong)\n", bad ? "MISMATCH" : "OK", bad);
        }
    }
    {
        float *d; CK(hipMalloc(&d, 256));
        hipLaunchKernelGGL(q3, dim3(1), dim3(64), 0, 0, d);
        float h[64]; CK(hipMemcpy(h, d, 256, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) if ((int)h[l] != (l ^ 8)) ++bad;
        printf("Q3 DPP row_ror:8 = lane ^ 8: %s", bad ? "MISMATCH:" : "OK\n");
        if (bad) { for (int l = 0; l < 64; ++l) printf(" %d", (int)h[l]); printf("\n"); }
    }
    {
        unsigned *d; CK(hipMalloc(&d, 4)); CK(hipMemset(d, 0, 4));
        for (int kb = 160; kb >= 128; kb -= 16) {
            const int bytes = kb * 1024;
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&q4), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e == hipSuccess) { hipLaunchKernelGGL(q4, dim3(2), dim3(1024), bytes, 0, d, bytes); e = hipGetLastError(); }
            if (e == hipSuccess) e = hipDeviceSynchronize();
            unsigned h = 1; if (e == hipSuccess) CK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
            printf("Q4 1024 lanes + %d KiB dynamic LDS: %s%s\n", kb, e == hipSuccess ? "launches" : hipGetErrorString(e), e == hipSuccess ? (h == 0 ? ", contents OK" : ", CONTENTS WRONG") : "");
            if (e == hipSuccess) break;
        }
    }
    return 0;
}
