// Why does the forward's row stream run at ~25 TB/s when gather2's random rows reach 31-34 TB/s out
// of the same L2-resident slab?  This ubench starts from the forward's exact launch shape (one
// 256-lane workgroup per 16 queries of a (b, h), 16 taps x 4 rows per query, 64 loads per wave, then
// exit) and changes one thing at a time towards gather2's persistent loop.
//   rows     5440 (north-star S, hash % S) | 4096 (power of two)
//   slabs    8 batches walked in launch order | one slab
//   life     64 loads per wave (the forward) | x4 | x16 | persistent grid of 2048 workgroups
//   flight   4+4 pipelined (vmcnt(4))       | 8, wait for all
//   corners  4 independent random rows      | the 2x2 footprint of one random pixel (x, x+1, x+W, x+W+1)
// Build: hipcc --offload-arch=gfx950 -O3 gather3.hip -o gather3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 ld16(const char *p)
{
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p));
    return v;
}
__device__ __forceinline__ void keep(const u32x4 &v) { asm volatile("" :: "v"(v)); }
__device__ __forceinline__ unsigned mix(unsigned x)
{
    x *= 0x9E3779B1u; x ^= x >> 15; x *= 0x85EBCA77u; x ^= x >> 13;
    return x;
}

struct P {
    int S, log2S;          // rows per slab; log2S > 0: row = hash >> (32 - log2S), else hash % S
    int B, QT, H;          // launch shape: tiles = B * QT * H
    int taps;              // taps per query (x 4 rows)
    int reps;              // tiles per workgroup, consecutive (life x reps)
    int persistent;        // 1: grid = 2048, workgroup g walks tiles g, g + 2048, ...
    int wait_all;          // 1: 8 loads, vmcnt(0); 0: 4 + 4 pipelined
    int footprint;         // 1: 2x2 footprint of a random pixel in a W x W level
    int W;
    int buffer;            // 1: buffer_load_dwordx4 through a slab descriptor (32-bit offsets) instead of global_load
    int store;             // 1: every lane stores 16 bytes per tile (the forward's output row)
    int lds_kb;            // static LDS claimed per workgroup (occupancy like the forward's 18 KB) + 2 barriers per tile
};

__device__ __forceinline__ u32x4 ldb(__amdgpu_buffer_rsrc_t r, unsigned off)
{
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
}

template <int LDS_KB>
__global__ void __launch_bounds__(256) fwd_like(const char *base, P p, unsigned salt, char *outp)
{
    __shared__ char lds[LDS_KB ? LDS_KB * 1024 : 16];
    if (LDS_KB) { lds[threadIdx.x * 64 % (LDS_KB * 1024)] = 1; __syncthreads(); }
    const int lane16 = threadIdx.x % 16, qi = threadIdx.x / 16;
    const long long n_tiles = (long long)p.B * p.QT * p.H;
    long long tile = p.persistent ? blockIdx.x : (long long)blockIdx.x * p.reps;
    const long long step = p.persistent ? gridDim.x : 1;
    const long long end = p.persistent ? n_tiles : min(n_tiles, tile + p.reps);
    for (; tile < end; tile += step) {
        const int h = (int)(tile % p.H);
        const long long t = tile / p.H;
        const int b = (int)(t / p.QT);
        const char *slab = base + ((size_t)b * p.S * p.H + h) * 256 + lane16 * 16;
        const unsigned q = (unsigned)(t * 16 + qi);
        const size_t rstride = (size_t)p.H * 256;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(slab - lane16 * 16), (short)0, (int)(p.S * rstride), 0x00020000);
        if (LDS_KB) __syncthreads();
        auto rows = [&](int k, unsigned (&r)[4]) {
            unsigned x = mix(q * 64u + (unsigned)k + salt);
            if (p.footprint) {
                // footprint == 2: the north-star pyramid, taps 0-3 in 64x64, 4-7 in 32x32, 8-11 in 16x16, 12-15 in 8x8
                const int lvl = p.footprint == 2 ? (k >> 2) & 3 : 0;
                const unsigned W = p.footprint == 2 ? 64u >> lvl : (unsigned)p.W;
                const unsigned st = p.footprint == 2 ? (lvl == 0 ? 0u : lvl == 1 ? 4096u : lvl == 2 ? 5120u : 5376u) : 0u;
                const unsigned px = ((x & 0xffffu) * (W - 1)) >> 16, py = ((x >> 16) * (W - 1)) >> 16;   // multiply-shift, no division
                const unsigned r0 = st + py * W + px;
                const unsigned Wp = W; (void)Wp;
#define W_ W
                r[0] = r0; r[1] = r0 + 1; r[2] = r0 + W_; r[3] = r0 + W_ + 1;
#undef W_
            } else {
                for (int c = 0; c < 4; ++c) {
                    x = mix(x + c);
                    r[c] = p.log2S > 0 ? x >> (32 - p.log2S) : x % (unsigned)p.S;
                }
            }
        };
        if (p.wait_all) {
            for (int k = 0; k < p.taps; k += 2) {
                unsigned r0[4], r1[4];
                rows(k, r0); rows(k + 1, r1);
                u32x4 v[8];
                for (int c = 0; c < 4; ++c) v[c] = ld16(slab + r0[c] * rstride);
                for (int c = 0; c < 4; ++c) v[4 + c] = ld16(slab + r1[c] * rstride);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                for (int c = 0; c < 8; ++c) keep(v[c]);
            }
        } else {
            unsigned r[4];
            u32x4 a[4], bb[4];
            rows(0, r);
#define LD(dst, rr) dst = p.buffer ? ldb(rsrc, (unsigned)(rr * rstride) + lane16 * 16) : ld16(slab + rr * rstride)
            for (int c = 0; c < 4; ++c) LD(a[c], r[c]);
            for (int k = 0; k < p.taps; k += 2) {
                rows(k + 1, r);
                for (int c = 0; c < 4; ++c) LD(bb[c], r[c]);
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                for (int c = 0; c < 4; ++c) keep(a[c]);
                rows(k + 2, r);
                if (k + 2 < p.taps) { for (int c = 0; c < 4; ++c) LD(a[c], r[c]); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                for (int c = 0; c < 4; ++c) keep(bb[c]);
            }
            if (p.store) *(u32x4 *)(outp + ((size_t)tile * 256 + threadIdx.x) * 16) = bb[0];
        }
        if (LDS_KB) __syncthreads();
    }
}

static char *g_out;
static void run(const char *name, const char *buf, P p)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long long tiles = (long long)p.B * p.QT * p.H;
    const int grid = p.persistent ? 2048 : (int)((tiles + p.reps - 1) / p.reps);
    auto launch = [&](unsigned salt) {
        switch (p.lds_kb) {
        case 0: hipLaunchKernelGGL(fwd_like<0>, dim3(grid), dim3(256), 0, 0, buf, p, salt, g_out); break;
        case 8: hipLaunchKernelGGL(fwd_like<8>, dim3(grid), dim3(256), 0, 0, buf, p, salt, g_out); break;
        case 16: hipLaunchKernelGGL(fwd_like<16>, dim3(grid), dim3(256), 0, 0, buf, p, salt, g_out); break;
        case 18: hipLaunchKernelGGL(fwd_like<18>, dim3(grid), dim3(256), 0, 0, buf, p, salt, g_out); break;
        case 19: hipLaunchKernelGGL(fwd_like<19>, dim3(grid), dim3(256), 0, 0, buf, p, salt, g_out); break;
        case 20: hipLaunchKernelGGL(fwd_like<20>, dim3(grid), dim3(256), 0, 0, buf, p, salt, g_out); break;
        case 24: hipLaunchKernelGGL(fwd_like<24>, dim3(grid), dim3(256), 0, 0, buf, p, salt, g_out); break;
        case 40: hipLaunchKernelGGL(fwd_like<40>, dim3(grid), dim3(256), 0, 0, buf, p, salt, g_out); break;
        default: printf("no such LDS size\n"); exit(1);
        }
    };
    for (int w = 0; w < 2; ++w) launch(17u + w);
    CK(hipEventRecord(e0));
    const int n = 10;
    for (int i = 0; i < n; ++i) launch(100u + i);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= n;
    const double bytes = (double)tiles * 16 * p.taps * 4 * 256;
    printf("%-100s %8.1f us  %6.2f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

int main()
{
    const size_t bytes = 8ull * 8192 * 2048;
    char *buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 1, bytes));
    CK(hipMalloc(&g_out, 8ull * 256 * 8 * 256 * 16));
    P f = {5440, 0, 8, 256, 8, 16, 1, 0, 0, 0, 64, 0, 0, 0};
    {
        P q = f; q.footprint = 2;
        run("pyramid rows, forward launch shape, global_load", buf, q);
        { P p = q; p.buffer = 1; run("  + buffer_load through a slab descriptor", buf, p); }
        { P p = q; p.store = 1; run("  + 16-byte store per lane per tile (output row)", buf, p); }
        for (int kb : {8, 16, 18, 19, 20, 24, 40}) { P p = q; p.lds_kb = kb; char nm[128]; snprintf(nm, sizeof nm, "  + %d KB LDS per workgroup, 2 barriers per tile", kb); run(nm, buf, p); }
        { P p = q; p.buffer = 1; p.store = 1; p.lds_kb = 18; run("  + buffer, store, 18 KB LDS", buf, p); }
        { P p = q; p.taps = 64; p.QT = 64; run("  + 64 taps per query (4x longer waves, quarter the tiles)", buf, p); }
    }
    run("forward shape: S=5440 (hash % S), 8 slabs, 64 loads per wave then exit, 4+4 pipelined", buf, f);
    { P p = f; p.wait_all = 1; run("  + 8 in flight, wait for all", buf, p); }
    { P p = f; p.S = 4096; p.log2S = 12; run("  + S = 4096 (power of two rows)", buf, p); }
    { P p = f; p.S = 8192; p.log2S = 13; run("  + S = 8192", buf, p); }
    { P p = f; p.B = 1; p.QT = 2048; run("  + one slab (B=1, 8x the tiles)", buf, p); }
    { P p = f; p.reps = 4; run("  + 4 tiles per workgroup", buf, p); }
    { P p = f; p.reps = 16; run("  + 16 tiles per workgroup", buf, p); }
    { P p = f; p.persistent = 1; run("  + persistent grid of 2048 workgroups (tile = g, g + 2048, ..)", buf, p); }
    { P p = f; p.persistent = 1; p.wait_all = 1; run("  + persistent, 8 in flight wait for all", buf, p); }
    { P p = f; p.persistent = 1; p.S = 4096; p.log2S = 12; run("  + persistent, S = 4096", buf, p); }
    { P p = f; p.persistent = 1; p.S = 4096; p.log2S = 12; p.B = 1; p.QT = 2048; p.wait_all = 1; run("  + persistent, S = 4096, one slab, wait for all (= gather2)", buf, p); }
    { P p = f; p.footprint = 1; p.W = 64; run("  + 2x2 footprints in a 64x64 level (rows x, x+1, x+W, x+W+1)", buf, p); }
    { P p = f; p.footprint = 1; p.W = 64; p.persistent = 1; run("  + 2x2 footprints, persistent", buf, p); }
    { P p = f; p.footprint = 1; p.W = 16; run("  + 2x2 footprints in a 16x16 level (64 KiB: L1-sized)", buf, p); }
    { P p = f; p.footprint = 2; run("  + the north-star pyramid: 4 taps each in 64^2, 32^2, 16^2, 8^2 (the real forward's rows)", buf, p); }
    { P p = f; p.footprint = 2; p.wait_all = 1; run("  + pyramid, 8 in flight wait for all", buf, p); }
    { P p = f; p.footprint = 2; p.persistent = 1; run("  + pyramid, persistent", buf, p); }
    { P p = f; p.footprint = 1; p.W = 32; run("  + 2x2 footprints in a 32x32 level", buf, p); }
    { P p = f; p.footprint = 1; p.W = 8; run("  + 2x2 footprints in an 8x8 level", buf, p); }
    return 0;
}
