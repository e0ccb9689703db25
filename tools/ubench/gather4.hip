// From the bare row stream to the forward kernel, one feature at a time (compile-time switches, so a
// feature that is off costs nothing): which part of msda_fwd_vec's 175 us is NOT its 4.29 GB of rows?
// Launch shape, slab layout and row pattern are the forward's at the north-star shape (see gather3.hip):
// one 256-lane workgroup per 16 queries of a (b, h), 16 taps x 4 rows of 256 B per query, rows = the 2x2
// footprints of random pixels in the 64^2 / 32^2 / 16^2 / 8^2 pyramid of an L2-resident slab, 4 + 4 reads
// in flight per lane through a buffer descriptor.
//   F_LEVELS   a level-table fetch (3 global loads by 4 lanes) + LDS write + barrier when the workgroup starts
//   F_STAGE    every lane loads its sample's (x, y) word and weight from streamed tensors, does the
//              bilinear arithmetic (~50 vector instructions) and writes a 32-byte record to LDS; barrier
//   F_RECORDS  the gather loop reads its row offsets and weights from those LDS records (2 x ds_read_b128 per tap)
//   F_FMA      the multiply-adds of the forward (unpack 8 bf16 channels, 8 FMAs per row)
//   F_STORE    the output row (16 bytes per lane)
//   F_BORDER_* the pixel is drawn from [-1, W-1] per axis like a uniform location (2/W of the samples have corners
//              outside the map): _OOB the outside corners get an offset past the descriptor (hardware zero, what the
//              forward does), _CLAMP they re-read the sample's inside neighbour, _MASK their lanes are switched off
// Build: hipcc --offload-arch=gfx950 -O3 gather4.hip -o gather4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

enum { F_LEVELS = 1, F_STAGE = 2, F_RECORDS = 4, F_FMA = 8, F_STORE = 16, F_BORDER_OOB = 32, F_BORDER_CLAMP = 64, F_BORDER_MASK = 128 };
constexpr int S = 5440, H = 8, B = 8, QT = 256, TAPS = 16;
constexpr unsigned ROW = H * 256;

__device__ __forceinline__ unsigned mix(unsigned x)
{
    x *= 0x9E3779B1u; x ^= x >> 15; x *= 0x85EBCA77u; x ^= x >> 13;
    return x;
}
template <int F>
__device__ __forceinline__ void footprint(unsigned x, int k, unsigned (&r)[4])
{
    const int lvl = (k >> 2) & 3;
    const unsigned W = 64u >> lvl;
    const unsigned st = lvl == 0 ? 0u : lvl == 1 ? 4096u : lvl == 2 ? 5120u : 5376u;
    if (F & (F_BORDER_OOB | F_BORDER_CLAMP | F_BORDER_MASK)) {
        // x0, y0 uniform in [-1, W-1]: W + 1 positions, the first and the last have a column / row outside
        const int x0 = (int)(((x & 0xffffu) * (W + 1)) >> 16) - 1, y0 = (int)(((x >> 16) * (W + 1)) >> 16) - 1;
        for (int c = 0; c < 4; ++c) {
            int xx = x0 + (c & 1), yy = y0 + (c >> 1);
            const bool ok = xx >= 0 && xx < (int)W && yy >= 0 && yy < (int)W;
            if (F & F_BORDER_CLAMP) { xx = min(max(xx, 0), (int)W - 1); yy = min(max(yy, 0), (int)W - 1); }
            r[c] = (ok || (F & F_BORDER_CLAMP)) ? (st + (unsigned)yy * W + (unsigned)xx) * ROW : 0x80000000u;
        }
        return;
    }
    const unsigned px = ((x & 0xffffu) * (W - 1)) >> 16, py = ((x >> 16) * (W - 1)) >> 16;
    const unsigned r0 = st + py * W + px;
    r[0] = r0 * ROW; r[1] = (r0 + 1) * ROW; r[2] = (r0 + W) * ROW; r[3] = (r0 + W + 1) * ROW;
}

template <int F>
__global__ void __launch_bounds__(256)
fwd_steps(const char *__restrict__ value, const unsigned *__restrict__ locw, const unsigned short *__restrict__ attn,
          const long long *__restrict__ shapes, char *__restrict__ out, unsigned salt)
{
    __shared__ uint4 recs[16 * 33];
    __shared__ int lvl[16];
    const int tid = threadIdx.x, lig = tid % 16, qi = tid / 16;
    const int bid = blockIdx.x;
    const int h = bid % H, t = bid / H, qt = t % QT, b = t / QT;
    const char *slab = value + ((size_t)b * S * H + h) * 256;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)slab, (short)0, (int)((size_t)S * ROW - h * 256), 0x00020000);
    const unsigned lane_off = lig * 16;
    const unsigned q = (unsigned)(qt * 16 + qi) + (unsigned)b * 4096u;

    if (F & F_LEVELS) {
        if (tid < 4) { lvl[3 * tid] = (int)shapes[2 * tid]; lvl[3 * tid + 1] = (int)shapes[2 * tid + 1]; lvl[3 * tid + 2] = (int)shapes[8 + tid]; }
        __syncthreads();
    }
    if (F & F_STAGE) {
        // one sample per lane: (rq, kk) = (tid / 16, tid % 16); streamed inputs, the forward's arithmetic
        const int rq = tid / 16, kk = tid % 16;
        const size_t s = (((size_t)b * 4096 + qt * 16 + rq) * H + h) * 16 + kk;
        const unsigned w = locw[s];
        const float a = (float)attn[s] * (1.f / 65536.f);
        const float lx = __uint_as_float((w & 0xffffu) << 16), ly = __uint_as_float(w & 0xffff0000u);
        const int l = kk >> 2;
        const int Hl = (F & F_LEVELS) ? lvl[3 * l] : 64 >> l, Wl = (F & F_LEVELS) ? lvl[3 * l + 1] : 64 >> l;
        const float y = ly * (float)Hl - 0.5f, x = lx * (float)Wl - 0.5f;
        const bool inside = (y > -1.f) && (x > -1.f) && (y < (float)Hl) && (x < (float)Wl);
        const float yf = floorf(y), xf = floorf(x);
        const float fy = inside ? y - yf : 0.f, fx = inside ? x - xf : 0.f;
        unsigned r[4];
        footprint<F>(mix((q + rq - qi) * 64u + (unsigned)kk + salt), kk, r);       // (the rows stay the hash's: same stream as the other variants)
        const float gy = 1.f - fy, gx = 1.f - fx;
        recs[rq * 33 + 2 * kk] = make_uint4(r[0], r[1], r[2], r[3]);
        recs[rq * 33 + 2 * kk + 1] = make_uint4(__float_as_uint(gy * gx * a), __float_as_uint(gy * fx * a),
                                                 __float_as_uint(fy * gx * a), __float_as_uint(fy * fx * a));
        __syncthreads();
    }

    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    u32x4 rawA[4], rawB[4];
    uint4 wA, wB;
    auto issue = [&](int k, u32x4 (&raw)[4], uint4 &ww) {
        unsigned r[4];
        if (F & F_RECORDS) {
            const uint4 rr = recs[qi * 33 + 2 * k];
            ww = recs[qi * 33 + 2 * k + 1];
            r[0] = rr.x; r[1] = rr.y; r[2] = rr.z; r[3] = rr.w;
        } else {
            const unsigned x = mix(q * 64u + (unsigned)k + salt);
            footprint<F>(x, k, r);
            ww = make_uint4(x, x >> 3, x >> 5, x >> 7);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (F & F_BORDER_MASK) {
                raw[c] = u32x4{0u, 0u, 0u, 0u};
                if (r[c] != 0x80000000u) raw[c] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(r[c] + lane_off), 0, 0);
            } else {
                raw[c] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(r[c] + lane_off), 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto consume = [&](const u32x4 (&raw)[4], const uint4 &ww) {
        const float w4[4] = {__uint_as_float(ww.x), __uint_as_float(ww.y), __uint_as_float(ww.z), __uint_as_float(ww.w)};
        if (F & F_FMA) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[2 * i] = fmaf(w4[c], __uint_as_float(raw[c][i] << 16), acc[2 * i]);
                    acc[2 * i + 1] = fmaf(w4[c], __uint_as_float(raw[c][i] & 0xffff0000u), acc[2 * i + 1]);
                }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] += w4[c] + __uint_as_float(raw[c][0] ^ raw[c][1] ^ raw[c][2] ^ raw[c][3]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(acc[i]));
        __builtin_amdgcn_sched_barrier(0);
    };
    issue(0, rawA, wA);
    int k = 0;
    for (; k + 2 < TAPS; k += 2) {
        issue(k + 1, rawB, wB);
        consume(rawA, wA);
        issue(k + 2, rawA, wA);
        consume(rawB, wB);
    }
    issue(k + 1, rawB, wB);
    consume(rawA, wA);
    consume(rawB, wB);
    if (F & F_STORE) {
        uint4 o = make_uint4(__float_as_uint(acc[0] + acc[1]), __float_as_uint(acc[2] + acc[3]),
                             __float_as_uint(acc[4] + acc[5]), __float_as_uint(acc[6] + acc[7]));
        *(uint4 *)(out + ((((size_t)b * 4096 + qt * 16 + qi) * H + h) * 16 + lig) * 16) = o;
    } else if (acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + acc[6] + acc[7] == 123.456f) {
        out[tid] = 1;
    }
}

static char *g_value, *g_out;
static unsigned *g_loc;
static unsigned short *g_attn;
static long long *g_shapes;

template <int F>
static void run(const char *name)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = B * QT * H;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(fwd_steps<F>, dim3(grid), dim3(256), 0, 0, g_value, g_loc, g_attn, g_shapes, g_out, 17u + w);
    CK(hipEventRecord(e0));
    const int n = 20;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(fwd_steps<F>, dim3(grid), dim3(256), 0, 0, g_value, g_loc, g_attn, g_shapes, g_out, 100u + i);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= n;
    printf("%-84s %7.1f us  %6.2f TB/s of rows\n", name, ms * 1e3, (double)grid * 256 * TAPS * 4 * 16 / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

int main()
{
    const size_t samples = (size_t)B * 4096 * H * 16;
    CK(hipMalloc(&g_value, (size_t)B * S * ROW)); CK(hipMemset(g_value, 1, (size_t)B * S * ROW));
    CK(hipMalloc(&g_out, (size_t)B * 4096 * H * 256));
    CK(hipMalloc(&g_loc, samples * 4)); CK(hipMemset(g_loc, 0x3e, samples * 4));
    CK(hipMalloc(&g_attn, samples * 2)); CK(hipMemset(g_attn, 0x11, samples * 2));
    long long hs[12] = {64, 64, 32, 32, 16, 16, 8, 8, 0, 4096, 5120, 5376};
    CK(hipMalloc(&g_shapes, sizeof hs)); CK(hipMemcpy(g_shapes, hs, sizeof hs, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; ++rep) {
        run<F_FMA | F_STORE>("interior footprints, FMA, store");
        run<F_FMA | F_STORE | F_BORDER_OOB>("  footprints over the borders like uniform locations: outside corners = offset past the descriptor");
        run<F_FMA | F_STORE | F_BORDER_CLAMP>("  outside corners re-read the inside neighbour (L1 hit)");
        run<F_FMA | F_STORE | F_BORDER_MASK>("  outside corners: lanes switched off");
        run<0>("rows only (pyramid footprints, buffer loads, 4 + 4 in flight)");
        run<F_STORE>("+ output row store");
        run<F_FMA | F_STORE>("+ unpack + FMA");
        run<F_LEVELS | F_FMA | F_STORE>("+ level table fetch + barrier at workgroup start");
        run<F_STAGE | F_FMA | F_STORE>("+ staging (sample loads, arithmetic, LDS records, barrier), rows still from registers");
        run<F_STAGE | F_RECORDS | F_FMA | F_STORE>("+ row offsets and weights read from the LDS records");
        run<F_LEVELS | F_STAGE | F_RECORDS | F_FMA | F_STORE>("+ everything (= the forward's structure)");
        run<F_LEVELS | F_STAGE | F_RECORDS | F_STORE>("everything but the FMAs");
        run<F_RECORDS | F_STAGE>("staging + records only");
    }
    return 0;
}
