// mfmabench: can the idle FP32 matrix pipe take a radix-16 stage of the inverse column pass?
// (round-4 verdict, item 1a - the go / no-go numbers of profiles/r05_mfma_dft.txt)
//
// A twiddle-free 16-point complex DFT of 16 columns is Y = F X with F the 16 x 16 DFT matrix:
//     Yr = Fr Xr - Fi Xi,   Yi = Fi Xr + Fr Xi
// four real 16 x 16 x 16 products = 16 v_mfma_f32_16x16x4_f32 (K = 4 per instruction), exact f32
// (an fmaf chain per output).  With X as the B operand the input of a lane is (column n = lane & 15,
// points k = (lane >> 4) + 4 s in register s) and its output (column n, points m = 4 (lane >> 4) + i):
// the instruction moves the point index between register and lane-quad as a side effect.
//
// Legs (every leg: ONE workgroup per CU - 96 KB of dynamic LDS each pins that -, `iters` rounds; device time by HIP
// events, cycles by s_memtime, the clock held by s_memtime / s_memrealtime (100 MHz)):
//   check     the DFT through the matrix pipe against a double-precision DFT on the host (layout + error)
//   mfma      one wave per SIMD (256 threads) and two (512): 8 independent DFT groups per round
//             (= stage 1 of one 2048-point column, one plane): 128 MFMAs per round and wave
//   valu      a stream of independent v_pk_fma_f32 (what the butterflies are made of), same shapes
//   sibling   512 threads: waves 0 - 3 run the mfma leg, waves 4 - 7 the valu leg - one of each per SIMD.
//             Co-issue shows as both finishing in about their stand-alone time
//   mixed     ONE wave interleaves an MFMA with q packed FMAs, q = 2, 4, 6, 8 (the in-order issue of a
//             single wave: what a column wave that keeps its butterflies would see)
//
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfmabench.hip -o tools/bin/mfmabench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// F = exp(-2 pi i m k / 16) * scale; lane l holds A[m = l & 15][k = (l >> 4) + 4 s] for slice s
__device__ __forceinline__ void dft_consts(float (&fr)[4], float (&fi)[4], float scale) {
    const int l = threadIdx.x & 63, m = l & 15, q = l >> 4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int k = q + 4 * s, t = (m * k) & 15;
        float sn, cs;
        sincospif(-(float)t / 8.0f, &sn, &cs);
        fr[s] = cs * scale;
        fi[s] = sn * scale;
    }
}

// one DFT group: 16 columns x 16 points, X in (xr[s], xi[s]) as B operands; 16 MFMAs
__device__ __forceinline__ void dft_group(const float (&fr)[4], const float (&fi)[4], const float (&nfi)[4],
                                          const float (&xr)[4], const float (&xi)[4], f32x4& yr, f32x4& yi) {
    yr = (f32x4){0.f, 0.f, 0.f, 0.f};
    yi = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        yr = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[s], xr[s], yr, 0, 0, 0);
        yi = __builtin_amdgcn_mfma_f32_16x16x4f32(fi[s], xr[s], yi, 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        yr = __builtin_amdgcn_mfma_f32_16x16x4f32(nfi[s], xi[s], yr, 0, 0, 0);
        yi = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[s], xi[s], yi, 0, 0, 0);
    }
}

// ---- correctness: one wave, one group -------------------------------------------------------
__global__ void k_check(const float2* __restrict__ x /* [16 points][16 columns] */, float2* __restrict__ y) {
    float fr[4], fi[4], nfi[4], xr[4], xi[4];
    dft_consts(fr, fi, 1.0f);
    const int l = threadIdx.x, n = l & 15, q = l >> 4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        nfi[s] = -fi[s];
        const float2 v = x[(q + 4 * s) * 16 + n];          // B[k = q + 4 s][n]
        xr[s] = v.x; xi[s] = v.y;
    }
    f32x4 yr, yi;
    dft_group(fr, fi, nfi, xr, xi, yr, yi);
#pragma unroll
    for (int i = 0; i < 4; ++i) y[(4 * q + i) * 16 + n] = make_float2(yr[i], yi[i]);   // D[m = 4 q + i][n]
}

// ---- timing legs -------------------------------------------------------------------------------
constexpr int GROUPS = 8;       // DFT groups per round and wave: 128 MFMAs = stage 1 of a 2048-point column, one plane

__device__ __forceinline__ void mfma_rounds(int iters, float seed, float* sink, long long* cyc) {
    float fr[4], fi[4], nfi[4];
    dft_consts(fr, fi, 0.25f);                              // |F| = 1/4: the values stay bounded round after round
#pragma unroll
    for (int s = 0; s < 4; ++s) nfi[s] = -fi[s];
    float xr[GROUPS][4], xi[GROUPS][4];
#pragma unroll
    for (int g = 0; g < GROUPS; ++g)
#pragma unroll
        for (int s = 0; s < 4; ++s) { xr[g][s] = seed + 0.01f * (g + s); xi[g][s] = seed - 0.02f * (g - s); }
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            f32x4 yr, yi;
            dft_group(fr, fi, nfi, xr[g], xi[g], yr, yi);
#pragma unroll
            for (int s = 0; s < 4; ++s) { xr[g][s] = yr[s]; xi[g][s] = yi[s]; }   // the next round's input
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0.f;
#pragma unroll
    for (int g = 0; g < GROUPS; ++g)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc += xr[g][s] + xi[g][s];
    if (acc == 12345.678f) *sink = acc;
    if ((threadIdx.x & 63) == 0) cyc[0] = t1 - t0;
}

constexpr int PKS = 128;        // packed FMAs per round and wave in the valu leg

__device__ __forceinline__ void valu_rounds(int iters, float seed, float* sink, long long* cyc) {
    f32x2 r[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) r[j] = (f32x2){seed + j, seed - j};
    const f32x2 a = {0.999f, 1.001f}, b = {1e-3f, -1e-3f};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < PKS / 32; ++u)
#pragma unroll
            for (int j = 0; j < 32; ++j)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r[j]) : "v"(r[j]), "v"(a), "v"(b));
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    f32x2 acc = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 32; ++j) acc += r[j];
    if (acc.x + acc.y == 12345.678f) *sink = acc.x;
    if ((threadIdx.x & 63) == 0) cyc[0] = t1 - t0;
}

extern __shared__ float pin_lds[];          // 96 KB per workgroup: one workgroup per CU, so "waves per SIMD" is what it says
__global__ void __launch_bounds__(512) k_mfma(int iters, float seed, float* sink, long long* cyc, long long* real) {
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    mfma_rounds(iters, seed, sink, cyc + (size_t)blockIdx.x * 8 + (threadIdx.x >> 6));
    if (threadIdx.x == 0) real[blockIdx.x] = __builtin_amdgcn_s_memrealtime() - r0;
    if (seed == 12345.f) pin_lds[threadIdx.x] = seed;
}
__global__ void __launch_bounds__(512) k_valu(int iters, float seed, float* sink, long long* cyc, long long* real) {
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    valu_rounds(iters, seed, sink, cyc + (size_t)blockIdx.x * 8 + (threadIdx.x >> 6));
    if (threadIdx.x == 0) real[blockIdx.x] = __builtin_amdgcn_s_memrealtime() - r0;
    if (seed == 12345.f) pin_lds[threadIdx.x] = seed;
}
// waves 0 - 3: matrix pipe, waves 4 - 7: vector pipe (one of each per SIMD)
__global__ void __launch_bounds__(512) k_sibling(int it_m, int it_v, float seed, float* sink, long long* cyc, long long* real) {
    long long* c = cyc + (size_t)blockIdx.x * 8 + (threadIdx.x >> 6);
    if ((threadIdx.x >> 6) < 4) mfma_rounds(it_m, seed, sink, c);
    else valu_rounds(it_v, seed, sink, c);
    if (seed == 12345.f) pin_lds[threadIdx.x] = seed;
}

// one wave: an MFMA, then Q independent packed FMAs, 128 times per round
template <int Q>
__global__ void __launch_bounds__(512) k_mixed(int iters, float seed, float* sink, long long* cyc, long long* real) {
    if (seed == 12345.f) pin_lds[threadIdx.x] = seed;
    float fr[4], fi[4], nfi[4];
    dft_consts(fr, fi, 0.25f);
#pragma unroll
    for (int s = 0; s < 4; ++s) nfi[s] = -fi[s];
    float xr[GROUPS][4], xi[GROUPS][4];
#pragma unroll
    for (int g = 0; g < GROUPS; ++g)
#pragma unroll
        for (int s = 0; s < 4; ++s) { xr[g][s] = seed + 0.01f * (g + s); xi[g][s] = seed - 0.02f * (g - s); }
    f32x2 r[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) r[j] = (f32x2){seed + j, seed - j};
    const f32x2 a = {0.999f, 1.001f}, b = {1e-3f, -1e-3f};
    auto fill = [&](int at) {
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            const int k = (at * Q + j) & 15;
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r[k]) : "v"(r[k]), "v"(a), "v"(b));
        }
    };
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            f32x4 yr = {0.f, 0.f, 0.f, 0.f}, yi = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                yr = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[s], xr[g][s], yr, 0, 0, 0); fill(4 * s);
                yi = __builtin_amdgcn_mfma_f32_16x16x4f32(fi[s], xr[g][s], yi, 0, 0, 0); fill(4 * s + 1);
                yr = __builtin_amdgcn_mfma_f32_16x16x4f32(nfi[s], xi[g][s], yr, 0, 0, 0); fill(4 * s + 2);
                yi = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[s], xi[g][s], yi, 0, 0, 0); fill(4 * s + 3);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) { xr[g][s] = yr[s]; xi[g][s] = yi[s]; }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0.f;
#pragma unroll
    for (int g = 0; g < GROUPS; ++g)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc += xr[g][s] + xi[g][s];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc += r[j].x + r[j].y;
    if (acc == 12345.678f) *sink = acc;
    if ((threadIdx.x & 63) == 0) cyc[(size_t)blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

// ---- host --------------------------------------------------------------------------------------
struct Timing { double ms; double cyc_a, cyc_b; double ghz; };
constexpr size_t PIN_LDS = 96 * 1024;

template <class K, class... A>
static Timing run(K kernel, int blocks, int threads, long long* d_cyc, long long* d_real, int split, A... args) {
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PIN_LDS));
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
    HIPCHECK(hipMemset(d_real, 0, sizeof(long long) * blocks));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), PIN_LDS, 0, args..., d_cyc, d_real);      // warm-up
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), PIN_LDS, 0, args..., d_cyc, d_real);
    HIPCHECK(hipEventRecord(e1));
    HIPCHECK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    const int waves = threads / 64;
    std::vector<long long> h((size_t)blocks * 8), hr(blocks);
    HIPCHECK(hipMemcpy(h.data(), d_cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(hr.data(), d_real, sizeof(long long) * blocks, hipMemcpyDeviceToHost));
    double a = 0, b = 0, ratio = 0; long na = 0, nb = 0, nr = 0;
    for (int bl = 0; bl < blocks; ++bl) {
        for (int w = 0; w < waves; ++w) {
            if (split && w >= split) { b += (double)h[(size_t)bl * 8 + w]; ++nb; }
            else { a += (double)h[(size_t)bl * 8 + w]; ++na; }
        }
        if (hr[bl] > 0) { ratio += (double)h[(size_t)bl * 8] / (double)hr[bl]; ++nr; }     // shader ticks per 10 ns
    }
    return {ms, na ? a / na : 0, nb ? b / nb : 0, nr ? ratio / nr * 0.1 : 0.0};
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t p;
    HIPCHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("device %s, %d CUs, one workgroup per CU (96 KB of LDS each); %d rounds per launch; cycles = s_memtime ticks of the "
           "wave, GHz = s_memtime / s_memrealtime\n", p.gcnArchName, cus, iters);

    // ---- check
    {
        std::vector<float2> x(256), y(256);
        srand(7);
        for (auto& v : x) v = make_float2(rand() / (float)RAND_MAX - 0.5f, rand() / (float)RAND_MAX - 0.5f);
        float2 *dx, *dy;
        HIPCHECK(hipMalloc(&dx, sizeof(float2) * 256)); HIPCHECK(hipMalloc(&dy, sizeof(float2) * 256));
        HIPCHECK(hipMemcpy(dx, x.data(), sizeof(float2) * 256, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dx, dy);
        HIPCHECK(hipMemcpy(y.data(), dy, sizeof(float2) * 256, hipMemcpyDeviceToHost));
        double err = 0, mag = 0;
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) {
                double re = 0, im = 0;
                for (int k = 0; k < 16; ++k) {
                    const double a = -2.0 * M_PI * ((m * k) & 15) / 16.0;
                    re += x[k * 16 + n].x * cos(a) - x[k * 16 + n].y * sin(a);
                    im += x[k * 16 + n].x * sin(a) + x[k * 16 + n].y * cos(a);
                }
                err = fmax(err, fmax(fabs(re - y[m * 16 + n].x), fabs(im - y[m * 16 + n].y)));
                mag = fmax(mag, fmax(fabs(re), fabs(im)));
            }
        printf("check: DFT-16 of 16 columns through 16 v_mfma_f32_16x16x4_f32: max |error| %.3g of max |Y| %.3g (%.2g relative): %s\n",
               err, mag, err / mag, err / mag < 1e-6 ? "layout and arithmetic as described" : "WRONG");
    }

    float* sink;
    long long *d_cyc, *d_real;
    const int blocks = cus;
    HIPCHECK(hipMalloc(&sink, 4));
    HIPCHECK(hipMalloc(&d_cyc, sizeof(long long) * (size_t)blocks * 8));
    HIPCHECK(hipMalloc(&d_real, sizeof(long long) * (size_t)blocks));
    const double mf = 16.0 * GROUPS;                      // MFMAs per round and wave

    for (int waves : {4, 8}) {
        const int wps = waves / 4;
        Timing t = run(k_mfma, blocks, 64 * waves, d_cyc, d_real, 0, iters, 0.5f, sink);
        printf("mfma     %d wave(s) per SIMD: %8.3f ms at %.2f GHz: %6.1f cycles per MFMA and wave = %5.1f per MFMA on the SIMD's matrix pipe; "
               "a DFT group (16 MFMAs, 16 columns x 16 points) %6.0f cycles of the pipe\n", wps, t.ms, t.ghz,
               t.cyc_a / iters / mf, t.cyc_a / iters / mf / wps, 16 * t.cyc_a / iters / mf / wps);
        t = run(k_valu, blocks, 64 * waves, d_cyc, d_real, 0, iters, 0.5f, sink);
        printf("valu     %d wave(s) per SIMD: %8.3f ms at %.2f GHz: %6.2f cycles per v_pk_fma_f32 and wave = %5.2f per instruction on the SIMD\n",
               wps, t.ms, t.ghz, t.cyc_a / iters / PKS, t.cyc_a / iters / PKS / wps);
    }
    // sibling: one mfma wave and one valu wave per SIMD; `ratio` valu rounds (128 pk each) per mfma round (128 MFMAs)
    for (int ratio : {2, 4, 6, 8, 12}) {
        Timing t = run(k_sibling, blocks, 512, d_cyc, d_real, 4, iters, iters * ratio, 0.5f, sink);
        printf("sibling  mfma wave + valu wave per SIMD, %2d pk per MFMA offered: %8.3f ms; the mfma wave %5.1f cycles per MFMA, "
               "the valu wave %5.2f cycles per v_pk_fma_f32 (= %4.1f pk issued per MFMA slot while both run)\n", ratio, t.ms,
               t.cyc_a / iters / mf, t.cyc_b / (iters * (double)ratio) / PKS,
               (t.cyc_a / iters / mf) / (t.cyc_b / (iters * (double)ratio) / PKS));
    }
#define MIXED(Q) { Timing t = run(k_mixed<Q>, blocks, 256, d_cyc, d_real, 0, iters, 0.5f, sink); \
        Timing u = run(k_mixed<Q>, blocks, 512, d_cyc, d_real, 0, iters, 0.5f, sink); \
        printf("mixed    %d v_pk_fma_f32 after every MFMA in ONE instruction stream: one wave per SIMD %5.1f cycles per MFMA + fillers; " \
               "two waves per SIMD %5.1f per wave = %5.1f on the SIMD\n", Q, t.cyc_a / iters / mf, u.cyc_a / iters / mf, u.cyc_a / iters / mf / 2); }
    MIXED(0) MIXED(2) MIXED(4) MIXED(6) MIXED(8)
    return 0;
}
