// radixbench: what a radix-15 last stage would cost against the radix-8 one the 2048-point transforms end in (round 6,
// verdict item 5: T = 3840 = 15 * 16 * 16 tiles would pad the 10000 x 10000 DEM to 1.33 x instead of 1.51 x its cells).
// Per-point issue cost of the three butterflies on the packed float32 pipe, data in registers, as the FFT kernels of
// sc_fft.hip hold it: every thread keeps N complex values, applies the butterfly and a twiddle per value (what a Stockham
// stage does between butterflies) ITER times.  One wave per SIMD x 4 waves per CU, 256 CUs: issue-bound, no memory.
//   hipcc -O3 --offload-arch=gfx950 tools/radixbench.hip -o /tmp/radixbench && /tmp/radixbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef float v2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2 cmul(v2 a, v2 w) { return v2{a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x}; }
__device__ __forceinline__ v2 muli(v2 a) { return v2{-a.y, a.x}; }           // i a
__device__ __forceinline__ v2 mulmi(v2 a) { return v2{a.y, -a.x}; }          // -i a

template <int N> struct Dft;
template <> struct Dft<2> {
    static __device__ __forceinline__ void run(v2* x) { v2 a = x[0] + x[1], b = x[0] - x[1]; x[0] = a; x[1] = b; }
};
template <> struct Dft<4> {
    static __device__ __forceinline__ void run(v2* x) {
        v2 a = x[0] + x[2], b = x[0] - x[2], c = x[1] + x[3], d = mulmi(x[1] - x[3]);
        x[0] = a + c; x[2] = a - c; x[1] = b + d; x[3] = b - d;
    }
};
template <> struct Dft<8> {
    static __device__ __forceinline__ void run(v2* x) {
        v2 e[4] = {x[0], x[2], x[4], x[6]}, o[4] = {x[1], x[3], x[5], x[7]};
        Dft<4>::run(e); Dft<4>::run(o);
        const float h = 0.70710678118654752f;
        o[1] = v2{(o[1].x + o[1].y) * h, (o[1].y - o[1].x) * h};
        o[2] = mulmi(o[2]);
        o[3] = v2{(o[3].y - o[3].x) * h, -(o[3].x + o[3].y) * h};
        for (int k = 0; k < 4; ++k) { x[k] = e[k] + o[k]; x[k + 4] = e[k] - o[k]; }
    }
};
template <> struct Dft<16> {
    static __device__ __forceinline__ void run(v2* x) {
        v2 e[8], o[8];
        for (int k = 0; k < 8; ++k) { e[k] = x[2 * k]; o[k] = x[2 * k + 1]; }
        Dft<8>::run(e); Dft<8>::run(o);
        const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
        const v2 w[8] = {{1.f, 0.f}, {c1, -s1}, {h, -h}, {s1, -c1}, {0.f, -1.f}, {-s1, -c1}, {-h, -h}, {-c1, -s1}};
        for (int k = 0; k < 8; ++k) {
            const v2 t = (k == 0) ? o[0] : (k == 4) ? mulmi(o[4]) : cmul(o[k], w[k]);
            x[k] = e[k] + t; x[k + 8] = e[k] - t;
        }
    }
};
template <> struct Dft<3> {
    static __device__ __forceinline__ void run(v2* x) {
        const float s = 0.86602540378443865f;
        v2 t = x[1] + x[2], m = x[0] - 0.5f * t, d = mulmi((x[1] - x[2]) * s);
        x[0] = x[0] + t; x[1] = m + d; x[2] = m - d;
    }
};
template <> struct Dft<5> {
    static __device__ __forceinline__ void run(v2* x) {
        const float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f, s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;
        v2 a1 = x[1] + x[4], a2 = x[2] + x[3], b1 = x[1] - x[4], b2 = x[2] - x[3];
        v2 m1 = x[0] + c1 * a1 + c2 * a2, m2 = x[0] + c2 * a1 + c1 * a2;
        v2 d1 = mulmi(s1 * b1 + s2 * b2), d2 = mulmi(s2 * b1 - s1 * b2);
        x[0] = x[0] + a1 + a2; x[1] = m1 + d1; x[4] = m1 - d1; x[2] = m2 + d2; x[3] = m2 - d2;
    }
};
// 15 = 3 x 5, prime factor map: input n = (5 n1 + 3 n2) mod 15, output k = (10 k1 + 6 k2) mod 15; no twiddles between
template <> struct Dft<15> {
    static __device__ __forceinline__ void run(v2* x) {
        v2 y[15];
        for (int n2 = 0; n2 < 5; ++n2) {
            v2 t[3] = {x[(3 * n2) % 15], x[(5 + 3 * n2) % 15], x[(10 + 3 * n2) % 15]};
            Dft<3>::run(t);
            for (int k1 = 0; k1 < 3; ++k1) y[k1 * 5 + n2] = t[k1];
        }
        for (int k1 = 0; k1 < 3; ++k1) {
            Dft<5>::run(y + 5 * k1);
            for (int k2 = 0; k2 < 5; ++k2) x[(10 * k1 + 6 * k2) % 15] = y[5 * k1 + k2];
        }
    }
};

template <int N, int SETS>
__global__ void __launch_bounds__(256) k_bench(float2* out, int iters, float eps) {
    v2 x[SETS][N];
    for (int s = 0; s < SETS; ++s)
        for (int k = 0; k < N; ++k) x[s][k] = v2{(float)(threadIdx.x + k + s) * 1e-3f, (float)(k - s) * 1e-3f};
    const v2 w = v2{1.f - eps, eps};                                          // a "twiddle" (keeps the values bounded)
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int s = 0; s < SETS; ++s) {
            Dft<N>::run(x[s]);
#pragma unroll
            for (int k = 1; k < N; ++k) x[s][k] = cmul(x[s][k], w);           // the stage's twiddles, one per value
        }
    v2 acc = v2{0.f, 0.f};
    for (int s = 0; s < SETS; ++s)
        for (int k = 0; k < N; ++k) acc += x[s][k];
    out[blockIdx.x * 256 + threadIdx.x] = make_float2(acc.x, acc.y);
}

template <int N, int SETS>
static double run(const char* name, float2* d, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 4;                                               // four workgroups of 4 waves per CU: 4 waves per SIMD
    hipLaunchKernelGGL((k_bench<N, SETS>), dim3(blocks), dim3(256), 0, 0, d, 8, 1e-3f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k_bench<N, SETS>), dim3(blocks), dim3(256), 0, 0, d, iters, 1e-3f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    const double points = (double)blocks * 256 * SETS * N * iters;
    const double ns_per_kpoint = 1e6 * ms / points * 1e3;
    printf("%-28s %8.3f ms  %8.4f ns per 1000 points (whole chip)\n", name, ms, ns_per_kpoint);
    return ns_per_kpoint;
}

int main() {
    float2* d;
    hipMalloc(&d, sizeof(float2) * 256 * 4 * 256);
    const int iters = 20000;
    // points per thread as in the kernels: 32 (two sets of 16, four of 8) / 30 (two sets of 15)
    for (int rep = 0; rep < 2; ++rep) {
        const double r16 = run<16, 2>("radix 16 + twiddles (2 sets)", d, iters);
        const double r8 = run<8, 4>("radix 8 + twiddles (4 sets)", d, iters);
        const double r15 = run<15, 2>("radix 15 + twiddles (2 sets)", d, iters);
        const double p2048 = 2 * r16 + r8, p3840 = 2 * r16 + r15;
        printf("  per point, 16 x 16 x 8 (2048): %.4f   16 x 16 x 15 (3840): %.4f   ratio %.3f;  x cells 1.327 / 1.510 = %.3f of the transform work\n",
               p2048, p3840, p3840 / p2048, p3840 / p2048 * 1.327 / 1.510);
    }
    return 0;
}
