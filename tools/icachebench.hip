// What does a kernel's FIRST wave on a CU pay for instructions that are not in the instruction cache?
// (developer probe, round 5).  The FFT kernels of this library are 19 - 37 KB of straight-line code each; a small search
// launches six of them in turn, each ONCE per orientation batch, so every launch starts with a cold instruction cache
// (64 KB shared by two CUs) and - on the first step - a cold L2.
//   k_code<KB>: KB kilobytes of s_nop (4 bytes, one issue cycle each) between two clock reads, one wave per CU.
//   run (a) twice in a row (warm), (b) after a different kernel of 128 KB (instruction cache cold, L2 warm),
//   (c) after that and a 1 GB fill (L2 and Infinity Cache cold as well).
//   hipcc --offload-arch=gfx950 -O3 tools/icachebench.hip -o tools/bin/icachebench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define NOPS_1K  asm volatile(".rept 256\n s_nop 0\n .endr");
template <int KB, int TAG>
__global__ void __launch_bounds__(64) k_code(long long* cyc, int n) {
    long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < KB; ++i) NOPS_1K
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && (int)blockIdx.x < n) cyc[blockIdx.x] = t1 - t0;
}

__global__ void k_fill(float4* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

template <int KB>
static void run(int wgs, long long* dcyc, float4* big, size_t nbig) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<long long> h(wgs);
    auto timed = [&](const char* what, int mode) {
        float best = 1e9f, sum = 0.f; long long cmax = 0, cmed = 0;
        const int reps = 7;
        for (int r = 0; r < reps; ++r) {
            if (mode >= 1) hipLaunchKernelGGL((k_code<128, 1>), dim3(1024), dim3(64), 0, 0, dcyc + 4096, 0);
            if (mode >= 2) hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, big, nbig);
            if (mode == 0) hipLaunchKernelGGL((k_code<KB, 0>), dim3(wgs), dim3(64), 0, 0, dcyc, wgs);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((k_code<KB, 0>), dim3(wgs), dim3(64), 0, 0, dcyc, wgs);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms); sum += ms;
            CK(hipMemcpy(h.data(), dcyc, wgs * sizeof(long long), hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            cmax = h[wgs - 1]; cmed = h[wgs / 2];
        }
        // s_memtime counts at 100 MHz: 10 ns per tick
        printf("  %3d KB x %4d wg  %-34s event %7.1f us (min %7.1f)   in-kernel median %7.2f us  max %7.2f us\n", KB, wgs, what,
               1e3f * sum / reps, 1e3f * best, cmed * 0.01, cmax * 0.01);
    };
    timed("after itself (warm)", 0);
    timed("after 128 KB of other code", 1);
    timed("after other code + 1 GB fill", 2);
}

int main() {
    long long* dcyc; CK(hipMalloc(&dcyc, 8192 * sizeof(long long)));
    const size_t nbig = (size_t)1 << 26;                 // 1 GB of float4
    float4* big; CK(hipMalloc(&big, nbig * sizeof(float4)));
    hipLaunchKernelGGL((k_code<1, 0>), dim3(256), dim3(64), 0, 0, dcyc, 256);
    CK(hipDeviceSynchronize());
    for (int wgs : {256, 2048}) {
        run<1>(wgs, dcyc, big, nbig);
        run<8>(wgs, dcyc, big, nbig);
        run<16>(wgs, dcyc, big, nbig);
        run<32>(wgs, dcyc, big, nbig);
        run<64>(wgs, dcyc, big, nbig);
    }
    return 0;
}
