// HBM ceilings of this chip for the access shapes of the inverse passes (developer probe):
// read-only, write-only (plain / non-temporal), copy, and I1's mix (0.37 read : 0.63 write).
//   hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/bin/membench && tools/bin/membench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0 read, 1 write plain, 2 write nt, 3 copy (nt store), 4 mix: 3 reads per 5 writes
__global__ void __launch_bounds__(512) k(const f4* __restrict__ src, f4* __restrict__ dst, size_t n, float* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    f4 acc = {0, 0, 0, 0};
    for (; i < n; i += st) {
        if (MODE == 0) acc += src[i];
        if (MODE == 1) dst[i] = f4{1.f, 2.f, 3.f, (float)i};
        if (MODE == 2) __builtin_nontemporal_store(f4{1.f, 2.f, 3.f, (float)i}, dst + i);
        if (MODE == 3) __builtin_nontemporal_store(src[i], dst + i);
        if (MODE == 4) { f4 v = ((i & 7) < 3) ? src[i] : f4{1.f, 2.f, 3.f, 4.f}; if ((i & 7) >= 3) __builtin_nontemporal_store(v, dst + i); else acc += v; }
    }
    if (MODE == 0 || MODE == 4) if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}

template <int MODE>
static void run(const char* name, const f4* s, f4* d, size_t n, float* sink, int blocks, double bytes_per_elem) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, s, d, n, sink);
    hipEventRecord(a, 0);
    const int reps = 20;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, s, d, n, sink);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-34s blocks %5d  %.3f ms/launch  %.2f TB/s\n", name, blocks, ms / reps, bytes_per_elem * n * reps / (ms * 1e-3) / 1e12);
}

int main() {
    const size_t n = (size_t)3 << 26;                 // 3 GiB per buffer (f4 elements: 201M)
    f4 *s, *d; float* sink;
    hipMalloc(&s, n * 16); hipMalloc(&d, n * 16); hipMalloc(&sink, 4);
    hipMemset(s, 1, n * 16); hipMemset(d, 0, n * 16);
    for (int blocks : {256, 512, 2048}) {
        run<0>("read only", s, d, n, sink, blocks, 16);
        run<1>("write only, plain stores", s, d, n, sink, blocks, 16);
        run<2>("write only, non-temporal stores", s, d, n, sink, blocks, 16);
        run<3>("copy (read + nt write)", s, d, n, sink, blocks, 32);
        run<4>("mix 3 reads : 5 writes", s, d, n, sink, blocks, 16);
    }
    return 0;
}
