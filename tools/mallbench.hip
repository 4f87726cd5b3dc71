// Does the Infinity Cache (256 MiB) carry a write -> read hand-off between two kernels?
// For a buffer of S bytes: write it all (kernel A), read it all (kernel B), repeat; report both rates.
//   hipcc --offload-arch=gfx950 -O3 tools/mallbench.hip -o tools/bin/mallbench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NT_STORE>
__global__ void __launch_bounds__(512) kw(f4* __restrict__ dst, size_t n, float v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) {
        f4 x = {v, 2.f, 3.f, (float)i};
        if (NT_STORE) __builtin_nontemporal_store(x, dst + i); else dst[i] = x;
    }
}
__global__ void __launch_bounds__(512) kr(const f4* __restrict__ src, size_t n, float* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    f4 acc = {0, 0, 0, 0};
    for (; i + 3 * st < n; i += 4 * st) acc += src[i] + src[i + st] + src[i + 2 * st] + src[i + 3 * st];
    for (; i < n; i += st) acc += src[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}
#define CK(x) do { hipError_t err__ = (x); if (err__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err__)); return 1; } } while (0)
int main() {
    f4* buf; float* sink; f4* other;
    const size_t cap = (size_t)3 << 30;
    CK(hipMalloc(&buf, cap)); CK(hipMalloc(&other, (size_t)1 << 30)); CK(hipMalloc(&sink, 4));
    hipEvent_t e[4]; for (auto& x : e) CK(hipEventCreate(&x));
    for (int nt = 0; nt < 2; ++nt)
    for (size_t mb : {16, 32, 64, 96, 128, 192, 256, 384, 1024, 3072}) {
        const size_t n = (mb << 20) / 16;
        const int reps = mb >= 1024 ? 6 : 40;
        double tw = 0, tr = 0;
        for (int r = 0; r < reps + 2; ++r) {
            CK(hipEventRecord(e[0], 0));
            if (nt) hipLaunchKernelGGL(kw<1>, dim3(1024), dim3(512), 0, 0, buf, n, (float)r);
            else    hipLaunchKernelGGL(kw<0>, dim3(1024), dim3(512), 0, 0, buf, n, (float)r);
            CK(hipEventRecord(e[1], 0));
            hipLaunchKernelGGL(kr, dim3(1024), dim3(512), 0, 0, buf, n, sink);
            CK(hipEventRecord(e[2], 0));
            CK(hipEventSynchronize(e[2]));
            float a, b; CK(hipEventElapsedTime(&a, e[0], e[1])); CK(hipEventElapsedTime(&b, e[1], e[2]));
            if (r >= 2) { tw += a; tr += b; }
        }
        printf("%s stores, buffer %5zu MB: write %6.2f TB/s   read-back %6.2f TB/s\n", nt ? "nt   " : "plain", mb,
               (double)(mb << 20) * reps / (tw * 1e-3) / 1e12, (double)(mb << 20) * reps / (tr * 1e-3) / 1e12);
    }
    return 0;
}
