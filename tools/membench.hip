// HBM ceilings of this chip (developer probe).  Round 3: the round-2 probe kept ONE 16-byte access
// in flight per lane on 256 - 2048 blocks and read 16 % below the guide's copy figure
// (MI355X_MICROARCH.md: 6.29 TB/s float4 copy, 6.0 - 6.2 TB/s plain stores).  Here every lane keeps
// U independent 16-byte accesses in flight, the grid is swept over, and the inverse passes' mixes
// are streams of whole 128-byte lines: I1 = 1 line read : 2 lines written, I2 = 38 read : 1 written.
//   hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/bin/membench && tools/bin/membench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));

// MODE 0 read, 1 write plain, 2 write nt, 3 copy (plain store), 4 copy (nt store),
//      5 I1 mix: per 3 f4 slots 1 read + 2 nt writes, 6 I2 mix: 38 reads per nt write
template <int MODE, int U>
__global__ void __launch_bounds__(256) k(const f4* __restrict__ src, f4* __restrict__ dst, size_t n, float* sink) {
    const size_t chunk = (size_t)blockDim.x * U;
    f4 acc = {0, 0, 0, 0};
    for (size_t c = blockIdx.x; c * chunk < n; c += gridDim.x) {
        const size_t i0 = c * chunk + threadIdx.x;
        f4 v[U];
        if (MODE == 0 || MODE == 3 || MODE == 4 || MODE == 5 || MODE == 6) {
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = src[i0 + (size_t)u * blockDim.x];
        }
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u];
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const f4 w = {1.f, 2.f, 3.f, (float)(i0 + u)};
                if (MODE == 1) dst[i0 + (size_t)u * blockDim.x] = w;
                else __builtin_nontemporal_store(w, dst + i0 + (size_t)u * blockDim.x);
            }
        }
        if (MODE == 3 || MODE == 4) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (MODE == 3) dst[i0 + (size_t)u * blockDim.x] = v[u];
                else __builtin_nontemporal_store(v[u], dst + i0 + (size_t)u * blockDim.x);
            }
        }
        if (MODE == 5) {          // read chunk c of src, write chunks 2c and 2c+1 of dst (2n elements)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                __builtin_nontemporal_store(v[u], dst + 2 * c * chunk + threadIdx.x + (size_t)u * blockDim.x);
                __builtin_nontemporal_store(v[u] + 1.f, dst + (2 * c + 1) * chunk + threadIdx.x + (size_t)u * blockDim.x);
            }
        }
        if (MODE == 6) {
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u];
            if (c % 38 == 0) __builtin_nontemporal_store(acc, dst + (c / 38) * blockDim.x + threadIdx.x);
        }
    }
    if (MODE == 0 || MODE == 6) if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}

template <int MODE, int U>
static void run(const char* name, const f4* s, f4* d, size_t n, float* sink, int blocks, double bytes_per_elem) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<MODE, U>), dim3(blocks), dim3(256), 0, 0, s, d, n, sink);
    hipEventRecord(a, 0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<MODE, U>), dim3(blocks), dim3(256), 0, 0, s, d, n, sink);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-30s U %d  blocks %6d  %.3f ms/launch  %.2f TB/s\n", name, U, blocks, ms / reps,
           bytes_per_elem * n * reps / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

template <int U>
static void sweep(const f4* s, f4* d, size_t n, float* sink) {
    for (int blocks : {1024, 2048, 4096, 16384}) {
        run<0, U>("read only", s, d, n, sink, blocks, 16);
        run<1, U>("write only, plain stores", s, d, n, sink, blocks, 16);
        run<2, U>("write only, nt stores", s, d, n, sink, blocks, 16);
        run<3, U>("copy, plain stores", s, d, n, sink, blocks, 32);
        run<4, U>("copy, nt stores", s, d, n, sink, blocks, 32);
        run<5, U>("I1 mix 1 read : 2 nt writes", s, d, n / 2, sink, blocks, 48);
        run<6, U>("I2 mix 38 reads : 1 nt write", s, d, n, sink, blocks, 16.0 * 39 / 38);
    }
}

int main() {
    const size_t n = (size_t)3 << 26;                 // 3 GiB per buffer (f4 elements: 201M)
    f4 *s, *d; float* sink;
    hipMalloc(&s, n * 16); hipMalloc(&d, n * 16); hipMalloc(&sink, 4);
    hipMemset(s, 1, n * 16); hipMemset(d, 0, n * 16);
    sweep<1>(s, d, n, sink);
    sweep<4>(s, d, n, sink);
    sweep<8>(s, d, n, sink);
    return 0;
}
