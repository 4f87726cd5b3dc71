// Do reads served by the Infinity Cache add to HBM bandwidth, or share it?  (developer probe)
// Half of the blocks stream a 3 GiB buffer (HBM), the other half sweep a small buffer again and
// again (resident in the 256 MiB Infinity Cache, far beyond the 32 MiB of L2).  If the total rate
// exceeds the 6.3 TB/s of a pure HBM stream, re-reads that hit the Infinity Cache (the mirror
// workgroups' coefficient lines in I1, the sibling rows' lines in I2) cost less than their bytes.
//   hipcc --offload-arch=gfx950 -O3 tools/mallmix.hip -o tools/bin/mallmix
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));

// MODE 0: all blocks stream the big buffer; 1: all blocks sweep the small one; 2: even blocks stream, odd sweep
template <int MODE>
__global__ void __launch_bounds__(256) k(const f4* __restrict__ big, size_t nbig, const f4* __restrict__ small_,
                                         size_t nsmall, float* sink) {
    constexpr int U = 8;
    f4 acc = {0, 0, 0, 0};
    const bool sweeper = MODE == 1 || (MODE == 2 && (blockIdx.x & 1));
    const size_t nb = gridDim.x, b = blockIdx.x;
    const size_t chunk = (size_t)256 * U;
    // every block reads nbig / nb elements in all
    const size_t per_block = nbig / nb / chunk;
    for (size_t c = 0; c < per_block; ++c) {
        size_t base;
        if (!sweeper) base = (c * nb + b) * chunk;
        else base = ((c * nb + b) * chunk) % nsmall;
        const f4* p = (sweeper ? small_ : big) + base + threadIdx.x;
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[(size_t)u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}

template <int MODE>
static void run(const char* name, const f4* big, size_t nbig, const f4* sm, size_t nsmall, float* sink, double bytes) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, dim3(2048), dim3(256), 0, 0, big, nbig, sm, nsmall, sink);
    (void)hipEventRecord(a, 0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<MODE>, dim3(2048), dim3(256), 0, 0, big, nbig, sm, nsmall, sink);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-64s %.3f ms  %.2f TB/s\n", name, ms / reps, bytes * reps / (ms * 1e-3) / 1e12);
}

int main() {
    const size_t nbig = (size_t)3 << 26;             // 3 GiB of f4
    f4 *big, *sm; float* sink;
    (void)hipMalloc(&big, nbig * 16); (void)hipMalloc(&sm, (size_t)256 << 20); (void)hipMalloc(&sink, 4);
    (void)hipMemset(big, 1, nbig * 16); (void)hipMemset(sm, 1, (size_t)256 << 20);
    for (size_t mb : {48, 96, 160}) {
        const size_t nsmall = (mb << 20) / 16;
        printf("-- small buffer %zu MiB\n", mb);
        run<0>("all blocks stream 3 GiB (HBM)", big, nbig, sm, nsmall, sink, 16.0 * nbig);
        run<1>("all blocks sweep the small buffer (Infinity Cache)", big, nbig, sm, nsmall, sink, 16.0 * nbig);
        run<2>("half of the blocks stream, half sweep (1.5 GiB each)", big, nbig, sm, nsmall, sink, 16.0 * nbig);
    }
    return 0;
}
