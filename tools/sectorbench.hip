// Does the memory side of gfx950 move 64-byte sectors or whole 128-byte lines?  (developer probe)
// The I1 -> I2 hand-off stores 128-byte blocks of 2 rows x 8 columns; a row workgroup of I2 uses
// one row of a block, i.e. half of every line it touches.  If the fabric fetches 64-byte sectors,
// a block layout that keeps a row's 8 columns in ONE half of the line lets a row workgroup fetch
// only its own bytes, whatever its sibling does.
//   hipcc --offload-arch=gfx950 -O3 tools/sectorbench.hip -o tools/bin/sectorbench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// MODE 0: every 8-byte word (all of every line)        useful = n * 8
//      1: the first 64 bytes of every 128-byte line     useful = n * 4
//      2: the even 8-byte words of every line (rows2 as I2 reads it today)   useful = n * 4
//      3: first halves by even blocks, second halves by odd blocks - both halves are read, by
//         workgroups 1 block id apart (different XCDs)   useful = n * 8
//      4: the same with the two readers 8 block ids apart (same XCD)          useful = n * 8
template <int MODE, int U>
__global__ void __launch_bounds__(256) kread(const f2* __restrict__ src, size_t nwords, float* sink) {
    // a "slot" is one 8-byte word a lane reads; a wave instruction covers 64 slots
    f2 acc = {0, 0};
    const size_t nslots = (MODE == 0 || MODE == 3 || MODE == 4) ? nwords : nwords / 2;
    const size_t chunk = (size_t)256 * U;
    for (size_t c = blockIdx.x; c * chunk < nslots; c += gridDim.x) {
        f2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            size_t s = c * chunk + (size_t)u * 256 + threadIdx.x;
            size_t w;
            if (MODE == 0) w = s;
            else if (MODE == 1) w = (s >> 3) * 16 + (s & 7);
            else if (MODE == 2) w = s * 2;
            else {
                // block pair (b, b') shares the lines of chunk pair; reader parity picks the half
                const int sh = MODE == 3 ? 0 : 3;
                const size_t cp = ((c >> (sh + 1)) << sh) | (c & ((1u << sh) - 1));   // chunk pair index
                const int half = (c >> sh) & 1;
                size_t t = cp * chunk + (size_t)u * 256 + threadIdx.x;                // slot within the halves
                w = (t >> 3) * 16 + half * 8 + (t & 7);
                if (w >= nwords) w = 0;
            }
            v[u] = src[w];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc.x + acc.y == 12345.678f) *sink = acc.x;
}

// stores: MODE 0 whole lines (16 B per lane, 8 lanes per line)
//         1 64-byte pieces: first halves by even chunks, second halves by odd chunks (1 block id apart)
//         2 the same, the two writers 8 block ids apart
template <int MODE, int U>
__global__ void __launch_bounds__(256) kwrite(f4* __restrict__ dst, size_t nq) {
    const size_t chunk = (size_t)256 * U;
    for (size_t c = blockIdx.x; c * chunk < nq; c += gridDim.x) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            size_t s = c * chunk + (size_t)u * 256 + threadIdx.x, w;
            if (MODE == 0) w = s;
            else {
                const int sh = MODE == 1 ? 0 : 3;
                const size_t cp = ((c >> (sh + 1)) << sh) | (c & ((1u << sh) - 1));
                const int half = (c >> sh) & 1;
                size_t t = cp * chunk + (size_t)u * 256 + threadIdx.x;
                w = (t >> 2) * 8 + half * 4 + (t & 3);
                if (w >= nq) w = 0;
            }
            __builtin_nontemporal_store(f4{1.f, 2.f, 3.f, (float)s}, dst + w);
        }
    }
}

template <int MODE, int U>
static void runr(const char* name, const f2* s, size_t nwords, float* sink, int blocks, double useful) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((kread<MODE, U>), dim3(blocks), dim3(256), 0, 0, s, nwords, sink);
    (void)hipEventRecord(a, 0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((kread<MODE, U>), dim3(blocks), dim3(256), 0, 0, s, nwords, sink);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("read  %-52s U %d blocks %5d  %.3f ms  useful %.2f TB/s\n", name, U, blocks, ms / reps, useful * reps / (ms * 1e-3) / 1e12);
    fflush(stdout);
}
template <int MODE, int U>
static void runw(const char* name, f4* d, size_t nq, int blocks) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((kwrite<MODE, U>), dim3(blocks), dim3(256), 0, 0, d, nq);
    (void)hipEventRecord(a, 0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((kwrite<MODE, U>), dim3(blocks), dim3(256), 0, 0, d, nq);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("write %-52s U %d blocks %5d  %.3f ms  %.2f TB/s\n", name, U, blocks, ms / reps, 16.0 * nq * reps / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

int main() {
    const size_t bytes = (size_t)3 << 30;
    void* p; float* sink;
    (void)hipMalloc(&p, bytes); (void)hipMalloc(&sink, 4);
    (void)hipMemset(p, 1, bytes);
    const size_t nwords = bytes / 8, nq = bytes / 16;
    for (int blocks : {1024, 4096}) {
        runr<0, 8>("all words", (const f2*)p, nwords, sink, blocks, 8.0 * nwords);
        runr<1, 8>("first 64 B of every 128-B line", (const f2*)p, nwords, sink, blocks, 4.0 * nwords);
        runr<2, 8>("even 8-B words of every line", (const f2*)p, nwords, sink, blocks, 4.0 * nwords);
        runr<3, 8>("both halves, readers 1 block id apart", (const f2*)p, nwords, sink, blocks, 8.0 * nwords);
        runr<4, 8>("both halves, readers 8 block ids apart (one XCD)", (const f2*)p, nwords, sink, blocks, 8.0 * nwords);
        runw<0, 4>("whole 128-B lines", (f4*)p, nq, blocks);
        runw<1, 4>("64-B pieces, writers 1 block id apart", (f4*)p, nq, blocks);
        runw<2, 4>("64-B pieces, writers 8 block ids apart (one XCD)", (f4*)p, nq, blocks);
    }
    return 0;
}
