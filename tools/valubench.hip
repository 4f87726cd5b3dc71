// Issue cost of packed vs plain float32 VALU on gfx950 (developer probe): the FFT kernels do their
// complex arithmetic in v_pk_add/mul/fma_f32 (one instruction per complex add).  How many cycles
// does a SIMD spend per instruction, at 1, 2 and 4 waves per SIMD?
//   hipcc --offload-arch=gfx950 -O3 tools/valubench.hip -o tools/bin/valubench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2 __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X
template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, int iters, long long* cyc) {
    v2 a0 = {1.f + threadIdx.x, 2.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f,
       a6 = a0 + 6.f, a7 = a0 + 7.f;
    const v2 b = {1.0001f, 0.9999f}, c = {1e-7f, -1e-7f};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#define OPS(INS, A) asm volatile(INS : "+v"(A) : "v"(b), "v"(c));
        if (MODE == 0) {         // v_pk_fma_f32: 8 independent per REP
            REP8(OPS("v_pk_fma_f32 %0, %0, %1, %2", a0) OPS("v_pk_fma_f32 %0, %0, %1, %2", a1)
                 OPS("v_pk_fma_f32 %0, %0, %1, %2", a2) OPS("v_pk_fma_f32 %0, %0, %1, %2", a3)
                 OPS("v_pk_fma_f32 %0, %0, %1, %2", a4) OPS("v_pk_fma_f32 %0, %0, %1, %2", a5)
                 OPS("v_pk_fma_f32 %0, %0, %1, %2", a6) OPS("v_pk_fma_f32 %0, %0, %1, %2", a7))
        } else if (MODE == 1) {  // v_pk_add_f32
            REP8(OPS("v_pk_add_f32 %0, %0, %1", a0) OPS("v_pk_add_f32 %0, %0, %1", a1)
                 OPS("v_pk_add_f32 %0, %0, %1", a2) OPS("v_pk_add_f32 %0, %0, %1", a3)
                 OPS("v_pk_add_f32 %0, %0, %1", a4) OPS("v_pk_add_f32 %0, %0, %1", a5)
                 OPS("v_pk_add_f32 %0, %0, %1", a6) OPS("v_pk_add_f32 %0, %0, %1", a7))
        } else if (MODE == 2) {  // v_pk_add_f32 with op_sel / neg modifiers (the +-j d form)
            REP8(OPS("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]", a0)
                 OPS("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]", a1)
                 OPS("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]", a2)
                 OPS("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]", a3)
                 OPS("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]", a4)
                 OPS("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]", a5)
                 OPS("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]", a6)
                 OPS("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]", a7))
        } else if (MODE == 3) {  // v_pk_mul_f32
            REP8(OPS("v_pk_mul_f32 %0, %0, %1", a0) OPS("v_pk_mul_f32 %0, %0, %1", a1)
                 OPS("v_pk_mul_f32 %0, %0, %1", a2) OPS("v_pk_mul_f32 %0, %0, %1", a3)
                 OPS("v_pk_mul_f32 %0, %0, %1", a4) OPS("v_pk_mul_f32 %0, %0, %1", a5)
                 OPS("v_pk_mul_f32 %0, %0, %1", a6) OPS("v_pk_mul_f32 %0, %0, %1", a7))
        } else if (MODE == 4) {  // v_fma_f32 on the low halves: 8 independent
#define OPL(INS, A) asm volatile(INS : "+v"(A.x) : "v"(b.x), "v"(c.x));
            REP8(OPL("v_fma_f32 %0, %0, %1, %2", a0) OPL("v_fma_f32 %0, %0, %1, %2", a1)
                 OPL("v_fma_f32 %0, %0, %1, %2", a2) OPL("v_fma_f32 %0, %0, %1, %2", a3)
                 OPL("v_fma_f32 %0, %0, %1, %2", a4) OPL("v_fma_f32 %0, %0, %1, %2", a5)
                 OPL("v_fma_f32 %0, %0, %1, %2", a6) OPL("v_fma_f32 %0, %0, %1, %2", a7))
        } else if (MODE == 5) {  // v_add_f32
            REP8(OPL("v_add_f32 %0, %0, %1", a0) OPL("v_add_f32 %0, %0, %1", a1)
                 OPL("v_add_f32 %0, %0, %1", a2) OPL("v_add_f32 %0, %0, %1", a3)
                 OPL("v_add_f32 %0, %0, %1", a4) OPL("v_add_f32 %0, %0, %1", a5)
                 OPL("v_add_f32 %0, %0, %1", a6) OPL("v_add_f32 %0, %0, %1", a7))
        } else if (MODE == 6) {  // integer v_add_u32 (address arithmetic)
            REP8(OPL("v_add_u32 %0, %0, %1", a0) OPL("v_add_u32 %0, %0, %1", a1)
                 OPL("v_add_u32 %0, %0, %1", a2) OPL("v_add_u32 %0, %0, %1", a3)
                 OPL("v_add_u32 %0, %0, %1", a4) OPL("v_add_u32 %0, %0, %1", a5)
                 OPL("v_add_u32 %0, %0, %1", a6) OPL("v_add_u32 %0, %0, %1", a7))
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    v2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
static void run(const char* name, float* out, long long* cyc, int threads) {
    const int iters = 2000, per_iter = 64;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double wps = threads / 64 / 4.0;                     // waves per SIMD
    const double ins_per_simd = (double)iters * per_iter * wps;
    printf("%-34s waves/SIMD %.0f  %.3f ms  s_memtime ticks per instruction per SIMD %.2f  (wall: %.2f ns)\n", name, wps, ms,
           (double)c / ins_per_simd, ms * 1e6 / ins_per_simd);
}

int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&cyc, 8);
    for (int threads : {256, 512, 1024}) {
        run<0>("v_pk_fma_f32", out, cyc, threads);
        run<1>("v_pk_add_f32", out, cyc, threads);
        run<2>("v_pk_add_f32 op_sel/neg", out, cyc, threads);
        run<3>("v_pk_mul_f32", out, cyc, threads);
        run<4>("v_fma_f32", out, cyc, threads);
        run<5>("v_add_f32", out, cyc, threads);
        run<6>("v_add_u32", out, cyc, threads);
    }
    return 0;
}
