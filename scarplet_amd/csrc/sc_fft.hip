// FFT path: overlap-save tiles, row / column FFTs done entirely in LDS.
//
// Data flow for one orientation (all templates of the orientation share it):
//
//   F1c  k_fwd_rows_curv    curvature of TWO tiles packed as re/im  -> row FFT
//   F2   k_fwd_cols         column FFT -> uc, uc2 (spectra of curv, curv^2)
//   F1t  k_fwd_rows_templ   template tile v = alpha*W + iM          -> row FFT
//   F2   k_fwd_cols         column FFT -> vh
//   S    k_split_templ_sym  Scarp / Ricker: FFT(W) = {i} a P, FFT(M) = b P with a, b
//                           real (flip symmetry of the template): store a, b only
//        k_split_templ      any other template: FFT(W), FFT(M) as complex half planes
//   I1   k_inv_cols_symx    Y = a * (X P {i}) per column block, inverse column FFT; column
//                           blocks and their mirrors in one launch, paired per XCD
//        k_inv_cols_sym     the same as two launches (small tiles; cross-check variant 6)
//        k_inv_cols         the same with complex spectra
//   I2   k_inv_rows_fast    inverse row FFT -> xcorr(A) + i xcorr(B), T3 likewise;
//        k_inv_rows         float32 amp/SNR epilogue, masks, running-best fold
//                           (fast: T in {512, 1024, 2048}; generic: the rest)
//
// Two real tiles ride in one complex transform (the template is real), so no
// real-to-complex bookkeeping is needed anywhere; an unpaired tile carries two
// templates instead (template parameter PT of I1 / I2).
//
// Layouts (all complex float32):
//   "spectrum"  [fx][fy]: a column of the 2-D spectrum is contiguous; written
//               by the column kernels, read by I1;
//   "blocked"   4x4 cells per 128-byte block, blocks row-major: the hand-off
//               between row kernels (which own 4 full rows = one contiguous
//               4*Tx run) and column kernels (which own 4 full columns = one
//               128-byte block per 4 rows).  Both sides move whole 128-byte
//               lines, which is what fuses the transpose into the FFT kernels.
//
//   "rows2"     hand-off between I1 and I2: 128-byte blocks of 2 rows x 8
//               columns, cell (r, c) at ((r/2)*(Tx/8) + c/8)*16 + (c%8)*2 + r%2.
//               I1 (4 columns) writes 64-byte half blocks; I2 reads whole rows,
//               the two rows of a pair from sibling workgroups on one XCD.
//
// The in-LDS FFT is a Stockham autosort with radices 16,16,..,r (see below);
// complex arithmetic is packed (namespace pk).
#include "sc_internal.h"
#include <math.h>
#include <type_traits>
#include <string.h>

struct TileDev {
    int i0, j0, vy, vx, gi0, gj0;
};

#ifndef SC_I1_XLANE
#define SC_I1_XLANE 1      // k_inv_cols_w8 / w4: stage 2 -> 3 exchanged across lanes (v_permlane swaps), not through the LDS
#endif
#ifndef SC_I2_RD1
#define SC_I2_RD1 1        // the row pass's stage-2 and stage-3 cells read singly too
#endif
#ifndef SC_LDS_WR1
#define SC_LDS_WR1 0      // ... and written one ds_write_b64 each (wave-per-column kernels)
#endif
#ifndef SC_LDS_RD1
#define SC_LDS_RD1 1      // LDS cells read one ds_read_b64 each (0: the compiler's ds_read2_b64 pairs)
#endif
#ifndef SC_H2_XLANE
#define SC_H2_XLANE 1      // k_inv_cols_h2: the last stage (radix 2) across lanes with v_permlane16_swap instead of through the LDS
#endif
#ifndef SC_F1C_HOLD
#define SC_F1C_HOLD 1      // k_fwd_rows_curv<.., MIX>: the mixed curvature held in registers for the second plane
#endif
#ifndef SC_I1_TWTAB
#ifndef SC_Y_ROWMAJOR
#define SC_Y_ROWMAJOR 0    // lab (k_inv_cols_w8 + k_inv_rows_fast only): the I1 -> I2 hand-off row-major instead of rows2 blocks
#endif
#define SC_I1_TWTAB 0      // 1: the wave-per-column kernels read all fifteen twiddles of a set from LDS tables
#endif


// ---------------------------------------------------------------------------
// In-LDS FFT of 4 lines of length T (complex float32).
//
// Stockham autosort, radices 16,16,..,r (r = T / 16^a in {1,2,4,8}); natural
// order in and out.  Every thread owns one 16-point set per stage - elements
// tt + j*T/16 - which is one radix-16 butterfly, or 16/r radix-r butterflies in
// the last stage: the load pattern is the same in every stage and is contiguous
// across lanes.  Outputs go to q + st*(R*p + m).  Lines are stored with one pad
// element per 16 (index i lives at i + i/16): the first stage's stride-16
// stores then fall on distinct banks (17*tt + m), later stages are contiguous.
// Twiddles exist only in radix-16 stages: w^(e*m), e = p*st, from four table
// loads (m = 1, 2, 4, 8) and eleven products.
// ---------------------------------------------------------------------------
__host__ __device__ constexpr int fft_threads(int T) {
    return (T / 4) >= 512 ? 512 : ((T / 4) < 64 ? 64 : (T / 4));
}
// minimum waves per SIMD the forward kernels are compiled for (the second
// __launch_bounds__ argument of hipcc): 4 = at most 128 VGPRs, i.e. two
// 512-thread workgroups per CU where LDS allows it (T <= 2048)
__host__ __device__ constexpr int fft_waves(int T) { return T >= 4096 ? 2 : 4; }
// padded line: one pad per 16 elements, plus 8 so that consecutive lines start
// 16 banks apart (lanes of one wave may alternate between two lines)
__host__ __device__ constexpr int fft_line(int T) { return T + T / 16 + 8; }
__host__ __device__ constexpr size_t fft_lds_bytes(int T) {
    return (size_t)4 * fft_line(T) * sizeof(float2);
}
__device__ __forceinline__ int ph(int i) { return i + (i >> 4); }

// Workgroup barrier for LDS hand-offs that leaves global loads in flight.
// __syncthreads() also drains vmcnt (its fence covers global memory), which
// would serialise the register prefetch of the next step behind every FFT
// stage; here only this wave's LDS traffic is waited for (lgkmcnt) before the
// barrier.  The "memory" clobbers keep the compiler from moving LDS accesses
// across it; waits for prefetched registers are inserted by the compiler at
// their first use.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// 16-byte store of two cells of Y, non-temporal: Y is written once and read by
// the row pass two gigabytes of traffic later, so it should not displace anything
// in L2 (sustained C3 run: 4.49 -> 4.42 s; non-temporal LOADS of the coefficient
// stream cost 16 % instead - they stop the compiler from hoisting the prefetch).
__device__ __forceinline__ void store_stream(float2* dst, float2 a, float2 b) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = {a.x, a.y, b.x, b.y};
    __builtin_nontemporal_store(v, reinterpret_cast<f4*>(dst));
}
// ---- sibling rendezvous (a speed hint, never needed for correctness) ---------------------------
// Two workgroups that stream the SAME 128-byte lines - the two row workgroups of a rows2 pair in
// I2, a column block and its mirror block in I1 - sit on one XCD (block ids 8 apart) so that the
// second reader hits that XCD's L2.  It only does while the two stay within a fraction of a
// template of each other: an XCD's 4 MB of L2 holds about ONE template step of its workgroups'
// lines (tools/sectorbench.hip: co-timed readers of the two halves share perfectly, 6.1 - 6.3 TB/s
// useful; the fabric moves whole lines, half-line reads cost the full line).  Left alone the
// siblings drift and a third of the second reads go to memory again (I2: 1.17x its algorithmic
// bytes).  So each workgroup publishes how many template fetches it has issued, and does not issue
// fetch f before its sibling has issued fetch f - 1.  The wait is bounded: a sibling that is not
// resident, or sits on another XCD and is therefore never seen, costs time, never the result.
struct SibSync {
    uint32_t* slots;       // one word per workgroup of the launch (nullptr: no rendezvous)
    uint32_t base;         // launch epoch << 8: words of earlier launches compare below it
};
// The word travels through the XCD's L2: a plain store leaves it there, an sc1 load reads it from
// there past the reader's L1 (sc1 on both sides goes to the memory side instead: 1 - 2 us per look,
// measured as +3 % on the row pass).  Siblings on different XCDs would never see each other's word:
// the first wait that runs out switches the waiting off for the rest of the workgroup's life.
__device__ __forceinline__ void sib_publish(uint32_t* mine, uint32_t v) {
    asm volatile("global_store_dword %0, %1, off" :: "v"(mine), "v"(v) : "memory");
}
__device__ __forceinline__ uint32_t sib_peek(const uint32_t* theirs) {
    return __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// false: gave up (the caller stops waiting from here on)
__device__ __forceinline__ bool sib_wait(const uint32_t* theirs, uint32_t target) {
    for (int budget = 64; budget > 0; --budget) {
        if ((int32_t)(sib_peek(theirs) - target) >= 0) return true;
        __builtin_amdgcn_s_sleep(4);
    }
    return false;
}
constexpr uint32_t SIB_DONE = 0xFFu;

#ifdef SC_ABLATE
// store flavours for the ablation build (tools/ablate.sh): 0 non-temporal (production),
// 1 plain, 2 sc1 (write-through, line dropped from L2), 3 sc0 sc1
__device__ __forceinline__ void store_flavour(float2* dst, float2 a, float2 b, int mode) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = {a.x, a.y, b.x, b.y};
    if (mode == 1) { *reinterpret_cast<f4*>(dst) = v; return; }
    if (mode == 2) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(dst), "v"(v) : "memory"); return; }
    if (mode == 3) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(dst), "v"(v) : "memory"); return; }
    __builtin_nontemporal_store(v, reinterpret_cast<f4*>(dst));
}
#endif
template <int T>
__device__ __forceinline__ int lidx(int line, int i) { return line * fft_line(T) + ph(i); }

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) {
    return make_float2(a.x + b.x, a.y + b.y);
}
__device__ __forceinline__ float2 csub(float2 a, float2 b) {
    return make_float2(a.x - b.x, a.y - b.y);
}
// multiply by -j (forward) / +j (inverse)
template <bool INV>
__device__ __forceinline__ float2 mulj(float2 a) {
    return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}
template <bool INV>
__device__ __forceinline__ float2 cw(float c, float s) {      // exp(-/+ i*angle)
    return make_float2(c, INV ? s : -s);
}
template <bool INV>
__device__ __forceinline__ void dft4(float2& a0, float2& a1, float2& a2, float2& a3) {
    float2 apc = cadd(a0, a2), amc = csub(a0, a2);
    float2 bpd = cadd(a1, a3), bmd = mulj<INV>(csub(a1, a3));
    a0 = cadd(apc, bpd);
    a1 = cadd(amc, bmd);
    a2 = csub(apc, bpd);
    a3 = csub(amc, bmd);
}

// radix-R butterfly on v[0..R-1] (natural order in); result X[m] is returned
// through out(m).  All indices are compile-time after unrolling.
template <int R, bool INV>
struct Bfly;

template <bool INV>
struct Bfly<2, INV> {
    static __device__ __forceinline__ void run(float2* v) {
        float2 a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    }
    static __device__ __forceinline__ constexpr int pos(int m) { return m; }
};
template <bool INV>
struct Bfly<4, INV> {
    static __device__ __forceinline__ void run(float2* v) { dft4<INV>(v[0], v[1], v[2], v[3]); }
    static __device__ __forceinline__ constexpr int pos(int m) { return m; }
};
template <bool INV>
struct Bfly<8, INV> {
    // j = c + 2d, m = r + 4s: DFT4 over d, twiddle w8^(c r), DFT2 over c
    static __device__ __forceinline__ void run(float2* v) {
        dft4<INV>(v[0], v[2], v[4], v[6]);
        dft4<INV>(v[1], v[3], v[5], v[7]);
        const float h = 0.70710678118654752f;
        v[3] = cmul(v[3], cw<INV>(h, h));
        v[5] = mulj<INV>(v[5]);
        v[7] = cmul(v[7], cw<INV>(-h, h));
        Bfly<2, INV>::run(v + 0);
        Bfly<2, INV>::run(v + 2);
        Bfly<2, INV>::run(v + 4);
        Bfly<2, INV>::run(v + 6);
    }
    // X[r + 4s] sits at v[2r + s]
    static __device__ __forceinline__ constexpr int pos(int m) { return 2 * (m & 3) + (m >> 2); }
};
template <bool INV>
struct Bfly<16, INV> {
    // j = c + 4d, m = r + 4s: DFT4 over d, twiddle w16^(c r), DFT4 over c
    static __device__ __forceinline__ void run(float2* v) {
#pragma unroll
        for (int c = 0; c < 4; ++c) dft4<INV>(v[c], v[c + 4], v[c + 8], v[c + 12]);
        const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f;
        const float h = 0.70710678118654752f;
        v[1 + 4] = cmul(v[1 + 4], cw<INV>(c1, s1));       // w^1
        v[1 + 8] = cmul(v[1 + 8], cw<INV>(h, h));         // w^2
        v[1 + 12] = cmul(v[1 + 12], cw<INV>(s1, c1));     // w^3
        v[2 + 4] = cmul(v[2 + 4], cw<INV>(h, h));         // w^2
        v[2 + 8] = mulj<INV>(v[2 + 8]);                   // w^4
        v[2 + 12] = cmul(v[2 + 12], cw<INV>(-h, h));      // w^6
        v[3 + 4] = cmul(v[3 + 4], cw<INV>(s1, c1));       // w^3
        v[3 + 8] = cmul(v[3 + 8], cw<INV>(-h, h));        // w^6
        v[3 + 12] = cmul(v[3 + 12], cw<INV>(-c1, -s1));   // w^9
#pragma unroll
        for (int r = 0; r < 4; ++r) dft4<INV>(v[4 * r], v[4 * r + 1], v[4 * r + 2], v[4 * r + 3]);
    }
    // X[r + 4s] sits at v[4r + s]
    static __device__ __forceinline__ constexpr int pos(int m) { return 4 * (m & 3) + (m >> 2); }
};

// ---------------------------------------------------------------------------
// Packed complex arithmetic: a complex value is a native 2-vector, so a complex
// add is one v_pk_add_f32 and a complex product two instructions (v_pk_mul_f32 +
// v_pk_fma_f32), the half-swaps and signs riding on the op_sel / neg modifiers.
// Left to itself the compiler pairs unrelated scalars into packed operations and
// pays for it in register moves.
// ---------------------------------------------------------------------------
namespace pk {
typedef float v2 __attribute__((ext_vector_type(2)));

// a * w
__device__ __forceinline__ v2 cmul(v2 a, v2 w) {
    v2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// a * conj(w)
__device__ __forceinline__ v2 cmulc(v2 a, v2 w) {
    v2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]"
        : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// a * w with w a compile-time constant held in a scalar register pair
__device__ __forceinline__ v2 cmul_k(v2 a, v2 w) {
    v2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "s"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=v"(r) : "v"(a), "s"(w), "v"(t));
    return r;
}
// multiply by -j (forward) / +j (inverse); exp(-/+ i angle) from (cos, sin)
template <bool INV>
__device__ __forceinline__ v2 mulj(v2 a) { return INV ? v2{-a.y, a.x} : v2{a.y, -a.x}; }
template <bool INV>
__device__ __forceinline__ v2 cw(float c, float s) { return v2{c, INV ? s : -s}; }
// a + j*d and a - j*d in one instruction each: the swap of d's halves and the sign
// ride on the modifiers (written out, the compiler spends an extra v_xor per j)
__device__ __forceinline__ v2 add_jd(v2 a, v2 d) {           // (a.x - d.y, a.y + d.x)
    v2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(d));
    return r;
}
__device__ __forceinline__ v2 sub_jd(v2 a, v2 d) {           // (a.x + d.y, a.y - d.x)
    v2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(d));
    return r;
}
template <bool INV>
__device__ __forceinline__ void dft4(v2& a0, v2& a1, v2& a2, v2& a3) {
    const v2 apc = a0 + a2, amc = a0 - a2, bpd = a1 + a3, d = a1 - a3;
    a0 = apc + bpd;
    a2 = apc - bpd;
    a1 = INV ? add_jd(amc, d) : sub_jd(amc, d);
    a3 = INV ? sub_jd(amc, d) : add_jd(amc, d);
}
// radix-R butterfly on v[0..R-1] (natural order in); X[m] ends up in v[pos(m)]
template <int R, bool INV>
struct B;
template <bool INV>
struct B<2, INV> {
    static __device__ __forceinline__ void run(v2* v) {
        v2 a = v[0], b = v[1];
        v[0] = a + b;
        v[1] = a - b;
    }
    static __device__ __forceinline__ constexpr int pos(int m) { return m; }
};
template <bool INV>
struct B<4, INV> {
    static __device__ __forceinline__ void run(v2* v) { dft4<INV>(v[0], v[1], v[2], v[3]); }
    static __device__ __forceinline__ constexpr int pos(int m) { return m; }
};
template <bool INV>
struct B<8, INV> {
    // j = c + 2d, m = r + 4s: DFT4 over d, twiddle w8^(c r), DFT2 over c
    static __device__ __forceinline__ void run(v2* v) {
        dft4<INV>(v[0], v[2], v[4], v[6]);
        dft4<INV>(v[1], v[3], v[5], v[7]);
        const float h = 0.70710678118654752f;
        v[3] = cmul_k(v[3], cw<INV>(h, h));
        v[5] = mulj<INV>(v[5]);
        v[7] = cmul_k(v[7], cw<INV>(-h, h));
        B<2, INV>::run(v + 0);
        B<2, INV>::run(v + 2);
        B<2, INV>::run(v + 4);
        B<2, INV>::run(v + 6);
    }
    // X[r + 4s] sits at v[2r + s]
    static __device__ __forceinline__ constexpr int pos(int m) { return 2 * (m & 3) + (m >> 2); }
};
template <bool INV>
struct B<16, INV> {
    // j = c + 4d, m = r + 4s: DFT4 over d, twiddle w16^(c r), DFT4 over c
    static __device__ __forceinline__ void run(v2* v) {
#pragma unroll
        for (int c = 0; c < 4; ++c) dft4<INV>(v[c], v[c + 4], v[c + 8], v[c + 12]);
        const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f;
        const float h = 0.70710678118654752f;
        v[5] = cmul_k(v[5], cw<INV>(c1, s1));        // w^1
        v[9] = cmul_k(v[9], cw<INV>(h, h));          // w^2
        v[13] = cmul_k(v[13], cw<INV>(s1, c1));      // w^3
        v[6] = cmul_k(v[6], cw<INV>(h, h));          // w^2
        v[10] = mulj<INV>(v[10]);                    // w^4
        v[14] = cmul_k(v[14], cw<INV>(-h, h));       // w^6
        v[7] = cmul_k(v[7], cw<INV>(s1, c1));        // w^3
        v[11] = cmul_k(v[11], cw<INV>(-h, h));       // w^6
        v[15] = cmul_k(v[15], cw<INV>(-c1, -s1));    // w^9
#pragma unroll
        for (int r = 0; r < 4; ++r) dft4<INV>(v[4 * r], v[4 * r + 1], v[4 * r + 2], v[4 * r + 3]);
    }
    // X[r + 4s] sits at v[4r + s]
    static __device__ __forceinline__ constexpr int pos(int m) { return 4 * (m & 3) + (m >> 2); }
};
}  // namespace pk

// Twiddle bases of a thread, loaded ONCE per kernel: in the radix-16 stage at
// stride 2^LST a thread needs w^(e), w^(2e), w^(4e), w^(8e) with e = p << LST,
// which depends only on the thread (its set index tt), not on the data.  Kept
// in registers so that no global load sits inside the transform: a load there
// would drag every outstanding prefetch with it (vmcnt completes in order).
template <int T>
struct FftTw {
    static constexpr int S = T / 16;
    static constexpr int NT = fft_threads(T);
    static constexpr int U = (4 * S + NT - 1) / NT;
    static constexpr int LOGT = __builtin_ctz(T);
    static constexpr int NST = (LOGT + 3) / 4;        // stages
    static constexpr int NTW = NST - 1;               // all but the last carry twiddles
    float2 w[NTW > 0 ? NTW : 1][U][4];
    __device__ __forceinline__ const float2 (&get(int k, int u, float2 (&)[4]) const)[4] { return w[k][u]; }
    __device__ __forceinline__ void load(const float2* __restrict__ tw) {
#pragma unroll
        for (int k = 0; k < NTW; ++k) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int id = threadIdx.x + u * NT;
                int tt = id % S;                       // bt == tt in radix-16 stages
                int e = (tt >> (4 * k)) << (4 * k);
                w[k][u][0] = tw[e];
                w[k][u][1] = tw[2 * e];
                w[k][u][2] = tw[4 * e];
                w[k][u][3] = tw[8 * e];
            }
        }
    }
};

// ---- one Stockham stage, per 16-point set -------------------------------------
// For T >= 256 every LDS address of a stage is (one per-thread base) +
// (compile-time offset): S = T/16 and the strides are multiples of 16, so the
// padding term (i >> 4) is affine in j and m, and the accesses compile to
// ds_read/ds_write with immediate offsets instead of 32 address registers.

// a[j] = line[tt + j*S]
// (SC_LDS_RD1: every cell with a ds_read_b64 of its own.  Left to itself the compiler pairs the reads of one base
//  into ds_read2_b64 - 8 LDS cycles per wave instruction for 16 bytes per lane, against 2 x 2 cycles for two
//  ds_read_b64 (MI355X_MICROARCH.md, LDS table: 128 against 256 B/clk/CU).  The wave-per-column pass issues 39 of
//  them per transform and plane, eight waves at a time: a sixth of its LDS cycles.  RD1: that kernel only - the
//  forward row pass is 13 % slower with single reads.)
template <bool RD1, typename V2>
__device__ __forceinline__ V2 lds_cell2(const V2* p) {            // the same for the packed two-float vector type
    if constexpr (!RD1 || !SC_LDS_RD1) return *p;
    typedef const volatile __attribute__((address_space(3))) unsigned long long* lds_u64p;
    const unsigned long long v = *(lds_u64p)(p);
    return V2{__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32))};
}
template <bool RD1>
__device__ __forceinline__ float2 lds_cell(const float2* p) {
    if constexpr (!RD1) return *p;
#if SC_LDS_RD1
    typedef const volatile __attribute__((address_space(3))) unsigned long long* lds_u64p;    // (volatile keeps the reads single; the LDS address space keeps them ds_ loads)
    const unsigned long long v = *(lds_u64p)(p);
    return make_float2(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
#else
    return *p;
#endif
}
template <int T, bool RD1 = false>
__device__ __forceinline__ void set_load(const float2* line, int tt, float2 (&a)[16]) {
    constexpr int S = T / 16;
    if ((S % 16) == 0) {
        const float2* rb = line + ph(tt);
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j] = lds_cell<RD1>(rb + j * (S + S / 16));
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j] = lds_cell<RD1>(line + ph(tt + j * S));
    }
}

// twiddle applied to output m of a radix-16 butterfly, from the four bases
// w^(e), w^(2e), w^(4e), w^(8e) of the forward table: w^k for k = 1..7 one at a
// time, each also serving k + 8; the inverse transform multiplies by the
// conjugate.  `put(m, value)` receives the finished outputs.
template <bool INV, typename PUT>
__device__ __forceinline__ void twiddle16(const pk::v2 (&v)[16], const float2 (&w)[4], PUT put) {
    using pk::v2;
    const v2 w1 = v2{w[0].x, w[0].y}, w2 = v2{w[1].x, w[1].y}, w4 = v2{w[2].x, w[2].y},
             w8 = v2{w[3].x, w[3].y};
    auto tw = [](v2 a, v2 wk) { return INV ? pk::cmulc(a, wk) : pk::cmul(a, wk); };
    put(0, v[pk::B<16, INV>::pos(0)]);
    put(8, tw(v[pk::B<16, INV>::pos(8)], w8));
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        v2 wk = (k & 1) ? w1 : v2{1.f, 0.f};
        if (k == 2 || k == 6) wk = w2;
        if (k == 3 || k == 7) wk = pk::cmul(w1, w2);
        if (k == 4) wk = w4;
        if (k >= 5) wk = pk::cmul(wk, w4);
        put(k, tw(v[pk::B<16, INV>::pos(k)], wk));
        put(k + 8, tw(v[pk::B<16, INV>::pos(k + 8)], pk::cmul(wk, w8)));
    }
}

// The fifteen twiddles w^k, k = 1 .. 15, of a radix-16 set as twiddle16 forms them from the four bases -
// the same products in the same order, so that a table of them gives the same bits as forming them in place.
__device__ __forceinline__ void twiddle16_expand(const float2 (&w)[4], pk::v2 (&wk)[16]) {
    using pk::v2;
    const v2 w1 = v2{w[0].x, w[0].y}, w2 = v2{w[1].x, w[1].y}, w4 = v2{w[2].x, w[2].y}, w8 = v2{w[3].x, w[3].y};
    wk[0] = v2{1.f, 0.f};
    wk[8] = w8;
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        v2 x = (k & 1) ? w1 : v2{1.f, 0.f};
        if (k == 2 || k == 6) x = w2;
        if (k == 3 || k == 7) x = pk::cmul(w1, w2);
        if (k == 4) x = w4;
        if (k >= 5) x = pk::cmul(x, w4);
        wk[k] = x;
        wk[k + 8] = pk::cmul(x, w8);
    }
}
// twiddle16 with the fifteen twiddles read from a table (get(k), k = 1 .. 15) instead of formed from four bases
template <bool INV, typename GET, typename PUT>
__device__ __forceinline__ void twiddle16_tab(const pk::v2 (&v)[16], GET get, PUT put) {
    using pk::v2;
    auto tw = [](v2 a, v2 wk) { return INV ? pk::cmulc(a, wk) : pk::cmul(a, wk); };
    put(0, v[pk::B<16, INV>::pos(0)]);
    put(8, tw(v[pk::B<16, INV>::pos(8)], get(8)));
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        put(k, tw(v[pk::B<16, INV>::pos(k)], get(k)));
        put(k + 8, tw(v[pk::B<16, INV>::pos(k + 8)], get(k + 8)));
    }
}

// butterflies of the set (tt) in the stage (R, LST) and store to `line`; TAB: the twiddles of a radix-16 stage
// come from a table of all fifteen (wtab[k * wstride], k = 1 .. 15) instead of the four bases w
// ---- stage 2 -> stage 3 of a length-2048 (1024) line WITHOUT the LDS (k_inv_cols_w8, SC_I1_XLANE) --------------------------
// (written for 2048 = 16 x 16 x 8, two sets per lane; 1024 = 16 x 16 x 4 is the same exchange with one set and four butterflies)
// Stage 2 (radix 16, stride 16) of the wave's line leaves output m of lane L's set u at element
//   (L & 15) + 16 m + 256 ((L >> 4) + 4 u);
// stage 3 (radix 8, stride 256) wants, in lane L3, for its butterflies bt3 = L3 + 64 u' + 128 b, the elements bt3 + 256 j:
// output m = (L3 >> 4) + 4 (u' + 2 b) of the lanes (L3 & 15) + 16 k, sets u, with j = k + 4 u.  The exchange stays inside the
// four lanes that share L & 15 - a 4 x 4 transpose between the 16-lane row L >> 4 and m & 3 for each of (u, m >> 2, re / im):
// two v_permlane32_swap and two v_permlane16_swap per four registers, 64 of them for the 32 cells, instead of 32 cells
// written to the LDS (6 cycles each, eight waves at a time) and read back.  Data movement only: the same values.
__device__ __forceinline__ void xlane_swap32(float& a, float& b) {        // a's lanes 32-63 <-> b's lanes 0-31
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void xlane_swap16(float& a, float& b) {        // a's odd rows of 16 lanes <-> b's even rows
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
// r[k] of the lane in row q <- r[q] of the lane in row k (rows of 16 lanes, same lane & 15)
__device__ __forceinline__ void xlane_transpose4(float& r0, float& r1, float& r2, float& r3) {
    xlane_swap32(r0, r2);
    xlane_swap32(r1, r3);
    xlane_swap16(r0, r1);
    xlane_swap16(r2, r3);
}
// a radix-16 stage's butterfly and twiddles with the outputs kept: out[m] = output m (set_compute_store's arithmetic)
template <bool INV, bool TAB = false>
__device__ __forceinline__ void set_compute_regs(const float2 (&a)[16], const float2 (&w)[4], float2 (&out)[16],
                                                 const float2* wtab = nullptr, int wstride = 0) {
    using pk::v2;
    v2 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = v2{a[k].x, a[k].y};
    pk::B<16, INV>::run(v);
    auto put = [&](int m, v2 val) { out[m] = make_float2(val.x, val.y); };
    if constexpr (TAB)
        twiddle16_tab<INV>(v, [&](int k) { const float2 x = lds_cell<true>(wtab + k * wstride); return v2{x.x, x.y}; }, put);
    else
        twiddle16<INV>(v, w, put);
}

template <int T, int R, int LST, bool INV, bool TAB = false, bool WR1 = false>
__device__ __forceinline__ void set_compute_store(float2* line_, int tt, float2 (&a)[16],
                                                  const float2 (&w)[4], const float2* wtab = nullptr, int wstride = 0) {
    using pk::v2;
    constexpr int S = T / 16;
    constexpr int NB = 16 / R;
    constexpr int LR = __builtin_ctz(R);
    constexpr int ST = 1 << LST;
    constexpr bool LAST = (R << LST) == T;     // n == R: all twiddles are 1
    static_assert(LAST || R == 16, "twiddled stages are radix 16");
    v2* line = reinterpret_cast<v2*>(line_);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        v2 v[R];
#pragma unroll
        for (int k = 0; k < R; ++k) v[k] = v2{a[b + NB * k].x, a[b + NB * k].y};
        pk::B<R, INV>::run(v);
        const int bt = tt + b * S;             // butterfly index in [0, T/R)
        const int p = bt >> LST, q = bt & (ST - 1);
        const int o = q + (p << (LST + LR));
        // output m goes to element o + m*ST: one base per butterfly and compile-time offsets where
        // the padding allows it.  (The choice is made at compile time: a null base as the "no base"
        // flag was not folded - LDS address 0 is a valid one - and cost two selects per output.)
        constexpr bool STRIDED = LST >= 4, UNIT = LST == 0 && R == 16;
        v2* const wb = STRIDED ? line + ph(o) : (UNIT ? line + 17 * bt : line);
        constexpr int wstep = STRIDED ? ST + ST / 16 : 1;      // padded distance of ST elements
        auto put = [&](int m, v2 val) {
            v2* dst = (STRIDED || UNIT) ? wb + m * wstep : line + ph(o + (m << LST));
            if constexpr (WR1 && SC_LDS_WR1) {       // (one ds_write_b64 per cell, see lds_cell)
                typedef volatile __attribute__((address_space(3))) unsigned long long* lds_u64w;
                *(lds_u64w)(dst) = (unsigned long long)__float_as_uint(val.x) | ((unsigned long long)__float_as_uint(val.y) << 32);
            } else {
                *dst = val;
            }
        };
        if constexpr (LAST) {
#pragma unroll
            for (int m = 0; m < R; ++m) put(m, v[pk::B<R, INV>::pos(m)]);
        } else if constexpr (TAB) {
            twiddle16_tab<INV>(v, [&](int k) { const float2 x = wtab[k * wstride]; return v2{x.x, x.y}; }, put);
        } else {
            twiddle16<INV>(v, w, put);
        }
    }
}

// one Stockham stage of radix R at stride 2^LST over the 4 lines in LDS,
// standard mapping: set id -> line id / S, tt = id % S
template <int T, int R, int LST, bool INV, bool REBUILD, typename TW>
__device__ __forceinline__ void fft_stage(float2* s, const TW& twr) {
    constexpr int S = T / 16;                  // 16-point sets per line
    constexpr int NT = fft_threads(T);
    constexpr int U = (4 * S + NT - 1) / NT;   // sets per thread
    float2 a[U][16];
    // REBUILD: the thread id as a value the compiler cannot trace, so that the stage's LDS
    // addresses are rebuilt from it (a few integer operations) where they are used.  Traced,
    // the addresses of all stages are hoisted to the top of the kernel and out of its loops:
    // right where a loop runs over many templates, a loss where it runs twice (the forward
    // kernels: the hoisted values were spilled to scratch there).
    int tid = threadIdx.x;
    if constexpr (REBUILD) asm volatile("" : "+v"(tid));
    __builtin_assume(tid >= 0 && tid < NT);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        int id = tid + u * NT;
        if (id < 4 * S) set_load<T>(s + (id / S) * fft_line(T), id % S, a[u]);
    }
    // the last stage works in place: butterfly bt reads the elements bt + k*T/R and
    // writes bt + m*T/R, the same cells - no other thread is waiting for them
    float2 wl[U][4];
    if constexpr ((R << LST) != T) lds_barrier();
#pragma unroll
    for (int u = 0; u < U; ++u) {
        int id = tid + u * NT;
        if (id < 4 * S)
            set_compute_store<T, R, LST, INV>(s + (id / S) * fft_line(T), id % S, a[u],
                                              twr.get((LST / 4) < TW::NTW ? LST / 4 : 0, u, wl[u]));
    }
    lds_barrier();
}

template <int T, int LST, bool INV, bool REBUILD, typename TW>
__device__ __forceinline__ void fft_stages(float2* s, const TW& twr) {
    if constexpr ((1 << LST) < T) {
        constexpr int REM = T >> LST;
        constexpr int R = REM >= 16 ? 16 : REM;
        fft_stage<T, R, LST, INV, REBUILD>(s, twr);
        fft_stages<T, LST + __builtin_ctz(R), INV, REBUILD>(s, twr);
    }
}

// Transform the 4 lines at s (padded layout, see lidx).  Ends with a barrier.
template <int T, bool INV, bool REBUILD = false, typename TW>
__device__ __forceinline__ void fft4_lines(float2* s, const TW& twr) {
    fft_stages<T, 0, INV, REBUILD>(s, twr);
}

// ---- F1c: curvature of a tile pair -> row FFT -> blocked ---------------------
// grid = (Ty/4/FWD_ROWS_RBW, npairs); out plane index = pair*2 + {0: curv, 1: curv^2}
constexpr int FWD_ROWS_RBW = 2;            // row blocks (of 4 tile rows) per workgroup
// MIX: the orientation's curvature plane is not read but formed here, cell by cell, from the three
// alpha-independent stencil planes: curv = cc A - sc2 B + ss C (dem.py:103-104) in float32, the very
// expression and evaluation order of k_curv_alpha (no contraction: the file is built with
// -ffp-contract=off) - the same bits.  k_curv_alpha wrote a plane per orientation that this kernel
// read straight back (1.2 GB in, 0.4 GB out, 0.4 GB in again at 10000 x 10000); fused, the planes are
// read once per tile cell and nothing is written.
struct CurvMix { float c[SC_MAX_ORIENT][3]; };
template <int TX, bool MIX>
__global__ void __launch_bounds__(fft_threads(TX), 4)
k_fwd_rows_curv(const float* __restrict__ curv, const float* __restrict__ pB, const float* __restrict__ pC,
                CurvMix mixc, Geom g,
                const TileDev* __restrict__ tiles, int Ty,
                const float2* __restrict__ tw, float2* __restrict__ blk,
                double* __restrict__ norm_part, int dbg, int np, size_t curv_stride) {
    // blockIdx.y = b * np + p: tile pair p of the b-th orientation of the launch (its
    // curvature plane lies curv_stride floats further on; MIX: its coefficients are mixc.c[b],
    // `curv` is plane A); small searches batch several orientations per launch (sc_api.hip,
    // "orientation batching")
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    FftTw<TX> twr;
    twr.load(tw);
    constexpr int NT = fft_threads(TX);
    constexpr int RBW = FWD_ROWS_RBW;
    const int rb0 = blockIdx.x * RBW, pair = blockIdx.y;
    const int ob = pair / np, ptile = pair - ob * np;
    float m_cc = 0.f, m_sc2 = 0.f, m_ss = 0.f;
    if constexpr (MIX) {
        m_cc = mixc.c[ob][0]; m_sc2 = mixc.c[ob][1]; m_ss = mixc.c[ob][2];
    } else {
        curv += (size_t)ob * curv_stride;
    }
    const TileDev ta = tiles[2 * ptile], tb = tiles[2 * ptile + 1];
    const size_t plane = (size_t)Ty * TX;
    constexpr int E = 4 * TX / NT;
    static_assert(TX % NT == 0, "a thread's cell u lies in row (u*NT)/TX, column (u*NT)%TX + tid");
    // A thread's cells lie in CPR columns (tid + k*NT) of each of the 4 rows: the CPR column
    // offsets are worked out once per tile (on the periodic DEM one division per thread, then
    // a step of NT mod nx with one conditional subtraction), the row bases are uniform.
    constexpr int CPR = TX / NT;
    auto load_tile = [&](const TileDev& t, float (&v)[E], int tid, int rb) {
        if (t.vy <= 0) {
#pragma unroll
            for (int u = 0; u < E; ++u) v[u] = 0.f;
            return;
        }
        // every load is issued unconditionally, from an address made valid beforehand, and
        // masked afterwards with an AND: written as "inside ? load : 0" the loads end up in
        // branches, two at a time with a wait for each pair
        unsigned col[CPR], cmask[CPR];
        if (g.wrap) {
            unsigned c = (unsigned)wrap_index(t.gj0 + tid, g.nx);
            const unsigned step = (unsigned)NT % (unsigned)g.nx;
#pragma unroll
            for (int k = 0; k < CPR; ++k) {
                col[k] = c;
                cmask[k] = ~0u;
                c += step;
                c = min(c, c - (unsigned)g.nx);     // (c - nx wraps past c while c < nx)
            }
        } else {
#pragma unroll
            for (int k = 0; k < CPR; ++k) {
                const int lj = t.gj0 - g.gx0 + k * NT + tid;
                const bool ok = lj >= 0 && lj < g.lx;
                cmask[k] = ok ? ~0u : 0u;
                col[k] = ok ? lj : 0;
            }
        }
        const unsigned* cu = reinterpret_cast<const unsigned*>(curv);
        unsigned x[E];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int gi = t.gi0 + 4 * rb + rr;
            const int li = g.wrap ? wrap_index(gi, g.ny) : gi - g.gy0;
            const bool row_ok = g.wrap || (li >= 0 && li < g.ly);
            const size_t ro = (size_t)(row_ok ? li : 0) * g.lx;
            const unsigned* row = cu + ro;
            const unsigned rmask = row_ok ? ~0u : 0u;
            if constexpr (MIX) {
                const float *rA = curv + ro, *rB = pB + ro, *rC = pC + ro;
#pragma unroll
                for (int k = 0; k < CPR; ++k) {
                    const float mixed = m_cc * rA[col[k]] - m_sc2 * rB[col[k]] + m_ss * rC[col[k]];
                    x[rr * CPR + k] = __float_as_uint(mixed) & (rmask & cmask[k]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < CPR; ++k) x[rr * CPR + k] = row[col[k]] & (rmask & cmask[k]);
            }
        }
#pragma unroll
        for (int u = 0; u < E; ++u) v[u] = __uint_as_float(x[u]);
    };
    // A workgroup takes RBW row blocks and the two planes (curv, curv^2) of each, one after
    // the other.  The values of the next step are fetched (for the second plane: fetched
    // again, they come from L2, rather than held in 32 more registers across the transform)
    // as soon as the transform's registers are free, so that they arrive under the stores
    // of this one: left where they are needed, every load was waited for with nothing else
    // to do (12 of a plane's 21 us at T = 2048).
    float va[E], vb[E];
    auto fetch = [&](int rbk) {
        int tid = threadIdx.x;                  // (opaque: addresses are rebuilt per step, see fft_stage)
        asm volatile("" : "+v"(tid));
        if (!SC_DBGBIT(dbg, 8)) { load_tile(ta, va, tid, rbk); load_tile(tb, vb, tid, rbk); }
        else {
#pragma unroll
            for (int u = 0; u < E; ++u) va[u] = vb[u] = 1.f + u;
        }
    };
    fetch(rb0);
    for (int step = 0; step < 2 * RBW; ++step) {
        const int rb = rb0 + (step >> 1), pl = step & 1;
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        __builtin_assume(tid >= 0 && tid < NT);
        if (pl == 0 && !SC_DBGBIT(dbg, 1)) {  // |curv|_2^2 and |curv^2|_2^2 of the tile pair (resolution floor, sc_epi_floor)
            // a thread's 2E cells in float32 (the norms only scale the resolution floor, a
            // 1e-7 relative error there moves nothing), float64 from the wave upwards
            float f2 = 0.f, f4 = 0.f;
#pragma unroll
            for (int u = 0; u < E; ++u) {
                float a2 = va[u] * va[u], b2 = vb[u] * vb[u];
                f2 += a2 + b2;
                f4 = fmaf(a2, a2, fmaf(b2, b2, f4));
            }
            double s2 = f2, s4 = f4;
            for (int sft = 32; sft > 0; sft >>= 1) {
                s2 += __shfl_down(s2, sft, 64);
                s4 += __shfl_down(s4, sft, 64);
            }
            // one pair of partial sums per row block, added up by k_tile_norms: same-address
            // float64 atomics serialise (one pair per workgroup was 20 us of this kernel at
            // T = 512 and at T = 2048 alike) and leave the sum's last bits to the arrival order
            __shared__ double red[2 * (NT / 64)];
            if ((tid & 63) == 0) {
                red[2 * (tid >> 6)] = s2;
                red[2 * (tid >> 6) + 1] = s4;
            }
            lds_barrier();
            if (tid == 0) {
                double t2 = 0.0, t4 = 0.0;
#pragma unroll
                for (int w = 0; w < NT / 64; ++w) {
                    t2 += red[2 * w];
                    t4 += red[2 * w + 1];
                }
                reinterpret_cast<double2*>(norm_part)[(size_t)pair * (Ty >> 2) + rb] = make_double2(t2, t4);
            }
        }
#pragma unroll
        for (int u = 0; u < E; ++u) {
            int e = tid + u * NT;
            sm[lidx<TX>(e / TX, e % TX)] = pl ? make_float2(va[u] * va[u], vb[u] * vb[u])
                                              : make_float2(va[u], vb[u]);
        }
        lds_barrier();
        if (!SC_DBGBIT(dbg, 2)) fft4_lines<TX, false, true>(sm, twr);
#if SC_F1C_HOLD
        // MIX: the second plane (curv^2) is filled from the SAME values - held across the first plane's transform - instead
        // of mixing the three stencil planes again: fetched twice, the 12 B per cell came from memory twice (the workgroups
        // of an XCD have more rows in flight than its L2 holds: FETCH_SIZE 3.5 GB per launch for 1.8 GB of planes)
        if (step + 1 < 2 * RBW && (!MIX || ((step + 1) & 1) == 0)) fetch(rb0 + ((step + 1) >> 1));
#else
        if (step + 1 < 2 * RBW) fetch(rb0 + ((step + 1) >> 1));
#endif
        float2* out = blk + (size_t)(pair * 2 + pl) * plane + (size_t)rb * 4 * TX;
        if (!SC_DBGBIT(dbg, 4))
#pragma unroll 4
        for (int e = 2 * tid; e < 4 * TX; e += 2 * NT) {
            int cb = e >> 4, rr = (e >> 2) & 3, cc = e & 3;
            float2 x0 = sm[lidx<TX>(rr, 4 * cb + cc)], x1 = sm[lidx<TX>(rr, 4 * cb + cc + 1)];
            *reinterpret_cast<float4*>(out + e) = make_float4(x0.x, x0.y, x1.x, x1.y);
        }
        lds_barrier();
    }
}

// |curv|_2^2 and |curv^2|_2^2 of every tile pair from k_fwd_rows_curv's partial sums (nrb per
// pair), in a fixed order: one wave per pair.
__global__ void __launch_bounds__(64)
k_tile_norms(const double* __restrict__ part, int nrb, double* __restrict__ norms) {
    const double2* p = reinterpret_cast<const double2*>(part) + (size_t)blockIdx.x * nrb;
    double s2 = 0.0, s4 = 0.0;
    for (int i = threadIdx.x; i < nrb; i += 64) {
        const double2 v = p[i];
        s2 += v.x;
        s4 += v.y;
    }
    for (int sft = 32; sft > 0; sft >>= 1) {
        s2 += __shfl_down(s2, sft, 64);
        s4 += __shfl_down(s4, sft, 64);
    }
    if (threadIdx.x == 0) {
        norms[2 * blockIdx.x] = s2;
        norms[2 * blockIdx.x + 1] = s4;
    }
}

// Tile rows 4rb .. 4rb+3 hold template rows (tile row r holds template row p = r
// for p >= 0, r - Ty for p < 0)?  The row kernel does not write blocks outside the
// support and the column kernel does not read them: 85 % of a template tile at
// C3 is zero rows.
__device__ __forceinline__ bool templ_rowblock_used(const TemplDev& t, int rb, int Ty) {
    bool any = false;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        int r = 4 * rb + rr;
        int p = (r <= t.pmax) ? r : r - Ty;
        any |= p >= t.pmin && p <= t.pmax;
    }
    return any;
}

// ---- F1t: template tile v = W + iM -> row FFT -> blocked ---------------------
// grid = (Ty/4, n_templates)
template <int TX>
__global__ void __launch_bounds__(fft_threads(TX), fft_waves(TX))
k_fwd_rows_templ(const TemplDev* __restrict__ templ, int first,
                 const float* __restrict__ win_w,
                 const uint8_t* __restrict__ win_m,
                 const double* __restrict__ sums, int Ty,
                 const float2* __restrict__ tw, float2* __restrict__ blk) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    FftTw<TX> twr;
    twr.load(tw);
    constexpr int NT = fft_threads(TX);
    const int rb = blockIdx.x;
    const TemplDev t = templ[first + blockIdx.y];
    const float alpha = sc_fft_alpha(sums, first + blockIdx.y);
    const size_t plane = (size_t)Ty * TX;
    float2* out = blk + (size_t)blockIdx.y * plane + (size_t)rb * 4 * TX;
    // tile row r holds template row p with p = r (p >= 0) or r - Ty (p < 0)
    bool any = false;
    int prow[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        int r = 4 * rb + rr;
        int p = (r <= t.pmax) ? r : r - Ty;
        prow[rr] = (p >= t.pmin && p <= t.pmax) ? p : INT_MIN;
        any |= prow[rr] != INT_MIN;
    }
    if (!any) return;                 // block-uniform: rows outside the support (templ_rowblock_used)
    {
        // all window loads first, then the LDS stores (see k_fwd_cols_tsym)
        constexpr int NF = 4 * TX / NT;
        float w[NF];
        uint8_t m[NF];
#pragma unroll
        for (int u = 0; u < NF; ++u) {
            const int e = threadIdx.x + u * NT;
            const int rr = e / TX, s = e - rr * TX;
            const int q = (s <= t.qmax) ? s : s - TX;
            w[u] = 0.f;
            m[u] = 0;
            if (prow[rr] != INT_MIN && q >= t.qmin && q <= t.qmax) {
                size_t o = (size_t)t.win_off + (size_t)(prow[rr] - t.pmin) * t.ww + (q - t.qmin);
                w[u] = win_w[o];
                m[u] = win_m[o];
            }
        }
#pragma unroll
        for (int u = 0; u < NF; ++u) {
            const int e = threadIdx.x + u * NT;
            const int rr = e / TX, s = e - rr * TX;
            sm[lidx<TX>(rr, s)] = make_float2(alpha * w[u], m[u] ? 1.f : 0.f);
        }
    }
    lds_barrier();
    fft4_lines<TX, false>(sm, twr);
#pragma unroll 4
    for (int e = 2 * threadIdx.x; e < 4 * TX; e += 2 * NT) {
        int cb = e >> 4, rr = (e >> 2) & 3, cc = e & 3;
        float2 x0 = sm[lidx<TX>(rr, 4 * cb + cc)], x1 = sm[lidx<TX>(rr, 4 * cb + cc + 1)];
        *reinterpret_cast<float4*>(out + e) = make_float4(x0.x, x0.y, x1.x, x1.y);
    }
}

// ---- F2: blocked -> column FFT -> spectrum [fx][fy] --------------------------
// grid = (Tx/4, nplanes).  split2: input plane q goes to (q&1 ? out1 : out0)
// at plane index q>>1 (curvature: curv / curv^2), else out0 plane q.
template <int TY>
__global__ void __launch_bounds__(fft_threads(TY), fft_waves(TY))
k_fwd_cols(const float2* __restrict__ blk, int Tx, const float2* __restrict__ tw,
           float2* __restrict__ out0, float2* __restrict__ out1, int split2,
           const TemplDev* __restrict__ templ) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    FftTw<TY> twr;
    twr.load(tw);
    constexpr int NT = fft_threads(TY);
    const int cb = blockIdx.x, q = blockIdx.y;
    const size_t plane = (size_t)TY * Tx;
    const float2* in = blk + (size_t)q * plane;
    const int nbx = Tx >> 2;
    // template planes (templ != nullptr, plane q = template q of the chunk): row
    // blocks outside the template's support were not written - they are zero
    TemplDev t{};
    if (templ) t = templ[q];
    // (all loads of the fill first, then the LDS stores: in one loop the second half of the loads
    //  was issued only once the first half had arrived and been stored)
    constexpr int EP = (4 * TY + 2 * NT - 1) / (2 * NT);
    float4 x[EP];
#pragma unroll
    for (int u = 0; u < EP; ++u) {                                // 2 cells = 16 B per lane
        const int e = 2 * ((int)threadIdx.x + u * NT);
        x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < 4 * TY && (!templ || templ_rowblock_used(t, e >> 4, TY)))
            x[u] = *reinterpret_cast<const float4*>(in + ((size_t)(e >> 4) * nbx + cb) * 16 + (e & 15));
    }
#pragma unroll
    for (int u = 0; u < EP; ++u) {
        const int e = 2 * ((int)threadIdx.x + u * NT);
        if (e >= 4 * TY) continue;
        const int rbk = e >> 4, rr = (e >> 2) & 3, cc = e & 3;
        sm[lidx<TY>(cc, 4 * rbk + rr)] = make_float2(x[u].x, x[u].y);
        sm[lidx<TY>(cc + 1, 4 * rbk + rr)] = make_float2(x[u].z, x[u].w);
    }
    lds_barrier();
    fft4_lines<TY, false, true>(sm, twr);
    float2* out = split2 ? ((q & 1) ? out1 : out0) + (size_t)(q >> 1) * plane
                         : out0 + (size_t)q * plane;
    out += (size_t)cb * 4 * TY;                 // columns 4cb..4cb+3 are contiguous
#pragma unroll 4
    for (int e = 2 * threadIdx.x; e < 4 * TY; e += 2 * NT) {
        int cc = e / TY, fy = e - cc * TY;
        float2 x0 = sm[lidx<TY>(cc, fy)], x1 = sm[lidx<TY>(cc, fy + 1)];
        *reinterpret_cast<float4*>(out + e) = make_float4(x0.x, x0.y, x1.x, x1.y);
    }
}

// ---- split: vh = FFT(alpha*W + iM)  ->  wh = FFT(alpha*W), mh = FFT(M) -------
// FFT(W)[f] = (v[f] + conj v[-f])/2, FFT(M)[f] = (v[f] - conj v[-f])/(2i).
// Done once per template so that the inverse kernels (18 tile pairs per
// template at C3) stream aligned spectra instead of gathering v[-f].  W and M
// are real, so their spectra are Hermitian: only columns fx = 0 .. Tx/2+3 are
// stored (the 4-column blocks up to the one holding Tx/2); I1 rebuilds the rest
// as conj(H[-fy, Tx-fx]) - half the bytes of the dominant HBM stream.
// grid = ((Tx/2+4)*Ty/2/256, n_templates), block = 256, 2 cells per thread.
__host__ __device__ constexpr size_t half_plane(int Ty, int Tx) { return (size_t)(Tx / 2 + 4) * Ty; }

__global__ void __launch_bounds__(256)
k_split_templ(const float2* __restrict__ vh, int Ty, int Tx,
              float2* __restrict__ wh, float2* __restrict__ mh) {
    const size_t plane = (size_t)Ty * Tx, hplane = half_plane(Ty, Tx);
    const float2* v = vh + (size_t)blockIdx.y * plane;
    size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (e >= hplane) return;
    int fx = (int)(e / Ty), fy = (int)(e - (size_t)fx * Ty);         // fy even
    const float2* vm = v + (size_t)((Tx - fx) & (Tx - 1)) * Ty;
    float4 a = *reinterpret_cast<const float4*>(v + e);
    float2 b0 = vm[(Ty - fy) & (Ty - 1)];
    float2 b1 = vm[Ty - fy - 1];
    float4 w = make_float4(0.5f * (a.x + b0.x), 0.5f * (a.y - b0.y),
                           0.5f * (a.z + b1.x), 0.5f * (a.w - b1.y));
    float4 m = make_float4(0.5f * (a.y + b0.y), -0.5f * (a.x - b0.x),
                           0.5f * (a.w + b1.y), -0.5f * (a.z - b1.x));
    *reinterpret_cast<float4*>(wh + (size_t)blockIdx.y * hplane + e) = w;
    *reinterpret_cast<float4*>(mh + (size_t)blockIdx.y * hplane + e) = m;
}

// ---- split, symmetric templates ------------------------------------------------
// Scarp (and its UpperBreak variants) is odd, Ricker is even under the flip of
// the DEM grid about its centre ((n-1)/2 per axis; exact in float64, the grid
// axes are antisymmetric), and M = (W != 0) is even either way.  In the tile the
// template sits wrapped around index 0, so the flip is p -> -p - k per axis with
// k = 1 - n % 2, and
//   FFT(W)[f] = i * a[f] * P[f]   (odd)   or   a[f] * P[f]   (even),
//   FFT(M)[f] =     b[f] * P[f],          P[f] = exp(i pi (ky fy/Ty + kx fx/Tx))
// with a, b REAL.  Only a and b are stored (half plane, 4 bytes per cell): the
// stream I1 reads per template is a quarter of the full complex spectra; the
// phase P is folded into the parked curvature spectrum there.  Taking the real /
// imaginary part also drops the rounding noise of the transform that breaks the
// symmetry.  parity: 1 odd, 2 even.
// P[fy][fx] = phy[fy] * phx[fx]: the two factors are tabulated behind the twiddles of their
// axis (upload_twiddles: entries T .. 2T-1 hold exp(i pi k f / T)).  A sincospif per cell cost
// the template split more instructions than its transforms.
__device__ __forceinline__ float2 phase_tab(const float2* __restrict__ phy, const float2* __restrict__ phx,
                                            int fy, int fx) {
    const float2 a = phy[fy], b = phx[fx];
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__global__ void __launch_bounds__(256)
k_split_templ_sym(const float2* __restrict__ vh, int Ty, int Tx, const float2* __restrict__ phy,
                  const float2* __restrict__ phx, int parity,
                  float* __restrict__ wa, float* __restrict__ mb) {
    const size_t plane = (size_t)Ty * Tx, hplane = half_plane(Ty, Tx);
    const float2* v = vh + (size_t)blockIdx.y * plane;
    size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (e >= hplane) return;
    int fx = (int)(e / Ty), fy = (int)(e - (size_t)fx * Ty);         // fy even
    const float2* vm = v + (size_t)((Tx - fx) & (Tx - 1)) * Ty;
    float4 a = *reinterpret_cast<const float4*>(v + e);
    float2 b0 = vm[(Ty - fy) & (Ty - 1)];
    float2 b1 = vm[Ty - fy - 1];
    float2 w[2] = {make_float2(0.5f * (a.x + b0.x), 0.5f * (a.y - b0.y)),
                   make_float2(0.5f * (a.z + b1.x), 0.5f * (a.w - b1.y))};
    float2 m[2] = {make_float2(0.5f * (a.y + b0.y), -0.5f * (a.x - b0.x)),
                   make_float2(0.5f * (a.w + b1.y), -0.5f * (a.z - b1.x))};
    float ra[2], rb[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float2 ph_ = phase_tab(phy, phx, fy + k, fx);
        // x * conj(P)
        float wr = w[k].x * ph_.x + w[k].y * ph_.y, wi = w[k].y * ph_.x - w[k].x * ph_.y;
        ra[k] = parity == 1 ? wi : wr;
        rb[k] = m[k].x * ph_.x + m[k].y * ph_.y;
    }
    *reinterpret_cast<float2*>(wa + (size_t)blockIdx.y * hplane + e) = make_float2(ra[0], ra[1]);
    *reinterpret_cast<float2*>(mb + (size_t)blockIdx.y * hplane + e) = make_float2(rb[0], rb[1]);
}

// ---- F2 + S fused for symmetric templates ----------------------------------------
// k_fwd_cols followed by k_split_templ_sym writes the full complex spectrum of every
// template (8 B per cell), reads it back and its mirror, and keeps 2 x 4 B per cell of half
// the plane: 86 MB of traffic per 2048^2 template for 17 MB of result.  Large searches
// amortise that over their tile pairs; a one-tile search (BASELINE config C2) spent a third
// of its time there.  Here one workgroup transforms the four mirror columns Tx - fx first,
// keeps the cells it will pair with in registers (16 complex values per thread), transforms
// its own four columns in the same LDS lines and writes the real coefficients a, b straight
// away.  Same transforms, same split arithmetic: the coefficients are bit-identical.
// grid = (Tx/8 + 1, n_templates): column blocks up to the one holding Tx/2.
template <int TY>
__global__ void __launch_bounds__(fft_threads(TY), fft_waves(TY))
k_fwd_cols_tsym(const float2* __restrict__ blk, int Tx, const float2* __restrict__ tw,
                const TemplDev* __restrict__ templ, const float2* __restrict__ phx, int parity,
                float* __restrict__ wa, float* __restrict__ mb) {
    const float2* phy = tw + TY;
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    FftTw<TY> twr;
    twr.load(tw);
    constexpr int NT = fft_threads(TY);
    constexpr int EP = (4 * TY + 2 * NT - 1) / (2 * NT);     // 2-cell pieces per thread
    const int cb = blockIdx.x, q = blockIdx.y;
    const size_t plane = (size_t)TY * Tx, hplane = half_plane(TY, Tx);
    const float2* in = blk + (size_t)q * plane;
    const int nbx = Tx >> 2;
    const TemplDev t = templ[q];
    // ---- mirror columns: line c holds column (Tx - (4cb + c)) mod Tx
    // (all loads of a fill are issued before its first LDS store: a load per loop trip, waited
    //  for before the next, made this kernel a chain of sixteen memory latencies; and the
    //  own columns' loads are issued before the mirror columns' transform, which hides them)
    float4 x[EP];
    {
        constexpr int NM = 4 * TY / NT;
        float2 v[NM];
#pragma unroll
        for (int u = 0; u < NM; ++u) {
            const int e = threadIdx.x + u * NT;
            const int c = e / TY, r = e - c * TY;
            const int col = (Tx - (4 * cb + c)) & (Tx - 1);
            v[u] = make_float2(0.f, 0.f);
            if (templ_rowblock_used(t, r >> 2, TY))
                v[u] = in[((size_t)(r >> 2) * nbx + (col >> 2)) * 16 + (r & 3) * 4 + (col & 3)];
        }
#pragma unroll
        for (int u = 0; u < NM; ++u) {
            const int e = threadIdx.x + u * NT;
            const int c = e / TY, r = e - c * TY;
            sm[lidx<TY>(c, r)] = v[u];
        }
        // own columns 4cb .. 4cb+3 (k_fwd_cols' fill), 2 cells = 16 B per lane
#pragma unroll
        for (int u = 0; u < EP; ++u) {
            const int e = 2 * (threadIdx.x + u * NT);
            x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < 4 * TY && templ_rowblock_used(t, e >> 4, TY))
                x[u] = *reinterpret_cast<const float4*>(in + ((size_t)(e >> 4) * nbx + cb) * 16 + (e & 15));
        }
    }
    lds_barrier();
    fft4_lines<TY, false, true>(sm, twr);
    float2 vm0[EP], vm1[EP];
#pragma unroll
    for (int u = 0; u < EP; ++u) {
        const int e = 2 * (threadIdx.x + u * NT);
        if (e < 4 * TY) {
            const int cc = e / TY, fy = e - cc * TY;          // fy even
            vm0[u] = sm[lidx<TY>(cc, (TY - fy) & (TY - 1))];
            vm1[u] = sm[lidx<TY>(cc, TY - fy - 1)];
        }
    }
    lds_barrier();
#pragma unroll
    for (int u = 0; u < EP; ++u) {
        const int e = 2 * (threadIdx.x + u * NT);
        if (e >= 4 * TY) continue;
        const int rbk = e >> 4, rr = (e >> 2) & 3, cc = e & 3;
        sm[lidx<TY>(cc, 4 * rbk + rr)] = make_float2(x[u].x, x[u].y);
        sm[lidx<TY>(cc + 1, 4 * rbk + rr)] = make_float2(x[u].z, x[u].w);
    }
    lds_barrier();
    fft4_lines<TY, false, true>(sm, twr);
    // ---- split (k_split_templ_sym, cell for cell) and store
#pragma unroll
    for (int u = 0; u < EP; ++u) {
        const int e = 2 * (threadIdx.x + u * NT);
        if (e >= 4 * TY) continue;
        const int cc = e / TY, fy = e - cc * TY, fx = 4 * cb + cc;
        const float2 a0 = sm[lidx<TY>(cc, fy)], a1 = sm[lidx<TY>(cc, fy + 1)];
        const float2 b0 = vm0[u], b1 = vm1[u];
        float2 w[2] = {make_float2(0.5f * (a0.x + b0.x), 0.5f * (a0.y - b0.y)),
                       make_float2(0.5f * (a1.x + b1.x), 0.5f * (a1.y - b1.y))};
        float2 m[2] = {make_float2(0.5f * (a0.y + b0.y), -0.5f * (a0.x - b0.x)),
                       make_float2(0.5f * (a1.y + b1.y), -0.5f * (a1.x - b1.x))};
        float ra[2], rb[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float2 ph_ = phase_tab(phy, phx, fy + k, fx);
            float wr = w[k].x * ph_.x + w[k].y * ph_.y, wi = w[k].y * ph_.x - w[k].x * ph_.y;
            ra[k] = parity == 1 ? wi : wr;
            rb[k] = m[k].x * ph_.x + m[k].y * ph_.y;
        }
        const size_t o = (size_t)q * hplane + (size_t)cb * 4 * TY + e;
        *reinterpret_cast<float2*>(wa + o) = make_float2(ra[0], ra[1]);
        *reinterpret_cast<float2*>(mb + o) = make_float2(rb[0], rb[1]);
    }
}

// ---- I1: spectra product -> inverse column FFT -> blocked --------------------
// grid = (Tx/4): one workgroup per block of 4 columns.  For each plane (W: xcorr,
// M: T3) the block's 4 columns of the curvature spectrum are parked in LDS once
// and reused by all G templates of the launch; per template only FFT(W) or
// FFT(M) is streamed from HBM.  The template loop is software pipelined: the
// spectrum of template g+1 is fetched into registers while template g is
// transformed in LDS (barriers that do not drain vmcnt: lds_barrier), so the HBM
// stream and the LDS/VALU work overlap inside one 512-thread workgroup per CU.
// Only the row pairs [rp_lo, rp_hi] that hold valid outputs are written.
template <int TY>
__host__ __device__ constexpr bool inv_cols_park() { return TY <= 2048; }
template <int TY>
__host__ __device__ constexpr size_t inv_cols_lds() {
    return fft_lds_bytes(TY) + (inv_cols_park<TY>() ? (size_t)4 * TY * sizeof(float2) : 0);
}

template <int TY, bool MIRROR>
__global__ void __launch_bounds__(fft_threads(TY), 2)
k_inv_cols(const float2* __restrict__ uc, const float2* __restrict__ uc2,
           const float2* __restrict__ wh, const float2* __restrict__ mh, int Tx,
           int cb0, int pair, int vfirst, int G, int rp_lo, int rp_hi, int dbg,
           const float2* __restrict__ tw, float2* __restrict__ yw,
           float2* __restrict__ ym, int ystride, int np, int pcj, int tstride,
           const TileDev* __restrict__ tiles, int py_valid) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    FftTw<TY> twr;
    twr.load(tw);
    constexpr int NT = fft_threads(TY);
    constexpr int EP = 4 * TY / (2 * NT);     // float4 (2-cell) loads per thread per stream
    constexpr bool PARK = inv_cols_park<TY>();
    float4* xs = reinterpret_cast<float4*>(sm + 4 * fft_line(TY));   // parked spectrum, linear
    // MIRROR = false: blocks 0 .. Tx/8-1 (fx < Tx/2, spectrum stored);
    // MIRROR = true : blocks Tx/8 .. Tx/4-1 (fx >= Tx/2, rebuilt from the stored
    //                 columns Tx-fx <= Tx/2; Tx/2 mirrors onto itself).
    // Two launches instead of one branchy kernel: the branch cost registers.
    const int cb = cb0 + blockIdx.x;
    // several jobs per launch (grid.y): job j = ob * pcj + q is tile pair `pair + q` of the
    // ob-th orientation of the launch (curvature spectra of an orientation: np planes; its
    // templates start tstride further on); the job's Y block lies j * ystride planes on
    {
        const int ob = blockIdx.y / pcj, q = blockIdx.y - ob * pcj;
        // rows beyond the valid extent of both tiles of the pair (the DEM's last tile row) are
        // never read by the row pass: do not store them (py_valid < 0: circular axis, all rows)
        if (py_valid >= 0) {
            const int vy = max(tiles[2 * (pair + q)].vy, tiles[2 * (pair + q) + 1].vy);
            rp_hi = min(rp_hi, (py_valid + vy - 1) >> 1);
        }
        pair += ob * np + q;
        vfirst += ob * tstride;
    }
    yw += (size_t)blockIdx.y * ystride * ((size_t)TY * Tx);
    ym += (size_t)blockIdx.y * ystride * ((size_t)TY * Tx);
    constexpr bool mirrored = MIRROR;
    const size_t plane = (size_t)TY * Tx;
    const size_t col = (size_t)cb * 4 * TY;
    float4 hreg[EP];
    for (int pl = 0; pl < 2; ++pl) {
        const float4* uu = reinterpret_cast<const float4*>((pl ? uc2 : uc) + (size_t)pair * plane + col);
        // template spectrum: stored for fx <= Tx/2+3; blocks beyond take the
        // mirrored columns Tx-fx (a contiguous 4-column run, not block aligned),
        // reversed in fy and conjugated
        const size_t hplane = half_plane(TY, Tx);
        const float2* hsrc = (pl ? mh : wh) + (size_t)vfirst * hplane +
                             (mirrored ? (size_t)(Tx - 4 * cb - 3) * TY : col);
        // the fetch is the same aligned linear stream for both kinds of block;
        // a mirrored block applies the reversal when it fills the LDS lines
        auto fetch = [&](int gi_) {
            const float2* p = hsrc + (size_t)gi_ * hplane;
#pragma unroll
            for (int u = 0; u < EP; ++u)
                hreg[u] = *reinterpret_cast<const float4*>(p + 2 * (threadIdx.x + u * NT));
        };
        if (!SC_DBGBIT(dbg, 16)) {
            if (PARK) {
#pragma unroll
                for (int u = 0; u < EP; ++u) xs[threadIdx.x + u * NT] = uu[threadIdx.x + u * NT];
            }
            fetch(0);
        }
        if (PARK && mirrored) lds_barrier();      // mirrored fills read other threads' cells of xs
        for (int gi_ = 0; gi_ < G; ++gi_) {
            if (!mirrored) {
#pragma unroll
                for (int u = 0; u < EP; ++u) {
                    int e = 2 * (threadIdx.x + u * NT);
                    int cc = e / TY, fy = e - cc * TY;
                    float4 x = PARK ? xs[threadIdx.x + u * NT] : uu[threadIdx.x + u * NT];
                    sm[lidx<TY>(cc, fy)] = cmul(make_float2(hreg[u].x, hreg[u].y), make_float2(x.x, x.y));
                    sm[lidx<TY>(cc, fy + 1)] = cmul(make_float2(hreg[u].z, hreg[u].w), make_float2(x.z, x.w));
                }
            } else {
                // source cell (column 3-cc of the run, row m) is the conjugate of
                // target cell (column cc, row (TY - m) % TY)
                const float2* x2 = PARK ? reinterpret_cast<const float2*>(xs)
                                        : reinterpret_cast<const float2*>(uu);
#pragma unroll
                for (int u = 0; u < EP; ++u) {
                    int e = 2 * (threadIdx.x + u * NT);
                    int sc = e / TY, m = e - sc * TY;
                    int cc = 3 - sc;
                    int f0 = (TY - m) & (TY - 1), f1 = (TY - m - 1) & (TY - 1);
                    float2 x0 = x2[cc * TY + f0], x1 = x2[cc * TY + f1];
                    sm[lidx<TY>(cc, f0)] = cmul(make_float2(hreg[u].x, -hreg[u].y), x0);
                    sm[lidx<TY>(cc, f1)] = cmul(make_float2(hreg[u].z, -hreg[u].w), x1);
                }
            }
            lds_barrier();
            if (gi_ + 1 < G && !SC_DBGBIT(dbg, 16)) fetch(gi_ + 1);
            if (!SC_DBGBIT(dbg, 32)) fft4_lines<TY, true>(sm, twr);
            // rows2 layout: this block's 4 columns x 2 rows of a row pair are 64
            // contiguous bytes; a thread stores (row 2rp, row 2rp+1) of one column
            float2* o = (pl ? ym : yw) + (size_t)gi_ * plane + (size_t)(cb >> 1) * 16 + (cb & 1) * 8;
            const int e_lo = 4 * rp_lo, e_hi = 4 * (rp_hi + 1);
            if (!SC_DBGBIT(dbg, 64))
#pragma unroll 2
            for (int e = e_lo + threadIdx.x; e < e_hi; e += NT) {
                int rp = e >> 2, k = e & 3;
                float2 x0 = sm[lidx<TY>(k, 2 * rp)], x1 = sm[lidx<TY>(k, 2 * rp + 1)];
                store_stream(o + (size_t)rp * (Tx >> 3) * 16 + 2 * k, x0, x1);
            }
            lds_barrier();
        }
    }
}

// ---- I1 for symmetric templates -----------------------------------------------
// Same structure as k_inv_cols; the template stream is the real coefficient a
// (W plane) or b (M plane) of k_split_templ_sym, and the phase - with the factor
// i of an odd W - is multiplied into the curvature spectrum when it is parked:
//   direct   blocks: Y[f]  = a[f]  * (X[f]  * P[f]  * {i})
//   mirrored blocks: Y[ft] = a[fs] * (X[ft] * conj(P[fs]) * {-i}),  fs = -ft mod T
// (FFT(W)[ft] = conj(FFT(W)[fs]); conj(P[fs]) is not P[ft]: the half-cell phase
// has period 2T.)
// PT ("paired templates"): the launch serves a tile pair whose second tile is empty
// (odd tile count, single-tile DEMs).  Instead of leaving the imaginary half of
// the transform idle, two TEMPLATES ride in it: Y = IFFT(X (a_g + i a_g+1) P), whose
// real part is template g's result and whose imaginary part is template g+1's
// (X is the spectrum of ONE real tile).  Plane k of Y holds templates 2k, 2k+1.
// XP ("paired orientations", with PT): searches with ONE template per orientation would leave
// that imaginary half idle all the same.  There two ORIENTATIONS ride together: job j serves
// orientations 2j and 2j+1 of the batch, Y = IFFT((X_2j a_2j + i X_2j+1 a_2j+1) P) - each
// product is the spectrum of a real plane, so the real part is orientation 2j's result and the
// imaginary part orientation 2j+1's.  Two spectra are parked; `tstride` carries the batch's
// orientation count, `ystride` is 1 (plane j of Y), one tile pair per launch.
// A column-pass launch whose workgroups do not fill the chip (a small DEM with many templates per
// orientation: 900 x 505 at 35 ages is 384 two-wave workgroups for 1 024 SIMDs, each a chain of 70
// transforms) deals its transforms out along grid.z: part z of nz takes transforms
// [NG z / nz, NG (z + 1) / nz) of the launch - its coefficient planes, its planes of Y.  Every part parks
// the spectrum for itself; the transforms are the same instructions on the same operands: Y is
// bit-identical whatever nz.
struct TemplShare { int g0, g, t0; };      // first template, templates, first transform of this workgroup's part
template <bool PT>
__device__ __forceinline__ TemplShare template_share(int G) {
    const int nz = gridDim.z;
    if (nz <= 1) return {0, G, 0};
    const int NG = PT ? (G + 1) / 2 : G;
    const int t0 = (int)(((long long)NG * blockIdx.z) / nz), t1 = (int)(((long long)NG * (blockIdx.z + 1)) / nz);
    const int g0 = PT ? 2 * t0 : t0, g1 = PT ? min(G, 2 * t1) : t1;
    return {g0, g1 - g0, t0};
}
#define TAKE_TEMPLATE_SHARE(PTV, plane_)                                       \
    {                                                                          \
        const TemplShare sh_ = template_share<PTV>(G);                         \
        vfirst += sh_.g0; G = sh_.g;                                           \
        yw += (size_t)sh_.t0 * (plane_); ym += (size_t)sh_.t0 * (plane_);      \
    }

template <int TY, bool MIRROR, bool PT, bool XP = false>
__device__ __forceinline__ void
inv_cols_sym_body(const int cbx, const float2* __restrict__ uc, const float2* __restrict__ uc2,
               const float* __restrict__ wa, const float* __restrict__ mb, int Tx,
               int cb0, int pair, int vfirst, int G, int rp_lo, int rp_hi, const float2* __restrict__ phx,
               int parity, const float2* __restrict__ tw, float2* __restrict__ yw,
               float2* __restrict__ ym, int ystride, int dbg, int np, int pcj, int tstride,
               const TileDev* __restrict__ tiles, int py_valid) {
    // dbg: timing-only ablation bits of an SC_ABLATE build (tools/ablate.sh), folded away otherwise:
    //   1 no coefficient fetch in mirrored launches   2 no coefficient fetch at all   4 no stores
    //   8 no transform   16 stores paired into whole 128-byte lines (a bijection onto the same plane)
    //   32 plain stores   64 sc1 stores   128 one contiguous run per workgroup and template   256 sc0 sc1 stores
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    static_assert(inv_cols_park<TY>(), "symmetric I1 parks the spectrum");
    FftTw<TY> twr;
    twr.load(tw);
    constexpr int NT = fft_threads(TY);
    constexpr int EP = 4 * TY / (2 * NT);     // 2-cell loads per thread per stream
    static_assert(!XP || PT, "paired orientations are a paired-template mode");
    float4* xs = reinterpret_cast<float4*>(sm + 4 * fft_line(TY));   // parked spectrum, linear
    float4* xsB = xs + 2 * TY;                                       // XP: the second orientation's
    const int cb = cb0 + cbx;
    int pairB = 0;
    // several jobs per launch (grid.y), see k_inv_cols
    if constexpr (XP) {
        const int oA = 2 * (int)blockIdx.y;                          // the job's first orientation
        G = min(2, tstride - oA);                                    // templates of the job (1: no partner)
        if (py_valid >= 0) {
            const int vy = max(tiles[2 * pair].vy, tiles[2 * pair + 1].vy);
            rp_hi = min(rp_hi, (py_valid + vy - 1) >> 1);
        }
        pair += oA * np;
        pairB = pair + (G > 1 ? np : 0);
        vfirst += oA;
    } else {
        const int ob = blockIdx.y / pcj, q = blockIdx.y - ob * pcj;
        // rows beyond the valid extent of both tiles of the pair (the DEM's last tile row) are
        // never read by the row pass: do not store them (py_valid < 0: circular axis, all rows)
        if (py_valid >= 0) {
            const int vy = max(tiles[2 * (pair + q)].vy, tiles[2 * (pair + q) + 1].vy);
            rp_hi = min(rp_hi, (py_valid + vy - 1) >> 1);
        }
        pair += ob * np + q;
        vfirst += ob * tstride;
    }
    yw += (size_t)blockIdx.y * ystride * ((size_t)TY * Tx);
    ym += (size_t)blockIdx.y * ystride * ((size_t)TY * Tx);
    constexpr bool mirrored = MIRROR;
    const size_t plane = (size_t)TY * Tx;
    const size_t col = (size_t)cb * 4 * TY;
    const size_t hplane = half_plane(TY, Tx);
    float2 hreg[EP], hreg2[PT ? EP : 1];
    const int NG = PT ? (G + 1) / 2 : G;       // inverse transforms of the launch
    for (int pl = 0; pl < 2; ++pl) {
        const float4* uu = reinterpret_cast<const float4*>((pl ? uc2 : uc) + (size_t)pair * plane + col);
        const float* hsrc = (pl ? mb : wa) + (size_t)vfirst * hplane +
                            (mirrored ? (size_t)(Tx - 4 * cb - 3) * TY : col);
        auto fetch = [&](int gi_) {
            const float* p = hsrc + (size_t)(PT ? 2 * gi_ : gi_) * hplane;
#pragma unroll
            for (int u = 0; u < EP; ++u)
                hreg[u] = *reinterpret_cast<const float2*>(p + 2 * (threadIdx.x + u * NT));
            if constexpr (PT) {
                // (an odd template count: the last transform's second plane is zero.  Loaded
                //  unconditionally - from the first plane again - and masked: "has2 ? load : 0"
                //  put the loads in a branch and a wait for them right behind it, i.e. no prefetch)
                const bool has2 = 2 * gi_ + 1 < G;
                const uint2* p2 = reinterpret_cast<const uint2*>(has2 ? p + hplane : p);
                const unsigned keep = has2 ? ~0u : 0u;
#pragma unroll
                for (int u = 0; u < EP; ++u) {
                    const uint2 r = p2[threadIdx.x + u * NT];
                    hreg2[u] = make_float2(__uint_as_float(r.x & keep), __uint_as_float(r.y & keep));
                }
            }
        };
        // cell value: x * a (one template), x * (a + i a2) (two templates) or x a + i xb a2 (two orientations)
        auto prod = [&](float2 x, float2 xb, float a, float a2) {
            if constexpr (XP) return make_float2(x.x * a - xb.y * a2, x.y * a + xb.x * a2);
            return PT ? make_float2(x.x * a - x.y * a2, x.x * a2 + x.y * a) : make_float2(a * x.x, a * x.y);
        };
        const float4* uuB = reinterpret_cast<const float4*>((pl ? uc2 : uc) + (size_t)pairB * plane + col);
        const bool rot = pl == 0 && parity == 1;          // odd W: factor i (direct) / -i (mirrored)
        // (the spectrum's loads all first: fetched where they are used, one instantiation of this
        //  loop waited for each of them in turn - eight latencies per plane, a quarter of a
        //  fifteen-template launch at T = 512)
        // order of issue: the phase factors first (table entries, back quickly), then the spectrum -
        // waiting for a phase factor then does not wait for the spectrum loads behind it
        float2 pvv[EP][2];
#pragma unroll
        for (int u = 0; u < EP; ++u) {
            const int e = 2 * (threadIdx.x + u * NT);
            const int cc = e / TY, fy = e - cc * TY, fx = 4 * cb + cc;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (!mirrored)
                    pvv[u][k] = phase_tab(tw + TY, phx, fy + k, fx);
                else {
                    pvv[u][k] = phase_tab(tw + TY, phx, (TY - fy - k) & (TY - 1), (Tx - fx) & (Tx - 1));
                    pvv[u][k].y = -pvv[u][k].y;
                }
            }
        }
        asm volatile("" ::: "memory");
        float4 xall[EP], xallB[XP ? EP : 1];
#pragma unroll
        for (int u = 0; u < EP; ++u) xall[u] = uu[threadIdx.x + u * NT];
        if constexpr (XP) {
#pragma unroll
            for (int u = 0; u < EP; ++u) xallB[u] = uuB[threadIdx.x + u * NT];
        }
        auto park = [&](const float4 x, const float2 (&pv)[2], float4* dst, int u) {
            float2 xv[2] = {make_float2(x.x, x.y), make_float2(x.z, x.w)};
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float2 v = cmul(xv[k], pv[k]);
                if (rot) v = mirrored ? make_float2(v.y, -v.x) : make_float2(-v.y, v.x);
                xv[k] = v;
            }
            dst[threadIdx.x + u * NT] = make_float4(xv[0].x, xv[0].y, xv[1].x, xv[1].y);
        };
#pragma unroll
        for (int u = 0; u < EP; ++u) park(xall[u], pvv[u], xs, u);
        if constexpr (XP) {
#pragma unroll
            for (int u = 0; u < EP; ++u) park(xallB[u], pvv[u], xsB, u);
        }
        if (SC_DBGBIT(dbg, 2) || (mirrored && SC_DBGBIT(dbg, 1))) {
#pragma unroll
            for (int u = 0; u < EP; ++u) hreg[u] = make_float2(1.f + u, 2.f);
        } else
        fetch(0);
        if (mirrored) lds_barrier();      // mirrored fills read other threads' cells of xs
        for (int gi_ = 0; gi_ < NG; ++gi_) {
            if (!mirrored) {
#pragma unroll
                for (int u = 0; u < EP; ++u) {
                    int e = 2 * (threadIdx.x + u * NT);
                    int cc = e / TY, fy = e - cc * TY;
                    const float4 x = xs[threadIdx.x + u * NT], xb = XP ? xsB[threadIdx.x + u * NT] : x;
                    sm[lidx<TY>(cc, fy)] = prod(make_float2(x.x, x.y), make_float2(xb.x, xb.y), hreg[u].x, hreg2[PT ? u : 0].x);
                    sm[lidx<TY>(cc, fy + 1)] = prod(make_float2(x.z, x.w), make_float2(xb.z, xb.w), hreg[u].y, hreg2[PT ? u : 0].y);
                }
            } else {
                // source cell (column 3-cc of the run, row m) pairs with target cell
                // (column cc, row (TY - m) % TY)
                const float2* x2 = reinterpret_cast<const float2*>(xs);
                const float2* x2B = reinterpret_cast<const float2*>(XP ? xsB : xs);
#pragma unroll
                for (int u = 0; u < EP; ++u) {
                    int e = 2 * (threadIdx.x + u * NT);
                    int sc = e / TY, m = e - sc * TY;
                    int cc = 3 - sc;
                    int f0 = (TY - m) & (TY - 1), f1 = (TY - m - 1) & (TY - 1);
                    float2 x0 = x2[cc * TY + f0], x1 = x2[cc * TY + f1];
                    sm[lidx<TY>(cc, f0)] = prod(x0, x2B[cc * TY + f0], hreg[u].x, hreg2[PT ? u : 0].x);
                    sm[lidx<TY>(cc, f1)] = prod(x1, x2B[cc * TY + f1], hreg[u].y, hreg2[PT ? u : 0].y);
                }
            }
            lds_barrier();
            if (gi_ + 1 < NG && !SC_DBGBIT(dbg, 2) && !(mirrored && SC_DBGBIT(dbg, 1))) fetch(gi_ + 1);
            if (!SC_DBGBIT(dbg, 8)) fft4_lines<TY, true>(sm, twr);
            float2* o = (pl ? ym : yw) + (size_t)gi_ * plane + (size_t)(cb >> 1) * 16 + (cb & 1) * 8;
            const int e_lo = 4 * rp_lo, e_hi = 4 * (rp_hi + 1);
            if (!SC_DBGBIT(dbg, 4))
#pragma unroll 2
            for (int e = e_lo + threadIdx.x; e < e_hi; e += NT) {
                int rp = e >> 2, k = e & 3;
                float2 x0 = sm[lidx<TY>(k, 2 * rp)], x1 = sm[lidx<TY>(k, 2 * rp + 1)];
#ifdef SC_ABLATE
                if (dbg & (16 | 32 | 64 | 128 | 256)) {
                    float2* dst = o + (size_t)rp * (Tx >> 3) * 16 + 2 * k;
                    if (dbg & 16)        // row pairs (2q, 2q+1) of this block -> one whole line in row pair 2q + (cb & 1)
                        dst = (pl ? ym : yw) + (size_t)gi_ * plane + (size_t)(cb >> 1) * 16 +
                              (size_t)((rp & ~1) + (cb & 1)) * (Tx >> 3) * 16 + (rp & 1) * 8 + 2 * k;
                    if (dbg & 128)       // the workgroup's cells of this template as one contiguous run
                        dst = (pl ? ym : yw) + (size_t)gi_ * plane + (size_t)cb * (TY * 4) + 2 * (size_t)e;
                    store_flavour(dst, x0, x1, (dbg & 32) ? 1 : (dbg & 64) ? 2 : (dbg & 256) ? 3 : 0);
                    continue;
                }
#endif
                store_stream(o + (size_t)rp * (Tx >> 3) * 16 + 2 * k, x0, x1);
            }
            lds_barrier();
        }
    }
}


template <int TY, bool MIRROR, bool PT>
__global__ void __launch_bounds__(fft_threads(TY), 2)
k_inv_cols_sym(const float2* __restrict__ uc, const float2* __restrict__ uc2,
               const float* __restrict__ wa, const float* __restrict__ mb, int Tx,
               int cb0, int pair, int vfirst, int G, int rp_lo, int rp_hi, const float2* __restrict__ phx,
               int parity, const float2* __restrict__ tw, float2* __restrict__ yw,
               float2* __restrict__ ym, int ystride, int dbg, int np, int pcj, int tstride,
               const TileDev* __restrict__ tiles, int py_valid) {
    inv_cols_sym_body<TY, MIRROR, PT>((int)blockIdx.x, uc, uc2, wa, mb, Tx, cb0, pair, vfirst, G, rp_lo, rp_hi, phx,
                                      parity, tw, yw, ym, ystride, dbg, np, pcj, tstride, tiles, py_valid);
}

// Both halves in ONE launch, a column block and the mirror block that streams (three of four
// columns of) the same coefficients eight workgroup ids apart: the same XCD, started together,
// so that the second read of a coefficient line hits that XCD's L2 instead of going to HBM
// again (k_inv_rows_fast pairs its sibling rows the same way).  Workgroup j: group j / 16,
// kind (j / 8) & 1, index i = 8 (j / 16) + j % 8; kind 0 is column block i, kind 1 the mirror
// block Tx/4 - 1 - i, whose coefficient columns are 4i+1 .. 4i+4.  Measured on the sustained
// C3 run: 728 us per tile pair against 745 (a kernel with block AND mirror in one workgroup,
// since removed) and 2 x 374 (two launches); C2,
// paired templates: 985 against 2 x 557.  (Giving every XCD a CONTIGUOUS range of blocks, so
// that the fourth column is shared too, is 13 % slower: each XCD then writes a 2-KB stripe of
// every row and loads a few L2 channels only.)  profiles/r02_i1_xcd_paired.txt
template <int TY, bool PT, bool XP = false>
__global__ void __launch_bounds__(fft_threads(TY), 2)      // (XP spills 32 B at 256 registers; uncapped - 274 - it loses a wave per SIMD: C5 1.65 -> 1.87 ms)
k_inv_cols_symx(const float2* __restrict__ uc, const float2* __restrict__ uc2,
                const float* __restrict__ wa, const float* __restrict__ mb, int Tx,
                int pair, int vfirst, int G, int rp_lo, int rp_hi, const float2* __restrict__ phx,
                int parity, const float2* __restrict__ tw, float2* __restrict__ yw,
                float2* __restrict__ ym, int ystride, int dbg, int np, int pcj, int tstride,
                const TileDev* __restrict__ tiles, int py_valid) {
    const int j = blockIdx.x, i = ((j >> 4) << 3) | (j & 7);
    if constexpr (!XP) TAKE_TEMPLATE_SHARE(PT, (size_t)TY * Tx)
    if ((j >> 3) & 1)
        inv_cols_sym_body<TY, true, PT, XP>((Tx >> 2) - 1 - i, uc, uc2, wa, mb, Tx, 0, pair, vfirst, G, rp_lo, rp_hi, phx,
                                            parity, tw, yw, ym, ystride, dbg, np, pcj, tstride, tiles, py_valid);
    else
        inv_cols_sym_body<TY, false, PT, XP>(i, uc, uc2, wa, mb, Tx, 0, pair, vfirst, G, rp_lo, rp_hi, phx,
                                             parity, tw, yw, ym, ystride, dbg, np, pcj, tstride, tiles, py_valid);
}

// ---- I1 for symmetric templates, one WAVE per column ----------------------------------
// The kernels above transform four columns with 512 threads: every thread owns one 16-point
// set per stage, and the eight waves meet at a workgroup barrier between reading and writing
// every stage (seven barriers per template and plane) - LDS passes and butterflies of the
// whole workgroup alternate instead of overlapping, and the transforms alone cost 4 us per
// four columns where the LDS traffic is worth 1.8 (ISA count: 336 packed VALU and 90 LDS
// instructions per wave and template).  A column's transform only needs the threads working
// on that column to agree.  Here a workgroup takes EIGHT columns and each of its eight waves
// owns one: two sets per lane and stage, ordered by the wave's own instruction stream (LDS
// executes a wave's instructions in order) - no barrier inside a transform, the waves drift
// apart and one wave's LDS pass runs under another's butterflies.
//  * The lane's cells of its column are fy = lane + 64 k, k = 0 .. 31 - exactly the inputs of
//    its two stage-1 sets (set tt reads tt + 128 j): the coefficient products go straight into
//    the first butterflies, no fill pass through LDS.  The phase-multiplied curvature spectrum
//    of those cells is parked in 64 registers (LDS holds the eight lines and nothing else).
//  * Two workgroup barriers per template and plane remain: lines complete -> store pass
//    (a wave stores whole 128-byte lines: 8 columns x 2 rows, the rows2 block I2 reads),
//    and store pass done -> next stage-1 writes.
//  * Same butterflies, twiddles and operand order as fft4_lines: Y is bit-identical.
// Column blocks and mirror blocks are paired per XCD as in k_inv_cols_symx.
template <int TY>
__host__ __device__ constexpr int w8_line() { return TY + TY / 16 + 4; }   // lines 8 banks apart: the store pass reads 8 lines x 2 cells
template <int TY>
__host__ __device__ constexpr size_t w8_lds() {          // eight lines + the twiddle bases of stages 1 and 2
    return ((size_t)8 * w8_line<TY>() + (SC_I1_TWTAB ? 16 : 4) * (TY / 16 + TY / 256) + SC_MAX_GROUP) * sizeof(float2);   // + the stored row range per transform
}

template <int TY, bool MIRROR, bool PT, int NC>
__device__ __forceinline__ void
inv_cols_w8_body(const int B, const int jobx, const float2* __restrict__ uc, const float2* __restrict__ uc2,
                 const float* __restrict__ wa, const float* __restrict__ mb, int Tx,
                 int pair, int vfirst, int G, int rp_lo, int rp_hi, const float2* __restrict__ phx,
                 int parity, const float2* __restrict__ tw, float2* __restrict__ yw,
                 float2* __restrict__ ym, int ystride, int np, int pcj, int tstride,
                 const TileDev* __restrict__ tiles, int py_valid, const TemplDev* __restrict__ tl) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    constexpr int S = TY / 16;                 // 16-point sets per line
    static_assert(S % 64 == 0 && S / 64 <= 2, "one wave per line: 64 or 128 sets");
    constexpr int U = S / 64;                  // sets per lane and stage
    constexpr int NK = 16 * U;                 // cells per lane
    constexpr int LINE = w8_line<TY>();
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float2* line = sm + w * LINE;
    // rows the templates' window limits mask out (WindowedTemplate.py:66-84: up to 350 rows at either
    // end of the DEM at scale 100, 2.4 % of all rows on average over the 35 x 181 grid) are not stored
    // either: the row pass scores nothing there (its range test fails for every cell of such a row,
    // whatever the transform of the stale cells says).  i0A / i0B: first global row of the pair's tiles.
    int i0A = 0, i0B = 0, vyA = 0, vyB = 0;
    {
        const int ob = jobx / pcj, q = jobx - ob * pcj;
        const TileDev tA_ = tiles[2 * (pair + q)], tB_ = tiles[2 * (pair + q) + 1];
        i0A = tA_.i0; i0B = tB_.i0; vyA = tA_.vy; vyB = tB_.vy;
        if (py_valid >= 0) {
            const int vy = max(tA_.vy, tB_.vy);
            rp_hi = min(rp_hi, (py_valid + vy - 1) >> 1);
        }
        pair += ob * np + q;
        vfirst += ob * tstride;
    }
    // [lo, hi] of the row pairs transform gi_ has to store: rp_lo .. rp_hi cut to the rows some
    // template of the transform keeps on some tile of the pair (tile row ri = 2 rp + {0, 1} - py_valid
    // is global row i0 + ri); circular axes (py_valid < 0) and launches without descriptors: all.
    // Worked out ONCE, by thread gi_, before the first coefficient is fetched, and parked in LDS behind
    // the twiddle tables: read from the descriptors inside the transform loop, the loads made the
    // compiler lose count of the coefficient prefetch in flight - every use of a prefetched value
    // then waited for ALL outstanding loads (s_waitcnt vmcnt(0): 70 of them in the four-wave
    // paired-template kernel instead of 8, 1 416 us per C2 launch instead of 878).
    int2* const rng = reinterpret_cast<int2*>(sm + NC * LINE + (SC_I1_TWTAB ? 16 : 4) * (S + S / 16));
    {
        const int NGt = PT ? (G + 1) / 2 : G;
        const int gi_ = threadIdx.x;
        if (gi_ < NGt) {
            int lo = rp_lo, hi = rp_hi;
            if (tl && py_valid >= 0) {
                int klo = INT_MAX, khi = INT_MIN;
#pragma unroll
                for (int k = 0; k < (PT ? 2 : 1); ++k) {
                    const int ti = vfirst + (PT ? 2 * gi_ + k : gi_);
                    if (PT && k == 1 && 2 * gi_ + 1 >= G) break;
                    const int ilo = tl[ti].ilo, ihi = tl[ti].ihi;
                    if (vyA > 0) { klo = min(klo, ilo - i0A); khi = max(khi, ihi - i0A); }
                    if (vyB > 0) { klo = min(klo, ilo - i0B); khi = max(khi, ihi - i0B); }
                }
                if (khi < klo) { lo = 1; hi = 0; }
                else {
                    lo = max(lo, (max(klo, 0) + py_valid) >> 1);
                    hi = min(hi, (min(khi, TY) + py_valid) >> 1);
                }
            }
            rng[gi_] = make_int2(lo, hi);
        }
    }
    const size_t plane = (size_t)TY * Tx, hplane = half_plane(TY, Tx);
    yw += (size_t)jobx * ystride * plane;
    ym += (size_t)jobx * ystride * plane;
    const int ft = NC * B + w;                 // this wave's column of Y (NC columns, NC waves per workgroup)
    const int fs = MIRROR ? Tx - ft : ft;      // the coefficient column it pairs with
    // twiddle bases of the sets (FftTw's, per set: stage 1 w^tt, stage 2 w^(16 (tt >> 4)), times
    // 1, 2, 4, 8) in an LDS table behind the lines, [m][tt] and [m][tt >> 4]: 64 registers of
    // parked spectrum leave no room for them
    float2* t1 = sm + NC * LINE;
#if SC_I1_TWTAB
    // all fifteen twiddles per set, [k][tt] and [k][tt >> 4] (entry k = 0 unused): forming w^3, w^5 .. w^15 from the
    // four bases in every transform was 88 of its 640 packed instructions.  The same products in the same order
    // (twiddle16_expand): the same bits.
    float2* t2 = t1 + 16 * S;
    for (int i = threadIdx.x; i < S + S / 16; i += 64 * NC) {
        const bool st2 = i >= S;
        const int e = st2 ? (i - S) << 4 : i, n = st2 ? S / 16 : S;
        float2* t = st2 ? t2 + (i - S) : t1 + i;
        const float2 wb[4] = {tw[e], tw[2 * e], tw[4 * e], tw[8 * e]};
        pk::v2 wk[16];
        twiddle16_expand(wb, wk);
#pragma unroll
        for (int k = 1; k < 16; ++k) t[k * n] = make_float2(wk[k].x, wk[k].y);
    }
#else
    float2* t2 = t1 + 4 * S;
    for (int i = threadIdx.x; i < 4 * S; i += 64 * NC) t1[i] = tw[(i % S) << (i / S)];
    for (int i = threadIdx.x; i < 4 * (S / 16); i += 64 * NC) t2[i] = tw[((i % (S / 16)) << 4) << (i / (S / 16))];
#endif
    auto tw_of = [&](const float2* t, int n, int idx, float2 (&wq)[4]) {
#pragma unroll
        for (int m = 0; m < 4; ++m) wq[m] = lds_cell<true>(t + m * n + idx);
    };
    float c[NK], c2[PT ? NK : 1];
    const int NG = PT ? (G + 1) / 2 : G;       // inverse transforms per plane (PT: templates 2k, 2k+1 in one)
    for (int pl = 0; pl < 2; ++pl) {
        const float2* xcol = (pl ? uc2 : uc) + (size_t)pair * plane + (size_t)ft * TY;
        const float* hsrc = (pl ? mb : wa) + (size_t)vfirst * hplane + (size_t)fs * TY;
        const bool rot = pl == 0 && parity == 1;          // odd W: factor i (own columns) / -i (mirrors)
        // ---- park X P {i} of the lane's cells (the arithmetic of k_inv_cols_sym, cell for cell)
        float2 xp[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int fy = lane + 64 * k;
            float2 pv;
            if (!MIRROR) {
                pv = phase_tab(tw + TY, phx, fy, ft);
            } else {
                pv = phase_tab(tw + TY, phx, (TY - fy) & (TY - 1), (Tx - ft) & (Tx - 1));
                pv.y = -pv.y;
            }
            float2 v = cmul(xcol[fy], pv);
            if (rot) v = MIRROR ? make_float2(v.y, -v.x) : make_float2(-v.y, v.x);
            xp[k] = v;
        }
        // coefficient of cell fy: a[fs][fy], mirrors a[fs][-fy mod TY] = a[fs][TY - lane - 64 k] but for
        // fy = 0 - one base per lane and compile-time offsets either way
        const float* cbase = hsrc + (MIRROR ? TY - lane : lane);
        const int c0off = (MIRROR && lane == 0) ? -TY : 0;
        // the cells of the lane's set u (k = u mod U) of transform gi_; u < 0: of all its sets
        auto fetch = [&](int gi_, int u) {
            const float* p = cbase + (size_t)(PT ? 2 * gi_ : gi_) * hplane;
            const bool has2 = PT && 2 * gi_ + 1 < G;
            const float* p2 = has2 ? p + hplane : p;
            const unsigned keep2 = has2 ? ~0u : 0u;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                if (u >= 0 && k % U != u) continue;
                const int off = k ? (MIRROR ? -64 * k : 64 * k) : c0off;
                c[k] = p[off];
                if constexpr (PT) c2[k] = __uint_as_float(__float_as_uint(p2[off]) & keep2);   // (see inv_cols_sym_body's fetch)
            }
        };
        fetch(0, -1);
        for (int gi_ = 0; gi_ < NG; ++gi_) {
            lds_barrier();                                   // the store pass of the previous transform is done with the lines
            float2 wq[4];
            // (LDS addresses rebuilt from the lane id in every transform, see the store pass)
            int lt = lane;
            asm volatile("" : "+v"(lt));
            // stage 1 set by set, straight from the products: cell k = u + U j is input j of set u
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float2 a1[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int k = u + U * j;
                    // x * a (one template) or x * (a + i a2) (two templates, see k_inv_cols_sym)
                    a1[j] = PT ? make_float2(xp[k].x * c[k] - xp[k].y * c2[k], xp[k].x * c2[k] + xp[k].y * c[k])
                               : make_float2(c[k] * xp[k].x, c[k] * xp[k].y);
                }
#if SC_I1_TWTAB
                set_compute_store<TY, 16, 0, true, true>(line, lt + 64 * u, a1, wq, t1 + lt + 64 * u, S);
#else
                tw_of(t1, S, lt + 64 * u, wq);
                set_compute_store<TY, 16, 0, true, false, true>(line, lt + 64 * u, a1, wq);
#endif
            }
            // next coefficients: all of them now - or, two planes of them (PT) and two sets per lane,
            // the second set's only once the transform's registers are free again
            constexpr bool SPLIT = PT && U > 1 && NC == 8;   // (the four-wave form has the registers)
            if (gi_ + 1 < NG) fetch(gi_ + 1, SPLIT ? 0 : -1);
            asm volatile("" ::: "memory");
            float2 a[U][16];
#pragma unroll
            for (int u = 0; u < U; ++u) set_load<TY, true>(line, lt + 64 * u, a[u]);
            constexpr bool XLANE = SC_I1_XLANE && (TY == 2048 || TY == 1024);
            if constexpr (XLANE) {
                // stage 2 into registers, the exchange across lanes, stage 3 from registers (see xlane_transpose4)
                float2 o2[U][16];
#pragma unroll
                for (int u = 0; u < U; ++u) {
#if SC_I1_TWTAB
                    set_compute_regs<true, true>(a[u], wq, o2[u], t2 + ((lt + 64 * u) >> 4), S / 16);
#else
                    tw_of(t2, S / 16, (lt + 64 * u) >> 4, wq);
                    set_compute_regs<true>(a[u], wq, o2[u]);
#endif
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int mh = 0; mh < 4; ++mh) {
                        xlane_transpose4(o2[u][4 * mh].x, o2[u][4 * mh + 1].x, o2[u][4 * mh + 2].x, o2[u][4 * mh + 3].x);
                        xlane_transpose4(o2[u][4 * mh].y, o2[u][4 * mh + 1].y, o2[u][4 * mh + 2].y, o2[u][4 * mh + 3].y);
                    }
                constexpr int R3x = TY / 256, NB3 = 16 / R3x;      // stage 3: radix 8 / 4, two / four butterflies per set
#pragma unroll
                for (int u3 = 0; u3 < U; ++u3) {
                    // butterfly b of set u3 takes a[b + NB3 j] = element (lt + 64 u3 + S b) + 256 j: set j >> 2, output
                    // 4 (u3 + U b) + (lane row), of the lane in row j & 3
#pragma unroll
                    for (int b = 0; b < NB3; ++b)
#pragma unroll
                        for (int j = 0; j < R3x; ++j) a[u3][b + NB3 * j] = o2[j >> 2][4 * (u3 + U * b) + (j & 3)];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) set_compute_store<TY, TY / 256, 8, true, false, true>(line, lt + 64 * u, a[u], wq);
            } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
#if SC_I1_TWTAB
                set_compute_store<TY, 16, 4, true, true>(line, lt + 64 * u, a[u], wq, t2 + ((lt + 64 * u) >> 4), S / 16);
#else
                tw_of(t2, S / 16, (lt + 64 * u) >> 4, wq);
                set_compute_store<TY, 16, 4, true, false, true>(line, lt + 64 * u, a[u], wq);
#endif
            }
            asm volatile("" ::: "memory");
            }
            if constexpr (TY > 256 && !XLANE) {
                constexpr int R3 = TY / 256;
#pragma unroll
                for (int u = 0; u < U; ++u) set_load<TY, true>(line, lt + 64 * u, a[u]);
#pragma unroll
                for (int u = 0; u < U; ++u) set_compute_store<TY, R3, 8, true, false, true>(line, lt + 64 * u, a[u], wq);
            }
            lds_barrier();                                   // all eight lines are complete
            if (SPLIT && gi_ + 1 < NG) fetch(gi_ + 1, 1);
            // ---- store: lane (q, c) of wave w takes column c of row pair rp_lo + 8 w + 64 it + q:
            // eight lanes write one 128-byte rows2 block
            // (addresses rebuilt from the lane id every time: kept across the transform they would be
            //  spilled, and a scratch reload here waits for the coefficient prefetch and the stores)
            int ln = lane;
            asm volatile("" : "+v"(ln));
            // (NC = 4: the workgroup's four columns are half a rows2 block, 64 bytes, like the four-column kernels')
            float2* o = (pl ? ym : yw) + (size_t)gi_ * plane +
                        (NC == 8 ? (size_t)B * 16 : (size_t)(B >> 1) * 16 + (B & 1) * 8) + 2 * (ln & (NC - 1));
            const float2* lc = sm + (ln & (NC - 1)) * LINE;
            constexpr int RQ = 64 / NC;                      // row pairs per store instruction
            const int2 sr = rng[gi_];                        // (one LDS word pair, the same for every lane)
            const int s_lo = __builtin_amdgcn_readfirstlane(sr.x), s_hi = __builtin_amdgcn_readfirstlane(sr.y);
#ifdef SC_I1_PRIO
            __builtin_amdgcn_s_setprio(3);                   // (experiment: the store pass ahead of other waves' butterflies)
#endif
#if SC_Y_ROWMAJOR
            if constexpr (NC == 8) {                         // (lab: the eight-column kernel only)
                float2* orow = (pl ? ym : yw) + (size_t)gi_ * plane + (size_t)B * 8 + (ln & 7);
#pragma unroll 2
                for (int rp = s_lo + RQ * w + ln / NC; rp <= s_hi; rp += RQ * NC) {
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    const float2 c0 = lds_cell<true>(lc + ph(2 * rp)), c1 = lds_cell<true>(lc + ph(2 * rp + 1));
                    float2* p = orow + (size_t)(2 * rp) * Tx;
                    __builtin_nontemporal_store(f2{c0.x, c0.y}, reinterpret_cast<f2*>(p));
                    __builtin_nontemporal_store(f2{c1.x, c1.y}, reinterpret_cast<f2*>(p + Tx));
                }
            } else
#endif
#pragma unroll 2
            for (int rp = s_lo + RQ * w + ln / NC; rp <= s_hi; rp += RQ * NC)
                store_stream(o + (size_t)rp * (Tx >> 3) * 16, lds_cell<true>(lc + ph(2 * rp)), lds_cell<true>(lc + ph(2 * rp + 1)));
#ifdef SC_I1_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        }
        lds_barrier();                                       // (the next plane's first barrier would do; kept simple)
    }
}

// grid.x = Tx/8 workgroups j: index i = 8 (j / 16) + j % 8 in [0, Tx/16); (j / 8) & 1 = 0: column
// block i (columns 8i .. 8i+7), 1: the mirror block Tx/8 - 1 - i, whose coefficient columns
// are 8i+1 .. 8i+8 - the partner's but one, eight workgroup ids away on the same XCD.
// (column length 1024, one template per transform: capped at 128 registers = two workgroups per CU, four waves per SIMD -
//  6 % faster on a C3-like load with 1024-long columns, profiles/r04_c3_plans.txt; at 2048 and with paired templates the
//  kernel needs its 256)
template <int TY, bool PT>
__global__ void __launch_bounds__(512, (TY == 1024 && !PT) ? 4 : 2)
k_inv_cols_w8(const float2* __restrict__ uc, const float2* __restrict__ uc2,
              const float* __restrict__ wa, const float* __restrict__ mb, int Tx,
              int pair, int vfirst, int G, int rp_lo, int rp_hi, const float2* __restrict__ phx,
              int parity, const float2* __restrict__ tw, float2* __restrict__ yw,
              float2* __restrict__ ym, int ystride, int np, int pcj, int tstride,
              const TileDev* __restrict__ tiles, int py_valid, const TemplDev* __restrict__ tl, int jil) {
    // jil jobs (tile pairs of one orientation) interleaved along x: the 16 workgroups of eight column blocks and their
    // mirrors are followed by the same 16 of the next job, so that the workgroups which stream the SAME coefficient
    // lines - they do not depend on the tile pair - run on one XCD at the same time, 2 jil of them per column block
    // (grid.x = jil Tx/8, grid.y = jobs / jil).  jil = 1: one job per blockIdx.y as before.
    const int L = blockIdx.x, g16 = L / (16 * jil), r16 = L - g16 * 16 * jil, jq = r16 >> 4, j = r16 & 15;
    const int i = (g16 << 3) | (j & 7), jobx = (int)blockIdx.y * jil + jq;
    TAKE_TEMPLATE_SHARE(PT, (size_t)TY * Tx)
    if ((j >> 3) & 1)
        inv_cols_w8_body<TY, true, PT, 8>((Tx >> 3) - 1 - i, jobx, uc, uc2, wa, mb, Tx, pair, vfirst, G, rp_lo,
                                   rp_hi, phx, parity, tw, yw, ym, ystride, np, pcj, tstride, tiles, py_valid, tl);
    else
        inv_cols_w8_body<TY, false, PT, 8>(i, jobx, uc, uc2, wa, mb, Tx, pair, vfirst, G, rp_lo,
                                    rp_hi, phx, parity, tw, yw, ym, ystride, np, pcj, tstride, tiles, py_valid, tl);
}

// The same with FOUR columns and four waves per workgroup, one wave per SIMD: a wave may then use
// the whole register file (256 VGPRs + 256 AGPRs), which paired-template mode at column length
// 2048 needs - the parked spectrum, two coefficient planes and two sets of data do not fit 256.
// grid.x = Tx/4 workgroups, paired per XCD like the above (block i: columns 4i .. 4i+3; mirror
// block Tx/4 - 1 - i, coefficient columns 4i+1 .. 4i+4).
template <int TY, bool PT>
__global__ void __launch_bounds__(256, 1)
k_inv_cols_w4(const float2* __restrict__ uc, const float2* __restrict__ uc2,
              const float* __restrict__ wa, const float* __restrict__ mb, int Tx,
              int pair, int vfirst, int G, int rp_lo, int rp_hi, const float2* __restrict__ phx,
              int parity, const float2* __restrict__ tw, float2* __restrict__ yw,
              float2* __restrict__ ym, int ystride, int np, int pcj, int tstride,
              const TileDev* __restrict__ tiles, int py_valid, const TemplDev* __restrict__ tl) {
    const int j = blockIdx.x, i = ((j >> 4) << 3) | (j & 7);
    if ((j >> 3) & 1)
        inv_cols_w8_body<TY, true, PT, 4>((Tx >> 2) - 1 - i, (int)blockIdx.y, uc, uc2, wa, mb, Tx, pair, vfirst, G, rp_lo,
                                          rp_hi, phx, parity, tw, yw, ym, ystride, np, pcj, tstride, tiles, py_valid, tl);
    else
        inv_cols_w8_body<TY, false, PT, 4>(i, (int)blockIdx.y, uc, uc2, wa, mb, Tx, pair, vfirst, G, rp_lo,
                                           rp_hi, phx, parity, tw, yw, ym, ystride, np, pcj, tstride, tiles, py_valid, tl);
}

// ---- I1 at column length 512: HALF a wave per column (round 5) --------------------------------------
// Every DEM the reference ships (carrizo 900 x 505, grandcanyon 512 x 512, synthetic 200 x 200 ...) plans tiles of
// column length 512, and there the column pass ran the four-column kernels: 128 threads, seven workgroup barriers
// per transform, 1.7 - 2.7 TB/s (C1F: 19 of 43 ms; C5: a third of the search).  The wave-per-column idea needs one
// 16-point set per lane and stage: a 512-point column has 32 sets - half a wave.  So a wave takes TWO adjacent
// columns, lanes 0 - 31 one and lanes 32 - 63 the other, one set per lane and stage (cells fy = hl + 32 k of its
// column, k = 0 .. 15: exactly the inputs of stage-1 set hl), and a workgroup of eight waves takes SIXTEEN columns:
//  * no barrier inside a transform (a wave's LDS instructions execute in order; the two halves of a wave never
//    touch each other's line), two per template and plane around the store pass, as in k_inv_cols_w8;
//  * the store pass writes 16 columns x 2 rows = two whole 128-byte rows2 blocks per row pair, 1 KB per instruction;
//  * 16 cells per lane: the parked spectrum is 32 registers (64 with a second orientation, XP), the coefficient
//    prefetch 16 (32 with paired templates): two workgroups per CU, four waves per SIMD.
// Same products, butterflies, twiddle bases and operand order as inv_cols_sym_body / fft4_lines<512>: Y is
// bit-identical (option "variant" 18 keeps the four-column kernels at 512 for the cross-check).
// PT: templates 2k, 2k+1 of the orientation in one transform (x (a + i a2)); XP: the job is a PAIR of orientations
// with one template each (x a + i xb a2), see inv_cols_sym_body.
constexpr int H2_TY = 512, H2_COLS = 16;
__host__ __device__ constexpr size_t h2_lds() {
    return ((size_t)H2_COLS * w8_line<H2_TY>() + 4 * (H2_TY / 16 + H2_TY / 256) + SC_MAX_GROUP) * sizeof(float2);
}
template <bool MIRROR, bool PT, bool XP>
__device__ __forceinline__ void
inv_cols_h2_body(const int B, const int jobx, const float2* __restrict__ uc, const float2* __restrict__ uc2,
                 const float* __restrict__ wa, const float* __restrict__ mb, int Tx,
                 int pair, int vfirst, int G, int rp_lo, int rp_hi, const float2* __restrict__ phx,
                 int parity, const float2* __restrict__ tw, float2* __restrict__ yw,
                 float2* __restrict__ ym, int ystride, int np, int pcj, int tstride,
                 const TileDev* __restrict__ tiles, int py_valid, const TemplDev* __restrict__ tl) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    constexpr int TY = H2_TY, S = TY / 16, NK = 16, NC = H2_COLS;
    constexpr int LINE = w8_line<TY>();
    static_assert(S == 32 && !(XP && !PT), "half a wave per column; paired orientations are a paired-template mode");
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int hl = lane & 31, cw = 2 * w + (lane >> 5);          // set / cell index within the column; column of the workgroup
    float2* line = sm + cw * LINE;
    int i0A = 0, i0B = 0, vyA = 0, vyB = 0, pairB = 0;
    if constexpr (XP) {
        const int oA = 2 * jobx;                                 // the job's first orientation
        G = min(2, tstride - oA);                                // templates of the job (1: no partner)
        if (py_valid >= 0) {
            const int vy = max(tiles[2 * pair].vy, tiles[2 * pair + 1].vy);
            rp_hi = min(rp_hi, (py_valid + vy - 1) >> 1);
        }
        pair += oA * np;
        pairB = pair + (G > 1 ? np : 0);
        vfirst += oA;
    } else {
        const int ob = jobx / pcj, q = jobx - ob * pcj;
        const TileDev tA_ = tiles[2 * (pair + q)], tB_ = tiles[2 * (pair + q) + 1];
        i0A = tA_.i0; i0B = tB_.i0; vyA = tA_.vy; vyB = tB_.vy;
        if (py_valid >= 0) {
            const int vy = max(tA_.vy, tB_.vy);
            rp_hi = min(rp_hi, (py_valid + vy - 1) >> 1);
        }
        pair += ob * np + q;
        vfirst += ob * tstride;
    }
    // rows to store per transform (see inv_cols_w8_body): worked out once, parked in LDS behind the tables
    int2* const rng = reinterpret_cast<int2*>(sm + NC * LINE + 4 * (S + S / 16));
    const int NG = (PT && !XP) ? (G + 1) / 2 : (XP ? 1 : G);    // inverse transforms per plane
    {
        const int gi_ = threadIdx.x;
        if (gi_ < NG) {
            int lo = rp_lo, hi = rp_hi;
            if (!XP && tl && py_valid >= 0) {
                int klo = INT_MAX, khi = INT_MIN;
#pragma unroll
                for (int k = 0; k < (PT ? 2 : 1); ++k) {
                    const int ti = vfirst + (PT ? 2 * gi_ + k : gi_);
                    if (PT && k == 1 && 2 * gi_ + 1 >= G) break;
                    const int ilo = tl[ti].ilo, ihi = tl[ti].ihi;
                    if (vyA > 0) { klo = min(klo, ilo - i0A); khi = max(khi, ihi - i0A); }
                    if (vyB > 0) { klo = min(klo, ilo - i0B); khi = max(khi, ihi - i0B); }
                }
                if (khi < klo) { lo = 1; hi = 0; }
                else {
                    lo = max(lo, (max(klo, 0) + py_valid) >> 1);
                    hi = min(hi, (min(khi, TY) + py_valid) >> 1);
                }
            }
            rng[gi_] = make_int2(lo, hi);
        }
    }
    const size_t plane = (size_t)TY * Tx, hplane = half_plane(TY, Tx);
    yw += (size_t)jobx * ystride * plane;
    ym += (size_t)jobx * ystride * plane;
    const int ft = NC * B + cw;                // this half-wave's column of Y
    const int fs = MIRROR ? Tx - ft : ft;      // the coefficient column it pairs with
    float2* t1 = sm + NC * LINE;
    float2* t2 = t1 + 4 * S;
    for (int i = threadIdx.x; i < 4 * S; i += 512) t1[i] = tw[(i % S) << (i / S)];
    for (int i = threadIdx.x; i < 4 * (S / 16); i += 512) t2[i] = tw[((i % (S / 16)) << 4) << (i / (S / 16))];
    auto tw_of = [&](const float2* t, int n, int idx, float2 (&wq)[4]) {
#pragma unroll
        for (int m = 0; m < 4; ++m) wq[m] = lds_cell<true>(t + m * n + idx);
    };
    float c[NK], c2[PT ? NK : 1];
    for (int pl = 0; pl < 2; ++pl) {
        const float2* xcol = (pl ? uc2 : uc) + (size_t)pair * plane + (size_t)ft * TY;
        const float2* xcolB = (pl ? uc2 : uc) + (size_t)pairB * plane + (size_t)ft * TY;
        const float* hsrc = (pl ? mb : wa) + (size_t)vfirst * hplane + (size_t)fs * TY;
        const bool rot = pl == 0 && parity == 1;          // odd W: factor i (own columns) / -i (mirrors)
        // ---- park X P {i} of the lane's cells (the arithmetic of inv_cols_sym_body, cell for cell): the phase
        // factors first (table entries, back quickly), then the spectrum
        float2 pvv[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int fy = hl + 32 * k;
            if (!MIRROR) {
                pvv[k] = phase_tab(tw + TY, phx, fy, ft);
            } else {
                pvv[k] = phase_tab(tw + TY, phx, (TY - fy) & (TY - 1), (Tx - ft) & (Tx - 1));
                pvv[k].y = -pvv[k].y;
            }
        }
        asm volatile("" ::: "memory");
        float2 xp[NK], xpB[XP ? NK : 1];
#pragma unroll
        for (int k = 0; k < NK; ++k) xp[k] = xcol[hl + 32 * k];
        if constexpr (XP) {
#pragma unroll
            for (int k = 0; k < NK; ++k) xpB[k] = xcolB[hl + 32 * k];
        }
        auto park = [&](float2 x, float2 pv) {
            float2 v = cmul(x, pv);
            if (rot) v = MIRROR ? make_float2(v.y, -v.x) : make_float2(-v.y, v.x);
            return v;
        };
#pragma unroll
        for (int k = 0; k < NK; ++k) xp[k] = park(xp[k], pvv[k]);
        if constexpr (XP) {
#pragma unroll
            for (int k = 0; k < NK; ++k) xpB[k] = park(xpB[k], pvv[k]);
        }
        // coefficient of cell fy: a[fs][fy], mirrors a[fs][-fy mod TY] = a[fs][TY - hl - 32 k] but for fy = 0
        const float* cbase = hsrc + (MIRROR ? TY - hl : hl);
        const int c0off = (MIRROR && hl == 0) ? -TY : 0;
        auto fetch = [&](int gi_) {
            const float* p = cbase + (size_t)((PT && !XP) ? 2 * gi_ : gi_) * hplane;
            const bool has2 = PT && (XP ? G > 1 : 2 * gi_ + 1 < G);
            const float* p2 = has2 ? p + hplane : p;
            const unsigned keep2 = has2 ? ~0u : 0u;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int off = k ? (MIRROR ? -32 * k : 32 * k) : c0off;
                c[k] = p[off];
                if constexpr (PT) c2[k] = __uint_as_float(__float_as_uint(p2[off]) & keep2);
            }
        };
        fetch(0);
        for (int gi_ = 0; gi_ < NG; ++gi_) {
            lds_barrier();                                   // the store pass of the previous transform is done with the lines
            float2 wq[4];
            int lt = hl;                                     // (LDS addresses rebuilt from the lane id in every transform)
            asm volatile("" : "+v"(lt));
            {
                float2 a1[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float2 x = xp[j];
                    if constexpr (XP) a1[j] = make_float2(x.x * c[j] - xpB[j].y * c2[j], x.y * c[j] + xpB[j].x * c2[j]);
                    else a1[j] = PT ? make_float2(x.x * c[j] - x.y * c2[j], x.x * c2[j] + x.y * c[j])
                                    : make_float2(c[j] * x.x, c[j] * x.y);
                }
                tw_of(t1, S, lt, wq);
                set_compute_store<TY, 16, 0, true, false, true>(line, lt, a1, wq);
            }
            if (gi_ + 1 < NG) fetch(gi_ + 1);
            asm volatile("" ::: "memory");
            float2 a[16];
            set_load<TY, true>(line, lt, a);
            tw_of(t2, S / 16, lt >> 4, wq);
#if SC_H2_XLANE
            // Stage 2 into registers; stage 3 - radix 2 at stride 256 - across lanes.  Output m of lane hl's stage-2 set
            // is element (hl & 15) + 16 m + 256 (hl >> 4) of its column: the butterfly partner of every one of them sits
            // in the lane 16 further on (the other 16-lane row of the half-wave), same m.  v_permlane16_swap of a value
            // with its own copy leaves the even row's value in one register and the odd row's in the other, in both
            // rows; the even row keeps the sum (element e), the odd row the difference (element e + 256) - where their
            // stage-2 outputs were.  One LDS round trip (16 cells written, 16 read back per lane) less per transform;
            // x + y and x - y as fma(+-1, y, x): the same bits.
            {
                float2 o2[16];
                set_compute_regs<true>(a, wq, o2);
                const float sgn = (lt & 16) ? -1.f : 1.f;
                float2* const wb = line + ph((lt & 15) + ((lt >> 4) << 8));
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    float lo_x = o2[m].x, hi_x = o2[m].x, lo_y = o2[m].y, hi_y = o2[m].y;
                    xlane_swap16(lo_x, hi_x);                // lo: the even row's value, hi: the odd row's - in both rows
                    xlane_swap16(lo_y, hi_y);
                    wb[17 * m] = make_float2(fmaf(sgn, hi_x, lo_x), fmaf(sgn, hi_y, lo_y));
                }
            }
#else
            set_compute_store<TY, 16, 4, true, false, true>(line, lt, a, wq);
            asm volatile("" ::: "memory");
            set_load<TY, true>(line, lt, a);
            set_compute_store<TY, 2, 8, true, false, true>(line, lt, a, wq);
#endif
            lds_barrier();                                   // all sixteen lines are complete
            // ---- store: lane (q, c) of wave w takes column c of row pair s_lo + 4 w + 32 it + q: sixteen lanes write
            // the two 128-byte rows2 blocks of a row pair
            int ln = lane;
            asm volatile("" : "+v"(ln));
            float2* o = (pl ? ym : yw) + (size_t)gi_ * plane + (size_t)B * 32 + 2 * (ln & 15);
            const float2* lc = sm + (ln & 15) * LINE;
            const int2 sr = rng[gi_];
            const int s_lo = __builtin_amdgcn_readfirstlane(sr.x), s_hi = __builtin_amdgcn_readfirstlane(sr.y);
#pragma unroll 2
            for (int rp = s_lo + 4 * w + (ln >> 4); rp <= s_hi; rp += 32)
                store_stream(o + (size_t)rp * (Tx >> 3) * 16, lds_cell<true>(lc + ph(2 * rp)), lds_cell<true>(lc + ph(2 * rp + 1)));
        }
        lds_barrier();
    }
}

// grid.x = Tx/16 workgroups j: group j / 16, kind (j / 8) & 1, index i = 8 (j / 16) + j % 8: kind 0 is column block i
// (columns 16 i .. 16 i + 15), kind 1 the mirror block Tx/16 - 1 - i whose coefficient columns are 16 i + 1 .. 16 i + 16 -
// eight workgroup ids from the block that streams (all but one of) the same lines, on the same XCD.  grid.y: jobs
// (tile pairs x batched orientations; XP: pairs of orientations); grid.z: parts of the template loop (template_share).
template <bool PT, bool XP>
__global__ void __launch_bounds__(512, 4)
k_inv_cols_h2(const float2* __restrict__ uc, const float2* __restrict__ uc2,
              const float* __restrict__ wa, const float* __restrict__ mb, int Tx,
              int pair, int vfirst, int G, int rp_lo, int rp_hi, const float2* __restrict__ phx,
              int parity, const float2* __restrict__ tw, float2* __restrict__ yw,
              float2* __restrict__ ym, int ystride, int np, int pcj, int tstride,
              const TileDev* __restrict__ tiles, int py_valid, const TemplDev* __restrict__ tl) {
    const int j = blockIdx.x, i = ((j >> 4) << 3) | (j & 7);
    if constexpr (!XP) TAKE_TEMPLATE_SHARE(PT, (size_t)H2_TY * Tx)
    if ((j >> 3) & 1)
        inv_cols_h2_body<true, PT, XP>((Tx >> 4) - 1 - i, (int)blockIdx.y, uc, uc2, wa, mb, Tx, pair, vfirst, G, rp_lo,
                                       rp_hi, phx, parity, tw, yw, ym, ystride, np, pcj, tstride, tiles, py_valid, tl);
    else
        inv_cols_h2_body<false, PT, XP>(i, (int)blockIdx.y, uc, uc2, wa, mb, Tx, pair, vfirst, G, rp_lo,
                                        rp_hi, phx, parity, tw, yw, ym, ystride, np, pcj, tstride, tiles, py_valid, tl);
}

// ---- I2: inverse row FFT -> epilogue -> fold ---------------------------------
// grid = (valid row blocks): one workgroup per block of 4 tile rows.  The rows
// are done as two sub-batches of 2 rows x {W plane, M plane} = 4 LDS lines, so
// xcorr and T3 of a cell come out of the same transform call.  Per sub-batch
// the workgroup loops over the G templates of the launch, in fold order, with
// the running best SNR of its cells in registers: the SNR plane is read once
// per launch, and a template that wins a cell stores its (snr, amp, id) right
// away (a few wins per cell per launch).  Software pipelined like I1.
// FULL = false is the lean variant for templates whose only mask is the
// window-limit rectangle (Scarp, Ricker); FULL = true adds the error masks and
// the explicit per-cell masks of generic plugins.
struct RowArgs {
    int Ty, Py, Qx, circ_y, circ_x;     // tile geometry
    int cy0, cx0, cw;                   // core origin and width
    int pair, first, G;
    int rp_lo, rp_n;                    // valid row pairs of the tile: [rp_lo, rp_lo + rp_n)
    int dbg;                            // diagnostic ablation bits (SC_DBG), 0 in production
    int ystride;                        // planes between the Y blocks of consecutive jobs
    // orientation batching (fast kernel): the launch folds nb orientations of G templates each
    // (templates first + b*G + g, consecutive); the Y block of (orientation b, pair q of the
    // chunk) is job b*pcj + q, its tile norms are entry b*np + pair
    int nb, np, pcj;
    SibSync sib;                        // sibling rendezvous of the fast kernel (rows 2rp, 2rp+1)
    unsigned long long* stats;          // {wins, wins near the resolution floor} of the search (sc_get_resolution_stats)
    int xp;                             // paired ORIENTATIONS (k_inv_cols_sym, XP): G templates, one per orientation; norms entry id*np + pair
    int skip;                           // fast kernel: skip the templates whose window limits mask the workgroup's whole row
    // fast kernel, SPLITK (small grids): the launch's transforms are dealt out over nsplit workgroups per row
    // (blockIdx.z), part h > 0 folding into a scratch record of its own - planes (h - 1) * nc .. of s2 / a2 / i2,
    // every valid cell written - that k_merge_split folds into the record in order afterwards
    int nsplit;
    size_t nc;
    float *s2, *a2;
    uint32_t* i2;
    // fast kernel, NEAR (the exact mode of the host layer): a byte per core cell, set where a template scored within
    // near_w (relative) of the running best without equalling it - the cells whose argmax the float32 convolution
    // cannot vouch for and that the host re-scores on the real-space path
    float near_w;
    uint8_t* near;
    // ... and, per near-tie, an EVENT (cell, id of the template scored, id of the record's holder at that moment) appended to a
    // list (round 5, end): the candidates of a flagged cell - the only templates its float64 argmax can be, sc_get_near_events
    uint32_t* ev;                       // 3 words per event (nullptr: flags only)
    unsigned long long* ev_count;       // events so far (counts on beyond the capacity: the host sees the overflow)
    unsigned long long ev_cap;
};
// One launch may serve several tile pairs (grid.y): pair = ra.pair + blockIdx.y,
// its Y planes ystride planes further on.  More workgroups per launch fill the
// last round of the chip better (a row's workgroup lives for the whole template
// loop, and 1816 rows on 768 slots would leave a third of the last round idle).

template <int TX, bool FULL>
__global__ void __launch_bounds__(fft_threads(TX), 2)
k_inv_rows(const float2* __restrict__ yw, const float2* __restrict__ ym,
           RowArgs ra, Geom g, const TileDev* __restrict__ tiles,
           const TemplDev* __restrict__ templ, const double* __restrict__ sums,
           const double* __restrict__ wl1, const double* __restrict__ norms, float kappa,
           const double* __restrict__ xaxis, const double* __restrict__ yaxis,
           const float2* __restrict__ tw, float* __restrict__ best_snr,
           float* __restrict__ best_amp, uint32_t* __restrict__ best_id,
           float* __restrict__ map_amp, float* __restrict__ map_snr) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    FftTw<TX> twr;
    twr.load(tw);
    constexpr int NT = fft_threads(TX);
    constexpr int E = 2 * TX / NT;          // cells per thread per sub-batch
    constexpr int EP = 2 * TX / (2 * NT);   // float4 loads per thread per plane
    const size_t plane = (size_t)ra.Ty * TX;
    ra.pair += blockIdx.y;
    yw += (size_t)blockIdx.y * ra.ystride * plane;
    ym += (size_t)blockIdx.y * ra.ystride * plane;
    const TileDev tA = tiles[2 * ra.pair], tB = tiles[2 * ra.pair + 1];
    const float scale = 1.0f / ((float)ra.Ty * (float)TX);
    float4 xreg[EP], yreg[EP];
    for (int sb = 0; sb < 2; ++sb) {
        // this workgroup takes row pairs 2*blockIdx.x and 2*blockIdx.x + 1 of the
        // valid range, one per sub-batch (a row pair is a contiguous 2*TX run)
        const int rp = ra.rp_lo + 2 * blockIdx.x + sb;
        if (rp >= ra.rp_lo + ra.rp_n) break;              // block-uniform
        const float2* base1 = yw + (size_t)rp * 2 * TX;
        const float2* base2 = ym + (size_t)rp * 2 * TX;
        auto fetch = [&](int gi_) {
#pragma unroll
            for (int u = 0; u < EP; ++u) {
                size_t src = (size_t)gi_ * plane + 2 * (threadIdx.x + u * NT);
                xreg[u] = *reinterpret_cast<const float4*>(base1 + src);
                yreg[u] = *reinterpret_cast<const float4*>(base2 + src);
            }
        };
        // cell u of this thread is tile-local (row0 + e/TX - Py, e%TX - Qx);
        // slot 2u + part is that cell of tile A / B (re / im of the transforms)
        const int row0 = 2 * rp - ra.Py;
        auto locate = [&](int u, int& ri, int& cj) {
            int e = threadIdx.x + u * NT;
            int r2 = e / TX;
            ri = row0 + r2;
            cj = e - r2 * TX - ra.Qx;
            if (ra.circ_y) ri &= (ra.Ty - 1);
            if (ra.circ_x) cj &= (TX - 1);
        };
        float b_snr[2 * E];
        unsigned valid = 0;
#pragma unroll
        for (int u = 0; u < E; ++u) {
            int ri, cj;
            locate(u, ri, cj);
            bool okA = ri >= 0 && ri < tA.vy && cj >= 0 && cj < tA.vx;
            bool okB = ri >= 0 && ri < tB.vy && cj >= 0 && cj < tB.vx;
            if (okA) valid |= 1u << (2 * u);
            if (okB) valid |= 2u << (2 * u);
            const bool rd = !map_amp && !SC_DBGBIT(ra.dbg, 8);
            b_snr[2 * u] = (okA && rd) ? best_snr[(size_t)(tA.i0 + ri - ra.cy0) * ra.cw + (tA.j0 + cj - ra.cx0)] : 0.f;
            b_snr[2 * u + 1] = (okB && rd) ? best_snr[(size_t)(tB.i0 + ri - ra.cy0) * ra.cw + (tB.j0 + cj - ra.cx0)] : 0.f;
        }
        if (!SC_DBGBIT(ra.dbg, 1)) fetch(0);
        for (int gi_ = 0; gi_ < ra.G; ++gi_) {
            const TemplDev* tp = templ + ra.first + gi_;
            EpiScal es = sc_epi_scalars(sums, ra.first + gi_);
            sc_epi_floor(es, sums[2 * (ra.first + gi_)], sums[2 * (ra.first + gi_) + 1],
                         wl1[ra.first + gi_], norms[2 * ra.pair], norms[2 * ra.pair + 1],
                         (double)ra.Ty * TX, kappa);
            const float scale_w = scale / sc_fft_alpha(sums, ra.first + gi_);
            const uint32_t tid_ = tp->id;
            // window-limit rectangle in tile-local coordinates (scalar)
            const int rloA = tp->ilo - tA.i0, rhiA = tp->ihi - tA.i0, cloA = tp->jlo - tA.j0, chiA = tp->jhi - tA.j0;
            const int rloB = tp->ilo - tB.i0, rhiB = tp->ihi - tB.i0, cloB = tp->jlo - tB.j0, chiB = tp->jhi - tB.j0;
#pragma unroll
            for (int u = 0; u < EP; ++u) {
                int c = threadIdx.x + u * NT;       // column; the float4 holds rows 0 and 1
                sm[lidx<TX>(0, c)] = make_float2(xreg[u].x, xreg[u].y);
                sm[lidx<TX>(1, c)] = make_float2(xreg[u].z, xreg[u].w);
                sm[lidx<TX>(2, c)] = make_float2(yreg[u].x, yreg[u].y);
                sm[lidx<TX>(3, c)] = make_float2(yreg[u].z, yreg[u].w);
            }
            lds_barrier();
            if (gi_ + 1 < ra.G && !SC_DBGBIT(ra.dbg, 1)) fetch(gi_ + 1);
            if (!SC_DBGBIT(ra.dbg, 2)) fft4_lines<TX, true>(sm, twr);
            if (!SC_DBGBIT(ra.dbg, 4)) {
                // branch-free scoring of the 2E slots, then the (rare) stores
                float t_amp[2 * E], t_snr[2 * E];
#pragma unroll
                for (int u = 0; u < E; ++u) {
                    int e = threadIdx.x + u * NT;
                    int r2 = e / TX, s = e - r2 * TX;
                    float2 xc = sm[lidx<TX>(r2, s)];
                    float2 t3 = sm[lidx<TX>(2 + r2, s)];
                    int ri, cj;
                    locate(u, ri, cj);
#pragma unroll
                    for (int part = 0; part < 2; ++part) {
                        const int c = 2 * u + part;
                        float amp, snr;
                        sc_epilogue((part ? xc.y : xc.x) * scale_w,
                                    (part ? t3.y : t3.x) * scale, es, amp, snr);
                        bool keep = (valid >> c) & 1u;
                        if (FULL) {
                            if (keep)
                                sc_apply_masks(*tp, g, xaxis, yaxis, (part ? tB.i0 : tA.i0) + ri,
                                               (part ? tB.j0 : tA.j0) + cj, amp, snr);
                        } else {
                            keep = keep && (part ? (ri >= rloB && ri <= rhiB && cj >= cloB && cj <= chiB)
                                                 : (ri >= rloA && ri <= rhiA && cj >= cloA && cj <= chiA));
                        }
                        t_amp[c] = keep ? amp : 0.f;
                        t_snr[c] = keep ? snr : 0.f;
                    }
                }
                unsigned won = 0;
                if (map_amp) {
                    won = valid;
                } else {
#pragma unroll
                    for (int c = 0; c < 2 * E; ++c) {
                        // sc_fold on the SNR alone: take if greater; a NaN score
                        // poisons the cell once (core.py:230-240, see sc_fold)
                        float bs = b_snr[c], ts = t_snr[c];
                        bool take = bs < ts;
                        bool poison = (ts != ts) && (bs == bs);
                        b_snr[c] = (take || poison) ? ts : bs;
                        if (take || poison) won |= 1u << c;
                    }
                }
                if (won && !SC_DBGBIT(ra.dbg, 8)) {
#pragma unroll
                    for (int u = 0; u < E; ++u) {
                        int ri, cj;
                        locate(u, ri, cj);
#pragma unroll
                        for (int part = 0; part < 2; ++part) {
                            const int c = 2 * u + part;
                            if (!((won >> c) & 1u)) continue;
                            size_t o = (size_t)((part ? tB.i0 : tA.i0) + ri - ra.cy0) * ra.cw +
                                       ((part ? tB.j0 : tA.j0) + cj - ra.cx0);
                            if (map_amp) {
                                map_amp[o] = t_amp[c];
                                map_snr[o] = t_snr[c];
                            } else {
                                bool nan = t_snr[c] != t_snr[c];
                                best_snr[o] = t_snr[c];
                                best_amp[o] = nan ? 0.f : t_amp[c];
                                best_id[o] = nan ? SC_ID_NONE : tid_;
                                if (ra.stats) {       // a win; near the resolution floor? (see k_inv_rows_fast)
                                    const float xr_ = t_amp[c] / es.inv_ts, T1 = xr_ * t_amp[c];
                                    const float fl = es.d3 + fabsf(xr_) * es.dx2 + es.dxx;
                                    const float d = (T1 / t_snr[c] - (float)SC_EPS) / es.inv_n;
                                    atomicAdd(ra.stats, 1ull);
                                    if (fl > 0.f && d < 256.f * fl) atomicAdd(ra.stats + 1, 1ull);
                                }
                            }
                        }
                    }
                }
            }
            lds_barrier();
        }
    }
}

// ---- I2 (fast): T in {512, 1024, 2048} ---------------------------------------
// One workgroup (T/8 threads) per tile ROW, 2 LDS lines = {W plane, M plane}.
// Compared with the generic kernel the transform is unrolled into its three
// stages so that
//   * stage 1 takes its 16 points straight from global memory into registers -
//     no LDS fill pass;
//   * stage 3 is fused with the epilogue: a thread computes the same butterflies
//     of the W and the M line, so xcorr and T3 of its cells meet in registers -
//     no LDS write of the result, no read back;
//   * the running best record of a thread's 16 cells stays in registers across
//     the G templates of the launch and is written back once.
// Small workgroups on purpose: four of them share a CU (39 KB of LDS and at most
// 128 VGPRs each - hence the packed winner index and the two-entry twiddle
// table), so one computes while another sits in a barrier or waits for LDS or
// HBM; a 512-thread workgroup per row pair left the SIMDs idle two thirds of the
// time, three workgroups per CU were 6 % slower than four.  Complex values are native 2-vectors (v_pk_add/mul/fma_f32: one
// instruction per complex add, two per complex product) and twiddles come from
// small LDS tables.  The two rows of a rows2 pair share their 128-byte lines:
// their workgroups are given block ids 8 apart, i.e. the same XCD at the same
// time, so the second reader hits in that XCD's L2.
#ifndef SC_I2_WAVES
#define SC_I2_WAVES 4      // waves per SIMD the fast row kernel is compiled for (LDS fits 4 workgroups per CU)
#endif
#ifndef SC_I2_FETCH_AT
#define SC_I2_FETCH_AT 0
#endif
#ifndef SC_I2_GRP
#define SC_I2_GRP 8         // output pairs per record branch of the row pass (SC_I2_RAREWIN); 4, 2, 1: the 2048 kernel spills 44 - 92 B
#endif
#ifndef SC_I2_RAREWIN
#ifndef SC_I2_NEAR_WAVES
#define SC_I2_NEAR_WAVES 4  // waves per SIMD the near-tie variant is compiled for (round 6: four, like the plain kernel - event loop, amplitude stored at a win)
#endif
#ifndef SC_I2_SITE
#define SC_I2_SITE 0        // lab (round 6): the record's update under a branch per output inside the record branch - measured, not kept
#endif
#ifndef SC_I2_AMPW_PLAIN
#define SC_I2_AMPW_PLAIN 0  // lab (round 6): the plain row kernel stores a winner's amplitude at the win as well
#endif
#ifndef SC_I2_NEAR_AMPW
#define SC_I2_NEAR_AMPW 1
#endif
#ifndef SC_I2_NEAR_LOOP
#define SC_I2_NEAR_LOOP 1   // (round 6) the near-tie events of a record branch from a loop over a mask instead of one site per output
#endif
#ifndef SC_I2_NEAR_RARE
#define SC_I2_NEAR_RARE 1   // the near-tie variant of the row pass on the deferred record update too (0: selects and a test per output)
#endif
#ifndef SC_I2_RARE512
#define SC_I2_RARE512 1     // the deferred record update in the 512-cell row kernels too (round 5, end: row pass of the small grids -4.5 %; 0: selects)
#endif
#define SC_I2_RAREWIN 1     // the row pass records a winner's output and index under a rarely taken branch (0: selects per output)
#endif
#ifndef SC_I2_STATIC
#define SC_I2_STATIC 0     // 1: templates whose window limits cover a whole tile row skip the range test per output (stage 3); measured, see DESIGN.md
#endif

template <int TX>
__host__ __device__ constexpr bool inv_rows_fast_ok() { return TX == 512 || TX == 1024 || TX == 2048; }

// Byte-offset access to the best-record planes: one 32-bit offset per lane on
// top of a workgroup-uniform row pointer.
template <typename V>
__device__ __forceinline__ V& at_bytes(V* base, uint32_t off) {
    return *reinterpret_cast<V*>(reinterpret_cast<char*>(base) + off);
}
constexpr int EPI_FLOATS = 5;                  // per-template scalars staged in LDS
// LDS: 2 lines | per-template scalars | stage-1 twiddle bases (w^tt, w^8tt per set index)
//      | stage-2 twiddles (16 per p = 0 .. S/16-1)
template <int TX>
__host__ __device__ constexpr int inv_rows_fast_threads() { return TX / 8; }
template <int TX>
__host__ __device__ constexpr size_t inv_rows_fast_lds() {
    return fft_lds_bytes(TX) / 2 + (size_t)SC_MAX_GROUP * EPI_FLOATS * sizeof(float) +
           (size_t)(TX / 16) * 2 * sizeof(float2) + (size_t)(TX / 256) * 16 * sizeof(float2) +
           sizeof(unsigned long long);                  // + the mask of the transforms this row takes part in
}

// SPLITK (round 5): a launch may carry up to 255 templates, every share of it at most SC_MAX_GROUP TRANSFORMS (the
// 64-bit mask of a workgroup's transforms is relative to its share) - with paired templates twice that many scalar sets
template <int TX>
__host__ __device__ constexpr size_t inv_rows_fast_lds_split() {
    return inv_rows_fast_lds<TX>() + (size_t)SC_MAX_GROUP * EPI_FLOATS * sizeof(float);
}

// (FULL - error masks, per-cell masks of generic plugins - carries the mask test's float64 coordinates: at four
//  waves per SIMD it spilled 76 - 92 B per lane inside the template loop; SC_I2_WAVES_FULL waves, 168 registers)
#ifndef SC_I2_WAVES_FULL
#define SC_I2_WAVES_FULL 3
#endif
template <int TX, bool FULL, bool MAPS, bool PT, bool SPLITK = false, bool NEAR = false>
__global__ void __launch_bounds__(inv_rows_fast_threads<TX>(), NEAR ? (PT ? 3 : SC_I2_NEAR_WAVES) : (FULL && !MAPS) ? SC_I2_WAVES_FULL : SC_I2_WAVES)
k_inv_rows_fast(const float2* __restrict__ yw, const float2* __restrict__ ym,
                RowArgs ra, Geom g, const TileDev* __restrict__ tiles,
                const TemplDev* __restrict__ templ, const double* __restrict__ sums,
                const double* __restrict__ wl1, const double* __restrict__ norms, float kappa,
                const double* __restrict__ xaxis, const double* __restrict__ yaxis,
                const float2* __restrict__ tw, float* __restrict__ best_snr,
                float* __restrict__ best_amp, uint32_t* __restrict__ best_id,
                float* __restrict__ map_amp, float* __restrict__ map_snr) {
    using pk::v2;
    extern __shared__ __attribute__((aligned(16))) float2 sm_[];
    v2* sm = reinterpret_cast<v2*>(sm_);
    constexpr int S = TX / 16;                 // 16-point sets per line; NT = 2S threads
    constexpr int NT = 2 * S;
    constexpr int R3 = TX / 256;               // radix of the last stage (2, 4, 8)
    constexpr int NB3 = 16 / R3;               // its butterflies per set
    constexpr int NU = NB3 / 2;                // butterflies per plane per thread in stage 3
    constexpr int NC = NU * R3;                // cells per thread (= 8)
    constexpr int LINE = fft_line(TX);
    constexpr int FETCH_AT = SC_I2_FETCH_AT;   // next template's points: 0 after stage 1, 1 after stage 2
    static_assert(inv_rows_fast_threads<TX>() == NT && NC == 8, "fast I2 geometry");
    const int id = threadIdx.x;
    // blocks b and b + 8 take the two rows of one pair (same XCD, same time)
    const int pi = (blockIdx.x >> 4) * 8 + (blockIdx.x & 7), rh = (blockIdx.x >> 3) & 1;
    // sibling rendezvous (SibSync): this workgroup's word and the word of the other row of the pair
    uint32_t* const sib_mine = ra.sib.slots ? ra.sib.slots + (size_t)blockIdx.y * gridDim.x + blockIdx.x : nullptr;
    const uint32_t* const sib_theirs = ra.sib.slots ? ra.sib.slots + (size_t)blockIdx.y * gridDim.x + (blockIdx.x ^ 8) : nullptr;
    if (pi >= ra.rp_n) {
        if (sib_mine && id == 0) sib_publish(sib_mine, ra.sib.base + SIB_DONE);
        return;
    }
    const int rp = ra.rp_lo + pi;
    const size_t plane = (size_t)ra.Ty * TX;
    ra.pair += blockIdx.y;
    // job of orientation b: b * pcj + blockIdx.y (RowArgs); orientation 0 here, fetch() adds the rest
    yw += (size_t)blockIdx.y * ra.ystride * plane;
    ym += (size_t)blockIdx.y * ra.ystride * plane;
    const int GT = ra.nb * ra.G;               // templates folded by this launch, first + 0 .. GT-1
    const TileDev tA = tiles[2 * ra.pair], tB = tiles[2 * ra.pair + 1];
    const float scale = 1.0f / ((float)ra.Ty * (float)TX);
    const int NGO = PT ? (ra.G + 1) / 2 : ra.G;                  // transforms per orientation
    // SPLITK: this workgroup's share [k0, k1) of the launch's transforms - at most SC_MAX_GROUP of them (the host
    // sees to it) - and the templates [tlo, thi] they carry.  Everything per template below - the scalar table, the
    // mask of transforms - is relative to the share, so that the LAUNCH may carry up to 255 templates (round 5)
    int k0 = 0, k1 = ra.nb * NGO;
    if constexpr (SPLITK) {
        const int ng = k1;
        k0 = (int)(((long long)ng * blockIdx.z) / ra.nsplit);
        k1 = (int)(((long long)ng * (blockIdx.z + 1)) / ra.nsplit);
    }
    auto templ_of = [&](int k, int part) {                       // template of the launch in transform k (PT: part 0 / 1)
        const int ob_ = k / NGO, ok__ = k - ob_ * NGO;
        return ob_ * ra.G + (PT ? min(2 * ok__ + part, ra.G - 1) : ok__);
    };
    const int tlo = SPLITK ? (k1 > k0 ? templ_of(k0, 0) : 0) : 0;
    const int thi = SPLITK ? (k1 > k0 ? templ_of(k1 - 1, 1) : -1) : GT - 1;

    float* epi_ = reinterpret_cast<float*>(sm + 2 * LINE);
    v2* tw1 = reinterpret_cast<v2*>(epi_ + (SPLITK ? 2 : 1) * SC_MAX_GROUP * EPI_FLOATS);
    v2* tw2 = tw1 + 2 * S;
    float* const epi = epi_ - EPI_FLOATS * tlo;                  // indexed by the template of the LAUNCH, tlo .. thi
    // per-template scalars of the epilogue (float64 arithmetic on the template
    // sums) are the same for every cell: thread t prepares template t once and
    // parks the five floats in LDS.  xcorr = xr*scale_w, T3 = tr*scale
    //   =>  amp = xr*ka, T1 = xr^2*kt, floor = |xr|*kx2 + fl0
    for (int tl_ = tlo + id; tl_ <= thi; tl_ += NT) {
        const int it = ra.first + tl_;
        const int nrm = (ra.xp ? tl_ : tl_ / ra.G) * ra.np + ra.pair;    // tile-pair norms of the template's orientation
        EpiScal es = sc_epi_scalars(sums, it);
        sc_epi_floor(es, sums[2 * it], sums[2 * it + 1], wl1[it], norms[2 * nrm],
                     norms[2 * nrm + 1], (double)ra.Ty * TX, kappa);
        const float scale_w = scale / sc_fft_alpha(sums, it);
        const float ka = scale_w * es.inv_ts;
        float* e = epi + EPI_FLOATS * tl_;
        e[0] = ka;
        e[1] = scale_w * ka;
        e[2] = scale_w * es.dx2;
        e[3] = es.d3 + es.dxx;
        e[4] = es.inv_n;
    }
    // twiddle tables (inverse transform: conjugates of the forward table):
    //   stage 1 (stride 1):  set tt multiplies output m by w^(tt*m); bases m = 1,2,4,8
    //   stage 2 (stride 16): set tt multiplies output m by w^(16*(tt>>4)*m)
    for (int i = id; i < 2 * S; i += NT) {                       // bases w^tt and w^(8 tt) per set tt
        float2 w = tw[(i >> 1) << (3 * (i & 1))];
        tw1[i] = v2{w.x, -w.y};
    }
    for (int i = id; i < S; i += NT) {
        float2 w = tw[((i >> 4) << 4) * (i & 15)];
        tw2[i] = v2{w.x, -w.y};
    }

    // stage-1 mapping: plane and set of this thread (the plane is the same for a
    // whole wave when S is a multiple of 64: keep its base pointer scalar then)
    const int pl1 = (S % 64 == 0) ? __builtin_amdgcn_readfirstlane(id / S) : id / S;
    const int tt1 = id % S;
    v2* line1 = sm + pl1 * LINE + 17 * tt1;
#if SC_Y_ROWMAJOR
    // lab: the hand-off row-major, Y[row][column] - the row reads whole 128-byte lines of its own
    const char* src1 = reinterpret_cast<const char*>((pl1 ? ym : yw) + (size_t)(2 * rp + rh) * TX);
    const uint32_t voff1 = (uint32_t)(tt1 * sizeof(float2));
    constexpr size_t JSTRIDE1 = (size_t)S * sizeof(float2);
#else
    const char* src1 = reinterpret_cast<const char*>((pl1 ? ym : yw) + (size_t)rp * 2 * TX);
    const uint32_t voff1 = (uint32_t)(((tt1 >> 3) * 16 + (tt1 & 7) * 2 + rh) * sizeof(float2));
    constexpr size_t JSTRIDE1 = (size_t)2 * S * sizeof(float2);
#endif
    // stage-2 mapping (standard): line id / S, set tt2
    const int tt2 = id % S;
    v2* line2 = sm + (id / S) * LINE;
    const v2* rd2 = line2 + ph(tt2);
    v2* wr2 = line2 + ph((tt2 & 15) + ((tt2 >> 4) << 8));
    const v2* twp2 = tw2 + ((tt2 >> 4) << 4);
    // stage-3 mapping: butterflies id + u*2S of both planes
    const int rem3 = id;
    const v2* lineW = sm + ph(rem3);
    const v2* lineM = lineW + LINE;
    // cells of this thread: row 2rp + rh (tile row ri), columns
    // cj0 + u*2S + 256 m, c = u*R3 + m; part 0 belongs to tile A (real part of
    // the packed transform), part 1 to tile B
    int ri = 2 * rp + rh - ra.Py;
    if (ra.circ_y) ri &= (ra.Ty - 1);
    const bool rowA = ri >= 0 && ri < tA.vy, rowB = ri >= 0 && ri < tB.vy;
    if (!rowA && !rowB) {                                        // whole workgroup
        if (sib_mine && id == 0) sib_publish(sib_mine, ra.sib.base + SIB_DONE);
        return;
    }
    // Transforms this row takes part in, one bit each (at most SC_MAX_GROUP = 64).  A template whose
    // window limits (WindowedTemplate.py:66-84) mask the whole row on both tiles scores nothing here:
    // its planes are not fetched, not transformed (and k_inv_cols_w8 has not stored the row for it).
    // 2.4 % of all (row, template) pairs of the 10000 x 10000 benchmark; one orientation per launch only.
    unsigned long long* const actm = reinterpret_cast<unsigned long long*>(tw2 + S);
    const bool skip_ok = !PT && !FULL && !MAPS && ra.skip && ra.nb == 1 && !ra.xp && ra.sib.slots == nullptr;
    // (a row no tile of the pair holds has returned above; with PT / FULL / MAPS the mask is all ones)
    if (id < 64) {                                               // the first wave, whole: bit id = transform k0 + id
        bool act = k0 + id < k1;
        if (skip_ok && act) {
            const TemplDev* tp = templ + ra.first + k0 + id;
            const int ilo = tp->ilo, ihi = tp->ihi, jlo = tp->jlo, jhi = tp->jhi;
            const bool a0 = rowA && ri >= ilo - tA.i0 && ri <= ihi - tA.i0 && min(tA.vx - 1, jhi - tA.j0) >= max(0, jlo - tA.j0);
            const bool a1 = rowB && ri >= ilo - tB.i0 && ri <= ihi - tB.i0 && min(tB.vx - 1, jhi - tB.j0) >= max(0, jlo - tB.j0);
            act = a0 || a1;
        }
        const unsigned long long m = __ballot(act);
        if (id == 0) *actm = m;
    }
    int cj0 = rem3 - ra.Qx;
    const int cmask = ra.circ_x ? TX - 1 : -1;
    auto col_of = [&](int c) { return (cj0 + (c / R3) * 2 * S + 256 * (c % R3)) & cmask; };
    // element offset of tile column 0 of this workgroup's row in the best-record planes
    // (the row is the same for the whole workgroup: a scalar 64-bit element offset;
    //  the column adds a 32-bit byte offset per lane)
    const size_t offA = (size_t)(tA.i0 + ri - ra.cy0) * ra.cw + (tA.j0 - ra.cx0);
    const size_t offB = (size_t)(tB.i0 + ri - ra.cy0) * ra.cw + (tB.j0 - ra.cx0);
    // The two halves ("parts") of a packed output: tiles A and B under one
    // template, or - PT, the pair's second tile is empty - templates 2k and 2k+1
    // on tile A (see k_inv_cols_sym).  Everything below is written per part.
    auto tile_of = [&](int part) -> const TileDev& { return (PT || part == 0) ? tA : tB; };
    auto off_of = [&](int part) { return (PT || part == 0) ? offA : offB; };
    auto row_of = [&](int part) { return (PT || part == 0) ? rowA : rowB; };
    // running best record of the thread's cells (16: 8 columns x 2 tiles; PT: 8),
    // in registers across the templates of the launch; written back once at the
    // end, and only where a template of this launch won.  A stored NaN SNR
    // stays: nothing compares greater than it (sc_fold's sticky NaN); a NaN score
    // never wins here - it cannot arise from the finite DEMs the host lets through.
    // The winner is remembered as its index in the launch (NONE: unchanged) and by
    // its raw transform output; its amplitude xr * ka[winner] is formed at the
    // write-back (the kernel is VALU-bound: every instruction per cell counts).
    constexpr int NBEST = PT ? NC : 2 * NC;
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    // NEAR (round 6): a winner's amplitude is STORED when it wins (the record branch) instead of its transform output
    // being carried in sixteen registers to the write-back - the same product of the same two floats; with the event
    // loop's temporaries the near-tie variant then fits the plain kernel's four waves per SIMD
    constexpr bool AMPW = (NEAR && !PT && SC_I2_NEAR_LOOP && SC_I2_NEAR_AMPW) ||
                          (SC_I2_AMPW_PLAIN && !NEAR && !PT && !MAPS && !FULL && !SPLITK);      // (lab: the plain kernel too)
    static_assert(!(AMPW && SPLITK && !NEAR), "a later share's amplitudes go to its own plane");
    auto best_of = [](int c, int part) { return PT ? c : 2 * c + part; };
    float b_snr[NBEST], b_xr[NBEST];
    uint32_t b_ix[NBEST / 4];                  // one byte per cell (0xFF: unchanged)
    uint32_t nearm = 0;                        // NEAR: bit k - some template came within near_w of cell k's running best
    static_assert(!NEAR || (!MAPS && !FULL), "near-tie flags: the plain fold only");
    // SPLITK + NEAR (round 6): a later share starts from the record as the launch found it - a FLOOR without a holder - instead
    // of from nothing: its templates win (and are near-ties) against what the search has reached, not against the best of the
    // few templates of the share (a share that starts from zero meets a "record" an order of magnitude more often, and
    // names candidates the final record is far above).  The merged result is the same: the share hands on only what beat
    // the floor, and the merge takes the shares in order.
    // EVERY share of such a launch - the first too - hands its winners to k_merge_split in a scratch record of its own: a first
    // share that wrote the record itself would move the floor under the workgroups of the other shares that start later
    // (the same final record, but other near-ties listed from run to run).
    const bool later = SPLITK && (NEAR || blockIdx.z > 0);
    const size_t share_plane = SPLITK ? (size_t)(NEAR ? blockIdx.z : blockIdx.z - 1) * ra.nc : 0;
    float* const amp_dst = later ? ra.a2 + share_plane : best_amp;                            // (AMPW: where a winner's amplitude goes)
#pragma unroll
    for (int c = 0; c < NBEST / 4; ++c) b_ix[c] = NONE;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int cj = col_of(c);
#pragma unroll
        for (int part = 0; part < (PT ? 1 : 2); ++part) {
            const bool ok = !MAPS && row_of(part) && cj >= 0 && cj < tile_of(part).vx && !(SPLITK && !NEAR && blockIdx.z > 0);
            // (a cell outside the tile's valid extent holds +inf: nothing compares greater, so the
            //  templates whose window-limit rectangle covers the whole tile row need no range test
            //  per output - STATIC below; such a cell is never written back, its winner byte stays NONE)
            b_snr[best_of(c, part)] = ok ? at_bytes(best_snr + off_of(part), 4u * (uint32_t)cj)
                                         : ((SC_I2_STATIC && !MAPS && !FULL) ? __builtin_inff() : 0.f);
            b_xr[best_of(c, part)] = 0.f;
        }
    }

    const size_t ostep = ((size_t)ra.pcj * ra.ystride - NGO) * plane * sizeof(float2);   // last plane of a job -> first of the next
    v2 a[16];
    // transforms are fetched in order: fp walks the planes of a job, then steps to the same
    // pair's job of the next orientation (uniform scalars; nb = 1: a plain plane walk)
    const char* fp = src1;
    int fk = 0;
    auto fetch = [&]() {
#pragma unroll
        for (int j = 0; j < 16; ++j)                             // column tt1 + j*S
            a[j] = *reinterpret_cast<const v2*>(fp + (size_t)j * JSTRIDE1 + voff1);
        fp += plane * sizeof(float2);
        if (++fk == NGO) { fk = 0; fp += ostep; }
    };
    constexpr bool CAN_SKIP = !PT && !FULL && !MAPS;             // (else every transform is taken: the first fetch need not wait for the mask)
    int ob = 0, ok_ = 0;                                         // orientation / transform within it of the one in hand
    // transform k of the launch: job k / NGO (the same pair's block of the next orientation), plane k % NGO of it
    auto seek = [&](int k) {
        ob = k / NGO; ok_ = k - ob * NGO; fk = ok_;
        fp = src1 + ((size_t)ob * ra.pcj * ra.ystride + ok_) * plane * sizeof(float2);
    };
    if constexpr (!CAN_SKIP) {
        if constexpr (SPLITK) seek(k0);
        fetch();
    }
    lds_barrier();                                               // tables, scalars and the mask are in place
    unsigned long long am;                                       // transforms still to do (workgroup-uniform: scalar)
    {
        const uint32_t* mp = reinterpret_cast<const uint32_t*>(actm);
        am = ((unsigned long long)__builtin_amdgcn_readfirstlane(mp[1]) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(mp[0]);
    }
    if (am == 0 && !(SPLITK && blockIdx.z > 0)) {                // every template masks this row
        if (sib_mine && id == 0) sib_publish(sib_mine, ra.sib.base + SIB_DONE);
        return;
    }
    int cur = am ? k0 + (int)__builtin_ctzll(am) : -1;           // the transform in hand (the first one unless transforms are skipped)
    am &= am - 1;
    if constexpr (CAN_SKIP) {
        if constexpr (SPLITK) {
            if (cur >= 0) {
                seek(cur);
                fetch();
            }
        } else {
            fp += (size_t)cur * plane * sizeof(float2);          // (transforms are skipped in one-orientation launches only)
            fetch();
        }
    }
    if (sib_mine && id == 0) sib_publish(sib_mine, ra.sib.base + 1);
    bool sib_on = sib_mine != nullptr;
    uint32_t sib_seen = 0;
    for (int gi_ = 0; !SPLITK || cur >= 0; ++gi_) {              // gi_: transforms done so far
        const int nxt = am ? k0 + (int)__builtin_ctzll(am) : -1; // the next one this row takes part in
        am &= am - 1;
        const bool more = nxt >= 0;
        if (skip_ok) ok_ = cur;                                  // (one orientation, one template per transform)
        // the columns are two instructions away from cj0: keep them out of the
        // loop-invariant registers (eight of them would not fit)
        asm volatile("" : "+v"(cj0));
        if (sib_on && id == 0 && more) sib_seen = sib_peek(sib_theirs);

        // ---- stage 1 (radix 16, stride 1) from registers: outputs 16 tt1 + m
        {
            pk::B<16, true>::run(a);
            const v2 w1 = tw1[2 * tt1], w8 = tw1[2 * tt1 + 1], w2 = pk::cmul(w1, w1), w4 = pk::cmul(w2, w2);
            line1[0] = a[pk::B<16, true>::pos(0)];
            line1[8] = pk::cmul(a[pk::B<16, true>::pos(8)], w8);
#pragma unroll
            for (int k = 1; k < 8; ++k) {
                v2 wk = (k & 1) ? w1 : v2{1.f, 0.f};
                if (k == 2 || k == 6) wk = w2;
                if (k == 3 || k == 7) wk = pk::cmul(w1, w2);
                if (k == 4) wk = w4;
                if (k >= 5) wk = pk::cmul(wk, w4);
                line1[k] = pk::cmul(a[pk::B<16, true>::pos(k)], wk);
                line1[k + 8] = pk::cmul(a[pk::B<16, true>::pos(k + 8)], pk::cmul(wk, w8));
            }
        }
        // (the sibling must have issued fetch gi_ before this workgroup issues fetch gi_ + 1;
        //  its word was looked at before stage 1, the load has had the stage to come back)
        if (sib_on && id == 0 && more && (int32_t)(sib_seen - (ra.sib.base + gi_ + 1)) < 0)
            sib_on = sib_wait(sib_theirs, ra.sib.base + gi_ + 1);
        lds_barrier();
        if (more) fp += (size_t)(nxt - cur - 1) * plane * sizeof(float2);      // (planes of skipped transforms)
        if (FETCH_AT == 0 && more) {
            fetch();                                             // in flight through stages 2-3
            if (sib_mine && id == 0) sib_publish(sib_mine, ra.sib.base + gi_ + 2);
        }
        // ---- stage 2 (radix 16, stride 16): elements tt2 + j*S -> o + 16 m
        {
            v2 b[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) b[j] = lds_cell2<SC_I2_RD1 != 0>(rd2 + j * (S + S / 16));
            lds_barrier();
            pk::B<16, true>::run(b);
            wr2[0] = b[pk::B<16, true>::pos(0)];
#pragma unroll
            for (int m = 1; m < 16; ++m) wr2[17 * m] = pk::cmul(b[pk::B<16, true>::pos(m)], twp2[m]);
        }
        if (FETCH_AT == 1 && more) fetch();                      // in flight through stage 3
        lds_barrier();
        // ---- stage 3 (radix R3, stride 256) fused with the epilogue: every
        // output is scored as soon as its two butterflies have produced it.
        // Per part: the template, its scalars, and the cells that may score -
        // inside the tile's valid extent AND (lean variant) inside the template's
        // window-limit rectangle, as one unsigned range test on the column; a row
        // outside (or a missing second template) makes the range empty.
        const TemplDev* tpp[2];
        float ka[2], kt[2], kx2[2], fl0[2], inv_n[2];
        unsigned span[2];
        int base[2];
        uint32_t tix[2];
        bool covers[2];
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            const int tgo = PT ? 2 * ok_ + part : ok_;            // template within its orientation
            const bool have = !PT || tgo < ra.G;
            const int tgc = ob * ra.G + (have ? tgo : 2 * ok_);   // template of the launch
            const TemplDev* tp = templ + ra.first + tgc;
            const TileDev& t = tile_of(part);
            const float* e = epi + EPI_FLOATS * tgc;
            tpp[part] = tp;
            ka[part] = e[0]; kt[part] = e[1]; kx2[part] = e[2]; fl0[part] = e[3]; inv_n[part] = e[4];
            tix[part] = (uint32_t)tgc * 0x01010101u;
            int lo = 0, hi = t.vx - 1;
            bool r = row_of(part) && have;
            if (!FULL && !MAPS) {
                lo = max(lo, tp->jlo - t.j0);
                hi = min(hi, tp->jhi - t.j0);
                r = r && ri >= tp->ilo - t.i0 && ri <= tp->ihi - t.i0;
            }
            span[part] = (r && hi >= lo) ? (unsigned)(hi - lo) : 0u;
            base[part] = (r && hi >= lo) ? lo : 0x40000000;      // no column reaches it
            // STATIC: every cell of the part's valid extent may score (the rectangle covers columns
            // 0 .. vx-1 of this row), or none can whatever the test says (the row lies outside the tile:
            // its cells hold +inf)
            covers[part] = !row_of(part) || (r && lo == 0 && hi == t.vx - 1);
        }
        const bool all_in = SC_I2_STATIC && !FULL && !MAPS && covers[0] && covers[1];     // workgroup-uniform
        const float near_lo = NEAR ? 1.f - 1.0001f * ra.near_w - 2e-7f : 1.f;             // (NEAR, deferred form: candidates score above the record times this)
        const float near_up = 1.f - ra.near_w;                                            // (a winner times this at or below the record it beat: a near-tie)
        auto stage3 = [&](auto static_tag) {
        constexpr bool STATIC = decltype(static_tag)::value;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            v2 vw[R3], vm[R3];
#pragma unroll
            for (int k = 0; k < R3; ++k) {
                vw[k] = lds_cell2<SC_I2_RD1 != 0>(lineW + u * (2 * S + 2 * S / 16) + k * 272);
                vm[k] = lds_cell2<SC_I2_RD1 != 0>(lineM + u * (2 * S + 2 * S / 16) + k * 272);
            }
            pk::B<R3, true>::run(vw);
            pk::B<R3, true>::run(vm);
            // (RARE: every row length since the end of round 5.  The 512-cell kernels kept the selects while the deferred form
            //  cost them a scratch reload inside the template loop; it no longer does - 16 B outside the loop in the dealt-out
            //  form - and their row pass is 4.5 % faster with it: C1F 35.2 -> 34.9 ms, profiles/r05_launch_forms.txt)
            // (NEAR, SC_I2_NEAR_RARE: the branch also stands for the near-ties - a score above the record LESS the window)
            constexpr bool RARE = SC_I2_RAREWIN && (TX >= 1024 || SC_I2_RARE512) && (!NEAR || SC_I2_NEAR_RARE);
            // (GRP outputs pairs per branch: the fewer cells a branch stands for, the more rarely it is taken)
            constexpr int GRP = (RARE && SC_I2_GRP < R3) ? SC_I2_GRP : R3;
#pragma unroll
            for (int m0 = 0; m0 < R3; m0 += GRP) {
            bool wonm[GRP][2];                 // lane masks (scalar register pairs): which outputs won their cell
            float snrs[GRP][2];
            bool anyw = false;
#pragma unroll
            for (int m = m0; m < m0 + GRP; ++m) {
                const int c = u * R3 + m;
                const v2 xc = vw[pk::B<R3, true>::pos(m)], t3 = vm[pk::B<R3, true>::pos(m)];
                const int cj = STATIC ? 0 : col_of(c);
#pragma unroll
                for (int part = 0; part < 2; ++part) {
                    const int k = best_of(c, part);
                    const TileDev& t = tile_of(part);
                    const TemplDev* tp = tpp[part];
                    const float xr = part ? xc.y : xc.x, tr = part ? t3.y : t3.x;
                    // core.py:360-367 with the float32 resolution floor (sc_epilogue)
                    const float T1 = xr * xr * kt[part];
                    const float d = fmaxf(fmaf(tr, scale, -T1), fmaf(fabsf(xr), kx2[part], fl0[part]));
                    float snr = fabsf(T1 * __builtin_amdgcn_rcpf(fmaf(d, inv_n[part], (float)SC_EPS)));
                    const bool in = STATIC || (unsigned)(cj - base[part]) <= span[part];
                    if (MAPS || FULL) {
                        float amp = xr * ka[part];
                        // `in` is the tile's valid extent here; masks per cell
                        if (in) {
                            if (FULL)
                                sc_apply_masks(*tp, g, xaxis, yaxis, t.i0 + ri, t.j0 + cj, amp, snr);
                            else if (!(ri >= tp->ilo - t.i0 && ri <= tp->ihi - t.i0 &&
                                       cj >= tp->jlo - t.j0 && cj <= tp->jhi - t.j0)) {
                                amp = 0.f;
                                snr = 0.f;
                            }
                            if (MAPS) {
                                at_bytes(map_amp + off_of(part), 4u * (uint32_t)cj) = amp;
                                at_bytes(map_snr + off_of(part), 4u * (uint32_t)cj) = snr;
                            }
                        } else {
                            snr = 0.f;
                        }
                    }
                    if (!MAPS) {
                        // sc_fold: take if greater, ties keep the incumbent; a cell outside
                        // the range (lean variant) or masked to 0 never wins
                        const bool won = (FULL || in) && snr > b_snr[k];
                        if constexpr (NEAR && !RARE) {
                            // a score within the window of the running best, either side of it - EQUAL scores included since the
                            // end of round 5: two templates proportional to each other on a degenerate support (a Ricker window two
                            // cells wide) score the same bits and differ at 1e-8 in float64; the grid's end twins (-pi/2, +pi/2: one
                            // template) cost the event route two float64 pairs per cell they hold
                            const float top = fmaxf(snr, b_snr[k]);
                            const bool nt = in && snr > 0.f && fabsf(snr - b_snr[k]) <= ra.near_w * top;
                            nearm |= nt ? (1u << k) : 0u;
                            if (nt && ra.ev) {             // (before the record moves: b_ix still names the holder)
                                const unsigned long long slot = atomicAdd(ra.ev_count, 1ull);
                                if (slot < ra.ev_cap) {
                                    const uint32_t hx = (b_ix[k >> 2] >> (8 * (k & 3))) & 0xFFu;
                                    uint32_t* e = ra.ev + SC_EVENT_WORDS * slot;
                                    e[0] = (uint32_t)(off_of(part) + (size_t)cj);
                                    e[1] = tp->id;
                                    // a holder from an earlier launch (or none yet: SC_ID_NONE) stands in the record's id plane
                                    e[2] = hx != 0xFFu ? templ[ra.first + hx].id : at_bytes(best_id + off_of(part), 4u * (uint32_t)cj);
                                    e[3] = __float_as_uint(top);
                                }
                            }
                        }
                        if constexpr (RARE) {
                            // (NEAR: a candidate - a win, or a score inside the window below the record; near_lo = 1 - near_w less a hair)
#ifdef SC_I2_NEAR_NOMUL                    // timing probe only: the near-tie variant without its window (flags nothing below the record)
                            const bool cand = won;
#else
                            const bool cand = NEAR ? (in && snr > b_snr[k] * near_lo) : won;
#endif
                            wonm[m - m0][part] = cand;
                            snrs[m - m0][part] = snr;
                            anyw = anyw || cand;
                        } else {
                            b_snr[k] = won ? snr : b_snr[k];
                            b_xr[k] = won ? xr : b_xr[k];
                            const uint32_t bm = 0xFFu << (8 * (k & 3));
                            b_ix[k >> 2] = won ? ((b_ix[k >> 2] & ~bm) | (tix[part] & bm)) : b_ix[k >> 2];
                        }
                    }
                }
            }
            // A cell is won a handful of times in a whole search: the record - best SNR, the winner's transform
            // output, its index - is updated under a wave-uniform branch, taken when one of the GRP x 128 cells the
            // branch stands for was won, instead of with three selects and two bit operations per output of every
            // template (the kernel is bound by vector issue).  The compare against the record is repeated under the
            // branch: where two templates ride one transform (PT) the second meets the first's update there.
            // Same order, same values: the record is identical.
            if (RARE && !MAPS && __builtin_amdgcn_ballot_w64(anyw) != 0ull) {
                // NEAR, one template per transform (round 6): the events of a branch's near-ties come from a LOOP over a mask
                // after the record's update (sixteen unrolled emission sites, each with its atomic and its addresses, cost the
                // kernel 41 registers and its fourth wave per SIMD: 168 -> 128).
                constexpr bool NEAR_LOOP = NEAR && !PT && SC_I2_NEAR_LOOP;
                // NEAR_LOOP: which of the branch's outputs are near-ties costs no vector instruction of its own - a candidate
                // that does NOT win lies inside the window below the record (that is what made it a candidate); one that
                // wins is tested against the window's upper side under the branch its amplitude's store takes anyway.
                // Bits of ntm: the lane's near-tie outputs; old_ix: the holders as the branch found them (an event names
                // the holder it met).  The branch is NOT rare (a cell is won dozens of times in a search: most waves
                // see a win per template); a near-tie is - three in a million outputs.
                uint32_t ntm = 0;
                float top1 = 0.f;
                uint32_t old_ix[NBEST / 4];
                if constexpr (NEAR_LOOP) {
#pragma unroll
                    for (int q = 0; q < NBEST / 4; ++q) old_ix[q] = b_ix[q];
                }
#pragma unroll
                for (int m = m0; m < m0 + GRP; ++m) {
                    const v2 xc = vw[pk::B<R3, true>::pos(m)];
#pragma unroll
                    for (int part = 0; part < 2; ++part) {
                        const int k = best_of(u * R3 + m, part);
                        const float snr = snrs[m - m0][part];
                        if constexpr (NEAR && !NEAR_LOOP) {
                            // the near-tie test of the select form on the candidates (all of them inside the cell's range), against
                            // the record as it stands now - where two templates ride one transform the second meets the first's update
                            const float top = fmaxf(snr, b_snr[k]);
                            const bool nt = wonm[m - m0][part] && snr > 0.f && fabsf(snr - b_snr[k]) <= ra.near_w * top;
                            nearm |= nt ? (1u << k) : 0u;
                            if (nt && ra.ev) {
                                const unsigned long long slot = atomicAdd(ra.ev_count, 1ull);
                                if (slot < ra.ev_cap) {
                                    const int cj = col_of(u * R3 + m);
                                    const uint32_t hx = (b_ix[k >> 2] >> (8 * (k & 3))) & 0xFFu;
                                    uint32_t* e = ra.ev + SC_EVENT_WORDS * slot;
                                    e[0] = (uint32_t)(off_of(part) + (size_t)cj);
                                    e[1] = tpp[part]->id;
                                    e[2] = hx != 0xFFu ? templ[ra.first + hx].id : at_bytes(best_id + off_of(part), 4u * (uint32_t)cj);
                                    e[3] = __float_as_uint(top);
                                }
                            }
                        }
                        const bool won = wonm[m - m0][part] && snr > b_snr[k];
                        // (SC_I2_SITE, lab: the record moves under a branch of its own per output.  A cell is won nine times in a
                        //  search of 6335 templates - ln 6335: the float32 SNRs come in no particular order, whatever order
                        //  the orientations are searched in - 0.14 % of all (cell, template) pairs; the branch above stands
                        //  for 1024 cells of the wave and is taken three times in four, one output's 64 cells hold a win one
                        //  time in twelve.  Skipping the update's 80 vector instructions eleven times in twelve is worth
                        //  nothing: C3 3.16 -> 3.16 .. 3.22 s with it, five record entries in scratch -
                        //  profiles/r06_row_pass_probes.txt: the row pass is not bound by its vector instruction count)
                        const bool any_won = !SC_I2_SITE || PT || __builtin_amdgcn_ballot_w64(won) != 0ull;
                        if constexpr (NEAR_LOOP) {
                            bool nt = wonm[m - m0][part] && !won;                 // below the record, inside the window (equal scores too)
                            if (any_won) {
                                nt = nt || (won && snr * near_up <= b_snr[k]);    // won, and the record it beat lies inside the window
                                if constexpr (AMPW) {
                                    if (won)
                                        at_bytes(amp_dst + off_of(part), 4u * (uint32_t)col_of(u * R3 + m)) = (part ? xc.y : xc.x) * ka[part];
                                } else {
                                    b_xr[k] = won ? (part ? xc.y : xc.x) : b_xr[k];
                                }
                            }
                            if (__builtin_amdgcn_ballot_w64(nt) != 0ull) {
                                top1 = (nt && ntm == 0u) ? fmaxf(snr, b_snr[k]) : top1;     // (b_snr[k]: still the record the output met)
                                ntm |= nt ? (1u << (2 * (m - m0) + part)) : 0u;
                            }
                        }
                        if (any_won) {
                            b_snr[k] = won ? snr : b_snr[k];
                            if constexpr (!NEAR_LOOP) {
                                if constexpr (AMPW) {
                                    if (won)
                                        at_bytes(amp_dst + off_of(part), 4u * (uint32_t)col_of(u * R3 + m)) = (part ? xc.y : xc.x) * ka[part];
                                } else {
                                    b_xr[k] = won ? (part ? xc.y : xc.x) : b_xr[k];
                                }
                            }
                            const uint32_t bm = 0xFFu << (8 * (k & 3));
                            b_ix[k >> 2] = won ? ((b_ix[k >> 2] & ~bm) | (tix[part] & bm)) : b_ix[k >> 2];
                        }
                    }
                }
                if constexpr (NEAR_LOOP) {
                    // the events, after the update: the outputs' scores and values are dead by now - the loop's temporaries
                    // take their registers instead of a fourth wave per SIMD (round 6: 168 registers -> 128)
                    bool first_ev = true;
                    while (ntm) {                          // (per lane; a near-tie is three in a million outputs)
                        const int bit = __builtin_ctz(ntm);
                        ntm &= ntm - 1;
                        const int part = bit & 1, c = u * R3 + m0 + (bit >> 1), k = 2 * c + part;
                        nearm |= 1u << k;
                        if (ra.ev) {
                            const unsigned long long slot = atomicAdd(ra.ev_count, 1ull);
                            if (slot < ra.ev_cap) {
                                const uint32_t hx = (old_ix[k >> 2] >> (8 * (k & 3))) & 0xFFu;
                                const int cj = col_of(c);
                                const size_t cell = (part ? offB : offA) + (size_t)cj;
                                // the larger of the two scores: taken where the lane's FIRST near-tie of the branch was found; a second
                                // one in the same lane and branch (one in 1e11) says +inf - never dropped by the settle
                                const float top = first_ev ? top1 : __builtin_inff();
                                first_ev = false;
                                uint32_t* e = ra.ev + SC_EVENT_WORDS * slot;
                                e[0] = (uint32_t)cell;
                                e[1] = part ? tpp[1]->id : tpp[0]->id;
                                // a holder from an earlier launch (or none yet: SC_ID_NONE) stands in the record's id plane
                                e[2] = hx != 0xFFu ? templ[ra.first + hx].id : best_id[cell];
                                e[3] = __float_as_uint(top);
                            }
                        }
                    }
                }
            }
            }
        }
        };
#if SC_I2_STATIC == 2                     // timing probe only: every template on the static path (wrong at the DEM's borders)
        stage3(std::true_type{});
#else
        if (all_in) stage3(std::true_type{});
        else stage3(std::false_type{});
#endif
        lds_barrier();
        if (!more) break;
        cur = nxt;
        if (++ok_ == NGO) { ok_ = 0; ++ob; }
    }
    if (sib_mine && id == 0) sib_publish(sib_mine, ra.sib.base + SIB_DONE);
    if (!MAPS) {
        // resolution statistic (sc_get_resolution_stats): of the cells this launch's templates won, how
        // many hold a residual T3 - T1 within 256 floors of the float32 resolution floor (sc_epi_floor) -
        // there the SNR is off by more than the stated tolerance (error ~ floor / residual) and the
        // argmax is the transform's rounding noise.  Recovered from the record at the write-back
        // (d = (T1 / snr - eps) n), not in the template loop.
        int n_won = 0, n_near = 0;
        // one won cell into the statistic: template ix of the launch holds it with raw output b_xr[k], SNR b_snr[k]
        auto count = [&](int k, const float* e, float xr_) {
            const float T1 = xr_ * xr_ * e[1], fl = fmaf(fabsf(xr_), e[2], e[3]);
            const float d = (T1 / b_snr[k] - (float)SC_EPS) / e[4];
            ++n_won;
            n_near += (fl > 0.f && d < 256.f * fl) ? 1 : 0;
        };
        bool later_share = false;
        if constexpr (SPLITK) {
            if (later) {
                // a later share of the launch's transforms: its own record, EVERY valid cell written (zero where
                // none of its templates scored), for k_merge_split to fold into the record in order.  Its winners
                // go into the statistic like the first share's (a cell counts once per share that scored on it:
                // the statistic is a fraction of wins, and every share's wins are wins of this launch's templates)
                later_share = true;
                const size_t pl_ = share_plane;
#pragma unroll
                for (int k = 0; k < NBEST; ++k) {
                    const int c = PT ? k : k >> 1, part = PT ? 0 : (k & 1);
                    const int cj = col_of(c);
                    if (!(row_of(part) && cj >= 0 && cj < tile_of(part).vx)) continue;
                    const uint32_t ix = (b_ix[k >> 2] >> (8 * (k & 3))) & 0xFFu;
                    const bool won = ix != 0xFFu;
                    const uint32_t o = 4u * (uint32_t)cj;
                    const float* e = epi + EPI_FLOATS * (won ? ix : (uint32_t)tlo);
                    at_bytes(ra.s2 + pl_ + off_of(part), o) = won ? b_snr[k] : 0.f;
                    if constexpr (AMPW) {                    // (a winner's amplitude is in the share's plane since it won)
                        if (!won) at_bytes(ra.a2 + pl_ + off_of(part), o) = 0.f;
                    } else {
                        at_bytes(ra.a2 + pl_ + off_of(part), o) = won ? b_xr[k] * e[0] : 0.f;
                    }
                    at_bytes(ra.i2 + pl_ + off_of(part), o) = won ? templ[ra.first + (won ? ix : 0)].id : SC_ID_NONE;
                    if (won && b_snr[k] > 0.f) count(k, e, AMPW ? at_bytes(ra.a2 + pl_ + off_of(part), o) / e[0] : b_xr[k]);
                }
            }
        }
        if constexpr (NEAR) {
#pragma unroll
            for (int k = 0; k < NBEST; ++k)
                if (nearm & (1u << k)) {
                    const int c = PT ? k : k >> 1, part = PT ? 0 : (k & 1);
                    at_bytes(ra.near + off_of(part), (uint32_t)col_of(c)) = (uint8_t)1;
                }
        }
        if (!later_share) {
#pragma unroll
        for (int k = 0; k < NBEST; ++k) {
            const uint32_t ix = (b_ix[k >> 2] >> (8 * (k & 3))) & 0xFFu;
            if (ix != 0xFFu) {
                const int c = PT ? k : k >> 1, part = PT ? 0 : (k & 1);
                const uint32_t o = 4u * (uint32_t)col_of(c);
                const float* e = epi + EPI_FLOATS * ix;
                at_bytes(best_snr + off_of(part), o) = b_snr[k];
                if constexpr (!AMPW) at_bytes(best_amp + off_of(part), o) = b_xr[k] * e[0];
                at_bytes(best_id + off_of(part), o) = templ[ra.first + ix].id;
                // (AMPW: the statistic's transform output back from the stored amplitude - a count, not a result)
                count(k, e, AMPW ? at_bytes(best_amp + off_of(part), o) / e[0] : b_xr[k]);
            }
        }
        }
        if (ra.stats) {
            for (int sft = 32; sft > 0; sft >>= 1) {
                n_won += __shfl_down(n_won, sft, 64);
                n_near += __shfl_down(n_near, sft, 64);
            }
            if ((id & 63) == 0 && n_won) {
                atomicAdd(ra.stats, (unsigned long long)n_won);
                atomicAdd(ra.stats + 1, (unsigned long long)n_near);
            }
        }
    }
}

// The shares h = 1 .. nparts of a split row pass (k_inv_rows_fast, SPLITK) into the record, in order: a later
// share's template takes a cell only where it scored strictly higher - what the one fold over all of the
// launch's templates does (sc_fold).  Cells the launch did not cover hold what an earlier launch of the
// search left there, already merged: nothing is greater than itself.
template <bool NEAR>
__global__ void __launch_bounds__(256)
k_merge_split(float* __restrict__ best_snr, float* __restrict__ best_amp, uint32_t* __restrict__ best_id,
              float* __restrict__ s2, const float* __restrict__ a2, const uint32_t* __restrict__ i2,
              size_t nc, int nparts, float near_w, uint8_t* __restrict__ near, unsigned long long* __restrict__ ev_count,
              uint32_t* __restrict__ ev, unsigned long long ev_cap) {
    // NEAR (round 6): a share's winner within the window of the record it meets here, either side, is a near-tie like any
    // other - flagged and listed (cell, the share's template, the record's holder) - and a share's entry is consumed (zeroed)
    // once read: a cell a later launch does not cover would otherwise meet its own old score again, as a tie, launch after launch.
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nc; i += (size_t)gridDim.x * 256) {
        float bs = best_snr[i], ba = 0.f;
        uint32_t bi = 0;
        bool took = false;
        for (int h = 0; h < nparts; ++h) {
            const float s = s2[(size_t)h * nc + i];
            if (NEAR && s > 0.f) {
                s2[(size_t)h * nc + i] = 0.f;
                if (fabsf(s - bs) <= near_w * fmaxf(s, bs)) {
                    near[i] = (uint8_t)1;
                    const unsigned long long slot = atomicAdd(ev_count, 1ull);
                    if (slot < ev_cap) {
                        uint32_t* e = ev + SC_EVENT_WORDS * slot;
                        e[0] = (uint32_t)i;
                        e[1] = i2[(size_t)h * nc + i];
                        e[2] = took ? bi : best_id[i];
                        e[3] = __float_as_uint(fmaxf(s, bs));
                    }
                }
            }
            if (s > bs) { bs = s; ba = a2[(size_t)h * nc + i]; bi = i2[(size_t)h * nc + i]; took = true; }
        }
        if (took) { best_snr[i] = bs; best_amp[i] = ba; best_id[i] = bi; }
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
bool fft_size_supported(int T) {
    return T >= 64 && T <= 4096 && (T & (T - 1)) == 0;
}

// Entries 0 .. T-1: the forward twiddles exp(-2 pi i k / T).  Entries T .. 2T-1: the phase
// factor exp(i pi kph f / T) of the symmetric-template path along this axis (phase_tab).
static int upload_twiddles(sc_ctx* ctx, DevBuf& buf, int& have, int T, int kph) {
    const int key = T * 8 + (kph & 7);
    if (have == key) return SC_OK;
    std::vector<float2> h(2 * (size_t)T);
    for (int k = 0; k < T; ++k) {
        double a = -2.0 * M_PI * (double)k / (double)T;
        h[k] = make_float2((float)cos(a), (float)sin(a));
        // kph * k mod 2T keeps the argument small (exact in integers)
        long long m = ((long long)kph * k) % (2LL * T);
        double b = M_PI * (double)m / (double)T;
        h[T + k] = make_float2((float)cos(b), (float)sin(b));
    }
    int rc = sc_ensure(ctx, buf, sizeof(float2) * h.size());
    if (rc) return rc;
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SC_HIP(ctx, hipMemcpy(buf.p, h.data(), sizeof(float2) * h.size(), hipMemcpyHostToDevice));
    have = key;
    return SC_OK;
}

static int npairs_of(const FftGeom& fg) { return (fg.ntiles + 1) / 2; }

void fft_spectra_forget(sc_ctx* ctx) {
    std::fill(ctx->spec_key.begin(), ctx->spec_key.end(), NAN);
}

// n_slots: orientations of the search whose curvature spectra would be kept (0: none asked for)
int fft_prepare(sc_ctx* ctx, const FftGeom& fg, int n_templ_chunk, int group, int nb, int n_slots) {
    // nb: orientations batched per launch (fft_batch_orientations); 1 for the large searches
    if (!fft_size_supported(fg.Ty) || !fft_size_supported(fg.Tx))
        return sc_fail(ctx, SC_ERR_UNSUPPORTED, "FFT tile %dx%d not supported", fg.Ty, fg.Tx);
    int rc;
    if ((rc = upload_twiddles(ctx, ctx->tw_y, ctx->tw_Ty, fg.Ty, 1 - ctx->g.oy))) return rc;
    if ((rc = upload_twiddles(ctx, ctx->tw_x, ctx->tw_Tx, fg.Tx, 1 - ctx->g.ox))) return rc;
    const Geom& g = ctx->g;
    int np = npairs_of(fg);
    std::vector<TileDev> h(2 * np);
    for (int k = 0; k < 2 * np; ++k) {
        TileDev t{0, 0, 0, 0, 0, 0};
        if (k < fg.ntiles) {
            int ty = k / fg.ntx, tx = k % fg.ntx;
            t.i0 = g.cy0 + ty * fg.Vy;
            t.j0 = g.cx0 + tx * fg.Vx;
            t.vy = std::min(fg.Vy, g.cy1 - t.i0);
            t.vx = std::min(fg.Vx, g.cx1 - t.j0);
            t.gi0 = t.i0 + g.oy - fg.Py;
            t.gj0 = t.j0 + g.ox - fg.Qx;
        }
        h[k] = t;
    }
    {
        // (uploaded only when it differs from what the device holds: a multi-scale job plans the same
        //  tiles search after search; the host copy lives in the context, the upload is asynchronous)
        const size_t bytes = sizeof(TileDev) * h.size();
        const void* before = ctx->tiles.p;
        if ((rc = sc_ensure(ctx, ctx->tiles, bytes))) return rc;
        if (before != ctx->tiles.p || ctx->h_tiles.size() != bytes || memcmp(ctx->h_tiles.data(), h.data(), bytes) != 0) {
            SC_HIP(ctx, hipStreamSynchronize(ctx->stream));      // (h_tiles may still be the source of a copy in flight)
            ctx->h_tiles.assign((const unsigned char*)h.data(), (const unsigned char*)h.data() + bytes);
            SC_HIP(ctx, hipMemcpyAsync(ctx->tiles.p, ctx->h_tiles.data(), bytes, hipMemcpyHostToDevice, ctx->stream));
        }
    }
    size_t plane = (size_t)fg.Ty * fg.Tx * sizeof(float2);
    size_t nblk = std::max((size_t)2 * np * nb, (size_t)n_templ_chunk);
    if ((rc = sc_ensure(ctx, ctx->blk, plane * nblk))) return rc;
    // spectra kept across searches: a slot per orientation, if the option's budget holds them
    {
        const Geom& gg = ctx->g;
        const long long sig[24] = {fg.Ty, fg.Tx, fg.Vy, fg.Vx, fg.nty, fg.ntx, fg.circ_y, fg.circ_x, fg.Py, fg.Qx,
                                   gg.ly, gg.lx, gg.gy0, gg.gx0, gg.ny, gg.nx, gg.cy0, gg.cy1, gg.cx0, gg.cx1,
                                   gg.wrap, n_slots, 0, 0};
        bool keep = n_slots > 0 && 2.0 * (double)plane * np * n_slots <= ctx->spec_mb * 1048576.0;
        if (keep) {
            // (and only while what it adds is a small part of the memory that is free: the slots are a
            //  convenience, a search must not fail for them)
            const double add = 2.0 * (double)plane * np * n_slots - (double)(ctx->uc.cap + ctx->uc2.cap);
            size_t free_b = 0, total_b = 0;
            if (add > 0.0 && (hipMemGetInfo(&free_b, &total_b) != hipSuccess || add > 0.25 * (double)free_b))
                keep = false;
        }
        const int slots = keep ? n_slots : 0;
        const size_t have = std::max(nb, slots);
        // a search that keeps nothing does not sit on the slots of an earlier one that did (up to 8 GiB
        // taken from other contexts on this GPU and from this search's own hand-off buffers)
        if (ctx->uc.cap > 2 * plane * np * have + ((size_t)256 << 20)) {
            SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
            for (DevBuf* b : {&ctx->uc, &ctx->uc2}) {
                if (b->p) SC_HIP(ctx, hipFree(b->p));
                b->p = nullptr;
                b->cap = 0;
            }
        }
        const void *p0 = ctx->uc.p, *p1 = ctx->uc2.p, *p2 = ctx->norms.p;
        if ((rc = sc_ensure(ctx, ctx->norms, sizeof(double) * 2 * np * have))) return rc;
        if ((rc = sc_ensure(ctx, ctx->uc, plane * np * have))) return rc;
        if ((rc = sc_ensure(ctx, ctx->uc2, plane * np * have))) return rc;
        if (!keep || slots != ctx->spec_slots || memcmp(sig, ctx->spec_sig, sizeof(sig)) != 0 ||
            p0 != ctx->uc.p || p1 != ctx->uc2.p || p2 != ctx->norms.p) {
            ctx->spec_key.assign((size_t)3 * slots, NAN);
            memcpy(ctx->spec_sig, sig, sizeof(sig));
        }
        ctx->spec_slots = slots;
        ctx->spec_uc_stride = (size_t)np * fg.Ty * fg.Tx;
        ctx->spec_norm_stride = (size_t)2 * np;
        ctx->uc_off = ctx->norms_off = 0;
    }
    if ((rc = sc_ensure(ctx, ctx->norm_part, sizeof(double) * 2 * np * nb * (fg.Ty / 4)))) return rc;
    if ((rc = sc_ensure(ctx, ctx->curv, sizeof(float) * (size_t)ctx->g.ly * ctx->g.lx * nb))) return rc;
    if ((rc = sc_ensure(ctx, ctx->vh, plane * n_templ_chunk))) return rc;
    size_t hplane = half_plane(fg.Ty, fg.Tx) * sizeof(float2);
    if ((rc = sc_ensure(ctx, ctx->wh, hplane * n_templ_chunk))) return rc;
    if ((rc = sc_ensure(ctx, ctx->mh, hplane * n_templ_chunk))) return rc;
    // Y blocks of several tile pairs per inverse launch (see RowArgs): as many as
    // a quarter of the free device memory (at most 32 GB; sc_set_option "y_gb" overrides) holds
    int pb = 1;
    {
        size_t free_b = 0, total_b = 0;
        double budget = 0.0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            budget = std::min(32e9, 0.25 * (double)(free_b + ctx->yw.cap + ctx->ym.cap));
        if (ctx->y_gb > 0.0) budget = ctx->y_gb * 1e9;
        double per_pair = 2.0 * (double)plane * group * nb;      // a pair's Y blocks of all batched orientations
        pb = (int)std::max(1.0, std::min((double)np, floor(budget / per_pair)));
    }
    ctx->fft_pb = pb;
    if ((rc = sc_ensure(ctx, ctx->yw, plane * group * pb * nb))) return rc;
    if ((rc = sc_ensure(ctx, ctx->ym, plane * group * pb * nb))) return rc;
    return SC_OK;
}

// Orientation batching.  A search whose single orientation does not fill the chip (one or
// two tiles, one age: BASELINE configs C1, C2, C5) is launch-bound: ten launches of a few
// microseconds of work per orientation.  Such searches send nb orientations through every
// launch: nb curvature planes and spectra, the templates of all of them in one forward pass,
// the inverse column pass with one job per (orientation, tile pair), and the row pass folding
// nb * n templates in the order the orientations come - the same cells in the same order as
// orientation by orientation, so the result is bit-identical.  Conditions: the fast row
// kernel (its scalar table and winner byte hold SC_MAX_GROUP templates), all templates of an
// orientation in one inverse launch (group >= n), and at most ~4096 column workgroups.
int fft_batch_orientations(const sc_ctx* ctx, const FftGeom& fg, int n_per, int group) {
    if (ctx->batch_off || n_per < 1 || n_per > group) return 1;
    const bool fast = (fg.Tx == 512 || fg.Tx == 1024 || fg.Tx == 2048) && ctx->variant != 9;
    if (!fast) return 1;
    const int np = (fg.ntiles + 1) / 2;
    // What one launch sequence may carry.  The row pass folds at most SC_MAX_GROUP templates per launch: where three
    // or more orientations fit that (C2: six of ten templates), the batch stops there and the row pass takes it in ONE
    // launch (more, sliced over two row-pass launches, cost C2's row pass 14 %); where they do not (C1F: 35 ages, one
    // orientation per row-pass launch either way) the forward passes and the column pass batch up to SC_MAX_BATCH
    // templates - 181 five-kernel launch sequences of a few microseconds each were 15 of C1F's 52 ms, now 6
    const int cap = ctx->batch_templ > 0 ? ctx->batch_templ : SC_MAX_BATCH;
    const int by_table = (SC_MAX_GROUP / n_per >= 3 || cap <= SC_MAX_GROUP) ? std::min(cap, SC_MAX_GROUP) / n_per : cap / n_per;
    // (round 4: 4096 workgroups and up to SC_MAX_ORIENT orientations - BASELINE config C5, one template per
    //  orientation on a 512 x 512 tile, runs 64 orientations per launch sequence instead of 32: 5.95 -> 5.2 ms;
    //  C1 and C2 are where they were with either)
    const int by_fill = (ctx->batch_fill > 0 ? ctx->batch_fill : 4096) / std::max(1, np * (fg.Tx / 8));
    return std::max(1, std::min(std::min(by_table, by_fill), SC_MAX_ORIENT));
}

// Rendezvous words for a launch of n workgroups (SibSync): one buffer per context, a new epoch per
// launch - words left by earlier launches compare below the new base, so nothing is cleared
// between launches; the buffer is zeroed when it grows and when the 24-bit epoch wraps.
static int sib_slots(sc_ctx* ctx, size_t n, SibSync& out) {
    const size_t bytes = n * sizeof(uint32_t);
    if (bytes > ctx->sib_buf.cap || ctx->sib_epoch >= 0xFFFFFEu) {
        int rc = sc_ensure(ctx, ctx->sib_buf, std::max(bytes, (size_t)1 << 20));
        if (rc) return rc;
        SC_HIP(ctx, hipMemsetAsync(ctx->sib_buf.p, 0, ctx->sib_buf.cap, ctx->stream));
        ctx->sib_epoch = 0;
    }
    out.slots = (uint32_t*)ctx->sib_buf.p;
    out.base = (++ctx->sib_epoch) << 8;
    return SC_OK;
}

template <typename K>
static int set_lds(sc_ctx* ctx, K kernel, size_t bytes) {
    return sc_lds_attr(ctx, (const void*)kernel, bytes);
}

#define DISPATCH_T(T, FN)                                                     \
    switch (T) {                                                              \
        case 64: FN(64); break;                                               \
        case 128: FN(128); break;                                             \
        case 256: FN(256); break;                                             \
        case 512: FN(512); break;                                             \
        case 1024: FN(1024); break;                                           \
        case 2048: FN(2048); break;                                           \
        case 4096: FN(4096); break;                                           \
        default: return sc_fail(ctx, SC_ERR_UNSUPPORTED, "tile size %d", T);  \
    }

static int launch_fwd_cols(sc_ctx* ctx, const FftGeom& fg, int nplanes,
                           float2* out0, float2* out1, int split2,
                           const TemplDev* templ = nullptr) {
    size_t lds = fft_lds_bytes(fg.Ty);
    dim3 grid(fg.Tx / 4, nplanes);
    sc_prof_begin(ctx, SC_K_FWD_COLS);
#define FN(T)                                                                  \
    {                                                                          \
        int rc = set_lds(ctx, k_fwd_cols<T>, lds);                             \
        if (rc) return rc;                                                     \
        hipLaunchKernelGGL(k_fwd_cols<T>, grid, dim3(fft_threads(T)), lds,     \
                           ctx->stream, (const float2*)ctx->blk.p, fg.Tx,      \
                           (const float2*)ctx->tw_y.p, out0, out1, split2,     \
                           templ);                                             \
    }
    DISPATCH_T(fg.Ty, FN)
#undef FN
    sc_prof_end(ctx);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

// coef: the nb orientations' (cc, sc2, ss) - the curvature is mixed from the stencil planes inside the
// row kernel (k_fwd_rows_curv<.., true>); nullptr: the plane(s) k_curv_alpha left in ctx->curv are read
int fft_forward_curv(sc_ctx* ctx, const FftGeom& fg, int nb, const float (*coef)[3]) {
    int np = npairs_of(fg);
    size_t lds = fft_lds_bytes(fg.Tx);
    dim3 grid(fg.Ty / 4 / FWD_ROWS_RBW, np * nb);
    CurvMix mixc;
    memset(&mixc, 0, sizeof(mixc));
    if (coef) {
        if (nb > SC_MAX_ORIENT) return sc_fail(ctx, SC_ERR_INVALID, "curvature batch of %d planes", nb);
        for (int b = 0; b < nb; ++b)
            for (int j = 0; j < 3; ++j) mixc.c[b][j] = coef[b][j];
    }
    sc_prof_begin(ctx, SC_K_FWD_ROWS);
#define FN2(T, MIXV)                                                           \
    {                                                                          \
        int rc = set_lds(ctx, k_fwd_rows_curv<T, MIXV>, lds);                  \
        if (rc) return rc;                                                     \
        hipLaunchKernelGGL((k_fwd_rows_curv<T, MIXV>), grid, dim3(fft_threads(T)), \
                           lds, ctx->stream, (const float*)(MIXV ? ctx->A.p : ctx->curv.p), \
                           (const float*)ctx->B.p, (const float*)ctx->C.p, mixc, \
                           ctx->g, (const TileDev*)ctx->tiles.p, fg.Ty,        \
                           (const float2*)ctx->tw_x.p, (float2*)ctx->blk.p,    \
                           (double*)ctx->norm_part.p, ctx->dbg, np,            \
                           (size_t)ctx->g.ly * ctx->g.lx);                     \
    }
#define FN(T) { if (coef) FN2(T, true) else FN2(T, false) }
    DISPATCH_T(fg.Tx, FN)
#undef FN
#undef FN2
    hipLaunchKernelGGL(k_tile_norms, dim3(np * nb), dim3(64), 0, ctx->stream,
                       (const double*)ctx->norm_part.p, fg.Ty / 4, (double*)ctx->norms.p + ctx->norms_off);
    sc_prof_end(ctx);
    SC_HIP(ctx, hipGetLastError());
    return launch_fwd_cols(ctx, fg, 2 * np * nb, (float2*)ctx->uc.p + ctx->uc_off, (float2*)ctx->uc2.p + ctx->uc_off, 1);
}

// Symmetric fast path (k_split_templ_sym / k_inv_cols_sym): all templates of the
// chunk share one parity and the tile is small enough for the parked spectrum.
static bool fft_use_sym(const sc_ctx* ctx, const FftGeom& fg, int parity) {
    return parity != 0 && fg.Ty <= 2048 && ctx->variant != 8;
}

int fft_forward_templates(sc_ctx* ctx, const FftGeom& fg, int first, int n, int parity) {
    size_t lds = fft_lds_bytes(fg.Tx);
    dim3 grid(fg.Ty / 4, n);
    sc_prof_begin(ctx, SC_K_FWD_ROWS);
#define FN(T)                                                                  \
    {                                                                          \
        int rc = set_lds(ctx, k_fwd_rows_templ<T>, lds);                       \
        if (rc) return rc;                                                     \
        hipLaunchKernelGGL(k_fwd_rows_templ<T>, grid, dim3(fft_threads(T)),    \
                           lds, ctx->stream, (const TemplDev*)ctx->templ.p,    \
                           first, (const float*)ctx->win_w.p,                  \
                           (const uint8_t*)ctx->win_m.p,                       \
                           (const double*)ctx->sums.p, fg.Ty,                  \
                           (const float2*)ctx->tw_x.p, (float2*)ctx->blk.p);   \
    }
    DISPATCH_T(fg.Tx, FN)
#undef FN
    sc_prof_end(ctx);
    SC_HIP(ctx, hipGetLastError());
    if (fft_use_sym(ctx, fg, parity) && ctx->variant != 7) {
        // symmetric templates: column transform and split in one kernel (k_fwd_cols_tsym)
        size_t ldsc = fft_lds_bytes(fg.Ty);
        dim3 gridc(fg.Tx / 8 + 1, n);
        sc_prof_begin(ctx, SC_K_FWD_COLS);
#define FN(T)                                                                  \
    {                                                                          \
        int rc = set_lds(ctx, k_fwd_cols_tsym<T>, ldsc);                       \
        if (rc) return rc;                                                     \
        hipLaunchKernelGGL(k_fwd_cols_tsym<T>, gridc, dim3(fft_threads(T)), ldsc, ctx->stream, \
                           (const float2*)ctx->blk.p, fg.Tx, (const float2*)ctx->tw_y.p, \
                           (const TemplDev*)ctx->templ.p + first, (const float2*)ctx->tw_x.p + fg.Tx, \
                           parity, (float*)ctx->wh.p, (float*)ctx->mh.p);      \
    }
        switch (fg.Ty) {
            case 64: FN(64); break;
            case 128: FN(128); break;
            case 256: FN(256); break;
            case 512: FN(512); break;
            case 1024: FN(1024); break;
            default: FN(2048); break;
        }
#undef FN
        sc_prof_end(ctx);
        SC_HIP(ctx, hipGetLastError());
        return SC_OK;
    }
    int rc = launch_fwd_cols(ctx, fg, n, (float2*)ctx->vh.p, nullptr, 0,
                             (const TemplDev*)ctx->templ.p + first);
    if (rc) return rc;
    size_t cells = half_plane(fg.Ty, fg.Tx);
    dim3 grid_s((unsigned)((cells / 2 + 255) / 256), n);
    sc_prof_begin(ctx, SC_K_FWD_COLS);
    if (fft_use_sym(ctx, fg, parity))
        hipLaunchKernelGGL(k_split_templ_sym, grid_s, dim3(256), 0, ctx->stream,
                           (const float2*)ctx->vh.p, fg.Ty, fg.Tx, (const float2*)ctx->tw_y.p + fg.Ty,
                           (const float2*)ctx->tw_x.p + fg.Tx,
                           parity, (float*)ctx->wh.p, (float*)ctx->mh.p);
    else
        hipLaunchKernelGGL(k_split_templ, grid_s, dim3(256), 0, ctx->stream, (const float2*)ctx->vh.p,
                           fg.Ty, fg.Tx, (float2*)ctx->wh.p, (float2*)ctx->mh.p);
    sc_prof_end(ctx);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

// Templates [first, first+n) of the current batch have their spectra in vh
// planes [0, n).  For every tile pair: inverse transforms in groups of `group`
// templates, folded in template order.
int fft_inverse_fold(sc_ctx* ctx, const FftGeom& fg, int first, int n,
                     int group, bool to_maps, bool full_masks, int parity, int nb) {
    // nb > 1: templates [first, first + nb*n) are nb orientations of n templates each
    // (fft_batch_orientations); n <= group then, and the row kernel is the fast one
    const bool sym = fft_use_sym(ctx, fg, parity);
    int np = npairs_of(fg);
    size_t lds_r = fft_lds_bytes(fg.Tx);
    // row pairs of a tile that hold valid outputs: r' in [Py, Py + Vy)
    int rp_lo = 0, rp_hi = fg.Ty / 2 - 1;
    if (!fg.circ_y) {
        rp_lo = fg.Py / 2;
        rp_hi = std::min(fg.Ty - 1, fg.Py + fg.Vy - 1) / 2;
    }
    const int rp_n = rp_hi - rp_lo + 1;
    if (group > SC_MAX_GROUP)
        return sc_fail(ctx, SC_ERR_INVALID, "group %d exceeds %d", group, SC_MAX_GROUP);
    const int pb = std::max(1, ctx->fft_pb);
    const size_t yblock = (size_t)fg.Ty * fg.Tx * group;          // cells per pair in yw / ym
    const bool fast = (fg.Tx == 512 || fg.Tx == 1024 || fg.Tx == 2048) && ctx->variant != 9;
    // block and mirror workgroups of the two-launch form in one launch, paired per XCD (k_inv_cols_symx)
    const bool symx = sym && ctx->variant != 6 && fg.Ty >= 512 && fg.Ty <= 2048 && (fg.Tx / 8) % 8 == 0;
    // one wave per column (k_inv_cols_w8): column length 1024 / 2048.  Paired-template chunks at 2048 take
    // the four-wave form k_inv_cols_w4 (one wave per SIMD, 512 registers): with eight waves the second
    // coefficient plane does not fit - 32 spilled values reloaded per transform wait for the stores in
    // flight, 1 455 us at C2 against k_inv_cols_symx's 1 000; k_inv_cols_w4: 915
    const bool w8 = symx && ctx->variant != 2 && (fg.Ty == 2048 || fg.Ty == 1024) && (fg.Tx / 16) % 8 == 0;
    if (nb > 1 && (!fast || n > group || nb * n > SC_MAX_BATCH || n > SC_MAX_GROUP))
        return sc_fail(ctx, SC_ERR_INVALID, "orientation batching outside its conditions");
    // One chunk = pc tile pairs through I1 and I2, group by group.  PTV: the chunk is a
    // single pair whose second tile is empty; templates ride in pairs instead (see
    // k_inv_cols_sym) - the symmetric I1 and the fast I2 know that mode.
    auto chunk = [&](int pair0, int pc, auto ptc) -> int {
        constexpr bool PTV = decltype(ptc)::value;
        for (int g0 = 0; g0 < n; g0 += group) {
            int G = std::min(group, n - g0);
            // Paired orientations (inv_cols_sym_body, XP): a paired-template chunk with ONE template per
            // orientation and a batch of orientations would run every transform half empty
            // (option "variant" 12: off, for the cross-check of the two forms)
            const bool xp = PTV && n == 1 && nb >= 2 && pc == 1 && sym && symx && fg.Ty == 512 &&
                            ctx->variant != 12 && !to_maps;
            // I1: as many tile pairs per launch as it takes to fill the chip once
            // (one 2048-tile pair does; longer launches lost 6 % on the sustained C3 run)
            const size_t lds_c = (size_t)4 * fft_line(fg.Ty) * sizeof(float2) +
                                 (fg.Ty <= 2048 ? (size_t)4 * fg.Ty * sizeof(float2) : 0);
            const int slots = 256 * (int)std::max<size_t>(1, (size_t)(160 * 1024) / lds_c);
            // (nb > 1, batched orientations: every job of the chunk in one launch - job
            //  ob * pc + q, the numbering the row kernel expects)
            int pi1 = nb > 1 ? pc : std::max(1, std::min(pc, (slots + fg.Tx / 8 - 1) / (fg.Tx / 8)));
            // Wave-per-column kernel, one orientation: i1_pairs tile pairs per launch, INTERLEAVED along x (k_inv_cols_w8,
            // jil) - a launch of several pairs one after the other along y lost 6 % (the pairs' workgroups drift apart);
            // interleaved, the 2 x i1_pairs workgroups that stream the same coefficient lines run together on one XCD
            int jil = 1;
            if (nb == 1 && w8 && (!PTV || fg.Ty == 1024) && ctx->variant != 1 && !xp && ctx->i1_pairs > 1) {
                jil = std::min(pc, ctx->i1_pairs);
                pi1 = jil;
            }
            // rows masked by the templates' window limits are neither stored by the wave-per-column
            // kernels nor scored by the row pass (not with explicit per-cell masks or single-template
            // maps: those write every cell; option "variant" 13 switches it off for the cross-check)
            const bool row_skip = !full_masks && !to_maps && ctx->variant != 13;
            // An under-filled column pass deals its transforms out along grid.z (take_template_share): nz parts
            // so that the launch's workgroups come up to the chip's resident capacity for the kernel (the
            // four-column kernels: 256 CUs x what LDS and 256 registers allow; wave-per-column: one or two
            // 512-thread workgroups per CU), every part at least four transforms.  Option "split_i1" 0: off.
            auto parts_for = [&](long long workgroups, long long capacity, int transforms) {
                if (!ctx->split_i1 || workgroups <= 0) return 1;
                if (ctx->split_i1 > 1) return std::max(1, std::min(ctx->split_i1, transforms));   // (lab: a given number of parts)
                // the number of parts (1 .. 8, every part at least four transforms) that fills the launch's rounds of
                // resident workgroups best: 672 workgroups on 512 slots are 1.31 rounds - a third of the chip idles through
                // the second -, in three parts 2 016 workgroups are 3.94 rounds of a third the length (C1F: seven batched
                // orientations x three tile pairs x 32 column blocks).  Fewer parts win ties: every part parks the spectrum.
                int best = 1;
                double best_u = 0.0;
                for (int nz = 1; nz <= 8 && transforms / nz >= 4; ++nz) {
                    const long long w = workgroups * nz, rounds = (w + capacity - 1) / capacity;
                    const double u = (double)w / (double)(rounds * capacity);
                    if (u > best_u + 0.03) { best_u = u; best = nz; }
                }
                return best;
            };
            const int NGl = PTV ? (G + 1) / 2 : G;
            sc_prof_begin(ctx, SC_K_INV_COLS);
            int n_i1 = 0;
            for (int pl0 = 0; pl0 < pc; pl0 += pi1) {
            const int pcc = std::min(pi1, pc - pl0);
            const int jilc = jil > 1 ? pcc : 1;                 // (the last launch of a chunk may hold fewer pairs)
            const int pair = pair0 + pl0;
            float2* ywp = (float2*)ctx->yw.p + (size_t)pl0 * yblock;
            float2* ymp = (float2*)ctx->ym.p + (size_t)pl0 * yblock;
            n_i1 += symx ? 1 : 2;            // one paired launch, or own columns + mirrors
#define COL_ARGS(CB0)                                                          \
    ctx->stream, (const float2*)ctx->uc.p + ctx->uc_off, (const float2*)ctx->uc2.p + ctx->uc_off, (const float2*)ctx->wh.p, \
        (const float2*)ctx->mh.p, fg.Tx, CB0, pair, g0, G, rp_lo, rp_hi, ctx->dbg,        \
        (const float2*)ctx->tw_y.p, ywp, ymp, group, np, pcc, n, (const TileDev*)ctx->tiles.p, fg.circ_y ? -1 : fg.Py
#define SYM_ARGS(CB0)                                                          \
    ctx->stream, (const float2*)ctx->uc.p + ctx->uc_off, (const float2*)ctx->uc2.p + ctx->uc_off, (const float*)ctx->wh.p, \
        (const float*)ctx->mh.p, fg.Tx, CB0, pair, g0, G, rp_lo, rp_hi,                    \
        (const float2*)ctx->tw_x.p + fg.Tx, parity, (const float2*)ctx->tw_y.p, ywp, ymp, group
#define SYM_ARGS_D(CB0) SYM_ARGS(CB0), ctx->dbg, np, pcc, n, (const TileDev*)ctx->tiles.p, fg.circ_y ? -1 : fg.Py
#define FN_SYMX(T)                                                             \
    {                                                                          \
        int rc = set_lds(ctx, k_inv_cols_symx<T, PTV>, inv_cols_lds<T>());     \
        if (rc) return rc;                                                     \
        const int nz_ = parts_for((long long)(fg.Tx / 4) * nb * pcc,                                \
                                  256LL * std::max<size_t>(1, std::min<size_t>((160 * 1024) / inv_cols_lds<T>(), \
                                                                               2048 / fft_threads(T) / 2)), NGl); \
        hipLaunchKernelGGL((k_inv_cols_symx<T, PTV>), dim3(fg.Tx / 4, nb * pcc, nz_), dim3(fft_threads(T)), \
                           inv_cols_lds<T>(), ctx->stream, (const float2*)ctx->uc.p + ctx->uc_off, (const float2*)ctx->uc2.p + ctx->uc_off, \
                           (const float*)ctx->wh.p, (const float*)ctx->mh.p, fg.Tx, pair, g0, G, rp_lo, rp_hi, \
                           (const float2*)ctx->tw_x.p + fg.Tx, parity, (const float2*)ctx->tw_y.p, ywp, ymp, group, \
                           ctx->dbg, np, pcc, n, (const TileDev*)ctx->tiles.p, fg.circ_y ? -1 : fg.Py); \
    }
#define FN_W8(T)                                                               \
    {                                                                          \
        int rc = set_lds(ctx, k_inv_cols_w8<T, PTV>, w8_lds<T>());             \
        if (rc) return rc;                                                     \
        const int nz_ = parts_for((long long)(fg.Tx / 8) * nb * pcc, 256LL * ((T == 1024 && !PTV) ? 2 : 1), NGl); \
        hipLaunchKernelGGL((k_inv_cols_w8<T, PTV>), dim3(fg.Tx / 8 * jilc, nb * pcc / jilc, nz_), dim3(512),  \
                           w8_lds<T>(), ctx->stream, (const float2*)ctx->uc.p + ctx->uc_off, (const float2*)ctx->uc2.p + ctx->uc_off, \
                           (const float*)ctx->wh.p, (const float*)ctx->mh.p, fg.Tx, pair, g0, G, rp_lo, rp_hi, \
                           (const float2*)ctx->tw_x.p + fg.Tx, parity, (const float2*)ctx->tw_y.p, ywp, ymp, group, \
                           np, pcc, n, (const TileDev*)ctx->tiles.p, fg.circ_y ? -1 : fg.Py, \
                           row_skip ? (const TemplDev*)ctx->templ.p + first : nullptr, jilc); \
    }
#define FN_W4(T)                                                               \
    {                                                                          \
        /* (more LDS than it uses: one workgroup per CU, one wave per SIMD) */ \
        const size_t lds4 = (size_t)88 * 1024;                                 \
        int rc = set_lds(ctx, k_inv_cols_w4<T, PTV>, lds4);                    \
        if (rc) return rc;                                                     \
        hipLaunchKernelGGL((k_inv_cols_w4<T, PTV>), dim3(fg.Tx / 4, nb * pcc), dim3(256),  \
                           lds4, ctx->stream, (const float2*)ctx->uc.p + ctx->uc_off, (const float2*)ctx->uc2.p + ctx->uc_off, \
                           (const float*)ctx->wh.p, (const float*)ctx->mh.p, fg.Tx, pair, g0, G, rp_lo, rp_hi, \
                           (const float2*)ctx->tw_x.p + fg.Tx, parity, (const float2*)ctx->tw_y.p, ywp, ymp, group, \
                           np, pcc, n, (const TileDev*)ctx->tiles.p, fg.circ_y ? -1 : fg.Py, \
                           row_skip ? (const TemplDev*)ctx->templ.p + first : nullptr); \
    }
#define FN_SYM(T)                                                              \
    {                                                                          \
        int rc = set_lds(ctx, k_inv_cols_sym<T, false, PTV>, inv_cols_lds<T>());    \
        if (rc) return rc;                                                     \
        rc = set_lds(ctx, k_inv_cols_sym<T, true, PTV>, inv_cols_lds<T>());         \
        if (rc) return rc;                                                     \
        const int nlo = fg.Tx / 8, nhi = fg.Tx / 4 - nlo;                      \
        hipLaunchKernelGGL((k_inv_cols_sym<T, false, PTV>), dim3(nlo, nb * pcc), dim3(fft_threads(T)), \
                           inv_cols_lds<T>(), SYM_ARGS_D(0));                  \
        if (nhi > 0)                                                           \
            hipLaunchKernelGGL((k_inv_cols_sym<T, true, PTV>), dim3(nhi, nb * pcc), dim3(fft_threads(T)), \
                               inv_cols_lds<T>(), SYM_ARGS_D(nlo));            \
    }
#define FN(T)                                                                  \
    {                                                                          \
        int rc = set_lds(ctx, k_inv_cols<T, false>, inv_cols_lds<T>());        \
        if (rc) return rc;                                                     \
        rc = set_lds(ctx, k_inv_cols<T, true>, inv_cols_lds<T>());             \
        if (rc) return rc;                                                     \
        const int nlo = fg.Tx / 8, nhi = fg.Tx / 4 - nlo;                      \
        hipLaunchKernelGGL((k_inv_cols<T, false>), dim3(nlo, nb * pcc), dim3(fft_threads(T)), \
                           inv_cols_lds<T>(), COL_ARGS(0));                    \
        if (nhi > 0)                                                           \
            hipLaunchKernelGGL((k_inv_cols<T, true>), dim3(nhi, nb * pcc), dim3(fft_threads(T)), \
                               inv_cols_lds<T>(), COL_ARGS(nlo));              \
    }
            // column length 512: half a wave per column (k_inv_cols_h2), where the grid pairs up per XCD - sixteen
            // workgroup ids = eight column blocks and their mirrors: (Tx / 32) % 8 == 0; "variant" 18: the four-column kernels
            const bool h2 = sym && symx && fg.Ty == 512 && (fg.Tx / 32) % 8 == 0 && ctx->variant != 18 && ctx->variant != 1 &&
                            ctx->variant != 2;
#define FN_H2(XPV, GY, YSTR, PCJ, TSTR, TLP)                                   \
    {                                                                          \
        int rc = set_lds(ctx, k_inv_cols_h2<PTV, XPV>, h2_lds());              \
        if (rc) return rc;                                                     \
        const int nz_ = XPV ? 1 : parts_for((long long)(fg.Tx / 16) * (GY), 512, NGl); \
        hipLaunchKernelGGL((k_inv_cols_h2<PTV, XPV>), dim3(fg.Tx / 16, (GY), nz_), dim3(512), h2_lds(), ctx->stream, \
                           (const float2*)ctx->uc.p + ctx->uc_off, (const float2*)ctx->uc2.p + ctx->uc_off, \
                           (const float*)ctx->wh.p, (const float*)ctx->mh.p, fg.Tx, pair, g0, G, rp_lo, rp_hi, \
                           (const float2*)ctx->tw_x.p + fg.Tx, parity, (const float2*)ctx->tw_y.p, ywp, ymp, (YSTR), \
                           np, (PCJ), (TSTR), (const TileDev*)ctx->tiles.p, fg.circ_y ? -1 : fg.Py, (TLP)); \
    }
            // (paired orientations - ONE transform per plane and job, all prologue - stay on the four-column kernel: on the
            //  sixteen-column workgroups C5's column pass took 2.07 ms against 1.62; option "variant" 19 takes them anyway)
            if (xp && h2 && ctx->variant == 19) {
                if constexpr (PTV) FN_H2(true, (nb + 1) / 2, 1, 1, nb, (const TemplDev*)nullptr)
            } else if (h2 && !xp) {
                FN_H2(false, nb * pcc, group, pcc, n, row_skip ? (const TemplDev*)ctx->templ.p + first : (const TemplDev*)nullptr)
            } else if (xp) {
                // paired orientations: job j = orientations 2j, 2j+1; plane j of Y; tstride carries nb
                const size_t ldsx = inv_cols_lds<512>() + (size_t)4 * 512 * sizeof(float2);
                int rc = set_lds(ctx, k_inv_cols_symx<512, PTV, PTV>, ldsx);
                if (rc) return rc;
                hipLaunchKernelGGL((k_inv_cols_symx<512, PTV, PTV>), dim3(fg.Tx / 4, (nb + 1) / 2), dim3(fft_threads(512)),
                                   ldsx, ctx->stream, (const float2*)ctx->uc.p + ctx->uc_off, (const float2*)ctx->uc2.p + ctx->uc_off,
                                   (const float*)ctx->wh.p, (const float*)ctx->mh.p, fg.Tx, pair, g0, G, rp_lo, rp_hi,
                                   (const float2*)ctx->tw_x.p + fg.Tx, parity, (const float2*)ctx->tw_y.p, ywp, ymp, 1,
                                   ctx->dbg, np, 1, nb, (const TileDev*)ctx->tiles.p, fg.circ_y ? -1 : fg.Py);
            } else if (w8 && (!PTV || (fg.Ty == 1024 && ctx->variant != 1))) {
                if (fg.Ty == 2048) FN_W8(2048) else FN_W8(1024)
            } else if (w8 && PTV && fg.Ty == 2048 && ctx->variant != 1) {
                FN_W4(2048)
            } else if (sym && symx) {
                switch (fg.Ty) {
                    case 512: FN_SYMX(512); break;
                    case 1024: FN_SYMX(1024); break;
                    default: FN_SYMX(2048); break;
                }
            } else if (sym) {
                switch (fg.Ty) {
                    case 64: FN_SYM(64); break;
                    case 128: FN_SYM(128); break;
                    case 256: FN_SYM(256); break;
                    case 512: FN_SYM(512); break;
                    case 1024: FN_SYM(1024); break;
                    default: FN_SYM(2048); break;
                }
            } else {
                DISPATCH_T(fg.Ty, FN)
            }
#undef FN
#undef FN_SYM
#undef FN_SYMX
#undef FN_W8
#undef FN_W4
#undef FN_H2
#undef SYM_ARGS_D
#undef SYM_ARGS
#undef COL_ARGS
            }
            sc_prof_end(ctx, n_i1);
            const int pair = pair0;
            // The row pass folds at most SC_MAX_GROUP templates per launch (its scalar table, its 64-bit mask of
            // transforms, the winner's byte): a batch of more - nb orientations of G templates each, round 5 -
            // goes through it in slices of whole orientations, in order; the running best lives in the record
            // between launches, so the fold is the one of a single launch (and of no batching at all).
            // (slices of equal size: 8 orientations of 10 templates go 4 + 4, not 6 + 2 - a short last launch leaves the
            //  chip part empty)
            // Round 5, second step: where the dealt-out row pass applies (small grids: SPLITK) a launch carries up to
            // 255 templates - the winner's byte - in shares of at most SC_MAX_GROUP transforms each; C1F's 7 batched
            // orientations of 35 ages are then ONE row-pass launch of four shares at four waves per SIMD instead of
            // seven launches of two
            // (with near-tie flags on - the exact mode - only for searches of several templates per orientation: a share meets the
            //  record as the launch found it, not what the shares before it have reached, and lists near-ties against that floor
            //  that the sequential fold would not; a Ricker's SNR varies slowly with the orientation - on C5, one template per
            //  orientation, the split cost more in float64 pairs than it saved in the pass: 14.0 -> 17.3 ms; C1F 51.7 -> 46.6)
            const bool near_split_ok = !(ctx->near_w > 0.f) || (!xp && G >= 4);
            const bool can_split = fast && near_split_ok && !to_maps && !full_masks && fg.Tx <= 1024 && ctx->variant != 15 &&
                                   !(ctx->sib & 1) && ctx->variant != 20;
            // How many shares: up to four (any number, not only powers of two) while the launch stays within ~4 300 waves -
            // the four per SIMD the kernel's 128 registers allow and 5 % (measured at C1F, 1 044 single-wave rows: two
            // shares 19.0 ms, three 13.8, four 13.5; option "split_fill" sets another bound)
            const int wpw = inv_rows_fast_threads<512>() * (fg.Tx / 512) / 64;          // waves per row workgroup
            const long long row_wgs = (long long)rp_n * 2 * pc;
            const long long cap_wg = (ctx->split_fill > 0 ? ctx->split_fill : 4300) / wpw;
            const int nsplit_max = can_split ? (int)std::max<long long>(1, std::min<long long>(4, cap_wg / std::max<long long>(1, row_wgs))) : 1;
            const int tper = PTV ? 2 : 1;                      // templates per transform
            int nbs = nb;
            if (nb > 1 && !xp) {
                const int cap_t = nsplit_max > 1 ? std::min(255, SC_MAX_GROUP * nsplit_max * tper) : SC_MAX_GROUP;
                const int cap = std::max(1, cap_t / std::max(1, G)), nsl = (nb + cap - 1) / cap;
                nbs = (nb + nsl - 1) / nsl;
            }
            for (int b0 = 0; b0 < nb; b0 += nbs) {
            const int nbc = std::min(nbs, nb - b0);
            const size_t yoff = (size_t)b0 * pc * group * (size_t)fg.Ty * fg.Tx;        // job (b0, 0) of yw / ym
            const float2* yw_s = (const float2*)ctx->yw.p + yoff;
            const float2* ym_s = (const float2*)ctx->ym.p + yoff;
            const double* norms_s = (const double*)ctx->norms.p + ctx->norms_off + (size_t)2 * np * b0;
            RowArgs ra{fg.Ty, fg.Py, fg.Qx, fg.circ_y, fg.circ_x, ctx->g.cy0, ctx->g.cx0,
                       ctx->g.cx1 - ctx->g.cx0, pair, first + g0 + b0 * n, G, rp_lo, rp_n, ctx->dbg, group,
                       nbc, np, pc, SibSync{nullptr, 0}, (unsigned long long*)ctx->res_stats.p, 0, row_skip ? 1 : 0,
                       1, 0, nullptr, nullptr, nullptr, 0.f, nullptr};
            if (xp) {                            // to the row pass: ONE orientation of nb templates, in pairs
                ra.G = nb; ra.nb = 1; ra.ystride = 1; ra.xp = 1;
            }
            dim3 gridr(fast ? ((rp_n + 7) / 8) * 16 : (rp_n + 1) / 2, pc);
            // Small grids: a row workgroup folds the launch's transforms one after the other, and a 512 x 512
            // search has 512 rows of ONE wave each for 1 024 SIMDs (BASELINE config C5: the row pass was a
            // third of the search, a chain of single-wave transforms at half the issue rate).  There the
            // transforms are dealt out over nsplit workgroups per row, each folding its share in order into a
            // record of its own; k_merge_split folds the shares into the record in order.  Same winners, same
            // ties (option "variant" 15: off).
            // Near-tie flags (option "near_window" > 0: the host layer's exact mode): the plain fast kernel only
            const bool near = ctx->near_w > 0.f && !to_maps;
            if (near) {
                if (!fast || full_masks)
                    return sc_fail(ctx, SC_ERR_UNSUPPORTED, "near-tie flags need the fast row kernel without per-cell masks");
                int rc = sc_near_buffers(ctx, &ra.ev_count, &ra.ev, &ra.ev_cap);
                if (rc) return rc;
                ra.near_w = ctx->near_w;
                ra.near = (uint8_t*)ctx->near.p;
            }
            int nsplit = 1;
            if (fast && near_split_ok && !to_maps && !full_masks && fg.Tx <= 1024 && ctx->variant != 15 && !(ctx->sib & 1)) {
                const int ngl = nbc * (PTV ? (G + 1) / 2 : G);
                // (option "split_fill": the waves the dealt-out row pass may come to instead of the chip's resident
                //  capacity for the kernel; every share at least four transforms, at most SC_MAX_GROUP - its 64-bit mask)
                nsplit = std::max(1, std::min(nsplit_max, ngl / 4));
                nsplit = std::max(nsplit, (ngl + SC_MAX_GROUP - 1) / SC_MAX_GROUP);
                if (nsplit > 4 || (ngl + nsplit - 1) / nsplit > SC_MAX_GROUP)
                    return sc_fail(ctx, SC_ERR_INVALID, "row pass: %d transforms in %d shares", ngl, nsplit);
            }
            if (nsplit > 1) {
                const size_t nc = (size_t)(ctx->g.cy1 - ctx->g.cy0) * (ctx->g.cx1 - ctx->g.cx0);
                const size_t need = (size_t)4 * nc * sizeof(float);          // up to four shares: three scratch records (four with near-tie flags on)
                const bool fresh = ctx->split_s.cap < need;
                int rc;
                if ((rc = sc_ensure(ctx, ctx->split_s, need))) return rc;
                if ((rc = sc_ensure(ctx, ctx->split_a, need))) return rc;
                if ((rc = sc_ensure(ctx, ctx->split_i, need))) return rc;
                if (fresh) SC_HIP(ctx, hipMemsetAsync(ctx->split_s.p, 0, need, ctx->stream));   // (sc_reset_best clears it from then on)
                ra.nsplit = nsplit;
                ra.nc = nc;
                ra.s2 = (float*)ctx->split_s.p;
                ra.a2 = (float*)ctx->split_a.p;
                ra.i2 = (uint32_t*)ctx->split_i.p;
                gridr.z = nsplit;
            }
            if (fast && (ctx->sib & 1)) {
                int rc = sib_slots(ctx, (size_t)gridr.x * gridr.y, ra.sib);
                if (rc) return rc;
            }
            sc_prof_begin(ctx, SC_K_INV_ROWS);
#define ROW_ARGS lds_r, FAST_ARGS
#define FAST_ARGS                                                              \
    ctx->stream, yw_s, ym_s, ra, ctx->g,                                       \
        (const TileDev*)ctx->tiles.p, (const TemplDev*)ctx->templ.p, (const double*)ctx->sums.p, \
        (const double*)ctx->wl1.p, norms_s, ctx->kappa,                        \
        (const double*)ctx->xaxis.p, (const double*)ctx->yaxis.p, (const float2*)ctx->tw_x.p, \
        (float*)ctx->best_snr.p, (float*)ctx->best_amp.p, (uint32_t*)ctx->best_id.p,        \
        to_maps ? (float*)ctx->map_amp.p : nullptr, to_maps ? (float*)ctx->map_snr.p : nullptr
#define LAUNCH_ROWS(T, FULLV)                                                  \
    {                                                                          \
        int rc = set_lds(ctx, k_inv_rows<T, FULLV>, lds_r);                    \
        if (rc) return rc;                                                     \
        hipLaunchKernelGGL((k_inv_rows<T, FULLV>), gridr, dim3(fft_threads(T)), ROW_ARGS); \
    }
#define LAUNCH_FAST2(T, FULLV, MAPSV)                                          \
    {                                                                          \
        int rc = set_lds(ctx, k_inv_rows_fast<T, FULLV, MAPSV, PTV>, inv_rows_fast_lds<T>()); \
        if (rc) return rc;                                                     \
        hipLaunchKernelGGL((k_inv_rows_fast<T, FULLV, MAPSV, PTV>), gridr,          \
                           dim3(inv_rows_fast_threads<T>()), inv_rows_fast_lds<T>(), FAST_ARGS); \
    }
#define LAUNCH_FAST(T, FULLV)                                                  \
    { if (to_maps) LAUNCH_FAST2(T, FULLV, true) else LAUNCH_FAST2(T, FULLV, false) }
#define LAUNCH_SPLIT2(T, NEARV)                                                \
    {                                                                          \
        int rc = set_lds(ctx, k_inv_rows_fast<T, false, false, PTV, true, NEARV>, inv_rows_fast_lds_split<T>()); \
        if (rc) return rc;                                                     \
        hipLaunchKernelGGL((k_inv_rows_fast<T, false, false, PTV, true, NEARV>), gridr,   \
                           dim3(inv_rows_fast_threads<T>()), inv_rows_fast_lds_split<T>(), FAST_ARGS); \
        const unsigned mb_ = (unsigned)std::min<size_t>((ra.nc + 255) / 256, 2048); \
        hipLaunchKernelGGL(k_merge_split<NEARV>, dim3(mb_), dim3(256), 0, ctx->stream, (float*)ctx->best_snr.p, \
                           (float*)ctx->best_amp.p, (uint32_t*)ctx->best_id.p, ra.s2,                \
                           (const float*)ra.a2, (const uint32_t*)ra.i2, ra.nc, (NEARV) ? nsplit : nsplit - 1, ra.near_w, ra.near, \
                           ra.ev_count, ra.ev, ra.ev_cap);                                           \
    }
#define LAUNCH_SPLIT(T) { if (near) LAUNCH_SPLIT2(T, true) else LAUNCH_SPLIT2(T, false) }
#define LAUNCH_NEAR(T)                                                         \
    {                                                                          \
        int rc = set_lds(ctx, k_inv_rows_fast<T, false, false, PTV, false, true>, inv_rows_fast_lds<T>()); \
        if (rc) return rc;                                                     \
        hipLaunchKernelGGL((k_inv_rows_fast<T, false, false, PTV, false, true>), gridr,   \
                           dim3(inv_rows_fast_threads<T>()), inv_rows_fast_lds<T>(), FAST_ARGS); \
    }
            if (nsplit > 1) {
                if (fg.Tx == 512) LAUNCH_SPLIT(512) else LAUNCH_SPLIT(1024)
            } else if (near) {
                switch (fg.Tx) {
                    case 512: LAUNCH_NEAR(512) break;
                    case 1024: LAUNCH_NEAR(1024) break;
                    default: LAUNCH_NEAR(2048) break;
                }
            } else if (fast) {
                switch (fg.Tx) {
                    case 512: if (full_masks) LAUNCH_FAST(512, true) else LAUNCH_FAST(512, false) break;
                    case 1024: if (full_masks) LAUNCH_FAST(1024, true) else LAUNCH_FAST(1024, false) break;
                    default: if (full_masks) LAUNCH_FAST(2048, true) else LAUNCH_FAST(2048, false) break;
                }
            } else {
#define FN(T) { if (full_masks) LAUNCH_ROWS(T, true) else LAUNCH_ROWS(T, false) }
                DISPATCH_T(fg.Tx, FN)
#undef FN
            }
#undef LAUNCH_ROWS
#undef LAUNCH_FAST
#undef LAUNCH_FAST2
#undef LAUNCH_SPLIT
#undef LAUNCH_SPLIT2
#undef LAUNCH_NEAR
#undef ROW_ARGS
#undef FAST_ARGS
            sc_prof_end(ctx);
            }                                    // (row-pass slices)
        }
        return SC_OK;
    };
    const bool pt = sym && fast && (fg.ntiles & 1) && ctx->variant != 5;
    const int np_main = pt ? np - 1 : np;
    for (int pair0 = 0; pair0 < np_main; pair0 += pb) {
        int rc = chunk(pair0, std::min(pb, np_main - pair0), std::false_type{});
        if (rc) return rc;
    }
    if (pt) {
        int rc = chunk(np - 1, 1, std::true_type{});
        if (rc) return rc;
    }
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}
