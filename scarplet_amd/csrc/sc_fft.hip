// FFT path: overlap-save tiles, row / column FFTs done entirely in LDS.
//
// Data flow for one orientation (all templates of the orientation share it):
//
//   F1c  k_fwd_rows_curv   curvature of TWO tiles packed as re/im  -> row FFT
//   F2   k_fwd_cols        column FFT -> uc, uc2 (spectra of curv, curv^2)
//   F1t  k_fwd_rows_templ  template tile v = W + iM                -> row FFT
//   F2   k_fwd_cols        column FFT -> vh
//   I1   k_inv_cols        FFT(W) = (vh[f]+conj vh[-f])/2, FFT(M) likewise;
//                          P1 = FFT(W)*uc, P2 = FFT(M)*uc2; inverse column FFT
//   I2   k_inv_rows        inverse row FFT -> xcorr(A) + i xcorr(B), T3 likewise;
//                          float64 amp/SNR epilogue, masks, running-best fold
//
// Two real tiles ride in one complex transform (the template is real), so no
// real-to-complex bookkeeping is needed anywhere.
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
// The in-LDS FFT is a radix-4 Stockham autosort (natural order in and out):
// every stage reads element t + k*T/4 (conflict-free) and writes
// q + s*(4p + k); a final radix-2 stage handles odd log2(T).
#include "sc_internal.h"
#include <math.h>

struct TileDev {
    int i0, j0, vy, vx, gi0, gj0;
};

__host__ __device__ constexpr int fft_threads(int T) {
    return T >= 512 ? 512 : (T < 64 ? 64 : T);
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) {
    return make_float2(a.x + b.x, a.y + b.y);
}
__device__ __forceinline__ float2 csub(float2 a, float2 b) {
    return make_float2(a.x - b.x, a.y - b.y);
}

// Transform 4 independent lines of length T held in LDS (line l at s + l*T).
// tw[k] = exp(-2 pi i k / T).  All NT threads of the workgroup take part.
template <int T, bool INV>
__device__ __forceinline__ void fft4_lines(float2* s, const float2* __restrict__ tw) {
    constexpr int NT = fft_threads(T);
    constexpr int Q = T / 4;
    constexpr int U = (T + NT - 1) / NT;       // butterflies per thread per stage
    const int tid = threadIdx.x;
    int lst = 0;
#pragma unroll 1
    for (int n = T; n >= 4; n >>= 2, lst += 2) {
        float2 a[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int b = tid + u * NT;
            if (b < T) {
                int line = b / Q, t = b - line * Q;
                const float2* base = s + line * T;
#pragma unroll
                for (int k = 0; k < 4; ++k) a[u][k] = base[t + k * Q];
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int b = tid + u * NT;
            if (b < T) {
                int line = b / Q, t = b - line * Q;
                float2* base = s + line * T;
                int p = t >> lst, q = t & ((1 << lst) - 1);
                float2 w1 = tw[p << lst], w2 = tw[(2 * p) << lst], w3 = tw[(3 * p) << lst];
                if (INV) { w1.y = -w1.y; w2.y = -w2.y; w3.y = -w3.y; }
                float2 apc = cadd(a[u][0], a[u][2]), amc = csub(a[u][0], a[u][2]);
                float2 bpd = cadd(a[u][1], a[u][3]), bmd = csub(a[u][1], a[u][3]);
                float2 jbmd = make_float2(-bmd.y, bmd.x);      // j*(b-d)
                float2 x1 = INV ? cadd(amc, jbmd) : csub(amc, jbmd);
                float2 x3 = INV ? csub(amc, jbmd) : cadd(amc, jbmd);
                int o = q + ((4 * p) << lst);
                base[o] = cadd(apc, bpd);
                base[o + (1 << lst)] = cmul(w1, x1);
                base[o + (2 << lst)] = cmul(w2, csub(apc, bpd));
                base[o + (3 << lst)] = cmul(w3, x3);
            }
        }
        __syncthreads();
    }
    // odd power of two: one radix-2 stage, in place per thread
    constexpr bool ODD = (__builtin_ctz(T) & 1) != 0;
    if (ODD) {
        constexpr int H = T / 2;
        for (int b = tid; b < 4 * H; b += NT) {
            int line = b / H, t = b - line * H;
            float2* base = s + line * T;
            float2 x = base[t], y = base[t + H];
            base[t] = cadd(x, y);
            base[t + H] = csub(x, y);
        }
        __syncthreads();
    }
}

// ---- F1c: curvature of a tile pair -> row FFT -> blocked ---------------------
// grid = (Ty/4, npairs); out plane index = pair*2 + {0: curv, 1: curv^2}
template <int TX>
__global__ void __launch_bounds__(fft_threads(TX))
k_fwd_rows_curv(const float* __restrict__ curv, Geom g,
                const TileDev* __restrict__ tiles, int Ty,
                const float2* __restrict__ tw, float2* __restrict__ blk) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    constexpr int NT = fft_threads(TX);
    const int rb = blockIdx.x, pair = blockIdx.y;
    const TileDev ta = tiles[2 * pair], tb = tiles[2 * pair + 1];
    const size_t plane = (size_t)Ty * TX;
    // curvature values of this thread's cells stay in registers for plane 1
    constexpr int E = 4 * TX / NT;
    float va[E], vb[E];
#pragma unroll
    for (int u = 0; u < E; ++u) {
        int e = threadIdx.x + u * NT;
        int rr = e / TX, s = e - rr * TX, r = 4 * rb + rr;
        va[u] = ta.vy > 0 ? load_curv(curv, g, ta.gi0 + r, ta.gj0 + s) : 0.f;
        vb[u] = tb.vy > 0 ? load_curv(curv, g, tb.gi0 + r, tb.gj0 + s) : 0.f;
    }
    for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
        for (int u = 0; u < E; ++u) {
            int e = threadIdx.x + u * NT;
            sm[e] = pl ? make_float2(va[u] * va[u], vb[u] * vb[u])
                       : make_float2(va[u], vb[u]);
        }
        __syncthreads();
        fft4_lines<TX, false>(sm, tw);
        float2* out = blk + (size_t)(pair * 2 + pl) * plane + (size_t)rb * 4 * TX;
        for (int e = threadIdx.x; e < 4 * TX; e += NT) {
            int cb = e >> 4, rr = (e >> 2) & 3, cc = e & 3;
            out[e] = sm[rr * TX + 4 * cb + cc];
        }
        __syncthreads();
    }
}

// ---- F1t: template tile v = W + iM -> row FFT -> blocked ---------------------
// grid = (Ty/4, n_templates)
template <int TX>
__global__ void __launch_bounds__(fft_threads(TX))
k_fwd_rows_templ(const TemplDev* __restrict__ templ, int first,
                 const float* __restrict__ win_w,
                 const uint8_t* __restrict__ win_m,
                 const double* __restrict__ sums, int Ty,
                 const float2* __restrict__ tw, float2* __restrict__ blk) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
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
    if (!any) {                       // block-uniform: rows outside the support
        for (int e = threadIdx.x; e < 4 * TX; e += NT) out[e] = make_float2(0.f, 0.f);
        return;
    }
    for (int e = threadIdx.x; e < 4 * TX; e += NT) {
        int rr = e / TX, s = e - rr * TX;
        int q = (s <= t.qmax) ? s : s - TX;
        float2 v = make_float2(0.f, 0.f);
        if (prow[rr] != INT_MIN && q >= t.qmin && q <= t.qmax) {
            size_t o = (size_t)t.win_off + (size_t)(prow[rr] - t.pmin) * t.ww + (q - t.qmin);
            v = make_float2(alpha * win_w[o], win_m[o] ? 1.f : 0.f);
        }
        sm[e] = v;
    }
    __syncthreads();
    fft4_lines<TX, false>(sm, tw);
    for (int e = threadIdx.x; e < 4 * TX; e += NT) {
        int cb = e >> 4, rr = (e >> 2) & 3, cc = e & 3;
        out[e] = sm[rr * TX + 4 * cb + cc];
    }
}

// ---- F2: blocked -> column FFT -> spectrum [fx][fy] --------------------------
// grid = (Tx/4, nplanes).  split2: input plane q goes to (q&1 ? out1 : out0)
// at plane index q>>1 (curvature: curv / curv^2), else out0 plane q.
template <int TY>
__global__ void __launch_bounds__(fft_threads(TY))
k_fwd_cols(const float2* __restrict__ blk, int Tx, const float2* __restrict__ tw,
           float2* __restrict__ out0, float2* __restrict__ out1, int split2) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    constexpr int NT = fft_threads(TY);
    const int cb = blockIdx.x, q = blockIdx.y;
    const size_t plane = (size_t)TY * Tx;
    const float2* in = blk + (size_t)q * plane;
    const int nbx = Tx >> 2;
    for (int e = threadIdx.x; e < 4 * TY; e += NT) {
        int rbk = e >> 4, rr = (e >> 2) & 3, cc = e & 3;
        sm[cc * TY + 4 * rbk + rr] = in[((size_t)rbk * nbx + cb) * 16 + (e & 15)];
    }
    __syncthreads();
    fft4_lines<TY, false>(sm, tw);
    float2* out = split2 ? ((q & 1) ? out1 : out0) + (size_t)(q >> 1) * plane
                         : out0 + (size_t)q * plane;
    out += (size_t)cb * 4 * TY;                 // columns 4cb..4cb+3 are contiguous
    for (int e = threadIdx.x; e < 4 * TY; e += NT) out[e] = sm[e];
}

// ---- I1: spectra product -> inverse column FFT -> blocked --------------------
// grid = (Tx/4, G): template first+g against tile pair `pair`.
template <int TY>
__global__ void __launch_bounds__(fft_threads(TY))
k_inv_cols(const float2* __restrict__ uc, const float2* __restrict__ uc2,
           const float2* __restrict__ vh, int Tx, int pair, int vfirst,
           const float2* __restrict__ tw, float2* __restrict__ yw,
           float2* __restrict__ ym) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    constexpr int NT = fft_threads(TY);
    constexpr int E = 4 * TY / NT;
    const int cb = blockIdx.x, gidx = blockIdx.y;
    const size_t plane = (size_t)TY * Tx;
    const float2* u1 = uc + (size_t)pair * plane + (size_t)cb * 4 * TY;
    const float2* u2 = uc2 + (size_t)pair * plane + (size_t)cb * 4 * TY;
    const float2* v = vh + (size_t)(vfirst + gidx) * plane;
    const int nbx = Tx >> 2;
    for (int pl = 0; pl < 2; ++pl) {
        const float2* uu = pl ? u2 : u1;
#pragma unroll
        for (int u = 0; u < E; ++u) {
            int e = threadIdx.x + u * NT;
            int cc = e / TY, fy = e - cc * TY;
            int fx = 4 * cb + cc;
            float2 a = v[(size_t)fx * TY + fy];
            float2 b = v[(size_t)((Tx - fx) & (Tx - 1)) * TY + ((TY - fy) & (TY - 1))];
            // FFT(W) = (a + conj b)/2 ; FFT(M) = (a - conj b)/(2i)
            float2 h = pl ? make_float2(0.5f * (a.y + b.y), -0.5f * (a.x - b.x))
                          : make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
            sm[e] = cmul(h, uu[e]);
        }
        __syncthreads();
        fft4_lines<TY, true>(sm, tw);
        float2* o = (pl ? ym : yw) + (size_t)gidx * plane;
        for (int e = threadIdx.x; e < 4 * TY; e += NT) {
            int rbk = e >> 4, rr = (e >> 2) & 3, cc = e & 3;
            o[((size_t)rbk * nbx + cb) * 16 + (e & 15)] = sm[cc * TY + 4 * rbk + rr];
        }
        __syncthreads();
    }
}

// ---- I2: inverse row FFT -> epilogue -> fold ---------------------------------
// grid = (Ty/4); loops over the G templates of the launch (fold order).
// FULL = false is the lean variant for templates whose only mask is the
// window-limit rectangle (Scarp, Ricker); FULL = true adds the error masks and
// the explicit per-cell masks of generic plugins.
struct RowArgs {
    int Ty, Py, Qx, circ_y, circ_x;     // tile geometry
    int cy0, cx0, cw;                   // core origin and width
    int pair, first, G;
};

template <int TX, bool FULL>
__global__ void __launch_bounds__(fft_threads(TX))
k_inv_rows(const float2* __restrict__ yw, const float2* __restrict__ ym,
           RowArgs ra, Geom g, const TileDev* __restrict__ tiles,
           const TemplDev* __restrict__ templ, const double* __restrict__ sums,
           const double* __restrict__ xaxis, const double* __restrict__ yaxis,
           const float2* __restrict__ tw, float* __restrict__ best_snr,
           float* __restrict__ best_amp, uint32_t* __restrict__ best_id,
           float* __restrict__ map_amp, float* __restrict__ map_snr) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    constexpr int NT = fft_threads(TX);
    constexpr int E = 4 * TX / NT;
    const int rb = blockIdx.x;
    const size_t plane = (size_t)ra.Ty * TX;
    const TileDev* tl = tiles + 2 * ra.pair;
    const float scale = 1.0f / ((float)ra.Ty * (float)TX);
    for (int gi_ = 0; gi_ < ra.G; ++gi_) {
        const TemplDev* tp = templ + ra.first + gi_;
        const EpiScal es = sc_epi_scalars(sums, ra.first + gi_);
        const float scale_w = scale / sc_fft_alpha(sums, ra.first + gi_);
        const float2* in1 = yw + (size_t)gi_ * plane + (size_t)rb * 4 * TX;
        const float2* in2 = ym + (size_t)gi_ * plane + (size_t)rb * 4 * TX;
        __syncthreads();
        for (int e = threadIdx.x; e < 4 * TX; e += NT) {
            int cb = e >> 4, rr = (e >> 2) & 3, cc = e & 3;
            sm[rr * TX + 4 * cb + cc] = in1[e];
        }
        __syncthreads();
        fft4_lines<TX, true>(sm, tw);
        float2 xc[E];
#pragma unroll
        for (int u = 0; u < E; ++u) xc[u] = sm[threadIdx.x + u * NT];
        __syncthreads();
        for (int e = threadIdx.x; e < 4 * TX; e += NT) {
            int cb = e >> 4, rr = (e >> 2) & 3, cc = e & 3;
            sm[rr * TX + 4 * cb + cc] = in2[e];
        }
        __syncthreads();
        fft4_lines<TX, true>(sm, tw);
        const int ilo = tp->ilo, ihi = tp->ihi, jlo = tp->jlo, jhi = tp->jhi;
        const uint32_t tid_ = tp->id;
#pragma unroll
        for (int u = 0; u < E; ++u) {
            int e = threadIdx.x + u * NT;
            int rr = e / TX, s = e - rr * TX;
            float2 t3 = sm[e];
            int ri = 4 * rb + rr - ra.Py;
            int cj = s - ra.Qx;
            if (ra.circ_y) ri &= (ra.Ty - 1);
            if (ra.circ_x) cj &= (TX - 1);
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const TileDev* tile = tl + part;
                if (ri < 0 || ri >= tile->vy || cj < 0 || cj >= tile->vx) continue;
                int gi = tile->i0 + ri, gj = tile->j0 + cj;
                float amp, snr;
                sc_epilogue((part ? xc[u].y : xc[u].x) * scale_w,
                            (part ? t3.y : t3.x) * scale, es, amp, snr);
                if (FULL) {
                    sc_apply_masks(*tp, g, xaxis, yaxis, gi, gj, amp, snr);
                } else if (gi < ilo || gi > ihi || gj < jlo || gj > jhi) {
                    amp = 0.f;
                    snr = 0.f;
                }
                size_t o = (size_t)(gi - ra.cy0) * ra.cw + (gj - ra.cx0);
                if (map_amp) {
                    map_amp[o] = amp;
                    map_snr[o] = snr;
                } else if (snr != 0.f) {
                    // snr == 0 can change nothing (best >= 0, or sticky NaN)
                    float b_snr = best_snr[o], b_amp = 0.f;
                    uint32_t b_id = SC_ID_NONE;
                    if (sc_fold(b_snr, b_amp, b_id, snr, amp, tid_)) {
                        best_snr[o] = b_snr;
                        best_amp[o] = b_amp;
                        best_id[o] = b_id;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
bool fft_size_supported(int T) {
    return T >= 64 && T <= 4096 && (T & (T - 1)) == 0;
}

static int upload_twiddles(sc_ctx* ctx, DevBuf& buf, int& have, int T) {
    if (have == T) return SC_OK;
    std::vector<float2> h(T);
    for (int k = 0; k < T; ++k) {
        double a = -2.0 * M_PI * (double)k / (double)T;
        h[k] = make_float2((float)cos(a), (float)sin(a));
    }
    int rc = sc_ensure(ctx, buf, sizeof(float2) * T);
    if (rc) return rc;
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SC_HIP(ctx, hipMemcpy(buf.p, h.data(), sizeof(float2) * T, hipMemcpyHostToDevice));
    have = T;
    return SC_OK;
}

static int npairs_of(const FftGeom& fg) { return (fg.ntiles + 1) / 2; }

int fft_prepare(sc_ctx* ctx, const FftGeom& fg, int n_templ_chunk, int group) {
    if (!fft_size_supported(fg.Ty) || !fft_size_supported(fg.Tx))
        return sc_fail(ctx, SC_ERR_UNSUPPORTED, "FFT tile %dx%d not supported", fg.Ty, fg.Tx);
    int rc;
    if ((rc = upload_twiddles(ctx, ctx->tw_y, ctx->tw_Ty, fg.Ty))) return rc;
    if ((rc = upload_twiddles(ctx, ctx->tw_x, ctx->tw_Tx, fg.Tx))) return rc;
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
    if ((rc = sc_ensure(ctx, ctx->tiles, sizeof(TileDev) * h.size()))) return rc;
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SC_HIP(ctx, hipMemcpy(ctx->tiles.p, h.data(), sizeof(TileDev) * h.size(), hipMemcpyHostToDevice));
    size_t plane = (size_t)fg.Ty * fg.Tx * sizeof(float2);
    size_t nblk = std::max((size_t)2 * np, (size_t)n_templ_chunk);
    if ((rc = sc_ensure(ctx, ctx->blk, plane * nblk))) return rc;
    if ((rc = sc_ensure(ctx, ctx->uc, plane * np))) return rc;
    if ((rc = sc_ensure(ctx, ctx->uc2, plane * np))) return rc;
    if ((rc = sc_ensure(ctx, ctx->vh, plane * n_templ_chunk))) return rc;
    if ((rc = sc_ensure(ctx, ctx->yw, plane * group))) return rc;
    if ((rc = sc_ensure(ctx, ctx->ym, plane * group))) return rc;
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
                           float2* out0, float2* out1, int split2) {
    size_t lds = (size_t)4 * fg.Ty * sizeof(float2);
    dim3 grid(fg.Tx / 4, nplanes);
    sc_prof_begin(ctx, SC_K_FWD_COLS);
#define FN(T)                                                                  \
    {                                                                          \
        int rc = set_lds(ctx, k_fwd_cols<T>, lds);                             \
        if (rc) return rc;                                                     \
        hipLaunchKernelGGL(k_fwd_cols<T>, grid, dim3(fft_threads(T)), lds,     \
                           ctx->stream, (const float2*)ctx->blk.p, fg.Tx,      \
                           (const float2*)ctx->tw_y.p, out0, out1, split2);    \
    }
    DISPATCH_T(fg.Ty, FN)
#undef FN
    sc_prof_end(ctx);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

int fft_forward_curv(sc_ctx* ctx, const FftGeom& fg) {
    int np = npairs_of(fg);
    size_t lds = (size_t)4 * fg.Tx * sizeof(float2);
    dim3 grid(fg.Ty / 4, np);
    sc_prof_begin(ctx, SC_K_FWD_ROWS);
#define FN(T)                                                                  \
    {                                                                          \
        int rc = set_lds(ctx, k_fwd_rows_curv<T>, lds);                        \
        if (rc) return rc;                                                     \
        hipLaunchKernelGGL(k_fwd_rows_curv<T>, grid, dim3(fft_threads(T)),     \
                           lds, ctx->stream, (const float*)ctx->curv.p,        \
                           ctx->g, (const TileDev*)ctx->tiles.p, fg.Ty,        \
                           (const float2*)ctx->tw_x.p, (float2*)ctx->blk.p);   \
    }
    DISPATCH_T(fg.Tx, FN)
#undef FN
    sc_prof_end(ctx);
    SC_HIP(ctx, hipGetLastError());
    return launch_fwd_cols(ctx, fg, 2 * np, (float2*)ctx->uc.p, (float2*)ctx->uc2.p, 1);
}

int fft_forward_templates(sc_ctx* ctx, const FftGeom& fg, int first, int n) {
    size_t lds = (size_t)4 * fg.Tx * sizeof(float2);
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
    return launch_fwd_cols(ctx, fg, n, (float2*)ctx->vh.p, nullptr, 0);
}

// Templates [first, first+n) of the current batch have their spectra in vh
// planes [0, n).  For every tile pair: inverse transforms in groups of `group`
// templates, folded in template order.
int fft_inverse_fold(sc_ctx* ctx, const FftGeom& fg, int first, int n,
                     int group, bool to_maps, bool full_masks) {
    int np = npairs_of(fg);
    size_t lds_c = (size_t)4 * fg.Ty * sizeof(float2);
    size_t lds_r = (size_t)4 * fg.Tx * sizeof(float2);
    for (int pair = 0; pair < np; ++pair) {
        for (int g0 = 0; g0 < n; g0 += group) {
            int G = std::min(group, n - g0);
            dim3 gridc(fg.Tx / 4, G);
            sc_prof_begin(ctx, SC_K_INV_COLS);
#define FN(T)                                                                  \
    {                                                                          \
        int rc = set_lds(ctx, k_inv_cols<T>, lds_c);                           \
        if (rc) return rc;                                                     \
        hipLaunchKernelGGL(k_inv_cols<T>, gridc, dim3(fft_threads(T)), lds_c,  \
                           ctx->stream, (const float2*)ctx->uc.p,              \
                           (const float2*)ctx->uc2.p, (const float2*)ctx->vh.p,\
                           fg.Tx, pair, g0, (const float2*)ctx->tw_y.p,        \
                           (float2*)ctx->yw.p, (float2*)ctx->ym.p);            \
    }
            DISPATCH_T(fg.Ty, FN)
#undef FN
            sc_prof_end(ctx);
            dim3 gridr(fg.Ty / 4);
            RowArgs ra{fg.Ty, fg.Py, fg.Qx, fg.circ_y, fg.circ_x, ctx->g.cy0, ctx->g.cx0,
                       ctx->g.cx1 - ctx->g.cx0, pair, first + g0, G};
            sc_prof_begin(ctx, SC_K_INV_ROWS);
#define LAUNCH_ROWS(T, FULLV)                                                  \
    {                                                                          \
        int rc = set_lds(ctx, k_inv_rows<T, FULLV>, lds_r);                    \
        if (rc) return rc;                                                     \
        hipLaunchKernelGGL((k_inv_rows<T, FULLV>), gridr, dim3(fft_threads(T)),\
                           lds_r, ctx->stream, (const float2*)ctx->yw.p,       \
                           (const float2*)ctx->ym.p, ra, ctx->g,               \
                           (const TileDev*)ctx->tiles.p,                       \
                           (const TemplDev*)ctx->templ.p,                      \
                           (const double*)ctx->sums.p,                         \
                           (const double*)ctx->xaxis.p,                        \
                           (const double*)ctx->yaxis.p,                        \
                           (const float2*)ctx->tw_x.p,                         \
                           (float*)ctx->best_snr.p, (float*)ctx->best_amp.p,   \
                           (uint32_t*)ctx->best_id.p,                          \
                           to_maps ? (float*)ctx->map_amp.p : nullptr,         \
                           to_maps ? (float*)ctx->map_snr.p : nullptr);        \
    }
#define FN(T) { if (full_masks) LAUNCH_ROWS(T, true) else LAUNCH_ROWS(T, false) }
            DISPATCH_T(fg.Tx, FN)
#undef FN
#undef LAUNCH_ROWS
            sc_prof_end(ctx);
        }
    }
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}
