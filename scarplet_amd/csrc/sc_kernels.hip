// Curvature stencils, template-window synthesis and the real-space
// (sliding window) matcher.  gfx950 only.
#include "sc_internal.h"
#include <algorithm>

// ---------------------------------------------------------------------------
// K1: the three alpha-independent curvature stencils (dem.py:88-101).
// float64 in (the reference differences float64 elevations), float32 out.
// Borders follow the GLOBAL cell index: d2z/dx2 is zero in the first and last
// DEM column, d2z/dxdy in the first row/column, d2z/dy2 in the first and last
// row.  One thread per cell, rows of the block are read as coalesced float64.
// ---------------------------------------------------------------------------
template <typename OUT>          // float: the matcher's planes; double: the float64 scorer's (sc_score_cells_f64)
__global__ void __launch_bounds__(256)
k_curv_planes(const double* __restrict__ z, Geom g, double dx, double dy,
              OUT* __restrict__ A, OUT* __restrict__ B,
              OUT* __restrict__ C) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int i = blockIdx.y;
    if (j >= g.lx) return;
    int gi = wrap_index(g.gy0 + i, g.ny);
    int gj = wrap_index(g.gx0 + j, g.nx);
    // neighbours inside the local block (clamped: cells whose neighbour is
    // missing lie in the outermost ring of a halo block and are never used)
    int im = max(i - 1, 0), ip = min(i + 1, g.ly - 1);
    int jm = max(j - 1, 0), jp = min(j + 1, g.lx - 1);
    const double* r0 = z + (size_t)im * g.lx;
    const double* r1 = z + (size_t)i * g.lx;
    const double* r2 = z + (size_t)ip * g.lx;
    double z11 = r1[j];
    double a = 0.0, b = 0.0, c = 0.0;
    if (gj >= 1 && gj <= g.nx - 2)          // np.diff(z, 2, 1) / dx**2
        a = __ddiv_rn(__dsub_rn(__dsub_rn(r1[jp], z11), __dsub_rn(z11, r1[jm])),
                      __dmul_rn(dx, dx));
    if (gi >= 1 && gj >= 1) {               // np.diff(np.diff(z,1,1)/dx,1,0)/dx
        double d1 = __ddiv_rn(__dsub_rn(z11, r1[jm]), dx);
        double d0 = __ddiv_rn(__dsub_rn(r0[j], r0[jm]), dx);
        b = __ddiv_rn(__dsub_rn(d1, d0), dx);
    }
    if (gi >= 1 && gi <= g.ny - 2)          // np.diff(z, 2, 0) / dy**2
        c = __ddiv_rn(__dsub_rn(__dsub_rn(r2[j], z11), __dsub_rn(z11, r0[j])),
                      __dmul_rn(dy, dy));
    size_t o = (size_t)i * g.lx + j;
    A[o] = (OUT)a;
    B[o] = (OUT)b;
    C[o] = (OUT)c;
}

// The reference's curvature as its data object returns it (dem.py:68-107), float64 out: the three
// stencils of k_curv_planes and dem.py:103-104's combination
//   d2z_dx2 * cos(a)**2 - 2 * d2z_dxdy * sin(a) * cos(a) + d2z_dy2 * sin(a)**2
// in numpy's evaluation order (no contraction); c2 = cos(a)**2, s2 = sin(a)**2 come from the host.
// Serves DEMGrid._calculate_directional_laplacian; the matcher keeps its float32 planes.
__global__ void __launch_bounds__(256)
k_curv_f64(const double* __restrict__ z, Geom g, double dx, double dy, double c2, double sn,
           double cs, double s2, double* __restrict__ out) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int i = blockIdx.y;
    if (j >= g.lx) return;
    int gi = wrap_index(g.gy0 + i, g.ny);
    int gj = wrap_index(g.gx0 + j, g.nx);
    int im = max(i - 1, 0), ip = min(i + 1, g.ly - 1);
    int jm = max(j - 1, 0), jp = min(j + 1, g.lx - 1);
    const double* r0 = z + (size_t)im * g.lx;
    const double* r1 = z + (size_t)i * g.lx;
    const double* r2 = z + (size_t)ip * g.lx;
    double z11 = r1[j];
    double a = 0.0, b = 0.0, c = 0.0;
    if (gj >= 1 && gj <= g.nx - 2)
        a = __ddiv_rn(__dsub_rn(__dsub_rn(r1[jp], z11), __dsub_rn(z11, r1[jm])),
                      __dmul_rn(dx, dx));
    if (gi >= 1 && gj >= 1) {
        double d1 = __ddiv_rn(__dsub_rn(z11, r1[jm]), dx);
        double d0 = __ddiv_rn(__dsub_rn(r0[j], r0[jm]), dx);
        b = __ddiv_rn(__dsub_rn(d1, d0), dx);
    }
    if (gi >= 1 && gi <= g.ny - 2)
        c = __ddiv_rn(__dsub_rn(__dsub_rn(r2[j], z11), __dsub_rn(z11, r0[j])),
                      __dmul_rn(dy, dy));
    const double t1 = __dmul_rn(a, c2);
    const double t2 = __dmul_rn(__dmul_rn(__dmul_rn(2.0, b), sn), cs);
    const double t3 = __dmul_rn(c, s2);
    out[(size_t)i * g.lx + j] = __dadd_rn(__dsub_rn(t1, t2), t3);
}

// ---------------------------------------------------------------------------
// match_template() at single cells in FLOAT64 (core.py:297-377 as the real-space closed form): the last step of
// the host layer's exact mode.  The float32 paths decide a cell's argmax to within their own rounding - 2e-4
// (FFT tiles), 1e-5 (real space) of the SNR; where two templates lie closer than that the answer is settled here
// the way the reference settles it, in float64: for every (cell, template) of the list
//   xcorr = sum W[p, q] curv[(i - p + oy) % ny, (j - q + ox) % nx],  T3 likewise with (W != 0), curv**2
// over the template's support box, W evaluated with k_windows' float64 expressions, the curvature with
// k_curv_f64's (stencils of dem.py:88-101 on the float64 elevations, global borders zero), then core.py:360-375.
// grid = (cells, templates), one workgroup each; the support box is dealt out over the threads.
// Built-in templates only (a generic plugin's window exists in float32 on the device).
// ---------------------------------------------------------------------------
// The float64 windows of the scorer, once per template instead of once per (cell, template): W over the template's support
// box, 0 where the reference's W is 0 (outside the grid, outside |xr| < c & |yr| < d, at xr = 0, beyond the float64
// underflow of a Ricker's exponential - core.py:348's M is literally W != 0).  grid = (ceil(largest box / 256), templates).
__global__ void __launch_bounds__(256)
k_window_f64(const TemplDev* __restrict__ templ, Geom g, const double* __restrict__ xaxis, const double* __restrict__ yaxis,
             const unsigned long long* __restrict__ woff, double* __restrict__ wbuf) {
    const TemplDev t = templ[blockIdx.y];
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= t.wh * t.ww) return;
    const int a = e / t.ww, b = e - a * t.ww;
    const int k = g.ny / 2 + t.pmin + a, l = g.nx / 2 + t.qmin + b;
    double w = 0.0;
    if (k >= 0 && k < g.ny && l >= 0 && l < g.nx) {
        const double x = xaxis[l], y = yaxis[k];
        const double xr = __dadd_rn(__dmul_rn(x, t.cos_a), __dmul_rn(y, t.sin_a));
        const double yr = __dadd_rn(__dmul_rn(-x, t.sin_a), __dmul_rn(y, t.cos_a));
        if ((fabs(xr) < t.c) && (fabs(yr) < t.d)) {
            if (t.kind == SC_KIND_SCARP) {
                if (xr != 0.0) w = __dmul_rn(__ddiv_rn(-xr, t.p0), exp(__ddiv_rn(-__dmul_rn(xr, xr), t.p1)));
            } else {
                const double u = __dmul_rn(t.p0, xr), u2 = __dmul_rn(u, u);
                const double poly = __dsub_rn(1.0, __dmul_rn(2.0, u2));
                if ((u2 < SC_EXP_UNDERFLOW) && (poly != 0.0)) w = __dmul_rn(poly, exp(-u2));
            }
            if (t.flags & SC_FLAG_NEGATE) w = -w;
        }
    }
    wbuf[woff[blockIdx.y] + e] = w;
}

__global__ void __launch_bounds__(256)
k_score_f64(const double* __restrict__ pa, const double* __restrict__ pb, const double* __restrict__ pc, Geom g,
            const TemplDev* __restrict__ templ, int n_templ, const double* __restrict__ sums,
            const double* __restrict__ xaxis, const double* __restrict__ yaxis,
            const unsigned long long* __restrict__ woff, const double* __restrict__ wbuf,
            const int* __restrict__ cells, const int* __restrict__ tsel, double* __restrict__ amp_out, double* __restrict__ snr_out) {
    // tsel: (cell, template) PAIRS - block ci scores cell ci against template tsel[ci] (grid.y = 1) - instead of the full table
    const int ci = blockIdx.x, it = tsel ? tsel[ci] : (int)blockIdx.y;
    const int i = cells[2 * ci], j = cells[2 * ci + 1];                  // global cell
    const TemplDev t = templ[it];
    // curvature mix of this template's orientation: cc, sc2, ss are not in TemplDev - the same expression as the
    // reference's from its alpha = -orientation: cos(a)^2, 2 sin(a) cos(a), sin(a)^2 with a = -alpha, i.e. the
    // orientation; cos_a / sin_a of the descriptor are those of alpha
    const double ca = t.cos_a, sa = -t.sin_a;                             // cos / sin of the ORIENTATION
    const double k_cc = __dmul_rn(ca, ca), k_ss = __dmul_rn(sa, sa);
    const double* __restrict__ wt = wbuf + woff[it];
    double xc = 0.0, t3 = 0.0;
    const int box = t.wh * t.ww;
    for (int e = threadIdx.x; e < box; e += 256) {
        const double w = wt[e];
        if (w == 0.0) continue;                                           // (outside the support: W != 0 is the mask M)
        const int a = e / t.ww, b = e - a * t.ww;
        const int p = t.pmin + a, q = t.qmin + b;
        // curvature at global ((i - p + oy) mod ny, (j - q + ox) mod nx): its local position in the block
        int gi = i - p + g.oy, gj = j - q + g.ox;
        int li, lj;
        if (g.wrap) { gi = wrap_index(gi, g.ny); gj = wrap_index(gj, g.nx); li = gi; lj = gj; }
        else { li = gi - g.gy0; lj = gj - g.gx0; gi = wrap_index(gi, g.ny); gj = wrap_index(gj, g.nx); }
        if (li < 0 || li >= g.ly || lj < 0 || lj >= g.lx) continue;       // (outside the block: the host sized the halo)
        // the stencils of dem.py:88-101 on the float64 elevations: k_curv_planes<double>, once per call (four float64
        // divisions per tap, done here, were most of this kernel)
        const size_t o = (size_t)li * g.lx + lj;
        const double A = pa[o], Bc = pb[o], C = pc[o];
        // dem.py:103-104: d2z_dx2 cos^2 - 2 d2z_dxdy sin cos + d2z_dy2 sin^2 (numpy's order)
        const double cv = __dadd_rn(__dsub_rn(__dmul_rn(A, k_cc), __dmul_rn(__dmul_rn(__dmul_rn(2.0, Bc), sa), ca)),
                                    __dmul_rn(C, k_ss));
        xc = fma(w, cv, xc);
        t3 = fma(cv, cv, t3);
    }
    __shared__ double red[2][4];
    for (int sft = 32; sft > 0; sft >>= 1) {
        xc += __shfl_down(xc, sft, 64);
        t3 += __shfl_down(t3, sft, 64);
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = xc; red[1][threadIdx.x >> 6] = t3; }
    __syncthreads();
    if (threadIdx.x == 0) {
        xc = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        t3 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const double n = sums[2 * it] + SC_EPS, ts = sums[2 * it + 1];
        double amp = xc / ts;
        const double T1 = ts * (amp * amp);
        const double err = (1.0 / n) * (T1 - 2.0 * amp * xc + t3) + SC_EPS;
        double snr = fabs(T1 / err);
        if (t.flags & (SC_FLAG_ERR_XR_LE0 | SC_FLAG_ERR_XR_GE0)) {
            const double xr = __dadd_rn(__dmul_rn(xaxis[j], t.cos_a), __dmul_rn(yaxis[i], t.sin_a));
            if ((t.flags & SC_FLAG_ERR_XR_LE0) ? (xr <= 0.0) : (xr >= 0.0)) snr = 0.0;
        }
        if (!(i >= t.ilo && i <= t.ihi && j >= t.jlo && j <= t.jhi)) { amp = 0.0; snr = 0.0; }
        const size_t oo = tsel ? (size_t)ci : (size_t)ci * n_templ + it;
        amp_out[oo] = amp;
        snr_out[oo] = snr;
    }
}

// What the float64 scorers read, rebuilt at every call: the templates' float64 windows (offsets from the host's copy of the
// last search's descriptors, one kernel for all templates) and the three stencil planes of the block in float64.
int score_prepare_f64(sc_ctx* ctx, int n_templ, const unsigned long long** woff_out, const double** wbuf_out, const double** planes_out) {
    std::vector<unsigned long long> off((size_t)n_templ + 1, 0ull);
    int maxbox = 1;
    for (int k = 0; k < n_templ; ++k) {
        const int box = ctx->h_templ[k].wh * ctx->h_templ[k].ww;
        off[k + 1] = off[k] + (unsigned long long)box;
        maxbox = std::max(maxbox, box);
    }
    const size_t obytes = sizeof(unsigned long long) * ((size_t)n_templ + 1);
    int rc = sc_ensure(ctx, ctx->score_w, obytes + sizeof(double) * (size_t)off[n_templ] + 64);
    if (rc) return rc;
    unsigned long long* woff = (unsigned long long*)ctx->score_w.p;
    double* wbuf = (double*)((char*)ctx->score_w.p + ((obytes + 63) & ~(size_t)63));
    SC_HIP(ctx, hipMemcpyAsync(woff, off.data(), obytes, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));                       // (`off` is a local)
    hipLaunchKernelGGL(k_window_f64, dim3((maxbox + 255) / 256, n_templ), dim3(256), 0, ctx->stream,
                       (const TemplDev*)ctx->templ.p, ctx->g, (const double*)ctx->xaxis.p, (const double*)ctx->yaxis.p,
                       (const unsigned long long*)woff, wbuf);
    SC_HIP(ctx, hipGetLastError());
    const size_t nc = (size_t)ctx->g.ly * ctx->g.lx;
    if ((rc = sc_ensure(ctx, ctx->score_abc, 3 * nc * sizeof(double)))) return rc;
    double* pa = (double*)ctx->score_abc.p;
    hipLaunchKernelGGL(k_curv_planes<double>, dim3((ctx->g.lx + 255) / 256, ctx->g.ly), dim3(256), 0, ctx->stream,
                       ctx->z_dev, ctx->g, ctx->dx, ctx->dy, pa, pa + nc, pa + 2 * nc);
    SC_HIP(ctx, hipGetLastError());
    *woff_out = woff;
    *wbuf_out = wbuf;
    *planes_out = pa;
    return SC_OK;
}

int launch_score_f64(sc_ctx* ctx, const int* cells_dev, const int* tsel_dev, int m, int n_templ, double* amp_dev, double* snr_dev) {
    const unsigned long long* woff = nullptr;
    const double *wbuf = nullptr, *pa = nullptr;
    int rc = score_prepare_f64(ctx, n_templ, &woff, &wbuf, &pa);
    if (rc) return rc;
    const size_t nc = (size_t)ctx->g.ly * ctx->g.lx;
    hipLaunchKernelGGL(k_score_f64, dim3(m, tsel_dev ? 1 : n_templ), dim3(256), 0, ctx->stream, pa, pa + nc, pa + 2 * nc, ctx->g,
                       (const TemplDev*)ctx->templ.p, n_templ, (const double*)ctx->sums.p, (const double*)ctx->xaxis.p,
                       (const double*)ctx->yaxis.p, woff, wbuf, cells_dev, tsel_dev, amp_dev, snr_dev);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

// Digest of the elevation block as it sits in HBM: the number of NaN cells (one NaN turns every reference
// output NaN, core.py:349-363 - the host answers such a DEM without a search) and a 128-bit fingerprint of
// the float64 bit patterns, position-dependent and summed (order-free, so plain atomics do).  The host used
// to walk the 800 MB of a 10000 x 10000 DEM twice for the two (np.isnan, a hash): 53 ms of every call; here
// it is one pass at HBM speed over data the device holds anyway.
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__global__ void __launch_bounds__(256)
k_dem_digest(const unsigned long long* __restrict__ z, size_t n, unsigned long long* __restrict__ out) {
    unsigned long long h0 = 0, h1 = 0, nan = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const unsigned long long w = z[i];
        h0 += mix64(w ^ (0x9E3779B97F4A7C15ull * (i + 1)));
        h1 += mix64((w + 0xD6E8FEB86659FD93ull) ^ (0xC2B2AE3D27D4EB4Full * (i + 1)));
        nan += ((w & 0x7FFFFFFFFFFFFFFFull) > 0x7FF0000000000000ull) ? 1ull : 0ull;
    }
    for (int sft = 32; sft > 0; sft >>= 1) {
        h0 += __shfl_down(h0, sft, 64);
        h1 += __shfl_down(h1, sft, 64);
        nan += __shfl_down(nan, sft, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(out, h0);
        atomicAdd(out + 1, h1);
        if (nan) atomicAdd(out + 2, nan);
    }
}
int launch_dem_digest(sc_ctx* ctx, unsigned long long* out_dev) {
    const size_t n = (size_t)ctx->g.ly * ctx->g.lx;
    SC_HIP(ctx, hipMemsetAsync(out_dev, 0, 3 * sizeof(unsigned long long), ctx->stream));
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(k_dem_digest, dim3(blocks), dim3(256), 0, ctx->stream,
                       (const unsigned long long*)ctx->z_dev, n, out_dev);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

int launch_curv_f64(sc_ctx* ctx, double c2, double sn, double cs, double s2, double* out_dev) {
    const Geom& g = ctx->g;
    dim3 grid((g.lx + 255) / 256, g.ly);
    sc_prof_begin(ctx, SC_K_CURV);
    hipLaunchKernelGGL(k_curv_f64, grid, dim3(256), 0, ctx->stream, ctx->z_dev, g, ctx->dx, ctx->dy,
                       c2, sn, cs, s2, out_dev);
    sc_prof_end(ctx);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

// curv = cc*A - sc2*B + ss*C (dem.py:103-104), float4 per thread.
__global__ void __launch_bounds__(256)
k_curv_alpha(const float* __restrict__ A, const float* __restrict__ B,
             const float* __restrict__ C, float cc, float sc2, float ss,
             float* __restrict__ out, size_t n) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (; i < n; i += stride) {
        if (i + 3 < n) {
            float4 a = *reinterpret_cast<const float4*>(A + i);
            float4 b = *reinterpret_cast<const float4*>(B + i);
            float4 c = *reinterpret_cast<const float4*>(C + i);
            float4 o;
            o.x = cc * a.x - sc2 * b.x + ss * c.x;
            o.y = cc * a.y - sc2 * b.y + ss * c.y;
            o.z = cc * a.z - sc2 * b.z + ss * c.z;
            o.w = cc * a.w - sc2 * b.w + ss * c.w;
            *reinterpret_cast<float4*>(out + i) = o;
        } else {
            for (size_t k = i; k < n; ++k)
                out[k] = cc * A[k] - sc2 * B[k] + ss * C[k];
        }
    }
}

// ---------------------------------------------------------------------------
// K2: template windows.  For every cell of a template's support bounding box
// evaluate, in float64 and with the reference's operation order, the rotated
// coordinates (WindowedTemplate.py:56-57), the window test (l.63) and the
// profile (Scarp l.177-178, Ricker l.514-515), and write the dense window
// W (float32) and M = (W != 0) (uint8).  count(M) and sum(W**2) are reduced
// per wave and added with one float64 atomic per wave.
// grid = (ceil(ww_max/64), ceil(wh_max/16), n_templates), block = 64 (one wave, 16 window rows).
// ---------------------------------------------------------------------------
#define SC_WIN_ROWS 16
__global__ void __launch_bounds__(64)
k_windows(const TemplDev* __restrict__ templ, int first,
          const double* __restrict__ xaxis, const double* __restrict__ yaxis,
          int ny, int nx, float* __restrict__ win_w,
          uint8_t* __restrict__ win_m, double* __restrict__ sums,
          double* __restrict__ wl1) {
    const int it = first + blockIdx.z;
    const TemplDev t = templ[it];
    if (t.kind == SC_KIND_WINDOW) return;         // uploaded by the host
    const int b = blockIdx.x * 64 + threadIdx.x;
    double cnt = 0.0, sq = 0.0, ab = 0.0;
    // SC_WIN_ROWS window rows per wave: a sixteenth of the same-address float64 atomics below
    // (1 540 waves per 308 x 308 window each added three of them: 2.5 of C2's 40 ms)
    for (int a = blockIdx.y * SC_WIN_ROWS; a < min(t.wh, (int)(blockIdx.y + 1) * SC_WIN_ROWS); ++a) {
        if (b >= t.ww) break;
        int k = ny / 2 + t.pmin + a;
        int l = nx / 2 + t.qmin + b;
        double x = xaxis[l], y = yaxis[k];
        double xr = __dadd_rn(__dmul_rn(x, t.cos_a), __dmul_rn(y, t.sin_a));
        double yr = __dadd_rn(__dmul_rn(-x, t.sin_a), __dmul_rn(y, t.cos_a));
        bool inside = (fabs(xr) < t.c) && (fabs(yr) < t.d);
        double w = 0.0;
        bool m = false;
        if (inside) {
            if (t.kind == SC_KIND_SCARP) {
                w = __dmul_rn(__ddiv_rn(-xr, t.p0),
                              exp(__ddiv_rn(-__dmul_rn(xr, xr), t.p1)));
                m = (xr != 0.0);
            } else {
                double u = __dmul_rn(t.p0, xr);
                double u2 = __dmul_rn(u, u);
                double poly = __dsub_rn(1.0, __dmul_rn(2.0, u2));
                m = (u2 < SC_EXP_UNDERFLOW) && (poly != 0.0);
                w = m ? __dmul_rn(poly, exp(-u2)) : 0.0;
            }
            if (!m) w = 0.0;
            if (t.flags & SC_FLAG_NEGATE) w = -w;
        }
        size_t o = (size_t)t.win_off + (size_t)a * t.ww + b;
        win_w[o] = (float)w;
        win_m[o] = m ? 1 : 0;
        cnt += m ? 1.0 : 0.0;
        sq += w * w;
        ab += fabs(w);
    }
    for (int s = 32; s > 0; s >>= 1) {
        cnt += __shfl_down(cnt, s, 64);
        sq += __shfl_down(sq, s, 64);
        ab += __shfl_down(ab, s, 64);
    }
    if (threadIdx.x == 0 && (cnt != 0.0 || sq != 0.0)) {
        atomicAdd(&sums[2 * it + 0], cnt);
        atomicAdd(&sums[2 * it + 1], sq);
        atomicAdd(&wl1[it], ab);
    }
}

// ---------------------------------------------------------------------------
// K3 (real-space): sliding-window correlation of a batch of templates that
// share one curvature plane, with the amp/SNR epilogue and the running-best
// fold fused.  A workgroup owns a TY x TX patch of output cells and stages
// the curvature it needs in LDS in slabs of template rows:
//
//   for each template t of the batch (all read the same curvature plane)
//     for each slab of SR template rows
//       LDS <- curvature rows/cols the patch needs for these template rows
//       every thread accumulates its PX cells:  xc += W*c,  t3 += M*c*c
//     epilogue (float64) + masks + fold into the thread's running best
//   one read-modify-write of the best planes per cell
//
// Curvature rows are loaded as coalesced float runs; window taps are
// wave-uniform (scalar loads); each thread keeps PX adjacent outputs so an
// LDS value is reused PX times from registers.
// ---------------------------------------------------------------------------
#define DR_TX 64          // patch width  (cells)
#define DR_TY 16          // patch height (cells)
#define DR_PX 4           // outputs per thread along x
#define DR_THREADS (DR_TX / DR_PX * DR_TY)     // 256
#define DR_LDS_FLOATS (36 * 1024)              // 144 KiB

__global__ void __launch_bounds__(DR_THREADS)
k_direct(const float* __restrict__ curv, Geom g,
         const TemplDev* __restrict__ templ, int first, int n_templ,
         const float* __restrict__ win_w, const uint8_t* __restrict__ win_m,
         const double* __restrict__ sums, const double* __restrict__ xaxis,
         const double* __restrict__ yaxis, float* __restrict__ best_snr,
         float* __restrict__ best_amp, uint32_t* __restrict__ best_id,
         float* __restrict__ map_amp, float* __restrict__ map_snr) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tx = threadIdx.x % (DR_TX / DR_PX);
    const int ty = threadIdx.x / (DR_TX / DR_PX);
    const int i0 = g.cy0 + blockIdx.y * DR_TY;       // patch origin (global)
    const int j0 = g.cx0 + blockIdx.x * DR_TX;
    const int cw = g.cx1 - g.cx0;
    const int gi = i0 + ty;
    const int gjb = j0 + tx * DR_PX;

    float b_snr[DR_PX], b_amp[DR_PX];
    uint32_t b_id[DR_PX];
    bool dirty[DR_PX];
#pragma unroll
    for (int u = 0; u < DR_PX; ++u) {
        bool in = gi < g.cy1 && (gjb + u) < g.cx1;
        size_t o = (size_t)(gi - g.cy0) * cw + (gjb + u - g.cx0);
        b_snr[u] = (in && !map_amp) ? best_snr[o] : 0.f;
        b_amp[u] = 0.f;
        b_id[u] = SC_ID_NONE;
        dirty[u] = false;
    }

    for (int it = 0; it < n_templ; ++it) {
        const TemplDev t = templ[first + it];
        const EpiScal es = sc_epi_scalars(sums, first + it);
        float xc[DR_PX], t3[DR_PX];
#pragma unroll
        for (int u = 0; u < DR_PX; ++u) { xc[u] = 0.f; t3[u] = 0.f; }

        // LDS slab: rows for template rows [a0, a0+sr): output row r, template
        // row p reads curvature row r - p + oy.  Slab covers curvature rows
        // i0 - (pmin+a0+sr-1) + oy ... i0 + DR_TY-1 - (pmin+a0) + oy
        const int lw = DR_TX + t.ww - 1;              // slab width in cells
        const int lwp = lw | 1;                       // odd pitch
        int sr = DR_LDS_FLOATS / lwp - (DR_TY - 1);
        if (sr > t.wh) sr = t.wh;
        if (sr < 1) sr = 1;
        // leftmost curvature column: j0 - qmax + ox
        const int gj_left = j0 - t.qmax + g.ox;
        for (int a0 = 0; a0 < t.wh; a0 += sr) {
            const int na = min(sr, t.wh - a0);
            const int rows = DR_TY + na - 1;
            const int gi_top = i0 - (t.pmin + a0 + na - 1) + g.oy;
            __syncthreads();
            for (int e = threadIdx.x; e < rows * lw; e += DR_THREADS) {
                int r = e / lw, c = e - r * lw;
                lds[r * lwp + c] = load_curv(curv, g, gi_top + r, gj_left + c);
            }
            __syncthreads();
            for (int a = 0; a < na; ++a) {
                // template row p = pmin + a0 + a; curvature row for output
                // row (i0+ty): i0 + ty - p + oy  -> slab row ty + (na-1-a)
                const float* lrow = lds + (ty + (na - 1 - a)) * lwp;
                const float* wrow = win_w + t.win_off + (size_t)(a0 + a) * t.ww;
                const uint8_t* mrow = win_m + t.win_off + (size_t)(a0 + a) * t.ww;
                // output col gjb+u, template col q = qmin + b reads curvature
                // col gjb + u - q + ox -> slab col (tx*PX + u) + (ww-1-b)
                const int cb = tx * DR_PX + (t.ww - 1);
                // register window: entering iteration b, v[u] holds slab col
                // cb + u - b + 1; the shift below turns that into cb + u - b.
                float v[DR_PX];
#pragma unroll
                for (int u = 0; u < DR_PX - 1; ++u) v[u] = lrow[cb + u + 1];
                v[DR_PX - 1] = 0.f;
                for (int b = 0; b < t.ww; ++b) {
#pragma unroll
                    for (int u = DR_PX - 1; u > 0; --u) v[u] = v[u - 1];
                    v[0] = lrow[cb - b];
                    if (mrow[b]) {
                        const float w = wrow[b];
#pragma unroll
                        for (int u = 0; u < DR_PX; ++u) {
                            xc[u] = fmaf(w, v[u], xc[u]);
                            t3[u] = fmaf(v[u], v[u], t3[u]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < DR_PX; ++u) {
            int gj = gjb + u;
            if (gi >= g.cy1 || gj >= g.cx1) continue;
            float amp, snr;
            sc_epilogue(xc[u], t3[u], es, amp, snr);
            sc_apply_masks(t, g, xaxis, yaxis, gi, gj, amp, snr);
            if (map_amp) {
                size_t o = (size_t)(gi - g.cy0) * cw + (gj - g.cx0);
                map_amp[o] = amp;
                map_snr[o] = snr;
            } else if (sc_fold(b_snr[u], b_amp[u], b_id[u], snr, amp, t.id)) {
                dirty[u] = true;
            }
        }
    }
    if (!map_amp) {
#pragma unroll
        for (int u = 0; u < DR_PX; ++u) {
            if (!dirty[u]) continue;
            size_t o = (size_t)(gi - g.cy0) * cw + (gjb + u - g.cx0);
            best_snr[o] = b_snr[u];
            best_amp[o] = b_amp[u];
            best_id[o] = b_id[u];
        }
    }
}

// ---------------------------------------------------------------------------
// K3, round 3: the real-space path rebuilt around TAPS instead of the support box.
//
// The kernel above walks the whole bounding box of a template - an LDS read and a register
// shift per box cell, the FMAs only where the window is non-zero; a rotated Scarp window fills
// 2 - 5 % of its box, and the path ran at 1 - 3 % of the chip's FP32 rate.  Here
//
//  * k_direct_prep turns every window into rows REVERSED in x (xcorr[x] = sum_b' R[b'] *
//    slab[x + b'] is a plain correlation), padded to groups of four taps, as float2
//    (w, m = W != 0) - and records per window row the span of groups that hold a non-zero:
//    the work of a template is its taps (rounded up to four per row end), not its box;
//  * a lane owns NB blocks of FOUR adjacent outputs (columns 4 lane + 256 n) on RW rows.
//    A group of four taps needs the eight slab cells x .. x + 7 of the row: four carried from
//    the previous group, four new ones from ONE ds_read_b128 (64 lanes read 1 KB of one slab
//    row: conflict-free) - 32 FMAs for xcorr and 32 for T3 per LDS read instruction;
//  * the four (w, m) pairs of a group are wave-uniform.  Round 3: one s_load_dwordx8, scalar operands
//    of the FMAs.  Round 4: vector loads from one address (two register buffers, one group ahead) -
//    scalar loads share lgkmcnt with the LDS reads and return out of order, so every group waited
//    for all of them;
//  * T3 over a run without holes is SHARED by a lane's four adjacent outputs: the cells common to
//    the four (all but three at either end) are summed once per block, whole chunks at a time
//    (3 adds + 1 per chunk instead of 16 weighted FMAs), the end cells are read again after the
//    groups - the same addends.  +25 - 35 % on supports of thousands of taps.  (A first form with a
//    0/1 mask per cell was 25 % SLOWER than no sharing: the masks became 28 v_cndmask per chunk.)
//    Rows with holes and runs below 16 taps accumulate m * curv^2 tap by tap; runs whose every
//    template is thin take a kernel without the shared form (fewer registers: 10 % faster there)
//    - decided per orientation run, so that a template's sums never depend on its batch;
//  * eight waves per workgroup (two per SIMD: one wave alone issues a VALU instruction every
//    four cycles, two share the SIMD at two), the patch is 8 RW rows x 256 NB columns, slabs of
//    as many template rows as the 158 KB of LDS hold;
//  * per-cell float32 sums as before: exact per cell, no resolution floor;
//  * of the running best only the SNR lives in registers; the amplitude and the id of a cell are
//    stored when a template wins it (round 4: no scratch in any form of the kernel).
// ---------------------------------------------------------------------------
#define DR2_LDS_FLOATS (39 * 1024 + 512)        // 158 KB
#ifndef SC_DR_VMEMW
#define SC_DR_VMEMW 1      // the real-space kernel's weights through vector loads (0: scalar loads, rounds 3 - 4a)
#endif
#define DR2_WAVES 8
#ifndef SC_DR_SHARE_MIN
#define SC_DR_SHARE_MIN 16    // shortest hole-free run (taps) whose T3 takes the shared form (8, the least the end-cell pieces allow, is slower:
                              // 0.26 against 0.19 ms at 46 taps, 0.56 against 0.50 at 294 - profiles/r06_crossover.txt)
#endif

// grid = (ceil(wh_max / 4), n_templates), block = 256: one wave per window row.
// spans[row] = (first group, groups, s, e): taps s .. e of the span (counted from its first group's
// first tap) are the row's support when that is ONE run without holes, e = -1 otherwise.
__global__ void __launch_bounds__(256)
k_direct_prep(const TemplDev* __restrict__ templ, int first, const float* __restrict__ win_w,
              const uint8_t* __restrict__ win_m, float2* __restrict__ dwin, int4* __restrict__ spans) {
    const TemplDev t = templ[first + blockIdx.y];
    const int a = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (a >= t.wh) return;
    const float* wrow = win_w + t.win_off + (size_t)a * t.ww;
    const uint8_t* mrow = win_m + t.win_off + (size_t)a * t.ww;
    float2* out = dwin + t.dwin_off + (size_t)a * t.dpitch;
    int lo = INT_MAX, hi = -1, cnt = 0;
    for (int bp = lane; bp < t.dpitch; bp += 64) {
        float2 v = make_float2(0.f, 0.f);
        if (bp < t.ww) {
            const int b = t.ww - 1 - bp;
            const bool m = mrow[b] != 0;
            v = make_float2(m ? wrow[b] : 0.f, m ? 1.f : 0.f);
            if (m) { lo = min(lo, bp); hi = max(hi, bp); ++cnt; }
        }
        out[bp] = v;
    }
    for (int sft = 32; sft > 0; sft >>= 1) {
        lo = min(lo, __shfl_xor(lo, sft, 64));
        hi = max(hi, __shfl_xor(hi, sft, 64));
        cnt += __shfl_xor(cnt, sft, 64);
    }
    if (lane == 0) {
        int4 r = make_int4(0, 0, 0, -1);
        if (hi >= 0) {
            const int g0 = lo >> 2;
            r = make_int4(g0, (hi >> 2) - g0 + 1, lo - 4 * g0, cnt == hi - lo + 1 ? hi - 4 * g0 : -1);
        }
        spans[t.span_off + a] = r;
    }
}

// (W4: built for four waves per SIMD, 128 registers - with a slab of half the LDS two workgroups share a CU, launch_direct;
//  the shared form then keeps 16 dwords in scratch and is still the faster one on windows that small)
template <int NB, int RW, bool SHARE, bool W4 = false>
__global__ void __launch_bounds__(64 * DR2_WAVES, W4 ? 4 : 2)
k_direct2(const float* __restrict__ curv0, size_t curv_stride, Geom g,
          const TemplDev* __restrict__ templ, int first, int n_per, int nb,
          const float2* __restrict__ dwin, const int4* __restrict__ spans,
          const double* __restrict__ sums, const double* __restrict__ xaxis,
          const double* __restrict__ yaxis, float* __restrict__ best_snr,
          float* __restrict__ best_amp, uint32_t* __restrict__ best_id,
          float* __restrict__ map_amp, float* __restrict__ map_snr,
          float near_w, uint8_t* __restrict__ near, unsigned long long* __restrict__ ev_count,
          uint32_t* __restrict__ ev, unsigned long long ev_cap, int lds_floats) {
    // near_w > 0 (the host layer's exact mode): a byte per core cell, set where a template scored within near_w
    // (relative) of the cell's running best, equal scores included - the cells whose argmax is decided inside THIS path's
    // own float32 error - and an event (cell, template scored, holder of the record) per near-tie, as the FFT row pass
    // lists them (k_inv_rows_fast, NEAR): sc_settle_exact scores exactly those pairs in float64
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TXW = 256 * NB, TY = DR2_WAVES * RW;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int zlane = 0;                                   // zero the compiler cannot see through: makes a load a vector load
    asm volatile("" : "+v"(zlane));
    // nb orientations per launch (small DEMs: fewer, longer launches), folded IN ORDER by the one
    // workgroup that owns the patch: orientation b's curvature plane lies b * curv_stride floats
    // further on, its n_per templates follow those of orientation b - 1
    const int i0 = g.cy0 + blockIdx.y * TY, j0 = g.cx0 + blockIdx.x * TXW;
    const int cw = g.cx1 - g.cx0;

    // running best of the lane's cells: row w RW + rr, columns j0 + 256 n + 4 lane + u
    // (kept in memory instead - read and written at every template's fold - the 512 x 16 form spills MORE:
    //  268 B of scratch against 84; the 48 registers are not what it runs out of)
    // (round 4b: only the SNR; the amplitude and the id of a cell are stored when a template WINS it - a win
    //  needs no read of them, and 32 registers fewer are live across the template loops)
    float b_snr[RW][NB][4];
    unsigned dirty = 0;
    auto cell = [&](int rr, int n, int u, int& gi, int& gj) {
        gi = i0 + w * RW + rr;
        gj = j0 + 256 * n + 4 * lane + u;
        return gi < g.cy1 && gj < g.cx1;
    };
#pragma unroll
    for (int rr = 0; rr < RW; ++rr)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int gi, gj;
                const bool in = cell(rr, n, u, gi, gj);
                b_snr[rr][n][u] = (in && !map_amp) ? best_snr[(size_t)(gi - g.cy0) * cw + (gj - g.cx0)] : 0.f;
            }

    const int n_templ = n_per * nb;
    for (int it = 0; it < n_templ; ++it) {
        const TemplDev* tp = templ + first + it;
        const float* curv = curv0 + (size_t)(it / n_per) * curv_stride;
        const int wh = tp->wh, P = tp->dpitch, pmin = tp->pmin, qmax = tp->qmax, span_off = tp->span_off;
        const float2* dw = dwin + tp->dwin_off;
        float xc[RW][NB][4], t3[RW][NB][4];
#pragma unroll
        for (int rr = 0; rr < RW; ++rr)
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int u = 0; u < 4; ++u) xc[rr][n][u] = t3[rr][n][u] = 0.f;
        // slab: curvature columns gj_left .. gj_left + lwp - 1 (output column x, tap b' reads
        // slab column x + b'), rows for template rows [a0, a0 + na): output row r, template
        // row a reads slab row r + (na - 1 - a)
        const int lwp = TXW + P;                                  // a multiple of 4
        int na_max = lds_floats / lwp - (TY - 1);            // (lds_floats: the slab this launch was given, launch_direct)
        na_max = max(1, min(na_max, wh));
        const int gj_left = j0 - qmax + g.ox;
        for (int a0 = 0; a0 < wh; a0 += na_max) {
            const int na = min(na_max, wh - a0);
            const int rows = TY + na - 1;
            const int gi_top = i0 - (pmin + a0 + na - 1) + g.oy;
            __syncthreads();
            for (int r = w; r < rows; r += DR2_WAVES) {            // a wave per slab row, 64 cells at a time
                const int gi = gi_top + r;
                int li;
                bool row_ok = true;
                if (g.wrap) li = wrap_index(gi, g.ny);
                else { li = gi - g.gy0; row_ok = li >= 0 && li < g.ly; }
                const float* src = curv + (size_t)(row_ok ? li : 0) * g.lx;
                float* dst = lds + r * lwp;
                // (periodic DEM at least as wide as the slab: the left edge is reduced once, a cell
                //  wraps at most once - no division per cell)
                const bool once = g.wrap && g.nx >= lwp;
                const int jl = once ? wrap_index(gj_left, g.nx) : 0;
                for (int c = lane; c < lwp; c += 64) {
                    const int gj = gj_left + c;
                    int lj;
                    bool ok = row_ok;
                    if (once) { lj = jl + c; lj = lj >= g.nx ? lj - g.nx : lj; }
                    else if (g.wrap) lj = wrap_index(gj, g.nx);
                    else { lj = gj - g.gx0; ok = ok && lj >= 0 && lj < g.lx; }
                    dst[c] = ok ? src[ok ? lj : 0] : 0.f;
                }
            }
            __syncthreads();
            // (the span of row a + 1 is fetched while row a is worked on: read where it is needed, every
            //  row of a thin window - a handful of groups - began with a scalar-load round trip)
            int4 sp_next = spans[span_off + a0];
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 ca[RW][NB], cb[RW][NB];                            // two chunks of four cells (across rows: see `pre`)
            f4 pwl = {0.f, 0.f, 0.f, 0.f}, pwh = pwl;             // first weights of the next row, fetched ahead
            bool pre = false;                                     // ca / pwl / pwh hold the NEXT row's first chunk and weights
            for (int a = 0; a < na; ++a) {
                const int4 sp = sp_next;                          // (first group, groups, s, e): wave-uniform
                sp_next = spans[span_off + a0 + min(a + 1, na - 1)];
                const int ng = sp.y;
                if (ng == 0) continue;
                const float2* wrow = dw + (size_t)(a0 + a) * P + 4 * sp.x;
                // the lane's slab cells of its rows: row (w RW + rr) + (na - 1 - a), column 4 lane + 4 g (+ 256 n)
                const float* lrow = lds + (w * RW + (na - 1 - a)) * lwp + 4 * lane + 4 * sp.x;
                f4 qa[RW][NB], qb[RW][NB];                        // the chunks' squares (rows without the shared form)
                const bool was_pre = pre;
                pre = false;
                auto chunk = [&](int j, f4 (&v)[RW][NB], f4 (&q)[RW][NB]) {
#pragma unroll
                    for (int rr = 0; rr < RW; ++rr)
#pragma unroll
                        for (int n = 0; n < NB; ++n) {
                            v[rr][n] = *reinterpret_cast<const f4*>(lrow + rr * lwp + 256 * n + 4 * j);
                            q[rr][n] = v[rr][n] * v[rr][n];
                        }
                };
#if !SC_DR_VMEMW
                // the four taps of group gq on the carried cells v0 (the first four) and the new cells v1
                auto group = [&](int gq, const f4 (&v0)[RW][NB], const f4 (&v1)[RW][NB], const f4 (&q0)[RW][NB],
                                 const f4 (&q1)[RW][NB]) {
                    float2 wm[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) wm[k] = wrow[4 * gq + k];
#pragma unroll
                    for (int rr = 0; rr < RW; ++rr)
#pragma unroll
                        for (int n = 0; n < NB; ++n) {
                            const float v[8] = {v0[rr][n].x, v0[rr][n].y, v0[rr][n].z, v0[rr][n].w,
                                                v1[rr][n].x, v1[rr][n].y, v1[rr][n].z, v1[rr][n].w};
                            const float q[8] = {q0[rr][n].x, q0[rr][n].y, q0[rr][n].z, q0[rr][n].w,
                                                q1[rr][n].x, q1[rr][n].y, q1[rr][n].z, q1[rr][n].w};
#pragma unroll
                            for (int k = 0; k < 4; ++k)
#pragma unroll
                                for (int u = 0; u < 4; ++u) {
                                    xc[rr][n][u] = fmaf(wm[k].x, v[u + k], xc[rr][n][u]);
                                    t3[rr][n][u] = fmaf(wm[k].y, q[u + k], t3[rr][n][u]);
                                }
                        }
                };
#endif
                // T3 of a lane's four adjacent outputs over a run WITHOUT holes, taps s .. e: output u sums
                // curv^2 over the cells s + u .. e + u.  The cells s + 3 .. e are common to the four: whole
                // chunks of them go into ONE sum per block (3 adds for the chunk's four squares + 1,
                // instead of 16 weighted FMAs), the 3 - 6 cells at either end are read again after the
                // groups and added to the outputs they belong to - the same addends, no per-cell masks.
                // Rows with holes (Scarp's xr = 0 column at -pi/2, 0, pi/2; generic windows) and runs
                // shorter than SC_DR_SHARE_MIN taps keep the weighted form.  (From 8 taps on the pieces below tile the run: cells
                // s .. s + 2 and e + 1 .. e + 3 per output, [s + 3, 4 jlo) and [4 (jhi + 1), e] - at most three cells each - one
                // by one into the common sum, whole chunks jlo .. jhi in between; below 8 the two short pieces could overlap.)
                const bool shared = SHARE && sp.w >= 0 && sp.w - sp.z >= SC_DR_SHARE_MIN - 1;
                if (!shared) {
#if SC_DR_VMEMW
                  // (the group's four (w, m) pairs through the vector memory path, one group ahead: see the shared form.  While the
                    //  512 x 16 patch spilled, the kernels that carry the shared form kept scalar loads here; without scratch
                    //  - round 4b - both forms on vector loads are 1 % faster on every support)
                  const float* wv = reinterpret_cast<const float*>(wrow) + zlane;
                  auto wload = [&](int gq, f4& lo, f4& hi) {
                      lo = *reinterpret_cast<const f4*>(wv + 8 * gq);
                      hi = *reinterpret_cast<const f4*>(wv + 8 * gq + 4);
                  };
                  auto group_v = [&](const float (&w4)[4], const float (&m4)[4], const f4 (&v0)[RW][NB], const f4 (&v1)[RW][NB],
                                     const f4 (&q0)[RW][NB], const f4 (&q1)[RW][NB]) {
#pragma unroll
                      for (int rr = 0; rr < RW; ++rr)
#pragma unroll
                          for (int n = 0; n < NB; ++n) {
                              const float v[8] = {v0[rr][n].x, v0[rr][n].y, v0[rr][n].z, v0[rr][n].w,
                                                  v1[rr][n].x, v1[rr][n].y, v1[rr][n].z, v1[rr][n].w};
                              const float q[8] = {q0[rr][n].x, q0[rr][n].y, q0[rr][n].z, q0[rr][n].w,
                                                  q1[rr][n].x, q1[rr][n].y, q1[rr][n].z, q1[rr][n].w};
#pragma unroll
                              for (int k = 0; k < 4; ++k)
#pragma unroll
                                  for (int u = 0; u < 4; ++u) {
                                      xc[rr][n][u] = fmaf(w4[k], v[u + k], xc[rr][n][u]);
                                      t3[rr][n][u] = fmaf(m4[k], q[u + k], t3[rr][n][u]);
                                  }
                          }
                  };
                  f4 wAl, wAh, wBl, wBh;
                  wload(0, wAl, wAh);
                  chunk(0, ca, qa);
                  for (int gq = 0; gq < ng; gq += 2) {
                      wload(min(gq + 1, ng - 1), wBl, wBh);
                      chunk(gq + 1, cb, qb);
                      {
                          const float w4[4] = {wAl.x, wAl.z, wAh.x, wAh.z}, m4[4] = {wAl.y, wAl.w, wAh.y, wAh.w};
                          group_v(w4, m4, ca, cb, qa, qb);
                      }
                      if (gq + 1 < ng) {
                          wload(min(gq + 2, ng - 1), wAl, wAh);
                          chunk(gq + 2, ca, qa);
                          const float w4[4] = {wBl.x, wBl.z, wBh.x, wBh.z}, m4[4] = {wBl.y, wBl.w, wBh.y, wBh.w};
                          group_v(w4, m4, cb, ca, qb, qa);
                      }
                  }
#else
                    chunk(0, ca, qa);
                    // two groups per trip: the chunks swap roles (carried / new) instead of being copied
                    // (the copies were 16 v_mov_b64 per 128 FMAs: +9 % on large supports)
                    for (int gq = 0; gq < ng; gq += 2) {
                        chunk(gq + 1, cb, qb);
                        group(gq, ca, cb, qa, qb);
                        if (gq + 1 < ng) {
                            chunk(gq + 2, ca, qa);
                            group(gq + 1, cb, ca, qb, qa);
                        }
                    }
#endif
                } else {
                    const int jlo = (sp.z + 6) >> 2;              // first chunk inside s + 3 .. e
                    const int jhi = ((sp.w + 1) >> 2) - 1;        // last chunk inside it
                    float csum[RW][NB];                       // squares of the cells common to a lane's four outputs, this row
#pragma unroll
                    for (int rr = 0; rr < RW; ++rr)
#pragma unroll
                        for (int n = 0; n < NB; ++n) csum[rr][n] = 0.f;
                    auto cload = [&](const float* lr, int j, f4 (&v)[RW][NB]) {
#pragma unroll
                        for (int rr = 0; rr < RW; ++rr)
#pragma unroll
                            for (int n = 0; n < NB; ++n) v[rr][n] = *reinterpret_cast<const f4*>(lr + rr * lwp + 256 * n + 4 * j);
                    };
                    auto caccum = [&](int j, const f4 (&v)[RW][NB]) {
                        if (j >= jlo && j <= jhi) {
#pragma unroll
                            for (int rr = 0; rr < RW; ++rr)
#pragma unroll
                                for (int n = 0; n < NB; ++n) {
                                    // (four fused multiply-adds: as squares first - two packed multiplies - and a
                                    //  sum tree this was 8 issue slots per chunk next to the 16 of its xcorr FMAs)
                                    const f4 c4 = v[rr][n];
                                    csum[rr][n] = fmaf(c4.w, c4.w, fmaf(c4.z, c4.z, fmaf(c4.y, c4.y, fmaf(c4.x, c4.x, csum[rr][n]))));
                                }
                        }
                    };
                    auto chunk_s = [&](int j, f4 (&v)[RW][NB]) {
                        cload(lrow, j, v);
                        caccum(j, v);
                    };
#if SC_DR_VMEMW
                    // The four weights of a group through the VECTOR memory path (every lane the same address), one
                    // group ahead: as scalar loads they shared the lgkmcnt counter with the slab's LDS reads, scalar
                    // loads return out of order, so every group waited for lgkmcnt(0) - its weights AND every LDS
                    // read in flight - before its first FMA.  vmcnt counts in order and counts nothing else here.
                    const float* wv = reinterpret_cast<const float*>(wrow) + zlane;
                    auto wload = [&](int gq, f4& lo, f4& hi) {
                        lo = *reinterpret_cast<const f4*>(wv + 8 * gq);
                        hi = *reinterpret_cast<const f4*>(wv + 8 * gq + 4);
                    };
                    auto group_x = [&](const float (&w4)[4], const f4 (&v0)[RW][NB], const f4 (&v1)[RW][NB]) {
#pragma unroll
                        for (int rr = 0; rr < RW; ++rr)
#pragma unroll
                            for (int n = 0; n < NB; ++n) {
                                const float v[8] = {v0[rr][n].x, v0[rr][n].y, v0[rr][n].z, v0[rr][n].w,
                                                    v1[rr][n].x, v1[rr][n].y, v1[rr][n].z, v1[rr][n].w};
#pragma unroll
                                for (int k = 0; k < 4; ++k)
#pragma unroll
                                    for (int u = 0; u < 4; ++u) xc[rr][n][u] = fmaf(w4[k], v[u + k], xc[rr][n][u]);
                            }
                    };
                    // (two buffers in vector registers, used from there: moved on to scalar registers first - one
                    //  buffer, v_readfirstlane - every group waited for its own load: 84 TFLOP/s against 90)
                    // The row's first weights and first chunk were fetched at the end of the previous row where that
                    // one had the shared form too (`pre`): a row used to begin by waiting for a vector load and an
                    // LDS read with nothing to do - a fifth of a row of a dozen taps.
                    f4 wAl, wAh, wBl, wBh;
                    if (was_pre) { wAl = pwl; wAh = pwh; }
                    else { wload(0, wAl, wAh); cload(lrow, 0, ca); }
                    caccum(0, ca);
                    for (int gq = 0; gq < ng; gq += 2) {
                        wload(min(gq + 1, ng - 1), wBl, wBh);
                        chunk_s(gq + 1, cb);
                        {
                            const float w4[4] = {wAl.x, wAl.z, wAh.x, wAh.z};
                            group_x(w4, ca, cb);
                        }
                        if (gq + 1 < ng) {
                            wload(min(gq + 2, ng - 1), wAl, wAh);
                            chunk_s(gq + 2, ca);
                            const float w4[4] = {wBl.x, wBl.z, wBh.x, wBh.z};
                            group_x(w4, cb, ca);
                        }
                    }
                    // the next row's first loads, under this row's end cells
                    if (a + 1 < na && sp_next.y > 0 && sp_next.w >= 0 && sp_next.w - sp_next.z >= SC_DR_SHARE_MIN - 1) {
                        const float* wvn = reinterpret_cast<const float*>(dw + (size_t)(a0 + a + 1) * P + 4 * sp_next.x) + zlane;
                        pwl = *reinterpret_cast<const f4*>(wvn);
                        pwh = *reinterpret_cast<const f4*>(wvn + 4);
                        cload(lds + (w * RW + (na - 2 - a)) * lwp + 4 * lane + 4 * sp_next.x, 0, ca);
                        pre = true;
                    }
#else
                    auto group_x = [&](int gq, const f4 (&v0)[RW][NB], const f4 (&v1)[RW][NB]) {
                        float2 wm[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) wm[k] = wrow[4 * gq + k];
#pragma unroll
                        for (int rr = 0; rr < RW; ++rr)
#pragma unroll
                            for (int n = 0; n < NB; ++n) {
                                const float v[8] = {v0[rr][n].x, v0[rr][n].y, v0[rr][n].z, v0[rr][n].w,
                                                    v1[rr][n].x, v1[rr][n].y, v1[rr][n].z, v1[rr][n].w};
#pragma unroll
                                for (int k = 0; k < 4; ++k)
#pragma unroll
                                    for (int u = 0; u < 4; ++u) xc[rr][n][u] = fmaf(wm[k].x, v[u + k], xc[rr][n][u]);
                            }
                    };
                    chunk_s(0, ca);
                    for (int gq = 0; gq < ng; gq += 2) {
                        chunk_s(gq + 1, cb);
                        group_x(gq, ca, cb);
                        if (gq + 1 < ng) {
                            chunk_s(gq + 2, ca);
                            group_x(gq + 1, cb, ca);
                        }
                    }
#endif
                    // The end cells (read again, one at a time, all wave-uniform addresses).  Cells s, s + 1, s + 2
                    // belong to the outputs u <= i and cells e + 1 .. e + 3 to the outputs u >= k; the cells between
                    // s + 3 and the first whole chunk and between the last whole chunk and e are common to the four
                    // like the chunks themselves: one FMA into the common sum each.  (Every end cell as four masked
                    // adds was 60 instructions per block and row - four and a half tap groups' worth on a row of a
                    // dozen; this is 24.)
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int rr = 0; rr < RW; ++rr)
#pragma unroll
                            for (int n = 0; n < NB; ++n) {
                                const float v = lrow[rr * lwp + 256 * n + sp.z + i], q = v * v;
#pragma unroll
                                for (int u = 0; u <= i; ++u) t3[rr][n][u] += q;
                            }
#pragma unroll
                    for (int i = 3; i < 6; ++i) {
                        const int c = sp.z + i;
                        if (c >= 4 * jlo) break;
#pragma unroll
                        for (int rr = 0; rr < RW; ++rr)
#pragma unroll
                            for (int n = 0; n < NB; ++n) {
                                const float v = lrow[rr * lwp + 256 * n + c];
                                csum[rr][n] = fmaf(v, v, csum[rr][n]);
                            }
                    }
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const int c = 4 * (jhi + 1) + i;
                        if (c > sp.w) break;
#pragma unroll
                        for (int rr = 0; rr < RW; ++rr)
#pragma unroll
                            for (int n = 0; n < NB; ++n) {
                                const float v = lrow[rr * lwp + 256 * n + c];
                                csum[rr][n] = fmaf(v, v, csum[rr][n]);
                            }
                    }
#pragma unroll
                    for (int k = 1; k < 4; ++k)
#pragma unroll
                        for (int rr = 0; rr < RW; ++rr)
#pragma unroll
                            for (int n = 0; n < NB; ++n) {
                                const float v = lrow[rr * lwp + 256 * n + sp.w + k], q = v * v;
#pragma unroll
                                for (int u = k; u < 4; ++u) t3[rr][n][u] += q;
                            }
#pragma unroll
                    for (int rr = 0; rr < RW; ++rr)
#pragma unroll
                        for (int n = 0; n < NB; ++n)
#pragma unroll
                            for (int u = 0; u < 4; ++u) t3[rr][n][u] += csum[rr][n];
                }
            }
        }
        const EpiScal es = sc_epi_scalars(sums, first + it);
        const TemplDev t = *tp;
        int zt = 0;
        asm volatile("" : "+v"(zt));
#pragma unroll
        for (int rr = 0; rr < RW; ++rr)
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    int gi, gj;
                    if (!cell(rr, n, u, gi, gj)) continue;
                    float amp, snr;
                    sc_epilogue(xc[rr][n][u], t3[rr][n][u], es, amp, snr);
                    sc_apply_masks(t, g, xaxis, yaxis, gi, gj, amp, snr);
                    if (map_amp) {
                        const size_t o = (size_t)(gi - g.cy0 + zt) * cw + (gj - g.cx0);
                        map_amp[o] = amp;
                        map_snr[o] = snr;
                    } else {
                        float w_amp = 0.f;
                        uint32_t w_id = SC_ID_NONE;
                        if (near_w > 0.f) {                  // (wave-uniform)
                            const float bs = b_snr[rr][n][u];
                            // (EQUAL float32 scores are flagged too: two templates a rounding apart can score the same bits here
                            //  and differ in float64 - exact twins, Scarp at -pi/2 and +pi/2, cost the float64 pass their cells)
                            if (snr > 0.f && fabsf(snr - bs) <= near_w * fmaxf(snr, bs)) {
                                const size_t o = (size_t)(gi - g.cy0 + zt) * cw + (gj - g.cx0);
                                near[o] = (uint8_t)1;
                                if (ev) {                    // (before the fold: the id plane still names the holder - this lane's own store)
                                    const unsigned long long slot = atomicAdd(ev_count, 1ull);
                                    if (slot < ev_cap) {
                                        uint32_t* e = ev + SC_EVENT_WORDS * slot;
                                        e[0] = (uint32_t)o;
                                        e[1] = t.id;
                                        e[2] = best_id[o];
                                        e[3] = __float_as_uint(fmaxf(snr, bs));
                                    }
                                }
                            }
                        }
                        if (sc_fold(b_snr[rr][n][u], w_amp, w_id, snr, amp, t.id)) {
                            // (zt: the cells' offsets are worked out here, at a win - hoisted out of the template loop
                            //  they are 32 registers that live through every row of every template)
                            const size_t o = (size_t)(gi - g.cy0 + zt) * cw + (gj - g.cx0);
                            best_amp[o] = w_amp;
                            best_id[o] = w_id;
                            dirty |= 1u << ((rr * NB + n) * 4 + u);
                        }
                    }
                }
    }
    if (!map_amp && dirty) {
#pragma unroll
        for (int rr = 0; rr < RW; ++rr)
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (!((dirty >> ((rr * NB + n) * 4 + u)) & 1u)) continue;
                    int gi, gj;
                    cell(rr, n, u, gi, gj);
                    const size_t o = (size_t)(gi - g.cy0) * cw + (gj - g.cx0);
                    best_snr[o] = b_snr[rr][n][u];
                }
    }
}

// ---------------------------------------------------------------------------
// compare() on host-provided float64 results (core.py:230-240), written the
// way numexpr evaluates it: boolean * value + boolean * value, snr last.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_compare_fold(double* __restrict__ b_amp, double* __restrict__ b_age,
               double* __restrict__ b_ang, double* __restrict__ b_snr,
               const double* __restrict__ t_amp, const double* __restrict__ t_snr,
               const double* __restrict__ t_age, const double* __restrict__ t_ang,
               double age, double angle, size_t n) {
    // t_age / t_ang: per-cell planes (results of an earlier fold, as match()
    // feeds them to compare(), core.py:288-292) or null for scalar age / angle
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        double bs = b_snr[i], ts = t_snr[i];
        double keep = (bs > ts) ? 1.0 : 0.0, take = (bs < ts) ? 1.0 : 0.0;
        b_amp[i] = __dadd_rn(__dmul_rn(keep, b_amp[i]), __dmul_rn(take, t_amp[i]));
        b_age[i] = __dadd_rn(__dmul_rn(keep, b_age[i]), __dmul_rn(take, t_age ? t_age[i] : age));
        b_ang[i] = __dadd_rn(__dmul_rn(keep, b_ang[i]), __dmul_rn(take, t_ang ? t_ang[i] : angle));
        b_snr[i] = __dadd_rn(__dmul_rn(keep, bs), __dmul_rn(take, ts));
    }
}

// ---------------------------------------------------------------------------
// Nodata fill (DEMGrid._fill_nodata, dem.py:388-414 -> rasterio.fill.fillnodata ->
// GDALFillNodata): four-quadrant inverse-distance interpolation from the nearest valid
// cell of every column within the search distance, as restated in
// the oracle's fill_nodata_pass (parity unpinned: GDAL is not in the image, no GDAL-written
// fixture exists).  float32 work values like GDAL's scanlines, sums in float64, the oracle's
// operation order.
//   k_fill_scan : per column, the nearest valid row at or above / at or below every row
//   k_fill_idw  : per nodata cell, the search over columns x +- step and the weighted mean
//   k_fill_smooth: optional 3x3 means over filled cells
// ---------------------------------------------------------------------------
#define SC_FILL_NONE_UP (-(1 << 30))
#define SC_FILL_NONE_DN (1 << 30)

__global__ void __launch_bounds__(256)
k_fill_scan(const double* __restrict__ z, int ny, int nx, int* __restrict__ up, int* __restrict__ dn) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= nx) return;
    int last = SC_FILL_NONE_UP;
    for (int y = 0; y < ny; ++y) {
        const double v = z[(size_t)y * nx + x];
        if (v == v) last = y;
        up[(size_t)y * nx + x] = last;
    }
    last = SC_FILL_NONE_DN;
    for (int y = ny - 1; y >= 0; --y) {
        const double v = z[(size_t)y * nx + x];
        if (v == v) last = y;
        dn[(size_t)y * nx + x] = last;
    }
}

// One nodata cell of GDALFillNodata's second pass (alg/rasterfill.cpp as the oracle's
// fill_nodata_pass restates it, statement by statement): float32 work values, quadrants TL, BL,
// TR, BR, left quadrants at every step and right ones from step 1, columns clamped at the raster
// edge, QUAD_CHECK on squared distances, the search limit refreshed every four steps, weights
// 1 / distance accumulated in quadrant order.  Valid cells pass through float32 as well (GDAL
// writes the whole float32 scanline back).
__global__ void __launch_bounds__(256)
k_fill_idw(const double* __restrict__ z, int ny, int nx, const int* __restrict__ up,
           const int* __restrict__ dn, double maxd, int R, double* __restrict__ out,
           unsigned long long* __restrict__ remaining) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= nx) return;
    const size_t o = (size_t)y * nx + x;
    const double v0 = z[o];
    if (v0 == v0) { out[o] = (double)(float)v0; return; }
    const double far = __dadd_rn(maxd, 1.0);
    double qd[4] = {far, far, far, far};
    float qv[4] = {0.f, 0.f, 0.f, 0.f};
    auto check = [&](int q, int tx, int ty) {
        if (ty < 0 || ty >= ny) return;
        const double ddx = (double)tx - (double)x, ddy = (double)ty - (double)y;
        const double dsq = __dadd_rn(__dmul_rn(ddx, ddx), __dmul_rn(ddy, ddy));
        if (dsq < __dmul_rn(qd[q], qd[q])) {
            qd[q] = __dsqrt_rn(dsq);
            qv[q] = (float)z[(size_t)ty * nx + tx];
        }
    };
    int this_max = R;
    for (int step = 0; step <= this_max; ++step) {
        const int lx = max(0, x - step), rx = min(nx - 1, x + step);
        check(0, lx, up[(size_t)y * nx + lx]);
        check(1, lx, dn[(size_t)y * nx + lx]);
        if (step == 0) continue;
        check(2, rx, up[(size_t)y * nx + rx]);
        check(3, rx, dn[(size_t)y * nx + rx]);
        if ((step & 3) == 0) this_max = (int)floor(fmax(fmax(qd[0], qd[1]), fmax(qd[2], qd[3])));
    }
    double ws = 0.0, vs = 0.0;
    bool have = false;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (qd[q] <= maxd) {
            const double w = __ddiv_rn(1.0, qd[q]);
            have = true;
            ws = __dadd_rn(ws, w);
            vs = __dadd_rn(vs, __dmul_rn((double)qv[q], w));
        }
    if (have) {
        out[o] = (double)(float)__ddiv_rn(vs, ws);
    } else {
        out[o] = v0;                                  // stays nodata
        atomicAdd(remaining, 1ull);
    }
}

// mean of the non-nodata cells of the 3x3 neighbourhood, filled cells only (row-major sum
// like numpy's nanmean over the window)
__global__ void __launch_bounds__(256)
k_fill_smooth(const double* __restrict__ src, const double* __restrict__ orig, int ny, int nx,
              double* __restrict__ dst) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= nx) return;
    const size_t o = (size_t)y * nx + x;
    const double v = src[o], z0 = orig[o];
    if (z0 == z0 || v != v) { dst[o] = v; return; }   // an original cell, or still nodata
    double sum = 0.0;
    int cnt = 0;
    for (int j = max(y - 1, 0); j <= min(y + 1, ny - 1); ++j)
        for (int i = max(x - 1, 0); i <= min(x + 1, nx - 1); ++i) {
            const double w = src[(size_t)j * nx + i];
            if (w == w) { sum = __dadd_rn(sum, w); ++cnt; }
        }
    dst[o] = __ddiv_rn(sum, (double)cnt);
}

int launch_fill_nodata(sc_ctx* ctx, double* zdev, double* tmp, int* up, int* dn, int ny, int nx,
                       double maxd, int smoothing, unsigned long long* remaining_dev) {
    SC_HIP(ctx, hipMemsetAsync(remaining_dev, 0, sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(k_fill_scan, dim3((nx + 255) / 256), dim3(256), 0, ctx->stream,
                       (const double*)zdev, ny, nx, up, dn);
    int R = (int)floor(maxd);
    if (R < 0) R = 0;
    if (R > nx) R = nx;
    dim3 grid((nx + 255) / 256, ny);
    hipLaunchKernelGGL(k_fill_idw, grid, dim3(256), 0, ctx->stream, (const double*)zdev, ny, nx,
                       (const int*)up, (const int*)dn, maxd, R, tmp, remaining_dev);
    // result in tmp; smoothing ping-pongs tmp <-> a second buffer carved from `up`/`dn`? no: the
    // original z (zdev) is still needed as the "filled cells only" mask, so smooth into zdev last
    double* cur = tmp;
    if (smoothing > 0) {
        // needs a third plane: reuse the up/dn storage (2 x int32 per cell = one float64 plane)
        double* alt = reinterpret_cast<double*>(up);
        for (int k = 0; k < smoothing; ++k) {
            hipLaunchKernelGGL(k_fill_smooth, grid, dim3(256), 0, ctx->stream, (const double*)cur,
                               (const double*)zdev, ny, nx, alt);
            std::swap(cur, alt);
        }
    }
    SC_HIP(ctx, hipMemcpyAsync(zdev, cur, sizeof(double) * (size_t)ny * nx, hipMemcpyDeviceToDevice, ctx->stream));
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
int launch_compare_fold(sc_ctx* ctx, double age, double angle, bool planes) {
    size_t n = ctx->cmp_n;
    size_t blocks = std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_compare_fold, dim3((unsigned)std::max<size_t>(blocks, 1)), dim3(256), 0,
                       ctx->stream, (double*)ctx->cmp[0].p, (double*)ctx->cmp[1].p,
                       (double*)ctx->cmp[2].p, (double*)ctx->cmp[3].p,
                       (const double*)ctx->cmp_in[0].p, (const double*)ctx->cmp_in[1].p,
                       planes ? (const double*)ctx->cmp_in[2].p : nullptr,
                       planes ? (const double*)ctx->cmp_in[3].p : nullptr, age, angle, n);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

int launch_curv_planes(sc_ctx* ctx) {
    const Geom& g = ctx->g;
    dim3 grid((g.lx + 255) / 256, g.ly);
    sc_prof_begin(ctx, SC_K_CURV);
    hipLaunchKernelGGL(k_curv_planes<float>, grid, dim3(256), 0, ctx->stream,
                       ctx->z_dev, g, ctx->dx, ctx->dy, (float*)ctx->A.p,
                       (float*)ctx->B.p, (float*)ctx->C.p);
    sc_prof_end(ctx);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

// the same for the nb orientations of a batched launch sequence: plane blockIdx.y with its own
// coefficients (a 512^2 search spent 40 % of its time in 905 five-microsecond launches of k_curv_alpha)
struct CurvCoefs { float c[SC_MAX_ORIENT][3]; };
__global__ void __launch_bounds__(256)
k_curv_alpha_batch(const float* __restrict__ A, const float* __restrict__ B,
                   const float* __restrict__ C, CurvCoefs k, float* __restrict__ out, size_t n) {
    const float cc = k.c[blockIdx.y][0], sc2 = k.c[blockIdx.y][1], ss = k.c[blockIdx.y][2];
    out += (size_t)blockIdx.y * n;
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (; i < n; i += stride) {
        if (i + 3 < n) {
            float4 a = *reinterpret_cast<const float4*>(A + i);
            float4 b = *reinterpret_cast<const float4*>(B + i);
            float4 c = *reinterpret_cast<const float4*>(C + i);
            float4 o;
            o.x = cc * a.x - sc2 * b.x + ss * c.x;
            o.y = cc * a.y - sc2 * b.y + ss * c.y;
            o.z = cc * a.z - sc2 * b.z + ss * c.z;
            o.w = cc * a.w - sc2 * b.w + ss * c.w;
            *reinterpret_cast<float4*>(out + i) = o;
        } else {
            for (size_t j = i; j < n; ++j)
                out[j] = cc * A[j] - sc2 * B[j] + ss * C[j];
        }
    }
}

int launch_curv_alpha_batch(sc_ctx* ctx, const float (*coef)[3], int nb) {
    if (nb < 1 || nb > SC_MAX_ORIENT) return sc_fail(ctx, SC_ERR_INVALID, "curvature batch of %d planes", nb);
    size_t n = (size_t)ctx->g.ly * ctx->g.lx;
    // plane p starts at element p * n: float4 accesses need n to be a multiple of 4
    if (nb > 1 && (n & 3)) {
        for (int b = 0; b < nb; ++b) {
            int rc = launch_curv_alpha(ctx, coef[b][0], coef[b][1], coef[b][2], b);
            if (rc) return rc;
        }
        return SC_OK;
    }
    CurvCoefs k;
    for (int b = 0; b < nb; ++b)
        for (int j = 0; j < 3; ++j) k.c[b][j] = coef[b][j];
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    sc_prof_begin(ctx, SC_K_CURV);
    hipLaunchKernelGGL(k_curv_alpha_batch, dim3((unsigned)blocks, nb), dim3(256), 0, ctx->stream,
                       (const float*)ctx->A.p, (const float*)ctx->B.p, (const float*)ctx->C.p, k,
                       (float*)ctx->curv.p, n);
    sc_prof_end(ctx);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

int launch_curv_alpha(sc_ctx* ctx, float cc, float sc2, float ss, int plane) {
    // plane: which of the context's curvature planes receives it (batched orientations)
    size_t n = (size_t)ctx->g.ly * ctx->g.lx;
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    sc_prof_begin(ctx, SC_K_CURV);
    hipLaunchKernelGGL(k_curv_alpha, dim3((unsigned)blocks), dim3(256), 0,
                       ctx->stream, (const float*)ctx->A.p,
                       (const float*)ctx->B.p, (const float*)ctx->C.p, cc, sc2,
                       ss, (float*)ctx->curv.p + (size_t)plane * n, n);
    sc_prof_end(ctx);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

int launch_windows(sc_ctx* ctx, int first, int n, int wh_max, int ww_max) {
    if (n <= 0 || wh_max <= 0 || ww_max <= 0) return SC_OK;
    dim3 grid((ww_max + 63) / 64, (wh_max + SC_WIN_ROWS - 1) / SC_WIN_ROWS, n);
    sc_prof_begin(ctx, SC_K_WINDOWS);
    hipLaunchKernelGGL(k_windows, grid, dim3(64), 0, ctx->stream,
                       (const TemplDev*)ctx->templ.p, first,
                       (const double*)ctx->xaxis.p, (const double*)ctx->yaxis.p,
                       ctx->g.ny, ctx->g.nx, (float*)ctx->win_w.p,
                       (uint8_t*)ctx->win_m.p, (double*)ctx->sums.p,
                       (double*)ctx->wl1.p);
    sc_prof_end(ctx);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

// widest template window the real-space kernels can stage: the slab must hold at least one
// template row for every output row of the patch (k_direct2: 256 + the padded window width
// cells, 16 rows; the box kernel, variant 10: DR_TX + ww - 1 cells, odd pitch, DR_TY rows)
bool direct_window_fits(int ww) {
    const long long p2 = 256 + ((ww + 3) & ~3);
    return p2 * 16 <= DR2_LDS_FLOATS && (long long)(((DR_TX + ww - 1) | 1)) * DR_TY <= DR_LDS_FLOATS;
}

static int launch_direct_box(sc_ctx* ctx, int first, int n, bool to_maps) {
    const Geom& g = ctx->g;
    int ch = g.cy1 - g.cy0, cw = g.cx1 - g.cx0;
    dim3 grid((cw + DR_TX - 1) / DR_TX, (ch + DR_TY - 1) / DR_TY);
    size_t lds = (size_t)DR_LDS_FLOATS * sizeof(float) + 4096;
    int rc = sc_lds_attr(ctx, (const void*)k_direct, lds);
    if (rc) return rc;
    sc_prof_begin(ctx, SC_K_DIRECT);
    hipLaunchKernelGGL(k_direct, grid, dim3(DR_THREADS), lds, ctx->stream,
                       (const float*)ctx->curv.p, g,
                       (const TemplDev*)ctx->templ.p, first, n,
                       (const float*)ctx->win_w.p, (const uint8_t*)ctx->win_m.p,
                       (const double*)ctx->sums.p, (const double*)ctx->xaxis.p,
                       (const double*)ctx->yaxis.p, (float*)ctx->best_snr.p,
                       (float*)ctx->best_amp.p, (uint32_t*)ctx->best_id.p,
                       to_maps ? (float*)ctx->map_amp.p : nullptr,
                       to_maps ? (float*)ctx->map_snr.p : nullptr);
    sc_prof_end(ctx);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

// templates [first, first + nb * n): nb orientations of n templates, orientation b on curvature
// plane b (nb = 1 unless the launch sequence batches orientations); wh_max: tallest window
int launch_direct(sc_ctx* ctx, int first, int n, bool to_maps, int nb, int wh_max, int ww_max, bool long_runs) {
    if (ctx->variant == 10) {
        if (nb != 1) return sc_fail(ctx, SC_ERR_INVALID, "the box kernel takes one orientation per launch");
        return launch_direct_box(ctx, first, n, to_maps);
    }
    const Geom& g = ctx->g;
    const int ch = g.cy1 - g.cy0, cw = g.cx1 - g.cx0;
    sc_prof_begin(ctx, SC_K_WINDOWS);
    hipLaunchKernelGGL(k_direct_prep, dim3((wh_max + 3) / 4, nb * n), dim3(256), 0, ctx->stream,
                       (const TemplDev*)ctx->templ.p, first, (const float*)ctx->win_w.p,
                       (const uint8_t*)ctx->win_m.p, (float2*)ctx->dwin.p, (int4*)ctx->spans.p);
    sc_prof_end(ctx);
    SC_HIP(ctx, hipGetLastError());
    // near-tie flags of the exact mode (option "near_window"): the byte plane of sc_get_near_ties, and the event list
    const bool near_on = ctx->near_w > 0.f && !to_maps;
    unsigned long long* ev_count = nullptr;
    uint32_t* ev = nullptr;
    unsigned long long ev_cap = 0;
    if (near_on) {
        int rc = sc_near_buffers(ctx, &ev_count, &ev, &ev_cap);
        if (rc) return rc;
    }
    // patch: 512 x 16 cells where that still gives every CU a workgroup, else 256 x 16, else 256 x 8
    // The slab: all of the LDS (one workgroup per CU) unless the patch selection below gives the launch a smaller one.  Its
    // size decides only how many template rows are staged at a time, not the order of any sum.
    size_t lds = (size_t)DR2_LDS_FLOATS * sizeof(float);
    const size_t plane = (size_t)g.ly * g.lx;
    auto wgs = [&](int txw, int ty) { return (long long)((cw + txw - 1) / txw) * ((ch + ty - 1) / ty); };
#define DR2_LAUNCH(NBV, RWV, SHV, ...)                                                                     \
    {                                                                                              \
        int rc = sc_lds_attr(ctx, (const void*)k_direct2<NBV, RWV, SHV __VA_OPT__(,) __VA_ARGS__>, lds);                   \
        if (rc) return rc;                                                                         \
        dim3 grid((cw + 256 * NBV - 1) / (256 * NBV), (ch + 8 * RWV - 1) / (8 * RWV));            \
        sc_prof_begin(ctx, SC_K_DIRECT);                                                           \
        hipLaunchKernelGGL((k_direct2<NBV, RWV, SHV __VA_OPT__(,) __VA_ARGS__>), grid, dim3(64 * DR2_WAVES), lds, ctx->stream, \
                           (const float*)ctx->curv.p, plane, g, (const TemplDev*)ctx->templ.p, first, n, nb, \
                           (const float2*)ctx->dwin.p, (const int4*)ctx->spans.p,                  \
                           (const double*)ctx->sums.p, (const double*)ctx->xaxis.p,                \
                           (const double*)ctx->yaxis.p, (float*)ctx->best_snr.p,                   \
                           (float*)ctx->best_amp.p, (uint32_t*)ctx->best_id.p,                     \
                           to_maps ? (float*)ctx->map_amp.p : nullptr,                             \
                           to_maps ? (float*)ctx->map_snr.p : nullptr,                             \
                           near_on ? ctx->near_w : 0.f, near_on ? (uint8_t*)ctx->near.p : nullptr, ev_count, ev, ev_cap, \
                           (int)(lds / sizeof(float)));                                            \
        sc_prof_end(ctx);                                                                          \
    }
    // (variant 11: T3 in the weighted form on every row; long_runs: some template of the launch has
    //  rows of 16 taps or more - the shared form's kernel carries more registers, thin windows are
    //  10 % faster on the plain one)
    const bool share = ctx->variant != 11 && long_runs;
    // (the 512-wide patch needs a slab of 512 + the padded window width cells x 16 rows)
    const bool fits512 = (long long)(512 + ((ww_max + 3) & ~3)) * 16 <= DR2_LDS_FLOATS;
    // (round 6) windows whose 256 x 16 slab fits half the LDS in one piece - up to about a thousand taps - take that patch
    // with a slab of their own size: 128 registers and half the LDS are two workgroups per CU, four waves per SIMD, and the
    // kernel is bound by instruction issue there (-4 .. -25 % against the 512 x 16 patch at two waves per SIMD,
    // profiles/r06_crossover.txt; option "variant" 19: the whole LDS as before)
    const long long need256 = (long long)(256 + ((ww_max + 3) & ~3)) * (16 + wh_max - 1);
    const bool two_wg = need256 <= DR2_LDS_FLOATS / 2 - 256 && wgs(256, 16) >= 256 && ctx->variant != 19;
    if (two_wg) {
        lds = (size_t)need256 * sizeof(float);
        if (share) DR2_LAUNCH(1, 2, true, true) else DR2_LAUNCH(1, 2, false, true)
    } else if (fits512 && wgs(512, 16) >= 512 && ctx->variant != 16) { if (share) DR2_LAUNCH(2, 2, true) else DR2_LAUNCH(2, 2, false) }
    else if (wgs(256, 16) >= 256) { if (share) DR2_LAUNCH(1, 2, true) else DR2_LAUNCH(1, 2, false) }
    else { if (share) DR2_LAUNCH(1, 1, true) else DR2_LAUNCH(1, 1, false) }
#undef DR2_LAUNCH
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}
