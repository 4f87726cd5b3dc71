// C ABI of libscarplet_hip.so (include/scarplet_hip.h): context, DEM hand-over,
// batching of templates into orientation runs, and measurement hooks.
#include "sc_internal.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include <algorithm>

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------
int sc_fail(sc_ctx* ctx, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

int sc_ensure(sc_ctx* ctx, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap && b.p) return SC_OK;
    if (b.p) {
        SC_HIP(ctx, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    if (bytes == 0) bytes = 16;
    SC_HIP(ctx, hipMalloc(&b.p, bytes));
    b.cap = bytes;
    return SC_OK;
}

static void buf_free(DevBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

size_t sc_total_bytes(sc_ctx* c) {
    DevBuf* arr[] = {&c->z, &c->xaxis, &c->yaxis, &c->A, &c->B, &c->C, &c->curv,
                     &c->best_snr, &c->best_amp, &c->best_id, &c->map_amp,
                     &c->map_snr, &c->templ, &c->sums, &c->wl1, &c->norms, &c->norm_part, &c->win_w, &c->win_m,
                     &c->tw_y, &c->tw_x, &c->blk, &c->uc, &c->uc2, &c->vh, &c->wh, &c->mh,
                     &c->yw, &c->ym, &c->tiles, &c->halo_z, &c->halo_stage, &c->res, &c->sib_buf, &c->dwin, &c->spans, &c->res_stats, &c->digest, &c->split_s, &c->split_a, &c->split_i,
                     &c->near, &c->near_ev, &c->score, &c->score_w, &c->score_abc, &c->st_slot, &c->st_work, &c->st_pairs, &c->st_patch, &c->st_spans, &c->snap, &c->xch, &c->xch_cnt};
    size_t s = 0;
    for (DevBuf* b : arr) s += b->cap;
    for (auto& w : c->windows) s += (size_t)w.h * w.wd * 5;
    return s;
}

int sc_lds_attr(sc_ctx* ctx, const void* kernel, size_t bytes) {
    auto it = ctx->lds_attr.find(kernel);
    if (it != ctx->lds_attr.end() && it->second >= bytes) return SC_OK;
    SC_HIP(ctx, hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)bytes));
    ctx->lds_attr[kernel] = bytes;
    return SC_OK;
}

// ---------------------------------------------------------------------------
// profiling: HIP events on the context's stream around sampled launches
// ---------------------------------------------------------------------------
// A bracket may hold several launches of the same kernel (sc_prof_end's n):
// launches are counted one by one, and a sampled bracket contributes its
// elapsed time divided over its n launches.
void sc_prof_begin(sc_ctx* ctx, int kernel) {
    ctx->prof_cur = -1;
    ctx->prof_kernel = kernel;
    const long long b = ctx->k_brackets[kernel]++;
    if (!ctx->prof) return;
    // one bracket in `prof`, chosen by a hash of the counter: the launches of a
    // kernel come in periodic patterns (curvature / template forward passes, first
    // / second chunk of an orientation) that a fixed stride would alias with
    uint32_t h = (uint32_t)b * 2654435761u;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    if (ctx->prof > 1 && h % (uint32_t)ctx->prof) return;
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (!ctx->ev_pool.empty()) {
        ev = ctx->ev_pool.back();
        ctx->ev_pool.pop_back();
    } else {
        if (hipEventCreate(&ev.first) != hipSuccess) return;
        if (hipEventCreate(&ev.second) != hipSuccess) return;
    }
    (void)hipEventRecord(ev.first, ctx->stream);
    ctx->prof_cur = kernel;
    ctx->prof_ev0 = ev.first;
    ctx->prof_ev1 = ev.second;
}

void sc_prof_end(sc_ctx* ctx, int n) {
    ctx->k_launches[ctx->prof_kernel] += n;
    if (ctx->prof_cur < 0) return;
    (void)hipEventRecord(ctx->prof_ev1, ctx->stream);
    ctx->pending.push_back({ctx->prof_cur, {ctx->prof_ev0, ctx->prof_ev1}});
    ctx->pending_n.push_back(n);
    ctx->prof_cur = -1;
}

void sc_prof_collect(sc_ctx* ctx) {
    for (size_t i = 0; i < ctx->pending.size(); ++i) {
        auto& p = ctx->pending[i];
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.second.first, p.second.second) == hipSuccess) {
            ctx->k_ms[p.first] += ms;
            ctx->k_sampled[p.first] += ctx->pending_n[i];
        }
        ctx->ev_pool.push_back(p.second);
    }
    ctx->pending.clear();
    ctx->pending_n.clear();
}

// ---------------------------------------------------------------------------
// lifetime
// ---------------------------------------------------------------------------
extern "C" int sc_abi_version(void) { return SC_ABI_VERSION; }

extern "C" int sc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int sc_create(int device, sc_ctx** out) {
    if (!out) return SC_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SC_ERR_HIP;
    if (device < 0 || device >= n) return SC_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return SC_ERR_HIP;
    sc_ctx* c = new sc_ctx();
    c->device = device;
#ifdef SC_ABLATE
    if (const char* d = getenv("SC_DBG")) c->dbg = atoi(d);
#endif
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return SC_ERR_HIP;
    }
    *out = c;
    return SC_OK;
}

extern "C" int sc_set_option(sc_ctx* ctx, const char* name, double value) {
    if (!ctx || !name) return SC_ERR_INVALID;
    if (!strcmp(name, "kappa")) {
        if (!(value >= 0.0)) return sc_fail(ctx, SC_ERR_INVALID, "kappa must be >= 0");
        ctx->kappa = (float)value;
    } else if (!strcmp(name, "variant")) {
        ctx->variant = (int)value;
#ifdef SC_ABLATE
    } else if (!strcmp(name, "dbg")) {
        ctx->dbg = (int)value;
#endif
    } else if (!strcmp(name, "batch")) {
        ctx->batch_off = value == 0.0;
    } else if (!strcmp(name, "batch_fill")) {
        if (!(value >= 0.0)) return sc_fail(ctx, SC_ERR_INVALID, "batch_fill must be >= 0");
        ctx->batch_fill = (int)value;
    } else if (!strcmp(name, "i1_pairs")) {
        if (!(value >= 1.0 && value <= 64.0)) return sc_fail(ctx, SC_ERR_INVALID, "i1_pairs must be 1 .. 64");
        ctx->i1_pairs = (int)value;
    } else if (!strcmp(name, "near_window")) {
        if (!(value >= 0.0 && value < 1.0)) return sc_fail(ctx, SC_ERR_INVALID, "near_window must be in [0, 1)");
        ctx->near_w = (float)value;
    } else if (!strcmp(name, "batch_templ")) {
        if (!(value >= 0.0 && value <= (double)SC_MAX_BATCH)) return sc_fail(ctx, SC_ERR_INVALID, "batch_templ must be 0 .. %d", SC_MAX_BATCH);
        ctx->batch_templ = (int)value;
    } else if (!strcmp(name, "split_i1")) {
        if (!(value >= 0.0 && value <= 64.0)) return sc_fail(ctx, SC_ERR_INVALID, "split_i1 must be 0 .. 64");
        ctx->split_i1 = (int)value;
    } else if (!strcmp(name, "split_fill")) {
        if (!(value >= 0.0)) return sc_fail(ctx, SC_ERR_INVALID, "split_fill must be >= 0");
        ctx->split_fill = (long long)value;
    } else if (!strcmp(name, "sib")) {
        ctx->sib = (int)value;
    } else if (!strcmp(name, "spectra_mb")) {
        if (!(value >= 0.0)) return sc_fail(ctx, SC_ERR_INVALID, "spectra_mb must be >= 0");
        ctx->spec_mb = value;
        fft_spectra_forget(ctx);
    } else if (!strcmp(name, "y_gb")) {
        if (!(value >= 0.0)) return sc_fail(ctx, SC_ERR_INVALID, "y_gb must be >= 0");
        ctx->y_gb = value;
    } else {
        return sc_fail(ctx, SC_ERR_INVALID, "unknown option '%s'", name);
    }
    return SC_OK;
}

extern "C" int sc_clear_windows(sc_ctx* ctx) {
    if (!ctx) return SC_ERR_INVALID;
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& w : ctx->windows) {
        if (w.w) (void)hipFree(w.w);
        if (w.m) (void)hipFree(w.m);
        if (w.mask_lim) (void)hipFree(w.mask_lim);
        if (w.mask_err) (void)hipFree(w.mask_err);
    }
    ctx->windows.clear();
    return SC_OK;
}

extern "C" void sc_destroy(sc_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    sc_comm_destroy(c);
    sc_prof_collect(c);
    for (auto& e : c->ev_pool) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    sc_clear_windows(c);
    DevBuf* arr[] = {&c->z, &c->xaxis, &c->yaxis, &c->A, &c->B, &c->C, &c->curv,
                     &c->best_snr, &c->best_amp, &c->best_id, &c->map_amp,
                     &c->map_snr, &c->templ, &c->sums, &c->wl1, &c->norms, &c->norm_part, &c->win_w, &c->win_m,
                     &c->tw_y, &c->tw_x, &c->blk, &c->uc, &c->uc2, &c->vh, &c->wh, &c->mh,
                     &c->yw, &c->ym, &c->tiles, &c->halo_z, &c->halo_stage, &c->res, &c->sib_buf, &c->dwin, &c->spans, &c->res_stats, &c->digest, &c->split_s, &c->split_a, &c->split_i,
                     &c->near, &c->near_ev, &c->score, &c->score_w, &c->score_abc, &c->st_slot, &c->st_work, &c->st_pairs, &c->st_patch, &c->st_spans, &c->snap, &c->xch, &c->xch_cnt};
    for (DevBuf* b : arr) buf_free(*b);
    for (int k = 0; k < 4; ++k) buf_free(c->cmp[k]);
    for (int k = 0; k < 4; ++k) buf_free(c->cmp_in[k]);
    for (int k = 0; k < 4; ++k) buf_free(c->fill[k]);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" const char* sc_last_error(sc_ctx* ctx) {
    return ctx ? ctx->err.c_str() : "null context";
}

extern "C" size_t sc_device_bytes(sc_ctx* ctx) { return ctx ? sc_total_bytes(ctx) : 0; }

// ---------------------------------------------------------------------------
// DEM hand-over
// ---------------------------------------------------------------------------
static int set_dem_body(sc_ctx* ctx, int ly, int lx, int gy0, int gx0, int ny,
                        int nx, int cy0, int cy1, int cx0, int cx1, double dx,
                        double dy, int wrap, const double* xaxis,
                        const double* yaxis);

// A hand-over that fails part-way (an allocation, the digest, the curvature planes, the record's reset) leaves
// NO DEM behind: the elevations and the geometry of the old one are overwritten by then, and a context that kept
// have_dem and the new block's fingerprint would answer the same block handed over again with "unchanged" and
// search on curvature planes and kept spectra of a half-installed state.
static int set_dem_common(sc_ctx* ctx, int ly, int lx, int gy0, int gx0, int ny,
                          int nx, int cy0, int cy1, int cx0, int cx1, double dx,
                          double dy, int wrap, const double* xaxis,
                          const double* yaxis) {
    const int rc = set_dem_body(ctx, ly, lx, gy0, gx0, ny, nx, cy0, cy1, cx0, cx1, dx, dy, wrap, xaxis, yaxis);
    if (rc != SC_OK) {
        ctx->have_dem = false;
        ctx->dem_unchanged = false;
        ctx->dem_hash[0] = ctx->dem_hash[1] = 0;
        memset(ctx->dem_sig, 0, sizeof(ctx->dem_sig));
        fft_spectra_forget(ctx);
    }
    return rc;
}

static int set_dem_body(sc_ctx* ctx, int ly, int lx, int gy0, int gx0, int ny,
                        int nx, int cy0, int cy1, int cx0, int cx1, double dx,
                        double dy, int wrap, const double* xaxis,
                        const double* yaxis) {
    if (ly < 3 || lx < 3 || ny < 3 || nx < 3 || !xaxis || !yaxis)
        return sc_fail(ctx, SC_ERR_INVALID, "sc_set_dem: bad sizes or null axes");
    if (cy0 < 0 || cy1 > ny || cx0 < 0 || cx1 > nx || cy0 >= cy1 || cx0 >= cx1)
        return sc_fail(ctx, SC_ERR_INVALID, "sc_set_dem: bad core rectangle");
    if (wrap && (ly != ny || lx != nx || gy0 != 0 || gx0 != 0))
        return sc_fail(ctx, SC_ERR_INVALID, "sc_set_dem: wrap needs the whole DEM");
    if (!wrap && (cy0 < gy0 || cy1 > gy0 + ly || cx0 < gx0 || cx1 > gx0 + lx))
        return sc_fail(ctx, SC_ERR_INVALID, "sc_set_dem: core outside the block");
    if (dx == 0.0 || dy == 0.0)
        return sc_fail(ctx, SC_ERR_INVALID, "sc_set_dem: zero cell size");
    Geom& g = ctx->g;
    g.ly = ly; g.lx = lx; g.gy0 = gy0; g.gx0 = gx0; g.ny = ny; g.nx = nx;
    g.cy0 = cy0; g.cy1 = cy1; g.cx0 = cx0; g.cx1 = cx1; g.wrap = wrap ? 1 : 0;
    g.oy = ny % 2; g.ox = nx % 2;
    ctx->dx = dx; ctx->dy = dy;
    size_t nl = (size_t)ly * lx, nc = (size_t)(cy1 - cy0) * (cx1 - cx0);
    int rc;
    if ((rc = sc_ensure(ctx, ctx->xaxis, sizeof(double) * nx))) return rc;
    if ((rc = sc_ensure(ctx, ctx->yaxis, sizeof(double) * ny))) return rc;
    if ((rc = sc_ensure(ctx, ctx->A, sizeof(float) * nl))) return rc;
    if ((rc = sc_ensure(ctx, ctx->B, sizeof(float) * nl))) return rc;
    if ((rc = sc_ensure(ctx, ctx->C, sizeof(float) * nl))) return rc;
    if ((rc = sc_ensure(ctx, ctx->curv, sizeof(float) * nl))) return rc;
    if ((rc = sc_ensure(ctx, ctx->best_snr, sizeof(float) * nc))) return rc;
    if ((rc = sc_ensure(ctx, ctx->best_amp, sizeof(float) * nc))) return rc;
    if ((rc = sc_ensure(ctx, ctx->best_id, sizeof(uint32_t) * nc))) return rc;
    SC_HIP(ctx, hipMemcpyAsync(ctx->xaxis.p, xaxis, sizeof(double) * nx,
                               hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(ctx->yaxis.p, yaxis, sizeof(double) * ny,
                               hipMemcpyHostToDevice, ctx->stream));
    // the block's digest; a block equal to the one this context already holds - same geometry, same cell
    // size, same bits: the next sl.match of a multi-scale job on the same data (the reference runs one
    // call per scale, docs/source/examples/channels.ipynb) - keeps its curvature planes and the
    // curvature spectra kept from the last search (option "spectra_mb")
    if ((rc = sc_ensure(ctx, ctx->digest, 3 * sizeof(unsigned long long)))) return rc;
    if ((rc = launch_dem_digest(ctx, (unsigned long long*)ctx->digest.p))) return rc;
    unsigned long long dg[3] = {0, 0, 0};
    SC_HIP(ctx, hipMemcpyAsync(dg, ctx->digest.p, sizeof(dg), hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const double sig[16] = {(double)ly, (double)lx, (double)gy0, (double)gx0, (double)ny, (double)nx, (double)cy0,
                            (double)cy1, (double)cx0, (double)cx1, dx, dy, (double)(wrap ? 1 : 0), 0, 0, 0};
    ctx->dem_unchanged = ctx->have_dem && dg[2] == 0 && dg[0] == ctx->dem_hash[0] && dg[1] == ctx->dem_hash[1] &&
                         memcmp(sig, ctx->dem_sig, sizeof(sig)) == 0;
    ctx->dem_hash[0] = dg[0];
    ctx->dem_hash[1] = dg[1];
    ctx->dem_nan = (long long)dg[2];
    memcpy(ctx->dem_sig, sig, sizeof(sig));
    if (!ctx->dem_unchanged) {
        if ((rc = launch_curv_planes(ctx))) return rc;
        fft_spectra_forget(ctx);
    }
    ctx->have_dem = true;
    if ((rc = sc_reset_best(ctx))) return rc;
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SC_OK;
}

extern "C" int sc_dem_info(sc_ctx* ctx, long long* nan_cells, unsigned long long* hash2, int* unchanged) {
    if (!ctx) return SC_ERR_INVALID;
    if (!ctx->have_dem) return sc_fail(ctx, SC_ERR_NO_DEM, "no DEM set");
    if (nan_cells) *nan_cells = ctx->dem_nan;
    if (hash2) { hash2[0] = ctx->dem_hash[0]; hash2[1] = ctx->dem_hash[1]; }
    if (unchanged) *unchanged = ctx->dem_unchanged ? 1 : 0;
    return SC_OK;
}

extern "C" int sc_set_dem(sc_ctx* ctx, const double* z, int ly, int lx, int gy0,
                          int gx0, int ny, int nx, int cy0, int cy1, int cx0,
                          int cx1, double dx, double dy, int wrap,
                          const double* xaxis, const double* yaxis) {
    if (!ctx || !z) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    if (ly < 3 || lx < 3) return sc_fail(ctx, SC_ERR_INVALID, "sc_set_dem: block too small");
    int rc = sc_ensure(ctx, ctx->z, sizeof(double) * (size_t)ly * lx);
    if (rc) return rc;
    SC_HIP(ctx, hipMemcpyAsync(ctx->z.p, z, sizeof(double) * (size_t)ly * lx,
                               hipMemcpyHostToDevice, ctx->stream));
    ctx->z_dev = (const double*)ctx->z.p;
    return set_dem_common(ctx, ly, lx, gy0, gx0, ny, nx, cy0, cy1, cx0, cx1, dx,
                          dy, wrap, xaxis, yaxis);
}

extern "C" int sc_set_dem_device(sc_ctx* ctx, const void* z_dev, int ly, int lx,
                                 int gy0, int gx0, int ny, int nx, int cy0,
                                 int cy1, int cx0, int cx1, double dx, double dy,
                                 int wrap, const double* xaxis,
                                 const double* yaxis) {
    if (!ctx || !z_dev) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    ctx->z_dev = (const double*)z_dev;
    return set_dem_common(ctx, ly, lx, gy0, gx0, ny, nx, cy0, cy1, cx0, cx1, dx,
                          dy, wrap, xaxis, yaxis);
}

extern "C" int sc_curvature(sc_ctx* ctx, double cc, double sc2, double ss, float* out) {
    if (!ctx || !out) return SC_ERR_INVALID;
    if (!ctx->have_dem) return sc_fail(ctx, SC_ERR_NO_DEM, "no DEM set");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    int rc = launch_curv_alpha(ctx, (float)cc, (float)sc2, (float)ss);
    if (rc) return rc;
    SC_HIP(ctx, hipMemcpyAsync(out, ctx->curv.p, sizeof(float) * (size_t)ctx->g.ly * ctx->g.lx,
                               hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SC_OK;
}

extern "C" int sc_curvature_f64(sc_ctx* ctx, double cos2, double sin_a, double cos_a, double sin2,
                               double* out) {
    if (!ctx || !out) return SC_ERR_INVALID;
    if (!ctx->have_dem) return sc_fail(ctx, SC_ERR_NO_DEM, "no DEM set");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = sizeof(double) * (size_t)ctx->g.ly * ctx->g.lx;
    int rc = sc_ensure(ctx, ctx->res, bytes);        // (the result planes' buffer: nothing else is live here)
    if (rc) return rc;
    rc = launch_curv_f64(ctx, cos2, sin_a, cos_a, sin2, (double*)ctx->res.p);
    if (rc) return rc;
    SC_HIP(ctx, hipMemcpyAsync(out, ctx->res.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SC_OK;
}

// ---------------------------------------------------------------------------
// generic plugin windows
// ---------------------------------------------------------------------------
extern "C" int sc_upload_window(sc_ctx* ctx, const double* w, int h, int wd, int* slot) {
    if (!ctx || !w || h <= 0 || wd <= 0 || !slot) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    size_t n = (size_t)h * wd;
    std::vector<float> wf(n);
    std::vector<uint8_t> wm(n);
    double l1 = 0.0;
    for (size_t i = 0; i < n; ++i) {
        wf[i] = (float)w[i];
        wm[i] = (w[i] != 0.0) ? 1 : 0;
        l1 += fabs(w[i]);
    }
    WindowSlot s;
    s.l1 = l1;
    s.h = h;
    s.wd = wd;
    SC_HIP(ctx, hipMalloc((void**)&s.w, n * sizeof(float)));
    SC_HIP(ctx, hipMalloc((void**)&s.m, n));
    SC_HIP(ctx, hipMemcpy(s.w, wf.data(), n * sizeof(float), hipMemcpyHostToDevice));
    SC_HIP(ctx, hipMemcpy(s.m, wm.data(), n, hipMemcpyHostToDevice));
    ctx->windows.push_back(s);
    *slot = (int)ctx->windows.size() - 1;
    return SC_OK;
}

extern "C" int sc_set_masks(sc_ctx* ctx, int slot, const uint8_t* limits, const uint8_t* err) {
    if (!ctx || slot < 0 || slot >= (int)ctx->windows.size()) return SC_ERR_INVALID;
    if (!ctx->have_dem) return sc_fail(ctx, SC_ERR_NO_DEM, "no DEM set");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    WindowSlot& s = ctx->windows[slot];
    size_t n = (size_t)ctx->g.ny * ctx->g.nx;
    const uint8_t* src[2] = {limits, err};
    uint8_t** dst[2] = {&s.mask_lim, &s.mask_err};
    for (int k = 0; k < 2; ++k) {
        if (*dst[k]) {
            SC_HIP(ctx, hipFree(*dst[k]));
            *dst[k] = nullptr;
        }
        if (src[k]) {
            SC_HIP(ctx, hipMalloc((void**)dst[k], n));
            SC_HIP(ctx, hipMemcpy(*dst[k], src[k], n, hipMemcpyHostToDevice));
        }
    }
    return SC_OK;
}

// ---------------------------------------------------------------------------
// the hot path
// ---------------------------------------------------------------------------
// compare()'s start state (core.py:222-225) in ONE launch: the record (snr 0, amp 0, no id), the resolution
// statistic, and the scratch records of a split row pass (a share left by an earlier search must not score
// against the new record).  Five memsets before: a fifth of a millisecond per five-scale C5 step in fills and gaps.
__global__ void __launch_bounds__(256)
k_reset_best(float* __restrict__ snr, float* __restrict__ amp, uint32_t* __restrict__ id, size_t nc,
             float* __restrict__ s2, size_t n2, unsigned long long* __restrict__ stats) {
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    for (size_t i = i0; i < nc; i += step) { snr[i] = 0.f; amp[i] = 0.f; id[i] = SC_ID_NONE; }
    for (size_t i = i0; i < n2; i += step) s2[i] = 0.f;
    if (i0 < 2) stats[i0] = 0ull;
}

extern "C" int sc_reset_best(sc_ctx* ctx) {
    if (!ctx) return SC_ERR_INVALID;
    if (!ctx->have_dem) return sc_fail(ctx, SC_ERR_NO_DEM, "no DEM set");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    size_t nc = (size_t)(ctx->g.cy1 - ctx->g.cy0) * (ctx->g.cx1 - ctx->g.cx0);
    int rc = sc_ensure(ctx, ctx->res_stats, 2 * sizeof(unsigned long long));
    if (rc) return rc;
    const size_t n2 = ctx->split_s.p ? ctx->split_s.cap / sizeof(float) : 0;
    const unsigned blocks = (unsigned)std::min<size_t>((std::max(nc, n2) + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(k_reset_best, dim3(blocks), dim3(256), 0, ctx->stream, (float*)ctx->best_snr.p,
                       (float*)ctx->best_amp.p, (uint32_t*)ctx->best_id.p, nc, (float*)ctx->split_s.p, n2,
                       (unsigned long long*)ctx->res_stats.p);
    SC_HIP(ctx, hipGetLastError());
    if (ctx->near.p) SC_HIP(ctx, hipMemsetAsync(ctx->near.p, 0, ctx->near.cap, ctx->stream));
    if (ctx->near_ev.p) SC_HIP(ctx, hipMemsetAsync(ctx->near_ev.p, 0, 16, ctx->stream));
    ctx->patch_n = 0;
    return SC_OK;
}

int sc_near_buffers(sc_ctx* ctx, unsigned long long** ev_count, uint32_t** ev, unsigned long long* ev_cap) {
    const size_t nc = (size_t)(ctx->g.cy1 - ctx->g.cy0) * (ctx->g.cx1 - ctx->g.cx0);
    const bool fresh = ctx->near.cap < nc;
    int rc = sc_ensure(ctx, ctx->near, nc);
    if (rc) return rc;
    if (fresh) SC_HIP(ctx, hipMemsetAsync(ctx->near.p, 0, ctx->near.cap, ctx->stream));
    // the event list: two per core cell or a million, whichever is more (12 bytes each); counter in front
    const unsigned long long cap = std::max<unsigned long long>(2ull * nc, 1ull << 20);
    const bool fresh_ev = ctx->near_ev.cap < 16 + 4 * SC_EVENT_WORDS * cap;
    if ((rc = sc_ensure(ctx, ctx->near_ev, 16 + 4 * SC_EVENT_WORDS * cap))) return rc;
    if (fresh_ev) SC_HIP(ctx, hipMemsetAsync(ctx->near_ev.p, 0, 16, ctx->stream));
    *ev_count = (unsigned long long*)ctx->near_ev.p;
    *ev = (uint32_t*)((char*)ctx->near_ev.p + 16);
    *ev_cap = (ctx->near_ev.cap - 16) / (4 * SC_EVENT_WORDS);
    ctx->near_w_used = ctx->near_w;
    return SC_OK;
}

extern "C" int sc_get_near_ties(sc_ctx* ctx, uint8_t* out) {
    if (!ctx || !out) return SC_ERR_INVALID;
    if (!ctx->have_dem) return sc_fail(ctx, SC_ERR_NO_DEM, "no DEM set");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nc = (size_t)(ctx->g.cy1 - ctx->g.cy0) * (ctx->g.cx1 - ctx->g.cx0);
    if (!ctx->near.p || ctx->near.cap < nc) {        // no search has run with the option on: nothing flagged
        memset(out, 0, nc);
        return SC_OK;
    }
    SC_HIP(ctx, hipMemcpyAsync(out, ctx->near.p, nc, hipMemcpyDeviceToHost, ctx->stream));
    return sc_sync(ctx);
}

extern "C" int sc_get_near_events(sc_ctx* ctx, uint32_t* events, long long capacity, long long* n_events) {
    if (!ctx || !n_events || capacity < 0 || (capacity > 0 && !events)) return SC_ERR_INVALID;
    *n_events = 0;
    if (!ctx->near_ev.p) return SC_OK;                   // no FFT search has run with the option on
    SC_HIP(ctx, hipSetDevice(ctx->device));
    unsigned long long n = 0;
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SC_HIP(ctx, hipMemcpy(&n, ctx->near_ev.p, sizeof(n), hipMemcpyDeviceToHost));
    *n_events = (long long)n;
    const unsigned long long held = std::min<unsigned long long>(n, (ctx->near_ev.cap - 16) / (4 * SC_EVENT_WORDS));
    const unsigned long long take = std::min<unsigned long long>(held, (unsigned long long)capacity);
    if (n > held)                                        // the device list overflowed: events were dropped - said, not left to the caller's arithmetic
        return sc_fail(ctx, SC_ERR_UNSUPPORTED, "sc_get_near_events: the event list overflowed (%llu near-ties, room for %llu)", n, held);
    if (held > (unsigned long long)capacity) return SC_OK;                  // the caller asks again with room (*n_events says how much)
    if (take) SC_HIP(ctx, hipMemcpy(events, (const char*)ctx->near_ev.p + 16, 4 * SC_EVENT_WORDS * take, hipMemcpyDeviceToHost));
    return SC_OK;
}

// shared by sc_score_cells_f64 (tsel = nullptr: every cell against every template) and sc_score_pairs_f64
static int score_f64(sc_ctx* ctx, const int32_t* cells, const int32_t* tsel, int m, double* amp, double* snr, const char* who) {
    if (!ctx->have_dem) return sc_fail(ctx, SC_ERR_NO_DEM, "no DEM set");
    const int n = ctx->last_batch;
    if (n <= 0) return sc_fail(ctx, SC_ERR_INVALID, "%s: no search has run in this context", who);
    if (ctx->templ_windows)
        return sc_fail(ctx, SC_ERR_UNSUPPORTED, "%s: built-in templates only (a plugin's window is float32 on the device)", who);
    SC_HIP(ctx, hipSetDevice(ctx->device));
    for (int k = 0; k < m; ++k) {
        if (cells[2 * k] < 0 || cells[2 * k] >= ctx->g.ny || cells[2 * k + 1] < 0 || cells[2 * k + 1] >= ctx->g.nx)
            return sc_fail(ctx, SC_ERR_INVALID, "%s: cell %d outside the DEM", who, k);
        if (tsel && (tsel[k] < 0 || tsel[k] >= n))
            return sc_fail(ctx, SC_ERR_INVALID, "%s: pair %d names template %d of %d", who, k, tsel[k], n);
    }
    const size_t nout = tsel ? (size_t)m : (size_t)m * n;
    int rc = sc_ensure(ctx, ctx->score, sizeof(int32_t) * 3 * (size_t)m + 16 + sizeof(double) * 2 * nout);
    if (rc) return rc;
    double* d_amp = (double*)ctx->score.p;
    double* d_snr = d_amp + nout;
    int* d_cells = (int*)(d_snr + nout);
    int* d_tsel = tsel ? d_cells + 2 * (size_t)m : nullptr;
    SC_HIP(ctx, hipMemcpyAsync(d_cells, cells, sizeof(int32_t) * 2 * (size_t)m, hipMemcpyHostToDevice, ctx->stream));
    if (tsel) SC_HIP(ctx, hipMemcpyAsync(d_tsel, tsel, sizeof(int32_t) * (size_t)m, hipMemcpyHostToDevice, ctx->stream));
    // (grid.y is limited to 65535: the templates of a search are at most a few thousand)
    if (!tsel && n > 65535) return sc_fail(ctx, SC_ERR_UNSUPPORTED, "%s: %d templates", who, n);
    if ((rc = launch_score_f64(ctx, d_cells, d_tsel, m, n, d_amp, d_snr))) return rc;
    SC_HIP(ctx, hipMemcpyAsync(amp, d_amp, sizeof(double) * nout, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(snr, d_snr, sizeof(double) * nout, hipMemcpyDeviceToHost, ctx->stream));
    return sc_sync(ctx);
}

extern "C" int sc_score_pairs_f64(sc_ctx* ctx, const int32_t* cells, const int32_t* templates, int m, double* amp, double* snr) {
    if (!ctx || !cells || !templates || m <= 0 || !amp || !snr) return SC_ERR_INVALID;
    return score_f64(ctx, cells, templates, m, amp, snr, "sc_score_pairs_f64");
}

extern "C" int sc_score_cells_f64(sc_ctx* ctx, const int32_t* cells, int m, int n_templates, double* amp, double* snr) {
    if (!ctx || !cells || m <= 0 || !amp || !snr) return SC_ERR_INVALID;
    // (the caller sized amp / snr for m x n_templates values: it must be the number this context will write)
    if (n_templates != ctx->last_batch)
        return sc_fail(ctx, SC_ERR_INVALID, "sc_score_cells_f64: the caller expects %d templates, the last search held %d",
                       n_templates, ctx->last_batch);
    return score_f64(ctx, cells, nullptr, m, amp, snr, "sc_score_cells_f64");
}

extern "C" int sc_get_resolution_stats(sc_ctx* ctx, long long* wins, long long* near_floor) {
    if (!ctx || !wins || !near_floor) return SC_ERR_INVALID;
    *wins = *near_floor = 0;
    if (!ctx->res_stats.p) return SC_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    unsigned long long h[2] = {0, 0};
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SC_HIP(ctx, hipMemcpy(h, ctx->res_stats.p, sizeof(h), hipMemcpyDeviceToHost));
    *wins = (long long)h[0];
    *near_floor = (long long)h[1];
    return SC_OK;
}

static int check_templates(sc_ctx* ctx, const sc_template* t, int n, const sc_plan* plan) {
    const Geom& g = ctx->g;
    for (int i = 0; i < n; ++i) {
        const sc_template& s = t[i];
        if (s.kind < SC_KIND_SCARP || s.kind > SC_KIND_WINDOW)
            return sc_fail(ctx, SC_ERR_INVALID, "template %d: unknown kind %d", i, s.kind);
        if (s.pmax < s.pmin || s.qmax < s.qmin)
            return sc_fail(ctx, SC_ERR_INVALID, "template %d: empty support box", i);
        // support cells must exist on the ny x nx template grid
        if (g.ny / 2 + s.pmin < 0 || g.ny / 2 + s.pmax >= g.ny ||
            g.nx / 2 + s.qmin < 0 || g.nx / 2 + s.qmax >= g.nx)
            return sc_fail(ctx, SC_ERR_INVALID, "template %d: support box leaves the grid", i);
        if (s.kind == SC_KIND_WINDOW) {
            if (s.window < 0 || s.window >= (int)ctx->windows.size())
                return sc_fail(ctx, SC_ERR_INVALID, "template %d: bad window slot", i);
            const WindowSlot& w = ctx->windows[s.window];
            if (w.h != s.pmax - s.pmin + 1 || w.wd != s.qmax - s.qmin + 1)
                return sc_fail(ctx, SC_ERR_INVALID, "template %d: window size mismatch", i);
        }
        if (plan->method == SC_METHOD_DIRECT && !direct_window_fits(s.qmax - s.qmin + 1))
            return sc_fail(ctx, SC_ERR_UNSUPPORTED, "template %d: a %d-cell wide window exceeds the "
                           "real-space kernel's LDS slab; use SC_METHOD_FFT", i, s.qmax - s.qmin + 1);
        if (plan->method == SC_METHOD_FFT) {
            if (s.pmax > plan->Py || s.qmax > plan->Qx ||
                plan->Py - s.pmin > plan->Ty - plan->Vy + (plan->circ_y ? plan->Ty : 0) ||
                plan->Qx - s.qmin > plan->Tx - plan->Vx + (plan->circ_x ? plan->Tx : 0))
                return sc_fail(ctx, SC_ERR_INVALID, "template %d: support exceeds the plan", i);
            if (s.pmax - s.pmin >= plan->Ty || s.qmax - s.qmin >= plan->Tx)
                return sc_fail(ctx, SC_ERR_INVALID, "template %d: support exceeds the tile", i);
        }
    }
    return SC_OK;
}

// One descriptor as the device's table entry (offsets into the per-chunk window buffers are match_impl's to fill)
static void templ_from_descriptor(const sc_ctx* ctx, const sc_template& s, TemplDev& d, double* sums2, double* wl1) {
    const Geom& g = ctx->g;
    d.kind = s.kind; d.flags = s.flags;
    d.cos_a = s.cos_a; d.sin_a = s.sin_a; d.c = s.c; d.d = s.d;
    d.p0 = s.p0; d.p1 = s.p1;
    d.ilo = s.ilo; d.ihi = s.ihi; d.jlo = s.jlo; d.jhi = s.jhi;
    if (s.flags & SC_FLAG_NO_LIMITS) { d.ilo = 0; d.ihi = g.ny - 1; d.jlo = 0; d.jhi = g.nx - 1; }
    d.pmin = s.pmin; d.pmax = s.pmax; d.qmin = s.qmin; d.qmax = s.qmax;
    d.id = s.id;
    d.wh = s.pmax - s.pmin + 1;
    d.ww = s.qmax - s.qmin + 1;
    d.win_off = 0;
    d.mask_lim = nullptr; d.mask_err = nullptr;
    if (s.kind == SC_KIND_WINDOW) {
        d.mask_lim = ctx->windows[s.window].mask_lim;
        d.mask_err = ctx->windows[s.window].mask_err;
        sums2[0] = s.p0;
        sums2[1] = s.p1;
        *wl1 = ctx->windows[s.window].l1;
    }
}

// The descriptors of a whole search as the template table the float64 scorer reads, without matching them
// (sc_settle_pairs: a rank of an orientation-sharded search settles candidates of templates other ranks matched)
int sc_load_templates(sc_ctx* ctx, const sc_template* t, int n) {
    sc_plan plan{};
    plan.method = -1;                            // (neither path's support limits: the float64 scorer has none)
    int rc = check_templates(ctx, t, n, &plan);
    if (rc) return rc;
    if (ctx->async_in_flight) {
        SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->async_in_flight = false;
    }
    std::vector<TemplDev>& h = ctx->h_templ;
    h.assign(n, TemplDev{});
    ctx->h_sums.assign(2 * (size_t)n, 0.0);
    ctx->h_wl1.assign(n, 0.0);
    ctx->templ_windows = false;
    for (int j = 0; j < n; ++j) {
        templ_from_descriptor(ctx, t[j], h[j], &ctx->h_sums[2 * (size_t)j], &ctx->h_wl1[j]);
        ctx->templ_windows = ctx->templ_windows || t[j].kind == SC_KIND_WINDOW;
    }
    if ((rc = sc_ensure(ctx, ctx->templ, sizeof(TemplDev) * n))) return rc;
    ctx->async_in_flight = true;
    SC_HIP(ctx, hipMemcpyAsync(ctx->templ.p, h.data(), sizeof(TemplDev) * n, hipMemcpyHostToDevice, ctx->stream));
    ctx->last_batch = n;
    return SC_OK;
}

static int match_impl(sc_ctx* ctx, const sc_template* t, int n, const sc_plan* plan,
                      bool to_maps) {
    if (!ctx || !t || !plan || n <= 0) return SC_ERR_INVALID;
    if (!ctx->have_dem) return sc_fail(ctx, SC_ERR_NO_DEM, "no DEM set");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    if (plan->method != SC_METHOD_DIRECT && plan->method != SC_METHOD_FFT)
        return sc_fail(ctx, SC_ERR_INVALID, "unknown method %d", plan->method);
    int rc = check_templates(ctx, t, n, plan);
    if (rc) return rc;
    if (!to_maps) ctx->patch_n = 0;            // (the record moves on: what sc_settle_exact patched no longer describes it)
    const Geom& g = ctx->g;
    FftGeom fg{};
    const int group = std::max(1, plan->group);
    const int CHUNK = SC_MAX_GROUP;
    if (plan->method == SC_METHOD_FFT) {
        fg.Ty = plan->Ty; fg.Tx = plan->Tx; fg.Vy = plan->Vy; fg.Vx = plan->Vx;
        fg.nty = plan->nty; fg.ntx = plan->ntx; fg.circ_y = plan->circ_y;
        fg.circ_x = plan->circ_x; fg.Py = plan->Py; fg.Qx = plan->Qx;
        fg.ntiles = fg.nty * fg.ntx;
        if (fg.nty < 1 || fg.ntx < 1 || fg.Vy < 1 || fg.Vx < 1)
            return sc_fail(ctx, SC_ERR_INVALID, "bad tile plan");
        if ((fg.circ_y && (fg.Ty != g.ny || !g.wrap || fg.nty != 1)) ||
            (fg.circ_x && (fg.Tx != g.nx || !g.wrap || fg.ntx != 1)))
            return sc_fail(ctx, SC_ERR_INVALID, "circular axis needs T == n and the whole DEM");
        if ((long long)fg.nty * fg.Vy < g.cy1 - g.cy0 || (long long)fg.ntx * fg.Vx < g.cx1 - g.cx0)
            return sc_fail(ctx, SC_ERR_INVALID, "tiles do not cover the core");
    }
    if (to_maps) {
        size_t nc = (size_t)(g.cy1 - g.cy0) * (g.cx1 - g.cx0);
        if ((rc = sc_ensure(ctx, ctx->map_amp, sizeof(float) * nc))) return rc;
        if ((rc = sc_ensure(ctx, ctx->map_snr, sizeof(float) * nc))) return rc;
    }

    // Orientation runs: consecutive templates with equal (cc, sc2, ss) share one curvature
    // plane (at most CHUNK of them per run).
    // (the context's own vectors: the uploads below are asynchronous and nothing waits for them before
    //  the launches; the stream was drained at the end of the previous call, so they are free to reuse -
    //  after an sc_match_async without sc_sync it is drained here)
    if (ctx->async_in_flight) {
        SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->async_in_flight = false;
    }
    std::vector<TemplDev>& h = ctx->h_templ;
    std::vector<double>&sums = ctx->h_sums, &wl1 = ctx->h_wl1;
    h.assign(n, TemplDev{});
    sums.assign(2 * (size_t)n, 0.0);
    wl1.assign(n, 0.0);
    struct Run { int first, n, parity; bool full, long_runs; };
    std::vector<Run> runs;
    for (int i = 0; i < n;) {
        int j = i;
        int parity = -1;
        bool full = false;
        while (j < n && j - i < CHUNK && t[j].cc == t[i].cc && t[j].sc2 == t[i].sc2 &&
               t[j].ss == t[i].ss) {
            const sc_template& s = t[j];
            TemplDev& d = h[j];
            templ_from_descriptor(ctx, s, d, &sums[2 * (size_t)j], &wl1[j]);
            // flip symmetry of the built-in templates (k_split_templ_sym): the same parity
            // for the whole run, and support boxes that map onto themselves
            int pj = s.kind == SC_KIND_SCARP ? 1 : (s.kind == SC_KIND_RICKER ? 2 : 0);
            if (s.pmin + s.pmax != -(1 - g.oy) || s.qmin + s.qmax != -(1 - g.ox)) pj = 0;
            parity = (parity == -1 || parity == pj) ? pj : 0;
            full |= (d.flags & (SC_FLAG_ERR_XR_LE0 | SC_FLAG_ERR_XR_GE0)) != 0 ||
                    d.mask_lim != nullptr || d.mask_err != nullptr;
            ++j;
        }
        // real-space path: does some template of the run have window rows of 16 taps or more along x (SC_DR_SHARE_MIN)
        // (about min(2c / |cos a|, 2d / |sin a|) cells for the built-in rectangles; a window uploaded
        // by the host: unknown, taken as long)?  Decides the kernel form of the run's launch
        // (launch_direct) - per RUN, so that a template's sums do not depend on what it is batched with
        bool long_runs = false;
        for (int k = i; k < j && !long_runs; ++k) {
            if (t[k].kind == SC_KIND_WINDOW) { long_runs = true; break; }
            const double ca = fabs(t[k].cos_a) + 1e-9, sa = fabs(t[k].sin_a) + 1e-9;
            long_runs = std::min(2.0 * t[k].c / ca, 2.0 * t[k].d / sa) / fabs(ctx->dx) >= 16.0;
        }
        runs.push_back({i, j - i, parity < 0 ? 0 : parity, full, long_runs});
        i = j;
    }
    // Chunks: one run, or - small FFT searches - nb consecutive runs of equal length, parity
    // and mask kind sent through every launch together (sc_fft.hip, "Orientation batching")
    struct Chunk { int first, n, nb, wh, ww, parity; bool full, long_runs; size_t cells; int run0; };
    std::vector<Chunk> chunks;
    size_t max_cells = 0, max_dcells = 0, max_spans = 0;
    int nb_max = 1, max_batch_templ = 1;
    for (size_t r = 0; r < runs.size();) {
        int nb = 1;
        if ((plan->method == SC_METHOD_FFT || (!ctx->batch_off && ctx->variant != 10)) && !to_maps) {
            // (real space: small DEMs fold up to 32 orientations per launch, in order, inside the
            //  workgroup that owns a patch - a 512 x 512 search is 905 launches of a few microseconds
            //  otherwise; DEMs of a thousand workgroups per orientation gain nothing)
            const long long wg1 = (long long)((g.cx1 - g.cx0 + 255) / 256) * ((g.cy1 - g.cy0 + 7) / 8);
            const int want = plan->method == SC_METHOD_FFT ? fft_batch_orientations(ctx, fg, runs[r].n, group)
                                                           : (wg1 <= 1024 ? 32 : 1);
            // compatible runs ahead, up to what one launch can hold (64 templates, 64 curvature planes)
            // (round 5: SC_MAX_BATCH templates per launch sequence - the row pass takes them in slices of whole
            //  orientations, at most SC_MAX_GROUP templates each: fft_inverse_fold)
            const int hard = plan->method == SC_METHOD_FFT ? std::min(SC_MAX_ORIENT, SC_MAX_BATCH / std::max(1, runs[r].n)) : 32;
            // (counted beyond what one launch can hold, up to two full batches: whether a remainder rides
            //  along or the rest is split evenly is decided on what is really left - capped at `hard`, 91
            //  orientations of ten templates went six at a time as 3 + 3 instead of 4 + 4 + ...)
            const int look = std::max(hard, 2 * want + 1);
            int avail = 1;
            while (avail < look && r + avail < runs.size() && runs[r + avail].n == runs[r].n &&
                   runs[r + avail].parity == runs[r].parity && runs[r + avail].full == runs[r].full &&
                   (plan->method == SC_METHOD_FFT || runs[r + avail].long_runs == runs[r].long_runs))
                ++avail;
            // `want` fills the chip; a tail of a few orientations costs a whole launch sequence of its
            // own (C1: 35 orientations were 32 + 3), so a remainder up to a quarter of `want` rides along
            // and a larger one is split evenly
            if (want <= 1) nb = 1;
            else if (avail <= want + want / 4) nb = avail;
            else if (avail < 2 * want) nb = (avail + 1) / 2;
            else nb = want;
            nb = std::min(nb, hard);
        }
        size_t off = 0, doff = 0;
        int wh = 0, ww = 0, soff = 0;
        for (int j = runs[r].first; j < runs[r].first + nb * runs[r].n; ++j) {
            h[j].win_off = (long long)off;
            off += (size_t)h[j].wh * h[j].ww;
            off = (off + 3) & ~(size_t)3;
            wh = std::max(wh, h[j].wh);
            ww = std::max(ww, h[j].ww);
            // real-space path: rows reversed and padded to groups of four taps (k_direct_prep)
            h[j].dpitch = (h[j].ww + 3) & ~3;
            h[j].dwin_off = (long long)doff;
            h[j].span_off = soff;
            doff += (size_t)h[j].wh * h[j].dpitch;
            soff += h[j].wh;
        }
        max_dcells = std::max(max_dcells, doff);
        max_spans = std::max(max_spans, (size_t)soff);
        chunks.push_back({runs[r].first, runs[r].n, nb, wh, ww, runs[r].parity, runs[r].full, runs[r].long_runs, off, (int)r});
        max_cells = std::max(max_cells, off);
        nb_max = std::max(nb_max, nb);
        max_batch_templ = std::max(max_batch_templ, nb * runs[r].n);
        r += nb;
    }
    // Spectra kept across searches (option "spectra_mb"): slot of run r = its orientation's index among the
    // search's distinct consecutive orientations; a batch must sit in consecutive slots
    std::vector<int> slot_of(runs.size(), 0);
    int n_slots = 0;
    if (plan->method == SC_METHOD_FFT && !to_maps && ctx->spec_mb > 0.0) {
        for (size_t r = 0; r < runs.size(); ++r) {
            const sc_template &a = t[runs[r].first];
            const bool same = r > 0 && a.cc == t[runs[r - 1].first].cc && a.sc2 == t[runs[r - 1].first].sc2 &&
                              a.ss == t[runs[r - 1].first].ss;
            slot_of[r] = same ? slot_of[r - 1] : n_slots++;
        }
        for (const Chunk& c : chunks)
            for (int b = 1; b < c.nb; ++b)
                if (slot_of[c.run0 + b] != slot_of[c.run0] + b) n_slots = 0;     // (an orientation twice in a batch)
    }
    if (plan->method == SC_METHOD_FFT &&
        (rc = fft_prepare(ctx, fg, std::min(n, std::max(CHUNK, max_batch_templ)), group, nb_max, n_slots)))
        return rc;
    const bool keep = plan->method == SC_METHOD_FFT && ctx->spec_slots > 0;
    if ((rc = sc_ensure(ctx, ctx->templ, sizeof(TemplDev) * n))) return rc;
    if ((rc = sc_ensure(ctx, ctx->sums, sizeof(double) * 2 * n))) return rc;
    if ((rc = sc_ensure(ctx, ctx->wl1, sizeof(double) * n))) return rc;
    if ((rc = sc_ensure(ctx, ctx->win_w, sizeof(float) * max_cells))) return rc;
    if ((rc = sc_ensure(ctx, ctx->win_m, max_cells))) return rc;
    if (plan->method == SC_METHOD_DIRECT) {
        if ((rc = sc_ensure(ctx, ctx->curv, sizeof(float) * (size_t)g.ly * g.lx * nb_max))) return rc;
        if ((rc = sc_ensure(ctx, ctx->dwin, sizeof(float2) * std::max<size_t>(max_dcells, 4)))) return rc;
        if ((rc = sc_ensure(ctx, ctx->spans, sizeof(int4) * std::max<size_t>(max_spans, 1)))) return rc;
    }
    // from here on the context's host vectors are the source of copies in flight: whatever way this call ends
    // (a launch failure or an allocation failing in a later chunk returns without draining the stream), the
    // next call's prologue waits before it reuses them; sc_sync, at the end of every successful synchronous
    // call, clears the flag
    ctx->async_in_flight = true;
    SC_HIP(ctx, hipMemcpyAsync(ctx->templ.p, h.data(), sizeof(TemplDev) * n,
                               hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(ctx->sums.p, sums.data(), sizeof(double) * 2 * n,
                               hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(ctx->wl1.p, wl1.data(), sizeof(double) * n,
                               hipMemcpyHostToDevice, ctx->stream));
    ctx->last_batch = n;
    ctx->templ_windows = false;
    for (int k = 0; k < n; ++k) ctx->templ_windows = ctx->templ_windows || t[k].kind == SC_KIND_WINDOW;

    double cur[3] = {0, 0, 0};
    bool have_curv = false;
    for (const Chunk& c : chunks) {
        const int n_all = c.nb * c.n;
        // kept spectra: this chunk's slots, and whether they already hold its orientations
        bool hit = false;
        if (keep) {
            const int s0 = slot_of[c.run0];
            ctx->uc_off = (size_t)s0 * ctx->spec_uc_stride;
            ctx->norms_off = (size_t)s0 * ctx->spec_norm_stride;
            hit = true;
            for (int b = 0; b < c.nb; ++b) {
                const sc_template& sb = t[c.first + b * c.n];
                const double* k = &ctx->spec_key[3 * (size_t)(s0 + b)];
                hit = hit && k[0] == sb.cc && k[1] == sb.sc2 && k[2] == sb.ss;
            }
        }
        if (hit) {
            have_curv = false;                    // (the curvature plane itself was not rebuilt)
        } else if (c.nb > 1) {
            float coef[SC_MAX_ORIENT][3];
            for (int b = 0; b < c.nb; ++b) {
                const sc_template& sb = t[c.first + b * c.n];
                coef[b][0] = (float)sb.cc; coef[b][1] = (float)sb.sc2; coef[b][2] = (float)sb.ss;
            }
            // FFT tiles: the orientations' curvature is mixed from the stencil planes inside the forward row
            // kernel (no plane written and read back; option "variant" 17: the separate k_curv_alpha pass, for the
            // cross-check - same bits); the real-space kernel reads the plane(s)
            const bool fused = plan->method == SC_METHOD_FFT && ctx->variant != 17;
            if (!fused && (rc = launch_curv_alpha_batch(ctx, coef, c.nb))) return rc;
            if (plan->method == SC_METHOD_FFT && (rc = fft_forward_curv(ctx, fg, c.nb, fused ? coef : nullptr))) return rc;
            have_curv = false;                    // plane 0 no longer belongs to a single run
        } else {
            const sc_template& s0 = t[c.first];
            if (!have_curv || s0.cc != cur[0] || s0.sc2 != cur[1] || s0.ss != cur[2]) {
                const bool fused = plan->method == SC_METHOD_FFT && ctx->variant != 17;
                const float coef1[1][3] = {{(float)s0.cc, (float)s0.sc2, (float)s0.ss}};
                if (!fused && (rc = launch_curv_alpha(ctx, coef1[0][0], coef1[0][1], coef1[0][2]))) return rc;
                if (plan->method == SC_METHOD_FFT && (rc = fft_forward_curv(ctx, fg, 1, fused ? coef1 : nullptr))) return rc;
                cur[0] = s0.cc; cur[1] = s0.sc2; cur[2] = s0.ss;
                have_curv = true;
            }
        }
        if (keep && !hit)
            for (int b = 0; b < c.nb; ++b) {
                const sc_template& sb = t[c.first + b * c.n];
                double* k = &ctx->spec_key[3 * (size_t)(slot_of[c.run0] + b)];
                k[0] = sb.cc; k[1] = sb.sc2; k[2] = sb.ss;
            }
        for (int j = c.first; j < c.first + n_all; ++j) {
            if (t[j].kind != SC_KIND_WINDOW) continue;
            const WindowSlot& w = ctx->windows[t[j].window];
            size_t cells = (size_t)w.h * w.wd;
            SC_HIP(ctx, hipMemcpyAsync((float*)ctx->win_w.p + h[j].win_off, w.w,
                                       cells * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
            SC_HIP(ctx, hipMemcpyAsync((uint8_t*)ctx->win_m.p + h[j].win_off, w.m, cells,
                                       hipMemcpyDeviceToDevice, ctx->stream));
        }
        if ((rc = launch_windows(ctx, c.first, n_all, c.wh, c.ww))) return rc;
        if (plan->method == SC_METHOD_DIRECT) {
            if ((rc = launch_direct(ctx, c.first, c.n, to_maps, c.nb, c.wh, c.ww, c.long_runs))) return rc;
        } else {
            if ((rc = fft_forward_templates(ctx, fg, c.first, n_all, c.parity))) return rc;
            if ((rc = fft_inverse_fold(ctx, fg, c.first, c.n, group, to_maps, c.full, c.parity, c.nb))) return rc;
        }
    }
    return SC_OK;
}

extern "C" int sc_match_async(sc_ctx* ctx, const sc_template* t, int n, const sc_plan* plan) {
    int rc = match_impl(ctx, t, n, plan, false);
    if (ctx) ctx->async_in_flight = true;
    return rc;
}

extern "C" int sc_sync(sc_ctx* ctx) {
    if (!ctx) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->async_in_flight = false;
    sc_prof_collect(ctx);
    return SC_OK;
}

extern "C" int sc_match(sc_ctx* ctx, const sc_template* t, int n, const sc_plan* plan) {
    int rc = match_impl(ctx, t, n, plan, false);
    if (rc) return rc;
    return sc_sync(ctx);
}

extern "C" int sc_match_template(sc_ctx* ctx, const sc_template* t, const sc_plan* plan,
                                 float* amp, float* snr) {
    if (!amp || !snr) return SC_ERR_INVALID;
    int rc = match_impl(ctx, t, 1, plan, true);
    if (rc) return rc;
    size_t nc = (size_t)(ctx->g.cy1 - ctx->g.cy0) * (ctx->g.cx1 - ctx->g.cx0);
    SC_HIP(ctx, hipMemcpyAsync(amp, ctx->map_amp.p, sizeof(float) * nc, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(snr, ctx->map_snr.p, sizeof(float) * nc, hipMemcpyDeviceToHost, ctx->stream));
    return sc_sync(ctx);
}

extern "C" int sc_get_best(sc_ctx* ctx, float* amp, float* snr, uint32_t* id) {
    if (!ctx || !amp || !snr || !id) return SC_ERR_INVALID;
    if (!ctx->have_dem) return sc_fail(ctx, SC_ERR_NO_DEM, "no DEM set");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    size_t nc = (size_t)(ctx->g.cy1 - ctx->g.cy0) * (ctx->g.cx1 - ctx->g.cx0);
    SC_HIP(ctx, hipMemcpyAsync(amp, ctx->best_amp.p, sizeof(float) * nc, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(snr, ctx->best_snr.p, sizeof(float) * nc, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(id, ctx->best_id.p, sizeof(uint32_t) * nc, hipMemcpyDeviceToHost, ctx->stream));
    return sc_sync(ctx);
}

// (amp, snr, id) float32 record -> the reference's four float64 planes
__global__ void __launch_bounds__(256)
k_result(const float* __restrict__ amp, const float* __restrict__ snr,
         const uint32_t* __restrict__ id, const double* __restrict__ par,
         const double* __restrict__ ang, uint32_t n_ids, size_t nc, double* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nc; i += (size_t)gridDim.x * 256) {
        const uint32_t t = id[i];
        const bool won = t < n_ids;                 // SC_ID_NONE is larger than any id
        out[i] = (double)amp[i];
        out[nc + i] = won ? par[t] : 0.0;
        out[2 * nc + i] = won ? ang[t] : 0.0;
        out[3 * nc + i] = (double)snr[i];
    }
}

// k_result on any record of n cells (sc_gather_result converts the other ranks' records at the root)
int sc_launch_result(sc_ctx* ctx, const float* amp, const float* snr, const uint32_t* id, const double* tab_par,
                     const double* tab_ang, int n_ids, size_t n, double* planes) {
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 256 * 64);
    hipLaunchKernelGGL(k_result, dim3(blocks), dim3(256), 0, ctx->stream, amp, snr, id, tab_par, tab_ang,
                       (uint32_t)n_ids, n, planes);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

// the record of this context's core as four float64 planes in ctx->res (device)
int sc_result_planes(sc_ctx* ctx, const double* param_of_id, const double* angle_of_id, int n_ids,
                     double** planes_out, size_t* nc_out) {
    if (!ctx->have_dem) return sc_fail(ctx, SC_ERR_NO_DEM, "no DEM set");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nc = (size_t)(ctx->g.cy1 - ctx->g.cy0) * (ctx->g.cx1 - ctx->g.cx0);
    int rc = sc_ensure(ctx, ctx->res, sizeof(double) * (4 * nc + 2 * (size_t)n_ids));
    if (rc) return rc;
    double* planes = (double*)ctx->res.p;
    double* tab = planes + 4 * nc;
    SC_HIP(ctx, hipMemcpyAsync(tab, param_of_id, sizeof(double) * n_ids, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(tab + n_ids, angle_of_id, sizeof(double) * n_ids, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = sc_launch_result(ctx, (const float*)ctx->best_amp.p, (const float*)ctx->best_snr.p,
                               (const uint32_t*)ctx->best_id.p, tab, tab + n_ids, n_ids, nc, planes)))
        return rc;
    if (ctx->patch_n && (rc = sc_apply_patches(ctx, tab, tab + n_ids, n_ids, nc, planes))) return rc;
    *planes_out = planes;
    *nc_out = nc;
    return SC_OK;
}

extern "C" int sc_get_result(sc_ctx* ctx, const double* param_of_id, const double* angle_of_id,
                             int n_ids, double* out) {
    if (!ctx || !param_of_id || !angle_of_id || n_ids <= 0 || !out) return SC_ERR_INVALID;
    double* planes = nullptr;
    size_t nc = 0;
    int rc = sc_result_planes(ctx, param_of_id, angle_of_id, n_ids, &planes, &nc);
    if (rc) return rc;
    SC_HIP(ctx, hipMemcpyAsync(out, planes, sizeof(double) * 4 * nc, hipMemcpyDeviceToHost, ctx->stream));
    return sc_sync(ctx);
}

extern "C" int sc_get_template_sums(sc_ctx* ctx, int n, double* n_out, double* ts_out) {
    if (!ctx || !n_out || !ts_out || n <= 0 || n > ctx->last_batch) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<double> h(2 * (size_t)n);
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SC_HIP(ctx, hipMemcpy(h.data(), ctx->sums.p, sizeof(double) * 2 * n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {
        n_out[i] = h[2 * (size_t)i] + SC_EPS;
        ts_out[i] = h[2 * (size_t)i + 1];
    }
    return SC_OK;
}

// ---------------------------------------------------------------------------
// nodata fill (pre-step of the matcher: DEMGrid._fill_nodata)
// ---------------------------------------------------------------------------
extern "C" int sc_fill_nodata(sc_ctx* ctx, double* z, int ny, int nx, double max_search_distance,
                              int smoothing_iterations, long long* remaining) {
    if (!ctx || !z || ny <= 0 || nx <= 0 || !(max_search_distance >= 0.0) || smoothing_iterations < 0)
        return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)ny * nx;
    int rc;
    if ((rc = sc_ensure(ctx, ctx->fill[0], sizeof(double) * n))) return rc;
    if ((rc = sc_ensure(ctx, ctx->fill[1], sizeof(double) * n))) return rc;
    if ((rc = sc_ensure(ctx, ctx->fill[2], sizeof(int) * 2 * n + 16))) return rc;      // up | down, contiguous
    if ((rc = sc_ensure(ctx, ctx->fill[3], 16))) return rc;
    SC_HIP(ctx, hipMemcpyAsync(ctx->fill[0].p, z, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    int* up = (int*)ctx->fill[2].p;
    rc = launch_fill_nodata(ctx, (double*)ctx->fill[0].p, (double*)ctx->fill[1].p, up, up + n, ny, nx,
                            max_search_distance, smoothing_iterations, (unsigned long long*)ctx->fill[3].p);
    if (rc) return rc;
    unsigned long long left = 0;
    SC_HIP(ctx, hipMemcpyAsync(z, ctx->fill[0].p, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(&left, ctx->fill[3].p, sizeof(left), hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (remaining) *remaining = (long long)left;
    return SC_OK;
}

// ---------------------------------------------------------------------------
// compare() on host results
// ---------------------------------------------------------------------------
extern "C" int sc_compare_begin(sc_ctx* ctx, int ny, int nx) {
    if (!ctx || ny <= 0 || nx <= 0) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    size_t n = (size_t)ny * nx;
    int rc;
    for (int k = 0; k < 4; ++k) {
        if ((rc = sc_ensure(ctx, ctx->cmp[k], sizeof(double) * n))) return rc;
        SC_HIP(ctx, hipMemsetAsync(ctx->cmp[k].p, 0, sizeof(double) * n, ctx->stream));
    }
    for (int k = 0; k < 2; ++k)
        if ((rc = sc_ensure(ctx, ctx->cmp_in[k], sizeof(double) * n))) return rc;
    ctx->cmp_n = n;
    return SC_OK;
}

static int compare_fold_impl(sc_ctx* ctx, const double* amp, const double* snr,
                             const double* age_p, const double* angle_p, double age, double angle) {
    if (!ctx || !amp || !snr || ctx->cmp_n == 0) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    size_t bytes = sizeof(double) * ctx->cmp_n;
    const bool planes = age_p && angle_p;
    SC_HIP(ctx, hipMemcpyAsync(ctx->cmp_in[0].p, amp, bytes, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(ctx->cmp_in[1].p, snr, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (planes) {
        int rc;
        for (int k = 2; k < 4; ++k)
            if ((rc = sc_ensure(ctx, ctx->cmp_in[k], bytes))) return rc;
        SC_HIP(ctx, hipMemcpyAsync(ctx->cmp_in[2].p, age_p, bytes, hipMemcpyHostToDevice, ctx->stream));
        SC_HIP(ctx, hipMemcpyAsync(ctx->cmp_in[3].p, angle_p, bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    int rc = launch_compare_fold(ctx, age, angle, planes);
    if (rc) return rc;
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SC_OK;
}

extern "C" int sc_compare_fold(sc_ctx* ctx, const double* amp, const double* snr,
                               double age, double angle) {
    return compare_fold_impl(ctx, amp, snr, nullptr, nullptr, age, angle);
}

extern "C" int sc_compare_fold_planes(sc_ctx* ctx, const double* amp, const double* age,
                                      const double* angle, const double* snr) {
    if (!age || !angle) return SC_ERR_INVALID;
    return compare_fold_impl(ctx, amp, snr, age, angle, 0.0, 0.0);
}

extern "C" int sc_compare_end(sc_ctx* ctx, double* amp, double* age, double* angle, double* snr) {
    if (!ctx || !amp || !age || !angle || !snr || ctx->cmp_n == 0) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    size_t bytes = sizeof(double) * ctx->cmp_n;
    double* out[4] = {amp, age, angle, snr};
    for (int k = 0; k < 4; ++k)
        SC_HIP(ctx, hipMemcpyAsync(out[k], ctx->cmp[k].p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->cmp_n = 0;
    return SC_OK;
}

// ---------------------------------------------------------------------------
// measurement
// ---------------------------------------------------------------------------
extern "C" int sc_profile(sc_ctx* ctx, int enable) {
    if (!ctx) return SC_ERR_INVALID;
    sc_prof_collect(ctx);
    ctx->prof = enable < 0 ? 0 : enable;
    for (int k = 0; k < SC_K_COUNT; ++k) {
        ctx->k_launches[k] = 0;
        ctx->k_brackets[k] = 0;
        ctx->k_sampled[k] = 0;
        ctx->k_ms[k] = 0.0;
    }
    return SC_OK;
}

extern "C" int sc_profile_get(sc_ctx* ctx, int kernel, long long* launches, double* total_ms) {
    if (!ctx || kernel < 0 || kernel >= SC_K_COUNT) return SC_ERR_INVALID;
    sc_prof_collect(ctx);
    if (launches) *launches = ctx->k_launches[kernel];
    if (total_ms) {
        // sampled mean duration times the number of launches
        double mean = ctx->k_sampled[kernel] ? ctx->k_ms[kernel] / (double)ctx->k_sampled[kernel] : 0.0;
        *total_ms = mean * (double)ctx->k_launches[kernel];
    }
    return SC_OK;
}

extern "C" const char* sc_kernel_name(int kernel) {
    static const char* names[SC_K_COUNT] = {"k_curv", "k_windows", "k_direct", "k_fwd_rows",
                                            "k_fwd_cols", "k_inv_cols", "k_inv_rows", "k_settle"};
    return (kernel >= 0 && kernel < SC_K_COUNT) ? names[kernel] : "?";
}

