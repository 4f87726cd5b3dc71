// Internal declarations shared by the translation units of libscarplet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <map>
#include <limits.h>
#include "../../include/scarplet_hip.h"

#define SC_EPS 2.220446049250313e-16      // np.spacing(1), core.py:340
// a near-tie event: cell, id of the template scored, id of the record's holder, float32 bits of the LARGER of the two scores
// (round 6: the settle drops an event whose scores the final record has left behind by more than the window)
#define SC_EVENT_WORDS 4

// Timing-only ablation bits (skip loads / transforms / stores of a kernel; the
// results are wrong while one is set).  They exist only in a -DSC_ABLATE build
// (tools/ablate.sh); in the shipped library the tests fold to constants.
#ifdef SC_ABLATE
#define SC_DBGBIT(d, b) (((d) & (b)) != 0)
#else
#define SC_DBGBIT(d, b) false
#endif
#define SC_EXP_UNDERFLOW 745.1332191019412 // exp(-u) != 0  <=>  u < 1075 ln 2

// Device-side view of one template of the current batch.
struct TemplDev {
    int32_t kind, flags;
    double cos_a, sin_a, c, d, p0, p1;
    int32_t ilo, ihi, jlo, jhi;
    int32_t pmin, pmax, qmin, qmax;
    uint32_t id;
    int32_t wh, ww;            // window (bbox) height / width
    long long win_off;         // element offset of the window in win_w / win_m
    long long dwin_off;        // real-space path: element offset of the reversed, padded (w, m) rows in dwin
    int32_t dpitch, span_off;  //   their pitch (a multiple of 4) and the first entry of the row-span table
    const uint8_t* mask_lim;   // optional explicit masks (generic plugins)
    const uint8_t* mask_err;
};

// Geometry of the block held by a context, passed by value to kernels.
struct Geom {
    int ly, lx;        // local block
    int gy0, gx0;      // global index of local (0,0)
    int ny, nx;        // DEM size
    int cy0, cy1, cx0, cx1;   // core
    int wrap;
    int oy, ox;        // ny % 2, nx % 2
};

struct FftGeom {
    int Ty, Tx, Vy, Vx, nty, ntx, circ_y, circ_x, Py, Qx;
    int ntiles;        // nty * ntx
};

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct WindowSlot {
    double l1 = 0.0;           // sum |W|
    float* w = nullptr;        // h x wd float32
    uint8_t* m = nullptr;      // h x wd
    int h = 0, wd = 0;
    uint8_t* mask_lim = nullptr;
    uint8_t* mask_err = nullptr;
};

// templates per inverse launch (sc_match batches an orientation run in chunks)
#define SC_MAX_GROUP 64
// templates one batched launch SEQUENCE can carry (forward passes and column pass of nb orientations x n
// templates each); the row pass folds them in launches of at most SC_MAX_GROUP, orientation slice after slice
#define SC_MAX_BATCH 256
#define SC_MAX_ORIENT 64         // orientations (curvature planes) one launch sequence can carry

struct sc_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    bool have_dem = false;
    Geom g{};
    double dx = 1, dy = 1;
    DevBuf z, xaxis, yaxis, A, B, C, curv;
    bool z_borrowed = false;
    const double* z_dev = nullptr;
    // digest of the elevation block (k_dem_digest): fingerprint, NaN cells; dem_sig = the geometry and cell
    // size it belongs to; dem_unchanged = the last sc_set_dem found the block the context already held
    unsigned long long dem_hash[2] = {0, 0};
    long long dem_nan = 0;
    double dem_sig[16] = {0};
    bool dem_unchanged = false;
    DevBuf digest;
    DevBuf best_snr, best_amp, best_id;
    DevBuf map_amp, map_snr;
    DevBuf cmp[4], cmp_in[4];   // sc_compare_*: amp, age, angle, snr (float64); inputs amp, snr, age, angle
    size_t cmp_n = 0;
    DevBuf res_stats;           // two counters: wins of the FFT path, wins near its float32 resolution floor
    DevBuf dwin, spans;         // real-space path: k_direct_prep's rows and row spans
    DevBuf templ, sums, wl1, norms, win_w, win_m;   // wl1: sum|W| per template; norms: per tile pair
    DevBuf norm_part;                               // k_fwd_rows_curv's partial sums of the norms
    // Curvature spectra kept across searches (sc_set_option "spectra_mb" > 0): uc / uc2 / norms hold one
    // slot per orientation of the search instead of one per batched orientation, and a later search on
    // the same DEM with the same plan and orientations - a multi-scale job is one search per scale -
    // finds them there and skips the curvature passes (k_curv_alpha, F1c, F2).
    double spec_mb = 0.0;
    int spec_slots = 0;                 // slots the buffers hold (0: not keeping)
    size_t spec_uc_stride = 0;          // float2 elements of uc / uc2 per slot; norms: 2 * npairs doubles
    size_t spec_norm_stride = 0;
    size_t uc_off = 0, norms_off = 0;   // the current chunk's first slot, in elements of uc / uc2 and of norms
    std::vector<double> spec_key;       // (cc, sc2, ss) per slot, NaN while empty
    long long spec_sig[24] = {0};       // tile plan and block geometry the slots were computed for
    DevBuf tw_y, tw_x;
    int tw_Ty = 0, tw_Tx = 0;
    int fft_pb = 1;            // tile pairs per inverse launch (fft_prepare)
    DevBuf blk, uc, uc2, vh, wh, mh, yw, ym, tiles;
    std::vector<WindowSlot> windows;
    // host copies of what a search uploads asynchronously (descriptors, sums, tile list): they must outlive
    // the copy, so they live here - no stream synchronisation between the upload and the launches
    std::vector<TemplDev> h_templ;
    std::vector<double> h_sums, h_wl1;
    std::vector<unsigned char> h_tiles;  // the tile list the device holds (bytes), to skip its re-upload
    bool async_in_flight = false;        // an sc_match_async has not been followed by sc_sync yet
    int last_batch = 0;
    float kappa = 4.f;         // float32 resolution floor of the FFT path, in units of eps (sc_set_option "kappa")
    float near_w = 0.f;        // sc_set_option "near_window": the FFT row pass flags near-ties (sc_get_near_ties)
    float near_w_used = 0.f;   // the window the events of the current record were listed with (the last sc_match's, when on)
    DevBuf near;               // one byte per core cell
    DevBuf near_ev;            // the near-tie events of the FFT row pass: a 64-bit count, then 3 words per event (sc_get_near_events)
    // sc_settle_exact: slot of every flagged cell (one word per core cell, valid at flagged cells), the work lists, and the
    // patches - per flagged cell its index and the float64 (amp, snr, id) sc_get_result writes over the converted record;
    // patch_n: how many the current record carries (0 after any sc_match / sc_reset_best)
    DevBuf st_slot, st_work, st_pairs, st_patch, st_spans;
    DevBuf snap;               // sc_snapshot_best: the record's (snr, id) planes as they stood (before sc_fold_ranks)
    size_t snap_cells = 0;
    long long cand_n = -1;      // sc_rank_candidates: pairs of the list it left on the device (st_pairs), -1: none
    DevBuf xch, xch_cnt;        // sc_exchange_candidates: all ranks' lists (slots of equal size, padded with invalid cells), their counts
    long long xch_n = -1;       // pairs in xch, padding included; -1: nothing exchanged
    size_t patch_n = 0;
    DevBuf score;              // sc_score_cells_f64: the cell list and the two float64 outputs
    DevBuf score_w;            // ... and the templates' float64 windows (offsets, then the windows)
    DevBuf score_abc;          // ... and the three stencil planes of the block in float64 (rebuilt at every call)
    bool templ_windows = false;   // the last sc_match carried host-uploaded windows (no float64 form on the device)
    int batch_templ = 0;       // sc_set_option "batch_templ": templates one batched launch sequence may carry (0: SC_MAX_BATCH)
    int split_i1 = 1;          // sc_set_option "split_i1": under-filled column passes deal their transforms out along grid.z
    long long split_fill = 0;  // sc_set_option "split_fill": waves a dealt-out row pass may come to (0: 4096)
    int variant = 0;           // sc_set_option "variant": alternative kernel paths kept for cross-checks
    int batch_off = 0;         // sc_set_option "batch" = 0: no orientation batching (cross-check in the tests)
    int i1_pairs = 2;          // sc_set_option "i1_pairs": tile pairs per launch of the wave-per-column pass, interleaved (1: one pair per launch)
    int batch_fill = 0;        // sc_set_option "batch_fill": column workgroups a batched launch aims at (0: 4096)
    double y_gb = 0.0;         // sc_set_option "y_gb": memory budget of the I1 -> I2 hand-off (0: automatic)
    int sib = 0;               // sc_set_option "sib": sibling rendezvous, bit 0 row pass, bit 1 column pass (sc_fft.hip SibSync)
    DevBuf split_s, split_a, split_i;   // scratch records of a split row pass (k_inv_rows_fast SPLITK, k_merge_split)
    DevBuf sib_buf;
    uint32_t sib_epoch = 0;
    int dbg = 0;               // timing-only ablation bits; only an SC_ABLATE build reads them (tools/ablate.sh)
    // profiling
    int prof = 0;              // 0 off, k: time every k-th launch of a kernel
    int prof_cur = -1;         // kernel being bracketed (-1: not sampled)
    hipEvent_t prof_ev0 = nullptr, prof_ev1 = nullptr;
    long long k_launches[SC_K_COUNT] = {0};
    long long k_brackets[SC_K_COUNT] = {0};
    int prof_kernel = 0;
    std::vector<int> pending_n;
    long long k_sampled[SC_K_COUNT] = {0};
    double k_ms[SC_K_COUNT] = {0};
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    std::map<const void*, size_t> lds_attr;
    // comm
    void* comm = nullptr;      // ncclComm_t
    int rank = 0, nranks = 1;
    DevBuf halo_z, halo_stage;
    DevBuf res;                // sc_get_result: four float64 planes + the id tables
    DevBuf fill[4];            // sc_fill_nodata: z, result, up / down tables (+ the counter)
};

int sc_fail(sc_ctx* ctx, int code, const char* fmt, ...);
int sc_ensure(sc_ctx* ctx, DevBuf& b, size_t bytes);
size_t sc_total_bytes(sc_ctx* ctx);
// raise a kernel's dynamic-LDS limit once (cached per kernel)
int sc_lds_attr(sc_ctx* ctx, const void* kernel, size_t bytes);

#define SC_HIP(ctx, call)                                                     \
    do {                                                                      \
        hipError_t e__ = (call);                                              \
        if (e__ != hipSuccess)                                                \
            return sc_fail(ctx, SC_ERR_HIP, "%s: %s (%s:%d)", #call,          \
                           hipGetErrorString(e__), __FILE__, __LINE__);       \
    } while (0)

// profiling brackets around kernel launches
void sc_prof_begin(sc_ctx* ctx, int kernel);
void sc_prof_end(sc_ctx* ctx, int n = 1);
void sc_prof_collect(sc_ctx* ctx);
// bytes per cell of the running-best record as it travels between ranks: amp f32, snr f32, id u32
#define SC_RECORD_BYTES 12
int sc_launch_result(sc_ctx* ctx, const float* amp, const float* snr, const uint32_t* id, const double* tab_par,
                     const double* tab_ang, int n_ids, size_t n, double* planes);
// flag plane and event list of option "near_window" (allocated and cleared on first use; sc_reset_best clears them from then on)
// the descriptors of a search as the device's template table (TemplDev), without matching them: sc_settle_pairs
int sc_load_templates(sc_ctx* ctx, const sc_template* t, int n);
int sc_near_buffers(sc_ctx* ctx, unsigned long long** ev_count, uint32_t** ev, unsigned long long* ev_cap);
// sc_settle.hip: the patches of sc_settle_exact over four converted float64 planes of nc cells each
int sc_apply_patches(sc_ctx* ctx, const double* tab_par, const double* tab_ang, int n_ids, size_t nc, double* planes);
int sc_result_planes(sc_ctx* ctx, const double* param_of_id, const double* angle_of_id, int n_ids,
                     double** planes_out, size_t* nc_out);

// ---- launchers implemented in sc_kernels.hip --------------------------------
int launch_curv_planes(sc_ctx* ctx);
int launch_dem_digest(sc_ctx* ctx, unsigned long long* out_dev);
int launch_curv_alpha(sc_ctx* ctx, float cc, float sc2, float ss, int plane = 0);
int launch_curv_f64(sc_ctx* ctx, double c2, double sn, double cs, double s2, double* out_dev);
int launch_curv_alpha_batch(sc_ctx* ctx, const float (*coef)[3], int nb);
// tsel_dev: nullptr - every cell against every template (m x n_templ outputs); else pair k = (cell k, template tsel[k]) (m outputs)
int score_prepare_f64(sc_ctx* ctx, int n_templ, const unsigned long long** woff_out, const double** wbuf_out, const double** planes_out);
int launch_score_f64(sc_ctx* ctx, const int* cells_dev, const int* tsel_dev, int m, int n_templ, double* amp_dev, double* snr_dev);
int launch_windows(sc_ctx* ctx, int first, int n, int wh_max, int ww_max);
int launch_direct(sc_ctx* ctx, int first, int n, bool to_maps, int nb, int wh_max, int ww_max, bool long_runs);
bool direct_window_fits(int ww);
int launch_fill_nodata(sc_ctx* ctx, double* zdev, double* tmp, int* up, int* dn, int ny, int nx,
                       double maxd, int smoothing, unsigned long long* remaining_dev);     // template window width the real-space kernel can stage in LDS
int launch_compare_fold(sc_ctx* ctx, double age, double angle, bool planes);

// ---- launchers implemented in sc_fft.hip ------------------------------------
int fft_prepare(sc_ctx* ctx, const FftGeom& fg, int n_templ_chunk, int group, int nb, int n_slots);
void fft_spectra_forget(sc_ctx* ctx);
int fft_batch_orientations(const sc_ctx* ctx, const FftGeom& fg, int n_per, int group);
int fft_forward_curv(sc_ctx* ctx, const FftGeom& fg, int nb, const float (*coef)[3] = nullptr);
int fft_forward_templates(sc_ctx* ctx, const FftGeom& fg, int first, int n, int parity);
int fft_inverse_fold(sc_ctx* ctx, const FftGeom& fg, int first, int n,
                     int group, bool to_maps, bool full_masks, int parity, int nb);
bool fft_size_supported(int T);

// ---- device helpers shared by both paths -------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ int wrap_index(int g, int n) {
    g %= n;
    return g < 0 ? g + n : g;
}

// curvature at GLOBAL cell (gi, gj) of the block held by the context: modulo
// the DEM size when the block is the whole periodic DEM, otherwise relative to
// the block origin (cells outside the block only feed discarded outputs).
__device__ __forceinline__ float load_curv(const float* __restrict__ curv,
                                           const Geom& g, int gi, int gj) {
    int li, lj;
    if (g.wrap) {
        li = wrap_index(gi, g.ny);
        lj = wrap_index(gj, g.nx);
    } else {
        li = gi - g.gy0;
        lj = gj - g.gx0;
        if (li < 0 || li >= g.ly || lj < 0 || lj >= g.lx) return 0.f;
    }
    return curv[(size_t)li * g.lx + lj];
}

// Per-template scalars of the epilogue, prepared once per template.
struct EpiScal {
    float inv_ts;      // 1 / sum(W**2)                       (core.py:356)
    float inv_n;       // 1 / (count(W != 0) + eps)           (core.py:350)
    float d3;          // float32 resolution of T3 (FFT path; 0 on the exact path)
    float dx2;         // 2 * resolution of xcorr / sum(W**2)
    float dxx;         // resolution of xcorr squared / sum(W**2)
};
__device__ __forceinline__ EpiScal sc_epi_scalars(const double* __restrict__ sums, int it) {
    EpiScal s;
    s.inv_n = (float)(1.0 / (sums[2 * it] + SC_EPS));
    s.inv_ts = (float)(1.0 / sums[2 * it + 1]);
    s.d3 = s.dx2 = s.dxx = 0.f;
    return s;
}

// Resolution floor of the float32 transforms.  A length-N float32 FFT
// convolution returns every output with an ABSOLUTE error of order
// eps32 * |kernel|_1 * |data|_2 / sqrt(N): relative to the plane, not to the
// cell.  Where a DEM has no noise floor of its own (synthetic test surfaces
// with exactly flat regions) T3 and T1 of a far-away template both sink below
// that error, their difference is rounding noise and T1/(T3 - T1) explodes -
// the reference sees the same at 1e-16 and adds eps for it (core.py:365-366).
// So T3 - T1 is not allowed below the resolution of its two terms:
//   d3  for T3  = M * curv^2   (|M|_1 = n,       |curv^2|_2 over the tile pair)
//   dx  for xcorr = W * curv   (|W|_1,           |curv|_2)
// and T1 = xcorr^2 / ts carries 2|xcorr| dx / ts + dx^2 / ts.
// kappa scales eps32 (calibrated against the reference golden).
__device__ __forceinline__ void sc_epi_floor(EpiScal& s, double n, double ts, double l1,
                                             double norm_c2, double norm_c4, double cells,
                                             float kappa) {
    const double e = (double)kappa * 5.9604644775390625e-08;
    double d3 = e * n * sqrt(norm_c4 / cells);
    double dx = e * l1 * sqrt(norm_c2 / cells);
    s.d3 = (float)d3;
    s.dx2 = (float)(2.0 * dx / ts);
    s.dxx = (float)(dx * dx / ts);
}

// FFT path: W and M = (W != 0) ride in ONE complex transform (W + iM) and are
// separated afterwards as (v[f] +- conj v[-f])/2.  In float32 that difference
// is only as accurate as the LARGER of the two spectra, and sum(M^2) = n is
// 10^3..10^6 times sum(W^2); so W is scaled by alpha = sqrt(n / sum W^2) to
// the same energy before the transform and divided out again after the inverse transform.
__device__ __forceinline__ float sc_fft_alpha(const double* __restrict__ sums, int it) {
    double n = sums[2 * it], ts = sums[2 * it + 1];
    return (n > 0.0 && ts > 0.0) ? (float)sqrt(n / ts) : 1.0f;
}
// core.py:360-367 for one cell.  The reference forms
//   amp = xcorr/ts, T1 = ts*amp**2, error = (T1 - 2*amp*xcorr + T3)/n + eps,
//   snr = |T1/error|;  T1 - 2*amp*xcorr is -T1, so error = (T3 - T1)/n + eps.
// float32 here: xcorr and T3 arrive as float32 sums, and the cancellation in
// T3 - T1 amplifies THEIR rounding, not the epilogue's.
__device__ __forceinline__ void sc_epilogue(float xc, float t3, const EpiScal& s,
                                            float& amp_out, float& snr_out) {
    float amp = xc * s.inv_ts;
    float T1 = xc * amp;
    float d = fmaxf(t3 - T1, s.d3 + fabsf(xc) * s.dx2 + s.dxx);   // floor is 0 on the exact path
    float err = d * s.inv_n + (float)SC_EPS;
    amp_out = amp;
    snr_out = fabsf(T1 * __builtin_amdgcn_rcpf(err));      // v_rcp_f32, 1 ulp
}

// masks of core.py:369-375 for global cell (gi, gj)
__device__ __forceinline__ void sc_apply_masks(const TemplDev& t, const Geom& g,
                                               const double* __restrict__ xaxis,
                                               const double* __restrict__ yaxis,
                                               int gi, int gj, float& amp,
                                               float& snr) {
    if (t.flags & (SC_FLAG_ERR_XR_LE0 | SC_FLAG_ERR_XR_GE0)) {
        double xr = __dadd_rn(__dmul_rn(xaxis[gj], t.cos_a),
                              __dmul_rn(yaxis[gi], t.sin_a));
        bool m = (t.flags & SC_FLAG_ERR_XR_LE0) ? (xr <= 0.0) : (xr >= 0.0);
        if (m) snr = 0.f;
    }
    if (t.mask_err && t.mask_err[(size_t)gi * g.nx + gj]) snr = 0.f;
    bool keep = gi >= t.ilo && gi <= t.ihi && gj >= t.jlo && gj <= t.jhi;
    if (t.mask_lim && t.mask_lim[(size_t)gi * g.nx + gj]) keep = false;
    if (!keep) { amp = 0.f; snr = 0.f; }
}

// One step of compare() (core.py:230-240) on a (snr, amp, id) record:
//   best > this: keep;   best < this: take;   NaN on either side: the record
//   becomes (snr = NaN, amp = 0, no id) and stays so (0*x + 0*NaN).
// Exact ties: the reference's two strict compares zero the record.  In its
// float64 arithmetic a tie between two different templates is a rounding
// accident; in float32 on the GPU it is systematic (a template that is even
// or odd in xr gives bit-identical SNR at -pi/2 and +pi/2), and a zeroed
// record is then overtaken by whatever template comes next.  So a tie keeps
// the incumbent here - one of the two tied candidates, which is what the
// reference returns in all but the accidental case ("Parity").
// The literal rule is kept where it is exact: sc_compare_* (float64).
__device__ __forceinline__ bool sc_fold(float& b_snr, float& b_amp,
                                        uint32_t& b_id, float t_snr,
                                        float t_amp, uint32_t t_id) {
    if (b_snr < t_snr) {                          // take
        b_snr = t_snr; b_amp = t_amp; b_id = t_id;
        return true;
    }
    if (b_snr >= t_snr) return false;             // keep (ties included)
    if (b_snr != b_snr) return false;             // already NaN: sticky
    b_snr = __builtin_nanf("");                   // this is NaN
    b_amp = 0.f;
    b_id = SC_ID_NONE;
    return true;
}
#endif
