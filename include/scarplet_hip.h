/*
 * scarplet_hip.h - C ABI of libscarplet_hip.so, the MI355X (gfx950) engine
 * behind scarplet's template-matching hot path.
 *
 * The reference (stgl/scarplet) is pure Python; the interface this library
 * replaces is the body of
 *
 *     scarplet/core.py:297-377   match_template()      (per-template kernel)
 *     scarplet/core.py:198-243   compare()             (running-best fold)
 *     scarplet/dem.py:68-107     _calculate_directional_laplacian()
 *     scarplet/WindowedTemplate.py:159-183, 498-520, 66-84, 257-304
 *                                template() / get_window_limits() /
 *                                get_err_mask() of the built-in plugins
 *
 * i.e. everything that runs once per (age, orientation) template.  The search
 * drivers above it (core.py:139-195 calculate_best_fit_parameters,
 * core.py:266-294 match) stay host code: they build the parameter grids,
 * turn every template into an sc_template descriptor and make ONE call.
 * INTEGRATION.md shows the ctypes binding a scarplet maintainer would add.
 *
 * Conventions
 *   - plain C types only; all host buffers are caller-owned, C order;
 *   - every call returns SC_OK (0) or a negative SC_ERR_* code and never
 *     throws; sc_last_error() gives the message for the last failure;
 *   - calls are synchronous unless the name ends in _async; an sc_ctx is
 *     bound to one GPU and is not thread-safe; distinct contexts are
 *     independent (one context per GPU / per process).
 */
#ifndef SCARPLET_HIP_H
#define SCARPLET_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SC_ABI_VERSION 9

#define SC_OK               0
#define SC_ERR_INVALID     -1   /* bad argument                                */
#define SC_ERR_HIP         -2   /* HIP runtime failure (see sc_last_error)     */
#define SC_ERR_NO_DEM      -3   /* sc_set_dem has not been called              */
#define SC_ERR_UNSUPPORTED -4   /* plan outside what the kernels are built for */
#define SC_ERR_COMM        -5   /* RCCL failure                                */

/* template kinds (WindowedTemplate.py classes) */
#define SC_KIND_SCARP   0       /* Scarp and the UpperBreak variants (l.87-304) */
#define SC_KIND_RICKER  1       /* Ricker / Channel (l.434-525)                 */
#define SC_KIND_WINDOW  2       /* any other plugin: W window uploaded by host  */

/* template flags */
#define SC_FLAG_NEGATE      1   /* W -> -W            (WindowedTemplate.py:254) */
#define SC_FLAG_ERR_XR_LE0  2   /* snr=0 where xr<=0  (WindowedTemplate.py:265) */
#define SC_FLAG_ERR_XR_GE0  4   /* snr=0 where xr>=0  (WindowedTemplate.py:302) */
#define SC_FLAG_NO_LIMITS   8   /* get_window_limits() all False (l.495-496)    */

/* matching method */
#define SC_METHOD_DIRECT 0      /* real-space sliding window                    */
#define SC_METHOD_FFT    1      /* overlap-save tiles, LDS FFTs                 */

#define SC_ID_NONE 0xFFFFFFFFu  /* best-id value of a cell no template has won  */

typedef struct sc_ctx sc_ctx;

/*
 * One (scale, age, orientation) template = one call of match_template()
 * in the reference (core.py:297).  All doubles are evaluated by the host with
 * numpy exactly as the reference evaluates them, so that the support W != 0
 * decided on the device in float64 is bit-for-bit the reference's.
 */
typedef struct sc_template {
    int32_t  kind;            /* SC_KIND_*                                      */
    int32_t  flags;           /* SC_FLAG_*                                      */
    double   cos_a, sin_a;    /* cos/sin of the template's alpha = -orientation */
    double   c, d;            /* window half-widths along xr / yr (l.61-64)     */
    double   p0, p1;          /* SCARP: 2*kt**1.5*sqrt(pi), 4*kt (l.177-178)
                                 RICKER: pi*f, unused (l.514-515)
                                 WINDOW: n = count(W != 0), sum(W**2)           */
    double   cc, sc2, ss;     /* curvature mix, dem.py:103-104:
                                 curv = cc*d2z_dx2 - sc2*d2z_dxdy + ss*d2z_dy2  */
    int32_t  ilo, ihi, jlo, jhi;     /* cells kept by get_window_limits()
                                        (l.66-84): ilo<=i<=ihi && jlo<=j<=jhi   */
    int32_t  pmin, pmax, qmin, qmax; /* support bounding box in offsets from
                                        the template centre (ny//2, nx//2)      */
    uint32_t id;              /* recorded in the best-id plane when it wins     */
    int32_t  window;          /* SC_KIND_WINDOW: slot given to sc_upload_window */
} sc_template;

/* Geometry of one sc_match call (computed by the host planner). */
typedef struct sc_plan {
    int32_t method;           /* SC_METHOD_*                                    */
    int32_t Ty, Tx;           /* FFT tile size (powers of two, 64..4096)        */
    int32_t Vy, Vx;           /* valid outputs per tile = T - support span      */
    int32_t nty, ntx;         /* tiles covering the core region                 */
    int32_t circ_y, circ_x;   /* axis handled by its own periodicity (T == n)   */
    int32_t Py, Qx;           /* max pmax / qmax over the batch                 */
    int32_t group;            /* templates per inverse-transform launch (>= 1)  */
} sc_plan;

/* ---- lifetime ---------------------------------------------------------- */
int  sc_abi_version(void);
/* Which sources this binary was compiled from: the first 16 hex digits of the SHA-256 over the
 * library's sources (the five .hip files of csrc, sc_internal.h, this header) in the Makefile's order, worked out
 * at build time - `make -C scarplet_amd/csrc print-build-id` prints the same string for the tree
 * at hand.  bench.py prints it in every line and __graft_entry__.build() compares the two: a
 * stale prebuilt binary is visible instead of silently benchmarked. */
const char* sc_build_id(void);
int  sc_device_count(void);
int  sc_create(int device, sc_ctx** out);
void sc_destroy(sc_ctx* ctx);
const char* sc_last_error(sc_ctx* ctx);

/*
 * Hand over the elevation block this context works on.
 *   z            ly x lx float64, rows gy0.. / columns gx0.. of the ny x nx DEM
 *                (global indices may lie outside [0,n): halo cells of a
 *                periodic DEM); replaces DEMGrid._griddata (dem.py:84)
 *   core         [cy0,cy1) x [cx0,cx1): global cells whose results this
 *                context owns
 *   wrap         1: the block IS the whole DEM (ly==ny, lx==nx, gy0==gx0==0)
 *                and neighbours are found modulo the DEM size
 *   xaxis,yaxis  centred cell coordinates (WindowedTemplate.py:50-53)
 * Computes the three curvature stencils of dem.py:88-101 on the device.
 */
int sc_set_dem(sc_ctx* ctx, const double* z, int ly, int lx, int gy0, int gx0,
               int ny, int nx, int cy0, int cy1, int cx0, int cx1,
               double dx, double dy, int wrap,
               const double* xaxis, const double* yaxis);

/*
 * What the device found in the block handed over last (one pass over it in HBM, part of
 * sc_set_dem / sc_set_dem_device):
 *   nan_cells  cells that are NaN.  One NaN turns every output of the reference NaN (its
 *              whole-grid FFTs spread it, core.py:349-363): the host layer answers such a DEM
 *              with the reference's maps and does not search it (scarplet_amd/core.py);
 *   hash2      128-bit fingerprint of the float64 bit patterns (two 64-bit words);
 *   unchanged  1: the block, its geometry and cell size equal what the context held before this
 *              hand-over - the curvature planes were kept, and so were the curvature spectra of
 *              the last search (option "spectra_mb"): the next scale of a multi-scale job, which
 *              the reference runs as one sl.match per scale on the same data
 *              (docs/source/examples/channels.ipynb).
 * Replaces two host passes over the DEM (np.isnan(z).any(), a hash of the values) per call.
 * Any pointer may be NULL.
 */
int sc_dem_info(sc_ctx* ctx, long long* nan_cells, unsigned long long* hash2, int* unchanged);

/* Same, with z already in device memory (used after sc_halo_exchange). */
int sc_set_dem_device(sc_ctx* ctx, const void* z_dev, int ly, int lx, int gy0,
                      int gx0, int ny, int nx, int cy0, int cy1, int cx0,
                      int cx1, double dx, double dy, int wrap,
                      const double* xaxis, const double* yaxis);

/* Explicit template window for SC_KIND_WINDOW (generic plugins): w is the
 * h x w float64 block W[ny//2+pmin .., nx//2+qmin ..]; returns the slot. */
int sc_upload_window(sc_ctx* ctx, const double* w, int h, int wd, int* slot);
/* Optional per-cell masks for generic plugins (ny x nx uint8, global):
 * get_window_limits() and get_err_mask() results; pass NULL to clear. */
int sc_set_masks(sc_ctx* ctx, int slot, const uint8_t* limits,
                 const uint8_t* err);
int sc_clear_windows(sc_ctx* ctx);

/*
 * Engine options, set explicitly (nothing is read from the environment):
 *   "kappa"    float32 resolution floor of the FFT epilogue in units of eps32
 *              (default 4; 0 switches the floor off; sc_internal.h sc_epi_floor)
 *   "variant"  alternative kernel paths kept for cross-checks in the tests:
 *              0 default, 1 paired-template chunks by the four-column LDS-parked kernel at
 *              every tile size, 2 inverse column pass by the four-column kernels throughout
 *              (no wave-per-column kernel), 5 no paired-template mode, 6 inverse
 *              column pass as two launches per tile pair (own columns, mirrors) instead
 *              of one launch with the two kinds paired per XCD, 7 template spectra by the
 *              separate column-transform and split kernels, 8 complex-spectrum I1 for
 *              symmetric templates, 9 generic row kernel at every tile size, 10 the round-2
 *              real-space kernel (walks the support box; one orientation per launch), 11 the
 *              real-space kernel with T3 accumulated tap by tap on every row (no sum shared
 *              between a lane's adjacent outputs), 12 no paired orientations (one-tile searches
 *              with one template per orientation then leave half of every transform empty),
 *              13 the inverse passes store and transform every valid tile row (default: rows
 *              that a template's window limits mask are skipped for that template), 15 the row
 *              pass of small grids folds a launch's transforms in ONE workgroup per row (default:
 *              dealt out over up to four, the shares merged in order), 16 the real-space kernel's
 *              256 x 16 patch also where the 512 x 16 patch would be taken, 17 the orientation's
 *              curvature plane written by a pass of its own and read back by the forward row pass
 *              (default: mixed from the three stencil planes inside that pass - the same bits), 18 the
 *              inverse column pass at column length 512 by the four-column kernels (default: half a wave
 *              per column, k_inv_cols_h2 - the same bits), 19 paired orientations on k_inv_cols_h2 too,
 *              20 at most 64 templates per row-pass launch (default: the dealt-out row pass of small grids
 *              takes up to 255 in shares of at most 64 transforms - the same bits)
 *   "batch"    1 (default): searches whose single orientation does not fill the
 *              chip send several orientations through every launch; 0: one
 *              orientation per launch sequence.  Results are bit-identical.
 *   "batch_fill"  column-pass workgroups a batched launch sequence aims at (0: the default, 4096):
 *              orientations per launch = batch_fill / (tile pairs x Tx / 8), at most 64 and what the
 *              row pass's tables hold.  Results are bit-identical whatever the batch.
 *   "i1_pairs" tile pairs per launch of the wave-per-column inverse pass (default 2; 1: one pair per
 *              launch), interleaved so that the workgroups which stream the same template
 *              coefficients run on one XCD at the same time.  Results are bit-identical.
 *   "near_window"  > 0: the FFT row pass flags near-ties within this relative window (sc_get_near_ties); needs
 *              the fast row kernel (tile widths 512 / 1024 / 2048) and templates without per-cell masks, else
 *              sc_match answers SC_ERR_UNSUPPORTED.  Default 0.
 *   "split_i1" 1 (default): a column pass whose workgroups do not fill the chip (a small DEM with many
 *              templates per orientation) deals its transforms out over up to eight workgroups per column
 *              block; 0: one workgroup per column block walks all of them.  Results are bit-identical.
 *   "split_fill"  waves the dealt-out row pass of small grids may come to (0: the default, 4300 - four
 *              per SIMD and 5 %; round 4: 2048).  Results are bit-identical.
 *   "y_gb"     memory budget of the column -> row pass hand-off buffers in GB
 *              (0: a quarter of the free memory, at most 32)
 *   "sib"      sibling rendezvous (bit 0: row pass): the two workgroups that read the two
 *              halves of the same 128-byte hand-off lines keep within one template of each
 *              other so that the second read hits the XCD's L2.  Default 0: it moves 10 % fewer
 *              bytes and takes 3 % longer (profiles/r03_sibling_rendezvous.txt).  Results are
 *              bit-identical either way.
 *   "spectra_mb"  > 0: a search whose orientations' curvature spectra fit this many MiB keeps
 *              them (one slot per orientation instead of one per batched orientation), and a
 *              later sc_match on the same DEM with the same tile plan and the same orientations
 *              - the next scale of a multi-scale job, the reference's one sl.match per scale
 *              (docs/source/examples/channels.ipynb) - skips the curvature passes.  The spectra
 *              are the same bits either way, so are the results.  Setting the option (to any
 *              value) and loading a DEM drop what was kept.  Default 0 in the library; the
 *              Python host sets 8192.
 */
int sc_set_option(sc_ctx* ctx, const char* name, double value);

/*
 * One pass of the nodata fill that precedes the matcher (DEMGrid._fill_nodata,
 * dem.py:388-414: rasterio.fill.fillnodata -> GDALFillNodata).  z: ny x nx
 * float64 host array, NaN = nodata, filled in place the way GDAL's second pass
 * does it (alg/rasterfill.cpp as restated in oracle/scarplet_oracle.py
 * fill_nodata_pass): float32 work values (valid cells come back rounded through
 * float32 as well), per nodata cell the nearest valid cell of each quadrant over
 * the columns x -+ step, step <= floor(max_search_distance) - left quadrants from
 * step 0, right quadrants from step 1, columns clamped at the raster edge - and
 * the mean weighted by 1 / distance over the quadrants within
 * max_search_distance; then smoothing_iterations 3x3 means over the filled cells.
 * *remaining = cells still nodata (no source within reach); the host repeats with
 * a new distance as the reference does.  Independent of the DEM held by the
 * context.  PARITY UNPINNED: GDAL is not available to check against and no
 * GDAL-written fixture exists.
 */
int sc_fill_nodata(sc_ctx* ctx, double* z, int ny, int nx, double max_search_distance,
                   int smoothing_iterations, long long* remaining);

/* ---- the hot path ------------------------------------------------------ */
/* Zero the running-best record (compare() start state, core.py:222-225). */
int sc_reset_best(sc_ctx* ctx);

/* Match n templates and fold them, in the order given, into the running
 * best record (snr, amp, id), float32 per cell.  Fold rule (compare(),
 * core.py:227-240, as far as it is meaningful in float32):
 *   - a template takes a cell when its SNR is strictly greater than the
 *     record's; cells it masks (window limits, error mask) score 0 and never
 *     take a cell;
 *   - an exact SNR tie KEEPS THE INCUMBENT.  The reference's two strict
 *     compares zero the record on a tie; in its float64 arithmetic that is a
 *     rounding accident, in float32 it is systematic (an even or odd template
 *     gives bit-identical SNR at -pi/2 and +pi/2) and the zeroed record would
 *     be overtaken by an arbitrary later template.  The literal float64 rule,
 *     ties and sticky NaNs included, is sc_compare_*;
 *   - the order of the fold is the order of t[]; it matters on ties only;
 *   - NaN cannot arise: sc_set_dem's elevations must be finite (the host layer
 *     answers a DEM with NaNs the way the reference does, without the device).
 * Templates with equal (cc, sc2, ss) share one curvature plane; send them
 * adjacent. */
int sc_match(sc_ctx* ctx, const sc_template* t, int n, const sc_plan* plan);
int sc_match_async(sc_ctx* ctx, const sc_template* t, int n,
                   const sc_plan* plan);
int sc_sync(sc_ctx* ctx);

/* amp / snr maps of ONE template over the core region (match_template(),
 * core.py:377), float32, (cy1-cy0) x (cx1-cx0). */
int sc_match_template(sc_ctx* ctx, const sc_template* t, const sc_plan* plan,
                      float* amp, float* snr);

/* Copy the running best of the core region to the host. */
int sc_get_best(sc_ctx* ctx, float* amp, float* snr, uint32_t* id);

/* The running best as the reference returns it (core.py:243, 194): four float64
 * planes [amp, age, angle, snr] of the core region, out = 4 x (cy1-cy0) x
 * (cx1-cx0) doubles.  param_of_id / angle_of_id map a template id (< n_ids) to
 * its (age, orientation); cells no template has won are all zero. */
int sc_get_result(sc_ctx* ctx, const double* param_of_id, const double* angle_of_id,
                  int n_ids, double* out);

/*
 * compare() for results that already sit on the host (core.py:198-243), e.g.
 * user code that calls match_template() per orientation and folds itself.
 * float64, literally best = (best_snr > snr)*best + (best_snr < snr)*this for
 * the four planes, snr last.  Independent of the DEM state.
 */
int sc_compare_begin(sc_ctx* ctx, int ny, int nx);
int sc_compare_fold(sc_ctx* ctx, const double* amp, const double* snr,
                    double age, double angle);
/* The same step for a result whose age and angle are per-cell planes - the
 * output of an earlier fold, as match() feeds calculate_best_fit_parameters'
 * (4, ny, nx) arrays to compare() (core.py:288-292). */
int sc_compare_fold_planes(sc_ctx* ctx, const double* amp, const double* age,
                           const double* angle, const double* snr);
int sc_compare_end(sc_ctx* ctx, double* amp, double* age, double* angle,
                   double* snr);

/* Directional curvature of the block (dem.py:68-107), float32 ly x lx. */
int sc_curvature(sc_ctx* ctx, double cc, double sc2, double ss, float* out);
/* The same as the reference's data object returns it - float64, dem.py:103-104's expression in
 * numpy's evaluation order: out = d2z_dx2 * cos2 - 2 * d2z_dxdy * sin_a * cos_a + d2z_dy2 * sin2,
 * cos2 = cos(alpha)**2 and sin2 = sin(alpha)**2 evaluated by the caller.  Replaces
 * CalculationMixin._calculate_directional_laplacian (dem.py:68-107) and _calculate_laplacian
 * (dem.py:62-66) behind scarplet_amd.dem.DEMGrid's methods of those names. */
int sc_curvature_f64(sc_ctx* ctx, double cos2, double sin_a, double cos_a, double sin2,
                     double* out);

/* Float32 resolution of the FFT path on THIS surface, measured by the searches since the last
 * sc_reset_best: *wins = cells a template of the FFT path won, *near_floor = those whose residual
 * T3 - T1 (what the SNR divides by, core.py:362-366) lies within 256 x the transforms' float32
 * resolution floor (sc_internal.h sc_epi_floor) - their SNR is off by more than the stated
 * tolerance and their argmax is rounding noise.  ~0 on DEMs with a noise floor of their own
 * (lidar, the benchmark DEM), tens of per cent on synthetic surfaces stored without one; the
 * real-space path has no such limit.  scarplet_amd.match(method="auto") reads it to fall back. */
int sc_get_resolution_stats(sc_ctx* ctx, long long* wins, long long* near_floor);

/* Near-ties of the FFT searches since the last sc_reset_best, one byte per core cell, (cy1-cy0) x (cx1-cx0):
 * 1 where some template scored within the relative window of option "near_window" of the cell's running best
 * (either side of it; equal scores included since ABI 7).  A float32 FFT convolution carries an SNR error of up to half the
 * path's tie window (scarplet_amd: oracle-measured): between two templates closer than
 * that, which one the record holds is rounding noise.  The real-space path flags the same way with the option on - and
 * equal scores as well (its per-cell float32 sums can give two templates a rounding apart the same bits).
 * scarplet_amd.match(..., exact=True) settles the flagged cells in float64 (sc_get_near_events + sc_score_pairs_f64;
 * or, where the event list overflowed, a real-space search of them + sc_score_cells_f64): the argmax of every cell
 * is then the float64 reference's.  All zero when the option is 0 (the default: flags and events cost the row pass 13 %). */
int sc_get_near_ties(sc_ctx* ctx, uint8_t* out);

/* match_template() at single cells in FLOAT64 - core.py:297-377 as the real-space closed form, the template
 * evaluated with the reference's float64 expressions, the curvature from the float64 elevations (dem.py:88-104) -
 * for the m cells given (global row, column pairs) and EVERY template of the last sc_match in this context, in
 * the order they were handed over: amp, snr = m x n doubles each, masks applied (core.py:369-375).  The last step
 * of scarplet_amd.match(..., exact=True): the cells where two templates lie inside the float32 paths' own rounding
 * are settled the way the reference settles them.  Built-in templates only (SC_ERR_UNSUPPORTED otherwise); the
 * context must hold the cells' neighbourhoods (a whole DEM does).  n_templates: what the caller sized amp / snr for -
 * SC_ERR_INVALID unless it is the number of templates of that last sc_match (ABI 8). */
int sc_score_cells_f64(sc_ctx* ctx, const int32_t* cells, int m, int n_templates, double* amp, double* snr);

/* The near-ties of the searches since the last sc_reset_best as EVENTS (round 5; ABI 8: FOUR 32-bit words each, both paths) -
 * the cell (index into the core planes, row-major), the id of the template that was being scored, the id of the template
 * that held the cell's record at that moment (SC_ID_NONE: none yet), the float32 bits of the LARGER of their two scores -
 * one per (cell, template) whose score came within option "near_window" of the record, either side.  The true float64
 * argmax of a flagged cell is the record's final holder or one of the templates its events name (two templates further
 * apart than the window differ by more than twice the path's error: the lower one cannot be the argmax), and only events
 * whose larger score lies within the window of the FINAL record can name it: sc_settle_exact scores exactly those (cell,
 * template) pairs in float64.  *n_events = events recorded; the device list holds two per core cell (a million at least).
 * A count above `capacity` copies nothing and returns SC_OK (the caller asks again with room); a list that OVERFLOWED on
 * the device answers SC_ERR_UNSUPPORTED (ABI 8: said by the call, not left to the caller's arithmetic) - the caller takes
 * the route without events (sc_get_near_ties + sc_score_cells_f64). */
int sc_get_near_events(sc_ctx* ctx, uint32_t* events, long long capacity, long long* n_events);

/* sc_score_cells_f64 for (cell, template) PAIRS: pair k = global cell (cells[2k], cells[2k+1]) against template
 * templates[k] of the last sc_match in this context (its index in hand-over order); amp, snr = m doubles each. */
int sc_score_pairs_f64(sc_ctx* ctx, const int32_t* cells, const int32_t* templates, int m, double* amp, double* snr);

/*
 * exact=True on the device (round 6, ABI 8): the reference's fold is an argmax over float64 SNR maps (compare(),
 * core.py:230-240); this call makes the record's (age, orientation) of every near-tie cell that argmax.  After an sc_match
 * with option "near_window" on (either path lists its near-ties since ABI 8): the flagged cells become slots in cell
 * order, every cell's candidates - the record's final holder and the templates named by its events (those whose larger score
 * the final record has left behind by more than the window are dropped: neither template can be the argmax) - become a list, exactly
 * those (cell, template) pairs are scored in float64 (sc_score_pairs_f64's arithmetic, one workgroup per pair; a template
 * named twice in a list, or a list of one template, is not scored), and every cell takes the largest float64 SNR, ties to
 * the earlier template of the hand-over order.  The winner's id goes into the record (amp and snr rounded to float32: what
 * sc_get_best, sc_gather_result and sc_fold_ranks then see) and its float64 (amp, snr) are kept as patches that
 * sc_get_result lays over the converted planes - until the next sc_match or sc_reset_best.  No host pass over the planes:
 * three 8-byte read-backs size the buffers.
 *   n_twin    the last n_twin templates of the search are one CLASS with its first n_twin (the orientation grid's two ends
 *             are one template for the symmetric built-ins: +pi/2 against -pi/2, core.py:173-175 - their float64 SNRs
 *             differ by rounding noise, one maximum by the parity policy); 0: none.  One member of a class is scored per
 *             cell - the record's holder where its class is named - as itself (its amplitude carries its own sign)
 *   max_work  > 0: nothing is scored when pairs x the largest support box exceeds it (SC_ERR_UNSUPPORTED)
 *   stats     8 values: flagged cells, pairs listed, pairs scored, cells scored, cells whose template changed, events,
 *             0, taps the scores weighed
 * SC_ERR_UNSUPPORTED also when the event list overflowed (the caller takes a longer route: sc_get_near_ties +
 * sc_score_cells_f64) and for templates with host-uploaded windows.
 */
int sc_settle_exact(sc_ctx* ctx, int n_twin, double max_work, long long* stats);

/*
 * exact=True for an ORIENTATION-SHARDED search (scarplet_amd.dist.OrientationMatcher; the reference's pool over orientations,
 * core.py:180-183, whose compare() folds float64 maps).  Every rank searched its share of the templates with "near_window"
 * on and sc_fold_ranks made every record the fold of all - a near-tie between templates of two ranks is in no rank's list.
 * But a template that can be the float64 argmax scores, in float32, within the window of the FOLDED record, and the rank that
 * matched it knows: it is named by one of that rank's events whose larger score lies within the window of the folded record,
 * or it held the rank's own record.  The sequence, the same on every rank:
 *   sc_match (near_window on) -> sc_snapshot_best -> sc_fold_ranks (or sc_get_best / host fold / sc_set_best)
 *   -> sc_rank_candidates -> sc_exchange_candidates (RCCL; or the launcher's transport concatenates the ranks' pair
 *   lists, in rank order) -> sc_settle_pairs with the descriptors of the WHOLE search.
 * Every rank scores the same pairs with the same float64 arithmetic: the records agree bit for bit without a further
 * collective, and equal what sc_settle_exact leaves in a single context that searched all the templates wherever the
 * float64 argmax is concerned (tests/test_gpu_exact.py).
 */
/* the record's (snr, id) planes as they stand, kept on the device (call it BEFORE the fold) */
int sc_snapshot_best(sc_ctx* ctx);
/* upload a record - core cells, row-major: amplitude, SNR, template id - as this context's running best (the host
 * backend's fold, or a record saved earlier); patches of an earlier settle are dropped */
int sc_set_best(sc_ctx* ctx, const float* amp, const float* snr, const uint32_t* id);
/* this rank's candidates against the folded record: (core cell index, template id) pairs, 2 x uint32 each.  *n_pairs is
 * the number found; they are copied to `pairs` when capacity (in pairs) holds them - call with capacity 0 to size the
 * buffer (the list is kept on the device in between).  SC_ERR_UNSUPPORTED when the event list overflowed. */
int sc_rank_candidates(sc_ctx* ctx, uint32_t* pairs, long long capacity, long long* n_pairs);
/* settle the union of all ranks' candidates: t[0..n) = the descriptors of the WHOLE search in fold order (they become the
 * context's template table; nothing is matched), pairs as sc_rank_candidates wrote them (ids = sc_template.id);
 * n_twin, max_work, stats as sc_settle_exact. */
int sc_settle_pairs(sc_ctx* ctx, const sc_template* t, int n, const uint32_t* pairs, long long n_pairs, int n_twin,
                    double max_work, long long* stats);
/* the exchange on the devices: after sc_rank_candidates (capacity 0 will do: the list stays on the device) every rank's
 * list becomes one list on every device - two ncclAllGather over RCCL/xGMI: the counts, then slots of the largest count in
 * rank order, short lists padded with cells no DEM has.  *n_union = pairs in it, padding included; sc_settle_pairs with
 * pairs == NULL and n_pairs == *n_union settles it.  Collective: all ranks of the communicator call it - a rank whose
 * sc_rank_candidates failed (an overflowed event list) too: it takes part in the counts' all-gather with a marker and every rank
 * returns SC_ERR_UNSUPPORTED, none is left waiting.  No communicator: the union is the rank's own list. */
int sc_exchange_candidates(sc_ctx* ctx, long long* n_union);

/* Per-template scalars of the last sc_match / sc_match_template call:
 * n = count(W != 0) + eps (core.py:350) and sum(W**2) (core.py:356). */
int sc_get_template_sums(sc_ctx* ctx, int n, double* n_out, double* ts_out);

/* ---- measurement ------------------------------------------------------- */
#define SC_K_CURV        0
#define SC_K_WINDOWS     1
#define SC_K_DIRECT      2
#define SC_K_FWD_ROWS    3
#define SC_K_FWD_COLS    4
#define SC_K_INV_COLS    5
#define SC_K_INV_ROWS    6
#define SC_K_SETTLE      7      /* sc_settle_exact: all its kernels as one bracket */
#define SC_K_COUNT       8
/* HIP-event timing of every launch on the context's stream. */
int sc_profile(sc_ctx* ctx, int enable);
int sc_profile_get(sc_ctx* ctx, int kernel, long long* launches,
                   double* total_ms);
const char* sc_kernel_name(int kernel);
/* device memory currently held by the context, bytes */
size_t sc_device_bytes(sc_ctx* ctx);

/* ---- multi-GPU: RCCL halo exchange (one process per GPU) ---------------- */
#define SC_COMM_ID_BYTES 128
int sc_comm_unique_id(void* id_out /* SC_COMM_ID_BYTES */);
int sc_comm_init(sc_ctx* ctx, const void* id, int rank, int nranks);
/*
 * One rectangle of the halo exchange, in cells of this rank's halo-extended
 * block.  The host (scarplet_amd/dist.py) derives the list for every rank from
 * the tile grid; all ranks walk the same global order, so sends and receives
 * between a pair of ranks match up.
 */
#define SC_XFER_RECV  0   /* receive h x w cells from `peer` into (dy0, dx0)       */
#define SC_XFER_SEND  1   /* send the h x w cells at (sy0, sx0) to `peer`          */
#define SC_XFER_LOCAL 2   /* copy (sy0, sx0) -> (dy0, dx0) inside the block
                             (periodic image of the rank's own core)              */
typedef struct sc_xfer {
    int32_t peer, kind;
    int32_t sy0, sx0, dy0, dx0, h, w;
} sc_xfer;

/*
 * Assemble this rank's halo-extended elevation block on the device.
 *   core      this rank's own cells, core_h x core_w float64 (host)
 *   h*_lo/hi  halo cells around the core; the block is
 *             (hy_lo + core_h + hy_hi) x (hx_lo + core_w + hx_hi)
 *   x, n      the rank's transfers; sends/receives run as one grouped
 *             ncclSend/ncclRecv over xGMI, halo rectangles are packed into a
 *             contiguous staging buffer first
 * On return *z_dev is the float64 device block (owned by the context), ready
 * for sc_set_dem_device.
 */
int sc_halo_exchange(sc_ctx* ctx, const double* core, int core_h, int core_w,
                     int hy_lo, int hy_hi, int hx_lo, int hx_hi,
                     const sc_xfer* x, int n, void** z_dev);
/*
 * Final gather of a tiled search (results are disjoint rectangles): every rank
 * sends the float32 record of its core - amplitude, SNR, template id: 12 bytes
 * per cell, three grouped ncclSend - to `root` over RCCL; root receives all ranks
 * at once, converts each record like sc_get_result (the id tables are the same on
 * every rank) and places the four float64 planes in out = 4 x ny x nx doubles (host).
 *   cores   nranks x 4 ints: [cy0, cy1, cx0, cx1) of every rank, same on all ranks
 *   out     root only (ignored elsewhere)
 * Collective: all ranks of the communicator call it.  Without a communicator
 * (single context) it is sc_get_result into the core's place.
 */
int sc_gather_result(sc_ctx* ctx, int root, const int32_t* cores, int ny, int nx,
                     const double* param_of_id, const double* angle_of_id, int n_ids,
                     double* out);
/*
 * Fold of an orientation-sharded search: every rank holds the WHOLE DEM and searched its share
 * of the templates, numbered globally in fold order (the reference's own parallelism: a pool
 * over orientations, core.py:180-183, folded by compare(), core.py:198-243).  On return every
 * rank's running-best record is the fold of all ranks' records: per cell the greatest SNR
 * wins, equal SNRs go to the smaller id (the earlier template of the fold order - what a
 * single context folding all templates keeps), a NaN SNR beats every number (sc_match's
 * sticky NaN).  Two all-reduces over RCCL/xGMI: ncclMax on the 64-bit key
 * (SNR bits << 32 | ~id), ncclSum on the amplitude (zeroed on the ranks that lost the cell).
 * Collective: all ranks of the communicator call it, with the same core.  No communicator:
 * nothing to do.
 */
int sc_fold_ranks(sc_ctx* ctx);
/* What RCCL reports for this context's communicator: ncclCommCount, ncclCommUserRank,
 * ncclCommCuDevice, and the PCI bus id of the context's device (bus_id: at least 16 bytes, may
 * be NULL).  Without a communicator *nranks = 0, *rank = -1. */
int sc_comm_info(sc_ctx* ctx, int* nranks, int* rank, int* device, char* bus_id, int bus_id_len);
int sc_comm_destroy(sc_ctx* ctx);

/* (The TIFF LZW decoder of the GeoTIFF reader moved to libscarplet_host.so in round 3:
 * include/scarplet_host.h - reading a DEM must not need the HIP and RCCL runtimes.) */

#ifdef __cplusplus
}
#endif
#endif /* SCARPLET_HIP_H */
