// sc_settle_exact: the float64 argmax of every near-tie cell, settled on the device (round 6).
//
// The reference folds float64 SNR maps (compare(), core.py:230-240: the argmax over templates of float64 numbers); the
// float32 searches here decide a cell to within their own rounding.  With option "near_window" on, both paths flag the
// cells where a template scored within the window of the running best and list an event (cell, template scored, holder of
// the record, the larger of their two scores) per near-tie.  With float32 scores off by at most e and a window of 2 e, the
// float64 argmax of a cell scores within the window of the record's FINAL holder, hence within the window of whatever held
// the record when it was scored or displaced: it is the final holder or a template named by an event whose larger score
// lies within the window of the final record (events the search has left behind are dropped).  This file turns the flags and the
// events into per-cell candidate lists, scores exactly those (cell, template) pairs with match_template()'s float64
// arithmetic (core.py:340-377 as the real-space closed form: k_st_score, the same expressions as k_score_f64), takes
// the argmax in fold order, writes it into the record and keeps the float64 (amp, snr) as patches that sc_get_result
// lays over the converted planes.  Round 5 did the list building on the host (1.2 GB of record copied out, np.argwhere
// over the flag plane, np.unique / np.lexsort over millions of keys): 1.9 s of the 5.1 s the exact C3 search took.
//
// Pipeline (all on the context's stream; three 8-byte read-backs size the next step's buffers):
//   k_st_flag_count / k_st_scan1 / k_st_slots   flagged cells -> slots, numbered 64 x 64 tile by tile (neighbouring pairs share
//                                               their curvature neighbourhood in L2), cnt[slot] = 1 (the final holder)
//   k_st_events<false>                          cnt[slot] += candidates the slot's events add (events left behind by the final
//                                               record: dropped - nine in ten on the benchmark search)
//   k_st_sum / k_st_scan1 / k_st_offsets        exclusive scan -> off[slot]
//   k_st_init_lists, k_st_events<true>          pair lists: entry 0 the final holder, then the events' templates
//   k_window_f64, k_curv_planes<double>         (score_prepare_f64); k_st_sums: n and sum(W**2) in a fixed order
//   k_st_spans                                  the windows' row runs (their supports are a seventh of their boxes)
//   k_st_score                                  one wave per pair; repeats of a template in a list and lists of one
//                                               template are not scored
//   k_st_resolve                                per slot the largest float64 SNR, ties to the earlier template
#include "sc_internal.h"
#include <algorithm>

namespace {

constexpr int ST_CH = 4096;                    // cells / counts per workgroup of the scans: 256 threads x 16
constexpr unsigned ST_STATS = 8;               // 64-bit counters: 0 flagged cells, 1 pairs listed, 2 pairs scored, 3 cells scored, 4 changed

__device__ __forceinline__ unsigned wave_incl_scan(unsigned v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned u = __shfl_up(v, d, 64);
        if ((int)(threadIdx.x & 63) >= d) v += u;
    }
    return v;
}

// exclusive scan of one value per thread over a workgroup of NW waves; *total = the workgroup's sum
template <int NW>
__device__ __forceinline__ unsigned block_excl_scan(unsigned v, unsigned* total) {
    __shared__ unsigned wsum[NW];
    const unsigned inc = wave_incl_scan(v);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) wsum[w] = inc;
    __syncthreads();
    unsigned base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const unsigned s = wsum[k];
        base += k < w ? s : 0u;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// Slots are numbered TILE by tile: workgroup b of the flag scans takes the 64 x 64 cells of tile (b / tiles_x, b % tiles_x),
// thread t the 16 cells of columns 16 (t & 3) .. of its row t >> 2.  Neighbouring slots are then neighbouring cells in BOTH
// directions, and the scorer's waves - which walk the curvature planes around their cells - find each other's lines in L2
// (row-major order: a thousand waves per XCD strung along one DEM row, 17 MB of planes between a line's first and second use).
struct TileMap {
    int ch, cw, tiles_x;
    __device__ __forceinline__ bool cell0(unsigned b, unsigned t, int& row, int& col) const {
        row = (int)(b / (unsigned)tiles_x) * 64 + (int)(t >> 2);
        col = (int)(b % (unsigned)tiles_x) * 64 + 16 * (int)(t & 3);
        return row < ch && col < cw;
    }
};

// the 16 flag bytes of a thread (flags are 0 or 1), as four words; cells beyond the row's end read 0
__device__ __forceinline__ uint4 flags16(const uint8_t* __restrict__ near, const TileMap& tm, unsigned b, unsigned t, size_t& c0) {
    int row, col;
    unsigned w[4] = {0, 0, 0, 0};
    c0 = 0;
    if (tm.cell0(b, t, row, col)) {
        c0 = (size_t)row * tm.cw + col;
        const int n = min(16, tm.cw - col);
        if (n == 16 && (c0 & 15) == 0) {
            const uint4 f = *reinterpret_cast<const uint4*>(near + c0);
            w[0] = f.x; w[1] = f.y; w[2] = f.z; w[3] = f.w;
        } else {
            for (int k = 0; k < n; ++k) w[k >> 2] |= (unsigned)(near[c0 + k] != 0) << (8 * (k & 3));
        }
    }
    return make_uint4(w[0] & 0x01010101u, w[1] & 0x01010101u, w[2] & 0x01010101u, w[3] & 0x01010101u);
}

__global__ void __launch_bounds__(256)
k_st_flag_count(const uint8_t* __restrict__ near, TileMap tm, unsigned* __restrict__ blk) {
    size_t c0;
    const uint4 f = flags16(near, tm, blockIdx.x, threadIdx.x, c0);
    unsigned tot;
    block_excl_scan<4>(__popc(f.x) + __popc(f.y) + __popc(f.z) + __popc(f.w), &tot);
    if (threadIdx.x == 0) blk[blockIdx.x] = tot;
}

// in-place exclusive scan of n workgroup sums by ONE workgroup of 1024 threads; *total = their sum
__global__ void __launch_bounds__(1024)
k_st_scan1(unsigned* __restrict__ blk, unsigned n, unsigned long long* __restrict__ total) {
    const unsigned per = (n + 1023) / 1024;
    const unsigned lo = min(n, threadIdx.x * per), hi = min(n, lo + per);
    unsigned s = 0;
    for (unsigned k = lo; k < hi; ++k) s += blk[k];
    unsigned tot;
    unsigned base = block_excl_scan<16>(s, &tot);
    for (unsigned k = lo; k < hi; ++k) {
        const unsigned v = blk[k];
        blk[k] = base;
        base += v;
    }
    if (threadIdx.x == 0) *total = tot;
}

__global__ void __launch_bounds__(256)
k_st_slots(const uint8_t* __restrict__ near, TileMap tm, const unsigned* __restrict__ blk, uint32_t* __restrict__ cell_of,
           uint32_t* __restrict__ slot_of, unsigned* __restrict__ cnt) {
    size_t c0;
    const uint4 f = flags16(near, tm, blockIdx.x, threadIdx.x, c0);
    unsigned tot;
    unsigned s = blk[blockIdx.x] + block_excl_scan<4>(__popc(f.x) + __popc(f.y) + __popc(f.z) + __popc(f.w), &tot);
    const unsigned w[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if ((w[k >> 2] >> (8 * (k & 3))) & 1u) {
            cell_of[s] = (uint32_t)(c0 + k);
            slot_of[c0 + k] = s;
            cnt[s] = 1u;                       // (the record's final holder)
            ++s;
        }
}

__global__ void __launch_bounds__(256)
k_st_sum(const unsigned* __restrict__ in, unsigned n, unsigned* __restrict__ blk) {
    const unsigned i0 = blockIdx.x * ST_CH + 16 * threadIdx.x;
    unsigned s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += i0 + k < n ? in[i0 + k] : 0u;
    unsigned tot;
    block_excl_scan<4>(s, &tot);
    if (threadIdx.x == 0) blk[blockIdx.x] = tot;
}

// out[i] = exclusive prefix of in[0 .. n) (out has n + 1 entries: the last is the total)
__global__ void __launch_bounds__(256)
k_st_offsets(const unsigned* __restrict__ in, unsigned n, const unsigned* __restrict__ blk, unsigned* __restrict__ out) {
    const unsigned i0 = blockIdx.x * ST_CH + 16 * threadIdx.x;
    unsigned v[16], s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { v[k] = i0 + k < n ? in[i0 + k] : 0u; s += v[k]; }
    unsigned tot;
    unsigned base = blk[blockIdx.x] + block_excl_scan<4>(s, &tot);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (i0 + k <= n) out[i0 + k] = base;
        base += v[k];
    }
}

// template id -> its index in the last search's hand-over order (-1: not a template of this search), and the CLASS of an
// index: the grid's end twins are one class (the last n_twin templates stand for the first n_twin - Scarp at +pi/2 is
// minus Scarp at -pi/2, a Ricker the same template: their float64 SNRs differ by rounding noise, one maximum by the parity
// policy).  The lists hold real indices - a template is scored as itself, its amplitude carries its own sign - and at
// most one member of a class per cell: the record's holder where its class is named, else the first listed.
struct IdMap {
    const int32_t* tab;
    uint32_t n_ids;
    int32_t n, n_twin;
    __device__ __forceinline__ int32_t operator()(uint32_t id) const { return id < n_ids ? tab[id] : -1; }
    __device__ __forceinline__ int32_t cls(int32_t t) const { return t >= n - n_twin ? t - (n - n_twin) : t; }
};

// FILL = false: cnt[slot] += what the event adds to its cell's list; true: the same candidates into the lists
template <bool FILL>
__global__ void __launch_bounds__(256)
k_st_events(const uint32_t* __restrict__ ev, unsigned long long n_ev, const uint8_t* __restrict__ near, size_t nc,
            const uint32_t* __restrict__ slot_of, const float* __restrict__ best_snr, float keep,
            const uint32_t* __restrict__ best_id, IdMap map, unsigned* __restrict__ cnt, const unsigned* __restrict__ off,
            int32_t* __restrict__ pair_t, uint32_t* __restrict__ pair_slot) {
    const unsigned long long k = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= n_ev) return;
    const uint32_t cell = ev[SC_EVENT_WORDS * k];
    if (cell >= nc || !near[cell]) return;         // (an event's cell is flagged by the kernel that lists it: slot_of holds a slot there only)
    // An event the search has left behind: the larger of its two float32 scores lies further below the FINAL record than the
    // window.  With scores off by at most e (window >= 2 e) the float64 argmax scores within 2 e of the final record: neither
    // template of such an event can be it.  (The first orientations of a search tie among themselves in every cell - on the
    // carrizo DEM at 2 m five ages are PROPORTIONAL templates at -pi/2: 85 % of all events - long before the record gets
    // where it ends.)
    if (__uint_as_float(ev[SC_EVENT_WORDS * k + 3]) < best_snr[cell] * keep) return;
    const int32_t ms = map(ev[SC_EVENT_WORDS * k + 1]), mh = map(ev[SC_EVENT_WORDS * k + 2]), mf = map(best_id[cell]);
    const int32_t cf = mf >= 0 ? map.cls(mf) : -1;
    const bool a = ms >= 0 && map.cls(ms) != cf, b = mh >= 0 && map.cls(mh) != cf && map.cls(mh) != (ms >= 0 ? map.cls(ms) : -1);
    if (!a && !b) return;
    const uint32_t slot = slot_of[cell];
    if (!FILL) {
        atomicAdd(cnt + slot, (unsigned)a + (unsigned)b);
    } else {
        unsigned pos = off[slot] + atomicAdd(cnt + slot, (unsigned)a + (unsigned)b);
        if (a) { pair_t[pos] = ms; pair_slot[pos] = slot; ++pos; }
        if (b) { pair_t[pos] = mh; pair_slot[pos] = slot; }
    }
}

// entry 0 of every list: the record's final holder; the fill counters start behind it
__global__ void __launch_bounds__(256)
k_st_init_lists(unsigned n_slots, const uint32_t* __restrict__ cell_of, const uint32_t* __restrict__ best_id, IdMap map,
                const unsigned* __restrict__ off, unsigned* __restrict__ fill, int32_t* __restrict__ pair_t,
                uint32_t* __restrict__ pair_slot) {
    const unsigned s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_slots) return;
    const unsigned pos = off[s];
    pair_t[pos] = map(best_id[cell_of[s]]);
    pair_slot[pos] = s;
    fill[s] = 1u;
}

// n = count(W != 0) and sum(W**2) of every template from its float64 window, in a FIXED order (one workgroup per template,
// strided partial sums, a shuffle tree, the four waves in order).  The search's own sums (k_windows) are accumulated with
// float64 atomics: their last bit changes from run to run - and where templates are proportional to each other (a window one
// cell wide at +-pi/2 for the youngest ages on a coarse DEM: the same SNR in exact arithmetic) that bit would decide the
// argmax differently every run.  With these the settle is the same in every bit every time.
__global__ void __launch_bounds__(256)
k_st_sums(const TemplDev* __restrict__ templ, const unsigned long long* __restrict__ woff, const double* __restrict__ wbuf,
          double* __restrict__ sums) {
    const int it = blockIdx.x;
    const int box = templ[it].wh * templ[it].ww;
    const double* __restrict__ wt = wbuf + woff[it];
    double n = 0.0, ts = 0.0;
    for (int e = threadIdx.x; e < box; e += 256) {
        const double w = wt[e];
        n += w != 0.0 ? 1.0 : 0.0;
        ts = fma(w, w, ts);
    }
    __shared__ double red[2][4];
    for (int sft = 32; sft > 0; sft >>= 1) {
        n += __shfl_down(n, sft, 64);
        ts += __shfl_down(ts, sft, 64);
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = n; red[1][threadIdx.x >> 6] = ts; }
    __syncthreads();
    if (threadIdx.x == 0) {
        sums[2 * it] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        sums[2 * it + 1] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
    }
}

// The support of every row of every float64 window: first and last non-zero column (a window is |xr| < c & |yr| < d - a
// convex region: one run per row, with the odd zero inside where xr == 0 exactly), and per template the longest run.  The
// supports of the searches' windows are a SEVENTH of their boxes (a thin young scarp at 45 degrees: seven taps a row in a
// box 145 wide): the scorer walks the runs, not the boxes.  grid = (rows / 4, templates), one wave per row.
__global__ void __launch_bounds__(256)
k_st_spans(const TemplDev* __restrict__ templ, const unsigned long long* __restrict__ woff, const double* __restrict__ wbuf,
           const unsigned* __restrict__ soff, int2* __restrict__ spans, int* __restrict__ maxlen) {
    const int it = blockIdx.y, a = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wh = templ[it].wh, ww = templ[it].ww;
    if (a >= wh) return;
    const double* __restrict__ row = wbuf + woff[it] + (size_t)a * ww;
    int first = 0x7FFFFFFF, last = -1;
    for (int b0 = 0; b0 < ww; b0 += 64) {
        const int b = b0 + lane;
        const unsigned long long m = __ballot(b < ww && row[b < ww ? b : 0] != 0.0);
        if (m) {
            first = min(first, b0 + (int)__builtin_ctzll(m));
            last = b0 + 63 - (int)__builtin_clzll(m);
        }
    }
    if (lane == 0) {
        spans[soff[it] + a] = last >= first ? make_int2(first, last) : make_int2(0, -1);     // (an empty row: a run of no cells)
        if (last >= first) atomicMax(maxlen + it, last - first + 1);
    }
}

__device__ __forceinline__ int wrap1(int x, int n) { return x < 0 ? x + n : (x >= n ? x - n : x); }

// match_template() of one (cell, template) pair in float64 - k_score_f64's arithmetic on the lists of sc_settle_exact.
// One WAVE per list entry (four entries per workgroup: the windows that meet in near-ties are thin - 1 700 taps on average
// on the C3 search - and a workgroup per pair spent its time starting up and reducing); an entry whose template (or its
// end twin) the list holds a second time, or whose list names one template only, is marked (snr = -1) and not scored.
// WPP > 1 (few pairs - a five-scale Channel search settles 350 pairs of 28 000 taps at a time, a wave each took 350 us):
// WPP waves per entry, a workgroup each; wave w takes every WPP-th row group, the partial sums meet in LDS and are added in
// wave order.  Which form runs depends on the number of pairs alone: the same bits from run to run.
template <int WPP>
__global__ void __launch_bounds__(WPP == 1 ? 256 : 64 * WPP)
k_st_score(const double* __restrict__ pa, const double* __restrict__ pb, const double* __restrict__ pc, Geom g,
           const TemplDev* __restrict__ templ, const double* __restrict__ sums,
           const double* __restrict__ xaxis, const double* __restrict__ yaxis,
           const unsigned long long* __restrict__ woff, const double* __restrict__ wbuf,
           const unsigned* __restrict__ soff, const int2* __restrict__ spans, const int* __restrict__ maxlen,
           const unsigned* __restrict__ off, const int32_t* __restrict__ pair_t, const uint32_t* __restrict__ pair_slot,
           const uint32_t* __restrict__ cell_of, IdMap map, unsigned n_pairs, double* __restrict__ amp_out,
           double* __restrict__ snr_out) {
    // workgroups are dealt round-robin to the eight XCDs (each with an L2 of its own): XCD x takes the x-th EIGHTH of the pair
    // list in order, so that the waves that share an L2 work on neighbouring cells
    const unsigned nb8 = (gridDim.x + 7) / 8;
    const unsigned lb = (blockIdx.x & 7) * nb8 + (blockIdx.x >> 3);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned pos = WPP == 1 ? lb * 4 + wv : lb;                                            // (wave-uniform: scalar loads below)
    const int wsel = WPP == 1 ? 0 : wv;                                                           // this wave's share of the rows
    const int lane = threadIdx.x & 63;
    if (pos >= n_pairs) return;
    const uint32_t slot = pair_slot[pos];
    const int it = pair_t[pos];
    const unsigned lo = off[slot], hi = off[slot + 1];
    bool dup = false, other = false;
    const int ci = it >= 0 ? map.cls(it) : -1;
    for (unsigned k = lo; k < hi; ++k) {
        const int tk = pair_t[k];
        const int ck = tk >= 0 ? map.cls(tk) : -2;
        // of a class one member is scored: the record's holder (entry 0), else the smallest index listed - whatever order
        // the events' atomics filled the list in (the same template listed twice: the earlier entry)
        dup = dup || (ck == ci && k != pos && (k == lo || (pos != lo && (tk < it || (tk == it && k < pos)))));
        other = other || (tk >= 0 && ck != ci);
    }
    if (it < 0 || dup || !other) {
        if (lane == 0) { snr_out[pos] = -1.0; amp_out[pos] = 0.0; }
        return;
    }
    const int cw = g.cx1 - g.cx0;
    const uint32_t cell = cell_of[slot];
    const int i = g.cy0 + (int)(cell / (uint32_t)cw), j = g.cx0 + (int)(cell % (uint32_t)cw);     // global cell
    const TemplDev t = templ[it];
    // the orientation's curvature mix, dem.py:103-104 (cos_a / sin_a of the descriptor are those of alpha = -orientation)
    const double ca = t.cos_a, sa = -t.sin_a;
    const double k_cc = __dmul_rn(ca, ca), k_ss = __dmul_rn(sa, sa);
    const double* __restrict__ wt = wbuf + woff[it];
    double xc = 0.0, t3 = 0.0;
    // The runs of the window's rows (k_st_spans).  L = the template's longest run rounded up to a power of two lanes per
    // row, 64 / L rows at a time.  (The bytes through L2 are what this kernel costs: fetching the plane values whether or
    // not a tap weighs anything took 180 ms instead of 78 on the C3 search, walking whole boxes - seven cells for every
    // tap - 78 instead of 43, a workgroup per pair 43 instead of what a wave per pair takes: profiles/r06_settle.txt.)
    // One tap: curvature at global ((i - p + oy) mod ny, (j - q + ox) mod nx), p = pmin + a, q = qmin + b
    auto row_of = [&](int a, size_t& orow) {
        const int gi = i - (t.pmin + a) + g.oy;
        const int li = g.wrap ? wrap1(gi, g.ny) : gi - g.gy0;
        orow = (size_t)(li < 0 || li >= g.ly ? 0 : li) * g.lx;
        return li >= 0 && li < g.ly;                                       // (outside a halo block: the host sized the halo)
    };
    auto col_of = [&](int b, int& lj) {
        const int gj = j - (t.qmin + b) + g.ox;
        lj = g.wrap ? wrap1(gj, g.nx) : gj - g.gx0;
        return lj >= 0 && lj < g.lx;
    };
    const int2* __restrict__ sp_t = spans + soff[it];
    const int ml = maxlen[it];
    if (ml <= 64) {
        // every run fits the lanes of its row: one tap per lane and row, ROWS row groups in flight - the loads of a group are a
        // chain (run -> weight -> three plane values), and what bounds the kernel is how many of them it keeps in the air
        int ls = 0;
        while ((1 << ls) < ml) ++ls;
        const int R = 64 >> ls, bl = lane & ((1 << ls) - 1);
        constexpr int ROWS = 4;
        for (int a0 = (lane >> ls) + wsel * ROWS * R; a0 < t.wh; a0 += WPP * ROWS * R) {
            int2 sp[ROWS];
            size_t orow[ROWS];
            bool v[ROWS];
            double w[ROWS];
#pragma unroll
            for (int u = 0; u < ROWS; ++u) {
                const int a = a0 + u * R;
                sp[u] = a < t.wh ? sp_t[a] : make_int2(0, -1);
                v[u] = a < t.wh && row_of(a, orow[u]);
            }
#pragma unroll
            for (int u = 0; u < ROWS; ++u) {
                const int b = sp[u].x + bl;
                v[u] = v[u] && b <= sp[u].y;
                w[u] = v[u] ? wt[(size_t)(a0 + u * R) * t.ww + b] : 0.0;
            }
            double A[ROWS], Bc[ROWS], C[ROWS];
#pragma unroll
            for (int u = 0; u < ROWS; ++u) {
                int lj = 0;
                v[u] = v[u] && w[u] != 0.0 && col_of(sp[u].x + bl, lj);    // (W != 0 is the mask M, core.py:348)
                const size_t o = v[u] ? orow[u] + lj : 0;
                if (v[u]) { A[u] = pa[o]; Bc[u] = pb[o]; C[u] = pc[o]; }
            }
#pragma unroll
            for (int u = 0; u < ROWS; ++u)
                if (v[u]) {
                    const double cv = __dadd_rn(__dsub_rn(__dmul_rn(A[u], k_cc), __dmul_rn(__dmul_rn(__dmul_rn(2.0, Bc[u]), sa), ca)),
                                                __dmul_rn(C[u], k_ss));
                    xc = fma(w[u], cv, xc);
                    t3 = fma(cv, cv, t3);
                }
        }
    } else {
        // runs longer than the wave: a lane takes the columns lane, lane + 64, ... of a row; FOUR rows and two columns of each
        // in flight (a wide window - a Ricker's support, an old scarp's - is tens of thousands of taps for one wave: row
        // after row with a load chain each it took a third of a millisecond a pair)
        constexpr int ROWS = 4, CH = 2;
        for (int a0 = wsel * ROWS; a0 < t.wh; a0 += WPP * ROWS) {
            int2 sp[ROWS];
            size_t orow[ROWS];
            bool rv[ROWS];
            int more = 0;
#pragma unroll
            for (int u = 0; u < ROWS; ++u) {
                sp[u] = a0 + u < t.wh ? sp_t[a0 + u] : make_int2(0, -1);
                rv[u] = a0 + u < t.wh && row_of(a0 + u, orow[u]);
                more = max(more, rv[u] ? sp[u].y - sp[u].x + 1 : 0);
            }
            for (int c0 = 0; c0 < more; c0 += 64 * CH) {                 // (wave-uniform bound: the longest of the four runs)
                double w[ROWS][CH], A[ROWS][CH], Bc[ROWS][CH], C[ROWS][CH];
                bool v[ROWS][CH];
                size_t o[ROWS][CH];
#pragma unroll
                for (int u = 0; u < ROWS; ++u)
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const int bcol = sp[u].x + c0 + 64 * c + lane;
                        v[u][c] = rv[u] && bcol <= sp[u].y;
                        w[u][c] = v[u][c] ? wt[(size_t)(a0 + u) * t.ww + bcol] : 0.0;
                    }
#pragma unroll
                for (int u = 0; u < ROWS; ++u)
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        int lj = 0;
                        v[u][c] = v[u][c] && w[u][c] != 0.0 && col_of(sp[u].x + c0 + 64 * c + lane, lj);
                        o[u][c] = v[u][c] ? orow[u] + lj : 0;
                        if (v[u][c]) { A[u][c] = pa[o[u][c]]; Bc[u][c] = pb[o[u][c]]; C[u][c] = pc[o[u][c]]; }
                    }
#pragma unroll
                for (int u = 0; u < ROWS; ++u)
#pragma unroll
                    for (int c = 0; c < CH; ++c)
                        if (v[u][c]) {
                            const double cv = __dadd_rn(__dsub_rn(__dmul_rn(A[u][c], k_cc), __dmul_rn(__dmul_rn(__dmul_rn(2.0, Bc[u][c]), sa), ca)),
                                                        __dmul_rn(C[u][c], k_ss));
                            xc = fma(w[u][c], cv, xc);
                            t3 = fma(cv, cv, t3);
                        }
            }
        }
    }
    for (int sft = 32; sft > 0; sft >>= 1) {
        xc += __shfl_down(xc, sft, 64);
        t3 += __shfl_down(t3, sft, 64);
    }
    if constexpr (WPP > 1) {
        __shared__ double part[2 * WPP];
        if (lane == 0) { part[2 * wv] = xc; part[2 * wv + 1] = t3; }
        __syncthreads();
        if (wv != 0) return;
        xc = part[0]; t3 = part[1];
        for (int k = 1; k < WPP; ++k) { xc += part[2 * k]; t3 += part[2 * k + 1]; }
    }
    if (lane == 0) {
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
        if (!(snr >= 0.0)) snr = 0.0;                                      // (a NaN never wins a cell here)
        amp_out[pos] = amp;
        snr_out[pos] = snr;
    }
}

// Per slot: the largest float64 SNR of its list, ties to the earlier template of the fold order (the hand-over order:
// what the device's own fold keeps).  The winner goes into the record (id; amp and snr rounded to float32) and into the
// patch (float64).  A class is scored once per cell, the record's holder standing for its own: an unchanged cell keeps
// its id and gets the float64 values of that very template.
__global__ void __launch_bounds__(256)
k_st_resolve(unsigned n_slots, const unsigned* __restrict__ off, const int32_t* __restrict__ pair_t,
             const double* __restrict__ pair_amp, const double* __restrict__ pair_snr, const uint32_t* __restrict__ cell_of,
             const TemplDev* __restrict__ templ, const double* __restrict__ sums, float* __restrict__ best_snr,
             float* __restrict__ best_amp, uint32_t* __restrict__ best_id, double* __restrict__ p_amp,
             double* __restrict__ p_snr, uint32_t* __restrict__ p_id, unsigned long long* __restrict__ stats) {
    const unsigned s = blockIdx.x * 256 + threadIdx.x;
    unsigned scored = 0, changed = 0;
    unsigned long long taps = 0;                   // (what the scores cost: the taps they weighed)
    if (s < n_slots) {
        const unsigned lo = off[s], hi = off[s + 1];
        double bs = -1.0, ba = 0.0;
        int bt = -1;
        for (unsigned k = lo; k < hi; ++k) {
            const double v = pair_snr[k];
            if (v < 0.0) continue;
            ++scored;
            const int tk = pair_t[k];
            taps += (unsigned long long)sums[2 * tk];
            if (v > bs || (v == bs && tk < bt)) { bs = v; ba = pair_amp[k]; bt = tk; }
        }
        uint32_t id = SC_ID_NONE;
        if (bs > 0.0) {
            const uint32_t cell = cell_of[s];
            changed = bt != pair_t[lo];
            id = templ[bt].id;
            best_id[cell] = id;
            best_snr[cell] = (float)bs;
            best_amp[cell] = (float)ba;
        }
        p_amp[s] = ba;
        p_snr[s] = bs;
        p_id[s] = id;
    }
    unsigned c1 = scored ? 1u : 0u;
    for (int sft = 32; sft > 0; sft >>= 1) {
        scored += __shfl_down(scored, sft, 64);
        changed += __shfl_down(changed, sft, 64);
        c1 += __shfl_down(c1, sft, 64);
        taps += __shfl_down(taps, sft, 64);
    }
    if ((threadIdx.x & 63) == 0 && scored) {
        atomicAdd(stats + 2, (unsigned long long)scored);
        atomicAdd(stats + 3, (unsigned long long)c1);
        atomicAdd(stats + 7, taps);
        if (changed) atomicAdd(stats + 4, (unsigned long long)changed);
    }
}

__global__ void __launch_bounds__(256)
k_st_apply(unsigned n_slots, const uint32_t* __restrict__ cell_of, const double* __restrict__ p_amp,
           const double* __restrict__ p_snr, const uint32_t* __restrict__ p_id, const double* __restrict__ par,
           const double* __restrict__ ang, uint32_t n_ids, size_t nc, double* __restrict__ out) {
    const unsigned s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_slots) return;
    const uint32_t id = p_id[s];
    if (!(p_snr[s] > 0.0) || id >= n_ids) return;
    const size_t c = cell_of[s];
    out[c] = p_amp[s];
    out[nc + c] = par[id];
    out[2 * nc + c] = ang[id];
    out[3 * nc + c] = p_snr[s];
}

inline size_t up64(size_t b) { return (b + 63) & ~(size_t)63; }

}  // namespace

// the patch buffer: cell_of (u32 x n), id (u32 x n), amp, snr (f64 x n)
static void patch_views(sc_ctx* ctx, size_t n, uint32_t** cell_of, uint32_t** p_id, double** p_amp, double** p_snr) {
    char* p = (char*)ctx->st_patch.p;
    *p_amp = (double*)p;
    *p_snr = (double*)(p + up64(8 * n));
    *cell_of = (uint32_t*)(p + 2 * up64(8 * n));
    *p_id = (uint32_t*)(p + 2 * up64(8 * n) + up64(4 * n));
}

int sc_apply_patches(sc_ctx* ctx, const double* tab_par, const double* tab_ang, int n_ids, size_t nc, double* planes) {
    const size_t n = ctx->patch_n;
    if (!n) return SC_OK;
    uint32_t *cell_of, *p_id;
    double *p_amp, *p_snr;
    patch_views(ctx, n, &cell_of, &p_id, &p_amp, &p_snr);
    hipLaunchKernelGGL(k_st_apply, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (unsigned)n, (const uint32_t*)cell_of,
                       (const double*)p_amp, (const double*)p_snr, (const uint32_t*)p_id, tab_par, tab_ang, (uint32_t)n_ids, nc, planes);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

static int settle_impl(sc_ctx* ctx, int n_twin, double max_work, long long* stats_out);

extern "C" int sc_settle_exact(sc_ctx* ctx, int n_twin, double max_work, long long* stats_out) {
    if (!ctx || !stats_out || n_twin < 0) return SC_ERR_INVALID;
    return settle_impl(ctx, n_twin, max_work, stats_out);
}

static int settle_impl(sc_ctx* ctx, int n_twin, double max_work, long long* stats_out) {
    for (unsigned k = 0; k < ST_STATS; ++k) stats_out[k] = 0;
    if (!ctx->have_dem) return sc_fail(ctx, SC_ERR_NO_DEM, "no DEM set");
    const int n = ctx->last_batch;
    if (n <= 0) return sc_fail(ctx, SC_ERR_INVALID, "sc_settle_exact: no search has run in this context");
    if (n_twin > n / 2) return sc_fail(ctx, SC_ERR_INVALID, "sc_settle_exact: %d end twins of %d templates", n_twin, n);
    if (ctx->templ_windows)
        return sc_fail(ctx, SC_ERR_UNSUPPORTED, "sc_settle_exact: built-in templates only (a plugin's window is float32 on the device)");
    ctx->patch_n = 0;
    if (!ctx->near.p || !ctx->near_ev.p) return SC_OK;                    // no search has run with the option on: nothing flagged
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const Geom& g = ctx->g;
    const size_t nc = (size_t)(g.cy1 - g.cy0) * (g.cx1 - g.cx0);
    unsigned long long n_ev = 0;
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SC_HIP(ctx, hipMemcpy(&n_ev, ctx->near_ev.p, sizeof(n_ev), hipMemcpyDeviceToHost));
    stats_out[5] = (long long)n_ev;
    if (n_ev > (ctx->near_ev.cap - 16) / (4 * SC_EVENT_WORDS))
        return sc_fail(ctx, SC_ERR_UNSUPPORTED, "sc_settle_exact: the event list overflowed (%llu near-ties, room for %llu)",
                       n_ev, (unsigned long long)((ctx->near_ev.cap - 16) / (4 * SC_EVENT_WORDS)));
    if (3 * n_ev + nc >= 0xFFFFFFFFull) return sc_fail(ctx, SC_ERR_UNSUPPORTED, "sc_settle_exact: %llu events", n_ev);
    const uint32_t* ev = (const uint32_t*)((const char*)ctx->near_ev.p + 16);
    const uint8_t* near = (const uint8_t*)ctx->near.p;

    // id -> index table from the host's copy of the last search's descriptors
    uint32_t max_id = 0;
    for (int k = 0; k < n; ++k) max_id = std::max(max_id, ctx->h_templ[k].id);
    if (max_id > (1u << 26)) return sc_fail(ctx, SC_ERR_UNSUPPORTED, "sc_settle_exact: template id %u", max_id);
    std::vector<int32_t> idtab((size_t)max_id + 1, -1);
    for (int k = 0; k < n; ++k) idtab[ctx->h_templ[k].id] = k;

    // ---- flagged cells -> slots --------------------------------------------------------------------------------
    const int ch_ = g.cy1 - g.cy0, cw_ = g.cx1 - g.cx0;
    const TileMap tm{ch_, cw_, (cw_ + 63) / 64};
    const unsigned nblk = (unsigned)(((ch_ + 63) / 64) * tm.tiles_x);          // one scan workgroup per 64 x 64 tile
    int rc;
    if ((rc = sc_ensure(ctx, ctx->st_slot, 4 * nc))) return rc;
    // work buffer, first part: stats | block sums (flags) | id table
    const size_t o_blk = up64(8 * ST_STATS), o_tab = o_blk + up64(4 * (size_t)nblk), o_end0 = o_tab + up64(4 * idtab.size());
    // (sized once for the most it can need given the events: slots <= events)
    const size_t ns_max = (size_t)std::min<unsigned long long>(n_ev, nc);
    const unsigned nblk2_max = (unsigned)((ns_max + 1 + ST_CH - 1) / ST_CH);
    const size_t o_cnt = o_end0, o_off = o_cnt + up64(4 * (ns_max + 1)), o_blk2 = o_off + up64(4 * (ns_max + 2)),
                 o_sums = o_blk2 + up64(4 * (size_t)nblk2_max), o_soff = o_sums + up64(16 * (size_t)n),
                 o_mlen = o_soff + up64(4 * ((size_t)n + 1)), o_end = o_mlen + up64(4 * (size_t)n);
    if ((rc = sc_ensure(ctx, ctx->st_work, o_end))) return rc;
    char* wk = (char*)ctx->st_work.p;
    unsigned long long* stats = (unsigned long long*)wk;
    unsigned* blk = (unsigned*)(wk + o_blk);
    int32_t* d_tab = (int32_t*)(wk + o_tab);
    unsigned* cnt = (unsigned*)(wk + o_cnt);
    unsigned* off = (unsigned*)(wk + o_off);
    unsigned* blk2 = (unsigned*)(wk + o_blk2);
    double* sums64 = (double*)(wk + o_sums);
    unsigned* soff = (unsigned*)(wk + o_soff);
    int* maxlen = (int*)(wk + o_mlen);
    // rows of every template's window before it: the offsets of the run table
    std::vector<unsigned> h_soff((size_t)n + 1, 0u);
    int wh_max = 1;
    for (int k = 0; k < n; ++k) {
        h_soff[k + 1] = h_soff[k] + (unsigned)ctx->h_templ[k].wh;
        wh_max = std::max(wh_max, ctx->h_templ[k].wh);
    }
    if ((rc = sc_ensure(ctx, ctx->st_patch, 2 * up64(8 * ns_max) + 2 * up64(4 * ns_max) + 64))) return rc;
    sc_prof_begin(ctx, SC_K_SETTLE);              // (one bracket over the whole call: every return below closes it)
    struct ProfEnd { sc_ctx* c; ~ProfEnd() { sc_prof_end(c); } } prof_end{ctx};
    SC_HIP(ctx, hipMemsetAsync(stats, 0, 8 * ST_STATS, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(d_tab, idtab.data(), 4 * idtab.size(), hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(soff, h_soff.data(), 4 * h_soff.size(), hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemsetAsync(maxlen, 0, 4 * (size_t)n, ctx->stream));
    hipLaunchKernelGGL(k_st_flag_count, dim3(nblk), dim3(256), 0, ctx->stream, near, tm, blk);
    hipLaunchKernelGGL(k_st_scan1, dim3(1), dim3(1024), 0, ctx->stream, blk, nblk, stats);
    SC_HIP(ctx, hipGetLastError());
    unsigned long long n_slots = 0;
    SC_HIP(ctx, hipMemcpyAsync(&n_slots, stats, 8, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));                       // (also: idtab is a local)
    stats_out[0] = (long long)n_slots;
    if (!n_slots) return SC_OK;
    if (n_slots > ns_max) return sc_fail(ctx, SC_ERR_INVALID, "sc_settle_exact: %llu flagged cells but %llu events", n_slots, n_ev);
    const unsigned ns = (unsigned)n_slots;
    uint32_t *cell_of, *p_id;
    double *p_amp, *p_snr;
    patch_views(ctx, ns, &cell_of, &p_id, &p_amp, &p_snr);
    uint32_t* slot_of = (uint32_t*)ctx->st_slot.p;
    const IdMap map{d_tab, max_id + 1, n, n_twin};
    hipLaunchKernelGGL(k_st_slots, dim3(nblk), dim3(256), 0, ctx->stream, near, tm, (const unsigned*)blk, cell_of, slot_of, cnt);
    const unsigned evb = (unsigned)((n_ev + 255) / 256);
    // (the window the events were listed with, less a hair for the float32 product)
    const float keep = (1.f - ctx->near_w_used) * (1.f - 4e-7f);
    hipLaunchKernelGGL(k_st_events<false>, dim3(evb), dim3(256), 0, ctx->stream, ev, n_ev, near, nc, (const uint32_t*)slot_of,
                       (const float*)ctx->best_snr.p, keep,
                       (const uint32_t*)ctx->best_id.p, map, cnt, (const unsigned*)nullptr, (int32_t*)nullptr, (uint32_t*)nullptr);
    // ---- list offsets ------------------------------------------------------------------------------------------
    const unsigned nblk2 = (ns + 1 + ST_CH - 1) / ST_CH;
    hipLaunchKernelGGL(k_st_sum, dim3(nblk2), dim3(256), 0, ctx->stream, (const unsigned*)cnt, ns, blk2);
    hipLaunchKernelGGL(k_st_scan1, dim3(1), dim3(1024), 0, ctx->stream, blk2, nblk2, stats + 1);
    hipLaunchKernelGGL(k_st_offsets, dim3(nblk2), dim3(256), 0, ctx->stream, (const unsigned*)cnt, ns, (const unsigned*)blk2, off);
    SC_HIP(ctx, hipGetLastError());
    unsigned long long n_pairs = 0;
    SC_HIP(ctx, hipMemcpyAsync(&n_pairs, stats + 1, 8, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    stats_out[1] = (long long)n_pairs;
    // how much float64 work that is at most (the caller's bound: nothing is started beyond it): the pairs that can come to be
    // scored - every list entry behind the holder's, and the holder's own where there is one - x the largest support box.
    // (Most lists of a search over few orientations hold the record's holder alone: the grid's end twins tie in every cell
    //  either of them holds, and a list of one class is not scored.)
    int maxbox = 1;
    for (int k = 0; k < n; ++k) maxbox = std::max(maxbox, ctx->h_templ[k].wh * ctx->h_templ[k].ww);
    const unsigned long long may_score = std::min<unsigned long long>(n_pairs, 2ull * (n_pairs - std::min(n_pairs, n_slots)));
    if (max_work > 0.0 && (double)may_score * (double)maxbox > max_work)
        return sc_fail(ctx, SC_ERR_UNSUPPORTED, "sc_settle_exact: too much float64 work (up to %llu pairs on %llu cells, boxes of up to %d cells)",
                       may_score, n_slots, maxbox);
    // ---- pair lists ----------------------------------------------------------------------------------------------
    const size_t np = (size_t)n_pairs;
    if ((rc = sc_ensure(ctx, ctx->st_pairs, 2 * up64(8 * np) + 2 * up64(4 * np) + 64))) return rc;
    char* pp = (char*)ctx->st_pairs.p;
    double* pair_amp = (double*)pp;
    double* pair_snr = (double*)(pp + up64(8 * np));
    int32_t* pair_t = (int32_t*)(pp + 2 * up64(8 * np));
    uint32_t* pair_slot = (uint32_t*)(pp + 2 * up64(8 * np) + up64(4 * np));
    hipLaunchKernelGGL(k_st_init_lists, dim3((ns + 255) / 256), dim3(256), 0, ctx->stream, ns, (const uint32_t*)cell_of,
                       (const uint32_t*)ctx->best_id.p, map, (const unsigned*)off, cnt, pair_t, pair_slot);
    hipLaunchKernelGGL(k_st_events<true>, dim3(evb), dim3(256), 0, ctx->stream, ev, n_ev, near, nc, (const uint32_t*)slot_of,
                       (const float*)ctx->best_snr.p, keep,
                       (const uint32_t*)ctx->best_id.p, map, cnt, (const unsigned*)off, pair_t, pair_slot);
    SC_HIP(ctx, hipGetLastError());
    // ---- float64 scores ------------------------------------------------------------------------------------------
    const unsigned long long* woff = nullptr;
    const double *wbuf = nullptr, *pa = nullptr;
    if ((rc = score_prepare_f64(ctx, n, &woff, &wbuf, &pa))) return rc;
    const size_t npl = (size_t)g.ly * g.lx;
    hipLaunchKernelGGL(k_st_sums, dim3(n), dim3(256), 0, ctx->stream, (const TemplDev*)ctx->templ.p, woff, wbuf, sums64);
    if ((rc = sc_ensure(ctx, ctx->st_spans, sizeof(int2) * (size_t)h_soff[n] + 64))) return rc;
    hipLaunchKernelGGL(k_st_spans, dim3((wh_max + 3) / 4, n), dim3(256), 0, ctx->stream, (const TemplDev*)ctx->templ.p, woff, wbuf,
                       (const unsigned*)soff, (int2*)ctx->st_spans.p, maxlen);
    // (a multiple of eight workgroups: the kernel deals the pair list out over the XCDs in eighths)
#define ST_SCORE(WPPV, BLOCKS, THREADS)                                                                                       \
    hipLaunchKernelGGL(k_st_score<WPPV>, dim3((unsigned)(((BLOCKS) + 7) / 8 * 8)), dim3(THREADS), 0, ctx->stream, pa, pa + npl, \
                       pa + 2 * npl, g, (const TemplDev*)ctx->templ.p, (const double*)sums64, (const double*)ctx->xaxis.p,      \
                       (const double*)ctx->yaxis.p, woff, wbuf, (const unsigned*)soff, (const int2*)ctx->st_spans.p,            \
                       (const int*)maxlen, (const unsigned*)off, (const int32_t*)pair_t,                                        \
                       (const uint32_t*)pair_slot, (const uint32_t*)cell_of, map, (unsigned)np, pair_amp, pair_snr)
    // a wave per pair fills the chip from a few thousand pairs on; shorter lists of WIDE windows spread every pair over 4 or
    // 16 waves (taps of the widest window from its descriptor: the rotated rectangle 2c x 2d cut to its box - a Scarp of a
    // few hundred taps settles faster on the one wave: C1 0.20 against 0.23 ms)
    double taps_max = 0.0;
    for (int k = 0; k < n; ++k) {
        const TemplDev& tk = ctx->h_templ[k];
        taps_max = std::max(taps_max, std::min((double)tk.wh * tk.ww, 4.0 * tk.c * tk.d / fabs(ctx->dx * ctx->dy)));
    }
    const bool wide = taps_max >= 8192.0 && ctx->variant != 20;
    if (wide && np < 4096) ST_SCORE(16, np, 1024);
    else if (wide && np < 32768) ST_SCORE(4, np, 256);
    else ST_SCORE(1, (np + 3) / 4, 256);
#undef ST_SCORE
    hipLaunchKernelGGL(k_st_resolve, dim3((ns + 255) / 256), dim3(256), 0, ctx->stream, ns, (const unsigned*)off, (const int32_t*)pair_t,
                       (const double*)pair_amp, (const double*)pair_snr, (const uint32_t*)cell_of, (const TemplDev*)ctx->templ.p,
                       (const double*)sums64, (float*)ctx->best_snr.p, (float*)ctx->best_amp.p, (uint32_t*)ctx->best_id.p, p_amp, p_snr, p_id, stats);
    SC_HIP(ctx, hipGetLastError());
    unsigned long long h[ST_STATS] = {0};
    SC_HIP(ctx, hipMemcpyAsync(h, stats, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    stats_out[2] = (long long)h[2];
    stats_out[3] = (long long)h[3];
    stats_out[4] = (long long)h[4];
    stats_out[7] = (long long)h[7];
    ctx->patch_n = ns;
    ctx->async_in_flight = false;
    return SC_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// The settle of an ORIENTATION-SHARDED search (scarplet_amd.dist.OrientationMatcher, exact mode).  Every rank holds the
// whole DEM and searched its share of the templates with the near-tie window on; sc_fold_ranks has made every rank's
// record the fold of all of them.  A near-tie between templates of two ranks is in neither rank's list - but every
// template that can be the float64 argmax scores, in float32, within the window of the FOLDED record, and its rank
// knows it: it is named by one of the rank's events whose larger score lies within the window of the folded record, or
// it is the holder of the rank's own record (sc_snapshot_best, taken before the fold) where that lies within the window.
// sc_rank_candidates lists those (cell, template id) pairs; the ranks exchange their lists (a few megabytes, through the
// launcher's transport) and EVERY rank settles the union with the whole search's descriptors (sc_settle_pairs): the same
// pairs, the same float64 scores, the same record on every rank - no further collective.
// ---------------------------------------------------------------------------------------------------------------
namespace {

__global__ void __launch_bounds__(256)
k_rank_pairs_events(const uint32_t* __restrict__ ev, unsigned long long n_ev, const float* __restrict__ best_snr, float keep,
                    size_t nc, uint32_t* __restrict__ out, unsigned long long cap, unsigned long long* __restrict__ count) {
    const unsigned long long k = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= n_ev) return;
    const uint32_t cell = ev[SC_EVENT_WORDS * k];
    if (cell >= nc || __uint_as_float(ev[SC_EVENT_WORDS * k + 3]) < best_snr[cell] * keep) return;
    for (int w = 1; w <= 2; ++w) {
        const uint32_t id = ev[SC_EVENT_WORDS * k + w];
        if (id == SC_ID_NONE) continue;
        const unsigned long long slot = atomicAdd(count, 1ull);
        if (slot < cap) { out[2 * slot] = cell; out[2 * slot + 1] = id; }
    }
}

// the rank's own holder where the folded record's holder is another template and the two lie within the window
__global__ void __launch_bounds__(256)
k_rank_pairs_holders(const float* __restrict__ snap_snr, const uint32_t* __restrict__ snap_id, const float* __restrict__ best_snr,
                     const uint32_t* __restrict__ best_id, float keep, size_t nc, uint32_t* __restrict__ out,
                     unsigned long long cap, unsigned long long* __restrict__ count) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nc; i += (size_t)gridDim.x * 256) {
        const uint32_t id = snap_id[i];
        if (id == SC_ID_NONE || id == best_id[i]) continue;
        const float s = snap_snr[i];
        if (!(s > 0.f) || s < best_snr[i] * keep) continue;
        const unsigned long long slot = atomicAdd(count, 1ull);
        if (slot < cap) { out[2 * slot] = (uint32_t)i; out[2 * slot + 1] = id; }
    }
}

// the exchanged pairs as events the settle reads: (cell, template, no holder, a score nothing drops).  Appended through the
// counter (the settle does not depend on the order of its events); the padding of sc_exchange_candidates' slots - cells
// beyond the core - is left out
__global__ void __launch_bounds__(256)
k_pairs_to_events(const uint32_t* __restrict__ pairs, unsigned long long n, size_t nc, uint32_t* __restrict__ ev,
                  unsigned long long* __restrict__ ev_count, uint8_t* __restrict__ near) {
    const unsigned long long k = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const uint32_t cell = pairs[2 * k];
    if (cell >= nc) return;
    const unsigned long long slot = atomicAdd(ev_count, 1ull);
    ev[SC_EVENT_WORDS * slot] = cell;
    ev[SC_EVENT_WORDS * slot + 1] = pairs[2 * k + 1];
    ev[SC_EVENT_WORDS * slot + 2] = SC_ID_NONE;
    ev[SC_EVENT_WORDS * slot + 3] = 0x7F800000u;                            // +inf
    near[cell] = (uint8_t)1;
}

}  // namespace

extern "C" int sc_set_best(sc_ctx* ctx, const float* amp, const float* snr, const uint32_t* id) {
    if (!ctx || !amp || !snr || !id) return SC_ERR_INVALID;
    if (!ctx->have_dem) return sc_fail(ctx, SC_ERR_NO_DEM, "no DEM set");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nc = (size_t)(ctx->g.cy1 - ctx->g.cy0) * (ctx->g.cx1 - ctx->g.cx0);
    SC_HIP(ctx, hipMemcpyAsync(ctx->best_amp.p, amp, sizeof(float) * nc, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(ctx->best_snr.p, snr, sizeof(float) * nc, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(ctx->best_id.p, id, sizeof(uint32_t) * nc, hipMemcpyHostToDevice, ctx->stream));
    ctx->patch_n = 0;
    return sc_sync(ctx);
}

extern "C" int sc_snapshot_best(sc_ctx* ctx) {
    if (!ctx) return SC_ERR_INVALID;
    if (!ctx->have_dem) return sc_fail(ctx, SC_ERR_NO_DEM, "no DEM set");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nc = (size_t)(ctx->g.cy1 - ctx->g.cy0) * (ctx->g.cx1 - ctx->g.cx0);
    int rc;
    if ((rc = sc_ensure(ctx, ctx->snap, 8 * nc))) return rc;
    SC_HIP(ctx, hipMemcpyAsync(ctx->snap.p, ctx->best_snr.p, 4 * nc, hipMemcpyDeviceToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync((char*)ctx->snap.p + 4 * nc, ctx->best_id.p, 4 * nc, hipMemcpyDeviceToDevice, ctx->stream));
    ctx->snap_cells = nc;
    return SC_OK;
}

extern "C" int sc_rank_candidates(sc_ctx* ctx, uint32_t* pairs, long long capacity, long long* n_pairs) {
    if (!ctx || !n_pairs || capacity < 0 || (capacity > 0 && !pairs)) return SC_ERR_INVALID;
    *n_pairs = 0;
    ctx->cand_n = -1;                                                     // (until this call has listed them)
    if (!ctx->have_dem) return sc_fail(ctx, SC_ERR_NO_DEM, "no DEM set");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nc = (size_t)(ctx->g.cy1 - ctx->g.cy0) * (ctx->g.cx1 - ctx->g.cx0);
    if (ctx->snap_cells != nc) return sc_fail(ctx, SC_ERR_INVALID, "sc_rank_candidates: no snapshot of this rank's record (sc_snapshot_best)");
    unsigned long long n_ev = 0;
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->near_ev.p) {
        SC_HIP(ctx, hipMemcpy(&n_ev, ctx->near_ev.p, sizeof(n_ev), hipMemcpyDeviceToHost));
        if (n_ev > (ctx->near_ev.cap - 16) / (4 * SC_EVENT_WORDS))
            return sc_fail(ctx, SC_ERR_UNSUPPORTED, "sc_rank_candidates: the event list overflowed (%llu near-ties)", n_ev);
    }
    const unsigned long long cap = 2 * n_ev + nc;
    int rc;
    if ((rc = sc_ensure(ctx, ctx->st_pairs, 16 + 8 * cap))) return rc;
    unsigned long long* count = (unsigned long long*)ctx->st_pairs.p;
    uint32_t* out = (uint32_t*)((char*)ctx->st_pairs.p + 16);
    const float keep = (1.f - ctx->near_w_used) * (1.f - 4e-7f);
    SC_HIP(ctx, hipMemsetAsync(count, 0, 16, ctx->stream));
    if (n_ev)
        hipLaunchKernelGGL(k_rank_pairs_events, dim3((unsigned)((n_ev + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const uint32_t*)((const char*)ctx->near_ev.p + 16), n_ev, (const float*)ctx->best_snr.p, keep, nc, out, cap, count);
    hipLaunchKernelGGL(k_rank_pairs_holders, dim3((unsigned)std::min<size_t>((nc + 255) / 256, 16384)), dim3(256), 0, ctx->stream,
                       (const float*)ctx->snap.p, (const uint32_t*)((const char*)ctx->snap.p + 4 * nc), (const float*)ctx->best_snr.p,
                       (const uint32_t*)ctx->best_id.p, keep, nc, out, cap, count);
    SC_HIP(ctx, hipGetLastError());
    unsigned long long n = 0;
    SC_HIP(ctx, hipMemcpyAsync(&n, count, 8, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_pairs = (long long)n;
    ctx->cand_n = -1;
    if (n > cap) return sc_fail(ctx, SC_ERR_INVALID, "sc_rank_candidates: %llu pairs", n);
    ctx->cand_n = (long long)n;
    if ((unsigned long long)capacity >= n && n)
        SC_HIP(ctx, hipMemcpy(pairs, out, 8 * n, hipMemcpyDeviceToHost));
    return SC_OK;
}

extern "C" int sc_settle_pairs(sc_ctx* ctx, const sc_template* t, int n, const uint32_t* pairs, long long n_pairs, int n_twin,
                               double max_work, long long* stats_out) {
    if (!ctx || !t || n <= 0 || n_pairs < 0 || n_twin < 0 || !stats_out) return SC_ERR_INVALID;
    if (!ctx->have_dem) return sc_fail(ctx, SC_ERR_NO_DEM, "no DEM set");
    const bool from_device = pairs == nullptr && n_pairs > 0;             // what sc_exchange_candidates left on the device
    if (from_device && n_pairs != ctx->xch_n)
        return sc_fail(ctx, SC_ERR_INVALID, "sc_settle_pairs: %lld pairs without a host list, %lld exchanged", n_pairs, ctx->xch_n);
    SC_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = sc_load_templates(ctx, t, n))) return rc;                    // the WHOLE search's descriptors: the scorer's table
    const size_t nc = (size_t)(ctx->g.cy1 - ctx->g.cy0) * (ctx->g.cx1 - ctx->g.cx0);
    unsigned long long* ev_count = nullptr;
    uint32_t* ev = nullptr;
    unsigned long long ev_cap = 0;
    const float w_used = ctx->near_w_used;
    if ((rc = sc_near_buffers(ctx, &ev_count, &ev, &ev_cap))) return rc;
    ctx->near_w_used = w_used;                                            // (the window the ranks searched with, not the option as it stands now)
    if ((unsigned long long)n_pairs > ev_cap) return sc_fail(ctx, SC_ERR_UNSUPPORTED, "sc_settle_pairs: %lld pairs, room for %llu", n_pairs, ev_cap);
    SC_HIP(ctx, hipMemsetAsync(ctx->near.p, 0, nc, ctx->stream));
    SC_HIP(ctx, hipMemsetAsync(ev_count, 0, 16, ctx->stream));
    const unsigned long long np = (unsigned long long)n_pairs;
    if (np) {
        const uint32_t* d_pairs = (const uint32_t*)ctx->xch.p;
        if (!from_device) {
            if ((rc = sc_ensure(ctx, ctx->st_pairs, 16 + 8 * np))) return rc;
            d_pairs = (const uint32_t*)((char*)ctx->st_pairs.p + 16);
            ctx->cand_n = -1;                                             // (the rank's own list lived there)
            SC_HIP(ctx, hipMemcpyAsync((void*)d_pairs, pairs, 8 * np, hipMemcpyHostToDevice, ctx->stream));
        }
        hipLaunchKernelGGL(k_pairs_to_events, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, ctx->stream, d_pairs, np, nc, ev, ev_count,
                           (uint8_t*)ctx->near.p);
        SC_HIP(ctx, hipGetLastError());
    }
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));                        // (st_pairs is taken again by the settle)
    return settle_impl(ctx, n_twin, max_work, stats_out);
}
