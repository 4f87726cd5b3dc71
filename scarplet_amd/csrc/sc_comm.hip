// Multi-GPU: RCCL (xGMI) halo exchange of elevation cells between the ranks of
// a tiled DEM.  One process per GPU; the communicator is created from an
// ncclUniqueId that the host distributes (scarplet_amd/dist.py).
#include "sc_internal.h"
#include <rccl/rccl.h>
#include <string.h>

#define SC_NCCL(ctx, call)                                                    \
    do {                                                                      \
        ncclResult_t r__ = (call);                                            \
        if (r__ != ncclSuccess)                                               \
            return sc_fail(ctx, SC_ERR_COMM, "%s: %s (%s:%d)", #call,         \
                           ncclGetErrorString(r__), __FILE__, __LINE__);      \
    } while (0)

static_assert(sizeof(ncclUniqueId) <= SC_COMM_ID_BYTES, "unique id size");

extern "C" int sc_comm_unique_id(void* id_out) {
    if (!id_out) return SC_ERR_INVALID;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return SC_ERR_COMM;
    memset(id_out, 0, SC_COMM_ID_BYTES);
    memcpy(id_out, &id, sizeof(id));
    return SC_OK;
}

extern "C" int sc_comm_init(sc_ctx* ctx, const void* id, int rank, int nranks) {
    if (!ctx || !id || nranks < 1 || rank < 0 || rank >= nranks) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->comm) sc_comm_destroy(ctx);
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm;
    SC_NCCL(ctx, ncclCommInitRank(&comm, nranks, uid, rank));
    ctx->comm = (void*)comm;
    ctx->rank = rank;
    ctx->nranks = nranks;
    return SC_OK;
}

// What RCCL itself reports about this rank's communicator (bench.py prints it with every
// multi-GPU line: "did RCCL see N ranks, and which devices").
extern "C" int sc_comm_info(sc_ctx* ctx, int* nranks, int* rank, int* device, char* bus_id, int bus_id_len) {
    if (!ctx || !nranks || !rank || !device) return SC_ERR_INVALID;
    *nranks = 0; *rank = -1; *device = ctx->device;
    if (bus_id && bus_id_len > 0) {
        bus_id[0] = 0;
        (void)hipDeviceGetPCIBusId(bus_id, bus_id_len, ctx->device);
    }
    if (!ctx->comm) return SC_OK;                       // no communicator: nranks = 0
    SC_NCCL(ctx, ncclCommCount((ncclComm_t)ctx->comm, nranks));
    SC_NCCL(ctx, ncclCommUserRank((ncclComm_t)ctx->comm, rank));
    SC_NCCL(ctx, ncclCommCuDevice((ncclComm_t)ctx->comm, device));
    return SC_OK;
}

extern "C" int sc_comm_destroy(sc_ctx* ctx) {
    if (!ctx) return SC_ERR_INVALID;
    if (ctx->comm) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        ncclCommDestroy((ncclComm_t)ctx->comm);
        ctx->comm = nullptr;
    }
    return SC_OK;
}

extern "C" int sc_halo_exchange(sc_ctx* ctx, const double* core, int core_h, int core_w,
                                int hy_lo, int hy_hi, int hx_lo, int hx_hi,
                                const sc_xfer* x, int n, void** z_dev) {
    if (!ctx || !core || !z_dev || core_h <= 0 || core_w <= 0 || hy_lo < 0 || hy_hi < 0 ||
        hx_lo < 0 || hx_hi < 0 || n < 0 || (n > 0 && !x))
        return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const int H = hy_lo + core_h + hy_hi, W = hx_lo + core_w + hx_hi;
    int rc = sc_ensure(ctx, ctx->halo_z, sizeof(double) * (size_t)H * W);
    if (rc) return rc;
    double* blk = (double*)ctx->halo_z.p;
    const size_t pitch = sizeof(double) * (size_t)W;
    SC_HIP(ctx, hipMemsetAsync(blk, 0, sizeof(double) * (size_t)H * W, ctx->stream));
    SC_HIP(ctx, hipMemcpy2DAsync(blk + (size_t)hy_lo * W + hx_lo, pitch, core,
                                 sizeof(double) * (size_t)core_w, sizeof(double) * (size_t)core_w,
                                 core_h, hipMemcpyHostToDevice, ctx->stream));
    size_t cells = 0;
    bool remote = false;
    for (int i = 0; i < n; ++i) {
        const sc_xfer& t = x[i];
        if (t.h <= 0 || t.w <= 0) return sc_fail(ctx, SC_ERR_INVALID, "transfer %d: empty", i);
        bool src_ok = t.sy0 >= 0 && t.sx0 >= 0 && t.sy0 + t.h <= H && t.sx0 + t.w <= W;
        bool dst_ok = t.dy0 >= 0 && t.dx0 >= 0 && t.dy0 + t.h <= H && t.dx0 + t.w <= W;
        if ((t.kind != SC_XFER_RECV && !src_ok) || (t.kind != SC_XFER_SEND && !dst_ok))
            return sc_fail(ctx, SC_ERR_INVALID, "transfer %d: rectangle outside the block", i);
        if (t.kind == SC_XFER_LOCAL) continue;
        if (t.kind != SC_XFER_SEND && t.kind != SC_XFER_RECV)
            return sc_fail(ctx, SC_ERR_INVALID, "transfer %d: bad kind", i);
        if (!ctx->comm || t.peer < 0 || t.peer >= ctx->nranks || t.peer == ctx->rank)
            return sc_fail(ctx, SC_ERR_COMM, "transfer %d: no communicator / bad peer", i);
        cells += (size_t)t.h * t.w;
        remote = true;
    }
    if ((rc = sc_ensure(ctx, ctx->halo_stage, sizeof(double) * std::max<size_t>(cells, 1)))) return rc;
    double* stage = (double*)ctx->halo_stage.p;
    // pack outgoing rectangles (they lie in the core part, which is complete)
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        const sc_xfer& t = x[i];
        if (t.kind == SC_XFER_LOCAL) continue;
        if (t.kind == SC_XFER_SEND)
            SC_HIP(ctx, hipMemcpy2DAsync(stage + off, sizeof(double) * (size_t)t.w,
                                         blk + (size_t)t.sy0 * W + t.sx0, pitch,
                                         sizeof(double) * (size_t)t.w, t.h,
                                         hipMemcpyDeviceToDevice, ctx->stream));
        off += (size_t)t.h * t.w;
    }
    if (remote) {
        ncclComm_t comm = (ncclComm_t)ctx->comm;
        SC_NCCL(ctx, ncclGroupStart());
        off = 0;
        for (int i = 0; i < n; ++i) {
            const sc_xfer& t = x[i];
            if (t.kind == SC_XFER_LOCAL) continue;
            size_t cnt = (size_t)t.h * t.w;
            if (t.kind == SC_XFER_SEND)
                SC_NCCL(ctx, ncclSend(stage + off, cnt, ncclDouble, t.peer, comm, ctx->stream));
            else
                SC_NCCL(ctx, ncclRecv(stage + off, cnt, ncclDouble, t.peer, comm, ctx->stream));
            off += cnt;
        }
        SC_NCCL(ctx, ncclGroupEnd());
    }
    // unpack received rectangles, then the periodic images of the own core
    off = 0;
    for (int i = 0; i < n; ++i) {
        const sc_xfer& t = x[i];
        if (t.kind == SC_XFER_LOCAL) continue;
        if (t.kind == SC_XFER_RECV)
            SC_HIP(ctx, hipMemcpy2DAsync(blk + (size_t)t.dy0 * W + t.dx0, pitch, stage + off,
                                         sizeof(double) * (size_t)t.w, sizeof(double) * (size_t)t.w,
                                         t.h, hipMemcpyDeviceToDevice, ctx->stream));
        off += (size_t)t.h * t.w;
    }
    for (int i = 0; i < n; ++i) {
        const sc_xfer& t = x[i];
        if (t.kind != SC_XFER_LOCAL) continue;
        SC_HIP(ctx, hipMemcpy2DAsync(blk + (size_t)t.dy0 * W + t.dx0, pitch,
                                     blk + (size_t)t.sy0 * W + t.sx0, pitch,
                                     sizeof(double) * (size_t)t.w, t.h,
                                     hipMemcpyDeviceToDevice, ctx->stream));
    }
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *z_dev = (void*)blk;
    return SC_OK;
}

// The final gather moves the float32 RECORD - amplitude, SNR, template id: 12 bytes per cell - and the root
// converts it into the reference's four float64 planes (the id -> (age, orientation) tables are the same on
// every rank): 2.7 x fewer bytes over xGMI than the converted planes (32 B per cell), and one conversion
// kernel per rank at the root instead of one on every rank.  All receives are posted in ONE group - the
// senders run in parallel over their own links - into one staging block per rank; the conversions and the
// device-to-host placements follow on the stream, no host synchronisation between ranks.
extern "C" int sc_gather_result(sc_ctx* ctx, int root, const int32_t* cores, int ny, int nx,
                                const double* param_of_id, const double* angle_of_id, int n_ids,
                                double* out) {
    if (!ctx || !cores || ny <= 0 || nx <= 0 || !param_of_id || !angle_of_id || n_ids <= 0)
        return SC_ERR_INVALID;
    if (!ctx->have_dem) return sc_fail(ctx, SC_ERR_NO_DEM, "no DEM set");
    const int nranks = ctx->comm ? ctx->nranks : 1, rank = ctx->comm ? ctx->rank : 0;
    if (root < 0 || root >= nranks || (rank == root && !out)) return SC_ERR_INVALID;
    const int32_t* mine = cores + 4 * rank;
    if (mine[0] != ctx->g.cy0 || mine[1] != ctx->g.cy1 || mine[2] != ctx->g.cx0 || mine[3] != ctx->g.cx1)
        return sc_fail(ctx, SC_ERR_INVALID, "cores[%d] is not this context's core", rank);
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nc = (size_t)(mine[1] - mine[0]) * (mine[3] - mine[2]);
    if (rank != root) {
        ncclComm_t comm = (ncclComm_t)ctx->comm;
        SC_NCCL(ctx, ncclGroupStart());
        SC_NCCL(ctx, ncclSend(ctx->best_amp.p, nc, ncclFloat, root, comm, ctx->stream));
        SC_NCCL(ctx, ncclSend(ctx->best_snr.p, nc, ncclFloat, root, comm, ctx->stream));
        SC_NCCL(ctx, ncclSend(ctx->best_id.p, nc, ncclUint32, root, comm, ctx->stream));
        SC_NCCL(ctx, ncclGroupEnd());
        SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return SC_OK;
    }
    // root: cells of every other rank, the largest core, the staging offsets (in cells)
    size_t others = 0, max_nc = nc;
    for (int r = 0; r < nranks; ++r) {
        const int32_t* c = cores + 4 * r;
        const int h = c[1] - c[0], w = c[3] - c[2];
        if (h <= 0 || w <= 0 || c[0] < 0 || c[1] > ny || c[2] < 0 || c[3] > nx)
            return sc_fail(ctx, SC_ERR_INVALID, "cores[%d] outside the DEM", r);
        if (r != root) others += (size_t)h * w;
        max_nc = std::max(max_nc, (size_t)h * w);
    }
    int rc;
    if ((rc = sc_ensure(ctx, ctx->halo_stage, SC_RECORD_BYTES * std::max<size_t>(others, 1)))) return rc;
    if ((rc = sc_ensure(ctx, ctx->res, sizeof(double) * (4 * max_nc + 2 * (size_t)n_ids)))) return rc;
    double* planes = (double*)ctx->res.p;
    double* tab = planes + 4 * max_nc;
    SC_HIP(ctx, hipMemcpyAsync(tab, param_of_id, sizeof(double) * n_ids, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(tab + n_ids, angle_of_id, sizeof(double) * n_ids, hipMemcpyHostToDevice, ctx->stream));
    char* stage = (char*)ctx->halo_stage.p;
    if (nranks > 1) {
        ncclComm_t comm = (ncclComm_t)ctx->comm;
        SC_NCCL(ctx, ncclGroupStart());
        size_t off = 0;
        for (int r = 0; r < nranks; ++r) {
            if (r == root) continue;
            const int32_t* c = cores + 4 * r;
            const size_t n = (size_t)(c[1] - c[0]) * (c[3] - c[2]);
            // a rank's record in the staging block: amp[n] | snr[n] | id[n], the order of its three sends
            SC_NCCL(ctx, ncclRecv(stage + off, n, ncclFloat, r, comm, ctx->stream));
            SC_NCCL(ctx, ncclRecv(stage + off + 4 * n, n, ncclFloat, r, comm, ctx->stream));
            SC_NCCL(ctx, ncclRecv(stage + off + 8 * n, n, ncclUint32, r, comm, ctx->stream));
            off += SC_RECORD_BYTES * n;
        }
        SC_NCCL(ctx, ncclGroupEnd());
    }
    const size_t full = (size_t)ny * nx;
    // one rank's record -> four float64 planes (device) -> its rectangle of the root's host planes; the
    // planes buffer is reused rank after rank: kernel and copies are ordered by the stream
    auto place = [&](const float* amp, const float* snr, const uint32_t* id, const int32_t* c) -> int {
        const int h = c[1] - c[0], w = c[3] - c[2];
        const size_t n = (size_t)h * w;
        int rc_ = sc_launch_result(ctx, amp, snr, id, tab, tab + n_ids, n_ids, n, planes);
        if (rc_) return rc_;
        for (int k = 0; k < 4; ++k)
            SC_HIP(ctx, hipMemcpy2DAsync(out + k * full + (size_t)c[0] * nx + c[2], sizeof(double) * nx,
                                         planes + (size_t)k * n, sizeof(double) * w,
                                         sizeof(double) * w, h, hipMemcpyDeviceToHost, ctx->stream));
        return SC_OK;
    };
    if ((rc = place((const float*)ctx->best_amp.p, (const float*)ctx->best_snr.p, (const uint32_t*)ctx->best_id.p, mine)))
        return rc;
    size_t off = 0;
    for (int r = 0; r < nranks; ++r) {
        if (r == root) continue;
        const int32_t* c = cores + 4 * r;
        const size_t n = (size_t)(c[1] - c[0]) * (c[3] - c[2]);
        if ((rc = place((const float*)(stage + off), (const float*)(stage + off + 4 * n),
                        (const uint32_t*)(stage + off + 8 * n), c)))
            return rc;
        off += SC_RECORD_BYTES * n;
    }
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SC_OK;
}

// ---- fold of the running-best records of orientation-sharded ranks ---------------------
// key = SNR bits (order-preserving for the non-negative SNRs; a NaN sorts above every number,
// as the sticky NaN of sc_fold does) << 32 | ~id: ncclMax picks the greatest SNR and, among
// equal SNRs, the smallest id - what one context folding all templates in id order keeps.
__global__ void __launch_bounds__(256)
k_fold_pack(const float* __restrict__ snr, const uint32_t* __restrict__ id,
            unsigned long long* __restrict__ key, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    {
        // the bit order is the numeric order only for non-negative numbers: a record SNR is |..| >= 0
        // by construction (sc_epilogue), but a -0.0 or a negative value would sort above every
        // positive one - anything that is not > 0 and not a NaN packs as 0
        const float s = snr[i];
        const uint32_t bits = (s > 0.f || s != s) ? __float_as_uint(s) : 0u;
        key[i] = ((unsigned long long)bits << 32) | (unsigned long long)(0xFFFFFFFFu - id[i]);
    }
}
// the winner's SNR and id to every rank; the amplitude stays only where this rank held the winner
__global__ void __launch_bounds__(256)
k_fold_unpack(const unsigned long long* __restrict__ key, float* __restrict__ snr,
              float* __restrict__ amp, uint32_t* __restrict__ id, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned long long k = key[i];
        const uint32_t w = 0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFull);
        if (id[i] != w || w == 0xFFFFFFFFu) amp[i] = 0.f;
        id[i] = w;
        snr[i] = __uint_as_float((uint32_t)(k >> 32));
    }
}

extern "C" int sc_fold_ranks(sc_ctx* ctx) {
    if (!ctx) return SC_ERR_INVALID;
    if (!ctx->have_dem) return sc_fail(ctx, SC_ERR_NO_DEM, "no DEM set");
    if (!ctx->comm) return SC_OK;          // (a one-rank communicator runs the whole sequence: the tests' handle on it)
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nc = (size_t)(ctx->g.cy1 - ctx->g.cy0) * (ctx->g.cx1 - ctx->g.cx0);
    int rc = sc_ensure(ctx, ctx->halo_stage, sizeof(unsigned long long) * nc);
    if (rc) return rc;
    unsigned long long* key = (unsigned long long*)ctx->halo_stage.p;
    const int blocks = (int)std::min<size_t>((nc + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(k_fold_pack, dim3(blocks), dim3(256), 0, ctx->stream, (const float*)ctx->best_snr.p,
                       (const uint32_t*)ctx->best_id.p, key, nc);
    SC_HIP(ctx, hipGetLastError());
    SC_NCCL(ctx, ncclAllReduce(key, key, nc, ncclUint64, ncclMax, (ncclComm_t)ctx->comm, ctx->stream));
    hipLaunchKernelGGL(k_fold_unpack, dim3(blocks), dim3(256), 0, ctx->stream, (const unsigned long long*)key,
                       (float*)ctx->best_snr.p, (float*)ctx->best_amp.p, (uint32_t*)ctx->best_id.p, nc);
    SC_HIP(ctx, hipGetLastError());
    SC_NCCL(ctx, ncclAllReduce(ctx->best_amp.p, ctx->best_amp.p, nc, ncclFloat, ncclSum,
                               (ncclComm_t)ctx->comm, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SC_OK;
}

/*
 * The ranks' candidate lists of an orientation-sharded exact search (sc_rank_candidates: each left on its device) as one
 * list on every device.  Two all-gathers over RCCL/xGMI: the counts (one 64-bit word per rank), then slots of the largest
 * count - in place, every rank's own list already in its slot, the rest of a slot filled with cells no DEM has
 * (0xFFFFFFFF: sc_settle_pairs leaves them out).  Every rank ends with the same slots in rank order.  Without a
 * communicator the union is the rank's own list.
 */
extern "C" int sc_exchange_candidates(sc_ctx* ctx, long long* n_union) {
    if (!ctx || !n_union) return SC_ERR_INVALID;
    *n_union = 0;
    ctx->xch_n = -1;
    if (!ctx->have_dem) return sc_fail(ctx, SC_ERR_NO_DEM, "no DEM set");
    // (a rank whose sc_rank_candidates failed - its event list overflowed - still takes part in the counts' all-gather, with a
    //  count no list can have: every rank then returns the same error instead of one rank leaving the others in a collective)
    const bool mine_bad = ctx->cand_n < 0;
    if (mine_bad && !ctx->comm) return sc_fail(ctx, SC_ERR_INVALID, "sc_exchange_candidates: no candidate list on the device (sc_rank_candidates)");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    int nr = 1, me = 0;
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    if (comm) {
        SC_NCCL(ctx, ncclCommCount(comm, &nr));
        SC_NCCL(ctx, ncclCommUserRank(comm, &me));
    }
    const unsigned long long mine = mine_bad ? ~0ull : (unsigned long long)ctx->cand_n;
    std::vector<unsigned long long> counts((size_t)nr, mine);
    int rc;
    if (comm) {
        if ((rc = sc_ensure(ctx, ctx->xch_cnt, 8 * (size_t)(nr + 1)))) return rc;
        unsigned long long* d_cnt = (unsigned long long*)ctx->xch_cnt.p;
        SC_HIP(ctx, hipMemcpyAsync(d_cnt + nr, &mine, 8, hipMemcpyHostToDevice, ctx->stream));
        SC_NCCL(ctx, ncclAllGather(d_cnt + nr, d_cnt, 1, ncclUint64, comm, ctx->stream));
        SC_HIP(ctx, hipMemcpyAsync(counts.data(), d_cnt, 8 * (size_t)nr, hipMemcpyDeviceToHost, ctx->stream));
        SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    unsigned long long slot = 0;
    for (int r = 0; r < nr; ++r) {
        if (counts[r] == ~0ull)
            return sc_fail(ctx, SC_ERR_UNSUPPORTED, "sc_exchange_candidates: rank %d has no candidate list (its event list overflowed)", r);
        slot = std::max(slot, counts[r]);
    }
    if (slot == 0) { ctx->xch_n = 0; return SC_OK; }
    if ((rc = sc_ensure(ctx, ctx->xch, 8 * slot * (size_t)nr))) return rc;
    uint32_t* all = (uint32_t*)ctx->xch.p;
    uint32_t* own = all + 2 * slot * (size_t)me;
    SC_HIP(ctx, hipMemsetAsync(own, 0xFF, 8 * slot, ctx->stream));
    if (mine)
        SC_HIP(ctx, hipMemcpyAsync(own, (const char*)ctx->st_pairs.p + 16, 8 * mine, hipMemcpyDeviceToDevice, ctx->stream));
    if (comm) SC_NCCL(ctx, ncclAllGather(own, all, 2 * slot, ncclUint32, comm, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->xch_n = (long long)(slot * (unsigned long long)nr);
    *n_union = ctx->xch_n;
    return SC_OK;
}
