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
