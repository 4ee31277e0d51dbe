// pmx_comm.hip - the multi-GPU exchange steps of the path, inside the C ABI: one process per GPU, RCCL over xGMI.  gfx950.
//
// The reference has no collective: one process, one image pair, ROI tiles with a margin as its only scaling
// convention.  This build shards one pair two ways:
//   * over D, exactly, for pipelines without SGM: every rank builds the costs of its disparity slice, one packed (cost, index)
//     key per pixel goes through ONE ncclAllReduce(min, uint64), the rank that owns a pixel's winner refines it and the refined
//     values meet in one ncclAllReduce(sum) (zeros elsewhere: exact);
//   * over rows (tiles + the steps' margin, the reference's convention, optimization/optimization.py:43 + marge.py:86-101) for
//     everything, SGM included: no data-path collective, one ncclAllGather of the owned rows of the 2-D results.
// Every buffer a collective touches lives on the device and belongs to the context ("exchange buffers", PMX_XBUF_*): keys, NaN
// flags, refinement packs and the full-size result maps never visit the host between the kernels that fill them, the
// collective and the kernels that read them.  librccl.so is loaded on first use (dlopen), so a single-GPU process never pays for
// it.  pmx_xbuf_download / pmx_xbuf_upload exist for ONE purpose: a test transport that lets two ranks share the single GPU of a
// test box (RCCL refuses two ranks on one device); the product path never calls them.
#include <dlfcn.h>

#include <cstring>

#include <rccl/rccl.h>

#include "pmx_internal.h"

struct pmx_comm {
    void* lib = nullptr;
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

#define PMX_NCCL(c, expr)                                                                                       \
    do {                                                                                                        \
        ncclResult_t r_ = (expr);                                                                               \
        if (r_ != ncclSuccess) {                                                                                \
            pmx_set_error("%s failed: %s (%s:%d)", #expr, (c)->GetErrorString ? (c)->GetErrorString(r_) : "?", __FILE__, __LINE__); \
            return PMX_ERR_HIP;                                                                                 \
        }                                                                                                       \
    } while (0)

static pmx_comm* g_loader = nullptr;  // function table shared by the contexts of a process

static int load_rccl(pmx_comm** out) {
    if (!g_loader) {
        pmx_comm* c = new pmx_comm;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)
            if (!c->lib) c->lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!c->lib) {
            pmx_set_error("pmx_comm: cannot load librccl.so (%s)", dlerror());
            delete c;
            return PMX_ERR_UNSUPPORTED;
        }
#define PMX_SYM(field, name)                                                       \
    c->field = (decltype(c->field))dlsym(c->lib, name);                            \
    if (!c->field) {                                                               \
        pmx_set_error("pmx_comm: librccl.so lacks %s", name);                      \
        return PMX_ERR_UNSUPPORTED;                                                \
    }
        PMX_SYM(GetUniqueId, "ncclGetUniqueId")
        PMX_SYM(CommInitRank, "ncclCommInitRank")
        PMX_SYM(CommDestroy, "ncclCommDestroy")
        PMX_SYM(CommCount, "ncclCommCount")
        PMX_SYM(AllReduce, "ncclAllReduce")
        PMX_SYM(AllGather, "ncclAllGather")
        PMX_SYM(ReduceScatter, "ncclReduceScatter")
        PMX_SYM(Send, "ncclSend")
        PMX_SYM(Recv, "ncclRecv")
        PMX_SYM(GroupStart, "ncclGroupStart")
        PMX_SYM(GroupEnd, "ncclGroupEnd")
        PMX_SYM(GetErrorString, "ncclGetErrorString")
#undef PMX_SYM
        g_loader = c;
    }
    *out = g_loader;
    return PMX_OK;
}

extern "C" int pmx_comm_unique_id(void* id_out, size_t bytes) {
    PMX_CHECK(id_out && bytes >= sizeof(ncclUniqueId), PMX_ERR_ARG, "pmx_comm_unique_id: need a buffer of %zu bytes", sizeof(ncclUniqueId));
    pmx_comm* l = nullptr;
    int rc = load_rccl(&l);
    if (rc) return rc;
    ncclUniqueId id;
    PMX_NCCL(l, l->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return PMX_OK;
}

extern "C" int pmx_comm_init(pmx_ctx* ctx, const void* id, size_t bytes, int world, int rank) {
    PMX_CHECK(ctx && id && bytes >= sizeof(ncclUniqueId), PMX_ERR_ARG, "pmx_comm_init: bad argument");
    PMX_CHECK(world >= 1 && rank >= 0 && rank < world, PMX_ERR_ARG, "pmx_comm_init: rank %d of %d", rank, world);
    PMX_CHECK(!ctx->comm, PMX_ERR_STATE, "pmx_comm_init: the context already has a communicator");
    PMX_HIP(hipSetDevice(ctx->device));
    pmx_comm* l = nullptr;
    int rc = load_rccl(&l);
    if (rc) return rc;
    pmx_comm* c = new pmx_comm(*l);
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclResult_t r = c->CommInitRank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) {
        pmx_set_error("ncclCommInitRank failed: %s", c->GetErrorString(r));
        delete c;
        return PMX_ERR_HIP;
    }
    c->world = world;
    c->rank = rank;
    ctx->comm = c;
    return PMX_OK;
}

extern "C" int pmx_comm_destroy(pmx_ctx* ctx) {
    PMX_CHECK(ctx, PMX_ERR_ARG, "pmx_comm_destroy: null context");
    if (!ctx->comm) return PMX_OK;
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm_stream) {
        (void)hipStreamSynchronize(ctx->comm_stream);
        (void)hipStreamDestroy(ctx->comm_stream);
        ctx->comm_stream = nullptr;
    }
    if (ctx->placed_ev) (void)hipEventDestroy(ctx->placed_ev);
    if (ctx->gathered_ev) (void)hipEventDestroy(ctx->gathered_ev);
    ctx->placed_ev = ctx->gathered_ev = nullptr;
    ctx->gather_pending = false;
    if (ctx->comm->comm) ctx->comm->CommDestroy(ctx->comm->comm);
    delete ctx->comm;
    ctx->comm = nullptr;
    return PMX_OK;
}

extern "C" int pmx_comm_info(const pmx_ctx* ctx, int* world, int* rank) {
    PMX_CHECK(ctx, PMX_ERR_ARG, "pmx_comm_info: null context");
    if (world) *world = ctx->comm ? ctx->comm->world : 1;
    if (rank) *rank = ctx->comm ? ctx->comm->rank : 0;
    return PMX_OK;
}

extern "C" int pmx_comm_count(const pmx_ctx* ctx, int* nranks) {
    PMX_CHECK(ctx && nranks, PMX_ERR_ARG, "pmx_comm_count: null argument");
    *nranks = 1;
    if (ctx->comm && ctx->comm->comm) PMX_NCCL(ctx->comm, ctx->comm->CommCount(ctx->comm->comm, nranks));
    return PMX_OK;
}

// ---- exchange buffers ------------------------------------------------------------------------------------------------
static const size_t kElem[PMX_XBUF_COUNT] = {8, 1, 4, 8, 4, 8, 4, 8, 2};

static size_t xbuf_count(const pmx_ctx* ctx, int which) {
    const size_t npix = (size_t)ctx->H * ctx->W, nfull = (size_t)ctx->full_H * ctx->W;
    switch (which) {
        case PMX_XBUF_KEYS: case PMX_XBUF_NANPIX: case PMX_XBUF_REFINE_FLAGS: return npix;
        case PMX_XBUF_REFINE_PACK: return 4 * npix;
        case PMX_XBUF_FULL_DISP: case PMX_XBUF_FULL_VALIDITY: case PMX_XBUF_FULL_ITP: case PMX_XBUF_FULL_VALIDITY16: return nfull;
        case PMX_XBUF_SCALARS: return 8;
        default: return 0;
    }
}

static int xbuf_need(pmx_ctx* ctx, int which) {
    PMX_CHECK(which >= 0 && which < PMX_XBUF_COUNT, PMX_ERR_ARG, "exchange buffer %d does not exist", which);
    const size_t bytes = xbuf_count(ctx, which) * kElem[which];
    PMX_CHECK(bytes > 0, PMX_ERR_STATE, "exchange buffer %d has no size yet (resident pair / pmx_tile_place first)", which);
    if (ctx->xbuf_bytes[which] < bytes) {
        if (ctx->xbuf[which]) PMX_HIP(hipFree(ctx->xbuf[which]));
        ctx->xbuf[which] = nullptr;
        ctx->xbuf_bytes[which] = 0;
        PMX_HIP(hipMalloc(&ctx->xbuf[which], bytes));
        ctx->xbuf_bytes[which] = bytes;
    }
    return PMX_OK;
}

void pmx_comm_release(pmx_ctx* ctx) {
    for (int i = 0; i < PMX_XBUF_COUNT; ++i) {
        if (ctx->xbuf[i]) (void)hipFree(ctx->xbuf[i]);
        ctx->xbuf[i] = nullptr;
        ctx->xbuf_bytes[i] = 0;
    }
    for (void*& p : ctx->refine_saved) {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    ctx->refine_saved_bytes = 0;
}

extern "C" int pmx_xbuf_info(pmx_ctx* ctx, int which, size_t* count, int* elem_bytes) {
    PMX_CHECK(ctx && which >= 0 && which < PMX_XBUF_COUNT, PMX_ERR_ARG, "pmx_xbuf_info: bad argument");
    if (count) *count = xbuf_count(ctx, which);
    if (elem_bytes) *elem_bytes = (int)kElem[which];
    return PMX_OK;
}

extern "C" int pmx_xbuf_download(pmx_ctx* ctx, int which, void* host) {
    PMX_CHECK(ctx && host, PMX_ERR_ARG, "pmx_xbuf_download: null argument");
    PMX_HIP(hipSetDevice(ctx->device));
    int rc = xbuf_need(ctx, which);
    if (rc) return rc;
    if ((rc = pmx_comm_join(ctx))) return rc;
    PMX_HIP(hipMemcpyAsync(host, ctx->xbuf[which], xbuf_count(ctx, which) * kElem[which], hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
}

extern "C" int pmx_xbuf_upload(pmx_ctx* ctx, int which, const void* host) {
    PMX_CHECK(ctx && host, PMX_ERR_ARG, "pmx_xbuf_upload: null argument");
    PMX_HIP(hipSetDevice(ctx->device));
    int rc = xbuf_need(ctx, which);
    if (rc) return rc;
    if ((rc = pmx_comm_join(ctx))) return rc;
    PMX_HIP(hipMemcpyAsync(ctx->xbuf[which], host, xbuf_count(ctx, which) * kElem[which], hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
}

// A gather queued on the communication stream (pmx_comm_gather_rows) reads and writes the full-size maps and uses the
// communicator: whatever touches either on the context's own stream waits for it first (an event wait on the device, the host
// goes on).
int pmx_comm_join(pmx_ctx* ctx) {
    if (ctx->gather_pending) PMX_HIP(hipStreamWaitEvent(ctx->stream, ctx->gathered_ev, 0));
    return PMX_OK;
}

// ---- collectives on exchange buffers -----------------------------------------------------------------------------------
extern "C" int pmx_comm_allreduce(pmx_ctx* ctx, int which, int op) {
    PMX_CHECK(ctx && ctx->comm, PMX_ERR_STATE, "pmx_comm_allreduce: no communicator (pmx_comm_init)");
    PMX_HIP(hipSetDevice(ctx->device));
    int rc = xbuf_need(ctx, which);
    if (rc) return rc;
    ncclDataType_t dt;
    switch (which) {
        case PMX_XBUF_KEYS: dt = ncclUint64; break;
        case PMX_XBUF_NANPIX: dt = ncclUint8; break;
        case PMX_XBUF_REFINE_PACK: dt = ncclFloat32; break;
        case PMX_XBUF_REFINE_FLAGS: dt = ncclInt64; break;
        case PMX_XBUF_SCALARS: dt = ncclFloat64; break;
        default: pmx_set_error("pmx_comm_allreduce: exchange buffer %d is gathered, not reduced", which); return PMX_ERR_ARG;
    }
    PMX_CHECK(op == PMX_OP_MIN || op == PMX_OP_SUM || op == PMX_OP_MAX, PMX_ERR_ARG, "pmx_comm_allreduce: unknown operation %d", op);
    const ncclRedOp_t ro = op == PMX_OP_MIN ? ncclMin : op == PMX_OP_SUM ? ncclSum : ncclMax;
    if ((rc = pmx_comm_join(ctx))) return rc;
    pmx_stage_scope t(ctx, PMX_STAGE_COLLECTIVE);
    PMX_NCCL(ctx->comm, ctx->comm->AllReduce(ctx->xbuf[which], ctx->xbuf[which], xbuf_count(ctx, which), dt, ro, ctx->comm->comm, ctx->stream));
    return PMX_OK;
}

// eight doubles per rank through RCCL (timings, counters): host in, host out
extern "C" int pmx_comm_allreduce_scalars(pmx_ctx* ctx, double* inout8, int op) {
    PMX_CHECK(ctx && inout8, PMX_ERR_ARG, "pmx_comm_allreduce_scalars: null argument");
    if (!ctx->comm) return PMX_OK;  // one rank: identity
    int rc = pmx_xbuf_upload(ctx, PMX_XBUF_SCALARS, inout8);
    if (rc) return rc;
    rc = pmx_comm_allreduce(ctx, PMX_XBUF_SCALARS, op);
    if (rc) return rc;
    return pmx_xbuf_download(ctx, PMX_XBUF_SCALARS, inout8);
}

static void shard_rows(int n, int world, int rank, int* lo, int* hi) {  // pandora_amd.dist.shard_range
    const int base = n / world, rem = n % world;
    *lo = rank * base + (rank < rem ? rank : rem);
    *hi = *lo + base + (rank < rem ? 1 : 0);
}

// Every rank holds its owned rows (shard_rows of full_H) of the three full-size maps in place; afterwards every rank holds all
// rows.  Rows divide evenly: in-place ncclAllGather straight into the maps.  Otherwise: gather padded slices into the pool and
// copy them into place (device to device).
extern "C" int pmx_comm_allgather_rows(pmx_ctx* ctx, int with_itp) {
    PMX_CHECK(ctx && ctx->comm, PMX_ERR_STATE, "pmx_comm_allgather_rows: no communicator (pmx_comm_init)");
    PMX_CHECK(ctx->full_H > 0, PMX_ERR_STATE, "pmx_comm_allgather_rows: pmx_tile_place first");
    PMX_HIP(hipSetDevice(ctx->device));
    if (int jr = pmx_comm_join(ctx)) return jr;
    pmx_comm* c = ctx->comm;
    const int which[3] = {PMX_XBUF_FULL_DISP, PMX_XBUF_FULL_VALIDITY, PMX_XBUF_FULL_ITP};
    const int H = ctx->full_H, W = ctx->W;
    int lo, hi;
    shard_rows(H, c->world, c->rank, &lo, &hi);
    pmx_stage_scope t(ctx, PMX_STAGE_COLLECTIVE);
    for (int m = 0; m < (with_itp ? 3 : 2); ++m) {
        int rc = xbuf_need(ctx, which[m]);
        if (rc) return rc;
        const size_t es = kElem[which[m]], row = (size_t)W * es;
        char* full = (char*)ctx->xbuf[which[m]];
        if (H % c->world == 0) {
            PMX_NCCL(c, c->AllGather(full + (size_t)lo * row, full, (size_t)(hi - lo) * row, ncclInt8, c->comm, ctx->stream));
        } else {
            const size_t slot = (size_t)(H / c->world + 1) * row;
            char* stage = nullptr;
            PMX_HIP(pmx_pool_alloc(ctx, (void**)&stage, slot * (c->world + 1)));
            struct stage_guard {  // the staging block goes back to the pool on every way out (stream-ordered reuse)
                pmx_ctx* ctx; char* p;
                ~stage_guard() { pmx_pool_free(ctx, p); }
            } guard{ctx, stage};
            PMX_HIP(hipMemcpyAsync(stage + slot * c->world, full + (size_t)lo * row, (size_t)(hi - lo) * row, hipMemcpyDeviceToDevice, ctx->stream));
            PMX_NCCL(c, c->AllGather(stage + slot * c->world, stage, slot, ncclInt8, c->comm, ctx->stream));
            for (int r = 0; r < c->world; ++r) {
                int rlo, rhi;
                shard_rows(H, c->world, r, &rlo, &rhi);
                if (r != c->rank)
                    PMX_HIP(hipMemcpyAsync(full + (size_t)rlo * row, stage + slot * r, (size_t)(rhi - rlo) * row, hipMemcpyDeviceToDevice, ctx->stream));
            }
        }
    }
    return PMX_OK;
}

// int64 validity bits <-> the 16 bits they need (the reference stores the mask as uint16, common.py:160-170): what travels
__global__ __launch_bounds__(256) void narrow_validity_kernel(const int64_t* __restrict__ in, size_t n, uint16_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (uint16_t)in[i];
}
__global__ __launch_bounds__(256) void widen_validity_kernel(const uint16_t* __restrict__ in, size_t n, int64_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (int64_t)in[i];
}

// Every rank holds its owned rows of the full-size maps; afterwards rank `root` holds all rows (the other ranks' maps are
// unchanged).  One group of ncclSend / ncclRecv: every peer sends straight to the root over its own xGMI link, which an
// all-gather's ring would not do.  The validity mask travels as 16 bits per pixel: 10 B/pixel in all (6 without the coefficient).
extern "C" int pmx_comm_gather_rows(pmx_ctx* ctx, int root, int with_itp) {
    PMX_CHECK(ctx && ctx->comm, PMX_ERR_STATE, "pmx_comm_gather_rows: no communicator (pmx_comm_init)");
    PMX_CHECK(ctx->full_H > 0, PMX_ERR_STATE, "pmx_comm_gather_rows: pmx_tile_place first");
    pmx_comm* c = ctx->comm;
    PMX_CHECK(root >= 0 && root < c->world, PMX_ERR_ARG, "pmx_comm_gather_rows: root %d of %d ranks", root, c->world);
    PMX_HIP(hipSetDevice(ctx->device));
    const int H = ctx->full_H, W = ctx->W;
    int lo, hi;
    shard_rows(H, c->world, c->rank, &lo, &hi);
    int rc = xbuf_need(ctx, PMX_XBUF_FULL_VALIDITY16);
    if (!rc) rc = xbuf_need(ctx, PMX_XBUF_FULL_VALIDITY);
    if (!rc) rc = xbuf_need(ctx, PMX_XBUF_FULL_DISP);
    if (!rc && with_itp) rc = xbuf_need(ctx, PMX_XBUF_FULL_ITP);
    if (rc) return rc;
    // On a stream of its own, ordered by events: it starts when this rank's rows are in the full-size maps (whatever is queued on
    // the context's stream so far) and whoever touches those maps or the communicator next waits for it (pmx_comm_join) - the
    // kernels of the NEXT pair do neither and run under the transfer.  PMX_COMM_OVERLAP=0: on the context's stream, as a plain
    // sequence.
    const char* eo = pmx_opt(ctx, "COMM_OVERLAP");
    const bool overlap = !(eo && eo[0] == '0');
    hipStream_t st = ctx->stream;
    if (overlap) {
        if (!ctx->comm_stream) {
            PMX_HIP(hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
            PMX_HIP(hipEventCreateWithFlags(&ctx->placed_ev, hipEventDisableTiming));
            PMX_HIP(hipEventCreateWithFlags(&ctx->gathered_ev, hipEventDisableTiming));
        }
        st = ctx->comm_stream;
        PMX_HIP(hipEventRecord(ctx->placed_ev, ctx->stream));
        PMX_HIP(hipStreamWaitEvent(st, ctx->placed_ev, 0));  // (gathers follow one another on this stream anyway)
    } else if ((rc = pmx_comm_join(ctx))) {
        return rc;
    }
    {
        pmx_stage_scope t(ctx, PMX_STAGE_COLLECTIVE, st);
        uint16_t* v16 = (uint16_t*)ctx->xbuf[PMX_XBUF_FULL_VALIDITY16];
        int64_t* v64 = (int64_t*)ctx->xbuf[PMX_XBUF_FULL_VALIDITY];
        const size_t own_n = (size_t)(hi - lo) * W;
        if (c->rank != root)
            hipLaunchKernelGGL(narrow_validity_kernel, dim3((unsigned)((own_n + 255) / 256)), dim3(256), 0, st, v64 + (size_t)lo * W, own_n,
                               v16 + (size_t)lo * W);
        struct { void* p; size_t es; } maps[3] = {{ctx->xbuf[PMX_XBUF_FULL_DISP], 4}, {v16, 2}, {with_itp ? ctx->xbuf[PMX_XBUF_FULL_ITP] : nullptr, 4}};
        PMX_NCCL(c, c->GroupStart());
        for (auto& m : maps) {
            if (!m.p) continue;
            if (c->rank == root) {
                for (int r = 0; r < c->world; ++r) {
                    if (r == root) continue;
                    int rlo, rhi;
                    shard_rows(H, c->world, r, &rlo, &rhi);
                    PMX_NCCL(c, c->Recv((char*)m.p + (size_t)rlo * W * m.es, (size_t)(rhi - rlo) * W * m.es, ncclInt8, r, c->comm, st));
                }
            } else {
                PMX_NCCL(c, c->Send((const char*)m.p + (size_t)lo * W * m.es, own_n * m.es, ncclInt8, root, c->comm, st));
            }
        }
        PMX_NCCL(c, c->GroupEnd());
        if (c->rank == root) {  // the peers' 16-bit rows into the int64 map (the root's own rows are already there)
            for (int r = 0; r < c->world; ++r) {
                if (r == root) continue;
                int rlo, rhi;
                shard_rows(H, c->world, r, &rlo, &rhi);
                const size_t n = (size_t)(rhi - rlo) * W;
                hipLaunchKernelGGL(widen_validity_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, v16 + (size_t)rlo * W, n,
                                   v64 + (size_t)rlo * W);
            }
        }
        PMX_HIP(hipGetLastError());
    }
    if (overlap) {
        PMX_HIP(hipEventRecord(ctx->gathered_ev, st));
        ctx->gather_pending = true;
    }
    return PMX_OK;
}

// ---- row tiles -----------------------------------------------------------------------------------------------------------
// The context's maps (disp, validity, itp) are those of a tile that starts at image row tile_lo; its rows [own_lo, own_hi) of
// the image go to their place in the full-size maps (full_H rows).
extern "C" int pmx_tile_place(pmx_ctx* ctx, int full_H, int own_lo, int own_hi, int tile_lo, int with_itp) {
    PMX_CHECK(ctx && ctx->left && ctx->disp_ready, PMX_ERR_STATE, "pmx_tile_place: no disparity map resident");
    PMX_CHECK(full_H > 0 && 0 <= tile_lo && tile_lo <= own_lo && own_lo < own_hi && own_hi <= full_H && own_hi - tile_lo <= ctx->H, PMX_ERR_ARG,
              "pmx_tile_place: rows [%d, %d) of %d do not lie inside the tile [%d, %d)", own_lo, own_hi, full_H, tile_lo, tile_lo + ctx->H);
    PMX_HIP(hipSetDevice(ctx->device));
    if (int jr = pmx_comm_join(ctx)) return jr;  // (a gather still reading / writing the full-size maps)
    ctx->full_H = full_H;
    const void* src[3] = {ctx->disp, ctx->validity, ctx->itp};
    const int which[3] = {PMX_XBUF_FULL_DISP, PMX_XBUF_FULL_VALIDITY, PMX_XBUF_FULL_ITP};
    for (int m = 0; m < (with_itp ? 3 : 2); ++m) {
        int rc = xbuf_need(ctx, which[m]);
        if (rc) return rc;
        const size_t row = (size_t)ctx->W * kElem[which[m]];
        PMX_HIP(hipMemcpyAsync((char*)ctx->xbuf[which[m]] + (size_t)own_lo * row, (const char*)src[m] + (size_t)(own_lo - tile_lo) * row,
                               (size_t)(own_hi - own_lo) * row, hipMemcpyDeviceToDevice, ctx->stream));
    }
    return PMX_OK;
}

// the same from host rows (a machine-level run ends with its maps on the host: filters, validation)
extern "C" int pmx_set_full_rows(pmx_ctx* ctx, int full_H, int own_lo, int own_hi, const float* disp, const int64_t* validity,
                                 const float* itp) {
    PMX_CHECK(ctx && ctx->left && disp && validity, PMX_ERR_ARG, "pmx_set_full_rows: bad argument");
    PMX_CHECK(full_H > 0 && 0 <= own_lo && own_lo < own_hi && own_hi <= full_H, PMX_ERR_ARG, "pmx_set_full_rows: rows [%d, %d) of %d", own_lo,
              own_hi, full_H);
    PMX_HIP(hipSetDevice(ctx->device));
    if (int jr = pmx_comm_join(ctx)) return jr;
    ctx->full_H = full_H;
    const void* src[3] = {disp, validity, itp};
    const int which[3] = {PMX_XBUF_FULL_DISP, PMX_XBUF_FULL_VALIDITY, PMX_XBUF_FULL_ITP};
    for (int m = 0; m < 3; ++m) {
        if (!src[m]) continue;
        int rc = xbuf_need(ctx, which[m]);
        if (rc) return rc;
        const size_t row = (size_t)ctx->W * kElem[which[m]];
        PMX_HIP(hipMemcpyAsync((char*)ctx->xbuf[which[m]] + (size_t)own_lo * row, src[m], (size_t)(own_hi - own_lo) * row, hipMemcpyHostToDevice,
                               ctx->stream));
    }
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
}

extern "C" int pmx_get_full_maps(pmx_ctx* ctx, float* disp, int64_t* validity, float* itp) {
    PMX_CHECK(ctx && ctx->full_H > 0, PMX_ERR_STATE, "pmx_get_full_maps: pmx_tile_place first");
    PMX_HIP(hipSetDevice(ctx->device));
    if (int jr = pmx_comm_join(ctx)) return jr;
    void* dst[3] = {disp, validity, itp};
    const int which[3] = {PMX_XBUF_FULL_DISP, PMX_XBUF_FULL_VALIDITY, PMX_XBUF_FULL_ITP};
    for (int m = 0; m < 3; ++m)
        if (dst[m]) {
            PMX_CHECK(ctx->xbuf[which[m]], PMX_ERR_STATE, "pmx_get_full_maps: map %d was never placed", m);
            PMX_HIP(hipMemcpyAsync(dst[m], ctx->xbuf[which[m]], xbuf_count(ctx, which[m]) * kElem[which[m]], hipMemcpyDeviceToHost, ctx->stream));
        }
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return pmx_check_async_error(ctx, "pmx_get_full_maps");
}

// ---- D shards ------------------------------------------------------------------------------------------------------------
extern "C" int pmx_shard_minkey(pmx_ctx* ctx, const pmx_cv* cv, int is_max, int global_index_offset) {
    PMX_CHECK(ctx && cv && cv->ctx == ctx, PMX_ERR_ARG, "pmx_shard_minkey: bad argument");
    PMX_HIP(hipSetDevice(ctx->device));
    int rc = xbuf_need(ctx, PMX_XBUF_KEYS);
    if (rc) return rc;
    return pmx_wta_minkey(ctx, cv, is_max, global_index_offset, (uint64_t*)ctx->xbuf[PMX_XBUF_KEYS]);
}

extern "C" int pmx_shard_from_keys(pmx_ctx* ctx, double d0_global, int subpix, float invalid_disparity) {
    PMX_CHECK(ctx && ctx->xbuf[PMX_XBUF_KEYS], PMX_ERR_STATE, "pmx_shard_from_keys: pmx_shard_minkey first");
    return pmx_wta_from_keys(ctx, (const uint64_t*)ctx->xbuf[PMX_XBUF_KEYS], d0_global, subpix, invalid_disparity);
}

// pixels that are NaN for every disparity of THIS shard -> exchange buffer (1 / 0); a MIN all-reduce makes it "of every shard"
extern "C" int pmx_shard_nan_pixels(pmx_ctx* ctx, const pmx_cv* cv) {
    PMX_CHECK(ctx && cv && cv->ctx == ctx, PMX_ERR_ARG, "pmx_shard_nan_pixels: bad argument");
    PMX_HIP(hipSetDevice(ctx->device));
    int rc = xbuf_need(ctx, PMX_XBUF_NANPIX);
    if (rc) return rc;
    if (cv->repr == PMX_REPR_CENSUS_DEFERRED || cv->repr == PMX_REPR_SGM_U8X8) return pmx_launch_census_nan_pixels(ctx, cv, (uint8_t*)ctx->xbuf[PMX_XBUF_NANPIX]);
    rc = pmx_cv_materialize(ctx, const_cast<pmx_cv*>(cv));
    if (rc) return rc;
    return pmx_launch_nan_pixels(ctx, cv, (uint8_t*)ctx->xbuf[PMX_XBUF_NANPIX]);
}

constexpr int64_t kMskInvalid = 0x3C3;  // constants.py:31

// pack[0] refined disparity, [1] interpolated coefficient (NaN sent as 0 + a flag in [2]), [3] 1 = this rank owns the pixel
__global__ __launch_bounds__(256) void refine_pack_kernel(const float* __restrict__ disp0, const int64_t* __restrict__ val0,
                                                          const float* __restrict__ rdisp, const int64_t* __restrict__ rval,
                                                          const float* __restrict__ ritp, size_t npix, float own_lo, float own_hi,
                                                          int last, float* __restrict__ pack, int64_t* __restrict__ flags) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const float d = disp0[i];
    const int64_t v = val0[i];
    const bool own = d >= own_lo && (last ? d <= own_hi : d < own_hi + 1.0f) && (v & kMskInvalid) == 0;
    const float it = ritp[i];
    const bool itnan = it != it;
    pack[i] = own ? rdisp[i] : 0.f;
    pack[npix + i] = (own && !itnan) ? it : 0.f;
    pack[2 * npix + i] = (own && itnan) ? 1.f : 0.f;
    pack[3 * npix + i] = own ? 1.f : 0.f;
    flags[i] = own ? rval[i] - v : 0;
}

__global__ __launch_bounds__(256) void refine_unpack_kernel(const float* __restrict__ disp0, const int64_t* __restrict__ val0,
                                                            const float* __restrict__ pack, const int64_t* __restrict__ flags,
                                                            size_t npix, float* __restrict__ disp, int64_t* __restrict__ validity,
                                                            float* __restrict__ itp) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const bool owned = pack[3 * npix + i] > 0.f;
    disp[i] = owned ? pack[i] : disp0[i];
    itp[i] = owned ? (pack[2 * npix + i] > 0.f ? __int_as_float(0x7fc00000) : pack[npix + i]) : __int_as_float(0x7fc00000);
    validity[i] = val0[i] + flags[i];
}

// The merged winner map is resident on every rank (pmx_shard_from_keys).  This rank refines the pixels whose winner lies in the
// disparities it owns, [own_lo, own_hi] (the last rank includes own_hi + sub-pixel steps below own_hi + 1 otherwise), and packs
// value-or-zero maps for the SUM all-reduce; pmx_shard_refine_unpack writes the merged maps back into the context.
extern "C" int pmx_shard_refine_pack(pmx_ctx* ctx, const pmx_cv* cv, int method, int is_max, double own_lo, double own_hi, int last) {
    PMX_CHECK(ctx && cv && cv->ctx == ctx && ctx->disp_ready, PMX_ERR_STATE, "pmx_shard_refine_pack: no merged disparity map resident");
    PMX_HIP(hipSetDevice(ctx->device));
    const size_t npix = (size_t)ctx->H * ctx->W;
    int rc = xbuf_need(ctx, PMX_XBUF_REFINE_PACK);
    if (!rc) rc = xbuf_need(ctx, PMX_XBUF_REFINE_FLAGS);
    if (rc) return rc;
    if (ctx->refine_saved_bytes < npix * 8) {
        for (void*& p : ctx->refine_saved) {
            if (p) PMX_HIP(hipFree(p));
            p = nullptr;
        }
        PMX_HIP(hipMalloc(&ctx->refine_saved[0], npix * 4));
        PMX_HIP(hipMalloc(&ctx->refine_saved[1], npix * 8));
        ctx->refine_saved_bytes = npix * 8;
    }
    PMX_HIP(hipMemcpyAsync(ctx->refine_saved[0], ctx->disp, npix * 4, hipMemcpyDeviceToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(ctx->refine_saved[1], ctx->validity, npix * 8, hipMemcpyDeviceToDevice, ctx->stream));
    rc = pmx_refine(ctx, cv, method, is_max);  // leaves winners outside the local volume alone
    if (rc) return rc;
    hipLaunchKernelGGL(refine_pack_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, ctx->stream, (const float*)ctx->refine_saved[0],
                       (const int64_t*)ctx->refine_saved[1], ctx->disp, ctx->validity, ctx->itp, npix, (float)own_lo, (float)own_hi, last,
                       (float*)ctx->xbuf[PMX_XBUF_REFINE_PACK], (int64_t*)ctx->xbuf[PMX_XBUF_REFINE_FLAGS]);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

extern "C" int pmx_shard_refine_unpack(pmx_ctx* ctx) {
    PMX_CHECK(ctx && ctx->refine_saved[0] && ctx->xbuf[PMX_XBUF_REFINE_PACK], PMX_ERR_STATE, "pmx_shard_refine_unpack: pmx_shard_refine_pack first");
    PMX_HIP(hipSetDevice(ctx->device));
    const size_t npix = (size_t)ctx->H * ctx->W;
    hipLaunchKernelGGL(refine_unpack_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, ctx->stream, (const float*)ctx->refine_saved[0],
                       (const int64_t*)ctx->refine_saved[1], (const float*)ctx->xbuf[PMX_XBUF_REFINE_PACK],
                       (const int64_t*)ctx->xbuf[PMX_XBUF_REFINE_FLAGS], npix, ctx->disp, ctx->validity, ctx->itp);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}
