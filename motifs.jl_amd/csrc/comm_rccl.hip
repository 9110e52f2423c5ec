// comm_rccl.hip — the collectives of the data-parallel path behind the C ABI (include/motifs_hip.h, "multi-GPU").
//
// The reference is single-GPU: src/MOTIFs.jl:4-8 imports no communication package, so there is no call site to
// mirror.  What the sharded path exchanges is fixed by the algorithm: reads shard in contiguous
// blocks, the parameters and the PWM bank are replicated, and per optimiser step ONE sum of the flat gradient
// [dD | dF | dvecs] (124 833 floats at BASELINE configs[1]) crosses the links, per scan ONE sum of K int64 hit
// counts per strand.  Both are far below the size at which xGMI's per-link rate matters (0.5 MB = a few
// microseconds of wire time on a 7 x ~153 GB/s fabric): they are latency-bound, so each is a single
// ncclAllReduce on the context's stream, queued directly behind the kernel that produced the operand — no bucketing,
// no second stream, no host wait.
//
// RCCL is opened with dlopen on first use: a single-GPU user never needs librccl, and a host process that has
// already loaded one (torch ships its own copy) gets that copy rather than a second one.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "api_common.h"

using namespace motifs;

struct motifs_comm {
    motifs_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
};

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    char why[256] = "";
};

Rccl g_rccl;
std::once_flag g_once;

void open_rccl() {
    Rccl& r = g_rccl;
    // a copy the process already holds first (RTLD_NOLOAD), then the loader's search path, then the ROCm tree
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (r.handle) break;
    }
    for (const char* n : names) {
        if (r.handle) break;
        r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!r.handle) {
        snprintf(r.why, sizeof(r.why), "librccl.so could not be opened: %s", dlerror());
        return;
    }
    auto sym = [&](const char* name) -> void* {
        void* p = dlsym(r.handle, name);
        if (!p && !r.why[0]) snprintf(r.why, sizeof(r.why), "librccl.so lacks %s", name);
        return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
}

int need_rccl(const char* who) {
    std::call_once(g_once, open_rccl);
    if (g_rccl.why[0]) {
        set_error("%s: %s", who, g_rccl.why);
        return MOTIFS_ERR_COMM;
    }
    return MOTIFS_OK;
}

int nccl_check(ncclResult_t e, const char* what) {
    if (e == ncclSuccess) return MOTIFS_OK;
    set_error("%s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "RCCL error");
    return MOTIFS_ERR_COMM;
}

int allreduce(motifs_comm* c, const void* send, void* recv, int64_t n, ncclDataType_t ty, const char* who) {
    if (!c || !c->ctx || n < 0 || (n > 0 && (!send || !recv))) {
        set_error("%s: bad argument", who);
        return MOTIFS_ERR_INVALID;
    }
    if (n == 0) return MOTIFS_OK;
    int r = need_rccl(who);
    if (r) return r;
    MOTIFS_HIP_CHECK(hipSetDevice(c->ctx->device));
    return nccl_check(g_rccl.AllReduce(send, recv, (size_t)n, ty, ncclSum, c->comm, c->ctx->stream), who);
}

thread_local int g_group_depth = 0;     // ncclGroupStart / ncclGroupEnd are per thread

}  // namespace

int motifs::comm_group_depth() { return g_group_depth; }
motifs_ctx* motifs::comm_ctx(motifs_comm* c) { return c ? c->ctx : nullptr; }

extern "C" {

int motifs_comm_unique_id(uint8_t id[MOTIFS_COMM_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) == MOTIFS_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (!id) return MOTIFS_ERR_INVALID;
    int r = need_rccl("motifs_comm_unique_id");
    if (r) return r;
    ncclUniqueId u;
    r = nccl_check(g_rccl.GetUniqueId(&u), "ncclGetUniqueId");
    if (r) return r;
    memcpy(id, &u, sizeof(u));
    return MOTIFS_OK;
}

int motifs_comm_create(motifs_ctx* ctx, const uint8_t id[MOTIFS_COMM_ID_BYTES], int nranks, int rank, motifs_comm** out) {
    if (!out) return MOTIFS_ERR_INVALID;
    *out = nullptr;
    if (!ctx || !id || nranks < 1 || rank < 0 || rank >= nranks) {
        set_error("motifs_comm_create: bad argument (nranks=%d rank=%d)", nranks, rank);
        return MOTIFS_ERR_INVALID;
    }
    int r = need_rccl("motifs_comm_create");
    if (r) return r;
    MOTIFS_HIP_CHECK(hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t comm = nullptr;
    r = nccl_check(g_rccl.CommInitRank(&comm, nranks, u, rank), "ncclCommInitRank");
    if (r) return r;
    motifs_comm* c = new motifs_comm();
    c->ctx = ctx;
    c->comm = comm;
    c->rank = rank;
    c->nranks = nranks;
    *out = c;
    return MOTIFS_OK;
}

int motifs_comm_create_all(motifs_ctx* const* ctxs, int n_dev, motifs_comm** out) {
    if (!ctxs || !out || n_dev < 1) {
        set_error("motifs_comm_create_all: bad argument (n_dev=%d)", n_dev);
        return MOTIFS_ERR_INVALID;
    }
    std::vector<int> devs(n_dev);
    for (int i = 0; i < n_dev; i++) {
        out[i] = nullptr;
        if (!ctxs[i]) {
            set_error("motifs_comm_create_all: ctxs[%d] is NULL", i);
            return MOTIFS_ERR_INVALID;
        }
        devs[i] = ctxs[i]->device;
        for (int j = 0; j < i; j++)
            if (devs[j] == devs[i]) {
                set_error("motifs_comm_create_all: device %d appears twice (one rank per device)", devs[i]);
                return MOTIFS_ERR_INVALID;
            }
    }
    int r = need_rccl("motifs_comm_create_all");
    if (r) return r;
    std::vector<ncclComm_t> comms(n_dev, nullptr);
    r = nccl_check(g_rccl.CommInitAll(comms.data(), n_dev, devs.data()), "ncclCommInitAll");
    if (r) return r;
    for (int i = 0; i < n_dev; i++) {
        motifs_comm* c = new motifs_comm();
        c->ctx = ctxs[i];
        c->comm = comms[i];
        c->rank = i;
        c->nranks = n_dev;
        out[i] = c;
    }
    return MOTIFS_OK;
}

void motifs_comm_destroy(motifs_comm* c) {
    if (!c) return;
    if (c->comm && g_rccl.CommDestroy) {
        (void)hipSetDevice(c->ctx->device);
        (void)hipStreamSynchronize(c->ctx->stream);
        (void)g_rccl.CommDestroy(c->comm);
    }
    delete c;
}

int motifs_comm_rank(motifs_comm* c, int* rank, int* nranks) {
    if (!c) return MOTIFS_ERR_INVALID;
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
    return MOTIFS_OK;
}

int motifs_comm_group_start(void) {
    int r = need_rccl("motifs_comm_group_start");
    if (r) return r;
    r = nccl_check(g_rccl.GroupStart(), "ncclGroupStart");
    if (r == MOTIFS_OK) g_group_depth++;
    return r;
}

int motifs_comm_group_end(void) {
    int r = need_rccl("motifs_comm_group_end");
    if (r) return r;
    if (g_group_depth <= 0) {
        set_error("motifs_comm_group_end without motifs_comm_group_start on this thread");
        return MOTIFS_ERR_INVALID;
    }
    g_group_depth--;
    return nccl_check(g_rccl.GroupEnd(), "ncclGroupEnd");
}

int motifs_comm_allreduce_sum_f32_dev(motifs_comm* c, float* buf_dev, int64_t n) {
    return allreduce(c, buf_dev, buf_dev, n, ncclFloat32, "motifs_comm_allreduce_sum_f32_dev");
}

int motifs_comm_allreduce_sum_i64_dev(motifs_comm* c, int64_t* buf_dev, int64_t n) {
    return allreduce(c, buf_dev, buf_dev, n, ncclInt64, "motifs_comm_allreduce_sum_i64_dev");
}

int motifs_comm_allreduce_sum_u32_dev(motifs_comm* c, uint32_t* buf_dev, int64_t n) {
    return allreduce(c, buf_dev, buf_dev, n, ncclUint32, "motifs_comm_allreduce_sum_u32_dev");
}

int motifs_comm_allreduce_sum_f32_to_dev(motifs_comm* c, const float* send_dev, float* recv_dev, int64_t n) {
    return allreduce(c, send_dev, recv_dev, n, ncclFloat32, "motifs_comm_allreduce_sum_f32_to_dev");
}

int motifs_hist_allreduce(motifs_comm* c, int64_t* per_pwm_counts_dev, int K, int n_strands) {
    if (K < 0 || n_strands < 1 || n_strands > 2) {
        set_error("motifs_hist_allreduce: bad argument (K=%d n_strands=%d)", K, n_strands);
        return MOTIFS_ERR_INVALID;
    }
    // (Round 5 ran this sum on a side stream behind an event recorded in front of emit_records, beside the record writes.  The timeline of a step says
    // no: an event record in the stream leaves a 15-20 us hole in front of the next kernel - as much as the sum's latency it was meant to hide.)
    return allreduce(c, per_pwm_counts_dev, per_pwm_counts_dev, (int64_t)K * n_strands, ncclInt64, "motifs_hist_allreduce");
}

}  // extern "C"
