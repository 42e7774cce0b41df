// multigpu.hip — the multi-GPU entries of the C ABI (host code only).
//
// Windows are independent units (the reference's `while read chr start end` loops carry no state from one
// window to the next: run_tajd.sh:103-196, run_h-fst.sh:155-190, run_pica2_impg.sh:126-190), so the window list
// is cut into contiguous ranges, one per GPU, each GPU keeps only the slab of sites its windows touch, and the
// only exchange is ONE all-gather of the fixed-size per-window records.  Two forms:
//   * one process driving several devices: impop_scan_sharded (a context + a slab per device, launches on all
//     streams before the first fetch, records land in the caller's host array in global window order);
//   * one process per GPU: impop_comm_* + impop_gather* over ncclAllGather (RCCL over xGMI), plus
//     impop_allreduce_i64 for the one case with a reduction — the K-split Gram of a single giant window.
// RCCL is bound at run time (dlopen of librccl.so.1) and only when a communicator is created: a process that
// already holds a copy (PyTorch-ROCm ships its own under the same soname) keeps exactly that one, and the
// single-GPU CLIs do not pay for loading a collective library they never call.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "internal.h"

using namespace impop;

// ---- shard arithmetic (the rule of impop_amd/distributed.py:shard_range / shard_windows) ---------------------

IMPOP_API int impop_shard_range(uint64_t n_items, int n_shards, int shard, uint64_t *first, uint64_t *count) {
    REQUIRE(n_shards >= 1 && shard >= 0 && shard < n_shards, "impop_shard_range: shard %d of %d", shard, n_shards);
    REQUIRE(first && count, "impop_shard_range: NULL output");
    const uint64_t base = n_items / (uint64_t)n_shards, extra = n_items % (uint64_t)n_shards;
    *first = (uint64_t)shard * base + std::min<uint64_t>((uint64_t)shard, extra);  // the first `extra` shards hold one more
    *count = base + ((uint64_t)shard < extra ? 1 : 0);
    return IMPOP_OK;
}

IMPOP_API int impop_shard_windows(const impop_window *windows, uint64_t n_windows, int n_shards, int shard,
                                  uint64_t *first_window, uint64_t *n_shard_windows, uint64_t *slab_begin,
                                  uint64_t *slab_end) {
    REQUIRE(n_windows == 0 || windows, "impop_shard_windows: windows is NULL");
    uint64_t lo = 0, cnt = 0;
    int rc = impop_shard_range(n_windows, n_shards, shard, &lo, &cnt);
    if (rc) return rc;
    uint64_t s0 = 0, s1 = 0;
    if (cnt) {
        s0 = windows[lo].site_begin;
        s1 = windows[lo].site_end;
        for (uint64_t i = lo; i < lo + cnt; ++i) {  // sliding windows: neighbouring slabs overlap by the halo
            s0 = std::min(s0, windows[i].site_begin);
            s1 = std::max(s1, windows[i].site_end);
        }
    }
    if (first_window) *first_window = lo;
    if (n_shard_windows) *n_shard_windows = cnt;
    if (slab_begin) *slab_begin = s0;
    if (slab_end) *slab_end = s1;
    return IMPOP_OK;
}

// ---- one process, several devices ----------------------------------------------------------------------------

IMPOP_API int impop_scan_sharded(impop_ctx *const *ctxs, const impop_matrix *const *slabs, const uint64_t *slab_site_begin,
                                 int n_ctx, const impop_window *windows, uint64_t n_windows, const uint64_t *mask_p,
                                 const uint64_t *mask_a, const uint64_t *mask_b, const impop_scan_params *params,
                                 impop_window_stats *out_host) {
    REQUIRE(ctxs && slabs && slab_site_begin && n_ctx >= 1, "impop_scan_sharded: NULL argument or n_ctx < 1");
    REQUIRE(n_windows == 0 || (windows && out_host), "impop_scan_sharded: NULL windows/out");
    std::vector<impop_scan_plan *> plans((size_t)n_ctx, nullptr);
    std::vector<uint64_t> first((size_t)n_ctx, 0), count((size_t)n_ctx, 0);
    auto cleanup = [&]() {
        for (impop_scan_plan *p : plans) impop_scan_plan_destroy(p);
    };
    // 1. plans: shard k = windows [first_k, first_k + count_k) rebased to the coordinates of slab k
    for (int k = 0; k < n_ctx; ++k) {
        REQUIRE(ctxs[k] && slabs[k], "impop_scan_sharded: context or slab %d is NULL", k);
        uint64_t s0 = 0, s1 = 0;
        int rc = impop_shard_windows(windows, n_windows, n_ctx, k, &first[k], &count[k], &s0, &s1);
        if (rc) { cleanup(); return rc; }
        if (!count[k]) continue;
        uint64_t n_site = 0;
        impop_matrix_info(slabs[k], nullptr, &n_site, nullptr, nullptr);
        if (slabs[k]->compact || s0 < slab_site_begin[k] || s1 > slab_site_begin[k] + n_site) {
            cleanup();
            set_error("impop_scan_sharded: slab %d ([%llu, %llu)%s) does not cover its windows' sites [%llu, %llu)", k,
                      (unsigned long long)slab_site_begin[k], (unsigned long long)(slab_site_begin[k] + n_site),
                      slabs[k]->compact ? ", compacted" : "", (unsigned long long)s0, (unsigned long long)s1);
            return IMPOP_E_INVALID;
        }
        std::vector<impop_window> loc(windows + first[k], windows + first[k] + count[k]);
        for (impop_window &w : loc) { w.site_begin -= slab_site_begin[k]; w.site_end -= slab_site_begin[k]; }
        rc = impop_scan_plan_create(ctxs[k], slabs[k], loc.data(), count[k], mask_p, mask_a, mask_b, params, &plans[k]);
        if (rc) { cleanup(); return rc; }
    }
    // 2. every device starts its pass before any result is waited for
    for (int k = 0; k < n_ctx; ++k)
        if (plans[k]) {
            const int rc = impop_scan_plan_launch(plans[k], nullptr);
            if (rc) { cleanup(); return rc; }
        }
    // 3. records into the caller's array, global window order
    for (int k = 0; k < n_ctx; ++k)
        if (plans[k]) {
            const int rc = impop_scan_plan_fetch(plans[k], out_host + first[k]);
            if (rc) { cleanup(); return rc; }
        }
    cleanup();
    return IMPOP_OK;
}

// The all-pairs mode over several devices: impop_pairwise_scan waits for its own results (chunks of Gram matrices go
// through one scratch), so every context's shard runs on a host thread of its own; the devices work side by side.
IMPOP_API int impop_pairwise_scan_sharded(impop_ctx *const *ctxs, const impop_matrix *const *slabs, const uint64_t *slab_site_begin,
                                          int n_ctx, const impop_window *windows, uint64_t n_windows, const uint64_t *mask_p,
                                          const uint64_t *mask_a, const uint64_t *mask_b, const impop_pairwise_params *params,
                                          impop_pairwise_stats *out_host) {
    REQUIRE(ctxs && slabs && slab_site_begin && n_ctx >= 1 && params, "impop_pairwise_scan_sharded: NULL argument or n_ctx < 1");
    REQUIRE(n_windows == 0 || (windows && out_host), "impop_pairwise_scan_sharded: NULL windows/out");
    std::vector<uint64_t> first((size_t)n_ctx, 0), count((size_t)n_ctx, 0);
    std::vector<std::vector<impop_window>> loc((size_t)n_ctx);
    for (int k = 0; k < n_ctx; ++k) {
        REQUIRE(ctxs[k] && slabs[k], "impop_pairwise_scan_sharded: context or slab %d is NULL", k);
        for (int j = 0; j < k; ++j)
            REQUIRE(ctxs[j] != ctxs[k], "impop_pairwise_scan_sharded: contexts %d and %d are the same (a context serves one shard at a time)", j, k);
        uint64_t s0 = 0, s1 = 0;
        const int rc = impop_shard_windows(windows, n_windows, n_ctx, k, &first[k], &count[k], &s0, &s1);
        if (rc) return rc;
        if (!count[k]) continue;
        uint64_t n_site = 0;
        impop_matrix_info(slabs[k], nullptr, &n_site, nullptr, nullptr);
        if (slabs[k]->compact || s0 < slab_site_begin[k] || s1 > slab_site_begin[k] + n_site) {
            set_error("impop_pairwise_scan_sharded: slab %d ([%llu, %llu)%s) does not cover its windows' sites [%llu, %llu)", k,
                      (unsigned long long)slab_site_begin[k], (unsigned long long)(slab_site_begin[k] + n_site),
                      slabs[k]->compact ? ", compacted" : "", (unsigned long long)s0, (unsigned long long)s1);
            return IMPOP_E_INVALID;
        }
        loc[k].assign(windows + first[k], windows + first[k] + count[k]);
        for (impop_window &w : loc[k]) { w.site_begin -= slab_site_begin[k]; w.site_end -= slab_site_begin[k]; }
    }
    std::vector<int> rcs((size_t)n_ctx, IMPOP_OK);
    std::vector<std::string> msgs((size_t)n_ctx);
    auto work = [&](int k) {
        rcs[k] = impop_pairwise_scan(ctxs[k], slabs[k], loc[k].data(), count[k], mask_p, mask_a, mask_b, params, out_host + first[k]);
        if (rcs[k]) msgs[k] = impop_last_error();  // the message lives in the worker's thread: carry it over
    };
    std::vector<std::thread> th;
    int own = -1;
    for (int k = 0; k < n_ctx; ++k) {
        if (!count[k]) continue;
        if (own < 0) own = k;  // the first shard runs on the calling thread
        else th.emplace_back(work, k);
    }
    if (own >= 0) work(own);
    for (std::thread &t : th) t.join();
    for (int k = 0; k < n_ctx; ++k)
        if (rcs[k]) {
            set_error("impop_pairwise_scan_sharded: shard %d: %s", k, msgs[k].c_str());
            return rcs[k];
        }
    return IMPOP_OK;
}

// ---- one process per GPU: RCCL ---------------------------------------------------------------------------------

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int rccl_load() {
    if (g_rccl.handle) return IMPOP_OK;
    // by soname first: a copy that is already in the process (PyTorch's) is returned as is
    const char *cands[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    void *h = nullptr;
    for (const char *c : cands)
        if ((h = dlopen(c, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) {
        set_error("RCCL not found (dlopen librccl.so.1: %s)", dlerror());
        return IMPOP_E_UNSUPPORTED;
    }
#define SYM(field, name)                                                     \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name)); \
    if (!g_rccl.field) {                                                     \
        set_error("RCCL: symbol %s missing", name);                          \
        return IMPOP_E_UNSUPPORTED;                                          \
    }
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllGather, "ncclAllGather")
    SYM(AllReduce, "ncclAllReduce")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    g_rccl.handle = h;
    return IMPOP_OK;
}

int rccl_fail(ncclResult_t r, const char *what) {
    set_error("RCCL error %d (%s) in %s", (int)r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?", what);
    return IMPOP_E_HIP;
}
#define RCCL_TRY(expr)                                        \
    do {                                                      \
        ncclResult_t _r = (expr);                             \
        if (_r != ncclSuccess) return rccl_fail(_r, #expr);   \
    } while (0)

}  // namespace

struct impop_comm {
    impop_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
    void *d_stage = nullptr;  // growable device staging for impop_gather_records
    size_t stage_bytes = 0;
};

static_assert(IMPOP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "impop_hip.h and rccl.h disagree on the id size");

IMPOP_API int impop_comm_unique_id(void *id_out) {
    REQUIRE(id_out, "impop_comm_unique_id: id_out is NULL");
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId id;
    RCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return IMPOP_OK;
}

IMPOP_API int impop_comm_create(impop_ctx *ctx, const void *unique_id, int world, int rank, impop_comm **out) {
    REQUIRE(ctx && unique_id && out, "impop_comm_create: NULL argument");
    *out = nullptr;
    REQUIRE(world >= 1 && rank >= 0 && rank < world, "impop_comm_create: rank %d of %d", rank, world);
    int rc = rccl_load();
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    ncclComm_t c = nullptr;
    RCCL_TRY(g_rccl.CommInitRank(&c, world, id, rank));
    impop_comm *cm = new impop_comm();
    cm->ctx = ctx; cm->comm = c; cm->world = world; cm->rank = rank;
    *out = cm;
    return IMPOP_OK;
}

IMPOP_API int impop_comm_destroy(impop_comm *comm) {
    if (!comm) return IMPOP_OK;
    hipSetDevice(comm->ctx->device);
    hipStreamSynchronize(comm->ctx->stream);
    if (comm->d_stage) hipFree(comm->d_stage);
    if (comm->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(comm->comm);
    delete comm;
    return IMPOP_OK;
}

IMPOP_API int impop_gather(impop_comm *comm, const void *d_local, size_t bytes_per_rank, void *d_all) {
    REQUIRE(comm && (bytes_per_rank == 0 || (d_local && d_all)), "impop_gather: NULL argument");
    if (!bytes_per_rank) return IMPOP_OK;
    HIP_TRY(hipSetDevice(comm->ctx->device));
    // on the context's stream: ordered behind the scans that wrote d_local, no host synchronisation
    RCCL_TRY(g_rccl.AllGather(d_local, d_all, bytes_per_rank, ncclUint8, comm->comm, comm->ctx->stream));
    return IMPOP_OK;
}

IMPOP_API int impop_allreduce_i64(impop_comm *comm, int64_t *d_values, size_t count) {
    REQUIRE(comm && (count == 0 || d_values), "impop_allreduce_i64: NULL argument");
    if (!count) return IMPOP_OK;
    HIP_TRY(hipSetDevice(comm->ctx->device));
    RCCL_TRY(g_rccl.AllReduce(d_values, d_values, count, ncclInt64, ncclSum, comm->comm, comm->ctx->stream));
    return IMPOP_OK;
}

// shards may differ by one window: every rank sends max-shard records (the tail is padding that is dropped here)
IMPOP_API int impop_gather_records(impop_comm *comm, const void *d_local_records, uint64_t n_total_windows,
                                   impop_window_stats *out_host) {
    REQUIRE(comm, "impop_gather_records: comm is NULL");
    if (!n_total_windows) return IMPOP_OK;
    REQUIRE(out_host, "impop_gather_records: out_host is NULL");
    const uint64_t cap = (n_total_windows + (uint64_t)comm->world - 1) / (uint64_t)comm->world;
    uint64_t lo = 0, cnt = 0;
    impop_shard_range(n_total_windows, comm->world, comm->rank, &lo, &cnt);
    REQUIRE(cnt == 0 || d_local_records, "impop_gather_records: local records are NULL");
    const size_t rec = sizeof(impop_window_stats), slot = (size_t)cap * rec;
    HIP_TRY(hipSetDevice(comm->ctx->device));
    const size_t need = slot * ((size_t)comm->world + 1);
    if (need > comm->stage_bytes) {
        HIP_TRY(hipStreamSynchronize(comm->ctx->stream));
        if (comm->d_stage) HIP_TRY(hipFree(comm->d_stage));
        comm->d_stage = nullptr; comm->stage_bytes = 0;
        HIP_TRY(hipMalloc(&comm->d_stage, need));
        comm->stage_bytes = need;
    }
    char *d_send = (char *)comm->d_stage, *d_recv = d_send + slot;
    hipStream_t st = comm->ctx->stream;
    if (cnt < cap) HIP_TRY(hipMemsetAsync(d_send + cnt * rec, 0, (cap - cnt) * rec, st));
    if (cnt) HIP_TRY(hipMemcpyAsync(d_send, d_local_records, cnt * rec, hipMemcpyDeviceToDevice, st));
    RCCL_TRY(g_rccl.AllGather(d_send, d_recv, slot, ncclUint8, comm->comm, st));
    for (int r = 0; r < comm->world; ++r) {
        uint64_t rlo = 0, rcnt = 0;
        impop_shard_range(n_total_windows, comm->world, r, &rlo, &rcnt);
        if (rcnt) HIP_TRY(hipMemcpyAsync(out_host + rlo, d_recv + (size_t)r * slot, rcnt * rec, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    return IMPOP_OK;
}
