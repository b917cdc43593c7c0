// lm_comm.cpp - the ONE collective of the sharded search behind the C-ABI (include/lexicmap_hip.h):
// a gatherv of lm_hsp row records over RCCL (xGMI inside a node).  Every rank searched the same query batch against its
// genome shard; the merging rank needs every shard's rows, the others need nothing back - so the collective is: one
// all-gather of the row counts (8 bytes per rank), then ONE group of point-to-point transfers (ncclSend on the ranks,
// ncclRecv x (N-1) on the root: (N-1) payloads over the root's xGMI links, nothing to the ranks that do not merge).  What
// it merges into: lm_merge_sharded = the order of lib-index-search.go:2919-2921 / merge-search-results.go:142-194.
//
// RCCL is bound at run time (dlopen of librccl.so.1): a host process that already carries an RCCL (a PyTorch process
// carries its own copy) keeps using that ONE instance, and a single-GPU user of the library needs no RCCL at all.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/lexicmap_hip.h"
#include "lm_merge.h"

// The few RCCL types and constants this file needs, declared here (values of rccl.h / nccl.h, stable since NCCL 2.0): the
// library is bound with dlopen, so the single-GPU build must not need the RCCL headers either.  lm_comm_* check ncclGetVersion
// after binding (point-to-point transfers exist since 2.7).
typedef struct ncclComm *ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclUint8 = 1, ncclUint64 = 5 } ncclDataType_t;

extern thread_local std::string g_open_error; // text of the last failure without a handle (lm_last_error(NULL))

namespace {
struct Rccl {
    void *so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    int version = 0;
    std::string err;
    bool ok = false;
};
Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {getenv("LM_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            r.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.so) break;
        }
        if (!r.so) {
            r.err = std::string("RCCL not found (librccl.so.1): ") + (dlerror() ? dlerror() : "");
            return;
        }
#define LM_SYM(field, name)                                                      \
    r.field = (decltype(r.field))dlsym(r.so, name);                              \
    if (!r.field) {                                                              \
        r.err = std::string("RCCL symbol missing: ") + name;                     \
        return;                                                                  \
    }
        LM_SYM(GetUniqueId, "ncclGetUniqueId")
        LM_SYM(CommInitRank, "ncclCommInitRank")
        LM_SYM(CommDestroy, "ncclCommDestroy")
        LM_SYM(AllGather, "ncclAllGather")
        LM_SYM(Send, "ncclSend")
        LM_SYM(Recv, "ncclRecv")
        LM_SYM(GroupStart, "ncclGroupStart")
        LM_SYM(GroupEnd, "ncclGroupEnd")
        LM_SYM(GetErrorString, "ncclGetErrorString")
        LM_SYM(GetVersion, "ncclGetVersion")
#undef LM_SYM
        // 2.7.0 is 2700 in the old numbering (major * 1000 + minor * 100 + patch), 2.9+ count major * 10000: both are >= 2700
        if (r.GetVersion(&r.version) != ncclSuccess || r.version < 2700) {
            r.err = "the RCCL that was found is older than 2.7 (no ncclSend / ncclRecv): version code " + std::to_string(r.version);
            return;
        }
        r.ok = true;
    });
    return r;
}
} // namespace

struct lm_comm {
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0, device = 0;
    hipStream_t st = nullptr;
    // grow-only staging: device send / receive buffers, pinned host mirror of the received rows, the counts
    void *d_send = nullptr, *d_recv = nullptr, *h_recv = nullptr, *h_send = nullptr;
    size_t send_cap = 0, recv_cap = 0, hrecv_cap = 0, hsend_cap = 0;
    unsigned long long *d_counts = nullptr; // [nranks + 1]: the gathered counts, then this rank's own
    std::vector<size_t> counts;
    // lm_gather_merge_rows: all ranks' rows in rank order, the merged rows, their pinned host mirror, the merge's scratch
    void *d_all = nullptr, *d_merged = nullptr, *h_merged = nullptr;
    size_t all_cap = 0, merged_cap = 0, hmerged_cap = 0;
    lm::MergeScratch ms;
    std::string err;
    std::mutex mu;
};

static_assert(sizeof(ncclUniqueId) == LM_COMM_ID_BYTES, "lm_comm_unique_id hands out an ncclUniqueId");

#define CK_HIP(c, expr)                                                                                \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            (c)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                              \
            return LM_ERR_HIP;                                                                         \
        }                                                                                              \
    } while (0)
#define CK_NCCL(c, expr)                                                                               \
    do {                                                                                               \
        ncclResult_t e_ = (expr);                                                                      \
        if (e_ != ncclSuccess) {                                                                       \
            (c)->err = std::string(#expr) + ": " + rccl().GetErrorString(e_);                          \
            return LM_ERR_HIP;                                                                         \
        }                                                                                              \
    } while (0)

extern "C" {

lm_status lm_comm_unique_id(uint8_t id[LM_COMM_ID_BYTES]) {
    if (!id) return LM_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        (void)hipGetLastError();
        g_open_error = "no HIP device: the row gather runs over RCCL between GPUs";
        return LM_ERR_NO_DEVICE;
    }
    Rccl &r = rccl();
    if (!r.ok) {
        g_open_error = r.err;
        return LM_ERR_HIP;
    }
    ncclUniqueId u;
    ncclResult_t e = r.GetUniqueId(&u);
    if (e != ncclSuccess) {
        g_open_error = std::string("ncclGetUniqueId: ") + r.GetErrorString(e);
        return LM_ERR_HIP;
    }
    memcpy(id, &u, LM_COMM_ID_BYTES);
    return LM_OK;
}

lm_status lm_comm_init(const uint8_t id[LM_COMM_ID_BYTES], int nranks, int rank, int device, lm_comm **out) {
    if (!out) return LM_ERR_ARG;
    *out = nullptr;
    if (!id || nranks < 1 || rank < 0 || rank >= nranks) return LM_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        (void)hipGetLastError();
        g_open_error = "no HIP device: the row gather runs over RCCL between GPUs";
        return LM_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) {
        g_open_error = "lm_comm_init: no such device";
        return LM_ERR_ARG;
    }
    Rccl &r = rccl();
    if (!r.ok) {
        g_open_error = r.err;
        return LM_ERR_HIP;
    }
    lm_comm *c = new lm_comm();
    c->nranks = nranks;
    c->rank = rank;
    c->device = device;
    c->counts.assign((size_t)nranks, 0);
    auto fail = [&](const std::string &m) {
        g_open_error = m;
        lm_comm_free(c);
        return LM_ERR_HIP;
    };
    if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice failed");
    ncclUniqueId u;
    memcpy(&u, id, LM_COMM_ID_BYTES);
    ncclResult_t e = r.CommInitRank(&c->comm, nranks, u, rank);
    if (e != ncclSuccess) {
        c->comm = nullptr;
        return fail(std::string("ncclCommInitRank: ") + r.GetErrorString(e));
    }
    if (hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking) != hipSuccess) return fail("hipStreamCreate failed");
    if (hipMalloc((void **)&c->d_counts, sizeof(unsigned long long) * (size_t)(nranks + 1)) != hipSuccess) return fail("hipMalloc failed");
    *out = c;
    return LM_OK;
}

void lm_comm_free(lm_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->st) (void)hipStreamSynchronize(c->st);
    if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    if (c->d_counts) (void)hipFree(c->d_counts);
    if (c->h_recv) (void)hipHostFree(c->h_recv);
    if (c->h_send) (void)hipHostFree(c->h_send);
    if (c->d_all) (void)hipFree(c->d_all);
    if (c->d_merged) (void)hipFree(c->d_merged);
    if (c->h_merged) (void)hipHostFree(c->h_merged);
    c->ms.release();
    if (c->st) (void)hipStreamDestroy(c->st);
    delete c;
}

const char *lm_comm_last_error(const lm_comm *c) { return c ? c->err.c_str() : g_open_error.c_str(); }
int lm_comm_rank(const lm_comm *c) { return c ? c->rank : -1; }
int lm_comm_size(const lm_comm *c) { return c ? c->nranks : 0; }

static lm_status grow_dev(lm_comm *c, void **p, size_t *cap, size_t need) {
    if (need <= *cap && *p) return LM_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    const size_t want = need + need / 4 + 4096;
    CK_HIP(c, hipMalloc(p, want));
    *cap = want;
    return LM_OK;
}
static lm_status grow_host(lm_comm *c, void **p, size_t *cap, size_t need) {
    if (need <= *cap && *p) return LM_OK;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr;
    *cap = 0;
    const size_t want = need + need / 4 + 4096;
    CK_HIP(c, hipHostMalloc(p, want, hipHostMallocDefault));
    *cap = want;
    return LM_OK;
}

// rows / n: this rank's rows (host memory, as lm_result_rows returns them).  On `root`: *all_rows = the rows of rank 0,
// then rank 1, ... (nrows[r] of each; pointer columns cleared - they are addresses of another process), valid until the next
// call on this communicator; on the other ranks *all_rows = NULL and nrows[] still holds every rank's count.
lm_status lm_gather_rows(lm_comm *c, const lm_hsp *rows, size_t n, int root, const lm_hsp **all_rows, size_t *nrows) {
    if (!c || !all_rows || !nrows || (n > 0 && !rows) || root < 0 || root >= c->nranks) return LM_ERR_ARG;
    *all_rows = nullptr;
    std::lock_guard<std::mutex> lock(c->mu);
    Rccl &r = rccl();
    CK_HIP(c, hipSetDevice(c->device));
    const int N = c->nranks;
    // 1. the counts: all-gather of one 64-bit word per rank
    unsigned long long mine = (unsigned long long)n;
    CK_HIP(c, hipMemcpyAsync(c->d_counts + N, &mine, sizeof mine, hipMemcpyHostToDevice, c->st));
    CK_NCCL(c, r.AllGather(c->d_counts + N, c->d_counts, 1, ncclUint64, c->comm, c->st));
    std::vector<unsigned long long> cnt((size_t)N);
    CK_HIP(c, hipMemcpyAsync(cnt.data(), c->d_counts, sizeof(unsigned long long) * (size_t)N, hipMemcpyDeviceToHost, c->st));
    CK_HIP(c, hipStreamSynchronize(c->st));
    size_t total = 0;
    std::vector<size_t> off((size_t)N + 1, 0);
    for (int i = 0; i < N; i++) {
        nrows[i] = (size_t)cnt[(size_t)i];
        off[(size_t)i + 1] = off[(size_t)i] + nrows[i];
    }
    total = off[(size_t)N];
    const size_t item = sizeof(lm_hsp);
    // 2. the payloads: the root keeps its own rows on the host; every other rank stages its rows on the device and sends
    if (c->rank != root) {
        if (n > 0) {
            lm_status s = grow_dev(c, &c->d_send, &c->send_cap, n * item);
            if (s != LM_OK) return s;
            s = grow_host(c, &c->h_send, &c->hsend_cap, n * item); // pinned: the upload is one DMA
            if (s != LM_OK) return s;
            memcpy(c->h_send, rows, n * item);
            CK_HIP(c, hipMemcpyAsync(c->d_send, c->h_send, n * item, hipMemcpyHostToDevice, c->st));
            CK_NCCL(c, r.Send(c->d_send, n * item, ncclUint8, root, c->comm, c->st));
        }
        CK_HIP(c, hipStreamSynchronize(c->st));
        return LM_OK;
    }
    lm_status s = grow_host(c, &c->h_recv, &c->hrecv_cap, std::max<size_t>(total, 1) * item);
    if (s != LM_OK) return s;
    const size_t remote = total - nrows[root];
    if (remote > 0) {
        s = grow_dev(c, &c->d_recv, &c->recv_cap, remote * item);
        if (s != LM_OK) return s;
        // one group: the (N-1) receives progress together over the root's links
        CK_NCCL(c, r.GroupStart());
        size_t doff = 0;
        for (int i = 0; i < N; i++) {
            if (i == root || nrows[i] == 0) continue;
            ncclResult_t e = r.Recv((char *)c->d_recv + doff * item, nrows[i] * item, ncclUint8, i, c->comm, c->st);
            if (e != ncclSuccess) {
                (void)r.GroupEnd();
                c->err = std::string("ncclRecv: ") + r.GetErrorString(e);
                return LM_ERR_HIP;
            }
            doff += nrows[i];
        }
        CK_NCCL(c, r.GroupEnd());
        // device -> the pinned host mirror, every rank's block at its place in rank order
        doff = 0;
        for (int i = 0; i < N; i++) {
            if (i == root || nrows[i] == 0) continue;
            CK_HIP(c, hipMemcpyAsync((char *)c->h_recv + off[(size_t)i] * item, (char *)c->d_recv + doff * item, nrows[i] * item,
                                     hipMemcpyDeviceToHost, c->st));
            doff += nrows[i];
        }
    }
    if (n > 0) memcpy((char *)c->h_recv + off[(size_t)root] * item, rows, n * item); // (beside the transfers)
    CK_HIP(c, hipStreamSynchronize(c->st));
    lm_hsp *all = (lm_hsp *)c->h_recv;
    for (size_t i = 0; i < total; i++) { // addresses of another process (and of results the caller may free)
        all[i].genome_id = nullptr;
        all[i].seq_id = nullptr;
        all[i].cigar = nullptr;
        all[i].qseq = nullptr;
        all[i].sseq = nullptr;
        all[i].align = nullptr;
    }
    *all_rows = all;
    return LM_OK;
}

// the handle's idle scratch slabs as the allocator of one merge (lm_merge.h)
static void *borrow_cb(void *ctx, size_t bytes) { return lm_scratch_borrow((lm_index *)ctx, bytes); }
static void return_cb(void *ctx, void *p) { lm_scratch_return((lm_index *)ctx, p); }

// rows of all shards in device memory, rank order -> the final order in the communicator's pinned buffer (lm_merge.hip + one
// download + the names); the caller holds c->mu and, when idx is given, the handle's scratch session (the buffers of the merge
// are then borrowed from the handle's scratch slabs)
static lm_status merge_on_device(lm_comm *c, lm_index *idx, const lm_hsp *d_rows, const int64_t *off, int N, const lm_hsp **merged, size_t *total_out) {
    const size_t total = (size_t)off[N], item = sizeof(lm_hsp);
    lm_status s = grow_host(c, &c->h_merged, &c->hmerged_cap, total * item);
    if (s != LM_OK) return s;
    lm_hsp *d_out = nullptr;
    if (idx) {
        c->ms.borrow = borrow_cb;
        c->ms.give_back = return_cb;
        c->ms.ctx = idx;
        d_out = (lm_hsp *)c->ms.take(total * item);
        if (!d_out) {
            c->ms.end_call();
            c->err = "the merge's buffers do not fit the index handle's scratch";
            return LM_ERR_NOMEM;
        }
    } else {
        s = grow_dev(c, &c->d_merged, &c->merged_cap, total * item);
        if (s != LM_OK) return s;
        d_out = (lm_hsp *)c->d_merged;
    }
    const bool dbg = getenv("LM_DEBUG") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    hipError_t e = lm::merge_rows_device(c->st, d_rows, total, off, N, d_out, c->ms);
    if (e == hipSuccess && dbg) e = hipStreamSynchronize(c->st);
    const double t1 = now();
    if (e == hipSuccess) e = hipMemcpyAsync(c->h_merged, d_out, total * item, hipMemcpyDeviceToHost, c->st);
    if (e == hipSuccess) e = hipStreamSynchronize(c->st);
    else (void)hipStreamSynchronize(c->st);
    c->ms.end_call(); // (borrowed buffers go back to the handle whatever happened)
    if (e != hipSuccess) {
        c->err = std::string("device merge: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? LM_ERR_NOMEM : LM_ERR_HIP;
    }
    const double t2 = now();
    lm_attach_names(idx, (lm_hsp *)c->h_merged, total);
    if (dbg)
        fprintf(stderr, "[lm] device merge of %zu rows: order %.1f ms, download %.1f ms (%.1f GB/s), names %.1f ms\n", total, t1 - t0, t2 - t1,
                (double)(total * item) / 1e6 / std::max(t2 - t1, 1e-3), now() - t2);
    *merged = (const lm_hsp *)c->h_merged;
    *total_out = total;
    return LM_OK;
}

// The gather and the merge in one call: the rows of the other ranks are received into device memory at their place in rank order,
// this rank's own rows are uploaded beside them, the final order is made on the device (lm_merge.hip) and downloaded ONCE into
// the communicator's pinned buffer; the names are re-attached by the host threads.  Same rows, same order, same `hits` as
// lm_gather_rows + lm_merge_sharded.  On `root`: *merged / *total; elsewhere *merged = NULL, *total = 0.  idx (may be NULL:
// names stay NULL) is the root's index handle.  All ranks call it, in the same order as their other collective calls.
lm_status lm_gather_merge_rows(lm_comm *c, lm_index *idx, const lm_hsp *rows, size_t n, int root, const lm_hsp **merged, size_t *total_out) {
    if (!c || !merged || !total_out || (n > 0 && !rows) || root < 0 || root >= c->nranks) return LM_ERR_ARG;
    *merged = nullptr;
    *total_out = 0;
    std::lock_guard<std::mutex> lock(c->mu);
    Rccl &r = rccl();
    CK_HIP(c, hipSetDevice(c->device));
    const int N = c->nranks;
    unsigned long long mine = (unsigned long long)n;
    CK_HIP(c, hipMemcpyAsync(c->d_counts + N, &mine, sizeof mine, hipMemcpyHostToDevice, c->st));
    CK_NCCL(c, r.AllGather(c->d_counts + N, c->d_counts, 1, ncclUint64, c->comm, c->st));
    std::vector<unsigned long long> cnt((size_t)N);
    CK_HIP(c, hipMemcpyAsync(cnt.data(), c->d_counts, sizeof(unsigned long long) * (size_t)N, hipMemcpyDeviceToHost, c->st));
    CK_HIP(c, hipStreamSynchronize(c->st));
    std::vector<int64_t> off((size_t)N + 1, 0);
    for (int i = 0; i < N; i++) off[(size_t)i + 1] = off[(size_t)i] + (int64_t)cnt[(size_t)i];
    const size_t total = (size_t)off[(size_t)N], item = sizeof(lm_hsp);
    lm_status s = LM_OK;
    if (n > 0) { // this rank's rows to the device through the pinned mirror (one DMA): the payload of a send, or the root's own block
        s = grow_host(c, &c->h_send, &c->hsend_cap, n * item);
        if (s != LM_OK) return s;
        memcpy(c->h_send, rows, n * item);
    }
    if (c->rank != root) {
        if (n > 0) {
            s = grow_dev(c, &c->d_send, &c->send_cap, n * item);
            if (s != LM_OK) return s;
            CK_HIP(c, hipMemcpyAsync(c->d_send, c->h_send, n * item, hipMemcpyHostToDevice, c->st));
            CK_NCCL(c, r.Send(c->d_send, n * item, ncclUint8, root, c->comm, c->st));
        }
        CK_HIP(c, hipStreamSynchronize(c->st));
        return LM_OK;
    }
    if (total == 0) return LM_OK;
    // the gathered rows: in the handle's scratch when there is one (idle between two searches), else in a buffer of the communicator
    struct Session {
        lm_index *ix;
        void *blk = nullptr;
        explicit Session(lm_index *i) : ix(i) {
            if (ix) lm_scratch_session_begin(ix);
        }
        ~Session() {
            if (ix && blk) lm_scratch_return(ix, blk);
            if (ix) lm_scratch_session_end(ix);
        }
    } session(idx);
    if (idx) {
        session.blk = lm_scratch_borrow(idx, total * item);
        if (!session.blk) {
            c->err = "the gathered rows do not fit the index handle's scratch";
            // (the other ranks are sending: receive into nothing is not possible - the job fails, as any error inside a collective)
            return LM_ERR_NOMEM;
        }
    } else {
        s = grow_dev(c, &c->d_all, &c->all_cap, total * item);
        if (s != LM_OK) return s;
    }
    char *const d_all = idx ? (char *)session.blk : (char *)c->d_all;
    if (n > 0) CK_HIP(c, hipMemcpyAsync(d_all + (size_t)off[(size_t)root] * item, c->h_send, n * item, hipMemcpyHostToDevice, c->st));
    if (total > n) {
        CK_NCCL(c, r.GroupStart());
        for (int i = 0; i < N; i++) {
            if (i == root || cnt[(size_t)i] == 0) continue;
            ncclResult_t e = r.Recv(d_all + (size_t)off[(size_t)i] * item, (size_t)cnt[(size_t)i] * item, ncclUint8, i, c->comm, c->st);
            if (e != ncclSuccess) {
                (void)r.GroupEnd();
                c->err = std::string("ncclRecv: ") + r.GetErrorString(e);
                return LM_ERR_HIP;
            }
        }
        CK_NCCL(c, r.GroupEnd());
    }
    return merge_on_device(c, idx, (const lm_hsp *)d_all, off.data(), N, merged, total_out);
}

// What the merging rank does once the rows have arrived, by itself: `d_rows` = the rows of shard 0, 1, ... back to back IN DEVICE
// MEMORY (nrows[r] of each; each block grouped by query ascending, a genome's rows together), merged on the device on the
// communicator's stream (a single-rank communicator will do), downloaded once, names re-attached from idx.  *merged as in
// lm_gather_merge_rows.  (bench.py times the merge of N shards' worth of rows on one GPU with it.)
lm_status lm_merge_sharded_device(lm_comm *c, lm_index *idx, const void *d_rows, const size_t *nrows, int nshards, const lm_hsp **merged,
                                  size_t *total_out) {
    if (!c || !merged || !total_out || !nrows || nshards < 1) return LM_ERR_ARG;
    *merged = nullptr;
    *total_out = 0;
    std::lock_guard<std::mutex> lock(c->mu);
    CK_HIP(c, hipSetDevice(c->device));
    std::vector<int64_t> off((size_t)nshards + 1, 0);
    for (int i = 0; i < nshards; i++) off[(size_t)i + 1] = off[(size_t)i] + (int64_t)nrows[i];
    if (off[(size_t)nshards] == 0) return LM_OK;
    if (!d_rows) return LM_ERR_ARG;
    if (idx) lm_scratch_session_begin(idx);
    const lm_status st = merge_on_device(c, idx, (const lm_hsp *)d_rows, off.data(), nshards, merged, total_out);
    if (idx) lm_scratch_session_end(idx);
    return st;
}

} // extern "C"
