/* fake_rccl.c - TEST INFRASTRUCTURE: the nine nccl* entry points lexicmap_amd/csrc/lm_comm.cpp binds at run time (+ ncclGetVersion),
 * implemented over Unix-domain sockets + hipMemcpy, so that the multi-rank branch of lm_gather_rows / lm_gather_merge_rows runs
 * with two (or more) PROCESSES ON ONE GPU - RCCL itself refuses two ranks on one device, and a gpurun box has one GPU.  Loaded
 * through LM_RCCL_LIB (tests/test_gpu_gather_two_procs.py); never part of the product.
 *
 * Semantics kept from NCCL as far as lm_comm.cpp relies on them: buffers are DEVICE pointers; an operation is ordered after
 * the work already on `stream` (the fake waits for the stream, then moves the bytes synchronously: when the call returns the
 * data is where it belongs, which is stronger than NCCL's "when the stream gets there"); send / recv pairs match by peer in
 * program order; the group calls are no-ops (receives inside a group run one after the other - the senders block until read).
 * Topology: every rank listens on <id>.<rank>; rank j connects to every i < j.  The unique id is the socket path prefix. */
#define __HIP_PLATFORM_AMD__ 1
#include <errno.h>
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <time.h>
#include <unistd.h>

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5 } ncclDataType_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct fakeComm {
    int nranks, rank;
    int fd[64]; /* connection to every peer (-1: self) */
    int listen_fd;
    char path[160];
};
typedef struct fakeComm *ncclComm_t;

static size_t dt_size(ncclDataType_t t) { return t == ncclInt8 || t == ncclUint8 ? 1 : (t == ncclInt32 || t == ncclUint32 ? 4 : 8); }

static int write_all(int fd, const void *p, size_t n) {
    const char *c = (const char *)p;
    while (n) {
        ssize_t w = write(fd, c, n);
        if (w < 0) {
            if (errno == EINTR) continue;
            return -1;
        }
        c += w;
        n -= (size_t)w;
    }
    return 0;
}
static int read_all(int fd, void *p, size_t n) {
    char *c = (char *)p;
    while (n) {
        ssize_t r = read(fd, c, n);
        if (r < 0) {
            if (errno == EINTR) continue;
            return -1;
        }
        if (r == 0) return -1; /* the peer is gone */
        c += r;
        n -= (size_t)r;
    }
    return 0;
}

ncclResult_t ncclGetVersion(int *v) {
    if (!v) return ncclInvalidArgument;
    *v = 22606; /* what the image's RCCL reports */
    return ncclSuccess;
}
const char *ncclGetErrorString(ncclResult_t e) {
    switch (e) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "fake rccl: HIP call failed";
    case ncclSystemError: return "fake rccl: socket error";
    case ncclInvalidArgument: return "fake rccl: invalid argument";
    default: return "fake rccl: internal error";
    }
}
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    const char *dir = getenv("LM_FAKE_RCCL_DIR");
    snprintf(id->internal, sizeof id->internal, "%s/lmfake_%d_%ld", dir && *dir ? dir : "/tmp", (int)getpid(), (long)(ts.tv_nsec & 0xffffff));
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
    if (!out || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    struct fakeComm *c = (struct fakeComm *)calloc(1, sizeof *c);
    c->nranks = nranks;
    c->rank = rank;
    c->listen_fd = -1;
    for (int i = 0; i < 64; i++) c->fd[i] = -1;
    id.internal[127] = 0;
    snprintf(c->path, sizeof c->path, "%s.%d", id.internal, rank);
    if (rank < nranks - 1) { /* somebody connects to me */
        c->listen_fd = socket(AF_UNIX, SOCK_STREAM, 0);
        struct sockaddr_un a;
        memset(&a, 0, sizeof a);
        a.sun_family = AF_UNIX;
        strncpy(a.sun_path, c->path, sizeof a.sun_path - 1);
        unlink(c->path);
        if (c->listen_fd < 0 || bind(c->listen_fd, (struct sockaddr *)&a, sizeof a) != 0 || listen(c->listen_fd, 64) != 0) return ncclSystemError;
    }
    for (int i = 0; i < rank; i++) { /* connect to every lower rank (it may not be listening yet: retry for a minute) */
        char peer[200];
        snprintf(peer, sizeof peer, "%s.%d", id.internal, i);
        struct sockaddr_un a;
        memset(&a, 0, sizeof a);
        a.sun_family = AF_UNIX;
        strncpy(a.sun_path, peer, sizeof a.sun_path - 1);
        int fd = -1;
        for (int t = 0; t < 6000; t++) {
            fd = socket(AF_UNIX, SOCK_STREAM, 0);
            if (fd >= 0 && connect(fd, (struct sockaddr *)&a, sizeof a) == 0) break;
            if (fd >= 0) close(fd);
            fd = -1;
            usleep(10000);
        }
        if (fd < 0) return ncclSystemError;
        int32_t me = rank;
        if (write_all(fd, &me, sizeof me) != 0) return ncclSystemError;
        c->fd[i] = fd;
    }
    for (int n = rank + 1; n < nranks; n++) { /* accept every higher rank */
        int fd = accept(c->listen_fd, NULL, NULL);
        int32_t who = -1;
        if (fd < 0 || read_all(fd, &who, sizeof who) != 0 || who <= rank || who >= nranks || c->fd[who] != -1) return ncclSystemError;
        c->fd[who] = fd;
    }
    *out = c;
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    for (int i = 0; i < 64; i++)
        if (c->fd[i] >= 0) close(c->fd[i]);
    if (c->listen_fd >= 0) {
        close(c->listen_fd);
        unlink(c->path);
    }
    free(c);
    return ncclSuccess;
}
ncclResult_t ncclGroupStart(void) { return ncclSuccess; }
ncclResult_t ncclGroupEnd(void) { return ncclSuccess; }

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t st) {
    if (!c || peer < 0 || peer >= c->nranks || peer == c->rank) return ncclInvalidArgument;
    const size_t n = count * dt_size(dt);
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    void *h = malloc(n ? n : 1);
    if (n && hipMemcpy(h, buf, n, hipMemcpyDeviceToHost) != hipSuccess) {
        free(h);
        return ncclUnhandledCudaError;
    }
    uint64_t len = n;
    const int bad = write_all(c->fd[peer], &len, sizeof len) != 0 || write_all(c->fd[peer], h, n) != 0;
    free(h);
    return bad ? ncclSystemError : ncclSuccess;
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t st) {
    if (!c || peer < 0 || peer >= c->nranks || peer == c->rank) return ncclInvalidArgument;
    const size_t n = count * dt_size(dt);
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    uint64_t len = 0;
    if (read_all(c->fd[peer], &len, sizeof len) != 0) return ncclSystemError;
    if (len != n) return ncclInvalidArgument; /* the sizes of a send / receive pair must agree */
    void *h = malloc(n ? n : 1);
    if (read_all(c->fd[peer], h, n) != 0) {
        free(h);
        return ncclSystemError;
    }
    const int bad = n && hipMemcpy(buf, h, n, hipMemcpyHostToDevice) != hipSuccess;
    free(h);
    return bad ? ncclUnhandledCudaError : ncclSuccess;
}
ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclComm_t c, hipStream_t st) {
    if (!c) return ncclInvalidArgument;
    const size_t n = count * dt_size(dt);
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    char *all = (char *)malloc(n * (size_t)c->nranks + 1);
    if (n && hipMemcpy(all + n * (size_t)c->rank, send, n, hipMemcpyDeviceToHost) != hipSuccess) {
        free(all);
        return ncclUnhandledCudaError;
    }
    /* small blocks (the row counts): everybody writes to everybody, then reads - the socket buffers hold them */
    for (int i = 0; i < c->nranks; i++)
        if (i != c->rank && write_all(c->fd[i], all + n * (size_t)c->rank, n) != 0) {
            free(all);
            return ncclSystemError;
        }
    for (int i = 0; i < c->nranks; i++)
        if (i != c->rank && read_all(c->fd[i], all + n * (size_t)i, n) != 0) {
            free(all);
            return ncclSystemError;
        }
    const int bad = n && hipMemcpy(recv, all, n * (size_t)c->nranks, hipMemcpyHostToDevice) != hipSuccess;
    free(all);
    return bad ? ncclUnhandledCudaError : ncclSuccess;
}
