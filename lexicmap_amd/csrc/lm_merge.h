// lm_merge.h - the shard merge on the device (lm_merge.hip), used by lm_gather_merge_rows (lm_comm.cpp)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/lexicmap_hip.h"

namespace lm {
struct MergeScratch { // grow-only device buffers, owned by the communicator
    void *head = nullptr, *gid = nullptr, *sim = nullptr, *keys = nullptr, *keys2 = nullptr, *first = nullptr, *cnt = nullptr, *sizes = nullptr,
         *hits = nullptr, *outpos = nullptr, *tmp = nullptr, *off = nullptr;
    size_t cap[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // Set for the duration of ONE call when the merging rank's index handle is known: the buffers are then borrowed from the
    // handle's scratch slabs - idle between two searches - and given back at the end, instead of hipMalloc'ed beside them (a
    // production-size handle holds 72 % of the device in its lane slabs: at shard-of-4 the 2.5 GB of the merge did not fit).
    void *(*borrow)(void *ctx, size_t bytes) = nullptr;
    void (*give_back)(void *ctx, void *p) = nullptr;
    void *ctx = nullptr;
    void *borrowed[16];
    int nborrowed = 0;
    void *take(size_t bytes); // one buffer for this call (borrow mode only)
    void end_call();          // give everything borrowed back, forget the pointers
    void release();
};
hipError_t merge_rows_device(hipStream_t st, const lm_hsp *d_rows, size_t n, const int64_t *off_host, int nranks, lm_hsp *d_out, MergeScratch &S);
} // namespace lm

struct lm_index;
// genome_id / seq_id of rows that came from other processes (lm_pipeline.hip): every shard holds the names of all genomes; the
// names of a synthetic set are made once per genome and kept with the handle.  Threaded over the rows.
extern "C" void lm_attach_names(lm_index *ix, lm_hsp *rows, size_t n); // (internal: not in the public header)
// The handle's scratch slabs lent to the caller between two searches (lm_pipeline.hip; internal).  begin takes the handle's
// mutex - no search can start meanwhile - end releases it; borrow returns NULL when the scratch cannot hold the block.
extern "C" void lm_scratch_session_begin(lm_index *ix);
extern "C" void lm_scratch_session_end(lm_index *ix);
extern "C" void *lm_scratch_borrow(lm_index *ix, size_t bytes);
extern "C" void lm_scratch_return(lm_index *ix, void *p);
