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
    void release();
};
hipError_t merge_rows_device(hipStream_t st, const lm_hsp *d_rows, size_t n, const int64_t *off_host, int nranks, lm_hsp *d_out, MergeScratch &S);
} // namespace lm

struct lm_index;
// genome_id / seq_id of rows that came from other processes (lm_pipeline.hip): every shard holds the names of all genomes; the
// names of a synthetic set are made once per genome and kept with the handle.  Threaded over the rows.
extern "C" void lm_attach_names(lm_index *ix, lm_hsp *rows, size_t n); // (internal: not in the public header)
