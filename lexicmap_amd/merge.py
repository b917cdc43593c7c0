"""Merging per-rank HSP rows: one all-gather of variable-length row records over torch.distributed
(RCCL on GPUs, gloo in the CPU tests), then the reference's final ordering per query — genomes by their best
SimilarityScore = bitscore*pident descending (lib-index-search.go:2919-2921; ties by genome key) — and the `hits`
column recomputed as the global number of genomes (search.go:463,494).  This is what `lexicmap utils
merge-search-results` does offline for several indexes (merge-search-results.go:142-194)."""
import numpy as np

def _row_dtype():
    """numpy view of the C struct lm_hsp (same offsets, pointers as u8) so rows move without per-field packing"""
    import ctypes as C
    from .api import Hsp
    names, formats, offsets = [], [], []
    fmt = {C.c_uint32: "<u4", C.c_int32: "<i4", C.c_uint64: "<u8", C.c_double: "<f8", C.c_char_p: "<u8"}
    for name, ct in Hsp._fields_:
        names.append(name)
        formats.append(fmt[ct])
        offsets.append(getattr(Hsp, name).offset)
    return np.dtype(dict(names=names, formats=formats, offsets=offsets, itemsize=C.sizeof(Hsp)))


ROW_DTYPE = _row_dtype()
PTR_FIELDS = ("genome_id", "seq_id", "cigar", "qseq", "sseq", "align")  # process-local addresses


def pack_rows(rows):
    arr = np.zeros(len(rows), dtype=ROW_DTYPE)
    for i, r in enumerate(rows):
        for name in ROW_DTYPE.names:
            v = r.get(name, 0)
            arr[name][i] = 0 if isinstance(v, (bytes, type(None))) else v
    return arr


def _cat(parts):
    """concatenate row arrays keeping the exact lm_hsp layout (np.concatenate may repack padded struct dtypes)"""
    parts = [p for p in parts if len(p)]
    out = np.zeros(sum(len(p) for p in parts), dtype=ROW_DTYPE)
    o = 0
    for p in parts:
        out[o:o + len(p)] = p.astype(ROW_DTYPE, copy=False)
        o += len(p)
    return out


_PINNED = None  # grow-only pinned staging buffer for the gathered rows (GPU path)


def all_gather_rows(arr, device="cpu", host_on=None):
    """gatherv of row records over torch.distributed (RCCL on GPUs, gloo in the CPU tests): the row counts first, then the
    payloads.  `host_on` = r: a gather to rank r only (`dist.gather` of the payloads padded to the largest: (N-1) payloads
    over r's xGMI links, nothing to the ranks that do not merge); the other ranks return their own rows only.
    `host_on` = None: all-gather of padded payloads, every rank gets every rank's rows.
    Returns the list of per-rank arrays.  On the GPU path the returned arrays are views of a reused pinned buffer: valid
    until the next call."""
    global _PINNED
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    rank = dist.get_rank()
    # (no private copy: the pointer columns hold process-local addresses that mean nothing on another rank, and every
    # consumer of the gathered rows - lm_merge_sharded, merge_sharded, merge_query_sharded - clears them itself)
    arr = np.ascontiguousarray(arr, dtype=ROW_DTYPE)
    item = ROW_DTYPE.itemsize
    payload = torch.from_numpy(arr.view(np.uint8).reshape(-1)) if len(arr) else torch.zeros(0, dtype=torch.uint8)
    size = torch.tensor([len(arr)], dtype=torch.int64, device=device)
    sizes = torch.zeros(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(sizes, size)
    sizes = [int(x) for x in sizes.cpu().tolist()]

    def to_host(t):
        global _PINNED
        if t.is_cuda:
            if _PINNED is None or _PINNED.numel() < t.numel():
                _PINNED = torch.empty(int(t.numel() * 1.25) + 1024, dtype=torch.uint8, pin_memory=True)
            host_t = _PINNED[:t.numel()]
            host_t.copy_(t, non_blocking=True)
            torch.cuda.synchronize()
            return host_t.numpy()
        return t.numpy()

    mx = max(1, max(sizes)) * item
    pad = torch.zeros(mx, dtype=torch.uint8, device=device)
    pad[:payload.numel()] = payload.to(device, non_blocking=True)
    if host_on is not None:
        # gather to the merging rank only: (N-1) payloads over ITS links, nothing to the ranks that do not merge (the shards
        # are balanced - genome g on rank g % N - so padding to the largest payload costs a few per cent)
        if rank != host_on:
            dist.gather(pad, gather_list=None, dst=host_on)
            return [arr]
        allb = torch.empty(world * mx, dtype=torch.uint8, device=device)
        dist.gather(pad, gather_list=[allb[r * mx:(r + 1) * mx] for r in range(world)], dst=host_on)
        host = to_host(allb)
        return [host[r * mx:r * mx + sizes[r] * item].view(ROW_DTYPE) for r in range(world)]
    allb = torch.empty(world * mx, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(allb, pad)
    host = to_host(allb)
    return [host[r * mx:r * mx + sizes[r] * item].view(ROW_DTYPE) for r in range(world)]


def merge_sharded(per_rank):
    """rows of disjoint genome shards -> one row array in the reference's output order, `hits` recomputed:
    per query, genomes by their best bitscore*pident descending (ties by genome key), each genome's rows in their
    (already final) order"""
    allr = _cat(per_rank)
    n = len(allr)
    if n == 0:
        return allr
    q = allr["query"].astype(np.int64)
    g = allr["batch_genome"].astype(np.int64)
    sim = allr["bitscore"].astype(np.float64) * allr["pident"]
    idx = np.arange(n)
    # groups of (query, genome): best similarity per group
    o = np.lexsort((idx, g, q))
    qs, gs = q[o], g[o]
    new = np.ones(n, dtype=bool)
    new[1:] = (qs[1:] != qs[:-1]) | (gs[1:] != gs[:-1])
    starts = np.flatnonzero(new)
    gid = np.cumsum(new) - 1                      # group id of every sorted row
    best = np.maximum.reduceat(sim[o], starts)    # per group
    # order: query asc, best desc, genome asc, original row order
    o2 = np.lexsort((idx[o], gs, -best[gid], qs))
    out = _cat([allr[o][o2]])  # fresh zeroed records: fancy indexing leaves the struct padding undefined
    for f in PTR_FIELDS:
        out[f] = 0
    # hits = genomes per query
    gq = qs[starts]
    uq, cnt = np.unique(gq, return_counts=True)
    hits = dict(zip(uq.tolist(), cnt.tolist()))
    out["hits"] = np.array([hits[int(x)] for x in out["query"]], dtype=out["hits"].dtype) if n < 4096 else \
        cnt[np.searchsorted(uq, out["query"].astype(np.int64))].astype(out["hits"].dtype)
    return out


def merge_query_sharded(per_rank):
    """query-sharded ranks: rows are already final per query and grouped per query; the order across queries is
    arrival order in the reference (search.go:47), here rank order"""
    out = _cat(per_rank)
    for f in PTR_FIELDS:
        out[f] = 0
    return out


def merge_sharded_c(per_rank, index=None):
    """the same merge through the library's C entry point lm_merge_sharded (what a Go host calls); `index` (an
    lexicmap_amd.Index or None) re-attaches genome / sequence names. Returns a fresh row array."""
    import ctypes as C
    from .api import Hsp, lib
    L = lib()
    arrs = [np.ascontiguousarray(p, dtype=ROW_DTYPE) for p in per_rank]
    n = len(arrs)
    ptrs = (C.POINTER(Hsp) * n)(*[a.ctypes.data_as(C.POINTER(Hsp)) for a in arrs])
    cnts = (C.c_size_t * n)(*[len(a) for a in arrs])
    res = C.c_void_p()
    st = L.lm_merge_sharded(index.h if index is not None else None, ptrs, cnts, n, C.byref(res))
    if st != 0:
        raise RuntimeError("lm_merge_sharded failed (%d)" % st)
    rows_p = C.POINTER(Hsp)()
    k = L.lm_result_rows(res, C.byref(rows_p))
    if index is None:
        # a view of the library's result (its pointer columns are NULL without an index): no copy of the merged rows - at
        # 8 shards x 6e5 rows the copy out and the six strided column writes were a third of the host-side merge
        if not k:
            L.lm_result_free(res)
            return np.zeros(0, dtype=ROW_DTYPE)
        import weakref
        buf = (C.c_char * (k * C.sizeof(Hsp))).from_address(C.addressof(rows_p.contents))
        out = np.frombuffer(buf, dtype=ROW_DTYPE)
        weakref.finalize(buf, L.lm_result_free, res)  # out -> buf keeps the rows alive
        return out
    out = np.zeros(k, dtype=ROW_DTYPE)
    if k:
        C.memmove(out.ctypes.data, rows_p, k * C.sizeof(Hsp))
    names = [(rows_p[i].genome_id, rows_p[i].seq_id) for i in range(k)] if k else None
    L.lm_result_free(res)
    for f in PTR_FIELDS:
        out[f] = 0
    return out, names


def topn_merge(cands, top_n):
    """cands: per shard (query u32, batch_genome u64, score f32) arrays -> (keep_query, keep_bg) of the global top-N"""
    import ctypes as C
    from .api import lib
    L = lib()
    n = len(cands)
    qs = [np.ascontiguousarray(c[0], dtype=np.uint32) for c in cands]
    gs = [np.ascontiguousarray(c[1], dtype=np.uint64) for c in cands]
    ss = [np.ascontiguousarray(c[2], dtype=np.float32) for c in cands]
    pq = (C.POINTER(C.c_uint32) * n)(*[a.ctypes.data_as(C.POINTER(C.c_uint32)) for a in qs])
    pg = (C.POINTER(C.c_uint64) * n)(*[a.ctypes.data_as(C.POINTER(C.c_uint64)) for a in gs])
    ps = (C.POINTER(C.c_float) * n)(*[a.ctypes.data_as(C.POINTER(C.c_float)) for a in ss])
    cn = (C.c_size_t * n)(*[len(a) for a in qs])
    oq, og, on = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint64)(), C.c_size_t()
    st = L.lm_topn_merge(n, pq, pg, ps, cn, top_n, C.byref(oq), C.byref(og), C.byref(on))
    if st != 0:
        raise RuntimeError("lm_topn_merge failed (%d)" % st)
    k = on.value
    kq = np.ctypeslib.as_array(oq, shape=(k,)).copy() if k else np.zeros(0, np.uint32)
    kg = np.ctypeslib.as_array(og, shape=(k,)).copy() if k else np.zeros(0, np.uint64)
    L.lm_free(oq)
    L.lm_free(og)
    return kq, kg
