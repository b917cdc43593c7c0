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


def all_gather_rows(arr, device="cpu"):
    """all-gatherv of row records: sizes first, then padded payloads. Returns the list of per-rank arrays."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    arr = _cat([arr])
    for f in ("genome_id", "seq_id", "cigar", "qseq", "sseq", "align"):
        arr[f] = 0  # process-local addresses mean nothing on another rank
    payload = torch.from_numpy(np.frombuffer(arr.tobytes(), dtype=np.uint8).copy()).to(device)
    size = torch.tensor([payload.numel()], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(size) for _ in range(world)]
    dist.all_gather(sizes, size)
    mx = max(1, int(max(s.item() for s in sizes)))
    pad = torch.zeros(mx, dtype=torch.uint8, device=device)
    pad[:payload.numel()] = payload
    bufs = [torch.zeros(mx, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(bufs, pad)
    out = []
    for b, s in zip(bufs, sizes):
        raw = b[:int(s.item())].cpu().numpy().tobytes()
        out.append(np.frombuffer(raw, dtype=ROW_DTYPE).copy())
    return out


def merge_sharded(per_rank):
    """rows of disjoint genome shards -> one row array in the reference's output order, `hits` recomputed"""
    allr = _cat(per_rank)
    if len(allr) == 0:
        return allr
    out = []
    for q in np.unique(allr["query"]):
        rq = allr[allr["query"] == q]
        genomes = {}
        for i, r in enumerate(rq):  # keep each genome's rows in their (already final) per-genome order
            genomes.setdefault(int(r["batch_genome"]), []).append(i)
        best = {g: max(float(rq["bitscore"][i]) * float(rq["pident"][i]) for i in ix) for g, ix in genomes.items()}
        order = sorted(genomes, key=lambda g: (-best[g], g))
        for g in order:
            blk = rq[genomes[g]].copy()
            blk["hits"] = len(order)
            out.append(blk)
    return _cat(out)


def merge_query_sharded(per_rank):
    """query-sharded ranks: rows are already final per query; just concatenate in query order"""
    allr = _cat(per_rank)
    return allr[np.argsort(allr["query"], kind="stable")]
