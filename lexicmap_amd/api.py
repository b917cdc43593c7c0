"""ctypes view of include/lexicmap_hip.h.  No compute happens in Python and there is no fallback: if the HIP library is
missing or no GPU is present the calls raise."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(HERE, "liblexicmap_hip.so")
# LEXICMAP_HIP_LIB=<path>: another build of the same C-ABI (an experiment from experiments/, an A-B build) instead of the in-tree
# library; said on stderr so that no measurement is attributed to the wrong sources
if os.environ.get("LEXICMAP_HIP_LIB"):
    LIB_PATH = os.path.abspath(os.environ["LEXICMAP_HIP_LIB"])
    import sys as _sys
    print("[lexicmap_amd] library override: %s" % LIB_PATH, file=_sys.stderr)


class HipLibraryMissing(RuntimeError):
    pass


class Options(C.Structure):
    _fields_ = [("min_prefix", C.c_int32), ("min_single_prefix", C.c_int32), ("top_n_genomes", C.c_int32),
                ("top_n_chains", C.c_int32), ("max_gap", C.c_double), ("max_distance", C.c_double),
                ("ext_len", C.c_int32), ("ext_len2", C.c_int32), ("min_qcov_per_genome", C.c_double),
                ("max_evalue", C.c_double), ("output_seq", C.c_int32), ("align_max_gap", C.c_int32),
                ("align_band", C.c_int32), ("align_min_match_len", C.c_int32), ("align_min_pident", C.c_double),
                ("min_qcov_per_hsp", C.c_double), ("shard_rank", C.c_int32), ("shard_count", C.c_int32),
                ("total_bases_override", C.c_int64)]


class IndexInfo(C.Structure):
    _fields_ = [("k", C.c_int32), ("masks", C.c_int32), ("mask_prefix", C.c_int32), ("anchor_prefix", C.c_int32),
                ("total_bases", C.c_int64), ("genomes", C.c_int64), ("seeds", C.c_int64), ("genome_bases", C.c_int64),
                ("hbm_bytes", C.c_int64), ("seed_bytes", C.c_int64), ("outlier_seeds", C.c_int64),
                ("key_bits", C.c_int32), ("val_bits", C.c_int32), ("partition_bases", C.c_int32), ("pad", C.c_int32)]


class Query(C.Structure):
    _fields_ = [("seq", C.c_char_p), ("len", C.c_uint32)]


class Hsp(C.Structure):
    _fields_ = [("query", C.c_uint32), ("hits", C.c_uint32), ("batch_genome", C.c_uint64), ("qcov_genome", C.c_double),
                ("cls", C.c_int32), ("hsp", C.c_int32), ("seq_idx", C.c_int32), ("nseqs", C.c_int32),
                ("seq_len", C.c_int32), ("nchunks", C.c_int32), ("chunk_idx", C.c_int32), ("rc", C.c_int32),
                ("qcov_hsp", C.c_double), ("aligned_length", C.c_int32), ("pident", C.c_double), ("gaps", C.c_int32),
                ("qbegin", C.c_int32), ("qend", C.c_int32), ("tbegin", C.c_int32), ("tend", C.c_int32),
                ("evalue", C.c_double), ("bitscore", C.c_int32), ("score", C.c_int32), ("matched_bases", C.c_int32),
                ("genome_id", C.c_char_p), ("seq_id", C.c_char_p), ("cigar", C.c_char_p), ("qseq", C.c_char_p),
                ("sseq", C.c_char_p), ("align", C.c_char_p)]


class StageStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("query_bases", "query_kmers", "seed_lookups", "seed_values", "anchors_raw",
                                         "genome_pairs", "anchors_cleared", "chains", "window_bases", "pa_anchors",
                                         "hsps_aligned", "wfa_retries", "rows", "aligned_bases")] + \
               [(n, C.c_double) for n in ("ms_mask", "ms_lookup", "ms_chain", "ms_window", "ms_pseudo", "ms_glue",
                                          "ms_extend_wfa", "ms_finalize", "ms_total")]


class Anchor(C.Structure):
    _fields_ = [("qbegin", C.c_int32), ("tbegin", C.c_int32), ("len", C.c_uint8), ("trc", C.c_uint8),
                ("qrc", C.c_uint8), ("pad", C.c_uint8)]


class Pair(C.Structure):
    _fields_ = [("query", C.c_uint32), ("batch_genome", C.c_uint64), ("raw_off", C.c_int64), ("raw_n", C.c_int64),
                ("clr_off", C.c_int64), ("clr_n", C.c_int64), ("score", C.c_float), ("chain_off", C.c_int64),
                ("chain_n", C.c_int64)]


class Chain2(C.Structure):
    _fields_ = [("qbegin", C.c_int32), ("qend", C.c_int32), ("tbegin", C.c_int32), ("tend", C.c_int32),
                ("nanchors", C.c_int32), ("matched_bases", C.c_int32), ("aligned_bases_q", C.c_int32),
                ("aligned_bases_t", C.c_int32), ("pident", C.c_double)]


class Wfa(C.Structure):
    _fields_ = [("status", C.c_int32), ("score", C.c_int32), ("qbegin", C.c_int32), ("qend", C.c_int32),
                ("tbegin", C.c_int32), ("tend", C.c_int32), ("align_len", C.c_uint32), ("matches", C.c_uint32),
                ("gaps", C.c_uint32), ("gap_regions", C.c_uint32), ("ops_off", C.c_int64), ("nops", C.c_int32)]


class SynthSpec(C.Structure):
    _fields_ = [("k", C.c_int32), ("masks", C.c_int32), ("mask_seed", C.c_int64), ("genomes", C.c_int64),
                ("genome_len", C.c_int32), ("families", C.c_int32), ("max_div", C.c_double), ("seed", C.c_int64),
                ("max_desert", C.c_int32), ("seed_dist", C.c_int32)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char_p), ("launches", C.c_int64), ("total_ms", C.c_double), ("bytes", C.c_int64)]


def build_library(force=False):
    """hipcc --offload-arch=gfx950 build of the in-tree shared library (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h"))]
    srcs.append(os.path.join(os.path.dirname(HERE), "include", "lexicmap_hip.h"))
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", CSRC])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing("lexicmap_amd/liblexicmap_hip.so is missing: run __graft_entry__.build() "
                                "(there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.lm_options_default.argtypes = [C.POINTER(Options)]
    L.lm_index_open.argtypes = [C.c_char_p, C.POINTER(Options), C.c_int, C.POINTER(vp)]
    L.lm_index_close.argtypes = [vp]
    L.lm_index_get_info.argtypes = [vp, C.POINTER(IndexInfo)]
    L.lm_index_masks.argtypes = [vp]
    L.lm_index_masks.restype = C.POINTER(C.c_uint64)
    L.lm_last_error.argtypes = [vp]
    L.lm_last_error.restype = C.c_char_p
    L.lm_qbatch_upload.argtypes = [vp, C.POINTER(Query), C.c_size_t, C.POINTER(vp)]
    L.lm_qbatch_free.argtypes = [vp]
    L.lm_search_resident.argtypes = [vp, vp, C.POINTER(vp)]
    L.lm_search_batch.argtypes = [vp, C.POINTER(Query), C.c_size_t, C.POINTER(vp)]
    L.lm_result_rows.argtypes = [vp, C.POINTER(C.POINTER(Hsp))]
    L.lm_result_rows.restype = C.c_size_t
    L.lm_result_stats.argtypes = [vp, C.POINTER(StageStats)]
    L.lm_result_free.argtypes = [vp]
    L.lm_format_row.argtypes = [C.POINTER(Hsp), C.c_char_p, C.c_uint32, C.c_int, C.c_char_p, C.c_size_t]
    L.lm_stage_free.argtypes = [vp]
    L.lm_mask_batch.argtypes = [vp, C.POINTER(Query), C.c_size_t, C.POINTER(vp), C.POINTER(C.POINTER(C.c_uint64)),
                                C.POINTER(C.POINTER(C.c_int64)), C.POINTER(C.POINTER(C.c_int32))]
    L.lm_seed_chain_batch.argtypes = [vp, C.POINTER(Query), C.c_size_t, C.POINTER(vp), C.POINTER(C.c_size_t),
                                      C.POINTER(C.POINTER(Pair)), C.POINTER(C.POINTER(Anchor)),
                                      C.POINTER(C.POINTER(Anchor)), C.POINTER(C.POINTER(C.c_int64)),
                                      C.POINTER(C.POINTER(C.c_int32))]
    L.lm_pseudoalign_batch.argtypes = [vp, C.POINTER(Query), C.c_size_t, C.POINTER(Query), C.POINTER(C.c_uint32),
                                       C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(vp),
                                       C.POINTER(C.POINTER(C.c_int64)), C.POINTER(C.POINTER(Chain2))]
    L.lm_wfa_batch.argtypes = [vp, C.POINTER(Query), C.POINTER(Query), C.c_size_t, C.POINTER(vp),
                               C.POINTER(C.POINTER(Wfa)), C.POINTER(C.POINTER(C.c_uint64))]
    L.lm_index_build_synthetic.argtypes = [C.POINTER(SynthSpec), C.POINTER(Options), C.c_int, C.POINTER(vp)]
    L.lm_merge_sharded.argtypes = [vp, C.POINTER(C.POINTER(Hsp)), C.POINTER(C.c_size_t), C.c_int, C.POINTER(vp)]
    L.lm_search_scores.argtypes = [vp, vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(C.POINTER(C.c_uint32)),
                                   C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.POINTER(C.c_float))]
    L.lm_topn_merge.argtypes = [C.c_int, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.POINTER(C.c_uint64)),
                                C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_size_t), C.c_int,
                                C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_size_t)]
    L.lm_search_resident_keep.argtypes = [vp, vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(vp)]
    L.lm_free.argtypes = [vp]
    L.lm_index_set_genome_filter.argtypes = [vp, C.POINTER(C.c_uint64), C.c_size_t]
    L.lm_index_mask_seeds.argtypes = [vp, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_size_t,
                                      C.POINTER(C.c_size_t)]
    L.lm_index_fetch.argtypes = [vp, C.c_int64, C.c_int64, C.c_int64, C.c_char_p]
    L.lm_index_save.argtypes = [vp, C.c_char_p, C.c_int]
    L.lm_profile_enable.argtypes = [vp, C.c_int]
    L.lm_profile_reset.argtypes = [vp]
    L.lm_profile_exclusive.argtypes = [vp, C.c_int]
    L.lm_format_rows.argtypes = [C.POINTER(Hsp), C.c_size_t, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.c_size_t, C.c_int,
                                 C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.lm_free.argtypes = [vp]
    L.lm_free.restype = None
    L.lm_profile_mark.argtypes = [vp, C.c_int]
    L.lm_profile_mark.restype = None
    L.lm_tuning_reload.argtypes = [vp]
    L.lm_tuning_reload.restype = None
    L.lm_profile_get.argtypes = [vp, C.POINTER(C.POINTER(KernelTime))]
    L.lm_profile_get.restype = C.c_size_t
    L.lm_comm_unique_id.argtypes = [C.c_char_p]
    L.lm_comm_init.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.lm_comm_free.argtypes = [vp]
    L.lm_comm_free.restype = None
    L.lm_comm_rank.argtypes = [vp]
    L.lm_comm_size.argtypes = [vp]
    L.lm_comm_last_error.argtypes = [vp]
    L.lm_comm_last_error.restype = C.c_char_p
    L.lm_gather_rows.argtypes = [vp, C.POINTER(Hsp), C.c_size_t, C.c_int, C.POINTER(C.POINTER(Hsp)), C.POINTER(C.c_size_t)]
    L.lm_gather_merge_rows.argtypes = [vp, vp, C.POINTER(Hsp), C.c_size_t, C.c_int, C.POINTER(C.POINTER(Hsp)), C.POINTER(C.c_size_t)]
    L.lm_merge_sharded_device.argtypes = [vp, vp, vp, C.POINTER(C.c_size_t), C.c_int, C.POINTER(C.POINTER(Hsp)), C.POINTER(C.c_size_t)]
    _lib = L
    return L


def _queries(seqs):
    arr = (Query * max(len(seqs), 1))()
    keep = []
    for i, s in enumerate(seqs):
        b = bytes(s)
        keep.append(b)
        arr[i].seq = b
        arr[i].len = len(b)
    return arr, keep


def default_options(**kw):
    o = Options()
    lib().lm_options_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class Index:
    """lm_index handle (replaces cmd.NewIndexSearcher / Index.Search / Index.Close of the reference)."""

    def __init__(self, path, options=None, device=0, _handle=None):
        L = lib()
        self.opt = options or default_options()
        if _handle is not None:
            self.h = _handle
            return
        h = C.c_void_p()
        st = L.lm_index_open(path.encode(), C.byref(self.opt), device, C.byref(h))
        if st != 0:
            raise RuntimeError("lm_index_open failed (%d): %s" % (st, L.lm_last_error(None).decode()))
        self.h = h

    @classmethod
    def synthetic(cls, genomes, genome_len, families, seed=1000, max_div=0.10, masks=20000, options=None, device=0):
        """genomes + seed index generated directly in HBM by lm_index_build_synthetic (bench input)"""
        L = lib()
        opt = options or default_options()
        sp = SynthSpec(31, masks, 1, genomes, genome_len, families, max_div, seed, 100, 50)
        h = C.c_void_p()
        st = L.lm_index_build_synthetic(C.byref(sp), C.byref(opt), device, C.byref(h))
        if st != 0:
            raise RuntimeError("lm_index_build_synthetic failed (%d): %s" % (st, L.lm_last_error(None).decode()))
        return cls(None, opt, device, _handle=h)

    def save(self, path, chunks=16):
        """the resident index written in the reference's on-disk format (lm_index_save)"""
        st = lib().lm_index_save(self.h, path.encode(), chunks)
        if st != 0:
            self._err(st)

    def fetch(self, local_genome, start, length):
        buf = C.create_string_buffer(length)
        st = lib().lm_index_fetch(self.h, local_genome, start, length, buf)
        if st != 0:
            self._err(st)
        return buf.raw

    def close(self):
        if self.h:
            lib().lm_index_close(self.h)
            self.h = None

    def _err(self, st):
        raise RuntimeError("liblexicmap_hip error %d: %s" % (st, lib().lm_last_error(self.h).decode()))

    def info(self):
        i = IndexInfo()
        lib().lm_index_get_info(self.h, C.byref(i))
        return {f[0]: getattr(i, f[0]) for f in IndexInfo._fields_}

    def mask_seeds(self, mask):
        """(k-mers, values) stored under one mask as numpy uint64 arrays (reference value layout)"""
        import numpy as np
        n = C.c_size_t()
        st = lib().lm_index_mask_seeds(self.h, mask, None, None, 0, C.byref(n))
        if st != 0:
            self._err(st)
        k = np.zeros(max(n.value, 1), dtype=np.uint64)
        v = np.zeros(max(n.value, 1), dtype=np.uint64)
        st = lib().lm_index_mask_seeds(self.h, mask, k.ctypes.data_as(C.POINTER(C.c_uint64)),
                                       v.ctypes.data_as(C.POINTER(C.c_uint64)), n.value, C.byref(n))
        if st != 0:
            self._err(st)
        return k[:n.value], v[:n.value]

    def upload(self, seqs):
        arr, keep = _queries(seqs)
        qb = C.c_void_p()
        st = lib().lm_qbatch_upload(self.h, arr, len(seqs), C.byref(qb))
        if st != 0:
            self._err(st)
        return qb

    def free_batch(self, qb):
        lib().lm_qbatch_free(qb)

    def search_resident(self, qb, want_rows=True):
        L = lib()
        res = C.c_void_p()
        st = L.lm_search_resident(self.h, qb, C.byref(res))
        if st != 0:
            self._err(st)
        out = self._collect(res, want_rows)
        L.lm_result_free(res)
        return out

    def search_resident_np(self, qb):
        """rows as a numpy structured array VIEW of the library's lm_hsp rows (no copy; the lm_result is released when
        the array is garbage collected) + stats. The six `char *` columns are process-local addresses (valid while the
        array / the index are alive); lexicmap_amd.merge zeroes them before rows leave the process."""
        import weakref
        import numpy as np
        from .merge import ROW_DTYPE
        L = lib()
        res = C.c_void_p()
        st = L.lm_search_resident(self.h, qb, C.byref(res))
        if st != 0:
            self._err(st)
        rows_p = C.POINTER(Hsp)()
        n = L.lm_result_rows(res, C.byref(rows_p))
        stats = StageStats()
        L.lm_result_stats(res, C.byref(stats))
        if n:
            buf = (C.c_char * (n * C.sizeof(Hsp))).from_address(C.addressof(rows_p.contents))
            arr = np.frombuffer(buf, dtype=ROW_DTYPE)
            weakref.finalize(buf, L.lm_result_free, res)  # arr -> buf keeps the rows alive
        else:
            arr = np.zeros(0, dtype=ROW_DTYPE)
            L.lm_result_free(res)
        return arr, {f[0]: getattr(stats, f[0]) for f in StageStats._fields_}

    def set_genome_filter(self, keys):
        """genome whitelist (batch<<17|index keys) for the searches that follow; empty / None clears it"""
        keys = list(keys or [])
        arr = (C.c_uint64 * max(len(keys), 1))(*keys)
        st = lib().lm_index_set_genome_filter(self.h, arr, len(keys))
        if st != 0:
            self._err(st)

    def search_scores(self, qb):
        """(query, batch_genome, score) numpy arrays: this shard's candidates for the -n cut"""
        import numpy as np
        L = lib()
        sg, n = C.c_void_p(), C.c_size_t()
        q, g, s = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint64)(), C.POINTER(C.c_float)()
        st = L.lm_search_scores(self.h, qb, C.byref(sg), C.byref(n), C.byref(q), C.byref(g), C.byref(s))
        if st != 0:
            self._err(st)
        k = n.value
        out = (np.ctypeslib.as_array(q, shape=(k,)).copy() if k else np.zeros(0, np.uint32),
               np.ctypeslib.as_array(g, shape=(k,)).copy() if k else np.zeros(0, np.uint64),
               np.ctypeslib.as_array(s, shape=(k,)).copy() if k else np.zeros(0, np.float32))
        L.lm_stage_free(sg)
        return out

    def search_resident_keep(self, qb, keep_query, keep_bg):
        import numpy as np
        L = lib()
        kq = np.ascontiguousarray(keep_query, dtype=np.uint32)
        kg = np.ascontiguousarray(keep_bg, dtype=np.uint64)
        res = C.c_void_p()
        st = L.lm_search_resident_keep(self.h, qb, kq.ctypes.data_as(C.POINTER(C.c_uint32)),
                                       kg.ctypes.data_as(C.POINTER(C.c_uint64)), len(kq), C.byref(res))
        if st != 0:
            self._err(st)
        out = self._collect(res)
        L.lm_result_free(res)
        return out

    def _collect(self, res, want_rows=True):
        L = lib()
        rows_p = C.POINTER(Hsp)()
        n = L.lm_result_rows(res, C.byref(rows_p))
        stats = StageStats()
        L.lm_result_stats(res, C.byref(stats))
        rows = []
        if want_rows:
            for i in range(n):
                r = rows_p[i]
                rows.append({f[0]: getattr(r, f[0]) for f in Hsp._fields_})
        return rows, {f[0]: getattr(stats, f[0]) for f in StageStats._fields_}

    def search(self, seqs):
        """rows (list of dict) for a batch of query sequences + stage statistics"""
        L = lib()
        arr, keep = _queries(seqs)
        res = C.c_void_p()
        st = L.lm_search_batch(self.h, arr, len(seqs), C.byref(res))
        if st != 0:
            self._err(st)
        out = self._collect(res)
        L.lm_result_free(res)
        return out

    def search_tsv(self, ids, seqs, more_columns=False):
        L = lib()
        arr, keep = _queries(seqs)
        res = C.c_void_p()
        st = L.lm_search_batch(self.h, arr, len(seqs), C.byref(res))
        if st != 0:
            self._err(st)
        rows_p = C.POINTER(Hsp)()
        n = L.lm_result_rows(res, C.byref(rows_p))
        size = 1 << 22 if more_columns else 1 << 16
        buf = C.create_string_buffer(size)
        lines = []
        for i in range(n):
            q = rows_p[i].query
            L.lm_format_row(C.byref(rows_p[i]), ids[q].encode(), len(seqs[q]), int(more_columns), buf, size)
            lines.append(buf.value.decode())
        L.lm_result_free(res)
        return lines

    # ---- stage-level entry points (parity tests) ----
    def mask(self, seqs):
        L = lib()
        arr, keep = _queries(seqs)
        sg, km, lo, lc = C.c_void_p(), C.POINTER(C.c_uint64)(), C.POINTER(C.c_int64)(), C.POINTER(C.c_int32)()
        st = L.lm_mask_batch(self.h, arr, len(seqs), C.byref(sg), C.byref(km), C.byref(lo), C.byref(lc))
        if st != 0:
            self._err(st)
        M = self.info()["masks"]
        n = len(seqs) * M
        kmers = [km[i] for i in range(n)]
        off = [lo[i] for i in range(n + 1)]
        locs = [lc[i] for i in range(off[n])]
        L.lm_stage_free(sg)
        return kmers, off, locs

    def seed_chain(self, seqs):
        L = lib()
        arr, keep = _queries(seqs)
        sg, npairs = C.c_void_p(), C.c_size_t()
        pairs, raw, clr = C.POINTER(Pair)(), C.POINTER(Anchor)(), C.POINTER(Anchor)()
        cptr, cidx = C.POINTER(C.c_int64)(), C.POINTER(C.c_int32)()
        st = L.lm_seed_chain_batch(self.h, arr, len(seqs), C.byref(sg), C.byref(npairs), C.byref(pairs), C.byref(raw),
                                   C.byref(clr), C.byref(cptr), C.byref(cidx))
        if st != 0:
            self._err(st)
        out = []
        tup = lambda a: (a.qbegin, a.tbegin, a.len, a.qrc, a.trc)
        for i in range(npairs.value):
            p = pairs[i]
            chains = []
            for c in range(p.chain_n):
                chains.append([cidx[j] for j in range(cptr[p.chain_off + c], cptr[p.chain_off + c + 1])])
            out.append(dict(query=p.query, genome=p.batch_genome,
                            raw=[tup(raw[p.raw_off + j]) for j in range(p.raw_n)],
                            cleared=[tup(clr[p.clr_off + j]) for j in range(p.clr_n)],
                            score=p.score, chains=chains))
        L.lm_stage_free(sg)
        return out

    def pseudoalign(self, queries, problems):
        """problems: list of (query index, qbegin, qend, target bytes)"""
        L = lib()
        qarr, k1 = _queries(queries)
        tarr, k2 = _queries([p[3] for p in problems])
        n = len(problems)
        qi = (C.c_uint32 * max(n, 1))(*[p[0] for p in problems])
        qb = (C.c_uint32 * max(n, 1))(*[p[1] for p in problems])
        qe = (C.c_uint32 * max(n, 1))(*[p[2] for p in problems])
        sg, ro, rv = C.c_void_p(), C.POINTER(C.c_int64)(), C.POINTER(Chain2)()
        st = L.lm_pseudoalign_batch(self.h, qarr, len(queries), tarr, qi, qb, qe, n, C.byref(sg), C.byref(ro),
                                    C.byref(rv))
        if st != 0:
            self._err(st)
        out = []
        for i in range(n):
            out.append([{f[0]: getattr(rv[j], f[0]) for f in Chain2._fields_} for j in range(ro[i], ro[i + 1])])
        L.lm_stage_free(sg)
        return out

    def wfa(self, pairs):
        L = lib()
        qarr, k1 = _queries([p[0] for p in pairs])
        tarr, k2 = _queries([p[1] for p in pairs])
        sg, rv, ops = C.c_void_p(), C.POINTER(Wfa)(), C.POINTER(C.c_uint64)()
        st = L.lm_wfa_batch(self.h, qarr, tarr, len(pairs), C.byref(sg), C.byref(rv), C.byref(ops))
        if st != 0:
            self._err(st)
        out = []
        for i in range(len(pairs)):
            r = rv[i]
            out.append(dict(status=r.status, score=r.score, ops=[ops[r.ops_off + j] for j in range(r.nops)],
                            qbegin=r.qbegin, qend=r.qend, tbegin=r.tbegin, tend=r.tend, align_len=r.align_len,
                            matches=r.matches, gaps=r.gaps, gap_regions=r.gap_regions))
        L.lm_stage_free(sg)
        return out

    def profile(self, on=True):
        lib().lm_profile_enable(self.h, int(on))

    def profile_reset(self):
        lib().lm_profile_reset(self.h)

    def profile_exclusive(self, on=True):
        """kernels of the following searches one after the other: exclusive per-kernel times (measurement only)"""
        lib().lm_profile_exclusive(self.h, int(on))

    def profile_mark(self, ident):
        """an empty marker kernel between idle points of the device: where a rocprofv3 pass is cut (measurement only)"""
        lib().lm_profile_mark(self.h, int(ident))

    def tuning_reload(self):
        """re-read the LM_* experiment switches from the environment (measurement only: results do not depend on them)"""
        lib().lm_tuning_reload(self.h)

    def profile_get(self):
        p = C.POINTER(KernelTime)()
        n = lib().lm_profile_get(self.h, C.byref(p))
        return [dict(name=p[i].name.decode(), launches=p[i].launches, total_ms=p[i].total_ms, bytes=p[i].bytes)
                for i in range(n)]


def format_rows(rows, ids, lens, flags=0, want_text=True):
    """lm_format_rows: the TSV text (bytes) of a numpy row array (merge.ROW_DTYPE, pointer columns live in this process); ids /
    lens: id and length of every batch query.  want_text=False: only (bytes, seconds) - the text is released unseen."""
    import time
    import numpy as np
    from .merge import ROW_DTYPE
    L = lib()
    arr = np.ascontiguousarray(rows, dtype=ROW_DTYPE)
    idarr = (C.c_char_p * len(ids))(*[i if isinstance(i, bytes) else str(i).encode() for i in ids])
    lnarr = (C.c_uint32 * len(lens))(*[int(x) for x in lens])
    text, n = C.c_void_p(), C.c_size_t(0)
    t0 = time.time()
    st = L.lm_format_rows(arr.ctypes.data_as(C.POINTER(Hsp)), len(arr), idarr, lnarr, len(ids), flags, C.byref(text), C.byref(n))
    dt = time.time() - t0
    if st != 0:
        raise RuntimeError("lm_format_rows failed (%d)" % st)
    out = C.string_at(text, n.value) if want_text else None
    L.lm_free(text)
    return (out if want_text else n.value), dt


def row_names(rows, i):
    """(genome_id, seq_id) of row i of a numpy row array (merge.ROW_DTYPE) whose pointer columns are live in THIS process"""
    out = []
    for f in ("genome_id", "seq_id"):
        a = int(rows[f][i])
        out.append(C.string_at(a).decode() if a else None)
    return tuple(out)


COMM_ID_BYTES = 128


class Comm:
    """ctypes view of lm_comm (include/lexicmap_hip.h): the RCCL communicator of the sharded search's row gather.
    Rank 0 calls Comm.unique_id() and hands the 128 bytes to the other ranks by the host's own means (bench.py: a
    torch.distributed broadcast); every rank then opens Comm(id, nranks, rank, device)."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(COMM_ID_BYTES)
        st = lib().lm_comm_unique_id(buf)
        if st != 0:
            raise RuntimeError("lm_comm_unique_id failed (%d): %s" % (st, (lib().lm_comm_last_error(None) or b"").decode()))
        return buf.raw

    def __init__(self, uid, nranks, rank, device=0):
        self.L = lib()
        h = C.c_void_p()
        st = self.L.lm_comm_init(bytes(uid), nranks, rank, device, C.byref(h))
        if st != 0:
            raise RuntimeError("lm_comm_init failed (%d): %s" % (st, (self.L.lm_comm_last_error(None) or b"").decode()))
        self.h = h
        self.nranks, self.rank = nranks, rank

    def close(self):
        if self.h:
            self.L.lm_comm_free(self.h)
            self.h = None

    def gather_rows(self, arr, root=0):
        """arr: numpy rows of merge.ROW_DTYPE (this rank's).  -> (list of per-rank row arrays on `root` - views of the
        communicator's buffer, valid until the next call - or None elsewhere, the per-rank counts)"""
        import numpy as np
        from .merge import ROW_DTYPE
        arr = np.ascontiguousarray(arr, dtype=ROW_DTYPE)
        allp = C.POINTER(Hsp)()
        cnt = (C.c_size_t * self.nranks)()
        st = self.L.lm_gather_rows(self.h, arr.ctypes.data_as(C.POINTER(Hsp)), len(arr), root, C.byref(allp), cnt)
        if st != 0:
            raise RuntimeError("lm_gather_rows failed (%d): %s" % (st, (self.L.lm_comm_last_error(self.h) or b"").decode()))
        counts = [int(x) for x in cnt]
        if self.rank != root:
            return None, counts
        total = sum(counts)
        if total == 0:
            return [np.zeros(0, dtype=ROW_DTYPE) for _ in counts], counts
        buf = (C.c_char * (total * C.sizeof(Hsp))).from_address(C.addressof(allp.contents))
        allr = np.frombuffer(buf, dtype=ROW_DTYPE)
        out, o = [], 0
        for n in counts:
            out.append(allr[o:o + n])
            o += n
        return out, counts

    def merge_sharded_device(self, dev_ptr, counts, index=None):
        """lm_merge_sharded_device: rows of shard 0, 1, ... back to back in device memory at address dev_ptr (counts[r] rows each)
        -> the merged rows (a view of the communicator's pinned buffer, valid until its next call)"""
        import numpy as np
        from .merge import ROW_DTYPE
        outp = C.POINTER(Hsp)()
        total = C.c_size_t(0)
        cnt = (C.c_size_t * len(counts))(*[int(x) for x in counts])
        st = self.L.lm_merge_sharded_device(self.h, index.h if index is not None else None, C.c_void_p(int(dev_ptr)), cnt, len(counts),
                                            C.byref(outp), C.byref(total))
        if st != 0:
            raise RuntimeError("lm_merge_sharded_device failed (%d): %s" % (st, (self.L.lm_comm_last_error(self.h) or b"").decode()))
        if total.value == 0:
            return np.zeros(0, dtype=ROW_DTYPE)
        buf = (C.c_char * (total.value * C.sizeof(Hsp))).from_address(C.addressof(outp.contents))
        return np.frombuffer(buf, dtype=ROW_DTYPE)

    def gather_merge_rows(self, arr, root=0, index=None):
        """lm_gather_merge_rows: the gather and the merge (on the device) in one call.  -> on `root` the merged rows of all ranks
        in output order (a view of the communicator's pinned buffer, valid until its next call), None elsewhere.  index: the
        root's Index (names are re-attached from it) or None."""
        import numpy as np
        from .merge import ROW_DTYPE
        arr = np.ascontiguousarray(arr, dtype=ROW_DTYPE)
        outp = C.POINTER(Hsp)()
        total = C.c_size_t(0)
        st = self.L.lm_gather_merge_rows(self.h, index.h if index is not None else None, arr.ctypes.data_as(C.POINTER(Hsp)), len(arr), root,
                                         C.byref(outp), C.byref(total))
        if st != 0:
            raise RuntimeError("lm_gather_merge_rows failed (%d): %s" % (st, (self.L.lm_comm_last_error(self.h) or b"").decode()))
        if self.rank != root:
            return None
        if total.value == 0:
            return np.zeros(0, dtype=ROW_DTYPE)
        buf = (C.c_char * (total.value * C.sizeof(Hsp))).from_address(C.addressof(outp.contents))
        return np.frombuffer(buf, dtype=ROW_DTYPE)
