"""ctypes binding of the CPU oracle (oracle/liblmo.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; nothing in
lexicmap_amd/ does.
"""
import ctypes as C
import gzip
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(HERE), "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liblmo.so")


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(os.path.join(ORACLE_DIR, f)) > os.path.getmtime(LIB_PATH)
        for f in os.listdir(ORACLE_DIR)
        if f.endswith((".c", ".h"))
    ):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])
    return LIB_PATH


class BuildOpt(C.Structure):
    _fields_ = [("k", C.c_int), ("masks", C.c_int), ("rand_seed", C.c_int64), ("max_desert", C.c_int),
                ("seed_dist", C.c_int), ("chunks", C.c_int), ("partitions", C.c_int), ("batch_size", C.c_int),
                ("contig_interval", C.c_int), ("max_genome", C.c_int)]


class SearchOpt(C.Structure):
    _fields_ = [("min_prefix", C.c_int), ("min_single_prefix", C.c_int), ("top_n", C.c_int),
                ("top_n_chains", C.c_int), ("max_gap", C.c_double), ("max_distance", C.c_double),
                ("ext_len", C.c_int), ("ext_len2", C.c_int), ("min_qcov_genome", C.c_double),
                ("max_evalue", C.c_double), ("output_seq", C.c_int), ("align_max_gap", C.c_int),
                ("align_band", C.c_int), ("align_min_match_len", C.c_int), ("align_min_pident", C.c_double),
                ("min_qcov_hsp", C.c_double)]


class Hsp(C.Structure):
    _fields_ = [("batch_genome", C.c_uint64), ("qcov_genome", C.c_double), ("cls", C.c_int), ("hsp", C.c_int),
                ("seq_idx", C.c_int), ("nseqs", C.c_int), ("seq_len", C.c_int), ("nchunks", C.c_int),
                ("chunk_idx", C.c_int), ("rc", C.c_int), ("qcov_hsp", C.c_double), ("aligned_length", C.c_int),
                ("pident", C.c_double), ("gaps", C.c_int), ("qbegin", C.c_int), ("qend", C.c_int),
                ("tbegin", C.c_int), ("tend", C.c_int), ("evalue", C.c_double), ("bitscore", C.c_int),
                ("score", C.c_int), ("matched_bases", C.c_int), ("cigar", C.c_char_p), ("qseq", C.c_char_p),
                ("tseq", C.c_char_p), ("align", C.c_char_p), ("genome_id", C.c_char_p), ("seq_id", C.c_char_p)]


class Result(C.Structure):
    _fields_ = [("rows", C.POINTER(Hsp)), ("n", C.c_int), ("cap", C.c_int), ("ngenomes", C.c_int),
                ("n_seed_values", C.c_int64), ("n_anchors", C.c_int64), ("n_genomes_seeded", C.c_int64),
                ("n_chains", C.c_int64)]


class Sub(C.Structure):
    _fields_ = [("qbegin", C.c_int32), ("tbegin", C.c_int32), ("len", C.c_uint8), ("trc", C.c_uint8),
                ("qrc", C.c_uint8), ("_pad", C.c_uint8)]


class Anchor(C.Structure):
    _fields_ = [("genome", C.c_uint64), ("sub", Sub)]


class WfaResult(C.Structure):
    _fields_ = [("ops", C.POINTER(C.c_uint64)), ("nops", C.c_int), ("qbegin", C.c_int), ("qend", C.c_int),
                ("tbegin", C.c_int), ("tend", C.c_int), ("align_len", C.c_uint32), ("matches", C.c_uint32),
                ("gaps", C.c_uint32), ("gap_regions", C.c_uint32), ("score", C.c_int)]


class Chain2(C.Structure):
    _fields_ = [("nanchors", C.c_int), ("aligned_fraction", C.c_double), ("matched_bases", C.c_int),
                ("aligned_bases_q", C.c_int), ("aligned_bases_t", C.c_int), ("pident", C.c_double),
                ("aligned_length", C.c_int), ("gaps", C.c_int), ("qbegin", C.c_int), ("qend", C.c_int),
                ("tbegin", C.c_int), ("tend", C.c_int), ("max_ext_len", C.c_int), ("tpos_offset_begin", C.c_int),
                ("score", C.c_int), ("bitscore", C.c_int), ("evalue", C.c_double), ("cigar", C.c_char_p),
                ("qseq", C.c_char_p), ("tseq", C.c_char_p), ("align", C.c_char_p), ("alive", C.c_int)]


class Chain2Opt(C.Structure):
    _fields_ = [("max_gap", C.c_int), ("min_score", C.c_int), ("min_align_len", C.c_int),
                ("min_identity", C.c_double), ("band_count", C.c_int), ("band_base", C.c_int),
                ("heuristic_pident", C.c_double)]


class CmpOpt(C.Structure):
    _fields_ = [("k", C.c_int), ("min_prefix", C.c_int), ("c2", Chain2Opt), ("min_aligned_fraction", C.c_double),
                ("min_identity", C.c_double)]


class TreeSr(C.Structure):
    _fields_ = [("kmer", C.c_uint64), ("len_prefix", C.c_uint8), ("vals", C.POINTER(C.c_uint32)), ("nvals", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.lmo_builder_new.restype = C.c_void_p
        L.lmo_builder_new.argtypes = [C.c_char_p, C.POINTER(BuildOpt)]
        L.lmo_builder_add.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_char_p),
                                      C.POINTER(C.c_char_p), C.POINTER(C.c_int)]
        L.lmo_builder_finish.argtypes = [C.c_void_p]
        L.lmo_builder_set_masks.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
        L.lmo_index_open.restype = C.c_void_p
        L.lmo_index_open.argtypes = [C.c_char_p, C.POINTER(SearchOpt)]
        L.lmo_index_close.argtypes = [C.c_void_p]
        L.lmo_index_nmasks.argtypes = [C.c_void_p]
        L.lmo_index_k.argtypes = [C.c_void_p]
        L.lmo_index_masks.argtypes = [C.c_void_p]
        L.lmo_index_masks.restype = C.POINTER(C.c_uint64)
        L.lmo_index_total_bases.argtypes = [C.c_void_p]
        L.lmo_index_total_bases.restype = C.c_int64
        L.lmo_search.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(Result)]
        L.lmo_result_free.argtypes = [C.POINTER(Result)]
        L.lmo_format_row.argtypes = [C.POINTER(Hsp), C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
        L.lmo_stage_mask.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_uint64),
                                     C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.POINTER(C.c_int))]
        L.lmo_stage_anchors.restype = C.c_int64
        L.lmo_stage_anchors.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                        C.POINTER(C.POINTER(Anchor))]
        L.lmo_wfa_align.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.POINTER(WfaResult)]
        L.lmo_wfa_result_free.argtypes = [C.POINTER(WfaResult)]
        L.lmo_score_evalue.argtypes = [C.POINTER(WfaResult), C.c_int, C.c_int64, C.POINTER(C.c_int),
                                       C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.lmo_kmer_encode.restype = C.c_uint64
        L.lmo_kmer_encode.argtypes = [C.c_char_p, C.c_int]
        L.lmo_kmer_reverse.restype = C.c_uint64
        L.lmo_kmer_reverse.argtypes = [C.c_uint64, C.c_int]
        L.lmo_kmer_revcomp.restype = C.c_uint64
        L.lmo_kmer_revcomp.argtypes = [C.c_uint64, C.c_int]
        L.lmo_dust.argtypes = [C.c_uint64, C.c_int]
        L.lmo_low_complexity.argtypes = [C.c_uint64, C.c_int]
        L.lmo_gen_masks.argtypes = [C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_uint64)]
        L.lmo_seed_weight.restype = C.c_float
        L.lmo_seed_weight.argtypes = [C.c_float]
        L.lmo_gap_score.restype = C.c_float
        L.lmo_gap_score.argtypes = [C.c_float]
        L.lmo_go_log2.restype = C.c_double
        L.lmo_go_log2.argtypes = [C.c_double]
        L.lmo_clear_subs.argtypes = [C.POINTER(Sub), C.c_int, C.c_int]
        L.lmo_chainer.restype = C.c_float
        L.lmo_chainer.argtypes = [C.POINTER(Sub), C.c_int, C.c_float, C.c_float, C.c_float, C.c_int,
                                  C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.c_int)]
        L.lmo_chainer2.argtypes = [C.POINTER(Sub), C.c_int, C.POINTER(Chain2Opt), C.POINTER(C.POINTER(Chain2)),
                                   C.POINTER(C.c_int)]
        L.lmo_cmp_new.restype = C.c_void_p
        L.lmo_cmp_new.argtypes = [C.POINTER(CmpOpt)]
        L.lmo_cmp_free.argtypes = [C.c_void_p]
        L.lmo_cmp_index.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.lmo_cmp_compare.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_int, C.c_int,
                                      C.POINTER(C.POINTER(Chain2)), C.POINTER(C.POINTER(Sub)), C.POINTER(C.c_int)]
        L.lmo_extend_match.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int] + [C.c_int] * 8 + \
            [C.POINTER(C.c_int)] * 8
        L.lmo_tree_new.restype = C.c_void_p
        L.lmo_tree_new.argtypes = [C.c_int]
        L.lmo_tree_free.argtypes = [C.c_void_p]
        L.lmo_tree_insert.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
        L.lmo_tree_search.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.POINTER(TreeSr)), C.POINTER(C.c_int)]
        L.lmo_trim_subs.argtypes = [C.POINTER(Sub), C.c_int, C.c_int, C.c_float, C.POINTER(C.c_int)]
        L.lmo_lh_new.restype = C.c_void_p
        L.lmo_lh_new.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.c_int]
        L.lmo_lh_free.argtypes = [C.c_void_p]
        L.lmo_lh_mask.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int,
                                  C.POINTER(C.c_uint64), C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.POINTER(C.c_int))]
        L.free = C.CDLL(None).free
        L.free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def read_fasta(path):
    """returns list of (id, seq bytes upper-cased)"""
    op = gzip.open if path.endswith(".gz") else open
    recs, name, chunks = [], None, []
    with op(path, "rb") as f:
        for line in f:
            line = line.rstrip()
            if line.startswith(b">"):
                if name is not None:
                    recs.append((name, b"".join(chunks).upper()))
                name = line[1:].split()[0].decode()
                chunks = []
            elif line:
                chunks.append(line)
    if name is not None:
        recs.append((name, b"".join(chunks).upper()))
    return recs


def default_build_opt(**kw):
    o = BuildOpt()
    lib().lmo_build_opt_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def default_search_opt(**kw):
    o = SearchOpt()
    lib().lmo_search_opt_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def build_index(outdir, genomes, opt=None, masks=None):
    """genomes: list of (genome_id, [(contig_id, seq bytes), ...]); masks: mask set to use instead of the generated one"""
    L = lib()
    opt = opt or default_build_opt()
    b = L.lmo_builder_new(outdir.encode(), C.byref(opt))
    if masks is not None:
        arr = (C.c_uint64 * len(masks))(*masks)
        if L.lmo_builder_set_masks(b, arr, len(masks)) != 0:
            raise RuntimeError("lmo_builder_set_masks failed")
    for gid, contigs in genomes:
        n = len(contigs)
        ids = (C.c_char_p * n)(*[c[0].encode() for c in contigs])
        seqs = (C.c_char_p * n)(*[c[1] for c in contigs])
        lens = (C.c_int * n)(*[len(c[1]) for c in contigs])
        rc = L.lmo_builder_add(b, gid.encode(), n, ids, seqs, lens)
        if rc != 0:
            raise RuntimeError("lmo_builder_add failed for %s" % gid)
    L.lmo_builder_finish(b)


class Index:
    def __init__(self, path, opt=None):
        self.opt = opt or default_search_opt()
        self.h = lib().lmo_index_open(path.encode(), C.byref(self.opt))
        if not self.h:
            raise RuntimeError("cannot open index %s" % path)

    def close(self):
        if self.h:
            lib().lmo_index_close(self.h)
            self.h = None

    @property
    def nmasks(self):
        return lib().lmo_index_nmasks(self.h)

    def search(self, seq):
        """returns (rows as list of dicts, stats)"""
        L = lib()
        res = Result()
        rc = L.lmo_search(self.h, seq, len(seq), C.byref(res))
        if rc != 0:
            raise RuntimeError("lmo_search failed")
        rows = []
        for i in range(res.n):
            h = res.rows[i]
            rows.append({f[0]: getattr(h, f[0]) for f in Hsp._fields_})
        stats = dict(ngenomes=res.ngenomes, n_anchors=res.n_anchors, n_genomes_seeded=res.n_genomes_seeded,
                     n_chains=res.n_chains)
        L.lmo_result_free(C.byref(res))
        return rows, stats

    def search_tsv(self, qid, seq, more_columns=False):
        L = lib()
        res = Result()
        L.lmo_search(self.h, seq, len(seq), C.byref(res))
        out = []
        size = 1 << 16
        if more_columns:
            size = 1 << 22
        buf = C.create_string_buffer(size)
        for i in range(res.n):
            L.lmo_format_row(C.byref(res.rows[i]), qid.encode(), len(seq), res.ngenomes, int(more_columns), buf, size)
            out.append(buf.value.decode())
        L.lmo_result_free(C.byref(res))
        return out
