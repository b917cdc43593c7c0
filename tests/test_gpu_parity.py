"""GPU parity tests: the HIP path (through the C-ABI of liblexicmap_hip.so) against the CPU oracle on the same seeded
inputs.  Bit-exact for every integer / index / float32 quantity; e-value within 1e-9 relative (Go's math.Pow/Log vs libm).

Nothing here reads /root/reference: indexes are written by the oracle's synthetic-index writer into a tmp dir and
opened by both sides; the demo goldens are the committed fixtures under tests/golden/demo.
"""
import ctypes as C
import os
import random

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "demo")


def _la():
    import lexicmap_amd as la
    return la


@pytest.fixture(scope="module")
def small_index(tmp_path_factory):
    """24 genomes x ~120 kb in 4 families, 1-4 contigs each, some with N runs; M=20000 masks like the reference default"""
    from lexicmap_amd import synth
    d = str(tmp_path_factory.mktemp("idx") / "small.lmi")
    genomes = synth.make_genomes(24, 120000, 4, seed=11, max_div=0.12, contigs=(1, 4), with_n=True)
    O.build_index(d, genomes, O.default_build_opt(chunks=4))
    return d, genomes


@pytest.fixture(scope="module")
def queries(small_index):
    from lexicmap_amd import synth
    d, genomes = small_index
    qs = synth.make_gene_queries(genomes, 24, seed=5, len_range=(300, 2000), max_div=0.12)
    qs += synth.make_reads(genomes, 4, seed=6, len_range=(3000, 12000))
    rng = random.Random(3)
    qs.append(("short_lt_k", b"ACGTACGTACGTACGTACGT"))                      # shorter than k: no result
    qs.append(("exact_k", genomes[0][1][0][1][100:131]))                     # exactly one k-mer
    qs.append(("random_nohit", bytes(rng.choice(b"ACGT") for _ in range(700))))
    qs.append(("polyA", b"A" * 200))
    qs.append(("lowcomplex", b"ACACACACACACACACACACACACACACACACACACACACACACACACAC" * 3))
    with_n = bytearray(qs[0][1])
    with_n[50:55] = b"NNNNN"
    qs.append(("with_N", bytes(with_n)))
    return qs


@pytest.fixture(scope="module")
def both(small_index):
    la = _la()
    d, _ = small_index
    oi = O.Index(d)
    gi = la.Index(d)
    yield oi, gi
    gi.close()
    oi.close()


def test_index_info_and_masks(both, small_index):
    oi, gi = both
    info = gi.info()
    L = O.lib()
    assert info["k"] == L.lmo_index_k(oi.h) == 31
    assert info["masks"] == oi.nmasks == 20000
    assert info["total_bases"] == L.lmo_index_total_bases(oi.h)
    assert info["genomes"] == len(small_index[1])
    om = L.lmo_index_masks(oi.h)
    gm = _la().lib().lm_index_masks(gi.h)
    assert [om[i] for i in range(0, 20000, 97)] == [gm[i] for i in range(0, 20000, 97)]
    assert info["seeds"] > 0 and info["hbm_bytes"] > info["seed_bytes"] > 0
    # packed seed image: K-p-a bases per key, local genome | position | strand per value
    assert info["partition_bases"] == 6 and info["key_bits"] == 2 * (31 - 7 - 6)
    assert info["val_bits"] in (5 + 17 + 1, 5 + 18 + 1)  # 24 genomes, ~2^17 bases each (contigs + spacers), strand
    assert 0 <= info["outlier_seeds"] < info["seeds"]


class _KvMem(C.Structure):  # lmo_kv_mem (oracle/lmo.h): the reference's RAM form of a chunk file, kv-reader.go:762
    _fields_ = [("k", C.c_int), ("chunk_index", C.c_int), ("chunk_size", C.c_int), ("mask_prefix", C.c_int),
                ("anchor_prefix", C.c_int), ("use7", C.c_int), ("kv", C.POINTER(C.POINTER(C.c_uint64))),
                ("kvlen", C.POINTER(C.c_int64)), ("index", C.POINTER(C.POINTER(C.c_int64)))]


def test_packed_seed_image_holds_exactly_the_seeds_of_the_chunk_files(both, small_index):
    """a20 / f1: every (k-mer, value) the reference format stores under a mask comes back out of the packed HBM image
    (partition table + 36-bit k-mer remainders + packed values, outliers flat), nothing more, each half sorted"""
    _, gi = both
    d, _ = small_index
    L = O.lib()
    L.lmo_kv_load.restype = C.POINTER(_KvMem)
    L.lmo_kv_load.argtypes = [C.c_char_p]
    L.lmo_kv_free.argtypes = [C.POINTER(_KvMem)]
    total = 0
    for chunk in sorted(os.listdir(os.path.join(d, "seeds"))):
        if not chunk.endswith(".bin"):
            continue
        m = L.lmo_kv_load(os.path.join(d, "seeds", chunk).encode())
        km = m.contents
        for i in list(range(0, km.chunk_size, 41)) + [km.chunk_size - 1]:
            n = km.kvlen[i]
            flat = np.ctypeslib.as_array(km.kv[i], shape=(n,)).copy() if n else np.zeros(0, dtype=np.uint64)
            exp = sorted(zip(flat[0::2].tolist(), flat[1::2].tolist()))
            k, v = gi.mask_seeds(km.chunk_index + i)
            assert sorted(zip(k.tolist(), v.tolist())) == exp, km.chunk_index + i
            nrm = (v & np.uint64(1)) == 0
            first_rev = int(np.argmax(~nrm)) if (~nrm).any() else len(v)
            assert nrm[:first_rev].all() and (~nrm[first_rev:]).all()
            total += len(k)
        L.lmo_kv_free(m)
    assert total > 1000
    info = gi.info()
    assert info["seed_bytes"] > 0 and info["outlier_seeds"] >= 0


def test_mask_parity(both, queries):
    """a1+a2: LexicHash capture of every mask + low-complexity zeroing + all locations"""
    oi, gi = both
    L = O.lib()
    M = oi.nmasks
    seqs = [q[1] for q in queries]
    kmers, off, locs = gi.mask(seqs)
    for qi, s in enumerate(seqs):
        gk = kmers[qi * M:(qi + 1) * M]
        if len(s) < 31:
            assert not any(gk)
            continue
        ok = (C.c_uint64 * M)()
        ooff, olocs = C.POINTER(C.c_int)(), C.POINTER(C.c_int)()
        assert L.lmo_stage_mask(oi.h, s, len(s), ok, C.byref(ooff), C.byref(olocs)) == 0
        assert gk == list(ok), queries[qi][0]
        for m in range(M):
            a = locs[off[qi * M + m]:off[qi * M + m + 1]]
            b = [olocs[j] for j in range(ooff[m], ooff[m + 1])]
            assert a == b, (queries[qi][0], m)
        L.free(ooff)
        L.free(olocs)


def _oracle_pairs(oi, seq):
    """oracle per (genome): raw anchors (clear order), cleared anchors, score, chains"""
    L = O.lib()
    M = oi.nmasks
    if len(seq) < 31:
        return {}
    ok = (C.c_uint64 * M)()
    ooff, olocs = C.POINTER(C.c_int)(), C.POINTER(C.c_int)()
    L.lmo_stage_mask(oi.h, seq, len(seq), ok, C.byref(ooff), C.byref(olocs))
    anc = C.POINTER(O.Anchor)()
    na = L.lmo_stage_anchors(oi.h, ok, ooff, olocs, C.byref(anc))
    out = {}
    i = 0
    tup = lambda s: (s.qbegin, s.tbegin, s.len, s.qrc, s.trc)
    while i < na:
        j = i
        while j < na and anc[j].genome == anc[i].genome:
            j += 1
        n = j - i
        subs = (O.Sub * n)()
        for t in range(n):
            subs[t] = anc[i + t].sub
        raw = [tup(subs[t]) for t in range(n)]
        nn = L.lmo_clear_subs(subs, n, 31) if n > 1 else n
        cleared = [tup(subs[t]) for t in range(nn)]
        coff, cidx, nch = C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.c_int()
        sc = L.lmo_chainer(subs, nn, 50.0, L.lmo_seed_weight(17.0), 1000.0, 0, C.byref(coff), C.byref(cidx), C.byref(nch))
        chains = [[cidx[x] for x in range(coff[c], coff[c + 1])] for c in range(nch.value)]
        out[anc[i].genome] = dict(raw=raw, cleared=cleared, score=sc, chains=chains)
        L.free(coff)
        L.free(cidx)
        i = j
    L.free(anc)
    L.free(ooff)
    L.free(olocs)
    return out


def test_seed_lookup_and_chaining_parity(both, queries):
    """a3-a7: reverse re-bucketing, prefix+suffix lookup, anchor assembly, ClearSubstrPairs, Chainer.Chain"""
    oi, gi = both
    seqs = [q[1] for q in queries]
    pairs = gi.seed_chain(seqs)
    by_q = {}
    for p in pairs:
        by_q.setdefault(p["query"], {})[p["genome"]] = p
    total = 0
    for qi, s in enumerate(seqs):
        exp = _oracle_pairs(oi, s)
        got = by_q.get(qi, {})
        assert sorted(exp.keys()) == sorted(got.keys()), queries[qi][0]
        for g, e in exp.items():
            p = got[g]
            assert p["raw"] == e["raw"], (queries[qi][0], g)
            assert p["cleared"] == e["cleared"]
            assert np.float32(p["score"]).tobytes() == np.float32(e["score"]).tobytes()
            assert p["chains"] == e["chains"]
            total += len(e["raw"])
    assert total > 1000  # the test really exercised the path


def test_pseudoalign_parity(both, queries, small_index):
    """a10-a12: SeqComparator.Index/Compare incl. tree.Search quirk emulation, Clear, Trim, Chainer2"""
    oi, gi = both
    L = O.lib()
    d, genomes = small_index
    rng = random.Random(8)
    qseqs = [q[1] for q in queries if len(q[1]) >= 31][:20]
    problems = []
    for qi, q in enumerate(qseqs):
        # windows: a mutated copy of the query embedded in random flanks, an unrelated window, and a poly-A rich one
        from lexicmap_amd import synth
        nprng = np.random.default_rng(qi)
        core = synth.mutate(nprng, np.frombuffer(q, dtype=np.uint8), sub=0.08, ins=0.01, dele=0.01).tobytes()
        fl = bytes(rng.choice(b"ACGT") for _ in range(400))
        problems.append((qi, 0, len(q) - 1, fl + core + fl[::-1]))
        problems.append((qi, len(q) // 4, (3 * len(q)) // 4, core[:len(core) // 2] + b"A" * 45 + core[len(core) // 2:]))
        problems.append((qi, 0, len(q) - 1, bytes(rng.choice(b"ACGT") for _ in range(900))))
    got = gi.pseudoalign(qseqs, problems)
    opt = O.CmpOpt()
    opt.k, opt.min_prefix = 31, 11
    opt.c2.max_gap, opt.c2.min_score, opt.c2.min_align_len = 20, 35, 50
    opt.c2.min_identity, opt.c2.band_count, opt.c2.band_base, opt.c2.heuristic_pident = 70.0, 50, 100, 15.0
    nchains = 0
    cmp_cache = {}
    for (qi, qb, qe, t), g in zip(problems, got):
        if qi not in cmp_cache:
            c = L.lmo_cmp_new(C.byref(opt))
            L.lmo_cmp_index(c, qseqs[qi], len(qseqs[qi]))
            cmp_cache[qi] = c
        chains = C.POINTER(O.Chain2)()
        nc = L.lmo_cmp_compare(cmp_cache[qi], qb, qe, t, len(t), len(qseqs[qi]), C.byref(chains), None, None)
        assert len(g) == nc
        for i in range(nc):
            o = chains[i]
            assert (g[i]["qbegin"], g[i]["qend"], g[i]["tbegin"], g[i]["tend"], g[i]["nanchors"], g[i]["matched_bases"],
                    g[i]["aligned_bases_q"]) == (o.qbegin, o.qend, o.tbegin, o.tend, o.nanchors, o.matched_bases,
                                                 o.aligned_bases_q)
            assert g[i]["pident"] == o.pident
        nchains += nc
    for c in cmp_cache.values():
        L.lmo_cmp_free(c)
    assert nchains >= 20


def test_wfa_parity(both):
    """a15: gap-affine WFA (CIGAR ops, bounds, statistics) incl. the scratch-overflow retry protocol"""
    oi, gi = both
    from lexicmap_amd import synth
    L = O.lib()
    rng = np.random.default_rng(21)
    pairs = []
    for n, div in [(1, 0), (40, 0.0), (60, 0.3), (300, 0.05), (1500, 0.1), (1500, 0.25), (5000, 0.12), (200, 0.6),
                   (9000, 0.03), (800, 0.15)]:
        for rep in range(3):
            q = synth.random_seq(rng, n + rep)
            t = synth.mutate(rng, q, sub=div, ins=div / 4, dele=div / 4)
            if rep == 1 and n > 50:
                t = np.concatenate([t[:len(t) // 2], synth.random_seq(rng, 40), t[len(t) // 2:]])
            if len(t) == 0:
                t = synth.random_seq(rng, 3)
            pairs.append((q.tobytes(), t.tobytes()) if rep != 2 else (t.tobytes(), q.tobytes()))
    got = gi.wfa(pairs)
    for (q, t), g in zip(pairs, got):
        r = O.WfaResult()
        assert L.lmo_wfa_align(q, len(q), t, len(t), 1, C.byref(r)) == 0
        assert g["score"] == r.score
        assert g["ops"] == [r.ops[i] for i in range(r.nops)]
        assert (g["qbegin"], g["qend"], g["tbegin"], g["tend"], g["align_len"], g["matches"], g["gaps"],
                g["gap_regions"]) == (r.qbegin, r.qend, r.tbegin, r.tend, r.align_len, r.matches, r.gaps, r.gap_regions)
        L.lmo_wfa_result_free(C.byref(r))


ROW_INT = ["batch_genome", "cls", "hsp", "seq_idx", "nseqs", "seq_len", "rc", "aligned_length", "gaps", "qbegin",
           "qend", "tbegin", "tend", "bitscore", "score", "matched_bases"]
ROW_F64 = ["qcov_genome", "qcov_hsp", "pident"]


def _cmp_rows(exp, got, label):
    assert len(exp) == len(got), (label, len(exp), len(got))
    for e, g in zip(exp, got):
        for f in ROW_INT:
            assert e[f] == g[f], (label, f, e[f], g[f])
        for f in ROW_F64:
            assert e[f] == g[f], (label, f)
        assert g["evalue"] == pytest.approx(e["evalue"], rel=1e-9, abs=0)  # float tolerance of the north-star
        assert e["genome_id"] == g["genome_id"] and e["seq_id"] == g["seq_id"]


def test_full_search_parity(both, queries):
    """whole path: every HSP row of every query, in output order"""
    oi, gi = both
    seqs = [q[1] for q in queries]
    rows, stats = gi.search(seqs)
    by_q = {}
    for r in rows:
        by_q.setdefault(r["query"], []).append(r)
    nrows = 0
    for qi, s in enumerate(seqs):
        exp, st = oi.search(s)
        got = by_q.get(qi, [])
        _cmp_rows(exp, got, queries[qi][0])
        for g in got:
            assert g["hits"] == st["ngenomes"]
        nrows += len(exp)
    assert nrows > 50
    assert stats["rows"] == nrows and stats["chains"] > 0 and stats["hsps_aligned"] >= nrows


def test_chunked_alignment_and_anchor_buffer_rerun_give_the_same_rows(both, queries, monkeypatch):
    """the alignment half runs in chunks of whole (query, genome) segments when the windows of a batch exceed its window
    budget, and re-runs the anchor kernel when its output buffer estimate was too small: both paths, forced here through
    the library's test hooks, must return exactly the rows of the single-chunk run"""
    _oi, gi = both
    seqs = [q[1] for q in queries]
    base, st0 = gi.search(seqs)
    monkeypatch.setenv("LM_DEBUG_MAX_WINDOW_BYTES", "20000")   # a few chain windows per chunk
    monkeypatch.setenv("LM_DEBUG_PA_CAP", "64")                # anchor buffer far too small -> counted, re-run
    monkeypatch.setenv("LM_DEBUG_PA_WIDE_KEYS", "1")           # two-key anchors (the layout for > 64 key bits)
    monkeypatch.setenv("LM_DEBUG_ROUND_HSPS", "7")             # several extendMatch / WFA rounds, each over a few chunks
    got, st1 = gi.search(seqs)
    monkeypatch.delenv("LM_DEBUG_ROUND_HSPS")
    monkeypatch.delenv("LM_DEBUG_MAX_WINDOW_BYTES")
    monkeypatch.delenv("LM_DEBUG_PA_CAP")
    monkeypatch.delenv("LM_DEBUG_PA_WIDE_KEYS")
    assert len(base) == len(got) and len(base) > 50
    for b, g in zip(base, got):
        assert b == g
    assert st0["rows"] == st1["rows"] and st0["pa_anchors"] == st1["pa_anchors"]


def test_out_of_memory_during_a_search_halves_the_batch_and_gives_the_same_rows(both, queries, monkeypatch):
    """a scratch allocation that fails (forced through the library's test hook, between the seeding and the alignment
    halves of every part above the given number of queries) makes the search drop its scratch and retry on half the
    queries; the rows are those of the undisturbed run, in the caller's query order"""
    _oi, gi = both
    seqs = [q[1] for q in queries]
    base, st0 = gi.search(seqs)
    monkeypatch.setenv("LM_DEBUG_OOM_ABOVE_QUERIES", str(max(1, len(seqs) // 3)))  # the batch ends up in quarters
    got, st1 = gi.search(seqs)
    monkeypatch.delenv("LM_DEBUG_OOM_ABOVE_QUERIES")
    assert len(base) == len(got) and len(base) > 50
    for b, g in zip(base, got):
        assert b == g
    assert st0["rows"] == st1["rows"]
    again, _ = gi.search(seqs)  # and the handle is as good as new afterwards
    assert again == base


def test_search_options_topn_and_all_columns(small_index, queries):
    """-n/--top-n-genomes, -N/--top-n-chains and -a/--all (CIGAR/qseq/sseq/align strings)"""
    la = _la()
    d, _ = small_index
    oi = O.Index(d, O.default_search_opt(top_n=3, top_n_chains=2, output_seq=1, min_qcov_hsp=5.0, min_qcov_genome=10.0))
    gi = la.Index(d, la.api.default_options(top_n_genomes=3, top_n_chains=2, output_seq=1, min_qcov_per_hsp=5.0,
                                            min_qcov_per_genome=10.0))
    seqs = [q[1] for q in queries[:12]]
    rows, _ = gi.search(seqs)
    by_q = {}
    for r in rows:
        by_q.setdefault(r["query"], []).append(r)
    n = 0
    for qi, s in enumerate(seqs):
        exp, _st = oi.search(s)
        got = by_q.get(qi, [])
        _cmp_rows(exp, got, queries[qi][0])
        for e, g in zip(exp, got):
            assert (e["cigar"], e["qseq"], e["tseq"], e["align"]) == (g["cigar"], g["qseq"], g["sseq"], g["align"])
            n += 1
    assert n > 5
    gi.close()
    oi.close()


def test_demo_golden_rows(tmp_path):
    """committed golden of the reference (demo/q.gene.fasta.lexicmap_top-2-genomes_all.tsv): 14 rows with CIGAR, qseq,
    sseq and alignment text reproduced byte for byte through the HIP path"""
    la = _la()
    d = str(tmp_path / "demo2.lmi")
    genomes = []
    for f in ("GCF_002949675.1.fa.gz", "GCF_003697165.2.fa.gz"):
        genomes.append((f[:-6], O.read_fasta(os.path.join(GOLD, f))))
    O.build_index(d, genomes, O.default_build_opt(chunks=4))
    q = O.read_fasta(os.path.join(GOLD, "q.gene.fasta"))[0]
    gi = la.Index(d, la.api.default_options(top_n_genomes=2, output_seq=1))
    lines = gi.search_tsv([q[0]], [q[1]], more_columns=True)
    gold = open(os.path.join(GOLD, "q.gene.fasta.lexicmap_top-2-genomes_all.tsv")).read().rstrip("\n").split("\n")[1:]
    assert lines == gold
    gi.close()


def test_demo_c1_all_84_gene_rows_through_the_hip_path(tmp_path):
    """BASELINE configs[0]: demo/q.gene.fasta against all 15 genomes of demo/refs (committed fixtures): the 84 rows of the
    reference's own q.gene.fasta.lexicmap.tsv, every column, in order, through the C-ABI on the GPU"""
    la = _la()
    d = str(tmp_path / "demo15.lmi")
    files = [os.path.join(GOLD, f) for f in os.listdir(GOLD) if f.endswith(".fa.gz")]
    files += [os.path.join(GOLD, "refs", f) for f in os.listdir(os.path.join(GOLD, "refs")) if f.endswith(".fa.gz")]
    files = sorted(files, key=os.path.basename)
    assert len(files) == 15
    O.build_index(d, [(os.path.basename(f)[:-6], O.read_fasta(f)) for f in files], O.default_build_opt(chunks=8))
    qs = O.read_fasta(os.path.join(GOLD, "q.gene.fasta"))
    gi = la.Index(d)
    lines = gi.search_tsv([q[0] for q in qs], [q[1] for q in qs])
    gi.close()
    gold = open(os.path.join(GOLD, "q.gene.fasta.lexicmap.tsv")).read().rstrip("\n").split("\n")[1:]
    assert len(gold) == 84 and lines == gold


def test_sharded_index_union_equals_whole(small_index, queries):
    """§8e: genomes sharded over 2 'ranks' (same device here); the union of per-shard rows, re-sorted by the final
    ordering rule, equals the single-shard result (hits column recomputed)"""
    la = _la()
    d, _ = small_index
    whole = la.Index(d)
    tb = whole.info()["total_bases"]
    seqs = [q[1] for q in queries[:10]]
    rows_w, _ = whole.search(seqs)
    whole.close()
    shard_rows = []
    for r in range(2):
        si = la.Index(d, la.api.default_options(shard_rank=r, shard_count=2, total_bases_override=tb))
        rr, _ = si.search(seqs)
        shard_rows += rr
        si.close()
    key = lambda r: (r["query"], r["batch_genome"], r["cls"], r["hsp"])
    a = sorted(rows_w, key=key)
    b = sorted(shard_rows, key=key)
    assert len(a) == len(b)
    for x, y in zip(a, b):
        for f in ROW_INT + ROW_F64 + ["evalue"]:
            assert x[f] == y[f]


def test_sharded_rows_through_the_c_merge_equal_the_unsharded_output(small_index, queries):
    """§8e / boundary: the per-shard row lists of two genome shards go through lm_merge_sharded (what a Go host calls
    after its all-gatherv) and come out as the unsharded index prints them: same rows, same ORDER, same `hits`, and the
    genome / sequence names re-attached on a rank that does not hold the genome"""
    la = _la()
    from lexicmap_amd import merge
    d, _ = small_index
    whole = la.Index(d)
    tb = whole.info()["total_bases"]
    seqs = [q[1] for q in queries[:14]]
    rows_w, _ = whole.search(seqs)
    whole.close()
    shards = [la.Index(d, la.api.default_options(shard_rank=r, shard_count=2, total_bases_override=tb)) for r in range(2)]
    per_rank = []
    for si in shards:
        qb = si.upload(seqs)
        arr, _ = si.search_resident_np(qb)
        per_rank.append(arr.copy())
        si.free_batch(qb)
    merged, names = merge.merge_sharded_c(per_rank, shards[0])   # rank 0 merges: it holds only the even genomes
    ref = merge.merge_sharded(per_rank)                          # the numpy statement of the same rule
    for si in shards:
        si.close()
    assert len(merged) == len(rows_w) > 50
    for i, w in enumerate(rows_w):
        for f in ROW_INT + ROW_F64 + ["evalue", "hits", "query"]:
            assert merged[f][i] == w[f], (i, f)
        assert names[i] == (w["genome_id"], w["seq_id"]), i
    for f in ROW_INT + ROW_F64 + ["hits", "query"]:
        assert (merged[f] == ref[f]).all(), f


def test_top_n_genomes_over_shards_equals_the_unsharded_cut(small_index, queries):
    """§8e(1): -n/--top-n-genomes with a sharded index = gather of the per-shard chaining scores (lm_search_scores),
    global cut (lm_topn_merge), search with the keep list (lm_search_resident_keep), merge: the rows of the unsharded
    index searched with -n"""
    la = _la()
    from lexicmap_amd import merge
    d, _ = small_index
    N = 3
    whole = la.Index(d, la.api.default_options(top_n_genomes=N))
    tb = whole.info()["total_bases"]
    seqs = [q[1] for q in queries[:14]]
    rows_w, _ = whole.search(seqs)
    whole.close()
    shards = [la.Index(d, la.api.default_options(shard_rank=r, shard_count=2, total_bases_override=tb, top_n_genomes=N))
              for r in range(2)]
    qbs = [si.upload(seqs) for si in shards]
    cands = [si.search_scores(qb) for si, qb in zip(shards, qbs)]
    assert all(len(c[0]) > 0 for c in cands)
    kq, kg = merge.topn_merge(cands, N)
    per_rank = []
    for si, qb in zip(shards, qbs):
        rows, _ = si.search_resident_keep(qb, kq, kg)
        per_rank.append(merge.pack_rows(rows))
        si.free_batch(qb)
    merged = merge.merge_sharded_c(per_rank)
    for si in shards:
        si.close()
    assert len(merged) == len(rows_w) > 10
    for i, w in enumerate(rows_w):
        for f in ROW_INT + ROW_F64 + ["evalue", "hits", "query"]:
            assert merged[f][i] == w[f], (i, f)


def test_genome_whitelist_filter(small_index, queries):
    """a5: the `genomeIds` whitelist of (*Index).Search (lib-index-search.go:1191,1425-1489) drops the seeds of other
    genomes at anchor assembly. Every later stage works per genome, so the filtered rows must be exactly the unfiltered
    rows of the listed genomes, with `hits` = the number of listed genomes that hit"""
    la = _la()
    d, genomes = small_index
    gi = la.Index(d)
    seqs = [q[1] for q in queries[:12]]
    allrows, _ = gi.search(seqs)
    keep = sorted({r["batch_genome"] for r in allrows})[::2]      # every other genome that hit
    gi.set_genome_filter(keep)
    got, st = gi.search(seqs)
    gi.set_genome_filter(None)
    again, _ = gi.search(seqs)
    gi.close()
    assert again == allrows
    exp = [dict(r) for r in allrows if r["batch_genome"] in set(keep)]
    hits = {}
    for r in exp:
        hits.setdefault(r["query"], set()).add(r["batch_genome"])
    for r in exp:
        r["hits"] = len(hits[r["query"]])
    assert len(got) == len(exp) > 10
    assert got == exp


def _split_genomes():
    """8 genomes x 150 kb cut into 5 (even) / 3 (odd) equal contigs: with max_genome = 70000 the index writer splits them
    into 3 / 3 genome chunks (lib-index-build.go:1581-1658)"""
    from lexicmap_amd import synth
    g0 = synth.make_genomes(8, 150000, 2, seed=51, max_div=0.08, contigs=(1, 1))
    genomes = []
    for gi, (gid, contigs) in enumerate(g0):
        s = contigs[0][1]
        n = 5 if gi % 2 == 0 else 3
        L = len(s) // n
        genomes.append((gid, [("g%d_c%d" % (gi, i), s[i * L:(i + 1) * L if i < n - 1 else len(s)]) for i in range(n)]))
    return genomes


def test_split_genomes_chunk_merge(tmp_path):
    """a18 / a20: an index whose genomes were split into chunks (genomes.chunks.bin): rows = oracle (which restates the merge
    of lib-index-search.go:2798-2913), chunk columns filled, and - the property the merge exists for - everything but the
    chunk bookkeeping equals the rows of the same genomes indexed unsplit.  Sharded, the chunks of a genome stay on one
    shard: the two shards through lm_merge_sharded give the unsharded rows."""
    la = _la()
    from lexicmap_amd import merge, synth
    genomes = _split_genomes()
    qs = synth.make_gene_queries(genomes, 24, seed=52, len_range=(500, 3000), max_div=0.08)
    seqs = [q[1] for q in qs]
    d_split, d_whole = str(tmp_path / "split.lmi"), str(tmp_path / "whole.lmi")
    O.build_index(d_split, genomes, O.default_build_opt(chunks=2, max_genome=70000))
    O.build_index(d_whole, genomes, O.default_build_opt(chunks=2))
    assert os.path.getsize(os.path.join(d_split, "genomes.chunks.bin")) == 8 * (8 + 3 * 8)
    oi = O.Index(d_split, O.default_search_opt(min_qcov_genome=1.0))
    gi = la.Index(d_split, la.api.default_options(min_qcov_per_genome=1.0))
    assert gi.info()["genomes"] == 24
    rows, _ = gi.search(seqs)
    by_q = {}
    for r in rows:
        by_q.setdefault(r["query"], []).append(r)
    n = 0
    for qi, s in enumerate(seqs):
        exp, st = oi.search(s)
        got = by_q.get(qi, [])
        _cmp_rows(exp, got, qs[qi][0])
        for e, g in zip(exp, got):
            assert (e["nchunks"], e["chunk_idx"]) == (g["nchunks"], g["chunk_idx"]) and g["nchunks"] == 3
            assert g["hits"] == st["ngenomes"]
        n += len(exp)
    oi.close()
    assert n > 50
    gw = la.Index(d_whole, la.api.default_options(min_qcov_per_genome=1.0))
    rows_w, _ = gw.search(seqs)
    gw.close()
    assert len(rows_w) == len(rows)
    for a, b in zip(rows_w, rows):
        for f in a:
            if f in ("seq_idx", "nseqs", "nchunks", "chunk_idx", "batch_genome"):
                continue
            assert a[f] == b[f], f
    # two shards: every genome's chunks on one shard
    tb = gi.info()["total_bases"]
    gi.close()
    per_rank, held = [], []
    for r in range(2):
        si = la.Index(d_split, la.api.default_options(shard_rank=r, shard_count=2, total_bases_override=tb, min_qcov_per_genome=1.0))
        held.append(si.info()["genomes"])
        rr, _ = si.search(seqs)
        per_rank.append(merge.pack_rows(rr))
        si.close()
    assert sorted(held) == [12, 12]
    merged = merge.merge_sharded_c(per_rank)
    assert len(merged) == len(rows)
    for i, w in enumerate(rows):
        for f in ROW_INT + ROW_F64 + ["evalue", "hits", "query", "nchunks", "chunk_idx"]:
            assert merged[f][i] == w[f], (i, f)


def test_concurrent_calls_serialise(both, queries):
    """the reference calls Search from up to --max-query-conc goroutines (search.go:549,577-602); the handle owns all device
    scratch, so concurrent calls on one lm_index* are serialised by its mutex: two threads searching different batches at
    the same time both get exactly their single-threaded rows"""
    import threading
    _oi, gi = both
    a = [q[1] for q in queries[:12]]
    b = [q[1] for q in queries[12:24]]
    want_a, _ = gi.search(a)
    want_b, _ = gi.search(b)
    got = {}

    def run(name, seqs, reps):
        out = []
        for _ in range(reps):
            rows, _st = gi.search(seqs)   # ctypes releases the GIL during the call
            out.append(rows)
        got[name] = out

    ta = threading.Thread(target=run, args=("a", a, 4))
    tb = threading.Thread(target=run, args=("b", b, 4))
    ta.start()
    tb.start()
    ta.join()
    tb.join()
    assert all(r == want_a for r in got["a"]) and all(r == want_b for r in got["b"])
    assert len(want_a) > 10 and len(want_b) > 10


def test_errors_fail_loudly(tmp_path):
    la = _la()
    with pytest.raises(RuntimeError):
        la.Index(str(tmp_path / "does_not_exist"))


def test_index_format_variants_and_broken_files(small_index, queries, tmp_path):
    """a20: (1) masks.bin as a headered big-endian mask list (8-byte magic, meta bytes with k at byte 10, u64 count, u64 seed:
    the layout lexichash's WriteToFile is believed to use) opens and searches like this build's own layout - accepted only
    because k / count agree with info.toml and the masks ascend; (2) an unknown layout, a truncated seed chunk and a
    genomes.chunks.bin that this build cannot honour yet fail with an error instead of reading out of bounds / wrong rows"""
    import shutil
    import struct
    la = _la()
    d, _ = small_index
    base = la.Index(d)
    seqs = [q[1] for q in queries[:6]]
    want, _ = base.search(seqs)
    M = base.info()["masks"]
    mp = la.lib().lm_index_masks(base.h)
    masks = [mp[i] for i in range(M)]
    base.close()
    alt = str(tmp_path / "alt.lmi")
    shutil.copytree(d, alt)
    with open(os.path.join(alt, "masks.bin"), "wb") as f:
        f.write(b"kmermask" + bytes([0, 1, 31, 0, 0, 0, 0, 0]) + struct.pack(">QQ", M, 1))
        f.write(b"".join(struct.pack(">Q", m) for m in masks))
    gi = la.Index(alt)
    got, _ = gi.search(seqs)
    gi.close()
    assert got == want and len(want) > 0
    # unknown masks layout
    bad = str(tmp_path / "bad1.lmi")
    shutil.copytree(d, bad)
    with open(os.path.join(bad, "masks.bin"), "wb") as f:
        f.write(b"whatever" + bytes(40))
    with pytest.raises(RuntimeError, match="masks"):
        la.Index(bad)
    # truncated seed chunk: cut inside the records
    bad2 = str(tmp_path / "bad2.lmi")
    shutil.copytree(d, bad2)
    chunk = sorted(f for f in os.listdir(os.path.join(bad2, "seeds")) if f.endswith(".bin"))[0]
    path = os.path.join(bad2, "seeds", chunk)
    data = open(path, "rb").read()
    for cut in (len(data) // 2, len(data) - 3, 41):
        open(path, "wb").write(data[:cut])
        with pytest.raises(RuntimeError, match="broken|invalid"):
            la.Index(bad2)
    # an index of format 3.4 or older: the reference masks its queries with MaskKnownDistinctPrefixesWithStrandBias
    # (lib-index-search.go:1212-1215), which is not restated here: refused by the library AND by the oracle, never searched
    # with the 3.5 masking
    import re
    for repl in ("minor-version = 4", ""):
        old = str(tmp_path / ("old%d.lmi" % len(repl)))
        shutil.copytree(d, old)
        t = open(os.path.join(old, "info.toml")).read()
        assert "minor-version = 5" in t
        open(os.path.join(old, "info.toml"), "w").write(re.sub(r"minor-version = 5\n", repl + "\n" if repl else "", t))
        with pytest.raises(RuntimeError, match="1212-1215"):
            la.Index(old)
        with pytest.raises(Exception):
            O.Index(old)


def test_prophage_golden_rows_through_the_hip_stages(small_index):
    """All 9 rows of the reference's demo/q.prophage.fasta.lexicmap.tsv through the HIP stage exports: the windows the
    reference's seed chains opened (tests/test_oracle_reference_vectors.py pins them with the oracle) go through
    lm_pseudoalign_batch (SeqComparator.Compare + Chainer2 on the device) and lm_wfa_batch (the LDS wavefront kernels, 64 bp to
    9.4 kb), extendMatch between them by the oracle (no stage export), and the rows assembled from the device results -
    coordinates, alignment length, identity, gaps, e-value, bitscore - are the golden's to the digit."""
    la = _la()
    L = O.lib()
    d, _ = small_index
    gi = la.Index(d)                      # (any handle: the stage exports need k and the device only)
    q = O.read_fasta(os.path.join(GOLD, "q.prophage.fasta"))[0][1]
    gold = [r.split("\t") for r in open(os.path.join(GOLD, "q.prophage.fasta.lexicmap.tsv")).read().rstrip("\n").split("\n")[1:]]
    as_row = lambda g: (int(g[12]), int(g[13]), int(g[14]), int(g[15]), int(g[9]), g[10], int(g[11]), g[18], int(g[19]))
    g1 = O.read_fasta(os.path.join(GOLD, "GCF_003697165.2.fa.gz"))[0][1]
    g2 = O.read_fasta(os.path.join(GOLD, "GCF_002949675.1.fa.gz"))[0][1]
    g3 = [O.read_fasta(os.path.join(GOLD, "refs", f)) for f in os.listdir(os.path.join(GOLD, "refs")) if f.startswith("GCF_002950215.1")][0][0][1]
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    # (genome, rc, query window, target window (0-based, inclusive), the golden rows expected from it in order)
    wins = [(g1, False, (0, 10407), (1864410 - 1000, 1873945), [0, 1]),
            (g1, False, (17440 - 1000, 24381 + 1000), (1882010 - 1000, 1888947 + 1000), [2]),
            (g1, False, (24354 - 1000, 30294 + 1000), (1853097 - 1000, 1859037 + 1000), [3]),
            (g1, False, (11000 - 1000, 12289 + 1000), (1873845 + (11000 - 10307) - 1000, 1876827), [4]),   # the window-clipped row
            (g1, False, (14539 - 1000, 15357 + 1000), (1878797 - 1000, 1879616 + 1000), [5]),
            (g2, True, (13918 - 1000, 14245 + 1000), (3704318 - 1000, 3704648 + 1000), [6]),
            (g3, False, (14836 - 1000, 14897 + 1000), (71091 - 1000, 71152 + 1000), [7]),
            (g3, False, (14836 - 1000, 14897 + 1000), (4261070 - 1000, 4261131 + 1000), [8])]
    so = O.default_search_opt()
    total_bases = 54142446
    problems, targets = [], []
    for g, rc, (qb, qe), (tb, te), _rows in wins:
        w = g[tb:te + 1]
        if rc:
            w = w.translate(comp)[::-1]
        targets.append(w)
        problems.append((0, qb, min(qe, len(q) - 1), w))
    chains = gi.pseudoalign([q], problems)
    pairs, meta = [], []
    for (g, rc, (qb, qe), (tb, te), rows), w, ch in zip(wins, targets, chains):
        assert len(ch) == len(rows), (rows, ch)
        for c in ch:
            ctb, cte = (te - c["tend"], te - c["tbegin"]) if rc else (tb + c["tbegin"], tb + c["tend"])
            o = [C.c_int() for _ in range(8)]
            L.lmo_extend_match(q, len(q), w, len(w), c["qbegin"], c["qend"] + 1, c["tbegin"], c["tend"] + 1, so.ext_len2, ctb,
                               len(g) - 1 - cte, 1 if rc else 0, *[C.byref(x) for x in o])
            qs, qe2, ts, te2, s1, e1, s2, e2 = [x.value for x in o]
            pairs.append((q[qs:qe2], w[ts:te2]))
            meta.append((c, rc, ctb, cte, s1, e1, s2, e2, qe2 - qs, te2 - ts))
    res = gi.wfa(pairs)
    gi.close()
    got = []
    for r, (c, rc, ctb, cte, s1, e1, s2, e2, lq, lt) in zip(res, meta):
        wr = O.WfaResult()     # lmo_score_evalue reads the run list: hand it the device's
        wr.score, wr.nops = r["score"], len(r["ops"])
        arr = (C.c_uint64 * max(1, len(r["ops"])))(*r["ops"])
        wr.ops = C.cast(arr, C.POINTER(C.c_uint64))
        score, bits, ev = C.c_int(), C.c_int(), C.c_double()
        L.lmo_score_evalue(C.byref(wr), lq, total_bases, C.byref(score), C.byref(bits), C.byref(ev))
        if rc:      # lib-index-search.go:2541-2556, the - strand form
            t1, t2 = ctb - e2 + (lt - r["tend"]) + 1, cte + s2 - (r["tbegin"] - 1) + 1
        else:
            t1, t2 = ctb - s2 + r["tbegin"], cte + e2 - (lt - r["tend"]) + 1
        got.append((c["qbegin"] - s1 + r["qbegin"], c["qend"] + e1 - (lq - r["qend"]) + 1, t1, t2, r["align_len"],
                    "%.3f" % (100.0 * r["matches"] / r["align_len"]), r["gaps"], "%.2e" % ev.value, bits.value))
    want = [as_row(gold[i]) for _g, _rc, _q, _t, rows in wins for i in rows]
    assert got == want


def test_a_saved_index_keeps_its_genome_chunk_lists(tmp_path):
    """lm_index_save writes genomes.chunks.bin (lib-index-build.go:1787-1808: per split genome the number of its chunks and their
    keys in chunk order): the file of an index with split genomes comes back byte for byte, the saved index gives the rows of
    the original through the HIP path and through the oracle; an index without split genomes gets an empty file"""
    la = _la()
    from lexicmap_amd import synth
    genomes = _split_genomes()
    qs = synth.make_gene_queries(genomes, 12, seed=53, len_range=(500, 3000), max_div=0.08)
    ids, seqs = [q[0] for q in qs], [q[1] for q in qs]
    d0, d1, d2 = str(tmp_path / "split.lmi"), str(tmp_path / "saved.lmi"), str(tmp_path / "plain_saved.lmi")
    O.build_index(d0, genomes, O.default_build_opt(chunks=2, max_genome=70000))
    gi = la.Index(d0)
    want = gi.search_tsv(ids, seqs)
    gi.save(d1, chunks=3)
    gi.close()
    assert open(os.path.join(d1, "genomes.chunks.bin"), "rb").read() == open(os.path.join(d0, "genomes.chunks.bin"), "rb").read()
    g1, o1 = la.Index(d1), O.Index(d1)
    got = g1.search_tsv(ids, seqs)
    exp = []
    for i, s in zip(ids, seqs):
        exp += o1.search_tsv(i, s)
    g1.close()
    o1.close()
    assert len(want) > 0 and got == want == exp
    d3 = str(tmp_path / "plain.lmi")
    O.build_index(d3, genomes[:3], O.default_build_opt(chunks=2))
    g3 = la.Index(d3)
    g3.save(d2, chunks=2)
    g3.close()
    assert os.path.getsize(os.path.join(d2, "genomes.chunks.bin")) == 0


def test_an_allocation_failure_inside_the_loader_starts_the_seed_passes_over_without_kept_copies(tmp_path, monkeypatch):
    """the loader keeps the decoded seeds of a chunk file on the device between its count and place passes while an ESTIMATE of
    what the packed image will need leaves room; when an allocation fails all the same, the passes start over without the kept
    copies (every file decoded twice) instead of failing the open (LM_DEBUG_LOADER_OOM: the failure injected once, at the end of
    the first attempt).  Same image, same rows."""
    la = _la()
    from lexicmap_amd import synth
    genomes = synth.make_genomes(5, 70_000, 2, seed=61, max_div=0.05)
    qs = synth.make_gene_queries(genomes, 8, seed=62, len_range=(500, 2000), max_div=0.06)
    ids, seqs = [q[0] for q in qs], [q[1] for q in qs]
    d = str(tmp_path / "oom.lmi")
    O.build_index(d, genomes, O.default_build_opt(chunks=3))
    gi = la.Index(d)
    want, info0 = gi.search_tsv(ids, seqs), gi.info()
    seeds0 = [gi.mask_seeds(m) for m in (0, 7, 99)]
    gi.close()
    monkeypatch.setenv("LM_DEBUG_LOADER_OOM", "1")
    gi = la.Index(d)
    monkeypatch.delenv("LM_DEBUG_LOADER_OOM")
    try:
        assert gi.info() == info0
        for m, s0 in zip((0, 7, 99), seeds0):
            s1 = gi.mask_seeds(m)
            assert all(np.array_equal(a, b) for a, b in zip(s0, s1))
        assert len(want) > 0 and gi.search_tsv(ids, seqs) == want
    finally:
        gi.close()
