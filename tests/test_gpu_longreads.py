"""GPU parity on the shapes of BASELINE configs 3 and 4: ONT-style long reads (5-50 kb, sub 2 % / ins 2 % / del 3 %) and a
circular 120-kb query, HIP path (C-ABI) vs the CPU oracle, row for row.

What these inputs reach that the gene-sized tests do not: target windows >= 10 kb and >= 50 kb (minimum pseudo-alignment
prefix 13 / 15, lib-seq_compare.go:339-348), the extendMatch extension steps for HSPs above 10 kb and 50 kb
(lib-index-search.go:2245-2254), WFA problems of tens of kb (the global-memory fallback of the wavefront kernel), chains
of hundreds of seeds per (query, genome), and the internal split of a batch into parts.
"""
import os

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


def _la():
    import lexicmap_amd as la
    return la


@pytest.fixture(scope="module")
def lr_index(tmp_path_factory):
    """8 genomes x ~400 kb in 2 families (<= 6 % divergence), 1-2 contigs"""
    from lexicmap_amd import synth
    d = str(tmp_path_factory.mktemp("lridx") / "lr.lmi")
    genomes = synth.make_genomes(8, 400_000, 2, seed=21, max_div=0.06, contigs=(1, 2))
    O.build_index(d, genomes, O.default_build_opt(chunks=4))
    return d, genomes


@pytest.fixture(scope="module")
def lr_queries(lr_index):
    from lexicmap_amd import synth
    _, genomes = lr_index
    qs = synth.make_reads(genomes, 5, seed=31, len_range=(5000, 50000))
    # force the extremes of the C3 range: one read at the 50-kb end, one at 5 kb
    rng = np.random.default_rng(32)
    gid, contigs = genomes[3]
    s = np.frombuffer(max(contigs, key=lambda c: len(c[1]))[1], dtype=np.uint8)
    qs.append(("r50k", synth.mutate(rng, s[1000:51000], sub=0.02, ins=0.02, dele=0.03).tobytes()))
    qs.append(("r5k", synth.mutate(rng, s[60000:65000], sub=0.02, ins=0.02, dele=0.03).tobytes()))
    # C4 shape: a circular 120-kb query = a rotated copy of a genome region (lightly mutated), reverse strand
    region = s[100000:220000]
    rot = np.concatenate([region[70000:], region[:70000]])
    rot = synth.mutate(rng, rot, sub=0.01, ins=0.002, dele=0.002)
    rc = rot.tobytes().translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1]
    qs.append(("circ120k_rc", rc))
    return qs


def _cmp(exp, got, name):
    assert len(exp) == len(got), (name, len(exp), len(got))
    for e, g in zip(exp, got):
        for f in ("batch_genome", "cls", "hsp", "seq_idx", "nseqs", "seq_len", "rc", "aligned_length", "gaps", "qbegin", "qend",
                  "tbegin", "tend", "bitscore", "score", "matched_bases"):
            assert e[f] == g[f], (name, f, e[f], g[f], e, g)
        for f in ("qcov_genome", "qcov_hsp", "pident"):
            assert e[f] == g[f], (name, f, e[f], g[f])
        assert abs(e["evalue"] - g["evalue"]) <= 1e-9 * max(abs(e["evalue"]), 1e-300), (name, e["evalue"], g["evalue"])


def test_long_read_rows_equal_oracle(lr_index, lr_queries):
    la = _la()
    d, _ = lr_index
    oi = O.Index(d)
    gi = la.Index(d)
    seqs = [q[1] for q in lr_queries]
    rows, stats = gi.search(seqs)
    by_q = {}
    for r in rows:
        by_q.setdefault(r["query"], []).append(r)
    nrows, longest, widest = 0, 0, 0
    for qi, s in enumerate(seqs):
        exp, st = oi.search(s)
        got = by_q.get(qi, [])
        _cmp(exp, got, lr_queries[qi][0])
        for g in got:
            assert g["hits"] == st["ngenomes"]
            longest = max(longest, g["aligned_length"])
            widest = max(widest, g["tend"] - g["tbegin"] + 1)
        nrows += len(exp)
    gi.close()
    oi.close()
    assert nrows >= 20
    assert longest > 40000 and widest > 40000   # HSPs beyond the 10-kb extension step, windows beyond 50 kb
    assert stats["rows"] == nrows


def test_batch_split_into_parts_gives_the_same_rows(lr_index, lr_queries, monkeypatch):
    """a batch above the per-pass limits is searched as consecutive parts (lm_qbatch_upload): same rows, same order"""
    la = _la()
    d, _ = lr_index
    gi = la.Index(d)
    seqs = [q[1] for q in lr_queries[:6]]
    base, st0 = gi.search(seqs)
    monkeypatch.setenv("LM_MAX_PART_KMERS", "60000")   # ~one long read per part
    got, st1 = gi.search(seqs)
    monkeypatch.delenv("LM_MAX_PART_KMERS")
    assert len(base) == len(got) and len(base) > 10
    for b, g in zip(base, got):
        assert b == g
    assert st0["rows"] == st1["rows"] and st0["chains"] == st1["chains"]
    # a part whose seed anchors outgrow the scratch budget is halved on the fly (forced here): same rows again
    monkeypatch.setenv("LM_DEBUG_MAX_ANCHORS", "1500")
    got2, st2 = gi.search(seqs)
    monkeypatch.delenv("LM_DEBUG_MAX_ANCHORS")
    assert len(base) == len(got2)
    for b, g in zip(base, got2):
        assert b == g
    assert st0["rows"] == st2["rows"]
    gi.close()
    # two lanes (the default for a batch of several parts: the parts searched side by side by two host threads, own scratch
    # and streams, half of the scratch budget each, parts halved on the fly inside the lanes) were what ran above; a handle
    # opened with LM_TWO_LANES=0 searches the parts one after the other: the same rows in the same order
    monkeypatch.setenv("LM_TWO_LANES", "0")
    gi = la.Index(d)
    monkeypatch.delenv("LM_TWO_LANES")
    monkeypatch.setenv("LM_MAX_PART_KMERS", "60000")
    got3, st3 = gi.search(seqs)
    monkeypatch.setenv("LM_DEBUG_MAX_ANCHORS", "1500")
    got4, _ = gi.search(seqs)
    monkeypatch.delenv("LM_DEBUG_MAX_ANCHORS")
    monkeypatch.delenv("LM_MAX_PART_KMERS")
    gi.close()
    assert got3 == base and got4 == base
    assert st0["rows"] == st3["rows"] and st0["chains"] == st3["chains"]


def test_rows_do_not_depend_on_which_device_implementation_of_a_stage_runs(lr_index, lr_queries, monkeypatch):
    """LM_WFA_R16 (16- or 32-bit ring cells in the short WFA classes), LM_LOOKUP_FLAT
    (seed anchors emitted with the lanes over the output or over the lookups), LM_PA_FILTER_ROLL (window positions consecutive
    per lane or strided) and LM_ARENA_RESERVE_PCT (lane slabs or slabs on demand) choose between device paths that must agree
    to the byte: the rows of the long-read fixture (equal to the oracle's by the first test) with each switch flipped"""
    la = _la()
    d, _ = lr_index
    seqs = [q[1] for q in lr_queries]
    gi = la.Index(d)
    gi.profile(True)
    base, st0 = gi.search(seqs)
    ran = {p["name"] for p in gi.profile_get() if p["launches"] > 0}
    gi.close()
    assert any(n in ("k_wfa_lean512", "k_wfa_win512", "k_wfa_lean1024", "k_wfa_win1024") for n in ran), ran   # the fixture does reach the wide passes
    for var, off in (("LM_WFA_R16", "0"), ("LM_SCRATCH_BUDGET_MB", "3072"),
                     ("LM_LOOKUP_FLAT", "0"), ("LM_PA_FILTER_ROLL", "0"), ("LM_ARENA_RESERVE_PCT", "0")):
        monkeypatch.setenv(var, off)
        gi = la.Index(d)      # the switches are read once per handle
        gi.profile(True)
        got, st1 = gi.search(seqs)
        ran1 = {p["name"] for p in gi.profile_get() if p["launches"] > 0}
        gi.close()
        monkeypatch.delenv(var)
        assert got == base, var
        assert st0["rows"] == st1["rows"] and st0["chains"] == st1["chains"] and st0["pa_anchors"] == st1["pa_anchors"]


def test_many_small_rounds_give_the_rows_of_one_round(lr_index, lr_queries, monkeypatch):
    """With the alignment half in chunks (forced here: several chunks, a round of extendMatch / WFA / finalisation every few
    HSPs) the rows, CIGAR / alignment strings included, must be those of the single-round run."""
    la = _la()
    d, _ = lr_index
    seqs = [q[1] for q in lr_queries]
    opt = dict(output_seq=1)
    gi = la.Index(d, la.api.default_options(**opt))
    base, st0 = gi.search(seqs)
    gi.close()
    assert len(base) >= 20 and max(r["aligned_length"] for r in base) > 40000
    monkeypatch.setenv("LM_DEBUG_MAX_WINDOW_BYTES", "150000")   # a few chain windows per chunk
    monkeypatch.setenv("LM_DEBUG_ROUND_HSPS", "4")              # a round every few HSPs
    monkeypatch.setenv("LM_DEBUG_MIN_ROUND_HSPS", "1")
    gi = la.Index(d, la.api.default_options(**opt))
    got, st1 = gi.search(seqs)
    gi.close()
    assert got == base
    assert st0["rows"] == st1["rows"] and st0["hsps_aligned"] == st1["hsps_aligned"]
    for v in ("LM_DEBUG_MAX_WINDOW_BYTES", "LM_DEBUG_ROUND_HSPS", "LM_DEBUG_MIN_ROUND_HSPS"):
        monkeypatch.delenv(v)
