"""GPU edge cases of the batch interface (through the C-ABI): empty batch, empty / too short / all-N queries inside a batch
(the reference's reader loop counts queries shorter than k and prints nothing for them, search.go:571-575), a query that is a
whole contig, duplicated queries; the rows of the ordinary queries of such a batch equal the oracle's and keep their batch
positions."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

FIELDS = ("batch_genome", "cls", "hsp", "seq_idx", "rc", "aligned_length", "gaps", "qbegin", "qend", "tbegin", "tend",
          "bitscore", "pident", "qcov_hsp", "qcov_genome")


@pytest.fixture(scope="module")
def idx(tmp_path_factory):
    import lexicmap_amd as la
    from lexicmap_amd import synth
    d = str(tmp_path_factory.mktemp("edge") / "edge.lmi")
    genomes = synth.make_genomes(6, 90_000, 2, seed=91, max_div=0.06, contigs=(2, 3))
    O.build_index(d, genomes, O.default_build_opt(chunks=2))
    gi, oi = la.Index(d), O.Index(d)
    yield gi, oi, genomes
    gi.close()
    oi.close()


def test_empty_batch_and_degenerate_queries(idx):
    gi, oi, genomes = idx
    rows, st = gi.search([])
    assert rows == [] and st["rows"] == 0
    for s in (b"", b"A", b"ACGT" * 7, b"N" * 500, b"ACGTN" * 6):   # 0, 1, 28, 500 (all N) and exactly 30 bases: all < k or nothing to seed
        rows, st = gi.search([s])
        assert rows == [], s[:10]
    rows, _ = gi.search([b"", b"", b"AC"])
    assert rows == []


def test_ordinary_queries_keep_their_rows_and_positions_among_degenerate_ones(idx):
    gi, oi, genomes = idx
    g = genomes[1][1][0][1]
    h = len(g) // 2
    assert len(g) > 4000
    good = [g[1000:2200], g[h:h + 900][::-1].translate(bytes.maketrans(b"ACGT", b"TGCA")), genomes[4][1][1][1][500:2500]]
    batch = [b"", good[0], b"ACGTACGT", good[1], b"N" * 100, good[2], good[0], b"A" * 31]
    rows, st = gi.search(batch)
    by_q = {}
    for r in rows:
        by_q.setdefault(r["query"], []).append(r)
    assert set(by_q) == {1, 3, 5, 6}
    for qi in (1, 3, 5, 6):
        exp, stq = oi.search(batch[qi])
        assert len(exp) == len(by_q[qi]) > 0
        for e, r in zip(exp, by_q[qi]):
            for f in FIELDS:
                assert e[f] == r[f], (qi, f)
            assert r["hits"] == stq["ngenomes"]
    # the duplicated query gets the same rows twice
    strip = lambda rs: [{k: v for k, v in r.items() if k != "query"} for r in rs]
    assert strip(by_q[1]) == strip(by_q[6])


def test_whole_contig_as_query(idx):
    gi, oi, genomes = idx
    contig = max((c for _, cs in genomes for c in cs), key=lambda c: len(c[1]))[1]
    assert len(contig) > 30_000
    rows, _ = gi.search([contig])
    exp, st = oi.search(contig)
    assert len(rows) == len(exp) > 0
    for e, r in zip(exp, rows):
        for f in FIELDS:
            assert e[f] == r[f], f
    best = rows[0]
    assert best["pident"] == 100.0 and best["aligned_length"] == len(contig) and best["qcov_hsp"] == 100.0


@pytest.mark.parametrize("k", [21, 27])
def test_an_index_with_k_below_31_gives_the_oracle_rows_for_windows_beyond_2048_positions(tmp_path, k):
    """k is index-defined (info.toml max-K; SURVEY section 8: 'always read it from the index').  The anchor filter's rolling form is
    written for K == 31 and slices windows into pieces of up to 2048 positions; any other k must run the strided form with its
    1920-position slices (lm_kernels.hip launch_pa_filter) - reads of 4-6 kb give chain windows well beyond 2048 positions"""
    import lexicmap_amd as la
    from lexicmap_amd import synth
    genomes = synth.make_genomes(5, 80_000, 1, seed=13, max_div=0.05)
    queries = synth.make_gene_queries(genomes, 6, seed=14, len_range=(4000, 6000), max_div=0.06)
    d = str(tmp_path / ("k%d.lmi" % k))
    O.build_index(d, genomes, O.default_build_opt(chunks=2, k=k))
    gi, oi = la.Index(d), O.Index(d)
    try:
        assert gi.info()["k"] == k
        ids, seqs = [q[0] for q in queries], [q[1] for q in queries]
        got = gi.search_tsv(ids, seqs)
        exp = []
        for i, s in zip(ids, seqs):
            exp += oi.search_tsv(i, s)
        assert len(exp) > 10 and max(int(r.split("\t")[9]) for r in exp) > 2048   # alignments beyond 2048 bases
        assert got == exp
    finally:
        gi.close()
        oi.close()
