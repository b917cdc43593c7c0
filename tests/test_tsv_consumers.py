"""the search TSV as its consumers (`utils 2blast`, `2sam`, `merge-search-results`) need it.
CPU: the reference's own -a golden and the oracle's rows for gapped alignments; GPU: the HIP path's rows."""
import os
import random

import numpy as np
import pytest

import oracle as O
import tsvcheck

HERE = os.path.dirname(os.path.abspath(__file__))


def test_reference_golden_all_columns_satisfies_the_consumer_contract():
    lines = open(os.path.join(HERE, "golden", "demo", "q.gene.fasta.lexicmap_top-2-genomes_all.tsv")).read().splitlines()
    assert lines[0].split("\t") == tsvcheck.COLUMNS          # header of search.go:440-446
    for l in lines[1:]:
        r = tsvcheck.check_line(l)
        assert r["M"] > 0
    lines = open(os.path.join(HERE, "golden", "demo", "q.gene.fasta.lexicmap.tsv")).read().splitlines()
    assert lines[0].split("\t") == tsvcheck.COLUMNS[:20]
    for l in lines[1:]:
        tsvcheck.check_line(l, all_columns=False)


@pytest.fixture(scope="module")
def gapped_case(tmp_path_factory):
    from lexicmap_amd import synth
    d = str(tmp_path_factory.mktemp("tsv") / "g.lmi")
    genomes = synth.make_genomes(6, 60000, 2, seed=31, max_div=0.08, contigs=(1, 2))
    O.build_index(d, genomes, O.default_build_opt(chunks=2))
    rng = np.random.default_rng(9)
    qs = []
    for i in range(10):
        g = genomes[i % len(genomes)]
        seq = np.frombuffer(g[1][0][1], dtype=np.uint8)
        st = int(rng.integers(0, max(1, len(seq) - 2500)))
        q = synth.mutate(rng, seq[st:st + int(rng.integers(600, 2400))], sub=0.04, ins=0.012, dele=0.012)
        if i % 2:
            q = np.frombuffer(q.tobytes().translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1], dtype=np.uint8)
        qs.append(("q%d" % i, q.tobytes()))
    return d, qs


def test_oracle_rows_with_gaps_satisfy_the_consumer_contract(gapped_case):
    d, qs = gapped_case
    oi = O.Index(d, O.default_search_opt(output_seq=1))
    n = gaps = 0
    for qid, s in qs:
        for l in oi.search_tsv(qid, s, more_columns=True):
            r = tsvcheck.check_line(l)
            n += 1
            gaps += r["I"] + r["D"]
    oi.close()
    assert n >= 10 and gaps > 20        # the case really exercises I / D runs


@pytest.mark.gpu
def test_hip_rows_satisfy_the_consumer_contract_and_equal_the_oracle(gapped_case):
    import lexicmap_amd as la
    d, qs = gapped_case
    oi = O.Index(d, O.default_search_opt(output_seq=1))
    gi = la.Index(d, la.api.default_options(output_seq=1))
    got = gi.search_tsv([q[0] for q in qs], [q[1] for q in qs], more_columns=True)
    exp = []
    for qid, s in qs:
        exp += oi.search_tsv(qid, s, more_columns=True)
    assert len(got) == len(exp) >= 10
    for g, e in zip(got, exp):
        tsvcheck.check_line(g)
        ge, ee = g.split("\t"), e.split("\t")
        assert ge[:18] == ee[:18] and ge[19:] == ee[19:]     # everything but the e-value text ...
        assert abs(float(ge[18]) - float(ee[18])) <= 1e-9 * max(abs(float(ee[18])), 1e-300) + 1e-320  # ... within 1e-9
    gi.close()
    oi.close()
