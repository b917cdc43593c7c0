"""GPU parity at the last step of the two length ladders of the reference: a target window >= 1 Mb - minimum pseudo-alignment
prefix 11 + 8 = 19 (lib-seq_compare.go:339-348) - and an HSP above 1 Mb - extendMatch flank 50 + 80 (lib-index-search.go:
2245-2254: beyond the 128-column grid of the flank chainer, so its plain pair list runs) - with a WFA problem of 1.15 Mb
through the windowed LDS kernels.  A 1.15-Mb query cut from one genome of a two-member family, HIP path vs oracle, row for row."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


def test_a_megabase_query_aligns_like_the_oracle(tmp_path):
    import lexicmap_amd as la
    from lexicmap_amd import synth
    d = str(tmp_path / "mb.lmi")
    genomes = synth.make_genomes(2, 1_600_000, 1, seed=91, max_div=0.02, contigs=(1, 1))
    O.build_index(d, genomes, O.default_build_opt(chunks=4))
    rng = np.random.default_rng(92)
    s = np.frombuffer(genomes[1][1][0][1], dtype=np.uint8)
    q = synth.mutate(rng, s[200_000:1_350_000], sub=0.003, ins=0.0005, dele=0.0005).tobytes()
    short = synth.mutate(rng, s[50_000:52_000], sub=0.02).tobytes()       # an ordinary query in the same batch
    oi = O.Index(d)
    gi = la.Index(d)
    rows, stats = gi.search([q, short])
    by_q = {}
    for r in rows:
        by_q.setdefault(r["query"], []).append(r)
    n = 0
    for qi, seq in enumerate([q, short]):
        exp, st = oi.search(seq)
        got = by_q.get(qi, [])
        assert len(exp) == len(got) and len(exp) >= 2
        for e, g in zip(exp, got):
            for f in ("batch_genome", "cls", "hsp", "seq_idx", "nseqs", "seq_len", "rc", "aligned_length", "gaps", "qbegin", "qend",
                      "tbegin", "tend", "bitscore", "score", "matched_bases", "qcov_genome", "qcov_hsp", "pident"):
                assert e[f] == g[f], (qi, f, e[f], g[f])
            assert abs(e["evalue"] - g["evalue"]) <= 1e-9 * max(abs(e["evalue"]), 1e-300)
            assert g["hits"] == st["ngenomes"]
            n += 1
    gi.close()
    oi.close()
    long_rows = by_q[0]
    assert all(r["aligned_length"] > 1_000_000 and r["tend"] - r["tbegin"] + 1 > 1_000_000 for r in long_rows)
    assert stats["rows"] == n
