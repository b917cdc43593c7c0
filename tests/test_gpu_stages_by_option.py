"""GPU parity tests that isolate the stages which have no stage-level export of their own - extendMatch (a14), contig
resolution / coordinate conversion / dedup (a13) and the HSP statistics + filters (a17) - by driving them through the
options and inputs that only they react to, HIP path (C-ABI) vs oracle, row for row.

  a14  extendMatch flank length ext_len2 (lib-index-search.go:2245-2254, hard-coded 50 in search.go:325): 0 (stage off),
       10, 50, 130 - the rows must equal the oracle's for every value AND differ between values (the stage matters)
  a13  queries laid across contig boundaries of multi-contig genomes (the contigs are joined by 1000 x 'A',
       lib-index-build.go:924): HSPs clipped at the spacer, clusters split per contig, (qb, qe, tb, te, seq, rc) dedup
       (lib-index-search.go:2083-2468), incl. a query that contains the spacer itself and reverse-strand versions
  a17  every HSP filter on its own (lib-index-search.go:2274-2312,2541-2581): --max-evalue, --min-qcov-per-hsp,
       --align-min-match-pident, --min-qcov-per-genome
"""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

COMP = bytes.maketrans(b"ACGT", b"TGCA")
FIELDS = ("batch_genome", "cls", "hsp", "seq_idx", "nseqs", "seq_len", "rc", "aligned_length", "gaps", "qbegin", "qend",
          "tbegin", "tend", "bitscore", "score", "matched_bases", "qcov_genome", "qcov_hsp", "pident")


def _la():
    import lexicmap_amd as la
    return la


@pytest.fixture(scope="module")
def mc_index(tmp_path_factory):
    """10 genomes x ~150 kb, 2 families, 3-5 contigs each"""
    from lexicmap_amd import synth
    d = str(tmp_path_factory.mktemp("mcidx") / "mc.lmi")
    genomes = synth.make_genomes(10, 150_000, 2, seed=81, max_div=0.08, contigs=(3, 5))
    O.build_index(d, genomes, O.default_build_opt(chunks=3))
    return d, genomes


def _equal(oi, gi, seqs, names):
    rows, _ = gi.search(seqs)
    by_q = {}
    for r in rows:
        by_q.setdefault(r["query"], []).append(r)
    out = []
    for qi, s in enumerate(seqs):
        exp, st = oi.search(s)
        got = by_q.get(qi, [])
        assert len(exp) == len(got), (names[qi], len(exp), len(got))
        for e, g in zip(exp, got):
            for f in FIELDS:
                assert e[f] == g[f], (names[qi], f, e[f], g[f])
            assert abs(e["evalue"] - g["evalue"]) <= 1e-9 * max(abs(e["evalue"]), 1e-300)
            assert g["hits"] == st["ngenomes"]
        out.append(got)
    return out


def _queries(genomes):
    from lexicmap_amd import synth
    rng = np.random.default_rng(82)
    qs = synth.make_gene_queries(genomes, 10, seed=83, len_range=(400, 1800), max_div=0.10)
    qs += synth.make_reads(genomes, 2, seed=84, len_range=(3000, 9000))
    return qs, rng


def test_extend_match_flank_lengths(mc_index):
    la = _la()
    d, genomes = mc_index
    qs, _ = _queries(genomes)
    seqs, names = [q[1] for q in qs], [q[0] for q in qs]
    spans = {}
    for ext2 in (0, 10, 50, 130):
        oi = O.Index(d, O.default_search_opt(ext_len2=ext2))
        gi = la.Index(d, la.api.default_options(ext_len2=ext2))
        got = _equal(oi, gi, seqs, names)
        spans[ext2] = [(r["qbegin"], r["qend"], r["tbegin"], r["tend"], r["aligned_length"]) for rows in got for r in rows]
        gi.close()
        oi.close()
    assert len(spans[50]) > 20
    # the stage does something: without it HSPs end where the pseudo-alignment chain ends, with it most reach further
    assert spans[0] != spans[50] and spans[10] != spans[130]
    assert sum(a[4] for a in spans[50]) > sum(a[4] for a in spans[0])


def test_queries_across_contig_boundaries(mc_index):
    la = _la()
    d, genomes = mc_index
    rng = np.random.default_rng(85)
    from lexicmap_amd import synth
    seqs, names = [], []
    for gi_, (gid, contigs) in enumerate(genomes[:6]):
        for c in range(len(contigs) - 1):
            a, b = contigs[c][1], contigs[c + 1][1]
            if len(a) < 1200 or len(b) < 1200:
                continue
            left = np.frombuffer(a[-700:], dtype=np.uint8)
            right = np.frombuffer(b[:650], dtype=np.uint8)
            joined = np.concatenate([left, right])                       # the two contig ends back to back
            spacer = np.concatenate([left, np.full(1000, ord("A"), np.uint8), right])  # ... with the index's own spacer
            for nm, s in (("join", joined), ("spacer", spacer)):
                m = synth.mutate(rng, s, sub=0.02, ins=0.002, dele=0.002).tobytes()
                seqs.append(m)
                names.append("%s_g%d_c%d" % (nm, gi_, c))
                seqs.append(m.translate(COMP)[::-1])
                names.append("%s_rc_g%d_c%d" % (nm, gi_, c))
            if len(seqs) >= 24:
                break
        if len(seqs) >= 24:
            break
    assert len(seqs) >= 12
    oi, gi = O.Index(d), la.Index(d)
    got = _equal(oi, gi, seqs, names)
    gi.close()
    oi.close()
    # the inputs do what they are meant to: some query has HSPs on two different contigs of one genome, none crosses a spacer
    two_contigs = 0
    for rows in got:
        per_genome = {}
        for r in rows:
            per_genome.setdefault(r["batch_genome"], set()).add(r["seq_idx"])
            assert 0 <= r["tbegin"] <= r["tend"] < r["seq_len"]
        two_contigs += any(len(v) >= 2 for v in per_genome.values())
    assert two_contigs >= 4


@pytest.mark.parametrize("opt", [dict(max_evalue=1e-200), dict(min_qcov_hsp=60.0), dict(align_min_pident=95.0),
                                 dict(min_qcov_genome=99.9), dict(max_evalue=1e-50, min_qcov_hsp=20.0, align_min_pident=90.0)])
def test_each_hsp_filter_alone(mc_index, opt):
    la = _la()
    d, genomes = mc_index
    qs, _ = _queries(genomes)
    seqs, names = [q[1] for q in qs], [q[0] for q in qs]
    hip_names = dict(max_evalue="max_evalue", min_qcov_hsp="min_qcov_per_hsp", align_min_pident="align_min_pident",
                     min_qcov_genome="min_qcov_per_genome")
    base_o, base_g = O.Index(d), la.Index(d)
    unfiltered = _equal(base_o, base_g, seqs, names)
    base_g.close()
    base_o.close()
    oi = O.Index(d, O.default_search_opt(**opt))
    gi = la.Index(d, la.api.default_options(**{hip_names[k]: v for k, v in opt.items()}))
    got = _equal(oi, gi, seqs, names)
    gi.close()
    oi.close()
    n0, n1 = sum(len(r) for r in unfiltered), sum(len(r) for r in got)
    assert 0 < n1 < n0, (opt, n0, n1)   # the filter removes rows, and not all of them
