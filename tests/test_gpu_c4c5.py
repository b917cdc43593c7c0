"""GPU parity on the shapes of BASELINE configs 4 and 5: circular plasmid/prophage-style queries of
50, 120 and 200 kb and a 300-kb query against an index SHARDED OVER 4 handles (g % 4, all on this device), and a mixed
gene + read batch against the same shards.  For every batch: rows of the unsharded HIP index == oracle rows (row for row),
and per-shard rows -> lm_merge_sharded (what the Go host calls after its all-gatherv) == the unsharded rows, in order.

What these inputs reach that nothing else does: a target window and an HSP >= 250 kb - minimum pseudo-alignment prefix 17
(lib-seq_compare.go:339-348) and the extendMatch flank 50+40 (lib-index-search.go:2245-2254) - WFA problems of 200-300 kb,
circular queries whose two arcs chain separately on each of the 4 family members, shards that hold only part of a family.
"""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

NSHARD = 4
COMP = bytes.maketrans(b"ACGT", b"TGCA")


def _la():
    import lexicmap_amd as la
    return la


@pytest.fixture(scope="module")
def c4_index(tmp_path_factory):
    """8 genomes x 600 kb, 2 families of 4 (<= 4 % divergence), one contig each: shard r of 4 holds one member of each"""
    from lexicmap_amd import synth
    d = str(tmp_path_factory.mktemp("c4idx") / "c4.lmi")
    genomes = synth.make_genomes(8, 600_000, 2, seed=71, max_div=0.04, contigs=(1, 1))
    O.build_index(d, genomes, O.default_build_opt(chunks=4))
    return d, genomes


def _circular(rng, s, start, length, rot, rc, sub=0.01, indel=0.002):
    from lexicmap_amd import synth
    region = s[start:start + length]
    q = np.concatenate([region[rot:], region[:rot]])
    q = synth.mutate(rng, q, sub=sub, ins=indel, dele=indel).tobytes()
    return q.translate(COMP)[::-1] if rc else q


@pytest.fixture(scope="module")
def c4_queries(c4_index):
    _, genomes = c4_index
    rng = np.random.default_rng(72)
    s2 = np.frombuffer(genomes[2][1][0][1], dtype=np.uint8)
    s5 = np.frombuffer(genomes[5][1][0][1], dtype=np.uint8)
    return [
        ("circ50k", _circular(rng, s2, 20_000, 50_000, 17_000, False)),
        ("circ120k_rc", _circular(rng, s5, 100_000, 120_000, 70_000, True)),
        ("circ200k", _circular(rng, s2, 350_000, 200_000, 111_000, False)),
        # not rotated: ONE chain of 300 kb -> window and HSP >= 250 kb (min prefix 17, flank 50+40)
        ("lin300k_rc", _circular(rng, s5, 250_000, 300_000, 0, True, sub=0.004, indel=0.001)),
    ]


@pytest.fixture(scope="module")
def c5_queries(c4_index):
    """C5 batch shape: 90 % gene queries (1-2 kb) + 10 % ONT-style reads"""
    from lexicmap_amd import synth
    _, genomes = c4_index
    qs = synth.make_gene_queries(genomes, 18, seed=73, len_range=(1000, 2000), max_div=0.08)
    qs[9:9] = synth.make_reads(genomes, 2, seed=74, len_range=(5000, 30000))  # reads in the middle of the batch
    return qs


ROW_FIELDS = ("batch_genome", "cls", "hsp", "seq_idx", "nseqs", "seq_len", "rc", "aligned_length", "gaps", "qbegin", "qend",
              "tbegin", "tend", "bitscore", "score", "matched_bases", "qcov_genome", "qcov_hsp", "pident")


def _check_batch(d, qs, expect_min_rows):
    la = _la()
    from lexicmap_amd import merge
    seqs = [q[1] for q in qs]
    oi = O.Index(d)
    whole = la.Index(d)
    tb = whole.info()["total_bases"]
    qb = whole.upload(seqs)
    arr_w, _ = whole.search_resident_np(qb)
    arr_w = arr_w.copy()
    whole.free_batch(qb)
    rows_w, _ = whole.search(seqs)
    whole.close()
    # 1) unsharded HIP rows == oracle rows, row for row
    by_q = {}
    for r in rows_w:
        by_q.setdefault(r["query"], []).append(r)
    nrows = 0
    for qi, s in enumerate(seqs):
        exp, st = oi.search(s)
        got = by_q.get(qi, [])
        assert len(exp) == len(got), (qs[qi][0], len(exp), len(got))
        for e, g in zip(exp, got):
            for f in ROW_FIELDS:
                assert e[f] == g[f], (qs[qi][0], f, e[f], g[f])
            assert abs(e["evalue"] - g["evalue"]) <= 1e-9 * max(abs(e["evalue"]), 1e-300)
            assert g["hits"] == st["ngenomes"]
        nrows += len(exp)
    oi.close()
    assert nrows >= expect_min_rows
    # 2) 4 shard handles on this device -> lm_merge_sharded == the unsharded rows, same order, same hits
    shards = [la.Index(d, la.api.default_options(shard_rank=r, shard_count=NSHARD, total_bases_override=tb))
              for r in range(NSHARD)]
    per_rank = []
    for si in shards:
        assert si.info()["genomes"] == 2
        qb = si.upload(seqs)
        arr, _ = si.search_resident_np(qb)
        per_rank.append(arr.copy())
        si.free_batch(qb)
    merged = merge.merge_sharded_c(per_rank)
    for si in shards:
        si.close()
    assert len(merged) == len(arr_w) == nrows
    for f in merged.dtype.names:
        if f in ("genome_id", "seq_id", "cigar", "qseq", "sseq", "align"):
            continue
        assert (merged[f] == arr_w[f]).all(), f
    return rows_w


def test_c4_circular_long_queries_on_4_shards_equal_oracle(c4_index, c4_queries):
    d, _ = c4_index
    rows = _check_batch(d, c4_queries, expect_min_rows=20)
    longest = max(r["aligned_length"] for r in rows)
    widest = max(r["tend"] - r["tbegin"] + 1 for r in rows)
    assert longest > 250_000 and widest > 250_000   # min prefix 17 + flank 50+40 were exercised
    # every circular query hits each family member with (at least) its two arcs
    for qi in range(3):
        per_genome = {}
        for r in rows:
            if r["query"] == qi:
                per_genome[r["batch_genome"]] = per_genome.get(r["batch_genome"], 0) + 1
        assert len(per_genome) == 4 and min(per_genome.values()) >= 2, (qi, per_genome)


def test_c5_mixed_gene_and_read_batch_on_4_shards_equals_oracle(c4_index, c5_queries):
    d, _ = c4_index
    rows = _check_batch(d, c5_queries, expect_min_rows=60)
    assert {r["query"] for r in rows} >= set(range(len(c5_queries))) - {-1}
