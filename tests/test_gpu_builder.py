"""GPU synthetic-index builder (lexicmap_amd/csrc/lm_builder.hip): structural invariants of the HBM index it produces,
exactness of its LexicHash captures against the oracle, and end-to-end search sanity on it."""
import ctypes as C

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def synth_index():
    import lexicmap_amd as la
    gi = la.Index.synthetic(genomes=12, genome_len=200_000, families=3, seed=77, max_div=0.10)
    yield gi
    gi.close()


def test_synthetic_genomes_family_structure(synth_index):
    gi = synth_index
    a = np.frombuffer(gi.fetch(0, 0, 200_000), dtype=np.uint8)      # ancestor of family 0
    b = np.frombuffer(gi.fetch(3, 0, 200_000), dtype=np.uint8)      # member of family 0
    c = np.frombuffer(gi.fetch(1, 0, 200_000), dtype=np.uint8)      # other family
    assert set(np.unique(a)) <= set(b"ACGT")
    # composition ~uniform
    assert all(abs((a == x).mean() - 0.25) < 0.01 for x in b"ACGT")
    ident_c = (a[:5000] == c[:5000]).mean()
    assert 0.2 < ident_c < 0.3                                      # unrelated
    # members are substituted + indel-shifted copies: some small shift aligns the first block
    ident_b = max((a[8 + sh:408 + sh] == b[8:408]).mean() for sh in range(-8, 9))
    assert ident_b > 0.85


def test_captures_match_oracle_lexichash(synth_index):
    """normal (non-desert) seeds are the exact LexicHash capture of the genome: for every mask, the k-mer the oracle's
    lexichash captures on the fetched genome is stored under that mask with exactly the oracle's (position, strand) set
    (lib-index-build.go:1028-1046), read back through lm_index_mask_seeds"""
    import hostalgos as H
    import lexicmap_amd as la
    gi = synth_index
    L = O.lib()
    info = gi.info()
    M = info["masks"]
    masks_p = la.lib().lm_index_masks(gi.h)
    masks = (C.c_uint64 * M)(*[masks_p[i] for i in range(M)])
    lh = L.lmo_lh_new(31, masks, M)
    g = 4
    seq = gi.fetch(g, 0, 200_000)
    kmers = (C.c_uint64 * M)()
    off, locs = C.POINTER(C.c_int)(), C.POINTER(C.c_int)()
    assert L.lmo_lh_mask(lh, seq, len(seq), None, 0, 1, kmers, C.byref(off), C.byref(locs)) == 0
    Hh = H.lib()
    checked = 0
    for m in range(0, M, 3):
        if kmers[m] == 0 or Hh.ha_low_complexity(kmers[m], 31):
            continue
        exp = sorted(int(locs[i]) for i in range(off[m], off[m + 1]))       # pos<<1 | strand
        k, v = gi.mask_seeds(m)
        sel = (k == np.uint64(kmers[m])) & ((v >> np.uint64(30)) == np.uint64(g)) & ((v & np.uint64(1)) == 0)
        got = sorted(int(x) for x in ((v[sel] >> np.uint64(1)) & np.uint64((1 << 29) - 1)))
        assert got == exp, (m, got[:4], exp[:4])
        # and its reversed twin sits in some list with the reversed flag (lib-index-build.go:776-890)
        checked += 1
    assert checked > 0.25 * M
    # the lists are sorted by k-mer inside each direction, values carry the direction of their half
    k, v = gi.mask_seeds(123)
    nrm = (v & np.uint64(1)) == 0
    assert nrm.any() and (~nrm).any()
    first_rev = int(np.argmax(~nrm))
    assert nrm[:first_rev].all() and (~nrm[first_rev:]).all()
    assert (np.diff(k[:first_rev].astype(np.int64)) >= 0).all() and (np.diff(k[first_rev:].astype(np.int64)) >= 0).all()
    # a query cut from this genome must recover anchors on genome g at the right coordinates
    rows, st = gi.search([seq[50_000:51_500]])
    assert st["rows"] >= 1
    best = [r for r in rows if r["batch_genome"] == g]
    assert best and best[0]["pident"] == 100.0 and best[0]["tbegin"] == 50_000 and best[0]["tend"] == 51_499
    assert best[0]["genome_id"] == b"SYN_%09d.1" % g
    # family members are found too (3 families x 4 members)
    fam = {r["batch_genome"] for r in rows}
    assert fam == {1, 4, 7, 10}
    L.free(off)
    L.free(locs)
    L.lmo_lh_free(lh)


def test_every_mask_list_equals_the_oracle_writer(synth_index, tmp_path):
    """f4: the GPU builder stores, under EVERY mask, exactly the (k-mer, value) pairs the oracle's writer
    (lib-index-build.go restated: captures, desert filling with the window-capture condition :1094-1407, reversed twins
    :776-890) stores for the same genomes and the same mask set - compared list by list through lm_index_mask_seeds, the
    oracle-written index being loaded through the reference-format loader."""
    import lexicmap_amd as la
    gi = synth_index
    M = gi.info()["masks"]
    masks_p = la.lib().lm_index_masks(gi.h)
    masks = [masks_p[i] for i in range(M)]
    ng = gi.info()["genomes"]
    genomes = [("SYN_%09d.1" % g, [("syn%09d_c1" % g, gi.fetch(g, 0, 200_000))]) for g in range(ng)]
    d = str(tmp_path / "same.lmi")
    O.build_index(d, genomes, O.default_build_opt(chunks=4), masks=masks)
    oi = la.Index(d)
    assert oi.info()["seeds"] == gi.info()["seeds"]
    nrev = ndesert = 0
    for m in list(range(0, M, 7)) + [M - 1]:
        k1, v1 = gi.mask_seeds(m)
        k2, v2 = oi.mask_seeds(m)
        a = sorted(zip(k1.tolist(), v1.tolist()))
        b = sorted(zip(k2.tolist(), v2.tolist()))
        assert a == b, (m, len(a), len(b), [x for x in a if x not in b][:3], [x for x in b if x not in a][:3])
        nrev += int((v1 & np.uint64(1)).sum())
        ndesert += len(set(k1[(v1 & np.uint64(1)) == 0].tolist())) - 1
    oi.close()
    assert nrev > 1000 and ndesert > 1000   # reversed twins and desert seeds were among what was compared


def test_saved_index_round_trip_and_oracle_reads_it(synth_index, tmp_path):
    """f1 / a20: lm_index_save writes the HBM image in the reference's on-disk format; the loader rebuilds the SAME packed
    image from it (same seeds under every sampled mask, same rows), and the oracle - an independent reader of that format -
    opens it and returns the same rows"""
    import lexicmap_amd as la
    gi = synth_index
    d = str(tmp_path / "saved.lmi")
    gi.save(d, chunks=5)
    li = la.Index(d)
    a, b = gi.info(), li.info()
    for f in ("k", "masks", "genomes", "seeds", "genome_bases", "total_bases", "outlier_seeds", "key_bits", "partition_bases"):
        assert a[f] == b[f], f
    for m in (0, 1, 77, 4099, 12345, 19999):
        k1, v1 = gi.mask_seeds(m)
        k2, v2 = li.mask_seeds(m)
        assert sorted(zip(k1.tolist(), v1.tolist())) == sorted(zip(k2.tolist(), v2.tolist())), m
    seqs = [gi.fetch(4, 50_000, 1500), gi.fetch(7, 120_000, 3000)[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA")),
            gi.fetch(2, 10, 800)]
    r1, _ = gi.search(seqs)
    r2, _ = li.search(seqs)
    assert len(r1) == len(r2) > 6
    for x, y in zip(r1, r2):
        for f in x:
            if f not in ("genome_id", "seq_id"):
                assert x[f] == y[f], f
        assert x["genome_id"] == y["genome_id"] and x["seq_id"] == y["seq_id"]
    li.close()
    oi = O.Index(d)
    n = 0
    for qi, s in enumerate(seqs):
        exp, st = oi.search(s)
        got = [r for r in r1 if r["query"] == qi]
        assert len(exp) == len(got)
        for e, g in zip(exp, got):
            for f in ("batch_genome", "aligned_length", "qbegin", "qend", "tbegin", "tend", "bitscore", "gaps", "pident"):
                assert e[f] == g[f], (qi, f)
        n += len(exp)
    oi.close()
    assert n == len(r1)


def test_seed_density_like_reference_builder(synth_index):
    """~2*(M + 0.9*L/50) seeds per genome: the desert filling and the reversed copies are in place"""
    info = synth_index.info()
    per_genome = info["seeds"] / info["genomes"]
    expect = 2 * (20000 * 0.9 + 0.8 * 200_000 / 50)
    assert 0.6 * expect < per_genome < 1.6 * expect, per_genome
