"""GPU parity of the workgroup WFA passes (k_wfa_mw2<2 / 4, WIN>: four wavefronts per alignment, 512 / 1024 diagonals) and of
the switch that chooses between two device implementations of the wide passes (LM_WFA_MW): either form against
the oracle (lmo_wfa_align = the restatement of wfa v0.5.0 as lib-index-search.go:2261 calls it), and the rows of the long-read
fixture with each switch on and off.

The forced-path test builds pairs whose final diagonal tlen - qlen lies 300-700 diagonals away from diagonal 0: the wavefront
must span both, so a pair outgrows the 256-diagonal ring (and with |tlen - qlen| > 510 the 512-diagonal ring too) whatever
its divergence.  8-32-kb pairs go through the windowed kernels (k_wfa_mww512 / 1024), 32-65-kb pairs through the
whole-sequence ones (k_wfa_mw512 / 1024); the profile of the call says which kernels ran.
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


def _la():
    import lexicmap_amd as la
    return la


def _pair(rng, n, ak):
    """ONT-style pair of ~n bases whose length difference is exactly ak (an insertion / deletion of random bases in the middle
    makes up for what the mutations left)"""
    from lexicmap_amd import synth
    q = synth.random_seq(rng, n)
    t = synth.mutate(rng, q, sub=0.02, ins=0.02, dele=0.03)
    d = ak - (len(t) - len(q))
    mid = len(t) // 2
    if d > 0:
        t = np.concatenate([t[:mid], synth.random_seq(rng, d), t[mid:]])
    elif d < 0:
        t = np.concatenate([t[:mid], t[mid - d:]])
    assert len(t) - len(q) == ak
    return q.tobytes(), t.tobytes()


@pytest.fixture(scope="module")
def tiny_index(tmp_path_factory):
    from lexicmap_amd import synth
    d = str(tmp_path_factory.mktemp("mwidx") / "t.lmi")
    genomes = synth.make_genomes(2, 60000, 1, seed=3, max_div=0.05)
    O.build_index(d, genomes, O.default_build_opt(chunks=2))
    return d


def _check(pairs, got):
    L = O.lib()
    for (q, t), g in zip(pairs, got):
        r = O.WfaResult()
        assert L.lmo_wfa_align(q, len(q), t, len(t), 1, C.byref(r)) == 0
        assert g["status"] in (0, 2)   # 2: an alignment without a single match operation
        assert g["score"] == r.score
        assert g["ops"] == [r.ops[i] for i in range(r.nops)]
        assert (g["qbegin"], g["qend"], g["tbegin"], g["tend"], g["align_len"], g["matches"], g["gaps"],
                g["gap_regions"]) == (r.qbegin, r.qend, r.tbegin, r.tend, r.align_len, r.matches, r.gaps, r.gap_regions)
        L.lmo_wfa_result_free(C.byref(r))


def test_forced_workgroup_passes_equal_the_oracle(tiny_index, monkeypatch):
    la = _la()
    rng = np.random.default_rng(77)
    pairs = []
    for n, ak in [(20000, 300), (26000, -420), (30000, 560), (22000, -700),      # 8-32 kb: windowed
                  (34000, 330), (45000, -400), (38000, 540), (44000, -650)]:     # 32-65 kb: whole sequences in LDS
        pairs.append(_pair(rng, n, ak))
    pairs.append(_pair(rng, 70000, 350))                                          # beyond 65 kb: windowed, starts at 512
    names = {}
    for mw in ("1", "0"):
        monkeypatch.setenv("LM_WFA_MW", mw)
        gi = la.Index(tiny_index)
        gi.profile(True)
        got = gi.wfa(pairs)
        names[mw] = {p["name"]: p["launches"] for p in gi.profile_get()}
        gi.close()
        _check(pairs, got)
    monkeypatch.delenv("LM_WFA_MW")
    for k in ("k_wfa_mw512", "k_wfa_mw1024", "k_wfa_mww512", "k_wfa_mww1024"):
        assert names["1"].get(k, 0) >= 1, (k, names["1"])
        assert names["0"].get(k, 0) == 0
    for k in ("k_wfa_lean512", "k_wfa_lean1024", "k_wfa_win512", "k_wfa_win1024"):
        assert names["0"].get(k, 0) >= 1, (k, names["0"])
        assert names["1"].get(k, 0) == 0


def test_workgroup_passes_on_small_and_degenerate_problems(tiny_index, monkeypatch):
    """what the wide passes see when a short problem reaches them: LM_WFA_FIRST_NC makes every class start at 512 diagonals"""
    la = _la()
    from lexicmap_amd import synth
    rng = np.random.default_rng(5)
    pairs = []
    for n, div in [(1, 0.0), (3, 0.5), (40, 0.0), (64, 0.3), (300, 0.05), (1500, 0.25), (2500, 0.12), (9000, 0.03)]:
        q = synth.random_seq(rng, n)
        t = synth.mutate(rng, q, sub=div, ins=div / 4, dele=div / 4)
        if len(t) == 0:
            t = synth.random_seq(rng, 2)
        pairs.append((q.tobytes(), t.tobytes()))
    monkeypatch.setenv("LM_WFA_FIRST_NC", "8,8,8,8,8")
    gi = la.Index(tiny_index)
    gi.profile(True)
    got = gi.wfa(pairs)
    names = {p["name"]: p["launches"] for p in gi.profile_get()}
    gi.close()
    monkeypatch.delenv("LM_WFA_FIRST_NC")
    _check(pairs, got)
    assert names.get("k_wfa_mw512", 0) >= 1 and names.get("k_wfa_lean", 0) == 0, names


def test_16_bit_ring_cells_equal_the_oracle_and_hand_over_before_they_could_wrap(tiny_index, monkeypatch):
    """k_wfa_lean<2 / 4, false, int16_t> (half the LDS per wavefront for the classes up to 8 kb): same results as the oracle with
    the switch on and off; a pair whose score passes 24 000 (where a 16-bit cell could wrap) must leave the 16-bit pass with
    status 3 and come back from the 32-bit one with the oracle's alignment"""
    la = _la()
    from lexicmap_amd import synth
    rng = np.random.default_rng(123)
    pairs = []
    for n, div in [(500, 0.1), (1900, 0.2), (2048, 0.05), (3000, 0.15), (6000, 0.1), (8000, 0.2), (8190, 0.02)]:
        q = synth.random_seq(rng, n)
        t = synth.mutate(rng, q, sub=div, ins=div / 4, dele=div / 4)
        pairs.append((q.tobytes(), t.tobytes()))
    pairs.append((synth.random_seq(rng, 7800).tobytes(), synth.random_seq(rng, 8000).tobytes()))   # unrelated sequences
    q = synth.random_seq(rng, 7000)
    pairs.append((q.tobytes(), synth.mutate(rng, q, sub=0.5, ins=0.1, dele=0.1).tobytes()))
    # 8 000 mismatches in a row between matching ends: score 32 000, nothing a shift or a gap could save
    pairs.append((b"ACGTTGCAGT" + b"A" * 8000 + b"TGCATTGACC", b"ACGTTGCAGT" + b"C" * 8000 + b"TGCATTGACC"))
    pairs.append((b"A" * 8100, b"C" * 8100))
    res = {}
    for r16 in ("1", "0"):
        monkeypatch.setenv("LM_WFA_R16", r16)
        gi = la.Index(tiny_index)
        res[r16] = gi.wfa(pairs)
        gi.close()
        _check(pairs, res[r16])
    monkeypatch.delenv("LM_WFA_R16")
    assert res["1"] == res["0"]
    assert max(g["score"] for g in res["1"]) >= 24000
