"""The INPUT data of two of the reference's own unit tests (tests/golden/ref_test_inputs.json, extracted by
tools/make_ref_test_inputs.py) through the oracle and through the product's device algorithms compiled for the host.

The reference's tests only log their results; what they DO hold as expectations is in their comments:
  * lib-chaining_test.go:41-44 - of the anchors (552, 3798905), (667, 3799019), (1332, 3799686) the comment says an
    earlier chainer gave "two chains: 0,1 and 2., while it should be one": the chainer as it is in the tree (and its
    restatement) gives the ONE chain;
  * lib-seq_compare_test.go:56-75 - the BLAST-style alignment of the two sequences: Query (s2) 8..295, Sbjct (s1) 15..294.
Both are asserted here, beside the committed oracle outputs and the oracle = product equality."""
import ctypes as C
import json
import os

import numpy as np

import hostalgos as H
import oracle as O
from test_device_algos_cpu import K, compare_device_vs_oracle, gap_lut

FIX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_test_inputs.json")))


def test_chaining_test_anchors_clear_and_chain():
    L, Hh = O.lib(), H.lib()
    subs = FIX["chaining"]["subs"]
    assert len(subs) == 35
    oa = (O.Sub * len(subs))()
    for i, (q, t, ln) in enumerate(subs):
        oa[i].qbegin, oa[i].tbegin, oa[i].len = q, t, ln
    n = L.lmo_clear_subs(oa, len(subs), K)
    assert [[oa[i].qbegin, oa[i].tbegin, oa[i].len] for i in range(n)] == FIX["chaining"]["cleared"]
    coff, cidx, nch = C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.c_int()
    so = L.lmo_chainer(oa, n, 50.0, L.lmo_seed_weight(17.0), 1000.0, 0, C.byref(coff), C.byref(cidx), C.byref(nch))
    chains = [[cidx[j] for j in range(coff[c], coff[c + 1])] for c in range(nch.value)]
    assert chains == FIX["chaining"]["chains"]
    assert int(np.float32(so).view(np.uint32)) == FIX["chaining"]["score_f32_bits"]
    # what the reference's test says the outcome SHOULD be: the three collinear anchors in one chain
    at = {(oa[i].qbegin, oa[i].tbegin): i for i in range(n)}
    a0, a1, a2 = at[(552, 3798905)], at[(667, 3799019)], at[(1332, 3799686)]
    assert sorted([a0, a1, a2]) in [sorted(c) for c in chains]
    # the product: pack -> sort -> unpack -> lm_clear_sorted -> lm_run_chain1 (what k_chain1 / k_chain1_wave run)
    packed = sorted(Hh.ha_pack_anchor(q, ln, t, 0, 0) for (q, t, ln) in subs)
    da = (H.Sub * len(subs))()
    for i, v in enumerate(packed):
        Hh.ha_unpack_anchor(v, C.byref(da[i]))
    nd = Hh.ha_clear_sorted(da, len(subs), K)
    assert nd == n and [[da[i].qbegin, da[i].tbegin, da[i].len] for i in range(nd)] == FIX["chaining"]["cleared"]
    doff = (C.c_int32 * (n + 4))()
    didx = (C.c_int32 * (2 * n + 6))()
    dn = C.c_int()
    sd = Hh.ha_chain1(da, n, 50.0, L.lmo_seed_weight(17.0), 1000.0, 0, gap_lut(), 64, doff, didx, C.byref(dn))
    assert np.float32(sd).tobytes() == np.float32(so).tobytes()
    assert [[didx[j] for j in range(doff[c], doff[c + 1])] for c in range(dn.value)] == chains
    L.free(coff)
    L.free(cidx)


def test_seq_compare_test_sequences_give_the_alignment_of_the_reference_comment():
    s1, s2 = FIX["compare"]["s1"].encode(), FIX["compare"]["s2"].encode()
    # cpr.Index(s1); cpr.Compare(0, len(s2)-1, s2, len(s2))  (lib-seq_compare_test.go:77-88)
    chains = compare_device_vs_oracle(s1, s2, 0, len(s2) - 1, query_len=len(s2))   # asserts oracle == device formulation
    assert [list(c[:7]) + [repr(c[7])] for c in chains] == FIX["compare"]["chains"]
    assert len(chains) == 1
    qb, qe, tb, te = chains[0][:4]
    # the alignment in the reference's comment, 1-based: Sbjct (the indexed s1) 15..294, Query (s2) 8..295
    assert (qb + 1, qe + 1, tb + 1, te + 1) == (15, 294, 8, 295)
