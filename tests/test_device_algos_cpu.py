"""CPU checks of the product's per-work-item DEVICE algorithms (lexicmap_amd/csrc/lm_algos.h, compiled for the host
by tests/host_algos.cpp) against the oracle.  The kernels in lm_kernels.hip are thin wrappers around these functions,
so a green run here means the GPU parity tests only have to catch launch/indexing mistakes."""
import ctypes as C
import random

import numpy as np
import pytest

import hostalgos as H
import oracle as O

K = 31


def rand_seq(rng, n, alphabet=b"ACGT"):
    return bytes(rng.choice(list(alphabet)) for _ in range(n))


def mutate(rng, s, sub=0.05, ins=0.01, dele=0.01):
    out = bytearray()
    for c in s:
        r = rng.random()
        if r < dele:
            continue
        if r < dele + sub:
            out.append(rng.choice([x for x in b"ACGT" if x != c]))
        else:
            out.append(c)
        if rng.random() < ins:
            out.append(rng.choice(list(b"ACGT")))
    return bytes(out)


def test_kmer_bitops_match_oracle():
    L, Hh = O.lib(), H.lib()
    rng = random.Random(1)
    for k in (5, 21, 31, 32):
        for _ in range(300):
            x = rng.getrandbits(2 * k)
            assert Hh.ha_revcomp(x, k) == L.lmo_kmer_revcomp(x, k)
            assert Hh.ha_reverse(x, k) == L.lmo_kmer_reverse(x, k)
    # low complexity: random, homopolymers, short tandem repeats
    cases = [rng.getrandbits(62) for _ in range(2000)]
    for unit in (b"A", b"C", b"G", b"T", b"AC", b"AT", b"CG", b"AAC", b"ACG", b"AACG", b"ACGTT"):
        s = (unit * 40)[:K]
        cases.append(L.lmo_kmer_encode(s, K))
        s2 = bytearray(s)
        s2[7] = ord("G")
        cases.append(L.lmo_kmer_encode(bytes(s2), K))
    for x in cases:
        assert bool(Hh.ha_dust(x, K)) == bool(L.lmo_dust(x, K))
        assert bool(Hh.ha_low_complexity(x, K)) == bool(L.lmo_low_complexity(x, K))
    for c in range(256):
        enc = L.lmo_kmer_encode(bytes([c]), 1)
        assert Hh.ha_base2bit(c) == enc


@pytest.mark.parametrize("qlen,M", [(40, 64), (300, 256), (1500, 4096), (1500, 20000), (9000, 20000)])
def test_xor_argmin_reproduces_lexichash_mask(qlen, M):
    """device masking = sort the query's k-mers + lm_xor_argmin per mask; must equal the oracle's Mask()"""
    L, Hh = O.lib(), H.lib()
    rng = random.Random(qlen * 7 + M)
    masks = (C.c_uint64 * M)()
    L.lmo_gen_masks(K, M, 1, masks)
    lh = L.lmo_lh_new(K, masks, M)
    seq = rand_seq(rng, qlen)
    if qlen >= 300:  # plant repeats and a poly-A stretch
        seq = seq[:100] + seq[20:80] + b"A" * 40 + seq[100:]
        seq = seq[:qlen]
    kmers = (C.c_uint64 * M)()
    off = C.POINTER(C.c_int)()
    locs = C.POINTER(C.c_int)()
    assert L.lmo_lh_mask(lh, seq, len(seq), None, 0, 1, kmers, C.byref(off), C.byref(locs)) == 0
    # device-side formulation
    n = len(seq) - K + 1
    keys, vals = [], []
    for i in range(n):
        f = L.lmo_kmer_encode(seq[i:i + K], K)
        keys.append(f)
        vals.append(i << 1)
        keys.append(L.lmo_kmer_revcomp(f, K))
        vals.append(i << 1 | 1)
    order = sorted(range(2 * n), key=lambda j: (keys[j], j))  # stable
    skeys = (C.c_uint64 * (2 * n))(*[keys[j] for j in order])
    svals = [vals[j] for j in order]
    lo, hi = C.c_int(), C.c_int()
    for m in range(M):
        w = Hh.ha_xor_argmin(skeys, 2 * n, masks[m], C.byref(lo), C.byref(hi))
        assert w == kmers[m], (m, w, kmers[m])
        assert svals[lo.value:hi.value] == [locs[j] for j in range(off[m], off[m + 1])]
    L.free(off)
    L.free(locs)
    L.lmo_lh_free(lh)


def random_anchors(rng, n, qspan, tspan, diag_frac=0.7, rc_frac=0.2):
    subs = []
    for _ in range(n):
        ln = rng.randint(15, 31)
        if rng.random() < diag_frac:
            q = rng.randrange(qspan)
            off = rng.choice([0, 0, 0, 1, -2, 5, 37, -60])
            if rng.random() < rc_frac:
                t = max(0, tspan // 2 - q + off)
            else:
                t = max(0, q + 1000 + off)
        else:
            q, t = rng.randrange(qspan), rng.randrange(tspan)
        subs.append((q, t, ln, rng.random() < 0.3, rng.random() < 0.3))
    # exact duplicates and nested anchors
    for _ in range(n // 10):
        q, t, ln, a, b = rng.choice(subs)
        subs.append((q, t, ln, a, b))
        if ln > 17:
            subs.append((q + 1, t + 1, ln - 2, a, b))
    return subs


def to_oracle_subs(subs):
    arr = (O.Sub * len(subs))()
    for i, (q, t, ln, qrc, trc) in enumerate(subs):
        arr[i].qbegin, arr[i].tbegin, arr[i].len, arr[i].qrc, arr[i].trc = q, t, ln, int(qrc), int(trc)
    return arr


def gap_lut(n=64):
    L = O.lib()
    return (C.c_float * n)(*[L.lmo_gap_score(float(g)) for g in range(n)])


@pytest.mark.parametrize("n,qspan,tspan,seed", [(1, 100, 100, 0), (2, 50, 50, 1), (12, 300, 3000, 2), (80, 1500, 6000, 3),
                                               (400, 1500, 4000, 4), (1500, 20000, 50000, 5), (300, 200, 400, 6)])
def test_clear_and_chain1_match_oracle(n, qspan, tspan, seed):
    L, Hh = O.lib(), H.lib()
    rng = random.Random(seed)
    subs = random_anchors(rng, n, qspan, tspan)
    # oracle: clear (sorts itself) + chain
    oa = to_oracle_subs(subs)
    n_o = L.lmo_clear_subs(oa, len(subs), K) if len(subs) > 1 else len(subs)
    # device formulation: pack -> sort u64 -> unpack -> clear_sorted
    packed = sorted(Hh.ha_pack_anchor(q, ln, t, int(a), int(b)) for (q, t, ln, a, b) in subs)
    da = (H.Sub * len(subs))()
    for i, v in enumerate(packed):
        Hh.ha_unpack_anchor(v, C.byref(da[i]))
    n_d = Hh.ha_clear_sorted(da, len(subs), K) if len(subs) > 1 else len(subs)
    assert n_d == n_o
    for i in range(n_o):
        assert (da[i].qbegin, da[i].tbegin, da[i].len, da[i].qrc, da[i].trc) == \
               (oa[i].qbegin, oa[i].tbegin, oa[i].len, oa[i].qrc, oa[i].trc)
    lut = gap_lut()
    for (max_gap, max_dist, top) in ((50.0, 1000.0, 0), (50.0, 1000.0, 1), (20.0, 300.0, 2)):
        min_score = L.lmo_seed_weight(17.0)
        coff, cidx, nch = C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.c_int()
        so = L.lmo_chainer(oa, n_o, max_gap, min_score, max_dist, top, C.byref(coff), C.byref(cidx), C.byref(nch))
        doff = (C.c_int32 * (n_o + 4))()
        didx = (C.c_int32 * (2 * n_o + 6))()
        dn = C.c_int()
        sd = Hh.ha_chain1(da, n_o, max_gap, min_score, max_dist, top, lut, 64, doff, didx, C.byref(dn))
        assert np.float32(sd).tobytes() == np.float32(so).tobytes()
        assert dn.value == nch.value
        assert [doff[i] for i in range(dn.value + 1)] == [coff[i] for i in range(nch.value + 1)]
        assert [didx[i] for i in range(doff[dn.value])] == [cidx[i] for i in range(coff[nch.value])]
        L.free(coff)
        L.free(cidx)


def test_go_log2_lut_is_what_c_log2_gives_after_float32_rounding():
    """appendix B.5: gapScore uses Go's math.Log2; record whether it ever differs from libm after rounding"""
    L = O.lib()
    import math
    for g in range(1, 5001):
        a = np.float32(L.lmo_go_log2(float(g)))
        b = np.float32(math.log2(g))
        assert a == b, g


def compare_setup(rng, qlen, div, tflank=300):
    core = rand_seq(rng, qlen)
    if qlen > 400:  # tandem repeat + low complexity inside the query
        core = core[:150] + b"ACGACGACGACGACGACGACGACGACGACGACGACG" + core[150:300] + b"A" * 33 + core[300:]
    t = rand_seq(rng, tflank) + mutate(rng, core, sub=div, ins=div / 6, dele=div / 6) + rand_seq(rng, tflank)
    return core, t


def compare_device_vs_oracle(q, t, begin, end, query_len=None):
    """SeqComparator.Index(q) + Compare(begin, end, t): the oracle's radix-tree pipeline (lmo_cmp_compare) against the
    device formulation (sorted-array tree.Search emulation -> anchors -> clear -> trim -> chain2); asserts equality and
    returns the chains (qbegin, qend, tbegin, tend, nanchors, matched_bases, aligned_bases_q, pident)"""
    L, Hh = O.lib(), H.lib()
    opt = O.CmpOpt()
    opt.k, opt.min_prefix = K, 11
    opt.c2.max_gap, opt.c2.min_score, opt.c2.min_align_len = 20, 35, 50
    opt.c2.min_identity, opt.c2.band_count, opt.c2.band_base, opt.c2.heuristic_pident = 70.0, 50, 100, 15.0
    cmp_ = L.lmo_cmp_new(C.byref(opt))
    assert L.lmo_cmp_index(cmp_, q, len(q)) == 0
    chains = C.POINTER(O.Chain2)()
    osubs = C.POINTER(O.Sub)()
    nosubs = C.c_int()
    nc = L.lmo_cmp_compare(cmp_, begin, end, t, len(t), query_len if query_len is not None else len(q), C.byref(chains), C.byref(osubs), C.byref(nosubs))
    # ---- device formulation ----
    keys, vals = [], []
    for i in range(len(q) - K + 1):
        f = L.lmo_kmer_encode(q[i:i + K], K)
        if f == 0 or L.lmo_low_complexity(f, K):
            continue
        keys += [f, L.lmo_kmer_revcomp(f, K)]
        vals += [i << 1, i << 1 | 1]
    order = sorted(range(len(keys)), key=lambda j: (keys[j], j))
    skeys = (C.c_uint64 * len(keys))(*[keys[j] for j in order])
    svals = [vals[j] for j in order]
    ccc, ggg, ttt = [int(c * K, 4) for c in "123"]
    packed = []
    lo, hi = C.c_int(), C.c_int()
    m = 11 + (2 if len(t) >= 10000 else 0)
    for idx in range(len(t) - K + 1):
        f = L.lmo_kmer_encode(t[idx:idx + K], K)
        if f in (0, ccc, ggg, ttt):
            continue
        rc = L.lmo_kmer_revcomp(f, K)
        if Hh.ha_tree_search_range(skeys, len(keys), f, m, K, C.byref(lo), C.byref(hi)):
            for j in range(lo.value, hi.value):
                v = svals[j]
                lp = (64 - (skeys[j] ^ f).bit_length()) // 2 + K - 32 if skeys[j] != f else K
                p = v >> 1
                if v & 1 or p < begin or p + lp > end:
                    continue
                packed.append(Hh.ha_pack_anchor(p, lp, idx, 0, 0))
        if Hh.ha_tree_search_range(skeys, len(keys), rc, m, K, C.byref(lo), C.byref(hi)):
            for j in range(lo.value, hi.value):
                v = svals[j]
                lp = (64 - (skeys[j] ^ rc).bit_length()) // 2 + K - 32 if skeys[j] != rc else K
                p = (v >> 1) + K - lp
                if not (v & 1) or p + lp < begin or p > end:
                    continue
                packed.append(Hh.ha_pack_anchor(p, lp, idx + K - lp, 1, 1))
    packed.sort()
    n = len(packed)
    da = (H.Sub * max(n, 1))()
    for i, v in enumerate(packed):
        Hh.ha_unpack_anchor(v, C.byref(da[i]))
    if n > 1:
        n = Hh.ha_clear_sorted(da, n, K)
    start = C.c_int()
    n2 = Hh.ha_trim(da, n, 100.0, C.byref(start)) if n > 0 else 0
    assert n2 == nosubs.value
    kept = [(da[start.value + i].qbegin, da[start.value + i].tbegin, da[start.value + i].len, da[start.value + i].qrc)
            for i in range(n2)]
    assert kept == [(osubs[i].qbegin, osubs[i].tbegin, osubs[i].len, osubs[i].qrc) for i in range(n2)]
    out = (H.Chain2 * max(n2, 1))()
    ka = (H.Sub * max(n2, 1))(*[da[start.value + i] for i in range(n2)])
    dn = Hh.ha_chain2(ka, n2, 20, 35, 50, 50, 100, 15.0, out) if n2 > 0 else 0
    dchains = sorted([(out[i].qbegin, i) for i in range(dn)])  # stable sort by qbegin
    assert dn == nc
    for rank, (_, i) in enumerate(dchains):
        o, d = chains[rank], out[i]
        assert (d.qbegin, d.qend, d.tbegin, d.tend, d.nanchors, d.matched_bases, d.aligned_bases_q) == \
               (o.qbegin, o.qend, o.tbegin, o.tend, o.nanchors, o.matched_bases, o.aligned_bases_q)
        assert d.pident == o.pident
    res = [(chains[i].qbegin, chains[i].qend, chains[i].tbegin, chains[i].tend, chains[i].nanchors, chains[i].matched_bases,
            chains[i].aligned_bases_q, chains[i].pident) for i in range(nc)]
    L.lmo_cmp_free(cmp_)
    return res




@pytest.mark.parametrize("qlen,div,seed", [(120, 0.0, 1), (400, 0.03, 2), (1500, 0.08, 3), (1500, 0.2, 4), (5000, 0.1, 5)])
def test_tree_range_trim_chain2_match_oracle(qlen, div, seed):
    """pseudo-alignment: sorted-array tree.Search emulation (incl. the partial-prefix quirk) -> anchors -> clear ->
    trim -> chain2, against the oracle's radix tree pipeline (lmo_cmp_compare)."""
    rng = random.Random(seed)
    q, t = compare_setup(rng, qlen, div)
    chains = compare_device_vs_oracle(q, t, 0, len(q) - 1)
    if qlen >= 400:
        assert len(chains) >= 1


def test_tree_search_quirk_matches_radix_tree():
    """tree.go:496-500: with a short node edge the uint8 arithmetic accepts subtrees sharing < p bases when the key has
    a run of A's; the sorted-array emulation must return exactly what the radix tree returns."""
    L, Hh = O.lib(), H.lib()
    rng = random.Random(99)
    hits_quirk = 0
    hits_quirk_filter = 0
    for trial in range(60):
        keyset = set()
        base = rng.getrandbits(62)
        for _ in range(rng.randint(2, 300)):
            # many keys share 4-9 leading bases with `base`, so nodes with short edges exist
            share = rng.randint(0, 10)
            x = rng.getrandbits(62)
            sh = 62 - 2 * share
            keyset.add(((base >> sh) << sh) | (x & ((1 << sh) - 1)) if share else x)
        keys = sorted(keyset)
        t = L.lmo_tree_new(K)
        for i, x in enumerate(keys):
            L.lmo_tree_insert(t, x, i)
        arr = (C.c_uint64 * len(keys))(*keys)
        out = C.POINTER(O.TreeSr)()
        cap = C.c_int(0)
        lo, hi = C.c_int(), C.c_int()
        # the pseudo-alignment prefix filter of this key set (k_build_cmp_bits): small map -> hashed, 22 -> exact
        flog = rng.choice([13, 16, 20, 21])  # 20, 21: above the Bloom filter's size, the 11-base bitmap is asked too
        fbits = (C.c_uint32 * int(Hh.ha_pa_bits_words(flog)))()
        Hh.ha_pa_filter_build(arr, len(keys), K, flog, fbits)
        tabs = {}
        for tb in (12, 16):
            tabs[tb] = (C.c_uint32 * ((1 << tb) + 1))()
            Hh.ha_build_tab(arr, len(keys), K, tb, tabs[tb])
        for _ in range(400):
            share = rng.randint(1, 12)
            sh = 62 - 2 * share
            arun = rng.randint(0, 12)
            q = ((base >> sh) << sh)  # shared prefix then A-run then random
            tail_bits = max(0, sh - 2 * arun)
            q |= rng.getrandbits(tail_bits) if tail_bits else 0
            for p in (7, 11, 13, 15):
                n = L.lmo_tree_search(t, q, p, C.byref(out), C.byref(cap))
                got = Hh.ha_tree_search_range(arr, len(keys), q, p, K, C.byref(lo), C.byref(hi))
                exp = [out[i].kmer for i in range(n)]
                res = keys[lo.value:hi.value] if got else []
                assert res == exp, (trial, hex(q), p)
                if p >= 11:  # the form k_pa_search runs: bucket table + lower bound only, matches enumerated
                    for tb, tab in tabs.items():
                        got2 = Hh.ha_tree_search_first_tab(arr, len(keys), q, p, K, tab, tb, C.byref(lo), C.byref(hi))
                        assert (keys[lo.value:hi.value] if got2 else []) == exp, (trial, hex(q), p, tb)
                if n and out[0].len_prefix < p:
                    hits_quirk += 1
                    if p >= 11:
                        hits_quirk_filter += 1
                # k_pa_search only runs on positions k_pa_filter lets through on positions the filter lets through: never a false negative
                if n and p >= 11:
                    assert Hh.ha_pa_candidate(fbits, flog, q, p, K), (trial, hex(q), p, flog)
                    if p <= 15:  # the two-level form with the LDS-resident Bloom filter and exact 9-base map
                        assert Hh.ha_pa_candidate2(fbits, flog, q, p, K), (trial, hex(q), p, flog)
        L.free(out)
        L.lmo_tree_free(t)
    assert hits_quirk > 0  # the quirk path was really exercised
    assert hits_quirk_filter > 0  # also at the prefix lengths the filter is used with


def test_pa_filter_is_selective_not_only_necessary():
    """the necessity of lm_pa_candidate2 is checked against the radix tree above; this checks that it also REJECTS: for a
    gene-sized and a read-sized key set nearly all random window k-mers are turned away (what makes k_pa_filter worth
    running), while every k-mer of the set itself passes"""
    Hh = H.lib()
    rng = random.Random(4)
    for nkeys, flog in ((3000, 16), (54000, 20)):
        keys = sorted({rng.getrandbits(62) for _ in range(nkeys)})
        arr = (C.c_uint64 * len(keys))(*keys)
        fbits = (C.c_uint32 * int(Hh.ha_pa_bits_words(flog)))()
        Hh.ha_pa_filter_build(arr, len(keys), K, flog, fbits)
        for p in (11, 13, 15):
            assert all(Hh.ha_pa_candidate2(fbits, flog, x, p, K) for x in keys[::37])
            passed = sum(1 for _ in range(20000) if Hh.ha_pa_candidate2(fbits, flog, rng.getrandbits(62), p, K))
            # true 11-base prefix matches of random probes: nkeys / 4^11 (0.07 % / 1.3 %), plus the filters' false positives
            assert passed < (0.02 if nkeys < 10000 else 0.06) * 20000, (nkeys, p, passed)


@pytest.mark.parametrize("seed", range(6))
def test_extend_match_matches_oracle(seed):
    L, Hh = O.lib(), H.lib()
    rng = random.Random(100 + seed)
    for _ in range(40):
        core = rand_seq(rng, rng.randint(60, 400))
        lf, rf = rand_seq(rng, rng.randint(0, 90)), rand_seq(rng, rng.randint(0, 90))
        q = lf + core + rf
        t = mutate(rng, lf, 0.1, 0.03, 0.03) + core + mutate(rng, rf, 0.1, 0.03, 0.03)
        if rng.random() < 0.2:
            t = b"A" * 70 + t + b"A" * 70
            q = b"A" * 60 + q + b"A" * 65
        s1 = q.find(core)
        s2 = t.find(core)
        e1, e2 = s1 + len(core), s2 + len(core)
        ext = rng.choice([50, 60, 130])
        tb, mx, rc = rng.randint(0, 200), rng.randint(0, 200), rng.random() < 0.5
        o = [C.c_int() for _ in range(8)]
        L.lmo_extend_match(q, len(q), t, len(t), s1, e1, s2, e2, ext, tb, mx, int(rc), *[C.byref(x) for x in o])
        d = (C.c_int * 8)()
        Hh.ha_extend_match(q, len(q), t, len(t), s1, e1, s2, e2, ext, tb, mx, int(rc), d)
        assert [x.value for x in o] == list(d)


@pytest.mark.parametrize("seed", range(4))
def test_extend_flank_grid_equals_list_chainer(seed):
    """a14: the grid form of the flank chainer the HIP kernel runs (bit-parallel pairs + windowed predecessor search on
    the (q,t) grid, strided scratch) returns what the list form (Chainer3 as written, lib-chaining3.go) returns"""
    Hh = H.lib()
    rng = random.Random(900 + seed)
    for it in range(600):
        n1 = rng.choice([2, 3, 10, 30, 50, 50, 50, 60, 90, 130, 131, 132, 140])
        n2 = rng.choice([2, 5, 30, 50, 50, 50, 64, 65, 100, 127, 128, 129, 130])
        kind = rng.random()
        if kind < 0.15:      # low complexity: many pairs
            a = bytes(rng.choice(b"AT") for _ in range(n1))
            b = bytes(rng.choice(b"AT") for _ in range(n2))
        elif kind < 0.25:
            a = b"A" * n1
            b = b"A" * (n2 - 1) + b"C"
        elif kind < 0.75:    # homologous flanks with substitutions / indels
            a = rand_seq(rng, n1)
            b = mutate(rng, a, 0.12, 0.04, 0.04)
            b = (b + rand_seq(rng, n2))[:n2] if len(b) < n2 else b[:n2]
            if len(b) < 2:
                b = rand_seq(rng, 2)
        else:
            a, b = rand_seq(rng, n1), rand_seq(rng, n2)
        o = (C.c_int * 4)()
        Hh.ha_extend_flank_both(a, len(a), b, len(b), int(rng.random() < 0.5), o)
        assert (o[0], o[1]) == (o[2], o[3]), (it, len(a), len(b), a, b, list(o))


def run_oracle_wfa(q, t):
    L = O.lib()
    r = O.WfaResult()
    rc = L.lmo_wfa_align(q, len(q), t, len(t), 1, C.byref(r))
    ops = [r.ops[i] for i in range(r.nops)]
    res = (rc, r.score, ops, r.qbegin, r.qend, r.tbegin, r.tend, r.align_len, r.matches, r.gaps, r.gap_regions)
    L.lmo_wfa_result_free(C.byref(r))
    return res


@pytest.mark.parametrize("n,div,seed", [(0, 0, 0), (50, 0.0, 1), (60, 0.3, 2), (300, 0.05, 3), (1500, 0.1, 4),
                                        (1500, 0.25, 5), (4000, 0.12, 6), (200, 0.6, 7)])
def test_wfa_matches_oracle(n, div, seed):
    Hh = H.lib()
    rng = random.Random(seed)
    for rep in range(6 if n <= 300 else 2):
        q = rand_seq(rng, n + rep)
        t = mutate(rng, q, div, div / 4, div / 4)
        if rep == 1 and n:
            t = t[:len(t) // 2] + rand_seq(rng, 40) + t[len(t) // 2:]  # long insertion
        if rep == 2 and n:
            q, t = t, q
        if len(q) == 0 or len(t) == 0:
            q, t = b"ACGT", b"ACGGT"
        exp = run_oracle_wfa(q, t)
        max_score = 64
        arena = 1 << 12
        while True:
            ops = (C.c_uint64 * (len(q) + len(t) + 8))()
            out = H.WfaOut()
            st = Hh.ha_wfa(q, len(q), t, len(t), max_score, arena, ops, len(ops), C.byref(out))
            if st != 1:
                break
            max_score *= 2  # the retry protocol the host pipeline uses on status 1
            arena *= 4
        assert exp[0] == 0 and st in (0, 2)
        got = (0, out.score, [ops[i] for i in range(out.nops)], out.qbegin, out.qend, out.tbegin, out.tend,
               out.align_len, out.matches, out.gaps, out.gap_regions)
        assert got == exp


# ---- packed seed image (lm_seedpack.hip, k_lookup_count): bit streams, in-place partition rewrite, range query -------
@pytest.mark.parametrize("width", [1, 7, 31, 36, 40, 47, 63, 64])
def test_bit_stream_store_and_get_roundtrip(width):
    """elements of any width written partition by partition (full words stored, shared words merged under their mask,
    the way the in-place partition sort writes) read back unchanged, and neighbours are never disturbed"""
    Hh = H.lib()
    rng = random.Random(width)
    n = 1000
    mask = (1 << width) - 1
    elems = [rng.getrandbits(64) & mask for _ in range(n)]
    words = (n * width + 63) // 64 + 2
    stream = (C.c_uint64 * words)()
    # partitions of random sizes (including 1-element ones that share a word with both neighbours), in random order
    cuts = sorted(set([0, n] + [rng.randrange(1, n) for _ in range(120)]))
    parts = list(zip(cuts[:-1], cuts[1:]))
    rng.shuffle(parts)
    for b, e in parts:
        arr = (C.c_uint64 * (e - b))(*elems[b:e])
        Hh.ha_bits_store_range(stream, b, e - b, width, arr)
    assert [Hh.ha_bits_get(stream, i, width) for i in range(n)] == elems
    # rewrite one partition with other values: only its elements change
    b, e = parts[0]
    new = [rng.getrandbits(64) & mask for _ in range(e - b)]
    Hh.ha_bits_store_range(stream, b, e - b, width, (C.c_uint64 * (e - b))(*new))
    exp = elems[:b] + new + elems[e:]
    assert [Hh.ha_bits_get(stream, i, width) for i in range(n)] == exp


def test_partition_range_matches_flat_search():
    """k_lookup_count's search of one sorted partition = all keys in [left, right] (kv-searcher2.go:105-323)"""
    Hh = H.lib()
    rng = random.Random(9)
    kb = 36
    for trial in range(200):
        n = rng.randrange(0, 90)
        keys = sorted(rng.getrandbits(kb) >> rng.choice([0, 0, 20, 30]) for _ in range(n))
        pad_before = rng.randrange(0, 5)
        allk = [rng.getrandbits(kb) for _ in range(pad_before)] + keys + [rng.getrandbits(kb) for _ in range(3)]
        stream = (C.c_uint64 * ((len(allk) * kb + 63) // 64 + 2))()
        Hh.ha_bits_store_range(stream, 0, len(allk), kb, (C.c_uint64 * len(allk))(*allk))
        x = rng.choice(keys) if keys and rng.random() < 0.7 else rng.getrandbits(kb)
        low = (1 << (2 * rng.choice([0, 1, 4, 8, 16]))) - 1
        left, right = x & ~low, x | low
        first = C.c_int64()
        cnt = Hh.ha_partition_range(stream, kb, pad_before, pad_before + n, left, right, C.byref(first))
        exp = [i for i, k in enumerate(keys) if left <= k <= right]
        assert cnt == len(exp)
        if exp:
            assert first.value == pad_before + exp[0]


def test_packed_seed_value_roundtrip():
    Hh = H.lib()
    rng = random.Random(4)
    for _ in range(500):
        pos_bits = rng.randrange(10, 29)
        g = rng.randrange(0, 1 << 18)
        bg = rng.getrandbits(34)
        pos, rc, rv = rng.getrandbits(pos_bits), rng.getrandbits(1), rng.getrandbits(1)
        v64 = (bg << 30) | (pos << 2) | (rc << 1) | rv  # lib-index-build.go:412-455
        pv = Hh.ha_pack_seed_val(g, v64, pos_bits)
        assert pv >> (pos_bits + 1) == g
        assert Hh.ha_unpack_seed_val(pv, bg, pos_bits, rv) == v64
