"""Seeded synthetic genomes / queries: i.i.d. ACGT ancestors, families of mutated descendants,
queries = mutated substrings (genes) or ONT-like reads.  numpy only; data generation, no search logic."""
import numpy as np

_B = np.frombuffer(b"ACGT", dtype=np.uint8)


def random_seq(rng, n):
    return _B[rng.integers(0, 4, size=n)]


def mutate(rng, s, sub=0.0, ins=0.0, dele=0.0):
    """s: uint8 array of ACGT. returns mutated uint8 array"""
    n = len(s)
    if n == 0:
        return s.copy()
    out = s.copy()
    if sub > 0:
        m = rng.random(n) < sub
        k = int(m.sum())
        if k:
            # substitute with a different base
            cur = np.searchsorted(_B, out[m])
            out[m] = _B[(cur + rng.integers(1, 4, size=k)) % 4]
    if ins <= 0 and dele <= 0:
        return out
    keep = rng.random(n) >= dele if dele > 0 else np.ones(n, dtype=bool)
    nins = (rng.random(n) < ins) if ins > 0 else np.zeros(n, dtype=bool)
    # build: each kept base, optionally followed by one inserted base
    reps = keep.astype(np.int64) + nins.astype(np.int64)
    idx = np.repeat(np.arange(n), reps)
    res = out[idx]
    # positions that are insertions: the last copy of an index with nins
    if nins.any():
        ends = np.cumsum(reps) - 1
        ins_pos = ends[nins & (reps > 0)]
        res[ins_pos] = _B[rng.integers(0, 4, size=len(ins_pos))]
    return res


def make_genomes(n_genomes, genome_len, n_families, seed, max_div=0.10, contigs=(1, 1), with_n=False):
    """returns list of (genome_id, [(contig_id, bytes), ...]) — family members are mutated copies of an ancestor
    (substitutions U(0,max_div), indels at a tenth of that rate), split into 1..k contigs."""
    rng = np.random.default_rng(seed)
    n_families = max(1, min(n_families, n_genomes))
    ancestors = [random_seq(rng, genome_len) for _ in range(n_families)]
    genomes = []
    for g in range(n_genomes):
        fam = g % n_families
        d = rng.random() * max_div if g >= n_families else 0.0
        s = mutate(rng, ancestors[fam], sub=d, ins=d / 20, dele=d / 20)
        if with_n and g % 5 == 3 and len(s) > 2000:
            p = int(rng.integers(500, len(s) - 600))
            s = s.copy()
            s[p:p + 37] = ord("N")
        nc = int(rng.integers(contigs[0], contigs[1] + 1))
        cuts = sorted(set([0, len(s)] + [int(x) for x in rng.integers(200, max(201, len(s) - 200), size=nc - 1)]))
        parts = [(("g%05d_c%d" % (g, i)), s[a:b].tobytes()) for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:]))
                 if b - a >= 31]
        genomes.append(("GCF_%07d.1" % g, parts))
    return genomes


def make_gene_queries(genomes, n, seed, len_range=(1000, 2000), max_div=0.10):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        gid, contigs = genomes[int(rng.integers(0, len(genomes)))]
        cid, s = contigs[int(rng.integers(0, len(contigs)))]
        L = int(rng.integers(len_range[0], len_range[1] + 1))
        L = min(L, len(s))
        st = int(rng.integers(0, len(s) - L + 1))
        q = np.frombuffer(s, dtype=np.uint8)[st:st + L]
        q = np.where(q == ord("N"), ord("A"), q).astype(np.uint8)
        d = rng.random() * max_div
        q = mutate(rng, q, sub=d, ins=d / 10, dele=d / 10)
        if rng.random() < 0.5:
            comp = {65: 84, 67: 71, 71: 67, 84: 65}
            q = np.array([comp[int(c)] for c in q[::-1]], dtype=np.uint8)
        out.append(("q%05d_%s" % (i, gid), q.tobytes()))
    return out


def make_reads(genomes, n, seed, len_range=(5000, 50000), sub=0.02, ins=0.02, dele=0.03):
    """ONT-like long reads with log-uniform lengths"""
    rng = np.random.default_rng(seed)
    out = []
    lo, hi = np.log(len_range[0]), np.log(len_range[1])
    for i in range(n):
        gid, contigs = genomes[int(rng.integers(0, len(genomes)))]
        cid, s = max(contigs, key=lambda c: len(c[1]))
        L = int(np.exp(rng.uniform(lo, hi)))
        L = min(L, len(s))
        st = int(rng.integers(0, len(s) - L + 1))
        q = np.frombuffer(s, dtype=np.uint8)[st:st + L]
        q = np.where(q == ord("N"), ord("A"), q).astype(np.uint8)
        q = mutate(rng, q, sub=sub, ins=ins, dele=dele)
        out.append(("r%05d_%s" % (i, gid), q.tobytes()))
    return out
