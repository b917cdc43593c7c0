"""debug helper: run a few WFA problems through lm_wfa_batch one at a time (each in its own short-lived process)"""
import ctypes as C
import os, subprocess, sys
import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = ["ident", "mm1", "ins", "del", "rand5", "rand10", "batch"]


def make(case):
    import lexicmap_amd.synth as S
    rng = np.random.default_rng(5)
    a = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 1500))
    if case == "ident":
        return [(a, a)]
    if case == "mm1":
        b = bytearray(a); b[700] = ord("A") if a[700] != ord("A") else ord("C")
        return [(a, bytes(b))]
    if case == "ins":
        return [(a, a[:500] + b"ACGTTGCA" + a[500:])]
    if case == "del":
        return [(a, a[:500] + a[520:])]
    if case == "rand5":
        return [(a, S.mutate(rng, np.frombuffer(a, dtype=np.uint8), 0.05, 0.01, 0.01).tobytes())]
    if case == "rand10":
        return [(a, S.mutate(rng, np.frombuffer(a, dtype=np.uint8), 0.10, 0.02, 0.02).tobytes())]
    out = []
    for i in range(300):
        x = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), int(rng.integers(200, 2000))))
        out.append((x, S.mutate(rng, np.frombuffer(x, dtype=np.uint8), float(rng.uniform(0, 0.12)), 0.01, 0.01).tobytes()))
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1:
        import lexicmap_amd as la
        sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
        import oracle as O
        gi = la.Index.synthetic(genomes=2, genome_len=50_000, families=1, seed=1, max_div=0.05)
        pairs = make(sys.argv[1])
        res = gi.wfa(pairs)
        bad = 0
        for (q, t), r in zip(pairs, res):
            o = O.WfaResult()
            O.lib().lmo_wfa_align(q, len(q), t, len(t), 1, C.byref(o))
            same = (r["score"], r["ops"]) == (o.score, [o.ops[i] for i in range(o.nops)])
            bad += not same
        print(sys.argv[1], "n", len(pairs), "mismatching", bad, "first", {k: res[0][k] for k in ("status", "score", "align_len", "matches", "gaps")}, flush=True)
        gi.close()
    else:
        for c in CASES:
            try:
                p = subprocess.run([sys.executable, __file__, c], timeout=40, capture_output=True, text=True)
                print(c, "rc", p.returncode, p.stdout.strip()[-300:], p.stderr.strip()[-300:], flush=True)
            except subprocess.TimeoutExpired:
                print(c, "TIMEOUT", flush=True)
