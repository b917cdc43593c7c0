"""tools/stress_emu_wfa_lean2.py [seconds] [seed]: random pairs of every shape through the staged forward passes on the
host SIMT emulator (single wavefront: 64-512 diagonals, 16- / 32-bit cells, whole / windowed) against the oracle, for as long as asked.  Not part of the test suite (the suite runs fixed cases of the same
harnesses); run before adopting:  python tools/stress_emu_wfa_lean2.py 600"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from test_device_algos_cpu import mutate, rand_seq, run_oracle_wfa  # noqa: E402
import test_wfa_lean2_emulated_cpu as L2  # noqa: E402


def make_pair(rng):
    shape = rng.randint(0, 6)
    n = rng.choice((40, 200, 700, 1500, 3000, 5000))
    if shape >= 5:
        n = rng.choice((4500, 6000, 9000))  # beyond one window
    q = rand_seq(rng, rng.randint(max(1, n // 2), n))
    d = rng.choice((0.0, 0.01, 0.03, 0.07, 0.12, 0.2, 0.35))
    if shape == 0:
        t = mutate(rng, q, d, d / 3, d / 3)
    elif shape == 1:  # one-sided indels: drift
        t = mutate(rng, q, d / 2, 0.0, min(0.08, d))
    elif shape == 2:
        t = mutate(rng, q, d / 2, min(0.08, d), 0.0)
    elif shape == 3:  # end gap on either side
        t = mutate(rng, q, d, d / 3, d / 3)
        e = rand_seq(rng, rng.randint(1, 300))
        if rng.random() < 0.5:
            t = t + e
        else:
            q = q + e
    elif shape == 4:  # wandering: deletions then insertions
        h = len(q) // 2
        t = mutate(rng, q[:h], 0.02, 0.0, 0.05) + mutate(rng, q[h:], 0.02, 0.05, 0.0)
    else:
        t = mutate(rng, q, min(d, 0.1), 0.02, 0.03)
    return q, (t if t else b"A")


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    t0 = time.time()
    n = nl = nm = nrec = nwide = 0
    while time.time() - t0 < budget:
        q, t = make_pair(rng)
        exp = run_oracle_wfa(q, t)
        if exp[0] != 0:
            continue
        n += 1
        # single wavefront
        nc = rng.choice((1, 2, 4, 8))
        win = rng.random() < 0.4
        r16 = (not win) and nc in (2, 4) and rng.random() < 0.6 and max(len(q), len(t)) <= 12000
        while True:
            st, got, rec = L2.run1(q, t, nc, r16, win=win)
            if st != 3 or nc >= 8:
                break
            nwide += 1
            nc *= 2
            r16 = r16 and nc <= 4
        if st == 0:
            assert got == exp, ("lean2", len(q), len(t), nc, r16, win)
            nl += 1
            nrec += rec
        else:
            assert st == 3 and nc >= 8, ("lean2 status", st, len(q), len(t), nc)
    print("pairs %d, single-wavefront alignments equal %d, recentres %d, width retries %d, %.0f s" % (n, nl, nrec, nwide, time.time() - t0))


main()
