"""wfa_lean2_gpu_check.py - the forced-path check of k_wfa_lean2 (lexicmap_amd/csrc/lm_wfa_lean2.h): every
instantiation forced through lm_wfa_batch (la.Index.wfa) against the oracle's lmo_wfa_align, plus the time of two class-shaped
batches.  tests/test_gpu_wfa_lean2.py runs main(timing=False).

    python tests/wfa_lean2_gpu_check.py            (on the GPU box, from the repository root: with the timing batches)

Pairs: gene-sized (<= 2 kb: 128 diagonals, 16-bit cells), 2-8 kb (128 / 256 diagonals, 16-bit cells up to 12 000 bases), 8-32
kb (windowed, 256 diagonals), 32-65 kb (whole sequences, 512 / 1024 diagonals), beyond 65 kb (windowed);
divergence 1-15 %; one-sided indels (the ring is recentred); length differences that outgrow 128 / 256 diagonals (status 3 ->
next width); LM_WFA_FIRST_NC=1,1,1,1,1 to run the 64-diagonal kernels too."""
import ctypes as C
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle as O  # noqa: E402
import lexicmap_amd as la  # noqa: E402
from lexicmap_amd import synth  # noqa: E402


def pair(rng, n, sub, ins, dele, extra=0):
    q = synth.random_seq(rng, n)
    t = synth.mutate(rng, q, sub=sub, ins=ins, dele=dele)
    if extra > 0:
        t = np.concatenate([t, synth.random_seq(rng, extra)])
    elif extra < 0:
        q = np.concatenate([q, synth.random_seq(rng, -extra)])
    return q.tobytes(), t.tobytes()


def check(pairs, got):
    L = O.lib()
    bad = 0
    for i, ((q, t), g) in enumerate(zip(pairs, got)):
        r = O.WfaResult()
        assert L.lmo_wfa_align(q, len(q), t, len(t), 1, C.byref(r)) == 0
        ok = (g["status"] in (0, 2) and g["score"] == r.score and g["ops"] == [r.ops[j] for j in range(r.nops)] and
              (g["qbegin"], g["qend"], g["tbegin"], g["tend"], g["align_len"], g["matches"], g["gaps"], g["gap_regions"]) ==
              (r.qbegin, r.qend, r.tbegin, r.tend, r.align_len, r.matches, r.gaps, r.gap_regions))
        if not ok:
            bad += 1
            print("DIFFERENT", i, len(q), len(t), g["status"], g["score"], r.score)
        L.lmo_wfa_result_free(C.byref(r))
    return bad


def main(timing=True):
    rng = np.random.default_rng(5)
    d = os.path.join(tempfile.mkdtemp(), "t.lmi")
    O.build_index(d, synth.make_genomes(2, 60000, 1, seed=3, max_div=0.05), O.default_build_opt(chunks=2))
    pairs = []
    for n in (300, 900, 1500, 1900):
        for dv in (0.01, 0.05, 0.10, 0.15):
            pairs.append(pair(rng, n, dv, dv / 3, dv / 3))
    for n in (2500, 5000, 7800, 11000):
        for dv in (0.02, 0.07, 0.12):
            pairs.append(pair(rng, n, dv, dv / 2, dv / 2))
    pairs += [pair(rng, 6000, 0.02, 0.0, 0.06), pair(rng, 6000, 0.02, 0.06, 0.0), pair(rng, 1800, 0.03, 0.0, 0.08)]   # drift: recentres
    pairs += [pair(rng, 3000, 0.05, 0.02, 0.02, extra=150), pair(rng, 3000, 0.05, 0.02, 0.02, extra=-300)]            # outgrow 128 / 256
    pairs += [pair(rng, 12000, 0.02, 0.02, 0.03), pair(rng, 25000, 0.02, 0.02, 0.03), pair(rng, 30000, 0.03, 0.01, 0.05)]  # windowed
    pairs += [pair(rng, 40000, 0.02, 0.02, 0.03), pair(rng, 70000, 0.02, 0.02, 0.03)]
    # the wide passes: final diagonals 300-700 away - 512 / 1024 diagonals, windowed (8-32 kb) and whole (32-65 kb)
    pairs += [pair(rng, 20000, 0.02, 0.02, 0.02, extra=400), pair(rng, 26000, 0.02, 0.02, 0.02, extra=-450), pair(rng, 30000, 0.02, 0.02, 0.02, extra=650),
              pair(rng, 36000, 0.02, 0.02, 0.02, extra=380), pair(rng, 44000, 0.02, 0.02, 0.02, extra=-640)]
    total_bad = 0
    report = {}
    for env in ({}, {"LM_WFA_FIRST_NC": "1,1,1,1,1"}, {"LM_WFA_R16": "0"}, {"LM_WFA_WIN": "11111"}, {"LM_WFA_WIN": "00000"}):
        for k in ("LM_WFA_FIRST_NC", "LM_WFA_R16", "LM_WFA_WIN"):
            os.environ.pop(k, None)
        os.environ.update(env)
        gi = la.Index(d)
        gi.profile(True)
        got = gi.wfa(pairs)
        names = {p["name"]: p["launches"] for p in gi.profile_get() if p["name"].startswith("k_wfa")}
        bad = check(pairs, got)
        total_bad += bad
        report[json.dumps(env, sort_keys=True)] = {"different": bad, "kernels": names}
        print(env, "different:", bad, names)
        gi.close()
    # time: 8192 gene-sized pairs and 512 5-kb pairs, k_wfa_lean2
    timing_s = {}
    for label, batch in () if not timing else (("genes_1500bp_x8192", [pair(rng, 1500, 0.05, 0.02, 0.02) for _ in range(256)] * 32),
                         ("reads_5kb_x512", [pair(rng, 5000, 0.03, 0.02, 0.03) for _ in range(64)] * 8)):
        for k in ("LM_WFA_FIRST_NC", "LM_WFA_R16", "LM_WFA_WIN"):
            os.environ.pop(k, None)
        gi = la.Index(d)
        gi.wfa(batch[:64])
        t0 = time.time()
        gi.wfa(batch)
        timing_s[label] = round(time.time() - t0, 4)
        gi.close()
    for k in ("LM_WFA_FIRST_NC", "LM_WFA_R16", "LM_WFA_WIN"):
        os.environ.pop(k, None)
    print(json.dumps({"report": report, "seconds": timing_s}, indent=1))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump({"report": report, "seconds": timing_s}, open("gpurun_out/r06_wfa_lean2_check.json", "w"), indent=1)
    return total_bad, report


if __name__ == "__main__":
    sys.exit(1 if main()[0] else 0)
