"""tools/wfa_counter_probe.py: batches of IDENTICAL pairs through lm_wfa_batch so that the SQ counters of one k_wfa_lean2 launch can
be divided by a known number of alignments and score steps.  Run under rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS:
    rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d /tmp/p -o p -- python tools/wfa_counter_probe.py
prints one JSON line per batch: length, divergence, alignments, score (steps = score / 2); the dispatch order is the batch order."""
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import oracle as O  # noqa: E402
import lexicmap_amd as la  # noqa: E402
from lexicmap_amd import synth  # noqa: E402

rng = np.random.default_rng(9)
d = os.path.join(tempfile.mkdtemp(), "t.lmi")
O.build_index(d, synth.make_genomes(2, 60000, 1, seed=3, max_div=0.05), O.default_build_opt(chunks=2))
gi = la.Index(d)
gi.wfa([(b"ACGT" * 100, b"ACGT" * 100)] * 64)   # warm-up (first-use costs)
out = []
for n, sub, ins, dele in ((4000, 0.0, 0.0, 0.0), (4000, 0.02, 0.0, 0.0), (4000, 0.06, 0.02, 0.03), (4000, 0.10, 0.03, 0.04), (1500, 0.06, 0.02, 0.03)):
    q = synth.random_seq(rng, n)
    t = synth.mutate(rng, q, sub=sub, ins=ins, dele=dele)
    pairs = [(q.tobytes(), t.tobytes())] * 16384
    got = gi.wfa(pairs)
    assert all(g["status"] == 0 and g["score"] == got[0]["score"] for g in got)
    rec = dict(qlen=len(q), tlen=len(t), sub=sub, ins=ins, dele=dele, alignments=len(pairs), score=got[0]["score"], steps=got[0]["score"] // 2,
               ops=len(got[0]["ops"]))
    out.append(rec)
    print(json.dumps(rec), flush=True)
gi.close()
