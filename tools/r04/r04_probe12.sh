#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
LM_DEBUG=1 timeout 600 python -m pytest tests/test_gpu_builder.py -m gpu -x -q -k "saved_index" > gpurun_out/r04_dbg_builder.log 2>&1; tail -3 gpurun_out/r04_dbg_builder.log
grep -E "pseudo-alignment:|loader:|uneven|chunk halved|Error|error" gpurun_out/r04_dbg_builder.log | cut -c1-260 | tail -40
python - <<'PY'
import sys, os
sys.path.insert(0, "tests")
import numpy as np
import lexicmap_amd as la
gi = la.Index.synthetic(12, 200000, 3, seed=5, masks=20000)
d = "/tmp/saved_dbg.lmi"
import shutil; shutil.rmtree(d, ignore_errors=True)
gi.save(d, chunks=5)
li = la.Index(d)
bad = 0
for g in range(12):
    a = gi.fetch(g, 0, 200000); b = li.fetch(g, 0, 200000)
    if a != b:
        bad += 1
        x = np.frombuffer(a, np.uint8); y = np.frombuffer(b, np.uint8)
        print("genome", g, "differs at", int(np.argmax(x != y)), "of", len(a), "ndiff", int((x != y).sum()))
print("genomes differing:", bad, gi.info()["genomes"], li.info()["genomes"])
seqs = [gi.fetch(4, 50_000, 1500), gi.fetch(2, 10, 800)]
try:
    r2, _ = li.search(seqs); print("loaded search rows", len(r2))
except Exception as e: print("loaded search failed:", e)
r1, _ = gi.search(seqs); print("built search rows", len(r1))
PY
