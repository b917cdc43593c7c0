import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import lexicmap_amd as la
rng = np.random.default_rng(5)
a = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 1500))
gi = la.Index.synthetic(genomes=2, genome_len=50_000, families=1, seed=1, max_div=0.05)
print("built", flush=True)
r = gi.wfa([(a, a)])
print({k: r[0][k] for k in ("status", "score", "qbegin", "qend", "align_len", "matches")}, len(r[0]["ops"]), flush=True)
