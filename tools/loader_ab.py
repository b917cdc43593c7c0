"""tools/loader_ab.py [genomes] [genome_len] [shards]: one GPU-built synthetic set written by lm_index_save, then opened by lm_index_open
under several loader settings (environment switches), each open timed; LM_DEBUG=1 prints the loader's own breakdown.  How the
round-6 loader variants were compared on one set of files (the C2-size set: 10000 x 5 Mb = 39 GB of index files)."""
import json
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lexicmap_amd as la  # noqa: E402

genomes = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
glen = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
variants = [dict(), dict()]
t0 = time.time()
gi = la.Index.synthetic(genomes, glen, max(1, genomes // 100), seed=1000)
info = gi.info()
d = os.path.join(tempfile.mkdtemp(prefix="lm_loader_ab_"), "set.lmi")
t1 = time.time()
gi.save(d, chunks=32)
t2 = time.time()
gi.close()
size = sum(os.path.getsize(os.path.join(r, f)) for r, _d, fs in os.walk(d) for f in fs)
out = dict(genomes=genomes, genome_len=glen, seeds=info["seeds"], files_GB=round(size / 1e9, 2), build_s=round(t1 - t0, 1), save_s=round(t2 - t1, 1), opens=[])
for v in variants + [dict(shard=4)]:
    env = {k: val for k, val in v.items() if k != "shard"}
    for k, val in env.items():
        os.environ[k] = val
    opt = la.api.default_options(shard_count=v["shard"], shard_rank=1) if "shard" in v else None
    t = time.time()
    g2 = la.Index(d, opt) if opt is not None else la.Index(d)
    dt = time.time() - t
    print("open returned after %.2f s" % dt, file=sys.stderr, flush=True)
    t = time.time()
    qb = g2.upload([b"ACGT" * 300])   # how long does the first use of the handle wait for the clean-up that runs behind the open ?
    g2.search_resident_np(qb)
    first_use = time.time() - t
    out["opens"].append(dict(variant=v, open_s=round(dt, 2), GBps_of_files=round(size / dt / 1e9, 2), seeds=g2.info()["seeds"], first_search_s=round(first_use, 2)))
    print(out["opens"][-1], file=sys.stderr, flush=True)
    g2.close()
    for k in env:
        os.environ.pop(k, None)
shutil.rmtree(os.path.dirname(d), ignore_errors=True)
print(json.dumps(out))
