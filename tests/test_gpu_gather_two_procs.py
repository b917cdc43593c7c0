"""The MULTI-RANK branch of lm_gather_rows / lm_gather_merge_rows (lexicmap_amd/csrc/lm_comm.cpp) executed: two PROCESSES on the
one GPU of the box, each with its own communicator rank, through the C entry points a Go host would call.  RCCL refuses two
ranks on one device, so the nine nccl* symbols the library binds at run time come from tests/fake_rccl.c (Unix sockets +
hipMemcpy) through LM_RCCL_LIB - everything on the library's side (count all-gather, offsets, grouped receives into the root,
rank-order placement, the device merge, names) is the product's code.

Each rank opens its genome shard (g % 2) of one small index, searches the same batch, and the ranks gather to root 0 and to
root 1; the merged rows must be the unsharded handle's rows (all columns but the process-local pointers), by lm_gather_rows +
lm_merge_sharded and by lm_gather_merge_rows (device merge) alike; a rank with zero rows and the counts on every rank are
checked with synthetic rows."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))

WORKER = r'''
import os, pickle, sys, time
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(here)r)
import numpy as np
import lexicmap_amd as la
from lexicmap_amd import merge
from lexicmap_amd.api import Comm
rank, d, idfile, outfile = int(sys.argv[1]), sys.argv[2], sys.argv[3], sys.argv[4]
if rank == 0:
    uid = Comm.unique_id()
    open(idfile + ".tmp", "wb").write(uid)
    os.rename(idfile + ".tmp", idfile)
else:
    for _ in range(3000):
        if os.path.exists(idfile):
            break
        time.sleep(0.02)
    uid = open(idfile, "rb").read()
comm = Comm(uid, 2, rank, device=0)
queries = pickle.load(open(os.path.join(os.path.dirname(d), "queries.pkl"), "rb"))
gi = la.Index(d, la.api.default_options(shard_count=2, shard_rank=rank))
qb = gi.upload([q[1] for q in queries])
rows, _st = gi.search_resident_np(qb)
rows = rows.copy()
out = {"own": len(rows)}
for root in (0, 1):
    per_rank, counts = comm.gather_rows(rows, root=root)
    out["counts_%%d" %% root] = counts
    if rank == root:
        m, nm = merge.merge_sharded_c(per_rank, index=gi)
        out["host_%%d" %% root] = m.copy()
        out["host_names_%%d" %% root] = [tuple(x.decode() if x else None for x in ab) for ab in nm[:50]]
    m2 = comm.gather_merge_rows(rows, root=root, index=gi)
    if rank == root:
        out["dev_%%d" %% root] = m2.copy()
        out["dev_names_%%d" %% root] = [la.api.row_names(m2, i) for i in range(min(len(m2), 50))]
# synthetic rows: rank 1 has none; then rank 0 has none
rng = np.random.default_rng(7 + rank)
syn = np.zeros(3000, dtype=merge.ROW_DTYPE)
qs, gs = rng.integers(0, 40, 3000), rng.integers(0, 500, 3000) * 2 + rank      # a genome lives in ONE shard
o = np.lexsort((gs, qs))
syn["query"], syn["batch_genome"] = qs[o], gs[o]
syn["bitscore"] = rng.integers(50, 3000, 3000)
syn["pident"] = rng.integers(70, 101, 3000).astype(np.float64)
out["syn"] = syn
for empty in (1, 0):
    mine = syn[:0] if rank == empty else syn
    per_rank, counts = comm.gather_rows(mine, root=0)
    out["syn_counts_%%d" %% empty] = counts
    if rank == 0:
        out["syn_host_%%d" %% empty] = merge.merge_sharded_c(per_rank).copy()
    m2 = comm.gather_merge_rows(mine, root=0)
    if rank == 0:
        out["syn_dev_%%d" %% empty] = m2.copy()
both = comm.gather_merge_rows(syn[:0], root=1)      # nobody has rows
out["nothing"] = None if both is None else len(both)
comm.close()
gi.close()
pickle.dump(out, open(outfile, "wb"))
'''


class _Cols:
    """the columns of a row array without the process-local pointers (and without the padding bytes of the C struct)"""

    def __init__(self, a):
        from lexicmap_amd import merge
        self.c = [(f, np.ascontiguousarray(a[f]).tobytes()) for f in merge.ROW_DTYPE.names if f not in merge.PTR_FIELDS]

    def tobytes(self):
        return self.c


def _strip(a):
    return _Cols(a)


@pytest.fixture(scope="module")
def fake_rccl(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("fake") / "libfake_rccl.so")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O2", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(HERE, "fake_rccl.c"), "-o", so,
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
    return so


def test_two_ranks_through_the_c_gather_and_the_device_merge_give_the_unsharded_rows(tmp_path, fake_rccl):
    import oracle as O
    import lexicmap_amd as la
    from lexicmap_amd import merge, synth
    genomes = synth.make_genomes(12, 70_000, 3, seed=5, max_div=0.06, contigs=(1, 2))
    queries = synth.make_gene_queries(genomes, 24, seed=6, len_range=(500, 1600), max_div=0.08)
    d = str(tmp_path / "two.lmi")
    O.build_index(d, genomes, O.default_build_opt(chunks=2))
    pickle.dump(queries, open(str(tmp_path / "queries.pkl"), "wb"))
    gi = la.Index(d)
    ref, _ = gi.search_resident_np(gi.upload([q[1] for q in queries]))
    ref = ref.copy()
    ref_names = [la.api.row_names(ref, i) for i in range(min(len(ref), 50))]
    gi.close()
    assert len(ref) > 40
    script = str(tmp_path / "worker.py")
    open(script, "w").write(WORKER % dict(root=ROOT, here=HERE))
    env = dict(os.environ, LM_RCCL_LIB=fake_rccl, LM_FAKE_RCCL_DIR=str(tmp_path))
    procs = [subprocess.Popen([sys.executable, script, str(r), d, str(tmp_path / "id.bin"), str(tmp_path / ("out%d.pkl" % r))], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in (0, 1)]
    outs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, se[-3000:]
    for r in (0, 1):
        outs.append(pickle.load(open(str(tmp_path / ("out%d.pkl" % r)), "rb")))
    own = [outs[0]["own"], outs[1]["own"]]
    assert own[0] > 0 and own[1] > 0 and own[0] + own[1] == len(ref)
    for root in (0, 1):
        for r in (0, 1):
            assert outs[r]["counts_%d" % root] == own                     # every rank's count on every rank
        host, dev = outs[root]["host_%d" % root], outs[root]["dev_%d" % root]
        assert _strip(host).tobytes() == _strip(ref).tobytes()               # lm_gather_rows + lm_merge_sharded = the unsharded rows
        assert _strip(dev).tobytes() == _strip(ref).tobytes()                # lm_gather_merge_rows (device merge) likewise
        assert outs[root]["host_names_%d" % root] == ref_names == outs[root]["dev_names_%d" % root]
        assert "dev_%d" % root not in outs[1 - root]                         # nothing comes back on the rank that does not merge
    # one rank without rows (either one), the merge of what is left = the host merge of the same rows
    syn = [outs[0]["syn"], outs[1]["syn"]]
    for empty in (1, 0):
        want = [0 if r == empty else 3000 for r in (0, 1)]
        assert outs[0]["syn_counts_%d" % empty] == want == outs[1]["syn_counts_%d" % empty]
        host, dev = outs[0]["syn_host_%d" % empty], outs[0]["syn_dev_%d" % empty]
        assert len(host) == 3000 and _strip(host).tobytes() == _strip(dev).tobytes()
        exp = merge.merge_sharded_c([syn[r][:0] if r == empty else syn[r] for r in (0, 1)])
        assert _strip(exp).tobytes() == _strip(dev).tobytes()
    assert outs[1]["nothing"] == 0 and outs[0]["nothing"] is None
