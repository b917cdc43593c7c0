"""lm_gather_rows (include/lexicmap_hip.h): the row gather of the sharded search behind the C-ABI, over RCCL.  One GPU here,
so the communicator has ONE rank (RCCL refuses two ranks on one device): the count all-gather runs, the root's own rows come
back in place with the pointer columns cleared, the staging buffers are reused, and the result feeds lm_merge_sharded.  The
multi-rank transfers are exercised by `bench.py --gpus N` on the driver's multi-GPU node; tests/test_merge_gloo.py covers the
N > 1 merge logic on the CPU."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_single_rank_communicator_gathers_counts_and_rows_and_feeds_the_c_merge():
    from lexicmap_amd import merge
    from lexicmap_amd.api import Comm
    uid = Comm.unique_id()
    assert len(uid) == 128
    comm = Comm(uid, 1, 0, device=0)
    try:
        rng = np.random.default_rng(4)
        rows = np.zeros(5000, dtype=merge.ROW_DTYPE)
        qs, gs = rng.integers(0, 50, 5000), rng.integers(0, 1000, 5000)
        o = np.lexsort((gs, qs))   # as a search returns them: grouped by query, a genome's rows together
        rows["query"], rows["batch_genome"] = qs[o], gs[o]
        rows["bitscore"] = rng.integers(50, 3000, 5000)
        rows["pident"] = rng.integers(70, 101, 5000).astype(np.float64)
        rows["genome_id"] = 12345  # a process-local address: cleared by the gather
        expect = rows.copy()
        expect["genome_id"] = 0
        for rep in range(2):   # the second call reuses the staging buffers
            got, counts = comm.gather_rows(rows, root=0)
            assert counts == [5000] and len(got) == 1
            assert merge._cat([got[0]]).tobytes() == merge._cat([expect]).tobytes()
        merged = merge.merge_sharded_c([got[0]])
        assert len(merged) == 5000 and (np.diff(merged["query"].astype(np.int64)) >= 0).all()
        key = lambda a: sorted(zip(a["query"].tolist(), a["batch_genome"].tolist(), a["bitscore"].tolist()))  # noqa: E731
        assert key(merged) == key(rows)                               # the same rows, in the merge's order
        hits = {int(q): len(set(rows["batch_genome"][rows["query"] == q].tolist())) for q in set(rows["query"].tolist())}
        assert all(int(h) == hits[int(q)] for q, h in zip(merged["query"], merged["hits"]))   # global hits per query
        got, counts = comm.gather_rows(rows[:0], root=0)
        assert counts == [0] and len(got[0]) == 0
        with pytest.raises(RuntimeError):
            comm.gather_rows(rows, root=3)   # no such rank: LM_ERR_ARG, nothing hangs
    finally:
        comm.close()


def test_device_merge_of_four_shards_equals_the_host_merge():
    """lm_merge_sharded_device (lm_merge.hip: heads, scan, group keys, rocPRIM merge sort, emit) against lm_merge_sharded on the
    same rows: four shards, 200 queries, genomes with several rows, equal similarities (ties go by genome key), a shard without
    rows, queries that only some shards hit"""
    import torch
    from lexicmap_amd import merge
    from lexicmap_amd.api import Comm
    rng = np.random.default_rng(11)
    shards = []
    for r in range(4):
        n = 0 if r == 2 else 20000
        qs = rng.integers(0, 200, n) * (1 if r != 3 else 2)         # shard 3 only hits even queries
        gs = rng.integers(0, 3000, n) * 4 + r                         # a genome lives in ONE shard
        o = np.lexsort((gs, qs))
        a = np.zeros(n, dtype=merge.ROW_DTYPE)
        a["query"], a["batch_genome"] = qs[o], gs[o]
        a["bitscore"] = rng.integers(1, 40, n) * 50                   # few distinct values: many equal similarities
        a["pident"] = rng.integers(8, 11, n).astype(np.float64) * 10.0
        a["hsp"] = np.arange(n)                                       # tells the rows of a genome apart: their order must be kept
        a["genome_id"] = 777
        shards.append(a)
    want = merge.merge_sharded_c(shards)
    comm = Comm(Comm.unique_id(), 1, 0, device=0)
    try:
        cat = np.concatenate([s.view(np.uint8).reshape(-1) for s in shards])
        dev = torch.from_numpy(cat).cuda()
        torch.cuda.synchronize()
        for _ in range(2):   # the second call reuses every buffer
            got = comm.merge_sharded_device(dev.data_ptr(), [len(s) for s in shards])
            assert len(got) == len(want) == 60000
            for f in merge.ROW_DTYPE.names:
                assert got[f].tobytes() == want[f].tobytes(), f
        assert len(comm.merge_sharded_device(0, [0, 0])) == 0
    finally:
        comm.close()
