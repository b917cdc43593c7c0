"""N>1 path on CPU: two gloo ranks all-gather their HSP row records and merge them with the reference's final ordering
(lexicmap_amd/merge.py); the result must equal the single-process merge of the same rows."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rows(seed, genomes, nq=5):
    from lexicmap_amd.merge import ROW_DTYPE
    rng = np.random.default_rng(seed)
    out = []
    for q in range(nq):
        for g in genomes:
            if rng.random() < 0.3:
                continue
            nh = int(rng.integers(1, 4))
            for h in range(nh):
                r = np.zeros(1, dtype=ROW_DTYPE)
                r["query"], r["batch_genome"], r["hits"] = q, g, 0
                r["cls"], r["hsp"] = h + 1, h + 1
                r["bitscore"] = int(rng.integers(50, 3000)) if h == 0 else int(rng.integers(50, 200))
                r["pident"] = float(rng.integers(70, 101))
                r["qbegin"], r["qend"] = 0, 99
                out.append(r)
    from lexicmap_amd.merge import _cat
    return _cat(out)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from lexicmap_amd import merge
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = _rows(100 + rank, [g for g in range(40) if g % world == rank])
    per_rank = merge.all_gather_rows(mine)
    merged = merge.merge_sharded(per_rank)
    # every rank must hold the identical merged table
    chk = merged.tobytes()
    gathered = [None] * world
    dist.all_gather_object(gathered, chk)
    assert all(g == chk for g in gathered)
    if rank == 0:
        q.put(merged.tobytes())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_merge_equals_single_process():
    sys.path.insert(0, ROOT)
    from lexicmap_amd import merge
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    shards = [_rows(100 + r, [g for g in range(40) if g % 2 == r]) for r in range(2)]
    exp = merge.merge_sharded(shards)
    assert got == exp.tobytes()
    # ordering rule: per query, genomes by best bitscore*pident descending; hits = number of genomes
    for qq in np.unique(exp["query"]):
        rq = exp[exp["query"] == qq]
        gs = []
        for g in rq["batch_genome"]:
            if not gs or gs[-1] != g:
                gs.append(int(g))
        assert len(set(gs)) == len(gs)
        best = [max(float(b) * float(p) for b, p in zip(rq["bitscore"][rq["batch_genome"] == g], rq["pident"][rq["batch_genome"] == g])) for g in gs]
        assert best == sorted(best, reverse=True)
        assert set(rq["hits"]) == {len(gs)}


# ---- real search output through the gather + the library's C merge ------------------------------------------------------
def _oracle_rows():
    """HSP rows of a small oracle-built index for a few queries, as lm_hsp records (CPU only: the oracle is the search
    engine here, the thing under test is the gather and lm_merge_sharded)"""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    from lexicmap_amd import merge, synth
    genomes = synth.make_genomes(10, 60000, 2, seed=41, max_div=0.10, contigs=(1, 2))
    queries = synth.make_gene_queries(genomes, 8, seed=42, len_range=(500, 1500), max_div=0.10)
    with tempfile.TemporaryDirectory() as tmp:
        d = os.path.join(tmp, "m.lmi")
        O.build_index(d, genomes, O.default_build_opt(chunks=2))
        oi = O.Index(d)
        rows = []
        for qi, (_, s) in enumerate(queries):
            rr, st = oi.search(s)
            for r in rr:
                r = dict(r)
                r["query"] = qi
                r["hits"] = st["ngenomes"]
                rows.append(r)
        oi.close()
    return merge.pack_rows(rows)


def _shard_of(rows, rank, world):
    """what rank `rank` of a genome-sharded index would report: its genomes' rows, `hits` counted over its own genomes"""
    g = (rows["batch_genome"] >> 17) * 5000 + (rows["batch_genome"] & 0x1ffff)  # dense genome number (one batch here)
    mine = rows[(g % world) == rank].copy()
    for q in np.unique(mine["query"]):
        m = mine["query"] == q
        mine["hits"][m] = len(np.unique(mine["batch_genome"][m]))
    return mine


def _worker_c(rank, world, port, q, blob):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from lexicmap_amd import merge
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = np.frombuffer(blob, dtype=merge.ROW_DTYPE)
    per_rank = merge.all_gather_rows(_shard_of(rows, rank, world), host_on=0)
    if rank == 0:
        q.put(merge.merge_sharded_c(per_rank).tobytes())
    dist.barrier()
    dist.destroy_process_group()


def test_search_output_of_two_shards_through_the_c_merge_equals_the_unsharded_rows():
    """world 2, gloo: every rank holds the rows of its genome shard (real search output, from the oracle), one all-gatherv
    (merge.all_gather_rows), then the library's lm_merge_sharded on rank 0: the unsharded rows in the unsharded order
    with the unsharded `hits`"""
    sys.path.insert(0, ROOT)
    from lexicmap_amd import merge
    rows = _oracle_rows()
    assert len(rows) > 20 and len(np.unique(rows["batch_genome"])) > 3
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_c, args=(r, 2, port, q, rows.tobytes())) for r in range(2)]
    for p in procs:
        p.start()
    got = np.frombuffer(q.get(timeout=180), dtype=merge.ROW_DTYPE)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert len(got) == len(rows)
    for f in merge.ROW_DTYPE.names:
        if f in ("genome_id", "seq_id", "cigar", "qseq", "sseq", "align"):
            continue
        assert (got[f] == rows[f]).all(), f
