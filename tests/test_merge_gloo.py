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
