"""The RCCL leg of the hit-list merge on one GPU: a single-rank `nccl` process group runs the same all-gatherv code the
multi-GPU bench uses (device tensors, pinned host staging); the gathered rows must equal what went in."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_single_rank_rccl_all_gather_rows_roundtrip():
    import torch
    import torch.distributed as dist
    from lexicmap_amd import merge
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        rng = np.random.default_rng(4)
        rows = np.zeros(5000, dtype=merge.ROW_DTYPE)
        rows["query"] = rng.integers(0, 50, 5000)
        rows["batch_genome"] = rng.integers(0, 1000, 5000)
        rows["bitscore"] = rng.integers(50, 3000, 5000)
        rows["pident"] = rng.integers(70, 101, 5000).astype(np.float64)
        rows["genome_id"] = 12345  # a process-local address: means nothing on another rank - every consumer clears it
        for rep in range(2):       # the second call reuses the pinned staging buffer
            got = merge.all_gather_rows(rows, device="cuda", host_on=0)
            assert len(got) == 1 and len(got[0]) == 5000
            assert merge._cat([got[0]]).tobytes() == merge._cat([rows]).tobytes()
        merged = merge.merge_sharded(got)
        assert len(merged) == 5000 and (np.diff(merged["query"].astype(np.int64)) >= 0).all()
        order = np.lexsort((np.arange(5000), rows["query"]))
        merged_c = merge.merge_sharded_c([rows[order]])       # (the C merge wants each shard's rows grouped by query)
        for out in (merged, merged_c, merge.merge_query_sharded(got)):
            assert len(out) == 5000 and all((out[f] == 0).all() for f in merge.PTR_FIELDS)
        empty = merge.all_gather_rows(rows[:0], device="cuda", host_on=0)
        assert len(empty) == 1 and len(empty[0]) == 0
    finally:
        dist.destroy_process_group()
