#!/usr/bin/env python3
"""bench.py — queries/s of the MI355X `lexicmap search` hot path (BASELINE.json metric), one JSON line on rank 0.

A "step" = one pass of the whole hot path (mask -> lookup -> chain -> pseudo-align -> extend/WFA -> HSP rows) over one
batch of synthetic queries whose bases are already resident in HBM.  N>1: one process per GPU; the index is replicated,
the query batch is sharded (strong scaling: total work fixed) and per-rank HSP rows are merged with one RCCL all-gather
per step.  `--shard index` shards the genomes instead (index larger than one GPU).

The CPU baseline is the oracle (oracle/liblmo.so, a C restatement of the reference; kind "port") timed on the host cores
on a bounded sample of the same workload.  The Go reference cannot be built here (`go` absent) — probed and printed.
"""
import argparse
import json
import math
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--workload", default=os.environ.get("LM_BENCH_WORKLOAD", "c2"), choices=["c2", "small", "tiny", "c3mini"])
    p.add_argument("--queries", type=int, default=0, help="override the number of queries")
    p.add_argument("--genomes", type=int, default=0, help="override the number of genomes")
    p.add_argument("--genome-len", type=int, default=0)
    p.add_argument("--shard", default="queries", choices=["queries", "index"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0)
    p.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                   help="N>1 with --shard queries: weak = every GPU searches its own batch of the workload's size (the "
                        "queries are independent units: no data-path collective, only the final row gather); strong = "
                        "the one batch is divided over the GPUs")
    p.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                   help="nccl = RCCL (one GPU per rank); gloo only to exercise the N>1 logic with ranks sharing a GPU")
    p.add_argument("--builder", default=None, choices=["gpu", "oracle"],
                   help="gpu: genomes+index generated in HBM (default for c2); oracle: CPU writer + on-disk format")
    p.add_argument("--cpu-sample-genomes", type=int, default=12)
    p.add_argument("--cpu-sample-queries", type=int, default=512)
    return p.parse_args()


WORKLOADS = {
    # BASELINE.json configs[1]: 1k gene queries (1-2 kb) vs 10k synthetic 5-Mb genomes, index HBM-resident
    "c2": dict(genomes=10000, genome_len=5_000_000, families=100, queries=1000, qlen=(1000, 2000)),
    # scaled-down shapes for development (NOT the headline config)
    "small": dict(genomes=200, genome_len=500_000, families=4, queries=1000, qlen=(1000, 2000)),
    "tiny": dict(genomes=24, genome_len=100_000, families=4, queries=64, qlen=(300, 1500)),
    # shape of BASELINE.json configs[2] (ONT-style long reads, chaining + WFA stress) at a size that builds in seconds;
    # a robustness run, not a headline number
    "c3mini": dict(genomes=2000, genome_len=5_000_000, families=40, queries=500, qlen=(5000, 50000)),
}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/r01_c2_pmc_{fetch,write}.json,
    two separate --pmc runs of this same workload): (FETCH_SIZE + WRITE_SIZE) * 1024. On gfx950 FETCH_SIZE tallies
    128-B read requests as 64 B for wide streaming reads (MI355X_MICROARCH.md, HBM), so the read part is a lower
    bound (at most 2x low). Returns (bytes or None, note)."""
    here = os.path.dirname(os.path.abspath(__file__))
    tot, found = 0.0, False
    for fn, ctr in (("r01_c2_pmc_fetch.json", "FETCH_SIZE"), ("r01_c2_pmc_write.json", "WRITE_SIZE")):
        try:
            pm = json.load(open(os.path.join(here, "profiles", fn))).get("pmc", {})
        except (OSError, ValueError):
            return None, "no committed PMC pass"
        best = None
        for k, v in pm.items():  # rocprof reports template kernels without the bench's suffix (k_wfa_lean vs k_wfa_lean128)
            if ctr in v and (kernel.startswith(k) or k.startswith(kernel)) and (best is None or len(k) > len(best[0])):
                best = (k, v[ctr])
        if best is None:
            return None, "kernel not in the committed PMC pass"
        tot += best[1]["mean"] * 1024.0
        found = True
    return (int(tot) if found else None), "(FETCH_SIZE+WRITE_SIZE)*1024 from profiles/r01_c2_pmc_*.json; read part is a lower bound on gfx950"


def usable_cores():
    """cores this process may really use: the CPU count capped by the affinity mask and the cgroup quota"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(index_dir, queries, seconds, ncores):
    """oracle (port) on the host cores, bounded sample; returns dict"""
    import multiprocessing as mp
    n = len(queries)
    # calibrate on a few queries with one core
    import oracle as O
    oi = O.Index(index_dir)
    t0 = time.time()
    k = 0
    while k < min(n, 4):
        oi.search(queries[k][1])
        k += 1
    per = (time.time() - t0) / max(k, 1)
    oi.close()
    # enough searches for ~`seconds` of wall time on all cores (the batch is repeated when it is too short for that)
    nsample = int(max(ncores, min(40 * n, seconds * ncores / max(per, 1e-4))))
    sample = [queries[i % n] for i in range(nsample)]
    with mp.Pool(ncores, initializer=_cpu_init, initargs=(index_dir,)) as pool:
        t0 = time.time()
        res = pool.map(_cpu_one, [q[1] for q in sample], chunksize=max(1, len(sample) // (ncores * 8)))
        dt = time.time() - t0
    rows = sum(r[0] for r in res)
    return dict(value=len(sample) / dt, unit="queries/s", cores=ncores, kind="port",
                sample="%d searches over the %d sample queries of the same batch, oracle/liblmo.so (C restatement of the Go reference, "
                       "RAM-resident index = the reference's -w regime) on %d processes, %.1f s, %d HSP rows"
                       % (len(sample), n, ncores, dt, rows))


_CPU_IDX = None


def _cpu_init(index_dir):
    global _CPU_IDX
    import oracle as O
    _CPU_IDX = O.Index(index_dir)


def _cpu_one(seq):
    rows, st = _CPU_IDX.search(seq)
    return (len(rows), sum(r["aligned_length"] for r in rows))


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import numpy as np
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and args.dist_backend == "nccl":
        raise SystemExit("rank %d has no GPU of its own (%d visible)" % (rank, ndev))
    local_rank = local_rank % ndev  # only differs with --dist-backend gloo (ranks sharing a GPU: a test of the N>1 logic)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    if world > 1:  # the ranks of a node share its cores: split them instead of oversubscribing the host-side stages
        os.environ.setdefault("LM_HOST_THREADS", str(max(1, usable_cores() // world)))
    import lexicmap_amd as la
    from lexicmap_amd import synth

    if args.builder is None:
        args.builder = "gpu" if args.workload in ("c2", "c3mini") else "oracle"
    wl = dict(WORKLOADS[args.workload])
    if args.queries:
        wl["queries"] = args.queries
    if args.genomes:
        wl["genomes"] = args.genomes
    if args.genome_len:
        wl["genome_len"] = args.genome_len

    t_setup = time.time()
    tmpdir = None
    index_dir = None
    opt_kw = {}
    if args.shard == "index" and world > 1:
        opt_kw = dict(shard_rank=rank, shard_count=world)
    gpu_built = args.builder == "gpu"
    weak = world > 1 and args.shard == "queries" and args.scaling == "weak"
    cpu_queries = None
    if gpu_built:
        # synthetic genomes + index generated directly in HBM by the GPU builder (no disk): the only way to have the
        # C2-size index on a fresh box within minutes
        gi = la.Index.synthetic(wl["genomes"], wl["genome_len"], wl["families"], seed=1000, max_div=0.10,
                                options=la.api.default_options(**opt_kw), device=local_rank)
        nloc = gi.info()["genomes"]

        def draw_queries(rng, n, locals_):
            out = []
            for i in range(n):
                l = int(locals_[int(rng.integers(0, len(locals_)))])
                L = int(rng.integers(wl["qlen"][0], wl["qlen"][1] + 1))
                L = min(L, wl["genome_len"])
                st = int(rng.integers(0, wl["genome_len"] - L + 1))
                q = np.frombuffer(gi.fetch(l, st, L), dtype=np.uint8)
                d = rng.random() * 0.10
                q = synth.mutate(rng, q, sub=d, ins=d / 10, dele=d / 10)
                if rng.random() < 0.5:
                    q = np.frombuffer(q.tobytes().translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1], dtype=np.uint8)
                out.append(("q%05d" % i, q.tobytes()))
            return out

        if weak:
            queries = draw_queries(np.random.default_rng(2000 + rank), wl["queries"], np.arange(nloc))  # this GPU's own batch
        else:
            if rank == 0:
                queries = draw_queries(np.random.default_rng(2000), wl["queries"], np.arange(nloc))
            else:
                queries = None
            if world > 1:
                box = [queries]
                dist.broadcast_object_list(box, src=0)
                queries = box[0]
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            # CPU-baseline sample: the first members of family 0 are fetched back from HBM and indexed by the ORACLE's
            # writer (reference on-disk format); sample queries are drawn from them with the same generator.
            import oracle as O
            fam = wl["families"]
            members = [g for g in range(0, wl["genomes"], fam)][:args.cpu_sample_genomes]
            tmpdir = tempfile.mkdtemp(prefix="lm_cpu_sample_")
            index_dir = os.path.join(tmpdir, "sample.lmi")
            t_s = time.time()
            gl = [("SYN_%09d.1" % g, [("syn%09d_c1" % g, gi.fetch(g, 0, wl["genome_len"]))]) for g in members]
            O.build_index(index_dir, gl, O.default_build_opt(chunks=8))
            cpu_queries = draw_queries(np.random.default_rng(2001), args.cpu_sample_queries, np.array(members))
            log("[rank 0] CPU sample index (%d genomes of family 0) built by the oracle in %.1f s" % (len(members), time.time() - t_s))
    else:
        import oracle as O
        tmpdir = os.path.join(tempfile.gettempdir(), "lm_bench_%s_%d_%d" % (args.workload, wl["genomes"], wl["genome_len"]))
        index_dir = os.path.join(tmpdir, "index.lmi")
        genomes = synth.make_genomes(wl["genomes"], wl["genome_len"], wl["families"], seed=1000, max_div=0.10)
        if rank == 0 and not os.path.exists(os.path.join(index_dir, "info.toml")):
            os.makedirs(tmpdir, exist_ok=True)
            O.build_index(index_dir, genomes, O.default_build_opt(chunks=8))
        if world > 1:
            dist.barrier()
        queries = synth.make_gene_queries(genomes, wl["queries"], seed=2000 + (rank if weak else 0), len_range=wl["qlen"],
                                          max_div=0.10)
        if opt_kw:
            whole_bases = sum(sum(len(c[1]) for c in g[1]) for g in genomes)
            opt_kw["total_bases_override"] = whole_bases
        gi = la.Index(index_dir, la.api.default_options(**opt_kw), device=local_rank)
    info = gi.info()
    log("[rank %d] index ready in %.1f s: %s" % (rank, time.time() - t_setup, info))

    # queries of this rank
    if weak:
        my = queries
    elif args.shard == "queries" and world > 1:
        my = [q for i, q in enumerate(queries) if i % world == rank]
    else:
        my = queries
    seqs = [q[1] for q in my]
    qb = gi.upload(seqs)  # bases resident in HBM before the timed region

    from lexicmap_amd import merge

    def step():
        """one pass of the hot path over this rank's resident batch, then (N>1) the hit-list merge: one all-gather of
        the per-rank HSP row records over RCCL and the reference's final ordering"""
        rows, st = gi.search_resident_np(qb)
        if world > 1:
            if args.shard == "queries":
                rows = rows.copy()  # the array is a view of the library's result
                if weak:
                    rows["query"] = rows["query"] + rank * len(queries)  # every rank has its own batch
                else:
                    rows["query"] = rows["query"] * world + rank  # local -> global query number (round-robin sharding)
            per_rank = merge.all_gather_rows(rows, device="cuda" if args.dist_backend == "nccl" else "cpu", host_on=0)
            rows = merge.merge_sharded(per_rank) if args.shard == "index" else merge.merge_query_sharded(per_rank)
        return rows, st

    for _ in range(args.warmup):
        step()
    gi.profile(True)
    gi.profile_reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    stats = None
    rows_np = None
    for _ in range(args.steps):
        rows_np, stats = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.time() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    prof = gi.profile_get()
    gi.profile(False)
    try:
        import resource
        log("[rank %d] peak host RSS %.1f GB after %d steps" % (rank, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576.0,
                                                             args.steps + args.warmup))
    except Exception:
        pass
    rows_total, aligned_total = int(len(rows_np)), int(rows_np["aligned_length"].sum())

    nq_total = len(queries) * (world if weak else 1)
    value = nq_total * args.steps / dt
    result = None
    if rank == 0:
        kern = [p for p in prof if p["name"].startswith("k_")]
        kern.sort(key=lambda p: -p["total_ms"])
        dom = kern[0] if kern else None
        kernels = []
        for p in kern:
            avg_ms = p["total_ms"] / max(p["launches"], 1)
            gbs = (p["bytes"] / max(p["launches"], 1)) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            kernels.append(dict(name=p["name"], launches=p["launches"], avg_ms=round(avg_ms, 4),
                                algorithmic_bytes_per_launch=int(p["bytes"] / max(p["launches"], 1)),
                                achieved_GBs=round(gbs, 3), frac_of_hbm_peak=round(gbs / HBM_PEAK_GBS, 6)))
        prims = [dict(name=p["name"], launches=p["launches"], avg_ms=round(p["total_ms"] / max(p["launches"], 1), 4))
                 for p in sorted(prof, key=lambda p: -p["total_ms"]) if not p["name"].startswith("k_")]
        roofline = None
        if dom:
            avg_ms = dom["total_ms"] / max(dom["launches"], 1)
            ach = (dom["bytes"] / max(dom["launches"], 1)) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            traffic, tnote = pmc_traffic(dom["name"]) if args.workload == "c2" else (None, "PMC passes are taken on c2")
            roofline = dict(bound="hbm", kernel=dom["name"], achieved=round(ach, 3), peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=round(ach / HBM_PEAK_GBS, 6), traffic=traffic, traffic_source=tnote,
                            avg_launch_ms=round(avg_ms, 4), launches=dom["launches"])
        # the HBM-bound kernel of the path (north_star: seed lookup against the in-HBM index) reported next to the dominant one
        roofline_lookup = None
        for pk in kern:
            if pk["name"] == "k_lookup_count":
                avg_ms = pk["total_ms"] / max(pk["launches"], 1)
                ach = (pk["bytes"] / max(pk["launches"], 1)) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
                tr, tn = pmc_traffic(pk["name"]) if args.workload == "c2" else (None, "PMC passes are taken on c2")
                roofline_lookup = dict(bound="hbm", kernel=pk["name"], achieved=round(ach, 3), peak=HBM_PEAK_GBS, unit="GB/s",
                                       frac=round(ach / HBM_PEAK_GBS, 6), traffic=tr, traffic_source=tn,
                                       avg_launch_ms=round(avg_ms, 4), launches=pk["launches"],
                                       traffic_GBs=(round(tr / (avg_ms * 1e-3) / 1e9, 1) if tr else None))
        go = shutil.which("go")
        result = {
            "metric": "queries/sec (lexicmap search hot path, seed index HBM-resident)",
            "value": round(value, 3), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            # weak: every GPU searches its own batch of the workload's size; strong: one batch (or the index) is divided
            "scaling": "weak" if (weak or (world == 1 and args.shard == "queries" and args.scaling == "weak")) else "strong",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "gbp_aligned_per_s": round(aligned_total * args.steps / dt / 1e9 * (1 if world == 1 else 1), 6),
            "config": {"workload": "%s: %d gene queries (%d-%d bp, <=10%% divergence) vs %d synthetic genomes x %d bp "
                                   "(%d families), M=%d masks, k=%d, index %s" %
                                   (args.workload, nq_total, wl["qlen"][0], wl["qlen"][1], wl["genomes"], wl["genome_len"],
                                    wl["families"], info["masks"], info["k"],
                                    "GPU-built in HBM" if gpu_built else "oracle-built, loaded from the reference format"),
                       "seeds_resident": info["seeds"], "index_hbm_bytes": info["hbm_bytes"], "parallelism":
                           (("q-shard x%d (index replicated), %d queries per GPU" % (world, len(my))) if args.shard == "queries"
                            else ("index-shard x%d" % world)),
                       "go_toolchain": go or "absent (reference Go binary cannot be built; CPU baseline is the C port)"},
            "stage_ms": {k: round(v, 3) for k, v in stats.items() if k.startswith("ms_")},
            "work": {k: v for k, v in stats.items() if not k.startswith("ms_")},
            "rows": rows_total,
            "roofline": roofline,
            "roofline_seed_lookup": roofline_lookup,
            "kernels": kernels,
            "rocprim_calls": prims,
        }
    gi.free_batch(qb)
    # CPU baseline on rank 0, N=1 only
    if rank == 0 and world == 1 and not args.no_cpu_baseline and index_dir:
        try:
            ncores = usable_cores()
            cb = cpu_baseline(index_dir, cpu_queries or queries, args.cpu_seconds, ncores)
            if cpu_queries is not None:
                cb["sample"] += ("; SAMPLE INDEX = %d members of one family fetched from the GPU-built set and indexed by "
                                 "the oracle writer (~%d genome hits/query instead of ~%d in the GPU run, so this CPU "
                                 "number is an upper bound for the full index)" %
                                 (args.cpu_sample_genomes, args.cpu_sample_genomes, wl["genomes"] // wl["families"]))
            if cpu_queries is not None:
                # the same sample (same index directory, same queries) through the HIP path: a like-for-like pair of numbers
                try:
                    g2 = la.Index(index_dir, device=local_rank)
                    qb2 = g2.upload([q[1] for q in cpu_queries])
                    g2.search_resident_np(qb2)
                    torch.cuda.synchronize()
                    t1 = time.time()
                    reps = 5
                    for _ in range(reps):
                        r2, _s2 = g2.search_resident_np(qb2)
                    torch.cuda.synchronize()
                    cb["gpu_on_same_sample"] = dict(value=round(len(cpu_queries) * reps / (time.time() - t1), 1),
                                                    unit="queries/s", rows_per_pass=int(len(r2)))
                    g2.free_batch(qb2)
                    g2.close()
                except Exception as e:
                    cb["gpu_on_same_sample"] = "failed: %r" % (e,)
            result["cpu_baseline"] = cb
        except Exception as e:  # the baseline must not kill the bench line
            result["cpu_baseline"] = dict(value=None, unit="queries/s", cores=0, kind="port", sample="failed: %r" % (e,))
    gi.close()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
