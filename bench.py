#!/usr/bin/env python3
"""bench.py — queries/s of the MI355X `lexicmap search` hot path (BASELINE.json metric), one JSON line on rank 0.

A "step" = one pass of the whole hot path (mask -> lookup -> chain -> pseudo-align -> extend/WFA -> HSP rows) over one
batch of synthetic queries whose bases are already resident in HBM.

Default workload = BASELINE.json configs[2] ("c3"): 10 000 ONT-style long reads (5-50 kb, sub 2 % / ins 2 % / del 3 %)
against 100 000 synthetic genomes on ONE MI355X — the largest single-GPU configuration; the genome length that fits the
288 GB next to the packed seed image is stated in config.workload.  `--workload c2` is configs[1] (1 000 gene queries vs
10 000 x 5-Mb genomes).

N > 1 (`--gpus N`: the script launches its own N ranks, one per GPU, unless it already runs under torchrun): the north-star
layout — the genome set and its seed index are SHARDED over the GPUs (genome g on rank g % N), every rank searches the whole
batch against its shard, per-shard HSP rows are merged with one RCCL all-gatherv per step; total work is fixed ("strong").
`--shard queries` (index replicated, queries divided; `--scaling weak|strong`) is an explicitly labelled extra.

The CPU baseline is the oracle (oracle/liblmo.so, a C restatement of the reference; kind "port") timed on the host cores
on a bounded sample of the same workload.  The Go reference cannot be built here (`go` absent) — probed and printed.
"""
import argparse
import hashlib
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PROFILE_ROUND = "r06"  # profiles/<round>_<workload>_pmc_*.json: the committed counter passes roofline.traffic is read from
# instruction-issue peaks of the chip (MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32, a wave64 VALU instruction issues over 2
# cycles; ONE scalar unit per CU), wave-instructions per second at 2.4 GHz
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2
SALU_ISSUE_PEAK = 256 * 2.4e9
# ... and what a SIMD was MEASURED to sustain on the kinds of instruction the WFA kernels are made of (experiments/valu_rate,
# profiles/r03_valu_rate.jsonl, 8 waves per SIMD): v_add_u32 2.5 cycles, but v_cmp + v_cndmask, DPP move + min, v_pk_min_u16,
# 64-bit shifts and VOP3 f32 FMA 3.8 - 4.2 cycles per wave64 instruction; s_add_u32 1.04 cycles per CU (= SALU_ISSUE_PEAK)
VALU_ISSUE_PEAK_MIX = 256 * 4 * 2.4e9 / 4


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=3,
                   help="untimed steps (default 3: with two lanes the scratch arena of a fresh handle keeps growing for the first "
                        "two or three steps of a C3-sized batch - 17.5, 14.5, then 12.2 s - before the steady state)")
    p.add_argument("--workload", default=os.environ.get("LM_BENCH_WORKLOAD", "c3"),
                   choices=["c3", "c2", "c4", "c5", "small", "tiny", "c3mini"])
    p.add_argument("--shard-of", type=int, default=0,
                   help="ONE process searches shard --shard-rank of an index sharded over this many GPUs (genome g on shard "
                        "g %% N, global total_bases): what one rank of the N-GPU run does, measurable on one GPU; adds the "
                        "host-timed lm_merge_sharded over N shards' worth of rows and a predicted N-GPU step (DESIGN.md 8)")
    p.add_argument("--shard-rank", type=int, default=0)
    p.add_argument("--queries", type=int, default=0, help="override the number of queries")
    p.add_argument("--genomes", type=int, default=0, help="override the number of genomes")
    p.add_argument("--genome-len", type=int, default=0)
    p.add_argument("--families", type=int, default=0, help="override the number of genome families")
    p.add_argument("--tag", default="", help="free text copied into config.tag (labels A/B runs)")
    p.add_argument("--shard", default="index", choices=["index", "queries"],
                   help="N>1: index = genomes + seed index sharded over the GPUs, queries broadcast (north-star layout, "
                        "default); queries = index replicated, query batch divided")
    p.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                   help="only with --shard queries: weak = every GPU searches its own batch of the workload's size")
    p.add_argument("--loader-check", action="store_true",
                   help="GPU-built workloads, N=1: write the index in the reference's on-disk format (lm_index_save), time "
                        "lm_index_open on it and compare the packed image it builds with the one made in HBM (loader GB/s)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-exclusive-step", action="store_true",
                   help="skip the extra serialised step (outside the timed region) that gives the exclusive kernel durations")
    p.add_argument("--cpu-seconds", type=float, default=12.0)
    p.add_argument("--gather", default="c", choices=["c", "torch"],
                   help="N > 1: the row gather through the library's lm_gather_rows (RCCL behind the C-ABI; default) or through "
                        "torch.distributed (lexicmap_amd/merge.py)")
    p.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                   help="nccl = RCCL (one GPU per rank); gloo only to exercise the N>1 logic with ranks sharing a GPU")
    p.add_argument("--builder", default=None, choices=["gpu", "oracle"],
                   help="gpu: genomes+index generated in HBM (default for c2/c3); oracle: CPU writer + on-disk format")
    p.add_argument("--cpu-sample-genomes", type=int, default=0,
                   help="genomes of the CPU-baseline sample index (default: one whole family + 16 genomes of other families)")
    p.add_argument("--cpu-sample-queries", type=int, default=0)
    p.add_argument("--cpu-subsets", default="auto",
                   help="CPU baseline on comparable work: genome counts of the subset indexes (GPU-built with the workload's own "
                        "generator, written by lm_index_save, opened by the oracle RAM-resident), e.g. 1250,5000; auto: two sizes "
                        "whose RAM-resident seeds fit the host; 0: skip")
    p.add_argument("--ab", default="", help="development: after the timed steps, time the same resident batch under other "
                   "experiment switches, e.g. 'LM_WFA_R16=0|LM_TWO_LANES=0 LM_WFA_R16=0' (variants separated by |, one warm-up + "
                   "--ab-steps steps each; reported under 'ab', not part of the metric)")
    p.add_argument("--ab-steps", type=int, default=2)
    return p.parse_args()


WORKLOADS = {
    # BASELINE.json configs[2]: 10k ONT-style long reads (5-50 kb) vs 100k genomes, 1 MI355X (chaining + WFA stress).
    # 100 000 x 2-Mb genomes: packed seed image ~105 GB + 2-bit genomes 50 GB of the 288 GB (5-Mb genomes would need
    # 125 GB of genomes + ~230 GB of seeds even at 9.5 B/seed: SURVEY.md §8d asks to state the shape that fits).
    # 1001 families (~100 members each; 1001 is coprime with 2/4/8 so family members spread over index shards).
    "c3": dict(genomes=100000, genome_len=2_000_000, families=1001, queries=10000, qlen=(5000, 50000), kind="reads"),
    # BASELINE.json configs[1]: 1k gene queries (1-2 kb) vs 10k synthetic 5-Mb genomes, index HBM-resident
    "c2": dict(genomes=10000, genome_len=5_000_000, families=100, queries=1000, qlen=(1000, 2000), kind="genes"),
    # BASELINE.json configs[3]: plasmid/prophage queries (50-200 kb, circular) vs 1M genomes, index sharded over 4 MI355X.
    # 800-kb genomes: a shard (250 000 genomes) is ~1.5e10 seeds = ~144 GB packed + 50 GB of 2-bit genomes of the 288 GB.
    # One GPU runs ONE shard of it (--shard-of 4 --shard-rank r); the driver's 4-GPU run shards it over the ranks.
    "c4": dict(genomes=1_000_000, genome_len=800_000, families=10007, queries=100, qlen=(50000, 200000), kind="circular",
               shards=4),
    # BASELINE.json configs[4]: AllTheBacteria-scale 1.9M genomes over 8 x 288 GB, mixed gene + read batch (90 % / 10 %)
    "c5": dict(genomes=1_900_000, genome_len=800_000, families=19001, queries=10000, qlen=(1000, 2000), kind="mixed",
               read_qlen=(5000, 50000), shards=8),
    # scaled-down shapes for development (NOT headline configs)
    "small": dict(genomes=200, genome_len=500_000, families=4, queries=1000, qlen=(1000, 2000), kind="genes"),
    "tiny": dict(genomes=24, genome_len=100_000, families=4, queries=64, qlen=(300, 1500), kind="genes"),
    "c3mini": dict(genomes=2000, genome_len=2_000_000, families=21, queries=500, qlen=(5000, 50000), kind="reads"),
}


def _code_only(text):
    """C / C++ source without comments, whitespace collapsed: what source_hash() hashes"""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == '"' or c == "'":  # string / character literal: copied as it is
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
        else:
            out.append(c)
            i += 1
    return " ".join("".join(out).split())


def source_hash(root=None):
    """identity of the kernels a PMC pass was taken on: sha256 over the CODE of the library sources (comments and whitespace do
    not count: a comment fixed after the counter passes were taken does not orphan them)"""
    h = hashlib.sha256()
    d = os.path.join(root or ROOT, "lexicmap_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            h.update(_code_only(open(os.path.join(d, f), encoding="utf-8", errors="replace").read()).encode())
    return h.hexdigest()[:16]


def source_hash_raw(root=None):
    """the hash of rounds 1-5a: over the raw bytes of the same files (tools/restamp_hash.py maps one to the other)"""
    h = hashlib.sha256()
    d = os.path.join(root or ROOT, "lexicmap_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, workload, per_step=False):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of THIS workload and THESE sources
    (profiles/r02_<workload>_pmc_{fetch,write}.json, two separate --pmc runs): (FETCH_SIZE + WRITE_SIZE) * 1024.  On
    gfx950 FETCH_SIZE tallies 128-B read requests as 64 B for wide streaming reads (MI355X_MICROARCH.md, HBM), so the read
    part is a lower bound (at most 2x low).  Passes recorded on other sources are refused.  Returns (bytes|None, note)."""
    tot, found = 0.0, False
    for fn, ctr in (("%s_%s_pmc_fetch.json" % (PROFILE_ROUND, workload), "FETCH_SIZE"), ("%s_%s_pmc_write.json" % (PROFILE_ROUND, workload), "WRITE_SIZE")):
        try:
            doc = json.load(open(os.path.join(ROOT, "profiles", fn)))
        except (OSError, ValueError):
            return None, "no committed PMC pass for workload %s" % workload
        if doc.get("source_hash") != source_hash():
            return None, "committed PMC pass was taken on other kernel sources (%s != %s): refused" % (
                doc.get("source_hash"), source_hash())
        pm = doc.get("pmc", {})
        # exact name only: a kernel the pass does not list yields null, never a neighbour's totals (round 5 matched by prefix and
        # gave the whole k_wfa_lean2 family's traffic to its 128-diagonal instantiation)
        best = (kernel, pm[kernel][ctr]) if kernel in pm and ctr in pm[kernel] else None
        if best is None:
            return None, "kernel not in the committed PMC pass"
        # per STEP when the caller asks for it: the passes time one step (--steps 1 --warmup 0), and the number of launches a
        # step is cut into differs between runs (batch parts halved under memory pressure stay halved), so a per-launch mean of
        # one run over the per-launch bytes of another would mix granularities
        tot += (best[1]["total"] / max(int(doc.get("window_steps", 1)), 1) if per_step else best[1]["mean"]) * 1024.0
        found = True
    return (int(tot) if found else None), ("(FETCH_SIZE+WRITE_SIZE)*1024 from profiles/%s_%s_pmc_*.json (same sources%s); "
                                           "read part is a lower bound on gfx950" %
                                           (PROFILE_ROUND, workload, "; the pass's whole step / this run's launches per step" if per_step else ""))


def pmc_issue(kernel, workload, per_step_launches=0):
    """instruction issue of `kernel` from the committed SQ pass of this workload and these sources
    (profiles/r02_<workload>_pmc_sq.json): the WFA and anchor-filter kernels are bound by instruction issue (scalar + vector
    ALU), not by HBM; a CU issues at most one vector and one scalar instruction per cycle (four SIMDs, a wavefront's vector
    instruction occupies its SIMD for four cycles)"""
    try:
        doc = json.load(open(os.path.join(ROOT, "profiles", "%s_%s_pmc_sq.json" % (PROFILE_ROUND, workload))))
    except (OSError, ValueError):
        return None
    if doc.get("source_hash") != source_hash():
        return None
    pm = doc.get("pmc", {})
    best = (kernel, pm[kernel]) if kernel in pm and "SQ_INSTS_VALU" in pm[kernel] else None  # (exact name only, see pmc_traffic)
    if best is None:
        return None
    v = best[1]
    wsteps = max(int(doc.get("window_steps", 1)), 1)  # steps between the pass's markers (1 in the passes without markers)
    out = {c.lower() + "_per_launch": int(v[c]["total"] / wsteps / per_step_launches if per_step_launches else v[c]["mean"])
           for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS") if c in v}
    out["note"] = "profiles/%s_%s_pmc_sq.json (kernels serialised by the counter pass%s)" % (
        PROFILE_ROUND, workload, "; " + doc["window"] if doc.get("window") else "")
    return out


def usable_cores():
    """cores this process may really use: the CPU count capped by the affinity mask and the cgroup quota"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(index_dir, queries, seconds, ncores):
    """oracle (port) on the host cores, bounded sample; returns dict"""
    import multiprocessing as mp
    import oracle as O
    n = len(queries)
    oi = O.Index(index_dir)  # calibrate on a few queries with one core
    t0 = time.time()
    k = 0
    while k < min(n, 3):
        oi.search(queries[k][1])
        k += 1
    per = (time.time() - t0) / max(k, 1)
    oi.close()
    nsample = int(max(ncores, min(40 * n, seconds * ncores / max(per, 1e-4))))
    sample = [queries[i % n] for i in range(nsample)]
    with mp.Pool(ncores, initializer=_cpu_init, initargs=(index_dir,)) as pool:
        t0 = time.time()
        res = pool.map(_cpu_one, [q[1] for q in sample], chunksize=max(1, len(sample) // (ncores * 8)))
        dt = time.time() - t0
    rows = sum(r[0] for r in res)
    bases = sum(r[1] for r in res)
    chains = sum(r[3] for r in res)
    # the oracle's rows of every sample query that was searched (query i of the sample list = queries[i % n])
    oracle_rows = {i: res[i][2] for i in range(min(n, nsample))}
    return dict(value=len(sample) / dt, unit="queries/s", cores=ncores, kind="port", _oracle_rows=oracle_rows,
                gbp_aligned_per_s=bases / dt / 1e9, rows_per_query=rows / max(len(sample), 1),
                chains_per_query=round(chains / max(len(sample), 1), 1),
                sample="%d searches over %d sample queries, oracle/liblmo.so (C restatement of the Go reference, RAM-resident "
                       "index = the reference's -w regime) on %d processes, %.1f s, %d HSP rows"
                       % (len(sample), n, ncores, dt, rows))


_CPU_IDX = None


def _cpu_init(index_dir):
    global _CPU_IDX
    import oracle as O
    _CPU_IDX = O.Index(index_dir)


ROW_CHECK_FIELDS = ("batch_genome", "cls", "hsp", "seq_idx", "nseqs", "seq_len", "rc", "aligned_length", "gaps", "qbegin",
                    "qend", "tbegin", "tend", "bitscore", "score", "matched_bases", "qcov_genome", "qcov_hsp", "pident")


def _cpu_one(seq):
    rows, st = _CPU_IDX.search(seq)
    compact = [tuple(r[f] for f in ROW_CHECK_FIELDS) + (r["evalue"], st["ngenomes"]) for r in rows]
    return (len(rows), sum(r["aligned_length"] for r in rows), compact, st["n_chains"])


def rows_equal_oracle(gpu_rows, oracle_rows, bg_map=None):
    """HIP rows (numpy lm_hsp array of one search over the sample queries) vs the oracle's rows of the same queries, row
    for row in output order: every integer / float64 column exact, e-value within 1e-9 relative, `hits` = the oracle's
    genome count.  bg_map: genome key in the oracle's (sample) index -> key of the same genome in the index the HIP rows
    come from (the full bench index).  Returns (equal, rows compared, first difference or None)."""
    import numpy as np
    q = gpu_rows["query"].astype(np.int64)
    ncmp = 0
    for qi, exp in sorted(oracle_rows.items()):
        got = gpu_rows[q == qi]  # rows of a query are contiguous and in final order
        if len(got) != len(exp):
            return False, ncmp, "query %d: %d HIP rows vs %d oracle rows" % (qi, len(got), len(exp))
        for j, e in enumerate(exp):
            g = got[j]
            if bg_map is not None:
                e = (bg_map[e[0]],) + tuple(e[1:])
            for k, f in enumerate(ROW_CHECK_FIELDS):
                if g[f] != e[k]:
                    return False, ncmp, "query %d row %d: %s = %r (HIP) vs %r (oracle)" % (qi, j, f, g[f].item(), e[k])
            ev = e[len(ROW_CHECK_FIELDS)]
            if abs(float(g["evalue"]) - ev) > 1e-9 * max(abs(ev), 1e-300):
                return False, ncmp, "query %d row %d: evalue %r vs %r" % (qi, j, float(g["evalue"]), ev)
            if int(g["hits"]) != e[len(ROW_CHECK_FIELDS) + 1]:
                return False, ncmp, "query %d row %d: hits %d vs %d" % (qi, j, int(g["hits"]), e[len(ROW_CHECK_FIELDS) + 1])
            ncmp += 1
    return True, ncmp, None


def set_input_bases(info_toml, total_bases):
    """the e-value database size of an index (info.toml input-bases, lib-index-search.go:1918) set to that of the index it is
    a part of"""
    import re
    t = open(info_toml).read()
    t2, n = re.subn(r"(?m)^(\s*input-bases\s*=\s*)\d+", lambda m: m.group(1) + str(int(total_bases)), t)
    if n != 1:
        raise SystemExit("bench.py: no input-bases line in %s" % info_toml)
    open(info_toml, "w").write(t2)


def genome_key(g):
    """batch << 17 | index of synthetic genome g (lm_builder.hip: 5000 genomes per batch, the reference's default)"""
    return ((g // 5000) << 17) | (g % 5000)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def maybe_spawn(args):
    """`bench.py --gpus N` outside torchrun: launch N ranks of this script (one per GPU) and wait"""
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is not None:
        if int(world_env) != args.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s" % (args.gpus, world_env))
        return
    if args.gpus <= 1:
        return
    import torch
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev < args.gpus and args.dist_backend == "nccl":
        raise SystemExit("bench.py: --gpus %d but only %d GPU(s) visible" % (args.gpus, ndev))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def draw_query(np, synth, rng, seq, wl, as_read=False):
    """one synthetic query from genome bases `seq` (uint8 array)"""
    if wl["kind"] == "reads" or as_read:  # ONT-like: sub 2 %, ins 2 %, del 3 %
        q = synth.mutate(rng, seq, sub=0.02, ins=0.02, dele=0.03)
    elif wl["kind"] == "circular":  # a plasmid / prophage: the region read from a random rotation point, lightly diverged
        rot = int(rng.integers(0, len(seq)))
        q = synth.mutate(rng, np.concatenate([seq[rot:], seq[:rot]]), sub=0.01, ins=0.002, dele=0.002)
    else:
        d = rng.random() * 0.10
        q = synth.mutate(rng, seq, sub=d, ins=d / 10, dele=d / 10)
    if rng.random() < 0.5:
        q = np.frombuffer(q.tobytes().translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1], dtype=np.uint8)
    return q.tobytes()


def main():
    args = parse()
    maybe_spawn(args)
    # ONE JSON line on stdout, whatever the libraries print: RCCL writes a version banner to stdout when a communicator is made
    # (lm_comm_init on every rank of an N-GPU run; the single-rank communicator of the shard model).  Everything written to file
    # descriptor 1 from here on goes to stderr; the line is written to the real stdout at the end.  (After maybe_spawn: the
    # ranks it starts must inherit the real stdout.)
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
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
        # the ranks of a node share its cores: split them instead of oversubscribing the host-side stages
        os.environ.setdefault("LM_HOST_THREADS", str(max(1, usable_cores() // world)))
    import lexicmap_amd as la
    from lexicmap_amd import merge, synth

    # the row gather of the sharded search through the library's C entry point (lm_gather_rows over RCCL: what the Go host of
    # INTEGRATION.md calls); the communicator's id travels by one torch.distributed broadcast.  A failure to set it up is
    # reported in the line (config.gather) and the torch.distributed gather of lexicmap_amd/merge.py takes over.
    comm = None
    gather_kind = "none (1 rank)"
    if world > 1:
        gather_kind = "torch.distributed gather (lexicmap_amd/merge.py)"
        # (gloo = ranks sharing a GPU, which RCCL refuses: the C gather only with a stand-in behind LM_RCCL_LIB - tests/fake_rccl.c)
        if args.gather == "c" and (args.dist_backend == "nccl" or os.environ.get("LM_RCCL_LIB")):
            cdev = "cuda" if args.dist_backend == "nccl" else "cpu"
            try:
                from lexicmap_amd.api import Comm, COMM_ID_BYTES
                # every rank must be able to bind an RCCL before ANY rank enters ncclCommInitRank (a rank that failed there would
                # leave the others waiting inside it): agree on a "loadable" flag first
                import ctypes
                loadable = 0
                for name in (os.environ.get("LM_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"):
                    if not name:
                        continue
                    try:
                        ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL)
                        loadable = 1
                        break
                    except OSError:
                        pass
                okl = torch.tensor([loadable], device=cdev)
                dist.all_reduce(okl, op=dist.ReduceOp.MIN)
                if int(okl.item()) == 0:
                    raise RuntimeError("no RCCL library could be loaded on some rank")
                idt = torch.zeros(COMM_ID_BYTES + 1, dtype=torch.uint8, device=cdev)  # [0]: rank 0 has an id to offer
                if rank == 0:
                    try:
                        uid = Comm.unique_id()
                        idt.copy_(torch.frombuffer(bytearray(b"\x01" + uid), dtype=torch.uint8))
                    except Exception as e:  # noqa: BLE001 (every rank learns it from the flag: nobody waits in ncclCommInitRank)
                        gather_kind += "; lm_comm_unique_id failed on rank 0: %r" % (e,)
                dist.broadcast(idt, src=0)
                idb = bytes(idt.cpu().numpy().tobytes())
                if idb[0] == 1:
                    comm = Comm(idb[1:], world, rank, local_rank)
                    gather_kind = "lm_gather_merge_rows (C-ABI: RCCL send/recv into the merging rank, merge on its device)"
                elif rank != 0:
                    gather_kind += "; rank 0 offered no communicator id"
            except Exception as e:  # noqa: BLE001
                comm = None
                gather_kind += "; lm_gather_merge_rows unavailable: %r" % (e,)
            ok = torch.tensor([1 if comm is not None else 0], device=cdev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # every rank or none
            if int(ok.item()) == 0 and comm is not None:
                comm.close()
                comm = None
                gather_kind = "torch.distributed gather (lexicmap_amd/merge.py); the C gather is unavailable on another rank"

    comm_box = [comm, gather_kind]  # (step() may give the C gather up at run time)

    if args.builder is None:
        args.builder = "gpu" if args.workload in ("c2", "c3", "c4", "c5", "c3mini") else "oracle"
    wl = dict(WORKLOADS[args.workload])
    if args.queries:
        wl["queries"] = args.queries
    if args.genomes:
        wl["genomes"] = args.genomes
    if args.genome_len:
        wl["genome_len"] = args.genome_len
    if args.families:
        wl["families"] = args.families
    wl["families"] = min(wl["families"], wl["genomes"])

    t_setup = time.time()
    tmpdir = None
    index_dir = None
    opt_kw = {}
    index_sharded = args.shard == "index" and world > 1
    if index_sharded:
        opt_kw = dict(shard_rank=rank, shard_count=world)
    # one shard of an N-GPU run on this one GPU: c4 / c5 only exist sharded (a shard is what fits 288 GB)
    if world == 1 and not args.shard_of and wl.get("shards"):
        args.shard_of = wl["shards"]
    shard_of = args.shard_of if (world == 1 and args.shard_of > 1) else 0
    if shard_of:
        if not (0 <= args.shard_rank < shard_of):
            raise SystemExit("bench.py: --shard-rank must be in [0, --shard-of)")
        if args.builder != "gpu":
            raise SystemExit("bench.py: --shard-of needs the GPU builder")
        opt_kw = dict(shard_rank=args.shard_rank, shard_count=shard_of)
        args.no_cpu_baseline = True  # the CPU leg belongs to the N=1 line of the unsharded workloads
    gpu_built = args.builder == "gpu"
    weak = world > 1 and args.shard == "queries" and args.scaling == "weak"
    cpu_queries = None
    lo_len, hi_len = wl["qlen"]

    def is_read(i):  # mixed batch (c5): every tenth query is an ONT-style read, the rest are genes
        return wl["kind"] == "mixed" and i % 10 == 9

    def qlen_of(rng, i=0):
        if wl["kind"] == "reads" or is_read(i):  # log-uniform lengths
            lo, hi = wl["read_qlen"] if is_read(i) else (lo_len, hi_len)
            return int(np.exp(rng.uniform(np.log(lo), np.log(hi))))
        return int(rng.integers(lo_len, hi_len + 1))

    if gpu_built:
        # synthetic genomes + index generated directly in HBM by the GPU builder (no disk): the only way to have a
        # C2/C3-size index on a fresh box within minutes
        gi = la.Index.synthetic(wl["genomes"], wl["genome_len"], wl["families"], seed=1000, max_div=0.10,
                                options=la.api.default_options(**opt_kw), device=local_rank)
        nloc = gi.info()["genomes"]
        nshard = world if index_sharded else (shard_of or 1)
        my_shard = rank if index_sharded else (args.shard_rank if shard_of else 0)

        def query_i(i, seed_base):
            """query number i of the batch: the same bases whatever the number of ranks; None on ranks that do not hold
            its source genome (index sharding).  --shard-of: the source is moved to the member of its residue class that
            this shard holds (the family members on the other shards are what the other ranks would find)."""
            rng = np.random.default_rng([seed_base, i])
            g = int(rng.integers(0, wl["genomes"]))
            L = min(qlen_of(rng, i), wl["genome_len"])
            st = int(rng.integers(0, wl["genome_len"] - L + 1))
            if shard_of:
                g = min(g - g % nshard + my_shard, (wl["genomes"] - 1 - my_shard) // nshard * nshard + my_shard)
            if g % nshard != my_shard:
                return None
            src = np.frombuffer(gi.fetch(g // nshard, st, L), dtype=np.uint8)
            return ("q%05d" % i, draw_query(np, synth, rng, src, wl, as_read=is_read(i)))

        if weak:
            queries = [query_i(i, 2000 + rank) for i in range(wl["queries"])]  # this GPU's own batch
        elif index_sharded:
            mine = [(i, query_i(i, 2000)) for i in range(wl["queries"])]
            mine = [(i, q) for i, q in mine if q is not None]
            box = [None] * world
            dist.all_gather_object(box, mine)
            queries = [None] * wl["queries"]
            for part in box:
                for i, q in part:
                    queries[i] = q
        else:
            queries = [query_i(i, 2000) for i in range(wl["queries"])] if rank == 0 else None
            if world > 1:
                box = [queries]
                dist.broadcast_object_list(box, src=0)
                queries = box[0]
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            # CPU-baseline sample index: ONE WHOLE FAMILY (so that a sample query sees the same ~genomes/family hits as
            # in the GPU run) + a slice of other families, fetched back from HBM and indexed by the ORACLE's writer
            # (reference on-disk format); sample queries are drawn from the family with the same generator.
            import oracle as O
            fam = wl["families"]
            members = [g for g in range(0, wl["genomes"], fam)]
            others = [g for g in range(1, wl["genomes"]) if g % fam != 0][:16]
            if args.cpu_sample_genomes:
                members = members[:args.cpu_sample_genomes]
                others = others[:max(0, args.cpu_sample_genomes - len(members))]
            tmpdir = tempfile.mkdtemp(prefix="lm_cpu_sample_")
            index_dir = os.path.join(tmpdir, "sample.lmi")
            t_s = time.time()
            # ascending genome numbers (ties between genomes break on the genome key: the same relative order in both
            # indexes), the bench index's own mask set, and its database size for the e-values: the sample index is then
            # exactly the part of the bench index that belongs to these genomes (per-genome independence of the search,
            # lib-index-search.go:1537-1556), which is what `full_index_rows_equal` below rests on
            sample_genomes = sorted(members + others)
            assert len(sample_genomes) < 5000  # one genome batch: key of the j-th sample genome = j
            gl = [("SYN_%09d.1" % g, [("syn%09d_c1" % g, gi.fetch(g, 0, wl["genome_len"]))]) for g in sample_genomes]
            M_ = gi.info()["masks"]
            mp_ = la.lib().lm_index_masks(gi.h)
            O.build_index(index_dir, gl, O.default_build_opt(chunks=8), masks=[mp_[i] for i in range(M_)])
            set_input_bases(os.path.join(index_dir, "info.toml"), gi.info()["total_bases"])
            nsq = args.cpu_sample_queries or (64 if wl["kind"] == "reads" else 512)
            cpu_queries = []
            for i in range(nsq):
                rng = np.random.default_rng([2001, i])
                g = members[int(rng.integers(0, len(members)))]
                L = min(qlen_of(rng), wl["genome_len"])
                st = int(rng.integers(0, wl["genome_len"] - L + 1))
                src = np.frombuffer(gi.fetch(g, st, L), dtype=np.uint8)
                cpu_queries.append(("s%05d" % i, draw_query(np, synth, rng, src, wl)))
            cpu_sample_note = ("SAMPLE INDEX = the %d members of family 0 + %d genomes of other families, fetched from the "
                               "GPU-built set and indexed by the oracle writer with the bench index's masks and database size "
                               "(%.0f s); sample queries drawn from the family, so hits/query match the GPU run" %
                               (len(members), len(others), time.time() - t_s))
            log("[rank 0] CPU sample index (%d genomes) built by the oracle in %.1f s" % (len(gl), time.time() - t_s))
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
        if wl["kind"] == "reads":
            queries = synth.make_reads(genomes, wl["queries"], seed=2000 + (rank if weak else 0), len_range=wl["qlen"])
        else:
            queries = synth.make_gene_queries(genomes, wl["queries"], seed=2000 + (rank if weak else 0), len_range=wl["qlen"],
                                              max_div=0.10)
        cpu_sample_note = "the bench index itself (oracle-built)"
        if opt_kw:
            opt_kw["total_bases_override"] = sum(sum(len(c[1]) for c in g[1]) for g in genomes)
        gi = la.Index(index_dir, la.api.default_options(**opt_kw), device=local_rank)
    info = gi.info()
    log("[rank %d] index ready in %.1f s: %s" % (rank, time.time() - t_setup, info))
    loader = None
    if args.loader_check and gpu_built and world == 1 and not shard_of:
        sdir = os.path.join(tempfile.mkdtemp(prefix="lm_saved_"), "saved.lmi")

        def content_digest(ix_):
            """sha256 over the (k-mer, value) lists of 64 evenly spaced masks and four slices of genome bases: the CONTENTS of
            the packed image, not only its sizes"""
            hh = hashlib.sha256()
            M_ = ix_.info()["masks"]
            for m_ in range(0, M_, max(1, M_ // 64)):
                k_, v_ = ix_.mask_seeds(m_)
                o_ = np.lexsort((v_, k_))      # seeds with one k-mer come out in the order they were placed: not part of the image
                hh.update(k_[o_].tobytes())
                hh.update(v_[o_].tobytes())
            ng_ = ix_.info()["genomes"]
            for g_ in sorted({0, ng_ // 3, (2 * ng_) // 3, ng_ - 1}):
                hh.update(ix_.fetch(g_, 0, min(100000, wl["genome_len"])))
            return hh.hexdigest()

        digest0 = content_digest(gi)
        t0 = time.time()
        gi.save(sdir, chunks=32)
        t_save = time.time() - t0
        disk = sum(os.path.getsize(os.path.join(r, f)) for r, _d, fs in os.walk(sdir) for f in fs)
        gi.close()  # one image at a time in HBM
        t0 = time.time()
        gi = la.Index(sdir, device=local_rank)
        t_load = time.time() - t0
        info2 = gi.info()
        sizes_same = all(info[f] == info2[f] for f in ("seeds", "genomes", "genome_bases", "seed_bytes", "outlier_seeds", "key_bits", "val_bits"))
        same = sizes_same and content_digest(gi) == digest0
        loader = dict(index_files_bytes=int(disk), save_s=round(t_save, 2), open_s=round(t_load, 2),
                      open_GBps_of_files=round(disk / t_load / 1e9, 3), seeds_per_s=round(info2["seeds"] / t_load),
                      image_sizes_equal_to_hbm_built=bool(sizes_same), sampled_contents_equal_to_hbm_built=bool(same),
                      contents_compared="sha256 over the (k-mer, value) lists of 64 evenly spaced masks (lm_index_mask_seeds) and "
                                        "100-kb slices of four genomes, before the save and after the load",
                      note="lm_index_save -> lm_index_open on the box's local disk; the searches below run on the LOADED index")
        log("[rank 0] loader check: %s" % loader)
        shutil.rmtree(os.path.dirname(sdir), ignore_errors=True)
        if not same:
            raise SystemExit("bench.py: the index loaded from disk differs from the one built in HBM: %s vs %s" % (info, info2))
        info = info2

    if gpu_built and rank == 0:
        log("[rank 0] %d queries drawn in %.1f s" % (len(queries), time.time() - t_setup))
    # queries of this rank
    if weak or index_sharded or world == 1:
        my = queries
    else:
        my = [q for i, q in enumerate(queries) if i % world == rank]
    seqs = [q[1] for q in my]
    t_up = time.time()
    qb = gi.upload(seqs)  # bases resident in HBM before the timed region
    upload_s = time.time() - t_up
    query_bases = sum(len(s) for s in seqs)

    def step():
        """one pass of the hot path over this rank's resident batch, then (N>1) the hit-list merge: one all-gather of
        the per-rank HSP row records over RCCL and the reference's final ordering"""
        rows, st = gi.search_resident_np(qb)
        if world > 1:
            if not index_sharded:
                rows = rows.copy()  # the array is a view of the library's result
                if weak:
                    rows["query"] = rows["query"] + rank * len(queries)  # every rank has its own batch
                else:
                    rows["query"] = rows["query"] * world + rank  # local -> global query number (round-robin sharding)
            if comm_box[0] is not None and index_sharded:
                # the gather and the merge behind the C-ABI in one call (lm_gather_merge_rows).  A failure here is fatal for the
                # whole job: the ranks cannot agree on another path in the middle of a collective (one that switched alone would
                # leave the others blocked in their receives)
                merged = comm_box[0].gather_merge_rows(rows, root=0, index=gi)
                return (merged if rank == 0 else rows), st
            per_rank = None
            if comm_box[0] is not None:
                per_rank, _cnt = comm_box[0].gather_rows(rows, root=0)  # (a failure raises: fatal, see above)
                if per_rank is None:
                    per_rank = [rows]
            if per_rank is None:
                per_rank = merge.all_gather_rows(rows, device="cuda" if args.dist_backend == "nccl" else "cpu", host_on=0)
            # index shards: the library's C merge (lm_merge_sharded: final order per query + global hits) on rank 0
            rows = (merge.merge_sharded_c(per_rank) if rank == 0 else per_rank[0]) if index_sharded else \
                merge.merge_query_sharded(per_rank)
        return rows, st

    warmup_ms = []  # (the first of them is the cold step of a fresh handle: reported as first_step_ms)
    for _ in range(args.warmup):
        t_s0 = time.time()
        step()
        warmup_ms.append(round((time.time() - t_s0) * 1e3, 1))
    gi.profile(True)
    gi.profile_reset()
    gi.profile_mark(1)  # (where tools/summarize_rocprof.py cuts a rocprofv3 pass: the timed - warm - steps only)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    stats = None
    rows_np = None
    step_ms = []
    for _ in range(args.steps):
        t_s0 = time.time()
        rows_np, stats = step()
        step_ms.append(round((time.time() - t_s0) * 1e3, 1))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.time() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    gi.profile_mark(2)
    prof = gi.profile_get()
    # the printer (search.go:437-533: one writer goroutine) on the rows of the last timed step: lm_format_rows (host threads, one
    # buffer) + one write to /dev/null - what the reference's queries/s includes and the timed region above does not
    host_e2e = None
    if rank == 0 and world == 1 and rows_np is not None and len(rows_np):
        try:
            ids_ = [q[0] for q in my]
            lens_ = [len(q[1]) for q in my]
            text_, fmt_s = la.api.format_rows(rows_np, ids_, lens_)
            t_w = time.time()
            with open(os.devnull, "wb") as fh:
                fh.write(text_)
            wr_s = time.time() - t_w
            st_s = dt / args.steps
            host_e2e = dict(format_rows_s=round(fmt_s, 4), write_devnull_s=round(wr_s, 4), tsv_bytes=len(text_),
                            rows_per_s_formatted=round(len(rows_np) / max(fmt_s, 1e-9)),
                            share_of_step=round((fmt_s + wr_s) / st_s, 4),
                            queries_per_s_with_printer=round(len(my) / (st_s + fmt_s + wr_s), 3),
                            note="lm_format_rows + one write after the step (serial: the reference prints beside its searches); "
                                 "the host PROGRAM (tests/cabi_shim.c as its own process) is timed on a subset index: "
                                 "cpu_baseline.subsets[].host_program")
            del text_
        except Exception as e:  # noqa: BLE001
            host_e2e = dict(failed=repr(e))
    gi.profile(False)
    # One more step OUTSIDE the timed region with the kernels serialised (no overlapped streams): in the timed steps the WFA
    # length classes, the anchor kernels of the next chunk and the fallback share the chip, so a kernel's HIP-event time there
    # includes the time it spent waiting for issue slots; the roofline figures use the exclusive durations of this step.
    prof_x = None
    if rank == 0 and world == 1 and not args.no_exclusive_step:
        gi.profile_exclusive(True)
        gi.profile(True)
        gi.profile_reset()
        step()
        torch.cuda.synchronize()
        prof_x = {p["name"]: p for p in gi.profile_get()}
        gi.profile(False)
        gi.profile_exclusive(False)
    ab = None
    if args.ab and world == 1:
        ab = []
        for variant in args.ab.split("|"):
            kv = dict(x.split("=", 1) for x in variant.split())
            saved = {k: os.environ.get(k) for k in kv}
            os.environ.update(kv)
            gi.tuning_reload()
            step()
            ts = []
            for _ in range(args.ab_steps):
                t_s0 = time.time()
                r_ab, _st = step()
                ts.append(round((time.time() - t_s0) * 1e3, 1))
            ab.append(dict(env=kv, step_ms=ts, rows=int(len(r_ab))))
            log("[rank 0] A/B %s: %s ms per step, %d rows" % (variant, ts, len(r_ab)))
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        gi.tuning_reload()
    try:
        import resource
        log("[rank %d] peak host RSS %.1f GB after %d steps" % (rank, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576.0,
                                                             args.steps + args.warmup))
    except Exception:
        pass
    shard_model = None
    if shard_of and rank == 0:
        # What the N-GPU step adds on top of this shard's search: the row gather and lm_merge_sharded on rank 0.  The other
        # shards' rows are emulated by re-keyed copies of this shard's rows (same number, other genomes), so the merge runs
        # on a full N-shard row set; timed on the host.
        base = rows_np.copy()
        per_rank = []
        for r in range(shard_of):
            c = base.copy()
            # the genomes shard r holds are those with number = r modulo N: this shard's genome g stands for its neighbour g + r
            # (a real genome of the set: its names resolve as they would on the merging rank)
            gnum = ((c["batch_genome"] >> np.uint64(17)) * np.uint64(5000) + (c["batch_genome"] & np.uint64(0x1ffff))).astype(np.int64)
            gnum = np.clip(gnum - args.shard_rank + r, 0, wl["genomes"] - 1).astype(np.uint64)
            c["batch_genome"] = ((gnum // np.uint64(5000)) << np.uint64(17)) | (gnum % np.uint64(5000))
            per_rank.append(c)
        t_m = time.time()
        merged = merge.merge_sharded_c(per_rank)
        merge_host_s = time.time() - t_m
        # ... and by what the N-GPU run does (lm_gather_merge_rows): the rows where the receives put them - in device memory, rank
        # order - merged on the device, downloaded once, names re-attached (lm_merge_sharded_device on a single-rank communicator;
        # the second call: the first sizes the buffers).  Rows must equal the host merge's.
        merge_s, merge_kind = merge_host_s, "lm_merge_sharded (host threads)"
        try:
            from lexicmap_amd.api import Comm
            cm = Comm(Comm.unique_id(), 1, 0, local_rank)
            cat = np.concatenate([c_.view(np.uint8).reshape(-1) for c_ in per_rank]) if len(base) else np.zeros(0, np.uint8)
            dev = torch.from_numpy(cat).cuda()
            torch.cuda.synchronize()
            counts_ = [len(c_) for c_ in per_rank]
            cm.merge_sharded_device(dev.data_ptr(), counts_, index=gi)
            t_m = time.time()
            md = cm.merge_sharded_device(dev.data_ptr(), counts_, index=gi)
            merge_s = time.time() - t_m
            strip = lambda a: [a[f].tobytes() for f in merge.ROW_DTYPE.names if f not in merge.PTR_FIELDS]  # noqa: E731
            if strip(md) != strip(merged):
                raise SystemExit("bench.py: the device merge and lm_merge_sharded disagree")
            merge_kind = "lm_merge_sharded_device (device merge + one download + names); rows equal lm_merge_sharded's"
            del md, dev, cat
            cm.close()
        except (RuntimeError, OSError) as e:
            merge_kind += "; device merge unavailable: %r" % (e,)
        row_bytes = int(base.dtype.itemsize) * len(base)
        # gather to rank 0 over xGMI: (N-1) shards' rows into one GPU, ~153 GB/s per link (MI355X_MICROARCH.md); the download of
        # the merged rows is inside the timed device merge
        gather_s = (shard_of - 1) * row_bytes / 153e9
        step_s0 = dt / args.steps
        fe_ms = sum(v for k, v in stats.items() if k in ("ms_mask", "ms_lookup"))
        shard_model = dict(shards=shard_of, shard_rank=args.shard_rank, shard_step_ms=round(step_s0 * 1e3, 3),
                           replicated_front_end_ms=round(fe_ms, 3),
                           rows_this_shard=int(len(base)), rows_merged=int(len(merged)), row_bytes_per_shard=row_bytes,
                           merge_ms=round(merge_s * 1e3, 3), merge_kind=merge_kind, merge_ms_host=round(merge_host_s * 1e3, 3),
                           gather_ms_estimated=round(gather_s * 1e3, 3),
                           predicted_step_ms=round((step_s0 + merge_s + gather_s) * 1e3, 3),
                           predicted_queries_per_s=round(len(queries) / (step_s0 + merge_s + gather_s), 3),
                           note="one shard measured on one GPU; predicted N-GPU step = this shard's step + gather (estimated "
                                "from the row bytes) + the timed merge (merge_kind) over N shards' worth of rows; the efficiency "
                                "against the 1-GPU line is computed in DESIGN.md section 8, not here")
        del merged, per_rank, base
    rows_total, aligned_total = int(len(rows_np)), int(rows_np["aligned_length"].sum())

    nq_total = len(queries) * (world if weak else 1)
    value = nq_total * args.steps / dt
    result = None
    if rank == 0:
        step_s = dt / args.steps
        kern = [p for p in prof if p["name"].startswith("k_")]
        kern.sort(key=lambda p: -p["total_ms"])

        def excl(name):
            """(exclusive avg ms per launch, launches, bytes per launch) of the serialised extra step, or None"""
            if not prof_x or name not in prof_x or not prof_x[name]["launches"]:
                return None
            q = prof_x[name]
            return q["total_ms"] / q["launches"], q["launches"], q["bytes"] / q["launches"]

        def entry(p):
            per_step_ms = p["total_ms"] / args.steps
            avg_ms = p["total_ms"] / max(p["launches"], 1)
            alg = p["bytes"] / max(p["launches"], 1)
            e = excl(p["name"])
            ex_ms, ex_alg = (e[0], e[2]) if e else (None, None)
            gbs = (ex_alg / (ex_ms * 1e-3) / 1e9) if e and ex_ms > 0 else (alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0)
            return dict(name=p["name"], launches=p["launches"], avg_ms=round(avg_ms, 4), ms_per_step=round(per_step_ms, 3),
                        exclusive_avg_ms=(round(ex_ms, 4) if e else None),
                        exclusive_ms_per_step=(round(e[0] * e[1], 3) if e else None),
                        algorithmic_bytes_per_launch=int(ex_alg if e else alg),
                        achieved_GBs=round(gbs, 3), frac_of_hbm_peak=round(gbs / HBM_PEAK_GBS, 6))

        kernels = [entry(p) for p in kern]
        prims = [entry(p) for p in sorted(prof, key=lambda p: -p["total_ms"]) if not p["name"].startswith("k_")]

        def roof(pk):
            """roofline entry of ONE kernel (instantiation): achieved = algorithmic bytes per launch / EXCLUSIVE launch
            duration (serialised extra step; the co-scheduled duration of the timed steps is given beside it)"""
            avg_ms = pk["total_ms"] / max(pk["launches"], 1)
            e = excl(pk["name"])
            dur_ms = e[0] if e else avg_ms
            alg = e[2] if e else pk["bytes"] / max(pk["launches"], 1)
            ach = alg / (dur_ms * 1e-3) / 1e9 if dur_ms > 0 else 0.0
            # counters of the pass's one step spread over THIS run's launches of a (serialised) step: same granularity as
            # algorithmic_bytes_per_launch and avg_launch_ms
            nl = e[1] if e else 0
            tr, tn = pmc_traffic(pk["name"], args.workload, per_step=bool(nl))
            if tr and nl:
                tr = int(tr / nl)
            issue = pmc_issue(pk["name"], args.workload, per_step_launches=nl)
            if issue and dur_ms > 0:
                # the kernel's wave-instructions per second against what the chip can issue (MI355X_MICROARCH.md)
                issue["issue_frac_valu"] = round(issue.get("sq_insts_valu_per_launch", 0) / (dur_ms * 1e-3) / VALU_ISSUE_PEAK, 4)
                issue["issue_frac_salu"] = round(issue.get("sq_insts_salu_per_launch", 0) / (dur_ms * 1e-3) / SALU_ISSUE_PEAK, 4)
                issue["issue_frac_valu_measured_mix"] = round(issue.get("sq_insts_valu_per_launch", 0) / (dur_ms * 1e-3) / VALU_ISSUE_PEAK_MIX, 4)
                # a fraction above 1 of a hardware issue rate is an accounting error (round 5: a family's counters over one
                # instantiation's launches), never a finding: refused
                over = [k for k in ("issue_frac_valu", "issue_frac_salu") if issue[k] > 1.0]
                if over:
                    issue = dict(refused="%s above 1.0 (%s): the counters and the launches do not describe the same work" % (
                        ", ".join(over), ", ".join("%s=%s" % (k, issue[k]) for k in over)), note=issue.get("note"))
            return dict(bound="hbm", kernel=pk["name"], achieved=round(ach, 3), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(ach / HBM_PEAK_GBS, 6), traffic=tr, traffic_source=tn,
                        traffic_over_algorithmic=(round(tr / alg, 2) if tr and alg else None),
                        traffic_GBs=(round(tr / (dur_ms * 1e-3) / 1e9, 1) if tr else None),
                        algorithmic_bytes_per_launch=int(alg), avg_launch_ms=round(dur_ms, 4),
                        duration_kind=("exclusive (serialised extra step)" if e else "co-scheduled (timed steps)"),
                        avg_launch_ms_coscheduled=round(avg_ms, 4), launches=pk["launches"], instruction_issue=issue)

        # The dominant kernel = the single kernel (one instantiation of a template counts by itself) with the largest
        # exclusive time per step; the other instantiations of its template are listed beside it, each with its own figures.
        def family(name):
            return "k_wfa" if name.startswith(("k_wfa_lean", "k_wfa_win", "k_wfa_mw")) else name

        def xtime(pk):
            e = excl(pk["name"])
            return e[0] * e[1] if e else pk["total_ms"] / args.steps

        roofline = None
        if kern:
            top = max(kern, key=xtime)
            roofline = roof(top)
            sibs = [roof(m) for m in kern if m is not top and family(m["name"]) == family(top["name"])]
            if sibs:
                roofline["other_instantiations"] = sibs
            if family(top["name"]) == "k_wfa":
                roofline["note"] = ("integer wavefront recurrences out of LDS: bound by instruction issue (see instruction_issue: "
                                    "the scalar unit first), not by HBM - DESIGN.md section 4")
        # the HBM-bound stage of the path (north_star: seed lookup against the in-HBM index): the search kernel and the
        # whole stage (prep + sort + count + scan + emit) against the same SURVEY 8(d) bytes
        roofline_lookup = None
        byname = {p["name"]: p for p in prof}
        lk = "k_lookup_fused" if "k_lookup_fused" in byname else ("k_lookup_count" if "k_lookup_count" in byname else None)
        if lk:
            roofline_lookup = roof(byname[lk])
            names = ("k_lookup_prep", "sort_lookups", "k_lookup_fused", "k_lookup_count", "scan", "k_lookup_emit")
            nl = max(byname[lk]["launches"], 1)
            if prof_x and lk in prof_x:
                stage_ms = sum(prof_x[n]["total_ms"] for n in names if n in prof_x) / max(prof_x[lk]["launches"], 1)
                stage_bytes = sum(prof_x[n]["bytes"] for n in (lk, "k_lookup_emit") if n in prof_x) / max(prof_x[lk]["launches"], 1)
            else:
                stage_ms = sum(byname[n]["total_ms"] for n in names if n in byname) / nl
                stage_bytes = sum(byname[n]["bytes"] for n in (lk, "k_lookup_emit") if n in byname) / nl
            roofline_lookup["stage"] = "k_lookup_prep + sort_lookups + %s%s" % (lk, "" if lk == "k_lookup_fused" else " + scan + k_lookup_emit")
            roofline_lookup["stage_ms"] = round(stage_ms, 4)
            roofline_lookup["stage_frac"] = round(stage_bytes / (stage_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if stage_ms > 0 else None
        # whole pipeline: sum of the algorithmic bytes of all kernels AND rocPRIM calls per step over the step time
        alg_step = sum(p["bytes"] for p in prof) / args.steps
        kern_ms_step = sum(p["total_ms"] for p in prof) / args.steps
        roofline_pipeline = dict(bound="hbm", achieved=round(alg_step / step_s / 1e9, 3), peak=HBM_PEAK_GBS, unit="GB/s",
                                 frac=round(alg_step / step_s / 1e9 / HBM_PEAK_GBS, 6),
                                 algorithmic_bytes_per_step=int(alg_step),
                                 kernel_ms_per_step=round(kern_ms_step, 3),
                                 kernel_time_fraction_of_step=round(kern_ms_step / (step_s * 1e3), 4),
                                 exclusive_kernel_ms_per_step=(round(sum(q["total_ms"] for q in prof_x.values()), 3) if prof_x else None),
                                 note="kernel_ms_per_step sums co-scheduled durations (streams overlap: it may exceed the step); "
                                      "exclusive_kernel_ms_per_step is the same sum with the kernels serialised")
        go = shutil.which("go")
        qdesc = {"reads": "ONT-style reads (%d-%d bp log-uniform, sub 2%% ins 2%% del 3%%)",
                 "circular": "circular plasmid/prophage queries (%d-%d bp, random rotation, sub 1%% indel 0.4%%)",
                 "mixed": "mixed batch: 90%% gene queries (%d-%d bp, <=10%% divergence) + 10%% ONT-style reads (5-50 kb)",
                 }.get(wl["kind"], "gene queries (%d-%d bp, <=10%% divergence)") % (lo_len, hi_len)
        seed_B = info["seed_bytes"] / max(info["seeds"], 1)
        result = {
            "metric": "queries/sec (lexicmap search hot path, seed index HBM-resident)",
            "value": round(value, 3), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(step_s * 1e3, 3), "step_ms": step_ms,
            # the cold step: the first search of this batch on the fresh handle (scratch slabs cut, streams created), untimed
            "first_step_ms": (warmup_ms[0] if warmup_ms else step_ms[0]), "warmup_step_ms": warmup_ms, "higher_is_better": True,
            # strong: the batch (and with index sharding the genome set) is fixed as N grows; weak: own batch per GPU
            "scaling": "weak" if weak else "strong",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "gbp_aligned_per_s": round(aligned_total * args.steps / dt / 1e9, 6),
            "config": {"workload": "%s: %d %s vs %d synthetic genomes x %d bp (%d families), M=%d masks, k=%d, index %s" %
                                   (args.workload, nq_total, qdesc, wl["genomes"], wl["genome_len"],
                                    wl["families"], info["masks"], info["k"],
                                    "GPU-built in HBM" if gpu_built else "oracle-built, loaded from the reference format"),
                       "seeds_resident": info["seeds"], "index_hbm_bytes": info["hbm_bytes"],
                       "seed_image_bytes": info["seed_bytes"], "seed_image_bytes_per_seed": round(seed_B, 3),
                       "index_hbm_bytes_per_seed": round(info["hbm_bytes"] / max(info["seeds"], 1), 3),
                       "seed_layout": "partition table (%d bases) + %d-bit k-mer remainder + %d-bit value per seed; %d outlier seeds flat" %
                                      (info["partition_bases"], info["key_bits"], info["val_bits"], info["outlier_seeds"]),
                       "genomes_resident": info["genomes"], "query_bases": query_bases,
                       "parallelism": (("1 GPU holding shard %d of %d (genome g on shard g %% %d, global total_bases): one rank of "
                                        "the %d-GPU index-sharded run" % (args.shard_rank, shard_of, shard_of, shard_of)) if shard_of
                                       else "1 GPU" if world == 1 else
                                       ("index-shard x%d (genome g on rank g %% %d), queries broadcast, one all-gatherv of HSP rows per step" % (world, world))
                                       if index_sharded else
                                       ("q-shard x%d (index replicated), %d queries per GPU" % (world, len(my)))),
                       "pcie_upload_s": round(upload_s, 4), "tag": args.tag, "gather": comm_box[1],
                       "go_toolchain": go or "absent (reference Go binary cannot be built; CPU baseline is the C port)"},
            "host_end_to_end": host_e2e,
            "stage_ms": {k: round(v, 3) for k, v in stats.items() if k.startswith("ms_")},
            "work": {k: v for k, v in stats.items() if not k.startswith("ms_")},
            "rows": rows_total,
            "roofline": roofline,
            "roofline_seed_lookup": roofline_lookup,
            "roofline_pipeline": roofline_pipeline,
            "kernels": kernels,
            "rocprim_calls": prims,
            "sharding_model": shard_model,
            "loader": loader,
            "ab": ab,
            "source_hash": source_hash(),
        }
    # The FULL-SIZE index against the oracle (the sample leg below only meets a second, small index): the search is independent
    # per genome (lib-index-search.go:1537-1556,1743-1747), so (1) the sample queries searched on the full index with the genome
    # whitelist = the sample genomes must give the oracle's rows on the sample index (compared after the CPU leg), and (2) the
    # rows of those genomes inside the timed batch must be the rows of the batch searched under the same whitelist (all
    # columns but `hits`, which counts the genomes of the whole index).
    full_sample_rows = None
    full_check = None
    if rank == 0 and world == 1 and gpu_built and cpu_queries and not shard_of:
        t_f = time.time()
        keys = [genome_key(g) for g in sample_genomes]
        gi.set_genome_filter(keys)
        qb_s = gi.upload([q[1] for q in cpu_queries])
        r_s, _ = gi.search_resident_np(qb_s)
        full_sample_rows = r_s.copy()
        gi.free_batch(qb_s)
        r_b, _ = gi.search_resident_np(qb)
        gi.set_genome_filter(None)
        sel = rows_np[np.isin(rows_np["batch_genome"], np.array(keys, dtype=np.uint64))]
        names = [n for n in rows_np.dtype.names if n not in ("hits", "genome_id", "seq_id", "cigar", "qseq", "sseq", "align")
                 and not n.startswith("pad")]
        same_n = len(sel) == len(r_b)
        same = same_n and all(np.array_equal(sel[n], r_b[n]) for n in names)
        full_check = dict(batch_rows_of_sample_genomes=int(len(sel)), batch_rows_under_whitelist=int(len(r_b)),
                          batch_rows_equal=bool(same), seconds=round(time.time() - t_f, 2))
        log("[rank 0] full-index check: %s" % full_check)
        del r_b, sel
    gi.free_batch(qb)
    # the bench index and its scratch slabs leave HBM before the sample leg: a second handle beside a resident 137-GB index
    # gets a sliver of scratch and searches in tiny chunks (it made the like-for-like number 5x too low)
    gi.close()
    gi = None
    sample_mismatch = None

    def shim_leg(sdir, S):
        """the whole host program as its own process (tests/cabi_shim.c = the Go host's call sequence in C99): opens the subset
        index from disk, reads >= 1000 reads from a FASTA file, searches in batches, formats with lm_format_rows, writes /dev/null"""
        import re
        import subprocess
        try:
            d0 = tempfile.mkdtemp(prefix="lm_shim_")
            exe = os.path.join(d0, "cabi_shim")
            subprocess.check_call(["gcc", "-std=c99", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cabi_shim.c"),
                                   "-L", os.path.dirname(la.LIB_PATH), "-llexicmap_hip", "-Wl,-rpath," + os.path.dirname(la.LIB_PATH), "-o", exe])
            sub2 = la.Index(sdir, device=local_rank)
            nrd = 1000 if wl["kind"] == "reads" else 4000
            fa = os.path.join(d0, "reads.fa")
            with open(fa, "wb") as fh:
                for i in range(nrd):
                    rng = np.random.default_rng([2003, S, i])
                    g = int(rng.integers(0, S))
                    L = min(qlen_of(rng), wl["genome_len"])
                    st = int(rng.integers(0, wl["genome_len"] - L + 1))
                    src = np.frombuffer(sub2.fetch(g, st, L), dtype=np.uint8)
                    fh.write(b">r%05d\n" % i + draw_query(np, synth, rng, src, wl) + b"\n")
            sub2.close()
            r = subprocess.run([exe, "-d", sdir, "-o", os.devnull, fa],  # (one batch: up to 4096 records / 64 Mb by default)
                               capture_output=True, text=True, timeout=900)
            shutil.rmtree(d0, ignore_errors=True)
            if r.returncode != 0:
                return dict(failed=r.stderr[-500:])
            m = re.search(r"timing: (.*)", r.stderr)
            t = {k: float(v) for k, v in (x.split("=") for x in m.group(1).split())}
            return dict(program="tests/cabi_shim.c (C99 restatement of the Go host: reader loop, lm_search_batch, lm_format_rows, one writer)",
                        reads=int(t["queries"]), index_genomes=S, open_s=t["open_s"], search_s=t["search_s"], format_s=t["format_s"],
                        write_s=t["write_s"], rows=int(t["rows"]), tsv_bytes=int(t["tsv_bytes"]),
                        queries_per_s_search_only=round(t["queries"] / max(t["search_s"], 1e-9), 2),
                        queries_per_s_search_format_write=round(t["queries"] / max(t["search_s"] + t["format_s"] + t["write_s"], 1e-9), 2),
                        printer_share=round((t["format_s"] + t["write_s"]) / max(t["search_s"] + t["format_s"] + t["write_s"], 1e-9), 4))
        except Exception as e:  # noqa: BLE001
            return dict(failed=repr(e))

    def cpu_subset_legs():
        """a CPU number beside the GPU number ON THE SAME WORK.  A subset index of S genomes is built on the GPU
        with the workload's generator (same genome length, same ~members per family, same masks parameters), written in the
        reference's on-disk format by lm_index_save and opened RAM-resident by the oracle (= the reference's -w regime,
        kv/kv-searcher2.go:52-88); the same sample reads are searched by both, rows compared, at two sizes."""
        import psutil
        out = []
        per_genome_seeds = 2.0 * (20000 + wl["genome_len"] / 77.0)   # captures + desert seeds, reversed twins
        if args.cpu_subsets == "auto":
            smax = int(8e9 / 16 / per_genome_seeds)                   # <= 8 GB of RAM-resident seeds per oracle process
            sizes = [max(200, smax // 4 // 50 * 50), max(400, smax // 50 * 50)]
            if args.workload == "c3mini":
                sizes = [400, 1600]
        else:
            sizes = [int(x) for x in args.cpu_subsets.split(",") if int(x) > 0]
        members = max(1, wl["genomes"] // wl["families"])
        for S in sizes:
            rec = dict(genomes=S, genome_len=wl["genome_len"])
            sdir = None
            try:
                t_b = time.time()
                sub = la.Index.synthetic(S, wl["genome_len"], max(1, S // members), seed=1000, max_div=0.10, device=local_rank)
                nsq = args.cpu_sample_queries or (64 if wl["kind"] == "reads" else 512)
                qs = []
                for i in range(nsq):
                    rng = np.random.default_rng([2002, S, i])
                    g = int(rng.integers(0, S))
                    L = min(qlen_of(rng), wl["genome_len"])
                    st = int(rng.integers(0, wl["genome_len"] - L + 1))
                    src = np.frombuffer(sub.fetch(g, st, L), dtype=np.uint8)
                    qs.append(("u%05d" % i, draw_query(np, synth, rng, src, wl)))
                sdir = os.path.join(tempfile.mkdtemp(prefix="lm_cpu_subset_"), "subset.lmi")
                sub.save(sdir, chunks=8)
                inf = sub.info()
                rec.update(seeds=int(inf["seeds"]), index_files_GB=round(sum(os.path.getsize(os.path.join(r_, f_)) for r_, _d, fs in os.walk(sdir) for f_ in fs) / 1e9, 2),
                           build_and_save_s=round(time.time() - t_b, 1))
                # the HIP path on the subset (its own handle: the index it built)
                qb3 = sub.upload([q[1] for q in qs])
                sub.search_resident_np(qb3)
                torch.cuda.synchronize()
                t1 = time.time()
                reps = 3
                for _ in range(reps):
                    r3, st3 = sub.search_resident_np(qb3)
                torch.cuda.synchronize()
                rec["gpu_queries_per_s"] = round(nsq * reps / (time.time() - t1), 2)
                rec["gpu_chains_per_query"] = round(st3.get("chains", 0) / nsq, 1)
                r3 = r3.copy()
                sub.free_batch(qb3)
                sub.close()
                if S == max(sizes):
                    rec["host_program"] = shim_leg(sdir, S)
                # the oracle on the host cores, as many processes as the host's free memory holds copies of the RAM-resident index
                rss = inf["seeds"] * 16 + S * wl["genome_len"] * 1.3
                nproc = int(max(1, min(usable_cores(), psutil.virtual_memory().available * 0.6 / rss)))
                c3_ = cpu_baseline(sdir, qs, max(5.0, args.cpu_seconds / 2), nproc)
                orows = c3_.pop("_oracle_rows")
                eq, ncmp, diff = rows_equal_oracle(r3, orows)
                rec.update(cpu_queries_per_s=float(round(c3_["value"], 3)), cores=nproc, cpu_chains_per_query=c3_["chains_per_query"],
                           rows_per_query=round(c3_["rows_per_query"], 1), gpu_over_cpu=round(rec["gpu_queries_per_s"] / max(c3_["value"], 1e-9), 1),
                           rows_equal=bool(eq), rows_compared=int(ncmp), sample=c3_["sample"])
                if not eq:
                    rec["first_difference"] = diff
            except Exception as e:  # noqa: BLE001 (reported in the line)
                rec["failed"] = repr(e)
            finally:
                if sdir:
                    shutil.rmtree(os.path.dirname(sdir), ignore_errors=True)
            log("[rank 0] CPU / GPU on a subset index: %s" % {k: v for k, v in rec.items() if k != "sample"})
            out.append(rec)
        return out

    # CPU baseline on rank 0, N=1 only
    if rank == 0 and world == 1 and not args.no_cpu_baseline and index_dir:
        try:
            ncores = usable_cores()
            cb = cpu_baseline(index_dir, cpu_queries or queries, args.cpu_seconds, ncores)
            cb["sample"] += "; " + cpu_sample_note
            # like for like or not: the seed chains (= windows the pseudo-alignment has to examine) per query on either side
            cb["gpu_chains_per_query"] = round(stats.get("chains", 0) / max(len(queries), 1), 1) if stats else None
            if cb.get("chains_per_query") and cb["gpu_chains_per_query"] and cb["gpu_chains_per_query"] > 2 * cb["chains_per_query"]:
                cb["note"] = ("UPPER BOUND, not a like-for-like ratio: the sample index holds %.0f seed chains per query where the "
                              "timed index holds %.0f - the chain windows of unrelated genomes (random seed matches, rejected by "
                              "the pseudo-alignment) grow with the size of the index and are nearly absent from the sample; the "
                              "like-for-like pair is cpu_baseline.value against gpu_on_same_sample" %
                              (cb["chains_per_query"], cb["gpu_chains_per_query"]))
            oracle_rows = cb.pop("_oracle_rows")
            if True:
                # the same sample (same index directory, same queries) through the HIP path: a like-for-like pair of numbers
                try:
                    sq = cpu_queries or queries
                    g2 = la.Index(index_dir, device=local_rank)
                    qb2 = g2.upload([q[1] for q in sq])
                    g2.search_resident_np(qb2)
                    torch.cuda.synchronize()
                    t1 = time.time()
                    reps = 3
                    for _ in range(reps):
                        r2, _s2 = g2.search_resident_np(qb2)
                    torch.cuda.synchronize()
                    cb["gpu_on_same_sample"] = dict(value=round(len(sq) * reps / (time.time() - t1), 1),
                                                    unit="queries/s", rows_per_query=round(len(r2) / len(sq), 2))
                    # parity at the bench's own shape: the HIP rows of the sample queries against the oracle's, row for row
                    eq, ncmp, diff = rows_equal_oracle(r2, oracle_rows)
                    cb["sample_rows_equal"] = bool(eq)
                    cb["sample_rows"] = int(ncmp)
                    cb["sample_queries_compared"] = len(oracle_rows)
                    if not eq:
                        cb["sample_rows_first_difference"] = diff
                        sample_mismatch = diff
                    if full_sample_rows is not None:
                        # the bench's own full-size index, restricted to the sample genomes, against the oracle's sample rows
                        bg_map = {j: genome_key(g) for j, g in enumerate(sample_genomes)}
                        eq_f, ncmp_f, diff_f = rows_equal_oracle(full_sample_rows, oracle_rows, bg_map)
                        cb["full_index_rows_equal"] = bool(eq_f and full_check["batch_rows_equal"])
                        cb["full_index_rows"] = int(ncmp_f)
                        cb["full_index_check"] = dict(full_check, sample_queries_on_full_index_equal_oracle=bool(eq_f),
                                                      how="sample queries searched on the FULL bench index under the genome whitelist "
                                                          "of the sample genomes vs the oracle on the sample index (same masks, same "
                                                          "database size), row for row; and the timed batch's rows of those genomes "
                                                          "vs the batch searched under the whitelist")
                        if not eq_f:
                            cb["full_index_rows_first_difference"] = diff_f
                            sample_mismatch = sample_mismatch or ("full index: " + diff_f)
                        elif not full_check["batch_rows_equal"]:
                            sample_mismatch = sample_mismatch or "full index: the timed batch's rows of the sample genomes differ from the whitelisted search"
                    g2.free_batch(qb2)
                    g2.close()
                except Exception as e:
                    cb["gpu_on_same_sample"] = "failed: %r" % (e,)
                    cb["sample_rows_equal"] = None
            if gpu_built and args.cpu_subsets != "0" and args.workload in ("c2", "c3", "c3mini"):
                cb["subsets"] = cpu_subset_legs()
                ok_pairs = [x for x in cb["subsets"] if isinstance(x.get("cpu_queries_per_s"), float)]
                if ok_pairs:
                    cb["note_subsets"] = ("LIKE-FOR-LIKE pairs: the same reads on the same index (a subset of the workload's genome set: "
                                          "same generator, same family size) by the oracle on the host cores and by the HIP path, at two "
                                          "index sizes - chains per query are equal on both sides by construction; rows compared")
                    for x in ok_pairs:
                        if x.get("rows_equal") is False:
                            sample_mismatch = sample_mismatch or ("subset of %d genomes: %s" % (x["genomes"], x.get("first_difference")))
            result["cpu_baseline"] = cb
        except Exception as e:  # the baseline must not kill the bench line
            result["cpu_baseline"] = dict(value=None, unit="queries/s", cores=0, kind="port", sample="failed: %r" % (e,))
    if gi is not None:
        gi.close()
    if tmpdir and gpu_built:
        shutil.rmtree(tmpdir, ignore_errors=True)
    if rank == 0:
        print(json.dumps(result), file=real_stdout, flush=True)
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if sample_mismatch:  # a fast path that differs from the reference is not done: fail loudly (the line is printed first)
        raise SystemExit("bench.py: HIP rows differ from the oracle on the CPU sample: %s" % sample_mismatch)


if __name__ == "__main__":
    main()
