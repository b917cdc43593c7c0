"""f2: `lexicmap search` as the Go host of INTEGRATION.md would run it, restated in C99 over the C-ABI
(tests/cabi_shim.c: the flags of search.go:159-229, the reader loop of :548-608 as a batch loop with repeated flushes, the
printer of :426-533 with -a/--all and --show-sseq-idx) and run as its OWN PROCESS on the GPU against the reference's own
golden TSVs (tests/golden/demo), byte for byte including the header line - not against another view of the HIP path."""
import os
import subprocess

import pytest

import oracle as O

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "demo")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    import lexicmap_amd as la
    exe = str(tmp_path_factory.mktemp("shim") / "cabi_shim")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cabi_shim.c"), "-L", os.path.dirname(la.LIB_PATH), "-llexicmap_hip",
                           "-Wl,-rpath," + os.path.dirname(la.LIB_PATH), "-o", exe])
    return exe


@pytest.fixture(scope="module")
def demo15(tmp_path_factory):
    """all 15 genomes of demo/refs (committed fixtures) indexed by the oracle's writer"""
    d = str(tmp_path_factory.mktemp("demo") / "demo15.lmi")
    files = [os.path.join(GOLD, f) for f in os.listdir(GOLD) if f.endswith(".fa.gz")]
    files += [os.path.join(GOLD, "refs", f) for f in os.listdir(os.path.join(GOLD, "refs")) if f.endswith(".fa.gz")]
    files = sorted(files, key=os.path.basename)
    assert len(files) == 15
    O.build_index(d, [(os.path.basename(f)[:-6], O.read_fasta(f)) for f in files], O.default_build_opt(chunks=8))
    return d


def run(exe, args):
    r = subprocess.run([exe] + args, capture_output=True, text=True)
    if r.returncode != 0:   # (what the device looked like to the other process: a failure here has been a starved scratch budget)
        import torch
        free, total = torch.cuda.mem_get_info()
        raise AssertionError("%s\n[device memory seen from the test process: %.1f of %.1f GB free]" % (r.stderr, free / 1e9, total / 1e9))
    return r.stdout, r.stderr


def test_shim_prints_the_reference_golden_tsv(shim, demo15):
    """BASELINE configs[0] through the shim binary: the file the reference's own run wrote (header + 84 rows)"""
    out, err = run(shim, ["-d", demo15, os.path.join(GOLD, "q.gene.fasta")])
    gold = open(os.path.join(GOLD, "q.gene.fasta.lexicmap.tsv")).read()
    assert gold.count("\n") == 85 and out == gold
    assert "processed queries: 1\n" in err and "100.0000% (1/1) queries matched" in err


def test_shim_reader_loop_flushes_batches_counts_short_records_and_reads_fastq(shim, demo15, tmp_path):
    """several files, a flush after every record (--batch-queries 1) and after every 2 kb (--batch-bases), a record shorter
    than k (counted, not searched: search.go:571-575), lower-case FASTA split over lines (upper-cased: :580-587), FASTQ:
    every record of the gene prints the golden rows under its own id, in input order"""
    gid, gseq = O.read_fasta(os.path.join(GOLD, "q.gene.fasta"))[0]
    gold = open(os.path.join(GOLD, "q.gene.fasta.lexicmap.tsv")).read().split("\n")
    header, rows = gold[0], [r for r in gold[1:] if r]
    low = gseq.decode().lower()
    f2 = tmp_path / "more.fa"
    f2.write_text(">short too short to hold a k-mer\nACGTACGTAC\n>lc lower case, 60 per line\n" +
                  "\n".join(low[i:i + 60] for i in range(0, len(low), 60)) + "\n>nohit\n" + "ACGTTGCA" * 40 + "\n")
    f3 = tmp_path / "reads.fq"
    f3.write_text("@fq1 a FASTQ record\n%s\n+\n%s\n@fq2\n%s\n+fq2\n%s\n" % (gseq.decode(), "I" * len(gseq), low, "@" * len(gseq)))
    want = [header] + rows
    for qid in ("lc", "fq1", "fq2"):
        want += [qid + r[len(gid):] for r in rows]
    # five records reach the library (1542, 1542, 320, 1542, 1542 bases): a flush per record / whenever 2 000 bases have
    # gathered (after the 2nd and the 5th) / one at the end
    for extra, batches in ((["--batch-queries", "1"], 5), (["--batch-bases", "2000"], 2), ([], 1)):
        out, err = run(shim, ["-d", demo15] + extra + [os.path.join(GOLD, "q.gene.fasta"), str(f2), str(f3)])
        assert out.split("\n")[:-1] == want, extra
        assert "processed queries: 6\n" in err and "(4/6) queries matched" in err
        assert int(err.split("batches=")[1].split()[0]) == batches, (extra, err)


def test_shim_all_columns_and_top_n_print_the_reference_golden(shim, tmp_path):
    """-a -n 2: demo/q.gene.fasta.lexicmap_top-2-genomes_all.tsv (CIGAR, qseq, sseq, align columns) byte for byte"""
    d = str(tmp_path / "demo2.lmi")
    genomes = [(f[:-6], O.read_fasta(os.path.join(GOLD, f))) for f in ("GCF_002949675.1.fa.gz", "GCF_003697165.2.fa.gz")]
    O.build_index(d, genomes, O.default_build_opt(chunks=4))
    out, _ = run(shim, ["--index", d, "--all", "--top-n-genomes", "2", os.path.join(GOLD, "q.gene.fasta")])
    assert out == open(os.path.join(GOLD, "q.gene.fasta.lexicmap_top-2-genomes_all.tsv")).read()


def test_shim_show_sseq_idx_and_option_flags(shim, demo15):
    """--show-sseq-idx (search.go:483-494): sseqid = c<chunk>/<chunks>:s<seq>/<seqs>:<id> from the row's own fields; the
    filter flags reach lm_options (rows = the ctypes path's under the same options)"""
    import lexicmap_amd as la
    gid, gseq = O.read_fasta(os.path.join(GOLD, "q.gene.fasta"))[0]
    gi = la.Index(demo15, la.api.default_options(min_qcov_per_hsp=90.0, align_min_pident=85.0, top_n_chains=2))
    rows, _ = gi.search([gseq])
    want = gi.search_tsv([gid], [gseq])
    gi.close()
    out, _ = run(shim, ["-d", demo15, "--show-sseq-idx", "-q", "90", "-i", "85", "-N", "2", os.path.join(GOLD, "q.gene.fasta")])
    got = out.split("\n")[1:-1]
    assert len(got) == len(want) == len(rows) and 0 < len(rows) < 84
    for g, w, r in zip(got, want, rows):
        gc, wc = g.split("\t"), w.split("\t")
        assert gc[4] == "c%d/%d:s%d/%d:%s" % (r["chunk_idx"] + 1, r["nchunks"], r["seq_idx"] + 1, r["nseqs"], wc[4])
        assert gc[:4] + gc[5:] == wc[:4] + wc[5:]


def test_shim_fails_like_checkerror(shim, demo15):
    """flag checks of search.go:159-229 with the reference's messages and exit status (util-cli.go:35: -1)"""
    for args, msg in ((["q.fa"], "flag -d/--index needed"),
                      (["-d", demo15, "-p", "4", "q.fa"], "-p/--seed-min-prefix (4) should be in the range of [5, 32]"),
                      (["-d", demo15, "-p", "20", "-P", "18", "q.fa"], "should be >= that of -p/--seed-min-prefix (20)"),
                      (["-d", demo15, "-i", "50", "q.fa"], "-i/--align-min-match-pident"),
                      (["-d", demo15, "--align-band", "10", "--align-max-gap", "20", "q.fa"], "--align-band should not be smaller"),
                      (["-d", os.path.join(demo15, "nope"), "q.fa"], "info.toml")):
        r = subprocess.run([shim] + args, capture_output=True, text=True)
        assert r.returncode == 255 and msg in r.stderr, (args, r.stderr)
