"""CPU-side checks of the product: the C-ABI library loads without a GPU, exports every symbol the header declares,
fails loudly (no CPU fallback), and the host-side pieces that need no GPU behave (options, row formatting)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "lexicmap_hip.h")


def _declared_functions():
    txt = open(HDR).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lm_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import lexicmap_amd as la
    la.build_library()
    out = subprocess.check_output(["nm", "-D", "--defined-only", la.LIB_PATH]).decode()
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    declared = _declared_functions()
    assert len(declared) >= 20
    missing = [f for f in declared if f not in exported]
    assert not missing, missing


def test_header_is_plain_c_and_the_shim_sequence_links(tmp_path):
    """f2: cgo compiles its preamble with a C compiler. The header must be valid C99 (no C++-isms, no duplicate
    typedefs) and the call sequence of the cgo shim of INTEGRATION.md - restated in C in tests/cabi_shim.c, since no Go
    toolchain exists here - must compile without warnings and link against the library with C linkage."""
    import lexicmap_amd as la
    la.build_library()
    inc = os.path.join(ROOT, "include")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", HDR],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    exe = str(tmp_path / "cabi_shim")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc,
                        os.path.join(ROOT, "tests", "cabi_shim.c"), "-L", os.path.dirname(la.LIB_PATH), "-llexicmap_hip",
                        "-Wl,-rpath," + os.path.dirname(la.LIB_PATH), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # without a GPU the shim must fail the way the Go shim would report it: status != OK and the library's message
    import torch
    if not torch.cuda.is_available():
        fa = tmp_path / "q.fa"
        fa.write_text(">q\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n")
        r = subprocess.run([exe, str(tmp_path), str(fa)], capture_output=True, text=True)
        assert r.returncode == 255 and "no CPU path" in r.stderr      # checkError: message + exit status -1
        # the flag checks of search.go:159-229 run before the index is opened: the reference's messages
        for args, msg in ((["q.fa"], "flag -d/--index needed"),
                          (["-d", "x", "-p", "4", "q.fa"], "-p/--seed-min-prefix (4) should be in the range of [5, 32]"),
                          (["-d", "x", "-p", "20", "-P", "18", "q.fa"], "should be >= that of -p/--seed-min-prefix (20)"),
                          (["-d", "x", "-l", "10", "q.fa"], "-l/--align-min-match-len (10) should be >="),
                          (["-d", "x", "-Q", "101", "q.fa"], "-Q/--min-qcov-per-genome"),
                          (["-d", "x", "--frobnicate", "q.fa"], "unknown flag")):
            r = subprocess.run([exe] + args, capture_output=True, text=True)
            assert r.returncode == 255 and msg in r.stderr, (args, r.stderr)


def test_no_gpu_means_loud_failure_not_fallback(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import lexicmap_amd as la
    L = la.lib()
    o = la.api.default_options()
    h = C.c_void_p()
    st = L.lm_index_open(str(tmp_path).encode(), C.byref(o), 0, C.byref(h))
    assert st == 4 and not h  # LM_ERR_NO_DEVICE
    assert b"no CPU path" in L.lm_last_error(None)
    with pytest.raises(RuntimeError):
        la.Index.synthetic(4, 1000, 2)
    # the row gather (RCCL between GPUs) likewise: no device, no communicator
    buf = C.create_string_buffer(128)
    assert L.lm_comm_unique_id(buf) == 4
    hc = C.c_void_p()
    assert L.lm_comm_init(buf, 1, 0, 0, C.byref(hc)) == 4 and not hc
    assert b"RCCL" in L.lm_comm_last_error(None)
    assert L.lm_comm_init(buf, 2, 5, 0, C.byref(hc)) == 7  # LM_ERR_ARG: rank outside the communicator


def test_option_defaults_match_reference_flags():
    """search.go:631-731"""
    import lexicmap_amd as la
    o = la.api.default_options()
    assert (o.min_prefix, o.min_single_prefix, o.top_n_genomes, o.top_n_chains) == (15, 17, 0, 0)
    assert (o.max_gap, o.max_distance, o.ext_len, o.ext_len2) == (50.0, 1000.0, 1000, 50)
    assert (o.align_max_gap, o.align_band, o.align_min_match_len, o.align_min_pident) == (20, 100, 50, 70.0)
    assert (o.max_evalue, o.min_qcov_per_hsp, o.min_qcov_per_genome, o.output_seq) == (10.0, 0.0, 0.0, 0)


def test_row_formatting_matches_reference_printf():
    """search.go:506-516 format string, checked on the first row of demo/q.gene.fasta.lexicmap.tsv"""
    import lexicmap_amd as la
    L = la.lib()
    r = la.api.Hsp()
    r.query, r.hits, r.qcov_genome, r.cls, r.hsp = 0, 15, 100.0, 1, 1
    r.qcov_hsp, r.aligned_length, r.pident, r.gaps = 100.0, 1542, 1539 / 1542 * 100, 0
    r.qbegin, r.qend, r.tbegin, r.tend, r.rc, r.seq_len = 0, 1541, 458558, 460099, 0, 4903501
    r.evalue, r.bitscore = 0.0, 2767
    r.genome_id, r.seq_id = b"GCF_003697165.2", b"NZ_CP033092.2"
    buf = C.create_string_buffer(4096)
    L.lm_format_row(C.byref(r), b"NC_000913.3:4166659-4168200", 1542, 0, buf, 4096)
    gold = open(os.path.join(ROOT, "tests", "golden", "demo", "q.gene.fasta.lexicmap.tsv")).read().split("\n")[1]
    assert buf.value.decode() == gold
    r.evalue = 1.72e-43
    L.lm_format_row(C.byref(r), b"q", 10, 0, buf, 4096)
    assert buf.value.decode().split("\t")[18] == "1.72e-43"
    # --show-sseq-idx (search.go:483-494) and -a/--all (:519-521); the header line (:426-430) is the golden's first line
    r.chunk_idx, r.nchunks, r.seq_idx, r.nseqs = 1, 3, 0, 10
    r.cigar, r.qseq, r.sseq, r.align = b"10M", b"ACGTACGTAC", b"ACGTACGTAC", b"||||||||||"
    L.lm_format_row_ex.argtypes = L.lm_format_row.argtypes
    L.lm_format_row_ex(C.byref(r), b"q", 10, 2, buf, 4096)
    cols = buf.value.decode().split("\t")
    assert cols[4] == "c2/3:s1/10:NZ_CP033092.2" and len(cols) == 20
    L.lm_format_row_ex(C.byref(r), b"q", 10, 3, buf, 4096)
    cols = buf.value.decode().split("\t")
    assert cols[4] == "c2/3:s1/10:NZ_CP033092.2" and cols[20:] == ["10M", "ACGTACGTAC", "ACGTACGTAC", "||||||||||"]
    L.lm_tsv_header.restype = C.c_char_p
    L.lm_tsv_header.argtypes = [C.c_int]
    gold_dir = os.path.join(ROOT, "tests", "golden", "demo")
    assert L.lm_tsv_header(0).decode() == open(os.path.join(gold_dir, "q.gene.fasta.lexicmap.tsv")).readline().rstrip("\n")
    assert L.lm_tsv_header(1).decode() == open(os.path.join(gold_dir, "q.gene.fasta.lexicmap_top-2-genomes_all.tsv")).readline().rstrip("\n")


def test_oracle_is_not_referenced_by_the_product():
    """the product path must not import, link or call anything under oracle/"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "lexicmap_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liblmo" not in txt and "lmo_" not in txt and "import oracle" not in txt, os.path.join(dirpath, f)
    import lexicmap_amd as la
    out = subprocess.check_output(["ldd", la.LIB_PATH]).decode()
    assert "liblmo" not in out
