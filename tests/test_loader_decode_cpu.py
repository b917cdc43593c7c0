"""The loader's host-side chunk decoder (lm_format.cpp: decode_seed_chunk - group-varint k-mer deltas and value counts, 7-byte
values, read with unaligned 8-byte loads from a padded copy of the file into a RE-USED chunk slot) against the oracle's reader of
the same files (lmo_kv_load = kv-reader.go:762-1021): every (mask, k-mer, value) of every chunk, slots re-used across files of
different sizes, the seeds of a shard only, and truncated files refused instead of read past their end."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "format_host.cpp")
CSRC = os.path.join(os.path.dirname(HERE), "lexicmap_amd", "csrc")
LIB = os.path.join(HERE, "libformat_host.so")


class _KvMem(C.Structure):
    _fields_ = [("k", C.c_int), ("chunk_index", C.c_int), ("chunk_size", C.c_int), ("mask_prefix", C.c_int),
                ("anchor_prefix", C.c_int), ("use7", C.c_int), ("kv", C.POINTER(C.POINTER(C.c_uint64))),
                ("kvlen", C.POINTER(C.c_int64)), ("index", C.POINTER(C.POINTER(C.c_int64)))]


@pytest.fixture(scope="module")
def F():
    deps = [SRC, os.path.join(CSRC, "lm_format.cpp"), os.path.join(CSRC, "lm_format.h")]
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-o", LIB, SRC, os.path.join(CSRC, "lm_format.cpp")])
    lib = C.CDLL(LIB)
    lib.fh_error.restype = C.c_char_p
    lib.fh_file.restype = C.c_char_p
    lib.fh_decode.restype = C.c_longlong
    lib.fh_decode.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.POINTER(C.c_uint16))]
    return lib


@pytest.fixture(scope="module")
def index(tmp_path_factory):
    from lexicmap_amd import synth
    d = str(tmp_path_factory.mktemp("ldidx") / "i.lmi")
    genomes = synth.make_genomes(12, 60000, 3, seed=17, max_div=0.08, contigs=(1, 3))
    O.build_index(d, genomes, O.default_build_opt(chunks=5))   # 5 files of unequal size (20 000 masks)
    return d


def decode(F, path):
    k, v, m = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint16)()
    n = F.fh_decode(path, C.byref(k), C.byref(v), C.byref(m))
    if n < 0:
        return n, F.fh_error().decode()
    as_np = lambda p, t: np.ctypeslib.as_array(p, shape=(n,)).astype(t) if n else np.zeros(0, t)
    return n, (as_np(k, np.uint64), as_np(v, np.uint64), as_np(m, np.uint16))


def oracle_seeds(path):
    L = O.lib()
    L.lmo_kv_load.restype = C.POINTER(_KvMem)
    L.lmo_kv_load.argtypes = [C.c_char_p]
    L.lmo_kv_free.argtypes = [C.POINTER(_KvMem)]
    m = L.lmo_kv_load(path)
    km = m.contents
    out = []
    for i in range(km.chunk_size):
        n = km.kvlen[i]
        if n:
            flat = np.ctypeslib.as_array(km.kv[i], shape=(n,))
            out += [(km.chunk_index + i, int(a), int(b)) for a, b in zip(flat[0::2], flat[1::2])]
    L.lmo_kv_free(m)
    return sorted(out)


def test_every_seed_of_every_chunk_with_one_reused_slot(F, index):
    assert F.fh_open(index.encode(), 0, 1) == 0, F.fh_error()
    files = [F.fh_file(i) for i in range(F.fh_nfiles())]
    assert len(files) == 5
    total = 0
    for path in files + files[::-1]:   # a smaller file after a larger one leaves the slot's arrays longer than its seeds
        n, (k, v, m) = decode(F, path)
        got = sorted(zip(m.tolist(), k.tolist(), v.tolist()))
        assert got == oracle_seeds(path) and n > 1000
        total += n
    assert total > 100000


def test_a_shard_keeps_only_its_genomes(F, index):
    per_shard = []
    for r in range(3):
        assert F.fh_open(index.encode(), r, 3) == 0, F.fh_error()
        n, (k, v, m) = decode(F, F.fh_file(0))
        per_shard.append(set(zip(m.tolist(), k.tolist(), v.tolist())))
    assert F.fh_open(index.encode(), 0, 1) == 0
    whole = set(oracle_seeds(F.fh_file(0)))
    assert set.union(*per_shard) == whole and sum(len(s) for s in per_shard) == len(whole)
    for r in range(3):   # genome g (batch 0: the value's genome field) lives on shard g % 3
        assert all(((v >> 30) & 0x1ffff) % 3 == r for _m, _k, v in per_shard[r])


def test_truncated_and_foreign_files_are_refused(F, index, tmp_path):
    assert F.fh_open(index.encode(), 0, 1) == 0
    src = F.fh_file(1).decode()
    data = open(src, "rb").read()
    bad = str(tmp_path / "chunk_001.bin")
    shutil.copy(src + ".idx", bad + ".idx")
    for cut in (len(data) // 2, len(data) - 3, len(data) - 1, 41, 33, 31):
        open(bad, "wb").write(data[:cut])
        n, err = decode(F, bad.encode())
        assert n < 0 and ("broken" in err or "invalid" in err), (cut, n, err)
    open(bad, "wb").write(b"notakvfile" + data[10:])
    n, err = decode(F, bad.encode())
    assert n == -2 and "invalid binary format" in err
    open(bad, "wb").write(data)   # and the untouched copy decodes again (the slot survived the failures)
    n, (k, v, m) = decode(F, bad.encode())
    assert sorted(zip(m.tolist(), k.tolist(), v.tolist())) == oracle_seeds(src.encode())


def _genomes_of(F):
    out = {}
    for i in range(F.fh_ngenomes()):
        bg, ln, ns, gid = C.c_uint64(), C.c_int(), C.c_int(), C.c_char_p()
        F.fh_genome.restype = C.POINTER(C.c_ubyte)
        p = F.fh_genome(i, C.byref(bg), C.byref(ln), C.byref(ns), C.byref(gid))
        packed = bytes(p[:(ln.value + 3) // 4])
        bases = "".join("ACGT"[(packed[j >> 2] >> ((3 - (j & 3)) << 1)) & 3] for j in range(ln.value))
        out[bg.value] = (gid.value.decode(), ln.value, ns.value, bases)
    others = {}
    for i in range(F.fh_nothers()):
        bg, ln, ns, gid = C.c_uint64(), C.c_int(), C.c_int(), C.c_char_p()
        F.fh_other(i, C.byref(bg), C.byref(ln), C.byref(ns), C.byref(gid))
        others[bg.value] = (gid.value.decode(), ln.value, ns.value)
    return out, others


@pytest.mark.parametrize("run_bytes,head_bytes", [(None, None), (40000, 64), (1, 1), (20000, 100000)])
def test_genome_batches_read_in_runs_and_a_shard_reads_only_the_heads_of_foreign_records(F, index, monkeypatch, run_bytes, head_bytes):
    """load_index_genomes reads a batch file in runs of consecutive records (LM_LOADER_RUN_BYTES: here a few records or one
    per run) and, of a record that belongs to another shard, only its head (LM_LOADER_HEAD_BYTES: here too short for the head,
    so the whole record is read after all): names, lengths, contig counts and every base of every local genome equal what the
    index was built from, whatever the run and head sizes, unsharded and on each of three shards"""
    from lexicmap_amd import synth
    genomes = synth.make_genomes(12, 60000, 3, seed=17, max_div=0.08, contigs=(1, 3))
    if run_bytes is not None:
        monkeypatch.setenv("LM_LOADER_RUN_BYTES", str(run_bytes))
        monkeypatch.setenv("LM_LOADER_HEAD_BYTES", str(head_bytes))
    want = {}
    for g, (gid, contigs) in enumerate(genomes):
        want[g] = (gid, len(contigs))
    assert F.fh_open(index.encode(), 0, 1) == 0, F.fh_error()
    loc, oth = _genomes_of(F)
    assert len(loc) == 12 and not oth
    for bg, (gid, ln, ns, bases) in loc.items():
        g = bg & 0x1ffff
        assert (gid, ns) == want[g]
        seqs = [c[1].decode() if isinstance(c[1], bytes) else c[1] for c in genomes[g][1]]
        # contigs are joined by the reference's interval of N (stored as A): the contig bases must appear in order
        pos = 0
        for sq in seqs:
            k = bases.find(sq, pos)
            assert k >= 0, (g, len(sq))
            pos = k + len(sq)
    whole = loc
    for r in range(3):
        assert F.fh_open(index.encode(), r, 3) == 0, F.fh_error()
        loc, oth = _genomes_of(F)
        assert sorted(bg & 0x1ffff for bg in loc) == [g for g in range(12) if g % 3 == r]
        assert sorted(bg & 0x1ffff for bg in oth) == [g for g in range(12) if g % 3 != r]
        for bg, v in loc.items():
            assert v == whole[bg]
        for bg, (gid, ln, ns) in oth.items():
            assert (gid, ln, ns) == whole[bg][:3]


@pytest.mark.parametrize("run_bytes", [None, 40000, 1])
def test_the_bases_handed_to_a_sink_are_the_host_store(F, index, monkeypatch, run_bytes):
    """with a sink (the loader: the device, zero-filled beforehand) load_index_genomes hands every local genome's packed bases
    over with the offset it has in the store: what the sink received is the host store of the sink-less reader, byte for byte,
    on every shard, for runs of all, a few and one record"""
    if run_bytes is not None:
        monkeypatch.setenv("LM_LOADER_RUN_BYTES", str(run_bytes))
    F.fh_store.restype = C.POINTER(C.c_ubyte)
    F.fh_store_bytes.restype = C.c_longlong
    F.fh_sink_calls.restype = C.c_long
    for r, n in ((0, 1), (0, 3), (1, 3), (2, 3)):
        assert F.fh_open(index.encode(), r, n) == 0, F.fh_error()
        nb = F.fh_store_bytes()
        host = bytes(F.fh_store()[:nb])
        loc_h, oth_h = _genomes_of(F)
        assert F.fh_open_sink(index.encode(), r, n) == 0, F.fh_error()
        assert F.fh_store_bytes() == nb
        assert bytes(F.fh_store()[:nb]) == host
        loc_s, oth_s = _genomes_of(F)
        assert loc_s == loc_h and oth_s == oth_h
        assert F.fh_sink_calls() == len(loc_h)
