"""ctypes binding of tests/libhost_algos.so: the product's per-work-item DEVICE algorithms (lexicmap_amd/csrc/lm_algos.h)
compiled for the host, so their logic can be checked against the oracle without a GPU.  Test infrastructure only."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_algos.cpp")
HDR = os.path.join(os.path.dirname(HERE), "lexicmap_amd", "csrc", "lm_algos.h")
LIB = os.path.join(HERE, "libhost_algos.so")


class Sub(C.Structure):
    _fields_ = [("qbegin", C.c_int32), ("tbegin", C.c_int32), ("len", C.c_uint8), ("trc", C.c_uint8),
                ("qrc", C.c_uint8), ("pad", C.c_uint8)]


class Chain2(C.Structure):
    _fields_ = [("qbegin", C.c_int32), ("qend", C.c_int32), ("tbegin", C.c_int32), ("tend", C.c_int32),
                ("nanchors", C.c_int32), ("matched_bases", C.c_int32), ("aligned_bases_q", C.c_int32),
                ("aligned_bases_t", C.c_int32), ("pident", C.c_double)]


class WfaOut(C.Structure):
    _fields_ = [("status", C.c_int32), ("score", C.c_int32), ("nops", C.c_int32), ("qbegin", C.c_int32),
                ("qend", C.c_int32), ("tbegin", C.c_int32), ("tend", C.c_int32), ("align_len", C.c_uint32),
                ("matches", C.c_uint32), ("gaps", C.c_uint32), ("gap_regions", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if (not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(HDR))):
            subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                                   "-o", LIB, SRC])
        L = C.CDLL(LIB)
        L.ha_revcomp.restype = C.c_uint64
        L.ha_revcomp.argtypes = [C.c_uint64, C.c_int]
        L.ha_reverse.restype = C.c_uint64
        L.ha_reverse.argtypes = [C.c_uint64, C.c_int]
        L.ha_dust.argtypes = [C.c_uint64, C.c_int]
        L.ha_low_complexity.argtypes = [C.c_uint64, C.c_int]
        L.ha_xor_argmin.restype = C.c_uint64
        L.ha_xor_argmin.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.c_uint64, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ha_pack_anchor.restype = C.c_uint64
        L.ha_unpack_anchor.argtypes = [C.c_uint64, C.POINTER(Sub)]
        L.ha_clear_sorted.argtypes = [C.POINTER(Sub), C.c_int, C.c_int]
        L.ha_chain1.restype = C.c_float
        L.ha_chain1.argtypes = [C.POINTER(Sub), C.c_int, C.c_float, C.c_float, C.c_float, C.c_int,
                                C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                C.POINTER(C.c_int)]
        L.ha_trim.argtypes = [C.POINTER(Sub), C.c_int, C.c_float, C.POINTER(C.c_int)]
        L.ha_chain2.argtypes = [C.POINTER(Sub), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                C.POINTER(Chain2)]
        L.ha_extend_match.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int] + [C.c_int] * 8 + [C.POINTER(C.c_int)]
        L.ha_extend_flank_both.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.ha_build_tab.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32)]
        L.ha_tree_search_first_tab.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.c_uint64, C.c_int, C.c_int,
                                               C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ha_tree_search_range.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.c_uint64, C.c_int, C.c_int,
                                           C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ha_wfa.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_uint64),
                             C.c_int, C.POINTER(WfaOut)]
        L.ha_pa_filter_build.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32)]
        L.ha_pa_candidate.argtypes = [C.POINTER(C.c_uint32), C.c_int, C.c_uint64, C.c_int, C.c_int]
        L.ha_pa_candidate2.argtypes = [C.POINTER(C.c_uint32), C.c_int, C.c_uint64, C.c_int, C.c_int]
        L.ha_pa_bits_words.argtypes = [C.c_int]
        L.ha_pa_bits_words.restype = C.c_uint64
        L.ha_bits_get.restype = C.c_uint64
        L.ha_bits_get.argtypes = [C.POINTER(C.c_uint64), C.c_int64, C.c_int]
        L.ha_bits_store_range.argtypes = [C.POINTER(C.c_uint64), C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_uint64)]
        L.ha_partition_range.restype = C.c_int32
        L.ha_partition_range.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64,
                                         C.POINTER(C.c_int64)]
        L.ha_pack_seed_val.restype = C.c_uint64
        L.ha_pack_seed_val.argtypes = [C.c_uint64, C.c_uint64, C.c_int]
        L.ha_unpack_seed_val.restype = C.c_uint64
        L.ha_unpack_seed_val.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int]
        _lib = L
    return _lib
