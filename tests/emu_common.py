"""What the emulator tests share: where the host SIMT emulator and its harnesses live (tests/emu), the result record of a
harness run, and the wide-wavefront pair generator."""
import ctypes as C
import os

from test_device_algos_cpu import mutate, rand_seq

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "emu")   # the host SIMT emulator (simt_emu.h) and the harnesses built on it


class EmuOut(C.Structure):
    _fields_ = [("status", C.c_int32), ("score", C.c_int32), ("nops", C.c_int32), ("qbegin", C.c_int32), ("qend", C.c_int32),
                ("tbegin", C.c_int32), ("tend", C.c_int32), ("align_len", C.c_uint32), ("matches", C.c_uint32),
                ("gaps", C.c_uint32), ("gap_regions", C.c_uint32), ("used", C.c_int32)]


def with_insertion(rng, q, at, n, div):
    """target = mutated query with n extra bases at `at` (-1: at the END - the final diagonal n is then never trimmed away
    (lm_wfa_align keeps the range open towards it), so the wavefront grows one diagonal per score all along the alignment
    and ends up |n| + ~50 wide: the way the 512 / 1024-diagonal rings get used)"""
    t = mutate(rng, q, div, div / 4, div / 4)
    return t + rand_seq(rng, n) if at < 0 else t[:at] + rand_seq(rng, n) + t[at:]
