"""lexicmap_amd — MI355X-native `lexicmap search` hot path.

The product is the C-ABI shared library lexicmap_amd/liblexicmap_hip.so (include/lexicmap_hip.h), built from
lexicmap_amd/csrc by `__graft_entry__.build()` / `make -C lexicmap_amd/csrc`.  This package is only the ctypes view of
that ABI used by tests/ and bench.py; the host logic above the ABI is C++ inside the library.
"""
from .api import (HipLibraryMissing, Index, Options, build_library, lib, LIB_PATH)  # noqa: F401
