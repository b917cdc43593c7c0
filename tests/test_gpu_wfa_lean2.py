"""GPU parity of k_wfa_lean2 (the LDS wavefront kernel): every instantiation forced through lm_wfa_batch against
the oracle's lmo_wfa_align - ring widths 64-1024 diagonals, 16- and 32-bit cells, whole sequences and sliding windows, drifting
wavefronts (the ring is recentred), pairs that outgrow a ring.  The pairs and the checks: tests/wfa_lean2_gpu_check.py."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_every_instantiation_equals_the_oracle():
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("wfa_lean2_gpu_check", os.path.join(here, "wfa_lean2_gpu_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad, report = mod.main(timing=False)
    assert bad == 0, report
    assert report and all(v["kernels"] for v in report.values())
