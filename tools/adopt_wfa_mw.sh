#!/bin/bash
# Wires k_wfa_mw (experiments/wfa_row, four wavefronts per long WFA alignment) into the product sources: applies the staged
# patch, adds its two headers to the Makefile rule of lm_kernels.o, rebuilds the library and runs the CPU tests.  Afterwards:
# tools/validate_gpu.sh (GPU tests), tools/final_runs.sh / tools/profile.sh (bench + counter passes on the new source hash).
set -euo pipefail
cd "$(dirname "$0")/.."
python experiments/wfa_row/make_integrated.py --no-build >/dev/null   # the patch against the CURRENT sources
rm -rf experiments/csrc_mw
git apply --check experiments/wfa_row/integrate_mw.patch
git apply experiments/wfa_row/integrate_mw.patch
sed -i 's|^lm_kernels.o: lm_kernels.hip lm_kernels.h lm_algos.h$|lm_kernels.o: lm_kernels.hip lm_kernels.h lm_algos.h lm_wfa_mw.h lm_wfa_mw_fwd.h|' lexicmap_amd/csrc/Makefile
grep -q "lm_wfa_mw.h lm_wfa_mw_fwd.h" lexicmap_amd/csrc/Makefile
python -c "import __graft_entry__ as g; g.build()"
python -m pytest tests -x -q -m "not gpu"
echo "k_wfa_mw is in lexicmap_amd/csrc (switch: LM_WFA_MW=0 turns it off).  Next: GPU tests, C2 / C3 bench, new profiles."
