"""Index format versions (CPU, oracle side): lib-index-search.go:1212-1215 masks the queries of an index older than format 3.5
with lexichash's MaskKnownDistinctPrefixesWithStrandBias - a function of the un-vendored lexichash module that is not restated
in oracle/ - so such an index must be REFUSED by the checker (and by the library: tests/test_gpu_parity.py), never searched
with the 3.5 masking."""
import os
import re
import shutil

import pytest

import oracle as O


def test_oracle_refuses_an_index_older_than_format_3_5(tmp_path):
    from lexicmap_amd import synth
    d = str(tmp_path / "i.lmi")
    genomes = synth.make_genomes(2, 30000, 1, seed=9, max_div=0.03)
    O.build_index(d, genomes, O.default_build_opt(chunks=2))
    oi = O.Index(d)
    rows, _ = oi.search(genomes[0][1][0][1][2000:3200])
    oi.close()
    assert rows
    t = open(os.path.join(d, "info.toml")).read()
    assert "main-version = 3\n" in t and "minor-version = 5\n" in t
    for name, text in (("v34", t.replace("minor-version = 5\n", "minor-version = 4\n")),
                       ("nominor", t.replace("minor-version = 5\n", "")),
                       ("v2", t.replace("main-version = 3\n", "main-version = 2\n"))):
        o = str(tmp_path / (name + ".lmi"))
        shutil.copytree(d, o)
        open(os.path.join(o, "info.toml"), "w").write(text)
        with pytest.raises(RuntimeError):
            O.Index(o)
