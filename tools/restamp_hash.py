#!/usr/bin/env python3
"""tools/restamp_hash.py <commit> <profiles/file.json ...>: the counter-pass summaries of a round are stamped with the hash of the
library sources they were taken on (bench.py refuses passes taken on other sources).  Until round 5 that hash was over the raw
bytes, so a COMMENT fixed afterwards orphaned the passes.  This tool checks, for the sources of <commit> (the commit the passes
ran on): (1) their raw hash IS the stamp in the file, (2) their code-only hash (bench.source_hash: comments and whitespace
stripped) EQUALS the code-only hash of the working tree - i.e. nothing but comments changed since - and only then rewrites the
stamp to the code-only hash, keeping the raw one as `source_hash_raw_at_run`.  Anything else: it refuses."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

commit, files = sys.argv[1], sys.argv[2:]
tmp = tempfile.mkdtemp(prefix="lm_restamp_")
d = os.path.join(tmp, "lexicmap_amd", "csrc")
os.makedirs(d)
names = subprocess.check_output(["git", "ls-tree", "--name-only", commit, "lexicmap_amd/csrc/"], cwd=ROOT, text=True).split()
for n in names:
    if n.endswith((".hip", ".h", ".cpp")):
        open(os.path.join(d, os.path.basename(n)), "wb").write(subprocess.check_output(["git", "show", "%s:%s" % (commit, n)], cwd=ROOT))
raw_then, code_then, code_now = bench.source_hash_raw(tmp), bench.source_hash(tmp), bench.source_hash()
print("sources of %s: raw %s, code-only %s; working tree code-only %s" % (commit, raw_then, code_then, code_now))
if code_then != code_now:
    sys.exit("the CODE changed since %s: the passes are not this tree's - refused" % commit)
for f in files:
    text = open(f).read()
    lines = text.strip().splitlines()
    whole = None
    try:
        whole = json.loads(text)   # a summary (one indented document)
        doc = whole
    except ValueError:
        doc = json.loads(lines[-1])  # a bench output: the JSON line is the last one
    if doc.get("source_hash") == code_now:
        print(f, "already stamped")
        continue
    if doc.get("source_hash") != raw_then and doc.get("source_hash_raw_at_run") != raw_then:
        sys.exit("%s is stamped %s, not the raw hash of %s (%s) - refused" % (f, doc.get("source_hash"), commit, raw_then))
    doc["source_hash_raw_at_run"] = raw_then
    doc["source_hash"] = code_now
    doc["source_hash_note"] = "code-only hash (comments / whitespace stripped) of the sources of commit %s, verified equal to the tree's by tools/restamp_hash.py" % commit[:12]
    if whole is not None:
        open(f, "w").write(json.dumps(doc, indent=1) + "\n")
    else:
        lines[-1] = json.dumps(doc)
        open(f, "w").write("\n".join(lines) + "\n")
    print(f, "restamped")
