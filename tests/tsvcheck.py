"""What the reference's TSV consumers rely on (`lexicmap utils 2blast / 2sam / merge-search-results`,
lexicmap/cmd/2sam.go:105-307, 2blast.go, merge-search-results.go:142-194; column list search.go:82-115):
24 tab-separated columns with -a/--all (20 without), 1-based inclusive coordinates, `%.3f` percentages, `%.2e` e-value,
a CIGAR made of `(\\d+)([M=XIDNSHP])` tokens in SAM convention (I consumes the query, D the subject; X kept), and
qseq / sseq / align strings of the alignment length. check_line() asserts all of it for one row."""
import re

RE_CIGAR = re.compile(r"(\d+)([M=XIDNSHP])")       # 2sam.go:445
RE_F3 = re.compile(r"^\d+\.\d{3}$")
RE_E2 = re.compile(r"^\d\.\d{2}e[+-]\d{2,3}$")
COLUMNS = ["query", "qlen", "hits", "sgenome", "sseqid", "qcovGnm", "cls", "hsp", "qcovHSP", "alenHSP", "pident", "gaps",
           "qstart", "qend", "sstart", "send", "sstr", "slen", "evalue", "bitscore", "cigar", "qseq", "sseq", "align"]


def check_line(line, all_columns=True):
    c = line.rstrip("\n").split("\t")
    assert len(c) == (24 if all_columns else 20), len(c)
    qlen, hits, cls, hsp, alen, gaps = int(c[1]), int(c[2]), int(c[6]), int(c[7]), int(c[9]), int(c[11])
    qs, qe, ss, se, slen, bits = int(c[12]), int(c[13]), int(c[14]), int(c[15]), int(c[17]), int(c[19])
    for j in (5, 8, 10):
        assert RE_F3.match(c[j]), (j, c[j])
        assert 0.0 <= float(c[j]) <= 100.0
    assert RE_E2.match(c[18]), c[18]
    assert c[16] in "+-" and len(c[16]) == 1
    assert hits >= 1 and cls >= 1 and hsp >= 1 and bits >= 0
    assert 1 <= qs <= qe <= qlen and 1 <= ss <= se <= slen
    if not all_columns:
        return None
    toks = RE_CIGAR.findall(c[20])
    assert "".join(n + o for n, o in toks) == c[20] and toks, c[20]
    tot = {"M": 0, "X": 0, "I": 0, "D": 0}
    for n, o in toks:
        assert o in tot, o                         # lexicmap only emits M, X, I, D
        tot[o] += int(n)
    for (n1, o1), (n2, o2) in zip(toks, toks[1:]):
        assert o1 != o2                            # runs are merged
    assert toks[0][1] == "M" and toks[-1][1] == "M"  # trimmed to the first / last match (lib-index-search.go:2327-2340)
    assert alen == sum(tot.values())
    assert gaps == tot["I"] + tot["D"]
    assert qe - qs + 1 == tot["M"] + tot["X"] + tot["I"]   # SAM convention: I consumes the query
    assert se - ss + 1 == tot["M"] + tot["X"] + tot["D"]
    assert abs(float(c[10]) - 100.0 * tot["M"] / alen) < 0.0006
    qseq, sseq, al = c[21], c[22], c[23]
    assert len(qseq) == len(sseq) == len(al) == alen
    assert qseq.count("-") == tot["D"] and sseq.count("-") == tot["I"] and al.count("|") == tot["M"]
    # walk the CIGAR along the strings
    p = 0
    for n, o in toks:
        n = int(n)
        seg_q, seg_s, seg_a = qseq[p:p + n], sseq[p:p + n], al[p:p + n]
        if o == "M":
            assert seg_q == seg_s and set(seg_a) == {"|"}
        elif o == "X":
            assert all(a != b for a, b in zip(seg_q, seg_s)) and "-" not in seg_q + seg_s and set(seg_a) == {" "}
        elif o == "I":
            assert set(seg_s) == {"-"} and "-" not in seg_q
        else:
            assert set(seg_q) == {"-"} and "-" not in seg_s
        p += n
    # edit distance the way 2sam derives NM (2sam.go:433-443): X + I + D
    return dict(nm=tot["X"] + tot["I"] + tot["D"], **tot)
