"""CPU: presence-matrix extractors and the matrix container (SURVEY §8f-2/3 host side)."""

import numpy as np
import pytest

import impop_amd
from impop_amd import extract, matrixio

GFA = """H\tVN:Z:1.0
S\t1\tACGT
S\t2\tA
S\t3\tG
S\t4\tTTTTT
S\t5\t*\tLN:i:3
P\tCHM13#0#chr1:1000-1013\t1+,2+,4+,5+\t*
P\tHG002#1#ctgA:0-13\t1+,3+,4+,5+\t*
W\tHG003\t2\tctgB\t0\t9\t>1>2>4
P\tHG002#2#ctgC:5-10\t4-,2-\t*
"""


def test_from_gfa(tmp_path):
    p = tmp_path / "t.gfa"
    p.write_text(GFA)
    mf = extract.from_gfa(str(p), ref_prefix="CHM13#0#", expand_bp=False)
    assert mf.names == ["CHM13#0#chr1:1000-1013", "HG002#1#ctgA:0-13", "HG002#2#ctgC:5-10", "HG003#2#ctgB:0-9"]
    m = impop_amd.unpack_hap_major(mf.bits, mf.n_site)
    assert m.tolist() == [[1, 1, 0, 1, 1], [1, 0, 1, 1, 1], [0, 1, 0, 1, 0], [1, 1, 0, 1, 0]]
    assert mf.site_pos.tolist() == [1000, 1004, 1004, 1005, 1010]
    assert mf.site_weight.tolist() == [4, 1, 1, 5, 3]  # node lengths: the weights of the node-level columns
    e = extract.from_gfa(str(p), ref_prefix="CHM13#0#", expand_bp=True)
    assert e.n_site == 4 + 1 + 1 + 5 + 3
    me = impop_amd.unpack_hap_major(e.bits, e.n_site)
    assert (me[:, :4] == m[:, [0]]).all() and (me[:, 6:11] == m[:, [3]]).all()
    assert e.site_pos.tolist() == [1000, 1001, 1002, 1003, 1004, 1004, 1005, 1006, 1007, 1008, 1009, 1010, 1011, 1012]
    assert e.site_range(1004, 1010) == (4, 11) and e.site_range(0, 999) == (0, 0) and e.site_range(1012, 5000) == (13, 14)
    # container round trip
    matrixio.save_matrix(str(tmp_path / "m.npz"), e)
    z = matrixio.load_matrix(str(tmp_path / "m.npz"))
    assert z.names == e.names and z.n_site == e.n_site and (z.bits == e.bits).all() and (z.site_pos == e.site_pos).all()
    assert e.site_weight is None and z.site_weight is None
    matrixio.save_matrix(str(tmp_path / "n.npz"), mf)
    assert matrixio.load_matrix(str(tmp_path / "n.npz")).site_weight.tolist() == [4, 1, 1, 5, 3]


def test_from_paths_table(tmp_path):
    p = tmp_path / "paths.tsv"
    p.write_text("path.name\tpath.length\tnode.count\tnode.1\tnode.2\tnode.3\n"
                 "b#1#x\t10\t2\t1\t0\t2\n"
                 "a#1#x\t12\t3\t1\t1\t1\n")
    mf = extract.from_paths_table(str(p))
    assert mf.names == ["a#1#x", "b#1#x"]
    assert impop_amd.unpack_hap_major(mf.bits, 3).tolist() == [[1, 1, 1], [1, 0, 1]]
    assert mf.site_range(1, 3) == (1, 3)


def _random_gfa(rng, n_seg, n_path, numeric=True, walks=True):
    ids = [str(int(x)) for x in rng.choice(np.arange(1, 10 * n_seg), size=n_seg, replace=False)] if numeric else \
          [f"s{int(x)}" if rng.random() < 0.5 else str(int(x)) for x in rng.choice(np.arange(1, 10 * n_seg), size=n_seg, replace=False)]
    lines = ["H\tVN:Z:1.1"]
    for i in ids:
        L = int(rng.integers(1, 30))
        k = rng.random()
        if k < 0.6:
            lines.append(f"S\t{i}\t{'ACGT' * 8}"[: len(f"S\t{i}\t") + L])
        elif k < 0.8:
            lines.append(f"S\t{i}\t*\tLN:i:{L}")
        else:
            lines.append(f"S\t{i}\t{'A' * 3}\tXX:Z:foo\tLN:i:{L}\tLN:i:{L + 2}")   # the last LN tag wins
    if rng.random() < 0.5:
        lines.append(f"S\t{ids[0]}\t{'C' * 7}")  # a repeated id: later line overwrites the length, keeps the first position
    names = []
    for r in range(n_path):
        steps = [ids[int(j)] for j in np.sort(rng.choice(n_seg, size=int(rng.integers(1, n_seg + 1)), replace=False))]
        if rng.random() < 0.3:
            steps = steps + steps[: 2]   # revisits
        if walks and rng.random() < 0.4:
            w = "".join((">" if rng.random() < 0.7 else "<") + s for s in steps)
            coords = ("0", str(len(steps))) if rng.random() < 0.7 else ("*", "*")
            lines.append(f"W\tHG{r:03d}\t{int(rng.integers(1, 3))}\tctg{r}\t{coords[0]}\t{coords[1]}\t{w}")
        else:
            nm = f"HG{int(rng.integers(0, n_path)):03d}#{int(rng.integers(1, 3))}#c{r}:{int(rng.integers(0, 50))}-{int(rng.integers(50, 99))}"
            lines.append(f"P\t{nm}\t" + ",".join(s + ("+" if rng.random() < 0.7 else "-") for s in steps) + "\t*")
            names.append(nm)
    rng.shuffle(lines[1:])  # S / P / W lines in any order
    return "\n".join(lines) + "\n", names


def test_native_gfa_parser_equals_python_extractor(tmp_path):
    """impop_gfa_parse (host C++ in libimpop_hip.so) against extract.from_gfa(native=False), the definition: names, packed
    bits, node lengths (site weights) and reference coordinates on random graphs — numeric and mixed ids, P and W lines in
    any order, revisited and repeated segments, several LN tags — and the fall-back on files the parser declines."""
    import numpy as np
    rng = np.random.default_rng(3)
    for trial in range(40):
        text, pnames = _random_gfa(rng, int(rng.integers(1, 200)), int(rng.integers(1, 12)), numeric=trial % 3 != 0)
        p = tmp_path / f"g{trial}.gfa"
        p.write_text(text)
        ref = None
        if pnames and trial % 2 == 0:
            ref = sorted(pnames)[0].split(":")[0][:6]
        want = extract.from_gfa(str(p), ref_prefix=ref, expand_bp=False, native=False)
        got = extract._from_gfa_native(str(p), ref)
        assert got is not None, trial
        assert got.names == want.names and got.n_site == want.n_site
        assert (got.bits == want.bits).all()
        assert (got.site_weight == want.site_weight).all() and got.site_weight.dtype == want.site_weight.dtype
        if ref is None:
            assert got.site_pos is None and want.site_pos is None
        else:
            assert (got.site_pos == want.site_pos).all() and got.contig == want.contig
        assert (extract.from_gfa(str(p), ref_prefix=ref, expand_bp=False).bits == want.bits).all()   # the default path
    # declined / failing inputs fall back to the Python reader, whose errors are the interface
    bad = tmp_path / "bad.gfa"
    bad.write_text("S\t1\tACGT\nP\tx\t1+,2+\t*\n")                       # a step on a segment without S line
    assert extract._from_gfa_native(str(bad), None) is None
    import pytest
    with pytest.raises(KeyError):
        extract.from_gfa(str(bad), expand_bp=False)
    crlf = tmp_path / "crlf.gfa"
    crlf.write_bytes(b"S\t1\tACGT\r\nP\tx\t1+\t*\r\n")
    assert extract._from_gfa_native(str(crlf), None) is None
    assert extract._from_gfa_native(str(tmp_path / "t.gfa"), "NOPE#") is None if (tmp_path / "t.gfa").exists() else True
    (tmp_path / "t2.gfa").write_text(GFA)
    assert extract._from_gfa_native(str(tmp_path / "t2.gfa"), "NOPE#") is None
    with pytest.raises(ValueError):
        extract.from_gfa(str(tmp_path / "t2.gfa"), ref_prefix="NOPE#", expand_bp=False)


def test_native_paths_table_parser_equals_python_reader(tmp_path):
    """impop_paths_table_parse (host C++ in libimpop_hip.so, rows parsed by several threads) against
    extract.from_paths_table(native=False), the definition: names, order, packed bits — with visit counts > 1, empty
    fields, duplicate names (stable order), no trailing newline, blank lines; files the native parser declines (ragged
    rows, CRLF, too few columns) fall back to the Python reader and its error texts."""
    rng = np.random.default_rng(9)
    for trial, (n, W) in enumerate([(1, 1), (7, 63), (23, 64), (40, 1000), (3, 129)]):
        p = tmp_path / f"t{trial}.tsv"
        names = [f"s{int(rng.integers(0, 12))}#{int(rng.integers(1, 3))}#c" for _ in range(n)]
        rows = ["path.name\tpath.length\tnode.count\t" + "\t".join(f"node.{k + 1}" for k in range(W))]
        for nm in names:
            fields = [("" if rng.random() < 0.05 else str(int(v))) for v in rng.choice([0, 0, 0, 1, 1, 2, 10, 100], size=W)]
            rows.append(f"{nm}\t{int(rng.integers(1, 999))}\t{W}\t" + "\t".join(fields))
            if rng.random() < 0.2:
                rows.append("")   # blank line: skipped
        text = "\n".join(rows) + ("\n" if trial % 2 else "")
        p.write_text(text)
        want = extract.from_paths_table(str(p), native=False)
        got = extract._from_paths_table_native(str(p))
        assert got is not None
        assert got.names == want.names and got.n_site == want.n_site == W
        assert got.bits.shape == want.bits.shape and (got.bits == want.bits).all()
        assert got.site_weight is None and got.site_pos is None
        assert (extract.from_paths_table(str(p)).bits == want.bits).all()
    ragged = tmp_path / "ragged.tsv"
    ragged.write_text("path.name\tpath.length\tnode.count\tn1\tn2\na\t1\t2\t1\t0\nb\t1\t2\t1\n")
    assert extract._from_paths_table_native(str(ragged)) is None
    with pytest.raises(ValueError, match="fields, expected 5"):
        extract.from_paths_table(str(ragged))
    long_row = tmp_path / "long.tsv"
    long_row.write_text("path.name\tpath.length\tnode.count\tn1\na\t1\t1\t1\t0\n")
    assert extract._from_paths_table_native(str(long_row)) is None
    crlf = tmp_path / "crlf.tsv"
    crlf.write_bytes(b"path.name\tpath.length\tnode.count\tn1\tn2\r\na\t1\t2\t1\t0\r\n")
    assert extract._from_paths_table_native(str(crlf)) is None
    assert impop_amd.unpack_hap_major(extract.from_paths_table(str(crlf)).bits, 2).tolist() == [[1, 0]]
    few = tmp_path / "few.tsv"
    few.write_text("a\tb\tc\n")
    assert extract._from_paths_table_native(str(few)) is None
    with pytest.raises(ValueError, match="expected >= 4"):
        extract.from_paths_table(str(few))
    assert extract._from_paths_table_native(str(tmp_path / "missing.tsv")) is None


def test_paths_table_readers_vs_reference_op_afs(tmp_path):
    """tests/golden/afs_table.json: an `odgi paths -H` table and what the REAL scripts/wip/op-afs.py (read_file_to_matrix +
    allele_freq per node column, op-afs.py:112-118) returned for it — per column the count and frequency of the value the
    file's first data row holds.  Both readers (native impop_paths_table_parse and the Python definition) must yield a
    matrix whose column sums reproduce every captured count: count = c_s where the first row carries the node, n - c_s
    where it does not."""
    import json
    import os
    from conftest import ROOT
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "afs_table.json")))
    p = tmp_path / "paths.tsv"
    p.write_text(g["table_text"])
    first_name = g["table_text"].split("\n")[1].split("\t")[0]
    for native in (True, False):
        mf = extract.from_paths_table(str(p), native=native)
        assert mf.n_hap == g["n_path"] and mf.n_site == g["n_node"] and mf.names == sorted(mf.names)
        m = impop_amd.unpack_hap_major(mf.bits, mf.n_site).astype(np.int64)
        first = m[mf.names.index(first_name)]
        c = m.sum(0)
        for k, col in enumerate(g["columns"]):
            assert int(first[k]) == col["value"], (native, col)
            count = int(c[k]) if col["value"] == 1 else mf.n_hap - int(c[k])
            assert count == col["count"] and count / mf.n_hap == float.fromhex(col["freq"]), (native, col, count)
