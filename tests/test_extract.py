"""CPU: presence-matrix extractors and the matrix container (SURVEY §8f-2/3 host side)."""

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
