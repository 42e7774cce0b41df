"""CPU: the native .sim parser (impop_sim_parse, host C++) against the Python readers that
mirror the reference line by line, on clean, ragged and hostile files."""
import random

import numpy as np
import pytest

from conftest import load_golden


def write(tmp_path, name, text):
    p = tmp_path / name
    p.write_bytes(text.encode() if isinstance(text, str) else text)
    return str(p)


def via_python(path, flavor):
    from impop_amd import simfile
    if flavor == "pica2":
        d, el, pc = simfile.read_similarity_file_pica2(path)
    else:
        d, el = simfile.read_similarity_file_hfst(path)
        pc = None
    names = sorted(el)
    return names, simfile.densify(d, names), pc


def same(a, b):
    return a.shape == b.shape and bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())


@pytest.mark.parametrize("flavor", ["pica2", "hfst"])
def test_native_equals_python_on_clean_files(tmp_path, flavor):
    from impop_amd import simfile
    g = load_golden("cli_pansn.json")
    p = write(tmp_path, "win8.sim", g["sim_text"])
    names, dense, rows = simfile.read_dense(p, flavor)
    pn, pd, pc = via_python(p, flavor)
    assert names == pn and same(dense, pd) and rows == 64
    # random table: duplicates (later row wins), self pairs, extra columns, CRLF, exponents, blank lines
    rnd = random.Random(5)
    nm = [f"S{i:03d}#{h}#chr{rnd.randint(1, 3)}:{rnd.randint(0, 9)}-{rnd.randint(10, 99)}" for i in range(40) for h in (1, 2)]
    lines = ["x\tgroup.b\tjunk\testimated.identity\tgroup.a"]
    for _ in range(3000):
        a, b = rnd.choice(nm), rnd.choice(nm)
        v = rnd.choice([repr(rnd.random()), "1", "0.99950", "1e-3", "9.99E-01", " 0.5 ", "+.5", "5.", "1E+0", "nan", "inf", "-Infinity"])
        lines.append(f"q\t{b}\tzz\t{v}\t{a}\textra\tmore")
        if rnd.random() < 0.02:
            lines.append("")
    p2 = write(tmp_path, "rand.sim", "\r\n".join(lines) + "\r\n")
    names, dense, rows = simfile.read_dense(p2, flavor)
    pn, pd, pc = via_python(p2, flavor)
    assert names == pn and same(dense, pd)
    if flavor == "pica2":
        assert rows == pc == 3000


def test_native_declines_what_it_is_not_sure_about(tmp_path, capsys):
    import ctypes as C

    from impop_amd import _lib, simfile
    lib = _lib.load()

    def rc_of(text, flavor=0):
        p = write(tmp_path, "t.sim", text)
        h = C.c_void_p()
        rc = lib.impop_sim_parse(p.encode(), flavor, C.byref(h))
        if rc == 0:
            lib.impop_sim_free(h)
        return rc, p
    hdr = "group.a\tgroup.b\testimated.identity\n"
    assert rc_of(hdr + "a\tb\t0.5\n")[0] == 0
    assert rc_of(hdr + 'a\t"b"\t0.5\n')[0] == _lib.E_UNSUPPORTED       # csv quoting
    assert rc_of(hdr + "a\tb\n")[0] == _lib.E_UNSUPPORTED               # short row
    assert rc_of(hdr + "a\tb\t1_0\n")[0] == _lib.E_UNSUPPORTED          # float('1_0') == 10.0 in Python
    assert rc_of(hdr + "a\tb\t0x10\n")[0] == _lib.E_UNSUPPORTED
    assert rc_of("a\tb\tc\nx\ty\t0.5\n")[0] == _lib.E_UNSUPPORTED      # missing columns
    assert rc_of("")[0] == _lib.E_UNSUPPORTED
    # ... and read_dense then behaves exactly like the reference through the Python reader
    rc, p = rc_of(hdr + "a\tb\t1_0\n")
    names, dense, rows = simfile.read_dense(p, "pica2")
    assert names == ["a", "b"] and dense[0, 1] == 10.0 and rows == 1
    rc, p = rc_of(hdr + "a\tb\tzzz\n")
    assert rc == 0  # parsed, bad value recorded -> the Python reader prints the reference's message
    with pytest.raises(SystemExit) as e:
        simfile.read_dense(p, "pica2")
    assert e.value.code == 1 and capsys.readouterr().out == "Error: Invalid similarity value on line 2: zzz\n"
    names, dense, rows = simfile.read_dense(p, "hfst")  # h-fst flavour: warned and skipped
    assert names == [] and "Warning: Invalid similarity value: zzz" in capsys.readouterr().err
    with pytest.raises(SystemExit):
        simfile.read_dense(str(tmp_path / "nope.sim"), "pica2")
    assert capsys.readouterr().out == f"Error: File not found {tmp_path / 'nope.sim'}\n"


def test_ingest_speed_465(tmp_path):
    """n = 465: 216 225 rows (SURVEY §3.1 measured 0.556 s in the reference reader)."""
    import time

    from impop_amd import simfile
    rng = np.random.default_rng(1)
    n = 465
    nm = [f"HG{i // 2:05d}#{i % 2 + 1}#CM0{i:05d}.1:1000-51000" for i in range(n)]
    sim = 0.998 + 0.002 * rng.random((n, n))
    sim = np.minimum(sim, sim.T)
    with open(tmp_path / "big.sim", "w") as f:
        f.write("group.a\tgroup.b\tgroup.a.length\tgroup.b.length\tintersection\testimated.identity\n")
        for i in range(n):
            f.write("".join(f"{nm[i]}\t{nm[j]}\t50000\t50000\t49900\t{float(sim[i, j])!r}\n" for j in range(n)))
    p = str(tmp_path / "big.sim")
    t0 = time.perf_counter()
    names, dense, rows = simfile.read_dense(p, "pica2")
    t_native = time.perf_counter() - t0
    t0 = time.perf_counter()
    pn, pd, pc = via_python(p, "pica2")
    t_python = time.perf_counter() - t0
    assert names == pn and same(dense, pd) and rows == pc == n * n
    assert (dense == sim).all()
    print(f"native {t_native * 1e3:.1f} ms vs python {t_python * 1e3:.1f} ms ({t_python / t_native:.1f}x)")
    assert t_native < t_python
