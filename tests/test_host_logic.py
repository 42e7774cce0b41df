"""CPU tests of the host layer: name handling, .sim ingest and its error texts, the C-ABI
library loads and exports every declared symbol (no compute without a GPU)."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden


def test_popnames_golden():
    from impop_amd import popnames
    g = load_golden("popnames.json")
    assert [popnames.canonicalize_identifier(r) for r in g["raw"]] == g["canonical"]
    exp, missing = popnames.expand_population(g["raw"], set(g["sequences"]))
    assert sorted(exp) == g["expanded"] and missing == g["missing"]


def test_abi_exports_every_declared_symbol():
    from impop_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "impop_hip.h")).read()
    declared = set(re.findall(r"^(?:int|const char \*)\s*\*?\s*(impop_[a-z0-9_]+)\(", hdr, flags=re.M))
    assert len(declared) >= 28
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.SO_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert declared <= exported
    assert all(s.startswith("impop_") for s in exported if not s.startswith("_")), exported  # nothing else leaks
    assert lib.impop_version() == _lib.ABI_VERSION == int(re.search(r"#define IMPOP_ABI_VERSION (\d+)", hdr).group(1))


def test_no_cpu_fallback_without_gpu():
    import impop_amd
    from impop_amd import _lib
    import ctypes as C
    c = C.c_int(-1)
    rc = _lib.load().impop_device_count(C.byref(c))
    if rc == 0 and c.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(impop_amd.ImpopError) as ei:
        impop_amd.Context(0)
    assert ei.value.code == _lib.E_NODEVICE and "no CPU fallback" in str(ei.value)
    from impop_amd import runtime, tj_d
    runtime.set_default_context(None)
    with pytest.raises(impop_amd.ImpopError):
        tj_d.tajimas_d(10, 5.0, 0.1)


def test_product_never_imports_oracle():
    bad = []
    for base in ("impop_amd", "scripts"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".hip", ".h", ".cpp")):
                    txt = open(os.path.join(dp, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "impop_oracle" in txt.replace("oracle/impop_oracle.c", ""):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_sim_ingest_and_error_texts(tmp_path, capsys):
    from impop_amd import simfile
    g = load_golden("cli_pansn.json")
    p = tmp_path / "win8.sim"
    p.write_text(g["sim_text"])
    d, elements, pc = simfile.read_similarity_file_pica2(str(p))
    assert pc == 64 and len(elements) == 8 and len(d) == 36
    names = sorted(elements)
    dense = simfile.densify(d, names)
    assert dense.shape == (8, 8) and not np.isnan(dense).any() and (dense == dense.T).all()
    d2, seqs = simfile.read_similarity_file_hfst(str(p))
    assert d2 == d and seqs == elements
    # error behaviour of pica2.read_similarity_file (messages on stdout, exit 1)
    errs = g["errors"]
    with pytest.raises(SystemExit) as e:
        simfile.read_similarity_file_pica2(str(tmp_path / "nope.sim"))
    assert e.value.code == 1
    assert capsys.readouterr().out == f"Error: File not found {tmp_path / 'nope.sim'}\n"
    bad = tmp_path / "bad.sim"
    bad.write_text("a\tb\tc\nx\ty\t0.5\n")
    with pytest.raises(SystemExit):
        simfile.read_similarity_file_pica2(str(bad))
    assert capsys.readouterr().out == errs[2]["stdout"]
    bad2 = tmp_path / "bad2.sim"
    bad2.write_text("group.a\tgroup.b\testimated.identity\nx\ty\tzzz\n")
    with pytest.raises(SystemExit):
        simfile.read_similarity_file_pica2(str(bad2))
    assert capsys.readouterr().out == errs[3]["stdout"]
    # h-fst flavour: bad float is warned and skipped, errors on stderr
    d3, _ = simfile.read_similarity_file_hfst(str(bad2))
    assert d3 == {} and "Warning: Invalid similarity value: zzz" in capsys.readouterr().err
    with pytest.raises(SystemExit):
        simfile.read_similarity_file_hfst(str(tmp_path / "nope.sim"))
    assert capsys.readouterr().err == f"Error: File not found: {tmp_path / 'nope.sim'}\n"


def test_window_helpers():
    import impop_amd
    w = impop_amd.fixed_windows(242_700_000, 50_000)
    assert len(w) == 4854 and int(w[-1]["site_end"]) == 242_700_000 and (w["seq_len"] == 50_000).all()
    s = impop_amd.fixed_windows(100_003, 10_000, 5_000)
    assert int(s[0]["site_end"]) == 10_000 and int(s[1]["site_begin"]) == 5_000 and int(s[-1]["site_end"]) == 100_003
    m = np.random.default_rng(0).integers(0, 2, size=(7, 130), dtype=np.uint8)
    assert (impop_amd.unpack_hap_major(impop_amd.pack_hap_major(m), 130) == m).all()
    mk = impop_amd.pack_mask(np.array([1, 0, 1] + [0] * 62 + [1]), 66)
    assert mk.tolist() == [5, 2] and impop_amd.mask_from_indices([0, 2, 65], 66).tolist() == [5, 2]
    mw = impop_amd.make_windows([(0, 10), (5, 20, 777)])
    assert mw["seq_len"].tolist() == [10, 777]


def test_fst_3pi_known_answer():
    """doc/how_fst.md:59-65: piA 0.00000279, piB 0.00000577, piC 0.00000528 -> 0.1893939 (run_fst_impg.sh:207-218)."""
    from impop_amd.drivers import fst_3pi_fields, pi_union_site, pica_cell
    ta, tb, tc, avg, fst = fst_3pi_fields(0.00000279, 0.00000577, 0.00000528)
    assert (ta, tb, tc, avg) == ("0.00000279", "0.00000577", "0.00000528", "0.00000428")
    assert fst == "0.18939394" and abs(float(fst) - 0.1893939) < 1e-7
    assert fst_3pi_fields(0.1, 0.2, 0.0)[4] == "NA"
    assert pica_cell(2.093e-05, 1000) == "0.00002093 (sequence length: 1000)"
    rec = {"n_sites": 100, "sum_a": 30, "sum_b": 12, "sum_ab": 58}
    assert pi_union_site(rec, 3, 2, 100) == 100 / (10.0 * 100) / 100


def test_c_abi_shard_rule_equals_python_rule():
    """impop_shard_range / impop_shard_windows are host arithmetic (no GPU): the C rule behind impop_scan_sharded and
    impop_gather_records is the rule of impop_amd.distributed (first n % shards ranges one window longer; the slab of a
    shard is the union of its windows' sites, halo included)."""
    import ctypes as C

    import impop_amd
    from impop_amd import _lib
    from impop_amd.distributed import shard_range, shard_windows
    lib = _lib.load()
    for n in (0, 1, 7, 64, 1000, 4854):
        for world in (1, 2, 3, 8, 13):
            covered = 0
            for r in range(world):
                f, c = C.c_uint64(), C.c_uint64()
                assert lib.impop_shard_range(n, world, r, C.byref(f), C.byref(c)) == 0
                lo, hi = shard_range(n, world, r)
                assert (f.value, f.value + c.value) == (lo, hi)
                covered += c.value
            assert covered == n
    f, c = C.c_uint64(), C.c_uint64()
    assert lib.impop_shard_range(10, 0, 0, C.byref(f), C.byref(c)) == _lib.E_INVALID
    assert lib.impop_shard_range(10, 4, 4, C.byref(f), C.byref(c)) == _lib.E_INVALID
    for size, step in ((1000, None), (1000, 500), (700, 300)):
        wins = impop_amd.fixed_windows(12345, size, step)
        for world in (1, 2, 5):
            for r in range(world):
                got = impop_amd.shard_windows_c(wins, world, r)
                loc, s0, s1, (lo, hi) = shard_windows(wins, world, r)
                assert got == (lo, hi - lo, s0, s1), (size, step, world, r, got)


def test_config4_sharding_geometry_of_the_c_abi():
    """BASELINE configs[3] (whole-genome 10 kb windows at a 5 kb step over N GPUs): what bench.py --config4 and
    scripts/impop_scan.py --devices N ask of impop_shard_windows — host arithmetic, no GPU.  Contiguous window ranges (the first
    n % shards ranges one window longer), every shard's slab = exactly the sites its windows touch, neighbouring slabs overlapping
    by the (window - step) halo, the same ranges the torch.distributed path (impop_amd.distributed.shard_windows) cuts."""
    import impop_amd
    from impop_amd import engine
    from impop_amd.distributed import shard_range, shard_windows
    G, Wn, step = 20_000_123, 10_000, 5_000
    wins = impop_amd.fixed_windows(G, Wn, step)
    assert len(wins) == 4001 and int(wins[-1]["site_end"]) == G
    for world in (1, 2, 3, 8):
        nxt, covered = 0, 0
        for r in range(world):
            first, cnt, s0, s1 = engine.shard_windows_c(wins, world, r)
            lo, hi = shard_range(len(wins), world, r)
            assert (first, first + cnt) == (lo, hi) and first == nxt
            nxt += cnt
            assert s0 == int(wins[first]["site_begin"]) and s1 == int(wins[first + cnt - 1]["site_end"])
            loc, t0, t1, _ = shard_windows(wins, world, r)
            assert (t0, t1) == (s0, s1) and len(loc) == cnt and int(loc[0]["site_begin"]) == 0
            if r:  # halo: this slab starts (window - step) sites before the previous one ends
                assert prev_end - s0 == Wn - step
            prev_end = s1
            covered += s1 - s0
        assert nxt == len(wins)
        assert covered == G + (world - 1) * (Wn - step)  # every site once per slab, the halos twice
    # more shards than windows: trailing shards are empty, nobody reads out of range
    few = impop_amd.fixed_windows(25_000, Wn, step)
    cnts = [engine.shard_windows_c(few, 8, r)[1] for r in range(8)]
    assert sum(cnts) == len(few) and cnts == sorted(cnts, reverse=True)


def test_batch_driver_bed_warnings_are_the_reference_drivers(tmp_path, capsys):
    """scripts/impop_scan.py reads the BED like the driver whose table it prints: same rows kept, same stderr texts
    (run_tajd.sh:104-117, run_h-fst.sh:159-181, run_fst_impg.sh:166-179, run_pica2_impg.sh:128-136)."""
    import importlib.util
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    spec = importlib.util.spec_from_file_location("impop_scan_cli", os.path.join(ROOT, "scripts", "impop_scan.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bed = tmp_path / "w.bed"
    bed.write_text("# c\n\nchr1\t10\t20\nchr1\t30\nchr1\tx\t40\nchr1\t50\t50\nchr2\t60\t70\textra\n")
    want_rows = [("chr1", 10, 20), ("chr2", 60, 70)]
    texts = {
        "tajd": ["Warning: Skipping malformed BED entry: chr1 30 ", "Warning: Skipping malformed BED entry: chr1 x 40",
                 "Warning: Skipping non-positive interval length for chr1:50-50"],
        "hfst": ["Warning: Incomplete BED entry at line 4, skipping", "Warning: Non-integer coordinates at line 5: chr1:x-40, skipping",
                 "Warning: Invalid interval at line 6: chr1:50-50, skipping"],
        "fst3pi": ["Warning: Incomplete BED entry for chromosome chr1, skipping",
                   "Warning: Non-integer coordinates in BED entry chr1\\tx\\t40, skipping",
                   "Warning: Non-positive interval length for chr1:50-50, skipping"],
        "pica2": ["Warning: Skipping malformed BED entry: chr1 30 ", "Warning: Skipping malformed BED entry: chr1 x 40",
                  "Warning: Skipping region with non-positive length: chr1:50-50"],
    }
    for fmt, lines in texts.items():
        assert mod.read_bed(str(bed), fmt) == want_rows
        err = capsys.readouterr().err.strip().split("\n")
        assert err == lines, (fmt, err)
    assert mod.threshold_arg("0.9990") == (0.999, "0.9990") and mod.round_arg("none") == "none" and mod.round_arg("5") == 5
