"""GPU: scripts/impop_scan.py (the batch replacement of the run_*.sh window loops) prints the
driver TSV schemas with the values the per-window reference chain would print (checked through
the CPU oracle, which is pinned to the reference)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_batch_driver_tables(tmp_path, oracle):
    import impop_amd
    from impop_amd import matrixio
    rng = np.random.default_rng(12)
    n, W = 24, 5000
    anc = rng.integers(0, 2, size=W, dtype=np.uint8)
    f = np.repeat(anc[None], 4, axis=0) ^ (rng.random((4, W)) < 0.01).astype(np.uint8)
    m = f[rng.integers(0, 4, size=n)] ^ (rng.random((n, W)) < 0.002).astype(np.uint8)
    names = [f"S{i // 2:03d}#{i % 2 + 1}#chr9:{1000}-{1000 + W}" for i in range(n)]
    mf = matrixio.from_dense(m, names, origin=1000, contig="CHM13#0#chr9")
    matrixio.save_matrix(str(tmp_path / "m.npz"), mf)
    (tmp_path / "w.bed").write_text("# comment\nchr9\t1000\t3000\nchr9\t3000\t6000\nchr9\tx\ty\nCHM13#0#chr9\t2000\t2500\tname\n")
    (tmp_path / "A.txt").write_text("S000\nS001_hap1_hprc_r2\nS002#2\n")
    (tmp_path / "B.txt").write_text("S005\nS006\nS007_mat\n")
    (tmp_path / "all.txt").write_text("\n".join(f"S{i:03d}" for i in range(12)) + "\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "impop_scan.py"), "--matrix", str(tmp_path / "m.npz"),
                        "--bed", str(tmp_path / "w.bed"), "--format", "all", "-A", str(tmp_path / "A.txt"), "-B", str(tmp_path / "B.txt"),
                        "-l", str(tmp_path / "all.txt"), "-t", "1.0", "-r", "none"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().split("\n")
    assert lines[0] == "REGION\tLENGTH\tTHRESHOLD\tR_VALUE\tPICA_OUTPUT"       # run_pica2_impg.sh:122
    assert lines[4] == "REGION\tLENGTH\tFST\tPI_A\tPI_B\tPI_XY\tDXY\tDA"          # run_h-fst.sh:148
    assert lines[8] == "REGION\tLENGTH\tSAMPLES\tSEGREGATING_SITES\tPI\tTAJIMAS_D"  # run_tajd.sh:101
    assert "Skipping malformed BED entry" in r.stderr
    inA = np.array([1 if nm.startswith(("S000#", "S001#1#", "S002#2#")) else 0 for nm in names], np.uint8)
    inB = np.array([1 if nm.startswith(("S005#", "S006#", "S007#1#")) else 0 for nm in names], np.uint8)
    bits = oracle.pack_hap_major(m)
    ones = oracle.pack_mask(np.ones(n, np.uint8))
    wins = [(0, 2000, 2000, "CHM13#0#chr9:1000-3000"), (2000, 5000, 3000, "CHM13#0#chr9:3000-6000"), (1000, 1500, 500, "CHM13#0#chr9:2000-2500")]
    for k, (s0, s1, L, reg) in enumerate(wins):
        w = oracle.window_allpairs(bits, n, s0, s1, ones, oracle.pack_mask(inA), oracle.pack_mask(inB), L)
        assert lines[1 + k] == f"{reg}\t{L}\t1.0\t\t{w['pi_site']:.8f} (sequence length: {L})"
        h = lines[5 + k].split("\t")
        assert h[:2] == [reg, str(L)] and len(h) == 8
        for got, key in zip(h[2:], ("fst", "pi_a", "pi_b", "pi_xy", "dxy", "da")):
            # ".8f" text: equal up to one unit of the last printed digit (a value that sits on a
            # rounding boundary may print either way; the doubles agree to 1e-9 relative elsewhere)
            assert abs(float(got) - w[key]) <= 1.0000001e-8, (key, got, w[key])
        t = lines[9 + k].split("\t")
        assert t[:5] == [reg, str(L), "12", str(w["s_all"]), f"{w['pi_site']:.8f}"]
        # D is printed with repr(); run_tajd.sh:180 gives tj_d.py `-n SAMPLE_COUNT` = the 12 list LINES
        # (run_tajd.sh:83), not the 24 haplotypes they select => recompute through the oracle with n = 12
        D, _ = oracle.tajimas_d(12, float(w["s_all"]), oracle.py_round(w["pi_site"], 8))
        assert abs(float(t[5]) - D) <= 1e-9 * abs(D)
    assert "sample list has 12 lines but selects 24 haplotypes" in r.stderr
    # haplotype-level list with a repeated line, an unknown name, a comment and a blank: SAMPLES = awk's count (5)
    (tmp_path / "hap.txt").write_text("# panel\nS000_hap1\nS001#2\n\nS001#2\nNOPE_hap1\n  S003_mat\n")
    rh = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "impop_scan.py"), "--matrix", str(tmp_path / "m.npz"),
                         "--bed", str(tmp_path / "w.bed"), "--format", "tajd", "-l", str(tmp_path / "hap.txt"), "-t", "1", "-r", "none"], capture_output=True, text=True)
    assert rh.returncode == 0, rh.stderr
    lh = rh.stdout.strip().split("\n")
    sel = np.array([1 if nm.startswith(("S000#1#", "S001#2#", "S003#1#")) else 0 for nm in names], np.uint8)
    for k, (s0, s1, L, reg) in enumerate(wins):
        w = oracle.window_allpairs(bits, n, s0, s1, oracle.pack_mask(sel), oracle.pack_mask(inA), oracle.pack_mask(inB), L)
        t = lh[1 + k].split("\t")
        assert t[:5] == [reg, str(L), "5", str(w["s_all"]), f"{w['pi_site']:.8f}"]
        D, _ = oracle.tajimas_d(5, float(w["s_all"]), oracle.py_round(w["pi_site"], 8))
        assert (t[5] == "NA" and D != D) or abs(float(t[5]) - D) <= 1e-9 * abs(D)
    assert "sample list has 5 lines but selects 3 haplotypes" in rh.stderr
    # a list whose line count equals the matched haplotypes takes the scan's own D, without a warning
    (tmp_path / "hap3.txt").write_text("S000_hap1\nS001#2\nS003_mat\n")
    r3h = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "impop_scan.py"), "--matrix", str(tmp_path / "m.npz"),
                          "--bed", str(tmp_path / "w.bed"), "--format", "tajd", "-l", str(tmp_path / "hap3.txt"), "-t", "1", "-r", "none"], capture_output=True, text=True)
    assert r3h.returncode == 0 and "sample list has" not in r3h.stderr
    for k, (s0, s1, L, reg) in enumerate(wins):
        w = oracle.window_allpairs(bits, n, s0, s1, oracle.pack_mask(sel), oracle.pack_mask(inA), oracle.pack_mask(inB), L)
        t = r3h.stdout.strip().split("\n")[1 + k].split("\t")
        assert t[2] == "3" and ((t[5] == "NA" and w["tajima_d"] != w["tajima_d"]) or abs(float(t[5]) - w["tajima_d"]) <= 1e-9 * abs(w["tajima_d"]))
    # fewer than two lines: run_tajd.sh:84-87
    (tmp_path / "one.txt").write_text("S000_hap1\n")
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "impop_scan.py"), "--matrix", str(tmp_path / "m.npz"),
                         "--bed", str(tmp_path / "w.bed"), "--format", "tajd", "-l", str(tmp_path / "one.txt")], capture_output=True, text=True)
    assert r1.returncode == 1 and "Need at least two samples" in r1.stderr
    # --compact: the same tables from the matrix compacted to its variable sites
    rc = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "impop_scan.py"), "--matrix", str(tmp_path / "m.npz"),
                         "--bed", str(tmp_path / "w.bed"), "--format", "all", "-A", str(tmp_path / "A.txt"), "-B", str(tmp_path / "B.txt"),
                         "-l", str(tmp_path / "all.txt"), "--compact", "-t", "1.0", "-r", "none"], capture_output=True, text=True)
    assert rc.returncode == 0, rc.stderr
    assert rc.stdout == r.stdout
    # 3 x pi table (run_fst_impg.sh): PI_C = pica2 on the union list, here against the oracle's pica2
    r3 = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "impop_scan.py"), "--matrix", str(tmp_path / "m.npz"),
                         "--bed", str(tmp_path / "w.bed"), "--format", "fst3pi", "-A", str(tmp_path / "A.txt"), "-B", str(tmp_path / "B.txt")],
                        capture_output=True, text=True)
    assert r3.returncode == 0, r3.stderr
    l3 = r3.stdout.strip().split("\n")
    assert l3[0] == "REGION\tLENGTH\tTHRESHOLD\tR_VALUE\tPI_A\tPI_B\tPI_C\tPI_AB_AVG\tFST"  # run_fst_impg.sh:158
    for k, (s0, s1, L, reg) in enumerate(wins):
        sim = oracle.identity(oracle.pairwise_counts(bits, n, s0, s1), s1 - s0, 0)
        t = l3[1 + k].split("\t")
        for col, sel in ((4, inA), (5, inB), (6, inA | inB)):
            idx = np.nonzero(sel)[0]
            _, ps, _, _ = oracle.pica2(sim[np.ix_(idx, idx)], 1.0, L, None)
            assert abs(float(t[col]) - ps) <= 1.0000001e-8, (reg, col, t[col], ps)
        fa, fb, fc = float(t[4]), float(t[5]), float(t[6])
        assert t[7] == f"{0.5 * (fa + fb):.8f}" and t[8] == ("NA" if fc == 0 else f"{(fc - 0.5 * (fa + fb)) / fc:.8f}")
    # ---- the reference's DEFAULT chain (run_tajd.sh:9-10,166-180): --format tajd without -t / -r is pica2 -t 0.999 -r 5
    # on the all-pairs path -> "%.8f" -> tj_d with n = the list's line count
    rd = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "impop_scan.py"), "--matrix", str(tmp_path / "m.npz"),
                         "--bed", str(tmp_path / "w.bed"), "--format", "tajd", "-l", str(tmp_path / "hap3.txt")], capture_output=True, text=True)
    assert rd.returncode == 0, rd.stderr
    ld = rd.stdout.strip().split("\n")
    assert ld[0] == "REGION\tLENGTH\tSAMPLES\tSEGREGATING_SITES\tPI\tTAJIMAS_D"
    sel3 = np.array([1 if nm.startswith(("S000#1#", "S001#2#", "S003#1#")) else 0 for nm in names], np.uint8)
    idx3 = np.nonzero(sel3)[0]
    for k, (s0, s1, L, reg) in enumerate(wins):
        sim = oracle.identity(oracle.pairwise_counts(bits, n, s0, s1), s1 - s0, 0)
        pi, ps, _, G = oracle.pica2(sim[np.ix_(idx3, idx3)], 0.999, L, 5)
        w1 = oracle.window_allpairs(bits, n, s0, s1, oracle.pack_mask(sel3), oracle.pack_mask(inA), oracle.pack_mask(inB), L)
        D, _ = oracle.tajimas_d(3, float(w1["s_all"]), oracle.py_round(ps, 8))
        t = ld[1 + k].split("\t")
        assert t[:5] == [reg, str(L), "3", str(w1["s_all"]), f"{ps:.8f}"], (t, ps)
        assert (t[5] == "NA" and D != D) or abs(float(t[5]) - D) <= 1e-9 * abs(D)
    # --format all at the defaults: THRESHOLD / R_VALUE say 0.999 / 5 and the PICA_OUTPUT / PI columns are computed there;
    # the h-fst table stays unrounded (run_h-fst.sh passes -r only when given); with --fst-round-digits it is h-fst.py -r
    for fr in (None, 5, 3):
        ra = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "impop_scan.py"), "--matrix", str(tmp_path / "m.npz"),
                             "--bed", str(tmp_path / "w.bed"), "--format", "all", "-A", str(tmp_path / "A.txt"), "-B", str(tmp_path / "B.txt")]
                            + ([] if fr is None else ["--fst-round-digits", str(fr)]), capture_output=True, text=True)
        assert ra.returncode == 0, ra.stderr
        la = ra.stdout.strip().split("\n")
        for k, (s0, s1, L, reg) in enumerate(wins):
            sim = oracle.identity(oracle.pairwise_counts(bits, n, s0, s1), s1 - s0, 0)
            pi, ps, _, G = oracle.pica2(sim, 0.999, L, 5)
            assert la[1 + k] == f"{reg}\t{L}\t0.999\t5\t{ps:.8f} (sequence length: {L})"
            want, _ = oracle.hfst(sim, inA, inB, L, fr)
            h = la[5 + k].split("\t")
            for got, key in zip(h[2:], ("fst", "pi_a", "pi_b", "pi_xy", "dxy", "da")):
                assert abs(float(got) - want[key]) <= 1.0000001e-8, (fr, key, got, want[key])
            S = oracle.window_allpairs(bits, n, s0, s1, ones, oracle.pack_mask(inA), oracle.pack_mask(inB), L)["s_all"]
            D, _ = oracle.tajimas_d(n, float(S), oracle.py_round(ps, 8))
            t = la[9 + k].split("\t")
            assert t[:5] == [reg, str(L), str(n), str(S), f"{ps:.8f}"]
            assert (t[5] == "NA" and D != D) or abs(float(t[5]) - D) <= 1e-9 * abs(D)
    # --format hfst -r N (run_h-fst.sh:76-78) on the all-pairs path
    rh5 = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "impop_scan.py"), "--matrix", str(tmp_path / "m.npz"),
                          "--bed", str(tmp_path / "w.bed"), "--format", "hfst", "-A", str(tmp_path / "A.txt"), "-B", str(tmp_path / "B.txt"),
                          "-r", "3"], capture_output=True, text=True)
    assert rh5.returncode == 0, rh5.stderr
    changed = False
    for k, (s0, s1, L, reg) in enumerate(wins):
        sim = oracle.identity(oracle.pairwise_counts(bits, n, s0, s1), s1 - s0, 0)
        want, _ = oracle.hfst(sim, inA, inB, L, 3)
        want0, _ = oracle.hfst(sim, inA, inB, L, None)
        h = rh5.stdout.strip().split("\n")[1 + k].split("\t")
        for got, key in zip(h[2:], ("fst", "pi_a", "pi_b", "pi_xy", "dxy", "da")):
            assert abs(float(got) - want[key]) <= 1.0000001e-8, (key, got, want[key])
        changed = changed or abs(want["pi_a"] - want0["pi_a"]) > 1.5e-8
    assert changed  # rounding to 3 digits really changes the printed table somewhere
    bad_t = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "impop_scan.py"), "--matrix", str(tmp_path / "m.npz"),
                            "--bed", str(tmp_path / "w.bed"), "--format", "hfst", "-A", str(tmp_path / "A.txt"), "-B", str(tmp_path / "B.txt"),
                            "-t", "0.9"], capture_output=True, text=True)
    assert bad_t.returncode == 2 and "only with --fst-method grouped" in bad_t.stderr
    # the 3 x pi table with pica2's -t / -r (run_fst_impg.sh:73 passes them to all three runs)
    r3t = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "impop_scan.py"), "--matrix", str(tmp_path / "m.npz"),
                          "--bed", str(tmp_path / "w.bed"), "--format", "fst3pi", "-A", str(tmp_path / "A.txt"), "-B", str(tmp_path / "B.txt"),
                          "-t", "0.995", "-r", "4"], capture_output=True, text=True)
    assert r3t.returncode == 0, r3t.stderr
    for k, (s0, s1, L, reg) in enumerate(wins):
        sim = oracle.identity(oracle.pairwise_counts(bits, n, s0, s1), s1 - s0, 0)
        t = r3t.stdout.strip().split("\n")[1 + k].split("\t")
        assert t[:4] == [reg, str(L), "0.995", "4"]
        for c, selx in ((4, inA), (5, inB), (6, inA | inB)):
            ix = np.nonzero(selx)[0]
            _, ps, _, _ = oracle.pica2(sim[np.ix_(ix, ix)], 0.995, L, 4)
            assert abs(float(t[c]) - ps) <= 1.0000001e-8, (reg, c, t[c], ps)
    # thresholded pica2 goes through the all-pairs path
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "impop_scan.py"), "--matrix", str(tmp_path / "m.npz"),
                         "--bed", str(tmp_path / "w.bed"), "--format", "pica2", "-t", "0.9950", "-r", "4"], capture_output=True, text=True)
    assert r2.returncode == 0, r2.stderr
    l2 = r2.stdout.strip().split("\n")
    for k, (s0, s1, L, reg) in enumerate(wins):
        sim = oracle.identity(oracle.pairwise_counts(bits, n, s0, s1), s1 - s0, 0)
        pi, ps, _, _ = oracle.pica2(sim, 0.995, L, 4)
        # THRESHOLD is echoed AS TYPED, like "${THRESHOLD}" in run_pica2_impg.sh:185-187
        assert l2[1 + k] == f"{reg}\t{L}\t0.9950\t4\t{ps:.8f} (sequence length: {L})"
    # run_pica2_impg.sh -l: one EFFECTIVE_LENGTH for pica2's -l and for the LENGTH column (:153-157,185-187)
    for extra in (["-t", "0.9950", "-r", "4"], []):
        r3 = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "impop_scan.py"), "--matrix", str(tmp_path / "m.npz"),
                             "--bed", str(tmp_path / "w.bed"), "--format", "pica2", "--sequence-length", "777"] + extra,
                            capture_output=True, text=True)
        assert r3.returncode == 0, r3.stderr
        l3 = r3.stdout.strip().split("\n")
        for k, (s0, s1, L, reg) in enumerate(wins):
            sim = oracle.identity(oracle.pairwise_counts(bits, n, s0, s1), s1 - s0, 0)
            pi, ps, _, _ = oracle.pica2(sim, 0.995 if extra else 1.0, 777, 4 if extra else None)
            t = l3[1 + k].split("\t")
            assert t[:2] == [reg, "777"] and t[4].endswith(" (sequence length: 777)"), l3[1 + k]
            assert abs(float(t[4].split()[0]) - ps) <= (0 if extra else 1.0000001e-8) or t[4].split()[0] == f"{ps:.8f}", (t, ps)


def test_batch_driver_grouped_fst(tmp_path, oracle):
    """--format hfst --fst-method grouped: hud.py's grouped Fst per BED row (oracle_hud_grouped on the
    oracle's identity of the window)."""
    from impop_amd import matrixio
    rng = np.random.default_rng(5)
    n, W = 20, 3000
    anc = rng.integers(0, 2, size=W, dtype=np.uint8)
    f = np.repeat(anc[None], 3, axis=0) ^ (rng.random((3, W)) < 0.02).astype(np.uint8)
    m = f[rng.integers(0, 3, size=n)] ^ (rng.random((n, W)) < 0.001).astype(np.uint8)
    names = [f"S{i // 2:03d}#{i % 2 + 1}#chr9:0-{W}" for i in range(n)]
    matrixio.save_matrix(str(tmp_path / "m.npz"), matrixio.from_dense(m, names, origin=0, contig="CHM13#0#chr9"))
    (tmp_path / "w.bed").write_text("chr9\t0\t1500\nchr9\t1000\t3000\n")
    (tmp_path / "A.txt").write_text("S000\nS001\nS002\nS003\n")
    (tmp_path / "B.txt").write_text("S005\nS006\nS007\nS008\nS009\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "impop_scan.py"), "--matrix", str(tmp_path / "m.npz"),
                        "--bed", str(tmp_path / "w.bed"), "--format", "hfst", "--fst-method", "grouped", "-t", "0.995",
                        "-A", str(tmp_path / "A.txt"), "-B", str(tmp_path / "B.txt")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().split("\n")
    assert lines[0] == "REGION\tLENGTH\tFST\tPI_A\tPI_B\tPI_XY\tDXY\tDA"
    inA = np.array([1 if int(nm[1:4]) <= 3 else 0 for nm in names], np.uint8)
    inB = np.array([1 if 5 <= int(nm[1:4]) <= 9 else 0 for nm in names], np.uint8)
    bits = oracle.pack_hap_major(m)
    for k, (s0, s1) in enumerate(((0, 1500), (1000, 3000))):
        sim = oracle.identity(oracle.pairwise_counts(bits, n, s0, s1), s1 - s0, 0)
        want, _ = oracle.hud_grouped(sim, inA, inB, 0.995, s1 - s0, None)
        h = lines[1 + k].split("\t")
        for got, key in zip(h[2:], ("fst", "pi_a", "pi_b", "pi_xy", "dxy", "da")):
            assert abs(float(got) - want[key]) <= 1.0000001e-8, (key, got, want[key])


def test_batch_driver_two_ranks_equal_one(tmp_path):
    """The driver sharded over 2 ranks (gloo rehearsal: both ranks share the one GPU of the test
    box; on a node it is one rank per GPU over RCCL) prints exactly what 1 rank prints."""
    import socket

    from impop_amd import matrixio
    rng = np.random.default_rng(21)
    n, W = 40, 64 * 257 + 5
    m = (rng.random((n, W)) < 0.3).astype(np.uint8)
    names = [f"S{i // 2:03d}#{i % 2 + 1}#chr9:0-{W}" for i in range(n)]
    matrixio.save_matrix(str(tmp_path / "m.npz"), matrixio.from_dense(m, names, origin=0, contig="CHM13#0#chr9"))
    with open(tmp_path / "w.bed", "w") as f:
        for s in range(0, W - 1000, 700):  # overlapping windows, unaligned
            f.write(f"chr9\t{s}\t{min(s + 1000, W)}\n")
    (tmp_path / "A.txt").write_text("\n".join(f"S{i:03d}" for i in range(0, 6)) + "\n")
    (tmp_path / "B.txt").write_text("\n".join(f"S{i:03d}" for i in range(8, 15)) + "\n")
    base = [os.path.join(ROOT, "scripts", "impop_scan.py"), "--matrix", str(tmp_path / "m.npz"), "--bed", str(tmp_path / "w.bed"),
            "--format", "all", "-A", str(tmp_path / "A.txt"), "-B", str(tmp_path / "B.txt")]
    one = subprocess.run([sys.executable] + base, capture_output=True, text=True)
    assert one.returncode == 0, one.stderr
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port)] + base + ["--backend", "gloo", "-o", str(tmp_path / "two.tsv")],
                         capture_output=True, text=True)
    assert two.returncode == 0, two.stderr[-2000:]
    assert open(tmp_path / "two.tsv").read() == one.stdout
    # ONE process driving several contexts through the C ABI (impop_scan_sharded; the contexts share the box's one GPU)
    for nd in (2, 3):
        many = subprocess.run([sys.executable] + base + ["--devices", str(nd)], capture_output=True, text=True)
        assert many.returncode == 0, many.stderr
        assert many.stdout == one.stdout, nd
    bad = subprocess.run([sys.executable] + base + ["--devices", "2", "--compact"], capture_output=True, text=True)
    assert bad.returncode == 2 and "--devices N" in bad.stderr
    # ... and the all-pairs formats (thresholded pica2, grouped Fst) through impop_pairwise_scan_sharded
    for extra in (["--format", "pica2", "-t", "0.99", "-r", "4"], ["--format", "hfst", "--fst-method", "grouped", "-t", "0.95"]):
        ap1 = subprocess.run([sys.executable] + base + extra, capture_output=True, text=True)
        assert ap1.returncode == 0, ap1.stderr
        ap3 = subprocess.run([sys.executable] + base + extra + ["--devices", "3"], capture_output=True, text=True)
        assert ap3.returncode == 0, ap3.stderr
        assert ap3.stdout == ap1.stdout and len(ap1.stdout.splitlines()) > 10, extra
    # the K-population panel (all pairs per window) sharded the same way
    (tmp_path / "C.txt").write_text("\n".join(f"S{i:03d}" for i in range(15, 20)) + "\n")
    panel = [os.path.join(ROOT, "scripts", "impop_scan.py"), "--matrix", str(tmp_path / "m.npz"), "--bed", str(tmp_path / "w.bed"),
             "--format", "hfst", "--panel", str(tmp_path / "A.txt"), str(tmp_path / "B.txt"), str(tmp_path / "C.txt")]
    one_p = subprocess.run([sys.executable] + panel, capture_output=True, text=True)
    assert one_p.returncode == 0, one_p.stderr
    assert one_p.stdout.count("# ") == 3
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    two_p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                            "127.0.0.1", "--master-port", str(port)] + panel + ["--backend", "gloo", "-o", str(tmp_path / "two_p.tsv")],
                           capture_output=True, text=True)
    assert two_p.returncode == 0, two_p.stderr[-2000:]
    assert open(tmp_path / "two_p.tsv").read() == one_p.stdout


def test_batch_driver_node_level_matrix_with_weights(tmp_path):
    """A GFA extracted with one weighted column per node (--no-expand-bp) gives the same pica2 / h-fst
    tables as the bp-expanded extraction when the BED rows fall on node boundaries."""
    rng = np.random.default_rng(3)
    n_node, n_hap = 60, 10
    lens = rng.integers(1, 30, size=n_node)
    seqs = ["".join(rng.choice(list("ACGT"), size=int(L))) for L in lens]
    lines = ["H\tVN:Z:1.0"] + [f"S\t{i + 1}\t{seqs[i]}" for i in range(n_node)]
    ref_nodes = list(range(0, n_node, 2)) + []  # the reference takes the even nodes (odd ones are alternative alleles)
    ref_nodes = sorted(ref_nodes)
    ref_len = int(sum(lens[i] for i in ref_nodes))
    lines.append("P\tCHM13#0#chr7:0-%d\t%s\t*" % (ref_len, ",".join(f"{i + 1}+" for i in ref_nodes)))
    for h in range(n_hap):
        steps = []
        for i in range(0, n_node, 2):
            steps.append(i if rng.random() < 0.7 else min(i + 1, n_node - 1))
        lines.append("P\tS%03d#%d#ctg:0-1\t%s\t*" % (h // 2, h % 2 + 1, ",".join(f"{i + 1}+" for i in steps)))
    (tmp_path / "g.gfa").write_text("\n".join(lines) + "\n")
    ex = os.path.join(ROOT, "scripts", "impop_extract.py")
    for out, extra in (("exp.npz", []), ("node.npz", ["--no-expand-bp"])):
        r = subprocess.run([sys.executable, ex, "--gfa", str(tmp_path / "g.gfa"), "--ref-prefix", "CHM13#0#", "-o", str(tmp_path / out)] + extra,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    # BED rows cut at reference-node starts (coordinates where a new reference node begins)
    starts = np.concatenate(([0], np.cumsum([lens[i] for i in ref_nodes])))
    bed = [(int(starts[2]), int(starts[12])), (int(starts[12]), int(starts[25])), (0, int(starts[-1]))]
    (tmp_path / "w.bed").write_text("".join(f"chr7\t{a}\t{b}\n" for a, b in bed))
    (tmp_path / "A.txt").write_text("S000\nS001\n")
    (tmp_path / "B.txt").write_text("S002\nS003\nS004\n")
    outs = {}
    for name in ("exp.npz", "node.npz"):
        for fmt in ("pica2", "hfst"):
            r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "impop_scan.py"), "--matrix", str(tmp_path / name), "--bed",
                                str(tmp_path / "w.bed"), "--format", fmt, "-A", str(tmp_path / "A.txt"), "-B", str(tmp_path / "B.txt")],
                               capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            outs[(name, fmt)] = r.stdout
    assert outs[("exp.npz", "pica2")] == outs[("node.npz", "pica2")]
    assert outs[("exp.npz", "hfst")] == outs[("node.npz", "hfst")]
    assert outs[("exp.npz", "hfst")].count("\n") == 4


def test_batch_driver_rows_go_to_their_chromosome(tmp_path, oracle):
    """run_pica2_impg.sh:139-151 builds REGION from each BED row's own chromosome: with one matrix per chromosome a
    whole-genome BED works in one call, rows come back in BED order, and a row of a chromosome nobody holds is skipped
    with a warning instead of being clamped onto another chromosome's matrix."""
    from impop_amd import matrixio
    rng = np.random.default_rng(77)
    n = 16
    mats = {}
    for chrom, W, origin in (("chr3", 4000, 500), ("chr8", 2500, 0)):
        m = (rng.random((n, W)) < 0.2).astype(np.uint8)
        m[1] = m[0]
        names = [f"S{i // 2:03d}#{i % 2 + 1}#{chrom}:{origin}-{origin + W}" for i in range(n)]
        matrixio.save_matrix(str(tmp_path / f"{chrom}.npz"), matrixio.from_dense(m, names, origin=origin, contig=f"CHM13#0#{chrom}"))
        mats[chrom] = (m, origin)
    (tmp_path / "g.bed").write_text("chr8\t100\t900\nchr3\t1000\t2000\nchrX\t5\t50\nCHM13#0#chr8\t900\t2500\nchr3\t2000\t4500\n")
    base = [sys.executable, os.path.join(ROOT, "scripts", "impop_scan.py"), "--bed", str(tmp_path / "g.bed"), "--format", "tajd"]
    r = subprocess.run(base + ["--matrix", str(tmp_path / "chr3.npz"), str(tmp_path / "chr8.npz")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Skipping region CHM13#0#chrX:5-50: no matrix holds chromosome chrX" in r.stderr
    lines = r.stdout.strip().split("\n")
    want_rows = [("chr8", 100, 900), ("chr3", 1000, 2000), ("chr8", 900, 2500), ("chr3", 2000, 4500)]
    assert len(lines) == 1 + len(want_rows)
    ones = oracle.pack_mask(np.ones(n, np.uint8))
    for line, (chrom, s, e) in zip(lines[1:], want_rows):
        m, origin = mats[chrom]
        bits = oracle.pack_hap_major(m)
        s0, s1 = s - origin, e - origin
        L = e - s
        sim = oracle.identity(oracle.pairwise_counts(bits, n, s0, s1), s1 - s0, 0)
        _, ps, _, _ = oracle.pica2(sim, 0.999, L, 5)
        S = oracle.window_sitecount(bits, n, s0, s1, ones, ones, ones, L)["s_all"]
        D, _ = oracle.tajimas_d(n, float(S), oracle.py_round(ps, 8))
        t = line.split("\t")
        assert t[:5] == [f"CHM13#0#{chrom}:{s}-{e}", str(L), str(n), str(S), f"{ps:.8f}"], (t, ps, S)
        assert (t[5] == "NA" and D != D) or abs(float(t[5]) - D) <= 1e-9 * abs(D)
    # one matrix only: the other chromosome's rows are skipped, not clamped
    r1 = subprocess.run(base + ["--matrix", str(tmp_path / "chr8.npz")], capture_output=True, text=True)
    assert r1.returncode == 0, r1.stderr
    assert r1.stderr.count("no matrix holds chromosome chr3") == 2
    assert r1.stdout.strip().split("\n")[1:] == [lines[1], lines[3]]
    dup = subprocess.run(base + ["--matrix", str(tmp_path / "chr8.npz"), str(tmp_path / "chr8.npz")], capture_output=True, text=True)
    assert dup.returncode == 2 and "two matrices for contig" in dup.stderr
