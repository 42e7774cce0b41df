#!/usr/bin/env python3
"""Generate tests/golden/*.json by running the REAL reference (pangenome/impop)
functions on seeded inputs.  Runs only in the build container, where
/root/reference exists; the GPU box only ever sees the JSON it writes.

The reference is imported by file path; nothing from it
is copied: the fixtures hold inputs (bit matrices, identity tables, argument
values) and the reference's full-precision outputs (float.hex()).

Usage: python3 oracle/gen_golden.py [--ref /root/reference] [--out tests/golden]
"""
from __future__ import annotations

import argparse
import base64
import importlib.util
import io
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

sys.dont_write_bytecode = True


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m  # tj_d.py's @dataclass needs the module registered
    spec.loader.exec_module(m)
    return m


def hx(x):
    return None if x is None else float(x).hex()


def names_for(n, ctg="chr1", s0=0, s1=1000):
    # PanSN-like, already in lexicographic order: sample#hap#contig:start-end
    out = []
    for i in range(n):
        out.append(f"S{(i // 2):04d}#{(i % 2) + 1}#{ctg}:{s0}-{s1}")
    assert out == sorted(out)
    return out


def b64(a: np.ndarray) -> str:
    return base64.b64encode(np.ascontiguousarray(a).tobytes()).decode()


def pack_rows(m01):
    n, W = m01.shape
    words = (W + 63) // 64
    pad = np.zeros((n, words * 64), dtype=np.uint8)
    pad[:, :W] = m01
    return np.packbits(pad, axis=1, bitorder="little").view(np.uint64).reshape(n, words)


def np_counts(m01):
    m = m01.astype(np.int64)
    return m @ m.T  # I_ij


def np_identity(I, W, kind):
    a = np.diag(I)
    n = I.shape[0]
    sim = np.zeros((n, n))
    for i in range(n):
        for j in range(n):
            if kind == "match":
                H = int(a[i] + a[j] - 2 * I[i, j])
                sim[i, j] = (W - H) / W
            else:
                d = int(a[i] + a[j])
                sim[i, j] = (2 * int(I[i, j])) / d if d else 1.0
    return sim


def sim_dict(sim, names):
    d = {}
    n = len(names)
    for i in range(n):
        for j in range(i, n):
            d[(names[i], names[j])] = float(sim[i, j])
    return d


def is_equivalence(sim, thr, rd):
    n = sim.shape[0]
    s = sim if rd is None else np.vectorize(lambda v: round(float(v), rd))(sim)
    adj = s > thr
    np.fill_diagonal(adj, True)
    # transitive <=> adj @ adj has the same support as adj
    reach = (adj.astype(np.int64) @ adj.astype(np.int64)) > 0
    return bool((reach == adj).all())


def founder_matrix(rng, n, W, n_founder, p_founder, p_private):
    anc = rng.integers(0, 2, size=W, dtype=np.uint8)
    f = np.repeat(anc[None, :], n_founder, axis=0)
    f ^= (rng.random((n_founder, W)) < p_founder).astype(np.uint8)
    who = rng.integers(0, n_founder, size=n)
    m = f[who].copy()
    if p_private > 0:
        m ^= (rng.random((n, W)) < p_private).astype(np.uint8)
    return m, who


# Every child process runs under a FIXED hash seed: where the reference's result depends on set order (pica2 /
# hud grouping on non-transitive tables) the fixture records which seed produced it, and a regeneration is
# byte-identical whatever seed this parent process happens to run under.
CHILD_ENV = dict(os.environ, PYTHONHASHSEED="0")


def seeded_child(spec_path):
    """Runs in a child under PYTHONHASHSEED=k: the REAL pica2.analyze_similarity_matrix / hud.calculate_fst on a
    non-transitive table, plus the set iteration orders they seeded their greedy groups from."""
    spec = json.load(open(spec_path))
    sc = spec["scripts"]
    pica2 = load("ref_pica2", os.path.join(sc, "pica2.py"))
    hud = load("ref_hud", os.path.join(sc, "hudson", "hud.py"))
    tjd = load("ref_tj_d", os.path.join(sc, "tj_d.py"))
    names = spec["names"]
    sim = [[float.fromhex(v) for v in row] for row in spec["sim"]]
    n = len(names)
    d = {(names[i], names[j]): sim[i][j] for i in range(n) for j in range(i, n)}
    elements = set()
    for k in spec["insert_order"]:  # the order a reader met the names in
        elements.add(names[k])
    out = {"hashseed": os.environ.get("PYTHONHASHSEED"), "order": list(set(elements)), "pica2": [], "hud": []}
    for thr, rd in spec["pica2_cases"]:
        log = io.StringIO()
        pi, ps = pica2.analyze_similarity_matrix(dict(d), elements, len(d), thr, spec["L"], log, rd)
        groups = [ln.strip() for ln in log.getvalue().splitlines() if ln.startswith("  G") and "(size:" in ln]
        # the rest of run_tajd.sh:166-180: pica2's "%.8f" stdout token -> tj_d.py -p, n = the list's line count, S given
        pi_text = f"{ps:.8f}"
        D = tjd.tajimas_d(n, float(spec["S"]), float(pi_text))
        out["pica2"].append({"threshold": float(thr).hex(), "round": rd, "pi": hx(pi), "pi_site": hx(ps), "n_groups": len(groups),
                             "pi_text": pi_text, "S": spec["S"], "D": hx(D)})
    A = set()
    for k in spec["insert_order"]:
        if spec["in_a"][k]:
            A.add(names[k])
    B = set()
    for k in spec["insert_order"]:
        if spec["in_b"][k]:
            B.add(names[k])
    out["order_a"], out["order_b"] = list(set(A)), list(set(B))
    for thr, rd in spec["hud_cases"]:
        r = hud.calculate_fst(d, A, B, spec["L"], rd, None, "grouped", thr)
        out["hud"].append({"threshold": float(thr).hex(), "round": rd, "out": {k: hx(v) for k, v in r.items()}})
    print(json.dumps(out))


def seeded_fixture(sc, out_dir, meta):
    """tests/golden/pica2_seeded.json: tables where "> threshold" is NOT transitive (the normal case at the
    pipeline defaults -t 0.999 -r 5).  The reference's greedy grouping then depends on set iteration order;
    for PYTHONHASHSEED = 0..K-1 record the order it iterated and the value it returned (library level), and
    the stdout + log of the real pica2.py CLI on the same table written as a .sim file."""
    rng = np.random.default_rng(20251102)
    tables = []
    def chain(n, step):  # SURVEY §8a-a3's example: identity 1 - step*|i-j|
        return np.array([[1.0 - step * abs(i - j) for j in range(n)] for i in range(n)])
    m, _ = founder_matrix(rng, 36, 900, 5, 0.01, 0.002)
    I = np_counts(m)
    # (0.999, 5) = the defaults of run_tajd.sh:9-10, last so that the CLI captures (first two cases) stay what they were
    specs = [("chain5", chain(5, 0.0004), 40, [(0.999, None), (0.999, 5)], [(0.999, None)]),
             ("chain12", chain(12, 0.0004), 10, [(0.999, None), (0.9985, 5), (0.999, 5)], [(0.999, None)]),
             ("chain31", chain(31, 0.00035), 10, [(0.999, None), (0.999, 3), (0.9975, None), (0.999, 5)], [(0.999, None), (0.998, 4)]),
             ("founders36_match", np_identity(I, 900, "match"), 10, None, None),
             ("founders36_dice", np_identity(I, 900, "dice"), 10, None, None)]
    for name, sim, n_seeds, pcases, hcases in specs:
        n = sim.shape[0]
        names = names_for(n, "chr7", 1000, 51000)
        off = sim[~np.eye(n, dtype=bool)]
        if pcases is None:
            q = [float(np.quantile(off, x)) for x in (0.5, 0.8)]
            pcases = [(q[0], None), (q[1], None), (q[1], 3), (q[1], 5)]
            hcases = [(q[0], None), (q[1], 4)]
        pcases = [c for c in pcases if not is_equivalence(sim, c[0], c[1])]
        assert pcases, name
        ins = rng.permutation(n).tolist()
        in_a = [1 if (i % 3) != 2 and i < (2 * n) // 3 else 0 for i in range(n)]
        in_b = [1 if not in_a[i] else 0 for i in range(n)]
        spec = {"scripts": sc, "names": names, "sim": [[hx(v) for v in row] for row in sim], "insert_order": ins,
                "pica2_cases": pcases, "hud_cases": hcases, "L": 50000, "in_a": in_a, "in_b": in_b, "S": 3 * n + 7}
        runs = []
        with tempfile.TemporaryDirectory() as td:
            sp = os.path.join(td, "spec.json")
            json.dump(spec, open(sp, "w"))
            simfile = os.path.join(td, f"{name}.sim")
            with open(simfile, "w") as f:  # rows in a shuffled order: the reader's insertion order is not the sorted one
                f.write("group.a\tgroup.b\testimated.identity\n")
                pairs = [(i, j) for i in range(n) for j in range(i, n)]
                for k in rng.permutation(len(pairs)):
                    i, j = pairs[k]
                    f.write(f"{names[i]}\t{names[j]}\t{float(sim[i, j])!r}\n")
            sim_text = open(simfile).read()
            for seed in range(n_seeds):
                env = dict(os.environ, PYTHONHASHSEED=str(seed))
                r = subprocess.run([sys.executable, "-B", os.path.abspath(__file__), "--seeded-child", sp], capture_output=True,
                                   text=True, env=env, check=True)
                rec = json.loads(r.stdout)
                rec["cli"] = []
                for thr, rd in pcases[:2]:
                    argv = [os.path.join(sc, "pica2.py"), simfile, "-t", repr(float(thr)), "-l", "50000", "-d", td] + (["-r", str(rd)] if rd is not None else [])
                    c = subprocess.run([sys.executable, "-B"] + argv, capture_output=True, text=True, env=env, cwd=td)
                    logp = os.path.join(td, f"{name}.log")
                    rec["cli"].append({"t": repr(float(thr)), "r": rd, "l": 50000, "stdout": c.stdout, "rc": c.returncode,
                                       "log": open(logp).read().replace(td, "<TMP>")})
                    os.remove(logp)
                runs.append(rec)
        tables.append({"name": name, "n": n, "names": names, "sim": spec["sim"], "insert_order": ins, "L": 50000,
                       "in_a": in_a, "in_b": in_b, "sim_text": sim_text, "runs": runs})
    json.dump({"meta": dict(meta, PYTHONHASHSEED="per run"), "tables": tables}, open(os.path.join(out_dir, "pica2_seeded.json"), "w"), indent=1)


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--seeded-child":
        return seeded_child(sys.argv[2])
    if os.environ.get("PYTHONHASHSEED") != "0":
        # h-fst / hud sum over Python sets, so even their deterministic results move in the last bits with the
        # string hash seed: the whole generator runs under seed 0 and a regeneration is byte-identical
        return sys.exit(subprocess.run([sys.executable, "-B", os.path.abspath(__file__)] + sys.argv[1:], env=CHILD_ENV).returncode)
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
    args = ap.parse_args()
    sc = os.path.join(args.ref, "scripts")
    pica2 = load("ref_pica2", os.path.join(sc, "pica2.py"))
    hfst = load("ref_hfst", os.path.join(sc, "h-fst.py"))
    tjd = load("ref_tj_d", os.path.join(sc, "tj_d.py"))
    af = load("ref_af", os.path.join(sc, "af.py"))
    hud = load("ref_hud", os.path.join(sc, "hudson", "hud.py"))
    os.makedirs(args.out, exist_ok=True)
    meta = {"python": sys.version.split()[0], "PYTHONHASHSEED": "0 (the generator re-executes itself under it)",
            "generator": "oracle/gen_golden.py", "reference": "pangenome/impop @ /root/reference (2025-10-31 snapshot)"}

    # ------------------------------------------------------------------ tajima
    taj = []
    cases = [(446, 20.0, 0.59146123),  # doc/how_tjd.md:45
             (465, 2543.0, 1.234e-06), (465, 2543.0, 61.7), (2, 1.0, 0.5), (3, 0.0, 0.0), (10, 1.0, 0.0),
             (8, 17.0, 3.3e-07), (100, 1e6, 2.5e-3), (465, 1.0, 1e-8), (5000, 12345.0, 0.001)]
    rng = np.random.default_rng(7)
    for _ in range(30):
        cases.append((int(rng.integers(2, 3000)), float(rng.integers(0, 100000)), float(rng.random() * 10.0 ** int(rng.integers(-9, 2)))))
    for n, S, pi in cases:
        D, c = tjd.tajimas_d(n, S, pi, return_components=True)
        taj.append({"n": n, "S": hx(S), "pi": hx(pi), "D": hx(D),
                    "comps": [hx(v) for v in (c.a1, c.a2, c.b1, c.b2, c.c1, c.c2, c.e1, c.e2, c.numerator, c.denominator)]})
    errs = []
    for n, S, pi in [(1, 1.0, 0.1), (5, -1.0, 0.1), (5, 1.0, -0.1)]:
        try:
            tjd.tajimas_d(n, S, pi)
            errs.append({"n": n, "S": S, "pi": pi, "error": None})
        except ValueError as e:
            errs.append({"n": n, "S": S, "pi": pi, "error": str(e)})
    json.dump({"meta": meta, "cases": taj, "errors": errs}, open(os.path.join(args.out, "tajima.json"), "w"), indent=1)

    # ------------------------------------------- the 6-sequence table
    # data rows of scripts/hudson/example_fst_methods.py:8-24 (a data table, not code)
    rows6 = [("seq1_popA", "seq2_popA", 0.9995), ("seq1_popA", "seq3_popA", 0.9993), ("seq2_popA", "seq3_popA", 0.9998),
             ("seq1_popA", "seq4_popB", 0.9950), ("seq1_popA", "seq5_popB", 0.9948), ("seq1_popA", "seq6_popB", 0.9952),
             ("seq2_popA", "seq4_popB", 0.9951), ("seq2_popA", "seq5_popB", 0.9949), ("seq2_popA", "seq6_popB", 0.9953),
             ("seq3_popA", "seq4_popB", 0.9949), ("seq3_popA", "seq5_popB", 0.9947), ("seq3_popA", "seq6_popB", 0.9951),
             ("seq4_popB", "seq5_popB", 0.9996), ("seq4_popB", "seq6_popB", 0.9994), ("seq5_popB", "seq6_popB", 0.9997)]
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "example_similarities.tsv")
        with open(p, "w") as f:
            f.write("group.a\tgroup.b\testimated.identity\n")
            for a, b, v in rows6:
                f.write(f"{a}\t{b}\t{v!r}\n")
        d, elements, pc = pica2.read_similarity_file(p)
        six = {"rows": [[a, b, hx(v)] for a, b, v in rows6], "pica2": [], "hfst": [], "af": []}
        el6 = sorted(elements)
        dense6 = np.eye(6)
        for a, b, v in rows6:
            dense6[el6.index(a), el6.index(b)] = dense6[el6.index(b), el6.index(a)] = v
        for thr in (1.0, 0.9996, 0.999, 0.99):
            for rd in (None, 3):
                if not is_equivalence(dense6, thr, rd):
                    continue  # seed-order dependent in the reference: no golden
                pi, ps = pica2.analyze_similarity_matrix(dict(d), set(elements), pc, thr, 1000, io.StringIO(), rd)
                six["pica2"].append({"threshold": hx(thr), "round": rd, "L": 1000, "pi": hx(pi), "pi_site": hx(ps)})
        A = {"seq1_popA", "seq2_popA", "seq3_popA"}
        B = {"seq4_popB", "seq5_popB", "seq6_popB"}
        hd, hs = hfst.read_similarity_file(p)
        for L, rd in ((None, None), (1000, None), (1000000, 3)):
            r = hfst.calculate_fst(hd, set(A), set(B), L, rd)
            six["hfst"].append({"L": L, "round": rd, "out": {k: hx(v) for k, v in r.items()}})
        rows, samples = af.load_pairs(p)
        for thr in (0.999, 0.9996, 1.0, 0.99):
            cl = af.cluster(rows, samples, thr)
            six["af"].append({"threshold": hx(thr), "clusters": [sorted(c) for c in cl]})
        # CLI-level goldens (stdout text) for the drop-in scripts
        pa, pb = os.path.join(td, "pop_A.txt"), os.path.join(td, "pop_B.txt")
        open(pa, "w").write("seq1_popA\nseq2_popA\nseq3_popA\n")
        open(pb, "w").write("seq4_popB\nseq5_popB\nseq6_popB\n")
        def run(argv, log=None):
            r = subprocess.run([sys.executable, "-B"] + argv, capture_output=True, text=True, cwd=td, env=CHILD_ENV)
            out = {"argv": ["<TMP>" if a == td else os.path.basename(a) if a.startswith(td) or a.startswith(sc) else a for a in argv],
                   "stdout": r.stdout.replace(td, "<TMP>"), "rc": r.returncode}
            if log and os.path.exists(os.path.join(td, log)):  # the script's log file, temp dir masked
                out["log"] = open(os.path.join(td, log)).read().replace(td, "<TMP>")
                os.remove(os.path.join(td, log))
            return out
        six["cli"] = {
            "pica2": [run([os.path.join(sc, "pica2.py"), p, "-t", t, "-l", "1000", "-d", td] + (["-r", r] if r else []),
                          log="example_similarities.log")
                      for t, r in (("0.999", "5"), ("1.0", None), ("0.9996", "4"), ("0.99", None))]
                     + [run([os.path.join(sc, "pica2.py"), p, "-t", "0.999", "-d", td], log="example_similarities.log")],
            # NB: bare names become the prefix 'seq1_popA#' (h-fst.py:57-61) and match nothing -> rc 1
            "hfst": [run([os.path.join(sc, "h-fst.py"), p, "-a", pa, "-b", pb, "-l", "1000000", "-d", td])],
            "tj_d": [run([os.path.join(sc, "tj_d.py"), "-n", "446", "-p", "0.59146123", "-S", "20"]),
                     run([os.path.join(sc, "tj_d.py"), "-n", "465", "-p", "0.00000123", "-S", "2543", "--show-components"]),
                     run([os.path.join(sc, "tj_d.py"), "-n", "10", "-p", "0.1", "-S", "0"])],
            "af": [run([os.path.join(sc, "af.py"), "--input", p, "--threshold", "0.999"]),
                   run([os.path.join(sc, "af.py"), "--input", p, "--threshold", "0.9996"])],
            # the three invocations of scripts/hudson/example_fst_methods.py:47-58 (its `fst.py` is hud.py);
            # inside both populations every identity is > 0.999, so the grouping does not depend on set order
            "hud": [run([os.path.join(sc, "hudson", "hud.py"), p, "-a", pa, "-b", pb, "-l", "1000000", "-d", td] + extra)
                    for extra in (["-m", "direct"], ["-m", "grouped", "-t", "0.999"], ["-m", "grouped", "-t", "0.996"])],
        }
        # library-level values of the same three calls
        six["hud"] = []
        hd2, _ = hud.read_similarity_file(p)
        for method, thr in (("direct", 0.999), ("grouped", 0.999), ("grouped", 0.996)):
            r = hud.calculate_fst(hd2, set(A), set(B), 1000000, None, None, method, thr)
            six["hud"].append({"method": method, "threshold": hx(thr), "L": 1000000, "out": {k: hx(v) for k, v in r.items()}})
    json.dump({"meta": meta, **six}, open(os.path.join(args.out, "six_seq.json"), "w"), indent=1)

    # ------------------------------------------ population-name expansion (a6)
    seqs = ["HG00097#1#CM094061.1:100-200", "HG00097#2#CM094062.1:100-200", "HG01891#1#JA1:5-9", "HG01891#2#JA2:5-9",
            "NA12878#1#c:1-2", "NA128#1#c:1-2", "CHM13#0#chr1:100-200"]
    raw = ["HG00097_hap1_hprc_r2_v1.0.1", "HG01891_pat_hprc_r2_v1.0.1", "HG01891_mat", "NA128", "NA12878#1", "CHM13#0#",
           "missing_sample", "", "#comment", "  HG00097_hap2  ", "NA12878#1#c:1-2"]
    canon = [hfst.canonicalize_identifier(r) for r in raw]
    exp, missing = hfst.expand_population(raw, set(seqs))
    json.dump({"meta": meta, "sequences": seqs, "raw": raw, "canonical": canon, "expanded": sorted(exp), "missing": missing},
              open(os.path.join(args.out, "popnames.json"), "w"), indent=1)

    # ------------------------------------------------ bit-matrix -> statistics
    rng = np.random.default_rng(20251031)
    mats = []
    specs = [
        # name, n, W, founders, p_founder, p_private, L
        ("n8_w1000", 8, 1000, 3, 0.02, 0.002, 1000),     # BASELINE config 1 scale (8 haplotypes)
        ("n61_w777", 61, 777, 5, 0.03, 0.004, 777),
        ("n24_w300_clones", 24, 300, 4, 0.05, 0.0, 300),   # exact clones: '>thr' is an equivalence relation
        ("n40_w2000_clones", 40, 2000, 6, 0.01, 0.0, 50000),
        ("n130_w640", 130, 640, 8, 0.02, 0.003, 640),    # >128 haplotypes: second mask word
        # founders 1-2 % apart, members of a founder a handful of private sites apart: at run_tajd.sh's defaults
        # (-t 0.999 -r 5) the groups are the founder classes (non-identical members, transitive) — asserted below
        ("n48_w12000_tight", 48, 12000, 5, 0.01, 0.00015, 12000),
    ]
    for name, n, W, nf, pf, pp, L in specs:
        m, who = founder_matrix(rng, n, W, nf, pf, pp)
        names = names_for(n, "chr1", 0, W)
        I = np_counts(m)
        inA = np.zeros(n, dtype=np.uint8)
        inB = np.zeros(n, dtype=np.uint8)
        perm = rng.permutation(n)
        inA[perm[: n // 3]] = 1
        inB[perm[n // 3: n // 3 + n // 4]] = 1
        rec = {"name": name, "n": n, "W": W, "L": L, "names": names, "bits_u64_b64": b64(pack_rows(m)),
               "in_a": inA.tolist(), "in_b": inB.tolist(), "founder_of": who.tolist(),
               "S_all": int(((m.sum(0) > 0) & (m.sum(0) < n)).sum()), "I_b64": b64(I.astype(np.int64)), "kinds": {}}
        for kind in ("match", "dice"):
            sim = np_identity(I, W, kind)
            d = sim_dict(sim, names)
            A = {names[i] for i in range(n) if inA[i]}
            B = {names[i] for i in range(n) if inB[i]}
            out = {"pica2": [], "hfst": [], "af": []}
            thr_list = [1.0, 1.5]
            offdiag = sim[~np.eye(n, dtype=bool)]
            for thr in (0.999, 0.99, float(np.median(offdiag)), float(offdiag.max()) - 1e-12):
                thr_list.append(thr)
            for thr in thr_list:
                for rd in (None, 5, 2):
                    eq = is_equivalence(sim, thr, rd)
                    if not eq:
                        continue  # seed-order dependent in the reference: no golden
                    for Lx in (L, None):
                        pi, ps = pica2.analyze_similarity_matrix(dict(d), set(names), len(d), thr, Lx, io.StringIO(), rd)
                        out["pica2"].append({"threshold": hx(thr), "round": rd, "L": Lx, "pi": hx(pi), "pi_site": hx(ps)})
            for Lx, rd in ((L, None), (None, None), (L, 5), (L, 3)):
                r = hfst.calculate_fst(d, set(A), set(B), Lx, rd)
                out["hfst"].append({"L": Lx, "round": rd, "out": {k: hx(v) for k, v in r.items()}})
            # overlapping populations (h-fst.py:181-185)
            Bov = set(B) | set(sorted(A)[:2])  # sorted(): the fixture must not depend on the hash seed
            r = hfst.calculate_fst(d, set(A), set(Bov), L, None)
            out["hfst_overlap"] = {"extra_in_b": sorted(A)[:2], "L": L, "out": {k: hx(v) for k, v in r.items()}}
            # hud.py grouped method: only where '> thr' is an equivalence relation inside BOTH populations
            out["hud_grouped"] = []
            for thr in (0.999, 0.99, float(np.median(offdiag)), 1.0):
                for rd in (None, 4):
                    ia, ib = [i for i in range(n) if inA[i]], [i for i in range(n) if inB[i]]
                    if not (is_equivalence(sim[np.ix_(ia, ia)], thr, rd) and is_equivalence(sim[np.ix_(ib, ib)], thr, rd)):
                        continue
                    r = hud.calculate_fst(d, set(A), set(B), L, rd, None, "grouped", thr)
                    out["hud_grouped"].append({"threshold": hx(thr), "round": rd, "L": L, "out": {k: hx(v) for k, v in r.items()}})
            rows = [(a.split(":", 1)[0], b.split(":", 1)[0], v) for (a, b), v in d.items()]
            samples = sorted({a for a, _, _ in rows} | {b for _, b, _ in rows})
            for thr in (1.0, 0.999, float(np.median(offdiag)), 0.0):
                cl = af.cluster(rows, samples, thr)
                out["af"].append({"threshold": hx(thr), "clusters": [sorted(c) for c in cl]})
            rec["kinds"][kind] = out
        # The DEFAULT chain of run_tajd.sh (:9-10 THRESHOLD=0.999 R_VALUE=5; :166 pica2.py -t T -l LENGTH -r R; :174 first
        # stdout token; :180 tj_d.py -n SAMPLE_COUNT -p PI -S S_COUNT), where the grouping does not depend on set order.
        # "all": every haplotype listed; "subset_a": the sample list = population A (impg similarity --subset-sequence-list,
        # :160), n = its size, S still from the whole graph (:126,148).
        rec["tajd_chain_default"] = {}
        for kind in ("match", "dice"):
            sim = np_identity(I, W, kind)
            ent = {}
            for label, idx in (("all", list(range(n))), ("subset_a", [i for i in range(n) if inA[i]])):
                sub = sim[np.ix_(idx, idx)]
                if len(idx) < 2 or not is_equivalence(sub, 0.999, 5):
                    ent[label] = None
                    continue
                nm = [names[i] for i in idx]
                log = io.StringIO()
                pi, ps = pica2.analyze_similarity_matrix(sim_dict(sub, nm), set(nm), 0, 0.999, L, log, 5)
                groups = [ln for ln in log.getvalue().splitlines() if ln.startswith("  G") and "(size:" in ln]
                pi_text = f"{ps:.8f}"
                D = tjd.tajimas_d(len(idx), float(rec["S_all"]), float(pi_text))
                ent[label] = {"threshold": "0.999", "round": 5, "L": L, "n": len(idx), "S": rec["S_all"], "pi": hx(pi), "pi_site": hx(ps),
                              "pi_text": pi_text, "n_groups": len(groups), "D": hx(D)}
            rec["tajd_chain_default"][kind] = ent
        if name == "n48_w12000_tight":
            for kind in ("match", "dice"):
                e = rec["tajd_chain_default"][kind]["all"]
                assert e is not None and 1 < e["n_groups"] < n, (name, kind, e)
        # full chain as wired by run_tajd.sh:166-180 on the `match` identity at t>=1
        sim = np_identity(I, W, "match")
        pi, ps = pica2.analyze_similarity_matrix(sim_dict(sim, names), set(names), 0, 1.0, L, io.StringIO(), None)
        pi_text = f"{ps:.8f}"  # pica2.py:226 -> run_tajd.sh:174
        D = tjd.tajimas_d(n, float(rec["S_all"]), float(pi_text))
        rec["tajd_chain"] = {"pi_text": pi_text, "S": rec["S_all"], "n": n, "D": hx(D)}
        mats.append(rec)
    json.dump({"meta": meta, "matrices": mats}, open(os.path.join(args.out, "bitmatrix.json"), "w"), indent=1)

    # -------------------------- CLI goldens on a PanSN-named .sim (n8_w1000, match)
    m8 = mats[0]
    bits8 = np.frombuffer(base64.b64decode(m8["bits_u64_b64"]), dtype=np.uint64).reshape(m8["n"], -1)
    m01 = np.unpackbits(bits8.view(np.uint8), axis=1, bitorder="little")[:, : m8["W"]]
    sim8 = np_identity(np_counts(m01), m8["W"], "match")
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "win8.sim")
        lines = ["group.a\tgroup.b\tgroup.a.length\tgroup.b.length\tintersection\testimated.identity"]
        for i in range(8):
            for j in range(8):  # impg-style: all ordered pairs, extra columns ignored (pica2.py:22)
                lines.append(f"{m8['names'][i]}\t{m8['names'][j]}\t1000\t1000\t0\t{float(sim8[i, j])!r}")
        open(p, "w").write("\n".join(lines) + "\n")
        pa, pb = os.path.join(td, "popA.txt"), os.path.join(td, "popB.txt")
        open(pa, "w").write("# population A\nS0000_hap1_hprc_r2_v1.0.1\nS0001\n\n")
        open(pb, "w").write("S0002_mat_hprc_r2_v1.0.1\nS0003#2\nS0002_pat\nNOPE_hap1\n")
        def run2(argv, log=None):
            r = subprocess.run([sys.executable, "-B"] + argv, capture_output=True, text=True, cwd=td, env=CHILD_ENV)
            out = {"argv": ["<TMP>" if a == td else os.path.basename(a) if a.startswith(td) or a.startswith(sc) else a for a in argv],
                   "stdout": r.stdout.replace(td, "<TMP>"), "stderr": r.stderr.replace(td, "<TMP>"), "rc": r.returncode}
            if log and os.path.exists(os.path.join(td, log)):
                out["log"] = open(os.path.join(td, log)).read().replace(td, "<TMP>")
                os.remove(os.path.join(td, log))
            return out
        cli = {"sim_text": open(p).read(), "popA": open(pa).read(), "popB": open(pb).read(),
               "pica2": [run2([os.path.join(sc, "pica2.py"), p, "-t", t, "-l", "1000", "-d", td] + (["-r", r] if r else []),
                              log="win8.log")
                         for t, r in (("1.0", None), ("0.999", "5"), ("0.99", None))],
               "hfst": [run2([os.path.join(sc, "h-fst.py"), p, "-a", pa, "-b", pb, "-l", "1000", "-d", td], log="win8_fst.log"),
                        run2([os.path.join(sc, "h-fst.py"), p, "-a", pa, "-b", pb, "-d", td, "-r", "4"], log="win8_fst.log"),
                        run2([os.path.join(sc, "h-fst.py"), p, "-a", pa, "-b", pa, "-d", td], log="win8_fst.log")],
               "af": [run2([os.path.join(sc, "af.py"), "--input", p, "--threshold", "1.0"]),
                      run2([os.path.join(sc, "af.py"), "--input", p, "--threshold", "0.99"])],
               "errors": [run2([os.path.join(sc, "pica2.py"), os.path.join(td, "nope.sim"), "-d", td]),
                          run2([os.path.join(sc, "h-fst.py"), os.path.join(td, "nope.sim"), "-a", pa, "-b", pb, "-d", td])]}
        bad = os.path.join(td, "bad.sim")
        open(bad, "w").write("a\tb\tc\nx\ty\t0.5\n")
        cli["errors"].append(run2([os.path.join(sc, "pica2.py"), bad, "-d", td]))
        bad2 = os.path.join(td, "bad2.sim")
        open(bad2, "w").write("group.a\tgroup.b\testimated.identity\nx\ty\tzzz\n")
        cli["errors"].append(run2([os.path.join(sc, "pica2.py"), bad2, "-d", td]))
        # scripts/hudson/hud.py CLI (exact-name populations): direct + grouped; log file captured too
        ha, hb = os.path.join(td, "hudA.txt"), os.path.join(td, "hudB.txt")
        open(ha, "w").write("# A\n" + "\n".join(m8["names"][:4]) + "\nNOPE#1#x:0-1\n")
        open(hb, "w").write("\n".join(m8["names"][3:8]) + "\n")
        hud_py = os.path.join(sc, "hudson", "hud.py")
        def run_hud(extra):
            r = run2([hud_py, p, "-a", ha, "-b", hb, "-d", td] + extra)
            r["log"] = open(os.path.join(td, "win8_fst.log")).read()
            return r
        cli["hudA"], cli["hudB"] = open(ha).read(), open(hb).read()
        hud_runs = [run_hud(["-l", "1000"]), run_hud(["-m", "direct", "-r", "3", "-v"]), run_hud(["-m", "grouped", "-t", "1.0", "-l", "1000"])]
        for t, r in (("0.999", "5"), ("0.99", None), ("0.995", None)):
            subA = [i for i in range(0, 3)]; subB = [i for i in range(4, 8)]  # index 3 is in both -> dropped
            ok = all(is_equivalence(sim8[np.ix_(sub, sub)], float(t), int(r) if r else None) for sub in (subA, subB))
            if ok:
                hud_runs.append(run_hud(["-m", "grouped", "-t", t] + (["-r", r] if r else []) + ["-v"]))
        cli["hud"] = hud_runs
        # is pica2 @ t=0.999 -r 5 order-independent on this table?
        cli["pica2_equivalence"] = [is_equivalence(sim8, 1.0, None), is_equivalence(sim8, 0.999, 5), is_equivalence(sim8, 0.99, None)]
    json.dump({"meta": meta, **cli}, open(os.path.join(args.out, "cli_pansn.json"), "w"), indent=1)

    # ---------------------------------------------- EHH (scripts/wip/ehhgfa.py calc_EHH + CLI)
    ehh_mod = load("ref_ehhgfa", os.path.join(sc, "wip", "ehhgfa.py"))
    rng = np.random.default_rng(20251101)
    ehh = {"meta": meta, "calc": [], "cli": []}
    for n, W, nf, pf, pp in ((1, 5, 1, 0.0, 0.0), (2, 9, 2, 0.2, 0.0), (5, 19, 2, 0.15, 0.02), (8, 40, 3, 0.08, 0.01),
                             (12, 130, 3, 0.03, 0.004), (7, 64, 2, 0.05, 0.0), (9, 65, 3, 0.02, 0.003), (6, 1, 2, 0.5, 0.0)):
        h = founder_matrix(rng, n, W, nf, pf, pp)[0].astype(np.int64)
        ehh["calc"].append({"n": n, "W": W, "rows": ["".join(map(str, r)) for r in h.tolist()],
                            "fwd": [hx(float(v)) for v in ehh_mod.calc_EHH(h)],
                            "rev": [hx(float(v)) for v in ehh_mod.calc_EHH(np.flip(h, axis=1))]})
    hv = rng.integers(0, 10, size=(6, 14))  # value matrix in the style of ehh2.py's digit examples
    hv[1] = hv[0]; hv[2, :9] = hv[0, :9]; hv[4, :5] = hv[3, :5]
    ehh["values"] = {"rows": hv.tolist(), "fwd": [hx(float(v)) for v in ehh_mod.calc_EHH(hv)],
                     "rev": [hx(float(v)) for v in ehh_mod.calc_EHH(np.flip(hv, axis=1))]}
    with tempfile.TemporaryDirectory() as td:
        for n, Wt, w, ptest, refpos in ((10, 60, 20, 5, 1), (8, 48, 16, 1, 3), (9, 50, 20, 8, 2), (6, 30, 10, 10, 1)):
            h = founder_matrix(rng, n, Wt, 3, 0.06, 0.01)[0].astype(np.int64)
            h[h[:, 0] == 1, 3] = 7  # a non 0/1 entry: ehhgfa.py:50 maps non-zero to 1
            f, o = os.path.join(td, "hap.txt"), os.path.join(td, "out.txt")
            np.savetxt(f, h, fmt="%d")
            if os.path.exists(o):
                os.remove(o)
            r = subprocess.run([sys.executable, "-B", os.path.join(sc, "wip", "ehhgfa.py"), "-i", f, "-p", str(ptest), "-w", str(w),
                                "-refpos", str(refpos), "-o", o], capture_output=True, text=True, cwd=td)
            ehh["cli"].append({"matrix_text": open(f).read(), "p": ptest, "w": w, "refpos": refpos, "rc": r.returncode,
                               "out": open(o).read() if os.path.exists(o) else None,
                               "stderr_last": r.stderr.strip().splitlines()[-1] if r.stderr.strip() else ""})
    json.dump(ehh, open(os.path.join(args.out, "ehh.json"), "w"), indent=1)

    # ---------------------------------------------- missing pairs / ragged .sim
    n = 7
    names = names_for(n)
    rng = np.random.default_rng(5)
    sim = np.round(0.99 + 0.01 * rng.random((n, n)), 6)
    sim = np.triu(sim, 1) + np.triu(sim, 1).T + np.eye(n)
    d = sim_dict(sim, names)
    drop = [(names[0], names[3]), (names[2], names[5]), (names[1], names[1])]
    for k in drop:
        d.pop(k, None)
    rag = {"names": names, "sim": [[hx(v) for v in row] for row in sim], "dropped": [list(k) for k in drop], "pica2": [], "hfst": []}
    for thr in (1.0,):
        pi, ps = pica2.analyze_similarity_matrix(dict(d), set(names), len(d), thr, 500, io.StringIO(), None)
        rag["pica2"].append({"threshold": hx(thr), "round": None, "L": 500, "pi": hx(pi), "pi_site": hx(ps)})
    r = hfst.calculate_fst(d, set(names[:3]), set(names[3:]), 500, None)
    rag["hfst"].append({"a": names[:3], "b": names[3:], "L": 500, "out": {k: hx(v) for k, v in r.items()}})
    # degenerate inputs
    pi0 = pica2.analyze_similarity_matrix({}, set(), 0, 1.0, 100, io.StringIO(), None)
    pi1 = pica2.analyze_similarity_matrix({("x", "x"): 1.0}, {"x"}, 1, 1.0, 100, io.StringIO(), None)
    rag["degenerate"] = {"empty": [hx(v) for v in pi0], "single": [hx(v) for v in pi1]}
    json.dump({"meta": meta, **rag}, open(os.path.join(args.out, "ragged.json"), "w"), indent=1)
    # ---------------------------------------------- Fst where Dxy - pi_xy cancels (tolerance policy)
    # Every pair of haplotypes differs at exactly the same number of sites, so pi_A = pi_B = Dxy in exact arithmetic and
    # Fst = Da = 0; h-fst.py sums (1 - sim) over Python sets (h-fst.py:141-171), so what it RETURNS is 0 or a value of the
    # order of one rounding error of the sums (it moves with PYTHONHASHSEED).  Captured here under hash seed 0.
    n, k_priv, k_shared = 30, 7, 100
    W = k_shared + n * k_priv + 13
    m = np.zeros((n, W), dtype=np.uint8)
    m[:, :k_shared] = 1
    for i in range(n):
        m[i, k_shared + i * k_priv: k_shared + (i + 1) * k_priv] = 1
    names = names_for(n, "chr4", 0, W)
    inA = np.array([1 if i < 12 else 0 for i in range(n)], dtype=np.uint8)
    inB = 1 - inA
    I = np_counts(m)
    canc = {"meta": meta, "n": n, "W": W, "names": names, "bits_u64_b64": b64(pack_rows(m)), "in_a": inA.tolist(), "in_b": inB.tolist(),
            "kinds": {}}
    for kind in ("match", "dice"):
        sim = np_identity(I, W, kind)
        d = sim_dict(sim, names)
        A = {names[i] for i in range(n) if inA[i]}
        B = {names[i] for i in range(n) if inB[i]}
        runs = []
        for Lx, rd in ((W, None), (W, 5), (None, None), (50000, 3)):
            r = hfst.calculate_fst(d, set(A), set(B), Lx, rd)
            runs.append({"L": Lx, "round": rd, "out": {k: hx(v) for k, v in r.items()}})
        canc["kinds"][kind] = runs
    json.dump(canc, open(os.path.join(args.out, "fst_cancel.json"), "w"), indent=1)

    # ---------------------------------------------- allele counts per node column (scripts/wip/op-afs.py)
    # The reference reads an `odgi paths -H` table (3 metadata columns, then one 0/1 column per node: op-afs.py:112) and, per
    # column, returns the count and frequency of the value the FIRST data row holds (allele_freq returns inside the first
    # iteration over its dict, op-afs.py:26-44); a monomorphic column makes main() fail (None is unpacked, :115), so the
    # fixture has none.  Stored: the table text, what allele_freq returned per column, and main()'s counts_d / counts_f.
    os.environ.setdefault("MPLBACKEND", "Agg")
    afs_mod = load("ref_op_afs", os.path.join(sc, "wip", "op-afs.py"))
    rng = np.random.default_rng(20251103)
    n_path, n_node = 13, 57
    tab = (rng.random((n_path, n_node)) < rng.random(n_node)[None, :] * 0.8 + 0.1).astype(np.int64)
    for c in range(n_node):  # no monomorphic column
        if tab[:, c].min() == tab[:, c].max():
            tab[int(rng.integers(0, n_path)), c] ^= 1
    pnames = [f"S{int(k) // 2:03d}#{int(k) % 2 + 1}#chr5:0-{n_node}" for k in rng.permutation(n_path)]  # file order != sorted order
    header = ["path.name", "path.length", "node.count"] + [f"node.{c + 1}" for c in range(n_node)]
    text = "\t".join(header) + "\n" + "".join(
        "\t".join([pnames[r], str(100 + r), str(int(tab[r].sum()))] + [str(int(v)) for v in tab[r]]) + "\n" for r in range(n_path))
    with tempfile.TemporaryDirectory() as td:
        fp = os.path.join(td, "paths.tsv")
        open(fp, "w").write(text)
        df = afs_mod.read_file_to_matrix(fp)
        cols, counts_d, counts_f = [], {}, {}
        for column in df.columns[3:]:  # op-afs.py:112-118
            label, value, count, freq = afs_mod.allele_freq(df[column].iloc[0:].tolist(), column)
            cols.append({"label": str(label), "value": int(value), "count": int(count), "freq": hx(freq)})
            counts_d.setdefault(int(value), []).append(int(count))
            counts_f.setdefault(int(value), []).append(hx(freq))
    json.dump({"meta": meta, "table_text": text, "n_path": n_path, "n_node": n_node, "columns": cols,
               "counts_d": {str(k): v for k, v in sorted(counts_d.items())}, "counts_f": {str(k): v for k, v in sorted(counts_f.items())}},
              open(os.path.join(args.out, "afs_table.json"), "w"), indent=1)
    seeded_fixture(sc, args.out, meta)
    print("wrote goldens to", os.path.abspath(args.out))


if __name__ == "__main__":
    main()
