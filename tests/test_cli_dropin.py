"""GPU: the drop-in CLIs (scripts/*.py) must print what the reference's CLIs printed for the
same files (stdout/stderr/exit code captured by oracle/gen_golden.py from the real reference)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT, fh, load_golden

pytestmark = pytest.mark.gpu
SCRIPTS = os.path.join(ROOT, "scripts")


def run(script, argv, cwd, hashseed="0"):
    # the goldens were captured under PYTHONHASHSEED=0 (oracle/gen_golden.py CHILD_ENV): where the reference's
    # result depends on set iteration order (pica2 / hud grouping) the drop-in reproduces it under the same seed
    r = subprocess.run([sys.executable, os.path.join(SCRIPTS, script)] + argv, capture_output=True, text=True, cwd=cwd,
                       env=dict(os.environ, PYTHONHASHSEED=str(hashseed)))
    # libdrm on the GPU box prints a notice about a missing amdgpu.ids table at device open:
    # environment noise, not something the scripts write
    r.stderr = "".join(l for l in r.stderr.splitlines(True) if "amdgpu.ids" not in l)
    return r


def remap(argv, td):
    """golden argv holds basenames for files that lived in the generator's temp dir"""
    out = []
    for a in argv[1:]:
        if a == "<TMP>":
            out.append(td)  # the -d log dir
        elif os.path.exists(os.path.join(td, a)):
            out.append(os.path.join(td, a))
        elif a.endswith(".sim") or a.endswith(".tsv"):
            out.append(os.path.join(td, a))
        else:
            out.append(a)
    return out


def test_six_sequence_cli(tmp_path):
    g = load_golden("six_seq.json")
    td = str(tmp_path)
    with open(os.path.join(td, "example_similarities.tsv"), "w") as f:
        f.write("group.a\tgroup.b\testimated.identity\n")
        for a, b, v in g["rows"]:
            f.write(f"{a}\t{b}\t{fh(v)!r}\n")
    open(os.path.join(td, "pop_A.txt"), "w").write("seq1_popA\nseq2_popA\nseq3_popA\n")
    open(os.path.join(td, "pop_B.txt"), "w").write("seq4_popB\nseq5_popB\nseq6_popB\n")
    for script, key in (("pica2.py", "pica2"), ("h-fst.py", "hfst"), ("tj_d.py", "tj_d"), ("af.py", "af"), ("hud.py", "hud")):
        for c in g["cli"][key]:
            r = run(script, remap(c["argv"], td), td)
            assert r.returncode == c["rc"], (script, c["argv"], r.stderr)
            if key == "tj_d":
                # repr(float) text: allow the last digits to differ within 1e-12 relative
                gl, wl = r.stdout.strip().splitlines(), c["stdout"].strip().splitlines()
                assert len(gl) == len(wl)
                for a, b in zip(gl, wl):
                    if a != b:
                        ta, tb = a.replace("=", " ").split(), b.replace("=", " ").split()
                        assert len(ta) == len(tb)
                        for x, y in zip(ta, tb):
                            try:
                                fx, fy = float(x), float(y)
                            except ValueError:
                                assert x == y
                                continue
                            assert (fx != fx and fy != fy) or abs(fx - fy) <= 1e-12 * max(abs(fx), abs(fy)), (a, b)
            else:
                assert r.stdout == c["stdout"], (script, c["argv"], r.stdout, c["stdout"], r.stderr)
            if key == "pica2":  # the log file body is the reference's, byte for byte (pica2.py:113-167, 200-222)
                assert open(os.path.join(td, "example_similarities.log")).read().replace(td, "<TMP>") == c["log"], c["argv"]


def test_pansn_cli(tmp_path):
    g = load_golden("cli_pansn.json")
    td = str(tmp_path)
    open(os.path.join(td, "win8.sim"), "w").write(g["sim_text"])
    open(os.path.join(td, "popA.txt"), "w").write(g["popA"])
    open(os.path.join(td, "popB.txt"), "w").write(g["popB"])
    open(os.path.join(td, "bad.sim"), "w").write("a\tb\tc\nx\ty\t0.5\n")
    open(os.path.join(td, "bad2.sim"), "w").write("group.a\tgroup.b\testimated.identity\nx\ty\tzzz\n")
    for script, key in (("pica2.py", "pica2"), ("h-fst.py", "hfst"), ("af.py", "af")):
        for c in g[key]:
            r = run(script, remap(c["argv"], td), td)
            assert r.returncode == c["rc"], (script, c["argv"], r.stderr)
            assert r.stdout == c["stdout"], (script, c["argv"], r.stdout, c["stdout"])
            assert r.stderr == c["stderr"], (script, c["argv"], r.stderr, c["stderr"])
            if "log" in c:  # pica2.py:113-167 / h-fst.py:187-231: the log bodies are the reference's
                logname = "win8.log" if key == "pica2" else "win8_fst.log"
                assert open(os.path.join(td, logname)).read().replace(td, "<TMP>") == c["log"], (script, c["argv"])
    # scripts/hudson/hud.py: direct + grouped, stdout / stderr / the log file text
    open(os.path.join(td, "hudA.txt"), "w").write(g["hudA"])
    open(os.path.join(td, "hudB.txt"), "w").write(g["hudB"])
    for c in g["hud"]:
        r = run("hud.py", remap(c["argv"], td), td)
        assert r.returncode == c["rc"], (c["argv"], r.stderr)
        assert r.stdout == c["stdout"], (c["argv"], r.stdout, c["stdout"])
        assert r.stderr.replace(td, "<TMP>") == c["stderr"], (c["argv"], r.stderr, c["stderr"])
        assert open(os.path.join(td, "win8_fst.log")).read() == c["log"], c["argv"]
    for c in g["errors"]:
        script = c["argv"][0]
        r = run(script, remap(c["argv"], td), td)
        assert r.returncode == c["rc"] == 1
        norm = lambda s: s.replace(td, "<TMP>")
        want_out = c["stdout"]
        # the golden stdout of pica2's "File not found" holds the generator's temp path
        if "File not found" in want_out:
            assert norm(r.stdout) == want_out
        else:
            assert r.stdout == want_out
        assert norm(r.stderr) == c["stderr"]
    assert os.path.exists(os.path.join(td, "win8.log")) and os.path.exists(os.path.join(td, "win8_fst.log"))


def test_ehhgfa_cli(tmp_path):
    """scripts/ehhgfa.py writes the rows the reference's scripts/wip/ehhgfa.py wrote for the same
    matrix file (goldens captured by oracle/gen_golden.py), incl. the crash on an empty half."""
    g = load_golden("ehh.json")
    td = str(tmp_path)
    for k, c in enumerate(g["cli"]):
        f, o = os.path.join(td, f"hap{k}.txt"), os.path.join(td, f"out{k}.txt")
        open(f, "w").write(c["matrix_text"])
        r = run("ehhgfa.py", ["-i", f, "-p", str(c["p"]), "-w", str(c["w"]), "-refpos", str(c["refpos"]), "-o", o], td)
        assert r.returncode == c["rc"], (c["p"], c["w"], r.stderr)
        assert open(o).read() == c["out"], (c["p"], c["w"])
        if c["rc"]:
            assert r.stderr.strip().splitlines()[-1] == c["stderr_last"]


def test_pica2_cli_nontransitive_tables_per_hash_seed(tmp_path):
    """pica2.py on tables where "> threshold" is not transitive prints a value that depends on PYTHONHASHSEED
    (set.pop() order).  The drop-in, started under the same seed on the same .sim file, rebuilds the reader's
    set (same insertion order) and prints the same stdout and writes the same log, for every captured seed."""
    g = load_golden("pica2_seeded.json")
    td = str(tmp_path)
    n_runs, outcomes = 0, {}
    for t in g["tables"]:
        p = os.path.join(td, t["name"] + ".sim")
        open(p, "w").write(t["sim_text"])
        for run_ in t["runs"][:10]:
            for c in run_["cli"]:
                argv = [p, "-t", c["t"], "-l", str(c["l"]), "-d", td] + (["-r", str(c["r"])] if c["r"] is not None else [])
                r = run("pica2.py", argv, td, hashseed=run_["hashseed"])
                assert r.returncode == c["rc"], r.stderr
                assert r.stdout == c["stdout"], (t["name"], run_["hashseed"], c["t"], r.stdout, c["stdout"])
                assert open(os.path.join(td, t["name"] + ".log")).read().replace(td, "<TMP>") == c["log"], (t["name"], run_["hashseed"])
                outcomes.setdefault((t["name"], c["t"], c["r"]), set()).add(c["stdout"])
                n_runs += 1
    assert n_runs >= 80 and max(len(v) for v in outcomes.values()) >= 3  # the captured values really differ by seed
