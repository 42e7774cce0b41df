"""CPU: sanitizer + fuzz target of the host-side text parsers (SURVEY §5 "build the C ABI with -fsanitize=address,undefined").

csrc/simparse.hip (the replacement of pica2.read_similarity_file, pica2.py:6-58, and h-fst.read_similarity_file,
h-fst.py:84-119) and csrc/gfaparse.hip (GFA / `odgi paths -H` readers) mmap and hand-parse untrusted text.  They hold no
device code, so they are compiled here AS C++ with g++ -fsanitize=address,undefined next to tests/fuzz/parsers_fuzz.cc and fed
corner cases (0-byte file, missing trailing newline, NUL bytes, truncated rows, duplicate headers, huge numeric fields,
10^6-column rows), every prefix of valid files and deterministic random mutations: every call must return IMPOP_OK or an
IMPOP_E_* code (on which the Python mirror takes over with the reference's messages) — never a sanitizer report.  The
reference behaviour that must survive (pica2.py:18-27,39-41,53-58; h-fst.py:90-98,105-109) is asserted on the SAME
sanitizer build through the library's own accessors in the driver, and at Python level by tests/test_sim_ingest.py."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "impop_amd", "csrc")


@pytest.fixture(scope="module")
def fuzz_binary(tmp_path_factory):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    d = tmp_path_factory.mktemp("parser_fuzz")
    flags = ["-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-D__HIP_PLATFORM_AMD__",
             "-I/opt/rocm/include", "-I" + CSRC]
    objs, procs = [], []
    for src, lang in ((os.path.join(CSRC, "simparse.hip"), ["-x", "c++"]), (os.path.join(CSRC, "gfaparse.hip"), ["-x", "c++"]),
                      (os.path.join(ROOT, "tests", "fuzz", "parsers_fuzz.cc"), [])):
        obj = str(d / (os.path.basename(src) + ".o"))
        objs.append(obj)
        procs.append(subprocess.Popen([gxx] + flags + ["-c"] + lang + [src, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate()
        assert p.returncode == 0, out
    exe = str(d / "parsers_fuzz")
    r = subprocess.run([gxx, "-fsanitize=address,undefined"] + objs + ["-o", exe, "-lpthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_host_parsers_survive_malformed_input_under_asan_ubsan(fuzz_binary):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([fuzz_binary, "12", "4000"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-6000:])
    assert "parsers_fuzz ok" in r.stdout and "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    # the driver saw both outcomes for every parser: accepted files and declined ones
    tail = r.stdout.strip().splitlines()[-1]
    import re
    nums = [int(x) for x in re.findall(r"(\d+)", tail)]
    assert nums[0] >= 4000 and all(v > 0 for v in nums[1:]), tail
