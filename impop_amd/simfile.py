"""Host-side `.sim` TSV ingest (pica2.py:6-58, h-fst.py:84-119, af.py:7-19) and the
densification the GPU kernels consume.  Pure marshalling: no statistics here."""
from __future__ import annotations

import csv
import os
import sys
from typing import Dict, Iterable, List, Sequence, Set, Tuple

import numpy as np

REQUIRED_COLS = {"group.a", "group.b", "estimated.identity"}


def read_similarity_file_pica2(filename):
    """pica2.read_similarity_file (pica2.py:6-58): same return value, same messages
    (stdout) and exit codes."""
    try:
        with open(filename, newline="") as handle:
            reader = csv.DictReader(handle, delimiter="\t")
            if reader.fieldnames is None:
                print(f"Error: File {filename} is empty or missing a header")
                sys.exit(1)
            missing_cols = REQUIRED_COLS - set(reader.fieldnames)
            if missing_cols:
                print(f"Error: File must contain columns: {sorted(REQUIRED_COLS)}")
                print(f"Found columns: {reader.fieldnames}")
                sys.exit(1)
            similarity_dict: Dict[Tuple[str, str], float] = {}
            elements: Set[str] = set()
            pair_count = 0
            for row_number, row in enumerate(reader, start=2):
                pair_count += 1
                e1, e2 = row["group.a"], row["group.b"]
                try:
                    similarity = float(row["estimated.identity"])
                except (TypeError, ValueError):
                    print(f"Error: Invalid similarity value on line {row_number}: {row['estimated.identity']}")
                    sys.exit(1)
                key = (e1, e2) if e1 <= e2 else (e2, e1)
                similarity_dict[key] = similarity
                elements.add(e1)
                elements.add(e2)
            if pair_count == 0:
                print(f"Warning: No similarity entries found in {filename}")
            return similarity_dict, elements, pair_count
    except FileNotFoundError:
        print(f"Error: File not found {filename}")
        sys.exit(1)
    except SystemExit:
        raise
    except Exception as e:  # pica2.py:56-58
        print(f"Error reading file {filename}: {e}")
        sys.exit(1)


def read_similarity_file_hfst(filename):
    """h-fst.read_similarity_file (h-fst.py:84-119): bad floats are warned and skipped,
    diagnostics go to stderr."""
    try:
        with open(filename, newline="") as f:
            reader = csv.DictReader(f, delimiter="\t")
            if not reader.fieldnames:
                print(f"Error: Empty file {filename}", file=sys.stderr)
                sys.exit(1)
            if not REQUIRED_COLS.issubset(set(reader.fieldnames)):
                print(f"Error: File must contain columns: {REQUIRED_COLS}", file=sys.stderr)
                print(f"Found: {reader.fieldnames}", file=sys.stderr)
                sys.exit(1)
            similarities: Dict[Tuple[str, str], float] = {}
            all_sequences: Set[str] = set()
            for row in reader:
                seq1, seq2 = row["group.a"], row["group.b"]
                try:
                    sim = float(row["estimated.identity"])
                except ValueError:
                    print(f"Warning: Invalid similarity value: {row['estimated.identity']}", file=sys.stderr)
                    continue
                key = (seq1, seq2) if seq1 <= seq2 else (seq2, seq1)
                similarities[key] = sim
                all_sequences.update([seq1, seq2])
            return similarities, all_sequences
    except FileNotFoundError:
        print(f"Error: File not found: {filename}", file=sys.stderr)
        sys.exit(1)


def densify(similarity_dict, names: Sequence[str]) -> np.ndarray:
    """dict[(min,max)] -> float  ==> dense symmetric [n,n] doubles, NaN = pair absent.
    `names` must be sorted (index order == lexicographic order)."""
    n = len(names)
    ix = {s: i for i, s in enumerate(names)}
    out = np.full((n, n), np.nan)
    for (a, b), v in similarity_dict.items():
        i, j = ix.get(a), ix.get(b)
        if i is None or j is None:
            continue
        out[i, j] = v
        out[j, i] = v
    return out


def read_dense(filename, flavor: str = "pica2"):
    """Fast ingest used by the drop-in CLIs: (sorted names, dense [n,n] identity with NaN for absent
    pairs, number of data rows).  Clean tab-separated files go through the native parser in
    libimpop_hip.so (impop_sim_parse); any file shape it declines (quotes, short rows, unusual
    number syntax, missing columns) — and every error path — goes through the reference-faithful
    Python readers above, so messages and exit codes are the reference's."""
    import ctypes as C

    from . import _lib
    lib = _lib.load()
    h = C.c_void_p()
    rc = lib.impop_sim_parse(os.fsencode(filename), 0 if flavor == "pica2" else 1, C.byref(h))
    if rc == 0:
        try:
            n, rows, nb, bad_line, n_bad = C.c_uint32(), C.c_uint64(), C.c_uint64(), C.c_int64(), C.c_uint64()
            _lib.check(lib.impop_sim_info(h, C.byref(n), C.byref(rows), C.byref(nb), C.byref(bad_line), C.byref(n_bad)))
            if bad_line.value < 0 and n_bad.value == 0:
                buf = C.create_string_buffer(max(nb.value, 1))
                _lib.check(lib.impop_sim_names(h, buf))
                names = [x.decode("utf-8", "surrogateescape") for x in buf.raw[: nb.value].split(b"\0")[: n.value]]
                dense = np.empty((n.value, n.value))
                _lib.check(lib.impop_sim_dense(h, dense.ctypes.data_as(C.POINTER(C.c_double))))
                if n.value or rows.value:
                    return names, dense, int(rows.value)
        finally:
            lib.impop_sim_free(h)
    # fall back: exact reference behaviour (including its messages / sys.exit) for everything else
    if flavor == "pica2":
        d, elements, pair_count = read_similarity_file_pica2(filename)
    else:
        d, elements = read_similarity_file_hfst(filename)
        pair_count = len(d)
    names = sorted(elements)
    return names, densify(d, names), pair_count
