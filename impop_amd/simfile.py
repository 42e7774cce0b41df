"""Host-side `.sim` TSV ingest (pica2.py:6-58, h-fst.py:84-119, af.py:7-19) and the
densification the GPU kernels consume.  Pure marshalling: no statistics here."""
from __future__ import annotations

import csv
import os
import sys
from typing import Dict, Sequence, Set, Tuple

import numpy as np

REQUIRED_COLS = {"group.a", "group.b", "estimated.identity"}


class _Flavor:
    """How one of the reference's two `.sim` readers reports problems.  The message texts, the stream
    they go to and which conditions are fatal are the scripts' observable interface (the bash drivers
    and the CLI goldens see them); the reading loop itself is shared."""

    def __init__(self, stream, empty_msg, empty_when_falsy, cols_msgs, bad_value_msg, bad_value_fatal, bad_value_errors,
                 not_found_msg, wrap_other_errors, no_rows_msg):
        self.stream = stream                      # None = stdout
        self.empty_msg = empty_msg
        self.empty_when_falsy = empty_when_falsy  # h-fst treats [] like None; pica2 only None
        self.cols_msgs = cols_msgs
        self.bad_value_msg = bad_value_msg
        self.bad_value_fatal = bad_value_fatal
        self.bad_value_errors = bad_value_errors
        self.not_found_msg = not_found_msg
        self.wrap_other_errors = wrap_other_errors
        self.no_rows_msg = no_rows_msg

    def say(self, text):
        print(text, file=self.stream if self.stream is not None else sys.stdout)

    def die(self, *texts):
        for t in texts:
            self.say(t)
        sys.exit(1)


def _flavor(kind: str) -> _Flavor:
    if kind == "pica2":  # pica2.py:6-58: everything on stdout, a bad number is fatal, any other exception is reported
        return _Flavor(None, "Error: File {f} is empty or missing a header", False,
                       lambda found: (f"Error: File must contain columns: {sorted(REQUIRED_COLS)}", f"Found columns: {found}"),
                       "Error: Invalid similarity value on line {line}: {val}", True, (TypeError, ValueError),
                       "Error: File not found {f}", "Error reading file {f}: {e}", "Warning: No similarity entries found in {f}")
    # h-fst.py:84-119 (and hud.py:18-53): stderr, a bad number is skipped with a warning
    return _Flavor(sys.stderr, "Error: Empty file {f}", True,
                   lambda found: (f"Error: File must contain columns: {REQUIRED_COLS}", f"Found: {found}"),
                   "Warning: Invalid similarity value: {val}", False, (ValueError,),
                   "Error: File not found: {f}", None, None)


def _read_table(filename, fl: _Flavor):
    """-> (dict keyed by the name pair in string order, set of names, number of data rows read)"""
    table: Dict[Tuple[str, str], float] = {}
    seen: Set[str] = set()
    n_rows = 0
    try:
        with open(filename, newline="") as handle:
            rows = csv.DictReader(handle, delimiter="\t")
            header = rows.fieldnames
            if header is None or (fl.empty_when_falsy and not header):
                fl.die(fl.empty_msg.format(f=filename))
            if not REQUIRED_COLS <= set(header):
                fl.die(*fl.cols_msgs(header))
            for line_no, rec in enumerate(rows, start=2):  # line 1 is the header
                n_rows += 1
                first, second, text = rec["group.a"], rec["group.b"], rec["estimated.identity"]
                try:
                    value = float(text)
                except fl.bad_value_errors:
                    msg = fl.bad_value_msg.format(line=line_no, val=text)
                    if fl.bad_value_fatal:
                        fl.die(msg)
                    fl.say(msg)
                    continue
                table[(first, second) if first <= second else (second, first)] = value  # later rows overwrite
                seen.update((first, second))
    except FileNotFoundError:
        fl.die(fl.not_found_msg.format(f=filename))
    except SystemExit:
        raise
    except Exception as e:
        if fl.wrap_other_errors is None:
            raise
        fl.die(fl.wrap_other_errors.format(f=filename, e=e))
    if n_rows == 0 and fl.no_rows_msg:
        fl.say(fl.no_rows_msg.format(f=filename))
    return table, seen, n_rows


def read_similarity_file_pica2(filename):
    """Behaves like pica2.read_similarity_file (pica2.py:6-58) -> (dict, set of names, row count)."""
    return _read_table(filename, _flavor("pica2"))


def read_similarity_file_hfst(filename):
    """Behaves like h-fst.read_similarity_file (h-fst.py:84-119) -> (dict, set of names)."""
    table, seen, _ = _read_table(filename, _flavor("hfst"))
    return table, seen


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


def read_dense(filename, flavor: str = "pica2", with_elements: bool = False):
    """Fast ingest used by the drop-in CLIs: (sorted names, dense [n,n] identity with NaN for absent
    pairs, number of data rows[, elements]).  `elements` (with_elements=True) is the set of names built the
    way the reference's reader builds it — one add() per distinct name in file order (pica2.py:45-46) — so
    that its iteration order, which seeds pica2's greedy grouping, is the reference's.  Clean tab-separated files go through the native parser in
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
                    if not with_elements:
                        return names, dense, int(rows.value)
                    seen = np.zeros(max(n.value, 1), dtype=np.uint32)
                    _lib.check(lib.impop_sim_first_seen(h, seen.ctypes.data_as(C.POINTER(C.c_uint32))))
                    elements = set()
                    for k in seen[: n.value]:
                        elements.add(names[int(k)])
                    return names, dense, int(rows.value), elements
        finally:
            lib.impop_sim_free(h)
    # fall back: exact reference behaviour (including its messages / sys.exit) for everything else
    if flavor == "pica2":
        d, elements, pair_count = read_similarity_file_pica2(filename)
    else:
        d, elements = read_similarity_file_hfst(filename)
        pair_count = len(d)
    names = sorted(elements)
    if with_elements:
        return names, densify(d, names), pair_count, elements
    return names, densify(d, names), pair_count
