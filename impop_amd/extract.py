"""Presence-matrix extractors: the data formats on the producer side of the
hot path.  They turn what `impg query -o gfa` / `odgi paths -H` emit (run_tajd.sh:126-140,
scripts/wip/op-afs.py:112) into the haplotype x site bit matrix the engine scans, which makes
the engine independent of `impg similarity` / `odgi similarity` and gives S without `povu`.

Host-side text parsing only; no statistics here.  `expand_bp=True` repeats every node column
`length` times so that a site is one bp column, the unit the identity kinds are defined on
(a node of length L then weighs L, exactly like a site weight w_s = L)."""
from __future__ import annotations

import re
from typing import Dict, List, Optional, Tuple

import numpy as np

from .matrixio import MatrixFile, from_dense


def _contig_of(ref_path_name: str) -> str:
    """The contig a matrix's coordinates refer to: the reference path's name without its ":start-end" range
    (`CHM13#0#chr7:1000-51000` -> `CHM13#0#chr7`) — what the batch driver matches a BED row's chromosome with
    (REGION = prefix + chr, run_pica2_impg.sh:139-151)."""
    import re
    return re.sub(r":\d+-\d+$", "", ref_path_name)


def _handle_to_matrix(lib, h, with_lengths: bool, ref_prefix: Optional[str]) -> MatrixFile:
    """names / bits / (lengths, positions) out of a native parser handle (impop_gfa_*)."""
    import ctypes as C

    from . import _lib
    n, n_seg, nb, ref_row = C.c_uint32(), C.c_uint64(), C.c_uint64(), C.c_int64()
    _lib.check(lib.impop_gfa_info(h, C.byref(n), C.byref(n_seg), C.byref(nb), C.byref(ref_row)))
    buf = C.create_string_buffer(max(nb.value, 1))
    _lib.check(lib.impop_gfa_names(h, buf))
    names = [x.decode("utf-8", "surrogateescape") for x in buf.raw[: nb.value].split(b"\0")[: n.value]]
    words = max((n_seg.value + 63) // 64, 1)
    bits = np.zeros((n.value, words), dtype=np.uint64)
    _lib.check(lib.impop_gfa_bits(h, bits.ctypes.data_as(C.POINTER(C.c_uint64)), words))
    mf = MatrixFile(bits=bits, n_site=int(n_seg.value), names=names)
    if with_lengths:
        lens = np.zeros(n_seg.value, dtype=np.uint32)
        _lib.check(lib.impop_gfa_lengths(h, lens.ctypes.data_as(C.POINTER(C.c_uint32))))
        mf.site_weight = lens
    if ref_prefix is not None:
        pos = np.zeros(n_seg.value, dtype=np.int64)
        _lib.check(lib.impop_gfa_positions(h, pos.ctypes.data_as(C.POINTER(C.c_int64))))
        mf.site_pos = pos
        mf.contig = _contig_of(names[ref_row.value]) if 0 <= ref_row.value < len(names) else (ref_prefix or "")
    return mf


def _from_paths_table_native(path: str) -> Optional[MatrixFile]:
    """The table through the native parser in libimpop_hip.so (impop_paths_table_parse: rows parsed by several threads
    straight into bit rows).  None on ANY non-zero status: the Python code below then handles the file and raises
    whatever it raises."""
    import ctypes as C
    import os

    from . import _lib
    lib = _lib.load()
    h = C.c_void_p()
    if lib.impop_paths_table_parse(os.fsencode(path), C.byref(h)) != 0:
        return None
    try:
        return _handle_to_matrix(lib, h, False, None)
    finally:
        lib.impop_gfa_free(h)


def from_paths_table(path: str, native: bool = True) -> MatrixFile:
    """`odgi paths -H`-style table: a header row, three metadata columns (path.name, path.length,
    node.count), then one column per node holding 0 / visit counts (op-afs.py:112 reads it the
    same way).  Presence = count != 0; one site per node.  native=True reads through
    impop_paths_table_parse (same matrix, two orders of magnitude faster); any file it declines comes here."""
    if native:
        mf = _from_paths_table_native(path)
        if mf is not None:
            return mf
    names: List[str] = []
    rows: List[np.ndarray] = []
    with open(path) as f:
        header = f.readline().rstrip("\n").split("\t")
        if len(header) < 4:
            raise ValueError(f"{path}: expected >= 4 tab-separated columns (3 metadata + nodes)")
        n_node = len(header) - 3
        for ln, line in enumerate(f, start=2):
            line = line.rstrip("\n")
            if not line:
                continue
            parts = line.split("\t")
            if len(parts) != n_node + 3:
                raise ValueError(f"{path}:{ln}: {len(parts)} fields, expected {n_node + 3}")
            names.append(parts[0])
            rows.append(np.array([p != "0" and p != "" for p in parts[3:]], dtype=np.uint8))
    mat = np.vstack(rows) if rows else np.zeros((0, n_node), np.uint8)
    order = sorted(range(len(names)), key=lambda i: names[i])  # engine convention: lexicographic name order
    return from_dense(mat[order], [names[i] for i in order])


_STEP = re.compile(r"([<>])([^<>]+)")


def _from_gfa_native(path: str, ref_prefix: Optional[str]) -> Optional[MatrixFile]:
    """Node-level extraction through the native parser in libimpop_hip.so (impop_gfa_parse: one mmap pass, bits packed
    directly — no dense byte matrix, no Python step lists).  None on ANY non-zero status: the Python code below then
    handles the file and raises whatever it raises."""
    import ctypes as C
    import os

    from . import _lib
    lib = _lib.load()
    h = C.c_void_p()
    if lib.impop_gfa_parse(os.fsencode(path), None if ref_prefix is None else ref_prefix.encode(), C.byref(h)) != 0:
        return None
    try:
        return _handle_to_matrix(lib, h, True, ref_prefix)
    finally:
        lib.impop_gfa_free(h)


def from_gfa(path: str, ref_prefix: Optional[str] = None, expand_bp: bool = True, native: bool = True) -> MatrixFile:
    """GFA 1.x: S (segments), P (paths, `id+,id-,...`) and W (walks, `>id<id...`) lines.
    Rows = paths/walks (W names become PanSN `sample#hap#seqid[:start-end]`), columns = segments in
    numeric-id order (odgi sort order), optionally expanded to bp columns.  If `ref_prefix` names a
    path (prefix match, e.g. 'CHM13#0#'), site_pos holds each column's coordinate on it (columns
    off the reference inherit the coordinate of the preceding reference column).
    Node-level extraction (expand_bp=False, the scalable form: node lengths become site weights) goes through the
    native parser when it accepts the file; this function is the definition and the fallback."""
    if not expand_bp and native:
        mf = _from_gfa_native(path, ref_prefix)
        if mf is not None:
            return mf
    seg_len: Dict[str, int] = {}
    paths: List[Tuple[str, List[str]]] = []
    ref_start = 0
    with open(path) as f:
        for line in f:
            if not line or line[0] not in "SPW":
                continue
            p = line.rstrip("\n").split("\t")
            if p[0] == "S":
                L = len(p[2]) if p[2] != "*" else 0
                for tag in p[3:]:
                    if tag.startswith("LN:i:"):
                        L = int(tag[5:])
                seg_len[p[1]] = L
            elif p[0] == "P":
                steps = [s[:-1] for s in p[2].split(",") if s]
                paths.append((p[1], steps))
            elif p[0] == "W":
                name = f"{p[1]}#{p[2]}#{p[3]}"
                if p[4] != "*" and p[5] != "*":
                    name += f":{p[4]}-{p[5]}"
                steps = [m.group(2) for m in _STEP.finditer(p[6])]
                paths.append((name, steps))
    def key(s):
        return (0, int(s)) if s.isdigit() else (1, s)
    segs = sorted(seg_len, key=key)
    col = {s: i for i, s in enumerate(segs)}
    paths.sort(key=lambda t: t[0])
    mat = np.zeros((len(paths), len(segs)), dtype=np.uint8)
    for r, (_, steps) in enumerate(paths):
        for s in steps:
            mat[r, col[s]] = 1
    lens = np.array([seg_len[s] for s in segs], dtype=np.int64)
    node_pos = None
    if ref_prefix is not None:
        ref = [t for t in paths if t[0].startswith(ref_prefix)]
        if not ref:
            raise ValueError(f"no path starts with {ref_prefix!r}")
        m = re.search(r":(\d+)-(\d+)$", ref[0][0])
        ref_start = int(m.group(1)) if m else 0
        node_pos = np.full(len(segs), -1, dtype=np.int64)
        off = ref_start
        for s in ref[0][1]:
            if node_pos[col[s]] < 0:
                node_pos[col[s]] = off
            off += seg_len[s]
        last = ref_start
        for i in range(len(segs)):  # off-reference columns inherit the preceding reference coordinate
            if node_pos[i] < 0:
                node_pos[i] = last
            else:
                last = node_pos[i]
        node_pos = np.maximum.accumulate(node_pos)
    names = [t[0] for t in paths]
    if expand_bp:
        reps = np.maximum(lens, 0)
        mat = np.repeat(mat, reps, axis=1)
        if node_pos is not None:
            # inside a reference node the coordinate advances per bp; off-reference columns stay put
            on_ref = np.zeros(len(segs), dtype=bool)
            for s in [t for t in paths if t[0].startswith(ref_prefix)][0][1]:
                on_ref[col[s]] = True
            pos = np.repeat(node_pos, reps)
            inner = np.concatenate([np.arange(r) if o else np.zeros(r, np.int64) for r, o in zip(reps, on_ref)]) if len(segs) else np.zeros(0, np.int64)
            node_pos = np.maximum.accumulate(pos + inner)
    mf = from_dense(mat, names)
    if not expand_bp:
        # one column per node, weighted by its length: BitMatrix.set_site_weights gives the records of the
        # expanded matrix without repeating columns (zero-length segments weigh nothing)
        mf.site_weight = np.maximum(lens, 0).astype(np.uint32)
    if node_pos is not None:
        mf.site_pos = node_pos
        mf.contig = _contig_of([t for t in paths if t[0].startswith(ref_prefix)][0][0])
    return mf
