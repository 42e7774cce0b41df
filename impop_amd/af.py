"""GPU-backed mirror of the reference's scripts/af.py (af.py:7-68)."""
from __future__ import annotations

import csv

import numpy as np

from .runtime import default_context


def _sample_of(name: str) -> str:
    return name.split(":", 1)[0]  # the PanSN name without its `:start-end` range


def load_pairs(path):
    """-> ([(sample_a, sample_b, identity), ...] in file order, sorted sample names); like af.load_pairs
    (af.py:7-19) a missing column is a KeyError and a bad number a ValueError."""
    with open(path) as handle:
        triples = [(_sample_of(rec["group.a"]), _sample_of(rec["group.b"]), float(rec["estimated.identity"]))
                   for rec in csv.DictReader(handle, delimiter="\t")]
    return triples, sorted({t[0] for t in triples} | {t[1] for t in triples})


def cluster(rows, samples, threshold, ctx=None):
    """af.cluster (af.py:35-44): components of {identity >= threshold}, ordered by
    (-size, members).  The union-find is replaced by GPU label propagation."""
    ctx = ctx or default_context()
    samples = list(samples)
    n = len(samples)
    if n == 0:
        return []
    ix = {s: i for i, s in enumerate(samples)}
    # several rows may map to one pair after ':' truncation: any row >= threshold links,
    # i.e. the maximum decides
    dense = np.full((n, n), np.nan)
    for a, b, val in rows:
        i, j = ix[a], ix[b]
        if np.isnan(dense[i, j]) or val > dense[i, j]:
            dense[i, j] = dense[j, i] = val
    order = sorted(range(n), key=lambda i: samples[i])  # cluster order compares sorted member lists
    rank = {i: r for r, i in enumerate(order)}
    perm = np.array(order)
    cl, K, sizes = ctx.cluster_from_identity(dense[perm][:, perm], threshold)
    comps = [[] for _ in range(K)]
    for i in range(n):
        comps[int(cl[rank[i]])].append(samples[i])
    return comps


def build_summary(clusters):
    """-> [(cluster id c1.., size, frequency, sorted members)] in the given order (af.py:46-54)"""
    sizes = [len(c) for c in clusters]
    total = sum(sizes)
    return [(f"c{k}", size, (size / total if total else 0.0), sorted(members))
            for k, (members, size) in enumerate(zip(clusters, sizes), start=1)]


def write_summary(summary, out_file):
    """TSV `cluster_id count frequency` (frequency with 6 decimals), csv-module line endings like af.py:56-60"""
    csv.writer(out_file, delimiter="\t").writerows(
        [("cluster_id", "count", "frequency")] + [(cid, size, f"{freq:.6f}") for cid, size, freq, _ in summary])


def write_details(summary, threshold, path):
    """TSV `sample_id cluster_id threshold`, one row per sample (af.py:62-68)"""
    with open(path, "w", newline="") as handle:
        csv.writer(handle, delimiter="\t").writerows(
            [("sample_id", "cluster_id", "threshold")]
            + [(sample, cid, threshold) for cid, _, _, members in summary for sample in members])
