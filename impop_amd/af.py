"""GPU-backed mirror of the reference's scripts/af.py (af.py:7-68)."""
from __future__ import annotations

import csv

import numpy as np

from .runtime import default_context


def load_pairs(path):
    """af.load_pairs (af.py:7-19): names are cut at the first ':'."""
    rows = []
    samples = set()
    with open(path) as f:
        reader = csv.DictReader(f, delimiter="\t")
        for row in reader:
            a = row["group.a"].split(":", 1)[0]
            b = row["group.b"].split(":", 1)[0]
            val = float(row["estimated.identity"])
            rows.append((a, b, val))
            samples.add(a)
            samples.add(b)
    return rows, sorted(samples)


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
    """af.build_summary (af.py:46-54)"""
    total = sum(len(c) for c in clusters)
    summary = []
    for idx, members in enumerate(clusters, 1):
        count = len(members)
        freq = (count / total) if total else 0.0
        summary.append((f"c{idx}", count, freq, sorted(members)))
    return summary


def write_summary(summary, out_file):
    writer = csv.writer(out_file, delimiter="\t")
    writer.writerow(["cluster_id", "count", "frequency"])
    for cid, count, freq, _ in summary:
        writer.writerow([cid, count, f"{freq:.6f}"])


def write_details(summary, threshold, path):
    with open(path, "w", newline="") as fh:
        writer = csv.writer(fh, delimiter="\t")
        writer.writerow(["sample_id", "cluster_id", "threshold"])
        for cid, _, _, members in summary:
            for sample in members:
                writer.writerow([sample, cid, threshold])
