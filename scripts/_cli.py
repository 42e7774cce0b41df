"""Shared pieces of the drop-in CLIs: table-driven flag definitions and the `<basename><suffix>.log`
convention of the reference's scripts (pica2.py:197-199, h-fst.py:324-326)."""
import argparse
import os


def make_parser(summary, flags):
    """flags: (names..., dict of argparse keywords) tuples; keeps every script's flag set in one table."""
    ap = argparse.ArgumentParser(description=summary)
    for *names, kw in flags:
        ap.add_argument(*names, **kw)
    return ap


def log_path_for(input_path, log_dir, suffix=""):
    stem = os.path.splitext(os.path.basename(input_path))[0]
    os.makedirs(log_dir, exist_ok=True)
    return os.path.join(log_dir, stem + suffix + ".log")
