#!/usr/bin/env python3
"""Prim's stage of posthoc.hdbscan_device alone, on planted blobs or tight clusters in 64 dimensions, with the launch statistics of
the lazy form:   python tools/time_prim.py [--n 1000000] [--tight]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1000000)
    ap.add_argument("--clusters", type=int, default=8)
    ap.add_argument("--tight", action="store_true")
    a = ap.parse_args()
    from idelucs_amd import posthoc
    rng = np.random.default_rng(7)
    centres = rng.normal(size=(a.clusters, 64)) * (30.0 if a.tight else 3.0)
    truth = rng.integers(0, a.clusters, a.n)
    x = (centres[truth] + rng.normal(size=(a.n, 64)) * (0.05 if a.tight else 0.6)).astype(np.float32).astype(np.float64)
    k = a.n // 100 + 1
    stats = {}
    t0 = time.perf_counter()
    labels, prob = posthoc.hdbscan_device(x, k, stats=stats)
    t1 = time.perf_counter()
    e = stats.pop("mst_edges")
    print(f"n = {a.n}: total {t1 - t0:.1f} s;", {kk: (round(v, 2) if isinstance(v, float) else v) for kk, v in stats.items()},
          f"clusters {len(np.unique(labels[labels >= 0]))}; edge weight sum {e['distance'].sum():.9f}", flush=True)


if __name__ == "__main__":
    main()
