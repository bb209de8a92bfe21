#!/usr/bin/env python3
"""Core distances at BASELINE cfg5's size (10^6 x 64 latent rows, k = 10 001): the one-pass window kernels (csrc/knn.hip) timed
stage by stage, and (--matrix-rows R) R rows through the float64 matrix path for comparison.
  python3 tools/bench_knn.py [--n 1000000] [--matrix-rows 20000]"""
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
    ap.add_argument("--matrix-rows", type=int, default=20000)
    ap.add_argument("--families", type=int, default=8)
    a = ap.parse_args()
    from idelucs_amd import posthoc
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu"); g.manual_seed(5)
    centres = torch.randn(a.families, 64, generator=g) * 3.0
    fam = torch.randint(0, a.families, (a.n,), generator=g)
    x = (centres[fam] + torch.randn(a.n, 64, generator=g) * 0.6).to(torch.float32)
    xd = x.to(dev).double()
    k = a.n // 100 + 1
    for rep in range(2):
        stats = {}
        torch.cuda.synchronize(); t0 = time.time()
        core = torch.empty(a.n, dtype=torch.float64, device=dev)
        missed = posthoc._core_distances_window(xd, k, dev, core, stats=stats)
        torch.cuda.synchronize(); t1 = time.time()
        print(f"window path, n = {a.n}, k = {k}: {t1 - t0:.2f} s  {stats}", flush=True)
    if a.matrix_rows:
        rows = torch.randperm(a.n, generator=g)[:a.matrix_rows].to(dev)
        sq = (xd * xd).sum(1)
        ref = torch.zeros(a.n, dtype=torch.float64, device=dev)
        torch.cuda.synchronize(); t0 = time.time()
        posthoc._core_distances_rows(xd, sq, rows, k, dev, ref)
        torch.cuda.synchronize(); t1 = time.time()
        done = torch.ones(a.n, dtype=torch.bool, device=dev); done[missed] = False
        sel = rows[done[rows]]
        same = bool((ref[sel] == core[sel]).all())
        print(f"matrix path, {a.matrix_rows} rows: {t1 - t0:.2f} s (=> {(t1 - t0) * a.n / a.matrix_rows:.1f} s for all rows); equal on those rows: {same}", flush=True)


if __name__ == "__main__":
    main()
