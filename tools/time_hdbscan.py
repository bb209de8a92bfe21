#!/usr/bin/env python3
"""Wall time of posthoc.hdbscan_device (exact HDBSCAN with its O(N^2) stages on the GPU) on planted blobs in 64 dimensions:
  python tools/time_hdbscan.py [--n 200000]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200000)
    ap.add_argument("--clusters", type=int, default=12)
    a = ap.parse_args()
    from idelucs_amd import posthoc
    from sklearn.metrics import adjusted_rand_score
    rng = np.random.default_rng(7)
    centres = rng.normal(size=(a.clusters, 64)) * 3.0
    truth = rng.integers(0, a.clusters, a.n)
    x = (centres[truth] + rng.normal(size=(a.n, 64)) * 0.6).astype(np.float32).astype(np.float64)
    dev = torch.device("cuda:0")
    k = a.n // 100 + 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    core = posthoc.core_distances_device(torch.from_numpy(x).to(dev), k, dev)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"n = {a.n}, k = {k}: core distances {t1 - t0:.1f} s", flush=True)
    del core
    t0 = time.perf_counter()
    labels, prob = posthoc.hdbscan_device(x, k)
    t2 = time.perf_counter()
    keep = labels >= 0
    print(f"hdbscan_device total {t2 - t0:.1f} s; {len(np.unique(labels[keep]))} clusters, {100 * (1 - keep.mean()):.2f} % noise, "
          f"ARI on the clustered points {adjusted_rand_score(truth[keep], labels[keep]):.4f}", flush=True)


if __name__ == "__main__":
    main()
