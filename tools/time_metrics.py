#!/usr/bin/env python3
"""Wall time of each metric of posthoc.compute_results at cfg5's size (10^6 x 64 latent, 8 clusters):  python tools/time_metrics.py [--n 1000000]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1000000)
a = ap.parse_args()
import sklearn.metrics.cluster as metrics  # noqa: E402
from idelucs_amd import posthoc  # noqa: E402
from idelucs_amd.utils import cluster_acc  # noqa: E402
rng = np.random.default_rng(7)
centres = rng.normal(size=(8, 64)) * 3.0
truth = rng.integers(0, 8, a.n)
x = (centres[truth] + rng.normal(size=(a.n, 64)) * 0.6).astype(np.float32).astype(np.float64)
pred = (truth + 1) % 8 + 1
for name, fn in (("silhouette (device)", lambda: posthoc.silhouette_score_device(x, pred)),
                 ("silhouette (device) again", lambda: posthoc.silhouette_score_device(x, pred)),
                 ("davies_bouldin", lambda: metrics.davies_bouldin_score(x, pred)),
                 ("adjusted_mutual_info", lambda: metrics.adjusted_mutual_info_score(truth, pred)),
                 ("adjusted_rand", lambda: metrics.adjusted_rand_score(truth, pred)),
                 ("homogeneity", lambda: metrics.homogeneity_score(truth, pred)),
                 ("completeness", lambda: metrics.completeness_score(truth, pred)),
                 ("cluster_acc", lambda: cluster_acc(truth, pred))):
    t0 = time.perf_counter()
    v = fn()
    print(f"{name:28s} {time.perf_counter() - t0:8.2f} s   -> {v if not isinstance(v, tuple) else v[1]}", flush=True)
