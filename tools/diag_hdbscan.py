#!/usr/bin/env python3
"""The device HDBSCAN on a saved latent (tools/run_cfg5_cli.py --save-latent): what the points look like, stage timings, and
(--compare) the matrix / unfiltered paths next to the default ones.
  python3 tools/diag_hdbscan.py /dev/shm/latent.npy [--compare]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from idelucs_amd import posthoc
    x = np.load(sys.argv[1]).astype(np.float64)
    n = len(x)
    k = n // 100 + 1
    norms = np.linalg.norm(x - x.mean(0), axis=1)
    rng = np.random.default_rng(0)
    a, b = rng.integers(0, n, 200000), rng.integers(0, n, 200000)
    dd = np.linalg.norm(x[a] - x[b], axis=1)
    print(f"n = {n}, k = {k}; centred norms: median {np.median(norms):.4g}, max {norms.max():.4g}; random pair distances: "
          f"1 % {np.quantile(dd, 0.01):.4g}, 10 % {np.quantile(dd, 0.1):.4g}, median {np.median(dd):.4g}; singular values (top 8 of 64): "
          f"{np.round(np.linalg.svd(x[rng.integers(0, n, 20000)] - x.mean(0), compute_uv=False)[:8], 2)}", flush=True)
    runs = [("default", {})]
    if "--compare" in sys.argv:
        runs += [("IDELUCS_DEV=mst_filter=0", {"IDELUCS_DEV": "mst_filter=0"}), ("IDELUCS_DEV=knn=matrix", {"IDELUCS_DEV": "knn=matrix"})]
    ref = None
    for name, env in runs:
        os.environ.update(env)
        stats = {}
        t0 = time.time()
        labels, prob = posthoc.hdbscan_device(x, k, stats=stats)
        wall = time.time() - t0
        edges = stats.pop("mst_edges")
        for kk in env:
            del os.environ[kk]
        print(f"{name}: {wall:.1f} s  " + str({kk: (round(v, 2) if isinstance(v, float) else v) for kk, v in stats.items()}) +
              f"  clusters {len(np.unique(labels[labels >= 0]))} noise {float((labels < 0).mean()):.4f} core[:3] {stats and ''}", flush=True)
        if ref is None:
            ref = (labels, prob, edges)
        else:
            same_edges = all(np.array_equal(ref[2][f], edges[f]) for f in ("current_node", "next_node", "distance"))
            print(f"    labels equal {np.array_equal(ref[0], labels)}, probabilities equal {np.array_equal(ref[1], prob)}, tree edges equal {same_edges}", flush=True)


if __name__ == "__main__":
    main()
