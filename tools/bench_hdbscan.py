#!/usr/bin/env python3
"""The device HDBSCAN (posthoc.hdbscan_device) stage by stage -- core distances, Prim's tree, sklearn's tree code -- at several sizes.
  python3 tools/bench_hdbscan.py 100000 300000 1000000"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from idelucs_amd import posthoc
    for n in [int(v) for v in sys.argv[1:]] or [100000]:
        g = torch.Generator(device="cpu"); g.manual_seed(5)
        centres = torch.randn(8, 64, generator=g) * 3.0
        fam = torch.randint(0, 8, (n,), generator=g)
        x = (centres[fam] + torch.randn(n, 64, generator=g) * 0.6).to(torch.float32).double().numpy()
        stats = {}
        t0 = time.time()
        labels, prob = posthoc.hdbscan_device(x, n // 100 + 1, stats=stats)
        wall = time.time() - t0
        keep = {k: (round(v, 2) if isinstance(v, float) else v) for k, v in stats.items() if k in ("core_s", "prim_s", "tree_s", "missed", "prim_launches", "prim_stalls", "prim_censuses")}
        print(f"n = {n}: {wall:.1f} s  {keep}  clusters {len(np.unique(labels[labels >= 0]))}  noise {float((labels < 0).mean()):.3f}", flush=True)


if __name__ == "__main__":
    main()
