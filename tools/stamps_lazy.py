#!/usr/bin/env python3
"""Where a step of the lazy Prim spends its time (a `make -C idelucs_amd/csrc STAMPS=1` build): tools/stamps_lazy.py [N]
Per kind of workgroup (workgroup 0 / one with an awake run / one that only keeps sleeping groups' bounds), the mean time between
the marks of lazy_step_kernel, and for workgroup 0 the idle time between two launches and the length of a launch (first entry to
last exit)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from idelucs_amd import _lib, posthoc
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
    rng = np.random.default_rng(7)
    centres = rng.normal(size=(8, 64)) * 3.0
    truth = rng.integers(0, 8, n)
    x = (centres[truth] + rng.normal(size=(n, 64)) * 0.6).astype(np.float32).astype(np.float64)
    out = (ctypes.c_ulonglong * 36)()
    _lib.check(_lib.lib.idl_debug_lazy_phases(ctypes.cast(out, ctypes.c_void_p)))
    stats = {}
    posthoc.hdbscan_device(x, n // 100 + 1, stats=stats)
    _lib.check(_lib.lib.idl_debug_lazy_phases(ctypes.cast(out, ctypes.c_void_p)))
    o = np.array(list(out), dtype=np.float64).reshape(3, 12)
    multi = False        # (the multi-node and fold variants of the lazy step left the library in round 6)
    names = ["run flags", "trip 2 answered", "decided", "node + boxes + ball", "bounds, queue", "exact distances", "candidate left"]
    fold = False
    if fold:
        names = ["trip 1", "trip 2 answered", "chain's candidates", "coordinates, how many go", "boxes, recorded, balls", "floor, bounds, exact, applied", "record left"]
    if multi:       # lazy_multi_kernel (kinds 0, 1) and lazy_reduce_kernel (kind 2) have marks of their own
        names = ["trip 1", "trip 2 answered", "nodes + boxes in LDS", "recorded, balls", "floor, bounds, exact, applied", "list + rest left", "-"]
        rnames = ["lists in LDS, bounds loaded", "block minima", "tournaments", "merged by rank", "umin", "distances, coordinates, sources", "decision written"]
    print(f"n = {n}: Prim {stats['prim_s']:.2f} s = {stats['prim_s'] / n * 1e6:.2f} us a step ({stats['prim_launches']} launches)")
    for kind, label in enumerate(["workgroup 0", "awake workgroups", "lazy_reduce_kernel" if multi else "ball keepers"]):
        m = max(o[kind][7], 1.0)
        nn = rnames if (multi and kind == 2) else names
        print(f"  {label:17s} ({int(o[kind][7])} samples), us: " + ", ".join(f"{nm} {o[kind][i] / m / 100:.2f}" for i, nm in enumerate(nn))
              + f"; sum {o[kind][:7].sum() / m / 100:.2f}; exact distances per step {o[kind][11] / m:.1f}"
              + (f"; inside the pairs' phase: floor tests + queued {o[kind][8] / m / 100:.2f}, bounds + exact {o[kind][9] / m / 100:.2f}, applied {o[kind][10] / m / 100:.2f}" if (fold and kind < 2) else ""))
    g = max(o[0][10], 1.0)
    print(f"  between launches (last exit -> workgroup 0's entry) {o[0][8] / g / 100:.2f} us; a launch, first entry -> last exit {o[0][9] / g / 100:.2f} us ({int(o[0][10])} samples)")


if __name__ == "__main__":
    main()
