#!/usr/bin/env python3
"""Where a Prim step spends its time (a `make -C idelucs_amd/csrc STAMPS=1` build): tools/stamps_prim.py latent.npy | N (blobs)."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from idelucs_amd import _lib, posthoc
    arg = sys.argv[1]
    if arg.endswith(".npy"):
        x = np.load(arg).astype(np.float64)
    else:
        n = int(arg)
        g = torch.Generator(device="cpu"); g.manual_seed(5)
        centres = torch.randn(8, 64, generator=g) * 3.0
        fam = torch.randint(0, 8, (n,), generator=g)
        x = (centres[fam] + torch.randn(n, 64, generator=g) * 0.6).to(torch.float32).double().numpy()
    n = len(x)
    for env in ({}, {"IDELUCS_DEV": "mst_filter=0"}):
        os.environ.update(env)
        out = (ctypes.c_ulonglong * 8)()
        _lib.check(_lib.lib.idl_debug_prim_phases(ctypes.cast(out, ctypes.c_void_p)))
        stats = {}
        posthoc.hdbscan_device(x, n // 100 + 1, stats=stats)
        _lib.check(_lib.lib.idl_debug_prim_phases(ctypes.cast(out, ctypes.c_void_p)))
        for k in env:
            del os.environ[k]
        m = max(out[6], 1)
        names = ["candidates reduced", "added node + boxes in LDS", "bounds evaluated", "exact distances", "candidate left"]
        print(f"{env or 'default'}: n = {n}, Prim {stats['prim_s']:.2f} s = {stats['prim_s'] / n * 1e6:.1f} us / step; sampled workgroups, us per step: " +
              ", ".join(f"{nm} {out[i] / m / 100:.2f}" for i, nm in enumerate(names)) + f"; exact distances per workgroup and step {out[5] / m:.1f}", flush=True)


if __name__ == "__main__":
    main()
