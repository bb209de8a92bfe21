#!/usr/bin/env python3
"""Diagnostic: where and when the workgroups of the optimizer launch (dW1 tiles + the rest) run.
   IDELUCS_DEV=stamps=1 python tools/stamps_step.py"""
import os, sys, ctypes
os.environ["IDELUCS_DEV"] = ",".join(x for x in (os.environ.get("IDELUCS_DEV", ""), "stamps=1") if x)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from idelucs_amd import _lib, utils as U, models
from idelucs_amd.PytorchUtils import NetLinear
from idelucs_amd.fused import FusedLinearTrainer
dev = torch.device("cuda:0")
P, n, F, C, B = 4, 6000, 4096, 20, 512
g = torch.Generator(device=dev); g.manual_seed(1)
feats = torch.rand((P, n, F), device=dev, generator=g) * 2e-4 + 1e-4
mean, scale = U.col_stats(feats[0])
store = U.FeatureStore(None, None, feats, mean, scale, 6, False)
net = NetLinear(F, C).to(dev); net.apply(models.weights_init)
tr = FusedLinearTrainer(net, 1e-3, 0.25, 2.8, seed=3)
import time
t_end = time.time() + float(os.environ.get("WARM_SECONDS", "2.5"))     # a burst from idle runs at a lower clock (round 4: 2.17 GHz in the
while time.time() < t_end:                                             # first dozen steps, 2.41 GHz sustained -- the same 77 k cycles
    tr.run_epoch(store, B, use_graph=False)                            # per tile workgroup, 35.5 us against 32.1): stamp the steady state
torch.cuda.synchronize()
out = np.zeros((1024, 4), np.uint64)
_lib.check(_lib.lib.idl_debug_stamps(out.ctypes.data_as(ctypes.c_void_p)))
used = out[:, 1] > 0
t0 = out[used, 0].min()
s = (out[:, 0].astype(np.int64) - int(t0)) * 0.01; e = (out[:, 1].astype(np.int64) - int(t0)) * 0.01     # us
hw = out[:, 2]; cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1; xcc = out[:, 3] & 0xF; cyc = (out[:, 3] >> 8).astype(np.float64)
nb = int(used.sum())
print(f"{nb} workgroups; launch spans {e[used].max():.2f} us")
for lo, hi, name in ((0, 256, "dW1 tiles"), (256, nb, "the rest")):
    if hi > lo:
        print(f"{name:10s}: start {s[lo:hi].min():6.2f} .. {s[lo:hi].max():6.2f}   end {e[lo:hi].min():6.2f} .. {e[lo:hi].max():6.2f}   "
              f"duration {np.min(e[lo:hi] - s[lo:hi]):6.2f} .. {np.max(e[lo:hi] - s[lo:hi]):6.2f} (median {np.median(e[lo:hi] - s[lo:hi]):.2f})")
d = e - s
clk = cyc[:nb] / np.maximum((out[:nb, 1] - out[:nb, 0]).astype(np.float64), 1.0) * 0.1       # GHz: shader cycles per 10 ns tick
print(f"clock held while a workgroup ran (s_memtime / s_memrealtime): dW1 tiles median {np.median(clk[:256]):.3f} GHz "
      f"({clk[:256].min():.3f} .. {clk[:256].max():.3f}); shader cycles per tile workgroup median {np.median(cyc[:256]):.0f}")
if nb > 256 + 128:
    print(f"dW2 tiles (256..383): duration {d[256:384].min():.2f} .. {d[256:384].max():.2f} median {np.median(d[256:384]):.2f}")
    print("the small tensors / loss (384..): " + " ".join(f"{d[b]:.1f}" for b in range(384, nb)))
place = {}
for b in range(nb):
    place.setdefault((int(xcc[b]), int(se[b]), int(sh[b]), int(cu[b])), []).append(b)
tiles_per_cu = [sum(1 for b in v if b < 256) for v in place.values()]
print(f"{len(place)} distinct (xcc, se, sh, cu); dW1 tiles per CU: " + ", ".join(f"{k}: {tiles_per_cu.count(k)} CUs" for k in sorted(set(tiles_per_cu))))
shared = {k: v for k, v in place.items() if any(b >= 256 for b in v)}
alone = [d[v[0]] for k, v in place.items() if len(v) == 1 and v[0] < 256]
withx = [d[b] for k, v in shared.items() for b in v if b < 256]
print(f"dW1 tile alone on its CU: n={len(alone)} median {np.median(alone) if alone else 0:.2f} max {max(alone) if alone else 0:.2f};  beside other workgroups: n={len(withx)} "
      f"median {np.median(withx):.2f} max {max(withx):.2f}")
for b in list(range(0, 8)) + list(range(256, min(nb, 272))) + list(range(max(256, nb - 6), nb)):
    print(f"  wg {b:4d}  xcc {xcc[b]} se {se[b]} sh {sh[b]} cu {cu[b]:2d}  start {s[b]:6.2f}  end {e[b]:6.2f}")
