#!/usr/bin/env python3
"""Diagnostic: phase marks inside the four middle kernels of the training step (workgroups 0..63), in microseconds from the
kernel's first workgroup start.   IDELUCS_DEV=stamps=1 python tools/stamps_mid.py"""
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
names = {1: ["start", "loads in, relu/dropout done (wave 0)", "MFMA + partials written", "barrier 1", "row sums, f/r2 stored, R2t (2 barriers)",
             "logits MFMA + barrier", "end"],
         2: ["start", "prologue requests issued (wave 0)", "barrier: sP / sW3 staged", "head backward of the row", "barrier",
             "dr1 MFMA + stores issued", "end"],
         3: ["start", "tiles' products + exponentials (wave 0)", "barrier: the four waves' sums", "partial row sums stored = end"],
         4: ["start", "partial row sums read, logarithms in LDS (barrier)", "both products of its tiles (wave 0)", "barrier: the four waves' G tiles",
             "G_part stored = end"]}
kern = {1: "mid_fwd", 2: "mid_bwd", 3: "nce_pass1 (workgroups (x < 64, y = 0))", 4: "nce_pass2 (workgroups (x < 64, y = 0))"}
for mode in (1, 2, 3, 4):
    _lib.check(_lib.lib.idl_debug_phase_stamps(mode))
    tr.run_epoch(store, B, use_graph=False)
    torch.cuda.synchronize()
    out = np.zeros((1024, 4), np.uint64)
    _lib.check(_lib.lib.idl_debug_stamps(out.ctypes.data_as(ctypes.c_void_p)))
    st = out.reshape(-1)[:64 * 8].reshape(64, 8).astype(np.int64)
    st = st[np.abs(st[:, 0] - np.median(st[:, 0])) < 100000]      # (the epoch's last, partial batch has fewer workgroups: the others' marks are one step older)
    t0 = st[:, 0].min()
    us = (st - t0) * 0.01
    print(f"== {kern[mode]}: marks (median / max over the workgroups of the last full step, us after the first workgroup's start)")
    for i, nm in enumerate(names[mode]):
        print(f"   {i} {nm:55s} {np.median(us[:, i]):6.2f} {us[:, i].max():6.2f}")
_lib.check(_lib.lib.idl_debug_phase_stamps(0))
