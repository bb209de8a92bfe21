#!/usr/bin/env python3
"""Diagnostic: time vectorise2_kernel at cfg2 under the IDELUCS_DEV=vec_ablate=<bits> (1 no row stores, 2 no H0 count,
4 no deltas, 8 raw epilogue) to split the launch into its phases.  Not part of the product or the bench."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idelucs_amd import _lib, utils as U
from tools.bench_vectorise import synth_input, timeit

dev = torch.device("cuda")
n, L, k, P = 100000, 10000, int(os.environ.get("IDELUCS_ABLATE_K", "6")), 4
din = synth_input(n, L, dev)
specs = [t.spec() for t in U.mimic_transforms(P - 1)]
edits, edit_off = U._philox_edits(din, specs, 7)
out = torch.empty((P, n, 4 ** k), dtype=torch.float32, device=dev)
for ab in [int(x) for x in (sys.argv[1:] or "0 1 2 4 8 3 5 6 7 12 13 15".split())]:
    cur = dict(i.split("=", 1) for i in os.environ.get("IDELUCS_DEV", "").split(",") if "=" in i); cur["vec_ablate"] = str(ab)
    os.environ["IDELUCS_DEV"] = ",".join(f"{k_}={v_}" for k_, v_ in cur.items())
    mn, av = timeit(lambda: U._vectorise(din, k, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_FREQ_F32, P, edits, edit_off, out), 8)
    print(f"ablate={ab:2d}  min {mn:7.3f} ms  avg {av:7.3f} ms", flush=True)
