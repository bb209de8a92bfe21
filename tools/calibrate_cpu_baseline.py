#!/usr/bin/env python3
"""SURVEY 8(d) calibration of bench.py's `cpu_baseline` ("port") against the REAL reference, in this
container only (the reference cannot travel to the GPU box).  Same synthetic sample for both:
n x 10 kbp uniform ACGT, k=6, n_mimics=3, B=512, NetLinear C=20, one epoch.

  reference : idelucs.models.IID_model(...).build_dataloader() + contrastive_training_epoch()
              imported from the /tmp build made by tests/golden/make_golden.py
  port      : bench.cpu_baseline (oracle C vectorise + torch-CPU epoch)

  python tools/calibrate_cpu_baseline.py [--n 2000]      -> prints both seq/s and the ratio
"""
import argparse, os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=2000)
ap.add_argument("--len", type=int, default=10000)
ap.add_argument("--ref-build", default="/tmp/idelucs_ref_build")
a = ap.parse_args()

path = f"/tmp/calib_{a.n}x{a.len}.fas"
rng = np.random.default_rng(12345)
with open(path, "wb") as f:
    for i in range(a.n):
        f.write(b">seq%06d\n" % i)
        f.write(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=a.len).tobytes()); f.write(b"\n")

import torch
cores = os.cpu_count()

# ---- the port, exactly as bench.py runs it
import bench
ns = types.SimpleNamespace(cpu_sample=a.n, n=a.n, len=a.len, k=6, n_mimics=3, n_clusters=20, batch_sz=512, cpu_ref_steps=10 ** 9)
port = bench.cpu_baseline(ns)
print("port     :", port["value"], "seq/s;", port["sample"], flush=True)

# ---- the reference (CLI thread setting: cpu_count()-2, idelucs/__main__.py:316)
sys.path.insert(0, a.ref_build)
torch.set_num_threads(max(1, cores - 2))
from idelucs import models as RM
args = {'sequence_file': path, 'GT_file': None, 'n_clusters': 20, 'k': 6, 'model_size': 'linear', 'n_mimics': 3,
        'batch_sz': 512, 'optimizer': 'RMSprop', 'lambda': 2.8, 'lr': 1e-3, 'weight': 0.25, 'scheduler': None,
        'n_epochs': 1, 'n_voters': 1, 'noise': 0}
t0 = time.perf_counter()
m = RM.IID_model(args)
m.build_dataloader()
t1 = time.perf_counter()
m.contrastive_training_epoch()
t2 = time.perf_counter()
ref = a.n / (t2 - t0)
print(f"reference: {ref:.1f} seq/s; vectorise+augment {t1 - t0:.2f} s + epoch {t2 - t1:.2f} s on {cores} cores "
      f"({max(1, cores - 2)} torch threads, 4 loader workers)")
ratio = port['value_reference_threads'] / ref
print(f"ratio port/reference = {ratio:.2f} (both with {max(1, cores - 2)} torch threads)")
import json
out = {"ratio": round(ratio, 3), "port_seq_per_s": port['value_reference_threads'], "reference_seq_per_s": ref, "cores": cores,
       "torch_threads": max(1, cores - 2), "sample": f"{a.n} x {a.len} bp, k=6, n_mimics=3, B=512, C=20, one epoch",
       "reference": f"vectorise+augment {t1 - t0:.2f} s + epoch {t2 - t1:.2f} s (4 loader workers)", "port": port["sample"]}
with open(os.path.join(ROOT, "profiles", "r03_cpu_calibration.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps(out))
