#!/usr/bin/env python3
"""End-to-end (FASTA text in the page cache -> last optimizer.step of epoch 1) timing: SURVEY 8(d)'s T_e2e.
Generates the synthetic FASTA of BASELINE.md section 3 (seed 12345, >seq%06d, one line per sequence) under /tmp.
  python tools/bench_e2e.py [--n 100000] [--len 10000]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from idelucs_amd import models, utils as U, gemm_tuning

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100000); ap.add_argument("--len", type=int, default=10000)
a = ap.parse_args()
path = f"/tmp/synth_{a.n}x{a.len}.fas"
if not os.path.exists(path):
    rng = np.random.default_rng(12345)
    t0 = time.time()
    with open(path, "wb") as f:
        for i0 in range(0, a.n, 2000):
            blk = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=(min(2000, a.n - i0), a.len))
            for j in range(blk.shape[0]):
                f.write(b">seq%06d\n" % (i0 + j)); f.write(blk[j].tobytes()); f.write(b"\n")
    print(f"generated {path}: {os.path.getsize(path) / 1e9:.2f} GB in {time.time() - t0:.1f} s", flush=True)
os.environ.setdefault("IDELUCS_TUNABLEOP", "1")
gemm_tuning.maybe_enable()
args = {'sequence_file': path, 'GT_file': None, 'n_clusters': 20, 'k': 6, 'model_size': 'linear', 'n_mimics': 3, 'batch_sz': 512,
        'optimizer': 'RMSprop', 'lambda': 2.8, 'lr': 1e-3, 'weight': 0.25, 'scheduler': None, 'n_epochs': 1, 'n_voters': 1}
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ff = U.FastaFile(path); t1 = time.perf_counter()
    m = models.IID_model(args)
    m.store = U.build_feature_store(path, 3, k=6, fasta=ff, device=m.device)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    m.contrastive_training_epoch()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"rep {rep}: parse+pack {t1 - t0:.3f} s ({os.path.getsize(path) / (t1 - t0) / 1e9:.2f} GB/s), H2D+vectorise+scaler {t2 - t1:.3f} s, "
          f"epoch {t3 - t2:.3f} s, T_e2e {t3 - t0:.3f} s -> {a.n / (t3 - t0):,.0f} sequences/s end to end", flush=True)
