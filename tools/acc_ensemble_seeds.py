#!/usr/bin/env python3
"""Ensemble accuracy of the CLI run of Example/ALL_RESULTS.tsv:19 (Influenza-A, k=6, 5 clusters, 35 epochs x 5 voters) over several
seeds, voters batched (default) and one after the other; the imported reference's three ensembles score 0.934 / 0.928 / 0.994
(tests/golden/anchor_seeds.json).   python tools/acc_ensemble_seeds.py [--seeds 6]"""
import argparse
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pandas as pd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=6)
a = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from idelucs_amd.__main__ import main  # noqa: E402
import time  # noqa: E402
work = tempfile.mkdtemp(prefix="idelucs_acc_")
os.chdir(work)
for lanes in ("8", "1"):
    os.environ["IDELUCS_VOTER_LANES"] = lanes
    accs = []
    for seed in range(a.seeds):
        out = main(["--sequence_file", os.path.join(root, "tests/data/Influenza-A.fas"), "--GT_file", os.path.join(root, "tests/data/Influenza-A_GT.tsv"),
                    "--n_clusters", "5", "--n_epochs", "35", "--n_voters", "5", "--batch_sz", "512", "--k", "6", "--seed", str(seed)])
        accs.append(float(pd.read_csv(os.path.join(out, "metrics.tsv"), sep="\t", index_col=0).loc["ACC", "Value"]))
        time.sleep(1.1)
    print(f"\nvoters per batch {lanes}: ensemble ACC over seeds 0..{a.seeds - 1}: {[round(x, 4) for x in accs]}  mean {sum(accs) / len(accs):.4f}", flush=True)
