#!/usr/bin/env python3
"""Generate tests/golden/anchor_seeds.json from the REAL reference (build container only; same scratch build as make_golden.py):
the distribution of the reference's own clustering accuracy over seeds, which the statistical end-to-end test of the MI355X path
is held to (the bar must come from the reference, not from this repo's own sweeps).

  single : Influenza-A, k=6, n_clusters=5, 10 epochs, n_mimics=3, batch 512, weight 0.25, ONE voter (BASELINE cfg1's parameters on
           the data file that is present), seeds 0..9.  Seed s = a fresh process that imports the reference (which seeds torch /
           numpy / random with 0 at import, models.py:17-21) and then re-seeds all three with s before building the model.
  voters5: the same data with the reference CLI's defaults for Example/ALL_RESULTS.tsv:19 (35 epochs, 5 voters, ensemble by
           utils.label_features), driven through idelucs.models / idelucs.utils exactly as idelucs/__main__.py:106-151 does
           (the CLI module itself needs `hdbscan`, which is absent here), seeds 0..2.

Usage:  python tests/golden/make_anchor_seeds.py      (~15 min on 8 cores)
        python tests/golden/make_anchor_seeds.py --extend-single 40      (round 3: seeds 10..39 of `single` appended)
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import build_reference, SCRATCH, DATA   # noqa: E402

SINGLE = """
import sys, json, random; sys.path.insert(0, {scratch!r})
import numpy as np, pandas as pd, torch
from idelucs.cluster import iDeLUCS_cluster
from idelucs.utils import cluster_acc
seed = {seed}
torch.manual_seed(seed); np.random.seed(seed); random.seed(seed)
m = iDeLUCS_cluster({fas!r}, n_clusters=5, n_epochs=10, n_mimics=3, batch_sz=512, k=6, weight=0.25, n_voters=1)
y, lat = m.fit_predict(None)
df = pd.read_csv({gt!r}, sep='\\t')
u = {{v: i for i, v in enumerate(sorted(set(df.cluster_id)))}}
gt = np.array([u[v] for v in df.cluster_id])
print('RESULT', json.dumps({{'seed': seed, 'acc': float(cluster_acc(gt, y)[1])}}))
"""

VOTERS5 = """
import sys, json, random; sys.path.insert(0, {scratch!r})
import numpy as np, pandas as pd, torch
from idelucs import models
from idelucs.models import IID_model, weights_init
from idelucs.utils import SummaryFasta, label_features, cluster_acc
seed = {seed}
torch.manual_seed(seed); np.random.seed(seed); random.seed(seed)
args = dict(sequence_file={fas!r}, GT_file=None, n_clusters=5, n_epochs=35, n_mimics=3, batch_sz=512, k=6, optimizer='RMSprop',
            scheduler='None', weight=0.25, lr=1e-3, n_voters=5, model_size='linear')
args['lambda'] = 2.8
model = IID_model(args)
model.names, model.lengths, model.GT, model.cluster_dis = SummaryFasta(model.sequence_file, None)
model.build_dataloader()
predictions, accs = [], []
df = pd.read_csv({gt!r}, sep='\\t')
u = {{v: i for i, v in enumerate(sorted(set(df.cluster_id)))}}
gt = np.array([u[v] for v in df.cluster_id])
for voter in range(args['n_voters']):                     # idelucs/__main__.py:106-146
    model.net.apply(weights_init)
    model.epoch = 0
    for i in range(args['n_epochs']):
        model.contrastive_training_epoch()
    y_pred, probabilities, latent = model.predict()
    accs.append(float(cluster_acc(gt, y_pred)[1]))
    y_pred = y_pred.astype(np.int32)
    d, count = {{}}, 0
    for i in range(y_pred.shape[0]):
        if y_pred[i] in d:
            y_pred[i] = d[y_pred[i]]
        else:
            d[y_pred[i]] = count; y_pred[i] = count; count += 1
    predictions.append(y_pred)
y, prob = label_features(np.array(predictions), args['n_clusters'])
print('RESULT', json.dumps({{'seed': seed, 'acc_ensemble': float(cluster_acc(gt, y)[1]), 'acc_voters': accs}}))
"""


def run(code):
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp")
    line = [l for l in r.stdout.splitlines() if "RESULT" in l]
    if not line:
        raise RuntimeError(r.stderr[-3000:])
    return json.loads(line[0].split("RESULT", 1)[1])


def main():
    build_reference()
    fas, gt = os.path.join(DATA, "Influenza-A.fas"), os.path.join(DATA, "Influenza-A_GT.tsv")
    out = {"single": [], "voters5": [],
           "doc": "reference (Kari-Genomics-Lab/iDeLUCS) accuracy on tests/data/Influenza-A.fas over seeds; see make_anchor_seeds.py"}
    for seed in range(10):
        out["single"].append(run(SINGLE.format(scratch=SCRATCH, seed=seed, fas=fas, gt=gt)))
        print(out["single"][-1], flush=True)
        json.dump(out, open(os.path.join(HERE, "anchor_seeds.json"), "w"), indent=1)
    for seed in range(3):
        out["voters5"].append(run(VOTERS5.format(scratch=SCRATCH, seed=seed, fas=fas, gt=gt)))
        print(out["voters5"][-1], flush=True)
        json.dump(out, open(os.path.join(HERE, "anchor_seeds.json"), "w"), indent=1)


def extend_single(upto):
    """Round 3: more single-voter seeds (10 .. upto-1) appended to the existing fixture -- ten runs of a quantity with a standard
    deviation of 0.06 cannot tell 'the same distribution' from '0.03 worse' (VERDICT r2 weak #1)."""
    build_reference()
    fas, gt = os.path.join(DATA, "Influenza-A.fas"), os.path.join(DATA, "Influenza-A_GT.tsv")
    path = os.path.join(HERE, "anchor_seeds.json")
    out = json.load(open(path))
    have = {r["seed"] for r in out["single"]}
    for seed in range(upto):
        if seed in have:
            continue
        out["single"].append(run(SINGLE.format(scratch=SCRATCH, seed=seed, fas=fas, gt=gt)))
        print(out["single"][-1], flush=True)
        json.dump(out, open(path, "w"), indent=1)


def extend_voters5(upto):
    """More 5-voter ensembles (seeds 3 .. upto-1) appended to the existing fixture."""
    build_reference()
    fas, gt = os.path.join(DATA, "Influenza-A.fas"), os.path.join(DATA, "Influenza-A_GT.tsv")
    path = os.path.join(HERE, "anchor_seeds.json")
    out = json.load(open(path))
    have = {r["seed"] for r in out["voters5"]}
    for seed in range(upto):
        if seed in have:
            continue
        out["voters5"].append(run(VOTERS5.format(scratch=SCRATCH, seed=seed, fas=fas, gt=gt)))
        print(out["voters5"][-1], flush=True)
        json.dump(out, open(path, "w"), indent=1)


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--extend-single":
        extend_single(int(sys.argv[2]))
    elif len(sys.argv) == 3 and sys.argv[1] == "--extend-voters5":
        extend_voters5(int(sys.argv[2]))
    elif len(sys.argv) > 1:          # anything else on the command line gets the usage, not a 15-minute rerun that rewrites the fixture
        sys.exit(__doc__)
    else:
        main()
