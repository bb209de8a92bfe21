#!/usr/bin/env python3
"""ACC of the Influenza-A anchor (k=6, C=5, 10 epochs, 1 voter) over seeds, fused vs autograd step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd, torch
import idelucs_amd
from idelucs_amd import models

D = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "data")
df = pd.read_csv(os.path.join(D, "Influenza-A_GT.tsv"), sep="\t")
u = {v: i for i, v in enumerate(sorted(set(df.cluster_id)))}
gt = np.array([u[v] for v in df.cluster_id])
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for fused in (0, 1):
    accs = []
    for seed in range(int(os.environ.get("SEEDS", "16"))):
        torch.manual_seed(seed)
        args = {'sequence_file': os.path.join(D, "Influenza-A.fas"), 'GT_file': None, 'n_clusters': 5, 'k': 6, 'model_size': 'linear',
                'n_mimics': 3, 'batch_sz': 512, 'optimizer': 'RMSprop', 'lambda': 2.8, 'lr': 1e-3, 'weight': 0.25, 'scheduler': None,
                'n_epochs': epochs, 'n_voters': 1, 'seed': seed}
        m = models.IID_model(args)
        m._use_fused = bool(fused)
        m.build_dataloader()
        m.net.apply(models.weights_init)
        losses = [m.contrastive_training_epoch() for _ in range(epochs)]
        y, _, _ = m.predict()
        accs.append(idelucs_amd.cluster_acc(gt, y)[1])
        print(f"fused={fused} seed={seed} acc={accs[-1]:.4f} loss0={losses[0]:.4f} lossN={losses[-1]:.4f}", flush=True)
    print(f"== fused={fused}: n {len(accs)} mean {np.mean(accs):.4f} sd {np.std(accs, ddof=1):.4f} se {np.std(accs, ddof=1) / np.sqrt(len(accs)):.4f} "
          f"min {np.min(accs):.4f} max {np.max(accs):.4f}")
    res = globals().setdefault("res", {})
    res[fused] = np.array(accs)
if len(res) == 2:
    from scipy import stats
    a, b = res[0], res[1]
    t = stats.ttest_ind(a, b, equal_var=False)
    ks = stats.ks_2samp(a, b)
    print(f"== autograd - fused: mean difference {np.mean(a) - np.mean(b):+.4f} (Welch t = {t.statistic:.2f}, p = {t.pvalue:.3f}); "
          f"two-sample KS D = {ks.statistic:.3f}, p = {ks.pvalue:.3f}")
