#!/usr/bin/env python3
"""Generate tests/golden/small_k5.npz from the REAL reference (build container only; same scratch build as make_golden.py):
BASELINE config 4's k=5 point through model_size='small' -- the reference's kmersFasta(reduce=True, k=5) rows of the first 32
Influenza records (canonical 5-mers: 512 features), a myNet(512, 5) with seeded Kaiming weights, its eval forward and ONE training
step (dropout off, eval mode with autograd): loss, every gradient, parameters after RMSprop(lr=1e-3, weight_decay=0.01).

Usage:  python tests/golden/make_golden_small_k5.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import build_reference, DATA   # noqa: E402


def main():
    build_reference()
    import torch
    from idelucs import utils as U
    from idelucs import models as M
    from idelucs.PytorchUtils import myNet
    from idelucs.LossFunctions import IID_loss, info_nce_loss
    names, rows = U.kmersFasta(os.path.join(DATA, "influenza_64.fas"), k=5, reduce=True)
    rows = rows[:32]
    assert rows.shape == (32, 512)
    mean, std = rows.mean(0), rows.std(0)
    x = ((rows - mean) / np.where(std < 1e-15, 1.0, std)).astype(np.float32)      # StandardScaler, as SequenceDataset does
    g = {"rows": rows, "x": x}
    torch.manual_seed(4321)
    net = myNet(512, 5)
    net.apply(M.weights_init)
    for n_, p in net.state_dict().items():
        g[f"w.{n_}"] = p.numpy().copy()
    x1, x2 = torch.from_numpy(x[:16]), torch.from_numpy(x[16:])
    net.eval()
    out, lat = net(x1.view(-1, 1, 512))
    g["eval_out"], g["eval_latent"] = out.detach().numpy().copy(), lat.detach().numpy().copy()
    opt = torch.optim.RMSprop(net.parameters(), lr=1e-3, weight_decay=0.01)
    opt.zero_grad()
    z1, h1 = net(x1.view(-1, 1, 512)); z2, h2 = net(x2.view(-1, 1, 512))
    loss = (1 - 0.25) * info_nce_loss(h1, h2, 0.85) + 0.25 * IID_loss(z1, z2, lamb=2.8)
    loss.backward()
    g["loss"] = np.float32(loss.item())
    for n_, p in net.named_parameters():          # (every 8th row of the two big weight gradients: keeps the fixture near 1 MB)
        gr = p.grad.numpy()
        g[f"g.{n_}"] = gr[::8].copy() if gr.ndim == 2 and gr.shape[0] >= 128 else gr.copy()
    opt.step()
    for n_, p in net.named_parameters():
        if p.dim() == 1:
            g[f"p.{n_}"] = p.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, "small_k5.npz"), **g)
    print("small_k5.npz: loss", g["loss"], "bytes", os.path.getsize(os.path.join(HERE, "small_k5.npz")))


if __name__ == "__main__":
    main()
