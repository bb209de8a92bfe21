#!/usr/bin/env python3
"""Generate tests/golden/optimizers.npz + optimizers.json from the REAL reference (build container only; same scratch build as
make_golden.py).  Pins what reference idelucs/models.py:87-99 configures besides the default RMSprop:

  * SGD(lr, weight_decay=0.01, momentum=0.9) and Adam(lr): three optimizer steps of IID_model's OWN optimizer object on a fixed
    batch, dropout off -- loss of every step, parameters after every step;
  * ReduceLROnPlateau(optimizer, 'min') and CyclicLR(base 1e-3, max 1e-1, step_size_up 5, triangular2): the learning rate after
    each of 30 scheduler steps, driven exactly as contrastive_training_epoch drives them (models.py:137-140);
  * the reference's cross-voter optimizer state: ONE RMSprop object serves every voter
    (models.py:87-88; __main__.py:109 re-initialises only the weights) -- voter 1 trains two epochs of three batches, voter 2
    starts from fresh weights with voter 1's square_avg; parameters after voter 2's first step and first epoch, next to the same
    voter 2 trained with a fresh optimizer.

Usage:  python tests/golden/make_golden_optimizers.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import build_reference   # noqa: E402

FIN, C, B = 16, 5, 9          # k = 2 widths keep the fixture small


def model_args(opt, sched):
    return {'sequence_file': None, 'GT_file': None, 'n_clusters': C, 'k': 2, 'model_size': 'linear', 'n_mimics': 3, 'batch_sz': B,
            'optimizer': opt, 'lambda': 2.8, 'lr': 1e-3, 'weight': 0.25, 'scheduler': sched}


def main():
    build_reference()
    import torch
    import torch.nn as nn
    from idelucs import models as M
    from idelucs.LossFunctions import IID_loss, info_nce_loss
    g, meta = {}, {}
    torch.manual_seed(2024)
    batches = []
    for i in range(3):
        x1 = torch.randn(B, FIN); x2 = x1 + 0.1 * torch.randn(B, FIN)
        batches.append((x1, x2))
        g[f"x1.{i}"] = x1.numpy().copy(); g[f"x2.{i}"] = x2.numpy().copy()

    def fresh_weights(model, seed, tag):
        torch.manual_seed(seed)
        model.net.apply(M.weights_init)
        for n_, p in model.net.state_dict().items():
            g[f"{tag}.w.{n_}"] = p.numpy().copy()

    def no_dropout(model):
        for mod in model.net.modules():
            if isinstance(mod, nn.Dropout):
                mod.p = 0.0

    # ---- SGD / Adam: three steps of the model's own optimizer
    for opt in ("SGD", "Adam"):
        model = M.IID_model(model_args(opt, None))
        fresh_weights(model, 11, "init11")
        model.net.eval()
        x1, x2 = batches[0]
        for it in range(3):
            model.optimizer.zero_grad()
            z1, h1 = model.net(x1.view(-1, 1, FIN)); z2, h2 = model.net(x2.view(-1, 1, FIN))
            loss = (1 - model.weight) * info_nce_loss(h1, h2, 0.85) + model.weight * IID_loss(z1, z2, lamb=model.l)
            loss.backward()
            model.optimizer.step()
            g[f"{opt}.step{it}.loss"] = np.float32(loss.item())
            for n_, p in model.net.named_parameters():          # (biases after every step, everything after the last: keeps the fixture small)
                if it == 2 or p.dim() == 1:
                    g[f"{opt}.step{it}.p.{n_}"] = p.detach().numpy().copy()
        meta[f"{opt}.defaults"] = {k: (v if not isinstance(v, tuple) else list(v)) for k, v in model.optimizer.defaults.items()
                                   if isinstance(v, (int, float, bool, tuple))}

    # ---- schedulers: the learning rate after each epoch's scheduler step
    losses = [2.0 - 0.1 * e for e in range(6)] + [1.5] * 24            # improves for six epochs, then a plateau
    g["plateau.losses"] = np.array(losses, np.float32)
    for opt, sched in (("RMSprop", "Plateau"), ("RMSprop", "Triangle"), ("SGD", "Triangle")):
        model = M.IID_model(model_args(opt, sched))
        trace = []
        for e in range(30):
            # (an optimizer step first, as in an epoch: CyclicLR warns otherwise; gradients are zero -- the trace is the lr's)
            model.optimizer.zero_grad()
            for p in model.net.parameters():
                p.grad = torch.zeros_like(p)
            model.optimizer.step()
            if sched == "Plateau":
                model.scheduler.step(torch.tensor(losses[e]))
            else:
                model.scheduler.step()
            trace.append(model.optimizer.param_groups[0]['lr'])
        g[f"{opt}.{sched}.lr"] = np.array(trace, np.float64)
        meta[f"{opt}.{sched}.lr_at_construction"] = None
    for opt, sched in (("RMSprop", "Triangle"), ("SGD", "Triangle"), ("RMSprop", "Plateau")):
        model = M.IID_model(model_args(opt, sched))
        meta[f"{opt}.{sched}.lr_at_construction"] = model.optimizer.param_groups[0]['lr']

    # ---- one optimizer for all voters (reference behaviour) vs a fresh one for voter 2
    def run_epoch(model):
        model.dataloader = [{'true': a, 'modified': b} for a, b in batches]
        return model.contrastive_training_epoch()

    model = M.IID_model(model_args("RMSprop", None))
    no_dropout(model)
    fresh_weights(model, 21, "carry.v1")
    g["carry.v1.epoch_loss"] = np.array([run_epoch(model), run_epoch(model)], np.float32)
    fresh_weights(model, 22, "carry.v2")                               # __main__.py:109: only the weights start over
    model.dataloader = [{'true': batches[0][0], 'modified': batches[0][1]}]
    # first step of voter 2 alone (an "epoch" of one batch would divide by i_batch = 0: step by hand, as the epoch body does)
    model.net.train()
    model.optimizer.zero_grad()
    z1, h1 = model.net(batches[0][0].view(-1, 1, FIN)); z2, h2 = model.net(batches[0][1].view(-1, 1, FIN))
    loss = (1 - model.weight) * info_nce_loss(h1, h2, 0.85) + model.weight * IID_loss(z1, z2, lamb=model.l)
    loss.backward(); model.optimizer.step()
    g["carry.v2.step0.loss"] = np.float32(loss.item())
    for n_, p in model.net.named_parameters():
        g[f"carry.v2.step0.p.{n_}"] = p.detach().numpy().copy()
    # ... the same voter 2 with a FRESH optimizer (what a sharded ensemble does: every voter starts where voter 1 started)
    fresh = M.IID_model(model_args("RMSprop", None))
    no_dropout(fresh)
    sd = {k[len("carry.v2.w."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("carry.v2.w.")}
    fresh.net.load_state_dict(sd)
    fresh.net.train()
    fresh.optimizer.zero_grad()
    z1, h1 = fresh.net(batches[0][0].view(-1, 1, FIN)); z2, h2 = fresh.net(batches[0][1].view(-1, 1, FIN))
    loss = (1 - fresh.weight) * info_nce_loss(h1, h2, 0.85) + fresh.weight * IID_loss(z1, z2, lamb=fresh.l)
    loss.backward(); fresh.optimizer.step()
    for n_, p in fresh.net.named_parameters():
        g[f"fresh.v2.step0.p.{n_}"] = p.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, "optimizers.npz"), **g)
    json.dump(meta, open(os.path.join(HERE, "optimizers.json"), "w"), indent=1)
    print("optimizers.npz:", os.path.getsize(os.path.join(HERE, "optimizers.npz")), "bytes;", json.dumps(meta))
    print("Plateau lr:", g["RMSprop.Plateau.lr"][[0, 10, 16, 17, 28, 29]], " Triangle lr:", g["RMSprop.Triangle.lr"][:12])


if __name__ == "__main__":
    main()
