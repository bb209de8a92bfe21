"""Shared driver pieces of the two front-ends (library entry point and CLI): model preparation and the
per-voter training run (reference idelucs/cluster.py:33-50 and idelucs/__main__.py:78-123 do the same
things inline, twice)."""
import os
import sys

import torch

from . import models
from .utils import SummaryFasta


def prepare_model(args):
    """IID_model + FASTA summary + device feature store (reference cluster.py:33-37 / __main__.py:78-95)."""
    model = models.IID_model(args)
    model.names, model.lengths, model.GT, model.cluster_dis = SummaryFasta(model.sequence_file, model.GT_file)
    model.build_dataloader()
    return model


def train_voter(model, n_epochs, voter=0, n_voters=1, progress=True):
    """One voter: fresh Kaiming init, n_epochs epochs, predict.  -> (loss curve, y_pred, probabilities, latent)."""
    if progress:
        sys.stdout.write(f"\r........... Training Model ({voter + 1}/{n_voters})................")
        sys.stdout.flush()
    model.begin_voter(voter)
    curve = [model.contrastive_training_epoch() for _ in range(n_epochs)]
    return (curve,) + tuple(model.predict())


def voter_lanes(n_voters_here):
    """How many voters of one rank train side by side, each on its own HIP stream (IDELUCS_VOTER_LANES, default 1).
    Measured on MI355X (tools/concurrent_voters.py, tools/stream_overlap.hip): streams and graphs of different streams do overlap
    on this GPU, but the training step's two GEMMs put one workgroup on every CU and its 1024-thread middle kernels do not fit
    beside them, so lanes only fill launch gaps -- 1.17 x at n_clusters = 20 with 50 000 sequences, nothing at the cfg2 size (a
    16-step graph has no gaps), 1.47 x on the one-step-per-replay path.  Off by default; the results are identical either way."""
    lanes = int(os.environ.get("IDELUCS_VOTER_LANES", "1"))
    return max(1, min(lanes, n_voters_here))


def train_voters(model, voters, n_epochs, n_voters=None, lanes=None, progress=True):
    """The voters this rank owns -> {voter: (loss curve, y_pred, probabilities, latent)}, the same results as train_voter()
    one voter after the other: every voter draws from its own RNG streams and starts from fresh optimizer state
    (IID_model.begin_voter), so when and beside whom it trains does not matter.  Voters run `lanes` at a time, each lane on its
    own HIP stream with its own network, step buffers and captured graph (IID_model.lane()); the epochs of one wave are
    enqueued round-robin and nothing waits on the host until the wave's predicts."""
    voters = list(voters)
    n_voters = n_voters if n_voters is not None else len(voters)
    lanes = voter_lanes(len(voters)) if lanes is None else max(1, min(int(lanes), len(voters)))
    if not model._use_fused:                 # the autograd paths draw dropout masks from the process-wide generator: one at a time
        lanes = 1
    if lanes <= 1:
        out = {}
        for v in voters:
            r = train_voter(model, n_epochs, v, n_voters, progress)
            out[v] = r
        return out
    models_ = [model] + [model.lane() for _ in range(lanes - 1)]
    streams = [torch.cuda.Stream(device=model.device) for _ in models_]
    cur = torch.cuda.current_stream()
    out = {}
    for w in range(0, len(voters), lanes):
        wave = voters[w:w + lanes]
        if progress:
            sys.stdout.write(f"\r........... Training Models ({wave[0] + 1}-{wave[-1] + 1}/{n_voters})................")
            sys.stdout.flush()
        curves = {v: [] for v in wave}
        for m, s, v in zip(models_, streams, wave):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                m.begin_voter(v)
        for _ in range(n_epochs):
            pending = []
            for m, s, v in zip(models_, streams, wave):
                with torch.cuda.stream(s):
                    pending.append(m.enqueue_epoch())
            for m, s, v, loss in zip(models_, streams, wave, pending):     # schedulers may read the loss: after every lane is enqueued
                with torch.cuda.stream(s):
                    curves[v].append(m._finish_epoch(loss, sync=False))
        for m, s, v in zip(models_, streams, wave):
            with torch.cuda.stream(s):
                out[v] = ([float(x) for x in curves[v]],) + tuple(m.predict())
        for s in streams:
            cur.wait_stream(s)
    # the caller goes on with `model`: leave it holding the LAST voter's weights, as after a sequential run
    last = models_[(len(voters) - 1) % lanes]
    if last is not model:
        model.net.load_state_dict(last.net.state_dict())
    return out
