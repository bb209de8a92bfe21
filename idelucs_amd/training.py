"""Shared driver pieces of the two front-ends (library entry point and CLI): model preparation and the
per-voter training run (reference idelucs/cluster.py:33-50 and idelucs/__main__.py:78-123 do the same
things inline, twice)."""
import sys

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
