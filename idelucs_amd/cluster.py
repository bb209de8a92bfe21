"""idelucs_amd.cluster -- library entry point, same signature as reference idelucs/cluster.py:9-52."""
import sys

from .utils import SummaryFasta
from . import models


class iDeLUCS_cluster():
    def __init__(self, sequence_file, n_clusters=4, n_epochs=500, n_mimics=3, batch_sz=512, k=4, weight=0.25,
                 n_voters=1):
        self.args = {
            'sequence_file': sequence_file, 'n_clusters': n_clusters, 'n_epochs': n_epochs, 'n_mimics': n_mimics,
            'batch_sz': batch_sz, 'GT_file': None, 'k': k,
            # hard-coded in the reference (cluster.py:24-30)
            'optimizer': "RMSprop", 'lambda': 2.8, 'weight': weight, 'n_voters': n_voters, 'lr': 1e-3,
            'model_size': "linear", 'scheduler': None,
        }

    def fit_predict(self, kmers=None):
        """Returns (y_pred int64 [N], latent float64 [N, 64]) of the LAST voter (cluster.py:32-52);
        the positional argument is ignored, as in the reference."""
        model = models.IID_model(self.args)
        model.names, model.lengths, model.GT, model.cluster_dis = SummaryFasta(model.sequence_file, model.GT_file)
        model.build_dataloader()
        y_pred = latent = None
        for voter in range(self.args['n_voters']):
            sys.stdout.write(f"\r........... Training Model ({voter + 1}/{self.args['n_voters']})................")
            sys.stdout.flush()
            model.net.apply(models.weights_init)
            model.epoch = 0
            for _ in range(self.args['n_epochs']):
                model.contrastive_training_epoch()
            y_pred, _, latent = model.predict()
        return y_pred, latent
