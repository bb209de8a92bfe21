"""idelucs_amd.cluster -- the scikit-learn-like entry point, `iDeLUCS_cluster(...).fit_predict()`, with the
constructor signature and return values of reference idelucs/cluster.py:9-52."""
from .models import IID_model
from .training import prepare_model, train_voter

# what the reference hard-codes for this entry point (cluster.py:20-30)
_PINNED = {"GT_file": None, "optimizer": "RMSprop", "lambda": 2.8, "lr": 1e-3, "model_size": "linear", "scheduler": None}


class iDeLUCS_cluster():
    def __init__(self, sequence_file, n_clusters=4, n_epochs=500, n_mimics=3, batch_sz=512, k=4, weight=0.25, n_voters=1):
        chosen = dict(sequence_file=sequence_file, n_clusters=n_clusters, n_epochs=n_epochs, n_mimics=n_mimics,
                      batch_sz=batch_sz, k=k, weight=weight, n_voters=n_voters)
        self.args = {**chosen, **_PINNED}

    def fit_predict(self, kmers=None):
        """(y_pred int64 [N], latent float64 [N, 64]) of the LAST voter; no ensembling here, and the
        positional argument is ignored -- both as in the reference (cluster.py:32,47-52)."""
        model = prepare_model(self.args)
        # every voter is a function of (seed, voter index) alone (IID_model.begin_voter), so the voters the reference trains
        # and then discards here need not be trained at all
        last = self.args["n_voters"] - 1
        if IID_model.voter_state_carried():
            # IDELUCS_VOTER_STATE=carry: the reference's sequence -- one optimizer for all voters (reference models.py:87-88), so
            # voter v starts from voter v-1's RMSprop state and every voter has to be trained (cluster.py:39-52)
            for v in range(last):
                train_voter(model, self.args["n_epochs"], v, self.args["n_voters"])
        _, y_pred, _, latent = train_voter(model, self.args["n_epochs"], last, self.args["n_voters"])
        return y_pred, latent
