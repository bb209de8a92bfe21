"""idelucs_amd.models -- IID_model: the training driver of the hot path on one MI355X.

Same constructor dict keys and the same five methods as reference idelucs/models.py:46-195
(IID_model.__init__, build_dataloader, contrastive_training_epoch, predict, calculate_probs).
Differences are all below the interface: features live in HBM (utils.FeatureStore), batches are
assembled by a HIP gather kernel instead of DataLoader workers, the two views go through the encoder
as one [2B, F] batch, losses are mask-free device code, and no step synchronises with the host.
"""
import os
import random
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

from . import _lib
from . import utils
from .LossFunctions import IID_loss, info_nce_loss
from .PytorchUtils import NetLinear, myNet

# Random seeds for reproducibility, at import, as the reference does (models.py:17-21): the
# "compat" mimic RNG relies on numpy/random being seeded 0 exactly like a fresh `import idelucs`.
torch.manual_seed(0)
np.random.seed(0)
random.seed(0)

EPS = sys.float_info.epsilon


PREDICT_CACHE_BYTES = 32 << 30      # standardised un-mutated vectors kept between the voters' predicts (1.6 GB at 100k x 4^6)


def weights_init(m, generator=None):
    """Reference models.py:36-44: Kaiming-normal weights, zero biases, for every nn.Linear."""
    if isinstance(m, nn.Linear):
        torch.nn.init.kaiming_normal_(m.weight, generator=generator)
        torch.nn.init.zeros_(m.bias)


def _fused_on():
    from . import fused
    return fused.VARIANTS["fused"] != "0"


class PlanesOverflow(RuntimeError):
    """An operand of the training step's two big products left the range of its fp16 planes (IID_model._check_planes)."""


class IID_model():
    def __init__(self, args: dict):
        _lib.require_gpu()          # no CPU path: fail here, loudly, rather than train on the host
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.sequence_file = args['sequence_file']
        self.GT_file = args['GT_file']
        self.n_clusters = args['n_clusters']
        self.k = args['k']

        if args['model_size'] == 'linear':                      # models.py:54-57
            self.n_features = 4 ** self.k
            self.net = NetLinear(self.n_features, args['n_clusters'])
            self.reduce = False
        elif args['model_size'] == 'small':                     # models.py:59-66
            self.n_features = (4 ** self.k + 4 ** (self.k // 2)) // 2 if self.k % 2 == 0 else (4 ** self.k) // 2
            self.net = myNet(self.n_features, args['n_clusters'])
            self.reduce = True
        else:
            # 'full' builds a ResNet18 in the reference but then crashes in build_dataloader
            # (models.py:68-69,111: n_features/reduce never set); 'conv' hits this same ValueError.
            raise ValueError("Invalid Model Type")

        self.net.apply(weights_init)
        self.net.to(self.device)
        self.epoch = 0
        self.EPS = EPS
        self.n_mimics = args['n_mimics']
        self.batch_sz = args['batch_sz']
        self.optimizer = args['optimizer']
        self.l = args['lambda']
        self.lr = args['lr']
        self.weight = args['weight']
        self.schedule = args['scheduler']
        self.mutate = True
        self.n_epochs = args.get('n_epochs', 1)
        self.n_voters = args.get('n_voters', 1)
        self.rng = args.get('rng')            # None -> $IDELUCS_RNG or "philox"
        self.seed = args.get('seed', 0)

        if self.optimizer == 'RMSprop':                         # models.py:87-94
            self.optimizer = optim.RMSprop(self.net.parameters(), lr=self.lr, weight_decay=0.01)
        elif self.optimizer == 'SGD':
            self.optimizer = optim.SGD(self.net.parameters(), lr=self.lr, weight_decay=0.01, momentum=0.9)
        elif self.optimizer == 'Adam':
            self.optimizer = optim.Adam(self.net.parameters(), lr=self.lr)
        else:
            raise ValueError("Optimizer not supported")

        self._make_scheduler()
        self.store = None
        self._args = dict(args)
        self._shared = {}           # what the lanes of one ensemble share besides the store (see lane()): the predict inputs
        self._gen = None            # this voter's private device generator (begin_voter)
        self._voter = 0
        # default configuration (NetLinear + RMSprop): explicit fused step replayed as a HIP graph
        self._fused = None
        self._use_fused = (args['model_size'] == 'linear' and args['optimizer'] == 'RMSprop'
                           and args['n_clusters'] <= 256 and _fused_on())

    def _make_scheduler(self):
        if self.schedule == 'Plateau':                          # models.py:96-99
            self.scheduler = optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, 'min')
        elif self.schedule == 'Triangle':
            self.scheduler = optim.lr_scheduler.CyclicLR(self.optimizer, base_lr=0.001, max_lr=0.1, step_size_up=5,
                                                         mode="triangular2")

    def lane(self):
        """A sibling model for training another voter of the same ensemble AT THE SAME TIME on this GPU (training.train_voters):
        its own network, optimizer state, step buffers and captured graph; the feature store, the FASTA summary and the
        predict inputs are shared, read-only."""
        other = IID_model(self._args)
        other.store, other._shared = self.store, self._shared
        other.dataloader = getattr(self, "dataloader", None)
        for name in ("names", "lengths", "GT", "cluster_dis"):
            if hasattr(self, name):
                setattr(other, name, getattr(self, name))
        return other

    # ------------------------------------------------------------------ data
    def build_dataloader(self):
        """Reference models.py:101-111: vectorise all mimic views + fit the scaler (one kernel launch
        each, utils.build_feature_store) and expose an iterable of device batches."""
        # (a second call -- the same file again, or another of the same record count -- refits the previous store in place: the
        #  captured step graph bakes the store's addresses and is then not captured again)
        self.store = utils.build_feature_store(self.sequence_file, self.n_mimics, k=self.k, reduce=self.reduce,
                                               rng=self.rng, seed=self.seed, device=self.device, streamed=True, reuse=self.store)
        self.dataloader = utils.DeviceBatchLoader(self.store, self.batch_sz)
        self._shared.clear()

    # ------------------------------------------------------------------ training
    @staticmethod
    def voter_state_carried():
        """IDELUCS_VOTER_STATE=carry: the reference's behaviour when all voters train in ONE process -- a single optimizer (and
        scheduler) object serves every voter (reference models.py:87-99; __main__.py:109 re-initialises only the weights), so
        voter v starts with voter v-1's RMSprop running averages, learning rate and scheduler counters.  Only meaningful when the
        voters run one after the other on one rank (training.train_voters and cluster.fit_predict see to that and refuse a sharded
        run); the default, "fresh", makes every voter an independent run that can train anywhere."""
        return os.environ.get("IDELUCS_VOTER_STATE", "fresh") == "carry"

    def begin_voter(self, voter=0):
        """A fresh voter (reference __main__.py:109: weights_init between voters).  The reference's voters differ because
        its one process consumes the torch RNG sequentially; here voters may run on different ranks, or side by side on one GPU,
        so every stream a voter draws from (Kaiming init, batch permutations, dropout) is a function of (seed, voter index) and
        voter v is the same run wherever and whenever it trains.  For the same reason the optimizer and scheduler state start
        from scratch: the reference carries RMSprop's running averages (and the scheduler's counters) from voter v-1 into voter
        v, which a sharded ensemble cannot do; every voter here starts where the reference's FIRST voter starts.  The
        mimic/data seed stays shared: every rank builds the same feature store."""
        self._voter = int(voter)
        vseed = (int(self.seed) * 1000003 + 1 + self._voter) & (2 ** 63 - 1)
        torch.manual_seed(vseed)                                  # CPU and every CUDA generator (the unfused paths draw from these)
        self._gen = torch.Generator(device=self.device).manual_seed(vseed)
        self.net.apply(lambda mod: weights_init(mod, self._gen))
        self.epoch = 0
        carry = self.voter_state_carried() and self._voter > 0
        if not carry:
            self.optimizer.state.clear()
            for grp in self.optimizer.param_groups:
                grp['lr'] = self.lr
            self._make_scheduler()
        if self._fused is not None:
            self._fused.begin_voter(self._voter, keep_state=carry)

    def _step(self, x):
        """One optimizer step on a [2b, F] batch (rows [0,b) "true", [b,2b) "modified")."""
        b = x.shape[0] // 2
        self.optimizer.zero_grad(set_to_none=True)
        z, h = self.net(x)                      # one pass for both views (independent dropout masks per row)
        loss = (1 - self.weight) * info_nce_loss(h[:b], h[b:], 0.85) + self.weight * IID_loss(z[:b], z[b:], lamb=self.l)
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def contrastive_training_epoch(self, sync=True):
        """Reference models.py:113-143: one pass over the shuffled N*n_mimics pairs -> the epoch loss as a Python float
        (sync=False: as a device scalar, without waiting for the epoch to finish)."""
        return self._finish_epoch(self.enqueue_epoch(), sync)

    def enqueue_epoch(self):
        """The epoch's launches on the current stream -> its loss as a device scalar (scheduler step and epoch counter:
        _finish_epoch)."""
        self.net.train()
        st = self.store
        if self._use_fused:
            from .fused import FusedLinearTrainer
            if self._fused is None:
                from . import gemm_tuning
                n_batches = (st.n_pairs + self.batch_sz - 1) // self.batch_sz
                gemm_tuning.maybe_enable(n_batches * self.n_epochs * self.n_voters)
                self._fused = FusedLinearTrainer(self.net, self.lr, self.weight, self.l, seed=self.seed)
                self._fused.begin_voter(self._voter)
            self._fused.set_lr(self.optimizer.param_groups[0]['lr'])       # schedulers act on the torch optimizer
            total, n_batches = self._fused.run_epoch(st, self.batch_sz, generator=self._gen)
            return total / (n_batches - 1)                                  # models.py:135 quirk (divide by last index)
        running_loss = torch.zeros((), device=self.device)
        perm = torch.randperm(st.n_pairs, device=self.device, generator=self._gen)
        i_batch = 0
        for i_batch, i in enumerate(range(0, st.n_pairs, self.batch_sz)):
            x = st.gather_pairs(perm[i:i + self.batch_sz])
            running_loss += self._step(x)
        return running_loss / i_batch              # models.py:135 divides by the LAST INDEX (n_batches-1): kept

    def _finish_epoch(self, running_loss, sync=True):
        if self.schedule == 'Plateau':
            self.scheduler.step(running_loss)
        elif self.schedule == 'Triangle':
            self.scheduler.step()
        self.epoch += 1
        if sync:
            loss = running_loss.item()
            self._check_planes()
            return loss
        return running_loss

    def _check_planes(self):
        """The two-plane step form (fused.FusedLinearTrainer._planes) clamps what leaves its planes' range -- a weight of Linear(F,512) beyond +-15.8, a
        standardised feature beyond +-8125 (a mimic's k-mer thousands of the originals' standard deviations out), a gradient entry more than 128 x the
        previous step's largest, anything not finite -- and raises a flag: checked wherever this model waits for the device anyway (a synchronous
        epoch, predict).  training.train_voter(s) catch PlanesOverflow, switch the process to the fp32 tiles and train the voter again from its
        start (every stream a voter draws from is a function of (seed, voter): the rerun IS the IDELUCS_PLANES=0 run)."""
        if self._fused is not None and self._fused.planes_overflowed():
            raise PlanesOverflow("a weight of the first layer (|w| >= 15.8), a standardised feature (|x| > 8125) or an entry of the layer-1 gradient left "
                                 "the range of the fp16 planes: train this voter again on the fp32 tiles (training.train_voter does; or IDELUCS_PLANES=0)")

    # ------------------------------------------------------------------ inference
    def _predict_inputs(self, rows=None):
        """utils.predict_features of this model's file.  The reference re-reads and re-vectorises the file for every voter's
        predict (models.py:147-163); the result is a pure function of the file, so one ensemble computes it once (kept while it
        fits PREDICT_CACHE_BYTES, shared between lanes)."""
        key = (self.sequence_file, self.k, self.reduce, rows)
        hit = self._shared.get("predict_inputs")
        if hit is not None and hit[0] == key:
            torch.cuda.current_stream().wait_event(hit[2])
            return hit[1]
        feats = utils.predict_features(self.sequence_file, k=self.k, reduce=self.reduce, device=self.device, rows=rows,
                                       with_names=False)[2]
        if feats.numel() * 4 <= PREDICT_CACHE_BYTES:
            ready = torch.cuda.Event()
            ready.record()
            self._shared["predict_inputs"] = (key, feats, ready)
        return feats

    def _predict_outputs(self, rows=None):
        feats = self._predict_inputs(rows)
        outs, lats = [], []
        with torch.no_grad():
            self.net.eval()
            # (the reference walks predict in batch_sz rows, models.py:158-170; every op of the eval forward is row-wise, so the chunk
            #  only sets the launch count: 32768 rows per chunk, 3.08 ms per 100 000 rows against 3.32 at 8192 and ~10 at 512)
            chunk = max(int(self.batch_sz), 32768)
            for i in range(0, feats.shape[0], chunk):
                o, l = self.net(feats[i:i + chunk])
                outs.append(o)
                lats.append(l)
        return torch.cat(outs), torch.cat(lats)

    def predict(self, data=None):
        """Reference models.py:145-172 -> (int64 y_pred[N], float64 probs[N], float64 latent[N,64])."""
        outputs, latent = self._predict_outputs()
        self._check_planes()
        probs, predicted = torch.max(outputs, 1)
        return (predicted.cpu().numpy().astype(np.int64), probs.double().cpu().numpy(),
                latent.double().cpu().numpy())

    def predict_latent_shard(self, lo, hi):
        """Rows [lo, hi) of predict()'s latent as a float32 device tensor [hi-lo, 64] (n_clusters=0 mode on several GPUs:
        predict is sharded by sequence and the shards are all-gathered, idelucs_amd.dist.all_gather_rows)."""
        if hi <= lo:
            return torch.empty((0, 64), dtype=torch.float32, device=self.device)
        return self._predict_outputs(rows=(lo, hi))[1].float().contiguous()

    def calculate_probs(self, data=None):
        """Reference models.py:175-195 -> float64 [N, n_clusters] softmax outputs."""
        outputs, _ = self._predict_outputs()
        return outputs.double().cpu().numpy()
