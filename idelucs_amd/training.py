"""Shared driver pieces of the two front-ends (library entry point and CLI): model preparation and the
per-voter training run (reference idelucs/cluster.py:33-50 and idelucs/__main__.py:78-123 do the same
things inline, twice)."""
import os
import sys

from . import gemm_tuning, models
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
    for attempt in (0, 1):
        model.begin_voter(voter)
        try:
            curve = [model.contrastive_training_epoch() for _ in range(n_epochs)]
            return (curve,) + tuple(model.predict())
        except models.PlanesOverflow as err:
            # the data (or a diverging run) does not fit the fp16 planes: the fp32 tiles for the rest of this process, and this voter again from
            # its start -- begin_voter re-seeds every stream it draws from, so the rerun is exactly the IDELUCS_PLANES=0 run
            if attempt or not _leave_planes([model], err):
                raise


def _leave_planes(lane_models, err):
    """Switch this process (and the given models' trainers) to the fp32 step form after a PlanesOverflow; False where the voter cannot simply be
    trained again (IDELUCS_VOTER_STATE=carry: its start was the previous voter's optimizer state, which the failed attempt overwrote)."""
    from . import fused
    if models.IID_model.voter_state_carried():
        return False
    sys.stderr.write(f"\n[idelucs_amd] {err}\n[idelucs_amd] -> the fp32 tiles from here on; training the voter(s) again from the start\n")
    fused.disable_planes()
    for m in lane_models:
        if m._fused is not None:
            m._fused.drop_planes()
    return True


def plane_step_applies(model):
    """Whether a lone voter of this model trains with the two big products of its step on the fp16 matrix cores (fused.FusedLinearTrainer._planes:
    NetLinear, full batches of 2 x batch_sz rows with 128 | 2 batch_sz, 512 | F, F >= 1024)."""
    from . import fused
    if not fused.planes_default() or not getattr(model, "_use_fused", False):
        return False
    try:
        lin1 = model.net.layers[0]
        F, H1 = int(lin1.in_features), int(lin1.out_features)
    except (AttributeError, IndexError, TypeError):
        return False
    m = 2 * int(model.batch_sz)
    return H1 == 512 and F >= 1024 and F % 512 == 0 and m >= 256 and m % 128 == 0


def voter_lanes(n_voters_here, model=None):
    """How many voters of one rank train in lockstep as one batch (IDELUCS_VOTER_LANES; 1 = one after the other).
    One training step is a handful of launches, most of them latency-bound; batched, each launch serves every voter of the batch
    (fused.BatchedLinearTrainer: blockIdx.y = voter), the two big products included when the step takes them from fp16 planes
    (plane_step_applies).  Default: all of a rank's voters, up to 8: at cfg2 a voter-epoch costs 54.2 ms alone and 47.5 / 45.4 / 43.6 ms in
    lockstep batches of 2 / 4 / 8 (bench.py: predicted_fixed_job).  With IDELUCS_DEV=lockstep_planes=0 a batch runs the products as batched
    fp32 library GEMMs (58.9 / 54.8 / 52.6 ms): a lone voter on planes then beats a batch of 2, and two voters train one after the other."""
    env = os.environ.get("IDELUCS_VOTER_LANES")
    lanes = max(1, min(int(env) if env is not None else 8, n_voters_here))
    from . import fused
    if env is None and model is not None and lanes == 2 and plane_step_applies(model) and fused.VARIANTS["lockstep_planes"] == "0":
        return 1
    return lanes


def can_batch(model):
    """Voters can be batched when the model runs the default fused launch sequence (NetLinear + RMSprop, n_clusters <= 48, full
    batches a multiple of 16 rows) and no scheduler reads the epoch loss on the host between epochs."""
    # (the CLI hands the scheduler over as the reference does, as a string: "None" unless Plateau / Triangle was asked for)
    from . import fused
    return bool(model._use_fused and model.n_clusters <= 48 and model.batch_sz % 16 == 0
                and model.schedule not in ('Plateau', 'Triangle')
                and fused.VARIANTS["pipeline"] != "0" and fused.VARIANTS["early_gather"] == "1" and fused.VARIANTS["mid_fused"] != "0")


def train_voters(model, voters, n_epochs, n_voters=None, lanes=None, progress=True):
    """The voters this rank owns -> {voter: (loss curve, y_pred, probabilities, latent)}.  Every voter draws from its own RNG
    streams and starts from fresh optimizer state (IID_model.begin_voter), so when and beside whom it trains does not change
    what it is.  Voters run `lanes` at a time in lockstep (fused.BatchedLinearTrainer: one launch sequence for the whole batch of
    voters, each on its own network / permutation / dropout stream / optimizer state; IID_model.lane()); nothing waits on the
    host until a batch's predicts."""
    from .fused import BatchedLinearTrainer
    voters = list(voters)
    n_voters = n_voters if n_voters is not None else len(voters)
    lanes = voter_lanes(len(voters), model) if lanes is None else max(1, min(int(lanes), len(voters)))
    if models.IID_model.voter_state_carried():
        # the reference's one-optimizer-for-all-voters behaviour: voters strictly one after the other, all of them here
        if voters != list(range(n_voters)):
            raise ValueError("IDELUCS_VOTER_STATE=carry needs every voter on one rank (the optimizer state of voter v-1 is voter v's start): "
                             "run without a launcher, or leave the default (independent voters)")
        lanes = 1
    if lanes <= 1 or not can_batch(model):
        return {v: train_voter(model, n_epochs, v, n_voters, progress) for v in voters}
    out = {}
    lane_models, batched = None, None
    for w in range(0, len(voters), lanes):
        wave = voters[w:w + lanes]
        if len(wave) == 1:                                   # a last voter on its own: the single-voter path
            out[wave[0]] = train_voter(model, n_epochs, wave[0], n_voters, progress)
            continue
        if batched is None or batched.L != len(wave):
            lane_models = [model.lane() for _ in wave]
            try:
                batched = BatchedLinearTrainer([m.net for m in lane_models], model.lr, model.weight, model.l, seed=model.seed)
            except ValueError:       # an opt-in launch variant (IDELUCS_DEV=overlap, IDELUCS_DEV=wgrad_fused, ...) the batched step does not take
                for v in voters[w:]:
                    out[v] = train_voter(model, n_epochs, v, n_voters, progress)
                return out
            for m, t in zip(lane_models, batched.trainers):
                m._fused = t
        if progress:
            sys.stdout.write(f"\r........... Training Models ({wave[0] + 1}-{wave[-1] + 1}/{n_voters})................")
            sys.stdout.flush()
        for attempt in (0, 1):
            for m, v in zip(lane_models, wave):
                m.begin_voter(v)
            curves = {v: [] for v in wave}
            n_batches = (model.store.n_pairs + model.batch_sz - 1) // model.batch_sz
            gemm_tuning.maybe_enable(n_batches * n_epochs * len(voters))
            for _ in range(n_epochs):
                for m in lane_models:
                    m.net.train()
                res = batched.run_epoch(model.store, model.batch_sz, [m._gen for m in lane_models])
                for m, v, (total, nb) in zip(lane_models, wave, res):
                    curves[v].append(m._finish_epoch(total / (nb - 1), sync=False))     # models.py:135 quirk (divide by last index)
            try:
                for m, v in zip(lane_models, wave):
                    out[v] = ([float(x) for x in curves[v]],) + tuple(m.predict())
                break
            except models.PlanesOverflow as err:     # (as train_voter: the fp32 form, and the whole batch of voters again)
                if attempt or not _leave_planes(lane_models, err):
                    raise
                batched.drop_planes()
    # the caller goes on with `model`: leave it holding the LAST voter's weights, as after a sequential run
    if lane_models is not None and len(voters) % lanes != 1:
        last = lane_models[(len(voters) - 1) % lanes if len(voters) % lanes else lanes - 1]
        model.net.load_state_dict(last.net.state_dict())
    return out
