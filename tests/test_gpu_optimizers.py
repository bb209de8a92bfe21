"""The optimizers and schedulers reference idelucs/models.py:87-99 configures besides the default, and the reference's cross-voter
optimizer state, against goldens from the imported reference (tests/golden/make_golden_optimizers.py -> optimizers.npz / .json):
SGD(momentum 0.9, weight_decay 0.01) and Adam steps, the ReduceLROnPlateau and CyclicLR learning-rate traces, and a two-voter
sequence with ONE RMSprop object (IDELUCS_VOTER_STATE=carry) next to the same voter with a fresh one (the default)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

FIN, C, B = 16, 5, 9
NAMES = ["layers.0.weight", "layers.0.bias", "layers.3.weight", "layers.3.bias", "classifier.2.weight", "classifier.2.bias"]


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLDEN, "optimizers.npz"))


def _args(opt, sched):
    return {'sequence_file': None, 'GT_file': None, 'n_clusters': C, 'k': 2, 'model_size': 'linear', 'n_mimics': 3, 'batch_sz': B,
            'optimizer': opt, 'lambda': 2.8, 'lr': 1e-3, 'weight': 0.25, 'scheduler': sched, 'n_epochs': 1, 'n_voters': 2}


def _load(net, g, tag):
    import torch
    sd = {n: torch.from_numpy(g[f"{tag}.w.{n}"]) for n in net.state_dict()}
    net.load_state_dict(sd)


def _close_but_for_flips(got, want, name, rtol=1e-4, atol=1e-6, frac=2e-3):
    # sign-like first steps (Adam: lr * m / (sqrt(v) + eps) = lr * sign(g); RMSprop alike): a near-zero gradient may flip between
    # the CPU and the GPU sums and moves that one weight by 2 lr; everything else must agree to rounding
    bad = ~np.isclose(got, want, rtol=rtol, atol=atol)
    assert bad.mean() <= frac, (name, float(bad.mean()), float(np.abs(got - want).max()))


@pytest.mark.parametrize("opt", ["SGD", "Adam"])
def test_sgd_and_adam_steps_vs_reference(g, opt):
    """Three steps of IID_model's own optimizer object on the golden batch, dropout off (reference models.py:89-92)."""
    import torch
    from idelucs_amd import models
    model = models.IID_model(_args(opt, None))
    meta = json.load(open(os.path.join(GOLDEN, "optimizers.json")))[f"{opt}.defaults"]
    for k, v in meta.items():                                   # the same optimizer configuration, key by key
        have = model.optimizer.defaults[k]
        assert (list(have) if isinstance(have, tuple) else have) == v, (k, have, v)
    assert not model._use_fused
    _load(model.net, g, "init11")
    model.net.eval()
    x = torch.cat([torch.from_numpy(g["x1.0"]), torch.from_numpy(g["x2.0"])]).to(model.device)
    for it in range(3):
        loss = model._step(x)
        ref = float(g[f"{opt}.step{it}.loss"])
        assert abs(loss.item() - ref) <= 3e-4 * abs(ref), (opt, it, loss.item(), ref)
        for n_, p in zip(NAMES, model.net.parameters()):
            key = f"{opt}.step{it}.p.{n_}"
            if key in g.files:
                _close_but_for_flips(p.detach().cpu().numpy(), g[key], key, frac=2e-3 if opt == "Adam" else 0.0)


@pytest.mark.parametrize("opt,sched", [("RMSprop", "Plateau"), ("RMSprop", "Triangle"), ("SGD", "Triangle")])
def test_scheduler_learning_rate_traces_vs_reference(g, opt, sched):
    """The learning rate after each of 30 epochs, the scheduler stepped exactly as contrastive_training_epoch steps it
    (reference models.py:96-99,137-140; Triangle ignores --lr: quirk #14)."""
    import torch
    from idelucs_amd import models
    model = models.IID_model(_args(opt, sched))
    meta = json.load(open(os.path.join(GOLDEN, "optimizers.json")))
    assert model.optimizer.param_groups[0]['lr'] == meta[f"{opt}.{sched}.lr_at_construction"]
    want = g[f"{opt}.{sched}.lr"]
    losses = g["plateau.losses"]
    got = []
    for e in range(30):
        model.optimizer.zero_grad()
        for p in model.net.parameters():
            p.grad = torch.zeros_like(p)
        model.optimizer.step()
        model._finish_epoch(torch.tensor(float(losses[e]), device=model.device), sync=False)
        got.append(model.optimizer.param_groups[0]['lr'])
    np.testing.assert_allclose(np.array(got), want, rtol=1e-12, atol=0)
    assert model.epoch == 30
    # the fused RMSprop step takes its learning rate from the torch optimizer the scheduler acts on
    if opt == "RMSprop":
        from idelucs_amd.fused import FusedLinearTrainer
        tr = FusedLinearTrainer(model.net, model.lr, model.weight, model.l)
        tr.set_lr(model.optimizer.param_groups[0]['lr'])
        assert abs(tr.hyper[0].item() - want[-1]) <= 1e-7 * want[-1]


@pytest.mark.parametrize("state", ["carry", "fresh"])
def test_two_voters_one_optimizer_vs_reference(g, state, monkeypatch):
    """the reference keeps ONE RMSprop object across voters (models.py:87-88, __main__.py:109).
    IDELUCS_VOTER_STATE=carry reproduces that on one rank: voter 1 trains two epochs of three batches, voter 2 starts from fresh
    weights WITH voter 1's square_avg -- parameters after its first step equal the reference's; the default ("fresh": a voter is
    an independent run, the only form a sharded ensemble can have) equals the reference's voter 2 with a fresh optimizer, and
    the two differ."""
    import torch
    from idelucs_amd import models
    from idelucs_amd.fused import FusedLinearTrainer
    monkeypatch.setenv("IDELUCS_VOTER_STATE", state)
    model = models.IID_model(_args("RMSprop", None))
    assert model._use_fused and models.IID_model.voter_state_carried() == (state == "carry")
    tr = FusedLinearTrainer(model.net, model.lr, model.weight, model.l, seed=0)
    model._fused = tr
    bf = tr.buffers(2 * B)
    xs = [torch.cat([torch.from_numpy(g[f"x1.{i}"]), torch.from_numpy(g[f"x2.{i}"])]).to(model.device) for i in range(3)]
    model.begin_voter(0)
    _load(model.net, g, "carry.v1")
    for epoch in range(2):
        tr.out[1] = 0.0
        for i in range(3):
            bf.x.copy_(xs[i])
            tr.step_on_batch(bf, train=False)
        ref = float(g["carry.v1.epoch_loss"][epoch])
        got = tr.out[1].item() / 2                               # reference models.py:135: divided by the last batch index
        assert abs(got - ref) <= 5e-4 * abs(ref), (epoch, got, ref)
    model.begin_voter(1)                                         # carry: only the weights start over
    assert (tr.square_avg[0].abs().max().item() > 0) == (state == "carry")
    _load(model.net, g, "carry.v2")
    bf.x.copy_(xs[0])
    tr.step_on_batch(bf, train=False)
    torch.cuda.synchronize()
    if state == "carry":
        ref = float(g["carry.v2.step0.loss"])
        assert abs(tr.out[0].item() - ref) <= 2e-4 * abs(ref)
    for n_, p in zip(NAMES, tr.params):
        got = p.detach().cpu().numpy()
        mine, other = g[f"{state}.v2.step0.p.{n_}"], g[f"{'fresh' if state == 'carry' else 'carry'}.v2.step0.p.{n_}"]
        _close_but_for_flips(got, mine, n_, rtol=1e-3, atol=1e-6, frac=5e-3)
        if p.dim() == 2:
            assert np.abs(got - other).mean() > 10 * np.abs(got - mine).mean(), n_      # the two behaviours are told apart


def test_carried_state_refuses_a_sharded_run(monkeypatch):
    from idelucs_amd import models, training
    monkeypatch.setenv("IDELUCS_VOTER_STATE", "carry")
    model = models.IID_model(_args("RMSprop", None))
    with pytest.raises(ValueError, match="every voter on one rank"):
        training.train_voters(model, [1], 1, n_voters=2, progress=False)


@pytest.mark.gpu
def test_gemm_tuning_only_from_long_jobs(monkeypatch):
    """idelucs_amd.gemm_tuning: a job of fewer than MIN_STEPS optimizer steps USES the shipped GEMM solutions (TunableOp enabled) but
    tunes nothing -- on the reference's own 949-sequence example the CLI's defaults are 3 000 steps, and tuning their partial-batch
    and batched-voter shapes took 6 of the run's 8.6 s; a longer job in the same process switches tuning on; stop_tuning() ends it."""
    import torch.cuda.tunable as tn
    from idelucs_amd import gemm_tuning as g
    monkeypatch.delenv("IDELUCS_TUNABLEOP", raising=False)
    saved = (g._enabled, g._tuning, tn.is_enabled(), tn.tuning_is_enabled())
    try:
        assert g.MIN_STEPS >= 100000
        assert g.maybe_enable(3000) and tn.is_enabled()
        if not saved[1]:
            assert not g._tuning and not tn.tuning_is_enabled()
        assert g.maybe_enable(g.MIN_STEPS) and g._tuning and tn.tuning_is_enabled()
        g.stop_tuning()
        assert not g._tuning and not tn.tuning_is_enabled() and tn.is_enabled()
        monkeypatch.setenv("IDELUCS_TUNABLEOP", "0")
        assert g.maybe_enable(10 ** 9) is False
    finally:
        g._tuning = saved[1]
        tn.tuning_enable(saved[3])
        tn.enable(saved[2] or g._enabled)
