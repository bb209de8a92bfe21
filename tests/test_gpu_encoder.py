"""GPU tests (-m gpu) of the encoder side of the hot path: nets, losses, one full training step and
the end-to-end quality anchor, against golden vectors produced by the real reference on CPU.
Floating point: tolerances are stated per test (fp32 GEMMs on MFMA vs CPU BLAS summation order)."""
import json
import os

import numpy as np
import pytest

from conftest import DATA, GOLDEN

pytestmark = pytest.mark.gpu

G = None


def g5():
    global G
    if G is None:
        G = np.load(os.path.join(GOLDEN, "nets.npz"))
    return G


@pytest.fixture(scope="module")
def dev():
    import torch
    from idelucs_amd import _lib
    _lib.require_gpu()
    torch.backends.cuda.matmul.allow_tf32 = False
    return torch.device("cuda")


def _load_net(tag, dev):
    import torch
    from idelucs_amd.PytorchUtils import NetLinear, myNet
    g = g5()
    net = (NetLinear(16, 5) if tag == "linear" else myNet(10, 7))
    sd = {k[len(tag) + 3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + ".w.")}
    net.load_state_dict(sd)          # reference parameter names/shapes load unchanged
    return net.to(dev)


@pytest.mark.parametrize("tag", ["linear", "small"])
def test_net_forward_eval(dev, tag):
    import torch
    g = g5()
    net = _load_net(tag, dev).eval()
    with torch.no_grad():
        out, lat = net(torch.from_numpy(g[f"{tag}.x1"]).to(dev).view(-1, 1, g[f"{tag}.x1"].shape[1]))
    np.testing.assert_allclose(out.cpu().numpy(), g[f"{tag}.eval_out"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(lat.cpu().numpy(), g[f"{tag}.eval_latent"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("B", [7, 128, 64])
def test_info_nce_value_and_grad(dev, B):
    import torch
    from idelucs_amd.LossFunctions import info_nce_loss
    g = g5()
    h1 = torch.from_numpy(g[f"nce.B{B}.h1"]).to(dev).requires_grad_()
    h2 = torch.from_numpy(g[f"nce.B{B}.h2"]).to(dev).requires_grad_()
    l = info_nce_loss(h1, h2, 0.85)
    l.backward()
    assert abs(l.item() - float(g[f"nce.B{B}.loss"])) <= 1e-4 * abs(float(g[f"nce.B{B}.loss"]))
    np.testing.assert_allclose(h1.grad.cpu().numpy(), g[f"nce.B{B}.g1"], rtol=1e-3, atol=1e-7)
    np.testing.assert_allclose(h2.grad.cpu().numpy(), g[f"nce.B{B}.g2"], rtol=1e-3, atol=1e-7)


@pytest.mark.parametrize("B,C", [(7, 5), (128, 20), (64, 200)])
def test_iic_value_and_grad(dev, B, C):
    import torch
    from idelucs_amd.LossFunctions import IID_loss
    g = g5()
    a = torch.from_numpy(g[f"iic.B{B}.C{C}.a"]).to(dev).requires_grad_()
    b = torch.from_numpy(g[f"iic.B{B}.C{C}.b"]).to(dev).requires_grad_()
    l = IID_loss(torch.softmax(a, 1), torch.softmax(b, 1), lamb=2.8)
    l.backward()
    assert abs(l.item() - float(g[f"iic.B{B}.C{C}.loss"])) <= 1e-4 * abs(float(g[f"iic.B{B}.C{C}.loss"]))
    np.testing.assert_allclose(a.grad.cpu().numpy(), g[f"iic.B{B}.C{C}.ga"], rtol=1e-3, atol=2e-7)
    np.testing.assert_allclose(b.grad.cpu().numpy(), g[f"iic.B{B}.C{C}.gb"], rtol=1e-3, atol=2e-7)


def test_iic_analytic_kats(dev):
    import math
    import torch
    from idelucs_amd.LossFunctions import IID_loss
    g = g5()
    for C in (5, 20):
        u = torch.full((10 * C, C), 1.0 / C, device=dev)
        oh = torch.eye(C, device=dev).repeat(10, 1)
        assert abs(IID_loss(u, u, lamb=2.8).item() - float(g[f"iic.uniform.C{C}"])) < 1e-4
        assert abs(IID_loss(oh, oh, lamb=2.8).item() - float(g[f"iic.onehot.C{C}"])) < 1e-4
        assert abs(IID_loss(u, u, lamb=2.8).item() - (2 - 2 * 2.8) * math.log(C)) < 1e-4


@pytest.mark.parametrize("tag", ["linear", "small"])
def test_training_step_matches_reference(dev, tag):
    """Two optimizer steps with dropout disabled (eval mode, autograd on): loss, every parameter
    gradient and every parameter after RMSprop(lr=1e-3, weight_decay=0.01) vs the reference."""
    import torch
    from idelucs_amd.LossFunctions import IID_loss, info_nce_loss
    g = g5()
    net = _load_net(tag, dev).eval()
    opt = torch.optim.RMSprop(net.parameters(), lr=1e-3, weight_decay=0.01)
    x = torch.cat([torch.from_numpy(g[f"{tag}.x1"]), torch.from_numpy(g[f"{tag}.x2"])]).to(dev)
    b = x.shape[0] // 2
    for it in range(2):
        opt.zero_grad()
        z, h = net(x)                       # both views in one [2B, F] pass, as models.IID_model._step does
        loss = 0.75 * info_nce_loss(h[:b], h[b:], 0.85) + 0.25 * IID_loss(z[:b], z[b:], lamb=2.8)
        loss.backward()
        ref = float(g[f"{tag}.step{it}.loss"])
        assert abs(loss.item() - ref) <= 2e-4 * abs(ref), (it, loss.item(), ref)
        if it == 0:
            for n_, p in net.named_parameters():
                np.testing.assert_allclose(p.grad.cpu().numpy(), g[f"{tag}.step0.g.{n_}"], rtol=2e-3, atol=2e-6)
        if it == 0:
            # isolate the optimizer: step on the REFERENCE gradients (our own were just checked), so a
            # near-zero gradient (first RMSprop step = lr*g/(0.1|g|+eps), ill-conditioned at g~0) cannot flip
            for n_, p in net.named_parameters():
                p.grad.copy_(torch.from_numpy(g[f"{tag}.step0.g.{n_}"]).to(dev))
        opt.step()
        if it == 0:
            for n_, p in net.named_parameters():
                np.testing.assert_allclose(p.detach().cpu().numpy(), g[f"{tag}.step0.p.{n_}"], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("planes", ["1", "0"])
def test_end_to_end_quality_anchor(dev, monkeypatch, planes):
    """Statistical end-to-end anchor, pinned to the REFERENCE's own seed distribution (tests/golden/anchor_seeds.json, written by
    tests/golden/make_anchor_seeds.py from the imported reference on CPU): Influenza-A, k=6, C=5, 10 epochs, ONE voter.
    Round 3: 40 reference seeds instead of 10.  The reference at one voter is much noisier than its first ten seeds suggested --
    over 40 seeds ACC = 0.655 .. 0.994, mean 0.902, standard deviation 0.088, 15 % of the runs >= 0.99 (the first ten had mean
    0.948 and minimum 0.80: a lucky draw), which is why it ensembles five -- so bit parity of a run is not definable (dropout /
    shuffle streams differ on the GPU) and the bar is distributional, over 48 seeds of this implementation: mean within 3 standard errors of the difference (0.056) of
    the reference's, worst run no more than 0.05 under the reference's worst, best run >= 0.985, and the two empirical
    distributions no further apart than the two-sample Kolmogorov-Smirnov bound at alpha = 0.01.  (The 0.72 run round 2 reported
    is the reference's own behaviour -- its seeds 14 and 25 score 0.655 and 0.675 -- and not the fused step's: the same seeds
    through torch autograd, tools/acc_sweep.py / profiles/r03_acc_sweep.txt, give 0.72 .. 0.995, mean 0.926, the fused step
    0.77 .. 0.992, mean 0.902.)"""
    import pandas as pd
    import torch
    import idelucs_amd
    from idelucs_amd import models
    # (VERDICT r5 #2d) the anchor holds for BOTH forms of the step's two big products -- the fp16 planes (default) and the fp32 tiles -- and the two
    # samples of 48 runs are themselves no further apart than the same Kolmogorov-Smirnov bound (checked when the second form has run)
    monkeypatch.setenv("IDELUCS_PLANES", planes)
    anchor = json.load(open(os.path.join(GOLDEN, "anchor.json")))
    ref = np.array([r["acc"] for r in json.load(open(os.path.join(GOLDEN, "anchor_seeds.json")))["single"]])
    assert len(ref) >= 40 and abs(ref[0] - anchor["acc"]) < 1e-12          # seed 0 of the sweep is the round-1 anchor run
    df = pd.read_csv(os.path.join(DATA, "Influenza-A_GT.tsv"), sep="\t")
    u = {v: i for i, v in enumerate(sorted(set(df.cluster_id)))}
    gt = np.array([u[v] for v in df.cluster_id])
    # API-level run (shapes / dtypes of the drop-in entry point)
    y, lat = idelucs_amd.iDeLUCS_cluster(os.path.join(DATA, "Influenza-A.fas"), n_clusters=5, n_epochs=10, n_mimics=3,
                                         batch_sz=512, k=6, weight=0.25, n_voters=1).fit_predict(None)
    assert y.dtype == np.int64 and y.shape == (949,) and lat.dtype == np.float64 and lat.shape == tuple(anchor["latent_shape"])
    accs = [idelucs_amd.cluster_acc(gt, y)[1]]
    for seed in range(1, 48):
        m = models.IID_model({'sequence_file': os.path.join(DATA, "Influenza-A.fas"), 'GT_file': None, 'n_clusters': 5, 'k': 6,
                              'model_size': 'linear', 'n_mimics': 3, 'batch_sz': 512, 'optimizer': 'RMSprop', 'lambda': 2.8,
                              'lr': 1e-3, 'weight': 0.25, 'scheduler': None, 'n_epochs': 10, 'n_voters': 1, 'seed': seed})
        m.build_dataloader()
        m.begin_voter(0)
        for _ in range(10):
            m.contrastive_training_epoch()
        accs.append(idelucs_amd.cluster_acc(gt, m.predict()[0])[1])
    accs = np.array(accs)
    grid = np.sort(np.concatenate([accs, ref]))
    ks = np.max(np.abs(np.searchsorted(np.sort(accs), grid, side="right") / len(accs) - np.searchsorted(np.sort(ref), grid, side="right") / len(ref)))
    ks_bound = 1.63 * np.sqrt((len(accs) + len(ref)) / (len(accs) * len(ref)))
    print(f"ACC over {len(accs)} seeds: mean {accs.mean():.4f} std {accs.std(ddof=1):.4f} min {accs.min():.4f} max {accs.max():.4f} | reference over "
          f"{len(ref)}: mean {ref.mean():.4f} std {ref.std(ddof=1):.4f} min {ref.min():.4f} max {ref.max():.4f} | KS {ks:.3f} (bound {ks_bound:.3f})")
    # the mean bar: 3 standard errors of the difference of two means of runs this noisy (sd 0.085 each: 0.056 with 48 + 40 seeds).
    # Round 3's fixed 0.03 was 1.6 standard errors wide -- a legitimately different software stack could trip it;
    # 64 seeds each of the fused and the autograd step differ by 0.012 +- 0.015 (profiles/r04_acc_sweep.txt)
    se_diff = float(np.sqrt(accs.var(ddof=1) / len(accs) + ref.var(ddof=1) / len(ref)))
    assert abs(accs.mean() - ref.mean()) <= 3.0 * se_diff, (accs.mean(), ref.mean(), se_diff)
    assert accs.max() >= 0.985 and accs.min() >= ref.min() - 0.05, (accs.max(), accs.min(), ref.min())
    assert ks <= ks_bound, (ks, ks_bound)
    _ANCHOR_SAMPLES[planes] = accs
    if len(_ANCHOR_SAMPLES) == 2:
        a, b = np.sort(_ANCHOR_SAMPLES["1"]), np.sort(_ANCHOR_SAMPLES["0"])
        grid = np.sort(np.concatenate([a, b]))
        ks2 = np.max(np.abs(np.searchsorted(a, grid, side="right") / len(a) - np.searchsorted(b, grid, side="right") / len(b)))
        print(f"planes vs fp32 tiles over {len(a)} seeds each: means {a.mean():.4f} / {b.mean():.4f}, KS {ks2:.3f} (bound {1.63 * np.sqrt(2.0 / len(a)):.3f})")
        assert ks2 <= 1.63 * np.sqrt(2.0 / len(a)), ks2


_ANCHOR_SAMPLES = {}


# ------------------------------------------------------------------------------------------------
# the fused explicit step (idelucs_amd/fused.py + csrc/train_step.hip)
# ------------------------------------------------------------------------------------------------
def _fused_trainer(dev):
    from idelucs_amd.fused import FusedLinearTrainer
    net = _load_net("linear", dev)
    return net, FusedLinearTrainer(net, lr=1e-3, weight=0.25, lamb=2.8, seed=5)


def test_fused_step_matches_reference(dev):
    """Explicit forward/backward + fused RMSprop, dropout off, vs the reference's autograd step."""
    import torch
    g = g5()
    net, tr = _fused_trainer(dev)
    bf = tr.buffers(18)
    bf.x.copy_(torch.cat([torch.from_numpy(g["linear.x1"]), torch.from_numpy(g["linear.x2"])]).to(dev))
    tr.step_on_batch(bf, train=False)
    torch.cuda.synchronize()
    ref = float(g["linear.step0.loss"])
    assert abs(tr.out[0].item() - ref) <= 2e-4 * abs(ref), (tr.out[0].item(), ref)
    names = ["layers.0.weight", "layers.0.bias", "layers.3.weight", "layers.3.bias", "classifier.2.weight", "classifier.2.bias"]
    for i, n_ in enumerate(names):
        np.testing.assert_allclose(tr.gradient(i).cpu().numpy(), g[f"linear.step0.g.{n_}"], rtol=2e-3, atol=2e-6, err_msg=n_)
    # parameters after the update: all but the near-zero-gradient elements (first RMSprop step is lr*g/(0.1|g|+eps))
    for n_, p in zip(names, tr.params):
        got, want = p.detach().cpu().numpy(), g[f"linear.step0.p.{n_}"]
        bad = ~np.isclose(got, want, rtol=1e-3, atol=1e-6)
        assert bad.mean() < 2e-3, (n_, bad.mean())
    # second step: loss after one update
    tr.step_on_batch(bf, train=False)
    ref1 = float(g["linear.step1.loss"])
    assert abs(tr.out[0].item() - ref1) <= 5e-4 * abs(ref1), (tr.out[0].item(), ref1)
    assert abs(tr.out[1].item() - (tr.out[0].item() + ref)) < 1e-3          # running sum
    assert tr.ctl.tolist() == [2, 0]


def test_fused_rmsprop_kernel_exact(dev):
    """idl_rmsprop_step on the reference gradients reproduces torch.optim.RMSprop's update."""
    import torch
    g = g5()
    net, tr = _fused_trainer(dev)
    names = ["layers.0.weight", "layers.0.bias", "layers.3.weight", "layers.3.bias", "classifier.2.weight", "classifier.2.bias"]
    for n_, gr, parts in zip(names, tr.grads, tr.parts):
        ref = torch.from_numpy(g[f"linear.step0.g.{n_}"]).to(dev)
        if parts > 1:                       # spread a bias gradient over the stacked partials
            gr.zero_(); gr[0].copy_(ref * 0.25); gr[parts - 1].copy_(ref * 0.75)
        else:
            gr.copy_(ref)
    from idelucs_amd import _lib
    from idelucs_amd.fused import _p, _stream
    _lib.check(_lib.lib.idl_rmsprop_step(6, tr._pp, tr._gp, tr._parts, tr._vp, tr._sz, _p(tr.hyper), _p(tr.ctl), 7, None, 0, 0.0, 0.0, None,
                                         _stream()))
    for n_, p in zip(names, tr.params):
        np.testing.assert_allclose(p.detach().cpu().numpy(), g[f"linear.step0.p.{n_}"], rtol=1e-5, atol=1e-7, err_msg=n_)
    assert tr.ctl.tolist() == [1, 7]


@pytest.mark.parametrize("B,C", [(7, 5), (128, 20), (64, 200)])
def test_fused_loss_kernels_vs_reference(dev, B, C):
    """idl_nce_rows / idl_iic_core / idl_head_bwd pieces against the reference's loss values and input gradients."""
    import torch
    from idelucs_amd import _lib
    from idelucs_amd.fused import _p, _stream, EPS
    g = g5()
    L = _lib.lib
    m = 2 * B
    # InfoNCE: loss and d/d(latent) through the normalisation
    h = torch.cat([torch.from_numpy(g[f"nce.B{B}.h1"]), torch.from_numpy(g[f"nce.B{B}.h2"])]).to(dev)
    nrm = h.norm(dim=1, keepdim=True).clamp_min(1e-12)
    f = (h / nrm).contiguous()
    S = (f @ f.t()).contiguous()
    lse = torch.empty(m, device=dev); rows = torch.empty(m, device=dev)
    _lib.check(L.idl_nce_rows(_p(S), m, 0.85, _p(lse), _p(rows), _stream()))
    assert abs(rows.mean().item() - float(g[f"nce.B{B}.loss"])) <= 1e-4 * abs(float(g[f"nce.B{B}.loss"]))
    G = S @ f
    pos = (torch.arange(m, device=dev) + B) % m
    df = (G - 2 * f[pos]) / (m * 0.85)
    dh = (df - f * (f * df).sum(1, keepdim=True)) / nrm
    np.testing.assert_allclose(dh[:B].cpu().numpy(), g[f"nce.B{B}.g1"], rtol=1e-3, atol=2e-7)
    np.testing.assert_allclose(dh[B:].cpu().numpy(), g[f"nce.B{B}.g2"], rtol=1e-3, atol=2e-7)
    # IIC: loss and dP0 -> dz -> d(logits) (softmax backward done here in torch)
    a = torch.from_numpy(g[f"iic.B{B}.C{C}.a"]).to(dev); b = torch.from_numpy(g[f"iic.B{B}.C{C}.b"]).to(dev)
    z1, z2 = torch.softmax(a, 1), torch.softmax(b, 1)
    P0 = (z1.t() @ z2).contiguous()
    scratch = torch.zeros(C * C + 2 * C, device=dev); out = torch.zeros(4, device=dev)
    _lib.check(L.idl_iic_core(_p(P0), C, 2.8, EPS, 1.0, _p(scratch), _p(out), _stream()))
    ref = float(g[f"iic.B{B}.C{C}.loss"])
    assert abs(out[3].item() - ref) <= 1e-4 * abs(ref), (out[3].item(), ref)
    dz1, dz2 = z2 @ P0.t(), z1 @ P0
    da = z1 * (dz1 - (dz1 * z1).sum(1, keepdim=True)); db = z2 * (dz2 - (dz2 * z2).sum(1, keepdim=True))
    np.testing.assert_allclose(da.cpu().numpy(), g[f"iic.B{B}.C{C}.ga"], rtol=1e-3, atol=2e-7)
    np.testing.assert_allclose(db.cpu().numpy(), g[f"iic.B{B}.C{C}.gb"], rtol=1e-3, atol=2e-7)


def test_fused_dropout_statistics(dev):
    import torch
    from idelucs_amd import _lib
    from idelucs_amd.fused import _p, _stream
    L = _lib.lib
    ctl = torch.zeros(2, dtype=torch.int64, device=dev)
    a = torch.ones(1024 * 512, device=dev)
    _lib.check(L.idl_relu_dropout_fwd(_p(a), a.numel(), 1, 9, _p(ctl), 1, _stream()))
    assert set(a.unique().tolist()) == {0.0, 2.0} and abs(a.mean().item() - 1.0) < 0.01
    b = torch.ones(1024 * 512, device=dev)
    _lib.check(L.idl_relu_dropout_fwd(_p(b), b.numel(), 1, 9, _p(ctl), 1, _stream()))
    assert torch.equal(a, b)                                  # same (seed, layer, step) -> same mask
    ctl[0] = 1
    c = torch.ones(1024 * 512, device=dev)
    _lib.check(L.idl_relu_dropout_fwd(_p(c), c.numel(), 1, 9, _p(ctl), 1, _stream()))
    assert 0.45 < (a != c).float().mean().item() < 0.55       # next step -> independent mask
    d = -torch.ones(4096, device=dev)
    _lib.check(L.idl_relu_dropout_fwd(_p(d), d.numel(), 0, 9, _p(ctl), 1, _stream()))
    assert torch.all(d == 0)                                  # eval: plain ReLU


def test_fused_graph_replay_equals_eager(dev):
    """An epoch replayed from the captured HIP graph gives the same parameters as the same epoch
    launched eagerly (same permutation, dropout stream and initial state)."""
    import copy
    import torch
    from idelucs_amd import utils as U
    from idelucs_amd.PytorchUtils import NetLinear
    from idelucs_amd.fused import FusedLinearTrainer
    from idelucs_amd import models
    torch.manual_seed(3)
    P, n, F, B = 4, 700, 256, 64
    feats = (torch.rand((P, n, F), device=dev) * 1e-2).contiguous()
    mean, scale = U.col_stats(feats[0])
    store = U.FeatureStore(None, None, feats, mean, scale, 4, False)
    net0 = NetLinear(F, 6).to(dev); net0.apply(models.weights_init)
    results = []
    for use_graph in (False, True):
        net = copy.deepcopy(net0)
        tr = FusedLinearTrainer(net, lr=1e-3, weight=0.25, lamb=2.8, seed=11)
        gen = torch.Generator(device=dev); gen.manual_seed(77)
        total, nb = tr.run_epoch(store, B, use_graph=use_graph, generator=gen)
        torch.cuda.synchronize()
        assert nb == (3 * n + B - 1) // B and tr.ctl.tolist() == [nb, 3 * n]
        results.append(([p.detach().clone() for p in tr.params], total.item()))
    for a, b in zip(results[0][0], results[1][0]):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
    assert abs(results[0][1] - results[1][1]) <= 1e-4 * abs(results[0][1])
    assert np.isfinite(results[0][1])


def _cfg2_store_and_net(dev, n, seed=3, C=20):
    """A feature store of cfg2's row shape (4 views x n x 4096, frequency-like rows) and a NetLinear(4096, C)."""
    import torch
    from idelucs_amd import utils as U, models
    from idelucs_amd.PytorchUtils import NetLinear
    g = torch.Generator(device=dev); g.manual_seed(seed)
    P, F = 4, 4096
    base = torch.rand((1, n, F), device=dev, generator=g) + 0.5
    feats = base * (1.0 + 0.05 * torch.randn((P, n, F), device=dev, generator=g))       # mimic views = perturbed copies
    feats = (feats / feats.sum(2, keepdim=True)).contiguous()
    mean, scale = U.col_stats(feats[0])
    store = U.FeatureStore(None, None, feats, mean, scale, 6, False)
    torch.manual_seed(seed)
    net = NetLinear(F, C).to(dev); net.apply(models.weights_init)
    return store, net


@pytest.mark.parametrize("C", [20, 200])
def test_default_fused_step_at_cfg2_shape_vs_autograd(dev, C):
    """C = 20: the launch sequence bench.py times -- FusedLinearTrainer._full_step(pipelined=True) with every default (transposed
    layer-1 product -> mid_fwd_gather -> nce_fused_iic_z -> mid_bwd_gather -> wgrad_rmsprop_step: dW1 tiles at the head of the
    optimizer launch) at cfg2's
    shape m=1024, F=4096, C=20 -- against torch autograd over idelucs_amd.LossFunctions (pinned to the reference goldens
    above), dropout off: loss rel 2e-4, the six gradients rel 2e-3, parameters after RMSprop rel 1e-5 on identical
    gradients; and the batch the step assembled for the NEXT step is the gather of the next 512 pairs.
    C = 200: the fine-grained mode's sequence (cfg5, --n_clusters 0): layer-1 product -> mid_fwd_gather carrying ALL of the next
    batch's tiles -> joint GEMM + idl_iic_core -> idl_nce_fused -> z dP0 GEMM -> head_bwd_dz -> dW3 / dr1 GEMMs -> bias_grads ->
    wgrad_rmsprop_step."""
    import copy
    import torch
    from idelucs_amd.fused import FusedLinearTrainer
    from idelucs_amd.LossFunctions import IID_loss, info_nce_loss
    store, net = _cfg2_store_and_net(dev, 2000, C=C)
    ref_net = copy.deepcopy(net)
    tr = FusedLinearTrainer(net, lr=1e-3, weight=0.25, lamb=2.8, seed=5)
    if C == 20:
        assert tr._early_gather and tr._early_split and tr._transposed_l1 and tr._dw2_inlaunch and tr._mid_fused and tr._dw3_partial \
            and tr._joint_inlaunch and not tr._nce_bwd_fused and tr._wgrad_fused and not tr._wgrad_own_launch, "not the default launch sequence"
    else:
        assert tr._early_fwd and not tr._early_gather and tr._dw2_inlaunch and tr._mid_fused and not tr._dw3_partial
    tr._keep_w1_grad = True                                     # the tiles also write dW1 out (the timed step never does)
    B = 512
    gen = torch.Generator(device=dev); gen.manual_seed(9)
    tr._perm = torch.randperm(store.n_pairs, device=dev, generator=gen)
    tr.ctl[1] = 0; tr.out[1] = 0.0
    bf = tr.buffers(2 * B)
    assert bf.nce_fused
    tr._gather(store, bf)                                       # prologue: batch 0 into bf.xs[0]
    x = bf.xs[0].clone()
    tr._full_step(store, bf, train=False, pipelined=True, xi=0)
    torch.cuda.synchronize()
    # --- autograd reference on the same batch, same initial parameters
    ref_net.eval()
    z, h = ref_net(x)
    loss = 0.75 * info_nce_loss(h[:B], h[B:], 0.85) + 0.25 * IID_loss(z[:B], z[B:], lamb=2.8)
    loss.backward()
    assert abs(tr.out[0].item() - loss.item()) <= 2e-4 * abs(loss.item()), (tr.out[0].item(), loss.item())
    ref_params = [ref_net.layers[0].weight, ref_net.layers[0].bias, ref_net.layers[3].weight, ref_net.layers[3].bias,
                  ref_net.classifier[2].weight, ref_net.classifier[2].bias]
    for i, p in enumerate(ref_params):
        got, want = tr.gradient(i), p.grad
        err = (got - want).abs().max().item()
        assert err <= 2e-3 * want.abs().max().item() + 1e-9, (i, err, want.abs().max().item())
        assert torch.allclose(got, want, rtol=2e-3, atol=2e-3 * want.abs().mean().item()), i
    # --- RMSprop on IDENTICAL gradients (the trainer's own): torch.optim vs the fused optimizer launch
    for i, p in enumerate(ref_params):
        p.grad.copy_(tr.gradient(i))
    torch.optim.RMSprop(ref_net.parameters(), lr=1e-3, weight_decay=0.01).step()
    for i, (p, q) in enumerate(zip(tr.params, ref_params)):
        bad = ~torch.isclose(p.detach(), q.detach(), rtol=1e-5, atol=1e-7)
        # a first RMSprop step is lr * g / (0.1 |g| + eps): where the bias partials are summed in another order a near-zero
        # gradient may flip; everything else must agree to 1e-5
        assert bad.float().mean().item() < (2e-3 if p.dim() == 1 else 1e-5), (i, bad.float().mean().item())
    assert tr.ctl.tolist() == [1, B]
    # --- the next batch, assembled by spare workgroups of the two middle launches into the other x buffer
    want_next = store.gather_pairs(tr._perm[B:2 * B])
    pb = getattr(bf, "_planes", None)
    if pb is not None and not pb["x32"][1]:     # the two-plane step form (the default at this shape): the next batch exists as its planes only
        from idelucs_amd import _lib
        k = int(_lib.lib.idl_planes_exponent(0))
        back = (pb["xh"][1].view(torch.float16).double() + pb["xl"][1].view(torch.float16).double()) * 2.0 ** -k
        err = (back - want_next.double()).abs()
        assert pb["valid"][1] and bool((err <= torch.clamp(want_next.double().abs() * 2.0 ** -21, min=2.0 ** (-25 - k))).all())
    else:
        assert torch.equal(bf.xs[1], want_next)


@pytest.mark.parametrize("C,n", [(20, 1500), (200, 1500), (20, 4200), (200, 4200)])
def test_graph_replay_equals_eager_at_cfg2_shape(dev, C, n):
    """A whole epoch at cfg2's step shape (F=4096, batch 512; n = 1500: 8 full batches + a partial one, replayed two steps per graph;
    n = 4200: 24 full batches + a partial one, eight steps per graph; dropout ON) replayed from the captured HIP graph vs launched
    eagerly: same permutation, same dropout stream, same start.  C = 200: the fine-grained mode's launch sequence."""
    import copy
    import torch
    from idelucs_amd.fused import FusedLinearTrainer
    store, net0 = _cfg2_store_and_net(dev, n, seed=4, C=C)
    B = 512
    nb_want = (store.n_pairs + B - 1) // B
    results = []
    for use_graph in (False, True):
        net = copy.deepcopy(net0)
        tr = FusedLinearTrainer(net, lr=1e-3, weight=0.25, lamb=2.8, seed=11)
        gen = torch.Generator(device=dev); gen.manual_seed(77)
        total, nb = tr.run_epoch(store, B, use_graph=use_graph, generator=gen)
        torch.cuda.synchronize()
        assert nb == nb_want and tr.ctl.tolist() == [nb_want, store.n_pairs]
        if use_graph:
            assert len(tr._graphs) == 1, "the epoch did not go through a captured graph"
            assert next(iter(tr._graphs))[-1] == (2 if n == 1500 else 8)
        results.append(([p.detach().clone() for p in tr.params], total.item()))
    for a, b in zip(results[0][0], results[1][0]):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
    assert np.isfinite(results[0][1]) and abs(results[0][1] - results[1][1]) <= 1e-4 * abs(results[0][1])
    # a second epoch on the same store replays the SAME graph (nothing re-captured) ...
    g_before = next(iter(tr._graphs.values()))
    tr.run_epoch(store, B, generator=gen)
    assert next(iter(tr._graphs.values())) is g_before
    # ... also after the scaler was refitted IN PLACE (bench.py / a new voter): every baked address is unchanged
    from idelucs_amd import utils as U
    U.col_stats(store.feats[1], out=(store.mean, store.scale)); store.refresh()
    tr.run_epoch(store, B, generator=gen)
    torch.cuda.synchronize()
    assert next(iter(tr._graphs.values())) is g_before and all(torch.isfinite(p).all() for p in tr.params)


@pytest.mark.parametrize("n,use_graph", [(1500, False), (1500, True), (4200, True)])
def test_batched_voters_step_like_single_voters(dev, monkeypatch, n, use_graph):
    """fused.BatchedLinearTrainer (three voters in lockstep: batched GEMMs + recorded launches with the voter index in the grid)
    against three FusedLinearTrainer runs of the same voters: same initial weights, same permutations (one generator per voter),
    same dropout streams -- after a whole epoch (8 or 24 full batches + a partial one) the parameters agree to a few 1e-4 of their
    scale and the losses to 1e-4.  That needs the library GEMMs of both sides to be the heuristic's: the first RMSprop steps from a
    fresh state move every weight by ~10 lr in the direction of its gradient's sign, and a rounding difference in the layer-1
    product (a shipped TunableOp solution on one side, the strided-batched kernel on the other) grows by 3-40 x per step (3.8e-5 of
    W1 after one step, 1.9e-2 after four: measured) -- two valid runs, no longer comparable element by element.  TunableOp is
    therefore off inside this test."""
    import copy
    import torch
    import torch.cuda.tunable as tunable
    from idelucs_amd.fused import FusedLinearTrainer, BatchedLinearTrainer
    was_on = tunable.is_enabled()
    tunable.enable(False)
    monkeypatch.setenv("IDELUCS_PLANES", "0")       # (... and the single voters to form their products in fp32 as the batched GEMMs do)
    try:
        _batched_like_single(dev, n, use_graph, copy, torch, FusedLinearTrainer, BatchedLinearTrainer)
    finally:
        tunable.enable(was_on)


def _batched_like_single(dev, n, use_graph, copy, torch, FusedLinearTrainer, BatchedLinearTrainer, L=3):
    store, net0 = _cfg2_store_and_net(dev, n, seed=4)
    B = 512
    nets_a, nets_b = [], []
    for l in range(L):
        net = copy.deepcopy(net0)
        with torch.no_grad():
            for p_ in net.parameters():                       # different voters: different weights
                p_.mul_(1.0 + 0.05 * l)
        nets_a.append(net)
        nets_b.append(copy.deepcopy(net))
    single = []
    for l, net in enumerate(nets_a):
        tr = FusedLinearTrainer(net, lr=1e-3, weight=0.25, lamb=2.8, seed=11)
        tr.begin_voter(l)
        gen = torch.Generator(device=dev); gen.manual_seed(100 + l)
        total, nb = tr.run_epoch(store, B, use_graph=use_graph, generator=gen)
        single.append((total.item(), nb, [p_.detach().clone() for p_ in tr.params], tr.ctl.tolist()))
    bt = BatchedLinearTrainer(nets_b, lr=1e-3, weight=0.25, lamb=2.8, seed=11)
    gens = []
    for l, tr in enumerate(bt.trainers):
        tr.begin_voter(l)
        g = torch.Generator(device=dev); g.manual_seed(100 + l)
        gens.append(g)
    res = bt.run_epoch(store, B, gens, use_graph=use_graph)
    torch.cuda.synchronize()
    if use_graph:
        assert len(bt._graphs) == 1, "the epoch did not go through a captured graph"
    for l, ((total, nb), tr) in enumerate(zip(res, bt.trainers)):
        want_total, want_nb, want_params, want_ctl = single[l]
        assert nb == want_nb and tr.ctl.tolist() == want_ctl
        assert abs(total.item() - want_total) <= 2e-4 * abs(want_total), (l, total.item(), want_total)
        for i, (a, b) in enumerate(zip(tr.params, want_params)):
            err = (a.detach() - b).abs().max().item()
            assert err <= 5e-4 * b.abs().max().item() + 1e-7, (l, i, err, b.abs().max().item())
        assert nets_b[l].layers[0].weight.data_ptr() == bt.W1s[l].data_ptr()          # the network's parameter IS the stacked slice
    # voters are different runs
    assert not torch.allclose(bt.trainers[0].params[2], bt.trainers[1].params[2])
    # a second epoch replays the same graph
    if use_graph:
        g_before = next(iter(bt._graphs.values()))
        bt.run_epoch(store, B, gens)
        torch.cuda.synchronize()
        assert next(iter(bt._graphs.values())) is g_before and all(torch.isfinite(p_).all() for t in bt.trainers for p_ in t.params)
    return bt


def test_plan_records_are_checked(dev):
    """idl_plan_begin / idl_plan_end: a launcher that cannot be recorded, or none at all, is an error; records of different
    launch shapes cannot be launched together."""
    import ctypes
    import torch
    from idelucs_amd import _lib
    L = _lib.lib
    nb = int(L.idl_plan_bytes())
    blob = torch.zeros(nb, dtype=torch.uint8)
    _lib.check(L.idl_plan_begin(ctypes.c_void_p(blob.data_ptr())))
    with pytest.raises(ValueError):
        _lib.check(L.idl_plan_end())                      # nothing was recorded
    two = torch.zeros((2, nb), dtype=torch.uint8)
    two[1, 8] = 1                                         # another grid
    with pytest.raises(ValueError):
        _lib.check(L.idl_plan_launch(ctypes.c_void_p(two.data_ptr()), ctypes.c_void_p(two.to(dev).data_ptr()), 2, None))


# ------------------------------------------------------------------------------------------------
# the other configurations of the reference surface
# ------------------------------------------------------------------------------------------------
def _args(**kw):
    a = {'sequence_file': os.path.join(DATA, "Influenza-A.fas"), 'GT_file': None, 'n_clusters': 5, 'k': 4, 'model_size': 'linear',
         'n_mimics': 3, 'batch_sz': 256, 'optimizer': 'RMSprop', 'lambda': 2.8, 'lr': 1e-3, 'weight': 0.25, 'scheduler': None,
         'n_epochs': 3, 'n_voters': 1}
    a.update(kw)
    return a


@pytest.mark.parametrize("kw", [dict(model_size='small', k=4), dict(model_size='small', k=5), dict(optimizer='SGD'), dict(optimizer='Adam'),
                                dict(scheduler='Plateau'), dict(scheduler='Triangle'), dict(n_mimics=1), dict(n_mimics=5, k=5),
                                dict(n_clusters=200, k=4)])
def test_all_reference_configurations_train_and_predict(dev, kw):
    """model_size small (canonical k-mers + myNet, reference models.py:59-66), the three optimizers
    (:87-94), both schedulers (:96-99), n_mimics < 2 (still two mimic passes, utils.py:336-344) and the
    n_clusters=0 head width (200): every one trains on the GPU, loss decreases, outputs have the
    reference's shapes/dtypes."""
    import torch
    from idelucs_amd import models
    torch.manual_seed(1)
    m = models.IID_model(_args(**kw))
    m.build_dataloader()
    assert m.store.n_pairs == 949 * max(m.n_mimics, 2)
    if kw.get('model_size') == 'small':
        assert m.store.f == {4: 136, 5: 512}[m.k] and m._use_fused is False
    losses = [m.contrastive_training_epoch() for _ in range(3)]
    assert all(np.isfinite(l) for l in losses)
    if kw.get('scheduler') != 'Triangle':          # CyclicLR ramps the LR from 1e-3 towards 0.1 (models.py:99): the loss rises
        assert losses[-1] < losses[0]
    else:
        assert m.optimizer.param_groups[0]['lr'] > 1e-3 and losses[-1] != losses[0]
    y, p, lat = m.predict()
    assert y.dtype == np.int64 and y.shape == (949,) and p.dtype == np.float64 and lat.shape == (949, 64) and lat.dtype == np.float64
    probs = m.calculate_probs()
    assert probs.shape == (949, m.n_clusters) and np.allclose(probs.sum(1), 1.0, atol=1e-5)


def test_invalid_configurations_raise_like_the_reference(dev):
    from idelucs_amd import models
    with pytest.raises(ValueError, match="Invalid Model Type"):
        models.IID_model(_args(model_size='conv'))
    with pytest.raises(ValueError, match="Optimizer not supported"):
        models.IID_model(_args(optimizer='LBFGS'))


def test_single_batch_epoch_returns_inf_like_the_reference(dev, tmp_path):
    """models.py:135 divides the summed loss by the LAST batch index: one batch -> division by zero -> inf."""
    from idelucs_amd import models
    m = models.IID_model(_args(sequence_file=os.path.join(DATA, "influenza_64.fas"), batch_sz=512))
    m.build_dataloader()
    assert m.store.n_pairs == 192
    assert np.isinf(m.contrastive_training_epoch())


@pytest.mark.parametrize("B", [16, 64, 512, 480])
def test_fused_infonce_kernels_vs_torch(dev, B):
    """idl_nce_fused (fp32 MFMA, S never materialised) against the mask-free torch formulation, which the
    tests above tie to the reference: loss rows, lse and (E + E^T) f."""
    import torch
    from idelucs_amd import _lib
    from idelucs_amd.fused import _p, _stream
    L = _lib.lib
    m = 2 * B
    torch.manual_seed(B)
    h = torch.randn(m, 64, device=dev)
    h[B:] = h[:B] + 0.3 * torch.randn(B, 64, device=dev)           # positives are correlated, like real pairs
    f = torch.nn.functional.normalize(h, dim=1).contiguous()
    T = 0.85
    S = (f.double() @ f.double().t()) / T
    eye = torch.eye(m, dtype=torch.bool, device=dev)
    lse_ref = torch.logsumexp(S.masked_fill(eye, float("-inf")), dim=1)
    pos = (torch.arange(m, device=dev) + B) % m
    loss_ref = lse_ref - S[torch.arange(m, device=dev), pos]
    E = torch.exp(S - lse_ref[:, None]).masked_fill(eye, 0.0)
    G_ref = (E + E.t()) @ f.double()
    parts = L.idl_nce_fused_parts()
    ws = torch.empty(L.idl_nce_fused_workspace(m) // 4, device=dev)
    lse = torch.empty(m, device=dev); rows = torch.empty(m, device=dev); G = torch.empty((parts, m, 64), device=dev)
    _lib.check(L.idl_nce_fused(_p(f), m, T, _p(lse), _p(rows), _p(G), _p(ws), _stream()))
    np.testing.assert_allclose(lse.cpu().numpy(), lse_ref.float().cpu().numpy(), rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(rows.cpu().numpy(), loss_ref.float().cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(G.sum(0).cpu().numpy(), G_ref.float().cpu().numpy(), rtol=1e-4, atol=2e-6)
    assert L.idl_nce_fused_workspace(574) == -1 and L.idl_nce_fused_workspace(4096) == -1      # fallback shapes


@pytest.mark.parametrize("m,C,train", [(16, 5, 0), (1024, 20, 0), (960, 20, 1), (64, 200, 1)])
def test_fused_mid_forward_equals_separate_kernels(dev, m, C, train):
    """idl_mid_fwd (ReLU/Dropout + Linear(512,64) on fp32 MFMA + head) == idl_relu_dropout_fwd + addmm + idl_head_fwd,
    including the dropout masks (same Philox streams)."""
    import torch
    from idelucs_amd import _lib
    from idelucs_amd.fused import _p, _stream
    L = _lib.lib
    torch.manual_seed(m + C)
    a1 = torch.randn(m, 512, device=dev)
    W2 = torch.randn(64, 512, device=dev) * 0.06; b2 = torch.randn(64, device=dev) * 0.1
    W3 = torch.randn(C, 64, device=dev) * 0.2; b3 = torch.randn(C, device=dev) * 0.1
    ctl = torch.tensor([3, 0], dtype=torch.int64, device=dev)
    seed = 12345
    def outs():
        return [torch.empty(m, 64, device=dev), torch.empty(m, device=dev), torch.empty(m, 64, device=dev), torch.empty(m, C, device=dev)]
    # separate kernels
    r1a = a1.clone()
    _lib.check(L.idl_relu_dropout_fwd(_p(r1a), r1a.numel(), train, seed, _p(ctl), 1, _stream()))
    lat = torch.addmm(b2, r1a, W2.t())
    fa, ia, r2a, za = outs()
    _lib.check(L.idl_head_fwd(_p(lat), _p(W3), _p(b3), m, C, train, seed, _p(ctl), _p(fa), _p(ia), _p(r2a), _p(za), _stream()))
    # fused
    r1b = a1.clone()
    fb, ib, r2b, zb = outs()
    _lib.check(L.idl_mid_fwd(_p(r1b), _p(W2), _p(b2), _p(W3), _p(b3), m, C, train, seed, _p(ctl), _p(fb), _p(ib), _p(r2b), _p(zb), _stream()))
    assert torch.equal(r1a, r1b)                                        # same ReLU/Dropout result, bit for bit
    assert torch.equal(r2a != 0, r2b != 0) or (r2a != 0).ne(r2b != 0).float().mean() < 1e-3   # same latent dropout mask (sign flips at ~0 aside)
    for x, y, tol in ((fa, fb, 2e-5), (ia, ib, 2e-5), (r2a, r2b, 2e-4), (za, zb, 2e-5)):
        np.testing.assert_allclose(y.cpu().numpy(), x.cpu().numpy(), rtol=2e-4, atol=tol)


@pytest.mark.gpu
@pytest.mark.parametrize("m,C,train,dw3", [(32, 5, 0, 1), (1024, 20, 1, 1), (960, 20, 1, 1), (14, 7, 1, 1), (64, 200, 1, 0), (2080, 20, 1, 1)])
def test_fused_mid_backward_equals_separate_kernels(dev, m, C, train, dw3):
    """idl_mid_bwd (head backward + dr1 = dlat W2 on fp32 MFMA + ReLU/Dropout backward + all partial bias sums + dW3 partials)
    == idl_head_bwd + torch.mm + idl_bias_grads, on ragged / empty row chunks too."""
    import torch
    from idelucs_amd import _lib
    from idelucs_amd.fused import _p, _stream
    L = _lib.lib
    parts = L.idl_col_sum_parts()
    torch.manual_seed(m * 7 + C)
    z = torch.softmax(torch.randn(m, C, device=dev), 1)
    r2 = torch.relu(torch.randn(m, 64, device=dev)) * (torch.rand(m, 64, device=dev) > 0.5) * 2
    lat = torch.randn(m, 64, device=dev)
    nrm = lat.norm(dim=1).clamp_min(1e-12)
    f = lat / nrm[:, None]; inv = 1.0 / nrm
    gp = 4
    G = torch.randn(gp, m, 64, device=dev) * 0.3
    dP0 = torch.randn(C, C, device=dev) * 0.1; dP0 = dP0 + dP0.t()
    W3 = torch.randn(C, 64, device=dev) * 0.2
    W2 = torch.randn(64, 512, device=dev) * 0.06
    act1 = torch.relu(torch.randn(m, 512, device=dev)) * (torch.rand(m, 512, device=dev) > 0.5)
    coef = 0.75 / (m * 0.85)
    # separate kernels
    dlg_a = torch.empty(m, C, device=dev); dlat_a = torch.empty(m, 64, device=dev)
    _lib.check(L.idl_head_bwd(_p(z), _p(r2), _p(f), _p(inv), _p(G), gp, _p(dP0), _p(W3), m, C, train, coef, _p(dlg_a), _p(dlat_a), _stream()))
    dr1_a = dlat_a @ W2
    p1a = torch.zeros(parts, 512, device=dev); p2a = torch.zeros(parts, 64, device=dev); p3a = torch.zeros(parts, C, device=dev)
    w3a = torch.zeros(parts, C, 64, device=dev)
    ctl_a = torch.tensor([3, 100], dtype=torch.int64, device=dev)
    _lib.check(L.idl_bias_grads(_p(dr1_a), _p(act1), 512, _p(p1a), _p(dlat_a), 64, _p(p2a), _p(dlg_a), C, _p(p3a), m, train,
                                _p(ctl_a), 7, _p(r2) if dw3 else None, _p(w3a) if dw3 else None, _stream()))
    # fused
    dlg_b = torch.empty(m, C, device=dev); dlat_b = torch.empty(m, 64, device=dev); dr1_b = torch.full((m, 512), float('nan'), device=dev)
    p1b = torch.full((parts, 512), float('nan'), device=dev); p2b = torch.full((parts, 64), float('nan'), device=dev)
    p3b = torch.full((parts, C), float('nan'), device=dev); w3b = torch.full((parts, C, 64), float('nan'), device=dev)
    ctl_b = torch.tensor([3, 100], dtype=torch.int64, device=dev)
    _lib.check(L.idl_mid_bwd(_p(z), _p(r2), _p(f), _p(inv), _p(G), gp, _p(dP0), _p(W3), _p(W2), _p(act1), m, C, train, coef,
                             _p(dlg_b), _p(dlat_b), _p(dr1_b), _p(p1b), _p(p2b), _p(p3b), _p(w3b) if dw3 else None, _p(ctl_b), 7, _stream()))
    torch.cuda.synchronize()
    assert torch.equal(ctl_a, ctl_b) and int(ctl_b[1]) == 107
    # same per-row arithmetic; the two kernels may contract a*b+c into an FMA at different places: a few ulp
    assert torch.allclose(dlg_a, dlg_b, rtol=1e-5, atol=1e-8) and torch.allclose(dlat_a, dlat_b, rtol=1e-5, atol=1e-8)
    assert torch.equal(dr1_a == 0, dr1_b == 0) or ((dr1_a == 0) != (dr1_b == 0)).float().mean() < 1e-4
    np.testing.assert_allclose(dr1_b.cpu().numpy(), dr1_a.cpu().numpy(), rtol=2e-4, atol=2e-6)
    for x, y in ((p1a, p1b), (p2a, p2b), (p3a, p3b)) + (((w3a, w3b),) if dw3 else ()):
        np.testing.assert_allclose(y.sum(0).cpu().numpy(), x.sum(0).cpu().numpy(), rtol=2e-4, atol=2e-5)
        assert torch.isfinite(y).all()                                          # every chunk slab written, empty chunks as zeros


@pytest.mark.parametrize("m,C,train", [(1024, 20, 1), (96, 5, 0), (160, 48, 1)])
def test_transposed_layer1_activations(dev, m, C, train):
    """The layer-1 activations may live transposed ([512, m], the orientation hipBLASLt runs W1 x^T fastest in): idl_mid_fwd_gather
    (bias added in the kernel) and idl_mid_bwd_gather on the transposed image give exactly what idl_mid_fwd / idl_mid_bwd give
    on the row-major one, and leave the transposed image of the same masked activations behind."""
    import torch
    from idelucs_amd import _lib
    from idelucs_amd.fused import _p, _stream
    L = _lib.lib
    torch.manual_seed(3 * m + C)
    a1 = torch.randn(m, 512, device=dev); b1 = torch.randn(512, device=dev) * 0.1
    W2 = torch.randn(64, 512, device=dev) * 0.06; b2 = torch.randn(64, device=dev) * 0.01
    W3 = torch.randn(C, 64, device=dev) * 0.2; b3 = torch.randn(C, device=dev) * 0.01
    ctl = torch.tensor([5, 0], dtype=torch.int64, device=dev)
    outs = []
    for tr_ in (0, 1):
        buf = (a1.t().contiguous() if tr_ else (a1 + b1).contiguous())
        f = torch.empty(m, 64, device=dev); inv = torch.empty(m, device=dev); r2 = torch.empty(m, 64, device=dev); z = torch.empty(m, C, device=dev)
        if tr_:
            _lib.check(L.idl_mid_fwd_gather(_p(buf), _p(b1), 1, _p(W2), _p(b2), _p(W3), _p(b3), m, C, train, 9, _p(ctl), _p(f), _p(inv), _p(r2), _p(z),
                                            None, 0, 0, 0, None, None, 0, 0, 0, None, None, None, None, 0, 1, 1, _stream()))
        else:
            _lib.check(L.idl_mid_fwd(_p(buf), _p(W2), _p(b2), _p(W3), _p(b3), m, C, train, 9, _p(ctl), _p(f), _p(inv), _p(r2), _p(z), _stream()))
        torch.cuda.synchronize()
        outs.append((buf.t().contiguous() if tr_ else buf, f, inv, r2, z))
    for x, y in zip(outs[0], outs[1]):
        assert torch.equal(x, y)
    # backward: same tensors, act1 row-major vs transposed
    r1, f, inv, r2, z = outs[0]
    parts = L.idl_col_sum_parts(); gp = L.idl_nce_fused_parts()
    G = torch.randn(gp, m, 64, device=dev) * 0.3
    dP0 = torch.randn(C, C, device=dev) * 0.1; dP0 = dP0 + dP0.t()
    res = []
    for tr_ in (0, 1):
        act = r1.t().contiguous() if tr_ else r1
        dlg = torch.empty(m, C, device=dev); dlat = torch.empty(m, 64, device=dev); dr1 = torch.empty(m, 512, device=dev)
        p1 = torch.empty(parts, 512, device=dev); p2 = torch.empty(parts, 64, device=dev); p3 = torch.empty(parts, C, device=dev)
        w3 = torch.empty(parts, C, 64, device=dev)
        if tr_:
            _lib.check(L.idl_mid_bwd_gather(_p(z), _p(r2), _p(f), _p(inv), _p(G), gp, _p(dP0), _p(W3), _p(W2), _p(act), m, C, train, 1e-3,
                                            _p(dlg), _p(dlat), _p(dr1), _p(p1), _p(p2), _p(p3), _p(w3),
                                            None, 0, 0, 0, None, None, 0, 0, 0, None, None, None, None, 0, 1, 1, 1, _stream()))
        else:
            _lib.check(L.idl_mid_bwd(_p(z), _p(r2), _p(f), _p(inv), _p(G), gp, _p(dP0), _p(W3), _p(W2), _p(act), m, C, train, 1e-3,
                                     _p(dlg), _p(dlat), _p(dr1), _p(p1), _p(p2), _p(p3), _p(w3), None, 0, _stream()))
        torch.cuda.synchronize()
        res.append((dlg, dlat, dr1, p1, p2, p3, w3))
    for x, y in zip(res[0], res[1]):
        assert torch.equal(x, y)


@pytest.mark.parametrize("m,F", [(1024, 4096), (96, 1024), (160, 256), (64, 448)])
def test_own_layer1_forward_tiles(dev, m, F):
    """idl_l1_fwd (csrc/l1_fwd.hip: a1^T = W1 x^T on own fp32 MFMA tiles, the first launch of the step's fp32 form; reference idelucs/PytorchUtils.py:38 at
    models.py:124-125) against float64 and against the library product it replaces; an unsupported shape is an error, never a silent other path."""
    import torch
    from idelucs_amd import _lib
    from idelucs_amd.fused import _p, _stream
    L = _lib.lib
    assert L.idl_l1_fwd_supported(m, 512, F) == 1 and L.idl_l1_fwd_supported(m, 512, F + 32) == 0 and L.idl_l1_fwd_supported(m, 512, 128) == 0 and L.idl_l1_fwd_supported(m + 8, 512, F) == 0
    torch.manual_seed(m + F)
    x = torch.randn(m, F, device=dev)
    W1 = torch.randn(512, F, device=dev) * (2.0 / F) ** 0.5
    out = torch.full((512, m), -3.0, device=dev)
    _lib.check(L.idl_l1_fwd(_p(W1), _p(x), m, F, _p(out), _stream()))
    torch.cuda.synchronize()
    want = (W1.double() @ x.double().t()).cpu().numpy()
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-4, atol=5e-5)
    e_own = np.abs(out.cpu().numpy() - want).max()
    e_lib = np.abs(torch.mm(W1, x.t()).cpu().numpy() - want).max()
    assert e_own <= 4 * e_lib + 1e-6, (e_own, e_lib)
    assert L.idl_l1_fwd(_p(W1), _p(x), m + 8, F, _p(out), _stream()) == _lib.IDL_ERR_ARG


@pytest.mark.parametrize("m,xt", [(1024, 0), (960, 0), (70, 0), (1024, 1), (960, 1), (72, 1)])
def test_dw2_inside_the_optimizer_launch(dev, m, xt):
    """idl_rmsprop_step_gather_wgrad: the gradient of one tensor computed in the launch as dy^T x (MFMA tiles) and applied
    there equals torch's product followed by the RMSprop formula; the other tensors are updated as by idl_rmsprop_step."""
    import ctypes
    import torch
    from idelucs_amd import _lib
    from idelucs_amd.fused import _p, _stream
    L = _lib.lib
    torch.manual_seed(m)
    dy = torch.randn(m, 64, device=dev); x = torch.relu(torch.randn(m, 512, device=dev))
    xarg = x.t().contiguous() if xt else x              # the product's second operand may be stored transposed, [512, m]
    W = [torch.randn(64, 512, device=dev) * 0.05, torch.randn(512, device=dev) * 0.1]
    V = [torch.rand(64, 512, device=dev) * 1e-3, torch.rand(512, device=dev) * 1e-3]
    gb = torch.randn(512, device=dev)
    grads = [torch.zeros(64, 512, device=dev), gb]
    W0, V0 = [w.clone() for w in W], [v.clone() for v in V]
    hyper = torch.tensor([1e-3, 0.99, 1e-8, 0.01, 0.01], device=dev)
    ctl = torch.zeros(2, dtype=torch.int64, device=dev)
    gout = torch.empty(64, 512, device=dev)
    arr = lambda ts: (ctypes.c_void_p * 2)(*[t_.data_ptr() for t_ in ts])
    _lib.check(L.idl_rmsprop_step_gather_wgrad(2, arr(W), arr(grads), (ctypes.c_int32 * 2)(1, 1), arr(V), (ctypes.c_int64 * 2)(64 * 512, 512),
                                               _p(hyper), _p(ctl), None, 0, 0.0, 0.0, None,
                                               None, 0, 0, 0, None, 0, 0, None, None, None, None,
                                               0, _p(dy), _p(xarg), xt, m, 64, 512, _p(gout), 5, _stream()))
    torch.cuda.synchronize()
    g64 = (dy.double().t() @ x.double())
    assert (gout.double() - g64).abs().max().item() <= 2e-6 * g64.abs().max().item() + 1e-5

    def upd(w, v, g):
        gi = g + 0.01 * w
        v2 = v * 0.99 + 0.01 * gi * gi
        return w - 1e-3 * (gi / (v2.sqrt() + 1e-8)), v2
    for w, v, w0, v0, g in zip(W, V, W0, V0, [gout, gb]):
        wr, vr = upd(w0, v0, g)
        assert torch.allclose(w, wr, rtol=1e-5, atol=1e-7) and torch.allclose(v, vr, rtol=1e-5, atol=1e-9)
    assert ctl.tolist() == [1, 5]


@pytest.mark.parametrize("m,fused", [(1024, True), (960, True), (128, False)])
def test_wgrad_kernel_vs_torch(dev, m, fused):
    """idl_wgrad_rmsprop (the dW1 tiles on their own): dy^T x on the fp32 matrix cores within fp32 summation error of a float64 product,
    and its fused RMSprop epilogue equals the formula applied to that gradient; unsupported shapes are refused."""
    import torch
    from idelucs_amd import _lib
    from idelucs_amd.fused import _p, _stream
    L = _lib.lib
    torch.manual_seed(m)
    H, F = 512, 1024
    dy = torch.randn(m, H, device=dev) * (torch.rand(m, H, device=dev) < 0.25); x = torch.randn(m, F, device=dev)
    W = torch.randn(H, F, device=dev) * 0.02; V = torch.rand(H, F, device=dev) * 1e-3
    W0, V0 = W.clone(), V.clone()
    hyper = torch.tensor([1e-3, 0.99, 1e-8, 0.01, 0.01], device=dev)
    g = torch.empty(H, F, device=dev)
    assert L.idl_wgrad_supported(m, H, F) == 1 and L.idl_wgrad_supported(m + 2, H, F) == 0 and L.idl_wgrad_supported(m, H + 32, F) == 0 and L.idl_wgrad_supported(m, H, F + 64) == 0
    with pytest.raises(ValueError):
        _lib.check(L.idl_wgrad_rmsprop(_p(dy), _p(x), m + 2, H, F, _p(g), None, None, None, _stream()))
    _lib.check(L.idl_wgrad_rmsprop(_p(dy), _p(x), m, H, F, _p(g), _p(W) if fused else None, _p(V) if fused else None,
                                   _p(hyper) if fused else None, _stream()))
    torch.cuda.synchronize()
    g64 = dy.double().t() @ x.double()
    assert (g.double() - g64).abs().max().item() <= 1e-6 * (dy.abs().double().t() @ x.abs().double()).max().item()
    if fused:
        gi = g + 0.01 * W0
        vr = V0 * 0.99 + 0.01 * gi * gi
        wr = W0 - 1e-3 * (gi / (vr.sqrt() + 1e-8))
        assert torch.allclose(W, wr, rtol=1e-5, atol=1e-7) and torch.allclose(V, vr, rtol=1e-5, atol=1e-9)
    else:
        assert torch.equal(W, W0) and torch.equal(V, V0)


@pytest.mark.parametrize("C", [49, 77, 120, 200, 256])
def test_iic_core_large_joint_vs_torch(dev, C):
    """idl_iic_core beyond the 48-class kernels (joint in LDS on ceil(C/8) workgroups up to C = 200, register-resident rows up to
    256): the loss and d(loss)/dP0 equal torch autograd through the reference formula (LossFunctions.py:20-62 restated on the
    C x C joint); a second call on the same scratch gives the same answer (the completion counter was left at zero)."""
    import torch
    from idelucs_amd import _lib
    from idelucs_amd.fused import _p, _stream, EPS
    torch.manual_seed(C)
    B = 300
    z1 = torch.softmax(torch.randn(B, C, device=dev) * 2, 1); z2 = torch.softmax(torch.randn(B, C, device=dev) * 2, 1)
    P0 = (z1.t() @ z2).contiguous().requires_grad_(True)
    P = (P0 + P0.t()) / 2.0
    P = P / P.sum()
    Pc = torch.where(P < EPS, torch.full_like(P, EPS), P)
    pi = P.sum(1, keepdim=True).expand(C, C); pj = P.sum(0, keepdim=True).expand(C, C)
    pic = torch.where(pi < EPS, torch.full_like(pi, EPS), pi); pjc = torch.where(pj < EPS, torch.full_like(pj, EPS), pj)
    loss = -(Pc * (torch.log(Pc) - 2.8 * torch.log(pjc) - 2.8 * torch.log(pic))).sum()
    loss.backward()
    scratch = torch.zeros(C * C + 2 * C, device=dev)
    want = 0.25 * P0.grad
    first = None
    for _ in range(3):
        buf = P0.detach().clone(); out = torch.zeros(4, device=dev)
        _lib.check(_lib.lib.idl_iic_core(_p(buf), C, 2.8, EPS, 0.25, _p(scratch), _p(out), _stream()))
        torch.cuda.synchronize()
        assert abs(out[3].item() - loss.item()) <= 1e-4 * abs(loss.item())
        assert torch.allclose(buf, want, rtol=2e-3, atol=2e-4 * want.abs().max().item())
        first = buf if first is None else first
        assert torch.equal(buf, first)


def test_head_bwd_with_precomputed_product(dev):
    """idl_head_bwd_dz (z dP0 as one GEMM by the caller, fine-grained mode) == idl_head_bwd with the per-row product inside."""
    import torch
    from idelucs_amd import _lib
    from idelucs_amd.fused import _p, _stream
    L = _lib.lib
    torch.manual_seed(2)
    m, C, gp = 256, 200, L.idl_nce_fused_parts()
    z = torch.softmax(torch.randn(m, C, device=dev), 1)
    lat = torch.randn(m, 64, device=dev); nrm = lat.norm(dim=1); f = lat / nrm[:, None]; inv = 1.0 / nrm
    r2 = torch.relu(torch.randn(m, 64, device=dev)); G = torch.randn(gp, m, 64, device=dev) * 0.3
    dP0 = torch.randn(C, C, device=dev) * 0.1; dP0 = (dP0 + dP0.t()).contiguous()
    W3 = torch.randn(C, 64, device=dev) * 0.2
    res = []
    for pre in (False, True):
        dlg = torch.empty(m, C, device=dev); dlat = torch.empty(m, 64, device=dev)
        if pre:
            dzs = z @ dP0
            _lib.check(L.idl_head_bwd_dz(_p(z), _p(r2), _p(f), _p(inv), _p(G), gp, _p(dzs), _p(W3), m, C, 1, 1e-3, _p(dlg), _p(dlat), _stream()))
        else:
            _lib.check(L.idl_head_bwd(_p(z), _p(r2), _p(f), _p(inv), _p(G), gp, _p(dP0), _p(W3), m, C, 1, 1e-3, _p(dlg), _p(dlat), _stream()))
        torch.cuda.synchronize()
        res.append((dlg, dlat))
    assert torch.allclose(res[0][0], res[1][0], rtol=1e-4, atol=1e-7) and torch.allclose(res[0][1], res[1][1], rtol=1e-4, atol=1e-7)


def test_opt_in_step_variants_agree_with_the_default(dev):
    """The opt-in / fallback ways of running the step (dW1 on hipBLASLt + a plain optimizer launch, the dW1 tiles as a launch of
    their own, batch assembly in the optimizer launch,
    row-major layer-1 activations, dW2 and the IIC joint as GEMM launches) train to the same parameters as the default launch
    sequence -- same batches, same dropout streams; only the summation orders inside the products differ."""
    import copy
    import torch
    from idelucs_amd import utils as U, models
    from idelucs_amd.PytorchUtils import NetLinear
    from idelucs_amd.fused import FusedLinearTrainer
    torch.manual_seed(5)
    P, n, F, B = 4, 1400, 256, 64            # m = 128: supported by idl_wgrad_rmsprop
    feats = (torch.rand((P, n, F), device=dev) * 1e-2).contiguous()
    mean, scale = U.col_stats(feats[0])
    store = U.FeatureStore(None, None, feats, mean, scale, 4, False)
    net0 = NetLinear(F, 6).to(dev); net0.apply(models.weights_init)

    def run(**flags):
        net = copy.deepcopy(net0)
        tr = FusedLinearTrainer(net, lr=1e-3, weight=0.25, lamb=2.8, seed=11)
        for k, v in flags.items():
            assert hasattr(tr, k), k
            setattr(tr, k, v)
        gen = torch.Generator(device=dev); gen.manual_seed(77)
        total, nb = tr.run_epoch(store, B, generator=gen)
        torch.cuda.synchronize()
        return [p.detach().clone() for p in tr.params], total.item()

    ref_p, ref_l = run()
    # RMSprop's first steps are sign-like, lr * g / (0.1 |g| + eps): rounding noise on a near-zero gradient moves a weight by up to
    # ~lr * 10, and 65 steps amplify it -- two correct runs whose products round differently end 1e-7 .. 4e-4 apart in the mean
    # (measured: the own layer-1 tiles, whose head adds lat up in another order, against the default 3.5e-4; the joint as a GEMM
    # against the default 3.5e-4 in one run and 2.5e-7 in another; dW1 on hipBLASLt: bit-identical at this shape), max 5e-3.  A wrong kernel
    # moves the weights (scale 3e-2) by >= 1e-2.  Bar: mean <= 1e-3, max <= 1e-2, epoch loss within 2e-3.
    variants = (dict(_wgrad_fused=False), dict(_wgrad_own_launch=True), dict(_early_gather=False), dict(_transposed_l1=False),
                dict(_dw2_inlaunch=False, _early_gather=False), dict(_joint_inlaunch=False), dict(_pipeline=False, _early_gather=False))
    for flags in variants:
        p, l = run(**flags)
        assert abs(l - ref_l) <= 2e-3 * abs(ref_l), (flags, l, ref_l)
        for a, b in zip(p, ref_p):
            assert (a - b).abs().max().item() <= 1e-2 and (a - b).abs().mean().item() <= 1e-3, flags
    again, l2 = run()
    assert l2 == ref_l and all(torch.equal(a, b) for a, b in zip(again, ref_p)), "the default step is not deterministic"


def test_batch_assembly_riding_in_the_middle_launches_is_the_gather(dev):
    """The tiles of the next batch assembled by the spare workgroups of idl_mid_fwd_gather (shares [0, 3) of 8) and
    idl_mid_bwd_gather (shares [3, 8)) are bit for bit idl_gather_pairs_at's batch, including the offset (base + base_add) and the
    rows past the end of the pair list, which are left alone."""
    import torch
    from idelucs_amd import _lib, utils as U
    from idelucs_amd.fused import _p, _stream
    L = _lib.lib
    torch.manual_seed(9)
    P, n, F, B, C, m = 4, 300, 1024, 512, 20, 1024
    feats = (torch.rand((P, n, F), device=dev) * 1e-3).contiguous()
    mean, scale = U.col_stats(feats[0]); inv_scale = (1.0 / scale).contiguous()
    n_pairs = (P - 1) * n                                   # 900 pairs: the batch at offset 600 has only 300 of its 512 rows
    # (the list is padded with valid indices: idl_gather_pairs_at, the reference here, has no end-of-list limit and reads 512 entries)
    perm = torch.cat([torch.randperm(n_pairs, device=dev), torch.zeros(512, dtype=torch.int64, device=dev)])
    ctl = torch.tensor([0, 88], dtype=torch.int64, device=dev)
    want = torch.full((m, F), -7.0, device=dev)
    base = ctl[1:].clone(); base += 512
    _lib.check(L.idl_gather_pairs_at(_p(feats), n, F, n * F, _p(perm), _p(base), B, _p(mean), _p(scale), _p(inv_scale), _p(want), _stream()))
    # (idl_gather_pairs_at has no pair-list limit: rows beyond it are compared only where the riding version writes)
    got = torch.full((m, F), -7.0, device=dev)
    a1 = torch.randn(m, 512, device=dev); W2 = torch.randn(64, 512, device=dev) * 0.06; b2 = torch.zeros(64, device=dev)
    W3 = torch.randn(C, 64, device=dev) * 0.2; b3 = torch.zeros(C, device=dev)
    f = torch.empty(m, 64, device=dev); inv = torch.empty(m, device=dev); r2 = torch.empty(m, 64, device=dev); z = torch.empty(m, C, device=dev)
    g_args = lambda p0, p1: (_p(feats), n, F, n * F, _p(perm), _p(ctl[1:]), 512, n_pairs, B, _p(mean), _p(scale), _p(inv_scale), _p(got), p0, p1, 8)
    _lib.check(L.idl_mid_fwd_gather(_p(a1), None, 0, _p(W2), _p(b2), _p(W3), _p(b3), m, C, 1, 3, _p(ctl), _p(f), _p(inv), _p(r2), _p(z),
                                    *g_args(0, 3), _stream()))
    parts = L.idl_col_sum_parts(); gp = L.idl_nce_fused_parts()
    G = torch.randn(gp, m, 64, device=dev); dP0 = torch.randn(C, C, device=dev); dP0 = dP0 + dP0.t()
    dlg = torch.empty(m, C, device=dev); dlat = torch.empty(m, 64, device=dev); dr1 = torch.empty(m, 512, device=dev)
    p1 = torch.empty(parts, 512, device=dev); p2 = torch.empty(parts, 64, device=dev); p3 = torch.empty(parts, C, device=dev)
    _lib.check(L.idl_mid_bwd_gather(_p(z), _p(r2), _p(f), _p(inv), _p(G), gp, _p(dP0), _p(W3), _p(W2), _p(a1), m, C, 1, 1e-3, _p(dlg), _p(dlat),
                                    _p(dr1), _p(p1), _p(p2), _p(p3), None, *g_args(3, 8), 0, _stream()))
    torch.cuda.synchronize()
    live = n_pairs - (88 + 512)                               # rows of each half that exist in the pair list
    assert 0 < live < B
    for half in (0, B):
        assert torch.equal(got[half:half + live], want[half:half + live])
        assert torch.all(got[half + live:half + B] == -7.0)   # beyond the end of the list: untouched
    assert ctl.tolist() == [0, 88]                            # the riding gather moves no offset


def test_small_model_k5_first_step_vs_reference(dev):
    """BASELINE config 4, k=5 through model_size='small' (tests/golden/small_k5.npz, from the imported reference): the device's
    canonical 5-mer frequency rows are the reference's kmersFasta(reduce=True) rows bit for bit, and myNet(512, 5) on them gives
    the reference's eval forward, first-step loss, gradients (rel 2e-3) and bias parameters after RMSprop."""
    import torch
    from idelucs_amd import utils as U
    from idelucs_amd.PytorchUtils import myNet
    from idelucs_amd.LossFunctions import IID_loss, info_nce_loss
    g = np.load(os.path.join(GOLDEN, "small_k5.npz"))
    _, rows = U.kmersFasta(os.path.join(DATA, "influenza_64.fas"), k=5, reduce=True)
    assert np.array_equal(rows[:32], g["rows"])
    net = myNet(512, 5)
    net.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")})
    net = net.to(dev).eval()
    x = torch.from_numpy(g["x"]).to(dev)
    with torch.no_grad():
        out, lat = net(x[:16].view(-1, 1, 512))
    np.testing.assert_allclose(out.cpu().numpy(), g["eval_out"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(lat.cpu().numpy(), g["eval_latent"], rtol=1e-4, atol=1e-5)
    opt = torch.optim.RMSprop(net.parameters(), lr=1e-3, weight_decay=0.01)
    opt.zero_grad()
    z, h = net(x)                                   # both views in one [2B, F] pass, as IID_model._step does
    loss = 0.75 * info_nce_loss(h[:16], h[16:], 0.85) + 0.25 * IID_loss(z[:16], z[16:], lamb=2.8)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) <= 2e-4 * abs(float(g["loss"]))
    for n_, p in net.named_parameters():
        want = g[f"g.{n_}"]
        got = p.grad.cpu().numpy()
        got = got[::8] if got.ndim == 2 and got.shape[0] >= 128 else got
        np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-3 * np.abs(want).mean(), err_msg=n_)
    opt.step()
    for n_, p in net.named_parameters():
        if p.dim() == 1:
            bad = ~np.isclose(p.detach().cpu().numpy(), g[f"p.{n_}"], rtol=1e-3, atol=1e-6)
            assert bad.mean() < 0.02, (n_, bad.mean())


@pytest.mark.parametrize("kw", [dict(k=4, batch_sz=256), dict(k=5, batch_sz=512, n_mimics=5), dict(k=6, batch_sz=48)])
def test_train_voters_batched_on_other_shapes(dev, kw):
    """training.train_voters with three voters in one batch on shapes other than cfg2's (F = 256 / 1024, batch 256, five mimics;
    batch 48: 96-row steps that the InfoNCE kernels still take): finite decreasing losses, distinct voters, predictions of the
    right shape, and the caller's model left with the LAST voter's weights."""
    import torch
    from idelucs_amd import training
    args = _args(n_epochs=4, n_voters=3, **kw)
    model = training.prepare_model(args)
    assert training.can_batch(model)
    out = training.train_voters(model, [0, 1, 2], args['n_epochs'], 3, lanes=3, progress=False)
    assert sorted(out) == [0, 1, 2]
    n = len(model.names)
    for v, (curve, y_pred, prob, latent) in out.items():
        assert len(curve) == 4 and all(np.isfinite(curve)) and curve[-1] < curve[0]
        assert y_pred.shape == (n,) and prob.shape == (n,) and latent.shape == (n, 64) and y_pred.max() < 5
    assert not np.array_equal(out[0][3], out[1][3])
    y_last = model.predict()[0]
    assert np.array_equal(y_last, out[2][1])


@pytest.mark.gpu
@pytest.mark.parametrize("m,C", [(1024, 200), (256, 64), (128, 49), (96, 130)])
def test_iic_core_dz_equals_the_core_and_the_product(dev, m, C):
    """idl_iic_core_dz (round 6; the step of n_clusters > 48): the IIC loss of idl_iic_core, and z dP0 for all rows with dP0 the gradient that
    launch pair leaves in P0 -- the shift by the gradient's global sum folded into the product's epilogue (a softmax row sums to 1)."""
    import torch
    from idelucs_amd import _lib
    from idelucs_amd.fused import _p, _stream
    L = _lib.lib
    g = torch.Generator(device="cpu"); g.manual_seed(C + m)
    z = torch.softmax(torch.randn(m, C, generator=g) * 2.0, 1).to(dev)
    P0 = (z[:m // 2].t() @ z[m // 2:]).contiguous()
    scratch = torch.zeros(C * C + 2 * C + 8, device=dev); out_a = torch.zeros(4, device=dev); out_b = torch.zeros(4, device=dev)
    dP0 = P0.clone()
    _lib.check(L.idl_iic_core(_p(dP0), C, 2.8, 2.2e-16, 0.25, _p(scratch), _p(out_a), _stream()))
    want = z.double() @ dP0.double()
    dzs = torch.full((m, C), float("nan"), device=dev)
    P0b = P0.clone()
    _lib.check(L.idl_iic_core_dz(_p(P0b), C, 2.8, 2.2e-16, 0.25, _p(scratch), _p(out_b), _p(z), m, _p(dzs), _stream()))
    torch.cuda.synchronize()
    assert torch.equal(P0b, P0) and out_a[3].item() == out_b[3].item()
    assert ((dzs.double() - want).abs().max() / want.abs().max()).item() < 2e-6
    assert L.idl_iic_core_dz(_p(P0b), 40, 2.8, 2.2e-16, 0.25, _p(scratch), _p(out_b), _p(z), m, _p(dzs), _stream()) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("K,M,N", [(1024, 200, 64), (300, 17, 70), (4, 1, 1), (1000, 256, 64)])
def test_at_b_equals_the_product(dev, K, M, N):
    """idl_at_b: out = A^T B (dW3 = dlogits^T r2 of the n_clusters > 48 step) against the float64 product."""
    import torch
    from idelucs_amd import _lib
    from idelucs_amd.fused import _p, _stream
    g = torch.Generator(device="cpu"); g.manual_seed(K + M)
    A = torch.randn(K, M + 3, generator=g).to(dev); B = torch.randn(K, N, generator=g).to(dev)
    out = torch.full((M, N + 2), 7.0, device=dev)
    _lib.check(_lib.lib.idl_at_b(_p(A), M + 3, _p(B), N, K, M, N, _p(out), N + 2, _stream()))
    torch.cuda.synchronize()
    want = A[:, :M].double().t() @ B.double()
    assert ((out[:, :N].double() - want).abs().max() / want.abs().max()).item() < 2e-6 and bool((out[:, N:] == 7.0).all())


@pytest.mark.gpu
@pytest.mark.parametrize("m,C", [(1024, 200), (1024, 37), (96, 5), (1000, 64)])
def test_iic_joint_kernel_equals_the_product(m, C):
    """idl_iic_joint: P0 = z[0:m/2]^T z[m/2:m] (reference LossFunctions.py:57-58) on one wave per 16 x 16 MFMA tile, for any
    n_clusters and any even batch (ragged tiles, a batch half that is not a multiple of 32)."""
    import ctypes
    import torch
    from idelucs_amd import _lib
    g = torch.Generator(device="cpu"); g.manual_seed(m + C)
    z = torch.softmax(torch.randn(m, C, generator=g), 1).cuda()
    P0 = torch.full((C, C), -1.0, device="cuda")
    _lib.check(_lib.lib.idl_iic_joint(ctypes.c_void_p(z.data_ptr()), m, C, ctypes.c_void_p(P0.data_ptr()),
                                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    want = (z[:m // 2].double().t() @ z[m // 2:].double())
    assert torch.allclose(P0.double(), want, rtol=1e-5, atol=1e-7)
    # ... and the same joint formed by spare workgroups of InfoNCE pass 1 (idl_nce_fused_joint: the step of n_clusters > 48, round 6), the InfoNCE
    # results being those of idl_nce_fused
    if m % 32 == 0 and _lib.lib.idl_nce_fused_workspace(m) > 0:
        from idelucs_amd.fused import _p, _stream
        L = _lib.lib
        f = torch.nn.functional.normalize(torch.randn(m, 64, generator=g), dim=1).cuda()
        ws = torch.empty(int(L.idl_nce_fused_workspace(m)) // 4, device="cuda")
        outs = []
        for joint in (False, True):
            lse = torch.empty(m, device="cuda"); rows = torch.empty(m, device="cuda"); G = torch.empty(int(L.idl_nce_fused_parts()), m, 64, device="cuda")
            P1 = torch.full((C, C), -1.0, device="cuda")
            if joint:
                _lib.check(L.idl_nce_fused_joint(_p(f), m, 0.85, _p(lse), _p(rows), _p(G), _p(ws), _p(z), _p(P1), C, _stream()))
            else:
                _lib.check(L.idl_nce_fused(_p(f), m, 0.85, _p(lse), _p(rows), _p(G), _p(ws), _stream()))
            torch.cuda.synchronize()
            outs.append((lse, rows, G, P1))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
        assert torch.allclose(outs[1][3].double(), want, rtol=1e-5, atol=1e-7) and bool((outs[0][3] == -1.0).all())


@pytest.mark.parametrize("n,use_graph", [(1500, False), (4200, True)])
def test_tail_riding_in_the_layer1_launch_changes_no_bit(dev, monkeypatch, n, use_graph):
    """Round 5: by default the layer-1 product runs on this package's own tiles and the optimizer's tail (dW2 tiles, RMSprop on the
    small tensors, step loss, step counter / batch offset) rides in the NEXT step's layer-1 launch (idl_l1_fwd_rms), a replayed
    graph ending with its last step's tail as a launch of its own.  Against the same step with the tail where it was -- behind
    the dW1 tiles of the optimizer launch -- and the same own layer-1 tiles (IDELUCS_TAIL_L1=0, IDELUCS_L1_FUSED=bare): every
    product is formed by the same instructions in the same order, so a whole epoch (8 or 24 full batches + a partial one,
    dropout on) leaves the SAME bits in every parameter, every running average, the loss sum and the counters."""
    import copy
    import torch
    from idelucs_amd.fused import FusedLinearTrainer
    store, net0 = _cfg2_store_and_net(dev, n, seed=6, C=20)
    B = 512
    out = []
    monkeypatch.setenv("IDELUCS_PLANES", "0")       # (the fp32 form of the step: the two-plane form has its own tests, test_gpu_planes.py)
    for tail, bare in (("1", "0"), ("0", "bare")):
        monkeypatch.setitem(__import__("idelucs_amd.fused", fromlist=["VARIANTS"]).VARIANTS, "tail_l1", tail)
        monkeypatch.setitem(__import__("idelucs_amd.fused", fromlist=["VARIANTS"]).VARIANTS, "l1_fused", bare)
        net = copy.deepcopy(net0)
        tr = FusedLinearTrainer(net, lr=1e-3, weight=0.25, lamb=2.8, seed=5)
        assert tr._tail_l1 == (tail == "1") and tr._l1_bare == (bare == "bare")
        gen = torch.Generator(device=dev); gen.manual_seed(123)
        total, nb = tr.run_epoch(store, B, use_graph=use_graph, generator=gen)
        total2, _ = tr.run_epoch(store, B, use_graph=use_graph, generator=gen)          # a second epoch: the graph replayed from its start
        torch.cuda.synchronize()
        assert tr._pending is None
        out.append(([p.detach().clone() for p in tr.params], [v.clone() for v in tr.square_avg], total2.item(), tr.ctl.tolist(), tr.out.tolist()))
    (pa, va, ta, ca, oa), (pb, vb, tb, cb, ob) = out
    assert ca == cb and ta == tb and oa == ob
    for a, b in zip(pa + va, pb + vb):
        assert torch.equal(a, b)
    # a step taken alone (tests, callers outside run_epoch) is complete when it returns: nothing stays pending
    monkeypatch.setitem(__import__("idelucs_amd.fused", fromlist=["VARIANTS"]).VARIANTS, "tail_l1", "1"); monkeypatch.setitem(__import__("idelucs_amd.fused", fromlist=["VARIANTS"]).VARIANTS, "l1_fused", "0")
    tr = FusedLinearTrainer(copy.deepcopy(net0), lr=1e-3, weight=0.25, lamb=2.8, seed=5)
    tr._perm = torch.randperm(store.n_pairs, device=dev)
    bf = tr.buffers(2 * B)
    tr._gather(store, bf)
    before = [p.detach().clone() for p in tr.params]
    tr._full_step(store, bf, pipelined=True)
    torch.cuda.synchronize()
    assert tr._pending is None and tr.ctl.tolist() == [1, B] and all(not torch.equal(a, b) for a, b in zip(before, tr.params))


@pytest.mark.parametrize("planes,C", [("1", 20), ("0", 20), ("1", 200)])
def test_default_step_contains_no_library_gemm(dev, monkeypatch, planes, C):
    """Round 5: the speed of the step's largest forward kernel must not depend on a hipBLASLt build, a TunableOp
    seed file or its validators.  The default step's launches, by kernel name (torch.profiler): the layer-1 product is
    l1_planes_kernel (+ reduce_rms_kernel) / wgrad_dplanes_rms_kernel in the two-plane form, l1_fwd_kernel / l1_rms_kernel / wgrad_q16_kernel
    with IDELUCS_PLANES=0 (own tiles either way), and NO rocBLAS / hipBLASLt kernel (Cijk_*)
    runs in a full-batch step -- also with the shipped solutions switched off (IDELUCS_TUNABLEOP_SEED=0), as on a box whose library
    differs from the seed file's."""
    import copy
    import torch
    from torch.profiler import profile, ProfilerActivity
    from idelucs_amd.fused import FusedLinearTrainer
    monkeypatch.setenv("IDELUCS_TUNABLEOP_SEED", "0")
    monkeypatch.setenv("IDELUCS_PLANES", planes)                        # the default (two-plane products) and the fp32 tiles
    store, net0 = _cfg2_store_and_net(dev, 1100, seed=8, C=C)           # 6 full batches + a partial one
    B = 512
    tr = FusedLinearTrainer(copy.deepcopy(net0), lr=1e-3, weight=0.25, lamb=2.8, seed=3)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    tr.run_epoch(store, B, use_graph=False, generator=gen)              # warm-up: allocations, library initialisation
    torch.cuda.synchronize()
    tr._perm = torch.randperm(store.n_pairs, device=dev)
    tr.ctl[1] = 0
    bf = tr.buffers(2 * B)
    tr._gather(store, bf)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for i in range(4):
            tr._full_step(store, bf, pipelined=True, xi=i % 2, defer_tail=True)
        tr.flush_tail()
        torch.cuda.synchronize()
    names = [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    if not names:
        pytest.skip("the profiler reported no device kernels on this box")
    if C > 48:      # (round 6) the step of the CLI's default mode -- 200 output units -- too: the middle backward in one launch, z dP0 and dW3 on own tiles
        assert any("l1_planes_kernel" in n for n in names) and any("wgrad_dplanes_rms" in n for n in names), sorted(set(names))
        assert any("mid_bwd_kernel<true, true>" in n or "mid_bwd_kernelILb1ELb1E" in n for n in names), sorted(set(names))
        assert any("iic_dz_kernel" in n for n in names) and any("at_b_kernel" in n for n in names) and not any("iic_joint_kernel" in n for n in names)
    elif planes == "1":
        assert any("l1_planes_kernel" in n for n in names) and any("reduce_rms_kernel" in n for n in names), sorted(set(names))
        assert any("wgrad_dplanes_rms" in n for n in names)     # (the tiles with both operands as planes + the step's optimizer tail on their loader waves)
        assert any("mid_bwd_kernel<false, true>" in n or "mid_bwd_kernelILb0ELb1E" in n for n in names), sorted(set(names))      # (... dr1 written as planes)
    else:
        assert any("l1_rms_kernel" in n for n in names) and any("l1_fwd_kernel" in n for n in names), sorted(set(names))
        assert any("wgrad_q16_kernel" in n for n in names)
    assert not any("Cijk_" in n for n in names), [n for n in names if "Cijk_" in n][:3]
