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


def test_end_to_end_quality_anchor(dev):
    """Reference on CPU: Influenza-A, k=6, C=5, 10 epochs, 1 voter -> ACC 0.99368 (tests/golden/anchor.json).
    Dropout/shuffle RNG differ on the GPU, so the bar is statistical: ACC >= 0.97."""
    import pandas as pd
    import torch
    import idelucs_amd
    anchor = json.load(open(os.path.join(GOLDEN, "anchor.json")))
    torch.manual_seed(0)
    m = idelucs_amd.iDeLUCS_cluster(os.path.join(DATA, "Influenza-A.fas"), n_clusters=5, n_epochs=10, n_mimics=3,
                                    batch_sz=512, k=6, weight=0.25, n_voters=1)
    y, lat = m.fit_predict(None)
    assert y.dtype == np.int64 and y.shape == (949,) and lat.dtype == np.float64 and lat.shape == tuple(anchor["latent_shape"])
    df = pd.read_csv(os.path.join(DATA, "Influenza-A_GT.tsv"), sep="\t")
    u = {v: i for i, v in enumerate(sorted(set(df.cluster_id)))}
    gt = np.array([u[v] for v in df.cluster_id])
    _, acc = idelucs_amd.cluster_acc(gt, y)
    print("ACC", acc, "reference", anchor["acc"])
    assert acc >= 0.97
