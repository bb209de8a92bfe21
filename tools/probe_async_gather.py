#!/usr/bin/env python3
"""Feasibility probe (round 4): would the next batch's assembly hide if it ran as kernels of a SECOND stream, free-running beside the
step's graph (device-side flags instead of graph edges)?  Stream A replays the default step graph with its riders switched off
(feats = NULL in the two middle launches: wrong training, right timing); stream B meanwhile runs idl_gather_pairs_at kernels
back to back, or one per step's time.  Reports A's time per step alone and beside B, and B's time per gather."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idelucs_amd import _lib, utils as U, models, fused
from idelucs_amd.PytorchUtils import NetLinear
from idelucs_amd.fused import FusedLinearTrainer, _p
L = _lib.lib
dev = torch.device("cuda:0")
P, n, F, C, B = 4, 100000, 4096, 20, 512
g = torch.Generator(device=dev); g.manual_seed(1)
feats = torch.rand((P, n, F), device=dev, generator=g) * 2e-4 + 1e-4
mean, scale = U.col_stats(feats[0])
store = U.FeatureStore(None, None, feats, mean, scale, 6, False)
net = NetLinear(F, C).to(dev); net.apply(models.weights_init)

def trainer(riders):
    tr = FusedLinearTrainer(net, 1e-3, 0.25, 2.8, seed=3)
    if not riders:
        real = fused._p
        class NoFeats:
            pass
        # the two middle launches get feats = NULL (no batch assembly in the launch): patch _p for store.feats only
        def p2(t):
            return None if (t is store.feats) else real(t)
        tr._p_patch = p2
    return tr

def epoch_time(tr, reps=3, side=None):
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    tr.run_epoch(store, B, generator=gen); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        stop = False
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if side is not None:
            side()
        tr.run_epoch(store, B, generator=gen)
        torch.cuda.current_stream().synchronize(); t1 = time.perf_counter()
        torch.cuda.synchronize()
        best = min(best, t1 - t0)
    return best / 586 * 1e6

tr = trainer(True)
print(f"A alone, riders in the middle launches (the default): {epoch_time(tr):.2f} us per step")
orig = fused._p
patched = lambda t: None if (t is store.feats) else orig(t)
class NoRiders(FusedLinearTrainer):
    def _gather(self, st, bf):                       # (the prologue's eager gather keeps its store)
        fused._p = orig
        try:
            super()._gather(st, bf)
        finally:
            fused._p = patched
fused._p = patched
tr2 = NoRiders(net, 1e-3, 0.25, 2.8, seed=3)
a_alone = epoch_time(tr2)
print(f"A alone, riders OFF (no assembly at all: the floor): {a_alone:.2f} us per step")
# stream B: gathers into a scratch x buffer, captured as a graph of 64, replayed while A runs
sB = torch.cuda.Stream()
xb = torch.empty((2 * B, F), device=dev)
perm = torch.randperm(store.n_pairs, device=dev)
ctl = torch.zeros(2, dtype=torch.int64, device=dev)
def one_gather():
    _lib.check(L.idl_gather_pairs_at(orig(store.feats), store.n, store.f, store.n * store.f, orig(perm), orig(ctl[1:]), B, orig(store.mean), orig(store.scale),
                                     orig(store.inv_scale), orig(xb), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
with torch.cuda.stream(sB):
    one_gather(); one_gather()
    gB = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gB):
        for _ in range(64):
            one_gather()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): gB.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"B alone: {e0.elapsed_time(e1) * 1000 / 256:.2f} us per whole-batch gather (back to back)")
for n_rep, label in ((10, "586+ gathers, back to back (more than one per step)"),):
    def side():
        with torch.cuda.stream(sB):
            for _ in range(n_rep): gB.replay()
    t = epoch_time(tr2, side=side)
    print(f"A beside B ({label}): {t:.2f} us per step  (+{t - a_alone:.2f})")
fused._p = orig
