#!/usr/bin/env python3
"""Diagnostic: the optimizer launch of the fused step (idl_rmsprop_step_gather_wgrad with the in-launch dW2 tiles) timed alone,
back to back inside a HIP graph, at the cfg2 shapes -- with its operands cache-hot (same buffers every launch) and cache-cold
(cycling through enough copies of W1 / v / g to exceed the 256 MB Infinity Cache)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idelucs_amd import _lib, models
from idelucs_amd.PytorchUtils import NetLinear
from idelucs_amd.fused import FusedLinearTrainer, _p, _stream
L = _lib.lib
dev = torch.device("cuda:0")
def make():
    net = NetLinear(4096, 20).to(dev); net.apply(models.weights_init)
    tr = FusedLinearTrainer(net, lr=1e-3, weight=0.25, lamb=2.8, seed=1)
    bf = tr.buffers(1024)
    bf.dlat.normal_(); bf.r1.normal_(); bf.loss_rows.normal_()
    for g in tr.grads: g.normal_(std=1e-3)
    return tr, bf
def launch(tr, bf, tiles=True):
    m = 1024
    if tiles:
        _lib.check(L.idl_rmsprop_step_gather_wgrad(len(tr.params), tr._pp, tr._gp, tr._parts, tr._vp, tr._sz, _p(tr.hyper), _p(tr.ctl), _p(bf.loss_rows), m,
                                                   0.75, 0.25, _p(tr.out), None, 0, 0, 0, None, 0, 0, None, None, None, None,
                                                   2, _p(bf.dlat), _p(bf.r1), 1, m, 64, 512, _p(tr.grads[2]), m // 2, _stream()))
    else:
        _lib.check(L.idl_rmsprop_step(len(tr.params), tr._pp, tr._gp, tr._parts, tr._vp, tr._sz, _p(tr.hyper), _p(tr.ctl), m // 2, _p(bf.loss_rows), m, 0.75, 0.25,
                                      _p(tr.out), _stream()))
def timeit(fn, n=200):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
tr, bf = make()
print(f"hot, with dW2 tiles     {timeit(lambda: launch(tr, bf, True)):6.2f} us")
print(f"hot, without dW2 tiles  {timeit(lambda: launch(tr, bf, False)):6.2f} us")
sets = [make() for _ in range(16)]          # 16 x 25 MB of W1 / v / g = 400 MB: every launch finds its operands out of the caches
it = [0]
def cold(tiles):
    t, b = sets[it[0] % 16]; it[0] += 1
    launch(t, b, tiles)
print(f"cold, with dW2 tiles    {timeit(lambda: cold(True), 160):6.2f} us")
print(f"cold, without dW2 tiles {timeit(lambda: cold(False), 160):6.2f} us")
