#!/usr/bin/env python3
"""idl_wgrad_rmsprop vs torch.mm (+ the separate optimizer) at the cfg2 shapes: correctness and time per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idelucs_amd import _lib
from idelucs_amd.fused import _p, _stream
L = _lib.lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
m, H, F = int(os.environ.get("M", 1024)), 512, int(os.environ.get("F", 4096))
dy = torch.randn(m, H, device=dev) * (torch.rand(m, H, device=dev) < 0.25)
x = torch.randn(m, F, device=dev)
dlat = torch.randn(m, 64, device=dev); r1 = torch.relu(torch.randn(m, H, device=dev))
W = torch.randn(H, F, device=dev) * 0.02; V = torch.rand(H, F, device=dev) * 1e-3
W2 = torch.randn(64, H, device=dev) * 0.06; V2 = torch.rand(64, H, device=dev) * 1e-3
hyper = torch.tensor([1e-3, 0.99, 1e-8, 0.01, 0.01], device=dev)
g = torch.empty(H, F, device=dev); g2 = torch.empty(64, H, device=dev)

def ref_update(W, V, g):
    gi = g + 0.01 * W
    V2_ = V * 0.99 + 0.01 * gi * gi
    return W - 1e-3 * (gi / (V2_.sqrt() + 1e-8)), V2_

# 1. plain gradient
_lib.check(L.idl_wgrad_rmsprop(_p(dy), _p(x), m, H, F, _p(g), None, None, None, _stream()))
ref = dy.t() @ x
ref64 = (dy.double().t() @ x.double())
print("grad  max|mine-f64|", (g.double() - ref64).abs().max().item(), " max|torch-f64|", (ref.double() - ref64).abs().max().item())
# 2. fused update
Wc, Vc = W.clone(), V.clone()
_lib.check(L.idl_wgrad_rmsprop(_p(dy), _p(x), m, H, F, _p(g), _p(Wc), _p(Vc), _p(hyper), _stream()))
Wr, Vr = ref_update(W, V, g)
print("fused W  max rel", ((Wc - Wr).abs().max() / Wr.abs().max()).item(), " V", ((Vc - Vr).abs().max() / Vr.abs().max()).item())

def bench(name, fn, n=50):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); fn()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(n): fn()
        gr.replay(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): gr.replay()
        e1.record(); torch.cuda.synchronize()
    print(f"{name:44s} {e0.elapsed_time(e1) * 1000 / (5 * n):7.2f} us / launch")

bench("torch.mm(dy.t(), x)", lambda: torch.mm(dy.t(), x, out=g))
bench("wgrad plain", lambda: L.idl_wgrad_rmsprop(_p(dy), _p(x), m, H, F, _p(g), None, None, None, _stream()))
bench("wgrad + rmsprop", lambda: L.idl_wgrad_rmsprop(_p(dy), _p(x), m, H, F, None, _p(Wc), _p(Vc), _p(hyper), _stream()))
bench("torch.mm(dlat.t(), r1)", lambda: torch.mm(dlat.t(), r1, out=g2))
