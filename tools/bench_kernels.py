#!/usr/bin/env python3
"""Per-kernel timing of the fused step's own kernels at the cfg2 shapes (m = 1024, F = 4096, C = 20), each replayed
100x inside a HIP graph:  python tools/bench_kernels.py [--C 20]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idelucs_amd import _lib
from idelucs_amd.fused import _p, _stream, EPS
L = _lib.lib
ap = argparse.ArgumentParser(); ap.add_argument("--C", type=int, default=20); ap.add_argument("--m", type=int, default=1024)
a = ap.parse_args()
dev = torch.device("cuda:0"); m, C = a.m, a.C
torch.manual_seed(0)
a1 = torch.randn(m, 512, device=dev); W2 = torch.randn(64, 512, device=dev) * .06; b2 = torch.zeros(64, device=dev)
W3 = torch.randn(C, 64, device=dev) * .2; b3 = torch.zeros(C, device=dev)
ctl = torch.tensor([3, 0], dtype=torch.int64, device=dev)
f = torch.empty(m, 64, device=dev); inv = torch.empty(m, device=dev); r2 = torch.empty(m, 64, device=dev); z = torch.empty(m, C, device=dev)
parts = L.idl_col_sum_parts(); gp = L.idl_nce_fused_parts()
G = torch.randn(gp, m, 64, device=dev); P0 = torch.randn(C, C, device=dev); P0 = P0 + P0.t()
dlg = torch.empty(m, C, device=dev); dlat = torch.empty(m, 64, device=dev); dr1 = torch.empty(m, 512, device=dev)
p1 = torch.empty(parts, 512, device=dev); p2 = torch.empty(parts, 64, device=dev); p3 = torch.empty(parts, C, device=dev); w3p = torch.empty(parts, C, 64, device=dev)
lse = torch.empty(m, device=dev); rows = torch.empty(m, device=dev); ws = torch.empty(max(L.idl_nce_fused_workspace(m), 4) // 4, device=dev)
scr = torch.zeros(C * C + 2 * C, device=dev); out = torch.zeros(4, device=dev)
lat = torch.randn(m, 64, device=dev)

def k_mid_fwd():
    _lib.check(L.idl_mid_fwd(_p(a1), _p(W2), _p(b2), _p(W3), _p(b3), m, C, 1, 7, _p(ctl), _p(f), _p(inv), _p(r2), _p(z), _stream()))
def k_head_fwd():
    _lib.check(L.idl_head_fwd(_p(lat), _p(W3), _p(b3), m, C, 1, 7, _p(ctl), _p(f), _p(inv), _p(r2), _p(z), _stream()))
def k_relu():
    _lib.check(L.idl_relu_dropout_fwd(_p(a1), a1.numel(), 1, 7, _p(ctl), 1, _stream()))
def k_mid_bwd():
    _lib.check(L.idl_mid_bwd(_p(z), _p(r2), _p(f), _p(inv), _p(G), gp, _p(P0), _p(W3), _p(W2), _p(a1), m, C, 1, 1e-3, _p(dlg), _p(dlat),
                             _p(dr1), _p(p1), _p(p2), _p(p3), _p(w3p) if C <= 48 else None, None, 0, _stream()))
def k_head_bwd():
    _lib.check(L.idl_head_bwd(_p(z), _p(r2), _p(f), _p(inv), _p(G), gp, _p(P0), _p(W3), m, C, 1, 1e-3, _p(dlg), _p(dlat), _stream()))
def k_nce():
    _lib.check(L.idl_nce_fused_iic(_p(f), m, 0.85, _p(lse), _p(rows), _p(G), _p(ws), _p(P0), C, 2.8, EPS, 0.25, _p(scr), _p(out), _stream()))
def k_empty():
    _lib.check(L.idl_iic_core(_p(P0), C, 2.8, EPS, 0.25, _p(scr), _p(out), _stream()))

k_mid_fwd()      # f, inv, r2, z become consistent
for name, fn in (("iic_core (1 WG: launch floor)", k_empty), ("relu_dropout_fwd", k_relu), ("head_fwd", k_head_fwd), ("mid_fwd", k_mid_fwd),
                 ("head_bwd", k_head_bwd), ("mid_bwd", k_mid_bwd), ("nce_fused_iic", k_nce)):
    if name.startswith("nce") and (ws.numel() <= 1 or C > 48): continue
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(100): fn()
        g.replay(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): g.replay()
        e1.record(); torch.cuda.synchronize()
    print(f"{name:32s} {e0.elapsed_time(e1) * 1000 / 500:7.2f} us / launch (back to back in a graph)")
