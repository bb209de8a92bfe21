#!/usr/bin/env python3
"""idl_l1_fwd (own fp32 MFMA tiles for the layer-1 forward, csrc/l1_fwd.hip) against torch.mm (hipBLASLt) at cfg2's shape:
correctness against float64 and time per launch back to back in a HIP graph.   python tools/bench_l1_fwd.py"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idelucs_amd import _lib
from idelucs_amd.fused import _p, _stream
L = _lib.lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
m, H, F = int(os.environ.get("M", 1024)), 512, int(os.environ.get("F", 4096))
x = torch.randn(m, F, device=dev)
W1 = torch.randn(H, F, device=dev) * 0.02
r1T = torch.empty(H, m, device=dev)

ref64 = W1.double() @ x.double().t()
_lib.check(L.idl_l1_fwd(_p(W1), _p(x), m, F, _p(r1T), _stream()))
tm = torch.mm(W1, x.t())
print("max|mine-f64|", (r1T.double() - ref64).abs().max().item(), "  max|torch-f64|", (tm.double() - ref64).abs().max().item())


def bench(name, fn, n=50):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); fn()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(n): fn()
        gr.replay(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        for _ in range(40): gr.replay()
        e0.record()
        for _ in range(10): gr.replay()
        e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / (10 * n)
    print(f"{name:52s} {us:7.2f} us / launch   ({2.0 * m * H * F / us / 1e6:.1f} TFLOP/s)")

bench("torch.mm(W1, x.t()) -> a1^T (untuned heuristic)", lambda: torch.mm(W1, x.t(), out=r1T))
bench("idl_l1_fwd (own fp32 MFMA tiles)", lambda: L.idl_l1_fwd(_p(W1), _p(x), m, F, _p(r1T), _stream()))
