#!/usr/bin/env python3
"""idl_l1_fwd (own fp32 MFMA tiles for the layer-1 forward, csrc/l1_fwd.hip) against torch.mm (hipBLASLt) at cfg2's shape:
correctness of the bare product and of the fused epilogue (bias + ReLU + Dropout + the K-split of Linear(512, 64)), and time per
launch back to back in a HIP graph.   python tools/bench_l1_fwd.py"""
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
b1 = torch.randn(H, device=dev) * 0.1
W2 = torch.randn(64, H, device=dev) * 0.06
ctl = torch.tensor([7, 0], dtype=torch.int64, device=dev)
seed = 12345
P = L.idl_l1_fwd_parts()
r1T = torch.empty(H, m, device=dev); r1 = torch.empty(m, H, device=dev)
lat_part = torch.empty(P, m, 64, device=dev)

ref64 = W1.double() @ x.double().t()
# 1. bare product
_lib.check(L.idl_l1_fwd(_p(W1), _p(x), None, None, m, F, 0, ctypes.c_uint64(seed), None, _p(r1T), 1, None, _stream()))
tm = torch.mm(W1, x.t())
print("bare   max|mine-f64|", (r1T.double() - ref64).abs().max().item(), "  max|torch-f64|", (tm.double() - ref64).abs().max().item())
# 2. epilogue, eval and train, both orientations
for train in (0, 1):
    a1 = (x @ W1.t() + b1).contiguous()                      # [m, 512] row-major
    _lib.check(L.idl_relu_dropout_fwd(_p(a1), a1.numel(), train, ctypes.c_uint64(seed), _p(ctl), 1, _stream()))
    for tl in (1, 0):
        out = r1T if tl else r1
        out.fill_(-7.0); lat_part.fill_(-7.0)
        _lib.check(L.idl_l1_fwd(_p(W1), _p(x), _p(b1), _p(W2), m, F, train, ctypes.c_uint64(seed), _p(ctl), _p(out), tl, _p(lat_part), _stream()))
        got = out.t() if tl else out
        # (a pre-activation within rounding of zero may fall on the other side of the ReLU: compare where the reference is clear of it)
        pre = (x.double() @ W1.double().t() + b1.double())
        clear = pre.abs() > 1e-4
        err = ((got.double() - a1.double()).abs() * clear).max().item()
        mask_same = ((got != 0) == (a1 != 0))[clear].float().mean().item()
        lat_ref = a1.double() @ W2.double().t()
        lat = lat_part.double().sum(0)
        print(f"train={train} transposed={tl}: r1 max err {err:.3e} (masks equal on {100 * mask_same:.4f} % of the clear elements), "
              f"lat max err {(lat - lat_ref).abs().max().item():.3e} of scale {lat_ref.abs().max().item():.2f}")


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
bench("idl_l1_fwd bare product", lambda: L.idl_l1_fwd(_p(W1), _p(x), None, None, m, F, 0, ctypes.c_uint64(seed), None, _p(r1T), 1, None, _stream()))
bench("idl_l1_fwd + epilogue (train)", lambda: L.idl_l1_fwd(_p(W1), _p(x), _p(b1), _p(W2), m, F, 1, ctypes.c_uint64(seed), _p(ctl), _p(r1T), 1, _p(lat_part), _stream()))
bench("idl_l1_fwd + epilogue (eval)", lambda: L.idl_l1_fwd(_p(W1), _p(x), _p(b1), _p(W2), m, F, 0, ctypes.c_uint64(seed), _p(ctl), _p(r1T), 1, _p(lat_part), _stream()))
