#!/usr/bin/env python3
"""Layer-1 GEMM shapes under the layouts available to the step (TunableOp on): which orientation hipBLASLt runs fastest."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault("IDELUCS_TUNABLEOP", "1")
from idelucs_amd import gemm_tuning
gemm_tuning.maybe_enable()
dev = torch.device("cuda:0")
m, H, F = 1024, 512, 4096
x = torch.randn(m, F, device=dev); xT = x.t().contiguous()
W = torch.randn(H, F, device=dev) * 0.02; WT = W.t().contiguous()
b = torch.zeros(H, device=dev)
dy = torch.randn(m, H, device=dev); dyT = dy.t().contiguous()
o1 = torch.empty(m, H, device=dev); o1T = torch.empty(H, m, device=dev); g = torch.empty(H, F, device=dev); gT = torch.empty(F, H, device=dev)

def bench(name, fn, n=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
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
    print(f"{name:52s} {e0.elapsed_time(e1) * 1000 / (5 * n):7.2f} us")

bench("fwd  addmm(b, x, W.t())            [NT] (now)", lambda: torch.addmm(b, x, W.t(), out=o1))
bench("fwd  mm(x, W.t())  no bias          [NT]", lambda: torch.mm(x, W.t(), out=o1))
bench("fwd  mm(x, WT)                     [NN]", lambda: torch.mm(x, WT, out=o1))
bench("fwd  mm(W, x.t()) -> r1^T          [NT']", lambda: torch.mm(W, x.t(), out=o1T))
bench("fwd  mm(xT.t(), WT)                [TN]", lambda: torch.mm(xT.t(), WT, out=o1))
bench("bwd  mm(dy.t(), x)                 [TN] (now)", lambda: torch.mm(dy.t(), x, out=g))
bench("bwd  mm(x.t(), dy) -> gW^T         [TN']", lambda: torch.mm(x.t(), dy, out=gT))
bench("bwd  mm(dyT, x)                    [NN]", lambda: torch.mm(dyT, x, out=g))
