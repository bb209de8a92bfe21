#!/usr/bin/env python3
"""Probe: the layer-1 product with fp32 results from the bf16 matrix cores -- every operand split three ways (x = x0 + x1 + x2, 8
mantissa bits each), the six largest partial products as ONE bf16 GEMM over a six-fold K (fp32 accumulator and output).  Times the
library GEMM of that shape against the fp32 one and measures the error against float64.   python3 tools/probe_split_gemm.py"""
import torch

dev = torch.device("cuda")


def t(fn, n=50):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1000


def split3(x):
    x0 = x.bfloat16(); r = x - x0.float()
    x1 = r.bfloat16(); r = r - x1.float()
    x2 = r.bfloat16()
    return x0, x1, x2


def main():
    g = torch.Generator(device="cpu"); g.manual_seed(0)
    m, F, H = 1024, 4096, 512
    x = torch.randn(m, F, generator=g).to(dev); W = (torch.randn(H, F, generator=g) * 0.02).to(dev)
    ref = W.double() @ x.double().t()
    out32 = torch.empty(H, m, device=dev)
    print("fp32 mm(W, x.t()): %.1f us, max rel err %.2e" % (t(lambda: torch.mm(W, x.t(), out=out32)), float((out32.double() - ref).abs().max() / ref.abs().max())))
    x0, x1, x2 = split3(x); w0, w1, w2 = split3(W)
    for name, wa, xa in (("x6 (drop w1x2, w2x1, w2x2)", [w0, w0, w0, w1, w1, w2], [x0, x1, x2, x0, x1, x0]),
                         ("x9", [w0, w0, w0, w1, w1, w1, w2, w2, w2], [x0, x1, x2, x0, x1, x2, x0, x1, x2]),
                         ("x3 (w0x0, w0x1, w1x0)", [w0, w0, w1], [x0, x1, x0])):
        Wc = torch.cat(wa, 1).contiguous(); Xc = torch.cat(xa, 1).contiguous()
        try:
            o = torch.mm(Wc, Xc.t(), out_dtype=torch.float32)
            us = t(lambda: torch.mm(Wc, Xc.t(), out_dtype=torch.float32))
            print("bf16 %-28s K' = %6d: %.1f us, max rel err %.2e" % (name, Wc.shape[1], us, float((o.double() - ref).abs().max() / ref.abs().max())))
        except Exception as err:
            print("bf16", name, "failed:", type(err).__name__, str(err)[:160])
    print("split of x (3 parts, torch ops): %.1f us" % t(lambda: split3(x)))


if __name__ == "__main__":
    main()
