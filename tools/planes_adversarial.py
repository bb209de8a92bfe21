#!/usr/bin/env python3
"""The two plane products against float64 on adversarial operands (VERDICT r5 #2b): columns spanning 1e-6 .. 30, near-cancelling rows, entries in the
low plane's subnormal range, a few huge outliers.  Prints, per case, the error of the plane product and of the fp32 reference (torch's GEMM for the
layer-1 product, the fp32 tiles for dW1), both relative to the float64 product's largest entry.     python3 tools/planes_adversarial.py"""
import ctypes
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def cases(m, H, F, g):
    """(name, W [H, F], x [m, F], dy [m, H])"""
    W0 = (torch.rand(H, F, generator=g) * 2 - 1) / F ** 0.5
    x0 = torch.randn(m, F, generator=g)
    dy0 = torch.randn(m, H, generator=g) * 1e-3
    out = [("plain", W0, x0, dy0)]
    out.append(("columns 1e-6 .. 30", W0, x0 * torch.logspace(-6, 1.5, F), dy0 * torch.logspace(-6, 0, H)))
    out.append(("rows 1e-6 .. 30", W0 * torch.logspace(-4, 0, H)[:, None], x0 * torch.logspace(-6, 1.5, m)[:, None], dy0 * torch.logspace(-6, 0, m)[:, None]))
    xc = x0.clone(); xc[:, 1::2] = -xc[:, 0::2] * (1 + 1e-6 * torch.randn(m, F // 2, generator=g))      # products that cancel pairwise when W's columns pair up
    Wc = W0.clone(); Wc[:, 1::2] = Wc[:, 0::2]
    dyc = dy0.clone(); dyc[1::2] = -dyc[0::2] * (1 + 1e-6 * torch.randn(m // 2, H, generator=g))
    xc2 = x0.clone(); xc2[1::2] = xc2[0::2]
    out.append(("near-cancelling sums", Wc, xc, dy0))
    out.append(("near-cancelling rows of dy", W0, xc2, dyc))
    out.append(("low plane subnormal (|x| ~ 1e-4)", W0 * 1e-3, x0 * 1e-4, dy0 * 1e-4))
    xo = x0.clone(); xo[3, 5] = 8000.0; xo[100, 4000 % F] = -7000.0
    Wo = W0.clone(); Wo[7, 5] = 15.0
    dyo = dy0.clone(); dyo[3, 7] = 0.4
    out.append(("outliers near the range's end", Wo, xo, dyo))
    out.append(("sparse (95 % zeros)", W0, x0 * (torch.rand(m, F, generator=g) > 0.95), dy0 * (torch.rand(m, H, generator=g) > 0.95)))
    return out


def main():
    from idelucs_amd import _lib
    L = _lib.lib
    dev = torch.device("cuda")
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cpu"); g.manual_seed(11)
    m, H, F = 1024, 512, 4096
    h16 = lambda t: torch.empty(t.shape, dtype=torch.int16, device=dev)
    rows = []
    for name, W, x, dy in cases(m, H, F, g):
        W, x, dy = W.to(dev).contiguous(), x.to(dev).contiguous(), dy.to(dev).contiguous()
        wh, wl, xh, xl, flag = h16(W), h16(W), h16(x), h16(x), torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(L.idl_split_planes(p(W), W.numel(), L.idl_planes_exponent(1), p(wh), p(wl), p(flag), st()))
        _lib.check(L.idl_split_planes(p(x), x.numel(), L.idl_planes_exponent(0), p(xh), p(xl), p(flag), st()))
        part = torch.empty(int(L.idl_l1_planes_parts()), H, m, device=dev)
        _lib.check(L.idl_l1_planes(p(wh), p(wl), F, p(xh), p(xl), F, m, H, F, p(part), st()))
        ref = W.double() @ x.double().t()
        s1 = ref.abs().max().item()
        e_pl = (part.double().sum(0) - ref).abs().max().item() / s1
        e_lib = ((W @ x.t()).double() - ref).abs().max().item() / s1
        # dW1 = dy^T x: dy as planes with the scale mid_bwd would give it (2^k max in [2^8, 2^9)) and with four times less headroom used
        refg = dy.double().t() @ x.double()
        s2 = refg.abs().max().item()
        g32 = torch.empty(H, F, device=dev)
        _lib.check(L.idl_wgrad_rmsprop(p(dy), p(x), m, H, F, p(g32), None, None, None, st()))
        e_32 = (g32.double() - refg).abs().max().item() / s2
        kd = 9 - math.frexp(dy.abs().max().item())[1]
        dyh, dyl = h16(dy), h16(dy)
        _lib.check(L.idl_split_planes(p(dy), dy.numel(), kd, p(dyh), p(dyl), p(flag), st()))
        sc = torch.zeros(int(L.idl_dr1_scale_words()), dtype=torch.int32, device=dev); sc[0] = kd
        gpl = torch.empty(H, F, device=dev)
        _lib.check(L.idl_wgrad_rmsprop_xplanes(p(dyh), p(dyl), p(sc), p(xh), p(xl), F, m, H, F, p(gpl), None, None, None, None, None, None, st()))
        torch.cuda.synchronize()
        e_g = (gpl.double() - refg).abs().max().item() / s2
        rows.append((name, e_pl, e_lib, e_g, e_32, int(flag.item())))
    print(f"{'case':36s} {'l1 planes':>10s} {'fp32 GEMM':>10s} | {'dW1 planes':>10s} {'fp32 tiles':>10s}  flag")
    for name, a, b, c, d, f in rows:
        print(f"{name:36s} {a:10.2e} {b:10.2e} | {c:10.2e} {d:10.2e}  {f}")


if __name__ == "__main__":
    main()
