#!/usr/bin/env python3
"""The layer-1 product from two-plane operands (csrc/l1_planes_device.h) beside the fp32 tiles (idl_l1_fwd) at the step's shape, the
producers' side (the dW1 tiles with and without W1's planes), and the epoch in both step forms (IDELUCS_PLANES=0 / 1).
    python3 tools/bench_planes.py [--no-epoch]"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402


def timed(fn, n=20, reps=10):
    """us per launch, n launches captured in a HIP graph (the host's ~15 us a ctypes call would otherwise pace short kernels)"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            g.replay()
        e1.record(s)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3


def epochs(dev, flags):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import copy
    import test_gpu_encoder as E
    from idelucs_amd.fused import FusedLinearTrainer
    store, net0 = E._cfg2_store_and_net(dev, 33334, seed=6, C=20)               # 100 002 pairs: 195 full batches of 512, as cfg2
    for flag_ in flags:
        os.environ["IDELUCS_PLANES"] = flag_
        tr = FusedLinearTrainer(copy.deepcopy(net0), lr=1e-3, weight=0.25, lamb=2.8, seed=5)
        gen = torch.Generator(device=dev); gen.manual_seed(1)
        losses, ts = [], []
        for ep in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            total, nb = tr.run_epoch(store, 512, use_graph=True, generator=gen)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            losses.append(total.item() / (nb - 1))
        print(f"IDELUCS_PLANES={flag_}: epoch {min(ts[1:]):.2f} ms ({nb} batches: {min(ts[1:]) / nb * 1e3:.1f} us a step), losses "
              + " ".join(f"{l:.4f}" for l in losses) + (f", overflow {tr.planes_overflowed()}" if flag_ == "1" else ""))


def main():
    from idelucs_amd import _lib
    L = _lib.lib
    dev = torch.device("cuda")
    if "--epoch-only" in sys.argv:                      # (under rocprofv3: the step's kernels of ONE form; IDELUCS_PLANES from the environment)
        return epochs(dev, (os.environ.get("IDELUCS_PLANES", "0"),))
    g = torch.Generator(device="cpu"); g.manual_seed(5)
    m, H, F = 1024, 512, 4096
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    W = ((torch.rand(H, F, generator=g) * 2 - 1) / F ** 0.5).to(dev)
    x = torch.randn(m, F, generator=g).to(dev)
    ref = W.double() @ x.double().t()
    scale = ref.abs().max().item()
    h16 = lambda t: torch.empty(t.shape, dtype=torch.int16, device=dev)
    wh, wl, xh, xl = h16(W), h16(W), h16(x), h16(x)
    _lib.check(L.idl_split_planes(p(W), W.numel(), L.idl_planes_exponent(1), p(wh), p(wl), None, st()))
    _lib.check(L.idl_split_planes(p(x), x.numel(), L.idl_planes_exponent(0), p(xh), p(xl), None, st()))
    part = torch.empty(8, H, m, device=dev)
    r1T = torch.empty(H, m, device=dev)
    t_pl = timed(lambda: _lib.check(L.idl_l1_planes(p(wh), p(wl), F, p(xh), p(xl), F, m, H, F, p(part), st())))
    for pad in (8, 32, 64, 128, 256, 512):               # rows a power of two apart all fall into one L2 channel: what a padded pitch buys
        def padded(t):
            b = torch.zeros(t.shape[0], F + pad, dtype=torch.int16, device=dev); b[:, :F] = t
            return b
        pw, pl_, px, py = padded(wh), padded(wl), padded(xh), padded(xl)
        t_pad = timed(lambda: _lib.check(L.idl_l1_planes(p(pw), p(pl_), F + pad, p(px), p(py), F + pad, m, H, F, p(part), st())))
        torch.cuda.synchronize()
        print(f"   pitch {F} + {pad} elements: {t_pad:.1f} us (error {(part.double().sum(0) - ref).abs().max().item() / scale:.1e})")
    t_32 = timed(lambda: _lib.check(L.idl_l1_fwd(p(W), p(x), m, F, p(r1T), st())))
    t_sx = timed(lambda: _lib.check(L.idl_split_planes(p(x), x.numel(), 3, p(xh), p(xl), None, st())))
    torch.cuda.synchronize()
    e_pl = (part.double().sum(0) - ref).abs().max().item() / scale
    e_32 = (r1T.double() - ref).abs().max().item() / scale
    e_lib = ((W @ x.t()).double() - ref).abs().max().item() / scale
    print(f"layer-1 product 512 x {m} x {F}: two-plane tiles {t_pl:.1f} us (error {e_pl:.1e} of the largest entry), fp32 tiles {t_32:.1f} ({e_32:.1e}), "
          f"fp32 library GEMM error {e_lib:.1e}; idl_split_planes of a batch {t_sx:.1f} us")
    dy = (torch.randn(m, H, generator=g) * 1e-4).to(dev)
    hyper = torch.tensor([1e-3, 0.99, 1e-8, 0.01, 0.01], dtype=torch.float32, device=dev)
    Wc, Vc = W.clone(), torch.zeros_like(W)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    t_w = timed(lambda: _lib.check(L.idl_wgrad_rmsprop(p(dy), p(x), m, H, F, None, p(Wc), p(Vc), p(hyper), st())))
    t_wp = timed(lambda: _lib.check(L.idl_wgrad_rmsprop_planes(p(dy), p(x), m, H, F, None, p(Wc), p(Vc), p(hyper), p(wh), p(wl), p(flag), st())))
    dyh, dyl = h16(dy), h16(dy)
    kd = 9 - int(__import__("math").frexp(dy.abs().max().item())[1])
    _lib.check(L.idl_split_planes(p(dy), dy.numel(), kd, p(dyh), p(dyl), None, st()))
    dsc = torch.zeros(int(L.idl_dr1_scale_words()), dtype=torch.int32, device=dev); dsc[0] = kd
    grad = torch.empty(H, F, device=dev)
    t_d = timed(lambda: _lib.check(L.idl_wgrad_rmsprop_xplanes(p(dyh), p(dyl), p(dsc), p(xh), p(xl), F, m, H, F, None, p(Wc), p(Vc), p(hyper), p(wh), p(wl), p(flag), st())))
    t_dg = timed(lambda: _lib.check(L.idl_wgrad_rmsprop_xplanes(p(dyh), p(dyl), p(dsc), p(xh), p(xl), F, m, H, F, p(grad), None, None, None, None, None, None, st())))
    torch.cuda.synchronize()
    e_d = (grad.double() - dy.double().t() @ x.double()).abs().max().item() / (dy.double().t() @ x.double()).abs().max().item()
    print(f"dW1 tiles + RMSprop (fp32 tiles): {t_w:.1f} us; also writing W1's planes {t_wp:.1f}; from both operands' planes (fp16 matrix cores) {t_d:.1f}, its gradient alone {t_dg:.1f} "
          f"(error {e_d:.1e} of the largest entry)")
    if "--no-epoch" in sys.argv:
        return
    epochs(dev, ("0", "1"))


if __name__ == "__main__":
    main()
