#!/usr/bin/env python3
"""The weight gradient + RMSprop launch on the fp16 matrix cores from split operands (csrc/wgrad_split.hip) beside the fp32 tiles
(idl_wgrad_rmsprop), at the step's shape: time per launch and the gradient's error against a float64 product.
    python3 tools/bench_wgrad_split.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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


def main():
    from idelucs_amd import _lib
    L = _lib.lib
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu"); g.manual_seed(5)
    m, H, F = 1024, 512, 4096
    dy = (torch.randn(m, H, generator=g) * 1e-4 * torch.rand(m, 1, generator=g) ** 4).to(dev)
    dy = dy * (torch.rand(m, H, generator=g).to(dev) > 0.5)            # ReLU / dropout zeros
    x = torch.randn(m, F, generator=g).to(dev)
    ref = dy.double().t() @ x.double()
    scale = ref.abs().max().item()
    hyper = torch.tensor([1e-3, 0.99, 1e-8, 0.01, 0.01], dtype=torch.float32, device=dev)
    W0 = (torch.randn(H, F, generator=g) * (2.0 / F) ** 0.5).to(dev)
    stream_of = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    state = torch.zeros(L.idl_wgrad_split_state_words(), dtype=torch.int64, device=dev)
    out = {}
    for name in ("fp32 tiles", "split fp16"):
        grad = torch.empty(H, F, dtype=torch.float32, device=dev)
        W, V = W0.clone(), torch.zeros_like(W0)

        def run(with_update=True, with_grad=True):
            if name == "fp32 tiles":
                _lib.check(L.idl_wgrad_rmsprop(p(dy), p(x), m, H, F, p(grad) if with_grad else None, p(W) if with_update else None,
                                               p(V) if with_update else None, p(hyper), stream_of()))
            else:
                _lib.check(L.idl_wgrad_rmsprop_split(p(dy), p(x), m, H, F, p(grad) if with_grad else None, p(W) if with_update else None,
                                                     p(V) if with_update else None, p(hyper), None, p(state), stream_of()))
        run()
        torch.cuda.synchronize()
        err = (grad.double() - ref).abs()
        out[name] = (grad.clone(), W.clone(), V.clone())
        t_full = timed(lambda: run(True, False))
        t_grad = timed(lambda: run(False, True))
        print(f"{name}: update in the epilogue {t_full:.1f} us a launch, gradient only {t_grad:.1f}; gradient against float64, relative to the largest "
              f"entry: max {err.max().item() / scale:.2e}, rms {(err ** 2).mean().sqrt().item() / scale:.2e}")
    if int(os.environ.get("IDELUCS_WGS_DBG", "0")) & 128:
        o = state[-8:].cpu().tolist()
        n = max(o[4], 1)
        print(f"loader wave 0 of workgroup 0, shader cycles a chunk over {o[4]} chunks: wait for the requests {o[0] / n:.0f}, split + LDS stores issued {o[1] / n:.0f}, "
              f"next requests + LDS stores done {o[2] / n:.0f}, barrier {o[3] / n:.0f}")
    (g0, w0, v0), (g1, w1, v1) = out["fp32 tiles"], out["split fp16"]
    print(f"after one update: max |W_split - W_fp32| {(w1 - w0).abs().max().item():.3e} (lr 1e-3), max |V_split - V_fp32| / max V {((v1 - v0).abs().max() / v0.abs().max()).item():.3e}")


if __name__ == "__main__":
    main()
