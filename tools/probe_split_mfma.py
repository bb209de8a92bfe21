#!/usr/bin/env python3
"""What the step's big fp32 products would cost on the bf16 matrix cores with three-way split operands (tools/probe_split.hip: a probe, built here with hipcc and linked against the product library; not part of it):
    python3 tools/probe_split_mfma.py
C = A B^T at the layer-1 forward's shape (A = a standardised batch [1024, 4096], B = W1 [512, 4096]), operands split
x = x0 + x1 + x2 into bf16 planes on the device beforehand, 6 / 3 / 1 of the split products on v_mfma_f32_32x32x16_bf16, fp32
accumulators, 8-way split K (partial sums [8][M][N], summed by torch for the check).  Prints per variant: us per launch (HIP events
over 200 launches behind 20 warm-ups) and the error against a float64 product, beside the fp32 library GEMM's own time and error."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def build_probe():
    """hipcc tools/probe_split.hip -> /tmp/libidelucs_probe_split.so (linked against libidelucs_hip.so for its error text), loaded with ctypes."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    csrc = os.path.join(ROOT, "idelucs_amd", "csrc")
    out = "/tmp/libidelucs_probe_split.so"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", os.path.join(ROOT, "tools", "probe_split.hip"), "-o", out,
                    "-I", os.path.join(ROOT, "include"), "-L", csrc, "-lidelucs_hip", "-Wl,-rpath," + csrc], check=True)
    lib = ctypes.CDLL(out)
    lib.idl_debug_split_gemm.restype = ctypes.c_int
    return lib


def split3(x):
    x0 = x.to(torch.bfloat16)
    r = x - x0.float()
    x1 = r.to(torch.bfloat16)
    r2 = r - x1.float()
    x2 = r2.to(torch.bfloat16)
    return [t.contiguous() for t in (x0, x1, x2)]


def split2_f16(x):
    """x = 2^-k (x0 + x1), fp16 both (22 significand bits), the tensor scaled so that its largest entry sits near 2^12."""
    k = int(torch.floor(torch.log2(torch.tensor(4096.0) / x.abs().max().cpu())).item())
    xs = x * (2.0 ** k)
    x0 = xs.to(torch.float16)
    x1 = (xs - x0.float()).to(torch.float16)
    return [x0.contiguous(), x1.contiguous(), x1.contiguous()], k


def timed(fn, n=200, warm=20):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="fwd", choices=["fwd", "wgrad", "wgrad_t"],
                    help="fwd: 1024 x 512 x 4096, 8-way split K (a1 = x W1^T); wgrad: 512 x 4096 x 1024, 2-way split K (dW1 = dr1^T x, with both "
                         "operands given contraction-contiguous, i.e. TRANSPOSED copies of dr1 and x: their producers' problem); wgrad_t: the same product from dr1 [1024][512] and x [1024][4096] AS THEY LIE (transposed LDS reads)")
    args = ap.parse_args()
    from idelucs_amd import _lib
    L = _lib.lib
    PROBE = build_probe()
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu"); g.manual_seed(3)
    if args.shape in ("fwd", "wgrad_t"):
        M, N, K, S = 1024, 512, 4096, 8
        a = torch.randn(M, K, generator=g).to(dev)                       # a standardised batch
        b = (torch.randn(N, K, generator=g) * (2.0 / K) ** 0.5).to(dev)  # Kaiming-normal W1
    else:
        M, N, K, S = 512, 4096, 1024, 2
        a = (torch.randn(M, K, generator=g) * 1e-4 * torch.rand(1, K, generator=g) ** 4).to(dev)   # dr1^T: rows of very different size
        b = torch.randn(N, K, generator=g).to(dev)                                                # x^T
    tr = args.shape == "wgrad_t"
    if tr:
        M, N, K, S = 512, 4096, 1024, 2
        a = (torch.randn(K, M, generator=g) * 1e-4 * torch.rand(K, 1, generator=g) ** 4).to(dev)   # dr1: rows of very different size
        b = torch.randn(K, N, generator=g).to(dev)                                                # x
    ref = (a.double().t() @ b.double()) if tr else (a.double() @ b.double().t())
    scale = ref.abs().max().item()
    pa, pb = split3(a), split3(b)
    cpart = torch.empty(S, M, N, dtype=torch.float32, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(products):
        _lib.check(PROBE.idl_debug_split_gemm(*(ctypes.c_void_p(t.data_ptr()) for t in pa), *(ctypes.c_void_p(t.data_ptr()) for t in pb),
                                          ctypes.c_void_p(cpart.data_ptr()), M, N, K, S, products, st))

    out = torch.empty(M, N, dtype=torch.float32, device=dev)
    t_lib = timed(lambda: torch.mm(a.t(), b, out=out)) if tr else timed(lambda: torch.mm(a, b.t(), out=out))
    e_lib = ((out.double() - ref).abs().max().item() / scale, ((out.double() - ref) ** 2).mean().sqrt().item() / scale)
    print(f"fp32 library GEMM {M} x {N} x {K}: {t_lib:.1f} us; error against float64, relative to the largest entry: max {e_lib[0]:.2e}, rms {e_lib[1]:.2e}")
    pa16, ka = split2_f16(a)
    pb16, kb = split2_f16(b)
    bf_planes = (pa, pb)
    for products in ((64 + 48 + 3, 64 + 48 + 1) if tr else (6, 3, 1, 16 + 6, 16 + 3, 16 + 1, 48 + 3, 48 + 1)):
        if products & 32:
            pa, pb = pa16, pb16
        else:
            pa, pb = bf_planes
        run(products)
        torch.cuda.synchronize()
        c = cpart.double().sum(0) * (2.0 ** -(ka + kb) if products & 32 else 1.0)
        err = ((c - ref).abs().max().item() / scale, ((c - ref) ** 2).mean().sqrt().item() / scale)
        pcode = products
        t = timed(lambda: run(pcode))
        t_sum = timed(lambda: torch.sum(cpart, 0, out=out))
        form = "loader waves, 3 chunks resident" if products & 16 else "every wave loads and computes, 2 chunks"
        kind = ("fp16 MFMA, two planes" + (" as they lie (transposed LDS reads)," if products & 64 else ",")) if products & 32 else "bf16 MFMA,"
        products &= 15
        flops = 2.0 * M * N * K * products
        print(f"{kind} {products} split product(s) ({form}), {S}-way split K: {t:.1f} us = {flops / t / 1e6:.0f} TFLOP/s of 16-bit products "
              f"(+ {t_sum:.1f} us for a separate sum of the {S} partials); error max {err[0]:.2e}, rms {err[1]:.2e}")


if __name__ == "__main__":
    main()
