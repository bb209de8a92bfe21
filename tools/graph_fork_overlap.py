#!/usr/bin/env python3
"""Does a memory-streaming kernel overlap with a big GEMM when it sits on a forked branch of the same captured graph?
(diagnostic for moving the batch assembly of the batched-voter step under the batched GEMMs)"""
import time
import torch
dev = torch.device("cuda:0")
L, m, F, H = 4, 1024, 4096, 512
W = torch.randn(L, H, F, device=dev); X = torch.randn(L, m, F, device=dev); R = torch.empty(L, H, m, device=dev)
src = torch.randn(L * 2, m, F, device=dev); dst = torch.empty_like(src)          # 134 MB read + 134 MB written


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def capture(body):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            body()
    return g


side = torch.cuda.Stream()


def gemm_only():
    torch.bmm(W, X.transpose(1, 2), out=R)


def copy_only():
    dst.copy_(src)


def serial():
    gemm_only(); copy_only()


def forked():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        copy_only()
    gemm_only()
    cur.wait_stream(side)


for name, body in (("gemm only", gemm_only), ("copy only", copy_only), ("serial", serial), ("forked", forked)):
    g = capture(body)
    print(f"{name:10s}: {timeit(g.replay) / 10:8.1f} us per iteration (graph of 10)")
for name, body in (("serial eager", serial), ("forked eager", forked)):
    print(f"{name:12s}: {timeit(body):8.1f} us")
