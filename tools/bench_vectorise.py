#!/usr/bin/env python3
"""Micro-benchmark of the stage-1 kernels (mimic edit generation, vectorise, scaler stats, gather) on
synthetic packed input resident in HBM.  Prints per-stage ms and algorithmic GB/s.
  python tools/bench_vectorise.py [--n 100000] [--len 10000] [--k 6] [--views 4] [--reps 5]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from idelucs_amd import _lib, utils as U


def synth_input(n, L, dev, seed=12345):
    slots = (L + 63) // 64
    g = torch.Generator(device=dev); g.manual_seed(seed)
    codes = torch.randint(-2 ** 31, 2 ** 31 - 1, (n * slots * 4,), dtype=torch.int32, device=dev, generator=g)
    mask = torch.zeros((n, slots, 2), dtype=torch.int32, device=dev)
    tail = L % 64
    if tail:
        w = [0, 0]
        for j in range(tail, 64):
            w[j // 32] |= 1 << (31 - (j % 32))
        for i in (0, 1):
            mask[:, -1, i] = w[i] - (1 << 32) if w[i] >= 2 ** 31 else w[i]
    class D: pass
    d = D(); d.n = n; d.codes = codes; d.mask = mask.view(-1)
    d.slot_off = torch.arange(0, (n + 1) * slots, slots, dtype=torch.int64, device=dev)
    d.lengths = torch.full((n,), L, dtype=torch.int64, device=dev)
    d.max_len = L
    return d


def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return min(ts), sum(ts) / len(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100000); ap.add_argument("--len", type=int, default=10000)
    ap.add_argument("--k", type=int, default=6); ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda")
    din = synth_input(a.n, a.len, dev)
    F = 4 ** a.k
    specs = [t.spec() for t in U.mimic_transforms(a.views - 1)][:a.views]
    t0 = time.time(); edits, edit_off = U._philox_edits(din, specs, 7); torch.cuda.synchronize()
    print(f"edits: {edits.numel()} total ({edits.numel() / (a.n * a.views):.1f} per view-seq), first call {1e3 * (time.time() - t0):.1f} ms")
    mn, av = timeit(lambda: U._philox_edits(din, specs, 7), a.reps)
    print(f"mimic_edits (count+scan+fill)   min {mn:8.3f} ms  avg {av:8.3f} ms")
    out = torch.empty((a.views, a.n, F), dtype=torch.float32, device=dev)
    bytes_alg = a.n * ((a.len + 3) // 4 + a.views * F * 4)
    for label, e, eo in (("no edits", None, None), ("philox edits", edits, edit_off)):
        mn, av = timeit(lambda: U._vectorise(din, a.k, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_FREQ_F32, a.views, e, eo, out), a.reps)
        print(f"vectorise f32 x{a.views} views [{label:12s}] min {mn:8.3f} ms  avg {av:8.3f} ms  -> {bytes_alg / mn / 1e6:8.1f} GB/s algorithmic, "
              f"{a.n / mn * 1e3:,.0f} seq/s")
    mn, av = timeit(lambda: U._vectorise(din, a.k, _lib.MODE_KMER, _lib.INIT_ZERO, _lib.OUT_COUNTS_I32, 1, None, None, out.view(torch.int32)[:1]), a.reps)
    print(f"vectorise i32 x1 view                 min {mn:8.3f} ms  avg {av:8.3f} ms  -> {a.n * ((a.len + 3) // 4 + F * 4) / mn / 1e6:8.1f} GB/s")
    for mname, mode in (("cgr", _lib.MODE_CGR), ("canonical", _lib.MODE_CANONICAL)):       # (k = 4, 5: vectorise4_kernel's epilogues; otherwise the delta-view kernel v2)
        rl = int(_lib.lib.idl_row_len(mode, a.k))
        o2 = out.view(-1)[:a.views * a.n * rl].view(a.views, a.n, rl)
        mn, av = timeit(lambda: U._vectorise(din, a.k, mode, _lib.INIT_ONE, _lib.OUT_FREQ_F32, a.views, edits, edit_off, o2), a.reps)
        print(f"vectorise {mname:9s} f32 x{a.views} views [philox edits] min {mn:8.3f} ms  avg {av:8.3f} ms")
    mn, av = timeit(lambda: U.col_stats(out[0]), a.reps)
    print(f"col_stats (one pass over view 0)        min {mn:8.3f} ms  avg {av:8.3f} ms  -> {2 * a.n * F * 4 / mn / 1e6:8.1f} GB/s")
    mean, scale = U.col_stats(out[0])
    st = U.FeatureStore(None, None, out, mean, scale, a.k, False)
    idx = torch.randperm(st.n_pairs, device=dev)[:512]
    y = torch.empty((1024, F), dtype=torch.float32, device=dev)
    mn, av = timeit(lambda: st.gather_pairs(idx, y), a.reps * 4)
    print(f"gather_pairs B=512                    min {mn:8.3f} ms  avg {av:8.3f} ms  -> {2 * 1024 * F * 4 / mn / 1e6:8.1f} GB/s")


if __name__ == "__main__":
    main()
