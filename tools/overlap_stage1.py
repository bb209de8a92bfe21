#!/usr/bin/env python3
"""Do the mimic generator and the vectoriser overlap when they are launched on two streams?  (cfg2: 100 000 x 10 kbp, k = 6)
Times each alone, back to back on one stream, and concurrently on two streams (the vectoriser reads the edits of an earlier run)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    import bench
    from idelucs_amd import _lib, utils as U
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    n, L, k = 100000, 10000, 6
    din = bench.synth_packed(n, L, dev)
    specs = [t.spec() for t in U.mimic_transforms(3)]
    P = len(specs)
    feats = torch.empty((P, n, 4 ** k), dtype=torch.float32, device=dev)
    expect = sum(n * (L * (1.0 - (1.0 - s[0]) * (1.0 - s[1])) + s[2]) for s in specs)
    cap = int(1.25 * expect + 64 * n * P + 1024)

    def gen(seed):
        return U._philox_edits(din, specs, seed, capacity=cap)

    def vec(edits, off):
        U._vectorise(din, k, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_FREQ_F32, P, edits, off, feats)

    e0, o0 = gen(1)
    vec(e0, o0)
    torch.cuda.synchronize()

    def timed(fn, reps=5):
        ts = []
        for _ in range(reps):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            s.record(); fn(); e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e))
        return sorted(ts)[len(ts) // 2]

    side = torch.cuda.Stream(device=dev)

    def both():
        main_s = torch.cuda.current_stream()
        side.wait_stream(main_s)
        with torch.cuda.stream(side):
            gen(2)
        vec(e0, o0)
        main_s.wait_stream(side)

    print(f"generator alone {timed(lambda: gen(2)):.3f} ms, vectoriser alone {timed(lambda: vec(e0, o0)):.3f} ms, "
          f"back to back {timed(lambda: (gen(2), vec(e0, o0))):.3f} ms, on two streams {timed(both):.3f} ms")


if __name__ == "__main__":
    main()
