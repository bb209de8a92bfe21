#!/usr/bin/env python3
"""Do several voters' epochs overlap when each runs its own captured step graph on its own HIP stream?

The small launches of one training step (InfoNCE passes, middle layers, optimizer) occupy a fraction of the 256 CUs each,
so a second voter's step could fill the rest.  Times V voter-epochs back to back on one stream against the same V epochs on
V streams (one model, one trainer and one graph per voter; the feature store is shared and read-only).

  python tools/concurrent_voters.py [--n 100000] [--voters 4]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100000)
    ap.add_argument("--voters", type=int, default=4)
    ap.add_argument("--k", type=int, default=6)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--n-clusters", dest="n_clusters", type=int, default=20)
    ap.add_argument("--cu-mask", dest="cu_mask", default="", help="block | strided: one slice of the CUs per lane")
    a = ap.parse_args()
    from idelucs_amd import models, utils as U
    dev = torch.device("cuda:0")
    P, F = 3, 4 ** a.k
    g = torch.Generator(device=dev).manual_seed(1)
    feats = torch.rand((P, a.n, F), device=dev, generator=g) * 1e-3
    mean, scale = U.col_stats(feats[0])
    store = U.FeatureStore(None, None, feats, mean, scale, a.k, False)

    def make(v):
        m = models.IID_model({'sequence_file': None, 'GT_file': None, 'n_clusters': a.n_clusters, 'k': a.k, 'model_size': 'linear',
                              'n_mimics': 3, 'batch_sz': 512, 'optimizer': 'RMSprop', 'lambda': 2.8, 'lr': 1e-3, 'weight': 0.25,
                              'scheduler': None, 'n_epochs': 1, 'n_voters': a.voters})
        m.store = store
        m.begin_voter(v)
        return m

    ms = [make(v) for v in range(a.voters)]
    if a.cu_mask:
        # each lane on its own slice of the CUs (hipExtStreamCreateWithCUMask): the GEMMs of the lanes then run side by side instead
        # of each one taking every CU in turn
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        n_cu = torch.cuda.get_device_properties(0).multi_processor_count
        words = (n_cu + 31) // 32
        streams = []
        for i in range(len(ms)):
            bits = 0
            if a.cu_mask == "block":
                for cu in range(i * n_cu // len(ms), (i + 1) * n_cu // len(ms)):
                    bits |= 1 << cu
            else:                                       # strided: CU j belongs to lane j mod L
                for cu in range(i, n_cu, len(ms)):
                    bits |= 1 << cu
            mask = (ctypes.c_uint32 * words)(*[(bits >> (32 * w)) & 0xFFFFFFFF for w in range(words)])
            h = ctypes.c_void_p()
            rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), words, mask)
            assert rc == 0, rc
            streams.append(torch.cuda.ExternalStream(h.value))
    else:
        streams = [torch.cuda.Stream() for _ in ms]
    for m in ms:                                   # capture every voter's graph
        m.contrastive_training_epoch(sync=False)
    torch.cuda.synchronize()

    def seq():
        return [m.contrastive_training_epoch(sync=False) for m in ms]

    def conc():
        cur = torch.cuda.current_stream()
        out = []
        for m, s in zip(ms, streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                out.append(m.contrastive_training_epoch(sync=False))
        for s in streams:
            cur.wait_stream(s)
        return out

    # the same voters batched inside the launches (fused.BatchedLinearTrainer): one launch sequence for all of them
    from idelucs_amd.fused import BatchedLinearTrainer
    bms = [make(v) for v in range(a.voters)]
    modes = [("sequential", seq), ("concurrent", conc)]
    if a.n_clusters <= 48 and a.voters > 1:
        bt = BatchedLinearTrainer([m.net for m in bms], 1e-3, 0.25, 2.8, seed=0)
        for m, t in zip(bms, bt.trainers):
            m._fused = t
        for v, m in enumerate(bms):
            m.begin_voter(v)

        def batched():
            res = bt.run_epoch(store, 512, [m._gen for m in bms])
            return [tot / (nb - 1) for tot, nb in res]
        batched()
        torch.cuda.synchronize()
        modes.append(("batched", batched))
    for name, fn in modes + modes:
        ts = []
        for _ in range(a.reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            losses = fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print(f"{name}: {a.voters} voter-epochs in {min(ts) * 1e3:.1f} ms (min of {a.reps}); losses {[round(float(x), 4) for x in losses]}", flush=True)


if __name__ == "__main__":
    main()
