#!/usr/bin/env python3
"""Where the host side of T_e2e goes: the stages of utils.build_feature_store(streamed=True) timed one by one on the synthetic
cfg2 FASTA (100 000 x 10 kbp, one line per sequence, tmpfs).   python tools/time_ingest.py [--n 100000] [--len 10000]"""
import argparse
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def write_fasta(path, n, L):
    rng = np.random.default_rng(12345)
    with open(path, "wb") as f:
        for i0 in range(0, n, 2000):
            nb = min(2000, n - i0)
            blk = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=(nb, L), dtype=np.uint8)]
            rec = np.empty((nb, 11 + L + 1), np.uint8)
            rec[:, :11] = np.frombuffer(b"".join(b">seq%06d\n" % (i0 + j) for j in range(nb)), np.uint8).reshape(nb, 11)
            rec[:, 11:-1] = blk
            rec[:, -1] = 10
            f.write(rec.tobytes())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100000)
    ap.add_argument("--len", type=int, default=10000)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    from idelucs_amd import utils as U, _lib
    L = _lib.lib
    dev = torch.device("cuda:0")
    path = f"/dev/shm/idelucs_time_ingest_{os.getpid()}.fas"
    write_fasta(path, a.n, a.len)
    try:
        for rep in range(a.reps):
            t = [time.perf_counter()]

            def lap():
                torch.cuda.synchronize()
                t.append(time.perf_counter())
            h = ctypes.c_void_p()
            _lib.check(L.idl_fasta_open(os.fsencode(path), 1, ctypes.byref(h)))
            L.idl_fasta_close(h)
            lap()                                                   # 1: idl_fasta_open alone (mmap, scan, validate + count)
            ff = U.FastaFile(path, check=True, pack="deferred")
            lap()                                                   # 2: FastaFile (open again + export + names as Python strings)
            din = U._StreamedInput(ff, dev)
            lap()                                                   # 3: buffers (pinned + device), lengths to the device
            edits, edit_off = U._philox_edits(din, [x.spec() for x in U.mimic_transforms(3)], 0)
            lap()                                                   # 4: mimic sites
            for lo, hi in zip(din._cuts[:-1], din._cuts[1:]):
                ff.pack_range(lo, hi, din._hc, din._hm)
            lap()                                                   # 5: packing alone (all chunks, no copies)
            din.fill()
            lap()                                                   # 6: fill = packing + H2D overlapped
            if rep == a.reps - 1:                                   # where fill's time goes: host time of each call
                tp = tc = 0.0
                for lo, hi in zip(din._cuts[:-1], din._cuts[1:]):
                    q0 = time.perf_counter()
                    ff.pack_range(lo, hi, din._hc, din._hm)
                    q1 = time.perf_counter()
                    x, y = int(ff.slot_off[lo]), int(ff.slot_off[hi])
                    with torch.cuda.stream(din._copy):
                        din.codes[x * 16:y * 16].copy_(din._hc[x * 16:y * 16], non_blocking=True)
                        din.mask[x * 8:y * 8].copy_(din._hm[x * 8:y * 8], non_blocking=True)
                    q2 = time.perf_counter()
                    tp += q1 - q0
                    tc += q2 - q1
                torch.cuda.synchronize()
                q3 = time.perf_counter()
                print(f"   fill again, host time: pack calls {1e3 * tp:.1f} ms, copy calls {1e3 * tc:.1f} ms, final wait {1e3 * (q3 - q2):.1f} ms ({len(din._cuts) - 1} chunks)")
                t[-1] = time.perf_counter()
            ff.close()
            lap()                                                   # 7: close (munmap of the file)
            feats = U._vectorise(din, 6, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_FREQ_F32, 4, edits, edit_off)
            lap()                                                   # 8: vectorise (+ 6.5 GB output allocation)
            U.col_stats(feats[0])
            lap()                                                   # 9
            names = ("idl_fasta_open", "FastaFile", "buffers", "mimic sites", "pack only", "fill (pack + H2D)", "close", "vectorise", "col_stats")
            print(f"rep {rep}: " + ", ".join(f"{nm} {1e3 * (t[i + 1] - t[i]):.1f}" for i, nm in enumerate(names)) + " ms", flush=True)
            del feats, din, edits, edit_off, ff
            for one_pass in ("0", "1"):                             # the whole of build_feature_store(streamed=True), both readers
                U.OPTIONS["one_pass"] = one_pass
                torch.cuda.synchronize(); q0 = time.perf_counter()
                st = U.build_feature_store(path, 3, k=6, device=dev, streamed=True)
                torch.cuda.synchronize(); q1 = time.perf_counter()
                print(f"   build_feature_store(streamed) with one_pass={one_pass}: {1e3 * (q1 - q0):.1f} ms", flush=True)
                del st
    finally:
        os.unlink(path)


if __name__ == "__main__":
    main()
