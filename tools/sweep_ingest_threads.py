#!/usr/bin/env python3
"""T_e2e's host stage against the number of reader threads: the one-pass FASTA reader (idl_fasta_parse_pack:
validate + count + 2-bit pack + H2D in flight) on the synthetic cfg2 file (100 000 x 10 kbp, tmpfs) for IDELUCS_THREADS in a
sweep -- (a) parse + pack into pinned host arenas only, (b) with the device copies, (c) the whole of
utils.build_feature_store(streamed=True) = ingest-to-features -- best of `reps`, plus what the box gives this process (affinity,
cgroup CPU quota, NUMA nodes).   python tools/sweep_ingest_threads.py > profiles/r04_ingest_threads.txt"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from time_ingest import write_fasta


def box():
    out = [f"os.cpu_count() {os.cpu_count()}, affinity {len(os.sched_getaffinity(0))} cpus"]
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
        try:
            out.append(f"{p}: {open(p).read().strip()}")
        except OSError:
            pass
    try:
        nodes = sorted(d for d in os.listdir("/sys/devices/system/node") if d.startswith("node"))
        out.append(f"NUMA nodes: {len(nodes)}")
    except OSError:
        pass
    return "; ".join(out)


def main():
    from idelucs_amd import utils as U, _lib
    L = _lib.lib
    dev = torch.device("cuda:0")
    n, length, reps = 100000, 10000, 4
    path = f"/dev/shm/idelucs_sweep_{os.getpid()}.fas"
    write_fasta(path, n, length)
    size = os.path.getsize(path)
    print(f"# {box()}")
    print(f"# file {size / 1e9:.3f} GB in /dev/shm; default threads = {U.ingest_threads()}")
    print(f"# {'threads':>7s} {'parse+pack (host arenas) ms':>28s} {'with H2D ms':>12s} {'ingest-to-features ms':>22s}")
    try:
        sweep = [int(x) for x in os.environ.get("SWEEP", "4,8,12,16,24,32,48,64").split(",")]
        for T in sweep:
            os.environ["IDELUCS_THREADS"] = str(T)
            cap = size // 48 + 4096 * T + 1024
            hc = torch.empty(cap * 16, dtype=torch.uint8, pin_memory=True); hm = torch.empty(cap * 8, dtype=torch.uint8, pin_memory=True)
            dc = torch.empty(cap * 16, dtype=torch.uint8, device=dev); dm = torch.empty(cap * 8, dtype=torch.uint8, device=dev)
            copy = torch.cuda.Stream(device=dev)
            best = [1e9, 1e9, 1e9]
            for rep in range(reps):
                for arm in (0, 1):
                    L.idl_ingest_release()
                    h = ctypes.c_void_p()
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    rc = L.idl_fasta_parse_pack(os.fsencode(path), U._ptr(hc), U._ptr(hm), cap, U._ptr(dc) if arm else None, U._ptr(dm) if arm else None,
                                                ctypes.c_void_p(copy.cuda_stream) if arm else None, ctypes.byref(h))
                    _lib.check(rc)
                    torch.cuda.synchronize(); t1 = time.perf_counter()
                    L.idl_fasta_close(h)
                    if rep:
                        best[arm] = min(best[arm], 1e3 * (t1 - t0))
                L.idl_ingest_release()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                st = U.build_feature_store(path, 3, k=6, device=dev, streamed=True)
                torch.cuda.synchronize(); t1 = time.perf_counter()
                del st
                if rep:
                    best[2] = min(best[2], 1e3 * (t1 - t0))
            print(f"  {T:7d} {best[0]:28.1f} {best[1]:12.1f} {best[2]:22.1f}", flush=True)
            del hc, hm, dc, dm
    finally:
        os.unlink(path)


if __name__ == "__main__":
    main()
