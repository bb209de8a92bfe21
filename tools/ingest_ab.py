#!/usr/bin/env python3
"""A/B runs of T_e2e's host stage: utils.build_feature_store(streamed=True) on the synthetic cfg2 FASTA (100 000 x 10 kbp, tmpfs)
under a list of environment settings, one child process per setting (most knobs are read once per process), with the reader's
own timeline (IDELUCS_INGEST_TIMING) of the best repetition.
   python tools/ingest_ab.py "IDELUCS_DEV=numa=off" "IDELUCS_DEV=reader_pool=0" "IDELUCS_DEV=copy_div=4" ...      (each argument: VAR=VAL[;VAR=VAL...]; "" = defaults)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import io
    import torch
    from time_ingest import write_fasta
    from idelucs_amd import utils as U
    path = os.environ["IDL_AB_FASTA"]
    dev = torch.device("cuda:0")
    reps = int(os.environ.get("IDL_AB_REPS", "6"))
    best, best_log, store = 1e9, "", None
    for rep in range(reps):
        U._L.idl_ingest_release()
        torch.cuda.synchronize()
        # the C++ side writes its timeline to fd 2: catch it per repetition
        r, w = os.pipe()
        saved = os.dup(2)
        os.dup2(w, 2)
        t0 = time.perf_counter()
        store = U.build_feature_store(path, 3, k=6, device=dev, streamed=True, reuse=store)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0)
        os.dup2(saved, 2); os.close(w); os.close(saved)
        log = os.read(r, 1 << 16).decode(errors="replace"); os.close(r)
        if rep and ms < best:
            best, best_log = ms, log
    print(f"{best:8.2f} ms ingest-to-features (best of {reps - 1}); threads {U.ingest_threads()}, node {U._L.idl_ingest_numa_node()}")
    for line in best_log.splitlines():
        print("      | " + line)


def main():
    if os.environ.get("IDL_AB_CHILD"):
        return child()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from time_ingest import write_fasta
    path = f"/dev/shm/idelucs_ab_{os.getpid()}.fas"
    node = os.environ.get("IDL_AB_FILE_NODE")          # write the file from a thread on that NUMA node: its page-cache pages land there
    if node is not None:
        cpus = []
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.extend(range(int(a), int(b or a) + 1))
        before = os.sched_getaffinity(0)
        os.sched_setaffinity(0, set(cpus) & before)
        write_fasta(path, 100000, 10000)
        os.sched_setaffinity(0, before)
        print(f"# the file's pages: written from NUMA node {node}", flush=True)
    else:
        write_fasta(path, 100000, 10000)
    try:
        for spec in (sys.argv[1:] or [""]):
            env = dict(os.environ, IDL_AB_CHILD="1", IDL_AB_FASTA=path, IDELUCS_INGEST_TIMING="1")
            for kv in filter(None, spec.split(";")):
                k, _, v = kv.partition("=")
                env[k] = v
            out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
            print(f"## {spec or '(defaults)'}")
            print(out.stdout.rstrip() or out.stderr[-2000:], flush=True)
    finally:
        os.unlink(path)


if __name__ == "__main__":
    main()
