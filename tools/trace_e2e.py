#!/usr/bin/env python3
"""T_e2e alone (bench.t_e2e), for a kernel trace of the hand-over between the ingest and the epoch:
    rocprofv3 --kernel-trace -f csv -d /tmp/e2e -o e2e -- python3 tools/trace_e2e.py
    python3 tools/trace_e2e.py --report /tmp/e2e/e2e_kernel_trace.csv
The report lists, for the LAST repetition, every launch from the site generator to the 12th launch of the epoch and the last
three of the epoch, with the idle time in front of each."""
import argparse
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def short(name):
    """`void (anonymous namespace)::foo<...>(...)` -> `foo`; torch's elementwise kernels keep their functor's name."""
    import re
    m = re.search(r"\w+Functor\w*", name) or re.search(r"\w+_kernel\w*", name)
    if m:
        return m.group(0)
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return re.split(r"[<(]", n)[0][-50:] or name[:50]


def report(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if "mask_from_lengths" in r[2]]
    i0 = starts[-1] if starts else 0
    t0 = rows[i0][0]
    last_end = rows[i0][0]
    print(f"{'start us':>10} {'idle us':>8} {'dur us':>8}  kernel")
    n_epoch = 0
    tail = rows[-3:]
    for r in rows[i0:]:
        name = short(r[2])
        show = n_epoch < 14 or r in tail
        if show:
            print(f"{(r[0] - t0) / 1e3:10.1f} {(r[0] - last_end) / 1e3:8.1f} {(r[1] - r[0]) / 1e3:8.1f}  {name}")
        if "l1_rms" in r[2] or "wgrad" in r[2]:
            n_epoch += 1
        last_end = max(last_end, r[1])
    print(f"first launch -> last end: {(last_end - t0) / 1e3:.1f} us")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--report")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--sync-debug", action="store_true", help="torch.cuda.set_sync_debug_mode('warn'): every host wait, with its stack")
    a = ap.parse_args()
    if a.report:
        return report(a.report)
    import torch
    import bench
    args = argparse.Namespace(n=100000, len=10000, k=6, n_clusters=20, n_mimics=3, batch_sz=512)
    if a.sync_debug:
        import traceback
        import warnings
        torch.cuda.set_sync_debug_mode("warn")

        def show(message, category, filename, lineno, file=None, line=None):
            print(f"SYNC: {message}", file=sys.stderr)
            traceback.print_stack(limit=9, file=sys.stderr)
        warnings.showwarning = show
        warnings.simplefilter("always")
    r = bench.t_e2e(args, torch.device("cuda:0"), reps=a.reps)
    print(r)


if __name__ == "__main__":
    main()
