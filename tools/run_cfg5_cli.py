#!/usr/bin/env python3
"""BASELINE cfg5 through the CLI at full size: 10^6 sequences x 5 kbp (a 5 GB FASTA of planted families, written here),
--n_clusters 0 => 200 output units, exact HDBSCAN of the 10^6 x 64 latent on the GPU, metrics, TSVs.
  python tools/run_cfg5_cli.py [--n 1000000] [--epochs 2] [--families 8]"""
import argparse
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def write_family_fasta(path, gt_path, n, L, n_families, seed):
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    anc = rng.integers(0, 4, size=(n_families, L), dtype=np.uint8)
    fam = rng.integers(0, n_families, n)
    with open(path, "wb") as f, open(gt_path, "w") as g:
        g.write("sequence_id\tcluster_id\n")
        for lo in range(0, n, 10000):
            hi = min(lo + 10000, n)
            codes = anc[fam[lo:hi]]
            mut = rng.random(codes.shape) < 0.03
            codes = np.where(mut, (codes + rng.integers(1, 4, size=codes.shape, dtype=np.uint8)) & 3, codes)
            rec = np.empty((hi - lo, 12 + L + 1), np.uint8)
            rec[:, :12] = np.frombuffer(b"".join(b">seq%07d\n" % i for i in range(lo, hi)), np.uint8).reshape(hi - lo, 12)
            rec[:, 12:-1] = acgt[codes]
            rec[:, -1] = 10
            f.write(rec.tobytes())
            g.write("".join("seq%07d\tfam%d\n" % (i, fam[i]) for i in range(lo, hi)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1000000)
    ap.add_argument("--len", type=int, default=5000)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--families", type=int, default=8)
    ap.add_argument("--n-clusters", dest="n_clusters", type=int, default=0, help="0 = fine-grained mode (cfg5); e.g. 20 with --voters 5 = a cfg2/cfg3-like run")
    ap.add_argument("--voters", type=int, default=1)
    ap.add_argument("--save-latent", default=None, help="write the latent that goes into the fine-grained clustering here (.npy, float32)")
    a = ap.parse_args()
    base = "/dev/shm" if os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    work = tempfile.mkdtemp(prefix="idelucs_cfg5_", dir=base)
    fas, gt = os.path.join(work, "fam.fas"), os.path.join(work, "fam_GT.tsv")
    t0 = time.time()
    write_family_fasta(fas, gt, a.n, a.len, a.families, seed=11)
    print(f"wrote {os.path.getsize(fas) / 1e9:.2f} GB FASTA in {time.time() - t0:.0f} s", flush=True)
    from idelucs_amd.__main__ import main as cli
    if a.save_latent:
        from idelucs_amd import posthoc
        inner = posthoc.fine_grained_clusters

        def saving(latent, *args, **kw):
            np.save(a.save_latent, np.asarray(latent, dtype=np.float32))
            return inner(latent, *args, **kw)
        posthoc.fine_grained_clusters = saving
    import pandas as pd
    cwd = os.getcwd()
    os.chdir(work)
    try:
        t0 = time.time()
        out_dir = cli(["--sequence_file", fas, "--GT_file", gt, "--n_clusters", str(a.n_clusters), "--n_epochs", str(a.epochs),
                       "--n_voters", str(a.voters), "--batch_sz", "512", "--k", "6"])
        wall = time.time() - t0
        m = pd.read_csv(os.path.join(out_dir, "metrics.tsv"), sep="\t", index_col=0)
        df = pd.read_csv(os.path.join(out_dir, "assignments.tsv"), sep="\t", index_col=0)
        print(f"\nthrough the CLI ({a.n} x {a.len} bp, n_clusters {a.n_clusters}, {a.voters} voter(s), {a.epochs} epochs): {len(df)} sequences, {df['assignment'].nunique()} clusters, wall {wall:.0f} s")
        print(m.to_string())
    finally:
        os.chdir(cwd)
        for root, dirs, files in os.walk(work, topdown=False):
            for fn in files:
                os.unlink(os.path.join(root, fn))
            for dn in dirs:
                os.rmdir(os.path.join(root, dn))
        os.rmdir(work)


if __name__ == "__main__":
    main()
