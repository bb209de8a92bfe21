#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference, Cython, gcc).  It
  1. copies the reference package to a scratch dir under /tmp and compiles its
     Cython module there (nothing from the reference is written into this repo),
  2. imports that scratch copy,
  3. drives the hot-path functions on the data fixtures in tests/data/ and on
     seeded synthetic inputs,
  4. writes inputs + expected outputs (data only) as .npz/.json next to this file.

The fixtures pin oracle/ (tests/test_oracle_*.py) and, through it and directly,
the HIP path (tests/test_gpu_*.py).  /root/reference is never read by any test.

Usage:  python tests/golden/make_golden.py        (takes ~1-2 min)
"""
import hashlib
import json
import os
import random
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(os.path.dirname(HERE), "data")
REF = "/root/reference"
SCRATCH = "/tmp/idelucs_ref_build"


def build_reference():
    if os.path.isdir(SCRATCH):
        shutil.rmtree(SCRATCH)
    os.makedirs(os.path.join(SCRATCH, "idelucs"))
    for fn in os.listdir(os.path.join(REF, "idelucs")):
        if fn.endswith(".py") or fn.endswith(".pyx"):
            shutil.copy(os.path.join(REF, "idelucs", fn), os.path.join(SCRATCH, "idelucs", fn))
    with open(os.path.join(SCRATCH, "setup_ref.py"), "w") as f:
        f.write(
            "from setuptools import setup, Extension\n"
            "from Cython.Build import cythonize\n"
            "setup(ext_modules=cythonize([Extension('idelucs.kmers', ['idelucs/kmers.pyx'])]))\n"
        )
    subprocess.run([sys.executable, "setup_ref.py", "build_ext", "--inplace"], cwd=SCRATCH,
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sys.path.insert(0, SCRATCH)


def sha16(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def read_records_like_reference(fname):
    """(id, cleaned bytearray) per record, using the reference's own check_sequence."""
    from idelucs.utils import check_sequence
    out, lines, seq_id = [], [], ""
    for line in open(fname, "rb"):
        if line.startswith(b"#"):
            continue
        if line.startswith(b">"):
            if seq_id != "":
                out.append((seq_id, check_sequence(seq_id, bytearray().join(lines))))
                lines = []
            seq_id = line[1:-1].decode()
        else:
            lines.append(line.strip())
    out.append((seq_id, check_sequence(seq_id, bytearray().join(lines))))
    return out


def main():
    build_reference()
    import torch
    import idelucs  # noqa: F401  (the reference, from SCRATCH)
    from idelucs.kmers import kmer_counts, cgr
    from idelucs import utils as U
    from idelucs import models as M
    from idelucs.LossFunctions import IID_loss, info_nce_loss
    from idelucs.PytorchUtils import NetLinear, myNet

    # ------------------------------------------------------------------ G0: KATs
    kat = {"cases": []}
    for s in [b"ACGTNACGTTGCA", b"", b"A", b"acgt", b"NNNN", b"ACG-T", b"ACGTACGT", b"AAAAAAA",
              b"ACGTNNACGTACGTNACG", b"TTTTTTTTTTTTTTTTTTTTTTTTT", b"ACGTXYZACGTACGT@ACGTT"]:
        for k in (1, 2, 3, 4, 5, 6, 7):
            c = np.zeros(4 ** k, np.int32)
            kmer_counts(bytearray(s), k, c)
            g = np.zeros(4 ** k, np.int32)
            cgr(bytearray(s), k, g)
            kat["cases"].append({"seq": s.decode("latin1"), "k": k,
                                 "kmer": np.flatnonzero(c).tolist(), "kmer_v": c[c != 0].tolist(),
                                 "cgr": np.flatnonzero(g).tolist(), "cgr_v": g[g != 0].tolist()})
    # accumulate semantics: counts are added on top of what the caller passes
    c = np.full(16, 7, np.int32)
    kmer_counts(bytearray(b"ACGTNACGTTGCA"), 2, c)
    kat["accumulate_k2_from7"] = c.tolist()
    # reverse_complement / kmer_rev_comp
    kat["revcomp"] = {str(k): [U.reverse_complement(x, k) for x in range(4 ** k)] for k in (1, 2, 3)}
    kat["revcomp_k6_sha16"] = sha16(np.array([U.reverse_complement(x, 6) for x in range(4096)], np.int32))
    c = np.zeros(16, np.int32); c[0] = 3; c[15] = 2
    kat["kmer_rev_comp_k2_trunc"] = U.kmer_rev_comp(c.copy(), 2).tolist()
    kat["canonical_len"] = {}
    for k in (2, 3, 4, 5, 6):
        kat["canonical_len"][str(k)] = int(len(U.kmer_rev_comp(np.ones(4 ** k, np.int32), k)))
    # check_sequence behaviour
    cs = []
    for hdr, s in [("h", b"acgtuUswkmyrbdhvnSWKMYRBDHV-ACGTN \t\r\n"), ("h", b""), ("h", b"ACGT ACGT")]:
        cs.append({"header": hdr, "seq": s.decode("latin1"), "out": bytes(U.check_sequence(hdr, bytearray(s))).decode()})
    errs = []
    for hdr, s in [("h", b"ACGTXACGT"), (">h", b"ACGT"), ("#h", b"ACGT"), (" h", b"ACGT"), ("a\tb", b"ACGT"), ("h", b"ACG*T")]:
        try:
            U.check_sequence(hdr, bytearray(s))
            errs.append({"header": hdr, "seq": s.decode(), "error": None})
        except ValueError as e:
            errs.append({"header": hdr, "seq": s.decode(), "error": str(e)})
    kat["check_sequence"] = cs
    kat["check_sequence_errors"] = errs
    json.dump(kat, open(os.path.join(HERE, "kat.json"), "w"), indent=0)

    # ------------------------------------------------ G1/G2: per-file count fixtures
    hashes = {}
    for name in ["edge", "edge_nonl", "empty", "influenza_64", "actino_8"]:
        fn = os.path.join(DATA, name + ".fas")
        recs = read_records_like_reference(fn)
        out = {"names": np.array([r[0] for r in recs]), "lengths": np.array([len(r[1]) for r in recs], np.int64)}
        for k in (4, 5, 6):
            km = np.zeros((len(recs), 4 ** k), np.int32)
            cg = np.zeros((len(recs), 4 ** k), np.int32)
            red = []
            for i, (_, s) in enumerate(recs):
                kmer_counts(bytearray(s), k, km[i])
                cgr(bytearray(s), k, cg[i])
                red.append(U.kmer_rev_comp(km[i].copy() + 1, k))
            out[f"kmer_k{k}"] = km
            out[f"cgr_k{k}"] = cg
            out[f"canon_k{k}"] = np.array(red, np.int32)
            names, f64 = U.kmersFasta(fn, k=k)
            assert list(names) == [r[0] for r in recs]
            out[f"freq_k{k}"] = f64
            _, f64r = U.kmersFasta(fn, k=k, reduce=True)
            out[f"freq_canon_k{k}"] = f64r
            if name != "empty":
                _, c64 = U.cgrFasta(fn, k=k)   # NB: cgrFasta skips check_sequence (survey quirk 5)
                out[f"cgrfreq_k{k}"] = c64
        np.savez_compressed(os.path.join(HERE, f"counts_{name}.npz"), **out)
        names, lengths, _, _ = U.SummaryFasta(fn)
        assert list(names) == [r[0] for r in recs] and list(lengths) == out["lengths"].tolist()

    # full Influenza-A: hashes only (the data file itself is in tests/data)
    fn = os.path.join(DATA, "Influenza-A.fas")
    recs = read_records_like_reference(fn)
    for k in (4, 5, 6):
        km = np.zeros((len(recs), 4 ** k), np.int32)
        cg = np.zeros((len(recs), 4 ** k), np.int32)
        for i, (_, s) in enumerate(recs):
            kmer_counts(bytearray(s), k, km[i])
            cgr(bytearray(s), k, cg[i])
        _, f64 = U.kmersFasta(fn, k=k)
        hashes[f"influenza_full_k{k}"] = {"n": len(recs), "kmer_sum": int(km.sum()), "kmer_max": int(km.max()),
                                          "kmer_sha16": sha16(km), "cgr_sha16": sha16(cg), "freq_sha16": sha16(f64),
                                          "freq_f32_sha16": sha16(f64.astype(np.float32)),
                                          "freq_row0_head": f64[0, :4].tolist()}
    names, lengths, gt, dis = U.SummaryFasta(fn, os.path.join(DATA, "Influenza-A_GT.tsv"))
    hashes["influenza_full_summary"] = {"n": len(names), "first": names[0], "last": names[-1],
                                        "len_sum": int(sum(lengths)), "gt_head": list(gt[:3]), "cluster_dis": dis}
    json.dump(hashes, open(os.path.join(HERE, "hashes.json"), "w"), indent=0)

    # ------------------------------------------------------- G3: mutation compat
    recs = read_records_like_reference(os.path.join(DATA, "influenza_64.fas"))[:6]
    recs.append(("withN", bytearray(b"ACGTNNACGTACGTTGCANNNACGATCGATCGATTTAGCNACGT" * 20)))
    mut = {"seqs": [bytes(r[1]).decode() for r in recs], "cases": []}
    for seed in (0, 7):
        for tname, mk in [("transition", lambda: U.transition(1e-2)),
                          ("transversion", lambda: U.transversion(0.5e-2)),
                          ("transition_transversion", lambda: U.transition_transversion(1e-2, 0.5e-2)),
                          ("Random_N", lambda: U.Random_N(20)),
                          ("transition_hi", lambda: U.transition(0.3)),
                          ("transversion_hi", lambda: U.transversion(0.3)),
                          ("tt_hi", lambda: U.transition_transversion(0.3, 0.3))]:
            np.random.seed(seed); random.seed(seed)
            tf = mk()
            outs = []
            for _, s in recs:          # one RNG stream across records, like a kmersFasta pass
                b = bytearray(s)
                tf(b)
                outs.append(bytes(b).decode())
            mut["cases"].append({"seed": seed, "transform": tname, "out": outs})
    json.dump(mut, open(os.path.join(HERE, "mutations.json"), "w"))

    # ------------------------------------------------------- G4: AugmentFasta
    small = os.path.join(HERE, "_tmp_small.fas")
    def write_subset(n):
        recs = read_records_like_reference(os.path.join(DATA, "influenza_64.fas"))[:n]
        with open(small, "wb") as f:
            for i, s in recs:
                f.write(b">" + i.encode() + b"\n" + bytes(s) + b"\n")
    aug = {}
    for (n, n_mimics, k, reduce) in [(16, 3, 4, False), (16, 1, 4, False), (16, 5, 4, True), (6, 3, 6, False), (6, 3, 6, True)]:
        write_subset(n)
        np.random.seed(0); random.seed(0)      # what a fresh `import idelucs.models` leaves behind (models.py:20-21)
        x = U.AugmentFasta(small, n_mimics, k=k, reduce=reduce)
        key = f"n{n}_m{n_mimics}_k{k}_r{int(reduce)}"
        aug[key] = x
    # the EDGE file through AugmentFasta (N runs, short/empty records) with k=4
    # n_mimics=2 has no Random_N pass; with n_mimics>=3 the reference raises on the EMPTY record
    # (np.random.randint(0, 0, 20) -> ValueError("high <= 0"), utils.py:93) -- pinned as error behaviour.
    np.random.seed(0); random.seed(0)
    aug["edge_m2_k4_r0"] = U.AugmentFasta(os.path.join(DATA, "edge.fas"), 2, k=4, reduce=False)
    try:
        np.random.seed(0); random.seed(0)
        U.AugmentFasta(os.path.join(DATA, "edge.fas"), 3, k=4, reduce=False)
        edge_err = None
    except ValueError as e:
        edge_err = str(e)
    json.dump({"edge_m3_error": edge_err}, open(os.path.join(HERE, "augment_errors.json"), "w"))
    np.savez_compressed(os.path.join(HERE, "augment.npz"), **aug)
    # predict-side features: SequenceDataset (unmutated, own float64 scaler)
    write_subset(16)
    ds = U.SequenceDataset(small, k=4)
    np.savez_compressed(os.path.join(HERE, "seqdataset.npz"), n16_k4=ds.kmers,
                        n16_k4_f32=ds.kmers.astype(np.float32))
    os.remove(small)

    # ------------------------------------------------------- G5: nets / losses
    g5 = {}
    torch.manual_seed(1234)
    for tag, net_cls, fin, C in [("linear", NetLinear, 16, 5), ("small", myNet, 10, 7)]:   # k=2 widths keep the fixture small
        net = net_cls(fin, C)
        net.apply(M.weights_init)
        for n_, p in net.state_dict().items():
            g5[f"{tag}.w.{n_}"] = p.numpy().copy()
        x1 = torch.randn(9, fin); x2 = x1 + 0.1 * torch.randn(9, fin)
        g5[f"{tag}.x1"] = x1.numpy().copy(); g5[f"{tag}.x2"] = x2.numpy().copy()
        net.eval()
        out, lat = net(x1.view(-1, 1, fin))
        g5[f"{tag}.eval_out"] = out.detach().numpy().copy(); g5[f"{tag}.eval_latent"] = lat.detach().numpy().copy()
        # one full training step with dropout disabled (eval mode keeps autograd): loss, grads, RMSprop update
        opt = torch.optim.RMSprop(net.parameters(), lr=1e-3, weight_decay=0.01)
        for it in range(2):
            opt.zero_grad()
            z1, h1 = net(x1.view(-1, 1, fin)); z2, h2 = net(x2.view(-1, 1, fin))
            loss = (1 - 0.25) * info_nce_loss(h1, h2, 0.85) + 0.25 * IID_loss(z1, z2, lamb=2.8)
            loss.backward()
            g5[f"{tag}.step{it}.loss"] = np.float32(loss.item())
            if it == 0:
                for n_, p in net.named_parameters():
                    g5[f"{tag}.step{it}.g.{n_}"] = p.grad.numpy().copy()
            opt.step()
            if it == 0:
                for n_, p in net.named_parameters():
                    g5[f"{tag}.step{it}.p.{n_}"] = p.detach().numpy().copy()
    for (B, C) in [(7, 5), (128, 20), (64, 200)]:
        h1 = torch.randn(B, 64, requires_grad=True); h2 = torch.randn(B, 64, requires_grad=True)
        l = info_nce_loss(h1, h2, 0.85); l.backward()
        g5[f"nce.B{B}.h1"] = h1.detach().numpy().copy(); g5[f"nce.B{B}.h2"] = h2.detach().numpy().copy()
        g5[f"nce.B{B}.loss"] = np.float32(l.item())
        g5[f"nce.B{B}.g1"] = h1.grad.numpy().copy(); g5[f"nce.B{B}.g2"] = h2.grad.numpy().copy()
        a = torch.randn(B, C, requires_grad=True); b = torch.randn(B, C, requires_grad=True)
        z1 = torch.softmax(a, 1); z2 = torch.softmax(b, 1)
        l = IID_loss(z1, z2, lamb=2.8); l.backward()
        g5[f"iic.B{B}.C{C}.a"] = a.detach().numpy().copy(); g5[f"iic.B{B}.C{C}.b"] = b.detach().numpy().copy()
        g5[f"iic.B{B}.C{C}.loss"] = np.float32(l.item())
        g5[f"iic.B{B}.C{C}.ga"] = a.grad.numpy().copy(); g5[f"iic.B{B}.C{C}.gb"] = b.grad.numpy().copy()
    # analytic IIC KATs
    for C in (5, 20):
        u = torch.full((10 * C, C), 1.0 / C)
        g5[f"iic.uniform.C{C}"] = np.float32(IID_loss(u, u, lamb=2.8).item())
        oh = torch.eye(C).repeat(10, 1)
        g5[f"iic.onehot.C{C}"] = np.float32(IID_loss(oh, oh, lamb=2.8).item())
    np.savez_compressed(os.path.join(HERE, "nets.npz"), **g5)

    # ------------------------------------------------------- G6: end-to-end anchor
    # Done in a FRESH process so that import-time seeding (models.py:17-21) is what the run sees.
    code = f"""
import sys, json; sys.path.insert(0, {SCRATCH!r})
import numpy as np, pandas as pd
from idelucs.cluster import iDeLUCS_cluster
from idelucs.utils import cluster_acc
m = iDeLUCS_cluster({os.path.join(DATA, 'Influenza-A.fas')!r}, n_clusters=5, n_epochs=10, n_mimics=3, batch_sz=512, k=6, weight=0.25, n_voters=1)
y, lat = m.fit_predict(None)
df = pd.read_csv({os.path.join(DATA, 'Influenza-A_GT.tsv')!r}, sep='\\t')
u = {{v: i for i, v in enumerate(sorted(set(df.cluster_id)))}}
gt = np.array([u[v] for v in df.cluster_id])
ind, acc = cluster_acc(gt, y)
print('RESULT', json.dumps({{'acc': float(acc), 'latent_shape': list(lat.shape), 'n_labels': int(len(set(y.tolist())))}}))
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp")
    line = [l for l in r.stdout.splitlines() if "RESULT" in l]
    anchor = json.loads(line[0].split("RESULT", 1)[1]) if line else {"error": r.stderr[-2000:]}
    json.dump(anchor, open(os.path.join(HERE, "anchor.json"), "w"))
    print("anchor:", anchor)
    print("done; fixtures in", HERE)


if __name__ == "__main__":
    main()
