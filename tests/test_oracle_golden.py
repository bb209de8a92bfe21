"""Pins oracle/ against the golden vectors generated from the REAL reference
(tests/golden/make_golden.py).  CPU only; bit-exact for integer/byte work."""
import hashlib
import json
import os
import random

import numpy as np
import pytest

from oracle import oracle as O
from conftest import DATA, GOLDEN


def sha16(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


KAT = json.load(open(os.path.join(GOLDEN, "kat.json")))


def test_kat_counts_and_cgr():
    for case in KAT["cases"]:
        s, k = case["seq"].encode("latin1"), case["k"]
        c = np.zeros(4 ** k, np.int32); O.kmer_counts(bytearray(s), k, c)
        g = np.zeros(4 ** k, np.int32); O.cgr(bytearray(s), k, g)
        assert np.flatnonzero(c).tolist() == case["kmer"] and c[c != 0].tolist() == case["kmer_v"], (s, k)
        assert np.flatnonzero(g).tolist() == case["cgr"] and g[g != 0].tolist() == case["cgr_v"], (s, k)


def test_kat_survey_values():
    c = np.zeros(16, np.int32); O.kmer_counts(b"ACGTNACGTTGCA", 2, c)
    assert c.tolist() == [0, 2, 0, 0, 1, 0, 2, 0, 0, 1, 0, 2, 0, 0, 1, 1]
    g = np.zeros(16, np.int32); O.cgr(b"ACGTNACGTTGCA", 2, g)
    assert g.tolist() == [0, 1, 2, 0, 2, 0, 0, 1, 1, 0, 0, 2, 0, 0, 0, 1]
    c = np.full(16, 7, np.int32); O.kmer_counts(b"ACGTNACGTTGCA", 2, c)
    assert c.tolist() == KAT["accumulate_k2_from7"]


def test_revcomp_and_collapse():
    for k in (1, 2, 3):
        assert [O.reverse_complement(x, k) for x in range(4 ** k)] == KAT["revcomp"][str(k)]
    assert sha16(np.array([O.reverse_complement(x, 6) for x in range(4096)], np.int32)) == KAT["revcomp_k6_sha16"]
    c = np.zeros(16, np.int32); c[0] = 3; c[15] = 2
    assert O.kmer_rev_comp(c, 2).tolist() == KAT["kmer_rev_comp_k2_trunc"]
    for k, n in KAT["canonical_len"].items():
        assert len(O.kmer_rev_comp(np.ones(4 ** int(k), np.int32), int(k))) == n == O.n_canonical(int(k))


def test_check_sequence():
    for c in KAT["check_sequence"]:
        assert bytes(O.check_sequence(c["header"], bytearray(c["seq"].encode("latin1")))).decode() == c["out"]
    for c in KAT["check_sequence_errors"]:
        if c["error"] is None:
            O.check_sequence(c["header"], bytearray(c["seq"].encode()))
        else:
            with pytest.raises(ValueError) as e:
                O.check_sequence(c["header"], bytearray(c["seq"].encode()))
            assert str(e.value) == c["error"]


@pytest.mark.parametrize("name", ["edge", "edge_nonl", "empty", "influenza_64", "actino_8"])
def test_file_fixtures(name):
    g = np.load(os.path.join(GOLDEN, f"counts_{name}.npz"))
    fn = os.path.join(DATA, name + ".fas")
    recs = list(O.fasta_records(fn))
    assert [r[0] for r in recs] == g["names"].tolist()
    assert [len(r[1]) for r in recs] == g["lengths"].tolist()
    names, lengths, gt, dis = O.SummaryFasta(fn)
    assert names == g["names"].tolist() and lengths == g["lengths"].tolist() and gt is None and dis is None
    for k in (4, 5, 6):
        km = np.zeros((len(recs), 4 ** k), np.int32)
        cg = np.zeros((len(recs), 4 ** k), np.int32)
        canon = []
        for i, (_, s) in enumerate(recs):
            O.kmer_counts(s, k, km[i]); O.cgr(s, k, cg[i])
            canon.append(O.kmer_rev_comp(km[i] + 1, k))
        assert np.array_equal(km, g[f"kmer_k{k}"])
        assert np.array_equal(cg, g[f"cgr_k{k}"])
        assert np.array_equal(np.array(canon, np.int32), g[f"canon_k{k}"])
        n2, f = O.kmersFasta(fn, k)
        assert n2 == names and f.dtype == np.float64 and np.array_equal(f, g[f"freq_k{k}"])
        _, fr = O.kmersFasta(fn, k, reduce=True)
        assert np.array_equal(fr, g[f"freq_canon_k{k}"])
        if name != "empty":
            _, cf = O.cgrFasta(fn, k)
            assert np.array_equal(cf, g[f"cgrfreq_k{k}"])


def test_cgr_is_permutation_of_kmer():
    rng = np.random.default_rng(5)
    for k in range(1, 8):
        for _ in range(10):
            s = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=300, p=[.24, .24, .24, .24, .04])
            c = np.zeros(4 ** k, np.int32); O.kmer_counts(s, k, c)
            g = np.zeros(4 ** k, np.int32); O.cgr(s, k, g)
            assert sorted(c.tolist()) == sorted(g.tolist()) and c.sum() == g.sum()


def test_full_influenza_hashes():
    H = json.load(open(os.path.join(GOLDEN, "hashes.json")))
    fn = os.path.join(DATA, "Influenza-A.fas")
    recs = list(O.fasta_records(fn))
    for k in (4, 5, 6):
        h = H[f"influenza_full_k{k}"]
        km = np.zeros((len(recs), 4 ** k), np.int32); cg = np.zeros_like(km)
        for i, (_, s) in enumerate(recs):
            O.kmer_counts(s, k, km[i]); O.cgr(s, k, cg[i])
        assert (len(recs), int(km.sum()), int(km.max())) == (h["n"], h["kmer_sum"], h["kmer_max"])
        assert sha16(km) == h["kmer_sha16"] and sha16(cg) == h["cgr_sha16"]
        _, f = O.kmersFasta(fn, k)
        assert sha16(f) == h["freq_sha16"] and sha16(f.astype(np.float32)) == h["freq_f32_sha16"]
    names, lengths, gt, dis = O.SummaryFasta(fn, os.path.join(DATA, "Influenza-A_GT.tsv"))
    s = H["influenza_full_summary"]
    assert (len(names), names[0], names[-1], sum(lengths)) == (s["n"], s["first"], s["last"], s["len_sum"])
    assert gt[:3] == s["gt_head"] and dis == s["cluster_dis"]


def test_mutation_compat():
    M = json.load(open(os.path.join(GOLDEN, "mutations.json")))
    mk = {"transition": lambda: O.transition(1e-2), "transversion": lambda: O.transversion(0.5e-2),
          "transition_transversion": lambda: O.transition_transversion(1e-2, 0.5e-2),
          "Random_N": lambda: O.Random_N(20), "transition_hi": lambda: O.transition(0.3),
          "transversion_hi": lambda: O.transversion(0.3), "tt_hi": lambda: O.transition_transversion(0.3, 0.3)}
    for case in M["cases"]:
        np.random.seed(case["seed"]); random.seed(case["seed"])
        tf = mk[case["transform"]]()
        for s, want in zip(M["seqs"], case["out"]):
            b = bytearray(s.encode()); tf(b)
            assert bytes(b).decode() == want, case["transform"]


def _write_subset(tmp_path, n):
    recs = list(O.fasta_records(os.path.join(DATA, "influenza_64.fas")))[:n]
    p = tmp_path / "small.fas"
    with open(p, "wb") as f:
        for i, s in recs:
            f.write(b">" + i.encode() + b"\n" + bytes(s) + b"\n")
    return str(p)


@pytest.mark.parametrize("n,m,k,r", [(16, 3, 4, False), (16, 1, 4, False), (16, 5, 4, True), (6, 3, 6, False), (6, 3, 6, True)])
def test_augment_fasta(tmp_path, n, m, k, r):
    g = np.load(os.path.join(GOLDEN, "augment.npz"))[f"n{n}_m{m}_k{k}_r{int(r)}"]
    np.random.seed(0); random.seed(0)
    x = O.AugmentFasta(_write_subset(tmp_path, n), m, k=k, reduce=r)
    assert x.dtype == np.float32 and x.shape == g.shape
    # the pair vectors are exact; the scaler's float64 column sums may differ in summation order
    np.testing.assert_allclose(x, g, rtol=0, atol=2e-6)


def test_augment_edge_and_error():
    g = np.load(os.path.join(GOLDEN, "augment.npz"))["edge_m2_k4_r0"]
    np.random.seed(0); random.seed(0)
    x = O.AugmentFasta(os.path.join(DATA, "edge.fas"), 2, k=4)
    np.testing.assert_allclose(x, g, rtol=0, atol=2e-6)
    err = json.load(open(os.path.join(GOLDEN, "augment_errors.json")))["edge_m3_error"]
    np.random.seed(0); random.seed(0)
    with pytest.raises(ValueError) as e:
        O.AugmentFasta(os.path.join(DATA, "edge.fas"), 3, k=4)
    assert str(e.value) == err


def test_sequence_dataset_features(tmp_path):
    g = np.load(os.path.join(GOLDEN, "seqdataset.npz"))
    f = O.sequence_dataset_features(_write_subset(tmp_path, 16), k=4)
    np.testing.assert_allclose(f, g["n16_k4"], rtol=0, atol=1e-12)


def test_edit_semantics_reproduce_transforms():
    """pos|op<<30 edits (op0 = N, op1..3 = XOR on A0 C1 G2 T3) can express every reference transform."""
    enc = {65: 0, 67: 1, 71: 2, 84: 3}
    M = json.load(open(os.path.join(GOLDEN, "mutations.json")))
    for case in M["cases"]:
        for s, want in zip(M["seqs"], case["out"]):
            a, b = s.encode(), want.encode()
            edits = []
            for i, (x, y) in enumerate(zip(a, b)):
                if x != y:
                    edits.append(i | ((0 if y == 78 else enc[x] ^ enc[y]) << 30))
            assert bytes(O.apply_edits(a, np.array(edits, np.uint32))) == b
