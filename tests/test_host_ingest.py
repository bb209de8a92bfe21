"""CPU tests of the host side of libidelucs_hip.so: the C ABI loads and exports every declared
symbol, the FASTA reader / check_sequence / packer agree with the golden vectors and the oracle.
No compute entry point is called here (there is no GPU in this tier)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

from conftest import DATA, GOLDEN, ROOT
from oracle import oracle as O

import idelucs_amd
from idelucs_amd import _lib, utils as U

KAT = json.load(open(os.path.join(GOLDEN, "kat.json")))


def test_every_declared_symbol_is_exported_and_bound():
    hdr = open(os.path.join(ROOT, "include", "idelucs_hip.h")).read()
    declared = set(re.findall(r"\b(idl_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"idl_fasta"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(_lib.lib, name), f"{name} declared in include/idelucs_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in idelucs_amd/_lib.py"
    assert set(_lib.SIGNATURES) <= declared
    assert _lib.lib.idl_abi_version() == 1


def test_row_len():
    for k, n in KAT["canonical_len"].items():
        assert _lib.lib.idl_row_len(_lib.MODE_CANONICAL, int(k)) == n
    assert _lib.lib.idl_row_len(_lib.MODE_KMER, 6) == 4096 and _lib.lib.idl_row_len(_lib.MODE_CGR, 4) == 256


def test_reference_import_surface():
    for name in ["check_sequence", "SummaryFasta", "reverse_complement", "kmer_rev_comp", "kmersFasta", "cgrFasta",
                 "cluster_acc", "SequenceDataset", "kmer_counts", "cgr", "IID_model", "IID_loss", "info_nce_loss",
                 "iDeLUCS_cluster"]:
        assert hasattr(idelucs_amd, name), name
    assert idelucs_amd.__version__ == (1, 2, 6)


def test_check_sequence_matches_reference():
    for c in KAT["check_sequence"]:
        assert bytes(U.check_sequence(c["header"], bytearray(c["seq"].encode("latin1")))).decode() == c["out"]
    for c in KAT["check_sequence_errors"]:
        if c["error"] is None:
            U.check_sequence(c["header"], bytearray(c["seq"].encode()))
        else:
            with pytest.raises(ValueError) as e:
                U.check_sequence(c["header"], bytearray(c["seq"].encode()))
            assert str(e.value) == c["error"]


def test_reverse_complement_matches_reference():
    for k in (1, 2, 3):
        assert [U.reverse_complement(x, k) for x in range(4 ** k)] == KAT["revcomp"][str(k)]
    assert all(U.reverse_complement(x, 6) == O.reverse_complement(x, 6) for x in range(4096))


@pytest.mark.parametrize("name", ["edge", "edge_nonl", "empty", "influenza_64", "actino_8"])
def test_fasta_reader_matches_golden(name):
    g = np.load(os.path.join(GOLDEN, f"counts_{name}.npz"))
    fn = os.path.join(DATA, name + ".fas")
    ff = U.FastaFile(fn, check=True, keep_bytes=True)
    assert ff.names == g["names"].tolist()
    assert ff.lengths.tolist() == g["lengths"].tolist()
    recs = list(O.fasta_records(fn))
    for i, (_, s) in enumerate(recs):
        assert bytes(ff.record(i)) == bytes(s)
        # packed layout == the oracle's statement of it
        codes, mask = O.pack(s)
        a, b = ff.slot_off[i], ff.slot_off[i + 1]
        assert b - a == (len(s) + 63) // 64
        assert np.array_equal(ff.codes[a * 16:b * 16], codes) and np.array_equal(ff.mask[a * 8:b * 8], mask)
    names, lengths, gt, dis = U.SummaryFasta(fn)
    assert names == g["names"].tolist() and lengths == g["lengths"].tolist() and gt is None and dis is None


def test_fasta_reader_no_check_keeps_raw_bytes():
    ff = U.FastaFile(os.path.join(DATA, "edge.fas"), check=False, keep_bytes=True)
    recs = list(O.fasta_records(os.path.join(DATA, "edge.fas"), check=False))
    assert [bytes(ff.record(i)) for i in range(ff.n)] == [bytes(s) for _, s in recs]


def test_fasta_errors_match_reference(tmp_path):
    with pytest.raises(ValueError) as e:
        U.FastaFile(os.path.join(DATA, "bad_char.fas"))
    with pytest.raises(ValueError) as e2:
        list(O.fasta_records(os.path.join(DATA, "bad_char.fas")))
    assert str(e.value) == str(e2.value) == "Invalid DNA byte in sequence bad one: 'X'"
    for content, msg in [(b">a\tb\nACGT\n", "tab included in header"), (b"> lead\nACGT\n", "Bad character in sequence header"),
                         (b">>x\nACGT\n", "Bad character in sequence header")]:
        p = tmp_path / "h.fas"
        p.write_bytes(content)
        with pytest.raises(ValueError) as e:
            U.FastaFile(str(p))
        assert str(e.value) == msg
        with pytest.raises(ValueError) as e2:
            list(O.fasta_records(str(p)))
        assert str(e2.value) == msg
    with pytest.raises(FileNotFoundError):
        U.FastaFile(str(tmp_path / "missing.fas"))


def test_summary_fasta_with_gt():
    H = json.load(open(os.path.join(GOLDEN, "hashes.json")))["influenza_full_summary"]
    names, lengths, gt, dis = U.SummaryFasta(os.path.join(DATA, "Influenza-A.fas"), os.path.join(DATA, "Influenza-A_GT.tsv"))
    assert (len(names), names[0], names[-1], sum(lengths)) == (H["n"], H["first"], H["last"], H["len_sum"])
    assert gt[:3] == H["gt_head"] and dis == H["cluster_dis"]


def test_compat_transforms_match_reference_streams():
    """Host-RNG transforms (vectorised numpy) consume numpy/random exactly like the reference."""
    import random
    M = json.load(open(os.path.join(GOLDEN, "mutations.json")))
    mk = {"transition": lambda: U.transition(1e-2), "transversion": lambda: U.transversion(0.5e-2),
          "transition_transversion": lambda: U.transition_transversion(1e-2, 0.5e-2),
          "Random_N": lambda: U.Random_N(20), "transition_hi": lambda: U.transition(0.3),
          "transversion_hi": lambda: U.transversion(0.3), "tt_hi": lambda: U.transition_transversion(0.3, 0.3)}
    for case in M["cases"]:
        np.random.seed(case["seed"]); random.seed(case["seed"])
        tf = mk[case["transform"]]()
        for s, want in zip(M["seqs"], case["out"]):
            b = bytearray(s.encode()); tf(b)
            assert bytes(b).decode() == want, case["transform"]
            # ... and the substitution-edit encoding of the change reproduces it (C oracle applies the edits)
            e = U._edits_from_diff(np.frombuffer(s.encode(), np.uint8), np.frombuffer(bytes(b), np.uint8))
            assert bytes(O.apply_edits(s.encode(), e)).decode() == want


def test_compute_entry_points_fail_loudly_without_gpu():
    if _lib.lib.idl_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError):
        idelucs_amd.kmer_counts(bytearray(b"ACGT"), 2, np.zeros(16, np.int32))
    with pytest.raises(RuntimeError):
        U.kmersFasta(os.path.join(DATA, "edge.fas"), k=4)
    with pytest.raises(RuntimeError):
        idelucs_amd.iDeLUCS_cluster(os.path.join(DATA, "edge.fas")).fit_predict(None)
    # argument validation happens before the device is touched, like the Cython buffer checks
    with pytest.raises(BufferError):
        idelucs_amd.kmer_counts(b"ACGT", 2, np.zeros(16, np.int32))
    with pytest.raises(ValueError) as e:
        idelucs_amd.kmer_counts(bytearray(b"ACGT"), 2, np.zeros(16, np.int64))
    assert "expected 'int' but got 'long'" in str(e.value)
