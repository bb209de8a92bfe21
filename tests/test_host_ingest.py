"""CPU tests of the host side of libidelucs_hip.so: the C ABI loads and exports every declared
symbol, the FASTA reader / check_sequence / packer agree with the golden vectors and the oracle.
No compute entry point is called here (there is no GPU in this tier)."""
import ctypes
from conftest import dev_env
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


def _random_fasta(path, rng, n, with_noise=True):
    alpha = np.frombuffer(b"ACGTacgtNnRYKM-Uu", np.uint8)
    with open(path, "wb") as f:
        for i in range(n):
            L = int(rng.integers(0, 400))
            pr = np.array([.2, .2, .2, .2, .03, .03, .03, .03, .01, .01, .01, .01, .01, .01, .01, .01, .01]); pr /= pr.sum()
            s = rng.choice(alpha, size=L, p=pr).tobytes()
            if with_noise and i % 7 == 3:
                f.write(b"# a comment between records\n")
            f.write(b">rec%d some description %d\n" % (i, i * i))
            w_ = int(rng.integers(20, 90))
            for a in range(0, L, w_):
                eol = b"\r\n" if (with_noise and i % 5 == 2) else b"\n"
                pad = b"  " if (with_noise and i % 11 == 4) else b""
                f.write(pad + s[a:a + w_] + pad + eol)


@pytest.mark.parametrize("threads", ["1", "3", "8"])
def test_threaded_reader_equals_oracle(tmp_path, monkeypatch, threads):
    """The parallel reader (forced on for small files) reproduces the reference's sequential state machine:
    wrapped lines, CRLF, padded lines, comments, lower case / IUPAC / gaps, empty records."""
    monkeypatch.setenv("IDELUCS_THREADS", threads)
    dev_env(monkeypatch, par_min="0")
    rng = np.random.default_rng(int(threads))
    p = str(tmp_path / "r.fas")
    _random_fasta(p, rng, 300)
    for check in (True, False):
        ff = U.FastaFile(p, check=check, keep_bytes=True)
        recs = list(O.fasta_records(p, check=check))
        assert ff.names == [r[0] for r in recs] and ff.lengths.tolist() == [len(r[1]) for r in recs]
        assert [bytes(ff.record(i)) for i in range(ff.n)] == [bytes(r[1]) for r in recs]
        for i, (_, s) in enumerate(recs):
            codes, mask = O.pack(s)
            a, b = ff.slot_off[i], ff.slot_off[i + 1]
            assert np.array_equal(ff.codes[a * 16:b * 16], codes) and np.array_equal(ff.mask[a * 8:b * 8], mask), i


@pytest.mark.parametrize("width", [0, 70, 33, 16, 64, 128, 200])
def test_reader_fast_paths_equal_oracle(tmp_path, width):
    """Long runs of upper-case A/C/G/T take the 64-base (AVX-512, at slot boundaries) / 32-base (AVX2) / 8-base (SWAR) paths of the reader; mixed with N runs, lower case,
    IUPAC codes and gaps at every alignment they must give exactly the bytes, lengths, codes and masks of the byte-wise walk."""
    rng = np.random.default_rng(100 + width)
    p = str(tmp_path / "fast.fas")
    with open(p, "wb") as f:
        for r in range(40):
            L = int(rng.integers(0, 4000))
            s = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=L)
            for _ in range(int(rng.integers(0, 6))):             # sprinkle short exceptional stretches at random offsets
                if L == 0: break
                a = int(rng.integers(0, L)); n_ = int(rng.integers(1, 40))
                s[a:a + n_] = rng.choice(np.frombuffer(b"NnacgtRYK-u", np.uint8), size=len(s[a:a + n_]))
            f.write(b">r%d\n" % r)
            raw = s.tobytes()
            if width == 0: f.write(raw + b"\n")
            else:
                for a in range(0, len(raw), width): f.write(raw[a:a + width] + b"\n")
    ff = U.FastaFile(p, check=True, keep_bytes=True)
    recs = list(O.fasta_records(p, check=True))
    assert ff.lengths.tolist() == [len(r[1]) for r in recs]
    assert [bytes(ff.record(i)) for i in range(ff.n)] == [bytes(r[1]) for r in recs]
    for i, (_, s) in enumerate(recs):
        codes, mask = O.pack(s)
        a, b = ff.slot_off[i], ff.slot_off[i + 1]
        assert np.array_equal(ff.codes[a * 16:b * 16], codes) and np.array_equal(ff.mask[a * 8:b * 8], mask), i


@pytest.mark.parametrize("content", [
    b"ACGT\nAC\n>first\nGGGG\n>second\nTT\n",              # sequence lines before the first header join the first record
    b">a\nAC\n>\nGG\n>b\nTT\n>c\nAA\n",                    # an empty-id header: its lines roll into the next flushed record
    b">\nAC\n",                                                # only an empty id: one record "" at EOF
    b">x\n>y\n>z\nACGT",                                       # empty records, no trailing newline
    b"\n\n>q\n\nAC\n\n\nGT\n\n",                              # blank lines
    b">only header no newline",                                 # id loses its last byte (line[1:-1])
])
def test_reader_quirks_match_reference_state_machine(tmp_path, monkeypatch, content):
    p = tmp_path / "q.fas"
    p.write_bytes(content)
    want = list(O.fasta_records(str(p)))
    for threads, par_min in (("1", None), ("4", "0")):
        monkeypatch.setenv("IDELUCS_THREADS", threads)
        if par_min is not None:
            dev_env(monkeypatch, par_min=par_min)
        ff = U.FastaFile(str(p), keep_bytes=True)
        assert ff.names == [r[0] for r in want], (threads, ff.names)
        assert [bytes(ff.record(i)) for i in range(ff.n)] == [bytes(r[1]) for r in want]


def test_first_error_in_file_order_wins(tmp_path, monkeypatch):
    monkeypatch.setenv("IDELUCS_THREADS", "6")
    dev_env(monkeypatch, par_min="0")
    body = b"".join(b">r%d\nACGTACGTAC\n" % i for i in range(200))
    bad = body.replace(b">r150\nACGT", b">r150\nAC!T").replace(b">r40\nACGT", b">r40\nAZGT").replace(b">r90\n", b">r\t90\n")
    p = tmp_path / "e.fas"
    p.write_bytes(bad)
    with pytest.raises(ValueError) as e:
        U.FastaFile(str(p))
    assert str(e.value) == "Invalid DNA byte in sequence r40: 'Z'"
    with pytest.raises(ValueError) as e2:
        list(O.fasta_records(str(p)))
    assert str(e2.value) == str(e.value)


@pytest.mark.parametrize("threads", ["1", "5"])
def test_ranged_packing_equals_whole_file_export(tmp_path, monkeypatch, threads):
    """idl_fasta_pack_range (the streamed ingest: record chunks packed into the whole-file buffers at their slot offsets)
    gives the same bytes as idl_fasta_export, whatever the chunking, also on Influenza-A."""
    monkeypatch.setenv("IDELUCS_THREADS", threads)
    dev_env(monkeypatch, par_min="0")
    rng = np.random.default_rng(5)
    fn = str(tmp_path / "r.fas")
    _random_fasta(fn, rng, 41)
    for path in (fn, os.path.join(DATA, "Influenza-A.fas")):
        whole = U.FastaFile(path)
        ff = U.FastaFile(path, pack="deferred")
        assert np.array_equal(ff.slot_off, whole.slot_off) and ff.names == whole.names
        codes = np.full(whole.codes.size, 0xAB, np.uint8); mask = np.full(whole.mask.size, 0xCD, np.uint8)
        cuts = sorted(set([0, ff.n] + rng.integers(0, ff.n + 1, 6).tolist()))
        for lo, hi in list(zip(cuts[:-1], cuts[1:]))[::-1]:         # any order
            ff.pack_range(lo, hi, codes, mask)
        ff.pack_range(3, 3, codes, mask)                             # empty range: no-op
        assert np.array_equal(codes, whole.codes) and np.array_equal(mask, whole.mask)
        with pytest.raises(ValueError):
            ff.pack_range(0, ff.n + 1, codes, mask)
        ff.close()
        with pytest.raises(ValueError):
            ff.pack_range(0, 1, codes, mask)
    assert 1 <= U.ingest_threads() <= 256


def _one_pass(path, cap_slots=None):
    """idl_fasta_parse_pack on host arenas (no device) -> (FastaFile, codes, mask) or None on IDL_FALLBACK."""
    size = os.path.getsize(path)
    cap = cap_slots if cap_slots is not None else size // 48 + 4096 * U.ingest_threads() + 1024
    codes = np.full(cap * 16, 0xAB, np.uint8)
    mask = np.full(cap * 8, 0xCD, np.uint8)
    h = ctypes.c_void_p()
    rc = _lib.lib.idl_fasta_parse_pack(os.fsencode(path), U._ptr(codes), U._ptr(mask), cap, None, None, None, ctypes.byref(h))
    if rc == _lib.IDL_FALLBACK:
        return None
    _lib.check(rc)
    return U.FastaFile.from_handle(h, arena=True), codes, mask


@pytest.mark.parametrize("threads", ["1", "3", "8"])
def test_one_pass_reader_equals_general_reader(tmp_path, monkeypatch, threads):
    """idl_fasta_parse_pack (validate + count + pack in one pass, every thread into its own region of the arenas) gives the names,
    lengths and packed bytes of the general reader, record by record, at the slots it reports -- wrapped lines, CRLF, padded lines,
    comments, lower case / IUPAC / gaps, empty records, Influenza-A; and records of any one thread sit back to back."""
    monkeypatch.setenv("IDELUCS_THREADS", threads)
    dev_env(monkeypatch, par_min="0")
    rng = np.random.default_rng(40 + int(threads))
    fn = str(tmp_path / "r.fas")
    _random_fasta(fn, rng, 300)
    for path in (fn, os.path.join(DATA, "Influenza-A.fas")):
        whole = U.FastaFile(path)
        got = _one_pass(path)
        assert got is not None, "the one-pass reader declined a file it should take"
        ff, codes, mask = got
        assert ff.names == whole.names and np.array_equal(ff.lengths, whole.lengths) and ff.slot_off.shape == (ff.n + 1,)
        for i in range(ff.n):
            a, b = whole.slot_off[i], whole.slot_off[i + 1]
            s = int(ff.slot_off[i])
            assert np.array_equal(codes[s * 16:(s + b - a) * 16], whole.codes[a * 16:b * 16]), i
            assert np.array_equal(mask[s * 8:(s + b - a) * 8], whole.mask[a * 8:b * 8]), i
        starts = ff.slot_off[:-1]
        assert np.all(np.diff(starts) >= (whole.slot_off[1:-1] - whole.slot_off[:-2]))        # never overlapping, in file order


@pytest.mark.parametrize("content", [
    b"ACGT\nAC\n>first\nGGGG\n>second\nTT\n",              # sequence lines before the first header
    b">a\nAC\n>\nGG\n>b\nTT\n>c\nAA\n",                    # an empty-id header
    b">\nAC\n",                                                # only an empty id
    b"\n\n>q\n\nAC\n\n\nGT\n\n",                              # blank lines before the first header
    b"ACGT\n",                                                 # no header at all
    b"",                                                       # empty file
])
def test_one_pass_reader_leaves_rolling_layouts_to_the_general_reader(tmp_path, monkeypatch, content):
    p = tmp_path / "q.fas"
    p.write_bytes(content)
    for threads in ("1", "4"):
        monkeypatch.setenv("IDELUCS_THREADS", threads)
        dev_env(monkeypatch, par_min="0")
        assert _one_pass(str(p)) is None


@pytest.mark.parametrize("content", [
    b"# a comment first\n#another\n>x\n>y\n>z\nACGT",         # comments before the first header, empty records, no trailing newline
    b">only header no newline",                                 # id loses its last byte (line[1:-1])
    b">q\n\nAC\n\n\nGT\n\n>r\n  ac gt\t\n",                   # blank lines and interior white space inside records
])
def test_one_pass_reader_on_small_layouts(tmp_path, monkeypatch, content):
    p = tmp_path / "q.fas"
    p.write_bytes(content)
    want = list(O.fasta_records(str(p)))
    for threads in ("1", "4"):
        monkeypatch.setenv("IDELUCS_THREADS", threads)
        dev_env(monkeypatch, par_min="0")
        got = _one_pass(str(p))
        assert got is not None
        ff, codes, mask = got
        assert ff.names == [r[0] for r in want] and ff.lengths.tolist() == [len(r[1]) for r in want]
        for i, (_, s) in enumerate(want):
            c, m = O.pack(s)
            a = int(ff.slot_off[i])
            assert np.array_equal(codes[a * 16:a * 16 + c.size], c) and np.array_equal(mask[a * 8:a * 8 + m.size], m)


def test_one_pass_reader_errors_and_full_regions(tmp_path, monkeypatch):
    """Same first-error-in-file-order rule and messages as the general reader; a region that runs out of slots is a fallback."""
    monkeypatch.setenv("IDELUCS_THREADS", "6")
    dev_env(monkeypatch, par_min="0")
    body = b"".join(b">r%d\nACGTACGTAC\n" % i for i in range(200))
    bad = body.replace(b">r150\nACGT", b">r150\nAC!T").replace(b">r40\nACGT", b">r40\nAZGT").replace(b">r90\n", b">r\t90\n")
    p = tmp_path / "e.fas"
    p.write_bytes(bad)
    with pytest.raises(ValueError) as e:
        _one_pass(str(p))
    assert str(e.value) == "Invalid DNA byte in sequence r40: 'Z'"
    p.write_bytes(body.replace(b">r7\n", b">#r7\n"))
    with pytest.raises(ValueError) as e:
        _one_pass(str(p))
    assert str(e.value) == "Bad character in sequence header"
    p.write_bytes(body)
    assert _one_pass(str(p), cap_slots=60) is None               # 200 records of one slot each do not fit 60 slots
    assert _one_pass(str(p)) is not None


def test_kept_mapping_never_serves_a_rewritten_file(tmp_path, monkeypatch):
    """The readers keep the mapping of the last file after its handles close.  A file rewritten in place (same path, same size,
    same inode) must be read anew: the key holds the modification time; idl_ingest_release drops the mapping."""
    monkeypatch.setenv("IDELUCS_THREADS", "2")
    dev_env(monkeypatch, par_min="0")
    p = tmp_path / "m.fas"
    p.write_bytes(b">a\nACGTACGT\n>b\nGGGGCCCC\n")
    first = U.FastaFile(str(p), keep_bytes=True)
    assert bytes(first.record(0)) == b"ACGTACGT"
    again = U.FastaFile(str(p), keep_bytes=True)                     # served from the kept mapping
    assert bytes(again.record(1)) == b"GGGGCCCC"
    st = os.stat(p)
    with open(p, "r+b") as f:                                        # in place: same inode, same size
        f.write(b">a\nTTTTTTTT\n>b\nAAAAAAAA\n")
    os.utime(p, ns=(st.st_atime_ns, st.st_mtime_ns + 1_000_000))     # (a coarse clock could repeat the time stamp)
    changed = U.FastaFile(str(p), keep_bytes=True)
    assert bytes(changed.record(0)) == b"TTTTTTTT" and bytes(changed.record(1)) == b"AAAAAAAA"
    got = _one_pass(str(p))
    assert got is not None and got[0].lengths.tolist() == [8, 8]
    U.release_ingest_buffers()
    assert bytes(U.FastaFile(str(p), keep_bytes=True).record(0)) == b"TTTTTTTT"
    other = tmp_path / "n.fas"
    other.write_bytes(b">z\nACACACAC\n")
    assert bytes(U.FastaFile(str(other), keep_bytes=True).record(0)) == b"ACACACAC"


def test_lazy_prim_refuses_inputs_beyond_its_look_ahead():
    """the sleeping-groups Prim scans a thread's 4 look-ahead points only, i.e. 1024 x 256 x 4 = 2^20 positions;
    beyond that points would never become candidates and the run would end in 'no progress' after minutes.  The entry point
    now rejects such a call in its argument checks (nothing is launched: this runs without a GPU), and the host gate sends
    those inputs to idl_mst_prim_local, whose step kernel strides on behind the look-ahead."""
    from idelucs_amd import posthoc
    assert posthoc.MST_LAZY_MAX == 1 << 20 and posthoc.MST_LAZY_MIN <= posthoc.MST_LAZY_MAX
    buf = (ctypes.c_uint8 * 512)()
    p = (ctypes.addressof(buf) + 255) & ~255
    for n, ok in (((1 << 20) + 4096, False), (1 << 23, False)):
        rc = _lib.lib.idl_mst_prim_lazy(p, p, p, n, 64, p, 0, p, p, p, p, p, 4, p, p, p, p, p, p, p, None, None)
        assert rc == _lib.IDL_ERR_ARG and "2^20" in _lib.last_error(), (n, rc, _lib.last_error())
    src = open(os.path.join(ROOT, "idelucs_amd", "posthoc.py")).read()
    assert "MST_LAZY_MIN <= n <= MST_LAZY_MAX" in src


def test_reader_thread_default_follows_quota_and_ranks():
    """Round 4: the reader's default thread count is min(32, hardware threads, 2 x the cgroup's CPU quota) shared out over the ranks
    of the node (LOCAL_WORLD_SIZE: every rank of a multi-GPU job parses the file itself); IDELUCS_THREADS overrides.  Checked in
    child processes (the default is computed once per process)."""
    import subprocess
    import sys

    def threads(**env):
        e = {k: v for k, v in os.environ.items() if k not in ("IDELUCS_THREADS", "LOCAL_WORLD_SIZE")}
        e.update(env, PYTHONPATH=ROOT)
        r = subprocess.run([sys.executable, "-c", "from idelucs_amd import utils as U; print(U.ingest_threads())"], env=e, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        return int(r.stdout.strip().splitlines()[-1])
    base = threads()
    hw = os.cpu_count() or 1
    assert 1 <= base <= min(32, hw)
    t2, t8 = threads(LOCAL_WORLD_SIZE="2"), threads(LOCAL_WORLD_SIZE="8")
    assert 1 <= t8 <= t2 <= base and t8 <= max(1, base // 2)          # (the share of a rank shrinks with the ranks of the node)
    assert threads(IDELUCS_THREADS="5", LOCAL_WORLD_SIZE="8") == 5


def test_reader_cpu_plan_one_core_per_thread(monkeypatch):
    """Round 5: a device job's reader threads are bound to the NUMA node of the device; with IDELUCS_DEV=numa_pin=1 one core each, cores
    dealt over the L3 domains (idl_ingest_cpu_plan reports the placement without changing anything).  Here, without a device: IDELUCS_DEV=numa=<node>
    names the node; every thread's set is a non-empty part of this process's CPUs, and as many threads as the node has cores get
    cores of their own; IDELUCS_DEV=numa=off plans nothing."""
    if not os.path.isdir("/sys/devices/system/node/node0"):
        pytest.skip("no NUMA topology in sysfs")
    mine = os.sched_getaffinity(0)
    nt = 12
    first = np.full(nt, -2, np.int32); count = np.zeros(nt, np.int32)
    dev_env(monkeypatch, numa="0")
    dev_env(monkeypatch, numa_pin="1")
    node = _lib.lib.idl_ingest_cpu_plan(-1, nt, U._ptr(first), U._ptr(count))
    if node < 0:
        pytest.skip("node 0 holds none of this process's CPUs")
    assert node == 0 and np.all(count >= 1) and all(int(c) in mine for c in first)
    cores = set()
    for c in first:                                             # threads beyond the node's cores wrap around
        sib = open(f"/sys/devices/system/cpu/cpu{int(c)}/topology/thread_siblings_list").read().strip()
        cores.add(sib)
    n_cores = len({open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip() for c in mine
                   if os.path.exists(f"/sys/devices/system/node/node0/cpu{c}")})
    assert len(cores) == min(nt, n_cores)
    dev_env(monkeypatch, numa_pin="0")                  # (the default) the node's whole set for everybody, or nothing to do at all
    assert _lib.lib.idl_ingest_cpu_plan(-1, nt, U._ptr(first), U._ptr(count)) == 0 and len(set(count.tolist())) == 1
    dev_env(monkeypatch, numa="off")
    assert _lib.lib.idl_ingest_cpu_plan(-1, nt, U._ptr(first), U._ptr(count)) == -1 and np.all(count == 0)


def test_reader_pool_is_reused_and_survives_fork(tmp_path, monkeypatch):
    """Round 5: the reader's threads persist between calls.  Jobs of different widths follow each other, and a forked child
    (whose parent's pool threads do not exist there) reads with a pool of its own."""
    dev_env(monkeypatch, par_min="0")
    rng = np.random.default_rng(77)
    fn = str(tmp_path / "p.fas")
    _random_fasta(fn, rng, 200)
    want = None
    for threads in ("6", "2", "8", "3"):
        monkeypatch.setenv("IDELUCS_THREADS", threads)
        ff = U.FastaFile(fn)
        got = (ff.names, ff.lengths.tolist(), ff.codes.tobytes(), ff.mask.tobytes())
        want = want or got
        assert got == want
    pid = os.fork()
    if pid == 0:                                                 # the child: must not hang on the parent's sleeping workers
        code = 1
        try:
            ff = U.FastaFile(fn)
            one = _one_pass(fn)
            ok = (ff.names, ff.lengths.tolist(), ff.codes.tobytes(), ff.mask.tobytes()) == want and one is not None
            code = 0 if ok else 2
        finally:
            os._exit(code)
    import time
    t0 = time.time()
    while True:
        done, status = os.waitpid(pid, os.WNOHANG)
        if done:
            break
        if time.time() - t0 > 60:
            os.kill(pid, 9)
            os.waitpid(pid, 0)
            pytest.fail("the forked child hung in the reader")
        time.sleep(0.05)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, status


def test_arena_meta_and_deferred_names(tmp_path, monkeypatch):
    """Round 5: idl_fasta_arena_meta hands out lengths, arena slots and the length range right after the one pass; the names come
    later (FastaFile.from_handle(meta=...): on first use or at close()) and are the general reader's."""
    monkeypatch.setenv("IDELUCS_THREADS", "4")
    dev_env(monkeypatch, par_min="0")
    for path in (os.path.join(DATA, "Influenza-A.fas"),):
        whole = U.FastaFile(path)
        size = os.path.getsize(path)
        cap = size // 48 + 4096 * 4 + 1024
        codes = np.zeros(cap * 16, np.uint8); mask = np.zeros(cap * 8, np.uint8)
        for use in ("names", "close"):
            h = ctypes.c_void_p()
            _lib.check(_lib.lib.idl_fasta_parse_pack(os.fsencode(path), U._ptr(codes), U._ptr(mask), cap, None, None, None, ctypes.byref(h)))
            n = ctypes.c_int64()
            _lib.check(_lib.lib.idl_fasta_sizes(h, ctypes.byref(n), None, None, None))
            meta = np.empty(2 * n.value + 1, np.int64)
            lo, hi = ctypes.c_int64(), ctypes.c_int64()
            _lib.check(_lib.lib.idl_fasta_arena_meta(h, U._ptr(meta), ctypes.c_void_p(meta.ctypes.data + 8 * n.value), ctypes.byref(lo), ctypes.byref(hi)))
            assert n.value == whole.n and np.array_equal(meta[:n.value], whole.lengths)
            assert (lo.value, hi.value) == (int(whole.lengths.min()), int(whole.lengths.max()))
            slots = np.empty(n.value + 1, np.int64)
            _lib.check(_lib.lib.idl_fasta_arena_slots(h, U._ptr(slots)))
            assert np.array_equal(meta[n.value:], slots)
            ff = U.FastaFile.from_handle(h, arena=True, meta=(meta[:n.value], meta[n.value:]))
            assert ff._names_raw is None and ff.total_bases == whole.total_bases
            if use == "close":
                ff.close()                                        # reads the names out before the handle goes
                assert ff._h is None
            assert ff.names == whole.names
            ff.close()
    # a handle of the general reader has no arena meta
    h = ctypes.c_void_p()
    _lib.check(_lib.lib.idl_fasta_open(os.fsencode(path), 1, ctypes.byref(h)))
    assert _lib.lib.idl_fasta_arena_meta(h, None, None, None, None) != _lib.IDL_OK
    _lib.lib.idl_fasta_close(h)


def test_file_page_node_probe(tmp_path):
    """Round 5: the reader binds itself beside the FILE's page-cache pages when it can tell where they are (a sample of resident
    pages asked with move_pages).  A file just written is resident: the probe names a node that exists (or -1 where the runtime
    refuses the query); a file too small to sample, a missing file: -1."""
    big = tmp_path / "big.bin"
    big.write_bytes(os.urandom(4 << 20))
    node = _lib.lib.idl_ingest_probe_file_node(os.fsencode(str(big)))
    nodes = []
    if os.path.isdir("/sys/devices/system/node"):
        nodes = [int(d[4:]) for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()]
    assert node == -1 or node in nodes
    small = tmp_path / "small.bin"
    small.write_bytes(b"x" * 1000)
    assert _lib.lib.idl_ingest_probe_file_node(os.fsencode(str(small))) == -1
    assert _lib.lib.idl_ingest_probe_file_node(os.fsencode(str(tmp_path / "nope"))) == -1
    assert _lib.lib.idl_ingest_file_node() in [-1] + nodes


def test_non_ascii_headers_are_checked_when_the_file_is_read(tmp_path, monkeypatch):
    """in the one-pass ingest the names stay in the handle until somebody asks for them -- but the two checks of the reference that need DECODED
    names (invalid UTF-8; a unicode space as a header's first character, which `line[1:].strip()` semantics of idelucs/utils.py:26-51 reject) must fail
    when the file is read, not when the CLI writes its TSV after training.  idl_fasta_names_high says whether any header byte is >= 0x80; FastaFile.from_handle(meta=...)
    then exports and validates at once.  ASCII files keep the deferral (names exported on first use)."""
    monkeypatch.setenv("IDELUCS_THREADS", "4")
    dev_env(monkeypatch, par_min="0")

    def open_deferred(content):
        p = tmp_path / "u.fas"
        p.write_bytes(content)
        size = os.path.getsize(p)
        cap = size // 48 + 4096 * U.ingest_threads() + 1024
        codes = np.zeros(cap * 16, np.uint8); mask = np.zeros(cap * 8, np.uint8)
        h = ctypes.c_void_p()
        _lib.check(_lib.lib.idl_fasta_parse_pack(os.fsencode(str(p)), U._ptr(codes), U._ptr(mask), cap, None, None, None, ctypes.byref(h)))
        n = ctypes.c_int64()
        _lib.check(_lib.lib.idl_fasta_sizes(h, ctypes.byref(n), None, None, None))
        lengths = np.empty(n.value, np.int64); slot_off = np.empty(n.value + 1, np.int64)
        lo, hi = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(_lib.lib.idl_fasta_arena_meta(h, U._ptr(lengths), U._ptr(slot_off), ctypes.byref(lo), ctypes.byref(hi)))
        high = int(_lib.lib.idl_fasta_names_high(h))
        return high, (lambda: U.FastaFile.from_handle(h, arena=True, meta=(lengths, slot_off)))

    body = b"".join(b">r%d\nACGTACGTAC\n" % i for i in range(50))
    high, make = open_deferred(body)
    ff = make()
    assert high == 0 and ff._names_raw is None                  # ASCII: deferred
    assert ff.names[:2] == ["r0", "r1"]
    high, make = open_deferred(body.replace(b">r7\n", ">r7é\n".encode()))
    ff = make()
    assert high == 1 and ff._names_raw is not None and ff.names[7] == "r7é"      # valid non-ASCII: read out at once, accepted
    high, make = open_deferred(body.replace(b">r9\n", "> r9\n".encode()))       # an EM SPACE heads the name
    assert high == 1
    with pytest.raises(ValueError) as e:
        make()
    assert str(e.value) == "Bad character in sequence header"
    high, make = open_deferred(body.replace(b">r11\n", b">r11\xff\xfe\n"))
    assert high == 1
    with pytest.raises((UnicodeDecodeError, ValueError)):
        make()
