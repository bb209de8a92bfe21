"""GPU parity tests (run with -m gpu on an MI355X): the HIP vectoriser, called through the C ABI,
against (a) the golden vectors generated from the real reference and (b) the CPU oracle on seeded
random inputs.  Integer/byte results must be bit-exact."""
import json
from conftest import dev_env
import os
import random

import numpy as np
import pytest

from conftest import DATA, GOLDEN
from oracle import oracle as O

pytestmark = pytest.mark.gpu

KAT = json.load(open(os.path.join(GOLDEN, "kat.json")))


@pytest.fixture(scope="module")
def gpu():
    import torch
    import idelucs_amd
    from idelucs_amd import _lib
    _lib.require_gpu()
    assert torch.cuda.is_available()
    return idelucs_amd


def test_scalar_api_kats(gpu):
    for case in KAT["cases"]:
        s, k = case["seq"].encode("latin1"), case["k"]
        c = np.zeros(4 ** k, np.int32); gpu.kmer_counts(bytearray(s), k, c)
        g = np.zeros(4 ** k, np.int32); gpu.cgr(bytearray(s), k, g)
        assert np.flatnonzero(c).tolist() == case["kmer"] and c[c != 0].tolist() == case["kmer_v"], (s, k)
        assert np.flatnonzero(g).tolist() == case["cgr"] and g[g != 0].tolist() == case["cgr_v"], (s, k)
    c = np.full(16, 7, np.int32); gpu.kmer_counts(bytearray(b"ACGTNACGTTGCA"), 2, c)
    assert c.tolist() == KAT["accumulate_k2_from7"]
    # numpy uint8 arrays are accepted like bytearrays; seq is not modified
    s = np.frombuffer(b"ACGTNACGTTGCA", np.uint8).copy(); keep = s.copy()
    c = np.zeros(16, np.int32); gpu.kmer_counts(s, 2, c)
    assert c.tolist() == [0, 2, 0, 0, 1, 0, 2, 0, 0, 1, 0, 2, 0, 0, 1, 1] and np.array_equal(s, keep)


def test_scalar_api_random_vs_oracle(gpu):
    rng = np.random.default_rng(11)
    alphabet = np.frombuffer(b"ACGTNacgtX-", np.uint8)
    for L in [1, 2, 5, 6, 7, 63, 64, 65, 127, 128, 129, 1000, 4095, 4096, 4097, 4102, 8191, 8192, 8200, 20000]:
        for k in (1, 2, 3, 4, 5, 6, 7):
            s = rng.choice(alphabet, size=L, p=[.23, .23, .23, .23, .03, .01, .01, .01, .01, .005, .005])
            init = rng.integers(0, 5, 4 ** k).astype(np.int32)
            want = init.copy(); O.kmer_counts(s, k, want)
            got = init.copy(); gpu.kmer_counts(s.copy(), k, got)
            assert np.array_equal(got, want), (L, k)
            want = init.copy(); O.cgr(s, k, want)
            got = init.copy(); gpu.cgr(s.copy(), k, got)
            assert np.array_equal(got, want), (L, k)


def test_kmer_rev_comp_api(gpu):
    c = np.zeros(16, np.int32); c[0] = 3; c[15] = 2
    assert gpu.kmer_rev_comp(c, 2).tolist() == KAT["kmer_rev_comp_k2_trunc"]
    rng = np.random.default_rng(3)
    for k in (1, 2, 3, 4, 5, 6, 7):
        c = rng.integers(0, 1000, 4 ** k).astype(np.int32)
        c2 = c.copy(); want = O.kmer_rev_comp(c2, k)
        c3 = c.copy(); got = gpu.kmer_rev_comp(c3, k)
        assert np.array_equal(got, want) and np.array_equal(c2, c3), k   # in-place side effect identical too


@pytest.mark.parametrize("name", ["edge", "edge_nonl", "empty", "influenza_64", "actino_8"])
def test_batched_vectoriser_vs_golden(gpu, name):
    import torch
    from idelucs_amd import _lib, utils as U
    g = np.load(os.path.join(GOLDEN, f"counts_{name}.npz"))
    fn = os.path.join(DATA, name + ".fas")
    ff = U.FastaFile(fn)
    din = U._DeviceInput(ff, torch.device("cuda"))
    for k in (4, 5, 6):
        c = U._vectorise(din, k, _lib.MODE_KMER, _lib.INIT_ZERO, _lib.OUT_COUNTS_I32)[0].cpu().numpy()
        assert np.array_equal(c, g[f"kmer_k{k}"]), ("kmer", k)
        c = U._vectorise(din, k, _lib.MODE_CGR, _lib.INIT_ZERO, _lib.OUT_COUNTS_I32)[0].cpu().numpy()
        assert np.array_equal(c, g[f"cgr_k{k}"]), ("cgr", k)
        c = U._vectorise(din, k, _lib.MODE_CANONICAL, _lib.INIT_ONE, _lib.OUT_COUNTS_I32)[0].cpu().numpy()
        assert np.array_equal(c, g[f"canon_k{k}"]), ("canon", k)
        # reference-level functions: float64 rows, bit-exact (exact ints divided in float64)
        names, f = gpu.kmersFasta(fn, k=k)
        assert names == g["names"].tolist() and f.dtype == np.float64 and np.array_equal(f, g[f"freq_k{k}"])
        _, fr = gpu.kmersFasta(fn, k=k, reduce=True)
        assert np.array_equal(fr, g[f"freq_canon_k{k}"])
        if name != "empty":
            _, cf = gpu.cgrFasta(fn, k=k)
            assert np.array_equal(cf, g[f"cgrfreq_k{k}"])
        # float32 output == float32(float64 division)  (what astype('float32') does at utils.py:353)
        f32 = U._vectorise(din, k, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_FREQ_F32)[0].cpu().numpy()
        assert np.array_equal(f32, g[f"freq_k{k}"].astype(np.float32))


def test_full_influenza_hashes(gpu):
    import hashlib
    import torch
    from idelucs_amd import _lib, utils as U
    H = json.load(open(os.path.join(GOLDEN, "hashes.json")))
    sha16 = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
    ff = U.FastaFile(os.path.join(DATA, "Influenza-A.fas"))
    din = U._DeviceInput(ff, torch.device("cuda"))
    for k in (4, 5, 6):
        h = H[f"influenza_full_k{k}"]
        km = U._vectorise(din, k, _lib.MODE_KMER, _lib.INIT_ZERO, _lib.OUT_COUNTS_I32)[0].cpu().numpy()
        cg = U._vectorise(din, k, _lib.MODE_CGR, _lib.INIT_ZERO, _lib.OUT_COUNTS_I32)[0].cpu().numpy()
        assert (int(km.sum()), int(km.max()), sha16(km), sha16(cg)) == (h["kmer_sum"], h["kmer_max"], h["kmer_sha16"], h["cgr_sha16"])
        f64 = U._vectorise(din, k, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_FREQ_F64)[0].cpu().numpy()
        f32 = U._vectorise(din, k, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_FREQ_F32)[0].cpu().numpy()
        assert sha16(f64) == h["freq_sha16"] and sha16(f32) == h["freq_f32_sha16"]


def _random_batch(rng, n, lmin, lmax, p_n=0.002):
    seqs = []
    for _ in range(n):
        L = int(rng.integers(lmin, lmax + 1))
        s = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=L, p=[(1 - p_n) / 4] * 4 + [p_n])
        seqs.append(s)
    return seqs


def _pack_batch(seqs):
    from idelucs_amd import _lib
    import ctypes
    n = len(seqs)
    byte_off = np.zeros(n + 1, np.int64); np.cumsum([len(s) for s in seqs], out=byte_off[1:])
    data = np.concatenate(seqs) if n else np.empty(0, np.uint8)
    slots = int(sum((len(s) + 63) // 64 for s in seqs))
    codes = np.zeros(max(slots, 1) * 16, np.uint8); mask = np.zeros(max(slots, 1) * 8, np.uint8)
    slot_off = np.zeros(n + 1, np.int64)
    p = lambda a: ctypes.c_void_p(a.ctypes.data)
    _lib.check(_lib.lib.idl_pack(p(data), p(byte_off), n, p(codes), p(mask), p(slot_off)))
    class FF: pass
    ff = FF(); ff.n = n; ff.codes = codes; ff.mask = mask; ff.slot_off = slot_off
    ff.lengths = np.array([len(s) for s in seqs], np.int64)
    return ff


@pytest.mark.parametrize("sc_slots,bins", [(None, None), ("1", None), ("2", "16"), ("5", None), (None, "16")])
def test_random_batches_with_edits_vs_oracle(gpu, monkeypatch, sc_slots, bins):
    """Multi-view launch with explicit substitution edits (all four ops, chunk boundaries, duplicates).
    sc_slots shrinks the LDS super-chunk of the delta-view kernel so that halo / boundary handling is
    exercised on small inputs."""
    import torch
    if sc_slots is not None:
        dev_env(monkeypatch, sc_slots=sc_slots)
    if bins is not None:
        dev_env(monkeypatch, bins=bins)          # opt-in 16-bit LDS bins (default: 32-bit)
    from idelucs_amd import _lib, utils as U
    rng = np.random.default_rng(2024)
    seqs = _random_batch(rng, 40, 0, 300) + _random_batch(rng, 12, 3900, 4300) + _random_batch(rng, 6, 8000, 13000, 0.0) \
        + [np.frombuffer(b"A" * 5000, np.uint8), np.frombuffer(b"ACGT" * 1024, np.uint8)]
    n, P = len(seqs), 4
    edits, counts, mutated = [], [], [[None] * n for _ in range(P)]
    for v in range(P):
        for i, s in enumerate(seqs):
            L = len(s)
            m = 0 if (v == 0 or L == 0) else int(rng.integers(0, max(2, L // 20)))
            pos = np.sort(rng.integers(0, max(L, 1), m)).astype(np.uint32)     # duplicates allowed
            op = rng.integers(0, 4, m).astype(np.uint32)
            e = pos | (op << np.uint32(30))
            edits.append(e); counts.append(m)
            mutated[v][i] = O.apply_edits(s.tobytes(), e)
    edit_off = np.zeros(P * n + 1, np.int64); np.cumsum(counts, out=edit_off[1:])
    e_all = np.concatenate(edits)
    dev = torch.device("cuda")
    din = U._DeviceInput(_pack_batch(seqs), dev)
    d_e = torch.from_numpy(e_all.view(np.int32)).to(dev); d_eo = torch.from_numpy(edit_off).to(dev)
    for k in (3, 4, 6, 7):
        for mode, fn in ((_lib.MODE_KMER, O.kmer_counts), (_lib.MODE_CGR, O.cgr)):
            got = U._vectorise(din, k, mode, _lib.INIT_ZERO, _lib.OUT_COUNTS_I32, P, d_e, d_eo).cpu().numpy()
            for v in range(P):
                for i in range(n):
                    want = np.zeros(4 ** k, np.int32); fn(mutated[v][i], k, want)
                    assert np.array_equal(got[v, i], want), (k, mode, v, i, len(seqs[i]))
        got = U._vectorise(din, k, _lib.MODE_CANONICAL, _lib.INIT_ONE, _lib.OUT_FREQ_F64, P, d_e, d_eo).cpu().numpy()
        for v in (0, 3):
            for i in range(0, n, 7):
                c = np.ones(4 ** k, np.int32); O.kmer_counts(mutated[v][i], k, c)
                c = O.kmer_rev_comp(c, k)
                assert np.array_equal(got[v, i], c / np.sum(c)), (k, v, i)


@pytest.mark.parametrize("sc_slots,long_seq", [(None, False), ("1", False), ("3", False), (None, True)])
def test_adversarial_edits_at_boundaries(gpu, monkeypatch, sc_slots, long_seq):
    """Edits packed around slot / super-chunk boundaries, runs of adjacent edits (overlapping windows),
    duplicates, N edits on top of XOR edits, edits in the last k bases and past-the-end positions."""
    import torch
    from idelucs_amd import _lib, utils as U
    if sc_slots is not None:
        dev_env(monkeypatch, sc_slots=sc_slots)
    rng = np.random.default_rng(77)
    seqs = _random_batch(rng, 6, 380, 400, 0.01) + _random_batch(rng, 2, 64, 64, 0.0) + _random_batch(rng, 2, 128, 129, 0.0) \
        + _random_batch(rng, 2, 17000, 17010, 0.001)
    if long_seq:                     # > 65000 bases: counts may exceed 16 bits -> the launcher must pick 32-bit bins
        seqs += [np.frombuffer(b"AC" * 40000, np.uint8), rng.choice(np.frombuffer(b"ACGT", np.uint8), size=70001)]
    n, P = len(seqs), 3
    edits, counts, mutated = [], [], [[None] * n for _ in range(P)]
    for v in range(P):
        for i, s in enumerate(seqs):
            L = len(s)
            cand = []
            for b in (0, 63, 64, 65, 127, 128, 191, 192, 193, 16383, 16384, 16385, L - 1, L - 2, L - 6, L - 7):
                cand += [b + d for d in range(-2, 3)]
            cand += list(range(100, 100 + 12))                     # a dense run: every window overlaps several edits
            cand += [150, 150, 150, 151]                            # duplicates
            pos = np.array(sorted(c for c in cand if 0 <= c < L), np.uint32)
            if v == 0:
                pos = pos[::3]
            op = rng.integers(0, 4, pos.size).astype(np.uint32)
            e = pos | (op << np.uint32(30))
            edits.append(e); counts.append(e.size)
            mutated[v][i] = O.apply_edits(s.tobytes(), e)
    edit_off = np.zeros(P * n + 1, np.int64); np.cumsum(counts, out=edit_off[1:])
    dev = torch.device("cuda")
    din = U._DeviceInput(_pack_batch(seqs), dev)
    d_e = torch.from_numpy(np.concatenate(edits).view(np.int32)).to(dev); d_eo = torch.from_numpy(edit_off).to(dev)
    for k in (1, 2, 4, 6, 7):
        got = U._vectorise(din, k, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_COUNTS_I32, P, d_e, d_eo).cpu().numpy()
        f32 = U._vectorise(din, k, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_FREQ_F32, P, d_e, d_eo).cpu().numpy()
        for v in range(P):
            for i in range(n):
                want = np.ones(4 ** k, np.int32); O.kmer_counts(mutated[v][i], k, want)
                assert np.array_equal(got[v, i], want), (k, v, i, len(seqs[i]))
                assert np.array_equal(f32[v, i], (want / np.sum(want)).astype(np.float32)), (k, v, i)


def _write_subset(tmp_path, n):
    recs = list(O.fasta_records(os.path.join(DATA, "influenza_64.fas")))[:n]
    p = tmp_path / "small.fas"
    with open(p, "wb") as f:
        for i, s in recs:
            f.write(b">" + i.encode() + b"\n" + bytes(s) + b"\n")
    return str(p)


@pytest.mark.parametrize("n,m,k,r", [(16, 3, 4, False), (16, 1, 4, False), (16, 5, 4, True), (6, 3, 6, False), (6, 3, 6, True)])
def test_augment_fasta_compat_vs_reference(gpu, tmp_path, n, m, k, r):
    """Host-RNG compat mode: the device result equals the reference's AugmentFasta output
    (golden, generated with numpy/random seeded 0 like a fresh import of the reference)."""
    from idelucs_amd import utils as U
    g = np.load(os.path.join(GOLDEN, "augment.npz"))[f"n{n}_m{m}_k{k}_r{int(r)}"]
    np.random.seed(0); random.seed(0)
    x = U.AugmentFasta(_write_subset(tmp_path, n), m, k=k, reduce=r, rng="compat")
    assert x.dtype == np.float32 and x.shape == g.shape
    # tolerance: the scaler's float64 column sums are reduced in a different order on the device
    np.testing.assert_allclose(x, g, rtol=0, atol=2e-6)


def test_augment_edge_file_and_error(gpu):
    from idelucs_amd import utils as U
    g = np.load(os.path.join(GOLDEN, "augment.npz"))["edge_m2_k4_r0"]
    np.random.seed(0); random.seed(0)
    x = U.AugmentFasta(os.path.join(DATA, "edge.fas"), 2, k=4, rng="compat")
    np.testing.assert_allclose(x, g, rtol=0, atol=2e-6)
    err = json.load(open(os.path.join(GOLDEN, "augment_errors.json")))["edge_m3_error"]
    for mode in ("compat", "philox"):
        np.random.seed(0); random.seed(0)
        with pytest.raises(ValueError) as e:
            U.AugmentFasta(os.path.join(DATA, "edge.fas"), 3, k=4, rng=mode)
        assert str(e.value) == err


def test_scaler_and_gather_vs_oracle(gpu):
    import torch
    from idelucs_amd import utils as U
    rng = np.random.default_rng(9)
    dev = torch.device("cuda")
    for (n, f) in [(1, 4), (3, 10), (37, 136), (300, 256), (1000, 4096)]:
        x32 = (rng.random((n, f)) * 1e-3).astype(np.float32)
        x32[:, 0] = 0.25                                   # a zero-variance column -> scale 1
        mean, scale = U.col_stats(torch.from_numpy(x32).to(dev))
        m_ref, s_ref = O.scaler_fit(x32)
        np.testing.assert_allclose(mean.cpu().numpy(), m_ref, rtol=1e-13, atol=0)
        np.testing.assert_allclose(scale.cpu().numpy(), s_ref, rtol=1e-9, atol=0)
        assert scale[0].item() == 1.0
        # transform with the ORACLE's statistics uploaded: then float32 results must be bit-exact
        dm, ds = torch.from_numpy(m_ref).to(dev), torch.from_numpy(s_ref).to(dev)
        y = U.standardise(torch.from_numpy(x32).to(dev), dm, ds).cpu().numpy()
        assert np.array_equal(y, O.scaler_transform(x32, m_ref, s_ref))
        x64 = x32.astype(np.float64) * 1.0000001
        m64, s64 = O.scaler_fit(x64)
        y = U.standardise(torch.from_numpy(x64).to(dev), torch.from_numpy(m64).to(dev), torch.from_numpy(s64).to(dev)).cpu().numpy()
        assert np.array_equal(y, O.scaler_transform(x64, m64, s64).astype(np.float32))
    # gather_pairs == standardise + the reference's pair layout
    P, n, f = 4, 50, 256
    feats = (rng.random((P, n, f)) * 1e-3).astype(np.float32)
    m_ref, s_ref = O.scaler_fit(feats[0])
    st = U.FeatureStore(None, None, torch.from_numpy(feats).to(dev), torch.from_numpy(m_ref).to(dev),
                        torch.from_numpy(s_ref).to(dev), 4, False)
    idx = rng.permutation((P - 1) * n)[:70].astype(np.int64)
    y = st.gather_pairs(torch.from_numpy(idx).to(dev)).cpu().numpy()
    want_true = O.scaler_transform(feats[0][idx % n], m_ref, s_ref)
    want_mod = O.scaler_transform(feats[1 + idx // n, idx % n], m_ref, s_ref)
    assert np.array_equal(y[:70], want_true) and np.array_equal(y[70:], want_mod)
    # the reciprocal form used by the training step (idl_gather_pairs_at with inv_scale) gives the same bits
    from idelucs_amd import _lib
    from idelucs_amd.utils import _ptr, _stream_ptr
    y2 = torch.empty_like(torch.from_numpy(y)).to(dev)
    didx = torch.from_numpy(idx).to(dev)
    _lib.check(_lib.lib.idl_gather_pairs_at(_ptr(st.feats), st.n, st.f, st.n * st.f, _ptr(didx), None, 70, _ptr(st.mean), _ptr(st.scale),
                                            _ptr(st.inv_scale), _ptr(y2), _stream_ptr()))
    assert np.array_equal(y2.cpu().numpy(), y)


def test_sequence_dataset_features(gpu, tmp_path):
    from idelucs_amd import utils as U
    g = np.load(os.path.join(GOLDEN, "seqdataset.npz"))
    fn = _write_subset(tmp_path, 16)
    ds = U.SequenceDataset(fn, k=4)
    assert ds.kmers.dtype == np.float64 and len(ds) == 16
    np.testing.assert_allclose(ds.kmers, g["n16_k4"], rtol=0, atol=1e-10)
    _, _, f32 = U.predict_features(fn, k=4)
    np.testing.assert_allclose(f32.cpu().numpy(), g["n16_k4_f32"], rtol=0, atol=2e-6)


def test_size_independent_properties_at_full_size(gpu):
    """cfg2-sized check (100k x 10 kbp is too slow for the scalar oracle): sum of counts ==
    sum(L) - N*(k-1) for N-free input; CGR is a permutation of k-mer counts; rows of frequencies
    sum to 1; canonical rows sum to the same totals."""
    import torch
    from idelucs_amd import _lib, utils as U
    dev = torch.device("cuda")
    n, L, k = 20000, 10000, 6
    slots = (L + 63) // 64
    g = torch.Generator(device=dev); g.manual_seed(1)
    codes = torch.randint(-2 ** 31, 2 ** 31 - 1, (n * slots * 4,), dtype=torch.int32, device=dev, generator=g)
    mask = torch.zeros(n * slots * 2, dtype=torch.int32, device=dev)
    # mark the padding of the last slot invalid: bases [L % 64, 64) of the last slot
    tail = L % 64
    if tail:
        lo = torch.tensor([0, 0], dtype=torch.int64)
        for j in range(tail, 64):
            lo[j // 32] |= (1 << (31 - (j % 32)))
        m = mask.view(n, slots, 2)
        m[:, -1, 0] = int(lo[0]) - (1 << 32) if int(lo[0]) >= 2 ** 31 else int(lo[0])
        m[:, -1, 1] = int(lo[1]) - (1 << 32) if int(lo[1]) >= 2 ** 31 else int(lo[1])
    class D: pass
    din = D(); din.n = n; din.codes = codes; din.mask = mask
    din.slot_off = torch.arange(0, (n + 1) * slots, slots, dtype=torch.int64, device=dev)
    din.lengths = torch.full((n,), L, dtype=torch.int64, device=dev)
    km = U._vectorise(din, k, _lib.MODE_KMER, _lib.INIT_ZERO, _lib.OUT_COUNTS_I32)[0]
    assert torch.all(km.sum(1) == L - (k - 1))
    cg = U._vectorise(din, k, _lib.MODE_CGR, _lib.INIT_ZERO, _lib.OUT_COUNTS_I32)[0]
    assert torch.equal(torch.sort(km, dim=1).values, torch.sort(cg, dim=1).values)
    fr = U._vectorise(din, k, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_FREQ_F32)[0]
    assert torch.allclose(fr.double().sum(1), torch.ones(n, dtype=torch.float64, device=dev), atol=1e-6)
    assert torch.equal(fr, ((km + 1).double() / float(L - (k - 1) + 4 ** k)).float())
    ca = U._vectorise(din, k, _lib.MODE_CANONICAL, _lib.INIT_ZERO, _lib.OUT_COUNTS_I32)[0]
    assert ca.shape[1] == 2080 and torch.all(ca.sum(1) <= (L - (k - 1)) // 2 + 64) and torch.all(ca.sum(1) >= (L - k + 1 - 2080) // 2)


def test_cfg2_full_size_store_properties(gpu):
    """The exact workload bench.py times (BASELINE configs[1]: 100 000 x 10 kbp, k=6, 4 views, device-drawn mimic sites; byte
    offsets of the 6.55 GB store pass 4 GiB), checked through size-independent properties: per view the counts sum to the
    number of valid windows (XOR-only views keep all L-k+1, the Random_N view loses <= 20*k), every float32 row equals
    (count + 1) / (windows + 4^k) recomputed from the integer counts, rows sum to 1, and the views differ."""
    import torch
    from idelucs_amd import _lib, utils as U
    import importlib.util
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    spec = importlib.util.spec_from_file_location("bench_vectorise", os.path.join(tools, "bench_vectorise.py"))
    bv = importlib.util.module_from_spec(spec); spec.loader.exec_module(bv)
    dev = torch.device("cuda")
    n, L, k, P = 100000, 10000, 6, 4
    F = 4 ** k
    din = bv.synth_input(n, L, dev)
    edits, edit_off = U._philox_edits(din, [t.spec() for t in U.mimic_transforms(P - 1)], 12345)
    feats = U._vectorise(din, k, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_FREQ_F32, P, edits, edit_off)
    counts = U._vectorise(din, k, _lib.MODE_KMER, _lib.INIT_ZERO, _lib.OUT_COUNTS_I32, P, edits, edit_off)
    assert feats.numel() * 4 > 2 ** 32 and tuple(feats.shape) == (P, n, F)
    for v in range(P):
        S = counts[v].sum(1, dtype=torch.int64)
        if v < 3:
            assert torch.all(S == L - (k - 1)), v
        else:
            assert int(S.max()) <= L - (k - 1) and int(S.min()) >= L - (k - 1) - 20 * k, v
        for lo in range(0, n, 25000):          # chunked: the float64 temporaries of 100 000 rows would be 3.3 GB each
            c = counts[v, lo:lo + 25000]
            want = ((c + 1).double() / (S[lo:lo + 25000] + F).double().unsqueeze(1)).float()
            assert torch.equal(feats[v, lo:lo + 25000], want), (v, lo)
            assert torch.allclose(feats[v, lo:lo + 25000].double().sum(1), torch.ones(c.shape[0], dtype=torch.float64, device=dev), atol=1e-6)
    assert not torch.equal(counts[0, :1000], counts[1, :1000]) and not torch.equal(counts[1, :1000], counts[2, :1000])
    # the last rows of the last view (highest addresses) are real rows, not left-overs
    assert int(counts[P - 1, n - 1].sum()) >= L - (k - 1) - 20 * k


@pytest.mark.parametrize("k", [6, 5, 4])
def test_cgr_and_canonical_rows_of_the_wave_per_sequence_kernel_are_v2s(gpu, monkeypatch, k):
    """Round 6: the CGR permutation (kmers.pyx:53-123) and the canonical collapse with its truncating halve and own normalisation
    (utils.py:208-221, 246-250) as epilogues of vectorise4_kernel (k = 4, 5, 6: a wavefront owns the finished histogram; k = 6 -- the reference's default k, model_size='small' -- in the instance that undoes a view by its edits) -- bit for bit the rows of the delta-view
    kernel v2 (the checker the oracle tests hold), counts and float32 frequencies, 20 000 x 10 kbp x 4 views with device-drawn mimic edits; also when some
    sequences go to the second pass."""
    import torch
    from idelucs_amd import _lib, utils as U
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_vectorise", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bench_vectorise.py"))
    bv = importlib.util.module_from_spec(spec); spec.loader.exec_module(bv)
    dev = torch.device("cuda")
    din = bv.synth_input(20000, 10000, dev)
    edits, edit_off = U._philox_edits(din, [t.spec() for t in U.mimic_transforms(3)], 13)
    for mode in (_lib.MODE_CGR, _lib.MODE_CANONICAL):
        outs = {}
        for which in ("2", "4"):
            dev_env(monkeypatch, vec=which)
            outs[which] = (U._vectorise(din, k, mode, _lib.INIT_ONE, _lib.OUT_FREQ_F32, 4, edits, edit_off).clone(),
                           U._vectorise(din, k, mode, _lib.INIT_ZERO, _lib.OUT_COUNTS_I32, 4, edits, edit_off).clone(),
                           U._vectorise(din, k, mode, _lib.INIT_ONE, _lib.OUT_FREQ_F32).clone())
        for a_, b_ in zip(outs["2"], outs["4"]):
            assert a_.shape == b_.shape and torch.equal(a_, b_), (mode, k)
        assert outs["4"][0].shape[-1] == int(_lib.lib.idl_row_len(mode, k))
        assert float(outs["4"][0][1:].sum(-1).sub(1).abs().max()) < 1e-5 and not torch.equal(outs["4"][0][0], outs["4"][0][1])
        dev_env(monkeypatch, vec="4"); dev_env(monkeypatch, v3_ec="320")
        assert torch.equal(U._vectorise(din, k, mode, _lib.INIT_ONE, _lib.OUT_FREQ_F32, 4, edits, edit_off), outs["2"][0]), (mode, k)
        dev_env(monkeypatch, v3_ec=None)


@pytest.mark.parametrize("k", [6, 5, 4])
def test_full_size_v1_v2_v3_kernels_agree_bitwise(gpu, monkeypatch, k):
    """cfg2-shaped batch (20 000 x 10 kbp, 4 views, device-drawn mimic edits): the single-pass kernel (v1: full recount per
    view, edits applied to a staged copy), the delta-view kernel (v2: one count + XOR-mask window moves, per-view pair passes)
    and the pipelined kernel (v3, the default here: LDS-DMA staging one sequence ahead, one pair pass for all views, recorded
    pair lists) are independent code paths and must produce identical bits -- counts and float32 frequencies."""
    import torch
    from idelucs_amd import _lib, utils as U
    sys_path_tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_vectorise", os.path.join(sys_path_tools, "bench_vectorise.py"))
    bv = importlib.util.module_from_spec(spec); spec.loader.exec_module(bv)
    dev = torch.device("cuda")
    din = bv.synth_input(20000, 10000, dev)
    specs = [t.spec() for t in U.mimic_transforms(3)]
    edits, edit_off = U._philox_edits(din, specs, 11)
    outs = {}
    for which in ("4", "3", "2", "1"):                        # (4: round 6's wave-per-sequence kernel, the default at k = 4 / 5; it does not take k = 6: v3 runs)
        dev_env(monkeypatch, vec=which)
        outs[which] = (U._vectorise(din, k, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_FREQ_F32, 4, edits, edit_off).clone(),
                       U._vectorise(din, k, _lib.MODE_KMER, _lib.INIT_ZERO, _lib.OUT_COUNTS_I32, 4, edits, edit_off).clone())
    assert torch.equal(outs["1"][1], outs["2"][1]) and torch.equal(outs["1"][1], outs["3"][1]) and torch.equal(outs["1"][1], outs["4"][1])
    assert torch.equal(outs["1"][0], outs["2"][0]) and torch.equal(outs["1"][0], outs["3"][0]) and torch.equal(outs["1"][0], outs["4"][0])
    if k in (4, 5):     # ... also when its tables take only some of the sequences (the rest: the second pass on v2), and with other numbers of histogram copies
        for ec, lc in (("0", "0"), ("320", "1800")):
            dev_env(monkeypatch, vec="4"); dev_env(monkeypatch, v3_ec=ec); dev_env(monkeypatch, v3_lc=lc)
            assert torch.equal(U._vectorise(din, k, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_FREQ_F32, 4, edits, edit_off), outs["1"][0]), (ec, lc)
        dev_env(monkeypatch, v3_ec=None); dev_env(monkeypatch, v3_lc=None)
        if k == 4:
            for copies in ("16", "8", "1"):
                dev_env(monkeypatch, vec="4"); dev_env(monkeypatch, v4_copies=copies)
                assert torch.equal(U._vectorise(din, k, _lib.MODE_KMER, _lib.INIT_ZERO, _lib.OUT_COUNTS_I32, 4, edits, edit_off), outs["1"][1]), copies
            dev_env(monkeypatch, v4_copies=None)
    if k == 4:      # round 5: the count goes to 16 copies of the 4^4-bin histogram (lane l adds to copy l mod 16); 32 and 8 copies: the same rows
        for copies in ("32", "8"):
            dev_env(monkeypatch, vec="3"); dev_env(monkeypatch, v3_copies=copies)
            assert torch.equal(U._vectorise(din, k, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_FREQ_F32, 4, edits, edit_off), outs["1"][0]), copies
            assert torch.equal(U._vectorise(din, k, _lib.MODE_KMER, _lib.INIT_ZERO, _lib.OUT_COUNTS_I32, 4, edits, edit_off), outs["1"][1]), copies
        dev_env(monkeypatch, v3_copies=None)
    # sequences whose edits / pairs do not fit v3's LDS tables are left to a second pass on the v2 kernel: same bits
    # (tables for no / half / nearly all of the sequences: the second pass scans, or walks the short list v3 left it)
    for ec, lc in (("0", "0"), ("320", "1800"), ("512", "1920"), ("448", "2304")):
        dev_env(monkeypatch, vec="3"); dev_env(monkeypatch, v3_ec=ec); dev_env(monkeypatch, v3_lc=lc)
        assert torch.equal(U._vectorise(din, k, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_FREQ_F32, 4, edits, edit_off), outs["1"][0]), (ec, lc)
    dev_env(monkeypatch, v3_ec=None); dev_env(monkeypatch, v3_lc=None)
    c = outs["2"][1]
    assert not torch.equal(c[0], c[1]) and int(c[3].sum(1).min()) >= 10000 - (k - 1) - 20 * k       # the views differ; Random_N kills <= 20*k windows


def test_kernel_projection_matches_numpy(gpu):
    """BASELINE config 4: kernels/kernel4.npz (the reference's data file) applied to k=4 frequency rows == np.dot."""
    from idelucs_amd import utils as U
    kfile = U.kernel_file(4)
    K = np.load(kfile)["arr_0"]
    assert K.shape == (256, 135)
    g = np.load(os.path.join(GOLDEN, "counts_influenza_64.npz"))
    want = np.dot(g["freq_k4"], K)
    got = U.project_kernel(g["freq_k4"], kfile).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-15)


def test_megabase_sequence_many_super_chunks(gpu):
    """A 3 Mbp sequence (293 super-chunks of the delta-view kernel) with N runs and ~45 000 edits per view, next to short
    sequences in the same launch, against the oracle."""
    import torch
    from idelucs_amd import _lib, utils as U
    rng = np.random.default_rng(31)
    big = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=3_000_007, p=[.2495, .2495, .2495, .2495, .002])
    big[1_500_000:1_500_300] = ord("N")
    seqs = [rng.choice(np.frombuffer(b"ACGT", np.uint8), size=777), big, rng.choice(np.frombuffer(b"ACGT", np.uint8), size=10_241)]
    n, P = len(seqs), 3
    edits, counts, mutated = [], [], [[None] * n for _ in range(P)]
    for v in range(P):
        for i, s in enumerate(seqs):
            L = len(s)
            m = 0 if v == 0 else int(L * 0.015)
            pos = np.sort(rng.integers(0, L, m)).astype(np.uint32)
            op = rng.integers(0, 4, m).astype(np.uint32)
            e = pos | (op << np.uint32(30))
            edits.append(e); counts.append(m)
            mutated[v][i] = O.apply_edits(s.tobytes(), e)
    edit_off = np.zeros(P * n + 1, np.int64); np.cumsum(counts, out=edit_off[1:])
    dev = torch.device("cuda")
    din = U._DeviceInput(_pack_batch(seqs), dev)
    d_e = torch.from_numpy(np.concatenate(edits).view(np.int32)).to(dev); d_eo = torch.from_numpy(edit_off).to(dev)
    for k in (4, 6):
        got = U._vectorise(din, k, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_COUNTS_I32, P, d_e, d_eo).cpu().numpy()
        f64 = U._vectorise(din, k, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_FREQ_F64, P, d_e, d_eo).cpu().numpy()
        for v in range(P):
            for i in range(n):
                want = np.ones(4 ** k, np.int32); O.kmer_counts(mutated[v][i], k, want)
                assert np.array_equal(got[v, i], want), (k, v, i)
                assert np.array_equal(f64[v, i], want / np.sum(want)), (k, v, i)


def test_arbitrary_transform_callables(gpu, tmp_path):
    """kmersFasta(fname, k, transform) takes ANY callable in the reference (utils.py:239-240).  Callables that are not base
    substitutions -- deleting bases, inserting, writing bytes kmer_counts skips -- run on the host on the cleaned record and the
    result is re-packed and counted on the device: equal to the oracle applying the same callable."""
    from idelucs_amd import utils as U
    fn = _write_subset(tmp_path, 12)

    def drop_every_7th(seq):                   # changes the length
        del seq[::7]

    def insert_gaps(seq):                      # longer, with bytes that restart the window (kmers.pyx:19-34: anything but ACGT)
        seq[10:10] = b"NN-xx"
        seq[-3:] = b"acg"

    calls = []

    def third_record_only(seq):                # substitutions for some records, a deletion for one: mutated exactly once each
        calls.append(len(seq))
        if len(calls) == 3:
            del seq[5:9]
        else:
            seq[0:1] = b"T"

    for tf in (drop_every_7th, insert_gaps, third_record_only):
        calls.clear()
        names, got = U.kmersFasta(fn, k=5, transform=tf)
        n_calls = len(calls)
        calls.clear()                                   # the oracle pass below must hit the same third record
        want = []
        for _, s in O.fasta_records(fn):
            b = bytearray(s)
            tf(b)
            c = np.ones(4 ** 5, np.int32)
            O.kmer_counts(b, 5, c)
            want.append(c / np.sum(c))
        assert np.array_equal(got, np.array(want)), tf.__name__
        if tf is third_record_only:
            assert n_calls == 12               # no record went through the callable twice


def test_random_n_beyond_the_device_sorter(gpu, tmp_path):
    """Random_N(n_bp > 64): more draws than the device generator sorts in a wave -- drawn on the host from numpy's global stream
    exactly like the reference (utils.py:78-95), whatever the rng mode; equal to the oracle under the same seed."""
    from idelucs_amd import _lib, utils as U
    assert _lib.lib.idl_mimic_max_random_n() == 64
    fn = _write_subset(tmp_path, 8)
    np.random.seed(5)
    _, got = U.kmersFasta(fn, k=6, transform=U.Random_N(150), rng="philox")
    np.random.seed(5)
    tf = O.Random_N(150)
    want = []
    for _, s in O.fasta_records(fn):
        b = bytearray(s)
        tf(b)
        c = np.ones(4 ** 6, np.int32)
        O.kmer_counts(b, 6, c)
        want.append(c / np.sum(c))
    assert np.array_equal(got, np.array(want))


@pytest.mark.parametrize("k,d", [(4, 135), (5, 511), (6, 2079)])
def test_kernel_projection_all_k(gpu, k, d):
    """BASELINE config 4 (kernels/kernel{4,5,6}.npz parity): kmersFasta(..., project=True) == np.dot(reference frequency rows,
    KERNEL) for the reference's three kernel files (the product the reference keeps commented out, utils.py:272-275)."""
    from idelucs_amd import utils as U
    K = np.load(U.kernel_file(k))["arr_0"]
    assert K.shape == (4 ** k, d)
    g = np.load(os.path.join(GOLDEN, "counts_influenza_64.npz"))
    want = np.dot(g[f"freq_k{k}"], K)
    names, got = U.kmersFasta(os.path.join(DATA, "influenza_64.fas"), k=k, project=True)
    assert got.dtype == np.float64 and got.shape == (64, d) and names == list(g["names"])
    np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-14)
    _, got_r = U.kmersFasta(os.path.join(DATA, "influenza_64.fas"), k=k, reduce=True, project=True)    # the projection is the reduction
    assert np.array_equal(got, got_r)
    with pytest.raises(ValueError):
        U.kernel_file(7)


def test_variant_n_synthetic_input_vs_oracle(gpu, monkeypatch):
    """SURVEY 8(d)'s variant "N" of the synthetic input (bench.py --n-rate: every base additionally an N w.p. 1e-2 here): the
    packed batch decoded back to bytes and counted by the oracle gives exactly the vectoriser's counts -- every N restarts the
    window -- for all views with device-drawn mimic sites, and the three kernels agree bit for bit."""
    import importlib.util
    import torch
    from idelucs_amd import _lib, utils as U
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    dev = torch.device("cuda")
    n, L, k, P = 300, 3000, 6, 4
    din = bench.synth_packed(n, L, dev, seed=7, n_rate=1e-2)
    codes = din.codes.view(torch.int32).cpu().numpy().view(np.uint32).reshape(n, -1)          # [n, slots * 4] words, first base on top
    mask = din.mask.view(torch.int32).cpu().numpy().view(np.uint32).reshape(n, -1)            # [n, slots * 2]
    shifts = np.arange(30, -2, -2, dtype=np.uint32)
    bases = ((codes[:, :, None] >> shifts) & 3).reshape(n, -1)[:, :L]
    inval = ((mask[:, :, None] >> np.arange(31, -1, -1, dtype=np.uint32)) & 1).reshape(n, -1)[:, :L].astype(bool)
    assert 0.005 < inval.mean() < 0.02
    seqs = np.frombuffer(b"ACGT", np.uint8)[bases]
    seqs[inval] = ord("N")
    counts = U._vectorise(din, k, _lib.MODE_KMER, _lib.INIT_ZERO, _lib.OUT_COUNTS_I32)[0].cpu().numpy()
    for i in range(0, n, 7):
        want = np.zeros(4 ** k, np.int32)
        O.kmer_counts(bytearray(seqs[i].tobytes()), k, want)
        assert np.array_equal(counts[i], want), i
        assert want.sum() < L - (k - 1)                                  # windows were lost to the Ns
    edits, edit_off = U._philox_edits(din, [t.spec() for t in U.mimic_transforms(P - 1)], 3)
    outs = {}
    for ver in ("1", "2", "3"):
        dev_env(monkeypatch, vec=ver)
        outs[ver] = U._vectorise(din, k, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_FREQ_F32, P, edits, edit_off)
    assert torch.equal(outs["1"], outs["2"]) and torch.equal(outs["1"], outs["3"])
    assert torch.allclose(outs["3"].double().sum(2), torch.ones((P, n), dtype=torch.float64, device=dev), atol=1e-6)


def test_two_vectorise_calls_in_flight_on_two_streams(gpu):
    """The hand-over words between the pipelined kernel and its second pass (queue head, redo count, redo list) belong to the
    (device, stream) of the launch: two calls in flight on two streams of one device -- a build on one, predict_features on
    another -- leave the rows each of them leaves alone."""
    import torch
    from idelucs_amd import _lib, utils as U
    rng = np.random.default_rng(31)
    dev = torch.device("cuda")
    batches = [_random_batch(rng, 1500, 2000, 6000, 0.001) + _random_batch(rng, 2, 60000, 70000, 0.0) for _ in range(2)]   # (the long ones go to the second pass)
    dins = [U._DeviceInput(_pack_batch(b), dev) for b in batches]
    want = [U._vectorise(d, 6, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_FREQ_F32).clone() for d in dins]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    for rep in range(4):
        got = [None, None]
        for i in (0, 1):
            with torch.cuda.stream(streams[i]):
                got[i] = U._vectorise(dins[i], 6, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_FREQ_F32)
        torch.cuda.synchronize()
        for i in (0, 1):
            assert torch.equal(got[i], want[i]), (rep, i)


@pytest.mark.parametrize("k", [4, 5, 6, 7])
def test_predict_inputs_from_counts_equal_the_float64_route(tmp_path, k):
    """Round 4 (the exchange step): the predict inputs of SequenceDataset (reference utils.py:400-405: float64 rows counts / sum(counts),
    StandardScaler fit_transform in float64, one rounding to float32) formed straight from int32 counts -- idl_counts_stats +
    idl_counts_standardise -- are BIT FOR BIT those of the materialised float64 rows (idl_vectorise OUT_FREQ_F64 + idl_col_stats +
    idl_standardise): statistics, row totals, output, also for a row shard; on Influenza-A and on records with N runs, a record
    shorter than k (only the pseudocount) and constant columns (scale 1)."""
    import torch
    from idelucs_amd import _lib, utils as U
    rng = np.random.default_rng(3 + k)
    recs = [b">short\nAC\n", b">allN\n" + b"N" * 40 + b"\n"]
    for i in range(700):
        L = int(rng.integers(k, 3000))
        s = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=L, p=[0.4, 0.1, 0.2, 0.3])
        if i % 5 == 0:
            s[L // 2: L // 2 + 1 + L // 20] = ord("N")
        recs.append(b">r%d\n" % i + s.tobytes() + b"\n")
    p = tmp_path / "c.fas"
    p.write_bytes(b"".join(recs))
    dev = torch.device("cuda")
    one = tmp_path / "one.fas"
    one.write_bytes(b">only\nACGTACGTTTGACCA\n")                      # a single row: variance 0 everywhere -> scale 1, output 0
    two = tmp_path / "two.fas"
    two.write_bytes(b">a\nACGTACGTTTGACCAGGT\n>b\nTTTTTTTTTTGACCAGGTAC\n")
    for path in (str(p), os.path.join(DATA, "Influenza-A.fas"), str(one), str(two)):
        ff = U.FastaFile(path, check=True)
        din = U._DeviceInput(ff, dev)
        f64 = U._vectorise(din, k, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_FREQ_F64)[0]
        mean, scale = U.col_stats(f64)
        want = U.standardise(f64, mean, scale)
        got = U.predict_inputs_from_counts(din, k)
        assert torch.equal(got, want), (path, k, (got - want).abs().max().item())
        lo, hi = min(17, ff.n - 1), min(ff.n, 403)
        assert torch.equal(U.predict_inputs_from_counts(din, k, (lo, hi)), want[lo:hi])
        assert U.predict_inputs_from_counts(din, k, (hi, hi)).shape == (0, 4 ** k)
        if ff.n == 1:
            assert not bool(want.any())
        # the pieces: row totals and statistics
        counts = U._vectorise(din, k, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_COUNTS_I32)[0]
        n, f = counts.shape
        m2 = torch.empty(f, dtype=torch.float64, device=dev); s2 = torch.empty(f, dtype=torch.float64, device=dev)
        tot = torch.empty(n, dtype=torch.int32, device=dev)
        ws = torch.empty(int(_lib.lib.idl_counts_stats_workspace(n, f)), dtype=torch.uint8, device=dev)
        _lib.check(_lib.lib.idl_counts_stats(U._ptr(counts), n, f, U._ptr(m2), U._ptr(s2), U._ptr(tot), U._ptr(ws), U._stream_ptr()))
        assert torch.equal(tot.long(), counts.long().sum(1)) and torch.equal(m2, mean) and torch.equal(s2, scale)
    # through the product's entry (SequenceDataset's numbers are pinned to the reference golden elsewhere: test_predict... / seqdataset.npz)
    names, lengths, x = U.predict_features(os.path.join(DATA, "Influenza-A.fas"), k=k)
    U.OPTIONS["predict_counts"] = "0"
    try:
        _, _, x_old = U.predict_features(os.path.join(DATA, "Influenza-A.fas"), k=k)
    finally:
        U.OPTIONS["predict_counts"] = "1"
    assert torch.equal(x, x_old)


@pytest.mark.parametrize("streams", ["1", "3"])
def test_one_pass_ingest_device_image_with_sparse_mask(gpu, tmp_path, monkeypatch, streams):
    """Round 5: the one-pass reader copies a piece's invalid-mask only when a record of the piece holds an N; the masks of the
    other records are written on the device from their lengths (idl_mask_from_lengths).  The device image -- packed bases and
    mask of every record, at its arena slot -- is the general reader's, byte for byte, with one copy stream and with several;
    and the feature store built from it is the one built with every mask copied."""
    import torch
    from idelucs_amd import _lib, utils as U
    dev_env(monkeypatch, par_min="0")
    monkeypatch.setenv("IDELUCS_THREADS", "6")
    dev_env(monkeypatch, copy_streams=streams)
    rng = np.random.default_rng(5)
    fn = str(tmp_path / "mix.fas")
    with open(fn, "wb") as f:                      # long clean stretches (pieces without an N), a few records with N runs / IUPAC, short and empty ones
        for i in range(6000):
            L = int(rng.choice([0, 1, 63, 64, 65, 700, 1000, 1013]))
            s = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=L)
            if i % 997 == 5 and L > 10:
                s[3:9] = ord("N"); s[-1] = ord("r")
            f.write(b">r%d\n" % i + s.tobytes() + b"\n")
    dev = torch.device("cuda")
    whole = U.FastaFile(fn)
    images = {}
    for sparse in ("1", "0"):
        dev_env(monkeypatch, sparse_mask=sparse)
        U.release_ingest_buffers()
        din = U._OnePassInput.create(fn, dev)
        assert din is not None
        din.fill()
        torch.cuda.synchronize()
        codes, mask, slot = din.codes.cpu().numpy(), din.mask.cpu().numpy(), din.ff.slot_off
        assert np.array_equal(din.ff.lengths, whole.lengths) and din.ff.names == whole.names
        for i in range(whole.n):
            a, b = whole.slot_off[i], whole.slot_off[i + 1]
            s0 = int(slot[i])
            assert np.array_equal(codes[s0 * 16:(s0 + b - a) * 16], whole.codes[a * 16:b * 16]), (sparse, i)
            assert np.array_equal(mask[s0 * 8:(s0 + b - a) * 8], whole.mask[a * 8:b * 8]), (sparse, i)
        images[sparse] = U._vectorise(din, 4, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_FREQ_F32).clone()
        din.ff.close()
    assert torch.equal(images["0"], images["1"])
    ref = U._vectorise(U._DeviceInput(whole, dev), 4, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_FREQ_F32)
    assert torch.equal(images["1"], ref)
    U.release_ingest_buffers()


def test_mask_from_lengths_is_the_packers_padding(gpu):
    """idl_mask_from_lengths == the host packer's mask on records without an invalid base, for every tail length 0..130, and it
    leaves flagged records alone."""
    import torch
    from idelucs_amd import _lib, utils as U
    rng = np.random.default_rng(9)
    seqs = [rng.choice(np.frombuffer(b"ACGT", np.uint8), size=L) for L in list(range(0, 131)) + [1000, 4096, 10000]]
    ff = _pack_batch(seqs)
    dev = torch.device("cuda")
    din = U._DeviceInput(ff, dev)
    out = torch.full_like(din.mask, 0x5A)
    sent = torch.zeros(ff.n, dtype=torch.uint8, device=dev)
    sent[7] = 1; sent[100] = 1
    _lib.check(_lib.lib.idl_mask_from_lengths(U._ptr(out), U._ptr(din.slot_off), U._ptr(din.lengths), U._ptr(sent), ff.n, U._stream_ptr()))
    got, want = out.cpu().numpy(), ff.mask
    for i in range(ff.n):
        a, b = int(ff.slot_off[i]) * 8, int(ff.slot_off[i + 1]) * 8
        if i in (7, 100):
            assert np.all(got[a:b] == 0x5A)
        else:
            assert np.array_equal(got[a:b], want[a:b]), (i, len(seqs[i]))
