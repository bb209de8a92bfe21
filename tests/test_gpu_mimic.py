"""GPU tests (-m gpu) of the fast-mode (device Philox) mimic generator: bit-exact against the C
oracle's restatement of the spec, and statistically equivalent to the reference transforms
(idelucs/utils.py:54-135: site probability per base, transition/transversion target law, Random_N)."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _generate(lengths, specs, seed, slots=False):
    import torch
    from idelucs_amd import utils as U
    dev = torch.device("cuda")

    class D:
        pass
    d = D()
    d.n = len(lengths)
    d.codes = torch.zeros(1, dtype=torch.int32, device=dev)
    d.lengths = torch.tensor(lengths, dtype=torch.int64, device=dev)
    edits, off = U._philox_edits(d, specs, seed, slots=slots)
    return edits.cpu().numpy().view(np.uint32), off.cpu().numpy()


def test_one_pass_slots_equal_the_two_pass_protocol():
    """idl_mimic_edits_slots (every site drawn once, into per-item slots; the default) against idl_mimic_edits (count, scan, fill):
    the same edits in the same order for every (view, sequence) -- including items whose lanes hold far more sites than the LDS
    buffer (the in-kernel count-then-fill path), one-base sequences, Random_N views -- and ranges that stay inside their
    slots.  An item that cannot fit its slot raises the overflow flag instead of writing past it."""
    import ctypes
    import torch
    from idelucs_amd import _lib, utils as U
    lengths = [1, 2, 3, 63, 64, 65, 127, 1000, 4096, 4097, 10000, 10000, 33333, 200000, 5]
    specs = [(1e-2, 0.5e-2, 0), (1e-2, 0.0, 0), (0.0, 0.5e-2, 0), (0.0, 0.0, 20), (0.3, 0.3, 0), (1e-6, 0.0, 0), (0.0, 0.0, 64), (0.0, 0.0, 0)]
    n = len(lengths)
    for seed in (0, 2 ** 40 + 12345):
        e2, off = _generate(lengths, specs, seed, slots=False)
        e1, rng = _generate(lengths, specs, seed, slots=True)
        assert off.ndim == 1 and rng.shape == (len(specs) * n, 2)
        prev_end = 0
        for it in range(len(specs) * n):
            a, b = rng[it]
            assert prev_end <= a <= b <= len(e1), (it, a, b)            # slots are disjoint and ascending
            prev_end = b
            assert np.array_equal(e1[a:b], e2[off[it]:off[it + 1]]), (seed, it)
    # overflow: a slot sized for 1 000 bases cannot take a 200 000-base sequence's sites -- flagged, nothing written past the buffer
    dev = torch.device("cuda")
    L = torch.tensor([200000], dtype=torch.int64, device=dev)
    p_ts, p_tv, n_rn = np.array([0.3]), np.array([0.0]), np.array([0], np.int32)
    P = lambda a: ctypes.c_void_p(a.ctypes.data)
    cap = int(_lib.lib.idl_mimic_slots_capacity(1, 1, P(p_ts), P(p_tv), P(n_rn), 1000))
    edits = torch.full((cap + 64,), -1, dtype=torch.int32, device=dev)
    ranges = torch.zeros((1, 2), dtype=torch.int64, device=dev); flag = torch.zeros((), dtype=torch.int32, device=dev)
    ws = torch.empty(int(_lib.lib.idl_mimic_slots_workspace(1)), dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib.idl_mimic_edits_slots(U._ptr(L), 1, 1, P(p_ts), P(p_tv), P(n_rn), ctypes.c_uint64(3), 1000, U._ptr(ranges), U._ptr(edits), cap,
                                              U._ptr(flag), U._ptr(ws), U._stream_ptr()))
    torch.cuda.synchronize()
    assert int(flag.item()) == 1 and ranges.tolist() == [[0, cap]] and bool((edits[cap:] == -1).all())


def test_vectoriser_takes_slot_ranges_like_csr():
    """idl_vectorise_ranges on the one-pass generator's slots == idl_vectorise on the packed CSR edits, bit for bit, through the
    pipelined kernel (k = 6, float32 rows, max_len known), the general kernel (canonical rows) and the accumulate kernel."""
    import torch
    from idelucs_amd import _lib, utils as U
    from conftest import ROOT
    import sys
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda")
    din = bench.synth_packed(3000, 2500, dev, seed=5, n_rate=1e-3)
    specs = [t.spec() for t in U.mimic_transforms(3)]
    e2, off = U._philox_edits(din, specs, 9, slots=False)
    e1, rng = U._philox_edits(din, specs, 9, slots=True)
    assert rng.dim() == 2 and off.dim() == 1
    for mode, init, kind in ((_lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_FREQ_F32), (_lib.MODE_CANONICAL, _lib.INIT_ONE, _lib.OUT_FREQ_F64),
                             (_lib.MODE_KMER, _lib.INIT_ZERO, _lib.OUT_COUNTS_I32)):
        a = U._vectorise(din, 6, mode, init, kind, len(specs), e1, rng)
        b = U._vectorise(din, 6, mode, init, kind, len(specs), e2, off)
        assert torch.equal(a, b), (mode, init, kind)
    acc = torch.ones_like(b)                                               # IDL_INIT_FROM_OUT: the single-pass kernel
    acc2 = acc.clone()
    U._vectorise(din, 6, _lib.MODE_KMER, _lib.INIT_FROM_OUT, _lib.OUT_COUNTS_I32, len(specs), e1, rng, out=acc)
    U._vectorise(din, 6, _lib.MODE_KMER, _lib.INIT_FROM_OUT, _lib.OUT_COUNTS_I32, len(specs), e2, off, out=acc2)
    assert torch.equal(acc, acc2) and torch.equal(acc, b + 1)


def test_bit_exact_vs_oracle():
    lengths = [1, 2, 63, 64, 65, 127, 1000, 4096, 4097, 10000, 10000, 33333, 200000, 5]
    specs = [(1e-2, 0.5e-2, 0), (1e-2, 0.0, 0), (0.0, 0.5e-2, 0), (0.0, 0.0, 20), (0.3, 0.3, 0), (1e-6, 0.0, 0), (0.0, 0.0, 64)]
    for seed in (0, 7, 2 ** 40 + 12345):
        edits, off = _generate(lengths, specs, seed)
        n = len(lengths)
        assert off[0] == 0 and off[-1] == len(edits) or (off[-1] == 0 and len(edits) == 1)
        for v, spec in enumerate(specs):
            for s, L in enumerate(lengths):
                got = edits[off[v * n + s]:off[v * n + s + 1]]
                want = O.mimic_edits(L, s, v, spec, seed)
                assert np.array_equal(got, want), (seed, v, s, L, len(got), len(want))
                pos = got & 0x3FFFFFFF
                assert np.all(pos < max(L, 1)) and np.all(np.diff(pos.astype(np.int64)) >= 0)       # in range, sorted


def test_statistics_match_reference_transforms():
    n, L = 2000, 10000
    specs = [(1e-2, 0.5e-2, 0), (1e-2, 0.0, 0), (0.0, 0.5e-2, 0), (0.0, 0.0, 20)]
    edits, off = _generate([L] * n, specs, 99)
    tot = n * L

    def view(v):
        e = edits[off[v * n]:off[(v + 1) * n]]
        return e & 0x3FFFFFFF, e >> 30
    # view 1: transition only, rate 1e-2, op always 2 (A<->G, C<->T)
    pos, op = view(1)
    assert np.all(op == 2) and abs(len(pos) / tot - 1e-2) < 4 * np.sqrt(1e-2 / tot)
    # view 2: transversion only, rate 0.5e-2, the two targets equally likely (random.choice, utils.py:118)
    pos, op = view(2)
    assert set(np.unique(op)) == {1, 3} and abs(len(pos) / tot - 0.5e-2) < 4 * np.sqrt(0.5e-2 / tot)
    assert abs(np.mean(op == 1) - 0.5) < 4 * 0.5 / np.sqrt(len(op))
    # view 0: both passes: P(site) = 1-(1-p1)(1-p2); type shares ts-only : tv-only : both = p1(1-p2) : (1-p1)p2 : p1p2
    pos, op = view(0)
    q = 1 - (1 - 1e-2) * (1 - 0.5e-2)
    assert abs(len(pos) / tot - q) < 4 * np.sqrt(q / tot)
    # ts-only -> op 2; tv-only -> 1|3; both -> 2^(1|3) = 3|1  => P(op==2) = p1(1-p2)/q
    assert abs(np.mean(op == 2) - 1e-2 * (1 - 0.5e-2) / q) < 4 * 0.5 / np.sqrt(len(op))
    # positions uniform over the sequence (10 bins)
    h = np.bincount((pos.astype(np.int64) * 10 // L), minlength=10)
    assert np.all(np.abs(h - len(pos) / 10) < 5 * np.sqrt(len(pos) / 10))
    # gaps between consecutive sites inside a sequence are geometric: mean 1/q
    e0 = edits[off[0]:off[1]] & 0x3FFFFFFF
    assert len(e0) > 50
    # view 3: Random_N(20): exactly 20 draws per sequence, uniform, duplicates allowed, op 0
    pos, op = view(3)
    assert np.all(op == 0) and len(pos) == 20 * n
    h = np.bincount((pos.astype(np.int64) * 10 // L), minlength=10)
    assert np.all(np.abs(h - len(pos) / 10) < 5 * np.sqrt(len(pos) / 10))
    # different seeds / sequences decorrelate
    e2, off2 = _generate([L] * 4, specs, 100)
    assert not np.array_equal(e2[off2[0]:off2[1]], edits[off[0]:off[1]])


def test_feature_level_equivalence_of_rng_modes(tmp_path):
    """Philox and compat (reference-stream) mimics give statistically the same features: mean L1 distance
    between the "true" view and each mimic view agrees within 10 % on real sequences."""
    import os
    import random
    from conftest import DATA
    from idelucs_amd import utils as U
    fn = os.path.join(DATA, "influenza_64.fas")
    np.random.seed(0); random.seed(0)
    a = U.build_feature_store(fn, 3, k=4, rng="compat").feats.cpu().numpy()
    b = U.build_feature_store(fn, 3, k=4, rng="philox", seed=3).feats.cpu().numpy()
    assert a.shape == b.shape == (4, 64, 256)
    for v in (1, 2, 3):
        da = np.abs(a[v] - a[0]).sum(1).mean()
        db = np.abs(b[v] - b[0]).sum(1).mean()
        assert abs(da - db) / da < 0.10, (v, da, db)


def test_skewed_lengths_take_the_exact_protocol_not_the_slots():
    """the one-pass generator sizes EVERY item's slot from the longest sequence.  One 300 kbp record among
    3 000 records of 200 bp would ask for n * P * cap(300 kbp) entries; the default now leaves the slots when that exceeds a small
    multiple of the expected sites (utils._slots_budget) and takes the exact CSR protocol -- the same sites, bit for bit, as
    the forced slots give on this (still allocatable) case.  The uniform-length hot shape keeps the slots."""
    import torch
    from idelucs_amd import utils as U
    lengths = [200] * 3000 + [300000]
    specs = [t.spec() for t in U.mimic_transforms(3)]
    dev = torch.device("cuda")

    class D:
        pass
    d = D()
    d.n, d.lengths = len(lengths), torch.tensor(lengths, dtype=torch.int64, device=dev)
    d.codes = torch.zeros(1, dtype=torch.int32, device=dev)
    d.total_len, d.max_len = sum(lengths), max(lengths)
    e_def, off_def = U._philox_edits(d, specs, 5)
    assert off_def.dim() == 1, "a skewed-length input still took the per-item slots"
    assert e_def.numel() < 4 * 0.04 * d.total_len + 64 * d.n * len(specs)          # proportional to the bases, not to n * cap(max)
    e_s, rng = U._philox_edits(d, specs, 5, slots=True)
    assert rng.dim() == 2 and e_s.numel() > 20 * e_def.numel()
    e_def, off_def, e_s, rng = e_def.cpu().numpy(), off_def.cpu().numpy(), e_s.cpu().numpy(), rng.cpu().numpy()
    for it in list(range(0, len(specs) * d.n, 97)) + [v * d.n + d.n - 1 for v in range(len(specs))]:
        assert np.array_equal(e_s[rng[it, 0]:rng[it, 1]], e_def[off_def[it]:off_def[it + 1]]), it
    # uniform lengths (the hot shape): the slots stay the default
    u = D()
    u.n, u.lengths = 500, torch.full((500,), 10000, dtype=torch.int64, device=dev)
    u.codes, u.total_len, u.max_len = d.codes, 500 * 10000, 10000
    assert U._philox_edits(u, specs, 5)[1].dim() == 2
