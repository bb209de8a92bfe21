"""idelucs_amd.utils -- the data layer of the hot path, behind the reference's names.

Mirrors reference idelucs/utils.py:26-429 (check_sequence, SummaryFasta, reverse_complement,
kmer_rev_comp, kmersFasta, cgrFasta, the four mimic transforms, AugmentFasta, AugmentedDataset,
SequenceDataset, create_dataloader): same names, arguments, return types and errors.  What differs
is where the work happens: the FASTA file is parsed and 2-bit packed ONCE by the C++ host reader in
libidelucs_hip.so, every counting / collapsing / normalising / scaling step runs on the MI355X, and
the feature matrix stays resident in HBM as a de-duplicated [views, N, F] store.

Mimic RNG modes (argument `rng`, default from $IDELUCS_RNG, else "philox"):
  "compat"  the host draws sites exactly like the reference (numpy global MT19937 + Python `random`,
            same consumption order, utils.py:69-118 / survey D.2) and ships them to the device as
            substitution edits -> outputs match the reference bit-for-bit up to the scaler's float64
            summation order;
  "philox"  sites are drawn on the device (idl_mimic_edits): statistically equivalent, ~1000x faster.
"""
import ctypes
import os
import random
import sys

import numpy as np
import torch

from . import _lib
from ._lib import lib as _L

A, C, G, T, N = (ord(c) for c in "ACGTN")
_CODE = np.full(256, 4, np.uint8)
_CODE[[A, C, G, T]] = [0, 1, 2, 3]


# Variants kept for the tests that compare them with the default path (not environment switches; a test sets an entry with monkeypatch.setitem)
OPTIONS = {"mimic_slots": "1",       # 0: the exact two-call protocol of the site generator (count, scan, fill) instead of one pass into slots
           "predict_counts": "1",    # 0: predict inputs through float64 rows instead of int32 counts
           "one_pass": "1"}          # 0: the general three-pass FASTA reader instead of idl_fasta_parse_pack
OPTIONS.update({k: v for k, v in _lib.DEV.items() if k in OPTIONS})

def _default_rng_mode():
    return os.environ.get("IDELUCS_RNG", "philox")


def _device(device=None):
    _lib.require_gpu()
    return torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())


def _stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        return ctypes.c_void_p(t.data_ptr())
    return ctypes.c_void_p(t.ctypes.data)


# ------------------------------------------------------------------------------------------------
# check_sequence / FASTA reading (host C++)
# ------------------------------------------------------------------------------------------------
def check_sequence(header, seq):
    """Reference idelucs/utils.py:26-51: validate the header, upper-case, U->T, IUPAC/'-' -> N,
    delete whitespace, reject anything else.  Returns a new bytearray."""
    if len(header) > 0 and (header[0] in (">", "#") or header[0].isspace()):
        raise ValueError("Bad character in sequence header")
    if "\t" in header:
        raise ValueError("tab included in header")
    src = np.frombuffer(bytes(seq), dtype=np.uint8)
    out = np.empty(max(src.size, 1), np.uint8)
    out_len, bad = ctypes.c_int64(0), ctypes.c_int64(-1)
    rc = _L.idl_check_sequence(_ptr(src) if src.size else None, src.size, _ptr(out), ctypes.byref(out_len), ctypes.byref(bad))
    if rc == _lib.IDL_ERR_BASE:
        # utils.py:46-50 formats the first byte of the translated string that is not in ACGTN
        raise ValueError("Invalid DNA byte in sequence {}: '{}'".format(header, chr(int(src[bad.value]))))
    _lib.check(rc)
    return bytearray(out[:out_len.value].tobytes())


class FastaFile:
    """One parse of a FASTA file by the C++ reader (idl_fasta_open): names, lengths, cleaned bytes
    (optional) and the packed slot layout.  Replaces the reference's re-parsing of the file in every
    pass (utils.py:137-188, :224-260, :279-317)."""

    def __init__(self, fname, check=True, keep_bytes=False, pack=True):
        """pack: True = packed codes/mask exported now; False = no packed data; "deferred" = the parsed file stays open and
        record ranges are packed on request (pack_range) -- the streamed ingest of build_feature_store."""
        h = ctypes.c_void_p()
        _lib.check(_L.idl_fasta_open(os.fsencode(fname), 1 if check else 0, ctypes.byref(h)))
        self._load(h, check, keep_bytes, pack)

    @classmethod
    def from_handle(cls, h, arena=False, meta=None):
        """A FastaFile over an idl_fasta handle the caller opened (idl_fasta_parse_pack: arena = True -> slot_off are the records'
        first slots in the caller's arenas); the handle is closed.
        meta = (lengths, slot_off) already read out with idl_fasta_arena_meta (the one-pass ingest: they go to the device first):
        the names are then exported on first use or at close() -- off the path between the file and the first kernel; a file with a non-ASCII header byte
        (idl_fasta_names_high) exports and validates them here (invalid UTF-8 / a unicode space heading a name fail when the file is read)."""
        obj = cls.__new__(cls)
        if arena and meta is not None:
            n, tb, ts, nb = (ctypes.c_int64() for _ in range(4))
            _lib.check(_L.idl_fasta_sizes(h, ctypes.byref(n), ctypes.byref(tb), ctypes.byref(ts), ctypes.byref(nb)))
            obj.n, obj.total_bases, obj.total_slots = n.value, tb.value, ts.value
            obj.lengths, obj.slot_off = meta
            obj.byte_off = obj.bytes = obj.codes = obj.mask = None
            obj._names_raw = obj._name_off = obj._names = None
            obj._names_bytes, obj._check = nb.value, True
            obj._h = h
            if int(_L.idl_fasta_names_high(h)) != 0:      # a non-ASCII header (rare): the checks on the decoded names run now, as the reference's do while it reads
                obj._export_names()
            return obj
        arena_slots = None
        if arena:
            n = ctypes.c_int64()
            _lib.check(_L.idl_fasta_sizes(h, ctypes.byref(n), None, None, None))
            arena_slots = np.empty(n.value + 1, np.int64)
            _lib.check(_L.idl_fasta_arena_slots(h, _ptr(arena_slots)))
        obj._load(h, True, False, "deferred" if arena else False)
        if arena:
            obj.slot_off = arena_slots
        return obj

    def _export_names(self):
        """The deferred half of from_handle(meta=...): names out of the open handle, with _load's checks."""
        names = np.empty(max(self._names_bytes, 1), np.uint8)
        name_off = np.empty(self.n + 1, np.int64)
        _lib.check(_L.idl_fasta_export(self._h, _ptr(names), _ptr(name_off), None, None, None, None, None, None))
        self._names_raw, self._name_off = names, name_off
        self._check_names(self._names_bytes, self._check)

    def _check_names(self, nb, check):
        if self.n and int(self._names_raw[:nb].max(initial=0)) >= 0x80:
            # non-ASCII names: decode now, so that invalid UTF-8 fails here as in the reference, and apply the one header check the
            # byte-level reader cannot do (unicode whitespace as first character)
            if check and any(len(nm) > 0 and nm[0].isspace() for nm in self.names):
                self.close(export=False)         # (export=False: this may BE close()'s export -- the handle is closed once, here)
                raise ValueError("Bad character in sequence header")

    def _load(self, h, check, keep_bytes, pack):
        self._h = None
        try:
            n, tb, ts, nb = (ctypes.c_int64() for _ in range(4))
            _lib.check(_L.idl_fasta_sizes(h, ctypes.byref(n), ctypes.byref(tb), ctypes.byref(ts), ctypes.byref(nb)))
            self.n, self.total_bases, self.total_slots = n.value, tb.value, ts.value
            names = np.empty(max(nb.value, 1), np.uint8)
            name_off = np.empty(self.n + 1, np.int64)
            self.lengths = np.empty(self.n, np.int64)
            self.byte_off = np.empty(self.n + 1, np.int64)
            self.bytes = np.empty(max(self.total_bases, 1), np.uint8) if keep_bytes else None
            if pack is True:
                self.codes = np.empty(max(self.total_slots, 1) * 16, np.uint8)
                self.mask = np.empty(max(self.total_slots, 1) * 8, np.uint8)
                self.slot_off = np.empty(self.n + 1, np.int64)
            else:
                self.codes = self.mask = self.slot_off = None
            _lib.check(_L.idl_fasta_export(h, _ptr(names), _ptr(name_off), _ptr(self.lengths), _ptr(self.bytes),
                                           _ptr(self.byte_off), _ptr(self.codes), _ptr(self.mask), _ptr(self.slot_off)))
            if pack == "deferred":
                self.slot_off = np.zeros(self.n + 1, np.int64)
                np.cumsum((self.lengths + 63) // 64, out=self.slot_off[1:])
                self._h, h = h, None
        finally:
            if h is not None:
                _L.idl_fasta_close(h)
        # names as Python strings (utils.py:172 .decode()) are built on first use: 100 000 decodes cost 12 ms, and the training
        # path needs them only when results are written.  The checks that need decoded text run now, on the few names concerned.
        self._names_raw, self._name_off, self._names = names, name_off, None
        self._check_names(nb.value, check)

    @property
    def names(self):
        if self._names is None:
            if self._names_raw is None:
                if self._h is None:
                    raise ValueError("the names of this file were never read out")
                self._export_names()
            raw, off = self._names_raw.tobytes(), self._name_off
            self._names = [raw[off[i]:off[i + 1]].decode() for i in range(self.n)]
        return self._names

    def pack_range(self, lo, hi, codes, mask):
        """Translate + 2-bit pack records [lo, hi) into the whole-file buffers `codes` / `mask` (host uint8 tensors or arrays
        of total_slots * 16 / * 8 bytes) at their slot offsets (idl_fasta_pack_range); needs pack="deferred"."""
        if self._h is None:
            raise ValueError("pack_range needs FastaFile(pack='deferred')")
        _lib.check(_L.idl_fasta_pack_range(self._h, int(lo), int(hi), _ptr(codes), _ptr(mask)))

    def close(self, export=True):
        if getattr(self, "_h", None) is not None:
            if export and getattr(self, "_names_raw", 0) is None:      # names deferred by from_handle(meta=...): last chance to read them
                h, self._h = self._h, None
                try:
                    self._h = h
                    self._export_names()
                finally:
                    if self._h is not None:      # (_check_names closes the handle itself before it raises)
                        self._h = None
                        _L.idl_fasta_close(h)
                return
            _L.idl_fasta_close(self._h)
            self._h = None

    def __del__(self):
        self.close(export=False)

    def record(self, i):
        return bytearray(self.bytes[self.byte_off[i]:self.byte_off[i + 1]].tobytes())


def SummaryFasta(fname, GT_file=None):
    """Reference idelucs/utils.py:137-188 -> (names, lengths, ground_truth | None, cluster_dis | None)."""
    gt_dict = cluster_dis = ground_truth = None
    if GT_file:
        import pandas as pd
        df = pd.read_csv(GT_file, sep="\t")
        gt_dict = dict(zip(df.sequence_id, df.cluster_id))
        cluster_dis = df["cluster_id"].value_counts().to_dict()
    ff = FastaFile(fname, check=True, pack=False)
    if GT_file:
        ground_truth = []
        for name in ff.names:
            if name not in gt_dict:
                raise ValueError("Check GT for sequence {}".format(name))
            ground_truth.append(gt_dict[name])
    return ff.names, ff.lengths.tolist(), ground_truth, cluster_dis


# ------------------------------------------------------------------------------------------------
# reverse complement helpers
# ------------------------------------------------------------------------------------------------
def reverse_complement(x, k):
    """Reference idelucs/utils.py:191-206 (index of the reverse-complement k-mer, A0 C1 G2 T3)."""
    x, rc = int(x), 0
    for _ in range(k):
        rc = (rc << 2) | (3 - (x & 3))
        x >>= 2
    return rc


def kmer_rev_comp(kmer_counts, k):
    """Reference idelucs/utils.py:208-221: canonical collapse with int truncation; `kmer_counts` is
    modified in place like the reference and the canonical entries are returned (device kernel)."""
    arr = np.asarray(kmer_counts)
    if arr.dtype != np.int32 or not arr.flags.c_contiguous or arr.ndim != 1 or arr.size < 4 ** k:
        raise ValueError("kmer_counts must be a contiguous int32 vector of 4**k entries")
    _lib.require_gpu()
    out = np.empty(_L.idl_row_len(_lib.MODE_CANONICAL, k), np.int32)
    _lib.check(_L.idl_kmer_rev_comp(_ptr(arr), int(k), _ptr(out)))
    return out


# ------------------------------------------------------------------------------------------------
# mimic transforms: host-RNG ("compat") application + parameters for the device generator
# ------------------------------------------------------------------------------------------------
def _as_u8(seq):
    return np.frombuffer(seq, dtype=np.uint8)   # writable view of a bytearray


_TRANSITION_LUT = np.arange(256, dtype=np.uint8)
_TRANSITION_LUT[[A, G, T, C]] = [G, A, C, T]
_TRANSVERSION_TABLE = {A: [T, C], G: [T, C], T: [A, G], C: [A, G], N: [N]}


class transition(object):
    """Reference idelucs/utils.py:54-76: each base mutates w.p. `threshold`; A<->G, C<->T, N->N."""

    def __init__(self, threshold):
        self.threshold = threshold

    def spec(self):
        return (float(self.threshold), 0.0, 0)

    def __call__(self, seq):
        a = _as_u8(seq)
        x = np.random.random(len(seq))
        idx = np.flatnonzero(x < self.threshold)
        bad = np.setdiff1d(a[idx], [A, C, G, T, N])
        if bad.size:
            raise KeyError(int(bad[0]))          # the reference's dict lookup (utils.py:76) raises KeyError
        a[idx] = _TRANSITION_LUT[a[idx]]


class transversion(object):
    """Reference idelucs/utils.py:98-118: site w.p. `threshold`; purine -> T|C, pyrimidine -> A|G
    via random.choice (an N site still consumes one choice)."""

    def __init__(self, threshold):
        self.threshold = threshold

    def spec(self):
        return (0.0, float(self.threshold), 0)

    def __call__(self, seq):
        x = np.random.random(len(seq))
        for i in np.flatnonzero(x < self.threshold):
            seq[i] = random.choice(_TRANSVERSION_TABLE.get(seq[i], [N]))


class transition_transversion(object):
    """Reference idelucs/utils.py:120-135: transition(threshold_1) then transversion(threshold_2)."""

    def __init__(self, threshold_1, threshold_2):
        self.tf1 = transition(threshold_1)
        self.tf2 = transversion(threshold_2)

    def spec(self):
        return (float(self.tf1.threshold), float(self.tf2.threshold), 0)

    def __call__(self, seq):
        self.tf1(seq)
        self.tf2(seq)


class Random_N(object):
    """Reference idelucs/utils.py:78-95: `n_bp` uniformly drawn positions (with replacement) -> N."""

    def __init__(self, n_bp):
        self.n_bp = n_bp

    def spec(self):
        return (0.0, 0.0, int(self.n_bp))

    def __call__(self, seq):
        a = _as_u8(seq)
        a[np.random.randint(0, len(seq), self.n_bp)] = N


class _NotSubstitutions(ValueError):
    """A transform did something other than ACGTN substitutions; .mutated holds the records mutated so far (the transform has
    already consumed its random numbers for them: they must not be mutated a second time)."""

    def __init__(self, msg, mutated):
        super().__init__(msg)
        self.mutated = mutated


def _edits_from_diff(orig, mutated):
    """Substitution edits (pos | op<<30) that turn `orig` into `mutated` (equal-length ACGTN byte
    arrays): op 0 = becomes N, else XOR of the 2-bit codes."""
    pos = np.flatnonzero(orig != mutated)
    if pos.size == 0:
        return np.empty(0, np.uint32)
    if pos[-1] >= (1 << 30):
        raise ValueError("mutated sequences longer than 2^30 bases are not supported")
    co, cm = _CODE[orig[pos]], _CODE[mutated[pos]]
    if np.any(co == 4) or np.any((cm == 4) & (mutated[pos] != N)):
        return None   # not expressible as edits (an invalid base became valid, or a non-ACGTN byte appeared)
    op = np.where(cm == 4, 0, co ^ cm).astype(np.uint32)
    return pos.astype(np.uint32) | (op << np.uint32(30))


# ------------------------------------------------------------------------------------------------
# device vectorisation of a parsed file
# ------------------------------------------------------------------------------------------------
class _DeviceInput:
    """Packed bases of one FASTA file, resident in HBM."""

    def __init__(self, ff, device):
        self.n = ff.n
        self.codes = torch.from_numpy(ff.codes).to(device)
        self.mask = torch.from_numpy(ff.mask).to(device)
        self.slot_off = torch.from_numpy(ff.slot_off).to(device)
        self.lengths = torch.from_numpy(ff.lengths).to(device)
        self.max_len = int(ff.lengths.max()) if ff.n else 0
        self.total_len = int(ff.lengths.sum()) if ff.n else 0


def _vectorise(dev_in, k, mode, init, out_kind, n_views=1, edits=None, edit_off=None, out=None):
    """idl_vectorise on torch-owned device buffers -> tensor [n_views, n, row_len]."""
    row = _L.idl_row_len(mode, k)
    if row < 0 or not 1 <= k <= _lib.MAX_K:
        raise ValueError(f"k={k} is outside 1..{_lib.MAX_K}")
    dtype = {_lib.OUT_COUNTS_I32: torch.int32, _lib.OUT_FREQ_F32: torch.float32, _lib.OUT_FREQ_F64: torch.float64}[out_kind]
    if out is None:
        out = torch.empty((n_views, dev_in.n, row), dtype=dtype, device=dev_in.codes.device)
    assert out.is_contiguous() and out.dtype == dtype and tuple(out.shape) == (n_views, dev_in.n, row)
    # edit_off [n_views * n + 1]: CSR;  [n_views * n, 2]: (begin, end) per item (the one-pass generator's slots)
    entry = _L.idl_vectorise_ranges if (edit_off is not None and edit_off.dim() == 2) else _L.idl_vectorise
    _lib.check(entry(_ptr(dev_in.codes), _ptr(dev_in.mask), _ptr(dev_in.slot_off), _ptr(dev_in.lengths),
                     dev_in.n, k, mode, init, out_kind, n_views, _ptr(edits), _ptr(edit_off),
                     _ptr(out), dev_in.n * row, int(getattr(dev_in, "max_len", 0) or 0), _stream_ptr()))
    return out


def _compat_edits(ff, transforms):
    """Host-RNG mimic sites for every (view, record), in the reference's consumption order
    (pass-major, records in file order; survey D.2).  `transforms[v]` is None or a callable that
    mutates a bytearray in place.  Returns (edits uint32, edit_off int64[n_views*n+1]) or raises if
    a transform cannot be expressed as substitutions."""
    chunks, counts = [], []
    for tf in transforms:
        done = []
        for i in range(ff.n):
            if tf is None:
                counts.append(0)
                continue
            orig = ff.bytes[ff.byte_off[i]:ff.byte_off[i + 1]]
            mut = bytearray(orig.tobytes())
            tf(mut)
            done.append(mut)
            if len(mut) != orig.size:
                raise _NotSubstitutions("a transform changed the length of a sequence", done)
            e = _edits_from_diff(orig, np.frombuffer(mut, np.uint8))
            if e is None:
                raise _NotSubstitutions("a transform produced bytes that are not substitutions among ACGTN", done)
            chunks.append(e)
            counts.append(e.size)
    edit_off = np.zeros(len(counts) + 1, np.int64)
    np.cumsum(counts, out=edit_off[1:])
    edits = np.concatenate(chunks) if chunks else np.empty(0, np.uint32)
    return edits, edit_off


def edits_overflowed(edits, edit_off, capacity=None):
    """Device bool: the edits of a no-sync _philox_edits call did not fit (slots: an item outgrew its slot; two-pass protocol with a
    caller's capacity: the total outgrew it).  The run that used them is then invalid."""
    flag = getattr(edits, "overflow_flag", None)
    if flag is not None:
        return flag != 0
    return edit_off[-1] > capacity if capacity is not None else torch.zeros((), dtype=torch.bool, device=edits.device)


def _slots_budget(dev_in, p_ts, p_tv, n_rn, max_len):
    """Entries the one-pass generator's slot buffer may take: 4 x the expected number of sites (from the bases actually present;
    an input that does not say is taken as n sequences of max_len) + 64 per item.  At uniform lengths the slots need 2-3 x the
    expectation (expected + 10 sigma + 32 per item) and pass; a skewed-length file does not and takes the exact protocol."""
    n, P = dev_in.n, len(p_ts)
    tot = getattr(dev_in, "total_len", None)
    if tot is None:
        tot = n * max_len
    q = 1.0 - (1.0 - p_ts) * (1.0 - p_tv)
    return int(4.0 * (float(q.sum()) * tot + float(n_rn.sum()) * n) + 64 * n * P + 1024)


def _philox_edits(dev_in, specs, seed, capacity=None, slots=None, sync=None):
    """Device-drawn mimic sites -> (edits, edit_off) on device.
    Default (slots; IDELUCS_DEV=mimic_slots=0 turns it off; slots=True forces it; otherwise only while the slot buffer stays within
    _slots_budget -- one long record among many short ones goes to the exact protocol): ONE pass (idl_mimic_edits_slots) into fixed per-item slots sized on the
    host from the longest sequence; edit_off is then the [n_views * n, 2] array of (begin, end) _vectorise understands.  An item
    that outgrows its slot (expected sites + 10 sigma + 32) raises a device flag: with sync (the default without a capacity)
    it is read here and the exact two-call protocol takes over; without, the caller checks edits_overflowed() afterwards.
    Two-call protocol (idl_mimic_edits): capacity=None: exact sizing (one host round trip to read the total).  With a capacity
    (an upper bound chosen by the caller) nothing synchronises; edit_off[-1] holds the true total and the caller must check it
    against the capacity afterwards (the fill kernel never writes past it)."""
    P = len(specs)
    p_ts = np.array([s[0] for s in specs], np.float64)
    p_tv = np.array([s[1] for s in specs], np.float64)
    n_rn = np.array([s[2] for s in specs], np.int32)
    if np.any(n_rn > 0) and dev_in.n > 0 and getattr(dev_in, "min_len", None) is None:
        dev_in.min_len = int(dev_in.lengths.min().item())
    if np.any(n_rn > 0) and dev_in.n > 0 and dev_in.min_len <= 0:
        raise ValueError("high <= 0")   # np.random.randint(0, 0, n) in the reference's Random_N (utils.py:93)
    # limits of the device generator (csrc/mimic.hip): positions are packed into 30 bits, per-lane site counts into 16
    max_len = int(getattr(dev_in, "max_len", 0) or (dev_in.lengths.max().item() if dev_in.n else 0))
    try:
        _lib.check(_L.idl_mimic_check_lengths(max_len, P, _ptr(p_ts), _ptr(p_tv)))          # the precondition idl_mimic_edits states
    except ValueError as err:
        raise ValueError(f"{err}; use rng='compat'") from None
    dev = dev_in.lengths.device
    args = (_ptr(dev_in.lengths), dev_in.n, P, _ptr(p_ts), _ptr(p_tv), _ptr(n_rn), ctypes.c_uint64(seed & (2 ** 64 - 1)))
    forced = slots is True
    if slots is None:
        slots = OPTIONS["mimic_slots"] != "0"
    edits = None
    if slots and dev_in.n > 0:
        total = int(_L.idl_mimic_slots_capacity(dev_in.n, P, _ptr(p_ts), _ptr(p_tv), _ptr(n_rn), max_len))
        # every item's slot is sized from the LONGEST sequence: n * P * cap(max_len) entries, not proportional to the bases.  One
        # long outlier among many short records (which the reference accepts) would ask for tens of GB where the exact CSR protocol
        # needs a few MB: the slots are taken only while they stay within a small multiple of the expected sites
        if not forced and total > _slots_budget(dev_in, p_ts, p_tv, n_rn, max_len):
            slots = False
    if slots and dev_in.n > 0:
        ws = torch.empty(max(int(_L.idl_mimic_slots_workspace(P)), 16), dtype=torch.uint8, device=dev)
        ranges = torch.empty((P * dev_in.n, 2), dtype=torch.int64, device=dev)
        try:
            edits = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
        except torch.cuda.OutOfMemoryError:
            if forced:
                raise
            edits = None                      # the exact two-pass protocol below sizes everything from the counts
    if edits is not None:
        flag = torch.zeros((), dtype=torch.int32, device=dev)
        _lib.check(_L.idl_mimic_edits_slots(*args, max_len, _ptr(ranges), _ptr(edits), total, _ptr(flag), _ptr(ws), _stream_ptr()))
        if sync if sync is not None else capacity is None:
            if int(flag.item()) == 0:
                return edits, ranges
            # (an item beyond +10 sigma: never seen; the exact protocol below sizes everything from the counts)
        else:
            edits.overflow_flag = flag
            return edits, ranges
    ws = torch.empty(_L.idl_mimic_workspace(dev_in.n, P), dtype=torch.uint8, device=dev)
    edit_off = torch.empty(P * dev_in.n + 1, dtype=torch.int64, device=dev)
    if capacity is None:
        total = ctypes.c_int64(0)
        _lib.check(_L.idl_mimic_edits(*args, _ptr(edit_off), None, 0, ctypes.byref(total), _ptr(ws), _stream_ptr()))
        capacity = total.value
    else:
        _lib.check(_L.idl_mimic_edits(*args, _ptr(edit_off), None, 0, None, _ptr(ws), _stream_ptr()))
    edits = torch.empty(max(int(capacity), 1), dtype=torch.int32, device=dev)
    _lib.check(_L.idl_mimic_edits(*args, _ptr(edit_off), _ptr(edits), int(capacity), None, _ptr(ws), _stream_ptr()))
    return edits, edit_off


def _features_one_pass(fname, k, transform, mode, check, out_kind, device=None, rng=None, seed=0):
    """One kmersFasta/cgrFasta-style pass: names + [N, row] device tensor."""
    rng = rng or _default_rng_mode()
    dev = _device(device)
    spec = transform.spec() if hasattr(transform, "spec") else None
    if spec is not None and spec[2] > _L.idl_mimic_max_random_n():
        spec = None                           # more Random_N draws than the device generator sorts in one wave: draw on the host
    host_tf = transform is not None and (rng == "compat" or spec is None)
    ff = FastaFile(fname, check=check, keep_bytes=host_tf)
    edits = edit_off = None
    if host_tf:
        try:
            e, eo = _compat_edits(ff, [transform])
        except _NotSubstitutions as err:      # (a ValueError raised by the user's own transform passes through untouched)
            # an arbitrary user callable (reference utils.py:239-240 takes any): apply it on the host, re-pack the mutated bytes
            return _features_user_transform(ff, k, transform, mode, out_kind, dev, applied=err.mutated)
        edits = torch.from_numpy(e.view(np.int32)).to(dev) if e.size else torch.zeros(1, dtype=torch.int32, device=dev)
        edit_off = torch.from_numpy(eo).to(dev)
    din = _DeviceInput(ff, dev)
    if transform is not None and not host_tf:
        edits, edit_off = _philox_edits(din, [spec], seed)
    out = _vectorise(din, k, mode, _lib.INIT_ONE, out_kind, 1, edits, edit_off)
    return ff.names, out[0]


def _pack_host(seqs):
    """Byte strings -> an object with the packed device-input fields as host arrays (idl_pack: bytes other than ACGT invalid,
    as kmers.pyx:19-34 treats them)."""
    n = len(seqs)
    lengths = np.array([len(x) for x in seqs], np.int64)
    byte_off = np.zeros(n + 1, np.int64)
    np.cumsum(lengths, out=byte_off[1:])
    flat = np.frombuffer(b"".join(bytes(x) for x in seqs), np.uint8) if byte_off[-1] else np.zeros(1, np.uint8)
    slots = int(((lengths + 63) // 64).sum())

    class Packed:
        pass
    pk = Packed()
    pk.n, pk.lengths = n, lengths
    pk.codes = np.empty(max(slots, 1) * 16, np.uint8)
    pk.mask = np.empty(max(slots, 1) * 8, np.uint8)
    pk.slot_off = np.empty(n + 1, np.int64)
    _lib.check(_L.idl_pack(_ptr(flat), _ptr(byte_off), n, _ptr(pk.codes), _ptr(pk.mask), _ptr(pk.slot_off)))
    return pk


def _features_user_transform(ff, k, transform, mode, out_kind, dev, applied):
    """kmersFasta / cgrFasta with a transform that is not a set of base substitutions (it inserts, deletes, writes other bytes...):
    the callable runs on the host on every cleaned record, as in the reference, and the result is packed and counted."""
    seqs = list(applied)
    for i in range(len(seqs), ff.n):
        b = ff.record(i)
        transform(b)
        seqs.append(b)
    out = _vectorise(_DeviceInput(_pack_host(seqs), dev), k, mode, _lib.INIT_ONE, out_kind)
    return ff.names, out[0]


def kmersFasta(fname, k=6, transform=None, reduce=False, rng=None, seed=0, project=False):
    """Reference idelucs/utils.py:224-277 -> (names, float64 [N, 4^k or n_canonical]).
    project=True (or a path to a kernel .npz): the compositional projection the reference keeps commented out at utils.py:272-275,
    `np.dot(kmers, kernels/kernel{k}.npz['arr_0'])` -> float64 [N, 135 / 511 / 2079] for k = 4 / 5 / 6.  The kernel has 4^k rows, so
    it applies to the un-collapsed frequency rows: with project the canonical collapse of `reduce` is not performed (the
    projection is the reduction)."""
    if project:
        names, feats = _features_one_pass(fname, k, transform, _lib.MODE_KMER, True, _lib.OUT_FREQ_F64, rng=rng, seed=seed)
        kernel = project if isinstance(project, (str, os.PathLike)) else kernel_file(k)
        return names, project_kernel(feats, kernel).cpu().numpy()
    mode = _lib.MODE_CANONICAL if reduce else _lib.MODE_KMER
    names, feats = _features_one_pass(fname, k, transform, mode, True, _lib.OUT_FREQ_F64, rng=rng, seed=seed)
    return names, feats.cpu().numpy()


def cgrFasta(fname, k=6, transform=None, rng=None, seed=0):
    """Reference idelucs/utils.py:279-317 (no check_sequence: lower-case bytes are skipped)."""
    names, feats = _features_one_pass(fname, k, transform, _lib.MODE_CGR, False, _lib.OUT_FREQ_F64, rng=rng, seed=seed)
    return names, feats.cpu().numpy()


# ------------------------------------------------------------------------------------------------
# the de-duplicated feature store (what AugmentFasta's [N*n_mimics, 2, F] array is a view of)
# ------------------------------------------------------------------------------------------------
def mimic_transforms(n_mimics):
    """The passes of reference AugmentFasta (utils.py:330-351): view 0 = "true" view (itself mutated),
    then transition, transversion, and max(n_mimics-2, 0) Random_N(20) views."""
    return ([transition_transversion(1e-2, 0.5e-2), transition(1e-2), transversion(0.5e-2)]
            + [Random_N(20) for _ in range(n_mimics - 2)])


class FeatureStore:
    """feats[P, N, F] float32 un-scaled frequencies (view 0 = "true") + the scaler fitted on view 0,
    all resident in HBM.  Pair p of the reference's x_train (utils.py:353) is
    (feats[0][p % N], feats[1 + p // N][p % N]), standardised."""

    def __init__(self, names, lengths, feats, mean, scale, k, reduce):
        self._names, self.lengths = names, lengths           # names: a list, or a FastaFile (its names are decoded on first use)
        self.feats, self.mean, self.scale = feats, mean, scale
        self.k, self.reduce = k, reduce
        self.n_views, self.n, self.f = feats.shape
        self.n_pairs = (self.n_views - 1) * self.n
        self.inv_scale = torch.empty_like(scale)
        self.refresh()

    @property
    def names(self):
        return self._names.names if isinstance(self._names, FastaFile) else self._names

    def refresh(self):
        """Recompute what is derived from `scale` IN PLACE (after col_stats(..., out=(mean, scale)) refitted the scaler into the
        same buffers): the addresses a captured training-step graph holds stay valid."""
        torch.div(1.0, self.scale, out=self.inv_scale)      # correctly rounded float64 reciprocals (IEEE division on the device)
        return self

    def gather_pairs(self, pair_idx, out=None):
        """-> [2*B, F] float32: rows [0,B) 'true', [B,2B) 'modified' (AugmentedDataset.__getitem__)."""
        b = pair_idx.numel()
        if out is None:
            out = torch.empty((2 * b, self.f), dtype=torch.float32, device=self.feats.device)
        _lib.check(_L.idl_gather_pairs(_ptr(self.feats), self.n, self.f, self.n * self.f, _ptr(pair_idx), b,
                                       _ptr(self.mean), _ptr(self.scale), _ptr(out), _stream_ptr()))
        return out


def col_stats(x, out=None):
    """StandardScaler().fit statistics of a device matrix [n, f] (float32 or float64) -> (mean, scale) float64.
    out = (mean, scale): existing float64 [f] device buffers to refit in place."""
    n, f = x.shape
    dev = x.device
    ws = torch.empty(max(_L.idl_col_stats_workspace(n, f), 8), dtype=torch.uint8, device=dev)
    if out is not None:
        mean, scale = out
        assert mean.dtype == scale.dtype == torch.float64 and mean.numel() == scale.numel() == f and mean.is_contiguous() and scale.is_contiguous()
    else:
        mean = torch.empty(f, dtype=torch.float64, device=dev)
        scale = torch.empty(f, dtype=torch.float64, device=dev)
    _lib.check(_L.idl_col_stats(_ptr(x), 1 if x.dtype == torch.float64 else 0, n, f, _ptr(mean), _ptr(scale), _ptr(ws),
                                _stream_ptr()))
    return mean, scale


def standardise(x, mean, scale, out=None):
    """StandardScaler().transform of a device matrix -> float32."""
    n, f = x.shape
    if out is None:
        out = torch.empty((n, f), dtype=torch.float32, device=x.device)
    _lib.check(_L.idl_standardise(_ptr(x), 1 if x.dtype == torch.float64 else 0, n, f, _ptr(mean), _ptr(scale), _ptr(out),
                                  _stream_ptr()))
    return out


def counts_route_ok(k, reduce=False):
    """The predict inputs can be formed from int32 counts (idl_counts_stats / idl_counts_standardise) for plain k-mer rows of 4^k
    columns with 4^k / 4 a multiple of 64: k = 4..7.  IDELUCS_DEV=predict_counts=0: always the float64 rows."""
    return (not reduce) and 4 <= k <= 7 and OPTIONS["predict_counts"] != "0"


def predict_inputs_from_counts(din, k, rows=None, workspace=None):
    """What SequenceDataset feeds the network (reference utils.py:400-405 + models.py:163) without materialising the float64 rows:
    int32 counts of the un-mutated sequences (pseudocount included) -> StandardScaler statistics of counts / sum(counts) in float64
    -> rows [lo, hi) standardised, one rounding to float32.  The same bits as _vectorise(OUT_FREQ_F64) + col_stats + standardise."""
    # workspace (a dict the caller keeps): the int32 counts [n, f] and the float32 result are 16 GB each at 10^6 x k = 6 -- a caller that
    # repeats the call keeps them, instead of depending on what the caching allocator happens to have left of them (a 10^6-row predict
    # took 49 or 175-190 ms by that alone)
    cbuf = None
    if workspace is not None:
        cbuf = workspace.get("counts")
        row = int(_L.idl_row_len(_lib.MODE_KMER, k))
        if cbuf is None or tuple(cbuf.shape) != (1, din.n, row) or cbuf.device != din.codes.device:
            cbuf = workspace["counts"] = torch.empty((1, din.n, row), dtype=torch.int32, device=din.codes.device)
    counts = _vectorise(din, k, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_COUNTS_I32, out=cbuf)[0]
    n, f = counts.shape
    dev = counts.device
    mean = torch.empty(f, dtype=torch.float64, device=dev)
    scale = torch.empty(f, dtype=torch.float64, device=dev)
    totals = torch.empty(n, dtype=torch.int32, device=dev)
    ws = torch.empty(max(_L.idl_counts_stats_workspace(n, f), 8), dtype=torch.uint8, device=dev)
    _lib.check(_L.idl_counts_stats(_ptr(counts), n, f, _ptr(mean), _ptr(scale), _ptr(totals), _ptr(ws), _stream_ptr()))
    lo, hi = (0, n) if rows is None else rows
    out = workspace.get("out") if workspace is not None else None
    if out is None or tuple(out.shape) != (hi - lo, f) or out.device != dev:
        out = torch.empty((hi - lo, f), dtype=torch.float32, device=dev)
        if workspace is not None:
            workspace["out"] = out
    if hi > lo:
        _lib.check(_L.idl_counts_standardise(_ptr(counts[lo:hi]), _ptr(totals[lo:hi]), hi - lo, f, _ptr(mean), _ptr(scale), _ptr(out), _stream_ptr()))
    return out


def ingest_threads():
    """Threads the C++ FASTA reader uses (IDELUCS_THREADS; default min(32, hardware threads, 2 x cgroup CPU quota) / ranks of the node)."""
    return int(_L.idl_ingest_threads())


class _StreamedInput:
    """_DeviceInput filled chunk by chunk: the host reader translates + packs records [lo, hi) into pinned buffers while the
    previous chunk's H2D copy is in flight on a copy stream (ingest overlapped with H2D).  The small arrays
    (lengths, slot offsets) go first, so the device can draw the mimic sites while the bases are still being parsed."""

    def __init__(self, ff, device, n_chunks=None):
        self.n = ff.n
        self.max_len = int(ff.lengths.max()) if ff.n else 0
        self.min_len = int(ff.lengths.min()) if ff.n else 0
        self.total_len = int(ff.lengths.sum()) if ff.n else 0
        self.lengths = torch.from_numpy(ff.lengths).to(device, non_blocking=True)
        self.slot_off = torch.from_numpy(ff.slot_off).to(device, non_blocking=True)
        slots = max(ff.total_slots, 1)
        self._hc = torch.empty(slots * 16, dtype=torch.uint8, pin_memory=True)
        self._hm = torch.empty(slots * 8, dtype=torch.uint8, pin_memory=True)
        self.codes = torch.empty(slots * 16, dtype=torch.uint8, device=device)
        self.mask = torch.empty(slots * 8, dtype=torch.uint8, device=device)
        self._ff, self._copy = ff, torch.cuda.Stream(device=device)
        if n_chunks is None:
            n_chunks = max(1, min(16, ff.total_slots * 24 // (32 << 20)))      # ~32 MB of packed data per chunk
        cuts = np.searchsorted(ff.slot_off, np.linspace(0, ff.total_slots, n_chunks + 1)[1:-1]).tolist()
        self._cuts = [0] + [int(c) for c in cuts] + [ff.n]

    def fill(self):
        ff = self._ff
        for lo, hi in zip(self._cuts[:-1], self._cuts[1:]):
            if hi <= lo:
                continue
            ff.pack_range(lo, hi, self._hc, self._hm)
            a, b = int(ff.slot_off[lo]), int(ff.slot_off[hi])
            with torch.cuda.stream(self._copy):
                self.codes[a * 16:b * 16].copy_(self._hc[a * 16:b * 16], non_blocking=True)
                self.mask[a * 8:b * 8].copy_(self._hm[a * 8:b * 8], non_blocking=True)
        torch.cuda.current_stream().wait_stream(self._copy)
        return self


_ARENAS = {}          # _OnePassInput's pinned / device arenas and staging, kept between calls (release_ingest_buffers() drops them)


def release_ingest_buffers():
    """Give back the arenas the one-pass reader keeps between files (24 bytes per 64 bases of the largest file, host and device)."""
    _ARENAS.clear()
    _L.idl_ingest_release()            # ... and the mapping of the last file read


class _OnePassInput:
    """_DeviceInput produced by ONE pass over the FASTA file (idl_fasta_parse_pack): the reader's threads validate, count and
    2-bit pack their share of the records straight into pinned arenas and send every few MB to the device arenas while they go
    on parsing; records sit back to back inside each thread's region, with gaps between the regions that nothing reads (the
    vectoriser takes a record's first slot from slot_off and its slot count from its length).  create() returns None when the
    file's layout needs the general reader (IDL_FALLBACK)."""

    @classmethod
    def create(cls, sequence_file, device):
        import time
        q = [time.perf_counter()]
        size = os.path.getsize(sequence_file)
        threads = ingest_threads()
        cap = size // 48 + 4096 * threads + 1024                   # slots; a third more than size / 64, + room per region
        # the arenas (24 bytes per slot on each side) are kept between calls: re-allocating them per file lets the caching allocator
        # carve them out of a freed feature-store block, and the next feature store then pays a fresh 6.5 GB allocation (121-262 ms)
        key = (str(device), cap)
        held = _ARENAS.get("key") == key
        if not held:
            _ARENAS.clear()
            _ARENAS.update(key=key, hc=torch.empty(cap * 16, dtype=torch.uint8, pin_memory=True),
                           hm=torch.empty(cap * 8, dtype=torch.uint8, pin_memory=True),
                           codes=torch.empty(cap * 16, dtype=torch.uint8, device=device),
                           mask=torch.empty(cap * 8, dtype=torch.uint8, device=device),
                           small=torch.empty(0, dtype=torch.int64, pin_memory=True))
        hc, hm, codes, mask = _ARENAS["hc"], _ARENAS["hm"], _ARENAS["codes"], _ARENAS["mask"]
        busy = _ARENAS.pop("busy", None)       # the previous file's copies out of the pinned buffers (arenas, staging): done?
        if busy is not None:
            for ev in busy:
                ev.synchronize()
        q.append(time.perf_counter())
        q.append(time.perf_counter())
        copy = torch.cuda.Stream(device=device)
        copy.wait_stream(torch.cuda.current_stream())              # (the previous file's kernels may still be reading the arenas)
        h = ctypes.c_void_p()
        rc = _L.idl_fasta_parse_pack(os.fsencode(sequence_file), _ptr(hc), _ptr(hm), cap, _ptr(codes), _ptr(mask),
                                     ctypes.c_void_p(copy.cuda_stream), ctypes.byref(h))
        if rc == _lib.IDL_FALLBACK:
            copy.synchronize()                                      # (copies of the part that was parsed may be in flight)
            return None
        if rc != _lib.IDL_OK:
            copy.synchronize()
            _lib.check(rc)
        q.append(time.perf_counter())
        self = cls()
        # lengths and first slots FIRST, straight into pinned staging and on to the device (the mimic-site generator needs nothing
        # else); a copy from pageable memory would wait behind the arena copies (16-20 ms).  The names are read out later, while the
        # device works (FastaFile.from_handle(meta=...)).
        n_rec = ctypes.c_int64()
        _lib.check(_L.idl_fasta_sizes(h, ctypes.byref(n_rec), None, None, None))
        n = n_rec.value
        words = 2 * n + 1 + (n + 7) // 8                            # lengths, slots, then one byte per record: its mask was copied
        if _ARENAS["small"].numel() < words:
            _ARENAS["small"] = torch.empty(words + 4096, dtype=torch.int64, pin_memory=True)
        stage = _ARENAS["small"]
        lo_len, hi_len = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(_L.idl_fasta_arena_meta(h, _ptr(stage), ctypes.c_void_p(stage.data_ptr() + 8 * n), ctypes.byref(lo_len), ctypes.byref(hi_len)))
        unsent = int(_L.idl_fasta_arena_mask_flags(h, ctypes.c_void_p(stage.data_ptr() + 8 * (2 * n + 1))))
        both = stage[:words].to(device, non_blocking=True)
        if unsent > 0:       # pieces without an N travelled without their invalid-mask: it is the tail padding the lengths imply
            _lib.check(_L.idl_mask_from_lengths(_ptr(mask), ctypes.c_void_p(both.data_ptr() + 8 * n), _ptr(both),
                                                ctypes.c_void_p(both.data_ptr() + 8 * (2 * n + 1)), n, _stream_ptr()))
        q.append(time.perf_counter())
        meta = stage[:2 * n + 1].numpy().copy()                     # (the staging is overwritten by the next file)
        self.ff = FastaFile.from_handle(h, arena=True, meta=(meta[:n], meta[n:]))
        ff = self.ff
        self.n = n
        self.max_len, self.min_len = hi_len.value, lo_len.value
        self.total_len = ff.total_bases
        self.lengths, self.slot_off = both[:n], both[n:2 * n + 1]
        self.codes, self.mask = codes, mask
        self._hold = (hc, hm, copy)
        ev_small, ev_copy = torch.cuda.Event(), torch.cuda.Event()
        ev_small.record()
        ev_copy.record(copy)
        _ARENAS["busy"] = (ev_small, ev_copy)   # the next file waits for these before it overwrites the pinned buffers
        if os.environ.get("IDELUCS_INGEST_TIMING") is not None:
            q.append(time.perf_counter())
            print("_OnePassInput: pinned arenas %.1f ms, device arenas %.1f, reader %.1f, lengths + slots to the device %.1f, host copy + handle %.1f"
                  % tuple(1e3 * (b - a) for a, b in zip(q[:-1], q[1:])), file=sys.stderr)
        return self

    def fill(self):
        torch.cuda.current_stream().wait_stream(self._hold[2])
        return self


def build_feature_store(sequence_file, n_mimics, k=6, reduce=False, rng=None, seed=0, device=None, fasta=None, streamed=False,
                        reuse=None):
    """Vectorise every mimic view of every sequence in one kernel launch and fit the scaler.
    streamed=True (device RNG only): parse once, pack record chunks into pinned memory and copy each chunk to the device
    while the next is being packed; the mimic sites are drawn meanwhile (they need only the lengths).
    reuse = a FeatureStore the caller is done with: when the new store has its shape (the same file again, another file of the
    same record count) the rows are written into ITS buffers and the scaler is refitted in place, and that object is returned
    -- every address a captured training-step graph holds stays valid (fused.run_epoch's key), so the graph is not captured again."""
    rng = rng or _default_rng_mode()
    dev = _device(device)
    tfs = mimic_transforms(n_mimics)

    def finish(names, lengths, feats, n):
        """Scaler fit + the store: into `reuse` when its buffers were taken (feats is reuse.feats then), else a new one."""
        if reuse is not None and feats is reuse.feats:
            col_stats(feats[0], out=(reuse.mean, reuse.scale))
            reuse._names, reuse.lengths = names, lengths
            return reuse.refresh()
        mean, scale = col_stats(feats[0])
        return FeatureStore(names, lengths, feats, mean, scale, k, reduce)

    def feature_buffer(n, row):
        if (reuse is not None and tuple(reuse.feats.shape) == (len(tfs), n, row) and reuse.feats.device == dev
                and reuse.k == k and reuse.reduce == reduce and reuse.feats.is_contiguous()):
            return reuse.feats
        return torch.empty((len(tfs), n, row), dtype=torch.float32, device=dev)
    if streamed and rng == "philox" and fasta is None and OPTIONS["one_pass"] != "0":
        import time
        timing = os.environ.get("IDELUCS_INGEST_TIMING") is not None
        q = [time.perf_counter()]
        din = _OnePassInput.create(sequence_file, dev)
        if din is not None:
            q.append(time.perf_counter())
            # the feature buffer first: it is the one allocation that must find the block a previous store left in the allocator's
            # cache whole (6.5 GB at cfg2: a fresh one costs 15-260 ms), before the smaller requests below can be carved out of it
            mode = _lib.MODE_CANONICAL if reduce else _lib.MODE_KMER
            row = _L.idl_row_len(mode, k)
            feats = feature_buffer(din.n, row)
            q.append(time.perf_counter())
            # the sites into their slots WITHOUT waiting for the overflow flag: that wait held the host for the generator's 0.5 ms
            # with the vectoriser not yet enqueued.  The flag travels to pinned memory and is read when everything is queued (an
            # item beyond its slot: expected sites + 10 sigma + 32 -- never seen); then the exact two-pass protocol redoes the build.
            # Nothing here waits for the vectoriser: the caller's next launches queue behind it
            edits, edit_off = _philox_edits(din, [t.spec() for t in tfs], seed, sync=False)
            flag = getattr(edits, "overflow_flag", None)
            if flag is not None:               # on its way to pinned memory behind the generator; looked at below, when it has long arrived
                if "flag" not in _ARENAS:
                    _ARENAS["flag"] = torch.zeros(1, dtype=torch.int32, pin_memory=True)
                _ARENAS["flag"].copy_(flag.reshape(1), non_blocking=True)
                flag_ready = torch.cuda.Event()
                flag_ready.record()
            din.fill()
            q.append(time.perf_counter())
            _vectorise(din, k, mode, _lib.INIT_ONE, _lib.OUT_FREQ_F32, len(tfs), edits, edit_off, feats)
            store = finish(din.ff, din.ff.lengths, feats, din.n)
            # (the names stay in the open handle until somebody asks for them -- FeatureStore.names, FastaFile.names -- or the handle
            #  goes: reading them out here held the host for ~1 ms right when the caller wants to queue its next launches behind the
            #  vectoriser; the handle keeps the file's mapping alive, which the reader keeps anyway)
            if flag is not None:
                flag_ready.synchronize()       # (waits for the site generator, not for the vectoriser behind it)
            if flag is not None and int(_ARENAS["flag"][0]) != 0:
                edits, edit_off = _philox_edits(din, [t.spec() for t in tfs], seed, slots=False)
                _vectorise(din, k, mode, _lib.INIT_ONE, _lib.OUT_FREQ_F32, len(tfs), edits, edit_off, feats)
                col_stats(feats[0], out=(store.mean, store.scale))
                store.refresh()
            if timing:
                q.append(time.perf_counter())
                torch.cuda.synchronize()
                q.append(time.perf_counter())
                print("build_feature_store (one pass): reader + arenas %.1f ms, feature buffer %.1f, mimic sites %.1f, enqueue %.1f, drain %.1f"
                      % tuple(1e3 * (b - a) for a, b in zip(q[:-1], q[1:])), file=sys.stderr)
            return store
    if streamed and rng == "philox" and fasta is None:
        ff = FastaFile(sequence_file, check=True, pack="deferred")
        try:
            din = _StreamedInput(ff, dev)
            edits, edit_off = _philox_edits(din, [t.spec() for t in tfs], seed)     # overlaps with the host packing below
            din.fill()
        finally:
            ff.close()
        mode = _lib.MODE_CANONICAL if reduce else _lib.MODE_KMER
        feats = _vectorise(din, k, mode, _lib.INIT_ONE, _lib.OUT_FREQ_F32, len(tfs), edits, edit_off,
                           feature_buffer(din.n, _L.idl_row_len(mode, k)))
        return finish(ff, ff.lengths, feats, din.n)
    ff = fasta if fasta is not None else FastaFile(sequence_file, check=True, keep_bytes=(rng == "compat"))
    mode = _lib.MODE_CANONICAL if reduce else _lib.MODE_KMER
    if rng == "compat":
        if ff.bytes is None:
            raise ValueError("compat RNG needs FastaFile(keep_bytes=True)")
        e, eo = _compat_edits(ff, tfs)
        edits = torch.from_numpy(e.view(np.int32)).to(dev) if e.size else torch.zeros(1, dtype=torch.int32, device=dev)
        edit_off = torch.from_numpy(eo).to(dev)
        din = _DeviceInput(ff, dev)
    elif rng == "philox":
        din = _DeviceInput(ff, dev)
        edits, edit_off = _philox_edits(din, [t.spec() for t in tfs], seed)
    else:
        raise ValueError("rng must be 'compat' or 'philox'")
    feats = _vectorise(din, k, mode, _lib.INIT_ONE, _lib.OUT_FREQ_F32, len(tfs), edits, edit_off,
                       feature_buffer(din.n, _L.idl_row_len(mode, k)))
    return finish(ff.names, ff.lengths, feats, din.n)


def AugmentFasta(sequence_file, n_mimics, k=6, reduce=False, rng=None, seed=0):
    """Reference idelucs/utils.py:321-368 -> float32 [N*max(n_mimics,2), 2, F] (host array), pairs
    mimic-major, both halves standardised with the scaler fitted on the "true" view."""
    st = build_feature_store(sequence_file, n_mimics, k=k, reduce=reduce, rng=rng, seed=seed)
    scaled = standardise(st.feats.view(-1, st.f), st.mean, st.scale).view(st.n_views, st.n, st.f).cpu().numpy()
    x = np.empty((st.n_pairs, 2, st.f), np.float32)
    for m in range(st.n_views - 1):
        x[m * st.n:(m + 1) * st.n, 0, :] = scaled[0]
        x[m * st.n:(m + 1) * st.n, 1, :] = scaled[m + 1]
    return x


def kernel_file(k):
    """kernels/kernel{k}.npz of the reference repository (data files, shipped in idelucs_amd/kernels/)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernels", f"kernel{int(k)}.npz")
    if not os.path.exists(path):
        raise ValueError(f"no projection kernel for k={k} (the reference ships kernel4/5/6.npz)")
    return path


def project_kernel(features, kernel):
    """The compositional projection the reference keeps commented out in kmersFasta (utils.py:272-275):
    `np.dot(kmers, KERNEL)` with KERNEL = kernels/kernel{k}.npz['arr_0'] ([4^k, d], d = 135/511/2079).
    features: [N, 4^k] tensor or array -> [N, d] on the device, one GEMM (float64 like the reference's rows)."""
    dev = _device()
    x = torch.as_tensor(features).to(dev).double()
    kern = torch.as_tensor(np.load(kernel)["arr_0"] if isinstance(kernel, (str, os.PathLike)) else kernel).to(dev).double()
    if x.shape[1] != kern.shape[0]:
        raise ValueError(f"kernel has {kern.shape[0]} rows, features have {x.shape[1]} columns")
    return x @ kern


class AugmentedDataset(torch.utils.data.Dataset):
    """Reference idelucs/utils.py:370-389."""

    def __init__(self, data):
        self.data = data

    def __len__(self):
        return self.data.shape[0]

    def __getitem__(self, idx):
        if torch.is_tensor(idx):
            idx = idx.tolist()
        return {"true": self.data[idx, 0, :], "modified": self.data[idx, 1, :]}


def predict_features(sequence_file, k=6, reduce=False, device=None, fasta=None, rows=None, with_names=True):
    """What SequenceDataset feeds the network (utils.py:400-405 + models.py:163): un-mutated float64
    frequencies, StandardScaler fit_transform in float64, rounded once to float32.  -> (names, lengths, [N,F] f32).
    rows=(lo, hi): the scaler is still fitted on ALL rows, only rows [lo, hi) are standardised and returned (sharded predict:
    every rank recomputes the cheap statistics locally -- bit-identical everywhere, no collective -- and embeds its shard)."""
    dev = _device(device)
    din = None
    if fasta is None and OPTIONS["one_pass"] != "0":
        din = _OnePassInput.create(sequence_file, dev)             # one pass over the file, copies in flight while parsing
        if din is not None:
            ff = din.ff
            din.fill()
    if din is None:
        ff = fasta if fasta is not None else FastaFile(sequence_file, check=True)
        din = _DeviceInput(ff, dev)
    mode = _lib.MODE_CANONICAL if reduce else _lib.MODE_KMER
    if din.n > 0 and counts_route_ok(k, reduce):                  # (round 4) straight from the integer counts: the same bits, 1.6 GB instead of 3.3 x 3
        out = predict_inputs_from_counts(din, k, rows)
    else:
        f64 = _vectorise(din, k, mode, _lib.INIT_ONE, _lib.OUT_FREQ_F64)[0]
        mean, scale = col_stats(f64)
        if rows is not None:
            f64 = f64[rows[0]:rows[1]]
        out = standardise(f64, mean, scale)
    if fasta is None:
        ff.close()
    return (ff.names if with_names else None), ff.lengths, out          # (with_names=False: the names are not decoded)


class SequenceDataset(torch.utils.data.Dataset):
    """Reference idelucs/utils.py:391-420: un-mutated k-mer vectors, own scaler (float64)."""

    def __init__(self, fasta_file, k=6, transform=None, GT_file=None, reduce=False):
        self.names, self.lengths, self.GT, self.cluster_dis = SummaryFasta(fasta_file, GT_file)
        mode = _lib.MODE_CANONICAL if reduce else _lib.MODE_KMER
        _, f64 = _features_one_pass(fasta_file, k, transform, mode, True, _lib.OUT_FREQ_F64)
        mean, scale = col_stats(f64)
        # StandardScaler on float64 keeps float64 (utils.py:404-405): (x - mean) / scale
        self.kmers = ((f64 - mean) / scale).cpu().numpy()

    def __len__(self):
        return len(self.lengths)

    def __getitem__(self, idx):
        if torch.is_tensor(idx):
            idx = idx.tolist()
        if self.GT:
            return {"kmer": self.kmers[idx, :], "name": self.names[idx], "cluster_id": self.GT[idx]}
        return {"kmer": self.kmers[idx, :], "name": self.names[idx]}


class DeviceBatchLoader:
    """Stands in for the reference's DataLoader(AugmentedDataset, shuffle=True) (utils.py:422-429):
    iterating yields {'true': [b,F], 'modified': [b,F]} float32 device tensors for a fresh random
    permutation of the N*n_mimics pairs; the last batch is partial (no drop_last).  Batches are
    assembled by idl_gather_pairs from the HBM-resident store -- no worker processes, no host copy."""

    def __init__(self, store, batch_size, generator=None):
        self.store, self.batch_size, self.generator = store, batch_size, generator

    def __len__(self):
        return (self.store.n_pairs + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        perm = torch.randperm(self.store.n_pairs, device=self.store.feats.device, generator=self.generator)
        for i in range(0, self.store.n_pairs, self.batch_size):
            idx = perm[i:i + self.batch_size]
            y = self.store.gather_pairs(idx)
            b = idx.numel()
            yield {"true": y[:b], "modified": y[b:]}


def create_dataloader(sequence_file, n_mimics, k=6, batch_size=512, GT_file=None, reduce=False, rng=None, seed=0):
    """Reference idelucs/utils.py:422-429."""
    return DeviceBatchLoader(build_feature_store(sequence_file, n_mimics, k=k, reduce=reduce, rng=rng, seed=seed), batch_size)


# ------------------------------------------------------------------------------------------------
# post-hoc helpers kept for API compatibility (run once on [N] ints; sklearn/scipy on the host, as in the reference)
# ------------------------------------------------------------------------------------------------
def cluster_acc(y_true, y_pred):
    """Reference idelucs/utils.py:489-508: best one-to-one label matching (Hungarian) accuracy."""
    from scipy.optimize import linear_sum_assignment
    y_true = np.asarray(y_true).astype(np.int64)
    y_pred = np.asarray(y_pred).astype(np.int64)
    d = max(y_pred.max(), y_true.max()) + 1
    w = np.zeros((d, d), dtype=np.int64)
    np.add.at(w, (y_pred, y_true), 1)
    ind = linear_sum_assignment(w.max() - w)
    ind = np.transpose(np.asarray(ind))
    return ind, sum(w[i, j] for i, j in ind) * 1.0 / y_pred.size
