"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY (see oracle/idelucs_oracle.c header).

CPU restatement of the reference hot path (Kari-Genomics-Lab/iDeLUCS @ 2024_08_07),
used as the checker for the HIP path.  Nothing under idelucs_amd/ may import it.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.

Parity status: PINNED by tests/golden/ (generated from the imported reference by
tests/golden/make_golden.py) -- see tests/test_oracle_golden.py.

Integer/byte arithmetic lives in idelucs_oracle.c (ctypes); the parts of the
reference that are numpy/Python by nature (host RNG draws, the FASTA line state
machine, StandardScaler arithmetic) are restated here in numpy.
Reference citations are `file:line` relative to the reference repo root.
"""
import ctypes
import os
import random
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile liboracle.so (gcc) if missing or stale."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "idelucs_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "liboracle.so"], check=True, stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        u8p, i32p, u32p = (ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_int32),
                           ctypes.POINTER(ctypes.c_uint32))
        L.orc_kmer_counts.argtypes = [u8p, ctypes.c_int64, ctypes.c_int, i32p]
        L.orc_kmer_counts.restype = ctypes.c_int
        L.orc_cgr.argtypes = [u8p, ctypes.c_int64, ctypes.c_int, i32p]
        L.orc_cgr.restype = ctypes.c_int
        L.orc_reverse_complement.argtypes = [ctypes.c_uint32, ctypes.c_int]
        L.orc_reverse_complement.restype = ctypes.c_uint32
        L.orc_kmer_rev_comp.argtypes = [i32p, ctypes.c_int, i32p]
        L.orc_kmer_rev_comp.restype = ctypes.c_int
        L.orc_check_sequence.argtypes = [u8p, ctypes.c_int64, u8p]
        L.orc_check_sequence.restype = ctypes.c_int64
        L.orc_normalise_f64.argtypes = [i32p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
        L.orc_normalise_f64.restype = None
        L.orc_apply_edits.argtypes = [u8p, ctypes.c_int64, u32p, ctypes.c_int64]
        L.orc_apply_edits.restype = None
        L.orc_pack.argtypes = [u8p, ctypes.c_int64, u8p, u8p]
        L.orc_pack.restype = None
        L.orc_mimic_sites.argtypes = [ctypes.c_int64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_double, ctypes.c_double,
                                      ctypes.c_uint64, u32p, ctypes.c_int64]
        L.orc_mimic_sites.restype = ctypes.c_int64
        L.orc_mimic_random_n.argtypes = [ctypes.c_int64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_uint64, u32p]
        L.orc_mimic_random_n.restype = ctypes.c_int64
        _LIB = L
    return _LIB


def _u8(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def _i32(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def _seq_array(seq):
    a = np.frombuffer(bytes(seq), dtype=np.uint8) if not isinstance(seq, np.ndarray) else seq
    return np.ascontiguousarray(a, dtype=np.uint8)


# ------------------------------------------------------------------ native counters
def kmer_counts(seq, k, counts):
    """idelucs/kmers.pyx:2-50 -- accumulates into the int32 array `counts` in place."""
    assert counts.dtype == np.int32 and counts.flags.c_contiguous and counts.size >= 4 ** k
    a = _seq_array(seq)
    if lib().orc_kmer_counts(_u8(a), a.size, k, _i32(counts)) != 0:
        raise ValueError("bad k")


def cgr(seq, k, out):
    """idelucs/kmers.pyx:53-123."""
    assert out.dtype == np.int32 and out.flags.c_contiguous and out.size >= 4 ** k
    a = _seq_array(seq)
    if lib().orc_cgr(_u8(a), a.size, k, _i32(out)) != 0:
        raise ValueError("bad k")


def reverse_complement(x, k):
    """idelucs/utils.py:191-206."""
    return int(lib().orc_reverse_complement(int(x), int(k)))


def kmer_rev_comp(counts, k):
    """idelucs/utils.py:208-221 (modifies `counts` in place like the reference, returns canonical part)."""
    assert counts.dtype == np.int32 and counts.flags.c_contiguous
    out = np.empty(4 ** k, np.int32)
    m = lib().orc_kmer_rev_comp(_i32(counts), k, _i32(out))
    return out[:m].copy()


def n_canonical(k):
    """idelucs/models.py:61-62."""
    return (4 ** k + 4 ** (k // 2)) // 2 if k % 2 == 0 else (4 ** k) // 2


# ------------------------------------------------------------------ FASTA layer
def check_sequence(header, seq):
    """idelucs/utils.py:26-51 (header checks :37-40, translate :42-45, error :46-50)."""
    if len(header) > 0 and (header[0] in (">", "#") or header[0].isspace()):
        raise ValueError("Bad character in sequence header")
    if "\t" in header:
        raise ValueError("tab included in header")
    a = _seq_array(seq)
    out = np.empty(max(a.size, 1), np.uint8)
    n = lib().orc_check_sequence(_u8(a), a.size, _u8(out))
    if n < 0:
        # the reference reports the first byte of the TRANSLATED string that is not ACGTN
        bad = bytes(a[-n - 1:-n]).translate(bytearray.maketrans(b"acgtuUswkmyrbdhvnSWKMYRBDHV-",
                                                                b"ACGTTTNNNNNNNNNNNNNNNNNNNNNN"))
        raise ValueError("Invalid DNA byte in sequence {}: '{}'".format(header, chr(bad[0])))
    return bytearray(out[:n].tobytes())


def fasta_records(fname, check=True):
    """The record state machine shared by SummaryFasta / kmersFasta / cgrFasta
    (idelucs/utils.py:145-186, :229-268, :284-316; survey appendix D.1):
    '#' lines ignored; '>' flushes the current record if an id exists, id = line[1:-1];
    other lines are .strip()ped and joined; unconditional flush at EOF."""
    lines, seq_id = [], ""
    with open(fname, "rb") as fh:
        for line in fh:
            if line.startswith(b"#"):
                continue
            if line.startswith(b">"):
                if seq_id != "":
                    seq = bytearray().join(lines)
                    yield seq_id, (check_sequence(seq_id, seq) if check else seq)
                    lines = []
                seq_id = line[1:-1].decode()
            else:
                lines.append(line.strip())
    seq = bytearray().join(lines)
    yield seq_id, (check_sequence(seq_id, seq) if check else seq)


def SummaryFasta(fname, GT_file=None):
    """idelucs/utils.py:137-188."""
    gt_dict, cluster_dis, ground_truth = None, None, None
    if GT_file:
        import pandas as pd
        df = pd.read_csv(GT_file, sep="\t")
        gt_dict = dict(zip(df.sequence_id, df.cluster_id))
        cluster_dis = df["cluster_id"].value_counts().to_dict()
        ground_truth = []
    names, lengths = [], []
    for seq_id, seq in fasta_records(fname, check=False):
        if GT_file and seq_id not in gt_dict:
            raise ValueError("Check GT for sequence {}".format(seq_id))
        seq = check_sequence(seq_id, seq)
        names.append(seq_id)
        lengths.append(len(seq))
        if GT_file:
            ground_truth.append(gt_dict[seq_id])
    return names, lengths, ground_truth, cluster_dis


def kmersFasta(fname, k=6, transform=None, reduce=False):
    """idelucs/utils.py:224-277: per record ones-initialised counts, optional collapse, / sum (float64)."""
    names, rows = [], []
    for seq_id, seq in fasta_records(fname):
        names.append(seq_id)
        if transform:
            transform(seq)
        counts = np.ones(4 ** k, np.int32)
        kmer_counts(seq, k, counts)
        if reduce:
            counts = kmer_rev_comp(counts, k)
        rows.append(counts / np.sum(counts))
    return names, np.array(rows)


def cgrFasta(fname, k=6, transform=None):
    """idelucs/utils.py:279-317 -- NB no check_sequence (lower-case bytes are skipped by the counter)."""
    names, rows = [], []
    for seq_id, seq in fasta_records(fname, check=False):
        names.append(seq_id)
        if transform:
            transform(seq)
        counts = np.ones(4 ** k, np.int32)
        cgr(seq, k, counts)
        rows.append(counts / np.sum(counts))
    return names, np.array(rows)


# ------------------------------------------------------------------ mimic transforms (host RNG)
_A, _C, _G, _T, _N = (ord(c) for c in "ACGTN")


class transition:
    """idelucs/utils.py:54-76: sites = where(np.random.random(L) < p); A<->G, C<->T, N->N."""

    def __init__(self, threshold):
        self.threshold = threshold

    def __call__(self, seq):
        x = np.random.random(len(seq))
        swap = {_A: _G, _G: _A, _T: _C, _C: _T, _N: _N}
        for i in np.where(x < self.threshold)[0]:
            seq[i] = swap[seq[i]]


class transversion:
    """idelucs/utils.py:98-118: same site draw; purine -> random.choice([T, C]), pyrimidine ->
    random.choice([A, G]); an N site still calls random.choice([N]) (consumes Python RNG)."""

    def __init__(self, threshold):
        self.threshold = threshold

    def __call__(self, seq):
        x = np.random.random(len(seq))
        table = {_A: [_T, _C], _G: [_T, _C], _T: [_A, _G], _C: [_A, _G], _N: [_N]}
        for i in np.where(x < self.threshold)[0]:
            seq[i] = random.choice(table.get(seq[i], [_N]))


class transition_transversion:
    """idelucs/utils.py:120-135: transition pass, then transversion pass, on the same buffer."""

    def __init__(self, threshold_1, threshold_2):
        self.t1, self.t2 = transition(threshold_1), transversion(threshold_2)

    def __call__(self, seq):
        self.t1(seq)
        self.t2(seq)


class Random_N:
    """idelucs/utils.py:78-95: np.random.randint(0, L, n) positions (with replacement) -> 'N'."""

    def __init__(self, n_bp):
        self.n_bp = n_bp

    def __call__(self, seq):
        for i in np.random.randint(0, len(seq), self.n_bp):
            seq[i] = _N


# ------------------------------------------------------------------ scaler + AugmentFasta
def scaler_fit(X):
    """sklearn StandardScaler().fit as used at idelucs/utils.py:357-359 (sklearn 1.7 behaviour,
    survey appendix C): float64 statistics of X (float32 or float64), population variance,
    scale = sqrt(var) with (scale < 10*eps) -> 1."""
    X64 = np.asarray(X, dtype=np.float64)
    n = X64.shape[0]
    s = X64.sum(axis=0)
    mean = s / n
    t = X64 - mean
    corr = t.sum(axis=0)
    var = ((t * t).sum(axis=0) - corr * corr / n) / n
    scale = np.sqrt(var)
    scale[scale < 10 * np.finfo(np.float64).eps] = 1.0
    return mean, scale


def scaler_transform(X, mean, scale):
    """StandardScaler.transform on a copy: X -= mean; X /= scale, each evaluated in float64 and
    rounded to X.dtype (idelucs/utils.py:361-366 for float32, :404-405 for float64)."""
    Y = np.array(X, copy=True)
    np.subtract(Y, mean, out=Y, casting="same_kind")
    np.divide(Y, scale, out=Y, casting="same_kind")
    return Y


def AugmentFasta(sequence_file, n_mimics, k=6, reduce=False):
    """idelucs/utils.py:321-368: pass0 t_t(1e-2, .5e-2) -> t_norm; pass1 transition(1e-2);
    pass2 transversion(.5e-2); passes 3.. Random_N(20) x (n_mimics-2); pairs (t_norm, mimic_m)
    mimic-major; float32; StandardScaler fit on float32 t_norm, applied to both halves."""
    _, t_norm = kmersFasta(sequence_file, k, transition_transversion(1e-2, 0.5e-2), reduce)
    mimics = [kmersFasta(sequence_file, k, transition(1e-2), reduce)[1],
              kmersFasta(sequence_file, k, transversion(0.5e-2), reduce)[1]]
    for _ in range(n_mimics - 2):
        mimics.append(kmersFasta(sequence_file, k, Random_N(20), reduce)[1])
    n, f = t_norm.shape
    x = np.empty((len(mimics) * n, 2, f), np.float32)
    for m, mim in enumerate(mimics):
        x[m * n:(m + 1) * n, 0, :] = t_norm
        x[m * n:(m + 1) * n, 1, :] = mim
    mean, scale = scaler_fit(t_norm.astype(np.float32))
    x[:, 0, :] = scaler_transform(x[:, 0, :], mean, scale)
    x[:, 1, :] = scaler_transform(x[:, 1, :], mean, scale)
    return x


def sequence_dataset_features(fasta_file, k=6, reduce=False):
    """idelucs/utils.py:400-405 (SequenceDataset): un-mutated float64 rows, fit_transform in float64."""
    _, kmers = kmersFasta(fasta_file, k, None, reduce)
    mean, scale = scaler_fit(kmers)
    return scaler_transform(kmers, mean, scale)


# ------------------------------------------------------------------ device-convention helpers
def apply_edits(seq, edits):
    """Apply this repo's substitution edits (pos | op<<30) to a cleaned byte string (C oracle)."""
    a = np.array(np.frombuffer(bytes(seq), np.uint8), copy=True)
    e = np.ascontiguousarray(edits, dtype=np.uint32)
    lib().orc_apply_edits(_u8(a), a.size, e.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), e.size)
    return bytearray(a.tobytes())


def pack(seq):
    """2-bit codes + invalid mask in 64-base slots (C oracle's statement of the device layout)."""
    a = _seq_array(seq)
    slots = (a.size + 63) // 64
    codes = np.zeros(slots * 16, np.uint8)
    mask = np.zeros(slots * 8, np.uint8)
    if slots:
        lib().orc_pack(_u8(a), a.size, _u8(codes), _u8(mask))
    return codes, mask


def mimic_edits(length, seq_index, view, spec, seed):
    """Fast-mode (Philox) mimic edits of one (view, sequence): spec = (p_transition, p_transversion, n_random_n).
    C-oracle restatement of the device generator's spec (idelucs_amd/csrc/mimic.hip)."""
    p_ts, p_tv, n_rand = spec
    u32p = ctypes.POINTER(ctypes.c_uint32)
    if n_rand > 0:
        out = np.empty(max(n_rand, 1), np.uint32)
        n = lib().orc_mimic_random_n(int(length), int(seq_index), int(view), int(n_rand), int(seed), out.ctypes.data_as(u32p))
        return out[:n].copy()
    cap = int(length) + 64
    out = np.empty(cap, np.uint32)
    n = lib().orc_mimic_sites(int(length), int(seq_index), int(view), float(p_ts), float(p_tv), int(seed), out.ctypes.data_as(u32p), cap)
    return out[:n].copy()
