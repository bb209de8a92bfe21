"""Core distances of the fine-grained mode's HDBSCAN (reference __main__.py:83,153-156) by the one-pass window kernels
(csrc/knn.hip: idl_knn_window + idl_knn_select) against the float64 matrix path and against the definition itself."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _blobs(n, seed, noise=20, spread=(0.3, 0.9), offset=0.0):
    rng = np.random.default_rng(seed)
    centres = rng.normal(size=(7, 64)) * 2.5 + offset
    truth = rng.integers(0, 7, n)
    x = centres[truth] + rng.normal(size=(n, 64)) * rng.uniform(*spread, size=(7,))[truth][:, None]
    if noise:
        x[: n // noise] = rng.uniform(-8, 8, size=(n // noise, 64)) + offset
    return x.astype(np.float32).astype(np.float64)


def _kth_by_definition(x, rows, k):
    """sqrt of the k-th smallest of sum_c (a_c - b_c)^2, the sum taken coordinate by coordinate in float64 (sklearn's
    EuclideanDistance: _dist_metrics.pyx.tp rdist loop), the row itself included."""
    out = np.empty(len(rows))
    for i, r in enumerate(rows):
        acc = np.zeros(len(x))
        for c in range(x.shape[1]):
            t = x[r, c] - x[:, c]
            acc += t * t
        out[i] = np.sqrt(np.partition(acc, k - 1)[k - 1])
    return out


@pytest.mark.parametrize("n,sample,offset", [(40000, None, 0.0), (9000, 1024, 0.0), (20000, 4096, 37.5)])
def test_window_core_distances_are_the_kth_neighbour_distances(n, sample, offset, monkeypatch):
    """Every row's value equals the definition bit for bit on a sample of rows, and the whole vector equals the matrix path's
    (which finds the neighbour by a float64 Gram form and then takes its distance from the difference vector).  The third case
    sits far from the origin (norms 10^5: without the centring the rounding bound would swallow the bracket); the second has a
    sample so small that the lower rank of the bracket does not exist (nothing counted below, everything under `hi` kept)."""
    import torch
    from idelucs_amd import posthoc
    x = _blobs(n, seed=n, offset=offset)
    k = n // 100 + 1
    dev = torch.device("cuda")
    xd = torch.from_numpy(x).to(dev)
    if sample:
        monkeypatch.setattr(posthoc, "KNN_SAMPLE", sample)
    stats = {}
    monkeypatch.setitem(__import__("idelucs_amd.posthoc", fromlist=["OPTIONS"]).OPTIONS, "knn", "window")
    got = posthoc.core_distances_device(xd, k, dev, stats=stats).cpu().numpy()
    monkeypatch.setitem(__import__("idelucs_amd.posthoc", fromlist=["OPTIONS"]).OPTIONS, "knn", "matrix")
    ref = posthoc.core_distances_device(xd, k, dev).cpu().numpy()
    print(stats)
    assert stats["missed"] <= max(8, n // 2000), "the bracket misses far more rows than its 4.5 sigma promise"
    rows = np.random.default_rng(1).choice(n, 48, replace=False)
    assert np.array_equal(got[rows], _kth_by_definition(x, rows, k))
    assert np.array_equal(got, ref)


def test_window_core_distances_with_duplicates_and_tiny_k(monkeypatch):
    """Half of the points are exact copies of a few others: whole runs of equal distances at the k-th rank (bands that overflow
    or touch the bracket's edge are handed to the matrix path), and a k of 2 (the HDBSCAN floor)."""
    import torch
    from idelucs_amd import posthoc
    x = _blobs(12000, seed=3, noise=0)
    x[6000:] = x[np.random.default_rng(0).integers(0, 40, 6000)]
    dev = torch.device("cuda")
    xd = torch.from_numpy(x).to(dev)
    for k in (121, 2):
        stats = {}
        monkeypatch.setitem(__import__("idelucs_amd.posthoc", fromlist=["OPTIONS"]).OPTIONS, "knn", "window")
        got = posthoc.core_distances_device(xd, k, dev, stats=stats).cpu().numpy()
        monkeypatch.setitem(__import__("idelucs_amd.posthoc", fromlist=["OPTIONS"]).OPTIONS, "knn", "matrix")
        ref = posthoc.core_distances_device(xd, k, dev).cpu().numpy()
        print(k, stats)
        assert np.array_equal(got, ref)
        rows = np.r_[0:8, 6000:6008]
        assert np.array_equal(got[rows], _kth_by_definition(x, rows, k))


def test_hdbscan_labels_do_not_depend_on_the_core_distance_path(monkeypatch):
    from idelucs_amd import posthoc
    x = _blobs(36000, seed=8)
    k = 36000 // 100 + 1
    monkeypatch.setitem(__import__("idelucs_amd.posthoc", fromlist=["OPTIONS"]).OPTIONS, "knn", "window")
    l1, p1 = posthoc.hdbscan_device(x, k)
    monkeypatch.setitem(__import__("idelucs_amd.posthoc", fromlist=["OPTIONS"]).OPTIONS, "knn", "matrix")
    l2, p2 = posthoc.hdbscan_device(x, k)
    assert np.array_equal(l1, l2) and np.array_equal(p1, p2)


@pytest.mark.parametrize("d,as_f32", [(64, True), (16, True), (64, False), (40, False)])
def test_prim_with_the_8_bit_filter_builds_the_same_tree(monkeypatch, d, as_f32):
    """idl_mst_prim_local skips exact distances that an 8-bit lower bound proves irrelevant: the edges -- nodes, order, float64
    weights -- are those of the unfiltered scan, also for data with outliers far outside the code range, for fewer than 64
    features and for float64 coordinates that float32 does not hold (the generic kernels)."""
    from idelucs_amd import posthoc
    rng = np.random.default_rng(4 + d)
    n = 24000
    centres = rng.normal(size=(7, d)) * 2.5
    truth = rng.integers(0, 7, n)
    x = centres[truth] + rng.normal(size=(n, d)) * rng.uniform(0.3, 0.9, size=(7,))[truth][:, None]
    x[: n // 20] = rng.uniform(-8, 8, size=(n // 20, d))
    x[:40] *= 30.0                                              # far outside the boxes of their groups
    if as_f32:
        x = x.astype(np.float32).astype(np.float64)
    k = 241
    with_filter, without = {}, {}
    l1, p1 = posthoc.hdbscan_device(x, k, stats=with_filter)
    monkeypatch.setitem(__import__("idelucs_amd.posthoc", fromlist=["OPTIONS"]).OPTIONS, "mst_filter", "0")
    l2, p2 = posthoc.hdbscan_device(x, k, stats=without)
    a, b = with_filter["mst_edges"], without["mst_edges"]
    for f in ("current_node", "next_node", "distance"):
        assert np.array_equal(a[f], b[f]), f
    assert np.array_equal(l1, l2) and np.array_equal(p1, p2)


def test_filtered_prim_is_sklearns(monkeypatch):
    from sklearn.cluster import HDBSCAN
    from sklearn.metrics import adjusted_rand_score
    from idelucs_amd import posthoc
    monkeypatch.setattr(posthoc, "MST_FILTER_MIN", 0)
    x = _blobs(6000, seed=12)
    ref = HDBSCAN(min_cluster_size=61).fit(x)
    labels, prob = posthoc.hdbscan_device(x, 61)
    assert np.array_equal(labels < 0, ref.labels_ < 0) and adjusted_rand_score(ref.labels_, labels) == 1.0
    assert np.allclose(prob, ref.probabilities_, atol=1e-9)


@pytest.mark.parametrize("kind", ["blobs", "tight", "duplicates", "uniform"])
def test_lazy_prim_builds_the_same_tree(monkeypatch, kind):
    """idl_mst_prim_lazy lets groups of points sleep while the tree grows elsewhere and has them catch up when the weight being
    added reaches their bound: the edges -- nodes, order, float64 weights -- are those of the scan that visits every point at
    every step; on Gaussian blobs, on tight far-apart clusters with background noise (stalls at every change of cluster), with
    half of the points exact copies of others (runs of equal weights: ties go by the original number) and on structureless data
    (nobody can sleep)."""
    from idelucs_amd import posthoc
    rng = np.random.default_rng(9)
    n = 70000
    if kind == "uniform":
        x = rng.uniform(-1, 1, size=(n, 64))
    else:
        tight = kind == "tight"
        centres = rng.normal(size=(6, 64)) * (30.0 if tight else 2.5)
        truth = rng.integers(0, 6, n)
        x = centres[truth] + rng.normal(size=(n, 64)) * (0.05 if tight else 0.6)
        x[: n // 50] = rng.uniform(-60, 60, size=(n // 50, 64)) if tight else rng.uniform(-8, 8, size=(n // 50, 64))
        if kind == "duplicates":
            x[n // 2:] = x[rng.integers(0, n // 2, n - n // 2)]
    x = x.astype(np.float32).astype(np.float64)
    k = n // 100 + 1
    lazy, plain = {}, {}
    monkeypatch.setitem(__import__("idelucs_amd.posthoc", fromlist=["OPTIONS"]).OPTIONS, "mst", "lazy")
    l1, p1 = posthoc.hdbscan_device(x, k, stats=lazy)
    assert "prim_stalls" in lazy, "the lazy path did not run"
    print({kk: lazy[kk] for kk in ("prim_launches", "prim_stalls", "prim_censuses", "prim_s")})
    monkeypatch.setitem(__import__("idelucs_amd.posthoc", fromlist=["OPTIONS"]).OPTIONS, "mst", "local")
    l2, p2 = posthoc.hdbscan_device(x, k, stats=plain)
    a, b = lazy["mst_edges"], plain["mst_edges"]
    for f in ("current_node", "next_node", "distance"):
        assert np.array_equal(a[f], b[f]), f
    assert np.array_equal(l1, l2) and np.array_equal(p1, p2)


def test_core_distances_when_no_bracket_exists(monkeypatch):
    """k close to n (no room for the upper rank of the bracket in the sample): the window path declines and the float64 matrix
    path answers; forced window mode says so."""
    import torch
    from idelucs_amd import posthoc
    x = _blobs(5003, seed=21)
    dev = torch.device("cuda")
    xd = torch.from_numpy(x).to(dev)
    k = 4990
    monkeypatch.setattr(posthoc, "KNN_WINDOW_MIN", 0)
    got = posthoc.core_distances_device(xd, k, dev).cpu().numpy()
    rows = np.arange(0, 5003, 500)
    assert np.array_equal(got[rows], _kth_by_definition(x, rows, k))
    monkeypatch.setitem(__import__("idelucs_amd.posthoc", fromlist=["OPTIONS"]).OPTIONS, "knn", "window")
    with pytest.raises(ValueError):
        posthoc.core_distances_device(xd, k, dev)


def test_core_distances_sharded_over_two_ranks(tmp_path):
    """the fine-grained mode's post-hoc stage over the ranks that are already there.  Two ranks (sharing this box's
    GPU, collectives over gloo) split the rows of the one-pass window kernels; every rank ends with exactly the single-rank
    core distances (each row's value is a function of the inputs alone), and HDBSCAN from them gives the same labels and
    probabilities (tests/dist_core_worker.py)."""
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_core_worker.py")]
    r = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SHARDED_CORE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_core_neighbour_rank_switch(monkeypatch):
    """IDELUCS_HDBSCAN_RANK=hdbscan takes the core distance one neighbour further (min_samples + 1 counting the point itself: what
    the `hdbscan` package's tree queries ask for; unpinned, the package is absent) -- on the device that is sklearn's HDBSCAN with
    min_samples = min_cluster_size + 1, label for label; the default stays sklearn's own rank."""
    from sklearn.cluster import HDBSCAN
    from sklearn.metrics import adjusted_rand_score
    from idelucs_amd import posthoc
    x = _blobs(4000, seed=5)
    k = 41
    assert posthoc.core_neighbour_rank(k) == k
    monkeypatch.setenv("IDELUCS_HDBSCAN_RANK", "hdbscan")
    assert posthoc.core_neighbour_rank(k) == k + 1
    labels, prob = posthoc.hdbscan_device(x, k)
    ref = HDBSCAN(min_cluster_size=k, min_samples=k + 1).fit(x)
    assert np.array_equal(labels < 0, ref.labels_ < 0) and adjusted_rand_score(ref.labels_, labels) == 1.0
    assert np.allclose(prob, ref.probabilities_, atol=1e-9)
    plain = HDBSCAN(min_cluster_size=k).fit(x)
    monkeypatch.delenv("IDELUCS_HDBSCAN_RANK")
    l0, _ = posthoc.hdbscan_device(x, k)
    assert np.array_equal(l0 < 0, plain.labels_ < 0) and adjusted_rand_score(plain.labels_, l0) == 1.0
