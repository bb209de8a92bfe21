"""idelucs_amd.posthoc -- what runs ONCE after the hot path, on [V, N] ints / [N, 64] floats
(reference idelucs/utils.py:582-623 and idelucs/__main__.py:129-156).  SURVEY 8(f) rows f2/f3: kept
on the host with sklearn/scipy, as in the reference, on the outputs gathered from the GPUs.
"""
import os

# Variants kept for the tests that compare them with the default path (not environment switches; a test sets an entry with monkeypatch.setitem)
OPTIONS = {"knn": "",            # matrix | window: force one of the core-distance paths
           "mst": "lazy",        # lazy | local | plain: which Prim
           "mst_filter": "1",    # 0: no 8-bit lower bound in front of the exact distances
           "silhouette": "",     # gemm: the float64 GEMM form instead of the one-pass kernel
           "ensemble": "device"}  # sklearn: label_features on the host


def _dev_options():
    from . import _lib
    OPTIONS.update({k: v for k, v in _lib.DEV.items() if k in OPTIONS})


_dev_options()


import numpy as np


def relabel_first_occurrence(y_pred):
    """Reference __main__.py:129-139: relabel clusters 0,1,2,... in order of first appearance (int32)."""
    y = np.asarray(y_pred).astype(np.int32)
    _, first_idx, inv = np.unique(y, return_index=True, return_inverse=True)
    order = np.argsort(np.argsort(first_idx))
    return order[inv].astype(np.int32)


def label_features(predictions, n_clusters):
    """Reference utils.py:582-602: one-hot votes [N, V*C], centred, KMeans(n_init=10) (unseeded in the
    reference), confidence = normalised inverse squared distance to the centres."""
    from sklearn.cluster import KMeans
    from sklearn.metrics.pairwise import euclidean_distances
    predictions = np.asarray(predictions)
    v, n = predictions.shape
    features = np.zeros((n, v * n_clusters))
    for i in range(v):
        features[np.arange(n), i * n_clusters + predictions[i]] = 1.0
    centred = features - features.sum(axis=0) / n
    km = KMeans(n_clusters=n_clusters, init="k-means++", n_init=10)
    y = km.fit_predict(centred)
    d = 1.0 / euclidean_distances(centred, km.cluster_centers_, squared=True)
    d /= d.sum(axis=1)[:, np.newaxis]
    return np.array(y), d.max(axis=1)


def label_features_device(predictions, n_clusters, device=None, n_init=10, max_iter=300, seed=None):
    """label_features (reference utils.py:582-602) on the GPU: the same centred one-hot vote matrix
    [N, V*C], k-means++ seeding + Lloyd iterations as dense torch ops, best of `n_init` restarts by
    inertia, and the same inverse-squared-distance confidence.  KMeans is unseeded in the reference, so
    parity is at partition level (ARI vs sklearn on the same votes, tests/test_cli_surface.py).
    predictions: [V, N] integer tensor/array of per-voter labels in [0, n_clusters)."""
    import torch
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    pred = torch.as_tensor(np.asarray(predictions) if not torch.is_tensor(predictions) else predictions).to(dev).long()
    v, n = pred.shape
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed) if seed is not None else int(torch.seed() % (2 ** 31)))
    x = torch.zeros((n, v * n_clusters), dtype=torch.float32, device=dev)
    x.scatter_(1, (pred.t() + torch.arange(v, device=dev) * n_clusters), 1.0)
    x = x - x.sum(0, keepdim=True) / n                                  # utils.py:592-594
    x2 = (x * x).sum(1)

    def sqdist(c):                                                       # [N, K] squared distances to centres c [K, D]
        return (x2[:, None] - 2.0 * (x @ c.t()) + (c * c).sum(1)[None, :]).clamp_min_(0.0)

    best = None
    for _ in range(n_init):
        # k-means++ seeding
        idx = torch.randint(0, n, (1,), device=dev, generator=g)
        centres = x[idx].clone()
        d = sqdist(centres).squeeze(1)
        for _k in range(1, n_clusters):
            tot = d.sum()
            if float(tot) <= 0.0:                                        # fewer distinct rows than clusters
                nxt = torch.randint(0, n, (1,), device=dev, generator=g)
            else:
                nxt = torch.multinomial(d / tot, 1, generator=g)
            centres = torch.cat([centres, x[nxt]], 0)
            d = torch.minimum(d, sqdist(x[nxt]).squeeze(1))
        # Lloyd
        labels = None
        for _it in range(max_iter):
            new_labels = sqdist(centres).argmin(1)
            if labels is not None and torch.equal(new_labels, labels):
                break
            labels = new_labels
            sums = torch.zeros_like(centres).index_add_(0, labels, x)
            cnt = torch.bincount(labels, minlength=n_clusters).to(x.dtype)
            keep = cnt > 0
            centres = torch.where(keep[:, None], sums / cnt.clamp_min(1.0)[:, None], centres)
        dist = sqdist(centres)
        inertia = float(dist.gather(1, labels[:, None]).sum())
        if best is None or inertia < best[0]:
            best = (inertia, labels.clone(), dist.clone())
    _, labels, dist = best
    w = 1.0 / dist                                                       # utils.py:597-600 (inf where a point sits on a centre, as in the reference)
    w = w / w.sum(1, keepdim=True)
    return labels.cpu().numpy(), w.max(1).values.double().cpu().numpy()


SILHOUETTE_HOST_MAX = 20000      # above this many points the silhouette is computed on the GPU (sklearn's is O(N^2) on the host)


def _silhouette_from_sums(sums, own, counts):
    """mean over the points of (b - a) / max(a, b) from the per-cluster distance sums [n, K] (float64), the points' own clusters
    and the cluster sizes: a = mean distance to the rest of the own cluster, b = smallest mean distance to another cluster;
    singletons score 0 (sklearn)."""
    import torch
    n_own = counts[own]
    a = sums.gather(1, own[:, None]).squeeze(1) / (n_own - 1.0).clamp_min(1.0)
    means = sums / counts[None, :]
    means.scatter_(1, own[:, None], float("inf"))
    b = means.min(1).values
    sil = (b - a) / torch.maximum(a, b)
    sil = torch.where(n_own > 1.0, sil, torch.zeros_like(sil))
    return torch.nan_to_num(sil).sum()


def _silhouette_one_pass(x, lab, k, dev):
    """The per-cluster distance sums by idl_silhouette_sums (csrc/knn.hip): the points cluster by cluster, every cluster padded to
    whole 64-row waves (weight 0; the padding repeats a point of the cluster: a wave centres its coordinates on its first row, and
    a wave that straddled two tight far-apart clusters would lose the second one's distances to rounding), one MFMA pass over
    all pairs.  -> sum over the points of their silhouette (float64 device scalar)."""
    import ctypes
    import torch
    from . import _lib
    n = x.shape[0]
    order = torch.argsort(lab, stable=True)
    lab_s = lab[order]
    counts = torch.bincount(lab, minlength=k)
    padded = (counts + 63) // 64 * 64                         # whole waves: a wave's 64 rows (and its centre) belong to one cluster
    first = torch.cumsum(counts, 0) - counts                  # first sorted index of each cluster
    start = torch.cumsum(padded, 0) - padded                  # first padded row of each cluster
    npad = int(padded.sum())
    pos = start[lab_s] + (torch.arange(n, device=dev) - first[lab_s])
    xs = x[order]
    row_cluster = torch.repeat_interleave(torch.arange(k, device=dev), padded)
    xp = xs[first[row_cluster]].contiguous()
    xp[pos] = xs
    w = torch.zeros(npad, dtype=torch.float32, device=dev)
    w[pos] = 1.0
    tile_cluster = torch.repeat_interleave(torch.arange(k, device=dev, dtype=torch.int32), (padded // 16))
    sums = torch.zeros((npad, k), dtype=torch.float32, device=dev)
    vp = ctypes.c_void_p
    _lib.check(_lib.lib.idl_silhouette_sums(vp(xp.data_ptr()), vp(w.data_ptr()), vp(tile_cluster.data_ptr()), npad, 64, k, vp(sums.data_ptr()),
                                            vp(torch.cuda.current_stream().cuda_stream)))
    return _silhouette_from_sums(sums[pos].double(), lab_s, counts.double())


SILHOUETTE_ONE_PASS_MIN = 4096   # points from which 64-dimensional data take the one-pass kernel


def silhouette_score_device(data, labels, device=None, block=4096):
    """sklearn.metrics.silhouette_score(data, labels) (euclidean, mean over all samples) on the GPU.  64-dimensional data (the
    latent) from SILHOUETTE_ONE_PASS_MIN points: one pass over all pairs on the fp32 matrix cores that adds every point's distances
    up per cluster in registers (_silhouette_one_pass; 10^6 points: 2 s).  Otherwise ($IDELUCS_DEV=silhouette=gemm forces it): for a
    block of rows the distances to every point come from one GEMM (||x||^2 + ||y||^2 - 2 x.y, clamped, square-rooted) and their
    per-cluster sums from a second GEMM with the one-hot label matrix (10 s at 10^6 points: five elementwise passes over 4 TB).
    float64 from the per-cluster sums on; agrees with sklearn to ~1e-6 (tests/test_cli_surface.py)."""
    import torch
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    x = torch.as_tensor(np.asarray(data)).to(dev, torch.float32)
    uniq, inv = np.unique(np.asarray(labels), return_inverse=True)
    n, k = x.shape[0], len(uniq)
    if not 2 <= k <= n - 1:
        raise ValueError("Number of labels is %d. Valid values are 2 to n_samples - 1 (inclusive)" % k)      # sklearn's check
    lab = torch.from_numpy(inv.astype(np.int64)).to(dev)
    if x.shape[1] == 64 and n >= SILHOUETTE_ONE_PASS_MIN and OPTIONS["silhouette"] != "gemm":
        return float(_silhouette_one_pass(x, lab, k, dev).item() / n)
    onehot = torch.zeros((n, k), dtype=torch.float32, device=dev)
    onehot[torch.arange(n, device=dev), lab] = 1.0
    counts = onehot.sum(0).double()
    x = x - x.mean(0, keepdim=True)                 # distances are translation invariant; centring keeps the cancellation small
    x2 = (x * x).sum(1)
    total = torch.zeros((), dtype=torch.float64, device=dev)
    block = max(64, min(block, (1 << 32) // max(n, 1)))        # one [block, n] float32 buffer (<= 16 GB), reused by every block
    buf = torch.empty((min(block, n), n), dtype=torch.float32, device=dev)
    xt = x.t()
    for lo in range(0, n, block):
        hi = min(lo + block, n)
        d2 = buf[:hi - lo]
        torch.mm(x[lo:hi], xt, out=d2)
        d2.mul_(-2.0).add_(x2[lo:hi, None]).add_(x2[None, :]).clamp_min_(0.0)
        d2[torch.arange(hi - lo, device=dev), torch.arange(lo, hi, device=dev)] = 0.0       # exact zeros on the diagonal
        sums = (d2.sqrt_() @ onehot).double()                                                # [rows, K]: sum of distances to each cluster
        total += _silhouette_from_sums(sums, lab[lo:hi], counts)
    return float(total.item() / n)


def compute_results(y_pred, data, y_true=None):
    """Reference utils.py:606-623.  (Silhouette: sklearn up to SILHOUETTE_HOST_MAX points, the same score on the GPU above.)"""
    import sklearn.metrics.cluster as metrics
    from .utils import cluster_acc
    if len(y_pred) > SILHOUETTE_HOST_MAX:
        sil = silhouette_score_device(data, y_pred)
    else:
        sil = metrics.silhouette_score(data, y_pred)
    d = {"Davies-Boulding": metrics.davies_bouldin_score(data, y_pred),
         "Silhouette-Score": sil}
    if y_true is None:
        return d, None
    d["NMI"] = metrics.adjusted_mutual_info_score(y_true, y_pred)
    d["ARI"] = metrics.adjusted_rand_score(y_true, y_pred)
    d["Homogeneity"] = metrics.homogeneity_score(y_true, y_pred)
    d["Completeness"] = metrics.completeness_score(y_true, y_pred)
    ind, acc = cluster_acc(y_true, y_pred)
    d["ACC"] = acc
    return d, ind


def plot_confusion_matrix(cm, target_names, pairs=None, title="Confusion matrix", cmap=None, normalize=False, ax=None):
    """The picture the reference CLI saves as contingency_matrix.jpg when n_clusters < 16 (utils.py:527-577, called at
    __main__.py:178): the matrix as a 'Blues' heat map with one text cell per entry, class names on the y axis, accuracy
    (trace, or the cells named by `pairs`) and misclassification rate under the x axis; tables with more than 16 classes are
    replaced by a pointer to the .tsv."""
    import matplotlib.pyplot as plt
    cm = np.asarray(cm)
    if len(target_names) > 16:
        ax.text(0, 0.5, "The confusion matrix is too big to display, see .tsv file instead", fontsize=12)
        ax.axis("off")
        return
    hits = np.trace(cm) if not isinstance(pairs, np.ndarray) else sum(cm[j][i] for i, j in pairs)
    accuracy = hits / float(cm.sum())
    ax.imshow(cm, interpolation="nearest", cmap=cmap if cmap is not None else plt.get_cmap("Blues"))
    ax.set_title(title)
    if target_names is not None:
        ticks = np.arange(len(target_names))
        ax.set_xticks(ticks); ax.set_xticklabels([""] * len(target_names))
        ax.set_yticks(ticks); ax.set_yticklabels(target_names)
    shown = cm.astype("float") / cm.sum(axis=1)[:, np.newaxis] if normalize else cm
    thresh = shown.max() / (1.5 if normalize else 2)
    fmt = "{:0.3f}" if normalize else "{:,}"
    for (i, j), val in np.ndenumerate(shown):
        ax.text(j, i, fmt.format(val), horizontalalignment="center", color="white" if val > thresh else "black")
    ax.set_ylabel("True label")
    ax.set_xlabel("Predicted label\naccuracy={:0.4f}; misclass={:0.4f}".format(accuracy, 1 - accuracy))


HDBSCAN_EXACT_MAX = 20000        # points the host HDBSCAN is run on directly


HDBSCAN_DEVICE_MIN = 2000        # without the `hdbscan` package: points from which sklearn's HDBSCAN is run on the GPU (same result, 30 x faster at 30 000)


def _hdbscan(points, min_cluster_size, device=None):
    """The reference's call (hdbscan.HDBSCAN, __main__.py:153) when that package is importable; else its stand-in
    sklearn.cluster.HDBSCAN -- on the host for small inputs, through hdbscan_device (the same algorithm, the same labels and
    probabilities: test_hdbscan_on_the_device_is_sklearns) from HDBSCAN_DEVICE_MIN points."""
    try:
        import hdbscan
        cl = hdbscan.HDBSCAN(min_cluster_size=min_cluster_size, gen_min_span_tree=True, prediction_data=True)
    except ImportError:
        if len(points) >= HDBSCAN_DEVICE_MIN and os.environ.get("IDELUCS_HDBSCAN", "device") != "host":
            return hdbscan_device(points, max(min_cluster_size, 2), device=device)
        from sklearn.cluster import HDBSCAN
        mcs = max(min_cluster_size, 2)
        cl = HDBSCAN(min_cluster_size=mcs, min_samples=min(core_neighbour_rank(mcs), len(points)))     # (min_samples = mcs unless IDELUCS_HDBSCAN_RANK=hdbscan)
    cl.fit(points)
    return cl.labels_, cl.probabilities_


def _core_distances_rows(x64, sq, rows_idx, k, device, out):
    """out[rows_idx] = distance of those rows to their k-th nearest neighbour, itself included: row blocks of the float64 Gram-form
    distance matrix (one GEMM each), the k + 32 smallest per row by radix select (torch.topk), and the k-th read off the exact
    distances (difference vector, sklearn's order of operations) of the 65 around rank k: the Gram form only FINDS them.
    rows_idx: None = every row, else an int64 tensor of row numbers."""
    import torch
    n = x64.shape[0]
    m = n if rows_idx is None else int(rows_idx.numel())
    if m == 0:
        return
    rows = max(64, min(m, (1 << 30) // max(n, 1)))            # 8 GB of float64 distances per block
    buf = torch.empty((min(rows, m), n), dtype=torch.float64, device=device)      # one block buffer, reused
    xt = x64.t()
    for lo in range(0, m, rows):
        ridx = torch.arange(lo, min(lo + rows, m), device=device) if rows_idx is None else rows_idx[lo:lo + rows]
        xb = x64[ridx]
        d2 = buf[:xb.shape[0]]
        torch.mm(xb, xt, out=d2)
        d2.mul_(-2.0).add_(sq[None, :]).add_(sq[ridx, None])
        d2[torch.arange(xb.shape[0], device=device), ridx] = 0.0     # a point is its own first neighbour
        # the k-th smallest per row: unsorted top-k (multi-block radix select, 2.5 x faster than torch.kthvalue on float64 rows of
        # 10^6); the Gram form's own rounding (1e-16 of the norms, which tight far-apart clusters make 1e-11 of the distance) can
        # swap near-ties, so the k-th is then read off the EXACT distances of the ranks k - pad .. k + pad
        pad = min(32, k - 1, n - k)
        vals, cols = torch.topk(d2, k + pad, dim=1, largest=False, sorted=False)
        near = cols.gather(1, torch.topk(vals, 2 * pad + 1, dim=1, largest=True, sorted=False).indices)     # the columns of those ranks
        ex = _distance_in_sklearns_order(xb[:, None, :].expand(-1, near.shape[1], -1).reshape(-1, xb.shape[1]),
                                         x64[near.reshape(-1)]).view(xb.shape[0], -1)
        out[ridx] = ex.kthvalue(pad + 1, dim=1).values
        del vals, cols, near, ex
    del buf


def _distance_in_sklearns_order(a, b):
    """sqrt(sum_c (a_c - b_c)^2) per row with the sum taken coordinate by coordinate, product and sum each rounded (sklearn's
    EuclideanDistance loop; a tree reduction would differ in the last bit)."""
    import torch
    acc = torch.zeros(a.shape[0], dtype=torch.float64, device=a.device)
    for c in range(a.shape[1]):
        t = a[:, c] - b[:, c]
        acc += t * t
    return acc.sqrt_()


KNN_WINDOW_MIN = 32768           # points from which the core distances take the one-pass window kernels (csrc/knn.hip)
KNN_SAMPLE = 16384               # columns sampled to bracket every row's k-th distance
KNN_SIGMAS = 4.5                 # half-width of the bracket in standard deviations of the sampled rank
KNN_SLOT_BYTES = 16 << 30        # candidate slots (8 bytes each) held at a time


def _spatial_order(x64, pivots=256, seed=0):
    """(perm, gid): a permutation that puts neighbours in space next to each other in memory -- the points grouped by their
    nearest of `pivots` random points -- and the group (0.., non-decreasing) of every position of that order.  idl_knn_window
    centres every 64 rows on the first of them and idl_mst_prim_local codes every point inside its group's box: the rounding
    bound of the one and the step of the other are the spread of a group."""
    import torch
    n = x64.shape[0]
    g = torch.Generator(device="cpu"); g.manual_seed(seed + 1)
    p = x64[torch.randperm(n, generator=g)[:min(pivots, n)].to(x64.device)]
    group = torch.empty(n, dtype=torch.int64, device=x64.device)
    for lo in range(0, n, 1 << 18):
        xb = x64[lo:lo + (1 << 18)]
        group[lo:lo + (1 << 18)] = ((p * p).sum(1)[None, :] - 2.0 * (xb @ p.t())).argmin(1)
    perm = torch.argsort(group, stable=True)
    gid = torch.unique_consecutive(group[perm], return_inverse=True)[1]
    return perm, gid


def _core_distances_window(x64, k, device, out, sample=None, seed=0, stats=None, order=None, shard=None):
    """Core distances without the distance matrix (csrc/knn.hip).  Needs 64 coordinates that float32 holds exactly.
      0. order the points by their nearest of 256 random pivots (_spatial_order), every group padded to whole waves of 64 rows;
      1. bracket: the squared distances (float64) of every row to KNN_SAMPLE random columns; the k-th of all n lies, with
         probability 1 - 7e-6 per row, between the sampled ranks r -+ 4.5 sqrt(r), r = sample * k / n  ->  [lo, hi) per row;
      2. idl_knn_window: one fp32 MFMA pass over all pairs: count of columns below lo, the columns inside [lo, hi) kept;
      3. idl_knn_select: radix select among the kept ones, then the float64 distances (difference vector, sklearn's order of
         operations) of everything within twice the pass's rounding bound of the selected value: the exact k-th.
    Rows the bracket missed (status != 0) are returned for the caller's matrix path: the int64 tensor of those rows, or None
    when k / n allows no bracket.
    shard = (rank, world): the ranks of a process group split the padded rows between them (whole 256-row blocks; brackets, window
    pass and selection for the own rows only -- each row's result is a function of the inputs alone, so it is the single-rank
    result bit for bit) and add their float64 / status vectors up (one all-reduce each): every rank returns the full vector."""
    import ctypes
    import math
    import torch
    from . import _lib
    L = _lib.lib
    n, d = x64.shape
    S = min(int(sample or KNN_SAMPLE), n)
    frac = k / n
    r = S * frac
    sd = math.sqrt(max(r * (1.0 - frac), 1e-9))
    r_lo, r_hi = int(math.floor(r - KNN_SIGMAS * sd)), int(math.ceil(r + KNN_SIGMAS * sd)) + 1
    if r_hi > S:
        return None                                            # k too close to n for a bracket: the matrix path
    sharded = shard is not None and shard[1] > 1
    agreed = [False]                                            # set once the ranks' agreement collective has run (see below)

    def body():
        perm, gid = order if order is not None else _spatial_order(x64, seed=seed)
        bufs = {"xo": x64[perm]}                                # the points in memory order (owned by the row part, which frees them when done with them)
        bufs["sq"] = (bufs["xo"] * bufs["xo"]).sum(1)
        g = torch.Generator(device="cpu"); g.manual_seed(seed)
        cols = torch.randperm(n, generator=g)[:S].to(device)
        xs_t, sqs = bufs["xo"][cols].t().contiguous(), bufs["sq"][cols]
        # every group padded to whole waves of 64 rows, so that no wave (which centres its coordinates on its first row) straddles two groups;
        # the padding is infinitely far from everything (never counted, never kept) and brackets nothing itself
        counts = torch.bincount(gid)
        padded = (counts + 63) // 64 * 64
        first, start = torch.cumsum(counts, 0) - counts, torch.cumsum(padded, 0) - padded
        pos = start[gid] + (torch.arange(n, device=device) - first[gid])
        npad = int(padded.sum())
        # this rank's share of the padded rows (whole 256-row blocks; everything without a process group)
        p_lo, p_hi = 0, npad
        if sharded:
            blocks = -(-npad // 256)
            per = -(-blocks // shard[1])
            p_lo, p_hi = min(npad, shard[0] * per * 256), min(npad, (shard[0] + 1) * per * 256)
        return _core_distances_window_rows(x64, k, device, out, stats, perm, gid, bufs, xs_t, sqs, pos, npad, p_lo, p_hi, r_lo, r_hi, S, n, d, shard, agreed)

    if not sharded:
        return body()
    #  the ranks add disjoint shards up at positions every rank derives for itself: the order must be the same everywhere
    # (core_distances_sharded broadcasts rank 0's), and a rank that fails in its share -- anywhere from its first allocation on (out of memory) --
    # must still reach the AGREEMENT collective its peers wait in: it says so there, then every rank raises.  A failure BEHIND the agreement
    # (in the sums, or in the bookkeeping after them) must not issue a second, unmatched collective: the peers are past it.
    try:
        return body()
    except ShardFailed:
        raise
    except Exception as err:      # noqa: BLE001 -- whatever it was, the other ranks may be waiting
        if not agreed[0]:
            _all_ranks_ok(False, device)
        raise ShardFailed(f"this rank failed in its share of the core distances: {err!r}") from err


class ShardFailed(RuntimeError):
    """A rank of the process group could not do its share of a sharded stage; every rank raises it (core_distances_sharded then
    leaves the stage to rank 0 alone)."""


def _all_ranks_ok(ok, device):
    """MIN all-reduce of a flag: True when every rank of the group says so."""
    import torch
    import torch.distributed as tdist
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    tdist.all_reduce(t, op=tdist.ReduceOp.MIN)
    return bool(int(t.item()))


def _core_distances_window_rows(x64, k, device, out, stats, perm, gid, bufs, xs_t, sqs, pos, npad, p_lo, p_hi, r_lo, r_hi, S, n, d, shard, agreed):
    """The row part of _core_distances_window: brackets, window pass and selection of the padded rows [p_lo, p_hi), then (sharded) the
    sum over the ranks.  bufs = {"xo", "sq"}: taken over (the only references, so that `del` below frees their 0.5 GB at 10^6 points);
    agreed[0] becomes True once the ranks' agreement collective has run."""
    xo, sq = bufs.pop("xo"), bufs.pop("sq")
    import ctypes
    import math
    import torch
    from . import _lib
    L = _lib.lib
    vp = ctypes.c_void_p
    mine = torch.nonzero((pos >= p_lo) & (pos < p_hi)).squeeze(1)            # positions (memory order) whose padded row is this rank's
    # the bracket of every own row, from the float64 distances to the sampled columns
    lo_all = torch.full((n,), -1.0, dtype=torch.float32, device=device)
    hi_all = torch.full((n,), -1.0, dtype=torch.float32, device=device)
    sub = 16384                                                 # rows of one sampled block: 16384 x 16384 float64 = 2 GB
    for b0 in range(0, int(mine.numel()), sub):
        idx = mine[b0:b0 + sub]
        ds = torch.mm(xo[idx], xs_t).mul_(-2.0).add_(sqs[None, :]).add_(sq[idx, None])
        vals = torch.topk(ds, r_hi, dim=1, largest=False, sorted=True).values
        hi_all[idx] = vals[:, r_hi - 1].to(torch.float32)
        lo_all[idx] = vals[:, r_lo - 1].to(torch.float32) if r_lo >= 1 else -1.0e30   # (no lower rank: nothing is below, every column under hi is kept)
        del ds, vals
    x32 = torch.zeros((npad, d), dtype=torch.float32, device=device)
    x32[:, 0] = 1.0e18
    x32[pos] = xo.to(torch.float32)
    lo_p = torch.full((npad,), -1.0, dtype=torch.float32, device=device)
    hi_p = torch.full((npad,), -1.0, dtype=torch.float32, device=device)
    lo_p[pos], hi_p[pos] = lo_all, hi_all
    del xo, sq, lo_all, hi_all
    m_expect = (r_hi - max(r_lo, 0)) / S * n
    cap = int(-(-int(2.0 * m_expect + 6.0 * math.sqrt(m_expect) + 1024) // 256) * 256)
    chunk = max(256, min(-(-npad // 256) * 256, (KNN_SLOT_BYTES // (8 * cap)) // 256 * 256))
    cand_d2 = torch.empty(chunk * cap, dtype=torch.float32, device=device)
    cand_ix = torch.empty(chunk * cap, dtype=torch.int32, device=device)
    delta = torch.empty(chunk, dtype=torch.float32, device=device)
    cnt_lo = torch.empty(chunk, dtype=torch.int32, device=device)
    cnt_in = torch.empty(chunk, dtype=torch.int32, device=device)
    status_p = torch.zeros(npad, dtype=torch.int32, device=device)
    core_p = torch.zeros(npad, dtype=torch.float64, device=device)
    stream = vp(torch.cuda.current_stream().cuda_stream)
    for row0 in range(p_lo, p_hi, chunk):
        rows = min(chunk, p_hi - row0)
        _lib.check(L.idl_knn_window(vp(x32.data_ptr()), npad, d, vp(lo_p[row0:].data_ptr()), vp(hi_p[row0:].data_ptr()), row0, rows, vp(cnt_lo.data_ptr()),
                                    vp(cnt_in.data_ptr()), vp(delta.data_ptr()), vp(cand_d2.data_ptr()), vp(cand_ix.data_ptr()), cap, stream))
        _lib.check(L.idl_knn_select(vp(x32.data_ptr()), npad, d, vp(lo_p[row0:].data_ptr()), vp(hi_p[row0:].data_ptr()), vp(delta.data_ptr()), row0, rows, k,
                                    vp(cnt_lo.data_ptr()), vp(cnt_in.data_ptr()), vp(cand_d2.data_ptr()), vp(cand_ix.data_ptr()), cap,
                                    vp(core_p.data_ptr()), vp(status_p[row0:].data_ptr()), stream))
        if stats is not None:
            stats["kept_max"] = max(stats.get("kept_max", 0), int(cnt_in[:rows].max()))
    if shard is not None and shard[1] > 1:                      # the other ranks' rows: zeros here, theirs there
        import torch.distributed as tdist
        ok = _all_ranks_ok(True, device)                        # (a rank that failed above says so here: nobody waits in the sums below)
        agreed[0] = True
        if not ok:
            raise ShardFailed("another rank failed in its share of the core distances")
        core_p[:p_lo] = 0.0; core_p[p_hi:] = 0.0
        status_p[:p_lo] = 0; status_p[p_hi:] = 0
        tdist.all_reduce(core_p)
        tdist.all_reduce(status_p)
    core_o, status = core_p[pos], status_p[pos]
    out[perm] = core_o
    missed = perm[torch.nonzero(status).squeeze(1)]
    if stats is not None:
        stats.update(sample=S, ranks=(r_lo, r_hi), cap=cap, chunk=chunk, missed=int(missed.numel()),
                     status_counts=torch.bincount(status, minlength=5).tolist())
    return missed


def core_distances_device(x64, k, device, f32_exact=None, stats=None, order=None, shard=None):
    """Distance of every point to its k-th nearest neighbour, itself included (sklearn _hdbscan_prims: NearestNeighbors(
    n_neighbors=k).kneighbors(X)[:, -1]).  From KNN_WINDOW_MIN points of 64 float32-exact coordinates (the latent of the
    reference's networks) the one-pass window kernels; else, and for the rows their bracket missed, the float64 Gram-form matrix
    in row blocks (_core_distances_rows).  $IDELUCS_DEV=knn = matrix | window forces one.
    shard = (rank, world) (every rank of the group must call, with the same points): the window path's rows are split over the
    ranks and the result all-reduced (_core_distances_window); the few rows the brackets missed, and the matrix path, are computed
    by every rank for itself -- the same values everywhere."""
    import torch
    n = x64.shape[0]
    sq = (x64 * x64).sum(1)
    core = torch.empty(n, dtype=torch.float64, device=device)
    mode = OPTIONS["knn"]
    if f32_exact is None:
        f32_exact = bool((x64.to(torch.float32).double() == x64).all())
    todo = None
    if mode != "matrix" and f32_exact and x64.shape[1] == 64 and (n >= KNN_WINDOW_MIN or mode == "window"):
        todo = _core_distances_window(x64, k, device, core, stats=stats, order=order, shard=shard)
        if todo is None and mode == "window":
            raise ValueError("core_distances_device: no bracket for this k / n (IDELUCS_DEV=knn=window)")
    if todo is None:
        _core_distances_rows(x64, sq, None, k, device, core)
    elif todo.numel():
        _core_distances_rows(x64, sq, todo, k, device, core)
    return core


MST_LAZY_MIN = 65536             # points from which groups of points may sleep during Prim's scan (idl_mst_prim_lazy)
MST_LAZY_MAX = 1 << 20           # its step kernel scans a thread's 4 look-ahead points only: 1024 workgroups x 256 threads x 4 (mst.hip
                                 # prim_grid / PRIM_NT / PRIM_AHEAD); larger inputs take idl_mst_prim_local, whose scan strides on
MST_FILTER_MIN = 20000           # points from which Prim's scan goes through the 8-bit lower-bound filter


def _local_q8(xo, gid):
    """The 8-bit picture of the points for idl_mst_prim_local.  xo: the points in memory order, gid: their groups.  Per group its
    box corner lo_g (float32) and code step scale_g = widest side / 255; per point codes = round((x - lo_g) / scale_g) and an upper
    bound of ||x - y||, y = lo_g + scale_g * code.  Any corner / step is CORRECT (the residual is the point's own).
    -> (codes int32 [d/4, n] four features per word, resid float32 [n], glo float32 [G, d], gscale float64 [G])"""
    import torch
    n, d = xo.shape
    G = int(gid[-1]) + 1
    idx = gid[:, None].expand(-1, d)
    lo = torch.full((G, d), float("inf"), dtype=torch.float64, device=xo.device).scatter_reduce_(0, idx, xo, "amin")
    hi = torch.full((G, d), float("-inf"), dtype=torch.float64, device=xo.device).scatter_reduce_(0, idx, xo, "amax")
    glo = lo.to(torch.float32)
    glo = torch.where(glo.double() > lo, torch.nextafter(glo, torch.full_like(glo, float("-inf"))), glo)    # a corner at or below every point
    lo = glo.double()
    span = (hi - lo).max(1).values
    gscale = torch.where(span > 0, span / 255.0, torch.ones_like(span))
    q = ((xo - lo[gid]) / gscale[gid, None]).round_().clamp_(0.0, 255.0)
    resid = (xo - (lo[gid] + gscale[gid, None] * q)).pow_(2).sum(1).sqrt_()
    resid = resid * (1.0 + 1e-6) + 1e-9 * (float(xo.abs().max()) + 255.0 * float(gscale.max()))     # covers the float64 rounding of y and of the exact distances
    r32 = resid.to(torch.float32)
    r32 = torch.where(r32.double() < resid, torch.nextafter(r32, torch.full_like(r32, float("inf"))), r32)
    q4 = q.to(torch.int64).view(n, d // 4, 4)
    del q
    word = q4[..., 0] | (q4[..., 1] << 8) | (q4[..., 2] << 16) | (q4[..., 3] << 24)
    word = torch.where(word >= (1 << 31), word - (1 << 32), word).to(torch.int32)             # the same 32 bits
    return word.t().contiguous(), r32.contiguous(), glo.contiguous(), gscale.contiguous()


def core_neighbour_rank(min_samples):
    """The neighbour whose distance is a point's core distance, counting the point itself.  Default: min_samples -- sklearn's
    _hdbscan_prims, `NearestNeighbors(n_neighbors=min_samples).kneighbors(X)[:, -1]`, the stand-in this package is pinned to.
    IDELUCS_HDBSCAN_RANK=hdbscan: min_samples + 1, what the `hdbscan` package the reference calls (idelucs/__main__.py:83,
    hdbscan==0.8.32) asks its trees for -- hdbscan/hdbscan_.py, _hdbscan_prims_kdtree / _hdbscan_prims_balltree:
    `core_distances = tree.query(X, k=min_samples + 1, dualtree=True, breadth_first=True)[0][:, -1]` (its generic path takes
    np.partition(distance_matrix, min_points, axis=0)[min_points], the same rank) -- one neighbour further.  The package is absent
    here, so this switch is UNPINNED (cited from the package's source as recalled), and the default stays sklearn's."""
    return int(min_samples) + (1 if os.environ.get("IDELUCS_HDBSCAN_RANK", "sklearn") == "hdbscan" else 0)


def hdbscan_device(points, min_cluster_size, device=None, stats=None, core=None, shard=None):
    """sklearn.cluster.HDBSCAN(min_cluster_size).fit(points) -> (labels_, probabilities_) with the two O(N^2) stages on the GPU:
    core distances (core_distances_device) and Prim's minimum spanning tree of the mutual-reachability graph (csrc/mst.hip,
    idl_mst_prim: sklearn's mst_from_data_matrix visit for visit, in float64); the edges then go through sklearn's own
    single-linkage / condensed-tree code (sklearn.cluster._hdbscan: make_single_linkage, tree_to_labels -- private names of
    sklearn 1.7, the stand-in for the absent `hdbscan` package: SURVEY 8c), so the labels are sklearn's.
    core: the core distances (float64 [n], device or host) when the caller has them already (the CLI computes them over all ranks of
    a multi-GPU job before the others leave: core_distances_sharded); shard = (rank, world): compute them here, split over the
    ranks of the process group (every rank must call with the same points; all return the same labels)."""
    import ctypes
    import torch
    try:
        from sklearn.cluster._hdbscan._linkage import MST_edge_dtype, make_single_linkage
        from sklearn.cluster._hdbscan._tree import tree_to_labels
    except ImportError as err:
        # another scikit-learn series: its own (host) HDBSCAN -- the same algorithm, O(N^2) on the CPU
        import warnings
        from sklearn.cluster import HDBSCAN
        warnings.warn(f"idelucs_amd: sklearn's private HDBSCAN tree code is not importable ({err}); running sklearn.cluster.HDBSCAN on the host")
        kk = max(int(min_cluster_size), 2)     # (min_samples as the device path takes it: IDELUCS_HDBSCAN_RANK -- ADVICE r4)
        cl = HDBSCAN(min_cluster_size=kk, min_samples=min(core_neighbour_rank(kk), len(points))).fit(np.asarray(points, dtype=np.float64))
        return cl.labels_, cl.probabilities_
    from . import _lib
    L = _lib.lib
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    pts = np.ascontiguousarray(np.asarray(points, dtype=np.float64))
    n, d = pts.shape
    k = max(int(min_cluster_size), 2)
    if not np.all(np.isfinite(pts)):
        raise ValueError("hdbscan_device: non-finite coordinates")
    if n < 2 or k > n:
        raise ValueError(f"hdbscan_device: min_samples ({k}) must be at most the number of points ({n})")
    x64 = torch.from_numpy(pts).to(dev)
    f32_exact = bool(np.array_equal(pts.astype(np.float32).astype(np.float64), pts))
    import time
    t0 = time.time()
    use_filter = d % 4 == 0 and d <= 64 and n >= MST_FILTER_MIN and OPTIONS["mst_filter"] != "0"
    order = _spatial_order(x64) if use_filter else None
    if core is None:
        core = core_distances_device(x64, min(core_neighbour_rank(k), n), dev, f32_exact=f32_exact, stats=stats, order=order, shard=shard)
    else:
        core = torch.as_tensor(core, dtype=torch.float64).to(dev).contiguous()
        assert core.numel() == n
    if stats is not None:
        torch.cuda.synchronize(dev); stats["core_s"] = time.time() - t0; t0 = time.time()
    vp = ctypes.c_void_p
    cur = torch.empty(n - 1, dtype=torch.int64, device=dev)
    nxt = torch.empty(n - 1, dtype=torch.int64, device=dev)
    w = torch.empty(n - 1, dtype=torch.float64, device=dev)
    ws = torch.empty(int(L.idl_mst_prim_workspace(n)) + 256, dtype=torch.uint8, device=dev)
    off = (-ws.data_ptr()) % 256
    stream = vp(torch.cuda.current_stream().cuda_stream)
    if use_filter:
        perm, gid = order
        xo = x64[perm]
        xt = xo.t().contiguous().to(torch.float32 if f32_exact else torch.float64)       # feature-major, float32 when that is lossless
        core_o = core[perm].contiguous()
        codes, resid, glo, gscale = _local_q8(xo, gid)
        orig, gid32 = perm.to(torch.int32), gid.to(torch.int32)
        start = int(torch.nonzero(perm == 0)[0, 0])
        del xo
        n_groups = int(gid32[-1]) + 1
        mode = OPTIONS["mst"]
        if mode == "lazy" and f32_exact and d == 64 and MST_LAZY_MIN <= n <= MST_LAZY_MAX and n_groups <= min(1024, -(-n // 256)):
            # groups of points may sleep while the tree grows elsewhere (idl_mst_prim_lazy): a ball around every group, the points row-major
            xo = x64[perm]
            cnt = torch.bincount(gid, minlength=n_groups)
            gfirst = torch.zeros(n_groups + 1, dtype=torch.int64, device=dev)
            gfirst[1:] = torch.cumsum(cnt, 0)
            gc = torch.zeros((n_groups, d), dtype=torch.float64, device=dev).index_add_(0, gid, xo) / cnt.clamp_min(1)[:, None].double()
            rad = (xo - gc[gid]).pow_(2).sum(1).sqrt_()
            gr = torch.zeros(n_groups, dtype=torch.float64, device=dev).scatter_reduce_(0, gid, rad, "amax")
            gr = gr * (1.0 + 1e-9) + 1e-9 * (float(xo.abs().max()) + 1.0)
            xrow = xo.to(torch.float32).contiguous()
            del xo, rad
            ws2 = torch.empty(int(L.idl_mst_prim_lazy_workspace(n, n_groups)) + 256, dtype=torch.uint8, device=dev)
            off2 = (-ws2.data_ptr()) % 256
            st3 = (ctypes.c_int64 * 3)()
            _lib.check(L.idl_mst_prim_lazy(vp(xt.data_ptr()), vp(xrow.data_ptr()), vp(core_o.data_ptr()), n, d, vp(orig.data_ptr()), start,
                                           vp(codes.data_ptr()), vp(resid.data_ptr()), vp(gid32.data_ptr()), vp(glo.data_ptr()), vp(gscale.data_ptr()),
                                           n_groups, vp(gfirst.data_ptr()), vp(gc.data_ptr()), vp(gr.data_ptr()), vp(cur.data_ptr()), vp(nxt.data_ptr()),
                                           vp(w.data_ptr()), vp(ws2.data_ptr() + off2), stream, ctypes.cast(st3, ctypes.c_void_p)))
            if stats is not None:
                stats["prim_launches"], stats["prim_stalls"], stats["prim_censuses"] = int(st3[0]), int(st3[1]), int(st3[2])
        else:
            _lib.check(L.idl_mst_prim_local(vp(xt.data_ptr()), 0 if f32_exact else 1, vp(core_o.data_ptr()), n, d, vp(orig.data_ptr()), start,
                                            vp(codes.data_ptr()), vp(resid.data_ptr()), vp(gid32.data_ptr()), vp(glo.data_ptr()),
                                            vp(gscale.data_ptr()), vp(cur.data_ptr()), vp(nxt.data_ptr()), vp(w.data_ptr()), vp(ws.data_ptr() + off), stream))
    else:
        xt = x64.t().contiguous().to(torch.float32 if f32_exact else torch.float64)
        _lib.check(L.idl_mst_prim(vp(xt.data_ptr()), 0 if f32_exact else 1, vp(core.data_ptr()), n, d, vp(cur.data_ptr()), vp(nxt.data_ptr()),
                                  vp(w.data_ptr()), vp(ws.data_ptr() + off), stream))
    mst = np.empty(n - 1, dtype=MST_edge_dtype)
    mst["current_node"], mst["next_node"], mst["distance"] = cur.cpu().numpy(), nxt.cpu().numpy(), w.cpu().numpy()
    if stats is not None:
        stats["prim_s"] = time.time() - t0; t0 = time.time()
        stats["mst_edges"] = mst.copy()
    mst = mst[np.argsort(mst["distance"])]                                            # sklearn hdbscan.py:_process_mst
    try:
        tree = make_single_linkage(mst)
        labels, prob = tree_to_labels(tree, k, "eom", False, 0.0, None)               # HDBSCAN's defaults (hdbscan.py:846-853)
    except TypeError as err:                                                          # a changed private signature
        import warnings
        from sklearn.cluster import HDBSCAN
        warnings.warn(f"idelucs_amd: sklearn's private tree_to_labels has another signature ({err}); running sklearn.cluster.HDBSCAN on the host")
        cl = HDBSCAN(min_cluster_size=k, min_samples=min(core_neighbour_rank(k), n)).fit(pts)
        return cl.labels_, cl.probabilities_
    if stats is not None:
        stats["tree_s"] = time.time() - t0
    return labels, prob


def core_distances_sharded(latent, device=None):
    """The core distances fine_grained_clusters' device HDBSCAN will need, computed by ALL ranks of the process group (rows of the
    one-pass window kernels split between them, one all-reduce): call on every rank with the same latent, before the ranks other
    than 0 leave.  None when the device path would not be taken (few points, approx / host mode) -- rank 0 then does as before."""
    import torch
    import torch.distributed as tdist
    latent = np.asarray(latent)
    n = len(latent)
    mode = os.environ.get("IDELUCS_HDBSCAN", "device")
    if n <= HDBSCAN_EXACT_MAX or mode == "approx" or not (tdist.is_available() and tdist.is_initialized()) or tdist.get_world_size() < 2:
        return None
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    pts = np.ascontiguousarray(latent, dtype=np.float64)
    x64 = torch.from_numpy(pts).to(dev)
    k = max(n // 100 + 1, 2)
    use_filter = pts.shape[1] % 4 == 0 and pts.shape[1] <= 64 and n >= MST_FILTER_MIN and OPTIONS["mst_filter"] != "0"
    # the memory order hdbscan_device will use (same seed, same groups) -- rank 0's, broadcast: it comes out of a float64 GEMM + argmin,
    # and two ranks whose libraries pick different kernels could place a boundary point in different groups.  Without
    # the filter the window pass builds its own order: the same broadcast
    rank = tdist.get_rank()
    if rank == 0:
        perm, gid = _spatial_order(x64)
    else:
        perm = torch.empty(n, dtype=torch.int64, device=dev)
        gid = torch.empty(n, dtype=torch.int64, device=dev)
    tdist.broadcast(perm, 0)
    tdist.broadcast(gid, 0)
    try:
        core = core_distances_device(x64, min(core_neighbour_rank(k), n), dev, order=(perm, gid), shard=(rank, tdist.get_world_size()))
    except ShardFailed as err:
        import warnings
        warnings.warn(f"idelucs_amd: sharded core distances abandoned ({err}); rank 0 computes them alone")
        return None
    return core.cpu().numpy()


def fine_grained_clusters(latent, exact_max=None, seed=0, device=None, mode=None, core=None):
    """n_clusters=0 mode (reference __main__.py:82-83,153-156): HDBSCAN(min_cluster_size=N//100+1) on the last voter's latent;
    labels+1, probabilities.  `hdbscan` is used when importable, else sklearn.cluster.HDBSCAN (parity with hdbscan==0.8.32 is
    unpinned -- SURVEY 8c).

    Up to `exact_max` points (default HDBSCAN_EXACT_MAX) that is the library call of _hdbscan (the `hdbscan` package on the host as
    in the reference; without it sklearn's HDBSCAN, on the GPU from HDBSCAN_DEVICE_MIN points -- identical results).  Beyond
    it the host algorithm is out of reach (BASELINE cfg5: 10^6 points, each needing its 10^4-th neighbour), and
      mode "device" (default; $IDELUCS_HDBSCAN): the same HDBSCAN with its two O(N^2) stages on the GPU (hdbscan_device: exact
          core distances, sklearn's Prim visit for visit, sklearn's own tree code on the edges) -- minutes at 10^6 points;
      mode "approx": the density clustering runs on a seeded uniform subsample of exact_max points -- min_cluster_size keeps its
          1 % meaning, S//100+1 -- and every other point takes the label of its nearest sampled point (one GEMM-shaped
          nearest-neighbour search on the GPU), with that point's membership probability scaled by how far it is compared with
          that point's own spacing.  Seconds, and an approximation, stated as such."""
    latent = np.asarray(latent)
    n = len(latent)
    exact_max = HDBSCAN_EXACT_MAX if exact_max is None else int(exact_max)
    if n <= exact_max:
        labels, prob = _hdbscan(latent, n // 100 + 1, device=device)
        return labels + 1, prob
    mode = mode or os.environ.get("IDELUCS_HDBSCAN", "device")
    if mode == "host":
        mode = "device"                  # (beyond exact_max the host algorithm is out of reach)
    if mode not in ("device", "approx"):
        raise ValueError("fine_grained_clusters: mode must be 'device' or 'approx'")
    if mode == "device":
        stats = {} if os.environ.get("IDELUCS_TIMING") else None
        labels, prob = hdbscan_device(latent, n // 100 + 1, device=device, stats=stats, core=core)
        if stats is not None:
            stats.pop("mst_edges", None)
            print("HDBSCAN on the device:", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in stats.items()})
        return labels + 1, prob
    import torch
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    rng = np.random.default_rng(seed)
    pick = np.sort(rng.choice(n, size=exact_max, replace=False))
    sub_labels, sub_prob = _hdbscan(latent[pick], exact_max // 100 + 1, device=dev)
    x = torch.from_numpy(latent).to(dev, torch.float32)
    s = x[torch.from_numpy(pick).to(dev)]
    s2 = (s * s).sum(1)
    # spacing of the sample around each sampled point: distance to its nearest other sampled point
    spacing = torch.empty(exact_max, dtype=torch.float32, device=dev)
    for lo in range(0, exact_max, 4096):
        d2 = (s2[lo:lo + 4096, None] + s2[None, :] - 2.0 * (s[lo:lo + 4096] @ s.t())).clamp_min_(0.0)
        d2[torch.arange(d2.shape[0], device=dev), torch.arange(lo, lo + d2.shape[0], device=dev)] = float("inf")
        spacing[lo:lo + 4096] = d2.min(1).values.sqrt_()
    near = torch.empty(n, dtype=torch.int64, device=dev)
    dist = torch.empty(n, dtype=torch.float32, device=dev)
    for lo in range(0, n, 16384):
        xb = x[lo:lo + 16384]
        d2 = ((xb * xb).sum(1)[:, None] + s2[None, :] - 2.0 * (xb @ s.t())).clamp_min_(0.0)
        m = d2.min(1)
        near[lo:lo + 16384] = m.indices
        dist[lo:lo + 16384] = m.values.sqrt_()
    lab_t = torch.from_numpy(sub_labels.astype(np.int64)).to(dev)[near]
    scale = (spacing[near] / dist.clamp_min(1e-30)).clamp_max_(1.0)
    prob_t = torch.from_numpy(sub_prob.astype(np.float32)).to(dev)[near] * scale
    labels = lab_t.cpu().numpy()
    prob = prob_t.double().cpu().numpy()
    labels[pick], prob[pick] = sub_labels, sub_prob            # the sampled points keep exactly what HDBSCAN gave them
    return labels + 1, prob
