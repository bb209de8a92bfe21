"""idelucs_amd.posthoc -- what runs ONCE after the hot path, on [V, N] ints / [N, 64] floats
(reference idelucs/utils.py:582-623 and idelucs/__main__.py:129-156).  SURVEY 8(f) rows f2/f3: kept
on the host with sklearn/scipy, as in the reference, on the outputs gathered from the GPUs.
"""
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


def compute_results(y_pred, data, y_true=None):
    """Reference utils.py:606-623."""
    import sklearn.metrics.cluster as metrics
    from .utils import cluster_acc
    d = {"Davies-Boulding": metrics.davies_bouldin_score(data, y_pred),
         "Silhouette-Score": metrics.silhouette_score(data, y_pred)}
    if y_true is None:
        return d, None
    d["NMI"] = metrics.adjusted_mutual_info_score(y_true, y_pred)
    d["ARI"] = metrics.adjusted_rand_score(y_true, y_pred)
    d["Homogeneity"] = metrics.homogeneity_score(y_true, y_pred)
    d["Completeness"] = metrics.completeness_score(y_true, y_pred)
    ind, acc = cluster_acc(y_true, y_pred)
    d["ACC"] = acc
    return d, ind


def fine_grained_clusters(latent):
    """n_clusters=0 mode (reference __main__.py:82-83,153-156): HDBSCAN(min_cluster_size=N//100+1) on the
    last voter's latent; labels+1, probabilities.  `hdbscan` is used when importable, else
    sklearn.cluster.HDBSCAN (parity with hdbscan==0.8.32 is unpinned -- SURVEY 8c)."""
    mcs = len(latent) // 100 + 1
    try:
        import hdbscan
        cl = hdbscan.HDBSCAN(min_cluster_size=mcs, gen_min_span_tree=True, prediction_data=True)
    except ImportError:
        from sklearn.cluster import HDBSCAN
        cl = HDBSCAN(min_cluster_size=max(mcs, 2))
    cl.fit(latent)
    return cl.labels_ + 1, cl.probabilities_
