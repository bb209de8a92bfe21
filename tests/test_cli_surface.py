"""The CLI keeps the reference's flag surface (idelucs/__main__.py:275-308); CPU-only checks plus one GPU run."""
import json
import os
import time

import numpy as np
import pytest

from conftest import DATA


def _free_port():
    """A rendezvous port nobody holds (a fixed one would be blocked by a lingering rank of the previous parametrisation)."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def test_flag_surface_and_defaults():
    from idelucs_amd.__main__ import build_parser
    a = vars(build_parser().parse_args(["--sequence_file", "x.fas"]))
    want = {"sequence_file": "x.fas", "n_clusters": 0, "n_epochs": 100, "n_mimics": 3, "batch_sz": 256, "GT_file": None, "k": 6,
            "optimizer": "RMSprop", "scheduler": "None", "weight": 0.25, "lambda": 2.8, "lr": 1e-3, "n_voters": 5,
            "model_size": "linear", "plot": False}
    for k, v in want.items():
        assert a[k] == v, k


def test_reference_package_name_resolves_to_this_implementation():
    """north_star: 'keeping the idelucs.cluster / __main__ CLI surface'.  `import idelucs`, its sub-modules and `python -m idelucs`
    are the reference's names (idelucs/__init__.py:3-8, pyproject.toml:16) served by idelucs_amd."""
    import importlib
    import idelucs
    import idelucs_amd
    from idelucs.cluster import iDeLUCS_cluster
    from idelucs.utils import kmersFasta, AugmentFasta                 # noqa: F401
    from idelucs.models import IID_model                               # noqa: F401
    from idelucs.LossFunctions import IID_loss, info_nce_loss          # noqa: F401
    assert iDeLUCS_cluster is idelucs_amd.cluster.iDeLUCS_cluster and kmersFasta is idelucs_amd.utils.kmersFasta
    assert importlib.import_module("idelucs.kmers") is idelucs_amd.kmers
    for name in ["check_sequence", "SummaryFasta", "reverse_complement", "kmer_rev_comp", "kmersFasta", "cgrFasta", "cluster_acc",
                 "SequenceDataset", "kmer_counts", "cgr", "IID_model", "IID_loss", "info_nce_loss", "iDeLUCS_cluster"]:
        assert getattr(idelucs, name) is getattr(idelucs_amd, name), name
    main = importlib.import_module("idelucs.__main__").main
    assert main is importlib.import_module("idelucs_amd.__main__").main
    from conftest import ROOT
    assert 'idelucs = "idelucs_amd.__main__:main"' in open(os.path.join(ROOT, "pyproject.toml")).read()


def test_lane_policy_of_a_ranks_voters(monkeypatch):
    """training.voter_lanes: all of a rank's voters in lockstep, up to 8; IDELUCS_VOTER_LANES overrides.  With the lockstep step on fp32
    library GEMMs (IDELUCS_LOCKSTEP_PLANES=0) two voters train one after the other when a lone voter's step takes the two-plane products
    (faster alone than in such a batch of 2: bench.py's predicted_fixed_job)."""
    import types
    import torch
    from idelucs_amd import training
    monkeypatch.delenv("IDELUCS_VOTER_LANES", raising=False)
    monkeypatch.delenv("IDELUCS_PLANES", raising=False)
    monkeypatch.setitem(__import__("idelucs_amd.fused", fromlist=["VARIANTS"]).VARIANTS, "lockstep_planes", "1")

    def model(F, batch_sz, fused=True, H1=512):
        net = types.SimpleNamespace(layers=[torch.nn.Linear(F, H1)])
        return types.SimpleNamespace(net=net, batch_sz=batch_sz, _use_fused=fused)
    cfg2 = model(4096, 512)
    assert training.plane_step_applies(cfg2) and training.plane_step_applies(model(1024, 128))
    assert not training.plane_step_applies(model(256, 512)) and not training.plane_step_applies(model(4096, 48))
    assert not training.plane_step_applies(model(4096, 512, fused=False)) and not training.plane_step_applies(model(4096, 512, H1=256))
    assert [training.voter_lanes(n, cfg2) for n in (1, 2, 3, 4, 8, 11)] == [1, 2, 3, 4, 8, 8]
    monkeypatch.setitem(__import__("idelucs_amd.fused", fromlist=["VARIANTS"]).VARIANTS, "lockstep_planes", "0")
    assert [training.voter_lanes(n, cfg2) for n in (1, 2, 3, 4, 8, 11)] == [1, 1, 3, 4, 8, 8]
    assert [training.voter_lanes(n, model(256, 512)) for n in (1, 2, 4, 8)] == [1, 2, 4, 8]        # k = 4: the fp32 step, lockstep pays
    assert [training.voter_lanes(n) for n in (1, 3, 9)] == [1, 3, 8]
    monkeypatch.setenv("IDELUCS_PLANES", "0")
    assert training.voter_lanes(2, cfg2) == 2
    monkeypatch.delenv("IDELUCS_PLANES")
    monkeypatch.setenv("IDELUCS_VOTER_LANES", "3")
    assert [training.voter_lanes(n, cfg2) for n in (2, 4, 8)] == [2, 3, 3]


def test_relabel_and_ensemble_helpers():
    from idelucs_amd import posthoc
    y = np.array([3, 3, 1, 0, 1, 3, 7])
    assert posthoc.relabel_first_occurrence(y).tolist() == [0, 0, 1, 2, 1, 0, 3]
    rng = np.random.default_rng(0)
    truth = rng.integers(0, 3, 300)
    votes = np.stack([(truth + s) % 3 for s in (0, 1, 2)])       # three voters, same partition, permuted labels
    flip = rng.random(votes.shape) < 0.05                         # (perfect agreement gives 1/0 distances -> NaN
    votes = np.where(flip, (votes + 1) % 3, votes)                #  confidences, in the reference too: utils.py:597-599)
    y, conf = posthoc.label_features(votes, 3)
    from idelucs_amd.utils import cluster_acc
    assert cluster_acc(truth, y)[1] >= 0.98 and conf.shape == (300,) and np.all(conf > 1 / 3)


def test_confusion_matrix_picture(tmp_path):
    """contingency_matrix.jpg of the reference CLI (n_clusters < 16; utils.py:527-577): a figure comes out, the accuracy in its
    x label is the trace share, and oversized tables are replaced by the pointer text."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    from idelucs_amd import posthoc
    cm = np.array([[50, 2, 0], [1, 40, 3], [0, 0, 60]])
    fig, ax = plt.subplots()
    posthoc.plot_confusion_matrix(cm, ["a", "b", "c"], ax=ax)
    assert "accuracy=0.9615" in ax.get_xlabel() and len(ax.texts) == 9 and [t.get_text() for t in ax.get_yticklabels()] == ["a", "b", "c"]
    fig.savefig(tmp_path / "cm.jpg")
    assert os.path.getsize(tmp_path / "cm.jpg") > 1000
    fig, ax = plt.subplots()
    posthoc.plot_confusion_matrix(np.eye(17, dtype=int), [str(i) for i in range(17)], ax=ax)
    assert len(ax.texts) == 1 and "too big" in ax.texts[0].get_text()
    plt.close("all")


@pytest.mark.gpu
def test_cli_end_to_end_writes_reference_outputs(tmp_path, monkeypatch):
    """The run of Example/ALL_RESULTS.tsv:19 (Influenza-A, k=6, 5 clusters, 35 epochs x 5 voters, batch 512): the reference's
    output files, and ensemble accuracies inside the reference's own range for this run -- tests/golden/anchor_seeds.json holds
    TEN 5-voter ensembles of the imported reference (round 3; three before): 0.928 .. 0.995, mean 0.965 (the published 0.9947 is the
    lucky end of it).  Three seeds here (the CLI run + two through the same driver code): every one within 0.03 of the
    reference's worst, their mean within 0.03 of the reference's mean."""
    import json
    import pandas as pd
    from conftest import GOLDEN
    from idelucs_amd.__main__ import main
    ref = [r["acc_ensemble"] for r in json.load(open(os.path.join(GOLDEN, "anchor_seeds.json")))["voters5"]]
    monkeypatch.chdir(tmp_path)
    out_dir = main(["--sequence_file", os.path.join(DATA, "Influenza-A.fas"), "--GT_file", os.path.join(DATA, "Influenza-A_GT.tsv"),
                    "--n_clusters", "5", "--n_epochs", "35", "--n_voters", "5", "--batch_sz", "512", "--k", "6"])
    for f in ("assignments.tsv", "metrics.tsv", "training_plots.jpg", "contingency_matrix.jpg", "contingency_matrix.tsv"):
        assert os.path.exists(os.path.join(out_dir, f)), f
    assert os.path.exists(tmp_path / "ALL_RESULTS.tsv")
    df = pd.read_csv(os.path.join(out_dir, "assignments.tsv"), sep="\t", index_col=0)
    assert list(df.columns) == ["sequence_id", "assignment", "confidence_score"] and len(df) == 949
    m = pd.read_csv(os.path.join(out_dir, "metrics.tsv"), sep="\t", index_col=0)
    assert {"ACC", "ARI", "NMI", "Silhouette-Score", "Davies-Boulding"} <= set(m.index)
    accs = [float(m.loc["ACC", "Value"])]
    for seed in (1, 2):
        time.sleep(1.1)                                       # the results folder is stamped to the second
        od = main(["--sequence_file", os.path.join(DATA, "Influenza-A.fas"), "--GT_file", os.path.join(DATA, "Influenza-A_GT.tsv"),
                   "--n_clusters", "5", "--n_epochs", "35", "--n_voters", "5", "--batch_sz", "512", "--k", "6", "--seed", str(seed)])
        accs.append(float(pd.read_csv(os.path.join(od, "metrics.tsv"), sep="\t", index_col=0).loc["ACC", "Value"]))
    print("5-voter ensemble ACC over 3 seeds", np.round(accs, 4), "| reference ensembles", np.round(ref, 4))
    assert len(ref) >= 10 and min(accs) >= min(ref) - 0.03 and abs(np.mean(accs) - np.mean(ref)) <= 0.03, (accs, np.mean(ref))


@pytest.mark.gpu
def test_cli_fine_grained_mode_n_clusters_0(tmp_path, monkeypatch):
    """--n_clusters 0 (reference __main__.py:75-83,153-156): 200 output units, clusters from HDBSCAN on the last
    voter's latent (sklearn's HDBSCAN stands in for the absent `hdbscan` package), labels shifted by +1."""
    import pandas as pd
    from idelucs_amd.__main__ import main
    monkeypatch.chdir(tmp_path)
    out_dir = main(["--sequence_file", os.path.join(DATA, "Influenza-A.fas"), "--GT_file", os.path.join(DATA, "Influenza-A_GT.tsv"),
                    "--n_clusters", "0", "--n_epochs", "8", "--n_voters", "2", "--batch_sz", "512", "--k", "6"])
    df = pd.read_csv(os.path.join(out_dir, "assignments.tsv"), sep="\t", index_col=0)
    assert len(df) == 949 and df["assignment"].min() >= 0 and df["assignment"].nunique() >= 2
    assert ((df["confidence_score"] >= 0) & (df["confidence_score"] <= 1)).all()
    m = pd.read_csv(os.path.join(out_dir, "metrics.tsv"), sep="\t", index_col=0)
    assert "ACC" in m.index and os.path.exists(os.path.join(out_dir, "contingency_matrix.tsv"))


@pytest.mark.gpu
def test_device_ensemble_matches_sklearn_partition():
    """label_features on the GPU vs the sklearn restatement of the reference (utils.py:582-602): the KMeans there is
    unseeded, so the bar is partition-level -- ARI >= 0.98 between the two and against the planted truth."""
    import time
    import torch
    from sklearn.metrics import adjusted_rand_score
    from idelucs_amd import posthoc
    rng = np.random.default_rng(4)
    n, C, V = 20000, 6, 5
    truth = rng.integers(0, C, n)
    votes = []
    for v in range(V):
        perm = rng.permutation(C)
        y = perm[truth]
        flip = rng.random(n) < 0.08
        y = np.where(flip, rng.integers(0, C, n), y)
        votes.append(posthoc.relabel_first_occurrence(y))
    votes = np.stack(votes)
    t0 = time.time(); y_sk, conf_sk = posthoc.label_features(votes, C); t_sk = time.time() - t0
    torch.cuda.synchronize(); t0 = time.time()
    y_dev, conf_dev = posthoc.label_features_device(votes, C, seed=1); t_dev = time.time() - t0
    print(f"ensemble of {V} voters x {n}: sklearn {t_sk:.2f} s, device {t_dev:.2f} s")
    assert adjusted_rand_score(truth, y_dev) >= 0.98 and adjusted_rand_score(y_sk, y_dev) >= 0.98
    assert conf_dev.shape == (n,) and np.all(conf_dev[np.isfinite(conf_dev)] > 1.0 / C - 1e-6)
    agree = (np.abs(conf_dev - conf_sk) < 0.05) | ~np.isfinite(conf_sk) | ~np.isfinite(conf_dev)
    assert agree.mean() > 0.95


@pytest.mark.gpu
def test_voters_batched_or_one_after_the_other(tmp_path, monkeypatch, capsys):
    """Several voters per GPU train in lockstep as one batch (training.train_voters -> fused.BatchedLinearTrainer).  Every voter
    draws from its own RNG streams and starts from fresh optimizer state, so it is the same run either way -- up to the rounding
    of the batched GEMMs, which training amplifies (a lone voter's layer-1 product runs on the shipped hipBLASLt solution, a
    batch's on the strided-batched heuristic's kernel): the votes of a voter trained in a batch and alone must describe the same
    partition (ARI: every voter >= 0.8, 0.9 on average -- two different voters of this job agree at 0.5-0.8), and the ensembles
    must be equally good."""
    import pandas as pd
    from sklearn.metrics import adjusted_rand_score
    from idelucs_amd.__main__ import main
    monkeypatch.chdir(tmp_path)
    votes, acc = {}, {}
    for lanes in (1, 3, 2):
        monkeypatch.setenv("IDELUCS_VOTER_LANES", str(lanes))
        monkeypatch.setenv("IDELUCS_DUMP_VOTES", str(tmp_path / f"votes{lanes}.npy"))
        out_dir = main(["--sequence_file", os.path.join(DATA, "Influenza-A.fas"), "--GT_file", os.path.join(DATA, "Influenza-A_GT.tsv"),
                        "--n_clusters", "5", "--n_epochs", "12", "--n_voters", "3", "--batch_sz", "512", "--k", "6"])
        # the CLI's own defaults (--scheduler "None", a string as in the reference) must reach the batched path
        said = capsys.readouterr().out
        assert ("Training Models (1-%d/3)" % lanes in said) == (lanes > 1), said[-600:]
        votes[lanes] = np.load(tmp_path / f"votes{lanes}.npy")
        acc[lanes] = float(pd.read_csv(os.path.join(out_dir, "metrics.tsv"), sep="\t", index_col=0).loc["ACC", "Value"])
        time.sleep(1.1)                                       # the results folder is stamped to the second
    assert votes[1].shape == (3, 949)
    assert not np.array_equal(votes[1][0], votes[1][1])
    for lanes in (3, 2):
        ari = [adjusted_rand_score(votes[lanes][v], votes[1][v]) for v in range(3)]
        print(f"{lanes} voters per batch vs one after the other: per-voter ARI {np.round(ari, 3)}, ensemble ACC {acc[lanes]:.4f} vs {acc[1]:.4f}")
        assert min(ari) >= 0.8 and np.mean(ari) >= 0.9 and abs(acc[lanes] - acc[1]) <= 0.05
    # with 2 voters per batch the third voter trains alone, on the single-voter kernels: exactly the sequential run's voter
    assert np.array_equal(votes[2][2], votes[1][2])


@pytest.mark.gpu
@pytest.mark.parametrize("n_clusters,n_voters", [(5, 3), (0, 3)])
def test_cli_two_ranks_voters_are_distinct_and_outputs_complete(tmp_path, n_clusters, n_voters):
    """`python -m torch.distributed.run --nproc-per-node 2 -m idelucs_amd ...`: the voter loop sharded over two ranks (both on
    this box's one GPU, collectives over gloo -- every line of the multi-GPU path except RCCL itself).  Voters on different
    ranks must be DIFFERENT runs (round-1 advice: they were seeded identically), rank 0 writes the reference's outputs, and in
    n_clusters=0 mode the latent comes from predict sharded by sequence + all_gather_rows."""
    import subprocess
    import sys
    import pandas as pd
    from conftest import ROOT
    votes = str(tmp_path / "votes.npy")
    # (voters one after the other on each rank: trained alone a voter is EXACTLY the same run wherever it trains; in a batch of
    # voters only the rounding of the batched GEMMs differs, see test_voters_batched_or_one_after_the_other)
    env = dict(os.environ, IDELUCS_BENCH_BACKEND="gloo", IDELUCS_BENCH_DEVICES="1", IDELUCS_DUMP_VOTES=votes, PYTHONPATH=ROOT,
               HSA_ENABLE_IPC_MODE_LEGACY="0", IDELUCS_VOTER_LANES="1" if n_clusters == 5 else "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "-m", "idelucs_amd", "--sequence_file", os.path.join(DATA, "Influenza-A.fas"),
           "--GT_file", os.path.join(DATA, "Influenza-A_GT.tsv"), "--n_clusters", str(n_clusters), "--n_epochs", "8",
           "--n_voters", str(n_voters), "--batch_sz", "512", "--k", "6"]
    r = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    v = np.load(votes)
    assert v.shape == (n_voters, 949)
    assert not np.array_equal(v[0], v[1]), "voters 0 and 1 (ranks 0 and 1) produced identical assignments"
    res = [d for d in (tmp_path / "Results" / "Influenza-A").iterdir()]
    assert len(res) == 1
    for f in ("assignments.tsv", "metrics.tsv", "training_plots.jpg", "contingency_matrix.tsv"):
        assert (res[0] / f).exists(), f
    df = pd.read_csv(res[0] / "assignments.tsv", sep="\t", index_col=0)
    assert len(df) == 949
    m = pd.read_csv(res[0] / "metrics.tsv", sep="\t", index_col=0)
    assert np.isfinite(float(m.loc["Silhouette-Score", "Value"]))        # computed on the gathered [N, 64] latent
    if n_clusters == 5:
        assert float(m.loc["ACC", "Value"]) > 0.80
        # voter v is the same run whichever rank trains it: the same job in ONE process gives the same vote matrix
        from idelucs_amd.__main__ import main
        one = str(tmp_path / "votes_one_process.npy")
        os.environ["IDELUCS_DUMP_VOTES"] = one
        os.environ["IDELUCS_VOTER_LANES"] = "1"
        cwd = os.getcwd()
        try:
            os.chdir(tmp_path)
            time.sleep(1.1)
            main(cmd[cmd.index("idelucs_amd") + 1:])
        finally:
            os.chdir(cwd)
            del os.environ["IDELUCS_DUMP_VOTES"], os.environ["IDELUCS_VOTER_LANES"]
        assert np.array_equal(np.load(one), v), (np.load(one) == v).mean(axis=1)


@pytest.mark.gpu
def test_silhouette_on_device_equals_sklearn():
    """posthoc.silhouette_score_device (two GEMMs per row block, used above 20 000 points where sklearn's O(N^2) host loop is
    impractical) against sklearn.metrics.silhouette_score, including a singleton cluster and noise-like label values."""
    from sklearn.metrics import silhouette_score
    from idelucs_amd import posthoc
    rng = np.random.default_rng(2)
    centres = rng.normal(size=(5, 64)) * 2.0
    lab = rng.integers(0, 5, 3000)
    x = centres[lab] + rng.normal(size=(3000, 64))
    lab[7] = 9                                                   # a singleton cluster scores 0
    got = posthoc.silhouette_score_device(x, lab, block=512)
    want = silhouette_score(x, lab)
    assert abs(got - want) < 2e-5, (got, want)
    with pytest.raises(ValueError):
        posthoc.silhouette_score_device(x, np.zeros(3000, int))


@pytest.mark.gpu
def test_silhouette_in_one_pass_equals_sklearn(monkeypatch):
    """The one-pass kernel (idl_silhouette_sums: per-cluster distance sums in registers, clusters padded to whole tiles) against
    sklearn on 6 000 points with cluster sizes that are not multiples of 16, a singleton and tight far-apart clusters (the cfg5
    latent's geometry, where a Gram form in global coordinates loses the within-cluster distances), and against the GEMM path at
    50 000 points."""
    from sklearn.metrics import silhouette_score
    from idelucs_amd import posthoc
    rng = np.random.default_rng(4)
    monkeypatch.setattr(posthoc, "SILHOUETTE_ONE_PASS_MIN", 0)
    for spread, scale in ((1.0, 2.0), (0.02, 40.0)):
        centres = rng.normal(size=(7, 64)) * scale
        lab = rng.integers(0, 7, 6000)
        x = (centres[lab] + rng.normal(size=(6000, 64)) * spread).astype(np.float32).astype(np.float64)
        lab[11] = 12                                             # a singleton cluster scores 0
        got, want = posthoc.silhouette_score_device(x, lab), silhouette_score(x, lab)
        assert abs(got - want) < 2e-5, (spread, got, want)
    centres = rng.normal(size=(9, 64)) * 3.0
    lab = rng.integers(0, 9, 50000)
    x = centres[lab] + rng.normal(size=(50000, 64))
    one = posthoc.silhouette_score_device(x, lab)
    monkeypatch.setitem(__import__("idelucs_amd.posthoc", fromlist=["OPTIONS"]).OPTIONS, "silhouette", "gemm")
    two = posthoc.silhouette_score_device(x, lab)
    assert abs(one - two) < 2e-5, (one, two)


@pytest.mark.gpu
def test_fine_grained_clusters_beyond_the_exact_limit():
    """n_clusters = 0 above HDBSCAN_EXACT_MAX points, mode "approx": density clustering of a seeded subsample on the host +
    nearest-sampled-point assignment on the GPU.  On planted blobs the partition is recovered, the sampled points keep HDBSCAN's own labels, and below
    the limit the function is the plain HDBSCAN call."""
    from sklearn.metrics import adjusted_rand_score
    from idelucs_amd import posthoc
    rng = np.random.default_rng(3)
    centres = rng.normal(size=(6, 64)) * 3.0
    truth = rng.integers(0, 6, 60000)
    x = (centres[truth] + rng.normal(size=(60000, 64)) * 0.6).astype(np.float64)
    y, p = posthoc.fine_grained_clusters(x, exact_max=6000, seed=1, mode="approx")
    assert y.shape == (60000,) and p.shape == (60000,) and y.min() >= 0 and np.all((p >= 0) & (p <= 1))
    keep = y > 0                                                 # label 0 = HDBSCAN noise (+1 shift, reference __main__.py:155)
    assert keep.mean() > 0.9 and adjusted_rand_score(truth[keep], y[keep]) > 0.98
    y_small, p_small = posthoc.fine_grained_clusters(x[:1500])                 # below HDBSCAN_DEVICE_MIN: the plain library call
    ref_l, ref_p = posthoc._hdbscan(x[:1500], 1500 // 100 + 1)
    assert np.array_equal(y_small, ref_l + 1) and np.array_equal(p_small, ref_p)
    from sklearn.cluster import HDBSCAN                                          # from there on: sklearn's result, computed on the GPU
    ref = HDBSCAN(min_cluster_size=3000 // 100 + 1).fit(x[:3000])
    y_mid, p_mid = posthoc.fine_grained_clusters(x[:3000])
    assert np.array_equal(y_mid, ref.labels_ + 1) and np.allclose(p_mid, ref.probabilities_, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("n,d,as_f32", [(3000, 64, True), (5000, 16, False), (12000, 64, True)])
def test_hdbscan_on_the_device_is_sklearns(n, d, as_f32):
    """posthoc.hdbscan_device (core distances by float64 GEMM + radix select, Prim's MST of the mutual-reachability graph as
    one HIP launch per node, then sklearn's own tree code) against sklearn.cluster.HDBSCAN on the same points with the
    reference's min_cluster_size rule N // 100 + 1: the MST weights are sklearn's (sorted, 1e-12), the labels are the same
    partition with the same noise set, the membership probabilities agree."""
    import torch
    from sklearn.cluster import HDBSCAN
    from sklearn.metrics import adjusted_rand_score
    from idelucs_amd import posthoc
    rng = np.random.default_rng(n + d)
    centres = rng.normal(size=(7, d)) * 2.5
    truth = rng.integers(0, 7, n)
    x = centres[truth] + rng.normal(size=(n, d)) * rng.uniform(0.3, 0.9, size=(7,))[truth][:, None]
    x[: n // 20] = rng.uniform(-8, 8, size=(n // 20, d))                   # background noise
    x = x.astype(np.float32).astype(np.float64) if as_f32 else x
    k = n // 100 + 1
    ref = HDBSCAN(min_cluster_size=k).fit(x)
    labels, prob = posthoc.hdbscan_device(x, k)
    assert labels.shape == (n,) and prob.shape == (n,)
    assert np.array_equal(labels < 0, ref.labels_ < 0), "different noise sets"
    assert adjusted_rand_score(ref.labels_, labels) == 1.0
    assert np.allclose(prob, ref.probabilities_, atol=1e-9)
    # the core distances on their own, against sklearn's neighbour search
    from sklearn.neighbors import NearestNeighbors
    want = NearestNeighbors(n_neighbors=k).fit(x).kneighbors(x, k)[0][:, -1]
    got = posthoc.core_distances_device(torch.from_numpy(x).cuda(), k, torch.device("cuda")).cpu().numpy()
    assert np.allclose(got, want, rtol=1e-12, atol=0)


@pytest.mark.gpu
def test_fine_grained_clusters_on_the_device_beyond_the_exact_limit():
    """n_clusters = 0 above HDBSCAN_EXACT_MAX points, default mode: the whole HDBSCAN with its O(N^2) stages on the GPU."""
    from sklearn.metrics import adjusted_rand_score
    from idelucs_amd import posthoc
    rng = np.random.default_rng(5)
    centres = rng.normal(size=(6, 64)) * 3.0
    truth = rng.integers(0, 6, 40000)
    x = (centres[truth] + rng.normal(size=(40000, 64)) * 0.6).astype(np.float32).astype(np.float64)
    t0 = time.time()
    y, p = posthoc.fine_grained_clusters(x)
    print(f"HDBSCAN of 40 000 x 64 on the device: {time.time() - t0:.1f} s")
    assert y.shape == (40000,) and y.min() >= 0 and np.all((p >= 0) & (p <= 1))
    keep = y > 0
    assert keep.mean() > 0.9 and adjusted_rand_score(truth[keep], y[keep]) > 0.99 and len(np.unique(y[keep])) == 6


def _write_family_fasta(path, gt_path, n, L, n_families, seed):
    """n sequences of L bases: n_families random ancestors, every member an independent 3 % point-mutated copy."""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    anc = rng.integers(0, 4, size=(n_families, L), dtype=np.uint8)
    fam = rng.integers(0, n_families, n)
    with open(path, "wb") as f, open(gt_path, "w") as g:
        g.write("sequence_id\tcluster_id\n")
        for lo in range(0, n, 10000):
            hi = min(lo + 10000, n)
            codes = anc[fam[lo:hi]]
            mut = rng.random(codes.shape) < 0.03
            codes = np.where(mut, (codes + rng.integers(1, 4, size=codes.shape, dtype=np.uint8)) & 3, codes)
            rec = np.empty((hi - lo, 11 + L + 1), np.uint8)
            rec[:, :11] = np.frombuffer(b"".join(b">seq%06d\n" % i for i in range(lo, hi)), np.uint8).reshape(hi - lo, 11)
            rec[:, 11:-1] = acgt[codes]
            rec[:, -1] = 10
            f.write(rec.tobytes())
            g.write("".join("seq%06d\tfam%d\n" % (i, fam[i]) for i in range(lo, hi)))
    return fam


@pytest.mark.gpu
def test_cli_fine_grained_mode_at_cfg5_scale(tmp_path, monkeypatch):
    """BASELINE cfg5's shape at a fifth of its size: 200 000 sequences x 5 kbp (a 1.0 GB FASTA of 8 planted families),
    --n_clusters 0 => 200 output units, fine-grained clusters on the latent (HDBSCAN with its O(N^2) stages on the GPU), metrics with
    the GPU silhouette: the whole FASTA-in / TSV-out path at a size the reference cannot run."""
    import pandas as pd
    from idelucs_amd.__main__ import main
    fas, gt = str(tmp_path / "fam.fas"), str(tmp_path / "fam_GT.tsv")
    _write_family_fasta(fas, gt, 200000, 5000, 8, seed=11)
    monkeypatch.chdir(tmp_path)
    out_dir = main(["--sequence_file", fas, "--GT_file", gt, "--n_clusters", "0", "--n_epochs", "2", "--n_voters", "1",
                    "--batch_sz", "512", "--k", "6"])
    df = pd.read_csv(os.path.join(out_dir, "assignments.tsv"), sep="\t", index_col=0)
    assert len(df) == 200000 and ((df["confidence_score"] >= 0) & (df["confidence_score"] <= 1)).all()
    m = pd.read_csv(os.path.join(out_dir, "metrics.tsv"), sep="\t", index_col=0)
    acc, sil = float(m.loc["ACC", "Value"]), float(m.loc["Silhouette-Score", "Value"])
    print(f"cfg5-scale CLI run: {df['assignment'].nunique()} clusters, ACC {acc:.4f}, silhouette {sil:.4f}")
    assert np.isfinite(sil) and acc > 0.9 and 4 <= df["assignment"].nunique() <= 40


@pytest.mark.gpu
def test_cfg5_at_full_size_hot_path(tmp_path):
    """BASELINE cfg5 at its FULL size on one GPU, through bench.py's own hot path (`--workload cfg5`): 1 000 000 sequences x 5 kbp
    packed in HBM -> device mimic sites -> all 4 views vectorised into the 65.5 GB feature store -> scaler fit -> one whole epoch
    at 200 output units (5 860 optimizer steps of the fine-grained mode's launch sequence) -> the last voter's weights broadcast,
    predict sharded by sequence, the fp32 latent shards [N/G, 64] all-gathered -- through a process group of one over RCCL, the
    collectives the 8-GPU run uses (reference __main__.py:75-83,106-146,153-156).  Size-independent properties at that size
    (bench.HotPath.validate on the store the run timed): every sampled frequency row sums to 1, views differ, each view's
    checksum is N (every row a distribution), finite epoch loss; here: the latent is [10^6, 64], finite, L2-bounded rows, and
    the job runs at more than 5 x 10^5 sequences/s.  (HDBSCAN on the 10^6 latent rows is post-hoc and takes a minute: it runs in
    tools/run_cfg5_cli.py -> profiles/r03_cfg5_cli_full.txt, and at 200 000 rows in the test above.)"""
    import subprocess
    import sys
    from conftest import ROOT
    free = 0
    try:
        import torch
        free = torch.cuda.mem_get_info(0)[0]
    except Exception:
        pass
    if free and free < 150 * 2 ** 30:
        pytest.skip(f"needs ~110 GB of device memory, {free / 2 ** 30:.0f} GB free")
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg5", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline", "--no-e2e"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(line) == 1, r.stdout[-2000:]
    j = json.loads(line[0])
    c = j["config"]
    assert c["n_sequences"] == 1_000_000 and c["seq_len"] == 5_000 and c["optimizer_steps_per_epoch"] == 5860 and "cfg5" in c["workload"]
    assert j["ranks"] == 1 and j["backend"] == "nccl"                       # the exchange went through RCCL
    v = j["validation"]
    assert all(abs(x - 1e6) < 1e3 for x in v["feats_checksum"]) and v["rows_checked"] >= 4096
    assert all(np.isfinite(x) for x in v["epoch_loss_last_step"])
    assert j["validation"].get("latent_shape") == [1_000_000, 64] and j["validation"]["latent_finite"] is True
    assert 0.0 < j["validation"]["latent_row_norm_max"] < 1e4
    if os.environ.get("IDELUCS_TEST_PERF_FLOORS"):      # (a functional test does not fail on a throttled or shared GPU)
        assert j["value"] > 5e5, j["value"]
    print(f"cfg5 full size: {j['value']:.0f} sequences/s, {j['ms_per_step']:.0f} ms per pass; stages {j['stage_ms']}")


@pytest.mark.gpu
def test_cfg3_at_full_size_one_gpu(tmp_path):
    """BASELINE cfg3 -- 100 000 sequences x 10 kbp, n_voters = 8 -- at its FULL size under the driver on the one GPU a test box has
    (the only BASELINE config no driver-run test exercised): `bench.py --gpus 1 --voters 8 --with-predict 1`, i.e.
    BASELINE.md section 3's fixed job with all 8 voters on this rank (batched in lockstep), each one epoch of 586 optimizer steps
    from its own init, predict per voter, and the exchange step -- the all-gather of the int32 assignments -- through the process
    group of one over RCCL (reference idelucs/__main__.py:106-146).  Checked: the gathered votes are [8, 100000] in range, the
    voters differ PAIRWISE (eight different runs), every epoch loss is finite, the timed store's rows are distributions, the
    backend is nccl, and the line says which job its value is."""
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--voters", "8", "--with-predict", "1", "--steps", "1",
                        "--warmup", "1", "--no-cpu-baseline", "--no-e2e"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(line) == 1, r.stdout[-2000:]
    j = json.loads(line[0])
    c = j["config"]
    assert c["n_sequences"] == 100_000 and c["seq_len"] == 10_000 and c["n_voters"] == 8 and c["optimizer_steps_per_epoch"] == 586
    assert "cfg3" in c["workload"] and j["job"].startswith("cfg3: 8 voters")
    assert j["ranks"] == 1 and j["backend"] == "nccl"                       # the exchange went through RCCL
    v = j["validation"]
    assert v["gathered_shape"] == [8, 100_000] and v["voters_pairwise_distinct"] is True
    assert len(v["epoch_loss_last_step"]) == 8 and all(np.isfinite(x) for x in v["epoch_loss_last_step"])
    assert all(abs(x - 1e5) < 1e2 for x in v["feats_checksum"]) and v["rows_checked"] >= 4096
    assert "fixed_job_8_voters" not in j                                    # (that field belongs to the default one-voter line)
    if os.environ.get("IDELUCS_TEST_PERF_FLOORS"):
        assert j["value"] > 8e5, j["value"]                                     # sequences x voters / s
    print(f"cfg3 full size on one GPU: {j['value']:.0f} sequences x voters / s, {j['ms_per_step']:.0f} ms per pass; stages {j['stage_ms']}")
