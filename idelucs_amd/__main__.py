#!/usr/bin/env python3
"""`python -m idelucs_amd` -- the reference CLI surface (idelucs/__main__.py:274-318): same flags,
FASTA in, ./Results/<stem>/<timestamp>/{assignments.tsv, metrics.tsv, training_plots.jpg,
contingency_matrix.*} and a row in ./ALL_RESULTS.tsv out.

Launched through `python -m torch.distributed.run --nproc-per-node G -m idelucs_amd ...` the voter
loop (reference __main__.py:106-146, sequential there) is sharded voter v -> rank v mod G, one
process per GPU; the [V, N] assignments are all-gathered over RCCL and rank 0 writes the results.
"""
import argparse
import csv
import os
import sys
import time

import numpy as np


def save_results_in_file(dataset_name, model_name, model_parameters, results, time_, memory, file_name):
    """Reference __main__.py:37-46."""
    with open(file_name, mode="a", newline="") as fh:
        w = csv.writer(fh, delimiter="\t")
        if fh.tell() == 0:
            w.writerow(["Dataset", "Model", "Parameters"] + list(results.keys()) + ["Time", "Memory"])
        w.writerow([dataset_name, model_name, model_parameters] + list(results.values()) + [time_, memory])


def run(args):
    import pandas as pd
    import torch
    import torch.distributed as dist
    from resource import getrusage, RUSAGE_SELF
    from . import gemm_tuning, posthoc, dist as D
    from .training import prepare_model, train_voters

    start_time = time.time()
    marks = [("start", start_time)]                               # stage timers, printed with IDELUCS_TIMING=1

    def mark(name):
        marks.append((name, time.time()))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))

    # the output folder first (reference __main__.py:60-73), BEFORE any collective exists: a rank 0 that cannot create it fails
    # here, alone, instead of leaving the other ranks waiting in a collective
    stamp = time.asctime().split(" ")
    stamp = [s for s in stamp if s]
    stamp[3] = "-".join(stamp[3].split(":"))
    time_stamp = "_".join(stamp[1:-1])
    results_folder = os.path.join(os.getcwd(), "Results", os.path.basename(args["sequence_file"]).split(".")[0])
    out_dir = os.path.join(results_folder, time_stamp)
    if rank == 0:
        os.makedirs(out_dir, exist_ok=False)

    if world > 1:
        local = int(os.environ.get("LOCAL_RANK", "0"))
        n_dev = int(os.environ.get("IDELUCS_BENCH_DEVICES", "0"))    # rehearsal on a box with fewer GPUs than ranks (with gloo)
        torch.cuda.set_device(local % n_dev if n_dev > 0 else local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("IDELUCS_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    use_hdbscan = False
    if args["n_clusters"] == 0:                                   # __main__.py:75-76
        args["n_clusters"], use_hdbscan = 200, True

    model = prepare_model(args)
    mark("FASTA summary + feature store")
    if rank == 0:
        print(model.cluster_dis)
        print(f"No. Sequences: \t {len(model.lengths):,}")
        print(f"Min. Length: \t {np.min(model.lengths):,}")
        print(f"Max. Length: \t {np.max(model.lengths):,}")
        print(f"Avg. Length: \t {round(np.mean(model.lengths), 2):,}")
    n = len(model.names)

    local_preds, curves, latent = {}, {}, None
    # this rank's voters, several at a time on their own HIP streams (training.train_voters)
    trained = train_voters(model, D.voters_of_rank(args["n_voters"], rank, world), args["n_epochs"], args["n_voters"], progress=rank == 0)
    for voter, (curve, y_pred, probabilities, lat) in trained.items():
        curves[voter] = curve
        if voter == args["n_voters"] - 1:
            latent = lat                                          # the reference scores/clusters the LAST voter's latent
        local_preds[voter] = torch.from_numpy(posthoc.relabel_first_occurrence(y_pred)).to(model.device)

    mark("training + predict")
    # nothing below reads the feature store or the predict inputs again: return their memory (65 + 16 GB at cfg5) before the
    # post-hoc stages allocate their own blocks -- left cached, the allocator ends up freeing and re-allocating per block
    model.store = model.dataloader = None
    model._shared.clear()
    if world == 1 or not use_hdbscan:                              # (the sharded predict of n_clusters = 0 reads the file once more)
        from .utils import release_ingest_buffers
        release_ingest_buffers()
    torch.cuda.empty_cache()
    gemm_tuning.stop_tuning()                                     # the GEMM shapes below are one-off and huge: not worth tuning
    preds = D.gather_voter_predictions(local_preds, args["n_voters"], n, device=model.device).cpu().numpy()
    if world > 1:
        owner = (args["n_voters"] - 1) % world
        if use_hdbscan:
            # n_clusters=0 clusters ONE model's latent (reference __main__.py:153-156, the last voter's): its weights go to every
            # rank, predict is sharded by sequence, the fp32 latent shards [N/G, 64] are all-gathered over RCCL
            D.broadcast_parameters(model.net, owner)
            lo, hi = D.shard_bounds(n, rank, world)
            latent = D.all_gather_rows(model.predict_latent_shard(lo, hi), n).double().cpu().numpy()
        else:                                                      # ship the last voter's latent to rank 0 (scores only)
            lat_t = torch.from_numpy(latent).to(model.device) if rank == owner else torch.empty((n, 64), dtype=torch.float64, device=model.device)
            dist.broadcast(lat_t, src=owner)
            latent = lat_t.cpu().numpy()
        all_curves = [None] * world
        dist.all_gather_object(all_curves, curves)
        curves = {k: v for c in all_curves for k, v in c.items()}
    core = None
    if world > 1 and use_hdbscan:
        # the fine-grained mode's post-hoc stage starts on ALL ranks: each takes its rows of the core-distance pass (3 s of the 21 s
        # HDBSCAN at cfg5 on one GPU) before the others leave; Prim and the tree code stay on rank 0
        core = posthoc.core_distances_sharded(latent, device=model.device)
    if rank != 0:
        dist.destroy_process_group()
        return
    if os.environ.get("IDELUCS_DUMP_VOTES"):                      # tests: the [V, N] vote matrix as gathered from the ranks
        np.save(os.environ["IDELUCS_DUMP_VOTES"], preds)

    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    fig, ax = plt.subplots(nrows=1, ncols=1)
    ax.grid(True); ax.set_title("Learning Curves"); ax.set_xlabel("Epoch"); ax.set_ylabel("Training Loss")
    for v in sorted(curves):
        ax.plot(curves[v], label=f"Model {v + 1}")
    ax.legend(loc=1)
    fig.savefig(os.path.join(out_dir, "training_plots.jpg"))

    if not use_hdbscan:
        if posthoc.OPTIONS["ensemble"] == "sklearn":
            y_pred, probabilities = posthoc.label_features(preds, args["n_clusters"])
        else:                                                      # same ensemble on the GPU
            y_pred, probabilities = posthoc.label_features_device(preds, args["n_clusters"], device=model.device, seed=args.get("seed", 0))
    else:
        y_pred, probabilities = posthoc.fine_grained_clusters(latent, core=core)
        args["n_clusters"] = int(np.max(y_pred) + 1)
    mark("ensemble / HDBSCAN")

    sys.stdout.write("\r........... Computing Results ................")
    sys.stdout.flush()
    if args["GT_file"] is not None:
        unique_arr, y = np.unique(np.asarray(model.GT), return_inverse=True)      # reference :160-161, as one sort
        unique_labels = list(unique_arr)
        y = np.asarray(y, dtype=np.int64)
        results, ind = posthoc.compute_results(y_pred, latent, y)
        mark("metrics")
        d = {i: j for i, j in ind}
        if -1 in y_pred:
            d[-1] = 0
        w = np.zeros((len(unique_labels), max(max(y_pred) + 1, max(y) + 1)), dtype=np.int64)
        lut = np.zeros(int(max(d)) + 2, dtype=np.int64)                          # reference :166-169 (w[y[i], d[y_pred[i]]] += 1)
        for a_, b_ in d.items():
            lut[a_] = b_                                                          # (a -1 key lands on the last slot)
        np.add.at(w, (y, lut[np.asarray(y_pred, dtype=np.int64)]), 1)
        if args["n_clusters"] < 16:                               # reference __main__.py:176-180: a picture for small tables ...
            fig, new_ax = plt.subplots(nrows=1, ncols=1)
            posthoc.plot_confusion_matrix(w, unique_labels, ax=new_ax, normalize=False)
            fig.savefig(os.path.join(out_dir, "contingency_matrix.jpg"))
        # ... and the table itself always (the reference writes the .tsv only when n_clusters >= 16, :181-186)
        w_df = pd.DataFrame(w)
        w_df.index = unique_labels
        w_df.to_csv(os.path.join(out_dir, "contingency_matrix.tsv"), sep="\t")
        print(f"ACC: {results['ACC']}")
    else:
        results, ind = posthoc.compute_results(y_pred, latent)
        mark("metrics")

    sys.stdout.write("\r........ Saving Results ..............\n")
    sys.stdout.flush()
    dataset_name = args["sequence_file"].split("/")[-1]
    params = {k: v for k, v in args.items() if k not in ("sequence_file", "GT_file")}
    info = getrusage(RUSAGE_SELF)
    t = time.time() - start_time
    hh, mm = int(t // 3600), int((t % 3600) // 60)
    ss = t - 3600 * hh - 60 * mm
    memory = info.ru_maxrss / 1e6
    print(f"training took: {hh}:{mm}:{round(ss)} (hh:mm:ss) and {memory} (GB)")

    names = np.array(model.names)
    data = np.concatenate((names[:, np.newaxis], y_pred[:, np.newaxis], probabilities[:, np.newaxis]), axis=1)
    pd.DataFrame(data, columns=["sequence_id", "assignment", "confidence_score"]).to_csv(
        os.path.join(out_dir, "assignments.tsv"), sep="\t")
    pd.Series(results, name="Value").to_csv(os.path.join(out_dir, "metrics.tsv"), sep="\t")
    save_results_in_file(dataset_name, "iDeLUCS", params, results, f"{hh}:{mm}:{round(ss)}", memory,
                         os.path.join(os.getcwd(), "ALL_RESULTS.tsv"))
    mark("tables")
    if os.environ.get("IDELUCS_TIMING"):
        print("stages: " + ", ".join(f"{name} {t1 - t0:.1f} s" for (_, t0), (name, t1) in zip(marks[:-1], marks[1:])))

    if args.get("plot"):
        try:
            import umap
        except ImportError:
            print("--plot needs the `umap-learn` package (reference __main__.py:239); skipping the plot")
        else:
            emb = umap.UMAP(random_state=42).fit_transform(latent)
            fig, ax = plt.subplots(nrows=1, ncols=1)
            ax.set_title("Representation of the Latent Space"); ax.set_xlabel("UMAP 1"); ax.set_ylabel("UMAP 2")
            ax.scatter(emb[:, 0], emb[:, 1], c=y_pred, s=1, alpha=0.5)
            fig.savefig(os.path.join(out_dir, "learned_representation.jpg"), dpi=150)
    if world > 1:
        dist.destroy_process_group()
    return out_dir


def build_parser():
    """Flag surface of reference __main__.py:275-308 (names, types, defaults)."""
    p = argparse.ArgumentParser(prog="idelucs_amd")
    p.add_argument("--sequence_file", action="store", type=str)
    p.add_argument("--n_clusters", action="store", type=int, default=0,
                   help="Expected or maximum number of clusters; 0 = fine-grained clusters found automatically")
    p.add_argument("--n_epochs", action="store", type=int, default=100)
    p.add_argument("--n_mimics", action="store", type=int, default=3)
    p.add_argument("--batch_sz", action="store", type=int, default=256)
    p.add_argument("--GT_file", action="store", type=str, default=None)
    p.add_argument("--k", action="store", type=int, default=6, help="k-mer length")
    p.add_argument("--optimizer", action="store", type=str, default="RMSprop")
    p.add_argument("--scheduler", action="store", type=str, default="None")
    p.add_argument("--weight", action="store", type=float, default=0.25)
    p.add_argument("--lambda", action="store", type=float, default=2.8)
    p.add_argument("--lr", action="store", type=float, default=1e-3, help="Learning Rate")
    p.add_argument("--n_voters", action="store", type=int, default=5, help="Number of Voters")
    p.add_argument("--model_size", action="store", type=str, default="linear", help="'small' or 'linear'")
    p.add_argument("--plot", action="store", type=bool, default=False)
    # additions (not in the reference)
    p.add_argument("--rng", action="store", type=str, default=None, choices=[None, "philox", "compat"],
                   help="mimic RNG: device Philox (default) or the reference's host numpy/random streams")
    p.add_argument("--seed", action="store", type=int, default=0)
    return p


def main(argv=None):
    args = vars(build_parser().parse_args(argv))
    if int(os.environ.get("RANK", "0")) == 0:
        print("\nTraining Parameters:")
        for key in args:
            print(f"{key} \t -> {args[key]}")
    return run(args)


if __name__ == "__main__":
    main()
