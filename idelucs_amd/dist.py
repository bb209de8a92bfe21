"""idelucs_amd.dist -- the multi-GPU shape of the hot path: the reference's sequential n_voters loop
(idelucs/__main__.py:106-146, cluster.py:40-50) sharded one-voter-per-GPU, one process per GPU
(torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The voters are independent (fresh Kaiming init each, __main__.py:109) and share only the read-only
feature store, which every rank rebuilds locally from the same packed input (vectorising is ~1 ms per
100k sequences; shipping 6.5 GB of features over xGMI would cost more than recomputing them).  So the
training epochs contain NO collective; the only exchange is at the end:
  * all-gather of the per-voter int32 assignments  [N] -> [V, N]   (feeds label_features, utils.py:582)
  * all-gather of fp32 latent shards [N/G, 64] -> [N, 64]          (n_clusters=0 / HDBSCAN mode)
Payloads are <= 0.4 MB and 32 MB per rank: latency-bound, one collective each.
"""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def voters_of_rank(n_voters, rank=None, world_size=None):
    """Voter v runs on rank v mod G (round-robin), in ascending order on each rank."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    return [v for v in range(n_voters) if v % world_size == rank]


def _no_group():
    """True when no process group exists: the single-process library / CLI use.  A group of ONE rank still goes through the
    collective (RCCL on a GPU box), so the N = 1 run exercises the same calls, dtypes and shapes as the N > 1 run."""
    return not (dist.is_available() and dist.is_initialized())


def all_gather_assignments(y_pred):
    """[N] int32 (this rank's voter) -> [G, N] on every rank.  One collective."""
    _, w = world()
    y = y_pred.contiguous()
    if _no_group():
        return y.unsqueeze(0)
    out = torch.empty(w * y.numel(), dtype=y.dtype, device=y.device)     # flat in / flat out: valid on nccl and gloo
    dist.all_gather_into_tensor(out, y.view(-1))
    return out.view((w,) + tuple(y.shape))


def gather_voter_predictions(local_preds, n_voters, n_items, device="cpu", dtype=torch.int32):
    """local_preds: {voter index: [n_items] int tensor} for the voters this rank trained (possibly
    none).  Returns the [n_voters, n_items] matrix on every rank, rows in voter order (ranks may own
    different numbers of voters: rounds of all-gather, padded with -1 rows that are dropped)."""
    r, w = world()
    if _no_group():
        return torch.stack([local_preds[v].to(dtype) for v in range(n_voters)])
    mine = voters_of_rank(n_voters, r, w)
    rounds = (n_voters + w - 1) // w
    rows = [None] * n_voters
    for i in range(rounds):
        y = local_preds[mine[i]].to(dtype) if i < len(mine) else torch.full((n_items,), -1, dtype=dtype, device=device)
        g = all_gather_assignments(y)
        for rk in range(w):
            v = i * w + rk
            if v < n_voters:
                rows[v] = g[rk]
    return torch.stack(rows)


def shard_bounds(n, rank=None, world_size=None):
    """Contiguous row shard [lo, hi) of N sequences for sharded predict (HDBSCAN mode)."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    per = (n + world_size - 1) // world_size
    return min(n, rank * per), min(n, (rank + 1) * per)


def all_gather_rows(x, n):
    """Row shards (shard_bounds) of an [N, d] matrix -> full [N, d] on every rank (padded all-gather)."""
    r, w = world()
    if _no_group():
        return x
    per = (n + w - 1) // w
    pad = torch.zeros((per,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    pad[:x.shape[0]] = x
    out = torch.empty(w * pad.numel(), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, pad.view(-1))
    return out.view((w * per,) + tuple(x.shape[1:]))[:n]


def broadcast_parameters(module, src):
    """The owner's weights to every rank (n_clusters = 0 mode: ONE model's latent is clustered, reference __main__.py:153-156).
    One broadcast per tensor; nothing to do without a process group."""
    if _no_group():
        return
    for p in module.parameters():
        dist.broadcast(p.data, src=src)
