"""Worker of tests/test_gpu_knn.py::test_core_distances_sharded_over_two_ranks (started by torch.distributed.run, two ranks sharing
the box's GPU over gloo): the window core-distance pass with its rows split over the ranks must return, on every rank, exactly the
single-rank vector, and the device HDBSCAN that takes them the same labels."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idelucs_amd import posthoc  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(31)
    n = 40000
    centres = rng.normal(size=(5, 64)) * 3.0
    x = (centres[rng.integers(0, 5, n)] + rng.normal(size=(n, 64)) * 0.5).astype(np.float32).astype(np.float64)
    k = n // 100 + 1
    xd = torch.from_numpy(x).to(dev)
    stats = {}
    sharded = posthoc.core_distances_device(xd, k, dev, stats=stats, shard=(rank, world)).cpu().numpy()
    assert stats.get("sample"), "the window path did not run"
    single = posthoc.core_distances_device(xd, k, dev).cpu().numpy()
    assert np.array_equal(sharded, single), f"rank {rank}: sharded core distances differ from the single-rank ones"
    # the whole thing through the entry the CLI uses: core distances by all ranks, HDBSCAN on rank 0 from them
    old = posthoc.HDBSCAN_EXACT_MAX
    posthoc.HDBSCAN_EXACT_MAX = 1000
    try:
        core = posthoc.core_distances_sharded(x, device=dev)
        assert core is not None and np.array_equal(core, single)
        if rank == 0:
            l1, p1 = posthoc.fine_grained_clusters(x, device=dev, core=core)
            l2, p2 = posthoc.fine_grained_clusters(x, device=dev)
            assert np.array_equal(l1, l2) and np.array_equal(p1, p2) and l1.max() >= 3
    finally:
        posthoc.HDBSCAN_EXACT_MAX = old
    dist.barrier()
    if rank == 0:
        print("SHARDED_CORE_OK", int(stats.get("missed", -1)))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
