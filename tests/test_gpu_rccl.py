"""RCCL itself, on the one GPU of the test box: a process group of ONE rank over backend "nccl" (== RCCL on ROCm) pushes every
payload of the path's exchange step through idelucs_amd.dist -- int32 assignments [N] -> [G, N], the rounds of
gather_voter_predictions, fp32 latent shards [N/G, 64] -> [N, 64], the weight broadcast of n_clusters = 0 mode -- so librccl is
loaded and every dtype / shape the N > 1 run will hand it has been through it once (reference loop being sharded:
idelucs/__main__.py:106-146, :153-156).  A group of one is not a shortcut in idelucs_amd.dist (dist._no_group)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


@pytest.fixture()
def rccl_group_of_one():
    from idelucs_amd import _lib
    _lib.require_gpu()
    torch.cuda.set_device(0)
    assert not dist.is_initialized()
    dist.init_process_group("nccl", rank=0, world_size=1, store=dist.HashStore(), device_id=torch.device("cuda", 0))
    try:
        yield torch.device("cuda", 0)
    finally:
        dist.destroy_process_group()


def _rccl_loaded():
    with open("/proc/self/maps") as f:
        return any("librccl" in line for line in f)


def test_exchange_payloads_through_rccl(rccl_group_of_one):
    dev = rccl_group_of_one
    from idelucs_amd import dist as D
    from idelucs_amd.PytorchUtils import NetLinear
    assert dist.get_backend() == "nccl" and D.world() == (0, 1) and not D._no_group()
    n = 100_000
    g = torch.Generator(device=dev); g.manual_seed(5)
    # (i) the int32 assignments of one voter: [N] -> [G, N]
    y = torch.randint(0, 20, (n,), dtype=torch.int32, device=dev, generator=g)
    got = D.all_gather_assignments(y)
    assert got.shape == (1, n) and got.dtype == torch.int32 and torch.equal(got[0], y)
    assert got.data_ptr() != y.data_ptr(), "the group of one took the local shortcut: RCCL did not run"
    # (ii) three voters on this rank: rounds of all-gather, rows in voter order
    preds = {v: torch.randint(0, 20, (n,), dtype=torch.int32, device=dev, generator=g) for v in range(3)}
    allp = D.gather_voter_predictions(preds, 3, n, device=dev)
    assert allp.shape == (3, n) and all(torch.equal(allp[v], preds[v]) for v in range(3))
    # (iii) fp32 latent shards: cfg5's [N/G, 64] at G = 8 and the whole [N, 64] of a group of one
    for rows, total in ((125_000, 125_000), (n, n), (0, 0)):
        lat = torch.randn((rows, 64), device=dev, generator=g)
        full = D.all_gather_rows(lat, total)
        assert full.shape == (total, 64) and full.dtype == torch.float32 and torch.equal(full, lat)
        if rows:
            assert full.data_ptr() != lat.data_ptr()
    # (iv) the weight broadcast of n_clusters = 0 mode (200 output units)
    net = NetLinear(4096, 200).to(dev)
    before = [p.detach().clone() for p in net.parameters()]
    D.broadcast_parameters(net, 0)
    torch.cuda.synchronize()
    assert all(torch.equal(a, p) for a, p in zip(before, net.parameters()))
    # the barrier / max-over-ranks reduction bench.py brackets its timed region with
    t = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t.item()) == 1.25
    assert _rccl_loaded(), "librccl is not mapped into the process: backend nccl did not load RCCL"


def test_no_group_is_the_local_path():
    """Without a process group (library / CLI on one GPU) the same calls stay local."""
    from idelucs_amd import dist as D
    assert not dist.is_initialized() and D._no_group()
    y = torch.arange(7, dtype=torch.int32, device="cuda")
    assert D.all_gather_assignments(y).shape == (1, 7)
    x = torch.ones((5, 64), device="cuda")
    assert D.all_gather_rows(x, 5) is x


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher: the parent (which never touches the GPU) starts two ranks as a child torchrun,
    relays rank 0's ONE JSON line and its exit code.  Two ranks share this box's one GPU over gloo (the rehearsal knobs
    IDELUCS_BENCH_BACKEND / IDELUCS_BENCH_DEVICES): every line of the N > 1 bench path except RCCL itself, which
    test_exchange_payloads_through_rccl covers.  `--n` must survive the launcher's own argument parser."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(IDELUCS_BENCH_BACKEND="gloo", IDELUCS_BENCH_DEVICES="1", PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--n", "5000", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith('{"metric"'), r.stdout[-1000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks"] == 2 and j["backend"] == "gloo" and [d["rank"] for d in j["devices"]] == [0, 1]
    assert j["config"]["n_voters"] == 8 and j["config"]["n_sequences"] == 5000 and j["scaling"] == "strong"
    # two processes time-slicing one GPU must still be within an order of magnitude of one (a gloo collective on a device tensor
    # right after init once left both ranks 100 x slower for the whole run)
    assert j["ms_per_step"] < 1000, j["ms_per_step"]
