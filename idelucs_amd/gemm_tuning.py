"""GEMM solution selection for the dense layers of the training step.

The step's seven skinny fp32 GEMMs (K or N = 64, or C x C outputs) are latency-bound; hipBLASLt's
default heuristic runs them in 8-12 us each, while the best rocBLAS/hipBLASLt solution found by
PyTorch's TunableOp needs 5-7 us (measured on MI355X: epoch 120.5 -> 108.1 ms).  This module turns
TunableOp on, seeds it with the solutions tuned for the default shapes (k=6, batch 512, NetLinear,
n_clusters=20 -- idelucs_amd/tunableop_gfx950.csv, valid for this image's PyTorch/rocBLAS/hipBLASLt
versions, re-tuned automatically when TunableOp's validators do not match) and leaves online tuning
enabled for any other shape (a few hundred ms per new GEMM shape, once per process).

IDELUCS_TUNABLEOP=0 disables it; =1 forces tuning on.  By default TUNING is on only for jobs of at least MIN_STEPS optimizer steps:
on the reference's own example (Influenza-A.fas, 949 sequences, the CLI's defaults: 5 voters x 100 epochs = 3 000 steps) tuning
the partial-batch and batched-voter shapes took 6.1 s of an 8.6 s run that needs 2.6 s without it.  Shorter jobs still USE the
shipped solutions for the shapes they cover (the layer-1 product of the default shape: 32.6 us instead of the heuristic's 41) and
run everything else on the library's heuristic.
"""
import atexit
import os
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
SEED_FILE = os.path.join(_HERE, "tunableop_gfx950.csv")
MIN_STEPS = 400000     # tuning the ~8 unseeded shapes of a job costs 4-6 s; tuned they save ~10 us a step
_enabled = False
_tuning = False


DUMP_TO = None          # a path: maybe_enable() writes the tuned solutions there at exit

def _unlink_quietly(path):
    try:
        os.unlink(path)
    except OSError:
        pass


def maybe_enable(total_steps=None):
    global _enabled, _tuning
    mode = os.environ.get("IDELUCS_TUNABLEOP", "auto")
    if mode == "0":
        return False
    tune = mode == "1" or (total_steps is not None and total_steps >= MIN_STEPS)
    try:
        import torch.cuda.tunable as tn
    except ImportError:
        return False
    if _enabled:
        if tune and not _tuning:                              # a longer job in the same process
            tn.tuning_enable(True)
            _tuning = True
        return True
    # one result file per process (ranks of a multi-GPU job must not share one), seeded from the shipped solutions
    dump = DUMP_TO                                        # maintainers (tools/): write the tuned solutions there at exit, to refresh SEED_FILE
    if dump:
        path = dump
        flags = os.O_WRONLY | os.O_CREAT | os.O_TRUNC | getattr(os, "O_NOFOLLOW", 0)
        fd = os.open(path, flags, 0o600)
    else:
        # a fresh 0600 file of an unpredictable name (never an existing path or a symlink), removed at exit
        fd, path = tempfile.mkstemp(prefix="idelucs_tunableop_", suffix=".csv")
        atexit.register(_unlink_quietly, path)
    with os.fdopen(fd, "wb") as out:
        if os.path.exists(SEED_FILE) and os.environ.get("IDELUCS_TUNABLEOP_SEED", "1") != "0":       # (0: tune every shape afresh)
            with open(SEED_FILE, "rb") as src:
                out.write(src.read())
    tn.set_filename(path, insert_device_ordinal=False)
    tn.enable(True)
    tn.tuning_enable(tune)                                    # a short job: the shipped solutions are used, nothing is tuned
    if hasattr(tn, "write_file_on_exit"):
        tn.write_file_on_exit(bool(dump))
    _enabled = True
    _tuning = tune
    return True


def stop_tuning():
    """After the training loop: keep using the solutions found, but do not tune the shapes that follow -- the post-hoc stages
    (silhouette, HDBSCAN core distances) run a handful of [4096 x 10^6]-sized GEMMs, and timing hundreds of candidate kernels on
    each of those shapes cost 230 s at cfg5 (measured: HDBSCAN 169 -> 104 s, metrics 180 -> 10 s without it)."""
    global _tuning
    if not _enabled:
        return
    import torch.cuda.tunable as tn
    tn.tuning_enable(False)
    _tuning = False
