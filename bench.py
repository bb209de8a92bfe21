#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json on MI355X.

metric   : sequences/sec (k-mer vectorise + 1 epoch), k=6, batch 512
workload : BASELINE.json configs[1] -- synthetic 100 000 sequences x 10 kbp (iid uniform ACGT),
           k=6, n_clusters=20, n_mimics=3 (CLI default), batch_sz=512, NetLinear, RMSprop, fp32.
step     : ONE pass of the hot path over the whole synthetic batch, with the packed bases already
           resident in HBM: device mimic-site generation + vectorisation of all 4 views + scaler fit
           + one full training epoch (586 optimizer steps), i.e. BASELINE.md section 3's timed region
           ("... to the optimizer.step() of the last batch of epoch 1; excludes predict, ensemble, TSV").
           Nothing is cached across steps (features are recomputed, weights re-initialised each step
           -- one step == one voter of the reference's voter loop, idelucs/__main__.py:106-146).
           --with-predict 1 adds predict (+ the all-gather) to the timed region.
N > 1    : one process per GPU (torch.distributed, backend nccl == RCCL); every rank runs its own
           voter on the full data set (the n_voters loop sharded over GPUs; weak scaling, no
           data-path collective inside the epoch).  value = (N_seq x ranks) / max-over-ranks time.
           The path's one exchange step -- predict + all-gather of the int32 assignments [V, N] over
           RCCL -- runs once after the timed loop at every N (reported as "exchange_ms").

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (the hand-written vectoriser kernel, HBM
bound), "roofline_epoch" (the encoder epoch, MFMA fp32 bound), "cpu_baseline" (oracle port on the
host cores, bounded sample, rank 0 at N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 matrix peak (v_mfma_f32_32x32x2_f32)


def synth_packed(n, L, dev, seed=12345):
    """iid-uniform ACGT == iid-uniform 2-bit codes, generated directly in the packed slot layout."""
    slots = (L + 63) // 64
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    codes = torch.randint(-2 ** 31, 2 ** 31 - 1, (n * slots * 4,), dtype=torch.int32, device=dev, generator=g)
    mask = torch.zeros((n, slots, 2), dtype=torch.int32, device=dev)
    tail = L % 64
    if tail:                     # bases past the end of the sequence are marked invalid
        w = [0, 0]
        for j in range(tail, 64):
            w[j // 32] |= 1 << (31 - (j % 32))
        for i in (0, 1):
            mask[:, -1, i] = w[i] - (1 << 32) if w[i] >= 2 ** 31 else w[i]

    class DeviceInput:
        pass
    d = DeviceInput()
    d.n, d.codes, d.mask, d.min_len = n, codes, mask.view(-1), L
    d.slot_off = torch.arange(0, (n + 1) * slots, slots, dtype=torch.int64, device=dev)
    d.lengths = torch.full((n,), L, dtype=torch.int64, device=dev)
    d.max_len = L
    return d


class HotPath:
    """The timed region: everything from packed bases in HBM to the last optimizer.step()."""

    def __init__(self, din, args, dev):
        from idelucs_amd import _lib, utils as U, models
        self.U, self._lib, self.models = U, _lib, models
        self.din, self.dev, self.a = din, dev, args
        self.specs = [t.spec() for t in U.mimic_transforms(args.n_mimics)]
        self.P = len(self.specs)
        self.F = 4 ** args.k
        self.model = models.IID_model({
            'sequence_file': None, 'GT_file': None, 'n_clusters': args.n_clusters, 'k': args.k, 'model_size': 'linear',
            'n_mimics': args.n_mimics, 'batch_sz': args.batch_sz, 'optimizer': 'RMSprop', 'lambda': 2.8, 'lr': 1e-3,
            'weight': 0.25, 'scheduler': None, 'n_epochs': 1, 'n_voters': 1})
        self.feats = torch.empty((self.P, din.n, self.F), dtype=torch.float32, device=dev)   # 6.55 GB at cfg2
        # mimic-site buffer sized by a safe bound (expected + 25 % + slack) instead of a host round trip per step; the
        # device-side totals are checked against it after the run (overflow flag, no sync inside the step)
        expect = sum(din.n * (args.len * (1.0 - (1.0 - s[0]) * (1.0 - s[1])) + s[2]) for s in self.specs)
        self.edit_capacity = int(1.25 * expect + 64 * din.n * self.P + 1024)
        self.edit_overflow = torch.zeros((), dtype=torch.bool, device=dev)
        self.ev = {k: [] for k in ("edits", "vectorise", "stats", "epoch", "predict")}
        self.y_pred = None

    def _timed(self, key, fn):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = fn()
        e.record()
        self.ev[key].append((s, e))
        return r

    def step(self, seed):
        U, _lib, m = self.U, self._lib, self.model
        edits, edit_off = self._timed("edits", lambda: U._philox_edits(self.din, self.specs, seed, capacity=self.edit_capacity))
        self.edit_overflow |= edit_off[-1] > self.edit_capacity
        self._timed("vectorise", lambda: U._vectorise(self.din, self.a.k, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_FREQ_F32,
                                                      self.P, edits, edit_off, self.feats))
        mean, scale = self._timed("stats", lambda: U.col_stats(self.feats[0]))
        m.store = U.FeatureStore(None, None, self.feats, mean, scale, self.a.k, False)
        m.net.apply(self.models.weights_init)          # a fresh voter (reference __main__.py:109)
        loss = self._timed("epoch", m.contrastive_training_epoch)
        if self.a.with_predict:
            self.y_pred = self._timed("predict", self.predict)
        return loss

    def predict(self):
        """Un-mutated vectors, own float64 scaler, eval forward, argmax (reference models.py:145-172)."""
        U, _lib, m = self.U, self._lib, self.model
        f64 = U._vectorise(self.din, self.a.k, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_FREQ_F64)[0]
        mean, scale = U.col_stats(f64)
        x = U.standardise(f64, mean, scale)
        del f64
        preds = []
        with torch.no_grad():
            m.net.eval()
            for i in range(0, x.shape[0], 8192):
                o, _ = m.net(x[i:i + 8192])
                preds.append(o.argmax(1).to(torch.int32))
        return torch.cat(preds)

    def mean_ms(self, key, skip):
        ev = self.ev[key][skip:]
        return sum(s.elapsed_time(e) for s, e in ev) / max(len(ev), 1)


def pmc_traffic_gb():
    """HBM traffic of one vectorise launch at cfg2, GB, from the committed rocprofv3 PMC passes
    (profiles/r01_k_vectorise_pmc.json, re-collected this round: WRITE_SIZE exact for 16-B stores, FETCH_SIZE as reported -- see DESIGN.md 4.1);
    None when the profile is not present or the workload is not cfg2."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_k_vectorise_pmc.json")) as fh:
            return json.load(fh)["traffic_gb_per_launch"]
    except Exception:
        return None


def cpu_baseline(args):
    """The oracle (CPU restatement of the reference algorithm) timed on this box's host cores, in the
    reference's shape: single-threaded per-record vectorise passes with numpy/random mimics
    (idelucs/utils.py:224-368), StandardScaler, then one PyTorch-CPU epoch with cores-2 threads
    (idelucs/__main__.py:316).  Bounded sample of the same workload; seq/s = sample / wall."""
    import random
    from oracle import oracle as O
    import torch.nn as nn
    from idelucs_amd.PytorchUtils import NetLinear
    from idelucs_amd.LossFunctions import IID_loss, info_nce_loss
    n, L, k = args.cpu_sample, args.len, args.k
    rng = np.random.default_rng(12345)
    seqs = [bytearray(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=L).tobytes()) for _ in range(n)]
    cores = os.cpu_count() or 1
    # the reference uses cpu_count()-2 torch threads (__main__.py:316); on a 256-core host that
    # oversubscribes these small GEMMs ~50x slower than 32 threads, so give the CPU its best shot
    threads = max(1, min(cores - 2, 32))
    torch.set_num_threads(threads)
    np.random.seed(0); random.seed(0)
    t0 = time.perf_counter()
    tfs = [O.transition_transversion(1e-2, 0.5e-2), O.transition(1e-2), O.transversion(0.5e-2)] + \
          [O.Random_N(20) for _ in range(args.n_mimics - 2)]
    views = []
    for tf in tfs:
        rows = []
        for s in seqs:
            b = bytearray(s)
            tf(b)
            c = np.ones(4 ** k, np.int32)
            O.kmer_counts(b, k, c)
            rows.append(c / np.sum(c))
        views.append(np.array(rows))
    t_norm = views[0]
    x = np.empty(((len(tfs) - 1) * n, 2, 4 ** k), np.float32)
    for m_, v in enumerate(views[1:]):
        x[m_ * n:(m_ + 1) * n, 0] = t_norm
        x[m_ * n:(m_ + 1) * n, 1] = v
    mean, scale = O.scaler_fit(t_norm.astype(np.float32))
    x[:, 0] = O.scaler_transform(x[:, 0], mean, scale)
    x[:, 1] = O.scaler_transform(x[:, 1], mean, scale)
    t_vec = time.perf_counter() - t0
    net = NetLinear(4 ** k, args.n_clusters)
    for mod in net.modules():
        if isinstance(mod, nn.Linear):
            nn.init.kaiming_normal_(mod.weight); nn.init.zeros_(mod.bias)
    opt = torch.optim.RMSprop(net.parameters(), lr=1e-3, weight_decay=0.01)
    xt = torch.from_numpy(x)
    net.train()
    perm = torch.randperm(xt.shape[0])
    for i in range(0, xt.shape[0], args.batch_sz):
        idx = perm[i:i + args.batch_sz]
        opt.zero_grad()
        z1, h1 = net(xt[idx, 0]); z2, h2 = net(xt[idx, 1])
        loss = 0.75 * info_nce_loss(h1, h2, 0.85) + 0.25 * IID_loss(z1, z2, lamb=2.8)
        loss.backward(); opt.step()
    t_all = time.perf_counter() - t0
    return {"value": n / t_all, "unit": "sequences/sec", "cores": threads, "kind": "port",
            "sample": f"{n} of the 100000 x {L} bp sequences (same generator family), k={k}, n_mimics={args.n_mimics}, "
                      f"B={args.batch_sz}: oracle vectorise single-threaded {t_vec:.1f} s + torch-CPU epoch "
                      f"({threads} threads of {cores} cores) {t_all - t_vec:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=100000)
    ap.add_argument("--len", type=int, default=10000)
    ap.add_argument("--k", type=int, default=6)
    ap.add_argument("--n-clusters", dest="n_clusters", type=int, default=20)
    ap.add_argument("--n-mimics", dest="n_mimics", type=int, default=3)
    ap.add_argument("--batch-sz", dest="batch_sz", type=int, default=512)
    ap.add_argument("--with-predict", dest="with_predict", type=int, default=0,
                    help="1: include predict + all-gather of assignments in the timed region (default 0 = BASELINE.md's region)")
    ap.add_argument("--cpu-sample", dest="cpu_sample", type=int, default=24000)
    ap.add_argument("--no-cpu-baseline", dest="cpu_base", action="store_false")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    from idelucs_amd import _lib, gemm_tuning
    _lib.require_gpu()
    # rehearsal knobs (not used by the driver): IDELUCS_BENCH_BACKEND=gloo and IDELUCS_BENCH_DEVICES=1 let two ranks share the one
    # GPU of a test box, to exercise every line of the N > 1 path except RCCL itself
    backend = os.environ.get("IDELUCS_BENCH_BACKEND", "nccl")
    n_dev = int(os.environ.get("IDELUCS_BENCH_DEVICES", "0"))
    if n_dev > 0:
        local_rank %= n_dev
    torch.cuda.set_device(local_rank)
    os.environ.setdefault("IDELUCS_TUNABLEOP", "1")      # GEMM solution selection (PyTorch TunableOp), done in the warm-up step
    gemm_tuning.maybe_enable()
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    din = synth_packed(args.n, args.len, dev)
    hp = HotPath(din, args, dev)
    gathered = None

    def one_step(i):
        nonlocal gathered
        hp.step(seed=1000 * rank + i)
        if world > 1 and args.with_predict:
            from idelucs_amd.dist import all_gather_assignments
            gathered = all_gather_assignments(hp.y_pred)        # [world, N] int32 over RCCL/xGMI

    for i in range(args.warmup):
        one_step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    # the exchange step of the path (untimed unless --with-predict 1): this rank's voter predicts, assignments are all-gathered
    from idelucs_amd.dist import all_gather_assignments
    all_gather_assignments(hp.predict())            # first call: library initialisation for the inference GEMM shapes (0.5 s), not the path
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    gathered = all_gather_assignments(hp.predict())
    torch.cuda.synchronize()
    exchange_ms = 1e3 * (time.perf_counter() - t1)
    assert tuple(gathered.shape) == (world, args.n)
    assert not bool(hp.edit_overflow.item()), "mimic edit buffer bound exceeded: the run is invalid"

    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        value = args.n * world / (elapsed / args.steps)
        P, F = hp.P, hp.F
        # B_vec = ceil(L/4) + P*F*4 bytes per sequence; one launch processes all N sequences
        b_vec = (args.len + 3) // 4 + P * F * 4
        t_vec = hp.mean_ms("vectorise", args.warmup)
        ach = args.n * b_vec / (t_vec * 1e-3) / 1e9
        # F_ep per sequence (fwd+bwd of both views through the 3 dense layers + InfoNCE GEMMs)
        H1, H2, C, B = 512, 64, args.n_clusters, args.batch_sz
        f_ep = args.n_mimics * 2 * 3 * 2 * (F * H1 + H1 * H2 + H2 * C) + args.n_mimics * 3 * (2 * B) * H2 * 2 * 2
        t_ep = hp.mean_ms("epoch", args.warmup)
        ach_ep = args.n * f_ep / (t_ep * 1e-3) / 1e12
        # what the step actually executes: layer 1 has no input gradient (x is data), so its backward is one product, not two
        nce = args.n_mimics * 3 * (2 * B) * H2 * 2 * 2
        f_exec = args.n_mimics * 2 * (2 * 2 * F * H1 + 3 * 2 * (H1 * H2 + H2 * C)) + nce
        ach_exec = args.n * f_exec / (t_ep * 1e-3) / 1e12
        out = {
            "metric": "sequences/sec (k-mer vectorise + 1 epoch), k=6 batch 512",
            "value": value, "unit": "sequences/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: synthetic {args.n} x {args.len} bp, k={args.k}, n_clusters={args.n_clusters}, "
                                   f"n_mimics={args.n_mimics} ({P} views), batch_sz={args.batch_sz}, NetLinear fp32, RMSprop; "
                                   f"one voter per GPU; timed = device mimic sites + vectorise + scaler fit + 1 epoch"
                                   + (" + predict + all-gather of assignments" if args.with_predict
                                      else " (BASELINE.md section 3 region; predict + RCCL all-gather of assignments run once after it)"),
                       "n_sequences": args.n, "seq_len": args.len, "k": args.k, "batch_sz": args.batch_sz,
                       "optimizer_steps_per_epoch": (args.n * args.n_mimics + args.batch_sz - 1) // args.batch_sz,
                       "parallelism": f"voters x{world}"},
            "roofline": {"kernel": "vectorise2_kernel<6> (hand-written HIP: one count + per-view window deltas + normalise, all views)",
                         "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": pmc_traffic_gb(), "ms_per_launch": t_vec, "bytes_per_seq_algorithmic": b_vec,
                         "share_of_step": t_vec / ms_step},
            "roofline_epoch": {"kernel": "training epoch (hipBLASLt fp32 GEMMs + gather + losses + RMSprop)", "bound": "mfma",
                               "achieved": ach_ep, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_ep / MFMA_F32_PEAK_TFLOPS,
                               "ms": t_ep, "flop_per_seq_algorithmic": f_ep, "share_of_step": t_ep / ms_step,
                               "flop_per_seq_executed": f_exec, "achieved_executed": ach_exec, "frac_executed": ach_exec / MFMA_F32_PEAK_TFLOPS,
                               "note": "algorithmic = SURVEY 8(d) (3 x forward for every layer); executed = without the input gradient of "
                                       "layer 1, which is not computed; matrix-pipe busy time from PMC: profiles/r01_i_epoch_pmc_mfma.json"},
            "stage_ms": {k: hp.mean_ms(k, args.warmup) for k in hp.ev if hp.ev[k]},
            "exchange_ms": exchange_ms,
        }
        if world == 1 and args.cpu_base:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
