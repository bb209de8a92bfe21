#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json on MI355X.

metric   : sequences/sec (k-mer vectorise + 1 epoch), k=6, batch 512
workload : N = 1  -> BASELINE.json configs[1] ("cfg2"): synthetic 100 000 sequences x 10 kbp (iid uniform ACGT), k=6,
                     n_clusters=20, n_mimics=3 (CLI default), batch_sz=512, NetLinear, RMSprop, fp32, ONE voter.
           N > 1  -> configs[2] ("cfg3", BASELINE.md section 3 "Multi-GPU reporting"): the same data, a FIXED job of 8 voters
                     sharded voter v -> rank v mod N (8/N voters per rank), predict + the RCCL all-gather of the int32
                     assignments [V, N_seq] INSIDE the timed region.  value = N_seq * V / wall, "scaling": "strong".
           --workload cfg5 -> configs[4]: 1 000 000 x 5 kbp, n_clusters=0 => 200 output units; voters as above, then the last
                     voter's weights are broadcast, predict is sharded by sequence and the fp32 latent shards [N_seq/G, 64] are
                     all-gathered (timed).  (HDBSCAN itself is post-hoc and not part of the metric.)
step     : ONE pass of the hot path over the whole synthetic batch, packed bases already resident in HBM: device mimic-site
           generation + vectorisation of all 4 views + scaler fit (once per step, as the reference builds x_train once,
           idelucs/__main__.py:95) + one full training epoch per voter (586 optimizer steps at cfg2), i.e. BASELINE.md
           section 3's timed region.  Nothing is cached across steps: features are recomputed and every voter starts from a
           fresh Kaiming init (reference __main__.py:109).  --with-predict 1 adds predict at N = 1 too.
checks   : the epoch loss of every timed step must be finite, and after the timed loop (untimed) the feature store of the last
           step is validated (rows sum to 1, view 0 != view 1) and a checksum of it is printed ("validation").
N > 1    : one process per GPU (torch.distributed, backend nccl == RCCL); no collective inside an epoch.

At N = 1 the line also carries "fixed_job_8_voters": BASELINE.md section 3's FIXED cfg3 job (8 voters, predict + the [8, N] all-gather
through the RCCL group of one) run on the one GPU after the headline region -- the like-for-like N = 1 anchor of the 1/2/4/8
curve: the N >= 2 lines ("job": "cfg3 ...") compare with IT, not with the one-voter headline ("job": "cfg2 ...").

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (the hand-written vectoriser kernel, HBM bound), "roofline_epoch"
(the encoder epoch, MFMA fp32 bound), "cpu_baseline" (oracle port on the host cores, bounded sample, rank 0 at N=1 only),
"t_e2e" (FASTA text in the page cache -> last optimizer.step, N=1 only; never `value`).
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
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense fp16 / bf16 matrix peak (v_mfma_f32_32x32x16_f16; the 2:1-sparsity figure is twice that)
PMC_PROFILE = "r06_vectorise_pmc_k6.json"     # profiles/: FETCH_SIZE / WRITE_SIZE passes of the vectoriser at cfg2


def synth_packed(n, L, dev, seed=12345, n_rate=0.0):
    """iid-uniform ACGT == iid-uniform 2-bit codes, generated directly in the packed slot layout.  n_rate > 0: SURVEY 8(d)'s
    variant "N" -- every base is additionally an N with that probability (an invalid-mask bit: the window restarts behind it)."""
    slots = (L + 63) // 64
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    codes = torch.empty(n * slots * 4, dtype=torch.int32, device=dev)
    step = 1 << 28
    for lo in range(0, codes.numel(), step):     # chunked: randint materialises an int64 temporary of the request
        hi = min(lo + step, codes.numel())
        codes[lo:hi] = torch.randint(-2 ** 31, 2 ** 31 - 1, (hi - lo,), dtype=torch.int32, device=dev, generator=g)
    mask = torch.zeros((n, slots, 2), dtype=torch.int32, device=dev)
    tail = L % 64
    if tail:                     # bases past the end of the sequence are marked invalid
        w = [0, 0]
        for j in range(tail, 64):
            w[j // 32] |= 1 << (31 - (j % 32))
        for i in (0, 1):
            mask[:, -1, i] = w[i] - (1 << 32) if w[i] >= 2 ** 31 else w[i]

    if n_rate > 0.0:             # a Bernoulli(n_rate) bit per base, 32 bases per mask word
        words = mask.view(-1)
        for lo in range(0, words.numel(), 1 << 24):
            hi = min(lo + (1 << 24), words.numel())
            bits = (torch.rand((hi - lo, 32), device=dev, generator=g) < n_rate).to(torch.int64)
            w = (bits << torch.arange(31, -1, -1, device=dev, dtype=torch.int64)).sum(1)
            words[lo:hi] |= torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32)

    class DeviceInput:
        pass
    d = DeviceInput()
    d.n, d.codes, d.mask, d.min_len = n, codes, mask.view(-1), L
    d.slot_off = torch.arange(0, (n + 1) * slots, slots, dtype=torch.int64, device=dev)
    d.lengths = torch.full((n,), L, dtype=torch.int64, device=dev)
    d.max_len = L
    return d


class HotPath:
    """The timed region: everything from packed bases in HBM to the last optimizer.step() (and, for a multi-voter job, to the
    gathered assignments)."""

    def __init__(self, din, args, dev, rank, world, force_lanes=None):
        from idelucs_amd import _lib, utils as U, models, dist as D
        self.U, self._lib, self.models, self.D = U, _lib, models, D
        self.din, self.dev, self.a, self.rank, self.world = din, dev, args, rank, world
        self.specs = [t.spec() for t in U.mimic_transforms(args.n_mimics)]
        self.P = len(self.specs)
        self.F = 4 ** args.k
        self.model = models.IID_model({
            'sequence_file': None, 'GT_file': None, 'n_clusters': args.n_clusters, 'k': args.k, 'model_size': 'linear',
            'n_mimics': args.n_mimics, 'batch_sz': args.batch_sz, 'optimizer': 'RMSprop', 'lambda': 2.8, 'lr': 1e-3,
            'weight': 0.25, 'scheduler': None, 'n_epochs': 1, 'n_voters': args.voters})
        self.feats = torch.empty((self.P, din.n, self.F), dtype=torch.float32, device=dev)   # 6.55 GB at cfg2, 65.5 GB at cfg5
        # the scaler lives in persistent buffers refitted in place every step: the captured training-step graph (which bakes
        # their addresses) is captured once, in the warm-up, and replayed afterwards
        self.mean = torch.zeros(self.F, dtype=torch.float64, device=dev)
        self.scale = torch.ones(self.F, dtype=torch.float64, device=dev)
        self.store = U.FeatureStore(None, None, self.feats, self.mean, self.scale, args.k, False)
        self.model.store = self.store
        # mimic-site buffer sized by a safe bound (expected + 25 % + slack) instead of a host round trip per step; the
        # device-side totals are checked against it after the run (overflow flag, no sync inside the step)
        expect = sum(din.n * (args.len * (1.0 - (1.0 - s[0]) * (1.0 - s[1])) + s[2]) for s in self.specs)
        self.edit_capacity = int(1.25 * expect + 64 * din.n * self.P + 1024)
        self.edit_overflow = torch.zeros((), dtype=torch.bool, device=dev)
        self.ev = {k: [] for k in ("edits", "vectorise", "stats", "epoch", "predict_inputs", "predict", "exchange")}
        self.my_voters = D.voters_of_rank(args.voters, rank, world)
        # several voters of one rank train in lockstep as one batch (idelucs_amd.training.train_voters does the same for the CLI):
        # one launch sequence for all of them, each on its own network / permutation / dropout stream / optimizer state
        from idelucs_amd.training import voter_lanes, can_batch
        from idelucs_amd.fused import BatchedLinearTrainer
        L = (force_lanes or voter_lanes(len(self.my_voters), self.model)) if can_batch(self.model) else 1
        L = max(1, min(L, len(self.my_voters)))
        self.lanes = [self.model]
        self.batched = None
        if L > 1:
            self.lanes = [self.model.lane() for _ in range(L)]
            self.batched = BatchedLinearTrainer([lm.net for lm in self.lanes], 1e-3, 0.25, 2.8, seed=self.model.seed)
            for lm, t in zip(self.lanes, self.batched.trainers):
                lm._fused = t
        self.last_model = self.model
        self.losses = []            # device scalars, one per voter-epoch (checked after the timed loop)
        self.gathered = None
        self.latent = None

    def _timed(self, key, fn):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = fn()
        e.record()
        self.ev[key].append((s, e))
        return r

    def features(self, seed):
        U, _lib = self.U, self._lib
        edits, edit_off = self._timed("edits", lambda: U._philox_edits(self.din, self.specs, seed, capacity=self.edit_capacity))
        self.edit_overflow |= U.edits_overflowed(edits, edit_off, self.edit_capacity)
        self._timed("vectorise", lambda: U._vectorise(self.din, self.a.k, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_FREQ_F32,
                                                      self.P, edits, edit_off, self.feats))

        def fit():
            U.col_stats(self.feats[0], out=(self.mean, self.scale))
            self.store.refresh()
        self._timed("stats", fit)

    def step(self, seed):
        m, a = self.model, self.a
        self.features(seed)
        preds = {}
        with_predict = a.exchange and a.workload != "cfg5"
        if len(self.lanes) == 1:
            # the predict inputs are a function of the input alone: once per job, not once per voter (IID_model._predict_inputs)
            x = self._timed("predict_inputs", self._predict_inputs) if with_predict else None
            for v in self.my_voters:
                m.begin_voter(v)                                   # fresh Kaiming init + this voter's RNG streams + optimizer state
                self.losses.append(self._timed("epoch", lambda: m.contrastive_training_epoch(sync=False)))
                if with_predict:
                    preds[v] = self._timed("predict", lambda: self.predict(m, x))
        else:
            # (the predict inputs are a function of the input alone: once per job, not once per voter)
            x = self._timed("predict_inputs", self._predict_inputs) if with_predict else None
            L = len(self.lanes)
            for w in range(0, len(self.my_voters), L):
                wave = self.my_voters[w:w + L]
                if len(wave) < L:                                   # a short last batch: one voter after the other
                    for v in wave:
                        m.begin_voter(v)
                        self.losses.append(self._timed("epoch", lambda: m.contrastive_training_epoch(sync=False)))
                        if with_predict:
                            preds[v] = self._timed("predict", lambda: self.predict(m, x))
                        self.last_model = m
                    continue
                for lm, v in zip(self.lanes, wave):
                    lm.begin_voter(v)
                    lm.net.train()
                res = self._timed("epoch", lambda: self.batched.run_epoch(self.store, a.batch_sz, [lm._gen for lm in self.lanes]))
                self.epochs_timed_as_batches = True
                for lm, v, (total, nb) in zip(self.lanes, wave, res):
                    self.losses.append(total / (nb - 1))
                    if with_predict:
                        preds[v] = self._timed("predict", lambda: self.predict(lm, x))
                    self.last_model = lm
        if with_predict:                                           # the path's one exchange step: [V, N] int32 assignments
            self.gathered = self._timed("exchange", lambda: self.D.gather_voter_predictions(preds, a.voters, self.din.n, device=self.dev))
        elif a.exchange:                                           # cfg5: last voter's model everywhere, predict sharded by sequence
            self.latent = self._timed("exchange", self.sharded_latent)

    def _predict_inputs(self, lo=0, hi=None):
        """Un-mutated float64 vectors, own float64 scaler fitted on ALL rows (reference utils.py:400-405), rows [lo, hi)
        standardised to float32."""
        U, _lib = self.U, self._lib
        hi = self.din.n if hi is None else hi
        if U.counts_route_ok(self.a.k):          # as utils.predict_features: from the int32 counts, the float64 rows never materialised
            if not hasattr(self, "_predict_ws"):
                self._predict_ws = {}
            return U.predict_inputs_from_counts(self.din, self.a.k, (lo, hi), workspace=self._predict_ws)
        f64 = U._vectorise(self.din, self.a.k, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_FREQ_F64)[0]
        mean, scale = U.col_stats(f64)
        return U.standardise(f64[lo:hi], mean, scale)

    def predict(self, m, x):
        """Eval forward, argmax (reference models.py:145-172) -> int32 [N]."""
        preds = []
        with torch.no_grad():
            m.net.eval()
            for i in range(0, x.shape[0], 32768):         # (as IID_model._predict_outputs: row-wise ops, the chunk only sets the launch count)
                o, _ = m.net(x[i:i + 32768])
                preds.append(o.argmax(1).to(torch.int32))
        return torch.cat(preds)

    def sharded_latent(self):
        """cfg5 (reference __main__.py:153-156 clusters ONE model's latent): the last voter's weights go to every rank, each
        rank embeds its N/G sequences, the fp32 shards are all-gathered over RCCL."""
        m, D = self.last_model, self.D
        owner = (self.a.voters - 1) % self.world
        D.broadcast_parameters(m.net, owner)
        lo, hi = D.shard_bounds(self.din.n, self.rank, self.world)
        x = self._predict_inputs(lo, hi)
        lats = []
        with torch.no_grad():
            m.net.eval()
            for i in range(0, x.shape[0], 32768):
                lats.append(m.net(x[i:i + 32768])[1])
        return D.all_gather_rows(torch.cat(lats) if lats else torch.empty((0, 64), device=self.dev), self.din.n)

    def mean_ms(self, key, skip_steps):
        """Mean duration of one stage; "epoch" is per voter-epoch (voters batched: the batch's epoch divided by its voters)."""
        nv = len(self.my_voters)
        ev = self.ev[key]
        if key in ("epoch", "predict"):
            per_step = len(ev) // max(len(self.ev["vectorise"]), 1)
            ev = ev[skip_steps * per_step:]
            t = sum(s.elapsed_time(e) for s, e in ev) / max(len(ev), 1)
            return t * per_step / nv if key == "epoch" else t
        ev = ev[skip_steps:]
        return sum(s.elapsed_time(e) for s, e in ev) / max(len(ev), 1)

    def validate(self):
        """Untimed checks of what the last timed step produced (the bench validated nothing it timed)."""
        n = self.din.n
        losses = torch.stack([l.reshape(()) for l in self.losses]).double().cpu().numpy()
        assert np.all(np.isfinite(losses)), f"non-finite epoch loss in a timed step: {losses}"
        rows = torch.arange(0, n, max(n // 4096, 1), device=self.dev)
        sums = self.feats[:, rows].double().sum(2)
        assert torch.allclose(sums, torch.ones_like(sums), atol=1e-6), "frequency rows of the timed store do not sum to 1"
        assert not torch.equal(self.feats[0, rows], self.feats[1, rows]), "view 0 == view 1: the mimic views were not applied"
        chk = [float(self.feats[v].sum(dtype=torch.float64).item()) for v in range(self.P)]      # == N for every view
        assert all(abs(c - n) < 1e-3 * n for c in chk), chk
        return {"epoch_loss_last_step": [float(x) for x in losses[-len(self.my_voters):]], "epoch_loss_first_timed": float(losses[0]),
                "feats_checksum": chk, "rows_checked": int(rows.numel())}


def check_votes(gathered, voters, n, n_clusters):
    """The exchange step's result: int32 [V, N] in range, and every pair of voters a different run (per-voter init / permutation /
    dropout streams; also across ranks)."""
    assert tuple(gathered.shape) == (voters, n), (tuple(gathered.shape), voters, n)
    assert int(gathered.min()) >= 0 and int(gathered.max()) < n_clusters
    distinct = True
    for i in range(voters):
        for j in range(i + 1, voters):
            distinct = distinct and not torch.equal(gathered[i], gathered[j])
    assert distinct, "two voters produced identical assignments"
    return {"gathered_shape": list(gathered.shape), "voters_pairwise_distinct": bool(distinct)}


def fixed_job_8_voters(din, args, dev, rank, world, passes=3):
    """BASELINE.md section 3's fixed job on ONE GPU: cfg3's 8 voters -- here batched in lockstep on the one rank
    -- each one epoch from a fresh init, then predict per voter and the all-gather of the int32 assignments [8, N] through
    the process group (RCCL, a group of one), everything inside the timed region as at N > 1.  One warm-up pass (graph capture,
    GEMM selection for the batched shapes), then `passes` timed passes."""
    import copy
    a = copy.copy(args)
    a.voters, a.exchange = 8, True
    hp = HotPath(din, a, dev, rank, world)
    hp.step(seed=1000)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(passes):
        hp.step(seed=1001 + i)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / passes
    assert not bool(hp.edit_overflow.item())
    v = hp.validate()
    v.update(check_votes(hp.gathered, 8, args.n, args.n_clusters))
    st = {k: hp.mean_ms(k, 1) for k in hp.ev if hp.ev[k]}
    return {"job": "cfg3: 8 voters x 1 epoch + predict + all-gather of int32 [8, N] (BASELINE.md section 3), all on this GPU",
            "value": args.n * 8 / (ms * 1e-3), "unit": "sequences x voters / sec", "sequences_per_sec_job": args.n / (ms * 1e-3),
            "ms_per_pass": ms, "passes": passes, "lanes": len(hp.lanes),
            "exchange_ms": st.get("predict_inputs", 0.0) + 8 * st.get("predict", 0.0) + st.get("exchange", 0.0),
            "stage_ms": st, "backend": dist.get_backend() if dist.is_initialized() else None, "validation": v}


def k_sweep(din, args, dev, ks=(4, 5), reps=6):
    """BASELINE configs[3] (cfg4) names the k = 4 / k = 5 path: the vectorise stage alone at this workload's size for those k,
    with its own roofline (B_vec = ceil(L / 4) + P * 4^k * 4 bytes a sequence -- at k = 4 the packed bases are a
    third of it and the kernel is bound by its LDS atomics, not by HBM: DESIGN 4.1).  HIP events on the launch stream."""
    from idelucs_amd import _lib, utils as U
    specs = [t.spec() for t in U.mimic_transforms(args.n_mimics)]
    P = len(specs)
    edits, edit_off = U._philox_edits(din, specs, 4242)
    out = {}
    for k in ks:
        F = 4 ** k
        feats = torch.empty((P, din.n, F), dtype=torch.float32, device=dev)
        run = lambda: U._vectorise(din, k, _lib.MODE_KMER, _lib.INIT_ONE, _lib.OUT_FREQ_F32, P, edits, edit_off, feats)
        run(); torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); run(); e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e))
        ms = sum(ts) / len(ts)
        b_vec = (args.len + 3) // 4 + P * F * 4
        ach = din.n * b_vec / (ms * 1e-3) / 1e9
        rows = torch.arange(0, din.n, max(din.n // 2048, 1), device=dev)
        sums = feats[:, rows].double().sum(2)
        assert torch.allclose(sums, torch.ones_like(sums), atol=1e-6) and not torch.equal(feats[0, rows], feats[1, rows])
        # (the k = 4 / 5 kernels are bound by their LDS atomics -- ~15 000 a sequence at 10 kbp and 4 views against the 4.5-6 a clock and CU the
        #  random-bin read-modify-writes sustain --, not by HBM: `frac` stays the fraction of the HBM peak the algorithmic bytes reach, for
        #  comparison with k = 6, and `bound` says what actually holds the kernel)
        atomics = (args.len - k + 1) + 2 * 0.015 * args.len * k * (P - 1)          # the count + an (old bin, new bin) pair per edit and window of every mimic view
        lds_floor_ms = din.n * atomics / (256 * 16 * 2.4e9) * 1e3                     # 16 a clock and CU: the instruction's issue rate
        del feats
        #  the canonical rows of model_size='small' (reverse-complement collapse, utils.py:208-221) at this k, on the same kernel's epilogue
        can_ms = _canonical_ms(din, k, P, edits, edit_off, dev, reps)
        #  the WHOLE path at this k -- sites + vectorise + scaler fit + one epoch of the same NetLinear(4^k -> 512 -> 64 -> C) on the same
        # 100 000 x 10 kbp: cfg4's shapes get a throughput figure, not only a vectorise stage.  At k = 5 (F = 1024) the step takes the two-plane
        # products; at k = 4 (F = 256: cluster.py's default k) no plane kernel applies (F >= 1024) and the step is its latency-bound launches
        import copy
        a = copy.copy(args)
        a.k, a.voters, a.exchange = k, 1, False
        hp = HotPath(din, a, dev, 0, 1)
        hp.step(seed=5000)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(2):
            hp.step(seed=5001 + i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
        v = hp.validate()
        whole = {"value": din.n / dt, "unit": "sequences/sec", "ms_per_pass": 1e3 * dt, "epoch_ms": hp.mean_ms("epoch", 1),
                 "stage_ms": {kk: hp.mean_ms(kk, 1) for kk in ("edits", "vectorise", "stats", "epoch")},
                 "step_form": "two-plane products" if getattr(hp.model._fused, "_planes", False) and F >= 1024 and F % 512 == 0 else "fp32 tiles / library GEMMs (F < 1024)",
                 "epoch_loss_last_step": v.get("epoch_loss_last_step")}
        del hp
        torch.cuda.empty_cache()
        feats = None
        out[str(k)] = {"kernel": "vectorise4_kernel<%d> (a wavefront per sequence)" % k, "bound": "lds", "ms_per_launch": ms, "ms_min": min(ts), "whole_path": whole,
                       "bytes_per_seq_algorithmic": b_vec, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                       "lds_atomics_per_seq": int(atomics), "lds_issue_floor_ms": lds_floor_ms, "frac_of_lds_issue_floor": lds_floor_ms / ms,
                       "sequences_per_sec_stage": din.n / (ms * 1e-3), "rows_checked": int(rows.numel()),
                       "canonical_rows_ms_per_launch": can_ms}
    # k = 6 (the reference's default k): the canonical rows alone -- the plain rows are `roofline`'s kernel
    out["6"] = {"kernel": "vectorise4_kernel<6> (canonical rows: a wavefront per sequence; plain rows at k = 6 are `roofline`)",
                "canonical_rows_ms_per_launch": _canonical_ms(din, 6, P, edits, edit_off, dev, reps)}
    return out


def _canonical_ms(din, k, P, edits, edit_off, dev, reps):
    from idelucs_amd import _lib, utils as U
    rl = int(_lib.lib.idl_row_len(_lib.MODE_CANONICAL, k))
    rows = torch.empty((P, din.n, rl), dtype=torch.float32, device=dev)
    run = lambda: U._vectorise(din, k, _lib.MODE_CANONICAL, _lib.INIT_ONE, _lib.OUT_FREQ_F32, P, edits, edit_off, rows)
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); run(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    sums = rows[:, ::max(din.n // 1024, 1)].double().sum(2)
    assert torch.allclose(sums, torch.ones_like(sums), atol=1e-6)
    return sum(ts) / len(ts)


def lanes_epoch_ms(din, args, dev, rank, world, voters):
    """Per-voter epoch time with `voters` voters batched in lockstep on this GPU (one warm-up pass + one timed): the
    lanes-dependence the scaling prediction needs (8 / N voters per rank at N ranks)."""
    import copy
    a = copy.copy(args)
    a.voters, a.exchange = voters, False
    hp = HotPath(din, a, dev, rank, world, force_lanes=voters)       # (in lockstep whatever training.voter_lanes would choose for so few)
    hp.step(seed=2000)
    hp.step(seed=2001)
    torch.cuda.synchronize()
    ms = hp.mean_ms("epoch", 1)
    del hp
    torch.cuda.empty_cache()
    return ms


def plane_kernel_leg(dev, m, H, F, n=20, reps=5):
    """The step's two plane kernels each on its own (operands as the step leaves them: planes of a standardised batch, of W1, of a dr1 with the scale
    mid_bwd gives it), `n` launches captured in a graph, HIP events around `reps` replays on the launch stream: us a launch, and the executed flops --
    three fp16 products of 2 m H F each -- as a fraction of the dense fp16 matrix peak (the pipe the products run on).  Alone, a
    kernel finds its operands in the caches the previous launch of itself left; inside the step (profiles/r06_step_kernels.txt) they come from the
    kernels in between, and the launches take 1-3 us longer."""
    import ctypes
    import math
    from idelucs_amd import _lib
    L = _lib.lib
    if not (L.idl_l1_planes_supported(m, H, F) and L.idl_wgrad_xplanes_supported(m, H, F) and m % 64 == 0):
        return None
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    g = torch.Generator(device="cpu"); g.manual_seed(5)
    W = ((torch.rand(H, F, generator=g) * 2 - 1) / F ** 0.5).to(dev)
    x = torch.randn(m, F, generator=g).to(dev)
    dy = (torch.randn(m, H, generator=g) * 1e-4).to(dev)
    h16 = lambda t: torch.empty(t.shape, dtype=torch.int16, device=dev)
    wh, wl, xh, xl, dyh, dyl = h16(W), h16(W), h16(x), h16(x), h16(dy), h16(dy)
    s = torch.cuda.Stream()
    out = {}
    with torch.cuda.stream(s):
        st = ctypes.c_void_p(s.cuda_stream)
        kd = 9 - math.frexp(dy.abs().max().item())[1]
        _lib.check(L.idl_split_planes(p(W), W.numel(), L.idl_planes_exponent(1), p(wh), p(wl), None, st))
        _lib.check(L.idl_split_planes(p(x), x.numel(), L.idl_planes_exponent(0), p(xh), p(xl), None, st))
        _lib.check(L.idl_split_planes(p(dy), dy.numel(), kd, p(dyh), p(dyl), None, st))
        sc = torch.zeros(int(L.idl_dr1_scale_words()), dtype=torch.int32, device=dev); sc[0] = kd
        part = torch.empty(int(L.idl_l1_planes_parts()), H, m, device=dev)
        V = torch.zeros_like(W); flag = torch.zeros(1, dtype=torch.int32, device=dev)
        hyper = torch.tensor([1e-3, 0.99, 1e-8, 0.01, 0.01], dtype=torch.float32, device=dev)
        fns = {"l1_planes_kernel": lambda: _lib.check(L.idl_l1_planes(p(wh), p(wl), F, p(xh), p(xl), F, m, H, F, p(part), st)),
               "wgrad_dplanes_kernel": lambda: _lib.check(L.idl_wgrad_rmsprop_xplanes(p(dyh), p(dyl), p(sc), p(xh), p(xl), F, m, H, F, None, p(W), p(V), p(hyper),
                                                                                       p(wh), p(wl), p(flag), st))}
        for name, fn in fns.items():
            for _ in range(3):
                fn()
            s.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                for _ in range(n):
                    fn()
            gr.replay(); s.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(reps):
                gr.replay()
            e1.record(s)
            s.synchronize()
            us = e0.elapsed_time(e1) / (n * reps) * 1e3
            gflop = 3 * 2.0 * m * H * F / 1e9
            out[name] = {"us_per_launch_alone": us, "executed_gflop": gflop, "achieved_tflops": gflop / us * 1e3,
                         "frac_of_fp16_peak": gflop / (us * 1e-6) / 1e3 / MFMA_F16_PEAK_TFLOPS}
    torch.cuda.current_stream().wait_stream(s)
    out["peak_tflops_fp16_dense"] = MFMA_F16_PEAK_TFLOPS
    return out


def fp32_form_leg(din, args, dev, rank, world, passes=4):
    """The headline region once more with the step's fp32 form (IDELUCS_PLANES=0: both big products on the fp32 matrix cores, round 4 / early
    round 5's default) -- reported beside `value`, never as `value`: what the two-plane products of the default step buy."""
    import copy
    a = copy.copy(args)
    a.voters, a.exchange = 1, False
    prev = os.environ.get("IDELUCS_PLANES")               # (restored below: a user's own setting must survive this leg -- ADVICE r5)
    os.environ["IDELUCS_PLANES"] = "0"
    try:
        hp = HotPath(din, a, dev, rank, world)
        hp.step(seed=3000)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(passes):
            hp.step(seed=3001 + i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / passes
        v = hp.validate()
        out = {"what": "IDELUCS_PLANES=0: layer 1 and dW1 on the fp32 matrix cores (own tiles: l1_rms_kernel, wgrad_q16_kernel) instead of the fp16 "
                       "matrix cores from two fp16 planes per operand (three products, fp32 accumulators); everything else as in `value`",
               "value": args.n / dt, "unit": "sequences/sec", "ms_per_pass": 1e3 * dt, "epoch_ms": hp.mean_ms("epoch", 1),
               "epoch_loss_last_step": v.get("epoch_loss_last_step")}
        del hp
    finally:
        if prev is None:
            os.environ.pop("IDELUCS_PLANES", None)
        else:
            os.environ["IDELUCS_PLANES"] = prev
    torch.cuda.empty_cache()
    return out


def predicted_fixed_job(din, args, dev, rank, world, st8, epoch1_ms, lane_model=None):
    """What the 1/2/4/8 curve of the FIXED 8-voter job (cfg3) should look like, from this GPU's own stage times:
    per-rank wall at N ranks = sites + vectorise + scaler fit + (8/N) x epoch(lanes = 8/N) + predict inputs + (8/N) x predict +
    all-gather.  Every rank vectorises the whole input; the voters of a rank train in lockstep, and a voter-epoch costs less in
    a batch of 8 than alone (the latency-bound launches are shared) -- so the curve is sub-linear by construction, before any
    communication: anything BELOW this prediction is communication or host contention."""
    per_lanes = {8: st8["epoch"], 1: epoch1_ms}
    for v in (4, 2):
        per_lanes[v] = lanes_epoch_ms(din, args, dev, rank, world, v)
    fixed = st8["edits"] + st8["vectorise"] + st8["stats"] + st8.get("predict_inputs", 0.0) + st8.get("exchange", 0.0)
    out = {"epoch_ms_per_voter_by_lanes": {str(k): per_lanes[k] for k in sorted(per_lanes)}, "fixed_ms_per_rank": fixed,
           "predict_ms_per_voter": st8.get("predict", 0.0)}
    # training.voter_lanes: how the 8 / N voters of a rank train (in lockstep as one batch, or one after the other)
    from idelucs_amd.training import voter_lanes
    policy = {}
    base = None
    for n in (1, 2, 4, 8):
        v = 8 // n
        lanes = 1 if v == 1 else voter_lanes(v, lane_model)
        policy[str(v)] = "one after the other" if lanes == 1 else "in lockstep"
        per_voter = per_lanes[1] if lanes == 1 else per_lanes[v]
        wall = fixed + v * per_voter + v * st8.get("predict", 0.0)
        val = args.n * 8 / (wall * 1e-3)
        base = base or val
        out[str(n)] = {"ms_per_pass": wall, "value": val, "speedup_vs_1": val / base}
    out["voters_of_a_rank_train"] = policy
    out["note"] = ("value = N_seq x 8 voters / predicted per-rank wall; %.1fx at N = 8 is the EXPECTED speed-up (not 8x): one GPU trains its 8 "
                   "voters in lockstep at %.1f ms a voter-epoch (4: %.1f, 2: %.1f), a lone voter takes %.1f; 2 voters of a rank train %s"
                   % (out["8"]["speedup_vs_1"], per_lanes[8], per_lanes[4], per_lanes[2], per_lanes[1], policy["2"]))
    return out


def cfg5_one_gpu(args, dev, rank, world, passes=2):
    """BASELINE configs[4] (cfg5) on this one GPU under the driver's clock: 10^6 x 5 kbp, 200 output units, one
    voter: sites + vectorise into the 65.5 GB store + scaler fit + a 5 860-step epoch + predict inputs + latent [10^6, 64] through
    the process group.  One warm-up pass, then `passes` timed."""
    import copy
    a = copy.copy(args)
    a.workload, a.n, a.len, a.n_clusters, a.voters, a.exchange = "cfg5", 1_000_000, 5_000, 200, 1, True
    din = synth_packed(a.n, a.len, dev, seed=54321)
    hp = HotPath(din, a, dev, rank, world)
    hp.step(seed=3000)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(passes):
        hp.step(seed=3001 + i)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / passes
    assert not bool(hp.edit_overflow.item())
    v = hp.validate()
    assert tuple(hp.latent.shape) == (a.n, 64) and bool(torch.isfinite(hp.latent).all())
    v.update(latent_shape=list(hp.latent.shape), latent_row_norm_max=float(hp.latent.norm(dim=1).max().item()))
    st = {k: hp.mean_ms(k, 1) for k in hp.ev if hp.ev[k]}
    F = hp.F
    b_vec = (a.len + 3) // 4 + hp.P * F * 4
    out = {"job": "cfg5: 10^6 x 5 kbp, 200 output units, 1 voter x 1 epoch (5 860 steps) + sharded predict + latent all-gather, all on this GPU",
           "value": a.n / (ms * 1e-3), "unit": "sequences/sec", "ms_per_pass": ms, "passes": passes, "stage_ms": st,
           "roofline_vectorise": {"bound": "hbm", "achieved": a.n * b_vec / (st["vectorise"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": a.n * b_vec / (st["vectorise"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "bytes_per_seq_algorithmic": b_vec},
           "feature_store_gb": hp.feats.numel() * 4 / 1e9, "validation": v}
    del hp, din
    torch.cuda.empty_cache()
    return out


def pmc_traffic_gb():
    """HBM traffic of one vectorise launch at cfg2, GB, from the committed rocprofv3 PMC passes (profiles/, WRITE_SIZE exact
    for 16-B stores, FETCH_SIZE doubled per MI355X_MICROARCH.md: DESIGN.md 4.1); None when the profile is not present."""
    for name in (PMC_PROFILE, "r04_vectorise_pmc.json", "r03_vectorise_pmc.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                return json.load(fh)["traffic_gb_per_launch"]
        except Exception:
            continue
    return None


CPU_CALIBRATION = "r03_cpu_calibration.json"   # profiles/: port vs the imported reference in the build container (tools/calibrate_cpu_baseline.py)


def cpu_calibration():
    """{"ratio": port seq/s / reference seq/s on the same sample, ...} as measured in the build container (the reference cannot
    travel to the GPU box), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", CPU_CALIBRATION)) as fh:
            return json.load(fh)
    except Exception:
        return None


def cpu_baseline(args):
    """The oracle (CPU restatement of the reference algorithm) timed on this box's host cores, in the
    reference's shape: single-threaded per-record vectorise passes with numpy/random mimics
    (idelucs/utils.py:224-368), StandardScaler, then one PyTorch-CPU epoch.  Two thread settings for the epoch: the reference's
    own cpu_count()-2 (idelucs/__main__.py:316; on a 256-core host that oversubscribes these small GEMMs badly, so it is timed
    on a bounded number of optimizer steps and scaled) and min(cores-2, 32), the CPU's best shot, which is `value`.
    Bounded sample of the same workload; seq/s = sample / wall."""
    import random
    from oracle import oracle as O
    import torch.nn as nn
    from idelucs_amd.PytorchUtils import NetLinear
    from idelucs_amd.LossFunctions import IID_loss, info_nce_loss
    n, L, k = args.cpu_sample, args.len, args.k
    rng = np.random.default_rng(12345)
    seqs = [bytearray(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=L).tobytes()) for _ in range(n)]
    cores = os.cpu_count() or 1
    threads = max(1, min(cores - 2, 32))
    ref_threads = max(1, cores - 2)
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
    xt = torch.from_numpy(x)
    n_steps = (xt.shape[0] + args.batch_sz - 1) // args.batch_sz

    def epoch(nthreads, max_steps, skip=0):
        """-> (seconds per epoch scaled from the timed steps, timed steps); the first `skip` steps run untimed (thread-pool spin-up)"""
        torch.set_num_threads(nthreads)
        torch.manual_seed(0)
        net = NetLinear(4 ** k, args.n_clusters)
        for mod in net.modules():
            if isinstance(mod, nn.Linear):
                nn.init.kaiming_normal_(mod.weight); nn.init.zeros_(mod.bias)
        opt = torch.optim.RMSprop(net.parameters(), lr=1e-3, weight_decay=0.01)
        net.train()
        perm = torch.randperm(xt.shape[0])
        t1 = time.perf_counter()
        done = -skip
        for i in range(0, xt.shape[0], args.batch_sz):
            if done >= max_steps:
                break
            if done == 0:
                t1 = time.perf_counter()
            idx = perm[i:i + args.batch_sz]
            opt.zero_grad()
            z1, h1 = net(xt[idx, 0]); z2, h2 = net(xt[idx, 1])
            loss = 0.75 * info_nce_loss(h1, h2, 0.85) + 0.25 * IID_loss(z1, z2, lamb=2.8)
            loss.backward(); opt.step()
            done += 1
        return (time.perf_counter() - t1) * n_steps / max(done, 1), done

    t_ep, _ = epoch(threads, n_steps)
    if ref_threads != threads:
        t_ep_ref, steps_ref = epoch(ref_threads, max(1, min(n_steps - 1, int(args.cpu_ref_steps))), skip=1)
    else:
        t_ep_ref, steps_ref = t_ep, n_steps
    cal = cpu_calibration()
    out = {"value": n / (t_vec + t_ep), "unit": "sequences/sec", "cores": cores, "threads": threads, "kind": "port",
           "value_reference_threads": n / (t_vec + t_ep_ref), "reference_threads": ref_threads,
           "calibration_ratio": cal["ratio"] if cal else None,
           "calibration_source": (f"profiles/{CPU_CALIBRATION}: port / imported reference on the build container's {cal.get('cores')} cores, "
                                  f"{cal.get('sample')}") if cal else None,
           "sample": f"{n} of the {args.n} x {L} bp sequences (same generator family), k={k}, n_mimics={args.n_mimics}, "
                     f"B={args.batch_sz}: oracle vectorise single-threaded {t_vec:.1f} s + torch-CPU epoch of {n_steps} steps "
                     f"{t_ep:.1f} s with {threads} threads of {cores} cores (value); with the reference's cpu_count()-2 = {ref_threads} threads "
                     f"{t_ep_ref:.1f} s (timed on {steps_ref} steps, scaled)"}
    return out


def t_e2e(args, dev, reps=3):
    """SURVEY 8(d) / BASELINE.md section 3 T_e2e: the synthetic FASTA text (seed 12345, >seq%06d, one line per sequence,
    ~1.0 GB at cfg2) is written to tmpfs once, untimed; timed = open (page cache) -> C++ parse/validate/2-bit pack in record
    chunks, each chunk's H2D copy and vectorisation overlapped with the parsing of the next -> scaler fit -> the last
    optimizer.step() of epoch 1, device-synchronised."""
    from idelucs_amd import models, utils as U
    d = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
    path = os.path.join(d, f"idelucs_synth_{args.n}x{args.len}_{os.getpid()}.fas")
    rng = np.random.default_rng(12345)
    with open(path, "wb") as f:
        blk_n = 2000
        for i0 in range(0, args.n, blk_n):
            nb = min(blk_n, args.n - i0)
            blk = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=(nb, args.len), dtype=np.uint8)]
            rec = np.empty((nb, 11 + args.len + 1), np.uint8)       # ">seq%06d\n" is 11 bytes for i < 10^6
            hdr = np.frombuffer(b"".join(b">seq%06d\n" % (i0 + j) for j in range(nb)), np.uint8).reshape(nb, 11)
            rec[:, :11] = hdr; rec[:, 11:-1] = blk; rec[:, -1] = 10
            f.write(rec.tobytes())
    size = os.path.getsize(path)
    margs = {'sequence_file': path, 'GT_file': None, 'n_clusters': args.n_clusters, 'k': args.k, 'model_size': 'linear',
             'n_mimics': args.n_mimics, 'batch_sz': args.batch_sz, 'optimizer': 'RMSprop', 'lambda': 2.8, 'lr': 1e-3,
             'weight': 0.25, 'scheduler': None, 'n_epochs': 1, 'n_voters': 1}
    best, runs = None, []
    try:
        m = models.IID_model(margs)
        per_rep = []
        for rep in range(reps + 1):                      # rep 0 = warm-up (page cache, allocator, graph capture)
            U._L.idl_ingest_release()                    # every rep maps the file afresh, as a new process would (and pays its page faults)
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            ev0.record()                                 # (on an idle device: the device-side image of t0)
            # (IID_model.build_dataloader's call: a store of the same shape is refitted in place, so the step graph captured in rep 0
            #  -- it bakes the store's addresses -- serves the timed reps; graph_captures_so_far says so)
            m.store = U.build_feature_store(path, args.n_mimics, k=args.k, device=m.device, streamed=True, reuse=m.store)
            ev1.record()                                 # the store is complete when the device gets here; the host does NOT wait:
            m.begin_voter(0)                             # the epoch's launches queue behind the vectoriser, as in a real run
            loss = m.contrastive_training_epoch()
            torch.cuda.synchronize(); t2 = time.perf_counter()
            assert np.isfinite(loss)
            feat_ms = ev0.elapsed_time(ev1)
            r = {"ms": 1e3 * (t2 - t0), "ingest_to_features_ms": feat_ms, "epoch_ms": 1e3 * (t2 - t0) - feat_ms,
                 "graph_captures_so_far": getattr(m._fused, "n_captures", 0),
                 "sequences_per_sec": args.n / (t2 - t0), "fasta_bytes": size, "host_threads": U.ingest_threads(),
                 "reader_numa_node": int(U._L.idl_ingest_numa_node()), "file_pages_numa_node": int(U._L.idl_ingest_file_node())}
            if rep > 0:
                per_rep.append(round(r["ms"], 2))
                runs.append(r)
        #  the reported repetition is the MEDIAN one, not the best: the first timed repetition runs ~10 % over the others (the
        # reader's arenas and the file's mapping are set up again after idl_ingest_release, and the pool's threads start cold); ms_best beside it
        runs.sort(key=lambda r: r["ms"])
        best = dict(runs[len(runs) // 2])
        best["ms_best"] = runs[0]["ms"]
        best["reported"] = "the median of the timed repetitions (ms_of_every_rep in order of execution)"
        best["ms_of_every_rep"] = per_rep
        best["split"] = "ingest_to_features_ms = device time from the start to the finished store (HIP events); epoch_ms = the rest of the wall time; no host wait between the two"
    finally:
        os.unlink(path)
    return best


class stdout_to_stderr:
    """RCCL prints its version banner to STDOUT when the environment sets NCCL_DEBUG=VERSION (this image does), at communicator
    creation; bench.py owes its caller exactly one line there.  File-descriptor level, so the library's own writes are caught."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def spawn_ranks(n, n_visible, n_shared):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD `python -m torch.distributed.run` (one process
    per GPU, RCCL), relay rank 0's JSON line, return the child's exit code.  The parent has only counted devices
    (torch.cuda.device_count() -- which on ROCm may already have brought the HIP runtime up): it must only ever SPAWN a child
    and wait for it, never replace itself with another program (os.exec* after the runtime is up is forbidden on this pool).
    Fewer visible GPUs than ranks is an error, never a smaller run."""
    import socket
    import subprocess
    if n_shared == 0 and n_visible < n:
        print(f"bench.py: --gpus {n} but only {n_visible} GPU(s) visible: refusing to run fewer ranks", file=sys.stderr)
        return 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # (the arguments travel in the environment: the launcher's own parser claims abbreviations such as --n even behind the script)
    env["IDELUCS_BENCH_ARGV"] = json.dumps(sys.argv[1:])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    for l in r.stdout.splitlines():
        if not l.startswith('{"metric"'):
            print(l, file=sys.stderr)
    if r.returncode != 0 or len(lines) != 1:
        print(f"bench.py: the {n}-rank run failed (exit {r.returncode}, {len(lines)} result line(s))", file=sys.stderr)
        return r.returncode or 1
    print(lines[0])
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["cfg2", "cfg5"], default="cfg2")
    ap.add_argument("--n", "--n-sequences", dest="n", type=int, default=None)
    ap.add_argument("--len", type=int, default=None)
    ap.add_argument("--k", type=int, default=6)
    ap.add_argument("--n-clusters", dest="n_clusters", type=int, default=None)
    ap.add_argument("--n-mimics", dest="n_mimics", type=int, default=3)
    ap.add_argument("--batch-sz", dest="batch_sz", type=int, default=512)
    ap.add_argument("--voters", type=int, default=None,
                    help="voters of the job (default: 1 on one GPU = cfg2; 8 on several GPUs = cfg3's fixed job)")
    ap.add_argument("--with-predict", dest="with_predict", type=int, default=None,
                    help="1: predict + all-gather inside the timed region (default: 0 at N=1 with one voter = BASELINE.md's region; 1 otherwise)")
    ap.add_argument("--cpu-sample", dest="cpu_sample", type=int, default=24000)
    ap.add_argument("--cpu-ref-steps", dest="cpu_ref_steps", type=int, default=2,
                    help="optimizer steps timed (after one untimed step) with the reference's cpu_count()-2 torch threads, scaled to the epoch")
    ap.add_argument("--no-k-sweep", dest="k_sweep", action="store_false", help="skip the k = 4 / k = 5 vectorise-stage rooflines (cfg4)")
    ap.add_argument("--no-split16", "--no-fp32-form", dest="split16", action="store_false", help="skip the leg with the step's fp32 form (IDELUCS_PLANES=0)")
    ap.add_argument("--no-cfg5", dest="cfg5_leg", action="store_false", help="skip the cfg5 job (10^6 x 5 kbp, 65.5 GB store) the default N = 1 run adds")
    ap.add_argument("--no-prediction", dest="prediction", action="store_false", help="skip the 2- and 4-lane passes behind predicted_fixed_job")
    ap.add_argument("--no-cpu-baseline", dest="cpu_base", action="store_false")
    ap.add_argument("--no-e2e", dest="e2e", action="store_false")
    ap.add_argument("--no-fixed-job", dest="fixed_job", action="store_false",
                    help="skip the fixed 8-voter job (cfg3) that the default N = 1 run adds after the headline region")
    ap.add_argument("--n-rate", dest="n_rate", type=float, default=0.0,
                    help="SURVEY 8(d) variant N: every synthetic base is an N with this probability (e.g. 1e-3); default 0 = BASELINE's input")
    forwarded = os.environ.pop("IDELUCS_BENCH_ARGV", None)      # set by spawn_ranks for its ranks
    args = ap.parse_args(json.loads(forwarded) if forwarded is not None and "WORLD_SIZE" in os.environ else None)

    cfg5 = args.workload == "cfg5"
    args.n = args.n or (1_000_000 if cfg5 else 100_000)
    args.len = args.len or (5_000 if cfg5 else 10_000)
    args.n_clusters = args.n_clusters or (200 if cfg5 else 20)
    # rehearsal knobs (not used by the driver): IDELUCS_BENCH_BACKEND=gloo and IDELUCS_BENCH_DEVICES=1 let two ranks share the one
    # GPU of a test box, to exercise every line of the N > 1 path except RCCL itself
    backend = os.environ.get("IDELUCS_BENCH_BACKEND", "nccl")
    n_shared = int(os.environ.get("IDELUCS_BENCH_DEVICES", "0"))
    n_visible = torch.cuda.device_count()            # (may initialise the HIP runtime: from here on this process spawns children only, never execs)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, n_visible, n_shared))      # the parent runs no kernels; the ranks are a child process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to report a line for a "
                 f"different rank count")
    if n_shared == 0 and n_visible < world:
        sys.exit(f"bench.py: {world} ranks need {world} GPUs, {n_visible} visible (one process per GPU; no fallback to fewer ranks)")
    if args.voters is None:
        args.voters = 1 if world == 1 else (world if cfg5 else 8)
    args.exchange = bool(args.with_predict) if args.with_predict is not None else (world > 1 or args.voters > 1 or cfg5)
    from idelucs_amd import _lib, gemm_tuning
    _lib.require_gpu()
    if n_shared > 0:
        local_rank %= n_shared
    torch.cuda.set_device(local_rank)
    os.environ.setdefault("IDELUCS_TUNABLEOP", "1")      # GEMM solution selection (PyTorch TunableOp), done in the warm-up step
    gemm_tuning.maybe_enable()
    dev = torch.device("cuda", local_rank)
    # one process per GPU over RCCL.  A single rank forms a group of one as well: the exchange step of the path then runs through
    # the same RCCL calls (dtypes, shapes) as at N > 1 instead of a local shortcut (idelucs_amd.dist._no_group)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    with stdout_to_stderr():
        if world > 1:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
        elif os.environ.get("IDELUCS_BENCH_GROUP_OF_ONE", "1") == "1":
            dist.init_process_group(backend, rank=0, world_size=1, store=dist.HashStore(),
                                    **({"device_id": dev} if backend == "nccl" else {}))
        if dist.is_initialized() and dist.get_backend() == "nccl":     # the communicator (and its banner) comes with the first collective
            dist.all_reduce(torch.zeros(1, device=dev))
            torch.cuda.synchronize()
    group = {"ranks": dist.get_world_size() if dist.is_initialized() else 1,
             "backend": dist.get_backend() if dist.is_initialized() else None}
    mine = {"rank": rank, "device": torch.cuda.current_device(), "name": torch.cuda.get_device_name(dev), "pid": os.getpid()}
    if dist.is_initialized():
        devs = [None] * group["ranks"]
        dist.all_gather_object(devs, mine)
        group["devices"] = devs
    else:
        group["devices"] = [mine]
    if n_shared == 0 and world > 1:
        assert len({d["device"] for d in group["devices"]}) == world, f"ranks share a GPU: {group['devices']}"

    din = synth_packed(args.n, args.len, dev, seed=54321 if cfg5 else 12345, n_rate=args.n_rate)
    hp = HotPath(din, args, dev, rank, world)

    for i in range(args.warmup):
        hp.step(seed=i)                 # the data/mimic seed is shared by all ranks: every rank builds the same store
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        hp.step(seed=args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    assert not bool(hp.edit_overflow.item()), "mimic edit buffer bound exceeded: the run is invalid"
    validation = hp.validate()
    exchange_ms = None
    if args.exchange:
        if cfg5:
            assert tuple(hp.latent.shape) == (args.n, 64) and bool(torch.isfinite(hp.latent).all())
            validation.update(latent_shape=list(hp.latent.shape), latent_finite=bool(torch.isfinite(hp.latent).all()),
                              latent_row_norm_max=float(hp.latent.norm(dim=1).max().item()))
        else:
            validation.update(check_votes(hp.gathered, args.voters, args.n, args.n_clusters))
    else:
        # the exchange step of the path, untimed at N = 1 with one voter: this rank's voter predicts, assignments are all-gathered
        from idelucs_amd.dist import all_gather_assignments
        all_gather_assignments(hp.predict(hp.model, hp._predict_inputs()))            # first call: library initialisation for the inference GEMM shapes (0.5 s)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        gathered = all_gather_assignments(hp.predict(hp.model, hp._predict_inputs()))
        torch.cuda.synchronize()
        exchange_ms = 1e3 * (time.perf_counter() - t1)
        assert tuple(gathered.shape) == (world, args.n)

    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        V = args.voters
        value = args.n * V / (elapsed / args.steps)
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
        plane_kernels = plane_kernel_leg(dev, 2 * B, H1, F) if world == 1 else None
        cfg_name = "configs[4] (cfg5)" if cfg5 else ("configs[1] (cfg2)" if V == 1 else "configs[2] (cfg3)")
        region = "device mimic sites + vectorise + scaler fit + 1 epoch per voter"
        if args.exchange:
            region += (" + last voter's weights broadcast + predict sharded by sequence + all-gather of fp32 latent shards" if cfg5
                       else " + predict + all-gather of the int32 assignments [V, N]")
        else:
            region += " (BASELINE.md section 3 region; predict + all-gather run once after it: exchange_ms)"
        job = ("cfg5: %d voter(s), sharded predict + latent all-gather" % V if cfg5 else
               ("cfg2: 1 voter (BASELINE.md section 3 region, no exchange inside)" if V == 1 and not args.exchange else
                "cfg3: %d voters x 1 epoch + predict + all-gather of int32 [V, N]" % V))
        out = {
            "metric": "sequences/sec (k-mer vectorise + 1 epoch), k=6 batch 512",
            "job": job + ("" if world == 1 or cfg5 else " -- compare with the N = 1 line's fixed_job_8_voters, not with its one-voter value"),
            "value": value, "unit": "sequences/sec", "n_gpus": world, "ranks": group["ranks"], "backend": group["backend"],
            "devices": group["devices"], "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak" if V == world and world > 1 and cfg5 else ("strong" if world > 1 else "weak"),
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "dtype_note": ("f32 storage, activations, accumulators and optimizer state throughout; the step's two big products (layer 1, dW1) run on the fp16 matrix "
                           "cores from operands kept as TWO fp16 planes each (22 significand bits), three products with fp32 accumulators: fp32-grade against float64 "
                           "-- within 7e-7 of the product's largest entry on standardised operands, closer than the fp32 arithmetic on the step's long dense sums (2.4e-7 "
                           "against 2.6e-6; dW1 4.5e-7 against 1.5e-6), up to twice as far on sums carried by a few terms (tests/test_gpu_planes.py, "
                           "profiles/r06_planes_adversarial.txt); an operand beyond the planes' range is flagged and the voter retrained on the fp32 tiles; the plain "
                           "fp32 form of the same step is timed in fp32_step_form"),
            "config": {"workload": f"BASELINE {cfg_name}{' (variant N: every base an N w.p. %g)' % args.n_rate if args.n_rate > 0 else ''}: "
                                   f"synthetic {args.n} x {args.len} bp, k={args.k}, n_clusters={args.n_clusters}, "
                                   f"n_mimics={args.n_mimics} ({P} views), batch_sz={args.batch_sz}, NetLinear fp32, RMSprop; "
                                   f"{V} voter(s) over {world} GPU(s), value = N_seq * voters / wall; timed = {region}",
                       "n_sequences": args.n, "seq_len": args.len, "k": args.k, "batch_sz": args.batch_sz, "n_voters": V,
                       "optimizer_steps_per_epoch": (args.n * args.n_mimics + args.batch_sz - 1) // args.batch_sz,
                       "parallelism": f"{V} voters / {world} ranks" + (f", batches of {len(hp.lanes)} voters in lockstep" if len(hp.lanes) > 1 else "")},
            "roofline": {"kernel": "vectorise kernel (hand-written HIP: one count + per-view window deltas + normalise, all views)",
                         "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": pmc_traffic_gb() if not cfg5 else None,
                         "traffic_source": (f"profiles/{PMC_PROFILE} (rocprofv3 --pmc passes of this kernel at this workload, per launch; "
                                            "not counters of this run)") if not cfg5 else None,
                         "ms_per_launch": t_vec, "bytes_per_seq_algorithmic": b_vec,
                         "share_of_step": t_vec / ms_step},
            "roofline_stage1": {"stage": "device mimic sites + vectorise (all views) + scaler fit: everything between the packed bases and the epoch",
                                "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                "ms": (lambda t: t)(hp.mean_ms("edits", args.warmup) + t_vec + hp.mean_ms("stats", args.warmup)),
                                "achieved": args.n * b_vec / ((hp.mean_ms("edits", args.warmup) + t_vec + hp.mean_ms("stats", args.warmup)) * 1e-3) / 1e9,
                                "frac": args.n * b_vec / ((hp.mean_ms("edits", args.warmup) + t_vec + hp.mean_ms("stats", args.warmup)) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "note": "the stage's algorithmic bytes are the vectoriser's (SURVEY 8d); the site generator is Philox-bound "
                                        "compute, the scaler fit reads view 0 once more (1.64 GB at cfg2)"},
            "roofline_epoch": {"kernel": "training epoch (seven launches a step: layer 1 and dW1 on the fp16 matrix cores from two fp16 planes per operand -- three products, fp32 accumulators, own tiles: l1_planes_kernel, wgrad_dplanes_rms_kernel (both operands by LDS-DMA: mid_bwd writes dr1 as planes) --, the sum of the K-slice partials, the middle, InfoNCE, IIC; IDELUCS_PLANES=0 and steps of other shapes: own fp32-MFMA tiles; the fixed job's lockstep voters run the same seven launches with a voter index in the grid)", "bound": "mfma",
                               "achieved": ach_exec, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_exec / MFMA_F32_PEAK_TFLOPS,
                               "ms": t_ep, "flop_per_seq_executed": f_exec, "share_of_step": t_ep * len(hp.my_voters) / ms_step,
                               "flop_per_seq_algorithmic": f_ep, "achieved_algorithmic": ach_ep, "frac_algorithmic": ach_ep / MFMA_F32_PEAK_TFLOPS,
                               "plane_kernels": plane_kernels,
                               "note": "frac = EXECUTED flops (SURVEY 8(d)'s count without the input gradient of layer 1, which is not computed) over the "
                                       "dense FP32 matrix peak -- the arithmetic the results are equivalent to (tests/test_gpu_planes.py: fp32-grade "
                                       "against float64); frac_algorithmic = SURVEY 8(d)'s own count (3 x forward for every layer) over the same peak, kept for "
                                       "comparison with earlier rounds.  Neither is a fraction of the pipe the two big products run on: each is three fp16 "
                                       "products (3 x 4.29 GFLOP a step) on the fp16 matrix cores -- plane_kernels has each kernel's time alone and its "
                                       "fraction of the 2.5 PFLOP/s fp16 peak; profiles/r06_step_kernels.txt their times inside the step"},
            "roofline_dominant": "roofline_epoch" if t_ep * len(hp.my_voters) > 0.5 * ms_step else "roofline",
            "roofline_dominant_note": ("`roofline` is the north_star's named kernel (the hand-written HBM-bound vectoriser); the epoch is "
                                       "%.0f %% of the timed step and its own object, roofline_epoch, is the one that prices the step"
                                       % (100.0 * t_ep * len(hp.my_voters) / ms_step)),
            "stage_ms": {k: hp.mean_ms(k, args.warmup) for k in hp.ev if hp.ev[k]},
            "validation": validation,
        }
        if exchange_ms is not None:
            out["exchange_ms"] = exchange_ms
        if world == 1 and args.e2e and not cfg5 and V == 1:
            del hp.feats, hp.store.feats
            hp.model.store = None
            torch.cuda.empty_cache()
            out["t_e2e"] = t_e2e(args, dev)
        if world == 1 and args.fixed_job and not cfg5 and V == 1 and not args.exchange:
            # (after t_e2e dropped the headline's feature store: the fixed job allocates its own)
            if "t_e2e" not in out:
                del hp.feats, hp.store.feats
                hp.model.store = None
            torch.cuda.empty_cache()
            out["fixed_job_8_voters"] = fixed_job_8_voters(din, args, dev, rank, world)
            if args.prediction:
                out["predicted_fixed_job"] = predicted_fixed_job(din, args, dev, rank, world, out["fixed_job_8_voters"]["stage_ms"], t_ep, lane_model=hp.model)
        if world == 1 and args.k_sweep and not cfg5 and V == 1:
            out["k_sweep"] = k_sweep(din, args, dev)
        if world == 1 and args.split16 and not cfg5 and V == 1 and args.k == 6:
            out["fp32_step_form"] = fp32_form_leg(din, args, dev, rank, world)
        if world == 1 and args.cfg5_leg and not cfg5 and V == 1 and not args.exchange and args.fixed_job:
            del din, hp
            torch.cuda.empty_cache()
            out["cfg5_one_gpu"] = cfg5_one_gpu(args, dev, rank, world)
        if world == 1 and args.cpu_base:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
