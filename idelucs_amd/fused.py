"""idelucs_amd.fused -- the explicit, fused optimizer step for the default configuration
(NetLinear encoder + RMSprop), replayed as a HIP graph.

One step of reference idelucs/models.py:117-133 is 6 launches by default: the layer-1 product on hipBLASLt (torch.mm with
out=, no allocation) and five fused HIP kernels from csrc/train_step.hip + nce_fused.hip + wgrad_device.h --
    W1 x^T  ->  idl_mid_fwd_gather  ->  idl_nce_fused_iic_z (two launches)  ->  idl_mid_bwd_gather
            ->  idl_wgrad_rmsprop_step (dW1 = dr1^T x as MFMA tiles with RMSprop in their epilogue + the rest of the optimizer)
(DESIGN.md 4.4 has the table of what each launch carries).  The unfused building blocks idl_relu_dropout_fwd, idl_head_fwd,
idl_nce_rows, idl_iic_core, idl_head_bwd, idl_bias_grads, idl_rmsprop_step remain for the shapes the fused kernels do not take
(n_clusters > 48, partial batches) and as their test references.  Batches are assembled from the HBM feature store at a
device-resident offset that the optimizer kernel advances, the next batch while the current step is between its two big GEMMs
(two x buffers), so an epoch is n_batches / 2 replays of one captured two-step graph with no host work between.

The parameters remain the nn.Parameters of model.net (state_dict / predict / weights_init unchanged).
RMSprop state lives here; begin_voter() clears it (every voter is an independent run, models.IID_model.begin_voter).

BatchedLinearTrainer steps several voters of one ensemble in lockstep: the layer-1 product becomes a batched GEMM and each of
the five kernels ONE launch with the voter index in its grid (recorded launches, idl_plan_*; the dW1 tiles ride at the head of
every voter's share of the optimizer launch as they do for a single voter: 5 voters 94.9 -> 91.7 ms an epoch of 50 000 sequences), so the latency-bound
launches -- 51 of the 116 us of a step, mostly launch boundaries and dependent-load chains on a quarter of the CUs -- are paid
once per step of the whole batch of voters instead of once per voter.
"""
import ctypes
import os
import sys

import torch

from . import _lib
from ._lib import lib as _L

EPS = sys.float_info.epsilon
TEMPERATURE = 0.85          # hard-coded at the reference call site, models.py:128


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


_PLANES_DISABLED = False


def planes_default():
    """Whether a new trainer takes the two-plane step form: IDELUCS_PLANES (default on) unless a run of this process already left the planes' range
    (disable_planes: training.train_voter's fall-back)."""
    return os.environ.get("IDELUCS_PLANES", "1") != "0" and not _PLANES_DISABLED


# Launch-sequence variants kept because tests compare them with the default form (each was measured and not adopted: DESIGN, History).  They are
# NOT environment switches (round 6: 27 of them used to be): a test sets an entry with monkeypatch.setitem(fused.VARIANTS, ...) before it builds a trainer.
VARIANTS = {
    "nce_fused": "1",        # 0: S = f f^T as a GEMM + row kernels instead of the fused InfoNCE passes
    "dw3_partial": "1",      # 0: dW3 as a GEMM instead of stacked partials from the middle-backward launch (n_clusters <= 48)
    "test_cold": "0",        # 1: a 512 MB fill in front of the hand-scheduled launches (tests/test_gpu_planes.py::test_cold_caches_*)
    "lockstep_planes": "1",  # 0: a rank's voters in lockstep on batched fp32 library GEMMs
    "planes_wgrad": "1",     # 0: dW1 on the fp32 tiles (writing W1's planes) beside the two-plane layer 1
    "planes_tail": "wgrad",  # reduce: the optimizer tail beside the next step's partial sums instead of on the dW1 tiles' loader waves
    "mid_fused": "1", "pipeline": "1", "dw2_inlaunch": "1", "overlap": "0", "early_gather": "1", "gather_split": "4", "transposed_l1": "1",
    "l1_fused": "0",         # bare: the fp32 form's own layer-1 tiles as a plain product without the riding tail (the comparison of test_tail_riding_*)
    "joint_inlaunch": "1", "wgrad_fused": "1", "keep_w1_grad": "0", "steps_per_graph": "16",
    "tail_l1": "1",          # 0 (fp32 form): the optimizer tail behind the dW1 tiles instead of riding in the next step's layer-1 launch
}


VARIANTS["fused"] = "1"      # 0: models.IID_model trains NetLinear + RMSprop through torch autograd instead of the fused explicit step
VARIANTS.update({k: v for k, v in _lib.DEV.items() if k in VARIANTS})


def _v(name):
    return VARIANTS[name]


def disable_planes():
    """From here on every trainer of this process runs the fp32 tiles (the data left the fp16 planes' range once: it will again)."""
    global _PLANES_DISABLED
    _PLANES_DISABLED = True


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _Buffers:
    """Activations / gradients of one batch shape (m = 2*B rows)."""

    def __init__(self, m, F, H1, H2, C, dev, shared=None):
        """shared: {'xs0', 'xs1', 'r1', 'dr1'} = this voter's slices of a BatchedLinearTrainer's stacked GEMM operands."""
        f32 = dict(dtype=torch.float32, device=dev)
        sh = shared or {}
        self.m = m
        # the batch of this step / the next one being assembled
        self.xs = [sh['xs0'] if 'xs0' in sh else torch.empty((m, F), **f32), sh['xs1'] if 'xs1' in sh else torch.empty((m, F), **f32)]
        self.x = self.xs[0]
        self.r1 = sh['r1'] if 'r1' in sh else torch.empty((m, H1), **f32)        # Linear1 output, then ReLU+Dropout in place
        self.r1T = self.r1.view(H1, m)                # the same memory as the transposed image [H1, m] (one of the two is in use)
        # tail-in-layer-1 (FusedLinearTrainer._tail_l1): the layer-1 launch of step t + 1 writes its activations while riders of the
        # same launch still read step t's (dW2 = dlat^T r1), so the steps alternate between two images; made on first use
        self._r1_other = None
        self._H1 = H1
        self.lat = torch.empty((m, H2), **f32)
        self._dims = (m, H2, dev)
        self.f = torch.empty((m, H2), **f32)
        self.inv = torch.empty((m,), **f32)
        self.r2 = torch.empty((m, H2), **f32)
        self.z = torch.empty((m, C), **f32)
        self.S = torch.empty((m, m) if not (_L.idl_nce_fused_workspace(m) > 0 and _v("nce_fused") != "0") else (1, 1), **f32)
        self.lse = torch.empty((m,), **f32)
        self.loss_rows = torch.empty((m,), **f32)
        self.nce_parts = _L.idl_nce_fused_parts()
        self.nce_ws_bytes = _L.idl_nce_fused_workspace(m)            # -1: shape not supported by the fused InfoNCE kernels
        self.nce_fused = self.nce_ws_bytes > 0 and _v("nce_fused") != "0"
        self.G = torch.empty((self.nce_parts if self.nce_fused else 1, m, H2), **f32)
        self.nce_ws = torch.empty(max(self.nce_ws_bytes, 4) // 4, **f32)
        self.P0 = torch.empty((C, C), **f32)
        self.iic_scratch = torch.empty((C * C + 2 * C,), **f32)
        self.dlogits = torch.empty((m, C), **f32)
        self.dzs = torch.empty((m, C), **f32) if C > 48 else None       # z dP0 (fine-grained mode)
        self.dlat = torch.empty((m, H2), **f32)
        self.dr1 = sh['dr1'] if 'dr1' in sh else torch.empty((m, H1), **f32)


def _r1_of(bf, xi):
    """The layer-1 activation image of a step of parity xi ([m, H1]; the transposed image is the same memory)."""
    if xi == 0:
        return bf.r1
    if bf._r1_other is None:
        bf._r1_other = torch.empty_like(bf.r1)
    return bf._r1_other


def _planes_of(bf, F):
    """Buffers of the two-plane step form (FusedLinearTrainer._planes), made on first use: the two batches' fp16 planes and the
    two images of idl_l1_planes' K-slice partial sums [8][H1][m] (slab 0 of an image becomes the step's transposed activations)."""
    if getattr(bf, "_planes", None) is None:
        m, H2, dev = bf._dims
        h16 = dict(dtype=torch.int16, device=dev)
        bf._planes = {"xh": [torch.empty((m, F), **h16) for _ in range(2)], "xl": [torch.empty((m, F), **h16) for _ in range(2)],
                      "part": [torch.empty((int(_L.idl_l1_planes_parts()), bf._H1, m), dtype=torch.float32, device=dev) for _ in range(2)],
                      "dh": torch.empty((m, bf._H1), **h16), "dl": torch.empty((m, bf._H1), **h16),      # dr1 as planes (mid_bwd -> the dW1 tiles)
                      "valid": [False, False],          # valid[i]: xh[i] / xl[i] hold the planes of the batch in bf.xs[i]
                      "x32": [True, True]}              # x32[i]: bf.xs[i] itself holds that batch (False: it was assembled as planes only)
    return bf._planes


class _Recorder:
    """step_on_batch in record mode: the big products are skipped (the batched trainer runs them as batched GEMMs) and every
    kernel launch is recorded as a plan (idl_plan_begin / idl_plan_end) instead of being performed."""

    def __init__(self):
        self.plans = []          # uint8 host tensors [PLAN_BYTES], in launch order
        self.mms = 0

    def launch(self, fn, *args):
        blob = torch.zeros(int(_L.idl_plan_bytes()), dtype=torch.uint8)
        _lib.check(_L.idl_plan_begin(ctypes.c_void_p(blob.data_ptr())))
        rc = fn(*args)
        end = _L.idl_plan_end()
        _lib.check(rc)
        _lib.check(end)
        self.plans.append(blob)


class FusedLinearTrainer:
    def __init__(self, net, lr, weight, lamb, weight_decay=0.01, alpha=0.99, eps=1e-8, seed=0, grad_w1=None, shared_buffers=None):
        lin1, lin2, lin3 = net.layers[0], net.layers[3], net.classifier[2]
        self.net = net
        self.W1, self.b1, self.W2, self.b2, self.W3, self.b3 = (lin1.weight, lin1.bias, lin2.weight, lin2.bias,
                                                                  lin3.weight, lin3.bias)
        self.params = [self.W1, self.b1, self.W2, self.b2, self.W3, self.b3]
        self.dev = self.W1.device
        self.F, self.H1, self.H2, self.C = lin1.in_features, lin1.out_features, lin2.out_features, lin3.out_features
        if self.H2 != 64 or self.C > 256 or lin2.in_features != self.H1 or lin3.in_features != 64:
            raise ValueError("FusedLinearTrainer needs NetLinear (latent 64) and n_clusters <= 256")
        # bias gradients are kept as COL_PARTS stacked partial column sums, added up inside idl_rmsprop_step
        self.parts = [1 if p.dim() == 2 else _L.idl_col_sum_parts() for p in self.params]
        # the last layer's weight gradient (C x 64) is produced as partials by the bias-gradient launch when C <= 48
        self._dw3_partial = self.C <= 48 and _v("dw3_partial") != "0"
        if self._dw3_partial:
            self.parts[4] = _L.idl_col_sum_parts()
        self.grads = [torch.zeros((q,) + tuple(p.shape), dtype=p.dtype, device=p.device) if q > 1 else torch.zeros_like(p)
                      for p, q in zip(self.params, self.parts)]
        if grad_w1 is not None:                  # this voter's slice of a BatchedLinearTrainer's stacked dW1
            assert tuple(grad_w1.shape) == tuple(self.W1.shape) and grad_w1.is_contiguous()
            self.grads[0] = grad_w1
        self._shared_buffers = shared_buffers    # {m: {...}} views handed to _Buffers
        self._rec = None                         # a _Recorder while a BatchedLinearTrainer records this voter's launches
        self.square_avg = [torch.zeros_like(p) for p in self.params]
        self.weight, self.lamb, self.seed = float(weight), float(lamb), int(seed) & (2 ** 64 - 1)
        self.hyper = torch.tensor([lr, alpha, eps, weight_decay, 1.0 - alpha], dtype=torch.float32, device=self.dev)
        self.ctl = torch.zeros(2, dtype=torch.int64, device=self.dev)        # [step counter, batch offset]
        self.out = torch.zeros(4, dtype=torch.float32, device=self.dev)      # [step loss, running sum, nce, iic]
        self._bufs = {}
        self._graphs = {}
        self._side = torch.cuda.Stream(device=self.dev)     # second branch of the step (see step_on_batch)
        # TEST HOOK (IDELUCS_DEV=test_cold=1; tests/test_gpu_planes.py): a 512 MB fill in front of the step's middle and of each plane kernel, so that every load of
        # the hand-scheduled kernels comes from HBM instead of a warm L2/MALL -- a load consumed before its wait is right when it landed
        # early and wrong when it did not (DESIGN.md History, round 5), and only cold caches show that
        self._cold = _v("test_cold") == "1"
        self._cold_buf = None
        # Round 5, default (IDELUCS_PLANES=0: the fp32 tiles below; csrc/planes.h): the two big products on the fp16 matrix cores from operands kept
        # as two fp16 planes (22 significand bits a factor, three products, fp32 accumulators: closer to a float64 product than an fp32 GEMM) --
        # the batch's planes written by the workgroups that assemble it, W1's by the epilogue of the dW1 tiles that update it.  A step
        # 100.6 us against 111.0 at cfg2 (tools/bench_planes.py).  Needs the default launch sequence of a single voter (tail-in-layer-1),
        # m % 128 == 0 and F % 512 == 0; any other step runs the fp32 tiles.
        self._planes = planes_default()
        self._planes_reduce_launch = True        # (round 6: the variant in which mid_fwd added the eight partial sums itself -- +8 us -- was removed)
        # ... and the same for a rank's voters in lockstep (BatchedLinearTrainer; IDELUCS_DEV=lockstep_planes=0: their products as batched fp32
        # library GEMMs instead): the six launches of the two-plane step recorded per voter and run once for all of them, blockIdx.y = voter
        # -- the lone voters' steps bit for bit (tests/test_gpu_planes.py), 47.5 / 45.4 / 43.6 ms a voter-epoch in batches of 2 / 4 / 8
        # against 54.2 alone (fp32 GEMMs: 58.9 / 54.8 / 52.6)
        self._planes_lockstep = _v("lockstep_planes") != "0"
        # ... and dW1 from the batch's planes too (csrc/wgrad_planes.hip); the assembling workgroups then write the planes ONLY
        self._planes_wgrad = _v("planes_wgrad") != "0"
        # ... whose loader waves run the step's optimizer tail under the tiles' epilogue (IDELUCS_DEV=planes_tail=reduce: the tail beside the
        # next step's partial sums instead, 9.4 us for that launch against 4.7)
        self._planes_tail_wgrad = _v("planes_tail") != "reduce"
        self._cus = torch.cuda.get_device_properties(self.dev).multi_processor_count if self.dev.type == "cuda" else 0
        self._ctl_snap = torch.zeros(1, dtype=torch.int64, device=self.dev)       # the step counter as the step's reduce launch saw it (idl_wgrad_xplanes_rms)
        self._w1_planes = None                   # (W1 hi, W1 lo, overflow flag)
        # the words of dr1's scale (csrc/planes.h): [0] the exponent mid_bwd gave this step's planes, [1] the next step's (the dW1 launch derives it), [4..] maxima
        self._dr1_scale = torch.zeros(int(_L.idl_dr1_scale_words()), dtype=torch.int32, device=self.dev)
        self._dr1_scale[1:2].fill_(int(_L.idl_planes_exponent(2)))
        self._w1_planes_fresh = False
        # the layers between the two big GEMMs as one 1024-thread MFMA kernel per direction (idl_mid_fwd / idl_mid_bwd)
        self._mid_fused = self.H1 == 512 and _v("mid_fused") != "0"
        self._pipeline = _v("pipeline") != "0"   # optimizer launch also assembles the next batch
        # dW2 = dlat^T r1 as 16 x 16 MFMA tiles inside the optimizer launch (idl_rmsprop_step_gather_wgrad) instead of a GEMM launch
        self._dw2_inlaunch = self.H1 % 16 == 0 and _v("dw2_inlaunch") != "0"
        self._overlap = _v("overlap") != "0"   # measured: no gain inside a HIP graph on ROCm 7.2
        # the NEXT batch is assembled by spare workgroups of the mid-forward / mid-backward launches into a second x buffer (instead
        # of by the optimizer launch, where it competed with RMSprop for HBM): needs the fused middle kernels and n_clusters <= 48
        self._early_gather = (self._pipeline and self._mid_fused and self.C <= 48 and self.F % 4 == 0
                              and _v("early_gather") != "0")
        self._early_split = _v("early_gather") != "2"       # 2: all of it in the mid-backward launch
        # n_clusters > 48 (fine-grained mode, 200 outputs): the backward runs as separate kernels, so ALL of the next batch's tiles ride
        # in the mid-forward launch
        self._early_fwd = (self._pipeline and self._mid_fused and self.C > 48 and self.F % 4 == 0 and self._dw2_inlaunch
                           and _v("early_gather") != "0")
        self._gsplit = min(max(int(_v("gather_split")), 0), 8)   # eighths of the tiles the mid-forward launch takes
        # layer-1 activations kept transposed ([512, m]) between the layer-1 product and its consumers
        self._transposed_l1 = _v("transposed_l1") != "0"
        # IDELUCS_DEV=l1_fused=bare: the fp32 form's own layer-1 tiles as a plain product (no riding tail) in place of the library GEMM, mid_fwd unchanged.
        # (Round 4's variant with bias / ReLU / Dropout and the K-split of Linear(512, 64) in those tiles' epilogue -- 114.7 us a step against 111.8: the
        #  mid-forward launch is as long as the batch assembly riding in it, not as its head -- left the library in round 6: DESIGN, History.)
        self._l1_bare = _v("l1_fused") == "bare"
        # opt-in: InfoNCE pass 2 + IIC core inside the mid-backward launch (one boundary less, but the InfoNCE tiles then run on
        # the 64 CUs of that launch instead of 256: the fused launch takes 32.8 us against 9.5 + 13.5 -- measured +8 us per step)
        self._nce_bwd_fused = False              # (round 6: InfoNCE pass 2 + the IIC core inside the mid-backward launch -- +8 us a step -- was removed)
        self._joint_inlaunch = _v("joint_inlaunch") != "0"   # IIC joint inside the InfoNCE pass-1 launch
        # dW1 on this package's own MFMA tiles with RMSprop in their epilogue, as the head of the optimizer launch
        # (csrc/wgrad_device.h, idl_wgrad_rmsprop_step): one launch instead of hipBLASLt's GEMM + the optimizer launch, and the 8 MB
        # gradient never goes to memory.  IDELUCS_DEV=wgrad_fused=0: hipBLASLt + optimizer launch; =2: the tiles as a launch of their own
        self._wgrad_fused = _v("wgrad_fused") != "0"
        self._wgrad_own_launch = _v("wgrad_fused") == "2"
        self._keep_w1_grad = _v("keep_w1_grad") != "0"       # tests: also write dW1 to grads[0]
        self._steps_per_graph = max(2, int(_v("steps_per_graph")) // 2 * 2)
        # Round 5 (IDELUCS_DEV=tail_l1, default on): the layer-1 product on this package's own tiles (idl_l1_fwd: no library build decides
        # its speed) and the optimizer's TAIL -- the dW2 tiles, the small tensors, step loss, step counter: 5.8 us behind the dW1 tiles
        # of the optimizer launch, where they cannot become resident beside a tile -- riding in the layer-1 launch of the NEXT step
        # (idl_l1_fwd_rms), where they have 30 us of slack.  A step then ends with the dW1 tiles alone; its tail is pending until the
        # next step's first launch, or flush_tail().  Only the default launch sequence of a single voter takes it.
        self._tail_l1 = _v("tail_l1") != "0"
        self._pending = None                     # (buffers, parity) of the step whose tail has not run yet
        self._perm = None
        n = len(self.params)
        self._pp = (ctypes.c_void_p * n)(*[p.data_ptr() for p in self.params])
        self._gp = (ctypes.c_void_p * n)(*[g.data_ptr() for g in self.grads])
        self._vp = (ctypes.c_void_p * n)(*[v.data_ptr() for v in self.square_avg])
        self._sz = (ctypes.c_int64 * n)(*[p.numel() for p in self.params])
        self._parts = (ctypes.c_int32 * n)(*self.parts)
        self._sz_no_w1 = (ctypes.c_int64 * n)(*([0] + [p.numel() for p in self.params[1:]]))     # W1 updated by idl_wgrad_rmsprop

    def begin_voter(self, voter, keep_state=False):
        """Dropout stream of voter v: the Philox counter word the kernels take from ctl[0] (its low 32 bits) starts at
        v << 24, so voters never share masks whichever rank runs them (16.7 M optimizer steps per voter, 256 voters).
        keep_state: the previous voter's RMSprop running averages stay (IDELUCS_VOTER_STATE=carry, models.IID_model)."""
        # (fill_ on a view, here and below: `tensor[i] = python_scalar` is a host-to-device copy from pageable memory, which holds the
        #  host until everything queued on the stream has run -- the vectoriser, when a voter begins right behind the store's build)
        self.ctl[0:1].fill_((int(voter) & 0xFF) << 24)
        self._dr1_scale[1:2].fill_(int(_L.idl_planes_exponent(2)))       # (a voter's first step does not inherit the last voter's gradient range)
        if keep_state:
            return
        for v in self.square_avg:               # a voter starts with fresh optimizer state (models.IID_model.begin_voter)
            v.zero_()

    def gradient(self, i):
        """Gradient of parameter i as a tensor of the parameter's shape (sums the stacked partials)."""
        return self.grads[i].sum(0) if self.parts[i] > 1 else self.grads[i]

    def set_lr(self, lr):
        self.hyper[0:1].fill_(float(lr))

    def buffers(self, m):
        if m not in self._bufs:
            self._bufs[m] = _Buffers(m, self.F, self.H1, self.H2, self.C, self.dev, shared=(self._shared_buffers or {}).get(m))
        return self._bufs[m]

    def _k(self, fn, *args):
        """One kernel launch of the step: performed, or recorded while a BatchedLinearTrainer is recording."""
        if self._rec is not None:
            self._rec.launch(fn, *args)
        else:
            _lib.check(fn(*args))

    def _mm(self, a, b, out):
        if self._rec is not None:
            self._rec.mms += 1                   # the batched trainer runs the big products as batched GEMMs
        else:
            torch.mm(a, b, out=out)

    # ------------------------------------------------------------------ one step on a filled bf.x
    @torch.no_grad()
    def step_on_batch(self, bf, train=True, batch_advance=0, next_from=None, xi=0, defer_tail=False):
        """Forward, backward and RMSprop update for the [m, F] batch in bf.x (rows [0,m/2) "true",
        [m/2,m) "modified").  Only enqueues work on the current stream.  next_from = a FeatureStore: the batch
        offset is advanced in the middle of the step and the optimizer launch also assembles the NEXT batch into
        bf.x (both are memory-bound and independent: one launch instead of two)."""
        m, C, tr = bf.m, self.C, 1 if train else 0
        x = bf.xs[xi]
        tl = (self._transposed_l1 and next_from is not None and self._early_gather and self._early_split and m % 16 == 0
              and self._dw2_inlaunch)
        early = next_from is not None and self._early_gather and m % 16 == 0   # next batch -> bf.xs[1 - xi] by the mid launches
        early_f = next_from is not None and self._early_fwd and m % 16 == 0    # ... by the mid-forward launch alone
        if (self._rec is not None and self._planes and self._planes_lockstep and tl and early and self._early_split and not self._nce_bwd_fused
                and not self._l1_bare and self._wgrad_fused and not self._wgrad_own_launch and not self._overlap
                and self._joint_inlaunch and self._dw3_partial and bf.nce_fused and C <= 48
                and bool(_L.idl_l1_planes_supported(m, self.H1, self.F)) and bool(_L.idl_wgrad_xplanes_supported(m, self.H1, self.F))
                and 144 <= (self.H1 // 64) * (self.F // 128) <= self._cus and next_from.n < 60_000_000):
            return self._record_planes_step(bf, tr, next_from, xi)
        # tail-in-layer-1: own layer-1 tiles, the dW1 tiles as the step's last launch, the rest of the optimizer in the NEXT layer-1 launch
        tm = (self._tail_l1 and tl and early and self._early_split and self._rec is None and not self._l1_bare
              and not self._nce_bwd_fused and self._wgrad_fused and not self._wgrad_own_launch and not self._shared_buffers
              and self._dw2_inlaunch and not self._overlap
              and bool(_L.idl_l1_fwd_supported(m, self.H1, self.F)) and bool(_L.idl_wgrad_supported(m, self.H1, self.F)))
        if not tm:
            self.flush_tail()                   # (a step of another form: whatever is pending goes first)
        # ... with the layer-1 product from two-plane operands (IDELUCS_PLANES=1)
        pl = (tm and self._planes and bool(_L.idl_l1_planes_supported(m, self.H1, self.F)) and next_from.n < 60_000_000)
        plw = pl and self._planes_wgrad and bool(_L.idl_wgrad_xplanes_supported(m, self.H1, self.F))
        dpl = plw                              # ... with dr1 written as planes by mid_bwd: both operands of dW1 reach its tiles by LDS-DMA (m % 128 == 0 here)
        # ... and the same two products in the step of n_clusters > 48 (the fine-grained mode's 200 output units: separate backward kernels, the
        # whole batch assembled by the mid-forward launch, activations NOT transposed): the layer-1 tiles with the operands' roles swapped
        # give part[8][m][512]; the dW1 kernel with the tail on its loader waves ends the step
        plf = (self._planes and self._planes_wgrad and self._planes_reduce_launch and self._planes_tail_wgrad and early_f and not early
               and not tm and self._rec is None and self._wgrad_fused and not self._wgrad_own_launch and self._dw2_inlaunch
               and not self._shared_buffers and bool(_L.idl_l1_planes_supported(self.H1, m, self.F))
               and bool(_L.idl_wgrad_xplanes_supported(m, self.H1, self.F)) and 144 <= (self.H1 // 64) * (self.F // 128) <= self._cus
               and next_from.n < 60_000_000)
        pb = _planes_of(bf, self.F) if (pl or plf) else None
        if not (plw or plf) and getattr(bf, "_planes", None) is not None and not bf._planes["x32"][xi]:
            raise RuntimeError("the batch in this buffer was assembled as planes only: a step of another form cannot read it")
        if pl:
            r1 = pb["part"][xi][0]              # [H1, m]: slab 0 of the partial sums, where mid_fwd leaves the activations
            self._prepare_planes(bf, pb, xi)
        elif plf:
            r1 = pb["part"][xi].view(-1, m, self.H1)[0]      # [m, H1]: slab 0 of part[8][m][512]
            self._prepare_planes(bf, pb, xi)
        else:
            r1 = _r1_of(bf, xi) if tm else bf.r1
            self._w1_planes_fresh = False       # (this step updates W1 without its planes)
            if getattr(bf, "_planes", None) is not None:
                bf._planes["valid"][1 - xi] = False      # (... and assembles the next batch without its planes)
        r1T = r1.view(self.H1, m)
        chk = _lib.check
        main = torch.cuda.current_stream()
        side = self._side if self._overlap else main
        # ---- forward
        # shares (eighths) of the next batch's assembly: [0, g2) the mid-forward launch, [g2, 8) mid-backward
        g1 = 0
        g2 = self._gsplit
        if pl:
            # a1^T = W1 x^T as eight K-slice partial sums on the fp16 matrix cores, then ONE launch that adds the eight up on every CU (and, in the
            # variant that keeps the tail off the dW1 tiles' loader waves, runs the previous step's optimizer tail beside that)
            wh, wl, _ = self._w1_planes
            if self._cold:
                self._evict()
            chk(_L.idl_l1_planes(_p(wh), _p(wl), self.F, _p(pb["xh"][xi]), _p(pb["xl"][xi]), self.F, m, self.H1, self.F, _p(pb["part"][xi]), _stream()))
            if self._pending is not None:
                pbf, pxi, pr1 = self._pending
                self._tail_launch(pbf, pxi, pr1, red=(pb["part"][xi], self.H1 * m))
            else:
                chk(_L.idl_reduce_parts_rms(_p(pb["part"][xi]), self.H1 * m, _p(self.ctl), _p(self._ctl_snap), 0, None, None, None, None, None, None, None, None, 0, 0.0, 0.0, None, 0,
                                            -1, None, None, 0, 0, 0, 0, None, 0, _stream()))
        elif tm:    # a1^T = W1 x^T on own tiles; the previous step's optimizer tail rides in the same launch
            if self._pending is not None:
                pbf, pxi, pr1 = self._pending
                self._tail_launch(pbf, pxi, pr1, l1=(x, m, r1T))
            else:
                chk(_L.idl_l1_fwd(_p(self.W1), _p(x), m, self.F, _p(r1T), _stream()))
        elif plf:   # a1 = x W1^T as eight K-slice partial sums [8][m][512] (the tiles of idl_l1_planes with the operands' roles swapped), then their sum
            wh, wl, _ = self._w1_planes
            if self._cold:
                self._evict()
            chk(_L.idl_l1_planes(_p(pb["xh"][xi]), _p(pb["xl"][xi]), self.F, _p(wh), _p(wl), self.F, self.H1, m, self.F, _p(pb["part"][xi]), _stream()))
            chk(_L.idl_reduce_parts_rms(_p(pb["part"][xi]), self.H1 * m, _p(self.ctl), _p(self._ctl_snap), 0, None, None, None, None, None, None, None, None, 0, 0.0, 0.0, None, 0,
                                        -1, None, None, 0, 0, 0, 0, None, 0, _stream()))
        elif tl and self._l1_bare and self._rec is None and bool(_L.idl_l1_fwd_supported(m, self.H1, self.F)):
            chk(_L.idl_l1_fwd(_p(self.W1), _p(x), m, self.F, _p(r1), _stream()))
        elif tl:    # a1^T = W1 x^T: the orientation hipBLASLt runs this product fastest in; mid_fwd adds the bias
            self._mm(self.W1, x.t(), r1T)
        else:
            torch.addmm(self.b1, x, self.W1.t(), out=r1)
        if self._cold:
            self._evict()
        if pl:      # mid_fwd adds the eight partial sums; its spare workgroups assemble the first half of the next batch AND its planes
            st = next_from
            chk(_L.idl_mid_fwd_gather_planes(_p(pb["part"][xi]), _p(self.b1), 1, _p(self.W2), _p(self.b2), _p(self.W3), _p(self.b3),
                                             m, C, tr, self.seed, _p(self.ctl), _p(bf.f), _p(bf.inv), _p(bf.r2), _p(bf.z),
                                             _p(st.feats), st.n, st.f, st.n * st.f, _p(self._perm), _p(self.ctl[1:]), m // 2, st.n_pairs, m // 2,
                                             _p(st.mean), _p(st.scale), _p(st.inv_scale), None if plw else _p(bf.xs[1 - xi]), _p(pb["xh"][1 - xi]),
                                             _p(pb["xl"][1 - xi]), _p(self._w1_planes[2]), g1, g2, 8, _stream()))
        elif early and self._early_split:         # ... and the first half of the next batch's tiles in its spare workgroups
            st = next_from
            self._k(_L.idl_mid_fwd_gather, _p(r1), _p(self.b1) if tl else None,
                    1 if tl else 0, _p(self.W2), _p(self.b2), _p(self.W3), _p(self.b3),
                    m, C, tr, self.seed, _p(self.ctl),
                    _p(bf.f), _p(bf.inv), _p(bf.r2), _p(bf.z),
                    _p(st.feats), st.n, st.f, st.n * st.f, _p(self._perm), _p(self.ctl[1:]), m // 2, st.n_pairs, m // 2,
                    _p(st.mean), _p(st.scale), _p(st.inv_scale), _p(bf.xs[1 - xi]), g1, g2, 8, _stream())
        elif plf:   # (the bias of Linear(F,512) is added here; the whole next batch assembled as planes only)
            st = next_from
            # (round 6: the fused middle-backward of this mode carries the other half of the next batch's assembly, as at n_clusters <= 48)
            chk(_L.idl_mid_fwd_gather_planes(_p(r1), _p(self.b1), 0, _p(self.W2), _p(self.b2), _p(self.W3), _p(self.b3),
                                             m, C, tr, self.seed, _p(self.ctl), _p(bf.f), _p(bf.inv), _p(bf.r2), _p(bf.z),
                                             _p(st.feats), st.n, st.f, st.n * st.f, _p(self._perm), _p(self.ctl[1:]), m // 2, st.n_pairs, m // 2,
                                             _p(st.mean), _p(st.scale), _p(st.inv_scale), None, _p(pb["xh"][1 - xi]), _p(pb["xl"][1 - xi]),
                                             _p(self._w1_planes[2]), 0, self._gsplit, 8, _stream()))
        elif early_f:
            st = next_from
            chk(_L.idl_mid_fwd_gather(_p(r1), None, 0, _p(self.W2), _p(self.b2), _p(self.W3), _p(self.b3),
                                      m, C, tr, self.seed, _p(self.ctl),
                                      _p(bf.f), _p(bf.inv), _p(bf.r2), _p(bf.z),
                                      _p(st.feats), st.n, st.f, st.n * st.f, _p(self._perm), _p(self.ctl[1:]), m // 2, st.n_pairs, m // 2,
                                      _p(st.mean), _p(st.scale), _p(st.inv_scale), _p(bf.xs[1 - xi]), 0, 8, 8, _stream()))
        elif self._mid_fused and m % 16 == 0:   # ReLU/Dropout + Linear(512,64) + head in one MFMA kernel
            chk(_L.idl_mid_fwd(_p(r1), _p(self.W2), _p(self.b2), _p(self.W3), _p(self.b3), m, C, tr, self.seed, _p(self.ctl),
                               _p(bf.f), _p(bf.inv), _p(bf.r2), _p(bf.z), _stream()))
        else:
            chk(_L.idl_relu_dropout_fwd(_p(r1), r1.numel(), tr, self.seed, _p(self.ctl), 1, _stream()))
            torch.addmm(self.b2, r1, self.W2.t(), out=bf.lat)
            chk(_L.idl_head_fwd(_p(bf.lat), _p(self.W3), _p(self.b3), m, C, tr, self.seed, _p(self.ctl),
                                _p(bf.f), _p(bf.inv), _p(bf.r2), _p(bf.z), _stream()))
        # ---- the two losses are independent: with the fused InfoNCE kernels the IIC core rides along as one extra workgroup
        if bf.nce_fused and C <= 48 and not self._overlap and self._joint_inlaunch:
            # the IIC workgroup of InfoNCE pass 1 forms the joint z1^T z2 itself (MFMA tiles) before the core
            self._k(_L.idl_nce_fused_iic_z, _p(bf.f), m, TEMPERATURE, _p(bf.lse), _p(bf.loss_rows), _p(bf.G), _p(bf.nce_ws), _p(bf.z),
                    _p(bf.P0), C, self.lamb, EPS, self.weight, _p(bf.iic_scratch), _p(self.out), _stream())
        elif bf.nce_fused and C <= 48 and not self._overlap:
            torch.mm(bf.z[:m // 2].t(), bf.z[m // 2:], out=bf.P0)            # IIC joint, one [C,B]x[B,C] GEMM
            chk(_L.idl_nce_fused_iic(_p(bf.f), m, TEMPERATURE, _p(bf.lse), _p(bf.loss_rows), _p(bf.G), _p(bf.nce_ws),
                                     _p(bf.P0), C, self.lamb, EPS, self.weight, _p(bf.iic_scratch), _p(self.out), _stream()))
        elif plf and bf.nce_fused and 48 < C <= 200:
            # n_clusters > 48, round 6: the joint's 16 x 16 tiles ride in InfoNCE pass 1 as spare workgroups; then the IIC core's first launch and z dP0
            # with the gradient's shift in its epilogue (no shift launch, no library GEMM)
            chk(_L.idl_nce_fused_joint(_p(bf.f), m, TEMPERATURE, _p(bf.lse), _p(bf.loss_rows), _p(bf.G), _p(bf.nce_ws), _p(bf.z), _p(bf.P0), C, _stream()))
            chk(_L.idl_iic_core_dz(_p(bf.P0), C, self.lamb, EPS, self.weight, _p(bf.iic_scratch), _p(self.out), _p(bf.z), m, _p(bf.dzs), _stream()))
        else:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                chk(_L.idl_iic_joint(_p(bf.z), m, C, _p(bf.P0), _stream()))       # (a library GEMM here: 118 us untuned at C = 200)
                if plf and 48 < C <= 200:    # the core's first launch, then z dP0 with the gradient's shift in its epilogue (no shift launch, no library GEMM)
                    chk(_L.idl_iic_core_dz(_p(bf.P0), C, self.lamb, EPS, self.weight, _p(bf.iic_scratch), _p(self.out), _p(bf.z), m, _p(bf.dzs), _stream()))
                else:
                    chk(_L.idl_iic_core(_p(bf.P0), C, self.lamb, EPS, self.weight, _p(bf.iic_scratch), _p(self.out), _stream()))
            if bf.nce_fused:       # S = f f^T, lse, E + E^T and (E + E^T) f in two MFMA kernels, S never written
                chk(_L.idl_nce_fused(_p(bf.f), m, TEMPERATURE, _p(bf.lse), _p(bf.loss_rows), _p(bf.G), _p(bf.nce_ws), _stream()))
            else:
                torch.mm(bf.f, bf.f.t(), out=bf.S)
                chk(_L.idl_nce_rows(_p(bf.S), m, TEMPERATURE, _p(bf.lse), _p(bf.loss_rows), _stream()))
                torch.mm(bf.S, bf.f, out=bf.G[0])                            # (E + E^T) f
        main.wait_stream(side)
        nce_coef = (1.0 - self.weight) / (m * TEMPERATURE)
        gW1, gb1, gW2, gb2, gW3, gb3 = self.grads
        adv_ctl = _p(self.ctl) if (next_from is not None and not early and not early_f) else None
        adv = batch_advance if (next_from is not None and not early and not early_f) else 0
        if pl:
            st = next_from
            chk(_L.idl_mid_bwd_gather_planes(_p(bf.z), _p(bf.r2), _p(bf.f), _p(bf.inv), _p(bf.G), bf.G.shape[0], _p(bf.P0), _p(self.W3), _p(self.W2),
                                             _p(r1), m, C, tr, nce_coef, _p(bf.dlogits), _p(bf.dlat), _p(bf.dr1), _p(gb1), _p(gb2), _p(gb3),
                                             _p(gW3) if self._dw3_partial else None,
                                             _p(st.feats), st.n, st.f, st.n * st.f, _p(self._perm), _p(self.ctl[1:]), m // 2, st.n_pairs, m // 2,
                                             _p(st.mean), _p(st.scale), _p(st.inv_scale), None if plw else _p(bf.xs[1 - xi]), _p(pb["xh"][1 - xi]),
                                             _p(pb["xl"][1 - xi]), _p(self._w1_planes[2]), g2, 8, 8, 1, *self._dr1_planes_args(pb, dpl), _stream()))
            pb["valid"][1 - xi] = True
            pb["x32"][1 - xi] = not plw
            if not self._dw3_partial:
                torch.mm(bf.dlogits.t(), bf.r2, out=gW3)
        elif early:
            st = next_from
            self._k(_L.idl_mid_bwd_gather, _p(bf.z), _p(bf.r2), _p(bf.f), _p(bf.inv), _p(bf.G), bf.G.shape[0], _p(bf.P0), _p(self.W3), _p(self.W2),
                    _p(r1), m, C, tr, nce_coef, _p(bf.dlogits), _p(bf.dlat), _p(bf.dr1), _p(gb1), _p(gb2), _p(gb3),
                    _p(gW3) if self._dw3_partial else None,
                    _p(st.feats), st.n, st.f, st.n * st.f, _p(self._perm), _p(self.ctl[1:]), m // 2, st.n_pairs, m // 2,
                    _p(st.mean), _p(st.scale), _p(st.inv_scale), _p(bf.xs[1 - xi]), g2 if self._early_split else 0,
                    8, 8, 1 if tl else 0, _stream())
            if not self._dw3_partial:
                torch.mm(bf.dlogits.t(), bf.r2, out=gW3)
            if not self._dw2_inlaunch:
                torch.mm(bf.dlat.t(), r1, out=gW2)
        elif plf:
            # n_clusters > 48 (the CLI's default mode: 200 output units), round 6: z dP0 for all rows as one GEMM, then ONE launch for the rest of the
            # middle backward -- softmax / Linear(64,C) / normalise backward per row with W3 in LDS, dr1 = dlat W2 on MFMA tiles masked by the layer-1
            # ReLU / Dropout and written as two fp16 planes for the dW1 tiles, every bias gradient as stacked partial sums, the second half of the
            # next batch assembled by spare workgroups -- in place of idl_head_bwd_dz + a library GEMM + idl_bias_grads (11.8 + 9.0 + 4.9 us)
            st = next_from
            if not 48 < C <= 200:
                torch.mm(bf.z, bf.P0, out=bf.dzs)
            dplf = True                        # (m % 128 == 0 in this form)
            chk(_L.idl_mid_bwd_gather_planes(_p(bf.z), _p(bf.r2), _p(bf.f), _p(bf.inv), _p(bf.G), bf.G.shape[0], _p(bf.P0), _p(self.W3), _p(self.W2),
                                             _p(r1), m, C, tr, nce_coef, _p(bf.dlogits), _p(bf.dlat), _p(bf.dr1), _p(gb1), _p(gb2), _p(gb3), None,
                                             _p(st.feats), st.n, st.f, st.n * st.f, _p(self._perm), _p(self.ctl[1:]), m // 2, st.n_pairs, m // 2,
                                             _p(st.mean), _p(st.scale), _p(st.inv_scale), None, _p(pb["xh"][1 - xi]), _p(pb["xl"][1 - xi]),
                                             _p(self._w1_planes[2]), self._gsplit, 8, 8, 0, *self._dr1_planes_args(pb, dplf, bf.dzs), _stream()))
            pb["valid"][1 - xi] = True
            pb["x32"][1 - xi] = False
            chk(_L.idl_at_b(_p(bf.dlogits), C, _p(bf.r2), self.H2, m, C, self.H2, _p(gW3), self.H2, _stream()))      # dW3 = dlogits^T r2 (the library's kernel: 9 us)
        elif self._mid_fused and C <= 48:     # (at n_clusters = 200 the per-row C x C products want all 256 CUs: separate kernels)
            # ---- head backward + dr1 = dlat W2 (MFMA) + ReLU/Dropout backward + every bias gradient (+ dW3) in one launch
            chk(_L.idl_mid_bwd(_p(bf.z), _p(bf.r2), _p(bf.f), _p(bf.inv), _p(bf.G), bf.G.shape[0], _p(bf.P0), _p(self.W3), _p(self.W2),
                               _p(r1), m, C, tr, nce_coef, _p(bf.dlogits), _p(bf.dlat), _p(bf.dr1), _p(gb1), _p(gb2), _p(gb3),
                               _p(gW3) if self._dw3_partial else None, adv_ctl, adv, _stream()))
            if not self._dw3_partial:
                torch.mm(bf.dlogits.t(), bf.r2, out=gW3)
            if not (self._dw2_inlaunch and next_from is not None):
                torch.mm(bf.dlat.t(), r1, out=gW2)
        else:
            if C > 64:      # fine-grained mode: z_partner dP0 for all rows as one GEMM instead of 40 000 FMAs per row inside the kernel
                torch.mm(bf.z, bf.P0, out=bf.dzs)
                chk(_L.idl_head_bwd_dz(_p(bf.z), _p(bf.r2), _p(bf.f), _p(bf.inv), _p(bf.G), bf.G.shape[0], _p(bf.dzs), _p(self.W3), m, C, tr,
                                       nce_coef, _p(bf.dlogits), _p(bf.dlat), _stream()))
            else:
                chk(_L.idl_head_bwd(_p(bf.z), _p(bf.r2), _p(bf.f), _p(bf.inv), _p(bf.G), bf.G.shape[0], _p(bf.P0), _p(self.W3), m, C, tr,
                                    nce_coef, _p(bf.dlogits), _p(bf.dlat), _stream()))
            # ---- parameter gradients (one launch for the three bias gradients + the ReLU/Dropout backward of layer 1)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                if not self._dw3_partial:
                    torch.mm(bf.dlogits.t(), bf.r2, out=gW3)
                if not (self._dw2_inlaunch and next_from is not None):
                    torch.mm(bf.dlat.t(), r1, out=gW2)
            torch.mm(bf.dlat, self.W2, out=bf.dr1)
            chk(_L.idl_bias_grads(_p(bf.dr1), _p(r1), self.H1, _p(gb1), _p(bf.dlat), self.H2, _p(gb2), _p(bf.dlogits), C, _p(gb3),
                                  m, tr, adv_ctl, adv, _p(bf.r2) if self._dw3_partial else None,
                                  _p(gW3) if self._dw3_partial else None, _stream()))
        if tm:      # the dW1 tiles end the step; everything else of the optimizer rides in the next step's layer-1 launch
            if plw:     # dW1 from the batch's planes on the fp16 matrix cores; the epilogue writes the updated W1 and its planes
                wh, wl, flag = self._w1_planes
                # (a tile per CU, and a tile for each of the tail's blocks: 128 dW2 tiles + at most 16 blocks for the small tensors)
                if self._planes_tail_wgrad and self._planes_reduce_launch and 144 <= (self.H1 // 64) * (self.F // 128) <= self._cus:
                    # ... and THIS step's optimizer tail is run by the tiles' loader waves under the tiles' epilogue: nothing is pending
                    tail = (len(self.params), self._pp, self._gp, self._parts, self._vp, self._sz, _p(self.hyper),
                            _p(self.ctl), _p(bf.loss_rows), m, 1.0 - self.weight, self.weight, _p(self.out))
                    wg = (2, _p(bf.dlat), _p(r1), 1, m, self.H2, self.H1, _p(self.grads[2]), m // 2, _stream())
                    if self._cold:
                        self._evict()
                    chk(_L.idl_wgrad_xplanes_rms(*self._dy_planes_args(pb), _p(pb["xh"][xi]), _p(pb["xl"][xi]), self.F, m, self.H1, self.F,
                                                 _p(gW1) if self._keep_w1_grad else None, _p(self.W1), _p(self.square_avg[0]),
                                                 _p(wh), _p(wl), _p(flag), *tail, 0, *wg))
                    self._pending = None
                    return
                chk(_L.idl_wgrad_rmsprop_xplanes(*self._dy_planes_args(pb), _p(pb["xh"][xi]), _p(pb["xl"][xi]), self.F, m, self.H1, self.F,
                                                 _p(gW1) if self._keep_w1_grad else None, _p(self.W1), _p(self.square_avg[0]), _p(self.hyper),
                                                 _p(wh), _p(wl), _p(flag), _stream()))
            elif pl:    # ... and write the updated W1's planes for the next layer-1 product
                wh, wl, flag = self._w1_planes
                chk(_L.idl_wgrad_rmsprop_planes(_p(bf.dr1), _p(x), m, self.H1, self.F, _p(gW1) if self._keep_w1_grad else None, _p(self.W1),
                                                _p(self.square_avg[0]), _p(self.hyper), _p(wh), _p(wl), _p(flag), _stream()))
            else:
                chk(_L.idl_wgrad_rmsprop(_p(bf.dr1), _p(x), m, self.H1, self.F, _p(gW1) if self._keep_w1_grad else None, _p(self.W1),
                                         _p(self.square_avg[0]), _p(self.hyper), _stream()))
            self._pending = (bf, xi, r1)
            if not defer_tail:
                self.flush_tail()
            return
        if plf:     # dW1 from the batch's planes with RMSprop and W1's planes in the epilogue; the rest of the optimizer on the loader waves
            main.wait_stream(side)
            wh, wl, flag = self._w1_planes
            tail = (len(self.params), self._pp, self._gp, self._parts, self._vp, self._sz, _p(self.hyper),
                    _p(self.ctl), _p(bf.loss_rows), m, 1.0 - self.weight, self.weight, _p(self.out))
            wg = (2, _p(bf.dlat), _p(r1), 0, m, self.H2, self.H1, _p(gW2), m // 2, _stream())
            if self._cold:
                self._evict()
            chk(_L.idl_wgrad_xplanes_rms(*self._dy_planes_args(pb), _p(pb["xh"][xi]), _p(pb["xl"][xi]), self.F, m,
                                         self.H1, self.F, _p(gW1) if self._keep_w1_grad else None, _p(self.W1), _p(self.square_avg[0]),
                                         _p(wh), _p(wl), _p(flag), *tail, 0, *wg))
            return
        w1_fusable = self._wgrad_fused and bool(_L.idl_wgrad_supported(m, self.H1, self.F))
        # the tiles ride at the head of the optimizer launch where that launch has the form below; else as a launch of their own
        w1_head = w1_fusable and (early or early_f) and self._dw2_inlaunch and not self._wgrad_own_launch
        w1_done = w1_fusable and not w1_head and self._rec is None     # (a recorded step takes the tiles only inside its optimizer launch)
        gw1_out = _p(gW1) if self._keep_w1_grad else None
        if w1_done:
            chk(_L.idl_wgrad_rmsprop(_p(bf.dr1), _p(x), m, self.H1, self.F, gw1_out, _p(self.W1), _p(self.square_avg[0]), _p(self.hyper),
                                     _stream()))
        elif not w1_head:
            self._mm(bf.dr1.t(), x, gW1)
        sz = self._sz_no_w1 if w1_done else self._sz
        main.wait_stream(side)
        # ---- RMSprop (and advance the device-side step counter / batch offset)
        if w1_head:
            self._k(_L.idl_wgrad_rmsprop_step, len(self.params), self._pp, self._gp, self._parts, self._vp, sz, _p(self.hyper),
                    _p(self.ctl), _p(bf.loss_rows), m, 1.0 - self.weight, self.weight, _p(self.out),
                    0, _p(bf.dr1), _p(x), m, self.H1, self.F, gw1_out,
                    2, _p(bf.dlat), _p(r1), 1 if tl else 0, m, self.H2, self.H1, _p(gW2), m // 2, _stream())
        elif (early or early_f) and self._dw2_inlaunch:      # no batch assembly here; the offset moves on at the end of the step
            self._k(_L.idl_rmsprop_step_gather_wgrad, len(self.params), self._pp, self._gp, self._parts, self._vp, sz, _p(self.hyper),
                    _p(self.ctl), _p(bf.loss_rows), m, 1.0 - self.weight, self.weight, _p(self.out),
                    None, 0, 0, 0, None, 0, 0, None, None, None, None,
                    2, _p(bf.dlat), _p(r1), 1 if tl else 0, m, self.H2, self.H1, _p(gW2), m // 2, _stream())
        elif early:
            chk(_L.idl_rmsprop_step(len(self.params), self._pp, self._gp, self._parts, self._vp, sz, _p(self.hyper),
                                    _p(self.ctl), m // 2, _p(bf.loss_rows), m, 1.0 - self.weight, self.weight, _p(self.out), _stream()))
        elif next_from is not None and self._dw2_inlaunch:
            st = next_from
            chk(_L.idl_rmsprop_step_gather_wgrad(len(self.params), self._pp, self._gp, self._parts, self._vp, sz, _p(self.hyper),
                                                 _p(self.ctl), _p(bf.loss_rows), m, 1.0 - self.weight, self.weight, _p(self.out),
                                                 _p(st.feats), st.n, st.f, st.n * st.f, _p(self._perm), st.n_pairs, m // 2,
                                                 _p(st.mean), _p(st.scale), _p(st.inv_scale), _p(bf.x),
                                                 2, _p(bf.dlat), _p(r1), 0, m, self.H2, self.H1, _p(gW2), 0, _stream()))
        elif next_from is not None:
            st = next_from
            chk(_L.idl_rmsprop_step_gather(len(self.params), self._pp, self._gp, self._parts, self._vp, sz, _p(self.hyper),
                                           _p(self.ctl), _p(bf.loss_rows), m, 1.0 - self.weight, self.weight, _p(self.out),
                                           _p(st.feats), st.n, st.f, st.n * st.f, _p(self._perm), st.n_pairs, m // 2,
                                           _p(st.mean), _p(st.scale), _p(st.inv_scale), _p(bf.x), _stream()))
        else:
            chk(_L.idl_rmsprop_step(len(self.params), self._pp, self._gp, self._parts, self._vp, sz, _p(self.hyper),
                                    _p(self.ctl), batch_advance, _p(bf.loss_rows), m, 1.0 - self.weight, self.weight, _p(self.out),
                                    _stream()))

    def _record_planes_step(self, bf, tr, st, xi):
        """The two-plane step (step_on_batch's default form of a lone voter: layer-1 tiles, the sum of their partials, mid_fwd, InfoNCE + IIC,
        mid_bwd, the dW1 tiles with the tail on their loader waves) as six RECORDED launches: BatchedLinearTrainer runs each once for all the
        voters of a rank.  Nothing is launched here but what allocates this voter's plane buffers."""
        m, C = bf.m, self.C
        pb = _planes_of(bf, self.F)
        if self._w1_planes is None:
            self._w1_planes = (torch.empty(self.W1.shape, dtype=torch.int16, device=self.dev), torch.empty(self.W1.shape, dtype=torch.int16, device=self.dev),
                               torch.zeros(1, dtype=torch.int32, device=self.dev))
        wh, wl, flag = self._w1_planes
        part = pb["part"][xi]
        r1 = part[0]
        gW1, gb1, gW2, gb2, gW3, gb3 = self.grads
        nce_coef = (1.0 - self.weight) / (m * TEMPERATURE)
        g2 = self._gsplit
        dpl = True
        self._k(_L.idl_l1_planes, _p(wh), _p(wl), self.F, _p(pb["xh"][xi]), _p(pb["xl"][xi]), self.F, m, self.H1, self.F, _p(part), _stream())
        self._k(_L.idl_reduce_parts_rms, _p(part), self.H1 * m, _p(self.ctl), _p(self._ctl_snap), 0, None, None, None, None, None, None, None, None, 0, 0.0, 0.0,
                None, 0, -1, None, None, 0, 0, 0, 0, None, 0, _stream())
        self._k(_L.idl_mid_fwd_gather_planes, _p(part), _p(self.b1), 1, _p(self.W2), _p(self.b2), _p(self.W3), _p(self.b3),
                m, C, tr, self.seed, _p(self.ctl), _p(bf.f), _p(bf.inv), _p(bf.r2), _p(bf.z),
                _p(st.feats), st.n, st.f, st.n * st.f, _p(self._perm), _p(self.ctl[1:]), m // 2, st.n_pairs, m // 2,
                _p(st.mean), _p(st.scale), _p(st.inv_scale), None, _p(pb["xh"][1 - xi]), _p(pb["xl"][1 - xi]), _p(flag), 0, g2, 8, _stream())
        self._k(_L.idl_nce_fused_iic_z, _p(bf.f), m, TEMPERATURE, _p(bf.lse), _p(bf.loss_rows), _p(bf.G), _p(bf.nce_ws), _p(bf.z),
                _p(bf.P0), C, self.lamb, EPS, self.weight, _p(bf.iic_scratch), _p(self.out), _stream())
        self._k(_L.idl_mid_bwd_gather_planes, _p(bf.z), _p(bf.r2), _p(bf.f), _p(bf.inv), _p(bf.G), bf.G.shape[0], _p(bf.P0), _p(self.W3), _p(self.W2),
                _p(r1), m, C, tr, nce_coef, _p(bf.dlogits), _p(bf.dlat), _p(bf.dr1), _p(gb1), _p(gb2), _p(gb3), _p(gW3),
                _p(st.feats), st.n, st.f, st.n * st.f, _p(self._perm), _p(self.ctl[1:]), m // 2, st.n_pairs, m // 2,
                _p(st.mean), _p(st.scale), _p(st.inv_scale), None, _p(pb["xh"][1 - xi]), _p(pb["xl"][1 - xi]), _p(flag), g2, 8, 8, 1,
                *self._dr1_planes_args(pb, dpl), _stream())
        tail = (len(self.params), self._pp, self._gp, self._parts, self._vp, self._sz, _p(self.hyper),
                _p(self.ctl), _p(bf.loss_rows), m, 1.0 - self.weight, self.weight, _p(self.out))
        wg = (2, _p(bf.dlat), _p(r1), 1, m, self.H2, self.H1, _p(gW2), m // 2, _stream())
        self._k(_L.idl_wgrad_xplanes_rms, *self._dy_planes_args(pb), _p(pb["xh"][xi]), _p(pb["xl"][xi]), self.F, m, self.H1, self.F, None, _p(self.W1),
                _p(self.square_avg[0]), _p(wh), _p(wl), _p(flag), *tail, 0, *wg)

    def _dr1_planes_args(self, pb, on, dzs=None):
        """idl_mid_bwd_gather_planes' last arguments: dr1's planes and the words of their scale (or none of them: dr1 in fp32), and z dP0 as an
        input (the step of n_clusters > 48)."""
        pb["dr1_as_planes"] = bool(on)
        return ((_p(pb["dh"]), _p(pb["dl"]), _p(self._dr1_scale)) if on else (None, None, None)) + (_p(dzs),)

    def dr1_of(self, bf):
        """dr1 of the last step on these buffers as an fp32 tensor (tests): the step's own tensor, or -- where mid_bwd wrote it as planes only --
        the planes put back together."""
        pb = getattr(bf, "_planes", None)
        if pb is None or not pb.get("dr1_as_planes", False):
            return bf.dr1
        return ((pb["dh"].view(torch.float16).double() + pb["dl"].view(torch.float16).double()) * 2.0 ** -int(self._dr1_scale[0].item())).float()

    def _dy_planes_args(self, pb):
        return (_p(pb["dh"]), _p(pb["dl"]), _p(self._dr1_scale))

    def _tail_launch(self, bf, xi, r1, l1=None, red=None):
        """The optimizer's tail of the step that ran on (bf, xi) with the activations r1: dW2 tiles + RMSprop on every tensor but W1 + step
        loss + step counter -- behind the layer-1 tiles of the next step (l1 = (x, m, r1T) of THAT step), beside the workgroups that add up the next
        step's layer-1 partial sums (red), or as a launch of its own."""
        m = bf.m
        tail = (len(self.params), self._pp, self._gp, self._parts, self._vp, self._sz, _p(self.hyper),
                _p(self.ctl), _p(bf.loss_rows), m, 1.0 - self.weight, self.weight, _p(self.out))
        wg = (2, _p(bf.dlat), _p(r1), 1, m, self.H2, self.H1, _p(self.grads[2]), m // 2, _stream())
        if red is not None:       # beside the workgroups that add up the next step's layer-1 partial sums (red = (part, elements of a slab))
            part, slab = red
            _lib.check(_L.idl_reduce_parts_rms(_p(part), slab, None, None, *tail, 0, *wg))
        elif l1 is not None:
            x, m1, r1T = l1
            _lib.check(_L.idl_l1_fwd_rms(_p(self.W1), _p(x), m1, self.F, _p(r1T), *tail, 0, *wg))
        else:       # (sizes without W1: the tiles' own launch updated it)
            tail = tail[:5] + (self._sz_no_w1,) + tail[6:]
            _lib.check(_L.idl_rmsprop_step_gather_wgrad(*tail, None, 0, 0, 0, None, 0, 0, None, None, None, None, *wg))
        self._pending = None

    def flush_tail(self):
        """Run the pending optimizer tail now (end of an epoch, before a step of another form, before anybody looks at the small
        tensors); a no-op when nothing is pending."""
        if self._pending is not None:
            bf, xi, r1 = self._pending
            self._tail_launch(bf, xi, r1)

    def _prepare_planes(self, bf, pb, xi):
        """Before a step of the two-plane form: W1's planes (made once; afterwards the dW1 tiles' epilogue keeps them) and the planes of the
        batch in bf.xs[xi] unless the step that assembled it wrote them."""
        if self._w1_planes is None:
            self._w1_planes = (torch.empty(self.W1.shape, dtype=torch.int16, device=self.dev), torch.empty(self.W1.shape, dtype=torch.int16, device=self.dev),
                               torch.zeros(1, dtype=torch.int32, device=self.dev))
        if not self._w1_planes_fresh:
            wh, wl, flag = self._w1_planes
            _lib.check(_L.idl_split_planes(_p(self.W1), self.W1.numel(), int(_L.idl_planes_exponent(1)), _p(wh), _p(wl), _p(flag), _stream()))
            self._w1_planes_fresh = True
        if not pb["valid"][xi]:
            x = bf.xs[xi]
            _lib.check(_L.idl_split_planes(_p(x), x.numel(), int(_L.idl_planes_exponent(0)), _p(pb["xh"][xi]), _p(pb["xl"][xi]), _p(self._w1_planes[2]), _stream()))
            pb["valid"][xi] = True

    def _evict(self):
        if self._cold_buf is None:
            self._cold_buf = torch.empty(128 << 20, dtype=torch.float32, device=self.dev)
        self._cold_buf.fill_(1.0)

    def planes_overflowed(self):
        """Whether an entry of W1 (|w| >= 15.8) or of a standardised batch (|x| > 8 125: a mimic's feature thousands of the originals' standard
        deviations out) ever left its planes' range (waits for the device).  Such an entry was clamped in the two big products: the run
        should be repeated with IDELUCS_PLANES=0."""
        return self._w1_planes is not None and bool(self._w1_planes[2].item())

    def drop_planes(self):
        """Leave the two-plane step form for good (an operand left the planes' range): the fp32 tiles from the next step on, the flag cleared, the
        captured graphs (they hold the plane launches) dropped.  The caller restarts its voter: what was trained on clamped operands is not kept."""
        self.flush_tail()
        self._planes = False
        self._planes_lockstep = False
        self._graphs.clear()
        self._w1_planes_fresh = False
        if self._w1_planes is not None:
            self._w1_planes[2].zero_()
        for bf in self._bufs.values():
            if getattr(bf, "_planes", None) is not None:
                bf._planes["valid"] = [False, False]
                bf._planes["x32"] = [True, True]
                bf._planes["dr1_as_planes"] = False

    def _gather(self, store, bf):
        b = bf.m // 2
        _lib.check(_L.idl_gather_pairs_at(_p(store.feats), store.n, store.f, store.n * store.f, _p(self._perm), _p(self.ctl[1:]),
                                          b, _p(store.mean), _p(store.scale), _p(store.inv_scale), _p(bf.x), _stream()))
        if getattr(bf, "_planes", None) is not None:
            bf._planes["valid"][0] = False
            bf._planes["x32"][0] = True

    def _full_step(self, store, bf, train=True, pipelined=False, xi=0, defer_tail=False):
        """pipelined: bf.xs[xi] already holds this batch (assembled by the previous step, or by the prologue gather);
        this step assembles the next one (into bf.xs[1 - xi] when the mid-backward launch does it, else into bf.xs[xi]).
        defer_tail (run_epoch's steps): the step's optimizer tail may wait for the next step's first launch (flush_tail)."""
        if pipelined:
            self.step_on_batch(bf, train=train, batch_advance=bf.m // 2, next_from=store,
                               xi=xi if (self._early_gather or self._early_fwd) else 0, defer_tail=defer_tail)
        else:
            self._gather(store, bf)
            self.step_on_batch(bf, train=train, batch_advance=bf.m // 2)

    # ------------------------------------------------------------------ one epoch over the store
    @torch.no_grad()
    def run_epoch(self, store, batch_sz, use_graph=True, generator=None):
        """One pass over a fresh permutation of the N*n_mimics pairs (models.py:117-133).
        Returns the device scalar sum of the per-step losses and the number of batches."""
        n_pairs = store.n_pairs
        if self._perm is None or self._perm.numel() != n_pairs:
            self._perm = torch.empty(n_pairs, dtype=torch.int64, device=self.dev)
            self._graphs.clear()
        # (the permutation on a stream of its own beside the vectoriser was measured: the epoch 67.2-67.4 ms against 64.8-65.0 on the
        #  main stream, T_e2e 76.8-79.2 against 76.1-76.3 -- the step graphs wait for the other stream's event)
        torch.randperm(n_pairs, device=self.dev, generator=generator, out=self._perm)
        self._w1_planes_fresh = False           # (whoever set the weights since the last epoch -- a voter's initialisation -- did not write planes)
        self.ctl[1:2].zero_()
        self.out[1:2].zero_()
        n_full, rem = divmod(n_pairs, batch_sz)
        pipe = self._pipeline
        if n_full:
            bf = self.buffers(2 * batch_sz)
            if pipe:
                self._gather(store, bf)         # prologue: batch 0; every later batch is assembled by the previous step
                if (self._planes and self._tail_l1 and bool(_L.idl_l1_planes_supported(bf.m, self.H1, self.F))):
                    self._prepare_planes(bf, _planes_of(bf, self.F), 0)      # (a replayed graph starts from valid planes)
            # steps per graph replay: an even number when two x buffers alternate.  Between two replays the GPU idles ~9 us
            # (profiles/r02_f: kernel trace), so a replay carries several steps
            per = self._steps_per_graph if pipe else 1
            while per > 2 and n_full < 2 + 2 * per:      # short epochs: the capture itself runs 2 + per real steps
                per = max(2, per // 4 * 2)
            # every address the captured launches bake in is part of the key (a store refitted in place keeps its graph)
            key = (2 * batch_sz, store.feats.data_ptr(), store.mean.data_ptr(), store.scale.data_ptr(), store.inv_scale.data_ptr(),
                   self._perm.data_ptr(), store.n, store.f, store.n_views, pipe, self._early_gather, self._early_fwd, per)
            if use_graph and n_full >= 8:
                g = self._graphs.get(key)
                if g is None:
                    g = self._capture(store, bf, pipe, per)
                    self._graphs = {key: g}             # one store at a time: drop graphs of older stores
                    n_done = 2 + per                     # the warm-up + capture already ran real steps (an even number when per == 2)
                else:
                    n_done = 0
                for _ in range((n_full - n_done) // per):
                    g.replay()
                for i in range((n_full - n_done) % per):
                    self._full_step(store, bf, pipelined=pipe, xi=i % 2, defer_tail=True)
            else:
                for i in range(n_full):
                    self._full_step(store, bf, pipelined=pipe, xi=i % 2, defer_tail=True)
        self.flush_tail()                       # (the last eager step's tail; a replayed graph ends with its own)
        if rem:
            self._full_step(store, self.buffers(2 * rem))
        return self.out[1], n_full + (1 if rem else 0)

    @torch.no_grad()
    def _capture(self, store, bf, pipe, per=1):
        """Warm up on a side stream (2 real steps), then capture the next `per` real steps into a HIP graph.
        Every launch is a genuine optimizer step on the next batch, so nothing is wasted or repeated."""
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for i in range(2):
                self._full_step(store, bf, pipelined=pipe, xi=i % 2, defer_tail=True)
            self.flush_tail()
        torch.cuda.current_stream().wait_stream(s)
        # (tail-in-layer-1: a replay is self-contained -- its first step has nothing pending in front of it, its last step's tail is
        #  a launch of its own at the end of the graph: one more launch per `per` steps, and a replay never applies a tail twice)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(per):
                self._full_step(store, bf, pipelined=pipe, xi=i % 2, defer_tail=True)
            self.flush_tail()
        g.replay()          # capture does not execute: run the captured step(s) once
        self.n_captures = getattr(self, "n_captures", 0) + 1
        return g


class BatchedLinearTrainer:
    """Several voters of one ensemble trained in lockstep on one GPU (each on its own network, permutation, dropout stream and
    optimizer state -- the same independent runs as one after the other; only the rounding of the batched GEMMs may differ
    from the single-voter ones).

    The step of voter l is the launch sequence of FusedLinearTrainer.step_on_batch; here each of its launches is issued ONCE for
    all voters: the two big products as batched GEMMs over stacked operands ([L, 512, F] weights, [L, m, F] batches), the five
    kernels as recorded launches (idl_plan_*: every voter's launch is recorded through the ordinary launcher, the records
    live on the device, the kernels take the voter index from blockIdx.y / .z).  Needs the default launch sequence
    (n_clusters <= 48, batch 2 x 512 rows assembled by the middle launches)."""

    def __init__(self, nets, lr, weight, lamb, seed=0):
        self.L = L = len(nets)
        lin1 = [n.layers[0] for n in nets]
        self.dev = dev = lin1[0].weight.device
        H1, F = lin1[0].weight.shape
        f32 = dict(dtype=torch.float32, device=dev)
        # stacked GEMM operands; every voter's parameters / gradients / buffers are views of them
        self.W1s = torch.empty((L, H1, F), **f32)
        self.gW1s = torch.zeros((L, H1, F), **f32)
        for l, lin in enumerate(lin1):
            self.W1s[l].copy_(lin.weight.data)
            lin.weight.data = self.W1s[l]
        self._stacks = {}
        self._H1, self._F = H1, F
        self.trainers = []
        for l, net in enumerate(nets):
            self.trainers.append(FusedLinearTrainer(net, lr, weight, lamb, seed=seed, grad_w1=self.gW1s[l], shared_buffers=_SharedViews(self, l)))
        t0 = self.trainers[0]
        if not (t0._early_gather and t0._early_split and t0._transposed_l1 and t0._dw2_inlaunch and t0._mid_fused and t0._dw3_partial
                and t0._joint_inlaunch and t0._pipeline and not t0._nce_bwd_fused and not t0._overlap):
            raise ValueError("BatchedLinearTrainer needs the default launch sequence (n_clusters <= 48, no opt-in variants)")
        self._programs = {}
        self._graphs = {}
        self._w1_in_launch = False
        self._planes_step = False

    def drop_planes(self):
        """After the voters' trainers left the two-plane form (FusedLinearTrainer.drop_planes): the recorded programs and captured graphs hold its launches."""
        self._programs = {}
        self._graphs.clear()
        self._planes_step = False

    def stack(self, m):
        """Stacked GEMM operands of batch shape m: XS[2][L, m, F], R1 [L, m, 512] (its transposed image [L, 512, m] is the layer-1
        product's output), DR1 [L, m, 512]."""
        if m not in self._stacks:
            f32 = dict(dtype=torch.float32, device=self.dev)
            self._stacks[m] = dict(xs=[torch.empty((self.L, m, self._F), **f32), torch.empty((self.L, m, self._F), **f32)],
                                   r1=torch.empty((self.L, m, self._H1), **f32), dr1=torch.empty((self.L, m, self._H1), **f32))
        return self._stacks[m]

    # ------------------------------------------------------------------ recording
    def _program(self, store, m):
        """The recorded launches of one pipelined step for the two x-buffer parities: [(host records [L, B], device records)] x 4."""
        key = (m, store.feats.data_ptr(), store.mean.data_ptr(), store.scale.data_ptr(), store.inv_scale.data_ptr(), store.n, store.f,
               store.n_views) + tuple(t._perm.data_ptr() for t in self.trainers)
        prog = self._programs.get(key)
        if prog is None:
            prog = []
            for xi in (0, 1):
                recs = []
                for t in self.trainers:
                    t._rec = _Recorder()
                    try:
                        t.step_on_batch(t.buffers(m), train=True, batch_advance=m // 2, next_from=store, xi=xi)
                    finally:
                        rec, t._rec = t._rec, None
                    # the two-plane step: six recorded launches, no library GEMM; the fp32 form: 4 kernel launches + the two big products as batched
                    # GEMMs (dW1 on own tiles at the head of the optimizer launch is a recorded launch instead: one GEMM)
                    if (len(rec.plans), rec.mms) not in ((4, 2), (4, 1), (6, 0)):
                        raise RuntimeError("the recorded step is not the default launch sequence")
                    self._planes_step = len(rec.plans) == 6
                    self._w1_in_launch = rec.mms == 1
                    recs.append(rec.plans)
                ops = []
                for k in range(len(recs[0])):
                    host = torch.stack([recs[l][k] for l in range(self.L)]).contiguous()
                    ops.append((host, host.to(self.dev)))
                prog.append(ops)
            self._programs = {key: prog}
            self._graphs.clear()
        return prog

    def _step(self, prog, st, xi):
        """One optimizer step of every voter on the batches in XS[xi] (assembling the next ones into XS[1 - xi])."""
        L = self.L
        ops = prog[xi]
        if self._planes_step:
            for k in range(6):                                # l1, reduce, mid_fwd, InfoNCE passes, mid_bwd, dW1 + RMSprop + tail: every voter's, one launch each
                if self.trainers[0]._cold:
                    self.trainers[0]._evict()
                _lib.check(_L.idl_plan_launch(ctypes.c_void_p(ops[k][0].data_ptr()), _p(ops[k][1]), L, _stream()))
            return
        k0 = 0
        r1T = st['r1'].view(L, self._H1, -1)
        torch.bmm(self.W1s, st['xs'][xi].transpose(1, 2), out=r1T)                        # a1^T = W1 x^T per voter
        for k in (k0, k0 + 1, k0 + 2):                                                   # mid_fwd, InfoNCE passes, mid_bwd
            _lib.check(_L.idl_plan_launch(ctypes.c_void_p(ops[k][0].data_ptr()), _p(ops[k][1]), L, _stream()))
        if not self._w1_in_launch:
            torch.bmm(st['dr1'].transpose(1, 2), st['xs'][xi], out=self.gW1s)            # dW1 = dr1^T x per voter
        _lib.check(_L.idl_plan_launch(ctypes.c_void_p(ops[k0 + 3][0].data_ptr()), _p(ops[k0 + 3][1]), L, _stream()))

    # ------------------------------------------------------------------ one epoch of every voter
    @torch.no_grad()
    def run_epoch(self, store, batch_sz, generators, use_graph=True):
        """One pass of every voter over its own fresh permutation -> [(device scalar sum of step losses, n_batches)] per voter."""
        n_pairs = store.n_pairs
        for t, g in zip(self.trainers, generators):
            if t._perm is None or t._perm.numel() != n_pairs:
                t._perm = torch.empty(n_pairs, dtype=torch.int64, device=self.dev)
            torch.randperm(n_pairs, device=self.dev, generator=g, out=t._perm)
            t.ctl[1:2].zero_()
            t.out[1:2].zero_()
        n_full, rem = divmod(n_pairs, batch_sz)
        m = 2 * batch_sz
        if n_full and m % 32 == 0:
            st = self.stack(m)
            for t in self.trainers:
                t._gather(store, t.buffers(m))            # prologue: batch 0 of every voter
            prog = self._program(store, m)
            if self._planes_step:                         # W1's planes (the weights were set since the last epoch) and those of every voter's batch 0
                for t in self.trainers:
                    t._w1_planes_fresh = False
                    bf = t.buffers(m)
                    t._prepare_planes(bf, _planes_of(bf, t.F), 0)
            per = self.trainers[0]._steps_per_graph
            while per > 2 and n_full < 2 + 2 * per:
                per = max(2, per // 4 * 2)
            done = 0
            if use_graph and n_full >= 8:
                g = self._graphs.get((m, per))
                if g is None:
                    s = torch.cuda.Stream()
                    s.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(s):
                        for i in range(2):
                            self._step(prog, st, i % 2)
                    torch.cuda.current_stream().wait_stream(s)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        for i in range(per):
                            self._step(prog, st, i % 2)
                    g.replay()
                    self._graphs[(m, per)] = g
                    done = 2 + per
                for _ in range((n_full - done) // per):
                    g.replay()
                done += (n_full - done) // per * per
            for i in range(n_full - done):
                self._step(prog, st, i % 2)
        elif n_full:
            for t in self.trainers:
                for i in range(n_full):
                    t._full_step(store, t.buffers(m), pipelined=t._pipeline, xi=i % 2)
        if rem:                                           # the partial last batch: voter by voter, the single-voter kernels
            for t in self.trainers:
                t._full_step(store, t.buffers(2 * rem))
        return [(t.out[1], n_full + (1 if rem else 0)) for t in self.trainers]


class _SharedViews:
    """{m: views} handed to a voter's _Buffers: its slices of the batched trainer's stacked GEMM operands (full batches only)."""

    def __init__(self, owner, l):
        self._owner, self._l = owner, l

    def get(self, m, default=None):
        if m % 32 != 0 or m < 64:
            return default
        st = self._owner.stack(m)
        l = self._l
        return dict(xs0=st['xs'][0][l], xs1=st['xs'][1][l], r1=st['r1'][l], dr1=st['dr1'][l])
