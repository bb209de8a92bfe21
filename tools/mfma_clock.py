#!/usr/bin/env python3
"""Reads the shader clock the chip holds under the fp32 matrix stream of the training step, instead of inferring it
from a kernel's wall time:   clock = delta s_memtime (shader cycles) / delta s_memrealtime (100 MHz) x 100 MHz
(MI355X_MICROARCH.md, DVFS give-back item 6), stamped right before the first and right after the last v_mfma_f32_16x16x4_f32 of every
compute wave of the dW1 tiles (idl_debug_wgrad_clock; 2048 MFMAs x 32 cycles = 65 536 matrix-pipe cycles per wave at cfg2's shape),
after >= 2 s of back-to-back launches, median over waves.  Arms:
  A  the MFMA stream alone, operands loaded once -- random data
  B  the same on all-zero operands (what an 'uninitialised registers' probe measures: round 3's 31.3 us figure)
  C  the product's loop (operands streamed through the register ring) -- random data
  D  arm C launched right behind the hipBLASLt layer-1 GEMM (the clock in the library kernel's neighbourhood)
  E  inside the real training step: the stamped optimizer launch (IDELUCS_DEV=stamps=1), whole-workgroup clock of the dW1 tiles
Prints a table; `python tools/mfma_clock.py > profiles/r04_mfma_clock.txt`."""
import ctypes
import os
import subprocess
import sys
import threading
import time

os.environ["IDELUCS_DEV"] = ",".join(x for x in (os.environ.get("IDELUCS_DEV", ""), "stamps=1") if x)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from idelucs_amd import _lib, utils as U, models
from idelucs_amd.fused import _p, _stream, FusedLinearTrainer
from idelucs_amd.PytorchUtils import NetLinear

L = _lib.lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
m, H, F = 1024, 512, 4096
TILES = (H // 64) * (F // 128)
FLOP = 2.0 * m * H * F


def smi_clocks(stop, out):
    """rocm-smi --showclocks sampled while a loop runs (the sysfs view; the guide notes it reads up to ~10 % above the in-kernel clock)"""
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=10)
            for line in r.stdout.splitlines():
                if "sclk" in line:
                    out.append(line.strip())
        except Exception as err:
            out.append(f"rocm-smi failed: {err}")
            return
        time.sleep(0.5)


def arm(name, dy, x, variant, before=None, seconds=2.5):
    g = torch.empty(H, F, device=dev)
    st = torch.zeros(TILES * 16, dtype=torch.int64, device=dev)

    def launch():
        if before is not None:
            before()
        _lib.check(L.idl_debug_wgrad_clock(_p(dy), _p(x), m, H, F, _p(g), variant, _p(st), _stream()))
    launch(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        launch(); launch()
        with torch.cuda.graph(gr):
            for _ in range(50):
                launch()
        torch.cuda.synchronize()
        t_end = time.time() + seconds
        n = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        while time.time() < t_end:                  # >= 2 s of back-to-back launches before the stamps that count
            gr.replay(); n += 1
            if n % 20 == 0:
                torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            gr.replay()
        e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / (10 * 50 * (2 if before is not None else 1))
    a = st.cpu().numpy().astype(np.uint64).reshape(TILES * 4, 4)
    cyc = (a[:, 1] - a[:, 0]).astype(np.float64)
    tick = (a[:, 3] - a[:, 2]).astype(np.float64)
    ghz = cyc / np.maximum(tick, 1) * 0.1
    us_stream = np.median(tick) * 0.01
    print(f"{name:58s} clock median {np.median(ghz):.3f} GHz (p5 {np.percentile(ghz, 5):.3f}, p95 {np.percentile(ghz, 95):.3f})  "
          f"cycles/wave median {np.median(cyc):.0f} (65536 = the pipe's count)  stream {us_stream:.2f} us  "
          f"launch-to-launch {us:.2f} us{' (mean of the pair)' if before is not None else ''}  "
          f"-> {FLOP / (us_stream * 1e-6) / 1e12:.1f} TFLOP/s inside the stream")
    return np.median(ghz), np.median(cyc), us_stream, us


print(f"# MI355X fp32 matrix pipe: the clock under the step's dW1 = dr1^T x product ({m} x {H} x {F}), {torch.cuda.get_device_name(0)}")
print("# peak 157.3 TFLOP/s = 256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz; 65 536 pipe cycles per wave / clock = the product's floor")
dy = torch.randn(m, H, device=dev) * (torch.rand(m, H, device=dev) < 0.25)      # dr1 as the step has it: ReLU/Dropout-masked (3/4 zeros)
dyd = torch.randn(m, H, device=dev)
x = torch.randn(m, F, device=dev)
z_dy, z_x = torch.zeros(m, H, device=dev), torch.zeros(m, F, device=dev)
W1 = torch.randn(H, F, device=dev) * 0.02
a1t = torch.empty(H, m, device=dev)

samples, stop = [], threading.Event()
th = threading.Thread(target=smi_clocks, args=(stop, samples), daemon=True)
th.start()
res = {}
res["A"] = arm("A  MFMA stream alone, random operands (dense)", dyd, x, 2)
res["A'"] = arm("A' MFMA stream alone, dy 3/4 zeros as in the step", dy, x, 2)
res["B"] = arm("B  MFMA stream alone, all-zero operands", z_dy, z_x, 2)
res["C"] = arm("C  product loop (operands streamed), random, dy 3/4 zeros", dy, x, 0)
res["C'"] = arm("C' product loop, dense random dy", dyd, x, 0)
res["D"] = arm("D  product loop right behind the hipBLASLt layer-1 GEMM", dy, x, 0, before=lambda: torch.mm(W1, x.t(), out=a1t))
stop.set(); th.join(timeout=15)
print("# rocm-smi --showclocks while the arms ran (sclk lines, first / middle / last):")
for l in (samples[:2] + samples[len(samples) // 2: len(samples) // 2 + 2] + samples[-2:]):
    print("#   " + l)

# E: inside the real step
P, n, C, B = 4, 6000, 20, 512
g = torch.Generator(device=dev); g.manual_seed(1)
feats = torch.rand((P, n, F), device=dev, generator=g) * 2e-4 + 1e-4
mean, scale = U.col_stats(feats[0])
store = U.FeatureStore(None, None, feats, mean, scale, 6, False)
net = NetLinear(F, C).to(dev); net.apply(models.weights_init)
tr = FusedLinearTrainer(net, 1e-3, 0.25, 2.8, seed=3)
t_end = time.time() + 2.5
while time.time() < t_end:
    tr.run_epoch(store, B, use_graph=False)
torch.cuda.synchronize()
out = np.zeros((1024, 4), np.uint64)
_lib.check(L.idl_debug_stamps(out.ctypes.data_as(ctypes.c_void_p)))
tick = (out[:256, 1] - out[:256, 0]).astype(np.float64)
cyc = (out[:256, 3] >> np.uint64(8)).astype(np.float64)
ghz = cyc / np.maximum(tick, 1) * 0.1
print(f"{'E  inside the training step: dW1 tile workgroups, whole life':58s} clock median {np.median(ghz):.3f} GHz (p5 {np.percentile(ghz, 5):.3f}, "
      f"p95 {np.percentile(ghz, 95):.3f})  cycles/workgroup median {np.median(cyc):.0f}  life {np.median(tick) * 0.01:.2f} us")
ca = res["A'"][0]
print(f"# floor of the 512 x 4096 x 1024 product at the clock arm A' holds: 65536 / {ca:.3f} GHz = {65536 / ca / 1e3:.2f} us; at 2.4 GHz 27.31 us")
