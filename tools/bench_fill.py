#!/usr/bin/env python3
"""Diagnostic: what the chip sustains for pure streaming stores / copies of the feature store's size (6.55 GB)."""
import torch
dev = torch.device("cuda")
x = torch.empty((4, 100000, 4096), dtype=torch.float32, device=dev)
y = torch.empty((2, 100000, 4096), dtype=torch.float32, device=dev)
def t(fn, reps=6):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); best = min(best, s.elapsed_time(e))
    return best
ms = t(lambda: x.fill_(1.0)); print(f"fill_ 6.55 GB: {ms:.3f} ms -> {x.numel() * 4 / ms / 1e6:.0f} GB/s")
ms = t(lambda: x.zero_()); print(f"zero_ 6.55 GB: {ms:.3f} ms -> {x.numel() * 4 / ms / 1e6:.0f} GB/s")
ms = t(lambda: y[1].copy_(y[0])); print(f"copy 1.64 GB -> 1.64 GB: {ms:.3f} ms -> {2 * y[0].numel() * 4 / ms / 1e6:.0f} GB/s (read + write)")
ms = t(lambda: torch.sum(x[0])); print(f"sum 1.64 GB: {ms:.3f} ms -> {x[0].numel() * 4 / ms / 1e6:.0f} GB/s read")
