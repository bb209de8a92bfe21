#!/usr/bin/env python3
"""Host-to-device copy rate from pinned memory on this box (diagnostic for the streamed ingest):  python tools/h2d_bw.py"""
import time
import torch
dev = torch.device("cuda:0")
for mb in (16, 64, 256):
    h = torch.empty(mb << 20, dtype=torch.uint8, pin_memory=True)
    h.fill_(1)
    d = torch.empty(mb << 20, dtype=torch.uint8, device=dev)
    for _ in range(2):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"H2D {mb} MB pinned: {dt * 1e3:.2f} ms = {mb / 1024 / dt:.1f} GB/s")
    t0 = time.perf_counter()
    for _ in range(5):
        h.copy_(d, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"D2H {mb} MB pinned: {dt * 1e3:.2f} ms = {mb / 1024 / dt:.1f} GB/s")
