#!/usr/bin/env python3
"""HBM write ceiling on this GPU: torch fill_ / zero_ / copy_ of a 3.2 GB buffer (the size of one 128-frame 4K slot)."""
import torch
n = 128 * 3840 * 2160 * 3
x = torch.empty(n, dtype=torch.uint8, device="cuda")
y = torch.empty(n, dtype=torch.uint8, device="cuda")
xi = x.view(torch.int32)
for name, fn, bytes_ in (("fill_ int32", lambda: xi.fill_(0x0d0d120d), n), ("zero_", lambda: x.zero_(), n),
                         ("copy_ (read+write)", lambda: y.copy_(x), 2 * n)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name:22s} {ms:.3f} ms  {bytes_ / ms / 1e6:.0f} GB/s")
