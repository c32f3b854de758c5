"""Practical streaming ceiling of the box, for context next to the aggregation kernel's
roofline fraction: plain device copies and a 2-read/1-write add at the aggregation's sizes,
timed with HIP events.      python tools/stream_bench.py
"""
import json

import torch


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    out = {}
    for V, C in ((1_000_000, 256), (1_000_000, 512), (4_000_000, 256)):
        x = torch.randn(V, C, device="cuda")
        y = torch.empty_like(x)
        z = torch.randn(V, C, device="cuda")
        nbytes = x.numel() * 4
        ms = timed(lambda: y.copy_(x))
        out[f"copy_{V}x{C}_f32_GBps"] = 2 * nbytes / ms / 1e6
        ms = timed(lambda: torch.add(x, z, out=y))
        out[f"add_{V}x{C}_f32_GBps"] = 3 * nbytes / ms / 1e6
        ms = timed(lambda: x.sum())
        out[f"sum_{V}x{C}_f32_GBps"] = nbytes / ms / 1e6
        del x, y, z
    print(json.dumps(out))


if __name__ == "__main__":
    main()
