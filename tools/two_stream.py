#!/usr/bin/env python3
"""Experiment: two half-batches stepped on two HIP streams vs one full batch on one stream."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.ab_step import make  # noqa: E402
n = 1 << 20
full, pool = make(n, "run64", torch.float64, "Navigator", 4)
def bench_full(steps=40):
    for k in range(3): full.step(pool[k % 4])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps): full.step(pool[k % 4])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e6
for parts in (2, 4, 8):
    envs = [make(n // parts, "run64", torch.float64, "Navigator", 4) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    def step_all(k):
        for (e, p), s in zip(envs, streams):
            with torch.cuda.stream(s):
                e.step(p[k % 4])
    for k in range(3): step_all(k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(40): step_all(k)
    torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 40 * 1e6
    print("%d streams x %d envs: %.1f us per full step" % (parts, n // parts, us))
    # staggered start: delay odd streams by half a kernel once
    torch.cuda.synchronize()
    for i, ((e, p), s) in enumerate(zip(envs, streams)):
        if i % 2:
            with torch.cuda.stream(s):
                torch.cuda._sleep(int(2.4e9 * 40e-6))
    t0 = time.perf_counter()
    for k in range(40): step_all(k)
    torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 40 * 1e6
    print("%d streams staggered: %.1f us per full step" % (parts, us))
    del envs
print("1 stream x %d envs: %.1f us" % (n, bench_full()))
