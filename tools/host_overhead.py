#!/usr/bin/env python3
"""Tuning aid: host-side cost of one BatchedT1DSimEnv.step call (tiny batch: the GPU work is negligible)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simglucose_amd.batch_env import BatchedT1DSimEnv
env = BatchedT1DSimEnv(patient=np.arange(64) % 30, sensor="Navigator", seed=1, extra_outputs=False)
env.reset()
a = torch.full((64,), 0.01, dtype=torch.float64, device=env.device)
for _ in range(200):
    env.step(a)
torch.cuda.synchronize()
for label, fn in (("env.step (tensor action)", lambda: env.step(a)),):
    t0 = time.perf_counter()
    for _ in range(5000):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-28s issue %.1f us/step, drained %.1f us/step" % (label, (t1 - t0) / 5000 * 1e6, (t2 - t0) / 5000 * 1e6))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(2000):
    env.step(a)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
