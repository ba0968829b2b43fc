#!/usr/bin/env python3
"""Tuning aid: the persistent single-minute kernel with pieces switched off and with several grid sizes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.ab_step import make  # noqa: E402
n = 1 << 20
env, pool = make(n, "mod30", torch.float64, "Navigator", 4)
base = env._flags0
env.set_option("integrator", 1)


def timeit():
    for k in range(3):
        env.step(pool[k % 4])
    ts = []
    for r in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for k in range(20):
            env.step(pool[k % 4])
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 20 * 1e3)
    return np.median(ts), np.min(ts)


for pm in (1, 0):
    env.set_option("params_mode", pm)
    for stg in (0, 1, 2, 3, 4, 6):
        env.set_option("pipe_stagger", stg)
        print("params_mode=%d stagger=%d       median %7.1f us  min %7.1f us" % ((pm, stg) + timeit()), flush=True)
    env.set_option("pipe_stagger", 0)
