#!/usr/bin/env python3
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.ab_step import make  # noqa: E402
n = 1 << 20
env, pool = make(n, "run64", torch.float64, "Navigator", 4)
for pm in (0, 1):
    env.set_option("params_mode", pm)
    for stag in (-1, 0, 1, 2, 3, 4, 6):
        env.set_option("pipeline", 0 if stag < 0 else 1); env.set_option("pipe_stagger", max(stag, 0))
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
        print("params_mode=%d stagger=%2d  median %7.1f us  min %7.1f us" % (pm, stag, np.median(ts), np.min(ts)))
assert env.sync(raise_on_status=False) == 0
