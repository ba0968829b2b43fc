#!/usr/bin/env python3
"""Time t1d_step for several minutes-per-launch values (separates per-launch memory cost from per-minute compute)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.ab_step import make  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
for var in ("reg", "split_reg", "split_lds"):
    env, pool = make(n, "run64", torch.float64, "Navigator", 4)
    env.set_option("scalar_params", 0); env.set_option("params_mode", 1 if var.endswith("reg") else 0)
    env.set_option("integrator", 1 if var.startswith("split") else 0)
    for minutes in (1, 2, 4, 8, 16):
        for k in range(3):
            env.step(pool[k % 4], minutes=minutes)
        ts = []
        for r in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); s.record()
            for k in range(10):
                env.step(pool[k % 4], minutes=minutes)
            e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / 10 * 1e3)
        print("%s minutes=%2d  %8.1f us/launch  %7.1f us/minute" % (var, minutes, np.median(ts), np.median(ts) / minutes))
    assert env.sync(raise_on_status=False) == 0
