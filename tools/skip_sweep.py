#!/usr/bin/env python3
"""Tuning aid: time the step kernel with pieces switched off (debug bits in batch.flags)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.ab_step import make  # noqa: E402
n = 1 << 20
env, pool = make(n, "run64", torch.float64, "Navigator", 4)
base = env._flags0
for pm, pipe in ((0, 1), (1, 1), (0, 0)):
    env.set_option("params_mode", pm); env.set_option("pipeline", pipe); env.set_option("pipe_stagger", 1 if pipe else 0)
    for name, bits in (("full", 0), ("no_risk", 0x100), ("no_pump", 0x200), ("no_noise", 0x400), ("no_rk4", 0x800),
                       ("no_risk_pump_noise", 0x700), ("only_memory", 0xF00)):
        env._flags0 = base | bits
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
        print("params_mode=%d pipe=%d %-20s median %7.1f us  min %7.1f us" % (pm, pipe, name, np.median(ts), np.min(ts)))
env._flags0 = base
