#!/usr/bin/env python3
"""Sweep the persistent kernel's grid size (GPU box)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.ab_step import make  # noqa: E402
n = 1 << 20
env, pool = make(n, "run64", torch.float64, "Navigator", 4)
for scalar in (0, 1):
    env.set_option("scalar_params", scalar)
    for blocks in (0, 256, 384, 512, 640, 768, 1024, 2048, 4096):
        env.set_option("pipeline", 1 if blocks else 0); env.set_option("pipe_blocks", blocks if blocks else 0)
        for k in range(3):
            env.step(pool[k % 4])
        ts = []
        for r in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); s.record()
            for k in range(20):
                env.step(pool[k % 4])
            e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / 20 * 1e3)
        print("scalar=%d pipe_blocks=%4d  %8.1f us" % (scalar, blocks, np.median(ts)))
assert env.sync(raise_on_status=False) == 0
