#!/usr/bin/env python3
"""Tuning aid (needs a -DT1D_AB_FLAGS=1 build in T1D_LIB_PATH): the one-minute launch at 1 024 and 131 072 envs as shipped, with
level 1 in every minute, without the integration (flag 0x800), without integration and risk index (0x900) and in place:
what the group of level-2 lanes that ends a small launch costs (profiles/r03/small_batches.log).  GPU box."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simglucose_amd.batch_env import BatchedT1DSimEnv
from simglucose_amd import params, scenario_batch
dt = torch.float64
_, tab = params.patient_table()
for n in (1024, 131072):
    for name, opts, flags in (("default", {}, 0), ("level 1 everywhere", {"adaptive_gut": 0}, 0), ("no integration", {}, 0x800), ("no integration, no risk", {}, 0x900), ("in place", {"adaptive_gut": 2}, 0)):
        pid = np.arange(n) % 30
        env = BatchedT1DSimEnv(patient=pid, sensor="Navigator", dtype=dt, n_sub=4, seed=5, extra_outputs=False)
        for k, v in opts.items(): env.set_option(k, v)
        env._flags0 |= flags; env._b.flags = env._flags0
        g0 = torch.Generator(device=env.device); g0.manual_seed(11)
        start = torch.randint(0, 1440, (n,), generator=g0, device=env.device, dtype=torch.int32)
        mt, ma = scenario_batch.random_meal_tables(n, days=3, start_minute_of_day=start, seed=3, device=env.device, dtype=dt)
        env.set_meals(mt, ma)
        b0 = torch.as_tensor(tab[pid, params.P_COL["u2ss"]] * tab[pid, params.P_COL["BW"]] / 6000.0, dtype=dt, device=env.device)
        g = torch.Generator(device=env.device); g.manual_seed(1)
        pool = [(b0 * 2 * torch.rand(n, generator=g, device=env.device, dtype=dt)).contiguous() for _ in range(4)]
        env.reset()
        for k in range(200): env.step(pool[k % 4])
        steps = 1500
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for k in range(steps): env.step(pool[k % 4])
        e.record(); torch.cuda.synchronize()
        print("n %8d %-24s %7.2f us/step" % (n, name, s.elapsed_time(e) / steps * 1e3), flush=True)
        del env, pool
