#!/usr/bin/env python3
"""Launch time of the one-minute step against batch size for several values of defer_min_chunks (from how many chunks per
workgroup the level-2 lanes are set aside; below: every lane in place): the crossover for the shards of a strong-scaling
run (1 Mi envs over 2/4/8 GPUs = 524 288 / 262 144 / 131 072 envs each).  GPU box."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simglucose_amd.batch_env import BatchedT1DSimEnv  # noqa: E402
from simglucose_amd import params, scenario_batch  # noqa: E402
dt = torch.float64 if (len(sys.argv) < 2 or sys.argv[1] == "f64") else torch.float32
_, tab = params.patient_table()
print("%9s" % "envs", " ".join("%8s" % ("d>=%d" % d) for d in (1, 1000)))
for n in (1024, 16384, 65536, 131072, 196608, 262144, 393216, 524288, 1048576):
    pid = np.arange(n) % 30
    row = []
    for d in (1, 1000):
        env = BatchedT1DSimEnv(patient=pid, sensor="Navigator", dtype=dt, n_sub=4, seed=5, extra_outputs=False)
        env.set_option("defer_min_chunks", d)
        g0 = torch.Generator(device=env.device); g0.manual_seed(11)
        start = torch.randint(0, 1440, (n,), generator=g0, device=env.device, dtype=torch.int32)
        mt, ma = scenario_batch.random_meal_tables(n, days=3, start_minute_of_day=start, seed=3, device=env.device, dtype=dt)
        env.set_meals(mt, ma)
        b0 = torch.as_tensor(tab[pid, params.P_COL["u2ss"]] * tab[pid, params.P_COL["BW"]] / 6000.0, dtype=dt, device=env.device)
        g = torch.Generator(device=env.device); g.manual_seed(1)
        pool = [(b0 * 2 * torch.rand(n, generator=g, device=env.device, dtype=dt)).contiguous() for _ in range(4)]
        env.reset()
        for k in range(200):
            env.step(pool[k % 4])
        steps = 1500 if n <= 262144 else 400
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for k in range(steps):
            env.step(pool[k % 4])
        e.record(); torch.cuda.synchronize()
        row.append(s.elapsed_time(e) / steps * 1e3)
        del env, pool
    print("%9d" % n, " ".join("%8.2f" % v for v in row), flush=True)
