#!/usr/bin/env python3
"""Where a small batch's launch time goes (GPU box): per batch size the time per env.step by HIP events around a loop of
steps (what a caller sees: includes the host's python + ctypes + launch path) and, when run under `rocprofv3 --kernel-trace
--stats`, the kernel's own duration to hold against it.  usage: launch_floor.py [f64|f32] [sizes...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simglucose_amd.batch_env import BatchedT1DSimEnv  # noqa: E402
from simglucose_amd import params, scenario_batch  # noqa: E402
dt = torch.float64 if (len(sys.argv) < 2 or sys.argv[1] == "f64") else torch.float32
sizes = [int(v) for v in sys.argv[2:]] or [64, 1024, 16384, 131072]
_, tab = params.patient_table()
for n in sizes:
    pid = np.arange(n) % 30
    env = BatchedT1DSimEnv(patient=pid, sensor="Navigator", dtype=dt, n_sub=4, seed=5, extra_outputs=False)
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
    steps = 1500
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t0 = time.perf_counter(); s.record()
    for k in range(steps):
        env.step(pool[k % 4])
    t1 = time.perf_counter()
    e.record(); torch.cuda.synchronize()
    print("n %8d  events %7.2f us/step   host loop (enqueue only) %7.2f us/step" % (n, s.elapsed_time(e) / steps * 1e3, (t1 - t0) / steps * 1e6), flush=True)
    del env, pool
