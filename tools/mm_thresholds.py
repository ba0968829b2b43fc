#!/usr/bin/env python3
"""Where the persistent multi-minute kernel beats the generic kernels (GPU box): Dexcom steps and PID roll-outs at several batch
sizes, fp64 and fp32, with multi_minute_kernel / rollout_launches forced off (0) and on (2).  The library's default thresholds
(t1d.h "multi_minute_min_envs", "rollout_launches_min_envs") come from this table."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simglucose_amd.batch_env import BatchedT1DSimEnv  # noqa: E402
from simglucose_amd import params, scenario_batch  # noqa: E402


def make(n, dt, opts):
    pid = np.arange(n) % 30
    env = BatchedT1DSimEnv(patient=pid, sensor="Dexcom", dtype=dt, n_sub=4, seed=5, extra_outputs=False)
    for k, v in opts.items():
        env.set_option(k, v)
    g0 = torch.Generator(device=env.device); g0.manual_seed(11)
    start_min = torch.randint(0, 1440, (n,), generator=g0, device=env.device, dtype=torch.int32)
    mt, ma = scenario_batch.random_meal_tables(n, days=3, start_minute_of_day=start_min, seed=3, device=env.device, dtype=dt)
    env.set_meals(mt, ma)
    _, tab = params.patient_table()
    b0 = torch.as_tensor(tab[pid, params.P_COL["u2ss"]] * tab[pid, params.P_COL["BW"]] / 6000.0, dtype=dt, device=env.device)
    g = torch.Generator(device=env.device); g.manual_seed(1)
    pool = [(b0 * 2 * torch.rand(n, generator=g, device=env.device, dtype=dt)).contiguous() for _ in range(4)]
    env.reset()
    return env, pool


print("%-4s %9s  %-22s %-22s" % ("", "envs", "Dexcom step, us: generic / persistent", "PID roll-out step, us: one launch / launch per step"))
for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
    for n in (1 << 16, 1 << 17, 1 << 18, 3 << 17, 1 << 19, 3 << 18, 1 << 20):
        row = []
        for mode in (0, 2):
            env, pool = make(n, dt, {"multi_minute_kernel": mode})
            for k in range(30):
                env.step(pool[k % 4])
            steps = max(40, min(300, (1 << 26) // n))
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); s.record()
            for k in range(steps):
                env.step(pool[k % 4])
            e.record(); torch.cuda.synchronize()
            row.append(s.elapsed_time(e) / steps * 1e3)
            assert env.sync(raise_on_status=False) == 0
            del env, pool
        for mode in (0, 2):
            env, _ = make(n, dt, {"rollout_launches": mode})
            st = env.rollout_pid(40, 1e-3, 1e-5, 1e-3, 140.0)
            st = env.rollout_pid(40, 1e-3, 1e-5, 1e-3, 140.0, pid_state=st)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); s.record()
            st = env.rollout_pid(240, 1e-3, 1e-5, 1e-3, 140.0, pid_state=st)
            e.record(); torch.cuda.synchronize(); row.append(s.elapsed_time(e) / 240 * 1e3)
            assert env.sync(raise_on_status=False) == 0
            del env
        torch.cuda.empty_cache()
        print("%-4s %9d  %9.1f / %-9.1f     %9.1f / %-9.1f" % (tag, n, *row), flush=True)
