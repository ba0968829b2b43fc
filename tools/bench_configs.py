#!/usr/bin/env python3
"""Secondary measurements of BASELINE.json's other configs (GPU box): launch-per-step at several batch
sizes / dtypes / sensors and the in-kernel PID roll-out (config 5).  Prints one JSON object."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simglucose_amd.batch_env import BatchedT1DSimEnv  # noqa: E402
from simglucose_amd import params, scenario_batch  # noqa: E402


def make(n, dt, sensor, n_sub=4, days=8):
    pid = np.arange(n) % 30
    env = BatchedT1DSimEnv(patient=pid, sensor=sensor, dtype=dt, n_sub=n_sub, seed=5, extra_outputs=False)
    # episodes start at a random minute of the day per env: every launch sees the day's mix of meal phases
    g0 = torch.Generator(device=env.device); g0.manual_seed(11)
    start_min = torch.randint(0, 1440, (n,), generator=g0, device=env.device, dtype=torch.int32)
    mt, ma = scenario_batch.random_meal_tables(n, days=days, start_minute_of_day=start_min, seed=3, device=env.device, dtype=dt)
    env.set_meals(mt, ma)
    _, tab = params.patient_table()
    b0 = torch.as_tensor(tab[pid, params.P_COL["u2ss"]] * tab[pid, params.P_COL["BW"]] / 6000.0, dtype=dt, device=env.device)
    g = torch.Generator(device=env.device); g.manual_seed(1)
    pool = [(b0 * 2 * torch.rand(n, generator=g, device=env.device, dtype=dt)).contiguous() for _ in range(4)]
    env.reset()
    return env, pool


def time_steps(env, pool, steps):
    for k in range(5):
        env.step(pool[k % 4])
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t0 = time.perf_counter(); s.record()
    for k in range(steps):
        env.step(pool[k % 4])
    e.record(); torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return s.elapsed_time(e) / steps * 1e3, wall / steps * 1e6


out = {}
for name, n, dt, sensor, steps in (("config2_1024_f64_navigator", 1024, torch.float64, "Navigator", 2000),
                                   ("config3_61440_f32_dexcom", 61440, torch.float32, "Dexcom", 500),
                                   ("n131072_f64_navigator", 131072, torch.float64, "Navigator", 500),
                                   ("n1M_f32_navigator", 1 << 20, torch.float32, "Navigator", 100),
                                   ("n1M_f64_dexcom", 1 << 20, torch.float64, "Dexcom", 100),
                                   ("n4M_f64_navigator", 1 << 22, torch.float64, "Navigator", 30)):
    env, pool = make(n, dt, sensor)
    gpu_us, wall_us = time_steps(env, pool, steps)
    m = env.minutes_per_step
    out[name] = {"us_per_launch_gpu": gpu_us, "us_per_launch_wall": wall_us, "minutes_per_launch": m,
                 "env_steps_per_s": n * m / (wall_us * 1e-6), "status": env.sync(raise_on_status=False)}
    del env, pool
    torch.cuda.empty_cache()
# config 5: 262 144 envs, in-kernel PID, 7 days, launches of 480 steps (= 1 day with Dexcom)
for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
    n = 262144
    env, _ = make(n, dt, "Dexcom", days=8)
    st = None
    env.rollout_pid(10, 1e-3, 1e-5, 1e-3, 140.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for day in range(7):
        st = env.rollout_pid(480, 1e-3, 1e-5, 1e-3, 140.0, pid_state=st)
    torch.cuda.synchronize(); wall = time.perf_counter() - t0
    out["config5_pid_262144_%s" % tag] = {"seconds_for_7_days": wall, "env_steps_per_s": n * 7 * 1440 / wall,
                                          "status": env.sync(raise_on_status=False),
                                          "bg_mean": float(env.bg.double().mean())}
    del env
    torch.cuda.empty_cache()
# config 1 at scale: 30 patients x 2 048 seeds, in-kernel BBController, 24 h (480 Dexcom steps) in one launch
n = 61440
env, _ = make(n, torch.float64, "Dexcom", days=2)
env.rollout_bb(10)
torch.cuda.synchronize(); t0 = time.perf_counter()
env.rollout_bb(480)
torch.cuda.synchronize(); wall = time.perf_counter() - t0
out["config1_bb_61440_f64_24h"] = {"seconds_for_24_h": wall, "env_steps_per_s": n * 1440 / wall,
                                   "status": env.sync(raise_on_status=False), "bg_mean": float(env.bg.double().mean())}
del env
torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
