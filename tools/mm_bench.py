#!/usr/bin/env python3
"""Timing of the multi-minute paths (GPU box): a step of sample_time minutes per launch and the closed-loop roll-outs, at the
shapes VERDICT/BASELINE name.  usage: mm_bench.py [name=value ctx options ...]   (T1D_LIB_PATH selects an A/B build)"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simglucose_amd.batch_env import BatchedT1DSimEnv  # noqa: E402
from simglucose_amd import params, scenario_batch  # noqa: E402
opts = [kv.split("=") for kv in sys.argv[1:]]


def make(n, dt, sensor, days=8):
    pid = np.arange(n) % 30
    env = BatchedT1DSimEnv(patient=pid, sensor=sensor, dtype=dt, n_sub=4, seed=5, extra_outputs=False)
    for k, v in opts:
        env.set_option(k, int(v))
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


out = {}
if os.environ.get("MM_SWEEP"):
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        for n in (1 << 16, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 22):
            env, pool = make(n, dt, "Dexcom", days=2)
            for k in range(20):
                env.step(pool[k % 4])
            steps = max(20, min(300, (1 << 25) // n))
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); s.record()
            for k in range(steps):
                env.step(pool[k % 4])
            e.record(); torch.cuda.synchronize()
            print("%s n=%8d  %8.2f us/step" % (tag, n, s.elapsed_time(e) / steps * 1e3), env.sync(raise_on_status=False))
            del env, pool
            torch.cuda.empty_cache()
    sys.exit(0)
for name, n, dt, sensor, steps in (("n1M_f64_dexcom", 1 << 20, torch.float64, "Dexcom", 150), ("n1M_f64_guardian", 1 << 20, torch.float64, "GuardianRT", 100),
                                   ("n256k_f64_dexcom", 1 << 18, torch.float64, "Dexcom", 300), ("config3_61440_f32_dexcom", 61440, torch.float32, "Dexcom", 500),
                                   ("n1M_f32_dexcom", 1 << 20, torch.float32, "Dexcom", 150)):
    env, pool = make(n, dt, sensor)
    for k in range(30):
        env.step(pool[k % 4])
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for k in range(steps):
        env.step(pool[k % 4])
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / steps * 1e3
    out[name] = {"us_per_step": round(us, 2), "env_steps_per_s": "%.3e" % (n * env.minutes_per_step / (us * 1e-6)), "status": env.sync(raise_on_status=False)}
    del env, pool
    torch.cuda.empty_cache()
for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
    n = 262144
    env, _ = make(n, dt, "Dexcom")
    st = env.rollout_pid(10, 1e-3, 1e-5, 1e-3, 140.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for day in range(3):
        st = env.rollout_pid(480, 1e-3, 1e-5, 1e-3, 140.0, pid_state=st)
    torch.cuda.synchronize(); wall = time.perf_counter() - t0
    out["config5_pid_262144_%s" % tag] = {"us_per_step": round(wall / 1440 * 1e6, 2), "env_steps_per_s": "%.3e" % (n * 3 * 1440 / wall), "status": env.sync(raise_on_status=False)}
    del env
    torch.cuda.empty_cache()
env, _ = make(61440, torch.float64, "Dexcom", days=2)
env.rollout_bb(10)
torch.cuda.synchronize(); t0 = time.perf_counter()
env.rollout_bb(480)
torch.cuda.synchronize(); wall = time.perf_counter() - t0
out["config1_bb_61440_f64_24h"] = {"us_per_step": round(wall / 480 * 1e6, 2), "env_steps_per_s": "%.3e" % (61440 * 1440 / wall), "status": env.sync(raise_on_status=False)}
for k, v in out.items():
    print("%-28s %s" % (k, json.dumps(v)))
