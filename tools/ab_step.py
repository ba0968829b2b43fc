#!/usr/bin/env python3
"""A/B timing of the step-kernel variants in ONE process, interleaved rounds (GPU box only).

    python tools/ab_step.py [--envs 1048576] [--rounds 5] [--steps 50] [--dtype f64]

Variants: ref = ocml tanh / IEEE division RHS; lds = fast RHS, parameters re-read from LDS;
scalar = fast RHS, wave-uniform patients with parameters in SGPRs.  Layouts: "mod30" patient =
i mod 30 (every wave mixes patients), "run64" patient = (i // 64) mod 30 (patient-homogeneous waves).
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simglucose_amd.batch_env import BatchedT1DSimEnv  # noqa: E402
from simglucose_amd import params, scenario_batch  # noqa: E402


def make(n, layout, dt, sensor, n_sub, start="random"):
    pid = (np.arange(n) % 30) if layout == "mod30" else ((np.arange(n) // 64) % 30)
    env = BatchedT1DSimEnv(patient=pid, sensor=sensor, dtype=dt, n_sub=n_sub, seed=5, extra_outputs=False)
    g0 = torch.Generator(device=env.device); g0.manual_seed(11)
    start_min = torch.randint(0, 1440, (n,), generator=g0, device=env.device, dtype=torch.int32) if start == "random" else 0
    mt, ma = scenario_batch.random_meal_tables(n, days=2, start_minute_of_day=start_min, seed=3, device=env.device, dtype=dt)
    env.set_meals(mt, ma)
    names, tab = params.patient_table()
    b0 = torch.as_tensor(tab[pid, params.P_COL["u2ss"]] * tab[pid, params.P_COL["BW"]] / 6000.0, dtype=dt, device=env.device)
    g = torch.Generator(device=env.device); g.manual_seed(1)
    pool = [(b0 * 2 * torch.rand(n, generator=g, device=env.device, dtype=dt)).contiguous() for _ in range(4)]
    env.reset()
    return env, pool


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1 << 20)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--sensor", default="Navigator")
    ap.add_argument("--n-sub", type=int, default=4)
    ap.add_argument("--variants", default="reg,g_split_reg,split_lds,split_reg")
    ap.add_argument("--start", choices=("random", "midnight"), default="random",
                    help="episode start: a random minute of the day per env (every launch sees the day's mix of meal phases) or 00:00 for all")
    a = ap.parse_args()
    dt = torch.float64 if a.dtype == "f64" else torch.float32
    cfgs = []
    for layout in ("mod30", "run64"):
        env, pool = make(a.envs, layout, dt, a.sensor, a.n_sub, a.start)
        for var in a.variants.split(","):
            if var in ("scalar", "pipe_scalar") and not env.wave_uniform:
                continue
            cfgs.append((layout, var, env, pool))
    res = {(l, v): [] for l, v, _, _ in cfgs}
    for r in range(a.rounds):
        for layout, var, env, pool in cfgs:
            env.set_option("math", 0 if var == "ref" else 1)
            env.set_option("scalar_params", 1 if var in ("scalar", "pipe_scalar") else 0)
            env.set_option("params_mode", 1 if var.endswith("reg") else 0)
            env.set_option("integrator", 1 if "split" in var else 0)
            env.set_option("adaptive_gut", (2 if "inplace" in var else 3 if "defer" in var else 1) if "adapt" in var else 0)   # adapt: the library's choice; adapt_defer: deferred refinement wherever it fits; adapt_inplace: masked half steps in place
            env.set_option("single_minute_kernel", 0 if var.startswith("g_") else 1)      # g_: generic one-tile-per-block kernel
            env.set_option("pipeline", 1 if var.startswith("pipe") else 0)
            for k in range(3):
                env.step(pool[k % 4])
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            s.record()
            for k in range(a.steps):
                env.step(pool[k % 4])
            e.record()
            torch.cuda.synchronize()
            res[(layout, var)].append(s.elapsed_time(e) / a.steps * 1e3)
    bytes_per = 352 if a.dtype == "f64" else 184
    out = {}
    for (l, v), ts in res.items():
        med = float(np.median(ts)); mn = float(np.min(ts))
        out["%s/%s" % (l, v)] = {"us_median": med, "us_min": mn,
                                 "env_steps_per_s": a.envs * env.minutes_per_step / (med * 1e-6),
                                 "algo_GBps": bytes_per * a.envs * env.minutes_per_step / (med * 1e-6) / 1e9}
        print("%-14s median %8.1f us  min %8.1f us  %.3e env-steps/s  %.0f GB/s algorithmic" % (
            l + "/" + v, med, mn, out[l + "/" + v]["env_steps_per_s"], out[l + "/" + v]["algo_GBps"]))
    for _, _, env, _ in cfgs:
        st = env.sync(raise_on_status=False)
        assert st == 0, st
    print(json.dumps({"lib": os.environ.get("T1D_LIB_PATH", "default"), "envs": a.envs, "dtype": a.dtype, "results": out}))


if __name__ == "__main__":
    main()
