#!/usr/bin/env python3
"""BASELINE config 5: 262 144 envs, in-kernel PID closed loop, 7-day horizon, fp32 vs fp64 (GPU box only).

Both precisions run the same episodes (same Philox streams, same meal tables); the BG history of every env is
kept on the device and compared step by step.  Prints one JSON object: per PID setting the BG range, the time
in range, and percentiles over envs of max_t |BG_f32 - BG_f64| after 1, 3 and 7 days."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simglucose_amd.batch_env import BatchedT1DSimEnv
from simglucose_amd import scenario_batch
from simglucose_amd.analysis import report

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
days, st = 7, 3
K = days * 1440 // st
pid = np.arange(n) % 30
out = {"envs": n, "days": days, "steps": K, "sensor": "Dexcom"}
for name, (P, I, D) in (("reference_example", (1e-3, 1e-5, 1e-3)), ("gentle", (1.5e-4, 4e-7, 5e-4))):
    tr = {}
    res = {}
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        env = BatchedT1DSimEnv(patient=pid, sensor="Dexcom", dtype=dt, n_sub=4, seed=5, extra_outputs=False)
        mt, ma = scenario_batch.random_meal_tables(n, days=days + 1, seed=3, device=env.device, dtype=dt)
        env.set_meals(mt, ma)
        env.reset()
        t = env.new_trace(K, columns=("bg",))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        state = None
        for d in range(days):
            state = env.rollout_pid(K // days, P, I, D, 140.0, pid_state=state, trace=t)
        torch.cuda.synchronize()
        res["seconds_" + tag] = time.perf_counter() - t0
        res["status_" + tag] = env.sync(raise_on_status=False)
        tr[tag] = t["bg"]
        del env
    b64 = tr["f64"]
    s = report.outcome_stats(b64, risk_trace=False)
    res["bg_min_f64"], res["bg_max_f64"] = float(b64.min()), float(b64.max())
    res["percent_in_range_mean"] = float(s["percent"][2].mean())
    res["cvga_zone_counts_A_to_E_none"] = [int((s["zone"] == k).sum()) for k in range(6)]
    diff = (tr["f32"].double() - b64).abs()
    for d in (1, 3, 7):
        m = diff[:d * 1440 // st + 1].max(0).values
        res["max_abs_diff_day%d" % d] = {"p50": float(m.median()), "p99": float(torch.quantile(m[:1 << 20], 0.99)) if m.numel() > (1 << 20) else float(torch.quantile(m, 0.99)), "max": float(m.max())}
    out[name] = res
    del tr, diff, b64
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
