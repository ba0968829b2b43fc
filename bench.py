#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the fused T1D step on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over the batch = ONE kernel launch advancing every env by
one simulated minute (1-min dt: Navigator-class sensor, sample_time = 1).  Workload at N = 1:
BASELINE.json configs[3]'s batch -- 1 048 576 concurrent envs, patient = i mod 30, random-action
policy (basal = U(0,2) x the patient's steady-state basal, from a pool of pre-generated action
tensors resident in HBM), per-env random meal tables, fp64, n_sub = 4 sub-steps per minute (the library's default
"split" fixed-step integrator: exact insulin propagator + RK4 gut/glucose, same error vs SciPy as RK4(4) on all
13 states -- DESIGN.md section 3; `--integrator rk4` times classical RK4; the gut sub-steps are halved in the
minutes that cross a gastric-emptying transition fast unless `--fixed-step`), Philox CGM noise.
With --gpus N each rank owns its own 1 Mi envs (weak scaling; independent episodes, no data-path
collective); value = all ranks' env-steps / max-over-ranks wall time.

    python bench.py --gpus 1 --steps 1000 --warmup 400
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES = {"f64": 352, "f32": 184}       # algorithmic HBM bytes per env-step
HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(n_envs, steps, n_sub, sensor, integ="split"):
    """The CPU oracle (oracle/t1d_oracle.c, a from-scratch port of the reference path with the same
    RK4 integrator) timed on one host core over a bounded sample of the same workload."""
    from oracle import t1d_oracle as O
    rs = np.random.RandomState(0)
    pid = np.arange(n_envs) % 30
    env = O.OracleEnv(pid, sensor=sensor, normals=rs.randn(64, n_envs), integrator=integ, n_sub=n_sub)
    env.reset()
    names, tab = O.patient_table()
    basal0 = tab[pid, O.IDX["u2ss"]] * tab[pid, O.IDX["BW"]] / 6000.0
    acts = [basal0 * rs.uniform(0, 2, n_envs) for _ in range(4)]
    cho = np.zeros((int(env.sample_time), n_envs))
    env.step(acts[0], None, cho)
    t0 = time.perf_counter()
    for k in range(steps):
        cho[:] = 0.0
        if k % 30 == 7:
            cho[0, (np.arange(n_envs) + k) % 5 == 0] = 50.0
        env.step(acts[k % 4], None, cho)
    dt = time.perf_counter() - t0
    return {"value": n_envs * steps * int(env.sample_time) / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": "%d envs x %d steps, %s integrator n_sub=%d (same scheme as the kernel), fp64, 1 thread, %.1f s" % (n_envs, steps, integ, n_sub, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 1 000 timed launches behind 400 untimed ones (~0.13 s in all).  The first few hundred launches after an
    # idle GPU run ~6 % slower whatever the integrator (clocks ramping: 87.7 us per launch with 20 warm-up launches,
    # 82.3 with 400 and with 2 000, fixed steps), so a short warm-up measures the ramp, not the steady state.
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=400)
    ap.add_argument("--prewarm", type=int, default=400,
                    help="launches ahead of the W warm-up steps that bring an idle GPU up to its running clocks (they advance the "
                         "same envs; reported in config.prewarm_launches)")
    ap.add_argument("--envs", type=int, default=1 << 20, help="envs per GPU")
    ap.add_argument("--dtype", choices=("f64", "f32"), default="f64")
    ap.add_argument("--n-sub", type=int, default=4)
    ap.add_argument("--sensor", default="Navigator")
    ap.add_argument("--integrator", choices=("auto", "rk4", "split"), default="auto")
    ap.add_argument("--fixed-step", action="store_true",
                    help="switch off the split integrator's adaptive gut refinement (the library default keeps it on: DESIGN.md section 4)")
    ap.add_argument("--midnight-start", action="store_true",
                    help="start every episode at 00:00 (default: a random minute of the day per env, like the reference's gym "
                         "wrapper draws a random start hour, so that every launch sees the day's mix of meal phases)")
    ap.add_argument("--in-place", action="store_true",
                    help="adaptive refinement in place (adaptive_gut = 2) instead of deferred to the end of the launch")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="t1d_ctx_set_option switches applied after the ones above (tuning runs), e.g. --opt dreg_max_chunks=0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch from a separate rocprofv3 --pmc run (FETCH_SIZE/WRITE_SIZE), copied into roofline.traffic")
    ap.add_argument("--cpu-envs", type=int, default=65536)
    ap.add_argument("--cpu-steps", type=int, default=600)
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from simglucose_amd.batch_env import BatchedT1DSimEnv
    from simglucose_amd import params, scenario_batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        print("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (a.gpus, a.gpus),
              file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # T1D_BENCH_FORCE_DIST=1 takes the multi-rank code path (RCCL init, barriers, max-reduce) with a single rank too:
    # a one-GPU box can rehearse what the N > 1 launches do
    use_dist = world > 1 or (os.environ.get("T1D_BENCH_FORCE_DIST") == "1" and "MASTER_ADDR" in os.environ)
    if use_dist:
        # RCCL prints its version banner on stdout while the communicator comes up: send that to stderr, so that
        # stdout carries nothing but the one JSON line
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=dev)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    n = a.envs
    dt = torch.float64 if a.dtype == "f64" else torch.float32
    names, tab = params.patient_table()
    pid = np.arange(n, dtype=np.int64) % 30
    env = BatchedT1DSimEnv(patient=pid, sensor=a.sensor, dtype=dt, device=dev, n_sub=a.n_sub, seed=1234,
                           env_offset=rank * n, noise="philox", extra_outputs=False)
    env.set_option("integrator", {"auto": -1, "rk4": 0, "split": 1}[a.integrator])
    integ = "rk4" if a.integrator == "rk4" or a.n_sub % 2 or a.n_sub > 8 else "split"
    env.set_option("adaptive_gut", 0 if a.fixed_step else (2 if a.in_place else 1))
    if integ == "split" and not a.fixed_step:
        integ = "split_adaptive"
    for kv in a.opt:
        name, value = kv.split("=")
        env.set_option(name, int(value))
    days = 1 + (a.steps + a.warmup + a.prewarm) * env.minutes_per_step // 1440
    gs = torch.Generator(device=dev); gs.manual_seed(99 + rank)
    start_min = 0 if a.midnight_start else torch.randint(0, 1440, (n,), generator=gs, device=dev, dtype=torch.int32)
    mt, ma = scenario_batch.random_meal_tables(n, days=days, start_minute_of_day=start_min, seed=1000, device=dev, dtype=dt, env_offset=rank * n)
    env.set_meals(mt, ma)
    basal0 = torch.as_tensor(tab[pid, params.P_COL["u2ss"]] * tab[pid, params.P_COL["BW"]] / 6000.0, dtype=dt, device=dev)
    g = torch.Generator(device=dev); g.manual_seed(7 + rank)
    pool = [(basal0 * 2.0 * torch.rand(n, generator=g, device=dev, dtype=dt)).contiguous() for _ in range(8)]
    env.reset()
    for k in range(a.prewarm + a.warmup):
        env.step(pool[k % 8])
    # One HIP event pair around the K launches of the timed region, on torch's current stream (the stream t1d_step
    # launches on): mean launch duration = elapsed / K.  (Bracketing single launches puts event packets between
    # consecutive kernels and reads 5-10 % high; the noise-block refill kernel runs once per 150 steps in between.)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for k in range(a.steps):
        env.step(pool[k % 8])
    ev1.record()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    status = env.sync(raise_on_status=False)
    if use_dist:
        tw = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
    kern_ms = ev0.elapsed_time(ev1) / a.steps
    minutes = env.minutes_per_step
    total_env_steps = world * n * a.steps * minutes
    bg = env.bg
    sane = bool(torch.isfinite(bg).all()) and status == 0

    if rank == 0:
        tname = "double" if a.dtype == "f64" else "float"
        if integ != "rk4" and minutes == 1:
            kernel_name = {"split": "t1d::step1_kernel<%s, 32, false, false>",
                           "split_adaptive": "t1d::step1_kernel<%s, 32, false, true>" if a.in_place
                                             else "t1d::step1d_kernel<%s, false, false>"}[integ] % tname
        else:
            kernel_name = "t1d::step_kernel<%d, %s, false>" % ({"rk4": 3, "split": 4, "split_adaptive": 7}[integ], tname)
        traffic = a.traffic_bytes
        if traffic is None:                               # last recorded PMC measurement of this exact configuration
            try:
                with open(os.path.join(ROOT, "profiles", "r01", "traffic.json")) as f:
                    tj = json.load(f)
                if (tj["envs"], tj["dtype"], tj["n_sub"], tj["minutes"], tj.get("integrator", "rk4"), tj.get("kernel", "")) == \
                        (n, a.dtype, a.n_sub, minutes, integ, kernel_name):
                    traffic = tj["traffic_bytes_per_launch"]
            except (OSError, KeyError, ValueError):
                traffic = None
        algo = ALGO_BYTES[a.dtype] * n * minutes          # algorithmic bytes per launch (per GPU)
        ach = algo / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "env-steps/sec at batch=1M patients (1-min dt)", "value": total_env_steps / wall,
            "unit": "env-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": wall / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": "configs[3]: %d envs per GPU, patient=i mod 30, random-action policy, "
                                   "random meal tables (episodes start at %s), %s sensor (sample_time %d min), %s integrator n_sub=%d, Philox CGM noise"
                                   % (n, "00:00" if a.midnight_start else "a random minute of the day per env", a.sensor, minutes, integ, a.n_sub),
                       "envs_per_gpu": n, "n_sub": a.n_sub, "integrator": integ, "minutes_per_launch": minutes, "prewarm_launches": a.prewarm, "parallelism": "env-shard x%d" % world},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": kernel_name,
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_env_step": ALGO_BYTES[a.dtype]},
            "sane": sane, "status_bits": status,
            "bg_mean": float(bg.mean()), "bg_min": float(bg.min()), "bg_max": float(bg.max()),
        }
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(a.cpu_envs, a.cpu_steps, a.n_sub, a.sensor, integ)
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
