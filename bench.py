#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the fused T1D step on MI355X, with the accuracy half of BASELINE.json's metric.

One "step" = one pass of the hot path over the batch = ONE kernel launch advancing every env by one simulated
minute (1-min dt: Navigator-class sensor, sample_time = 1).  Workload: BASELINE.json configs[3] -- 1 048 576
concurrent envs, patient = i mod 30, random-action policy (basal = U(0,2) x the patient's steady-state basal, from a
pool of pre-generated action tensors resident in HBM), per-env random meal tables with every episode starting at a
random minute of the day, fp64, n_sub = 4 (the library's default integrator: the split scheme with per-minute step
sizes, DESIGN.md sections 3-4; `--fixed-step` times level 1 in every minute, `--integrator rk4` classical RK4 on all
13 states), Philox CGM noise.

With --gpus N the 1 Mi envs are sharded over the N ranks (`--scaling strong`, the default: contiguous shards,
env_offset = first global env of the shard, so every env's Philox stream and meal table are those of the one-GPU run;
no data-path collective); `--scaling weak` gives every rank its own --envs instead.  value = all ranks' env-steps /
max-over-ranks wall time of the K timed launches.

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

METRIC = "env-steps/sec at batch=1M patients (1-min dt); fp64 glucose trace max-abs-err"     # BASELINE.json
ALGO_BYTES = {"f64": 352, "f32": 184}       # algorithmic HBM bytes per env-step
HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3}   # vector (non-MFMA) peaks; the fp32 figure counts packed FMAs
PROFILE_DIR = os.path.join(ROOT, "profiles", "r03")


def shard_range(n_total, rank, world):
    """contiguous shards whose sizes differ by at most one (as simglucose_amd/distributed.py; restated here so that the
    rank arithmetic of this file can be exercised without a GPU)"""
    base, extra = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def plan_shard(envs, scaling, rank, world):
    """-> (n_local, env_offset, n_global): what one rank simulates and where its envs sit in the global batch."""
    if scaling == "strong":
        lo, hi = shard_range(envs, rank, world)
        return hi - lo, lo, envs
    return envs, rank * envs, envs * world


def refills_in(t0, n_steps, st, samples_per_block):
    """launches of refill_kernel among steps t0 .. t0 + n_steps - 1 of a lock-step batch: one whenever the sample taken
    at the end of the step opens a new 150-minute noise block (batch_env.py keeps the same clock)"""
    cnt = 0
    for k in range(n_steps):
        t1 = (t0 + k + 1) * st
        if (1 + t1 // st) % samples_per_block == 0:
            cnt += 1
    return cnt


def _cpu_worker(args):
    n_envs, steps, n_sub, sensor, integ, seed = args
    from oracle import t1d_oracle as O
    rs = np.random.RandomState(seed)
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
    return n_envs * steps * int(env.sample_time), time.perf_counter() - t0


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores(cap=16):
    """cores this process can actually keep busy: the cgroup's CPU quota where there is one (a GPU box hands a 16-core
    share of a 256-thread host to each GPU: its affinity mask still shows every thread), else the affinity mask, and
    never more than `cap` workers -- the baseline has to finish in seconds"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = int(f.read()), int(g.read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return max(1, min(n, cap))


def cpu_baseline(n_envs, steps, n_sub, sensor, integ):
    """The CPU oracle (oracle/t1d_oracle.c: a from-scratch C port of the reference path running the SAME integrator as the
    kernel) timed on a bounded sample of the same workload: one host core, then the cores of this process's CPU share."""
    import multiprocessing as mp
    done, dt = _cpu_worker((n_envs, steps, n_sub, sensor, integ, 0))
    cores = usable_cores()
    out = {"value": done / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
           "sample": "%d envs x %d steps, %s integrator n_sub=%d (same scheme as the kernel), fp64, 1 thread, %.1f s" % (n_envs, steps, integ, n_sub, dt),
           "cpu_model": cpu_model(), "cpu_count": os.cpu_count()}
    if cores > 1:
        t0 = time.perf_counter()
        with mp.get_context("fork").Pool(cores) as pool:
            res = pool.map(_cpu_worker, [(n_envs // 2, steps, n_sub, sensor, integ, 1 + k) for k in range(cores)])
        wall = time.perf_counter() - t0
        out["all_cores"] = {"value": sum(r[0] for r in res) / max(r[1] for r in res), "unit": "env-steps/s", "cores": cores,
                            "sample": "%d processes x (%d envs x %d steps), %.1f s wall incl. start-up" % (cores, n_envs // 2, steps, wall)}
    return out


def _oracle_slice(args):
    """one thread of the accuracy replay: the sampled envs [lo, hi) on the CPU oracle, step by step -> BG per step of the
    SciPy-faithful DOPRI5 path, of a tight solve (classical RK4 at 48 sub-steps per minute) and of the kernel's own scheme"""
    pid, sensor, z, pool_s, cho, K, st, integ, n_sub, with_tight = args
    from oracle import t1d_oracle as O
    envs = {"dopri": O.OracleEnv(pid, sensor=sensor, normals=z, integrator="dopri"),
            "same": O.OracleEnv(pid, sensor=sensor, normals=z, integrator=integ, n_sub=n_sub)}
    if with_tight:
        envs["tight"] = O.OracleEnv(pid, sensor=sensor, normals=z, integrator="rk4", n_sub=48)
    out = {k: np.empty((K, len(pid))) for k in envs}
    for e in envs.values():
        e.reset()
    for k in range(K):
        c = cho[k * st:(k + 1) * st]
        for name, e in envs.items():
            out[name][k] = e.step(pool_s[k % len(pool_s)], None, c)["bg"]
    return out


def accuracy_block(env, pool, mt, ma, integ, n_sub, sensor, minutes, n_sample, seed=1, threads=8, rk4_too=True):
    """The accuracy half of the metric on the bench's own workload, outside the timed region: `n_sample` envs spread over
    the batch are replayed on the CPU oracle with the very normals, meals and actions the kernel used -- through the
    SciPy-faithful DOPRI5 path (pinned to the reference's fixtures to ~1e-9), through a tight solve of the same ODE (the
    floor of the comparison: SciPy's own distance from the exact solution) and through the oracle's restatement of the
    kernel's scheme.  The batch then runs the same minutes twice: with the integrator being benchmarked and, for the
    record, with north_star's literal "fixed-step RK4" (classical RK4 on all 13 states, n_sub sub-steps per minute)."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    n = env.n
    rs = np.random.RandomState(seed)
    sample = np.unique(np.concatenate([np.arange(0, min(64, n)), np.arange(max(n - 64, 0), n), rs.randint(0, n, n_sample)]))[:n_sample]
    sidx = torch.as_tensor(sample, device=env.device)
    st = env.minutes_per_step
    K = minutes // st
    z = env.philox_normals(1 + 10 * (2 + minutes // 150), draw0=0, episode=int(env.episode[0].item()) + 1)[:, sidx].cpu().numpy()
    t_s, a_s = mt[:, sidx].cpu().numpy().astype(np.int64), ma[:, sidx].double().cpu().numpy()
    cho = np.zeros((K * st, len(sample)))
    for j in range(len(sample)):
        for tt, aa in zip(t_s[:, j], a_s[:, j]):
            if 0 <= tt < K * st:
                cho[tt, j] = aa
    pool_s = [p[sidx].double().cpu().numpy() for p in pool]
    pid = env.patient_idx[sample]
    # the oracle replays on host threads (the C library releases the GIL; no fork once the GPU is initialised)
    t0 = time.perf_counter()
    nt = max(1, min(threads, len(sample) // 32))
    cuts = np.linspace(0, len(sample), nt + 1).astype(int)
    jobs = [(pid[lo:hi], sensor, np.ascontiguousarray(z[:, lo:hi]), [np.ascontiguousarray(p[lo:hi]) for p in pool_s],
             np.ascontiguousarray(cho[:, lo:hi]), K, st, integ, n_sub, True) for lo, hi in zip(cuts[:-1], cuts[1:])]
    with ThreadPoolExecutor(nt) as ex:
        parts = list(ex.map(_oracle_slice, jobs))
    ref = {k: np.concatenate([p[k] for p in parts], axis=1) for k in parts[0]}
    cpu_s = time.perf_counter() - t0

    def run(opts):
        for name, value in opts:
            env.set_option(name, value)
        env.set_meals(mt, ma)
        env.reset()
        got = torch.empty(K, len(sample), dtype=torch.float64, device=env.device)
        for k in range(K):
            env.step(pool[k % len(pool)])
            got[k] = env.bg[sidx].double()
        return got.cpu().numpy()

    def stats(got, against):
        w = np.abs(got - against).max(axis=0)                  # per-env max over the run
        return {"max_abs_err_mg_dl": float(w.max()), "p99_mg_dl": float(np.percentile(w, 99)), "median_mg_dl": float(np.median(w)),
                "frac_envs_within_1e-3": float((w <= 1e-3).mean())}

    got = run([])
    out = stats(got, ref["dopri"])
    out.update({"vs": "oracle DOPRI5 as SciPy drives it (rtol 1e-6; pinned to the reference's fixtures to ~1e-9)",
                "hip_vs_oracle_same_scheme_max_mg_dl": float(np.abs(got - ref["same"]).max()),
                "vs_tight_solve": stats(got, ref["tight"]),
                "scipy_default_vs_tight_solve": dict(stats(ref["dopri"], ref["tight"]),
                                                     note="the floor of the comparison: the reference's own integrator against classical RK4 at 48 sub-steps per minute, same envs"),
                "envs_sampled": int(len(sample)), "minutes": int(K * st), "quantity": "subcutaneous glucose (BG), per-env max over time",
                "oracle_cpu_seconds": cpu_s, "oracle_threads": nt})
    rk4 = None
    if rk4_too and integ != "rk4":
        got4 = run([("integrator", 0)])
        rk4 = dict(stats(got4, ref["dopri"]), vs_tight_solve=stats(got4, ref["tight"]), integrator="rk4", n_sub=n_sub,
                   note="north_star's literal fixed-step RK4: classical RK4 on all 13 states, same envs, actions, meals and noise")
    return out, rk4


def weak_probe(a, dev, dt, tab, rank, world, steps=300, warmup=300):
    """The workload with a.envs envs on EVERY rank (rank r = envs [r a.envs, (r + 1) a.envs) of the job): barrier, `steps`
    launches, barrier, MAX over ranks -- the figure `--scaling weak` reports, taken beside a strong-scaling run."""
    import torch
    import torch.distributed as dist
    from simglucose_amd.batch_env import BatchedT1DSimEnv
    from simglucose_amd import params, scenario_batch
    n, off = a.envs, rank * a.envs
    pid = (off + np.arange(n, dtype=np.int64)) % 30
    env = BatchedT1DSimEnv(patient=pid, sensor=a.sensor, dtype=dt, device=dev, n_sub=a.n_sub, seed=1234, env_offset=off,
                           noise="philox", extra_outputs=False)
    env.set_option("integrator", {"auto": -1, "rk4": 0, "split": 1}[a.integrator])
    env.set_option("adaptive_gut", 0 if a.fixed_step else (2 if a.in_place else 1))
    st = env.minutes_per_step
    g = torch.Generator(device=dev); g.manual_seed(99 + rank)
    start = torch.zeros(n, dtype=torch.int32, device=dev) if a.midnight_start else torch.randint(0, 1440, (n,), generator=g, device=dev, dtype=torch.int32)
    mt, ma = scenario_batch.random_meal_tables(n, days=1 + (steps + warmup) * st // 1440, start_minute_of_day=start, seed=1000, device=dev, dtype=dt, env_offset=off)
    env.set_meals(mt, ma)
    basal0 = torch.as_tensor(tab[pid, params.P_COL["u2ss"]] * tab[pid, params.P_COL["BW"]] / 6000.0, dtype=dt, device=dev)
    pool = [(basal0 * 2.0 * torch.rand(n, generator=g, device=dev, dtype=torch.float64).to(dt)).contiguous() for _ in range(4)]
    env.reset()
    for k in range(warmup):
        env.step(pool[k % 4])
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        env.step(pool[k % 4])
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    tw = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(tw, op=dist.ReduceOp.MAX)
    status = env.sync(raise_on_status=False)
    wall = float(tw.item())
    return {"scaling": "weak", "value": n * world * steps * st / wall, "unit": "env-steps/s", "envs_per_gpu": n, "envs_total": n * world,
            "steps": steps, "warmup": warmup, "ms_per_step": wall / steps * 1e3, "status_bits": status,
            "note": "the same workload with %d envs on every rank, timed after the headline region (barrier + MAX over ranks as there)" % n}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 1 000 timed launches behind 400 untimed ones (~0.15 s in all).  The first few hundred launches after an
    # idle GPU run ~6 % slower whatever the integrator (clocks ramping), so a short warm-up measures the ramp.
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=400)
    ap.add_argument("--prewarm", type=int, default=400,
                    help="launches ahead of the W warm-up steps that bring an idle GPU up to its running clocks (they advance the "
                         "same envs; reported in config.prewarm_launches)")
    ap.add_argument("--envs", type=int, default=1 << 20, help="envs of the whole job (--scaling strong) or per GPU (--scaling weak)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--dtype", choices=("f64", "f32"), default="f64")
    ap.add_argument("--n-sub", type=int, default=4)
    ap.add_argument("--sensor", default="Navigator")
    ap.add_argument("--integrator", choices=("auto", "rk4", "split"), default="auto")
    ap.add_argument("--fixed-step", action="store_true",
                    help="the split integrator at level 1 in every minute instead of per-minute step sizes (DESIGN.md section 4)")
    ap.add_argument("--midnight-start", action="store_true",
                    help="start every episode at 00:00 (default: a random minute of the day per env, like the reference's gym "
                         "wrapper draws a random start hour, so that every launch sees the day's mix of meal phases)")
    ap.add_argument("--in-place", action="store_true",
                    help="every lane takes its step-size level in place (adaptive_gut = 2) instead of levels 1 and 2 being set aside")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="t1d_ctx_set_option switches applied after the ones above (tuning runs), e.g. --opt minute_launches=0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-accuracy", action="store_true")
    ap.add_argument("--accuracy-envs", type=int, default=1024)
    ap.add_argument("--accuracy-minutes", type=int, default=1440, help="north_star: a 24 h scenario")
    ap.add_argument("--no-weak-probe", action="store_true", help="N > 1, strong scaling: skip the full-batch-per-rank measurement after the timed region")
    ap.add_argument("--rk4-steps", type=int, default=200, help="timed launches of the north_star_rk4 leg (classical RK4, same workload)")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch from a separate rocprofv3 --pmc run (FETCH_SIZE/WRITE_SIZE), copied into roofline.traffic")
    ap.add_argument("--cpu-envs", type=int, default=32768)
    ap.add_argument("--cpu-steps", type=int, default=400)
    ap.add_argument("--plan-only", action="store_true", help="print this rank's shard plan as JSON and exit (no GPU; tests)")
    a = ap.parse_args(argv)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        print("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (a.gpus, a.gpus),
              file=sys.stderr)
        sys.exit(2)
    n, env_offset, n_global = plan_shard(a.envs, a.scaling, rank, world)
    if a.plan_only:
        print(json.dumps({"rank": rank, "world": world, "n_local": n, "env_offset": env_offset, "n_global": n_global, "scaling": a.scaling}))
        return

    # the CPU baseline runs first: its all-core leg forks worker processes, which a process that has initialised the GPU
    # should not do
    integ_name = "rk4" if a.integrator == "rk4" or a.n_sub % 2 or a.n_sub > 8 else ("split" if a.fixed_step else "split_adaptive")
    cpu = None
    if rank == 0 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a.cpu_envs, a.cpu_steps, a.n_sub, a.sensor, integ_name)
        # the reference's own Python path, timed in the build container by tools/time_reference_cpu.py (it cannot travel)
        try:
            with open(os.path.join(PROFILE_DIR, "reference_cpu.json")) as f:
                cpu["reference_python"] = json.load(f)
        except (OSError, ValueError):
            pass

    import torch
    import torch.distributed as dist
    from simglucose_amd.batch_env import BatchedT1DSimEnv
    from simglucose_amd import params, scenario_batch

    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # T1D_BENCH_FORCE_DIST=1 takes the multi-rank code path (RCCL init, barriers, max-reduce) with a single rank too:
    # a one-GPU box can rehearse what the N > 1 launches do
    # (=2: and the weak-scaling probe that otherwise only runs with N > 1)
    use_dist = world > 1 or (os.environ.get("T1D_BENCH_FORCE_DIST") in ("1", "2") and "MASTER_ADDR" in os.environ)
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

    dt = torch.float64 if a.dtype == "f64" else torch.float32
    names, tab = params.patient_table()
    pid = (env_offset + np.arange(n, dtype=np.int64)) % 30                  # patient of the GLOBAL env id
    env = BatchedT1DSimEnv(patient=pid, sensor=a.sensor, dtype=dt, device=dev, n_sub=a.n_sub, seed=1234,
                           env_offset=env_offset, noise="philox", extra_outputs=False)
    env.set_option("integrator", {"auto": -1, "rk4": 0, "split": 1}[a.integrator])
    integ = integ_name
    env.set_option("adaptive_gut", 0 if a.fixed_step else (2 if a.in_place else 1))
    for kv in a.opt:
        name, value = kv.split("=")
        env.set_option(name, int(value))
    st = env.minutes_per_step
    days = 1 + max((a.steps + a.warmup + a.prewarm) * st, a.accuracy_minutes) // 1440
    # per-env inputs as functions of the GLOBAL env id: the shards of an N-rank run are slices of the one-rank run
    gs = torch.Generator(device="cpu"); gs.manual_seed(99)
    start_all = torch.zeros(n_global, dtype=torch.int32) if a.midnight_start else torch.randint(0, 1440, (n_global,), generator=gs, dtype=torch.int32)
    start_min = start_all[env_offset:env_offset + n].to(dev)
    mt, ma = scenario_batch.random_meal_tables(n, days=days, start_minute_of_day=start_min, seed=1000, device=dev, dtype=dt, env_offset=env_offset)
    env.set_meals(mt, ma)
    basal0 = torch.as_tensor(tab[pid, params.P_COL["u2ss"]] * tab[pid, params.P_COL["BW"]] / 6000.0, dtype=dt, device=dev)
    g = torch.Generator(device="cpu"); g.manual_seed(7)
    pool = [(basal0 * 2.0 * torch.rand(n_global, generator=g, dtype=torch.float64)[env_offset:env_offset + n].to(dev, dt)).contiguous() for _ in range(8)]

    accuracy = rk4 = None
    if rank == 0 and not a.no_accuracy:
        accuracy, rk4 = accuracy_block(env, pool, mt, ma, integ, a.n_sub, a.sensor, a.accuracy_minutes, a.accuracy_envs)
        if rk4 is not None:              # north_star's literal integrator on the same workload: time it too (integrator 0 is still set)
            env.set_meals(mt, ma); env.reset()
            for k in range(a.rk4_steps):
                env.step(pool[k % 8])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for k in range(a.rk4_steps):
                env.step(pool[k % 8])
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.rk4_steps
            rk4.update({"kernel_ms": ms, "env_steps_per_s_this_gpu": n * st / (ms * 1e-3), "launches_timed": a.rk4_steps,
                        "roofline_frac": ALGO_BYTES[a.dtype] * n * st / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
        env.set_option("integrator", {"auto": -1, "rk4": 0, "split": 1}[a.integrator])
        env.set_meals(mt, ma)            # restart the meal cursors for the timed episode
    if use_dist:
        # rank 0 spent seconds in the accuracy replay: nobody starts warming up until it is back, so that every rank
        # enters the timed region straight from its warm-up launches
        dist.barrier()
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
    total_env_steps = n_global * a.steps * st
    # the optional exchange of the N > 1 job (north_star: "an optional RCCL gather of observations over xGMI"), outside the
    # timed region: every rank's CGM slice gathered into one [N] tensor on every rank, as a central policy would ask for
    gather = None
    if use_dist:
        from simglucose_amd.distributed import gather_observations
        try:
            full = gather_observations(env.cgm, n_global, force_collective=True)
            ok = bool(torch.equal(full[env_offset:env_offset + n], env.cgm)) and full.numel() == n_global
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            tg = time.perf_counter()
            for _ in range(20):
                full = gather_observations(env.cgm, n_global, force_collective=True)
            torch.cuda.synchronize()
            tgw = torch.tensor([(time.perf_counter() - tg) / 20], dtype=torch.float64, device=dev)
            dist.all_reduce(tgw, op=dist.ReduceOp.MAX)
            gather = {"ms": float(tgw.item()) * 1e3, "bytes_gathered_per_rank": n_global * env.cgm.element_size(), "slices_in_place": ok,
                      "collective": "all_gather_into_tensor (RCCL)", "note": "not part of `value`: the step path needs no collective"}
        except Exception as e:
            gather = {"error": repr(e)}
    # N > 1 under strong scaling: each rank's shard is small and its launch latency-bound; the same job
    # with a full-size batch on every rank (what --scaling weak times) is measured beside it, outside the timed region
    weak = None
    if use_dist and (world > 1 or os.environ.get("T1D_BENCH_FORCE_DIST") == "2") and a.scaling == "strong" and not a.no_weak_probe:
        try:
            weak = weak_probe(a, dev, dt, tab, rank, world)
        except Exception as e:                      # (collective inside: every rank fails or none; the headline line survives)
            weak = {"error": repr(e)}
    bg = env.bg
    sane = bool(torch.isfinite(bg).all()) and status == 0

    if rank == 0:
        tname = "double" if a.dtype == "f64" else "float"
        if integ != "rk4" and st == 1:
            kernel_name = {"split": "void t1d::step1_kernel<%s, 32, false, false>(t1d::KArgs<%s>, int)",
                           "split_adaptive": "void t1d::step1_kernel<%s, 32, false, true>(t1d::KArgs<%s>, int)" if a.in_place
                                             else "void t1d::step1d_kernel<%s, false>(t1d::KArgs<%s>, int)"}[integ] % (tname, tname)
        elif integ != "rk4" and n >= (262144 if a.dtype == "f64" else 393216) and not a.in_place:       # (the library's default thresholds: t1d.h "multi_minute_kernel")
            kernel_name = "void t1d::stepn_kernel<%s, false, false>(t1d::KArgs<%s>, t1d::PidArgs<%s>, int, int, int)" % (tname, tname, tname)
        else:
            kernel_name = "void t1d::step_kernel<%d, %s, false>(t1d::KArgs<%s>)" % ({"rk4": 3, "split": 4, "split_adaptive": 7}[integ], tname, tname)
        # last recorded PMC measurements of this exact configuration (tools/profile_bench.sh): HBM bytes, VALU instructions
        traffic, valu_insts = a.traffic_bytes, None
        try:
            with open(os.path.join(PROFILE_DIR, "traffic.json")) as f:
                tj = json.load(f)
            if (tj["envs"], tj["dtype"], tj["n_sub"], tj["minutes"], tj.get("integrator"), tj.get("kernel")) == (n, a.dtype, a.n_sub, st, integ, kernel_name):
                traffic = tj["traffic_bytes_per_launch"] if traffic is None else traffic
                valu_insts = tj.get("valu_insts_per_launch")
        except (OSError, KeyError, ValueError):
            pass
        # algorithmic bytes per launch (per GPU): the state is read and written once per launch by the persistent multi-minute
        # kernel, once per minute otherwise
        algo = ALGO_BYTES[a.dtype] * n * (1 if "stepn_kernel" in kernel_name else st)
        ach = algo / (kern_ms * 1e-3) / 1e9
        valu = None
        if valu_insts:
            # upper bound on the arithmetic rate: every vector instruction counted as one FMA on 64 lanes
            tf = valu_insts * 64 * 2 / (kern_ms * 1e-3) / 1e12
            valu = {"insts_per_env_step": valu_insts * 64 / (n * st), "achieved": tf, "peak": VALU_PEAK_TFLOPS[a.dtype], "unit": "TFLOP/s",
                    "frac": tf / VALU_PEAK_TFLOPS[a.dtype], "note": "SQ_INSTS_VALU x 64 lanes x 2 flop (every vector instruction counted as an FMA): an upper bound"}
        clock0 = a.prewarm + a.warmup
        out = {
            "metric": METRIC, "value": total_env_steps / wall,
            "unit": "env-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": wall / a.steps * 1e3, "higher_is_better": True, "scaling": a.scaling,
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": "configs[3]: %d envs in all (%d on this GPU), patient=i mod 30, random-action policy, "
                                   "random meal tables (episodes start at %s), %s sensor (sample_time %d min), %s integrator n_sub=%d, Philox CGM noise"
                                   % (n_global, n, "00:00" if a.midnight_start else "a random minute of the day per env", a.sensor, st, integ, a.n_sub),
                       "envs_total": n_global, "envs_per_gpu": n, "n_sub": a.n_sub, "integrator": integ, "minutes_per_launch": st,
                       "prewarm_launches": a.prewarm, "untimed_launches_before_timing": a.prewarm + a.warmup,
                       "refill_kernel_launches_in_timed_region": refills_in(clock0, a.steps, st, int(150 // env.sample_time)),
                       "parallelism": "env-shard x%d" % world},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": kernel_name,
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_env_step": ALGO_BYTES[a.dtype], "valu": valu},
            "accuracy": accuracy, "north_star_rk4": rk4, "weak_scaling_probe": weak, "observation_gather": gather,
            "sane": sane, "status_bits": status,
            "bg_mean": float(bg.mean()), "bg_min": float(bg.min()), "bg_max": float(bg.max()),
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
