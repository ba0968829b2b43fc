"""GPU parity tests at the shapes BASELINE.json states for configs 3 and 5, and known answers of single pieces of the path
through the C ABI: the ODE right-hand side on its own (fixture G1) and the Philox draw of the random initial glucose.

Tolerances:
  * fp64 kernel vs the oracle's restatement of the same scheme: 1e-8 mg/dL open loop, 1e-6 over 7 days of closed loop
    with the reference test's PID gains (which wind up and take most virtual patients through BG = 0 within days).
  * fp32 kernel vs the fp64 oracle: DESIGN.md section 4 -- 0.05 mg/dL over a day of open loop with meals (measured
    8e-3), and distribution bounds over 7 days of closed loop (median 0.02, 90th percentile 0.3 mg/dL).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mk(**kw):
    import torch
    from simglucose_amd.batch_env import BatchedT1DSimEnv
    assert torch.cuda.is_available()
    return BatchedT1DSimEnv(**kw)


def _basal(pid):
    from simglucose_amd import params
    _, tab = params.patient_table()
    return tab[pid, params.P_COL["u2ss"]] * tab[pid, params.P_COL["BW"]] / 6000.0


def _dense_cho(mt, ma, sample, minutes):
    t_s, a_s = mt[:, sample].cpu().numpy().astype(np.int64), ma[:, sample].double().cpu().numpy()
    cho = np.zeros((minutes, len(sample)))
    for j in range(len(sample)):
        for tt, aa in zip(t_s[:, j], a_s[:, j]):
            if 0 <= tt < minutes:
                cho[tt, j] = aa
    return cho


def test_config3_61440_envs_fp32_24h_sampled_envs_match_fp64_oracle():
    """BASELINE config 3 at its stated shape: 30 virtual patients x 2 048 seeds = 61 440 envs, meal scenario (per-env
    random meal tables), fp32, Dexcom (3-minute env.steps), 24 h, random-action policy.  300 envs sampled across the
    batch are replayed on the fp64 oracle (the kernel's own scheme) with the very Philox normals, meals and actions the
    kernel used; the fp32 kernel must stay within 0.05 mg/dL on BG and CGM (measured ~0.01: DESIGN.md section 4)."""
    import torch
    from simglucose_amd import scenario_batch as sb
    from oracle import t1d_oracle as O
    n, K, st = 30 * 2048, 480, 3
    pid = np.arange(n) % 30                                  # patient = i mod 30, seed = i div 30
    e = _mk(patient=pid, sensor="Dexcom", dtype=torch.float32, noise="philox", seed=33, n_sub=4)
    mt, ma = sb.random_meal_tables(n, days=1, start_minute_of_day=0, seed=12, device=e.device, dtype=torch.float32)
    e.set_meals(mt, ma)
    rs = np.random.RandomState(2)
    sample = np.unique(np.concatenate([np.arange(0, 90), np.arange(n - 90, n), rs.randint(0, n, 140)]))[:300]
    sidx = torch.as_tensor(sample, device=e.device)
    z = e.philox_normals(1 + 10 * (2 + K * st // 150), draw0=0, episode=1)[:, sidx].cpu().numpy()
    cho = _dense_cho(mt, ma, sample, K * st)
    b0 = torch.as_tensor(_basal(pid), device=e.device, dtype=torch.float32)
    g = torch.Generator(device=e.device); g.manual_seed(5)
    pool = [(b0 * 2.0 * torch.rand(n, generator=g, device=e.device, dtype=torch.float32)).contiguous() for _ in range(8)]
    pool_s = [p[sidx].double().cpu().numpy() for p in pool]
    orc = O.OracleEnv(pid[sample], sensor="Dexcom", normals=z, integrator="split_adaptive", n_sub=4)
    o0, r0 = e.reset(), orc.reset()
    assert np.abs(o0[sidx].double().cpu().numpy() - r0["cgm"]).max() < 1e-3
    worst_bg = worst_cgm = 0.0
    for k in range(K):
        e.step(pool[k % 8])
        r = orc.step(pool_s[k % 8], None, cho[k * st:(k + 1) * st])
        if k % 8 == 7 or k == K - 1:
            worst_bg = max(worst_bg, np.abs(e.bg[sidx].double().cpu().numpy() - r["bg"]).max())
            worst_cgm = max(worst_cgm, np.abs(e.cgm[sidx].double().cpu().numpy() - r["cgm"]).max())
    print("config 3: fp32 vs fp64 oracle over 24 h: BG %.3e CGM %.3e mg/dL" % (worst_bg, worst_cgm))
    assert worst_bg < 0.05 and worst_cgm < 0.05, (worst_bg, worst_cgm)
    assert e.sync() == 0 and bool(torch.isfinite(e.bg).all()) and int(e.t.min()) == K * st == int(e.t.max())


@pytest.mark.parametrize("dtype_name", ["f64", "f32"])
def test_config5_pid_rollout_7_days_sampled_envs_vs_oracle(dtype_name):
    """BASELINE config 5: in-kernel PID closed loop (gains of the reference's tests/test_pid_controller.py:18: P 1e-3,
    I 1e-5, D 1e-3, target 140), Dexcom, 7 days = 3 360 env.steps in launches of 20 to 480 steps, fp64 and fp32, against
    the oracle's closed loop (same scheme, fp64) on 48 envs replayed with the kernel's own normals and meals.  These gains
    wind up: within days most virtual patients are driven through BG = 0 (the x3 >= 0 clamp, held from then on), some
    swing between 20 and 480 mg/dL.  fp64: the kernel follows the oracle to 1e-6 throughout (measured 3e-9: with x3 held
    exactly and set to -1e-10 where it crosses zero the clamp regime is deterministic); fp32 against the fp64 oracle:
    bounds on the distribution (measured median 2e-3, p90 3e-2, max 0.8 mg/dL)."""
    import torch
    from simglucose_amd import scenario_batch as sb
    from oracle import t1d_oracle as O
    dt = torch.float64 if dtype_name == "f64" else torch.float32
    n, days, st = 4096, 7, 3
    K = days * 1440 // st
    pid = np.arange(n) % 30
    e = _mk(patient=pid, sensor="Dexcom", dtype=dt, noise="philox", seed=55, n_sub=4)
    mt, ma = sb.random_meal_tables(n, days=days, start_minute_of_day=0, seed=21, device=e.device, dtype=dt)
    e.set_meals(mt, ma)
    sample = np.unique(np.concatenate([np.arange(0, 30), np.arange(n - 18, n)]))
    sidx = torch.as_tensor(sample, device=e.device)
    z = e.philox_normals(1 + 10 * (2 + K * st // 150), draw0=0, episode=1)[:, sidx].cpu().numpy()
    cho = _dense_cho(mt, ma, sample, K * st)
    orc = O.OracleEnv(pid[sample], sensor="Dexcom", normals=z, integrator="split_adaptive", n_sub=4)
    e.reset()
    r = orc.reset()
    P, I, D, target = 0.001, 0.00001, 0.001, 140.0
    obs = r["cgm"].copy(); integ = np.zeros(len(sample)); prev = np.zeros(len(sample))
    tr = e.new_trace(K, columns=("bg",))
    state = None
    done = 0
    for chunk in (20, 100, 480, 480, 480, 480, 480, 480, 360):
        state = e.rollout_pid(chunk, P, I, D, target, pid_state=state, trace=tr)
        done += chunk
    assert done == K and tr["row"] == K + 1
    ref = np.empty((K, len(sample)))
    for k in range(K):
        u = P * (obs - target) + I * integ + D * (obs - prev) / st            # pid_ctrller.py:17-36
        prev = obs.copy(); integ = integ + (obs - target) * st
        o = orc.step(u, None, cho[k * st:(k + 1) * st])
        obs = o["cgm"]; ref[k] = o["bg"]
    got = tr["bg"][1:, sidx].double().cpu().numpy()
    err = np.abs(got - ref).max(0)
    print("config 5 %s: |BG - oracle| over 7 days: median %.2e p90 %.2e max %.2e; BG range %.1f .. %.1f" % (
        dtype_name, np.median(err), np.percentile(err, 90), err.max(), ref.min(), ref.max()))
    if dtype_name == "f64":
        assert np.median(err) < 1e-8 and err.max() < 1e-6, (np.median(err), err.max())
    else:
        assert np.median(err) < 0.02 and np.percentile(err, 90) < 0.3 and err.max() < 5.0, (np.median(err), np.percentile(err, 90), err.max())
    assert e.sync() == 0 and bool(torch.isfinite(e.bg).all()) and int(e.t.min()) == K * st == int(e.t.max())


def test_pid_rollout_1mi_envs_launch_per_step_sampled_envs_vs_oracle():
    """The in-kernel PID closed loop at 1 048 576 fp64 envs, where the library's defaults run it as one launch of the persistent
    multi-minute kernel per step with the controller fused in (t1d.h "rollout_launches"): 8 h of Dexcom steps in calls of 1,
    39 and 120 steps, milder gains than config 5's; 200 envs sampled across the batch (first and last workgroups, every
    patient) follow the oracle's closed loop -- same scheme, the kernel's own normals and meals -- to 1e-8 mg/dL, and the
    controller state to rounding."""
    import torch
    from simglucose_amd import scenario_batch as sb
    from oracle import t1d_oracle as O
    n, st, K = 1 << 20, 3, 160
    pid = np.arange(n) % 30
    e = _mk(patient=pid, sensor="Dexcom", noise="philox", seed=56, n_sub=4)
    mt, ma = sb.random_meal_tables(n, days=1, start_minute_of_day=6 * 60, seed=22, device=e.device)
    e.set_meals(mt, ma)
    rs = np.random.RandomState(3)
    sample = np.unique(np.concatenate([np.arange(0, 70), np.arange(n - 70, n), rs.randint(0, n, 60)]))[:200]
    sidx = torch.as_tensor(sample, device=e.device)
    z = e.philox_normals(1 + 10 * (2 + K * st // 150), draw0=0, episode=1)[:, sidx].cpu().numpy()
    cho = _dense_cho(mt, ma, sample, K * st)
    orc = O.OracleEnv(pid[sample], sensor="Dexcom", normals=z, integrator="split_adaptive", n_sub=4)
    e.reset()
    r = orc.reset()
    P, I, D, target = 1.5e-4, 4e-7, 5e-4, 140.0
    obs = r["cgm"].copy(); integ = np.zeros(len(sample)); prev = np.zeros(len(sample))
    tr = e.new_trace(K, columns=("bg", "cgm"))
    state = None
    for chunk in (1, 39, 120):
        state = e.rollout_pid(chunk, P, I, D, target, pid_state=state, trace=tr)
    ref_bg, ref_cgm = np.empty((K, len(sample))), np.empty((K, len(sample)))
    for k in range(K):
        u = P * (obs - target) + I * integ + D * (obs - prev) / st            # pid_ctrller.py:17-36
        prev = obs.copy(); integ = integ + (obs - target) * st
        o = orc.step(u, None, cho[k * st:(k + 1) * st])
        obs = o["cgm"]; ref_bg[k] = o["bg"]; ref_cgm[k] = o["cgm"]
    assert np.abs(tr["bg"][1:, sidx].cpu().numpy() - ref_bg).max() < 1e-8
    assert np.abs(tr["cgm"][1:, sidx].cpu().numpy() - ref_cgm).max() < 1e-8
    assert np.abs(state["integ"][sidx].cpu().numpy() - integ).max() < 1e-6 * max(1.0, np.abs(integ).max())
    assert np.abs(state["prev"][sidx].cpu().numpy() - prev).max() < 1e-8
    assert e.sync() == 0 and bool(torch.isfinite(e.bg).all()) and int(e.t.min()) == K * st == int(e.t.max())


def test_config5_262144_envs_7_days_properties():
    """Config 5 at its stated size (262 144 envs, fp32, 7 days of in-kernel PID closed loop in launches of 480 steps):
    size-independent properties -- every env reaches minute 10 080 with a finite state and no status bit; identical
    envs (same patient, meal plan and noise stream is impossible, so: the run is reproducible bit for bit)."""
    import torch
    from simglucose_amd import scenario_batch as sb
    n, days, st = 262144, 7, 3
    K = days * 1440 // st
    pid = np.arange(n) % 30
    outs = []
    for rep in range(2):
        e = _mk(patient=pid, sensor="Dexcom", dtype=torch.float32, noise="philox", seed=9, n_sub=4, extra_outputs=False)
        mt, ma = sb.random_meal_tables(n, days=days, seed=3, device=e.device, dtype=torch.float32)
        e.set_meals(mt, ma); e.reset()
        state = None
        stats = {"n_low": torch.zeros(n, dtype=torch.int32, device=e.device), "n_high": torch.zeros(n, dtype=torch.int32, device=e.device)}
        for _ in range(K // 480):
            state = e.rollout_pid(480, 0.001, 0.00001, 0.001, 140.0, pid_state=state, stats=stats)
        assert e.sync() == 0
        assert bool(torch.isfinite(e.x).all()) and bool(torch.isfinite(e.bg).all())
        assert int(e.t.min()) == K * st == int(e.t.max()) == 10080
        assert int((stats["n_low"] + stats["n_high"]).max()) <= K
        outs.append((e.x.clone(), e.cgm.clone(), stats["n_low"].clone()))
        if rep == 0:
            del e
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])


@pytest.mark.parametrize("math,tol", [(0, 1e-12), (1, 1e-9)])
def test_rhs_known_answers_on_the_gpu(golden, math, tol):
    """Fixture G1 (T1DPatient.model on ~6 000 points recorded from the reference: both Dbar branches, x3 on both sides of
    ke2, EGP < 0, negative states) through t1d_model_rhs: the reference's own arithmetic to 1e-12 relative, the step
    kernels' fast arithmetic (exp-based gastric emptying, Newton-refined reciprocals) to 1e-9."""
    import torch
    g = golden("g1_rhs.npz")
    e = _mk(patient="adult#001", n_envs=1, sensor="Navigator")
    out = e.model_rhs(g["x"].T, g["patient_idx"], g["cho"], g["insulin"], g["last_qsto"], g["last_foodtaken"], math=math)
    got, ref = out.cpu().numpy().T, g["dxdt"]
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-6)
    print("G1 on the GPU, math %d: max relative error %.2e over %d points" % (math, rel.max(), len(ref)))
    assert rel.max() < tol, rel.max()
    # the host adapter's static T1DPatient.model (reference signature) goes the same way
    from simglucose_amd.patient.t1dpatient import T1DPatient, Action
    import pandas as pd
    from simglucose_amd import params
    row = pd.read_csv(params.PATIENT_PARA_FILE).iloc[int(g["patient_idx"][0])]
    d = T1DPatient.model(0, g["x"][0], Action(CHO=float(g["cho"][0]), insulin=float(g["insulin"][0])), row,
                         float(g["last_qsto"][0]), float(g["last_foodtaken"][0]))
    assert np.abs(d - ref[0]).max() <= 1e-12 * max(1.0, np.abs(ref[0]).max())


def test_random_init_bg_device_draw_replays_through_the_oracle():
    """t1d_reset(random_init_bg = 1) draws x3, x4, x12 ~ N(mu, 0.1 mu) from Philox draws -3..-1 of the env's stream:
    fetched through t1d_philox_normals they reproduce the initial states exactly, and the oracle reset from those states
    (with the same noise normals) gives the same first observation; the draws have the reference's moments
    (t1dpatient.py:256-270: mean mu, variance 0.1 mu)."""
    import torch
    from oracle import t1d_oracle as O
    n = 8192
    pid = np.arange(n) % 30
    e = _mk(patient=pid, sensor="Dexcom", noise="philox", seed=17, env_offset=12345, random_init_bg=True)
    obs = e.reset().cpu().numpy()
    z3 = e.philox_normals(3, draw0=-3, episode=1).cpu().numpy()
    zn = e.philox_normals(21, draw0=0, episode=1).cpu().numpy()
    names, tab = O.patient_table()
    x0 = tab[pid, :13].T.copy()
    for row, k in enumerate((3, 4, 12)):
        x0[k] = x0[k] + np.sqrt(0.1 * x0[k]) * z3[row]
    x = e.x.cpu().numpy()
    assert np.abs(x - x0).max() < 1e-10
    orc = O.OracleEnv(pid, sensor="Dexcom", normals=zn, integrator="split_adaptive", n_sub=4)
    r = orc.reset(x0=x0)
    assert np.abs(obs - r["cgm"]).max() < 1e-9 and np.abs(e.bg.cpu().numpy() - r["bg"]).max() < 1e-10
    for k in (3, 4, 12):
        mu = tab[pid, k]
        zz = (x[k] - mu) / np.sqrt(0.1 * mu)
        assert abs(zz.mean()) < 0.04 and abs(zz.std() - 1.0) < 0.04
    # a second episode draws anew; another shard of the same global batch reproduces its slice
    e.reset()
    assert not np.allclose(e.x.cpu().numpy()[3], x[3])
    e2 = _mk(patient=pid[100:], sensor="Dexcom", noise="philox", seed=17, env_offset=12345 + 100, random_init_bg=True)
    e2.reset()
    assert np.array_equal(e2.x.cpu().numpy(), x[:, 100:])
    assert e.sync() == 0 and e2.sync() == 0


def test_unknown_batch_flags_are_rejected():
    """t1d_batch.flags: anything outside the documented bits comes back as T1D_E_INVALID instead of running with frozen
    patients or noise-free sensors (the tuning bits exist only in -DT1D_AB_FLAGS builds)."""
    import ctypes as C
    import torch
    e = _mk(patient="adult#001", n_envs=64, sensor="Navigator")
    e.reset()
    a = torch.full((64,), 0.01, dtype=torch.float64, device=e.device)
    e.step(a)
    for bad in (0x1, 0x100, 0x200, 0x400, 0x800, 0x8, 1 << 20):
        e._b.flags = bad
        assert e._L.t1d_step(e._ctx, C.byref(e._b), 1, 4, None) == -1 and b"flags" in e._L.t1d_last_error(), hex(bad)
        assert e._L.t1d_reset(e._ctx, C.byref(e._b), None, 0, None) == -1
    e._b.flags = e._flags0
    e.step(a)
    assert e.sync() == 0 and int(e.t[0]) == 2


def test_batched_gym_env_shards_equal_slices_of_one_batch():
    """Two shards of a gym batch (env_offset 0 and n/2, as two ranks of a multi-GPU job would build them) hold what the
    two halves of one n-env instance hold: start hours, meal tables, initial glucose draws, first observations and the
    observations after a few steps -- every per-env stream is indexed by the GLOBAL env id."""
    import torch
    from simglucose_amd.envs import BatchedGymT1DSimEnv
    n = 512
    names = ["adolescent#001", "adult#004", "child#007", "adult#009"]
    pname = [names[i % 4] for i in range(n)]
    whole = BatchedGymT1DSimEnv(n, patient_name=pname, seed=11)
    parts = [BatchedGymT1DSimEnv(n // 2, patient_name=pname[lo:lo + n // 2], seed=11, env_offset=lo) for lo in (0, n // 2)]
    o = whole.reset()
    op = [p.reset() for p in parts]
    cat = lambda f: torch.cat([f(p) for p in parts], dim=-1)
    assert torch.equal(whole.start_hour, cat(lambda p: p.start_hour))
    assert len(torch.unique(whole.start_hour)) > 12 and int(whole.start_hour.min()) >= 0 and int(whole.start_hour.max()) <= 23
    assert torch.equal(whole.env.meal_time, cat(lambda p: p.env.meal_time)) and torch.equal(whole.env.meal_amt, cat(lambda p: p.env.meal_amt))
    assert torch.equal(whole.env.x, cat(lambda p: p.env.x)) and torch.equal(o, torch.cat(op))
    a = torch.full((n,), 0.012, dtype=torch.float64, device=o.device)
    for _ in range(5):
        ow = whole.step(a)[0]
        ops = [p.step(a[:n // 2])[0] for p in parts]
    assert torch.equal(ow, torch.cat(ops))
    # a second episode draws new hours, again the same on both sides
    whole.reset(); [p.reset() for p in parts]
    assert torch.equal(whole.start_hour, cat(lambda p: p.start_hour))
