"""GPU tests of the pieces around the RK4 core: meal tables, Philox noise replay, masked reset,
the in-kernel PID roll-out, fp32, checkpointing and size-independent properties at full batch."""
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


@pytest.mark.parametrize("sensor", ["Dexcom", "Navigator"])
def test_meal_table_equals_dense_cho_and_oracle(sensor):
    """Per-env meal tables (cursor + next_meal state) give exactly what dense per-minute CHO gives,
    including two meals in consecutive minutes, a meal at minute 0, ragged tables and an env with no
    meal; checked against the oracle too."""
    import torch
    from simglucose_amd import scenario_batch as sb
    from oracle import t1d_oracle as O
    n = 70
    pid = np.arange(n) % 30
    rs = np.random.RandomState(5)
    lists = []
    for i in range(n):
        k = int(rs.randint(0, 6)) if i != 3 else 0
        mins = sorted(set(int(m) for m in rs.randint(0, 300, k)))
        l = [(m, float(rs.randint(5, 90))) for m in mins]
        if i == 0:
            l = [(0, 40.0), (1, 20.0), (2, 10.0), (150, 75.0)]
        lists.append(l)
    z = rs.randn(40, n)
    st = int(O.sensor_row(sensor)[5])
    nstep = 300 // st
    dense = np.zeros((300, n))
    for i, l in enumerate(lists):
        for m, g in l:
            dense[m, i] = g
    envs = []
    for use_table in (True, False):
        e = _mk(patient=pid, sensor=sensor, noise="host", normals=z, n_sub=4)
        if use_table:
            mt, ma = sb.tables_from_minute_lists(lists, device=e.device)
            e.set_meals(mt, ma)
        e.reset()
        envs.append(e)
    orc = O.OracleEnv(pid, sensor=sensor, normals=z, integrator="split_adaptive", n_sub=4)     # the default integrator at even n_sub
    orc.reset()
    b = _basal(pid)
    for k in range(nstep):
        a = b * (0.6 + 0.8 * ((k * 7) % 5) / 4.0)
        o1 = envs[0].step(a)
        bg1, cgm1, meal1 = envs[0].bg.clone(), envs[0].cgm.clone(), envs[0].meal.clone()
        envs[1].step(a, cho=dense[k * st:(k + 1) * st])
        r = orc.step(a, None, dense[k * st:(k + 1) * st])
        assert torch.equal(bg1, envs[1].bg) and torch.equal(cgm1, envs[1].cgm) and torch.equal(meal1, envs[1].meal), k
        assert np.abs(bg1.cpu().numpy() - r["bg"]).max() < 1e-8
        assert np.abs(meal1.cpu().numpy() - r["meal"]).max() < 1e-12
    assert envs[0].sync() == 0 and envs[1].sync() == 0


def test_philox_noise_replays_through_host_normals_and_oracle():
    """Philox mode: the normals the kernel draws (exported by t1d_philox_normals) fed back as host
    normals reproduce the run bit for bit, the oracle agrees, and the draws look N(0,1)."""
    import torch
    from oracle import t1d_oracle as O
    n = 4096
    pid = (np.arange(n) // 64) % 30
    e1 = _mk(patient=pid, sensor="Dexcom", noise="philox", seed=99, env_offset=1000, n_sub=2)
    z = e1.philox_normals(31, draw0=0, episode=1)
    zz = z.cpu().numpy()
    assert abs(zz.mean()) < 0.02 and abs(zz.std() - 1.0) < 0.02
    from scipy import stats
    assert stats.kstest(zz[:, ::7].ravel(), "norm").pvalue > 1e-4
    assert abs(np.corrcoef(zz[0], zz[1])[0, 1]) < 0.06
    e2 = _mk(patient=pid, sensor="Dexcom", noise="host", normals=z, n_sub=2)
    orc = O.OracleEnv(pid, sensor="Dexcom", normals=zz, integrator="split_adaptive", n_sub=2)
    o1, o2, r = e1.reset().clone(), e2.reset().clone(), orc.reset()
    assert torch.equal(o1, o2)
    assert np.abs(o1.cpu().numpy() - r["cgm"]).max() < 1e-9
    b = _basal(pid)
    for k in range(120):           # 360 min: three noise blocks
        e1.step(b); e2.step(b)
        rr = orc.step(b)
        assert torch.equal(e1.cgm, e2.cgm), k
        assert np.abs(e1.cgm.cpu().numpy() - rr["cgm"]).max() < 1e-8
    # a different env_offset or seed gives different streams; the same gives the same
    e3 = _mk(patient=pid, sensor="Dexcom", noise="philox", seed=99, env_offset=1000 + 64, n_sub=2)
    z3 = e3.philox_normals(3).cpu().numpy()
    assert np.array_equal(z3[:, :-64], zz[:3, 64:])
    assert e1.sync() == 0 and e2.sync() == 0


def test_masked_reset_and_episode_streams():
    import torch
    n = 256
    e = _mk(patient="adult#001", n_envs=n, sensor="Dexcom", noise="philox", seed=5)
    o0 = e.reset().clone()
    b = float(_basal(np.array([10]))[0])
    for _ in range(20):
        e.step(torch.full((n,), 3 * b, dtype=torch.float64, device=e.device))
    snap = {k: getattr(e, k).clone() for k in ("x", "t", "cgm", "cgm0", "last_cgm", "prev_risk", "episode", "pts", "ar_e")}
    mask = torch.zeros(n, dtype=torch.uint8); mask[::2] = 1
    o1 = e.reset(mask=mask).clone()
    odd, even = slice(1, None, 2), slice(0, None, 2)
    for k, v in snap.items():        # unmasked envs untouched
        assert torch.equal(getattr(e, k)[..., odd], v[..., odd]), k
    assert bool((e.t[even] == 0).all()) and bool((e.episode[even] == 2).all()) and bool((e.episode[odd] == 1).all())
    assert torch.allclose(e.bg[even], torch.full_like(e.bg[even], 138.56), atol=1e-9)
    assert not torch.equal(o1[even], o0[even])          # new episode -> new noise stream
    assert e.sync() == 0


def test_rollout_pid_matches_step_by_step_and_oracle(golden):
    """t1d_rollout_pid (K closed-loop steps in one launch) == K x (host PID + t1d_step), and both
    follow the reference's PID golden trace G10 (adult#001, Dexcom seed 5, RandomScenario seed 9)."""
    import torch
    from simglucose_amd import scenario_batch as sb
    from oracle import t1d_oracle as O
    g = golden("g10_pid_actions.npz")
    z = g["randn"]
    n, K = 64, 480
    cho = O.random_scenario_cho(9, 0, K * 3)
    lst = [(int(m), float(cho[m])) for m in np.nonzero(cho)[0]]
    envs = []
    for _ in range(2):
        e = _mk(patient="adult#001", n_envs=n, sensor="Dexcom", noise="host", normals=np.repeat(z[:, None], n, 1), n_sub=4)
        mt, ma = sb.tables_from_minute_lists([lst] * n, device=e.device)
        e.set_meals(mt, ma)
        e.reset()
        envs.append(e)
    ea, eb = envs
    P, I, D, target = 0.001, 0.00001, 0.001, 140.0
    # (a) step by step with the PID on the host (pid_ctrller.py:17-36)
    integ = torch.zeros(n, dtype=torch.float64, device=ea.device); prev = torch.zeros_like(integ)
    obs = ea.cgm.clone()
    acts = []
    min_bg, max_bg, n_low, n_high = 1e9, 0.0, 0, 0
    for k in range(K):
        u = P * (obs - target) + I * integ + D * (obs - prev) / 3.0
        prev = obs.clone(); integ = integ + (obs - target) * 3.0
        acts.append(float(u[0]))
        ea.step(u, torch.zeros_like(u))
        obs = ea.cgm.clone()
        bg0 = float(ea.bg[0])
        min_bg, max_bg = min(min_bg, bg0), max(max_bg, bg0)
        n_low += bg0 < 70; n_high += bg0 > 180
    assert np.abs(np.array(acts) - g["actions"][:, 0]).max() < 1e-6          # same controls as the reference run
    # (b) in-kernel, in chunks of different lengths
    st = None
    stats = {"sum_risk": torch.zeros(n, dtype=torch.float64, device=eb.device),
             "min_bg": torch.full((n,), 1e9, dtype=torch.float64, device=eb.device),
             "max_bg": torch.zeros(n, dtype=torch.float64, device=eb.device),
             "n_low": torch.zeros(n, dtype=torch.int32, device=eb.device),
             "n_high": torch.zeros(n, dtype=torch.int32, device=eb.device)}
    for chunk in (1, 7, 100, 372):
        st = eb.rollout_pid(chunk, P, I, D, target, pid_state=st, stats=stats)
    for k in ("x", "t", "cgm", "bg", "last_cgm", "prev_risk", "reward", "planned"):
        assert torch.allclose(getattr(ea, k).double(), getattr(eb, k).double(), rtol=0, atol=1e-9), k
    assert torch.allclose(st["integ"], integ, atol=1e-6) and torch.allclose(st["prev"], prev, atol=1e-9)
    assert abs(float(stats["min_bg"][0]) - min_bg) < 1e-9 and abs(float(stats["max_bg"][0]) - max_bg) < 1e-9
    assert int(stats["n_low"][0]) == n_low and int(stats["n_high"][0]) == n_high
    assert bool((stats["n_low"] == stats["n_low"][0]).all())
    assert ea.sync() == 0 and eb.sync() == 0


def test_rollout_bb_config1_matches_reference_host_loop_and_oracle(golden):
    """BASELINE config 1 on the device: adult#001 + BBController + RandomScenario(seed 1), Dexcom seed 1, 24 h
    (fixture G6, recorded from the reference).  t1d_rollout_bb (K steps per launch) must equal K x (host
    BBController + t1d_step), follow the oracle's restatement of the same integrator to 1e-8, and stay
    within 1e-3 mg/dL of the reference's SciPy trace; other patients (heterogeneous batch) against the oracle."""
    import csv, os, torch
    from simglucose_amd import scenario_batch as sb, params
    from oracle import t1d_oracle as O
    K, st = 480, 3
    with open(os.path.join(os.path.dirname(__file__), "golden", "g6_config1_adult001_bb.csv"), newline="") as f:
        rows = list(csv.DictReader(f))
    ref_bg = np.array([float(r["BG"]) for r in rows]); ref_cgm = np.array([float(r["CGM"]) for r in rows])
    acts = golden("g6_config1_actions.npz")["actions"]
    names = ["adult#001", "adolescent#003", "child#008", "adult#010"]
    z = np.random.RandomState(1).randn(1 + 10 * (2 + (K * st) // 150))
    cho = O.random_scenario_cho(1, 0, K * st)
    lst = [(int(m), float(cho[m])) for m in np.nonzero(cho)[0]]
    n = len(names) * 16
    pname = [names[i // 16] for i in range(n)]
    envs = []
    for _ in range(2):
        e = _mk(patient=pname, sensor="Dexcom", noise="host", normals=np.repeat(z[:, None], n, 1), n_sub=4)
        mt, ma = sb.tables_from_minute_lists([lst] * n, device=e.device)
        e.set_meals(mt, ma)
        e.reset()
        envs.append(e)
    ea, eb = envs
    # (a) host BBController (basal_bolus_ctrller.py:34-80) around t1d_step
    c = ea.bb_constants()
    obs, meal = ea.cgm.clone(), torch.zeros(n, dtype=torch.float64, device=ea.device)
    bg_a, cgm_a, act_a = [], [], []
    for k in range(K):
        bolus = torch.where(meal > 0, ((meal * st) / c["cr"] + (obs > 150) * (obs - 140.0) / c["cf"]) / st, torch.zeros_like(meal))
        act_a.append((float(c["basal"][0]), float(bolus[0])))
        ea.step(c["basal"], bolus)
        obs, meal = ea.cgm.clone(), ea.meal.clone()
        bg_a.append(ea.bg.clone()); cgm_a.append(ea.cgm.clone())
    assert np.abs(np.array(act_a) - acts).max() < 1e-6                       # same controls as the reference run
    bg_a = torch.stack(bg_a).cpu().numpy(); cgm_a = torch.stack(cgm_a).cpu().numpy()
    assert np.abs(bg_a[:, 0] - ref_bg[1:]).max() < 1e-3 and np.abs(cgm_a[:, 0] - ref_cgm[1:]).max() < 1e-3
    # (b) the same in-kernel, in chunks of different lengths
    stt = None
    stats = {"n_high": torch.zeros(n, dtype=torch.int32, device=eb.device),
             "max_bg": torch.zeros(n, dtype=torch.float64, device=eb.device)}
    tr = eb.new_trace(K)
    for chunk in (1, 9, 170, 300):
        stt = eb.rollout_bb(chunk, bb_state=stt, stats=stats, trace=tr)
    for k in ("x", "t", "cgm", "bg", "last_cgm", "prev_risk", "reward", "planned", "meal", "insulin"):
        assert torch.allclose(getattr(ea, k).double(), getattr(eb, k).double(), rtol=0, atol=1e-9), k
    assert torch.allclose(stt["prev_meal"], meal, atol=1e-12)
    assert np.array_equal(stats["n_high"].cpu().numpy(), (bg_a > 180).sum(0))
    assert np.abs(stats["max_bg"].cpu().numpy() - bg_a.max(0)).max() < 1e-9
    # (b') the device-resident history as the reference's show_history() table / per-patient CSV
    import pandas as pd
    from datetime import datetime
    from simglucose_amd.analysis import report
    df = report.history_frame(tr, 0, datetime(2018, 1, 1, 0, 0, 0), st)
    ref = pd.read_csv(os.path.join(os.path.dirname(__file__), "golden", "g6_config1_adult001_bb.csv"), index_col="Time", parse_dates=True)
    assert list(df.columns) == list(ref.columns) and df.index.name == "Time" and len(df) == len(ref) == K + 1
    assert (df.index == ref.index).all()
    for col, tol in (("BG", 1e-3), ("CGM", 1e-3), ("LBGI", 2e-3), ("HBGI", 2e-3), ("Risk", 2e-3), ("CHO", 1e-12), ("insulin", 1e-9)):
        assert np.nanmax(np.abs(df[col].values - ref[col].values)) < tol, col
        assert np.array_equal(np.isnan(df[col].values), np.isnan(ref[col].values)), col
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        files = report.save_histories(tr, [0, 16], ["adult#001", names[1]], tmp, datetime(2018, 1, 1), st)
        back = pd.read_csv(files[0], index_col="Time", parse_dates=True)
        assert np.nanmax(np.abs(back["BG"].values - df["BG"].values)) < 1e-12 and os.path.basename(files[1]) == names[1] + ".csv"
    # (c) the oracle's closed loop with the same scheme, every patient of the batch
    for j, nm in enumerate(names):
        hist, a = O.closed_loop(nm, "Dexcom", 1, 1, K, lambda cgm, info, nm=nm: O.bb_policy(nm, info["meal"], cgm, info["sample_time"]),
                                integrator="split_adaptive", n_sub=4)
        assert np.abs(bg_a[:, 16 * j] - hist["BG"][1:]).max() < 1e-8, nm
        assert np.abs(cgm_a[:, 16 * j] - hist["CGM"][1:]).max() < 1e-8, nm
    assert ea.sync() == 0 and eb.sync() == 0


@pytest.mark.parametrize("n_sub", [2, 6, 8])
def test_split_integrator_other_substep_counts_and_table_rebuild(n_sub):
    """The split integrator at n_sub = 2, 6, 8 (tables rebuilt inside one ctx when n_sub changes) against the
    oracle's restatement with scipy-built tables; odd n_sub falls back to classical RK4 under "auto" and is an
    error when the split scheme is demanded."""
    import torch
    from simglucose_amd import _lib
    from oracle import t1d_oracle as O
    rs = np.random.RandomState(n_sub)
    n = 90
    pid = np.arange(n) % 30
    z = rs.randn(30, n)
    e = _mk(patient=pid, sensor="Navigator", noise="host", normals=z, n_sub=4)
    e.reset()
    b = _basal(pid)
    cho = np.zeros((60, n)); cho[5] = 60.0; cho[20, ::2] = 25.0
    e.step(torch.as_tensor(b, device=e.device), cho=cho[0:1])                 # builds the n_sub = 4 tables first
    orc4 = O.OracleEnv(pid, sensor="Navigator", normals=z, integrator="split_adaptive", n_sub=4)
    orc4.reset(); r = orc4.step(b, None, cho[0:1])
    assert np.abs(e.bg.cpu().numpy() - r["bg"]).max() < 1e-8
    e.n_sub = n_sub                                                            # forces a rebuild of the tables
    orc = O.OracleEnv(pid, sensor="Navigator", normals=z, integrator="split_adaptive", n_sub=n_sub)
    orc.reset()
    for k in ("x", "planned", "last_qsto", "last_food", "last_cgm", "ar_e", "pts", "prev_cgm"):
        getattr(orc, k)[...] = getattr(orc4, k)
    orc.was_eating[:] = orc4.was_eating; orc.t[:] = orc4.t; orc.n_samples[:] = orc4.n_samples; orc.n_draws[:] = orc4.n_draws
    for k in range(1, 60):
        a = b * (0.5 + (k % 5) * 0.3)
        e.step(torch.as_tensor(a, device=e.device), cho=cho[k:k + 1])
        r = orc.step(a, None, cho[k:k + 1])
        assert np.abs(e.bg.cpu().numpy() - r["bg"]).max() < 1e-8, k
        assert np.abs(e.cgm.cpu().numpy() - r["cgm"]).max() < 1e-8, k
    assert e.sync() == 0
    e.n_sub = 3
    e.step(torch.as_tensor(b, device=e.device), cho=cho[0:1])                 # auto: classical RK4
    e.set_option("integrator", 1)
    with pytest.raises(_lib.T1DError):
        e.step(torch.as_tensor(b, device=e.device), cho=cho[0:1])
    e.sync(raise_on_status=False)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_outcome_statistics_match_reference_report(golden, dtype):
    """t1d_outcome_stats on fixture G11: the reference's percent_stats and CVGA_analysis outputs
    exactly (percentiles by radix selection == np.percentile), zones, and the chunked risk trace against the
    oracle's restatement of risk_index_trace."""
    import torch
    from simglucose_amd.analysis import report
    from oracle import t1d_oracle as O
    g = golden("g11_report.npz")
    dt = torch.float64 if dtype == "f64" else torch.float32
    bg = torch.as_tensor(g["bg"], dtype=dt, device="cuda:0").contiguous()
    ref_bg = bg.double().cpu().numpy()                       # what the kernel sees (fp32 run: rounded inputs)
    r = report.outcome_stats(bg)
    want = O.report_percent_stats(ref_bg)
    assert np.array_equal(r["counts"].cpu().numpy(), np.rint(want * len(ref_bg) / 100.0).astype(np.int64))
    assert np.abs(r["percent"].cpu().numpy() - want).max() < 1e-12
    mn, mx, frac, zone = O.report_cvga(ref_bg)
    tol = 1e-12 if dtype == "f64" else 2e-5
    assert np.abs(r["bg_min"].double().cpu().numpy() - mn).max() < tol and np.abs(r["bg_max"].double().cpu().numpy() - mx).max() < tol
    if dtype == "f64":
        assert np.abs(r["percent"].cpu().numpy() - g["percent"]).max() < 1e-12
        assert np.abs(r["bg_min"].cpu().numpy() - g["bg_min"]).max() < 1e-12 and np.abs(r["bg_max"].cpu().numpy() - g["bg_max"]).max() < 1e-12
        assert np.array_equal(r["zone"].cpu().numpy(), zone)
        out = report.CVGA_analysis(bg)
        assert np.abs(np.array(out[2:]) - g["zones"]).max() < 1e-15
    L, H = O.report_risk_index_trace(ref_bg)
    rtol = 1e-10 if dtype == "f64" else 1e-4
    assert np.allclose(r["lbgi"].double().cpu().numpy(), L, rtol=rtol, atol=rtol) and np.allclose(r["hbgi"].double().cpu().numpy(), H, rtol=rtol, atol=rtol)
    # G12: the reference's own 2017 result files -- risk_trace.csv, performance_stats.csv, CVGA_stats.csv -- from the BG
    # columns of its per-patient CSVs (patients whose BG reached 0: chunks without a usable sample are NaN, as upstream)
    g12 = golden("g12_report_2017.npz")
    bg12 = torch.as_tensor(g12["bg"], dtype=dt, device="cuda:0").contiguous()
    r12 = report.outcome_stats(bg12)
    if dtype == "f64":
        for got, want in ((r12["lbgi"], g12["lbgi_trace"]), (r12["hbgi"], g12["hbgi_trace"])):
            got = got.cpu().numpy()
            assert np.array_equal(np.isnan(got), np.isnan(want)) and np.nanmax(np.abs(got - want)) < 1e-10
        cols = list(g12["perf_cols"])
        for row, name in enumerate(report.PERCENT_COLUMNS):
            assert np.abs(r12["percent"][row].cpu().numpy() - g12["perf"][:, cols.index(name)]).max() < 1e-12, name
        assert np.abs(np.array(report.CVGA_analysis(bg12)[2:]) - g12["cvga_zones"]).max() < 1e-15
    # a history written by the roll-out kernel: rows = reset + every step, equal to stepping on the host
    from simglucose_amd import scenario_batch as sb
    n, K = 256, 96
    envs = []
    for _ in range(2):
        e = _mk(patient=np.arange(n) % 30, sensor="Dexcom", dtype=dt, noise="philox", seed=7, n_sub=4)
        mt, ma = sb.random_meal_tables(n, days=1, seed=5, device=e.device, dtype=dt)
        e.set_meals(mt, ma); e.reset()
        envs.append(e)
    tr = {"bg": torch.zeros(K + 1, n, dtype=dt, device="cuda:0"), "cgm": torch.zeros(K + 1, n, dtype=dt, device="cuda:0"), "row": 1}
    tr["bg"][0] = envs[0].bg; tr["cgm"][0] = envs[0].cgm
    st = envs[0].rollout_bb(40, trace=tr)
    envs[0].rollout_bb(K - 40, bb_state=st, trace=tr)
    assert tr["row"] == K + 1
    st2 = None
    host = [envs[1].bg.clone()]
    for k in range(K):
        st2 = envs[1].rollout_bb(1, bb_state=st2)
        host.append(envs[1].bg.clone())
    assert torch.allclose(torch.stack(host).double(), tr["bg"].double(), rtol=0, atol=1e-9 if dtype == "f64" else 1e-3)
    s = report.outcome_stats(tr["bg"])
    assert int(s["counts"][:3].sum(0).min()) == K + 1 == int(s["counts"][:3].sum(0).max())      # the three ranges partition the samples
    assert envs[0].sync() == 0 and envs[1].sync() == 0


@pytest.mark.parametrize("sensor", ["Navigator", "Dexcom"])
def test_custom_patient_table_with_more_than_32_patients(sensor):
    """A caller-supplied table of 44 patients (the 30 of the reference + perturbed copies): the single-minute kernel
    then runs its 64-patient LDS layout, the generic kernels a 44-column propagator table; both against the oracle
    on the same table."""
    import torch
    from simglucose_amd import params
    from oracle import t1d_oracle as O
    names, tab = params.patient_table()
    rs = np.random.RandomState(44)
    extra = tab[rs.randint(0, 30, 14)].copy()
    for c in ("kabs", "kmax", "kp2", "k1", "k2", "m1", "m30", "ka2", "ksc", "p2u", "ki", "Vmx"):
        extra[:, params.P_COL[c]] *= rs.uniform(0.85, 1.15, 14)
    big = np.ascontiguousarray(np.vstack([tab, extra]))
    n = 88
    pid = np.arange(n) % 44
    z = rs.randn(20, n)
    e = _mk(patient=pid, patient_table=big, sensor=sensor, noise="host", normals=z, n_sub=4)
    orc = O.OracleEnv(pid, sensor=sensor, normals=z, integrator="split_adaptive", n_sub=4, ptab_override=big)
    o0, r0 = e.reset(), orc.reset()
    assert np.abs(o0.cpu().numpy() - r0["cgm"]).max() < 1e-9
    st = int(e.minutes_per_step)
    b = big[pid, params.P_COL["u2ss"]] * big[pid, params.P_COL["BW"]] / 6000.0
    for k in range(40):
        cho = np.zeros((st, n))
        if k == 4:
            cho[0] = 55.0
        a = b * (0.4 + 0.4 * (k % 4))
        e.step(torch.as_tensor(a, device=e.device), cho=cho)
        r = orc.step(a, None, cho)
        assert np.abs(e.bg.cpu().numpy() - r["bg"]).max() < 1e-8, k
        assert np.abs(e.cgm.cpu().numpy() - r["cgm"]).max() < 1e-8, k
    assert e.sync() == 0


@pytest.mark.parametrize("sensor,n_sub", [("Navigator", 4), ("Dexcom", 4), ("Navigator", 8), ("Dexcom", 8)])
def test_table_of_64_patients_runs_at_every_substep_count(sensor, n_sub):
    """The largest table the header allows (64 patients) at n_sub = 4 and 8: 133 / 245 propagator rows x 64 columns is
    more than 64 KiB of LDS.  One-minute launches take the persistent kernel's 64-patient layout; steps of several minutes
    take the generic kernel with its dynamic-LDS ceiling raised -- neither may refuse the call (integrator = -1 promises
    the split scheme whenever math and n_sub allow it) and both follow the oracle on the same table."""
    import torch
    from simglucose_amd import params
    from oracle import t1d_oracle as O
    names, tab = params.patient_table()
    rs = np.random.RandomState(64)
    extra = tab[rs.randint(0, 30, 34)].copy()
    for c in ("kabs", "kmax", "kp2", "k1", "k2", "m1", "m30", "ka2", "ksc", "p2u", "ki", "Vmx"):
        extra[:, params.P_COL[c]] *= rs.uniform(0.9, 1.1, 34)
    big = np.ascontiguousarray(np.vstack([tab, extra]))
    n = 128
    pid = np.arange(n) % 64
    z = rs.randn(20, n)
    e = _mk(patient=pid, patient_table=big, sensor=sensor, noise="host", normals=z, n_sub=n_sub)
    orc = O.OracleEnv(pid, sensor=sensor, normals=z, integrator="split_adaptive", n_sub=n_sub, ptab_override=big)
    o0, r0 = e.reset(), orc.reset()
    assert np.abs(o0.cpu().numpy() - r0["cgm"]).max() < 1e-9
    st = int(e.minutes_per_step)
    b = big[pid, params.P_COL["u2ss"]] * big[pid, params.P_COL["BW"]] / 6000.0
    for k in range(30):
        cho = np.zeros((st, n))
        if k == 3:
            cho[0] = 60.0
        a = b * (0.4 + 0.4 * (k % 4))
        e.step(torch.as_tensor(a, device=e.device), cho=cho)
        r = orc.step(a, None, cho)
        assert np.abs(e.bg.cpu().numpy() - r["bg"]).max() < 1e-8, k
        assert np.abs(e.cgm.cpu().numpy() - r["cgm"]).max() < 1e-8, k
    assert e.sync() == 0


@pytest.mark.parametrize("pump", ["Insulet", "Cozmo"])
@pytest.mark.parametrize("sensor", ["Navigator", "Dexcom"])
def test_pump_quantiser_in_kernel_matches_reference(golden, pump, sensor):
    """G3: the reference's InsulinPump.basal/.bolus on 364 amounts (negative, ties at half increments, beyond the
    limits) through the step kernels' quantiser (single-minute and generic kernel): the insulin output is the
    sum of the two quantised rates; which increment a tie rounds to must agree exactly, the value to 1 ulp."""
    import torch
    g = golden("g3_pump.npz")
    amt = g["amount"]
    n = len(amt)
    e = _mk(patient="adult#001", n_envs=n, sensor=sensor, pump=pump, n_sub=4)
    e.reset()
    a = torch.as_tensor(amt, dtype=torch.float64, device=e.device)
    e.step(a, torch.zeros_like(a))
    want_b = g["basal_" + pump] + np.maximum(float(e.pump_row[0]), 0.0)          # + pump.bolus(0)
    got = e.insulin.cpu().numpy()
    ulps = lambda x, y: np.max(np.abs(x - y) / np.spacing(np.maximum(np.abs(y), 1e-300)))
    assert ulps(got, want_b) <= 4, ulps(got, want_b)                             # fast reciprocal + mean over the step
    e.step(torch.zeros_like(a), a)
    want = g["bolus_" + pump] + float(np.clip(0.0, e.pump_row[3], e.pump_row[4]))
    got = e.insulin.cpu().numpy()
    assert ulps(got, want) <= 4, ulps(got, want)
    assert e.sync(raise_on_status=False) in (0, 2)                               # absurd doses may drive a state non-finite


def test_c_abi_argument_errors_are_reported_not_executed():
    """Error behaviour at the C ABI on a live device: bad arguments come back as negative codes with a message
    (the Python layer raises T1DError / ValueError) and leave the env usable."""
    import ctypes as C
    import torch
    from simglucose_amd import _lib
    e = _mk(patient="adult#001", n_envs=130, sensor="Dexcom", n_sub=4)
    e.reset()
    a = torch.full((130,), 0.01, dtype=torch.float64, device=e.device)
    L = e._L
    for minutes, n_sub in ((0, 4), (3, 0), (100001, 4), (3, 5000)):
        assert L.t1d_step(e._ctx, C.byref(e._b), minutes, n_sub, None) == -1 and b"t1d_step" in L.t1d_last_error()
    e._b.basal = None
    assert L.t1d_step(e._ctx, C.byref(e._b), 3, 4, None) == -1 and b"basal" in L.t1d_last_error()
    with pytest.raises(ValueError):
        e.step(a, cho=np.zeros((2, 130)))                       # cho must be [minutes, n]
    with pytest.raises(ValueError):
        e.reset(x0=np.zeros((12, 130)))
    with pytest.raises(_lib.T1DError):
        e.set_option("no_such_option", 1)
    with pytest.raises(_lib.T1DError):
        e.set_option("math", 7)
    bad = _lib.Bb()
    assert L.t1d_rollout_bb(e._ctx, C.byref(e._b), C.byref(bad), 4, 3, 4, None) == -1
    out = _lib.Outcome()
    assert L.t1d_outcome_stats(0, 0, 130, 0, C.c_void_p(e.bg.data_ptr()), C.byref(out), None) == -1
    assert L.t1d_random_meals(0, 1, 0, 130, 0, 0, None, 0, C.c_void_p(e.t.data_ptr()), C.c_void_p(e.bg.data_ptr()), None) == -1
    e.step(a)                                                    # still works
    assert e.sync() == 0 and bool(torch.isfinite(e.bg).all())


def test_random_meal_tables_match_reference_generator_statistics():
    """t1d_random_meals against the reference's RandomScenario.create_scenario as restated (and
    pinned by fixture G9) in the oracle: structure of the tables exactly, distributions per meal window within
    sampling error of 6 000 reference days."""
    import torch
    from simglucose_amd import scenario_batch as sb
    from oracle import t1d_oracle as O
    n, days = 1 << 17, 2
    mt, ma = sb.random_meal_tables(n, days=days, start_minute_of_day=0, seed=11, device="cuda:0")
    t, a = mt.cpu().numpy().astype(np.int64), ma.cpu().numpy()
    assert t.shape == (6 * (days + 1), n)
    used = t != 0x7FFFFFFF
    assert np.all(np.diff(t, axis=0)[used[1:]] > 0)                          # ascending, one meal per minute
    assert np.all(used[:-1] | ~used[1:])                                     # unused entries only at the end
    assert t[used].min() >= 5 * 60 and t[used].max() < days * 1440
    assert np.all(a[~used] == 0) and np.all(a[used] >= 0) and np.all(a[used] == np.round(a[used]))
    # deterministic; shifting env_offset shifts the streams; another seed differs
    mt2, ma2 = sb.random_meal_tables(n, days=days, start_minute_of_day=0, seed=11, device="cuda:0")
    assert torch.equal(mt, mt2) and torch.equal(ma, ma2)
    mt3, _ = sb.random_meal_tables(n - 64, days=days, start_minute_of_day=0, seed=11, device="cuda:0", env_offset=64)
    assert torch.equal(mt3, mt[:, 64:])
    assert not torch.equal(sb.random_meal_tables(n, days=days, seed=12, device="cuda:0")[0], mt)
    # reference sample
    rs = np.random.RandomState(123)
    ref_t, ref_a, ref_days = [], [], 6000
    for _ in range(ref_days):
        tt, aa = O.random_scenario_draw(rs)
        ref_t += list(tt); ref_a += list(aa)
    ref_t, ref_a = np.array(ref_t), np.array(ref_a)
    tod, grams = t[used] % 1440, a[used]
    lb = np.array([5, 9, 10, 14, 16, 20]) * 60; ub = np.array([9, 10, 14, 16, 20, 23]) * 60
    assert abs(used.sum() / (n * days) - len(ref_t) / ref_days) < 5 * np.sqrt(3.75 / ref_days)      # meals per day
    for k in range(6):
        sel, rsel = (tod > lb[k]) & (tod < ub[k]), (ref_t > lb[k]) & (ref_t < ub[k])    # open interval: boundary minutes are ambiguous
        f, rf = sel.sum() / (n * days), rsel.sum() / ref_days
        assert abs(f - rf) < 5 * np.sqrt(rf * (1 - rf) / ref_days) + 1e-3, (k, f, rf)
        for x, rx in ((tod[sel], ref_t[rsel]), (grams[sel], ref_a[rsel])):
            se = rx.std() / np.sqrt(len(rx))
            assert abs(x.mean() - rx.mean()) < 5 * se + 0.05, (k, x.mean(), rx.mean())
            assert abs(x.std() - rx.std()) < 0.06 * rx.std() + 0.05, (k, x.std(), rx.std())
    # a start at 14:00 drops the morning of day 0 and reaches into day `days`
    mt4, _ = sb.random_meal_tables(4096, days=1, start_minute_of_day=14 * 60, seed=3, device="cuda:0")
    t4 = mt4.cpu().numpy().astype(np.int64); u4 = t4 != 0x7FFFFFFF
    assert t4[u4].min() >= 0 and t4[u4].max() < 1440
    assert ((t4[u4] + 14 * 60) // 1440 == 1).any() and ((t4[u4] + 14 * 60) // 1440 == 0).any()
    st = torch.randint(0, 1440, (4096,), dtype=torch.int32, device="cuda:0")
    mt5, _ = sb.random_meal_tables(4096, days=1, start_minute_of_day=st, seed=3, device="cuda:0")
    t5 = mt5.cpu().numpy().astype(np.int64); u5 = t5 != 0x7FFFFFFF
    todd = (t5 + st.cpu().numpy()[None, :].astype(np.int64)) % 1440
    assert todd[u5].min() >= 5 * 60 and todd[u5].max() <= 23 * 60


def test_fp32_tracks_fp64():
    """fp32 variant (BASELINE configs 3/5): stays within 0.05 mg/dL of fp64 over 12 h with meals."""
    import torch
    from simglucose_amd import scenario_batch as sb
    n = 1920
    pid = np.arange(n) % 30
    mt, ma = sb.random_meal_tables(n, days=1, seed=4, device="cuda:0")
    envs = []
    for dt in (torch.float64, torch.float32):
        e = _mk(patient=pid, sensor="Dexcom", dtype=dt, noise="philox", seed=3, n_sub=4)
        e.set_meals(mt, ma.to(dt))
        e.reset()
        envs.append(e)
    b64 = torch.as_tensor(_basal(pid), device="cuda:0")
    worst = 0.0
    for k in range(240):
        envs[0].step(b64); envs[1].step(b64.float())
        worst = max(worst, float((envs[0].bg - envs[1].bg.double()).abs().max()))
    assert worst < 0.05, worst
    assert envs[0].sync() == 0 and envs[1].sync() == 0


def test_fp32_single_minute_kernel_tracks_fp64():
    """The persistent single-minute kernel in fp32 (1-min sensor) against its fp64 instantiation: 6 h with meals."""
    import torch
    from simglucose_amd import scenario_batch as sb
    n = 3000                                  # not a multiple of 64: the last chunk is partly masked
    pid = np.arange(n) % 30
    mt, ma = sb.random_meal_tables(n, days=1, start_minute_of_day=6 * 60, seed=9, device="cuda:0")
    envs = []
    for dt in (torch.float64, torch.float32):
        e = _mk(patient=pid, sensor="Navigator", dtype=dt, noise="philox", seed=3, n_sub=4)
        e.set_meals(mt, ma.to(dt))
        e.reset()
        envs.append(e)
    b64 = torch.as_tensor(_basal(pid), device="cuda:0")
    worst = 0.0
    for k in range(360):
        a = b64 * (0.6 + 0.2 * (k % 5))
        envs[0].step(a); envs[1].step(a.float())
        if k % 20 == 19:
            worst = max(worst, float((envs[0].bg - envs[1].bg.double()).abs().max()))
    assert worst < 0.05, worst
    assert torch.equal(envs[0].t, envs[1].t) and int(envs[0].t[0]) == 360
    assert envs[0].sync() == 0 and envs[1].sync() == 0


@pytest.mark.parametrize("dtype_name", ["f64", "f32"])
def test_deferred_refinement_equals_in_place_refinement(dtype_name):
    """The single-minute kernel with the adaptive scheme's refinement deferred to the end of the launch (default,
    step1d_kernel) against the in-place form (adaptive_gut = 2): the same lanes refine and take the same arithmetic,
    so the states agree to rounding (different instantiations may contract FMAs differently) -- 8 h with meals, with
    the default grid and with a 3-block grid (many chunks per wave, a long list per block), extra outputs on and off.
    Some envs share their meal times so that whole waves are flagged at once."""
    import torch
    from simglucose_amd import scenario_batch as sb
    dt = torch.float64 if dtype_name == "f64" else torch.float32
    n = 64 * 200 - 7                              # the last chunk is partly masked
    pid = np.arange(n) % 30
    mt, ma = sb.random_meal_tables(n, days=1, start_minute_of_day=6 * 60, seed=11, device="cuda:0", dtype=dt)
    mt[:, :640] = mt[:, :1]; ma[:, :640] = ma[:, :1]        # ten waves of envs with one meal plan
    for extra, blocks in ((False, 0), (True, 3)):
        envs = []
        for form in (2, 3):                       # 2: every lane in place; 3: levels 1, 2 set aside whatever the batch size
            e = _mk(patient=pid, sensor="Navigator", dtype=dt, noise="philox", seed=4, n_sub=4, extra_outputs=extra)
            e.set_option("adaptive_gut", form)
            e.set_option("s1_blocks", blocks)
            e.set_meals(mt, ma)
            e.reset()
            envs.append(e)
        b = torch.as_tensor(_basal(pid), device="cuda:0", dtype=dt)
        tol = 1e-9 if dtype_name == "f64" else 2e-3
        for k in range(480 if not extra else 150):
            a = b * (0.5 + 0.25 * (k % 7))
            envs[0].step(a); envs[1].step(a)
            if k % 30 == 29:
                assert float((envs[0].bg - envs[1].bg).abs().max()) < tol, k
        assert torch.equal(envs[0].t, envs[1].t) and torch.equal(envs[0].done, envs[1].done)
        assert float((envs[0].x - envs[1].x).abs().max()) < tol * 100
        assert float((envs[0].cgm - envs[1].cgm).abs().max()) < tol
        assert float((envs[0].reward - envs[1].reward).abs().max()) < tol
        if extra:
            assert float((envs[0].risk - envs[1].risk).abs().max()) < tol
            assert torch.equal(envs[0].meal, envs[1].meal)
        assert envs[0].sync() == 0 and envs[1].sync() == 0


def test_state_dict_roundtrip_and_determinism():
    import torch
    n = 512
    pid = (np.arange(n) // 64) % 30
    e = _mk(patient=pid, sensor="Dexcom", noise="philox", seed=8)
    e.reset()
    b = torch.as_tensor(_basal(pid), device=e.device)
    for _ in range(10):
        e.step(b)
    sd = e.state_dict()
    outs = []
    for _ in range(2):
        e.load_state_dict(sd)
        for _ in range(15):
            e.step(1.5 * b)
        outs.append((e.cgm.clone(), e.x.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_clock_edited_behind_the_wrapper_still_refills_noise_blocks():
    """The wrapper shadows a lock-step clock on the host to skip the noise-block refill pre-kernel.  A caller who moves
    env.t in place (instead of going through load_state_dict) must not get a stale noise block: the edit is noticed
    (version counter of the state tensors) and the envs' own clocks decide again.  Reference run: the same jump through
    state_dict / load_state_dict; the jump lands 4 minutes before a block boundary."""
    import torch
    n = 256
    pid = np.arange(n) % 30
    envs = [_mk(patient=pid, sensor="Navigator", noise="philox", seed=31, n_sub=4) for _ in range(2)]
    b = torch.as_tensor(_basal(pid), device="cuda:0")
    for e in envs:
        e.reset()
        for _ in range(100):
            e.step(b)
    assert envs[0]._clock == 100
    envs[0].t += 45                                   # behind the wrapper's back: sample #150 (a new block) is now 4 steps away
    sd = envs[1].state_dict(); sd["t"] = sd["t"] + 45
    envs[1].load_state_dict(sd)
    pts_before = envs[0].pts[:11].clone()
    for _ in range(12):
        envs[0].step(b); envs[1].step(b)
    assert envs[0]._clock is None
    assert not torch.equal(pts_before, envs[0].pts[:11])              # the block was rebuilt
    assert torch.equal(envs[0].pts, envs[1].pts) and torch.equal(envs[0].cgm, envs[1].cgm) and torch.equal(envs[0].t, envs[1].t)
    assert envs[0].sync() == 0 and envs[1].sync() == 0


@pytest.mark.parametrize("sensor", ["Navigator", "Dexcom"])
def test_dbar_row_and_planned_bit_invariants(sensor):
    """ABI 4: the state keeps Dbar = last_qsto + 1000 last_food beside the two words, and bit 9 of meta says
    planned_meal > 0.  Both hold after every step, through meals (table): that is what lets the one-minute kernels skip
    the three meal words of envs that neither eat nor have a meal planned (they read Dbar instead).  The run with the
    wrapper's shadow clock dropped before every step (the refill pre-kernel checks every env's clock) is bit-identical."""
    import torch
    from simglucose_amd import scenario_batch as sb
    n = 64 * 9 - 3
    pid = np.arange(n) % 30
    mt, ma = sb.random_meal_tables(n, days=1, start_minute_of_day=6 * 60 + 30, seed=17, device="cuda:0")
    envs = []
    for _ in range(2):
        e = _mk(patient=pid, sensor=sensor, noise="philox", seed=6, n_sub=4)
        e.set_meals(mt, ma)
        e.reset()
        envs.append(e)
    b = torch.as_tensor(_basal(pid), device="cuda:0")
    st = int(envs[0].sample_time)
    seen_live = False
    for k in range(240 // st):
        a = b * (0.5 + 0.25 * (k % 5))
        envs[0].step(a)
        envs[1].invalidate_clock()
        envs[1].step(a)
        if k % 7 == 0 or k == 240 // st - 1:
            for e in envs:
                assert float((e.dbar - (e.last_qsto + 1000.0 * e.last_food)).abs().max()) <= 1e-12 * float(e.dbar.abs().max() + 1)
                assert torch.equal((e.meta & 0x200) != 0, e.planned > 0)
            seen_live = seen_live or bool((envs[0].meta & 0x300).ne(0).any())
            for key in ("x", "cgm", "bg", "reward", "planned", "last_qsto", "last_food", "dbar", "meta", "next_meal", "pts", "t"):
                assert torch.equal(getattr(envs[0], key), getattr(envs[1], key)), (k, key)
    assert seen_live and int(envs[0].t[5]) == 240
    assert envs[0].sync() == 0 and envs[1].sync() == 0


def test_checkpoint_format_and_history_after_load():
    """state_dict carries a format number (a checkpoint of another layout is refused, not half-loaded) and loading a
    checkpoint without a CGM history into an env that keeps one re-seeds the ring from the restored observation."""
    import torch
    from simglucose_amd._lib import T1DError
    n = 128
    pid = np.arange(n) % 30
    a = _mk(patient=pid, sensor="Dexcom", noise="philox", seed=2)
    a.reset()
    b0 = torch.as_tensor(_basal(pid), device=a.device)
    for _ in range(5):
        a.step(b0)
    sd = a.state_dict()
    assert sd["format"] == a.STATE_FORMAT
    h = _mk(patient=pid, sensor="Dexcom", noise="philox", seed=2, cgm_history=True)
    h.reset()
    for _ in range(9):
        h.step(1.3 * b0)
    h.load_state_dict(sd)
    w = h.cgm_window()
    assert torch.equal(w[-1], a.cgm) and bool(torch.isnan(w[:-1]).all())
    old = dict(sd); del old["format"]
    with pytest.raises(T1DError):
        h.load_state_dict(old)
    with pytest.raises(ValueError):
        h.model_rhs(torch.zeros(13, 2), [0, 99], [0.0, 0.0], [0.0, 0.0], [0.0, 0.0], [0.0, 0.0])


def test_full_batch_properties_1m_envs():
    """BASELINE.json's full size (1 048 576 envs): size-independent properties.  (1) every env of a
    patient-homogeneous batch with identical inputs carries identical state; (2) shifting the batch
    (env_offset) permutes nothing but the noise; (3) steady state: basal-only adult#001 stays at
    138.56 mg/dL; (4) wave-uniform (SGPR) and LDS parameter paths agree to rounding."""
    import torch
    n = 1 << 20
    e = _mk(patient="adult#001", n_envs=n, sensor="Navigator", noise="philox", seed=1, extra_outputs=False)
    e.reset()
    b = float(_basal(np.array([10]))[0])
    a = torch.full((n,), b, dtype=torch.float64, device=e.device)
    for _ in range(30):
        e.step(a)
    assert float((e.bg - 138.56).abs().max()) < 1e-5      # RK4(4) drift from the CSV's rounded steady state
    assert bool((e.x == e.x[:, :1]).all())
    x_scalar = e.x.clone()
    e2 = _mk(patient="adult#001", n_envs=n, sensor="Navigator", noise="philox", seed=1, extra_outputs=False)
    e2.set_option("adaptive_gut", 2)
    e2.reset()
    for _ in range(30):
        e2.step(a)
    # separately compiled kernels may contract FMAs differently: agreement to rounding, not bitwise
    assert float((x_scalar - e2.x).abs().max()) < 1e-9 and float((e.cgm - e2.cgm).abs().max()) < 1e-9
    assert e.sync() == 0 and e2.sync() == 0


@pytest.mark.parametrize("adaptive", [False, True])
def test_full_size_24h_run_sampled_envs_match_oracle(adaptive):
    """The headline workload end to end at BASELINE's full size: 1 048 576 envs, 24 h of one-minute steps with a
    random-action policy, random meal tables and Philox CGM noise (1 440 launches of the single-minute kernel).
    300 envs sampled across the batch (every patient, first and last workgroups, wave edges) are replayed on the
    oracle with the very normals, meals and actions the kernel used and must agree to 1e-8 mg/dL throughout; the
    distance to the oracle's SciPy-faithful DOPRI5 path (pinned to the reference to 1e-9) is bounded as measured."""
    import torch
    from simglucose_amd import scenario_batch as sb
    from oracle import t1d_oracle as O
    n, K = 1 << 20, 1440
    pid = np.arange(n) % 30
    e = _mk(patient=pid, sensor="Navigator", noise="philox", seed=77, n_sub=4, extra_outputs=False, adaptive_gut=adaptive)
    mt, ma = sb.random_meal_tables(n, days=1, start_minute_of_day=0, seed=5, device=e.device)
    e.set_meals(mt, ma)
    rs = np.random.RandomState(1)
    sample = np.unique(np.concatenate([np.arange(0, 130), np.arange(n - 130, n), rs.randint(0, n, 60)]))[:300]
    z = e.philox_normals(1 + 10 * (1 + K // 150), draw0=0, episode=1)[:, torch.as_tensor(sample, device=e.device)].cpu().numpy()
    t_s, a_s = mt[:, sample].cpu().numpy().astype(np.int64), ma[:, sample].cpu().numpy()
    cho = np.zeros((K, len(sample)))
    for j in range(len(sample)):
        for tt, aa in zip(t_s[:, j], a_s[:, j]):
            if tt < K:
                cho[tt, j] = aa
    b0 = torch.as_tensor(_basal(pid), device=e.device)
    g = torch.Generator(device=e.device); g.manual_seed(3)
    pool = [(b0 * 2.0 * torch.rand(n, generator=g, device=e.device, dtype=torch.float64)).contiguous() for _ in range(8)]
    pool_s = [p[torch.as_tensor(sample, device=e.device)].cpu().numpy() for p in pool]
    orc = O.OracleEnv(pid[sample], sensor="Navigator", normals=z, integrator="split_adaptive" if adaptive else "split", n_sub=4)
    ref = O.OracleEnv(pid[sample], sensor="Navigator", normals=z, integrator="dopri")      # SciPy's DOPRI5 as the reference drives it
    tight = O.OracleEnv(pid[sample], sensor="Navigator", normals=z, integrator="rk4", n_sub=48) if adaptive else None   # the ODE's own solution
    sidx = torch.as_tensor(sample, device=e.device)
    o0, r0 = e.reset(), orc.reset()
    ref.reset()
    if tight is not None:
        tight.reset()
    assert np.abs(o0[sidx].cpu().numpy() - r0["cgm"]).max() < 1e-9
    worst, worst_scipy = 0.0, 0.0
    alive = np.ones(len(sample), bool)                       # fixed steps: compared while BG stays out of the clamp regime
    worst_env = np.zeros(len(sample)); worst_tight = np.zeros(len(sample)); scipy_tight = np.zeros(len(sample))
    for k in range(K):
        e.step(pool[k % 8])
        r = orc.step(pool_s[k % 8], None, cho[k:k + 1])
        rr = ref.step(pool_s[k % 8], None, cho[k:k + 1])
        rt = tight.step(pool_s[k % 8], None, cho[k:k + 1]) if tight is not None else None
        if not adaptive:
            alive &= rr["bg"] >= 20.0                        # level 1 everywhere is not held to the bar at the x3 >= 0 clamp; the default is
        if rt is not None:
            scipy_tight = np.maximum(scipy_tight, np.abs(rr["bg"] - rt["bg"]))
        if k % 16 == 15 or k == K - 1:
            bg = e.bg[sidx].cpu().numpy()
            worst = max(worst, np.abs(bg - r["bg"]).max(), np.abs(e.cgm[sidx].cpu().numpy() - r["cgm"]).max())
            if alive.any():
                worst_scipy = max(worst_scipy, np.abs(bg - rr["bg"])[alive].max())
                worst_env = np.maximum(worst_env, np.where(alive, np.abs(bg - rr["bg"]), 0.0))
            if rt is not None:
                worst_tight = np.maximum(worst_tight, np.abs(bg - rt["bg"]))
    assert worst < 1e-8, worst
    # Measured on this workload (bench.py, 1 024 envs x 24 h): against a tight solve of the ODE the default scheme is within
    # 1.04e-3 (99.8 % of the env-days within 1e-3, median 2.6e-5); SciPy's own default tolerance is up to 3.7e-3 from that
    # solve (99.2 % within 1e-3), so against SciPy 99.1 % of the env-days are within 1e-3 and the worst one is SciPy's own
    # worst.  The bounds below are those figures plus a margin: a scheme twice as far off fails them.
    # Level 1 in every minute (adaptive_gut = 0): about one env-day in twelve goes beyond 1e-3 (worst ~6e-3, steep
    # gastric-emptying patients after large meals) -- DESIGN.md section 4.
    if adaptive:
        assert worst_tight.max() < 1.5e-3 and (worst_tight <= 1e-3).mean() >= 0.99 and np.median(worst_tight) < 6e-5, \
            (worst_tight.max(), (worst_tight <= 1e-3).mean(), np.median(worst_tight))
        assert (worst_env <= 1e-3).mean() >= 0.985 and np.median(worst_env) < 1.0e-4, ((worst_env <= 1e-3).mean(), np.median(worst_env))
        # where the kernel is further than 1e-3 from SciPy, SciPy is at least as far from the ODE's solution
        far = worst_env > 1e-3
        assert (scipy_tight[far] > 0.7e-3).all(), (worst_env[far], scipy_tight[far])
        assert worst_scipy < 1.2 * max(scipy_tight.max(), 1e-3) + 1e-3, (worst_scipy, scipy_tight.max())
    else:
        assert alive.sum() > 250 and worst_scipy < 1e-2, (alive.sum(), worst_scipy)
        assert (worst_env <= 1e-3).mean() > 0.85 and np.median(worst_env) < 4e-4, ((worst_env <= 1e-3).mean(), np.median(worst_env))
    assert np.abs(e.x[:, sidx].cpu().numpy() - orc.x).max() < 1e-6
    assert e.sync() == 0 and bool(torch.isfinite(e.bg).all()) and int(e.t.min()) == K == int(e.t.max())


@pytest.mark.parametrize("sensor,hours", [("Dexcom", 12), ("GuardianRT", 6)])
def test_full_size_multi_minute_steps_sampled_envs_match_oracle(sensor, hours):
    """The reference's own step shape (env.py:75-81: sample_time mini-steps per env.step, Dexcom 3 / GuardianRT 5) at
    BASELINE's full size with the library's defaults -- 1 048 576 fp64 envs: the persistent multi-minute kernel, one launch
    per step --, random-action policy, random meal tables, Philox noise.  300 envs sampled across the batch (every patient,
    first and last workgroups, wave edges) replayed on the oracle with the very normals, meals and actions the kernel used:
    observation (mean of held and fresh CGM samples), mean BG, reward, done and meal agree to 1e-8 throughout, the final
    states to 1e-6; against the oracle's SciPy-faithful DOPRI5 path the BG stays within the bounds of the 1-minute test."""
    import torch
    from simglucose_amd import scenario_batch as sb
    from oracle import t1d_oracle as O
    n = 1 << 20
    pid = np.arange(n) % 30
    e = _mk(patient=pid, sensor=sensor, noise="philox", seed=78, n_sub=4, extra_outputs=True)
    st = int(e.sample_time)
    K = hours * 60 // st
    mt, ma = sb.random_meal_tables(n, days=1, start_minute_of_day=5 * 60, seed=6, device=e.device)
    e.set_meals(mt, ma)
    rs = np.random.RandomState(2)
    sample = np.unique(np.concatenate([np.arange(0, 130), np.arange(n - 130, n), rs.randint(0, n, 60)]))[:300]
    sidx = torch.as_tensor(sample, device=e.device)
    z = e.philox_normals(1 + 10 * (2 + K * st // 150), draw0=0, episode=1)[:, sidx].cpu().numpy()
    t_s, a_s = mt[:, sample].cpu().numpy().astype(np.int64), ma[:, sample].cpu().numpy()
    cho = np.zeros((K * st, len(sample)))
    for j in range(len(sample)):
        for tt, aa in zip(t_s[:, j], a_s[:, j]):
            if tt < K * st:
                cho[tt, j] = aa
    b0 = torch.as_tensor(_basal(pid), device=e.device)
    g = torch.Generator(device=e.device); g.manual_seed(4)
    pool = [(b0 * 2.0 * torch.rand(n, generator=g, device=e.device, dtype=torch.float64)).contiguous() for _ in range(8)]
    pool_s = [p[sidx].cpu().numpy() for p in pool]
    orc = O.OracleEnv(pid[sample], sensor=sensor, normals=z, integrator="split_adaptive", n_sub=4)
    ref = O.OracleEnv(pid[sample], sensor=sensor, normals=z, integrator="dopri")
    o0, r0 = e.reset(), orc.reset()
    ref.reset()
    assert np.abs(o0[sidx].cpu().numpy() - r0["cgm"]).max() < 1e-9
    worst = 0.0
    worst_env = np.zeros(len(sample))
    for k in range(K):
        e.step(pool[k % 8])
        r = orc.step(pool_s[k % 8], None, cho[k * st:(k + 1) * st])
        rr = ref.step(pool_s[k % 8], None, cho[k * st:(k + 1) * st])
        if k % 8 == 7 or k == K - 1:
            for key in ("cgm", "bg", "reward", "meal"):
                worst = max(worst, np.abs(getattr(e, key)[sidx].cpu().numpy() - r[key]).max())
            assert np.array_equal(e.done[sidx].cpu().numpy().astype(bool), np.asarray(r["done"], bool)), k
            worst_env = np.maximum(worst_env, np.abs(e.bg[sidx].cpu().numpy() - rr["bg"]))
    assert worst < 1e-8, worst
    assert (worst_env <= 1e-3).mean() >= 0.985 and np.median(worst_env) < 1.0e-4, ((worst_env <= 1e-3).mean(), np.median(worst_env), worst_env.max())
    assert np.abs(e.x[:, sidx].cpu().numpy() - orc.x).max() < 1e-6
    assert e.sync() == 0 and bool(torch.isfinite(e.bg).all()) and int(e.t.min()) == K * st == int(e.t.max())


@pytest.mark.parametrize("sensor,dtype_name,form", [("Dexcom", "f64", "default"), ("GuardianRT", "f64", "default"), ("Dexcom", "f32", "default"),
                                                    ("Dexcom", "f64", "small_park"), ("Dexcom", "f64", "in_place"), ("Dexcom", "f64", "fixed"),
                                                    ("GuardianRT", "f32", "small_park")])
def test_multi_minute_kernel_equals_generic_kernel(sensor, dtype_name, form):
    """A step of sample_time minutes in one launch of the persistent multi-minute kernel (state in registers across the
    minutes, lanes of level 2 parked in LDS and finished by the pass over the records) against the same step inside one
    launch of the generic kernel: the same lanes refine in the same minutes with the same arithmetic and the outputs are
    summed in the same order -- meals from tables, extra outputs, 8 h.  Forms: the default record capacity; 64 records per
    workgroup and a 3-block grid (most waves find no room and finish their chunks in place, the overflow path); every chunk
    in place (adaptive_gut = 2); level 1 in every minute (adaptive_gut = 0).  Some envs share their meal plan so that whole
    waves are flagged at once."""
    import torch
    from simglucose_amd import scenario_batch as sb
    dt = torch.float64 if dtype_name == "f64" else torch.float32
    n = 64 * 50 - 5
    pid = np.arange(n) % 30
    mt, ma = sb.random_meal_tables(n, days=1, start_minute_of_day=7 * 60, seed=3, device="cuda:0", dtype=dt)
    mt[:, :320] = mt[:, :1]; ma[:, :320] = ma[:, :1]        # five waves of envs with one meal plan
    envs = []
    for mode in (0, 2):
        e = _mk(patient=pid, sensor=sensor, dtype=dt, noise="philox", seed=9, n_sub=4, extra_outputs=True)
        e.set_option("multi_minute_kernel", mode)
        if form == "small_park":
            e.set_option("park_cap", 64); e.set_option("s1_blocks", 3)
        elif form == "in_place":
            e.set_option("adaptive_gut", 2)
        elif form == "fixed":
            e.set_option("adaptive_gut", 0)
        e.set_meals(mt, ma)
        e.reset()
        envs.append(e)
    b = torch.as_tensor(_basal(pid), device="cuda:0", dtype=dt)
    tol = 1e-9 if dtype_name == "f64" else 2e-3
    steps = 480 // int(envs[0].sample_time)
    for k in range(steps):
        a = b * (0.4 + 0.3 * (k % 5))
        envs[0].step(a); envs[1].step(a)
        if k % 20 == 19 or k == 0:
            for key in ("cgm", "bg", "reward", "risk", "lbgi", "hbgi", "meal", "insulin"):
                assert float((getattr(envs[0], key) - getattr(envs[1], key)).abs().max()) < tol, (k, key)
            assert torch.equal(envs[0].done, envs[1].done)
    assert torch.equal(envs[0].t, envs[1].t) and int(envs[0].t[0]) == steps * int(envs[0].sample_time)
    assert torch.equal(envs[0].meta, envs[1].meta) and torch.equal(envs[0].next_meal, envs[1].next_meal)
    assert float((envs[0].x - envs[1].x).abs().max()) < tol * 100
    for key in ("prev_risk", "last_cgm", "planned", "last_qsto", "last_food"):
        assert float((getattr(envs[0], key) - getattr(envs[1], key)).abs().max()) < tol, key
    assert envs[0].sync() == 0 and envs[1].sync() == 0


@pytest.mark.parametrize("sensor", ["Navigator", "Dexcom"])
def test_chunk_order_of_a_launch_does_not_change_results(sensor):
    """The persistent kernels walk a CU's chunks backwards in every other launch (t1d.h "pingpong": the next launch starts on
    what this one touched last, which is what the Infinity Cache still holds).  Envs are independent, so the order must not
    show: with and without it the outputs and states are bit-identical over 40 steps."""
    import torch
    from simglucose_amd import scenario_batch as sb
    n = 64 * 300 - 7
    pid = np.arange(n) % 30
    mt, ma = sb.random_meal_tables(n, days=1, start_minute_of_day=7 * 60, seed=4, device="cuda:0")
    envs = []
    for pp in (0, 1):
        e = _mk(patient=pid, sensor=sensor, noise="philox", seed=10, n_sub=4, extra_outputs=True)
        e.set_option("pingpong", pp); e.set_option("multi_minute_kernel", 2)
        e.set_meals(mt, ma); e.reset(); envs.append(e)
    b = torch.as_tensor(_basal(pid), device="cuda:0")
    for k in range(40):
        for e in envs:
            e.step(b * (0.5 + 0.25 * (k % 4)))
    for key in ("x", "cgm", "bg", "reward", "risk", "prev_risk", "planned", "last_qsto", "last_food", "dbar", "t", "meta", "next_meal", "done"):
        assert torch.equal(getattr(envs[0], key), getattr(envs[1], key)), key
    assert envs[0].sync() == 0 and envs[1].sync() == 0


def test_multi_minute_kernel_odd_shapes_equal_generic_kernel():
    """The persistent multi-minute kernel against the generic kernel over shapes and settings picked to hit its edges: 1,
    63, 65, 640, 4 099 and 70 001 envs (partial chunks, fewer chunks than waves), both sensors and dtypes, n_sub 2-8,
    record capacities 0 (everything through the redo map) to 300, records going ahead of chunks from 1 waiting, grids of 1,
    3 and 17 workgroups, and a masked reset of a third of the envs in the middle.  80 steps each; outputs, clocks, meta words
    and states agree (fp64: rounding; fp32: 5e-3) and no status bit is raised."""
    import torch
    from simglucose_amd import scenario_batch as sb
    rs = np.random.RandomState(0)
    cases = []
    for n in (1, 63, 65, 640, 4099, 70001):
        for sensor in ("Dexcom", "GuardianRT"):
            for dt in (torch.float64, torch.float32):
                cases.append((n, sensor, dt, int(rs.choice([2, 4, 6, 8])), int(rs.choice([0, 1, 7, 64, 300])), int(rs.choice([1, 5, 64])),
                              int(rs.choice([0, 1, 3, 17])), bool(rs.rand() < 0.5)))
    for (n, sensor, dt, n_sub, park, gmin, blocks, extra) in cases:
        pid = rs.randint(0, 30, n)
        mt, ma = sb.random_meal_tables(n, days=1, start_minute_of_day=7 * 60, seed=int(rs.randint(1 << 20)), device="cuda:0", dtype=dt)
        envs = []
        for mode in (0, 2):
            e = _mk(patient=pid, sensor=sensor, dtype=dt, noise="philox", seed=9, n_sub=n_sub, extra_outputs=extra)
            for k, v in (("multi_minute_kernel", mode), ("park_cap", park), ("record_group_min", gmin), ("s1_blocks", blocks)):
                e.set_option(k, v)
            e.set_meals(mt, ma); e.reset(); envs.append(e)
        b = torch.as_tensor(_basal(pid), device="cuda:0", dtype=dt)
        tol = 1e-9 if dt == torch.float64 else 5e-3
        case = (n, sensor, str(dt), n_sub, park, gmin, blocks, extra)
        for k in range(80):
            if k == 40:
                m = torch.as_tensor(np.arange(n) % 3 == 0, device="cuda:0")
                for e in envs:
                    e.reset(mask=m)
            for e in envs:
                e.step(b * (0.3 + 0.4 * (k % 5)))
            if k % 10 == 9 or k == 40:
                for key in ("cgm", "bg", "reward"):
                    assert float((getattr(envs[0], key) - getattr(envs[1], key)).abs().max()) < tol, (case, k, key)
        assert torch.equal(envs[0].t, envs[1].t) and torch.equal(envs[0].meta, envs[1].meta), case
        assert float((envs[0].x - envs[1].x).abs().max()) < tol * 100, case
        assert envs[0].sync() == 0 and envs[1].sync() == 0, case


@pytest.mark.parametrize("dtype_name,ctrl", [("f64", "pid"), ("f64", "bb"), ("f32", "pid")])
def test_rollout_as_one_launch_per_step_equals_single_launch_rollout(dtype_name, ctrl):
    """t1d_rollout_pid / t1d_rollout_bb as one launch of the multi-minute kernel per step (controller fused into the launch,
    what large batches take) against all steps inside one launch of the generic roll-out kernel: states, controller state,
    statistics and the device-resident histories agree -- 6 h of Dexcom steps in calls of different lengths."""
    import torch
    from simglucose_amd import scenario_batch as sb
    dt = torch.float64 if dtype_name == "f64" else torch.float32
    n = 64 * 20 - 3
    pid = np.arange(n) % 30
    mt, ma = sb.random_meal_tables(n, days=1, start_minute_of_day=6 * 60, seed=21, device="cuda:0", dtype=dt)
    res = []
    for mode in (0, 2):
        e = _mk(patient=pid, sensor="Dexcom", dtype=dt, noise="philox", seed=12, n_sub=4, extra_outputs=True)
        e.set_option("rollout_launches", mode)
        e.set_option("multi_minute_kernel", 2)
        e.set_meals(mt, ma)
        e.reset()
        K = 120
        tr = e.new_trace(K)
        stats = {"sum_risk": torch.zeros(n, dtype=dt, device=e.device), "min_bg": torch.full((n,), 1e9, dtype=dt, device=e.device),
                 "max_bg": torch.zeros(n, dtype=dt, device=e.device), "n_low": torch.zeros(n, dtype=torch.int32, device=e.device),
                 "n_high": torch.zeros(n, dtype=torch.int32, device=e.device)}
        st = None
        for chunk in (1, 19, 100):
            if ctrl == "pid":
                st = e.rollout_pid(chunk, 1.5e-4, 4e-7, 5e-4, 140.0, pid_state=st, stats=stats, trace=tr)
            else:
                st = e.rollout_bb(chunk, bb_state=st, stats=stats, trace=tr)
        assert e.sync() == 0
        res.append((e, st, stats, tr))
    (ea, sa, sta, tra), (eb, sb_, stb, trb) = res
    tol = 1e-9 if dtype_name == "f64" else 5e-3
    for k in ("x", "cgm", "bg", "last_cgm", "prev_risk", "reward", "planned", "last_qsto", "last_food", "meal", "insulin", "risk"):
        assert float((getattr(ea, k).double() - getattr(eb, k).double()).abs().max()) < tol * (100 if k == "x" else 1), k
    assert torch.equal(ea.t, eb.t) and torch.equal(ea.meta, eb.meta) and torch.equal(ea.done, eb.done)
    for k in (("integ", "prev") if ctrl == "pid" else ("prev_meal",)):
        assert float((sa[k].double() - sb_[k].double()).abs().max()) < tol * (1e4 if k == "integ" else 1), k
    for k in ("bg", "cgm", "cho", "insulin"):
        d = (tra[k].double() - trb[k].double()).abs()
        assert float(d[torch.isfinite(d)].max()) < tol, k
        assert torch.equal(torch.isnan(tra[k]), torch.isnan(trb[k])), k
    for k in ("min_bg", "max_bg"):
        assert float((sta[k].double() - stb[k].double()).abs().max()) < tol, k
    assert float((sta["sum_risk"].double() - stb["sum_risk"].double()).abs().max()) < tol * 1e3
    if dtype_name == "f64":
        assert torch.equal(sta["n_low"], stb["n_low"]) and torch.equal(sta["n_high"], stb["n_high"])
