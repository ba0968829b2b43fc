"""GPU tests of the drop-in Python surface: the reference's own tests, re-expressed on this package
(tests/test_sim_engine.py::test_results_consistency, test_seed.py, test_reset.py, test_reward_fun.py,
test_gym.py of the reference)."""
import csv
import os
from datetime import datetime, timedelta

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _hist(name):
    with open(os.path.join(GOLDEN, name), newline="") as f:
        rows = list(csv.DictReader(f))
    return {k: np.array([float(r[k]) if r[k] else np.nan for r in rows]) for k in rows[0] if k != "Time"}, \
        [r["Time"] for r in rows]


def test_results_consistency_with_upstream_golden_file():
    """Reference tests/test_sim_engine.py:87-113: adolescent#001, Dexcom(seed 1), Insulet,
    RandomScenario(2018-01-01 00:00, seed 1), BBController, 2 days == tests/sim_results.csv.
    Tolerances: glucose columns to BASELINE.json's 1e-3 mg/dL (RK4(4) in place of adaptive DOPRI5);
    the risk columns are 10 f(BG)^2 with |d risk / d BG| <= ~0.2 per mg/dL -> 2e-4; CHO/insulin exact.
    (The reference's own assert_frame_equal(rtol=1e-5) is met by the oracle's DOPRI5 path on CPU,
    tests/test_oracle_golden.py::test_upstream_golden_file_closed_loop.)"""
    from simglucose_amd.simulation.env import T1DSimEnv
    from simglucose_amd.controller.basal_bolus_ctrller import BBController
    from simglucose_amd.sensor.cgm import CGMSensor
    from simglucose_amd.actuator.pump import InsulinPump
    from simglucose_amd.patient.t1dpatient import T1DPatient
    from simglucose_amd.simulation.scenario_gen import RandomScenario
    from simglucose_amd.simulation.sim_engine import SimObj, sim
    start_time = datetime(2018, 1, 1, 0, 0, 0)
    env = T1DSimEnv(T1DPatient.withName("adolescent#001"), CGMSensor.withName("Dexcom", seed=1),
                    InsulinPump.withName("Insulet"), RandomScenario(start_time=start_time, seed=1))
    results = sim(SimObj(env, BBController(), timedelta(days=2), animate=False, path=None))
    exp, times = _hist("upstream_sim_results.csv")
    assert len(results) == 961 and list(results.columns) == ["BG", "CGM", "CHO", "insulin", "LBGI", "HBGI", "Risk"]
    assert str(results.index[1]) == times[1]
    for col in exp:
        got = results[col].to_numpy()
        assert np.array_equal(np.isnan(got), np.isnan(exp[col])), col
        ok = ~np.isnan(exp[col])
        tol = {"BG": 1e-3, "CGM": 1e-3, "CHO": 1e-12, "insulin": 1e-12}.get(col, 2e-4)
        assert np.abs(got[ok] - exp[col][ok]).max() < tol, (col, np.abs(got[ok] - exp[col][ok]).max())
    assert np.abs(results["BG"].to_numpy() - exp["BG"]).max() < 1e-3          # BASELINE.json's bar
    assert np.abs(results["CGM"].to_numpy() - exp["CGM"]).max() < 1e-3


def test_step_tuple_and_info_keys():
    from simglucose_amd.simulation.env import T1DSimEnv, Observation
    from simglucose_amd.controller.base import Action
    from simglucose_amd.sensor.cgm import CGMSensor
    from simglucose_amd.actuator.pump import InsulinPump
    from simglucose_amd.patient.t1dpatient import T1DPatient
    from simglucose_amd.simulation.scenario import CustomScenario
    t0 = datetime(2018, 1, 1, 6, 0, 0)
    env = T1DSimEnv(T1DPatient.withName("adult#001"), CGMSensor.withName("Dexcom", seed=1),
                    InsulinPump.withName("Insulet"), CustomScenario(t0, [(0.1, 56)]))
    r = env.reset()
    assert isinstance(r.observation, Observation) and r.reward == 0 and r.done is False
    assert set(r.info) == {"sample_time", "patient_name", "meal", "patient_state", "time", "bg", "lbgi", "hbgi", "risk"}
    assert abs(r.info["bg"] - 138.56) < 1e-9 and r.info["time"] == t0 and r.info["patient_name"] == "adult#001"
    assert abs(r.observation.CGM - 151.9733141211806) < 1e-9            # known answer
    s1 = env.step(Action(basal=0.02, bolus=0))
    s2 = env.step(Action(basal=0.02, bolus=0))
    s3 = env.step(Action(basal=0.02, bolus=0))                          # minute 6 = 0.1 h: the meal is announced
    assert s1.info["meal"] == 0 and abs(s3.info["meal"] - 56 / 3.0) < 1e-12
    assert env.time == t0 + timedelta(minutes=9) and s3.info["patient_state"].shape == (13,)
    assert env.patient.t == 9 and abs(env.patient.observation.Gsub - s3.info["patient_state"][12] / 1.9152) < 1e-9
    df = env.show_history()
    assert len(df) == 4 and np.isnan(df["CHO"].iloc[-1]) and abs(df["insulin"].iloc[0] - 0.02) < 1e-12
    # custom reward function sees the CGM history window (env.py:100-102)
    seen = []
    env.step(Action(basal=0.02, bolus=0), reward_fun=lambda w: seen.append(list(w)) or -len(w))
    assert seen[0] == env.CGM_hist[-20:] and len(seen[0]) == 5


def test_gym_wrapper_seed_reset_semantics():
    """Reference tests/test_seed.py and tests/test_reset.py."""
    from simglucose_amd.envs import T1DSimEnv
    env = T1DSimEnv(patient_name="adult#001")
    seeds = env.seed(0)
    assert len(seeds) == 4 and seeds[0] == 0
    obs0 = env.reset()
    assert env.env.scenario.start_time == datetime(2018, 1, 1, 23, 0, 0)
    env.seed(1000)
    obs1 = env.reset()
    assert env.env.scenario.start_time == datetime(2018, 1, 1, 14, 0, 0)
    assert obs0 != obs1
    # same seed => same sequence of (obs, start time, scenario) across resets; successive resets differ
    runs = []
    for _ in range(2):
        env.seed(7)
        seq = []
        for _ in range(3):
            o = env.reset()
            seq.append((o.CGM, env.env.scenario.start_time, tuple(env.env.scenario.scenario["meal"]["time"])))
        runs.append(seq)
    assert runs[0] == runs[1]
    assert runs[0][0] != runs[0][1] and runs[0][1] != runs[0][2]
    assert env.action_space.shape == (1,) and float(env.action_space.high[0]) == 30.0
    assert env.observation_space.shape == (1,)
    # 30 gym steps with a scalar basal action; custom reward fun contract (tests/test_reward_fun.py)
    def reward(window):
        return 1 if 70 <= window[-1] <= 180 else (-1 if window[-1] > 180 else -2)
    env2 = T1DSimEnv(patient_name="adolescent#002", reward_fun=reward, seed=3)
    env2.reset()
    for _ in range(30):
        obs, r, done, info = env2.step(np.array([0.015]))
        assert r == reward([obs.CGM]) and info["sample_time"] == 3.0 and isinstance(done, bool)


def test_standalone_patient_steps():
    """T1DPatient used alone (t1dpatient.py:297-320 style): step(Action(CHO, insulin)) once per minute."""
    from simglucose_amd.patient.t1dpatient import T1DPatient, Action
    from oracle import t1d_oracle as O
    p = T1DPatient.withName("child#005")
    names, tab = O.patient_table()
    orc = O.PatientOracle(tab[names.index("child#005")])
    basal = float(p._params.u2ss * p._params.BW / 6000)
    for t in range(90):
        cho = 50.0 if t == 10 else 0.0
        ins = basal * (3.0 if 10 <= t < 13 else 1.0)
        p.step(Action(CHO=cho, insulin=ins))
        orc.step(cho, ins, integrator="split_adaptive", n_sub=4)
    assert p.t == 90
    assert np.abs(p.state - orc.x).max() < 1e-7
    p.reset()
    assert p.t == 0 and np.array_equal(p.state, tab[names.index("child#005"), :13])


def test_batched_gym_env_exact_equals_n_single_wrappers():
    """BatchedGymT1DSimEnv(exact=True): env i == the single-env gym wrapper seeded with seed + i."""
    import torch
    from simglucose_amd.envs import T1DSimEnv, BatchedGymT1DSimEnv
    n, seed = 3, 11
    benv = BatchedGymT1DSimEnv(n, patient_name="adult#003", seed=seed, exact=True)
    singles = []
    for i in range(n):
        e = T1DSimEnv(patient_name="adult#003")
        e.seed(seed + i)
        singles.append(e)
    for episode in range(2):
        ob = benv.reset().cpu().numpy()
        os_ = [e.reset().CGM for e in singles]
        assert np.abs(ob - np.array(os_)).max() < 1e-9
        assert [int(h) for h in benv.start_hour] == [e.env.scenario.start_time.hour for e in singles]
        for k in range(60):
            a = 0.01 + 0.002 * (k % 5)
            obs, rew, done, info = benv.step(torch.full((n,), a, dtype=torch.float64))
            for i, e in enumerate(singles):
                o, r, d, inf = e.step(a)
                assert abs(float(obs[i]) - o.CGM) < 1e-9 and abs(float(rew[i]) - r) < 1e-9 and bool(done[i]) == d
                assert abs(float(info["meal"][i]) - inf["meal"]) < 1e-12


def test_batched_gym_env_device_mode_and_auto_reset():
    import torch
    from simglucose_amd.envs import BatchedGymT1DSimEnv
    n = 4096
    env = BatchedGymT1DSimEnv(n, patient_name=["child#001", "adult#001"] * (n // 2), seed=3, auto_reset=True)
    obs = env.reset()
    assert obs.shape == (n,) and bool((obs >= 39).all()) and int(env.start_hour.min()) >= 0 and int(env.start_hour.max()) <= 23
    bg0 = env.env.bg.clone()
    assert 2.3 < float(bg0[1::2].std()) < 3.1          # random_init_bg: sd = sqrt(0.1 x0_13)/Vg = 2.69 mg/dL around 138.56
    n_done = 0
    for k in range(200):
        obs, rew, done, info = env.step(torch.full((n,), 0.05, dtype=torch.float64, device=obs.device))   # heavy basal: hypos
        n_done += int(done.sum())
        assert bool(torch.isfinite(obs).all()) and bool(torch.isfinite(rew).all())
    assert n_done > 0                                  # some episodes ended and were re-started in place
    assert int(env.env.t.min()) < 600 and int(env.env.t.max()) == 600
    assert int(env.env.episode.max()) >= 2
    assert env.env.sync() == 0


def test_batched_reward_fun_equals_single_env_adapters():
    """Custom reward functions on the batch (reference: T1DSimEnv.step(action, reward_fun) hands the function the last
    hour of CGM_hist, simulation/env.py:100-102; rule of examples/custom_reward_function.py:5-14 and
    tests/test_reward_fun.py:6-12, plus one that uses the whole window): N envs of the batched env give what N
    single-env adapters give, step by step -- rewards, observations and the window itself, incl. the first steps
    of an episode, where the history is shorter than an hour, and across a reset."""
    import torch
    from simglucose_amd.batch_env import BatchedT1DSimEnv
    from simglucose_amd.envs import BatchedGymT1DSimEnv
    from simglucose_amd.simulation.env import T1DSimEnv
    from simglucose_amd.controller.base import Action
    from simglucose_amd.sensor.cgm import CGMSensor
    from simglucose_amd.actuator.pump import InsulinPump
    from simglucose_amd.patient.t1dpatient import T1DPatient
    from simglucose_amd.simulation.scenario import CustomScenario

    def custom_reward(BG_last_hour):                                   # the reference's example rule
        if BG_last_hour[-1] > 180:
            return -1
        elif BG_last_hour[-1] < 70:
            return -2
        return 1

    def mean_drop(BG_last_hour):                                       # uses the whole window
        return float(BG_last_hour[0] - np.mean(BG_last_hour))

    def custom_reward_b(w):
        return torch.where(w[-1] > 180, -1.0, torch.where(w[-1] < 70, -2.0, 1.0))

    def mean_drop_b(w):
        first = torch.gather(w, 0, torch.isnan(w).sum(0, keepdim=True).clamp(max=w.shape[0] - 1))[0]    # oldest valid sample
        return first - torch.nanmean(w, 0)

    names = ["adolescent#003", "adult#002", "child#006"]
    seeds = [3, 8, 21]
    t0 = datetime(2018, 1, 1, 6, 0, 0)
    scen = [(0.5, 45), (1.5, 20)]
    K, st = 45, 3
    n = len(names)
    z = np.stack([np.random.RandomState(sd).randn(1 + 10 * 16) for sd in seeds], axis=1)
    cho = np.zeros((K * st, n))
    for h, g in scen:
        cho[int(round(h * 60))] = g
    rs = np.random.RandomState(0)
    acts = rs.uniform(0.0, 0.04, (K, n))
    for rule, rule_b in ((custom_reward, custom_reward_b), (mean_drop, mean_drop_b)):
        singles = [T1DSimEnv(T1DPatient.withName(nm), CGMSensor.withName("Dexcom", seed=sd), InsulinPump.withName("Insulet"),
                             CustomScenario(start_time=t0, scenario=scen)) for nm, sd in zip(names, seeds)]
        be = BatchedT1DSimEnv(patient=names, sensor="Dexcom", noise="host", normals=z, cgm_history=True)
        assert be.window == 20
        for episode in range(2):
            for s in singles:
                s.reset()
            be.reset()
            w0 = be.cgm_window().cpu().numpy()
            assert np.isnan(w0[:-1]).all() and np.abs(w0[-1] - [s.CGM_hist[0] for s in singles]).max() < 1e-9
            for k in range(K if episode == 0 else 8):
                want = [s.step(Action(basal=acts[k, j], bolus=0), reward_fun=rule) for j, s in enumerate(singles)]
                obs, rew, done, info = be.step(acts[k], cho=cho[k * st:(k + 1) * st], reward_fun=rule_b)
                assert np.abs(obs.cpu().numpy() - [w.observation.CGM for w in want]).max() < 1e-9, k
                assert np.abs(rew.cpu().numpy() - [w.reward for w in want]).max() < 1e-9, (rule.__name__, k)
                win = be.cgm_window().cpu().numpy()
                for j, s in enumerate(singles):
                    h = np.array(s.CGM_hist[-20:])
                    assert np.abs(win[20 - len(h):, j] - h).max() < 1e-9 and np.isnan(win[:20 - len(h), j]).all()
        assert be.sync() == 0
    # the gym-style batch takes the function at construction, as the reference wrapper does
    genv = BatchedGymT1DSimEnv(256, patient_name="adolescent#002", seed=5, reward_fun=custom_reward_b)
    o = genv.reset()
    for _ in range(30):
        o, r, d, info = genv.step(torch.full((256,), 0.02, dtype=torch.float64, device=o.device))
        assert torch.equal(r, custom_reward_b(o.unsqueeze(0)))
    plain = BatchedT1DSimEnv(patient="adult#001", n_envs=4)
    plain.reset()
    with pytest.raises(Exception):
        plain.step(0.01, reward_fun=custom_reward_b)                    # needs cgm_history
