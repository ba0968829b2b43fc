#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING the reference.

TEST INFRASTRUCTURE -- build-container only.  This script imports the
read-only reference checkout at /root/reference (which never travels to the
GPU box) and records inputs + the reference's outputs as small .npz/.csv
fixtures.  Nothing from the reference's source text is copied: the fixtures
hold numbers only.  Re-run with

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python oracle/gen_golden.py [--only G1,G5] [--jobs 8]

The reference's top-level ``simglucose/__init__.py`` imports ``gym`` (absent in
this image), so an empty parent package is registered in-process and the
hot-path sub-modules are imported underneath it unmodified.

Fixture ids follow SURVEY.md §8(c): G1 RHS known answers, G2 open-loop
24 h traces (SciPy-default and tight), G3 pump, G4 sensor noise, G5 env-level
reset/step, G6 config-1 closed loop (BBController), G7 upstream golden CSV
re-generation check, G8 risk index, G9 random_init_bg + scenario draws,
G10 PID closed loop, G11 analysis/report.py statistics.
"""
import argparse
import os
import sys
import types
import warnings
from datetime import datetime, timedelta

import numpy as np

REF = os.environ.get("T1D_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

warnings.simplefilter("ignore")
os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True


def _import_reference():
    pkg = types.ModuleType("simglucose")
    pkg.__path__ = [os.path.join(REF, "simglucose")]
    pkg.__file__ = os.path.join(REF, "simglucose", "__init__.py")
    sys.modules["simglucose"] = pkg
    import logging
    logging.disable(logging.CRITICAL)


_import_reference()
import pandas as pd  # noqa: E402
from scipy.integrate import ode  # noqa: E402
from simglucose.patient.t1dpatient import T1DPatient, Action as PAction  # noqa: E402
from simglucose.sensor.cgm import CGMSensor  # noqa: E402
from simglucose.sensor.noise_gen import CGMNoise  # noqa: E402
from simglucose.actuator.pump import InsulinPump  # noqa: E402
from simglucose.simulation.env import T1DSimEnv  # noqa: E402
from simglucose.simulation.scenario import CustomScenario  # noqa: E402
from simglucose.simulation.scenario_gen import RandomScenario  # noqa: E402
from simglucose.controller.base import Action as CAction  # noqa: E402
from simglucose.controller.basal_bolus_ctrller import BBController  # noqa: E402
from simglucose.controller.pid_ctrller import PIDController  # noqa: E402
from simglucose.analysis.risk import risk_index  # noqa: E402

PTABLE = pd.read_csv(os.path.join(REF, "simglucose", "params", "vpatient_params.csv"))
NAMES = list(PTABLE.Name)
T0 = datetime(2018, 1, 1, 0, 0, 0)


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print("wrote", path, {k: np.asarray(v).shape for k, v in arrs.items()})


# --------------------------------------------------------------------------- G1
def g1_rhs():
    """T1DPatient.model known answers on random points covering every branch."""
    rs = np.random.RandomState(20240101)
    per = 40
    xs, chos, inss, lq, lf, pid, out = [], [], [], [], [], [], []
    for ip, name in enumerate(NAMES):
        row = PTABLE.iloc[ip]
        x0 = row.iloc[2:15].to_numpy(dtype=float)
        for j in range(per):
            x = x0 * rs.uniform(0.2, 2.5, 13)
            x[0:3] = rs.uniform(0, 60000, 3) * (rs.rand(3) < 0.7)
            x[6] = rs.uniform(-50, 300)
            kind = j % 8
            if kind == 1:      # x3 above renal threshold ke2=339
                x[3] = rs.uniform(340, 700)
            elif kind == 2:    # EGP clipped at zero (large x3 and x8)
                x[3] = rs.uniform(500, 900); x[8] = rs.uniform(300, 900)
            elif kind == 3:    # negative states -> derivative clamp active
                for i in (3, 4, 5, 9, 10, 11, 12):
                    if rs.rand() < 0.6:
                        x[i] = -abs(x[i]) * rs.uniform(0.01, 1)
            elif kind == 4:    # exact zeros (>= 0 keeps the derivative)
                for i in (3, 4, 5, 9, 10, 11, 12):
                    if rs.rand() < 0.5:
                        x[i] = 0.0
            cho = [0.0, 5.0, 2.5, 0.0][j % 4]
            ins = rs.uniform(0, 0.2) if j % 3 else 0.0
            if j % 5 == 0:
                last_qsto, last_food = 0.0, 0.0          # Dbar <= 0 branch
            else:
                last_qsto, last_food = rs.uniform(0, 30000), rs.uniform(0, 100) * (j % 2)
                if last_qsto + last_food * 1000 <= 0:
                    last_food = 5.0
            d = T1DPatient.model(0.0, x, PAction(CHO=cho, insulin=ins), row, last_qsto, last_food)
            xs.append(x); chos.append(cho); inss.append(ins); lq.append(last_qsto)
            lf.append(last_food); pid.append(ip); out.append(d)
    save("g1_rhs.npz", x=np.array(xs), cho=np.array(chos), insulin=np.array(inss),
         last_qsto=np.array(lq), last_foodtaken=np.array(lf),
         patient_idx=np.array(pid, dtype=np.int32), dxdt=np.array(out))


# --------------------------------------------------------------------------- G2
G2_MEALS = ((7 * 60, 45.0), (12 * 60, 70.0), (18 * 60, 80.0))   # minute, grams
G2_MIN = 1440


def g2_actions():
    """Config-2 random-action policy: one U(0,2) multiplier of basal per minute."""
    return np.random.RandomState(0).uniform(0.0, 2.0, G2_MIN)


def _g2_one(args):
    ip, tight = args
    _import_reference()
    row = PTABLE.iloc[ip]
    p = T1DPatient(row.copy())
    if tight:
        p._odesolver = ode(p.model).set_integrator("dopri5", rtol=1e-13, atol=1e-13, nsteps=100000)
        p._odesolver.set_initial_value(np.asarray(p.init_state, dtype=float), p.t0)
    basal = float(row.u2ss * row.BW / 6000.0)
    mult = g2_actions()
    meal_at = dict(G2_MEALS)
    X = np.empty((G2_MIN + 1, 13))
    X[0] = p.state
    for t in range(G2_MIN):
        p.step(PAction(CHO=meal_at.get(t, 0.0), insulin=basal * mult[t]))
        X[t + 1] = p.state
    return ip, tight, X


def g2_openloop(jobs):
    from multiprocessing import Pool
    tasks = [(ip, tight) for tight in (False, True) for ip in range(30)]
    with Pool(jobs) as pool:
        res = pool.map(_g2_one, tasks, chunksize=1)
    Xd = np.empty((30, G2_MIN + 1, 13)); Xt = np.empty_like(Xd)
    for ip, tight, X in res:
        (Xt if tight else Xd)[ip] = X
    vg = PTABLE.Vg.to_numpy(dtype=float)
    basal = (PTABLE.u2ss * PTABLE.BW / 6000.0).to_numpy(dtype=float)
    save("g2_openloop.npz",
         gsub_default=Xd[:, :, 12] / vg[:, None], gsub_tight=Xt[:, :, 12] / vg[:, None],
         state_default_10min=Xd[:, ::10, :], state_tight_10min=Xt[:, ::10, :],
         state_default_full_adult001=Xd[10], state_default_full_child001=Xd[20],
         state_default_full_adolescent001=Xd[0],
         action_mult=g2_actions(), basal=basal,
         meal_minute=np.array([m for m, _ in G2_MEALS], dtype=np.int32),
         meal_grams=np.array([g for _, g in G2_MEALS]))


# --------------------------------------------------------------------------- G3
def g3_pump():
    inc = 0.05 / 6000.0
    grid = [-1.0, -1e-9, 0.0, 1e-6, 0.0139333, 0.02, 0.5, 1.0, 29.999999, 30.0, 31.0, 35.0, 36.0, 74.9, 75.0, 80.0]
    grid += [(k + 0.5) * inc for k in range(0, 40)]                 # half-increment ties
    grid += [(k + 0.5) * inc * (1 + 1e-12) for k in range(0, 8)]
    grid += list(np.random.RandomState(3).uniform(0, 0.1, 200))
    grid += list(np.random.RandomState(4).uniform(0, 40, 100))
    grid = np.array(grid)
    out = {}
    for name in ("Insulet", "Cozmo"):
        pump = InsulinPump.withName(name)
        out["basal_" + name] = np.array([pump.basal(a) for a in grid], dtype=float)
        out["bolus_" + name] = np.array([pump.bolus(a) for a in grid], dtype=float)
    save("g3_pump.npz", amount=grid, **out)


# --------------------------------------------------------------------------- G4
def g4_sensor():
    sens = pd.read_csv(os.path.join(REF, "simglucose", "params", "sensor_params.csv"))
    out = {}
    nsamp = 500
    for _, row in sens.iterrows():
        st = float(row.sample_time)
        nblock = int(150 // st)
        # The spline block is a fixed linear map of the 11 points: probe it with unit vectors
        W = np.empty((nblock, 11))
        for k in range(11):
            g = CGMNoise(row, seed=0)
            g._noise_init = 1.0 if k == 0 else 0.0
            unit = iter([1.0 if (j + 1) == k else 0.0 for j in range(10)])
            g._noise15_gen = unit
            W[:, k] = np.array(g._get_noise_seq())
        out["W_" + row.Name] = W
        for seed in (0, 1, 7):
            g = CGMNoise(row, seed=seed)
            out["noise_%s_seed%d" % (row.Name, seed)] = np.array([next(g) for _ in range(nsamp)])
            ndraw = 1 + 10 * (nsamp // nblock + 1)
            out["randn_seed%d" % seed] = np.random.RandomState(seed).randn(max(ndraw, 200))[:200]
    save("g4_sensor.npz", **out)


# --------------------------------------------------------------------------- G5
G5_SCEN = ((1.0, 45.0), (5.5, 70.0), (11.0, 80.0), (16.25, 30.0))     # hours after start, grams


def g5_env():
    out = {}
    for sensor_name, seed, nstep in (("Dexcom", 1, 480), ("Navigator", 2, 600), ("GuardianRT", 3, 288)):
        for pname in ("adult#001", "child#003"):
            p = T1DPatient.withName(pname)
            s = CGMSensor.withName(sensor_name, seed=seed)
            pump = InsulinPump.withName("Insulet")
            scen = CustomScenario(start_time=T0, scenario=[(h, g) for h, g in G5_SCEN])
            env = T1DSimEnv(p, s, pump, scen)
            basal = float(p._params.u2ss * p._params.BW / 6000.0)
            rs = np.random.RandomState(100 + seed)
            act = basal * rs.uniform(0, 2, nstep)
            bol = np.where(rs.rand(nstep) < 0.02, rs.uniform(0, 2.0, nstep), 0.0)
            r0 = env.reset()
            cols = {k: [] for k in ("cgm", "bg", "reward", "done", "lbgi", "hbgi", "risk", "meal")}
            states = []
            for k in range(nstep):
                st = env.step(CAction(basal=act[k], bolus=bol[k]))
                cols["cgm"].append(st.observation.CGM); cols["bg"].append(st.info["bg"])
                cols["reward"].append(st.reward); cols["done"].append(st.done)
                cols["lbgi"].append(st.info["lbgi"]); cols["hbgi"].append(st.info["hbgi"])
                cols["risk"].append(st.info["risk"]); cols["meal"].append(st.info["meal"])
                states.append(np.array(st.info["patient_state"], dtype=float))
            tag = "%s_%s" % (sensor_name, pname.replace("#", ""))
            hist = env.show_history()
            out["reset_cgm_" + tag] = np.array(r0.observation.CGM)
            out["reset_info_" + tag] = np.array([r0.info["bg"], r0.info["lbgi"], r0.info["hbgi"], r0.info["risk"]])
            out["hist0_cgm_" + tag] = np.array(hist.CGM.iloc[0])
            out["basal_" + tag] = act; out["bolus_" + tag] = bol
            out["insulin_hist_" + tag] = hist.insulin.to_numpy()[:nstep]
            out["cho_hist_" + tag] = hist.CHO.to_numpy()[:nstep]
            out["state_" + tag] = np.array(states)
            for k, v in cols.items():
                out[k + "_" + tag] = np.array(v, dtype=float)
            out["randn_" + tag] = np.random.RandomState(seed).randn(1 + 10 * 12)
    out["scen_hours"] = np.array([h for h, _ in G5_SCEN]); out["scen_grams"] = np.array([g for _, g in G5_SCEN])
    save("g5_env.npz", **out)


# --------------------------------------------------------------------------- G6 / G7 / G10
def _closed_loop(pname, sensor_seed, scen_seed, controller, days, sensor_name="Dexcom"):
    p = T1DPatient.withName(pname)
    s = CGMSensor.withName(sensor_name, seed=sensor_seed)
    env = T1DSimEnv(p, s, InsulinPump.withName("Insulet"), RandomScenario(start_time=T0, seed=scen_seed))
    controller.reset()
    obs, reward, done, info = env.reset()
    acts = []
    while env.time < T0 + timedelta(days=days):
        a = controller.policy(obs, reward, done, **info)
        acts.append((a.basal, a.bolus))
        obs, reward, done, info = env.step(a)
    return env.show_history(), np.array(acts, dtype=float)


def g6_config1():
    df, acts = _closed_loop("adult#001", 1, 1, BBController(), 1)
    df.to_csv(os.path.join(OUT, "g6_config1_adult001_bb.csv"))
    save("g6_config1_actions.npz", actions=acts)


def g7_upstream():
    df, acts = _closed_loop("adolescent#001", 1, 1, BBController(), 2)
    exp = pd.read_csv(os.path.join(REF, "tests", "sim_results.csv"), index_col=0)
    err = {c: float(np.nanmax(np.abs(df[c].to_numpy() - exp[c].to_numpy()))) for c in exp.columns}
    print("G7 live reference vs upstream tests/sim_results.csv max-abs:", err)
    assert max(err.values()) < 1e-9
    save("g7_upstream_actions.npz", actions=acts, maxabs=np.array([err[c] for c in exp.columns]))


def g10_pid():
    df, acts = _closed_loop("adult#001", 5, 9, PIDController(P=0.001, I=0.00001, D=0.001, target=140), 1)
    df.to_csv(os.path.join(OUT, "g10_pid_adult001.csv"))
    save("g10_pid_actions.npz", actions=acts, randn=np.random.RandomState(5).randn(1 + 10 * 12))


# --------------------------------------------------------------------------- G8
def g8_risk():
    bg = np.array([1.0, 2.0, 10.0, 39.0, 69.99, 70.0, 100.0, 112.5, 112.51754, 140.0, 180.0, 350.0, 350.01, 600.0, 1000.0])
    out = np.array([risk_index([b], 1) for b in bg], dtype=float)
    save("g8_risk.npz", bg=bg, lbgi=out[:, 0], hbgi=out[:, 1], risk=out[:, 2])


# --------------------------------------------------------------------------- G9
def g9_seeding():
    out = {}
    # random_init_bg: first and second (compounded, quirk 9) reset draws
    names = ("adult#001", "adolescent#001", "child#001")
    seeds = (0, 1, 2, 3)
    first = np.empty((len(names), len(seeds), 13)); second = np.empty_like(first)
    for i, n in enumerate(names):
        for j, sd in enumerate(seeds):
            p = T1DPatient.withName(n, random_init_bg=True, seed=sd)
            first[i, j] = np.asarray(p.state, dtype=float)
            p.reset()
            second[i, j] = np.asarray(p.state, dtype=float)
    out["init_names"] = np.array(names); out["init_seeds"] = np.array(seeds)
    out["init_first"] = first; out["init_second"] = second
    # RandomScenario draws: constructor draw and following re-draws, several seeds
    sseeds = (0, 1, 2, 25, 1000)
    ndraw = 4
    T = np.full((len(sseeds), ndraw, 6), -1.0); A = np.full_like(T, -1.0); C = np.zeros((len(sseeds), ndraw), dtype=np.int32)
    for i, sd in enumerate(sseeds):
        sc = RandomScenario(start_time=T0, seed=sd)
        for d in range(ndraw):
            s = sc.scenario if d == 0 else sc.create_scenario()
            n = len(s["meal"]["time"]); C[i, d] = n
            T[i, d, :n] = s["meal"]["time"]; A[i, d, :n] = s["meal"]["amount"]
    out["scen_seeds"] = np.array(sseeds); out["scen_time"] = T; out["scen_amount"] = A; out["scen_count"] = C
    # per-minute announced meal for 2 days from a midnight start and a 14:00 start (quirk 4)
    for tag, start in (("00h", T0), ("14h", datetime(2018, 1, 1, 14, 0, 0))):
        sc = RandomScenario(start_time=start, seed=1)
        sc.reset()
        out["scen_minute_meal_" + tag] = np.array(
            [sc.get_action(start + timedelta(minutes=m)).meal for m in range(2 * 1440)], dtype=float)
    save("g9_seeding.npz", **out)


# --------------------------------------------------------------------------- G11
def g11_report():
    """analysis/report.py statistics on a 24 h x 30 patient BG table (the G2 traces, spread per column so that every
    range, risk sign and CVGA zone occurs)."""
    from simglucose.analysis.report import percent_stats, CVGA_analysis
    import matplotlib.pyplot as plt
    g2 = np.load(os.path.join(OUT, "g2_openloop.npz"))
    bg = g2["gsub_default"].T.copy()                                   # [1441, 30]
    rs = np.random.RandomState(11)
    scale = rs.uniform(0.1, 2.6, 30) * np.where(np.arange(30) % 3 == 2, -0.5, 1.0)     # every third column mirrored
    shift = rs.uniform(-70, 30, 30)
    bg = np.maximum(110.0 + (bg - 138.0) * scale[None, :] + shift[None, :], 5.0)
    bg[::97, 3] = bg[5, 3]                                             # ties among the order statistics
    df = pd.DataFrame(bg, columns=["p%02d" % k for k in range(30)])
    p_stats, _, _ = percent_stats(df)
    bmin, bmax, pa, pb, pc, pd_, pe = CVGA_analysis(df)
    plt.close("all")
    # risk_index_trace (report.py:95-133) cannot be recorded here: with the installed pandas 2.3 / numpy 2.2 its
    # np.mean(DataFrame) yields a scalar and the function raises TypeError in pd.concat (an ordinary error of the
    # reference under newer libraries).  It is pinned by fixture G12 instead: the output files the reference holds.
    cols = ["BG>180", "BG<70", "70<=BG<=180", "BG>250", "BG<50"]
    save("g11_report.npz", bg=bg, percent=np.stack([p_stats[c].values for c in cols]),
         bg_min=np.asarray(bmin, float), bg_max=np.asarray(bmax, float),
         zones=np.array([pa, pb, pc, pd_, pe], float))


def g12_report_2017():
    """G12: output files the reference itself holds (examples/results/2017-12-31_17-46-32: 30 patients x 24 h,
    BBController, produced by the reference's authors with the library versions of the time): the BG column of every
    per-patient <name>.csv (481 rows, 3-minute steps) as input, and what the reference's report functions wrote for
    them -- risk_trace.csv (risk_index_trace: LBGI / HBGI per 60-row chunk, report.py:95-110), performance_stats.csv
    (percent_stats + mean risk indices) and CVGA_stats.csv -- as expected outputs.  Data only: nothing is executed."""
    import csv
    d = os.path.join(REF, "examples", "results", "2017-12-31_17-46-32")
    with open(os.path.join(d, "performance_stats.csv"), newline="") as f:
        perf = list(csv.reader(f))
    names = [r[0] for r in perf[1:]]
    bg = []
    for nm in names:
        with open(os.path.join(d, nm + ".csv"), newline="") as f:
            bg.append([float(r["BG"]) for r in csv.DictReader(f)])
    bg = np.array(bg).T
    with open(os.path.join(d, "risk_trace.csv"), newline="") as f:
        rt = list(csv.reader(f))
    tr = {(r[0], r[1]): [float(v) if v != "" else np.nan for v in r[2:]] for r in rt[1:]}
    lbgi = np.array([tr[("LBGI", nm)] for nm in names]).T
    hbgi = np.array([tr[("HBGI", nm)] for nm in names]).T
    cols = perf[0][1:]
    pstat = np.array([[float(v) for v in r[1:]] for r in perf[1:]])
    with open(os.path.join(d, "CVGA_stats.csv"), newline="") as f:
        cv = list(csv.reader(f))
    save("g12_report_2017.npz", names=np.array(names), bg=bg, lbgi_trace=lbgi, hbgi_trace=hbgi,
         perf_cols=np.array(cols), perf=pstat, cvga_zones=np.array([float(v) for v in cv[1][1:]]))


ALL = {"G12": g12_report_2017, "G11": g11_report, "G1": g1_rhs, "G3": g3_pump, "G4": g4_sensor, "G5": g5_env, "G6": g6_config1,
       "G7": g7_upstream, "G8": g8_risk, "G9": g9_seeding, "G10": g10_pid}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--jobs", type=int, default=8)
    a = ap.parse_args()
    want = [w for w in a.only.split(",") if w] or list(ALL) + ["G2"]
    os.makedirs(OUT, exist_ok=True)
    for k in want:
        if k == "G2":
            g2_openloop(a.jobs)
        else:
            ALL[k]()
