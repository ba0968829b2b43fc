#!/usr/bin/env python3
"""CPU study (oracle only) of the split integrator's step-size rule: error against the SciPy-faithful
DOPRI5 path and a tight solve, and the share of env-minutes per level, on three workloads:
  open   RandomScenario days with a new random basal rate every minute (what bench.py times), 1-minute sensor
  bb     closed loop with the BBController (boluses: EGP floor, renal threshold), Dexcom, 30 patients x seeds
  pid    closed loop with the reference test's PID gains (winds up: BG -> 0, the x3 >= 0 clamp), Dexcom
usage: tier_study.py [open|bb|pid] [n_envs] [knob=value ...]     knobs: near move kink snap stiff"""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from oracle import t1d_oracle as O

KNOBS = {"near": 0, "move": 1, "kink": 2, "snap": 3, "stiff": 4}
what = sys.argv[1] if len(sys.argv) > 1 else "open"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 600
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    O.set_knob(KNOBS[k], float(v))
names, tab = O.patient_table()
rs = np.random.RandomState(2024)
pid = np.arange(n) % 30
days = 1 if what == "open" else 2
sensor = "Navigator" if what == "open" else "Dexcom"
st = int(O.sensor_row(sensor)[5])
K = days * 1440 // st
cho = np.zeros((K * st, n))
for j in range(n):
    for dday in range(days):
        t, a = O.random_scenario_draw(rs)
        for tt, aa in zip(t, a):
            cho[dday * 1440 + int(tt), j] = aa
basal0 = tab[pid, O.IDX["u2ss"]] * tab[pid, O.IDX["BW"]] / 6000.0
pool = [basal0 * 2 * rs.rand(n) for _ in range(8)]
z = rs.randn(1 + 10 * (2 + K * st // 150), n) if what != "open" else np.zeros((120, n))
quest = O.quest_table()
cr = np.array([quest[names[p]]["CR"] for p in pid]); cf = np.array([quest[names[p]]["CF"] for p in pid])


def run(integ, ns):
    t0 = time.time()
    e = O.OracleEnv(pid, sensor=sensor, normals=z, integrator=integ, n_sub=ns)
    r = e.reset()
    out = np.empty((K, n))
    obs, meal = r["cgm"], np.zeros(n)
    integ_s, prev = np.zeros(n), np.zeros(n)
    for k in range(K):
        if what == "open":
            bas, bol = pool[k % 8], None
        elif what == "bb":
            bas = basal0
            bol = np.where(meal > 0, (meal * st / cr + (obs > 150) * (obs - 140.0) / cf) / st, 0.0)
        else:
            bas = 1e-3 * (obs - 140.0) + 1e-5 * integ_s + 1e-3 * (obs - prev) / st
            prev = obs.copy(); integ_s = integ_s + (obs - 140.0) * st
            bol = None
        o = e.step(bas, bol, cho[k * st:(k + 1) * st])
        obs, meal = o["cgm"], o["meal"]
        out[k] = o["bg"]
    lc = e.level_count
    print("%-15s n_sub %-3d %4.0f s  levels %s" % (integ, ns, time.time() - t0, (lc / max(lc.sum(), 1)).round(4) if lc.sum() else "-"), flush=True)
    return out


def rep(name, o, r, label):
    w = np.abs(o - r).max(0)
    print("  %-15s %-9s median %.1e p95 %.1e p99 %.1e max %.1e frac<=1e-3 %.4f worst %s (bg min %.1f)" % (
        name, label, np.median(w), np.percentile(w, 95), np.percentile(w, 99), w.max(), (w <= 1e-3).mean(), names[pid[w.argmax()]], r.min()), flush=True)


ref = run("dopri", 4)
tight = run("rk4", 48)
rep("dopri", ref, tight, "vs tight")
for integ in ("split", "split_adaptive"):
    o = run(integ, 4)
    rep(integ, o, ref, "vs dopri"); rep(integ, o, tight, "vs tight")
    low = ref.min(0) < 20
    if low.any():
        w = np.abs(o - ref).max(0)
        print("  envs reaching BG < 20: %d; among them vs dopri max %.1e median %.1e; others max %.1e" % (low.sum(), w[low].max(), np.median(w[low]), w[~low].max() if (~low).any() else 0))
