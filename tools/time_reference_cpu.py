#!/usr/bin/env python3
"""Time the REFERENCE's own CPU path -- build container only.

TEST / MEASUREMENT INFRASTRUCTURE.  Imports the read-only reference checkout at /root/reference by absolute path (as
oracle/gen_golden.py does: an empty parent package stands in for simglucose/__init__.py, which needs gym) and times
`T1DSimEnv.step` (simulation/env.py:66-117) with a random-action policy on adult#001, Navigator (1-min) and Dexcom
(3-min) sensors, on one core and fanned over every core with one env per process -- the shape of the reference's own
`batch_sim(parallel=True)` (simulation/sim_engine.py:65-76).  The reference never travels: only the numbers are kept
(profiles/<round>/reference_cpu.json, quoted in BASELINE.md section 2 and attached by bench.py as
cpu_baseline.reference_python).  Exits 0 with a note when /root/reference is absent (GPU box).

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tools/time_reference_cpu.py [--steps 2000] [--out FILE]
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time
import types
import warnings
from datetime import datetime

REF = os.environ.get("T1D_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
warnings.simplefilter("ignore")
os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True


def _import_reference():
    import logging
    pkg = types.ModuleType("simglucose")
    pkg.__path__ = [os.path.join(REF, "simglucose")]
    pkg.__file__ = os.path.join(REF, "simglucose", "__init__.py")
    sys.modules["simglucose"] = pkg
    logging.disable(logging.CRITICAL)


def _run(args):
    """one process: one env, `warm` untimed + `steps` timed env.step calls -> (patient-minutes, seconds)"""
    sensor, steps, warm, seed = args
    _import_reference()
    import numpy as np
    from simglucose.patient.t1dpatient import T1DPatient
    from simglucose.sensor.cgm import CGMSensor
    from simglucose.actuator.pump import InsulinPump
    from simglucose.simulation.env import T1DSimEnv
    from simglucose.simulation.scenario import CustomScenario
    from simglucose.controller.base import Action
    start = datetime(2018, 1, 1, 0, 0, 0)
    patient = T1DPatient.withName("adult#001")
    sen = CGMSensor.withName(sensor, seed=seed)
    env = T1DSimEnv(patient, sen, InsulinPump.withName("Insulet"), CustomScenario(start_time=start, scenario=[(7, 45), (12, 70), (18, 80)]))
    env.reset()
    basal = patient._params.u2ss * patient._params.BW / 6000.0
    rs = np.random.RandomState(seed)
    for _ in range(warm):
        env.step(Action(basal=basal * rs.uniform(0, 2), bolus=0))
    t0 = time.perf_counter()
    for _ in range(steps):
        env.step(Action(basal=basal * rs.uniform(0, 2), bolus=0))
    return steps * int(sen.sample_time), time.perf_counter() - t0


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03", "reference_cpu.json"))
    a = ap.parse_args()
    if not os.path.isdir(os.path.join(REF, "simglucose")):
        print("reference checkout %s absent: nothing timed (this script runs in the build container only)" % REF)
        return
    cores = len(os.sched_getaffinity(0))
    import numpy, scipy, pandas
    out = {"what": "the reference's own T1DSimEnv.step (simulation/env.py:66-117; scipy dopri5 re-entered per minute), adult#001, "
                   "random basal U(0,2) x steady state each step, meals 45/70/80 g at 7/12/18 h; unit = patient-minutes (env-steps at 1-min dt) per second",
           "where": "build container, %d vCPU %s" % (cores, cpu_model()), "cores_available": cores,
           "versions": {"python": sys.version.split()[0], "numpy": numpy.__version__, "scipy": scipy.__version__, "pandas": pandas.__version__},
           "steps_timed": a.steps, "warmup_steps": a.warmup, "rows": []}
    for sensor in ("Navigator", "Dexcom"):
        done, dt = _run((sensor, a.steps, a.warmup, 1))
        row = {"sensor": sensor, "one_core": {"env_steps_per_s": done / dt, "gym_steps_per_s": a.steps / dt, "seconds": dt, "cores": 1}}
        t0 = time.perf_counter()
        with mp.get_context("fork").Pool(cores) as pool:
            res = pool.map(_run, [(sensor, a.steps, a.warmup, 10 + k) for k in range(cores)])
        wall = time.perf_counter() - t0
        row["all_cores"] = {"env_steps_per_s": sum(r[0] for r in res) / max(r[1] for r in res), "cores": cores, "processes": cores,
                            "seconds_slowest_process": max(r[1] for r in res), "wall_incl_startup_s": wall}
        out["rows"].append(row)
        print(json.dumps(row))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
