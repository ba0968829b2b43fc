#!/usr/bin/env python3
"""A/B of two builds of the library inside ONE process on ONE box (GPU box): the headline workload (1 Mi envs, fp64, one
minute per launch) on an env of each build, the two taking turns over several rounds, so that box-to-box and warm-up
differences cancel.  usage: ab_two_libs.py other_lib.so [rounds] [steps]   (the first env uses the in-tree library)"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simglucose_amd import _lib, params, scenario_batch  # noqa: E402
from simglucose_amd.batch_env import BatchedT1DSimEnv  # noqa: E402
other = sys.argv[1]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 450          # a multiple of 150: every round holds the same number of block builds
n = 1 << 20
pid = np.arange(n) % 30
_, tab = params.patient_table()


def make():
    env = BatchedT1DSimEnv(patient=pid, sensor="Navigator", n_sub=4, seed=1234, extra_outputs=False)
    g0 = torch.Generator(device="cpu"); g0.manual_seed(99)
    start = torch.randint(0, 1440, (n,), generator=g0, dtype=torch.int32).to(env.device)
    mt, ma = scenario_batch.random_meal_tables(n, days=14, start_minute_of_day=start, seed=1000, device=env.device)
    env.set_meals(mt, ma)
    env.reset()
    return env


envs = {"in-tree": make()}
L0 = _lib._lib
_lib._lib = None
_lib.LIB_PATH = os.path.abspath(other)
_lib._stale = lambda: False
envs[os.path.basename(other)] = make()
b0 = torch.as_tensor(tab[pid, params.P_COL["u2ss"]] * tab[pid, params.P_COL["BW"]] / 6000.0, device="cuda:0")
g = torch.Generator(device="cpu"); g.manual_seed(7)
pool = [(b0 * 2.0 * torch.rand(n, generator=g, dtype=torch.float64).to("cuda:0")).contiguous() for _ in range(8)]
for env in envs.values():
    for k in range(600):
        env.step(pool[k % 8])
res = {k: [] for k in envs}
for r in range(rounds):
    for name, env in envs.items():
        for k in range(30):
            env.step(pool[k % 8])
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for k in range(steps - 30):
            env.step(pool[k % 8])
        e.record(); torch.cuda.synchronize()
        res[name].append(s.elapsed_time(e) / (steps - 30) * 1e3)
for name, v in res.items():
    print("%-24s median %.2f us per step incl. block builds  (%s)  status %d" % (name, float(np.median(v)), " ".join("%.1f" % x for x in v), envs[name].sync(raise_on_status=False)))
