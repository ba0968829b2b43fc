#!/usr/bin/env python3
"""Tuning aid (needs a -DT1D_S1_TRACE=1 build in T1D_LIB_PATH): per-phase wall-clock of sampled waves of step1_kernel."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simglucose_amd import _lib, params, scenario_batch  # noqa: E402
from simglucose_amd.batch_env import BatchedT1DSimEnv  # noqa: E402
n = 1 << 20
pid = np.arange(n) % 30
env = BatchedT1DSimEnv(patient=pid, sensor="Navigator", n_sub=4, seed=1, extra_outputs=False)
env.set_option("adaptive_gut", int(sys.argv[1]) if len(sys.argv) > 1 else 0)       # the trace marks live in step1_kernel
mt, ma = scenario_batch.random_meal_tables(n, days=1, seed=5, device=env.device)
env.set_meals(mt, ma); env.reset()
_, tab = params.patient_table()
b0 = torch.as_tensor(tab[pid, params.P_COL["u2ss"]] * tab[pid, params.P_COL["BW"]] / 6000.0, device=env.device)
pool = [(b0 * 2.0 * torch.rand(n, device=env.device, dtype=torch.float64)).contiguous() for _ in range(4)]
for k in range(5):
    env.step(pool[k % 4])
torch.cuda.synchronize()
out = np.zeros(128 * 4 * 64, dtype=np.int64)
L = _lib.lib()
L.t1d_debug_trace.argtypes = [C.c_void_p, C.c_void_p]
assert L.t1d_debug_trace(env._ctx, out.ctypes.data_as(C.c_void_p)) == 0
tr = out[:96 * 4 * 64].reshape(96 * 4, 8, 8).astype(np.float64) * 0.01      # us (100 MHz)
t0 = tr[:, 0, 0].min()
names = ["load wait", "prologue (pump, meal, eat, early stores)", "integration", "x stores issued + sensor loads arrive", "epilogue compute", "stores drain"]
valid = tr[:, :, 6] > 0
print("tiles per wave:", valid.sum(1).min(), "-", valid.sum(1).max(), " kernel span %.1f us" % (tr[:, :, 6].max() - t0))
for m in range(6):
    d = (tr[:, :, m + 1] - tr[:, :, m])[valid]
    print("%-45s mean %6.2f us  p10 %6.2f  p90 %6.2f" % (names[m], d.mean(), np.percentile(d, 10), np.percentile(d, 90)))
tile = (tr[:, :, 6] - tr[:, :, 0])[valid]
print("whole tile: mean %.2f us;  first-tile start spread %.2f us" % (tile.mean(), tr[:, 0, 0].max() - t0))
for w in (0, 4, 8, 13, 17, 21):       # waves w, w + 4, w + 8 of a workgroup share a SIMD
    for m, nm in ((0, "chunk start"), (2, "integ start"), (3, "integ end"), (6, "chunk end")):
        print("block %3d wave %2d %-11s" % (w // 12, w % 12, nm), np.round(tr[w, :, m][valid[w]] - t0, 1))
