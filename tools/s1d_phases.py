#!/usr/bin/env python3
"""Tuning aid (needs a -DT1D_S1_TRACE=1 build in T1D_LIB_PATH): when the waves of step1d_kernel's first 32 workgroups
reach the phase boundaries of a launch -- main pass done, every chunk past its decision point, list pass done -- and how
long the lists were; then the per-chunk phase times of the main pass.   usage: s1d_phases.py [envs] [dtype f64|f32]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simglucose_amd import _lib, params, scenario_batch  # noqa: E402
from simglucose_amd.batch_env import BatchedT1DSimEnv  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
dt = torch.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else torch.float64
pid = np.arange(n) % 30
env = BatchedT1DSimEnv(patient=pid, sensor="Navigator", n_sub=4, seed=1, extra_outputs=False, dtype=dt)
start = torch.randint(0, 1440, (n,), device=env.device, dtype=torch.int32)
mt, ma = scenario_batch.random_meal_tables(n, days=2, start_minute_of_day=start, seed=5, device=env.device, dtype=dt)
env.set_meals(mt, ma); env.reset()
_, tab = params.patient_table()
b0 = torch.as_tensor(tab[pid, params.P_COL["u2ss"]] * tab[pid, params.P_COL["BW"]] / 6000.0, device=env.device, dtype=dt)
pool = [(b0 * 2.0 * torch.rand(n, device=env.device, dtype=dt)).contiguous() for _ in range(4)]
for k in range(600):
    env.step(pool[k % 4])
torch.cuda.synchronize()
out = np.zeros(128 * 4 * 64, dtype=np.int64)
L = _lib.lib()
L.t1d_debug_trace.argtypes = [C.c_void_p, C.c_void_p]
assert L.t1d_debug_trace(env._ctx, out.ctypes.data_as(C.c_void_p)) == 0
nw = int(os.environ.get('T1D_S1_WAVES', '4' if dt == torch.float32 else '3')) * 4
tr = out[:32 * nw * 64].reshape(32, nw, 64)
t = tr[:, :, :5].astype(np.float64) * 0.01                    # us (100 MHz)
t0 = t[:, :, 0].min()
names = ["start", "main pass done (wave)", "all chunks decided", "list pass done"]
for k in range(4):
    v = t[:, :, k] - t0
    print("%-26s mean %7.2f  min %7.2f  max %7.2f us" % (names[k], v.mean(), v.min(), v.max()))
print("per workgroup: last wave out   mean %.2f max %.2f us" % ((t[:, :, 3].max(1) - t0).mean(), (t[:, :, 3].max() - t0)))
print("listed envs per CU: mean %.1f max %d;  chunks drawn per wave %d-%d" % (tr[:, 0, 6].mean(), tr[:, 0, 6].max(), tr[:, :, 8].min(), tr[:, :, 8].max()))
print("main pass done, last wave of workgroups 0..31 (us):", np.round(t[:, :, 1].max(1) - t0, 1))
b = 0
print("workgroup 0, per wave (us since start): main done / decided / list pass done")
for w in range(nw):
    print("  wave %2d  " % w, " ".join("%7.2f" % (t[b, w, k] - t0) for k in range(1, 4)))

# per-chunk anatomy of the main pass (the first six chunks of every sampled wave; marks drain the memory counters)
ck = tr[:, :, 16:64].reshape(32, nw, 6, 8).astype(np.float64) * 0.01
ok = (ck[..., 6] > 0) & (ck[..., 0] >= t0)            # (marks older than this launch belong to earlier launches)
names = ["loads arrive", "pump, meal, step-size rule, early stores", "integration", "x stores issued + sensor loads arrive", "epilogue compute", "stores drain"]
for m in range(6):
    d = (ck[..., m + 1] - ck[..., m])[ok]
    print("%-42s mean %6.2f us  p10 %6.2f  p90 %6.2f" % (names[m], d.mean(), np.percentile(d, 10), np.percentile(d, 90)))
print("whole chunk mean %.2f us over %d chunks" % ((ck[..., 6] - ck[..., 0])[ok].mean(), ok.sum()))

good = ok & np.all(np.diff(ck[..., :7], axis=-1) >= 0, axis=-1)
for k in range(6):
    g = good[:, :, k]
    if g.sum():
        d = np.diff(ck[:, :, k, :7], axis=-1)[g]
        print("chunk #%d of a wave (%4d): start %6.1f us after launch; " % (k, g.sum(), (ck[:, :, k, 0][g] - t0).mean()) + "  ".join("%5.2f" % v for v in d.mean(0)) + "  = %.2f us" % d.sum(1).mean())
