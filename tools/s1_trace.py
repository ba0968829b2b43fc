#!/usr/bin/env python3
"""Tuning aid (needs a -DT1D_S1_TRACE=1 build in T1D_LIB_PATH): per-phase wall-clock of sampled waves of step1_kernel."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.ab_step import make  # noqa: E402
from simglucose_amd import _lib  # noqa: E402
n = 1 << 20
env, pool = make(n, "mod30", torch.float64, "Navigator", 4)
env.set_option("integrator", 1); env.set_option("params_mode", int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for k in range(5):
    env.step(pool[k % 4])
torch.cuda.synchronize()
out = np.zeros(96 * 4 * 64, dtype=np.int64)
L = _lib.lib()
L.t1d_debug_trace.argtypes = [C.c_void_p, C.c_void_p]
assert L.t1d_debug_trace(env._ctx, out.ctypes.data_as(C.c_void_p)) == 0
tr = out.reshape(96 * 4, 8, 8).astype(np.float64) * 0.01      # us (100 MHz)
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
