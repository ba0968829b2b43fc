"""ctypes/numpy front-end of the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; nothing under ``simglucose_amd/`` does.  The numerics live in ``t1d_oracle.c``
(each function cites the reference file:line it restates); this file only marshals numpy
arrays, builds the parameter rows from the shipped CSV data tables and restates the two small
host-side pieces of the path that are easier to state in numpy:

* the not-a-knot cubic-spline block operator ``W`` behind ``scipy.interpolate.interp1d(kind=
  'cubic')`` in ``sensor/noise_gen.py:45-47`` (pinned by ``tests/golden/g4_sensor.npz``);
* ``RandomScenario.create_scenario`` / ``get_action`` (``simulation/scenario_gen.py:15-60``) and
  ``CustomScenario.get_action`` (``simulation/scenario.py:33-59``) as per-minute CHO arrays
  (pinned by ``tests/golden/g9_seeding.npz``).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PARAMS = os.path.join(os.path.dirname(_HERE), "simglucose_amd", "params")

PAR_COLS = ["BW", "kabs", "kmax", "kmin", "b", "d", "Vg", "Vi", "Vmx", "Km0", "k2", "k1", "p2u",
            "m1", "m2", "m4", "m30", "Ib", "ki", "kp2", "kp3", "f", "ke1", "ke2", "Fsnc", "Vm0",
            "kd", "ksc", "ka1", "ka2", "kp1", "u2ss"]
NPAR = 13 + len(PAR_COLS)
IDX = {name: 13 + i for i, name in enumerate(PAR_COLS)}
SENSOR_COLS = ["PACF", "gamma", "lambda", "delta", "xi", "sample_time", "min", "max"]
PUMP_COLS = ["min_bolus", "max_bolus", "inc_bolus", "min_basal", "max_basal", "inc_basal"]


def build(force=False):
    so = os.path.join(_HERE, "libt1d_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("t1d_oracle.c", "t1d_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libt1d_oracle.so"])
    return so


def _read_csv(path):
    import csv
    with open(path, newline="") as f:
        rows = list(csv.reader(f))
    return rows[0], rows[1:]


def patient_table():
    """-> (names, table[30, NPAR]) in the column order of t1d_oracle.h."""
    hdr, rows = _read_csv(os.path.join(_PARAMS, "vpatient_params.csv"))
    col = {h: i for i, h in enumerate(hdr)}
    names = [r[col["Name"]] for r in rows]
    tab = np.empty((len(rows), NPAR))
    for i, r in enumerate(rows):
        tab[i, :13] = [float(v) for v in r[2:15]]
        tab[i, 13:] = [float(r[col[c]]) for c in PAR_COLS]
    return names, tab


def sensor_row(name):
    hdr, rows = _read_csv(os.path.join(_PARAMS, "sensor_params.csv"))
    r = next(r for r in rows if r[0] == name)
    return np.array([float(r[hdr.index(c)]) for c in SENSOR_COLS])


def pump_row(name):
    hdr, rows = _read_csv(os.path.join(_PARAMS, "pump_params.csv"))
    r = next(r for r in rows if r[0] == name)
    return np.array([float(r[hdr.index(c)]) for c in PUMP_COLS])


def spline_block_operator(sample_time):
    """W[(floor(150/st)) x 11]: noise block samples = W @ (11 points at 15-min spacing).

    Not-a-knot cubic spline through 11 equally spaced knots (what interp1d(kind='cubic')
    builds), evaluated at t = st, 2 st, ... (first sample dropped, noise_gen.py:47).
    """
    h, K = 15.0, 11
    A = np.zeros((K, K)); B = np.zeros((K, K))
    for k in range(1, K - 1):
        A[k, k - 1:k + 2] = (1.0, 4.0, 1.0)
        B[k, k - 1:k + 2] = np.array([1.0, -2.0, 1.0]) * 6.0 / h ** 2
    A[0, 0:3] = (1.0, -2.0, 1.0)
    A[K - 1, K - 3:K] = (1.0, -2.0, 1.0)
    M = np.linalg.solve(A, B)                     # second derivatives = M @ y
    nsample = int(np.floor(10 * 15 / sample_time)) + 1
    t = np.arange(1, nsample) * float(sample_time)
    W = np.zeros((len(t), K))
    for r, tt in enumerate(t):
        k = min(int(tt // h), K - 2)
        a, b = (k + 1) * h - tt, tt - k * h
        W[r] += M[k] * (a ** 3 / (6 * h) - a * h / 6) + M[k + 1] * (b ** 3 / (6 * h) - b * h / 6)
        W[r, k] += a / h
        W[r, k + 1] += b / h
    return W


# ----------------------------------------------------------------------------- ctypes structs
_d = C.POINTER(C.c_double); _i = C.POINTER(C.c_int32); _u = C.POINTER(C.c_uint8)


class _Batch(C.Structure):
    _fields_ = [("n", C.c_int32), ("w_rows", C.c_int32), ("ptab", _d), ("pid", _i), ("W", _d),
                ("normals", _d), ("sensor", C.c_double * 8), ("pump", C.c_double * 6),
                ("x", _d), ("planned", _d), ("last_qsto", _d), ("last_food", _d), ("was_eating", _u),
                ("t", _i), ("h_carry", _d), ("last_cgm", _d), ("ar_e", _d), ("pts", _d),
                ("n_samples", _i), ("n_draws", _i), ("prev_cgm", _d), ("split_tab", _d), ("split_stride", C.c_int32),
                ("level_count", C.POINTER(C.c_int64))]


class _Out(C.Structure):
    _fields_ = [(k, _d) for k in ("cgm", "bg", "reward", "lbgi", "hbgi", "risk", "meal", "insulin",
                                  "cgm_hist0")] + [("done", _u)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.t1d_o_rhs.argtypes = [_d, _d, C.c_double, C.c_double, C.c_double, C.c_double, _d]
        L.t1d_o_rk4_minute.argtypes = [_d, _d, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int]
        L.t1d_o_dopri5_minute.argtypes = [_d, _d, C.c_double, C.c_double, C.c_double, C.c_double, _d,
                                          C.c_double, C.c_double]
        L.t1d_o_dopri5_minute.restype = C.c_int
        L.t1d_o_pump.argtypes = [C.c_double] * 4; L.t1d_o_pump.restype = C.c_double
        L.t1d_o_risk.argtypes = [C.c_double, _d, _d, _d]
        L.t1d_o_reset.argtypes = [C.POINTER(_Batch), _d, C.POINTER(_Out)]
        L.t1d_o_step.argtypes = [C.POINTER(_Batch), _d, _d, _d, C.c_int, C.c_int, C.c_double, C.POINTER(_Out)]
        L.t1d_o_step.restype = C.c_int
        L.t1d_o_pid.argtypes = [_d, _d] + [C.c_double] * 6; L.t1d_o_pid.restype = C.c_double
        L.t1d_o_patient_minute.argtypes = [_d, _d, _d, _d, _d, _u, _d, C.c_int, C.c_double, C.c_double,
                                           C.c_int, C.c_int, C.c_double, _d]
        L.t1d_o_split_minute.argtypes = [_d, _d, _d, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
        L.t1d_o_split_minute.restype = C.c_int
        L.t1d_o_patient_minute.restype = C.c_int
        L.t1d_o_set_knob.argtypes = [C.c_int, C.c_double]
        L.t1d_o_get_knob.argtypes = [C.c_int]; L.t1d_o_get_knob.restype = C.c_double
        _lib = L
    return _lib


def _p(a, typ=_d):
    return a.ctypes.data_as(typ)


def rhs(prow, x, cho, ins, lq, lf):
    prow = np.ascontiguousarray(prow, dtype=np.float64); x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty(13)
    lib().t1d_o_rhs(_p(prow), _p(x), cho, ins, lq, lf, _p(out))
    return out


def pump(amount, inc, lo, hi):
    return lib().t1d_o_pump(float(amount), float(inc), float(lo), float(hi))


def risk(bg):
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    lib().t1d_o_risk(float(bg), C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


DOPRI_BETA = 0.04     # what the Fortran driver uses when scipy hands it beta = 0.0
# integrator codes of t1d_oracle.c; "mr" = multirate RK4 with n_sub = ng * 1000 + ns
_INTEG = {"rk4": 0, "dopri": 1, "mr": 2, "split": 3, "split_adaptive": 4}


def split_tables(ptab, n_sub):
    """Host tables of the split scheme (t1d_oracle.c, integrators 3 and 4) for every row of `ptab`:
    [2 n_sub][7][9] exact propagators Phi(k / (2 n_sub)) of the linear insulin sub-system
    s = (x5, x9, x10, x11, x6, x7, x8) augmented with (u, 1)  (t1dpatient.py:176-198), from
    scipy.linalg.expm, then the ETD-RK4 weights (E, wa, wm, wb) of x2' = -kabs x2 + F for the gut steps of level 1
    (h = 1/n_sub) and of level 2 (h/2), by Gauss-Legendre quadrature of the quadratic interpolant against
    exp(-kabs (h-s))."""
    from scipy.linalg import expm
    ptab = np.atleast_2d(ptab)
    nb = 2 * n_sub
    out = np.zeros((ptab.shape[0], nb * 63 + 8))
    gx, gw = np.polynomial.legendre.leggauss(32)
    for r, p in enumerate(ptab):
        g = lambda k: p[IDX[k]]
        A = np.zeros((9, 9))
        A[0, 0] = -(g("m2") + g("m4")); A[0, 1] = g("m1"); A[0, 2] = g("ka1"); A[0, 3] = g("ka2")     # x5  :176
        A[1, 1] = -(g("m1") + g("m30")); A[1, 0] = g("m2")                                            # x9  :190
        A[2, 2] = -(g("ka1") + g("kd")); A[2, 7] = 1.0                                                # x10 :194
        A[3, 2] = g("kd"); A[3, 3] = -g("ka2")                                                        # x11 :197
        A[4, 4] = -g("p2u"); A[4, 0] = g("p2u") / g("Vi"); A[4, 8] = -g("p2u") * g("Ib")              # x6  :182
        A[5, 5] = -g("ki"); A[5, 0] = g("ki") / g("Vi")                                               # x7  :185
        A[6, 6] = -g("ki"); A[6, 5] = g("ki")                                                         # x8  :187
        for k in range(1, nb + 1):
            out[r, (k - 1) * 63:k * 63] = expm(A * (k / nb))[:7].ravel()
        for part, hh in enumerate((1.0 / n_sub, 0.5 / n_sub)):                     # gut step of level 1, 2
            ss = (gx + 1.0) * hh / 2.0; ww = gw * hh / 2.0
            Lah = (ss - hh / 2) * (ss - hh) / ((0 - hh / 2) * (0 - hh))
            Lmh = (ss - 0) * (ss - hh) / ((hh / 2) * (hh / 2 - hh))
            Lbh = (ss - 0) * (ss - hh / 2) / (hh * (hh / 2))
            ker = np.exp(-g("kabs") * (hh - ss))
            out[r, nb * 63 + 4 * part:nb * 63 + 4 * part + 4] = (np.exp(-g("kabs") * hh), (ker * Lah * ww).sum(),
                                                                (ker * Lmh * ww).sum(), (ker * Lbh * ww).sum())
    return np.ascontiguousarray(out)


def set_knob(k, v):
    lib().t1d_o_set_knob(int(k), float(v))


class PatientOracle:
    """T1DPatient.step restated (patient only; no pump/sensor) -- pins G2."""

    def __init__(self, prow, x0=None):
        self._split = {}
        self.p = np.ascontiguousarray(prow, dtype=np.float64)
        self.x = np.array(self.p[:13] if x0 is None else x0, dtype=np.float64)
        self.planned = C.c_double(0.0); self.lq = C.c_double(self.x[0] + self.x[1]); self.lf = C.c_double(0.0)
        self.eat = C.c_uint8(0); self.h = C.c_double(0.0); self.t = 0
        self.nfcn = 0

    def step(self, meal, insulin, integrator="rk4", n_sub=4, beta=DOPRI_BETA):
        tab = None
        if integrator in ("split", "split_adaptive"):
            if n_sub not in self._split:
                self._split[n_sub] = split_tables(self.p, n_sub)[0]
            tab = _p(self._split[n_sub])
        r = lib().t1d_o_patient_minute(_p(self.p), _p(self.x), C.byref(self.planned), C.byref(self.lq),
                                       C.byref(self.lf), C.byref(self.eat), C.byref(self.h), self.t,
                                       float(meal), float(insulin), _INTEG[integrator],
                                       int(n_sub), float(beta), tab)
        if r < 0:
            raise RuntimeError("oracle integrator failed")
        self.nfcn = r
        self.t += 1
        return self.x


class OracleEnv:
    """Batch of T1DSimEnv restated on the CPU (env index fastest, fp64)."""

    OUT_KEYS = ("cgm", "bg", "reward", "lbgi", "hbgi", "risk", "meal", "insulin", "cgm_hist0")

    def __init__(self, patient_idx, sensor="Dexcom", pump="Insulet", normals=None, n_draws_max=256,
                 integrator="rk4", n_sub=4, beta=DOPRI_BETA, sensor_row_override=None, ptab_override=None):
        self.names, self.ptab = patient_table()
        if ptab_override is not None:
            self.ptab = np.ascontiguousarray(ptab_override, dtype=np.float64)
        self.pid = np.ascontiguousarray(patient_idx, dtype=np.int32)
        n = self.n = len(self.pid)
        self.sensor = np.array(sensor_row_override if sensor_row_override is not None else sensor_row(sensor))
        self.pump = pump_row(pump)
        self.W = np.ascontiguousarray(spline_block_operator(self.sensor[5]))
        self.normals = np.ascontiguousarray(
            normals if normals is not None else np.zeros((n_draws_max, n)), dtype=np.float64)
        assert self.normals.shape[1] == n
        self.integrator, self.n_sub, self.beta = integrator, n_sub, beta
        z = lambda *s: np.zeros(s)
        self.x = z(13, n); self.planned = z(n); self.last_qsto = z(n); self.last_food = z(n)
        self.was_eating = np.zeros(n, np.uint8); self.t = np.zeros(n, np.int32); self.h_carry = z(n)
        self.last_cgm = z(n); self.ar_e = z(n); self.pts = z(11, n)
        self.n_samples = np.zeros(n, np.int32); self.n_draws = np.zeros(n, np.int32); self.prev_cgm = z(n)
        self.out = {k: z(n) for k in self.OUT_KEYS}; self.out["done"] = np.zeros(n, np.uint8)
        b = self._b = _Batch()
        b.n, b.w_rows = n, self.W.shape[0]
        b.ptab, b.pid, b.W, b.normals = _p(self.ptab), _p(self.pid, _i), _p(self.W), _p(self.normals)
        for k in range(8): b.sensor[k] = self.sensor[k]
        for k in range(6): b.pump[k] = self.pump[k]
        for k in ("x", "planned", "last_qsto", "last_food", "h_carry", "last_cgm", "ar_e", "pts", "prev_cgm"):
            setattr(b, k, _p(getattr(self, k)))
        if integrator in ("split", "split_adaptive"):
            self.split_tab = split_tables(self.ptab, n_sub)
            b.split_tab = _p(self.split_tab); b.split_stride = self.split_tab.shape[1]
        self.level_count = np.zeros(3, np.int64)
        b.level_count = self.level_count.ctypes.data_as(C.POINTER(C.c_int64))
        b.was_eating = _p(self.was_eating, _u); b.t = _p(self.t, _i)
        b.n_samples = _p(self.n_samples, _i); b.n_draws = _p(self.n_draws, _i)
        o = self._o = _Out()
        for k in self.OUT_KEYS: setattr(o, k, _p(self.out[k]))
        o.done = _p(self.out["done"], _u)

    @property
    def sample_time(self):
        return self.sensor[5]

    def reset(self, x0=None):
        x0p = None
        if x0 is not None:
            self._x0 = np.ascontiguousarray(x0, dtype=np.float64); assert self._x0.shape == (13, self.n)
            x0p = _p(self._x0)
        lib().t1d_o_reset(C.byref(self._b), x0p, C.byref(self._o))
        return {k: v.copy() for k, v in self.out.items()}

    def step(self, basal, bolus=None, cho=None):
        n = self.n
        basal = np.ascontiguousarray(np.broadcast_to(np.asarray(basal, dtype=np.float64), (n,)))
        bp = None
        if bolus is not None:
            bolus = np.ascontiguousarray(np.broadcast_to(np.asarray(bolus, dtype=np.float64), (n,))); bp = _p(bolus)
        cp = None
        if cho is not None:
            cho = np.ascontiguousarray(cho, dtype=np.float64); assert cho.shape == (int(self.sample_time), n)
            cp = _p(cho)
        rc = lib().t1d_o_step(C.byref(self._b), _p(basal), bp, cp, _INTEG[self.integrator],
                              int(self.n_sub), float(self.beta), C.byref(self._o))
        if rc != 0:
            raise RuntimeError("oracle DOPRI5 failed")
        return {k: v.copy() for k, v in self.out.items()}


# ----------------------------------------------------------------------------- scenarios (M1)
def custom_scenario_cho(hours, grams, n_minutes):
    """CustomScenario with numeric times = hours after start (scenario.py:33-59): CHO per minute.
    First match wins when two entries round to the same minute (list.index, scenario.py:40)."""
    cho = np.zeros(n_minutes)
    seen = set()
    for h, g in zip(hours, grams):
        m = int(round(h * 60.0))
        if m in seen:
            continue
        seen.add(m)
        if 0 <= m < n_minutes:
            cho[m] = g
    return cho


def random_scenario_draw(rs):
    """One RandomScenario.create_scenario() draw from a numpy RandomState (scenario_gen.py:33-60).
    scipy.stats.truncnorm.rvs draws one uniform from the same RandomState and maps it through
    the truncated-normal inverse CDF."""
    from scipy.stats import truncnorm
    prob = [0.95, 0.3, 0.95, 0.3, 0.95, 0.3]
    lb = np.array([5, 9, 10, 14, 16, 20]) * 60; ub = np.array([9, 10, 14, 16, 20, 23]) * 60
    mu = np.array([7, 9.5, 12, 15, 18, 21.5]) * 60; sd = np.array([60, 30, 60, 30, 60, 30])
    amu = [45, 10, 70, 10, 80, 10]; asd = [10, 5, 10, 5, 10, 5]
    times, amounts = [], []
    for k in range(6):
        if rs.rand() < prob[k]:
            t = np.round(truncnorm.rvs(a=(lb[k] - mu[k]) / sd[k], b=(ub[k] - mu[k]) / sd[k], loc=mu[k],
                                       scale=sd[k], random_state=rs))
            times.append(float(t))
            amounts.append(float(max(round(rs.normal(amu[k], asd[k])), 0)))
    return times, amounts


def random_scenario_cho(seed, start_minute_of_day, n_minutes):
    """RandomScenario(start_time, seed) after reset(): per-minute announced CHO (scenario_gen.py:15-31).
    A fresh day is drawn at reset and again whenever the clock reads 00:00 (quirk 4)."""
    rs = np.random.RandomState(seed)
    times, amounts = random_scenario_draw(rs)
    cho = np.zeros(n_minutes)
    for m in range(n_minutes):
        tod = (start_minute_of_day + m) % 1440
        if tod == 0:
            times, amounts = random_scenario_draw(rs)
        if float(tod) in times:
            cho[m] = amounts[times.index(float(tod))]
    return cho


# ----------------------------------------------------------------------------- controllers (C1, C2)
def quest_table():
    hdr, rows = _read_csv(os.path.join(_PARAMS, "Quest.csv"))
    return {r[0]: {h: float(v) for h, v in zip(hdr[1:], r[1:])} for r in rows}


def bb_policy(name, meal, glucose, sample_time, target=140.0):
    """BBController._bb_policy (controller/basal_bolus_ctrller.py:34-80) -> (basal, bolus) U/min."""
    names, tab = patient_table()
    q = quest_table()[name]
    row = tab[names.index(name)]
    basal = row[IDX["u2ss"]] * row[IDX["BW"]] / 6000.0                        # :64
    if meal > 0:
        bolus = (meal * sample_time) / q["CR"] + (glucose > 150) * (glucose - target) / q["CF"]   # :69-71
    else:
        bolus = 0.0
    return basal, bolus / sample_time                                         # :79


def closed_loop(patient_name, sensor, sensor_seed, scen_seed, n_steps, policy, start_minute_of_day=0,
                integrator="dopri", n_sub=4):
    """SimObj.simulate (simulation/sim_engine.py:29-39) for one env on the oracle: reset, then
    n_steps x (policy(obs, info) -> env.step).  `policy(cgm, info) -> (basal, bolus)`.
    Returns the show_history() columns (simulation/env.py:169-180) as a dict of arrays (n_steps+1
    rows for BG/CGM/LBGI/HBGI/Risk, n_steps for CHO/insulin)."""
    names, _ = patient_table()
    st = int(sensor_row(sensor)[5])
    z = np.random.RandomState(sensor_seed).randn(1 + 10 * (2 + (n_steps * st) // 150))
    env = OracleEnv([names.index(patient_name)], sensor=sensor, normals=z[:, None], integrator=integrator, n_sub=n_sub)
    cho = random_scenario_cho(scen_seed, start_minute_of_day, n_steps * st)
    r = env.reset()
    hist = {"BG": [r["bg"][0]], "CGM": [r["cgm_hist0"][0]], "LBGI": [r["lbgi"][0]], "HBGI": [r["hbgi"][0]],
            "Risk": [r["risk"][0]], "CHO": [], "insulin": []}
    obs, info = r["cgm"][0], {"meal": 0.0, "sample_time": float(st), "patient_name": patient_name}
    actions = []
    for k in range(n_steps):
        basal, bolus = policy(obs, info)
        actions.append((basal, bolus))
        o = env.step(basal, bolus, cho[k * st:(k + 1) * st, None])
        obs = o["cgm"][0]
        info["meal"] = o["meal"][0]
        for key, src in (("BG", "bg"), ("CGM", "cgm"), ("LBGI", "lbgi"), ("HBGI", "hbgi"), ("Risk", "risk"),
                         ("CHO", "meal"), ("insulin", "insulin")):
            hist[key].append(o[src][0])
    return {k: np.array(v) for k, v in hist.items()}, np.array(actions)


# ----------------------------------------------------------------------------- analysis/report.py (numbers only)
def report_percent_stats(BG):
    """percent_stats (analysis/report.py:74-92) on BG [rows, envs] -> [5, envs] percentages in the order
    BG>180, BG<70, 70<=BG<=180, BG>250, BG<50.  Pinned by fixture G11."""
    BG = np.asarray(BG, dtype=np.float64); n = float(len(BG))
    return np.stack([(BG > 180).sum(0), (BG < 70).sum(0), ((BG >= 70) & (BG <= 180)).sum(0), (BG > 250).sum(0),
                     (BG < 50).sum(0)]) / n * 100.0


def report_cvga(BG):
    """CVGA_analysis (analysis/report.py:198-217) -> BG_min, BG_max (clipped 2.5th / 97.5th percentiles per env),
    zone fractions (A, B, C, D, E) and the per-env zone code 0..4 (5 = none).  Pinned by fixture G11."""
    BG = np.asarray(BG, dtype=np.float64)
    mn = np.clip(np.percentile(BG, 2.5, axis=0), 50, 400); mx = np.clip(np.percentile(BG, 97.5, axis=0), 50, 400)
    A = (mn > 90) & (mn <= 110) & (mx >= 110) & (mx < 180)
    B = (mn > 70) & (mn <= 110) & (mx >= 110) & (mx < 300)
    Cz = ((mn > 90) & (mn <= 110) & (mx >= 300)) | ((mn <= 70) & (mx >= 110) & (mx < 180))
    D = ((mn > 70) & (mn <= 90) & (mx >= 300)) | ((mn <= 70) & (mx >= 180) & (mx < 300))
    E = (mn <= 70) & (mx >= 300)
    m = float(len(mn))
    frac = np.array([A.sum() / m, B.sum() / m - A.sum() / m, Cz.sum() / m, D.sum() / m, E.sum() / m])
    zone = np.where(A, 0, np.where(B, 1, np.where(Cz, 2, np.where(D, 3, np.where(E, 4, 5)))))
    return mn, mx, frac, zone


def report_risk_index_trace(BG, chunk=60):
    """risk_index_trace (analysis/report.py:95-110): per chunk of `chunk` rows and per env, f = mean of
    1.509 (ln(BG)^1.084 - 5.381) over the rows where that is a number -- the reference masks BG <= 0 (`BG[BG > 0]` on a
    DataFrame leaves NaN there), 0 < BG < 1 gives NaN through the power of a negative logarithm, and pandas' mean skips
    NaN; LBGI = 10 (f (f<0))^2, HBGI = 10 (f (f>0))^2 -> (LBGI, HBGI) each [n_chunks, envs], NaN where a chunk has no
    usable row.  Pinned by fixture G12 (the reference's own 2017 result files) to 1e-13."""
    BG = np.asarray(BG, dtype=np.float64)
    L, H = [], []
    for i in range(0, len(BG), chunk):
        c = BG[i:i + chunk]
        with np.errstate(invalid="ignore", divide="ignore"):
            v = 1.509 * (np.log(np.where(c > 0, c, np.nan)) ** 1.084 - 5.381)
            cnt = (~np.isnan(v)).sum(0)
            f = np.where(cnt > 0, np.nansum(v, 0) / np.maximum(cnt, 1), np.nan)
        L.append(10.0 * (f * (f < 0)) ** 2); H.append(10.0 * (f * (f > 0)) ** 2)
    return np.array(L), np.array(H)
