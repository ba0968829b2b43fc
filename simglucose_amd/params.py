"""Constant tables of the simulator (the shipped copies of the reference's data files
``simglucose/params/{vpatient_params,sensor_params,pump_params,Quest}.csv``) in the column order
``include/t1d.h`` declares, and the cubic-spline block operator of the CGM noise model."""
import csv
import os

import numpy as np

PARAMS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "params")
PATIENT_PARA_FILE = os.path.join(PARAMS_DIR, "vpatient_params.csv")
SENSOR_PARA_FILE = os.path.join(PARAMS_DIR, "sensor_params.csv")
INSULIN_PUMP_PARA_FILE = os.path.join(PARAMS_DIR, "pump_params.csv")
CONTROL_QUEST = os.path.join(PARAMS_DIR, "Quest.csv")

# T1D_P_* (include/t1d.h): 13 initial-state columns, then these model parameters
MODEL_COLS = ("BW", "kabs", "kmax", "kmin", "b", "d", "Vg", "Vi", "Vmx", "Km0", "k2", "k1", "p2u", "m1",
              "m2", "m4", "m30", "Ib", "ki", "kp2", "kp3", "f", "ke1", "ke2", "Fsnc", "Vm0", "kd", "ksc",
              "ka1", "ka2", "kp1", "u2ss")
P_COL = {name: 13 + i for i, name in enumerate(MODEL_COLS)}
SENSOR_COLS = ("PACF", "gamma", "lambda", "delta", "xi", "sample_time", "min", "max")
PUMP_COLS = ("min_bolus", "max_bolus", "inc_bolus", "min_basal", "max_basal", "inc_basal")


def _rows(path):
    with open(path, newline="") as f:
        r = list(csv.reader(f))
    return r[0], r[1:]


_cache = {}


def patient_table():
    """-> (names list[30], table float64 [30, 45]) rows in file order (adolescent, adult, child)."""
    if "patients" not in _cache:
        hdr, rows = _rows(PATIENT_PARA_FILE)
        col = {h: i for i, h in enumerate(hdr)}
        tab = np.empty((len(rows), 13 + len(MODEL_COLS)))
        for i, r in enumerate(rows):
            tab[i, :13] = [float(v) for v in r[2:15]]          # x0_ 1 .. x0_13 (t1dpatient.py:252)
            tab[i, 13:] = [float(r[col[c]]) for c in MODEL_COLS]
        _cache["patients"] = ([r[col["Name"]] for r in rows], tab)
    names, tab = _cache["patients"]
    return list(names), tab.copy()


def patient_index(name):
    names, _ = patient_table()
    try:
        return names.index(name)
    except ValueError:
        raise ValueError("unknown patient %r" % (name,)) from None


def _named_row(path, cols, name):
    hdr, rows = _rows(path)
    for r in rows:
        if r[0] == name:
            return np.array([float(r[hdr.index(c)]) for c in cols])
    raise ValueError("unknown name %r in %s" % (name, os.path.basename(path)))


def sensor_row(name):
    return _named_row(SENSOR_PARA_FILE, SENSOR_COLS, name)


def pump_row(name):
    return _named_row(INSULIN_PUMP_PARA_FILE, PUMP_COLS, name)


def quest_table():
    """-> dict name -> (CR, CF, Age, TDI) from Quest.csv (basal_bolus_ctrller.py:22)."""
    hdr, rows = _rows(CONTROL_QUEST)
    return {r[0]: tuple(float(r[hdr.index(c)]) for c in ("CR", "CF", "Age", "TDI")) for r in rows}


def basal_rate(table_row):
    """u2ss * BW / 6000 U/min: the steady-state basal (basal_bolus_ctrller.py:64)."""
    return table_row[P_COL["u2ss"]] * table_row[P_COL["BW"]] / 6000.0


def spline_block_operator(sample_time, n_points=11, spacing=15.0):
    """The linear map from the 11 fifteen-minute noise points of one block to the sensor-rate
    samples the reference takes from ``interp1d(t15, noise15, kind='cubic')`` with the first
    sample dropped (noise_gen.py:38-47).  interp1d's cubic is the not-a-knot interpolating spline;
    it is built here in Hermite form (knot slopes from the continuity equations).

    -> W float64 [floor(150 / sample_time), 11]; block samples = W @ points.
    """
    K, h = n_points, float(spacing)
    A = np.zeros((K, K)); R = np.zeros((K, K))          # A @ slopes = R @ y
    for k in range(1, K - 1):
        A[k, k - 1], A[k, k], A[k, k + 1] = 1.0, 4.0, 1.0
        R[k, k + 1], R[k, k - 1] = 3.0 / h, -3.0 / h
    # not-a-knot: the third derivative is continuous across the first and last interior knots
    A[0, 0], A[0, 2] = 1.0, -1.0
    R[0, 0], R[0, 1], R[0, 2] = -2.0 / h, 4.0 / h, -2.0 / h
    A[K - 1, K - 3], A[K - 1, K - 1] = 1.0, -1.0
    R[K - 1, K - 3], R[K - 1, K - 2], R[K - 1, K - 1] = -2.0 / h, 4.0 / h, -2.0 / h
    slopes = np.linalg.solve(A, R)                       # [K, K]: slope_k = slopes[k] @ y
    nsample = int(np.floor((K - 1) * h / float(sample_time))) + 1
    eye = np.eye(K)
    W = np.empty((nsample - 1, K))
    for r in range(1, nsample):
        tt = r * float(sample_time)
        k = min(int(tt // h), K - 2)
        u = (tt - k * h) / h
        h00 = (1 + 2 * u) * (1 - u) ** 2; h10 = u * (1 - u) ** 2
        h01 = u * u * (3 - 2 * u); h11 = u * u * (u - 1)
        W[r - 1] = h00 * eye[k] + h01 * eye[k + 1] + h * (h10 * slopes[k] + h11 * slopes[k + 1])
    return W
