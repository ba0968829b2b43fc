"""Numeric part of the reference's post-run report (``simglucose/analysis/report.py``) for a batch of envs whose
BG history lives on the device: time-in-range percentages (``percent_stats``, report.py:74-92), the chunked risk
trace (``risk_index_trace``, :95-110) and the control-variability grid analysis (``CVGA_analysis``, :198-217).
One kernel launch (``t1d_outcome_stats``, include/t1d.h) reads the ``[rows, n]`` BG table -- written by
``BatchedT1DSimEnv.rollout_*(trace=...)`` or stacked by the caller -- and returns per-env results as tensors; the
matplotlib figures of the reference are out of scope.
"""
import ctypes as C

import torch

from .. import _lib

ZONES = ("A", "B", "C", "D", "E", "none")
PERCENT_COLUMNS = ("BG>180", "BG<70", "70<=BG<=180", "BG>250", "BG<50")        # report.py:77-86


def outcome_stats(bg, chunk=60, q_lo=2.5, q_hi=97.5, counts=True, percentiles=True, risk_trace=True):
    """bg: contiguous float tensor [rows, n] on a ROCm device.  -> dict with
    ``percent`` [5, n] (PERCENT_COLUMNS order), ``bg_min`` / ``bg_max`` [n] (q_lo / q_hi percentiles, clipped to
    [50, 400] as CVGA does), ``pct`` [2, n] (unclipped), ``zone`` uint8 [n] (index into ZONES),
    ``lbgi`` / ``hbgi`` [ceil(rows / chunk), n]."""
    if bg.dim() != 2 or not bg.is_contiguous() or bg.dtype not in (torch.float64, torch.float32):
        raise ValueError("bg must be a contiguous float32/float64 tensor [rows, n]")
    if bg.device.type != "cuda":
        raise _lib.T1DError("outcome_stats runs on the GPU (device=%s)" % bg.device)
    L = _lib.lib()
    rows, n = bg.shape
    o = _lib.Outcome()
    o.q_lo, o.q_hi, o.chunk = float(q_lo), float(q_hi), int(chunk)
    out = {}
    keep = []
    if counts:
        cnt = torch.empty(5, n, dtype=torch.int32, device=bg.device); o.counts = cnt.data_ptr(); keep.append(cnt)
    if percentiles:
        pct = torch.empty(2, n, dtype=bg.dtype, device=bg.device); o.pct = pct.data_ptr()
        zone = torch.empty(n, dtype=torch.uint8, device=bg.device); o.zone = zone.data_ptr()
    if risk_trace:
        nch = (rows + int(chunk) - 1) // int(chunk)
        rt = torch.empty(nch, 2, n, dtype=bg.dtype, device=bg.device); o.risk_trace = rt.data_ptr()
    dev_index = bg.device.index if bg.device.index is not None else torch.cuda.current_device()
    with torch.cuda.device(bg.device):
        stream = C.c_void_p(torch.cuda.current_stream(bg.device).cuda_stream)
        _lib.check(L.t1d_outcome_stats(dev_index, 0 if bg.dtype == torch.float64 else 1, n, rows,
                                       C.c_void_p(bg.data_ptr()), C.byref(o), stream))
    if counts:
        out["counts"] = cnt
        out["percent"] = cnt.to(torch.float64) / float(rows) * 100.0
    if percentiles:
        out["pct"], out["zone"] = pct, zone
        out["bg_min"], out["bg_max"] = pct[0].clamp(50, 400), pct[1].clamp(50, 400)
    if risk_trace:
        out["lbgi"], out["hbgi"] = rt[:, 0], rt[:, 1]
    return out


def percent_stats(bg):
    """report.py:74-92 -> [5, n] percentages (PERCENT_COLUMNS order)."""
    return outcome_stats(bg, percentiles=False, risk_trace=False)["percent"]


def risk_index_trace(bg, chunk=60):
    """report.py:95-110 -> (LBGI, HBGI, Risk Index) each [n_chunks, n]; the reference's 'hour' is 60 samples."""
    r = outcome_stats(bg, chunk=chunk, counts=False, percentiles=False)
    return r["lbgi"], r["hbgi"], r["lbgi"] + r["hbgi"]


def CVGA_analysis(bg):
    """report.py:198-217 -> BG_min, BG_max [n] and the zone fractions perA..perE (B excludes A, as upstream)."""
    r = outcome_stats(bg, counts=False, risk_trace=False)
    z = r["zone"]
    m = float(z.numel())
    frac = [float((z == k).sum()) / m for k in range(5)]
    return (r["bg_min"], r["bg_max"]) + tuple(frac)


def history_frame(trace, env_index, start_time, sample_time):
    """T1DSimEnv.show_history() (simulation/env.py:169-180) for one env of a device-resident history (see
    BatchedT1DSimEnv.new_trace): DataFrame indexed by Time with columns BG, CGM, CHO, insulin, LBGI, HBGI, Risk;
    the last row has no CHO / insulin, as upstream."""
    from datetime import timedelta
    import numpy as np
    import pandas as pd
    from .risk import risk_index
    rows = int(trace["row"])
    col = lambda k: trace[k][:rows, env_index].double().cpu().numpy()
    bg, cgm = col("bg"), col("cgm")
    cho = np.append(col("cho")[1:], np.nan)                 # CHO_hist / insulin_hist are one entry shorter (env.py:88-89,174-175)
    ins = np.append(col("insulin")[1:], np.nan)
    risk = np.array([risk_index([v], 1) for v in bg], dtype=float)
    df = pd.DataFrame({"Time": [start_time + timedelta(minutes=float(sample_time) * k) for k in range(rows)],
                       "BG": bg, "CGM": cgm, "CHO": cho, "insulin": ins,
                       "LBGI": risk[:, 0], "HBGI": risk[:, 1], "Risk": risk[:, 2]})
    return df.set_index("Time")


def save_histories(trace, env_indices, names, path, start_time, sample_time):
    """SimObj.save_results (simulation/sim_engine.py:44-49) for a selected subset of envs: <path>/<name>.csv."""
    import os
    os.makedirs(path, exist_ok=True)
    out = []
    for i, name in zip(env_indices, names):
        f = os.path.join(path, str(name) + ".csv")
        history_frame(trace, i, start_time, sample_time).to_csv(f)
        out.append(f)
    return out
