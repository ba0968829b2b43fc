"""Kovatchev blood-glucose risk index on the host (numpy).

Host-side helper for custom reward functions and reports; inside ``env.step`` the same quantity is
computed by the HIP kernel (``risk_index1`` in csrc/t1d_device.hpp).  Semantics of the reference's
``simglucose/analysis/risk.py:5-17``: the mean low/high risk of the last ``horizon`` samples, with
empty means and NaNs mapped to 0 as ``numpy.nan_to_num`` does."""
import warnings

import numpy as np


def risk_index(BG, horizon):
    window = np.asarray(BG, dtype=np.float64)[-horizon:]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        f = 1.509 * (np.log(window) ** 1.084 - 5.381)
        low, high = f[f < 0], f[f > 0]
        LBGI = np.nan_to_num(np.mean(10.0 * low ** 2))
        HBGI = np.nan_to_num(np.mean(10.0 * high ** 2))
    return (LBGI, HBGI, LBGI + HBGI)
