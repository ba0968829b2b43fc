"""InsulinPump descriptor (reference surface: ``simglucose/actuator/pump.py:11-43``).

Inside ``T1DSimEnv.step`` the quantiser runs in the HIP kernel (``pump_quantise``); this object
carries the hardware row and offers the same ``basal``/``bolus`` helpers for host code."""
import numpy as np
import pandas as pd

from ..params import INSULIN_PUMP_PARA_FILE


class InsulinPump(object):
    U2PMOL = 6000

    def __init__(self, params):
        self._params = params

    @classmethod
    def withName(cls, name):
        table = pd.read_csv(INSULIN_PUMP_PARA_FILE)
        row = table.loc[table.Name == name]
        if len(row) != 1:
            raise ValueError("unknown insulin pump %r" % (name,))
        return cls(row.squeeze())

    def _quantise(self, amount, inc, lo, hi):
        pmol = np.round(amount * self.U2PMOL / inc) * inc        # round-half-to-even, as numpy rounds
        return max(min(pmol / self.U2PMOL, hi), lo)

    def bolus(self, amount):
        p = self._params
        return self._quantise(amount, p["inc_bolus"], p["min_bolus"], p["max_bolus"])

    def basal(self, amount):
        p = self._params
        return self._quantise(amount, p["inc_basal"], p["min_basal"], p["max_basal"])

    def row(self):
        """-> [min_bolus, max_bolus, inc_bolus, min_basal, max_basal, inc_basal] for t1d_ctx_create."""
        p = self._params
        return np.array([p[k] for k in ("min_bolus", "max_bolus", "inc_bolus", "min_basal", "max_basal", "inc_basal")],
                        dtype=np.float64)

    def reset(self):
        pass
