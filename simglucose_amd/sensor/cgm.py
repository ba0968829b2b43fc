"""CGMSensor descriptor (reference surface: ``simglucose/sensor/cgm.py:12-50``).

The sensor model itself -- AR(1) -> Johnson-SU -> cubic-spline noise, clamp, zero-order hold -- is
fused into the step kernel.  This object holds the hardware row and the seed; ``normals(k)`` hands
out the first ``k`` values of ``numpy.random.RandomState(seed).randn()``, the exact stream the
reference's ``CGMNoise`` consumes (``sensor/noise_gen.py:75,86-88``), which the env uploads for the
kernel's host-normals mode."""
import numpy as np
import pandas as pd

from ..params import SENSOR_PARA_FILE


class CGMSensor(object):
    def __init__(self, params, seed=None):
        self._params = params
        self.name = params.Name
        self.sample_time = params.sample_time
        self._seed = seed
        self._env = None
        self._last_CGM = 0

    @classmethod
    def withName(cls, name, **kwargs):
        table = pd.read_csv(SENSOR_PARA_FILE)
        row = table.loc[table.Name == name]
        if len(row) != 1:
            raise ValueError("unknown sensor %r" % (name,))
        return cls(row.squeeze(), **kwargs)

    @property
    def seed(self):
        return self._seed

    @seed.setter
    def seed(self, seed):
        self._seed = seed

    def normals(self, count):
        return np.random.RandomState(self._seed).randn(int(count))

    def row(self):
        """-> [PACF, gamma, lambda, delta, xi, sample_time, min, max] for t1d_ctx_create."""
        p = self._params
        return np.array([p[k] for k in ("PACF", "gamma", "lambda", "delta", "xi", "sample_time", "min", "max")],
                        dtype=np.float64)

    def measure(self, patient):
        """Latest CGM value of the env this sensor is attached to (sampling happens inside env.step)."""
        if self._env is None:
            raise RuntimeError("CGMSensor.measure needs the sensor to be part of a T1DSimEnv: the sensor model "
                               "runs inside the fused step kernel")
        return self._env._last_cgm()

    def reset(self):
        self._last_CGM = 0
