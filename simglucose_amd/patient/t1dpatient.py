"""T1DPatient host object (reference surface: ``simglucose/patient/t1dpatient.py:18-281``).

The 13-state model, the meal bookkeeping and the integrator live in the HIP kernel; this object
holds the parameter row, the initial state and the seed, and owns a one-env device batch (created
on first use) so that ``patient.step(Action(CHO, insulin))`` / ``.state`` / ``.t`` /
``.observation`` keep working stand-alone.  Inside a ``T1DSimEnv`` the env's batch is used instead.
"""
import logging
from collections import namedtuple

import numpy as np
import pandas as pd

from .. import params as _params
from .base import Patient

logger = logging.getLogger(__name__)

Action = namedtuple("patient_action", ["CHO", "insulin"])
Observation = namedtuple("observation", ["Gsub"])
PATIENT_PARA_FILE = _params.PATIENT_PARA_FILE


class T1DPatient(Patient):
    SAMPLE_TIME = 1      # min
    EAT_RATE = 5         # g/min CHO

    def __init__(self, params, init_state=None, random_init_bg=False, seed=None, t0=0):
        self._params = params.copy()       # own copy: reset() writes random_init_bg draws back into it
        self._init_state = init_state
        self.random_init_bg = random_init_bg
        self._seed = seed
        self.t0 = t0
        self._env = None          # BatchedT1DSimEnv (n = 1) this patient currently lives in
        self._own_env = None
        self.reset()

    @classmethod
    def withID(cls, patient_id, **kwargs):
        table = pd.read_csv(PATIENT_PARA_FILE)
        return cls(table.iloc[patient_id - 1, :], **kwargs)

    @classmethod
    def withName(cls, name, **kwargs):
        table = pd.read_csv(PATIENT_PARA_FILE)
        row = table.loc[table.Name == name]
        if len(row) != 1:
            raise ValueError("unknown patient %r" % (name,))
        return cls(row.squeeze(), **kwargs)

    # ---------------------------------------------------------------- parameters for the device
    def table_row(self):
        """-> float64 [45] in include/t1d.h's T1D_P_* order, built from the (possibly edited) params."""
        p = self._params
        row = np.empty(13 + len(_params.MODEL_COLS))
        row[:13] = np.asarray(p.iloc[2:15], dtype=np.float64)
        row[13:] = [float(p[c]) for c in _params.MODEL_COLS]
        return row

    # ---------------------------------------------------------------- state access
    def _batch(self):
        if self._env is None:
            from ..batch_env import BatchedT1DSimEnv
            self._own_env = BatchedT1DSimEnv(patient="custom", n_envs=1, patient_table=self.table_row(),
                                             sensor="Navigator", noise="philox", use_pump=False)
            self._env = self._own_env
            self._env.reset(x0=np.asarray(self.init_state, dtype=np.float64).reshape(13, 1))
        return self._env

    def _attach(self, env):
        self._env = env

    @property
    def state(self):
        if self._env is None:
            return np.asarray(self.init_state, dtype=np.float64).copy()
        return self._env.x[:, 0].double().cpu().numpy()

    @property
    def t(self):
        if self._env is None:
            return self.t0
        return self.t0 + int(self._env.t[0])

    @property
    def sample_time(self):
        return self.SAMPLE_TIME

    @property
    def observation(self):
        return Observation(Gsub=self.state[12] / float(self._params.Vg))

    def step(self, action):
        """One simulated minute: announce action.CHO grams, infuse action.insulin U/min."""
        env = self._batch()
        if env is not self._own_env:
            raise RuntimeError("this patient is driven by its T1DSimEnv; call env.step")
        env.step(float(action.insulin), cho=np.full((1, 1), float(action.CHO)), minutes=1)

    @staticmethod
    def model(t, x, action, params, last_Qsto, last_foodtaken):
        """dx/dt of the 13-state model (reference: the static T1DPatient.model, t1dpatient.py:119-208), evaluated on the
        device through t1d_model_rhs with the reference's own arithmetic (ocml tanh, IEEE divisions).  `params` is a row of
        vpatient_params.csv (pandas Series); `action` has .CHO (g eaten this minute) and .insulin (U/min)."""
        from ..batch_env import BatchedT1DSimEnv
        row = np.empty(13 + len(_params.MODEL_COLS))
        row[:13] = np.asarray(params.iloc[2:15], dtype=np.float64)
        row[13:] = [float(params[c]) for c in _params.MODEL_COLS]
        key = row.tobytes()
        env = T1DPatient._model_envs.get(key)
        if env is None:
            env = T1DPatient._model_envs[key] = BatchedT1DSimEnv(patient="custom", n_envs=1, patient_table=row, sensor="Navigator")
        out = env.model_rhs(np.asarray(x, dtype=np.float64).reshape(13, 1), [0], [float(action.CHO)], [float(action.insulin)],
                            [float(last_Qsto)], [float(last_foodtaken)], math=0)
        return out[:, 0].cpu().numpy()

    _model_envs = {}

    # ---------------------------------------------------------------- reset
    @property
    def seed(self):
        return self._seed

    @seed.setter
    def seed(self, seed):
        self._seed = seed
        self.reset()

    def reset(self):
        """Initial state = params columns x0_1..x0_13 (or init_state); with random_init_bg the three
        glucose states are drawn ~ N(mu, diag(0.1 mu)) from RandomState(seed).multivariate_normal on the
        host (exact numpy stream) and, as in the reference (t1dpatient.py:252,268-270), written back
        into the parameter row, so a second reset draws around the first draw."""
        if self._init_state is None:
            self.init_state = self._params.iloc[2:15]
        else:
            self.init_state = self._init_state
        self.random_state = np.random.RandomState(self.seed)
        if self.random_init_bg:
            vals = np.asarray(self.init_state, dtype=np.float64)
            mean = [1.0 * vals[3], 1.0 * vals[4], 1.0 * vals[12]]
            cov = np.diag([0.1 * vals[3], 0.1 * vals[4], 0.1 * vals[12]])
            bg_init = self.random_state.multivariate_normal(mean, cov)
            if self._init_state is None:
                cols = list(self._params.index[2:15])
                for k, v in zip((3, 4, 12), bg_init):
                    self._params[cols[k]] = 1.0 * v
                self.init_state = self._params.iloc[2:15]
            else:
                self.init_state = np.array(vals)
                self.init_state[[3, 4, 12]] = bg_init
        self.name = self._params.Name
        if self._own_env is not None and self._env is self._own_env:
            self._own_env.reset(x0=np.asarray(self.init_state, dtype=np.float64).reshape(13, 1))
