"""T1DSimEnv: the single-episode environment of the reference (``simglucose/simulation/env.py:36-180``)
as a thin adapter over a one-env device batch.

``reset()`` / ``step(action, reward_fun=risk_diff)`` return the reference's ``Step(observation=
Observation(CGM), reward, done, info)`` with the same info keys; per step the adapter asks the scenario
for the meal announced in each simulated minute, hands the kernel one launch worth of work and appends
to the history lists behind ``show_history()``.  The sensor noise uses the exact
``numpy.random.RandomState(sensor.seed).randn()`` stream (host-normals mode of the kernel).
"""
import logging
from collections import namedtuple
from datetime import timedelta

import numpy as np
import pandas as pd

from ..analysis.risk import risk_index
from ..patient.t1dpatient import Action

try:
    from rllab.envs.base import Step
except ImportError:
    _Step = namedtuple("Step", ["observation", "reward", "done", "info"])

    def Step(observation, reward, done, **kwargs):
        return _Step(observation, reward, done, kwargs)

Observation = namedtuple("Observation", ["CGM"])
logger = logging.getLogger(__name__)


def risk_diff(BG_last_hour):
    """default reward: risk(previous CGM) - risk(current CGM); 0 until two samples exist."""
    if len(BG_last_hour) < 2:
        return 0
    _, _, now = risk_index([BG_last_hour[-1]], 1)
    _, _, before = risk_index([BG_last_hour[-2]], 1)
    return before - now


class T1DSimEnv(object):
    def __init__(self, patient, sensor, pump, scenario, n_sub=4):
        self.patient = patient
        self.sensor = sensor
        self.pump = pump
        self.scenario = scenario
        self.n_sub = n_sub
        self._batch = None
        self.viewer = None
        self._device_reset()
        self._start_history()

    # ------------------------------------------------------------------ device batch
    def _device_reset(self):
        from ..batch_env import BatchedT1DSimEnv
        self.sample_time = self.sensor.sample_time
        n_draws = 1 + 10 * 16                                   # 16 noise blocks = 40 h; grown on demand
        if self._batch is None:
            self._batch = BatchedT1DSimEnv(patient="custom", n_envs=1, patient_table=self.patient.table_row(),
                                           sensor_row=self.sensor.row(), pump_row=self.pump.row(), noise="host",
                                           normals=self.sensor.normals(n_draws).reshape(-1, 1), n_sub=self.n_sub)
        else:
            self._batch.set_normals(self.sensor.normals(n_draws).reshape(-1, 1))
        self._n_draws = n_draws
        self.patient._attach(self._batch)
        self.sensor._env = self
        x0 = np.asarray(self.patient.init_state, dtype=np.float64).reshape(13, 1)
        self._batch.reset(x0=x0)
        self._batch.sync()

    def _ensure_normals(self, minutes_ahead):
        need = 1 + 10 * (2 + int((self._batch.t[0].item() + minutes_ahead) // 150))
        if need > self._n_draws:
            self._n_draws = max(need, 2 * self._n_draws)
            self._batch.set_normals(self.sensor.normals(self._n_draws).reshape(-1, 1))

    def _last_cgm(self):
        return float(self._batch.last_cgm[0])

    def _scalar(self, name):
        return float(getattr(self._batch, name)[0])

    # ------------------------------------------------------------------ reference surface
    @property
    def time(self):
        return self.scenario.start_time + timedelta(minutes=self.patient.t)

    def _start_history(self):
        b = self._batch
        self.time_hist = [self.scenario.start_time]
        self.BG_hist = [self._scalar("bg")]
        self.CGM_hist = [float(b.cgm0[0])]        # sample #0 (env.py:126)
        self.risk_hist = [self._scalar("risk")]
        self.LBGI_hist = [self._scalar("lbgi")]
        self.HBGI_hist = [self._scalar("hbgi")]
        self.CHO_hist = []
        self.insulin_hist = []

    def step(self, action, reward_fun=risk_diff):
        """action: namedtuple with .basal and .bolus in U/min, held for int(sample_time) minutes."""
        minutes = int(self.sample_time)
        self._ensure_normals(minutes)
        now = self.time
        cho = np.array([[float(self.scenario.get_action(now + timedelta(minutes=m)).meal)] for m in range(minutes)])
        b = self._batch
        b.step(float(np.asarray(action.basal).reshape(-1)[0]), float(np.asarray(action.bolus).reshape(-1)[0]), cho=cho)
        status = b.sync(raise_on_status=False)
        if status & 2:
            logger.error("ODE state became non-finite")
            raise RuntimeError("patient state is no longer finite")
        CGM, BG, CHO, insulin = self._scalar("cgm"), self._scalar("bg"), self._scalar("meal"), self._scalar("insulin")
        LBGI, HBGI, risk = self._scalar("lbgi"), self._scalar("hbgi"), self._scalar("risk")
        self.CHO_hist.append(CHO)
        self.insulin_hist.append(insulin)
        self.time_hist.append(self.time)
        self.BG_hist.append(BG)
        self.CGM_hist.append(CGM)
        self.risk_hist.append(risk)
        self.LBGI_hist.append(LBGI)
        self.HBGI_hist.append(HBGI)
        window = int(60 / self.sample_time)
        reward = self._scalar("reward") if reward_fun is risk_diff else reward_fun(self.CGM_hist[-window:])
        done = bool(b.done[0])
        return Step(observation=Observation(CGM=CGM), reward=reward, done=done, sample_time=self.sample_time,
                    patient_name=self.patient.name, meal=CHO, patient_state=self.patient.state, time=self.time,
                    bg=BG, lbgi=LBGI, hbgi=HBGI, risk=risk)

    def reset(self):
        self.patient.reset()
        self.sensor.reset()
        self.pump.reset()
        self.scenario.reset()
        self._device_reset()
        self._start_history()
        return Step(observation=Observation(CGM=self._scalar("cgm")), reward=0, done=False,
                    sample_time=self.sample_time, patient_name=self.patient.name, meal=0,
                    patient_state=self.patient.state, time=self.time, bg=self.BG_hist[0], lbgi=self.LBGI_hist[0],
                    hbgi=self.HBGI_hist[0], risk=self.risk_hist[0])

    def render(self, close=False):
        """The matplotlib viewer of the reference is not part of this package."""
        return None

    def show_history(self):
        df = pd.DataFrame()
        df["Time"] = pd.Series(self.time_hist)
        df["BG"] = pd.Series(self.BG_hist)
        df["CGM"] = pd.Series(self.CGM_hist)
        df["CHO"] = pd.Series(self.CHO_hist)
        df["insulin"] = pd.Series(self.insulin_hist)
        df["LBGI"] = pd.Series(self.LBGI_hist)
        df["HBGI"] = pd.Series(self.HBGI_hist)
        df["Risk"] = pd.Series(self.risk_hist)
        return df.set_index("Time")
