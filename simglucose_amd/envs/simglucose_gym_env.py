"""gym-style single-patient environment (reference surface: ``simglucose/envs/simglucose_gym_env.py:18-85``).

Controls basal insulin only (bolus 0); every ``reset`` builds a fresh episode from the wrapper's
``np_random``: CGM seed, scenario seed, patient seed (``random_init_bg=True``) and a random start hour,
Dexcom sensor + Insulet pump.  Works without gym installed (gym 0.9.4 is pinned by the reference and
absent here): ``spaces.Box`` is replaced by a minimal stand-in and both the old (``_step``/``_reset``)
and the public method names are provided.  For many environments use ``envs.BatchedGymT1DSimEnv``.
"""
from datetime import datetime

import numpy as np

from ..actuator.pump import InsulinPump
from ..controller.base import Action
from ..patient.t1dpatient import T1DPatient
from ..sensor.cgm import CGMSensor
from ..simulation.env import T1DSimEnv as _T1DSimEnv
from ..simulation.scenario_gen import RandomScenario
from . import seeding

try:                                             # optional: real gym base class and spaces
    import gym as _gym
    from gym import spaces as _spaces
    _Base = _gym.Env
except Exception:                                # noqa: BLE001 - gym absent or incompatible
    _gym = None
    _Base = object

    class _Box(object):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low = np.full(shape, low, dtype=dtype)
            self.high = np.full(shape, high, dtype=dtype)
            self.shape, self.dtype = tuple(shape), dtype

        def sample(self):
            hi = np.where(np.isfinite(self.high), self.high, self.low + 1.0)
            return np.random.uniform(self.low, hi).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    class _spaces(object):
        Box = _Box


class T1DSimEnv(_Base):
    metadata = {"render.modes": ["human"]}
    SENSOR_HARDWARE = "Dexcom"
    INSULIN_PUMP_HARDWARE = "Insulet"

    def __init__(self, patient_name=None, custom_scenario=None, reward_fun=None, seed=None):
        self.patient_name = "adolescent#001" if patient_name is None else patient_name
        self.reward_fun = reward_fun
        self.np_random, _ = seeding.np_random(seed=seed)
        self.env, _, _, _ = self._create_env_from_random_state(custom_scenario)

    def _create_env_from_random_state(self, custom_scenario=None):
        seed2, seed3, seed4, hour = seeding.derive_episode(self.np_random)
        start_time = datetime(2018, 1, 1, hour, 0, 0)
        patient = T1DPatient.withName(self.patient_name, random_init_bg=True, seed=seed4)
        sensor = CGMSensor.withName(self.SENSOR_HARDWARE, seed=seed2)
        scenario = RandomScenario(start_time=start_time, seed=seed3) if custom_scenario is None else custom_scenario
        pump = InsulinPump.withName(self.INSULIN_PUMP_HARDWARE)
        return _T1DSimEnv(patient, sensor, pump, scenario), seed2, seed3, seed4

    # old-gym hook names -----------------------------------------------------
    def _step(self, action):
        act = Action(basal=action, bolus=0)
        if self.reward_fun is None:
            return self.env.step(act)
        return self.env.step(act, reward_fun=self.reward_fun)

    def _reset(self):
        self.env, _, _, _ = self._create_env_from_random_state()     # a custom_scenario is not carried over (as upstream)
        obs, _, _, _ = self.env.reset()
        return obs

    def _seed(self, seed=None):
        self.np_random, seed1 = seeding.np_random(seed=seed)
        self.env, seed2, seed3, seed4 = self._create_env_from_random_state()
        return [seed1, seed2, seed3, seed4]

    def _render(self, mode="human", close=False):
        self.env.render(close=close)

    # public names (gym >= 0.9.6 style) ---------------------------------------
    def step(self, action):
        return self._step(action)

    def reset(self):
        return self._reset()

    def seed(self, seed=None):
        return self._seed(seed)

    def render(self, mode="human", close=False):
        return self._render(mode=mode, close=close)

    @property
    def action_space(self):
        return _spaces.Box(low=0, high=self.env.pump._params["max_basal"], shape=(1,))

    @property
    def observation_space(self):
        return _spaces.Box(low=0, high=np.inf, shape=(1,))
