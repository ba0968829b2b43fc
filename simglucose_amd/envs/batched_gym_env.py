"""BatchedGymT1DSimEnv: N gym-style episodes (Dexcom + Insulet, basal-only action, RandomScenario,
random start hour, random initial glucose) advanced by one kernel launch per step.

Two ways to build the episodes:
  * ``exact=False`` (default, any N): per-env start hour, meal tables and initial glucose are drawn on
    the device (torch generator + Philox in the kernels) -- statistically what N instances of the
    reference's gym wrapper produce;
  * ``exact=True`` (small N): env i replays the reference wrapper seeded with ``seed + i`` bit for bit --
    gym's seeding chain, numpy's RandomState streams for sensor noise, scenario and initial glucose
    (``simglucose/envs/simglucose_gym_env.py:58-73``) -- through host-side set-up and the kernel's
    host-normals mode.
"""
from datetime import datetime, timedelta

import numpy as np
import torch

from .. import _lib, params, scenario_batch
from ..batch_env import BatchedT1DSimEnv
from . import seeding


class BatchedGymT1DSimEnv(object):
    SENSOR_HARDWARE = "Dexcom"
    INSULIN_PUMP_HARDWARE = "Insulet"

    def __init__(self, n_envs, patient_name="adolescent#001", seed=0, device="cuda:0", dtype=torch.float64,
                 exact=False, auto_reset=False, horizon_days=2, n_sub=4, env_offset=0, reward_fun=None):
        self.n = int(n_envs)
        names = [patient_name] * self.n if isinstance(patient_name, str) else list(patient_name)
        self.patient_names = names
        self.seed_value, self.exact, self.auto_reset = int(seed), bool(exact), bool(auto_reset)
        self.horizon_days = int(horizon_days)
        # as the reference wrapper's reward_fun (simglucose_gym_env.py:27-46), for the batch: f(window [20, n]) -> [n]
        self.reward_fun = reward_fun
        self._episode = 0
        self.env = BatchedT1DSimEnv(patient=names, sensor=self.SENSOR_HARDWARE, pump=self.INSULIN_PUMP_HARDWARE,
                                    dtype=dtype, device=device, n_sub=n_sub, seed=self.seed_value,
                                    env_offset=env_offset, noise="philox", random_init_bg=not exact,
                                    cgm_history=reward_fun is not None)
        self.start_hour = torch.zeros(self.n, dtype=torch.int64, device=self.env.device)
        self.max_basal = float(self.env.pump_row[4])

    # ------------------------------------------------------------------ episode construction
    def _build_exact(self):
        from ..patient.t1dpatient import T1DPatient
        from ..simulation.scenario_gen import RandomScenario
        n, st = self.n, int(self.env.sample_time)
        minutes = self.horizon_days * 1440
        n_draws = 1 + 10 * (2 + minutes // 150)
        z = np.empty((n_draws, n)); x0 = np.empty((13, n)); hours = np.empty(n, dtype=np.int64); lists = []
        for i in range(n):
            rng, _ = seeding.np_random(self.seed_value + i)
            for _ in range(self._episode + 1):                # seed() consumed one draw, every reset() one more
                seeding.derive_episode(rng)
            seed2, seed3, seed4, hour = seeding.derive_episode(rng)
            hours[i] = hour
            z[:, i] = np.random.RandomState(seed2).randn(n_draws)
            p = T1DPatient.withName(self.patient_names[i], random_init_bg=True, seed=seed4)
            p.reset()                                         # wrapper resets twice per episode (quirk 9)
            x0[:, i] = np.asarray(p.init_state, dtype=np.float64)
            start = datetime(2018, 1, 1, hour, 0, 0)
            sc = RandomScenario(start_time=start, seed=seed3)
            sc.reset()
            meals = []
            for m in range(minutes):
                g = sc.get_action(start + timedelta(minutes=m)).meal
                if g > 0:
                    meals.append((m, float(g)))
            lists.append(meals)
        self.env.set_normals(z)
        mt, ma = scenario_batch.tables_from_minute_lists(lists, device=self.env.device, dtype=self.env.dtype)
        self.env.set_meals(mt, ma)
        self.start_hour = torch.as_tensor(hours, device=self.env.device)
        return x0

    def _start_hours(self):
        """start hour of every env of this episode, a function of (seed, episode, GLOBAL env id) only -- splitmix64 of the
        id, keyed -- so that the shards of a multi-GPU job draw what the slices of one big batch would"""
        gid = torch.arange(self.n, dtype=torch.int64, device=self.env.device) + int(self.env.env_offset)
        lsr = lambda v, s: (v >> s) & ((1 << (64 - s)) - 1)               # logical shift on two's-complement int64
        wrap = lambda c: c - (1 << 64) if c >= (1 << 63) else c
        z = gid * wrap(0x9E3779B97F4A7C15) + wrap((self.seed_value * 1000003 + self._episode) & 0xFFFFFFFFFFFFFFFF)
        z = (z ^ lsr(z, 30)) * wrap(0xBF58476D1CE4E5B9)
        z = (z ^ lsr(z, 27)) * wrap(0x94D049BB133111EB)
        z = z ^ lsr(z, 31)
        return lsr(z, 11) % 24

    def _build_device(self, mask=None):
        hours = self._start_hours()
        if mask is not None:
            hours = torch.where(mask.bool(), hours, self.start_hour)
        self.start_hour = hours
        mt, ma = scenario_batch.random_meal_tables(self.n, days=self.horizon_days, start_minute_of_day=hours * 60,
                                                   seed=self.seed_value * 7919 + self._episode, device=self.env.device,
                                                   dtype=self.env.dtype, env_offset=self.env.env_offset)
        if mask is not None and self.env.meal_time is not None and self.env.meal_time.shape == mt.shape:
            keep = ~mask.bool()
            mt[:, keep] = self.env.meal_time[:, keep]; ma[:, keep] = self.env.meal_amt[:, keep]
            cursor_meta, nm = self.env.meta.clone(), self.env.next_meal.clone()
            self.env.set_meals(mt, ma)
            self.env.meta[keep] = cursor_meta[keep]; self.env.next_meal[keep] = nm[keep]
        else:
            self.env.set_meals(mt, ma)

    # ------------------------------------------------------------------ gym surface
    def seed(self, seed=None):
        self.seed_value = int(seed or 0)
        self.env.seed = self.seed_value & 0xFFFFFFFFFFFFFFFF
        self.env._b.seed = self.env.seed
        self._episode = 0
        return [self.seed_value]

    def reset(self, mask=None):
        """-> observation CGM [n].  mask (optional, device-built episodes only): reset just those envs."""
        if self.exact:
            if mask is not None:
                raise ValueError("masked reset is not available with exact=True")
            x0 = self._build_exact()
            obs = self.env.reset(x0=x0)
        else:
            m = None if mask is None else torch.as_tensor(mask, device=self.env.device)
            self._build_device(m)
            obs = self.env.reset(mask=m)
        self._episode += 1
        return obs

    def step(self, action):
        """action: basal U/min, tensor [n] or [n, 1] (or a scalar).  -> (obs [n], reward [n], done [n] bool, info)."""
        a = torch.as_tensor(action, dtype=self.env.dtype, device=self.env.device).reshape(-1)
        if a.numel() == 1:
            a = a.expand(self.n)
        obs, reward, done, info = self.env.step(a.contiguous(), reward_fun=self.reward_fun)
        done_b = done.bool()
        if self.auto_reset and not self.exact and bool(done_b.any()):
            info = dict(info, terminal_observation=obs.clone())
            obs = obs.clone(); reward = reward.clone()
            new_obs = self.reset(mask=done_b)
            obs = torch.where(done_b, new_obs, obs)
        return obs, reward, done_b, info

    def time(self):
        """per-env wall-clock as minutes since 2018-01-01 00:00."""
        return self.start_hour * 60 + self.env.t.long()

    @property
    def action_space(self):
        from .simglucose_gym_env import _spaces
        return _spaces.Box(low=0, high=self.max_basal, shape=(self.n, 1))

    @property
    def observation_space(self):
        from .simglucose_gym_env import _spaces
        return _spaces.Box(low=0, high=np.inf, shape=(self.n, 1))
