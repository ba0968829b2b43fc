"""RandomScenario (reference: ``simglucose/simulation/scenario_gen.py:10-73``): per day up to six meals
-- (breakfast, snack, lunch, snack, dinner, snack) taken with probability (.95,.3,.95,.3,.95,.3), at
truncated-normal times (minutes after midnight) with rounded-normal amounts -- drawn from
``numpy.random.RandomState(seed)`` at reset and again whenever the clock reads 00:00."""
from datetime import datetime

import numpy as np
from scipy.stats import truncnorm

from .scenario import Action, Scenario

_SLOTS = (  # (probability, earliest, latest, mean, sd [hours -> minutes below], mean grams, sd grams)
    (0.95, 5, 9, 7, 60, 45, 10), (0.3, 9, 10, 9.5, 30, 10, 5), (0.95, 10, 14, 12, 60, 70, 10),
    (0.3, 14, 16, 15, 30, 10, 5), (0.95, 16, 20, 18, 60, 80, 10), (0.3, 20, 23, 21.5, 30, 10, 5))


class RandomScenario(Scenario):
    def __init__(self, start_time, seed=None):
        Scenario.__init__(self, start_time=start_time)
        self.seed = seed

    def get_action(self, t):
        since_midnight = (t - datetime.combine(t.date(), datetime.min.time())).total_seconds()
        if since_midnight < 1:
            self.scenario = self.create_scenario()
        minute = np.floor(since_midnight / 60.0)
        times = self.scenario["meal"]["time"]
        if minute in times:
            return Action(meal=self.scenario["meal"]["amount"][times.index(minute)])
        return Action(meal=0)

    def create_scenario(self):
        day = {"meal": {"time": [], "amount": []}}
        for prob, lo, hi, mean, sd, grams, grams_sd in _SLOTS:
            if self.random_gen.rand() < prob:
                lo_m, hi_m, mean_m = lo * 60, hi * 60, mean * 60
                when = np.round(truncnorm.rvs(a=(lo_m - mean_m) / sd, b=(hi_m - mean_m) / sd, loc=mean_m, scale=sd,
                                              random_state=self.random_gen))
                day["meal"]["time"].append(when)
                day["meal"]["amount"].append(max(round(self.random_gen.normal(grams, grams_sd)), 0))
        return day

    def reset(self):
        self.random_gen = np.random.RandomState(self.seed)
        self.scenario = self.create_scenario()

    @property
    def seed(self):
        return self._seed

    @seed.setter
    def seed(self, seed):
        self._seed = seed
        self.reset()
