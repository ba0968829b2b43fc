"""Meal scenarios (reference surface: ``simglucose/simulation/scenario.py:7-59``).  Host objects; the
env turns them into per-minute CHO for the step kernel (batched form: ``scenario_batch.py``)."""
from collections import namedtuple
from datetime import datetime, timedelta

Action = namedtuple("scenario_action", ["meal"])


class Scenario(object):
    def __init__(self, start_time):
        self.start_time = start_time

    def get_action(self, t):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError


def parseTime(time, start_time):
    """numbers are hours after start_time, timedeltas are offsets (both rounded to the minute)."""
    if isinstance(time, (int, float)):
        return start_time + timedelta(minutes=round(time * 60.0))
    if isinstance(time, timedelta):
        return start_time + timedelta(minutes=round(time.total_seconds() / 60.0))
    if isinstance(time, datetime):
        return time
    raise ValueError("Expect time to be int, float, timedelta, datetime")


class CustomScenario(Scenario):
    """scenario: list of (time, grams)."""

    def __init__(self, start_time, scenario):
        Scenario.__init__(self, start_time=start_time)
        self.scenario = scenario

    def get_action(self, t):
        for when, grams in self.scenario or ():          # first matching entry wins
            if parseTime(when, self.start_time) == t:
                return Action(meal=grams)
        return Action(meal=0)

    def reset(self):
        pass
