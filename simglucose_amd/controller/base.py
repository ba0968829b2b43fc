"""Controller interface (reference surface: ``simglucose/controller/base.py:3-34``)."""
from collections import namedtuple

Action = namedtuple("ctrller_action", ["basal", "bolus"])


class Controller(object):
    def __init__(self, init_state):
        self.init_state = init_state
        self.state = init_state

    def policy(self, observation, reward, done, **info):
        """observation: namedtuple with .CGM; info carries sample_time, patient_name, meal, ...
        -> Action(basal, bolus) in U/min."""
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError
