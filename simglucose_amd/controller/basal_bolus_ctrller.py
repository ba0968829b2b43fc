"""Basal-bolus controller (reference: ``simglucose/controller/basal_bolus_ctrller.py:15-83``):
basal = u2ss * BW / 6000 U/min; when the previous step announced a meal, a bolus of
(carbs / CR + (CGM > 150) (CGM - target) / CF) units spread over one sample."""
from .. import params
from .base import Action, Controller


class BBController(Controller):
    def __init__(self, target=140):
        self.quest = params.quest_table()
        self.names, self.table = params.patient_table()
        self.target = target

    def policy(self, observation, reward, done, **kwargs):
        return self._bb_policy(kwargs.get("patient_name"), kwargs.get("meal"), observation.CGM,
                               kwargs.get("sample_time", 1))

    def _bb_policy(self, name, meal, glucose, env_sample_time):
        if name in self.quest:
            CR, CF = self.quest[name][0], self.quest[name][1]
            row = self.table[self.names.index(name)]
            u2ss, BW = row[params.P_COL["u2ss"]], row[params.P_COL["BW"]]
        else:                                   # the reference's "Average" patient
            CR, CF, u2ss, BW = 1 / 15, 1 / 50, 1.43, 57.0
        basal = u2ss * BW / 6000
        bolus = 0
        if meal > 0:
            bolus = (meal * env_sample_time) / CR + (glucose > 150) * (glucose - self.target) / CF
        return Action(basal=basal, bolus=bolus / env_sample_time)

    def reset(self):
        pass
