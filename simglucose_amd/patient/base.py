"""Abstract patient (reference surface: ``simglucose/patient/base.py:4-31``)."""


class Patient(object):
    def step(self, action):
        raise NotImplementedError

    @staticmethod
    def model(t, x, action, params):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError
