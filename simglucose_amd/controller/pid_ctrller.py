"""PID controller on the CGM reading (reference: ``simglucose/controller/pid_ctrller.py:6-40``).
The batched in-kernel form is ``BatchedT1DSimEnv.rollout_pid`` / ``t1d_rollout_pid``."""
from .base import Action, Controller


class PIDController(Controller):
    def __init__(self, P=1, I=0, D=0, target=140):
        self.P, self.I, self.D, self.target = P, I, D, target
        self.integrated_state = 0
        self.prev_state = 0

    def policy(self, observation, reward, done, **kwargs):
        dt = kwargs.get("sample_time")
        bg = observation.CGM
        u = self.P * (bg - self.target) + self.I * self.integrated_state + self.D * (bg - self.prev_state) / dt
        self.prev_state = bg
        self.integrated_state += (bg - self.target) * dt
        return Action(basal=u, bolus=0)

    def reset(self):
        self.integrated_state = 0
        self.prev_state = 0
