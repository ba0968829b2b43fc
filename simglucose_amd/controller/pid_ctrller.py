"""PID controller on the CGM reading, host form.

Mirrors the controller the reference ships (``simglucose/controller/pid_ctrller.py:6-40``) so that scripts written
against it keep working: same constructor keywords, same public attributes (``P``, ``I``, ``D``, ``target``,
``integrated_state``, ``prev_state``) and the same ``policy`` / ``reset`` calls.  The arithmetic is exactly the
per-env update of the in-kernel closed loop (``t1d_rollout_pid`` in ``include/t1d.h``, ``rollout_body`` in
``csrc/t1d_kernels.hpp``), which is what ``BatchedT1DSimEnv.rollout_pid`` runs for a whole batch:

    u_k   = P (y_k - target) + I S_k + D (y_k - y_{k-1}) / dt
    S_k+1 = S_k + (y_k - target) dt          (the integral is advanced AFTER it has been used)

with ``y`` the CGM reading in mg/dL, ``dt`` the sensor's sample time in minutes and ``u`` the basal rate in U/min.
The first call sees ``y_{-1} = 0`` and ``S_0 = 0``; the pump clamps whatever comes out (``actuator/pump.py``).
"""
from .base import Action, Controller


def pid_update(gains, target, integral, previous, reading, dt):
    """One PID step.  Returns (command, new_integral, new_previous); pure, so that tests can run it side by side with
    the device roll-out."""
    kp, ki, kd = gains
    error = reading - target
    command = kp * error + ki * integral + kd * (reading - previous) / dt
    return command, integral + error * dt, reading


class PIDController(Controller):
    def __init__(self, P=1, I=0, D=0, target=140):
        self.P = P
        self.I = I
        self.D = D
        self.target = target
        self.reset()

    def reset(self):
        self.integrated_state = 0
        self.prev_state = 0

    def policy(self, observation, reward, done, **kwargs):
        command, self.integrated_state, self.prev_state = pid_update(
            (self.P, self.I, self.D), self.target, self.integrated_state, self.prev_state,
            observation.CGM, kwargs.get("sample_time"))
        return Action(basal=command, bolus=0)
