"""simglucose_amd -- MI355X-native batched Type-1-Diabetes glucose-insulin simulator.

The hot path (pump -> meal bookkeeping -> 13-state ODE / RK4 -> CGM noise -> risk/reward) is one
hand-written HIP kernel launch per ``env.step`` for the whole batch (``csrc/``, C ABI in
``include/t1d.h``); this package is the host-side mirror of the reference's Python surface.
"""
from ._lib import T1DError, build  # noqa: F401

__version__ = "0.1.0"


def __getattr__(name):
    if name == "BatchedT1DSimEnv":
        from .batch_env import BatchedT1DSimEnv
        return BatchedT1DSimEnv
    raise AttributeError(name)
