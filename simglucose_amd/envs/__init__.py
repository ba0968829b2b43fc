from .simglucose_gym_env import T1DSimEnv  # noqa: F401


def __getattr__(name):
    if name == "BatchedGymT1DSimEnv":
        from .batched_gym_env import BatchedGymT1DSimEnv
        return BatchedGymT1DSimEnv
    raise AttributeError(name)
