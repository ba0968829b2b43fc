from .simglucose_gym_env import T1DSimEnv  # noqa: F401
