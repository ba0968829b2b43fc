"""Batched meal scenarios as per-env meal tables for the step kernel.

``random_meal_tables`` is the on-device (one HIP kernel, `t1d_random_meals`) statistical counterpart of the reference's
``RandomScenario.create_scenario`` (``simglucose/simulation/scenario_gen.py:33-60``): per day six
candidate meals with presence probabilities (.95,.3,.95,.3,.95,.3), truncated-normal times (minutes
after midnight) and ``max(round(N(mu, sigma)), 0)`` grams.  It does not reproduce numpy's MT19937
stream (exact-parity scenarios for small N come from ``simulation.scenario_gen.RandomScenario``);
it removes the per-env host loop when N is 10^6.

``tables_from_minute_lists`` packs explicit (minute, grams) lists, e.g. from CustomScenario.
"""
import torch

MEAL_UNUSED = 0x7FFFFFFF

def _finalise(times, amts):
    """sort per env by time (stable: earlier slots win ties), drop same-minute duplicates."""
    times, order = torch.sort(times, dim=0, stable=True)
    amts = torch.gather(amts, 0, order)
    dup = torch.zeros_like(times, dtype=torch.bool)
    dup[1:] = (times[1:] == times[:-1]) & (times[1:] != MEAL_UNUSED)
    times = torch.where(dup, torch.full_like(times, MEAL_UNUSED), times)
    times, order = torch.sort(times, dim=0, stable=True)
    amts = torch.gather(amts, 0, order)
    return times.to(torch.int32).contiguous(), amts.contiguous()


def random_meal_tables(n, days=1, start_minute_of_day=0, seed=0, device="cuda:0", dtype=torch.float64, env_offset=0):
    """-> (meal_time int32 [6*(days+1), n], meal_amt dtype [6*(days+1), n]) covering `days` days
    from an episode that starts at `start_minute_of_day` (scalar or int tensor [n]).  One launch of
    t1d_random_meals (include/t1d.h); env i draws from Philox subsequence env_offset + i."""
    import ctypes as C
    from . import _lib
    L = _lib.lib()
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.T1DError("random_meal_tables runs on the GPU (device=%s)" % device)
    rows = 6 * (int(days) + 1)
    mt = torch.empty(rows, n, dtype=torch.int32, device=device)
    ma = torch.empty(rows, n, dtype=dtype, device=device)
    start_t, start_s = None, 0
    if isinstance(start_minute_of_day, torch.Tensor) or hasattr(start_minute_of_day, "__len__"):
        start_t = torch.as_tensor(start_minute_of_day).to(device=device, dtype=torch.int32).contiguous()
        if start_t.numel() != n:
            raise ValueError("start_minute_of_day must be a scalar or have n entries")
    else:
        start_s = int(start_minute_of_day)
    dev_index = device.index if device.index is not None else torch.cuda.current_device()
    with torch.cuda.device(device):
        stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        _lib.check(L.t1d_random_meals(dev_index, int(seed) & 0xFFFFFFFFFFFFFFFF, int(env_offset), int(n),
                                      0 if dtype == torch.float64 else 1, int(days),
                                      C.c_void_p(start_t.data_ptr()) if start_t is not None else None, start_s,
                                      C.c_void_p(mt.data_ptr()), C.c_void_p(ma.data_ptr()), stream))
    return mt, ma


def tables_from_minute_lists(lists, device="cuda:0", dtype=torch.float64):
    """lists[i] = [(minute_since_start, grams), ...] for env i -> (meal_time, meal_amt)."""
    n = len(lists)
    m = max(1, max((len(l) for l in lists), default=1))
    t = torch.full((m, n), MEAL_UNUSED, dtype=torch.int64)
    a = torch.zeros((m, n), dtype=torch.float64)
    for i, l in enumerate(lists):
        for j, (minute, grams) in enumerate(l):
            t[j, i] = int(minute); a[j, i] = float(grams)
    t, a = _finalise(t, a)
    return t.to(device), a.to(dtype).to(device)
