"""Batched meal scenarios as per-env meal tables for the step kernel.

``random_meal_tables`` is the vectorised, on-device statistical counterpart of the reference's
``RandomScenario.create_scenario`` (``simglucose/simulation/scenario_gen.py:33-60``): per day six
candidate meals with presence probabilities (.95,.3,.95,.3,.95,.3), truncated-normal times (minutes
after midnight) and ``max(round(N(mu, sigma)), 0)`` grams.  It does not reproduce numpy's MT19937
stream (exact-parity scenarios for small N come from ``simulation.scenario_gen.RandomScenario``);
it removes the per-env host loop when N is 10^6.

``tables_from_minute_lists`` packs explicit (minute, grams) lists, e.g. from CustomScenario.
"""
import math

import torch

MEAL_UNUSED = 0x7FFFFFFF

_PROB = (0.95, 0.3, 0.95, 0.3, 0.95, 0.3)
_LB = (5 * 60, 9 * 60, 10 * 60, 14 * 60, 16 * 60, 20 * 60)
_UB = (9 * 60, 10 * 60, 14 * 60, 16 * 60, 20 * 60, 23 * 60)
_MU = (7 * 60, 9.5 * 60, 12 * 60, 15 * 60, 18 * 60, 21.5 * 60)
_SD = (60, 30, 60, 30, 60, 30)
_AMU = (45, 10, 70, 10, 80, 10)
_ASD = (10, 5, 10, 5, 10, 5)


def _phi(x):
    return 0.5 * (1.0 + math.erf(x / math.sqrt(2.0)))


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


def random_meal_tables(n, days=1, start_minute_of_day=0, seed=0, device="cuda:0", dtype=torch.float64):
    """-> (meal_time int32 [6*(days+1), n], meal_amt dtype [6*(days+1), n]) covering `days` days
    from an episode that starts at `start_minute_of_day` (scalar or int tensor [n])."""
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    start = torch.as_tensor(start_minute_of_day, device=device, dtype=torch.int64).expand(n)
    T, A = [], []
    for day in range(days + 1):          # a non-midnight start touches days+1 calendar days
        for k in range(6):
            present = torch.rand(n, generator=gen, device=device, dtype=torch.float64) < _PROB[k]
            a, b = (_LB[k] - _MU[k]) / _SD[k], (_UB[k] - _MU[k]) / _SD[k]
            u = torch.rand(n, generator=gen, device=device, dtype=torch.float64)
            q = _phi(a) + u * (_phi(b) - _phi(a))
            zt = math.sqrt(2.0) * torch.erfinv((2.0 * q - 1.0).clamp(-1 + 1e-15, 1 - 1e-15))
            tod = torch.round(_MU[k] + _SD[k] * zt).clamp(_LB[k], _UB[k]).to(torch.int64)
            grams = torch.round(_AMU[k] + _ASD[k] * torch.randn(n, generator=gen, device=device,
                                                                dtype=torch.float64)).clamp_min(0.0)
            minute = day * 1440 + tod - start
            ok = present & (minute >= 0) & (minute < days * 1440)
            T.append(torch.where(ok, minute, torch.full_like(minute, MEAL_UNUSED)))
            A.append(torch.where(ok, grams, torch.zeros_like(grams)))
    times, amts = _finalise(torch.stack(T), torch.stack(A))
    return times, amts.to(dtype)


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
