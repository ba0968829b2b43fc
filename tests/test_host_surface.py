"""CPU suite, part 3: the host-side mirror of the reference's Python surface that needs no GPU --
seeding chain, scenarios, pump/risk helpers, controllers -- against the reference's own known answers
(tests/test_seed.py) and the golden vectors."""
from datetime import datetime, timedelta

import numpy as np
import pytest


def test_gym_seeding_chain_known_answers():
    """tests/test_seed.py:19,23 of the reference: after seed(0)+reset() the episode starts at 23:00,
    after seed(1000)+reset() at 14:00 (the seed() call itself consumes one draw: 14:00 and 10:00)."""
    from simglucose_amd.envs import seeding
    rng, s1 = seeding.np_random(0)
    assert s1 == 0
    assert seeding.derive_episode(rng)[3] == 14 and seeding.derive_episode(rng)[3] == 23
    rng, _ = seeding.np_random(1000)
    assert seeding.derive_episode(rng)[3] == 10 and seeding.derive_episode(rng)[3] == 14
    a = seeding.derive_episode(seeding.np_random(5)[0])
    b = seeding.derive_episode(seeding.np_random(5)[0])
    assert a == b and all(0 <= v < 2 ** 31 for v in a[:3])
    with pytest.raises(ValueError):
        seeding.np_random(-1)


def test_random_scenario_matches_reference_draws(golden):
    from simglucose_amd.simulation.scenario_gen import RandomScenario
    g = golden("g9_seeding.npz")
    t0 = datetime(2018, 1, 1, 0, 0, 0)
    for i, sd in enumerate(g["scen_seeds"]):
        sc = RandomScenario(start_time=t0, seed=int(sd))
        for d in range(g["scen_time"].shape[1]):
            s = sc.scenario if d == 0 else sc.create_scenario()
            n = int(g["scen_count"][i, d])
            assert list(s["meal"]["time"]) == list(g["scen_time"][i, d, :n])
            assert list(s["meal"]["amount"]) == list(g["scen_amount"][i, d, :n])
    for tag, start in (("00h", t0), ("14h", datetime(2018, 1, 1, 14, 0, 0))):
        sc = RandomScenario(start_time=start, seed=1)
        sc.reset()
        meals = np.array([sc.get_action(start + timedelta(minutes=m)).meal for m in range(2880)], dtype=float)
        assert np.array_equal(meals, g["scen_minute_meal_" + tag])


def test_custom_scenario_time_forms():
    from simglucose_amd.simulation.scenario import CustomScenario, parseTime
    t0 = datetime(2018, 1, 1, 6, 0, 0)
    sc = CustomScenario(t0, [(1.5, 40), (timedelta(hours=3, seconds=20), 25), (datetime(2018, 1, 1, 12, 0), 60), (1.5, 99)])
    assert sc.get_action(t0 + timedelta(minutes=90)).meal == 40          # first entry wins
    assert sc.get_action(t0 + timedelta(minutes=180)).meal == 25         # rounded to the minute
    assert sc.get_action(datetime(2018, 1, 1, 12, 0)).meal == 60
    assert sc.get_action(t0).meal == 0
    assert CustomScenario(t0, []).get_action(t0).meal == 0
    with pytest.raises(ValueError):
        parseTime("noon", t0)


def test_pump_helper_and_risk_helper(golden):
    from simglucose_amd.actuator.pump import InsulinPump
    from simglucose_amd.analysis.risk import risk_index
    g = golden("g3_pump.npz")
    for name in ("Insulet", "Cozmo"):
        pump = InsulinPump.withName(name)
        assert np.array_equal([pump.basal(a) for a in g["amount"]], g["basal_" + name])
        assert np.array_equal([pump.bolus(a) for a in g["amount"]], g["bolus_" + name])
        assert pump.row().shape == (6,)
    g = golden("g8_risk.npz")
    got = np.array([risk_index([b], 1) for b in g["bg"]])
    assert np.allclose(got[:, 0], g["lbgi"], rtol=1e-13) and np.allclose(got[:, 1], g["hbgi"], rtol=1e-13)
    with pytest.raises(ValueError):
        InsulinPump.withName("nope")


def test_controllers_reproduce_reference_actions(golden):
    """BBController / PIDController fed the reference's own observation stream give the reference's
    action stream (G6, G10)."""
    import csv, os
    from collections import namedtuple
    from simglucose_amd.controller.basal_bolus_ctrller import BBController
    from simglucose_amd.controller.pid_ctrller import PIDController
    Obs = namedtuple("Observation", ["CGM"])
    here = os.path.dirname(os.path.abspath(__file__))

    def hist(name):
        with open(os.path.join(here, "golden", name), newline="") as f:
            rows = list(csv.DictReader(f))
        return {k: np.array([float(r[k]) if r[k] else np.nan for r in rows]) for k in rows[0] if k != "Time"}
    # NB: the first policy call sees reset()'s observation (CGM sample #1), which is not in the history
    # table (its row 0 is sample #0), so the comparison starts at the second action.
    h = hist("g6_config1_adult001_bb.csv"); acts = golden("g6_config1_actions.npz")["actions"]
    bb = BBController()
    for k in range(1, 480):
        a = bb.policy(Obs(CGM=h["CGM"][k]), 0, False, patient_name="adult#001", meal=h["CHO"][k - 1], sample_time=3.0)
        assert abs(a.basal - acts[k, 0]) < 1e-15 and abs(a.bolus - acts[k, 1]) < 1e-12
    assert BBController().policy(Obs(CGM=200.0), 0, False, patient_name="someone", meal=3.0, sample_time=3).bolus > 0
    h = hist("g10_pid_adult001.csv"); acts = golden("g10_pid_actions.npz")["actions"]
    pid = PIDController(P=0.001, I=0.00001, D=0.001, target=140)
    # replay the integral state: it depends on the unseen first observation, so recover it from action 0
    # closed form: u0 = P (c - 140) + D c / 3  ->  c
    c0 = (acts[0, 0] + 0.001 * 140) / (0.001 + 0.001 / 3.0)
    a0 = pid.policy(Obs(CGM=c0), 0, False, sample_time=3.0)
    assert abs(a0.basal - acts[0, 0]) < 1e-12
    for k in range(1, 480):
        a = pid.policy(Obs(CGM=h["CGM"][k]), 0, False, sample_time=3.0)
        assert abs(a.basal - acts[k, 0]) < 1e-9 and a.bolus == 0
    pid.reset()
    assert pid.integrated_state == 0 and pid.prev_state == 0


def test_patient_descriptor_rows_match_tables():
    """T1DPatient.table_row (pandas row -> T1D_P_* order) equals params.patient_table; random_init_bg
    draws equal the reference's (G9), including the compounding second reset."""
    from simglucose_amd import params
    from simglucose_amd.patient.t1dpatient import T1DPatient
    names, tab = params.patient_table()
    for pid in (1, 11, 30):
        p = T1DPatient.withID(pid)
        assert p.name == names[pid - 1]
        assert np.array_equal(p.table_row(), tab[pid - 1])
        assert np.array_equal(p.state, tab[pid - 1, :13])
        assert p.t == 0 and abs(p.observation.Gsub - tab[pid - 1, 12] / tab[pid - 1, params.P_COL["Vg"]]) < 1e-12


def test_random_init_bg_matches_reference(golden):
    from simglucose_amd.patient.t1dpatient import T1DPatient
    g = golden("g9_seeding.npz")
    for i, name in enumerate(g["init_names"]):
        for j, sd in enumerate(g["init_seeds"]):
            p = T1DPatient.withName(str(name), random_init_bg=True, seed=int(sd))
            assert np.allclose(p.state, g["init_first"][i, j], rtol=0, atol=1e-12)
            p.reset()
            assert np.allclose(p.state, g["init_second"][i, j], rtol=0, atol=1e-12)
