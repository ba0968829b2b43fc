"""Episode loop (reference surface: ``simglucose/simulation/sim_engine.py:15-76``)."""
import logging
import os
import time

logger = logging.getLogger(__name__)


class SimObj(object):
    def __init__(self, env, controller, sim_time, animate=True, path=None):
        self.env, self.controller, self.sim_time = env, controller, sim_time
        self.animate, self.path = animate, path

    def simulate(self):
        self.controller.reset()
        obs, reward, done, info = self.env.reset()
        tic = time.time()
        end = self.env.scenario.start_time + self.sim_time
        while self.env.time < end:                      # `done` does not stop the episode (as upstream)
            if self.animate:
                self.env.render()
            action = self.controller.policy(obs, reward, done, **info)
            obs, reward, done, info = self.env.step(action)
        logger.info("Simulation took %.2f seconds.", time.time() - tic)

    def results(self):
        return self.env.show_history()

    def save_results(self):
        if self.path is None:
            return
        os.makedirs(self.path, exist_ok=True)
        self.results().to_csv(os.path.join(self.path, str(self.env.patient.name) + ".csv"))

    def reset(self):
        self.env.reset()
        self.controller.reset()


def sim(sim_object):
    sim_object.simulate()
    sim_object.save_results()
    return sim_object.results()


def batch_sim(sim_instances, parallel=False):
    """Runs the instances one after another: each owns a device context, and many envs of one kind are
    better served by one BatchedT1DSimEnv than by a process pool."""
    return [sim(s) for s in sim_instances]
