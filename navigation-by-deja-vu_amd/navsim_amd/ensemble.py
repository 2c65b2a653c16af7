"""Ensembles of agents on one GPU: the reference farms independent trials over MPI ranks
(scripts/run_experiment.py:326-347); here N agents that share a landscape and a trained library step in lockstep,
and the device work of 64/A of them shares each pass over the library (dv_sense_step_batch).

Every agent is a full navsim_amd.NavBySceneFamiliarity (own pose, error metrics, stop conditions); only the sensing
and scoring of a step are batched.  An agent that stops (end of path, out of bounds, too far) keeps its final state
and no longer takes part.
"""
import numpy as np

from .agent import StopNavigationException


class NavEnsemble(object):
    def __init__(self, agents):
        if not agents:
            raise ValueError("no agents")
        eng = agents[0]._engine
        if eng is None or any(a._engine is not eng for a in agents):
            raise ValueError("the agents of an ensemble share one engine (NavEnsemble.from_agent)")
        if any(a.track_scene_familiarity for a in agents):
            raise ValueError("construct the agents with track_scene_familiarity=False: a batched pass keeps no per-view minimum")
        self.agents = list(agents)
        self.engine = eng
        self.stop_status = [0] * len(agents)                      # the reference's codes: 0 running / 1 / -1 / -2

    @classmethod
    def from_agent(cls, agent, poses):
        """`agent`: trained, with the GPU sensor model; poses: iterable of ((x, y), angle), one agent each
        (the first pose goes to `agent` itself, the others to clones on the same engine and library)."""
        poses = list(poses)
        agents = [agent] + [agent.clone_for_ensemble() for _ in poses[1:]]
        for a, (pos, ang) in zip(agents, poses):
            a.position = (float(pos[0]), float(pos[1]))
            a.angle = float(ang)
        return cls(agents)

    @property
    def active(self):
        return [i for i, s in enumerate(self.stop_status) if s == 0 and self.agents[i].stopped_with_exception is None]

    SENSE_ERROR_STATUS = -3      # not one of the reference's codes: its trial would have died of an IndexError

    def _stop(self, i, exc):
        self.stop_status[i] = exc.get_code() if hasattr(exc, "get_code") else self.SENSE_ERROR_STATUS
        self.agents[i].stopped_with_exception = exc

    def step_forward(self, fake=False):
        """One step of every running agent; returns the indices that are still running afterwards."""
        idx, xs, ys, angs = [], [], [], []
        for i in self.active:
            try:
                x, y, a = self.agents[i].headings_to_test()
            except StopNavigationException as e:                  # out of the landscape before anything is sensed
                self._stop(i, e)
                continue
            idx.append(i); xs.append(x); ys.append(y); angs.append(a)
        if idx:
            results = self.engine.sense_step_batch(xs, ys, np.stack(angs))
            # the records as arrays when the engine offers them (engine.BatchResults): no dictionary per agent and step
            lean = hasattr(results, "angle_familiarity")
            flags = results.flags.tolist() if lean else [r["flags"] for r in results]
            best = results.best_idex.tolist() if lean else None
            for k, i in enumerate(idx):
                if flags[k] & 16:                                 # DV_RES_SENSE_ERROR: this agent's footprint left the
                    # landscape (a corner reaches r*sqrt(2) > r past the bounds test); the reference's trial ends in an
                    # IndexError, the other trials go on
                    self._stop(i, IndexError("sensor footprint reaches past the end of the landscape (index out of bounds)"))
                    continue
                try:
                    if lean:
                        self.agents[i].apply_step_arrays(results.angle_familiarity[k], best[k], fake)
                    else:
                        self.agents[i].apply_step_result(results[k], fake)
                except StopNavigationException as e:
                    self._stop(i, e)
        return self.active

    def run(self, frames):
        """Up to `frames` steps; returns the per-agent number of completed steps."""
        done = [0] * len(self.agents)
        for _ in range(frames):
            before = self.active
            if not before:
                break
            self.step_forward()
            for i in before:
                if self.stop_status[i] == 0:
                    done[i] += 1
        return done
