"""Ensembles of agents on one GPU: the reference farms independent trials over MPI ranks
(scripts/run_experiment.py:326-347); here N agents that share a landscape and a trained library step in lockstep,
and the device work of 64/A of them shares each pass over the library (dv_sense_step_batch).

Every agent is a full navsim_amd.NavBySceneFamiliarity (own pose, error metrics, stop conditions); only the sensing
and scoring of a step are batched.  An agent that stops (end of path, out of bounds, too far) keeps its final state
and no longer takes part.
"""
import numpy as np

from .agent import StopNavigationException, OutOfLandscapeBoundsException


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
        # update_error (:252-276) of all members in ONE device call per ensemble step: every member gets a coverage array of its own on
        # the device (dv_path_slots) and hands its position in; 32 members x a NumPy pass over 50 000 training points each were
        # ~10 ms of host time beside a 0.85 ms device step
        # members made alike (from_agent: clones) have their candidate headings and the poses those lead to worked out as arrays for
        # all members at once (NumPy's elementwise loops give an element what they give the scalar)
        a0 = agents[0]
        self._uniform = all(a.step_size == a0.step_size and a.landscape is a0.landscape and a._sensor_r == a0._sensor_r and
                            np.array_equal(a.angle_offsets, a0.angle_offsets) for a in agents)
        self._stepping = False
        self._pending = []                                        # members whose position awaits its metrics
        self._too_far = {}
        if agents[0].training_path is not None and hasattr(eng, "path_slots") and all(a.training_path is agents[0].training_path for a in agents):
            for a in self.agents:
                if getattr(a, "_metrics_on_device", False):
                    a._collect_errors()                           # (answers still outstanding from steps it took on its own)
            if not any(getattr(a, "_metrics_on_device", False) for a in self.agents):
                eng.set_training_path(agents[0].training_path)    # (the engine holds the path already where an agent's metrics ran on it)
            eng.path_slots(len(agents))
            for j, a in enumerate(self.agents):
                a._metric_slot, a._ens = j, self
                a._metrics_on_device = False

    def _want_error(self, agent):
        self._pending.append((agent, agent.position[0], agent.position[1]))

    def _drop_errors(self, agent):
        self._pending = [p for p in self._pending if p[0] is not agent]

    def _flush_errors(self, raise_for=None):
        """The metrics of every position handed in since the last flush; a member too far from the path is remembered (_too_far) for
        the ensemble's step to stop it as the reference's update_error would have (:264) -- or raised at once for `raise_for`, a
        member stepping on its own."""
        if not self._pending:
            return
        pend, self._pending = self._pending, []
        a0 = pend[0][0]
        nearest = self.engine.path_error_batch([a._metric_slot for a, _, _ in pend], [x for _, x, _ in pend], [y for _, _, y in pend],
                                               a0.coverage_threshold_factor * a0.step_size)
        for (a, _, _), d in zip(pend, nearest.tolist()):
            try:
                a._take_error(d)
            except StopNavigationException as e:
                if a is raise_for:
                    raise
                self._too_far[id(a)] = e

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
        self._stepping = True
        try:
            return self._step_forward(fake)
        finally:
            self._stepping = False

    def _step_forward(self, fake):
        idx, xs, ys, angs = [], [], [], []
        cands = None
        act = self.active
        if self._uniform and act:
            a0 = self.agents[act[0]]
            b = a0._bounds_tuple()
            px = np.array([self.agents[i].position[0] for i in act], dtype=np.float64)
            py = np.array([self.agents[i].position[1] for i in act], dtype=np.float64)
            ang = np.array([self.agents[i].angle for i in act], dtype=np.float64)
            out = (px <= b[0]) | (py <= b[0]) | (px >= b[1]) | (py >= b[2])        # the bounds test of :153-158, before anything is sensed
            for k in np.nonzero(out)[0].tolist():
                self.agents[act[k]].angle_familiarity[:] = np.nan
                self._stop(act[k], OutOfLandscapeBoundsException())
            keep = np.nonzero(~out)[0]
            idx = [act[k] for k in keep.tolist()]
            if idx:
                xs, ys = px[keep], py[keep]
                cand_angle = (ang[keep][:, None] + a0.angle_offsets[None, :]) % (2 * np.pi)
                angs = cand_angle
                cands = (cand_angle, xs[:, None] + a0.step_size * np.cos(cand_angle), ys[:, None] + a0.step_size * np.sin(cand_angle))
        else:
            for i in act:
                try:
                    x, y, a = self.agents[i].headings_to_test()
                except StopNavigationException as e:              # out of the landscape before anything is sensed
                    self._stop(i, e)
                    continue
                idx.append(i); xs.append(x); ys.append(y); angs.append(a)
            if idx:
                angs = np.stack(angs)
        if idx:
            stops = {}
            results = self.engine.sense_step_batch(xs, ys, angs)
            # the records as arrays when the engine offers them (engine.BatchResults): no dictionary per agent and step
            lean = hasattr(results, "angle_familiarity")
            flags = results.flags.tolist() if lean else [r["flags"] for r in results]
            best = results.best_idex.tolist() if lean else None
            for k, i in enumerate(idx):
                if flags[k] & 16:                                 # DV_RES_SENSE_ERROR: this agent's footprint left the
                    # landscape (a corner reaches r*sqrt(2) > r past the bounds test); the reference's trial ends in an
                    # IndexError, the other trials go on
                    if cands is not None:
                        self.agents[i].angle_familiarity[:] = np.nan      # (headings_to_test's reset, :285)
                    self._stop(i, IndexError("sensor footprint reaches past the end of the landscape (index out of bounds)"))
                    continue
                try:
                    if lean:
                        self.agents[i].apply_step_arrays(results.angle_familiarity[k], best[k], fake,
                                                         None if cands is None else (cands[0][k], cands[1][k], cands[2][k]))
                    else:
                        self.agents[i].apply_step_result(results[k], fake)
                except StopNavigationException as e:
                    stops[i] = e
            # the members' error metrics, all in one device call; the reference takes them BEFORE its end-of-path test (:324-328), so
            # "too far from the path" wins where both would stop a member
            self._flush_errors()
            for i in idx:
                e = self._too_far.pop(id(self.agents[i]), None) or stops.get(i)
                if e is not None and self.stop_status[i] == 0:
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
