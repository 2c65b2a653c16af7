"""Experiment loop around the agent: what one trial of the reference's farm runs and reports.

Counterpart of scripts/run_experiment.py:235-258 (`run_experiment`) and :107-124 (`chop_path_to_len`); the MPI farm,
PNG loading and grain labelling around them are out of scope.  Result keys and stop codes are the reference's.
"""
import numpy as np

from .agent import StopNavigationException

FRAME_FACTOR = 3.0               # frames = FRAME_FACTOR * training_path_length / step_size (scripts/run_experiment.py:21,238)
N_CONSECUTIVE_SCENES = 0.05      # window of the forgiving coverage / capture metrics (:28)


def chop_path_to_len(path, length):
    """Trim a polyline to at most `length`, dropping points from the front and the back alternately (:107-124)."""
    seg = np.linalg.norm(path[1:] - path[:-1], axis=1)
    assert np.sum(seg) >= length
    lo, hi = 0, len(path)
    drop_front = True
    while np.sum(seg[lo:hi]) > length and hi - lo > 0:
        if drop_front:
            lo += 1
        else:
            hi -= 1
        drop_front = not drop_front
    return path[lo:hi]


def run_experiment(nsf, frames=None):
    """Step the agent until it stops or `frames` run out; returns the reference's result row (:251-258)."""
    if frames is None:
        frames = int(FRAME_FACTOR * nsf.training_path_length / nsf.step_size)
    status = 0
    completed = 0
    try:
        for _ in range(frames):
            nsf.step_forward()
            completed += 1
    except StopNavigationException as stop:
        status = stop.get_code()
        nsf.stopped_with_exception = stop
    return {
        "path_coverage": nsf.percent_recapitulated,
        "rmsd_error": nsf.navigation_error,
        "completed_frames": completed,
        "stop_status": status,
        "percent_forgiving": nsf.percent_recapitulated_forgiving(n_consecutive_scenes=N_CONSECUTIVE_SCENES),
        "n_captures": nsf.n_captures(n_consecutive_scenes=N_CONSECUTIVE_SCENES),
    }


def run_ensemble(ens, frames=None):
    """run_experiment for a navsim_amd.NavEnsemble: every agent gets the reference's result row (:251-258)."""
    first = ens.agents[0]
    if frames is None:
        frames = int(FRAME_FACTOR * first.training_path_length / first.step_size)
    done = ens.run(frames)
    rows = []
    for i, nsf in enumerate(ens.agents):
        scored = nsf._n_navigation_error > 0                      # an agent stopped before its first step has no error yet
        rows.append({
            "path_coverage": nsf.percent_recapitulated,
            "rmsd_error": nsf.navigation_error if scored else float("nan"),
            "completed_frames": done[i],
            "stop_status": ens.stop_status[i],
            "percent_forgiving": nsf.percent_recapitulated_forgiving(n_consecutive_scenes=N_CONSECUTIVE_SCENES),
            "n_captures": nsf.n_captures(n_consecutive_scenes=N_CONSECUTIVE_SCENES),
        })
    return rows
