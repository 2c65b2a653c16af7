"""navsim_amd -- MI355X-native scene-familiarity engine behind the navsim API.

    from navsim_amd import NavBySceneFamiliarity, StopNavigationException, sads_familiarity

mirrors `from navsim import ...` of the reference (navsim/__init__.py:1-3,
scripts/run_experiment.py:84).  Scoring runs only on the GPU through libdejavu_hip.so
(include/dejavu.h); importing this package does not need a GPU, scoring does.
"""
from .agent import (NavBySceneFamiliarity, StopNavigationException, ReachedEndOfTrainingPathException,
                    NavigatingFailedException, TooFarFromTrainingPathException,
                    OutOfLandscapeBoundsException, fill_sensor_from, downscale_chem)
from .util import sads_familiarity, hip_sads_familiarity, ssd_familiarity
from .engine import FamiliarityEngine
from .group import FamiliarityGroup
from ._native import EngineError
from . import synth
from .experiment import run_experiment, run_ensemble, chop_path_to_len
from .ensemble import NavEnsemble

__all__ = [
    "NavBySceneFamiliarity", "StopNavigationException", "ReachedEndOfTrainingPathException",
    "NavigatingFailedException", "TooFarFromTrainingPathException", "OutOfLandscapeBoundsException",
    "sads_familiarity", "hip_sads_familiarity", "ssd_familiarity", "FamiliarityEngine", "FamiliarityGroup", "EngineError",
    "fill_sensor_from", "downscale_chem", "synth", "run_experiment", "run_ensemble", "chop_path_to_len", "NavEnsemble",
]
