"""TEST DOUBLE of navsim_amd.FamiliarityEngine for CPU tests of the sharding protocol.

Reproduces the C ABI's step semantics (include/dejavu.h: integer-sum scores, candidate window `delta`,
exact re-scoring of candidates, flags) with NumPy + the oracle, so that navsim_amd/sharded.py can be
exercised with world_size > 1 over gloo on a machine without a GPU.  Never used by the product.
"""
import numpy as np

from oracle import oracle


class OracleBackedEngine(object):
    def __init__(self, always_resolve=False):
        self.always_resolve = always_resolve

    def set_library(self, scenes, chem_weight=0.0, first_view=0):
        self.lib = np.ascontiguousarray(scenes)
        self.cw = float(chem_weight)
        self.first = int(first_view)
        self.n_views = len(scenes)
        self.shape = scenes.shape[1:3]
        P = float(self.shape[0] * self.shape[1])
        self.delta = 4.0 * (P + 8.0) * 2.0 ** -53 * P          # dejavu_hip.hip: alloc_library

    def _approx(self, patches):
        P = float(self.shape[0] * self.shape[1])
        out = np.empty((len(patches), self.n_views))
        for a, p in enumerate(patches):
            s_hs, s_v = oracle.int_sums(self.lib, p)
            out[a] = P - ((0.5 * self.cw) * s_hs.astype(np.float64) + (1 - self.cw) * s_v.astype(np.float64)) / 255.
        return out

    def _exact(self, patches):
        return np.stack([oracle.sads_hsv(self.lib, p, self.cw) for p in patches])

    def step(self, patches, want_scene=False, force_resolve=False):
        self._patches = patches
        A = len(patches)
        fam = self._approx(patches)
        self._fam = fam
        gmax = fam.max()
        cand = np.argwhere(fam >= gmax - self.delta)
        self._cand = cand
        res = dict(angle_familiarity=fam.max(axis=1), angle_view=fam.argmax(axis=1) + self.first,
                   exact_familiarity=np.full(A, -np.inf), exact_view=np.full(A, -1, dtype=np.int64),
                   approx_max=float(gmax), delta=self.delta, n_candidates=len(cand), flags=0,
                   scene_familiarity=fam.min(axis=0) if want_scene else None)
        if len(cand) >= 2 or force_resolve or self.always_resolve:
            res = self._resolve_into(res)
        else:
            res["best_idex"] = int(np.argmax(res["angle_familiarity"]))
            res["best_view"] = int(res["angle_view"][res["best_idex"]])
            res["step_familiarity"] = float(res["angle_familiarity"][res["best_idex"]])
        self._last = res
        return res

    def step_batch(self, patches, force_resolve=False):
        return [self.step(p, force_resolve=force_resolve) for p in patches]

    def _resolve_into(self, res):
        res = dict(res)
        exact = self._exact(self._patches)
        ang = res["angle_familiarity"].copy()
        view = res["angle_view"].copy()
        ex = np.full(len(ang), -np.inf)
        exv = np.full(len(ang), -1, dtype=np.int64)
        for a in sorted(set(self._cand[:, 0])):
            fs = self._cand[self._cand[:, 0] == a][:, 1]
            vals = exact[a, fs]
            ex[a] = vals.max()
            exv[a] = fs[vals == vals.max()].min() + self.first
            ang[a], view[a] = ex[a], exv[a]
        best = int(np.argmax(ex))
        res.update(angle_familiarity=ang, angle_view=view, exact_familiarity=ex, exact_view=exv, flags=1,
                   best_idex=best, best_view=int(exv[best]), step_familiarity=float(ex[best]))
        return res

    def resolve(self):
        self._last = self._resolve_into(self._last)
        return self._last
