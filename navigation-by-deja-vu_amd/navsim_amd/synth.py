"""Seeded synthetic views, patches and landscapes (counter-based, NumPy-version independent).

Every byte is a pure function of (seed, global pixel counter), via the splitmix64 finaliser,
so the same library can be regenerated bit-for-bit by NumPy here and by the HIP generator
(`dv_generate_library`, csrc/dejavu_hip.hip: k_generate_tiles) on the GPU box without shipping
hundreds of megabytes.  The value distribution follows the reference's experiments
(SURVEY.md section 8d):

  V  uniform over the five levels n_sensor_levels=5 produces, {0,63,127,191,255}
     (navsim/NavBySceneFamiliarity.py:176-186: float32 rint then truncating uint8 cast)
  H  one of two chemicals, {0,127}   (scripts/run_experiment.py:131: integers(n)*(255//n))
  S  127 inside a "grain", else 0    (scripts/run_experiment.py:192: concentration_range=(127,128))
"""
import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
V_LEVELS = np.array([0, 63, 127, 191, 255], dtype=np.uint8)


def splitmix64(x):
    """splitmix64 output function on a uint64 array (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        z = np.asarray(x, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def _pixel_words(seed, first_pixel, n_pixels):
    with np.errstate(over="ignore"):
        idx = np.arange(n_pixels, dtype=np.uint64) + np.uint64(first_pixel)
        return splitmix64(idx + np.uint64(seed) * GOLDEN)


def hsv_from_words(z, full_range_s=False):
    """Map hash words to (H, S, V) bytes; same bit fields as k_generate_tiles.  full_range_s: the saturation takes every
    value 0..127 (a swept concentration range, scripts/run_experiment.py:132,192) instead of {0, 127}."""
    lvl = ((z & np.uint64(0xFFFF)) * np.uint64(5)) >> np.uint64(16)
    v = V_LEVELS[lvl.astype(np.int64)]
    h = (((z >> np.uint64(16)) & np.uint64(1)) * np.uint64(127)).astype(np.uint8)
    if full_range_s:
        s = ((z >> np.uint64(17)) & np.uint64(0x7F)).astype(np.uint8)
    else:
        s = (((z >> np.uint64(17)) & np.uint64(1)) * np.uint64(127)).astype(np.uint8)
    return h, s, v


def synth_views(seed, n_views, h, w, first_view=0, full_range_s=False):
    """uint8[n_views, h, w, 3] HSV views; view f depends only on (seed, first_view + f)."""
    npx = h * w
    out = np.empty((n_views, npx, 3), dtype=np.uint8)
    # generate in slabs to bound temporary memory
    slab = max(1, (1 << 22) // npx)
    for f0 in range(0, n_views, slab):
        f1 = min(n_views, f0 + slab)
        z = _pixel_words(seed, (first_view + f0) * npx, (f1 - f0) * npx)
        hh, ss, vv = hsv_from_words(z, full_range_s)
        o = out[f0:f1].reshape(-1, 3)
        o[:, 0] = hh
        o[:, 1] = ss
        o[:, 2] = vv
    return out.reshape(n_views, h, w, 3)


def synth_views_f32(seed, n_views, h, w, first_view=0):
    """float32[n_views, h, w] views for the ssd_f32 metric: pixel value = top 24 bits of its hash word / 2**24, in [0, 1)
    and exact in float32 -- the same values as the HIP generator (dv_generate_library_f32: k_generate_tiles_f32)."""
    npx = h * w
    out = np.empty((n_views, npx), dtype=np.float32)
    slab = max(1, (1 << 22) // npx)
    for f0 in range(0, n_views, slab):
        f1 = min(n_views, f0 + slab)
        z = _pixel_words(seed, (first_view + f0) * npx, (f1 - f0) * npx)
        out[f0:f1] = ((z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)).reshape(f1 - f0, npx)
    return out.reshape(n_views, h, w)


def synth_patches(seed, n_headings, h, w, full_range_s=False):
    """uint8[A, h, w, 3] sensor patches: the same distribution, stream seed+1."""
    return synth_views(seed + 1, n_headings, h, w, full_range_s=full_range_s)


def near_match_patch(view, seed, fraction=0.01):
    """Copy of `view` with about `fraction` of its pixels replaced by fresh random pixels."""
    h, w, _ = view.shape
    z = _pixel_words(seed ^ 0x5151, 0, h * w)
    pick = (z >> np.uint64(40)) % np.uint64(10000) < np.uint64(int(fraction * 10000))
    hh, ss, vv = hsv_from_words(splitmix64(z))
    out = view.reshape(-1, 3).copy()
    out[pick, 0] = hh[pick]
    out[pick, 1] = ss[pick]
    out[pick, 2] = vv[pick]
    return out.reshape(h, w, 3)


def random_hsv(seed, shape):
    """Uniform random uint8 array of `shape` (full-range H, S, V: exercises the generic-hue path)."""
    n = int(np.prod(shape))
    z = _pixel_words(seed, 0, (n + 7) // 8)
    return z.view(np.uint8)[:n].reshape(shape).copy()


def synth_landscape(seed, size, grain=4):
    """uint8[size, size, 3] HSV landscape of square grains of `grain` px.

    Each grain cell: V in {0,255} (Bernoulli 0.5, like navsim/generate_landscapes.py:66-72),
    bright cells carry a chemical: H in {0,127}, S=127 (scripts/run_experiment.py:126-142).
    """
    ncell = (size + grain - 1) // grain
    z = _pixel_words(seed, 0, ncell * ncell).reshape(ncell, ncell)
    bright = ((z >> np.uint64(8)) & np.uint64(1)).astype(bool)
    hue = (((z >> np.uint64(16)) & np.uint64(1)) * np.uint64(127)).astype(np.uint8)
    cell = np.zeros((ncell, ncell, 3), dtype=np.uint8)
    cell[..., 2] = np.where(bright, 255, 0)
    cell[..., 0] = np.where(bright, hue, 0)
    cell[..., 1] = np.where(bright, 127, 0)
    land = np.repeat(np.repeat(cell, grain, axis=0), grain, axis=1)[:size, :size]
    return np.ascontiguousarray(land)


def sin_training_path(curveness, start_x, length, arclen=2.0):
    """Points spaced `arclen` apart along y = x - (L/2)*c*sin(2*pi*(x - L/2 - x0)/L).

    Same construction as the reference's experiment driver (scripts/run_experiment.py:95-105):
    sample the curve densely (4 samples per arclen of x), then pick the dense sample at each
    multiple of `arclen` of cumulative chord length.
    """
    n_dense = 4 * int(np.floor(length / arclen))
    half = 0.5 * length
    xs = np.linspace(start_x, start_x + length, n_dense)
    ys = xs - half * curveness * np.sin((xs - half - start_x) * np.pi / half)
    dx, dy = np.diff(xs), np.diff(ys)
    chord = np.cumsum(np.sqrt(dx * dx + dy * dy))
    targets = arclen * np.arange(np.floor(chord[-1] / arclen))
    pick = np.searchsorted(chord, targets)
    return np.stack((xs[pick], ys[pick]), axis=1)
