"""ctypes binding of libdejavu_hip.so (C ABI: include/dejavu.h).

There is no CPU fallback: if the HIP library is missing or cannot be loaded this module raises,
and so does everything that scores views.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libdejavu_hip.so")

DV_MAX_HEADINGS = 64
DV_MAX_HUE_PLANES = 4
DV_MAX_WIDE_HEADINGS = 4096
DV_STEP_FORCE_RESOLVE = 1
DV_STEP_WANT_SCENE = 2
DV_RES_RESOLVED = 1
DV_RES_EXACT_ALL = 2
DV_RES_OVERFLOW = 4
DV_RES_SENSE_ERROR = 16

ERROR_NAMES = {-1: "DV_ERR_INVALID", -2: "DV_ERR_HIP", -3: "DV_ERR_STATE", -4: "DV_ERR_OOM", -5: "DV_ERR_INDEX"}


class EngineError(RuntimeError):
    """A call into libdejavu_hip.so failed (HIP runtime error, bad state, out of memory)."""


class StepResult(ctypes.Structure):
    _fields_ = [
        ("best_heading", ctypes.c_int32),
        ("flags", ctypes.c_uint32),
        ("best_view", ctypes.c_int64),
        ("best_fam", ctypes.c_double),
        ("approx_max", ctypes.c_double),
        ("delta", ctypes.c_double),
        ("n_candidates", ctypes.c_int64),
        ("n_headings", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
        ("angle_fam", ctypes.c_double * DV_MAX_HEADINGS),
        ("angle_view", ctypes.c_int64 * DV_MAX_HEADINGS),
        ("exact_fam", ctypes.c_double * DV_MAX_HEADINGS),
        ("exact_view", ctypes.c_int64 * DV_MAX_HEADINGS),
    ]


class WideResult(ctypes.Structure):
    _fields_ = [
        ("best_heading", ctypes.c_int32),
        ("flags", ctypes.c_uint32),
        ("best_view", ctypes.c_int64),
        ("best_fam", ctypes.c_double),
        ("n_headings", ctypes.c_int32),
        ("n_passes", ctypes.c_int32),
        ("n_contending", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


class LibInfo(ctypes.Structure):
    _fields_ = [
        ("n_views", ctypes.c_int64),
        ("first_view", ctypes.c_int64),
        ("h", ctypes.c_int32),
        ("w", ctypes.c_int32),
        ("n_planes", ctypes.c_int32),
        ("n_hue_planes", ctypes.c_int32),
        ("generic_hue", ctypes.c_int32),
        ("has_value_plane", ctypes.c_int32),
        ("tile_bytes", ctypes.c_int64),
        ("chem_weight", ctypes.c_double),
        ("delta", ctypes.c_double),
        ("hues", ctypes.c_uint8 * DV_MAX_HUE_PLANES),
        ("n_hues", ctypes.c_int32),
        ("signed_saturation", ctypes.c_int32),
        ("bit_planes_hs", ctypes.c_int32),
        ("bit_planes_v", ctypes.c_int32),
        ("has_bit_planes", ctypes.c_int32),
        ("fp4_form", ctypes.c_int32),
        ("bit_tile_bytes", ctypes.c_int64),
        ("code_tile_bytes", ctypes.c_int64),
        ("mixed_layout", ctypes.c_int32),
        ("reserved0", ctypes.c_int32),
    ]


class MergeOut(ctypes.Structure):
    _fields_ = [
        ("best_heading", ctypes.c_int32),
        ("resolved", ctypes.c_int32),
        ("best_view", ctypes.c_int64),
        ("best_fam", ctypes.c_double),
        ("needs_resolve", ctypes.c_int32),
        ("n_contending", ctypes.c_int32),
        ("contending_mask", ctypes.c_uint64),
        ("angle_fam", ctypes.c_double * DV_MAX_HEADINGS),
    ]


_ctx_p = ctypes.c_void_p
_u8p = ctypes.POINTER(ctypes.c_uint8)
_f64p = ctypes.POINTER(ctypes.c_double)
_i64p = ctypes.POINTER(ctypes.c_int64)

# name -> (restype, argtypes); every symbol include/dejavu.h declares
PROTOTYPES = {
    "dv_create": (ctypes.c_int, [ctypes.POINTER(_ctx_p), ctypes.c_int]),
    "dv_destroy": (None, [_ctx_p]),
    "dv_last_error": (ctypes.c_char_p, [_ctx_p]),
    "dv_set_stream": (ctypes.c_int, [_ctx_p, ctypes.c_void_p]),
    "dv_set_exact": (ctypes.c_int, [_ctx_p, ctypes.c_int]),
    "dv_set_library": (ctypes.c_int, [_ctx_p, _u8p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_double, ctypes.c_int64]),
    "dv_generate_library": (ctypes.c_int, [_ctx_p, ctypes.c_uint64, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_double, ctypes.c_int64]),
    "dv_generate_library_ex": (ctypes.c_int, [_ctx_p, ctypes.c_uint64, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_double, ctypes.c_int64, ctypes.c_int]),
    "dv_append_library": (ctypes.c_int, [_ctx_p, _u8p, ctypes.c_int64, ctypes.c_int]),
    "dv_append_library_from_poses": (ctypes.c_int, [_ctx_p, _f64p, _f64p, _f64p, ctypes.c_int64, _u8p]),
    "dv_clear_library": (ctypes.c_int, [_ctx_p]),
    "dv_get_library_info": (ctypes.c_int, [_ctx_p, ctypes.POINTER(LibInfo)]),
    "dv_read_planes": (ctypes.c_int, [_ctx_p, ctypes.c_int64, ctypes.c_int64, _u8p]),
    "dv_bitplane_plan": (ctypes.c_int, [ctypes.POINTER(ctypes.c_uint32), ctypes.c_int, _u8p, _u8p,
                                        ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "dv_set_landscape": (ctypes.c_int, [_ctx_p, _u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "dv_configure_sensor": (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _u8p,
                                           ctypes.c_int]),
    "dv_sense": (ctypes.c_int, [_ctx_p, _f64p, _f64p, _f64p, ctypes.c_int, _u8p]),
    "dv_sense_patches": (ctypes.c_int, [_ctx_p, ctypes.c_double, ctypes.c_double, _f64p, ctypes.c_int]),
    "dv_sense_step": (ctypes.c_int, [_ctx_p, ctypes.c_double, ctypes.c_double, _f64p, ctypes.c_int, ctypes.c_uint32,
                                     ctypes.POINTER(StepResult), _f64p]),
    "dv_sense_step_batch": (ctypes.c_int, [_ctx_p, _f64p, _f64p, _f64p, ctypes.c_int, ctypes.c_int, ctypes.c_uint32,
                                           ctypes.POINTER(StepResult)]),
    "dv_agent_step": (ctypes.c_int, [_ctx_p, ctypes.c_double, ctypes.c_double, ctypes.c_double, _f64p, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_double, ctypes.c_double, ctypes.c_double, _f64p, ctypes.POINTER(ctypes.c_int32), _f64p,
                                     ctypes.POINTER(ctypes.c_int32)]),
    "dv_agent_step_begin": (ctypes.c_int, [_ctx_p, ctypes.c_double, ctypes.c_double, ctypes.c_double, _f64p, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_double, ctypes.c_double, ctypes.c_double, _f64p, ctypes.POINTER(ctypes.c_int32)]),
    "dv_agent_step_end": (ctypes.c_int, [_ctx_p, _f64p, ctypes.POINTER(ctypes.c_int32)]),
    "dv_agent_step_end_begin": (ctypes.c_int, [_ctx_p, _f64p, ctypes.POINTER(ctypes.c_int32), _f64p, _f64p, _f64p, _f64p, ctypes.c_int, _f64p,
                                               ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_int32), _f64p, ctypes.POINTER(ctypes.c_int32)]),
    "dv_set_library_from_poses": (ctypes.c_int, [_ctx_p, _f64p, _f64p, _f64p, ctypes.c_int64, ctypes.c_double,
                                                 ctypes.c_int64, _u8p]),
    "dv_set_training_path": (ctypes.c_int, [_ctx_p, _f64p, ctypes.c_int64]),
    "dv_path_error_enqueue": (ctypes.c_int, [_ctx_p, ctypes.c_double, ctypes.c_double, ctypes.c_double]),
    "dv_path_error_wait": (ctypes.c_int, [_ctx_p, _f64p]),
    "dv_path_coverage": (ctypes.c_int, [_ctx_p, _u8p, ctypes.c_int64]),
    "dv_path_reset": (ctypes.c_int, [_ctx_p]),
    "dv_path_slots": (ctypes.c_int, [_ctx_p, ctypes.c_int]),
    "dv_path_error_batch": (ctypes.c_int, [_ctx_p, ctypes.POINTER(ctypes.c_int32), _f64p, _f64p, ctypes.c_int, ctypes.c_double, _f64p]),
    "dv_path_coverage_slot": (ctypes.c_int, [_ctx_p, ctypes.c_int, _u8p, ctypes.c_int64]),
    "dv_path_reset_slot": (ctypes.c_int, [_ctx_p, ctypes.c_int]),
    "dv_score": (ctypes.c_int, [_ctx_p, _u8p, _f64p]),
    "dv_step": (ctypes.c_int, [_ctx_p, _u8p, ctypes.c_int, ctypes.c_uint32, ctypes.POINTER(StepResult), _f64p]),
    "dv_step_batch": (ctypes.c_int, [_ctx_p, _u8p, ctypes.c_int, ctypes.c_int, ctypes.c_uint32,
                                     ctypes.POINTER(StepResult)]),
    "dv_resolve": (ctypes.c_int, [_ctx_p, ctypes.POINTER(StepResult)]),
    "dv_step_wide": (ctypes.c_int, [_ctx_p, _u8p, ctypes.c_int, ctypes.c_uint32, ctypes.POINTER(WideResult), _f64p, _i64p, _f64p]),
    "dv_sense_step_wide": (ctypes.c_int, [_ctx_p, ctypes.c_double, ctypes.c_double, _f64p, ctypes.c_int, ctypes.c_uint32,
                                          ctypes.POINTER(WideResult), _f64p, _i64p, _f64p]),
    "dv_set_library_f32": (ctypes.c_int, [_ctx_p, ctypes.POINTER(ctypes.c_float), ctypes.c_int64, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int64]),
    "dv_generate_library_f32": (ctypes.c_int, [_ctx_p, ctypes.c_uint64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int64]),
    "dv_score_f32": (ctypes.c_int, [_ctx_p, ctypes.POINTER(ctypes.c_float), _f64p]),
    "dv_step_f32": (ctypes.c_int, [_ctx_p, ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.c_uint32,
                                   ctypes.POINTER(StepResult), _f64p]),
    "dv_set_library_u8": (ctypes.c_int, [_ctx_p, _u8p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int64]),
    "dv_score_u8": (ctypes.c_int, [_ctx_p, _u8p, _f64p]),
    "dv_step_u8": (ctypes.c_int, [_ctx_p, _u8p, ctypes.c_int, ctypes.c_uint32, ctypes.POINTER(StepResult), _f64p]),
    "dv_set_library_u8_from_poses": (ctypes.c_int, [_ctx_p, _f64p, _f64p, _f64p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, _u8p]),
    "dv_sense_step_u8": (ctypes.c_int, [_ctx_p, ctypes.c_double, ctypes.c_double, _f64p, ctypes.c_int, ctypes.c_int, ctypes.c_uint32,
                                        ctypes.POINTER(StepResult), _f64p]),
    "dv_upload_patches": (ctypes.c_int, [_ctx_p, _u8p, ctypes.c_int]),
    "dv_generate_patches": (ctypes.c_int, [_ctx_p, ctypes.c_uint64, ctypes.c_int]),
    "dv_step_enqueue": (ctypes.c_int, [_ctx_p, ctypes.c_uint32]),
    "dv_step_wait": (ctypes.c_int, [_ctx_p, ctypes.POINTER(StepResult), _f64p]),
    "dv_step_record": (ctypes.c_int, [_ctx_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int)]),
    "dv_resolve_enqueue": (ctypes.c_int, [_ctx_p]),
    "dv_step_keys": (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
                                    ctypes.POINTER(ctypes.c_int)]),
    "dv_merge_keys": (ctypes.c_int, [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                     ctypes.c_int, ctypes.POINTER(MergeOut)]),
    "dv_publish": (ctypes.c_int, [_ctx_p, ctypes.c_void_p, ctypes.c_int64]),
    "dv_publish_wait": (ctypes.c_int, [_ctx_p, _f64p, ctypes.c_int64]),
    "dv_set_mailbox": (ctypes.c_int, [_ctx_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    "dv_mailbox_post": (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_uint64]),
    "dv_mailbox_wait": (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, _f64p, ctypes.c_int64, ctypes.c_int]),
    "dv_merge_records": (ctypes.c_int, [_f64p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_double,
                                        ctypes.POINTER(MergeOut)]),
    "dv_group_create": (ctypes.c_int, [ctypes.POINTER(_ctx_p), ctypes.POINTER(ctypes.c_int), ctypes.c_int]),
    "dv_group_destroy": (None, [_ctx_p]),
    "dv_group_last_error": (ctypes.c_char_p, [_ctx_p]),
    "dv_group_size": (ctypes.c_int, [_ctx_p]),
    "dv_group_member": (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.POINTER(_ctx_p), _i64p, _i64p]),
    "dv_group_set_library": (ctypes.c_int, [_ctx_p, _u8p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double]),
    "dv_group_set_landscape": (ctypes.c_int, [_ctx_p, _u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "dv_group_configure_sensor": (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _u8p, ctypes.c_int]),
    "dv_group_score": (ctypes.c_int, [_ctx_p, _u8p, _f64p]),
    "dv_group_step": (ctypes.c_int, [_ctx_p, _u8p, ctypes.c_int, ctypes.c_uint32, ctypes.POINTER(StepResult), _f64p]),
    "dv_group_sense_step": (ctypes.c_int, [_ctx_p, ctypes.c_double, ctypes.c_double, _f64p, ctypes.c_int, ctypes.c_uint32,
                                           ctypes.POINTER(StepResult), _f64p]),
    "dv_synchronize": (ctypes.c_int, [_ctx_p]),
    "dv_timer_start": (ctypes.c_int, [_ctx_p]),
    "dv_timer_stop": (ctypes.c_int, [_ctx_p, ctypes.POINTER(ctypes.c_float)]),
    "dv_profile_kernel": (ctypes.c_int, [_ctx_p, ctypes.c_int]),
    "dv_profile_read": (ctypes.c_int, [_ctx_p, _f64p, _i64p]),
    "dv_workgroup_shape": (ctypes.c_int, [_ctx_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    "dv_stream_read_gbps": (ctypes.c_int, [_ctx_p, ctypes.c_int64, ctypes.c_int, _f64p]),
    "dv_range_push": (ctypes.c_int, [ctypes.c_char_p]),
    "dv_range_pop": (ctypes.c_int, []),
    "dv_patches_on_level": (ctypes.c_int, [ctypes.c_void_p]),
    "dv_fp4_plan": (ctypes.c_int, [ctypes.POINTER(ctypes.c_uint32), ctypes.c_int, ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint8),
                    ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_int)]),
    "dv_scoring_form": (ctypes.c_int, [ctypes.c_void_p]),
    "dv_version": (ctypes.c_char_p, []),
}

_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so; if libdejavu_hip.so pulls in the system
    one first and torch is imported afterwards (the multi-GPU exchange needs it), the process ends up with two
    runtimes and torch finds "No HIP GPUs".  Loading torch's copy first -- without importing torch -- makes both
    resolve to the same library, as happens anyway whenever torch is imported first."""
    try:
        with open("/proc/self/maps") as f:
            if "libamdhip64" in f.read():
                return                                   # a runtime is already in: whoever loaded it decides
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(path):
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    except Exception:                                    # noqa: BLE001 - best effort; the system runtime still works alone
        pass


def load():
    """Load libdejavu_hip.so and declare its prototypes.  Raises EngineError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    _share_torch_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise EngineError(
            "HIP library not built: %s is missing (run `make -C navigation-by-deja-vu_amd/csrc` or "
            "__graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise EngineError("cannot load %s: %s" % (LIB_PATH, e)) from e
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)      # AttributeError here = header and library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def as_u8(a, what):
    a = np.ascontiguousarray(a)
    if a.dtype != np.uint8:
        # mirrors the reference's "Buffer dtype mismatch" ValueError (navsim/util.pyx:31-33)
        raise ValueError("Buffer dtype mismatch for %s, expected 'uint8_t' but got '%s'" % (what, a.dtype))
    return a


def u8ptr(a):
    return a.ctypes.data_as(_u8p)


def f64ptr(a):
    return a.ctypes.data_as(_f64p)


def i64ptr(a):
    return a.ctypes.data_as(_i64p)


# byte offset of the per-heading arrays inside a step record (angle_fam, angle_view, exact_fam, exact_view follow one another)
RESULT_ARRAYS_OFFSET = StepResult.angle_fam.offset
