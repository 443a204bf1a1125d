"""ctypes binding of libhydro.so (the C ABI declared in include/hydro.h).

The shared library is built in-tree by `silver2_isaacsim_amd.build.build()`
(`__graft_entry__.build()` calls it) into `silver2_isaacsim_amd/lib/`.  There is
no CPU fallback: if the library is missing or fails to load, importing the
binding raises, and every product entry point needs it.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libhydro.so")

STATE_FIELDS, PREV_FIELDS, PARAM_FIELDS, WRENCH_FIELDS, COMP_FIELDS = 13, 6, 11, 6, 24
TILE = 64
BATCH_MAX = 32


class Scene(ctypes.Structure):
    """hydro_scene_t (include/hydro.h): one scene of a hydro_step_wrench_tiled_batch launch."""
    _fields_ = [("engine", c_void_p), ("n", c_int64),
                ("state", c_void_p), ("state_tile_stride", c_int64),
                ("prev", c_void_p), ("prev_tile_stride", c_int64),
                ("wrench", c_void_p), ("wrench_tile_stride", c_int64)]


HYDRO_OK = 0
HYDRO_SEM_NUMBA, HYDRO_SEM_WARP = 0, 1
STATUS_NAMES = {0: "HYDRO_OK", -1: "HYDRO_E_ARG", -2: "HYDRO_E_ALLOC", -3: "HYDRO_E_LAUNCH",
                -4: "HYDRO_E_DEVICE", -5: "HYDRO_E_STATE"}

# every symbol include/hydro.h declares: (restype, argtypes)
_FP = POINTER(c_void_p)      # table of field pointers (const float *const [N])
SIGNATURES = {
    "hydro_version": (c_int, []),
    "hydro_status_string": (c_char_p, [c_int]),
    "hydro_device_count": (c_int, [POINTER(c_int)]),
    "hydro_create": (c_int, [c_int, c_int64, POINTER(c_void_p)]),
    "hydro_destroy": (c_int, [c_void_p]),
    "hydro_last_error": (c_char_p, [c_void_p]),
    "hydro_capacity": (c_int64, [c_void_p]),
    "hydro_set_scene": (c_int, [c_void_p, c_double, c_double]),
    "hydro_set_semantics": (c_int, [c_void_p, c_int]),
    "hydro_set_params_f32": (c_int, [c_void_p, c_int64, _FP, c_int]),
    "hydro_set_params_f16": (c_int, [c_void_p, c_int64, _FP, c_int]),
    "hydro_reset_prev_velocity": (c_int, [c_void_p]),
    "hydro_get_prev_velocity": (c_int, [c_void_p, c_int64, _FP, c_int]),
    "hydro_set_prev_velocity": (c_int, [c_void_p, c_int64, _FP, c_int]),
    "hydro_step_wrench": (c_int, [c_void_p, c_int64, _FP, c_double, _FP, c_void_p]),
    "hydro_step_wrench_ext": (c_int, [c_void_p, c_int64, _FP, _FP, c_double, _FP, c_void_p]),
    "hydro_step_wrench_tiled": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_double,
                                        c_void_p, c_int64, c_void_p]),
    "hydro_step_wrench_tiled_batch": (c_int, [c_int, POINTER(Scene), c_double, c_void_p]),
    "hydro_step_wrench_tiled_ke": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_double,
                                           c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "hydro_step_fused_tiled": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_double,
                                       c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p]),
    "hydro_step_fused_tiled_ke": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_double,
                                          c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "hydro_step_fused_tiled_multi": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_double, c_int,
                                             c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "hydro_reserve_soa": (c_int, [c_void_p]),
    "hydro_integrate_tiled": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_double,
                                      c_void_p, c_int64, c_void_p]),
    "hydro_pack_state_aos": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p]),
    "hydro_unpack_wrench_aos": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "hydro_repack": (c_int, [c_void_p, c_int64, c_int, _FP, c_void_p, c_int64, c_int, c_void_p]),
    "hydro_step_wrench_aos": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_double,
                                      c_void_p, c_void_p, c_void_p]),
    "hydro_step_components": (c_int, [c_void_p, c_int64, _FP, _FP, _FP, c_void_p, c_void_p]),
    "hydro_step_components_aos": (c_int, [c_void_p, c_int64] + [c_void_p] * 6 + [_FP, c_void_p, c_void_p]),
    "hydro_kinetic_energy": (c_int, [c_void_p, c_int64, _FP, c_int, c_void_p, c_void_p]),
    "hydro_kinetic_energy_tiled": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "hydro_ke_allreduce": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "hydro_bind_rccl": (c_int, [c_void_p, c_void_p]),
    "hydro_rccl_origin": (c_char_p, []),
    "hydro_ke_rearm": (c_int, [c_void_p, c_void_p]),
    "hydro_debug_ke_fault": (c_int, [c_void_p, c_int, ctypes.c_uint32, c_int]),
    "hydro_integrate": (c_int, [c_void_p, c_int64, _FP, _FP, c_double, _FP, c_void_p]),
    "hydro_set_tuning": (c_int, [c_void_p, c_int, c_int, c_int, c_int]),
    "hydro_sync": (c_int, [c_void_p]),
    "hydro_stream": (c_void_p, [c_void_p]),
}

_lib = None


class HydroError(RuntimeError):
    """Non-zero status from libhydro.  A RuntimeError on purpose: the reference plugin's
    state-fetch guard treats RuntimeError as "skip this step"
    (hydrodynamics_behavior.py:191-192)."""

    def __init__(self, status: int, message: str):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


def load(path: str | None = None) -> ctypes.CDLL:
    """Load libhydro.so and attach prototypes.  Raises OSError if it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("HYDRO_LIBRARY") or LIB_PATH     # HYDRO_LIBRARY: another build of libhydro.so (A/B runs)
    if not os.path.exists(p):
        raise OSError(f"{p} not found: the HIP extension is not built "
                      f"(run `python -c 'import __graft_entry__ as g; g.build()'`); there is no CPU fallback")
    # One HIP runtime per process: PyTorch ships its own libamdhip64.so, and whichever copy is mapped first serves
    # every later user of that SONAME.  If libhydro.so came first it would bring the system runtime in, torch would
    # then run on a runtime it was not built against, and device calls fail (hipGetDeviceCount -> HYDRO_E_DEVICE).
    # The Python host uses torch for memory and streams anyway, so load it first.
    import torch  # noqa: F401
    lib = ctypes.CDLL(p)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = restype
        fn.argtypes = argtypes
    if path is None:
        _lib = lib
    return lib


def pointer_table(ptrs) -> ctypes.Array:
    """ctypes array of raw addresses (ints) usable as `const float *const t[N]`."""
    arr = (c_void_p * len(ptrs))()
    for i, p in enumerate(ptrs):
        arr[i] = int(p)
    return arr
