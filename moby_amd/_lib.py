"""ctypes binding of libmoby_hip.so (the C ABI of include/moby_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or a call
fails this module raises.  (The CPU oracle under oracle/ is test
infrastructure and is never imported from here.)
"""
import ctypes
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MOBY_HIP_LIB") or os.path.join(_HERE, "libmoby_hip.so")   # override: kernel experiments (tools/variants.sh)

MH_OK = 0
MH_ERR_INVALID_ARG = -1
MH_ERR_UNSUPPORTED_N = -2
MH_ERR_HIP = -3
MH_ERR_NO_DEVICE = -4

MH_LCP_FAST, MH_LCP_FAST_REG, MH_LCP_LEMKE, MH_LCP_LEMKE_REG = 0, 1, 2, 3
MH_RAND_WORDS = 32
MH_LCP_MAX_N_WAVE = 64
MH_TRACE_ATTEMPT = 0x40000000


class MobyHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libmoby_hip error %d: %s" % (code, msg))
        self.code = code


class mh_lcp_opts(ctypes.Structure):
    _fields_ = [("min_exp", ctypes.c_int), ("step_exp", ctypes.c_uint), ("max_exp", ctypes.c_int),
                ("piv_tol", ctypes.c_double), ("zero_tol", ctypes.c_double)]


# every symbol include/moby_hip.h declares: name -> (restype, argtypes)
_vp, _i, _l, _d = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_double
_LCP_TAIL = [_i, _i, _vp, _i, _l, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, ctypes.POINTER(mh_lcp_opts)]
SYMBOLS = {
    "mh_version": (_i, []),
    "mh_last_error": (ctypes.c_char_p, []),
    "mh_device_count": (_i, []),
    "mh_device_get": (_i, []),
    "mh_device_set": (_i, [_i]),
    "mh_world_batch_device": (_i, [_vp]),
    "mh_world_batch_counters_dev": (_i, [_vp, _vp, _vp, _vp]),
    "mh_big_batch_device": (_i, [_vp]),
    "mh_artic_batch_device": (_i, [_vp]),
    "mh_impact_batch_device": (_i, [_vp]),
    "mh_rand_seed": (None, [_vp, ctypes.c_uint32]),
    "mh_rand_next": (_i, [_vp]),
    "mh_lcp_solve_batch_dev": (_i, [_vp, _i] + _LCP_TAIL),
    "mh_lcp_solve_batch": (_i, [_i] + _LCP_TAIL),
    "mh_debug_set": (_i, [_i, _i]),
    "mh_scene_defaults": (None, [_vp]),
    "mh_world_aux_init": (None, [_vp, ctypes.c_uint32]),
    "mh_world_batch_create": (_i, [_vp, _i, ctypes.POINTER(_vp)]),
    "mh_world_batch_destroy": (_i, [_vp]),
    "mh_world_batch_upload": (_i, [_vp, _vp, _vp]),
    "mh_world_batch_step": (_i, [_vp, _vp, _d, _i, _vp]),
    "mh_world_batch_step_ids": (_i, [_vp, _vp, _d, _i, _vp, _i]),
    "mh_world_batch_download": (_i, [_vp, _vp, _vp]),
    "mh_world_batch_occupancy": (_i, [_vp]),
    "mh_world_batch_profile": (_i, [_vp, _d, _i, _vp, _i]),
    "mh_world_profile_phase_count": (_i, []),
    "mh_world_batch_device_ptrs": (_i, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp)]),
    "mh_world_step_batch": (_i, [_vp, _i, _d, _i, _vp, _vp, _vp]),
    # include/moby_hip_impact.h
    "mh_impact_batch_create": (_i, [_i, _i, _i, _i, _vp, _vp, ctypes.POINTER(_vp)]),
    "mh_impact_batch_destroy": (_i, [_vp]),
    "mh_impact_batch_upload": (_i, [_vp, _vp, _vp]),
    "mh_impact_batch_process": (_i, [_vp, _vp]),
    "mh_impact_batch_download": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "mh_impact_batch_lcp_size": (_i, [_vp]),
    "mh_impact_batch_set_model": (_i, [_vp, _i]),
    "mh_impact_batch_debug_lcp": (_i, [_vp, _vp, _vp]),
    "mh_impact_batch_save_solver_state": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "mh_impact_batch_load_solver_state": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "mh_impact_batch_save_noslip_state": (_i, [_vp, _vp, _vp]),
    "mh_impact_batch_load_noslip_state": (_i, [_vp, _vp, _vp]),
    "mh_impact_batch_device_ptrs": (_i, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp)]),
    "mh_impact_process_batch": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    # include/moby_hip_stack.h
    "mh_big_batch_create": (_i, [_vp, _i, ctypes.POINTER(_vp)]),
    "mh_big_batch_destroy": (_i, [_vp]),
    "mh_big_batch_upload": (_i, [_vp, _vp, _vp]),
    "mh_big_batch_step": (_i, [_vp, _vp, _d, _i]),
    "mh_big_batch_stabilize": (_i, [_vp, _vp]),
    "mh_big_batch_download": (_i, [_vp, _vp, _vp]),
    "mh_big_batch_lcp_capacity": (_i, [_vp]),
    "mh_big_batch_lu_work": (_i, [_vp, _vp, _i]),
    "mh_impact_batch_lu_work": (_i, [_vp, _vp, _i]),
    "mh_big_batch_save_solver_state": (_i, [_vp, _vp, _vp, _vp]),
    "mh_big_batch_load_solver_state": (_i, [_vp, _vp, _vp, _vp]),
    # include/moby_hip_artic.h
    "mh_artic_batch_create": (_i, [_vp, _i, ctypes.POINTER(_vp)]),
    "mh_artic_batch_destroy": (_i, [_vp]),
    "mh_artic_batch_upload": (_i, [_vp, _vp, _vp, _vp]),
    "mh_artic_batch_step": (_i, [_vp, _vp, _d, _i]),
    "mh_artic_batch_fwd_dyn": (_i, [_vp, _vp, _vp, _vp]),
    "mh_artic_batch_download": (_i, [_vp, _vp, _vp, _vp]),
    "mh_artic_batch_link_poses": (_i, [_vp, _vp]),
    "mh_artic_batch_jacobian": (_i, [_vp, _i, _vp, _vp]),
}

_lib = None


def load():
    """Load libmoby_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "moby_amd: %s not found -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    # torch bundles its own HIP runtime: if this library were loaded first it would bring /opt/rocm's
    # copy into the process and torch's would come second -- two runtimes, and the second one to
    # initialise sees no device.  Loading torch first makes libmoby_hip bind to the runtime torch uses.
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    # test hook: MOBY_HIP_DEBUG="key=value,..." applies mh_debug_set switches (A/B runs of tools/ and bench legs in one process tree)
    for kv in filter(None, os.environ.get("MOBY_HIP_DEBUG", "").split(",")):
        k, v = kv.split("=")
        if lib.mh_debug_set(int(k), int(v)) != MH_OK:
            raise MobyHipError(MH_ERR_INVALID_ARG, "MOBY_HIP_DEBUG: mh_debug_set(%s) refused" % kv)
    return lib


def check(rc):
    if rc != MH_OK:
        raise MobyHipError(rc, load().mh_last_error().decode("utf-8", "replace"))
