"""ctypes binding of libmoby_hip_io.so (include/moby_hip_io.h): Moby XML scenes,
regress rows and compare-trajs.  Host only."""
import ctypes
import os

import numpy as np

from . import scene as S

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmoby_hip_io.so")
MH_IO_ID_LEN = 64


class mh_io_scene(ctypes.Structure):
    _fields_ = [("scene", S.mh_scene),
                ("state", ctypes.c_double * (S.MH_MAX_BODIES * S.MH_BODY_STATE)),
                ("body_id", (ctypes.c_char * MH_IO_ID_LEN) * (S.MH_MAX_BODIES + 1)),
                ("step_size", ctypes.c_double)]


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("moby_amd.io: %s not found -- build it with `make -C moby_amd/host` (or __graft_entry__.build())" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        lib.mh_io_load_xml.restype = ctypes.c_int
        lib.mh_io_load_xml.argtypes = [ctypes.c_char_p, ctypes.POINTER(mh_io_scene)]
        lib.mh_io_last_error.restype = ctypes.c_char_p
        lib.mh_io_format_row.restype = ctypes.c_int
        lib.mh_io_format_row.argtypes = [ctypes.c_double, ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
        lib.mh_io_compare_trajs.restype = ctypes.c_int
        lib.mh_io_compare_trajs.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_double,
                                            ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        _lib = lib
    return _lib


class SceneError(RuntimeError):
    pass


def load_xml(path):
    """-> (mh_scene, state (1, nb*13), body ids (enabled bodies in id order, then the ground), step size)."""
    lib = load()
    io = mh_io_scene()
    if lib.mh_io_load_xml(os.fsencode(path), ctypes.byref(io)) != 0:
        raise SceneError(lib.mh_io_last_error().decode("utf-8", "replace"))
    sc = S.mh_scene()
    ctypes.memmove(ctypes.addressof(sc), ctypes.addressof(io.scene), ctypes.sizeof(S.mh_scene))
    nb = sc.nb
    st = np.array(io.state[:nb * S.MH_BODY_STATE], dtype=np.float64).reshape(1, nb * S.MH_BODY_STATE)
    ids = [io.body_id[b].value.decode() for b in range(nb + sc.has_ground)]
    return sc, st, ids, io.step_size


def format_row(t, state, nb):
    lib = load()
    st = np.ascontiguousarray(state, dtype=np.float64)
    buf = ctypes.create_string_buffer(8192)
    lib.mh_io_format_row(float(t), st.ctypes.data, int(nb), buf, len(buf))
    return buf.value.decode()


def compare_trajs(f1, f2, tol):
    """-> (rc, max_diff, (timing1, timing2)); rc 0 = within tol, 1 = larger, -1 = error."""
    lib = load()
    md = ctypes.c_double(0.0)
    tm = (ctypes.c_double * 2)()
    rc = lib.mh_io_compare_trajs(os.fsencode(f1), os.fsencode(f2), float(tol), ctypes.byref(md), tm)
    return rc, md.value, (tm[0], tm[1])
