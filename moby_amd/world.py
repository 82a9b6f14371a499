"""Host-side mirror of the reference's simulator loop for a batch of worlds.

``WorldBatch.step(dt, nsteps)`` plays the role of calling
``TimeSteppingSimulator::step(dt)`` (include/Moby/TimeSteppingSimulator.h:36)
``nsteps`` times on every world; ``regress_rows`` formats the state the way
``programs/regress.cpp:82-93`` prints it (t, then x y z qx qy qz qw per body).
"""
import ctypes

import numpy as np

from . import _lib
from . import scene as S


class WorldBatch:
    """B worlds of one scene; numpy host arrays in/out (copies through the library)."""

    def __init__(self, scene, state, aux=None, seed=1):
        self.scene = scene
        self.state = np.ascontiguousarray(state, dtype=np.float64)
        self.B = self.state.shape[0]
        assert self.state.shape[1] == scene.nb * S.MH_BODY_STATE
        self.aux = S.new_aux(self.B, seed) if aux is None else aux

    def step(self, dt, nsteps=1, want_traj=False):
        lib = _lib.load()
        traj = np.zeros((self.B, nsteps, self.scene.nb, 7)) if want_traj else None
        rc = lib.mh_world_step_batch(ctypes.addressof(self.scene), self.B, float(dt), int(nsteps),
                                     self.state.ctypes.data, self.aux.ctypes.data,
                                     None if traj is None else traj.ctypes.data)
        _lib.check(rc)
        return traj

    def regress_rows(self):
        """(B, 1 + 7*nb): current_time followed by the Euler coordinates of every body."""
        q = self.state.reshape(self.B, self.scene.nb, S.MH_BODY_STATE)[:, :, :7].reshape(self.B, -1)
        return np.concatenate([self.aux["time"][:, None], q], axis=1)


class WorldBatchDevice:
    """Same, with state/aux resident in HBM behind an ``mh_world_batch`` handle;
    ``step`` is asynchronous on the given (torch) stream."""

    def __init__(self, scene, state, seed=1, aux=None):
        lib = _lib.load()
        self.scene = scene
        self.B = state.shape[0]
        self.handle = ctypes.c_void_p()
        _lib.check(lib.mh_world_batch_create(ctypes.addressof(scene), self.B, ctypes.byref(self.handle)))
        st = np.ascontiguousarray(state, dtype=np.float64)
        aux = S.new_aux(self.B, seed) if aux is None else np.ascontiguousarray(aux)      # aux given: resume from a checkpoint
        _lib.check(lib.mh_world_batch_upload(self.handle, st.ctypes.data, aux.ctypes.data))

    def step(self, dt, nsteps=1, stream=None, traj_ptr=None):
        _lib.check(_lib.load().mh_world_batch_step(self.handle, stream, float(dt), int(nsteps), traj_ptr))

    def step_ids(self, dt, nsteps, ids_dev_ptr, count, stream=None):
        """nsteps of the worlds listed in a DEVICE int32 array (e.g. a torch tensor's data_ptr()) on `stream` (mh_world_batch_step_ids)."""
        _lib.check(_lib.load().mh_world_batch_step_ids(self.handle, stream, float(dt), int(nsteps), ctypes.c_void_p(int(ids_dev_ptr)), int(count)))

    def download(self):
        st = np.zeros((self.B, self.scene.nb * S.MH_BODY_STATE))
        aux = np.zeros(self.B, dtype=S.AUX_DTYPE)
        _lib.check(_lib.load().mh_world_batch_download(self.handle, st.ctypes.data, aux.ctypes.data))
        return st, aux

    def close(self):
        if self.handle:
            _lib.load().mh_world_batch_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass



def save_checkpoint(path, scene, state, aux):
    """Everything a resumed run needs: the scene record, the body states and the per-world solver state
    (``mh_world_aux``: rand() stream, _zlast / _z / _v with their sizes, current_time, status, counters).  The reference's
    XML pickle (programs/driver.cpp:224-232) keeps none of the solver state, so its resumed runs diverge in pivot sequence."""
    np.savez(path, scene=np.frombuffer(bytes(scene), dtype=np.uint8), state=np.ascontiguousarray(state, dtype=np.float64),
             aux=np.frombuffer(np.ascontiguousarray(aux).tobytes(), dtype=np.uint8), nworlds=np.int64(len(aux)))


def load_checkpoint(path):
    """Returns (scene, state, aux) as ``save_checkpoint`` stored them."""
    z = np.load(path)
    scene = S.mh_scene.from_buffer_copy(z["scene"].tobytes())
    aux = np.frombuffer(z["aux"].tobytes(), dtype=S.AUX_DTYPE).copy()
    assert len(aux) == int(z["nworlds"])
    return scene, z["state"].copy(), aux
