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
    """Same, with state/aux resident in HBM (torch tensors); asynchronous on torch's current stream."""

    def __init__(self, scene, state, device="cuda", seed=1):
        import torch
        self.torch = torch
        self.scene = scene
        self.device = torch.device(device)
        self.B = state.shape[0]
        self.state = torch.from_numpy(np.ascontiguousarray(state, dtype=np.float64)).to(self.device)
        aux = S.new_aux(self.B, seed)
        self.aux = torch.from_numpy(aux.view(np.uint8).reshape(self.B, -1).copy()).to(self.device)

    def step(self, dt, nsteps=1, traj=None):
        lib = _lib.load()
        stream = self.torch.cuda.current_stream(self.device).cuda_stream
        rc = lib.mh_world_step_batch_dev(stream, ctypes.addressof(self.scene), self.B, float(dt), int(nsteps),
                                         self.state.data_ptr(), self.aux.data_ptr(),
                                         None if traj is None else traj.data_ptr())
        _lib.check(rc)

    def aux_host(self):
        return self.aux.cpu().numpy().view(S.AUX_DTYPE).reshape(self.B)
