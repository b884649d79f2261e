"""``SimEngine``: ``yaqs_amd.engine.BatchEngine`` bound to tests/hipsim/_build/libtjm_sim.so (TEST INFRASTRUCTURE ONLY).

libtjm_sim.so is the device code of yaqs_amd/csrc compiled for the host against tests/hipsim (see the header of
tests/hipsim/hip/hip_runtime.h): the same kernels, launch code and C ABI, executed by an interpreter of the HIP execution model.
It lets the CPU suite run the engine's code paths (at small sizes) when no MI355X is at hand; it is not a backend of the package -
nothing under yaqs_amd/ knows about it - and the ``-m gpu`` tests remain the parity tests proper.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "hipsim"))

from yaqs_amd import _lib  # noqa: E402
from yaqs_amd.engine import BatchEngine, _i32  # noqa: E402

_sim: dict = {}


def load_sim(dtype: str = "complex128") -> C.CDLL:
    """The simulated-device library of the fp64 build, or of the complex64 build (-DTJM_F32) for ``dtype="complex64"``."""
    if dtype not in _sim:
        import build as hipsim_build  # tests/hipsim/build.py

        lib = C.CDLL(hipsim_build.build(f32=(dtype == "complex64")))
        for name, (res, args) in _lib.EXPORTS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _sim[dtype] = lib
    return _sim[dtype]


class _NoStream:
    cuda_stream = 0

    def synchronize(self):
        pass


class SimEngine(BatchEngine):
    """Same methods as BatchEngine (they only use ``self.lib`` / ``self.h``); construction binds host memory instead of HBM."""

    CANARY = 4096

    def __init__(self, length, chi_max, batch, mpo, device="cpu", d=2, stream=None, cap_slack=1, dtype="complex128"):
        self.torch = None
        self.dtype = dtype
        self.lib = load_sim(dtype)
        self.L, self.d, self.chi_max, self.B = int(length), int(d), int(chi_max), int(batch)
        self.device = device
        self.mpo_bonds = _i32([int(mpo[0].shape[2])] + [int(w.shape[3]) for w in mpo])
        h = C.c_void_p()
        _lib.check(self.lib.tjm_engine_create_ex(C.byref(h), self.L, self.d, self.chi_max, self.B, self.mpo_bonds.ctypes.data, int(cap_slack)), "create")
        self.h = h
        nbytes = self.lib.tjm_engine_workspace_bytes(self.h)
        self.workspace_bytes = int(nbytes)
        self.stream = _NoStream()
        # the workspace sits between two canaries: a kernel that writes outside what the engine said it needs is caught at close()
        self.ws = np.zeros(nbytes + 256 + 2 * self.CANARY, dtype=np.uint8)
        base = (self.ws.ctypes.data + self.CANARY + 255) // 256 * 256
        self._lo, self._hi = base - self.ws.ctypes.data, base - self.ws.ctypes.data + nbytes
        self.ws[: self._lo] = 0xA5
        self.ws[self._hi:] = 0xA5
        _lib.check(self.lib.tjm_engine_bind(self.h, base, nbytes, None), "bind")
        packed = np.concatenate([np.ascontiguousarray(w, dtype=np.complex128).reshape(-1) for w in mpo])
        _lib.check(self.lib.tjm_engine_set_mpo(self.h, packed.ctypes.data), "set_mpo")
        self.mpo_tensors = [np.array(w, dtype=np.complex128) for w in mpo]
        caps = np.zeros(self.L + 1, dtype=np.int32)
        self.lib.tjm_engine_bond_caps(self.h, caps.ctypes.data)
        self.caps = caps
        self.padded_elems = int(self.lib.tjm_engine_padded_state_elems(self.h))

    @staticmethod
    def workspace_bytes_for(length, chi_max, batch, mpo, d=2, cap_slack=1, dtype="complex128"):
        lib = load_sim(dtype)
        bonds = _i32([int(mpo[0].shape[2])] + [int(w.shape[3]) for w in mpo])
        h = C.c_void_p()
        _lib.check(lib.tjm_engine_create_ex(C.byref(h), int(length), int(d), int(chi_max), int(batch), bonds.ctypes.data, int(cap_slack)), "create")
        try:
            return int(lib.tjm_engine_workspace_bytes(h))
        finally:
            lib.tjm_engine_destroy(h)

    def close(self):
        if getattr(self, "h", None):
            self.lib.tjm_engine_destroy(self.h)
            self.h = None
            intact = bool(np.all(self.ws[: self._lo] == 0xA5) and np.all(self.ws[self._hi:] == 0xA5))
            self.ws = None
            if not intact:
                raise AssertionError("a kernel wrote outside the workspace the engine asked for")

    def synchronize(self):
        pass
