"""Python handle on the batched HIP engine (one GPU, B trajectories in lock-step).

PyTorch is used for device storage only: the workspace is one uint8 tensor whose
``data_ptr`` is handed to the C ABI.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np

from . import _lib

TRUNC_MODES = {"discarded_weight": 0, "relative": 1, "hard_cutoff": 2, "relative_discarded_weight": 3}


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


class BatchEngine:
    def __init__(self, length: int, chi_max: int, batch: int, mpo: Sequence[np.ndarray], device: str = "cuda:0", d: int = 2, stream=None,
                 cap_slack: int = 1, dtype: str = "complex128"):
        import torch

        if not torch.cuda.is_available():
            raise _lib.TjmError("yaqs_amd needs a HIP device (torch.cuda.is_available() is False); there is no CPU path")
        self.torch = torch
        self.dtype = dtype  # "complex128" (the reference's precision) or "complex64" (libtjm_hip_f32.so: fp32 arithmetic and storage)
        self.lib = _lib.load(dtype)
        self.L, self.d, self.chi_max, self.B = int(length), int(d), int(chi_max), int(batch)
        self.device = torch.device(device)
        bonds = [int(mpo[0].shape[2])] + [int(w.shape[3]) for w in mpo]
        self.mpo_bonds = _i32(bonds)
        h = C.c_void_p()
        _lib.check(self.lib.tjm_engine_create_ex(C.byref(h), self.L, self.d, self.chi_max, self.B, self.mpo_bonds.ctypes.data, int(cap_slack)), "create")
        self.h = h
        nbytes = self.lib.tjm_engine_workspace_bytes(self.h)
        self.workspace_bytes = int(nbytes)
        with torch.cuda.device(self.device):
            # every engine launches on ONE stream; engines on different streams overlap on the device (the SVD kernels are
            # VALU-bound, the contractions MFMA-bound: the two pipes of a CU run side by side)
            self.stream = stream if stream is not None else torch.cuda.current_stream(self.device)
            with torch.cuda.stream(self.stream):
                self.ws = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)  # zero fill ordered before the engine's work
            self.stream.synchronize()
        _lib.check(self.lib.tjm_engine_bind(self.h, self.ws.data_ptr(), nbytes, C.c_void_p(self.stream.cuda_stream)), "bind")
        packed = np.concatenate([np.ascontiguousarray(w, dtype=np.complex128).reshape(-1) for w in mpo])
        _lib.check(self.lib.tjm_engine_set_mpo(self.h, packed.ctypes.data), "set_mpo")
        self.mpo_tensors = [np.array(w, dtype=np.complex128) for w in mpo]  # the Hamiltonian the engine currently holds
        caps = np.zeros(self.L + 1, dtype=np.int32)
        self.lib.tjm_engine_bond_caps(self.h, caps.ctypes.data)
        self.caps = caps
        self.padded_elems = int(self.lib.tjm_engine_padded_state_elems(self.h))

    @staticmethod
    def workspace_bytes_for(length: int, chi_max: int, batch: int, mpo: Sequence[np.ndarray], d: int = 2, cap_slack: int = 1,
                            dtype: str = "complex128") -> int:
        """Device bytes an engine of this shape binds (no GPU needed): used to size the batch to the free HBM."""
        lib = _lib.load(dtype)
        bonds = _i32([int(mpo[0].shape[2])] + [int(w.shape[3]) for w in mpo])
        h = C.c_void_p()
        _lib.check(lib.tjm_engine_create_ex(C.byref(h), int(length), int(d), int(chi_max), int(batch), bonds.ctypes.data, int(cap_slack)), "create")
        try:
            return int(lib.tjm_engine_workspace_bytes(h))
        finally:
            lib.tjm_engine_destroy(h)

    def set_mpo(self, mpo: Sequence[np.ndarray]):
        """Replace the Hamiltonian (piecewise-constant drives, analog_tjm.py:43-49); bond dimensions must match the engine's."""
        bonds = _i32([int(mpo[0].shape[2])] + [int(w.shape[3]) for w in mpo])
        if len(mpo) != self.L or not np.array_equal(bonds, self.mpo_bonds):
            raise NotImplementedError("piecewise Hamiltonians must share the MPO bond dimensions the engine was created with")
        packed = np.concatenate([np.ascontiguousarray(w, dtype=np.complex128).reshape(-1) for w in mpo])
        _lib.check(self.lib.tjm_engine_set_mpo(self.h, packed.ctypes.data), "set_mpo")
        self.mpo_tensors = [np.array(w, dtype=np.complex128) for w in mpo]

    def close(self):
        if getattr(self, "h", None):
            self.lib.tjm_engine_destroy(self.h)
            self.h = None
            self.ws = None
            self.torch.cuda.empty_cache()  # hand the workspace back to the device: the next engine may be larger

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration ---------------------------------------------------------------
    def set_params(self, *, dt, svd_threshold, trunc_mode="discarded_weight", max_bond_dim=None, krylov_tol=1e-4, tdvp_mode="2site",
                   tdvp_sweeps=1):
        if tdvp_mode not in ("1site", "2site", "dynamic"):
            raise ValueError(f'tdvp_mode must be one of ("1site", "2site", "dynamic"), got {tdvp_mode!r}.')  # tdvp.py:109-111
        self.tdvp_mode = tdvp_mode
        _lib.check(self.lib.tjm_engine_set_params(self.h, float(dt), float(svd_threshold), TRUNC_MODES[trunc_mode],
                                                  -1 if max_bond_dim is None else int(max_bond_dim), float(krylov_tol),
                                                  1 if tdvp_mode == "1site" else 2, int(tdvp_sweeps)),  # "dynamic": the host drives the site steps
                   "set_params")

    def set_noise(self, processes, is_pauli_flags):
        n = len(processes)
        nsites = np.zeros(max(n, 1), dtype=np.int32)
        sites = np.zeros(2 * max(n, 1), dtype=np.int32)
        gamma = np.zeros(max(n, 1))
        pauli = np.zeros(max(n, 1), dtype=np.int32)
        dd = self.d * self.d  # a one-site operator has d^2 entries, an operator on an adjacent pair d^4 (16 for qubits)
        mats = np.zeros((max(n, 1), dd * dd), dtype=np.complex128)
        facs = np.zeros((max(n, 1), 2 * dd), dtype=np.complex128)
        hasf = np.zeros(max(n, 1), dtype=np.int32)
        for k, p in enumerate(processes):
            s = list(p["sites"])
            nsites[k] = len(s)
            sites[2 * k] = s[0]
            sites[2 * k + 1] = s[1] if len(s) > 1 else s[0]
            gamma[k] = p["strength"]
            pauli[k] = int(bool(is_pauli_flags[k]))
            if "matrix" in p:
                m = np.asarray(p["matrix"], dtype=np.complex128).reshape(-1)
                mats[k, : m.size] = m
            if "factors" in p:
                facs[k, :dd] = np.asarray(p["factors"][0], dtype=np.complex128).reshape(-1)
                facs[k, dd:] = np.asarray(p["factors"][1], dtype=np.complex128).reshape(-1)
                hasf[k] = 1
        _lib.check(self.lib.tjm_engine_set_noise(self.h, n, nsites.ctypes.data, sites.ctypes.data, gamma.ctypes.data, pauli.ctypes.data,
                                                 mats.ctypes.data, facs.ctypes.data, hasf.ctypes.data), "set_noise")

    def load_state(self, tensors: Sequence[np.ndarray], set_index: int = 0):
        bonds = _i32([tensors[0].shape[1]] + [t.shape[2] for t in tensors])
        packed = np.concatenate([np.ascontiguousarray(t, dtype=np.complex128).reshape(-1) for t in tensors])
        _lib.check(self.lib.tjm_engine_load_state(self.h, set_index, packed.ctypes.data, bonds.ctypes.data), "load_state")

    def load_state_slot(self, b: int, tensors: Sequence[np.ndarray], set_index: int = 0):
        """One trajectory slot only (the others keep their states): per-trajectory initial states."""
        bonds = _i32([tensors[0].shape[1]] + [t.shape[2] for t in tensors])
        packed = np.concatenate([np.ascontiguousarray(t, dtype=np.complex128).reshape(-1) for t in tensors])
        _lib.check(self.lib.tjm_engine_load_state_slot(self.h, set_index, int(b), packed.ctypes.data, bonds.ctypes.data), "load_state_slot")

    def copy_state(self, dst: int, src: int):
        _lib.check(self.lib.tjm_engine_copy_state(self.h, dst, src), "copy_state")

    def export_state(self, b: int, set_index: int = 0) -> list[np.ndarray]:
        buf = np.zeros(self.padded_elems, dtype=np.complex128)
        bonds = np.zeros(self.L + 1, dtype=np.int32)
        _lib.check(self.lib.tjm_engine_export_state(self.h, set_index, b, buf.ctypes.data, bonds.ctypes.data), "export_state")
        out, off = [], 0
        for i in range(self.L):
            cl, cr = int(self.caps[i]), int(self.caps[i + 1])
            t = buf[off: off + self.d * cl * cr].reshape(self.d, cl, cr)
            off += self.d * cl * cr
            out.append(t[:, : bonds[i], : bonds[i + 1]].copy())
        return out

    def set_uniforms(self, u: np.ndarray):
        u = np.ascontiguousarray(u, dtype=np.float64)
        assert u.shape[0] == self.B
        _lib.check(self.lib.tjm_engine_set_uniforms(self.h, u.ctypes.data, u.shape[1]), "set_uniforms")

    # -- the path --------------------------------------------------------------------
    def tdvp(self, set_index: int = 0):
        _lib.check(self.lib.tjm_engine_tdvp(self.h, set_index), "tdvp")

    def dissipate(self, dt: float, set_index: int = 0):
        _lib.check(self.lib.tjm_engine_dissipate(self.h, set_index, float(dt)), "dissipate")

    def dissipate_from(self, dt: float, center: int, set_index: int = 0):
        _lib.check(self.lib.tjm_engine_dissipate_from(self.h, set_index, float(dt), int(center)), "dissipate_from")

    def set_noise_filter(self, indices=None):
        if indices is None:
            _lib.check(self.lib.tjm_engine_set_noise_filter(self.h, -1, None), "set_noise_filter")
        else:
            idx = _i32(indices)
            _lib.check(self.lib.tjm_engine_set_noise_filter(self.h, len(idx), idx.ctypes.data), "set_noise_filter")

    def normalize_qr(self, center: int, set_index: int = 0):
        _lib.check(self.lib.tjm_engine_normalize_qr(self.h, set_index, int(center)), "normalize_qr")

    def apply_single(self, site: int, matrix: np.ndarray, set_index: int = 0):
        m = np.ascontiguousarray(matrix, dtype=np.complex128)
        _lib.check(self.lib.tjm_engine_apply_single(self.h, set_index, int(site), m.ctypes.data), "apply_single")

    def tebd_gate(self, left: int, u4: np.ndarray, set_index: int = 0, center: int = 0):
        u = np.ascontiguousarray(np.asarray(u4, dtype=np.complex128).reshape(self.d ** 2, self.d ** 2))
        _lib.check(self.lib.tjm_engine_tebd_gate_at(self.h, set_index, int(left), int(center), u.ctypes.data), "tebd_gate")

    def apply_pair(self, left: int, matrix: np.ndarray, min_keep: int = 1, set_index: int = 0):
        m = np.ascontiguousarray(np.asarray(matrix, dtype=np.complex128).reshape(self.d ** 2, self.d ** 2))
        _lib.check(self.lib.tjm_engine_apply_pair(self.h, set_index, int(left), m.ctypes.data, int(min_keep)), "apply_pair")

    def canonicalize_qr(self, center: int, set_index: int = 0):
        _lib.check(self.lib.tjm_engine_canonicalize_qr(self.h, set_index, int(center)), "canonicalize_qr")

    def stochastic(self, dt: float, set_index: int = 0):
        jumped = np.zeros(self.B, dtype=np.int32)
        dp = np.zeros(self.B)
        _lib.check(self.lib.tjm_engine_stochastic(self.h, set_index, float(dt), jumped.ctypes.data, dp.ctypes.data), "stochastic")
        return jumped, dp

    def site_moments(self, set_index: int = 0) -> np.ndarray:
        m = np.zeros((self.L, self.B, self.d, self.d), dtype=np.complex128)
        _lib.check(self.lib.tjm_engine_site_moments(self.h, set_index, m.ctypes.data), "site_moments")
        return m

    def site_moments2(self, set_index: int = 0):
        m = np.zeros((self.L, self.B, self.d, self.d), dtype=np.complex128)
        m2 = np.zeros((self.L - 1, self.B, self.d * self.d, self.d * self.d), dtype=np.complex128)
        _lib.check(self.lib.tjm_engine_site_moments2(self.h, set_index, m.ctypes.data, m2.ctypes.data), "site_moments2")
        return m, m2

    def bond_dims(self, set_index: int = 0) -> np.ndarray:
        chi = np.zeros((self.B, self.L + 1), dtype=np.int32)
        _lib.check(self.lib.tjm_engine_bond_dims(self.h, set_index, chi.ctypes.data), "bond_dims")
        return chi

    def site0_normsq(self, set_index: int = 0) -> np.ndarray:
        out = np.zeros(self.B)
        _lib.check(self.lib.tjm_engine_site0_normsq(self.h, set_index, out.ctypes.data), "site0_normsq")
        return out

    def run(self, *, order: int, n_times: int, sample_timesteps: bool, has_noise: bool, seed, traj_indices, observables,
            start=(0, 0), rng_pos=None, results=None, diagnostics=None, status=None):
        """Whole trajectories in one C call (tjm_engine_run).  observables: [(first_site, matrix d x d | d^2 x d^2)] in site-sorted order.

        ``start`` / ``rng_pos`` / ``results`` / ``diagnostics`` continue a run that stopped with ``CapacityError`` (whose
        ``resume`` and ``rng_pos`` attributes carry the values to pass) after ``adopt`` has moved the states to this engine."""
        n_obs = len(observables)
        nsites = np.zeros(max(n_obs, 1), dtype=np.int32)
        site = np.zeros(max(n_obs, 1), dtype=np.int32)
        dd = self.d * self.d
        mats = np.zeros((max(n_obs, 1), dd * dd), dtype=np.complex128)
        for k, (s0, m) in enumerate(observables):
            m = np.asarray(m, dtype=np.complex128)
            nsites[k] = 2 if m.size == dd * dd else 1
            site[k] = s0
            mats[k, : m.size] = m.reshape(-1)
        pos = np.zeros(self.B, dtype=np.int64) if rng_pos is None else np.ascontiguousarray(rng_pos, dtype=np.int64).copy()
        resume = np.zeros(2, dtype=np.int32)
        cfg = _lib.RunConfig(order=int(order), n_times=int(n_times), sample_timesteps=int(bool(sample_timesteps)), has_noise=int(bool(has_noise)),
                             has_seed=int(seed is not None), seed=int(seed or 0), n_obs=n_obs, obs_nsites=nsites.ctypes.data,
                             obs_site=site.ctypes.data, obs_matrix=mats.ctypes.data, start_step=int(start[0]), start_phase=int(start[1]),
                             rng_pos=pos.ctypes.data, resume=resume.ctypes.data)
        traj = np.ascontiguousarray(np.asarray(traj_indices, dtype=np.int64))
        assert traj.shape == (self.B,) and pos.shape == (self.B,)
        cols = n_times if sample_timesteps else 1
        if results is None:
            results = np.zeros((self.B, n_obs, cols))
            diagnostics = np.zeros((self.B, 3, cols))
        assert results.shape == (self.B, n_obs, cols) and diagnostics.shape == (self.B, 3, cols)
        assert results.flags.c_contiguous and diagnostics.flags.c_contiguous
        try:
            if status is not None:  # int32 [B], filled with one code per trajectory: a non-finite trajectory is taken out, the others finish
                assert status.dtype == np.int32 and status.shape == (self.B,) and status.flags.c_contiguous
                _lib.check(self.lib.tjm_engine_run_status(self.h, C.byref(cfg), traj.ctypes.data, results.ctypes.data, diagnostics.ctypes.data,
                                                          status.ctypes.data), "run")
            else:
                _lib.check(self.lib.tjm_engine_run(self.h, C.byref(cfg), traj.ctypes.data, results.ctypes.data, diagnostics.ctypes.data), "run")
        except _lib.CapacityError as err:
            err.resume = (int(resume[0]), int(resume[1]))
            err.rng_pos = pos
            err.results, err.diagnostics = results, diagnostics
            raise
        return results, diagnostics

    def adopt(self, src: "BatchEngine", first: int = 0) -> None:
        """Set 0 of this engine = set 0 of ``src`` slots [first, first + B), zero-padded to this engine's larger capacities."""
        _lib.check(self.lib.tjm_engine_adopt_state(self.h, src.h, int(first)), "adopt_state")

    def capacity_overflow(self, clear: bool = False) -> bool:
        """True when a truncation since the last clear was clipped by the storage capacity ``chi_max`` of this engine."""
        flag = C.c_int32(0)
        _lib.check(self.lib.tjm_engine_capacity_overflow(self.h, C.byref(flag), int(clear)), "capacity_overflow")
        return bool(flag.value)

    def bond_spectrum(self, site: int, set_index: int = 0) -> np.ndarray:
        """Singular values of A_site A_{site+1} as a (d chi_l) x (d chi_r) matrix -> [B, min(m, n)] (mps.py:604-678)."""
        n = int(self.d * min(self.caps[site], self.caps[site + 2]))
        out = np.zeros((self.B, n))
        _lib.check(self.lib.tjm_engine_bond_spectrum(self.h, set_index, int(site), out.ctypes.data, n), "bond_spectrum")
        return out

    def bitstring_probability(self, bitstring: str, set_index: int = 0) -> np.ndarray:
        """project_onto_bitstring (mps.py:1495-1537): site 0 is the first character."""
        assert len(bitstring) == self.L, "Bitstring length must match number of sites"
        bits = np.array([int(c) for c in bitstring], dtype=np.uint8)
        out = np.zeros(self.B)
        _lib.check(self.lib.tjm_engine_bitstring_probability(self.h, set_index, bits.ctypes.data, out.ctypes.data), "bitstring_probability")
        return out

    BASIS_ROTATION = {
        "Z": np.eye(2, dtype=np.complex128),
        "X": np.array([[1, 1], [1, -1]], dtype=np.complex128) / np.sqrt(2),
        "Y": np.array([[1, -1j], [1, 1j]], dtype=np.complex128) / np.sqrt(2),
    }

    def sample_shots(self, uniforms: np.ndarray, basis: str = "Z", set_index: int = 0) -> np.ndarray:
        """measure_shots (mps.py:1282-1417): uniforms[B, shots, L] -> bits[B, shots, L] (uint8), site 0 first."""
        basis = basis.upper()
        if basis not in self.BASIS_ROTATION:
            raise ValueError(f"Invalid basis: {basis}. Expected 'X', 'Y', or 'Z'.")  # mps.py:1313-1315
        u = np.ascontiguousarray(uniforms, dtype=np.float64)
        assert u.ndim == 3 and u.shape[0] == self.B and u.shape[2] == self.L
        rot = np.ascontiguousarray(self.BASIS_ROTATION[basis])
        bits = np.zeros(u.shape, dtype=np.uint8)
        _lib.check(self.lib.tjm_engine_sample_shots(self.h, set_index, u.shape[1], rot.ctypes.data, u.ctypes.data, bits.ctypes.data), "sample_shots")
        return bits

    # -- site-level steps (sweeps scheduled by the host per trajectory: dynamic TDVP) ----
    @staticmethod
    def _ids(ids):
        if ids is None:
            return None, 0
        a = _i32(ids)
        return a, len(a)

    def step_env_init(self, set_index: int = 0):
        _lib.check(self.lib.tjm_engine_step_env_init(self.h, set_index), "step_env_init")

    def step_two_site(self, site: int, dt: float, dist: str, capped: bool, ids=None, set_index: int = 0):
        a, n = self._ids(ids)
        _lib.check(self.lib.tjm_engine_step_two_site(self.h, set_index, int(site), float(dt), 0 if dist == "right" else 1, int(bool(capped)),
                                                     None if a is None else a.ctypes.data, n), "step_two_site")

    def step_one_site(self, site: int, dt: float, ids=None, set_index: int = 0):
        a, n = self._ids(ids)
        _lib.check(self.lib.tjm_engine_step_one_site(self.h, set_index, int(site), float(dt), None if a is None else a.ctypes.data, n), "step_one_site")

    def step_env(self, site: int, left: bool, ids=None, set_index: int = 0):
        a, n = self._ids(ids)
        _lib.check(self.lib.tjm_engine_step_env(self.h, set_index, int(site), int(bool(left)), None if a is None else a.ctypes.data, n), "step_env")

    def step_qr_bond(self, site: int, right: bool, dt: float, ids=None, set_index: int = 0, max_bond_dim=None):
        """``max_bond_dim``: the cut of sweep_dynamic's one-site branch, a new bond above it is sliced back to it (integrators.py:361-364)."""
        a, n = self._ids(ids)
        _lib.check(self.lib.tjm_engine_step_qr_bond(self.h, set_index, int(site), int(bool(right)), float(dt), -1 if max_bond_dim is None else int(max_bond_dim),
                                                    None if a is None else a.ctypes.data, n), "step_qr_bond")

    def step_cap_bond(self, bond: int, target: int, ids=None, set_index: int = 0):
        a, n = self._ids(ids)
        _lib.check(self.lib.tjm_engine_step_cap_bond(self.h, set_index, int(bond), int(target), None if a is None else a.ctypes.data, n), "step_cap_bond")

    def sweep_dynamic(self, max_bond_dim, dt: float, set_index: int = 0):
        """One whole sweep of the dynamic TDVP in one C call (tjm_engine_sweep_dynamic): the branch lists of every site are formed
        inside the library."""
        _lib.check(self.lib.tjm_engine_sweep_dynamic(self.h, set_index, -1 if max_bond_dim is None else int(max_bond_dim), float(dt)), "sweep_dynamic")

    def bug_sweep(self, dt: float, set_index: int = 0):
        """One half-sweep of the BUG integrator in one C call (tjm_engine_bug_sweep)."""
        _lib.check(self.lib.tjm_engine_bug_sweep(self.h, set_index, float(dt)), "bug_sweep")

    # -- steps of the BUG integrator (engines created with cap_slack >= 2) -----------------
    def step_bug_prepare(self, set_index: int = 0):
        _lib.check(self.lib.tjm_engine_step_bug_prepare(self.h, set_index), "step_bug_prepare")

    def step_bug_site(self, site: int, dt: float, set_index: int = 0):
        _lib.check(self.lib.tjm_engine_step_bug_site(self.h, set_index, int(site), float(dt)), "step_bug_site")

    def step_bug_root(self, dt: float, set_index: int = 0):
        _lib.check(self.lib.tjm_engine_step_bug_root(self.h, set_index, float(dt)), "step_bug_root")

    def step_flip(self, set_index: int = 0):
        _lib.check(self.lib.tjm_engine_step_flip(self.h, set_index), "step_flip")

    def apply_gate_mpo(self, first: int, last: int, left_ops: np.ndarray, right_ops: np.ndarray, set_index: int = 0):
        """U = sum_k left_ops[k] (x) right_ops[k] on the distant pair (first, last) as an MPO product (tjm_engine_apply_gate_mpo); the
        bonds in between grow by the number of terms until ``step_compress``."""
        lo = np.ascontiguousarray(left_ops, dtype=np.complex128)
        ro = np.ascontiguousarray(right_ops, dtype=np.complex128)
        assert lo.shape == ro.shape == (lo.shape[0], self.d, self.d)
        _lib.check(self.lib.tjm_engine_apply_gate_mpo(self.h, set_index, int(first), int(last), int(lo.shape[0]), lo.ctypes.data, ro.ctypes.data), "apply_gate_mpo")

    def step_compress(self, threshold: float, max_bond_dim, trunc_mode: str = "discarded_weight", set_index: int = 0):
        _lib.check(self.lib.tjm_engine_step_compress(self.h, set_index, float(threshold), -1 if max_bond_dim is None else int(max_bond_dim),
                                                     TRUNC_MODES[trunc_mode]), "step_compress")

    def stats(self) -> dict:
        s = np.zeros(14, dtype=np.int64)
        self.lib.tjm_engine_stats_ex(self.h, s.ctypes.data, 14)
        return dict(matvecs=int(s[0]), krylov_calls=int(s[1]), svds=int(s[2]), svd_sweeps=int(s[3]), site_updates=int(s[4]),
                    matvecs_two_site=int(s[5]), env_updates=int(s[6]), direct_applies=int(s[7]), svd_matrices=int(s[8]), identity_checks=int(s[9]),
                    identity_channels=int(s[10]), certified_dissipations=int(s[11]), certified_jumps=int(s[12]), certificate_tests_blocked_cholesky=int(s[13]))

    def profile(self, enable: bool = True):
        """Bracket the kernel classes of every step with HIP events on the engine's stream (tjm_engine_profile)."""
        _lib.check(self.lib.tjm_engine_profile(self.h, int(bool(enable))), "profile")

    def profile_read(self) -> dict:
        ms = np.zeros(3)
        n = np.zeros(3, dtype=np.int64)
        _lib.check(self.lib.tjm_engine_profile_read(self.h, ms.ctypes.data, n.ctypes.data), "profile_read")
        return {k: {"ms": float(ms[i]), "regions": int(n[i])} for i, k in enumerate(("svd", "krylov", "env"))}

    def synchronize(self):
        self.torch.cuda.synchronize(self.device)
