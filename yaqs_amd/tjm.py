"""Batched TJM trajectory drivers on the HIP engine and the ``Simulator`` front end.

Restates the control flow of the reference drivers for a whole batch of trajectories
advancing in lock-step (paths relative to /root/reference/src/mqt/yaqs):

* ``analog_tjm_1`` / ``analog_tjm_2``        analog/analog_tjm.py:206-462
* RNG streams                                 core/random_utils.py:15-69
* ``Simulator.run`` (MPS analog branch)       simulator.py:1173-1312, 1444-1679
* trajectory sharding replaces                core/parallel_utils.py:331-390
"""
from __future__ import annotations

import os
import threading
from typing import Sequence

import numpy as np

from .api import META_OBSERVABLES, AnalogSimParams, MPO, MPS, NoiseModel, Result, is_pauli, validate_noise_model_for_run
from ._lib import CapacityError
from .engine import BatchEngine

TAG_TRAJ = 0x5452414A
TAG_SAMPLE = 0x53414D50
TAG_SHOT = 0x53484F54
TAG_DISORDER = 0x4449534F


def trajectory_uniforms(seed: int | None, traj: int, n: int) -> np.ndarray:
    """First ``n`` doubles of ``make_trajectory_rng(traj, base_seed=seed)`` (random_utils.py:20-37)."""
    rng = np.random.default_rng() if seed is None else np.random.default_rng(np.random.SeedSequence([seed, traj, TAG_TRAJ]))
    return rng.random(n)


def sample_uniforms(seed: int | None, traj: int, timestep: int, n: int = 2) -> np.ndarray:
    """First doubles of ``make_sample_rng`` (random_utils.py:40-69)."""
    rng = np.random.default_rng() if seed is None else np.random.default_rng(np.random.SeedSequence([seed, traj, timestep, TAG_SAMPLE]))
    return rng.random(n)


def disorder_rng(seed: int | None) -> np.random.Generator:
    """``make_disorder_rng`` (random_utils.py:72-87)."""
    return np.random.default_rng() if seed is None else np.random.default_rng(np.random.SeedSequence([seed, TAG_DISORDER]))


def _diagnostics_from_bonds(chi: np.ndarray, d: int) -> np.ndarray:
    """record_diagnostics (mps.py:549-602) from the bond table chi[B, L+1] -> [B, 3]."""
    inner = chi[:, 1:-1].astype(np.float64)
    cost = np.sum(inner ** 3, axis=1)
    max_bond = np.maximum(d, np.max(chi[:, 1:], axis=1)).astype(np.float64)
    total = np.sum(inner, axis=1)
    return np.stack([cost, max_bond, total], axis=1)


def dynamic_tdvp(e, set_index: int, max_bond_dim: int | None, dt: float, sweeps: int = 1) -> None:
    """``tdvp(tdvp_mode="dynamic")`` (tdvp/tdvp.py:69-111 -> integrators.py:294-511) for a lock-step batch.

    Site by site a trajectory takes the two-site branch (uncapped split, ``split_tdvp(dynamic=True)``) while the bond to its right
    (left, on the way back) is below ``max_bond_dim`` and the one-site branch with a QR bond transfer once it has reached it.  Bonds
    differ per trajectory, so the host reads the bond table before every site, forms the two index lists and calls the engine's
    site-level steps for each (``tjm_engine_step_*``); every contraction, exponential and factorisation is a HIP kernel."""
    for _ in range(sweeps):
        if HOST_SEQUENCED_SWEEPS:
            _sweep_dynamic(e, set_index, max_bond_dim, dt / sweeps)
        else:  # the same sweep sequenced inside the library (tjm_engine_sweep_dynamic): one C call, one bond column per site to the host
            e.sweep_dynamic(max_bond_dim, dt / sweeps, set_index)


# YAQS_AMD_HOST_SWEEPS=1: the site loops of the dynamic TDVP and of the BUG integrator sequenced from Python (the readable mirror of
# Engine::sweep_dynamic / bug_sweep, kept for the A/B and as documentation); default: one C entry per sweep
HOST_SEQUENCED_SWEEPS = os.environ.get("YAQS_AMD_HOST_SWEEPS") == "1"


def _sweep_dynamic(e, s: int, cap: int | None, dt: float) -> None:
    n = e.L
    everyone = np.arange(e.B)

    def lists(dims):
        """(one-site list, two-site list) for the bond dimensions ``dims[B]`` of the bond that decides the branch."""
        if cap is None:
            return everyone[:0], everyone
        at_cap = dims >= cap
        return everyone[at_cap], everyone[~at_cap]

    if cap is not None:  # _cap_bonds (sweep_utils.py:280-302): bonds the previous sweep left above the cap
        for bond in range(n - 1):
            over = everyone[e.bond_dims(s)[:, bond + 1] > cap]
            if len(over):
                e.step_cap_bond(bond, cap, over, s)
    e.step_env_init(s)
    for i in range(n):                     # left to right (integrators.py:340-424)
        one, two = lists(e.bond_dims(s)[:, i + 1])
        if len(one):
            e.step_one_site(i, 0.5 * dt, one, s)
            if i != n - 1:
                e.step_qr_bond(i, True, -0.5 * dt, one, s, max_bond_dim=cap)
        if len(two) and i != n - 1:
            e.step_two_site(i, 0.5 * dt, "right", False, two, s)
            if i == n - 2:
                e.step_env(i + 1, False, two, s)
                e.step_env(i, True, two, s)
            else:
                e.step_env(i, True, two, s)
                e.step_one_site(i + 1, -0.5 * dt, two, s)
    for i in reversed(range(n)):           # right to left (integrators.py:427-505)
        one, two = lists(e.bond_dims(s)[:, i])
        if len(one):
            e.step_one_site(i, 0.5 * dt, one, s)
            if i != 0:
                e.step_qr_bond(i, False, -0.5 * dt, one, s, max_bond_dim=cap)
        if len(two) and i != 0:
            e.step_two_site(i - 1, 0.5 * dt, "left", False, two, s)
            e.step_env(i, False, two, s)
            if i != 1:
                e.step_one_site(i - 1, -0.5 * dt, two, s)


def bug_step(e, set_index: int, params, mpo_tensors) -> None:
    """One physical step of the Basis-Update and Galerkin integrator, ``bug()`` (core/methods/bug.py:213-257), for a lock-step batch:
    two half-sweeps of ``dt / 2`` with alternating endpoints (the second on the site-reversed chain with the reflected MPO), one
    compression with the run's truncation settings, renormalisation.  Each half-sweep (``bug_sweep``, bug.py:128-196) prepares the
    coefficient-bearing centres and left environments, then walks from the last site to site 1 - Krylov predictor, left QR of the
    stacked basis ``[retained | predictor]``, basis-change matrix, transported centre, right block - and evolves the root.  The
    host only sequences the steps; every one of them is a set of HIP kernels behind ``tjm_engine_step_bug_* / _flip / _compress``."""
    n = e.L
    half = params.dt / 2.0
    reflected = [np.ascontiguousarray(np.transpose(np.asarray(w), (0, 1, 3, 2))) for w in reversed(list(mpo_tensors))]  # MPO.reflected, mpo.py:1612-1630

    def sweep():
        if not HOST_SEQUENCED_SWEEPS:
            e.bug_sweep(half, set_index)
            return
        e.step_bug_prepare(set_index)
        for site in range(n - 1, 0, -1):
            e.step_bug_site(site, half, set_index)
        e.step_bug_root(half, set_index)

    sweep()
    if n > 1:
        e.step_flip(set_index)
        e.set_mpo(reflected)
        e.canonicalize_qr(n - 1, set_index)
    sweep()
    if n > 1:
        e.step_flip(set_index)
        e.set_mpo(list(mpo_tensors))
        e.canonicalize_qr(n - 1, set_index)
    e.step_compress(params.svd_threshold, params.max_bond_dim, params.trunc_mode, set_index)
    e.normalize_qr(0, set_index)


class TrajectoryBatch:
    """Runs trajectories ``traj_indices`` (one per engine slot) through one TJM driver."""

    def __init__(self, engine: BatchEngine, params: AnalogSimParams, noise: NoiseModel | None):
        self.e = engine
        self.p = params
        self.noise = noise if (noise is not None and (noise.processes or getattr(noise, "scheduled_jumps", None))) else None
        if params.tdvp_mode not in ("1site", "2site", "dynamic"):
            raise ValueError(f'tdvp_mode must be one of ("1site", "2site", "dynamic"), got {params.tdvp_mode!r}.')  # tdvp.py:109-111
        self.dynamic = params.tdvp_mode == "dynamic" and engine.L > 1  # a one-site chain falls back to 1TDVP (tdvp.py:96-98)
        self.bug = str(getattr(getattr(params, "evolution_mode", "tdvp"), "value", getattr(params, "evolution_mode", "tdvp"))) == "bug"
        self.two_site_obs = False
        self.schmidt: dict = {}  # (sorted row, column) -> [B, 500] Schmidt spectra
        self.meta_obs = any(obs.gate.name in META_OBSERVABLES for obs in params.observables)
        for obs in params.observables:
            if obs.gate.name in META_OBSERVABLES:
                continue
            if isinstance(obs.sites, (list, tuple)) and len(obs.sites) == 2:
                if obs.sites[1] != obs.sites[0] + 1:
                    raise ValueError("Only nearest-neighbor observables are currently implemented.")  # mps.py:1012-1014
                self.two_site_obs = True
        engine.set_params(dt=params.dt, svd_threshold=params.svd_threshold, trunc_mode=params.trunc_mode,
                          max_bond_dim=params.max_bond_dim, krylov_tol=params.krylov_tol,
                          tdvp_mode=params.tdvp_mode if (self.dynamic or params.tdvp_mode != "dynamic") else "1site", tdvp_sweeps=params.tdvp_sweeps)
        procs = self.noise.processes if self.noise is not None else []
        engine.set_noise(procs, [is_pauli(q) for q in procs])
        self.sorted_obs = params.sorted_observables
        self.dp_log: list[np.ndarray] = []
        self.jump_log: list[np.ndarray] = []
        self.intervals = None  # piecewise-constant Hamiltonian: one MPO tensor list per time interval
        self._interval_loaded = None

    def set_intervals(self, mpos) -> None:
        """``hamiltonian`` given as a tuple of MPOs, one per interval of the time grid (analog_tjm.py:43-49)."""
        mpos = [m.tensors if hasattr(m, "tensors") else m for m in mpos]
        if len(mpos) != len(self.p.times) - 1:
            raise ValueError("a piecewise Hamiltonian needs one MPO per time interval")
        self.intervals = mpos
        self._interval_loaded = None

    def _tdvp(self, set_index: int, interval: int) -> None:
        if self.intervals is not None and self._interval_loaded != interval:
            self.e.set_mpo(self.intervals[interval])
            self._interval_loaded = interval
        if self.bug:  # apply_unitary_evolution (analog/evolution.py:24-51)
            mpo = [np.array(w) for w in self.e.mpo_tensors]  # the interval's Hamiltonian was loaded just above
            bug_step(self.e, set_index, self.p, mpo)
        elif self.dynamic:
            dynamic_tdvp(self.e, set_index, self.p.max_bond_dim, self.p.dt, self.p.tdvp_sweeps)
        else:
            self.e.tdvp(set_index)

    # ---- measurement ----------------------------------------------------------------
    def _measure(self, set_index: int, results: np.ndarray, diagnostics: np.ndarray, col: int) -> None:
        e = self.e
        if self.two_site_obs:
            M, M2 = e.site_moments2(set_index)
        else:
            M, M2 = e.site_moments(set_index), None  # [L, B, d, d]
        for row, obs in enumerate(self.sorted_obs):
            site = obs.first_site
            if obs.gate.name in ("entropy", "schmidt_spectrum"):  # mps.py:1200-1213
                assert isinstance(obs.sites, (list, tuple)) and len(obs.sites) == 2, "Given metric requires 2 sites to act on."
                lo, hi = min(obs.sites), max(obs.sites)
                assert hi - lo == 1, "Entropy and Schmidt cuts must be nearest neighbor."
                spec = e.bond_spectrum(lo, set_index)
                chi = e.bond_dims(set_index)
                if obs.gate.name == "entropy":
                    s2 = spec ** 2
                    norm = s2.sum(axis=1, keepdims=True)
                    pr = np.divide(s2, norm, out=np.zeros_like(s2), where=norm > 0)
                    ent = -np.sum(pr * np.log(pr + np.finfo(np.float64).tiny), axis=1)
                    ent[chi[:, lo + 1] == 1] = 0.0
                    results[:, row, col] = ent
                else:
                    padded = np.full((e.B, 500), np.nan)
                    for b in range(e.B):
                        nb_ = int(e.d * min(chi[b, lo], chi[b, lo + 2]))
                        if chi[b, lo + 1] == 1:
                            padded[b, 0] = 1.0
                        else:
                            padded[b, : min(500, nb_)] = spec[b, : min(500, nb_)]
                    self.schmidt[(row, col)] = padded
                    results[:, row, col] = np.nan  # the vector lives in self.schmidt[(row, col)]
                continue
            if obs.gate.name == "pvm":
                results[:, row, col] = e.bitstring_probability(obs.gate.bitstring, set_index)
                continue
            O = np.asarray(obs.gate.matrix, dtype=np.complex128)
            if isinstance(obs.sites, (list, tuple)) and len(obs.sites) == 2:
                val = np.einsum("pq,bpq->b", O, M2[site])   # <theta| O |theta> on the merged pair (mps.py:999-1047)
            else:
                val = np.einsum("pq,bpq->b", O, M[site])
            # "assert exp.imag < 1e-13" (mps.py:1233; an fp64 rounding bound: 1e-5 on the complex64 engine): a non-finite value fails it too
            if not np.all(val.imag < (1e-13 if getattr(e, "dtype", "complex128") == "complex128" else 1e-5)):
                raise AssertionError(f"Measurement should be real, got max imag {np.nanmax(val.imag) if not np.all(np.isnan(val.imag)) else np.nan:.3e}")
            results[:, row, col] = val.real
        diagnostics[:, :, col] = _diagnostics_from_bonds(e.bond_dims(set_index), e.d)

    def _stochastic(self, set_index: int, dt: float, u: np.ndarray, pos: np.ndarray | None) -> None:
        """One stochastic_process call; ``u[b, pos[b]:pos[b]+2]`` are the candidate draws."""
        e = self.e
        if self.noise is None:
            e.set_uniforms(np.zeros((e.B, 2)))
            e.stochastic(dt, set_index)
            return
        if pos is None:
            cand = u[:, :2]
        else:
            cand = np.stack([u[np.arange(e.B), pos], u[np.arange(e.B), pos + 1]], axis=1)
        e.set_uniforms(cand)
        jumped, dp = e.stochastic(dt, set_index)
        self.dp_log.append(dp.copy())
        self.jump_log.append(jumped.copy())
        if pos is not None:
            pos += 1 + jumped

    # ---- drivers --------------------------------------------------------------------
    def run(self, traj_indices: Sequence[int], initial: MPS | None, native: bool = False, resume: dict | None = None, *, sample_at=None,
            continue_trajectory: bool = False, sample_timestep_offset: int = 0, use_trajectory_rng_for_final_sample: bool = False,
            rng_pos=None):
        """``native=True`` hands the whole schedule to the C driver (``tjm_engine_run``): same results, no per-step host
        logic and no dp / jump logs.  The Python schedule below is the readable mirror of analog_tjm.py used by the tests.

        The keyword options are the reference's continuation options of the drivers (analog_tjm.py:206-255, 369-400), served by the
        Python schedule: ``sample_at`` = time indices to measure; ``continue_trajectory`` (order 2) = set 0 already holds the
        handed-off trajectory states phi (``initial`` may be None), no ``initialize``; ``sample_timestep_offset`` shifts the sample
        streams onto a global timeline; ``rng_pos`` = cursors into the trajectory streams (the reference's external ``rng``: the
        stream of trajectory t continues where the previous segment left it; the cursors after the run are in ``self.rng_pos``);
        ``use_trajectory_rng_for_final_sample`` lets the last measurement copy draw from that stream.  The trajectory states stay in
        set 0 of the engine after the run (``return_trajectory_state``)."""
        e, p = self.e, self.p
        assert len(traj_indices) == e.B
        n_t = len(p.times)
        self._validate_sample_at(sample_at)
        self.measure_at = frozenset(int(q) for q in sample_at) if sample_at is not None else None
        self.sample_offset = int(sample_timestep_offset)
        self.final_from_traj = bool(use_trajectory_rng_for_final_sample and rng_pos is not None)
        options = sample_at is not None or continue_trajectory or sample_timestep_offset or rng_pos is not None
        if continue_trajectory and p.order != 2:
            raise ValueError("continue_trajectory belongs to the order-2 driver (analog_tjm.py:206-255)")
        has_sched = self.noise is not None and bool(getattr(self.noise, "scheduled_jumps", None))
        if has_sched and p.order != 1:
            raise ValueError(f"scheduled_jumps are only supported for AnalogSimParams(order=1); got order={p.order}.")  # noise_model.py:758-765
        if has_sched:
            for j in self.noise.scheduled_jumps:
                if not np.any(np.isclose(p.times, j["time"], atol=p.dt * 1e-3, rtol=0.0)):
                    raise ValueError(f"Scheduled jump time {j['time']} is not on the simulation time grid.")  # noise_model.py:768-775
        if native and not options and not self.dynamic and not self.bug and not self.meta_obs and not has_sched and self.intervals is None:  # entropy / Schmidt spectrum / PVM are evaluated by the host schedule
            obs = [(o_.first_site, np.asarray(o_.gate.matrix, dtype=np.complex128)) for o_ in self.sorted_obs]
            kw = {}
            if resume is None:
                e.load_state(initial.tensors, 0)
            else:  # the states were adopted from the engine that ran out of capacity (CapacityError.resume / .rng_pos)
                kw = dict(start=resume["start"], rng_pos=resume["rng_pos"], results=resume["results"], diagnostics=resume["diagnostics"])
            return e.run(order=p.order, n_times=n_t, sample_timesteps=p.sample_timesteps, has_noise=self.noise is not None,
                         seed=p.random_seed, traj_indices=traj_indices, observables=obs, **kw)
        cols = n_t if p.sample_timesteps else 1
        first_pos = np.zeros(e.B, dtype=np.int64) if rng_pos is None else np.asarray(rng_pos, dtype=np.int64).copy()
        n_draw = 2 * n_t + 4 + int(first_pos.max(initial=0))
        u = np.stack([trajectory_uniforms(p.random_seed, int(t), n_draw) for t in traj_indices])
        e.capacity_overflow(clear=True)
        if resume is None:
            results = np.zeros((e.B, len(self.sorted_obs), cols))
            diagnostics = np.zeros((e.B, 3, cols))
            if not continue_trajectory:
                e.load_state(initial.tensors, 0)
            pos = first_pos
            start = (0, 0)
        else:  # continue on this (larger) engine at the time step that ran out of capacity on the previous one
            results, diagnostics = resume["results"], resume["diagnostics"]
            pos = np.asarray(resume["rng_pos"], dtype=np.int64).copy()
            start = tuple(resume["start"])
        self._out = (results, diagnostics)
        if p.order == 2:
            self._run_order2(traj_indices, results, diagnostics, u, pos, start, continue_trajectory)
        else:
            self._run_order1(results, diagnostics, u, pos, start)
        self.rng_pos = pos
        return results, diagnostics

    def _validate_sample_at(self, sample_at) -> None:
        """analog_tjm.py:69-84."""
        if sample_at is None:
            return
        n_times = len(self.p.times)
        for index in sample_at:
            if isinstance(index, bool) or not isinstance(index, (int, np.integer)) or index < 0 or index >= n_times:
                raise ValueError(f"sample_at index {index!r} is outside the time grid [0, {n_times}).")
        if not self.p.sample_timesteps and len(frozenset(sample_at)) > 1:
            raise ValueError("Selecting multiple sample_at indices requires sample_timesteps=True.")

    def _record(self, j: int) -> bool:
        """_measure_at (analog_tjm.py:52-66)."""
        if self.measure_at is not None:
            return j in self.measure_at
        return True if self.p.sample_timesteps else j == len(self.p.times) - 1

    # ---- storage capacity: per-step rollback of the host schedule (mirror of run_batch in tjm_run.hip) ----
    def _snapshot(self, pos):
        """Start of a time step: the trajectory states go to the measurement-copy set (idle then), the cursors and log lengths aside."""
        self.e.copy_state(1, 0)
        return pos.copy(), len(self.dp_log), len(self.jump_log), set(self.schmidt)

    def _clipped(self, step: int, phase: int, pos, snap=None) -> None:
        """After a step (phase 0) or its sampling (phase 1): on a clipped truncation roll back and hand over to a larger engine."""
        if not self.e.capacity_overflow():
            return
        if snap is not None:
            self.e.copy_state(0, 1)
            pos[:] = snap[0]
            del self.dp_log[snap[1]:]
            del self.jump_log[snap[2]:]
            for key in set(self.schmidt) - snap[3]:
                del self.schmidt[key]
        err = CapacityError(f"time step {step} needs a bond beyond the engine's capacity chi = {self.e.chi_max}")
        err.resume, err.rng_pos = (step, phase), pos.copy()
        err.results, err.diagnostics = self._out
        raise err

    # ---- scheduled jumps (core/methods/scheduled_jumps.py:28-119) ----------------------
    def _scheduled_at(self, time: float) -> list:
        if self.noise is None or not getattr(self.noise, "scheduled_jumps", None):
            return []
        return [j for j in self.noise.scheduled_jumps if np.isclose(j["time"], time, atol=self.p.dt * 1e-3, rtol=0.0)]

    def _apply_scheduled(self, jumps) -> None:
        e, p = self.e, self.p
        for j in jumps:
            if len(j["sites"]) == 1:
                e.apply_single(j["sites"][0], j["matrix"])
            else:
                e.apply_pair(j["sites"][0], j["matrix"], min_keep=1)
        e.canonicalize_qr(e.L - 1)                     # state.norm() needs no gauge; here it is ||A_0||^2 after the QR sweep
        nsq = e.site0_normsq()
        if not np.all(np.isfinite(nsq)) or np.any(nsq <= 0.0):
            raise ValueError("Scheduled jump produced a zero or non-finite squared norm. The jump operator annihilates the current state.")
        e.normalize_qr(0)                              # normalize("B")

    def _run_order1(self, results, diagnostics, u, pos, start=(0, 0)):
        """analog_tjm_1 (analog_tjm.py:369-462)."""
        e, p = self.e, self.p
        n_t = len(p.times)
        if start[0] == 0:
            first = self._scheduled_at(p.times[0])
            if first:
                self._apply_scheduled(first)
                self._clipped(0, 0, pos)
            self._measure_initial = (0 in self.measure_at) if self.measure_at is not None else p.sample_timesteps
            if self._measure_initial:
                self._measure(0, results, diagnostics, 0)
        for j in range(max(1, start[0]), n_t):
            snap = self._snapshot(pos)
            self._tdvp(0, j - 1)
            if self.noise is not None:
                e.dissipate(p.dt, 0)
                due = self._scheduled_at(p.times[j])
                if due:
                    self._apply_scheduled(due)
                else:
                    self._stochastic(0, p.dt, u, pos)
            self._clipped(j, 0, pos, snap)
            if self._record(j):
                self._measure(0, results, diagnostics, j if p.sample_timesteps else 0)
        if not getattr(self, "_measure_initial", False) and not p.sample_timesteps and n_t <= 1:
            self._measure(0, results, diagnostics, 0)

    def _run_order2(self, traj_indices, results, diagnostics, u, pos, start=(0, 0), continue_trajectory=False):
        """analog_tjm_2 (analog_tjm.py:206-366) with its continuation options."""
        e, p = self.e, self.p
        n_t = len(p.times)
        record = self._record

        def sample(j, interval):
            if not record(j):
                return
            e.copy_state(1, 0)  # psi = deepcopy(phi)
            self._tdvp(1, interval)  # capture_sample(phi, j, operator), analog_tjm.py:296-313
            e.dissipate(p.dt / 2, 1)
            n_dp, n_jump, keys = len(self.dp_log), len(self.jump_log), set(self.schmidt)
            before = pos.copy()
            if self.final_from_traj and j == n_t - 1:
                self._stochastic(1, p.dt, u, pos)  # the last measurement copy draws from the trajectory stream (programs' final segment)
            else:
                us = np.stack([sample_uniforms(p.random_seed, int(t), j + self.sample_offset) for t in traj_indices])
                self._stochastic(1, p.dt, us, None)
            if e.capacity_overflow():  # phi is untouched: only the sampling of step j is repeated on the larger engine
                del self.dp_log[n_dp:]
                del self.jump_log[n_jump:]
                pos[:] = before
                for key in set(self.schmidt) - keys:
                    del self.schmidt[key]
                first = 1 if continue_trajectory else 2
                self._clipped(j if j >= first else 0, 1 if j >= first else 0, pos)
            self._measure(1, results, diagnostics, j if p.sample_timesteps else 0)

        if n_t == 1:
            if record(0):
                self._measure(0, results, diagnostics, 0)
            return
        if continue_trajectory:
            # mid-Trotter hand-off: the junction is measured again on the global sample timeline without touching phi, then every
            # time step is a step_through with the static operator (analog_tjm.py:323-334)
            if start[0] == 0:
                sample(0, 0)
            for j in range(max(1, start[0]), n_t):
                if not (j == start[0] and start[1] == 1):
                    snap = self._snapshot(pos)
                    self._tdvp(0, 0)
                    e.dissipate(p.dt, 0)
                    self._stochastic(0, p.dt, u, pos)
                    self._clipped(j, 0, pos, snap)
                sample(j, 0)
            return
        if start[0] == 0:
            if record(0):
                self._measure(0, results, diagnostics, 0)
            e.dissipate(p.dt / 2, 0)
            self._stochastic(0, p.dt, u, pos)
            self._clipped(0, 0, pos)  # before the first full step: nothing to keep
            sample(1, 0)
        for j in range(max(2, start[0]), n_t):
            if not (j == start[0] and start[1] == 1):
                snap = self._snapshot(pos)
                self._tdvp(0, j - 2)  # step_through with interval j - 2, analog_tjm.py:351-358
                e.dissipate(p.dt, 0)
                self._stochastic(0, p.dt, u, pos)
                self._clipped(j, 0, pos, snap)
            sample(j, j - 1)


class DigitalBatch:
    """``digital_tjm`` (digital/digital_tjm.py:636-749) for a batch of trajectories: pre-compiled gate layers of
    single-qubit gates and nearest-neighbour two-qubit gates (TEBD), local noise after every two-qubit gate with dt = 1."""

    def __init__(self, engine: BatchEngine, params, noise: NoiseModel | None):
        self.e = engine
        self.p = params
        self.noise = noise if (noise is not None and noise.processes) else None
        self.noisy = self.noise is not None and any(q["strength"] != 0 for q in self.noise.processes)
        engine.set_params(dt=1.0, svd_threshold=params.svd_threshold, trunc_mode=params.trunc_mode, max_bond_dim=params.max_bond_dim,
                          krylov_tol=1e-4)
        procs = self.noise.processes if self.noise is not None else []
        engine.set_noise(procs, [is_pauli(q) for q in procs])
        self.procs = procs
        self.sorted_obs = params.sorted_observables
        self.two_site_obs = any(isinstance(o.sites, (list, tuple)) and len(o.sites) == 2 and o.gate.name not in META_OBSERVABLES
                                for o in params.observables)
        self.schmidt: dict = {}
        self.jump_log: list[np.ndarray] = []

    _measure = TrajectoryBatch._measure

    _SWAP = np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.complex128).reshape(2, 2, 2, 2)

    def _apply_two_qubit(self, entry):
        """``apply_two_qubit_gate_tebd`` (digital_tjm.py:455-533).  ``entry`` is ``(left, U[out_l,out_r,in_l,in_r])`` for a gate on
        (left, left+1), or ``(site0, site1, 4x4 matrix)`` with the matrix indexed ``2*q_site0 + q_site1`` as ``BaseGate.matrix``;
        non-adjacent sites are routed with adjacent SWAPs.  Returns the gate's sites and the centre position afterwards."""
        e = self.e
        if len(entry) == 2:
            left, u4 = entry
            e.tebd_gate(left, u4)
            return {left, left + 1}, left + 1
        s0, s1, mat = entry
        if s0 == s1:
            raise ValueError("a two-qubit gate needs two different sites")
        t = np.asarray(mat, dtype=np.complex128).reshape(2, 2, 2, 2)
        u_lr = t if s0 < s1 else t.transpose(1, 0, 3, 2)  # resolve_lr_tensor (mpo_utils.py:127-159)
        left, right = min(s0, s1), max(s0, s1)
        mode = getattr(self.p, "gate_mode", "mpo")
        if right - left > 1 and mode == "mpo":
            # the reference's default (digital_tjm.py:536-557, 616-620): MPO.from_gate(gate, L).multiply(state) and MPS.compress.
            # Operator Schmidt decomposition of the gate as split_tensor does it (gate_library.py:29-63: singular values <= 1e-6 dropped)
            sv_u, sv, sv_vh = np.linalg.svd(u_lr.transpose(0, 2, 1, 3).reshape(4, 4), full_matrices=False)
            keep = max(1, int(np.sum(sv > 1e-6)))
            e.apply_gate_mpo(left, right, sv_u[:, :keep].T.reshape(keep, 2, 2), (sv[:keep, None] * sv_vh[:keep]).reshape(keep, 2, 2))
            e.step_compress(self.p.svd_threshold, self.p.max_bond_dim, self.p.trunc_mode)
            return {s0, s1}, 0  # step_compress leaves the centre on site 0 (the reference moves it to L // 2: a gauge choice)
        if right - left > 1 and mode != "swaps":
            # "tdvp" / "full-tdvp" evolve a window with the gate's generator (digital_tjm.py:408-453): not built - the reference's own
            # numbers are rounding-defined at the 1e-4 level there (a zero singular value kept by min_keep = 2 seeds the next projector;
            # tests/test_oracle_golden.py::test_digital_gates_by_tdvp_on_a_window_are_rounding_defined), nothing to reproduce to 1e-8
            raise NotImplementedError(f"long-range gate on sites ({s0}, {s1}) needs gate_mode='mpo' or 'swaps' (got {self.p.gate_mode!r})")
        center = 0
        for i in range(right - 1, left, -1):  # bring the right qubit next to the left one
            e.tebd_gate(i, self._SWAP, center=center)
            center = i + 1
        e.tebd_gate(left, u_lr, center=center)
        center = left + 1
        for i in range(left + 1, right):      # and back
            e.tebd_gate(i, self._SWAP, center=center)
            center = i + 1
        return {s0, s1}, center

    def run(self, traj_indices: Sequence[int], initial: MPS | None, layers, shots_per_traj=None, basis: str = "Z", resume: dict | None = None):
        """``resume`` continues a run that stopped with ``CapacityError`` at the start of layer ``resume["start"][0] - 1`` after the
        states were adopted from the smaller engine (same contract as ``TrajectoryBatch.run``)."""
        e, p = self.e, self.p
        assert len(traj_indices) == e.B
        n_gates = sum(len(l.even) + len(l.odd) for l in layers)
        mid = p.num_mid_measurements if p.sample_layers else 0
        cols = (mid + 2) if p.sample_layers else 1
        e.capacity_overflow(clear=True)
        if resume is None:
            results = np.zeros((e.B, len(self.sorted_obs), cols))
            diagnostics = np.zeros((e.B, 3, cols))
            e.load_state(initial.tensors, 0)
            if p.sample_layers:
                self._measure(0, results, diagnostics, 0)
            pos = np.zeros(e.B, dtype=np.int64)
            col, first_layer = 0, 0
        else:
            results, diagnostics = resume["results"], resume["diagnostics"]
            pos = np.asarray(resume["rng_pos"], dtype=np.int64).copy()
            col, first_layer = int(resume["extra"]["col"]), int(resume["start"][0]) - 1
        u = np.stack([trajectory_uniforms(p.random_seed, int(t), 2 * n_gates + 2) for t in traj_indices])
        rows = np.arange(e.B)
        for li in range(first_layer, len(layers)):
            layer = layers[li]
            e.copy_state(1, 0)  # the layer is rolled back if one of its truncations is clipped by the storage
            pos_at_start, jumps_at_start = pos.copy(), len(self.jump_log)
            for site, m in layer.singles:
                e.apply_single(site, m)
            for group in (layer.even, layer.odd):
                for entry in group:
                    sites, center = self._apply_two_qubit(entry)
                    if not self.noisy:
                        e.normalize_qr(center)
                        continue
                    local = [k for k, q in enumerate(self.procs) if set(q["sites"]).issubset(sites)]  # digital_tjm.py:187-204
                    e.set_noise_filter(local)
                    e.dissipate_from(1.0, center)
                    e.set_uniforms(np.stack([u[rows, pos], u[rows, pos + 1]], axis=1))
                    jumped, _ = e.stochastic(1.0)
                    if not local:
                        jumped[:] = 0  # an empty local model can only renormalise (stochastic_process.py:236-243)
                    self.jump_log.append(jumped.copy())
                    pos += 1 + jumped
            if e.capacity_overflow():
                e.copy_state(0, 1)
                e.set_noise_filter(None)
                del self.jump_log[jumps_at_start:]
                err = CapacityError(f"layer {li} needs a bond beyond the engine's capacity chi = {e.chi_max}")
                err.resume, err.rng_pos, err.results, err.diagnostics, err.extra = (li + 1, 0), pos_at_start, results, diagnostics, {"col": col}
                raise err
            if p.sample_layers:
                for _ in range(layer.sample_points):
                    col += 1
                    self._measure(0, results, diagnostics, col)
        e.set_noise_filter(None)
        self._measure(0, results, diagnostics, cols - 1)
        self.counts = None
        if shots_per_traj is not None:
            self.counts = self._sample(traj_indices, np.asarray(shots_per_traj, dtype=np.int64), basis)
        return results, diagnostics

    def _sample(self, traj_indices, shots_per_traj, basis):
        """measure_shots on the final states (mps.py:1352-1417); trajectory b keeps its first shots_per_traj[b] samples.
        The reference draws from an unseeded generator; with a seed the draws come from SeedSequence([seed, traj, TAG_SHOT])."""
        e, p = self.e, self.p
        n = int(shots_per_traj.max()) if len(shots_per_traj) else 0
        counts: dict[int, int] = {}
        if n <= 0:
            return counts
        u = np.zeros((e.B, n, e.L))
        for b, t in enumerate(traj_indices):
            rng = np.random.default_rng() if p.random_seed is None else np.random.default_rng(np.random.SeedSequence([p.random_seed, int(t), TAG_SHOT]))
            u[b] = rng.random((n, e.L))
        bits = e.sample_shots(u, basis)
        weights = (1 << np.arange(e.L, dtype=object))
        for b in range(e.B):
            for s_ in range(int(shots_per_traj[b])):
                code = int(np.dot(bits[b, s_].astype(object), weights))  # sum(bit_i << i), arbitrary length
                counts[code] = counts.get(code, 0) + 1
        return counts


class Simulator:
    """``Simulator().run(state, hamiltonian, sim_params, noise_model) -> Result`` (simulator.py:1173-1312).

    ``batch`` trajectories are resident on the GPU at a time; with ``torch.distributed``
    initialised, trajectory indices are sharded contiguously over ranks and the observable /
    diagnostic sums are combined with one all-reduce.
    """

    def __init__(self, batch: int | None = None, device: str | None = None, show_progress: bool = False, native: bool = True,
                 parallel: bool = True, max_workers: int | None = None, dtype: str = "complex128", engines: int = 4):
        # parallel / max_workers configure the reference's process pool (simulator.py:60-130); here the trajectories of a run are
        # batched on the GPU instead, so the arguments are accepted for source compatibility and have no effect on the results
        self.parallel, self.max_workers = parallel, max_workers
        # "complex128": the reference's arithmetic (mps.py:231), trajectory-by-trajectory parity.  "complex64": fp32 arithmetic and
        # storage on the device (libtjm_hip_f32.so) - half the HBM per trajectory, twice the vector rate; jump decisions are
        # discontinuous, so parity with the reference is statistical (ensemble means), not per trajectory.
        if dtype not in ("complex128", "complex64"):
            raise ValueError(f'dtype must be "complex128" or "complex64", got {dtype!r}')
        self.dtype = dtype
        self.batch = batch
        self.device = device
        # engines: the trajectories resident at a time are split over this many engines, each with a host thread and a HIP stream of
        # its own, so that the host round trips and latency-bound kernels of one hide behind the kernels of the others (fp64 MFMA
        # and fp64 vector work share one datapath on gfx950, so this is latency hiding, not pipe overlap; measured on the MI355X
        # with the headline configuration:
        # 1 -> 4 engines +6 % at 1024 resident trajectories, +16 % at 128).  Results do not depend on it: a trajectory is a pure
        # function of (seed, index).
        self.engines = max(1, int(engines))
        self._alloc_lock = threading.Lock()
        self._threads_active = 1
        self._engine_kw: dict = {}
        self.show_progress = show_progress
        self.native = native  # True: the C driver tjm_engine_run runs the schedule; False: the Python mirror of it

    def _batch_for(self, remaining: int, length: int, chi: int, mpo, device) -> int:
        """Trajectories resident at once.  ``batch=None``: as many as fit in 60 % of the free HBM (at most ``AUTO_BATCH_MAX``) - small
        bonds are launch-latency-bound, so throughput grows with the batch until the chip is full; an explicit ``batch`` is kept."""
        if self.batch is not None:
            return max(1, min(int(self.batch), remaining, MAX_ENGINE_BATCH))
        import torch

        per_traj = BatchEngine.workspace_bytes_for(length, chi, 64, mpo, **self._engine_kw) / 64.0
        free, _total = torch.cuda.mem_get_info(torch.device(device))
        fit = int(0.6 * free / per_traj / max(1, self._threads_active))  # concurrent engines share what is free
        return max(1, min(remaining, AUTO_BATCH_MAX, fit))

    def _run_growing(self, chunk, chi, chi_top, length, mpo, make_batch, run_piece, device, cols, n_obs, keep_last=False, engine_kw=None,
                     first_fit=None):
        """One chunk of trajectories with storage grown on demand.  A piece that runs out of capacity at time step (gate layer) j
        hands its states, rolled back to the start of j, to engines of twice the capacity - several smaller ones when the memory
        asks for it - which continue from j; only a clip before the first full step starts the piece again from the initial state.
        ``run_piece(batch, lo, hi, resume)`` runs trajectories chunk[lo:hi] on ``batch`` (made by ``make_batch(engine)``).
        Returns (results, diagnostics, engine holding trajectory chunk[0] if ``keep_last``)."""
        res = np.zeros((len(chunk), n_obs, cols))
        dg = np.zeros((len(chunk), 3, cols))
        # (lo, hi, source engine or None, first source slot, (step, phase), rng cursors, extra, capacity)
        pending = [(0, len(chunk), None, 0, (0, 0), None, None, chi)]
        kept = None
        while pending:
            lo, hi, src, first, start, pos, extra, cap_now = pending.pop()
            with self._alloc_lock:  # sizing and allocation are one step: concurrent engines see each other's memory
                # the first piece of a part was sized by the caller for all concurrent parts together
                fit = first_fit if (first_fit is not None and src is None and cap_now == chi) else self._batch_for(hi - lo, length, cap_now, mpo, device)
                if fit < hi - lo:  # the larger engine holds fewer trajectories: the rest of the piece waits
                    pending.append((lo + fit, hi, src, first + fit, start, None if pos is None else pos[fit:], extra, cap_now))
                    hi = lo + fit
                    pos = None if pos is None else pos[:fit]
                engine = BatchEngine(length, cap_now, hi - lo, mpo, device=device, **dict(self._engine_kw, **(engine_kw or {})))
            batch = make_batch(engine)
            try:
                resume = None
                if src is not None:
                    engine.adopt(src, first)
                    resume = dict(start=start, rng_pos=pos, extra=extra, results=np.ascontiguousarray(res[lo:hi]), diagnostics=np.ascontiguousarray(dg[lo:hi]))
                res[lo:hi], dg[lo:hi] = run_piece(batch, lo, hi, resume)
            except CapacityError as err:
                bigger = grown_capacity(cap_now, chi_top, getattr(engine, "d", 2))
                if err.resume is not None and err.resume[0] > 0:
                    res[lo:hi], dg[lo:hi] = err.results, err.diagnostics  # the columns measured so far
                    pending.append((lo, hi, engine, 0, err.resume, err.rng_pos, getattr(err, "extra", None), bigger))
                    engine = None  # stays alive as the source of its successors
                else:
                    pending.append((lo, hi, None, 0, (0, 0), None, None, bigger))
            if src is not None and not any(q[2] is src for q in pending):
                src.close()
            if engine is not None:
                if keep_last and lo == 0 and not any(q[0] == 0 for q in pending):
                    kept = engine
                else:
                    engine.close()
        return res, dg, kept

    def _run_parts(self, chunk, chi, chi_top, length, mpo, make_batch, run_piece, device, cols, n_obs, keep_last=False):
        """``_run_growing`` on ``self.engines`` contiguous parts of the chunk side by side, one host thread and one HIP stream per part
        (results in chunk order; a trajectory is a pure function of (seed, index), so the split never shows in them)."""
        import torch

        E = min(self.engines, len(chunk) // 2)
        if E <= 1 or torch.cuda.device_count() == 0:
            return self._run_growing(chunk, chi, chi_top, length, mpo, make_batch, run_piece, device, cols, n_obs, keep_last)
        from concurrent.futures import ThreadPoolExecutor

        bounds = [len(chunk) * k // E for k in range(E + 1)]

        def work(k):
            o, e = bounds[k], bounds[k + 1]
            stream = torch.cuda.Stream(device=torch.device(device))
            return self._run_growing(chunk[o:e], chi, chi_top, length, mpo, make_batch, lambda tb, lo_, hi_, resume: run_piece(tb, o + lo_, o + hi_, resume),
                                     device, cols, n_obs, keep_last and k == 0, engine_kw={"stream": stream}, first_fit=e - o)

        self._threads_active = E
        try:
            with ThreadPoolExecutor(E) as pool:
                parts = [f.result() for f in [pool.submit(work, k) for k in range(E)]]
        finally:
            self._threads_active = 1
        return np.concatenate([q[0] for q in parts]), np.concatenate([q[1] for q in parts]), parts[0][2]

    def run(self, initial_state: MPS, hamiltonian: MPO, sim_params: AnalogSimParams, noise_model: NoiseModel | None = None, *,
            observables=None, num_traj=None, random_seed=None, get_state: bool = False) -> Result:
        """Same entry point as the reference for both paths (simulator.py:1173-1312): ``DigitalSimParams`` with a list of gate
        layers as operator goes to ``run_circuit``.  The keyword arguments belong to simulation programs (pair lists), which
        are the reference's control plane and not part of this path."""
        import torch

        from .api import DigitalSimParams

        if observables is not None or num_traj is not None or random_seed is not None or get_state:
            raise NotImplementedError("program-wide arguments belong to SimulationProgram runs, which are outside the hot path built here")
        if isinstance(initial_state, (list, tuple)):
            raise NotImplementedError("a list of initial states (deterministic unitary ensemble, simulator.py:1190-1196) is outside the TJM path built here")
        if not hasattr(initial_state, "tensors"):
            raise TypeError("initial_state must be a State (MPS representation).")  # simulator.py:1465-1476
        if isinstance(hamiltonian, (str, os.PathLike)):
            raise NotImplementedError("QASM circuits go through the reference's qiskit front end; pass gate layers (yaqs_amd.api.GateLayer)")
        if sim_params is None:
            raise NotImplementedError("SimulationProgram / pair-list runs are the reference's control plane, outside the path built here")
        if isinstance(sim_params, DigitalSimParams):
            if not isinstance(hamiltonian, (list, tuple)) or not all(hasattr(layer, "singles") for layer in hamiltonian):
                raise TypeError("a circuit run needs a list of gate layers as operator.")  # simulator.py:1537-1544
            top = max([q for layer in hamiltonian for q, _ in layer.singles] + [e[0] + 1 if len(e) == 2 else max(e[0], e[1]) for layer in hamiltonian for e in list(layer.even) + list(layer.odd)] + [0])
            if top >= initial_state.length:
                raise ValueError("State and circuit qubit counts do not match.")  # simulator.py:1738-1741
            return self.run_circuit(initial_state, hamiltonian, sim_params, noise_model)

        pieces = None
        if hasattr(hamiltonian, "per_interval"):  # Hamiltonian.piecewise([(H, duration), ...])
            hamiltonian = hamiltonian.per_interval(sim_params.dt, len(sim_params.times) - 1)
        if isinstance(hamiltonian, (tuple, list)):  # piecewise-constant drive: one MPO per interval
            pieces = list(hamiltonian)
            if not pieces:
                raise ValueError("a piecewise Hamiltonian needs at least one MPO")
            hamiltonian = pieces[0]
        if hamiltonian.length != initial_state.length:
            raise ValueError("State and Hamiltonian must have the same number of sites")  # tdvp.py:91-93
        hamiltonian = _closed_boundaries(hamiltonian)
        if pieces is not None:
            pieces = [_closed_boundaries(h_) for h_ in pieces]
        site_dims = [int(q) for q in (getattr(initial_state, "physical_dimensions", None) or [2] * initial_state.length)]
        d = max(site_dims)
        for h_ in (pieces if pieces is not None else [hamiltonian]):
            if any(int(w.shape[0]) != site_dims[i] or int(w.shape[1]) != site_dims[i] for i, w in enumerate(h_.tensors)):
                raise ValueError("State and Hamiltonian must have the same physical dimensions")
        validate_noise_model_for_run(noise_model, length=initial_state.length, physical_dimensions=site_dims if len(set(site_dims)) > 1 else d,
                                     is_digital=False, sim_params=sim_params)  # simulator.py:1488-1516
        if noise_model is not None:  # one realisation of static disorder per run (simulator.py:1269-1271)
            noise_model = noise_model.sample(rng=disorder_rng(sim_params.random_seed))
        initial_state = _encoded(initial_state)
        user_params, user_noise = sim_params, noise_model
        if len(set(site_dims)) > 1:  # sites of different dimension: zero-padded onto the engine's one-dimension storage
            initial_state, padded, sim_params, noise_model, _ = embed_mixed_dimensions(
                initial_state, [h_.tensors for h_ in (pieces if pieces is not None else [hamiltonian])], sim_params, noise_model)
            if pieces is not None:
                pieces = [MPO(t) for t in padded]
                hamiltonian = pieces[0]
            else:
                hamiltonian = MPO(padded[0])
        rank, world = 0, 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            rank, world = torch.distributed.get_rank(), torch.distributed.get_world_size()
        device = self.device or f"cuda:{int(os.environ.get('LOCAL_RANK', 0))}"
        noisy = noise_model is not None and any(q["strength"] != 0 for q in noise_model.processes)
        if noisy and sim_params.get_state:
            raise ValueError("Cannot return state in noisy analog simulation due to stochastics.")  # simulator.py:1555-1557
        num_traj = sim_params.num_traj if noisy else 1  # simulator.py:1549-1559
        lo, hi = shard_range(num_traj, rank, world)
        mine = list(range(lo, hi))
        chi, chi_top = engine_bond_caps(sim_params, initial_state, can_grow=_noise_can_grow_bonds(noise_model))
        mode = getattr(sim_params, "evolution_mode", "tdvp")
        self._engine_kw = {"cap_slack": 2} if str(getattr(mode, "value", mode)) == "bug" else {}
        if self.dtype != "complex128":
            self._engine_kw["dtype"] = self.dtype
        if d != 2:
            self._engine_kw["d"] = d  # qutrits / four-level sites: one local dimension per chain (engine storage [B][d][cap][cap])
        cols = len(sim_params.times) if sim_params.sample_timesteps else 1
        n_obs = len(sim_params.observables)
        res_all = np.zeros((len(mine), n_obs, cols))
        diag_all = np.zeros((len(mine), 3, cols))
        done = 0
        final = None
        schmidt: dict = {}  # (global trajectory index, sorted row, column) -> 500-entry Schmidt spectrum
        while done < len(mine):
            B = self._batch_for(len(mine) - done, initial_state.length, chi, hamiltonian.tensors, device)
            chunk = mine[done: done + B]

            def make_batch(engine):
                tb = TrajectoryBatch(engine, sim_params, noise_model)  # the backend sees the model as given (simulator.py:1549-1559)
                if pieces is not None:
                    tb.set_intervals(pieces)
                return tb

            def run_piece(tb, lo_, hi_, resume, chunk=chunk):
                try:
                    return tb.run(chunk[lo_:hi_], initial_state if resume is None else None, native=self.native, resume=resume)
                finally:  # also on a capacity hand-over: the columns measured before the clipped step stay valid
                    for (row, col), arr in tb.schmidt.items():
                        for b_, t_ in enumerate(chunk[lo_:hi_]):
                            schmidt[(t_, row, col)] = arr[b_]

            keep = sim_params.get_state and 0 in chunk
            r, dg, last = self._run_parts(chunk, chi, chi_top, initial_state.length, hamiltonian.tensors, make_batch, run_piece, device, cols,
                                          n_obs, keep)
            res_all[done: done + len(chunk)] = r
            diag_all[done: done + len(chunk)] = dg
            done += len(chunk)
            if last is not None:
                # the physical state at the last time: the trajectory state (order 1) or the last sampled copy psi (order 2,
                # analog_tjm.py:331-366); closed-system runs have one trajectory, slot 0 of the first chunk
                use_psi = sim_params.order == 2 and len(sim_params.times) > 1
                from .api import State

                out_t = last.export_state(0, 1 if use_psi else 0)
                if len(set(site_dims)) > 1:  # back to the chain's own dimensions (the added levels hold exact zeros)
                    out_t = [t[: site_dims[i]] for i, t in enumerate(out_t)]
                final = State(tensors=out_t, physical_dimensions=site_dims if d != 2 else None)  # result.output_state is a State (result.py:155-189)
                last.close()
        if world > 1:
            res_all, diag_all = gather_trajectories(res_all, diag_all, num_traj, lo, device)
            if any(ob.gate.name == "schmidt_spectrum" for ob in sim_params.observables):
                schmidt = gather_counts_like(schmidt, device)
        out = Result(user_params, res_all, diag_all, schmidt=schmidt)
        out.output_state = final
        out.noise_model = user_noise  # the sampled realisation the trajectories ran with (result.py:155-189)
        return out

    def run_circuit(self, initial_state: MPS, layers, sim_params, noise_model: NoiseModel | None = None, basis: str = "Z"):
        """Circuit runs (simulator.py:1681-1830 with gate layers instead of a qiskit circuit): observables, diagnostics and, with
        ``sim_params.shots``, the measurement histogram.  Trajectories in chunks of ``batch``; with ``torch.distributed``
        initialised they are sharded contiguously over the ranks, and the per-trajectory rows and the histogram are combined at the
        end."""
        import torch

        from .api import CircuitResult

        if any(int(q) != 2 for q in (getattr(initial_state, "physical_dimensions", None) or [2])):
            raise NotImplementedError("circuit runs are built for qubits (the gate library is 2 x 2 / 4 x 4)")
        validate_noise_model_for_run(noise_model, length=initial_state.length, is_digital=True, sim_params=sim_params)  # simulator.py:1867-1873
        if noise_model is not None:
            noise_model = noise_model.sample(rng=disorder_rng(sim_params.random_seed))
        initial_state = _encoded(initial_state)
        noisy = noise_model is not None and any(q["strength"] != 0 for q in noise_model.processes)
        num_traj, per_call, distribution = plan_digital_shots(sim_params, noisy)
        device = self.device or f"cuda:{int(os.environ.get('LOCAL_RANK', 0))}"
        self._engine_kw = {}
        distant = any(len(en) == 3 and abs(en[0] - en[1]) > 1 for layer in layers for en in list(layer.even) + list(layer.odd))
        slack = 1
        if distant and getattr(sim_params, "gate_mode", "mpo") == "mpo":
            # the gate-MPO product multiplies the bonds under the gate by its operator Schmidt rank (<= 4) until MPS.compress cuts
            # them back (digital_tjm.py:536-557): the storage holds four times the cap and four times the exact ranks near the ends
            slack = 4
            self._engine_kw = {"cap_slack": 4}
        if self.dtype != "complex128":
            self._engine_kw["dtype"] = self.dtype
        chi, chi_top = engine_bond_caps(sim_params, initial_state, can_grow=True, slack=slack)  # every TEBD gate is a truncated split
        mid = sim_params.num_mid_measurements if sim_params.sample_layers else 0
        cols = (mid + 2) if sim_params.sample_layers else 1
        res_all = np.zeros((num_traj, len(sim_params.observables), cols))
        diag_all = np.zeros((num_traj, 3, cols))
        counts: dict[int, int] = {}
        schmidt: dict = {}
        wants_shots = sim_params.shots is not None
        identity_mpo = [np.eye(2, dtype=np.complex128).reshape(2, 2, 1, 1)] * initial_state.length  # the circuit path never applies it
        rank, world = 0, 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            rank, world = torch.distributed.get_rank(), torch.distributed.get_world_size()
        first, last = shard_range(num_traj, rank, world)
        res_all = res_all[: last - first]
        diag_all = diag_all[: last - first]
        done = 0
        while done < last - first:
            B = self._batch_for(last - first - done, initial_state.length, chi, identity_mpo, device)
            chunk = list(range(first + done, first + min(done + B, last - first)))
            spt = [shots_for_trajectory(t, per_call, distribution) for t in chunk] if wants_shots else None

            def run_piece(db, lo_, hi_, resume, chunk=chunk, spt=spt):
                try:
                    out = db.run(chunk[lo_:hi_], initial_state, layers, shots_per_traj=None if spt is None else spt[lo_:hi_], basis=basis, resume=resume)
                finally:
                    for (row, col), arr in db.schmidt.items():
                        for b_, t_ in enumerate(chunk[lo_:hi_]):
                            schmidt[(t_, row, col)] = arr[b_]
                with self._alloc_lock:  # engines of one chunk finish on threads of their own
                    for k, v in (db.counts or {}).items():
                        counts[k] = counts.get(k, 0) + v
                return out

            r, dg, _ = self._run_parts(chunk, chi, chi_top, initial_state.length, identity_mpo, lambda eng: DigitalBatch(eng, sim_params, noise_model if noisy else None),
                                       run_piece, device, cols, len(sim_params.observables))
            res_all[done: done + len(chunk)] = r
            diag_all[done: done + len(chunk)] = dg
            done += len(chunk)
        if world > 1:
            res_all, diag_all = gather_trajectories(res_all, diag_all, num_traj, first, device)
            if wants_shots:
                counts = gather_counts(counts, device)
            if any(ob.gate.name == "schmidt_spectrum" for ob in sim_params.observables):
                schmidt = gather_counts_like(schmidt, device)
        has_obs = len(sim_params.observables) > 0  # a shots-only run reports no diagnostics (result.py:155-189)
        out = CircuitResult(sim_params, res_all if has_obs else None, diag_all, counts if wants_shots else None, schmidt=schmidt)
        out.noise_model = noise_model
        return out


def _closed_boundaries(mpo: MPO) -> MPO:
    """An MPO whose outer bonds are wider than 1 (the reference's coupled-transmon chain of even length ends on a resonator tensor with
    an open right bond of 4, mpo.py:549-668) meets boundary environments that are the identity for EVERY channel of that bond
    (integrators.py:186-193, primitives.py:139-174), i.e. the channels are summed: the same operator with outer bonds of 1 is the
    boundary tensor summed over them."""
    t = list(mpo.tensors)
    if t[0].shape[2] == 1 and t[-1].shape[3] == 1:
        return mpo
    t[0] = np.asarray(t[0]).sum(axis=2, keepdims=True)
    t[-1] = np.asarray(t[-1]).sum(axis=3, keepdims=True)
    return MPO(t)


def _pad_operator(matrix, site_dims, d: int) -> np.ndarray:
    """A local operator on sites of dimensions ``site_dims`` as an operator on sites of dimension ``d`` (zero on the added levels)."""
    k = len(site_dims)
    m = np.asarray(matrix, dtype=np.complex128).reshape(tuple(site_dims) * 2)
    out = np.zeros((d,) * (2 * k), dtype=np.complex128)
    out[tuple(slice(0, q) for q in site_dims) * 2] = m
    return out.reshape(d ** k, d ** k)


def embed_mixed_dimensions(initial_state, mpos, sim_params, noise_model):
    """Chains whose sites differ in dimension (the reference's coupled-transmon chains, mpo.py:549-668) run on the engine's uniform
    storage ``[B][d][cap][cap]`` with d = the largest local dimension: state tensors, MPO tensors, jump operators and observables are
    zero-padded on the physical legs.  A Hamiltonian, a dissipator and a jump operator that are zero on the added levels never
    populate them, so every expectation value, jump probability and singular value is that of the original chain; the output state is
    cut back to its own dimensions.  Returns (state, [mpo tensor lists], sim_params, noise_model, dims)."""
    import copy

    dims = [int(q) for q in initial_state.physical_dimensions]
    d = max(dims)
    st = MPS(initial_state.length, tensors=[np.pad(np.asarray(t, dtype=np.complex128), ((0, d - t.shape[0]), (0, 0), (0, 0))) for t in initial_state.tensors],
             physical_dimensions=[d] * initial_state.length)
    padded_mpos = []
    for tensors in mpos:
        for i, w in enumerate(tensors):
            if int(w.shape[0]) != dims[i] or int(w.shape[1]) != dims[i]:
                raise ValueError("State and Hamiltonian must have the same physical dimensions")
        padded_mpos.append([np.pad(np.asarray(w, dtype=np.complex128), ((0, d - w.shape[0]), (0, d - w.shape[1]), (0, 0), (0, 0))) for w in tensors])
    params = copy.copy(sim_params)
    obs = []
    for ob in sim_params.observables:
        if ob.gate.name in META_OBSERVABLES:
            obs.append(ob)
            continue
        sites = list(ob.sites) if isinstance(ob.sites, (list, tuple)) else [ob.sites]
        clone = copy.copy(ob)
        clone.gate = copy.copy(ob.gate)
        clone.gate.matrix = _pad_operator(ob.gate.matrix, [dims[q] for q in sites], d)
        obs.append(clone)
    params.observables = obs
    noise = noise_model
    if noise_model is not None:
        noise = copy.copy(noise_model)
        procs = []
        for proc in noise_model.processes:
            q = dict(proc)
            if "matrix" in q:
                q["matrix"] = _pad_operator(q["matrix"], [dims[s_] for s_ in q["sites"]], d)
            if "factors" in q:
                q["factors"] = tuple(_pad_operator(f, [dims[s_]], d) for f, s_ in zip(q["factors"], q["sites"]))
            procs.append(q)
        noise.processes = procs
        jumps = []
        for jump in getattr(noise_model, "scheduled_jumps", None) or []:
            q = dict(jump)
            if "matrix" in q:
                q["matrix"] = _pad_operator(q["matrix"], [dims[s_] for s_ in q["sites"]], d)
            jumps.append(q)
        if jumps:
            noise.scheduled_jumps = jumps
    return st, padded_mpos, params, noise, dims


def _encoded(state: MPS) -> MPS:
    """``State._encode("mps")`` (state.py:278-297): the run works on a copy brought to B-normal form (centre 0, unit norm)."""
    out = MPS(state.length, tensors=[np.array(t, dtype=np.complex128, copy=True) for t in state.tensors],
              physical_dimensions=list(getattr(state, "physical_dimensions", None) or [2] * state.length))
    out.normalize("B")
    return out


MAX_CHI = 512   # largest bond the engine serves: the kernels hold d * chi <= 1024 (1024 x 1024 two-site splits, 1024-row Householder
# panels).  The engine at chi = 512 - gate, SVD and QR centre shifts, a whole two-site TDVP sweep of a 20-site saturated chain -
# agrees with the oracle on the MI355X (tests/test_hip_round2.py::test_bonds_up_to_512_*, profiles/r03_gpu_logs/c9_chi512_*.log).


def max_chi(d: int = 2) -> int:
    """Largest bond for local dimension d: the two-site matrix has d * chi rows."""
    return min(MAX_CHI, 1024 // max(int(d), 2))


START_CHI = 8   # first storage capacity tried when the requested cap is larger
AUTO_BATCH_MAX = 16384  # trajectories in flight when Simulator(batch=None) sizes the batch itself
MAX_ENGINE_BATCH = 65535  # the trajectory index is a y / z grid dimension of the kernels: tjm_engine_create refuses more


def engine_bond_caps(sim_params, initial_state, can_grow: bool = False, slack: int = 1) -> tuple[int, int]:
    """``(first, top)`` storage capacities of the engine.

    The reference's bonds are dynamic and its presets ask for ``max_bond_dim`` = 128, 4096 or no cap at all
    (simulation_parameters.py:46-51) while the bonds of most runs stay far below that.  The engine's storage is static, so a
    run starts with a small capacity and is repeated with twice the capacity whenever a truncation was clipped by it
    (``CapacityError``): the final pass is one in which ``max_bond_dim`` and the threshold alone decided every truncation,
    exactly as in the reference.  Work grows with chi**3, so the discarded passes cost at most 1/7 of the last one.
    ``top`` is the largest capacity that can ever be needed: ``max_bond_dim``, the exact Schmidt-rank bound ``2**(L//2)`` and
    the bonds of the initial state.  ``can_grow``: something besides the TDVP sweep can enlarge a bond (an adjacent non-Pauli
    two-site process or a scheduled pair jump goes through a merged, truncated split; so does every TEBD gate), so a one-site
    TDVP run is not confined to the bonds of its initial state.
    """
    have = max(max(t.shape[1], t.shape[2]) for t in initial_state.tensors)
    d = int(initial_state.physical_dimensions[0]) if getattr(initial_state, "physical_dimensions", None) else 2
    exact = min(d ** min(initial_state.length // 2, 30), 1 << 30)
    mode = getattr(sim_params, "evolution_mode", "tdvp")
    bug = str(getattr(mode, "value", mode)) == "bug"
    if bug:
        exact *= 2  # a stacked trial basis holds up to twice the Schmidt rank of its cut until the next canonicalisation (cap_slack = 2)
    exact *= slack  # gate-MPO products: bonds up to `slack` times the exact rank / the cap between the product and its compression
    want = exact if sim_params.max_bond_dim is None else min(slack * int(sim_params.max_bond_dim), exact)
    if getattr(sim_params, "tdvp_mode", "2site") == "dynamic" and sim_params.max_bond_dim is not None and not bug:
        # the two-site branch of the dynamic sweep splits without a cap (split_tdvp(dynamic=True)): a bond next to one below the cap
        # can reach d * (max_bond_dim - 1) before _cap_bonds cuts it back at the start of the next sweep
        # (d = local dimension: 2 max_bond_dim for qubits, 3 / 4 max_bond_dim for qutrit / four-level chains)
        want = min(d * int(sim_params.max_bond_dim), exact)
        if d * want > 512 and want > int(sim_params.max_bond_dim):
            # _cap_bonds' sqrt-distributed split (tjm_engine_step_cap_bond) is served by the plain Jacobi split, which holds d * bond <= 512
            raise NotImplementedError(f"tdvp_mode='dynamic' with local dimension {d} and max_bond_dim {sim_params.max_bond_dim}: its uncapped "
                                      f"two-site splits need bonds up to {want}, the capped re-split holds d * bond <= 512")
    if bug and sim_params.max_bond_dim is not None:
        # each of the two half-sweeps of a BUG step stacks [retained | predictor] along the left bond of every site: bonds reach
        # 4 * max_bond_dim before the single compression at the end of the step (bug.py:213-257)
        want = min(4 * int(sim_params.max_bond_dim), exact)
    top = max(want, have)
    if have > max_chi(d):
        raise NotImplementedError(f"bond dimension {have} of the initial state exceeds the supported chi <= {max_chi(d)} for local dimension {d}")
    if getattr(sim_params, "tdvp_mode", "2site") == "1site" and not can_grow:
        return have, have  # one-site TDVP never changes a bond (integrators.py:44-158)
    return max(have, min(top, START_CHI)), top


CAPACITY_LADDER = (8, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512)  # steps of x1.5 / x1.33: work grows with chi**3, a re-padding copy is cheap


def grown_capacity(chi: int, top: int, d: int = 2) -> int:
    if chi >= top:
        raise RuntimeError("a truncation was clipped although the engine holds max_bond_dim")  # cannot happen: svd_finish_kernel
    limit = max_chi(d)
    if chi >= limit:
        raise NotImplementedError(f"the run needs bonds beyond {chi}; the HIP path holds chi <= {limit} for local dimension {d}")
    ladder = CAPACITY_LADDER if os.environ.get("TJM_CAPACITY_DOUBLING") is None else (8, 16, 32, 64, 128, 256, 512)
    return min(next(c for c in ladder if c > chi), top, limit)


def engine_bond_cap(sim_params, initial_state) -> int:
    """Largest capacity a run can need (``engine_bond_caps(...)[1]``), refused when beyond the supported size."""
    top = engine_bond_caps(sim_params, initial_state)[1]
    d = int(initial_state.physical_dimensions[0]) if getattr(initial_state, "physical_dimensions", None) else 2
    if top > max_chi(d):
        raise NotImplementedError(f"bond dimension {top} exceeds the supported chi <= {max_chi(d)} for local dimension {d}")
    return top


def _require_capacity(engine) -> None:
    if engine.capacity_overflow():
        raise CapacityError(f"a truncation needed a bond beyond the engine's capacity chi = {engine.chi_max}")


def plan_digital_shots(sim_params, noisy: bool):
    """``_plan_digital_shots`` (simulator.py:1001-1050): (effective_num_traj, per_call_shots | None, (total, n_traj) | None)."""
    wants_obs = bool(sim_params.observables)
    wants_shots = sim_params.shots is not None
    if wants_shots and not wants_obs:
        return (sim_params.shots, 1, None) if noisy else (1, sim_params.shots, None)
    if wants_obs:
        n = sim_params.num_traj if noisy else 1
        if wants_shots:
            return (n, None, (sim_params.shots, n)) if noisy else (n, sim_params.shots, None)
        return n, None, None
    return 1, None, None


def shots_for_trajectory(traj: int, per_call, distribution) -> int:
    """``_per_call_shots`` (digital_tjm.py:750-766)."""
    if per_call is not None:
        return int(per_call)
    if distribution is not None:
        base, rem = divmod(int(distribution[0]), int(distribution[1]))
        return base + (1 if traj < rem else 0)
    return 0


def shard_range(num_traj: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous trajectory-index range of ``rank``."""
    return (num_traj * rank) // world, (num_traj * (rank + 1)) // world


def _noise_can_grow_bonds(noise_model) -> bool:
    """True when a process of the model is applied through a merged two-site split (dissipation.py:158-171,
    stochastic_process.py:268-288, scheduled_jumps.py:88-106): adjacent non-Pauli pairs and scheduled pair jumps."""
    if noise_model is None:
        return False
    for q in noise_model.processes:
        if len(q["sites"]) == 2 and abs(q["sites"][1] - q["sites"][0]) == 1 and not is_pauli(q):
            return True
    return any(len(j["sites"]) == 2 for j in (getattr(noise_model, "scheduled_jumps", None) or []))


def gather_counts_like(table: dict, device) -> dict:
    """Union of per-rank dictionaries with disjoint keys (the Schmidt spectra of each rank's trajectories)."""
    import contextlib

    import torch
    import torch.distributed as dist

    parts = [None] * dist.get_world_size()
    ctx = torch.cuda.device(torch.device(device)) if dist.get_backend() == "nccl" else contextlib.nullcontext()
    with ctx:
        dist.all_gather_object(parts, dict(table))
    out: dict = {}
    for part in parts:
        out.update(part)
    return out


def gather_counts(counts: dict, device) -> dict:
    """Sum of the per-rank measurement histograms {basis state: occurrences}.  The keys are Python integers of L bits (more than 64
    on long chains), so the tables travel as objects."""
    import contextlib

    import torch
    import torch.distributed as dist

    parts = [None] * dist.get_world_size()
    ctx = torch.cuda.device(torch.device(device)) if dist.get_backend() == "nccl" else contextlib.nullcontext()
    with ctx:  # the object collectives of the nccl backend stage through the current device
        dist.all_gather_object(parts, dict(counts))
    total: dict[int, int] = {}
    for part in parts:
        for key, val in part.items():
            total[int(key)] = total.get(int(key), 0) + int(val)
    return total


def gather_trajectories(res: np.ndarray, diag: np.ndarray, num_traj: int, lo: int, device):
    """Place this rank's rows into the global buffers and all-reduce(sum) them over ranks."""
    import torch
    import torch.distributed as dist

    full_r = np.zeros((num_traj,) + res.shape[1:])
    full_d = np.zeros((num_traj,) + diag.shape[1:])
    full_r[lo: lo + res.shape[0]] = res
    full_d[lo: lo + diag.shape[0]] = diag
    dev = torch.device(device) if dist.get_backend() == "nccl" else torch.device("cpu")
    tr = torch.from_numpy(full_r).to(dev)
    td = torch.from_numpy(full_d).to(dev)
    dist.all_reduce(tr, op=dist.ReduceOp.SUM)
    dist.all_reduce(td, op=dist.ReduceOp.SUM)
    return tr.cpu().numpy(), td.cpu().numpy()
