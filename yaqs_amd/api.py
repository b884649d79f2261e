"""Host-side mirror of the reference's simulation interface for the TJM path.

Same names, argument meaning and error behaviour as the reference objects the path
touches (paths relative to /root/reference/src/mqt/yaqs):

* ``Observable``, ``AnalogSimParams``   core/data_structures/simulation_parameters.py:330-613
* ``NoiseModel`` / ``is_pauli``         core/data_structures/noise_model.py:227-665
* ``MPS`` (product / Haar states)       core/data_structures/mps.py:54-301
* ``MPO.ising`` / ``MPO.heisenberg``    core/data_structures/mpo.py:326-406 (plain FSM form)
* ``Result``                            core/data_structures/result.py:34-189
* ``Simulator.run``                     simulator.py:1173-1312, 1444-1679 (MPS analog branch)

Only host bookkeeping lives here; every tensor operation runs in the HIP library.
"""
from __future__ import annotations

import re
from dataclasses import dataclass
from typing import Any, Sequence

import numpy as np

C128 = np.complex128

_X = np.array([[0, 1], [1, 0]], dtype=C128)
_Y = np.array([[0, -1j], [1j, 0]], dtype=C128)
_Z = np.array([[1, 0], [0, -1]], dtype=C128)
_I = np.eye(2, dtype=C128)
PAULI_MAP = {"x": _X, "y": _Y, "z": _Z}


# ------------------------------------------------------------------ gates / observables
@dataclass
class BaseGate:
    name: str
    matrix: np.ndarray
    interaction: int = 1
    sites: Any = None

    def set_sites(self, sites):
        self.sites = sites


def X() -> BaseGate:
    return BaseGate("x", _X.copy())


def Y() -> BaseGate:
    return BaseGate("y", _Y.copy())


def Z() -> BaseGate:
    return BaseGate("z", _Z.copy())


def Id() -> BaseGate:
    return BaseGate("id", _I.copy())


def XX() -> BaseGate:
    """Two-site correlator X (x) X (gate_library.py:1674-1698)."""
    return BaseGate("xx", np.kron(_X, _X), interaction=2)


def YY() -> BaseGate:
    """Two-site correlator Y (x) Y (gate_library.py:1700-1724)."""
    return BaseGate("yy", np.kron(_Y, _Y), interaction=2)


def ZZ() -> BaseGate:
    """Two-site correlator Z (x) Z (gate_library.py:1726-1750)."""
    return BaseGate("zz", np.kron(_Z, _Z), interaction=2)


def Entropy() -> BaseGate:
    """Meta-observable: entanglement entropy across the cut (i, i+1) (gate_library.py:1873-1900, mps.py:604-641)."""
    return BaseGate("entropy", _I.copy(), interaction=2)


def SchmidtSpectrum() -> BaseGate:
    """Meta-observable: singular values across the cut (i, i+1), padded to 500 with NaN (gate_library.py:1903-1930, mps.py:643-678)."""
    return BaseGate("schmidt_spectrum", _I.copy(), interaction=2)


def PVM(bitstring: str) -> BaseGate:
    """Projection onto a computational-basis string, site 0 first (gate_library.py:1796-1811, mps.py:1495-1537)."""
    g = BaseGate("pvm", _I.copy())
    g.bitstring = bitstring
    return g


META_OBSERVABLES = ("entropy", "schmidt_spectrum", "pvm")


class Observable:
    """``Observable(gate, sites)`` (simulation_parameters.py:330-416): a gate instance, a gate name (``"x"``, ``"zz"``, ``"entropy"``,
    ``"schmidt_spectrum"``), a local operator matrix, or - any other string - a computational-basis projector (PVM) on that bitstring."""

    def __init__(self, gate: BaseGate | str | np.ndarray, sites: int | list[int] | None = None, **parameters):
        if isinstance(gate, str):
            table = {"x": X, "y": Y, "z": Z, "id": Id, "xx": XX, "yy": YY, "zz": ZZ, "entropy": Entropy, "schmidt_spectrum": SchmidtSpectrum}
            if gate == "position":  # gate_library.py:1845-1872: a one-site operator diagonal in a supplied position basis
                extra = set(parameters) - {"positions"}
                if extra or "positions" not in parameters:
                    raise TypeError("Observable 'position' takes exactly the keyword argument 'positions'")
                pos = np.asarray(parameters["positions"])
                if np.iscomplexobj(pos):
                    raise ValueError("positions must contain only real values.")
                pos = np.asarray(pos, dtype=np.float64)
                if pos.ndim != 1 or pos.size == 0:
                    raise ValueError("positions must be a non-empty one-dimensional array.")
                if not np.all(np.isfinite(pos)):
                    raise ValueError("positions must contain only finite values.")
                gate, parameters = BaseGate("position", np.diag(pos).astype(C128), interaction=1), {}
            if isinstance(gate, str) and parameters:
                if gate.lower() in table:
                    raise TypeError(f"Observable {gate!r} got an unexpected keyword argument {next(iter(parameters))!r}")
                if gate == "pvm" or set(gate) <= {"0", "1"}:
                    raise TypeError("'pvm' does not accept observable parameters")
                raise TypeError(f"Unknown observable {gate!r}")
            if isinstance(gate, str):
                gate = table[gate.lower()]() if gate.lower() in table else PVM(gate)  # simulation_parameters.py:392-401: fall back to a PVM
        elif isinstance(gate, np.ndarray):
            if parameters:
                raise TypeError("Observable parameters are only supported for named observables")
            m = np.asarray(gate, dtype=C128)
            if m.ndim != 2 or m.shape[0] != m.shape[1]:
                raise ValueError("Local operator matrix must be a square two-dimensional array")
            gate = BaseGate("local", m, interaction=2 if m.shape == (4, 4) else 1)
        elif parameters:
            raise TypeError("Observable parameters are only supported for named observables")
        self.gate = gate
        self.sites = sites
        if sites is not None:
            self.gate.set_sites(sites)

    @property
    def first_site(self) -> int:
        if self.sites is None:  # PVMs act on the whole chain; they sort behind the local observables
            return 0
        return self.sites[0] if isinstance(self.sites, (list, tuple)) else int(self.sites)


# ------------------------------------------------------------------ parameters
SIMULATION_PRESETS = {
    # simulation_parameters.py:46-51
    "fast": dict(svd_threshold=1e-3, max_bond_dim=16, num_traj=128, krylov_tol=1e-3),
    "balanced": dict(svd_threshold=1e-6, max_bond_dim=128, num_traj=256, krylov_tol=1e-4),
    "accurate": dict(svd_threshold=1e-9, max_bond_dim=4096, num_traj=1024, krylov_tol=1e-6),
    "exact": dict(svd_threshold=1e-13, max_bond_dim=None, num_traj=1024, krylov_tol=1e-12),
}
_USE_PRESET = object()
_TRUNC = ("discarded_weight", "relative", "hard_cutoff", "relative_discarded_weight")
_TDVP_MODES = ("1site", "2site", "dynamic")
_GATE_MODES = ("swaps", "tdvp", "full-tdvp", "mpo")


class EvolutionMode(str, __import__("enum").Enum):
    """simulation_parameters.py:300-336."""

    TDVP = "tdvp"
    BUG = "bug"


def _real_number(value, what: str) -> float:
    if isinstance(value, (bool, np.bool_)) or not isinstance(value, (int, float, np.integer, np.floating)):
        raise TypeError(f"{what} must be a real number, got {type(value).__name__}.")
    return float(value)


def _validate_time_grid(elapsed_time, dt) -> tuple[float, float, int]:
    """simulation_parameters.py:100-170: dt finite and positive, elapsed_time finite and non-negative, their ratio an integer up to
    float64 rounding dust (a few ulps of the larger of elapsed_time and dt), never a genuine fraction of a step."""
    elapsed_time, dt = _real_number(elapsed_time, "elapsed_time"), _real_number(dt, "dt")
    if not np.isfinite(elapsed_time) or not np.isfinite(dt):
        raise ValueError("elapsed_time and dt must be finite.")
    if elapsed_time < 0:
        raise ValueError("elapsed_time must be non-negative.")
    if dt <= 0:
        raise ValueError("dt must be positive.")
    with np.errstate(over="ignore"):
        ratio = elapsed_time / dt
    if not np.isfinite(ratio):
        raise ValueError("elapsed_time / dt must be finite.")
    n_steps = int(round(ratio))
    slack = 8 * np.finfo(np.float64).eps * max(abs(elapsed_time), dt * max(n_steps, 1))
    if abs(n_steps * dt - elapsed_time) > slack or (n_steps == 0 and elapsed_time != 0.0):
        raise ValueError("elapsed_time must be an integer multiple of dt.")
    return elapsed_time, dt, n_steps


def _validate_tdvp_sweeps(value) -> int:
    if isinstance(value, (bool, np.bool_)) or not isinstance(value, (int, np.integer)):
        raise TypeError("tdvp_sweeps must be an int.")
    if value < 1:
        raise ValueError("tdvp_sweeps must be at least 1.")
    return int(value)


def _common_knobs(self, *, preset, num_traj, max_bond_dim, trunc_mode, svd_threshold, krylov_tol, random_seed, tdvp_sweeps, tdvp_mode) -> None:
    """Validation shared by AnalogSimParams and DigitalSimParams (simulation_parameters.py:216-300)."""
    if not isinstance(preset, str) or preset not in SIMULATION_PRESETS:
        raise ValueError(f"preset must be one of {tuple(SIMULATION_PRESETS)}, got {preset!r}.")
    pv = SIMULATION_PRESETS[preset]
    if not isinstance(trunc_mode, str) or trunc_mode not in _TRUNC:
        raise ValueError(f"trunc_mode must be one of {_TRUNC}, got {trunc_mode!r}.")
    if not isinstance(tdvp_mode, str) or tdvp_mode not in _TDVP_MODES:
        raise ValueError(f"tdvp_mode must be one of {_TDVP_MODES}, got {tdvp_mode!r}.")
    if random_seed is not None:
        if isinstance(random_seed, (bool, np.bool_)) or not isinstance(random_seed, (int, np.integer)):
            raise TypeError("random_seed must be int or None.")
        if random_seed < 0:
            raise ValueError("random_seed must be non-negative.")
    if max_bond_dim is not _USE_PRESET and max_bond_dim is not None and (isinstance(max_bond_dim, bool) or not isinstance(max_bond_dim, (int, np.integer))):
        raise TypeError(f"max_bond_dim must be int, None, or omitted, got {type(max_bond_dim).__name__}.")
    self.preset = preset
    self.num_traj = num_traj if num_traj is not None else pv["num_traj"]
    self.max_bond_dim = pv["max_bond_dim"] if max_bond_dim is _USE_PRESET else max_bond_dim
    self.trunc_mode = trunc_mode
    self.svd_threshold = float(svd_threshold if svd_threshold is not None else pv["svd_threshold"])
    if not np.isfinite(self.svd_threshold) or self.svd_threshold < 0.0:
        raise ValueError(f"svd_threshold must be a finite non-negative float, got {svd_threshold!r}.")
    self.krylov_tol = float(krylov_tol if krylov_tol is not None else pv["krylov_tol"])
    if not np.isfinite(self.krylov_tol) or self.krylov_tol <= 0.0:
        raise ValueError(f"krylov_tol must be a finite positive float, got {krylov_tol!r}.")
    self.random_seed = None if random_seed is None else int(random_seed)
    self.tdvp_sweeps = _validate_tdvp_sweeps(tdvp_sweeps)
    self.tdvp_mode = tdvp_mode


class _ObservableOrdering:
    """Site-sorted worker order of the observables, PVMs last (simulation_parameters.py:419-475); derived on every access."""

    def _ordering(self):
        local = [i for i, ob in enumerate(self.observables) if ob.gate.name != "pvm"]
        pvms = [i for i, ob in enumerate(self.observables) if ob.gate.name == "pvm"]
        return sorted(local, key=lambda i: (self.observables[i].first_site, i)) + pvms

    @property
    def sorted_observables(self):
        return [self.observables[i] for i in self._ordering()]

    @property
    def observable_sorted_indices(self):
        out = [0] * len(self.observables)
        for row, user in enumerate(self._ordering()):
            out[user] = row
        return tuple(out)


class AnalogSimParams(_ObservableOrdering):
    """Numerical knobs of the analog TJM path (simulation_parameters.py:520-613)."""

    def __init__(self, observables=None, elapsed_time: float = 0.1, dt: float = 0.1, num_traj: int | None = None,
                 max_bond_dim=_USE_PRESET, trunc_mode: str = "discarded_weight", svd_threshold: float | None = None,
                 krylov_tol: float | None = None, order: int = 1, *, preset: str = "balanced", sample_timesteps: bool = True,
                 get_state: bool = False, random_seed: int | None = None, tdvp_sweeps: int = 1, tdvp_mode: str = "2site",
                 evolution_mode="tdvp"):
        try:
            self.evolution_mode = evolution_mode if isinstance(evolution_mode, EvolutionMode) else EvolutionMode(evolution_mode)
        except ValueError:
            raise ValueError(f"evolution_mode must be one of {tuple(m.value for m in EvolutionMode)}, got {evolution_mode!r}.") from None
        self.elapsed_time, self.dt, n_steps = _validate_time_grid(elapsed_time, dt)
        _common_knobs(self, preset=preset, num_traj=num_traj, max_bond_dim=max_bond_dim, trunc_mode=trunc_mode, svd_threshold=svd_threshold,
                      krylov_tol=krylov_tol, random_seed=random_seed, tdvp_sweeps=tdvp_sweeps, tdvp_mode=tdvp_mode)
        self.observables = [] if observables is None else list(observables)
        self.times = self.dt * np.arange(n_steps + 1, dtype=np.float64)
        if n_steps > 0:
            self.times[-1] = self.elapsed_time
        self.sample_timesteps = sample_timesteps
        self.order = order
        self.get_state = get_state


class DigitalSimParams(_ObservableOrdering):
    """Truncation / sampling knobs of the circuit path (simulation_parameters.py:616-745); keyword-only; ``dt`` is fixed to 1."""

    def __init__(self, *, observables=None, shots: int | None = None, num_traj: int | None = None, max_bond_dim=_USE_PRESET,
                 trunc_mode: str = "discarded_weight", svd_threshold: float | None = None, krylov_tol: float | None = None,
                 preset: str = "balanced", get_state: bool = False, sample_layers: bool = False, num_mid_measurements: int = 0,
                 random_seed: int | None = None, gate_mode: str = "mpo", tdvp_sweeps: int = 1, tdvp_mode: str = "2site"):
        if not isinstance(gate_mode, str) or gate_mode not in _GATE_MODES:
            raise ValueError(f"gate_mode must be one of {_GATE_MODES}, got {gate_mode!r}.")  # simulation_parameters.py:175-190
        self.gate_mode = gate_mode  # nearest-neighbour gates are TEBD in every mode; "swaps" also routes distant pairs with TEBD
        if shots is not None and (isinstance(shots, bool) or not isinstance(shots, (int, np.integer)) or shots < 1):
            raise ValueError("shots must be a positive int or None.")
        _common_knobs(self, preset=preset, num_traj=num_traj, max_bond_dim=max_bond_dim, trunc_mode=trunc_mode, svd_threshold=svd_threshold,
                      krylov_tol=krylov_tol, random_seed=random_seed, tdvp_sweeps=tdvp_sweeps, tdvp_mode=tdvp_mode)
        self.observables = [] if observables is None else list(observables)
        n_pvm = sum(ob.gate.name == "pvm" for ob in self.observables)
        assert n_pvm in (0, len(self.observables)), "PVM observables cannot be mixed with other observables"  # simulation_parameters.py:700-712
        self.shots = shots
        self.sample_layers = sample_layers
        self.num_mid_measurements = num_mid_measurements
        self.get_state = get_state
        self.dt = 1.0  # simulation_parameters.py:667


@dataclass
class GateLayer:
    """One pre-compiled execution layer (the build's replacement of the qiskit DAG front end; digital_tjm.py:49-68).

    ``singles``: [(site, 2x2 matrix)]; ``even`` / ``odd``: [(left_site, U[out_l, out_r, in_l, in_r])] nearest-neighbour gates on
    (left_site, left_site + 1) in the index order of mpo_utils.py:104-159; ``sample_points``: measurement barriers after the layer.
    """

    singles: list
    even: list
    odd: list
    sample_points: int = 0


def rx_matrix(theta: float) -> np.ndarray:
    """gate_library.py:949-978."""
    c, s_ = np.cos(theta / 2), np.sin(theta / 2)
    return np.array([[c, -1j * s_], [-1j * s_, c]], dtype=C128)


def rzz_tensor(theta: float) -> np.ndarray:
    """gate_library.py:1612-1646 as U[out_l, out_r, in_l, in_r]."""
    return np.diag(np.exp(-0.5j * theta * np.array([1, -1, -1, 1]))).astype(C128).reshape(2, 2, 2, 2)


def ising_trotter_layers(length: int, J: float, g: float, dt: float, steps: int, sample_each: bool = False) -> list:
    """Gate sequence of ``create_ising_circuit`` (circuit_library.py:28-79)."""
    out = []
    for _ in range(steps):
        singles = [(q, rx_matrix(-2.0 * dt * g)) for q in range(length)]
        even = [(q, rzz_tensor(-2.0 * dt * J)) for q in range(0, length - 1, 2)]
        odd = [(q, rzz_tensor(-2.0 * dt * J)) for q in range(1, length - 1, 2)]
        out.append(GateLayer(singles, even, odd, 1 if sample_each else 0))
    return out


# ------------------------------------------------------------------ noise
_LIB_OPS = {
    "pauli_x": _X, "pauli_y": _Y, "pauli_z": _Z,
    "x": _X, "y": _Y, "z": _Z,
    "lowering": np.array([[0, 1], [0, 0]], dtype=C128),
    "raising": np.array([[0, 0], [1, 0]], dtype=C128),
    "dephasing": _Z, "bitflip": _X, "bitphaseflip": _Y,
}


def _unit_phase_of(m: np.ndarray, refs) -> bool:
    for ref in refs:
        nz = np.abs(ref) > 0
        if np.any(np.abs(m[~nz]) > 1e-12):
            continue
        r = m[nz] / ref[nz]
        if np.allclose(r, r[0], atol=1e-12) and abs(abs(r[0]) - 1.0) < 1e-12:
            return True
    return False


def is_pauli(proc: dict[str, Any]) -> bool:
    """noise_model.py:644-665."""
    sites = proc["sites"]
    singles = [_X, _Y, _Z]
    if len(sites) == 1:
        return "matrix" in proc and np.shape(proc["matrix"]) == (2, 2) and _unit_phase_of(np.asarray(proc["matrix"], dtype=C128), singles)
    if len(sites) != 2:
        return False
    if abs(sites[1] - sites[0]) == 1 and "matrix" in proc:
        return np.shape(proc["matrix"]) == (4, 4) and _unit_phase_of(np.asarray(proc["matrix"], dtype=C128), [np.kron(a, b) for a in singles for b in singles])
    if abs(sites[1] - sites[0]) > 1 and "factors" in proc:
        return all(np.shape(f) == (2, 2) and _unit_phase_of(np.asarray(f, dtype=C128), singles) for f in proc["factors"])
    return False


_DISTRIBUTIONS = ("normal", "lognormal", "truncated_normal")
_TWO_SITE_LIBRARY = ("raising_two", "lowering_two")
_log = __import__("logging").getLogger(__name__)


def _is_bool(x) -> bool:
    return isinstance(x, (bool, np.bool_))


def _crosstalk_letters(name):
    m = re.fullmatch(r"(?:longrange_)?crosstalk_([xyz])([xyz])", str(name))
    return (m.group(1), m.group(2)) if m else None


class NoiseModel:
    """``NoiseModel(processes, scheduled_jumps)`` (noise_model.py:227-490): list of {name, sites, strength[, matrix | factors]} and
    of {time, sites, name[, matrix]}.  Construction normalises and validates: sites ascending (operators reordered with them),
    library operators attached as ``matrix`` (one site, adjacent pair) or ``factors`` (distant pair), the documented errors."""

    def __init__(self, processes: Sequence[dict[str, Any]] | None = None, scheduled_jumps: Sequence[dict[str, Any]] | None = None):
        self.processes: list[dict[str, Any]] = []
        self.scheduled_jumps: list[dict[str, Any]] = []
        if scheduled_jumps is not None:
            if not isinstance(scheduled_jumps, (list, tuple)):
                raise TypeError("scheduled_jumps must be a list or tuple of dictionaries.")
            for jump in scheduled_jumps:
                self.scheduled_jumps.append(self._normalize_scheduled_jump(jump))
        if processes is None:
            return
        if not isinstance(processes, (list, tuple)):
            raise TypeError("processes must be a list or tuple of dictionaries.")
        for original in processes:
            self.processes.append(self._normalize_process(original))

    # ---- library ----------------------------------------------------------------------
    @staticmethod
    def get_operator(name: str) -> np.ndarray:
        """An owned copy of a library operator by name (noise_model.py:262-296): one-site names, ``raising_two`` / ``lowering_two``
        and ``[longrange_]crosstalk_ab`` as the Kronecker product of its two Pauli letters."""
        if not isinstance(name, str):
            raise TypeError("Noise operator name must be a string.")
        if name in _LIB_OPS:
            return _LIB_OPS[name].copy()
        if name in _TWO_SITE_LIBRARY:
            one = _LIB_OPS[name[:-4]]
            return np.kron(one, one)
        letters = _crosstalk_letters(name)
        if letters:
            return np.kron(PAULI_MAP[letters[0]], PAULI_MAP[letters[1]])
        raise ValueError(f"Unknown noise operator {name!r}; supported: {sorted(_LIB_OPS) + list(_TWO_SITE_LIBRARY)} and crosstalk_ab with a, b in x, y, z.")

    # ---- validation helpers -------------------------------------------------------------
    @staticmethod
    def _sites_of(entry, what: str) -> tuple[list[int], bool]:
        raw = entry["sites"]
        if not isinstance(raw, (list, tuple)):
            raise TypeError(f"{what} 'sites' must be a list or tuple of integers.")
        if any(_is_bool(q) for q in raw):
            raise TypeError(f"{what} 'sites' must be integers, not booleans.")
        if any(not isinstance(q, (int, np.integer)) for q in raw):
            raise TypeError(f"{what} 'sites' must be a list or tuple of integers.")
        sites = [int(q) for q in raw]
        if len(sites) not in (1, 2):
            raise ValueError(f"{what} must act on exactly 1 or 2 sites.")
        if any(q < 0 for q in sites):
            raise ValueError(f"{what} site indices must be nonnegative.")
        if len(set(sites)) != len(sites):
            raise ValueError(f"{what} sites must be distinct.")
        swapped = len(sites) == 2 and sites[0] > sites[1]
        return sorted(sites), swapped

    @staticmethod
    def _name_of(entry, what: str) -> str:
        name = entry["name"]
        if not isinstance(name, str):
            raise TypeError(f"{what} 'name' must be a string.")
        if not name:
            raise ValueError(f"{what} 'name' must be nonempty.")
        return name

    @staticmethod
    def _matrix_of(value, what: str) -> np.ndarray:
        try:
            m = np.asarray(value, dtype=C128)
        except (TypeError, ValueError) as err:
            raise TypeError(f"{what} must be a numeric array.") from err
        if m.ndim != 2 or m.shape[0] != m.shape[1]:
            raise ValueError(f"{what} must be a square matrix.")
        if not np.all(np.isfinite(m)):
            raise ValueError(f"{what} must have finite entries.")
        return m

    @staticmethod
    def _strength_of(value):
        if isinstance(value, dict):  # static disorder: {"distribution", "mean", "std"}, resolved by sample()
            if "distribution" not in value:
                raise ValueError("Noise strength dict must contain 'distribution' key.")
            unknown = set(value) - {"distribution", "mean", "std"}
            if unknown:
                raise ValueError(f"Unknown distribution keys: {sorted(unknown)}")
            if value["distribution"] not in _DISTRIBUTIONS:
                raise ValueError(f"Unsupported distribution type: {value['distribution']}")
            for key in ("mean", "std"):
                if key not in value:
                    raise ValueError(f"Noise strength distribution needs a '{key}' value.")
                if _is_bool(value[key]) or not np.isfinite(float(value[key])):
                    raise ValueError(f"Noise strength distribution {key} must be a finite number.")
            if float(value["std"]) < 0:
                raise ValueError("Noise strength distribution std must be nonnegative.")
            return {"distribution": value["distribution"], "mean": float(value["mean"]), "std": float(value["std"])}
        if _is_bool(value):
            raise TypeError("Noise strengths must be numbers, not booleans.")
        g = float(value)
        if not np.isfinite(g):
            raise ValueError("Noise strengths must be finite.")
        if g < 0:
            raise ValueError("Noise strengths must be nonnegative.")
        return g

    def _normalize_process(self, original) -> dict[str, Any]:
        if not isinstance(original, dict):
            raise TypeError("Each noise process must be a dictionary.")
        for key in ("name", "sites", "strength"):
            if key not in original:
                raise ValueError(f"Each process must have a '{key}' key.")
        p = dict(original)
        name = self._name_of(p, "Noise process")
        sites, swapped = self._sites_of(p, "Noise process")
        p["strength"] = self._strength_of(p["strength"])
        if "matrix" in p and "factors" in p:
            raise ValueError("A noise process cannot specify both 'matrix' and 'factors'.")
        if "factors" in p and p["factors"] is None:
            raise ValueError("'factors' must be a pair of matrices, not None.")
        if len(sites) == 1:
            if "factors" in p:
                raise ValueError("One-site processes do not accept 'factors'; use 'matrix'.")
            p["matrix"] = self._matrix_of(p["matrix"], "A process 'matrix'") if "matrix" in p else self._library_one_site(name)
        elif sites[1] - sites[0] == 1:
            if "factors" in p:
                raise ValueError("Adjacent two-site processes use 'matrix', not 'factors'.")
            if "matrix" in p:
                if swapped:
                    raise ValueError("Custom full two-site matrices require ascending site order.")
                p["matrix"] = self._matrix_of(p["matrix"], "A process 'matrix'")
            else:
                p["matrix"] = self._library_pair(name, swapped)
        else:
            if "matrix" in p:
                raise ValueError("Non-adjacent two-site processes require 'factors' (one operator per site), not a full 'matrix'.")
            if "factors" in p:
                f = p["factors"]
                if not isinstance(f, (list, tuple)) or len(f) != 2:
                    raise ValueError("'factors' must hold exactly two matrices.")
                pair = (self._matrix_of(f[0], "A process factor"), self._matrix_of(f[1], "A process factor"))
                p["factors"] = (pair[1], pair[0]) if swapped else pair  # the operators follow their sites
            else:
                letters = _crosstalk_letters(name)
                if not letters:
                    raise ValueError(f"Long-range process {name!r} needs 'factors' or a crosstalk_ab name.")
                a, b = (letters[1], letters[0]) if swapped else letters
                p["factors"] = (PAULI_MAP[a].copy(), PAULI_MAP[b].copy())
        p["sites"] = sites
        return p

    @staticmethod
    def _library_one_site(name: str) -> np.ndarray:
        if name not in _LIB_OPS:
            raise ValueError(f"Unknown noise operator {name!r}; supported one-site operators: {sorted(_LIB_OPS)}.")
        return _LIB_OPS[name].copy()

    @staticmethod
    def _library_pair(name: str, swapped: bool) -> np.ndarray:
        letters = _crosstalk_letters(name)
        if letters:
            a, b = (letters[1], letters[0]) if swapped else letters
            return np.kron(PAULI_MAP[a], PAULI_MAP[b])
        if name in _TWO_SITE_LIBRARY:  # noise_library.py:88-106: symmetric under the exchange of the two sites
            one = _LIB_OPS[name[:-4]]
            return np.kron(one, one)
        raise ValueError(f"Unknown noise operator {name!r} for an adjacent pair; use crosstalk_ab, raising_two, lowering_two or 'matrix'.")

    def sample(self, rng=None) -> "NoiseModel":
        """One realisation of static disorder (noise_model.py:492-559): distribution-valued strengths become floats, drawn in
        process order from ``rng`` (``make_disorder_rng``: SeedSequence([seed, 0x4449534F]), random_utils.py:72-87)."""
        import copy

        generator = np.random.default_rng(rng)
        new = object.__new__(NoiseModel)
        new.processes = []
        new.scheduled_jumps = copy.deepcopy(self.scheduled_jumps)
        for proc in self.processes:
            q = copy.deepcopy(proc)
            sv = proc["strength"]
            if isinstance(sv, dict):
                kind, mean, std = sv["distribution"], sv["mean"], sv["std"]
                if kind == "normal":
                    val = float(generator.normal(loc=mean, scale=std))
                    if val < 0.0:
                        _log.warning("Sampled strength %.6g of process %r was negative and clamped to 0.0", val, proc.get("name"))
                        val = 0.0
                elif kind == "lognormal":
                    val = float(generator.lognormal(mean=mean, sigma=std))
                elif kind == "truncated_normal":
                    if abs(std) <= 1e-8:
                        val = float(max(0.0, mean))
                    else:
                        from scipy.stats import truncnorm

                        val = float(truncnorm.rvs((0.0 - mean) / std, np.inf, loc=mean, scale=std, random_state=generator))
                else:
                    raise ValueError(f"Unsupported distribution type: {kind}")
                if not np.isfinite(val) or val < 0:
                    raise ValueError("sampled process strength must be finite and nonnegative.")
                q["strength"] = val
            new.processes.append(q)
        return new

    @property
    def has_disorder(self) -> bool:
        return any(isinstance(q["strength"], dict) for q in self.processes)

    def _normalize_scheduled_jump(self, jump) -> dict[str, Any]:
        """noise_model.py:298-338: {time, sites, name[, matrix]}; two-site jumps act on adjacent sites in ascending order."""
        if not isinstance(jump, dict):
            raise TypeError("Each scheduled jump must be a dictionary.")
        for key in ("time", "sites", "name"):
            if key not in jump:
                raise ValueError(f"Each scheduled jump must have a '{key}' key.")
        if "factors" in jump:
            raise ValueError("Scheduled jumps do not accept 'factors'; use 'matrix' for custom operators.")
        j = dict(jump)
        if _is_bool(j["time"]):
            raise TypeError("Scheduled jump times must be numbers, not booleans.")
        j["time"] = float(j["time"])
        if not np.isfinite(j["time"]):
            raise ValueError("Scheduled jump time must be finite.")
        name = self._name_of(j, "Scheduled jump")
        sites, swapped = self._sites_of(j, "Scheduled jump")
        if len(sites) == 2 and sites[1] - sites[0] != 1:
            raise ValueError(f"Scheduled jump acts on non-adjacent sites {sites}. Only nearest-neighbor scheduled jumps are supported.")
        if "matrix" in j:
            if swapped:
                raise ValueError("Custom full scheduled-jump matrices require ascending site order.")
            j["matrix"] = self._matrix_of(j["matrix"], "A scheduled jump 'matrix'")
        elif len(sites) == 1:
            j["matrix"] = self._library_one_site(name)
        else:
            j["matrix"] = self._library_pair(name, swapped)
        j["sites"] = sites
        return j


def validate_noise_model_for_run(noise_model, *, length: int, physical_dimensions=2, representation: str = "mps", is_digital: bool = False,
                                 is_ensemble: bool = False, sim_params=None) -> None:
    """Run-context checks of a noise model (noise_model.py:668-790): site range, operator shapes, what the digital and the analog
    MPS paths support, and where scheduled jumps are allowed."""
    if noise_model is None:
        return
    dims = [int(physical_dimensions)] * length if np.isscalar(physical_dimensions) else [int(q) for q in physical_dimensions]
    for proc in noise_model.processes:
        sites = proc["sites"]
        if any(q >= length for q in sites):
            raise ValueError(f"Noise process {proc['name']!r} acts on site(s) {sites} out of range for {length} sites.")
        if "matrix" in proc:
            want = int(np.prod([dims[q] for q in sites]))
            if np.shape(proc["matrix"]) != (want, want):
                raise ValueError(f"Noise process {proc['name']!r}: matrix shape {np.shape(proc['matrix'])} does not match ({want}, {want}).")
        if "factors" in proc:
            for q, f in zip(sites, proc["factors"]):
                if np.shape(f) != (dims[q], dims[q]):
                    raise ValueError(f"Noise process {proc['name']!r}: factor on site {q} has shape {np.shape(f)}, expected ({dims[q]}, {dims[q]}).")
        if len(sites) == 2 and sites[1] - sites[0] > 1:
            if is_digital:
                raise ValueError("Digital TJM does not support non-adjacent two-site noise processes.")
            if representation == "mps" and not is_ensemble and not is_pauli(proc):
                raise ValueError("Analog MPS TJM does not support non-Pauli long-range processes (dissipation.py:136-138).")
    if noise_model.scheduled_jumps:
        if is_digital or is_ensemble or representation != "mps":
            raise ValueError("scheduled_jumps are only supported for single-State analog MPS runs.")
        if sim_params is None or not hasattr(sim_params, "times") or not hasattr(sim_params, "order"):
            raise ValueError("AnalogSimParams are required to validate scheduled_jumps.")
        if sim_params.order != 1:
            raise ValueError(f"scheduled_jumps are only supported for AnalogSimParams(order=1); got order={sim_params.order}.")
        for jump in noise_model.scheduled_jumps:
            if any(q >= length for q in jump["sites"]):
                raise ValueError(f"Scheduled jump acts on site(s) {jump['sites']} out of range for {length} sites.")
            if not np.any(np.isclose(sim_params.times, jump["time"], atol=sim_params.dt * 1e-3, rtol=0.0)):
                raise ValueError(f"Scheduled jump time {jump['time']} is not on the simulation time grid.")


# ------------------------------------------------------------------ states / operators
class MPS:
    """Tensor list with index order (sigma, chi_left, chi_right) (mps.py:58)."""

    def __init__(self, length: int, tensors: list[np.ndarray] | None = None, physical_dimensions: list[int] | int | None = None,
                 state: str = "zeros", pad: int | None = None, basis_string: str | None = None, rng: np.random.Generator | None = None):
        """Argument order of the reference (mps.py:71-79); ``rng`` (seeded "random" / "haar-random" states) is an addition.

        ``physical_dimensions``: one local dimension for every site (int or a uniform list), 2, 3 or 4 - the engine's storage is
        ``[B][d][cap][cap]`` with one d per chain.  The product presets fill a length-d vector exactly as mps.py:224-300 does
        (the qubit amplitudes in the first two levels)."""
        self.length = length
        if physical_dimensions is None:
            dims = [2] * length
        elif isinstance(physical_dimensions, (int, np.integer)):
            dims = [int(physical_dimensions)] * length
        else:
            dims = [int(q) for q in physical_dimensions]
        assert len(dims) == length
        if tensors is not None:
            assert len(tensors) == length
            self.tensors = [np.asarray(t, dtype=C128) for t in tensors]
            if physical_dimensions is None:
                dims = [int(t.shape[0]) for t in self.tensors]
        if length and not all(2 <= q <= 4 for q in dims):
            raise NotImplementedError("the HIP path holds local dimensions 2, 3 and 4 (larger ones are not built)")
        self.physical_dimensions = dims
        if tensors is not None:
            return
        s = 1 / np.sqrt(2)
        self.tensors = []
        if state == "haar-random":
            chi = 1 if pad is None else pad
            caps = self.bond_caps(length, chi, dims)
            rng = rng if rng is not None else np.random.default_rng()
            for i in range(length):
                d = dims[i]
                cl, cr = caps[i], caps[i + 1]
                x = rng.standard_normal((d * cl, cr)) + 1j * rng.standard_normal((d * cl, cr))
                q, r = np.linalg.qr(x, mode="reduced")
                dg = np.diag(r)
                ph = np.ones_like(dg, dtype=C128)
                nz = np.abs(dg) > 0
                ph[nz] = dg[nz] / np.abs(dg[nz])
                self.tensors.append((q / ph[np.newaxis, :]).reshape(d, cl, cr).astype(C128))
            return
        for i in range(length):
            d = dims[i]
            v = np.zeros(d, dtype=C128)
            if state == "zeros":
                v[0] = 1
            elif state == "ones":
                v[1] = 1
            elif state == "x+":
                v[:2] = (s, s)
            elif state == "x-":
                v[:2] = (s, -s)
            elif state == "y+":
                v[:2] = (s, 1j * s)
            elif state == "y-":
                v[:2] = (s, -1j * s)
            elif state == "Neel":
                v[0 if i % 2 else 1] = 1
            elif state == "wall":
                v[0 if i < length // 2 else 1] = 1
            elif state == "random":  # (r, 1 - r) per site, normalised afterwards (mps.py:266-269, 294-295)
                r = (rng if rng is not None else np.random.default_rng()).random()
                v[:2] = (r, 1 - r)
                v /= np.linalg.norm(v)
            elif state == "basis":   # one character per site, site 0 first (mps.py:395-408); digits up to d - 1 for qudits
                if basis_string is None or len(basis_string) != length or not (basis_string[i].isdigit() and int(basis_string[i]) < d):
                    raise ValueError("state='basis' needs basis_string of one digit below the local dimension per site")
                v[int(basis_string[i])] = 1
            else:
                raise ValueError("Invalid state string")
            self.tensors.append(v.reshape(d, 1, 1))
        if pad is not None:
            self.pad_bond_dimension(pad)

    @property
    def mps(self) -> "MPS":
        """``result.output_state.mps`` of the reference (state.py): the MPS behind a State - here the object itself."""
        return self

    def to_vec(self) -> np.ndarray:
        """Dense state vector with site 0 as the fastest (least significant) index, the reference's convention (mps.py:1633-1658:
        the network is flipped before the contraction); small chains only."""
        if self.length > 20:
            raise ValueError("to_vec is meant for small chains")
        v = np.ones((1, 1), dtype=C128)  # (physical indices of the sites so far, right bond)
        for t in self.tensors:  # (sigma, l, r): the new site becomes the MOST significant index so far
            v = np.einsum("xl,slr->sxr", v, t).reshape(-1, t.shape[2])
        return v.reshape(-1)

    # -- inspection helpers of the reference's MPS (host side, numpy over the site tensors) -----------------------------------------
    def bond_dimensions(self) -> list[int]:
        """Internal bond dimensions, left to right (mps.py:514-520)."""
        return [int(t.shape[2]) for t in self.tensors[:-1]]

    def get_max_bond(self) -> int:
        """max over sites of max(shape[0], shape[2]) exactly as mps.py:549-565 writes it (the physical dimension takes part, so a
        product state reports 2)."""
        return max(max(int(t.shape[0]), int(t.shape[2])) for t in self.tensors)

    def get_total_bond(self) -> int:
        """Sum of the left bonds of sites 1 .. L-1 (mps.py:567-578)."""
        return sum(int(t.shape[1]) for t in self.tensors[1:])

    def get_cost(self) -> int:
        """Sum of the cubed left bonds of sites 1 .. L-1 (mps.py:580-591)."""
        return sum(int(t.shape[1]) ** 3 for t in self.tensors[1:])

    def _bond_singular_values(self, sites, what: str):
        assert len(sites) == 2, f"{what} is defined on a bond (two adjacent sites)."
        i, j = sites
        assert i + 1 == j, f"{what} is only defined for nearest-neighbor cut."
        a, b = self.tensors[i], self.tensors[j]
        if a.shape[2] == 1:
            return None
        theta = np.tensordot(a, b, axes=(2, 1)).reshape(a.shape[0] * a.shape[1], b.shape[0] * b.shape[2])
        return np.linalg.svd(theta, compute_uv=False)

    def get_entropy(self, sites) -> np.float64:
        """Von Neumann entropy (natural log) of the normalised squared singular values of the two-site block (i, i+1) as it stands
        (mps.py:604-642): the entanglement entropy of the cut when the orthogonality centre is on one of the two sites."""
        sv = self._bond_singular_values(sites, "Entropy")
        if sv is None or not np.sum(sv ** 2) > 0:
            return np.float64(0.0)
        pr = sv ** 2 / np.sum(sv ** 2)
        return np.float64(-np.sum(pr * np.log(pr + np.finfo(np.float64).tiny)))

    def get_schmidt_spectrum(self, sites) -> np.ndarray:
        """Singular values of the two-site block (i, i+1), NaN-padded to 500 entries (mps.py:644-678)."""
        sv = self._bond_singular_values(sites, "Schmidt spectrum")
        out = np.full(500, np.nan)
        if sv is None:
            out[0] = 1.0
        else:
            out[:min(500, len(sv))] = sv[:500]
        return out

    def scalar_product(self, other: "MPS", sites: int | None = None) -> np.complex128:
        """<self|other> over the whole chain, or the local contraction of one site's tensors (mps.py:901-959)."""
        if sites is not None:
            i = int(sites[0]) if isinstance(sites, (list, tuple)) else int(sites)
            if isinstance(sites, (list, tuple)) and len(sites) != 1:
                raise ValueError("Invalid `sites` argument.")
            return np.complex128(np.vdot(self.tensors[i], other.tensors[i]))
        env = np.ones((1, 1), dtype=C128)  # (bra bond, ket bond)
        for a, b in zip(self.tensors, other.tensors):
            env = np.einsum("xy,pxa,pyb->ab", env, a.conj(), b)
        return np.complex128(env[0, 0])

    def norm(self, site: int | None = None) -> np.float64:
        """<psi|psi>, or the squared Frobenius norm of one site tensor (the norm when the centre sits there) (mps.py:1539-1565)."""
        return np.float64(np.real(self.scalar_product(self, site)))

    def check_if_valid_mps(self) -> None:
        """Neighbouring bonds must agree (mps.py:1567-1579)."""
        for a, b in zip(self.tensors, self.tensors[1:]):
            assert a.shape[2] == b.shape[1]

    def check_canonical_form(self) -> list[int]:
        """Every site that can serve as orthogonality centre: all sites left of it left-orthonormal, all sites right of it
        right-orthonormal - what the code of mps.py:1598-1630 returns (several sites for e.g. a normalised product state, an empty list
        when there is none)."""
        lefts = [np.allclose(np.einsum("pxa,pxb->ab", t.conj(), t), np.eye(t.shape[2])) for t in self.tensors]
        rights = [np.allclose(np.einsum("pax,pbx->ab", t.conj(), t), np.eye(t.shape[1])) for t in self.tensors]
        return [i for i in range(self.length) if all(lefts[:i]) and all(rights[i + 1:])]

    _MEASUREMENT_ROTATION = {"Z": np.eye(2, dtype=C128), "X": np.array([[1, 1], [1, -1]], dtype=C128) / np.sqrt(2),
                             "Y": np.array([[1, -1j], [1, 1j]], dtype=C128) / np.sqrt(2)}

    def _outcome_probabilities(self, basis: str) -> np.ndarray:
        basis = str(basis).upper()
        if basis not in self._MEASUREMENT_ROTATION:
            raise ValueError(f"Invalid basis: {basis}. Expected 'X', 'Y', or 'Z'.")  # mps.py:1306-1314
        if self.length > 24:
            raise ValueError("host-side sampling enumerates the 2^L outcomes; use DigitalSimParams(shots=...) for long chains")
        rot = self._MEASUREMENT_ROTATION[basis]
        pr = np.abs(MPS(self.length, tensors=[np.einsum("ab,bcd->acd", rot, t) for t in self.tensors]).to_vec()) ** 2
        return pr / pr.sum()

    def measure_single_shot(self, basis: str = "Z", rng: np.random.Generator | None = None) -> int:
        """One projective measurement of every site in the X, Y or Z basis; the outcome is sum(bit_i << i), site 0 the least
        significant bit (mps.py:1282-1345).  Host-side helper for small chains: the outcome is drawn from the exact distribution
        of the 2^L results instead of site by site, so a seeded generator gives the same statistics but not the same draws as the
        reference; ensembles of shots on the GPU are ``DigitalSimParams(shots=...)`` (``tjm_engine_sample_shots``)."""
        rng = rng if rng is not None else np.random.default_rng()
        pr = self._outcome_probabilities(basis)
        return int(rng.choice(pr.size, p=pr))

    def measure_shots(self, shots: int, basis: str = "Z", rng: np.random.Generator | None = None) -> dict[int, int]:
        """Histogram {outcome: count} of ``shots`` independent measurements (mps.py:1351-1382)."""
        rng = rng if rng is not None else np.random.default_rng()
        pr = self._outcome_probabilities(basis)
        draws, counts = np.unique(rng.choice(pr.size, size=int(shots), p=pr), return_counts=True)
        return {int(k): int(v) for k, v in zip(draws, counts)}

    def expect(self, observable) -> float:
        """<psi| O |psi> of a one-site or nearest-neighbour two-site observable (mps.py:961-1047), dense evaluation for small chains;
        a two-site matrix is indexed (s_i, s_{i+1}) with site i the major index."""
        L = self.length
        psi = self.to_vec().reshape([2] * L)  # axes (s_{L-1}, ..., s_0)
        sites = observable.sites if isinstance(observable.sites, (list, tuple)) else [observable.sites]
        m = np.asarray(observable.gate.matrix, dtype=C128)
        axes = [L - 1 - int(q) for q in sites]
        k = len(axes)
        out = np.tensordot(m.reshape([2] * (2 * k)), psi, axes=(list(range(k, 2 * k)), axes))
        out = np.moveaxis(out, list(range(k)), axes)
        return float(np.real(np.vdot(psi, out)))

    def pad_bond_dimension(self, target_dim: int) -> None:
        """Zero-pad every internal bond k to ``min(target_dim, 2**min(k, L-k))`` and re-canonicalise (mps.py:409-452): the start
        of fixed-chi runs such as one-site TDVP, whose tangent space is spanned by the padded isometries."""
        caps = self.bond_caps(self.length, target_dim, self.physical_dimensions)
        for i, t in enumerate(self.tensors):
            d, cl, cr = t.shape
            if cl > caps[i] or cr > caps[i + 1]:
                raise ValueError("Target bond dim must be at least current bond dim.")
            new = np.zeros((d, caps[i], caps[i + 1]), dtype=C128)
            new[:, :cl, :cr] = t
            self.tensors[i] = new
        self.normalize("B")

    def normalize(self, form: str = "B") -> None:
        """Right-canonical form with centre 0 and norm 1 (``MPS.normalize("B")``, mps.py:815-839).

        One-off host preparation of the input state (the reference does this in
        ``State.ensure_encoded``, state.py:278-297, above the trajectory path).
        """
        if form != "B":
            raise ValueError("only form 'B' is prepared on the host")
        t = self.tensors
        for i in range(self.length - 1, 0, -1):
            d, cl, cr = t[i].shape
            m = t[i].transpose(1, 0, 2).reshape(cl, d * cr)          # (chi_l, d*chi_r)
            q, r = np.linalg.qr(m.conj().T)                           # m^H = q r  ->  m = r^H q^H
            k = q.shape[1]
            t[i] = q.conj().T.reshape(k, d, cr).transpose(1, 0, 2)
            t[i - 1] = np.einsum("sab,bk->sak", t[i - 1], r.conj().T)
        t[0] = t[0] / np.linalg.norm(t[0])

    @staticmethod
    def bond_caps(length: int, target: int, d=2) -> list[int]:
        """Feasible bond dimensions for a target maximum (mps.py:130-168); ``d``: one local dimension or one per site."""
        dims = [int(d)] * length if np.isscalar(d) else [int(q) for q in d]
        caps = [1] * (length + 1)
        left = 1
        for i in range(1, length):
            left *= dims[i - 1]
            caps[i] = left
        right = 1
        for i in range(length - 1, 0, -1):
            right *= dims[i]
            caps[i] = min(caps[i], right, target)
        return caps


class MPO:
    """MPO tensors with index order (phys_out, phys_in, chi_left, chi_right) (mpo.py:45-50)."""

    def __init__(self, tensors: list[np.ndarray] | None = None):
        self.tensors = [] if tensors is None else [np.asarray(t, dtype=C128) for t in tensors]

    @property
    def length(self) -> int:
        return len(self.tensors)

    @classmethod
    def identity(cls, length: int, physical_dimension: int = 2) -> "MPO":
        """Identity operator with bond dimension 1 (mpo.py:1015-1028)."""
        d = int(physical_dimension)
        return cls([np.eye(d, dtype=C128).reshape(d, d, 1, 1).copy() for _ in range(length)])

    def custom(self, tensors, *, transpose: bool = True) -> None:
        """Adopt site tensors; ``transpose=True`` takes them as (chi_left, chi_right, phys_out, phys_in) (mpo.py:1146-1169)."""
        ts = [np.asarray(t, dtype=C128) for t in tensors]
        if transpose:
            ts = [t.transpose(2, 3, 0, 1) for t in ts]
        ok = all(t.ndim == 4 for t in ts) and ts[0].shape[2] == 1 and ts[-1].shape[3] == 1
        assert ok and all(a.shape[3] == b.shape[2] for a, b in zip(ts, ts[1:])), "MPO initialized wrong"
        self.tensors = ts

    def to_matrix(self) -> np.ndarray:
        """Dense operator with site 0 the MOST significant index (mpo.py:1755-1781)."""
        acc = np.ones((1, 1, 1), dtype=C128)
        for t in self.tensors:
            acc = np.einsum("oib,pqbc->opiqc", acc, t).reshape(acc.shape[0] * t.shape[0], acc.shape[1] * t.shape[1], t.shape[3])
        return acc[:, :, 0]

    def to_matrix_mps_order(self) -> np.ndarray:
        """Dense operator acting on ``MPS.to_vec`` vectors: site 0 the LEAST significant index (mpo.py:1783-1794)."""
        acc = np.ones((1, 1, 1), dtype=C128)
        for t in self.tensors:
            acc = np.einsum("pqbc,oib->poqic", t, acc).reshape(acc.shape[0] * t.shape[0], acc.shape[1] * t.shape[1], t.shape[3])
        return acc[:, :, 0]

    def from_pauli_sum(self, *, terms, length: int, physical_dimension: int = 2, tol: float = 1e-12, max_bond_dim: int | None = None,
                       n_sweeps: int = 2) -> None:
        """H = sum_k c_k P_k from ``(coeff, "X0 Z3 ...")`` terms, ``""`` being the identity (mpo.py:1171-1318).  The reference assembles a
        finite-state machine and compresses it; here every term is one channel of a direct sum that is compressed exactly while it is
        built: a left-to-right sweep of SVDs over (left basis x site, channels), then one right-to-left SVD sweep on the left-canonical
        result, both cut at ``tol`` relative to the largest singular value (and at ``max_bond_dim`` when given).  The operator is the
        same; the bond dimensions are the operator Schmidt ranks.  ``n_sweeps`` is accepted for signature parity."""
        if length <= 0:
            raise ValueError("L must be positive.")
        if physical_dimension != 2:
            raise ValueError("Only physical_dimension=2 is supported.")
        ops = {"I": _I, "X": _X, "Y": _Y, "Z": _Z}
        coef, table = [], []
        for c, spec in terms:
            row = {}
            for tok in str(spec).split():
                label, idx = tok[:1].upper(), tok[1:]
                if label not in ops or not idx.isdigit():
                    raise ValueError(f"Invalid term {spec!r}: expected tokens like 'X0' with an operator in {sorted(ops)}.")
                site = int(idx)
                if site >= length:
                    raise ValueError(f"Site index {site} out of bounds for length {length}.")
                if site in row:
                    raise ValueError(f"Invalid term {spec!r}: site {site} appears twice.")
                row[site] = label
            coef.append(complex(c))
            table.append(row)
        n = len(coef)
        if n == 0:
            self.tensors = [np.zeros((2, 2, 1, 1), dtype=C128) for _ in range(length)]
            return

        def rank(sv):
            k = int(np.sum(sv > tol * sv[0])) if sv.size and sv[0] > 0 else 0
            return max(1, k if max_bond_dim is None else min(k, int(max_bond_dim)))

        carry = np.asarray(coef, dtype=C128).reshape(1, n)  # (left basis, channel)
        left = []
        for i in range(length):
            site_ops = np.stack([ops[row.get(i, "I")] for row in table])  # (channel, out, in)
            m = np.einsum("kt,tpq->kpqt", carry, site_ops)
            k = carry.shape[0]
            if i == length - 1:
                left.append(m.sum(axis=3).reshape(k, 2, 2, 1))
                break
            u, sv, vh = np.linalg.svd(m.reshape(k * 4, n), full_matrices=False)
            r = rank(sv)
            left.append(u[:, :r].reshape(k, 2, 2, r))
            carry = sv[:r, None] * vh[:r]
        for i in range(length - 1, 0, -1):  # (l, out, in, r) tensors, left-canonical up to site i
            l, r_ = left[i].shape[0], left[i].shape[3]
            u, sv, vh = np.linalg.svd(left[i].reshape(l, 4 * r_), full_matrices=False)
            r = rank(sv)
            left[i] = vh[:r].reshape(r, 2, 2, r_)
            left[i - 1] = np.einsum("kpqm,mr->kpqr", left[i - 1], u[:, :r] * sv[:r])
        self.tensors = [t.transpose(1, 2, 0, 3).copy() for t in left]

    @classmethod
    def long_range_ising(cls, length: int, coeffs, decays, g: float) -> "MPO":
        """H = -sum_{i<j} f(j-i) Z_i Z_j - g sum_i X_i with f(r) = sum_k coeffs[k] * decays[k]**(r-1): the exponential-sum
        form of a power-law coupling as a finite-state-machine MPO of bond dimension K + 2."""
        K = len(coeffs)
        D = K + 2
        w = np.zeros((D, D, 2, 2), dtype=C128)
        w[0, 0] = _I
        w[D - 1, D - 1] = _I
        w[0, D - 1] = -g * _X
        for k in range(K):
            w[0, 1 + k] = -coeffs[k] * _Z
            w[1 + k, 1 + k] = decays[k] * _I
            w[1 + k, D - 1] = _Z
        return cls._fsm(length, w)

    @classmethod
    def bose_hubbard(cls, length: int, local_dim: int, omega: float, hopping_j: float, hubbard_u: float) -> "MPO":
        """H = sum_i [omega n_i + U/2 n_i (n_i - 1)] - J sum_i (a_i^dag a_{i+1} + h.c.) on sites with ``local_dim`` levels
        (at most local_dim - 1 bosons per site), the bond-dimension-4 automaton of mpo.py:670-745: state 1 has placed a^dag and
        waits for -J a, state 2 has placed a and waits for -J a^dag."""
        if length <= 0:
            raise ValueError("length must be positive.")
        d = int(local_dim)
        a = np.diag(np.sqrt(np.arange(1, d, dtype=np.float64)), 1).astype(C128)
        ad = a.conj().T
        n = ad @ a
        one = np.eye(d, dtype=C128)
        w = np.zeros((4, 4, d, d), dtype=C128)
        w[0, 0], w[3, 3] = one, one
        w[0, 1], w[1, 3] = ad, -hopping_j * a
        w[0, 2], w[2, 3] = a, -hopping_j * ad
        w[0, 3] = omega * n + 0.5 * hubbard_u * (n @ (n - one))
        return cls._fsm(length, w)

    @classmethod
    def fermi_hubbard_1d(cls, length: int, t: float, u: float) -> "MPO":
        """H = U sum_i n_{i,up} n_{i,down} - t sum_{i,s} (c^dag_{i,s} c_{i+1,s} + h.c.) on composite four-level sites |n_up n_down>, written
        with the site-local ladder operators c_up = c (x) 1, c_down = 1 (x) c (the fermionic embedding of mpo.py:472-520, not the
        Jordan-Wigner chain): the bond-dimension-6 automaton whose states 1-4 have placed c_up^dag, c_down^dag, c_up, c_down and wait for
        -t times the partner operator on the next site."""
        if length <= 0:
            raise ValueError("length must be positive.")
        c = np.array([[0, 1], [0, 0]], dtype=C128)
        one2 = np.eye(2, dtype=C128)
        c_up, c_dn = np.kron(c, one2), np.kron(one2, c)
        n_up, n_dn = np.kron(c.conj().T @ c, one2), np.kron(one2, c.conj().T @ c)
        one = np.eye(4, dtype=C128)
        w = np.zeros((6, 6, 4, 4), dtype=C128)
        w[0, 0], w[5, 5], w[0, 5] = one, one, u * (n_up @ n_dn)
        for k, (first, partner) in enumerate(((c_up.conj().T, c_up), (c_dn.conj().T, c_dn), (c_up, c_up.conj().T), (c_dn, c_dn.conj().T))):
            w[0, 1 + k] = first
            w[1 + k, 5] = -t * partner
        return cls._fsm(length, w)

    @classmethod
    def coupled_transmon(cls, length: int, qubit_dim: int, resonator_dim: int, qubit_freq: float, resonator_freq: float, anharmonicity: float,
                         coupling: float) -> "MPO":
        """Chain of transmons (even sites, ``qubit_dim`` levels, Duffing oscillators w n + alpha/2 n (n - 1)) and resonators (odd sites,
        ``resonator_dim`` levels, w n), neighbours coupled by g (b + b^dag)(a + a^dag): the bond-dimension-4 MPO of mpo.py:549-668, whose
        sites differ in dimension.  Channel 0 / 3 carry "nothing placed yet" / "term complete" between a qubit and the next; a resonator
        hands the channels on crosswise (its own term sits between two qubits)."""
        def ladder(d_):
            return np.diag(np.sqrt(np.arange(1, d_, dtype=np.float64)), 1).astype(C128)

        b, a = ladder(int(qubit_dim)), ladder(int(resonator_dim))
        one_q, one_r = np.eye(b.shape[0], dtype=C128), np.eye(a.shape[0], dtype=C128)
        n_q, n_r = b.conj().T @ b, a.conj().T @ a
        h_q = qubit_freq * n_q + 0.5 * anharmonicity * (n_q @ (n_q - one_q))
        h_r = resonator_freq * n_r
        x_q, x_r = b + b.conj().T, a + a.conj().T
        tensors = []
        for i in range(length):
            if i % 2 == 0:
                w = np.zeros((4, 4) + one_q.shape, dtype=C128)
                w[0, 0], w[0, 1], w[0, 2], w[0, 3] = h_q, one_q, coupling * x_q, one_q
                w[1, 3], w[3, 3] = coupling * x_q, one_q
                if i == 0:
                    w = w[0:1]
                elif i == length - 1:
                    w = np.stack([one_q, coupling * x_q, one_q, h_q])[:, None]
            else:
                w = np.zeros((4, 4) + one_r.shape, dtype=C128)
                w[0, 0], w[1, 2], w[2, 0], w[3, 1], w[3, 3] = one_r, h_r, x_r, x_r, one_r
            tensors.append(np.ascontiguousarray(w.transpose(2, 3, 0, 1)))
        return cls(tensors)

    @staticmethod
    def _check_bc(bc: str, length: int) -> bool:
        if bc not in ("open", "periodic"):
            raise ValueError("bc must be 'open' or 'periodic'.")  # mpo.py:288-290
        if bc == "periodic" and length < 2:
            raise ValueError("periodic boundary conditions need at least two sites")
        return bc == "periodic"

    @classmethod
    def _fsm(cls, length: int, w: np.ndarray, wrap=None) -> "MPO":
        """Site tensors of the automaton ``w[from, to]`` (state 0 = nothing placed yet, state D-1 = term complete).  ``wrap`` lists the
        closing bonds (A, B) of a periodic chain, the terms A_{L-1} B_0 (mpo.py:304-308): each gets one more state that site 0 enters
        with B, the bulk carries with the identity and the last site leaves with A."""
        if wrap:
            D0, n = w.shape[0], len(wrap)
            big = np.zeros((D0 + n, D0 + n, 2, 2), dtype=C128)
            big[:D0 - 1, :D0 - 1] = w[:D0 - 1, :D0 - 1]
            big[:D0 - 1, -1] = w[:D0 - 1, -1]
            big[-1, -1] = w[-1, -1]
            first, last = big.copy(), big.copy()
            for k, (a_op, b_op) in enumerate(wrap):
                big[D0 - 1 + k, D0 - 1 + k] = _I
                first[0, D0 - 1 + k] = b_op
                last[D0 - 1 + k, -1] = a_op
            D = D0 + n
            t = [first.transpose(2, 3, 0, 1)[:, :, 0:1, :].copy()]
            t += [big.transpose(2, 3, 0, 1).copy() for _ in range(length - 2)]
            t.append(last.transpose(2, 3, 0, 1)[:, :, :, D - 1:D].copy())
            return cls(t)
        D = w.shape[0]
        bulk = w.transpose(2, 3, 0, 1)
        t = []
        for i in range(length):
            if length == 1:
                t.append(bulk[:, :, 0:1, D - 1:D].copy())
            elif i == 0:
                t.append(bulk[:, :, 0:1, :].copy())
            elif i == length - 1:
                t.append(bulk[:, :, :, D - 1:D].copy())
            else:
                t.append(bulk.copy())
        return cls(t)

    @classmethod
    def pauli(cls, *, length: int, two_body=None, one_body=None, bc: str = "open", **_unused) -> "MPO":
        """H = sum_bonds c A_i B_{i+1} + sum_i c A_i from ``(coeff, op_i, op_j)`` / ``(coeff, op)`` terms (mpo.py:247-325), as an exact
        finite-state-machine MPO of bond dimension 2 + len(two_body) (the reference compresses a Pauli sum; the operator is the same)."""
        periodic = cls._check_bc(bc, length)
        ops = {"I": _I, "X": _X, "Y": _Y, "Z": _Z}

        def op(x):
            x = str(x).upper()
            if x not in ops:
                raise ValueError(f"Invalid operator {x!r}; expected one of {sorted(ops)}.")
            return ops[x]

        two_body, one_body = list(two_body or []), list(one_body or [])
        D = 2 + len(two_body)
        w = np.zeros((D, D, 2, 2), dtype=C128)
        w[0, 0] = _I
        w[D - 1, D - 1] = _I
        for k, (c, a, b) in enumerate(two_body):
            w[0, 1 + k] = c * op(a)
            w[1 + k, D - 1] = op(b)
        for c, a in one_body:
            w[0, D - 1] = w[0, D - 1] + c * op(a)
        return cls._fsm(length, w, [(c * op(a), op(b)) for c, a, b in two_body] if periodic else None)

    @classmethod
    def ising(cls, length: int, J: float, g: float, bc: str = "open") -> "MPO":
        """H = -J sum Z_i Z_{i+1} - g sum X_i (sign convention of mpo.py:326-363); ``bc="periodic"`` adds the bond (L-1, 0)."""
        periodic = cls._check_bc(bc, length)
        w = np.zeros((3, 3, 2, 2), dtype=C128)
        w[0, 0] = _I
        w[0, 1] = -J * _Z
        w[0, 2] = -g * _X
        w[1, 2] = _Z
        w[2, 2] = _I
        return cls._fsm(length, w, [(-J * _Z, _Z)] if periodic else None)

    @classmethod
    def heisenberg(cls, length: int, Jx: float, Jy: float, Jz: float, h: float = 0.0, bc: str = "open") -> "MPO":
        """H = -sum (Jx XX + Jy YY + Jz ZZ) - h sum Z (mpo.py:365-406); ``bc="periodic"`` adds the bond (L-1, 0)."""
        periodic = cls._check_bc(bc, length)
        w = np.zeros((5, 5, 2, 2), dtype=C128)
        w[0, 0] = _I
        w[0, 1] = -Jx * _X
        w[0, 2] = -Jy * _Y
        w[0, 3] = -Jz * _Z
        w[0, 4] = -h * _Z
        w[1, 4] = _X
        w[2, 4] = _Y
        w[3, 4] = _Z
        w[4, 4] = _I
        return cls._fsm(length, w, [(-Jx * _X, _X), (-Jy * _Y, _Y), (-Jz * _Z, _Z)] if periodic else None)


# ------------------------------------------------------------------ result
class Result:
    """Outcome of a run (result.py:34-189): ``sim_params`` (the caller's object, untouched), copies of the observables in the user's
    order, per-trajectory rows and their means, averaged diagnostics, the final state / sampled noise model / measurement histogram
    when the run produced them, ``None`` otherwise."""

    def __init__(self, sim_params, results_sorted, diagnostics, counts=None, schmidt=None):
        import copy

        # results_sorted: [num_traj, n_obs_sorted, T]; diagnostics: [num_traj, 3, T]
        self.sim_params = sim_params
        self.observables = [copy.copy(ob) for ob in sim_params.observables]
        self.output_state = None
        self.noise_model = None
        self.counts = counts
        self.multi_time_times = None    # correlator outputs of the reference's other solvers
        self.multi_time_results = None
        self.times = None
        if hasattr(sim_params, "times"):
            self.times = sim_params.times if sim_params.sample_timesteps else sim_params.times[-1:]
        self.trajectories, self.expectation_values = [], []
        if results_sorted is not None and len(self.observables):
            idx = sim_params.observable_sorted_indices
            self.trajectories = [results_sorted[:, idx[u], :] for u in range(len(self.observables))]
            self.expectation_values = [np.mean(t, axis=0) for t in self.trajectories]
            for u, ob in enumerate(self.observables):
                if ob.gate.name != "schmidt_spectrum":
                    continue
                # the reference keeps the 500-entry vectors in the results buffer and concatenates them over the trajectories
                # (mps.py:1211, result.py:127-139): trajectories[u] is [num_traj, T, 500], NaN-padded past the spectrum
                n_traj, cols = results_sorted.shape[0], results_sorted.shape[2]
                spec = np.full((n_traj, cols, 500), np.nan)
                for (t, row, col), vec in (schmidt or {}).items():
                    if row == idx[u]:
                        spec[t, col] = vec
                self.trajectories[u] = spec
                self.expectation_values[u] = np.concatenate([spec[t].ravel() for t in range(n_traj)]) if n_traj else np.zeros(0)
        self.runtime_cost = self.max_bond = self.total_bond = None
        self.trajectory_diagnostics = diagnostics  # per-trajectory rows (this package's addition), also for shots-only runs
        if diagnostics is not None and (len(self.observables) or counts is None):  # a shots-only run reports no averaged diagnostics
            d = np.mean(diagnostics, axis=0)
            self.runtime_cost, self.max_bond, self.total_bond = d[0], d[1], d[2]


CircuitResult = Result  # circuit runs return the same object, with ``counts`` when ``shots`` is set (result.py:155-189)


# ------------------------------------------------------------------ front-end wrappers of the reference's newer API
class State(MPS):
    """``State(length, initial="zeros", pad=..., tensors=...)`` (core/data_structures/state.py:50-140), MPS representation only:
    the TJM path works on matrix product states, the ``vector`` / ``density_matrix`` representations belong to the reference's
    other solvers."""

    def __init__(self, length: int | None = None, *, initial: str = "zeros", representation: str | None = None,
                 physical_dimensions=None, tensors=None, vector=None, density_matrix=None, pad: int | None = None,
                 basis_string: str | None = None, seed: int | None = None):
        if length is not None and length <= 0:
            raise ValueError("length must be a positive integer.")  # state.py:88-90
        if vector is not None or density_matrix is not None or representation not in (None, "mps"):
            raise NotImplementedError("only representation='mps' is part of the TJM path built here")
        if tensors is not None and (basis_string is not None or pad is not None or seed is not None or initial != "zeros"):
            raise ValueError("initial / pad / basis_string / seed describe a preset state; omit them with tensors=")  # state_utils.py:39-76
        if tensors is not None:
            if len(tensors) == 0:
                raise ValueError("tensors must be a non-empty list of MPS cores.")
            if length is not None and length != len(tensors):
                raise ValueError(f"length={length} does not match len(tensors)={len(tensors)}.")
            super().__init__(len(tensors), tensors=list(tensors), physical_dimensions=physical_dimensions)
        else:
            if length is None:
                raise ValueError("length is required for a preset state.")
            rng = np.random.default_rng(seed) if seed is not None else None
            super().__init__(length, physical_dimensions=physical_dimensions, state=initial, pad=pad, rng=rng, basis_string=basis_string)
        self.initial, self.representation, self.basis_string = initial, "mps", basis_string


class Hamiltonian(MPO):
    """``Hamiltonian.ising(...)`` / ``.heisenberg(...)`` / ``.from_mpo(...)`` / ``.piecewise(...)``
    (core/data_structures/hamiltonian.py:36-330); the factories are those of ``MPO``."""

    def __init__(self, tensors=None, *, matrix=None):
        """``Hamiltonian(matrix=H)`` (hamiltonian.py:49-120): a dense operator on L qubits in the convention of ``to_vec`` (site 0 the
        least significant index), as an exact MPO by successive SVDs (small chains)."""
        if matrix is None:
            super().__init__(tensors)
            return
        H = np.asarray(matrix, dtype=C128)
        L = int(round(np.log2(H.shape[0])))
        if H.ndim != 2 or H.shape[0] != H.shape[1] or 2 ** L != H.shape[0] or L < 1 or L > 10:
            raise ValueError("matrix must be a square operator on 1 to 10 qubits")
        # T[(o_0 i_0), (o_1 i_1), ...] then split site by site
        T = H.reshape([2] * (2 * L)).transpose([k for s_ in range(L) for k in (L - 1 - s_, 2 * L - 1 - s_)]).reshape([4] * L)
        out, left = [], 1
        rest = T.reshape(left * 4, -1)
        for s_ in range(L - 1):
            u, sv, vh = np.linalg.svd(rest, full_matrices=False)
            keep = max(1, int(np.sum(sv > 1e-14 * max(sv[0], 1e-300))))
            out.append(u[:, :keep].reshape(left, 2, 2, keep).transpose(1, 2, 0, 3))
            rest = (sv[:keep, None] * vh[:keep]).reshape(keep * 4, -1)
            left = keep
        out.append(rest.reshape(left, 2, 2, 1).transpose(1, 2, 0, 3))
        super().__init__(out)

    @classmethod
    def from_mpo(cls, mpo: MPO) -> "Hamiltonian":
        out = cls.__new__(cls)
        out.__dict__.update(mpo.__dict__)
        return out

    @staticmethod
    def piecewise(pieces):
        """``[(Hamiltonian, duration), ...]`` -> the tuple of per-interval MPOs ``Simulator.run`` takes (hamiltonian.py:179-230); the
        durations are checked against the time grid there."""
        out = []
        for ham, duration in pieces:
            if not duration > 0:
                raise ValueError("piece durations must be positive")
            out.append((ham, float(duration)))
        return PiecewiseHamiltonian(out)


class PiecewiseHamiltonian:
    def __init__(self, pieces):
        self.pieces = pieces
        self.length = pieces[0][0].length

    def per_interval(self, dt: float, n_intervals: int):
        """One MPO per interval of the dt grid; durations must be integer multiples of dt and add up to the run."""
        out = []
        for ham, duration in self.pieces:
            k = duration / dt
            if abs(k - round(k)) > 1e-9 or round(k) < 1:
                raise ValueError("every piece duration must be a positive integer multiple of dt")
            out.extend([ham] * int(round(k)))
        if len(out) != n_intervals:
            raise ValueError("piece durations must sum to elapsed_time")
        return out
