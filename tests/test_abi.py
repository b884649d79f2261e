"""CPU-side checks of the C-ABI library: it loads and exports every symbol the header declares."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "tjm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tjm_[a-z0-9_]+)\s*\(", text)))


@pytest.mark.parametrize("dtype", ["complex128", "complex64"])
def test_library_exports_every_declared_symbol(dtype):
    """Both builds of the sources - libtjm_hip.so (fp64) and libtjm_hip_f32.so (complex64, -DTJM_F32) - export the whole C ABI."""
    from yaqs_amd import _lib

    lib = _lib.load(dtype)
    names = header_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/tjm_hip.h but not exported"
        assert n in _lib.EXPORTS, f"{n} has no ctypes signature in yaqs_amd/_lib.py"
    assert lib.tjm_version() >= 100
    assert lib.tjm_error_string(-4) == b"not implemented"


def test_engine_argument_validation_without_gpu():
    from yaqs_amd import _lib

    lib = _lib.load()
    h = ctypes.c_void_p()
    bonds = np.array([1, 3, 3, 1], dtype=np.int32)
    assert lib.tjm_engine_create(ctypes.byref(h), 3, 2, 8, 4, bonds.ctypes.data) == 0
    assert lib.tjm_engine_workspace_bytes(h) > 0
    caps = np.zeros(4, dtype=np.int32)
    lib.tjm_engine_bond_caps(h, caps.ctypes.data)
    assert list(caps) == [1, 2, 2, 1]
    # params validation happens on the host
    assert lib.tjm_engine_set_params(h, -1.0, 1e-6, 0, 8, 1e-4, 2, 1) == -1
    assert lib.tjm_engine_set_params(h, 0.1, 1e-6, 7, 8, 1e-4, 2, 1) == -1
    assert lib.tjm_engine_set_params(h, 0.1, 1e-6, 0, 16, 1e-4, 2, 1) == 0  # a cap above chi_max is legal: clipped truncations are reported
    assert lib.tjm_engine_set_params(h, 0.1, 1e-6, 0, 8, 1e-4, 2, 1) == 0
    # operations before bind() are refused, not executed
    assert lib.tjm_engine_tdvp(h, 0) == -6
    lib.tjm_engine_destroy(h)
    bad = np.array([2, 3, 3, 1], dtype=np.int32)
    assert lib.tjm_engine_create(ctypes.byref(h), 3, 2, 8, 4, bad.ctypes.data) == -1


def test_product_path_has_no_cpu_fallback():
    import torch

    from yaqs_amd import _lib
    from yaqs_amd.engine import BatchEngine

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.TjmError):
        BatchEngine(4, 4, 2, [np.zeros((2, 2, 1, 1))] * 4)


def test_host_mirror_of_reference_interface():
    from yaqs_amd.api import AnalogSimParams, NoiseModel, Observable, X, Z, is_pauli

    p = AnalogSimParams(observables=[Observable(Z(), 2), Observable(X(), 0), Observable(Z(), 0)], elapsed_time=1.0, dt=0.1)
    assert len(p.times) == 11 and p.times[-1] == 1.0
    assert p.max_bond_dim == 128 and p.svd_threshold == 1e-6 and p.krylov_tol == 1e-4 and p.num_traj == 256  # "balanced"
    assert p.observable_sorted_indices == (2, 0, 1)  # site-sorted, stable (simulation_parameters.py:419-456)
    with pytest.raises(ValueError):
        AnalogSimParams(elapsed_time=0.25, dt=0.1)
    nm = NoiseModel([{"name": "pauli_z", "sites": [0], "strength": 0.1}, {"name": "lowering", "sites": [1], "strength": 0.1},
                     {"name": "crosstalk_xy", "sites": [1, 0], "strength": 0.1}, {"name": "longrange_crosstalk_zz", "sites": [0, 3], "strength": 0.1}])
    assert [is_pauli(q) for q in nm.processes] == [True, False, True, True]
    assert nm.processes[2]["sites"] == [0, 1]
    assert np.allclose(nm.processes[2]["matrix"], np.kron([[0, -1j], [1j, 0]], [[0, 1], [1, 0]]))  # swapped -> Y on 0, X on 1
    with pytest.raises(ValueError):
        NoiseModel([{"name": "pauli_z", "sites": [0], "strength": -1.0}])


def test_trajectory_uniform_streams_match_oracle():
    from oracle import tjm_oracle as o
    from yaqs_amd.tjm import sample_uniforms, shard_range, trajectory_uniforms

    assert np.array_equal(trajectory_uniforms(42, 3, 6), o.trajectory_rng(42, 3).random(6))
    assert np.array_equal(sample_uniforms(42, 3, 5), o.sample_rng(42, 3, 5).random(2))
    parts = [shard_range(1024, r, 8) for r in range(8)]
    assert parts[0][0] == 0 and parts[-1][1] == 1024 and all(parts[i][1] == parts[i + 1][0] for i in range(7))


def test_c_rng_streams_match_reference_fixture_bit_exactly():
    """tjm_rng_uniforms (host code of the C ABI, no GPU involved) reproduces make_trajectory_rng / make_sample_rng of
    core/random_utils.py:20-69 bit for bit: SeedSequence([seed, traj, TAG]) -> PCG64 -> random()."""
    import ctypes as C

    from yaqs_amd import _lib

    lib = _lib.load()
    g = np.load(os.path.join(GOLDEN, "rng_streams.npz"))
    out = np.zeros(8)
    for i, seed in enumerate(g["seeds"]):
        for j, traj in enumerate(g["trajs"]):
            _lib.check(lib.tjm_rng_uniforms(1, int(seed), int(traj), -1, 8, out.ctypes.data))
            assert np.array_equal(out, g["traj"][i, j]), (seed, traj)
            for k, step in enumerate(g["steps"]):
                _lib.check(lib.tjm_rng_uniforms(1, int(seed), int(traj), int(step), 8, out.ctypes.data))
                assert np.array_equal(out, g["sample"][i, j, k]), (seed, traj, step)
    # a seed beyond 32 bits splits into two entropy words exactly as NumPy does
    big = 2**40 + 12345
    ref = np.random.default_rng(np.random.SeedSequence([big, 7, 0x5452414A])).random(8)
    _lib.check(lib.tjm_rng_uniforms(1, big, 7, -1, 8, out.ctypes.data))
    assert np.array_equal(out, ref)
    # unseeded: two calls differ
    a, b = np.zeros(4), np.zeros(4)
    lib.tjm_rng_uniforms(0, 0, 0, -1, 4, a.ctypes.data)
    lib.tjm_rng_uniforms(0, 0, 0, -1, 4, b.ctypes.data)
    assert not np.array_equal(a, b) and np.all((a >= 0) & (a < 1))


def test_static_disorder_sampling_matches_reference_draws():
    """NoiseModel.sample (noise_model.py:492-559) with make_disorder_rng (random_utils.py:72-87): same NumPy calls in the same
    order, so the realisation equals the reference's for a given seed; checked here against the draws spelled out."""
    from yaqs_amd.api import NoiseModel
    from yaqs_amd.tjm import disorder_rng

    procs = [{"name": "pauli_z", "sites": [0], "strength": {"distribution": "normal", "mean": 0.1, "std": 0.02}},
             {"name": "pauli_x", "sites": [1], "strength": 0.3},
             {"name": "lowering", "sites": [2], "strength": {"distribution": "lognormal", "mean": -2.0, "std": 0.1}},
             {"name": "pauli_y", "sites": [3], "strength": {"distribution": "truncated_normal", "mean": 0.05, "std": 0.1}},
             {"name": "pauli_z", "sites": [4], "strength": {"distribution": "normal", "mean": -5.0, "std": 0.1}}]
    nm = NoiseModel(procs)
    assert nm.has_disorder
    got = nm.sample(rng=disorder_rng(42))
    from scipy.stats import truncnorm

    g = np.random.default_rng(np.random.SeedSequence([42, 0x4449534F]))
    want = [max(0.0, float(g.normal(loc=0.1, scale=0.02))), 0.3, float(g.lognormal(mean=-2.0, sigma=0.1)),
            float(truncnorm.rvs(-0.5, np.inf, loc=0.05, scale=0.1, random_state=g)), max(0.0, float(g.normal(loc=-5.0, scale=0.1)))]
    assert [q["strength"] for q in got.processes] == want and want[-1] == 0.0
    assert not got.has_disorder and nm.has_disorder  # the template is untouched
    with pytest.raises(ValueError):
        NoiseModel([{"name": "pauli_z", "sites": [0], "strength": {"distribution": "cauchy", "mean": 0, "std": 1}}])


def test_simulation_presets_are_the_reference_table():
    """simulation_parameters.py:46-51: the four presets, value by value; explicit arguments override them (:520-613)."""
    from yaqs_amd.api import AnalogSimParams, SIMULATION_PRESETS

    assert SIMULATION_PRESETS == {
        "fast": {"svd_threshold": 1e-3, "max_bond_dim": 16, "num_traj": 128, "krylov_tol": 1e-3},
        "balanced": {"svd_threshold": 1e-6, "max_bond_dim": 128, "num_traj": 256, "krylov_tol": 1e-4},
        "accurate": {"svd_threshold": 1e-9, "max_bond_dim": 4096, "num_traj": 1024, "krylov_tol": 1e-6},
        "exact": {"svd_threshold": 1e-13, "max_bond_dim": None, "num_traj": 1024, "krylov_tol": 1e-12},
    }
    p = AnalogSimParams(observables=[])
    assert (p.max_bond_dim, p.svd_threshold, p.num_traj, p.krylov_tol) == (128, 1e-6, 256, 1e-4)
    q = AnalogSimParams(observables=[], preset="exact", max_bond_dim=7)
    assert (q.max_bond_dim, q.svd_threshold) == (7, 1e-13)
    assert AnalogSimParams(observables=[], preset="accurate", max_bond_dim=None).max_bond_dim is None


def test_capacity_ladder_and_bounds():
    """Host logic of the storage capacity (yaqs_amd/tjm.py): first capacity, the largest one a run can need, the ladder."""
    from yaqs_amd.api import AnalogSimParams, MPS
    from yaqs_amd.tjm import CAPACITY_LADDER, MAX_CHI, engine_bond_caps, grown_capacity

    p = AnalogSimParams(observables=[], max_bond_dim=4096)
    assert engine_bond_caps(p, MPS(40, state="x+")) == (8, 4096)
    assert engine_bond_caps(p, MPS(6, state="x+")) == (8, 8)            # 2**(L//2) bounds every bond of a short chain
    assert engine_bond_caps(AnalogSimParams(observables=[], max_bond_dim=None), MPS(12, state="x+")) == (8, 64)
    assert engine_bond_caps(AnalogSimParams(observables=[], max_bond_dim=4), MPS(12, state="x+")) == (4, 4)
    assert engine_bond_caps(AnalogSimParams(observables=[], max_bond_dim=64, tdvp_mode="1site"), MPS(12, state="x+", pad=32))[0] == 32
    chi, seen = 8, [8]
    while chi < 200:
        chi = grown_capacity(chi, 200)
        seen.append(chi)
    assert seen == [c for c in CAPACITY_LADDER if c < 200] + [200]
    with pytest.raises(NotImplementedError):
        grown_capacity(MAX_CHI, 4096)
    with pytest.raises(RuntimeError):
        grown_capacity(16, 16)


def test_growing_run_bookkeeping_without_a_gpu(monkeypatch):
    """Simulator._run_growing with stand-in engines: a piece that runs out of capacity at step j hands its rows, cursors and
    measured columns to larger engines, is split when those hold fewer trajectories, and every source engine is closed once its
    last successor has taken over."""
    import yaqs_amd.tjm as tjm_mod
    from yaqs_amd._lib import CapacityError

    log = {"built": [], "closed": [], "adopted": []}

    class FakeEngine:
        def __init__(self, length, chi_max, batch, mpo, device=None):
            self.chi_max, self.B, self.rows = chi_max, batch, None
            log["built"].append((chi_max, batch))

        def adopt(self, src, first):
            self.rows = src.rows[first: first + self.B]
            log["adopted"].append((src.chi_max, self.chi_max, first, self.B))

        def close(self):
            log["closed"].append((self.chi_max, self.B))

    monkeypatch.setattr(tjm_mod, "BatchEngine", FakeEngine)
    monkeypatch.setattr(tjm_mod.Simulator, "_batch_for", lambda self, remaining, length, chi, mpo, device: min(remaining, {8: 6, 16: 4, 24: 2}[chi]))
    n_steps = 6
    need = {0: 8, 1: 8, 2: 16, 3: 16, 4: 24, 5: 24}  # capacity that time step j needs

    def run_piece(engine, lo, hi, resume):
        rows = list(range(lo, hi))
        engine.rows = rows
        start = 0 if resume is None else resume["start"][0]
        res = np.zeros((hi - lo, 1, n_steps)) if resume is None else resume["results"]
        dg = np.zeros((hi - lo, 3, n_steps)) if resume is None else resume["diagnostics"]
        pos = np.zeros(hi - lo, dtype=np.int64) if resume is None else np.asarray(resume["rng_pos"]).copy()
        if resume is not None:
            assert list(pos) == [start * (t + 1) for t in rows]  # the cursors travelled with their trajectories
        for j in range(start, n_steps):
            if need[j] > engine.chi_max:
                err = CapacityError("clipped")
                err.resume, err.rng_pos, err.results, err.diagnostics = ((j, 0) if j > 0 else (0, 0)), pos, res, dg
                raise err
            res[:, 0, j] = [100 * t + j for t in rows]
            pos += np.array([t + 1 for t in rows])
        return res, dg

    sim = tjm_mod.Simulator()
    res, dg, kept = sim._run_growing(list(range(6)), 8, 64, 10, None, lambda e: e, run_piece, "cpu", n_steps, 1)
    assert kept is None
    assert np.array_equal(res[:, 0, :], np.array([[100 * t + j for j in range(n_steps)] for t in range(6)]))
    assert log["built"][0] == (8, 6) and sorted(c for c, _ in log["built"]) == [8, 16, 16, 24, 24, 24]
    assert sorted(log["closed"]) == sorted(log["built"])           # nothing leaks
    assert all(a[0] < a[1] for a in log["adopted"]) and len(log["adopted"]) == 5


def test_noise_model_normalisation_and_error_behaviour():
    """Host logic of the data format on the boundary (noise_model.py:227-490, 668-790), behaviour by behaviour as the reference's own
    tests document it (tests/core/data_structures/test_noise_model.py): site order with the operators following their sites, library
    lookups, every construction error with its type and wording, run-context validation."""
    from yaqs_amd.api import AnalogSimParams, NoiseModel, Observable, Z, is_pauli, validate_noise_model_for_run

    X2, Y2, Z2 = (NoiseModel.get_operator(n) for n in ("x", "y", "z"))
    # normalisation
    p = NoiseModel([{"name": "custom_longrange_xy", "sites": [3, 1], "strength": 0.3, "factors": (X2, Y2)}]).processes[0]
    assert p["sites"] == [1, 3] and np.allclose(p["factors"][0], Y2) and np.allclose(p["factors"][1], X2) and "matrix" not in p
    p = NoiseModel([{"name": "longrange_crosstalk_xy", "sites": [0, 2], "strength": 0.3}]).processes[0]
    assert np.allclose(p["factors"][0], X2) and np.allclose(p["factors"][1], Y2) and "matrix" not in p
    p = NoiseModel([{"name": "crosstalk_xy", "sites": (1, 0), "strength": 0.1}]).processes[0]
    assert p["sites"] == [0, 1] and np.allclose(p["matrix"], np.kron(Y2, X2))
    custom = np.diag([1.0, 2.0, 3.0, 4.0]).astype(complex)
    assert np.allclose(NoiseModel([{"name": "crosstalk_xy", "sites": [0, 1], "strength": 0.1, "matrix": custom}]).processes[0]["matrix"], custom)
    assert NoiseModel([{"name": "raising_two", "sites": [0, 1], "strength": 0.1}]).processes[0]["matrix"].shape == (4, 4)
    assert np.allclose(NoiseModel.get_operator("crosstalk_xy"), np.kron(X2, Y2))
    op = NoiseModel.get_operator("pauli_x")
    op[0, 0] = 99.0
    assert NoiseModel.get_operator("pauli_x")[0, 0] == 0.0
    almost_x = X2.copy()
    almost_x[0, 1] += 5e-6
    assert is_pauli(NoiseModel([{"name": "almost_x", "sites": [0], "strength": 0.1, "matrix": almost_x}]).processes[0]) is False
    jumps = NoiseModel(scheduled_jumps=[{"time": 0.0, "sites": [0, 1], "name": "crosstalk_xx"}]).scheduled_jumps
    assert jumps[0]["sites"] == [0, 1] and np.allclose(jumps[0]["matrix"], np.kron(X2, X2))
    # construction errors: (exception, fragment of the message, keyword arguments)
    lr = {"name": "custom", "sites": [0, 2], "strength": 0.1}
    cases = [
        (ValueError, "factors", dict(processes=[{"name": "foo_bar", "sites": [0, 2], "strength": 0.1}])),
        (ValueError, "must contain 'distribution' key", dict(processes=[{"name": "pauli_x", "sites": [0], "strength": {"mean": 0.5, "std": 0.1}}])),
        (ValueError, "nonnegative", dict(processes=[{"name": "pauli_x", "sites": [0], "strength": -0.1}])),
        (ValueError, "finite", dict(processes=[{"name": "pauli_x", "sites": [0], "strength": np.nan}])),
        (TypeError, "booleans", dict(processes=[{"name": "pauli_x", "sites": [True], "strength": 0.1}])),
        (TypeError, "booleans", dict(processes=[{"name": "pauli_x", "sites": [0], "strength": True}])),
        (ValueError, "distinct", dict(processes=[{"name": "pauli_x", "sites": [1, 1], "strength": 0.1}])),
        (ValueError, "exactly 1 or 2", dict(processes=[{"name": "pauli_x", "sites": [], "strength": 0.1}])),
        (ValueError, "ascending site order", dict(processes=[{"name": "custom", "sites": [1, 0], "strength": 0.1, "matrix": np.kron(X2, Z2)}])),
        (ValueError, "Unknown noise operator", dict(processes=[{"name": "not_an_operator", "sites": [0], "strength": 0.1}])),
        (ValueError, "std must be nonnegative", dict(processes=[{"name": "pauli_x", "sites": [0], "strength": {"distribution": "normal", "mean": 0.1, "std": -0.1}}])),
        (ValueError, "Unknown distribution keys", dict(processes=[{"name": "pauli_x", "sites": [0], "strength": {"distribution": "normal", "mean": 0.1, "stdev": 0.1}}])),
        (ValueError, "non-adjacent", dict(scheduled_jumps=[{"time": 0.0, "sites": [0, 2], "name": "x"}])),
        (TypeError, "booleans", dict(scheduled_jumps=[{"time": True, "sites": [0], "name": "x"}])),
        (ValueError, "both 'matrix' and 'factors'", dict(processes=[dict(lr, matrix=np.eye(4, dtype=complex), factors=(X2, Y2))])),
        (ValueError, "not None", dict(processes=[dict(lr, factors=None)])),
        (ValueError, "do not accept 'factors'", dict(scheduled_jumps=[{"time": 0.0, "sites": [0], "name": "x", "factors": None}])),
        (TypeError, "dictionary", dict(processes=["not-a-dict"])),
        (TypeError, "list or tuple", dict(processes={"name": "pauli_x", "sites": [0], "strength": 0.1})),
        (TypeError, "list or tuple", dict(scheduled_jumps={"time": 0.0, "sites": [0], "name": "x"})),
        (TypeError, "must be a string", dict(processes=[{"name": 1, "sites": [0], "strength": 0.1}])),
        (ValueError, "nonempty", dict(processes=[{"name": "", "sites": [0], "strength": 0.1}])),
        (TypeError, "list or tuple of integers", dict(processes=[{"name": "pauli_x", "sites": 0, "strength": 0.1}])),
        (ValueError, "nonnegative", dict(processes=[{"name": "pauli_x", "sites": [-1], "strength": 0.1}])),
        (TypeError, "numeric array", dict(processes=[{"name": "custom", "sites": [0], "strength": 0.1, "matrix": object()}])),
        (ValueError, "square", dict(processes=[{"name": "custom", "sites": [0], "strength": 0.1, "matrix": np.ones((2, 3))}])),
        (ValueError, "finite", dict(processes=[{"name": "custom", "sites": [0], "strength": 0.1, "matrix": np.array([[np.nan, 0], [0, 1]])}])),
        (ValueError, "One-site processes do not accept", dict(processes=[{"name": "custom", "sites": [0], "strength": 0.1, "factors": (X2, Y2)}])),
        (ValueError, "use 'matrix', not 'factors'", dict(processes=[{"name": "custom", "sites": [0, 1], "strength": 0.1, "factors": (X2, Y2)}])),
        (ValueError, "require 'factors'", dict(processes=[dict(lr, matrix=np.eye(4, dtype=complex))])),
        (ValueError, "exactly two", dict(processes=[dict(lr, factors=(X2,))])),
        (ValueError, "'time' key", dict(scheduled_jumps=[{"sites": [0], "name": "x"}])),
        (ValueError, "ascending site order", dict(scheduled_jumps=[{"time": 0.0, "sites": [1, 0], "name": "custom", "matrix": np.kron(X2, Z2)}])),
        (ValueError, "finite", dict(scheduled_jumps=[{"time": np.nan, "sites": [0], "name": "x"}])),
    ]
    for exc, fragment, kw in cases:
        with pytest.raises(exc, match=fragment):
            NoiseModel(**kw)
    nm = NoiseModel([{"name": "pauli_x", "sites": [0], "strength": 0.1}])
    nm.processes[0]["strength"] = {"distribution": "bogus", "mean": 0.0, "std": 0.1}
    with pytest.raises(ValueError, match="Unsupported distribution type"):
        nm.sample(rng=0)
    # run-context validation
    sp = AnalogSimParams(observables=[Observable(Z(), 0)], dt=0.1, elapsed_time=0.2, order=1, get_state=True)
    validate_noise_model_for_run(NoiseModel([{"name": "pauli_x", "sites": [0], "strength": 0.1}]), length=2, physical_dimensions=2, representation="mps", sim_params=sp)
    lowering = np.array([[0, 1], [0, 0]], dtype=complex)
    sched = NoiseModel(scheduled_jumps=[{"time": 0.0, "sites": [0], "name": "x"}])
    run_cases = [
        ("out of range", NoiseModel([{"name": "pauli_x", "sites": [3], "strength": 0.1}]), dict(length=2)),
        ("matrix shape", NoiseModel([{"name": "custom", "sites": [0], "strength": 0.1, "matrix": np.eye(3, dtype=complex)}]), dict(length=2)),
        ("factor on site", NoiseModel([dict(lr, factors=(np.eye(3, dtype=complex), Y2))]), dict(length=3)),
        ("Digital TJM does not support non-adjacent", NoiseModel([{"name": "longrange_crosstalk_xy", "sites": [0, 2], "strength": 0.1}]), dict(length=3, is_digital=True)),
        ("non-Pauli long-range", NoiseModel([dict(lr, factors=(lowering, Y2))]), dict(length=3)),
        ("AnalogSimParams are required", sched, dict(length=2, sim_params=None)),
        ("only supported for single-State analog MPS", sched, dict(length=2, is_digital=True, sim_params=sp)),
        ("only supported for single-State analog MPS", sched, dict(length=2, representation="vector", sim_params=sp)),
        ("order=1", sched, dict(length=2, sim_params=AnalogSimParams(observables=[Observable(Z(), 0)], dt=0.1, elapsed_time=0.2, order=2, get_state=True))),
        ("not on the simulation time grid", NoiseModel(scheduled_jumps=[{"time": 0.05, "sites": [0], "name": "x"}]), dict(length=2, sim_params=sp)),
    ]
    for fragment, model, kw in run_cases:
        kw.setdefault("representation", "mps")
        with pytest.raises(ValueError, match=fragment):
            validate_noise_model_for_run(model, **kw)
    validate_noise_model_for_run(sched, length=2, representation="mps", sim_params=sp)


def test_normal_strength_below_zero_is_clamped_with_a_warning(caplog):
    """noise_model.py:521-533: a normal draw below zero becomes 0 and is reported."""
    import logging

    from yaqs_amd.api import NoiseModel

    nm = NoiseModel([{"name": "pauli_x", "sites": [0], "strength": {"distribution": "normal", "mean": -5.0, "std": 0.1}}])
    with caplog.at_level(logging.WARNING):
        out = nm.sample(rng=np.random.default_rng(1))
    assert "was negative and clamped to 0.0" in caplog.text
    assert out.processes[0]["strength"] == 0.0


def test_simulation_parameter_classes_validate_like_the_reference():
    """Host logic of AnalogSimParams / DigitalSimParams / Observable (simulation_parameters.py:100-745), behaviour by behaviour as the
    reference's tests document it (tests/core/data_structures/test_simulation_parameters.py): the time grid and its rounding dust,
    presets and overrides, every validation error with type and wording, observable construction from names / matrices / bitstrings,
    the site-sorted worker order with PVMs last."""
    from yaqs_amd.api import (SIMULATION_PRESETS, AnalogSimParams, DigitalSimParams, EvolutionMode, Observable, PVM, SchmidtSpectrum, X, Y, Z,
                              _validate_tdvp_sweeps)

    obs = [Observable(X(), 0)]
    p = AnalogSimParams(observables=obs, elapsed_time=1.0, dt=0.2, num_traj=50)
    assert np.allclose(p.times, [0.0, 0.2, 0.4, 0.6, 0.8, 1.0]) and p.sample_timesteps is True and p.num_traj == 50 and p.order == 1
    assert np.allclose(AnalogSimParams(observables=obs, elapsed_time=0.0, dt=0.1).times, [0.0])
    for elapsed, dt in ((100.1, 0.1), (1.0, 1.0 / 9015), (1.23456789, 1e-8)):  # float64 rounding dust is not a fraction of a step
        assert AnalogSimParams(observables=obs, elapsed_time=elapsed, dt=dt).times[-1] == elapsed
    for elapsed, dt in ((0.15, 0.1), (0.25, 0.1), (5e-13, 1e-12), (1.5e-12, 1e-12), (1.0, 1e9)):
        with pytest.raises(ValueError, match="integer multiple"):
            AnalogSimParams(observables=obs, elapsed_time=elapsed, dt=dt)
    for elapsed, dt, fragment in ((-0.1, 0.1, "non-negative"), (0.1, 0.0, "positive"), (0.1, -0.1, "positive"), (float("nan"), 0.1, "finite"),
                                  (0.1, float("inf"), "finite"), (1e308, 1e-308, "elapsed_time / dt must be finite")):
        with pytest.raises(ValueError, match=fragment):
            AnalogSimParams(observables=obs, elapsed_time=elapsed, dt=dt)
    for elapsed, dt in ((True, 0.1), (0.1, False), ("0.1", 0.1), (0.1, None)):
        with pytest.raises(TypeError, match="real number"):
            AnalogSimParams(observables=obs, elapsed_time=elapsed, dt=dt)
    for cls, kw in ((AnalogSimParams, {}), (DigitalSimParams, {"get_state": True}), (DigitalSimParams, {"shots": 100})):
        for preset, expected in SIMULATION_PRESETS.items():
            q = cls(preset=preset, **kw)
            assert (q.preset, q.svd_threshold, q.max_bond_dim, q.krylov_tol) == (preset, expected["svd_threshold"], expected["max_bond_dim"], expected["krylov_tol"])
        q = cls(preset="fast", svd_threshold=1e-8, max_bond_dim=512, krylov_tol=1e-12, **kw)
        assert (q.svd_threshold, q.max_bond_dim, q.krylov_tol) == (1e-8, 512, 1e-12)
        assert cls(preset="balanced", max_bond_dim=None, **kw).max_bond_dim is None
        assert (cls(**kw).tdvp_mode, cls(**kw).tdvp_sweeps) == ("2site", 1)
        for bad in ("invalid", None):
            with pytest.raises(ValueError, match="preset must be one of"):
                cls(preset=bad, **kw)
        for bad in ("nope", ["discarded_weight"], 1, None):
            with pytest.raises(ValueError, match="trunc_mode"):
                cls(trunc_mode=bad, **kw)
        with pytest.raises(ValueError, match="tdvp_mode"):
            cls(tdvp_mode="invalid", **kw)
        for bad in (0, -1):
            with pytest.raises(ValueError, match="tdvp_sweeps"):
                cls(tdvp_sweeps=bad, **kw)
        with pytest.raises(TypeError, match="random_seed must be int or None"):
            cls(random_seed="not-a-seed", **kw)
        with pytest.raises(ValueError, match="random_seed must be non-negative"):
            cls(random_seed=-1, **kw)
        for bad in (0.0, -1.0, float("inf"), float("nan")):
            with pytest.raises(ValueError, match="krylov_tol must be a finite positive float"):
                cls(krylov_tol=bad, **kw)
        for bad in (-1.0, float("inf"), float("nan")):
            with pytest.raises(ValueError, match="svd_threshold must be a finite non-negative float"):
                cls(svd_threshold=bad, **kw)
        assert cls(svd_threshold=0.0, **kw).svd_threshold == 0.0
    for bad in (1.5, True):
        with pytest.raises(TypeError, match="tdvp_sweeps"):
            _validate_tdvp_sweeps(bad)
    assert AnalogSimParams(evolution_mode="bug").evolution_mode is EvolutionMode.BUG
    with pytest.raises(ValueError, match="evolution_mode"):
        AnalogSimParams(evolution_mode="not-a-mode")
    assert DigitalSimParams(shots=1).gate_mode == "mpo" and DigitalSimParams(get_state=True, gate_mode="full-tdvp").gate_mode == "full-tdvp"
    with pytest.raises(ValueError, match="gate_mode"):
        DigitalSimParams(get_state=True, gate_mode="invalid")
    for bad in (0, -1):
        with pytest.raises(ValueError, match="shots must be a positive int"):
            DigitalSimParams(shots=bad)
    with pytest.raises(TypeError, match=r"keyword-only|takes 1 positional"):
        DigitalSimParams([Observable("z", 0)])
    d = DigitalSimParams()
    assert d.observables == [] and d.shots is None and not d.get_state
    # observables
    assert Observable("entropy", sites=[3, 4]).gate.name == "entropy" and Observable("schmidt_spectrum", sites=[3, 4]).sites == [3, 4]
    pvm = Observable("10101", sites=None)
    assert pvm.gate.name == "pvm" and pvm.gate.bitstring == "10101" and np.allclose(pvm.gate.matrix, np.eye(2))
    assert Observable("pvm").gate.bitstring == "pvm"
    local = Observable(np.diag([1.0, -1.0]), 0)
    assert local.gate.name == "local" and local.gate.interaction == 1
    for bad in (np.ones(3), np.ones((2, 3))):
        with pytest.raises(ValueError, match="Local operator matrix"):
            Observable(bad, 0)
    pos = Observable("position", 2, positions=[-1.0, 0.0, 1.0])  # gate_library.py:1845-1872
    assert pos.gate.name == "position" and pos.sites == 2 and np.array_equal(pos.gate.matrix, np.diag([-1.0, 0.0, 1.0]))
    for bad, err in (([1j, 0], "real"), ([], "non-empty"), ([[0.0, 1.0]], "non-empty"), ([0.0, np.inf], "finite")):
        with pytest.raises(ValueError, match=err):
            Observable("position", 0, positions=bad)
    with pytest.raises(TypeError, match="positions"):
        Observable("position", 0)
    with pytest.raises(TypeError, match="unexpected keyword argument 'positions'"):
        Observable("z", 0, positions=[0.0, 1.0])
    with pytest.raises(TypeError, match="only supported for named observables"):
        Observable(np.eye(2), 0, positions=[0.0, 1.0])
    # worker order: by first site, ties by user order, PVMs last; derived from the current list
    z3, x2, y1, ssp = Observable(Z(), sites=3), Observable(X(), sites=2), Observable(Y(), sites=1), Observable(SchmidtSpectrum(), sites=[1, 2])
    dp = DigitalSimParams(observables=[z3, x2, y1, ssp], num_traj=7, max_bond_dim=128, get_state=True, sample_layers=True, num_mid_measurements=2)
    assert dp.sorted_observables == [y1, ssp, x2, z3] and dp.observable_sorted_indices == (3, 2, 0, 1)
    dp.observables.append(Observable(Z(), sites=0))
    assert dp.observable_sorted_indices == (4, 3, 1, 2, 0)
    with pytest.raises(AssertionError):
        DigitalSimParams(observables=[Observable(PVM("101"), sites=None), Observable(Z(), sites=0)])
    DigitalSimParams(observables=[Observable(PVM("0"), sites=None), Observable(PVM("1"), sites=None)])


def test_preset_states_basis_and_random():
    """``MPS(length, state="basis", basis_string=...)`` puts character i of the string on site i (mps.py:395-408); ``state="random"``
    gives one normalised (r, 1 - r) vector per site (mps.py:266-269); ``State`` forwards both and rejects preset arguments next to
    ``tensors=`` (state_utils.py:39-76)."""
    from yaqs_amd.api import MPS, State

    m = MPS(5, state="basis", basis_string="01101")
    assert [t.shape for t in m.tensors] == [(2, 1, 1)] * 5
    assert [int(np.argmax(np.abs(t[:, 0, 0]))) for t in m.tensors] == [0, 1, 1, 0, 1]
    vec = m.to_vec()  # site 0 is the least significant bit of the dense index
    assert np.flatnonzero(np.abs(vec) > 0).tolist() == [0b10110]
    st = State(4, initial="basis", basis_string="1000")
    assert st.basis_string == "1000" and st.tensors[0][1, 0, 0] == 1 and all(t[0, 0, 0] == 1 for t in st.tensors[1:])
    for bad in ("010", "01x01", None):
        with pytest.raises(ValueError, match="basis_string"):
            MPS(5, state="basis", basis_string=bad)
    r = MPS(6, state="random", rng=np.random.default_rng(4))
    assert all(t.shape == (2, 1, 1) and abs(np.linalg.norm(t) - 1) < 1e-15 and np.all(t.real >= 0) and np.all(t.imag == 0) for t in r.tensors)
    assert abs(np.linalg.norm(r.to_vec()) - 1) < 1e-14
    for kw in (dict(initial="x+"), dict(pad=2), dict(basis_string="00"), dict(seed=1)):
        with pytest.raises(ValueError, match="preset"):
            State(tensors=m.tensors[:2], **kw)


def _dense_of(mpo):
    """Dense matrix of an MPO, site 0 the most significant index (test helper)."""
    acc = np.ones((1, 1, 1), dtype=complex)  # (out, in, bond)
    for t in mpo.tensors:
        acc = np.einsum("oib,pqbc->opiqc", acc, t).reshape(acc.shape[0] * 2, acc.shape[1] * 2, t.shape[3])
    return acc[:, :, 0]


def _embed(L, ops):
    out = np.ones((1, 1), dtype=complex)
    for s in range(L):
        out = np.kron(out, ops.get(s, np.eye(2)))
    return out


def test_periodic_boundary_conditions_close_the_chain():
    """``bc="periodic"`` adds the bond (L-1, 0) with the first operator of each two-body term on site L-1 (mpo.py:304-308): checked
    against explicit Kronecker sums for ising, heisenberg and a non-symmetric pauli term, for L = 2 (the bond counted twice) to 5."""
    from yaqs_amd.api import MPO, Hamiltonian

    X, Y, Z = (np.array(m, dtype=complex) for m in ([[0, 1], [1, 0]], [[0, -1j], [1j, 0]], [[1, 0], [0, -1]]))
    for L in (2, 3, 5):
        bonds = [(i, (i + 1) % L) for i in range(L)]
        want = sum(-1.3 * _embed(L, {i: Z}) @ _embed(L, {j: Z}) for i, j in bonds) + sum(-0.4 * _embed(L, {i: X}) for i in range(L))
        assert np.allclose(_dense_of(MPO.ising(L, 1.3, 0.4, bc="periodic")), want, atol=1e-14)
        want = sum(-(0.5 * _embed(L, {i: X}) @ _embed(L, {j: X}) + 0.7 * _embed(L, {i: Y}) @ _embed(L, {j: Y}) + 1.1 * _embed(L, {i: Z}) @ _embed(L, {j: Z}))
                   for i, j in bonds) + sum(-0.2 * _embed(L, {i: Z}) for i in range(L))
        assert np.allclose(_dense_of(Hamiltonian.heisenberg(L, 0.5, 0.7, 1.1, 0.2, bc="periodic")), want, atol=1e-14)
        want = sum(0.9 * _embed(L, {i: X}) @ _embed(L, {j: Z}) for i, j in bonds) + sum(0.3 * _embed(L, {i: Y}) for i in range(L))
        got = MPO.pauli(length=L, two_body=[(0.9, "X", "Z")], one_body=[(0.3, "Y")], bc="periodic")
        assert np.allclose(_dense_of(got), want, atol=1e-14)
        assert all(t.shape[:2] == (2, 2) for t in got.tensors) and got.tensors[0].shape[2] == 1 and got.tensors[-1].shape[3] == 1
    assert np.allclose(_dense_of(MPO.ising(4, 1.3, 0.4)), _dense_of(MPO.ising(4, 1.3, 0.4, bc="open")))
    with pytest.raises(ValueError, match="bc must be"):
        MPO.ising(4, 1.0, 1.0, bc="twisted")
    with pytest.raises(ValueError, match="at least two sites"):
        MPO.ising(1, 1.0, 1.0, bc="periodic")


def test_pauli_sum_hamiltonians_and_dense_converters():
    """``MPO().from_pauli_sum(terms=..., length=...)`` (mpo.py:1171-1318) against explicit Kronecker sums, including long-range and
    three-body strings, complex coefficients and the identity term; the bond dimensions are the operator Schmidt ranks (3 for the Ising
    chain); ``to_matrix`` has site 0 as the most significant index, ``to_matrix_mps_order`` acts on ``MPS.to_vec`` vectors; malformed
    terms raise ValueError; ``identity`` and ``custom`` as in the reference."""
    from yaqs_amd.api import MPO, MPS, Observable, Z as Zg, X as Xg

    pa = {"I": np.eye(2, dtype=complex), "X": np.array([[0, 1], [1, 0]], dtype=complex), "Y": np.array([[0, -1j], [1j, 0]]),
          "Z": np.diag([1.0, -1.0]).astype(complex)}
    L = 5
    terms = [(-1.0, f"Z{i} Z{i + 1}") for i in range(L - 1)] + [(-0.5, f"X{i}") for i in range(L)]
    ising = MPO()
    ising.from_pauli_sum(terms=terms, length=L)
    assert [t.shape[3] for t in ising.tensors] == [3, 3, 3, 3, 1] and ising.length == L
    assert np.allclose(ising.to_matrix(), MPO.ising(L, 1.0, 0.5).to_matrix(), atol=1e-13)
    more = terms + [(0.3 + 0.1j, "Y0 X2 Z4"), (0.7, "x1 z4"), (2.0, "")]
    m = MPO()
    m.from_pauli_sum(terms=more, length=L)
    want = sum(c * _embed(L, {int(t[1:]): pa[t[0].upper()] for t in spec.split()}) for c, spec in more)
    assert np.allclose(m.to_matrix(), want, atol=1e-12)
    assert np.allclose(_dense_of(m), m.to_matrix())
    # the two dense layouts: a haar state's <Z_0>, <X_3> through to_vec and the LSB-ordered matrix
    psi = MPS(L, state="haar-random", pad=4, rng=np.random.default_rng(0))
    vec = psi.to_vec()
    for label, gate, site in (("Z", Zg(), 0), ("X", Xg(), 3)):
        one = MPO()
        one.from_pauli_sum(terms=[(1.0, f"{label}{site}")], length=L)
        got = np.real(vec.conj() @ one.to_matrix_mps_order() @ vec)
        assert abs(got - psi.expect(Observable(gate, site))) < 1e-13
        assert np.allclose(one.to_matrix(), _embed(L, {site: pa[label]}))
    capped = MPO()
    capped.from_pauli_sum(terms=more, length=L, max_bond_dim=2)
    assert max(t.shape[3] for t in capped.tensors) == 2
    empty = MPO()
    empty.from_pauli_sum(terms=[], length=3)
    assert np.allclose(empty.to_matrix(), 0)
    for bad, msg in (([(1.0, "Q0")], "Invalid term"), ([(1.0, "X9")], "out of bounds"), ([(1.0, "X0 Z0")], "twice"), ([(1.0, "X")], "Invalid term")):
        with pytest.raises(ValueError, match=msg):
            MPO().from_pauli_sum(terms=bad, length=L)
    with pytest.raises(ValueError, match="positive"):
        MPO().from_pauli_sum(terms=terms, length=0)
    assert np.allclose(MPO.identity(3).to_matrix(), np.eye(8))
    c = MPO()
    c.custom([t.transpose(2, 3, 0, 1) for t in ising.tensors])
    assert np.allclose(c.to_matrix(), ising.to_matrix())
    c.custom(ising.tensors, transpose=False)
    assert np.allclose(c.to_matrix(), ising.to_matrix())


def test_mps_inspection_helpers():
    """Host-side helpers of the reference's MPS that users call on ``result.output_state`` (mps.py:514-678, 901-959, 1539-1630),
    checked against dense linear algebra on a Haar state brought to a known canonical form."""
    from yaqs_amd.api import MPS

    L = 6
    psi = MPS(L, state="haar-random", pad=8, rng=np.random.default_rng(5))  # left-orthonormal sites, norm 1
    assert psi.bond_dimensions() == [2, 4, 8, 4, 2] and psi.get_total_bond() == 20 and psi.get_cost() == 2 ** 3 * 2 + 4 ** 3 * 2 + 8 ** 3
    assert psi.get_max_bond() == 8 and MPS(3, state="zeros").get_max_bond() == 2
    psi.check_if_valid_mps()
    assert abs(psi.norm() - 1) < 1e-13 and abs(psi.norm(L - 1) - 1) < 1e-13
    assert psi.check_canonical_form() == [L - 1]
    assert MPS(4, state="x+").check_canonical_form() == [0, 1, 2, 3]
    broken = MPS(L, tensors=[1.5 * t for t in psi.tensors])
    assert broken.check_canonical_form() == []
    other = MPS(L, state="haar-random", pad=4, rng=np.random.default_rng(6))
    assert abs(psi.scalar_product(other) - np.vdot(psi.to_vec(), other.to_vec())) < 1e-13
    assert abs(psi.scalar_product(other, 0) - np.vdot(psi.tensors[0], other.tensors[0])) < 1e-15
    # the centre is on the last site: the block (L-2, L-1) carries the Schmidt values of that cut
    vec = psi.to_vec().reshape(2, 2 ** (L - 1))  # (s_{L-1}, rest): site L-1 is the most significant index
    sv = np.linalg.svd(vec, compute_uv=False)
    spec = psi.get_schmidt_spectrum([L - 2, L - 1])
    assert spec.shape == (500,) and np.allclose(spec[:2], sv, atol=1e-13) and np.all(np.isnan(spec[2:]))
    pr = sv ** 2
    assert abs(psi.get_entropy([L - 2, L - 1]) - (-np.sum(pr * np.log(pr)))) < 1e-12
    prod = MPS(3, state="zeros")
    assert prod.get_entropy([0, 1]) == 0.0 and prod.get_schmidt_spectrum([0, 1])[0] == 1.0 and np.all(np.isnan(prod.get_schmidt_spectrum([0, 1])[1:]))
    with pytest.raises(AssertionError):
        psi.get_entropy([0, 2])


def test_host_side_shot_measurement():
    """``MPS.measure_shots`` / ``measure_single_shot`` (mps.py:1282-1382): outcome = sum(bit_i << i); basis states are deterministic
    in Z, x+ is deterministic in X, y+ in Y; a GHZ-like state gives only the two extreme outcomes with frequencies near 1/2."""
    from yaqs_amd.api import MPS

    rng = np.random.default_rng(0)
    assert MPS(5, state="basis", basis_string="10110").measure_shots(50, rng=rng) == {0b01101: 50}
    assert MPS(4, state="x+").measure_shots(20, basis="X", rng=rng) == {0: 20}
    assert MPS(4, state="x-").measure_shots(20, basis="x", rng=rng) == {15: 20}
    assert MPS(3, state="y+").measure_shots(20, basis="Y", rng=rng) == {0: 20}
    assert MPS(3, state="y-").measure_single_shot("Y", rng) == 7
    z = MPS(3, state="x+").measure_shots(4000, rng=rng)
    assert set(z) == set(range(8)) and all(abs(v / 4000 - 0.125) < 0.03 for v in z.values())
    a = np.zeros((2, 1, 2), dtype=complex); a[0, 0, 0] = a[1, 0, 1] = 2 ** -0.5
    m = np.zeros((2, 2, 2), dtype=complex); m[0, 0, 0] = m[1, 1, 1] = 1
    b = np.zeros((2, 2, 1), dtype=complex); b[0, 0, 0] = b[1, 1, 0] = 1
    ghz = MPS(4, tensors=[a, m, m, b]).measure_shots(2000, rng=rng)
    assert set(ghz) == {0, 15} and abs(ghz[0] / 2000 - 0.5) < 0.05
    with pytest.raises(ValueError, match="Invalid basis"):
        MPS(2, state="zeros").measure_shots(1, basis="W")


def test_pmc_summary_divides_the_gui_counter_by_the_xcds(tmp_path):
    """tools/pmc_summary.py: rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs; MfmaUtil = busy / (gui / 8 * 1024 SIMDs)."""
    import subprocess
    import sys

    src = tmp_path / "counter_collection.csv"
    rows = ["Kernel_Name,Counter_Name,Counter_Value"]
    for _ in range(2):  # two dispatches of one kernel
        rows += ["k,GRBM_GUI_ACTIVE,8000", "k,SQ_VALU_MFMA_BUSY_CYCLES,512000", "k,SQ_INSTS_VALU_MFMA_MOPS_F64,10",
                 "k,SQ_WAVE_CYCLES,1000", "k,SQ_ACTIVE_INST_VALU,250", "k,SQ_WAIT_INST_LDS,100", "k,SQ_INSTS_LDS,50", "k,SQ_LDS_BANK_CONFLICT,25"]
    src.write_text("\n".join(rows) + "\n")
    dst = tmp_path / "out.csv"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), str(src), str(dst)], capture_output=True, text=True, check=True).stdout
    # busy 1 024 000 / (16 000 / 8 * 1024) = 50 %
    assert "MfmaUtil 50.0%" in out and "VALU active 25.0%" in out and "LDS wait 10.0%" in out, out
    assert dst.read_text().strip().splitlines()[1].endswith(",50.00")
