"""GPU parity at BASELINE.json's stated sizes, against outputs of the REFERENCE itself.

``tests/golden/fullsize.npz`` holds what the reference (imported in the build container by ``tools/make_golden.py fullsize``)
returns for ONE order-1 TJM step of configs 2, 3 and 4 at full size: per-site <Z>, the jump probability dp, the final bond
dimensions and the diagnostics.  Nothing large is stored: the chi-saturated Haar input is regenerated from its seed by the same
host-side builder the fixture script used.  Tolerance 1e-8 on <Z> and dp (the fp64 bound of ``north_star``), bonds exact.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

TOL = 1e-8


def _inputs(L, chi):
    from yaqs_amd import api

    st = api.MPS(L, state="haar-random", pad=chi, rng=np.random.default_rng(1))
    st.normalize("B")
    return api, [np.asarray(t, dtype=np.complex128) for t in st.tensors]


def _one_step(L, chi, mpo, proc, gamma, dt, tdvp_mode, tensors, trajs, dtype="complex128"):
    """tdvp -> dissipate -> stochastic -> <Z_i> through the stage entry points of the C ABI, one slot per trajectory."""
    from yaqs_amd.api import NoiseModel, is_pauli
    from yaqs_amd.engine import BatchEngine
    from yaqs_amd.tjm import trajectory_uniforms

    B = len(trajs)
    e = BatchEngine(L, chi, B, mpo.tensors, dtype=dtype)
    e.set_params(dt=dt, svd_threshold=1e-12, max_bond_dim=chi, krylov_tol=1e-10 if dtype == "complex128" else 1e-6, tdvp_mode=tdvp_mode)
    noise = NoiseModel([{"name": proc, "sites": [i], "strength": gamma} for i in range(L)])
    e.set_noise(noise.processes, [is_pauli(q) for q in noise.processes])
    e.load_state(tensors)
    e.tdvp()
    e.dissipate(dt)
    e.set_uniforms(np.stack([trajectory_uniforms(42, int(t), 2) for t in trajs]))
    jumped, dp = e.stochastic(dt)
    M = e.site_moments()
    z = (M[:, :, 0, 0] - M[:, :, 1, 1]).real.T  # [B][L]
    bonds = e.bond_dims()
    assert not e.capacity_overflow()
    e.close()
    return z, dp, jumped, bonds


def _check(name, z, dp, jumped, bonds):
    g = np.load(os.path.join(GOLDEN, "fullsize.npz"))
    assert np.allclose(dp, g[name + "_dp"], atol=TOL), (name, dp, g[name + "_dp"])
    assert np.array_equal(jumped.astype(bool), g[name + "_u0"] < g[name + "_dp"]), name
    assert np.array_equal(bonds, g[name + "_bonds"]), name
    err = np.abs(z - g[name + "_z"]).max()
    assert err < TOL, (name, err)
    # diagnostics of the reference (mps.py:549-591): sum chi^3, largest bond, sum chi over the inner bonds
    inner = bonds[:, 1:-1].astype(np.float64)
    assert np.array_equal((inner ** 3).sum(axis=1), g[name + "_diag"][:, 0]), name
    assert np.array_equal(inner.sum(axis=1), g[name + "_diag"][:, 2]), name


def test_config2_full_size_step_matches_the_reference():
    """64-site dissipative TFIM, chi = 128 saturated (BASELINE.json configs[1]): one trajectory that does not jump and one that does."""
    g = np.load(os.path.join(GOLDEN, "fullsize.npz"))
    api, t = _inputs(64, 128)
    out = _one_step(64, 128, api.MPO.ising(64, 1.0, 0.5), "pauli_z", 0.1, 0.1, "2site", t, list(g["cfg2_traj"]))
    assert out[2].tolist() == [0, 1]
    _check("cfg2", *out)


def test_config4_full_size_one_site_tdvp_step_matches_the_reference():
    """32-site long-range Ising (exponential-sum MPO), one-site TDVP at chi = 256 (configs[3]): 512 x 256 Householder panels and
    project_bond at 256."""
    g = np.load(os.path.join(GOLDEN, "fullsize.npz"))
    api, t = _inputs(32, 256)
    mpo = api.MPO.long_range_ising(32, [0.8792, 0.1208], [0.0717, 0.5136], 0.5)
    _check("cfg4", *_one_step(32, 256, mpo, "pauli_z", 0.05, 0.05, "1site", t, list(g["cfg4_traj"])))


def test_config3_full_size_step_matches_the_reference():
    """128-site XXZ chain with amplitude damping at chi = 256 (configs[2], in the reference's complex128): 512 x 512 two-site splits,
    the d x d dissipator path, environment lists at full length."""
    g = np.load(os.path.join(GOLDEN, "fullsize.npz"))
    if "cfg3_z" not in g:
        pytest.skip("cfg3 fixture not generated")
    api, t = _inputs(128, 256)
    _check("cfg3", *_one_step(128, 256, api.MPO.heisenberg(128, 1.0, 1.0, 0.5, 0.0), "lowering", 0.05, 0.05, "2site", t, list(g["cfg3_traj"])))


# ---- the state bench.py's timed region is in: ten CONSECUTIVE steps of config 2 -------------------------------------------------
STEADY = os.path.join(GOLDEN, "fullsize_steady.npz")
NO_CERT = os.environ.get("TJM_NO_CERT_DISSIPATION") is not None


def _steady_engine(g, sel=None):
    from yaqs_amd.api import NoiseModel, is_pauli
    from yaqs_amd.engine import BatchEngine

    api, t = _inputs(64, 128)
    trajs = [int(x) for x in g["traj"]]
    if sel is not None:
        trajs = [trajs[k] for k in sel]
    e = BatchEngine(64, 128, len(trajs), api.MPO.ising(64, 1.0, 0.5).tensors)
    e.set_params(dt=0.1, svd_threshold=1e-12, max_bond_dim=128, krylov_tol=float(g["krylov_tol"]) if "krylov_tol" in g else 1e-10, tdvp_mode="2site")
    noise = NoiseModel([{"name": "pauli_z", "sites": [i], "strength": 0.1} for i in range(64)])
    e.set_noise(noise.processes, [is_pauli(q) for q in noise.processes])
    e.load_state(t)
    return e, trajs


def _steady_counters(stats):
    if NO_CERT:
        assert stats["certified_dissipations"] == 0 and stats["certified_jumps"] == 0, stats
    else:  # the path that serves most trajectory-steps of the driver's bench run must be the one under test
        assert stats["certified_dissipations"] > 0 and stats["certified_jumps"] > 0, stats


def _steady_stages(g, sel, check=True):
    """tdvp -> dissipate -> stochastic, ten times, through the stage entry points; with ``check`` every step against the fixture."""
    from yaqs_amd.tjm import trajectory_uniforms

    e, trajs = _steady_engine(g, sel)
    steps = int(g["steps"])
    u = np.stack([trajectory_uniforms(42, t, 2 * steps + 4) for t in trajs])
    pos = np.zeros(len(trajs), dtype=np.int64)
    ar = np.arange(len(trajs))
    for k in range(steps):
        e.tdvp()
        e.dissipate(0.1)
        e.set_uniforms(np.stack([u[ar, pos], u[ar, pos + 1]], axis=1))
        jumped, dp = e.stochastic(0.1)
        pos += 1 + jumped
        if check:
            assert np.allclose(dp, g["dp"][sel, k], atol=TOL), (k, dp, g["dp"][sel, k])
            assert np.array_equal(jumped.astype(int), g["jumped"][sel, k]), (k, jumped, g["jumped"][sel, k])
            assert np.array_equal(e.bond_dims(), g["bonds"][sel, k]), k
    M = e.site_moments()
    z = (M[:, :, 0, 0] - M[:, :, 1, 1]).real.T
    stats = e.stats()
    assert not e.capacity_overflow()
    e.close()
    return z, stats


@pytest.mark.skipif(not os.path.exists(STEADY), reason="tests/golden/fullsize_steady.npz not generated")
def test_config2_ten_consecutive_steps_match_the_reference_through_the_stage_entry_points():
    """``tools/make_golden.py fullsize_steady[_final]``: the REFERENCE's analog_tjm_1 on config 2 (L = 64, chi = 128 Haar-saturated,
    krylov_tol 1e-10) for ten consecutive steps, three trajectories, every one of which jumps several times - also after step 6, where
    the certified scalar dissipation and the in-place jumps of the engine take over.  Per step: the jump
    probability dp (1e-8), the jump decision and the whole bond table (exact); at the end <Z_i> on every site (1e-8).
    Then the same trajectories in other batches (one alone, the other two together): whether a trajectory certifies - only part of
    the thirty trajectory-steps do - is its own affair, so the rows are bit-identical."""
    g = np.load(STEADY)
    sel = list(range(len(g["traj"])))
    z, stats = _steady_stages(g, sel)
    err = np.abs(z - g["z"][:, :, -1]).max()
    assert err < TOL, err
    assert g["jumped"][:, 6:].sum() > 0
    _steady_counters(stats)
    if not NO_CERT:
        assert 0 < stats["certified_dissipations"] < len(sel) * int(g["steps"]), stats  # the partial regime
    if not (os.environ.get("TJM_NO_DIRECT_HEFF") or os.environ.get("TJM_GEMM_16X16") or os.environ.get("TJM_NO_IDENTITY_CHANNELS")):
        # round 5: the H_eff applies of the bulk sites run in their direct form (no T2 tensor) - the path this fixture pins
        assert stats["direct_applies"] > 0.5 * stats["matvecs"], stats
    z0, _ = _steady_stages(g, sel[:1], check=False)
    z12, _ = _steady_stages(g, sel[1:], check=False)
    assert np.array_equal(z0[0], z[0]) and np.array_equal(z12, z[1:]), (np.abs(z0[0] - z[0]).max(), np.abs(z12 - z[1:]).max())


@pytest.mark.skipif(not os.path.exists(STEADY), reason="tests/golden/fullsize_steady.npz not generated")
def test_config2_ten_consecutive_steps_match_the_reference_through_the_c_driver():
    """The same run in ONE call of tjm_engine_run (the reference's random streams inside the library): final <Z_i> (1e-8) and the
    diagnostics row of the final time (exact)."""
    g = np.load(STEADY)
    e, trajs = _steady_engine(g)
    steps = int(g["steps"])
    zmat = np.diag([1.0, -1.0]).astype(np.complex128)
    res, diag = e.run(order=1, n_times=steps + 1, sample_timesteps=False, has_noise=True, seed=42, traj_indices=trajs,
                      observables=[(s, zmat) for s in range(64)])
    stats = e.stats()
    e.close()
    err = np.abs(res[:, :, 0] - g["z"][:, :, -1]).max()
    assert err < TOL, err
    assert np.array_equal(diag[:, :, 0], g["diag"][:, :, -1]), (diag[:, :, 0], g["diag"][:, :, -1])
    _steady_counters(stats)


@pytest.mark.skipif(not os.path.exists(STEADY) or NO_CERT, reason="fixture not generated / already the child run")
def test_config2_ten_consecutive_steps_also_without_the_certified_dissipation():
    """TJM_NO_CERT_DISSIPATION is read once per process: the two tests above once more in a child process with the switch set (every
    dissipation by the reference's 2 (L - 1) SVD shifts, every jump through the QR walk and the SVD sweep back)."""
    import subprocess
    import sys

    env = dict(os.environ, TJM_NO_CERT_DISSIPATION="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k", "config2_ten_consecutive and not also_without"],
                         env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__)))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "2 passed" in out.stdout, out.stdout[-2000:]


# ---- the same ten steps with the EXACT parameters of the timed region: krylov_tol 1e-4 (bench.py), eight trajectories ----
STEADY4 = os.path.join(GOLDEN, "fullsize_steady_tol4.npz")


@pytest.mark.skipif(not os.path.exists(STEADY4), reason="tests/golden/fullsize_steady_tol4.npz not generated")
def test_config2_ten_steps_at_the_bench_krylov_tolerance_match_the_reference():
    """``tools/make_golden.py fullsize_steady_bench``: the fixture above once more with ``krylov_tol = 1e-4`` - the value bench.py times
    (at 1e-10 a Lanczos call runs to ~12 vectors, at 1e-4 to ~5: another stopping point of the same recurrence, so the iteration count of
    every one of the 2 x 63 x 10 calls per trajectory has to agree with the reference's for the results to agree) - and EIGHT
    trajectories (80 trajectory-steps, 42 of them with a jump).  The stopping rule compares an a-posteriori estimate with 1e-4; the
    engine evaluates the same estimate from the same recurrence, so the bar stays the fp64 one: dp 1e-8 at every step, jump decisions
    and bond tables exact, final <Z_i> 1e-8.  Through the stage entry points and through the one-call C driver."""
    g = np.load(STEADY4)
    assert float(g["krylov_tol"]) == 1e-4 and len(g["traj"]) >= 8
    sel = list(range(len(g["traj"])))
    z, stats = _steady_stages(g, sel)
    err = np.abs(z - g["z"][:, :, -1]).max()
    assert err < TOL, err
    _steady_counters(stats)
    e, trajs = _steady_engine(g)
    zmat = np.diag([1.0, -1.0]).astype(np.complex128)
    res, diag = e.run(order=1, n_times=int(g["steps"]) + 1, sample_timesteps=False, has_noise=True, seed=42, traj_indices=trajs,
                      observables=[(s, zmat) for s in range(64)])
    e.close()
    assert np.abs(res[:, :, 0] - g["z"][:, :, -1]).max() < TOL
    assert np.array_equal(diag[:, :, 0], g["diag"][:, :, -1])


# ---- configs 4 and 3 in the state their bench lines time: consecutive steps at krylov_tol 1e-4 (round 6) ---------------------------
STEADY_CFG = {
    # name: (L, chi, MPO builder, noise process, gamma, dt, tdvp_mode)
    "cfg4": (32, 256, lambda api: api.MPO.long_range_ising(32, [0.8792, 0.1208], [0.0717, 0.5136], 0.5), "pauli_z", 0.05, 0.05, "1site"),
    "cfg3": (128, 256, lambda api: api.MPO.heisenberg(128, 1.0, 1.0, 0.5, 0.0), "lowering", 0.05, 0.05, "2site"),
}


def _cfg_fixture(cfg):
    path = os.path.join(GOLDEN, f"fullsize_steady_{cfg}.npz")
    if not os.path.exists(path):
        pytest.skip(f"tests/golden/fullsize_steady_{cfg}.npz not generated")
    return np.load(path)


def _cfg_engine(cfg, g, sel=None, dtype="complex128"):
    from yaqs_amd.api import NoiseModel, is_pauli
    from yaqs_amd.engine import BatchEngine

    L, chi, make_mpo, proc, gamma, dt, mode = STEADY_CFG[cfg]
    api, t = _inputs(L, chi)
    trajs = [int(x) for x in g["traj"]]
    if sel is not None:
        trajs = [trajs[k] for k in sel]
    e = BatchEngine(L, chi, len(trajs), make_mpo(api).tensors, dtype=dtype)
    e.set_params(dt=dt, svd_threshold=1e-12, max_bond_dim=chi, krylov_tol=float(g["krylov_tol"]), tdvp_mode=mode)
    noise = NoiseModel([{"name": proc, "sites": [i], "strength": gamma} for i in range(L)])
    e.set_noise(noise.processes, [is_pauli(q) for q in noise.processes])
    e.load_state(t)
    return e, trajs, dt


def _cfg_stages(cfg, g, sel, check=True, tol=TOL):
    """tdvp -> dissipate -> stochastic for every step of the fixture through the stage entry points; with ``check`` the jump
    probability (tol), the jump decision and the whole bond table (exact) of every step against the reference."""
    from yaqs_amd.tjm import trajectory_uniforms

    e, trajs, dt = _cfg_engine(cfg, g, sel)
    steps = int(g["steps"])
    u = np.stack([trajectory_uniforms(42, t, 2 * steps + 4) for t in trajs])
    pos = np.zeros(len(trajs), dtype=np.int64)
    ar = np.arange(len(trajs))
    for k in range(steps):
        e.tdvp()
        e.dissipate(dt)
        e.set_uniforms(np.stack([u[ar, pos], u[ar, pos + 1]], axis=1))
        jumped, dp = e.stochastic(dt)
        pos += 1 + jumped
        if check:
            assert np.allclose(dp, g["dp"][sel, k], atol=tol), (cfg, k, dp, g["dp"][sel, k])
            assert np.array_equal(jumped.astype(int), g["jumped"][sel, k]), (cfg, k, jumped, g["jumped"][sel, k])
            assert np.array_equal(e.bond_dims(), g["bonds"][sel, k]), (cfg, k)
    M = e.site_moments()
    z = (M[:, :, 0, 0] - M[:, :, 1, 1]).real.T
    stats = e.stats()
    assert not e.capacity_overflow()
    e.close()
    return z, stats


def test_config4_ten_consecutive_steps_match_the_reference():
    """``tools/make_golden.py fullsize_steady_cfg4``: the reference's analog_tjm_1 on BASELINE config 4 (L = 32, chi = 256
    Haar-saturated, long-range Ising MPO, one-site TDVP, pauli_z 0.05, krylov_tol 1e-4 as ``bench.py --config 4``) for ten consecutive
    steps, three trajectories (one jumps at steps 1, 2 and 5, one at step 2, one at step 5).  Per step dp (1e-8), jump decisions and
    bond tables (exact), final <Z_i> (1e-8), through the stage entry points and through the one-call C driver.  At chi = 256 the
    dissipation certificate runs on chol_pd_blocked_kernel (the packed triangle of a 256 x 256 Gram matrix does not fit the LDS): the
    counters say that it ran and that it certified part of the thirty trajectory-steps - the path that bought config 4 its 25 %."""
    g = _cfg_fixture("cfg4")
    sel = list(range(len(g["traj"])))
    assert g["jumped"].sum() >= 3 and g["jumped"][:, 1:].sum() > 0
    z, stats = _cfg_stages("cfg4", g, sel)
    err = np.abs(z - g["z"][:, :, -1]).max()
    assert err < TOL, err
    if NO_CERT:
        assert stats["certified_dissipations"] == 0, stats
    else:
        assert stats["certificate_tests_blocked_cholesky"] > 0, stats
        assert stats["certified_dissipations"] > 0, stats
    e, trajs, _ = _cfg_engine("cfg4", g)
    zmat = np.diag([1.0, -1.0]).astype(np.complex128)
    res, diag = e.run(order=1, n_times=int(g["steps"]) + 1, sample_timesteps=False, has_noise=True, seed=42, traj_indices=trajs,
                      observables=[(s, zmat) for s in range(32)])
    e.close()
    assert np.abs(res[:, :, 0] - g["z"][:, :, -1]).max() < TOL
    assert np.array_equal(diag[:, :, 0], g["diag"][:, :, -1])
    # batch independence: the first trajectory alone gives the same row bit for bit
    z0, _ = _cfg_stages("cfg4", g, sel[:1], check=False)
    assert np.array_equal(z0[0], z[0]), np.abs(z0[0] - z[0]).max()


@pytest.mark.skipif(NO_CERT, reason="already the child run")
def test_config4_ten_consecutive_steps_also_without_the_certified_dissipation():
    """The test above once more in a child process with TJM_NO_CERT_DISSIPATION=1 (read once per process): every dissipation by the
    reference's 2 (L - 1) SVD shifts."""
    import subprocess
    import sys

    _cfg_fixture("cfg4")
    env = dict(os.environ, TJM_NO_CERT_DISSIPATION="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k", "config4_ten_consecutive and not also_without"],
                         env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__)))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "1 passed" in out.stdout, out.stdout[-2000:]


def test_config3_three_consecutive_steps_match_the_reference():
    """``tools/make_golden.py fullsize_steady_cfg3``: the reference's analog_tjm_1 on BASELINE config 3 in complex128 (L = 128,
    chi = 256 Haar-saturated, XXZ, amplitude damping 0.05 on every site - non-Pauli jumps: the real probability sweep, the d x d
    dissipator, 512 x 512 two-site splits -, krylov_tol 1e-4) for three consecutive steps, two trajectories: from the second step on
    bonds and gauges are the run's own.  dp 1e-8 per step, jump decisions and bond tables exact, final <Z_i> 1e-8; stage entry
    points and the one-call C driver."""
    g = _cfg_fixture("cfg3")
    sel = list(range(len(g["traj"])))
    assert g["jumped"].sum() >= 2
    z, stats = _cfg_stages("cfg3", g, sel)
    err = np.abs(z - g["z"][:, :, -1]).max()
    assert err < TOL, err
    e, trajs, _ = _cfg_engine("cfg3", g)
    zmat = np.diag([1.0, -1.0]).astype(np.complex128)
    res, diag = e.run(order=1, n_times=int(g["steps"]) + 1, sample_timesteps=False, has_noise=True, seed=42, traj_indices=trajs,
                      observables=[(s, zmat) for s in range(128)])
    e.close()
    assert np.abs(res[:, :, 0] - g["z"][:, :, -1]).max() < TOL
    assert np.array_equal(diag[:, :, 0], g["diag"][:, :, -1])


def test_config3_three_consecutive_steps_in_complex64_follow_the_reference():
    """BASELINE quotes config 3 in fp32 and ``bench.py --config 3`` times the complex64 library: the three consecutive steps of the
    fixture above on ``libtjm_hip_f32.so`` (512 x 512 splits on the grouped four-block Jacobi kernel of round 6, Lanczos and
    environments in fp32) against the REFERENCE's complex128 outputs.  fp32 accuracy over 128 sites and three steps: dp of every step to
    2e-3, the jump decision wherever the draw is not that close to dp, final <Z_i> to 5e-3 for the trajectories whose decisions all
    coincided; bonds at the cap in the saturated bulk (towards the chain ends the complex64 build drops Schmidt values below the fp32
    resolution, TJM_RANK_TOL, so a bond may come out a few below the reference's)."""
    from yaqs_amd.tjm import trajectory_uniforms

    g = _cfg_fixture("cfg3")
    sel = list(range(len(g["traj"])))
    e, trajs, dt = _cfg_engine("cfg3", g, sel, dtype="complex64")
    steps = int(g["steps"])
    u = np.stack([trajectory_uniforms(42, t, 2 * steps + 4) for t in trajs])
    pos = np.zeros(len(trajs), dtype=np.int64)
    ar = np.arange(len(trajs))
    alive = np.ones(len(trajs), dtype=bool)  # decisions coincided so far
    for k in range(steps):
        e.tdvp()
        e.dissipate(dt)
        e.set_uniforms(np.stack([u[ar, pos], u[ar, pos + 1]], axis=1))
        jumped, dp = e.stochastic(dt)
        want = g["jumped"][sel, k].astype(bool)
        assert np.abs(dp - g["dp"][sel, k])[alive].max(initial=0.0) < 2e-3, (k, dp, g["dp"][sel, k])
        safe = np.abs(u[ar, pos] - g["dp"][sel, k]) > 5e-3
        assert np.array_equal(jumped.astype(bool)[alive & safe], want[alive & safe]), (k, jumped, want)
        alive &= jumped.astype(bool) == want
        pos += 1 + jumped
    assert alive.any()
    M = e.site_moments()
    z = (M[:, :, 0, 0] - M[:, :, 1, 1]).real.T
    bonds = e.bond_dims()
    assert not e.capacity_overflow()
    e.close()
    assert np.abs(z[alive] - g["z"][sel][alive][:, :, -1]).max() < 5e-3
    ref_bonds = g["bonds"][sel, steps - 1][alive]
    L = 128
    assert np.array_equal(bonds[alive][:, L // 4: 3 * L // 4], ref_bonds[:, L // 4: 3 * L // 4])
    assert np.all(bonds[alive] <= ref_bonds) and np.all(bonds[alive] >= ref_bonds - 8)
