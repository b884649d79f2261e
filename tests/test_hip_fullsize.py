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
