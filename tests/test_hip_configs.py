"""BASELINE.json configurations at their stated shapes that have no reference fixture of their own (the reference's circuit front end
needs qiskit, absent here): checked against the pinned oracle, which restates digital_tjm (digital/digital_tjm.py:636-749) and is
itself pinned to the reference's hand-built-layer fixtures (tests/golden/digital.npz, digital_mpo.npz)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _config5(dtype, ntraj, L=64, nlayers=20):
    import yaqs_amd.tjm as tjm
    from yaqs_amd.api import DigitalSimParams, MPS, NoiseModel, Observable, Z, ising_trotter_layers

    layers = ising_trotter_layers(L, 1.0, 0.5, 0.1, nlayers)
    noise = NoiseModel([{"name": name, "sites": [i], "strength": 0.001} for i in range(L) for name in ("pauli_x", "pauli_y", "pauli_z")])
    p = DigitalSimParams(observables=[Observable(Z(), s) for s in range(L)], num_traj=ntraj, max_bond_dim=512, svd_threshold=1e-9, random_seed=42)
    return tjm.Simulator(dtype=dtype).run_circuit(MPS(L, state="zeros"), layers, p, noise), layers


def test_config5_at_its_stated_shape_matches_the_oracle():
    """64-site noisy Trotter circuit (20 layers of rx / rzz-even / rzz-odd = 1260 two-qubit gates), depolarising noise gamma = 0.001 on
    the sites of every two-qubit gate, max_bond_dim 512, svd_threshold 1e-9 (BASELINE.json configs[4]).  64 trajectories
    through Simulator.run_circuit; the first two also through the oracle on the host: <Z_i> on every site within 1e-8 in fp64, within
    2e-3 with the complex64 library (the arithmetic the configuration is quoted in: fp32 rounding through 1260 gate updates; SURVEY 8d
    states the fp32 bar at ensemble level), bonds within the cap."""
    from oracle import tjm_oracle as o
    from yaqs_amd.api import ising_trotter_layers

    L, nlayers, check = 64, 20, 2
    olayers = [o.GateLayer(l.singles, l.even, l.odd, l.sample_points) for l in ising_trotter_layers(L, 1.0, 0.5, 0.1, nlayers)]
    on = [o.make_process(name, [i], 0.001) for i in range(L) for name in ("pauli_x", "pauli_y", "pauli_z")]
    op = o.DigitalParams(observables=[o.Obs(o.PAULI["z"], s) for s in range(L)], max_bond_dim=512, svd_threshold=1e-9, random_seed=42)
    cpu = [o.digital_tjm(t, o.MPSState.product(L, "zeros"), on, op, olayers) for t in range(check)]
    for dtype, tol in (("complex128", 1e-8), ("complex64", 2e-3)):  # fp32 rounding through 1260 gate updates and as many truncations
        res, _ = _config5(dtype, 64)
        z = np.stack([np.asarray(res.trajectories[u]) for u in range(L)])  # [L][ntraj][cols]
        assert np.all(np.isfinite(z))
        for t in range(check):
            ref = np.asarray(cpu[t][0], dtype=float)  # [L][cols]
            err = max(float(np.max(np.abs(z[u][t] - ref[u]))) for u in range(L))
            assert err < tol, (dtype, t, err)
        assert int(np.max(res.max_bond)) <= 512
        # the ensemble is physical: the depolarising channel only shrinks |<Z>|, and the noiseless value bounds it
        assert np.all(np.abs(np.asarray(res.expectation_values)) <= 1.0 + 1e-9)


def test_config1_tebd_variant_at_its_stated_shape_matches_the_reference():
    """BASELINE.json configs[0] in its TEBD form: the gate sequence of create_ising_circuit(10, 1, 0.5, 0.1, 10)
    (circuit_library.py:28-79) on 10 sites from |0...0>, max_bond_dim 16, measured after every time step - against outputs of the
    REFERENCE's digital_tjm (tests/golden/config1_tebd.npz, `tools/make_golden.py config1_tebd`): the closed system through
    Simulator.run_circuit, and the row's 8 trajectories with depolarising noise 0.01 after every gate (so that they differ) through
    the backend class.  <Z_i>, <X_3> at every sample point 1e-8; diagnostics exact."""
    import os

    from conftest import GOLDEN
    from yaqs_amd.api import DigitalSimParams, MPS, NoiseModel, Observable, X, Z, ising_trotter_layers
    from yaqs_amd.engine import BatchEngine
    from yaqs_amd.tjm import DigitalBatch, Simulator

    g = np.load(os.path.join(GOLDEN, "config1_tebd.npz"))
    L, steps = 10, 10
    obs = [Observable(Z(), s) for s in range(L)] + [Observable(X(), 3)]
    layers = ising_trotter_layers(L, 1.0, 0.5, 0.1, steps, sample_each=True)
    p = DigitalSimParams(observables=obs, num_traj=1, max_bond_dim=16, svd_threshold=1e-9, random_seed=5, sample_layers=True, num_mid_measurements=steps)
    res = Simulator().run_circuit(MPS(L, state="zeros"), layers, p, None)
    got = np.stack([np.asarray(res.trajectories[k])[0] for k in range(len(obs))])  # [n_obs][cols], in the caller's order
    # the backend returns its rows in the reference's worker order - sorted by site, X_3 behind Z_3 (simulation_parameters.py) - and
    # Result maps them back to the caller's: Z_s sits in row s (s <= 3) / s + 1 (s >= 4) of the fixture, X_3 in row 4
    row = [s_ if s_ <= 3 else s_ + 1 for s_ in range(L)] + [4]
    assert got.shape == g["closed_results"][0].shape
    assert np.abs(got - g["closed_results"][0][row]).max() < 1e-8
    noise = NoiseModel([{"name": n, "sites": [i], "strength": 0.01} for i in range(L) for n in ("pauli_x", "pauli_y", "pauli_z")])
    identity_mpo = [np.eye(2, dtype=np.complex128).reshape(2, 2, 1, 1)] * L
    e = BatchEngine(L, 16, 8, identity_mpo)
    try:
        r, d = DigitalBatch(e, p, noise).run(list(range(8)), MPS(L, state="zeros"), layers)
    finally:
        e.close()
    assert np.abs(r - g["noisy_results"]).max() < 1e-8
    assert np.array_equal(d, g["noisy_diag"])
