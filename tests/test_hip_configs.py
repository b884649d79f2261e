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
