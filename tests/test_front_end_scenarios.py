"""Front-end scenarios: what a user script gets back from ``Simulator.run`` for a handful of small, fully specified runs.

Expected numbers are OUTPUTS OF THE REFERENCE's own backends for the same inputs (``tools/make_golden.py scenarios`` ->
``tests/golden/front_end_scenarios.npz``) or closed-form values; the scenarios cover the situations the reference's end-to-end tests
exercise (noisy order-2 ensemble, observable order, final states, zero- and one-step order-2 runs, scheduled jumps, pair channels,
long-range Pauli noise, piecewise drives, what a Result carries).  Every scenario runs twice: on the CPU with the oracle-backed
stand-in engine (host logic only) and, marked ``gpu``, on the HIP engine.
"""
import os
import pickle

import numpy as np
import pytest

from conftest import GOLDEN
from standin import OracleEngine

G = np.load(os.path.join(GOLDEN, "front_end_scenarios.npz"))


@pytest.fixture(params=["standin", pytest.param("hip", marks=pytest.mark.gpu)])
def make_sim(request, monkeypatch):
    """Factory of Simulators: the stand-in answers the stage calls with the oracle and needs the Python schedule and an explicit batch."""
    import yaqs_amd.tjm as tjm_mod

    if request.param == "standin":
        monkeypatch.setattr(tjm_mod, "BatchEngine", OracleEngine)
        return lambda **kw: tjm_mod.Simulator(native=False, batch=kw.pop("batch", 32), **kw)
    pytest.importorskip("torch")
    return lambda **kw: tjm_mod.Simulator(**kw)


def _z_all(n):
    from yaqs_amd.api import Observable, Z

    return [Observable(Z(), s) for s in range(n)]


def test_noisy_order2_ensemble_with_final_time_sampling(make_sim):
    """Ten trajectories of a 5-site Ising chain with amplitude damping and dephasing, order 2, one column per observable: the
    ensemble means and every trajectory row equal the reference's (whose own end-to-end test pins these means to 2e-4)."""
    from yaqs_amd.api import AnalogSimParams, Hamiltonian, NoiseModel, State

    n = 5
    noise = NoiseModel([{"name": kind, "sites": [s], "strength": 0.1} for s in range(n) for kind in ("lowering", "pauli_z")])
    p = AnalogSimParams(observables=_z_all(n), elapsed_time=1, dt=0.1, num_traj=10, max_bond_dim=4, svd_threshold=1e-6, order=2,
                        sample_timesteps=False, random_seed=42)
    res = make_sim(batch=4).run(State(n, initial="zeros"), Hamiltonian.ising(n, J=1, g=0.5), p, noise)
    assert len(res.observables) == n
    for s in range(n):
        assert res.trajectories[s].shape == (10, 1) and res.expectation_values[s].shape == (1,)
        assert abs(res.expectation_values[s][0] - G["noisy_order2_mean"][s, 0]) < 1e-8
        assert np.allclose(res.trajectories[s][:, 0], G["noisy_order2_rows"][:, s, 0], atol=1e-8)


def test_results_come_back_in_the_users_observable_order(make_sim):
    """Observables given as (Z on 1, X on 0, Z on 0) are evaluated site-sorted inside the backend and handed back in the given order;
    the values agree with the dense final state the run returns, and with MPS.expect on it."""
    from yaqs_amd.api import AnalogSimParams, Hamiltonian, Observable, State, X, Z

    asked = [Observable(Z(), 1), Observable(X(), 0), Observable(Z(), 0)]
    p = AnalogSimParams(observables=asked, elapsed_time=0.1, dt=0.1, num_traj=1, get_state=True, sample_timesteps=False, preset="exact")
    res = make_sim().run(State(2, initial="zeros"), Hamiltonian.ising(2, J=1.0, g=0.7), p)
    sorted_rows, where = G["order_sorted_rows"], G["order_sorted_index"]
    for u, ob in enumerate(res.observables):
        assert (ob.gate.name, ob.sites) == (asked[u].gate.name, asked[u].sites)
        assert abs(res.expectation_values[u][-1] - sorted_rows[where[u]]) < 1e-10
    vec = res.output_state.mps.to_vec()
    assert abs(abs(np.vdot(vec, G["order_final_vec"])) - 1.0) < 1e-12
    for u, ob in enumerate(asked):  # site 0 is the least significant index of to_vec
        site = ob.sites if isinstance(ob.sites, int) else ob.sites[0]
        dense = np.kron(np.kron(np.eye(2 ** (1 - site)), ob.gate.matrix), np.eye(2 ** site))
        assert abs(np.real(np.vdot(vec, dense @ vec)) - res.expectation_values[u][-1]) < 1e-10
        assert abs(res.output_state.mps.expect(ob) - res.expectation_values[u][-1]) < 1e-10


def test_a_pair_observable_without_exchange_symmetry_and_a_dense_hamiltonian(make_sim):
    """X on site 0 times Z on site 1 pins the (s_i, s_i+1) index order of two-site observables; ``Hamiltonian(matrix=...)`` built
    from the dense matrix of an MPO drives the same evolution as that MPO."""
    from oracle import tjm_oracle as o  # checker only: dense matrix of the MPO
    from yaqs_amd.api import AnalogSimParams, BaseGate, Hamiltonian, Observable, State, X, Z

    xz = Observable(BaseGate("xz", np.kron(X().matrix, Z().matrix), interaction=2), [0, 1])
    p = AnalogSimParams(observables=[xz], elapsed_time=0.3, dt=0.1, num_traj=1, get_state=True, sample_timesteps=False, preset="exact")
    H = Hamiltonian.heisenberg(3, 1.0, 0.8, 0.5, 0.3)
    a = make_sim().run(State(3, initial="x+"), H, p)
    assert abs(a.expectation_values[0][-1] - a.output_state.mps.expect(xz)) < 1e-10
    b = make_sim().run(State(3, initial="x+"), Hamiltonian(matrix=o.mpo_to_matrix(H.tensors)), p)
    assert abs(b.expectation_values[0][-1] - a.expectation_values[0][-1]) < 1e-9


@pytest.mark.parametrize("order", [1, 2])
def test_final_state_of_a_closed_two_site_run(make_sim, order):
    """get_state on a closed run: the state vector at T = 1 (both drivers give the same one) against the reference's."""
    from yaqs_amd.api import AnalogSimParams, Hamiltonian, Observable, State, X

    p = AnalogSimParams(observables=[Observable(X(), 1)], elapsed_time=1, dt=0.1, num_traj=1, max_bond_dim=4, svd_threshold=1e-6, order=order,
                        get_state=True, sample_timesteps=False)
    res = make_sim().run(State(2, initial="zeros"), Hamiltonian.ising(2, J=1, g=0.5), p)
    assert isinstance(res.output_state, State)
    fid = abs(np.vdot(res.output_state.mps.to_vec(), G[f"closed2_order{order}_vec"])) ** 2
    assert abs(fid - 1.0) < 1e-10
    assert abs(res.expectation_values[0][0] - G[f"closed2_order{order}_x"][0]) < 1e-9


@pytest.mark.parametrize("T,sample", [(0.0, True), (0.0, False), (0.1, False), (0.1, True)])
def test_order2_runs_of_zero_and_one_step(make_sim, T, sample):
    """The order-2 driver at its edges: no step at all (the initial state is what is measured, and no half step of noise is applied
    to it), and exactly one step (final-only sampling returns the last column of the sampled run)."""
    from yaqs_amd.api import AnalogSimParams, Hamiltonian, NoiseModel, Observable, State, Z

    p = AnalogSimParams(observables=[Observable(Z(), 0)], dt=0.1, elapsed_time=T, num_traj=1, order=2, sample_timesteps=sample, get_state=True,
                        random_seed=0)
    res = make_sim().run(State(2, initial="zeros"), Hamiltonian.ising(2, J=1.0, g=0.5), p)
    assert res.output_state is not None
    assert np.allclose(np.asarray(res.expectation_values[0], dtype=float), G[f"short_T{T}_s{int(sample)}"], atol=1e-10)
    if T == 0.0:
        damped = NoiseModel([{"name": "lowering", "sites": [0], "strength": 1.0}])
        q = AnalogSimParams(observables=[Observable(Z(), 0)], dt=0.1, elapsed_time=0.0, num_traj=1, order=2, sample_timesteps=sample, random_seed=0)
        r = make_sim().run(State(1, initial="x+"), Hamiltonian(matrix=np.zeros((2, 2), dtype=complex)), q, damped)
        assert abs(np.asarray(r.expectation_values[0], dtype=float).reshape(-1)[0]) < 1e-10  # <Z> of |+> untouched


def test_scheduled_jumps_flip_qubits_at_their_times(make_sim):
    """Deterministic jumps under a vanishing Hamiltonian: an X at t = 0 acts before the first sample (also in a zero-length run with
    final-only sampling); an X at t = 0.5 flips <Z> from the sixth time point on; an XX on a pair at t = 0.2 flips both qubits, which
    <ZZ> cannot see and <Z_0> can."""
    from yaqs_amd.api import ZZ, AnalogSimParams, Hamiltonian, NoiseModel, Observable, State, Z

    vacuum1 = Hamiltonian(matrix=np.zeros((2, 2), dtype=complex))
    at_start = NoiseModel(scheduled_jumps=[{"time": 0.0, "sites": [0], "name": "x"}])
    p = AnalogSimParams(observables=[Observable(Z(), 0)], dt=0.1, elapsed_time=0.3, num_traj=1, order=1, get_state=True)
    r = make_sim().run(State(1, initial="zeros"), vacuum1, p, at_start)
    assert np.allclose(r.expectation_values[0], -1.0, atol=1e-10) and abs(r.output_state.mps.expect(Observable(Z(), 0)) + 1.0) < 1e-10
    p0 = AnalogSimParams(observables=[Observable(Z(), 0)], dt=0.1, elapsed_time=0.0, num_traj=1, order=1, sample_timesteps=False, get_state=True)
    r = make_sim().run(State(1, initial="zeros"), vacuum1, p0, at_start)
    assert abs(float(np.real(np.asarray(r.expectation_values[0]).reshape(-1)[0])) + 1.0) < 1e-10
    assert abs(r.output_state.mps.expect(Observable(Z(), 0)) + 1.0) < 1e-10

    mid = NoiseModel(scheduled_jumps=[{"time": 0.5, "sites": [0], "name": "x"}])
    r = make_sim().run(State(1, initial="zeros"), Hamiltonian.ising(1, 0.0, 0.0),
                       AnalogSimParams(elapsed_time=1.0, dt=0.1, num_traj=1, observables=[Observable(Z(), sites=0)]), noise_model=mid)
    assert np.allclose(r.expectation_values[0][:5], 1.0, atol=1e-10) and np.allclose(r.expectation_values[0][5:], -1.0, atol=1e-10)

    pair = NoiseModel(scheduled_jumps=[{"time": 0.2, "sites": [0, 1], "name": "crosstalk_xx"}])
    vacuum2 = Hamiltonian.ising(2, 0.0, 0.0)
    r = make_sim().run(State(2, initial="zeros"), vacuum2, AnalogSimParams(elapsed_time=0.4, dt=0.1, num_traj=1, observables=[Observable(ZZ(), sites=[0, 1])]),
                       noise_model=pair)
    assert np.allclose(r.expectation_values[0], 1.0, atol=1e-10)
    r = make_sim().run(State(2, initial="zeros"), vacuum2, AnalogSimParams(elapsed_time=0.4, dt=0.1, num_traj=1, observables=[Observable(Z(), sites=0)]),
                       noise_model=pair)
    assert np.allclose(r.expectation_values[0][:2], 1.0, atol=1e-10) and np.allclose(r.expectation_values[0][2:], -1.0, atol=1e-10)


@pytest.mark.parametrize("order", [1, 2])
@pytest.mark.parametrize("sample", [True, False])
def test_shapes_of_trajectories_and_means(make_sim, order, sample):
    from yaqs_amd.api import AnalogSimParams, Hamiltonian, State

    p = AnalogSimParams(observables=_z_all(5), elapsed_time=0.2, dt=0.2, num_traj=1, max_bond_dim=2, order=order, sample_timesteps=sample)
    res = make_sim().run(State(5, initial="zeros"), Hamiltonian.ising(5, J=1.0, g=0.5), p)
    cols = len(p.times) if sample else 1
    assert all(t.shape == (1, cols) for t in res.trajectories) and all(len(e) == cols for e in res.expectation_values)


@pytest.mark.parametrize("pair", ["crosstalk_xx", "lowering_two"])
def test_one_site_and_adjacent_pair_channels_together(make_sim, pair):
    """A Pauli flip channel on site 0 next to a two-site channel on (0, 1) - a Pauli pair or the non-Pauli double lowering - over 20
    trajectories: every row and the mean against the reference."""
    from yaqs_amd.api import AnalogSimParams, Hamiltonian, NoiseModel, Observable, State, Z

    p = AnalogSimParams(observables=[Observable(Z(), 0)], elapsed_time=0.1, dt=0.1, num_traj=20, max_bond_dim=8, order=2, sample_timesteps=False,
                        random_seed=42)
    noise = NoiseModel([{"name": "pauli_x", "sites": [0], "strength": 0.02}, {"name": pair, "sites": [0, 1], "strength": 0.01}])
    res = make_sim().run(State(2, initial="zeros"), Hamiltonian.ising(2, 1.0, 0.5), p, noise)
    assert np.allclose(res.trajectories[0][:, 0], G[f"pair_{pair}_rows"][:, 0, 0], atol=1e-8)
    assert abs(res.expectation_values[0][0] - G[f"pair_{pair}_mean"][0, 0]) < 1e-8


def test_long_range_pauli_crosstalk_runs_on_the_analog_path(make_sim):
    from yaqs_amd.api import AnalogSimParams, Hamiltonian, NoiseModel, Observable, State, Z

    noise = NoiseModel([{"name": "longrange_crosstalk_xy", "sites": [0, 2], "strength": 0.05}])
    p = AnalogSimParams(observables=[Observable(Z(), 0)], dt=0.1, elapsed_time=0.2, num_traj=2, random_seed=0)
    res = make_sim().run(State(3), Hamiltonian.ising(3, J=1.0, g=0.5), p, noise)
    assert np.allclose(res.expectation_values[0], G["longrange_mean"][0], atol=1e-8)


def test_piecewise_constant_drive_equals_its_pieces_run_in_sequence(make_sim):
    """Two X drives of different amplitude on one qubit, one-site TDVP: the piecewise run equals the second static run started from
    the final state of the first, and <Z> = cos(2 (0.1 * 1 + 0.1 * 2)); a piece that does not end on the time grid is rejected."""
    from yaqs_amd.api import AnalogSimParams, Hamiltonian, Observable, State

    weak, strong = (Hamiltonian.pauli(length=1, one_body=[(a, "X")]) for a in (1.0, 2.0))
    sim = make_sim()
    kw = dict(observables=[Observable("z", 0)], dt=0.1, order=1, tdvp_mode="1site", sample_timesteps=False)
    first = sim.run(State(1, initial="zeros"), weak, AnalogSimParams(elapsed_time=0.1, get_state=True, **kw))
    then = sim.run(first.output_state, strong, AnalogSimParams(elapsed_time=0.1, **kw))
    both = sim.run(State(1, initial="zeros"), Hamiltonian.piecewise([(weak, 0.1), (strong, 0.1)]), AnalogSimParams(elapsed_time=0.2, **kw))
    assert abs(both.expectation_values[0][-1] - then.expectation_values[0][-1]) < 1e-10
    assert abs(both.expectation_values[0][-1] - np.cos(0.6)) < 1e-9
    with pytest.raises(ValueError, match="integer multiple"):
        sim.run(State(1, initial="zeros"), Hamiltonian.piecewise([(weak, 0.15), (strong, 0.05)]),
                AnalogSimParams(elapsed_time=0.2, observables=[Observable("z", 0)], dt=0.1, order=1, tdvp_mode="1site"))


def test_what_an_analog_result_carries(make_sim):
    """Fields of a Result after a closed analog run: the caller's parameter object untouched, copies of the observables, averaged
    diagnostics on the time grid, no counts / noise model / correlator outputs, the final state on request; it survives pickling."""
    from yaqs_amd.api import AnalogSimParams, Hamiltonian, Observable, Result, State, Z

    mine = Observable(Z(), 0)
    p = AnalogSimParams(observables=[mine], elapsed_time=0.1, dt=0.1, num_traj=1000, get_state=True, sample_timesteps=False)
    res = make_sim().run(State(2, initial="zeros"), Hamiltonian.ising(2, J=1.0, g=0.5), p)
    assert isinstance(res, Result) and res.sim_params is p and p.num_traj == 1000
    assert res.observables is not p.observables and res.observables[0] is not mine and not hasattr(mine, "results")
    assert len(res.observables) == len(res.expectation_values) == len(res.trajectories) == 1
    assert res.output_state is not None and res.noise_model is None and res.counts is None
    assert res.multi_time_times is None and res.multi_time_results is None
    assert res.times is not None and len(res.runtime_cost) == len(res.max_bond) == len(res.total_bond) == len(res.times)
    back = pickle.loads(pickle.dumps(res))
    assert isinstance(back, Result) and isinstance(back.sim_params, AnalogSimParams)
    assert np.allclose(np.asarray(back.expectation_values[0]), np.asarray(res.expectation_values[0]))


@pytest.mark.parametrize("where", ["left_boundary", "center", "right_boundary"])
def test_pair_correlators_of_a_closed_chain_follow_the_pinned_series(make_sim, where):
    """<XX>, <YY>, <ZZ> of one neighbouring pair of a closed 4-site Ising chain at 21 time points, default preset, against the series
    the reference pins in its test suite (kept as data in tests/golden/reference_two_site_correlators.json; the reference's own
    tolerance is 1e-3)."""
    import json

    from yaqs_amd.api import XX, YY, ZZ, AnalogSimParams, Hamiltonian, Observable, State

    spec = json.load(open(os.path.join(GOLDEN, "reference_two_site_correlators.json")))["test_two_site_correlator_" + where]
    pair = spec["sites"]
    p = AnalogSimParams(observables=[Observable(g(), pair) for g in (XX, YY, ZZ)], elapsed_time=spec["elapsed_time"], dt=spec["dt"],
                        max_bond_dim=spec["max_bond_dim"], sample_timesteps=True)
    res = make_sim().run(State(spec["L"], initial="zeros"), Hamiltonian.ising(spec["L"], spec["J"], spec["g"]), p)
    for got, name in zip(res.expectation_values, ("xx", "yy", "zz")):
        assert np.allclose(got, np.array(spec[name]), atol=1e-3)


@pytest.mark.gpu
def test_what_a_circuit_result_carries():
    """Circuit runs (gate layers in place of a qiskit circuit): a shots-only run returns the histogram and no averaged diagnostics, an
    observables-only run the opposite; the noise model a noisy run sampled is on the Result, not on the caller's parameters."""
    pytest.importorskip("torch")
    from yaqs_amd.api import DigitalSimParams, NoiseModel, Observable, State, Z, ising_trotter_layers
    from yaqs_amd.tjm import Simulator

    layers = ising_trotter_layers(2, 1, 0.5, 0.1, 1)
    shots = DigitalSimParams(shots=16, max_bond_dim=4)
    r = Simulator().run(State(2, initial="zeros"), layers, shots)
    assert sum(r.counts.values()) == 16 and r.runtime_cost is None and r.max_bond is None and r.total_bond is None
    r = Simulator().run(State(2, initial="zeros"), layers, DigitalSimParams(observables=[Observable(Z(), 0)], num_traj=1, max_bond_dim=4))
    assert r.counts is None and r.runtime_cost is not None and r.max_bond is not None and r.total_bond is not None
    noisy = DigitalSimParams(shots=4, max_bond_dim=4, random_seed=0)
    r = Simulator().run(State(2, initial="zeros"), layers, noisy, NoiseModel([{"name": "pauli_z", "sites": [s], "strength": 1e-3} for s in range(2)]))
    assert r.noise_model is not None and not hasattr(noisy, "noise_model")
