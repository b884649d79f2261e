"""Host logic above the C ABI on the CPU: ``Simulator.run`` / ``TrajectoryBatch`` with the oracle-backed stand-in engine of
``tests/standin.py`` in place of the HIP engine (monkeypatched; the product itself has no CPU path).  What is checked here is the
Python schedule - drivers of both orders, uniform cursors, storage grown on demand with roll-back, result assembly, run-context
validation - against the oracle's own drivers; the kernels are the business of the ``-m gpu`` tests."""
import numpy as np
import pytest

from oracle import tjm_oracle as o
from standin import OracleEngine

Z = o.PAULI["z"]


@pytest.fixture()
def sim(monkeypatch):
    import yaqs_amd.tjm as tjm_mod

    monkeypatch.setattr(tjm_mod, "BatchEngine", OracleEngine)
    OracleEngine.instances.clear()
    return tjm_mod.Simulator


def _oracle_rows(L, noise, kw, mpo, initial, n):
    on = [o.make_process(q["name"], q["sites"], q["strength"], matrix=q.get("matrix"), factors=q.get("factors")) for q in noise.processes]
    op = o.Params(observables=[o.Obs(Z, s) for s in range(L)], **kw)
    return [o.run_trajectory(t, o.MPSState([x.copy() for x in initial], 0), on, op, mpo) for t in range(n)]


@pytest.mark.parametrize("order", [1, 2])
@pytest.mark.parametrize("sample", [True, False])
def test_python_schedule_with_growing_storage_matches_the_oracle_drivers(sim, order, sample):
    """8 sites from a product state, max_bond_dim 16: the run starts at capacity 8 and must hand over to 16 mid-run (roll-back of
    the clipped step, cursors travelling with the trajectories, chunks of 3 trajectories)."""
    from yaqs_amd.api import AnalogSimParams, MPO, MPS, NoiseModel, Observable, Z as Zg

    L = 8
    noise = NoiseModel([{"name": n, "sites": [i], "strength": 0.15} for i in range(L) for n in ("lowering", "pauli_x")])
    kw = dict(elapsed_time=1.2, dt=0.2, max_bond_dim=16, svd_threshold=1e-14, krylov_tol=1e-12, order=order, sample_timesteps=sample, random_seed=21)
    p = AnalogSimParams(observables=[Observable(Zg(), s) for s in range(L)], num_traj=5, **kw)
    mpo = MPO.ising(L, 1.0, 1.0)
    st = MPS(L, state="x+")
    res = sim(batch=3, native=False).run(st, mpo, p, noise)
    st.normalize("B")
    rows = _oracle_rows(L, noise, kw, [np.asarray(w) for w in mpo.tensors], st.tensors, 5)
    for t in range(5):
        for s_ in range(L):
            assert np.allclose(res.trajectories[s_][t], rows[t][0][s_], atol=1e-9), (t, s_)
    caps = sorted({e.chi_max for e in OracleEngine.instances})
    assert caps[0] == 8 and caps[-1] == 16, caps           # the storage ladder was climbed
    assert all(e.closed for e in OracleEngine.instances)    # and every engine handed back
    d = np.mean([rows[t][1] for t in range(5)], axis=0)
    assert np.allclose(res.total_bond, d[2]) and np.allclose(res.max_bond, d[1])


def test_schmidt_spectrum_entropy_and_pvm_through_the_front_end(sim):
    """trajectories[u] of a schmidt_spectrum observable is [traj, T, 500], expectation_values[u] the concatenation over trajectories
    (mps.py:1211, result.py:127-139); entropy and the projector stay scalars - none of them NaN means."""
    from yaqs_amd.api import AnalogSimParams, MPO, MPS, NoiseModel, Observable, Z as Zg

    L = 6
    noise = NoiseModel([{"name": "lowering", "sites": [s], "strength": 0.1} for s in range(L)])
    obs = [Observable(Zg(), 0), Observable("schmidt_spectrum", [2, 3]), Observable("entropy", [2, 3]), Observable(Zg(), 5)]
    kw = dict(elapsed_time=0.3, dt=0.1, max_bond_dim=8, svd_threshold=1e-12, krylov_tol=1e-12, order=1, sample_timesteps=True, random_seed=5)
    res = sim(batch=2, native=False).run(MPS(L, state="x+"), MPO.ising(L, 1.0, 0.5), AnalogSimParams(observables=obs, num_traj=3, **kw), noise)
    spec = res.trajectories[1]
    assert spec.shape == (3, 4, 500) and res.expectation_values[1].shape == (3 * 4 * 500,)
    on = [o.make_process("lowering", [s], 0.1) for s in range(L)]
    op = o.Params(observables=[o.Obs(Z, 0)], get_state=False, **kw)
    for t in range(3):
        for j in range(4):
            sv = spec[t, j][~np.isnan(spec[t, j])]
            pr = sv ** 2 / np.sum(sv ** 2)
            assert abs(-np.sum(pr * np.log(pr + np.finfo(float).tiny)) - res.trajectories[2][t, j]) < 1e-10
        r, _, _ = o.run_trajectory(t, o.MPSState.product(L, "x+"), on, op, o.ising_mpo(L, 1.0, 0.5))
        assert np.allclose(res.trajectories[0][t], r[0], atol=1e-9)
    assert np.isfinite(res.expectation_values[0]).all() and np.isfinite(res.expectation_values[2]).all()


def test_one_site_tdvp_with_a_pair_channel_is_not_confined_to_the_initial_bonds(sim):
    """tdvp_mode='1site' keeps the bonds of the sweep, but an adjacent non-Pauli pair channel is applied through a truncated merged
    split: the engine must be allowed to grow (engine_bond_caps(can_grow=True)) and the result is the oracle's."""
    from yaqs_amd.api import AnalogSimParams, MPO, MPS, NoiseModel, Observable, Z as Zg
    from yaqs_amd.tjm import _noise_can_grow_bonds, engine_bond_caps

    L = 4
    noise = NoiseModel([{"name": "lowering_two", "sites": [1, 2], "strength": 0.4}, {"name": "pauli_x", "sites": [0], "strength": 0.2}])
    assert _noise_can_grow_bonds(noise) and not _noise_can_grow_bonds(NoiseModel([{"name": "pauli_x", "sites": [0], "strength": 0.2}]))
    kw = dict(elapsed_time=0.3, dt=0.1, max_bond_dim=4, svd_threshold=1e-12, krylov_tol=1e-12, order=1, sample_timesteps=True, random_seed=3,
              tdvp_mode="1site")
    p = AnalogSimParams(observables=[Observable(Zg(), s) for s in range(L)], num_traj=4, **kw)
    st = MPS(L, state="x+")
    assert engine_bond_caps(p, st) == (1, 1) and engine_bond_caps(p, st, can_grow=True) == (4, 4)
    mpo = MPO.ising(L, 1.0, 0.5)
    res = sim(batch=4, native=False).run(st, mpo, p, noise)
    st.normalize("B")
    rows = _oracle_rows(L, noise, kw, [np.asarray(w) for w in mpo.tensors], st.tensors, 4)
    for t in range(4):
        for s_ in range(L):
            assert np.allclose(res.trajectories[s_][t], rows[t][0][s_], atol=1e-9), (t, s_)


def test_run_context_validation_happens_before_anything_runs(sim):
    """validate_noise_model_for_run (noise_model.py:668-790) at the top of Simulator.run / run_circuit: non-Pauli long-range noise
    on the analog MPS path and non-adjacent two-site noise on the digital path are refused with ValueError; no engine is built."""
    from yaqs_amd.api import AnalogSimParams, DigitalSimParams, MPO, MPS, NoiseModel, Observable, Z as Zg

    L = 4
    low = np.array([[0, 1], [0, 0]], dtype=complex)
    lr = NoiseModel([{"name": "custom", "sites": [0, 3], "strength": 0.2, "factors": (low, np.diag([1.0, -1.0]).astype(complex))}])
    p = AnalogSimParams(observables=[Observable(Zg(), 0)], elapsed_time=0.1, dt=0.1, max_bond_dim=4)
    with pytest.raises(ValueError, match="non-Pauli long-range"):
        sim(batch=1, native=False).run(MPS(L, state="x+"), MPO.ising(L, 1.0, 0.5), p, lr)
    pauli_lr = NoiseModel([{"name": "longrange_crosstalk_xy", "sites": [0, 2], "strength": 0.1}])
    dp = DigitalSimParams(observables=[Observable(Zg(), 0)], num_traj=2, max_bond_dim=4)
    with pytest.raises(ValueError, match="non-adjacent"):
        sim(batch=1).run_circuit(MPS(L, state="zeros"), [], dp, pauli_lr)
    assert OracleEngine.instances == []


def test_sample_at_and_segment_stitching_of_the_python_schedule(sim):
    """The continuation options of the drivers (analog_tjm.py:206-255, 369-400) in ``TrajectoryBatch.run``: ``sample_at``, and an
    order-2 run cut after 3 of 6 steps (cursors of the trajectory streams handed on, phi left in set 0, sample streams on the global
    timeline) - against the REFERENCE's outputs (tests/golden/continuation.npz), which stitch to the continuous run bit for bit."""
    import os

    from conftest import GOLDEN
    from yaqs_amd.api import AnalogSimParams, MPS, NoiseModel, Observable, Z as Zg
    from yaqs_amd.tjm import TrajectoryBatch

    g = np.load(os.path.join(GOLDEN, "continuation.npz"))
    L = 5
    mpo = [g[f"mpo{i}"] for i in range(L)]
    noise = NoiseModel([{"name": n, "sites": [i], "strength": 0.15} for i in range(L) for n in ("lowering", "pauli_z")])
    kw = dict(dt=0.1, max_bond_dim=4, svd_threshold=1e-9, krylov_tol=1e-12, random_seed=31)
    obs = [Observable(Zg(), s) for s in range(L)]
    st = MPS(L, state="x+")
    st.normalize("B")
    traj = [0, 1, 2, 3]
    for order in (1, 2):
        e = OracleEngine(L, 4, 4, mpo)
        p = AnalogSimParams(observables=obs, elapsed_time=0.6, sample_timesteps=True, order=order, **kw)
        r, _ = TrajectoryBatch(e, p, noise).run(traj, st, sample_at=[0, 2, 5])
        assert np.allclose(r, g[f"sample_at_order{order}"], atol=1e-9), order
        p1 = AnalogSimParams(observables=obs, elapsed_time=0.6, sample_timesteps=False, order=order, **kw)
        r, _ = TrajectoryBatch(e, p1, noise).run(traj, st, sample_at=[3])
        assert np.allclose(r, g[f"sample_at_single_order{order}"], atol=1e-9), order
        with pytest.raises(ValueError, match="outside the time grid"):
            TrajectoryBatch(e, p, noise).run(traj, st, sample_at=[7])
        with pytest.raises(ValueError, match="requires sample_timesteps=True"):
            TrajectoryBatch(e, p1, noise).run(traj, st, sample_at=[1, 2])
    seg = AnalogSimParams(observables=obs, elapsed_time=0.3, sample_timesteps=True, order=2, **kw)
    e = OracleEngine(L, 4, 4, mpo)
    tb = TrajectoryBatch(e, seg, noise)
    r1, _ = tb.run(traj, st, rng_pos=np.zeros(4, dtype=np.int64))
    tb2 = TrajectoryBatch(e, seg, noise)
    r2, _ = tb2.run(traj, None, continue_trajectory=True, sample_timestep_offset=3, rng_pos=tb.rng_pos)
    assert np.allclose(r1, g["segment1"], atol=1e-9) and np.allclose(r2, g["segment2"], atol=1e-9)
    assert np.allclose(r1, g["whole"][:, :, :4], atol=1e-9) and np.allclose(r2, g["whole"][:, :, 3:], atol=1e-9)
    assert np.array_equal(e.bond_dims(0)[:, 1:], g["phi_bonds"])
    assert np.all(tb2.rng_pos > tb.rng_pos)


def test_dynamic_tdvp_driven_from_the_host_matches_the_reference(sim):
    """tdvp_mode="dynamic" (integrators.py:294-511): the per-trajectory branching between the two-site and the one-site update is
    host logic on top of the engine's site-level steps.  One sweep on the chains of tests/golden/f3_dynamic_bug.npz (bonds below, at
    and above the cap) and whole noisy trajectories of both drivers, against the REFERENCE's outputs."""
    import os

    from conftest import GOLDEN
    from yaqs_amd.api import AnalogSimParams, MPS, NoiseModel, Observable, Z as Zg
    from yaqs_amd.tjm import dynamic_tdvp

    g = np.load(os.path.join(GOLDEN, "f3_dynamic_bug.npz"))
    for key in g["cases"]:
        key = str(key)
        L = int(key.split("_")[0][1:])
        cap = key.split("_")[2][3:]
        cap = None if cap == "None" else int(cap)
        mpo = [g[f"{key}_mpo{i}"] for i in range(L)]
        e = OracleEngine(L, 64, 2, mpo)
        e.set_params(dt=0.1, svd_threshold=1e-9, max_bond_dim=cap, krylov_tol=1e-12, tdvp_mode="dynamic")
        e.load_state([g[f"{key}_in{i}"] for i in range(L)])
        dynamic_tdvp(e, 0, cap, 0.1, 1)
        for b in range(2):
            out = o.MPSState(e.export_state(b), 0)
            assert [t.shape[2] for t in out.tensors] == list(g[f"{key}_dynamic_bonds"]), key
            v, ref = out.to_vec(), g[f"{key}_dynamic_vec"]
            assert abs(abs(np.vdot(ref, v)) - np.vdot(ref, ref).real) < 1e-9, key
    L = 6
    mpo = [g[f"traj_mpo{i}"] for i in range(L)]
    noise = NoiseModel([{"name": n, "sites": [i], "strength": 0.1} for i in range(L) for n in ("lowering", "pauli_z")])
    st = MPS(L, tensors=[g[f"traj_in{i}"] for i in range(L)])
    from yaqs_amd.api import MPO

    H = MPO(mpo)
    for order in (1, 2):
        p = AnalogSimParams(observables=[Observable(Zg(), s) for s in range(L)], elapsed_time=0.5, dt=0.1, num_traj=4, max_bond_dim=4, svd_threshold=1e-9,
                            krylov_tol=1e-12, order=order, sample_timesteps=True, random_seed=9, tdvp_mode="dynamic")
        res = sim(batch=4, native=False).run(st, H, p, noise)
        want = g[f"traj_dynamic_order{order}_results"]
        for s_ in range(L):
            assert np.allclose(res.trajectories[s_], want[:, s_, :], atol=1e-8), (order, s_, np.abs(res.trajectories[s_] - want[:, s_, :]).max())


def test_bug_integrator_driven_from_the_host_matches_the_reference(sim):
    """evolution_mode="bug" (core/methods/bug.py:128-257): the sequence of steps of one BUG time step - two half-sweeps with alternating
    endpoints, compression, renormalisation - is host logic over the engine's BUG steps.  Single steps on the generic-state chains of
    tests/golden/f3_dynamic_bug.npz and whole noisy trajectories of both drivers against the REFERENCE's outputs.  (From a product
    state the reference's own result is decided by rounding noise - exactly dependent columns in the stacked basis - so those cases
    are not compared.)"""
    import os

    from conftest import GOLDEN
    from yaqs_amd.api import AnalogSimParams, MPO, MPS, NoiseModel, Observable, Z as Zg
    from yaqs_amd.tjm import bug_step

    g = np.load(os.path.join(GOLDEN, "f3_dynamic_bug.npz"))
    for key in g["cases"]:
        key = str(key)
        if key.endswith("x+"):
            continue
        L = int(key.split("_")[0][1:])
        cap = key.split("_")[2][3:]
        cap = None if cap == "None" else int(cap)
        mpo = [g[f"{key}_mpo{i}"] for i in range(L)]
        e = OracleEngine(L, 64, 2, mpo)
        p = AnalogSimParams(observables=[Observable(Zg(), 0)], elapsed_time=0.1, dt=0.1, max_bond_dim=cap, svd_threshold=1e-9, krylov_tol=1e-12,
                            evolution_mode="bug")
        e.set_params(dt=0.1, svd_threshold=1e-9, max_bond_dim=cap, krylov_tol=1e-12)
        e.load_state([g[f"{key}_in{i}"] for i in range(L)])
        bug_step(e, 0, p, mpo)
        for b in range(2):
            out = o.MPSState(e.export_state(b), 0)
            assert [t.shape[2] for t in out.tensors] == list(g[f"{key}_bug_bonds"]), key
            v, ref = out.to_vec(), g[f"{key}_bug_vec"]
            assert abs(abs(np.vdot(ref, v)) - np.vdot(ref, ref).real) < 1e-9, key
    L = 6
    noise = NoiseModel([{"name": n, "sites": [i], "strength": 0.1} for i in range(L) for n in ("lowering", "pauli_z")])
    st = MPS(L, tensors=[g[f"traj_in{i}"] for i in range(L)])
    H = MPO([g[f"traj_mpo{i}"] for i in range(L)])
    for order in (1, 2):
        p = AnalogSimParams(observables=[Observable(Zg(), s) for s in range(L)], elapsed_time=0.5, dt=0.1, num_traj=4, max_bond_dim=4, svd_threshold=1e-9,
                            krylov_tol=1e-12, order=order, sample_timesteps=True, random_seed=9, evolution_mode="bug")
        res = sim(batch=4, native=False).run(st, H, p, noise)
        want = g[f"traj_bug_order{order}_results"]
        for s_ in range(L):
            assert np.allclose(res.trajectories[s_], want[:, s_, :], atol=1e-8), (order, s_, np.abs(res.trajectories[s_] - want[:, s_, :]).max())
