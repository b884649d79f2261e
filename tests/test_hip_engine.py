"""GPU parity tests of the batched engine against golden fixtures (reference outputs) and the oracle."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import tjm_oracle as o  # noqa: E402  (checker only)

Z = o.PAULI["z"]
X = o.PAULI["x"]


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def tensors(g, prefix):
    out, i = [], 0
    while f"{prefix}{i}" in g:
        out.append(g[f"{prefix}{i}"])
        i += 1
    return out


def phase_align(a, b):
    ov = np.vdot(b, a)
    return b * (ov / abs(ov)) if abs(ov) > 0 else b


def vec_of(tensor_list):
    return o.MPSState(tensor_list, 0).to_vec()


def make_engine(L, chi, B, mpo):
    from yaqs_amd.engine import BatchEngine

    assert torch.cuda.is_available()
    return BatchEngine(L, chi, B, mpo)


def make_engine_d(L, chi, B, mpo, d):
    from yaqs_amd.engine import BatchEngine

    assert torch.cuda.is_available()
    return BatchEngine(L, chi, B, mpo, d=d)


def test_one_tdvp_call_matches_reference_fixture():
    g = load("tdvp_step")
    for key in g["cases"]:
        key = str(key)
        L, chi, mode, sweeps = key.split("_")
        L, chi, sweeps = int(L[1:]), int(chi[3:]), int(sweeps[1:])
        mpo = tensors(g, key + "_mpo")
        e = make_engine(L, chi, 3, mpo)
        e.set_params(dt=0.1, svd_threshold=1e-9, max_bond_dim=chi, krylov_tol=1e-12, tdvp_sweeps=sweeps, tdvp_mode=mode)
        e.load_state(tensors(g, key + "_in"))
        e.tdvp()
        for b in (0, 2):
            out = e.export_state(b)
            assert [t.shape[2] for t in out] == list(g[key + "_bonds"]), key
            assert np.allclose(vec_of(out), g[key + "_vec"], atol=1e-9), key
        e.close()


def _noise_sets(L):
    kx = np.kron(X, X)
    return {
        "pauli": [o.make_process(n, [i], 0.1 + 0.01 * i) for i in range(L) for n in ("pauli_z", "pauli_x")],
        "lowering": [o.make_process("lowering", [i], 0.2) for i in range(L)],
        "mixed": [o.make_process(n, [i], 0.1) for i in range(L) for n in ("lowering", "pauli_z")],
        "twosite": [o.make_process("pauli_z", [i], 0.05) for i in range(L)]
        + [o.make_process("crosstalk_xx", [i, i + 1], 0.07, matrix=kx) for i in range(L - 1)]
        + [o.make_process("longrange_crosstalk_zz", [0, 3], 0.03, factors=(Z, Z))],
    }


def test_dissipation_and_jump_step_match_reference_fixture():
    g = load("noise_step")
    L = 6
    sets = _noise_sets(L)
    mpo = o.ising_mpo(L, 1.0, 0.5)
    for key in g["cases"]:
        key = str(key)
        nname, mode = key.split("_")
        if nname not in sets:
            continue
        procs = sets[nname]
        e = make_engine(L, 8, 2, mpo)
        e.set_params(dt=0.1, svd_threshold=1e-10, max_bond_dim=8, krylov_tol=1e-12)
        e.set_noise(procs, [o.is_pauli(p) for p in procs])
        e.load_state(tensors(g, key + "_in"))
        e.dissipate(0.1)
        ref = g[key + "_after_diss_vec"]
        assert np.allclose(phase_align(ref, vec_of(e.export_state(1))), ref, atol=1e-10), key
        u = np.tile(np.concatenate([g[key + "_u"], [0.5]])[:2], (2, 1))
        e.set_uniforms(u)
        jumped, dp = e.stochastic(0.1)
        assert abs(dp[0] - float(g[key + "_dp"])) < 1e-11, key
        assert bool(jumped[0]) == (mode != "nojump"), key
        out = e.export_state(0)
        ref = g[key + "_final_vec"]
        assert np.allclose(phase_align(ref, vec_of(out)), ref, atol=1e-9), key
        assert [t.shape[2] for t in out] == list(g[key + "_bonds"]), key
        e.close()


def test_unsupported_noise_raises_not_implemented():
    # non-Pauli long-range processes raise NotImplementedError in the reference too (dissipation.py:136-138)
    L = 6
    procs = [o.make_process("lr", [0, 3], 0.07, factors=(2.0 * X, X))]
    e = make_engine(L, 4, 1, o.ising_mpo(L, 1.0, 0.5))
    e.set_params(dt=0.1, svd_threshold=1e-10, max_bond_dim=4)
    e.set_noise(procs, [False] * len(procs))
    e.load_state(o.MPSState.product(L, "x+").tensors)
    with pytest.raises(NotImplementedError):
        e.dissipate(0.1)
    e.close()


def _run(L, init, noise, params, mpo, trajs, batch=None, native=False):
    from yaqs_amd.api import MPS
    from yaqs_amd.tjm import TrajectoryBatch

    e = make_engine(L, params.max_bond_dim, len(trajs), mpo)
    tb = TrajectoryBatch(e, params, noise)
    r, d = tb.run(trajs, MPS(L, tensors=init), native=native)
    e.close()
    return r, d, tb


def test_trajectories_match_reference_fixture_and_pinned_golden():
    from yaqs_amd.api import AnalogSimParams, NoiseModel, Observable, Z as Zg

    g = load("trajectories")
    L = 5
    mpo = tensors(g, "mpo")
    noise = NoiseModel([{"name": n, "sites": [i], "strength": 0.1} for i in range(L) for n in ("lowering", "pauli_z")])
    init = o.MPSState.product(L, "zeros").tensors
    for order in (1, 2):
        for sample in (False, True):
            p = AnalogSimParams(observables=[Observable(Zg(), s) for s in range(L)], elapsed_time=1, dt=0.1, num_traj=10, max_bond_dim=4,
                                svd_threshold=1e-6, order=order, sample_timesteps=sample, random_seed=42)
            key = f"order{order}_sample{int(sample)}"
            r, d, tb = _run(L, init, noise, p, mpo, list(range(10)))
            dps = np.array(tb.dp_log)  # [calls, B]
            for i in range(10):
                ref_dp = g[key + "_dp"][i]
                ref_dp = ref_dp[~np.isnan(ref_dp)]
                assert len(ref_dp) == dps.shape[0]
                # krylov_tol = 1e-4 here (the "balanced" preset): adaptive-stop decisions are shared, values agree far tighter
                assert np.allclose(dps[:, i], ref_dp, atol=1e-8), (key, i)
            assert np.allclose(r, g[key + "_results"], atol=1e-8), key
            assert np.array_equal(d, g[key + "_diag"]), key
            # the one-call C driver (tjm_engine_run: schedule, RNG streams and measurement inside the library)
            rn, dn, _ = _run(L, init, noise, p, mpo, list(range(10)), native=True)
            assert np.allclose(rn, g[key + "_results"], atol=1e-8), key
            assert np.array_equal(dn, g[key + "_diag"]), key
            if order == 2 and not sample:
                assert np.allclose(r.mean(axis=0).ravel(), g["pinned_expected_z"], atol=1e-8)  # tests/test_simulator.py:191-197


def test_closed_and_dephasing_configs_match_reference_fixture():
    from yaqs_amd.api import AnalogSimParams, NoiseModel, Observable, X as Xg, Z as Zg

    g = load("trajectories")
    mpo = tensors(g, "c1_mpo")
    init = o.MPSState.product(10, "zeros").tensors
    for order in (1, 2):
        p = AnalogSimParams(observables=[Observable(Zg(), s) for s in range(10)], elapsed_time=1.0, dt=0.1, max_bond_dim=16, svd_threshold=1e-9,
                            krylov_tol=1e-12, order=order, sample_timesteps=True, random_seed=42)
        # BASELINE.json configs[0]: all 8 trajectories (the reference collapses a closed run to one, simulator.py:1549-1554; called
        # per index they are eight identical jobs)
        trajs = list(range(8))
        r, d, _ = _run(10, init, None, p, mpo, trajs)
        rn, dn, _ = _run(10, init, None, p, mpo, trajs, native=True)
        assert np.allclose(rn, r, atol=1e-12) and np.array_equal(dn, d)
        for t in trajs:
            assert np.allclose(r[t], g[f"c1_order{order}_results"], atol=1e-9), t
            assert np.array_equal(r[t], r[0]), t  # a closed system is deterministic, and a slot's result does not depend on its position
            assert np.array_equal(d[t], g[f"c1_order{order}_diag"]), t
    mpo = tensors(g, "c2_mpo")
    init = o.MPSState.product(8, "x+").tensors
    noise = NoiseModel([{"name": "pauli_z", "sites": [i], "strength": 0.1} for i in range(8)])
    p = AnalogSimParams(observables=[Observable(Zg(), s) for s in range(8)] + [Observable(Xg(), s) for s in range(8)], elapsed_time=1.0, dt=0.1,
                        max_bond_dim=8, svd_threshold=1e-12, krylov_tol=1e-12, order=1, sample_timesteps=True, random_seed=42)
    r, d, tb = _run(8, init, noise, p, mpo, list(range(8)))
    assert np.allclose(r, g["c2_results"], atol=1e-8)
    assert np.array_equal(d, g["c2_diag"])


def test_simulator_front_end_runs_in_chunks():
    from yaqs_amd.api import AnalogSimParams, MPO, MPS, NoiseModel, Observable, Z as Zg
    from yaqs_amd.tjm import Simulator

    L = 6
    p = AnalogSimParams(observables=[Observable(Zg(), s) for s in (3, 0)], elapsed_time=0.3, dt=0.1, num_traj=7, max_bond_dim=8, svd_threshold=1e-10,
                        krylov_tol=1e-10, random_seed=5)
    noise = NoiseModel([{"name": "pauli_x", "sites": [i], "strength": 0.3} for i in range(L)])
    res = Simulator(batch=4).run(MPS(L, state="zeros"), MPO.ising(L, 1.0, 0.5), p, noise)
    assert len(res.trajectories) == 2 and res.trajectories[0].shape == (7, 4)
    # same run through the oracle
    op = o.Params(observables=[o.Obs(Z, 3), o.Obs(Z, 0)], elapsed_time=0.3, dt=0.1, max_bond_dim=8, svd_threshold=1e-10, krylov_tol=1e-10,
                  random_seed=5)
    on = [o.make_process("pauli_x", [i], 0.3) for i in range(L)]
    idx = op.observable_sorted_indices
    for t in range(7):
        r, _, _ = o.run_trajectory(t, o.MPSState.product(L, "zeros"), on, op, o.ising_mpo(L, 1.0, 0.5))
        assert np.allclose(res.trajectories[0][t], r[idx[0]], atol=1e-8)
        assert np.allclose(res.trajectories[1][t], r[idx[1]], atol=1e-8)


def test_one_site_tdvp_trajectories_match_oracle():
    """Config-4-like path (tdvp_mode="1site", frozen bonds) on a D=5 Heisenberg MPO with dephasing, against the oracle."""
    from yaqs_amd.api import AnalogSimParams, MPO, MPS, NoiseModel, Observable, X as Xg, Z as Zg

    L, chi = 6, 8
    rng = np.random.default_rng(9)
    st = o.MPSState.haar(L, chi, rng)
    st.normalize("B")
    init = [t.copy() for t in st.tensors]
    mpo = MPO.heisenberg(L, 1.0, 1.0, 0.5, 0.2)
    noise = NoiseModel([{"name": "pauli_z", "sites": [i], "strength": 0.05} for i in range(L)])
    p = AnalogSimParams(observables=[Observable(Zg(), s) for s in range(L)] + [Observable(Xg(), 2)], elapsed_time=0.3, dt=0.05, max_bond_dim=chi,
                        svd_threshold=1e-12, krylov_tol=1e-12, order=1, sample_timesteps=True, random_seed=11, tdvp_mode="1site")
    r, d, tb = _run(L, init, noise, p, mpo.tensors, list(range(4)))
    op = o.Params(observables=[o.Obs(Z, s) for s in range(L)] + [o.Obs(X, 2)], elapsed_time=0.3, dt=0.05, max_bond_dim=chi, svd_threshold=1e-12,
                  krylov_tol=1e-12, order=1, sample_timesteps=True, random_seed=11, tdvp_mode="1site")
    on = [o.make_process("pauli_z", [i], 0.05) for i in range(L)]
    omp = o.heisenberg_mpo(L, 1.0, 1.0, 0.5, 0.2)
    for t in range(4):
        ro, do, _ = o.run_trajectory(t, o.MPSState([x.copy() for x in init], 0), on, op, omp)
        assert np.allclose(r[t], ro, atol=1e-8), t
        assert np.array_equal(d[t], do), t


def test_adjacent_two_site_noise_and_two_site_observables_match_oracle():
    """Non-Pauli adjacent two-site processes (merged expm / jump + truncated split) and nearest-neighbour observables."""
    from yaqs_amd.api import AnalogSimParams, MPO, NoiseModel, Observable, Z as Zg

    L, chi = 6, 8
    low = o.JUMP_OPS["lowering"]
    two = np.kron(low, Z) + 0.3 * np.kron(X, low)          # non-Pauli, non-product two-site operator
    zz = np.kron(Z, Z)
    xz = np.kron(X, Z)
    procs = [{"name": "pauli_x", "sites": [i], "strength": 0.05} for i in range(L)] + \
            [{"name": "custom2", "sites": [i, i + 1], "strength": 0.4, "matrix": two} for i in range(0, L - 1, 2)] + \
            [{"name": "crosstalk_zy", "sites": [1, 2], "strength": 0.2}]
    noise = NoiseModel(procs)
    obs = [Observable(Zg(), 0), Observable(zz, [2, 3]), Observable(xz, [0, 1]), Observable(Zg(), 5), Observable(zz, [4, 5])]
    p = AnalogSimParams(observables=obs, elapsed_time=0.4, dt=0.1, max_bond_dim=chi, svd_threshold=1e-10, krylov_tol=1e-12, order=2,
                        sample_timesteps=True, random_seed=3)
    init = o.MPSState.product(L, "x+").tensors
    mpo = MPO.ising(L, 1.0, 0.5)
    r, d, tb = _run(L, init, noise, p, mpo.tensors, list(range(6)))
    assert np.array(tb.jump_log).sum() > 0  # the jump branch (incl. adjacent pairs) is exercised
    rn, dn, _ = _run(L, init, noise, p, mpo.tensors, list(range(6)), native=True)  # two-site observables through the C driver
    assert np.allclose(rn, r, atol=1e-11) and np.array_equal(dn, d)
    oobs = [o.Obs(Z, 0), o.Obs(zz, [2, 3]), o.Obs(xz, [0, 1]), o.Obs(Z, 5), o.Obs(zz, [4, 5])]
    op = o.Params(observables=oobs, elapsed_time=0.4, dt=0.1, max_bond_dim=chi, svd_threshold=1e-10, krylov_tol=1e-12, order=2,
                  sample_timesteps=True, random_seed=3)
    on = [o.make_process(q["name"], q["sites"], q["strength"], matrix=q.get("matrix"), factors=q.get("factors")) for q in noise.processes]
    for t in range(6):
        ro, do, _ = o.run_trajectory(t, o.MPSState.product(L, "x+"), on, op, o.ising_mpo(L, 1.0, 0.5))
        assert np.allclose(r[t], ro, atol=1e-8), t
        assert np.array_equal(d[t], do), t


def test_digital_tebd_trajectories_match_reference_fixture():
    """Circuit path (TEBD gates + per-gate local noise, digital_tjm.py:636-749) against outputs of the reference itself."""
    from yaqs_amd.api import DigitalSimParams, MPS, NoiseModel, Observable, X as Xg, Z as Zg, ising_trotter_layers
    from yaqs_amd.tjm import DigitalBatch

    g = load("digital")
    L, steps = 8, 5
    obs = [Observable(Zg(), s) for s in range(L)] + [Observable(Xg(), 3)]
    mpo = o.ising_mpo(L, 1.0, 0.5)  # the engine wants an MPO shape; the circuit path never applies it
    init = MPS(L, state="zeros")

    def run(noise, params, layers, trajs, chi):
        e = make_engine(L, chi, len(trajs), mpo)
        db = DigitalBatch(e, params, noise)
        r, d = db.run(trajs, init, layers)
        e.close()
        return r, d, db

    noise = NoiseModel([{"name": n, "sites": [i], "strength": 0.01} for i in range(L) for n in ("pauli_x", "pauli_y", "pauli_z")])
    p = DigitalSimParams(observables=obs, max_bond_dim=16, svd_threshold=1e-9, random_seed=3)
    r, d, _ = run(noise, p, ising_trotter_layers(L, 1.0, 0.5, 0.1, steps), list(range(6)), 16)
    assert np.allclose(r[:, :, 0], g["noisy_results"][:, :, 0], atol=1e-8)
    assert np.array_equal(d, g["noisy_diag"])
    p = DigitalSimParams(observables=obs, max_bond_dim=16, svd_threshold=1e-9, random_seed=3, sample_layers=True, num_mid_measurements=steps)
    r, d, _ = run(None, p, ising_trotter_layers(L, 1.0, 0.5, 0.1, steps, sample_each=True), [0, 1], 16)
    assert np.allclose(r[0], g["noiseless_results"][0], atol=1e-8)
    assert np.array_equal(d[0], g["noiseless_diag"][0])
    noise2 = NoiseModel([{"name": n, "sites": [i], "strength": 0.1} for i in range(L) for n in ("lowering", "pauli_z")])
    p = DigitalSimParams(observables=obs, max_bond_dim=4, svd_threshold=1e-6, random_seed=7)
    r, d, db = run(noise2, p, ising_trotter_layers(L, 1.0, 0.5, 0.1, 3), list(range(6)), 4)
    assert np.array(db.jump_log).sum() > 0
    assert np.allclose(r[:, :, 0], g["strong_results"][:, :, 0], atol=1e-8)
    assert np.array_equal(d, g["strong_diag"])
    # long-range gates in both site orders, routed with adjacent SWAPs (digital_tjm.py:476-499); local noise on the gate's own
    # sites only, including a long-range two-site Pauli channel and a non-Pauli one-site channel
    from yaqs_amd.api import GateLayer, rx_matrix

    cx, rzz = g["lr_cx_matrix"], g["lr_rzz_matrix"]
    lr_layers = [GateLayer([(q, rx_matrix(0.3 + 0.1 * q)) for q in range(L)], [(1, 5, cx), (6, 2, rzz)], [(4, 3, cx), (7, 0, cx)], 0)
                 for _ in range(2)]
    noise3 = NoiseModel([{"name": "pauli_x", "sites": [i], "strength": 0.05} for i in range(L)] +
                        [{"name": "crosstalk_zz", "sites": [1, 5], "strength": 0.1}, {"name": "lowering", "sites": [6], "strength": 0.2}])
    p = DigitalSimParams(observables=obs, max_bond_dim=4, svd_threshold=1e-8, random_seed=11, gate_mode="swaps")
    r, d, _ = run(None, p, lr_layers, [0], 4)
    assert np.allclose(r[0], g["lr_noiseless_results"][0], atol=1e-8)
    assert np.array_equal(d[0], g["lr_noiseless_diag"][0])
    r, d, db = run(noise3, p, lr_layers, list(range(6)), 4)
    assert np.allclose(r, g["lr_noisy_results"], atol=1e-8)
    assert np.array_equal(d, g["lr_noisy_diag"])
    with pytest.raises(NotImplementedError):  # the TDVP-window route of distant pairs (digital_tjm.py:408-453) is not built
        run(None, DigitalSimParams(observables=obs, max_bond_dim=4, svd_threshold=1e-8, random_seed=11, gate_mode="tdvp"), lr_layers, [0], 4)


def test_chi256_heisenberg_lowering_step_matches_oracle():
    """Config-3-like step in fp64 at the largest supported bond: Heisenberg D=5 MPO, `lowering` (non-Pauli) noise on every
    site, chi = 256, so the two-site split is 512 x 512 (split X / W Jacobi with 8 row groups, doubly QR-preconditioned)
    and the dissipation / jump shifts are 512 x 256."""
    from yaqs_amd.api import AnalogSimParams, NoiseModel, Observable, Z as Zg

    L, chi = 18, 256
    rng = np.random.default_rng(7)
    st = o.MPSState.haar(L, chi, rng)
    st.normalize("B")
    init = [t.copy() for t in st.tensors]
    mpo = o.heisenberg_mpo(L, 1.0, 1.0, 0.5, 0.0)
    noise = NoiseModel([{"name": "lowering", "sites": [i], "strength": 0.05} for i in range(L)])
    p = AnalogSimParams(observables=[Observable(Zg(), s) for s in range(L)], elapsed_time=0.05, dt=0.05, max_bond_dim=chi, svd_threshold=1e-12,
                        krylov_tol=1e-10, order=1, sample_timesteps=True, random_seed=42)
    r, d, tb = _run(L, init, noise, p, mpo, [0, 1])
    on = [o.make_process("lowering", [i], 0.05) for i in range(L)]
    op = o.Params(observables=[o.Obs(Z, s) for s in range(L)], elapsed_time=0.05, dt=0.05, max_bond_dim=chi, svd_threshold=1e-12, krylov_tol=1e-10,
                  random_seed=42, sample_timesteps=True)
    for t in range(2):
        rr, dd, _ = o.run_trajectory(t, o.MPSState([x.copy() for x in init], 0), on, op, mpo)
        assert np.allclose(r[t], rr, atol=1e-8), np.abs(r[t] - rr).max()
        assert np.array_equal(d[t], dd)


@pytest.mark.parametrize("basis", ["Z", "X", "Y"])
def test_shot_sampling_matches_oracle_and_born_probabilities(basis):
    """measure_shots (mps.py:1282-1417) batched on the GPU: with the same uniforms every shot equals the oracle's restatement of
    measure_single_shot, and the histogram follows the Born probabilities of the dense state."""
    L, chi, B, shots = 6, 8, 3, 4000
    rng = np.random.default_rng(5)
    st = o.MPSState.haar(L, chi, rng)
    st.normalize("B")
    e = make_engine(L, chi, B, o.ising_mpo(L, 1.0, 0.5))
    e.load_state(st.tensors)
    u = rng.random((B, shots, L))
    bits = e.sample_shots(u, basis)
    e.close()
    codes = (bits.astype(np.int64) << np.arange(L)).sum(axis=2)  # sum(bit_i << i), mps.py:1350
    for b in range(B):
        for s_ in range(0, 60):
            assert codes[b, s_] == o.measure_single_shot(st, u[b, s_], basis), (b, s_)
    # Born rule on the dense vector (site 0 = least significant index, mps.py:1633-1658), rotated into the measurement basis
    psi = st.to_vec()
    rot = o.BASIS_ROTATION[basis]
    full = np.array([[1.0]])
    for _ in range(L):
        full = np.kron(rot, full)  # site 0 is the fastest index
    prob = np.abs(full @ psi) ** 2
    hist = np.bincount(codes.ravel(), minlength=2 ** L) / codes.size
    assert np.abs(hist - prob).max() < 5 * np.sqrt(prob.max() / codes.size) + 1e-3
    with pytest.raises(ValueError):
        make_engine(L, chi, B, o.ising_mpo(L, 1.0, 0.5)).sample_shots(u, "Q")
    if basis == "Z":  # the reference's own outcomes for its recorded draws (tests/golden/shots.npz)
        g = load("shots")
        e = make_engine(6, 8, 1, o.ising_mpo(6, 1.0, 0.5))
        e.load_state([g[f"t{i}"] for i in range(6)])
        for bi, bs in enumerate("ZXY"):
            bits = e.sample_shots(g["uniforms"][bi][None], bs)
            assert np.array_equal((bits[0].astype(np.int64) << np.arange(6)).sum(axis=1), g["codes"][bi]), bs
        e.close()


def test_run_circuit_with_shots_matches_dense_probabilities():
    """Simulator.run_circuit: noiseless circuit -> one trajectory, the whole shot budget sampled from its final state
    (simulator.py:1001-1050); noisy shots-only run -> one stochastic state per shot; combined run splits the budget."""
    from yaqs_amd.api import DigitalSimParams, MPS, NoiseModel, Observable, Z as Zg, ising_trotter_layers
    from yaqs_amd.tjm import Simulator

    L = 6
    layers = ising_trotter_layers(L, 1.0, 0.5, 0.2, 3)
    p = DigitalSimParams(max_bond_dim=8, svd_threshold=1e-12, random_seed=1, shots=20000)
    res = Simulator(batch=4).run_circuit(MPS(L, state="zeros"), layers, p)
    assert sum(res.counts.values()) == 20000
    op = o.DigitalParams(observables=[o.Obs(Z, 0)], max_bond_dim=8, svd_threshold=1e-12, random_seed=1, get_state=True)
    _, _, final = o.digital_tjm(0, o.MPSState.product(L, "zeros"), None, op, o.ising_trotter_layers(L, 1.0, 0.5, 0.2, 3))
    prob = np.abs(final.to_vec()) ** 2
    hist = np.zeros(2 ** L)
    for k, v in res.counts.items():
        hist[k] = v / 20000
    assert np.abs(hist - prob).max() < 5 * np.sqrt(prob.max() / 20000) + 1e-3
    noise = NoiseModel([{"name": "pauli_x", "sites": [i], "strength": 0.05} for i in range(L)])
    p = DigitalSimParams(max_bond_dim=8, svd_threshold=1e-12, random_seed=1, shots=37)
    res = Simulator(batch=16).run_circuit(MPS(L, state="zeros"), layers, p, noise)
    assert sum(res.counts.values()) == 37 and res.trajectory_diagnostics.shape[0] == 37
    p = DigitalSimParams(observables=[Observable(Zg(), 2)], num_traj=8, max_bond_dim=8, svd_threshold=1e-12, random_seed=1, shots=20)
    res = Simulator(batch=8).run_circuit(MPS(L, state="zeros"), layers, p, noise)
    assert sum(res.counts.values()) == 20 and res.trajectories[0].shape == (8, 1)


def test_entropy_schmidt_spectrum_and_pvm_observables():
    """Meta-observables of evaluate_observables (mps.py:1200-1218) through the engine, against the reference fixture and, along
    a noisy trajectory, against the oracle."""
    from yaqs_amd.api import AnalogSimParams, Entropy, MPS, NoiseModel, Observable, PVM, SchmidtSpectrum, Z as Zg
    from yaqs_amd.tjm import TrajectoryBatch

    g = load("shots")
    L = 6
    tens = [g[f"t{i}"] for i in range(L)]
    e = make_engine(L, 8, 2, o.ising_mpo(L, 1.0, 0.5))
    e.load_state(tens)
    for i in range(L - 1):
        spec = e.bond_spectrum(i)
        ref = g["schmidt"][i]
        ref = ref[~np.isnan(ref)]
        assert np.allclose(spec[0, : len(ref)], ref, atol=1e-12) and np.allclose(spec[1], spec[0], atol=1e-14)
    for b, ref in zip(g["pvm_strings"], g["pvm"]):
        assert np.allclose(e.bitstring_probability(str(b)), ref, atol=1e-13)
    e.close()
    obs = [Observable(Entropy(), [2, 3]), Observable(Zg(), 1), Observable(SchmidtSpectrum(), [1, 2]), Observable(PVM("010101"), 0)]
    p = AnalogSimParams(observables=obs, elapsed_time=0.3, dt=0.1, max_bond_dim=8, svd_threshold=1e-10, krylov_tol=1e-10, order=1,
                        sample_timesteps=True, random_seed=9)
    noise = NoiseModel([{"name": "lowering", "sites": [i], "strength": 0.2} for i in range(L)])
    e = make_engine(L, 8, 3, o.ising_mpo(L, 1.0, 0.5))
    tb = TrajectoryBatch(e, p, noise)
    r, d = tb.run([0, 1, 2], MPS(L, tensors=tens), native=True)  # falls back to the host schedule for meta-observables
    e.close()
    on = [o.make_process("lowering", [i], 0.2) for i in range(L)]
    rows = {u: row for u, row in enumerate(p.observable_sorted_indices)}
    for t in range(3):
        # replay the trajectory in the oracle and evaluate the same quantities on its states
        st = o.MPSState([x.copy() for x in tens], 0)
        op = o.Params(observables=[o.Obs(Z, 1)], elapsed_time=0.3, dt=0.1, max_bond_dim=8, svd_threshold=1e-10, krylov_tol=1e-10,
                      random_seed=9, sample_timesteps=True)
        rng = o.trajectory_rng(9, t)
        for j in range(4):
            if j > 0:
                o.tdvp(st, o.ising_mpo(L, 1.0, 0.5), op)
                o.apply_dissipation(st, on, 0.1, op)
                st = o.stochastic_process(st, on, 0.1, op, rng)
            assert abs(r[t, rows[0], j] - o.get_entropy(st, [2, 3])) < 1e-8
            assert abs(r[t, rows[3], j] - o.project_onto_bitstring(st, "010101")) < 1e-9
            ref = o.get_schmidt_spectrum(st, [1, 2])
            assert np.allclose(tb.schmidt[(rows[2], j)][t], ref, atol=1e-9, equal_nan=True)


def test_scheduled_jumps_match_reference_fixture():
    """NoiseModel.scheduled_jumps through the engine (one-site at t = 0 and mid-run, adjacent two-site) against the reference."""
    from yaqs_amd.api import AnalogSimParams, MPS, NoiseModel, Observable, X as Xg, Z as Zg
    from yaqs_amd.tjm import Simulator

    g = load("scheduled")
    L = 6
    sched = [{"time": 0.0, "sites": [2], "name": "pauli_x"}, {"time": 0.2, "sites": [4], "name": "lowering"},
             {"time": 0.3, "sites": [1, 2], "name": "custom", "matrix": g["two"]}]
    noise = NoiseModel([{"name": "pauli_z", "sites": [i], "strength": 0.2} for i in range(L)], scheduled_jumps=sched)
    obs = [Observable(Zg(), s) for s in range(L)] + [Observable(Xg(), 0)]
    p = AnalogSimParams(observables=obs, elapsed_time=0.5, dt=0.1, max_bond_dim=8, svd_threshold=1e-10, krylov_tol=1e-10, order=1,
                        sample_timesteps=True, random_seed=21)
    r, d, _ = _run(L, o.MPSState.product(L, "x+").tensors, noise, p, [g[f"mpo{i}"] for i in range(L)], [0, 1, 2, 3])
    assert np.allclose(r, g["results"], atol=1e-8)
    assert np.array_equal(d, g["diag"])
    p2 = AnalogSimParams(observables=obs, elapsed_time=0.5, dt=0.1, max_bond_dim=8, order=2, random_seed=21)
    with pytest.raises(ValueError):
        _run(L, o.MPSState.product(L, "x+").tensors, noise, p2, [g[f"mpo{i}"] for i in range(L)], [0])
    bad = NoiseModel([], scheduled_jumps=[{"time": 0.1, "sites": [0], "name": "lowering"}, {"time": 0.1, "sites": [0], "name": "lowering"}])
    with pytest.raises(ValueError):  # sigma^- twice annihilates the state
        _run(L, o.MPSState.product(L, "zeros").tensors, bad, p, [g[f"mpo{i}"] for i in range(L)], [0])


def test_get_state_returns_the_final_mps_of_a_closed_run():
    """get_state (simulator.py:1438, 1555-1557): the final physical state of a noise-free run; noisy runs refuse it."""
    from yaqs_amd.api import AnalogSimParams, MPO, MPS, NoiseModel, Observable, Z as Zg
    from yaqs_amd.tjm import Simulator

    L = 6
    for order in (1, 2):
        p = AnalogSimParams(observables=[Observable(Zg(), 0)], elapsed_time=0.3, dt=0.1, max_bond_dim=8, svd_threshold=1e-12, krylov_tol=1e-12,
                            order=order, get_state=True, sample_timesteps=False)
        res = Simulator().run(MPS(L, state="x+"), MPO.ising(L, 1.0, 0.5), p)
        op = o.Params(observables=[o.Obs(Z, 0)], elapsed_time=0.3, dt=0.1, max_bond_dim=8, svd_threshold=1e-12, krylov_tol=1e-12, order=order,
                      get_state=True, sample_timesteps=False)
        _, _, ref = o.run_trajectory(0, o.MPSState.product(L, "x+"), None, op, o.ising_mpo(L, 1.0, 0.5))
        got = vec_of(res.output_state.tensors)
        want = ref.to_vec()
        assert np.allclose(phase_align(want, got), want, atol=1e-9), order
    with pytest.raises(ValueError):
        Simulator().run(MPS(L, state="x+"), MPO.ising(L, 1.0, 0.5), p, NoiseModel([{"name": "pauli_z", "sites": [0], "strength": 0.1}]))


def test_piecewise_hamiltonian_matches_reference_fixture():
    """A tuple of MPOs, one per interval, through TrajectoryBatch.set_intervals (engine re-loads the MPO between steps)."""
    from yaqs_amd.api import AnalogSimParams, MPS, NoiseModel, Observable, X as Xg, Z as Zg
    from yaqs_amd.tjm import TrajectoryBatch

    g = load("piecewise")
    L, n = 6, 4
    hams = [[g[f"h{k}_mpo{i}"] for i in range(L)] for k in range(n)]
    noise = NoiseModel([{"name": "pauli_z", "sites": [i], "strength": 0.1} for i in range(L)])
    obs = [Observable(Zg(), s) for s in range(L)] + [Observable(Xg(), 2)]
    for order in (1, 2):
        p = AnalogSimParams(observables=obs, elapsed_time=0.1 * n, dt=0.1, max_bond_dim=8, svd_threshold=1e-10, krylov_tol=1e-10, order=order,
                            sample_timesteps=True, random_seed=5)
        e = make_engine(L, 8, 3, hams[0])
        tb = TrajectoryBatch(e, p, noise)
        tb.set_intervals(hams)
        r, d = tb.run([0, 1, 2], MPS(L, state="x+"), native=True)  # the host schedule takes over for piecewise drives
        e.close()
        assert np.allclose(r, g[f"order{order}_results"], atol=1e-8), order
        assert np.array_equal(d, g[f"order{order}_diag"]), order


def test_config4_like_one_site_tdvp_long_range_mpo_padded_state():
    """Config 4 of SURVEY section 8d in small: exponential-sum long-range Ising MPO (D = 4) built on the host, "x+" padded with zeros
    to a fixed chi (tangent space of the padded isometries), tdvp_mode = "1site", dephasing; the same MPO tensors go to the oracle.
    The MPO is checked against the dense Hamiltonian it stands for."""
    from yaqs_amd.api import AnalogSimParams, MPO, MPS, NoiseModel, Observable, Z as Zg

    L, chi = 10, 32
    mpo = MPO.long_range_ising(L, [1.0, 0.25], [0.35, 0.7], 0.5)
    st = MPS(L, state="x+", pad=chi)
    assert max(t.shape[2] for t in st.tensors) == chi and st.tensors[4].shape == (2, 16, 32)
    # the MPO is the Hamiltonian it claims to be
    def embed(i, op):  # site 0 is the fastest index (mps.py:1633-1658)
        return np.kron(np.eye(2 ** (L - 1 - i)), np.kron(op, np.eye(2 ** i)))

    H = o.mpo_to_matrix(mpo.tensors)
    ref = np.zeros_like(H)
    zs = [embed(i, Z) for i in range(L)]
    for i in range(L):
        ref -= 0.5 * embed(i, X)
        for j in range(i + 1, L):
            ref -= (1.0 * 0.35 ** (j - i - 1) + 0.25 * 0.7 ** (j - i - 1)) * (zs[i] @ zs[j])
    assert np.allclose(H, ref, atol=1e-12)
    obs = [Observable(Zg(), s) for s in range(L)]
    noise = NoiseModel([{"name": "pauli_z", "sites": [i], "strength": 0.05} for i in range(L)])
    p = AnalogSimParams(observables=obs, elapsed_time=0.15, dt=0.05, max_bond_dim=chi, svd_threshold=1e-12, krylov_tol=1e-10, order=1,
                        sample_timesteps=True, random_seed=42, tdvp_mode="1site")
    init = [t.copy() for t in st.tensors]
    r, d, _ = _run(L, init, noise, p, mpo.tensors, [0, 1])
    op = o.Params(observables=[o.Obs(Z, s) for s in range(L)], elapsed_time=0.15, dt=0.05, max_bond_dim=chi, svd_threshold=1e-12, krylov_tol=1e-10,
                  order=1, sample_timesteps=True, random_seed=42, tdvp_mode="1site")
    on = [o.make_process("pauli_z", [i], 0.05) for i in range(L)]
    for t in range(2):
        ro, do, _ = o.run_trajectory(t, o.MPSState([x.copy() for x in init], 0), on, op, mpo.tensors)
        assert np.allclose(r[t], ro, atol=1e-8), np.abs(r[t] - ro).max()
        assert np.array_equal(d[t], do)


def test_entry_points_work_in_a_fresh_interpreter():
    """build() then smoke() in one fresh process, the way the driver calls them: the library is loaded before anything else has
    touched the GPU and must still share PyTorch's HIP runtime (a second runtime instance sees no device)."""
    import subprocess
    import sys

    from conftest import ROOT

    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke()"], cwd=ROOT, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "smoke ok" in out.stdout


@pytest.mark.parametrize("L", [1, 2, 3])
@pytest.mark.parametrize("order", [1, 2])
def test_tiny_chains_and_zero_duration(L, order):
    """Edge cases of the drivers: one-, two- and three-site chains (tdvp.py:96-100 falls back to 1TDVP on one site), a
    zero-duration run (analog_tjm.py:313-321), zero noise strength, a trajectory count that is not a multiple of the batch."""
    from yaqs_amd.api import AnalogSimParams, MPO, MPS, NoiseModel, Observable, X as Xg, Z as Zg
    from yaqs_amd.tjm import Simulator

    obs = [Observable(Zg(), L - 1), Observable(Xg(), 0)]
    oobs = [o.Obs(Z, L - 1), o.Obs(X, 0)]
    mpo_o = o.ising_mpo(L, 1.0, 0.7)
    for elapsed, gamma, ntraj in ((0.3, 0.3, 5), (0.0, 0.3, 3), (0.2, 0.0, 4)):
        p = AnalogSimParams(observables=obs, elapsed_time=elapsed, dt=0.1, num_traj=ntraj, max_bond_dim=4, svd_threshold=1e-10, krylov_tol=1e-10,
                            order=order, random_seed=13)
        noise = NoiseModel([{"name": "lowering", "sites": [i], "strength": gamma} for i in range(L)])
        res = Simulator(batch=2).run(MPS(L, state="x+"), MPO.ising(L, 1.0, 0.7), p, noise)
        op = o.Params(observables=oobs, elapsed_time=elapsed, dt=0.1, max_bond_dim=4, svd_threshold=1e-10, krylov_tol=1e-10, order=order,
                      random_seed=13)
        on = [o.make_process("lowering", [i], gamma) for i in range(L)]
        idx = op.observable_sorted_indices
        n_eff = ntraj if gamma > 0 else 1
        assert res.trajectories[0].shape[0] == n_eff
        for t in range(n_eff):
            r, _, _ = o.run_trajectory(t, o.MPSState.product(L, "x+"), on, op, mpo_o)
            for u in range(2):
                assert np.allclose(res.trajectories[u][t], r[idx[u]], atol=1e-8), (L, order, elapsed, gamma, t, u)


def test_unbounded_bond_dimension_is_exact():
    """max_bond_dim=None (the "exact" preset): no truncation by a cap anywhere; the result reproduces dense evolution."""
    import scipy.linalg

    from yaqs_amd.api import AnalogSimParams, MPO, MPS, Observable, Z as Zg
    from yaqs_amd.tjm import Simulator

    L = 8
    p = AnalogSimParams(observables=[Observable(Zg(), s) for s in range(L)], elapsed_time=0.5, dt=0.05, max_bond_dim=None, svd_threshold=1e-14,
                        krylov_tol=1e-12, order=2, sample_timesteps=False)
    res = Simulator().run(MPS(L, state="x+"), MPO.heisenberg(L, 1.0, 0.8, 0.5, 0.3), p)
    H = o.mpo_to_matrix(o.heisenberg_mpo(L, 1.0, 0.8, 0.5, 0.3))
    psi0 = o.MPSState.product(L, "x+").to_vec()
    psi = scipy.linalg.expm(-1j * 0.5 * H) @ psi0
    for s in range(L):
        zs = np.kron(np.eye(2 ** (L - 1 - s)), np.kron(Z, np.eye(2 ** s)))
        assert abs(res.expectation_values[s][0] - np.vdot(psi, zs @ psi).real) < 2e-5  # second-order splitting error of dt = 0.05


def _recording_engine(monkeypatch):
    """Records the storage capacity of every engine the Simulator builds."""
    import yaqs_amd.tjm as tjm_mod

    built = []

    class Recording(tjm_mod.BatchEngine):
        def __init__(self, length, chi_max, batch, mpo, **kw):
            built.append(int(chi_max))
            super().__init__(length, chi_max, batch, mpo, **kw)

    monkeypatch.setattr(tjm_mod, "BatchEngine", Recording)
    return built


@pytest.mark.parametrize("max_bond", [None, 4096, 24])
def test_storage_capacity_grows_on_demand(monkeypatch, max_bond):
    """The reference's presets ask for max_bond_dim = 4096 or None while the bonds stay small: the engine starts with a small
    static capacity and the chunk is repeated with twice the capacity whenever a truncation was clipped by it.  The final pass is
    the reference's run: per-trajectory observables and bond diagnostics equal the oracle's with the same max_bond_dim."""
    from yaqs_amd.api import AnalogSimParams, MPO, MPS, NoiseModel, Observable, X as Xg, Z as Zg
    from yaqs_amd.tjm import Simulator

    built = _recording_engine(monkeypatch)
    L, ntraj = 12, 3
    obs = [Observable(Zg(), s) for s in range(L)] + [Observable(Xg(), 3)]
    oobs = [o.Obs(Z, s) for s in range(L)] + [o.Obs(X, 3)]
    kw = dict(elapsed_time=1.2, dt=0.1, max_bond_dim=max_bond, svd_threshold=1e-10, krylov_tol=1e-11, order=1, random_seed=5)
    p = AnalogSimParams(observables=obs, num_traj=ntraj, sample_timesteps=True, **kw)
    noise = NoiseModel([{"name": "pauli_x", "sites": [i], "strength": 0.05} for i in range(L)])
    res = Simulator(batch=ntraj).run(MPS(L, state="Neel"), MPO.heisenberg(L, 1.0, 0.9, 0.7, 0.2), p, noise)
    op = o.Params(observables=oobs, sample_timesteps=True, **kw)
    on = [o.make_process("pauli_x", [i], 0.05) for i in range(L)]
    idx = op.observable_sorted_indices
    biggest = 0
    for t in range(ntraj):
        r, dg, _ = o.run_trajectory(t, o.MPSState.product(L, "Neel"), on, op, o.heisenberg_mpo(L, 1.0, 0.9, 0.7, 0.2))
        for u in range(len(obs)):
            assert np.allclose(res.trajectories[u][t], r[idx[u]], atol=1e-8), (t, u)
        biggest = max(biggest, int(np.max(dg[1])))
    assert biggest > 16, "the case must outgrow the first capacity to mean anything"
    assert built[0] == 8 and built == sorted(built) and len(built) >= 2, built
    assert built[-1] >= min(biggest, 64) and built[-1] <= 64  # 2**(L//2) bounds every bond of a 12-site chain
    if max_bond == 24:
        assert built[-1] == 24 and biggest == 24


@pytest.mark.parametrize("native", [True, False])
@pytest.mark.parametrize("order,sample_timesteps", [(1, False), (2, True), (2, False)])
def test_growth_continues_mid_run_and_splits_pieces(monkeypatch, order, sample_timesteps, native):
    """A run that outgrows its storage is not started again: the step that was clipped is rolled back, the states move to
    engines of twice the capacity (here forced to hold fewer trajectories each, so the piece is split) and the run continues from
    that step with the random-stream cursors it had - order 1 and both phases of order 2, against the oracle per trajectory."""
    import yaqs_amd.tjm as tjm_mod
    from yaqs_amd.api import AnalogSimParams, MPO, MPS, NoiseModel, Observable, X as Xg, Z as Zg

    built = _recording_engine(monkeypatch)
    monkeypatch.setattr(tjm_mod.Simulator, "_batch_for", lambda self, remaining, length, chi, mpo, device: min(remaining, 5 if chi <= 8 else 2))
    starts = []
    orig_run = tjm_mod.BatchEngine.run

    def spy(self, **kw):
        starts.append(tuple(kw.get("start", (0, 0))))
        return orig_run(self, **kw)

    monkeypatch.setattr(tjm_mod.BatchEngine, "run", spy)
    L, ntraj = 10, 5
    obs = [Observable(Zg(), s) for s in range(L)] + [Observable(Xg(), 4)]
    oobs = [o.Obs(Z, s) for s in range(L)] + [o.Obs(X, 4)]
    kw = dict(elapsed_time=1.5, dt=0.1, max_bond_dim=None, svd_threshold=1e-10, krylov_tol=1e-11, order=order, random_seed=9)
    p = AnalogSimParams(observables=obs, num_traj=ntraj, sample_timesteps=sample_timesteps, **kw)
    noise = NoiseModel([{"name": "lowering", "sites": [i], "strength": 0.08} for i in range(L)])
    resumed = []
    orig_host = tjm_mod.TrajectoryBatch.run

    def spy_host(self, traj, initial, native=False, resume=None):
        if resume is not None:
            resumed.append(tuple(resume["start"]))
        return orig_host(self, traj, initial, native=native, resume=resume)

    monkeypatch.setattr(tjm_mod.TrajectoryBatch, "run", spy_host)
    res = tjm_mod.Simulator(native=native).run(MPS(L, state="Neel"), MPO.heisenberg(L, 1.0, 0.9, 0.7, 0.2), p, noise)
    op = o.Params(observables=oobs, sample_timesteps=sample_timesteps, **kw)
    on = [o.make_process("lowering", [i], 0.08) for i in range(L)]
    idx = op.observable_sorted_indices
    for t in range(ntraj):
        r, dg, _ = o.run_trajectory(t, o.MPSState.product(L, "Neel"), on, op, o.heisenberg_mpo(L, 1.0, 0.9, 0.7, 0.2))
        for u in range(len(obs)):
            assert np.allclose(res.trajectories[u][t], r[idx[u]], atol=1e-8), (t, u)
        assert np.array_equal(res.max_bond_trajectories[t], dg[1]) if hasattr(res, "max_bond_trajectories") else True
    assert built[0] == 8 and max(built) >= 16 and len(built) >= 4, built          # 5 trajectories -> pieces of 2, 2 and 1
    assert any(st[0] > 0 for st in resumed), resumed                              # some piece continued mid-run (either driver)
    if native:
        assert any(st[0] > 0 for st in starts), starts
    if order == 2:
        assert all(st[1] in (0, 1) for st in resumed)


def test_ensemble_mean_converges_to_the_lindblad_solution():
    """The physics the method exists for (cf. tests/analog/test_analog_tjm.py:323-379 of the reference, TJM against a dense solver
    within 0.03): the mean over 16384 trajectories of a 4-site dissipative Ising chain (amplitude damping and dephasing on every site)
    against the exact solution of the Lindblad master equation, exp(t Liouvillian) applied to the vectorised density matrix."""
    import scipy.linalg

    from yaqs_amd import AnalogSimParams, Hamiltonian, NoiseModel, Observable, Simulator, State
    from yaqs_amd.api import Z as Zg

    L, T, gamma, n = 4, 1.0, 0.1, 16384
    p = AnalogSimParams(observables=[Observable(Zg(), s) for s in range(L)], elapsed_time=T, dt=0.05, num_traj=n, max_bond_dim=16, svd_threshold=1e-10,
                        order=2, random_seed=7, sample_timesteps=False)
    noise = NoiseModel([{"name": name, "sites": [i], "strength": gamma} for i in range(L) for name in ("lowering", "pauli_z")])
    res = Simulator(show_progress=False).run(State(L, initial="x+"), Hamiltonian.ising(L, J=1.0, g=0.5), p, noise)
    got = np.array([res.expectation_values[s][-1] for s in range(L)])
    # dense Lindblad: d rho / dt = -i [H, rho] + sum_k gamma (L_k rho L_k^dag - {L_k^dag L_k, rho} / 2), site 0 = least significant index
    dim = 2 ** L
    H = o.mpo_to_matrix(o.ising_mpo(L, 1.0, 0.5))
    lower = np.array([[0, 1], [0, 0]], dtype=complex)

    def embed(m, s):
        return np.kron(np.eye(2 ** (L - 1 - s)), np.kron(m, np.eye(2 ** s)))

    jumps = [embed(m, s) for s in range(L) for m in (lower, Z)]
    eye = np.eye(dim)
    liouv = -1j * (np.kron(eye, H) - np.kron(H.T, eye))  # column-stacking vec: vec(A rho B) = (B^T kron A) vec(rho)
    for Lk in jumps:
        LdL = Lk.conj().T @ Lk
        liouv += gamma * (np.kron(Lk.conj(), Lk) - 0.5 * np.kron(eye, LdL) - 0.5 * np.kron(LdL.T, eye))
    psi = o.MPSState.product(L, "x+").to_vec()
    rho = (scipy.linalg.expm(T * liouv) @ np.outer(psi, psi.conj()).reshape(-1, order="F")).reshape(dim, dim, order="F")
    exact = np.array([np.real(np.trace(rho @ embed(Z, s))) for s in range(L)])
    assert abs(np.trace(rho) - 1.0) < 1e-10
    spread = np.array([np.std(res.trajectories[s][:, -1]) for s in range(L)]) / np.sqrt(n)
    assert np.all(np.abs(got - exact) < 5 * spread + 2e-3), (got, exact, spread)  # 5 sigma of the mean + the O(dt^2) splitting error
    assert np.max(np.abs(exact)) > 0.05  # the dynamics is not trivial


def test_reruns_are_bit_identical_and_independent_of_the_batching():
    """tests/core/test_random_utils.py:72-100 and tests/test_simulator.py:87-117 of the reference (bit-identical reruns, parallel equals
    serial): here the pool is the batch axis - the same seed gives the same bits on a rerun, and a trajectory's numbers do not depend on
    how many others share its launches or on the chunk it lands in."""
    from yaqs_amd import AnalogSimParams, Hamiltonian, NoiseModel, Observable, Simulator, State
    from yaqs_amd.api import Z as Zg

    L = 6
    p = AnalogSimParams(observables=[Observable(Zg(), s) for s in range(L)], elapsed_time=0.5, dt=0.1, num_traj=11, max_bond_dim=8, svd_threshold=1e-10,
                        order=2, random_seed=42)
    noise = NoiseModel([{"name": n, "sites": [i], "strength": 0.1} for i in range(L) for n in ("lowering", "pauli_z")])
    H = Hamiltonian.ising(L, J=1.0, g=0.5)
    runs = [np.stack(Simulator(batch=b, show_progress=False).run(State(L, initial="x+"), H, p, noise).trajectories) for b in (11, 11, 3, 7, 1)]
    assert np.array_equal(runs[0], runs[1])                      # rerun: atol = 0
    for other in runs[2:]:
        assert np.array_equal(runs[0], other)                    # 11 at once, chunks of 3 / 7, one by one: the same bits


def test_piecewise_hamiltonian_through_the_reference_style_factory():
    """Hamiltonian.piecewise([(H, duration), ...]) (hamiltonian.py:179-230) equals the tuple-of-MPOs form."""
    from yaqs_amd.api import AnalogSimParams, Hamiltonian, MPO, Observable, State, Z as Zg
    from yaqs_amd.tjm import Simulator

    L = 5
    p = AnalogSimParams(observables=[Observable(Zg(), s) for s in range(L)], elapsed_time=0.5, dt=0.1, max_bond_dim=8, svd_threshold=1e-10,
                        krylov_tol=1e-10, sample_timesteps=False)
    a, b = Hamiltonian.ising(L, 1.0, 0.5), Hamiltonian.ising(L, 1.0, 1.3)
    r1 = Simulator().run(State(L, initial="zeros"), Hamiltonian.piecewise([(a, 0.2), (b, 0.3)]), p)
    r2 = Simulator().run(State(L, initial="zeros"), (a, a, b, b, b), p)
    assert np.allclose(np.stack(r1.expectation_values), np.stack(r2.expectation_values), atol=1e-12)
    with pytest.raises(ValueError):
        Simulator().run(State(L, initial="zeros"), Hamiltonian.piecewise([(a, 0.25), (b, 0.25)]), p)


def test_run_dispatches_on_the_parameter_type_like_the_reference():
    """Simulator.run(state, operator, sim_params, noise) is the one entry point of the reference (simulator.py:1173-1312): with
    DigitalSimParams the operator is the circuit (here: gate layers) and the call is the circuit run."""
    from yaqs_amd.api import DigitalSimParams, MPS, NoiseModel, Observable, Z as Zg, ising_trotter_layers
    from yaqs_amd.tjm import Simulator

    L = 5
    layers = ising_trotter_layers(L, 1.0, 0.5, 0.1, 2)
    noise = NoiseModel([{"name": "pauli_x", "sites": [i], "strength": 0.05} for i in range(L)])
    p = DigitalSimParams(observables=[Observable(Zg(), s) for s in range(L)], num_traj=3, max_bond_dim=8, svd_threshold=1e-10, random_seed=2)
    a = Simulator().run(MPS(L, state="zeros"), layers, p, noise)
    b = Simulator().run_circuit(MPS(L, state="zeros"), layers, p, noise)
    assert np.array_equal(np.stack(a.trajectories), np.stack(b.trajectories))
    with pytest.raises(NotImplementedError):
        Simulator().run(MPS(L, state="zeros"), layers, p, noise, num_traj=5)
    with pytest.raises(ValueError, match="qubit counts do not match"):  # tests/test_simulator.py:838-855
        Simulator().run(MPS(L - 1, state="zeros"), layers, p, noise)
    with pytest.raises(NotImplementedError):
        Simulator().run([MPS(L, state="zeros")], layers, p, noise)
    with pytest.raises(NotImplementedError):
        Simulator().run(MPS(L, state="zeros"), "OPENQASM 2.0;", p, noise)
    with pytest.raises(TypeError):
        Simulator().run("not a state", layers, p, noise)


def test_circuit_growth_continues_at_the_clipped_layer(monkeypatch):
    """run_circuit with storage grown on demand: the layer whose truncation was clipped is rolled back and repeated on engines of
    twice the capacity (split into smaller batches here), mid-circuit sampling columns and the jump streams carry over; per
    trajectory equal to the oracle, shot counts complete."""
    import yaqs_amd.tjm as tjm_mod
    from yaqs_amd.api import DigitalSimParams, GateLayer, MPS, NoiseModel, Observable, X as Xg, Z as Zg

    built = _recording_engine(monkeypatch)
    monkeypatch.setattr(tjm_mod.Simulator, "_batch_for", lambda self, remaining, length, chi, mpo, device: min(remaining, 4 if chi <= 8 else 3))
    rng = np.random.default_rng(99)

    def haar(n):
        q, r = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
        return q * (np.diag(r) / np.abs(np.diag(r)))

    L, ntraj, n_layers = 10, 4, 6
    layers, olayers = [], []
    for k in range(n_layers):
        singles = [(q, haar(2)) for q in range(L)]
        even = [(q, q + 1, haar(4)) for q in range(0, L - 1, 2)]
        odd = [(q, q + 1, haar(4)) for q in range(1, L - 1, 2)]
        layers.append(GateLayer(singles, even, odd, 1 if k in (1, 3) else 0))
        olayers.append(o.GateLayer(singles, even, odd, 1 if k in (1, 3) else 0))
    procs = [{"name": name, "sites": [i], "strength": 0.05} for i in range(L) for name in ("pauli_x", "lowering")]
    obs = [Observable(Zg(), s) for s in range(L)] + [Observable(Xg(), 2)]
    oobs = [o.Obs(Z, s) for s in range(L)] + [o.Obs(X, 2)]
    kw = dict(max_bond_dim=None, svd_threshold=1e-9, random_seed=21, sample_layers=True, num_mid_measurements=2)
    res = tjm_mod.Simulator().run_circuit(MPS(L, state="zeros"), layers, DigitalSimParams(observables=obs, num_traj=ntraj, shots=40, **kw),
                                          NoiseModel(procs))
    on = [o.make_process(q["name"], q["sites"], q["strength"]) for q in procs]
    op = o.DigitalParams(observables=oobs, **kw)
    order_ = sorted(range(len(oobs)), key=lambda i: (oobs[i].first_site, i))  # user index -> row of the site-sorted buffer
    idx = [order_.index(u) for u in range(len(oobs))]
    biggest = 0
    for t in range(ntraj):
        ro, do, _ = o.digital_tjm(t, o.MPSState.product(L, "zeros"), on, op, olayers)
        for u in range(len(obs)):
            assert np.allclose(res.trajectories[u][t], ro[idx[u]], atol=1e-8), (t, u)
        biggest = max(biggest, int(np.max(do[1])))
    assert biggest > 8 and built[0] == 8 and max(built) >= 16, (biggest, built)
    assert sum(res.counts.values()) == 40


def test_general_kernels_still_serve_small_bonds():
    """Small bonds take the fused one-kernel centre shifts, site QR, two-site split and Krylov exponential and the LDS-resident
    Jacobi; with the switches below the same cases run on the general GEMM / Jacobi / Householder path (the switches are read once
    per process, hence the child interpreter)."""
    import subprocess
    import sys

    from conftest import ROOT

    env = dict(os.environ, TJM_NO_SMALL_SHIFT="1", TJM_NO_SMALL_KRYLOV="1", TJM_NO_LDS_JACOBI="1", TJM_NO_SWEEP_FUSION="1", TJM_FUZZ_CASES="10")
    out = subprocess.run([sys.executable, "-m", "pytest", "tests/test_hip_engine.py", "-m", "gpu", "-x", "-q", "-k",
                          "randomised_configurations or randomised_circuits or tiny_chains"], cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=1500)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout


def _channel_switch_points(noise, chi, u2):
    """One forced jump per trajectory from |00> (dissipation with dt = 1, then a jump test with u1 = 0) with the channel draw u2:
    returns <Z_0> after the jump, from which the chosen channel can be read."""
    from yaqs_amd.api import MPO, MPS, is_pauli

    L = 2
    st = MPS(L, state="zeros")
    e = make_engine(L, chi, len(u2), MPO.ising(L, 1.0, 0.5).tensors)
    try:
        e.set_params(dt=1.0, svd_threshold=1e-12, max_bond_dim=chi, krylov_tol=1e-10)
        e.set_noise(noise.processes, [is_pauli(q) for q in noise.processes])
        e.load_state(st.tensors, 0)
        e.dissipate(1.0)
        e.set_uniforms(np.stack([np.zeros(len(u2)), np.asarray(u2)], axis=1))
        jumped, dp = e.stochastic(1.0)
        assert np.all(jumped == 1) and np.all(dp > 0.5)
        m = e.site_moments(0)
        return np.real(m[0, :, 0, 0] - m[0, :, 1, 1])
    finally:
        e.close()


def test_channel_weights_of_the_reference_known_answer_tests():
    """tests/core/methods/test_stochastic_process.py:264-297 of the reference, through the engine: X on site 0 and 2 I on the pair
    (0, 1) at equal rates from |00> have weights ||X|0>||^2 : ||2 I |00>||^2 = 1 : 4, i.e. [0.2, 0.8]; rng.choice switches channel
    where the draw crosses the cumulative weight (searchsorted, side = "right").  The xx / Bell-creation pair has [0.5, 0.5] - the
    weight of an adjacent non-Pauli channel is the Frobenius norm of the UNTRUNCATED block."""
    from yaqs_amd.api import NoiseModel

    u2 = np.array([0.0, 0.1, 0.19, 0.1999999, 0.2000001, 0.21, 0.5, 0.8, 0.999])
    noise = NoiseModel([{"name": "pauli_x", "sites": [0], "strength": 1.0},
                        {"name": "scaled_i", "sites": [0, 1], "strength": 1.0, "matrix": 2.0 * np.eye(4, dtype=np.complex128)}])
    z0 = _channel_switch_points(noise, 2, u2)
    assert np.allclose(z0, np.where(u2 < 0.2, -1.0, 1.0), atol=1e-10), z0  # X flips qubit 0, 2 I leaves |00>
    xx = np.kron(np.array([[0, 1], [1, 0]]), np.array([[0, 1], [1, 0]])).astype(np.complex128)
    bell = np.zeros((4, 4), dtype=np.complex128)
    bell[:, 0] = [1 / np.sqrt(2), 0, 0, 1 / np.sqrt(2)]
    bell[:, 1] = [0, 1, 0, 0]
    bell[:, 2] = [0, 0, 1, 0]
    bell[:, 3] = [1 / np.sqrt(2), 0, 0, -1 / np.sqrt(2)]
    noise = NoiseModel([{"name": "xx", "sites": [0, 1], "strength": 1.0, "matrix": xx},
                        {"name": "bell", "sites": [0, 1], "strength": 1.0, "matrix": bell}])
    u2 = np.array([0.0, 0.3, 0.4999999, 0.5000001, 0.7, 0.999])
    z0 = _channel_switch_points(noise, 2, u2)
    assert np.allclose(z0, np.where(u2 < 0.5, -1.0, 0.0), atol=1e-10), z0  # xx |00> = |11>;  Bell state: <Z_0> = 0


@pytest.mark.parametrize("entangled", [False, True])
def test_jump_probability_after_dissipation_matches_the_dense_master_equation_step(entangled):
    """tests/analog/test_analog_tjm.py:257-279 of the reference: for lowering noise on every site (H = 0) the norm lost in the
    dissipative sweep, dp = 1 - ||psi||^2, is the jump probability of one dense quantum-jump step, 1 - ||exp(-dt/2 sum gamma
    L^dag L) psi||^2 - from the product state |1...1> and from an entangled state (one TDVP step of the Ising chain).  The
    reference allows 5e-4; the one-site dissipators commute, so the two agree to rounding."""
    import scipy.linalg

    from yaqs_amd.api import MPO, MPS, NoiseModel, is_pauli

    L, dt, gamma = 5, 0.05, 1.0
    st = o.MPSState.product(L, "ones")
    if entangled:
        o.tdvp(st, o.ising_mpo(L, 1.0, 0.5), o.Params(elapsed_time=0.0, dt=dt, max_bond_dim=64, svd_threshold=1e-10))
    psi = st.to_vec()
    psi = psi / np.linalg.norm(psi)
    lower = np.array([[0, 1], [0, 0]], dtype=np.complex128)
    local = scipy.linalg.expm(-0.5 * dt * gamma * lower.conj().T @ lower)
    prop = np.array([[1.0]])
    for _ in range(L):
        prop = np.kron(prop, local)
    p_dense = 1.0 - np.linalg.norm(prop @ psi) ** 2
    noise = NoiseModel([{"name": "lowering", "sites": [i], "strength": gamma} for i in range(L)])
    tensors = [t.copy() for t in st.tensors]
    e = make_engine(L, 8, 2, MPO.ising(L, 1.0, 0.5).tensors)
    try:
        e.set_params(dt=dt, svd_threshold=1e-10, max_bond_dim=8, krylov_tol=1e-10)
        e.set_noise(noise.processes, [is_pauli(q) for q in noise.processes])
        e.load_state(MPS(L, tensors=tensors).tensors, 0)
        e.dissipate(dt)
        e.set_uniforms(np.ones((2, 2)))  # u = 1 >= dp: no jump, dp is only read
        jumped, dp = e.stochastic(dt)
    finally:
        e.close()
    assert not jumped.any()
    assert p_dense > 0.1 and np.allclose(dp, p_dense, rtol=0.0, atol=1e-10), (dp, p_dense)


@pytest.mark.parametrize("adjacent", [False, True])
def test_noncommuting_channels_use_one_exponential_of_the_summed_generator(adjacent):
    """tests/core/methods/test_dissipation.py:212-295 of the reference, through the engine: two channels on the same site (or the
    same adjacent pair) whose L^dag L do not commute are applied as ONE expm(-dt/2 sum gamma L^dag L), independent of their order
    in the noise model; the dissipated two-site state equals the dense result."""
    import scipy.linalg

    from yaqs_amd.api import MPO, MPS, NoiseModel, is_pauli

    lowering = np.array([[0, 1], [0, 0]], dtype=np.complex128)
    mixed = np.array([[0, 1], [1, 1]], dtype=np.complex128)
    eye = np.eye(2, dtype=np.complex128)
    la, lb = (np.kron(lowering, eye), np.kron(mixed, eye)) if adjacent else (lowering, mixed)
    a, b = la.conj().T @ la, lb.conj().T @ lb
    assert not np.allclose(a @ b, b @ a)
    gamma_a, gamma_b, dt = 0.4, 0.3, 0.2
    sites = [0, 1] if adjacent else [0]
    fwd = [{"name": "a", "sites": sites, "strength": gamma_a, "matrix": la}, {"name": "b", "sites": sites, "strength": gamma_b, "matrix": lb}]
    amp = np.array([1 / np.sqrt(2), 1 / np.sqrt(2)], dtype=np.complex128)
    t0 = amp.reshape(2, 1, 1)
    t1 = (amp if adjacent else np.array([1.0, 0.0], dtype=np.complex128)).reshape(2, 1, 1)
    vec = np.kron(amp, t1.reshape(2))
    gen = gamma_a * a + gamma_b * b
    expected = (scipy.linalg.expm(-0.5 * dt * gen) @ vec) if adjacent else np.kron(scipy.linalg.expm(-0.5 * dt * gen) @ amp, t1.reshape(2))
    got = []
    for procs in (fwd, fwd[::-1]):
        noise = NoiseModel(procs)
        e = make_engine(2, 4, 1, MPO.ising(2, 1.0, 0.5).tensors)
        try:
            e.set_params(dt=dt, svd_threshold=1e-14, max_bond_dim=4, krylov_tol=1e-10)
            e.set_noise(noise.processes, [is_pauli(q) for q in noise.processes])
            e.load_state(MPS(2, tensors=[t0.copy(), t1.copy()]).tensors, 0)
            e.dissipate(dt)
            ts = e.export_state(0, 0)
        finally:
            e.close()
        got.append(np.einsum("alr,brc->ab", ts[0], ts[1]).reshape(-1))
    assert np.allclose(got[0], got[1], atol=1e-12)
    assert np.allclose(got[0], expected, atol=1e-10), (got[0], expected)


@pytest.mark.parametrize("case", range(int(os.environ.get("TJM_FUZZ_GROWTH_CASES", "12"))))
def test_randomised_front_end_runs_with_growing_storage_match_oracle(case, monkeypatch):
    """Differential test of the whole front end on seeded random set-ups that outgrow their first storage capacity: random chain
    length, model, noise, order, sampling mode, truncation settings and max_bond_dim in {None, 4096, a binding cap}, forced piece
    splitting in half of the cases; per trajectory against the oracle, bond diagnostics included."""
    import yaqs_amd.tjm as tjm_mod
    from yaqs_amd import AnalogSimParams, Hamiltonian, NoiseModel, Observable, Simulator, State
    from yaqs_amd.api import X as Xg, Z as Zg

    rng = np.random.default_rng(9100 + case)
    L = int(rng.integers(8, 12))
    heis = bool(rng.integers(0, 2))
    order = int(rng.integers(1, 3))
    sample_timesteps = bool(rng.integers(0, 2))
    max_bond = [None, 4096, int(rng.integers(9, 20))][int(rng.integers(0, 3))]
    steps = int(rng.integers(8, 15))
    ntraj = int(rng.integers(2, 5))
    gamma = float(rng.uniform(0.02, 0.15))
    name = str(rng.choice(["lowering", "pauli_x", "pauli_z", "raising"]))
    if rng.integers(0, 2):
        monkeypatch.setattr(tjm_mod.Simulator, "_batch_for", lambda self, remaining, length, chi, mpo, device: min(remaining, 4 if chi <= 8 else 2))
    built = _recording_engine(monkeypatch)
    kw = dict(elapsed_time=0.1 * steps, dt=0.1, max_bond_dim=max_bond, svd_threshold=float(10.0 ** rng.uniform(-11, -8)), krylov_tol=1e-10, order=order,
              random_seed=int(rng.integers(0, 10 ** 6)))
    obs = [Observable(Zg(), s) for s in range(L)] + [Observable(Xg(), int(rng.integers(0, L)))]
    oobs = [o.Obs(Z, s) for s in range(L)] + [o.Obs(X, obs[-1].sites)]
    H = Hamiltonian.heisenberg(L, 1.0, 0.9, 0.7, 0.2) if heis else Hamiltonian.ising(L, 1.0, 0.8)
    Ho = o.heisenberg_mpo(L, 1.0, 0.9, 0.7, 0.2) if heis else o.ising_mpo(L, 1.0, 0.8)
    init = str(rng.choice(["Neel", "x+", "wall"]))
    p = AnalogSimParams(observables=obs, num_traj=ntraj, sample_timesteps=sample_timesteps, **kw)
    noise = NoiseModel([{"name": name, "sites": [i], "strength": gamma} for i in range(L)])
    res = Simulator(show_progress=False).run(State(L, initial=init), H, p, noise)
    op = o.Params(observables=oobs, sample_timesteps=sample_timesteps, **kw)
    on = [o.make_process(name, [i], gamma) for i in range(L)]
    idx = op.observable_sorted_indices
    for t in range(ntraj):
        r, dg, _ = o.run_trajectory(t, o.MPSState.product(L, init), on, op, Ho)
        for u in range(len(obs)):
            assert np.allclose(res.trajectories[u][t], r[idx[u]], atol=1e-8), (case, t, u)
        assert np.array_equal(res.trajectory_diagnostics[t], dg), (case, t)
    assert built[0] == 8 and max(built) <= 64, built  # every run starts small; 2**(L//2) bounds what it can ever need


def test_capacity_overflow_is_reported_by_the_engine_and_the_driver():
    """A two-site truncation that wants more values than the new bond stores sets the engine's flag (and only such a one), and
    tjm_engine_run stops after that time step with TJM_ERR_CAPACITY instead of finishing a run that is not the reference's."""
    from yaqs_amd._lib import CapacityError
    from yaqs_amd.api import MPO, MPS
    from yaqs_amd.engine import BatchEngine

    L = 8
    mpo = MPO.heisenberg(L, 1.0, 0.9, 0.7, 0.2).tensors
    init = MPS(L, state="Neel")
    init.normalize("B")
    e = BatchEngine(L, 4, 2, mpo)
    try:
        e.set_params(dt=0.1, svd_threshold=1e-12, max_bond_dim=64, krylov_tol=1e-10)
        e.set_noise([], [])
        e.load_state(init.tensors, 0)
        assert not e.capacity_overflow()
        e.tdvp(0)  # bonds 1 -> at most 4 with threshold 1e-12: nothing is clipped in the first step
        first = e.capacity_overflow()
        for _ in range(6):
            e.tdvp(0)
        assert e.capacity_overflow() and np.max(e.bond_dims(0)) == 4
        assert e.capacity_overflow(clear=True) and not e.capacity_overflow()
        e.load_state(init.tensors, 0)
        with pytest.raises(CapacityError):
            e.run(order=1, n_times=12, sample_timesteps=False, has_noise=False, seed=1, traj_indices=[0, 1],
                  observables=[(0, np.diag([1.0, -1.0]))])
        # with max_bond_dim = capacity the clip IS the requested truncation: no flag
        e.set_params(dt=0.1, svd_threshold=1e-12, max_bond_dim=4, krylov_tol=1e-10)
        e.load_state(init.tensors, 0)
        e.capacity_overflow(clear=True)
        for _ in range(6):
            e.tdvp(0)
        assert not e.capacity_overflow()
        assert first in (False, True)
    finally:
        e.close()


def test_default_presets_run_on_long_chains(monkeypatch):
    """The "accurate" (4096) and "exact" (None) presets on a 40-site chain: refused before (static capacity), now served within
    the capacity the run really needs and equal to the oracle."""
    from yaqs_amd.api import AnalogSimParams, MPO, MPS, Observable, Z as Zg
    from yaqs_amd.tjm import Simulator

    built = _recording_engine(monkeypatch)
    L = 40
    for max_bond in (None, 4096):
        p = AnalogSimParams(observables=[Observable(Zg(), s) for s in (0, 17, 39)], elapsed_time=0.3, dt=0.1, max_bond_dim=max_bond,
                            svd_threshold=1e-9, krylov_tol=1e-10, sample_timesteps=False)
        res = Simulator().run(MPS(L, state="x+"), MPO.ising(L, 1.0, 0.5), p)
        op = o.Params(observables=[o.Obs(Z, s) for s in (0, 17, 39)], elapsed_time=0.3, dt=0.1, max_bond_dim=max_bond, svd_threshold=1e-9,
                      krylov_tol=1e-10, sample_timesteps=False)
        r, _, _ = o.run_trajectory(0, o.MPSState.product(L, "x+"), [], op, o.ising_mpo(L, 1.0, 0.5))
        for u in range(3):
            assert abs(res.expectation_values[u][0] - r[op.observable_sorted_indices[u]][0]) < 1e-8
    assert max(built) <= 32, built


def test_simulator_normalises_the_initial_state_like_the_reference():
    """Simulator.run brings the given MPS to B-normal form on a copy (state.py:278-297): a left-canonical, unnormalised Haar state
    gives the same results as handing over its normalised form, and the caller's tensors stay untouched."""
    from yaqs_amd.api import AnalogSimParams, MPO, MPS, Observable, Z as Zg
    from yaqs_amd.tjm import Simulator

    L, chi = 6, 8
    raw = MPS(L, state="haar-random", pad=chi, rng=np.random.default_rng(3))
    raw.tensors[2] = 1.7 * raw.tensors[2]
    keep = [t.copy() for t in raw.tensors]
    p = AnalogSimParams(observables=[Observable(Zg(), s) for s in range(L)], elapsed_time=0.2, dt=0.1, max_bond_dim=chi, svd_threshold=1e-12,
                        krylov_tol=1e-12, order=1, sample_timesteps=True)
    a = Simulator().run(raw, MPO.ising(L, 1.0, 0.5), p)
    assert all(np.array_equal(x, y) for x, y in zip(keep, raw.tensors))
    st = o.MPSState([t.copy() for t in keep], None)
    st.normalize("B")
    op = o.Params(observables=[o.Obs(Z, s) for s in range(L)], elapsed_time=0.2, dt=0.1, max_bond_dim=chi, svd_threshold=1e-12, krylov_tol=1e-12,
                  order=1, sample_timesteps=True)
    r, _, _ = o.run_trajectory(0, st, None, op, o.ising_mpo(L, 1.0, 0.5))
    for s in range(L):
        assert np.allclose(a.trajectories[s][0], r[s], atol=1e-9)


@pytest.mark.parametrize("case", range(int(os.environ.get("TJM_FUZZ_CASES", "120"))))
def test_randomised_configurations_match_oracle(case):
    """Differential test on seeded random set-ups: chain length, bond cap, truncation mode and threshold, TDVP mode and substeps,
    driver order, initial state, and a noise model mixing Pauli / non-Pauli one-site, adjacent two-site (Pauli and custom) and
    long-range Pauli channels.  Every trajectory must agree with the oracle to 1e-8 and reproduce its bond dimensions."""
    from yaqs_amd.api import AnalogSimParams, MPO, MPS, NoiseModel, Observable, X as Xg, Y as Yg, Z as Zg

    rng = np.random.default_rng(1000 + case)
    L = int(rng.integers(2, 8))
    chi = int(rng.choice([2, 4, 8]))
    order = int(rng.choice([1, 2]))
    mode = str(rng.choice(["2site", "2site", "1site"]))
    sweeps = int(rng.choice([1, 1, 2]))
    trunc = str(rng.choice(["discarded_weight", "relative", "hard_cutoff", "relative_discarded_weight"]))
    thr = float(10.0 ** rng.uniform(-12, -5))
    state = str(rng.choice(["zeros", "x+", "y-", "Neel", "haar"]))
    one_site = ["lowering", "raising", "pauli_x", "pauli_y", "pauli_z"]
    procs = []
    for i in range(L):
        for name in rng.choice(one_site, size=int(rng.integers(0, 3)), replace=False):
            procs.append({"name": str(name), "sites": [i], "strength": float(rng.uniform(0.02, 0.4))})
    if L >= 3 and rng.random() < 0.7:
        i = int(rng.integers(0, L - 1))
        procs.append({"name": "crosstalk_xz", "sites": [i, i + 1], "strength": float(rng.uniform(0.05, 0.3))})
    if L >= 3 and rng.random() < 0.5:
        i = int(rng.integers(0, L - 1))
        m = np.kron(o.JUMP_OPS["lowering"], np.array([[1, 0], [0, -1]])) + 0.3 * np.kron(np.eye(2), o.JUMP_OPS["raising"])
        procs.append({"name": "custom", "sites": [i, i + 1], "strength": float(rng.uniform(0.05, 0.3)), "matrix": m})
    if L >= 4 and rng.random() < 0.5:
        procs.append({"name": "crosstalk_zy", "sites": [0, L - 1], "strength": float(rng.uniform(0.05, 0.3))})
    if not procs:
        procs.append({"name": "pauli_z", "sites": [0], "strength": 0.2})
    noise = NoiseModel(procs)
    if state == "haar":
        st = o.MPSState.haar(L, chi, np.random.default_rng(case))
        st.normalize("B")
        init = [t.copy() for t in st.tensors]
    else:
        init = MPS(L, state=state).tensors
    gates = {"x": (Xg, X), "y": (Yg, o.PAULI["y"]), "z": (Zg, Z)}
    picks = [(str(rng.choice(list(gates))), int(rng.integers(0, L))) for _ in range(4)]
    obs = [Observable(gates[g_][0](), s) for g_, s in picks]
    oobs = [o.Obs(gates[g_][1], s) for g_, s in picks]
    kw = dict(elapsed_time=0.3, dt=0.1, max_bond_dim=chi, svd_threshold=thr, trunc_mode=trunc, krylov_tol=1e-11, order=order,
              sample_timesteps=bool(rng.integers(0, 2)), random_seed=int(rng.integers(0, 10 ** 6)), tdvp_mode=mode, tdvp_sweeps=sweeps)
    mpo = MPO.heisenberg(L, 1.0, 0.7, 0.4, 0.25) if rng.random() < 0.5 else MPO.ising(L, 1.0, 0.6)
    r, d, _ = _run(L, init, noise, AnalogSimParams(observables=obs, **kw), mpo.tensors, [0, 1, 2], native=bool(rng.integers(0, 2)))
    op = o.Params(observables=oobs, **kw)
    on = [o.make_process(q["name"], q["sites"], q["strength"], matrix=q.get("matrix"), factors=q.get("factors")) for q in noise.processes]
    for t in range(3):
        ro, do, _ = o.run_trajectory(t, o.MPSState([x.copy() for x in init], 0), on, op, [w.copy() for w in mpo.tensors])
        assert np.allclose(r[t], ro, atol=1e-8), (case, t, np.abs(r[t] - ro).max(), kw, [q["name"] for q in procs])
        assert np.array_equal(d[t], do), (case, t)


@pytest.mark.parametrize("case", range(int(os.environ.get("TJM_FUZZ_CASES", "80"))))
def test_randomised_circuits_match_oracle(case):
    """Differential test of the circuit path on seeded random circuits: random one-qubit unitaries, random two-qubit unitaries on
    adjacent and distant pairs in both site orders, local noise with Pauli / non-Pauli / adjacent two-site / long-range channels,
    optional mid-circuit sampling points."""
    from yaqs_amd.api import DigitalSimParams, GateLayer, MPS, NoiseModel, Observable, X as Xg, Z as Zg
    from yaqs_amd.tjm import DigitalBatch

    rng = np.random.default_rng(7000 + case)
    L = int(rng.integers(3, 8))
    chi = int(rng.choice([2, 4, 8]))
    n_layers = int(rng.integers(1, 4))
    sample = bool(rng.integers(0, 2))

    def haar(n):
        q, r = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
        return q * (np.diag(r) / np.abs(np.diag(r)))

    layers, olayers = [], []
    for _ in range(n_layers):
        singles = [(int(q), haar(2)) for q in rng.choice(L, size=int(rng.integers(0, L + 1)), replace=False)]
        groups = []
        for _g in range(2):
            grp = []
            sites = list(rng.permutation(L))
            while len(sites) >= 2 and rng.random() < 0.7:
                a, b = int(sites.pop()), int(sites.pop())
                grp.append((a, b, haar(4)))
            groups.append(grp)
        sp_ = int(rng.integers(0, 2)) if sample else 0
        layers.append(GateLayer(singles, groups[0], groups[1], sp_))
        olayers.append(o.GateLayer(singles, groups[0], groups[1], sp_))
    mid = sum(l.sample_points for l in layers)
    procs = []
    for i in range(L):
        for name in rng.choice(["lowering", "pauli_x", "pauli_y", "pauli_z", "raising"], size=int(rng.integers(0, 3)), replace=False):
            procs.append({"name": str(name), "sites": [i], "strength": float(rng.uniform(0.01, 0.3))})
    if rng.random() < 0.6:
        i = int(rng.integers(0, L - 1))
        procs.append({"name": "crosstalk_zx", "sites": [i, i + 1], "strength": float(rng.uniform(0.05, 0.3))})
    if L >= 4 and rng.random() < 0.6:
        procs.append({"name": "crosstalk_yy", "sites": [0, L - 1], "strength": float(rng.uniform(0.05, 0.3))})
    noise = NoiseModel(procs) if procs and rng.random() < 0.85 else None
    obs = [Observable(Zg(), s) for s in range(L)] + [Observable(Xg(), int(rng.integers(0, L)))]
    oobs = [o.Obs(Z, s) for s in range(L)] + [o.Obs(X, obs[-1].sites)]
    kw = dict(max_bond_dim=chi, svd_threshold=float(10.0 ** rng.uniform(-12, -6)), random_seed=int(rng.integers(0, 10 ** 6)), sample_layers=sample,
              num_mid_measurements=mid if sample else 0)
    e = make_engine(L, chi, 3, o.ising_mpo(L, 1.0, 0.5))
    db = DigitalBatch(e, DigitalSimParams(observables=obs, gate_mode="swaps", **kw), noise)
    r, d = db.run([0, 1, 2], MPS(L, state="zeros"), layers)
    e.close()
    on = None if noise is None else [o.make_process(q["name"], q["sites"], q["strength"], matrix=q.get("matrix"), factors=q.get("factors"))
                                     for q in noise.processes]
    op = o.DigitalParams(observables=oobs, **kw)
    for t in range(3):
        ro, do, _ = o.digital_tjm(t, o.MPSState.product(L, "zeros"), on, op, olayers)
        assert np.allclose(r[t], ro, atol=1e-8), (case, t, np.abs(r[t] - ro).max())
        assert np.array_equal(d[t], do), (case, t)


@pytest.mark.parametrize("L,chi", [(12, 32), (14, 64), (16, 128), (12, 24), (14, 48), (16, 96), (10, 16), (10, 12)])
def test_medium_bond_dimensions_match_oracle(L, chi):
    """chi = 32 / 64 / 128: the two-site split is 64 / 128 / 256 square, so the doubly QR-preconditioned, accumulation-free path runs
    with the fused 16-column kernels (64, 128) and the split X kernel (256); chi-saturated Haar state, amplitude damping plus dephasing."""
    from yaqs_amd.api import AnalogSimParams, MPO, NoiseModel, Observable, Z as Zg

    rng = np.random.default_rng(L * 100 + chi)
    st = o.MPSState.haar(L, chi, rng)
    st.normalize("B")
    init = [t.copy() for t in st.tensors]
    noise = NoiseModel([{"name": n, "sites": [i], "strength": 0.08} for i in range(L) for n in ("lowering", "pauli_z")])
    kw = dict(elapsed_time=0.2, dt=0.1, max_bond_dim=chi, svd_threshold=1e-10, krylov_tol=1e-10, order=1, sample_timesteps=True, random_seed=77)
    mpo = MPO.ising(L, 1.0, 0.5)
    r, d, _ = _run(L, init, noise, AnalogSimParams(observables=[Observable(Zg(), s) for s in range(L)], **kw), mpo.tensors, [0, 1])
    on = [o.make_process(n, [i], 0.08) for i in range(L) for n in ("lowering", "pauli_z")]
    op = o.Params(observables=[o.Obs(Z, s) for s in range(L)], **kw)
    for t in range(2):
        ro, do, _ = o.run_trajectory(t, o.MPSState([x.copy() for x in init], 0), on, op, mpo.tensors)
        assert np.allclose(r[t], ro, atol=1e-8), (t, np.abs(r[t] - ro).max())
        assert np.array_equal(d[t], do), t


def test_error_behaviour_matches_reference_types():
    """Error mapping of the boundary: imaginary expectation value -> AssertionError (mps.py:1233) in both drivers, length mismatch ->
    ValueError (tdvp.py:91-93), non-Pauli long-range noise -> NotImplementedError (dissipation.py:136-138)."""
    from yaqs_amd.api import AnalogSimParams, MPO, MPS, NoiseModel, Observable
    from yaqs_amd.tjm import Simulator

    L = 4
    nonherm = np.array([[0, 1j], [0.5j, 0]], dtype=np.complex128)  # <x+| . |x+> = 0.75i
    p = AnalogSimParams(observables=[Observable(nonherm, 1)], elapsed_time=0.1, dt=0.1, max_bond_dim=4)
    for native in (True, False):
        with pytest.raises(AssertionError):
            Simulator(native=native).run(MPS(L, state="x+"), MPO.ising(L, 1.0, 0.5), p)
    with pytest.raises(ValueError):
        Simulator().run(MPS(L, state="x+"), MPO.ising(L + 1, 1.0, 0.5), p)
    from yaqs_amd.api import Z as Zg

    p2 = AnalogSimParams(observables=[Observable(Zg(), 0)], elapsed_time=0.1, dt=0.1, max_bond_dim=4)
    lr = NoiseModel([{"name": "custom", "sites": [0, 3], "strength": 0.2,
                      "factors": (np.array([[0, 1], [0, 0]], dtype=complex), np.array([[1, 0], [0, -1]], dtype=complex))}])
    # the front end refuses it in the run-context validation (noise_model.py:668-742 -> ValueError), the backend itself with the
    # reference's NotImplementedError (dissipation.py:136-138)
    with pytest.raises(ValueError, match="non-Pauli long-range"):
        Simulator().run(MPS(L, state="x+"), MPO.ising(L, 1.0, 0.5), p2, lr)
    from yaqs_amd.tjm import TrajectoryBatch

    eng = make_engine(L, 4, 1, MPO.ising(L, 1.0, 0.5).tensors)
    with pytest.raises(NotImplementedError):
        TrajectoryBatch(eng, p2, lr).run([0], MPS(L, state="x+"))
    eng.close()


def test_run_from_a_basis_state_matches_oracle():
    """A noisy run started from ``State(initial="basis", basis_string=...)`` against the oracle started from the same product state:
    the same per-trajectory observables with the same seeds."""
    from yaqs_amd.api import AnalogSimParams, MPO, NoiseModel, Observable, State, Z as Zg
    from yaqs_amd.tjm import Simulator

    L, bits = 6, "010011"
    noise = [{"name": "lowering", "sites": [s], "strength": 0.2} for s in range(L)]
    p = AnalogSimParams(observables=[Observable(Zg(), s) for s in range(L)], elapsed_time=0.4, dt=0.1, num_traj=3, max_bond_dim=8, svd_threshold=1e-12,
                        krylov_tol=1e-12, order=2, sample_timesteps=True, random_seed=11)
    a = Simulator().run(State(L, initial="basis", basis_string=bits), MPO.ising(L, 1.0, 0.7), p, NoiseModel(noise))
    assert np.allclose([a.trajectories[s][0][0] for s in range(L)], [1 - 2 * int(c) for c in bits], atol=1e-12)
    on = [o.make_process(q["name"], q["sites"], q["strength"]) for q in noise]
    op = o.Params(observables=[o.Obs(Z, s) for s in range(L)], elapsed_time=0.4, dt=0.1, max_bond_dim=8, svd_threshold=1e-12,
                  krylov_tol=1e-12, order=2, sample_timesteps=True, random_seed=11)
    for t in range(3):
        v = [np.eye(2)[int(c)].reshape(2, 1, 1).astype(complex) for c in bits]
        r, _, _ = o.run_trajectory(t, o.MPSState(v, 0), on, op, o.ising_mpo(L, 1.0, 0.7))
        for s in range(L):
            assert np.allclose(a.trajectories[s][t], r[s], atol=1e-8), (t, s)


def test_periodic_chain_matches_oracle_and_dense_evolution():
    """Closed periodic Ising ring of 5 sites (MPO.ising(..., bc="periodic")), full bond dimension: the HIP path against the oracle fed
    with the same MPO tensors (1e-8) and against exp(-iHt) on the dense ring Hamiltonian built from Kronecker products (2e-3: the closing
    bond is a long-range term for the chain, and a two-site sweep started from a product state carries a projection error there - the
    reference's own behaviour, which the oracle reproduces)."""
    import scipy.linalg as sla
    from yaqs_amd.api import AnalogSimParams, MPO, MPS, Observable, Z as Zg
    from yaqs_amd.tjm import Simulator

    L, J, gf, T = 5, 1.0, 0.8, 0.5
    mpo = MPO.ising(L, J, gf, bc="periodic")
    p = AnalogSimParams(observables=[Observable(Zg(), s) for s in range(L)], elapsed_time=T, dt=0.1, max_bond_dim=8, svd_threshold=1e-14,
                        krylov_tol=1e-12, order=2, sample_timesteps=False)
    start = MPS(L, state="wall")
    a = Simulator().run(start, mpo, p)
    got = np.array([a.expectation_values[s][-1] for s in range(L)])
    op = o.Params(observables=[o.Obs(Z, s) for s in range(L)], elapsed_time=T, dt=0.1, max_bond_dim=8, svd_threshold=1e-14, krylov_tol=1e-12,
                  order=2, sample_timesteps=False)
    r, _, _ = o.run_trajectory(0, o.MPSState([t.copy() for t in start.tensors], 0), None, op, [w.copy() for w in mpo.tensors])
    assert np.allclose(got, r[:, -1], atol=1e-8)

    def emb(s, m):
        out = np.ones((1, 1), dtype=complex)
        for q in range(L):
            out = np.kron(out, m if q == s else np.eye(2))
        return out

    H = sum(-J * emb(i, Z) @ emb((i + 1) % L, Z) for i in range(L)) + sum(-gf * emb(i, X) for i in range(L))
    psi = np.ones(1, dtype=complex)
    for s in range(L):
        psi = np.kron(psi, np.eye(2)[0 if s < L // 2 else 1])
    psi = sla.expm(-1j * T * H) @ psi
    want = np.array([np.real(psi.conj() @ emb(s, Z) @ psi) for s in range(L)])
    assert np.allclose(got, want, atol=2e-3), np.abs(got - want).max()
    b = Simulator().run(start, MPO.ising(L, J, gf), p)
    assert np.abs(got - np.array([b.expectation_values[s][-1] for s in range(L)])).max() > 1e-2  # the closing bond matters


def test_pauli_sum_hamiltonian_run_matches_oracle():
    """A noisy run under ``MPO().from_pauli_sum(...)`` with a next-nearest-neighbour and a three-site string (site-dependent MPO bond
    dimensions): per-trajectory observables against the oracle fed with the same MPO tensors."""
    from yaqs_amd.api import AnalogSimParams, MPO, MPS, NoiseModel, Observable, X as Xg, Z as Zg
    from yaqs_amd.tjm import Simulator

    L = 6
    terms = ([(-1.0, f"Z{i} Z{i + 1}") for i in range(L - 1)] + [(-0.6, f"X{i}") for i in range(L)] + [(0.4, f"X{i} X{i + 2}") for i in range(L - 2)]
             + [(0.25, "Y0 Z2 Y3")])
    mpo = MPO()
    mpo.from_pauli_sum(terms=terms, length=L)
    assert len({t.shape[3] for t in mpo.tensors[:-1]}) > 1
    noise = [{"name": "pauli_x", "sites": [s], "strength": 0.15} for s in range(L)]
    obs = [Observable(Zg(), s) for s in range(3)] + [Observable(Xg(), 2)] + [Observable(Zg(), s) for s in range(3, L)]  # in worker order
    p = AnalogSimParams(observables=obs, elapsed_time=0.4, dt=0.1, num_traj=3, max_bond_dim=8, svd_threshold=1e-12, krylov_tol=1e-12, order=2,
                        sample_timesteps=True, random_seed=3)
    start = MPS(L, state="Neel")
    a = Simulator().run(start, mpo, p, NoiseModel(noise))
    on = [o.make_process(q["name"], q["sites"], q["strength"]) for q in noise]
    op = o.Params(observables=[o.Obs(Z, s) for s in range(3)] + [o.Obs(X, 2)] + [o.Obs(Z, s) for s in range(3, L)], elapsed_time=0.4, dt=0.1,
                  max_bond_dim=8, svd_threshold=1e-12, krylov_tol=1e-12, order=2, sample_timesteps=True, random_seed=3)
    for t in range(3):
        r, _, _ = o.run_trajectory(t, o.MPSState([x.copy() for x in start.tensors], 0), on, op, [w.copy() for w in mpo.tensors])
        for u in range(len(obs)):
            assert np.allclose(a.trajectories[u][t], r[u], atol=1e-8), (t, u)


def test_output_state_answers_the_inspection_helpers():
    """``result.output_state`` of a closed run: unit norm, a valid canonical form, bond dimensions and entropies consistent with the
    dense vector, and <Z_s> of the returned state equal to the reported final expectation values."""
    from yaqs_amd.api import AnalogSimParams, MPO, MPS, Observable, Z as Zg
    from yaqs_amd.tjm import Simulator

    L = 6
    p = AnalogSimParams(observables=[Observable(Zg(), s) for s in range(L)], elapsed_time=0.5, dt=0.1, max_bond_dim=8, svd_threshold=1e-12,
                        krylov_tol=1e-12, order=2, sample_timesteps=False, get_state=True)
    res = Simulator().run(MPS(L, state="x+"), MPO.ising(L, 1.0, 0.5), p)
    out = res.output_state
    out.check_if_valid_mps()
    assert abs(out.norm() - 1) < 1e-10 and len(out.check_canonical_form()) >= 1
    assert out.get_total_bond() == sum(out.bond_dimensions()) and max(out.bond_dimensions()) <= 8
    for s in range(L):
        assert abs(out.expect(Observable(Zg(), s)) - res.expectation_values[s][-1]) < 1e-9
    c = out.check_canonical_form()[0]
    i = min(max(c, 0), L - 2) if c < L - 1 else L - 2
    vec = out.to_vec().reshape(2 ** (L - 1 - i), 2 ** (i + 1))  # rows: sites above the cut (site L-1 most significant)
    pr = np.linalg.svd(vec, compute_uv=False) ** 2
    pr = pr[pr > 1e-300]
    assert abs(out.get_entropy([i, i + 1]) - (-np.sum(pr * np.log(pr)))) < 1e-9
