"""Pin the CPU oracle (oracle/tjm_oracle.py) to fixtures generated from the reference.

Fixtures: tests/golden/*.npz, produced by tools/make_golden.py importing /root/reference.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import tjm_oracle as o

Z = o.PAULI["z"]
X = o.PAULI["x"]


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def tensors(g, prefix):
    out = []
    i = 0
    while f"{prefix}{i}" in g:
        out.append(g[f"{prefix}{i}"])
        i += 1
    return out


def phase_align(a, b):
    """Return b rotated by the global phase that best matches a."""
    ov = np.vdot(b, a)
    return b * (ov / abs(ov)) if abs(ov) > 0 else b


def test_rng_streams():
    g = load("rng_streams")
    for a, s in enumerate(g["seeds"]):
        for b, t in enumerate(g["trajs"]):
            assert np.array_equal(o.trajectory_rng(int(s), int(t)).random(8), g["traj"][a, b])
            for c, k in enumerate(g["steps"]):
                assert np.array_equal(o.sample_rng(int(s), int(t), int(k)).random(8), g["sample"][a, b, c])
    # first doubles quoted in SURVEY.md section 8(c)
    assert o.trajectory_rng(42, 0).random() == 0.8577260219363377
    assert o.sample_rng(42, 0, 1).random() == 0.6952053372292888


def test_choice_is_one_uniform_searchsorted():
    g = load("rng_streams")
    for case in range(len(g["choice_idx"])):
        p = g["choice_p"][case]
        p = p[p > 0]
        rng = o.trajectory_rng(7, case)
        rng.random()
        u = rng.random()
        cdf = np.cumsum(p)
        cdf /= cdf[-1]
        assert int(np.searchsorted(cdf, u, side="right")) == g["choice_idx"][case]
        assert rng.random() == g["choice_next"][case]


def test_truncate_kat_table():
    g = load("truncate_kat")
    modes = ["discarded_weight", "relative", "hard_cutoff", "relative_discarded_weight"]
    for row, spec in zip(g["spectra"], g["specs"]):
        n, mode, thr, cap, min_keep, keep = int(spec[0]), modes[int(spec[1])], spec[2], int(spec[3]), int(spec[4]), int(spec[5])
        got = o.truncate(row[:n], mode=mode, threshold=thr, max_bond_dim=None if cap < 0 else cap, min_keep=min_keep)
        assert got == keep


def test_truncate_reference_unit_cases():
    # hand-computed cases of tests/core/linalg/test_svd_utils.py:20-80
    t = o.truncate
    assert t(np.array([10.0, 3.0, 1.0, 0.5]), mode="discarded_weight", threshold=10.0) == 2
    assert t(np.array([2.0, 1.0, 0.4]), mode="relative", threshold=0.45) == 2
    assert t(np.array([0.0, 1.0]), mode="relative", threshold=0.1) == 1
    s = np.array([5.0, 2.0, 0.5, 0.1])
    assert t(s, mode="hard_cutoff", threshold=0.2) == 3
    assert t(s, mode="hard_cutoff", threshold=0.2, max_bond_dim=2) == 2
    s = np.array([10.0, 1.0, 0.1, 0.01])
    assert t(s, mode="relative_discarded_weight", threshold=1e-3) == 2
    assert t(s, mode="relative_discarded_weight", threshold=0.02) == 1
    assert t(np.array([3.0, 2.0, 0.0, 0.0]), mode="relative_discarded_weight", threshold=0.0) == 2
    assert t(np.zeros(4), mode="relative_discarded_weight", threshold=0.1) == 1
    # tests/core/methods/tdvp/test_sweep_utils.py:136-143
    assert t(np.array([1.0, 0.5, 0.1, 0.0100001]), mode="discarded_weight", threshold=1e-4, min_keep=2) == 4
    assert t(np.array([1.0, 0.5, 0.01, 0.001]), mode="discarded_weight", threshold=1e-4, min_keep=2) == 3
    with pytest.raises(ValueError):
        t(np.ones(3), mode="invalid", threshold=1.0)


def test_local_kernels():
    g = load("kernels")
    a, b = g["A"], g["B"]
    assert np.allclose(o.merge_two_site(a, b), g["merge"], atol=1e-13)
    w = o.merge_mpo_tensors(g["W1"], g["W2"])
    assert np.allclose(w, g["merge_mpo"], atol=1e-13)
    assert np.allclose(o.project_site(g["L"], g["R"], w, g["merge"]), g["project_site_2"], atol=1e-11)
    assert np.allclose(o.project_site(g["L"], g["R1"], g["W1"], a), g["project_site_1"], atol=1e-11)
    assert np.allclose(o.update_left_environment(a, a, g["W1"], g["L"]), g["env_left"], atol=1e-11)
    assert np.allclose(o.update_right_environment(b, b, g["W2"], g["R"]), g["env_right"], atol=1e-11)
    assert np.allclose(o.project_bond(g["LB"], g["R"], g["C"]), g["project_bond"], atol=1e-11)
    mps, mpo = tensors(g, "mps"), tensors(g, "mpo")
    rb = o.right_environments(mps, mpo)
    for mine, ref in zip(rb, tensors(g, "renv")):
        assert np.allclose(mine, ref, atol=1e-12)
    th = o.merge_two_site(mps[0], mps[1])
    w2 = o.merge_mpo_tensors(mpo[0], mpo[1])
    l0 = np.ones((1, 1, 1), dtype=complex)
    for tol in (1e-4, 1e-12):
        got = o.update_site(l0, rb[1], w2, th, 0.05, tol)
        assert np.allclose(got, g[f"krylov_site2_tol{tol:g}"], atol=1e-12)
    for dist in ("left", "right", "sqrt"):
        l_, r_ = o.split_two_site(g["theta_split"], [2, 2], svd_distribution=dist, trunc_mode="discarded_weight", threshold=1e-3, max_bond_dim=4)
        assert l_.shape[2] == int(g[f"split_{dist}_keep"])
        assert np.allclose(o.merge_two_site(l_, r_), g[f"split_{dist}_recon"], atol=1e-12)


def test_one_tdvp_call_matches_reference():
    g = load("tdvp_step")
    for key in g["cases"]:
        key = str(key)
        L, chi, mode, sweeps = key.split("_")
        chi = int(chi[3:])
        st = o.MPSState(tensors(g, key + "_in"), 0)
        mpo = tensors(g, key + "_mpo")
        p = o.Params(dt=0.1, elapsed_time=0.1, max_bond_dim=chi, svd_threshold=1e-9, krylov_tol=1e-12, tdvp_sweeps=int(sweeps[1:]), tdvp_mode=mode)
        o.tdvp(st, mpo, p)
        assert [t.shape[2] for t in st.tensors] == list(g[key + "_bonds"]), key
        assert np.allclose(st.to_vec(), g[key + "_vec"], atol=1e-10), key
        assert abs(st.norm_sq() - float(g[key + "_norm"])) < 1e-12


def test_tdvp_against_dense_exponential():
    # tests/core/methods/tdvp/test_integrators.py:74-105: one exact 2TDVP step == expm(-i dt H) psi
    import scipy.linalg

    L = 5
    rng = np.random.default_rng(0)
    st = o.MPSState.haar(L, 4, rng)
    st.normalize("B")
    mpo = o.ising_mpo(L, 1.0, 0.5)
    psi0 = st.to_vec()
    p = o.Params(dt=0.05, elapsed_time=0.05, max_bond_dim=None, svd_threshold=1e-16, krylov_tol=1e-12)
    o.tdvp(st, mpo, p)
    H = o.mpo_to_matrix(mpo)
    assert np.allclose(H, H.conj().T)
    exact = scipy.linalg.expm(-1j * 0.05 * H) @ psi0
    assert np.allclose(st.to_vec(), exact, atol=1e-6)


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


class Scripted:
    def __init__(self, vals):
        self.vals = list(vals)

    def random(self):
        return self.vals.pop(0)

    def choice(self, n, p=None):
        u = self.vals.pop(0)
        cdf = np.cumsum(p)
        cdf /= cdf[-1]
        return int(np.searchsorted(cdf, u, side="right"))


def test_dissipation_and_jump_step():
    g = load("noise_step")
    sets = _noise_sets(6)
    for key in g["cases"]:
        key = str(key)
        nname, mode = key.split("_")
        st = o.MPSState(tensors(g, key + "_in"), 0)
        p = o.Params(dt=0.1, max_bond_dim=8, svd_threshold=1e-10)
        o.apply_dissipation(st, sets[nname], 0.1, p)
        assert np.allclose(phase_align(g[key + "_after_diss_vec"], st.to_vec()), g[key + "_after_diss_vec"], atol=1e-11), key
        dp = 1.0 - st.norm_sq(0)
        assert abs(dp - float(g[key + "_dp"])) < 1e-12, key
        if mode != "nojump":
            _, probs = o.jump_distribution(st.copy(), sets[nname], 0.1, p)
            assert np.allclose(probs, g[key + "_probs"], atol=1e-12), key
        st = o.stochastic_process(st, sets[nname], 0.1, p, Scripted(g[key + "_u"]))
        assert [t.shape[2] for t in st.tensors] == list(g[key + "_bonds"]), key
        ref = g[key + "_final_vec"]
        assert np.allclose(phase_align(ref, st.to_vec()), ref, atol=1e-10), key


def test_trajectories_match_reference_and_pinned_golden():
    g = load("trajectories")
    L = 5
    mpo = tensors(g, "mpo")
    noise = [o.make_process(n, [i], 0.1) for i in range(L) for n in ("lowering", "pauli_z")]
    init = o.MPSState.product(L, "zeros")
    for order in (1, 2):
        for sample in (False, True):
            p = o.Params(observables=[o.Obs(Z, s) for s in range(L)], elapsed_time=1, dt=0.1, max_bond_dim=4, svd_threshold=1e-6,
                         krylov_tol=1e-4, order=order, sample_timesteps=sample, random_seed=42)
            key = f"order{order}_sample{int(sample)}"
            res = []
            for i in range(10):
                o.JUMP_LOG = []
                r, dg, _ = o.run_trajectory(i, init, noise, p, mpo)
                dps = [e["dp"] for e in o.JUMP_LOG]
                ref_dp = g[key + "_dp"][i]
                ref_dp = ref_dp[~np.isnan(ref_dp)]
                assert len(dps) == len(ref_dp), (key, i)
                assert np.allclose(dps, ref_dp, atol=1e-9), (key, i)
                assert np.allclose(r, g[key + "_results"][i], atol=1e-9), (key, i)
                assert np.array_equal(dg, g[key + "_diag"][i]), (key, i)
                res.append(r)
            o.JUMP_LOG = None
            if order == 2 and not sample:
                mean = np.mean(res, axis=0).ravel()
                # the reference's own pinned golden, tests/test_simulator.py:191-197 (its tolerance is 2e-4)
                assert np.allclose(mean, g["pinned_expected_z"], atol=1e-9)


def test_closed_and_dephasing_configs():
    g = load("trajectories")
    mpo = tensors(g, "c1_mpo")
    init = o.MPSState.product(10, "zeros")
    for order in (1, 2):
        p = o.Params(observables=[o.Obs(Z, s) for s in range(10)], elapsed_time=1.0, dt=0.1, max_bond_dim=16, svd_threshold=1e-9,
                     krylov_tol=1e-12, order=order, sample_timesteps=True, random_seed=42)
        r, dg, _ = o.run_trajectory(0, init, None, p, mpo)
        assert np.allclose(r, g[f"c1_order{order}_results"], atol=1e-10)
        assert np.array_equal(dg, g[f"c1_order{order}_diag"])
    mpo = tensors(g, "c2_mpo")
    init = o.MPSState.product(8, "x+")
    noise = [o.make_process("pauli_z", [i], 0.1) for i in range(8)]
    p = o.Params(observables=[o.Obs(Z, s) for s in range(8)] + [o.Obs(X, s) for s in range(8)], elapsed_time=1.0, dt=0.1, max_bond_dim=8,
                 svd_threshold=1e-12, krylov_tol=1e-12, order=1, sample_timesteps=True, random_seed=42)
    sorted_rows = p.observable_sorted_indices
    for i in range(8):
        r, dg, _ = o.run_trajectory(i, init, noise, p, mpo)
        assert np.allclose(r, g["c2_results"][i], atol=1e-9), i
        assert np.array_equal(dg, g["c2_diag"][i]), i
    assert sorted_rows[0] == 0 and sorted_rows[8] == 1  # Z0 -> row 0, X0 -> row 1 (site-sorted, stable)


def test_digital_tebd_trajectories_match_reference():
    g = load("digital")
    assert np.allclose(o.rx_matrix(-0.1), g["rx_matrix"], atol=1e-15)
    assert np.allclose(o.rzz_tensor(-0.2), g["rzz_tensor"], atol=1e-15)
    L, steps = 8, 5
    obs = [o.Obs(Z, s) for s in range(L)] + [o.Obs(X, 3)]
    init = o.MPSState.product(L, "zeros")
    noise = [o.make_process(n, [i], 0.01) for i in range(L) for n in ("pauli_x", "pauli_y", "pauli_z")]
    p = o.DigitalParams(observables=obs, max_bond_dim=16, svd_threshold=1e-9, random_seed=3)
    layers = o.ising_trotter_layers(L, 1.0, 0.5, 0.1, steps)
    for i in range(6):
        r, dg, _ = o.digital_tjm(i, init, noise, p, layers)
        assert np.allclose(r, g["noisy_results"][i], atol=1e-9), i
        assert np.array_equal(dg, g["noisy_diag"][i]), i
    p = o.DigitalParams(observables=obs, max_bond_dim=16, svd_threshold=1e-9, random_seed=3, sample_layers=True, num_mid_measurements=steps)
    r, dg, _ = o.digital_tjm(0, init, None, p, o.ising_trotter_layers(L, 1.0, 0.5, 0.1, steps, sample_each=True))
    assert np.allclose(r, g["noiseless_results"][0], atol=1e-9)
    assert np.array_equal(dg, g["noiseless_diag"][0])
    noise2 = [o.make_process(n, [i], 0.1) for i in range(L) for n in ("lowering", "pauli_z")]
    p = o.DigitalParams(observables=obs, max_bond_dim=4, svd_threshold=1e-6, random_seed=7)
    for i in range(6):
        r, dg, _ = o.digital_tjm(i, init, noise2, p, o.ising_trotter_layers(L, 1.0, 0.5, 0.1, 3))
        assert np.allclose(r, g["strong_results"][i], atol=1e-9), i
        assert np.array_equal(dg, g["strong_diag"][i]), i


def _long_range_layers(g, L):
    cx, rzz = g["lr_cx_matrix"], g["lr_rzz_matrix"]
    layers = []
    for _ in range(2):
        singles = [(q, o.rx_matrix(0.3 + 0.1 * q)) for q in range(L)]
        layers.append(o.GateLayer(singles, [(1, 5, cx), (6, 2, rzz)], [(4, 3, cx), (7, 0, cx)], 0))
    return layers


def test_digital_long_range_gates_match_reference():
    """SWAP-routed long-range gates in both site orders (digital_tjm.py:476-499) with local one- and two-site noise."""
    g = load("digital")
    L = 8
    obs = [o.Obs(Z, s) for s in range(L)] + [o.Obs(X, 3)]
    init = o.MPSState.product(L, "zeros")
    noise = [o.make_process("pauli_x", [i], 0.05) for i in range(L)] + [o.make_process("crosstalk_zz", [1, 5], 0.1, factors=(Z, Z)),
                                                                         o.make_process("lowering", [6], 0.2)]
    p = o.DigitalParams(observables=obs, max_bond_dim=4, svd_threshold=1e-8, random_seed=11)
    r, dg, _ = o.digital_tjm(0, init, None, p, _long_range_layers(g, L))
    assert np.allclose(r, g["lr_noiseless_results"][0], atol=1e-9)
    assert np.array_equal(dg, g["lr_noiseless_diag"][0])
    for i in range(6):
        r, dg, _ = o.digital_tjm(i, init, noise, p, _long_range_layers(g, L))
        assert np.allclose(r, g["lr_noisy_results"][i], atol=1e-9), i
        assert np.array_equal(dg, g["lr_noisy_diag"][i]), i


def test_digital_long_range_gates_through_the_gate_mpo_match_reference():
    """gate_mode="mpo", the reference's default (digital_tjm.py:536-557): distant pairs through MPO.from_gate(...).multiply(state) and
    MPS.compress; tests/golden/digital_mpo.npz holds the reference's trajectories for a cap that bites (4) and one that does not (16)."""
    g, gd = load("digital_mpo"), load("digital")
    L = 8
    obs = [o.Obs(Z, s) for s in range(L)] + [o.Obs(X, 3)]
    init = o.MPSState.product(L, "zeros")
    noise = [o.make_process("pauli_x", [i], 0.05) for i in range(L)] + [o.make_process("crosstalk_zz", [1, 5], 0.1, factors=(Z, Z)),
                                                                         o.make_process("lowering", [6], 0.2)]
    for chi in (4, 16):
        p = o.DigitalParams(observables=obs, max_bond_dim=chi, svd_threshold=1e-8, random_seed=11, gate_mode="mpo")
        r, dg, _ = o.digital_tjm(0, init, None, p, _long_range_layers(gd, L))
        assert np.allclose(r, g[f"chi{chi}_noiseless_results"][0], atol=1e-9), chi
        assert np.array_equal(dg, g[f"chi{chi}_noiseless_diag"][0]), chi
        for i in range(6):
            r, dg, _ = o.digital_tjm(i, init, noise, p, _long_range_layers(gd, L))
            assert np.allclose(r, g[f"chi{chi}_noisy_results"][i], atol=1e-9), (chi, i)
            assert np.array_equal(dg, g[f"chi{chi}_noisy_diag"][i]), (chi, i)


def _generator_layers(g, L):
    def gate(name, s0, s1):
        return (s0, s1, g[name + "_matrix"], (g[name + "_gen0"], g[name + "_gen1"]))

    return [o.GateLayer([(q, _rx(0.3 + 0.1 * q)) for q in range(L)], [gate("rzz07", 1, 5), gate("rxx04", 6, 2)], [gate("ryy09", 4, 3), gate("rzz11", 7, 0)], 0)
            for _ in range(2)]


def _rx(theta):
    c, s_ = np.cos(theta / 2), np.sin(theta / 2)
    return np.array([[c, -1j * s_], [-1j * s_, c]], dtype=np.complex128)


def test_digital_gates_by_tdvp_on_a_window_are_rounding_defined():
    """gate_mode="tdvp" / "full-tdvp" (digital_tjm.py:408-453, 592-614): gates with a product-form generator as two-site TDVP over
    unit time on the window around their support.  On the product-like states a circuit starts from, the splits inside the window keep
    min_keep = 2 singular values of which one is ZERO, its singular vectors are whatever rounding leaves, and the projector of the
    following TDVP steps is built on them: a relative perturbation of 1e-15 of the input moves the output by 1e-4 ... 1e-3 (shown
    below on the restated algorithm), and the reference and this restatement - the same operations on the same inputs - differ from
    each other by as much.  There is no number to reproduce to 1e-8, so the HIP engine does not build this route (it refuses the
    modes loudly); the restatement is kept and held to the reference's fixture (tests/golden/digital_tdvp.npz) within that
    conditioning."""
    g = load("digital_tdvp")
    L = 8
    obs = [o.Obs(Z, s) for s in range(L)] + [o.Obs(X, 3)]
    init = o.MPSState.product(L, "zeros")
    for mode in ("tdvp", "full-tdvp"):
        for chi in (4, None):
            p = o.DigitalParams(observables=obs, max_bond_dim=chi, svd_threshold=1e-8, krylov_tol=1e-10, random_seed=11, gate_mode=mode)
            r, dg, _ = o.digital_tjm(0, init, None, p, _generator_layers(g, L))
            assert np.allclose(r, g[f"{mode}_chi{chi}_noiseless_results"][0], atol=5e-3), (mode, chi)
    # the conditioning: one gate on a product state, input perturbed at the rounding level
    rng = np.random.default_rng(0)
    st = o.MPSState.product(L, "zeros")
    for q in range(L):
        o.apply_single_qubit_gate(st, q, _rx(0.3 + 0.1 * q))
    gen = (g["rzz07_gen0"], g["rzz07_gen1"])
    p = o.DigitalParams(max_bond_dim=4, svd_threshold=1e-8, krylov_tol=1e-10, gate_mode="tdvp")
    base = o.MPSState([t.copy() for t in st.tensors], None)
    o.apply_two_qubit_gate_tdvp(base, 1, 5, gen, p)
    moved = []
    for _ in range(3):
        pert = o.MPSState([t * (1 + 1e-15 * rng.standard_normal(t.shape)) for t in st.tensors], None)
        o.apply_two_qubit_gate_tdvp(pert, 1, 5, gen, p)
        va, vb = base.to_vec(), pert.to_vec()
        moved.append(np.linalg.norm(va - vb * np.exp(1j * np.angle(np.vdot(vb, va)))))
    assert max(moved) > 1e-6, moved  # ten orders of magnitude above the perturbation


def test_measure_single_shot_matches_reference():
    """Projective sampling of all sites (mps.py:1282-1350) with the recorded draws, in the Z, X and Y bases."""
    g = load("shots")
    st = o.MPSState([g[f"t{i}"] for i in range(6)], 0)
    for bi, basis in enumerate("ZXY"):
        for k in range(40):
            assert o.measure_single_shot(st, g["uniforms"][bi, k], basis) == g["codes"][bi, k], (basis, k)


def test_entropy_schmidt_and_bitstring_projection_match_reference():
    """MPS.get_entropy / get_schmidt_spectrum / project_onto_bitstring (mps.py:604-678, 1495-1537)."""
    g = load("shots")
    st = o.MPSState([g[f"t{i}"] for i in range(6)], 0)
    for i in range(5):
        assert abs(o.get_entropy(st, [i, i + 1]) - g["entropy"][i]) < 1e-12
        assert np.allclose(o.get_schmidt_spectrum(st, [i, i + 1]), g["schmidt"][i], atol=1e-13, equal_nan=True)
    for b, ref in zip(g["pvm_strings"], g["pvm"]):
        assert abs(o.project_onto_bitstring(st, str(b)) - ref) < 1e-14


def _scheduled_setup(g):
    L = 6
    mpo = [g[f"mpo{i}"] for i in range(L)]
    sched = [{"time": 0.0, "sites": [2], "matrix": X}, {"time": 0.2, "sites": [4], "matrix": o.JUMP_OPS["lowering"]},
             {"time": 0.3, "sites": [1, 2], "matrix": g["two"]}]
    noise = [o.make_process("pauli_z", [i], 0.2) for i in range(L)]
    p = o.Params(observables=[o.Obs(Z, s) for s in range(L)] + [o.Obs(X, 0)], elapsed_time=0.5, dt=0.1, max_bond_dim=8, svd_threshold=1e-10,
                 krylov_tol=1e-10, order=1, sample_timesteps=True, random_seed=21)
    return L, mpo, sched, noise, p


def test_scheduled_jumps_match_reference():
    """analog_tjm_1 with deterministic jumps replacing the stochastic step at their times (scheduled_jumps.py:51-119)."""
    g = load("scheduled")
    L, mpo, sched, noise, p = _scheduled_setup(g)
    for t in range(4):
        r, dg, _ = o.analog_tjm_1(t, o.MPSState.product(L, "x+"), noise, p, mpo, scheduled=sched)
        assert np.allclose(r, g["results"][t], atol=1e-9), t
        assert np.array_equal(dg, g["diag"][t]), t


def _piecewise_setup(g):
    L, n = 6, 4
    hams = tuple([g[f"h{k}_mpo{i}"] for i in range(L)] for k in range(n))
    noise = [o.make_process("pauli_z", [i], 0.1) for i in range(L)]
    obs = [o.Obs(Z, s) for s in range(L)] + [o.Obs(X, 2)]
    return L, n, hams, noise, obs


def test_piecewise_hamiltonian_matches_reference():
    """One MPO per time interval (analog_tjm.py:43-49): interval j-1 for step j (order 1); j-2 for the step and j-1 for the
    sample of order 2 (analog_tjm.py:351-360)."""
    g = load("piecewise")
    L, n, hams, noise, obs = _piecewise_setup(g)
    for order in (1, 2):
        p = o.Params(observables=obs, elapsed_time=0.1 * n, dt=0.1, max_bond_dim=8, svd_threshold=1e-10, krylov_tol=1e-10, order=order,
                     sample_timesteps=True, random_seed=5)
        for t in range(3):
            r, dg, _ = o.run_trajectory(t, o.MPSState.product(L, "x+"), noise, p, hams)
            assert np.allclose(r, g[f"order{order}_results"][t], atol=1e-9), (order, t)
            assert np.array_equal(dg, g[f"order{order}_diag"][t]), (order, t)


def test_one_and_two_site_chains_match_reference():
    """L = 1 takes the 1TDVP fallback (tdvp.py:96-98); L = 2 has a single bond."""
    g = load("tiny")
    for L in (1, 2):
        mpo = [g[f"L{L}_mpo{i}"] for i in range(L)]
        noise = [o.make_process("lowering", [i], 0.3) for i in range(L)]
        for order in (1, 2):
            p = o.Params(observables=[o.Obs(Z, L - 1), o.Obs(X, 0)], elapsed_time=0.3, dt=0.1, max_bond_dim=4, svd_threshold=1e-10,
                         krylov_tol=1e-10, order=order, sample_timesteps=True, random_seed=13)
            for t in range(5):
                r, _, _ = o.run_trajectory(t, o.MPSState.product(L, "x+"), noise, p, mpo)
                assert np.allclose(r, g[f"L{L}_order{order}"][t], atol=1e-9), (L, order, t)


@pytest.mark.parametrize("case", ["test_two_site_correlator_left_boundary", "test_two_site_correlator_center", "test_two_site_correlator_right_boundary"])
def test_oracle_reproduces_the_reference_two_site_correlator_series(case):
    """The pinned <XX>, <YY>, <ZZ> series of the reference's own tests (tests/test_simulator.py:858-1188; extracted as data by
    tools/extract_reference_test_vectors.py): closed 4-site Ising chain, default preset, 21 time points, the reference's tolerance."""
    import json

    g = json.load(open(os.path.join(GOLDEN, "reference_two_site_correlators.json")))[case]
    x, y, z = o.PAULI["x"], o.PAULI["y"], o.PAULI["z"]
    obs = [o.Obs(np.kron(m, m), list(g["sites"])) for m in (x, y, z)]
    p = o.Params(observables=obs, elapsed_time=g["elapsed_time"], dt=g["dt"], max_bond_dim=g["max_bond_dim"], svd_threshold=1e-6, krylov_tol=1e-4,
                 order=1, sample_timesteps=True)
    res, _, _ = o.run_trajectory(0, o.MPSState.product(g["L"], "zeros"), [], p, o.ising_mpo(g["L"], g["J"], g["g"]))
    idx = p.observable_sorted_indices
    for k, name in enumerate(("xx", "yy", "zz")):
        assert np.allclose(res[idx[k]], np.array(g[name]), atol=1e-3), name


def test_dynamic_tdvp_and_bug_match_reference():
    """tdvp(tdvp_mode="dynamic") (integrators.py:294-511) and bug() (bug.py:213-257): one call on small chains whose bonds sit below,
    at and above the cap, and whole noisy trajectories of both drivers in both modes (fixtures written by the reference)."""
    g = load("f3_dynamic_bug")
    for key in g["cases"]:
        key = str(key)
        cap = key.split("_")[2][3:]
        cap = None if cap == "None" else int(cap)
        mpo = tensors(g, key + "_mpo")
        for mode in ("dynamic", "bug"):
            if mode == "bug" and key.endswith("x+"):
                continue  # product state: BUG's stacked bases are exactly rank deficient, the reference's own result is rounding-dependent
            st = o.MPSState(tensors(g, key + "_in"), 0)
            p = o.Params(dt=0.1, elapsed_time=0.1, max_bond_dim=cap, svd_threshold=1e-9, krylov_tol=1e-12,
                         tdvp_mode="dynamic" if mode == "dynamic" else "2site", evolution_mode="bug" if mode == "bug" else "tdvp")
            o.apply_unitary_evolution(st, mpo, p)
            assert [t.shape[2] for t in st.tensors] == list(g[f"{key}_{mode}_bonds"]), (key, mode)
            v, ref = st.to_vec(), g[f"{key}_{mode}_vec"]
            ov = np.vdot(ref, v)
            assert np.allclose(v, ref * (ov / abs(ov)), atol=1e-9), (key, mode, np.abs(v - ref * (ov / abs(ov))).max())
            assert abs(st.norm_sq() - float(g[f"{key}_{mode}_norm"])) < 1e-10, (key, mode)
    L = 6
    mpo = tensors(g, "traj_mpo")
    noise = [o.make_process(n, [i], 0.1) for i in range(L) for n in ("lowering", "pauli_z")]
    for mode in ("dynamic", "bug"):
        for order in (1, 2):
            p = o.Params(observables=[o.Obs(Z, s) for s in range(L)], elapsed_time=0.5, dt=0.1, max_bond_dim=4, svd_threshold=1e-9, krylov_tol=1e-12,
                         order=order, sample_timesteps=True, random_seed=9, tdvp_mode="dynamic" if mode == "dynamic" else "2site",
                         evolution_mode="bug" if mode == "bug" else "tdvp")
            for t in range(4):
                r, dg, _ = o.run_trajectory(t, o.MPSState(tensors(g, "traj_in"), 0), noise, p, mpo)
                assert np.allclose(r, g[f"traj_{mode}_order{order}_results"][t], atol=1e-8), (mode, order, t, np.abs(r - g[f"traj_{mode}_order{order}_results"][t]).max())
                assert np.array_equal(dg, g[f"traj_{mode}_order{order}_diag"][t]), (mode, order, t)


def test_the_reference_dynamic_sweep_depends_on_the_gauge_of_its_input():
    """Why the HIP engine does not follow sweep_dynamic's leftward one-site branch to the letter (integrators.py:450-461: left_qr hands
    back R^T and the sweep transposes it once more, so R - with its indices the wrong way round - is evolved and absorbed).  The same
    physical state in two gauges - a random unitary inserted on one bond, both copies right-canonical with the centre on site 0 -
    gives two different states after one sweep of the restated reference (switch on), and one and the same state with the switch
    off, which is what the engine computes and is compared with.  A result that changes with the gauge depends on the signs and
    phases LAPACK's SVD and QR picked earlier in the run: no independent implementation can reproduce it."""
    g = load("f3_dynamic_bug")
    key = "L5_c4_cap4_haar"  # every bond at the cap: the one-site branch is taken on the way back
    mpo = tensors(g, key + "_mpo")
    base = [t.copy() for t in tensors(g, key + "_in")]
    rng = np.random.default_rng(2)
    chi = base[1].shape[2]
    u, _ = np.linalg.qr(rng.standard_normal((chi, chi)) + 1j * rng.standard_normal((chi, chi)))
    gauged = [t.copy() for t in base]
    gauged[1] = np.einsum("pab,bc->pac", base[1], u)
    gauged[2] = np.einsum("cb,pbd->pcd", u.conj().T, base[2])
    assert np.allclose(o.MPSState(base, 0).to_vec(), o.MPSState(gauged, 0).to_vec(), atol=1e-13)
    moved = {}
    for as_reference in (True, False):
        out = []
        for t in (base, gauged):
            st = o.MPSState([x.copy() for x in t], 0)
            o.tdvp(st, mpo, o.Params(dt=0.1, max_bond_dim=4, svd_threshold=1e-9, krylov_tol=1e-12, tdvp_mode="dynamic",
                                     reference_dynamic_transpose=as_reference))
            out.append(st.to_vec())
        ov = np.vdot(out[0], out[1])
        moved[as_reference] = float(np.linalg.norm(out[1] - out[0] * (ov / abs(ov))))
    assert moved[False] < 1e-10, moved
    assert moved[True] > 1e-3, moved


def test_the_reference_bug_step_from_a_product_state_is_rounding_defined():
    """Why the BUG fixtures start from generic states: from a product state the stacked trial basis [retained | predictor] of
    build_trial_basis (bug.py:65-90) has exactly dependent columns, its QR completes the basis with whatever rounding leaves, and the
    result moves by 1e-2 when the input is perturbed by 1e-15 (shown on the restatement; the reference's own vector for this case lies
    as far from the restatement's as the restatement's lies from itself)."""
    g = load("f3_dynamic_bug")
    key = "L6_c1_cap4_x+"
    mpo, base = tensors(g, key + "_mpo"), tensors(g, key + "_in")
    rng = np.random.default_rng(0)

    def step(ts):
        st = o.MPSState([x.copy() for x in ts], 0)
        o.bug(st, mpo, o.Params(dt=0.1, max_bond_dim=4, svd_threshold=1e-9, krylov_tol=1e-12))
        return st.to_vec()

    def dist(a, b):
        ov = np.vdot(a, b)
        return float(np.linalg.norm(b - a * (ov / abs(ov))))

    v0 = step(base)
    moved = [dist(v0, step([t * (1 + 1e-15 * rng.standard_normal(t.shape)) for t in base])) for _ in range(3)]
    assert max(moved) > 1e-4, moved
    assert dist(g[key + "_bug_vec"], v0) < 10 * max(moved) + 1e-2


def _mixed_case(g):
    dims = [int(x) for x in g["dims"]]
    L = len(dims)
    lower = {d_: np.diag(np.sqrt(np.arange(1, d_)), 1).astype(complex) for d_ in set(dims)}
    number = {d_: lower[d_].conj().T @ lower[d_] for d_ in lower}
    init = []
    for i, ch in enumerate(str(g["basis"])):
        v = np.zeros(dims[i], dtype=complex)
        v[int(ch)] = 1.0
        init.append(v.reshape(dims[i], 1, 1))
    return dims, L, lower, number, init


def test_mixed_local_dimensions_match_reference():
    """A chain whose sites differ in dimension (tests/golden/mixed_dims.npz: the reference on MPO.coupled_transmon, three-level transmons
    and two-level resonators alternating, loss on every site with its own ladder operator): one closed two-site TDVP step from a random
    state and noisy trajectories of both drivers, bond diagnostics included."""
    g = load("mixed_dims")
    dims, L, lower, number, init = _mixed_case(g)
    mpo = tensors(g, "mpo")
    st = o.MPSState([t.copy() for t in tensors(g, "in")], 0)
    o.tdvp(st, mpo, o.Params(dt=0.05, svd_threshold=1e-10, max_bond_dim=8, krylov_tol=1e-12))
    assert [t.shape[2] for t in st.tensors] == list(g["tdvp_bonds"])
    ref = g["tdvp_vec"]
    assert abs(abs(np.vdot(ref, st.to_vec())) - np.vdot(ref, ref).real) < 1e-10
    on = [o.make_process("loss", [i], 0.25, matrix=lower[dims[i]]) for i in range(L)]
    for order in (1, 2):
        op = o.Params(observables=[o.Obs(number[dims[s]], s) for s in range(L)], elapsed_time=0.4, dt=0.1, max_bond_dim=8, svd_threshold=1e-10,
                      krylov_tol=1e-12, order=order, sample_timesteps=True, random_seed=6)
        for t in range(3):
            r, dg, _ = o.run_trajectory(t, o.MPSState([x.copy() for x in init], 0), on, op, mpo)
            assert np.allclose(r, g[f"order{order}_results"][t], atol=1e-9), (order, t)
            assert np.array_equal(dg, g[f"order{order}_diag"][t]), (order, t)
    # scheduled jumps with operators of the sites' own dimensions
    sched = [{"time": 0.1, "sites": [2], "matrix": lower[3]},
             {"time": 0.2, "sites": [2, 3], "matrix": np.kron(number[3] + 0.5 * lower[3], lower[2].conj().T + np.eye(2))}]
    on = [o.make_process("loss", [i], 0.1, matrix=lower[dims[i]]) for i in range(L)]
    op = o.Params(observables=[o.Obs(number[dims[s]], s) for s in range(L)], elapsed_time=0.4, dt=0.1, max_bond_dim=8, svd_threshold=1e-10, krylov_tol=1e-12,
                  order=1, sample_timesteps=True, random_seed=8)
    for t in range(3):
        r, _, _ = o.analog_tjm_1(t, o.MPSState([x.copy() for x in init], 0), on, op, mpo, scheduled=sched)
        assert np.allclose(r, g["scheduled_results"][t], atol=1e-9), t


def test_fermi_hubbard_step_matches_reference():
    """Composite four-level sites, MPO of bond dimension 6 (tests/golden/fermi_hubbard.npz): one closed two-site TDVP step."""
    g = load("fermi_hubbard")
    st = o.MPSState([t.copy() for t in tensors(g, "in")], 0)
    o.tdvp(st, tensors(g, "mpo"), o.Params(dt=0.05, svd_threshold=1e-10, max_bond_dim=8, krylov_tol=1e-12))
    assert [t.shape[2] for t in st.tensors] == list(g["tdvp_bonds"])
    ref = g["tdvp_vec"]
    assert abs(abs(np.vdot(ref, st.to_vec())) - np.vdot(ref, ref).real) < 1e-10


def _continuation_setup(g):
    L = 5
    mpo = tensors(g, "mpo")
    noise = [o.make_process(n, [i], 0.15) for i in range(L) for n in ("lowering", "pauli_z")]
    kw = dict(dt=0.1, max_bond_dim=4, svd_threshold=1e-9, krylov_tol=1e-12, random_seed=31)
    return L, mpo, noise, kw


def test_sample_at_and_segment_stitching_match_reference():
    """The continuation options of analog_tjm_1 / analog_tjm_2 (analog_tjm.py:206-462): ``sample_at`` with and without
    ``sample_timesteps``, and an order-2 run cut after 3 of 6 steps - shared trajectory stream, phi handed over, sample streams on
    the global timeline - which the reference stitches to the continuous run bit for bit."""
    g = load("continuation")
    L, mpo, noise, kw = _continuation_setup(g)
    obs = [o.Obs(Z, s) for s in range(L)]
    init = o.MPSState.product(L, "x+")
    for order, fn in ((1, o.analog_tjm_1), (2, o.analog_tjm_2)):
        p = o.Params(observables=obs, elapsed_time=0.6, sample_timesteps=True, order=order, **kw)
        p1 = o.Params(observables=obs, elapsed_time=0.6, sample_timesteps=False, order=order, **kw)
        for t in range(4):
            assert np.allclose(fn(t, init, noise, p, mpo, sample_at=[0, 2, 5])[0], g[f"sample_at_order{order}"][t], atol=1e-9), (order, t)
            assert np.allclose(fn(t, init, noise, p1, mpo, sample_at=[3])[0], g[f"sample_at_single_order{order}"][t], atol=1e-9), (order, t)
        with pytest.raises(ValueError, match="outside the time grid"):
            fn(0, init, noise, p, mpo, sample_at=[7])
        with pytest.raises(ValueError, match="requires sample_timesteps=True"):
            fn(0, init, noise, p1, mpo, sample_at=[1, 2])
    seg = o.Params(observables=obs, elapsed_time=0.3, sample_timesteps=True, order=2, **kw)
    for t in range(4):
        rng = o.trajectory_rng(31, t)
        r1, _, phi = o.analog_tjm_2(t, init, noise, seg, mpo, rng=rng, return_trajectory_state=True)
        r2, _, phi2 = o.analog_tjm_2(t, phi, noise, seg, mpo, rng=rng, sample_timestep_offset=3, continue_trajectory=True, return_trajectory_state=True)
        assert np.allclose(r1, g["segment1"][t], atol=1e-9) and np.allclose(r2, g["segment2"][t], atol=1e-9), t
        assert np.allclose(r1, g["whole"][t][:, :4], atol=1e-9) and np.allclose(r2, g["whole"][t][:, 3:], atol=1e-9), t
        assert [x.shape[2] for x in phi2.tensors] == list(g["phi_bonds"][t])


def _qudit_case(g, key):
    d, L = int(key[1]), int(key.split("_L")[1])
    b = np.diag(np.sqrt(np.arange(1, d)), 1).astype(complex)
    n = b.conj().T @ b
    mpo = tensors(g, key + "_mpo")
    procs = [("loss", [i], 0.3, b) for i in range(L)] + [("dephasing", [i], 0.1, n) for i in range(L)]
    init = []
    for i in range(L):
        v = np.zeros(d, dtype=complex)
        v[(i + 1) % d] = 1.0
        init.append(v.reshape(d, 1, 1))
    return d, L, n, mpo, procs, init


def test_qutrit_and_four_level_chains_match_reference():
    """Local dimension 3 and 4 (tests/golden/qudit.npz: the reference on Bose-Hubbard chains with loss and dephasing): the oracle is
    dimension-generic as the reference is - one two-site TDVP step from a random state and noisy trajectories of both drivers."""
    g = load("qudit")
    for key in g["cases"]:
        key = str(key)
        d, L, n, mpo, procs, init = _qudit_case(g, key)
        chi = 9 if d == 3 else 8
        st = o.MPSState([t.copy() for t in tensors(g, key + "_in")], 0)
        o.tdvp(st, mpo, o.Params(dt=0.05, svd_threshold=1e-10, max_bond_dim=chi, krylov_tol=1e-12))
        assert [t.shape[2] for t in st.tensors] == list(g[key + "_tdvp_bonds"])
        ref = g[key + "_tdvp_vec"]
        assert abs(abs(np.vdot(ref, st.to_vec())) - np.vdot(ref, ref).real) < 1e-10
        on = [o.make_process(nm, s, gam, matrix=m) for nm, s, gam, m in procs]
        for order in (1, 2):
            op = o.Params(observables=[o.Obs(n, s) for s in range(L)], elapsed_time=0.4, dt=0.1, max_bond_dim=chi, svd_threshold=1e-10, krylov_tol=1e-12,
                          order=order, sample_timesteps=True, random_seed=4)
            want = g[f"{key}_order{order}_results"]
            for t in range(3):
                r, _, _ = o.run_trajectory(t, o.MPSState([x.copy() for x in init], 0), on, op, mpo)
                assert np.allclose(r, want[t], atol=1e-9), (key, order, t)
