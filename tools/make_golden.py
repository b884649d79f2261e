"""Generate golden fixtures under tests/golden/ by importing the REFERENCE itself.

Runs only in the build container (needs /root/reference); provenance script for the
committed ``tests/golden/*.npz`` data.  Fixtures hold inputs and reference outputs
only - never reference source text.

    python tools/make_golden.py
"""
from __future__ import annotations

import copy
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(__file__))
from ref_boot import ref  # noqa: E402

OUT = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)

MPS = ref("core.data_structures.mps").MPS
MPO = ref("core.data_structures.mpo").MPO
NoiseModel = ref("core.data_structures.noise_model").NoiseModel
sp = ref("core.data_structures.simulation_parameters")
gl = ref("core.libraries.gate_library")
tjm = ref("analog.analog_tjm")
tdvp_mod = ref("core.methods.tdvp.tdvp")
diss = ref("core.methods.dissipation")
stoch = ref("core.methods.stochastic_process")
rutil = ref("core.random_utils")
linalg = ref("core.linalg")
decomp = ref("core.methods.decompositions")
prim = ref("core.methods.tdvp.primitives")
mexp = ref("core.methods.matrix_exponential")


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path), "bytes")


def pack_tensors(prefix, tensors):
    return {f"{prefix}{i}": np.asarray(t, dtype=np.complex128) for i, t in enumerate(tensors)}


def haar_mps(length, chi, seed):
    """Seeded Haar MPS built with the same recipe as mps.py:170-221, then normalize('B')."""
    rng = np.random.default_rng(seed)
    caps = [1] * (length + 1)
    left = 1
    for i in range(1, length):
        left *= 2
        caps[i] = left
    right = 1
    for i in range(length - 1, 0, -1):
        right *= 2
        caps[i] = min(caps[i], right, chi)
    tensors = []
    for i in range(length):
        cl, cr = caps[i], caps[i + 1]
        x = rng.standard_normal((2 * cl, cr)) + 1j * rng.standard_normal((2 * cl, cr))
        q, r = np.linalg.qr(x, mode="reduced")
        d = np.diag(r)
        q = q / (d / np.abs(d))[np.newaxis, :]
        tensors.append(q.reshape(2, cl, cr).astype(np.complex128))
    m = MPS(length, tensors=tensors)
    m.normalize("B")
    return m


# ------------------------------------------------------------------ 1. RNG streams
def gen_rng():
    seeds = [0, 1, 42, 12345]
    trajs = [0, 1, 2, 7, 1023]
    steps = [0, 1, 2, 10]
    traj_tab = np.zeros((len(seeds), len(trajs), 8))
    samp_tab = np.zeros((len(seeds), len(trajs), len(steps), 8))
    for a, s in enumerate(seeds):
        for b, t in enumerate(trajs):
            traj_tab[a, b] = rutil.make_trajectory_rng(t, base_seed=s).random(8)
            for c, k in enumerate(steps):
                samp_tab[a, b, c] = rutil.make_sample_rng(t, base_seed=s, timestep=k).random(8)
    # rng.choice equivalence table: (p vectors, drawn index, next double)
    rng = np.random.default_rng(99)
    ps, idxs, nxt = [], [], []
    for case in range(64):
        n = int(rng.integers(2, 9))
        p = rng.random(n)
        p /= p.sum()
        g = rutil.make_trajectory_rng(case, base_seed=7)
        g.random()
        idxs.append(int(g.choice(n, p=p)))
        nxt.append(g.random())
        pp = np.zeros(8)
        pp[:n] = p
        ps.append(pp)
    save("rng_streams", seeds=np.array(seeds), trajs=np.array(trajs), steps=np.array(steps), traj=traj_tab, sample=samp_tab,
         choice_p=np.array(ps), choice_idx=np.array(idxs), choice_next=np.array(nxt))


# ------------------------------------------------------------------ 2. truncation KATs
def gen_truncate():
    rng = np.random.default_rng(5)
    rows = []
    specs = []
    modes = ["discarded_weight", "relative", "hard_cutoff", "relative_discarded_weight"]
    for case in range(200):
        n = int(rng.integers(1, 12))
        s = np.sort(np.abs(rng.standard_normal(n)) * 10.0 ** rng.integers(-8, 1, size=n))[::-1].copy()
        if case % 17 == 0:
            s[n // 2:] = 0.0
        mode = modes[case % 4]
        thr = float(10.0 ** rng.integers(-14, 1))
        cap = [None, 1, 2, 4, 8][case % 5]
        min_keep = [1, 2][case % 2]
        if cap is not None and cap < min_keep:
            cap = min_keep
        keep = linalg.truncate(s, mode=mode, threshold=thr, max_bond_dim=cap, min_keep=min_keep)
        pad = np.full(12, -1.0)
        pad[:n] = s
        rows.append(pad)
        specs.append([n, modes.index(mode), thr, -1 if cap is None else cap, min_keep, keep])
    save("truncate_kat", spectra=np.array(rows), specs=np.array(specs, dtype=np.float64))


# ------------------------------------------------------------------ 3. local kernels
def gen_kernels():
    rng = np.random.default_rng(11)

    def crand(*shape):
        return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)

    out = {}
    d, D = 2, 3
    cl, cm, cr = 5, 6, 4
    a = crand(d, cl, cm)
    b = crand(d, cm, cr)
    out["A"], out["B"] = a, b
    out["merge"] = decomp.merge_two_site(a, b)
    w1, w2 = crand(d, d, D, D), crand(d, d, D, D)
    out["W1"], out["W2"] = w1, w2
    out["merge_mpo"] = prim.merge_mpo_tensors(w1, w2)
    lenv = crand(cl, D, cl)
    renv = crand(cr, D, cr)
    out["L"], out["R"] = lenv, renv
    out["project_site_2"] = prim.project_site(lenv, renv, out["merge_mpo"], out["merge"])
    renv1 = crand(cm, D, cm)
    out["R1"] = renv1
    out["project_site_1"] = prim.project_site(lenv, renv1, w1, a)
    out["env_left"] = prim.update_left_environment(a, a, w1, lenv)
    out["env_right"] = prim.update_right_environment(b, b, w2, renv)
    c = crand(cl, cr)
    lb = crand(cl, D, cl)
    out["C"], out["LB"] = c, lb
    out["project_bond"] = prim.project_bond(lb, renv, c)
    # hermitian H_eff from real environments: build from an actual MPS / MPO
    L = 6
    m = haar_mps(L, 8, 3)
    H = MPO.ising(L, 1.0, 0.5)
    rb = prim.initialize_right_environments(m, H)
    out.update(pack_tensors("mps", m.tensors))
    out.update(pack_tensors("mpo", H.tensors))
    out.update(pack_tensors("renv", rb))
    th = decomp.merge_two_site(m.tensors[0], m.tensors[1])
    w = prim.merge_mpo_tensors(H.tensors[0], H.tensors[1])
    l0 = np.zeros((1, 1, 1), dtype=complex)
    l0[0, 0, 0] = 1
    for tol in (1e-4, 1e-12):
        out[f"krylov_site2_tol{tol:g}"] = prim.update_site(l0, rb[1], w, th, 0.05, krylov_tol=tol)
    # split with each distribution
    th2 = crand(4, 3, 5)
    out["theta_split"] = th2
    for dist in ("left", "right", "sqrt"):
        l_, r_ = decomp.split_two_site(th2, [2, 2], svd_distribution=dist, trunc_mode="discarded_weight", threshold=1e-3, max_bond_dim=4)
        out[f"split_{dist}_recon"] = decomp.merge_two_site(l_, r_)
        out[f"split_{dist}_keep"] = np.array(l_.shape[2])
    save("kernels", **out)


# ------------------------------------------------------------------ 4. one tdvp() call
def gen_tdvp():
    out = {}
    cases = []
    for (L, chi, mode, sweeps, seed) in [(2, 2, "2site", 1, 1), (4, 4, "2site", 1, 2), (5, 4, "2site", 2, 3), (10, 16, "2site", 1, 4),
                                          (4, 4, "1site", 1, 5), (6, 8, "1site", 1, 6), (10, 16, "1site", 2, 7), (6, 2, "2site", 1, 8)]:
        m = haar_mps(L, chi, seed)
        H = MPO.ising(L, 1.0, 0.5)
        params = sp.AnalogSimParams(observables=[sp.Observable(gl.Z(), 0)], elapsed_time=0.1, dt=0.1, max_bond_dim=chi,
                                    svd_threshold=1e-9, krylov_tol=1e-12, tdvp_sweeps=sweeps, tdvp_mode=mode, sample_timesteps=False)
        key = f"L{L}_chi{chi}_{mode}_s{sweeps}"
        cases.append(key)
        out.update(pack_tensors(key + "_in", m.tensors))
        out.update(pack_tensors(key + "_mpo", H.tensors))
        tdvp_mod.tdvp(m, H, params)
        out[key + "_vec"] = m.to_vec()
        out[key + "_bonds"] = np.array([t.shape[2] for t in m.tensors])
        out[key + "_norm"] = np.array(m.norm())
    out["cases"] = np.array(cases)
    save("tdvp_step", **out)


# ------------------------------------------------------------------ 5. dissipation + jumps
class ScriptedRng:
    def __init__(self, values):
        self.values = list(values)

    def random(self):
        return self.values.pop(0)

    def choice(self, n, p=None):
        u = self.values.pop(0)
        cdf = np.cumsum(p)
        cdf /= cdf[-1]
        return int(np.searchsorted(cdf, u, side="right"))


def gen_noise():
    out = {}
    L = 6
    noise_sets = {
        "pauli": [{"name": n, "sites": [i], "strength": 0.1 + 0.01 * i} for i in range(L) for n in ("pauli_z", "pauli_x")],
        "lowering": [{"name": "lowering", "sites": [i], "strength": 0.2} for i in range(L)],
        "mixed": [{"name": n, "sites": [i], "strength": 0.1} for i in range(L) for n in ("lowering", "pauli_z")],
        "twosite": [{"name": "pauli_z", "sites": [i], "strength": 0.05} for i in range(L)]
        + [{"name": "crosstalk_xx", "sites": [i, i + 1], "strength": 0.07} for i in range(L - 1)]
        + [{"name": "longrange_crosstalk_zz", "sites": [0, 3], "strength": 0.03}],
    }
    names = []
    for nname, procs in noise_sets.items():
        nm = NoiseModel(procs)
        for mode, uvals in (("nojump", [0.999999]), ("jump", [0.0, 0.37]), ("jump2", [0.0, 0.93])):
            m = haar_mps(L, 8, 21)
            params = sp.AnalogSimParams(observables=[sp.Observable(gl.Z(), 0)], elapsed_time=0.1, dt=0.1, max_bond_dim=8,
                                        svd_threshold=1e-10, sample_timesteps=False)
            key = f"{nname}_{mode}"
            names.append(key)
            out.update(pack_tensors(key + "_in", m.tensors))
            diss.apply_dissipation(m, nm, 0.1, params)
            out[key + "_after_diss_vec"] = m.to_vec()
            out[key + "_dp"] = np.array(float(stoch.calculate_stochastic_factor(m)))
            if mode != "nojump":
                mm = copy.deepcopy(m)
                _, probs = stoch.create_probability_distribution(mm, nm, 0.1, params)
                out[key + "_probs"] = np.array(probs)
            m = stoch.stochastic_process(m, nm, 0.1, params, rng=ScriptedRng(uvals))
            out[key + "_final_vec"] = m.to_vec()
            out[key + "_bonds"] = np.array([t.shape[2] for t in m.tensors])
            out[key + "_u"] = np.array(uvals)
    out["cases"] = np.array(names)
    save("noise_step", **out)


# ------------------------------------------------------------------ 6. full trajectories
def gen_traj():
    out = {}
    L = 5
    H = MPO.ising(L, 1, 0.5)
    out.update(pack_tensors("mpo", H.tensors))
    noise = NoiseModel([{"name": n, "sites": [i], "strength": 0.1} for i in range(L) for n in ("lowering", "pauli_z")])
    st = MPS(L, state="zeros")
    st.normalize("B")
    dps = []
    orig = stoch.calculate_stochastic_factor

    def spy(state):
        v = orig(state)
        dps.append(float(v))
        return v

    stoch.calculate_stochastic_factor = spy
    try:
        for order in (1, 2):
            for sample in (False, True):
                params = sp.AnalogSimParams(observables=[sp.Observable(gl.Z(), s) for s in range(L)], elapsed_time=1, dt=0.1, num_traj=10,
                                            max_bond_dim=4, svd_threshold=1e-6, order=order, sample_timesteps=sample, random_seed=42)
                backend = tjm.analog_tjm_2 if order == 2 else tjm.analog_tjm_1
                res, diag, dplog = [], [], []
                for i in range(10):
                    dps.clear()
                    r, dg, _ = backend((i, st, noise, params, H))
                    res.append(np.asarray(r, dtype=np.float64))
                    diag.append(dg)
                    dplog.append(np.array(dps + [np.nan] * (64 - len(dps)))[:64])
                key = f"order{order}_sample{int(sample)}"
                out[key + "_results"] = np.array(res)
                out[key + "_diag"] = np.array(diag)
                out[key + "_dp"] = np.array(dplog)
    finally:
        stoch.calculate_stochastic_factor = orig
    # the reference's own pinned golden (tests/test_simulator.py:191-197)
    out["pinned_expected_z"] = np.array([0.748947146695782, 0.8720515025769692, 0.8652609567462763, 0.8673233347433466, 0.6872036335377433])

    # closed system, config-1-like: L=10 TFIM chi 16 order 2, traj 0 (all trajectories identical)
    L2 = 10
    H2 = MPO.ising(L2, 1, 0.5)
    out.update(pack_tensors("c1_mpo", H2.tensors))
    st2 = MPS(L2, state="zeros")
    st2.normalize("B")
    for order in (1, 2):
        p2 = sp.AnalogSimParams(observables=[sp.Observable(gl.Z(), s) for s in range(L2)], elapsed_time=1.0, dt=0.1, max_bond_dim=16,
                                svd_threshold=1e-9, krylov_tol=1e-12, order=order, sample_timesteps=True, random_seed=42)
        backend = tjm.analog_tjm_2 if order == 2 else tjm.analog_tjm_1
        r, dg, _ = backend((0, st2, None, p2, H2))
        out[f"c1_order{order}_results"] = np.asarray(r, dtype=np.float64)
        out[f"c1_order{order}_diag"] = dg
    # dephasing only (config-2-like, small): L=8 chi 8
    L3 = 8
    H3 = MPO.ising(L3, 1, 0.5)
    out.update(pack_tensors("c2_mpo", H3.tensors))
    st3 = MPS(L3, state="x+")
    st3.normalize("B")
    n3 = NoiseModel([{"name": "pauli_z", "sites": [i], "strength": 0.1} for i in range(L3)])
    p3 = sp.AnalogSimParams(observables=[sp.Observable(gl.Z(), s) for s in range(L3)] + [sp.Observable(gl.X(), s) for s in range(L3)],
                            elapsed_time=1.0, dt=0.1, max_bond_dim=8, svd_threshold=1e-12, krylov_tol=1e-12, order=1,
                            sample_timesteps=True, random_seed=42)
    res, diag, dplog = [], [], []
    stoch.calculate_stochastic_factor = spy
    try:
        for i in range(8):
            dps.clear()
            r, dg, _ = tjm.analog_tjm_1((i, st3, n3, p3, H3))
            res.append(np.asarray(r, dtype=np.float64))
            diag.append(dg)
            dplog.append(np.array(dps + [np.nan] * 16)[:16])
    finally:
        stoch.calculate_stochastic_factor = orig
    out["c2_results"] = np.array(res)
    out["c2_diag"] = np.array(diag)
    out["c2_dp"] = np.array(dplog)
    save("trajectories", **out)




# ------------------------------------------------------------------ 7. digital TEBD trajectories
def gen_digital():
    dtm = ref("digital.digital_tjm")
    L, steps = 8, 5
    J, g, dt = 1.0, 0.5, 0.1

    def layer(sample):
        singles = []
        for q in range(L):
            gt = gl.GateLibrary.rx([-2 * dt * g]); gt.set_sites(q); singles.append(gt)
        even, odd = [], []
        for q in range(0, L - 1, 2):
            gt = gl.GateLibrary.rzz([-2 * dt * J]); gt.set_sites(q, q + 1); even.append(gt)
        for q in range(1, L - 1, 2):
            gt = gl.GateLibrary.rzz([-2 * dt * J]); gt.set_sites(q, q + 1); odd.append(gt)
        return dtm._CompiledCircuitLayer(tuple(singles), tuple(even), tuple(odd), 1 if sample else 0)

    out = {}
    g0 = gl.GateLibrary.rx([-2 * dt * g]); g0.set_sites(0)
    g1 = gl.GateLibrary.rzz([-2 * dt * J]); g1.set_sites(0, 1)
    out["rx_matrix"] = np.asarray(g0.tensor)
    out["rzz_tensor"] = np.asarray(g1.tensor)
    st = MPS(L, state="zeros")
    st.normalize("B")
    noise = NoiseModel([{"name": n, "sites": [i], "strength": 0.01} for i in range(L) for n in ("pauli_x", "pauli_y", "pauli_z")])
    obs = [sp.Observable(gl.Z(), s) for s in range(L)] + [sp.Observable(gl.X(), 3)]
    for name, nm, sample in (("noisy", noise, False), ("noiseless", None, True)):
        cc = dtm._CompiledCircuit(tuple(layer(sample) for _ in range(steps)), steps if sample else 0)
        p = sp.DigitalSimParams(observables=obs, max_bond_dim=16, svd_threshold=1e-9, random_seed=3, sample_layers=sample,
                                num_mid_measurements=steps if sample else 0)
        res, diag = [], []
        for i in range(6 if nm is not None else 1):
            r, dg, _, _ = dtm.digital_tjm((i, st, nm, p, None), compiled_circuit=cc)
            res.append(np.asarray(r, dtype=np.float64))
            diag.append(dg)
        out[name + "_results"] = np.array(res)
        out[name + "_diag"] = np.array(diag)
    # strong noise with non-Pauli channel and truncation: lowering + dephasing, chi 4
    noise2 = NoiseModel([{"name": n, "sites": [i], "strength": 0.1} for i in range(L) for n in ("lowering", "pauli_z")])
    cc = dtm._CompiledCircuit(tuple(layer(False) for _ in range(3)), 0)
    p = sp.DigitalSimParams(observables=obs, max_bond_dim=4, svd_threshold=1e-6, random_seed=7)
    res, diag = [], []
    for i in range(6):
        r, dg, _, _ = dtm.digital_tjm((i, st, noise2, p, None), compiled_circuit=cc)
        res.append(np.asarray(r, dtype=np.float64))
        diag.append(dg)
    out["strong_results"] = np.array(res)
    out["strong_diag"] = np.array(diag)
    # long-range gates (SWAP-routed TEBD, digital_tjm.py:476-499) in both site orders, with one- and two-site local noise
    def lr_layer():
        singles = []
        for q in range(L):
            gt = gl.GateLibrary.rx([0.3 + 0.1 * q]); gt.set_sites(q); singles.append(gt)
        a = gl.GateLibrary.cx(); a.set_sites(1, 5)
        b = gl.GateLibrary.rzz([0.7]); b.set_sites(6, 2)
        c = gl.GateLibrary.cx(); c.set_sites(4, 3)
        d_ = gl.GateLibrary.cx(); d_.set_sites(7, 0)
        return dtm._CompiledCircuitLayer(tuple(singles), (a, b), (c, d_), 0)

    out["lr_cx_matrix"] = np.asarray(gl.GateLibrary.cx().matrix)
    out["lr_rzz_matrix"] = np.asarray(gl.GateLibrary.rzz([0.7]).matrix)
    noise3 = NoiseModel([{"name": "pauli_x", "sites": [i], "strength": 0.05} for i in range(L)] +
                        [{"name": "crosstalk_zz", "sites": [1, 5], "strength": 0.1}, {"name": "lowering", "sites": [6], "strength": 0.2}])
    cc = dtm._CompiledCircuit(tuple(lr_layer() for _ in range(2)), 0)
    # gate_mode="swaps" is the routed-TEBD mode (digital_tjm.py:604-605); chi = 4 makes the truncations of the SWAP chain bite,
    # so the fixture distinguishes it from the default gate-MPO route
    p = sp.DigitalSimParams(observables=obs, max_bond_dim=4, svd_threshold=1e-8, random_seed=11, gate_mode="swaps")
    for name, nm, ntraj in (("lr_noisy", noise3, 6), ("lr_noiseless", None, 1)):
        res, diag = [], []
        for i in range(ntraj):
            r, dg, _, _ = dtm.digital_tjm((i, st, nm, p, None), compiled_circuit=cc)
            res.append(np.asarray(r, dtype=np.float64))
            diag.append(dg)
        out[name + "_results"] = np.array(res)
        out[name + "_diag"] = np.array(diag)
    save("digital", **out)


def gen_config1_tebd():
    """BASELINE config 1 in its TEBD variant at the stated shape: the gate sequence of
    create_ising_circuit(10, 1, 0.5, 0.1, 10) (circuit_library.py:28-79: per time step an rx layer, rzz on the even bonds, rzz on the
    odd bonds), max_bond_dim 16, from |0...0>, <Z_i> and <X_3> after every step: the closed system (one deterministic run) and, so that
    the 8 trajectories of the row differ, the same circuit with depolarising noise 0.01 after every gate (trajectories 0 ... 7)."""
    dtm = ref("digital.digital_tjm")
    L, steps = 10, 10
    J, g, dt = 1.0, 0.5, 0.1

    def layer():
        singles = []
        for q in range(L):
            gt = gl.GateLibrary.rx([-2 * dt * g]); gt.set_sites(q); singles.append(gt)
        even, odd = [], []
        for q in range(0, L - 1, 2):
            gt = gl.GateLibrary.rzz([-2 * dt * J]); gt.set_sites(q, q + 1); even.append(gt)
        for q in range(1, L - 1, 2):
            gt = gl.GateLibrary.rzz([-2 * dt * J]); gt.set_sites(q, q + 1); odd.append(gt)
        return dtm._CompiledCircuitLayer(tuple(singles), tuple(even), tuple(odd), 1)

    st = MPS(L, state="zeros")
    st.normalize("B")
    obs = [sp.Observable(gl.Z(), s) for s in range(L)] + [sp.Observable(gl.X(), 3)]
    noise = NoiseModel([{"name": n, "sites": [i], "strength": 0.01} for i in range(L) for n in ("pauli_x", "pauli_y", "pauli_z")])
    cc = dtm._CompiledCircuit(tuple(layer() for _ in range(steps)), steps)
    p = sp.DigitalSimParams(observables=obs, max_bond_dim=16, svd_threshold=1e-9, random_seed=5, sample_layers=True, num_mid_measurements=steps)
    out = {}
    for name, nm, ntraj in (("closed", None, 1), ("noisy", noise, 8)):
        res, diag = [], []
        for i in range(ntraj):
            r, dg, _, _ = dtm.digital_tjm((i, st, nm, p, None), compiled_circuit=cc)
            res.append(np.asarray(r, dtype=np.float64))
            diag.append(dg)
        out[name + "_results"] = np.array(res)
        out[name + "_diag"] = np.array(diag)
    save("config1_tebd", **out)


def gen_shots():
    """MPS.measure_single_shot (mps.py:1282-1350) with scripted draws: rng.choice(n, p) replaced by its definition
    (searchsorted on the normalised cumulative sum) so that the fixture records the uniforms."""

    class ScriptRng:
        def __init__(self, u):
            self.u = list(u)

        def choice(self, n, p):
            u = self.u.pop(0)
            cdf = np.cumsum(p)
            cdf /= cdf[-1]
            return int(np.searchsorted(cdf, u, side="right"))

    rng = np.random.default_rng(17)
    L, chi = 6, 8
    tensors = []
    for i in range(L):
        cl, cr = min(2 ** i, 2 ** (L - i), chi), min(2 ** (i + 1), 2 ** (L - i - 1), chi)
        tensors.append(rng.normal(size=(2, cl, cr)) + 1j * rng.normal(size=(2, cl, cr)))
    m = MPS(L, tensors=[t.copy() for t in tensors])
    m.normalize("B")
    out = {f"t{i}": m.tensors[i] for i in range(L)}
    u = rng.random((3, 40, L))
    codes = np.zeros((3, 40), dtype=np.int64)
    for bi, basis in enumerate("ZXY"):
        for k in range(40):
            codes[bi, k] = m.measure_single_shot(basis, rng=ScriptRng(u[bi, k]))
    out["uniforms"] = u
    out["codes"] = codes
    # entropy / Schmidt spectrum / bitstring projection on the same state (mps.py:604-678, 1495-1537)
    out["entropy"] = np.array([m.get_entropy([i, i + 1]) for i in range(L - 1)])
    out["schmidt"] = np.array([m.get_schmidt_spectrum([i, i + 1]) for i in range(L - 1)])
    strings = ["000000", "101010", "111111", "010011"]
    out["pvm_strings"] = np.array(strings)
    out["pvm"] = np.array([complex(m.project_onto_bitstring(b)).real for b in strings])
    save("shots", **out)


def gen_scheduled():
    """analog_tjm_1 with NoiseModel.scheduled_jumps (scheduled_jumps.py:51-119): one-site jumps at t = 0 and mid-run, an adjacent
    two-site jump, on top of stochastic dephasing."""
    atjm = ref("analog.analog_tjm")
    L = 6
    H = MPO.ising(L, 1.0, 0.5)
    st = MPS(L, state="x+")
    two = np.kron(np.array([[0, 1], [0, 0]]), np.array([[1, 0], [0, -1]])).astype(complex)
    sched = [{"time": 0.0, "sites": [2], "name": "pauli_x"}, {"time": 0.2, "sites": [4], "name": "lowering"},
             {"time": 0.3, "sites": [1, 2], "name": "custom", "matrix": two}]
    noise = NoiseModel([{"name": "pauli_z", "sites": [i], "strength": 0.2} for i in range(L)], scheduled_jumps=sched)
    obs = [sp.Observable(gl.Z(), s) for s in range(L)] + [sp.Observable(gl.X(), 0)]
    p = sp.AnalogSimParams(observables=obs, elapsed_time=0.5, dt=0.1, max_bond_dim=8, svd_threshold=1e-10, krylov_tol=1e-10, order=1,
                           sample_timesteps=True, random_seed=21)
    res, diag = [], []
    for i in range(4):
        r, dg, _ = atjm.analog_tjm_1((i, st, noise, p, H))
        res.append(np.asarray(r, dtype=np.float64))
        diag.append(dg)
    out = {"results": np.array(res), "diag": np.array(diag), "two": two}
    for i, w in enumerate(H.tensors):
        out[f"mpo{i}"] = w
    save("scheduled", **out)


def gen_piecewise():
    """Piecewise-constant Hamiltonian: a tuple of MPOs, one per time interval (analog_tjm.py:43-49, 351-360), order 1 and 2."""
    atjm = ref("analog.analog_tjm")
    L, n = 6, 4
    gs = [0.3, 0.9, -0.4, 0.6]
    hams = tuple(MPO.ising(L, 1.0, g) for g in gs)
    st = MPS(L, state="x+")
    noise = NoiseModel([{"name": "pauli_z", "sites": [i], "strength": 0.1} for i in range(L)])
    obs = [sp.Observable(gl.Z(), s) for s in range(L)] + [sp.Observable(gl.X(), 2)]
    out = {"g": np.array(gs)}
    for k, h in enumerate(hams):
        for i, w in enumerate(h.tensors):
            out[f"h{k}_mpo{i}"] = w
    for order in (1, 2):
        p = sp.AnalogSimParams(observables=obs, elapsed_time=0.1 * n, dt=0.1, max_bond_dim=8, svd_threshold=1e-10, krylov_tol=1e-10, order=order,
                               sample_timesteps=True, random_seed=5)
        fn = atjm.analog_tjm_1 if order == 1 else atjm.analog_tjm_2
        res, diag = [], []
        for i in range(3):
            r, dg, _ = fn((i, st, noise, p, hams))
            res.append(np.asarray(r, dtype=np.float64))
            diag.append(dg)
        out[f"order{order}_results"] = np.array(res)
        out[f"order{order}_diag"] = np.array(diag)
    save("piecewise", **out)


def gen_tiny():
    """One- and two-site chains (tdvp.py:96-100: a single site falls back to 1TDVP), order 1 and 2, amplitude damping."""
    atjm = ref("analog.analog_tjm")
    out = {}
    for L in (1, 2):
        H = MPO.ising(L, 1.0, 0.7)
        st = MPS(L, state="x+")
        noise = NoiseModel([{"name": "lowering", "sites": [i], "strength": 0.3} for i in range(L)])
        obs = [sp.Observable(gl.Z(), L - 1), sp.Observable(gl.X(), 0)]
        for i, w in enumerate(H.tensors):
            out[f"L{L}_mpo{i}"] = w
        for order in (1, 2):
            p = sp.AnalogSimParams(observables=obs, elapsed_time=0.3, dt=0.1, max_bond_dim=4, svd_threshold=1e-10, krylov_tol=1e-10, order=order,
                                   sample_timesteps=True, random_seed=13)
            fn = atjm.analog_tjm_1 if order == 1 else atjm.analog_tjm_2
            res = []
            for i in range(5):
                r, _, _ = fn((i, st, noise, p, H))
                res.append(np.asarray(r, dtype=np.float64))
            out[f"L{L}_order{order}"] = np.array(res)
    save("tiny", **out)


# ------------------------------------------------------------------ 12. BASELINE.json configurations at their stated sizes
def _fullsize_inputs(L, chi, seed=1):
    """The chi-saturated Haar state of bench.py / SURVEY section 8d, built by the package's own host-side builder (no GPU), so the
    GPU test regenerates the identical input from the seed; handed to the reference as MPS(tensors=...)."""
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from yaqs_amd import api

    st = api.MPS(L, state="haar-random", pad=chi, rng=np.random.default_rng(seed))
    st.normalize("B")
    return api, [np.array(t, dtype=np.complex128) for t in st.tensors]


def _one_step(name, L, chi, mpo_tensors, proc, gamma, dt, tdvp_mode, tensors, trajs, out, krylov_tol=1e-10):
    import time

    H = MPO()  # tensors in the reference's (phys_out, phys_in, chi_l, chi_r) order already
    H.tensors = [np.asarray(w, dtype=np.complex128) for w in mpo_tensors]
    H.length = L
    H.physical_dimension = 2
    noise = NoiseModel([{"name": proc, "sites": [i], "strength": gamma} for i in range(L)])
    obs = [sp.Observable(gl.Z(), s) for s in range(L)]
    p = sp.AnalogSimParams(observables=obs, elapsed_time=dt, dt=dt, max_bond_dim=chi, svd_threshold=1e-12, krylov_tol=krylov_tol, order=1,
                           sample_timesteps=False, random_seed=42, tdvp_mode=tdvp_mode, get_state=True)
    dps = []
    orig = stoch.calculate_stochastic_factor

    def spy(state):
        v = orig(state)
        dps.append(float(v))
        return v

    stoch.calculate_stochastic_factor = spy
    res, dplog, bonds, diag, ulog = [], [], [], [], []
    try:
        for t in trajs:
            dps.clear()
            t0 = time.time()
            st = MPS(L, tensors=[x.copy() for x in tensors])
            r, dg, final = tjm.analog_tjm_1((t, st, noise, p, H))
            print(f"  {name}: trajectory {t} took {time.time() - t0:.1f} s, dp {dps}", flush=True)
            res.append(np.asarray(r, dtype=np.float64)[:, 0])
            diag.append(np.asarray(dg, dtype=np.float64)[:, 0])
            dplog.append(dps[0])
            bonds.append([final.tensors[0].shape[1]] + [x.shape[2] for x in final.tensors])
            ulog.append(rutil.make_trajectory_rng(t, base_seed=42).random())
    finally:
        stoch.calculate_stochastic_factor = orig
    out[name + "_traj"] = np.array(trajs)
    out[name + "_z"] = np.array(res)
    out[name + "_diag"] = np.array(diag)
    out[name + "_dp"] = np.array(dplog)
    out[name + "_bonds"] = np.array(bonds)
    out[name + "_u0"] = np.array(ulog)


def gen_fullsize(which=("cfg2", "cfg4", "cfg3")):
    """One order-1 TJM step of the reference at BASELINE.json's sizes (krylov_tol 1e-10): per-site <Z>, dp, final bond dimensions,
    diagnostics.  cfg2: L=64 chi=128 dissipative TFIM; cfg4: L=32 chi=256 long-range Ising, one-site TDVP; cfg3: L=128 chi=256 XXZ
    with amplitude damping.  The inputs are regenerated from seeds by the test (nothing large is stored)."""
    path = os.path.join(OUT, "fullsize.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    if "cfg2" in which:
        api, tensors = _fullsize_inputs(64, 128)
        # trajectory 0 does not jump (u = 0.858 >= dp), trajectory 3 does (first uniform 0.0858...): chosen from the stream table
        us = [rutil.make_trajectory_rng(t, base_seed=42).random() for t in range(16)]
        jumper = next(t for t in range(16) if us[t] < 0.3)
        _one_step("cfg2", 64, 128, api.MPO.ising(64, 1.0, 0.5).tensors, "pauli_z", 0.1, 0.1, "2site", tensors, [0, jumper], out)
    if "cfg4" in which:
        api, tensors = _fullsize_inputs(32, 256)
        mpo = api.MPO.long_range_ising(32, [0.8792, 0.1208], [0.0717, 0.5136], 0.5)
        # dp = 0.077 per step here: trajectory 0 does not jump; the first one whose opening uniform is below 0.05 does
        us = [rutil.make_trajectory_rng(t, base_seed=42).random() for t in range(64)]
        jumper = next(t for t in range(64) if us[t] < 0.05)
        _one_step("cfg4", 32, 256, mpo.tensors, "pauli_z", 0.05, 0.05, "1site", tensors, [0, jumper], out)
    if "cfg3" in which:
        api, tensors = _fullsize_inputs(128, 256)
        # amplitude damping on a Haar state: dp ~ 0.15 per step; one trajectory that does not jump, one that does (non-Pauli jump
        # operator: the QR walk, the state-dependent jump weights and the SVD sweep back at chi = 256)
        us = [rutil.make_trajectory_rng(t, base_seed=42).random() for t in range(64)]
        jumper = next(t for t in range(64) if us[t] < 0.05)
        quiet = next(t for t in range(64) if us[t] > 0.5)
        _one_step("cfg3", 128, 256, api.MPO.heisenberg(128, 1.0, 1.0, 0.5, 0.0).tensors, "lowering", 0.05, 0.05, "2site", tensors, [quiet, jumper], out)
    save("fullsize", **out)


def _steady_one(traj, steps=10, krylov_tol=1e-10, sample=True):
    """One trajectory of gen_fullsize_steady (runs in a forked worker)."""
    import time

    api, tensors = _fullsize_inputs(64, 128)
    H = MPO()
    H.tensors = [np.asarray(w, dtype=np.complex128) for w in api.MPO.ising(64, 1.0, 0.5).tensors]
    H.length = 64
    H.physical_dimension = 2
    noise = NoiseModel([{"name": "pauli_z", "sites": [i], "strength": 0.1} for i in range(64)])
    obs = [sp.Observable(gl.Z(), s) for s in range(64)]
    p = sp.AnalogSimParams(observables=obs, elapsed_time=0.1 * steps, dt=0.1, max_bond_dim=128, svd_threshold=1e-12, krylov_tol=krylov_tol,
                           order=1, sample_timesteps=sample, random_seed=42, tdvp_mode="2site", get_state=True)
    dps, jumped, bonds = [], [], []
    orig_f, orig_pdf = stoch.calculate_stochastic_factor, stoch.create_probability_distribution

    def spy_f(state):
        v = orig_f(state)
        dps.append(float(v))
        jumped.append(0)
        return v

    def spy_pdf(state, *a, **k):
        jumped[-1] = 1
        # bonds of the state as the jump branch meets it (after the dissipation sweep of this step)
        return orig_pdf(state, *a, **k)

    orig_sp = stoch.stochastic_process

    def spy_sp(state, *a, **k):
        out = orig_sp(state, *a, **k)
        bonds.append([out.tensors[0].shape[1]] + [x.shape[2] for x in out.tensors])
        print(f"  fullsize_steady: trajectory {traj} step {len(bonds)} done after {time.time() - t0:.0f} s, dp {dps[-1]:.6f} jumped {jumped[-1]}", flush=True)
        return out

    stoch.calculate_stochastic_factor = spy_f
    stoch.create_probability_distribution = spy_pdf
    tjm.stochastic_process = spy_sp
    t0 = time.time()
    try:
        st = MPS(64, tensors=[x.copy() for x in tensors])
        r, dg, final = tjm.analog_tjm_1((traj, st, noise, p, H))
    finally:
        stoch.calculate_stochastic_factor, stoch.create_probability_distribution = orig_f, orig_pdf
        tjm.stochastic_process = orig_sp
    print(f"  fullsize_steady: trajectory {traj} took {time.time() - t0:.1f} s, dp {dps}, jumped {jumped}", flush=True)
    return dict(z=np.asarray(r, dtype=np.float64), diag=np.asarray(dg, dtype=np.float64), dp=np.array(dps), jumped=np.array(jumped),
                bonds=np.array(bonds))


def gen_fullsize_steady_final(trajs=(0, 1, 2), steps=10):
    """gen_fullsize_steady with final-time sampling only (sample_timesteps=False): the reference's measurement of a state whose gauge
    it does not know (the initial sample of a run from MPS(tensors=...)) contracts the whole chain once per observable, which at
    chi = 128 costs more than the ten steps; <Z_i> is then stored for the final time only, everything else as below."""
    gen_fullsize_steady(trajs, steps, sample=False)


def gen_fullsize_steady_bench(trajs=(0, 1, 2, 3, 4, 5, 6, 7), steps=10):
    """The same ten consecutive steps with the EXACT parameters of bench.py's timed region - krylov_tol 1e-4 (the fixtures
    above use 1e-10) - and eight trajectories: tests/golden/fullsize_steady_tol4.npz.  Final-time sampling as in
    gen_fullsize_steady_final.  About 5 minutes on eight cores."""
    gen_fullsize_steady(trajs, steps, sample=False, krylov_tol=1e-4, name="fullsize_steady_tol4")


def gen_fullsize_steady(trajs=(0, 1, 2), steps=10, sample=True, krylov_tol=1e-10, name="fullsize_steady"):
    """The reference's analog_tjm_1 on BASELINE's config 2 (L=64, chi=128 Haar-saturated, pauli_z 0.1 on every site, dt 0.1,
    svd_threshold 1e-12, krylov_tol 1e-10) for `steps` CONSECUTIVE steps: the state bench.py's timed region is in after its first
    steps, where the certified scalar dissipation / in-place jumps of the engine serve most trajectory-steps.  Per trajectory:
    <Z_i> at every time point, diagnostics, dp of every step, which steps jumped, the bond table after every step."""
    import multiprocessing as mp

    with mp.get_context("fork").Pool(len(trajs)) as pool:
        rows = pool.starmap(_steady_one, [(t, steps, krylov_tol, sample) for t in trajs])
    out = {"traj": np.array(trajs), "steps": np.array(steps), "sample_timesteps": np.array(int(sample))}
    if name != "fullsize_steady":  # (the round-4 fixture keeps its keys: it regenerates bit for bit)
        out["krylov_tol"] = np.array(krylov_tol)
    for k in rows[0]:
        out[k] = np.array([r[k] for r in rows])
    save(name, **out)


# ------------------------------------------------------------------ 12b. configs 4 and 3: consecutive steps at the bench's krylov_tol
STEADY_CFGS = {
    # name: (L, chi, MPO builder, noise process, gamma, dt, tdvp_mode, steps)
    "cfg4": (32, 256, lambda api: api.MPO.long_range_ising(32, [0.8792, 0.1208], [0.0717, 0.5136], 0.5), "pauli_z", 0.05, 0.05, "1site", 10),
    "cfg3": (128, 256, lambda api: api.MPO.heisenberg(128, 1.0, 1.0, 0.5, 0.0), "lowering", 0.05, 0.05, "2site", 3),
}


def _steady_cfg_one(cfg, traj, steps, krylov_tol):
    """One trajectory of gen_fullsize_steady_cfg (forked worker): the reference's analog_tjm_1 for `steps` consecutive steps of
    BASELINE config 3 or 4 at full size, final-time sampling; dp / jump decision / bond table of every step via spies."""
    import time

    L, chi, make_mpo, proc, gamma, dt, tdvp_mode, _ = STEADY_CFGS[cfg]
    api, tensors = _fullsize_inputs(L, chi)
    H = MPO()
    H.tensors = [np.asarray(w, dtype=np.complex128) for w in make_mpo(api).tensors]
    H.length = L
    H.physical_dimension = 2
    noise = NoiseModel([{"name": proc, "sites": [i], "strength": gamma} for i in range(L)])
    obs = [sp.Observable(gl.Z(), s) for s in range(L)]
    p = sp.AnalogSimParams(observables=obs, elapsed_time=dt * steps, dt=dt, max_bond_dim=chi, svd_threshold=1e-12, krylov_tol=krylov_tol,
                           order=1, sample_timesteps=False, random_seed=42, tdvp_mode=tdvp_mode, get_state=True)
    dps, jumped, bonds = [], [], []
    orig_f, orig_pdf, orig_sp = stoch.calculate_stochastic_factor, stoch.create_probability_distribution, tjm.stochastic_process

    def spy_f(state):
        v = orig_f(state)
        dps.append(float(v))
        jumped.append(0)
        return v

    def spy_pdf(state, *a, **k):
        jumped[-1] = 1
        return orig_pdf(state, *a, **k)

    def spy_sp(state, *a, **k):
        out = orig_sp(state, *a, **k)
        bonds.append([out.tensors[0].shape[1]] + [x.shape[2] for x in out.tensors])
        print(f"  {cfg}: trajectory {traj} step {len(bonds)} done after {time.time() - t0:.0f} s, dp {dps[-1]:.6f} jumped {jumped[-1]}", flush=True)
        return out

    stoch.calculate_stochastic_factor = spy_f
    stoch.create_probability_distribution = spy_pdf
    tjm.stochastic_process = spy_sp
    t0 = time.time()
    try:
        st = MPS(L, tensors=[x.copy() for x in tensors])
        r, dg, final = tjm.analog_tjm_1((traj, st, noise, p, H))
    finally:
        stoch.calculate_stochastic_factor, stoch.create_probability_distribution = orig_f, orig_pdf
        tjm.stochastic_process = orig_sp
    print(f"  {cfg}: trajectory {traj} took {time.time() - t0:.1f} s, dp {dps}, jumped {jumped}", flush=True)
    return dict(z=np.asarray(r, dtype=np.float64), diag=np.asarray(dg, dtype=np.float64), dp=np.array(dps), jumped=np.array(jumped),
                bonds=np.array(bonds))


def _pick_trajs(cfg):
    """Trajectories chosen from the reference's stream table alone (make_trajectory_rng, base seed 42): one that starts quietly,
    one whose opening uniform jumps at step 1, one that passes step 1 and has a small uniform right after (a jump once the bonds and
    gauges are the run's own)."""
    us = np.array([rutil.make_trajectory_rng(t, base_seed=42).random(4) for t in range(4000)])
    quiet = int(next(t for t in range(4000) if us[t, 0] > 0.5 and us[t, 1] > 0.5))
    early = int(next(t for t in range(4000) if us[t, 0] < 0.05 and us[t, 2] < 0.07))  # jumps at step 1 and again at step 2
    late = int(next(t for t in range(4000) if us[t, 0] > 0.3 and us[t, 1] < 0.04))  # quiet step 1, jumps at step 2
    return [quiet, early, late] if cfg == "cfg4" else [early, late]


def gen_fullsize_steady_cfg4():
    gen_fullsize_steady_cfg("cfg4")


def gen_fullsize_steady_cfg3():
    gen_fullsize_steady_cfg("cfg3")


def gen_fullsize_steady_cfg(cfg, krylov_tol=1e-4):
    """tests/golden/fullsize_steady_<cfg>.npz: what fullsize_steady_tol4 is for config 2, for the other two analog BASELINE rows - the
    states bench.py --config 3 / 4 time after their first step (krylov_tol 1e-4 as the bench).  config 4: L = 32, chi = 256, one-site
    TDVP, long-range Ising MPO, pauli_z 0.05, ten consecutive steps, three trajectories; config 3: L = 128, chi = 256, XXZ with
    amplitude damping (non-Pauli jumps), three consecutive steps, two trajectories.  One forked worker per trajectory."""
    import multiprocessing as mp

    steps = STEADY_CFGS[cfg][7]
    trajs = _pick_trajs(cfg)
    print(cfg, "trajectories", trajs, flush=True)
    with mp.get_context("fork").Pool(len(trajs)) as pool:
        rows = pool.starmap(_steady_cfg_one, [(cfg, t, steps, krylov_tol) for t in trajs])
    out = {"traj": np.array(trajs), "steps": np.array(steps), "krylov_tol": np.array(krylov_tol)}
    for k in rows[0]:
        out[k] = np.array([r[k] for r in rows])
    save("fullsize_steady_" + cfg, **out)


# ------------------------------------------------------------------ 13. dynamic TDVP and the BUG integrator
def gen_f3():
    """One call of tdvp(tdvp_mode="dynamic") (integrators.py:294-511) and of bug() (bug.py:213-257) on small chains - bonds below,
    at and above the cap so that both branches of the dynamic sweep run - and whole noisy trajectories in both modes."""
    bugmod = ref("core.methods.bug")
    out = {}
    cases = []
    for L, chi0, cap, state_kind in ((4, 2, 4, "haar"), (5, 4, 4, "haar"), (10, 4, 8, "haar"), (6, 1, 4, "x+"), (8, 8, 8, "haar"), (7, 2, None, "haar")):
        key = f"L{L}_c{chi0}_cap{cap}_{state_kind}"
        cases.append(key)
        H = MPO.ising(L, 1.0, 0.7)
        out.update(pack_tensors(key + "_mpo", H.tensors))
        m = haar_mps(L, chi0, 100 + L) if state_kind == "haar" else MPS(L, state="x+")
        m.normalize("B")
        out.update(pack_tensors(key + "_in", m.tensors))
        for mode in ("dynamic", "bug"):
            st = copy.deepcopy(m)
            p = sp.AnalogSimParams(observables=[sp.Observable(gl.Z(), 0)], elapsed_time=0.1, dt=0.1, max_bond_dim=cap, svd_threshold=1e-9,
                                   krylov_tol=1e-12, tdvp_mode="dynamic" if mode == "dynamic" else "2site",
                                   evolution_mode="bug" if mode == "bug" else "tdvp")
            if mode == "dynamic":
                tdvp_mod.tdvp(st, H, p)
            else:
                bugmod.bug(st, H, p)
            out[f"{key}_{mode}_vec"] = st.to_vec()
            out[f"{key}_{mode}_bonds"] = np.array([t.shape[2] for t in st.tensors])
            out[f"{key}_{mode}_norm"] = np.array(st.norm())
    out["cases"] = np.array(cases)
    # whole trajectories
    L = 6
    H = MPO.ising(L, 1.0, 0.5)
    out.update(pack_tensors("traj_mpo", H.tensors))
    noise = NoiseModel([{"name": n, "sites": [i], "strength": 0.1} for i in range(L) for n in ("lowering", "pauli_z")])
    # The trajectories start from a generic (Haar, chi = 2) state: from a product state the stacked trial bases of BUG contain
    # exactly dependent columns, their QR completion is decided by rounding noise and the reference's own result moves at the 1e-5
    # level with it (measured: two bit-different but equal inputs give overlap 0.99996 after one step).
    st = haar_mps(L, 2, 77)
    out.update(pack_tensors("traj_in", st.tensors))
    for mode in ("dynamic", "bug"):
        for order in (1, 2):
            p = sp.AnalogSimParams(observables=[sp.Observable(gl.Z(), s) for s in range(L)], elapsed_time=0.5, dt=0.1, num_traj=4, max_bond_dim=4,
                                   svd_threshold=1e-9, krylov_tol=1e-12, order=order, sample_timesteps=True, random_seed=9,
                                   tdvp_mode="dynamic" if mode == "dynamic" else "2site", evolution_mode="bug" if mode == "bug" else "tdvp")
            backend = tjm.analog_tjm_2 if order == 2 else tjm.analog_tjm_1
            res, diag = [], []
            for i in range(4):
                r, dg, _ = backend((i, st, noise, p, H))
                res.append(np.asarray(r, dtype=np.float64))
                diag.append(dg)
            out[f"traj_{mode}_order{order}_results"] = np.array(res)
            out[f"traj_{mode}_order{order}_diag"] = np.array(diag)
    save("f3_dynamic_bug", **out)


# ------------------------------------------------------------------ 14. continuation options of the drivers
def gen_continuation():
    """analog_tjm_1 / analog_tjm_2 with ``sample_at``, and an order-2 run cut into two segments the way the reference's program
    runner stitches them (analog_tjm.py:206-366): one trajectory stream shared by both segments, ``return_trajectory_state`` /
    ``continue_trajectory`` for the hand-off of phi, ``sample_timestep_offset`` for the global sample timeline."""
    L = 5
    H = MPO.ising(L, 1.0, 0.6)
    out = pack_tensors("mpo", H.tensors)
    noise = NoiseModel([{"name": n, "sites": [i], "strength": 0.15} for i in range(L) for n in ("lowering", "pauli_z")])
    st = MPS(L, state="x+")
    st.normalize("B")
    obs = [sp.Observable(gl.Z(), s) for s in range(L)]
    kw = dict(dt=0.1, max_bond_dim=4, svd_threshold=1e-9, krylov_tol=1e-12, random_seed=31)
    ntraj = 4
    # sample_at on both drivers
    for order, fn in ((1, tjm.analog_tjm_1), (2, tjm.analog_tjm_2)):
        p = sp.AnalogSimParams(observables=obs, elapsed_time=0.6, sample_timesteps=True, order=order, **kw)
        out[f"sample_at_order{order}"] = np.array([np.asarray(fn((i, st, noise, p, H), sample_at=[0, 2, 5])[0], dtype=np.float64) for i in range(ntraj)])
        p1 = sp.AnalogSimParams(observables=obs, elapsed_time=0.6, sample_timesteps=False, order=order, **kw)
        out[f"sample_at_single_order{order}"] = np.array([np.asarray(fn((i, st, noise, p1, H), sample_at=[3])[0], dtype=np.float64) for i in range(ntraj)])
    # one continuous order-2 run of 6 steps, and the same cut after 3 steps
    full = sp.AnalogSimParams(observables=obs, elapsed_time=0.6, sample_timesteps=True, order=2, **kw)
    seg = sp.AnalogSimParams(observables=obs, elapsed_time=0.3, sample_timesteps=True, order=2, **kw)
    whole, first, second, phi_bonds = [], [], [], []
    for i in range(ntraj):
        whole.append(np.asarray(tjm.analog_tjm_2((i, st, noise, full, H))[0], dtype=np.float64))
        rng = rutil.make_trajectory_rng(i, base_seed=31)
        r1, _, phi = tjm.analog_tjm_2((i, st, noise, seg, H), rng=rng, return_trajectory_state=True)
        r2, _, phi2 = tjm.analog_tjm_2((i, phi, noise, seg, H), rng=rng, sample_timestep_offset=3, continue_trajectory=True,
                                       return_trajectory_state=True)
        first.append(np.asarray(r1, dtype=np.float64))
        second.append(np.asarray(r2, dtype=np.float64))
        phi_bonds.append([t.shape[2] for t in phi2.tensors])
    out["whole"], out["segment1"], out["segment2"], out["phi_bonds"] = np.array(whole), np.array(first), np.array(second), np.array(phi_bonds)
    save("continuation", **out)


# ------------------------------------------------------------------ 15. front-end scenarios (replace transcribed reference tests)
def gen_scenarios():
    """Reference outputs for the scenarios of tests/test_front_end_scenarios.py: every scenario is described by its inputs in that
    test file; here the reference's own backends produce what `Simulator.run` must return for them."""
    out = {}

    def mean_rows(backend, n, st, noise, p, H):
        rows = [np.asarray(backend((i, st, noise, p, H))[0], dtype=np.float64) for i in range(n)]
        return np.mean(rows, axis=0), np.array(rows)

    # noisy order-2 run with final-time sampling (the configuration of the reference's own end-to-end analog test)
    L = 5
    H = MPO.ising(L, 1, 0.5)
    st = MPS(L, state="zeros")
    noise = NoiseModel([{"name": n, "sites": [i], "strength": 0.1} for i in range(L) for n in ("lowering", "pauli_z")])
    p = sp.AnalogSimParams(observables=[sp.Observable(gl.Z(), s) for s in range(L)], elapsed_time=1, dt=0.1, num_traj=10, max_bond_dim=4,
                           svd_threshold=1e-6, order=2, sample_timesteps=False, random_seed=42)
    out["noisy_order2_mean"], out["noisy_order2_rows"] = mean_rows(tjm.analog_tjm_2, 10, st, noise, p, H)
    # user order of the observables, closed two-site chain, exact preset, final state
    H2 = MPO.ising(2, 1.0, 0.7)
    p = sp.AnalogSimParams(observables=[sp.Observable(gl.Z(), 1), sp.Observable(gl.X(), 0), sp.Observable(gl.Z(), 0)], elapsed_time=0.1, dt=0.1,
                           num_traj=1, get_state=True, sample_timesteps=False, preset="exact")
    r, _, fin = tjm.analog_tjm_1((0, MPS(2, state="zeros"), None, p, H2))
    out["order_sorted_rows"] = np.asarray(r, dtype=np.float64)[:, 0]
    out["order_sorted_index"] = np.array(p.observable_sorted_indices if hasattr(p, "observable_sorted_indices") else [1, 2, 0])
    out["order_final_vec"] = fin.to_vec()
    # closed two-site run to T = 1, both orders: final state vector
    for order in (1, 2):
        p = sp.AnalogSimParams(observables=[sp.Observable(gl.X(), 1)], elapsed_time=1, dt=0.1, num_traj=1, max_bond_dim=4, svd_threshold=1e-6,
                               order=order, get_state=True, sample_timesteps=False)
        backend = tjm.analog_tjm_2 if order == 2 else tjm.analog_tjm_1
        r, _, fin = backend((0, MPS(2, state="zeros"), None, p, MPO.ising(2, 1, 0.5)))
        out[f"closed2_order{order}_vec"] = fin.to_vec()
        out[f"closed2_order{order}_x"] = np.asarray(r, dtype=np.float64)[:, 0]
    # order-2 short runs
    for T, sample in ((0.0, True), (0.0, False), (0.1, False), (0.1, True)):
        p = sp.AnalogSimParams(observables=[sp.Observable(gl.Z(), 0)], dt=0.1, elapsed_time=T, num_traj=1, order=2, sample_timesteps=sample,
                               get_state=True, random_seed=0)
        r, _, _ = tjm.analog_tjm_2((0, MPS(2, state="zeros"), None, p, MPO.ising(2, 1.0, 0.5)))
        out[f"short_T{T}_s{int(sample)}"] = np.asarray(r, dtype=np.float64)[0]
    # one-site plus adjacent two-site jump processes, 20 trajectories, order 2
    for name in ("crosstalk_xx", "lowering_two"):
        p = sp.AnalogSimParams(observables=[sp.Observable(gl.Z(), 0)], elapsed_time=0.1, dt=0.1, num_traj=20, max_bond_dim=8, order=2,
                               sample_timesteps=False, random_seed=42)
        nm = NoiseModel([{"name": "pauli_x", "sites": [0], "strength": 0.02}, {"name": name, "sites": [0, 1], "strength": 0.01}])
        out[f"pair_{name}_mean"], out[f"pair_{name}_rows"] = mean_rows(tjm.analog_tjm_2, 20, MPS(2, state="zeros"), nm, p, MPO.ising(2, 1.0, 0.5))
    # long-range Pauli crosstalk on the analog path
    p = sp.AnalogSimParams(observables=[sp.Observable(gl.Z(), 0)], dt=0.1, elapsed_time=0.2, num_traj=2, random_seed=0)
    nm = NoiseModel([{"name": "longrange_crosstalk_xy", "sites": [0, 2], "strength": 0.05}])
    out["longrange_mean"], _ = mean_rows(tjm.analog_tjm_2 if p.order == 2 else tjm.analog_tjm_1, 2, MPS(3, state="zeros"), nm, p, MPO.ising(3, 1.0, 0.5))
    save("front_end_scenarios", **out)


# ------------------------------------------------------------------ 16. qutrit / four-level chains
def gen_qudit():
    """The reference on chains with local dimension 3 and 4: MPO.bose_hubbard (mpo.py:670-745), Fock product states through
    MPS(physical_dimensions=...), one-site loss / dephasing with custom d x d matrices, occupation observables; one closed TDVP
    step from a random state and noisy trajectories of both drivers."""

    out = {}
    cases = []
    for d, L, chi in ((3, 5, 9), (4, 4, 8)):
        key = f"d{d}_L{L}"
        cases.append(key)
        H = MPO.bose_hubbard(L, d, 0.7, 0.6, 0.5)
        out.update(pack_tensors(key + "_mpo", H.tensors))
        b = np.diag(np.sqrt(np.arange(1, d)), 1).astype(complex)
        n = b.conj().T @ b
        # closed two-site TDVP step from a seeded random state
        rng = np.random.default_rng(d * 10 + L)
        caps = [1] * (L + 1)
        for i in range(1, L):
            caps[i] = min(d ** i, d ** (L - i), chi)
        st = MPS(L, tensors=[rng.standard_normal((d, caps[i], caps[i + 1])) + 1j * rng.standard_normal((d, caps[i], caps[i + 1])) for i in range(L)],
                 physical_dimensions=[d] * L)
        st.normalize("B")
        out.update(pack_tensors(key + "_in", st.tensors))
        p = sp.AnalogSimParams(observables=[sp.Observable(n, 0)], elapsed_time=0.05, dt=0.05, max_bond_dim=chi, svd_threshold=1e-10, krylov_tol=1e-12)
        work = MPS(L, tensors=[t.copy() for t in st.tensors], physical_dimensions=[d] * L)
        ref("core.methods.tdvp.tdvp").tdvp(work, H, p)
        out[key + "_tdvp_vec"] = work.to_vec()
        out[key + "_tdvp_bonds"] = np.array([t.shape[2] for t in work.tensors])
        # noisy trajectories from a Fock state
        basis = "".join(str((i + 1) % d) for i in range(L))
        fock = MPS(L, physical_dimensions=[d] * L, state="basis", basis_string=basis)
        noise = NoiseModel([{"name": "loss", "sites": [i], "strength": 0.3, "matrix": b} for i in range(L)]
                           + [{"name": "dephasing", "sites": [i], "strength": 0.1, "matrix": n} for i in range(L)])
        for order, fn in ((1, tjm.analog_tjm_1), (2, tjm.analog_tjm_2)):
            p = sp.AnalogSimParams(observables=[sp.Observable(n, s) for s in range(L)], elapsed_time=0.4, dt=0.1, num_traj=3, max_bond_dim=chi,
                                   svd_threshold=1e-10, krylov_tol=1e-12, order=order, sample_timesteps=True, random_seed=4)
            out[f"{key}_order{order}_results"] = np.array([np.asarray(fn((i, fock, noise, p, H))[0], dtype=np.float64) for i in range(3)])
    out["cases"] = np.array(cases)
    save("qudit", **out)


# ------------------------------------------------------------------ 17. long-range gates through the gate-MPO product (the default gate_mode)
def gen_digital_mpo():
    """digital_tjm with gate_mode="mpo" (the default; digital_tjm.py:536-557, 592-620): distant pairs go through
    MPO.from_gate(...).multiply(state, compress=True), nearest neighbours through TEBD.  Same layers as the SWAP-routed fixture of
    gen_digital, with a cap that bites (chi = 4) and one that does not (chi = 16)."""
    dtm = ref("digital.digital_tjm")
    L = 8

    def lr_layer():
        singles = []
        for q in range(L):
            gt = gl.GateLibrary.rx([0.3 + 0.1 * q]); gt.set_sites(q); singles.append(gt)
        a = gl.GateLibrary.cx(); a.set_sites(1, 5)
        b = gl.GateLibrary.rzz([0.7]); b.set_sites(6, 2)
        c = gl.GateLibrary.cx(); c.set_sites(4, 3)
        d_ = gl.GateLibrary.cx(); d_.set_sites(7, 0)
        return dtm._CompiledCircuitLayer(tuple(singles), (a, b), (c, d_), 0)

    out = {}
    st = MPS(L, state="zeros")
    st.normalize("B")
    obs = [sp.Observable(gl.Z(), s) for s in range(L)] + [sp.Observable(gl.X(), 3)]
    noise = NoiseModel([{"name": "pauli_x", "sites": [i], "strength": 0.05} for i in range(L)] +
                       [{"name": "crosstalk_zz", "sites": [1, 5], "strength": 0.1}, {"name": "lowering", "sites": [6], "strength": 0.2}])
    cc = dtm._CompiledCircuit(tuple(lr_layer() for _ in range(2)), 0)
    for chi in (4, 16):
        p = sp.DigitalSimParams(observables=obs, max_bond_dim=chi, svd_threshold=1e-8, random_seed=11)
        assert p.gate_mode == "mpo"
        for name, nm, ntraj in ((f"chi{chi}_noisy", noise, 6), (f"chi{chi}_noiseless", None, 1)):
            res, diag = [], []
            for i in range(ntraj):
                r, dg, _, _ = dtm.digital_tjm((i, st, nm, p, None), compiled_circuit=cc)
                res.append(np.asarray(r, dtype=np.float64))
                diag.append(dg)
            out[name + "_results"] = np.array(res)
            out[name + "_diag"] = np.array(diag)
    save("digital_mpo", **out)


# ------------------------------------------------------------------ 18. long-range gates by TDVP on a window (gate_mode "tdvp" / "full-tdvp")
def gen_digital_tdvp():
    """digital_tjm with gate_mode="tdvp" (distant pairs with a product-form generator by two-site TDVP on a window,
    digital_tjm.py:408-453; nearest neighbours by TEBD) and "full-tdvp" (nearest neighbours by TDVP too): rzz / rxx / ryy gates on
    distant and adjacent pairs in both site orders, local noise, with a cap (renorm_drift active) and without."""
    dtm = ref("digital.digital_tjm")
    L = 8

    def layer():
        singles = []
        for q in range(L):
            gt = gl.GateLibrary.rx([0.3 + 0.1 * q]); gt.set_sites(q); singles.append(gt)
        a = gl.GateLibrary.rzz([0.7]); a.set_sites(1, 5)
        b = gl.GateLibrary.rxx([0.4]); b.set_sites(6, 2)
        c = gl.GateLibrary.ryy([0.9]); c.set_sites(4, 3)
        d_ = gl.GateLibrary.rzz([1.1]); d_.set_sites(7, 0)
        return dtm._CompiledCircuitLayer(tuple(singles), (a, b), (c, d_), 0)

    out = {}
    for name, g_ in (("rzz07", gl.GateLibrary.rzz([0.7])), ("rxx04", gl.GateLibrary.rxx([0.4])), ("ryy09", gl.GateLibrary.ryy([0.9])), ("rzz11", gl.GateLibrary.rzz([1.1]))):
        g_.set_sites(0, 1)  # the generator is attached with the sites
        out[name + "_matrix"] = np.asarray(g_.matrix)
        out[name + "_gen0"] = np.asarray(g_.generator[0])
        out[name + "_gen1"] = np.asarray(g_.generator[1])
    st = MPS(L, state="zeros")
    st.normalize("B")
    obs = [sp.Observable(gl.Z(), s) for s in range(L)] + [sp.Observable(gl.X(), 3)]
    noise = NoiseModel([{"name": "pauli_x", "sites": [i], "strength": 0.05} for i in range(L)] + [{"name": "lowering", "sites": [6], "strength": 0.2}])
    cc = dtm._CompiledCircuit(tuple(layer() for _ in range(2)), 0)
    for mode in ("tdvp", "full-tdvp"):
        for chi in (4, None):
            p = sp.DigitalSimParams(observables=obs, max_bond_dim=chi, svd_threshold=1e-8, krylov_tol=1e-10, random_seed=11, gate_mode=mode)
            for name, nm, ntraj in ((f"{mode}_chi{chi}_noisy", noise, 4), (f"{mode}_chi{chi}_noiseless", None, 1)):
                res, diag = [], []
                for i in range(ntraj):
                    r, dg, _, _ = dtm.digital_tjm((i, st, nm, p, None), compiled_circuit=cc)
                    res.append(np.asarray(r, dtype=np.float64))
                    diag.append(dg)
                out[name + "_results"] = np.array(res)
                out[name + "_diag"] = np.array(diag)
    save("digital_tdvp", **out)


# ------------------------------------------------------------------ 19. mixed local dimensions (coupled transmon chain)
def gen_mixed_dims():
    """The reference on a chain whose sites differ in dimension: MPO.coupled_transmon (mpo.py:549-668; three-level transmons on the even
    sites, two-level resonators on the odd ones), Fock product state through MPS(physical_dimensions=[3, 2, 3, 2, ...]), loss on every
    site with that site's own ladder operator, occupation observables; one closed TDVP step from a random state and noisy
    trajectories of both drivers."""
    out = {}
    L, dq, dr, chi = 6, 3, 2, 8
    dims = [dq if i % 2 == 0 else dr for i in range(L)]
    H = MPO.coupled_transmon(L, dq, dr, 0.9, 0.7, -0.3, 0.25)
    out.update(pack_tensors("mpo", H.tensors))
    out["dims"] = np.array(dims)
    lower = {d_: np.diag(np.sqrt(np.arange(1, d_)), 1).astype(complex) for d_ in (dq, dr)}
    number = {d_: lower[d_].conj().T @ lower[d_] for d_ in (dq, dr)}
    rng = np.random.default_rng(17)
    caps = [1] * (L + 1)
    for i in range(1, L):
        caps[i] = min(int(np.prod(dims[:i])), int(np.prod(dims[i:])), chi)
    st = MPS(L, tensors=[rng.standard_normal((dims[i], caps[i], caps[i + 1])) + 1j * rng.standard_normal((dims[i], caps[i], caps[i + 1])) for i in range(L)],
             physical_dimensions=list(dims))
    st.normalize("B")
    out.update(pack_tensors("in", st.tensors))
    p = sp.AnalogSimParams(observables=[sp.Observable(number[dims[0]], 0)], elapsed_time=0.05, dt=0.05, max_bond_dim=chi, svd_threshold=1e-10, krylov_tol=1e-12)
    work = MPS(L, tensors=[t.copy() for t in st.tensors], physical_dimensions=list(dims))
    ref("core.methods.tdvp.tdvp").tdvp(work, H, p)
    out["tdvp_vec"] = work.to_vec()
    out["tdvp_bonds"] = np.array([t.shape[2] for t in work.tensors])
    basis = "".join(str(1 if dims[i] == 2 else 2 - (i // 2) % 2) for i in range(L))
    out["basis"] = np.array(basis)
    fock = MPS(L, physical_dimensions=list(dims), state="basis", basis_string=basis)
    noise = NoiseModel([{"name": "loss", "sites": [i], "strength": 0.25, "matrix": lower[dims[i]]} for i in range(L)])
    for order, fn in ((1, tjm.analog_tjm_1), (2, tjm.analog_tjm_2)):
        p = sp.AnalogSimParams(observables=[sp.Observable(number[dims[s]], s) for s in range(L)], elapsed_time=0.4, dt=0.1, num_traj=3, max_bond_dim=chi,
                               svd_threshold=1e-10, krylov_tol=1e-12, order=order, sample_timesteps=True, random_seed=6)
        rows, diags = [], []
        for i in range(3):
            r = fn((i, fock, noise, p, H))
            rows.append(np.asarray(r[0], dtype=np.float64))
            diags.append(np.asarray(r[1], dtype=np.float64))
        out[f"order{order}_results"] = np.array(rows)
        out[f"order{order}_diag"] = np.array(diags)
    # scheduled jumps (scheduled_jumps.py:51-119) with operators of the sites' own dimensions: one-site on a transmon, pair on (transmon, resonator)
    sched = [{"time": 0.1, "sites": [2], "name": "custom", "matrix": lower[dq]},
             {"time": 0.2, "sites": [2, 3], "name": "custom", "matrix": np.kron(number[dq] + 0.5 * lower[dq], lower[dr].conj().T + np.eye(dr))}]
    noise_s = NoiseModel([{"name": "loss", "sites": [i], "strength": 0.1, "matrix": lower[dims[i]]} for i in range(L)], scheduled_jumps=sched)
    p = sp.AnalogSimParams(observables=[sp.Observable(number[dims[s]], s) for s in range(L)], elapsed_time=0.4, dt=0.1, num_traj=3, max_bond_dim=chi,
                           svd_threshold=1e-10, krylov_tol=1e-12, order=1, sample_timesteps=True, random_seed=8)
    out["scheduled_results"] = np.array([np.asarray(tjm.analog_tjm_1((i, fock, noise_s, p, H))[0], dtype=np.float64) for i in range(3)])
    save("mixed_dims", **out)


# ------------------------------------------------------------------ 20. Fermi-Hubbard chain on composite four-level sites
def gen_fermi_hubbard():
    """MPO.fermi_hubbard_1d (mpo.py:409-520, fermionic ladder operators on dimension-4 sites, bond dimension 6): the MPO tensors and one
    closed two-site TDVP step from a seeded random state."""
    out = {}
    L, chi = 4, 8
    H = MPO.fermi_hubbard_1d(L, 1.0, 2.0)
    out.update(pack_tensors("mpo", H.tensors))
    rng = np.random.default_rng(23)
    caps = [1] * (L + 1)
    for i in range(1, L):
        caps[i] = min(4 ** i, 4 ** (L - i), chi)
    st = MPS(L, tensors=[rng.standard_normal((4, caps[i], caps[i + 1])) + 1j * rng.standard_normal((4, caps[i], caps[i + 1])) for i in range(L)],
             physical_dimensions=[4] * L)
    st.normalize("B")
    out.update(pack_tensors("in", st.tensors))
    n_up = np.kron(np.diag([0.0, 1.0]), np.eye(2))
    p = sp.AnalogSimParams(observables=[sp.Observable(n_up.astype(complex), 0)], elapsed_time=0.05, dt=0.05, max_bond_dim=chi, svd_threshold=1e-10, krylov_tol=1e-12)
    ref("core.methods.tdvp.tdvp").tdvp(st, H, p)
    out["tdvp_vec"] = st.to_vec()
    out["tdvp_bonds"] = np.array([t.shape[2] for t in st.tensors])
    save("fermi_hubbard", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["tiny", "rng", "truncate", "kernels", "tdvp", "noise", "traj", "digital", "shots", "scheduled", "piecewise"]
    for w in which:
        if w.startswith("fullsize:"):
            gen_fullsize(tuple(w.split(":")[1].split(",")))
        else:
            globals()["gen_" + w]()
