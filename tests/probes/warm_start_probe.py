"""How many one-sided Jacobi sweeps would a two-site split need if it started from the right-singular basis of the SAME bond's split
one time step earlier, instead of from scratch (sorted columns + two QR factorisations, what the HIP path does)?  CPU experiment on
the oracle (no GPU): a dissipative TFIM chain at a saturated bond dimension, order-1 TJM steps; every backward-sweep split of step k
is factorised twice by a plain cyclic one-sided Jacobi in NumPy - cold (columns sorted by norm, QR, QR of R^H: the preconditioning
of tjm_svd.hip) and warm (theta times the previous step's V at that bond, then the same QR pair) - and the sweeps to convergence
(relative off-diagonal 1e-13) are counted.

    python tests/probes/warm_start_probe.py [L=10] [chi=16] [steps=4]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import tjm_oracle as o  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 10
chi = int(sys.argv[2]) if len(sys.argv) > 2 else 16
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4


def jacobi_sweeps(x, tol=1e-13, max_sweeps=40):
    """cyclic one-sided Jacobi on the columns of x; returns the number of sweeps until a sweep rotates nothing"""
    x = x.copy()
    n = x.shape[1]
    for sweep in range(max_sweeps):
        rotated = 0
        for p in range(n - 1):
            for q in range(p + 1, n):
                a = np.vdot(x[:, p], x[:, p]).real
                d = np.vdot(x[:, q], x[:, q]).real
                g = np.vdot(x[:, p], x[:, q])
                if abs(g) ** 2 <= tol * tol * a * d or a < 1e-26 or d < 1e-26:
                    continue
                rotated += 1
                delta = 0.5 * (d - a)
                r = np.hypot(delta, abs(g))
                u = abs(delta) + r
                qq = 1.0 / np.sqrt(2 * r * u)
                c = u * qq
                s = (qq if delta >= 0 else -qq) * g
                xp = c * x[:, p] - np.conj(s) * x[:, q]
                xq = s * x[:, p] + c * x[:, q]
                x[:, p], x[:, q] = xp, xq
        if rotated == 0:
            return sweep + 1, rotated
    return max_sweeps, rotated


def precondition(z):
    """columns sorted by norm, Z = Q R, R^H = Q1 R1, Jacobi runs on X = R1^H (tjm_svd.hip)"""
    order = np.argsort(-np.linalg.norm(z, axis=0))
    r = np.linalg.qr(z[:, order])[1]
    r1 = np.linalg.qr(r.conj().T)[1]
    return r1.conj().T


rng = np.random.default_rng(7)
st = o.MPSState.haar(L, chi, rng)
st.normalize("B")
mpo = o.ising_mpo(L, 1.0, 0.5)
noise = [o.make_process("pauli_z", [i], 0.1) for i in range(L)]
params = o.Params(observables=[o.Obs(np.diag([1.0, -1.0]).astype(complex), 0)], elapsed_time=0.1 * steps, dt=0.1, max_bond_dim=chi, svd_threshold=1e-12,
                  krylov_tol=1e-4, order=1, sample_timesteps=False, random_seed=1)

captured = {}  # bond -> V of the last split at that bond (columns = right singular vectors, full square)
log = []
orig_split = o._split_tdvp


def spying_split(theta, p, dist, dims=None):
    out = orig_split(theta, p, dist, dims)
    d0, d1 = dims if dims is not None else (2, 2)
    m = theta.shape  # (d0*d1, chiL, chiR)
    chiL, chiR = m[1], m[2]
    mat = theta.reshape(d0, d1, chiL, chiR).transpose(0, 2, 1, 3).reshape(d0 * chiL, d1 * chiR)
    if dist == "left" and min(mat.shape) >= 16:  # backward-sweep splits of full size only
        key = (spying_split.site, mat.shape)
        cold = jacobi_sweeps(precondition(mat))[0]
        warm = None
        if key in captured:
            warm = jacobi_sweeps(precondition(mat @ captured[key]))[0]
            plain = jacobi_sweeps(mat @ captured[key])[0]
        else:
            plain = None
        log.append((spying_split.step, spying_split.site, mat.shape, cold, warm, plain))
        captured[key] = np.linalg.svd(mat)[2].conj().T
    return out


spying_split.site = 0
spying_split.step = 0
o._split_tdvp = spying_split

# the backward sweep visits bonds L-2 ... 0: count them to label the site
state = st
prng = o.trajectory_rng(1, 0)
for k in range(steps):
    spying_split.step = k
    calls = {"n": 0}
    orig = spying_split

    def counted(theta, p, dist, dims=None, _orig=orig, _calls=calls):
        if dist == "left":
            spying_split.site = _calls["n"]
            _calls["n"] += 1
        return _orig(theta, p, dist, dims)

    o._split_tdvp = counted
    o.apply_dissipation(state, noise, params.dt, params)
    state = o.stochastic_process(state, noise, params.dt, params, prng)
    o.apply_unitary_evolution(state, mpo, params)
    o._split_tdvp = spying_split

print("step site shape cold warm(+QR pair) warm(plain)")
for row in log:
    print(*row)
later = [r for r in log if r[4] is not None]
if later:
    print("mean sweeps, splits that had a previous basis: cold %.2f  warm + QR pair %.2f  warm plain %.2f" % (
        np.mean([r[3] for r in later]), np.mean([r[4] for r in later]), np.mean([r[5] for r in later])))
