"""How many fp64 one-sided Jacobi sweeps does a two-site split need when it starts from theta times an approximate right-singular
basis computed in complex64 (re-orthonormalised in fp64 by Newton-Schulz steps), instead of from the doubly QR-preconditioned
matrix?  CPU experiment on the oracle (no GPU), same harness as warm_start_probe.py: a dissipative TFIM chain at a saturated bond
dimension, order-1 TJM steps; every full-size backward-sweep split is factorised by a plain cyclic one-sided Jacobi in NumPy

  cold   columns sorted by norm, QR, QR of R^H (the preconditioning of tjm_svd.hip), tolerance 1e-13
  mixed  V32 = right singular basis of complex64(theta) (LAPACK in single precision stands in for the complex64 Jacobi), k Newton-Schulz
         steps V <- V (3 I - V^H V) / 2 in fp64, Jacobi on theta V

and the rotations of every sweep are counted (a sweep that rotates nothing ends the iteration).

    python tests/probes/mixed_precision_probe.py [L=10] [chi=16] [steps=3] [ns=2]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import tjm_oracle as o  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 10
chi = int(sys.argv[2]) if len(sys.argv) > 2 else 16
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ns = int(sys.argv[4]) if len(sys.argv) > 4 else 2


def jacobi_rotations(x, tol=1e-13, max_sweeps=40):
    """cyclic one-sided Jacobi on the columns of x; returns the rotations of every sweep (the last entry is 0) and the rotated matrix"""
    x = x.copy()
    n = x.shape[1]
    counts = []
    for _ in range(max_sweeps):
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
        counts.append(rotated)
        if rotated == 0:
            break
    return counts, x


def precondition(z):
    order = np.argsort(-np.linalg.norm(z, axis=0))
    r = np.linalg.qr(z[:, order])[1]
    r1 = np.linalg.qr(r.conj().T)[1]
    return r1.conj().T


def mixed_start(mat, ns_steps):
    v = np.linalg.svd(mat.astype(np.complex64))[2].conj().T.astype(np.complex128)
    eye = np.eye(v.shape[1])
    resid0 = np.abs(v.conj().T @ v - eye).max()
    for _ in range(ns_steps):
        v = v @ (1.5 * eye - 0.5 * (v.conj().T @ v))
    return mat @ v, resid0, np.abs(v.conj().T @ v - eye).max()


rng = np.random.default_rng(7)
st = o.MPSState.haar(L, chi, rng)
st.normalize("B")
mpo = o.ising_mpo(L, 1.0, 0.5)
noise = [o.make_process("pauli_z", [i], 0.1) for i in range(L)]
params = o.Params(observables=[o.Obs(np.diag([1.0, -1.0]).astype(complex), 0)], elapsed_time=0.1 * steps, dt=0.1, max_bond_dim=chi, svd_threshold=1e-12,
                  krylov_tol=1e-4, order=1, sample_timesteps=False, random_seed=1)

log = []
orig_split = o._split_tdvp


def spying_split(theta, p, dist, dims=None):
    out = orig_split(theta, p, dist, dims)
    d0, d1 = dims if dims is not None else (2, 2)
    chiL, chiR = theta.shape[1], theta.shape[2]
    mat = theta.reshape(d0, d1, chiL, chiR).transpose(0, 2, 1, 3).reshape(d0 * chiL, d1 * chiR)
    if dist == "left" and min(mat.shape) >= 2 * chi:
        cold, _ = jacobi_rotations(precondition(mat))
        x1, r0, r1 = mixed_start(mat, ns)
        mixed, y = jacobi_rotations(x1)
        sv = np.sort(np.linalg.norm(y, axis=0))[::-1]
        ref = np.linalg.svd(mat, compute_uv=False)
        rel = np.max(np.abs(sv - ref) / np.maximum(ref, 1e-300))
        log.append((mat.shape, ref[0] / ref[-1], cold, mixed, r0, r1, rel))
        print(mat.shape, "cond %.1e" % (ref[0] / ref[-1]), "cold", cold, "mixed", mixed, "ortho before/after NS %.1e %.1e" % (r0, r1), "max rel err of sigma %.1e" % rel,
              flush=True)
    return out


o._split_tdvp = spying_split
state = st
prng = o.trajectory_rng(1, 0)
for k in range(steps):
    o.apply_dissipation(state, noise, params.dt, params)
    state = o.stochastic_process(state, noise, params.dt, params, prng)
    o.apply_unitary_evolution(state, mpo, params)
if log:
    print("mean sweeps: cold %.2f  mixed %.2f ; mean rotating sweeps: cold %.2f mixed %.2f" % (
        np.mean([len(r[2]) for r in log]), np.mean([len(r[3]) for r in log]),
        np.mean([sum(1 for c in r[2] if c) for r in log]), np.mean([sum(1 for c in r[3] if c) for r in log])))
