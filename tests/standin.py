"""Oracle-backed stand-in for ``yaqs_amd.engine.BatchEngine`` (TEST INFRASTRUCTURE ONLY).

The product has no CPU path; this class exists so that the *host logic* above the C ABI - the Python schedule of
``TrajectoryBatch``, ``Simulator.run`` (chunking, storage grown on demand, result assembly, validation) - runs in the CPU suite.
Every stage entry point is answered by the CPU oracle on one ``MPSState`` per slot; the uniforms the host hands over are consumed
the way the engine consumes them (one for the jump test, one more for the channel choice).
"""
from __future__ import annotations

import numpy as np

from oracle import tjm_oracle as o


class _ScriptedRng:
    """The two draws of ``stochastic_process`` from the host's uniforms: ``random()`` and ``choice(n, p)`` (searchsorted of the
    normalised cumulative sum, side="right": what ``Generator.choice`` does with one double)."""

    def __init__(self, values):
        self.values, self.used = list(values), 0

    def random(self):
        v = self.values[self.used]
        self.used += 1
        return v

    def choice(self, n, p=None):
        cdf = np.cumsum(np.asarray(p, dtype=np.float64))
        cdf /= cdf[-1]
        return int(np.searchsorted(cdf, self.random(), side="right"))


class OracleEngine:
    instances: list = []

    # True: non-gauge-invariant lines of the reference are followed to the letter (step_qr_bond), so that a host schedule run on
    # this stand-in can be compared with the REFERENCE's outputs; False: what the HIP engine computes (see tjm_engine.hip).
    reference_quirks = True

    def __init__(self, length, chi_max, batch, mpo, device="cpu", d=2, stream=None, cap_slack=1, dtype="complex128"):
        self.dtype = "complex128"  # the oracle computes in the reference's precision whatever is asked for
        self.L, self.d, self.chi_max, self.B = int(length), int(d), int(chi_max), int(batch)
        self.mpo = [np.asarray(w, dtype=np.complex128) for w in mpo]
        exact = o.MPSState.bond_caps(self.L, 1 << 30)
        self.caps = np.array([min(self.chi_max, int(cap_slack) * c) if 0 < k < self.L else 1 for k, c in enumerate(exact)], dtype=np.int32)
        self.sets = [[None] * self.B, [None] * self.B]
        self.noise = None
        self.filter = None
        self.params = None
        self.u = None
        self._overflow = False
        self.closed = False
        OracleEngine.instances.append(self)

    # -- configuration -----------------------------------------------------------------
    def set_params(self, *, dt, svd_threshold, trunc_mode="discarded_weight", max_bond_dim=None, krylov_tol=1e-4, tdvp_mode="2site",
                   tdvp_sweeps=1):
        self.params = o.Params(dt=dt, svd_threshold=svd_threshold, trunc_mode=trunc_mode, max_bond_dim=max_bond_dim, krylov_tol=krylov_tol,
                               tdvp_mode=tdvp_mode, tdvp_sweeps=tdvp_sweeps)

    def set_mpo(self, mpo):
        self.mpo = [np.asarray(w, dtype=np.complex128) for w in mpo]

    @property
    def mpo_tensors(self):
        return self.mpo

    def set_noise(self, processes, flags):
        self.noise = [dict(q) for q in processes]

    def set_noise_filter(self, indices=None):
        self.filter = None if indices is None else list(indices)

    def load_state(self, tensors, set_index=0):
        for b in range(self.B):
            self.sets[set_index][b] = o.MPSState([np.array(t, dtype=np.complex128) for t in tensors], 0)
        self._check()

    def copy_state(self, dst, src):
        self.sets[dst] = [s.copy() for s in self.sets[src]]

    def export_state(self, b, set_index=0):
        return [t.copy() for t in self.sets[set_index][b].tensors]

    def adopt(self, src, first=0):
        for b in range(self.B):
            self.sets[0][b] = src.sets[0][first + b].copy()

    def set_uniforms(self, u):
        self.u = np.asarray(u, dtype=np.float64)

    def close(self):
        self.closed = True

    def synchronize(self):
        pass

    # -- storage capacity ----------------------------------------------------------------
    def _check(self):
        for s in self.sets[0] + self.sets[1]:
            if s is not None and any(t.shape[2] > self.caps[i + 1] for i, t in enumerate(s.tensors)):
                self._overflow = True

    def capacity_overflow(self, clear=False):
        f = self._overflow
        if clear:
            self._overflow = False
        return f

    # -- the path ------------------------------------------------------------------------
    def _active_noise(self):
        if self.noise is None:
            return None
        if self.filter is None:
            return self.noise
        return [self.noise[k] for k in self.filter]

    def tdvp(self, set_index=0):
        for s in self.sets[set_index]:
            o.tdvp(s, self.mpo, self.params)
        self._check()

    def dissipate(self, dt, set_index=0):
        for s in self.sets[set_index]:
            o.apply_dissipation(s, self._active_noise() or None, dt, self.params)
        self._check()

    def stochastic(self, dt, set_index=0):
        jumped = np.zeros(self.B, dtype=np.int32)
        dp = np.zeros(self.B)
        noise = self._active_noise()
        for b in range(self.B):
            s = self.sets[set_index][b]
            dp[b] = 1.0 - s.norm_sq(0)
            rng = _ScriptedRng(self.u[b])
            self.sets[set_index][b] = o.stochastic_process(s, noise if noise else None, dt, self.params, rng)
            jumped[b] = 1 if rng.used > 1 else 0
        self._check()
        return jumped, dp

    def site0_normsq(self, set_index=0):
        return np.array([s.norm_sq(0) for s in self.sets[set_index]])

    def canonicalize_qr(self, center, set_index=0):
        for s in self.sets[set_index]:
            s.set_canonical_form(0, "QR")
            s.center = 0

    def normalize_qr(self, center, set_index=0):
        for s in self.sets[set_index]:
            s.normalize("B", "QR")

    def apply_single(self, site, matrix, set_index=0):
        m = np.asarray(matrix, dtype=np.complex128)
        for s in self.sets[set_index]:
            s.tensors[site] = np.einsum("ab,bcd->acd", m, s.tensors[site])
            s.center = None

    def apply_gate_mpo(self, first, last, left_ops, right_ops, set_index=0):
        """The site-by-site product of mpo.py:1511-1548 for U = sum_k left_ops[k] (x) right_ops[k] on (first, last)."""
        lo, ro = np.asarray(left_ops, dtype=np.complex128), np.asarray(right_ops, dtype=np.complex128)
        r = lo.shape[0]
        for s in self.sets[set_index]:
            for site in range(first, last + 1):
                a = s.tensors[site]
                d = a.shape[0]
                if site == first:
                    w = lo.transpose(1, 2, 0).reshape(d, d, 1, r)
                elif site == last:
                    w = ro.transpose(1, 2, 0).reshape(d, d, r, 1)
                else:
                    w = np.zeros((d, d, r, r), dtype=np.complex128)
                    for k in range(r):
                        w[:, :, k, k] = np.eye(d)
                th = np.tensordot(w, a, axes=([1], [0]))
                po, wl, wr, ml, mr = th.shape
                s.tensors[site] = th.transpose(0, 3, 1, 4, 2).reshape(po, ml * wl, mr * wr)
                if s.tensors[site].shape[2] > self.caps[site + 1]:
                    self._overflow = True
            s.center = None

    def apply_pair(self, left, matrix, min_keep=1, set_index=0):
        m = np.asarray(matrix, dtype=np.complex128).reshape(4, 4)
        p = self.params
        for s in self.sets[set_index]:
            a, b = s.tensors[left], s.tensors[left + 1]
            th = np.einsum("ab,bcd->acd", m, o.merge_two_site(a, b))
            s.tensors[left], s.tensors[left + 1] = o.split_two_site(th, [2, 2], svd_distribution="right", trunc_mode=p.trunc_mode,
                                                                    threshold=p.svd_threshold, max_bond_dim=p.max_bond_dim, min_keep=min_keep)
            s.center = None
        self._check()

    # -- site-level steps (tjm_engine_step_*) --------------------------------------------
    def _slots(self, ids):
        return range(self.B) if ids is None else [int(b) for b in ids]

    def step_env_init(self, set_index=0):
        self.env = {}
        for b, s in enumerate(self.sets[set_index]):
            rb = o.right_environments(s.tensors, self.mpo)
            lb = [None] * self.L
            lb[0] = o.identity_env(s.tensors[0].shape[1], self.mpo[0].shape[2])
            self.env[b] = (lb, rb)

    def step_two_site(self, site, dt, dist, capped, ids=None, set_index=0):
        p = self.params
        for b in self._slots(ids):
            t = self.sets[set_index][b].tensors
            lb, rb = self.env[b]
            theta = o.update_site(lb[site], rb[site + 1], o.merge_mpo_tensors(self.mpo[site], self.mpo[site + 1]), o.merge_two_site(t[site], t[site + 1]),
                                  dt, p.krylov_tol)
            t[site], t[site + 1] = o.split_two_site(theta, [2, 2], svd_distribution=dist, trunc_mode=p.trunc_mode, threshold=p.svd_threshold,
                                                    max_bond_dim=p.max_bond_dim if capped else None, min_keep=o._min_keep(p))
        self._check()

    def step_one_site(self, site, dt, ids=None, set_index=0):
        for b in self._slots(ids):
            t = self.sets[set_index][b].tensors
            lb, rb = self.env[b]
            t[site] = o.update_site(lb[site], rb[site], self.mpo[site], t[site], dt, self.params.krylov_tol)

    def step_env(self, site, left, ids=None, set_index=0):
        for b in self._slots(ids):
            t = self.sets[set_index][b].tensors
            lb, rb = self.env[b]
            if left:
                lb[site + 1] = o.update_left_environment(t[site], t[site], self.mpo[site], lb[site])
            else:
                rb[site - 1] = o.update_right_environment(t[site], t[site], self.mpo[site], rb[site])

    def step_qr_bond(self, site, right, dt, ids=None, set_index=0, max_bond_dim=None):
        tol = self.params.krylov_tol
        cap = max_bond_dim
        for b in self._slots(ids):
            t = self.sets[set_index][b].tensors
            lb, rb = self.env[b]
            if right:
                q, c = o.right_qr(t[site])
                if cap is not None and q.shape[2] > cap:  # integrators.py:361-364
                    q, c = q[:, :, :cap], c[:cap, :]
                t[site] = q
                lb[site + 1] = o.update_left_environment(q, q, self.mpo[site], lb[site])
                c = o.update_bond(lb[site + 1], rb[site], c, dt, tol)
                t[site + 1] = np.einsum("adc,bd->abc", t[site + 1], c)
            else:
                q, c = o.left_qr(t[site])
                if cap is not None and q.shape[1] > cap:
                    # integrators.py:452-455 cuts "bond_tensor[:cap, :]", the LEFT index of left_qr's R^T = C[left][new]; the
                    # gauge-invariant step cuts the new index
                    q, c = q[:, :cap, :], (c[:cap, :] if self.reference_quirks else c[:, :cap])
                t[site] = q
                rb[site - 1] = o.update_right_environment(q, q, self.mpo[site], rb[site])
                # left_qr hands back R^T = C[left][new]; the reference's dynamic sweep transposes it once more (integrators.py:461).
                # reference_quirks = True does the same, so that the host schedule can be checked against the reference's outputs.
                c = o.update_bond(lb[site], rb[site - 1], c.transpose() if self.reference_quirks else c, dt, tol)
                t[site - 1] = np.einsum("abd,dc->abc", t[site - 1], c)

    def step_cap_bond(self, bond, target, ids=None, set_index=0):
        for b in self._slots(ids):
            o._sync_bond_dim(self.sets[set_index][b], bond, target, self.params)

    # -- steps of the BUG integrator (tjm_engine_step_bug_*, _flip, _compress) ------------
    def step_bug_prepare(self, set_index=0):
        """prepare_canonical_site_tensors (bug.py:35-62) for every slot; the basis-change matrix starts as the identity."""
        self.bug = {}
        for b, s in enumerate(self.sets[set_index]):
            t = s.tensors
            canon = list(t)
            lenv = [np.eye(t[0].shape[1], dtype=np.complex128).reshape(t[0].shape[1], 1, t[0].shape[1])]
            for i in range(1, self.L):
                q, r = o.right_qr(canon[i - 1])
                canon[i] = np.tensordot(r, canon[i], axes=(1, 1)).transpose(1, 0, 2)
                lenv.append(o.update_left_environment(q, q, self.mpo[i - 1], lenv[i - 1]))
            rdim = t[-1].shape[2]
            self.bug[b] = dict(canon=canon, lenv=lenv, rblock=np.eye(rdim, dtype=np.complex128).reshape(rdim, 1, rdim),
                               m=np.eye(rdim, dtype=np.complex128))

    def step_bug_site(self, site, dt, set_index=0):
        """_local_update (bug.py:93-125): predictor, stacked trial basis, basis-change matrix, transported centre, right block."""
        tol = self.params.krylov_tol
        for b, s in enumerate(self.sets[set_index]):
            t, w = s.tensors, self.bug[b]
            working = w["canon"][site]
            predictor = o.update_site(w["lenv"][site], w["rblock"], self.mpo[site], working, dt, tol)
            old_current = np.tensordot(t[site], w["m"], axes=(2, 0))
            retained = t[site] if site == self.L - 1 else working
            new_q, _ = o.left_qr(np.concatenate((retained, predictor), axis=1))
            w["m"] = np.tensordot(old_current, new_q.conj(), axes=([0, 2], [0, 2]))
            t[site] = new_q
            w["canon"][site - 1] = np.tensordot(w["canon"][site - 1], w["m"], axes=(2, 0))
            w["rblock"] = o.update_right_environment(new_q, new_q, self.mpo[site], w["rblock"])
        self._check()

    def step_bug_root(self, dt, set_index=0):
        for b, s in enumerate(self.sets[set_index]):
            w = self.bug[b]
            s.tensors[0] = o.update_site(w["lenv"][0], w["rblock"], self.mpo[0], w["canon"][0], dt, self.params.krylov_tol)
            s.center = 0

    def sweep_dynamic(self, max_bond_dim, dt, set_index=0):
        """The library's one-call sweep (tjm_engine_sweep_dynamic): here the Python mirror over this engine's site-level steps, so that
        the CPU suite keeps checking that sequencing against the reference."""
        from yaqs_amd.tjm import _sweep_dynamic

        _sweep_dynamic(self, set_index, max_bond_dim, dt)

    def bug_sweep(self, dt, set_index=0):
        self.step_bug_prepare(set_index)
        for site in range(self.L - 1, 0, -1):
            self.step_bug_site(site, dt, set_index)
        self.step_bug_root(dt, set_index)

    def step_flip(self, set_index=0):
        for s in self.sets[set_index]:
            s.flip()

    def step_compress(self, threshold, max_bond_dim, trunc_mode, set_index=0):
        for s in self.sets[set_index]:
            o.compress(s, threshold, max_bond_dim, trunc_mode)
        self._check()

    # -- measurement ---------------------------------------------------------------------
    def site_moments(self, set_index=0):
        d = self.d
        M = np.zeros((self.L, self.B, d, d), dtype=np.complex128)
        for b, s in enumerate(self.sets[set_index]):
            for i in range(self.L):
                for p in range(d):
                    for q in range(d):
                        unit = np.zeros((d, d), dtype=np.complex128)
                        unit[p, q] = 1.0
                        M[i, b, p, q] = s.full_expect(unit, [i])
        return M

    def site_moments2(self, set_index=0):
        M = self.site_moments(set_index)
        d2 = self.d * self.d
        M2 = np.zeros((self.L - 1, self.B, d2, d2), dtype=np.complex128)
        for b, s in enumerate(self.sets[set_index]):
            for i in range(self.L - 1):
                for p in range(d2):
                    for q in range(d2):
                        unit = np.zeros((d2, d2), dtype=np.complex128)
                        unit[p, q] = 1.0
                        M2[i, b, p, q] = s.full_expect(unit, [i, i + 1])
        return M, M2

    def bond_dims(self, set_index=0):
        chi = np.ones((self.B, self.L + 1), dtype=np.int32)
        for b, s in enumerate(self.sets[set_index]):
            chi[b, 0] = s.tensors[0].shape[1]
            chi[b, 1:] = [t.shape[2] for t in s.tensors]
        return chi

    def bond_spectrum(self, site, set_index=0):
        n = int(self.d * min(self.caps[site], self.caps[site + 2]))
        out = np.zeros((self.B, n))
        for b, s in enumerate(self.sets[set_index]):
            sv = o.bond_singular_values(s, site)
            out[b, : min(n, len(sv))] = sv[:n]
        return out

    def bitstring_probability(self, bitstring, set_index=0):
        return np.array([o.project_onto_bitstring(s, bitstring) for s in self.sets[set_index]])

    def stats(self):
        return {}
