"""Which stage of a TJM step is not run-to-run deterministic at chi = 32?  One engine, the same loaded state; every stage is applied
`reps` times from the same input (state set 1 keeps the input) and the exported state of every trajectory is compared bit for bit
with the first repetition.  Usage: python tests/probes/determinism_stage_probe.py [reps] [chi]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: F401,E402

from oracle import tjm_oracle as o  # noqa: E402
from yaqs_amd.api import NoiseModel, is_pauli  # noqa: E402
from yaqs_amd.engine import BatchEngine  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
chi = int(sys.argv[2]) if len(sys.argv) > 2 else 32
L, B = 12, 6
st = o.MPSState.haar(L, chi, np.random.default_rng(7))
st.normalize("B")
init = [t.copy() for t in st.tensors]
mpo = o.ising_mpo(L, 1.0, 0.5)
noise = NoiseModel([{"name": "pauli_z", "sites": [i], "strength": 0.1} for i in range(L)])
e = BatchEngine(L, chi, B, mpo)
e.set_params(dt=0.1, svd_threshold=1e-12, max_bond_dim=chi, krylov_tol=1e-10, tdvp_mode="2site")
e.set_noise(noise.processes, [is_pauli(q) for q in noise.processes])


def snapshot():
    return [np.concatenate([t.ravel() for t in e.export_state(b)]) for b in range(B)]


def stage(name, fn, prepare):
    ref = None
    bad = 0
    for k in range(reps):
        prepare()
        fn()
        cur = snapshot()
        if ref is None:
            ref = cur
        else:
            d = max(float(np.abs(c - r).max()) for c, r in zip(cur, ref))
            if d > 0:
                bad += 1
                who = [b for b in range(B) if np.abs(cur[b] - ref[b]).max() > 0]
                print(f"  {name}: repetition {k} differs by {d:.2e} in trajectories {who}")
    print(f"{name}: {bad} of {reps - 1} repetitions differ")


# inputs of the stages: the Haar state (tdvp), the state after tdvp (dissipate), after dissipation (stochastic with a forced jump)
e.load_state(init)
stage("tdvp", lambda: e.tdvp(), lambda: e.load_state(init))
e.load_state(init)
e.tdvp()
after_tdvp = [e.export_state(b) for b in range(B)]


def load_slots(states):
    for b in range(B):
        e.load_state_slot(b, states[b])


stage("dissipate", lambda: e.dissipate(0.1), lambda: load_slots(after_tdvp))
load_slots(after_tdvp)
e.dissipate(0.1)
after_diss = [e.export_state(b) for b in range(B)]


def jump():
    e.set_uniforms(np.tile(np.array([[0.0, 0.37]]), (B, 1)))  # u = 0 < dp: every trajectory jumps
    e.stochastic(0.1)


stage("stochastic (all jump)", jump, lambda: load_slots(after_diss))
e.close()
