"""The TDVP sweep of step 8 of tests/probes/determinism_steps_probe.py, repeated from the SAME input states (the states after seven
steps, exported once and reloaded slot by slot): how many repetitions differ from the first, and in which slots.
Usage: python tests/probes/determinism_tdvp_probe.py [reps]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: F401,E402

from oracle import tjm_oracle as o  # noqa: E402
from yaqs_amd.api import NoiseModel, is_pauli  # noqa: E402
from yaqs_amd.engine import BatchEngine  # noqa: E402
from yaqs_amd.tjm import trajectory_uniforms  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
L, chi, B, steps = 12, 32, 6, 7
st = o.MPSState.haar(L, chi, np.random.default_rng(7))
st.normalize("B")
init = [t.copy() for t in st.tensors]
mpo = o.ising_mpo(L, 1.0, 0.5)
noise = NoiseModel([{"name": "pauli_z", "sites": [i], "strength": 0.1} for i in range(L)])
u = np.stack([trajectory_uniforms(3, t, 2 * steps + 6) for t in range(B)])
e = BatchEngine(L, chi, B, mpo)
e.set_params(dt=0.1, svd_threshold=1e-12, max_bond_dim=chi, krylov_tol=1e-10, tdvp_mode="2site")
e.set_noise(noise.processes, [is_pauli(q) for q in noise.processes])
e.load_state(init)
pos = np.zeros(B, dtype=np.int64)
ar = np.arange(B)
for s_ in range(steps):
    e.tdvp()
    e.dissipate(0.1)
    e.set_uniforms(np.stack([u[ar, pos], u[ar, pos + 1]], axis=1))
    jumped, _ = e.stochastic(0.1)
    pos += 1 + jumped
inputs = [e.export_state(b) for b in range(B)]
print("bonds of slot 4:", [t.shape[2] for t in inputs[4]])
ref, bad = None, 0
for k in range(reps):
    for b in range(B):
        e.load_state_slot(b, inputs[b])
    e.tdvp()
    cur = [np.concatenate([t.ravel() for t in e.export_state(b)]) for b in range(B)]
    if ref is None:
        ref = cur
    else:
        d = [float(np.abs(c - r).max()) for c, r in zip(cur, ref)]
        if max(d) > 0:
            bad += 1
            print(f"  repetition {k}: {max(d):.2e} in slots {[b for b, x in enumerate(d) if x > 0]}")
print(f"tdvp of step 8: {bad} of {reps - 1} repetitions differ")
e.close()
