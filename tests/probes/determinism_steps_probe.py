"""Where does a multi-step run first differ between repetitions?  Six Pauli-noise trajectories (L = 12, chi = 32), eight steps with the
trajectories' own random streams; after every stage of every step the states (and the site moments) are compared bit for bit with the
first repetition.  Usage: python tests/probes/determinism_steps_probe.py [reps]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: F401,E402

from oracle import tjm_oracle as o  # noqa: E402
from yaqs_amd.api import NoiseModel, is_pauli  # noqa: E402
from yaqs_amd.engine import BatchEngine  # noqa: E402
from yaqs_amd.tjm import trajectory_uniforms  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
L, chi, B, steps = 12, 32, 6, 8
st = o.MPSState.haar(L, chi, np.random.default_rng(7))
st.normalize("B")
init = [t.copy() for t in st.tensors]
mpo = o.ising_mpo(L, 1.0, 0.5)
noise = NoiseModel([{"name": "pauli_z", "sites": [i], "strength": 0.1} for i in range(L)])
u = np.stack([trajectory_uniforms(3, t, 2 * steps + 4) for t in range(B)])
ref = {}
found = 0
for k in range(reps):
    e = BatchEngine(L, chi, B, mpo)
    e.set_params(dt=0.1, svd_threshold=1e-12, max_bond_dim=chi, krylov_tol=1e-10, tdvp_mode="2site")
    e.set_noise(noise.processes, [is_pauli(q) for q in noise.processes])
    e.load_state(init)
    pos = np.zeros(B, dtype=np.int64)
    ar = np.arange(B)
    first = None
    for s_ in range(steps):
        for stage in ("tdvp", "dissipate", "stochastic", "moments"):
            if stage == "tdvp":
                e.tdvp()
            elif stage == "dissipate":
                e.dissipate(0.1)
            elif stage == "stochastic":
                e.set_uniforms(np.stack([u[ar, pos], u[ar, pos + 1]], axis=1))
                jumped, _ = e.stochastic(0.1)
                pos += 1 + jumped
            if stage == "moments":
                cur = [np.asarray(e.site_moments()).ravel()]
            else:
                cur = [np.concatenate([t.ravel() for t in e.export_state(b)]) for b in range(B)]
            key = (s_, stage)
            if key not in ref:
                ref[key] = cur
            elif first is None:
                d = [float(np.abs(c - r).max()) for c, r in zip(cur, ref[key])]
                if max(d) > 0:
                    first = (s_, stage, max(d), [b for b, x in enumerate(d) if x > 0])
    e.close()
    if k > 0:
        if first:
            found += 1
            print(f"repetition {k}: first difference at step {first[0]} after {first[1]}: {first[2]:.2e} (slots {first[3]})")
print(f"{found} of {reps - 1} repetitions differ from the first")
