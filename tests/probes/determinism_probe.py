"""Run-to-run determinism probe: the same batch of Pauli-noise trajectories (certified dissipation path, chi = 32) several times in
one process; prints which trajectories / columns differ between repetitions.  Usage: python tests/probes/determinism_probe.py [reps]"""
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import torch  # noqa: F401,E402

from oracle import tjm_oracle as o  # noqa: E402
from yaqs_amd.api import AnalogSimParams, MPS, NoiseModel, Observable, Z as Zg  # noqa: E402
from yaqs_amd.engine import BatchEngine  # noqa: E402
from yaqs_amd.tjm import TrajectoryBatch  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
L, chi = 12, 32
st = o.MPSState.haar(L, chi, np.random.default_rng(7))
st.normalize("B")
init = [t.copy() for t in st.tensors]
noise = NoiseModel([{"name": "pauli_z", "sites": [i], "strength": 0.1} for i in range(L)])
mpo = o.ising_mpo(L, 1.0, 0.5)
p = AnalogSimParams(observables=[Observable(Zg(), s) for s in range(L)], num_traj=6, elapsed_time=0.8, dt=0.1, max_bond_dim=chi,
                    svd_threshold=1e-12, krylov_tol=1e-10, order=1, sample_timesteps=True, random_seed=3)
ref = None
for k in range(reps):
    e = BatchEngine(L, chi, 6, mpo)
    r, d = TrajectoryBatch(e, p, noise).run(list(range(6)), MPS(L, tensors=init), native=False)
    e.close()
    r = np.asarray(r)
    if ref is None:
        ref = r
        continue
    diff = np.abs(r - ref)
    if diff.max() > 0:
        t, s, c = np.unravel_index(np.argmax(diff), diff.shape)
        first = np.argwhere(diff > 0)
        print(f"rep {k}: max diff {diff.max():.3e} at traj {t} site {s} column {c}; first differing column {first[:, 2].min()}, trajectories {sorted(set(first[:, 0]))}")
    else:
        print(f"rep {k}: identical")
