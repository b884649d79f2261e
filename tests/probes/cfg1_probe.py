"""Config 1: 10-site closed TFIM from "zeros", order 2, two-site TDVP, chi = 16, dt = 0.1, T = 1, svd_threshold 1e-9,
krylov_tol 1e-12, Z on every site at every time: one trajectory on the GPU (latency-bound) next to the oracle on one host core."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import tjm_oracle as o  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 10
T = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
kw = dict(elapsed_time=T, dt=0.1, max_bond_dim=16, svd_threshold=1e-9, krylov_tol=1e-12, order=2, sample_timesteps=True, random_seed=42)
op = o.Params(observables=[o.Obs(o.PAULI["z"], s) for s in range(L)], **kw)
t0 = time.perf_counter()
ref = o.run_trajectory(0, o.MPSState.product(L, "zeros"), [], op, o.ising_mpo(L, 1.0, 0.5))
cpu_s = time.perf_counter() - t0

from yaqs_amd.api import AnalogSimParams, MPO, MPS, Observable, Z  # noqa: E402
from yaqs_amd.tjm import Simulator  # noqa: E402

p = AnalogSimParams(observables=[Observable(Z(), s) for s in range(L)], **kw)
sim = Simulator()
sim.run(MPS(L, state="zeros"), MPO.ising(L, 1.0, 0.5), p)  # warm-up (library load, first launches)
t0 = time.perf_counter()
res = sim.run(MPS(L, state="zeros"), MPO.ising(L, 1.0, 0.5), p)
gpu_s = time.perf_counter() - t0
err = max(float(np.max(np.abs(res.expectation_values[u] - ref[0][op.observable_sorted_indices[u]]))) for u in range(L))
print(json.dumps({"workload": f"config 1: {L}-site closed TFIM, order 2, chi=16, T={T:g}, dt=0.1, sampling at every step", "gpu_seconds_one_trajectory": round(gpu_s, 3),
                  "oracle_seconds_one_core": round(cpu_s, 3), "max_abs_difference": err, "max_bond": int(np.max(res.max_bond))}))
