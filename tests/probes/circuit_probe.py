"""Config-5-like circuit run: 20 Trotter layers of the 64-site Ising circuit, depolarising noise (pauli_x/y/z,
gamma = 0.001 each on every site, applied after each two-qubit gate on that gate's sites), max_bond_dim = 512, svd_threshold 1e-9,
fp64.  With a 4th argument N the first N trajectories also run through the oracle on one host core each (timed, compared).
Usage: python tests/probes/circuit_probe.py [L] [num_traj] [layers] [check] [complex128|complex64]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402,F401  (loaded before the timed region)

import yaqs_amd.tjm as tjm  # noqa: E402
from yaqs_amd.api import DigitalSimParams, MPS, NoiseModel, Observable, Z, ising_trotter_layers  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ntraj = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
nlayers = int(sys.argv[3]) if len(sys.argv) > 3 else 20
check = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dtype = sys.argv[5] if len(sys.argv) > 5 else "complex128"
cpu = None
if check:  # before torch / HIP are loaded
    from oracle import tjm_oracle as o

    olayers = [o.GateLayer(l.singles, l.even, l.odd, l.sample_points) for l in __import__("yaqs_amd.api", fromlist=["x"]).ising_trotter_layers(L, 1.0, 0.5, 0.1, nlayers)]
    on = [o.make_process(name, [i], 0.001) for i in range(L) for name in ("pauli_x", "pauli_y", "pauli_z")]
    op = o.DigitalParams(observables=[o.Obs(o.PAULI["z"], s) for s in range(L)], max_bond_dim=512, svd_threshold=1e-9, random_seed=42)
    t0 = time.perf_counter()
    cpu = [o.digital_tjm(t, o.MPSState.product(L, "zeros"), on, op, olayers) for t in range(check)]
    cpu_s = (time.perf_counter() - t0) / check
layers = ising_trotter_layers(L, 1.0, 0.5, 0.1, nlayers)
noise = NoiseModel([{"name": name, "sites": [i], "strength": 0.001} for i in range(L) for name in ("pauli_x", "pauli_y", "pauli_z")])
p = DigitalSimParams(observables=[Observable(Z(), s) for s in range(L)], num_traj=ntraj, max_bond_dim=512, svd_threshold=1e-9, random_seed=42)
built = []
orig = tjm.BatchEngine


class Rec(orig):
    def __init__(self, length, chi_max, batch, mpo, **kw):
        built.append((int(chi_max), int(batch)))
        super().__init__(length, chi_max, batch, mpo, **kw)


tjm.BatchEngine = Rec
t0 = time.perf_counter()
res = tjm.Simulator(dtype=dtype).run_circuit(MPS(L, state="zeros"), layers, p, noise)
dt = time.perf_counter() - t0
gates = sum(len(l.even) + len(l.odd) for l in layers)
extra = {}
if cpu is not None:
    err = max(float(np.max(np.abs(res.trajectories[u][t] - cpu[t][0][u]))) for t in range(check) for u in range(L))
    extra["oracle"] = {"trajectories": check, "seconds_per_trajectory_one_core": round(cpu_s, 2), "max_abs_difference": err}
print(json.dumps({"workload": f"{L}-site Ising Trotter circuit, {nlayers} layers ({gates} two-qubit gates), depolarising gamma=0.001, max_bond_dim=512, "
                              f"svd_threshold=1e-9, {dtype}", "trajectories": ntraj, "engines (capacity, batch)": built, "seconds": round(dt, 2),
                  "trajectories_per_sec": round(ntraj / dt, 2), "gate_updates_per_sec": round(ntraj * gates / dt, 1),
                  "max_bond": int(np.max(res.max_bond)), "mean_Z_site0": float(res.expectation_values[0][-1]), **extra}))
