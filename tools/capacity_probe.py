"""Times Simulator.run on a low-entanglement run with the reference's default ("balanced") preset: static capacity = max_bond_dim
versus the capacity grown on demand.  Usage: python tools/capacity_probe.py [L] [num_traj] [batch|0] [mode]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402,F401  (loaded before the timed region)

import yaqs_amd.tjm as tjm  # noqa: E402
from yaqs_amd.api import AnalogSimParams, MPO, MPS, NoiseModel, Observable, Z  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ntraj = int(sys.argv[2]) if len(sys.argv) > 2 else 64
batch = (int(sys.argv[3]) or None) if len(sys.argv) > 3 else None  # None: sized to the free HBM
p = AnalogSimParams(observables=[Observable(Z(), s) for s in range(L)], elapsed_time=1.0, dt=0.1, num_traj=ntraj, random_seed=3, sample_timesteps=False)
noise = NoiseModel([{"name": "pauli_z", "sites": [i], "strength": 0.1} for i in range(L)])
out = {}
modes = (("on-demand", tjm.START_CHI), ("static", 4096))
if len(sys.argv) > 4:  # e.g. "on-demand": one mode only (profiling runs)
    modes = tuple(m for m in modes if m[0] == sys.argv[4])
for label, start in modes:
    tjm.START_CHI = start
    built = []
    orig = tjm.BatchEngine

    class Rec(orig):
        def __init__(self, length, chi_max, batch, mpo, **kw):
            built.append(chi_max)
            super().__init__(length, chi_max, batch, mpo, **kw)

    tjm.BatchEngine = Rec
    t0 = time.perf_counter()
    res = tjm.Simulator(batch=batch).run(MPS(L, state="x+"), MPO.ising(L, 1.0, 0.5), p, noise)
    dt = time.perf_counter() - t0
    tjm.BatchEngine = orig
    out[label] = np.array([e[0] for e in res.expectation_values])
    print(f"{label:10s} capacities {built}  max bond {int(np.max(res.max_bond))}  {dt:.2f} s  ({ntraj / dt:.2f} trajectories/s)", flush=True)
if len(out) == 2:
    print("max |difference| of the ensemble means:", float(np.max(np.abs(out["on-demand"] - out["static"]))))
