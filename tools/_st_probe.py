import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import yaqs_amd.tjm as tjm
from yaqs_amd.api import AnalogSimParams, MPO, MPS, NoiseModel, Observable, Z
L, ntraj = 30, 16384
noise = NoiseModel([{"name": "pauli_z", "sites": [i], "strength": 0.1} for i in range(L)])
for st in (False, True, False, True):
    p = AnalogSimParams(observables=[Observable(Z(), s) for s in range(L)], elapsed_time=1.0, dt=0.1, num_traj=ntraj, random_seed=3, sample_timesteps=st)
    t0 = time.perf_counter()
    res = tjm.Simulator().run(MPS(L, state="x+"), MPO.ising(L, 1.0, 0.5), p, noise)
    dt = time.perf_counter() - t0
    print("sample_timesteps", st, round(dt, 3), "s", round(ntraj / dt, 1), "traj/s", flush=True)
