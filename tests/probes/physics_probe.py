"""Config-2 physics run: 64-site dissipative TFIM from the product state "x+", pauli_z gamma = 0.1 on every
site, max_bond_dim 128, dt = 0.1, 10 steps, svd_threshold 1e-12, krylov_tol 1e-4, order 1, final-time sampling - bonds grow from 1,
so this exercises the capacity-on-demand path end to end.  With --check N the first N trajectories are also run through the
oracle on the host (single core each, timed) and compared.
Usage: python tests/probes/physics_probe.py [num_traj] [--check N] [--threshold 1e-12] [--steps 10]"""
import argparse
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")

ap = argparse.ArgumentParser()
ap.add_argument("num_traj", nargs="?", type=int, default=1024)
ap.add_argument("--check", type=int, default=0)
ap.add_argument("--threshold", type=float, default=1e-12)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--length", type=int, default=64)
ap.add_argument("--engines", type=int, default=4)
args = ap.parse_args()
L, dt = args.length, 0.1

cpu = None
if args.check:  # before torch / HIP are loaded
    from oracle import tjm_oracle as o

    op = o.Params(observables=[o.Obs(o.PAULI["z"], s) for s in range(L)], elapsed_time=args.steps * dt, dt=dt, max_bond_dim=128, svd_threshold=args.threshold,
                  krylov_tol=1e-4, order=1, sample_timesteps=False, random_seed=42)
    on = [o.make_process("pauli_z", [i], 0.1) for i in range(L)]
    t0 = time.perf_counter()
    cpu = [o.run_trajectory(t, o.MPSState.product(L, "x+"), on, op, o.ising_mpo(L, 1.0, 0.5)) for t in range(args.check)]
    cpu_s = (time.perf_counter() - t0) / args.check

import torch  # noqa: E402,F401  (loaded before the timed region)

import yaqs_amd.tjm as tjm  # noqa: E402
from yaqs_amd.api import AnalogSimParams, MPO, MPS, NoiseModel, Observable, Z  # noqa: E402

p = AnalogSimParams(observables=[Observable(Z(), s) for s in range(L)], elapsed_time=args.steps * dt, dt=dt, num_traj=args.num_traj, max_bond_dim=128,
                    svd_threshold=args.threshold, krylov_tol=1e-4, order=1, random_seed=42, sample_timesteps=False)
noise = NoiseModel([{"name": "pauli_z", "sites": [i], "strength": 0.1} for i in range(L)])
built = []
orig = tjm.BatchEngine


class Rec(orig):
    def __init__(self, length, chi_max, batch, mpo, **kw):
        built.append((int(chi_max), int(batch)))
        super().__init__(length, chi_max, batch, mpo, **kw)


tjm.BatchEngine = Rec
t0 = time.perf_counter()
res = tjm.Simulator(engines=args.engines).run(MPS(L, state="x+"), MPO.ising(L, 1.0, 0.5), p, noise)
sec = time.perf_counter() - t0
out = {"workload": f"{L}-site dissipative TFIM from x+, pauli_z gamma=0.1, max_bond_dim=128, dt=0.1, {args.steps} steps, svd_threshold={args.threshold:g}, "
                   "krylov_tol=1e-4, order 1", "trajectories": args.num_traj, "engines (capacity, batch)": built, "seconds": round(sec, 2),
       "trajectories_per_sec": round(args.num_traj / sec, 3), "max_bond": int(np.max(res.max_bond)),
       "mean_Z_site0": float(res.expectation_values[0][-1])}
if cpu is not None:
    idx = op.observable_sorted_indices
    err = max(float(np.max(np.abs(res.trajectories[u][t] - cpu[t][0][idx[u]]))) for t in range(args.check) for u in range(L))
    out["oracle"] = {"trajectories": args.check, "seconds_per_trajectory_one_core": round(cpu_s, 2), "max_abs_difference": err}
print(json.dumps(out))
