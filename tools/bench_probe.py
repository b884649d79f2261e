"""Timing probe for the headline configuration (not part of the product)."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
from yaqs_amd.api import MPS, MPO, NoiseModel, is_pauli
from yaqs_amd.engine import BatchEngine

L = int(sys.argv[1]) if len(sys.argv) > 1 else 64
chi = int(sys.argv[2]) if len(sys.argv) > 2 else 128
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
tol = float(sys.argv[5]) if len(sys.argv) > 5 else 1e-4
mpo = MPO.ising(L, 1.0, 0.5)
st = MPS(L, state="haar-random", pad=chi, rng=np.random.default_rng(1))
st.normalize("B")
e = BatchEngine(L, chi, B, mpo.tensors)
print("workspace GB", e.workspace_bytes / 2**30, flush=True)
e.set_params(dt=0.1, svd_threshold=1e-12, max_bond_dim=chi, krylov_tol=tol)
nm = NoiseModel([{"name": "pauli_z", "sites": [i], "strength": 0.1} for i in range(L)])
e.set_noise(nm.processes, [is_pauli(p) for p in nm.processes])
e.load_state(st.tensors)
rng = np.random.default_rng(0)
for s in range(steps):
    s0 = e.stats()
    t0 = time.time(); e.tdvp(); e.synchronize(); t1 = time.time()
    s1 = e.stats()
    e.dissipate(0.1); e.synchronize(); t2 = time.time()
    s2 = e.stats()
    print("   tdvp svds", s1["svds"]-s0["svds"], "sweeps", s1["svd_sweeps"]-s0["svd_sweeps"], "| diss svds", s2["svds"]-s1["svds"], "sweeps", s2["svd_sweeps"]-s1["svd_sweeps"])
    e.set_uniforms(rng.random((B, 2)))
    j, dp = e.stochastic(0.1); e.synchronize(); t3 = time.time()
    print(f"step {s}: tdvp {t1-t0:.3f}s diss {t2-t1:.3f}s stoch {t3-t2:.3f}s  jumps {j.sum()}/{B} dp {dp[:3]}", e.stats(), flush=True)
chi_t = e.bond_dims()
print("bonds", chi_t[0][:10], chi_t[0][L//2-2:L//2+2])
