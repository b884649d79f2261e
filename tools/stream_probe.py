"""Null stream vs an explicit stream for one engine (not part of the product).   python tools/stream_probe.py B steps"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
from yaqs_amd.api import MPS, MPO, NoiseModel, is_pauli
from yaqs_amd.engine import BatchEngine

L, chi = 64, 128
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
mpo = MPO.ising(L, 1.0, 0.5)
st = MPS(L, state="haar-random", pad=chi, rng=np.random.default_rng(1))
st.normalize("B")
nm = NoiseModel([{"name": "pauli_z", "sites": [i], "strength": 0.1} for i in range(L)])
for label, mk in (("null stream", lambda: None), ("explicit stream", lambda: torch.cuda.Stream()), ("null stream", lambda: None)):
    e = BatchEngine(L, chi, B, mpo.tensors, stream=mk())
    e.set_params(dt=0.1, svd_threshold=1e-12, max_bond_dim=chi, krylov_tol=1e-4)
    e.set_noise(nm.processes, [is_pauli(p) for p in nm.processes])
    e.load_state(st.tensors)
    rng = np.random.default_rng(0)
    for k in range(steps + 1):
        if k == 1:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        e.tdvp(); e.dissipate(0.1)
        e.set_uniforms(rng.random((B, 2)))
        e.stochastic(0.1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{label}, B={B}: {dt:.3f} s/step -> {B / 10 / dt:.3f} traj/s", e.stats(), flush=True)
    e.close(); del e
    torch.cuda.empty_cache()
