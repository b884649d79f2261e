"""Do two engines on two HIP streams overlap (VALU-bound Jacobi next to MFMA-bound GEMMs)?  Not part of the product.

    python tools/overlap_probe.py B_total steps
"""
import os, sys, threading, time
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


def make(b, stream=None):
    e = BatchEngine(L, chi, b, mpo.tensors, stream=stream)
    e.set_params(dt=0.1, svd_threshold=1e-12, max_bond_dim=chi, krylov_tol=1e-4)
    e.set_noise(nm.processes, [is_pauli(p) for p in nm.processes])
    e.load_state(st.tensors)
    return e


def drive(e, n, seed, delay_steps=0.0):
    rng = np.random.default_rng(seed)
    for _ in range(n):
        e.tdvp(); e.dissipate(0.1)
        e.set_uniforms(rng.random((e.B, 2)))
        e.stochastic(0.1)


def timed(engines, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=drive, args=(e, n, k)) for k, e in enumerate(engines)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for n_eng in (1, 2, 3, 4):
    if B % n_eng: continue
    engines = [make(B // n_eng, torch.cuda.Stream()) for _ in range(n_eng)]
    timed(engines, 1)
    dt = timed(engines, steps)
    print(f"{n_eng} engine(s) x {B // n_eng}: {dt:.3f} s/step -> {B / 10 / dt:.3f} traj/s", flush=True)
    for e in engines: e.close()
    del engines
    torch.cuda.empty_cache()
