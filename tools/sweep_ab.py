"""Same-box A/B of the sweeps sequenced inside the library (tjm_engine_sweep_dynamic / tjm_engine_bug_sweep, round 6) against the
host-sequenced ones (yaqs_amd/tjm.py: _sweep_dynamic over the site-level entry points, one bond-table read-back per site):

    python tools/sweep_ab.py [L=16] [chi=16] [B=64] [sweeps=10]

A dissipation-free Ising chain from a Haar state of bond chi / 2 with max_bond_dim = chi, so that part of the trajectories' bonds sit
at the cap (both branches of the dynamic sweep run).  Prints ms per sweep and the engine's launch-independent counters.  Not part of
the product."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch

import yaqs_amd.tjm as T
from yaqs_amd import api
from yaqs_amd.engine import BatchEngine

L = int(sys.argv[1]) if len(sys.argv) > 1 else 16
chi = int(sys.argv[2]) if len(sys.argv) > 2 else 16
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
n = int(sys.argv[4]) if len(sys.argv) > 4 else 10
mpo = api.MPO.ising(L, 1.0, 0.5)
st = api.MPS(L, state="haar-random", pad=max(2, chi // 2), rng=np.random.default_rng(3))
st.normalize("B")
out = {}
for name in ("library", "host"):
    e = BatchEngine(L, chi, B, mpo.tensors)
    e.set_params(dt=0.05, svd_threshold=1e-9, max_bond_dim=chi, krylov_tol=1e-8, tdvp_mode="dynamic")
    e.load_state(st.tensors)
    run = (lambda: e.sweep_dynamic(chi, 0.05, 0)) if name == "library" else (lambda: T._sweep_dynamic(e, 0, chi, 0.05))
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        run()
    torch.cuda.synchronize()
    out[name] = 1e3 * (time.perf_counter() - t0) / n
    bonds = e.bond_dims()
    print(f"dynamic TDVP, {name}-sequenced: {out[name]:.1f} ms per sweep (L={L}, cap={chi}, B={B}; largest bond {bonds.max()}, trajectories at the cap somewhere: {(bonds.max(axis=1) >= chi).sum()})", flush=True)
    e.close()
print(f"dynamic sweep: library / host = {out['library'] / out['host']:.3f}")
for name in ("library", "host"):
    e = BatchEngine(L, chi, B, mpo.tensors, cap_slack=2)
    e.set_params(dt=0.05, svd_threshold=1e-9, max_bond_dim=chi, krylov_tol=1e-8)
    e.set_noise([], [])
    e.load_state(st.tensors)

    def half(e=e, name=name):
        if name == "library":
            e.bug_sweep(0.025, 0)
        else:
            e.step_bug_prepare(0)
            for site in range(L - 1, 0, -1):
                e.step_bug_site(site, 0.025, 0)
            e.step_bug_root(0.025, 0)
        e.step_compress(1e-9, chi, "discarded_weight", 0)
        e.canonicalize_qr(L - 1, 0)

    e.canonicalize_qr(L - 1, 0)
    half()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        half()
    torch.cuda.synchronize()
    out["bug_" + name] = 1e3 * (time.perf_counter() - t0) / n
    print(f"BUG half-sweep + compression, {name}-sequenced: {out['bug_' + name]:.1f} ms", flush=True)
    e.close()
print(f"BUG half-sweep: library / host = {out['bug_library'] / out['bug_host']:.3f}")
