"""Debug probe: one TJM step on the complex64 and the complex128 engine side by side, stage by stage (norm and bonds after the TDVP
sweep, after the dissipation sweep): where does the fp32 build leave the fp64 one?   python tests/probes/f32_step_probe.py L chi workload"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from yaqs_amd import api  # noqa: E402
from yaqs_amd.api import NoiseModel, is_pauli  # noqa: E402
from yaqs_amd.engine import BatchEngine  # noqa: E402


def vec_norm2(e):
    M = e.site_moments()
    return (M[0, :, 0, 0] + M[0, :, 1, 1]).real


def run(L, chi, workload, dtype):
    if workload == "xxz":
        mpo, proc, gamma, dt = api.MPO.heisenberg(L, 1.0, 1.0, 0.5, 0.0), "lowering", 0.05, 0.05
    elif workload == "xxz-z":
        mpo, proc, gamma, dt = api.MPO.heisenberg(L, 1.0, 1.0, 0.5, 0.0), "pauli_z", 0.05, 0.05
    elif workload == "tfim-low":
        mpo, proc, gamma, dt = api.MPO.ising(L, 1.0, 0.5), "lowering", 0.05, 0.05
    else:
        mpo, proc, gamma, dt = api.MPO.ising(L, 1.0, 0.5), "pauli_z", 0.1, 0.1
    st = api.MPS(L, state="haar-random", pad=chi, rng=np.random.default_rng(1))
    st.normalize("B")
    e = BatchEngine(L, chi, 1, mpo.tensors, dtype=dtype)
    e.set_params(dt=dt, svd_threshold=1e-12, max_bond_dim=chi, krylov_tol=1e-10 if dtype == "complex128" else 1e-6, tdvp_mode="2site")
    noise = NoiseModel([{"name": proc, "sites": [i], "strength": gamma} for i in range(L)])
    e.set_noise(noise.processes, [is_pauli(q) for q in noise.processes])
    e.load_state([np.asarray(t, dtype=np.complex128) for t in st.tensors])
    out = {"n0": vec_norm2(e)[0]}
    e.tdvp()
    out["n_tdvp"] = vec_norm2(e)[0]
    out["bonds_tdvp"] = e.bond_dims()[0].tolist()
    out["stats"] = e.stats()
    e.dissipate(dt)
    out["n_diss"] = vec_norm2(e)[0]
    out["bonds_diss"] = e.bond_dims()[0].tolist()
    e.close()
    return out


if __name__ == "__main__":
    L, chi, wl = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    a, b = run(L, chi, wl, "complex128"), run(L, chi, wl, "complex64")
    for k in ("n0", "n_tdvp", "n_diss"):
        print(f"{wl} L={L} chi={chi} {k}: f64 {a[k]:.6f}  f32 {b[k]:.6f}")
    print("  bonds equal after tdvp:", a["bonds_tdvp"] == b["bonds_tdvp"], " after dissipation:", a["bonds_diss"] == b["bonds_diss"])
    print("  f64 stats", a["stats"]); print("  f32 stats", b["stats"])
    if a["bonds_tdvp"] != b["bonds_tdvp"]:
        print("  f64", a["bonds_tdvp"]); print("  f32", b["bonds_tdvp"])
