"""Debug probe: the forward half of a two-site TDVP sweep driven site by site (tjm_engine_step_*) on the complex64 and the complex128
engine; squared norm of the centre tensor after every site update.   python tests/probes/f32_sweep_probe.py L chi"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from yaqs_amd import api  # noqa: E402
from yaqs_amd.engine import BatchEngine  # noqa: E402

L, chi = int(sys.argv[1]), int(sys.argv[2])
dt = 0.05
mpo = api.MPO.heisenberg(L, 1.0, 1.0, 0.5, 0.0)
st = api.MPS(L, state="haar-random", pad=chi, rng=np.random.default_rng(1))
st.normalize("B")
eng = {}
for dtype in ("complex128", "complex64"):
    e = BatchEngine(L, chi, 1, mpo.tensors, dtype=dtype)
    e.set_params(dt=dt, svd_threshold=1e-12, max_bond_dim=chi, krylov_tol=1e-10 if dtype == "complex128" else 1e-6, tdvp_mode="2site")
    e.load_state([np.asarray(t, dtype=np.complex128) for t in st.tensors])
    e.step_env_init()
    eng[dtype] = e


def centre_norm(e, site):
    t = e.export_state(0)[site]
    return float(np.sum(np.abs(t) ** 2)), t.shape


for i in range(L - 1):
    row = []
    for dtype, e in eng.items():
        s0 = e.stats()
        e.step_two_site(i, 0.5 * dt, "right", True)
        n_split, shp = centre_norm(e, i + 1)
        e.step_env(i, True)
        if i < L - 2:
            e.step_one_site(i + 1, -0.5 * dt)
        n_back, _ = centre_norm(e, i + 1)
        s1 = e.stats()
        row.append((n_split, n_back, shp, s1["matvecs"] - s0["matvecs"], s1["svd_sweeps"] - s0["svd_sweeps"]))
    a, b = row
    flag = "  <<<" if abs(a[0] - b[0]) > 1e-4 else ""
    print(f"site {i:3d} centre {a[2]}: after split f64 {a[0]:.6f} f32 {b[0]:.6f} | after backward step f64 {a[1]:.6f} f32 {b[1]:.6f} | matvecs {a[3]}/{b[3]} sweeps {a[4]}/{b[4]}{flag}")
for e in eng.values():
    e.close()
