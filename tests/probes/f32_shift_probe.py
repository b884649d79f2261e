"""Debug probe: SVD centre shifts of a chi-saturated Haar state on the complex64 engine, one at a time: the squared norm after each
shift (a gauge move: it must stay 1) and the isometry defect of the site left behind.  python tests/probes/f32_shift_probe.py L chi [dtype]"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from yaqs_amd import _lib, api  # noqa: E402
from yaqs_amd.engine import BatchEngine  # noqa: E402

L, chi = int(sys.argv[1]), int(sys.argv[2])
dtype = sys.argv[3] if len(sys.argv) > 3 else "complex64"
st = api.MPS(L, state="haar-random", pad=chi, rng=np.random.default_rng(1))
st.normalize("B")
e = BatchEngine(L, chi, 1, api.MPO.ising(L, 1.0, 0.5).tensors, dtype=dtype)
e.set_params(dt=0.1, svd_threshold=1e-12, max_bond_dim=chi, krylov_tol=1e-6, tdvp_mode="2site")
e.load_state([np.asarray(t, dtype=np.complex128) for t in st.tensors])


def norm2():
    M = e.site_moments()
    return float((M[0, 0, 0, 0] + M[0, 0, 1, 1]).real)


def iso_defect(t, left):
    m = t.transpose(1, 0, 2).reshape(-1, t.shape[2]) if left else t.transpose(1, 0, 2).reshape(t.shape[1], -1)  # (a,p)x b  |  a x (p,b)
    if left:
        m = np.concatenate([t[p] for p in range(t.shape[0])], axis=0)   # (p,a) x b
        g = m.conj().T @ m
    else:
        m = np.concatenate([t[p] for p in range(t.shape[0])], axis=1)   # a x (p,b)
        g = m @ m.conj().T
    return float(np.abs(g - np.eye(g.shape[0])).max())


print(f"{dtype} L={L} chi={chi}: norm2 at start {norm2():.7f}")
for i in range(L - 1):
    _lib.check(e.lib.tjm_engine_center_shift(e.h, 0, i, 1, 1), "shift right")
    out = e.export_state(0)
    vec_n = None
    print(f"  right shift at {i}: bonds {out[i].shape[1]}x{out[i].shape[2]}  isometry defect of site {i}: {iso_defect(out[i], True):.2e}  centre norm2 {float(np.sum(np.abs(out[i + 1]) ** 2)):.7f}")
for i in range(L - 1, 0, -1):
    _lib.check(e.lib.tjm_engine_center_shift(e.h, 0, i, -1, 1), "shift left")
    out = e.export_state(0)
    print(f"  left shift at {i}: bonds {out[i].shape[1]}x{out[i].shape[2]}  isometry defect of site {i}: {iso_defect(out[i], False):.2e}  centre norm2 {float(np.sum(np.abs(out[i - 1]) ** 2)):.7f}")
e.close()
