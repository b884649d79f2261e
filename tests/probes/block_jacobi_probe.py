"""Timing probe of the mixed-precision two-site split alone (GPU): 256 matrices of 256 x 256 with a spectrum like the evolved state's
(geometric decay over nine decades), through tjm_svd_split_qr.  Prints the wall time per batched split, the counters of the mixed
path and the sampled launch time of the complex64 Jacobi kernel.  Written for the block-Jacobi experiment of round 4 (DESIGN section 5:
measured, removed - its switches TJM_NO_BLOCK_JACOBI / TJM_MIXED_BLOCK_INNER / TJM_BJ_DEBUG are gone with it); the other switches of
the mixed split (TJM_MIXED_*) can be compared with it in seconds instead of a bench run.

    python tests/probes/block_jacobi_probe.py [B] [reps]
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))


def main():
    import torch

    import test_hip_kernels as k
    from yaqs_amd import _lib

    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    lib = _lib.load()
    rng = np.random.default_rng(7)
    d, cap = 2, 128
    n = d * cap
    sig = np.exp(-np.arange(n) / 12.0)
    base = []
    for _ in range(8):  # eight distinct matrices, repeated: the sweeps per trajectory differ a little, as in the engine
        u = np.linalg.qr(k.crand(rng, n, n))[0]
        v = np.linalg.qr(k.crand(rng, n, n))[0]
        base.append((u * sig) @ v.conj().T)
    theta = np.stack([base[b % 8] for b in range(B)])
    chi = np.full(B, cap, dtype=np.int32)
    out = (C.c_double * 10)()
    k.svd_split_gpu(lib, theta, d, cap, cap, cap, 0, 0, 1e-12, cap, 2, chi, chi, qr=True, want_spec=False)  # warm-up
    lib.tjm_svd_mixed_read(out, 1)
    lib.tjm_profile_cross_kernel(1)
    th = k.dev(theta)
    left = torch.zeros((B, d, cap, cap), dtype=torch.complex128, device=k.DEV)
    right = torch.zeros((B, d, cap, cap), dtype=torch.complex128, device=k.DEV)
    chid = k.dev(np.stack([chi, chi, np.zeros(B, dtype=np.int32)], axis=1).astype(np.int32))
    nbytes = lib.tjm_svd_qr_workspace_bytes(n, B)
    work = torch.zeros(nbytes, dtype=torch.uint8, device=k.DEV)
    sweeps = C.c_int32(0)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        rc = lib.tjm_svd_split_qr(th.data_ptr(), B, d, cap, cap, cap, left.data_ptr(), right.data_ptr(), 0, 0, 1e-12, cap, 2, chid.data_ptr(), None, n,
                                  work.data_ptr(), nbytes, C.byref(sweeps), None)
        assert rc == 0, rc
    torch.cuda.synchronize()
    wall = (time.time() - t0) / reps
    ms, by, ns = C.c_double(), C.c_double(), C.c_int64()
    lib.tjm_profile_cross_kernel_read_c64(C.byref(ms), C.byref(by), C.byref(ns))
    lib.tjm_svd_mixed_read(out, 0)
    L_ = left[0].cpu().numpy().reshape(n, cap)
    R_ = right[0].cpu().numpy().transpose(1, 0, 2).reshape(cap, n)
    ru, rs, rvh = np.linalg.svd(theta[0])
    kb = int(chid.cpu().numpy()[0, 2])
    err = float(np.abs(L_ @ R_ - (ru[:, :kb] * rs[:kb]) @ rvh[:kb]).max())
    print(json.dumps({"B": B, "ms_per_split": 1e3 * wall, "c64_sweeps_per_split": out[1] / max(out[0], 1), "fp64_sweeps": out[2], "fallbacks": out[3],
                      "c64_kernel_avg_us": 1e3 * ms.value / max(ns.value, 1), "c64_kernel_launches_sampled": ns.value, "keep0": kb, "err0": err,
                      "env": {e: os.environ[e] for e in os.environ if e.startswith("TJM_")}}))


if __name__ == "__main__":
    main()
