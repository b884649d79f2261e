"""Kernel-level checks of the complex64 library (libtjm_hip_f32.so) at the sizes of BASELINE's configs 3 / 5 (bonds 128 ... 256):
batched GEMM and the QR-preconditioned two-site split against NumPy in fp32 tolerance.  Debug probe; prints one line per case.
    python tests/probes/f32_kernel_probe.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from yaqs_amd import _lib  # noqa: E402
from yaqs_amd._lib import GemmDesc, check  # noqa: E402

DEV = "cuda:0"
lib = _lib.load("complex64")


def crand(rng, *shape):
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)


def gemm_case(M, N, K, nb=3):
    rng = np.random.default_rng(M + N + K)
    a, b = crand(rng, nb, M, K), crand(rng, nb, K, N)
    A, B = torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV)
    Cc = torch.zeros((nb, M, N), dtype=torch.complex64, device=DEV)
    d = GemmDesc()
    for k, v in dict(nks=1, nb0=nb, nb1=1, nb2=1, A=A.data_ptr(), B=B.data_ptr(), C=Cc.data_ptr(), M=M, N=N, K=K, a_rs=K, a_cs=1, b_rs=N, b_cs=1,
                     c_rs=N, a_b0=M * K, b_b0=K * N, c_b0=M * N).items():
        setattr(d, k, v)
    check(lib.tjm_zgemm_batched(C.byref(d), None), "gemm")
    torch.cuda.synchronize()
    ref = np.einsum("bmk,bkn->bmn", a.astype(np.complex128), b.astype(np.complex128))
    err = np.abs(Cc.cpu().numpy() - ref).max() / np.abs(ref).max()
    print(f"gemm {M}x{N}x{K}: rel err {err:.2e}", "OK" if err < 1e-5 else "FAIL", flush=True)


def split_case(capL, capR, dist, qr, graded=False, B=2):
    rng = np.random.default_rng(capL + 7 * capR + dist)
    d = 2
    m, n = d * capL, d * capR
    capM = min(m, n)
    theta = crand(rng, B, m, n)
    if graded:  # the spectrum of a TDVP step: half of the values large, half small
        u, _, vh = np.linalg.svd(theta.astype(np.complex128), full_matrices=False)
        s = np.concatenate([np.linspace(1.0, 0.3, capM // 2), np.geomspace(3e-2, 1e-4, capM - capM // 2)])
        theta = ((u * s) @ vh).astype(np.complex64)
    theta /= np.linalg.norm(theta.reshape(B, -1), axis=1)[:, None, None]
    th = torch.from_numpy(theta).to(DEV)
    left = torch.zeros((B, d, capL, capM), dtype=torch.complex64, device=DEV)
    right = torch.zeros((B, d, capM, capR), dtype=torch.complex64, device=DEV)
    chi = torch.from_numpy(np.stack([np.full(B, capL), np.full(B, capR), np.zeros(B)], axis=1).astype(np.int32)).to(DEV)
    spec_ld = d * max(capL, capR)
    spec = torch.zeros((B, spec_ld), dtype=torch.float32, device=DEV)
    nbytes = (lib.tjm_svd_qr_workspace_bytes if qr else lib.tjm_svd_workspace_bytes)(d * max(capL, capR), B)
    fn = lib.tjm_svd_split_qr if qr else lib.tjm_svd_split
    work = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
    sweeps = C.c_int32(0)
    maxb = capM // 2
    rc = fn(th.data_ptr(), B, d, capL, capR, capM, left.data_ptr(), right.data_ptr(), dist, 0, 1e-12, maxb, 1, chi.data_ptr(), spec.data_ptr(),
            spec_ld, work.data_ptr(), nbytes, C.byref(sweeps), None)
    torch.cuda.synchronize()
    keep = chi.cpu().numpy()[:, 2]
    L_, R_ = left.cpu().numpy().astype(np.complex128), right.cpu().numpy().astype(np.complex128)
    worst = 0.0
    iso = 0.0
    sv = 0.0
    for b in range(B):
        k = keep[b]
        s_ref = np.linalg.svd(theta[b].astype(np.complex128), compute_uv=False)
        u, s, vh = np.linalg.svd(theta[b].astype(np.complex128), full_matrices=False)
        best = (u[:, :k] * s[:k]) @ vh[:k]
        got = np.einsum("sak,tkc->satc", L_[b][:, :, :k], R_[b][:, :k, :]).reshape(m, n)
        worst = max(worst, np.abs(got - best).max())
        if dist == 0:
            q = L_[b][:, :, :k].reshape(m, k)
            iso = max(iso, np.abs(q.conj().T @ q - np.eye(k)).max())
        else:
            q = R_[b][:, :k, :].transpose(1, 0, 2).reshape(k, n)
            iso = max(iso, np.abs(q @ q.conj().T - np.eye(k)).max())
        sv = max(sv, np.abs(spec.cpu().numpy()[b, :capM] - s_ref).max())
    ok = rc == 0 and worst < 2e-4 and iso < 2e-4 and sv < 1e-4 and (keep == maxb).all()
    print(f"split {m}x{n} dist {dist} qr {qr} graded {graded}: rc {rc} keep {keep.tolist()} sweeps {sweeps.value} recon {worst:.2e} iso {iso:.2e} sv {sv:.2e}",
          "OK" if ok else "FAIL", flush=True)


if __name__ == "__main__":
    for M, N, K in [(512, 384, 128), (1024, 1280, 256), (256, 256, 1280), (128, 128, 384)]:
        gemm_case(M, N, K)
    for cap in (64, 128, 192, 256):
        for dist in (0, 1):
            split_case(cap, cap, dist, True)
            split_case(cap, cap, dist, True, graded=True)
    split_case(128, 128, 0, False)
    split_case(256, 128, 0, True)
    split_case(128, 256, 1, True)
