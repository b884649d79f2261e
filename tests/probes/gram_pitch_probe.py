"""Does the leading dimension matter for the Gram products of the mixed split (GPU probe)?

G = X^H X for column-major X (N x N, N = 256) is the `zgemm_kernel<false, false>` instance: both operands k-contiguous, rows of the
operand tiles one leading dimension apart.  With ld = 256 complex128 = 4 KB every row of a tile starts on the same address modulo
4 KB.  The probe times the same product with ld = 256, 264, 272, 288 (and the plain X T product of the rounds for comparison).
Result (round 4, one MI355X, 256 matrices): 0.48 - 0.57 ms whatever the leading dimension, the same as X T (0.49 ms = 70 TFLOP/s
nominal, 52 executed): the instance is not slower by itself.  Its 43 % MfmaUtil in the PMC table is the average over launches in which
6 of 16 workgroups (the mirror tiles of a Hermitian product) return at once.

    python tests/probes/gram_pitch_probe.py [B] [reps]
"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from yaqs_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lib = _lib.load()
N = 256


def run(name, ld, kind, herm):
    X = torch.randn(B, N * ld, 2, dtype=torch.float64, device="cuda")
    T = torch.randn(B, N * ld, 2, dtype=torch.float64, device="cuda")
    G = torch.zeros(B, N * ld, 2, dtype=torch.float64, device="cuda")
    g = _lib.GemmDesc()
    g.M, g.N, g.K = N, N, N
    g.nks, g.nb0, g.nb1, g.nb2 = 1, B, 1, 1
    g.a_b0 = g.b_b0 = g.c_b0 = N * ld
    g.C, g.c_rs = G.data_ptr(), ld
    if kind == "gram":  # G[i][j] = sum_r conj(X[r + i ld]) X[r + j ld]
        g.A, g.a_rs, g.a_cs, g.conjA = X.data_ptr(), ld, 1, 1
        g.B, g.b_rs, g.b_cs = X.data_ptr(), 1, ld
    else:               # C[k][r] = sum_j T[j][k] X[j ld + r]
        g.A, g.a_rs, g.a_cs = T.data_ptr(), 1, ld
        g.B, g.b_rs, g.b_cs = X.data_ptr(), ld, 1
    for _ in range(3):
        lib.tjm_zgemm_batched(C.byref(g), None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.tjm_zgemm_batched(C.byref(g), None)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    flop = 8.0 * N * N * N * B * ((10.0 / 16.0) if herm else 1.0)
    print(f"{name:34s} ld {ld}: {dt * 1e3:7.3f} ms  {flop / dt / 1e12:5.1f} TFLOP/s nominal")


for ld in (256, 264, 272, 288):  # (the C ABI's descriptor has no Hermitian flag: all 16 tiles)
    run("Gram X^H X (all tiles)", ld, "gram", False)
for ld in (256, 264):
    run("X T (m-contiguous x n-contiguous)", ld, "times", False)
