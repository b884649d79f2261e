"""Micro-benchmark of the batched complex GEMM at the shapes of the two-site H_eff apply (diagnostic tool).

    python tools/gemm_bench.py [B] [reps]
"""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from yaqs_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lib = _lib.load()
shapes = {
    "x*R  (A k-contig, B n-contig)": dict(M=512, N=384, K=128, a_rs=128, a_cs=1, b_rs=384, b_cs=1),
    "L*T  (A m-contig, B n-contig)": dict(M=384, N=512, K=384, a_rs=1, a_cs=384, b_rs=512, b_cs=1),
}
if len(sys.argv) > 3 and "," in sys.argv[3]:  # explicit shapes "M,N,K,am[;M,N,K,am...]" (am = 1: A stored [K][M], m contiguous)
    shapes = {}
    for spec in sys.argv[3].split(";"):
        f = [int(v) for v in spec.split(",")]
        M_, N_, K_, am_ = f[:4]
        nks_, nb1_ = (f[4], f[5]) if len(f) >= 6 else (1, 1)  # k-split terms; inner batches sharing A (the last product of an H_eff apply: 128,128,128,1,2,4)
        shapes[f"M={M_} N={N_} K={K_} x {nks_} terms, {nb1_} inner batches, {'A m-contig' if am_ else 'A k-contig'}"] = dict(
            M=M_, N=N_, K=K_, a_rs=(1 if am_ else K_ * nks_), a_cs=(M_ * nks_ if am_ else 1), b_rs=N_, b_cs=1, nks=nks_, nb1=nb1_)
elif len(sys.argv) > 3:  # K scan at the first shape: fixed cost per tile versus cost per k-tile
    ks = (int(sys.argv[3][2:]),) if sys.argv[3].startswith("K=") else (32, 64, 128, 256, 512)  # "K=512": that K alone (PMC passes)
    shapes = {f"K={k}": dict(M=512, N=384, K=k, a_rs=k, a_cs=1, b_rs=384, b_cs=1) for k in ks}
for name, sh in shapes.items():
    M, N, K = sh["M"], sh["N"], sh["K"]
    nks, nb1 = sh.get("nks", 1), sh.get("nb1", 1)
    A = torch.randn(B, M * K * nks, 2, dtype=torch.float64, device="cuda")
    Bm = torch.randn(B, nb1 * nks * K * N, 2, dtype=torch.float64, device="cuda")
    Cm = torch.zeros(B, nb1 * M * N, 2, dtype=torch.float64, device="cuda")
    g = _lib.GemmDesc()
    g.A, g.B, g.C = A.data_ptr(), Bm.data_ptr(), Cm.data_ptr()
    g.M, g.N, g.K = M, N, K
    g.a_rs, g.a_cs, g.b_rs, g.b_cs, g.c_rs = sh["a_rs"], sh["a_cs"], sh["b_rs"], sh["b_cs"], N
    g.nks, g.nb0, g.nb1, g.nb2 = nks, B, nb1, 1
    g.a_b0, g.b_b0, g.c_b0 = M * K * nks, nb1 * nks * K * N, nb1 * M * N
    g.a_ks, g.b_ks = (M if sh["a_rs"] == 1 else K), K * N   # A [K x nks][M] (m contiguous) or [M][nks x K]; B [nb1][nks][K][N]
    g.b_b1, g.c_b1 = nks * K * N, M * N
    for _ in range(3):
        lib.tjm_zgemm_batched(C.byref(g), None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.tjm_zgemm_batched(C.byref(g), None)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    flop = 8.0 * M * N * K * B * nks * nb1
    print(f"{name}: {dt * 1e3:.3f} ms  {flop / dt / 1e12:.1f} TFLOP/s  ({100 * flop / dt / 78.6e12:.0f} % of the fp64 MFMA peak)")
