"""Stand-alone timing of the batched two-site split of the complex64 library (libtjm_hip_f32.so) - the splits of BASELINE configs 3
(512 x 512) and 5 (1024 x 1024) - on ONE stream, nothing else on the GPU:

    python tools/svd_bench32.py [B=32] [chi=256] [reps=2]        # under rocprofv3 --kernel-trace --stats: launch times of the tile kernels alone
    TJM_Q64_ABLATE=1 python tools/svd_bench32.py ...             # the loads and column norms of jacobi_quad64_kernel only (wrong results)

theta as in tools/svd_bench.py (an evolved chi-saturated two-site tensor: chi large singular values, chi small ones).  Not part of the product."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch

from yaqs_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
chi = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
d = 2
n = d * chi
lib = _lib.load("complex64")
rng = np.random.default_rng(0)
g = torch.Generator(device="cuda:0").manual_seed(1)
a = np.linalg.qr(rng.standard_normal((n, chi)) + 1j * rng.standard_normal((n, chi)))[0]
c = (rng.standard_normal((chi, n)) + 1j * rng.standard_normal((chi, n))) / np.sqrt(chi * n)
base = torch.from_numpy((a @ c).astype(np.complex64)).to("cuda:0")
noise = torch.complex(torch.randn(B, n, n, dtype=torch.float32, device="cuda:0", generator=g), torch.randn(B, n, n, dtype=torch.float32, device="cuda:0", generator=g))
theta = (base[None] + 0.05 / n * noise).contiguous()
left = torch.zeros((B, d, chi, chi), dtype=torch.complex64, device="cuda:0")
right = torch.zeros((B, d, chi, chi), dtype=torch.complex64, device="cuda:0")
nbytes = lib.tjm_svd_qr_workspace_bytes(n, B)
work = torch.zeros(nbytes, dtype=torch.uint8, device="cuda:0")
chi_lrm = torch.tensor([[chi, chi, 0]] * B, dtype=torch.int32, device="cuda:0")
sweeps = C.c_int32(0)


def run():
    _lib.check(lib.tjm_svd_split_qr(theta.data_ptr(), B, d, chi, chi, chi, left.data_ptr(), right.data_ptr(), 0, 0, 1e-12, chi, 2, chi_lrm.data_ptr(),
                                    None, 0, work.data_ptr(), nbytes, C.byref(sweeps), None), "svd_split_qr")
    torch.cuda.synchronize()


run()
t0 = time.perf_counter()
for _ in range(reps):
    run()
dt = (time.perf_counter() - t0) / reps
print(f"complex64 B={B} n={n}: {1e3 * dt:.2f} ms per batched split, {sweeps.value} sweeps, keep {int(chi_lrm[0, 2])}", flush=True)
