"""Stand-alone timing of the batched two-site split (tjm_svd_split_qr) at the headline size, without the 6-minute bench:

    python tools/svd_bench.py [B=1024] [chi=128] [reps=3]           # env switches of DESIGN section 8 apply (TJM_NO_FOLD=1, TJM_NO_LATE_SWEEPS=1 ...)

theta = a chi-saturated two-site tensor after a short evolution: (A_i C) + eps * noise with A_i left-isometric, i.e. 128 large
singular values and 128 small ones, the regime of the TDVP splits of the bench.  Prints ms per batched SVD, TFLOP/s on the
nominal 88 n^3 and the sweep count.  Not part of the product."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch

from yaqs_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
chi = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
d = 2
n = d * chi
lib = _lib.load()
rng = np.random.default_rng(0)
g = torch.Generator(device="cuda:0").manual_seed(1)


def crandn(*shape):
    return torch.complex(torch.randn(*shape, dtype=torch.float64, device="cuda:0", generator=g), torch.randn(*shape, dtype=torch.float64, device="cuda:0", generator=g))


# one host-made pattern, varied per trajectory by a random unitary-ish mixing on the device (kept cheap: no batched QR on the GPU)
a = np.linalg.qr(rng.standard_normal((n, chi)) + 1j * rng.standard_normal((n, chi)))[0]
c = (rng.standard_normal((chi, n)) + 1j * rng.standard_normal((chi, n))) / np.sqrt(chi * n)
base = torch.from_numpy(a @ c).to("cuda:0")
theta = base[None] + 0.05 / n * crandn(B, n, n)
theta = theta.contiguous()
left = torch.zeros((B, d, chi, chi), dtype=torch.complex128, device="cuda:0")
right = torch.zeros((B, d, chi, chi), dtype=torch.complex128, device="cuda:0")
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
print(f"B={B} n={n}: {1e3 * dt:.2f} ms per batched SVD, {88.0 * n ** 3 * B / dt / 1e12:.1f} TFLOP/s nominal, {sweeps.value} sweeps, "
      f"keep {int(chi_lrm[0, 2])}", flush=True)
