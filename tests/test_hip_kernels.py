"""GPU parity tests for the individual HIP kernels, called through the C ABI, checked against
NumPy / the CPU oracle on the same seeded inputs."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from conftest import SIM  # noqa: E402  (TJM_SIM=1: the same tests on tests/hipsim, host memory for device memory)

DEV = "cpu" if SIM else "cuda:0"


def _sync():
    if not SIM:
        torch.cuda.synchronize()


def dev(a):
    return torch.from_numpy(np.array(a, order="C", copy=True)).to(DEV)  # a copy also on the host: kernels work in place


def crand(rng, *shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


@pytest.fixture(scope="module")
def lib():
    from yaqs_amd import _lib

    if SIM:
        from simengine import load_sim

        return load_sim()
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return _lib.load()


def run_gemm(lib, **kw):
    from yaqs_amd._lib import GemmDesc, check

    d = GemmDesc()
    defaults = dict(nks=1, nb0=1, nb1=1, nb2=1)
    defaults.update(kw)
    for k, v in defaults.items():
        setattr(d, k, v)
    check(lib.tjm_zgemm_batched(C.byref(d), None), "gemm")
    _sync()


@pytest.mark.parametrize("M,N,K", [(64, 64, 16), (70, 50, 37), (256, 384, 128), (3, 5, 2), (128, 128, 384), (512, 384, 128)])
@pytest.mark.parametrize("conjA,conjB", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_gemm_nn_batched(lib, M, N, K, conjA, conjB):
    rng = np.random.default_rng(M * 1000 + N + K)
    nb = 3
    a, b = crand(rng, nb, M, K), crand(rng, nb, K, N)
    A, B = dev(a), dev(b)
    Cc = torch.zeros((nb, M, N), dtype=torch.complex128, device=DEV)
    run_gemm(lib, A=A.data_ptr(), B=B.data_ptr(), C=Cc.data_ptr(), M=M, N=N, K=K, a_rs=K, a_cs=1, b_rs=N, b_cs=1, c_rs=N,
             nb0=nb, a_b0=M * K, b_b0=K * N, c_b0=M * N, conjA=conjA, conjB=conjB)
    ref = np.einsum("bmk,bkn->bmn", a.conj() if conjA else a, b.conj() if conjB else b)
    assert np.allclose(Cc.cpu().numpy(), ref, atol=1e-11 * K)


@pytest.mark.parametrize("M,N,K", [(32, 24, 8), (17, 200, 33), (130, 9, 4), (16, 16, 512), (1, 1, 1), (31, 31, 127)])
def test_gemm_small_tile_path(lib, M, N, K):
    """Products with M <= 32 or N <= 32 and K <= 512 run on the one-tile-per-wavefront kernel (small bonds): strided and
    conjugated operands and the K-split sum against numpy."""
    rng = np.random.default_rng(M * 131 + N * 7 + K)
    nb, O = 5, 2
    a, b = crand(rng, nb, O, K, M), crand(rng, nb, O, N, K)  # A stored [K][M] (m contiguous), B stored [N][K] (k contiguous)
    A, B = dev(a), dev(b)
    nks = O if K * O <= 512 else 1
    Cc = torch.zeros((nb, M, N), dtype=torch.complex128, device=DEV)
    run_gemm(lib, A=A.data_ptr(), B=B.data_ptr(), C=Cc.data_ptr(), M=M, N=N, K=K, a_rs=1, a_cs=M, b_rs=1, b_cs=K, c_rs=N,
             nks=nks, a_ks=K * M, b_ks=N * K, nb0=nb, a_b0=O * K * M, b_b0=O * N * K, c_b0=M * N, conjA=1, conjB=1)
    want = np.einsum("bokm,bonk->bmn", a[:, :nks].conj(), b[:, :nks].conj())
    assert np.allclose(Cc.cpu().numpy(), want, atol=1e-11 * max(K, 8))


def test_gemm_transposed_operands_ksplit_and_inner_batches(lib):
    rng = np.random.default_rng(7)
    # C[b][o] = A^T B_o  (A stored [K][M]), inner batch over o
    M, N, K, nb, P = 96, 80, 150, 2, 4
    a, b = crand(rng, nb, K, M), crand(rng, nb, P, K, N)
    A, B = dev(a), dev(b)
    Cc = torch.zeros((nb, P, M, N), dtype=torch.complex128, device=DEV)
    run_gemm(lib, A=A.data_ptr(), B=B.data_ptr(), C=Cc.data_ptr(), M=M, N=N, K=K, a_rs=1, a_cs=M, b_rs=N, b_cs=1, c_rs=N,
             nb0=nb, nb1=P, a_b0=K * M, b_b0=P * K * N, b_b1=K * N, c_b0=P * M * N, c_b1=M * N, conjA=1)
    ref = np.einsum("bkm,bpkn->bpmn", a.conj(), b)
    assert np.allclose(Cc.cpu().numpy(), ref, atol=1e-10)
    # C = sum_o A[:, o-block] B_o^H  (B stored [N][K], K-split over o)
    M, N, K, O = 40, 33, 29, 2
    a, b = crand(rng, nb, M, O, K), crand(rng, nb, O, N, K)
    A, B = dev(a), dev(b)
    Cc = torch.zeros((nb, M, N), dtype=torch.complex128, device=DEV)
    run_gemm(lib, A=A.data_ptr(), B=B.data_ptr(), C=Cc.data_ptr(), M=M, N=N, K=K, a_rs=O * K, a_cs=1, b_rs=1, b_cs=K, c_rs=N,
             nks=O, a_ks=K, b_ks=N * K, nb0=nb, a_b0=M * O * K, b_b0=O * N * K, c_b0=M * N, conjB=1)
    ref = np.einsum("bmok,bonk->bmn", a, b.conj())
    assert np.allclose(Cc.cpu().numpy(), ref, atol=1e-10)
    # two inner batch levels (merge_two_site into tensor layout)
    d, ca, cm, cc = 2, 7, 5, 6
    x, y = crand(rng, nb, d, ca, cm), crand(rng, nb, d, cm, cc)
    X, Y = dev(x), dev(y)
    Cc = torch.zeros((nb, d, d, ca, cc), dtype=torch.complex128, device=DEV)
    run_gemm(lib, A=X.data_ptr(), B=Y.data_ptr(), C=Cc.data_ptr(), M=ca, N=cc, K=cm, a_rs=cm, a_cs=1, b_rs=cc, b_cs=1, c_rs=cc,
             nb0=nb, nb1=d, nb2=d, a_b0=d * ca * cm, a_b1=ca * cm, b_b0=d * cm * cc, b_b2=cm * cc, c_b0=d * d * ca * cc,
             c_b1=d * ca * cc, c_b2=ca * cc)
    ref = np.einsum("bsam,btmc->bstac", x, y)
    assert np.allclose(Cc.cpu().numpy(), ref, atol=1e-11)


@pytest.mark.parametrize("M,N,K,nb,P", [(70, 130, 37, 24, 3), (128, 128, 32, 128, 1), (64, 192, 16, 40, 4), (96, 80, 150, 17, 2)])
def test_gemm_in_xcd_aware_order(lib, M, N, K, nb, P):
    """Whole-tile launches of at least 16 trajectories and 512 tiles deal their tiles to the persistent workgroups in XCD-aware order
    (round 5: a trajectory with all its inner batches goes to workgroups of equal index mod 8); beside them ragged shapes and smaller
    launches with inner batches sharing A, on either kernel."""
    rng = np.random.default_rng(M + 3 * N + 7 * K + nb)
    a, b = crand(rng, nb, M, K), crand(rng, nb, P, K, N)
    A, B = dev(a), dev(b)
    Cc = torch.zeros((nb, P, M, N), dtype=torch.complex128, device=DEV)
    run_gemm(lib, A=A.data_ptr(), B=B.data_ptr(), C=Cc.data_ptr(), M=M, N=N, K=K, a_rs=K, a_cs=1, b_rs=N, b_cs=1, c_rs=N,
             nb0=nb, nb1=P, a_b0=M * K, b_b0=P * K * N, b_b1=K * N, c_b0=P * M * N, c_b1=M * N, conjB=1)
    assert np.allclose(Cc.cpu().numpy(), np.einsum("bmk,bpkn->bpmn", a, b.conj()), atol=1e-11 * K)


def test_gemm_with_more_batches_than_one_grid_dimension_holds(lib):
    """3 x 150 x 150 = 67 500 batched products through the LDS-tiled kernel: more than the 65 535 of grid z (16 384 trajectories
    with two physical indices each reach it), so the batch index spills into grid y.  The innermost batch level writes the same C."""
    rng = np.random.default_rng(11)
    M, N, K, nb, P, Q = 40, 40, 3, 3, 150, 150
    a, b = crand(rng, nb, M, K), crand(rng, nb, P, K, N)
    A, B = dev(a), dev(b)
    Cc = torch.zeros((nb, P, M, N), dtype=torch.complex128, device=DEV)
    run_gemm(lib, A=A.data_ptr(), B=B.data_ptr(), C=Cc.data_ptr(), M=M, N=N, K=K, a_rs=K, a_cs=1, b_rs=N, b_cs=1, c_rs=N,
             nb0=nb, nb1=P, nb2=Q, a_b0=M * K, b_b0=P * K * N, b_b1=K * N, c_b0=P * M * N, c_b1=M * N)
    _sync()
    assert np.allclose(Cc.cpu().numpy(), np.einsum("bmk,bpkn->bpmn", a, b), atol=1e-12)


@pytest.mark.parametrize("k,dt,scale", [(1, 0.05, 1.0), (2, 0.1, 3.0), (7, -0.05, 20.0), (25, 0.1, 60.0), (12, 1.0, 40.0)])
def test_tridiag_expm_matches_dense(lib, k, dt, scale):
    import scipy.linalg

    from yaqs_amd._lib import check

    rng = np.random.default_rng(k)
    alpha = rng.standard_normal(25) * scale
    beta = np.abs(rng.standard_normal(25)) * scale * 0.5
    T = np.diag(alpha[:k]) + np.diag(beta[: k - 1], 1) + np.diag(beta[: k - 1], -1)
    ref = scipy.linalg.expm(-1j * dt * T)[:, 0]
    a, b = dev(alpha), dev(beta)
    out = torch.zeros(2 * k, dtype=torch.float64, device=DEV)
    check(lib.tjm_tridiag_expm(a.data_ptr(), b.data_ptr(), k, dt, out.data_ptr(), None), "expm")
    _sync()
    got = out.cpu().numpy().view(np.complex128)
    assert np.allclose(got, ref, atol=5e-13)


def svd_split_gpu(lib, theta, d, capL, capR, capM, dist, mode, thr, max_bond, min_keep, chiL, chiR, qr=False, want_spec=True, stream=None):
    from yaqs_amd._lib import check

    if stream is not None:  # the whole call - allocations, uploads, the split, the read-back - on the caller's HIP stream
        with torch.cuda.stream(stream):
            out = svd_split_gpu(lib, theta, d, capL, capR, capM, dist, mode, thr, max_bond, min_keep, chiL, chiR, qr, want_spec, None)
            stream.synchronize()
        return out

    B = theta.shape[0]
    th = dev(theta)
    left = torch.zeros((B, d, capL, capM), dtype=torch.complex128, device=DEV)
    right = torch.zeros((B, d, capM, capR), dtype=torch.complex128, device=DEV)
    chi = dev(np.stack([chiL, chiR, np.zeros(B, dtype=np.int32)], axis=1).astype(np.int32))
    spec_ld = d * max(capL, capR)
    spec = torch.zeros((B, spec_ld), dtype=torch.float64, device=DEV)
    nbytes = (lib.tjm_svd_qr_workspace_bytes if qr else lib.tjm_svd_workspace_bytes)(d * max(capL, capR), B)
    fn = lib.tjm_svd_split_qr if qr else lib.tjm_svd_split
    work = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
    sweeps = C.c_int32(0)
    cur = None if SIM else C.c_void_p(torch.cuda.current_stream().cuda_stream)  # (the default stream unless a caller's stream is current)
    check(fn(th.data_ptr(), B, d, capL, capR, capM, left.data_ptr(), right.data_ptr(), dist, mode, thr, max_bond, min_keep,
                            chi.data_ptr(), spec.data_ptr() if want_spec else None, spec_ld, work.data_ptr(), nbytes, C.byref(sweeps), cur), "svd_split")
    _sync()
    return left.cpu().numpy(), right.cpu().numpy(), chi.cpu().numpy()[:, 2], spec.cpu().numpy(), sweeps.value


@pytest.mark.parametrize("qr", [False, True])
@pytest.mark.parametrize("capL,capR,dist", [(4, 4, 0), (4, 4, 1), (8, 3, 0), (3, 8, 1), (16, 16, 0), (32, 32, 1), (1, 2, 0), (2, 1, 1), (32, 48, 0), (40, 32, 1)])
def test_svd_split_matches_oracle(lib, capL, capR, dist, qr):
    from oracle import tjm_oracle as o

    rng = np.random.default_rng(capL * 10 + capR + dist)
    d, B = 2, 5
    capM = min(d * capL, d * capR)
    theta = crand(rng, B, d * capL, d * capR)
    # make two of them low rank and one tiny
    theta[1] = crand(rng, d * capL, 1) @ crand(rng, 1, d * capR)
    theta[2] *= 1e-3
    chiL = np.full(B, capL, dtype=np.int32)
    chiR = np.full(B, capR, dtype=np.int32)
    thr, maxb = 1e-6, max(1, capM - 1)
    mk = min(2, maxb)
    left, right, keep, spec, sweeps = svd_split_gpu(lib, theta, d, capL, capR, capM, dist, 0, thr, maxb, mk, chiL, chiR, qr=qr)
    for b in range(B):
        merged = theta[b].reshape(d, capL, d, capR).transpose(0, 2, 1, 3).reshape(d * d, capL, capR)
        l_ref, r_ref, s_ref = o.split_two_site(merged, [d, d], svd_distribution="right" if dist == 0 else "left", trunc_mode="discarded_weight",
                                                threshold=thr, max_bond_dim=maxb, min_keep=mk, return_spectrum=True)
        k = l_ref.shape[2]
        assert keep[b] == k, (b, keep[b], k)
        assert np.allclose(spec[b, : len(s_ref)], s_ref, atol=1e-12 * max(1.0, s_ref[0])), b
        got = o.merge_two_site(left[b][:, :, :k], right[b][:, :k, :])
        ref = o.merge_two_site(l_ref, r_ref)
        assert np.allclose(got, ref, atol=1e-11 * max(1.0, s_ref[0])), b
        # padding beyond keep is exactly zero and the isometric side is isometric
        assert np.all(left[b][:, :, k:] == 0) and np.all(right[b][:, k:, :] == 0)
        if dist == 0:
            iso = left[b][:, :, :k].reshape(d * capL, k)
            assert np.allclose(iso.conj().T @ iso, np.eye(k), atol=1e-12)
        else:
            iso = right[b][:, :k, :].transpose(1, 0, 2).reshape(k, d * capR)
            assert np.allclose(iso @ iso.conj().T, np.eye(k), atol=1e-12)


@pytest.mark.parametrize("qr", [False, True])
def test_svd_split_ragged_bonds_and_modes(lib, qr):
    from oracle import tjm_oracle as o

    rng = np.random.default_rng(3)
    d, capL, capR, capM, B = 2, 8, 8, 8, 4
    chiL = np.array([8, 3, 1, 5], dtype=np.int32)
    chiR = np.array([8, 2, 4, 5], dtype=np.int32)
    theta = np.zeros((B, d * capL, d * capR), dtype=np.complex128)
    for b in range(B):
        t = crand(rng, d, chiL[b], d, chiR[b])
        full = np.zeros((d, capL, d, capR), dtype=np.complex128)
        full[:, : chiL[b], :, : chiR[b]] = t
        theta[b] = full.reshape(d * capL, d * capR)
    for mode, name, thr in [(0, "discarded_weight", 0.3), (1, "relative", 0.4), (2, "hard_cutoff", 1.5), (3, "relative_discarded_weight", 0.05)]:
        left, right, keep, spec, _ = svd_split_gpu(lib, theta, d, capL, capR, capM, 0, mode, thr, 8, 1, chiL, chiR, qr=qr)
        for b in range(B):
            t = theta[b].reshape(d, capL, d, capR)[:, : chiL[b], :, : chiR[b]]
            merged = t.transpose(0, 2, 1, 3).reshape(d * d, chiL[b], chiR[b])
            l_ref, r_ref = o.split_two_site(merged, [d, d], svd_distribution="right", trunc_mode=name, threshold=thr, max_bond_dim=8, min_keep=1)
            k = l_ref.shape[2]
            assert keep[b] == k, (name, b, keep[b], k)
            got = o.merge_two_site(left[b][:, : chiL[b], :k], right[b][:, :k, : chiR[b]])
            assert np.allclose(got, o.merge_two_site(l_ref, r_ref), atol=1e-11), (name, b)
            # zero padding stays exactly zero and the active block of the isometric factor is isometric on its own
            assert np.all(left[b][:, chiL[b]:, :] == 0) and np.all(right[b][:, :, chiR[b]:] == 0), (name, b)
            iso = left[b][:, : chiL[b], :k].reshape(d * chiL[b], k)
            assert np.allclose(iso.conj().T @ iso, np.eye(k), atol=1e-12), (name, b)


@pytest.mark.parametrize("qr", [False, True])
def test_svd_split_chi128_rank_deficient(lib, qr):
    """Full-size case of the headline config: 256 x 256 theta of rank 128 (a centre-shift merge)."""
    rng = np.random.default_rng(11)
    d, cap, B = 2, 128, 2
    a = crand(rng, B, d * cap, cap) / np.sqrt(d * cap * cap)
    q = np.linalg.qr(crand(rng, B, d * cap, cap))[0].conj().transpose(0, 2, 1)  # right-isometric (cap x d*cap)
    theta = a @ q
    chi = np.full(B, cap, dtype=np.int32)
    left, right, keep, spec, sweeps = svd_split_gpu(lib, theta, d, cap, cap, cap, 0, 0, 1e-12, 0, 1, chi, chi, qr=qr)
    print('sweeps', sweeps, 'qr', qr)
    for b in range(B):
        s_ref = np.linalg.svd(theta[b], compute_uv=False)
        assert keep[b] == cap
        assert np.allclose(spec[b, :cap], s_ref[:cap], atol=1e-12)
        rec = left[b].reshape(d * cap, cap) @ right[b].transpose(1, 0, 2).reshape(cap, d * cap)
        assert np.allclose(rec, theta[b], atol=1e-12)
    assert sweeps < 40


@pytest.mark.parametrize("dist", [0, 1])
def test_svd_split_qr_graded_full_size(lib, dist):
    """256 x 256 theta with a spectrum graded over 8 decades (the time-evolved two-site tensor regime)."""
    rng = np.random.default_rng(21 + dist)
    d, cap, B = 2, 128, 2
    n = d * cap
    theta = np.zeros((B, n, n), dtype=np.complex128)
    svs = []
    for b in range(B):
        u = np.linalg.qr(crand(rng, n, n))[0]
        v = np.linalg.qr(crand(rng, n, n))[0]
        s = np.concatenate([np.linspace(1.0, 0.05, cap), 10.0 ** rng.uniform(-8, -2, cap)])
        s = np.sort(s)[::-1]
        svs.append(s)
        theta[b] = (u * s) @ v.conj().T
    chi = np.full(B, cap, dtype=np.int32)
    out = {}
    for qr in (False, True):
        left, right, keep, spec, sweeps = svd_split_gpu(lib, theta, d, cap, cap, cap, dist, 0, 1e-12, cap, 2, chi, chi, qr=qr)
        out[qr] = sweeps
        for b in range(B):
            assert keep[b] == cap
            assert np.allclose(spec[b, :n], svs[b], atol=1e-12)
            L_ = left[b].reshape(n, cap)
            R_ = right[b].transpose(1, 0, 2).reshape(cap, n)
            best = (u_ := None)
            ref_u, ref_s, ref_vh = np.linalg.svd(theta[b])
            trunc = (ref_u[:, :cap] * ref_s[:cap]) @ ref_vh[:cap]
            assert np.allclose(L_ @ R_, trunc, atol=1e-10)
            iso = L_ if dist == 0 else R_.conj().T
            assert np.allclose(iso.conj().T @ iso, np.eye(cap), atol=1e-12)
    print("sweeps plain", out[False], "qr", out[True])
    assert out[True] < out[False]


@pytest.mark.parametrize("qr,dist", [(False, 0), (True, 0), (True, 1)])
def test_svd_split_chi256_full_size(lib, qr, dist):
    """d*chi = 512: the largest two-site split the register-resident Jacobi holds (16 row groups per stacked column,
    8 per X column in the split scheme); graded spectrum, truncation to chi = 256."""
    rng = np.random.default_rng(31 + dist)
    d, cap, B = 2, 256, 2
    n = d * cap
    theta = np.zeros((B, n, n), dtype=np.complex128)
    svs = []
    for b in range(B):
        u = np.linalg.qr(crand(rng, n, n))[0]
        v = np.linalg.qr(crand(rng, n, n))[0]
        s = np.sort(np.concatenate([np.linspace(1.0, 0.05, cap), 10.0 ** rng.uniform(-7, -2, cap)]))[::-1]
        svs.append(s)
        theta[b] = (u * s) @ v.conj().T
    chi = np.full(B, cap, dtype=np.int32)
    left, right, keep, spec, sweeps = svd_split_gpu(lib, theta, d, cap, cap, cap, dist, 0, 1e-12, cap, 2, chi, chi, qr=qr)
    print("sweeps", sweeps, "qr", qr)
    for b in range(B):
        assert keep[b] == cap
        assert np.allclose(spec[b, :n], svs[b], atol=1e-12)
        L_ = left[b].reshape(n, cap)
        R_ = right[b].transpose(1, 0, 2).reshape(cap, n)
        ref_u, ref_s, ref_vh = np.linalg.svd(theta[b])
        trunc = (ref_u[:, :cap] * ref_s[:cap]) @ ref_vh[:cap]
        assert np.allclose(L_ @ R_, trunc, atol=1e-10)
        iso = L_ if dist == 0 else R_.conj().T
        assert np.allclose(iso.conj().T @ iso, np.eye(cap), atol=1e-12)
    assert sweeps < 40


@pytest.mark.parametrize("capL,capR", [(256, 128), (128, 256), (200, 200), (256, 64)])
def test_svd_split_large_rectangular_and_odd_sizes(lib, capL, capR):
    """Shapes next to the bond cap and bond dimensions that are not powers of two (generic 8-column path at > 512 rows)."""
    rng = np.random.default_rng(capL + capR)
    d, B = 2, 2
    m, n = d * capL, d * capR
    capM = min(m, n, 256)
    theta = crand(rng, B, m, n) / np.sqrt(m * n)
    chiL = np.full(B, capL, dtype=np.int32)
    chiR = np.full(B, capR, dtype=np.int32)
    for qr in (False, True):
        left, right, keep, spec, sweeps = svd_split_gpu(lib, theta, d, capL, capR, capM, 0, 0, 1e-12, capM, 2, chiL, chiR, qr=qr)
        for b in range(B):
            ref_u, ref_s, ref_vh = np.linalg.svd(theta[b], full_matrices=False)
            assert keep[b] == capM
            assert np.allclose(spec[b, : len(ref_s)], ref_s, atol=1e-12)
            L_ = left[b].reshape(m, capM)
            R_ = right[b].transpose(1, 0, 2).reshape(capM, n)
            trunc = (ref_u[:, :capM] * ref_s[:capM]) @ ref_vh[:capM]
            assert np.allclose(L_ @ R_, trunc, atol=1e-11)
            assert np.allclose(L_.conj().T @ L_, np.eye(capM), atol=1e-12)


@pytest.mark.parametrize("dist", [0, 1])
def test_svd_split_qr_rank_deficient_and_ragged_square(lib, dist):
    """Square theta through the accumulation-free doubly preconditioned path: kept ZERO singular values (threshold 0 keeps
    everything up to max_bond, the rank is smaller) and ragged zero-padded bonds.  The isometric factor must stay exactly
    isometric (orthonormal completion by the QR), padding exactly zero, and the product must reproduce theta."""
    rng = np.random.default_rng(77 + dist)
    d, cap, B = 2, 32, 4
    n = d * cap
    chiL = np.array([32, 32, 11, 20], dtype=np.int32)
    chiR = np.array([32, 32, 17, 5], dtype=np.int32)
    theta = np.zeros((B, n, n), dtype=np.complex128)
    theta[0] = crand(rng, n, 10) @ crand(rng, 10, n)          # rank 10, keep 32
    theta[1] = crand(rng, n, n)                                # full rank
    for b in (2, 3):
        t = crand(rng, d, chiL[b], d, chiR[b])
        full = np.zeros((d, cap, d, cap), dtype=np.complex128)
        full[:, : chiL[b], :, : chiR[b]] = t
        theta[b] = full.reshape(n, n)
    capM = 32
    left, right, keep, spec, sweeps = svd_split_gpu(lib, theta, d, cap, cap, capM, dist, 0, 0.0, capM, 2, chiL, chiR, qr=True)
    for b in range(B):
        k = keep[b]
        nsv = min(d * chiL[b], d * chiR[b])
        assert k == min(capM, nsv), (b, k)
        s_ref = np.linalg.svd(theta[b], compute_uv=False)
        assert np.allclose(spec[b, :n], s_ref, atol=1e-12 * s_ref[0])
        L_ = left[b].reshape(n, capM)
        R_ = right[b].transpose(1, 0, 2).reshape(capM, n)
        u, s, vh = np.linalg.svd(theta[b])
        trunc = (u[:, :k] * s[:k]) @ vh[:k]
        assert np.allclose(L_ @ R_, trunc, atol=1e-11 * s_ref[0]), b
        iso = L_[:, :k] if dist == 0 else R_[:k].conj().T
        assert np.allclose(iso.conj().T @ iso, np.eye(k), atol=1e-12), b
        assert np.all(left[b][:, chiL[b]:, :] == 0) and np.all(right[b][:, :, chiR[b]:] == 0), b
        assert np.all(left[b][:, :, k:] == 0) and np.all(right[b][:, k:, :] == 0), b


@pytest.mark.parametrize("svs,mode,thr,expected", [
    ([1.0, 0.5, 0.1, 0.0100001], 0, 1e-4, 4), ([1.0, 0.5, 0.01, 0.001], 0, 1e-4, 3), ([1.0, 0.2, 0.2, 0.2], 0, 0.2 ** 2 * 3, 1),
    ([1.0, 0.6, 0.4, 0.1], 1, 0.5, 2), ([1.0, 0.99, 0.98], 1, 0.95, 3), ([1.0, 0.55, 0.3], 1, 0.5, 2)])
def test_truncation_known_answers_of_the_reference(lib, svs, mode, thr, expected):
    """The keep counts of the reference's own known-answer tests (tests/core/methods/tdvp/test_sweep_utils.py:136-222:
    discarded_weight and relative modes, min_keep = 2 as split_tdvp passes it) through the GPU split."""
    rng = np.random.default_rng(11)
    d, capL, capR = 2, 3, 3
    m, n = d * capL, d * capR
    s = np.zeros(min(m, n))
    s[: len(svs)] = svs
    u = np.linalg.qr(crand(rng, m, m))[0]
    v = np.linalg.qr(crand(rng, n, n))[0]
    theta = ((u[:, : len(s)] * s) @ v[:, : len(s)].conj().T)[None]
    chi = np.array([3], dtype=np.int32)
    for qr in (False, True):
        left, right, keep, spec, _ = svd_split_gpu(lib, theta, d, capL, capR, min(m, n), 0, mode, thr, 0, 2, chi, chi, qr=qr)
        boundary = mode == 0 and abs(np.sum(np.square(svs[expected:])) - thr) < 64 * np.finfo(float).eps
        assert keep[0] == max(expected, 2) or (boundary and abs(keep[0] - expected) <= 1), (keep, expected)
        assert np.allclose(spec[0, : len(svs)], svs, atol=1e-13)


@pytest.mark.parametrize("svs,mode,thr,max_bond,min_keep,expected", [
    ([10.0, 3.0, 1.0, 0.5], 0, 10.0, 0, 1, 2),             # discarded_weight: 0.25 + 1 < 10 <= 0.25 + 1 + 9
    ([2.0, 1.0, 0.4], 1, 0.45, 0, 1, 2),                   # relative: 0.5 >= 0.45 > 0.2
    ([5.0, 2.0, 0.5, 0.1], 2, 0.2, 0, 1, 3),               # hard_cutoff: s > 0.2
    ([5.0, 2.0, 0.5, 0.1], 2, 0.2, 2, 1, 2),               # ... capped by max_bond_dim
    ([10.0, 1.0, 0.1, 0.01], 3, 1e-3, 0, 1, 2),            # relative_discarded_weight: (0.01 + 0.0001) / 101.0101 <= 1e-3 < 1.0101 / 101.0101
    ([10.0, 1.0, 0.1, 0.01], 3, 0.02, 0, 1, 1),
    ([4.0, 2.0, 0.5, 0.1], 3, 0.05, 0, 1, 2),              # (0.25 + 0.01) / 20.26 <= 0.05 < 4.26 / 20.26
])
def test_truncation_modes_hand_computed_cases_of_the_reference(lib, svs, mode, thr, max_bond, min_keep, expected):
    """tests/core/linalg/test_svd_utils.py:20-80 of the reference (all four truncation modes, max_bond_dim, min_keep) through the
    GPU split: the spectrum is planted in a random two-site block; one-kernel and general paths.  (The reference's threshold-0 case
    needs exactly zero trailing values, which no factorisation of a matrix returns; it stays an oracle-level test.)"""
    rng = np.random.default_rng(23)
    d, capL, capR = 2, 3, 3
    m, n = d * capL, d * capR
    s = np.zeros(min(m, n))
    s[: len(svs)] = svs
    u = np.linalg.qr(crand(rng, m, m))[0]
    v = np.linalg.qr(crand(rng, n, n))[0]
    theta = ((u[:, : len(s)] * s) @ v[:, : len(s)].conj().T)[None]
    chi = np.array([3], dtype=np.int32)
    for qr in (False, True):
        _, _, keep, spec, _ = svd_split_gpu(lib, theta, d, capL, capR, min(m, n), 0, mode, thr, max_bond, min_keep, chi, chi, qr=qr)
        assert keep[0] == expected, (keep, expected, qr)
        assert np.allclose(spec[0, : len(svs)], svs, atol=1e-12)
    # relative_discarded_weight is scale invariant (test_svd_utils.py:62-69).  The reference scales the bare spectrum by 1e+-200; the
    # GPU path factorises a matrix, whose squared column norms and their products must stay representable: overall scales within
    # about 1e+-70 (the states of the path are normalised).
    for scale in (3.7, 1e40, 1e-40):
        if mode == 3:
            _, _, keep, _, _ = svd_split_gpu(lib, theta * scale, d, capL, capR, min(m, n), 0, mode, thr, max_bond, min_keep, chi, chi, qr=False)
            assert keep[0] == expected, (scale, keep)


@pytest.mark.parametrize("dist", [0, 1])
def test_svd_split_steeply_graded_spectrum_keeps_small_values_accurately(lib, dist):
    """Singular values over ten decades, all kept (threshold far below the smallest): the direct variant takes the isometric
    factor from normalised rotated columns, so this checks the relative accuracy of the small ones and the isometry."""
    rng = np.random.default_rng(91 + dist)
    d, cap, B = 2, 64, 2
    n = d * cap
    theta = np.zeros((B, n, n), dtype=np.complex128)
    svs = 10.0 ** (-10.0 * np.arange(n) / (n - 1))
    for b in range(B):
        u = np.linalg.qr(crand(rng, n, n))[0]
        v = np.linalg.qr(crand(rng, n, n))[0]
        theta[b] = (u * svs) @ v.conj().T
    chi = np.full(B, cap, dtype=np.int32)
    capM = n
    left, right, keep, spec, sweeps = svd_split_gpu(lib, theta, d, cap, cap, capM, dist, 0, 1e-26, 0, 2, chi, chi, qr=True)
    for b in range(B):
        assert keep[b] == n
        assert np.allclose(spec[b, :n] / svs, 1.0, atol=2e-6)  # small values: absolute accuracy eps * sigma_max
        assert np.allclose(spec[b, :40] / svs[:40], 1.0, atol=1e-10)
        L_ = left[b].reshape(n, capM)
        R_ = right[b].transpose(1, 0, 2).reshape(capM, n)
        assert np.allclose(L_ @ R_, theta[b], atol=1e-13)
        iso = L_ if dist == 0 else R_.conj().T
        assert np.allclose(iso.conj().T @ iso, np.eye(n), atol=1e-11)


# ---- kernel-level exports of the sweep against the REFERENCE's own outputs (tests/golden/kernels.npz, tools/make_golden.py) ----
def _kernel_engine(B=2):
    from oracle import tjm_oracle as o  # checker only: MPO builder for an engine of Dmax = 3, capacities up to 8
    from yaqs_amd.engine import BatchEngine

    return BatchEngine(8, 8, B, o.ising_mpo(8, 1.0, 0.5))


_KEEP = []  # device operands stay referenced until the test module is done (a temporary's memory would be reused mid-call)


def _slots(a, B=2):
    """Slot 0 = the fixture operand, slot 1 = a scaled copy (checks the batch strides)."""
    t = dev(np.stack([a * (1.0 + 0.5 * k) for k in range(B)]))
    _KEEP.append(t)
    return t


def _same(a, B=2):
    t = dev(np.stack([a] * B))
    _KEEP.append(t)
    return t


def test_project_site_env_updates_and_project_bond_match_reference_outputs(lib):
    """tjm_heff_apply / tjm_env_update / tjm_project_bond on the reference's operands (primitives.py:77-226), tolerance 1e-11."""
    from conftest import GOLDEN
    from yaqs_amd._lib import check

    g = np.load(os.path.join(GOLDEN, "kernels.npz"))
    e = _kernel_engine()
    H = lambda a: np.ascontiguousarray(a, dtype=np.complex128)  # noqa: E731
    # two-site project_site: x = merge (4, 5, 4), L (5, 3, 5), R (4, 3, 4), merged MPO (4, 4, 3, 3)
    x, L, R, W2 = g["merge"], g["L"], g["R"], H(g["merge_mpo"])
    y = torch.zeros((2,) + x.shape, dtype=torch.complex128, device=DEV)
    check(lib.tjm_heff_apply(e.h, 2, 5, 4, 3, 3, _slots(x).data_ptr(), _same(L).data_ptr(), _same(R).data_ptr(),
                             W2.ctypes.data, y.data_ptr(), 2), "heff2")
    got = y.cpu().numpy()
    assert np.allclose(got[0], g["project_site_2"], atol=1e-11) and np.allclose(got[1], 1.5 * g["project_site_2"], atol=1e-11)
    # one-site project_site: x = A (2, 5, 6), L (5, 3, 5), R1 (6, 3, 6), W1 (2, 2, 3, 3)
    a, R1, W1 = g["A"], g["R1"], H(g["W1"])
    y = torch.zeros((2,) + a.shape, dtype=torch.complex128, device=DEV)
    check(lib.tjm_heff_apply(e.h, 1, 5, 6, 3, 3, _slots(a).data_ptr(), _same(L).data_ptr(), _same(R1).data_ptr(),
                             W1.ctypes.data, y.data_ptr(), 2), "heff1")
    got = y.cpu().numpy()
    assert np.allclose(got[0], g["project_site_1"], atol=1e-11) and np.allclose(got[1], 1.5 * g["project_site_1"], atol=1e-11)
    # environments: left from (A, W1, L) -> (6, 3, 6); right from (B, W2, R) -> (6, 3, 6); quadratic in the site tensor
    out = torch.zeros((2, 6, 3, 6), dtype=torch.complex128, device=DEV)
    check(lib.tjm_env_update(e.h, 1, 5, 6, 3, 3, _slots(a).data_ptr(), _same(L).data_ptr(), W1.ctypes.data, out.data_ptr(), 2), "envL")
    got = out.cpu().numpy()
    assert np.allclose(got[0], g["env_left"], atol=1e-11) and np.allclose(got[1], 2.25 * g["env_left"], atol=1e-11)
    b, Wb = g["B"], H(g["W2"])
    out = torch.zeros((2, 6, 3, 6), dtype=torch.complex128, device=DEV)
    check(lib.tjm_env_update(e.h, 0, 6, 4, 3, 3, _slots(b).data_ptr(), _same(R).data_ptr(), Wb.ctypes.data, out.data_ptr(), 2), "envR")
    got = out.cpu().numpy()
    assert np.allclose(got[0], g["env_right"], atol=1e-11) and np.allclose(got[1], 2.25 * g["env_right"], atol=1e-11)
    # project_bond: C (5, 4), LB (5, 3, 5), R (4, 3, 4)
    c, LB = g["C"], g["LB"]
    y = torch.zeros((2, 5, 4), dtype=torch.complex128, device=DEV)
    check(lib.tjm_project_bond(e.h, 5, 4, 3, _slots(c).data_ptr(), _same(LB).data_ptr(), _same(R).data_ptr(), y.data_ptr(), 2),
          "project_bond")
    got = y.cpu().numpy()
    assert np.allclose(got[0], g["project_bond"], atol=1e-11) and np.allclose(got[1], 1.5 * g["project_bond"], atol=1e-11)
    e.close()


@pytest.mark.parametrize("D", [8, 20, 28])
def test_heff_apply_with_wide_mpo_bonds(lib, D):
    """project_site (primitives.py:180-204) on a qubit pair with MPO bonds of 8, 20 and 28: the merged operator of the MPO stage is
    (4 D)^2 complex numbers - 16 KiB (default LDS), 100 KiB (needs the kernel's LDS limit raised) and 196 KiB (more than a CU has: read
    through the caches).  Against the oracle's contraction."""
    from oracle import tjm_oracle as o  # checker only
    from yaqs_amd._lib import check
    from yaqs_amd.engine import BatchEngine

    rng = np.random.default_rng(D)
    Ls, chi, B = 4, 6, 2
    mpo = [crand(rng, 2, 2, 1 if i == 0 else D, 1 if i == Ls - 1 else D) for i in range(Ls)]
    e = BatchEngine(Ls, chi, B, mpo)
    ca, cb = 4, 4
    x = crand(rng, 4, ca, cb)
    lenv, renv = crand(rng, ca, D, ca), crand(rng, cb, D, cb)
    w2 = np.ascontiguousarray(o.merge_mpo_tensors(mpo[1], mpo[2]), dtype=np.complex128)  # (4, 4, D, D)
    y = torch.zeros((B, 4, ca, cb), dtype=torch.complex128, device=DEV)
    check(lib.tjm_heff_apply(e.h, 2, ca, cb, D, D, _slots(x).data_ptr(), _same(lenv).data_ptr(), _same(renv).data_ptr(),
                             w2.ctypes.data, y.data_ptr(), B), "heff2")
    _sync()
    ref = o.project_site(lenv, renv, w2, x)
    got = y.cpu().numpy()
    scale = np.abs(ref).max()
    assert np.allclose(got[0], ref, atol=1e-12 * scale) and np.allclose(got[1], 1.5 * ref, atol=1e-12 * scale)
    e.close()


def test_lanczos_expm_matches_reference_outputs(lib):
    """tjm_lanczos_expm = update_site (expm_krylov o project_site) on the reference's two-site block of a 6-site chain at both
    tolerances of the fixture (the fused small-bond kernel serves blocks of this size)."""
    from conftest import GOLDEN
    from oracle import tjm_oracle as o  # checker only: merge of the fixture's tensors
    from yaqs_amd._lib import check

    g = np.load(os.path.join(GOLDEN, "kernels.npz"))
    e = _kernel_engine()
    th = o.merge_two_site(g["mps0"], g["mps1"])           # (4, 1, 4)
    w2 = np.ascontiguousarray(o.merge_mpo_tensors(g["mpo0"], g["mpo1"]), dtype=np.complex128)  # (4, 4, 1, 3)
    l0 = np.ones((1, 1, 1), dtype=np.complex128)
    rb = g["renv1"]                                        # (4, 3, 4)
    for tol in (1e-4, 1e-12):
        y = torch.zeros((2, 4, 1, 4), dtype=torch.complex128, device=DEV)
        check(lib.tjm_lanczos_expm(e.h, 2, 1, 4, 1, 3, _slots(th).data_ptr(), _same(l0).data_ptr(), _same(rb).data_ptr(),
                                   w2.ctypes.data, 0.05, tol, y.data_ptr(), 2, None), "lanczos")
        got = y.cpu().numpy()
        ref = g[f"krylov_site2_tol{tol:g}"]
        assert np.allclose(got[0], ref, atol=1e-11), (tol, np.abs(got[0] - ref).max())
        assert np.allclose(got[1], 1.5 * ref, atol=1e-11)
    e.close()


@pytest.mark.parametrize("bond", [24, 64])
@pytest.mark.parametrize("nsites", [1, 2])
def test_lanczos_expm_general_path_matches_oracle(lib, nsites, bond):
    """The same export at bond 24 (block of 1152 / 2304 entries: MFMA GEMMs + Lanczos vector kernels, one host check per iteration)
    against the oracle's update_site (pinned to the reference by the fixture above), with the adaptive stop at 1e-4 and 1e-12; and
    at bond 64, where the last GEMM of the H_eff apply runs on the tiled kernel and delivers the Lanczos coefficient <v, H v> in its
    epilogue (round 5: per-tile partial sums instead of a dot-product pass)."""
    from oracle import tjm_oracle as o  # checker only
    from yaqs_amd._lib import check
    from yaqs_amd.engine import BatchEngine

    rng = np.random.default_rng(11 + nsites + bond)
    ca = cb = bond
    D, P = 3, 2 ** nsites
    mpo = o.ising_mpo(8, 1.0, 0.5)
    w = mpo[3] if nsites == 1 else o.merge_mpo_tensors(mpo[3], mpo[4])
    herm = lambda m: m + m.conj().transpose(2, 1, 0)  # noqa: E731  (environments of a Hermitian MPO are Hermitian in their outer legs)
    Lenv, Renv = herm(crand(rng, ca, D, ca)), herm(crand(rng, cb, D, cb))
    x = crand(rng, P, ca, cb)
    x /= np.linalg.norm(x)
    e = BatchEngine(12, max(32, bond), 2, o.ising_mpo(12, 1.0, 0.5))
    wh = np.ascontiguousarray(w, dtype=np.complex128)
    for tol in (1e-4, 1e-12):
        y = torch.zeros((2, P, ca, cb), dtype=torch.complex128, device=DEV)
        mv = C.c_int64(0)
        check(lib.tjm_lanczos_expm(e.h, nsites, ca, cb, D, D, _slots(x).data_ptr(), _same(Lenv).data_ptr(),
                                   _same(Renv).data_ptr(), wh.ctypes.data, 0.02, tol, y.data_ptr(), 2, C.byref(mv)), "lanczos")
        ref = o.update_site(Lenv, Renv, w, x, 0.02, tol)
        got = y.cpu().numpy()
        assert mv.value >= 2
        assert np.allclose(got[0], ref, atol=1e-10), (tol, np.abs(got[0] - ref).max())
        assert np.allclose(got[1], 1.5 * ref, atol=1e-10)
    e.close()


@pytest.mark.parametrize("bond", [64, 128])
@pytest.mark.parametrize("nsites", [1, 2])
def test_lanczos_expm_with_identity_channels_takes_the_direct_form(lib, nsites, bond):
    """Environments as a canonical chain gives them - L[:, 0, :] = 1 and R[:, D - 1, :] = 1 for the Ising MPO - so that the Krylov call
    certifies both identity channels; the MPO rows of the other left channels are monomial and no entry couples two non-identity
    channels, so the H_eff apply runs in its direct form (round 5: the fused stage kernel writes only y, the last product reads
    x[perm] x coef instead of T2; GemmDesc::b_perm / coef on zgemm4_kernel).  Checked against the oracle's update_site, which knows
    nothing of identity channels; the engine's counter says the direct form served the applies (fp64 library on the device only)."""
    from oracle import tjm_oracle as o  # checker only
    from yaqs_amd._lib import check
    from yaqs_amd.engine import BatchEngine

    rng = np.random.default_rng(211 + nsites + bond)
    ca = cb = bond
    D, P = 3, 2 ** nsites
    mpo = o.ising_mpo(8, 1.0, 0.5)
    w = mpo[3] if nsites == 1 else o.merge_mpo_tensors(mpo[3], mpo[4])
    herm = lambda m: m + m.conj().transpose(2, 1, 0)  # noqa: E731
    Lenv, Renv = herm(crand(rng, ca, D, ca)), herm(crand(rng, cb, D, cb))
    Lenv[:, 0, :] = np.eye(ca)
    Renv[:, D - 1, :] = np.eye(cb)
    x = crand(rng, P, ca, cb)
    x /= np.linalg.norm(x)
    e = BatchEngine(16, max(32, bond), 2, o.ising_mpo(16, 1.0, 0.5))  # (16 sites: the storage of the middle bonds reaches 128)
    wh = np.ascontiguousarray(w, dtype=np.complex128)
    for tol in (1e-4, 1e-12):
        y = torch.zeros((2, P, ca, cb), dtype=torch.complex128, device=DEV)
        mv = C.c_int64(0)
        check(lib.tjm_lanczos_expm(e.h, nsites, ca, cb, D, D, _slots(x).data_ptr(), _same(Lenv).data_ptr(),
                                   _same(Renv).data_ptr(), wh.ctypes.data, 0.02, tol, y.data_ptr(), 2, C.byref(mv)), "lanczos")
        ref = o.update_site(Lenv, Renv, w, x, 0.02, tol)
        got = y.cpu().numpy()
        assert np.allclose(got[0], ref, atol=1e-10), (tol, np.abs(got[0] - ref).max())
        assert np.allclose(got[1], 1.5 * ref, atol=1e-10)
    st = e.stats()
    assert st["identity_channels"] >= 2
    if not os.environ.get("TJM_NO_DIRECT_HEFF") and not os.environ.get("TJM_GEMM_16X16"):
        assert st["direct_applies"] >= 2, st
    e.close()


def test_center_shifts_are_gauge_moves_with_isometric_factors(lib):
    """tjm_engine_center_shift (shift_orthogonality_center_right / _left, mps.py:719-788), QR and SVD flavour: the state vector is
    unchanged, the tensor left behind is an isometry, the bond obeys the thin-QR rule; at bonds that take the general Householder /
    Jacobi kernels (capacity 32) and at small ones (fused one-wavefront kernels)."""
    from oracle import tjm_oracle as o  # checker only
    from yaqs_amd._lib import check
    from yaqs_amd.engine import BatchEngine

    for L, chi in ((8, 4), (12, 32)):
        rng = np.random.default_rng(L)
        st = o.MPSState.haar(L, chi, rng)
        st.normalize("B")
        v0 = st.to_vec() if L <= 12 else None
        e = BatchEngine(L, chi, 2, o.ising_mpo(L, 1.0, 0.5))
        e.set_params(dt=0.1, svd_threshold=1e-12, max_bond_dim=chi)
        for use_svd in (0, 1):
            e.load_state(st.tensors)
            for i in range(L - 1):
                check(lib.tjm_engine_center_shift(e.h, 0, i, 1, use_svd), "shift right")
                t = e.export_state(1)[i]
                m = t.reshape(-1, t.shape[2])
                assert np.allclose(m.conj().T @ m, np.eye(m.shape[1]), atol=1e-12), (L, use_svd, i)
            out = e.export_state(0)
            assert np.allclose(abs(np.vdot(o.MPSState(out, L - 1).to_vec(), v0)), 1.0, atol=1e-11)
            for i in range(L - 1, 0, -1):
                check(lib.tjm_engine_center_shift(e.h, 0, i, -1, use_svd), "shift left")
                t = e.export_state(0)[i]
                m = t.transpose(1, 0, 2).reshape(t.shape[1], -1)
                assert np.allclose(m @ m.conj().T, np.eye(m.shape[0]), atol=1e-12), (L, use_svd, i)
            out = e.export_state(1)
            assert np.allclose(abs(np.vdot(o.MPSState(out, 0).to_vec(), v0)), 1.0, atol=1e-11)
            assert [t.shape[2] for t in out] == [t.shape[2] for t in st.tensors]
        e.close()


def test_jump_weights_match_the_oracle_distribution(lib):
    """tjm_engine_jump_weights = create_probability_distribution (stochastic_process.py:139-187): channel order and normalised
    probabilities for one-site Pauli / non-Pauli, adjacent two-site non-Pauli and long-range Pauli processes."""
    from oracle import tjm_oracle as o  # checker only
    from yaqs_amd._lib import check
    from yaqs_amd.api import is_pauli
    from yaqs_amd.engine import BatchEngine

    L, chi = 6, 8
    X, Zm = o.PAULI["x"], o.PAULI["z"]
    low = np.array([[0, 1], [0, 0]], dtype=np.complex128)
    procs = [o.make_process("lowering", [2], 0.3), o.make_process("pauli_z", [0], 0.2), o.make_process("pauli_x", [2], 0.1),
             o.make_process("pair", [3, 4], 0.25, matrix=np.kron(low, low)), o.make_process("lr", [1, 5], 0.15, factors=(X, Zm)),
             o.make_process("raising", [5], 0.05)]
    rng = np.random.default_rng(4)
    st = o.MPSState.haar(L, chi, rng)
    st.normalize("B")
    for t in st.tensors[:1]:
        t *= 0.9  # an unnormalised centre, as after dissipation
    e = BatchEngine(L, chi, 2, o.ising_mpo(L, 1.0, 0.5))
    e.set_params(dt=0.1, svd_threshold=1e-12, max_bond_dim=chi)
    e.set_noise(procs, [is_pauli(q) for q in procs])
    e.load_state(st.tensors)
    order = np.zeros(len(procs), dtype=np.int32)
    w = np.zeros((2, len(procs)))
    n = C.c_int32(0)
    check(lib.tjm_engine_jump_weights(e.h, 0, 0.1, order.ctypes.data, w.ctypes.data, C.byref(n)), "jump_weights")
    assert n.value == len(procs)
    chosen, probs = o.jump_distribution(o.MPSState([x.copy() for x in st.tensors], 0), procs, 0.1, o.Params(dt=0.1, max_bond_dim=chi, svd_threshold=1e-12))
    assert [procs[k]["name"] for k in order] == [c["name"] for c in chosen]
    assert np.allclose(w[0] / w[0].sum(), probs, atol=1e-12) and np.allclose(w[1], w[0], atol=1e-14)
    e.close()


# ---- bonds up to 512 (BASELINE config 5's max_bond_dim): matrices of up to 1024 x 1024 --------------------------------------
@pytest.mark.parametrize("capL,capR,dist", [(512, 512, 0), (512, 512, 1), (256, 512, 0), (512, 256, 1)])
def test_svd_split_up_to_1024_matches_oracle(lib, capL, capR, dist):
    """split_two_site (decompositions.py:105-185) of a (2 capL) x (2 capR) theta with a spectrum graded over ten decades against the
    oracle: keep, singular values, the reconstructed truncated theta, exact isometry - square 1024 x 1024 and the rectangular
    512 x 1024 / 1024 x 512 shapes of the chain positions where the outer bonds differ (embedded in the square factorisation)."""
    from oracle import tjm_oracle as o

    rng = np.random.default_rng(capL + 2 * capR + dist)
    d, B = 2, 1
    m, n = d * capL, d * capR
    k0 = min(m, n)
    u = np.linalg.qr(crand(rng, m, k0))[0]
    v = np.linalg.qr(crand(rng, n, k0))[0]
    sv = 10.0 ** (-10.0 * np.arange(k0) / k0) * (1.0 + 0.1 * rng.random(k0))
    theta = ((u * sv) @ v.conj().T)[None]
    capM = 512
    chiL, chiR = np.full(B, capL, dtype=np.int32), np.full(B, capR, dtype=np.int32)
    thr, maxb = 1e-14, 400
    left, right, keep, spec, sweeps = svd_split_gpu(lib, theta, d, capL, capR, capM, dist, 0, thr, maxb, 2, chiL, chiR, qr=True)
    merged = theta[0].reshape(d, capL, d, capR).transpose(0, 2, 1, 3).reshape(d * d, capL, capR)
    l_ref, r_ref, s_ref = o.split_two_site(merged, [d, d], svd_distribution="right" if dist == 0 else "left", trunc_mode="discarded_weight",
                                            threshold=thr, max_bond_dim=maxb, min_keep=2, return_spectrum=True)
    k = l_ref.shape[2]
    assert keep[0] == k, (keep[0], k)
    assert np.allclose(spec[0, :k], s_ref[:k], rtol=1e-10, atol=1e-14)
    got = o.merge_two_site(left[0][:, :, :k], right[0][:, :k, :])
    assert np.allclose(got, o.merge_two_site(l_ref, r_ref), atol=1e-11)
    assert np.all(left[0][:, :, k:] == 0) and np.all(right[0][:, k:, :] == 0)
    iso = left[0][:, :, :k].reshape(m, k) if dist == 0 else right[0][:, :k, :].transpose(1, 0, 2).reshape(k, n).conj().T
    assert np.allclose(iso.conj().T @ iso, np.eye(k), atol=1e-12)


@pytest.mark.parametrize("capL,capR", [(4, 4), (16, 16), (8, 3), (48, 48), (128, 128)])
def test_svd_split_sqrt_distribution_matches_oracle(lib, capL, capR):
    """distribution 2 = "sqrt" of split_two_site (decompositions.py:166-171; _sync_bond_dim of the dynamic sweep): sqrt(S) in both
    factors, capped and thresholded like the oracle's; the merged pair, keep and the balance of the two factors."""
    from oracle import tjm_oracle as o

    rng = np.random.default_rng(capL * 7 + capR)
    d, B = 2, 3
    capM = min(d * capL, d * capR)
    theta = crand(rng, B, d * capL, d * capR) / np.sqrt(d * capL * d * capR)
    theta[1] *= 1e-2
    chiL, chiR = np.full(B, capL, dtype=np.int32), np.full(B, capR, dtype=np.int32)
    thr, maxb = 1e-8, max(1, capM // 2)
    left, right, keep, _, _ = svd_split_gpu(lib, theta, d, capL, capR, capM, 2, 0, thr, maxb, 1, chiL, chiR, qr=False)
    for b in range(B):
        merged = theta[b].reshape(d, capL, d, capR).transpose(0, 2, 1, 3).reshape(d * d, capL, capR)
        l_ref, r_ref = o.split_two_site(merged, [d, d], svd_distribution="sqrt", trunc_mode="discarded_weight", threshold=thr, max_bond_dim=maxb, min_keep=1)
        k = l_ref.shape[2]
        assert keep[b] == k
        assert np.allclose(o.merge_two_site(left[b][:, :, :k], right[b][:, :k, :]), o.merge_two_site(l_ref, r_ref), atol=1e-11)
        gl_ = left[b][:, :, :k].reshape(d * capL, k)
        gr_ = right[b][:, :k, :].transpose(1, 0, 2).reshape(k, d * capR)
        assert np.allclose(np.linalg.norm(gl_, axis=0), np.linalg.norm(gr_, axis=1), rtol=1e-9)   # sqrt(S) on both sides
        assert np.all(left[b][:, :, k:] == 0) and np.all(right[b][:, k:, :] == 0)


@pytest.fixture(scope="module")
def lib32():
    """libtjm_hip_f32.so (the complex64 build): its kernel-level exports take complex64 / float32 device arrays."""
    from yaqs_amd import _lib

    if SIM:
        from simengine import load_sim

        return load_sim("complex64")
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return _lib.load("complex64")


@pytest.mark.parametrize("M,N,K,conjA,conjB", [(64, 64, 16, 0, 0), (70, 50, 37, 1, 0), (128, 96, 64, 0, 1), (3, 5, 2, 1, 1), (17, 200, 33, 0, 0), (130, 9, 4, 1, 1),
                                               (31, 31, 127, 0, 1)])
def test_gemm_of_the_complex64_library(lib32, M, N, K, conjA, conjB):
    """The batched complex GEMM of the complex64 build (v_mfma_f32_16x16x4_f32, whose result registers map to rows 4 (l >> 4) + v
    instead of the f64 form's (l >> 4) + 4 v): tiled kernel and one-tile-per-wavefront kernel, conjugated operands, odd shapes."""
    rng = np.random.default_rng(M * 1000 + N + K)
    nb = 3
    a, b = crand(rng, nb, M, K).astype(np.complex64), crand(rng, nb, K, N).astype(np.complex64)
    A, B = dev(a), dev(b)
    Cc = torch.zeros((nb, M, N), dtype=torch.complex64, device=DEV)
    run_gemm(lib32, A=A.data_ptr(), B=B.data_ptr(), C=Cc.data_ptr(), M=M, N=N, K=K, a_rs=K, a_cs=1, b_rs=N, b_cs=1, c_rs=N,
             nb0=nb, a_b0=M * K, b_b0=K * N, c_b0=M * N, conjA=conjA, conjB=conjB)
    a64, b64 = a.astype(np.complex128), b.astype(np.complex128)
    ref = np.einsum("bmk,bkn->bmn", a64.conj() if conjA else a64, b64.conj() if conjB else b64)
    assert np.allclose(Cc.cpu().numpy(), ref, atol=3e-6 * K)


@pytest.mark.parametrize("capL,capR,qr", [(8, 8, False), (32, 32, False), (40, 32, True), (64, 64, True), (96, 96, True), (72, 80, True), (128, 128, True),
                                          (128, 128, False), (256, 256, True), (160, 160, True), (192, 256, True), (256, 256, False), (320, 320, True), (384, 384, True), (512, 512, True)])
def test_svd_split_of_the_complex64_library(lib32, capL, capR, qr):
    """Two-site split of the complex64 build (fused small kernel, LDS-resident and tiled Jacobi with fp32 tolerances, Householder
    panels): singular values to 1e-5 of the largest, isometric left factor, reconstruction of theta to fp32 accuracy.  Round 6: the
    sizes of BASELINE's configs 3 and 5 - 512 x 512 (four columns per wavefront, the grouped three-rounds-per-load schedule), 320 and
    640 rows (tile kernels between the group sizes), 1024 x 1024 (sixteen row groups per column in registers)."""
    from yaqs_amd._lib import check

    rng = np.random.default_rng(capL * 10 + capR)
    d, B = 2, 2
    capM = min(d * capL, d * capR)
    theta = crand(rng, B, d * capL, d * capR).astype(np.complex64)
    th = dev(theta)
    left = torch.zeros((B, d, capL, capM), dtype=torch.complex64, device=DEV)
    right = torch.zeros((B, d, capM, capR), dtype=torch.complex64, device=DEV)
    chi = dev(np.stack([np.full(B, capL), np.full(B, capR), np.zeros(B)], axis=1).astype(np.int32))
    spec_ld = d * max(capL, capR)
    spec = torch.zeros((B, spec_ld), dtype=torch.float32, device=DEV)
    nbytes = (lib32.tjm_svd_qr_workspace_bytes if qr else lib32.tjm_svd_workspace_bytes)(d * max(capL, capR), B)
    work = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
    sweeps = C.c_int32(0)
    fn = lib32.tjm_svd_split_qr if qr else lib32.tjm_svd_split
    check(fn(th.data_ptr(), B, d, capL, capR, capM, left.data_ptr(), right.data_ptr(), 0, 0, 0.0, capM, 1, chi.data_ptr(), spec.data_ptr(), spec_ld,
             work.data_ptr(), nbytes, C.byref(sweeps), None), "svd_split")
    _sync()
    # fp32 arithmetic: values to ~N eps of the largest (np.allclose's atol + rtol = 2e-5 of the tests before round 6, scaled with the size
    # above 256 rows: measured 1.9e-5 at 512, 2.6e-5 at 640 with the QR preconditioner).  Isometry: 2e-5 behind the QR preconditioner
    # (the isometric factor is reflectors times normalised columns); the plain split accumulates its ~13 sweeps x (N - 1) rotations per
    # column in fp32: 2.7e-5 at 256 rows, 4.5e-5 at 512 (measured) - the engine takes that path at these sizes only for listed trajectories
    N_ = d * max(capL, capR)
    spec_tol = 2e-5 * max(1.0, N_ / 256.0)
    iso_tol = 2e-5 * max(1.0, N_ / 256.0) if qr else 2e-5 * max(1.0, N_ / 128.0)  # (rectangular 384 x 512 behind the QR: 2.8e-5; square 512: 2e-6)
    for b in range(B):
        s_ref = np.linalg.svd(theta[b].astype(np.complex128), compute_uv=False)
        k = int(chi.cpu().numpy()[b, 2])
        assert k == capM
        got = spec.cpu().numpy()[b, :k]
        spec_err = np.abs(got - s_ref[:k]).max() / s_ref[0]
        assert spec_err <= spec_tol, (b, spec_err)
        lf = left.cpu().numpy()[b].astype(np.complex128).reshape(d * capL, capM)
        rf = right.cpu().numpy()[b].astype(np.complex128).transpose(1, 0, 2).reshape(capM, d * capR)
        iso_err = np.abs(lf.conj().T @ lf - np.eye(capM)).max()
        assert iso_err <= iso_tol, (b, iso_err)
        print(f"[c64 split {capL}x{capR} qr={qr}] b={b} spec_err {spec_err:.2e} iso_err {iso_err:.2e}")
        # left[(s,a),k] right[k,(t,c)] = theta[(s,a),(t,c)] with rows (s, a) and columns (t, c)
        assert np.allclose(lf @ rf, theta[b].astype(np.complex128), atol=2e-5 * s_ref[0])


# ---- the mixed-precision two-site split in the mode the ENGINE calls it in (no spectrum buffer) -----------------------------------
def _mixed_matrices(rng, n):
    """(name, theta, gap at the cut?) for square n x n thetas with the new bond capped at n / 2: full numerical rank without a gap at
    the cut (a Gaussian matrix), rank n / 4 (exactly zero singular values: unit-vector completion of the complex64 basis, the 'far'
    columns), a spectrum graded over six decades, and the exponentially decaying spectrum of an evolved two-site tensor."""
    cap = n // 2
    u = np.linalg.qr(crand(rng, n, n))[0]
    v = np.linalg.qr(crand(rng, n, n))[0]
    graded = (u * np.sort(np.concatenate([np.linspace(1.0, 0.05, cap), 10.0 ** rng.uniform(-6, -2, n - cap)]))[::-1]) @ v.conj().T
    low = (crand(rng, n, n // 4) / np.sqrt(n * n / 4)) @ np.linalg.qr(crand(rng, n, n // 4))[0].conj().T
    evolved = (u * np.exp(-np.arange(n) * 16.0 / n)) @ v.conj().T
    return ["gaussian", "rank n/4", "graded", "evolved"], np.stack([crand(rng, n, n) / n, low, graded, evolved])


def _check_engine_mode_split(theta, left, right, keep, d, cap, dist, tol_trunc, tol_iso=1e-13, tol_resid=1e-12):
    """Against LAPACK: the kept count, the truncated theta (tol_trunc x sigma_0; for a matrix without a gap at the cut the subspace of
    the kept values is only defined to eps / relative gap, so the gap-free measure is checked too: ||theta - L R||_F against the
    discarded tail), an isometric factor that is isometric to rounding, exact zero padding."""
    n = d * cap
    for b in range(theta.shape[0]):
        ru, rs, rvh = np.linalg.svd(theta[b])
        tail = np.cumsum((rs ** 2)[::-1])[::-1]
        kb_ref = max(2, min(cap, int(np.sum(tail >= 1e-12))))  # discarded_weight rule (svd_utils.py:22-104), min_keep 2, max_bond cap
        kb = int(keep[b])
        assert kb == kb_ref, (b, kb, kb_ref)
        L_ = left[b].reshape(n, cap)
        R_ = right[b].transpose(1, 0, 2).reshape(cap, n)
        trunc = (ru[:, :kb] * rs[:kb]) @ rvh[:kb]
        assert np.abs(L_ @ R_ - trunc).max() <= tol_trunc[b] * rs[0], (dist, b, np.abs(L_ @ R_ - trunc).max() / rs[0])
        resid = np.linalg.norm(theta[b] - L_ @ R_)
        assert abs(resid - np.sqrt(np.sum(rs[kb:] ** 2))) <= tol_resid * rs[0], (dist, b, resid, np.sqrt(np.sum(rs[kb:] ** 2)))
        iso = L_[:, :kb] if dist == 0 else R_[:kb].conj().T
        assert np.abs(iso.conj().T @ iso - np.eye(kb)).max() <= tol_iso, (dist, b, np.abs(iso.conj().T @ iso - np.eye(kb)).max())
        assert np.all(L_[:, kb:] == 0) and np.all(R_[kb:] == 0)


def _run_engine_mode(lib, n):
    rng = np.random.default_rng(n)
    d, cap = 2, n // 2
    names, theta = _mixed_matrices(rng, n)
    chi = np.full(len(names), cap, dtype=np.int32)
    # tolerance of the truncated theta: 1e-12 where the cut sits in a gap (or behind the rank), 1e-10 for the gap-free Gaussian matrix
    # (relative gap of neighbouring values ~ 1 / n: tilt 1e-13 / gap)
    tol = [1e-10, 1e-12, 1e-12, 1e-12]
    out = (C.c_double * 10)()
    lib.tjm_svd_mixed_read(out, 1)
    for dist in (0, 1):
        left, right, keep, _, _ = svd_split_gpu(lib, theta, d, cap, cap, cap, dist, 0, 1e-12, cap, 2, chi, chi, qr=True, want_spec=False)
        _check_engine_mode_split(theta, left, right, keep, d, cap, dist, tol)
    lib.tjm_svd_mixed_read(out, 0)
    return [int(round(out[i])) for i in range(6)] + [out[8]]


@pytest.mark.parametrize("n", [256, 512, 768, 1024])
def test_mixed_split_in_the_mode_of_the_engine_matches_lapack(lib, n):
    """svd_split_mixed as Engine::split calls it - no spectrum buffer, so columns that can never be kept stay uncorrected among
    themselves, the refinement works on X and its squares run on the complex64 GEMM (tjm_svd.hip: can_skip) - at the sizes of
    BASELINE's configs 2 (256 x 256) and 3 (512 x 512), on four kinds of matrices, both distributions, against LAPACK.  The counters
    say that the mixed path SERVED the calls: two batched solves, complex64 sweeps and fp64 products counted, no trajectory handed to
    the all-fp64 split."""
    solves, c64_sweeps, f64_sweeps, fallbacks, jacobi_traj, second_polar, gemms = _run_engine_mode(lib, n)
    assert solves == 2 and fallbacks == 0, (solves, fallbacks)
    assert c64_sweeps >= 8 and gemms >= 26, (c64_sweeps, gemms)  # at least four complex64 sweeps and 13 fp64 products per solve
    assert jacobi_traj <= 2 * 4  # (the fp64 Jacobi may finish single trajectories: the final check decides, per trajectory)


@pytest.mark.skipif(SIM, reason="child processes of the GPU run")
@pytest.mark.parametrize("cap_sweeps,branch", [(4, "second_polar"), (2, "fallback")])
def test_mixed_split_branches_under_a_loosened_complex64_basis(cap_sweeps, branch):
    """The rare branches of the mixed split, forced: with the complex64 iteration cut off after four sweeps (TJM_MIXED_C64_SWEEPS,
    read once per process: a child process) the basis is orthonormal to ~1e-3 only - the polar certificate fails, the listed second
    polar step runs and the fp64 Jacobi finishes what the final check does not pass; cut off after two sweeps the second step fails
    too and the trajectories go to the all-fp64 split (one code path whatever the size of the list).  Results against LAPACK as above
    (isometry 2e-13 where the plain fp64 split with its accumulated rotations served the trajectory)."""
    import json
    import subprocess
    import sys

    code = ("import json, sys; sys.path.insert(0, %r); import test_hip_kernels as k; from yaqs_amd import _lib; "
            "r = k._run_engine_mode_loose(_lib.load(), 256); print('RESULT ' + json.dumps(r))") % os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, TJM_MIXED_C64_SWEEPS=str(cap_sweeps))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__)))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    solves, c64_sweeps, f64_sweeps, fallbacks, jacobi_traj, second_polar, gemms = r
    assert solves == 2 and c64_sweeps == 2 * cap_sweeps, r
    assert second_polar == 2, r  # both batched solves took the listed second polar step
    if branch == "second_polar":
        assert fallbacks == 0 and jacobi_traj > 0 and f64_sweeps > 0, r  # served by the mixed path, finished by the fp64 Jacobi on X
    else:
        assert fallbacks > 0, r  # handed to the all-fp64 split, per trajectory


def _run_engine_mode_loose(lib, n):
    """_run_engine_mode with the tolerances of a trajectory the all-fp64 split may have served: isometry 2e-13 (accumulated rotations),
    residual 5e-11 sigma_0 (that path stops rotating a column once it is below its noise floor, 1e-13 ||theta||_F: on the rank n / 4
    matrix the 3 n / 4 null columns are left 1e-11 sigma_0 away from orthogonal - three decades inside the 1e-8 parity bar)."""
    rng = np.random.default_rng(n)
    d, cap = 2, n // 2
    names, theta = _mixed_matrices(rng, n)
    chi = np.full(len(names), cap, dtype=np.int32)
    out = (C.c_double * 10)()
    lib.tjm_svd_mixed_read(out, 1)
    for dist in (0, 1):
        left, right, keep, _, _ = svd_split_gpu(lib, theta, d, cap, cap, cap, dist, 0, 1e-12, cap, 2, chi, chi, qr=True, want_spec=False)
        _check_engine_mode_split(theta, left, right, keep, d, cap, dist, [1e-10, 5e-11, 5e-11, 5e-11], tol_iso=2e-13, tol_resid=5e-11)
    lib.tjm_svd_mixed_read(out, 0)
    return [int(round(out[i])) for i in range(6)] + [out[8]]


@pytest.mark.skipif(SIM, reason="two host threads on two HIP streams")
def test_svd_split_is_reentrant_across_host_threads_and_streams(lib):
    """'one HIP stream per call; re-entrant per device'.  Two host threads run 256 x 256 splits (the mixed path: five host
    round trips per call, each through a pinned flag block) on two streams at the same time, several times over; every result is
    bit-identical to the serial one.  (Until round 4 all calls shared one process-global pinned block.)"""
    import threading

    d, cap = 2, 128
    n = d * cap
    jobs = []
    for seed in (11, 12):
        rng = np.random.default_rng(seed)
        u = np.linalg.qr(crand(rng, n, n))[0]
        v = np.linalg.qr(crand(rng, n, n))[0]
        evolved = (u * np.exp(-np.arange(n) * (12.0 + seed) / n)) @ v.conj().T
        jobs.append(np.stack([crand(rng, n, n) / n, evolved, crand(rng, n, n) / n, evolved.conj().T]))
    chi = np.full(4, cap, dtype=np.int32)

    def run(theta, stream=None):
        return svd_split_gpu(lib, theta, d, cap, cap, cap, 0, 0, 1e-12, cap, 2, chi, chi, qr=True, want_spec=False, stream=stream)

    serial = [run(t) for t in jobs]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for rep in range(4):
        got, errs = [None, None], []

        def work(k):
            try:
                torch.cuda.set_device(0)
                got[k] = run(jobs[k], streams[k])
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errs, errs
        for k in range(2):
            for a, b in zip(serial[k][:3], got[k][:3]):
                assert np.array_equal(a, b), (rep, k)


def test_qr_apply_launch_sampler_reports_the_preconditioner_of_the_mixed_split(lib):
    """tjm_profile_qr_apply / _read (round 6; the source of bench.py's roofline.kernels.qr_apply_complex64): with the sampler on, a
    256 x 256 split of the fp64 library - whose preconditioner is the complex64 instance - reports launches, a positive summed duration
    and the nominal flops of its block reflectors in the complex64 block and nothing in the fp64 one; switched off it reports zeros."""
    rng = np.random.default_rng(3)
    d, cap = 2, 128
    theta = np.stack([crand(rng, d * cap, d * cap) / (d * cap) for _ in range(4)])
    chi = np.full(4, cap, dtype=np.int32)
    lib.tjm_profile_qr_apply(1)
    svd_split_gpu(lib, theta, d, cap, cap, cap, 0, 0, 1e-12, cap, 2, chi, chi, qr=True, want_spec=False)
    own, c64 = np.zeros(5), np.zeros(5)
    lib.tjm_profile_qr_apply_read(own.ctypes.data, c64.ctypes.data)
    lib.tjm_profile_qr_apply(0)
    assert c64[2] > 0 and c64[3] >= c64[2] and c64[0] > 0.0 and c64[1] > 0.0 and c64[4] >= c64[1], c64
    assert own[2] == 0 and own[0] == 0.0, own
    # two QR factorisations of 256 x 256 in groups of four 16-column panels + Q x C: per matrix ~ 2 x 8 x 16 x rows x columns per panel
    assert 1e8 < c64[4] / 4 < 1e10, c64
    lib.tjm_profile_qr_apply_read(own.ctypes.data, c64.ctypes.data)
    assert c64[2] == 0 and c64[3] == 0, c64
