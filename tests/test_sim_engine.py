"""The device code on tests/hipsim: a selection of the ``-m gpu`` tests run on the CPU (TEST INFRASTRUCTURE, no GPU needed).

tests/hipsim compiles the unchanged .hip sources of yaqs_amd/csrc for the host and interprets the HIP execution model (fibres for
threads, wavefront exchange for the cross-lane and MFMA instructions; see tests/hipsim/hip/hip_runtime.h).  The tests below are the
bodies of tests/test_hip_engine.py and tests/test_hip_kernels.py, unchanged, with ``BatchEngine`` bound to that library
(tests/simengine.py).  They check what an interpreter can check - indices, strides, control flow, arithmetic, the host schedules
on the real engine code - before the code reaches an MI355X; they are not the parity tests (those are ``-m gpu``) and say nothing
about races between workgroups or speed.  The whole ``-m gpu`` suite runs the same way with ``TJM_SIM=1`` (tests/conftest.py).
"""
import importlib

import pytest

torch = pytest.importorskip("torch")


@pytest.fixture()
def on_sim(monkeypatch):
    import conftest
    import yaqs_amd.engine as engine_mod
    import yaqs_amd.tjm as tjm_mod
    from simengine import SimEngine, load_sim

    load_sim()
    monkeypatch.setattr(engine_mod, "BatchEngine", SimEngine)
    monkeypatch.setattr(tjm_mod, "BatchEngine", SimEngine)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda device=None: (4 << 30, 4 << 30))
    monkeypatch.setattr(conftest, "SIM", True)
    mods = {}
    for name in ("test_hip_engine", "test_hip_kernels", "test_hip_round2"):
        mods[name] = importlib.import_module(name)
    k = mods["test_hip_kernels"]
    monkeypatch.setattr(k, "SIM", True)
    monkeypatch.setattr(k, "DEV", "cpu")
    return mods


ENGINE_CASES = [
    "test_one_tdvp_call_matches_reference_fixture",
    "test_dissipation_and_jump_step_match_reference_fixture",
    "test_scheduled_jumps_match_reference_fixture",
    "test_digital_tebd_trajectories_match_reference_fixture",
    "test_schmidt_spectrum_through_the_front_end",
    "test_one_site_tdvp_run_with_a_pair_channel_grows_its_storage",
    "test_sample_at_and_segment_stitching_match_reference_on_the_engine",
    "test_dynamic_tdvp_matches_reference_on_the_engine",
    "test_dynamic_tdvp_cuts_an_oversized_qr_bond_back_to_the_cap",
    "test_bug_integrator_matches_reference_on_the_engine",
    "test_bose_hubbard_qudit_chains_match_reference_fixture",
    "test_long_range_gates_through_the_gate_mpo_match_reference_fixture",
    "test_mixed_local_dimensions_match_reference_fixture",
    "test_dissipation_certificate_near_the_cut_follows_the_reference_rule",
    "test_sweeps_sequenced_inside_the_library_equal_the_host_sequenced_ones",
]


@pytest.mark.parametrize("name", ENGINE_CASES)
def test_engine_case_on_the_simulated_device(on_sim, name):
    mod = on_sim["test_hip_round2"] if hasattr(on_sim["test_hip_round2"], name) else on_sim["test_hip_engine"]
    getattr(mod, name)()


def test_complex64_build_on_the_simulated_device(on_sim, monkeypatch):
    """libtjm_hip_f32.so's sources (-DTJM_F32) on the simulated device: fp32-level agreement with the fp64 oracle, equal bonds; and
    the ensemble comparison at a size an interpreter can afford."""
    on_sim["test_hip_round2"].test_complex64_engine_tracks_the_fp64_oracle()
    monkeypatch.setenv("TJM_F32_ENSEMBLE", "8")
    on_sim["test_hip_round2"].test_complex64_ensemble_means_agree_with_the_fp64_ensemble()


def test_wide_mpo_bonds_and_four_level_pairs_on_the_simulated_device(on_sim):
    """MPO stages whose operator needs more than the default 64 KiB of dynamic LDS (the interpreter aborts such a launch unless the
    kernel's limit was raised, as the device refuses it): MPO bonds of 17 on qubits, and the Fermi-Hubbard chain on four-level sites."""
    on_sim["test_hip_round2"].test_two_site_sweep_with_wide_mpo_bonds_matches_oracle(32)
    on_sim["test_hip_round2"].test_fermi_hubbard_chain_on_four_level_sites_matches_reference_fixture()


@pytest.mark.parametrize("native", [False, True])
def test_non_finite_inputs_on_the_simulated_device(on_sim, native):
    on_sim["test_hip_round2"].test_non_finite_inputs_fail_loudly_like_the_reference(native)


@pytest.mark.parametrize("d,L,chi,order", [(3, 5, 9, 1), (4, 4, 8, 2)])
def test_qudit_chains_on_the_simulated_device(on_sim, d, L, chi, order):
    on_sim["test_hip_round2"].test_qutrit_and_four_level_chains_match_oracle(d, L, chi, order)


def test_mixed_precision_split_on_the_simulated_device(on_sim):
    """The mixed-precision two-site split (both arithmetic types in one library) at the smallest size it serves, 128 x
    128: a full-rank matrix, one of rank 64 (exactly zero singular values: the unit-vector completion of the complex64 basis, the
    'far' columns of the refinement) and one graded over six decades, for both distributions - truncated reconstruction against
    LAPACK, exactly isometric factor, zero padding; the counters say that the mixed path served them, without the fp64 Jacobi unless
    a spectrum is asked for (then every pair has to be diagonal: the graded matrix goes to the Jacobi kernels for the rest)."""
    import ctypes as C

    import numpy as np

    k = on_sim["test_hip_kernels"]
    from simengine import load_sim

    lib = load_sim()
    rng = np.random.default_rng(5)
    d, cap = 2, 64
    n = d * cap
    u = np.linalg.qr(k.crand(rng, n, n))[0]
    v = np.linalg.qr(k.crand(rng, n, n))[0]
    graded = (u * np.sort(np.concatenate([np.linspace(1.0, 0.05, cap), 10.0 ** rng.uniform(-6, -2, cap)]))[::-1]) @ v.conj().T
    low = (k.crand(rng, n, cap) / np.sqrt(n * cap)) @ np.linalg.qr(k.crand(rng, n, cap))[0].conj().T
    theta = np.stack([k.crand(rng, n, n) / n, low, graded])
    chi = np.full(3, cap, dtype=np.int32)
    out = (C.c_double * 10)()
    lib.tjm_svd_mixed_read(out, 1)
    for dist in (0, 1):
        for want_spec in (False, True):
            left, right, keep, spec, sweeps = k.svd_split_gpu(lib, theta, d, cap, cap, cap, dist, 0, 1e-12, cap, 2, chi, chi, qr=True, want_spec=want_spec)
            for b in range(3):
                ru, rs, rvh = np.linalg.svd(theta[b])
                kb = int(keep[b])
                assert kb == (cap if b != 1 else 64)
                L_ = left[b].reshape(n, cap)
                R_ = right[b].transpose(1, 0, 2).reshape(cap, n)
                trunc = (ru[:, :kb] * rs[:kb]) @ rvh[:kb]
                assert np.allclose(L_ @ R_, trunc, atol=1e-13 * max(1.0, rs[0])), (dist, want_spec, b, np.abs(L_ @ R_ - trunc).max())
                iso = L_[:, :kb] if dist == 0 else R_[:kb].conj().T
                assert np.allclose(iso.conj().T @ iso, np.eye(kb), atol=1e-13), (dist, want_spec, b)
                assert np.all(L_[:, kb:] == 0) and np.all(R_[kb:] == 0)
                if want_spec:
                    assert np.allclose(spec[b, :n], rs, rtol=0, atol=1e-12 * rs[0]), (dist, b, np.abs(spec[b, :n] - rs).max(), int(np.abs(spec[b, :n] - rs).argmax()))  # numerically null columns stay at the 1e-13 noise floor
    lib.tjm_svd_mixed_read(out, 0)
    assert out[0] == 4 and out[3] == 0, list(out)  # four batched splits served, none sent back to the fp64 path
    assert out[1] > 0 and out[8] > 0               # complex64 sweeps and fp64 GEMMs were counted


def test_mixed_precision_split_at_256_in_the_mode_of_the_engine_on_the_simulated_device(on_sim):
    """The size of the headline (256 x 256, no spectrum buffer): the complex64 phase runs the three-rounds-per-load Jacobi kernel
    (jacobi_quad64_kernel: the AG(2,4) schedule of block quads and of the columns inside a block, the block exchange between the
    rounds) and the grouped block reflectors (qr_block_apply_multi_kernel) - the body of the GPU test
    test_mixed_split_in_the_mode_of_the_engine_matches_lapack on the interpreter: LAPACK, isometry, padding, counters."""
    k = on_sim["test_hip_kernels"]
    from simengine import load_sim

    solves, c64_sweeps, f64_sweeps, fallbacks, jacobi_traj, second_polar, gemms = k._run_engine_mode(load_sim(), 256)
    assert solves == 2 and fallbacks == 0 and c64_sweeps >= 8 and gemms >= 26


@pytest.mark.parametrize("switch", ["TJM_NO_QUAD_TILE", "TJM_MIXED_UPDATE_V", "TJM_MIXED_NO_SKIP"])
def test_mixed_precision_split_under_its_switches(switch):
    """The A/B switches of the mixed split name code that is otherwise not run any more (the two-column tile kernel in its complex64
    instance, the refinement rounds on the basis, the refinement without the never-kept columns left out).  They are read once per
    process: the test above once more in a child process with the switch set."""
    import os
    import subprocess
    import sys

    env = dict(os.environ, **{switch: "1"})
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k", "mixed_precision_split_on_the_simulated_device"],
                         env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__)))
    assert out.returncode == 0 and "1 passed" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


def test_complex64_split_at_512_rows_on_the_simulated_device(on_sim):
    """Round 6: the two-site split of chi = 256 (BASELINE config 3) in the complex64 library - a 512 x 512 matrix runs the grouped
    schedule of jacobi_quad64_kernel<8> on the interpreter (five plane launches inside the two groups of 16 blocks, eight cross
    launches of two rounds between them, in-block pairs riding along, the block exchange between the rounds, the hand-overs of the J
    quads) behind the two Householder factorisations: singular values, isometry and reconstruction at fp32 accuracy.  One matrix: about
    a minute of interpreter time."""
    import ctypes as C

    import numpy as np

    k = on_sim["test_hip_kernels"]
    from simengine import load_sim
    from yaqs_amd._lib import check

    lib32 = load_sim("complex64")
    rng = np.random.default_rng(2560 + 256)
    d, cap, B = 2, 256, 1
    n = d * cap
    theta = k.crand(rng, B, n, n).astype(np.complex64)
    left = np.zeros((B, d, cap, n), dtype=np.complex64)
    right = np.zeros((B, d, n, cap), dtype=np.complex64)
    chi = np.stack([np.full(B, cap), np.full(B, cap), np.zeros(B)], axis=1).astype(np.int32)
    spec = np.zeros((B, n), dtype=np.float32)
    nbytes = lib32.tjm_svd_qr_workspace_bytes(n, B)
    work = np.zeros(nbytes, dtype=np.uint8)
    sweeps = C.c_int32(0)
    check(lib32.tjm_svd_split_qr(theta.ctypes.data, B, d, cap, cap, n, left.ctypes.data, right.ctypes.data, 0, 0, 0.0, n, 1, chi.ctypes.data,
                                 spec.ctypes.data, n, work.ctypes.data, nbytes, C.byref(sweeps), None), "svd_split_qr")
    s_ref = np.linalg.svd(theta[0].astype(np.complex128), compute_uv=False)
    assert int(chi[0, 2]) == n
    assert np.abs(spec[0] - s_ref).max() <= 4e-5 * s_ref[0]
    lf = left[0].astype(np.complex128).reshape(n, n)
    rf = right[0].astype(np.complex128).transpose(1, 0, 2).reshape(n, n)
    assert np.abs(lf.conj().T @ lf - np.eye(n)).max() <= 2e-5
    assert np.abs(lf @ rf - theta[0]).max() <= 2e-5 * s_ref[0]
