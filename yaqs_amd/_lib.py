"""ctypes binding of libtjm_hip.so (the C ABI in include/tjm_hip.h).

The HIP library is the product path: loading raises if the shared object is missing -
there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtjm_hip.so")
# the complex64 variant: the same sources and the same C ABI (host arrays stay complex128 / float64 and are converted at the boundary),
# arithmetic and device storage in fp32 (yaqs_amd/csrc/tjm_common.h: -DTJM_F32)
LIB_PATH_F32 = os.path.join(_HERE, "libtjm_hip_f32.so")
DTYPES = ("complex128", "complex64")

ERRORS = {
    -1: ValueError,
    -2: RuntimeError,
    -3: MemoryError,
    -4: NotImplementedError,
    -5: ValueError,
    -6: RuntimeError,
    -7: AssertionError,
}


class TjmError(RuntimeError):
    pass


class CapacityError(TjmError):
    """A truncation asked for a bond beyond the engine's chi_max (TJM_ERR_CAPACITY): continue (or re-run) on a larger engine.

    ``resume`` = (time step, phase) and ``rng_pos`` are set by ``BatchEngine.run``: set 0 of the engine then holds the states at
    the start of that step; ``resume[0] == 0`` (or None) means: start again from the initial state."""

    resume = None
    rng_pos = None


ERRORS[-8] = CapacityError


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("a_rs", C.c_int64), ("a_cs", C.c_int64), ("b_rs", C.c_int64), ("b_cs", C.c_int64), ("c_rs", C.c_int64),
        ("nks", C.c_int32), ("a_ks", C.c_int64), ("b_ks", C.c_int64),
        ("nb0", C.c_int32), ("nb1", C.c_int32), ("nb2", C.c_int32),
        ("a_b0", C.c_int64), ("a_b1", C.c_int64), ("a_b2", C.c_int64),
        ("b_b0", C.c_int64), ("b_b1", C.c_int64), ("b_b2", C.c_int64),
        ("c_b0", C.c_int64), ("c_b1", C.c_int64), ("c_b2", C.c_int64),
        ("conjA", C.c_int32), ("conjB", C.c_int32),
    ]


class RunConfig(C.Structure):
    """tjm_run_config of include/tjm_hip.h."""
    _fields_ = [("order", C.c_int32), ("n_times", C.c_int32), ("sample_timesteps", C.c_int32), ("has_noise", C.c_int32),
                ("has_seed", C.c_int32), ("seed", C.c_uint64), ("n_obs", C.c_int32), ("obs_nsites", C.c_void_p),
                ("obs_site", C.c_void_p), ("obs_matrix", C.c_void_p),
                ("start_step", C.c_int32), ("start_phase", C.c_int32), ("rng_pos", C.c_void_p), ("resume", C.c_void_p)]


V = C.c_void_p
I = C.c_int32
D = C.c_double
EXPORTS = {
    # name: (restype, argtypes) - every symbol declared in include/tjm_hip.h
    "tjm_version": (C.c_int, []),
    "tjm_error_string": (C.c_char_p, [C.c_int]),
    "tjm_engine_create": (C.c_int, [C.POINTER(V), I, I, I, I, V]),
    "tjm_engine_create_ex": (C.c_int, [C.POINTER(V), I, I, I, I, V, I]),
    "tjm_engine_destroy": (None, [V]),
    "tjm_engine_workspace_bytes": (C.c_size_t, [V]),
    "tjm_engine_bind": (C.c_int, [V, V, C.c_size_t, V]),
    "tjm_engine_set_params": (C.c_int, [V, D, D, I, I, D, I, I]),
    "tjm_engine_capacity_overflow": (C.c_int, [V, V, I]),
    "tjm_engine_adopt_state": (C.c_int, [V, V, I]),
    "tjm_engine_set_mpo": (C.c_int, [V, V]),
    "tjm_engine_set_noise": (C.c_int, [V, I, V, V, V, V, V, V, V]),
    "tjm_engine_load_state": (C.c_int, [V, I, V, V]),
    "tjm_engine_load_state_slot": (C.c_int, [V, I, I, V, V]),
    "tjm_engine_copy_state": (C.c_int, [V, I, I]),
    "tjm_engine_padded_state_elems": (C.c_size_t, [V]),
    "tjm_engine_bond_caps": (C.c_int, [V, V]),
    "tjm_engine_export_state": (C.c_int, [V, I, I, V, V]),
    "tjm_engine_set_uniforms": (C.c_int, [V, V, I]),
    "tjm_engine_tdvp": (C.c_int, [V, I]),
    "tjm_engine_dissipate": (C.c_int, [V, I, D]),
    "tjm_engine_dissipate_from": (C.c_int, [V, I, D, I]),
    "tjm_engine_set_noise_filter": (C.c_int, [V, I, V]),
    "tjm_engine_normalize_qr": (C.c_int, [V, I, I]),
    "tjm_engine_apply_single": (C.c_int, [V, I, I, V]),
    "tjm_engine_tebd_gate": (C.c_int, [V, I, I, V]),
    "tjm_engine_tebd_gate_at": (C.c_int, [V, I, I, I, V]),
    "tjm_engine_apply_pair": (C.c_int, [V, I, I, V, I]),
    "tjm_engine_canonicalize_qr": (C.c_int, [V, I, I]),
    "tjm_engine_stochastic": (C.c_int, [V, I, D, V, V]),
    "tjm_engine_site_moments": (C.c_int, [V, I, V]),
    "tjm_engine_site_moments2": (C.c_int, [V, I, V, V]),
    "tjm_engine_bond_dims": (C.c_int, [V, I, V]),
    "tjm_engine_site0_normsq": (C.c_int, [V, I, V]),
    "tjm_engine_bond_spectrum": (C.c_int, [V, I, I, V, I]),
    "tjm_engine_bitstring_probability": (C.c_int, [V, I, V, V]),
    "tjm_engine_sample_shots": (C.c_int, [V, I, I, V, V, V]),
    "tjm_engine_stats": (C.c_int, [V, V]),
    "tjm_engine_stats_ex": (C.c_int, [V, V, I]),
    "tjm_heff_apply": (C.c_int, [V, I, I, I, I, I, V, V, V, V, V, I]),
    "tjm_env_update": (C.c_int, [V, I, I, I, I, I, V, V, V, V, I]),
    "tjm_project_bond": (C.c_int, [V, I, I, I, V, V, V, V, I]),
    "tjm_lanczos_expm": (C.c_int, [V, I, I, I, I, I, V, V, V, V, D, D, V, I, V]),
    "tjm_engine_center_shift": (C.c_int, [V, I, I, I, I]),
    "tjm_engine_jump_weights": (C.c_int, [V, I, D, V, V, V]),
    "tjm_engine_step_env_init": (C.c_int, [V, I]),
    "tjm_engine_step_two_site": (C.c_int, [V, I, I, D, I, I, V, I]),
    "tjm_engine_step_one_site": (C.c_int, [V, I, I, D, V, I]),
    "tjm_engine_step_env": (C.c_int, [V, I, I, I, V, I]),
    "tjm_engine_step_qr_bond": (C.c_int, [V, I, I, I, D, I, V, I]),
    "tjm_engine_step_cap_bond": (C.c_int, [V, I, I, I, V, I]),
    "tjm_engine_sweep_dynamic": (C.c_int, [V, I, I, D]),
    "tjm_engine_bug_sweep": (C.c_int, [V, I, D]),
    "tjm_engine_step_bug_prepare": (C.c_int, [V, I]),
    "tjm_engine_step_bug_site": (C.c_int, [V, I, I, D]),
    "tjm_engine_step_bug_root": (C.c_int, [V, I, D]),
    "tjm_engine_step_flip": (C.c_int, [V, I]),
    "tjm_engine_apply_gate_mpo": (C.c_int, [V, I, I, I, I, V, V]),
    "tjm_engine_step_compress": (C.c_int, [V, I, D, I, I]),
    "tjm_engine_profile": (C.c_int, [V, I]),
    "tjm_engine_profile_read": (C.c_int, [V, V, V]),
    "tjm_engine_run": (C.c_int, [V, C.POINTER(RunConfig), V, V, V]),
    "tjm_engine_run_status": (C.c_int, [V, C.POINTER(RunConfig), V, V, V, V]),
    "tjm_rng_uniforms": (C.c_int, [I, C.c_uint64, C.c_uint64, C.c_int64, I, V]),
    "tjm_zgemm_batched": (C.c_int, [C.POINTER(GemmDesc), V]),
    "tjm_svd_workspace_bytes": (C.c_size_t, [I, I]),
    "tjm_svd_split": (C.c_int, [V, I, I, I, I, I, V, V, I, I, D, I, I, V, V, I, V, C.c_size_t, V, V]),
    "tjm_svd_qr_workspace_bytes": (C.c_size_t, [I, I]),
    "tjm_svd_split_qr": (C.c_int, [V, I, I, I, I, I, V, V, I, I, D, I, I, V, V, I, V, C.c_size_t, V, V]),
    "tjm_tridiag_expm": (C.c_int, [V, V, I, D, V, V]),
    "tjm_profile_cross_kernel": (C.c_int, [I]),
    "tjm_profile_cross_kernel_read": (C.c_int, [V, V, V]),
    "tjm_svd_work_read": (C.c_int, [V, I]),
    "tjm_svd_mixed_read": (C.c_int, [V, I]),
    "tjm_profile_cross_kernel_read_c64": (C.c_int, [V, V, V]),
    "tjm_profile_qr_apply": (C.c_int, [I]),
    "tjm_profile_qr_apply_read": (C.c_int, [V, V]),
    "tjm_profile_gemm": (C.c_int, [I]),
    "tjm_profile_gemm_read": (C.c_int, [V]),
}

_lib = None
_libs: dict = {}


def load(dtype: str = "complex128") -> C.CDLL:
    """Load libtjm_hip.so (``dtype="complex128"``, the reference's precision) or libtjm_hip_f32.so (``"complex64"``) and attach
    signatures.  Raises if the library is not built."""
    global _lib
    if dtype not in DTYPES:
        raise ValueError(f"dtype must be one of {DTYPES}, got {dtype!r}")
    if dtype in _libs:
        return _libs[dtype]
    path = LIB_PATH if dtype == "complex128" else LIB_PATH_F32
    if not os.path.exists(path):
        raise TjmError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C yaqs_amd/csrc).  yaqs_amd has no CPU fallback."
        )
    # PyTorch ships its own libamdhip64; load it FIRST so that this library binds to the same HIP runtime instance.  Loaded the
    # other way round, the process holds two runtimes and the second one reports "no ROCm-capable device is detected".
    import torch  # noqa: F401

    lib = C.CDLL(path)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _libs[dtype] = lib
    if dtype == "complex128":
        _lib = lib
    return lib


def check(code: int, what: str = "") -> None:
    if code == 0:
        return
    msg = load().tjm_error_string(code).decode()
    exc = ERRORS.get(code, TjmError)
    raise exc(f"tjm_hip {what}: {msg} (code {code})")
