// extern "C" surface of libtjm_hip.so (declared in include/tjm_hip.h).
#include <cstring>
#include <mutex>
#include <vector>
#include <new>

#include "../../include/tjm_hip.h"
#include "tjm_engine.h"

using namespace tjm;

struct tjm_engine {
  Engine impl;
};

namespace {
// Every entry point that touches the device runs on the device that owns the engine's workspace, whatever device the calling
// thread has current (one process may drive several GPUs, and a torchrun rank >= 1 starts with device 0 current).
struct DeviceGuard {
  int prev = -1, want = -1;
  explicit DeviceGuard(int device) : want(device) {
    if (want < 0) return;
    if (hipGetDevice(&prev) != hipSuccess) { prev = -1; return; }
    if (prev != want && hipSetDevice(want) != hipSuccess) prev = -1;
  }
  ~DeviceGuard() {
    if (want >= 0 && prev >= 0 && prev != want) (void)hipSetDevice(prev);
  }
};
int device_of(const void* dev_ptr) {
  hipPointerAttribute_t a;
  if (!dev_ptr || hipPointerGetAttributes(&a, dev_ptr) != hipSuccess) { (void)hipGetLastError(); return -1; }
  return a.device;
}
}  // namespace
#define TJM_ON_DEVICE(e) DeviceGuard tjm_guard_((e) ? (e)->impl.device_id : -1)

extern "C" {

int tjm_version(void) { return 100; }

const char* tjm_error_string(int code) {
  switch (code) {
    case TJM_OK: return "ok";
    case TJM_ERR_ARG: return "invalid argument";
    case TJM_ERR_HIP: return "HIP runtime error";
    case TJM_ERR_WORKSPACE: return "workspace too small";
    case TJM_ERR_NOT_IMPLEMENTED: return "not implemented";
    case TJM_ERR_NUMERIC: return "numerical failure";
    case TJM_ERR_STATE: return "engine state error";
    case TJM_ERR_ASSERT: return "measurement should be real";
    case TJM_ERR_CAPACITY: return "a truncation needs a bond beyond the engine's capacity";
    default: return "unknown";
  }
}

int tjm_engine_create(tjm_engine** out, int32_t L, int32_t d, int32_t chi_max, int32_t B, const int32_t* mpo_bond) {
  if (!out || !mpo_bond) return TJM_ERR_ARG;
  tjm_engine* e = new (std::nothrow) tjm_engine();
  if (!e) return TJM_ERR_ARG;
  const int rc = e->impl.create(L, d, chi_max, B, mpo_bond);
  if (rc != TJM_OK) { delete e; return rc; }
  *out = e;
  return TJM_OK;
}

int tjm_engine_create_ex(tjm_engine** out, int32_t L, int32_t d, int32_t chi_max, int32_t B, const int32_t* mpo_bond, int32_t cap_slack) {
  if (!out || !mpo_bond) return TJM_ERR_ARG;
  tjm_engine* e = new (std::nothrow) tjm_engine();
  if (!e) return TJM_ERR_ARG;
  const int rc = e->impl.create(L, d, chi_max, B, mpo_bond, cap_slack);
  if (rc != TJM_OK) { delete e; return rc; }
  *out = e;
  return TJM_OK;
}

void tjm_engine_destroy(tjm_engine* e) {
  TJM_ON_DEVICE(e);
  delete e;
}

size_t tjm_engine_workspace_bytes(const tjm_engine* e) { return e ? e->impl.workspace_bytes() : 0; }

int tjm_engine_bind(tjm_engine* e, void* ws, size_t bytes, void* stream) {
  if (!e || !ws) return TJM_ERR_ARG;
  e->impl.device_id = device_of(ws);
  TJM_ON_DEVICE(e);
  return e->impl.bind(ws, bytes, static_cast<hipStream_t>(stream));
}

int tjm_engine_set_params(tjm_engine* e, double dt, double svd_threshold, int32_t trunc_mode, int32_t max_bond, double krylov_tol,
                          int32_t tdvp_mode, int32_t tdvp_sweeps) {
  if (!e || !(dt > 0) || trunc_mode < 0 || trunc_mode > 3 || tdvp_sweeps < 1) return TJM_ERR_ARG;
  e->impl.dt = dt; e->impl.svd_threshold = svd_threshold; e->impl.trunc_mode = trunc_mode; e->impl.max_bond = max_bond;
  e->impl.krylov_tol = krylov_tol; e->impl.tdvp_mode = tdvp_mode; e->impl.tdvp_sweeps = tdvp_sweeps;
#ifdef TJM_F32
  // the adaptive stop of the Lanczos exponential cannot see below the rounding of its own vectors: a tolerance under ~100 eps would
  // only run every exponential to the iteration cap
  if (e->impl.krylov_tol < 100.0 * TJM_EPS) e->impl.krylov_tol = 100.0 * TJM_EPS;
#endif
  return TJM_OK;
}

int tjm_engine_capacity_overflow(tjm_engine* e, int32_t* flag, int32_t clear) {
  if (!e || !flag) return TJM_ERR_ARG;
  int f = 0;
  TJM_ON_DEVICE(e);
  const int rc = e->impl.capacity_overflow(&f, clear != 0);
  *flag = f;
  return rc;
}

int tjm_engine_adopt_state(tjm_engine* dst, tjm_engine* src, int32_t src_first) {
  if (!dst || !src || dst == src || dst->impl.device_id != src->impl.device_id) return TJM_ERR_ARG;
  TJM_ON_DEVICE(dst);
  return dst->impl.adopt(src->impl, src_first);
}

int tjm_engine_set_mpo(tjm_engine* e, const double* host_mpo) { TJM_ON_DEVICE(e); return (e && host_mpo) ? e->impl.set_mpo(host_mpo) : TJM_ERR_ARG; }

int tjm_engine_set_noise(tjm_engine* e, int32_t nproc, const int32_t* nsites, const int32_t* sites, const double* gamma,
                         const int32_t* pauli, const double* mats, const double* factors, const int32_t* has_factors) {
  if (!e || nproc < 0) return TJM_ERR_ARG;
  const size_t dd = (size_t)e->impl.d * e->impl.d, slot = dd * dd;
  std::vector<NoiseProc> v(nproc);
  for (int k = 0; k < nproc; ++k) {
    NoiseProc& p = v[k];
    std::memset(&p, 0, sizeof(p));
    p.nsites = nsites[k]; p.site0 = sites[2 * k]; p.site1 = sites[2 * k + 1]; p.gamma = gamma[k]; p.pauli = pauli[k];
    for (size_t q = 0; q < slot; ++q)  // d^4 complex128 entries per process (16 for qubits)
      p.mat[q] = cplx{(real)mats[2 * (slot * (size_t)k + q)], (real)mats[2 * (slot * (size_t)k + q) + 1]};
    p.has_factors = has_factors ? has_factors[k] : 0;
    if (p.has_factors && factors) {
      for (size_t q = 0; q < dd; ++q) {
        const double* f = factors + 4 * dd * (size_t)k;
        p.f0[q] = cplx{(real)f[2 * q], (real)f[2 * q + 1]};
        p.f1[q] = cplx{(real)f[2 * dd + 2 * q], (real)f[2 * dd + 2 * q + 1]};
      }
    }
    if (p.nsites == 2 && p.site1 - p.site0 > 1 && !p.has_factors) return TJM_ERR_ARG;
  }
  TJM_ON_DEVICE(e);
  return e->impl.set_noise(v);
}

int tjm_engine_load_state(tjm_engine* e, int32_t set, const double* t, const int32_t* bonds) {
  TJM_ON_DEVICE(e);
  return (e && t && bonds) ? e->impl.load_state(set, t, bonds) : TJM_ERR_ARG;
}
int tjm_engine_load_state_slot(tjm_engine* e, int32_t set, int32_t b, const double* t, const int32_t* bonds) {
  TJM_ON_DEVICE(e);
  return (e && t && bonds) ? e->impl.load_state_slot(set, b, t, bonds) : TJM_ERR_ARG;
}
int tjm_engine_copy_state(tjm_engine* e, int32_t dst, int32_t src) { TJM_ON_DEVICE(e); return e ? e->impl.copy_state(dst, src) : TJM_ERR_ARG; }

size_t tjm_engine_padded_state_elems(const tjm_engine* e) {
  size_t n = 0;
  for (int i = 0; i < e->impl.L; ++i) n += (size_t)e->impl.d * e->impl.cap[i] * e->impl.cap[i + 1];
  return n;
}
int tjm_engine_bond_caps(const tjm_engine* e, int32_t* caps) {
  for (int i = 0; i <= e->impl.L; ++i) caps[i] = e->impl.cap[i];
  return TJM_OK;
}
int tjm_engine_export_state(tjm_engine* e, int32_t set, int32_t b, double* out, int32_t* bonds) {
  TJM_ON_DEVICE(e);
  return (e && out && bonds && set >= 0 && set < 2) ? e->impl.export_state(set, b, out, bonds) : TJM_ERR_ARG;
}
int tjm_engine_set_uniforms(tjm_engine* e, const double* u, int32_t n) { TJM_ON_DEVICE(e); return (e && u && n > 0) ? e->impl.set_uniforms(u, n) : TJM_ERR_ARG; }
int tjm_engine_tdvp(tjm_engine* e, int32_t set) { TJM_ON_DEVICE(e); return (e && set >= 0 && set < 2) ? e->impl.tdvp(set) : TJM_ERR_ARG; }
int tjm_engine_dissipate(tjm_engine* e, int32_t set, double dt) { TJM_ON_DEVICE(e); return (e && set >= 0 && set < 2) ? e->impl.dissipate(set, dt) : TJM_ERR_ARG; }
int tjm_engine_dissipate_from(tjm_engine* e, int32_t set, double dt, int32_t center) {
  TJM_ON_DEVICE(e);
  return (e && set >= 0 && set < 2) ? e->impl.dissipate(set, dt, center) : TJM_ERR_ARG;
}
int tjm_engine_set_noise_filter(tjm_engine* e, int32_t n, const int32_t* idx) { TJM_ON_DEVICE(e); return e ? e->impl.set_noise_filter(n, idx) : TJM_ERR_ARG; }
int tjm_engine_normalize_qr(tjm_engine* e, int32_t set, int32_t center) {
  TJM_ON_DEVICE(e);
  return (e && set >= 0 && set < 2) ? e->impl.normalize_qr(set, center) : TJM_ERR_ARG;
}
int tjm_engine_apply_single(tjm_engine* e, int32_t set, int32_t site, const double* mat) {
  TJM_ON_DEVICE(e);
  return (e && mat && set >= 0 && set < 2) ? e->impl.apply_single(set, site, mat) : TJM_ERR_ARG;
}
int tjm_engine_tebd_gate(tjm_engine* e, int32_t set, int32_t left, const double* u) {
  TJM_ON_DEVICE(e);
  return (e && u && set >= 0 && set < 2) ? e->impl.tebd_gate(set, left, u) : TJM_ERR_ARG;
}
int tjm_engine_apply_pair(tjm_engine* e, int32_t set, int32_t left, const double* u, int32_t min_keep) {
  TJM_ON_DEVICE(e);
  return (e && u && set >= 0 && set < 2) ? e->impl.apply_pair(set, left, u, min_keep) : TJM_ERR_ARG;
}
int tjm_engine_canonicalize_qr(tjm_engine* e, int32_t set, int32_t center) {
  TJM_ON_DEVICE(e);
  return (e && set >= 0 && set < 2) ? e->impl.canonicalize_qr(set, center) : TJM_ERR_ARG;
}
int tjm_engine_tebd_gate_at(tjm_engine* e, int32_t set, int32_t left, int32_t center, const double* u) {
  TJM_ON_DEVICE(e);
  return (e && u && set >= 0 && set < 2) ? e->impl.tebd_gate(set, left, u, center) : TJM_ERR_ARG;
}
int tjm_engine_stochastic(tjm_engine* e, int32_t set, double dt, int32_t* jumped, double* dp) {
  TJM_ON_DEVICE(e);
  return (e && set >= 0 && set < 2) ? e->impl.stochastic(set, dt, jumped, dp) : TJM_ERR_ARG;
}
int tjm_engine_site_moments(tjm_engine* e, int32_t set, double* M) { TJM_ON_DEVICE(e); return (e && M) ? e->impl.site_moments(set, M) : TJM_ERR_ARG; }
int tjm_engine_site_moments2(tjm_engine* e, int32_t set, double* M, double* M2) {
  TJM_ON_DEVICE(e);
  return (e && M && M2) ? e->impl.site_moments(set, M, M2) : TJM_ERR_ARG;
}
int tjm_engine_bond_dims(tjm_engine* e, int32_t set, int32_t* chi) { TJM_ON_DEVICE(e); return (e && chi) ? e->impl.bond_dims(set, chi) : TJM_ERR_ARG; }
int tjm_engine_site0_normsq(tjm_engine* e, int32_t set, double* out) { TJM_ON_DEVICE(e); return (e && out) ? e->impl.site_normsq0(set, out) : TJM_ERR_ARG; }
int tjm_engine_bond_spectrum(tjm_engine* e, int32_t set, int32_t site, double* spectrum, int32_t n_out) {
  TJM_ON_DEVICE(e);
  return (e && set >= 0 && set < 2) ? e->impl.bond_spectrum(set, site, spectrum, n_out) : TJM_ERR_ARG;
}

int tjm_engine_bitstring_probability(tjm_engine* e, int32_t set, const uint8_t* bits, double* prob) {
  TJM_ON_DEVICE(e);
  return (e && set >= 0 && set < 2) ? e->impl.bitstring_probability(set, bits, prob) : TJM_ERR_ARG;
}

int tjm_engine_sample_shots(tjm_engine* e, int32_t set, int32_t shots, const double* rotation, const double* uniforms, uint8_t* bits) {
  TJM_ON_DEVICE(e);
  return (e && set >= 0 && set < 2) ? e->impl.sample_shots(set, shots, rotation, uniforms, bits) : TJM_ERR_ARG;
}

int tjm_engine_run(tjm_engine* e, const tjm_run_config* cfg, const int64_t* traj, double* results, double* diagnostics) {
  if (!e) return TJM_ERR_ARG;
  TJM_ON_DEVICE(e);
  return run_batch(e->impl, cfg, traj, results, diagnostics);
}

int tjm_engine_run_status(tjm_engine* e, const tjm_run_config* cfg, const int64_t* traj, double* results, double* diagnostics, int32_t* status) {
  if (!e || !status) return TJM_ERR_ARG;
  TJM_ON_DEVICE(e);
  return run_batch(e->impl, cfg, traj, results, diagnostics, status);
}

int tjm_rng_uniforms(int32_t has_seed, uint64_t seed, uint64_t traj, int64_t timestep, int32_t n, double* out) {
  if (!out || n < 0) return TJM_ERR_ARG;
  rng_uniforms(has_seed, seed, traj, timestep, n, out);
  return TJM_OK;
}

int tjm_engine_stats(const tjm_engine* e, int64_t* o) {
  if (!e || !o) return TJM_ERR_ARG;
  o[0] = e->impl.stat_matvecs; o[1] = e->impl.stat_krylov_calls; o[2] = e->impl.stat_svds; o[3] = e->impl.stat_svd_sweeps;
  o[4] = e->impl.stat_site_updates;
  return TJM_OK;
}

int tjm_engine_stats_ex(const tjm_engine* e, int64_t* o, int32_t n) {
  if (!e || !o || n < 0) return TJM_ERR_ARG;
  const int64_t v[14] = {e->impl.stat_matvecs, e->impl.stat_krylov_calls, e->impl.stat_svds, e->impl.stat_svd_sweeps,
                         e->impl.stat_site_updates, e->impl.stat_matvecs2, e->impl.stat_env_updates, e->impl.stat_direct_applies,
                         e->impl.stat_svd_mats, e->impl.stat_ident_calls, e->impl.stat_ident_hits, e->impl.stat_cert_traj, e->impl.stat_cert_jumps,
                         e->impl.stat_cert_blocked};
  for (int k = 0; k < n && k < 14; ++k) o[k] = v[k];
  return TJM_OK;
}

int tjm_heff_apply(tjm_engine* e, int32_t nsites, int32_t ca, int32_t cb, int32_t Dl, int32_t Dr, const void* x, const void* Lenv, const void* Renv,
                   const double* host_w, void* y, int32_t nb) {
  if (!e || !x || !Lenv || !Renv || !host_w || !y) return TJM_ERR_ARG;
  TJM_ON_DEVICE(e);
  return e->impl.x_heff_apply(nsites, ca, cb, Dl, Dr, static_cast<const cplx*>(x), static_cast<const cplx*>(Lenv), static_cast<const cplx*>(Renv), host_w,
                              static_cast<cplx*>(y), nb);
}

int tjm_env_update(tjm_engine* e, int32_t left, int32_t ca, int32_t cb, int32_t Dl, int32_t Dr, const void* A, const void* env, const double* host_w,
                   void* out, int32_t nb) {
  if (!e || !A || !env || !host_w || !out) return TJM_ERR_ARG;
  TJM_ON_DEVICE(e);
  return e->impl.x_env_update(left, ca, cb, Dl, Dr, static_cast<const cplx*>(A), static_cast<const cplx*>(env), host_w, static_cast<cplx*>(out), nb);
}

int tjm_project_bond(tjm_engine* e, int32_t cu, int32_t cv, int32_t D, const void* C, const void* Lenv, const void* Renv, void* y, int32_t nb) {
  if (!e || !C || !Lenv || !Renv || !y) return TJM_ERR_ARG;
  TJM_ON_DEVICE(e);
  return e->impl.x_project_bond(cu, cv, D, static_cast<const cplx*>(C), static_cast<const cplx*>(Lenv), static_cast<const cplx*>(Renv),
                                static_cast<cplx*>(y), nb);
}

int tjm_lanczos_expm(tjm_engine* e, int32_t nsites, int32_t ca, int32_t cb, int32_t Dl, int32_t Dr, const void* x, const void* Lenv, const void* Renv,
                     const double* host_w, double dt, double tol, void* y, int32_t nb, int64_t* matvecs) {
  if (!e || !x || !Lenv || !Renv || !host_w || !y) return TJM_ERR_ARG;
  TJM_ON_DEVICE(e);
  long mv = 0;
  const int rc = e->impl.x_lanczos_expm(nsites, ca, cb, Dl, Dr, static_cast<const cplx*>(x), static_cast<const cplx*>(Lenv),
                                        static_cast<const cplx*>(Renv), host_w, dt, tol, static_cast<cplx*>(y), nb, &mv);
  if (matvecs) *matvecs = mv;
  return rc;
}

int tjm_engine_center_shift(tjm_engine* e, int32_t set, int32_t site, int32_t direction, int32_t use_svd) {
  if (!e) return TJM_ERR_ARG;
  TJM_ON_DEVICE(e);
  return e->impl.x_center_shift(set, site, direction, use_svd);
}

int tjm_engine_jump_weights(tjm_engine* e, int32_t set, double dt, int32_t* order, double* weights, int32_t* n_out) {
  if (!e) return TJM_ERR_ARG;
  TJM_ON_DEVICE(e);
  int n = 0;
  const int rc = e->impl.x_jump_weights(set, dt, order, weights, &n);
  if (n_out) *n_out = n;
  return rc;
}

int tjm_engine_step_env_init(tjm_engine* e, int32_t set) {
  if (!e) return TJM_ERR_ARG;
  TJM_ON_DEVICE(e);
  return e->impl.step_env_init(set);
}
int tjm_engine_step_two_site(tjm_engine* e, int32_t set, int32_t site, double dt, int32_t dist, int32_t capped, const int32_t* ids, int32_t n) {
  if (!e) return TJM_ERR_ARG;
  TJM_ON_DEVICE(e);
  return e->impl.step_two_site(set, site, dt, dist, capped, ids, n);
}
int tjm_engine_step_one_site(tjm_engine* e, int32_t set, int32_t site, double dt, const int32_t* ids, int32_t n) {
  if (!e) return TJM_ERR_ARG;
  TJM_ON_DEVICE(e);
  return e->impl.step_one_site(set, site, dt, ids, n);
}
int tjm_engine_step_env(tjm_engine* e, int32_t set, int32_t site, int32_t left, const int32_t* ids, int32_t n) {
  if (!e) return TJM_ERR_ARG;
  TJM_ON_DEVICE(e);
  return e->impl.step_env(set, site, left, ids, n);
}
int tjm_engine_step_qr_bond(tjm_engine* e, int32_t set, int32_t site, int32_t right, double dt, int32_t max_bond, const int32_t* ids, int32_t n) {
  if (!e) return TJM_ERR_ARG;
  TJM_ON_DEVICE(e);
  return e->impl.step_qr_bond(set, site, right, dt, max_bond, ids, n);
}
int tjm_engine_step_cap_bond(tjm_engine* e, int32_t set, int32_t bond, int32_t target, const int32_t* ids, int32_t n) {
  if (!e) return TJM_ERR_ARG;
  TJM_ON_DEVICE(e);
  return e->impl.step_cap_bond(set, bond, target, ids, n);
}

int tjm_engine_step_bug_prepare(tjm_engine* e, int32_t set) {
  if (!e) return TJM_ERR_ARG;
  TJM_ON_DEVICE(e);
  return e->impl.step_bug_prepare(set);
}
int tjm_engine_step_bug_site(tjm_engine* e, int32_t set, int32_t site, double dt) {
  if (!e) return TJM_ERR_ARG;
  TJM_ON_DEVICE(e);
  return e->impl.step_bug_site(set, site, dt);
}
int tjm_engine_step_bug_root(tjm_engine* e, int32_t set, double dt) {
  if (!e) return TJM_ERR_ARG;
  TJM_ON_DEVICE(e);
  return e->impl.step_bug_root(set, dt);
}
int tjm_engine_step_flip(tjm_engine* e, int32_t set) {
  if (!e) return TJM_ERR_ARG;
  TJM_ON_DEVICE(e);
  return e->impl.step_flip(set);
}
int tjm_engine_sweep_dynamic(tjm_engine* e, int32_t set, int32_t max_bond_dim, double dt) {
  if (!e) return TJM_ERR_ARG;
  TJM_ON_DEVICE(e);
  return e->impl.sweep_dynamic(set, max_bond_dim, dt);
}
int tjm_engine_bug_sweep(tjm_engine* e, int32_t set, double dt) {
  if (!e) return TJM_ERR_ARG;
  TJM_ON_DEVICE(e);
  return e->impl.bug_sweep(set, dt);
}
int tjm_engine_step_compress(tjm_engine* e, int32_t set, double threshold, int32_t max_bond_dim, int32_t trunc_mode) {
  if (!e) return TJM_ERR_ARG;
  TJM_ON_DEVICE(e);
  return e->impl.step_compress(set, threshold, max_bond_dim, trunc_mode);
}

int tjm_engine_apply_gate_mpo(tjm_engine* e, int32_t set, int32_t first, int32_t last, int32_t rank, const double* left_ops, const double* right_ops) {
  if (!e) return TJM_ERR_ARG;
  TJM_ON_DEVICE(e);
  return e->impl.apply_gate_mpo(set, first, last, rank, left_ops, right_ops);
}

int tjm_engine_profile(tjm_engine* e, int32_t enable) {
  if (!e) return TJM_ERR_ARG;
  TJM_ON_DEVICE(e);
  e->impl.profile_enable(enable != 0);
  return TJM_OK;
}

int tjm_engine_profile_read(tjm_engine* e, double* ms3, int64_t* regions3) {
  if (!e || !ms3 || !regions3) return TJM_ERR_ARG;
  TJM_ON_DEVICE(e);
  long n[Engine::PROF_NCLASS];
  const int rc = e->impl.profile_read(ms3, n);
  for (int c = 0; c < Engine::PROF_NCLASS; ++c) regions3[c] = n[c];
  return rc;
}

int tjm_zgemm_batched(const tjm_gemm_desc* t, void* stream) {
  if (!t) return TJM_ERR_ARG;
  GemmDesc g;
  std::memset(&g, 0, sizeof(g));
  g.A = static_cast<const cplx*>(t->A); g.B = static_cast<const cplx*>(t->B); g.C = static_cast<cplx*>(t->C);
  g.M = t->M; g.N = t->N; g.K = t->K;
  g.a_rs = t->a_rs; g.a_cs = t->a_cs; g.b_rs = t->b_rs; g.b_cs = t->b_cs; g.c_rs = t->c_rs;
  g.nks = t->nks; g.a_ks = t->a_ks; g.b_ks = t->b_ks;
  g.nb0 = t->nb0; g.nb1 = t->nb1; g.nb2 = t->nb2;
  g.a_b0 = t->a_b0; g.a_b1 = t->a_b1; g.a_b2 = t->a_b2;
  g.b_b0 = t->b_b0; g.b_b1 = t->b_b1; g.b_b2 = t->b_b2;
  g.c_b0 = t->c_b0; g.c_b1 = t->c_b1; g.c_b2 = t->c_b2;
  g.conjA = t->conjA; g.conjB = t->conjB;
  DeviceGuard guard(device_of(t->C));
  return launch_gemm(g, static_cast<hipStream_t>(stream));
}

size_t tjm_svd_workspace_bytes(int32_t max_dim, int32_t B) { return svd_workspace_bytes(max_dim, B) + (size_t)B * 64 + 4096; }

namespace {
std::mutex g_pin_mutex;
std::vector<int*> g_pin_free;  // pinned blocks of 256 bytes, never released (a handful per process)
struct PinnedLease {
  int* p = nullptr;
  int acquire() {
    {
      std::lock_guard<std::mutex> lock(g_pin_mutex);
      if (!g_pin_free.empty()) { p = g_pin_free.back(); g_pin_free.pop_back(); return TJM_OK; }
    }
    TJM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&p), 256, hipHostMallocDefault));
    return TJM_OK;
  }
  ~PinnedLease() {
    if (!p) return;
    std::lock_guard<std::mutex> lock(g_pin_mutex);
    g_pin_free.push_back(p);
  }
};
}  // namespace

static int svd_split_impl(bool use_qr, const void* theta, int32_t B, int32_t d, int32_t capL, int32_t capR, int32_t capM, void* left, void* right,
                          int32_t distribution, int32_t trunc_mode, double threshold, int32_t max_bond, int32_t min_keep, int32_t* chi_lrm,
                          double* spectrum, int32_t spec_ld, void* work, size_t work_bytes, int32_t* sweeps_out, void* stream_) {
  if (!theta || !left || !right || !chi_lrm || !work) return TJM_ERR_ARG;
  DeviceGuard guard(device_of(work));
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int m = d * capL, n = d * capR;
  const int mx = m > n ? m : n;
  const size_t need = use_qr ? tjm_svd_qr_workspace_bytes(mx, B) : tjm_svd_workspace_bytes(mx, B);
  if (work_bytes < need) return TJM_ERR_WORKSPACE;
  char* w = static_cast<char*>(work);
  auto take = [&](size_t nbytes) { char* q = w; w += (nbytes + 255) / 256 * 256; return q; };
  SvdWorkspace sw;
  {
    char* sbase = take(svd_workspace_bytes(mx, B));
    svd_carve(sw, sbase, mx, B);
  }
  // the convergence flags of a call travel through pinned host words: one block per CALL (leased from a pool for its duration), so
  // that calls from several host threads on their own streams do not read each other's flags (re-entrant per device)
  PinnedLease lease;
  if (lease.acquire() != TJM_OK) return TJM_ERR_HIP;
  sw.h_pinned = lease.p;
  QrWorkspace qw;
  std::memset(&qw, 0, sizeof(qw));
  MixedWorkspace mw;
  if (use_qr) {
    char* qbase = take(qr_workspace_bytes(mx, B));
    qr_carve(qw, qbase, mx, B);
    const size_t mb = mixed_split_workspace_bytes(mx, B);
    if (mb > 0) { mw.base = take(mb); mw.bytes = mb; mw.max_dim = mx; mw.B = B; }
  }
  SvdSplitDesc s;
  s.theta = static_cast<const cplx*>(theta); s.theta_b0 = (long)m * n; s.ld_theta = n; s.m = m; s.n = n; s.d = d;
  s.capL = capL; s.capR = capR; s.capM = capM;
  s.left = static_cast<cplx*>(left); s.right = static_cast<cplx*>(right);
  s.left_b0 = (long)d * capL * capM; s.right_b0 = (long)d * capM * capR;
  s.distribution = distribution; s.trunc_mode = trunc_mode; s.threshold = threshold; s.max_bond = max_bond; s.min_keep = min_keep;
  s.chiL = chi_lrm; s.chiR = chi_lrm + 1; s.chiM = chi_lrm + 2; s.chi_stride = 3;
  s.spectrum = reinterpret_cast<real*>(spectrum); s.spec_ld = spec_ld; s.nb0 = B; s.ids = nullptr;  // device array of the build's real type
  int sweeps = 0;
  const int rc = use_qr ? svd_split_qr(s, sw, qw, stream, &sweeps, &mw) : svd_split(s, sw, stream, &sweeps);
  if (sweeps_out) *sweeps_out = sweeps;
  return rc;
}

size_t tjm_svd_qr_workspace_bytes(int32_t max_dim, int32_t B) {
  return tjm_svd_workspace_bytes(max_dim, B) + qr_workspace_bytes(max_dim, B) + mixed_split_workspace_bytes(max_dim, B) + 16384;
}

int tjm_svd_split(const void* theta, int32_t B, int32_t d, int32_t capL, int32_t capR, int32_t capM, void* left, void* right,
                  int32_t distribution, int32_t trunc_mode, double threshold, int32_t max_bond, int32_t min_keep, int32_t* chi_lrm,
                  double* spectrum, int32_t spec_ld, void* work, size_t work_bytes, int32_t* sweeps_out, void* stream_) {
  return svd_split_impl(false, theta, B, d, capL, capR, capM, left, right, distribution, trunc_mode, threshold, max_bond, min_keep, chi_lrm,
                        spectrum, spec_ld, work, work_bytes, sweeps_out, stream_);
}

int tjm_svd_split_qr(const void* theta, int32_t B, int32_t d, int32_t capL, int32_t capR, int32_t capM, void* left, void* right,
                     int32_t distribution, int32_t trunc_mode, double threshold, int32_t max_bond, int32_t min_keep, int32_t* chi_lrm,
                     double* spectrum, int32_t spec_ld, void* work, size_t work_bytes, int32_t* sweeps_out, void* stream_) {
  return svd_split_impl(true, theta, B, d, capL, capR, capM, left, right, distribution, trunc_mode, threshold, max_bond, min_keep, chi_lrm,
                        spectrum, spec_ld, work, work_bytes, sweeps_out, stream_);
}

int tjm_profile_cross_kernel(int32_t every) {
  profile_enable(every);
  mixed_profile_enable(every);
  return TJM_OK;
}

int tjm_profile_cross_kernel_read_c64(double* total_ms, double* total_bytes, int64_t* samples) {
  if (!total_ms || !total_bytes || !samples) return TJM_ERR_ARG;
  long n = 0;
  mixed_profile_get(total_ms, total_bytes, &n);
  *samples = n;
  return TJM_OK;
}

int tjm_svd_mixed_read(double* out10, int32_t reset) {
  if (!out10) return TJM_ERR_ARG;
  mixed_stats_get(out10, reset != 0);
  return TJM_OK;
}

int tjm_svd_work_read(double* out4, int32_t reset) {
  if (!out4) return TJM_ERR_ARG;
  jacobi_work_get(out4, reset != 0);
  return TJM_OK;
}

int tjm_profile_cross_kernel_read(double* total_ms, double* total_bytes, int64_t* samples) {
  if (!total_ms || !total_bytes || !samples) return TJM_ERR_ARG;
  long n = 0;
  profile_get(total_ms, total_bytes, &n);
  *samples = n;
  return TJM_OK;
}

int tjm_profile_qr_apply(int32_t every) {
  qr_profile_enable(every);
  mixed_qr_profile_enable(every);
  return TJM_OK;
}

int tjm_profile_qr_apply_read(double* out5, double* out5_c64) {
  if (!out5 || !out5_c64) return TJM_ERR_ARG;
  qr_profile_get(out5);
  mixed_qr_profile_get(out5_c64);
  return TJM_OK;
}

int tjm_profile_gemm(int32_t every) {
  gemm_profile_enable(every);
  return TJM_OK;
}

int tjm_profile_gemm_read(double* out6) {
  if (!out6) return TJM_ERR_ARG;
  gemm_profile_get(out6);
  return TJM_OK;
}

int tjm_tridiag_expm(const double* alpha, const double* beta, int32_t k, double dt, double* out, void* stream) {
  DeviceGuard guard(device_of(out));
  return launch_tridiag_expm_test(reinterpret_cast<const real*>(alpha), reinterpret_cast<const real*>(beta), k, dt, reinterpret_cast<real*>(out),
                                  static_cast<hipStream_t>(stream));
}

}  // extern "C"
