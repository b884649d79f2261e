// Engine: host-side control of the batched TJM sweep (see tjm_engine.h).
//
// Reference call stack restated here (all paths relative to /root/reference/src/mqt/yaqs):
//   tdvp()            core/methods/tdvp/tdvp.py:69-111 -> integrators.py:161-291 (sweep_2site)
//   dissipate()       core/methods/dissipation.py:50-183
//   stochastic()      core/methods/stochastic_process.py:190-292
//   site_moments()    replaces the QR centre walk of mps.py:1178-1234 (evaluate_observables) by
//                     left "density" environments; same expectation values, no gauge moves.
#include "tjm_engine.h"

#include <atomic>
#include <algorithm>
#include <cmath>
#include <mutex>
#include <cstdlib>
#include <cstring>

namespace tjm {

namespace {
inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
// The C ABI speaks float64 / complex128 whatever the build computes in (tjm_common.h: real): conversions at the host boundary.
inline void from_host_c(cplx* dst, const double* src, size_t n) {
  for (size_t i = 0; i < n; ++i) dst[i] = cplx{(real)src[2 * i], (real)src[2 * i + 1]};
}
inline void to_host_c(double* dst, const cplx* src, size_t n) {
  for (size_t i = 0; i < n; ++i) { dst[2 * i] = src[i].x; dst[2 * i + 1] = src[i].y; }
}
// n complex128 numbers of the caller into device memory
int upload_c(cplx* dev, const double* host, size_t n, hipStream_t s) {
#ifdef TJM_F32
  std::vector<cplx> tmp(n);
  from_host_c(tmp.data(), host, n);
  TJM_HIP_CHECK(hipMemcpyAsync(dev, tmp.data(), n * sizeof(cplx), hipMemcpyHostToDevice, s));
  TJM_HIP_CHECK(hipStreamSynchronize(s));  // the staging buffer goes out of scope
#else
  TJM_HIP_CHECK(hipMemcpyAsync(dev, host, n * sizeof(cplx), hipMemcpyHostToDevice, s));
#endif
  return TJM_OK;
}
inline int round16(int x) { return (x + 15) / 16 * 16; }

// dense expm for tiny matrices (<= 16 x 16: a pair of four-level sites) by scaling and squaring + Taylor
void small_expm(const cplx* in, int n, cplx* out) {
  double nrm = 0.0;
  for (int i = 0; i < n * n; ++i) nrm = std::max(nrm, (double)std::hypot(in[i].x, in[i].y));
  int s = 0;
  while (nrm > 0.25) { nrm *= 0.5; ++s; }
  const double sc = std::ldexp(1.0, -s);
  cplx a[MSLOT], term[MSLOT], res[MSLOT], tmp[MSLOT];
  for (int i = 0; i < n * n; ++i) {
    a[i] = cscale(in[i], sc);
    term[i] = cplx{(i / n == i % n) ? real(1) : real(0), 0.0};
    res[i] = term[i];
  }
  for (int k = 1; k <= 24; ++k) {
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) {
        cplx acc{0.0, 0.0};
        for (int l = 0; l < n; ++l) cfma(acc, term[i * n + l], a[l * n + j]);
        tmp[i * n + j] = cscale(acc, 1.0 / k);
      }
    for (int i = 0; i < n * n; ++i) { term[i] = tmp[i]; res[i] = cadd(res[i], term[i]); }
  }
  for (int q = 0; q < s; ++q) {
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) {
        cplx acc{0.0, 0.0};
        for (int l = 0; l < n; ++l) cfma(acc, res[i * n + l], res[l * n + j]);
        tmp[i * n + j] = acc;
      }
    for (int i = 0; i < n * n; ++i) res[i] = tmp[i];
  }
  for (int i = 0; i < n * n; ++i) out[i] = res[i];
}
}  // namespace

// ---- live class timing -------------------------------------------------------------------------------------------------
void Engine::profile_enable(bool on) {
  prof_collect();
  prof_.on = on;
  if (on) for (int c = 0; c < PROF_NCLASS; ++c) { prof_.ms[c] = 0.0; prof_.n[c] = 0; }
}

void Engine::prof_collect() {
  if (prof_.used == 0) return;
  (void)hipStreamSynchronize(stream);
  for (size_t k = 0; k < prof_.used; ++k) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, prof_.pool[2 * k], prof_.pool[2 * k + 1]) == hipSuccess) {
      prof_.ms[prof_.cls[k]] += ms;
      ++prof_.n[prof_.cls[k]];
    }
  }
  prof_.used = 0;
}

int Engine::profile_read(double* ms, long* regions) {
  prof_collect();
  for (int c = 0; c < PROF_NCLASS; ++c) { ms[c] = prof_.ms[c]; regions[c] = prof_.n[c]; }
  return TJM_OK;
}

Engine::Region::Region(Engine& eng, int cls) : e(eng), idx(-1) {
  if (!e.prof_.on || e.prof_.depth++ > 0) return;  // nested regions count for the outer class
  if (e.prof_.used >= 8192) e.prof_collect();
  idx = (int)e.prof_.used++;
  while (e.prof_.pool.size() < 2 * e.prof_.used) {
    hipEvent_t ev;
    if (hipEventCreate(&ev) != hipSuccess) { idx = -1; --e.prof_.used; return; }
    e.prof_.pool.push_back(ev);
  }
  if (e.prof_.cls.size() < e.prof_.used) e.prof_.cls.resize(e.prof_.used);
  e.prof_.cls[idx] = cls;
  (void)hipEventRecord(e.prof_.pool[2 * idx], e.stream);
}

Engine::Region::~Region() {
  if (e.prof_.on) {
    --e.prof_.depth;
    if (idx >= 0) (void)hipEventRecord(e.prof_.pool[2 * idx + 1], e.stream);
  }
}

Engine::~Engine() {
  direct_clear();
  if (h_pinned_) (void)hipHostFree(h_pinned_);
  for (hipEvent_t ev : krylov_ev_) if (ev) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : prof_.pool) if (ev) (void)hipEventDestroy(ev);
}

int Engine::create(int L_, int d_, int chi_, int B_, const int* mpo_bond, int cap_slack) {
  // uniform local dimension 2, 3 or 4; the trajectory index is the y or z dimension of most grids (at most 65535)
  if (L_ < 1 || d_ < 2 || d_ > 4 || chi_ < 1 || B_ < 1 || B_ > 65535 || cap_slack < 1) return TJM_ERR_ARG;
  L = L_; d = d_; chi_max = chi_; B = B_;
  n_sets = cap_slack > 1 ? 4 : 2;
  // storage of bond k: min(chi_max, slack * min(d^k, d^(L-k))).  slack = 1 is the exact Schmidt-rank bound; the stacked trial bases of
  // the BUG integrator hold up to twice that near the chain ends between a half-sweep and the next canonicalisation (slack = 2)
  cap.assign(L + 1, 1);
  long left = 1;
  for (int i = 1; i < L; ++i) { left = std::min<long>(left * d, (long)chi_max * 4); cap[i] = (int)std::min<long>(left * cap_slack, chi_max); }
  long right = 1;
  for (int i = L - 1; i > 0; --i) { right = std::min<long>(right * d, (long)chi_max * 4); cap[i] = std::min<int>(cap[i], (int)std::min<long>(right * cap_slack, chi_max)); }
  Dm.assign(mpo_bond, mpo_bond + L + 1);
  Dmax = *std::max_element(Dm.begin(), Dm.end());
  if (Dm[0] != 1 || Dm[L] != 1) return TJM_ERR_ARG;
  a_b0_.resize(L); l_b0_.resize(L); r_b0_.resize(L);
  for (int i = 0; i < L; ++i) {
    a_b0_[i] = (long)d * cap[i] * cap[i + 1];
    l_b0_[i] = (long)cap[i] * Dm[i] * cap[i];
    r_b0_[i] = (long)cap[i + 1] * Dm[i + 1] * cap[i + 1];
  }
  return TJM_OK;
}

size_t Engine::workspace_bytes() const {
  const int cm = *std::max_element(cap.begin(), cap.end());
  size_t tot = 0;
  for (int s = 0; s < n_sets; ++s) {
    for (int i = 0; i < L; ++i) tot += align_up((size_t)B * a_b0_[i] * sizeof(cplx));
    tot += align_up((size_t)B * (L + 1) * sizeof(int));
  }
  for (int i = 0; i < L; ++i) tot += align_up((size_t)B * l_b0_[i] * sizeof(cplx)) + align_up((size_t)B * r_b0_[i] * sizeof(cplx));
  const size_t tb = (size_t)d * d * cm * Dmax * cm;
  tot += 2 * align_up((size_t)B * tb * sizeof(cplx));
  const size_t nloc = (size_t)d * d * cm * cm;
  tot += align_up((size_t)B * (mmax + 1) * nloc * sizeof(cplx));
  tot += align_up((size_t)B * (size_t)(d * cm) * (d * cm) * sizeof(cplx));
  tot += align_up(svd_workspace_bytes(d * cm, B));
  tot += 8 * align_up((size_t)B * sizeof(double) * 4);  // small per-trajectory scalars and index lists
  tot += align_up(qr_workspace_bytes(d * cm, B)) + 4096;
  tot += align_up(mixed_split_workspace_bytes(d * cm, B));
  tot += 2 * align_up((size_t)B * TJM_MAX_PART * sizeof(double));
  tot += 4 * align_up((size_t)B * mmax * sizeof(cplx));                 // alpha, beta, coef, svec
  tot += 3 * align_up((size_t)B * cm * cm * sizeof(cplx));              // E ping-pong + bond matrix
  tot += align_up((size_t)L * B * d * d * sizeof(cplx));                // M
  tot += align_up((size_t)L * B * d * d * d * d * sizeof(cplx));        // M2
  // MPO matrices + operator table
  tot += (size_t)(3 * L) * align_up((size_t)(d * d * Dmax) * (d * d * Dmax) * sizeof(cplx));
  tot += align_up((size_t)(L + 64) * MSLOT * sizeof(cplx));
  tot += 2 * align_up((size_t)(d * d * Dmax) * (d * d * Dmax) * sizeof(cplx));  // MPO matrices of the kernel-level exports
  tot += align_up((size_t)4 * L * sizeof(SmallSiteRef)) + 4 * 256 + align_up((size_t)(4 * L + 8) * sizeof(SmallSweepStep));  // fused sweeps
  tot += align_up((size_t)B * (L + 1) * sizeof(int)) + 3 * align_up((size_t)B * sizeof(double));  // certified dissipation: virtual bond table, minima, flags, checksums
  tot += 1 << 16;
  return tot;
}

int Engine::bind(void* ws, size_t bytes, hipStream_t s) {
  if (bytes < workspace_bytes()) return TJM_ERR_WORKSPACE;
  stream = s;
  char* p = static_cast<char*>(ws);
  auto take = [&](size_t n) { char* q = p; p += align_up(n); return q; };
  const int cm = *std::max_element(cap.begin(), cap.end());
  for (int st = 0; st < n_sets; ++st) {
    sets[st].A.resize(L);
    for (int i = 0; i < L; ++i) sets[st].A[i] = reinterpret_cast<cplx*>(take((size_t)B * a_b0_[i] * sizeof(cplx)));
    sets[st].chi = reinterpret_cast<int*>(take((size_t)B * (L + 1) * sizeof(int)));
  }
  Lenv_.resize(L); Renv_.resize(L);
  for (int i = 0; i < L; ++i) {
    Lenv_[i] = reinterpret_cast<cplx*>(take((size_t)B * l_b0_[i] * sizeof(cplx)));
    Renv_[i] = reinterpret_cast<cplx*>(take((size_t)B * r_b0_[i] * sizeof(cplx)));
  }
  t_b0 = (long)d * d * cm * Dmax * cm;
  T1 = reinterpret_cast<cplx*>(take((size_t)B * t_b0 * sizeof(cplx)));
  T2 = reinterpret_cast<cplx*>(take((size_t)B * t_b0 * sizeof(cplx)));
  v_ld = (long)d * d * cm * cm;
  v_b0 = v_ld * (mmax + 1);
  V = reinterpret_cast<cplx*>(take((size_t)B * v_b0 * sizeof(cplx)));
  theta_b0 = (long)(d * cm) * (d * cm);
  theta = reinterpret_cast<cplx*>(take((size_t)B * theta_b0 * sizeof(cplx)));
  {
    char* sbase = take(svd_workspace_bytes(d * cm, B));
    svd_carve(svdw, sbase, d * cm, B);
  }
  {
    const int md = d * cm;
    char* qbase = take(qr_workspace_bytes(md, B));
    qr_carve(qrw, qbase, md, B);
    mixw = MixedWorkspace();
    const size_t mb = mixed_split_workspace_bytes(md, B);
    if (mb > 0) { mixw.base = take(mb); mixw.bytes = mb; mixw.max_dim = md; mixw.B = B; }
  }
  part1_ = reinterpret_cast<real*>(take((size_t)B * TJM_MAX_PART * sizeof(double)));
  part2_ = reinterpret_cast<real*>(take((size_t)B * TJM_MAX_PART * sizeof(double)));
  ks.mmax = mmax;
  ks.alpha = reinterpret_cast<real*>(take((size_t)B * mmax * sizeof(double)));
  ks.beta = reinterpret_cast<real*>(take((size_t)B * mmax * sizeof(double)));
  ks.coef = reinterpret_cast<cplx*>(take((size_t)B * mmax * sizeof(cplx)));
  ks.vnorm = reinterpret_cast<real*>(take((size_t)B * sizeof(double)));
  ks.scale = reinterpret_cast<real*>(take((size_t)B * sizeof(double)));
  ks.svec = reinterpret_cast<real*>(take((size_t)B * mmax * sizeof(double)));
  ks.status = reinterpret_cast<int*>(take((size_t)B * sizeof(int)));
  ks.kfinal = reinterpret_cast<int*>(take((size_t)B * sizeof(int)));
  ks.n_active = reinterpret_cast<int*>(take(256));
  nloc_ = reinterpret_cast<int*>(take((size_t)B * sizeof(int)));
  scal_ = reinterpret_cast<real*>(take((size_t)B * sizeof(double)));
  normsq_ = reinterpret_cast<real*>(take((size_t)B * sizeof(double)));
  ids_ = reinterpret_cast<int*>(take((size_t)B * sizeof(int)));
  opidx_ = reinterpret_cast<int*>(take((size_t)B * sizeof(int)));
  jsite_ = reinterpret_cast<int*>(take((size_t)B * sizeof(int)));
  overflow_ = reinterpret_cast<int*>(take(256));
  for (int st = 0; st < n_sets; ++st) site_refs_[st] = reinterpret_cast<SmallSiteRef*>(take((size_t)L * sizeof(SmallSiteRef)));
  sweep_steps_ = reinterpret_cast<SmallSweepStep*>(take((size_t)(4 * L + 8) * sizeof(SmallSweepStep)));
  TJM_HIP_CHECK(hipMemsetAsync(overflow_, 0, 2 * sizeof(int), s));
  E_ = reinterpret_cast<cplx*>(take((size_t)B * cm * cm * sizeof(cplx)));
  E2_ = reinterpret_cast<cplx*>(take((size_t)B * cm * cm * sizeof(cplx)));
  M_ = reinterpret_cast<cplx*>(take((size_t)L * B * d * d * sizeof(cplx)));
  Cm_ = reinterpret_cast<cplx*>(take((size_t)B * cm * cm * sizeof(cplx)));
  M2_ = reinterpret_cast<cplx*>(take((size_t)L * B * d * d * d * d * sizeof(cplx)));
  const size_t wsz = (size_t)(d * d * Dmax) * (d * d * Dmax) * sizeof(cplx);
  W_.resize(L); WenvL_.resize(L); W2_.resize(L);
  for (int i = 0; i < L; ++i) {
    W_[i] = reinterpret_cast<cplx*>(take(wsz));
    WenvL_[i] = reinterpret_cast<cplx*>(take(wsz));
    W2_[i] = reinterpret_cast<cplx*>(take(wsz));
  }
  ops_ = reinterpret_cast<cplx*>(take((size_t)(L + 64) * MSLOT * sizeof(cplx)));
  Wx_[0] = reinterpret_cast<cplx*>(take(wsz));
  Wx_[1] = reinterpret_cast<cplx*>(take(wsz));
  vchi_ = reinterpret_cast<int*>(take((size_t)B * (L + 1) * sizeof(int)));
  cert_min_ = reinterpret_cast<real*>(take((size_t)B * sizeof(double)));
  cert_flag_ = reinterpret_cast<int*>(take((size_t)B * sizeof(double)));
  csum_ = reinterpret_cast<unsigned long long*>(take((size_t)B * sizeof(double)));
  cert_ok_.assign(B, 0);
  cert_sum_.assign(B, 0);
  cert_set_ = -1;
  if ((size_t)(p - static_cast<char*>(ws)) > bytes) return TJM_ERR_WORKSPACE;
  if (!h_pinned_) TJM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h_pinned_), 256, hipHostMallocDefault));
  svdw.h_pinned = h_pinned_;
  ws_bytes_ = bytes;
  bound_ = true;
  // fused small-bond sweeps: every site must fit the one-wavefront kernels in both directions
  static const bool no_sweeps = getenv("TJM_NO_SWEEP_FUSION") != nullptr;
  sweep_ok_ = !no_sweeps && d == 2 && L >= 2;
  for (int i = 0; i < L && sweep_ok_; ++i)
    sweep_ok_ = svd_shift_small_fits(d, cap[i], cap[i + 1], false) && svd_shift_small_fits(d, cap[i], cap[i + 1], true);
  for (int st = 0; st < n_sets; ++st) {
    std::vector<SmallSiteRef> refs(L);
    for (int i = 0; i < L; ++i) refs[i] = SmallSiteRef{sets[st].A[i], a_b0_[i], cap[i], cap[i + 1]};
    TJM_HIP_CHECK(hipMemcpyAsync(site_refs_[st], refs.data(), refs.size() * sizeof(SmallSiteRef), hipMemcpyHostToDevice, stream));
  }
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  return TJM_OK;
}

// One launch for a list of small-bond centre shifts (with optional one-site factors) on set `set`.
int Engine::run_sweep(int set, const std::vector<SmallSweepStep>& steps, const int* ids, int nb0) {
  if (steps.empty() || nb0 <= 0) return TJM_OK;
  if ((int)steps.size() > 4 * L + 8) return TJM_ERR_ARG;
  TJM_HIP_CHECK(hipMemcpyAsync(sweep_steps_, steps.data(), steps.size() * sizeof(SmallSweepStep), hipMemcpyHostToDevice, stream));
  SmallSweepDesc q;
  q.sites = site_refs_[set]; q.steps = sweep_steps_; q.nsteps = (int)steps.size(); q.d = d;
  q.chi = sets[set].chi; q.chi_stride = L + 1; q.threshold = 1e-12; q.min_keep = 1;
  q.ids = ids; q.nb0 = nb0; q.flags = overflow_;
  q.pitch = small_sweep_pitch(d * *std::max_element(cap.begin(), cap.end()));
  stat_svds += (long)steps.size();
  stat_svd_mats += (long)steps.size() * nb0;
  return launch_small_sweep(q, stream);
}

void Engine::direct_clear() {
  for (auto& kv : direct_) {
    if (kv.second.perm) (void)hipFree(kv.second.perm);
    if (kv.second.coef) (void)hipFree(kv.second.coef);
  }
  direct_.clear();
}

// The direct form of (Wm, lch, rch), analysed once on the host (the matrix is a few hundred entries) and kept until the next upload.
const Engine::DirectForm* Engine::direct_form(const cplx* Wm, int P, int Dl, int Dr, int lch, int rch) {
  const auto key = std::make_tuple(Wm, P, Dl, Dr, lch, rch);
  auto it = direct_.find(key);
  if (it != direct_.end()) return it->second.ok ? &it->second : nullptr;
  DirectForm f;
  const int nin = P * Dr, nl = Dl - 1, first = (lch == 0) ? 1 : 0;
  std::vector<cplx> w((size_t)P * Dl * nin);
  bool ok = (lch == 0 || lch == Dl - 1) && nl >= 1 && hipMemcpyAsync(w.data(), Wm, w.size() * sizeof(cplx), hipMemcpyDeviceToHost, stream) == hipSuccess &&
            hipStreamSynchronize(stream) == hipSuccess;
  std::vector<int> perm((size_t)nl * P, 0);
  std::vector<cplx> coef((size_t)nl * P, cplx{0.0, 0.0});
  for (int ks = 0; ok && ks < nl; ++ks)
    for (int po = 0; ok && po < P; ++po) {
      const cplx* row = w.data() + (size_t)(po * Dl + first + ks) * nin;
      int hits = 0;
      for (int pi = 0; pi < P; ++pi)
        for (int r = 0; r < Dr; ++r) {
          const cplx v = row[pi * Dr + r];
          if (v.x == 0.0 && v.y == 0.0) continue;
          if (r != rch) ok = false;  // a term with operators in BOTH environments: the three-stage form
          ++hits;
          perm[(size_t)ks * P + po] = pi;
          coef[(size_t)ks * P + po] = v;
        }
      if (hits > 1) ok = false;
    }
  if (ok) {
    ok = hipMalloc(reinterpret_cast<void**>(&f.perm), perm.size() * sizeof(int)) == hipSuccess &&
         hipMalloc(reinterpret_cast<void**>(&f.coef), coef.size() * sizeof(cplx)) == hipSuccess &&
         hipMemcpyAsync(f.perm, perm.data(), perm.size() * sizeof(int), hipMemcpyHostToDevice, stream) == hipSuccess &&
         hipMemcpyAsync(f.coef, coef.data(), coef.size() * sizeof(cplx), hipMemcpyHostToDevice, stream) == hipSuccess &&
         hipStreamSynchronize(stream) == hipSuccess;
  }
  f.ok = ok;
  auto ins = direct_.emplace(key, f);
  return ok ? &ins.first->second : nullptr;
}

int Engine::set_mpo(const double* host) {
  if (!bound_) return TJM_ERR_STATE;
  direct_clear();
  Whost_.assign(L, {});
  const double* src = host;
  for (int i = 0; i < L; ++i) {
    const int Dl = Dm[i], Dr = Dm[i + 1];
    Whost_[i].resize((size_t)d * d * Dl * Dr);  // (o,p,l,r)
    from_host_c(Whost_[i].data(), src, Whost_[i].size());
    src += 2 * Whost_[i].size();
  }
  auto W4 = [&](int i, int o, int p, int l, int r) { return Whost_[i][(((size_t)o * d + p) * Dm[i] + l) * Dm[i + 1] + r]; };
  for (int i = 0; i < L; ++i) {
    const int Dl = Dm[i], Dr = Dm[i + 1];
    std::vector<cplx> mv((size_t)d * Dl * d * Dr), el((size_t)d * Dr * d * Dl);
    for (int o = 0; o < d; ++o) for (int p = 0; p < d; ++p) for (int l = 0; l < Dl; ++l) for (int r = 0; r < Dr; ++r) {
      mv[(size_t)(o * Dl + l) * (d * Dr) + (p * Dr + r)] = W4(i, o, p, l, r);
      el[(size_t)(p * Dr + r) * (d * Dl) + (o * Dl + l)] = W4(i, o, p, l, r);
    }
    TJM_HIP_CHECK(hipMemcpyAsync(W_[i], mv.data(), mv.size() * sizeof(cplx), hipMemcpyHostToDevice, stream));
    TJM_HIP_CHECK(hipMemcpyAsync(WenvL_[i], el.data(), el.size() * sizeof(cplx), hipMemcpyHostToDevice, stream));
    TJM_HIP_CHECK(hipStreamSynchronize(stream));
    if (i + 1 < L) {
      const int Dr2 = Dm[i + 2], P = d * d;
      std::vector<cplx> m2((size_t)P * Dl * P * Dr2, cplx{0.0, 0.0});
      for (int o = 0; o < d; ++o) for (int o2 = 0; o2 < d; ++o2) for (int p = 0; p < d; ++p) for (int p2 = 0; p2 < d; ++p2)
        for (int l = 0; l < Dl; ++l) for (int r = 0; r < Dr2; ++r) {
          cplx acc{0.0, 0.0};
          for (int m = 0; m < Dr; ++m) cfma(acc, W4(i, o, p, l, m), W4(i + 1, o2, p2, m, r));
          m2[(size_t)((o * d + o2) * Dl + l) * (P * Dr2) + ((p * d + p2) * Dr2 + r)] = acc;
        }
      TJM_HIP_CHECK(hipMemcpyAsync(W2_[i], m2.data(), m2.size() * sizeof(cplx), hipMemcpyHostToDevice, stream));
      TJM_HIP_CHECK(hipStreamSynchronize(stream));
    }
  }
  return TJM_OK;
}

int Engine::set_noise(const std::vector<NoiseProc>& procs) {
  noise_ = procs;
  proc_on_.assign(procs.size(), 1);
  one_by_site_.assign(L, {});
  two_by_right_.assign(L, {});
  for (size_t k = 0; k < noise_.size(); ++k) {
    const NoiseProc& p = noise_[k];
    if (p.nsites == 1) {
      if (p.site0 < 0 || p.site0 >= L) return TJM_ERR_ARG;
      one_by_site_[p.site0].push_back((int)k);
    } else if (p.nsites == 2) {
      if (p.site0 < 0 || p.site1 >= L || p.site1 <= p.site0) return TJM_ERR_ARG;
      two_by_right_[p.site1].push_back((int)k);
    } else {
      return TJM_ERR_ARG;
    }
  }
  return TJM_OK;
}

int Engine::load_state(int set, const double* host, const int* bonds) {
  if (!bound_ || set < 0 || set > 1) return TJM_ERR_STATE;
  StateSet& S = sets[set];
  const double* src = host;  // complex128: (re, im) pairs
  cert_size();
  for (int b = 0; b < B; ++b) { cert_wait_[(size_t)set * B + b] = 0; cert_back_[(size_t)set * B + b] = 0; }  // new trajectories: nobody sits out
  std::vector<int> chi((size_t)B * (L + 1));
  for (int b = 0; b < B; ++b) for (int k = 0; k <= L; ++k) chi[(size_t)b * (L + 1) + k] = bonds[k];
  for (int k = 0; k <= L; ++k) if (bonds[k] > cap[k] || bonds[k] < 1) return TJM_ERR_ARG;
  TJM_HIP_CHECK(hipMemcpyAsync(S.chi, chi.data(), chi.size() * sizeof(int), hipMemcpyHostToDevice, stream));
  for (int i = 0; i < L; ++i) {
    const int cl = bonds[i], cr = bonds[i + 1];
    std::vector<cplx> pad((size_t)a_b0_[i], cplx{0.0, 0.0});
    for (int p = 0; p < d; ++p) for (int a = 0; a < cl; ++a) for (int c = 0; c < cr; ++c) {
      const double* z = src + 2 * (((size_t)p * cl + a) * cr + c);
      pad[((size_t)p * cap[i] + a) * cap[i + 1] + c] = cplx{(real)z[0], (real)z[1]};
    }
    src += 2 * (size_t)d * cl * cr;
    // first slot from host, the others by device copies
    TJM_HIP_CHECK(hipMemcpyAsync(S.A[i], pad.data(), pad.size() * sizeof(cplx), hipMemcpyHostToDevice, stream));
    TJM_HIP_CHECK(hipStreamSynchronize(stream));
    long done = 1;
    while (done < B) {
      const long n = std::min<long>(done, B - done);
      TJM_HIP_CHECK(hipMemcpyAsync(S.A[i] + done * a_b0_[i], S.A[i], (size_t)n * a_b0_[i] * sizeof(cplx), hipMemcpyDeviceToDevice, stream));
      done += n;
    }
  }
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  return TJM_OK;
}

// One slot of a state set from host tensors (the others keep what they hold): per-trajectory initial states.
int Engine::load_state_slot(int set, int b, const double* host, const int* bonds) {
  if (!bound_ || set < 0 || set > 1 || b < 0 || b >= B) return TJM_ERR_STATE;
  StateSet& S = sets[set];
  for (int k = 0; k <= L; ++k) if (bonds[k] > cap[k] || bonds[k] < 1) return TJM_ERR_ARG;
  cert_size();
  cert_wait_[(size_t)set * B + b] = 0;
  cert_back_[(size_t)set * B + b] = 0;
  TJM_HIP_CHECK(hipMemcpyAsync(S.chi + (long)b * (L + 1), bonds, (size_t)(L + 1) * sizeof(int), hipMemcpyHostToDevice, stream));
  const double* src = host;
  for (int i = 0; i < L; ++i) {
    const int cl = bonds[i], cr = bonds[i + 1];
    std::vector<cplx> pad((size_t)a_b0_[i], cplx{0.0, 0.0});
    for (int p = 0; p < d; ++p) for (int a = 0; a < cl; ++a) for (int c = 0; c < cr; ++c) {
      const double* z = src + 2 * (((size_t)p * cl + a) * cr + c);
      pad[((size_t)p * cap[i] + a) * cap[i + 1] + c] = cplx{(real)z[0], (real)z[1]};
    }
    src += 2 * (size_t)d * cl * cr;
    TJM_HIP_CHECK(hipMemcpyAsync(S.A[i] + (long)b * a_b0_[i], pad.data(), pad.size() * sizeof(cplx), hipMemcpyHostToDevice, stream));
    TJM_HIP_CHECK(hipStreamSynchronize(stream));  // pad is a local
  }
  cert_set_ = -1;
  return TJM_OK;
}

// slot dst of a set = slot src of the same set (tensors and bond row)
int Engine::copy_slot(int set, int dst, int src) {
  if (!bound_ || set < 0 || set >= n_sets || dst < 0 || src < 0 || dst >= B || src >= B) return TJM_ERR_ARG;
  if (dst == src) return TJM_OK;
  StateSet& S = sets[set];
  for (int i = 0; i < L; ++i)
    TJM_HIP_CHECK(hipMemcpyAsync(S.A[i] + (long)dst * a_b0_[i], S.A[i] + (long)src * a_b0_[i], (size_t)a_b0_[i] * sizeof(cplx), hipMemcpyDeviceToDevice, stream));
  TJM_HIP_CHECK(hipMemcpyAsync(S.chi + (long)dst * (L + 1), S.chi + (long)src * (L + 1), (size_t)(L + 1) * sizeof(int), hipMemcpyDeviceToDevice, stream));
  cert_size();
  cert_wait_[(size_t)set * B + dst] = cert_wait_[(size_t)set * B + src];
  cert_back_[(size_t)set * B + dst] = cert_back_[(size_t)set * B + src];
  cert_set_ = -1;
  return TJM_OK;
}

// flags[b] |= 1 when a site tensor of trajectory b holds a NaN or an infinity
__global__ __launch_bounds__(256) void finite_flags_kernel(const cplx* __restrict__ A, long a_b0, int* __restrict__ flags) {
  const cplx* Ab = A + (long)blockIdx.y * a_b0;
  int bad = 0;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < a_b0; e += (long)gridDim.x * blockDim.x) {
    const cplx v = Ab[e];
    const real t = v.x - v.x + (v.y - v.y);  // 0 for finite entries, NaN otherwise
    bad |= (t == real(0.0)) ? 0 : 1;
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(&flags[blockIdx.y], 1);
}

// host_flags[b] = 1 when the state of trajectory b holds a non-finite number
int Engine::finite_check(int set, int* host_flags) {
  if (!bound_ || set < 0 || set >= n_sets) return TJM_ERR_STATE;
  StateSet& S = sets[set];
  TJM_HIP_CHECK(hipMemsetAsync(ids_, 0, (size_t)B * sizeof(int), stream));
  for (int i = 0; i < L; ++i) {
    int gx = (int)((a_b0_[i] + 4095) / 4096);
    if (gx < 1) gx = 1;
    if (gx > 16) gx = 16;
    hipLaunchKernelGGL(finite_flags_kernel, dim3(gx, B), dim3(256), 0, stream, S.A[i], a_b0_[i], ids_);
  }
  TJM_HIP_CHECK(hipGetLastError());
  TJM_HIP_CHECK(hipMemcpyAsync(host_flags, ids_, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, stream));
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  return TJM_OK;
}

int Engine::copy_state(int dst, int src) {
  if (!bound_ || dst == src || dst < 0 || src < 0 || dst >= n_sets || src >= n_sets) return TJM_ERR_ARG;
  for (int i = 0; i < L; ++i)
    TJM_HIP_CHECK(hipMemcpyAsync(sets[dst].A[i], sets[src].A[i], (size_t)B * a_b0_[i] * sizeof(cplx), hipMemcpyDeviceToDevice, stream));
  TJM_HIP_CHECK(hipMemcpyAsync(sets[dst].chi, sets[src].chi, (size_t)B * (L + 1) * sizeof(int), hipMemcpyDeviceToDevice, stream));
  cert_size();  // the copy inherits the certificate back-off of its source (psi = deepcopy(phi); snapshot / roll-back of a step)
  for (int b = 0; b < B; ++b) {
    cert_wait_[(size_t)dst * B + b] = cert_wait_[(size_t)src * B + b];
    cert_back_[(size_t)dst * B + b] = cert_back_[(size_t)src * B + b];
  }
  return TJM_OK;
}

// dst[b][p][a][c] (extents capL x capR) = src[b][p][a][c] (extents sl x sr) inside the source extents, zero outside
__global__ __launch_bounds__(256) void repad_kernel(const cplx* __restrict__ src, long src_b0, int sl, int sr, cplx* __restrict__ dst, long dst_b0,
                                                   int capL, int capR, int d) {
  const long n = (long)d * capL * capR;
  const int b = blockIdx.y;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
    const int c = (int)(t % capR);
    const long q = t / capR;
    const int a = (int)(q % capL), p = (int)(q / capL);
    cplx v{0.0, 0.0};
    if (a < sl && c < sr) v = src[(long)b * src_b0 + ((long)p * sl + a) * sr + c];
    dst[(long)b * dst_b0 + t] = v;
  }
}

// Take over the trajectory states (set 0) of slots [first, first + B) of a smaller-capacity engine.
int Engine::adopt(Engine& src, int first) {
  if (!bound_ || !src.bound_ || src.L != L || src.d != d || first < 0 || first + B > src.B) return TJM_ERR_ARG;
  for (int k = 0; k <= L; ++k) if (src.cap[k] > cap[k]) return TJM_ERR_ARG;
  TJM_HIP_CHECK(hipStreamSynchronize(src.stream));
  for (int i = 0; i < L; ++i) {
    const long n = (long)d * cap[i] * cap[i + 1];
    const int gx = (int)std::min<long>((n + 255) / 256, 64);
    hipLaunchKernelGGL(repad_kernel, dim3(gx, B), dim3(256), 0, stream, src.sets[0].A[i] + (long)first * src.a_b0_[i], src.a_b0_[i], src.cap[i],
                       src.cap[i + 1], sets[0].A[i], a_b0_[i], cap[i], cap[i + 1], d);
  }
  TJM_HIP_CHECK(hipGetLastError());
  TJM_HIP_CHECK(hipMemcpyAsync(sets[0].chi, src.sets[0].chi + (size_t)first * (L + 1), (size_t)B * (L + 1) * sizeof(int), hipMemcpyDeviceToDevice, stream));
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  cert_size();
  src.cert_size();
  for (int b = 0; b < B; ++b) {  // a run continued on a larger engine keeps every trajectory's back-off
    cert_wait_[b] = src.cert_wait_[(size_t)first + b];
    cert_back_[b] = src.cert_back_[(size_t)first + b];
  }
  return TJM_OK;
}

int Engine::export_state(int set, int b, double* out, int* bonds) {
  if (!bound_ || b < 0 || b >= B) return TJM_ERR_ARG;
  StateSet& S = sets[set];
  TJM_HIP_CHECK(hipMemcpyAsync(bonds, S.chi + (size_t)b * (L + 1), (L + 1) * sizeof(int), hipMemcpyDeviceToHost, stream));
  size_t total = 0;
  for (int i = 0; i < L; ++i) total += (size_t)a_b0_[i];
  std::vector<cplx> buf(total);
  cplx* dst = buf.data();
  for (int i = 0; i < L; ++i) {
    TJM_HIP_CHECK(hipMemcpyAsync(dst, S.A[i] + (size_t)b * a_b0_[i], (size_t)a_b0_[i] * sizeof(cplx), hipMemcpyDeviceToHost, stream));
    dst += a_b0_[i];
  }
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  to_host_c(out, buf.data(), buf.size());
  return TJM_OK;
}

int Engine::set_uniforms(const double* host_u, int n_per_traj) {
  uni_host_.assign(host_u, host_u + (size_t)B * n_per_traj);
  n_uniform_ = n_per_traj;
  cursor_.assign(B, 0);
  return TJM_OK;
}

int Engine::reset_cursor() {
  cursor_.assign(B, 0);
  return TJM_OK;
}

int Engine::bond_dims(int set, int* host_chi) {
  TJM_HIP_CHECK(hipMemcpyAsync(host_chi, sets[set].chi, (size_t)B * (L + 1) * sizeof(int), hipMemcpyDeviceToHost, stream));
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  return TJM_OK;
}

// ------------------------------------------------------------------------------------------
// contractions
// ------------------------------------------------------------------------------------------
static GemmDesc blank_gemm() {
  GemmDesc g;
  std::memset(&g, 0, sizeof(g));
  g.nks = 1; g.nb0 = 1; g.nb1 = 1; g.nb2 = 1;
  return g;
}

int Engine::merge_tensor_layout(StateSet& S, int i, cplx* out, long out_b0, const int* ids, int nb0) {
  const int ca = cap[i], cmid = cap[i + 1], cc = cap[i + 2];
  GemmDesc g = blank_gemm();
  g.A = S.A[i]; g.B = S.A[i + 1]; g.C = out;
  g.M = ca; g.K = cmid; g.N = cc;
  g.a_rs = cmid; g.a_cs = 1; g.b_rs = cc; g.b_cs = 1; g.c_rs = cc;
  g.nb0 = nb0; g.nb1 = d; g.nb2 = d;
  g.a_b0 = a_b0_[i]; g.a_b1 = (long)ca * cmid; g.a_b2 = 0;
  g.b_b0 = a_b0_[i + 1]; g.b_b1 = 0; g.b_b2 = (long)cmid * cc;
  g.c_b0 = out_b0; g.c_b1 = (long)d * ca * cc; g.c_b2 = (long)ca * cc;
  g.ids = ids;
  return gemm(g);
}

int Engine::merge_matrix_layout(StateSet& S, int i, const int* ids, int nb0) {
  const int ca = cap[i], cmid = cap[i + 1], cc = cap[i + 2];
  GemmDesc g = blank_gemm();
  g.A = S.A[i]; g.B = S.A[i + 1]; g.C = theta;
  g.M = d * ca; g.K = cmid; g.N = cc;
  g.a_rs = cmid; g.a_cs = 1; g.b_rs = cc; g.b_cs = 1; g.c_rs = (long)d * cc;
  g.nb0 = nb0; g.nb1 = d;
  g.a_b0 = a_b0_[i]; g.b_b0 = a_b0_[i + 1]; g.b_b1 = (long)cmid * cc;
  g.c_b0 = theta_b0; g.c_b1 = cc;
  g.ids = ids;
  return gemm(g);
}

int Engine::heff_apply(const cplx* x, long x_b0, int P, int ca, int cb, const cplx* Lenv, long l_b0, int Dl, const cplx* Renv,
                       long r_b0, int Dr, const cplx* Wm, cplx* y, long y_b0, int nb0, const int* ids, const int* active, int lch, int rch) {
  int rc;
  const bool fused12 = heff_stage12_fits(P, ca, cb, Dl, Dr, rch);
  // Direct form (round 5): when no entry of W couples a non-identity left channel to a non-identity right channel and the rows of
  // the non-identity left channels are monomial, T2[o][a][l][B] = coef(l, o) x[perm(l, o)][a][B] - the third stage reads x itself
  // (block perm, factor coef: GemmDesc::b_perm / coef) and T2 is neither written nor read: 7.3 instead of 10.3 MB of HBM traffic per
  // apply and trajectory at chi = 128 (these products run at 4 TB/s: they are bandwidth-bound).  Switch: TJM_NO_DIRECT_HEFF.
  static const bool no_direct = getenv("TJM_NO_DIRECT_HEFF") != nullptr;
  const DirectForm* df = nullptr;
  if (!no_direct && fused12 && lch >= 0 && rch >= 0 && Dl >= 2 && gemm4_serves(ca, cb, ca)) df = direct_form(Wm, P, Dl, Dr, lch, rch);
  if (df) ++stat_direct_applies;
  if (fused12) {  // stages 1 and 2 in one kernel: T1 stays in the accumulators of the GEMM tile (tjm_gemm.hip)
    HeffStage12Desc q;
    q.x = x; q.x_b0 = x_b0; q.R = Renv; q.r_b0 = r_b0; q.Wm = Wm; q.T2 = T2; q.t_b0 = t_b0; q.y = y; q.y_b0 = y_b0;
    q.P = P; q.ca = ca; q.cb = cb; q.Dl = Dl; q.Dr = Dr; q.rch = rch; q.lch = lch; q.nb0 = nb0; q.ids = ids; q.active = active;
    q.skip_t2 = df ? 1 : 0;
    if ((rc = launch_heff_stage12(q, stream)) != TJM_OK) return rc;
  }
  if (!fused12) {  // T1[(p,a),(r,B)] = x[(p,a),b] R[b,(r,B)]
    GemmDesc g = blank_gemm();
    g.A = x; g.B = Renv; g.C = T1;
    g.M = P * ca; g.K = cb; g.N = Dr * cb;
    g.a_rs = cb; g.a_cs = 1; g.b_rs = (long)Dr * cb; g.b_cs = 1; g.c_rs = (long)Dr * cb;
    g.nb0 = nb0; g.a_b0 = x_b0; g.b_b0 = r_b0; g.c_b0 = t_b0;
    g.ids = ids; g.active = active;
    if (rch >= 0) {  // R[b, rch, B] = delta(b, B): that slice of T1 is x itself (the MPO stage reads it there), the GEMM covers the other Dr - 1 channels
      g.N = (Dr - 1) * cb;
      if (rch == 0) { g.B = Renv + cb; g.C = T1 + cb; }
    }
    if ((rc = gemm(g)) != TJM_OK) return rc;
  }
  if (!fused12) {  // T2[o][a][l][B] = sum_{p,r} W[(o,l),(p,r)] T1[p][a][r][B]
    MpoApplyDesc m;
    m.in = T1; m.out = T2; m.Wm = Wm; m.P = P; m.din = Dr; m.dout = Dl; m.na = ca; m.nB = cb;
    m.in_sp = (long)ca * Dr * cb; m.in_sb = cb; m.in_sa = (long)Dr * cb;
    m.out_sp = (long)ca * Dl * cb; m.out_sb = cb; m.out_sa = (long)Dl * cb;
    m.in_b0 = t_b0; m.out_b0 = t_b0; m.nb0 = nb0; m.ids = ids; m.active = active;
    if (rch >= 0) { m.in_alt = x; m.in_alt_ch = rch; m.in_alt_b0 = x_b0; m.in_alt_sp = (long)ca * cb; m.in_alt_sa = cb; }
    if (lch >= 0) { m.out_alt = y; m.out_alt_ch = lch; m.out_alt_b0 = y_b0; m.out_alt_sp = (long)ca * cb; m.out_alt_sa = cb; }
    if ((rc = launch_mpo_apply(m, stream)) != TJM_OK) return rc;
  }
  {  // y[o][A][B] = sum_{(a,l)} L[(a,l)][A] T2[o][(a,l)][B]
    GemmDesc g = blank_gemm();
    g.A = Lenv; g.B = T2; g.C = y;
    g.M = ca; g.K = ca * Dl; g.N = cb;
    g.a_rs = 1; g.a_cs = ca; g.b_rs = cb; g.b_cs = 1; g.c_rs = cb;
    g.nb0 = nb0; g.nb1 = P;
    g.a_b0 = l_b0; g.b_b0 = t_b0; g.b_b1 = (long)ca * Dl * cb; g.c_b0 = y_b0; g.c_b1 = (long)ca * cb;
    g.ids = ids; g.active = active;
    if (lch >= 0) {  // L[a, lch, A] = delta(a, A): the MPO stage wrote that channel straight into y, the other Dl - 1 are a K-split sum on top of it
      const long first = (lch == 0) ? 1 : 0;
      g.A = Lenv + first * ca; g.B = T2 + first * cb;
      g.K = ca; g.a_cs = (long)Dl * ca; g.b_rs = (long)Dl * cb;
      g.nks = Dl - 1; g.a_ks = ca; g.b_ks = cb;
      g.accumulate = 1;
      if (df) {  // B = x[perm(l, o)] instead of T2[o][.][l][.]
        g.B = x; g.b_b0 = x_b0; g.b_b1 = 0; g.b_ks = 0; g.b_rs = cb;
        g.b_perm = df->perm; g.coef = df->coef; g.b_perm_stride = (long)ca * cb;
      }
    }
    static const bool no_dot_fusion = getenv("TJM_NO_DOT_EPILOGUE") != nullptr;
    const int tiles = ((ca + 63) / 64) * ((cb + 63) / 64) * P;
    if (!no_dot_fusion && dot_req_.v == x && x_b0 == y_b0 && ca >= 64 && cb >= 64 && tiles <= TJM_MAX_PART) {
      // the Lanczos coefficient <x, H x> in the epilogue of this product: y is final there, x has y's layout
      g.dot_with = x; g.dot_part = part1_; g.dot_ld = tiles;
      dot_req_.served = true;
      dot_req_.nblk1 = tiles;
    }
    if ((rc = gemm(g)) != TJM_OK) return rc;
  }
  return TJM_OK;
}

// Which channel of the two environments of a Krylov call is the identity matrix for EVERY trajectory of the call (see
// env_identity_check_kernel)?  One small kernel per environment and one host read per call; -1 when none is.
int Engine::identity_channels(int ca, int cb, const cplx* Lenv, long l_b0, int Dl, const cplx* Renv, long r_b0, int Dr, int nb0, const int* ids,
                              const int* chi_l, const int* chi_r, int* lch, int* rch) {
  *lch = -1;
  *rch = -1;
  static const bool off = getenv("TJM_NO_IDENTITY_CHANNELS") != nullptr;
  if (off || ca < 32 || cb < 32 || (Dl < 2 && Dr < 2)) return TJM_OK;
  int rc;
  int* flags = ks.n_active + 8;
  // The tolerance is a rounding bound, not an approximation the results lean on: an environment channel <A|A> over isometric sites
  // deviates from the identity by what the isometries and the GEMMs lose to rounding (about 1e-15 per entry and site in fp64, 1e-13
  // for a site whose isometry comes from a Jacobi sweep; 1e-6 in complex64), i.e. substituting the exact identity changes an H_eff
  // apply by no more than evaluating the GEMM would.  TJM_IDENTITY_TOL overrides it (diagnostic).
#ifdef TJM_F32
  static const real tol = getenv("TJM_IDENTITY_TOL") ? (real)atof(getenv("TJM_IDENTITY_TOL")) : real(3e-5);
#else
  static const real tol = getenv("TJM_IDENTITY_TOL") ? (real)atof(getenv("TJM_IDENTITY_TOL")) : real(1e-12);
#endif
  TJM_HIP_CHECK(hipMemsetAsync(flags, 0, 4 * sizeof(int), stream));
  if (Dl >= 2 && (rc = launch_env_identity_check(Lenv, l_b0, ca, Dl, chi_l, L + 1, tol, flags, nb0, ids, stream)) != TJM_OK) return rc;
  if (Dr >= 2 && (rc = launch_env_identity_check(Renv, r_b0, cb, Dr, chi_r, L + 1, tol, flags + 2, nb0, ids, stream)) != TJM_OK) return rc;
  TJM_HIP_CHECK(hipMemcpyAsync(h_pinned_ + 12, flags, 4 * sizeof(int), hipMemcpyDeviceToHost, stream));
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  const int* f = h_pinned_ + 12;
  if (Dl >= 2) *lch = !f[1] ? Dl - 1 : (!f[0] ? 0 : -1);
  if (Dr >= 2) *rch = !f[2] ? 0 : (!f[3] ? Dr - 1 : -1);
  ++stat_ident_calls;
  stat_ident_hits += (*lch >= 0) + (*rch >= 0);
  return TJM_OK;
}

int Engine::env_left(StateSet& S, int i, const int* ids, int nb0) {
  return env_left_at(S.A[i], a_b0_[i], cap[i], cap[i + 1], Dm[i], Dm[i + 1], Lenv_[i], l_b0_[i], WenvL_[i], Lenv_[i + 1], l_b0_[i + 1],
                     nb0 < 0 ? B : nb0, ids);
}

// update_left_environment (primitives.py:77-107) on explicit tensors: A [nb][d][ca][cb], Lin [nb][ca][Dl][ca] -> Lout [nb][cb][Dr][cb]
int Engine::env_left_at(const cplx* A, long a_b0, int ca, int cb, int Dl, int Dr, const cplx* Lin, long lin_b0, const cplx* WenvL, cplx* Lout,
                        long lout_b0, int nb, const int* ids) {
  int rc;
  Region prof(*this, PROF_ENV);
  ++stat_env_updates;
  {  // T1[(a,l),(o,B)] = L[(a,l),A] conj(A_i[o][A][B])
    GemmDesc g = blank_gemm();
    g.A = Lin; g.B = A; g.C = T1;
    g.M = ca * Dl; g.K = ca; g.N = cb;
    g.a_rs = ca; g.a_cs = 1; g.b_rs = cb; g.b_cs = 1; g.c_rs = (long)d * cb; g.conjB = 1;
    g.nb0 = nb; g.nb1 = d;
    g.a_b0 = lin_b0; g.b_b0 = a_b0; g.b_b1 = (long)ca * cb; g.c_b0 = t_b0; g.c_b1 = cb; g.ids = ids;
    if ((rc = gemm(g)) != TJM_OK) return rc;
  }
  {  // T2[p][a][r][B] = sum_{o,l} W[o,p,l,r] T1[a][l][o][B]
    MpoApplyDesc m;
    m.in = T1; m.out = T2; m.Wm = WenvL; m.P = d; m.din = Dl; m.dout = Dr; m.na = ca; m.nB = cb;
    m.in_sp = cb; m.in_sb = (long)d * cb; m.in_sa = (long)Dl * d * cb;
    m.out_sp = (long)ca * Dr * cb; m.out_sb = cb; m.out_sa = (long)Dr * cb;
    m.in_b0 = t_b0; m.out_b0 = t_b0; m.nb0 = nb; m.ids = ids; m.active = nullptr;
    if ((rc = launch_mpo_apply(m, stream)) != TJM_OK) return rc;
  }
  {  // L'[b,(r,B)] = sum_{(p,a)} A_i[(p,a),b] T2[(p,a),(r,B)]
    GemmDesc g = blank_gemm();
    g.A = A; g.B = T2; g.C = Lout;
    g.M = cb; g.K = d * ca; g.N = Dr * cb;
    g.a_rs = 1; g.a_cs = cb; g.b_rs = (long)Dr * cb; g.b_cs = 1; g.c_rs = (long)Dr * cb;
    g.nb0 = nb; g.a_b0 = a_b0; g.b_b0 = t_b0; g.c_b0 = lout_b0; g.ids = ids;
    if ((rc = gemm(g)) != TJM_OK) return rc;
  }
  return TJM_OK;
}

int Engine::env_right(StateSet& S, int i, const int* ids, int nb0) {
  // Renv_[i-1] (bond cap[i], Dm[i]) from Renv_[i] and A_i
  return env_right_at(S.A[i], a_b0_[i], cap[i], cap[i + 1], Dm[i], Dm[i + 1], Renv_[i], r_b0_[i], W_[i], Renv_[i - 1], r_b0_[i - 1],
                      nb0 < 0 ? B : nb0, ids);
}

// update_right_environment (primitives.py:110-136) on explicit tensors: A [nb][d][ca][cb], Rin [nb][cb][Dr][cb] -> Rout [nb][ca][Dl][ca]
int Engine::env_right_at(const cplx* A, long a_b0, int ca, int cb, int Dl, int Dr, const cplx* Rin, long rin_b0, const cplx* Wm, cplx* Rout,
                         long rout_b0, int nb, const int* ids) {
  int rc;
  Region prof(*this, PROF_ENV);
  ++stat_env_updates;
  {  // T1[(p,a),(r,B)] = A_i[(p,a),b] R[b,(r,B)]
    GemmDesc g = blank_gemm();
    g.A = A; g.B = Rin; g.C = T1;
    g.M = d * ca; g.K = cb; g.N = Dr * cb;
    g.a_rs = cb; g.a_cs = 1; g.b_rs = (long)Dr * cb; g.b_cs = 1; g.c_rs = (long)Dr * cb;
    g.nb0 = nb; g.a_b0 = a_b0; g.b_b0 = rin_b0; g.c_b0 = t_b0; g.ids = ids;
    if ((rc = gemm(g)) != TJM_OK) return rc;
  }
  {  // T2[a][l][o][B] = sum_{p,r} W[o,p,l,r] T1[p][a][r][B]
    MpoApplyDesc m;
    m.in = T1; m.out = T2; m.Wm = Wm; m.P = d; m.din = Dr; m.dout = Dl; m.na = ca; m.nB = cb;
    m.in_sp = (long)ca * Dr * cb; m.in_sb = cb; m.in_sa = (long)Dr * cb;
    m.out_sp = cb; m.out_sb = (long)d * cb; m.out_sa = (long)Dl * d * cb;
    m.in_b0 = t_b0; m.out_b0 = t_b0; m.nb0 = nb; m.ids = ids; m.active = nullptr;
    if ((rc = launch_mpo_apply(m, stream)) != TJM_OK) return rc;
  }
  {  // R'[(a,l),A] = sum_o sum_B T2[(a,l),(o,B)] conj(A_i[o][A][B])
    GemmDesc g = blank_gemm();
    g.A = T2; g.B = A; g.C = Rout;
    g.M = ca * Dl; g.K = cb; g.N = ca;
    g.a_rs = (long)d * cb; g.a_cs = 1; g.b_rs = 1; g.b_cs = cb; g.c_rs = ca; g.conjB = 1;
    g.nks = d; g.a_ks = cb; g.b_ks = (long)ca * cb;
    g.nb0 = nb; g.a_b0 = t_b0; g.b_b0 = a_b0; g.c_b0 = rout_b0; g.ids = ids;
    if ((rc = gemm(g)) != TJM_OK) return rc;
  }
  return TJM_OK;
}

// ------------------------------------------------------------------------------------------
// Krylov exponential of a site tensor held in V[:, 0]
// ------------------------------------------------------------------------------------------
int Engine::krylov_core(const ApplyFn& apply, int n, double dt_, const int* nloc_dev, cplx* out, long out_b0, int n0, int n1, int n2, int n3,
                        long o0, long o1, long o2, int nb0, const int* ids) {
  int rc, nblk = 1;
  Region prof(*this, PROF_KRYLOV);
  ++stat_krylov_calls;
  static const bool sync_each = getenv("TJM_KRYLOV_SYNC_EACH") != nullptr;
  bool pipelined = !sync_each;
  if (pipelined && !krylov_ev_[0]) {
    if (hipEventCreate(&krylov_ev_[0]) != hipSuccess || hipEventCreate(&krylov_ev_[1]) != hipSuccess) { (void)hipGetLastError(); pipelined = false; }
  }
  // one counter of still-active trajectories PER ITERATION (slots 16 ... 16 + mmax of the 64-int block behind ks.n_active), zeroed by one
  // fill for the whole call instead of one fill per iteration (round 6: the fills were 13 % of all launches of a step)
  const bool slots = 16 + mmax + 1 <= 64;
  TJM_HIP_CHECK(hipMemsetAsync(ks.n_active, 0, (slots ? 16 + mmax + 1 : 1) * sizeof(int), stream));
  if ((rc = launch_normsq_partial(V, v_b0, n, part2_, nb0, ids, nullptr, stream, &nblk)) != TJM_OK) return rc;
  if ((rc = launch_lanczos_init(ks, part2_, nblk, nb0, ids, stream)) != TJM_OK) return rc;
  // (no normalisation passes: the Krylov vectors stay unnormalised in V, their scales ks.svec ride along - round 5)
  for (int j = 0; j < mmax; ++j) {
    cplx* vj = V + (long)j * v_ld;
    cplx* w = V + (long)(j + 1) * v_ld;
    cplx* vjm1 = V + (long)(j > 0 ? j - 1 : 0) * v_ld;
    dot_req_.v = vj;
    dot_req_.served = false;
    rc = apply(vj, w, ks.status);
    dot_req_.v = nullptr;
    if (rc != TJM_OK) return rc;
    ++stat_matvecs;
    if (krylov_P_ == d * d) ++stat_matvecs2;
    int nblk1 = dot_req_.nblk1;  // <v_j, w>: per-tile partial sums from the epilogue of the apply's last GEMM, or a pass of its own
    if (!dot_req_.served) {
      if ((rc = launch_dot_partial(vj, w, v_b0, v_b0, n, part1_, nb0, ids, ks.status, stream, &nblk1)) != TJM_OK) return rc;
    }
    // (nblk: the grid of the vector kernels for n elements, from the norm pass above)
    if ((rc = launch_lanczos_axpy(w, vj, vjm1, v_b0, n, part1_, part2_, nblk, ks.beta, mmax, j, nb0, ids, ks.status, stream, ks.svec, nblk1)) != TJM_OK) return rc;
    KrylovState ksj = ks;
    if (slots) ksj.n_active = ks.n_active + 16 + j;
    else TJM_HIP_CHECK(hipMemsetAsync(ks.n_active, 0, sizeof(int), stream));
    if ((rc = launch_lanczos_finalize(ksj, part1_, part2_, nblk, j, dt_, krylov_tol, nloc_dev, nb0, ids, stream, nblk1)) != TJM_OK) return rc;
    if (!pipelined) {
      TJM_HIP_CHECK(hipMemcpyAsync(h_pinned_, ksj.n_active, sizeof(int), hipMemcpyDeviceToHost, stream));
      TJM_HIP_CHECK(hipStreamSynchronize(stream));
      if (*h_pinned_ == 0) break;
      continue;
    }
    // The count of still-active trajectories travels to the host behind the iteration; the host looks at it only after it has
    // queued the NEXT iteration, so the device never idles for the round trip.  When the count was zero, the iteration already
    // queued runs with every trajectory masked (ks.status) and changes nothing.
    TJM_HIP_CHECK(hipMemcpyAsync(h_pinned_ + 2 + (j & 1), ksj.n_active, sizeof(int), hipMemcpyDeviceToHost, stream));
    TJM_HIP_CHECK(hipEventRecord(krylov_ev_[j & 1], stream));
    if (j > 0) {
      TJM_HIP_CHECK(hipEventSynchronize(krylov_ev_[(j - 1) & 1]));
      if (h_pinned_[2 + ((j - 1) & 1)] == 0) break;
    }
    if (j + 1 == mmax) TJM_HIP_CHECK(hipStreamSynchronize(stream));
  }
  return launch_krylov_combine(V, v_b0, v_ld, ks, out, out_b0, n0, n1, n2, n3, o0, o1, o2, nb0, ids, stream);
}

int Engine::krylov_site(cplx* /*unused*/, int P, int ca, int cb, const cplx* Lenv, long l_b0, int Dl, const cplx* Renv, long r_b0,
                        int Dr, const cplx* Wm, double dt_, const int* nloc_dev, cplx* out, long out_b0, int n0, int n1, int n2,
                        int n3, long o0, long o1, long o2, int nb0, const int* ids, const int* chi_l, const int* chi_r) {
  if (krylov_small_fits(P, ca, cb, Dl, Dr, mmax, nb0)) {  // small bonds: contraction, recurrence, adaptive stop and combination in one kernel
    SmallKrylovDesc q;
    q.V = V; q.v_b0 = v_b0; q.v_ld = v_ld; q.P = P; q.ca = ca; q.cb = cb;
    q.Lenv = Lenv; q.l_b0 = l_b0; q.Dl = Dl; q.Renv = Renv; q.r_b0 = r_b0; q.Dr = Dr; q.Wm = Wm;
    q.dt = dt_; q.tol = krylov_tol; q.nloc = nloc_dev; q.mmax = mmax;
    q.out = out; q.out_b0 = out_b0; q.n1 = n1; q.n2 = n2; q.n3 = n3; q.o0 = o0; q.o1 = o1; q.o2 = o2;
    q.ids = ids; q.nb0 = nb0; q.matvecs = nullptr;
    q.chi_l = chi_l; q.chi_r = chi_r; q.chi_stride = L + 1;
    ++stat_krylov_calls;
    return launch_krylov_site_small(q, stream);
  }
  int lch = -1, rch = -1, rci;
  if ((rci = identity_channels(ca, cb, Lenv, l_b0, Dl, Renv, r_b0, Dr, nb0, ids, chi_l, chi_r, &lch, &rch)) != TJM_OK) return rci;
  ApplyFn f = [&](const cplx* x, cplx* y, const int* active) {
    return heff_apply(x, v_b0, P, ca, cb, Lenv, l_b0, Dl, Renv, r_b0, Dr, Wm, y, v_b0, nb0, ids, active, lch, rch);
  };
  krylov_P_ = P;
  const int rc = krylov_core(f, P * ca * cb, dt_, nloc_dev, out, out_b0, n0, n1, n2, n3, o0, o1, o2, nb0, ids);
  krylov_P_ = 0;
  return rc;
}

// project_bond (primitives.py:207-226): y[p][w] = sum L[u][a][p] C[u][v] R[v][a][w]
int Engine::bond_apply(const cplx* x, int cu, int cv, const cplx* Lenv, long l_b0, const cplx* Renv, long r_b0, int D, cplx* y,
                       const int* active, int nb0, const int* ids) {
  if (nb0 < 0) nb0 = B;
  int rc;
  {  // T[u][(a,w)] = C[u][v] R[v][(a,w)]
    GemmDesc g = blank_gemm();
    g.A = x; g.B = Renv; g.C = T1;
    g.M = cu; g.K = cv; g.N = D * cv;
    g.a_rs = cv; g.a_cs = 1; g.b_rs = (long)D * cv; g.b_cs = 1; g.c_rs = (long)D * cv;
    g.nb0 = nb0; g.ids = ids; g.a_b0 = v_b0; g.b_b0 = r_b0; g.c_b0 = t_b0; g.active = active;
    if ((rc = gemm(g)) != TJM_OK) return rc;
  }
  {  // y[p][w] = sum_{(u,a)} L[(u,a)][p] T[(u,a)][w]
    GemmDesc g = blank_gemm();
    g.A = Lenv; g.B = T1; g.C = y;
    g.M = cu; g.K = cu * D; g.N = cv;
    g.a_rs = 1; g.a_cs = cu; g.b_rs = cv; g.b_cs = 1; g.c_rs = cv;
    g.nb0 = nb0; g.ids = ids; g.a_b0 = l_b0; g.b_b0 = t_b0; g.c_b0 = v_b0; g.active = active;
    if ((rc = gemm(g)) != TJM_OK) return rc;
  }
  return TJM_OK;
}

// nloc[b] = P * chi[b][bl] * chi[b][br]   (actual local dimension; matrix_exponential.py:94 uses vec.size)
__global__ void nloc_kernel(const int* chi, int stride, int bl, int br, int P, int* nloc, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) nloc[b] = P * chi[(long)b * stride + bl] * chi[(long)b * stride + br];
}

// Has any truncation since the last clear asked for more singular values than the storage of its bond holds?  The states are then
// those of a run with max_bond_dim = chi_max rather than the requested one; the caller re-runs with a larger engine.
int Engine::capacity_overflow(int* host_flag, bool clear) {
  if (!bound_) return TJM_ERR_STATE;
  TJM_HIP_CHECK(hipMemcpyAsync(h_pinned_ + 8, overflow_, 2 * sizeof(int), hipMemcpyDeviceToHost, stream));
  if (clear) TJM_HIP_CHECK(hipMemsetAsync(overflow_, 0, sizeof(int), stream));
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  *host_flag = h_pinned_[8];
  if (h_pinned_[9] != 0) return TJM_ERR_NUMERIC;  // a fused small-bond factorisation did not converge (sticky)
  return TJM_OK;
}

int Engine::set_nloc(StateSet& S, int bl, int br, int P) {
  hipLaunchKernelGGL(nloc_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, S.chi, L + 1, bl, br, P, nloc_, B);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

int Engine::split(StateSet& S, int i, int dist, int mode, double thr, int maxb, int min_keep, const int* ids, int nb0) {
  SvdSplitDesc s;
  s.theta = theta; s.theta_b0 = theta_b0; s.ld_theta = d * cap[i + 2];
  s.m = d * cap[i]; s.n = d * cap[i + 2]; s.d = d;
  s.capL = cap[i]; s.capR = cap[i + 2]; s.capM = cap[i + 1];
  s.left = S.A[i]; s.right = S.A[i + 1]; s.left_b0 = a_b0_[i]; s.right_b0 = a_b0_[i + 1];
  s.distribution = dist; s.trunc_mode = mode; s.threshold = thr; s.max_bond = maxb; s.min_keep = min_keep;
  s.chiL = S.chi + i; s.chiR = S.chi + i + 2; s.chiM = S.chi + i + 1; s.chi_stride = L + 1;
  s.spectrum = nullptr; s.spec_ld = 0; s.nb0 = nb0; s.ids = ids;
  s.overflow = overflow_;
  int sweeps = 0;
  Region prof(*this, PROF_SVD);
  static const bool no_qr = getenv("TJM_NO_QR") != nullptr;
  static const bool force_large = getenv("TJM_FORCE_LARGE_SPLIT") != nullptr;
  const bool large = std::max(s.m, s.n) > 512 || (force_large && std::min(s.m, s.n) >= 32);  // bonds beyond 256: only the QR-preconditioned X-only variant holds the columns
  const bool use_qr = large || (!no_qr && dist != 2 && ids == nullptr && std::min(s.m, s.n) >= 64);  // the sqrt distribution is served by the plain split
  const int rc = use_qr ? svd_split_qr(s, svdw, qrw, stream, &sweeps, &mixw) : svd_split(s, svdw, stream, &sweeps);
  ++stat_svds;
  stat_svd_mats += nb0;
  stat_svd_sweeps += sweeps;
  return rc;
}

int Engine::two_site_update(StateSet& S, int i, double dt_, int dist, const int* ids, int nb0, bool capped) {
  if (nb0 < 0) nb0 = B;
  const int ca = cap[i], cc = cap[i + 2], P = d * d;
  int rc;
  if ((rc = merge_tensor_layout(S, i, V, v_b0, ids, nb0)) != TJM_OK) return rc;
  if ((rc = set_nloc(S, i, i + 2, P)) != TJM_OK) return rc;
  // result in matrix layout theta[(s,a),(t,c)] from tensor layout [s][t][a][c]
  if ((rc = krylov_site(nullptr, P, ca, cc, Lenv_[i], l_b0_[i], Dm[i], Renv_[i + 1], r_b0_[i + 1], Dm[i + 2], W2_[i], dt_, nloc_,
                        theta, theta_b0, d, d, ca, cc, (long)ca * d * cc, cc, (long)d * cc, nb0, ids, S.chi + i, S.chi + i + 2)) != TJM_OK) return rc;
  const int mk = (max_bond > 0) ? std::min(2, max_bond) : 2;  // get_min_keep (sweep_utils.py:33-44)
  // capped = false: split_tdvp(dynamic=True), no max_bond_dim in the truncation (sweep_utils.py:47-84)
  if ((rc = split(S, i, dist, trunc_mode, svd_threshold, capped ? max_bond : 0, mk, ids, nb0)) != TJM_OK) return rc;
  ++stat_site_updates;
  return TJM_OK;
}

int Engine::one_site_update(StateSet& S, int i, double dt_, const int* ids, int nb0) {
  if (nb0 < 0) nb0 = B;
  const int ca = cap[i], cb = cap[i + 1];
  int rc;
  TJM_HIP_CHECK(hipMemcpy2DAsync(V, (size_t)v_b0 * sizeof(cplx), S.A[i], (size_t)a_b0_[i] * sizeof(cplx), (size_t)a_b0_[i] * sizeof(cplx), B,
                                 hipMemcpyDeviceToDevice, stream));
  if ((rc = set_nloc(S, i, i + 1, d)) != TJM_OK) return rc;
  return krylov_site(nullptr, d, ca, cb, Lenv_[i], l_b0_[i], Dm[i], Renv_[i], r_b0_[i], Dm[i + 1], W_[i], dt_, nloc_, S.A[i], a_b0_[i], 1,
                     d, ca, cb, 0, (long)ca * cb, cb, nb0, ids, S.chi + i, S.chi + i + 1);
}

int Engine::sweep_2site(StateSet& S, double scale) {
  int rc;
  const double sdt = dt * scale;
  for (int i = 0; i < L - 2; ++i) {
    if ((rc = two_site_update(S, i, 0.5 * sdt, 0)) != TJM_OK) return rc;
    if ((rc = env_left(S, i)) != TJM_OK) return rc;
    if ((rc = one_site_update(S, i + 1, -0.5 * sdt)) != TJM_OK) return rc;
  }
  {
    const int i = L - 2;
    if ((rc = two_site_update(S, i, sdt, 1)) != TJM_OK) return rc;
    if ((rc = env_right(S, i + 1)) != TJM_OK) return rc;
  }
  for (int i = L - 3; i >= 0; --i) {
    if ((rc = one_site_update(S, i + 1, -0.5 * sdt)) != TJM_OK) return rc;
    if ((rc = two_site_update(S, i, 0.5 * sdt, 1)) != TJM_OK) return rc;
    if ((rc = env_right(S, i + 1)) != TJM_OK) return rc;
  }
  return TJM_OK;
}

// ---- helpers of the one-site sweep ---------------------------------------------------------------------------
// Z (column-major, bond-major rows) from a site tensor.  right: rows (a, p), columns b ; left: rows (b, p), columns a
__global__ __launch_bounds__(256) void site_to_z_kernel(const cplx* __restrict__ A, long a_b0, int d, int ca, int cb, int right, cplx* __restrict__ Z,
                                                       long z_b0, const int* ids) {
  const int b = ids ? ids[blockIdx.y] : blockIdx.y;
  const cplx* Ab = A + (long)b * a_b0;
  cplx* Zb = Z + (long)b * z_b0;
  const long total = (long)d * ca * cb;
  const int zr = right ? d * ca : d * cb;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int col = (int)(e / zr), rp = (int)(e % zr);
    const int bond = rp / d, p = rp % d;
    Zb[e] = right ? Ab[((long)p * ca + bond) * cb + col] : Ab[((long)p * ca + col) * cb + bond];
  }
}

// Cm from the R factor (upper triangle of Z).  right: Cm[j][c] = R[j][c] ; left: Cm[c][j] = R[j][c]   (ld = cdim)
__global__ __launch_bounds__(256) void r_to_bond_kernel(const cplx* __restrict__ Z, long z_b0, int zr, int zc, int right, cplx* __restrict__ Cm,
                                                       long c_b0, int cdim, const int* ids) {
  const int b = ids ? ids[blockIdx.y] : blockIdx.y;
  const cplx* Zb = Z + (long)b * z_b0;
  cplx* Cb = Cm + (long)b * c_b0;
  const long total = (long)cdim * cdim;
  const int kmax = zr < zc ? zr : zc;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int r = (int)(e / cdim), c = (int)(e % cdim);
    const int j = right ? r : c, col = right ? c : r;
    cplx v{0.0, 0.0};
    if (j < kmax && col < zc && j <= col) v = Zb[(long)col * zr + j];
    Cb[e] = v;
  }
}

__global__ __launch_bounds__(256) void z_identity_kernel(cplx* __restrict__ Z, long z_b0, int zr, int nc, const int* ids) {
  const int b = ids ? ids[blockIdx.y] : blockIdx.y;
  cplx* Zb = Z + (long)b * z_b0;
  const long total = (long)zr * nc;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x)
    Zb[e] = cplx{(e / zr == e % zr) ? real(1) : real(0), 0.0};
}

// new bond dimension after the thin QR (np.linalg.qr reduced: k = min(rows, cols)), and vec.size of the bond problem
__global__ void qr_bond_dims_kernel(int* chi, int stride, int i, int d, int right, int* nloc, int nb0, const int* ids) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nb0) return;
  const int b = ids ? ids[t] : t;
  int* c = chi + (long)b * stride;
  if (right) {
    const int k = min(d * c[i], c[i + 1]);
    nloc[b] = k * c[i + 1];
    c[i + 1] = k;
  } else {
    const int k = min(d * c[i + 1], c[i]);
    nloc[b] = c[i] * k;
    c[i] = k;
  }
}

// A_i = Q C with Q left-isometric (right = true, integrators.py:98-101) or A_i = C^T Q with Q right-isometric
// (right = false, integrators.py:128-136).  Householder QR; the bond matrix lands in Cm_ ([u][v] order of project_bond).
int Engine::qr_site(StateSet& S, int i, bool right, const int* ids, int nb0, bool absorb) {
  if (nb0 < 0) nb0 = B;
  const int ca = cap[i], cb = cap[i + 1];
  if (svd_shift_small_fits(d, ca, cb, !right)) {  // small bonds: factorisation, bond matrix, bond rule (and the shift) in one kernel
    SmallQrDesc q;
    q.site = S.A[i]; q.site_b0 = a_b0_[i]; q.bond = Cm_; q.nb = nullptr; q.nb_b0 = 0; q.cn = 0;
    if (absorb) {
      const int j = right ? i + 1 : i - 1;
      q.nb = S.A[j]; q.nb_b0 = a_b0_[j]; q.cn = right ? cap[i + 2] : cap[i - 1];
    }
    q.d = d; q.ca = ca; q.cb = cb;
    q.chi = S.chi + i; q.chi_stride = L + 1; q.nloc = nloc_; q.ids = ids; q.nb0 = nb0;
    return launch_qr_site_small(q, right, stream);
  }
  const int zr = right ? d * ca : d * cb;
  const int zc = right ? cb : ca;
  const int kmax = std::min(zr, zc);
  const int cdim = right ? cb : ca;
  int rc;
  const long total = (long)d * ca * cb;
  int gx = (int)((total + 1023) / 1024);
  if (gx < 1) gx = 1;
  if (gx > 128) gx = 128;
  hipLaunchKernelGGL(site_to_z_kernel, dim3(gx, nb0), dim3(256), 0, stream, S.A[i], a_b0_[i], d, ca, cb, right ? 1 : 0, qrw.Z, qrw.z_b0, ids);
  if ((rc = qr_factor(qrw, zr, zc, nb0, ids, stream)) != TJM_OK) return rc;
  {
    int g2 = (int)(((long)cdim * cdim + 1023) / 1024);
    if (g2 < 1) g2 = 1;
    hipLaunchKernelGGL(r_to_bond_kernel, dim3(g2, nb0), dim3(256), 0, stream, qrw.Z, qrw.z_b0, zr, zc, right ? 1 : 0, Cm_, (long)cdim * cdim, cdim, ids);
  }
  hipLaunchKernelGGL(qr_bond_dims_kernel, dim3((nb0 + 255) / 256), dim3(256), 0, stream, S.chi, L + 1, i, d, right ? 1 : 0, nloc_, nb0, ids);
  {
    int g3 = (int)(((long)zr * kmax + 1023) / 1024);
    if (g3 < 1) g3 = 1;
    if (g3 > 128) g3 = 128;
    hipLaunchKernelGGL(z_identity_kernel, dim3(g3, nb0), dim3(256), 0, stream, qrw.Z, qrw.z_b0, zr, kmax, ids);
  }
  if ((rc = qr_apply_q(qrw, zr, zc, qrw.Z, qrw.z_b0, kmax, nb0, ids, stream)) != TJM_OK) return rc;
  ExtractDesc x;
  x.out = S.A[i]; x.out_b0 = a_b0_[i]; x.row_off = 0; x.conj = 0; x.scale_mode = 0;
  if (right) {  // A_i[p][a][k] = Q[(a,p)][k]
    x.n_k = cb; x.o_k = 1; x.n_r1 = ca; x.n_r0 = d; x.o_r1 = cb; x.o_r0 = (long)ca * cb;
    rc = qr_scatter(qrw.Z, qrw.z_b0, zr, x, S.chi + i + 1, L + 1, nb0, ids, stream);
  } else {      // A_i[p][k][r] = Q[(r,p)][k]
    x.n_k = ca; x.o_k = cb; x.n_r1 = cb; x.n_r0 = d; x.o_r1 = 1; x.o_r0 = (long)ca * cb;
    rc = qr_scatter(qrw.Z, qrw.z_b0, zr, x, S.chi + i, L + 1, nb0, ids, stream);
  }
  TJM_HIP_CHECK(hipGetLastError());
  return rc;
}

// One symmetric 1TDVP sweep (integrators.py:44-158); bond dimensions are frozen apart from the thin-QR rule.
int Engine::sweep_1site(StateSet& S, double scale) {
  int rc;
  const double sdt = dt * scale;
  if ((rc = launch_identity_env(Renv_[L - 1], r_b0_[L - 1], cap[L], Dm[L], B, stream)) != TJM_OK) return rc;
  for (int i = L - 1; i >= 1; --i)
    if ((rc = env_right(S, i)) != TJM_OK) return rc;
  if ((rc = launch_identity_env(Lenv_[0], l_b0_[0], cap[0], Dm[0], B, stream)) != TJM_OK) return rc;
  for (int i = 0; i < L - 1; ++i) {
    const int cb = cap[i + 1], cc = cap[i + 2];
    if ((rc = one_site_update(S, i, 0.5 * sdt)) != TJM_OK) return rc;
    if ((rc = qr_site(S, i, true)) != TJM_OK) return rc;
    if ((rc = env_left(S, i)) != TJM_OK) return rc;
    // bond matrix C[u][v] (cb x cb padded) evolves backwards under project_bond(L_{i+1}, R_i)
    TJM_HIP_CHECK(hipMemcpy2DAsync(V, (size_t)v_b0 * sizeof(cplx), Cm_, (size_t)cb * cb * sizeof(cplx), (size_t)cb * cb * sizeof(cplx), B,
                                   hipMemcpyDeviceToDevice, stream));
    ApplyFn f = [&](const cplx* x, cplx* y, const int* active) {
      return bond_apply(x, cb, cb, Lenv_[i + 1], l_b0_[i + 1], Renv_[i], r_b0_[i], Dm[i + 1], y, active);
    };
    if ((rc = krylov_core(f, cb * cb, -0.5 * sdt, nloc_, Cm_, (long)cb * cb, 1, 1, cb, cb, 0, 0, cb, B, nullptr)) != TJM_OK) return rc;
    GemmDesc g = blank_gemm();  // T1[p][l][r] = C[l][x] A_{i+1}[p][x][r]
    g.A = Cm_; g.B = S.A[i + 1]; g.C = T1;
    g.M = cb; g.K = cb; g.N = cc;
    g.a_rs = cb; g.a_cs = 1; g.b_rs = cc; g.b_cs = 1; g.c_rs = cc;
    g.nb0 = B; g.nb1 = d; g.a_b0 = (long)cb * cb; g.b_b0 = a_b0_[i + 1]; g.b_b1 = (long)cb * cc; g.c_b0 = t_b0; g.c_b1 = (long)cb * cc;
    if ((rc = gemm(g)) != TJM_OK) return rc;
    if ((rc = copy_back(S.A[i + 1], a_b0_[i + 1], T1, t_b0, a_b0_[i + 1], nullptr, B)) != TJM_OK) return rc;
  }
  if ((rc = one_site_update(S, L - 1, sdt)) != TJM_OK) return rc;
  for (int i = L - 1; i >= 1; --i) {
    const int cz = cap[i - 1], ca = cap[i];
    if ((rc = qr_site(S, i, false)) != TJM_OK) return rc;
    if ((rc = env_right(S, i)) != TJM_OK) return rc;
    TJM_HIP_CHECK(hipMemcpy2DAsync(V, (size_t)v_b0 * sizeof(cplx), Cm_, (size_t)ca * ca * sizeof(cplx), (size_t)ca * ca * sizeof(cplx), B,
                                   hipMemcpyDeviceToDevice, stream));
    ApplyFn f = [&](const cplx* x, cplx* y, const int* active) {
      return bond_apply(x, ca, ca, Lenv_[i], l_b0_[i], Renv_[i - 1], r_b0_[i - 1], Dm[i], y, active);
    };
    if ((rc = krylov_core(f, ca * ca, -0.5 * sdt, nloc_, Cm_, (long)ca * ca, 1, 1, ca, ca, 0, 0, ca, B, nullptr)) != TJM_OK) return rc;
    GemmDesc g = blank_gemm();  // T1[(p,l)][r] = A_{i-1}[(p,l)][x] C^T[x][r]
    g.A = S.A[i - 1]; g.B = Cm_; g.C = T1;
    g.M = d * cz; g.K = ca; g.N = ca;
    g.a_rs = ca; g.a_cs = 1; g.b_rs = ca; g.b_cs = 1; g.c_rs = ca;
    g.nb0 = B; g.a_b0 = a_b0_[i - 1]; g.b_b0 = (long)ca * ca; g.c_b0 = t_b0;
    if ((rc = gemm(g)) != TJM_OK) return rc;
    if ((rc = copy_back(S.A[i - 1], a_b0_[i - 1], T1, t_b0, a_b0_[i - 1], nullptr, B)) != TJM_OK) return rc;
    if ((rc = one_site_update(S, i - 1, 0.5 * sdt)) != TJM_OK) return rc;
  }
  return TJM_OK;
}

int Engine::tdvp(int set) {
  cert_set_ = -1;
  if (!bound_) return TJM_ERR_STATE;
  if (tdvp_mode != 2 && tdvp_mode != 1) return TJM_ERR_NOT_IMPLEMENTED;
  StateSet& S = sets[set];
  int rc;
  if (tdvp_mode == 1 || L == 1) {  // a one-site chain falls back to 1TDVP (tdvp.py:96-98)
    for (int sw = 0; sw < tdvp_sweeps; ++sw)
      if ((rc = sweep_1site(S, 1.0 / tdvp_sweeps)) != TJM_OK) return rc;
    return TJM_OK;
  }
  // right environments (primitives.py:139-174), left boundary (integrators.py:186-193)
  if ((rc = launch_identity_env(Renv_[L - 1], r_b0_[L - 1], cap[L], Dm[L], B, stream)) != TJM_OK) return rc;
  for (int i = L - 1; i >= 1; --i)
    if ((rc = env_right(S, i)) != TJM_OK) return rc;
  if ((rc = launch_identity_env(Lenv_[0], l_b0_[0], cap[0], Dm[0], B, stream)) != TJM_OK) return rc;
  for (int s = 0; s < tdvp_sweeps; ++s)
    if ((rc = sweep_2site(S, 1.0 / tdvp_sweeps)) != TJM_OK) return rc;
  return TJM_OK;
}

// ------------------------------------------------------------------------------------------
// SVD centre shifts (mps.py:747-788): two-site merge + split, discarded_weight 1e-12, no cap
// ------------------------------------------------------------------------------------------
// Centre shift i -> i+1 when A_{i+1} is right-isometric: theta = A_i A_{i+1} has the singular values of the
// matrix A_i[(s,a), b], so the 256 x 256 two-site SVD reduces to a (d chi_l) x chi_r one:
//   A_i = U S V^H  ->  A_i <- U ,  A_{i+1} <- (S V^H) A_{i+1}        (same truncation rule, same state)
int Engine::svd_shift_right(StateSet& S, int i, const int* ids, int nb0) {
  const int ca = cap[i], cb = cap[i + 1], cc = cap[i + 2];
  int rc;
  Region prof(*this, PROF_SVD);
  if (svd_shift_small_fits(d, ca, cb, false)) {  // small bonds: factorisation, truncation and absorption in one kernel
    SmallShiftDesc q;
    q.site = S.A[i]; q.site_b0 = a_b0_[i]; q.nb = S.A[i + 1]; q.nb_b0 = a_b0_[i + 1];
    q.d = d; q.ca = ca; q.cb = cb; q.cn = cc;
    q.chi = S.chi + i; q.chi_stride = L + 1; q.threshold = 1e-12; q.min_keep = 1;
    q.ids = ids; q.nb0 = nb0; q.flags = overflow_;
    ++stat_svds; stat_svd_mats += nb0;
    return launch_svd_shift_small(q, false, stream);
  }
  JacobiSource src;
  src.src = S.A[i]; src.src_b0 = a_b0_[i]; src.rx = d * ca; src.ncols = cb; src.conj = 0; src.tri = 0;
  src.r_n0 = ca; src.s_r1 = (long)ca * cb; src.s_r0 = cb; src.c_n0 = cb; src.s_c1 = 0; src.s_c0 = 1;
  src.nb0 = nb0; src.ids = ids;
  TruncSpec tr;
  tr.trunc_mode = 0; tr.threshold = 1e-12; tr.max_bond = 0; tr.min_keep = 1;
  tr.chiA = S.chi + i; tr.mulA = d; tr.chiB = S.chi + i + 1; tr.mulB = 1; tr.chiOut = S.chi + i + 1; tr.chi_stride = L + 1;
  tr.spectrum = nullptr; tr.spec_ld = 0;
  JacobiShape sh;
  int sweeps = 0;
  // only X is rotated: U is the set of normalised rotated columns, and S V^H = U^H A_i comes from one GEMM on the input
  if ((rc = jacobi_solve(src, tr, svdw, stream, &sh, &sweeps, false)) != TJM_OK) return rc;
  ++stat_svds; stat_svd_mats += nb0; stat_svd_sweeps += sweeps;
  ExtractDesc xu;  // U[(s,a)][k] = X_final / sigma  (zero beyond keep), first into the temp: A_i is still needed
  xu.out = theta; xu.out_b0 = theta_b0; xu.n_k = cb; xu.o_k = 1; xu.n_r1 = 1; xu.n_r0 = d * ca; xu.o_r1 = 0; xu.o_r0 = cb;
  xu.row_off = 0; xu.conj = 0; xu.scale_mode = 2;
  if ((rc = svd_extract(xu, svdw, sh, S.chi + i + 1, L + 1, nb0, ids, stream)) != TJM_OK) return rc;
  {
    GemmDesc g = blank_gemm();  // G[k][j] = sum_{(s,a)} conj(U[(s,a)][k]) A_i[(s,a)][j]  (= sigma_k conj(V[j][k]))
    g.A = theta; g.B = S.A[i]; g.C = T2;
    g.M = cb; g.K = d * ca; g.N = cb;
    g.a_rs = 1; g.a_cs = cb; g.conjA = 1; g.b_rs = cb; g.b_cs = 1; g.c_rs = cb;
    g.nb0 = nb0; g.a_b0 = theta_b0; g.b_b0 = a_b0_[i]; g.c_b0 = t_b0;
    g.ids = ids;
    if ((rc = gemm(g)) != TJM_OK) return rc;
  }
  xu.out = S.A[i]; xu.out_b0 = a_b0_[i];
  if ((rc = svd_extract(xu, svdw, sh, S.chi + i + 1, L + 1, nb0, ids, stream)) != TJM_OK) return rc;
  GemmDesc g = blank_gemm();  // T1[t][k][c] = G[k][j] A_{i+1}[t][j][c]
  g.A = T2; g.B = S.A[i + 1]; g.C = T1;
  g.M = cb; g.K = cb; g.N = cc;
  g.a_rs = cb; g.a_cs = 1; g.b_rs = cc; g.b_cs = 1; g.c_rs = cc;
  g.nb0 = nb0; g.nb1 = d; g.a_b0 = t_b0; g.b_b0 = a_b0_[i + 1]; g.b_b1 = (long)cb * cc; g.c_b0 = t_b0; g.c_b1 = (long)cb * cc;
  g.ids = ids;
  if ((rc = gemm(g)) != TJM_OK) return rc;
  return copy_back(S.A[i + 1], a_b0_[i + 1], T1, t_b0, a_b0_[i + 1], ids, nb0);
}

// Centre shift i -> i-1 when A_{i-1} is left-isometric (mirror image of svd_shift_right):
//   A_i[a,(t,c)] = U S V^H  ->  A_i <- V^H ,  A_{i-1} <- A_{i-1} (U S)
int Engine::svd_shift_left(StateSet& S, int i, const int* ids, int nb0) {
  const int cz = cap[i - 1], ca = cap[i], cb = cap[i + 1];
  int rc;
  Region prof(*this, PROF_SVD);
  if (svd_shift_small_fits(d, ca, cb, true)) {
    SmallShiftDesc q;
    q.site = S.A[i]; q.site_b0 = a_b0_[i]; q.nb = S.A[i - 1]; q.nb_b0 = a_b0_[i - 1];
    q.d = d; q.ca = ca; q.cb = cb; q.cn = cz;
    q.chi = S.chi + i; q.chi_stride = L + 1; q.threshold = 1e-12; q.min_keep = 1;
    q.ids = ids; q.nb0 = nb0; q.flags = overflow_;
    ++stat_svds; stat_svd_mats += nb0;
    return launch_svd_shift_small(q, true, stream);
  }
  JacobiSource src;  // X = M^H : rows (t,c), columns a
  src.src = S.A[i]; src.src_b0 = a_b0_[i]; src.rx = d * cb; src.ncols = ca; src.conj = 1; src.tri = 0;
  src.r_n0 = cb; src.s_r1 = (long)ca * cb; src.s_r0 = 1; src.c_n0 = ca; src.s_c1 = 0; src.s_c0 = cb;
  src.nb0 = nb0; src.ids = ids;
  TruncSpec tr;
  tr.trunc_mode = 0; tr.threshold = 1e-12; tr.max_bond = 0; tr.min_keep = 1;
  tr.chiA = S.chi + i; tr.mulA = 1; tr.chiB = S.chi + i + 1; tr.mulB = d; tr.chiOut = S.chi + i; tr.chi_stride = L + 1;
  tr.spectrum = nullptr; tr.spec_ld = 0;
  JacobiShape sh;
  int sweeps = 0;
  // only X = M^H is rotated: its normalised columns are the right singular vectors V of M, and U S = M V is one GEMM
  if ((rc = jacobi_solve(src, tr, svdw, stream, &sh, &sweeps, false)) != TJM_OK) return rc;
  ++stat_svds; stat_svd_mats += nb0; stat_svd_sweeps += sweeps;
  ExtractDesc xt;  // Vt[(t,c)][k] = X_final / sigma into the temp (row-major, d*cb x ca)
  xt.out = theta; xt.out_b0 = theta_b0; xt.n_k = ca; xt.o_k = 1; xt.n_r1 = 1; xt.n_r0 = d * cb; xt.o_r1 = 0; xt.o_r0 = ca;
  xt.row_off = 0; xt.conj = 0; xt.scale_mode = 2;
  if ((rc = svd_extract(xt, svdw, sh, S.chi + i, L + 1, nb0, ids, stream)) != TJM_OK) return rc;
  {
    GemmDesc g = blank_gemm();  // G[a][k] = sum_{(t,c)} A_i[t][a][c] Vt[(t,c)][k]  (= U[a][k] sigma_k)
    g.A = S.A[i]; g.B = theta; g.C = T2;
    g.M = ca; g.K = cb; g.N = ca; g.nks = d;
    g.a_rs = cb; g.a_cs = 1; g.a_ks = (long)ca * cb; g.b_rs = ca; g.b_cs = 1; g.b_ks = (long)cb * ca; g.c_rs = ca;
    g.nb0 = nb0; g.a_b0 = a_b0_[i]; g.b_b0 = theta_b0; g.c_b0 = t_b0;
    g.ids = ids;
    if ((rc = gemm(g)) != TJM_OK) return rc;
  }
  ExtractDesc xv;  // A_i[t][k][c] = conj(X_final[(t,c)][k]) / sigma
  xv.out = S.A[i]; xv.out_b0 = a_b0_[i]; xv.n_k = ca; xv.o_k = cb; xv.n_r1 = d; xv.n_r0 = cb; xv.o_r1 = (long)ca * cb; xv.o_r0 = 1;
  xv.row_off = 0; xv.conj = 1; xv.scale_mode = 2;
  if ((rc = svd_extract(xv, svdw, sh, S.chi + i, L + 1, nb0, ids, stream)) != TJM_OK) return rc;
  GemmDesc g = blank_gemm();  // T1[s][z][k] = A_{i-1}[s][z][a] G[a][k]
  g.A = S.A[i - 1]; g.B = T2; g.C = T1;
  g.M = d * cz; g.K = ca; g.N = ca;
  g.a_rs = ca; g.a_cs = 1; g.b_rs = ca; g.b_cs = 1; g.c_rs = ca;
  g.nb0 = nb0; g.a_b0 = a_b0_[i - 1]; g.b_b0 = t_b0; g.c_b0 = t_b0;
  g.ids = ids;
  if ((rc = gemm(g)) != TJM_OK) return rc;
  return copy_back(S.A[i - 1], a_b0_[i - 1], T1, t_b0, a_b0_[i - 1], ids, nb0);
}

// ------------------------------------------------------------------------------------------
// Certified scalar dissipation.  With Pauli-only noise every local factor of apply_dissipation is a scalar (dissipation.py:117-119,
// 141, 156-157) and the sweep - centre to the last site and back by 2 (L - 1) truncating SVD shifts (mps.py:747-788, discarded weight
// 1e-12) - changes the state only where a shift truncates.  The right-going pass is run on scratch copies of the centre tensor
// (svd_shift_right_virtual: the state itself is not touched) and records, per trajectory, whether any bond would lose a singular
// value and the smallest squared singular value it met.  When nothing is truncated and that minimum times the total scalar factor
// is still above the threshold, neither pass of the reference's sweep truncates: both are gauge moves, the state the reference ends
// with is the input times the scalar, and this build leaves it at that (one scaling kernel instead of 63 more shifts).  Trajectories
// that do not certify take the sweep as before.  stochastic() uses the same certificate: a unitary jump on a certified state needs
// neither the QR walk to the last site nor the SVD sweep back (normalize("B", "SVD"), mps.py:815-839) - they, too, only move the gauge.
// The certificate travels with a checksum of the state (64-bit sum over all site tensors), so that any change between the two calls
// voids it.
// ------------------------------------------------------------------------------------------
__global__ void cert_update_kernel(const real* __restrict__ norms, int ncols_pad, const int* vchi_col, const int* chi_col, int stride, real* cert_min,
                                   int* cert_flag, int nb0, const int* ids) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nb0) return;
  const int b = ids ? ids[t] : t;
  const int keep = vchi_col[(long)b * stride], have = chi_col[(long)b * stride];
  if (keep != have) cert_flag[b] = 1;
  if (keep > 0) {
    const real s = norms[(long)b * ncols_pad + keep - 1];
    const real s2 = s * s;
    if (!(s2 >= cert_min[b])) cert_min[b] = s2;  // a NaN lowers it for good
  }
}

__global__ void cert_init_kernel(real* cert_min, int* cert_flag, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) { cert_min[b] = real(3.0e38); cert_flag[b] = 0; }
}

__global__ __launch_bounds__(256) void state_checksum_kernel(const cplx* __restrict__ A, long a_b0, unsigned long long* csum, const int* ids) {
  __shared__ unsigned long long sh[256];
  int b = blockIdx.y;
  if (ids) b = ids[b];
  const unsigned long long* w = reinterpret_cast<const unsigned long long*>(A + (long)b * a_b0);
  const long nwords = a_b0 * (long)(sizeof(cplx) / sizeof(unsigned long long));
  unsigned long long acc = 0;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < nwords; e += (long)gridDim.x * blockDim.x) acc += w[e] * (unsigned long long)(2 * e + 1);
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(&csum[b], sh[0]);
}

int Engine::state_checksum(int set, const int* ids, int n, unsigned long long* host_out) {
  StateSet& S = sets[set];
  TJM_HIP_CHECK(hipMemsetAsync(csum_, 0, (size_t)B * sizeof(unsigned long long), stream));
  for (int i = 0; i < L; ++i) {
    int gx = (int)((a_b0_[i] + 4095) / 4096);
    if (gx < 1) gx = 1;
    if (gx > 32) gx = 32;
    hipLaunchKernelGGL(state_checksum_kernel, dim3(gx, n), dim3(256), 0, stream, S.A[i] + 0, a_b0_[i], csum_ + 0, ids);
  }
  TJM_HIP_CHECK(hipGetLastError());
  TJM_HIP_CHECK(hipMemcpyAsync(host_out, csum_, (size_t)B * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  return TJM_OK;
}

// ---- the same certificate without a single SVD (round 5) ---------------------------------------------------------------------------
// What the virtual pass needs from bond k is ONE bit: does every squared singular value of the centre tensor clear the cut
// 1e-12 / scale^2 (then neither pass of the reference's sweep truncates there: the discarded-weight rule drops a value only while the
// running sum stays below 1e-12, so nothing goes when even the smallest one is above it).  The squared singular values of the centre
// tensor at bond k are the eigenvalues of its Gram matrix, and those Gram matrices obey the recursion of the left density
// environments: with the centre at site 0 and isometric tensors to its right,
//     G_0 = [1],   G_(k+1) = sum_s A_k[s]^H G_k A_k[s]     (C_k = M A_k with M^H M = G_k, whatever gauge M is in)
// - two small MFMA products per site, the recursion site_moments runs for the observables.  "lambda_min(G_k) >= cut" is the statement
// "G_k - cut I is positive definite": a Cholesky factorisation that meets no non-positive pivot (chol_pd_kernel, one workgroup per
// trajectory, the lower triangle in LDS).  The test only has to be SUFFICIENT (a trajectory that fails it takes the reference's
// sweep), so the cut carries an absolute margin for the rounding of the recursion and of the factorisation (2e-14: n eps ||G|| with
// ||G|| <= 1, against eigenvalues that matter at 1e-12; round 6, ADVICE r5: plus a term that follows the chain position, the size and
// the norm of the matrix, 4 (k + 1) n u ||G_k||_F with u = 2^-53 - the first-order bound of k + 1 recursion steps of inner products
// of length ~n and of the factorisation, 3e-13 at the end of the headline's chain; a false pass would skip a truncation the
// reference performs, a false fail only costs the reference's sweep for that trajectory).  Per trajectory-step: 63 x 2 products of 128^3 and 63 factorisations of
// 128 x 128 instead of 63 Jacobi SVDs of 256 x 128 (ten fp64 sweeps each in the evolved state).  fp64 build only: in complex64 the
// Gram matrix is not resolved at 1e-12.  TJM_CERT_SVD_PASS: the SVD pass of rounds 3 - 4.
// ||G||_F of the Hermitian matrix whose lower triangle is E (row-major, leading dimension ld), by the whole workgroup; the rounding
// margin of the positive-definiteness tests below is proportional to it (cert_pass_gram)
__device__ inline real gram_fro(const cplx* __restrict__ Eb, int ld, int n, real* s_red) {
  const int tid = threadIdx.x, nt = blockDim.x;
  real acc = 0.0;
  for (long e = tid; e < (long)n * n; e += nt) {
    const int i = (int)(e / n), j = (int)(e % n);
    if (j > i) continue;
    const cplx v = Eb[(long)i * ld + j];
    const real a = v.x * v.x + ((i == j) ? real(0.0) : v.y * v.y);
    acc += (i == j) ? a : a + a;
  }
  s_red[tid] = acc;
  __syncthreads();
  for (int h = nt >> 1; h > 0; h >>= 1) {
    if (tid < h) s_red[tid] += s_red[tid + h];
    __syncthreads();
  }
  const real f = sqrt(s_red[0]);
  __syncthreads();
  return f;
}

__global__ __launch_bounds__(256) void chol_pd_kernel(const cplx* __restrict__ E, long e_b0, int ld, const int* __restrict__ chi, int chi_stride, real cut0,
                                                     real rel_margin, int* __restrict__ flag, const int* ids) {
  extern __shared__ real chol_smem[];
  cplx* Lp = reinterpret_cast<cplx*>(chol_smem);  // packed lower triangle, row-major: (i, j <= i) at i (i + 1) / 2 + j
  __shared__ real s_piv;
  __shared__ real s_red[256];
  int b = blockIdx.x;
  if (ids) b = ids[b];
  const int n = chi[(long)b * chi_stride];
  const cplx* Eb = E + (long)b * e_b0;
  const int tid = threadIdx.x;
  const int ntri = n * (n + 1) / 2;
  const real cut = cut0 + rel_margin * gram_fro(Eb, ld, n, s_red);
  for (int e = tid; e < ntri; e += 256) {
    // row i of entry e: largest i with i (i + 1) / 2 <= e
    int i = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
    while ((i + 1) * (i + 2) / 2 <= e) ++i;
    while (i * (i + 1) / 2 > e) --i;
    const int j = e - i * (i + 1) / 2;
    cplx v = Eb[(long)i * ld + j];
    if (i == j) { v.x -= cut; v.y = 0.0; }
    Lp[e] = v;
  }
  __syncthreads();
  bool ok = true;
  for (int j = 0; j < n; ++j) {
    if (tid == 0) s_piv = Lp[j * (j + 1) / 2 + j].x;
    __syncthreads();
    const real piv = s_piv;
    if (!(piv > real(0.0))) { ok = false; break; }  // uniform: every thread reads the same pivot (a NaN fails too)
    const real inv = real(1.0) / sqrt(piv);
    // column j below the diagonal, scaled; every thread keeps its own copy of what it needs: entry (i, j) for its rows
    for (int i = j + 1 + tid; i < n; i += 256) {
      cplx& v = Lp[i * (i + 1) / 2 + j];
      v.x *= inv; v.y *= inv;
    }
    __syncthreads();
    // trailing update: (i, k) -= L(i, j) conj(L(k, j)) for j < k <= i; rows dealt round-robin, a row's entries by the threads of a
    // group of 16
    const int grp = tid >> 4, gl = tid & 15;
    for (int i = j + 1 + grp; i < n; i += 16) {
      const cplx lij = Lp[i * (i + 1) / 2 + j];
      for (int k = j + 1 + gl; k <= i; k += 16) {
        const cplx lkj = Lp[k * (k + 1) / 2 + j];
        cplx& v = Lp[i * (i + 1) / 2 + k];
        v.x -= lij.x * lkj.x + lij.y * lkj.y;
        v.y -= lij.y * lkj.x - lij.x * lkj.y;
      }
    }
    __syncthreads();
  }
  if (!ok && tid == 0) flag[b] = 1;
}

// The same test for bonds above 128, where the packed triangle (264 KB at 256 x 256) does not fit the LDS: a blocked right-looking
// Cholesky factorisation of a working copy in global memory (W: the lower triangle of G - cut I, row-major, leading dimension ld), one
// workgroup per trajectory.  Per block column of 32: the diagonal block is factored in LDS (32 sequential steps), the panel below it is
// solved against it row by row (X = A L^-H, a thread per row, L in LDS) and staged in LDS, the trailing triangle gets X X^H taken off.
// n^3 / 3 complex multiply-adds per matrix, 8 trips of the trailing triangle through L2 at n = 256.  Same verdict as chol_pd_kernel:
// flag[b] = 1 at the first non-positive pivot.
constexpr int CHB = 32;
__global__ __launch_bounds__(1024) void chol_pd_blocked_kernel(const cplx* __restrict__ E, long e_b0, int ld, cplx* __restrict__ Wk, long w_b0,
                                                              const int* __restrict__ chi, int chi_stride, real cut0, real rel_margin,
                                                              int* __restrict__ flag, const int* ids) {
  extern __shared__ real cholb_smem[];
  cplx* sD = reinterpret_cast<cplx*>(cholb_smem);  // [CHB][CHB + 1] diagonal block
  cplx* sX = sD + CHB * (CHB + 1);                 // [rows below][CHB + 1] solved panel
  __shared__ real s_piv;
  __shared__ int s_bad;
  __shared__ real s_red[1024];
  int b = blockIdx.x;
  if (ids) b = ids[b];
  const int n = chi[(long)b * chi_stride];
  const cplx* Eb = E + (long)b * e_b0;
  cplx* W = Wk + (long)b * w_b0;
  const int tid = threadIdx.x, nt = blockDim.x;
  const real cut = cut0 + rel_margin * gram_fro(Eb, ld, n, s_red);
  for (long e = tid; e < (long)n * n; e += nt) {  // working copy of the lower triangle, the cut off the diagonal
    const int i = (int)(e / n), j = (int)(e % n);
    if (j > i) continue;
    cplx v = Eb[(long)i * ld + j];
    if (i == j) { v.x -= cut; v.y = 0.0; }
    W[(long)i * ld + j] = v;
  }
  if (tid == 0) s_bad = 0;
  __syncthreads();
  for (int k0 = 0; k0 < n; k0 += CHB) {
    const int kb = (n - k0 < CHB) ? n - k0 : CHB;  // columns of this block
    const int below = n - k0 - kb;                  // rows under the diagonal block
    for (int e = tid; e < kb * kb; e += nt) {
      const int i = e / kb, j = e % kb;
      sD[i * (CHB + 1) + j] = (j <= i) ? W[(long)(k0 + i) * ld + k0 + j] : cplx{0.0, 0.0};
    }
    __syncthreads();
    for (int j = 0; j < kb; ++j) {  // unblocked factorisation of the diagonal block
      if (tid == 0) s_piv = sD[j * (CHB + 1) + j].x;
      __syncthreads();
      const real piv = s_piv;
      if (!(piv > real(0.0))) { if (tid == 0) s_bad = 1; break; }  // uniform (a NaN fails too)
      const real inv = real(1.0) / sqrt(piv);
      if (tid > j && tid < kb) { cplx& v = sD[tid * (CHB + 1) + j]; v.x *= inv; v.y *= inv; }
      if (tid == 0) sD[j * (CHB + 1) + j] = cplx{(real)sqrt(piv), 0};
      __syncthreads();
      for (int e = tid; e < kb * kb; e += nt) {
        const int i = e / kb, c = e % kb;
        if (c > j && c <= i) {
          const cplx lij = sD[i * (CHB + 1) + j], lcj = sD[c * (CHB + 1) + j];
          cplx& v = sD[i * (CHB + 1) + c];
          v.x -= lij.x * lcj.x + lij.y * lcj.y;
          v.y -= lij.y * lcj.x - lij.x * lcj.y;
        }
      }
      __syncthreads();
    }
    __syncthreads();
    if (s_bad) break;
    // panel below: row r of X solves x L^H = a, x_c = (a_c - sum_{k<c} x_k conj(L[c][k])) / L[c][c]
    for (int r = tid; r < below; r += nt) {
      const cplx* arow = W + (long)(k0 + kb + r) * ld + k0;
      cplx* xr = sX + r * (CHB + 1);
      for (int c = 0; c < kb; ++c) {
        cplx acc = arow[c];
        for (int k = 0; k < c; ++k) {
          const cplx xk = xr[k], l = sD[c * (CHB + 1) + k];
          acc.x -= xk.x * l.x + xk.y * l.y;
          acc.y -= xk.y * l.x - xk.x * l.y;
        }
        const real dinv = real(1.0) / sD[c * (CHB + 1) + c].x;
        xr[c] = cplx{acc.x * dinv, acc.y * dinv};
      }
    }
    __syncthreads();
    // trailing triangle: (i, j <= i) -= X_i . conj(X_j)
    const long ntri = (long)below * (below + 1) / 2;
    for (long e = tid; e < ntri; e += nt) {
      int i = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
      while ((long)(i + 1) * (i + 2) / 2 <= e) ++i;
      while ((long)i * (i + 1) / 2 > e) --i;
      const int j = (int)(e - (long)i * (i + 1) / 2);
      const cplx* xi = sX + i * (CHB + 1);
      const cplx* xj = sX + j * (CHB + 1);
      real ax = 0.0, ay = 0.0;
      for (int k = 0; k < kb; ++k) {
        ax += xi[k].x * xj[k].x + xi[k].y * xj[k].y;
        ay += xi[k].y * xj[k].x - xi[k].x * xj[k].y;
      }
      cplx& w = W[(long)(k0 + kb + i) * ld + k0 + kb + j];
      w.x -= ax;
      w.y -= ay;
    }
    __threadfence_block();
    __syncthreads();
  }
  if (s_bad && tid == 0) flag[b] = 1;
}

bool Engine::cert_gram_fits() const {
#ifdef TJM_F32
  return false;
#else
  static const bool svd_pass = getenv("TJM_CERT_SVD_PASS") != nullptr;
  if (svd_pass) return false;
  int cm = 1;
  for (int k = 0; k <= L; ++k) cm = cap[k] > cm ? cap[k] : cm;
  // up to 128: the packed triangle in LDS (chol_pd_kernel); up to 256: the blocked factorisation of a working copy (the panel of the
  // first block column, (256 - 32) x 33 complex, and the diagonal block have to fit the LDS; the copy lives in T2)
  return cm <= 256 && (size_t)cm * cm <= (size_t)t_b0;
#endif
}

__global__ void fill_cplx_kernel(cplx* p, cplx v, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// flags cert_flag_[b] for every listed trajectory one of whose bonds 1 ... L - 1 has a squared singular value below `cut`
int Engine::cert_pass_gram(StateSet& S, const int* ids, int nb0, double cut) {
  int rc;
  static std::atomic<bool> attr_set{false};
  if (!attr_set.load(std::memory_order_acquire)) {
    TJM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(chol_pd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
    TJM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(chol_pd_blocked_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
    attr_set.store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL(fill_cplx_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, E_, cplx{1.0, 0.0}, (long)B);  // G_0 = [1] (cap[0] = 1)
  cplx* E = E_;
  cplx* En = E2_;
  for (int i = 0; i < L - 1; ++i) {
    const int ca = cap[i], cb = cap[i + 1];
    {  // T[p][a][b] = sum_a' G[a][a'] A_i[p][a'][b]
      GemmDesc g = blank_gemm();
      g.A = E; g.B = S.A[i]; g.C = T1;
      g.M = ca; g.K = ca; g.N = cb;
      g.a_rs = ca; g.a_cs = 1; g.b_rs = cb; g.b_cs = 1; g.c_rs = cb;
      g.nb0 = nb0; g.ids = ids; g.nb1 = d;
      g.a_b0 = (long)ca * ca; g.b_b0 = a_b0_[i]; g.b_b1 = (long)ca * cb; g.c_b0 = t_b0; g.c_b1 = (long)ca * cb;
      if ((rc = gemm(g)) != TJM_OK) return rc;
    }
    {  // G'[b][b'] = sum_{(p,a)} conj(A_i[(p,a),b]) T[(p,a),b']
      GemmDesc g = blank_gemm();
      g.A = S.A[i]; g.B = T1; g.C = En;
      g.M = cb; g.K = d * ca; g.N = cb;
      g.a_rs = 1; g.a_cs = cb; g.b_rs = cb; g.b_cs = 1; g.c_rs = cb; g.conjA = 1;
      g.nb0 = nb0; g.ids = ids; g.a_b0 = a_b0_[i]; g.b_b0 = t_b0; g.c_b0 = (long)cb * cb;
      if ((rc = gemm(g)) != TJM_OK) return rc;
    }
    std::swap(E, En);
    const size_t lds = (size_t)cb * (cb + 1) / 2 * sizeof(cplx);
    static const bool force_blocked = getenv("TJM_CHOL_BLOCKED") != nullptr;  // diagnostic: the blocked kernel at every size
    // rounding margin of this bond's test: 4 (k + 1) n u ||G_k||_F (the kernels measure the norm)
    static const double margin_c = getenv("TJM_CERT_MARGIN_C") ? atof(getenv("TJM_CERT_MARGIN_C")) : 4.0;
    const real rel_margin = (real)(margin_c * (double)(i + 1) * (double)cb * 1.1102230246251565e-16);
    if (lds <= 140 * 1024 && !(force_blocked && cb >= 8))
      hipLaunchKernelGGL(chol_pd_kernel, dim3(nb0), dim3(256), lds, stream, E, (long)cb * cb, cb, S.chi + i + 1, L + 1, (real)cut, rel_margin, cert_flag_, ids);
    else {
      const size_t ldsb = ((size_t)CHB * (CHB + 1) + (size_t)(cb > CHB ? cb - CHB : 1) * (CHB + 1)) * sizeof(cplx);
      hipLaunchKernelGGL(chol_pd_blocked_kernel, dim3(nb0), dim3(1024), ldsb, stream, E, (long)cb * cb, cb, T2, t_b0, S.chi + i + 1, L + 1, (real)cut,
                         rel_margin, cert_flag_, ids);
      stat_cert_blocked += nb0;
    }
  }
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

// svd_shift_right on a scratch centre tensor: Cin [B][d][ca][cb] (stride cin_b0) = the centre tensor of site i; Cout [B][d][cb][cc] =
// (S V^H) A_{i+1}, the centre tensor of site i + 1.  The bond table is read, not written (the kept count goes to vchi_), and
// cert_min_ / cert_flag_ are updated.  Always the general kernels (the one-wavefront kernels work in place).
int Engine::svd_shift_right_virtual(StateSet& S, int i, const cplx* Cin, long cin_b0, cplx* Cout, long cout_b0, const int* ids, int nb0) {
  const int ca = cap[i], cb = cap[i + 1], cc = cap[i + 2];
  int rc;
  Region prof(*this, PROF_SVD);
  JacobiSource src;
  src.src = Cin; src.src_b0 = cin_b0; src.rx = d * ca; src.ncols = cb; src.conj = 0; src.tri = 0;
  src.r_n0 = ca; src.s_r1 = (long)ca * cb; src.s_r0 = cb; src.c_n0 = cb; src.s_c1 = 0; src.s_c0 = 1;
  src.nb0 = nb0; src.ids = ids;
  TruncSpec tr;
  tr.trunc_mode = 0; tr.threshold = 1e-12; tr.max_bond = 0; tr.min_keep = 1;
  tr.chiA = S.chi + i; tr.mulA = d; tr.chiB = S.chi + i + 1; tr.mulB = 1; tr.chiOut = vchi_ + i + 1; tr.chi_stride = L + 1;
  tr.spectrum = nullptr; tr.spec_ld = 0;
  JacobiShape sh;
  int sweeps = 0;
  if ((rc = jacobi_solve(src, tr, svdw, stream, &sh, &sweeps, false)) != TJM_OK) return rc;
  ++stat_svds; stat_svd_mats += nb0; stat_svd_sweeps += sweeps;
  hipLaunchKernelGGL(cert_update_kernel, dim3((nb0 + 255) / 256), dim3(256), 0, stream, svdw.norms, sh.ncols_pad, vchi_ + i + 1, S.chi + i + 1, L + 1,
                     cert_min_, cert_flag_, nb0, ids);
  ExtractDesc xu;  // U[(s,a)][k] = X_final / sigma  (zero beyond keep)
  xu.out = theta; xu.out_b0 = theta_b0; xu.n_k = cb; xu.o_k = 1; xu.n_r1 = 1; xu.n_r0 = d * ca; xu.o_r1 = 0; xu.o_r0 = cb;
  xu.row_off = 0; xu.conj = 0; xu.scale_mode = 2;
  if ((rc = svd_extract(xu, svdw, sh, vchi_ + i + 1, L + 1, nb0, ids, stream)) != TJM_OK) return rc;
  {
    GemmDesc g = blank_gemm();  // G[k][j] = sum_{(s,a)} conj(U[(s,a)][k]) C_i[(s,a)][j]
    g.A = theta; g.B = Cin; g.C = T2;
    g.M = cb; g.K = d * ca; g.N = cb;
    g.a_rs = 1; g.a_cs = cb; g.conjA = 1; g.b_rs = cb; g.b_cs = 1; g.c_rs = cb;
    g.nb0 = nb0; g.ids = ids; g.a_b0 = theta_b0; g.b_b0 = cin_b0; g.c_b0 = t_b0;
    if ((rc = gemm(g)) != TJM_OK) return rc;
  }
  GemmDesc g = blank_gemm();  // Cout[t][k][c] = G[k][j] A_{i+1}[t][j][c]
  g.A = T2; g.B = S.A[i + 1]; g.C = Cout;
  g.M = cb; g.K = cb; g.N = cc;
  g.a_rs = cb; g.a_cs = 1; g.b_rs = cc; g.b_cs = 1; g.c_rs = cc;
  g.nb0 = nb0; g.ids = ids; g.nb1 = d; g.a_b0 = t_b0; g.b_b0 = a_b0_[i + 1]; g.b_b1 = (long)cb * cc; g.c_b0 = cout_b0; g.c_b1 = (long)cb * cc;
  return gemm(g);
}

// General centre shift i -> i-1 by the two-site SVD (mps.py:771-788), any gauge.
int Engine::svd_shift_left_2site(StateSet& S, int i, const int* ids, int nb0) {
  int rc;
  if ((rc = merge_matrix_layout(S, i - 1, ids, nb0)) != TJM_OK) return rc;
  return split(S, i - 1, 1, 0, 1e-12, 0, 1, ids, nb0);
}

__global__ __launch_bounds__(256) void copy_back_kernel(cplx* __restrict__ dst, long dst_b0, const cplx* __restrict__ src, long src_b0, long n,
                                                       const int* ids) {
  int b = blockIdx.y;
  if (ids) b = ids[b];
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
    dst[(long)b * dst_b0 + e] = src[(long)b * src_b0 + e];
}

int Engine::copy_back(cplx* dst, long dst_b0, const cplx* src, long src_b0, long n, const int* ids, int nb0) {
  int gx = (int)((n + 1023) / 1024);
  if (gx < 1) gx = 1;
  if (gx > 128) gx = 128;
  hipLaunchKernelGGL(copy_back_kernel, dim3(gx, nb0), dim3(256), 0, stream, dst, dst_b0, src, src_b0, n, ids);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

// x *= s (uniform scalar over the batch)
__global__ void fill_kernel(real* p, real v, int n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) p[t] = v;
}
__global__ void rsqrt_kernel(const real* in, real* out, int n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) out[t] = (in[t] > 0.0) ? real(1.0 / sqrt(in[t])) : real(0);
}

int Engine::set_noise_filter(int n, const int* idx) {
  if (n < 0) { proc_on_.assign(noise_.size(), 1); return TJM_OK; }
  proc_on_.assign(noise_.size(), 0);
  for (int k = 0; k < n; ++k) {
    if (idx[k] < 0 || idx[k] >= (int)noise_.size()) return TJM_ERR_ARG;
    proc_on_[idx[k]] = 1;
  }
  return TJM_OK;
}

// QR centre shifts (mps.py:719-746 / 771-788 with decomposition="QR"): exact gauge moves, no truncation
int Engine::qr_shift_right(StateSet& S, int i, const int* ids, int nb0) {
  if (nb0 < 0) nb0 = B;
  const int cb = cap[i + 1], cc = cap[i + 2];
  int rc;
  if (svd_shift_small_fits(d, cap[i], cb, false)) return qr_site(S, i, true, ids, nb0, true);
  if ((rc = qr_site(S, i, true, ids, nb0)) != TJM_OK) return rc;
  GemmDesc g = blank_gemm();  // T1[p][l][r] = C[l][x] A_{i+1}[p][x][r]
  g.A = Cm_; g.B = S.A[i + 1]; g.C = T1;
  g.M = cb; g.K = cb; g.N = cc;
  g.a_rs = cb; g.a_cs = 1; g.b_rs = cc; g.b_cs = 1; g.c_rs = cc;
  g.nb0 = nb0; g.nb1 = d; g.a_b0 = (long)cb * cb; g.b_b0 = a_b0_[i + 1]; g.b_b1 = (long)cb * cc; g.c_b0 = t_b0; g.c_b1 = (long)cb * cc;
  g.ids = ids;
  if ((rc = gemm(g)) != TJM_OK) return rc;
  return copy_back(S.A[i + 1], a_b0_[i + 1], T1, t_b0, a_b0_[i + 1], ids, nb0);
}

int Engine::qr_shift_left(StateSet& S, int i) {
  const int cz = cap[i - 1], ca = cap[i];
  int rc;
  if (svd_shift_small_fits(d, ca, cap[i + 1], true)) return qr_site(S, i, false, nullptr, B, true);
  if ((rc = qr_site(S, i, false)) != TJM_OK) return rc;
  GemmDesc g = blank_gemm();  // T1[(p,l)][r] = A_{i-1}[(p,l)][x] C^T[x][r]
  g.A = S.A[i - 1]; g.B = Cm_; g.C = T1;
  g.M = d * cz; g.K = ca; g.N = ca;
  g.a_rs = ca; g.a_cs = 1; g.b_rs = ca; g.b_cs = 1; g.c_rs = ca;
  g.nb0 = B; g.a_b0 = a_b0_[i - 1]; g.b_b0 = (long)ca * ca; g.c_b0 = t_b0;
  if ((rc = gemm(g)) != TJM_OK) return rc;
  return copy_back(S.A[i - 1], a_b0_[i - 1], T1, t_b0, a_b0_[i - 1], nullptr, B);
}

// normalize("B", "QR") from a known centre (mps.py:815-839): QR shifts down to site 0, then drop R (unit norm)
int Engine::normalize_qr(int set, int center) {
  if (!bound_ || center < 0 || center >= L) return TJM_ERR_ARG;
  StateSet& S = sets[set];
  int rc;
  if (sweep_ok_ && center >= 1) {
    std::vector<SmallSweepStep> steps;
    for (int i = center; i >= 1; --i) {
      SmallSweepStep st{};
      st.site = i; st.kind = 4; st.op = 0;
      steps.push_back(st);
    }
    if ((rc = run_sweep(set, steps, nullptr, B)) != TJM_OK) return rc;
  } else {
    for (int i = center; i >= 1; --i)
      if ((rc = qr_shift_left(S, i)) != TJM_OK) return rc;
  }
  if ((rc = launch_normsq(S.A[0], a_b0_[0], a_b0_[0], normsq_, B, nullptr, stream)) != TJM_OK) return rc;
  hipLaunchKernelGGL(rsqrt_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, normsq_, scal_, B);
  return launch_scale(S.A[0], a_b0_[0], a_b0_[0], scal_, B, nullptr, nullptr, stream);
}

// _apply_single_qubit_gate (digital_tjm.py:304-309): a unitary on the physical leg keeps the gauge
int Engine::apply_single(int set, int site, const double* host_mat) {
  if (!bound_ || site < 0 || site >= L) return TJM_ERR_ARG;
  StateSet& S = sets[set];
  if (int rcu = upload_c(ops_ + (size_t)(L + 2) * d * d * d * d, host_mat, (size_t)d * d, stream)) return rcu;
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  return launch_apply_local(S.A[site], a_b0_[site], d, (long)cap[site] * cap[site + 1], ops_ + (size_t)(L + 2) * d * d * d * d, nullptr, B, nullptr, stream);
}

// QR shifts of the centre from site `from` to site `to` (either direction) on the whole batch: one launch at small bonds.
int Engine::qr_walk(int set, int from, int to) {
  StateSet& S = sets[set];
  int rc;
  if (from == to) return TJM_OK;
  if (sweep_ok_) {
    std::vector<SmallSweepStep> steps;
    if (from < to)
      for (int i = from; i < to; ++i) { SmallSweepStep st{}; st.site = i; st.kind = 3; steps.push_back(st); }
    else
      for (int i = from; i > to; --i) { SmallSweepStep st{}; st.site = i; st.kind = 4; steps.push_back(st); }
    return run_sweep(set, steps, nullptr, B);
  }
  if (from < to) {
    for (int i = from; i < to; ++i)
      if ((rc = qr_shift_right(S, i)) != TJM_OK) return rc;
  } else {
    for (int i = from; i > to; --i)
      if ((rc = qr_shift_left(S, i)) != TJM_OK) return rc;
  }
  return TJM_OK;
}

// apply_two_qubit_gate_tebd (digital_tjm.py:455-533) for a nearest-neighbour gate on (left, left+1), from a state with
// centre `center`: QR shifts put the centre on the pair, then merge, gate, truncated split to the right (min_keep = min(2, chi));
// the centre ends on left + 1.
int Engine::tebd_gate(int set, int left, const double* host_u, int center) {
  if (!bound_ || left < 0 || left + 1 >= L || center < 0 || center >= L) return TJM_ERR_ARG;
  StateSet& S = sets[set];
  int rc;
  // shift_center_to(left) unless the centre already sits on the pair (digital_tjm.py:503-506); QR shifts only move the gauge
  if (center < left || center > left + 1)
    if ((rc = qr_walk(set, center, left)) != TJM_OK) return rc;
  if (int rcu = upload_c(ops_ + (size_t)(L + 4) * d * d * d * d, host_u, (size_t)d * d * d * d, stream)) return rcu;
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  const int mk = (max_bond > 0) ? std::min(2, max_bond) : 2;
  return two_site_op(S, left, ops_ + (size_t)(L + 4) * d * d * d * d, nullptr, nullptr, B, mk);
}

// A d^2 x d^2 operator on the merged pair (left, left+1) followed by the truncated split to the right, in whatever gauge the
// state is in (scheduled two-site jumps, scheduled_jumps.py:88-106: sim_params truncation, min_keep 1).
int Engine::apply_pair(int set, int left, const double* host_u, int min_keep) {
  if (!bound_ || left < 0 || left + 1 >= L || !host_u || min_keep < 1) return TJM_ERR_ARG;
  StateSet& S = sets[set];
  if (int rcu = upload_c(ops_ + (size_t)(L + 4) * d * d * d * d, host_u, (size_t)d * d * d * d, stream)) return rcu;
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  return two_site_op(S, left, ops_ + (size_t)(L + 4) * d * d * d * d, nullptr, nullptr, B, min_keep);
}

// QR sweep from `center` down to site 0 without the final normalisation: with center = L-1 it right-canonicalises a state in
// any gauge (the sweep of normalize("B") / set_canonical_form, mps.py:790-839); ||A_0||^2 is then the squared norm of the state.
int Engine::canonicalize_qr(int set, int center) {
  if (!bound_ || center < 0 || center >= L) return TJM_ERR_ARG;
  return qr_walk(set, center, 0);
}

int Engine::dissipate(int set, double dt_, int start_center) {
  if (!bound_) return TJM_ERR_STATE;
  if (start_center < 0 || start_center >= L) return TJM_ERR_ARG;
  StateSet& S = sets[set];
  int rc;
  bool any = false;
  for (size_t k = 0; k < noise_.size(); ++k) any = any || (proc_on_[k] && noise_[k].gamma != 0.0);
  if (!any)  // dissipation.py:79-86: centre to 0 by QR, then QR at site 0 with R discarded = renormalise
    return normalize_qr(set, start_center);
  if (sweep_ok_) {  // small bonds: the whole sweep in one launch, unless an adjacent non-Pauli pair needs its merged two-site factor
    std::vector<SmallSweepStep> steps;
    bool fits = true;
    for (int i = start_center; i < L - 1; ++i) {
      SmallSweepStep st{};
      st.site = i; st.kind = 1; st.op = 0;
      steps.push_back(st);
    }
    for (int i = L - 1; i >= 0 && fits; --i) {
      double expo = 0.0;
      bool need_matrix = false, any_one = false;
      cplx gen[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
      for (int k : one_by_site_[i]) {
        if (!proc_on_[k]) continue;
        any_one = true;
        const NoiseProc& p = noise_[k];
        if (p.pauli) {
          gen[0].x += p.gamma; gen[3].x += p.gamma;
        } else {
          need_matrix = true;
          for (int a = 0; a < d; ++a) for (int c = 0; c < d; ++c) {
            cplx acc{0.0, 0.0};
            for (int r = 0; r < d; ++r) cfma(acc, cconj(p.mat[r * d + a]), p.mat[r * d + c]);
            gen[a * d + c] = cadd(gen[a * d + c], cscale(acc, p.gamma));
          }
        }
      }
      if (any_one && !need_matrix) expo += gen[0].x;
      if (i != 0)
        for (int k : two_by_right_[i]) {
          if (!proc_on_[k]) continue;
          const NoiseProc& p = noise_[k];
          if ((p.site1 - p.site0) > 1) {
            if (!p.pauli) return TJM_ERR_NOT_IMPLEMENTED;  // dissipation.py:136-138
            expo += p.gamma;
          } else if (p.pauli) {
            expo += p.gamma;  // all adjacent processes of this pair must be Pauli for the scalar form (checked below)
          } else {
            fits = false;  // merged two-site factor with a truncated split: general path
          }
        }
      if (!fits) break;
      if (need_matrix) {  // the matrix step first, then the scalar on the same site, then the shift (same order as below)
        SmallSweepStep sm{};
        sm.site = i; sm.kind = 0; sm.op = 1;
        cplx arg[4];
        for (int q = 0; q < 4; ++q) arg[q] = cscale(gen[q], -0.5 * dt_);
        small_expm(arg, d, sm.m);
        steps.push_back(sm);
      }
      SmallSweepStep st{};
      st.site = i; st.kind = (i != 0) ? 2 : 0;
      if (expo != 0.0) { st.op = 2; st.scal = std::exp(-0.5 * dt_ * expo); }
      if (st.kind != 0 || st.op != 0) steps.push_back(st);
    }
    if (fits) return run_sweep(set, steps, nullptr, B);
  }
  const int dd = d * d, slot = dd * dd;  // a one-site operator has dd entries, an operator on a merged pair dd x dd
  cert_set_ = -1;
  {  // ---- certified scalar dissipation (see svd_shift_right_virtual)
    static const bool cert_off = getenv("TJM_NO_CERT_DISSIPATION") != nullptr;
    bool scalar_only = !cert_off && start_center == 0 && L >= 2;
    std::vector<double> expo_site(L, 0.0);
    double expo_total = 0.0;
    for (int i = L - 1; i >= 0 && scalar_only; --i) {
      for (int k : one_by_site_[i])
        if (proc_on_[k]) { if (!noise_[k].pauli) scalar_only = false; else expo_site[i] += noise_[k].gamma; }
      if (i != 0)
        for (int k : two_by_right_[i])
          if (proc_on_[k]) { if (!noise_[k].pauli) scalar_only = false; else expo_site[i] += noise_[k].gamma; }
      expo_total += expo_site[i];
    }
    // Which trajectories try the certificate is a function of each trajectory's own history (after a failure a trajectory sits out
    // one call, after the next failure two, then four at most; a success clears it): the code path of a trajectory - and with it
    // the last bits of its result - must not depend on who else shares its engine.
    std::vector<int> trying;
    if (scalar_only) {
      cert_size();
      int* wait = &cert_wait_[(size_t)set * B];  // the back-off belongs to the state set: the sampling copy of the order-2 driver has
      // (the back-off dates from the SVD pass, which cost a third of a sweep: with the Gram pass every trajectory tries every time;
      // TJM_CERT_BACKOFF=1 keeps it)
      static const bool keep_backoff = getenv("TJM_CERT_BACKOFF") != nullptr;
      const bool backoff = keep_backoff || !cert_gram_fits();
      for (int b = 0; b < B; ++b) {              // its own, so the main trajectory's path does not depend on sample_timesteps
        if (backoff && wait[b] > 0) --wait[b];
        else trying.push_back(b);
      }
      if (trying.empty()) scalar_only = false;
    }
    if (scalar_only) {
      const int nt = (int)trying.size();
      int* try_ids = nullptr;  // device list of the trajectories that try (null: all of them)
      if (nt < B) {
        try_ids = cert_flag_ + B;  // second half of the flag buffer (carved as B doubles)
        TJM_HIP_CHECK(hipMemcpyAsync(try_ids, trying.data(), (size_t)nt * sizeof(int), hipMemcpyHostToDevice, stream));
      }
      hipLaunchKernelGGL(cert_init_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, cert_min_, cert_flag_, B);
      const double scale = std::exp(-0.5 * dt_ * expo_total);
      if (cert_gram_fits()) {
        // no SVD: Gram matrices of the centre tensors by the density-environment recursion, one positive-definiteness test per bond
        // against the cut the comparison below applies to the smallest singular value (cert_min_ stays at its initial 3e38)
        Region prof(*this, PROF_SVD);
        if ((rc = cert_pass_gram(S, try_ids, nt, 1e-12 * (1.0 + 1e-6) / (scale * scale) + 2e-14)) != TJM_OK) return rc;
      } else {
        const cplx* cin = S.A[0];
        long cin_b0 = a_b0_[0];
        for (int i = 0; i < L - 1; ++i) {  // the right-going pass on scratch centre tensors (two slots of the idle Krylov basis buffer)
          cplx* cout = V + (long)(i & 1) * v_ld;
          if ((rc = svd_shift_right_virtual(S, i, cin, cin_b0, cout, v_b0, try_ids, nt)) != TJM_OK) return rc;
          cin = cout;
          cin_b0 = v_b0;
        }
      }
      std::vector<real> mins(B);
      std::vector<int> flags(B);
      TJM_HIP_CHECK(hipMemcpyAsync(mins.data(), cert_min_, (size_t)B * sizeof(real), hipMemcpyDeviceToHost, stream));
      TJM_HIP_CHECK(hipMemcpyAsync(flags.data(), cert_flag_, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, stream));
      TJM_HIP_CHECK(hipStreamSynchronize(stream));
      std::vector<int> good, rest;
      std::vector<char> tried(B, 0);
      for (int b : trying) tried[b] = 1;
      for (int b = 0; b < B; ++b) {
        // the left-going pass of the reference sees the singular values scaled by the factors already applied (at most `scale`
        // in all): no truncation anywhere if even the smallest value, fully scaled, clears the threshold (margin: rounding)
        const bool ok = tried[b] && flags[b] == 0 && (double)mins[b] * scale * scale >= 1e-12 * (1.0 + 1e-6);
        if (tried[b]) {  // truncating bonds: this trajectory takes the plain sweep for its next call(s)
          int& back = cert_back_[(size_t)set * B + b];
          back = ok ? 0 : std::min(4, std::max(1, 2 * back));
          cert_wait_[(size_t)set * B + b] = back;
        }
        (ok ? good : rest).push_back(b);
      }
      if (getenv("TJM_DEBUG_CERT")) {
        int nf = 0; double mn = 1e300;
        for (int b = 0; b < B; ++b) { nf += flags[b]; mn = std::min(mn, (double)mins[b]); }
        fprintf(stderr, "[cert] B %d flagged %d min sigma^2 %.3e scale^2 %.3e certified %zu\n", B, nf, mn, scale * scale, good.size());
      }
      if (!good.empty()) {
        hipLaunchKernelGGL(fill_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, scal_, (real)scale, B);
        TJM_HIP_CHECK(hipMemcpyAsync(ids_, good.data(), good.size() * sizeof(int), hipMemcpyHostToDevice, stream));
        if ((rc = launch_scale(S.A[0], a_b0_[0], a_b0_[0], scal_, (int)good.size(), ids_, nullptr, stream)) != TJM_OK) return rc;
        TJM_HIP_CHECK(hipStreamSynchronize(stream));
      }
      if (!rest.empty()) {  // the reference's sweep for the others
        TJM_HIP_CHECK(hipMemcpyAsync(ids_, rest.data(), rest.size() * sizeof(int), hipMemcpyHostToDevice, stream));
        const int nr = (int)rest.size();
        for (int i = 0; i < L - 1; ++i)
          if ((rc = svd_shift_right(S, i, ids_, nr)) != TJM_OK) return rc;
        for (int i = L - 1; i >= 0; --i) {
          if (expo_site[i] != 0.0) {
            hipLaunchKernelGGL(fill_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, scal_, (real)std::exp(-0.5 * dt_ * expo_site[i]), B);
            if ((rc = launch_scale(S.A[i], a_b0_[i], a_b0_[i], scal_, nr, ids_, nullptr, stream)) != TJM_OK) return rc;
          }
          if (i != 0 && (rc = svd_shift_left(S, i, ids_, nr)) != TJM_OK) return rc;
        }
        TJM_HIP_CHECK(hipStreamSynchronize(stream));
      }
      stat_cert_traj += (long)good.size();
      std::fill(cert_ok_.begin(), cert_ok_.end(), 0);
      if (!good.empty()) {
        if ((rc = state_checksum(set, nullptr, B, cert_sum_.data())) != TJM_OK) return rc;
        for (int b : good) cert_ok_[b] = 1;
        cert_set_ = set;
      }
      return TJM_OK;
    }
  }
  for (int i = start_center; i < L - 1; ++i)
    if ((rc = svd_shift_right(S, i, nullptr, B)) != TJM_OK) return rc;
  for (int i = L - 1; i >= 0; --i) {
    double expo = 0.0;
    bool need_matrix = false;
    cplx gen[MAXDD];
    for (int q = 0; q < dd; ++q) gen[q] = cplx{0.0, 0.0};
    bool any_one = false;
    for (int k : one_by_site_[i]) {
      if (!proc_on_[k]) continue;
      any_one = true;
      const NoiseProc& p = noise_[k];
      if (p.pauli) {
        for (int q = 0; q < d; ++q) gen[q * d + q].x += p.gamma;
      } else {
        need_matrix = true;
        for (int a = 0; a < d; ++a) for (int c = 0; c < d; ++c) {
          cplx acc{0.0, 0.0};
          for (int r = 0; r < d; ++r) cfma(acc, cconj(p.mat[r * d + a]), p.mat[r * d + c]);
          gen[a * d + c] = cadd(gen[a * d + c], cscale(acc, p.gamma));
        }
      }
    }
    if (any_one && !need_matrix) expo += gen[0].x;
    bool need_matrix2 = false;
    cplx gen2[MSLOT];
    for (int q = 0; q < slot; ++q) gen2[q] = cplx{0.0, 0.0};
    if (i != 0) {
      double adj_pauli = 0.0;
      for (int k : two_by_right_[i]) {
        if (!proc_on_[k]) continue;
        const NoiseProc& p = noise_[k];
        const bool longrange = (p.site1 - p.site0) > 1;
        if (longrange) {
          if (!p.pauli) return TJM_ERR_NOT_IMPLEMENTED;  // dissipation.py:136-138
          expo += p.gamma;
        } else if (p.pauli) {
          adj_pauli += p.gamma;
          for (int q = 0; q < dd; ++q) gen2[q * dd + q].x += p.gamma;
        } else {
          need_matrix2 = true;
          for (int a = 0; a < dd; ++a) for (int c = 0; c < dd; ++c) {
            cplx acc{0.0, 0.0};
            for (int r = 0; r < dd; ++r) cfma(acc, cconj(p.mat[r * dd + a]), p.mat[r * dd + c]);
            gen2[a * dd + c] = cadd(gen2[a * dd + c], cscale(acc, p.gamma));
          }
        }
      }
      if (!need_matrix2) expo += adj_pauli;  // all adjacent processes Pauli: scalar (dissipation.py:156-157)
    }
    if (need_matrix) {
      cplx arg[MAXDD], m[MAXDD];
      for (int q = 0; q < dd; ++q) arg[q] = cscale(gen[q], -0.5 * dt_);
      small_expm(arg, d, m);
      TJM_HIP_CHECK(hipMemcpyAsync(ops_ + (size_t)i * slot, m, (size_t)dd * sizeof(cplx), hipMemcpyHostToDevice, stream));
      TJM_HIP_CHECK(hipStreamSynchronize(stream));
      if ((rc = launch_apply_local(S.A[i], a_b0_[i], d, (long)cap[i] * cap[i + 1], ops_ + (size_t)i * slot, nullptr, B, nullptr, stream)) != TJM_OK) return rc;
    }
    if (expo != 0.0) {
      hipLaunchKernelGGL(fill_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, scal_, std::exp(-0.5 * dt_ * expo), B);
      if ((rc = launch_scale(S.A[i], a_b0_[i], a_b0_[i], scal_, B, nullptr, nullptr, stream)) != TJM_OK) return rc;
    }
    if (need_matrix2) {  // merged pair (i-1, i): expm(-dt/2 sum gamma L^dag L), truncated split to the right
      cplx arg[MSLOT], m2[MSLOT];
      for (int q = 0; q < slot; ++q) arg[q] = cscale(gen2[q], -0.5 * dt_);
      small_expm(arg, dd, m2);
      TJM_HIP_CHECK(hipMemcpyAsync(ops_ + (size_t)L * slot, m2, (size_t)slot * sizeof(cplx), hipMemcpyHostToDevice, stream));
      TJM_HIP_CHECK(hipStreamSynchronize(stream));
      if ((rc = two_site_op(S, i - 1, ops_ + (size_t)L * slot, nullptr, nullptr, B, 1)) != TJM_OK) return rc;
    }
    if (i != 0)
      if ((rc = svd_shift_left(S, i, nullptr, B)) != TJM_OK) return rc;
  }
  return TJM_OK;
}

int Engine::site_normsq0(int set, double* host_out) {
  StateSet& S = sets[set];
  int rc;
  if ((rc = launch_normsq(S.A[0], a_b0_[0], a_b0_[0], normsq_, B, nullptr, stream)) != TJM_OK) return rc;
  std::vector<real> h(B);
  TJM_HIP_CHECK(hipMemcpyAsync(h.data(), normsq_, B * sizeof(real), hipMemcpyDeviceToHost, stream));
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  for (int b = 0; b < B; ++b) host_out[b] = h[b];
  return TJM_OK;
}

// ---- projective sampling of all sites (MPS.measure_single_shot / measure_shots, mps.py:1282-1417) -----------------------
// The reference walks the centre through the chain and projects site by site; from a state with centre 0 the projected
// left part is a row vector, so S shots of one trajectory are the rows of a matrix: per site two GEMMs
// T[sigma] = Vec (S x ca) A_i[sigma] (ca x cb), then per shot p(sigma') = || sum_sigma rot[sigma'][sigma] T[sigma][s] ||^2,
// rng.choice (searchsorted of the cumulative sum, one uniform) and Vec'[s] = rotated row / sqrt(p).
__global__ __launch_bounds__(256) void shot_select_kernel(const cplx* __restrict__ T, long t_b0, long t_sig, int cb, int ldt, cplx rot00, cplx rot01,
                                                         cplx rot10, cplx rot11, const double* __restrict__ u, int u_b0, int u_s, int site,
                                                         cplx* __restrict__ vec, long v_b0, int ldv, unsigned char* __restrict__ bits, long bits_b0,
                                                         int L, int shots) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int sidx = blockIdx.x * 4 + (threadIdx.x >> 6);  // one wavefront per shot
  if (sidx >= shots) return;
  const cplx* t0 = T + (long)b * t_b0 + (long)sidx * ldt;
  const cplx* t1 = t0 + t_sig;
  double p0 = 0.0, p1 = 0.0;
  for (int c = lane; c < cb; c += 64) {
    const cplx a = t0[c], q = t1[c];
    const cplx r0 = cadd(cmul(rot00, a), cmul(rot01, q));
    const cplx r1 = cadd(cmul(rot10, a), cmul(rot11, q));
    p0 = fma(r0.x, r0.x, fma(r0.y, r0.y, p0));
    p1 = fma(r1.x, r1.x, fma(r1.y, r1.y, p1));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    p0 += __shfl_xor(p0, o, 64);
    p1 += __shfl_xor(p1, o, 64);
  }
  // probabilities / sum, cdf = cumsum / cdf[-1], searchsorted(cdf, u, side="right")
  const double tot = p0 + p1;
  const double q0 = p0 / tot, q1 = p1 / tot;
  const double last = q0 + q1;
  const double uu = u[(long)b * u_b0 + (long)sidx * u_s + site];
  const int pick = (uu < q0 / last) ? 0 : 1;
  const double inv = 1.0 / sqrt(pick ? q1 * tot : q0 * tot);
  cplx* vo = vec + (long)b * v_b0 + (long)sidx * ldv;
  for (int c = lane; c < cb; c += 64) {
    const cplx a = t0[c], q = t1[c];
    const cplx r = pick ? cadd(cmul(rot10, a), cmul(rot11, q)) : cadd(cmul(rot00, a), cmul(rot01, q));
    vo[c] = cplx{real(r.x * inv), real(r.y * inv)};
  }
  if (lane == 0) bits[(long)b * bits_b0 + (long)sidx * L + site] = (unsigned char)pick;
}

__global__ void shot_init_kernel(cplx* vec, long v_b0, int ldv, int shots) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < shots) vec[(long)blockIdx.y * v_b0 + (long)s * ldv] = cplx{1.0, 0.0};
}

int Engine::sample_shots(int set, int shots, const double* host_rot, const double* host_u, unsigned char* host_bits) {
  if (!bound_) return TJM_ERR_STATE;
  if (shots <= 0 || !host_rot || !host_u || !host_bits || d != 2) return TJM_ERR_ARG;
  StateSet& S = sets[set];
  const int cm = *std::max_element(cap.begin(), cap.end());
  // buffers inside the Krylov basis area: Vec [S][cm] and T [d][S][cm] per trajectory, uniforms and bits behind them
  const long per_shot = (long)(1 + d) * cm + (L * (long)sizeof(double) + L + (long)sizeof(cplx) - 1) / (long)sizeof(cplx) + 1;
  const long s_max = v_b0 / per_shot;
  if (s_max < 1) return TJM_ERR_WORKSPACE;
  cplx rot[4];
  from_host_c(rot, host_rot, 4);
  int rc;
  for (int s0 = 0; s0 < shots; s0 += (int)s_max) {
    const int ns = (int)std::min<long>(s_max, shots - s0);
    cplx* vec = V;                                  // [B] stride v_b0: Vec at 0
    cplx* T = V + (long)ns * cm;                    // T[sigma][s][cm]
    double* du = reinterpret_cast<double*>(V + (long)(1 + d) * ns * cm);
    unsigned char* dbits = reinterpret_cast<unsigned char*>(du + (long)ns * L);
    const long vb = v_b0;                           // trajectory stride in complex elements
    for (int b = 0; b < B; ++b)
      TJM_HIP_CHECK(hipMemcpyAsync(reinterpret_cast<char*>(du) + (size_t)b * vb * sizeof(cplx), host_u + ((size_t)b * shots + s0) * L,
                                   (size_t)ns * L * sizeof(double), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(shot_init_kernel, dim3((ns + 255) / 256, B), dim3(256), 0, stream, vec, vb, cm, ns);
    for (int i = 0; i < L; ++i) {
      const int ca = cap[i], cb = cap[i + 1];
      GemmDesc g = blank_gemm();  // T[sigma][s][c] = sum_a Vec[s][a] A_i[sigma][a][c]
      g.A = vec; g.B = S.A[i]; g.C = T;
      g.M = ns; g.K = ca; g.N = cb;
      g.a_rs = cm; g.a_cs = 1; g.b_rs = cb; g.b_cs = 1; g.c_rs = cm;
      g.nb0 = B; g.nb1 = d;
      g.a_b0 = vb; g.b_b0 = a_b0_[i]; g.b_b1 = (long)ca * cb; g.c_b0 = vb; g.c_b1 = (long)ns * cm;
      if ((rc = gemm(g)) != TJM_OK) return rc;
      hipLaunchKernelGGL(shot_select_kernel, dim3((ns + 3) / 4, B), dim3(256), 0, stream, T, vb, (long)ns * cm, cb, cm, rot[0], rot[1], rot[2], rot[3],
                         du, (int)(vb * 2), L, i, vec, vb, cm, dbits, vb * (long)sizeof(cplx), L, ns);
    }
    TJM_HIP_CHECK(hipGetLastError());
    for (int b = 0; b < B; ++b)
      TJM_HIP_CHECK(hipMemcpyAsync(host_bits + ((size_t)b * shots + s0) * L, dbits + (size_t)b * vb * sizeof(cplx), (size_t)ns * L,
                                   hipMemcpyDeviceToHost, stream));
    TJM_HIP_CHECK(hipStreamSynchronize(stream));
  }
  return TJM_OK;
}

// Singular values of theta = A_i A_{i+1} as a (d cap_i) x (d cap_{i+2}) matrix, descending: the quantity behind
// MPS.get_entropy / get_schmidt_spectrum (mps.py:604-678), which the reference evaluates on the tensors as they stand.
int Engine::bond_spectrum(int set, int i, double* host_spec, int n_out) {
  if (!bound_) return TJM_ERR_STATE;
  if (i < 0 || i + 1 >= L || !host_spec || n_out < 1) return TJM_ERR_ARG;
  StateSet& S = sets[set];
  int rc;
  if ((rc = merge_matrix_layout(S, i, nullptr, B)) != TJM_OK) return rc;
  const int m = d * cap[i], n = d * cap[i + 2];
  const int nsv = std::min(m, n);
  real* dspec = reinterpret_cast<real*>(T2);
  int* dchi = reinterpret_cast<int*>(reinterpret_cast<char*>(T2) + align_up((size_t)B * nsv * sizeof(double)));
  TJM_HIP_CHECK(hipMemcpyAsync(dchi, S.chi, (size_t)B * (L + 1) * sizeof(int), hipMemcpyDeviceToDevice, stream));
  SvdSplitDesc sd;
  sd.theta = theta; sd.theta_b0 = theta_b0; sd.ld_theta = n;
  sd.m = m; sd.n = n; sd.d = d;
  sd.capL = cap[i]; sd.capR = cap[i + 2]; sd.capM = 1;
  sd.left = T1; sd.right = T1 + t_b0 / 2; sd.left_b0 = t_b0; sd.right_b0 = t_b0;
  sd.distribution = 0; sd.trunc_mode = 2; sd.threshold = 0.0; sd.max_bond = 0; sd.min_keep = 1;
  sd.chiL = dchi + i; sd.chiR = dchi + i + 2; sd.chiM = dchi + i + 1; sd.chi_stride = L + 1;
  sd.spectrum = dspec; sd.spec_ld = nsv; sd.nb0 = B; sd.ids = nullptr;
  int sweeps = 0;
  if ((rc = (std::max(m, n) > 512 ? svd_split_qr(sd, svdw, qrw, stream, &sweeps) : svd_split(sd, svdw, stream, &sweeps))) != TJM_OK) return rc;
  std::vector<real> h((size_t)B * nsv);
  TJM_HIP_CHECK(hipMemcpyAsync(h.data(), dspec, h.size() * sizeof(real), hipMemcpyDeviceToHost, stream));
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  for (int b = 0; b < B; ++b)
    for (int k = 0; k < n_out; ++k) host_spec[(size_t)b * n_out + k] = (k < nsv) ? h[(size_t)b * nsv + k] : 0.0;
  return TJM_OK;
}

// MPS.project_onto_bitstring (mps.py:1495-1537): |<bits|psi>|^2 as the squared norm of the product of the selected slices
int Engine::bitstring_probability(int set, const unsigned char* bits, double* host_prob) {
  if (!bound_) return TJM_ERR_STATE;
  if (!bits || !host_prob) return TJM_ERR_ARG;
  for (int i = 0; i < L; ++i) if (bits[i] >= d) return TJM_ERR_ARG;
  StateSet& S = sets[set];
  int rc;
  std::vector<cplx> ones((size_t)B, cplx{1.0, 0.0});
  TJM_HIP_CHECK(hipMemcpyAsync(E_, ones.data(), ones.size() * sizeof(cplx), hipMemcpyHostToDevice, stream));
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  cplx* v = E_;
  cplx* vn = E2_;
  long vs = 1;  // trajectory stride of the current row vector
  for (int i = 0; i < L; ++i) {
    const int ca = cap[i], cb = cap[i + 1];
    GemmDesc g = blank_gemm();  // vn[c] = sum_a v[a] A_i[bits_i][a][c]
    g.A = v; g.B = S.A[i] + (long)bits[i] * ca * cb; g.C = vn;
    g.M = 1; g.K = ca; g.N = cb;
    g.a_rs = ca; g.a_cs = 1; g.b_rs = cb; g.b_cs = 1; g.c_rs = cb;
    g.nb0 = B; g.a_b0 = vs; g.b_b0 = a_b0_[i]; g.c_b0 = cb;
    if ((rc = gemm(g)) != TJM_OK) return rc;
    std::swap(v, vn);
    vs = cb;
  }
  std::vector<cplx> h(B);
  TJM_HIP_CHECK(hipMemcpyAsync(h.data(), v, (size_t)B * sizeof(cplx), hipMemcpyDeviceToHost, stream));
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  for (int b = 0; b < B; ++b) host_prob[b] = h[b].x * h[b].x + h[b].y * h[b].y;
  return TJM_OK;
}

int Engine::site_moments(int set, double* host_M, double* host_M2) {
  if (!bound_) return TJM_ERR_STATE;
  StateSet& S = sets[set];
  int rc;
  cplx one{1.0, 0.0};
  // E_0 = [[1]] for every trajectory
  std::vector<cplx> ones((size_t)B, one);
  TJM_HIP_CHECK(hipMemcpyAsync(E_, ones.data(), ones.size() * sizeof(cplx), hipMemcpyHostToDevice, stream));
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  cplx* E = E_;
  cplx* En = E2_;
  for (int i = 0; i < L; ++i) {
    const int ca = cap[i], cb = cap[i + 1];
    {  // T[p][a][b] = sum_a' E[a][a'] A_i[p][a'][b]
      GemmDesc g = blank_gemm();
      g.A = E; g.B = S.A[i]; g.C = T1;
      g.M = ca; g.K = ca; g.N = cb;
      g.a_rs = ca; g.a_cs = 1; g.b_rs = cb; g.b_cs = 1; g.c_rs = cb;
      g.nb0 = B; g.nb1 = d;
      g.a_b0 = (long)ca * ca; g.b_b0 = a_b0_[i]; g.b_b1 = (long)ca * cb; g.c_b0 = t_b0; g.c_b1 = (long)ca * cb;
      if ((rc = gemm(g)) != TJM_OK) return rc;
    }
    if ((rc = launch_phys_overlap(S.A[i], T1, a_b0_[i], t_b0, d, (long)ca * cb, M_ + (size_t)i * B * d * d, B, nullptr, stream)) != TJM_OK) return rc;
    if (host_M2 && i + 1 < L) {
      // two-site moments M2[(s,t),(s',t')] = <theta_st | E | theta_s't'>: theta = A_i A_{i+1} into V[0], E theta = T A_{i+1} into V[1]
      const int cc = cap[i + 2];
      if ((rc = merge_tensor_layout(S, i, V, v_b0, nullptr, B)) != TJM_OK) return rc;
      GemmDesc g = blank_gemm();
      g.A = T1; g.B = S.A[i + 1]; g.C = V + v_ld;
      g.M = ca; g.K = cb; g.N = cc;
      g.a_rs = cb; g.a_cs = 1; g.b_rs = cc; g.b_cs = 1; g.c_rs = cc;
      g.nb0 = B; g.nb1 = d; g.nb2 = d;
      g.a_b0 = t_b0; g.a_b1 = (long)ca * cb; g.b_b0 = a_b0_[i + 1]; g.b_b2 = (long)cb * cc;
      g.c_b0 = v_b0; g.c_b1 = (long)d * ca * cc; g.c_b2 = (long)ca * cc;
      if ((rc = gemm(g)) != TJM_OK) return rc;
      if ((rc = launch_phys_overlap(V, V + v_ld, v_b0, v_b0, d * d, (long)ca * cc, M2_ + (size_t)i * B * d * d * d * d, B, nullptr, stream)) != TJM_OK)
        return rc;
    }
    if (i + 1 < L) {  // E'[b][b'] = sum_{(p,a)} conj(A_i[(p,a),b]) T[(p,a),b']
      GemmDesc g = blank_gemm();
      g.A = S.A[i]; g.B = T1; g.C = En;
      g.M = cb; g.K = d * ca; g.N = cb;
      g.a_rs = 1; g.a_cs = cb; g.b_rs = cb; g.b_cs = 1; g.c_rs = cb; g.conjA = 1;
      g.nb0 = B; g.a_b0 = a_b0_[i]; g.b_b0 = t_b0; g.c_b0 = (long)cb * cb;
      if ((rc = gemm(g)) != TJM_OK) return rc;
      std::swap(E, En);
    }
  }
  std::vector<cplx> hm((size_t)L * B * d * d), hm2(host_M2 ? (size_t)(L - 1) * B * d * d * d * d : 0);
  TJM_HIP_CHECK(hipMemcpyAsync(hm.data(), M_, hm.size() * sizeof(cplx), hipMemcpyDeviceToHost, stream));
  if (host_M2)
    TJM_HIP_CHECK(hipMemcpyAsync(hm2.data(), M2_, hm2.size() * sizeof(cplx), hipMemcpyDeviceToHost, stream));
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  to_host_c(host_M, hm.data(), hm.size());
  if (host_M2) to_host_c(host_M2, hm2.data(), hm2.size());
  return TJM_OK;
}

// theta[(s,a),(t,c)] <- sum_{s',t'} O_b[(s,t),(s',t')] theta[(s',a),(t',c)]   (two-site operator on the merged pair)
__global__ __launch_bounds__(256) void apply_phys2_kernel(cplx* __restrict__ theta, long th_b0, int d, int ca, int cc, const cplx* ops,
                                                         const int* op_index, const int* ids) {
  int b = blockIdx.y;
  if (ids) b = ids[b];
  const int oi = op_index ? op_index[b] : 0;
  if (oi < 0) return;
  const int P = d * d;
  const cplx* O = ops + (long)oi * P * P;
  cplx* tb = theta + (long)b * th_b0;
  const long n = (long)d * cc;
  const long total = (long)ca * cc;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int a = (int)(e / cc), c = (int)(e % cc);
    cplx v[16], y[16];
    for (int sp = 0; sp < d; ++sp)
      for (int tp = 0; tp < d; ++tp) v[sp * d + tp] = tb[((long)sp * ca + a) * n + (long)tp * cc + c];
    for (int q = 0; q < P; ++q) {
      cplx acc{0.0, 0.0};
      for (int r = 0; r < P; ++r) cfma(acc, O[q * P + r], v[r]);
      y[q] = acc;
    }
    for (int sp = 0; sp < d; ++sp)
      for (int tp = 0; tp < d; ++tp) tb[((long)sp * ca + a) * n + (long)tp * cc + c] = y[sp * d + tp];
  }
}

// merge (i, i+1), apply a d^2 x d^2 operator, split "right" with the run's truncation (dissipation.py:158-171,
// stochastic_process.py:268-288): centre ends on site i+1.
int Engine::two_site_op(StateSet& S, int i, const cplx* dev_ops, const int* op_index, const int* ids, int nb0, int min_keep) {
  int rc;
  if ((rc = merge_matrix_layout(S, i, ids, nb0)) != TJM_OK) return rc;
  const long total = (long)cap[i] * cap[i + 2];
  int gx = (int)((total + 255) / 256);
  if (gx > 128) gx = 128;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(apply_phys2_kernel, dim3(gx, nb0), dim3(256), 0, stream, theta, theta_b0, d, cap[i], cap[i + 2], dev_ops, op_index, ids);
  TJM_HIP_CHECK(hipGetLastError());
  return split(S, i, 0, trunc_mode, svd_threshold, max_bond, min_keep, ids, nb0);
}

// x_b <- O_b x_b at a per-trajectory site
__global__ __launch_bounds__(256) void apply_local_multi_kernel(cplx* const* site_ptr, const long* site_b0, const long* site_rest,
                                                               int d, const cplx* ops, const int* op_index, const int* site,
                                                               const int* ids) {
  const int b = ids[blockIdx.y];
  const int s = site[b];
  const int oi = op_index[b];
  if (oi < 0 || s < 0) return;
  const cplx* O = ops + (long)oi * d * d;
  cplx* xb = site_ptr[s] + (long)b * site_b0[s];
  const long rest = site_rest[s];
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < rest; r += (long)gridDim.x * blockDim.x) {
    cplx v[4], y[4];
    for (int q = 0; q < d; ++q) v[q] = xb[(long)q * rest + r];
    for (int p = 0; p < d; ++p) {
      cplx acc{0.0, 0.0};
      for (int q = 0; q < d; ++q) cfma(acc, O[p * d + q], v[q]);
      y[p] = acc;
    }
    for (int p = 0; p < d; ++p) xb[(long)p * rest + r] = y[p];
  }
}

// Jump weights dt * gamma_m * ||L_m psi||^2 of create_probability_distribution (stochastic_process.py:139-176) for the listed
// trajectories, in the reference's order: site sweep, one-site processes before the two-site ones starting at that site.
// nsq[B] = squared norms (||A_0||^2, centre 0).  w is [which.size()][order.size()].
int Engine::jump_weights(int set, double dt_, const std::vector<double>& nsq, const std::vector<int>& which, std::vector<int>& order,
                         std::vector<double>& w) {
  int rc;
  order.clear();
  for (int site = 0; site < L; ++site) {
    for (size_t k = 0; k < noise_.size(); ++k)
      if (proc_on_[k] && noise_[k].nsites == 1 && noise_[k].site0 == site) order.push_back((int)k);
    if (site < L - 1)
      for (size_t k = 0; k < noise_.size(); ++k)
        if (proc_on_[k] && noise_[k].nsites == 2 && noise_[k].site0 == site) order.push_back((int)k);
  }
  bool need_moments = false, need_moments2 = false;
  for (int k : order) {
    const NoiseProc& p = noise_[k];
    if (p.nsites == 1 && !p.pauli) need_moments = true;
    if (p.nsites == 2 && !p.pauli) {
      if (p.site1 != p.site0 + 1) return TJM_ERR_NOT_IMPLEMENTED;  // stochastic_process.py:171-176
      need_moments2 = true;
    }
  }
  std::vector<zc> Mh, Mh2;  // complex128, as site_moments hands them to the host
  if (need_moments || need_moments2) {
    Mh.resize((size_t)L * B * d * d);
    if (need_moments2) Mh2.resize((size_t)(L - 1) * B * d * d * d * d);
    if ((rc = site_moments(set, reinterpret_cast<double*>(Mh.data()), need_moments2 ? reinterpret_cast<double*>(Mh2.data()) : nullptr)) != TJM_OK)
      return rc;
  }
  w.assign(which.size() * order.size(), 0.0);
  for (size_t jb = 0; jb < which.size(); ++jb) {
    const int b = which[jb];
    for (size_t c = 0; c < order.size(); ++c) {
      const NoiseProc& p = noise_[order[c]];
      double nrm;
      if (p.pauli) {
        nrm = nsq[b];  // unitary jump operator: ||L psi||^2 = ||psi||^2
      } else if (p.nsites == 2) {
        // adjacent non-Pauli: Frobenius weight of the untruncated L theta (stochastic_process.py:53-83)
        const int dd = d * d;
        const zc* M2 = &Mh2[((size_t)p.site0 * B + b) * dd * dd];
        double acc = 0.0;
        for (int a = 0; a < dd; ++a) for (int c2 = 0; c2 < dd; ++c2) {
          cplx ll{0.0, 0.0};
          for (int r = 0; r < dd; ++r) cfma(ll, cconj(p.mat[r * dd + a]), p.mat[r * dd + c2]);
          acc += ll.x * M2[a * dd + c2].x - ll.y * M2[a * dd + c2].y;
        }
        nrm = acc;
      } else {
        // ||L psi||^2 = sum_{p,q} (L^dag L)[p][q] M[p][q]
        const zc* M = &Mh[((size_t)p.site0 * B + b) * d * d];
        double acc = 0.0;
        for (int a = 0; a < d; ++a) for (int c2 = 0; c2 < d; ++c2) {
          cplx ll{0.0, 0.0};
          for (int r = 0; r < d; ++r) cfma(ll, cconj(p.mat[r * d + a]), p.mat[r * d + c2]);
          acc += ll.x * M[a * d + c2].x - ll.y * M[a * d + c2].y;
        }
        nrm = acc;
      }
      w[jb * order.size() + c] = dt_ * p.gamma * nrm;
    }
  }
  return TJM_OK;
}

int Engine::stochastic(int set, double dt_, int* host_jumped, double* host_dp) {
  if (!bound_) return TJM_ERR_STATE;
  StateSet& S = sets[set];
  int rc;
  std::vector<double> nsq(B);
  if ((rc = site_normsq0(set, nsq.data())) != TJM_OK) return rc;
  const bool have_noise = !noise_.empty();
  std::vector<int> jumped;
  std::vector<double> scale(B, 1.0);
  std::vector<double> u_choice(B, 0.0);
  for (int b = 0; b < B; ++b) {
    const double dp = 1.0 - nsq[b];
    if (host_dp) host_dp[b] = dp;
    bool jump = false;
    if (have_noise) {
      if (cursor_[b] >= n_uniform_) return TJM_ERR_STATE;
      const double u = uni_host_[(size_t)b * n_uniform_ + cursor_[b]++];
      jump = !(u >= dp);  // stochastic_process.py:226
    }
    if (jump) {
      if (cursor_[b] >= n_uniform_) return TJM_ERR_STATE;
      u_choice[b] = uni_host_[(size_t)b * n_uniform_ + cursor_[b]++];
      jumped.push_back(b);
    } else {
      scale[b] = (nsq[b] > 0.0) ? 1.0 / std::sqrt(nsq[b]) : 0.0;
    }
    if (host_jumped) host_jumped[b] = jump ? 1 : 0;
  }
  // no-jump branch: QR at site 0 with R discarded (mps.py:736-746)
  {
    std::vector<real> sc(scale.begin(), scale.end());
    TJM_HIP_CHECK(hipMemcpyAsync(scal_, sc.data(), B * sizeof(real), hipMemcpyHostToDevice, stream));
    TJM_HIP_CHECK(hipStreamSynchronize(stream));
  }
  if ((rc = launch_scale(S.A[0], a_b0_[0], a_b0_[0], scal_, B, nullptr, nullptr, stream)) != TJM_OK) return rc;
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  if (jumped.empty()) return TJM_OK;

  // ---- channel weights in site-sweep order (stochastic_process.py:139-176)
  std::vector<int> order;
  std::vector<double> wall;
  if ((rc = jump_weights(set, dt_, nsq, jumped, order, wall)) != TJM_OK) return rc;
  // operator table: per process one (or two, for long-range factors) d x d matrices
  std::vector<cplx> optab;
  std::vector<int> op_first(noise_.size());
  for (size_t k = 0; k < noise_.size(); ++k) {
    op_first[k] = (int)(optab.size() / (d * d));
    const NoiseProc& p = noise_[k];
    if (p.nsites == 1) optab.insert(optab.end(), p.mat, p.mat + d * d);
    else { optab.insert(optab.end(), p.f0, p.f0 + d * d); optab.insert(optab.end(), p.f1, p.f1 + d * d); }
  }
  // adjacent two-site operators (d^2 x d^2) live in a second table behind the one-site ones
  const int slot = d * d * d * d;
  std::vector<cplx> optab2;
  std::vector<int> op2_index(noise_.size(), -1);
  for (size_t k = 0; k < noise_.size(); ++k) {
    const NoiseProc& p = noise_[k];
    if (p.nsites == 2 && p.site1 == p.site0 + 1) {
      op2_index[k] = (int)(optab2.size() / slot);
      optab2.insert(optab2.end(), p.mat, p.mat + slot);
    }
  }
  const size_t tab2_off = ((optab.size() + slot - 1) / slot) * slot;
  if (tab2_off + optab2.size() > (size_t)(L + 64) * MSLOT) return TJM_ERR_WORKSPACE;
  if (!optab2.empty()) TJM_HIP_CHECK(hipMemcpyAsync(ops_ + tab2_off, optab2.data(), optab2.size() * sizeof(cplx), hipMemcpyHostToDevice, stream));
  if (optab.size() > (size_t)(L + 64) * MSLOT) return TJM_ERR_WORKSPACE;
  TJM_HIP_CHECK(hipMemcpyAsync(ops_, optab.data(), optab.size() * sizeof(cplx), hipMemcpyHostToDevice, stream));

  std::vector<int> opi(B, -1), opi2(B, -1), js(B, -1), js2(B, -1), adj_site(B, -1), adj_op(B, -1);
  bool any_adjacent = false;
  unitary_jump_.assign(B, 0);
  bool any_second = false;
  std::vector<double> w(order.size());
  for (size_t jb = 0; jb < jumped.size(); ++jb) {
    const int b = jumped[jb];
    double tot = 0.0;
    for (size_t c = 0; c < order.size(); ++c) { w[c] = wall[jb * order.size() + c]; tot += w[c]; }
    if (!(tot > 0.0) || !std::isfinite(tot)) return TJM_ERR_NUMERIC;  // stochastic_process.py:178-186
    // rng.choice(n, p): cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(cdf, u, side="right")
    std::vector<double> cdf(order.size());
    double run = 0.0;
    for (size_t c = 0; c < order.size(); ++c) { run += w[c] / tot; cdf[c] = run; }
    const double last = cdf.back();
    size_t choice = order.size() - 1;
    for (size_t c = 0; c < order.size(); ++c) if (u_choice[b] < cdf[c] / last) { choice = c; break; }
    const NoiseProc& p = noise_[order[choice]];
    js[b] = p.site0;
    unitary_jump_[b] = p.pauli ? 1 : 0;
    if (p.nsites == 2 && p.site1 == p.site0 + 1) {
      // adjacent pair: merged application + truncated split, never the one-tensor shortcut at or left of the pair
      adj_site[b] = p.site0;
      adj_op[b] = op2_index[order[choice]];
      js2[b] = p.site1;
      unitary_jump_[b] = 0;
      any_adjacent = true;
      continue;
    }
    opi[b] = op_first[order[choice]];
    if (p.nsites == 2) { js2[b] = p.site1; opi2[b] = op_first[order[choice]] + 1; any_second = true; }
  }
  // device tables for the per-trajectory site application
  std::vector<cplx*> sp(L);
  std::vector<long> sb(L), sr(L);
  for (int i = 0; i < L; ++i) { sp[i] = S.A[i]; sb[i] = a_b0_[i]; sr[i] = (long)cap[i] * cap[i + 1]; }
  // reuse the tail of T2 as scratch for the small tables
  char* scratch = reinterpret_cast<char*>(T2);
  cplx** d_sp = reinterpret_cast<cplx**>(scratch);
  long* d_sb = reinterpret_cast<long*>(scratch + align_up(L * sizeof(cplx*)));
  long* d_sr = reinterpret_cast<long*>(scratch + 2 * align_up(L * sizeof(cplx*)));
  TJM_HIP_CHECK(hipMemcpyAsync(d_sp, sp.data(), L * sizeof(cplx*), hipMemcpyHostToDevice, stream));
  TJM_HIP_CHECK(hipMemcpyAsync(d_sb, sb.data(), L * sizeof(long), hipMemcpyHostToDevice, stream));
  TJM_HIP_CHECK(hipMemcpyAsync(d_sr, sr.data(), L * sizeof(long), hipMemcpyHostToDevice, stream));
  const int nj = (int)jumped.size();
  // Certified trajectories (dissipate: no bond of theirs is near the truncation threshold; the state is unchanged since - same
  // checksum) with a unitary jump: the QR walk to the last site, the jump and the SVD sweep back are a gauge move around a local
  // unitary - the operator is applied where the state is (right-canonical, centre 0) and the trajectory is done.
  std::vector<int> slow;
  std::vector<char> fast(B, 0);
  {
    bool cert_live = cert_set_ == set;
    std::vector<unsigned long long> now;
    if (cert_live) {
      bool any = false;
      for (int b : jumped) any = any || (cert_ok_[b] && unitary_jump_[b] && adj_site[b] < 0);
      if (any) {
        now.resize(B);
        if ((rc = state_checksum(set, nullptr, B, now.data())) != TJM_OK) return rc;
      } else cert_live = false;
    }
    for (int b : jumped) {
      if (cert_live && cert_ok_[b] && unitary_jump_[b] && adj_site[b] < 0 && now[b] == cert_sum_[b]) { fast[b] = 1; ++stat_cert_jumps; }
      else slow.push_back(b);
    }
    cert_set_ = -1;  // consumed: the jumps below change the state
  }
  const int ns = (int)slow.size();
  // create_probability_distribution (stochastic_process.py:139-176) walks the orthogonality centre 0 -> L-1 by QR on the state
  // itself, so the jump operator meets a LEFT-canonical chain with the centre on the last site.  The weights above do not
  // depend on the gauge, but the truncations of the renormalising sweep below do: reproduce the gauge move.
  if (ns > 0) {
    TJM_HIP_CHECK(hipMemcpyAsync(ids_, slow.data(), ns * sizeof(int), hipMemcpyHostToDevice, stream));
    if (sweep_ok_) {  // small bonds: the whole walk in one launch
      std::vector<SmallSweepStep> steps;
      for (int i = 0; i + 1 < L; ++i) {
        SmallSweepStep st{};
        st.site = i; st.kind = 3; st.op = 0;
        steps.push_back(st);
      }
      if ((rc = run_sweep(set, steps, ids_, ns)) != TJM_OK) return rc;
    } else {
      for (int i = 0; i + 1 < L; ++i)
        if ((rc = qr_shift_right(S, i, ids_, ns)) != TJM_OK) return rc;
    }
    TJM_HIP_CHECK(hipStreamSynchronize(stream));
  }
  TJM_HIP_CHECK(hipMemcpyAsync(ids_, jumped.data(), nj * sizeof(int), hipMemcpyHostToDevice, stream));
  TJM_HIP_CHECK(hipMemcpyAsync(opidx_, opi.data(), B * sizeof(int), hipMemcpyHostToDevice, stream));
  TJM_HIP_CHECK(hipMemcpyAsync(jsite_, js.data(), B * sizeof(int), hipMemcpyHostToDevice, stream));
  hipLaunchKernelGGL(apply_local_multi_kernel, dim3(64, nj), dim3(256), 0, stream, d_sp, d_sb, d_sr, d, ops_, opidx_, jsite_, ids_);
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  if (any_second) {
    TJM_HIP_CHECK(hipMemcpyAsync(opidx_, opi2.data(), B * sizeof(int), hipMemcpyHostToDevice, stream));
    TJM_HIP_CHECK(hipMemcpyAsync(jsite_, js2.data(), B * sizeof(int), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(apply_local_multi_kernel, dim3(64, nj), dim3(256), 0, stream, d_sp, d_sb, d_sr, d, ops_, opidx_, jsite_, ids_);
    TJM_HIP_CHECK(hipStreamSynchronize(stream));
  }
  if (any_adjacent) {
    // adjacent two-site jumps: group the trajectories by pair position
    for (int site = 0; site + 1 < L; ++site) {
      std::vector<int> lst;
      std::vector<int> opsel(B, -1);
      for (int b : jumped)
        if (adj_site[b] == site) { lst.push_back(b); opsel[b] = adj_op[b]; }
      if (lst.empty()) continue;
      TJM_HIP_CHECK(hipMemcpyAsync(ids_, lst.data(), lst.size() * sizeof(int), hipMemcpyHostToDevice, stream));
      TJM_HIP_CHECK(hipMemcpyAsync(opidx_, opsel.data(), B * sizeof(int), hipMemcpyHostToDevice, stream));
      if ((rc = two_site_op(S, site, ops_ + tab2_off, opidx_, ids_, (int)lst.size(), 1)) != TJM_OK) return rc;
      TJM_HIP_CHECK(hipStreamSynchronize(stream));
    }
    TJM_HIP_CHECK(hipMemcpyAsync(ids_, jumped.data(), nj * sizeof(int), hipMemcpyHostToDevice, stream));
  }
  // ---- normalize("B", "SVD") on the jumped trajectories (mps.py:815-839): two-site SVD sweep right -> left (discarded weight
  // 1e-12, no cap), then drop R at site 0.  The chain is left-canonical with the centre at L-1 except for the tensor the jump
  // made non-isometric.  Where the left tensor of the pair (i-1, i) is left-isometric, theta = A_{i-1} C_i has the singular
  // values of the centre tensor C_i alone (svd_shift_left); only the pair whose left tensor is the broken one needs the
  // two-site SVD.  Broken: the site of a non-unitary one-site jump; the right site of an adjacent pair (its split leaves
  // U on the left site and S V^H on the right one); nothing for Pauli jumps.
  {
    std::vector<int> broken(B, -1);
    for (int b : slow) {
      if (adj_site[b] >= 0) broken[b] = adj_site[b] + 1;
      else if (!unitary_jump_[b]) broken[b] = js[b];
    }
    std::vector<int> lst_short, lst_full;
    int* ids_full = opidx_;  // the operator-index table is no longer needed: reuse it as the second id list
    bool any_broken = false;
    for (int b : slow) any_broken = any_broken || broken[b] >= 0;
    if (ns == 0) {
      // every jump of this call was applied in place
    } else if (sweep_ok_ && !any_broken) {  // small bonds, unitary jumps only: every pair is a plain centre shift - one launch
      std::vector<SmallSweepStep> steps;
      for (int i = L - 1; i >= 1; --i) {
        SmallSweepStep st{};
        st.site = i; st.kind = 2; st.op = 0;
        steps.push_back(st);
      }
      TJM_HIP_CHECK(hipMemcpyAsync(ids_, slow.data(), ns * sizeof(int), hipMemcpyHostToDevice, stream));
      if ((rc = run_sweep(set, steps, ids_, ns)) != TJM_OK) return rc;
      TJM_HIP_CHECK(hipStreamSynchronize(stream));
    } else
    for (int i = L - 1; i >= 1; --i) {
      lst_short.clear();
      lst_full.clear();
      for (int b : slow) (broken[b] == i - 1 ? lst_full : lst_short).push_back(b);
      if (!lst_short.empty()) {
        TJM_HIP_CHECK(hipMemcpyAsync(ids_, lst_short.data(), lst_short.size() * sizeof(int), hipMemcpyHostToDevice, stream));
        if ((rc = svd_shift_left(S, i, ids_, (int)lst_short.size())) != TJM_OK) return rc;
      }
      if (!lst_full.empty()) {
        TJM_HIP_CHECK(hipMemcpyAsync(ids_full, lst_full.data(), lst_full.size() * sizeof(int), hipMemcpyHostToDevice, stream));
        if ((rc = svd_shift_left_2site(S, i, ids_full, (int)lst_full.size())) != TJM_OK) return rc;
      }
      TJM_HIP_CHECK(hipStreamSynchronize(stream));  // host vectors are reused next iteration
    }
    TJM_HIP_CHECK(hipMemcpyAsync(ids_, jumped.data(), nj * sizeof(int), hipMemcpyHostToDevice, stream));
  }
  if ((rc = launch_normsq(S.A[0], a_b0_[0], a_b0_[0], normsq_, nj, ids_, stream)) != TJM_OK) return rc;
  hipLaunchKernelGGL(rsqrt_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, normsq_, scal_, B);
  if ((rc = launch_scale(S.A[0], a_b0_[0], a_b0_[0], scal_, nj, ids_, nullptr, stream)) != TJM_OK) return rc;
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  return TJM_OK;
}

// ------------------------------------------------------------------------------------------
// Kernel-level parity exports: the single contractions of the sweep on explicit device tensors.
// All operands hold nb == B slots, contiguous per slot; W is a host MPO tensor in the reference's (o, p, l, r) order.
// ------------------------------------------------------------------------------------------
int Engine::upload_w(const double* host_w, int P, int Dl, int Dr, cplx** mv, cplx** envl) {
  if (!bound_ || !host_w || P < 1 || P > d * d || Dl < 1 || Dr < 1 || Dl > Dmax || Dr > Dmax) return TJM_ERR_ARG;
  direct_clear();  // the scratch matrices change content under the same device pointers
  std::vector<cplx> a((size_t)P * Dl * P * Dr), b((size_t)P * Dr * P * Dl);
  for (int o = 0; o < P; ++o) for (int p = 0; p < P; ++p) for (int l = 0; l < Dl; ++l) for (int r = 0; r < Dr; ++r) {
    const double* z = host_w + 2 * ((((size_t)o * P + p) * Dl + l) * Dr + r);
    const cplx v{(real)z[0], (real)z[1]};
    a[(size_t)(o * Dl + l) * (P * Dr) + (p * Dr + r)] = v;
    b[(size_t)(p * Dr + r) * (P * Dl) + (o * Dl + l)] = v;
  }
  TJM_HIP_CHECK(hipMemcpyAsync(Wx_[0], a.data(), a.size() * sizeof(cplx), hipMemcpyHostToDevice, stream));
  TJM_HIP_CHECK(hipMemcpyAsync(Wx_[1], b.data(), b.size() * sizeof(cplx), hipMemcpyHostToDevice, stream));
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  *mv = Wx_[0];
  *envl = Wx_[1];
  return TJM_OK;
}

// project_site (primitives.py:180-204): y = L (W (x R)); x, y [B][P][ca][cb], L [B][ca][Dl][ca], R [B][cb][Dr][cb]
int Engine::x_heff_apply(int nsites, int ca, int cb, int Dl, int Dr, const cplx* x, const cplx* Lenv, const cplx* Renv, const double* host_w, cplx* y,
                         int nb) {
  const int cm = *std::max_element(cap.begin(), cap.end());
  if (nb != B || (nsites != 1 && nsites != 2) || ca < 1 || cb < 1 || ca > cm || cb > cm) return TJM_ERR_ARG;
  const int P = nsites == 1 ? d : d * d;
  cplx *mv, *el;
  int rc;
  if ((rc = upload_w(host_w, P, Dl, Dr, &mv, &el)) != TJM_OK) return rc;
  const long xb = (long)P * ca * cb;
  if ((rc = heff_apply(x, xb, P, ca, cb, Lenv, (long)ca * Dl * ca, Dl, Renv, (long)cb * Dr * cb, Dr, mv, y, xb, nb, nullptr, nullptr)) != TJM_OK) return rc;
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  return TJM_OK;
}

// update_left_environment / update_right_environment (primitives.py:77-136) with bra = ket = A [B][d][ca][cb]
int Engine::x_env_update(int left, int ca, int cb, int Dl, int Dr, const cplx* A, const cplx* env, const double* host_w, cplx* out, int nb) {
  const int cm = *std::max_element(cap.begin(), cap.end());
  if (nb != B || ca < 1 || cb < 1 || ca > cm || cb > cm) return TJM_ERR_ARG;
  cplx *mv, *el;
  int rc;
  if ((rc = upload_w(host_w, d, Dl, Dr, &mv, &el)) != TJM_OK) return rc;
  const long ab = (long)d * ca * cb;
  if (left) rc = env_left_at(A, ab, ca, cb, Dl, Dr, env, (long)ca * Dl * ca, el, out, (long)cb * Dr * cb, nb);
  else rc = env_right_at(A, ab, ca, cb, Dl, Dr, env, (long)cb * Dr * cb, mv, out, (long)ca * Dl * ca, nb);
  if (rc != TJM_OK) return rc;
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  return TJM_OK;
}

// project_bond (primitives.py:207-226): y[p][w] = sum L[u][a][p] C[u][v] R[v][a][w]; C, y [B][cu][cv], L [B][cu][D][cu], R [B][cv][D][cv]
int Engine::x_project_bond(int cu, int cv, int D, const cplx* C, const cplx* Lenv, const cplx* Renv, cplx* y, int nb) {
  const int cm = *std::max_element(cap.begin(), cap.end());
  if (nb != B || cu < 1 || cv < 1 || cu > cm || cv > cm || D < 1 || D > Dmax) return TJM_ERR_ARG;
  int rc;
  const size_t row = (size_t)cu * cv * sizeof(cplx);
  TJM_HIP_CHECK(hipMemcpy2DAsync(V, (size_t)v_b0 * sizeof(cplx), C, row, row, B, hipMemcpyDeviceToDevice, stream));
  if ((rc = bond_apply(V, cu, cv, Lenv, (long)cu * D * cu, Renv, (long)cv * D * cv, D, V + v_ld, nullptr)) != TJM_OK) return rc;
  TJM_HIP_CHECK(hipMemcpy2DAsync(y, row, V + v_ld, (size_t)v_b0 * sizeof(cplx), row, B, hipMemcpyDeviceToDevice, stream));
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  return TJM_OK;
}

// update_site = expm_krylov(project_site) (primitives.py:484-520, matrix_exponential.py:33-173): y = exp(-i dt H_eff) x
int Engine::x_lanczos_expm(int nsites, int ca, int cb, int Dl, int Dr, const cplx* x, const cplx* Lenv, const cplx* Renv, const double* host_w,
                           double dt_, double tol, cplx* y, int nb, long* matvecs) {
  const int cm = *std::max_element(cap.begin(), cap.end());
  if (nb != B || (nsites != 1 && nsites != 2) || ca < 1 || cb < 1 || ca > cm || cb > cm) return TJM_ERR_ARG;
  const int P = nsites == 1 ? d : d * d;
  cplx *mv, *el;
  int rc;
  if ((rc = upload_w(host_w, P, Dl, Dr, &mv, &el)) != TJM_OK) return rc;
  const long n = (long)P * ca * cb;
  TJM_HIP_CHECK(hipMemcpy2DAsync(V, (size_t)v_b0 * sizeof(cplx), x, (size_t)n * sizeof(cplx), (size_t)n * sizeof(cplx), B, hipMemcpyDeviceToDevice, stream));
  std::vector<int> nl(B, (int)n);
  TJM_HIP_CHECK(hipMemcpyAsync(nloc_, nl.data(), B * sizeof(int), hipMemcpyHostToDevice, stream));
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  const double keep_tol = krylov_tol;
  const long mv0 = stat_matvecs;
  krylov_tol = tol;
  rc = krylov_site(nullptr, P, ca, cb, Lenv, (long)ca * Dl * ca, Dl, Renv, (long)cb * Dr * cb, Dr, mv, dt_, nloc_, y, n, 1, P, ca, cb, 0,
                   (long)ca * cb, cb, B, nullptr, nullptr, nullptr);
  krylov_tol = keep_tol;
  if (matvecs) *matvecs = stat_matvecs - mv0;
  if (rc != TJM_OK) return rc;
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  return TJM_OK;
}

// One orthogonality-centre shift of the loaded state (mps.py:719-788): direction +1 moves the centre site -> site + 1,
// -1 moves it site -> site - 1; use_svd selects the truncating SVD shift (discarded weight 1e-12, no cap) instead of QR.
int Engine::x_center_shift(int set, int site, int direction, int use_svd) {
  if (!bound_ || set < 0 || set > 1 || site < 0 || site >= L) return TJM_ERR_ARG;
  if ((direction == 1 && site + 1 >= L) || (direction == -1 && site < 1) || (direction != 1 && direction != -1)) return TJM_ERR_ARG;
  StateSet& S = sets[set];
  int rc;
  if (direction == 1) rc = use_svd ? svd_shift_right(S, site, nullptr, B) : qr_shift_right(S, site);
  else rc = use_svd ? svd_shift_left(S, site, nullptr, B) : qr_shift_left(S, site);
  if (rc != TJM_OK) return rc;
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  return TJM_OK;
}

// create_probability_distribution (stochastic_process.py:139-187) for every resident trajectory of a state with centre 0:
// host_order[k] = index of the k-th process in the reference's order, host_w[b][k] = its unnormalised weight dt * gamma * ||L psi||^2
int Engine::x_jump_weights(int set, double dt_, int* host_order, double* host_w, int* n_out) {
  if (!bound_ || set < 0 || set > 1 || !host_order || !host_w || !n_out) return TJM_ERR_ARG;
  std::vector<double> nsq(B);
  int rc;
  if ((rc = site_normsq0(set, nsq.data())) != TJM_OK) return rc;
  std::vector<int> all(B), order;
  for (int b = 0; b < B; ++b) all[b] = b;
  std::vector<double> w;
  if ((rc = jump_weights(set, dt_, nsq, all, order, w)) != TJM_OK) return rc;
  *n_out = (int)order.size();
  for (size_t k = 0; k < order.size(); ++k) host_order[k] = order[k];
  for (size_t k = 0; k < w.size(); ++k) host_w[k] = w[k];
  for (int b = 0; b < B; ++b) {
    double tot = 0.0;
    for (size_t k = 0; k < order.size(); ++k) tot += w[(size_t)b * order.size() + k];
    if (!order.empty() && (!(tot > 0.0) || !std::isfinite(tot))) return TJM_ERR_NUMERIC;  // stochastic_process.py:178-186
  }
  return TJM_OK;
}

// ------------------------------------------------------------------------------------------
// Site-level steps for sweeps whose schedule is decided on the host per trajectory (dynamic TDVP, integrators.py:294-511: a site
// takes the two-site branch while its bond is below max_bond_dim and the one-site branch with a QR bond transfer once it has
// reached it, so one lock-step batch splits into two index lists at every site).  host_ids = nullptr: every trajectory.
// ------------------------------------------------------------------------------------------
int Engine::upload_ids(const int* host_ids, int n, const int** dev) {
  *dev = nullptr;
  if (!host_ids) return TJM_OK;
  if (n < 1 || n > B) return TJM_ERR_ARG;
  for (int k = 0; k < n; ++k) if (host_ids[k] < 0 || host_ids[k] >= B) return TJM_ERR_ARG;
  TJM_HIP_CHECK(hipMemcpyAsync(ids_, host_ids, (size_t)n * sizeof(int), hipMemcpyHostToDevice, stream));
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  *dev = ids_;
  return TJM_OK;
}

// right environments of the whole chain and the left boundary (primitives.py:139-174, integrators.py:186-193)
int Engine::step_env_init(int set) {
  if (!bound_ || set < 0 || set > 1) return TJM_ERR_ARG;
  StateSet& S = sets[set];
  int rc;
  if ((rc = launch_identity_env(Renv_[L - 1], r_b0_[L - 1], cap[L], Dm[L], B, stream)) != TJM_OK) return rc;
  for (int i = L - 1; i >= 1; --i)
    if ((rc = env_right(S, i)) != TJM_OK) return rc;
  return launch_identity_env(Lenv_[0], l_b0_[0], cap[0], Dm[0], B, stream);
}

// merge (i, i+1), exp(-i dt H_eff), split_tdvp (dist 0 = "right", 1 = "left"; capped = 0: dynamic=True, no max_bond_dim)
int Engine::step_two_site(int set, int i, double dt_, int dist, int capped, const int* host_ids, int n) {
  if (!bound_ || set < 0 || set > 1 || i < 0 || i + 1 >= L || (dist != 0 && dist != 1)) return TJM_ERR_ARG;
  const int* dev;
  int rc;
  if ((rc = upload_ids(host_ids, n, &dev)) != TJM_OK) return rc;
  return two_site_update(sets[set], i, dt_, dist, dev, host_ids ? n : B, capped != 0);
}

int Engine::step_one_site(int set, int i, double dt_, const int* host_ids, int n) {
  if (!bound_ || set < 0 || set > 1 || i < 0 || i >= L) return TJM_ERR_ARG;
  const int* dev;
  int rc;
  if ((rc = upload_ids(host_ids, n, &dev)) != TJM_OK) return rc;
  return one_site_update(sets[set], i, dt_, dev, host_ids ? n : B);
}

// left = 1: Lenv[i+1] from Lenv[i] and A_i ; left = 0: Renv[i-1] from Renv[i] and A_i
int Engine::step_env(int set, int i, int left, const int* host_ids, int n) {
  if (!bound_ || set < 0 || set > 1 || i < 0 || i >= L || (left && i + 1 >= L) || (!left && i < 1)) return TJM_ERR_ARG;
  const int* dev;
  int rc;
  if ((rc = upload_ids(host_ids, n, &dev)) != TJM_OK) return rc;
  return left ? env_left(sets[set], i, dev, host_ids ? n : B) : env_right(sets[set], i, dev, host_ids ? n : B);
}

// The cut of the one-site branch of sweep_dynamic (integrators.py:361-364, 452-455): a thin QR whose new bond came out above
// max_bond_dim is sliced back to it, "site_tensor[:, :, :cap]; bond_tensor[:cap, :]".  In this storage: the new bond's table
// entry becomes cap, the dropped Q columns (rows of the left bond for the leftward step) and the dropped rows / columns of the bond
// matrix along the NEW index are set to zero (entries beyond a bond are zero everywhere), the Krylov length shrinks with it.
// One workgroup per trajectory.  right = 1: A_i[p][a][k >= cap] = 0, C[k >= cap][*] = 0 ; right = 0: A_i[p][k >= cap][r] = 0, C^T[*][k >= cap] = 0.
__global__ __launch_bounds__(256) void clip_new_bond_kernel(cplx* __restrict__ A, long a_b0, int d, int ca, int cb, int* chi, int stride, int site, int right,
                                                           int capv, cplx* __restrict__ Cm, int cdim, int* nloc, const int* ids) {
  int b = blockIdx.x;
  if (ids) b = ids[b];
  int* slot = chi + (long)b * stride + site + (right ? 1 : 0);
  const int k = *slot;
  __syncthreads();
  if (k <= capv) return;
  cplx* Ab = A + (long)b * a_b0;
  const long total = (long)d * ca * cb;
  for (long e = threadIdx.x; e < total; e += blockDim.x) {
    const int r = (int)(e % cb), l = (int)((e / cb) % ca);
    if ((right ? r : l) >= capv) Ab[e] = cplx{0.0, 0.0};
  }
  cplx* Cb = Cm + (long)b * cdim * cdim;
  for (int e = threadIdx.x; e < cdim * cdim; e += blockDim.x) {
    const int row = e / cdim, col = e - row * cdim;
    if ((right ? row : col) >= capv) Cb[e] = cplx{0.0, 0.0};
  }
  if (threadIdx.x == 0) {
    *slot = capv;
    if (nloc) nloc[b] = nloc[b] / k * capv;
  }
}

// The bond transfer of the one-site branch (integrators.py:352-377 / 441-466, the body of sweep_1site): thin QR of site i
// (right = 1: A_i = Q C, right = 0: A_i = C^T Q), environment update with Q, exp(-i dt H_bond) on C, C into the neighbour.
// right = 1: A_i = Q C, C evolves over dt, A_{i+1} <- C A_{i+1}.  right = 0: A_i = C^T Q, C evolves, A_{i-1} <- A_{i-1} C.
//
// Deviation from the reference, on purpose: sweep_dynamic's leftward one-site branch (integrators.py:450-474) calls left_qr, which
// hands back R^T = C[left][new] (decompositions.py:82), and then transposes once more - a line taken over from the fixed one-site
// sweep (integrators.py:140) where R comes straight from np.linalg.qr.  The matrix that is evolved and absorbed there is R, whose
// indices are contracted the wrong way round: the result changes under a change of gauge on the bond, so it depends on the signs
// and phases LAPACK's SVD and QR happened to pick in the steps before (and is only defined when the factor is square).  No other
// implementation can reproduce those numbers; this step does what the fixed one-site sweep does, the projector-splitting step.
// oracle/tjm_oracle.py restates the reference line for line behind Params.reference_dynamic_transpose (default on, pinned to the
// reference's fixtures); the engine is compared with the oracle with that switch off wherever a bond sits at the cap.
int Engine::step_qr_bond(int set, int i, int right, double dt_, int max_bond, const int* host_ids, int n) {
  if (!bound_ || set < 0 || set > 1 || i < 0 || i >= L || (right && i + 1 >= L) || (!right && i < 1)) return TJM_ERR_ARG;
  StateSet& S = sets[set];
  const int* dev;
  int rc;
  if ((rc = upload_ids(host_ids, n, &dev)) != TJM_OK) return rc;
  const int nb = host_ids ? n : B;
  if (right) {
    const int cb = cap[i + 1], cc = cap[i + 2];
    if ((rc = qr_site(S, i, true, dev, nb)) != TJM_OK) return rc;
    if (max_bond > 0 && max_bond < cb)
      hipLaunchKernelGGL(clip_new_bond_kernel, dim3(nb), dim3(256), 0, stream, S.A[i], a_b0_[i], d, cap[i], cb, S.chi, L + 1, i, 1, max_bond, Cm_, cb, nloc_, dev);
    if ((rc = env_left(S, i, dev, nb)) != TJM_OK) return rc;
    TJM_HIP_CHECK(hipMemcpy2DAsync(V, (size_t)v_b0 * sizeof(cplx), Cm_, (size_t)cb * cb * sizeof(cplx), (size_t)cb * cb * sizeof(cplx), B,
                                   hipMemcpyDeviceToDevice, stream));
    ApplyFn f = [&](const cplx* x, cplx* y, const int* active) {
      return bond_apply(x, cb, cb, Lenv_[i + 1], l_b0_[i + 1], Renv_[i], r_b0_[i], Dm[i + 1], y, active, nb, dev);
    };
    if ((rc = krylov_core(f, cb * cb, dt_, nloc_, Cm_, (long)cb * cb, 1, 1, cb, cb, 0, 0, cb, nb, dev)) != TJM_OK) return rc;
    GemmDesc g = blank_gemm();  // T1[p][l][r] = C[l][x] A_{i+1}[p][x][r]
    g.A = Cm_; g.B = S.A[i + 1]; g.C = T1;
    g.M = cb; g.K = cb; g.N = cc;
    g.a_rs = cb; g.a_cs = 1; g.b_rs = cc; g.b_cs = 1; g.c_rs = cc;
    g.nb0 = nb; g.nb1 = d; g.a_b0 = (long)cb * cb; g.b_b0 = a_b0_[i + 1]; g.b_b1 = (long)cb * cc; g.c_b0 = t_b0; g.c_b1 = (long)cb * cc;
    g.ids = dev;
    if ((rc = gemm(g)) != TJM_OK) return rc;
    return copy_back(S.A[i + 1], a_b0_[i + 1], T1, t_b0, a_b0_[i + 1], dev, nb);
  }
  const int cz = cap[i - 1], ca = cap[i];
  if ((rc = qr_site(S, i, false, dev, nb)) != TJM_OK) return rc;
  if (max_bond > 0 && max_bond < ca)
    hipLaunchKernelGGL(clip_new_bond_kernel, dim3(nb), dim3(256), 0, stream, S.A[i], a_b0_[i], d, ca, cap[i + 1], S.chi, L + 1, i, 0, max_bond, Cm_, ca, nloc_, dev);
  if ((rc = env_right(S, i, dev, nb)) != TJM_OK) return rc;
  TJM_HIP_CHECK(hipMemcpy2DAsync(V, (size_t)v_b0 * sizeof(cplx), Cm_, (size_t)ca * ca * sizeof(cplx), (size_t)ca * ca * sizeof(cplx), B,
                                 hipMemcpyDeviceToDevice, stream));
  ApplyFn f = [&](const cplx* x, cplx* y, const int* active) {
    return bond_apply(x, ca, ca, Lenv_[i], l_b0_[i], Renv_[i - 1], r_b0_[i - 1], Dm[i], y, active, nb, dev);
  };
  if ((rc = krylov_core(f, ca * ca, dt_, nloc_, Cm_, (long)ca * ca, 1, 1, ca, ca, 0, 0, ca, nb, dev)) != TJM_OK) return rc;
  GemmDesc g = blank_gemm();  // T1[(p,l)][r] = A_{i-1}[(p,l)][x] C^T[x][r]
  g.A = S.A[i - 1]; g.B = Cm_; g.C = T1;
  g.M = d * cz; g.K = ca; g.N = ca;
  g.a_rs = ca; g.a_cs = 1; g.b_rs = ca; g.b_cs = 1; g.c_rs = ca;
  g.nb0 = nb; g.a_b0 = a_b0_[i - 1]; g.b_b0 = (long)ca * ca; g.c_b0 = t_b0;
  g.ids = dev;
  if ((rc = gemm(g)) != TJM_OK) return rc;
  return copy_back(S.A[i - 1], a_b0_[i - 1], T1, t_b0, a_b0_[i - 1], dev, nb);
}

// _sync_bond_dim (sweep_utils.py:110-163) where it truncates: merge (bond, bond+1), split with sqrt(S) into both factors under the
// run's truncation rule, max_bond_dim = target, min_keep 1.  (Its padding branches do nothing here: a bond has ONE dimension per
// trajectory in this storage and the entries beyond it are zero.)
int Engine::step_cap_bond(int set, int bond, int target, const int* host_ids, int n) {
  if (!bound_ || set < 0 || set > 1 || bond < 0 || bond + 1 >= L || target < 1) return TJM_ERR_ARG;
  StateSet& S = sets[set];
  const int* dev;
  int rc;
  if ((rc = upload_ids(host_ids, n, &dev)) != TJM_OK) return rc;
  const int nb = host_ids ? n : B;
  if ((rc = merge_matrix_layout(S, bond, dev, nb)) != TJM_OK) return rc;
  return split(S, bond, 2, trunc_mode, svd_threshold, target, 1, dev, nb);
}

// ------------------------------------------------------------------------------------------
// One whole sweep of the dynamic TDVP in one call (round 6): sweep_dynamic (integrators.py:294-511) with its branch lists formed
// HERE - per site one column of the bond table (B integers) comes to the host instead of the whole table crossing the C ABI and
// ctypes several times, and the host-side sequencing that yaqs_amd/tjm.py did in Python (its _sweep_dynamic, kept as the readable
// mirror) is this loop.  max_bond < 1: no cap (every site takes the two-site branch).
// ------------------------------------------------------------------------------------------
int Engine::bond_column(int set, int bond, std::vector<int>& out) {
  out.resize(B);
  TJM_HIP_CHECK(hipMemcpy2DAsync(out.data(), sizeof(int), sets[set].chi + bond, (size_t)(L + 1) * sizeof(int), sizeof(int), B, hipMemcpyDeviceToHost, stream));
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  return TJM_OK;
}

int Engine::sweep_dynamic(int set, int max_bond, double dt_) {
  if (!bound_ || set < 0 || set > 1) return TJM_ERR_ARG;
  int rc;
  std::vector<int> dims, one, two;
  auto lists = [&](int bond) -> int {  // the bond that decides the branch of this site
    one.clear();
    two.clear();
    if (max_bond < 1) { for (int b = 0; b < B; ++b) two.push_back(b); return TJM_OK; }
    const int r = bond_column(set, bond, dims);
    if (r != TJM_OK) return r;
    for (int b = 0; b < B; ++b) (dims[b] >= max_bond ? one : two).push_back(b);
    return TJM_OK;
  };
  if (max_bond >= 1) {  // _cap_bonds (sweep_utils.py:280-302): bonds the previous sweep left above the cap
    for (int bond = 0; bond + 1 < L; ++bond) {
      if ((rc = bond_column(set, bond + 1, dims)) != TJM_OK) return rc;
      one.clear();
      for (int b = 0; b < B; ++b) if (dims[b] > max_bond) one.push_back(b);
      if (!one.empty() && (rc = step_cap_bond(set, bond, max_bond, one.data(), (int)one.size())) != TJM_OK) return rc;
    }
  }
  if ((rc = step_env_init(set)) != TJM_OK) return rc;
  for (int i = 0; i < L; ++i) {  // left to right (integrators.py:340-424)
    if ((rc = lists(i + 1)) != TJM_OK) return rc;
    if (!one.empty()) {
      if ((rc = step_one_site(set, i, 0.5 * dt_, one.data(), (int)one.size())) != TJM_OK) return rc;
      if (i != L - 1 && (rc = step_qr_bond(set, i, 1, -0.5 * dt_, max_bond, one.data(), (int)one.size())) != TJM_OK) return rc;
    }
    if (!two.empty() && i != L - 1) {
      const int nt = (int)two.size();
      if ((rc = step_two_site(set, i, 0.5 * dt_, 0, 0, two.data(), nt)) != TJM_OK) return rc;
      if (i == L - 2) {
        if ((rc = step_env(set, i + 1, 0, two.data(), nt)) != TJM_OK) return rc;
        if ((rc = step_env(set, i, 1, two.data(), nt)) != TJM_OK) return rc;
      } else {
        if ((rc = step_env(set, i, 1, two.data(), nt)) != TJM_OK) return rc;
        if ((rc = step_one_site(set, i + 1, -0.5 * dt_, two.data(), nt)) != TJM_OK) return rc;
      }
    }
  }
  for (int i = L - 1; i >= 0; --i) {  // right to left (integrators.py:427-505)
    if ((rc = lists(i)) != TJM_OK) return rc;
    if (!one.empty()) {
      if ((rc = step_one_site(set, i, 0.5 * dt_, one.data(), (int)one.size())) != TJM_OK) return rc;
      if (i != 0 && (rc = step_qr_bond(set, i, 0, -0.5 * dt_, max_bond, one.data(), (int)one.size())) != TJM_OK) return rc;
    }
    if (!two.empty() && i != 0) {
      const int nt = (int)two.size();
      if ((rc = step_two_site(set, i - 1, 0.5 * dt_, 1, 0, two.data(), nt)) != TJM_OK) return rc;
      if ((rc = step_env(set, i, 0, two.data(), nt)) != TJM_OK) return rc;
      if (i != 1 && (rc = step_one_site(set, i - 1, -0.5 * dt_, two.data(), nt)) != TJM_OK) return rc;
    }
  }
  return TJM_OK;
}

// One half-sweep of the BUG integrator (bug_sweep, bug.py:128-196) in one call: the prepared centres and left environments, the walk
// from the last site to site 1, the root.
int Engine::bug_sweep(int set, double dt_) {
  int rc;
  if ((rc = step_bug_prepare(set)) != TJM_OK) return rc;
  for (int site = L - 1; site >= 1; --site)
    if ((rc = step_bug_site(set, site, dt_)) != TJM_OK) return rc;
  return step_bug_root(set, dt_);
}

// ------------------------------------------------------------------------------------------
// Long-range gate as a matrix product operator (digital_tjm.py:536-557: MPO.from_gate(...).multiply(state), mpo.py:1511-1548)
// ------------------------------------------------------------------------------------------
// One site of the product.  The gate is U = sum_k P_k (x) Q_k on sites (first, last) with identity threads in between
// (extend_gate, gate_library.py:66-126); the fused virtual legs carry the MPS index first (mpo_utils.py:27-56):
//   mode 0 (first):  out[p][a][b r + k]       = sum_q P_k[p][q] A[q][a][b]
//   mode 1 (thread): out[p][a r + k][b r + k] = A[p][a][b]
//   mode 2 (last):   out[p][a r + k][b]       = sum_q Q_k[p][q] A[q][a][b]
// Entries beyond the storage are dropped (gate_mpo_dim_kernel raises the overflow flag).
__global__ __launch_bounds__(256) void gate_mpo_site_kernel(const cplx* __restrict__ A, long a_b0, int d, int ca, int cb, const int* chi, int stride, int site,
                                                           int mode, int r, const cplx* __restrict__ ops, cplx* __restrict__ out, long out_b0) {
  const int b = blockIdx.y;
  const int xa = chi[(long)b * stride + site], xb = chi[(long)b * stride + site + 1];
  const cplx* Ab = A + (long)b * a_b0;
  const long n = (long)d * ca * cb;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const int bo = (int)(e % cb);
    const long q0 = e / cb;
    const int ao = (int)(q0 % ca), p = (int)(q0 / ca);
    int a = ao, bi = bo, ka = -1, kb = -1;
    if (mode != 0) { ka = ao % r; a = ao / r; }
    if (mode != 2) { kb = bo % r; bi = bo / r; }
    cplx v{0.0, 0.0};
    if (a < xa && bi < xb && (mode != 1 || ka == kb)) {
      if (mode == 1) {
        v = Ab[((long)p * ca + a) * cb + bi];
      } else {
        const cplx* O = ops + (long)(mode == 0 ? kb : ka) * d * d;
        for (int q = 0; q < d; ++q) cfma(v, O[p * d + q], Ab[((long)q * ca + a) * cb + bi]);
      }
    }
    out[(long)b * out_b0 + e] = v;
  }
}
// bond k grows by the factor r; a product that does not fit the storage is flagged
__global__ void gate_mpo_dim_kernel(int* chi, int stride, int k, int r, int capk, int* overflow, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int x = chi[(long)b * stride + k] * r;
  if (x > capk) { x = capk; atomicOr(overflow, 1); }
  chi[(long)b * stride + k] = x;
}

int Engine::apply_gate_mpo(int set, int first, int last, int r, const double* host_left, const double* host_right) {
  if (!bound_ || set < 0 || set > 1 || first < 0 || last >= L || last - first < 1 || r < 1 || r > d * d || !host_left || !host_right) return TJM_ERR_ARG;
  StateSet& S = sets[set];
  const size_t dd = (size_t)d * d;
  cplx* tab = ops_ + (size_t)(L + 8) * dd * dd;  // behind the slots of apply_single / tebd_gate
  if (int rcu = upload_c(tab, host_left, (size_t)r * dd, stream)) return rcu;
  if (int rcu = upload_c(tab + (size_t)r * dd, host_right, (size_t)r * dd, stream)) return rcu;
  TJM_HIP_CHECK(hipStreamSynchronize(stream));
  int rc;
  for (int k = first; k <= last; ++k) {
    const int mode = (k == first) ? 0 : (k == last ? 2 : 1);
    const long total = a_b0_[k];
    int gx = (int)((total + 1023) / 1024);
    if (gx < 1) gx = 1;
    if (gx > 128) gx = 128;
    hipLaunchKernelGGL(gate_mpo_site_kernel, dim3(gx, B), dim3(256), 0, stream, S.A[k], a_b0_[k], d, cap[k], cap[k + 1], S.chi, L + 1, k, mode, r,
                       mode == 2 ? tab + (size_t)r * dd : tab, T1, t_b0);
    TJM_HIP_CHECK(hipGetLastError());
    if ((rc = copy_back(S.A[k], a_b0_[k], T1, t_b0, a_b0_[k], nullptr, B)) != TJM_OK) return rc;
  }
  for (int k = first + 1; k <= last; ++k)  // after every site has read the old dimensions
    hipLaunchKernelGGL(gate_mpo_dim_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, S.chi, L + 1, k, r, cap[k], overflow_, B);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

// ------------------------------------------------------------------------------------------
// Steps of the Basis-Update and Galerkin integrator (core/methods/bug.py:35-257) for the whole batch.
// Set `set` holds the state (right-canonical basis tensors, centre 0: the reference's state.tensors), set 2 the coefficient-bearing
// centres (canon_center_tensors), set 3 is scratch (the Q factors of the preparation, then predictor / stacked basis / new basis).
// The basis-change matrix M of the site below lives in E_ ([B][c][c'], leading dimension = capacity of that bond).
// ------------------------------------------------------------------------------------------
namespace {
__global__ void chi_col_copy_kernel(int* dst, const int* src, int stride, int col, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) dst[(long)b * stride + col] = src[(long)b * stride + col];
}
__global__ void chi_reverse_kernel(int* dst, const int* src, int stride, int L, int B) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < B * (L + 1)) {
    const int b = t / (L + 1), k = t % (L + 1);
    dst[(long)b * stride + k] = src[(long)b * stride + (L - k)];
  }
}
// out[b][p][c][a] = in[b][p][a][c]   (site tensor of the reversed chain, mps.py:680-698)
__global__ __launch_bounds__(256) void flip_site_kernel(const cplx* __restrict__ in, long b0, int d, int ca, int cb, cplx* __restrict__ out) {
  const int b = blockIdx.y;
  const long n = (long)d * ca * cb;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const int a = (int)(e % ca);
    const long q = e / ca;
    const int c = (int)(q % cb), p = (int)(q / cb);
    out[(long)b * b0 + e] = in[(long)b * b0 + ((long)p * ca + a) * cb + c];
  }
}
// Stack along the left bond without gaps (build_trial_basis, bug.py:78-80): out[p][a'][c] = ret[p][a'][c] for a' < n_ret,
// pred[p][a' - n_ret][c] for n_ret <= a' < n_ret + n_pred, zero beyond.  The bond table is updated by stack_dims_kernel AFTERWARDS
// (the workgroups of one trajectory read the old dimensions at different times).
__global__ __launch_bounds__(256) void stack_left_kernel(const cplx* __restrict__ ret, const cplx* __restrict__ pred, long b0, int d, int ca, int cb,
                                                        const int* chi_ret, const int* chi_pred, int stride, int col, cplx* __restrict__ out,
                                                        long out_b0) {
  const int b = blockIdx.y;
  const int nr = chi_ret[(long)b * stride + col], np_ = chi_pred[(long)b * stride + col];
  const int tot = (nr + np_ > ca) ? ca : nr + np_;
  const long n = (long)d * ca * cb;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % cb);
    const long q = e / cb;
    const int a = (int)(q % ca), p = (int)(q / ca);
    cplx v{0.0, 0.0};
    if (a < nr) v = ret[(long)b * b0 + e];
    else if (a < tot) v = pred[(long)b * b0 + ((long)p * ca + (a - nr)) * cb + c];
    out[(long)b * out_b0 + e] = v;
  }
}
// new left bond of the stack: n_ret + n_pred, clipped to the storage (flagged: the caller re-runs on a larger engine)
__global__ void stack_dims_kernel(const int* chi_ret, const int* chi_pred, int* chi_out, int stride, int col, int ca, int* overflow, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int tot = chi_ret[(long)b * stride + col] + chi_pred[(long)b * stride + col];
  if (tot > ca) { tot = ca; atomicOr(overflow, 1); }
  chi_out[(long)b * stride + col] = tot;
}
__global__ void bond_identity_kernel(cplx* M, long m_b0, int n, int B) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < (long)B * n * n) {
    const long r = t % ((long)n * n);
    M[(t / ((long)n * n)) * m_b0 + r] = cplx{(r / n == r % n) ? real(1) : real(0), 0.0};
  }
}
}  // namespace

int Engine::copy_chi_col(int dst, int src, int col) {
  hipLaunchKernelGGL(chi_col_copy_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, sets[dst].chi, sets[src].chi, L + 1, col, B);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

int Engine::copy_site(int dst, int src, int site) {
  TJM_HIP_CHECK(hipMemcpyAsync(sets[dst].A[site], sets[src].A[site], (size_t)B * a_b0_[site] * sizeof(cplx), hipMemcpyDeviceToDevice, stream));
  int rc;
  if ((rc = copy_chi_col(dst, src, site)) != TJM_OK) return rc;
  return copy_chi_col(dst, src, site + 1);
}

// prepare_canonical_site_tensors (bug.py:35-62): centres canon[i] = R_{i-1} A_i into set 2 and the left environments of the Q factors
int Engine::step_bug_prepare(int set) {
  if (!bound_ || n_sets < 4 || set < 0 || set > 1) return TJM_ERR_ARG;
  int rc;
  if ((rc = copy_state(2, set)) != TJM_OK) return rc;
  StateSet& Cs = sets[2];
  StateSet& Qs = sets[3];
  if ((rc = launch_identity_env(Lenv_[0], l_b0_[0], cap[0], Dm[0], B, stream)) != TJM_OK) return rc;
  for (int i = 1; i < L; ++i) {
    const int cb = cap[i], cc = cap[i + 1];
    if ((rc = copy_site(3, 2, i - 1)) != TJM_OK) return rc;
    if ((rc = qr_site(Qs, i - 1, true)) != TJM_OK) return rc;            // Q into set 3, R into Cm_, new bond into set 3's table
    GemmDesc g = blank_gemm();                                           // canon[i][p][l][r] = R[l][x] canon[i][p][x][r]
    g.A = Cm_; g.B = Cs.A[i]; g.C = T1;
    g.M = cb; g.K = cb; g.N = cc;
    g.a_rs = cb; g.a_cs = 1; g.b_rs = cc; g.b_cs = 1; g.c_rs = cc;
    g.nb0 = B; g.nb1 = d; g.a_b0 = (long)cb * cb; g.b_b0 = a_b0_[i]; g.b_b1 = (long)cb * cc; g.c_b0 = t_b0; g.c_b1 = (long)cb * cc;
    if ((rc = gemm(g)) != TJM_OK) return rc;
    if ((rc = copy_back(Cs.A[i], a_b0_[i], T1, t_b0, a_b0_[i], nullptr, B)) != TJM_OK) return rc;
    if ((rc = copy_chi_col(2, 3, i)) != TJM_OK) return rc;
    if ((rc = env_left_at(Qs.A[i - 1], a_b0_[i - 1], cap[i - 1], cap[i], Dm[i - 1], Dm[i], Lenv_[i - 1], l_b0_[i - 1], WenvL_[i - 1], Lenv_[i],
                          l_b0_[i], B)) != TJM_OK) return rc;
  }
  if ((rc = launch_identity_env(Renv_[L - 1], r_b0_[L - 1], cap[L], Dm[L], B, stream)) != TJM_OK) return rc;
  const int cm = *std::max_element(cap.begin(), cap.end());
  hipLaunchKernelGGL(bond_identity_kernel, dim3((unsigned)(((long)B * cap[L] * cap[L] + 255) / 256)), dim3(256), 0, stream, E_, (long)cm * cm, cap[L], B);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

// _local_update (bug.py:93-125) at `site` (L-1 ... 1) for every trajectory
int Engine::step_bug_site(int set, int site, double dt_) {
  if (!bound_ || n_sets < 4 || set < 0 || set > 1 || site < 1 || site >= L) return TJM_ERR_ARG;
  StateSet& Ts = sets[set];
  StateSet& Cs = sets[2];
  StateSet& Qs = sets[3];
  const int cz = cap[site - 1], ca = cap[site], cb = cap[site + 1];
  const int cm = *std::max_element(cap.begin(), cap.end());
  const long m_b0 = (long)cm * cm;
  int rc;
  // predictor = update_site(lenv[site], right_block, W, working, dt)
  if ((rc = copy_site(3, 2, site)) != TJM_OK) return rc;
  if ((rc = one_site_update(Qs, site, dt_)) != TJM_OK) return rc;
  {  // old_basis_current[p][a][c'] = sum_c old_q[p][a][c] M[c][c']  -> T2
    GemmDesc g = blank_gemm();
    g.A = Ts.A[site]; g.B = E_; g.C = T2;
    g.M = d * ca; g.K = cb; g.N = cb;
    g.a_rs = cb; g.a_cs = 1; g.b_rs = cb; g.b_cs = 1; g.c_rs = cb;
    g.nb0 = B; g.a_b0 = a_b0_[site]; g.b_b0 = m_b0; g.c_b0 = t_b0;
    if ((rc = gemm(g)) != TJM_OK) return rc;
  }
  {  // stacked = [retained | predictor] along the left bond; retained = old_q at the endpoint, the working centre elsewhere
    const bool endpoint = (site == L - 1);
    const StateSet& Rs = endpoint ? Ts : Cs;
    const long n = (long)d * ca * cb;
    int gx = (int)((n + 1023) / 1024);
    if (gx < 1) gx = 1;
    if (gx > 128) gx = 128;
    hipLaunchKernelGGL(stack_left_kernel, dim3(gx, B), dim3(256), 0, stream, Rs.A[site], Qs.A[site], a_b0_[site], d, ca, cb, Rs.chi, Qs.chi, L + 1, site, V,
                       v_b0);
    hipLaunchKernelGGL(stack_dims_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, Rs.chi, Qs.chi, Qs.chi, L + 1, site, ca, overflow_, B);
    TJM_HIP_CHECK(hipGetLastError());
    TJM_HIP_CHECK(hipMemcpy2DAsync(Qs.A[site], (size_t)a_b0_[site] * sizeof(cplx), V, (size_t)v_b0 * sizeof(cplx), (size_t)a_b0_[site] * sizeof(cplx), B,
                                   hipMemcpyDeviceToDevice, stream));
  }
  if ((rc = qr_site(Qs, site, false)) != TJM_OK) return rc;  // new_q = left_qr(stacked): right-isometric, new left bond into set 3's table
  {  // M'[a][b] = sum_{p,c} old_basis_current[p][a][c] conj(new_q[p][b][c])  -> E2_ (leading dimension ca)
    GemmDesc g = blank_gemm();
    g.A = T2; g.B = Qs.A[site]; g.C = E2_;
    g.M = ca; g.K = cb; g.N = ca; g.nks = d;
    g.a_rs = cb; g.a_cs = 1; g.a_ks = (long)ca * cb; g.b_rs = 1; g.b_cs = cb; g.b_ks = (long)ca * cb; g.conjB = 1; g.c_rs = ca;
    g.nb0 = B; g.a_b0 = t_b0; g.b_b0 = a_b0_[site]; g.c_b0 = m_b0;
    if ((rc = gemm(g)) != TJM_OK) return rc;
  }
  // state.tensors[site] = new_q
  TJM_HIP_CHECK(hipMemcpyAsync(Ts.A[site], Qs.A[site], (size_t)B * a_b0_[site] * sizeof(cplx), hipMemcpyDeviceToDevice, stream));
  if ((rc = copy_chi_col(set, 3, site)) != TJM_OK) return rc;
  {  // canon[site-1][p][z][b] = sum_a canon[site-1][p][z][a] M'[a][b]
    GemmDesc g = blank_gemm();
    g.A = Cs.A[site - 1]; g.B = E2_; g.C = T1;
    g.M = d * cz; g.K = ca; g.N = ca;
    g.a_rs = ca; g.a_cs = 1; g.b_rs = ca; g.b_cs = 1; g.c_rs = ca;
    g.nb0 = B; g.a_b0 = a_b0_[site - 1]; g.b_b0 = m_b0; g.c_b0 = t_b0;
    if ((rc = gemm(g)) != TJM_OK) return rc;
    if ((rc = copy_back(Cs.A[site - 1], a_b0_[site - 1], T1, t_b0, a_b0_[site - 1], nullptr, B)) != TJM_OK) return rc;
    if ((rc = copy_chi_col(2, 3, site)) != TJM_OK) return rc;
  }
  // right block of the new basis: Renv[site-1] from Renv[site] and new_q
  if ((rc = env_right_at(Ts.A[site], a_b0_[site], ca, cb, Dm[site], Dm[site + 1], Renv_[site], r_b0_[site], W_[site], Renv_[site - 1], r_b0_[site - 1],
                         B)) != TJM_OK) return rc;
  std::swap(E_, E2_);  // M of the next site
  return TJM_OK;
}

// root update (bug.py:188-196): state.tensors[0] = update_site(lenv[0], right_block, W_0, canon[0], dt)
int Engine::step_bug_root(int set, double dt_) {
  if (!bound_ || n_sets < 4 || set < 0 || set > 1) return TJM_ERR_ARG;
  int rc;
  if ((rc = copy_site(set, 2, 0)) != TJM_OK) return rc;
  return one_site_update(sets[set], 0, dt_);
}

// MPS.flip_network (mps.py:680-698): reversed site order, virtual legs exchanged (the storage capacities are symmetric)
int Engine::step_flip(int set) {
  if (!bound_ || n_sets < 4 || set < 0 || set > 1) return TJM_ERR_ARG;
  for (int i = 0; i < L; ++i) {
    const int j = L - 1 - i;
    if (cap[i] != cap[j + 1] || cap[i + 1] != cap[j]) return TJM_ERR_STATE;
    const long n = (long)d * cap[i] * cap[i + 1];
    int gx = (int)((n + 1023) / 1024);
    if (gx < 1) gx = 1;
    if (gx > 128) gx = 128;
    hipLaunchKernelGGL(flip_site_kernel, dim3(gx, B), dim3(256), 0, stream, sets[set].A[i], a_b0_[i], d, cap[i], cap[i + 1], sets[3].A[j]);
  }
  hipLaunchKernelGGL(chi_reverse_kernel, dim3((unsigned)(((long)B * (L + 1) + 255) / 256)), dim3(256), 0, stream, sets[3].chi, sets[set].chi, L + 1, L, B);
  TJM_HIP_CHECK(hipGetLastError());
  return copy_state(set, 3);
}

// MPS.compress (mps.py:841-899): right-canonical form by QR, truncated SVD sweep left to right ("right" distribution, min_keep 1),
// centre back to site 0 by QR
int Engine::step_compress(int set, double threshold, int max_bond_dim, int mode) {
  if (!bound_ || set < 0 || set > 1 || mode < 0 || mode > 3) return TJM_ERR_ARG;
  if (L == 1) return TJM_OK;
  StateSet& S = sets[set];
  int rc;
  if ((rc = canonicalize_qr(set, L - 1)) != TJM_OK) return rc;
  for (int site = 0; site + 1 < L; ++site) {
    if ((rc = merge_matrix_layout(S, site, nullptr, B)) != TJM_OK) return rc;
    if ((rc = split(S, site, 0, mode, threshold, max_bond_dim, 1, nullptr, B)) != TJM_OK) return rc;
  }
  return canonicalize_qr(set, L - 1);
}

}  // namespace tjm
