// One-call trajectory driver of the C ABI: tjm_engine_run evolves the B resident trajectories through the whole
// analog_tjm_1 / analog_tjm_2 schedule (analog/analog_tjm.py:206-462) and returns results + diagnostics, so a C caller
// needs no per-step host logic.  Also the host-side random streams of core/random_utils.py:20-69, bit-compatible with
// NumPy:  default_rng(SeedSequence([seed, traj, TAG])) -> PCG64 -> Generator.random().
#include <cmath>
#include <cstring>
#include <random>
#include <vector>

#include "../../include/tjm_hip.h"
#include "tjm_engine.h"

namespace tjm {

namespace {

// ---- numpy.random.SeedSequence (bit_generator.pyx): 4-word pool, hashmix / mix constants of the published algorithm
struct SeedSeq {
  static constexpr uint32_t INIT_A = 0x43b0d7e5u, MULT_A = 0x931e8875u, INIT_B = 0x8b51f9ddu, MULT_B = 0x58f38dedu;
  static constexpr uint32_t MIX_MULT_L = 0xca01f9ddu, MIX_MULT_R = 0x4973f715u;
  static constexpr int XSHIFT = 16, POOL = 4;
  uint32_t pool[POOL];

  static uint32_t hashmix(uint32_t value, uint32_t& hash_const) {
    value ^= hash_const;
    hash_const *= MULT_A;
    value *= hash_const;
    value ^= value >> XSHIFT;
    return value;
  }
  static uint32_t mix(uint32_t x, uint32_t y) {
    uint32_t r = MIX_MULT_L * x - MIX_MULT_R * y;
    r ^= r >> XSHIFT;
    return r;
  }
  explicit SeedSeq(const std::vector<uint32_t>& entropy) {
    uint32_t hash_const = INIT_A;
    for (int i = 0; i < POOL; ++i) pool[i] = hashmix(i < (int)entropy.size() ? entropy[i] : 0u, hash_const);
    for (int i_src = 0; i_src < POOL; ++i_src)
      for (int i_dst = 0; i_dst < POOL; ++i_dst)
        if (i_src != i_dst) pool[i_dst] = mix(pool[i_dst], hashmix(pool[i_src], hash_const));
    for (size_t i_src = POOL; i_src < entropy.size(); ++i_src)
      for (int i_dst = 0; i_dst < POOL; ++i_dst) pool[i_dst] = mix(pool[i_dst], hashmix(entropy[i_src], hash_const));
  }
  void generate(uint32_t* out, int n_words) const {
    uint32_t hash_const = INIT_B;
    for (int i = 0; i < n_words; ++i) {
      uint32_t v = pool[i % POOL];
      v ^= hash_const;
      hash_const *= MULT_B;
      v *= hash_const;
      v ^= v >> XSHIFT;
      out[i] = v;
    }
  }
};

// every Python int of the entropy list becomes its own little-endian run of 32-bit words (at least one)
void push_entropy(std::vector<uint32_t>& e, uint64_t v) {
  e.push_back((uint32_t)(v & 0xffffffffu));
  if (v >> 32) e.push_back((uint32_t)(v >> 32));
}

// ---- PCG64 (setseq 128, XSL-RR 64) as seeded by numpy.random.PCG64
struct Pcg64 {
  unsigned __int128 state = 0, inc = 0;
  static unsigned __int128 mult() { return ((unsigned __int128)2549297995355413924ULL << 64) | 4865540595714422341ULL; }
  void step() { state = state * mult() + inc; }
  explicit Pcg64(const SeedSeq& ss) {
    uint32_t w[8];
    ss.generate(w, 8);
    uint64_t v[4];
    for (int i = 0; i < 4; ++i) v[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
    const unsigned __int128 initstate = ((unsigned __int128)v[0] << 64) | v[1];
    const unsigned __int128 initseq = ((unsigned __int128)v[2] << 64) | v[3];
    state = 0;
    inc = (initseq << 1) | 1;
    step();
    state += initstate;
    step();
  }
  uint64_t next64() {
    step();
    const uint64_t hi = (uint64_t)(state >> 64), lo = (uint64_t)state;
    const uint64_t x = hi ^ lo;
    const unsigned rot = (unsigned)(hi >> 58);
    return (x >> rot) | (x << ((64 - rot) & 63));
  }
  double next_double() { return (double)(next64() >> 11) * (1.0 / 9007199254740992.0); }
};

constexpr uint64_t TAG_TRAJ = 0x5452414AULL, TAG_SAMPLE = 0x53414D50ULL;

}  // namespace

void rng_uniforms(int has_seed, uint64_t seed, uint64_t traj, int64_t timestep, int n, double* out) {
  if (!has_seed) {  // unseeded generator (random_utils.py:33-35): fresh OS entropy
    std::random_device rd;
    std::vector<uint32_t> e = {rd(), rd(), rd(), rd()};
    Pcg64 g{SeedSeq(e)};
    for (int i = 0; i < n; ++i) out[i] = g.next_double();
    return;
  }
  std::vector<uint32_t> e;
  push_entropy(e, seed);
  push_entropy(e, traj);
  if (timestep >= 0) { push_entropy(e, (uint64_t)timestep); push_entropy(e, TAG_SAMPLE); }
  else push_entropy(e, TAG_TRAJ);
  Pcg64 g{SeedSeq(e)};
  for (int i = 0; i < n; ++i) out[i] = g.next_double();
}

namespace {

struct RunCtx {
  Engine* e;
  const tjm_run_config* c;
  const std::vector<char>* dead = nullptr;  // trajectories taken out of the run (their rows are NaN)
  int T, cols;
  double* results;      // [B][n_obs][cols]
  double* diagnostics;  // [B][3][cols]
  std::vector<zc> M, M2;  // complex128 (host side of site_moments)
  std::vector<int> chi;
  bool need2 = false;
};

int measure(RunCtx& r, int set, int col) {
  Engine& e = *r.e;
  const int B = e.B, L = e.L, d = e.d, dd = d * d;
  int rc = e.site_moments(set, reinterpret_cast<double*>(r.M.data()), r.need2 ? reinterpret_cast<double*>(r.M2.data()) : nullptr);
  if (rc != TJM_OK) return rc;
  for (int k = 0; k < r.c->n_obs; ++k) {
    const int site = r.c->obs_site[k];
    const zc* O = reinterpret_cast<const zc*>(r.c->obs_matrix) + (size_t)k * dd * dd;
    const int n = (r.c->obs_nsites[k] == 2) ? dd : d;
    for (int b = 0; b < B; ++b) {
      const zc* Mb = (n == d) ? &r.M[((size_t)site * B + b) * dd] : &r.M2[((size_t)site * B + b) * dd * dd];
      double re = 0.0, im = 0.0;
      for (int p = 0; p < n; ++p)
        for (int q = 0; q < n; ++q) {
          const zc o = O[p * n + q], m = Mb[p * n + q];
          re += o.x * m.x - o.y * m.y;
          im += o.x * m.y + o.y * m.x;
        }
      if (r.dead && (*r.dead)[b]) { r.results[((size_t)b * r.c->n_obs + k) * r.cols + col] = __builtin_nan(""); continue; }
      if (!(im < TJM_IMAG_TOL)) return TJM_ERR_ASSERT;  // "assert exp.imag < 1e-13" (mps.py:1233): a NaN fails it too
      r.results[((size_t)b * r.c->n_obs + k) * r.cols + col] = re;
    }
  }
  if ((rc = e.bond_dims(set, r.chi.data())) != TJM_OK) return rc;
  for (int b = 0; b < B; ++b) {  // record_diagnostics (mps.py:549-602)
    const int* ch = &r.chi[(size_t)b * (L + 1)];
    double cost = 0.0, total = 0.0;
    int mx = d;
    for (int i = 1; i < L; ++i) { cost += (double)ch[i] * ch[i] * ch[i]; total += ch[i]; }
    for (int i = 1; i <= L; ++i) mx = ch[i] > mx ? ch[i] : mx;
    double* dg = r.diagnostics + (size_t)b * 3 * r.cols;
    dg[0 * r.cols + col] = cost;
    dg[1 * r.cols + col] = mx;
    dg[2 * r.cols + col] = total;
  }
  return TJM_OK;
}

}  // namespace

// status == nullptr: one return code for the batch (the first failure ends the call).  With a status array a trajectory whose state
// holds a non-finite number is TAKEN OUT instead - status[b] = TJM_ERR_NUMERIC, its rows NaN, its slot refilled with a copy of a healthy
// neighbour so that every kernel keeps seeing finite data - and the other trajectories finish exactly as they would have without it
// (a trajectory is a pure function of its own slot): the reference loses one job of its pool, not the pool
// (core/parallel_utils.py:361-383).  The states are screened before the first step and after every step.
int run_batch(Engine& e, const tjm_run_config* c, const int64_t* traj, double* results, double* diagnostics, int32_t* status) {
  if (!c || !traj || !results || !diagnostics) return TJM_ERR_ARG;
  if (c->n_times < 1 || (c->order != 1 && c->order != 2) || c->n_obs < 0) return TJM_ERR_ARG;
  const int B = e.B, L = e.L, d = e.d;
  RunCtx r;
  r.e = &e; r.c = c; r.T = c->n_times; r.cols = c->sample_timesteps ? c->n_times : 1;
  r.results = results; r.diagnostics = diagnostics;
  for (int k = 0; k < c->n_obs; ++k) {
    if (c->obs_nsites[k] != 1 && c->obs_nsites[k] != 2) return TJM_ERR_ARG;
    if (c->obs_site[k] < 0 || c->obs_site[k] + c->obs_nsites[k] > L) return TJM_ERR_ARG;
    if (c->obs_nsites[k] == 2) r.need2 = true;
  }
  r.M.resize((size_t)L * B * d * d);
  if (r.need2) r.M2.resize((size_t)(L > 1 ? L - 1 : 1) * B * d * d * d * d);
  r.chi.resize((size_t)B * (L + 1));
  const int n_t = c->n_times;
  const double dt = e.dt;
  const bool noisy = c->has_noise != 0;
  // trajectory streams: at most two draws per stochastic_process call (jump test, channel choice)
  const int n_draw = 2 * n_t + 2;
  std::vector<double> u((size_t)B * n_draw);
  for (int b = 0; b < B; ++b) rng_uniforms(c->has_seed, c->seed, (uint64_t)traj[b], -1, n_draw, &u[(size_t)b * n_draw]);
  std::vector<int> pos(B, 0), jumped(B, 0), pos_snap(B, 0);
  std::vector<double> cand((size_t)B * 2);
  int rc, clipped = 0;
  std::vector<char> dead(B, 0);
  std::vector<int> nonfinite(B, 0);
  // status is in / out: a continued run (start_step > 0, e.g. after a capacity rollback onto a larger engine) keeps the trajectories
  // that were taken out earlier out - their slots hold a finite copy of a donor that would pass the screen - and a fresh run starts clean
  if (status) {
    const bool continued = c->start_step > 0;
    for (int b = 0; b < B; ++b) {
      if (continued && status[b] != TJM_OK) dead[b] = 1;
      else status[b] = TJM_OK;
    }
    r.dead = &dead;
  }
  auto screen = [&](int set) -> int {
    if (!status) return TJM_OK;
    if ((rc = e.finite_check(set, nonfinite.data())) != TJM_OK) return rc;
    int donor = -1;
    for (int b = 0; b < B && donor < 0; ++b) if (!nonfinite[b] && !dead[b]) donor = b;
    for (int b = 0; b < B; ++b) {
      if (!nonfinite[b]) continue;
      if (!dead[b]) { dead[b] = 1; status[b] = TJM_ERR_NUMERIC; }
      if (donor < 0) return TJM_ERR_NUMERIC;  // nobody left to run
      if ((rc = e.copy_slot(set, b, donor)) != TJM_OK) return rc;
    }
    return TJM_OK;
  };
  if ((rc = screen(0)) != TJM_OK) return rc;
  const int j0 = c->start_step;
  if (j0 < 0 || j0 >= n_t || (j0 > 0 && !c->rng_pos) || (c->start_phase != 0 && (c->order != 2 || j0 < 2))) return TJM_ERR_ARG;
  if (j0 > 0)
    for (int b = 0; b < B; ++b) {
      if (c->rng_pos[b] < 0 || c->rng_pos[b] + 2 > n_draw) return TJM_ERR_ARG;
      pos[b] = (int)c->rng_pos[b];
    }
  if ((rc = e.capacity_overflow(&clipped, true)) != TJM_OK) return rc;  // start from a clean flag
  // A truncation clipped by the engine's storage makes the rest of the run pointless.  Time step j is then rolled back (set 1
  // holds the states of its start, nothing else uses that set in between) and the caller continues on a larger engine.
  auto stop_at = [&](int step, int phase) -> int {
    if (c->resume) { c->resume[0] = step; c->resume[1] = phase; }
    if (c->rng_pos) for (int b = 0; b < B; ++b) c->rng_pos[b] = pos[b];
    return TJM_ERR_CAPACITY;
  };
  auto clipped_now = [&](bool& yes) -> int {
    if ((rc = e.capacity_overflow(&clipped, false)) != TJM_OK) return rc;
    yes = clipped != 0;
    return TJM_OK;
  };
  auto snapshot = [&]() -> int { pos_snap = pos; return e.copy_state(1, 0); };
  auto roll_back = [&]() -> int { pos = pos_snap; return e.copy_state(0, 1); };
  auto record = [&](int j) { return c->sample_timesteps ? true : j == n_t - 1; };
  auto col_of = [&](int j) { return c->sample_timesteps ? j : 0; };
  auto stochastic_main = [&](int set) -> int {
    if (!noisy) {
      std::fill(cand.begin(), cand.end(), 0.0);
      if ((rc = e.set_uniforms(cand.data(), 2)) != TJM_OK) return rc;
      return e.stochastic(set, dt, nullptr, nullptr);
    }
    for (int b = 0; b < B; ++b) {
      cand[2 * b] = u[(size_t)b * n_draw + pos[b]];
      cand[2 * b + 1] = u[(size_t)b * n_draw + pos[b] + 1];
    }
    if ((rc = e.set_uniforms(cand.data(), 2)) != TJM_OK) return rc;
    if ((rc = e.stochastic(set, dt, jumped.data(), nullptr)) != TJM_OK) return rc;
    for (int b = 0; b < B; ++b) pos[b] += 1 + jumped[b];
    return TJM_OK;
  };
  bool over = false;

  if (c->order == 1) {  // analog_tjm_1 (analog_tjm.py:369-462)
    if (j0 == 0 && (c->sample_timesteps || n_t <= 1))
      if ((rc = measure(r, 0, 0)) != TJM_OK) return rc;
    for (int j = (j0 > 0 ? j0 : 1); j < n_t; ++j) {
      if ((rc = snapshot()) != TJM_OK) return rc;
      if ((rc = e.tdvp(0)) != TJM_OK) return rc;
      if (noisy) {
        if ((rc = e.dissipate(0, dt)) != TJM_OK) return rc;
        if ((rc = stochastic_main(0)) != TJM_OK) return rc;
      }
      if ((rc = clipped_now(over)) != TJM_OK) return rc;
      if (over) {
        if ((rc = roll_back()) != TJM_OK) return rc;
        return stop_at(j, 0);
      }
      if ((rc = screen(0)) != TJM_OK) return rc;
      if (record(j))
        if ((rc = measure(r, 0, col_of(j))) != TJM_OK) return rc;
    }
    return TJM_OK;
  }
  // analog_tjm_2 (analog_tjm.py:206-366), standalone form
  auto sample = [&](int j) -> int {
    if (!record(j)) return TJM_OK;
    if ((rc = e.copy_state(1, 0)) != TJM_OK) return rc;  // psi = deepcopy(phi)
    if ((rc = e.tdvp(1)) != TJM_OK) return rc;
    // dissipation and the stochastic step run with or without a noise model (analog_tjm.py:86-107, 179-203): without one
    // they reduce to the truncating gauge sweep and the renormalisation, and no random number is consumed
    if ((rc = e.dissipate(1, 0.5 * dt)) != TJM_OK) return rc;
    if (noisy) for (int b = 0; b < B; ++b) rng_uniforms(c->has_seed, c->seed, (uint64_t)traj[b], j, 2, &cand[2 * b]);
    else std::fill(cand.begin(), cand.end(), 0.0);
    if ((rc = e.set_uniforms(cand.data(), 2)) != TJM_OK) return rc;
    if ((rc = e.stochastic(1, dt, nullptr, nullptr)) != TJM_OK) return rc;
    if ((rc = clipped_now(over)) != TJM_OK) return rc;
    if (over) return TJM_OK;  // phi is untouched: the caller stops with phase 1
    if ((rc = screen(1)) != TJM_OK) return rc;  // a non-finite sampling copy takes its trajectory out, not the batch (measure would assert)
    return measure(r, 1, col_of(j));
  };
  if (j0 == 0) {
    if (record(0))
      if ((rc = measure(r, 0, 0)) != TJM_OK) return rc;
    if (n_t == 1) return TJM_OK;
    if ((rc = e.dissipate(0, 0.5 * dt)) != TJM_OK) return rc;
    if ((rc = stochastic_main(0)) != TJM_OK) return rc;
    if ((rc = clipped_now(over)) != TJM_OK) return rc;
    if (!over && (rc = screen(0)) != TJM_OK) return rc;  // the half-step prelude is screened like every full step
    if (!over && (rc = sample(1)) != TJM_OK) return rc;
    if (over) return stop_at(0, 0);  // before the first full step: nothing to keep
  }
  for (int j = (j0 > 1 ? j0 : 2); j < n_t; ++j) {
    if (!(j == j0 && c->start_phase == 1)) {
      if ((rc = snapshot()) != TJM_OK) return rc;
      if ((rc = e.tdvp(0)) != TJM_OK) return rc;
      if ((rc = e.dissipate(0, dt)) != TJM_OK) return rc;
      if ((rc = stochastic_main(0)) != TJM_OK) return rc;
      if ((rc = clipped_now(over)) != TJM_OK) return rc;
      if (over) {
        if ((rc = roll_back()) != TJM_OK) return rc;
        return stop_at(j, 0);
      }
      if ((rc = screen(0)) != TJM_OK) return rc;
    }
    if ((rc = sample(j)) != TJM_OK) return rc;
    if (over) return stop_at(j, 1);
  }
  return TJM_OK;
}

}  // namespace tjm
