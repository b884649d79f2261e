// Batched TJM engine: B trajectories advance the same site in lock-step on one GPU.
// Host-side control (this class) issues the kernels of tjm_gemm / tjm_kernels / tjm_svd on one
// HIP stream; all tensors live in a caller-provided device workspace.
#pragma once
#include <functional>
#include <map>
#include <tuple>
#include <vector>

#include "../../include/tjm_hip.h"
#include "tjm_kernels.h"

namespace tjm {

constexpr int MAXD = 4;               // local dimension 2, 3 or 4 (uniform along the chain)
constexpr int MAXDD = MAXD * MAXD;    // entries of a one-site operator
constexpr int MSLOT = MAXDD * MAXDD;  // entries of an operator on a merged pair

struct NoiseProc {
  int nsites;          // 1 or 2
  int site0, site1;
  double gamma;
  int pauli;           // unit-phase Pauli (L^dag L = 1)
  cplx mat[MSLOT];     // 1-site: d x d ; adjacent 2-site: d^2 x d^2
  cplx f0[MAXDD], f1[MAXDD];  // long-range factors (d x d each)
  int has_factors;
};

// One set of MPS tensors for the whole batch (the trajectory state phi, or the measurement copy psi).
struct StateSet {
  std::vector<cplx*> A;  // per site: [B][d][cap[i]][cap[i+1]]
  int* chi = nullptr;    // device [B][L+1] actual bond dimensions
};

class Engine {
 public:
  int L = 0, d = 2, chi_max = 0, B = 0, mmax = 25;
  std::vector<int> cap, Dm;
  int Dmax = 1;
  hipStream_t stream = nullptr;
  int device_id = -1;  // device that owns the workspace (set at bind from the workspace pointer)
  // numerical parameters
  double dt = 0.1, svd_threshold = 1e-6, krylov_tol = 1e-4;
  int trunc_mode = 0, max_bond = 0, tdvp_mode = 2, tdvp_sweeps = 1;
  // statistics
  long stat_matvecs = 0, stat_krylov_calls = 0, stat_svds = 0, stat_svd_sweeps = 0, stat_site_updates = 0;
  long stat_direct_applies = 0;  // H_eff applies served by the direct form (no T2)
  long stat_matvecs2 = 0, stat_env_updates = 0, stat_svd_mats = 0;  // two-site H_eff applies (subset of matvecs), environment updates, matrices factorised

  // Live timing of the kernel classes of a step with HIP events on the engine's stream (bench.py's roofline object):
  // class 0 = SVD family (two-site splits and SVD centre shifts: QR + Jacobi + finish + their GEMMs), 1 = Krylov exponentials
  // (H_eff applies, MPO stage, Lanczos vector kernels), 2 = environment updates and merges.
  enum { PROF_SVD = 0, PROF_KRYLOV = 1, PROF_ENV = 2, PROF_NCLASS = 3 };
  void profile_enable(bool on);
  int profile_read(double* ms /*[PROF_NCLASS]*/, long* regions /*[PROF_NCLASS]*/);
  struct Region {
    Engine& e; int idx;
    Region(Engine& eng, int cls);
    ~Region();
  };

  ~Engine();  // pinned host words and HIP events (the device workspace belongs to the caller)
  int create(int L, int d, int chi_max, int B, const int* mpo_bond, int cap_slack = 1);
  size_t workspace_bytes() const;
  int load_state_slot(int set, int b, const double* host, const int* bonds);
  int copy_slot(int set, int dst, int src);
  int finite_check(int set, int* host_flags);
  int bind(void* ws, size_t bytes, hipStream_t s);
  int set_mpo(const double* host_tensors);   // packed (o,p,l,r) complex128 per site
  int set_noise(const std::vector<NoiseProc>& procs);
  int load_state(int set, const double* host_tensors, const int* host_bonds);  // broadcast one MPS to all slots
  int copy_state(int dst, int src);
  int export_state(int set, int b, double* host_out, int* host_bonds);         // padded tensors of slot b
  int set_uniforms(const double* host_u, int n_per_traj);
  int reset_cursor();
  int capacity_overflow(int* host_flag, bool clear);
  int adopt(Engine& src, int first);   // set 0 <- set 0 of src slots [first, first + B), re-padded

  int tdvp(int set);
  int dissipate(int set, double dt_, int start_center = 0);
  int set_noise_filter(int n, const int* idx);   // n < 0: all processes active
  int normalize_qr(int set, int center);
  int apply_single(int set, int site, const double* host_mat);
  int tebd_gate(int set, int left, const double* host_u, int center = 0);
  int apply_pair(int set, int left, const double* host_u, int min_keep);
  int apply_gate_mpo(int set, int first, int last, int r, const double* host_left, const double* host_right);
  int canonicalize_qr(int set, int center);
  int stochastic(int set, double dt_, int* host_jumped /*B or null*/, double* host_dp /*B or null*/);
  int site_moments(int set, double* host_M /*[L][B][d][d] complex*/, double* host_M2 = nullptr /*[L-1][B][d^2][d^2] or null*/);
  int bond_dims(int set, int* host_chi /*[B][L+1]*/);
  int site_normsq0(int set, double* host_out);
  int bond_spectrum(int set, int i, double* host_spec /*[B][n_out]*/, int n_out);
  int bitstring_probability(int set, const unsigned char* bits /*[L]*/, double* host_prob /*[B]*/);
  int sample_shots(int set, int shots, const double* host_rot, const double* host_u /*[B][shots][L]*/, unsigned char* host_bits /*[B][shots][L]*/);

  // exposed for kernel-level parity tests
  int krylov_site(cplx* x_in_v0, int P, int ca, int cb, const cplx* Lenv, long l_b0, int Dl, const cplx* Renv, long r_b0,
                  int Dr, const cplx* Wm, double dt_, const int* nloc_dev, cplx* out, long out_b0, int n0, int n1, int n2,
                  int n3, long o0, long o1, long o2, int nb0, const int* ids, const int* chi_l = nullptr, const int* chi_r = nullptr);
  // lch / rch: channel of Lenv / Renv certified to be the identity matrix (first or last; -1: none) - its share of the GEMM is a copy
  int heff_apply(const cplx* x, long x_b0, int P, int ca, int cb, const cplx* Lenv, long l_b0, int Dl, const cplx* Renv,
                 long r_b0, int Dr, const cplx* Wm, cplx* y, long y_b0, int nb0, const int* ids, const int* active, int lch = -1, int rch = -1);
  int identity_channels(int ca, int cb, const cplx* Lenv, long l_b0, int Dl, const cplx* Renv, long r_b0, int Dr, int nb0, const int* ids,
                        const int* chi_l, const int* chi_r, int* lch, int* rch);
  long stat_cert_traj = 0, stat_cert_jumps = 0;  // trajectory-steps whose scalar dissipation sweep was certified away / jumps applied in place
  long stat_cert_blocked = 0;                    // trajectory-bonds whose certificate test ran on chol_pd_blocked_kernel (bonds above 128)
  long stat_ident_calls = 0, stat_ident_hits = 0;  // Krylov calls examined / channels certified (of two per call)

  struct Prof {
    bool on = false;
    std::vector<hipEvent_t> pool;       // pairs (begin, end)
    std::vector<int> cls;               // class of pair k
    size_t used = 0;
    double ms[PROF_NCLASS] = {0, 0, 0};
    long n[PROF_NCLASS] = {0, 0, 0};
    int depth = 0;
  } prof_;
  void prof_collect();

  // environment updates on explicit tensors (also the kernel-level parity exports of the C ABI)
  int env_left_at(const cplx* A, long a_b0, int ca, int cb, int Dl, int Dr, const cplx* Lin, long lin_b0, const cplx* WenvL, cplx* Lout,
                  long lout_b0, int nb, const int* ids = nullptr);
  int env_right_at(const cplx* A, long a_b0, int ca, int cb, int Dl, int Dr, const cplx* Rin, long rin_b0, const cplx* Wm, cplx* Rout,
                   long rout_b0, int nb, const int* ids = nullptr);
  // site-level steps of a sweep driven from the host (dynamic TDVP, integrators.py:294-511): host index lists of trajectories
  int step_env_init(int set);
  int step_two_site(int set, int i, double dt_, int dist, int capped, const int* host_ids, int n);
  int step_one_site(int set, int i, double dt_, const int* host_ids, int n);
  int step_env(int set, int i, int left, const int* host_ids, int n);
  int step_qr_bond(int set, int i, int right, double dt_, int max_bond, const int* host_ids, int n);
  int step_cap_bond(int set, int bond, int target, const int* host_ids, int n);
  // steps of the BUG integrator (core/methods/bug.py:35-257), whole batch; centres live in set 2, scratch in set 3
  int step_bug_prepare(int set);
  int step_bug_site(int set, int site, double dt_);
  int step_bug_root(int set, double dt_);
  int step_flip(int set);
  int step_compress(int set, double threshold, int max_bond_dim, int mode);
  int bond_column(int set, int bond, std::vector<int>& out);  // chi[b][bond] of every trajectory (host)
  int sweep_dynamic(int set, int max_bond, double dt_);      // one sweep of the dynamic TDVP, branch lists formed per site from one bond column
  int bug_sweep(int set, double dt_);                         // one half-sweep of the BUG integrator
  int copy_site(int dst, int src, int site);
  int copy_chi_col(int dst, int src, int col);
  int upload_ids(const int* host_ids, int n, const int** dev);
  // kernel-level parity exports (tjm_capi.hip): operands are device arrays of nb <= B slots, W is a host MPO tensor (o,p,l,r)
  int x_heff_apply(int nsites, int ca, int cb, int Dl, int Dr, const cplx* x, const cplx* Lenv, const cplx* Renv, const double* host_w, cplx* y, int nb);
  int x_env_update(int left, int ca, int cb, int Dl, int Dr, const cplx* A, const cplx* env, const double* host_w, cplx* out, int nb);
  int x_project_bond(int cu, int cv, int D, const cplx* C, const cplx* Lenv, const cplx* Renv, cplx* y, int nb);
  int x_lanczos_expm(int nsites, int ca, int cb, int Dl, int Dr, const cplx* x, const cplx* Lenv, const cplx* Renv, const double* host_w, double dt_,
                     double tol, cplx* y, int nb, long* matvecs);
  int x_center_shift(int set, int site, int direction, int use_svd);
  int x_jump_weights(int set, double dt_, int* host_order /*[nproc]*/, double* host_w /*[B][nproc]*/, int* n_out);
  int upload_w(const double* host_w, int P, int Dl, int Dr, cplx** mv, cplx** envl);
  int jump_weights(int set, double dt_, const std::vector<double>& nsq, const std::vector<int>& which, std::vector<int>& order,
                   std::vector<double>& w);

  StateSet sets[4];   // 0: trajectory state phi, 1: measurement copy psi; 2, 3: centres and scratch of the BUG steps (engines created with cap_slack > 1)
  int n_sets = 2;
  // work areas (public for tests)
  cplx *T1 = nullptr, *T2 = nullptr, *V = nullptr, *theta = nullptr;
  long v_b0 = 0, v_ld = 0, t_b0 = 0, theta_b0 = 0;
  SvdWorkspace svdw{};
  QrWorkspace qrw{};
  MixedWorkspace mixw{};  // complex64 phase of the mixed-precision two-site split (fp64 build, square splits of 128 ... 512 rows)
  KrylovState ks{};

 private:
  bool bound_ = false;
  int krylov_P_ = 0;             // physical dimension of the block the running Krylov exponential acts on (0: bond matrix)
  std::vector<long> a_b0_, l_b0_, r_b0_;
  std::vector<cplx*> Lenv_, Renv_;
  std::vector<cplx*> W_;         // per-site MPO as matrix [(o,l),(p,r)] : matvec / right-env form
  std::vector<cplx*> WenvL_;     // [(p,r),(o,l)] : left-env form
  std::vector<cplx*> W2_;        // merged two-site [(o o', l),(p p', r)]
  std::vector<std::vector<cplx>> Whost_;
  // Direct form of an H_eff apply (heff_apply): with certified identity channels lch / rch, an MPO matrix none of whose entries
  // couples a non-identity left channel to a non-identity right channel and whose rows (o, l != lch) hold at most one entry (Pauli-sum
  // Hamiltonians with nearest-neighbour terms) gives T2[o][.][l][.] = coef x[perm]: the third stage reads x itself.
  struct DirectForm { bool ok = false; int* perm = nullptr; cplx* coef = nullptr; };
  std::map<std::tuple<const cplx*, int, int, int, int, int>, DirectForm> direct_;  // (Wm, P, Dl, Dr, lch, rch); cleared whenever an MPO matrix is uploaded
  const DirectForm* direct_form(const cplx* Wm, int P, int Dl, int Dr, int lch, int rch);
  void direct_clear();
  real *part1_ = nullptr, *part2_ = nullptr;
  int* nloc_ = nullptr;
  real* scal_ = nullptr;         // [B] scratch scalars
  real* normsq_ = nullptr;       // [B]
  int* ids_ = nullptr;           // [B] compacted trajectory list
  int* opidx_ = nullptr;         // [B]
  int* jsite_ = nullptr;         // [B]
  int* overflow_ = nullptr;      // sticky flag: a truncation was clipped by the storage capacity
  hipEvent_t krylov_ev_[2] = {nullptr, nullptr};  // pipelined convergence check of the Lanczos loop (krylov_core)
  SmallSiteRef* site_refs_[4] = {nullptr, nullptr, nullptr, nullptr};   // device tables of the fused small-bond sweeps
  SmallSweepStep* sweep_steps_ = nullptr;
  bool sweep_ok_ = false;
  int run_sweep(int set, const std::vector<SmallSweepStep>& steps, const int* ids, int nb0);
  int qr_walk(int set, int from, int to);
  cplx* ops_ = nullptr;          // operator table (device)
  cplx* Wx_[2] = {nullptr, nullptr};  // MPO matrices of the kernel-level exports
  cplx* E_ = nullptr;            // [B][chi][chi] moment environment (x2 ping-pong)
  cplx* E2_ = nullptr;
  cplx* M_ = nullptr;            // [L][B][d][d]
  cplx* Cm_ = nullptr;           // [B][cap][cap] bond matrix of the one-site sweep
  cplx* M2_ = nullptr;           // [L-1][B][d^2][d^2] two-site moments
  int n_uniform_ = 0;
  std::vector<int> cursor_;      // host-side cursor per trajectory
  std::vector<double> uni_host_;
  int* h_pinned_ = nullptr;
  std::vector<NoiseProc> noise_;
  std::vector<std::vector<int>> one_by_site_, two_by_right_;
  size_t ws_bytes_ = 0;

  int gemm(const GemmDesc& g) { return launch_gemm(g, stream); }
  int merge_tensor_layout(StateSet& S, int i, cplx* out, long out_b0, const int* ids, int nb0);
  int merge_matrix_layout(StateSet& S, int i, const int* ids, int nb0);
  int env_left(StateSet& S, int i, const int* ids = nullptr, int nb0 = -1);    // Lenv[i+1] from Lenv[i], A_i
  int env_right(StateSet& S, int i, const int* ids = nullptr, int nb0 = -1);   // Renv[i-1] from Renv[i], A_i
  int two_site_update(StateSet& S, int i, double dt_, int dist, const int* ids = nullptr, int nb0 = -1, bool capped = true);
  int one_site_update(StateSet& S, int i, double dt_, const int* ids = nullptr, int nb0 = -1);
  int sweep_2site(StateSet& S, double scale);
  int split(StateSet& S, int i, int dist, int mode, double thr, int maxb, int min_keep, const int* ids, int nb0);
  int set_nloc(StateSet& S, int bl, int br, int P);
  using ApplyFn = std::function<int(const cplx* x, cplx* y, const int* active)>;
  // krylov_core asks the apply it is about to call for Re <x, H x> as well (x = the vector it passes): heff_apply serves it in the
  // epilogue of its last GEMM when that runs on the tiled kernel (per-tile partial sums in part1_), otherwise the dot-product kernel runs
  struct DotRequest { const cplx* v = nullptr; bool served = false; int nblk1 = 0; } dot_req_;
  int krylov_core(const ApplyFn& apply, int n, double dt_, const int* nloc_dev, cplx* out, long out_b0, int n0, int n1, int n2, int n3,
                  long o0, long o1, long o2, int nb0, const int* ids);
  int bond_apply(const cplx* x, int cu, int cv, const cplx* Lenv, long l_b0, const cplx* Renv, long r_b0, int D, cplx* y, const int* active,
                 int nb0 = -1, const int* ids = nullptr);
  int sweep_1site(StateSet& S, double scale);
  int qr_site(StateSet& S, int i, bool right, const int* ids = nullptr, int nb0 = -1, bool absorb = false);  // absorb: also multiply C into the neighbour (small bonds only)
    // A_i = Q C (right) or A_i = C^T Q (left); C into Cm_
  int two_site_op(StateSet& S, int i, const cplx* dev_ops, const int* op_index, const int* ids, int nb0, int min_keep);
  int qr_shift_right(StateSet& S, int i, const int* ids = nullptr, int nb0 = -1);
  int qr_shift_left(StateSet& S, int i);
  std::vector<char> proc_on_;
  int svd_shift_right(StateSet& S, int i, const int* ids, int nb0);
  int svd_shift_left(StateSet& S, int i, const int* ids, int nb0);
  int svd_shift_left_2site(StateSet& S, int i, const int* ids, int nb0);
  // Certified scalar dissipation (tjm_engine.hip: dissipate): the right-going SVD pass on scratch copies of the centre tensor
  int svd_shift_right_virtual(StateSet& S, int i, const cplx* Cin, long cin_b0, cplx* Cout, long cout_b0, const int* ids, int nb0);
  bool cert_gram_fits() const;
  int cert_pass_gram(StateSet& S, const int* ids, int nb0, double cut);
  int state_checksum(int set, const int* ids, int n, unsigned long long* host_out);
  int* vchi_ = nullptr;                  // [B][L+1] bond dimensions the virtual pass would leave
  real* cert_min_ = nullptr;             // [B] smallest squared singular value met by the virtual pass
  int* cert_flag_ = nullptr;             // [B] the virtual pass would have truncated a bond
  unsigned long long* csum_ = nullptr;   // [B] checksum scratch
  std::vector<char> cert_ok_;            // per trajectory: the state is what a certified dissipation left (valid while cert_set_ >= 0)
  std::vector<unsigned long long> cert_sum_;
  int cert_set_ = -1;
  std::vector<int> cert_wait_;           // [set][trajectory]: calls of dissipate it still sits out after its certificate failed
  std::vector<int> cert_back_;           // [set][trajectory]: length of its last sit-out (doubles with consecutive failures, at most 4)
  void cert_size() { if (cert_wait_.size() != (size_t)n_sets * B) { cert_wait_.assign((size_t)n_sets * B, 0); cert_back_.assign((size_t)n_sets * B, 0); } }
  int copy_back(cplx* dst, long dst_b0, const cplx* src, long src_b0, long n, const int* ids, int nb0);
  std::vector<int> unitary_jump_;
};

// tjm_run.hip
void rng_uniforms(int has_seed, uint64_t seed, uint64_t traj, int64_t timestep, int n, double* out);
int run_batch(Engine& e, const tjm_run_config* c, const int64_t* traj, double* results, double* diagnostics, int32_t* status = nullptr);

}  // namespace tjm
