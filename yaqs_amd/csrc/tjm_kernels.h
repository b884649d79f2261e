// Launch wrappers for the helper kernels (tjm_kernels.hip) and the Jacobi SVD (tjm_svd.hip).
#pragma once
#include "tjm_common.h"

namespace tjm {

constexpr int TJM_MAX_PART = 64;  // partial sums per trajectory in two-stage reductions

struct MpoApplyDesc {
  const cplx* in;
  cplx* out;
  const cplx* Wm;  // [(po, bo)][(pi, bi)] row-major, (P*dout) x (P*din)
  int P, din, dout;
  int na, nB;
  long in_sp, in_sb, in_sa;
  long out_sp, out_sb, out_sa;
  long in_b0, out_b0;
  int nb0;
  const int* ids;
  const int* active;
  // identity channels of heff_apply: input channel in_alt_ch is read from in_alt ([b][pi][a][B] with its own strides) instead of
  // `in`, output channel out_alt_ch is written to out_alt instead of `out` (-1: none)
  const cplx* in_alt = nullptr;
  cplx* out_alt = nullptr;
  int in_alt_ch = -1, out_alt_ch = -1;
  long in_alt_b0 = 0, in_alt_sp = 0, in_alt_sa = 0, out_alt_b0 = 0, out_alt_sp = 0, out_alt_sa = 0;
};
int launch_mpo_apply(const MpoApplyDesc& d, hipStream_t stream);

// Per-trajectory Krylov bookkeeping (device arrays, indexed by trajectory slot).
struct KrylovState {
  real* alpha;   // [B][mmax]
  real* beta;    // [B][mmax]
  cplx* coef;      // [B][mmax]
  real* vnorm;   // [B]
  real* scale;   // [B]  1/beta_j (or 1/|v|)
  real* svec;    // [B][mmax] scale of Krylov vector j: the basis is stored UNNORMALISED (v_j = svec[j] * V[j]; svec[0] = 1 / |v|,
                 // svec[j + 1] = 1 / beta_j), the scalars ride along in the vector kernels - no normalisation pass per iteration
  int* status;     // [B]  1 = still iterating, 0 = finished
  int* kfinal;     // [B]
  int* n_active;   // [1]
  int mmax;
};

// expm_krylov (matrix_exponential.py:60-163) of one site / two-site block with small bonds, everything in one kernel
// (krylov_site_small_kernel): V[0] holds the input, the Krylov vectors are written behind it, the result goes to `out` through the
// 4-level index permutation of krylov_combine_kernel.
struct SmallKrylovDesc {
  cplx* V; long v_b0, v_ld;
  int P, ca, cb;              // physical extent (d or d^2) and padded bond extents of the block
  const cplx* Lenv; long l_b0; int Dl;   // L[(a,l)][A]
  const cplx* Renv; long r_b0; int Dr;   // R[b][(r,B)]
  const cplx* Wm;             // [(o,l)][(p,r)] row-major
  real dt, tol;
  const int* nloc;            // actual local dimension per trajectory (breakdown threshold)
  int mmax;
  cplx* out; long out_b0;
  int n1, n2, n3; long o0, o1, o2;
  const int* ids; int nb0;
  unsigned long long* matvecs;   // device counter (statistics), may be null
  const int* chi_l; const int* chi_r; int chi_stride;   // actual bonds of the block per trajectory (null: the padded extents)
};
bool krylov_small_fits(int P, int ca, int cb, int Dl, int Dr, int mmax, int nb0);
int launch_krylov_site_small(const SmallKrylovDesc& p, hipStream_t s);

int launch_normsq_partial(const cplx* x, long x_b0, int n, real* part, int nb0, const int* ids, const int* active,
                          hipStream_t s, int* nblk_out);
int launch_dot_partial(const cplx* v, const cplx* w, long v_b0, long w_b0, int n, real* part, int nb0, const int* ids,
                       const int* active, hipStream_t s, int* nblk_out);
int launch_lanczos_axpy(cplx* w, const cplx* vj, const cplx* vjm1, long v_b0, int n, const real* part1, real* part2,
                        int nblk, const real* beta, int beta_ld, int j, int nb0, const int* ids, const int* active,
                        hipStream_t s, const real* svec, int nblk1);
int launch_scale(cplx* x, long x_b0, long n, const real* scale, int nb0, const int* ids, const int* active, hipStream_t s);
int launch_env_identity_check(const cplx* env, long b0, int c, int D, const int* chi, int chi_stride, real tol, int* flags, int nb0, const int* ids,
                              hipStream_t s);  // flags[0] / [1] raised when the first / last channel of env[c][D][c] is not the identity
int launch_lanczos_init(const KrylovState& ks, const real* part, int nblk, int nb0, const int* ids, hipStream_t s);
int launch_lanczos_finalize(const KrylovState& ks, const real* part1, const real* part2, int nblk, int j, real dt,
                            real tol, const int* nloc, int nb0, const int* ids, hipStream_t s, int nblk1);
int launch_krylov_combine(const cplx* V, long v_b0, long v_ld, const KrylovState& ks, cplx* out, long out_b0, int n0, int n1,
                          int n2, int n3, long o0, long o1, long o2, int nb0, const int* ids, hipStream_t s);
int launch_tridiag_expm_test(const real* alpha, const real* beta, int k, real dt, real* out, hipStream_t s);
int launch_normsq(const cplx* x, long x_b0, long n, real* out, int nb0, const int* ids, hipStream_t s);
int launch_apply_local(cplx* x, long x_b0, int d, long rest, const cplx* ops, const int* op_index, int nb0, const int* ids,
                       hipStream_t s);
int launch_identity_env(cplx* E, long e_b0, int n, int D, int nb0, hipStream_t s);
int launch_phys_overlap(const cplx* x, const cplx* y, long x_b0, long y_b0, int d, long rest, cplx* M, int nb0, const int* ids,
                        hipStream_t s);

// ---- batched one-sided block-Jacobi SVD (tjm_svd.hip) ---------------------------------------
// Splits theta[b] (m x n, row-major, row pitch ld_theta) = U S V^H, truncates per the
// reference's rule and writes the two site tensors.
struct SvdSplitDesc {
  const cplx* theta;   // [B] m x n matrices, rows (s, a), cols (t, c)
  long theta_b0;
  int ld_theta;
  int m, n;            // padded matrix extents (m = d*capL, n = d*capR)
  int d;               // physical dimension (rows = d x capL, cols = d x capR)
  int capL, capR;      // padded left / right bond extents
  int capM;            // padded middle bond extent of the output tensors
  cplx* left;          // [B][d][capL][capM]
  cplx* right;         // [B][d][capM][capR]
  long left_b0, right_b0;
  int distribution;    // 0 = "right" (left isometric), 1 = "left" (right isometric), 2 = "sqrt" (plain split only)
  int trunc_mode;      // 0 discarded_weight, 1 relative, 2 hard_cutoff, 3 relative_discarded_weight
  real threshold;
  int max_bond;        // <= 0: none
  int min_keep;
  int* overflow = nullptr;  // device flag: set when the truncation rule wanted more than capM values (may be null)
  const int* chiL;     // actual left bond per trajectory, element stride chi_stride
  const int* chiR;
  int* chiM;           // out: new middle bond
  int chi_stride;
  real* spectrum;    // optional out [B][spec_ld] singular values (descending), may be null
  int spec_ld;
  int nb0;
  const int* ids;
};
struct SvdWorkspace {
  cplx* Y;        // [B][ncols_pad][rtot]  column-major stacked [X; W]
  long y_b0;
  real* norms;  // [B][ncols_pad]
  real* fro2;   // [B] squared Frobenius norm of theta (noise floor for the rotations)
  int* perm;      // [B][ncols_pad]
  real* rec;    // [B][8][256][4] rotation record (split X / W Jacobi), may be null
  int* stamps;    // [B][1152] visit-pruning stamps of the Jacobi sweeps
  int* nrot;      // [B]
  int* done;      // [B]
  int* n_active;  // [1]
  int* h_pinned;  // host pinned int for the active counter
};
// Generic one-sided Jacobi problem: X[r][c] (rx x ncols) read through two-level strided indices
//   r = r1 * r_n0 + r0 ,  c = c1 * c_n0 + c0 ,  X[r][c] = op(src[b*src_b0 + r1*s_r1 + r0*s_r0 + c1*s_c1 + c0*s_c0])
struct JacobiSource {
  const cplx* src;
  long src_b0;
  int rx, ncols;
  int r_n0, c_n0;
  long s_r1, s_r0, s_c1, s_c0;
  int conj;
  int tri;  // 1: the source is triangular, keep X[r][c] only for c <= r
  int nb0;
  const int* ids;
};
// Truncation rule (core/linalg/svd_utils.py:22-104); number of singular values = min(mulA*chiA, mulB*chiB)
struct TruncSpec {
  int trunc_mode;
  real threshold;
  int max_bond, min_keep;
  int cap = 0;               // storage extent of the new bond (0: unbounded); a wish beyond it is clipped and reported
  int* overflow = nullptr;   // device flag, set when the clip changed the result
  int* overflow_each = nullptr;  // svd_finish_kernel only: [B] words written per trajectory (0 / 1) INSTEAD of the sticky flag
  const int* chiA; int mulA;
  const int* chiB; int mulB;
  int* chiOut;
  int chi_stride;
  real* spectrum;
  int spec_ld;
};
struct JacobiShape { int ncols_pad, rx_top, rtot; };
// Centre shift of one site with small bonds, everything in one kernel (svd_shift_small_kernel): the site tensor [d][ca][cb] is
// factorised, the isometric factor replaces it and the weighted factor is multiplied into the neighbour.
struct SmallShiftDesc {
  cplx* site; long site_b0;   // A_i  [d][ca][cb]
  cplx* nb;   long nb_b0;     // right shift: A_{i+1} [d][cb][cn];  left shift: A_{i-1} [d][cn][ca]
  int d, ca, cb, cn;
  int* chi; int chi_stride;   // chi[b * stride + 0] = left bond of the site, [+1] = right bond
  real threshold; int min_keep;
  const int* ids; int nb0;
  int* flags;                 // flags[1] |= 1 when the Jacobi iteration did not converge
};
int launch_svd_shift_small(const SmallShiftDesc& p, bool left, hipStream_t s);
// Householder QR of one site with small bonds in one kernel (qr_site_small_kernel): A_i <- Q, R into the padded bond matrix
// `bond` ([cap][cap], may be null) and, when `nb` is given, multiplied into the neighbour (the QR centre shift).
struct SmallQrDesc {
  cplx* site; long site_b0;   // A_i [d][ca][cb]
  cplx* bond;                 // [B][cap][cap] with cap = cb (right) or ca (left); right: C[k][j] = R[k][j], left: C[j][k] = R[k][j]
  cplx* nb;   long nb_b0;     // right: A_{i+1} [d][cb][cn]; left: A_{i-1} [d][cn][ca]
  int d, ca, cb, cn;
  int* chi; int chi_stride;   // chi[0] = left bond of the site, chi[1] = right bond
  int* nloc;                  // [B] size of the bond problem (may be null)
  const int* ids; int nb0;
};
int launch_qr_site_small(const SmallQrDesc& p, bool right, hipStream_t s);
// A whole sweep of small-bond centre shifts in one kernel (small_sweep_kernel): per step an optional one-site factor on the
// physical index of `site` (op 1: 2x2 matrix m, op 2: real scalar) followed by a shift of the centre away from `site`
// (kind 1: SVD to the right, 2: SVD to the left, 3: QR to the right, 4: QR to the left, 0: none).
struct SmallSiteRef { cplx* A; long b0; int ca, cb; };
struct SmallSweepStep { int site, kind, op; real scal; cplx m[4]; };
struct SmallSweepDesc {
  const SmallSiteRef* sites;     // device [L]
  const SmallSweepStep* steps;   // device [nsteps]
  int nsteps, d;
  int* chi; int chi_stride;      // chi[b * stride + k] = bond k
  real threshold; int min_keep;
  const int* ids; int nb0;
  int* flags;
  int pitch;                     // rows of the LDS columns: small_sweep_pitch(largest d * cap of the chain)
};
int small_sweep_pitch(int rows);
int launch_small_sweep(const SmallSweepDesc& p, hipStream_t s);
bool svd_shift_small_fits(int d, int ca, int cb, bool left);
// out[b][k*o_k + r1*o_r1 + r0*o_r0] = scale_k * op(Ycol[perm[k]][row_off + r1*n_r0 + r0]) for k < keep, 0 for keep <= k < n_k
// scale_mode: 0 none, 1 multiply by sigma_k, 2 divide by sigma_k, 3 multiply by sqrt(sigma_k), 4 divide by sqrt(sigma_k),
//             5 divide by sigma_k and replace an exactly zero column by the unit vector of its index (completes a square basis)
struct ExtractDesc {
  cplx* out;
  long out_b0;
  int n_k; long o_k;
  int n_r1, n_r0; long o_r1, o_r0;
  int row_off;
  int conj;
  int scale_mode;
  const int* row_map = nullptr;  // optional: matrix row r goes to output row row_map[b * row_map_ld + r]
  int row_map_ld = 0;
};
long svd_y_elems(int max_dim);  // complex elements of the stacked Jacobi matrix per trajectory
size_t svd_workspace_bytes(int max_dim, int B);
size_t svd_carve(SvdWorkspace& w, char* base, int max_dim, int B);  // lays the buffers out behind base, returns the bytes used
void profile_enable(int every);
void gemm_profile_enable(int every);   // launch sampler of zgemm4_kernel (tjm_gemm.hip); get: out6, see there
void gemm_profile_get(double* out6);
void profile_get(double* total_ms, double* total_bytes, long* samples);
void jacobi_work_get(double* out4, bool reset);  // slot x rows, applied rotations x rows, sweeps, solves of the tiled Jacobi since the last reset
// Knobs of the tiled iteration for callers that do not want the defaults (the mixed-precision split, tjm_mixed.h)
struct JacobiOpts {
  int max_sweeps = 40;
  real tol2 = TJM_JACOBI_TOL2;           // squared relative off-diagonal tolerance
  real floor_scale = TJM_NOISE_FLOOR2;   // columns below sqrt(floor_scale) ||X||_F are numerically null (never rotated)
  bool allow_unconverged = false;        // reaching max_sweeps is not an error (the caller refines the result anyway)
  double stop_fraction = 0.0;            // > 0: stop after a sweep that rotated less than this fraction of all pairs (no confirming sweep)
  bool late_start = false;               // check-first tile kernel from the first sweep on
  bool late_after_first = false;         // check-first tile kernel from the second sweep on, whatever the first one rotated
  bool preloaded = false;                // Y holds X already: column-major, pitch = rows, rows and columns multiples of 64 / 32
  bool quad = false;                     // complex64 build, X-only, up to 256 rows: the tile kernel with four columns of each block per wavefront
};
// accumulate = false: rotate X only (no W rows, no rotation record); the caller rebuilds the other factor from X
int jacobi_solve(const JacobiSource& src, const TruncSpec& tr, const SvdWorkspace& w, hipStream_t s, JacobiShape* shape_out,
                 int* sweeps_out, bool accumulate = true, const JacobiOpts* opts = nullptr);
int svd_extract(const ExtractDesc& x, const SvdWorkspace& w, const JacobiShape& sh, const int* chi_keep, int chi_stride, int nb0,
                const int* ids, hipStream_t s);
int svd_split(const SvdSplitDesc& d, const SvdWorkspace& w, hipStream_t s, int* sweeps_out);

// ---- blocked Householder QR preconditioner of the two-site split (tjm_qr.hip) -----------------
struct QrWorkspace {
  cplx* Z;   // [B][zr*zc] column-major work matrix (factor in place, later the isometric factor Q W)
  long z_b0;
  cplx* V;   // [B][panels][16][zr] reflector blocks with explicit zeros / unit diagonal
  long v_b0;
  cplx* T;   // [B][panels][16][16]
  long t_b0;
  cplx* W1;  // [B][16][w_ld] scratch; its head holds the column order of the sorted QR (int[B][w_ld])
  cplx* W2;
  int w_ld;
  cplx* Z2 = nullptr;  // second factorisation (R^H = Q1 R1) of the doubly preconditioned split: same shapes as Z / V / T
  cplx* V2 = nullptr;
  cplx* T2 = nullptr;
  int* colperm() const { return reinterpret_cast<int*>(W1); }
  QrWorkspace second() const { QrWorkspace q = *this; q.Z = Z2; q.V = V2; q.T = T2; return q; }
};
size_t qr_carve(QrWorkspace& q, char* base, int max_dim, int B);  // lays the buffers out behind base, returns the bytes used
// helpers of the accumulation-free split (tjm_svd.hip: svd_split_qr2)
int qr_gather_scaled(const cplx* G, long g_b0, int rows, int ncols, int d, const real* sigma, int sig_ld, const int* keep, int keep_stride,
                     cplx* Z, long z_b0, int nb0, hipStream_t s);  // Z[k][bond*d+p] = G[(p,bond)][k] / sigma_k (0 beyond keep)
int qr_identity(cplx* C, long c_b0, int rows, int ncols, int nb0, hipStream_t s, const int* ids = nullptr, const int* keep = nullptr,
                int keep_stride = 0);
int qr_r_times_sigma(const cplx* Z, long z_b0, int zr, int ncols, const real* sigma, int sig_ld, const int* keep, int keep_stride, cplx* Rs,
                     long rs_b0, int nb0, hipStream_t s);  // Rs[k][j] = R[k][j] sigma_j for k <= j < keep, else 0 (row-major ncols x ncols)
int qr_adjoint_triangle(const QrWorkspace& q, int n, int nb0, const int* ids, hipStream_t s);  // Z2 = R^H of the factored Z
size_t qr_workspace_bytes(int max_dim, int B);
// square > 0: embed the rectangular matrix in a square one of that size (zero rows / columns behind the data)
int qr_prepare(const cplx* theta, long th_b0, int m, int n, int dist, int d, const QrWorkspace& q, int nb0, const int* ids, hipStream_t s,
               int square = 0);
int qr_factor(const QrWorkspace& q, int zr, int zc, int nb0, const int* ids, hipStream_t s);
int qr_apply_q(const QrWorkspace& q, int zr, int zc, cplx* C, long c_b0, int nc, int nb0, const int* ids, hipStream_t s);
int qr_scatter(const cplx* in, long in_b0, int ld, const ExtractDesc& x, const int* chi_keep, int chi_stride, int nb0, const int* ids,
               hipStream_t s);
// Mixed-precision variant of the square two-site split (fp64 library only, tjm_mixed.h): workspace of its complex64 phase.
struct MixedWorkspace { void* base = nullptr; size_t bytes = 0; int max_dim = 0, B = 0; };
size_t mixed_split_workspace_bytes(int max_dim, int B);  // 0: not served (complex64 build, size out of range, TJM_NO_MIXED_SPLIT)
void mixed_stats_get(double* out10, bool reset);  // batched splits, complex64 sweeps, fp64 Jacobi sweeps, batches sent to the fp64 path,
                                                  // trajectories finished by the fp64 Jacobi, batches with a second polar step, executed
                                                  // complex64 rotation slots x rows, applied complex64 rotations x rows, GEMMs of the
                                                  // fp64 phase, their nominal real flops
void qr_profile_enable(int every);                // launch sampler of qr_block_apply_multi_kernel (tjm_qr.hip); get: out5, see there
void qr_profile_get(double* out5);
void mixed_qr_profile_enable(int every);          // ... of the complex64 instance inside the fp64 library (the mixed split's preconditioner)
void mixed_qr_profile_get(double* out5);
void mixed_profile_enable(int every);             // launch sampler of the complex64 tile kernel (as profile_enable for the fp64 one)
void mixed_profile_get(double* total_ms, double* total_bytes, long* samples);
int svd_split_qr(const SvdSplitDesc& d, const SvdWorkspace& w, const QrWorkspace& q, hipStream_t s, int* sweeps_out,
                 const MixedWorkspace* mx = nullptr);

}  // namespace tjm
