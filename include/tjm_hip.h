/* C ABI of the MI355X-native Tensor-Jump-Method hot path (libtjm_hip.so).
 *
 * Plain pointers and sizes only; device buffers are caller-allocated (the Python host
 * passes torch.Tensor.data_ptr()).  Every function returns 0 (TJM_OK) or a negative
 * error code.  All complex data is complex128 stored interleaved (re, im).
 *
 * libtjm_hip_f32.so is the same sources compiled with -DTJM_F32: fp32 arithmetic and complex64 device
 * storage behind the SAME entry points.  Host arrays of the engine-level calls stay float64 / complex128 and
 * are converted at the boundary; only the kernel-level parity exports, whose operands are device pointers,
 * take complex64 / float32 device arrays there.
 *
 * The reference (munich-quantum-toolkit/yaqs) has no FFI; the seam this library sits
 * behind is its backend-function contract, simulator.py:164-185 / 1539-1547:
 *     backend((traj_idx, MPS, NoiseModel|None, AnalogSimParams, MPO))
 *         -> (results[n_obs, T], diagnostics[3, T], MPS|None)
 * whose body is analog/analog_tjm.py:206-462.  Each entry point below cites the
 * reference function(s) it replaces (paths relative to src/mqt/yaqs).
 */
#ifndef TJM_HIP_H
#define TJM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TJM_OK 0
#define TJM_ERR_ARG (-1)
#define TJM_ERR_HIP (-2)
#define TJM_ERR_WORKSPACE (-3)
#define TJM_ERR_NOT_IMPLEMENTED (-4) /* maps to NotImplementedError (dissipation.py:136-138) */
#define TJM_ERR_NUMERIC (-5)         /* maps to ValueError (stochastic_process.py:178-186)   */
#define TJM_ERR_STATE (-6)
#define TJM_ERR_ASSERT (-7)          /* maps to AssertionError: imaginary expectation value (mps.py:1233) */
#define TJM_ERR_CAPACITY (-8)        /* a truncation wanted more singular values than the engine's chi_max holds: re-run larger */

/* ---- library ------------------------------------------------------------------------ */
int tjm_version(void);
const char* tjm_error_string(int code);

/* ---- batched engine: B trajectories in lock-step on one GPU ------------------------- *
 * Replaces the fork pool of core/parallel_utils.py:331-390 (run_backend_parallel): the
 * trajectory axis becomes the batch axis of every kernel launch.                          */
typedef struct tjm_engine tjm_engine;

/* mpo_bond[L+1]: MPO bond dimensions, mpo_bond[0] = mpo_bond[L] = 1 (mpo.py:45-50).  1 <= B <= 65535 (the trajectory index is a
 * grid dimension of the kernels; more trajectories run as further engines or further chunks). */
int tjm_engine_create(tjm_engine** out, int32_t L, int32_t d, int32_t chi_max, int32_t B, const int32_t* mpo_bond);
/* The same with storage of bond k = min(chi_max, cap_slack * min(d^k, d^(L-k))): cap_slack = 1 is the exact Schmidt-rank bound of
 * tjm_engine_create; the BUG integrator (core/methods/bug.py) needs 2 for its stacked trial bases near the chain ends. */
int tjm_engine_create_ex(tjm_engine** out, int32_t L, int32_t d, int32_t chi_max, int32_t B, const int32_t* mpo_bond, int32_t cap_slack);
void tjm_engine_destroy(tjm_engine* e);
size_t tjm_engine_workspace_bytes(const tjm_engine* e);
/* workspace: device memory of at least workspace_bytes; stream: hipStream_t (0 = default). */
int tjm_engine_bind(tjm_engine* e, void* dev_workspace, size_t bytes, void* hip_stream);
/* AnalogSimParams knobs of the path (simulation_parameters.py:520-613).
 * trunc_mode: 0 discarded_weight, 1 relative, 2 hard_cutoff, 3 relative_discarded_weight.
 * max_bond <= 0: no cap.  tdvp_mode: 2 = "2site", 1 = "1site" (integrators.py:44-158).
 * max_bond may exceed chi_max (the reference's presets ask for 4096 or no cap at all while the bonds of most runs stay far
 * smaller): the engine then works within chi_max and REPORTS the first truncation that needed more, see below. */
int tjm_engine_set_params(tjm_engine* e, double dt, double svd_threshold, int32_t trunc_mode, int32_t max_bond,
                          double krylov_tol, int32_t tdvp_mode, int32_t tdvp_sweeps);
/* *flag = 1 when, since the last clear, a truncation (svd_utils.py:22-104) kept fewer values than its rule and max_bond ask for
 * because the new bond only stores chi_max; the states are then not the reference's and the caller re-runs the trajectories
 * on a larger engine.  tjm_engine_run checks after every time step and returns TJM_ERR_CAPACITY. */
int tjm_engine_capacity_overflow(tjm_engine* e, int32_t* flag, int32_t clear);
/* Growing the storage: set 0 of dst slot k = set 0 of src slot (src_first + k), zero-padded to dst's larger bond capacities
 * (same L and d; dst.chi_max >= src.chi_max; src_first + dst.B <= src.B).  Device-to-device on dst's stream. */
int tjm_engine_adopt_state(tjm_engine* dst, tjm_engine* src, int32_t src_first);
/* host pointer: per site the tensor (phys_out, phys_in, chi_l, chi_r) C-contiguous, sites concatenated. */
int tjm_engine_set_mpo(tjm_engine* e, const double* host_mpo);
/* NoiseModel.processes (noise_model.py:227-243), one entry per process:
 *   nsites[k] in {1,2}; sites[2k], sites[2k+1]; gamma[k]; pauli[k] = is_pauli(process);
 *   mats: 2 d^4 doubles per process (32 for qubits; 1-site: d x d row-major in the leading entries, adjacent pair: d^2 x d^2);
 *   factors: 4 d^2 doubles per process (two d x d matrices; 16 for qubits) when has_factors[k]. */
int tjm_engine_set_noise(tjm_engine* e, int32_t nproc, const int32_t* nsites, const int32_t* sites, const double* gamma,
                         const int32_t* pauli, const double* mats, const double* factors, const int32_t* has_factors);
/* MPS tensors (sigma, chi_l, chi_r) C-contiguous (mps.py:58), sites concatenated; bonds[L+1].
 * The state is broadcast to all B slots of state set `set` (0 = trajectory state, 1 = measurement copy). */
int tjm_engine_load_state(tjm_engine* e, int32_t set, const double* host_tensors, const int32_t* bonds);
/* one slot only (per-trajectory initial states); same tensor layout */
int tjm_engine_load_state_slot(tjm_engine* e, int32_t set, int32_t b, const double* host_tensors, const int32_t* bonds);
int tjm_engine_copy_state(tjm_engine* e, int32_t dst_set, int32_t src_set); /* copy.deepcopy(phi), analog_tjm.py:179 */
size_t tjm_engine_padded_state_elems(const tjm_engine* e);                   /* complex elements per trajectory */
int tjm_engine_bond_caps(const tjm_engine* e, int32_t* caps);                /* L+1 padded bond extents */
int tjm_engine_export_state(tjm_engine* e, int32_t set, int32_t b, double* host_padded, int32_t* bonds);
/* host uniforms [B][n_per_traj]: the per-trajectory PCG64 double streams of core/random_utils.py:20-69 */
int tjm_engine_set_uniforms(tjm_engine* e, const double* host_u, int32_t n_per_traj);

/* apply_unitary_evolution -> tdvp -> sweep_2site | sweep_1site (analog/evolution.py:24-51, tdvp/tdvp.py:69-111,
 * tdvp/integrators.py:44-291) on every trajectory of the set. */
int tjm_engine_tdvp(tjm_engine* e, int32_t set);
/* apply_dissipation (core/methods/dissipation.py:50-183). */
int tjm_engine_dissipate(tjm_engine* e, int32_t set, double dt);
/* ---- digital (circuit) path: TEBD gates with per-gate local noise (digital/digital_tjm.py:636-749) ---- */
/* apply_dissipation started from a known centre (after a TEBD split the centre sits on the gate's right site). */
int tjm_engine_dissipate_from(tjm_engine* e, int32_t set, double dt, int32_t center);
/* create_local_noise_model (digital_tjm.py:187-204): restrict dissipation / jumps to the listed process indices;
 * n < 0 re-activates every process. */
int tjm_engine_set_noise_filter(tjm_engine* e, int32_t n, const int32_t* idx);
/* MPS.normalize("B", "QR") from a known centre (mps.py:815-839). */
int tjm_engine_normalize_qr(tjm_engine* e, int32_t set, int32_t center);
/* _apply_single_qubit_gate (digital_tjm.py:304-309); host 2x2 complex128 row-major. */
int tjm_engine_apply_single(tjm_engine* e, int32_t set, int32_t site, const double* host_mat);
/* apply_two_qubit_gate_tebd (digital_tjm.py:455-533) for a nearest-neighbour gate on (left, left+1) of a state with
 * centre 0; host U[(out_l,out_r),(in_l,in_r)] 4x4 complex128 row-major (mpo_utils.py:104-159 index order). */
int tjm_engine_tebd_gate(tjm_engine* e, int32_t set, int32_t left, const double* host_u);
/* The same gate from a state whose orthogonality centre is at `center` (chains of gates without noise in between, e.g. the
 * adjacent SWAPs that route a long-range gate, digital_tjm.py:476-499).  The centre ends on left + 1. */
int tjm_engine_tebd_gate_at(tjm_engine* e, int32_t set, int32_t left, int32_t center, const double* host_u);
/* stochastic_process (core/methods/stochastic_process.py:190-292); jumped[B], dp[B] optional host outputs. */
int tjm_engine_stochastic(tjm_engine* e, int32_t set, double dt, int32_t* jumped, double* dp);
/* Physical-leg moment matrices M[site][b][p][q] = <psi| |p><q|_site |psi> (host, complex128);
 * <O_site> = sum_pq O[p][q] M[p][q].  Replaces MPS.evaluate_observables / local_expect
 * (mps.py:961-1047, 1178-1234) for one-site observables. */
int tjm_engine_site_moments(tjm_engine* e, int32_t set, double* host_M);
/* Same plus the nearest-neighbour two-site moments M2[site][b][(s,t)][(s',t')] for two-site observables
 * (mps.py:999-1047) and adjacent two-site jump weights (stochastic_process.py:53-83). */
int tjm_engine_site_moments2(tjm_engine* e, int32_t set, double* host_M, double* host_M2);
int tjm_engine_bond_dims(tjm_engine* e, int32_t set, int32_t* host_chi); /* [B][L+1]; record_diagnostics mps.py:549-602 */
int tjm_engine_site0_normsq(tjm_engine* e, int32_t set, double* host_out); /* MPS.norm(0), mps.py:1539-1565 */
/* counters: matvecs, krylov calls, svds, svd sweeps, two-site updates */
/* Singular values (descending) of theta = A_site A_{site+1} reshaped (d chi_l) x (d chi_r): the spectrum behind
 * MPS.get_entropy and MPS.get_schmidt_spectrum (mps.py:604-678).  spectrum[B][n_out] (host), zero-filled past min(m, n). */
int tjm_engine_bond_spectrum(tjm_engine* e, int32_t set, int32_t site, double* spectrum, int32_t n_out);
/* MPS.project_onto_bitstring (mps.py:1495-1537): probability of the computational-basis outcome bits[L] (site 0 first)
 * for every resident trajectory; prob[B] (host). */
int tjm_engine_bitstring_probability(tjm_engine* e, int32_t set, const uint8_t* bits, double* prob);
/* Scheduled two-site jump (core/methods/scheduled_jumps.py:88-106): a d^2 x d^2 operator (row-major, index
 * d*sigma_left + sigma_right) on the merged pair (left, left+1), then split_two_site("right") with the engine's truncation
 * settings and `min_keep`; the gauge is not touched beforehand.  One-site scheduled jumps use tjm_engine_apply_single. */
int tjm_engine_apply_pair(tjm_engine* e, int32_t set, int32_t left, const double* host_u, int32_t min_keep);
/* QR sweep from `center` down to site 0 without normalising (set_canonical_form / the sweep of normalize("B"),
 * mps.py:790-839); with center = L-1 it works from any gauge.  tjm_engine_site0_normsq then returns the squared norm. */
int tjm_engine_canonicalize_qr(tjm_engine* e, int32_t set, int32_t center);
/* MPS.measure_shots / measure_single_shot (mps.py:1282-1417) for every resident trajectory, from a normalised state with
 * centre 0: `shots` projective samples of all L sites.  rotation: the 2x2 basis change of mps.py:1306-1311 (row-major
 * complex; identity for "Z"); uniforms[B][shots][L] (host): the draw of rng.choice at each site; bits[B][shots][L] (host):
 * outcomes, site 0 first (the reference packs them as sum(bit_i << i)). */
int tjm_engine_sample_shots(tjm_engine* e, int32_t set, int32_t shots, const double* rotation, const double* uniforms, uint8_t* bits);
int tjm_engine_stats(const tjm_engine* e, int64_t* out5);
/* the same five counters followed by: two-site H_eff applies (a subset of matvecs), environment updates, H_eff applies served by
 * the direct form (no T2 tensor: monomial MPO rows behind certified identity channels), matrices factorised (batched SVD calls x trajectories in the call), Krylov calls whose
 * environments were examined for identity channels, channels certified (at most two per call), trajectory-steps whose scalar dissipation sweep was certified away, jumps applied in
 * place on certified states, trajectory-bonds whose certificate test ran on the blocked Cholesky kernel (bonds above 128); writes min(n, 14) values */
int tjm_engine_stats_ex(const tjm_engine* e, int64_t* out, int32_t n);
/* Live device time per kernel class of a step, bracketed with HIP events on the engine's stream (the source of bench.py's
 * roofline object): class 0 = SVD family (split_two_site and the SVD centre shifts: QR, Jacobi, truncation, their GEMMs),
 * 1 = Krylov exponentials (project_site / project_bond applies, Lanczos vector kernels), 2 = environment updates.
 * read(): summed milliseconds and region counts since enable. */
int tjm_engine_profile(tjm_engine* e, int32_t enable);
int tjm_engine_profile_read(tjm_engine* e, double* ms3, int64_t* regions3);

/* ---- site-level steps: sweeps whose schedule the host decides per trajectory ---------- *
 * Dynamic TDVP (core/methods/tdvp/integrators.py:294-511) takes, site by site, the two-site branch for the trajectories whose
 * bond is below max_bond_dim and the one-site branch with a QR bond transfer for those that have reached it, so a lock-step
 * batch splits into two index lists at every site: the Python host reads the bond table (tjm_engine_bond_dims), forms the
 * lists and calls these steps.  ids = host int32 list of n trajectory slots, or NULL for all of them.
 *   step_env_init   initialize_right_environments + the left boundary (primitives.py:139-174, integrators.py:325-336)
 *   step_two_site   merge_two_site, update_site on the pair, split_tdvp: dist 0 = "right", 1 = "left"; capped = 0 is
 *                   dynamic=True (no max_bond_dim in the truncation, sweep_utils.py:47-84)
 *   step_one_site   update_site on one tensor (primitives.py:484-520)
 *   step_env        left = 1: update_left_environment into site + 1; left = 0: update_right_environment into site - 1
 *   step_qr_bond    right_qr / left_qr of the site, environment update with Q, update_bond on C over dt, C into the neighbour
 *                   (integrators.py:352-377, 441-466; leftward it is the projector-splitting step of the fixed one-site sweep,
 *                   integrators.py:126-158 - sweep_dynamic's own leftward line transposes left_qr's factor a second time and
 *                   is not gauge invariant, see tjm_engine.hip).  max_bond > 0: a new bond the thin QR left above it is cut back
 *                   to it along the NEW index, "site_tensor[:, :, :cap]; bond_tensor[:cap, :]" (integrators.py:361-364, 452-455)
 *   step_cap_bond   _sync_bond_dim where it truncates (sweep_utils.py:110-163): merged pair, sqrt-distributed split capped at
 *                   `target`, min_keep 1 */
int tjm_engine_step_env_init(tjm_engine* e, int32_t set);
int tjm_engine_step_two_site(tjm_engine* e, int32_t set, int32_t site, double dt, int32_t dist, int32_t capped, const int32_t* ids, int32_t n);
int tjm_engine_step_one_site(tjm_engine* e, int32_t set, int32_t site, double dt, const int32_t* ids, int32_t n);
int tjm_engine_step_env(tjm_engine* e, int32_t set, int32_t site, int32_t left, const int32_t* ids, int32_t n);
int tjm_engine_step_qr_bond(tjm_engine* e, int32_t set, int32_t site, int32_t right, double dt, int32_t max_bond, const int32_t* ids, int32_t n);
int tjm_engine_step_cap_bond(tjm_engine* e, int32_t set, int32_t bond, int32_t target, const int32_t* ids, int32_t n);

/* Steps of the Basis-Update and Galerkin integrator (core/methods/bug.py) for the whole batch, sequenced by the host
 * (yaqs_amd/tjm.py: bug_step = bug(), bug.py:213-257); the engine must have been created with cap_slack >= 2.
 *   step_bug_prepare  prepare_canonical_site_tensors (bug.py:35-62): coefficient-bearing centres and left environments
 *   step_bug_site     _local_update + build_trial_basis (bug.py:65-125) at `site` (L-1 ... 1): Krylov predictor, left QR of
 *                     [retained | predictor] stacked along the left bond, basis-change matrix, transported centre, right block
 *   step_bug_root     the root update of bug_sweep (bug.py:186-196)
 *   step_flip         MPS.flip_network (mps.py:680-698); the caller loads the reflected MPO (mpo.py:1612-1630) with set_mpo
 *   step_compress     MPS.compress (mps.py:841-899) with the given truncation settings (max_bond_dim <= 0: none) */
/* Whole sweeps in one call (round 6).  tjm_engine_sweep_dynamic: one sweep of tdvp(tdvp_mode="dynamic") with time step dt
 * (integrators.py:294-511; sweep_utils.py:280-302 for the bonds a previous sweep left above the cap) - the branch of every
 * trajectory at every site is decided inside the library from one column of the bond table (max_bond_dim < 1: no cap).
 * tjm_engine_bug_sweep: one half-sweep of the BUG integrator (bug_sweep, bug.py:128-196). */
int tjm_engine_sweep_dynamic(tjm_engine* e, int32_t set, int32_t max_bond_dim, double dt);
int tjm_engine_bug_sweep(tjm_engine* e, int32_t set, double dt);
int tjm_engine_step_bug_prepare(tjm_engine* e, int32_t set);
int tjm_engine_step_bug_site(tjm_engine* e, int32_t set, int32_t site, double dt);
int tjm_engine_step_bug_root(tjm_engine* e, int32_t set, double dt);
int tjm_engine_step_flip(tjm_engine* e, int32_t set);
int tjm_engine_step_compress(tjm_engine* e, int32_t set, double threshold, int32_t max_bond_dim, int32_t trunc_mode);
/* Long-range two-qubit gate as a matrix product operator (digital_tjm.py:536-557: MPO.from_gate(gate, L).multiply(state), mpo.py:1511-1548,
 * the reference's default route for distant pairs): U = sum_k left_ops[k] (x) right_ops[k] on sites (first, last), first < last, identity
 * threads in between (gate_library.py:29-126); left_ops / right_ops are [rank][d][d] row-major complex (out, in).  Bonds first+1 .. last
 * grow by the factor `rank` (MPS index first in the fused leg, mpo_utils.py:27-56); a product beyond the storage raises the capacity
 * flag.  The caller compresses afterwards (tjm_engine_step_compress = MPS.compress, mps.py:841-899). */
int tjm_engine_apply_gate_mpo(tjm_engine* e, int32_t set, int32_t first, int32_t last, int32_t rank, const double* left_ops, const double* right_ops);

/* ---- whole trajectories in one call -------------------------------------------------- *
 * The body of the backend contract: analog_tjm_1 / analog_tjm_2 (analog/analog_tjm.py:206-462) for the B resident
 * trajectories, after set_params / set_mpo / set_noise / load_state.  Observables are given in the reference's
 * site-sorted order (simulation_parameters.py:419-456); a two-site observable acts on (site, site+1) (mps.py:999-1047). */
typedef struct {
  int32_t order;            /* 1: analog_tjm_1, 2: analog_tjm_2 */
  int32_t n_times;          /* len(sim_params.times) = number of steps + 1 */
  int32_t sample_timesteps; /* 1: one column per time point, 0: final time only */
  int32_t has_noise;        /* 0: noise_model is None */
  int32_t has_seed;         /* 0: random_seed is None (fresh OS entropy, random_utils.py:33-35) */
  uint64_t seed;            /* sim_params.random_seed */
  int32_t n_obs;
  const int32_t* obs_nsites; /* [n_obs] 1 or 2 */
  const int32_t* obs_site;   /* [n_obs] first site */
  const double* obs_matrix;  /* [n_obs][2 d^4] doubles (32 for qubits): row-major complex d x d (leading entries) or d^2 x d^2 */
  /* Continuation after TJM_ERR_CAPACITY (all zero / null for a run from the initial state).
   * start_step = j > 0: set 0 already holds the trajectory states at the START of time step j (tjm_engine_adopt_state), the
   * columns measured before j are already in results / diagnostics, and rng_pos[B] holds every trajectory's cursor into its
   * random stream.  start_phase = 1 (order 2 only): the trajectory state is the one AFTER step j; only its sampling is redone.
   * rng_pos (in/out, may be null) and resume (out, may be null: int32[2] = {step, phase}) are written when the run stops with
   * TJM_ERR_CAPACITY; set 0 then holds the states to continue from.  resume[0] = 0 means: start again from the initial state. */
  int32_t start_step;
  int32_t start_phase;
  int64_t* rng_pos;
  int32_t* resume;
} tjm_run_config;
/* traj[B]: trajectory indices (seeds of the per-trajectory streams); results[B][n_obs][T], diagnostics[B][3][T] with
 * T = n_times if sample_timesteps else 1 (host, float64).  Returns TJM_ERR_ASSERT for an imaginary expectation
 * value (AssertionError in mps.py:1233), TJM_ERR_NUMERIC for zero / non-finite jump weights and TJM_ERR_CAPACITY as soon as
 * a time step needed a bond beyond chi_max (the step is rolled back: see start_step). */
int tjm_engine_run(tjm_engine* e, const tjm_run_config* cfg, const int64_t* traj, double* results, double* diagnostics);
/* The same with one status per trajectory (SURVEY 8b's out_status): a trajectory whose state holds a non-finite number - a poisoned
 * input, a blow-up - is taken out of the run (status[b] = TJM_ERR_NUMERIC, its result rows NaN, its slot refilled with a copy of a
 * healthy neighbour so that every kernel keeps seeing finite data) and the other B - 1 finish exactly as they would have without it;
 * the states are screened before the first step and after every step.  The reference loses one job of its pool in that case, not the
 * pool (core/parallel_utils.py:361-383).  The return code still reports failures that concern the batch (capacity, arguments, HIP).
 * status is IN / OUT for a continued run (cfg->start_step > 0, e.g. after TJM_ERR_CAPACITY on a larger engine): a trajectory whose
 * entry is not TJM_OK on entry stays out (its slot holds a donor's state); a fresh run (start_step == 0) overwrites every entry.  The
 * sampling copy of the order-2 driver and the state after its half-step prelude are screened like the state after every step. */
int tjm_engine_run_status(tjm_engine* e, const tjm_run_config* cfg, const int64_t* traj, double* results, double* diagnostics, int32_t* status);
/* The reference's host random streams, bit-compatible with NumPy (core/random_utils.py:20-69):
 * timestep < 0: make_trajectory_rng(traj, base_seed=seed).random(n); otherwise make_sample_rng(traj, timestep, seed). */
int tjm_rng_uniforms(int32_t has_seed, uint64_t seed, uint64_t traj, int64_t timestep, int32_t n, double* out);

/* ---- single kernels, exported for parity tests -------------------------------------- */
typedef struct {
  const void* A; const void* B; void* C;
  int32_t M, N, K;
  int64_t a_rs, a_cs, b_rs, b_cs, c_rs;
  int32_t nks; int64_t a_ks, b_ks;
  int32_t nb0, nb1, nb2;
  int64_t a_b0, a_b1, a_b2, b_b0, b_b1, b_b2, c_b0, c_b1, c_b2;
  int32_t conjA, conjB;
} tjm_gemm_desc;
/* np.tensordot / merge_two_site contractions (decompositions.py:87-102, primitives.py:77-226) */
int tjm_zgemm_batched(const tjm_gemm_desc* desc, void* hip_stream);
/* split_two_site (decompositions.py:105-185) + truncate (linalg/svd_utils.py:22-104):
 * theta[B][m][n] row-major with m = d*capL, n = d*capR;
 * left[B][d][capL][capM], right[B][d][capM][capR], chi_lrm int32[B][3] = (chiL, chiR, out chiM),
 * spectrum[B][spec_ld] optional, work = device scratch of tjm_svd_workspace_bytes(max(m,n), B). */
size_t tjm_svd_workspace_bytes(int32_t max_dim, int32_t B);
int tjm_svd_split(const void* theta, int32_t B, int32_t d, int32_t capL, int32_t capR, int32_t capM, void* left, void* right,
                  int32_t distribution, int32_t trunc_mode, double threshold, int32_t max_bond, int32_t min_keep,
                  int32_t* chi_lrm, double* spectrum, int32_t spec_ld, void* work, size_t work_bytes, int32_t* sweeps_out,
                  void* hip_stream);
/* Same split with the Householder-QR preconditioner the engine uses for d*cap >= 64 (theta = Q R, Jacobi on R^H,
 * isometric factor Q W); work must hold tjm_svd_qr_workspace_bytes. */
size_t tjm_svd_qr_workspace_bytes(int32_t max_dim, int32_t B);
int tjm_svd_split_qr(const void* theta, int32_t B, int32_t d, int32_t capL, int32_t capR, int32_t capM, void* left, void* right,
                     int32_t distribution, int32_t trunc_mode, double threshold, int32_t max_bond, int32_t min_keep,
                     int32_t* chi_lrm, double* spectrum, int32_t spec_ld, void* work, size_t work_bytes, int32_t* sweeps_out,
                     void* hip_stream);
/* The contractions of the sweep on explicit tensors.  Every operand is a device array of nb slots (nb must equal the engine's B),
 * contiguous per slot, within the engine's bond capacities; host_w is ONE host MPO tensor in the reference's (phys_out, phys_in,
 * chi_l, chi_r) order (mpo.py:45-50), for nsites = 2 the merged tensor of merge_mpo_tensors (primitives.py:54-74).
 *   tjm_heff_apply   project_site (core/methods/tdvp/primitives.py:180-204): x, y [P][ca][cb] (P = d or d^2),
 *                    Lenv [ca][Dl][ca], Renv [cb][Dr][cb]
 *   tjm_env_update   update_left_environment (left = 1, primitives.py:77-107): A [d][ca][cb], env [ca][Dl][ca] -> out [cb][Dr][cb];
 *                    update_right_environment (left = 0, primitives.py:110-136): env [cb][Dr][cb] -> out [ca][Dl][ca]; bra = ket = A
 *   tjm_project_bond project_bond (primitives.py:207-226): C, y [cu][cv], Lenv [cu][D][cu], Renv [cv][D][cv]
 *   tjm_lanczos_expm update_site = expm_krylov o project_site (primitives.py:484-520, matrix_exponential.py:33-173):
 *                    y = exp(-i dt H_eff) x with the reference's adaptive stop at `tol`; matvecs (optional) = Lanczos steps issued */
int tjm_heff_apply(tjm_engine* e, int32_t nsites, int32_t ca, int32_t cb, int32_t Dl, int32_t Dr, const void* x, const void* Lenv, const void* Renv,
                   const double* host_w, void* y, int32_t nb);
int tjm_env_update(tjm_engine* e, int32_t left, int32_t ca, int32_t cb, int32_t Dl, int32_t Dr, const void* A, const void* env, const double* host_w,
                   void* out, int32_t nb);
int tjm_project_bond(tjm_engine* e, int32_t cu, int32_t cv, int32_t D, const void* C, const void* Lenv, const void* Renv, void* y, int32_t nb);
int tjm_lanczos_expm(tjm_engine* e, int32_t nsites, int32_t ca, int32_t cb, int32_t Dl, int32_t Dr, const void* x, const void* Lenv, const void* Renv,
                     const double* host_w, double dt, double tol, void* y, int32_t nb, int64_t* matvecs);
/* One orthogonality-centre shift of the loaded state, shift_orthogonality_center_right / _left (mps.py:719-788): direction = +1
 * moves the centre from `site` to site + 1, -1 to site - 1; use_svd = 0: QR (exact gauge move), 1: SVD with the shift's own
 * truncation (discarded weight 1e-12, no cap). */
int tjm_engine_center_shift(tjm_engine* e, int32_t set, int32_t site, int32_t direction, int32_t use_svd);
/* create_probability_distribution (stochastic_process.py:139-187) of a state with centre 0: order[k] = index of the k-th channel
 * in the reference's sweep order, weights[B][n] = dt * gamma * ||L psi||^2 (unnormalised); *n_out = number of channels.
 * TJM_ERR_NUMERIC for a zero / non-finite total (stochastic_process.py:178-186). */
int tjm_engine_jump_weights(tjm_engine* e, int32_t set, double dt, int32_t* order, double* weights, int32_t* n_out);
/* exp(-i dt T_k) e_1 of the Lanczos tridiagonal (matrix_exponential.py:147-163); device pointers. */
int tjm_tridiag_expm(const double* alpha, const double* beta, int32_t k, double dt, double* out_k_complex, void* hip_stream);

/* ---- measurement ---------------------------------------------------------------------- *
 * Brackets every `every`-th launch of the dominant kernel (the Jacobi block-pair kernel of the SVD split)
 * with HIP events on the launch stream; 0 switches it off.  read(): summed duration [ms], summed
 * algorithmic bytes (2 x 16 columns x rows x 16 B per block-pair visit of a live trajectory), sample count. */
int tjm_profile_cross_kernel(int32_t every);
int tjm_profile_cross_kernel_read(double* total_ms, double* total_bytes, int64_t* samples);
/* Work the tiled Jacobi kernels really executed in this process since the last reset (all engines): out4 = { rotation slots x rows
 * (every column pair of a visited tile: one dot product and one - possibly identity - plane rotation of `rows` complex entries),
 * applied rotations x rows, sweeps, solves }.  28 real flops per slot-row (8 dot + 20 rotation): bench.py reports the executed
 * flops next to the nominal 88 n^3 of the reference's zgesdd (core/linalg/svd.py:51-104). */
int tjm_svd_work_read(double* out4, int32_t reset);
/* Mixed-precision two-site split (fp64 library; square splits of 128 ... 512 rows - replaces the same reference lines as
 * tjm_svd_split_qr, core/linalg/svd.py:51-104, core/methods/decompositions.py:105-185): an approximate singular basis from the complex64
 * Jacobi (the same kernels, compiled for complex64 into this library), made exactly unitary in fp64 (polar step) and refined to
 * rounding by first-order eigenvector corrections from the Gram matrix of theta x basis - all fp64 work on the matrix cores; the fp64
 * Jacobi kernels only finish single trajectories that fail the final check.  out10 = { batched splits served, complex64 sweeps, fp64
 * Jacobi sweeps, batches sent back to the all-fp64 path, trajectories finished by the fp64 Jacobi, batches that needed a second polar
 * step, executed complex64 rotation slots x rows, applied complex64 rotations x rows (28 fp32 flops per slot-row, as in
 * tjm_svd_work_read), GEMM launches of the fp64 phase, their nominal real flops (8 M N K per complex product) } since the last reset.
 * Zeros in the complex64 library.  Switch: TJM_NO_MIXED_SPLIT. */
int tjm_svd_mixed_read(double* out10, int32_t reset);
/* tjm_profile_cross_kernel also samples the complex64 instance of the tile kernel (the first phase of the mixed split); its totals: */
int tjm_profile_cross_kernel_read_c64(double* total_ms, double* total_bytes, int64_t* samples);
/* Launch sampler of the fp64 GEMM kernel on v_mfma_f64_4x4x4_4b_f64 (zgemm4_kernel, tjm_gemm.hip; the products of
 * core/methods/tdvp/primitives.py:77-226 and of the two-site split's fp64 phase): every `every`-th launch bracketed by HIP events on its
 * stream, 0 switches it off.  read(): out6 = { summed duration of the sampled launches [ms], sampled launches, all launches, executed output
 * tiles x K counted ON THE DEVICE over all launches (one 64 x 64 tile x one unit of K = 3 x 2 x 64 x 64 real matrix-core flops: three real
 * products per complex one; masked trajectories and mirror tiles of Hermitian products are not counted), algorithmic bytes of the sampled
 * launches (operands and result of a launch once each), the same for all launches }.  Call read() after synchronising the streams.
 * Zeros in the complex64 library. */
/* Launch sampler of the block-reflector apply of the QR preconditioner (qr_block_apply_multi_kernel): every N-th launch bracketed by
 * HIP events; out5 = summed duration of the sampled launches (ms), their nominal real flops (8 per complex multiply-add), samples,
 * all launches, nominal flops of all launches - of the library's own arithmetic and, in the fp64 library, of the complex64 instance
 * that preconditions the mixed-precision split (zeros in the complex64 library). */
int tjm_profile_qr_apply(int32_t every);
int tjm_profile_qr_apply_read(double* out5, double* out5_c64);
int tjm_profile_gemm(int32_t every);
int tjm_profile_gemm_read(double* out6);

#ifdef __cplusplus
}
#endif
#endif /* TJM_HIP_H */
