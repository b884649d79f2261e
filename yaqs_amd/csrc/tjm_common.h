// Shared declarations for the MI355X (gfx950) Tensor-Jump-Method kernels.
// All device data is complex128 stored interleaved (re, im) - the NumPy / torch layout.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#include "../../include/tjm_hip.h"

namespace tjm {

// Arithmetic type of the build.  libtjm_hip.so computes in fp64 / complex128, the reference's precision; the same sources compiled
// with -DTJM_F32 give libtjm_hip_f32.so, the complex64 variant (same C ABI: host arrays stay complex128 / float64 and are converted
// at the boundary).  Everything that depends on the type - the MFMA instruction and its result map, machine epsilon, the cross-lane
// helpers for 32- / 64-bit payloads - is named here.
#ifdef TJM_F32
typedef float real;
#define TJM_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
// v_mfma_f32_16x16x4_f32: register v of lane l is D[4 (l >> 4) + v][l & 15]
#define TJM_ACC_ROW(lane, reg) (4 * ((lane) >> 4) + (reg))
#define TJM_EPS 1.1920929e-7f
#define TJM_TINY 1e-30f
#define TJM_JACOBI_TOL2 4e-12f   // one-sided Jacobi: rotate while |<p,q>|^2 > tol2 |p|^2 |q|^2 (relative tolerance 2e-6)
#define TJM_NOISE_FLOOR2 1e-12f  // columns below sqrt(floor2) ||X||_F are numerically null
#define TJM_RANK_TOL 1e-5f       // a kept singular value below this fraction of the largest one: rank-deficient, complete the basis
#define TJM_IMAG_TOL 1e-5        // "measurement should be real": the reference's 1e-13 (mps.py:1233) is an fp64 rounding bound
#else
typedef double real;
#define TJM_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)
// v_mfma_f64_16x16x4_f64: register v of lane l is D[(l >> 4) + 4 v][l & 15]
#define TJM_ACC_ROW(lane, reg) (((lane) >> 4) + 4 * (reg))
#define TJM_EPS 2.220446049250313e-16
#define TJM_TINY 1e-300
#define TJM_JACOBI_TOL2 1e-26
#define TJM_NOISE_FLOOR2 1e-26
#define TJM_RANK_TOL 1e-11
#define TJM_IMAG_TOL 1e-13
#endif

struct cplx {
  real x, y;
};
// complex128 as the C ABI hands it over (host side only)
struct zc {
  double x, y;
};

__host__ __device__ inline cplx cmake(real a, real b) { return cplx{a, b}; }
__host__ __device__ inline cplx cadd(cplx a, cplx b) { return cplx{a.x + b.x, a.y + b.y}; }
__host__ __device__ inline cplx csub(cplx a, cplx b) { return cplx{a.x - b.x, a.y - b.y}; }
__host__ __device__ inline cplx cmul(cplx a, cplx b) { return cplx{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__host__ __device__ inline cplx cconj(cplx a) { return cplx{a.x, -a.y}; }
__host__ __device__ inline cplx cscale(cplx a, real s) { return cplx{a.x * s, a.y * s}; }
// acc += a*b
__host__ __device__ inline void cfma(cplx& acc, cplx a, cplx b) {
  acc.x = fma(a.x, b.x, acc.x);
  acc.x = fma(-a.y, b.y, acc.x);
  acc.y = fma(a.x, b.y, acc.y);
  acc.y = fma(a.y, b.x, acc.y);
}

typedef real real4 __attribute__((ext_vector_type(4)));

// global_load_lds_dwordx4: 16 bytes per lane from a per-lane global address straight into LDS at (wave-uniform base) + lane x 16.
#ifdef HIPSIM
#define TJM_GLDS16(gptr, lbase) hipsim_glds16((const void*)(gptr), (void*)(lbase))
#else
#define TJM_GLDS16(gptr, lbase)                                                                             \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                   \
                                   (__attribute__((address_space(3))) void*)(lbase), 16, 0, 0)
#endif

// ---- cross-lane moves of one `real` (device only): DPP row operations, v_readlane, the gfx950 row / half-wave swaps ----------------
#ifdef TJM_F32
template <int CTRL>
__device__ inline real tjm_dpp(real v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ inline real tjm_readlane(real v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ inline real tjm_xor16_sum(real v) {
  const int w = __float_as_int(v);
  const auto a = __builtin_amdgcn_permlane16_swap(w, w, false, false);
  return __int_as_float(a[0]) + __int_as_float(a[1]);
}
__device__ inline real tjm_xor32_sum(real v) {
  const int w = __float_as_int(v);
  const auto a = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return __int_as_float(a[0]) + __int_as_float(a[1]);
}
__device__ inline real tjm_rsq(real x) { return __builtin_amdgcn_rsqf(x); }
__device__ inline void tjm_sincos(real x, real* s, real* c) { sincosf(x, s, c); }
__device__ inline real tjm_rcp(real x) { return __builtin_amdgcn_rcpf(x); }
#else
__device__ inline real tjm_rsq(real x) { return __builtin_amdgcn_rsq(x); }
__device__ inline void tjm_sincos(real x, real* s, real* c) { sincos(x, s, c); }
__device__ inline real tjm_rcp(real x) { return __builtin_amdgcn_rcp(x); }
#endif

// error codes: the TJM_* macros of the C ABI (include/tjm_hip.h)

#define TJM_HIP_CHECK(expr)                                                                 \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      fprintf(stderr, "[tjm_hip] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(_e),   \
              __FILE__, __LINE__);                                                          \
      return TJM_ERR_HIP;                                                                   \
    }                                                                                       \
  } while (0)

// Lanczos breakdown test, "beta < 100 * vec.size * eps" (matrix_exponential.py:126-129).  With the fp64 epsilon this is a rounding
// bound (6e-9 for a two-site block at chi = 256); with the fp32 epsilon the same formula gives 3.1 there - above every beta of a bulk
// site, so the complex64 build declared "invariant subspace" after the first vector and left the block unevolved (found on the
// MI355X: config 3 at chi = 256 lost no norm in its truncations).  The complex64 build uses the size of an fp32 rounding residual
// instead: eps * sqrt(size) per unit of |H|, with the same factor 100 - and |H| is estimated by the first Lanczos coefficients,
// max(|alpha_0|, beta_0) (tjm_breakdown_scale), so that a weak Hamiltonian or a short sub-step, whose beta are all small, is not
// mistaken for an invariant subspace.  The fp64 build keeps the reference's absolute formula (scale 1): parity with it is the point.
__host__ __device__ inline real tjm_breakdown_scale(real alpha0, real beta0) {
#ifdef TJM_F32
  const real a = alpha0 < 0 ? -alpha0 : alpha0;
  return a > beta0 ? a : beta0;
#else
  (void)alpha0; (void)beta0;
  return 1.0;
#endif
}
__host__ __device__ inline real tjm_breakdown_cut(int nloc) {
#ifdef TJM_F32
  return 100.0f * sqrtf((float)nloc) * TJM_EPS;
#else
  return 100.0 * (real)nloc * TJM_EPS;
#endif
}

// Sort key of the rank sorts (squared norms): a NaN compares false with everything, which would leave two entries with the same rank
// and one slot of the permutation unwritten - a stale index that later addresses memory.  Non-finite input must end in an error
// (the reference stops at its first measurement or jump weight), never in a fault: NaN sorts as +infinity.
__device__ inline real tjm_sort_key(real v) { return (v == v) ? v : real(__builtin_huge_val()); }

// Strided batched complex GEMM descriptor.
//   C[m,n] (+)= sum_{ks} sum_k opA(A)[m,k] * opB(B)[k,n]
// Element addresses (in complex elements):
//   A(m,k) = A + m*a_rs + k*a_cs + ks*a_ks + b0*a_b0 + b1*a_b1 + b2*a_b2
//   B(k,n) = B + k*b_rs + n*b_cs + ks*b_ks + b0*b_b0 + b1*b_b1 + b2*b_b2
//   C(m,n) = C + m*c_rs + n        (C rows are contiguous)   + b0*c_b0 + b1*c_b1 + b2*c_b2
// b0 = trajectory slot (optionally remapped through ids[]), b1 / b2 = inner batches (physical indices).
struct GemmDesc {
  const cplx* A;
  const cplx* B;
  cplx* C;
  int M, N, K;
  long a_rs, a_cs, b_rs, b_cs, c_rs;
  int nks;
  long a_ks, b_ks;
  int nb0, nb1, nb2;
  long a_b0, a_b1, a_b2, b_b0, b_b1, b_b2, c_b0, c_b1, c_b2;
  int conjA, conjB;
  int accumulate;     // 0: C = A*B ; +1: C += A*B ; -1: C -= A*B
  const int* ids;     // optional trajectory remap for b0 (device pointer) or nullptr
  const int* active;  // optional per-trajectory mask (device, indexed by remapped id); 0 => skip
  int hermitian;      // 1: the result is Hermitian (a Gram matrix, M == N): only the tiles on and above the diagonal are computed, the
                      // others are written as their mirror images (10 of 16 tiles at 256 x 256)
  // Optional epilogue (tiled kernel only): Re <dot_with, C> over the workgroup's tile, with the FINAL values of C (after accumulate) -
  // the Lanczos coefficient alpha_j = <v_j, H v_j> without a pass of its own.  dot_with has the element offsets of C (same strides);
  // the partial sum of tile t of inner batch (b1, b2) goes to dot_part[b0 * dot_ld + (b1 * nb2 + b2) * tiles + t] (no atomics:
  // one slot per workgroup, added up in a fixed order by the consumer).
  const cplx* dot_with;
  real* dot_part;
  int dot_ld;
  // Optional (zgemm4_kernel only; launch_gemm refuses it elsewhere): the B operand of k-split term ks and inner batch b1 is block
  // b_perm[ks * nb1 + b1] of a tensor of blocks (B + b0 * b_b0 + block * b_perm_stride; b_b1 and b_ks are not used), and the term
  // is multiplied by coef[ks * nb1 + b1] - the direct form of an H_eff apply whose MPO rows are monomial (tjm_engine.hip: heff_apply).
  const int* b_perm;
  const cplx* coef;
  long b_perm_stride;
};

int launch_gemm(const GemmDesc& g, hipStream_t stream);
bool gemm4_serves(int M, int N, int K);  // the product goes to zgemm4_kernel (whole tiles; fp64 library)

// First two stages of an H_eff apply in one kernel (tjm_gemm.hip: heff_stage12_kernel): the product with the right environment over
// its non-identity channels and, as the epilogue of the same tile, the MPO stage - the intermediate T1 never leaves the chip.
//   T1[p][a][c][B] = sum_b x[p][a][b] R[b][chan(c)][B]          c over the Dr - 1 channels other than rch (R[:, rch, :] = 1)
//   out[o][a][l][B] = sum_{p,r} W[(o,l),(p,r)] in[p][a][r][B]    in[.., rch, ..] = x, the others T1;  l = lch goes to y, the rest to T2
struct HeffStage12Desc {
  const cplx* x;  long x_b0;      // [P][ca][cb]
  const cplx* R;  long r_b0;      // [cb][Dr][cb]
  const cplx* Wm;                 // [(o,l)][(p,r)] row-major, (P Dl) x (P Dr)
  cplx* T2;       long t_b0;      // [P][ca][Dl][cb]
  cplx* y;        long y_b0;      // [P][ca][cb]
  int P, ca, cb, Dl, Dr, rch, lch;
  int nb0;
  const int* ids;
  const int* active;
  int skip_t2;  // 1: only the output channel lch (-> y) is computed; the caller does not need T2 (direct form of heff_apply)
};
bool heff_stage12_fits(int P, int ca, int cb, int Dl, int Dr, int rch);
int launch_heff_stage12(const HeffStage12Desc& d, hipStream_t stream);

}  // namespace tjm
