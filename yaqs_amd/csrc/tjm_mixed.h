// Bridge between the two arithmetic types inside libtjm_hip.so (fp64 build only).
//
// The two-site split of the TJM sweep (decompositions.py:105-185: zgesdd + truncate) spends its time in fp64 Jacobi sweeps.  The
// mixed-precision split (tjm_svd.hip: svd_split_mixed) lets the complex64 code of this same source tree find an APPROXIMATE singular
// basis first - at the packed-fp32 rotation rate, 2.2 x the fp64 one on gfx950 - and spends fp64 work only on making that basis
// exactly unitary (one polar step, three GEMMs on the matrix cores) and on the two or three Jacobi sweeps that take a 1e-6-orthogonal
// matrix to 1e-13.  The result is an fp64 one-sided Jacobi SVD of theta times an exactly unitary matrix: the same singular values,
// the same isometric factor, to the same tolerance as the all-fp64 path.
//
// The complex64 code is the ordinary -DTJM_F32 build of tjm_gemm.hip / tjm_qr.hip / tjm_svd.hip compiled a second time into the
// namespace tjm32 (-Dtjm=tjm32: every `namespace tjm` of the sources becomes tjm32, the device symbols differ by their mangled
// names) and linked into libtjm_hip.so next to the fp64 objects.  This header is the only thing both sides see: plain types, no
// `real`, no `cplx`.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace tjm32 {

struct MixedBasisDesc {
  int N;          // theta is N x N (N = d * bond capacity, a multiple of 64)
  int d;          // physical dimension: rows of theta are (s, a), columns (t, c)
  int dist;       // the basis is the left singular basis of Z' = theta (0) or theta^H (1), rows of Z' bond-major (a * d + s / c * d + t)
  int nb0;        // trajectories (slots 0 ... nb0 - 1 of the workspace)
  int max_sweeps; // cap on the complex64 sweeps; stopping there is fine, fp64 finishes the job
  double stop_fraction;  // the complex64 iteration ends after a sweep that rotated less than this fraction of the pairs (0: run to convergence)
  int* h_pinned;  // pinned host ints (>= 8) for the sweep loop's convergence reads
};

// Workspace of the complex64 phase for matrices up to max_dim x max_dim and B trajectories.  Its head is the complex64 copy of theta
// that the caller fills: [B][max_dim * max_dim] float2, row-major N x N per trajectory with batch stride max_dim * max_dim.
size_t mixed_workspace_bytes(int max_dim, int B);

// All N left singular vectors of Z' (approximate: complex64 arithmetic, orthonormal to ~1e-6), column-major N x N per trajectory;
// *basis / *basis_b0 (elements of float2) point into the workspace.  Exactly zero singular directions (zero padding of the bond) are
// completed with unit vectors, so the basis is square and (approximately) unitary whatever the rank.
int mixed_left_basis(const MixedBasisDesc& m, void* ws, size_t ws_bytes, int max_dim, int B, hipStream_t s, const void** basis,
                     long* basis_b0, int* sweeps_out);

// out = in x in for N x N row-major complex64 matrices (hermitian != 0: the product is Hermitian, only the tiles on and above the
// diagonal are computed): the squares of the refinement rounds of the fp64 phase that need three digits only (C^2 / 2 next to
// I + C with |C| <= 0.01, corrected for exactly in the following round; W^2 / 2 with |W| <= 1e-4).  The complex64 GEMM runs at
// twice the fp64 rate.  in / out (device, float2, batch stride *b0 elements) are buffers of the workspace that are idle after
// mixed_left_basis has returned its result to the caller; the caller fills `in`.
int mixed_square_buffers(void* ws, size_t ws_bytes, int max_dim, int B, void** in, const void** out, long* b0);
int mixed_square(void* ws, size_t ws_bytes, int max_dim, int B, int N, int nb0, int hermitian, hipStream_t s);

// Counters and the launch sampler of the complex64 Jacobi kernels (the tjm32 instances of the functions of the same name in
// tjm_kernels.h; defined by the tjm32 compilation of tjm_svd.hip)
void jacobi_work_get(double* out4, bool reset);
void qr_profile_enable(int every);
void qr_profile_get(double* out5);
void profile_enable(int every);
void profile_get(double* total_ms, double* total_bytes, long* samples);

}  // namespace tjm32
