// complex64 half of the mixed-precision two-site split (see tjm_mixed.h).  Compiled ONLY as
//   hipcc -DTJM_F32 -Dtjm=tjm32 ...
// so that `namespace tjm` of the shared headers is tjm32 here and the functions below call the complex64 instantiation of the QR
// preconditioner and the tiled Jacobi (tjm_qr.hip, tjm_svd.hip) that is linked into the fp64 library under that namespace.
#include "tjm_kernels.h"
#include "tjm_mixed.h"

#include <cstring>

#ifndef TJM_F32
#error "tjm_mixed32.hip is the complex64 side of the bridge: compile with -DTJM_F32 -Dtjm=tjm32"
#endif

namespace tjm32 {

namespace {

struct Layout {
  SvdWorkspace w;
  QrWorkspace q;
  int* chi_all;  // [B] = N (every singular value is "kept": the basis is square)
  int* chi_out;  // [B]
  size_t bytes;
};

Layout carve(char* base, int max_dim, int B) {
  Layout l;
  size_t off = 0;
  auto take = [&](size_t n) { char* p = base ? base + off : nullptr; off += (n + 255) / 256 * 256; return p; };
  take((size_t)B * max_dim * max_dim * sizeof(cplx));  // theta in complex64, filled by the caller
  {
    const size_t n = svd_workspace_bytes(max_dim, B);
    char* p = take(n);
    svd_carve(l.w, p, max_dim, B);
  }
  {
    const size_t n = qr_workspace_bytes(max_dim, B);
    char* p = take(n);
    qr_carve(l.q, p, max_dim, B);
  }
  l.chi_all = reinterpret_cast<int*>(take((size_t)B * sizeof(int)));
  l.chi_out = reinterpret_cast<int*>(take((size_t)B * sizeof(int)));
  l.bytes = off + 4096;
  return l;
}

__global__ void fill_int_kernel(int* p, int v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

}  // namespace

size_t mixed_workspace_bytes(int max_dim, int B) { return carve(nullptr, max_dim, B).bytes; }

int mixed_left_basis(const MixedBasisDesc& m, void* ws, size_t ws_bytes, int max_dim, int B, hipStream_t s, const void** basis,
                     long* basis_b0, int* sweeps_out) {
  if (m.nb0 <= 0) return TJM_OK;
  if (m.N > max_dim || m.nb0 > B || m.N % 64 != 0 || m.N % m.d != 0) return TJM_ERR_ARG;
  Layout l = carve(static_cast<char*>(ws), max_dim, B);
  if (ws_bytes < l.bytes) return TJM_ERR_WORKSPACE;
  l.w.h_pinned = m.h_pinned;
  const int N = m.N;
  const cplx* theta = static_cast<const cplx*>(ws);
  const long th_b0 = (long)max_dim * max_dim;
  const QrWorkspace& q = l.q;
  const QrWorkspace q2 = q.second();
  int rc;
  hipLaunchKernelGGL(fill_int_kernel, dim3((m.nb0 + 255) / 256), dim3(256), 0, s, l.chi_all, N, m.nb0);
  // Z' (columns sorted by norm) = Q R, R^H = Q1 R1, Jacobi on X = R1^H: the doubly preconditioned direct variant of svd_split_qr
  if ((rc = qr_prepare(theta, th_b0, N, N, m.dist, m.d, q, m.nb0, nullptr, s)) != TJM_OK) return rc;
  if ((rc = qr_factor(q, N, N, m.nb0, nullptr, s)) != TJM_OK) return rc;
  if ((rc = qr_adjoint_triangle(q, N, m.nb0, nullptr, s)) != TJM_OK) return rc;
  if ((rc = qr_factor(q2, N, N, m.nb0, nullptr, s)) != TJM_OK) return rc;
  JacobiSource src;
  src.src = q2.Z; src.src_b0 = q2.z_b0; src.rx = N; src.ncols = N; src.conj = 1; src.tri = 1;
  src.r_n0 = N; src.s_r1 = 0; src.s_r0 = N; src.c_n0 = N; src.s_c1 = 0; src.s_c0 = 1;
  src.nb0 = m.nb0; src.ids = nullptr;
  TruncSpec tr;  // keep everything: hard cut-off below zero, min_keep = N
  tr.trunc_mode = 2; tr.threshold = -1.0f; tr.max_bond = 0; tr.min_keep = N; tr.cap = 0; tr.overflow = nullptr;
  tr.chiA = l.chi_all; tr.mulA = 1; tr.chiB = l.chi_all; tr.mulB = 1; tr.chiOut = l.chi_out; tr.chi_stride = 1;
  tr.spectrum = nullptr; tr.spec_ld = 0;
  JacobiOpts op;
  op.max_sweeps = m.max_sweeps > 0 ? m.max_sweeps : 12;
  op.allow_unconverged = true;
  op.stop_fraction = m.stop_fraction;
  // every non-zero column is rotated: a column at the fp32 rounding floor is noise, but noise that has been orthogonalised against
  // the rest is what the polar step of the fp64 side can make exactly unitary (an unrotated one is not); the sweep cap bounds the cost
  op.floor_scale = 0.0f;  // (the caller scales theta so that ||theta||_F ~ 2^24: tiny columns stay far above the fp32 underflow range)
  op.quad = true;
  JacobiShape sh;
  if ((rc = jacobi_solve(src, tr, l.w, s, &sh, sweeps_out, false, &op)) != TJM_OK) return rc;
  ExtractDesc xy;  // normalised columns of Y (unit vectors for the structurally zero ones) into Z, N x N column-major
  xy.out = q.Z; xy.out_b0 = q.z_b0; xy.n_k = N; xy.o_k = N; xy.n_r1 = 1; xy.n_r0 = N;
  xy.o_r1 = 0; xy.o_r0 = 1; xy.row_off = 0; xy.conj = 0; xy.scale_mode = 5;
  if ((rc = svd_extract(xy, l.w, sh, l.chi_out, 1, m.nb0, nullptr, s)) != TJM_OK) return rc;
  if ((rc = qr_apply_q(q, N, N, q.Z, q.z_b0, N, m.nb0, nullptr, s)) != TJM_OK) return rc;  // left singular basis of Z' = Q Ytilde
  *basis = q.Z;
  *basis_b0 = q.z_b0;
  return TJM_OK;
}

int mixed_square_buffers(void* ws, size_t ws_bytes, int max_dim, int B, void** in, const void** out, long* b0) {
  Layout l = carve(static_cast<char*>(ws), max_dim, B);
  if (ws_bytes < l.bytes) return TJM_ERR_WORKSPACE;
  *in = ws;  // the head (theta in complex64 during the factorisation)
  *out = l.q.Z2;
  *b0 = (long)max_dim * max_dim;
  return TJM_OK;
}

int mixed_square(void* ws, size_t ws_bytes, int max_dim, int B, int N, int nb0, int hermitian, hipStream_t s) {
  if (nb0 <= 0) return TJM_OK;
  Layout l = carve(static_cast<char*>(ws), max_dim, B);
  if (ws_bytes < l.bytes || N > max_dim || nb0 > B) return TJM_ERR_WORKSPACE;
  GemmDesc g;
  memset(&g, 0, sizeof(g));
  g.nks = 1; g.nb0 = nb0; g.nb1 = 1; g.nb2 = 1; g.M = N; g.N = N; g.K = N;
  g.A = static_cast<const cplx*>(ws); g.a_rs = N; g.a_cs = 1; g.a_b0 = (long)max_dim * max_dim;
  g.B = g.A; g.b_rs = N; g.b_cs = 1; g.b_b0 = g.a_b0;
  g.C = l.q.Z2; g.c_rs = N; g.c_b0 = l.q.z_b0;
  g.hermitian = hermitian ? 1 : 0;
  return launch_gemm(g, s);
}

}  // namespace tjm32
