// Batched truncated SVD split of the two-site tensor: one-sided block-Jacobi (Hestenes) on
// the fp64 matrix cores, with the reference's truncation rule applied on the device.
//
// Replaces decompositions.py:105-185 (split_two_site: scipy zgesdd + linalg.truncate,
// core/linalg/svd_utils.py:22-104) for a whole batch of trajectories at once.
//
// Method.  For "right" distribution (left tensor isometric) the kernel orthogonalises the
// columns of X = theta^H, for "left" distribution the columns of X = theta, by plane
// rotations accumulated in W (X W = Q Sigma).  The isometric output is always read from the
// accumulated unitary W and the sigma-weighted output from the rotated X, so no division by
// a singular value ever happens (a zero singular value kept by min_keep = 2 is harmless).
// Columns are processed as block pairs of 2 x 8 columns: the stacked tile [X; W] (16 columns)
// is staged in LDS, its 16 x 16 Gram matrix is formed with v_mfma_f64_16x16x4_f64, the Gram
// matrix is (nearly) diagonalised by a small cyclic Jacobi in one wavefront, and the
// resulting 16 x 16 unitary is applied to all rows of the tile with MFMAs again.  Block pairs
// of one round-robin round are independent, so one launch handles (pairs x trajectories)
// workgroups.  Trajectories whose sweep performed no rotation are flagged done and skipped.
#include "tjm_kernels.h"

namespace tjm {

namespace {

constexpr int NB = 8;       // columns per block
constexpr int TC = 2 * NB;  // columns per tile

struct JacobiArgs {
  cplx* Y;
  long y_b0;
  int rtot;     // rows of the stacked tile (multiple of 16)
  int rx;       // rows of X (top part, multiple of 16)
  int nblk;     // number of column blocks (even)
  int round;    // round-robin round
  int cs;       // LDS column pitch in doubles
  int max_inner;
  double tol2;  // squared relative tolerance
  const double* fro2;
  int* nrot;
  const int* done;
  const int* ids;
};

__device__ inline void pair_of(int nblk, int round, int p, int& I, int& J) {
  // circle method on nblk players, player nblk-1 fixed
  const int n1 = nblk - 1;
  if (p == 0) {
    I = round % n1;
    J = nblk - 1;
  } else {
    I = (round + p) % n1;
    J = (round - p + n1) % n1;
  }
  if (I > J) { int t = I; I = J; J = t; }
}

// cyclic Jacobi on the 16x16 Hermitian matrix in LDS (sA), accumulating W (sW); wave 0 only.
// Wave-synchronous: a single wavefront executes in lock-step; wave barriers order LDS traffic.
__device__ inline int inner_jacobi(cplx* sA, cplx* sW, double* sRot, int lane, int max_inner, double tol2, double floor2) {
  // sW = identity
  for (int t = lane; t < TC * TC; t += 64) sW[t] = cplx{(t / TC == t % TC) ? 1.0 : 0.0, 0.0};
  __builtin_amdgcn_wave_barrier();
  int first_count = 0;
  for (int sweep = 0; sweep < max_inner; ++sweep) {
    int sweep_count = 0;
    for (int step = 0; step < TC - 1; ++step) {
      // lanes 0..7: rotation for pair (p, q) of this step
      if (lane < TC / 2) {
        int p, q;
        pair_of(TC, step, lane, p, q);
        const double app = sA[p * TC + p].x, aqq = sA[q * TC + q].x;
        const cplx apq = sA[p * TC + q];
        const double mag2 = apq.x * apq.x + apq.y * apq.y;
        double c = 1.0, sr = 0.0, si = 0.0;
        int rot = 0;
        const double big = fmax(app, aqq);
        // rotate only if the pair is non-orthogonal at the 1e-14 level, the rotation angle is above
        // 1e-15 and the two columns are not both at the rounding-noise floor of the matrix
        if (mag2 > tol2 * app * aqq && mag2 > 1e-30 * big * big && app * aqq > floor2 && mag2 > 1e-300) {
          const double mag = sqrt(mag2);
          const double tau = (aqq - app) / (2.0 * mag);
          const double t = ((tau >= 0.0) ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
          c = 1.0 / sqrt(1.0 + t * t);
          const double s = t * c;
          sr = s * apq.x / mag;  // s * e^{i phi}
          si = s * apq.y / mag;
          rot = 1;
        }
        sRot[lane * 4 + 0] = c;
        sRot[lane * 4 + 1] = sr;
        sRot[lane * 4 + 2] = si;
        sRot[lane * 4 + 3] = (double)rot;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // lane -> column `col` (l & 15) and rows (l>>4) + 4t.  partner/role from the pairing.
      const int col = lane & 15;
      int myp = 0, pp = 0, qq = 0;
      for (int k = 0; k < TC / 2; ++k) {
        int p, q;
        pair_of(TC, step, k, p, q);
        if (p == col || q == col) { myp = k; pp = p; qq = q; }
      }
      const double c = sRot[myp * 4 + 0], sr = sRot[myp * 4 + 1], si = sRot[myp * 4 + 2];
      sweep_count += (lane < TC / 2) ? (int)sRot[lane * 4 + 3] : 0;
      const bool is_p = (col == pp);
      // ---- column phase on A and W:  y_p' = c y_p - conj(s) y_q ;  y_q' = s y_p + c y_q
      cplx na[4], nw[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int r = (lane >> 4) + 4 * t;
        const cplx ap = sA[r * TC + pp], aq = sA[r * TC + qq];
        const cplx wp = sW[r * TC + pp], wq = sW[r * TC + qq];
        if (is_p) {
          na[t] = cplx{c * ap.x - (sr * aq.x + si * aq.y), c * ap.y - (sr * aq.y - si * aq.x)};
          nw[t] = cplx{c * wp.x - (sr * wq.x + si * wq.y), c * wp.y - (sr * wq.y - si * wq.x)};
        } else {
          na[t] = cplx{(sr * ap.x - si * ap.y) + c * aq.x, (sr * ap.y + si * ap.x) + c * aq.y};
          nw[t] = cplx{(sr * wp.x - si * wp.y) + c * wq.x, (sr * wp.y + si * wp.x) + c * wq.y};
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int r = (lane >> 4) + 4 * t;
        sA[r * TC + col] = na[t];
        sW[r * TC + col] = nw[t];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // ---- row phase on A: row index `col` now plays the row role
      //   a_p' = c a_p - s a_q ; a_q' = conj(s) a_p + c a_q      (rows of J^H A)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int cc = (lane >> 4) + 4 * t;
        const cplx ap = sA[pp * TC + cc], aq = sA[qq * TC + cc];
        if (is_p) na[t] = cplx{c * ap.x - (sr * aq.x - si * aq.y), c * ap.y - (sr * aq.y + si * aq.x)};
        else      na[t] = cplx{(sr * ap.x + si * ap.y) + c * aq.x, (sr * ap.y - si * ap.x) + c * aq.y};
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int cc = (lane >> 4) + 4 * t;
        cplx v = na[t];
        if (sRot[myp * 4 + 3] != 0.0) {
          if ((col == pp && cc == qq) || (col == qq && cc == pp)) v = cplx{0.0, 0.0};
          if (cc == col) v.y = 0.0;
        }
        sA[col * TC + cc] = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    // total rotations of this sweep (lanes 0..7 hold counts)
    int tot = sweep_count;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
    if (sweep == 0) first_count = tot;
    if (tot == 0) break;
  }
  return first_count;
}

__global__ __launch_bounds__(256) void jacobi_round_kernel(JacobiArgs g) {
  extern __shared__ double smem[];
  int b = blockIdx.y;
  if (g.ids) b = g.ids[b];
  if (g.done[b]) return;
  const int cs = g.cs;
  double* sRe = smem;
  double* sIm = sRe + TC * cs;
  cplx* sA = reinterpret_cast<cplx*>(sIm + TC * cs);
  cplx* sW = sA + TC * TC;
  double* sRot = reinterpret_cast<double*>(sW + TC * TC);
  int* sFlag = reinterpret_cast<int*>(sRot + 64);

  int I, J;
  pair_of(g.nblk, g.round, blockIdx.x, I, J);
  cplx* __restrict__ Yb = g.Y + (long)b * g.y_b0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rtot = g.rtot;

  // ---- stage the 16 columns in LDS (planar re / im)
  for (int c = 0; c < TC; ++c) {
    const int gc = (c < NB) ? (I * NB + c) : (J * NB + (c - NB));
    const cplx* colp = Yb + (long)gc * rtot;
    for (int r = tid; r < rtot; r += 256) {
      cplx v = colp[r];
      sRe[c * cs + r] = v.x;
      sIm[c * cs + r] = v.y;
    }
  }
  for (int t = tid; t < TC * TC; t += 256) sA[t] = cplx{0.0, 0.0};
  __syncthreads();

  // ---- Gram matrix of the X part with MFMA: G = X^H X (16 x 16)
  {
    d4 P = {0, 0, 0, 0}, Q = {0, 0, 0, 0}, S1 = {0, 0, 0, 0}, S2 = {0, 0, 0, 0};
    const int c = lane & 15, rk = lane >> 4;
    const int nsteps = g.rx / 4;
    for (int s = wave; s < nsteps; s += 4) {
      const double xr = sRe[c * cs + 4 * s + rk];
      const double xi = sIm[c * cs + 4 * s + rk];
      P = __builtin_amdgcn_mfma_f64_16x16x4f64(xr, xr, P, 0, 0, 0);
      Q = __builtin_amdgcn_mfma_f64_16x16x4f64(xi, xi, Q, 0, 0, 0);
      S1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xr, xi, S1, 0, 0, 0);
      S2 = __builtin_amdgcn_mfma_f64_16x16x4f64(xi, xr, S2, 0, 0, 0);
    }
    // deterministic reduction over the four waves
    for (int w = 0; w < 4; ++w) {
      if (wave == w) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = (lane >> 4) + 4 * r, j = lane & 15;
          cplx v = sA[i * TC + j];
          v.x += P[r] + Q[r];
          v.y += S1[r] - S2[r];
          sA[i * TC + j] = v;
        }
      }
      __syncthreads();
    }
  }

  // ---- diagonalise the Gram matrix (wave 0), count significant rotations
  if (wave == 0) {
    const int cnt = inner_jacobi(sA, sW, sRot, lane, g.max_inner, g.tol2, 1e-60 * g.fro2[b] * g.fro2[b]);
    if (lane == 0) {
      sFlag[0] = cnt;
      if (cnt > 0) atomicAdd(&g.nrot[b], cnt);
    }
  }
  __syncthreads();
  if (sFlag[0] == 0) return;  // tile already orthogonal: nothing to update or store

  // ---- apply the 16x16 unitary to every row of the tile:  Y' = Y * W
  {
    const int j = lane & 15, kq = lane >> 4;
    double wr[4], wi[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const cplx v = sW[(4 * kk + kq) * TC + j];
      wr[kk] = v.x;
      wi[kk] = v.y;
    }
    const int nchunks = rtot / 16;
    for (int ch = wave; ch < nchunks; ch += 4) {
      const int r0 = ch * 16;
      double yr[4], yi[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        yr[kk] = sRe[(4 * kk + kq) * cs + r0 + j];  // A operand: row = l & 15, old column = 4kk + (l >> 4)
        yi[kk] = sIm[(4 * kk + kq) * cs + r0 + j];
      }
      d4 P = {0, 0, 0, 0}, Q = {0, 0, 0, 0}, S1 = {0, 0, 0, 0}, S2 = {0, 0, 0, 0};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        P = __builtin_amdgcn_mfma_f64_16x16x4f64(yr[kk], wr[kk], P, 0, 0, 0);
        Q = __builtin_amdgcn_mfma_f64_16x16x4f64(yi[kk], wi[kk], Q, 0, 0, 0);
        S1 = __builtin_amdgcn_mfma_f64_16x16x4f64(yr[kk], wi[kk], S1, 0, 0, 0);
        S2 = __builtin_amdgcn_mfma_f64_16x16x4f64(yi[kk], wr[kk], S2, 0, 0, 0);
      }
      // D layout: row = (l >> 4) + 4 r, new column = l & 15.  The chunk's rows belong to this wave only.
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + kq + 4 * r;
        sRe[j * cs + row] = P[r] - Q[r];
        sIm[j * cs + row] = S1[r] + S2[r];
      }
    }
  }
  __syncthreads();

  // ---- write the tile back
  for (int c = 0; c < TC; ++c) {
    const int gc = (c < NB) ? (I * NB + c) : (J * NB + (c - NB));
    cplx* colp = Yb + (long)gc * rtot;
    for (int r = tid; r < rtot; r += 256) colp[r] = cplx{sRe[c * cs + r], sIm[c * cs + r]};
  }
}

// Y[c][r] from theta.  dist 0: X = theta^H (columns = theta rows), dist 1: X = theta.
__global__ __launch_bounds__(256) void svd_load_kernel(const cplx* __restrict__ theta, long theta_b0, int ld, int m, int n, int dist,
                                                      cplx* __restrict__ Y, long y_b0, int ncols_pad, int rx, int rtot,
                                                      const int* ids) {
  int b = blockIdx.y;
  if (ids) b = ids[b];
  const cplx* th = theta + (long)b * theta_b0;
  cplx* Yb = Y + (long)b * y_b0;
  const long total = (long)ncols_pad * rtot;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e / rtot), r = (int)(e % rtot);
    cplx v{0.0, 0.0};
    if (r < rx) {
      if (dist == 0) {
        if (c < m && r < n) { v = th[(long)c * ld + r]; v.y = -v.y; }
      } else {
        if (r < m && c < n) v = th[(long)r * ld + c];
      }
    } else if (r - rx == c) {
      v.x = 1.0;
    }
    Yb[e] = v;
  }
}

__global__ void svd_sweep_check_kernel(int* nrot, int* done, int* n_active, int nb0, const int* ids) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nb0) return;
  const int b = ids ? ids[t] : t;
  if (!done[b]) {
    if (nrot[b] == 0) done[b] = 1;
    else atomicAdd(n_active, 1);
  }
  nrot[b] = 0;
}

__global__ void svd_reset_kernel(int* nrot, int* done, int nb0, const int* ids) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nb0) return;
  const int b = ids ? ids[t] : t;
  nrot[b] = 0;
  done[b] = 0;
}

// Column norms of the X part, descending rank sort, truncation (svd_utils.py:22-104).
__global__ __launch_bounds__(256) void svd_finish_kernel(SvdSplitDesc d, SvdWorkspace w, int ncols_pad, int rx, int rtot) {
  __shared__ double sN[512];
  __shared__ int sPerm[512];
  int b = blockIdx.x;
  if (d.ids) b = d.ids[b];
  const cplx* Yb = w.Y + (long)b * w.y_b0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c = wave; c < ncols_pad; c += 4) {
    const cplx* col = Yb + (long)c * rtot;
    double acc = 0.0;
    for (int r = lane; r < rx; r += 64) {
      cplx v = col[r];
      acc = fma(v.x, v.x, acc);
      acc = fma(v.y, v.y, acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) sN[c] = acc;
  }
  __syncthreads();
  for (int c = tid; c < ncols_pad; c += 256) {
    const double v = sN[c];
    int rank = 0;
    for (int o = 0; o < ncols_pad; ++o) {
      const double u = sN[o];
      rank += (u > v || (u == v && o < c)) ? 1 : 0;
    }
    sPerm[rank] = c;
  }
  __syncthreads();
  int* perm = w.perm + (long)b * ncols_pad;
  double* norms = w.norms + (long)b * ncols_pad;
  for (int k = tid; k < ncols_pad; k += 256) {
    perm[k] = sPerm[k];
    norms[k] = sqrt(sN[sPerm[k]]);
  }
  __syncthreads();
  if (tid == 0) {
    const int m_act = d.d * d.chiL[(long)b * d.chi_stride];
    const int n_act = d.d * d.chiR[(long)b * d.chi_stride];
    int nsv = m_act < n_act ? m_act : n_act;
    if (nsv > ncols_pad) nsv = ncols_pad;
    int keep = 0;
    if (nsv > 0) {
      auto sv = [&](int k) { return sqrt(sN[sPerm[k]]); };
      if (d.trunc_mode == 2) {  // hard_cutoff
        for (int k = 0; k < nsv; ++k) keep += (sv(k) > d.threshold) ? 1 : 0;
      } else if (d.trunc_mode == 1) {  // relative
        const double smax = sv(0);
        if (smax > 0.0)
          for (int k = 0; k < nsv; ++k) keep += ((sv(k) / smax) >= d.threshold) ? 1 : 0;
      } else if (d.trunc_mode == 0) {  // discarded_weight
        keep = nsv;
        double discard = 0.0;
        for (int idx = 0; idx < nsv; ++idx) {
          const double s = sv(nsv - 1 - idx);
          discard += s * s;
          if (discard >= d.threshold) {
            keep = nsv - idx;
            if (keep < d.min_keep) keep = d.min_keep;
            break;
          }
        }
      } else {  // relative_discarded_weight
        const double smax = sv(0);
        if (smax > 0.0) {
          double total = 0.0;
          for (int k = 0; k < nsv; ++k) { const double q = sv(k) / smax; total += q * q; }
          keep = nsv;
          double discard = 0.0;
          for (int idx = 0; idx < nsv; ++idx) {
            const double q = sv(nsv - 1 - idx) / smax;
            const double cand = discard + q * q;
            if (cand / total <= d.threshold) { discard = cand; keep = nsv - idx - 1; }
            else break;
          }
        }
      }
      if (d.max_bond > 0 && keep > d.max_bond) keep = d.max_bond;
      if (keep < d.min_keep) keep = d.min_keep;
      if (keep > nsv) keep = nsv;
    }
    d.chiM[(long)b * d.chi_stride] = keep;
  }
  if (d.spectrum) {
    for (int k = tid; k < d.spec_ld; k += 256) d.spectrum[(long)b * d.spec_ld + k] = (k < ncols_pad) ? sqrt(sN[sPerm[k]]) : 0.0;
  }
}

// left[b][s][a][k], right[b][t][k][c] from the rotated tile (zero beyond `keep`).
__global__ __launch_bounds__(256) void svd_write_kernel(SvdSplitDesc d, SvdWorkspace w, int ncols_pad, int rx, int rtot) {
  int b = blockIdx.y;
  if (d.ids) b = d.ids[b];
  const cplx* Yb = w.Y + (long)b * w.y_b0;
  const int* perm = w.perm + (long)b * ncols_pad;
  const int keep = d.chiM[(long)b * d.chi_stride];
  const long nl = (long)d.d * d.capL * d.capM, nr = (long)d.d * d.capM * d.capR;
  cplx* L = d.left + (long)b * d.left_b0;
  cplx* R = d.right + (long)b * d.right_b0;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < nl + nr; e += (long)gridDim.x * blockDim.x) {
    if (e < nl) {
      const int k = (int)(e % d.capM);
      const int row = (int)(e / d.capM);  // (s, a) -> s * capL + a
      cplx v{0.0, 0.0};
      if (k < keep) {
        const long col = perm[k];
        v = (d.distribution == 0) ? Yb[col * rtot + rx + row] : Yb[col * rtot + row];
      }
      L[e] = v;
    } else {
      const long f = e - nl;
      const int c = (int)(f % d.capR);
      const int k = (int)((f / d.capR) % d.capM);
      const int t = (int)(f / ((long)d.capR * d.capM));
      cplx v{0.0, 0.0};
      if (k < keep) {
        const long col = perm[k];
        const int row = t * d.capR + c;
        v = (d.distribution == 0) ? Yb[col * rtot + row] : Yb[col * rtot + rx + row];
        v.y = -v.y;
      }
      R[f] = v;
    }
  }
}

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

}  // namespace

size_t svd_workspace_bytes(int max_dim, int B) {
  const int p = round_up(max_dim, 16);
  size_t y = (size_t)B * p * (2 * p) * sizeof(cplx);
  size_t small = (size_t)B * p * (sizeof(double) + sizeof(int)) + (size_t)B * 2 * sizeof(int) + 64;
  return y + small + 1024;
}

int svd_split(const SvdSplitDesc& d, const SvdWorkspace& w, hipStream_t s, int* sweeps_out) {
  if (d.nb0 <= 0) return TJM_OK;
  const int ncols = d.distribution == 0 ? d.m : d.n;
  const int rxr = d.distribution == 0 ? d.n : d.m;
  const int ncols_pad = round_up(ncols, 16);
  const int rx = round_up(rxr, 16);
  const int rtot = rx + ncols_pad;
  if (rtot > 512 || ncols_pad > 512) return TJM_ERR_NOT_IMPLEMENTED;  // LDS-resident tile: d*chi <= 256
  if ((long)ncols_pad * rtot > w.y_b0) return TJM_ERR_WORKSPACE;
  const int cs = rtot + 2;
  const size_t lds = (size_t)2 * TC * cs * sizeof(double) + 2 * TC * TC * sizeof(cplx) + 64 * sizeof(double) + 16;
  static bool attr_set = false;
  if (!attr_set) {
    TJM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_round_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const int tb = (d.nb0 + 255) / 256;
  hipLaunchKernelGGL(svd_reset_kernel, dim3(tb), dim3(256), 0, s, w.nrot, w.done, d.nb0, d.ids);
  if (d.ld_theta != d.n) return TJM_ERR_ARG;
  {
    int rc = launch_normsq(d.theta, d.theta_b0, (long)d.m * d.n, w.fro2, d.nb0, d.ids, s);
    if (rc != TJM_OK) return rc;
  }
  {
    const long total = (long)ncols_pad * rtot;
    int gx = (int)((total + 1023) / 1024);
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(svd_load_kernel, dim3(gx, d.nb0), dim3(256), 0, s, d.theta, d.theta_b0, d.ld_theta, d.m, d.n,
                       d.distribution, w.Y, w.y_b0, ncols_pad, rx, rtot, d.ids);
  }
  JacobiArgs g;
  g.Y = w.Y;
  g.y_b0 = w.y_b0;
  g.rtot = rtot;
  g.rx = rx;
  g.nblk = ncols_pad / NB;
  g.cs = cs;
  g.max_inner = 3;
  g.tol2 = 1e-28;  // relative off-diagonal tolerance 1e-14
  g.nrot = w.nrot;
  g.fro2 = w.fro2;
  g.done = w.done;
  g.ids = d.ids;
  const int nrounds = g.nblk - 1;
  const int npairs = g.nblk / 2;
  const int max_sweeps = 30;
  int sweep = 0;
  for (; sweep < max_sweeps; ++sweep) {
    for (int r = 0; r < nrounds; ++r) {
      g.round = r;
      hipLaunchKernelGGL(jacobi_round_kernel, dim3(npairs, d.nb0), dim3(256), lds, s, g);
    }
    TJM_HIP_CHECK(hipMemsetAsync(w.n_active, 0, sizeof(int), s));
    hipLaunchKernelGGL(svd_sweep_check_kernel, dim3(tb), dim3(256), 0, s, w.nrot, w.done, w.n_active, d.nb0, d.ids);
    TJM_HIP_CHECK(hipMemcpyAsync(w.h_pinned, w.n_active, sizeof(int), hipMemcpyDeviceToHost, s));
    TJM_HIP_CHECK(hipStreamSynchronize(s));
    if (*w.h_pinned == 0) { ++sweep; break; }
  }
  if (sweeps_out) *sweeps_out = sweep;
  hipLaunchKernelGGL(svd_finish_kernel, dim3(d.nb0), dim3(256), 0, s, d, w, ncols_pad, rx, rtot);
  {
    const long total = (long)d.d * d.capL * d.capM + (long)d.d * d.capM * d.capR;
    int gx = (int)((total + 1023) / 1024);
    if (gx > 256) gx = 256;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(svd_write_kernel, dim3(gx, d.nb0), dim3(256), 0, s, d, w, ncols_pad, rx, rtot);
  }
  TJM_HIP_CHECK(hipGetLastError());
  return (sweep >= max_sweeps && *w.h_pinned != 0) ? TJM_ERR_NUMERIC : TJM_OK;
}

}  // namespace tjm
