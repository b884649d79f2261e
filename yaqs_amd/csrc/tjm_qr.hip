// Batched blocked Householder QR (compact WY) used to precondition the Jacobi SVD of the two-site split.
//
// Why: one-sided Jacobi on theta (or theta^H) of a time-evolved two-site tensor needs 16-19 sweeps because the
// singular spectrum is graded over many decades and both singular bases are dense.  After Z = Q R the Jacobi
// iteration on R^H starts from nearly orthogonal, norm-ordered columns and converges in about half the sweeps
// (Drmac-Veselic preconditioning).  The isometric factor is recovered exactly as Q * W (Q: product of Householder
// reflectors, W: accumulated plane rotations), so no division by a singular value appears anywhere.
//
// Structure: panels of 16 columns.  qr_panel_kernel factors one panel in LDS (one workgroup per trajectory),
// builds the triangular T of the compact WY form and writes the reflector block V with explicit zeros / unit
// diagonal; the trailing update and the application of Q are three launches of the batched MFMA zgemm each.
#include <cstdlib>
#include <cstring>

#include <atomic>
#include <mutex>
#include <vector>

#include "tjm_kernels.h"

namespace tjm {

namespace {

constexpr int PW = 16;  // panel width

__device__ inline real block_sum256(real v, real* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

__device__ inline cplx wave_csum(cplx v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    v.x += __shfl_xor(v.x, o, 64);
    v.y += __shfl_xor(v.y, o, 64);
  }
  return v;
}

template <int CTRL>
__device__ inline real dpp_pull(real v) {
#ifdef TJM_F32
  return tjm_dpp<CTRL>(v);
#else
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
#endif
}
__device__ inline real lane_value(real v, int lane) {
#ifdef TJM_F32
  return tjm_readlane(v, lane);
#else
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
#endif
}
// wavefront all-reduce: DPP butterfly inside rows of 16 lanes, then the four row totals via v_readlane
__device__ inline real wsum(real v) {
  v += dpp_pull<0xB1>(v);
  v += dpp_pull<0x4E>(v);
  v += dpp_pull<0x141>(v);
  v += dpp_pull<0x140>(v);
  return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}

// Factor panel columns [k0, k0 + pw) of A (column-major, zr x zc, leading dimension zr), rows k0 .. zr-1.
// ONE wavefront per trajectory with the panel in LDS: a single wavefront runs in lock-step, so the 16 sequential
// Householder steps need wavefront reductions only (no workgroup barrier), and the loops stay rolled (small code).
__global__ __launch_bounds__(64) void qr_panel_kernel(cplx* __restrict__ A, long a_b0, int zr, int k0, int pw, cplx* __restrict__ Vb,
                                                     long v_b0, cplx* __restrict__ Tb, long t_b0, int panel, const int* ids) {
  extern __shared__ real smem[];
  int b = blockIdx.x;
  if (ids) b = ids[b];
  const int lane = threadIdx.x;
  const int mp = zr - k0;  // panel rows, local row r = global row - k0
  cplx* P = reinterpret_cast<cplx*>(smem);  // [PW][mp]
  cplx* sG = P + PW * mp;                   // [PW][PW]
  cplx* sT = sG + PW * PW;                  // [PW][PW]
  cplx* sTau = sT + PW * PW;                // [PW]
  real* sBeta = reinterpret_cast<real*>(sTau + PW);
  cplx* Ab = A + (long)b * a_b0;
  for (int c = 0; c < pw; ++c)
    for (int r = lane; r < mp; r += 64) P[c * mp + r] = Ab[(long)(k0 + c) * zr + k0 + r];
  for (int t = lane; t < PW * PW; t += 64) { sG[t] = cplx{0.0, 0.0}; sT[t] = cplx{0.0, 0.0}; }
  if (lane < PW) { sTau[lane] = cplx{0.0, 0.0}; sBeta[lane] = 0.0; }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int j = 0; j < pw; ++j) {
    // ---- zlarfg on column j, rows j .. mp-1
    cplx* pj = P + j * mp;
    real acc = 0.0;
    for (int r = j + 1 + lane; r < mp; r += 64) {
      const cplx v = pj[r];
      acc = fma(v.x, v.x, fma(v.y, v.y, acc));
    }
    const real xn2 = wsum(acc);
    const cplx alpha = (j < mp) ? pj[j] : cplx{0.0, 0.0};
    cplx tau{0.0, 0.0}, scale{0.0, 0.0};
    real bt = alpha.x;
    if (j < mp && (xn2 > 0.0 || alpha.y != 0.0)) {
      const real an = sqrt(alpha.x * alpha.x + alpha.y * alpha.y + xn2);
      bt = (alpha.x >= 0.0) ? -an : an;
      tau = cplx{(bt - alpha.x) / bt, -alpha.y / bt};
      const cplx dnm{alpha.x - bt, alpha.y};
      const real d2 = dnm.x * dnm.x + dnm.y * dnm.y;
      scale = cplx{dnm.x / d2, -dnm.y / d2};
    }
    __builtin_amdgcn_wave_barrier();
    for (int r = j + 1 + lane; r < mp; r += 64) pj[r] = cmul(pj[r], scale);
    if (lane == 0) {
      if (j < mp) pj[j] = cplx{1.0, 0.0};
      sTau[j] = tau;
      sBeta[j] = bt;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- H^H = I - conj(tau) v v^H on the remaining panel columns
    if (tau.x != 0.0 || tau.y != 0.0) {
      // all dot products v^H a_c (c > j) first, reduced together (the reductions pipeline), then the rank-1 update
      cplx w[PW];
#pragma unroll
      for (int c = 0; c < PW; ++c) w[c] = cplx{0.0, 0.0};
      for (int r = j + lane; r < mp; r += 64) {
        const cplx vv = cconj(pj[r]);
#pragma unroll
        for (int c = 0; c < PW; ++c)
          if (c > j && c < pw) cfma(w[c], vv, P[c * mp + r]);
      }
#pragma unroll
      for (int c = 0; c < PW; ++c) {
        w[c].x = wsum(w[c].x);
        w[c].y = wsum(w[c].y);
        w[c] = cmul(cconj(tau), w[c]);
      }
      for (int r = j + lane; r < mp; r += 64) {
        const cplx vv = pj[r];
#pragma unroll
        for (int c = 0; c < PW; ++c)
          if (c > j && c < pw) {
            cplx x = P[c * mp + r];
            x.x -= w[c].x * vv.x - w[c].y * vv.y;
            x.y -= w[c].x * vv.y + w[c].y * vv.x;
            P[c * mp + r] = x;
          }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // ---- G[l][j] = V_l^H V_j (l < j); column c holds v_c below row c, 1 on it (R entries above are not part of V)
  for (int j = 1; j < pw; ++j) {
    cplx w[PW];
#pragma unroll
    for (int l = 0; l < PW; ++l) w[l] = cplx{0.0, 0.0};
    for (int r = j + lane; r < mp; r += 64) {
      const cplx vj = P[j * mp + r];
#pragma unroll
      for (int l = 0; l < PW; ++l)
        if (l < j) cfma(w[l], cconj(P[l * mp + r]), vj);
    }
#pragma unroll
    for (int l = 0; l < PW; ++l) {
      w[l].x = wsum(w[l].x);
      w[l].y = wsum(w[l].y);
      if (lane == 0 && l < j) sG[l * PW + j] = w[l];
    }
  }
  if (lane < pw) sT[lane * PW + lane] = sTau[lane];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // ---- T (zlarft forward / columnwise); lane i builds row i of column j
  for (int j = 1; j < pw; ++j) {
    const cplx tj = sT[j * PW + j];
    if (lane < j) {
      cplx accv{0.0, 0.0};
      for (int l = lane; l < j; ++l) cfma(accv, sT[lane * PW + l], sG[l * PW + j]);
      sT[lane * PW + j] = cplx{-(tj.x * accv.x - tj.y * accv.y), -(tj.x * accv.y + tj.y * accv.x)};
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // ---- write back: R part / reflectors into A, explicit V block (zeros above, unit diagonal), T
  cplx* Vp = Vb + (long)b * v_b0 + (long)panel * PW * zr;
  for (int c = 0; c < PW; ++c) {
    for (int r = lane; r < mp; r += 64) {
      cplx v{0.0, 0.0};
      if (c < pw) {
        cplx a = P[c * mp + r];
        if (r == c) a = cplx{sBeta[c], 0.0};
        Ab[(long)(k0 + c) * zr + k0 + r] = a;
        if (r == c) v = cplx{1.0, 0.0};
        else if (r > c) v = P[c * mp + r];
      }
      Vp[(long)c * zr + k0 + r] = v;
    }
    for (int gr = lane; gr < k0; gr += 64) Vp[(long)c * zr + gr] = cplx{0.0, 0.0};
  }
  cplx* Tp = Tb + (long)b * t_b0 + (long)panel * PW * PW;
  for (int t = lane; t < PW * PW; t += 64) Tp[t] = sT[t];
}

// Row-resident panel factorisation: 256 threads, thread t keeps rows t, t + 256, ... of the 16 panel columns in
// registers.  One step needs ONE workgroup reduction: with a_j the not yet scaled column j, every inner product of the
// step follows from raw sums over the rows below j,
//   xn2 = sum |a_j|^2,   d_l = sum conj(v_l) a_j  (l < j, for T),   e_c = sum conj(a_j) a_c  (c > j, for the update),
// because v_j = scale * a_j below the diagonal: v_j^H a_c = conj(scale) e_c + a_c[j],  v_l^H v_j = scale d_l + conj(v_l[j]).
// The 31 partial sums are transposed through LDS ([value][thread]), 8 threads add 32 entries each per value, a DPP
// butterfly joins them.  Row j of the panel (alpha, a_c[j], v_l[j]) is broadcast through LDS by its owner.
constexpr int NRED = 2 * PW;          // slot 0: xn2, slots 1..15: Re, 17..31: Im of the 15 complex sums (slot 16 unused)

// NT = 256 threads for panels of up to 512 rows; NT = 512 (RPT = 2) for up to 1024 rows (bonds up to 512): the 32 sums are then
// joined by 16 threads each
template <int RPT, int NT = 256>
__global__ __launch_bounds__(NT) void qr_panel_rows_kernel(cplx* __restrict__ A, long a_b0, int zr, int k0, int pw, cplx* __restrict__ Vb,
                                                           long v_b0, cplx* __restrict__ Tb, long t_b0, int panel, const int* ids) {
  extern __shared__ real smem[];
  int b = blockIdx.x;
  if (ids) b = ids[b];
  const int tid = threadIdx.x;
  const int mp = zr - k0;
  constexpr int RED_PITCH = NT + 1;
  constexpr int SEGS = NT / 32;                                   // threads that join one of the 32 sums
  real* sPart = smem;                                           // [NRED][RED_PITCH]
  real* sRed = sPart + NRED * RED_PITCH;                        // [2][NRED]
  cplx* sRow = reinterpret_cast<cplx*>(sRed + 2 * NRED);          // [2][PW]
  cplx* sG = sRow + 2 * PW;                                       // [PW][PW]
  cplx* sT = sG + PW * PW;                                        // [PW][PW]
  cplx* sTau = sT + PW * PW;                                      // [PW]
  real* sBeta = reinterpret_cast<real*>(sTau + PW);           // [PW]
  cplx* Ab = A + (long)b * a_b0;

  cplx P[PW][RPT];
#pragma unroll
  for (int c = 0; c < PW; ++c)
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      const int r = tid + NT * q;
      P[c][q] = (c < pw && r < mp) ? Ab[(long)(k0 + c) * zr + k0 + r] : cplx{0.0, 0.0};
    }
  for (int t = tid; t < PW * PW; t += NT) { sG[t] = cplx{0.0, 0.0}; sT[t] = cplx{0.0, 0.0}; }
  if (tid < PW) { sTau[tid] = cplx{0.0, 0.0}; sBeta[tid] = 0.0; }

#pragma unroll
  for (int j = 0; j < PW; ++j) {
    if (j < pw && j < mp) {   // uniform
      const int buf = j & 1;
      // ---- partial sums over this thread's rows below j
      real part[NRED];
#pragma unroll
      for (int v = 0; v < NRED; ++v) part[v] = 0.0;
#pragma unroll
      for (int q = 0; q < RPT; ++q) {
        const int r = tid + NT * q;
        if (r > j && r < mp) {
          const cplx aj = P[j][q];
          part[0] = fma(aj.x, aj.x, fma(aj.y, aj.y, part[0]));
#pragma unroll
          for (int c = 0; c < PW; ++c) {
            if (c == j) continue;
            const cplx x = P[c][q];
            const int slot = (c < j) ? c + 1 : c;   // 1..15
            if (c < j) {  // conj(v_l) * a_j
              part[slot] = fma(x.x, aj.x, fma(x.y, aj.y, part[slot]));
              part[slot + PW] = fma(x.x, aj.y, fma(-x.y, aj.x, part[slot + PW]));
            } else {      // conj(a_j) * a_c
              part[slot] = fma(aj.x, x.x, fma(aj.y, x.y, part[slot]));
              part[slot + PW] = fma(aj.x, x.y, fma(-aj.y, x.x, part[slot + PW]));
            }
          }
        }
      }
#pragma unroll
      for (int v = 0; v < NRED; ++v) sPart[v * RED_PITCH + tid] = part[v];
      if (tid == j) {  // row j < 16 of the panel lives in thread j, slot 0
#pragma unroll
        for (int c = 0; c < PW; ++c) sRow[buf * PW + c] = P[c][0];
      }
      __syncthreads();
      {
        const int v = tid / SEGS, seg = tid % SEGS;
        const real* src = sPart + v * RED_PITCH + seg * 32;
        real acc = 0.0;
#pragma unroll
        for (int i = 0; i < 32; ++i) acc += src[(i + 4 * seg) & 31];
        acc += dpp_pull<0xB1>(acc);
        acc += dpp_pull<0x4E>(acc);
        acc += dpp_pull<0x141>(acc);
        if (SEGS == 16) acc += dpp_pull<0x140>(acc);  // row_mirror joins the two halves of a 16-lane row
        if (seg == 0) sRed[buf * NRED + v] = acc;
      }
      __syncthreads();
      // ---- zlarfg (every thread, redundantly)
      const real xn2 = sRed[buf * NRED];
      const cplx alpha = sRow[buf * PW + j];
      cplx tau{0.0, 0.0}, scale{0.0, 0.0};
      real bt = alpha.x;
      if (xn2 > 0.0 || alpha.y != 0.0) {
        const real an = sqrt(alpha.x * alpha.x + alpha.y * alpha.y + xn2);
        bt = (alpha.x >= 0.0) ? -an : an;
        tau = cplx{(bt - alpha.x) / bt, -alpha.y / bt};
        const cplx dnm{alpha.x - bt, alpha.y};
        const real d2 = dnm.x * dnm.x + dnm.y * dnm.y;
        scale = cplx{dnm.x / d2, -dnm.y / d2};
      }
      if (tid == 0) { sTau[j] = tau; sBeta[j] = bt; }
      if (tid < j) {  // Gram entry for T: v_l^H v_j = scale * d_l + conj(v_l[j])
        const cplx dl{sRed[buf * NRED + tid + 1], sRed[buf * NRED + tid + 1 + PW]};
        const cplx vlj = sRow[buf * PW + tid];
        cplx gv = cmul(scale, dl);
        gv.x += vlj.x;
        gv.y -= vlj.y;
        sG[tid * PW + j] = gv;
      }
      // ---- v_j, then H^H = I - conj(tau) v v^H on the remaining panel columns
      cplx wv[PW];
#pragma unroll
      for (int c = 0; c < PW; ++c) {
        if (c > j) {
          const cplx e{sRed[buf * NRED + c], sRed[buf * NRED + c + PW]};
          const cplx acj = sRow[buf * PW + c];
          cplx t = cmul(cconj(scale), e);
          t.x += acj.x;
          t.y += acj.y;
          wv[c] = cmul(cconj(tau), t);
        }
      }
#pragma unroll
      for (int q = 0; q < RPT; ++q) {
        const int r = tid + NT * q;
        if (r >= j && r < mp) {
          const cplx vv = (r == j) ? cplx{1.0, 0.0} : cmul(P[j][q], scale);
          P[j][q] = vv;
#pragma unroll
          for (int c = 0; c < PW; ++c) {
            if (c > j) {
              P[c][q].x -= wv[c].x * vv.x - wv[c].y * vv.y;
              P[c][q].y -= wv[c].x * vv.y + wv[c].y * vv.x;
            }
          }
        }
      }
    }
  }
  __syncthreads();
  if (tid < pw) sT[tid * PW + tid] = sTau[tid];
  __syncthreads();
  // ---- T (zlarft forward / columnwise); thread i builds row i of column j
  for (int j = 1; j < pw; ++j) {
    const cplx tj = sT[j * PW + j];
    if (tid < j) {
      cplx accv{0.0, 0.0};
      for (int l = tid; l < j; ++l) cfma(accv, sT[tid * PW + l], sG[l * PW + j]);
      sT[tid * PW + j] = cplx{-(tj.x * accv.x - tj.y * accv.y), -(tj.x * accv.y + tj.y * accv.x)};
    }
    __syncthreads();
  }
  // ---- write back: R part / reflectors into A, explicit V block (zeros above, unit diagonal), T
  cplx* Vp = Vb + (long)b * v_b0 + (long)panel * PW * zr;
#pragma unroll
  for (int c = 0; c < PW; ++c) {
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      const int r = tid + NT * q;
      if (r < mp) {
        cplx v{0.0, 0.0};
        if (c < pw) {
          cplx a = P[c][q];
          if (r == c) a = cplx{sBeta[c], 0.0};
          Ab[(long)(k0 + c) * zr + k0 + r] = a;
          if (r == c) v = cplx{1.0, 0.0};
          else if (r > c) v = P[c][q];
        }
        Vp[(long)c * zr + k0 + r] = v;
      }
    }
    for (int gr = tid; gr < k0; gr += NT) Vp[(long)c * zr + gr] = cplx{0.0, 0.0};
  }
  cplx* Tp = Tb + (long)b * t_b0 + (long)panel * PW * PW;
  for (int t = tid; t < PW * PW; t += NT) Tp[t] = sT[t];
}

// Fused block reflector on a chunk of 16 columns of C (column-major, leading dimension zr), fp64 MFMA:
//   C_chunk <- C_chunk - V op(T) (V^H C_chunk),   op = T^H (factorisation) or T (Q * C)
__global__ __launch_bounds__(256) void qr_block_apply_kernel(const cplx* __restrict__ Vb, long v_b0, const cplx* __restrict__ Tb, long t_b0, int panel,
                                                            int zr, int t_herm, cplx* __restrict__ C, long c_b0, int col0, int nc,
                                                            const int* ids) {
  __shared__ cplx sW1[PW * PW];   // (i, c)
  __shared__ cplx sW2[PW * PW];
  __shared__ cplx sTm[PW * PW];
  int b = blockIdx.y;
  if (ids) b = ids[b];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = col0 + blockIdx.x * PW;
  const int ncw = (nc - blockIdx.x * PW < PW) ? nc - blockIdx.x * PW : PW;
  const cplx* __restrict__ Vp = Vb + (long)b * v_b0 + (long)panel * PW * zr;
  const cplx* Tp = Tb + (long)b * t_b0 + (long)panel * PW * PW;
  cplx* Cb = C + (long)b * c_b0;
  sTm[tid] = Tp[tid];
  sW1[tid] = cplx{0.0, 0.0};
  __syncthreads();
  const int li = lane & 15, lk = lane >> 4;
  const int row0 = panel * PW;  // the reflector block is zero above its first row: start there
  {  // W1[i][c] = sum_r conj(V[i][r]) C[c][r]   (K = rows in chunks of 16, split over the four wavefronts).  The order of the
     // K index is free, so lane (li, lk) takes the four CONSECUTIVE rows base + 4 lk + q: 64-byte runs instead of 16-byte ones.
    real4 P = {0, 0, 0, 0}, Q = {0, 0, 0, 0}, S1 = {0, 0, 0, 0}, S2 = {0, 0, 0, 0};
    const int nsteps = (zr - row0 + 15) >> 4;
    const bool cvalid = li < ncw;
    const cplx* vcol = Vp + (long)li * zr;
    const cplx* ccol = Cb + (long)(c0 + li) * zr;
    for (int s = wave; s < nsteps; s += 4) {
      const int rb = row0 + 16 * s + 4 * lk;
      cplx v[4], x[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = rb + q;
        v[q] = (r < zr) ? vcol[r] : cplx{0.0, 0.0};
        x[q] = (cvalid && r < zr) ? ccol[r] : cplx{0.0, 0.0};
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        P = TJM_MFMA(v[q].x, x[q].x, P);
        Q = TJM_MFMA(v[q].y, x[q].y, Q);
        S1 = TJM_MFMA(v[q].x, x[q].y, S1);
        S2 = TJM_MFMA(v[q].y, x[q].x, S2);
      }
    }
    for (int w = 0; w < 4; ++w) {  // deterministic reduction over the wavefronts
      if (wave == w) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = TJM_ACC_ROW(lane, q), c = li;
          cplx a = sW1[i * PW + c];
          a.x += P[q] + Q[q];
          a.y += S1[q] - S2[q];
          sW1[i * PW + c] = a;
        }
      }
      __syncthreads();
    }
  }
  {  // W2 = op(T) W1
    const int i = tid >> 4, c = tid & 15;
    cplx acc{0.0, 0.0};
    for (int l = 0; l < PW; ++l) {
      const cplx t = t_herm ? cconj(sTm[l * PW + i]) : sTm[i * PW + l];
      cfma(acc, t, sW1[l * PW + c]);
    }
    sW2[i * PW + c] = acc;
  }
  __syncthreads();
  {  // C[c][r] -= sum_i W2[i][c] V[i][r], computed transposed: D[c][row] = sum_i A[c][i] B[i][row] with A = W2^T, B = V^T, so that
     // the 16 lanes of a row group read and write 16 consecutive rows of one column (256-byte runs)
    real wr[4], wi[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const cplx t = sW2[(4 * kk + lk) * PW + li];  // A[c = li][i = 4kk + lk]
      wr[kk] = t.x;
      wi[kk] = t.y;
    }
    const int nchunks = (zr - row0 + 15) >> 4;
    for (int ch = wave; ch < nchunks; ch += 4) {
      const int r0 = row0 + ch * 16;
      real4 P = {0, 0, 0, 0}, Q = {0, 0, 0, 0}, S1 = {0, 0, 0, 0}, S2 = {0, 0, 0, 0};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const cplx v = (r0 + li < zr) ? Vp[(long)(4 * kk + lk) * zr + r0 + li] : cplx{0.0, 0.0};  // B[i = 4kk + lk][row = li]
        P = TJM_MFMA(wr[kk], v.x, P);
        Q = TJM_MFMA(wi[kk], v.y, Q);
        S1 = TJM_MFMA(wi[kk], v.x, S1);
        S2 = TJM_MFMA(wr[kk], v.y, S2);
      }
      if (r0 + li < zr) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = TJM_ACC_ROW(lane, q);  // D: row (= column of C) of result register q, column (= row of C) li
          if (c >= ncw) continue;
          const long idx = (long)(c0 + c) * zr + r0 + li;
          cplx x = Cb[idx];
          x.x -= P[q] - Q[q];
          x.y -= S1[q] + S2[q];
          Cb[idx] = x;
        }
      }
    }
  }
}

// SEVERAL block reflectors in a row on a chunk of 16 columns that stays in LDS in between: the chunk is read from and written to
// memory once for np panels instead of once per panel.  qr_block_apply_kernel streams the whole trailing matrix through the chip
// for every 16-column panel (16 passes per factorisation of a 256-column matrix, 16 more for Q * C); with the panels grouped in
// fours the trailing matrix makes 2.5 x fewer trips (qr_factor), and Q * C is a single launch (qr_apply_q).  Per panel the
// arithmetic is that of qr_block_apply_kernel, instruction for instruction (same MFMA sequence, same order of the partial sums over
// the four wavefronts): the results are bit-identical to the one-panel-per-launch form.
//   panels p_first, p_first + p_step, ... (np of them); rows row_lo ... zr - 1 of the chunk are kept (row_lo = 16 x the smallest
//   panel: the reflectors of a panel are zero above its first row)
// MAXT: 16-row steps of a panel per wavefront (4: chunks of up to 256 rows, 8: up to 512).  A workgroup walks its panels one after the
// other and every panel needs the reflector block V twice, in two register layouts, from L2: those loads are issued early - the
// operands of the rank-16 update right after the inner products have been issued (consumed three barriers later), the operands of
// the next panel's inner products before the update of this one.  Measured: no change (Q x C of 256 matrices 839 -> 878 us alone on
// the device): the kernel is not waiting for those loads, it is at half of its matrix-core floor - 268 MFLOP per matrix in four real
// MFMAs per complex multiply-add = 437 us at the fp32 MFMA peak for the launch; the three-product form would be the next step.
template <int MAXT>
__global__ __launch_bounds__(256) void qr_block_apply_multi_kernel(const cplx* __restrict__ Vb, long v_b0, const cplx* __restrict__ Tb, long t_b0,
                                                                  int p_first, int p_step, int np, int zr, int t_herm, cplx* __restrict__ C,
                                                                  long c_b0, int col0, int nc, const int* ids, int row_lo, int pitch, int nb0, int xcd_map) {
  extern __shared__ real smem[];
  cplx* sC = reinterpret_cast<cplx*>(smem);   // [PW][pitch]
  cplx* sPart = sC + PW * pitch;              // [4][PW * PW] partial sums of the four wavefronts
  cplx* sW1 = sPart + 4 * PW * PW;            // (i, c)
  cplx* sW2 = sW1 + PW * PW;
  cplx* sTm = sW2 + PW * PW;
  // XCD-aware order (1-D grid): the column chunks of one matrix all read its reflector blocks, so they go to workgroups of equal
  // blockIdx % 8 - one XCD, one L2 (dealt in launch order, chunk c of every matrix sits on XCD c % 8)
  int b = blockIdx.y, chunk = blockIdx.x;
  if (xcd_map) {
    const int nchunks = (nc + PW - 1) / PW, q = blockIdx.x >> 3;
    chunk = q % nchunks;
    b = (q / nchunks) * 8 + (blockIdx.x & 7);
    if (b >= nb0) return;
  }
  if (ids) b = ids[b];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = col0 + chunk * PW;
  const int ncw = (nc - chunk * PW < PW) ? nc - chunk * PW : PW;
  cplx* Cb = C + (long)b * c_b0;
  const int nrows = zr - row_lo;
  const int li = lane & 15, lk = lane >> 4;
  const cplx* __restrict__ Vall = Vb + (long)b * v_b0;
  cplx v1[MAXT][4], v3[MAXT][4];
  // operands of the inner products of a panel: lane (li, lk) takes rows row0 + 16 s + 4 lk + q of reflector column li, s = wave + 4 t
  auto load_v1 = [&](int panel) {
    const int row0 = panel * PW;
    const int nsteps = (zr - row0 + 15) >> 4;
    const cplx* vcol = Vall + (long)panel * PW * zr + (long)li * zr;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int rb = row0 + 16 * (wave + 4 * t) + 4 * lk;
#pragma unroll
      for (int q = 0; q < 4; ++q) v1[t][q] = (wave + 4 * t < nsteps && rb + q < zr) ? vcol[rb + q] : cplx{0.0, 0.0};
    }
  };
  // operands of the update: lane (li, lk) takes row r0 + li of reflector columns 4 kk + lk, r0 = row0 + 16 (wave + 4 t)
  auto load_v3 = [&](int panel) {
    const int row0 = panel * PW;
    const int nchunks = (zr - row0 + 15) >> 4;
    const cplx* Vp = Vall + (long)panel * PW * zr;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int r = row0 + 16 * (wave + 4 * t) + li;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) v3[t][kk] = (wave + 4 * t < nchunks && r < zr) ? Vp[(long)(4 * kk + lk) * zr + r] : cplx{0.0, 0.0};
    }
  };
  load_v1(p_first);
  for (int c = 0; c < PW; ++c) {
    const cplx* src = Cb + (long)(c0 + c) * zr + row_lo;
    for (int r = tid; r < nrows; r += 256) sC[c * pitch + r] = (c < ncw) ? src[r] : cplx{0.0, 0.0};
  }
  for (int ip = 0; ip < np; ++ip) {
    const int panel = p_first + ip * p_step;
    const cplx* Tp = Tb + (long)b * t_b0 + (long)panel * PW * PW;
    const int row0 = panel * PW;
    sTm[tid] = Tp[tid];
    __syncthreads();  // the chunk (first panel) / the update of the panel before is in LDS
    {  // W1[i][c] = sum_r conj(V[i][r]) C[c][r], K split over the four wavefronts
      real4 P = {0, 0, 0, 0}, Q = {0, 0, 0, 0}, S1 = {0, 0, 0, 0}, S2 = {0, 0, 0, 0};
      const int nsteps = (zr - row0 + 15) >> 4;
      const cplx* ccol = sC + li * pitch - row_lo;
#pragma unroll
      for (int t = 0; t < MAXT; ++t) {
        const int s = wave + 4 * t;
        if (s < nsteps) {
          const int rb = row0 + 16 * s + 4 * lk;
          cplx x[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) x[q] = (rb + q < zr) ? ccol[rb + q] : cplx{0.0, 0.0};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            P = TJM_MFMA(v1[t][q].x, x[q].x, P);
            Q = TJM_MFMA(v1[t][q].y, x[q].y, Q);
            S1 = TJM_MFMA(v1[t][q].x, x[q].y, S1);
            S2 = TJM_MFMA(v1[t][q].y, x[q].x, S2);
          }
        }
      }
      load_v3(panel);  // in flight during the reduction and the product with T
#pragma unroll
      for (int q = 0; q < 4; ++q) sPart[wave * PW * PW + TJM_ACC_ROW(lane, q) * PW + li] = cplx{P[q] + Q[q], S1[q] - S2[q]};
    }
    __syncthreads();
    {  // the four partial sums in the order of the one-panel kernel: (((0 + p0) + p1) + p2) + p3
      cplx a{0.0, 0.0};
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        a.x += sPart[w * PW * PW + tid].x;
        a.y += sPart[w * PW * PW + tid].y;
      }
      sW1[tid] = a;
    }
    __syncthreads();
    {  // W2 = op(T) W1
      const int i = tid >> 4, c = tid & 15;
      cplx acc{0.0, 0.0};
      for (int l = 0; l < PW; ++l) {
        const cplx t = t_herm ? cconj(sTm[l * PW + i]) : sTm[i * PW + l];
        cfma(acc, t, sW1[l * PW + c]);
      }
      sW2[i * PW + c] = acc;
    }
    __syncthreads();
    if (ip + 1 < np) load_v1(panel + p_step);  // the next panel's inner-product operands, in flight during this panel's update
    {  // C[c][r] -= sum_i W2[i][c] V[i][r] on the chunk in LDS
      real wr[4], wi[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const cplx t = sW2[(4 * kk + lk) * PW + li];
        wr[kk] = t.x;
        wi[kk] = t.y;
      }
      const int nchunks = (zr - row0 + 15) >> 4;
#pragma unroll
      for (int t = 0; t < MAXT; ++t) {
        const int ch = wave + 4 * t;
        if (ch < nchunks) {
          const int r0 = row0 + ch * 16;
          real4 P = {0, 0, 0, 0}, Q = {0, 0, 0, 0}, S1 = {0, 0, 0, 0}, S2 = {0, 0, 0, 0};
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            P = TJM_MFMA(wr[kk], v3[t][kk].x, P);
            Q = TJM_MFMA(wi[kk], v3[t][kk].y, Q);
            S1 = TJM_MFMA(wi[kk], v3[t][kk].x, S1);
            S2 = TJM_MFMA(wr[kk], v3[t][kk].y, S2);
          }
          if (r0 + li < zr) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int c = TJM_ACC_ROW(lane, q);
              cplx* px = sC + c * pitch + (r0 + li - row_lo);
              cplx x = *px;
              x.x -= P[q] - Q[q];
              x.y -= S1[q] + S2[q];
              *px = x;
            }
          }
        }
      }
    }
  }
  __syncthreads();
  for (int c = 0; c < ncw; ++c) {
    cplx* dst = Cb + (long)(c0 + c) * zr + row_lo;
    for (int r = tid; r < nrows; r += 256) dst[r] = sC[c * pitch + r];
  }
}

// Z (column-major zr x zc) from theta (row-major m x n, rows (s,a), columns (t,c)):
//   dist 0 -> Z = theta   with rows re-ordered bond-major:  r' = a * d + s
//   dist 1 -> Z = theta^H with rows re-ordered bond-major:  r' = c * d + t
// Bond-major rows make the zero padding (bond index >= actual bond dimension) a SUFFIX of the row range, so every
// Householder reflector stays inside the active rows and the padded rows of Q stay exactly untouched.
// Order of the columns of Z by decreasing norm: cperm[j] = source column that becomes column j.  One workgroup per
// trajectory; the rank of a column is the number of columns that precede it (ties broken by index: deterministic).
__global__ __launch_bounds__(256) void qr_colsort_kernel(const cplx* __restrict__ theta, long th_b0, int m, int n, int dist, int* __restrict__ cperm,
                                                        int ld, const int* ids, int zc_pad) {
  __shared__ real sn[1024];
  int b = blockIdx.x;
  if (ids) b = ids[b];
  const cplx* th = theta + (long)b * th_b0;
  const int tid = threadIdx.x;
  const int zc = (dist == 0) ? n : m;
  if (dist == 0) {  // column c of theta
    for (int c = tid; c < zc; c += 256) {
      real acc = 0.0;
      for (int r = 0; r < m; ++r) {
        const cplx v = th[(long)r * n + c];
        acc = fma(v.x, v.x, fma(v.y, v.y, acc));
      }
      sn[c] = acc;
    }
  } else {          // row i of theta, one wavefront per row
    const int lane = tid & 63, wave = tid >> 6;
    for (int i = wave; i < zc; i += 4) {
      real acc = 0.0;
      for (int k = lane; k < n; k += 64) {
        const cplx v = th[(long)i * n + k];
        acc = fma(v.x, v.x, fma(v.y, v.y, acc));
      }
      acc = wsum(acc);
      if (lane == 0) sn[i] = acc;
    }
  }
  __syncthreads();
  for (int c = tid; c < zc; c += 256) {
    const real mine = tjm_sort_key(sn[c]);
    int rank = 0;
    for (int o = 0; o < zc; ++o) {
      const real other = tjm_sort_key(sn[o]);
      rank += (other > mine || (other == mine && o < c)) ? 1 : 0;
    }
    cperm[(long)b * ld + rank] = c;
  }
  for (int c = zc + tid; c < zc_pad; c += 256) cperm[(long)b * ld + c] = c;  // zero columns of the square embedding
}

__global__ __launch_bounds__(256) void qr_prepare_kernel(const cplx* __restrict__ theta, long th_b0, int m, int n, int dist, int d, cplx* __restrict__ Z,
                                                        long z_b0, const int* __restrict__ cperm, int perm_ld, const int* ids, int zr_pad, int zc_pad) {
  int b = blockIdx.y;
  if (ids) b = ids[b];
  const cplx* th = theta + (long)b * th_b0;
  cplx* Zb = Z + (long)b * z_b0;
  const int* cp = cperm + (long)b * perm_ld;
  // Z is zr_pad x zc_pad (column-major); the block of rows < zr and columns < zc holds theta (dist 0) or theta^H (dist 1), the
  // rest is zero: a rectangular theta embedded in a square matrix keeps its singular values and vectors (zero rows stay a suffix)
  const int zr = (dist == 0) ? m : n, zc = (dist == 0) ? n : m;
  const long total = (long)zr_pad * zc_pad;
  const int capL = m / d, capR = n / d;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long c = e / zr_pad, rp = e % zr_pad;
    cplx v{0.0, 0.0};
    if (rp < zr && c < zc) {
      if (dist == 0) {                               // Z(rp, c), rp = a * d + s
        const int a = (int)(rp / d), sph = (int)(rp % d);
        v = th[((long)sph * capL + a) * n + cp[c]];
      } else {                                       // Z(rp, i) = conj(theta[i][(t,c)]), rp = c * d + t
        const int cc = (int)(rp / d), t = (int)(rp % d);
        v = th[(long)cp[c] * n + (long)t * capR + cc];
        v.y = -v.y;
      }
    }
    Zb[e] = v;
  }
}

// out[k*o_k + r1*o_r1 + r0*o_r0] = op(in[k*ld + r1*n_r0 + r0])  for k < n_k (zero beyond keep)
__global__ __launch_bounds__(256) void qr_scatter_kernel(const cplx* __restrict__ in, long in_b0, int ld, ExtractDesc x, const int* chi_keep,
                                                        int chi_stride, const int* ids) {
  int b = blockIdx.y;
  if (ids) b = ids[b];
  const cplx* ib = in + (long)b * in_b0;
  cplx* out = x.out + (long)b * x.out_b0;
  const int keep = chi_keep[(long)b * chi_stride];
  const long nrows = (long)x.n_r1 * x.n_r0;
  const long total = nrows * x.n_k;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e / nrows);
    const long r = e % nrows;
    const long ro = x.row_map ? x.row_map[(long)b * x.row_map_ld + r] : r;
    const int r1 = (int)(ro / x.n_r0), r0 = (int)(ro % x.n_r0);
    cplx v{0.0, 0.0};
    if (k < keep) {
      v = ib[(long)k * ld + r];
      if (x.conj) v.y = -v.y;
    }
    out[(long)k * x.o_k + (long)r1 * x.o_r1 + (long)r0 * x.o_r0] = v;
  }
}

GemmDesc blank() {
  GemmDesc g;
  memset(&g, 0, sizeof(g));
  g.nks = 1; g.nb0 = 1; g.nb1 = 1; g.nb2 = 1;
  return g;
}

// C (col-major zr x nc at C + col0 * zr) <- C - V_p op(T_p) V_p^H C     (op = T^H for the factorisation, T for Q * C)
int apply_block_reflector(const QrWorkspace& q, int zr, int panel, bool t_herm, cplx* C, long c_b0, int col0, int nc, int nb0, const int* ids,
                          hipStream_t s) {
  if (nc <= 0) return TJM_OK;
  hipLaunchKernelGGL(qr_block_apply_kernel, dim3((nc + PW - 1) / PW, nb0), dim3(256), 0, s, q.V, q.v_b0, q.T, q.t_b0, panel, zr, t_herm ? 1 : 0, C,
                     c_b0, col0, nc, ids);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

// LDS of qr_block_apply_multi_kernel for a chunk of `nrows` rows: the chunk (row pitch padded by 16 bytes: the 16 columns of a
// wavefront's operand read start in different banks) + partial sums, W1, W2, T
int multi_pitch(int nrows) { return nrows + (int)(16 / sizeof(cplx)); }
size_t multi_lds_bytes(int nrows) { return ((size_t)PW * multi_pitch(nrows) + 7 * PW * PW) * sizeof(cplx); }
// panels grouped per pass over the trailing matrix (TJM_QR_GROUP, default 4; 1: one panel per launch as in rounds 1 - 4); the
// chunk has to leave room for two workgroups per CU
// (TJM_QR_MULTI_LDS_KB: 160 by default - up to 80 KB two workgroups share a CU, above it one has the CU to itself: the 512-row chunks of
// the fp64 centre shifts at chi = 256, config 4: 5.03 against 4.96 trajectories/s with the limit at 80)
size_t multi_lds_limit() {
  static const size_t kb = getenv("TJM_QR_MULTI_LDS_KB") ? (size_t)atoi(getenv("TJM_QR_MULTI_LDS_KB")) : 160;
  return kb * 1024;
}
int multi_group(int zr) {
  static const int g = getenv("TJM_QR_GROUP") ? atoi(getenv("TJM_QR_GROUP")) : 4;
  return (g > 1 && zr <= 512 && multi_lds_bytes(zr) <= multi_lds_limit()) ? g : 1;  // (the kernel holds chunks of up to 512 rows)
}

// np panels p_first, p_first + p_step, ... on the columns [col0, col0 + nc) of C in one launch
// Launch sampler of qr_block_apply_multi_kernel (bench.py's roofline.kernels entry of the QR preconditioner): every N-th launch is
// bracketed by HIP events on its stream; the work of a launch is the nominal count of its block reflectors, 8 real flops per complex
// multiply-add: per panel V^H C and C - V (T V^H C), 2 x 16 x rows x columns.  Off unless qr_profile_enable was called.
struct QrProf {
  std::mutex m;
  int every = 0;
  long counter = 0, samples = 0, launches = 0;
  double ms = 0.0, flops = 0.0, flops_all = 0.0;
  std::vector<hipEvent_t> pool;
  std::vector<std::pair<int, double>> pending;  // (event pair, flops)
  size_t used = 0;
};
QrProf g_qrp;

void qr_prof_harvest_locked() {
  for (auto& p : g_qrp.pending) {
    float ms = 0.f;
    if (hipEventSynchronize(g_qrp.pool[2 * p.first + 1]) == hipSuccess && hipEventElapsedTime(&ms, g_qrp.pool[2 * p.first], g_qrp.pool[2 * p.first + 1]) == hipSuccess) {
      g_qrp.ms += ms;
      g_qrp.flops += p.second;
      ++g_qrp.samples;
    }
  }
  (void)hipGetLastError();
  g_qrp.pending.clear();
  g_qrp.used = 0;
}

int apply_block_reflectors(const QrWorkspace& q, int zr, int p_first, int p_step, int np, bool t_herm, cplx* C, long c_b0, int col0, int nc, int nb0,
                           const int* ids, hipStream_t s) {
  if (nc <= 0 || np <= 0) return TJM_OK;
  if (np == 1) return apply_block_reflector(q, zr, p_first, t_herm, C, c_b0, col0, nc, nb0, ids, s);
  static std::atomic<bool> attr_set{false};
  if (!attr_set.load(std::memory_order_acquire)) {
    TJM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(qr_block_apply_multi_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)multi_lds_limit()));
    TJM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(qr_block_apply_multi_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)multi_lds_limit()));
    attr_set.store(true, std::memory_order_release);
  }
  const int p_last = p_first + (np - 1) * p_step;
  const int row_lo = PW * (p_first < p_last ? p_first : p_last);
  const int nrows = zr - row_lo;
  if (nrows > 512) return TJM_ERR_NOT_IMPLEMENTED;  // (multi_group keeps the chunk within 80 KB: never reached)
  static const bool flat = getenv("TJM_GEMM_FLAT_TILES") != nullptr;
  const int nchunks = (nc + PW - 1) / PW;
  const int xcd_map = (!flat && nb0 >= 16) ? 1 : 0;
  const dim3 grid = xcd_map ? dim3((unsigned)(nchunks * ((nb0 + 7) / 8 * 8)), 1) : dim3(nchunks, nb0);
  int ev = -1;
  double fl = 0.0;
  std::unique_lock<std::mutex> plock(g_qrp.m, std::defer_lock);
  if (g_qrp.every > 0) {
    plock.lock();
    if (g_qrp.every > 0) {
      for (int k = 0; k < np; ++k) fl += 2.0 * 8.0 * PW * (double)(zr - PW * (p_first + k * p_step)) * nc * nb0;
      ++g_qrp.launches;
      g_qrp.flops_all += fl;
      if (g_qrp.counter++ % g_qrp.every == 0) {
        if (g_qrp.used >= 4096) qr_prof_harvest_locked();
        if (g_qrp.pool.size() < 2 * (g_qrp.used + 1)) {
          hipEvent_t a, b;
          if (hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) { g_qrp.pool.push_back(a); g_qrp.pool.push_back(b); }
        }
        if (g_qrp.pool.size() >= 2 * (g_qrp.used + 1)) {
          ev = (int)g_qrp.used++;
          (void)hipEventRecord(g_qrp.pool[2 * ev], s);
        }
      }
    }
  }
  if (nrows <= 256)
    hipLaunchKernelGGL(qr_block_apply_multi_kernel<4>, grid, dim3(256), multi_lds_bytes(nrows), s, q.V, q.v_b0, q.T, q.t_b0, p_first,
                       p_step, np, zr, t_herm ? 1 : 0, C, c_b0, col0, nc, ids, row_lo, multi_pitch(nrows), nb0, xcd_map);
  else
    hipLaunchKernelGGL(qr_block_apply_multi_kernel<8>, grid, dim3(256), multi_lds_bytes(nrows), s, q.V, q.v_b0, q.T, q.t_b0, p_first,
                       p_step, np, zr, t_herm ? 1 : 0, C, c_b0, col0, nc, ids, row_lo, multi_pitch(nrows), nb0, xcd_map);
  if (ev >= 0) {
    (void)hipEventRecord(g_qrp.pool[2 * ev + 1], s);
    g_qrp.pending.emplace_back(ev, fl);
  }
  if (plock.owns_lock()) plock.unlock();
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

}  // namespace

namespace {
// Z2 (n x n, column-major) = R^H where R is the upper triangle of the factored Z (leading dimension n)
__global__ __launch_bounds__(256) void qr_adjoint_triangle_kernel(const cplx* __restrict__ Z, long z_b0, cplx* __restrict__ Z2, int n, const int* ids) {
  int b = blockIdx.y;
  if (ids) b = ids[b];
  const cplx* Zb = Z + (long)b * z_b0;
  cplx* Ob = Z2 + (long)b * z_b0;
  const long total = (long)n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e / n), r = (int)(e % n);  // R^H[r][c] = conj(R[c][r]) , R[c][r] = Z[r * n + c] for c <= r
    cplx v{0.0, 0.0};
    if (r >= c) { v = Zb[(long)r * n + c]; v.y = -v.y; }
    Ob[e] = v;
  }
}
}  // namespace

namespace {
__global__ __launch_bounds__(256) void qr_gather_scaled_kernel(const cplx* __restrict__ G, long g_b0, int rows, int ncols, int d,
                                                              const real* __restrict__ sigma, int sig_ld, const int* __restrict__ keep,
                                                              int keep_stride, cplx* __restrict__ Z, long z_b0) {
  const int b = blockIdx.y;
  const cplx* Gb = G + (long)b * g_b0;
  cplx* Zb = Z + (long)b * z_b0;
  const int kp = keep[(long)b * keep_stride];
  const int cap = rows / d;
  const long total = (long)rows * ncols;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e / rows), rp = (int)(e % rows);  // destination: column k, bond-major row rp = bond * d + p
    const int bond = rp / d, ph = rp % d;
    cplx v{0.0, 0.0};
    if (k < kp) {
      const real sg = sigma[(long)b * sig_ld + k];
      if (sg > 0.0) {
        const real inv = 1.0 / sg;
        v = Gb[((long)ph * cap + bond) * ncols + k];
        v.x *= inv;
        v.y *= inv;
      }
    }
    Zb[e] = v;
  }
}

__global__ __launch_bounds__(256) void qr_identity_kernel(cplx* __restrict__ C, long c_b0, int rows, int ncols, const int* ids, const int* keep,
                                                         int keep_stride) {
  const int b = ids ? ids[blockIdx.y] : blockIdx.y;
  cplx* Cb = C + (long)b * c_b0;
  const long total = (long)rows * ncols;
  const int kp = keep ? keep[(long)b * keep_stride] : ncols;  // columns beyond the kept ones stay zero
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x)
    Cb[e] = cplx{(e / rows == e % rows && e / rows < kp) ? real(1) : real(0), 0.0};
}

__global__ __launch_bounds__(256) void qr_r_times_sigma_kernel(const cplx* __restrict__ Z, long z_b0, int zr, int ncols, const real* __restrict__ sigma,
                                                              int sig_ld, const int* __restrict__ keep, int keep_stride, cplx* __restrict__ Rs,
                                                              long rs_b0) {
  const int b = blockIdx.y;
  const cplx* Zb = Z + (long)b * z_b0;
  cplx* Rb = Rs + (long)b * rs_b0;
  const int kp = keep[(long)b * keep_stride];
  const long total = (long)ncols * ncols;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e / ncols), j = (int)(e % ncols);
    cplx v{0.0, 0.0};
    if (k <= j && j < kp && k < zr) {
      const real sg = sigma[(long)b * sig_ld + j];
      v = Zb[(long)j * zr + k];
      v.x *= sg;
      v.y *= sg;
    }
    Rb[e] = v;
  }
}
}  // namespace

int qr_gather_scaled(const cplx* G, long g_b0, int rows, int ncols, int d, const real* sigma, int sig_ld, const int* keep, int keep_stride,
                     cplx* Z, long z_b0, int nb0, hipStream_t s) {
  const long total = (long)rows * ncols;
  int gx = (int)((total + 1023) / 1024);
  if (gx > 128) gx = 128;
  hipLaunchKernelGGL(qr_gather_scaled_kernel, dim3(gx, nb0), dim3(256), 0, s, G, g_b0, rows, ncols, d, sigma, sig_ld, keep, keep_stride, Z, z_b0);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

int qr_identity(cplx* C, long c_b0, int rows, int ncols, int nb0, hipStream_t s, const int* ids, const int* keep, int keep_stride) {
  const long total = (long)rows * ncols;
  int gx = (int)((total + 1023) / 1024);
  if (gx > 128) gx = 128;
  hipLaunchKernelGGL(qr_identity_kernel, dim3(gx, nb0), dim3(256), 0, s, C, c_b0, rows, ncols, ids, keep, keep_stride);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

int qr_r_times_sigma(const cplx* Z, long z_b0, int zr, int ncols, const real* sigma, int sig_ld, const int* keep, int keep_stride, cplx* Rs,
                     long rs_b0, int nb0, hipStream_t s) {
  const long total = (long)ncols * ncols;
  int gx = (int)((total + 1023) / 1024);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(qr_r_times_sigma_kernel, dim3(gx, nb0), dim3(256), 0, s, Z, z_b0, zr, ncols, sigma, sig_ld, keep, keep_stride, Rs, rs_b0);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

int qr_adjoint_triangle(const QrWorkspace& q, int n, int nb0, const int* ids, hipStream_t s) {
  const long total = (long)n * n;
  int gx = (int)((total + 1023) / 1024);
  if (gx > 128) gx = 128;
  hipLaunchKernelGGL(qr_adjoint_triangle_kernel, dim3(gx, nb0), dim3(256), 0, s, q.Z, q.z_b0, q.Z2, n, ids);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

size_t qr_carve(QrWorkspace& q, char* base, int max_dim, int B) {
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += (bytes + 255) / 256 * 256; return p; };
  const int npan = max_dim / PW + 1;
  q.z_b0 = (long)max_dim * max_dim;
  q.v_b0 = (long)npan * PW * max_dim;
  q.t_b0 = (long)npan * PW * PW;
  q.w_ld = max_dim;
  q.Z = reinterpret_cast<cplx*>(take((size_t)B * q.z_b0 * sizeof(cplx)));
  q.V = reinterpret_cast<cplx*>(take((size_t)B * q.v_b0 * sizeof(cplx)));
  q.T = reinterpret_cast<cplx*>(take((size_t)B * q.t_b0 * sizeof(cplx)));
  q.W1 = reinterpret_cast<cplx*>(take((size_t)B * PW * max_dim * sizeof(cplx)));
  q.W2 = nullptr;
  q.Z2 = reinterpret_cast<cplx*>(take((size_t)B * q.z_b0 * sizeof(cplx)));
  q.V2 = reinterpret_cast<cplx*>(take((size_t)B * q.v_b0 * sizeof(cplx)));
  q.T2 = reinterpret_cast<cplx*>(take((size_t)B * q.t_b0 * sizeof(cplx)));
  return off;
}

size_t qr_workspace_bytes(int max_dim, int B) {
  QrWorkspace q;
  return qr_carve(q, nullptr, max_dim, B) + 4096;
}

int qr_prepare(const cplx* theta, long th_b0, int m, int n, int dist, int d, const QrWorkspace& q, int nb0, const int* ids, hipStream_t s, int square) {
  const int zr = (dist == 0) ? m : n, zc = (dist == 0) ? n : m;
  const int zr_pad = square > 0 ? square : zr, zc_pad = square > 0 ? square : zc;
  if (zr_pad < zr || zc_pad < zc) return TJM_ERR_ARG;
  const long total = (long)zr_pad * zc_pad;
  int gx = (int)((total + 1023) / 1024);
  if (gx > 128) gx = 128;
  if (zc_pad > 1024 || zc_pad > q.w_ld || total > q.z_b0) return TJM_ERR_NOT_IMPLEMENTED;
  hipLaunchKernelGGL(qr_colsort_kernel, dim3(nb0), dim3(256), 0, s, theta, th_b0, m, n, dist, q.colperm(), q.w_ld, ids, zc_pad);
  hipLaunchKernelGGL(qr_prepare_kernel, dim3(gx, nb0), dim3(256), 0, s, theta, th_b0, m, n, dist, d, q.Z, q.z_b0, q.colperm(), q.w_ld, ids, zr_pad,
                     zc_pad);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

// Z (zr x zc, column-major in q.Z) = Q R in place: R in the upper triangle, reflector blocks in q.V / q.T.
int qr_factor(const QrWorkspace& q, int zr, int zc, int nb0, const int* ids, hipStream_t s) {
  const int kmax = zr < zc ? zr : zc;
  static std::atomic<bool> attr_set{false};  // several engines of one process call this from their own host threads
  if (!attr_set.load(std::memory_order_acquire)) {
    TJM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(qr_panel_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    TJM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(qr_panel_rows_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    TJM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(qr_panel_rows_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    TJM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(qr_panel_rows_kernel<2, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set.store(true, std::memory_order_release);
  }
  static const bool no_rows = getenv("TJM_QR_LDS_PANEL") != nullptr;
  int rc;
  int panel = 0;
  const int group = multi_group(zr);
  for (int k0 = 0; k0 < kmax; k0 += PW, ++panel) {
    const int pw = (kmax - k0 < PW) ? kmax - k0 : PW;
    const int mp = zr - k0;
    if (mp > 1024) return TJM_ERR_NOT_IMPLEMENTED;  // panels of at most 1024 rows (bonds up to 512)
    if ((mp <= 512 && !no_rows) || mp > 384) {  // rows of the panel in registers: 1, 2 or 4 per thread
      static const bool wide = getenv("TJM_QR_WIDE_PANEL") != nullptr;  // diagnostic: the 512-thread kernel from 257 rows on
      const int nt = (mp <= 512 && !(wide && mp > 256)) ? 256 : 512;
      const size_t lds = (size_t)(NRED * (nt + 1) + 2 * NRED + PW) * sizeof(real) + (size_t)(2 * PW + 2 * PW * PW + PW) * sizeof(cplx);
      if (mp <= 256)
        hipLaunchKernelGGL(qr_panel_rows_kernel<1>, dim3(nb0), dim3(256), lds, s, q.Z, q.z_b0, zr, k0, pw, q.V, q.v_b0, q.T, q.t_b0, panel, ids);
      else if (nt == 256)
        hipLaunchKernelGGL(qr_panel_rows_kernel<2>, dim3(nb0), dim3(256), lds, s, q.Z, q.z_b0, zr, k0, pw, q.V, q.v_b0, q.T, q.t_b0, panel, ids);
      else
        hipLaunchKernelGGL((qr_panel_rows_kernel<2, 512>), dim3(nb0), dim3(512), lds, s, q.Z, q.z_b0, zr, k0, pw, q.V, q.v_b0, q.T, q.t_b0, panel, ids);
    } else {
      const size_t lds = (size_t)(PW * (zr - k0) + 2 * PW * PW + PW) * sizeof(cplx) + PW * sizeof(real) + 64;
      hipLaunchKernelGGL(qr_panel_kernel, dim3(nb0), dim3(64), lds, s, q.Z, q.z_b0, zr, k0, pw, q.V, q.v_b0, q.T, q.t_b0, panel, ids);
    }
    TJM_HIP_CHECK(hipGetLastError());
    const int col0 = k0 + pw;
    // two-level blocking: the reflectors of a panel go to the rest of its GROUP of panels at once, and to the columns right of the
    // group together with the other panels of the group, in one pass over those columns (every column still sees the panels in
    // ascending order: same arithmetic, same results)
    const int first = (panel / group) * group;
    const int group_end = (first + group) * PW < zc ? (first + group) * PW : zc;
    if ((rc = apply_block_reflector(q, zr, panel, true, q.Z, q.z_b0, col0, group_end - col0, nb0, ids, s)) != TJM_OK) return rc;
    if (panel + 1 == first + group || k0 + PW >= kmax)
      if ((rc = apply_block_reflectors(q, zr, first, 1, panel - first + 1, true, q.Z, q.z_b0, group_end, zc - group_end, nb0, ids, s)) != TJM_OK) return rc;
  }
  return TJM_OK;
}

// C (zr x nc, column-major, leading dimension zr) <- Q C
int qr_apply_q(const QrWorkspace& q, int zr, int zc, cplx* C, long c_b0, int nc, int nb0, const int* ids, hipStream_t s) {
  const int kmax = zr < zc ? zr : zc;
  const int npanels = (kmax + PW - 1) / PW;
  int rc;
  if (multi_group(zr) > 1)  // all panels in one launch, the chunk of C resident
    return apply_block_reflectors(q, zr, npanels - 1, -1, npanels, false, C, c_b0, 0, nc, nb0, ids, s);
  for (int p = npanels - 1; p >= 0; --p)
    if ((rc = apply_block_reflector(q, zr, p, false, C, c_b0, 0, nc, nb0, ids, s)) != TJM_OK) return rc;
  return TJM_OK;
}

void qr_profile_enable(int every) {
  std::lock_guard<std::mutex> lock(g_qrp.m);
  qr_prof_harvest_locked();
  g_qrp.every = every;
  g_qrp.counter = 0; g_qrp.samples = 0; g_qrp.launches = 0;
  g_qrp.ms = 0.0; g_qrp.flops = 0.0; g_qrp.flops_all = 0.0;
}

// out5: summed duration of the sampled launches (ms), their nominal flops, samples, all launches, nominal flops of all launches
void qr_profile_get(double* out5) {
  std::lock_guard<std::mutex> lock(g_qrp.m);
  qr_prof_harvest_locked();
  out5[0] = g_qrp.ms; out5[1] = g_qrp.flops; out5[2] = (double)g_qrp.samples; out5[3] = (double)g_qrp.launches; out5[4] = g_qrp.flops_all;
}

int qr_scatter(const cplx* in, long in_b0, int ld, const ExtractDesc& x, const int* chi_keep, int chi_stride, int nb0, const int* ids,
               hipStream_t s) {
  const long total = (long)x.n_r1 * x.n_r0 * x.n_k;
  if (total <= 0) return TJM_OK;
  int gx = (int)((total + 1023) / 1024);
  if (gx > 128) gx = 128;
  hipLaunchKernelGGL(qr_scatter_kernel, dim3(gx, nb0), dim3(256), 0, s, in, in_b0, ld, x, chi_keep, chi_stride, ids);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

}  // namespace tjm
