// Batched truncated SVD for the TJM sweep: register-resident one-sided Jacobi (Hestenes) with
// the reference's truncation rule applied on the device.
//
// Replaces decompositions.py:105-185 (split_two_site: scipy zgesdd + linalg.truncate,
// core/linalg/svd_utils.py:22-104) and the SVD centre shifts of mps.py:747-788 for a whole
// batch of trajectories at once.
//
// Method.  The columns of X (rx x ncols) are orthogonalised by plane rotations accumulated in
// W (X W = Q Sigma, W unitary).  The stacked matrix Y = [X; W] is stored column-major in HBM /
// Infinity Cache (2 MiB per trajectory at d*chi = 256).  Columns are grouped in blocks of 8:
//   * jacobi_cross_kernel: one workgroup (8 wavefronts) per block pair (I, J) of a round-robin
//     round.  Wavefront w keeps column w of block I and one column of block J in registers
//     (lane l holds rows l, l+64, ...), computes <y_I, y_J> over the X rows with a wavefront
//     reduction, rotates both columns in registers, then the J columns move one wavefront on
//     through LDS.  8 steps visit the 64 cross pairs of the block pair; each column is read
//     from and written to memory once per visit, with 1 KiB coalesced accesses.
//   * jacobi_diag_kernel: the 28 pairs inside each block, LDS-resident, once per sweep.
// A sweep is 1 diag launch + (nblk - 1) cross launches over (block pairs x trajectories)
// workgroups; trajectories whose sweep performed no rotation are flagged done and skipped.
// The isometric output is always read from W and the sigma-weighted output from the rotated
// X for the two-site split, so that path never divides by a singular value.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

#include <atomic>

#include "tjm_kernels.h"
#ifndef TJM_F32
#include "tjm_mixed.h"
#endif

namespace tjm {

namespace {

constexpr int NB = 8;        // columns per block
constexpr int MAXRK = 16;    // row groups of 64 per column held in registers (rtot <= 1024); kernels are instantiated for 8 and 16
constexpr int MAXBLK = 128;  // column blocks per matrix (ncols <= 1024: bonds up to 512)
constexpr int STAMP_STRIDE = 3 * MAXBLK + MAXBLK * MAXBLK;
constexpr int REC_PER_VISIT = 2 * NB * 2 * NB;  // pairs of a 16 x 16 column tile: 8 steps x 2 sub-steps x 8 wavefronts x 2 pairs = 256 rotations

struct JacobiArgs {
  cplx* Y;
  long y_b0;
  int rtot;     // rows of the stacked column (multiple of 64)
  int rx;       // rows of X (top part)
  int nblk;     // number of column blocks (even)
  int round;    // round-robin round
  real tol2;  // squared relative tolerance
  const real* fro2;
  int* nrot;
  const int* done;
  const int* ids;
  int mode;     // 0: round-robin pair of the round ; 1: sibling pair (2p, 2p+1) of 8-column blocks
  real* rec;  // [B][MAXBLK/4][256][4] rotation record of the split X / W scheme (c, sr, si, flag)
  int* stamps;  // [B][STAMP_STRIDE]: mod[MAXBLK] | verd[MAXBLK] | nz[MAXBLK] | ver[MAXBLK*MAXBLK]   (visit pruning)
  int clock;    // launch counter, strictly increasing inside one solve
  int fold;     // jacobi_cross16x_kernel: the pairs inside a 16-column block ride along with the tile visits (no diag / sibling launches)
  int* work;    // rotation slots executed in this sweep (tile visits x pairs per visit; the identity rotations of a visited tile count)
  real floor_scale;  // columns below sqrt(floor_scale) ||X||_F are numerically null (TJM_NOISE_FLOOR2 unless the caller says otherwise)
  int ngroups;  // jacobi_quad64_kernel: groups of 16 blocks (256 columns) of the matrix ...
  int ground;   // ... and, in its cross mode (mode 1), the round of the circle method on the groups (round = the shift 0 ... 7)
  int ablate;   // timing ablation of jacobi_quad64_kernel (TJM_Q64_ABLATE, wrong results): 1 = leave behind the loads and column norms
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

// ---- wavefront all-reduce without LDS traffic: four DPP butterfly stages inside each row of 16 lanes
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), then the four row totals are read with v_readlane.
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

__device__ inline real row_total(real v) {
  v += dpp_pull<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_pull<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_pull<0x141>(v);  // row_half_mirror
  v += dpp_pull<0x140>(v);  // row_mirror
  return v;
}

__device__ inline real lane_value(real v, int lane) {
#ifdef TJM_F32
  return tjm_readlane(v, lane);
#else
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
#endif
}

__device__ inline real wave_sum(real v) {
  v = row_total(v);
  return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}

// sum over lanes l ^ 16 and l ^ 32 with the gfx950 row / half-wave swap instructions (pure VALU, no LDS crossbar)
__device__ inline real xor16_sum(real v) {
#ifdef TJM_F32
  return tjm_xor16_sum(v);
#else
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
#endif
}
__device__ inline real xor32_sum(real v) {
#ifdef TJM_F32
  return tjm_xor32_sum(v);
#else
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
#endif
}

// Four wavefront sums for the price of two: a halving butterfly.  On return lane l holds the total of p[l & 3].
__device__ inline real wave_sum4(real p0, real p1, real p2, real p3, int lane) {
  const bool b0 = lane & 1, b1 = lane & 2;
  const real x01 = (b0 ? p1 : p0) + dpp_pull<0xB1>(b0 ? p0 : p1);
  const real x23 = (b0 ? p3 : p2) + dpp_pull<0xB1>(b0 ? p2 : p3);
  real y = (b1 ? x23 : x01) + dpp_pull<0x4E>(b1 ? x01 : x23);
  y += dpp_pull<0x124>(y);  // row_ror:4
  y += dpp_pull<0x128>(y);  // row_ror:8
  return xor32_sum(xor16_sum(y));
}

// One halving step of a butterfly with the gfx950 swap instructions - no selects, no LDS crossbar.
// swap32_add: lanes < 32 get a[l] + a[l + 32], lanes >= 32 get b[l - 32] + b[l].
// swap16_add: even rows of 16 lanes get a summed over their row pair, odd rows b summed over theirs.
__device__ inline real swap32_add(real a, real b) {
#ifdef TJM_F32
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(a), __float_as_int(b), false, false);
  return __int_as_float(r[0]) + __int_as_float(r[1]);
#else
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
#endif
}
__device__ inline real swap16_add(real a, real b) {
#ifdef TJM_F32
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(a), __float_as_int(b), false, false);
  return __int_as_float(r[0]) + __int_as_float(r[1]);
#else
  const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
#endif
}
// Four wavefront sums, one per row of 16 lanes: on return every lane of row r (lanes 16 r ... 16 r + 15) holds the total of p_r.
__device__ inline real wave_sum4_rows(real p0, real p1, real p2, real p3) {
  return row_total(swap16_add(swap32_add(p0, p2), swap32_add(p1, p3)));
}

// fp64 reciprocal square root: hardware estimate + two Newton steps (full real precision)
__device__ inline real fast_rsqrt(real x) {
  const real half = 0.5;  // typed constants: a double literal would pull the complex64 build into fp64 arithmetic
  real y = tjm_rsq(x);
  const real h = half * x;
  y = fma(y, fma(-(h * y), y, half), y);  // y (1 + (1/2 - x y^2 / 2))
#ifndef TJM_F32  // (v_rsq_f32 is good to one ulp: ONE step brings it to rounding, and the step sits on the serial chain of every sub-step)
  y = fma(y, fma(-(h * y), y, half), y);
#endif
  return y;
}

// sqrt(x), x > 0: the same estimate refined by two coupled (Goldschmidt) steps on g -> sqrt(x), h -> 1 / (2 sqrt(x)); two
// instructions fewer than x * fast_rsqrt(x)
__device__ inline real fast_sqrt(real x) {
  const real half = 0.5;
  const real y = tjm_rsq(x);
  real g = x * y, h = half * y;
  real e = fma(-g, h, half);
#ifdef TJM_F32  // (one step from the one-ulp estimate of v_rsq_f32)
  return fma(g, e, g);
#else
  g = fma(g, e, g);
  h = fma(h, e, h);
  e = fma(-g, h, half);
  return fma(g, e, g);
#endif
}

// Decide and build the rotation for the column pair with norms (a, d) and inner product g.
// Returns true when a rotation is applied; (c, sr + i si) as in
//   y_p' = c y_p - conj(s) y_q ,  y_q' = s y_p + c y_q ,  a' = a - t|g| ,  d' = d + t|g|.
__device__ inline bool make_rotation(real a, real d, real gx, real gy, real tol2, real nfloor, real& c, real& sr,
                                     real& si, real& tg) {
  const real mag2 = fma(gx, gx, gy * gy);
  // rotate only if the pair is non-orthogonal at the tolerance level and neither column sits at the rounding-noise floor of the
  // matrix.  (No lower bound on the angle: a rotation far below 1 ulp of the large column still carries the correction that makes
  // a column ten decades smaller orthogonal to it.) (sigma < 1e-13 ||X||_F: such columns are
  // numerically null, carry no weight, and would otherwise be rotated against rounding noise for ever)
  if (!(mag2 > tol2 * a * d && a > nfloor && d > nfloor && mag2 > TJM_TINY)) return false;
  // With delta = (d - a) / 2, r = sqrt(delta^2 + |g|^2), u = |delta| + r  (so u^2 + |g|^2 = 2 r u):
  //   t = sgn(delta) |g| / u ,  c = u / sqrt(2 r u) ,  s = t c g / |g| = sgn(delta) g / sqrt(2 r u) ,  t |g| = sgn(delta) |g|^2 / u.
  // Two reciprocal square roots and no division; c^2 + |s|^2 = (u^2 + |g|^2) / (2 r u) = 1 to the rounding of r.
  // delta^2 is a FOURTH power of the entries: fine in fp64 for any sensible input; the complex64 build needs ||X||_F^2 within about
  // 1e-15 ... 1e19 (the lower end was already set by TJM_TINY) - the matrices here are slices of normalised states.
  const real delta = real(0.5) * (d - a);
  const real x = fma(delta, delta, mag2);
  const real r = fast_sqrt(x);
  const real u = fabs(delta) + r;
  const real q = fast_rsqrt((r + r) * u);                      // 1 / sqrt(2 r u)
  const real qs = (delta >= real(0.0)) ? q : -q;
  c = u * q;
  sr = qs * gx;
  si = qs * gy;
  tg = mag2 * ((r + r) * q) * qs;                              // sgn(delta) |g|^2 / u ,  1 / u = 2 r q^2
  return true;
}

// The same decision and rotation, evaluated per lane on the output of wave_sum4_rows: rows 0 and 1 carry (Re g, Im g) of the
// first pair, rows 2 and 3 of the second.  own = this lane's component of g; (a, d) the norms of its pair.
// Gives c, sv = s * own / |g| (Re s in the even rows, Im s in the odd rows) and tg; (1, 0, 0) when no rotation applies.  Returns the
// decision of this lane's pair.
__device__ inline bool make_rotation_lanes(real a, real d, real own, real tol2, real nfloor, real& c, real& sv, real& tg) {
  const real mag2 = xor16_sum(own * own);  // identical in both rows of the pair (addition commutes)
  const bool rot = mag2 > tol2 * a * d && a > nfloor && d > nfloor && mag2 > TJM_TINY;
  const real delta = real(0.5) * (d - a);  // the formulas of make_rotation
  const real x = fma(delta, delta, mag2);
  const real r = fast_sqrt(x);
  const real u = fabs(delta) + r;
  const real q = fast_rsqrt((r + r) * u);
  const real qs = (delta >= real(0.0)) ? q : -q;
  c = rot ? u * q : real(1.0);
  sv = rot ? qs * own : real(0.0);
  tg = rot ? mag2 * ((r + r) * q) * qs : real(0.0);
  return rot;
}

__device__ inline void rotate_pair(cplx& p, cplx& q, real c, real sr, real si) {
  // y_p' = c y_p - conj(s) y_q ; y_q' = s y_p + c y_q
  const real npx = fma(-si, q.y, fma(-sr, q.x, c * p.x));
  const real npy = fma(si, q.x, fma(-sr, q.y, c * p.y));
  const real nqx = fma(c, q.x, fma(-si, p.y, sr * p.x));
  const real nqy = fma(c, q.y, fma(si, p.x, sr * p.y));
  p = cplx{npx, npy};
  q = cplx{nqx, nqy};
}

// ---- cross pairs of one block pair -----------------------------------------------------------
template <int RK>
__global__ __launch_bounds__(512) void jacobi_cross_kernel(JacobiArgs g) {
  extern __shared__ real smem[];
  int b = blockIdx.y;
  if (g.ids) b = g.ids[b];
  if (g.done[b]) return;
  const int rtot = g.rtot, rx = g.rx;
  const int nrk = rtot >> 6;
  cplx* slots = reinterpret_cast<cplx*>(smem);                  // [8][rtot]
  real* sN = reinterpret_cast<real*>(slots + NB * rtot);    // [8]
  int* sCnt = reinterpret_cast<int*>(sN + NB);                  // [8]

  int I, J;
  if (g.mode == 1) { I = 2 * blockIdx.x; J = I + 1; }
  else pair_of(g.nblk, g.round, blockIdx.x, I, J);
  // visit pruning: a block with only zero columns never rotates; a pair found orthogonal stays so until one of
  // its blocks is modified again
  int* st = g.stamps + (long)b * STAMP_STRIDE;
  if (!st[2 * MAXBLK + I] || !st[2 * MAXBLK + J]) return;
  {
    const int ver = st[3 * MAXBLK + I * MAXBLK + J];
    if (ver > st[I] && ver > st[J]) return;
  }
  cplx* __restrict__ Yb = g.Y + (long)b * g.y_b0;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  cplx* colI = Yb + (long)(I * NB + w) * rtot;
  cplx* colJ = Yb + (long)(J * NB + w) * rtot;

  cplx yI[RK], yJ[RK];
#pragma unroll
  for (int k = 0; k < RK; ++k) {
    if (k < nrk) {
      yI[k] = colI[lane + 64 * k];
      yJ[k] = colJ[lane + 64 * k];
    } else {
      yI[k] = cplx{0.0, 0.0};
      yJ[k] = cplx{0.0, 0.0};
    }
  }
  real nI = 0.0, nJ = 0.0;
#pragma unroll
  for (int k = 0; k < RK; ++k) {
    if (k < nrk && lane + 64 * k < rx) {
      nI = fma(yI[k].x, yI[k].x, fma(yI[k].y, yI[k].y, nI));
      nJ = fma(yJ[k].x, yJ[k].x, fma(yJ[k].y, yJ[k].y, nJ));
    }
  }
  nI = wave_sum(nI);
  nJ = wave_sum(nJ);
  const real floor2 = g.floor_scale * g.fro2[b];
  int cnt = 0;
  for (int s = 0; s < NB; ++s) {
    real gx = 0.0, gy = 0.0;
#pragma unroll
    for (int k = 0; k < RK; ++k) {
      if (k < nrk && lane + 64 * k < rx) {
        gx = fma(yI[k].x, yJ[k].x, fma(yI[k].y, yJ[k].y, gx));   // conj(yI) * yJ
        gy = fma(yI[k].x, yJ[k].y, fma(-yI[k].y, yJ[k].x, gy));
      }
    }
    gx = wave_sum(gx);
    gy = wave_sum(gy);
    real c, sr, si, tg;
    if (make_rotation(nI, nJ, gx, gy, g.tol2, floor2, c, sr, si, tg)) {
#pragma unroll
      for (int k = 0; k < RK; ++k)
        if (k < nrk) rotate_pair(yI[k], yJ[k], c, sr, si);
      nI -= tg;
      nJ += tg;
      ++cnt;
    }
    if (s + 1 < NB) {
      // hand the J column to the previous wavefront: wave w next needs the column held by wave w+1
#pragma unroll
      for (int k = 0; k < RK; ++k)
        if (k < nrk) slots[w * rtot + lane + 64 * k] = yJ[k];
      if (lane == 0) sN[w] = nJ;
      __syncthreads();
      const int src = (w + 1) & (NB - 1);
#pragma unroll
      for (int k = 0; k < RK; ++k)
        if (k < nrk) yJ[k] = slots[src * rtot + lane + 64 * k];
      nJ = sN[src];
      __syncthreads();
    }
  }
  if (lane == 0) sCnt[w] = cnt;
  __syncthreads();
  int total = 0;
#pragma unroll
  for (int q = 0; q < NB; ++q) total += sCnt[q];
  if (tid == 0 && g.work) atomicAdd(g.work, NB * NB);
  if (total == 0) {  // nothing rotated: memory is already up to date, remember the pair as verified
    if (tid == 0) st[3 * MAXBLK + I * MAXBLK + J] = g.clock;
    return;
  }
  if (tid == 0) { st[I] = g.clock; st[J] = g.clock; }
  // after 7 hand-overs wave w holds J column (w + 7) mod 8
  cplx* outJ = Yb + (long)(J * NB + ((w + NB - 1) & (NB - 1))) * rtot;
#pragma unroll
  for (int k = 0; k < RK; ++k) {
    if (k < nrk) {
      colI[lane + 64 * k] = yI[k];
      outJ[lane + 64 * k] = yJ[k];
    }
  }
  if (tid == 0) atomicAdd(&g.nrot[b], total);
}

// ---- cross pairs of a pair of 16-column blocks ---------------------------------------------------
// Same scheme with twice the tile: every wavefront keeps TWO columns of block I and TWO of block J, so each LDS
// hand-over (and its two barriers) is followed by four rotations, the two independent ones back to back, and a
// column is read from / written to memory once per 256 rotations of the tile instead of once per 64.
template <int RK>
__global__ __launch_bounds__(512) void jacobi_cross16_kernel(JacobiArgs g) {
  extern __shared__ real smem[];
  int b = blockIdx.y;
  if (g.ids) b = g.ids[b];
  if (g.done[b]) return;
  const int rtot = g.rtot, rx = g.rx;
  const int nrk = rtot >> 6;
  cplx* slots = reinterpret_cast<cplx*>(smem);                      // [8][2][rtot]
  real* sN = reinterpret_cast<real*>(slots + 2 * NB * rtot);    // [8][2]
  int* sCnt = reinterpret_cast<int*>(sN + 2 * NB);                  // [8]
  int I, J;
  pair_of(g.nblk / 2, g.round, blockIdx.x, I, J);                   // indices of 16-column blocks
  int* st = g.stamps + (long)b * STAMP_STRIDE;
  {
    const int nzI = st[2 * MAXBLK + 2 * I] | st[2 * MAXBLK + 2 * I + 1];
    const int nzJ = st[2 * MAXBLK + 2 * J] | st[2 * MAXBLK + 2 * J + 1];
    if (!nzI || !nzJ) return;
    const int ver = st[3 * MAXBLK + (2 * I) * MAXBLK + 2 * J];
    const int m1 = max(max(st[2 * I], st[2 * I + 1]), max(st[2 * J], st[2 * J + 1]));
    if (ver > m1) return;
  }
  cplx* __restrict__ Yb = g.Y + (long)b * g.y_b0;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  cplx yI[2][RK], yJ[2][RK];
  real nI[2] = {0.0, 0.0}, nJ[2] = {0.0, 0.0};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const cplx* cI = Yb + (long)(I * 16 + 2 * w + h) * rtot;
    const cplx* cJ = Yb + (long)(J * 16 + 2 * w + h) * rtot;
#pragma unroll
    for (int k = 0; k < RK; ++k) {
      if (k < nrk) {
        yI[h][k] = cI[lane + 64 * k];
        yJ[h][k] = cJ[lane + 64 * k];
      } else {
        yI[h][k] = cplx{0.0, 0.0};
        yJ[h][k] = cplx{0.0, 0.0};
      }
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int k = 0; k < RK; ++k) {
      if (k < nrk && lane + 64 * k < rx) {
        nI[h] = fma(yI[h][k].x, yI[h][k].x, fma(yI[h][k].y, yI[h][k].y, nI[h]));
        nJ[h] = fma(yJ[h][k].x, yJ[h][k].x, fma(yJ[h][k].y, yJ[h][k].y, nJ[h]));
      }
    }
  }
  nI[0] = wave_sum(nI[0]); nI[1] = wave_sum(nI[1]);
  nJ[0] = wave_sum(nJ[0]); nJ[1] = wave_sum(nJ[1]);
  const real floor2 = g.floor_scale * g.fro2[b];
  int cnt = 0;
  for (int s = 0; s < NB; ++s) {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      // sub 0: (I0,J0) and (I1,J1) ; sub 1: (I0,J1) and (I1,J0) -- the two pairs of a sub-step are independent
      real gx[2] = {0.0, 0.0}, gy[2] = {0.0, 0.0};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int hj = h ^ sub;
#pragma unroll
        for (int k = 0; k < RK; ++k) {
          if (k < nrk && lane + 64 * k < rx) {
            gx[h] = fma(yI[h][k].x, yJ[hj][k].x, fma(yI[h][k].y, yJ[hj][k].y, gx[h]));
            gy[h] = fma(yI[h][k].x, yJ[hj][k].y, fma(-yI[h][k].y, yJ[hj][k].x, gy[h]));
          }
        }
      }
      gx[0] = wave_sum(gx[0]); gy[0] = wave_sum(gy[0]);
      gx[1] = wave_sum(gx[1]); gy[1] = wave_sum(gy[1]);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int hj = h ^ sub;
        real c, sr, si, tg;
        if (make_rotation(nI[h], nJ[hj], gx[h], gy[h], g.tol2, floor2, c, sr, si, tg)) {
#pragma unroll
          for (int k = 0; k < RK; ++k)
            if (k < nrk) rotate_pair(yI[h][k], yJ[hj][k], c, sr, si);
          nI[h] -= tg;
          nJ[hj] += tg;
          ++cnt;
        }
      }
    }
    if (s + 1 < NB) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int k = 0; k < RK; ++k)
          if (k < nrk) slots[(w * 2 + h) * rtot + lane + 64 * k] = yJ[h][k];
      if (lane < 2) sN[w * 2 + lane] = (lane == 0) ? nJ[0] : nJ[1];
      __syncthreads();
      const int src = (w + 1) & (NB - 1);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int k = 0; k < RK; ++k)
          if (k < nrk) yJ[h][k] = slots[(src * 2 + h) * rtot + lane + 64 * k];
      nJ[0] = sN[src * 2];
      nJ[1] = sN[src * 2 + 1];
      __syncthreads();
    }
  }
  if (lane == 0) sCnt[w] = cnt;
  __syncthreads();
  int total = 0;
#pragma unroll
  for (int q = 0; q < NB; ++q) total += sCnt[q];
  if (tid == 0 && g.work) atomicAdd(g.work, REC_PER_VISIT);
  if (total == 0) {
    if (tid == 0) st[3 * MAXBLK + (2 * I) * MAXBLK + 2 * J] = g.clock;
    return;
  }
  if (tid == 0) { st[2 * I] = g.clock; st[2 * I + 1] = g.clock; st[2 * J] = g.clock; st[2 * J + 1] = g.clock; }
  const int wj = (w + NB - 1) & (NB - 1);  // after 7 hand-overs wave w holds the J column pair of wave w-1
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    cplx* oI = Yb + (long)(I * 16 + 2 * w + h) * rtot;
    cplx* oJ = Yb + (long)(J * 16 + 2 * wj + h) * rtot;
#pragma unroll
    for (int k = 0; k < RK; ++k) {
      if (k < nrk) {
        oI[lane + 64 * k] = yI[h][k];
        oJ[lane + 64 * k] = yJ[h][k];
      }
    }
  }
  if (tid == 0) atomicAdd(&g.nrot[b], total);
}

// ---- split scheme: rotate the X rows and RECORD the rotations, replay them on the W rows ------------
// The dot products only involve the X rows; the accumulated unitary W just follows.  Splitting the tile at the
// X / W boundary halves the registers and LDS of the latency-bound kernel (2 workgroups per CU) and turns the W
// half into a pure FMA stream: one lane per row, all 32 columns of the tile in registers, 256 rotations with
// wave-uniform parameters and compile-time column indices.

// One column of the X part in the registers of a wavefront: XRK row groups of 64 rows, lane = row within its group.  The kernel
// below is written against these operations; the two arithmetic types differ in how the registers hold the column.
template <int XRK>
struct ColFrag {
#ifdef TJM_F32
  // complex64: the real parts of two row groups share one register pair and so do the imaginary parts, every operation is a
  // v_pk_*_f32 on both row groups at once with no swizzle (half the issue slots of the scalar form).  An odd XRK carries a zero
  // row group, which contributes nothing to a dot product and rotates into zero.
  typedef float f2 __attribute__((ext_vector_type(2)));
  static constexpr int NP = (XRK + 1) / 2;
  static constexpr int LDS_REALS = 4 * 64 * NP;  // per column in the hand-over buffer
  f2 re[NP], im[NP];
  static __device__ inline f2 pk(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
  __device__ inline void load(const cplx* __restrict__ col, int lane) {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const cplx a = col[lane + 128 * k];
      const cplx b = (2 * k + 1 < XRK) ? col[lane + 128 * k + 64] : cplx{0.0f, 0.0f};
      re[k] = f2{a.x, b.x};
      im[k] = f2{a.y, b.y};
    }
  }
  __device__ inline void store(cplx* __restrict__ col, int lane) const {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      col[lane + 128 * k] = cplx{re[k].x, im[k].x};
      if (2 * k + 1 < XRK) col[lane + 128 * k + 64] = cplx{re[k].y, im[k].y};
    }
  }
  __device__ inline real norm2() const {
    f2 n = {0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < NP; ++k) n = pk(re[k], re[k], pk(im[k], im[k], n));
    return n.x + n.y;
  }
  // this lane's share of <p, q> = sum conj(p) q
  static __device__ inline void dot(const ColFrag& p, const ColFrag& q, real& gx, real& gy) {
    f2 x = {0.0f, 0.0f}, y = {0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      x = pk(p.re[k], q.re[k], pk(p.im[k], q.im[k], x));
      y = pk(p.re[k], q.im[k], pk(-p.im[k], q.re[k], y));
    }
    gx = x.x + x.y;
    gy = y.x + y.y;
  }
  // p' = c p - conj(s) q ; q' = s p + c q
  static __device__ inline void rotate(ColFrag& p, ColFrag& q, real c, real sr, real si) {
    const f2 C = {c, c}, SR = {sr, sr}, SI = {si, si};
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const f2 px = pk(-SI, q.im[k], pk(-SR, q.re[k], C * p.re[k]));
      const f2 py = pk(SI, q.re[k], pk(-SR, q.im[k], C * p.im[k]));
      const f2 qx = pk(C, q.re[k], pk(-SI, p.im[k], SR * p.re[k]));
      const f2 qy = pk(C, q.im[k], pk(SI, p.re[k], SR * p.im[k]));
      p.re[k] = px; p.im[k] = py; q.re[k] = qx; q.im[k] = qy;
    }
  }
  __device__ inline void to_lds(real* slot, int lane) const {
    f2* s2 = reinterpret_cast<f2*>(slot);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      s2[(2 * k) * 64 + lane] = re[k];
      s2[(2 * k + 1) * 64 + lane] = im[k];
    }
  }
  __device__ inline void from_lds(const real* slot, int lane) {
    const f2* s2 = reinterpret_cast<const f2*>(slot);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      re[k] = s2[(2 * k) * 64 + lane];
      im[k] = s2[(2 * k + 1) * 64 + lane];
    }
  }
#else
  static constexpr int LDS_REALS = 2 * 64 * XRK;
  cplx y[XRK];
  __device__ inline void load(const cplx* __restrict__ col, int lane) {
#pragma unroll
    for (int k = 0; k < XRK; ++k) y[k] = col[lane + 64 * k];
  }
  __device__ inline void store(cplx* __restrict__ col, int lane) const {
#pragma unroll
    for (int k = 0; k < XRK; ++k) col[lane + 64 * k] = y[k];
  }
  __device__ inline real norm2() const {
    real n = 0.0;
#pragma unroll
    for (int k = 0; k < XRK; ++k) n = fma(y[k].x, y[k].x, fma(y[k].y, y[k].y, n));
    return n;
  }
  static __device__ inline void dot(const ColFrag& p, const ColFrag& q, real& gx, real& gy) {
    gx = 0.0;
    gy = 0.0;
#pragma unroll
    for (int k = 0; k < XRK; ++k) {
      gx = fma(p.y[k].x, q.y[k].x, fma(p.y[k].y, q.y[k].y, gx));
      gy = fma(p.y[k].x, q.y[k].y, fma(-p.y[k].y, q.y[k].x, gy));
    }
  }
  static __device__ inline void rotate(ColFrag& p, ColFrag& q, real c, real sr, real si) {
#pragma unroll
    for (int k = 0; k < XRK; ++k) rotate_pair(p.y[k], q.y[k], c, sr, si);
  }
  __device__ inline void to_lds(real* slot, int lane) const {
    cplx* sc = reinterpret_cast<cplx*>(slot);
#pragma unroll
    for (int k = 0; k < XRK; ++k) sc[lane + 64 * k] = y[k];
  }
  __device__ inline void from_lds(const real* slot, int lane) {
    const cplx* sc = reinterpret_cast<const cplx*>(slot);
#pragma unroll
    for (int k = 0; k < XRK; ++k) y[k] = sc[lane + 64 * k];
  }
#endif
};

// Columns of a 16-column block that wavefront w holds in tournament round tr (circle method on 16 players, player 15 fixed): over
// the 15 rounds every pair of the block is the pair of exactly one wavefront.  tr < 0: the fixed assignment (2 w, 2 w + 1).
__device__ inline int fold_column(int tr, int w, int h) {
  if (tr < 0) return 2 * w + h;
  if (w == 0) return h == 0 ? tr : 15;
  return h == 0 ? (tr + w) % 15 : (tr - w + 15) % 15;
}

// XRK = row groups of 64 of the X part held in registers: 1 ... 8 (rx_top = 64 XRK; 4 at d*chi = 256)
//
// g.fold (matrices of at least 16 blocks, no accumulated unitary): the pairs INSIDE a 16-column block ride along.  In round r < 15
// the wavefronts hold the columns of both blocks in the pairing of tournament round r, and one extra sub-step rotates the two
// in-wavefront pairs (I_a, I_b), (J_a, J_b) before the J columns start to circulate: the 15 rounds of a sweep visit all 120 pairs
// of every block exactly once, with no launch, no load and no store of their own (the separate diag / sibling kernels cost two
// launches per sweep, each about as long as a round of this kernel).  The tile's verification stamp then also stands for those
// pairs; a block whose partner of the round is all zero still gets its in-block sub-step.
//
// LATE: the variant for the last sweeps (the host launches it once a sweep has rotated less than 70 % of its pairs): a sub-step
// whose two pairs are both orthogonal already - nearly all of them by then - stops after the inner products and the decision
// instead of applying two identity rotations.
template <int XRK, bool LATE>
__global__ __launch_bounds__(512, (XRK <= 4) ? 4 : 2) void jacobi_cross16x_kernel(JacobiArgs g) {
  extern __shared__ real smem[];
  int b = blockIdx.y;
  if (g.ids) b = g.ids[b];
  const bool record = g.rec != nullptr;  // without accumulation nothing replays the rotations
  real* rec = record ? g.rec + ((long)b * gridDim.x + blockIdx.x) * (REC_PER_VISIT * 4) : nullptr;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (g.done[b]) { if (tid == 0 && record) rec[3] = 0.0; return; }
  const int rtot = g.rtot;
  typedef ColFrag<XRK> Col;
  real* slots = smem;                                               // [8][2] columns of Col::LDS_REALS
  real* sN = slots + 2 * NB * Col::LDS_REALS;                       // [8][2]
  int* sCnt = reinterpret_cast<int*>(sN + 2 * NB);                  // [8]
  int I, J;
  pair_of(g.nblk / 2, g.round, blockIdx.x, I, J);
  int* st = g.stamps + (long)b * STAMP_STRIDE;
  const int tr = (g.fold && g.round < 15) ? g.round : -1;           // tournament round of the in-block pairs (-1: none in this visit)
  bool cross;
  {
    const int nzI = st[2 * MAXBLK + 2 * I] | st[2 * MAXBLK + 2 * I + 1];
    const int nzJ = st[2 * MAXBLK + 2 * J] | st[2 * MAXBLK + 2 * J + 1];
    const int ver = st[3 * MAXBLK + (2 * I) * MAXBLK + 2 * J];
    const int m1 = max(max(st[2 * I], st[2 * I + 1]), max(st[2 * J], st[2 * J + 1]));
    cross = nzI && nzJ;
    if ((!cross && (tr < 0 || (!nzI && !nzJ))) || ver > m1) { if (tid == 0 && record) rec[3] = 0.0; return; }
  }
  cplx* __restrict__ Yb = g.Y + (long)b * g.y_b0;
  Col yI[2], yJ[2];
  real nI[2], nJ[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    yI[h].load(Yb + (long)(I * 16 + fold_column(tr, w, h)) * rtot, lane);
    yJ[h].load(Yb + (long)(J * 16 + fold_column(tr, w, h)) * rtot, lane);
    nI[h] = yI[h].norm2();
    nJ[h] = yJ[h].norm2();
  }
  nI[0] = wave_sum(nI[0]); nI[1] = wave_sum(nI[1]);
  nJ[0] = wave_sum(nJ[0]); nJ[1] = wave_sum(nJ[1]);
  const real floor2 = g.floor_scale * g.fro2[b];
  int cnt = 0;
  // One sub-step: the independent pairs (p0, q0) and (p1, q1), norms (a0, d0), (a1, d1).  Both inner products are reduced together
  // (row r of 16 lanes ends up with component r of (Re g0, Im g0, Re g1, Im g1)), both rotations are set up side by side - pair h in
  // half-wave h - and applied unconditionally: (c, s) = (1, 0) leaves the columns bit-identical, and in the sweeps that matter
  // almost every pair rotates; a branch only bought register copies at its merge point.
  auto sub_step = [&](Col& p0, Col& q0, real& a0, real& d0, Col& p1, Col& q1, real& a1, real& d1, int rec_slot) {
    real gx[2], gy[2];
    Col::dot(p0, q0, gx[0], gy[0]);
    Col::dot(p1, q1, gx[1], gy[1]);
    const real gsum = wave_sum4_rows(gx[0], gy[0], gx[1], gy[1]);
    const bool second = lane & 32;
    real cv, sv, tv;
    const bool rot = make_rotation_lanes(second ? a1 : a0, second ? d1 : d0, gsum, g.tol2, floor2, cv, sv, tv);
    const unsigned long long both = __ballot(rot) & 0x100000001ull;  // lanes 0 and 32 speak for the two pairs (scalar unit: the mask is in SGPRs)
    cnt += __popcll(both);
    if (LATE && !record && both == 0) return;
    if (record && rec_slot >= 0 && (lane & 15) == 0) {
      real* r4 = rec + (((rec_slot * NB + w) * 2 + (lane >> 5)) * 4);
      if (lane & 16) r4[2] = sv;
      else { r4[0] = cv; r4[1] = sv; }
    }
    {
      const real sr = lane_value(sv, 0), si = lane_value(sv, 16), c = lane_value(cv, 0), tg = lane_value(tv, 0);
      Col::rotate(p0, q0, c, sr, si);
      a0 -= tg;
      d0 += tg;
    }
    {
      const real sr = lane_value(sv, 32), si = lane_value(sv, 48), c = lane_value(cv, 32), tg = lane_value(tv, 32);
      Col::rotate(p1, q1, c, sr, si);
      a1 -= tg;
      d1 += tg;
    }
  };
  if (tr >= 0) sub_step(yI[0], yI[1], nI[0], nI[1], yJ[0], yJ[1], nJ[0], nJ[1], -1);  // the in-block pairs of this round
  if (cross) {
    for (int s = 0; s < NB; ++s) {
      sub_step(yI[0], yJ[0], nI[0], nJ[0], yI[1], yJ[1], nI[1], nJ[1], s * 2);
      sub_step(yI[0], yJ[1], nI[0], nJ[1], yI[1], yJ[0], nI[1], nJ[0], s * 2 + 1);
      if (s + 1 < NB) {
#pragma unroll
        for (int h = 0; h < 2; ++h) yJ[h].to_lds(slots + (w * 2 + h) * Col::LDS_REALS, lane);
        if (lane < 2) sN[w * 2 + lane] = (lane == 0) ? nJ[0] : nJ[1];
        __syncthreads();
        const int src = (w + 1) & (NB - 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) yJ[h].from_lds(slots + (src * 2 + h) * Col::LDS_REALS, lane);
        nJ[0] = sN[src * 2];
        nJ[1] = sN[src * 2 + 1];
        __syncthreads();
      }
    }
  }
  if (lane == 0) sCnt[w] = cnt;
  __syncthreads();
  int total = 0;
#pragma unroll
  for (int q = 0; q < NB; ++q) total += sCnt[q];
  if (tid == 0 && record) rec[3] = (total > 0) ? 1.0 : 0.0;  // flag slot of the first record: does the W half have work
  if (tid == 0 && g.work) atomicAdd(g.work, (cross ? REC_PER_VISIT : 0) + (tr >= 0 ? 2 * NB : 0));
  if (total == 0) {
    if (tid == 0) st[3 * MAXBLK + (2 * I) * MAXBLK + 2 * J] = g.clock;
    return;
  }
  if (tid == 0) { st[2 * I] = g.clock; st[2 * I + 1] = g.clock; st[2 * J] = g.clock; st[2 * J + 1] = g.clock; }
  const int wj = cross ? ((w + NB - 1) & (NB - 1)) : w;  // after 7 hand-overs wave w holds the J column pair of wave w - 1
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    yI[h].store(Yb + (long)(I * 16 + fold_column(tr, w, h)) * rtot, lane);
    yJ[h].store(Yb + (long)(J * 16 + fold_column(tr, wj, h)) * rtot, lane);
  }
  if (tid == 0) atomicAdd(&g.nrot[b], total);
}

// ---- the same visit with FOUR columns of each block per wavefront (4 wavefronts per tile) ---------------------------------------
// The tile kernel above is bound by instruction issue, and of the ~150 instructions of a sub-step only ~56 are the inner products
// and rotations of its two pairs; the rest - the cross-lane reduction, the rotation set-up, the broadcasts, the hand-over - costs the
// same whether a sub-step carries two pairs or four.  Here a wavefront holds four columns of I and four of J: a step is four
// sub-steps of four independent pairs (the Latin square I_h x J_(h+t)), eight inner-product components are reduced together
// (wave_sum8_groups: group g of 8 lanes ends up with component g), the four rotations are set up side by side in the four rows of 16
// lanes, and the J quads circulate over 4 wavefronts in 3 hand-overs instead of 7.  Same pairs per visit (256 + 16 in-block), same
// rotation formulas; the order of the rotations inside a visit differs, so the iterates differ from the two-column kernel's in the
// last bits (both are deterministic).  Used without rotation record (X-only solves: the complex64 phase of the mixed split).

// Eight wavefront sums: on return every lane of group g = lane / 8 holds the total of p_g.
__device__ inline real wave_sum8_groups(real p0, real p1, real p2, real p3, real p4, real p5, real p6, real p7, int lane) {
  // halves: lower lanes keep p_k, upper lanes p_(k+4); rows: row 0 / 1 keep (p_j, p_(j+2)), rows 2 / 3 (p_(j+4), p_(j+6))
  const real y0 = swap16_add(swap32_add(p0, p4), swap32_add(p2, p6));  // rows: p0 p2 p4 p6
  const real y1 = swap16_add(swap32_add(p1, p5), swap32_add(p3, p7));  // rows: p1 p3 p5 p7
  const bool hi8 = lane & 8;
  real z = (hi8 ? y1 : y0) + dpp_pull<0x128>(hi8 ? y0 : y1);  // row_ror:8: the two halves of a row of 16 trade what the other one keeps
  z += dpp_pull<0x141>(z);  // row_half_mirror
  z += dpp_pull<0xB1>(z);   // quad_perm [1,0,3,2]
  z += dpp_pull<0x4E>(z);   // quad_perm [2,3,0,1]
  return z;
}

// make_rotation_lanes for four pairs: pair = lane / 16, groups of 8 lanes carry (Re g, Im g) of their pair
__device__ inline bool make_rotation_lanes4(real a, real d, real own, real tol2, real nfloor, real& c, real& sv, real& tg) {
  const real o2 = own * own;
  const real mag2 = o2 + dpp_pull<0x128>(o2);
  const bool rot = mag2 > tol2 * a * d && a > nfloor && d > nfloor && mag2 > TJM_TINY;
  const real delta = real(0.5) * (d - a);
  const real x = fma(delta, delta, mag2);
  const real r = fast_sqrt(x);
  const real u = fabs(delta) + r;
  const real q = fast_rsqrt((r + r) * u);
  const real qs = (delta >= real(0.0)) ? q : -q;
  c = rot ? u * q : real(1.0);
  sv = rot ? qs * own : real(0.0);
  tg = rot ? mag2 * ((r + r) * q) * qs : real(0.0);
  return rot;
}

// columns of a 16-column block that wavefront w (of 4) holds in tournament round tr: the pairs 2 w and 2 w + 1 of the round
__device__ inline int fold_column4(int tr, int w, int h) {
  if (tr < 0) return 4 * w + h;
  const int k = 2 * w + (h >> 1);
  if (k == 0) return (h & 1) == 0 ? tr : 15;
  return (h & 1) == 0 ? (tr + k) % 15 : (tr - k + 15) % 15;
}

template <int XRK, bool LATE>
__global__ __launch_bounds__(256, (XRK <= 4) ? 4 : 2) void jacobi_cross16q_kernel(JacobiArgs g) {
  extern __shared__ real smem[];
  int b = blockIdx.y;
  if (g.ids) b = g.ids[b];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (g.done[b]) return;
  const int rtot = g.rtot;
  typedef ColFrag<XRK> Col;
  real* slots = smem;                                    // [4][4] columns of Col::LDS_REALS
  real* sN = slots + 16 * Col::LDS_REALS;                // [4][4]
  int* sCnt = reinterpret_cast<int*>(sN + 16);           // [4]
  int I, J;
  pair_of(g.nblk / 2, g.round, blockIdx.x, I, J);
  int* st = g.stamps + (long)b * STAMP_STRIDE;
  const int tr = (g.fold && g.round < 15) ? g.round : -1;
  bool cross;
  {
    const int nzI = st[2 * MAXBLK + 2 * I] | st[2 * MAXBLK + 2 * I + 1];
    const int nzJ = st[2 * MAXBLK + 2 * J] | st[2 * MAXBLK + 2 * J + 1];
    const int ver = st[3 * MAXBLK + (2 * I) * MAXBLK + 2 * J];
    const int m1 = max(max(st[2 * I], st[2 * I + 1]), max(st[2 * J], st[2 * J + 1]));
    cross = nzI && nzJ;
    if ((!cross && (tr < 0 || (!nzI && !nzJ))) || ver > m1) return;
  }
  cplx* __restrict__ Yb = g.Y + (long)b * g.y_b0;
  Col yI[4], yJ[4];
  real nI[4], nJ[4];
  {
    real pn[8];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      yI[h].load(Yb + (long)(I * 16 + fold_column4(tr, w, h)) * rtot, lane);
      yJ[h].load(Yb + (long)(J * 16 + fold_column4(tr, w, h)) * rtot, lane);
      pn[h] = yI[h].norm2();
      pn[4 + h] = yJ[h].norm2();
    }
    const real z = wave_sum8_groups(pn[0], pn[1], pn[2], pn[3], pn[4], pn[5], pn[6], pn[7], lane);
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      nI[h] = lane_value(z, 8 * h);
      nJ[h] = lane_value(z, 32 + 8 * h);
    }
  }
  const real floor2 = g.floor_scale * g.fro2[b];
  int cnt = 0;
  // one sub-step: four independent pairs (p_i, q_i) with squared norms (a_i, d_i); applied unconditionally ((c, s) = (1, 0) leaves a
  // pair bit-identical), LATE: nothing is applied when none of the four needs it
  auto sub_step = [&](Col& p0, Col& q0, real& a0, real& d0, Col& p1, Col& q1, real& a1, real& d1, Col& p2, Col& q2, real& a2, real& d2, Col& p3, Col& q3,
                      real& a3, real& d3) {
    real gx[4], gy[4];
    Col::dot(p0, q0, gx[0], gy[0]);
    Col::dot(p1, q1, gx[1], gy[1]);
    Col::dot(p2, q2, gx[2], gy[2]);
    Col::dot(p3, q3, gx[3], gy[3]);
    const real gsum = wave_sum8_groups(gx[0], gy[0], gx[1], gy[1], gx[2], gy[2], gx[3], gy[3], lane);
    const int pr = lane >> 4;
    const real a = pr == 0 ? a0 : pr == 1 ? a1 : pr == 2 ? a2 : a3;
    const real d = pr == 0 ? d0 : pr == 1 ? d1 : pr == 2 ? d2 : d3;
    real cv, sv, tv;
    const bool rot = make_rotation_lanes4(a, d, gsum, g.tol2, floor2, cv, sv, tv);
    const unsigned long long any = __ballot(rot) & 0x0001000100010001ull;  // lanes 0, 16, 32, 48 speak for the four pairs
    cnt += __popcll(any);
    if (LATE && any == 0) return;
    {
      const real sr = lane_value(sv, 0), si = lane_value(sv, 8), c = lane_value(cv, 0), tg = lane_value(tv, 0);
      Col::rotate(p0, q0, c, sr, si);
      a0 -= tg; d0 += tg;
    }
    {
      const real sr = lane_value(sv, 16), si = lane_value(sv, 24), c = lane_value(cv, 16), tg = lane_value(tv, 16);
      Col::rotate(p1, q1, c, sr, si);
      a1 -= tg; d1 += tg;
    }
    {
      const real sr = lane_value(sv, 32), si = lane_value(sv, 40), c = lane_value(cv, 32), tg = lane_value(tv, 32);
      Col::rotate(p2, q2, c, sr, si);
      a2 -= tg; d2 += tg;
    }
    {
      const real sr = lane_value(sv, 48), si = lane_value(sv, 56), c = lane_value(cv, 48), tg = lane_value(tv, 48);
      Col::rotate(p3, q3, c, sr, si);
      a3 -= tg; d3 += tg;
    }
  };
  if (tr >= 0)  // the in-block pairs of this tournament round: two of I and two of J
    sub_step(yI[0], yI[1], nI[0], nI[1], yI[2], yI[3], nI[2], nI[3], yJ[0], yJ[1], nJ[0], nJ[1], yJ[2], yJ[3], nJ[2], nJ[3]);
  if (cross) {
    for (int s = 0; s < 4; ++s) {
      sub_step(yI[0], yJ[0], nI[0], nJ[0], yI[1], yJ[1], nI[1], nJ[1], yI[2], yJ[2], nI[2], nJ[2], yI[3], yJ[3], nI[3], nJ[3]);
      sub_step(yI[0], yJ[1], nI[0], nJ[1], yI[1], yJ[2], nI[1], nJ[2], yI[2], yJ[3], nI[2], nJ[3], yI[3], yJ[0], nI[3], nJ[0]);
      sub_step(yI[0], yJ[2], nI[0], nJ[2], yI[1], yJ[3], nI[1], nJ[3], yI[2], yJ[0], nI[2], nJ[0], yI[3], yJ[1], nI[3], nJ[1]);
      sub_step(yI[0], yJ[3], nI[0], nJ[3], yI[1], yJ[0], nI[1], nJ[0], yI[2], yJ[1], nI[2], nJ[1], yI[3], yJ[2], nI[3], nJ[2]);
      if (s + 1 < 4) {
#pragma unroll
        for (int h = 0; h < 4; ++h) yJ[h].to_lds(slots + (w * 4 + h) * Col::LDS_REALS, lane);
        if (lane < 4) sN[w * 4 + lane] = lane == 0 ? nJ[0] : lane == 1 ? nJ[1] : lane == 2 ? nJ[2] : nJ[3];
        __syncthreads();
        const int src = (w + 1) & 3;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          yJ[h].from_lds(slots + (src * 4 + h) * Col::LDS_REALS, lane);
          nJ[h] = sN[src * 4 + h];
        }
        __syncthreads();
      }
    }
  }
  if (lane == 0) sCnt[w] = cnt;
  __syncthreads();
  const int total = sCnt[0] + sCnt[1] + sCnt[2] + sCnt[3];
  if (tid == 0 && g.work) atomicAdd(g.work, (cross ? REC_PER_VISIT : 0) + (tr >= 0 ? 2 * NB : 0));
  if (total == 0) {
    if (tid == 0) st[3 * MAXBLK + (2 * I) * MAXBLK + 2 * J] = g.clock;
    return;
  }
  if (tid == 0) { st[2 * I] = g.clock; st[2 * I + 1] = g.clock; st[2 * J] = g.clock; st[2 * J + 1] = g.clock; }
  const int wj = cross ? ((w + 3) & 3) : w;  // after 3 hand-overs wavefront w holds the J quad of wavefront w - 1
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    yI[h].store(Yb + (long)(I * 16 + fold_column4(tr, w, h)) * rtot, lane);
    yJ[h].store(Yb + (long)(J * 16 + fold_column4(tr, wj, h)) * rtot, lane);
  }
  if (tid == 0) atomicAdd(&g.nrot[b], total);
}

// ---- THREE tournament rounds per load: four 16-column blocks per workgroup ---------------------------------------------------------
// jacobi_cross16q_kernel reads and writes a tile of 32 columns for one round of the tournament: 15 trips of the whole matrix through
// HBM per sweep, and by its arithmetic intensity (14 flop per byte) the kernel sits on the memory side of the fp32 ridge.  Here a
// workgroup of 8 wavefronts holds FOUR blocks (A, B, C, D) and plays the three rounds among them - (A,B)(C,D), (A,C)(B,D),
// (A,D)(B,C), each by two groups of four wavefronts exactly as two workgroups of the tile kernel would - with one block changing
// groups through LDS between the rounds: one load and one store per THREE rounds.  A sweep of a 256-column matrix is 5 launches of
// 4 such quads: the 16 blocks are the points of the affine plane AG(2,4), whose 20 lines (5 parallel classes of 4 disjoint lines)
// contain every pair of points exactly once - so every block pair meets in exactly one quad of one class, as in one round of the
// circle method.  The pairs INSIDE a block ride along as before; their schedule is the same plane on the 16 columns of a block:
// in the launch of class c wavefront w of a group holds the four columns of line (c, w), and the three rounds play the three
// perfect matchings of those four columns - no column changes wavefronts inside a launch, and the 5 x 3 rounds of a sweep visit all
// 120 pairs of the block once.  Pruning stamps, zero blocks, the rotation rule and the LATE variant are those of the tile kernel
// (clock + t stands for the launch counter of round t).
__device__ inline int gf4_mul(int a, int b) { return (0x9C78E400u >> (2 * (4 * a + b))) & 3; }  // GF(4) = {0, 1, x, x + 1}, 2 bits per product
// k-th point (0 ... 3) of line `line` of parallel class `cls` (0 ... 4) of AG(2,4); points are numbered 4 x + y
__device__ inline int ag_point(int cls, int line, int k) { return cls == 4 ? 4 * line + k : 4 * k + (gf4_mul(cls, k) ^ line); }

//
// Matrices of more than 256 columns (round 6: 512 x 512 is the two-site split of chi = 256, BASELINE config 3).  No resolvable design
// with lines of four covers 32 points, so the blocks are taken in GROUPS of 16 (256 columns each) and a sweep has two kinds of launches:
//   mode 0  the plane inside every group, all groups side by side (5 launches x 3 rounds: every pair of blocks of one group, and the
//           120 in-block pairs of every block, exactly as above);
//   mode 1  two groups against each other: a quad takes the blocks (2 i, 2 i + 1) of group a and (2 j, 2 j + 1) of group b,
//           j = i + shift mod 8, as A = a_2i, B = b_2j, C = b_2j+1, D = a_2i+1 and plays only the first TWO rounds - (A,B)(C,D) and
//           (A,C)(B,D), the four cross tiles; the third, (A,D)(B,C), lies inside the groups and belongs to mode 0.  8 shifts visit
//           the 256 cross tiles of a group pair once; the group pairs follow the circle method (one pair for 512 columns).  No
//           in-block pairs ride along here.
// 512 columns: 5 + 8 = 13 loads and stores of the matrix per sweep instead of 31; XRK = 8 (512 rows) keeps the eight columns of a
// wavefront in 128 registers and the hand-over buffer in 128 KB of LDS - one workgroup per CU, two wavefronts per SIMD.
template <int XRK, bool LATE>
__global__ __launch_bounds__(512, (XRK <= 4) ? 4 : 2) void jacobi_quad64_kernel(JacobiArgs g) {
  extern __shared__ real smem[];
  int b = blockIdx.y;
  if (g.ids) b = __builtin_amdgcn_readfirstlane(g.ids[b]);
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), grp = w >> 2, wl = w & 3;  // (scalar registers)
  if (g.done[b]) return;
  const int rtot = g.rtot;
  typedef ColFrag<XRK> Col;
  real* slots = smem;                                    // [8][4] columns of Col::LDS_REALS
  real* sN = slots + 32 * Col::LDS_REALS;                // [8][4]
  int* sCnt = reinterpret_cast<int*>(sN + 32);           // [8]
  const bool plane = g.mode == 0;
  const int cls = plane ? g.round : 4;                   // parallel class of this launch (0 ... 4); cross mode: the columns 4 w ... 4 w + 3
  const int last = plane ? 2 : 1;                        // last round played
  int blk[4];
  if (plane) {
    const int base = 16 * (blockIdx.x >> 2), line = blockIdx.x & 3;
#pragma unroll
    for (int k = 0; k < 4; ++k) blk[k] = base + ag_point(cls, line, k);
  } else {
    int ga, gb;
    pair_of(g.ngroups + (g.ngroups & 1), g.ground, blockIdx.x >> 3, ga, gb);
    if (gb >= g.ngroups) return;                         // (an odd number of groups: the bye of this round)
    const int i = blockIdx.x & 7, j = (i + g.round) & 7;
    blk[0] = 16 * ga + 2 * i; blk[3] = blk[0] + 1;
    blk[1] = 16 * gb + 2 * j; blk[2] = blk[1] + 1;
  }
  int* st = g.stamps + (long)b * STAMP_STRIDE;
  int mod[4], nz[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    mod[k] = __builtin_amdgcn_readfirstlane(max(st[2 * blk[k]], st[2 * blk[k] + 1]));
    nz[k] = __builtin_amdgcn_readfirstlane(st[2 * MAXBLK + 2 * blk[k]] | st[2 * MAXBLK + 2 * blk[k] + 1]);
  }
  auto ver_of = [&](int i, int j) -> int& { const int I = min(blk[i], blk[j]), J = max(blk[i], blk[j]); return st[3 * MAXBLK + (2 * I) * MAXBLK + 2 * J]; };
  // does the tile of positions (i, j) have anything to do: the rule of the tile kernel
  auto tile_open = [&](int i, int j) { return (nz[i] || nz[j]) && !(__builtin_amdgcn_readfirstlane(ver_of(i, j)) > max(mod[i], mod[j])); };
  {
    // nothing open in the first round and nothing that a rotation of this launch could re-open: leave before the load
    const bool any = tile_open(0, 1) || tile_open(2, 3) || tile_open(0, 2) || tile_open(1, 3) || (plane && (tile_open(0, 3) || tile_open(1, 2)));
    if (!any) return;
  }
  cplx* __restrict__ Yb = g.Y + (long)b * g.y_b0;
  // group 0 holds A as I throughout and B, C, D in turn as J; group 1 holds (C, D), then (B, D), then (B, C); qI / qJ: the lines of the
  // column plane the quads of this wavefront sit on
  int qI = wl, qJ = wl;
  Col yI[4], yJ[4];
  real nI[4], nJ[4];
  {
    const int bI = grp == 0 ? blk[0] : blk[2], bJ = grp == 0 ? blk[1] : blk[3];
    real pn[8];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      yI[h].load(Yb + (long)(bI * 16 + ag_point(cls, qI, h)) * rtot, lane);
      yJ[h].load(Yb + (long)(bJ * 16 + ag_point(cls, qJ, h)) * rtot, lane);
      pn[h] = yI[h].norm2();
      pn[4 + h] = yJ[h].norm2();
    }
    const real z = wave_sum8_groups(pn[0], pn[1], pn[2], pn[3], pn[4], pn[5], pn[6], pn[7], lane);
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      nI[h] = lane_value(z, 8 * h);
      nJ[h] = lane_value(z, 32 + 8 * h);
    }
  }
  const real floor2 = g.floor_scale * g.fro2[b];
  if (g.ablate == 1) {  // (timing only: what the loads and the norms cost)
    if (nI[0] + nJ[0] + nI[3] + nJ[3] == real(-1.0)) st[0] = 1;
    return;
  }
  int cnt = 0;
  auto sub_step = [&](Col& p0, Col& q0, real& a0, real& d0, Col& p1, Col& q1, real& a1, real& d1, Col& p2, Col& q2, real& a2, real& d2, Col& p3, Col& q3,
                      real& a3, real& d3) {
    real gx[4], gy[4];
    Col::dot(p0, q0, gx[0], gy[0]);
    Col::dot(p1, q1, gx[1], gy[1]);
    Col::dot(p2, q2, gx[2], gy[2]);
    Col::dot(p3, q3, gx[3], gy[3]);
    const real gsum = wave_sum8_groups(gx[0], gy[0], gx[1], gy[1], gx[2], gy[2], gx[3], gy[3], lane);
    const int pr = lane >> 4;
    const real a = pr == 0 ? a0 : pr == 1 ? a1 : pr == 2 ? a2 : a3;
    const real d = pr == 0 ? d0 : pr == 1 ? d1 : pr == 2 ? d2 : d3;
    real cv, sv, tv;
    const bool rot = make_rotation_lanes4(a, d, gsum, g.tol2, floor2, cv, sv, tv);
    const unsigned long long any = __ballot(rot) & 0x0001000100010001ull;
    cnt += __popcll(any);
    if (LATE && any == 0) return;
    {
      const real sr = lane_value(sv, 0), si = lane_value(sv, 8), c = lane_value(cv, 0), tg = lane_value(tv, 0);
      Col::rotate(p0, q0, c, sr, si);
      a0 -= tg; d0 += tg;
    }
    {
      const real sr = lane_value(sv, 16), si = lane_value(sv, 24), c = lane_value(cv, 16), tg = lane_value(tv, 16);
      Col::rotate(p1, q1, c, sr, si);
      a1 -= tg; d1 += tg;
    }
    {
      const real sr = lane_value(sv, 32), si = lane_value(sv, 40), c = lane_value(cv, 32), tg = lane_value(tv, 32);
      Col::rotate(p2, q2, c, sr, si);
      a2 -= tg; d2 += tg;
    }
    {
      const real sr = lane_value(sv, 48), si = lane_value(sv, 56), c = lane_value(cv, 48), tg = lane_value(tv, 48);
      Col::rotate(p3, q3, c, sr, si);
      a3 -= tg; d3 += tg;
    }
  };
  int total_all = 0, work = 0;
#pragma unroll 1
  for (int t = 0; t < 3; ++t) {
    // both tiles of the round (every thread keeps the stamps of all four blocks), read BEFORE anybody writes a stamp of this round;
    // constant indices per round: the four stamps stay in scalar registers
    bool open0, open1, cross0, cross1;
    if (t == 0) { open0 = tile_open(0, 1); open1 = tile_open(2, 3); cross0 = nz[0] && nz[1]; cross1 = nz[2] && nz[3]; }       // (A, B) (C, D)
    else if (t == 1) { open0 = tile_open(0, 2); open1 = tile_open(1, 3); cross0 = nz[0] && nz[2]; cross1 = nz[1] && nz[3]; }  // (A, C) (B, D)
    else { open0 = tile_open(0, 3); open1 = tile_open(1, 2); cross0 = nz[0] && nz[3]; cross1 = nz[1] && nz[2]; }              // (A, D) (B, C)
    cross0 = cross0 && open0;
    cross1 = cross1 && open1;
    const bool open = grp == 0 ? open0 : open1;
    const bool cross = grp == 0 ? cross0 : cross1;
    cnt = 0;
    // the in-block pairs of this round: slots (0, 1) and (2, 3) of each block - the slots are rotated between the rounds (below), so
    // that one piece of code plays the three perfect matchings of the four columns
    if (open && plane) sub_step(yI[0], yI[1], nI[0], nI[1], yI[2], yI[3], nI[2], nI[3], yJ[0], yJ[1], nJ[0], nJ[1], yJ[2], yJ[3], nJ[2], nJ[3]);
#pragma unroll 1
    for (int s = 0; s < 4; ++s) {
      if (cross) {
        sub_step(yI[0], yJ[0], nI[0], nJ[0], yI[1], yJ[1], nI[1], nJ[1], yI[2], yJ[2], nI[2], nJ[2], yI[3], yJ[3], nI[3], nJ[3]);
        sub_step(yI[0], yJ[1], nI[0], nJ[1], yI[1], yJ[2], nI[1], nJ[2], yI[2], yJ[3], nI[2], nJ[3], yI[3], yJ[0], nI[3], nJ[0]);
        sub_step(yI[0], yJ[2], nI[0], nJ[2], yI[1], yJ[3], nI[1], nJ[3], yI[2], yJ[0], nI[2], nJ[0], yI[3], yJ[1], nI[3], nJ[1]);
        sub_step(yI[0], yJ[3], nI[0], nJ[3], yI[1], yJ[0], nI[1], nJ[0], yI[2], yJ[1], nI[2], nJ[1], yI[3], yJ[2], nI[3], nJ[2]);
      }
      if (s + 1 < 4) {  // the J quads move one wavefront on inside their group (barriers are for the whole workgroup: both groups
                        // run the same sequence, a closed tile only skips the arithmetic and the traffic)
        if (cross) {
#pragma unroll
          for (int h = 0; h < 4; ++h) yJ[h].to_lds(slots + (w * 4 + h) * Col::LDS_REALS, lane);
          if (lane < 4) sN[w * 4 + lane] = lane == 0 ? nJ[0] : lane == 1 ? nJ[1] : lane == 2 ? nJ[2] : nJ[3];
        }
        __syncthreads();
        if (cross) {
          const int src = grp * 4 + ((wl + 1) & 3);
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            yJ[h].from_lds(slots + (src * 4 + h) * Col::LDS_REALS, lane);
            nJ[h] = sN[src * 4 + h];
          }
        }
        __syncthreads();
      }
    }
    if (cross) qJ = (qJ + 3) & 3;  // after three hand-overs wavefront w holds the J quad that started at wavefront w + 3
    if (lane == 0) sCnt[w] = cnt;
    __syncthreads();
    const int tot0 = __builtin_amdgcn_readfirstlane(sCnt[0] + sCnt[1] + sCnt[2] + sCnt[3]);
    const int tot1 = __builtin_amdgcn_readfirstlane(sCnt[4] + sCnt[5] + sCnt[6] + sCnt[7]);
    const int stamp = g.clock + t;
    const bool moved0 = open0 && tot0 > 0, moved1 = open1 && tot1 > 0;
    const int inb = plane ? 2 * NB : 0;  // rotation slots of the in-block sub-step
    work += (open0 ? (cross0 ? REC_PER_VISIT : 0) + inb : 0) + (open1 ? (cross1 ? REC_PER_VISIT : 0) + inb : 0);  // (scalar, every thread)
    if (moved0) total_all += tot0;
    if (moved1) total_all += tot1;
    if (t == 0) {
      if (tid == 0) {
        if (open0 && tot0 == 0) ver_of(0, 1) = stamp;
        if (open1 && tot1 == 0) ver_of(2, 3) = stamp;
      }
      if (moved0) { mod[0] = stamp; mod[1] = stamp; }
      if (moved1) { mod[2] = stamp; mod[3] = stamp; }
    } else if (t == 1) {
      if (tid == 0) {
        if (open0 && tot0 == 0) ver_of(0, 2) = stamp;
        if (open1 && tot1 == 0) ver_of(1, 3) = stamp;
      }
      if (moved0) { mod[0] = stamp; mod[2] = stamp; }
      if (moved1) { mod[1] = stamp; mod[3] = stamp; }
    } else {
      if (tid == 0) {
        if (open0 && tot0 == 0) ver_of(0, 3) = stamp;
        if (open1 && tot1 == 0) ver_of(1, 2) = stamp;
      }
      if (moved0) { mod[0] = stamp; mod[3] = stamp; }
      if (moved1) { mod[1] = stamp; mod[2] = stamp; }
    }
    if (t == last) break;
    // ---- between the rounds.  (i) Slots 1, 2, 3 of every block rotate: (c0 c1 c2 c3) -> (c0 c2 c3 c1) -> (c0 c3 c1 c2), the three
    // matchings.  (ii) One block of each group changes sides, always as the J block: group 0 gives B, then C; group 1 - whose fixed
    // block would have to change with the round - first trades the roles of its two blocks (registers only: (C, D) -> (D, C) gives
    // C; (D, B) -> (B, D) gives D).  The quads travel as they are: wavefront wl takes what wavefront wl of the other group held.
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      Col* y = q == 0 ? yI : yJ;
      real* nn = q == 0 ? nI : nJ;
      const Col c1 = y[1];
      const real n1 = nn[1];
      y[1] = y[2]; y[2] = y[3]; y[3] = c1;
      nn[1] = nn[2]; nn[2] = nn[3]; nn[3] = n1;
    }
    if (grp == 1) {
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const Col c = yI[h]; yI[h] = yJ[h]; yJ[h] = c;
        const real n = nI[h]; nI[h] = nJ[h]; nJ[h] = n;
      }
      const int q = qI; qI = qJ; qJ = q;
    }
    __syncthreads();  // (the counts above are read; the slots are free)
#pragma unroll
    for (int h = 0; h < 4; ++h) yJ[h].to_lds(slots + (w * 4 + h) * Col::LDS_REALS, lane);
    if (lane < 4) sN[w * 4 + lane] = lane == 0 ? nJ[0] : lane == 1 ? nJ[1] : lane == 2 ? nJ[2] : nJ[3];
    if (lane == 0) sCnt[w] = qJ;  // the line of the column plane the quad sits on travels with it
    __syncthreads();
    {
      const int src = (1 - grp) * 4 + wl;
      qJ = __builtin_amdgcn_readfirstlane(sCnt[src]);
#pragma unroll
      for (int h = 0; h < 4; ++h) { yJ[h].from_lds(slots + (src * 4 + h) * Col::LDS_REALS, lane); nJ[h] = sN[src * 4 + h]; }
    }
    __syncthreads();
  }
  if (tid == 0 && g.work && work) atomicAdd(g.work, work);
  if (total_all == 0) return;
  // blocks that were rotated in this launch go back (mod stamps of this launch), by whoever holds them now: after three rounds group 0
  // ends with (A, D), group 1 with (B, C); after the two rounds of the cross mode group 0 holds (A, C), group 1 (D, B)
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (mod[k] >= g.clock) { st[2 * blk[k]] = mod[k]; st[2 * blk[k] + 1] = mod[k]; }
    atomicAdd(&g.nrot[b], total_all);
  }
  const int kI = plane ? (grp == 0 ? 0 : 1) : (grp == 0 ? 0 : 3), kJ = plane ? (grp == 0 ? 3 : 2) : (grp == 0 ? 2 : 1);
  const int bI = blk[kI], bJ = blk[kJ];
  const int mI = mod[kI], mJ = mod[kJ];
  // after the two rotations slot h holds point (0, 3, 1, 2)[h] of its line; after one (cross mode) point (0, 2, 3, 1)[h]
  if (mI >= g.clock) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int pt = plane ? (h == 0 ? 0 : h == 1 ? 3 : h - 1) : (h == 0 ? 0 : h == 3 ? 1 : h + 1);
      yI[h].store(Yb + (long)(bI * 16 + ag_point(cls, qI, pt)) * rtot, lane);
    }
  }
  if (mJ >= g.clock) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int pt = plane ? (h == 0 ? 0 : h == 1 ? 3 : h - 1) : (h == 0 ? 0 : h == 3 ? 1 : h + 1);
      yJ[h].store(Yb + (long)(bJ * 16 + ag_point(cls, qJ, pt)) * rtot, lane);
    }
  }
}

// Replay of the recorded rotations on the W rows of the same tile: lane = one row, 32 columns in registers.
__global__ __launch_bounds__(64) void jacobi_cross16w_kernel(JacobiArgs g, int wrow0) {
  int b = blockIdx.z;
  if (g.ids) b = g.ids[b];
  const real* __restrict__ rec = g.rec + ((long)b * gridDim.x + blockIdx.x) * (REC_PER_VISIT * 4);
  if (rec[3] == 0.0) return;
  int I, J;
  pair_of(g.nblk / 2, g.round, blockIdx.x, I, J);
  const int rtot = g.rtot;
  const int row = wrow0 + blockIdx.y * 64 + threadIdx.x;
  cplx* __restrict__ Yb = g.Y + (long)b * g.y_b0 + row;
  cplx yI[16], yJ[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    yI[c] = Yb[(long)(I * 16 + c) * rtot];
    yJ[c] = Yb[(long)(J * 16 + c) * rtot];
  }
  // the 256 rotation records (c, sr, si) are spread over the lanes (4 records per lane) and broadcast with v_readlane
  real pc[4], psr[4], psi[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const real* r4 = rec + (long)(threadIdx.x + 64 * q) * 4;
    pc[q] = r4[0];
    psr[q] = r4[1];
    psi[q] = r4[2];
  }
#pragma unroll
  for (int s = 0; s < NB; ++s)
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int w = 0; w < NB; ++w)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int ridx = ((s * 2 + sub) * NB + w) * 2 + h;   // compile-time constant after unrolling
          const real c = lane_value(pc[ridx >> 6], ridx & 63);
          const real sr = lane_value(psr[ridx >> 6], ridx & 63);
          const real si = lane_value(psi[ridx >> 6], ridx & 63);
          // wavefront w rotated its I column 2w+h with the J column that started in wavefront (w+s) mod 8
          rotate_pair(yI[2 * w + h], yJ[2 * ((w + s) & (NB - 1)) + (h ^ sub)], c, sr, si);
        }
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    Yb[(long)(I * 16 + c) * rtot] = yI[c];
    Yb[(long)(J * 16 + c) * rtot] = yJ[c];
  }
}

// ---- medium bonds: the whole Jacobi iteration of one matrix inside one workgroup ---------------------
// At most 64 columns of at most 128 stacked rows (128 KiB of LDS): sixteen wavefronts take the disjoint pairs of every round of the
// circle ordering side by side, all sweeps run inside the launch (norms refreshed from the tile at the start of each sweep), and
// the host reads one convergence flag at the end instead of synchronising after every sweep.  Same rotation rule and noise floor as
// the tiled kernels; zero columns are skipped.
template <int RK>
__global__ __launch_bounds__(1024) void jacobi_lds_kernel(JacobiArgs g, int ncols, int max_sweeps, int* n_unconverged) {
  extern __shared__ real smem[];
  int b = blockIdx.x;
  if (g.ids) b = g.ids[b];
  const int rtot = g.rtot, rx = g.rx;
  cplx* tile = reinterpret_cast<cplx*>(smem);                     // [ncols][rtot]
  real* sN = reinterpret_cast<real*>(tile + (long)ncols * rtot);  // [ncols]
  int* sCnt = reinterpret_cast<int*>(sN + ncols);                 // [16]
  cplx* __restrict__ Yb = g.Y + (long)b * g.y_b0;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int c = w; c < ncols; c += 16)
#pragma unroll
    for (int k = 0; k < RK; ++k)
      if (lane + 64 * k < rtot) tile[c * rtot + lane + 64 * k] = Yb[(long)c * rtot + lane + 64 * k];
  __syncthreads();
  bool converged = false;
  real floor2 = 0.0;
  for (int sweep = 0; sweep < max_sweeps && !converged; ++sweep) {
    for (int c = w; c < ncols; c += 16) {  // column norms of the X part from the tile (no drift from the running updates)
      real n = 0.0;
#pragma unroll
      for (int k = 0; k < RK; ++k)
        if (lane + 64 * k < rx) {
          const cplx v = tile[c * rtot + lane + 64 * k];
          n = fma(v.x, v.x, fma(v.y, v.y, n));
        }
      n = wave_sum(n);
      if (lane == 0) sN[c] = n;
    }
    __syncthreads();
    if (sweep == 0) {
      real f = 0.0;
      for (int c = 0; c < ncols; ++c) f += sN[c];
      floor2 = g.floor_scale * f;
      // every wavefront has to have read the norms before the first rotation updates two of them: a wavefront that was still
      // summing saw a - t|g| next to the old d and got a different noise floor, and a column AT the floor (bonds with numerically
      // zero Schmidt values) was then rotated in one run and not in the next - last-bit differences between identical runs, one
      // run in ten (round 4, tests/probes/determinism_*_probe.py)
      __syncthreads();
    }
    int cnt = 0;
    for (int s = 0; s < ncols - 1; ++s) {
      for (int pi = w; pi < ncols / 2; pi += 16) {
        int p, q;
        pair_of(ncols, s, pi, p, q);
        const real a = sN[p], dd = sN[q];
        if (a == 0.0 || dd == 0.0) continue;  // padding columns
        cplx yp[RK], yq[RK];
        real gx = 0.0, gy = 0.0;
#pragma unroll
        for (int k = 0; k < RK; ++k) {
          if (lane + 64 * k < rtot) {
            yp[k] = tile[p * rtot + lane + 64 * k];
            yq[k] = tile[q * rtot + lane + 64 * k];
            if (lane + 64 * k < rx) {
              gx = fma(yp[k].x, yq[k].x, fma(yp[k].y, yq[k].y, gx));
              gy = fma(yp[k].x, yq[k].y, fma(-yp[k].y, yq[k].x, gy));
            }
          }
        }
        gx = wave_sum(gx);
        gy = wave_sum(gy);
        real c, sr, si, tg;
        if (make_rotation(a, dd, gx, gy, g.tol2, floor2, c, sr, si, tg)) {
#pragma unroll
          for (int k = 0; k < RK; ++k) {
            if (lane + 64 * k < rtot) {
              rotate_pair(yp[k], yq[k], c, sr, si);
              tile[p * rtot + lane + 64 * k] = yp[k];
              tile[q * rtot + lane + 64 * k] = yq[k];
            }
          }
          if (lane == 0) { sN[p] = a - tg; sN[q] = dd + tg; }
          ++cnt;
        }
      }
      __syncthreads();
    }
    if (lane == 0) sCnt[w] = cnt;
    __syncthreads();
    int total = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) total += sCnt[q];
    converged = total == 0;
    __syncthreads();
  }
  for (int c = w; c < ncols; c += 16)
#pragma unroll
    for (int k = 0; k < RK; ++k)
      if (lane + 64 * k < rtot) Yb[(long)c * rtot + lane + 64 * k] = tile[c * rtot + lane + 64 * k];
  if (!converged && tid == 0) atomicAdd(n_unconverged, 1);
}

// ---- pairs inside one block (LDS resident) -----------------------------------------------------
template <int RK>
__global__ __launch_bounds__(256) void jacobi_diag_kernel(JacobiArgs g) {
  extern __shared__ real smem[];
  int b = blockIdx.y;
  if (g.ids) b = g.ids[b];
  if (g.done[b]) return;
  const int rtot = g.rtot, rx = g.rx;
  const int nrk = rtot >> 6;
  cplx* tile = reinterpret_cast<cplx*>(smem);                  // [8][rtot]
  real* sN = reinterpret_cast<real*>(tile + NB * rtot);    // [8]
  int* sCnt = reinterpret_cast<int*>(sN + NB);                 // [4]
  int* st = g.stamps + (long)b * STAMP_STRIDE;
  const int I = blockIdx.x;
  if (!st[2 * MAXBLK + I]) return;
  if (st[MAXBLK + I] > st[I]) return;
  cplx* __restrict__ Yb = g.Y + (long)b * g.y_b0 + (long)blockIdx.x * NB * rtot;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int c = w; c < NB; c += 4) {
    real n = 0.0;
    for (int k = 0; k < nrk; ++k) {
      const cplx v = Yb[(long)c * rtot + lane + 64 * k];
      tile[c * rtot + lane + 64 * k] = v;
      if (lane + 64 * k < rx) n = fma(v.x, v.x, fma(v.y, v.y, n));
    }
    n = wave_sum(n);
    if (lane == 0) sN[c] = n;
  }
  __syncthreads();
  const real floor2 = g.floor_scale * g.fro2[b];
  int cnt = 0;
  for (int s = 0; s < NB - 1; ++s) {
    int p, q;
    pair_of(NB, s, w, p, q);
    cplx yp[RK], yq[RK];
    real gx = 0.0, gy = 0.0;
#pragma unroll
    for (int k = 0; k < RK; ++k) {
      if (k < nrk) {
        yp[k] = tile[p * rtot + lane + 64 * k];
        yq[k] = tile[q * rtot + lane + 64 * k];
        if (lane + 64 * k < rx) {
          gx = fma(yp[k].x, yq[k].x, fma(yp[k].y, yq[k].y, gx));
          gy = fma(yp[k].x, yq[k].y, fma(-yp[k].y, yq[k].x, gy));
        }
      }
    }
    gx = wave_sum(gx);
    gy = wave_sum(gy);
    real c, sr, si, tg;
    const real a = sN[p], d = sN[q];
    if (make_rotation(a, d, gx, gy, g.tol2, floor2, c, sr, si, tg)) {
#pragma unroll
      for (int k = 0; k < RK; ++k) {
        if (k < nrk) {
          rotate_pair(yp[k], yq[k], c, sr, si);
          tile[p * rtot + lane + 64 * k] = yp[k];
          tile[q * rtot + lane + 64 * k] = yq[k];
        }
      }
      if (lane == 0) { sN[p] = a - tg; sN[q] = d + tg; }
      ++cnt;
    }
    __syncthreads();
  }
  if (lane == 0) sCnt[w] = cnt;
  __syncthreads();
  const int total = sCnt[0] + sCnt[1] + sCnt[2] + sCnt[3];
  if (tid == 0 && g.work) atomicAdd(g.work, NB * (NB - 1) / 2);
  if (total == 0) {
    if (tid == 0) st[MAXBLK + I] = g.clock;
    return;
  }
  if (tid == 0) st[I] = g.clock;
  for (int c = w; c < NB; c += 4)
    for (int k = 0; k < nrk; ++k) Yb[(long)c * rtot + lane + 64 * k] = tile[c * rtot + lane + 64 * k];
  if (tid == 0) atomicAdd(&g.nrot[b], total);
}

// Y[c][r]:  rows [0, rx) = X (from the strided source, optionally conjugated), rows [rx, rx + ncols_pad) = identity
__global__ __launch_bounds__(256) void jacobi_load_kernel(JacobiSource src, cplx* __restrict__ Y, long y_b0, int ncols_pad, int rx_top,
                                                         int rtot) {
  int b = blockIdx.y;
  if (src.ids) b = src.ids[b];
  const cplx* sp = src.src + (long)b * src.src_b0;
  cplx* Yb = Y + (long)b * y_b0;
  const long total = (long)ncols_pad * rtot;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e / rtot), r = (int)(e % rtot);
    cplx v{0.0, 0.0};
    if (r < rx_top) {
      if (r < src.rx && c < src.ncols && !(src.tri && c > r)) {
        // two-level row / column indices let a (phys, bond) pair be flattened without a transpose
        const int r1 = r / src.r_n0, r0 = r % src.r_n0;
        const int c1 = c / src.c_n0, c0 = c % src.c_n0;
        v = sp[(long)r1 * src.s_r1 + (long)r0 * src.s_r0 + (long)c1 * src.s_c1 + (long)c0 * src.s_c0];
        if (src.conj) v.y = -v.y;
      }
    } else if (r - rx_top == c) {
      v.x = 1.0;
    }
    Yb[e] = v;
  }
}

// squared Frobenius norm of the source (noise floor of the rotations)
__global__ __launch_bounds__(256) void jacobi_fro_kernel(const cplx* __restrict__ Y, long y_b0, int ncols_pad, int rx_top, int rtot,
                                                        real* fro2, const int* ids) {
  __shared__ real sh[4];
  int b = blockIdx.x;
  if (ids) b = ids[b];
  const cplx* Yb = Y + (long)b * y_b0;
  real acc = 0.0;
  const long total = (long)ncols_pad * rx_top;
  for (long e = threadIdx.x; e < total; e += blockDim.x) {
    const long c = e / rx_top, r = e % rx_top;
    const cplx v = Yb[c * rtot + r];
    acc = fma(v.x, v.x, fma(v.y, v.y, acc));
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) fro2[b] = sh[0] + sh[1] + sh[2] + sh[3];
}

// stamps: mod = 1, verified = 0, nz[I] = block I has a non-zero X column
__global__ __launch_bounds__(256) void jacobi_stamp_init_kernel(const cplx* __restrict__ Y, long y_b0, int nblk, int rx_top, int rtot, int* stamps,
                                                               const int* ids) {
  int b = blockIdx.x;
  if (ids) b = ids[b];
  int* st = stamps + (long)b * STAMP_STRIDE;
  for (int t = threadIdx.x; t < STAMP_STRIDE; t += blockDim.x) st[t] = (t < MAXBLK) ? 1 : 0;
  __syncthreads();
  const cplx* Yb = Y + (long)b * y_b0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int I = wave; I < nblk; I += 4) {
    int nz = 0;
    for (int c = 0; c < NB; ++c)
      for (int r = lane; r < rx_top; r += 64) {
        const cplx v = Yb[(long)(I * NB + c) * rtot + r];
        nz |= (v.x != 0.0 || v.y != 0.0) ? 1 : 0;
      }
    nz = __any(nz);
    if (lane == 0) st[2 * MAXBLK + I] = nz;
  }
}

// stop_at: a trajectory is done when its sweep applied at most this many rotations (0: a sweep without rotations, the convergence
// criterion; > 0: callers that refine the result anyway).  The decision is per trajectory, so what a trajectory gets does not depend
// on the others in its batch.
// n_active: the counter slot of THIS sweep (live trajectories, rotations, -, -, rotation slots executed: the tile kernels add to [4]);
// clear: the slot of the next sweep, zeroed here - the sweeps alternate between two slots, so no fill command sits between them
__global__ void svd_sweep_check_kernel(int* nrot, int* done, int* n_active, int nb0, const int* ids, int stop_at, int* clear) {
  if (clear != nullptr && blockIdx.x == 0 && threadIdx.x < 5) clear[threadIdx.x] = 0;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nb0) return;
  const int b = ids ? ids[t] : t;
  if (!done[b]) {
    if (nrot[b] <= stop_at) done[b] = 1;
    else atomicAdd(n_active, 1);
    atomicAdd(n_active + 1, nrot[b]);
  }
  nrot[b] = 0;
}

__global__ void svd_reset_kernel(int* nrot, int* done, int nb0, const int* ids, int* n_active) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < 16) n_active[16 + t] = 0;  // the two counter slots of the sweeps (svd_sweep_check_kernel)
  if (t == 16) n_active[2] = 0;      // "a kept singular value sits at the noise floor" (svd_finish_kernel)
  if (t >= nb0) return;
  const int b = ids ? ids[t] : t;
  nrot[b] = 0;
  done[b] = 0;
}

constexpr int SMALL_MAXN = 16;

// ---- small bonds: the whole centre shift in one kernel -------------------------------------------
// One wavefront per trajectory, lane = row of the matrix being orthogonalised (at most 64 rows, 16 columns), columns in LDS.
// Right shift: X = A_i as (d ca) x cb;  left shift: X = A_i^H as (d cb) x ca.  One-sided Jacobi over the ACTUAL columns
// (the bond dimension of this trajectory, not the padded capacity), cyclic by rows, the rotation rule of the large kernels;
// then norms, rank sort and the discarded-weight truncation of svd_finish_kernel; the isometric factor is the set of
// normalised columns, the weighted factor is its overlap with the input (accumulation-free, as in the large path) and goes
// straight into the neighbouring tensor.  Replaces ~12 launches and one host synchronisation per sweep.
// Cyclic-by-rows one-sided Jacobi on the n columns Y[j * pitch + 0 .. pitch) held in LDS by ONE wavefront (lane = row), with the rotation rule of
// the large kernels.  Returns false when 40 sweeps did not converge.
__device__ inline bool small_jacobi(cplx* Y, int pitch, int n, int lane, real floor2) {
  bool converged = n < 2;
  const bool mine = lane < pitch;  // rows beyond the pitch do not exist (and would be zero)
  for (int sweep = 0; sweep < 40 && !converged; ++sweep) {
    int cnt = 0;
    for (int pc = 0; pc + 1 < n; ++pc)
      for (int qc = pc + 1; qc < n; ++qc) {
        cplx yp = mine ? Y[pc * pitch + lane] : cplx{0.0, 0.0}, yq = mine ? Y[qc * pitch + lane] : cplx{0.0, 0.0};
        // the four sums of the pair (two norms, the inner product) in ONE packed butterfly; lanes 0..3 hold the totals
        const real packed = wave_sum4(fma(yp.x, yp.x, yp.y * yp.y), fma(yq.x, yq.x, yq.y * yq.y), fma(yp.x, yq.x, yp.y * yq.y),
                                        fma(yp.x, yq.y, -yp.y * yq.x), lane);
        const real a = lane_value(packed, 0), dd = lane_value(packed, 1), gx = lane_value(packed, 2), gy = lane_value(packed, 3);
        real c, sr, si, tg;
        if (make_rotation(a, dd, gx, gy, TJM_JACOBI_TOL2, floor2, c, sr, si, tg)) {
          rotate_pair(yp, yq, c, sr, si);
          if (mine) {
            Y[pc * pitch + lane] = yp;
            Y[qc * pitch + lane] = yq;
          }
          ++cnt;
        }
        __syncthreads();
      }
    converged = cnt == 0;
  }
  return converged;
}

// Number of singular values to keep (svd_utils.py:22-104); sv(k) = k-th largest value, nsv = how many exist.
template <class SV>
__device__ inline int truncation_keep(const TruncSpec& d, int nsv, SV sv) {
  int keep = 0;
  if (nsv <= 0) return 0;
  if (d.trunc_mode == 2) {  // hard_cutoff
    for (int k = 0; k < nsv; ++k) keep += (sv(k) > d.threshold) ? 1 : 0;
  } else if (d.trunc_mode == 1) {  // relative
    const real smax = sv(0);
    if (smax > 0.0)
      for (int k = 0; k < nsv; ++k) keep += ((sv(k) / smax) >= d.threshold) ? 1 : 0;
  } else if (d.trunc_mode == 0) {  // discarded_weight
    keep = nsv;
    real discard = 0.0;
    for (int idx = 0; idx < nsv; ++idx) {
      const real s = sv(nsv - 1 - idx);
      discard += s * s;
      if (discard >= d.threshold) {
        keep = nsv - idx;
        if (keep < d.min_keep) keep = d.min_keep;
        break;
      }
    }
  } else {  // relative_discarded_weight
    const real smax = sv(0);
    if (smax > 0.0) {
      real total = 0.0;
      for (int k = 0; k < nsv; ++k) { const real q = sv(k) / smax; total += q * q; }
      keep = nsv;
      real discard = 0.0;
      for (int idx = 0; idx < nsv; ++idx) {
        const real q = sv(nsv - 1 - idx) / smax;
        const real cand = discard + q * q;
        if (cand / total <= d.threshold) { discard = cand; keep = nsv - idx - 1; }
        else break;
      }
    }
  }
  if (d.max_bond > 0 && keep > d.max_bond) keep = d.max_bond;
#ifdef TJM_F32
  // complex64 build: singular values below the resolution of the arithmetic (TJM_RANK_TOL of the largest one) are rounding noise -
  // their columns were never rotated and are not orthogonal to the rest (found on the MI355X: the 1e-12 discarded-weight rule of
  // the centre shifts kept values at 1e-6 of the largest, the "isometric" factor was 0.3 away from an isometry).  They carry
  // < 1e-10 of the weight each and are dropped with the discarded ones.
  while (keep > 1 && keep > d.min_keep && sv(keep - 1) <= TJM_RANK_TOL * sv(0)) --keep;
#endif
  if (keep < d.min_keep) keep = d.min_keep;
  if (keep > nsv) keep = nsv;
  if (d.cap > 0 && keep > d.cap) {  // the engine's storage is smaller than what the truncation rule asks for
    keep = d.cap;
    if (d.overflow) atomicOr(d.overflow, 1);
  }
  return keep;
}

// Neighbour update shared by the fused small-bond kernels.  G[k][j] (k < keep new, j < n old) is the weighted factor.
//   right shift (LEFT = false):  N[t][k][c] = sum_j G[k][j] N[t][j][c]   neighbour A_{i+1} [d][cb][cn], bond = its rows
//   left shift  (LEFT = true):   N[s][z][k] = sum_j N[s][z][j] G[k][j]   neighbour A_{i-1} [d][cn][ca], bond = its columns
template <bool LEFT>
__device__ inline void small_absorb(cplx* __restrict__ Nb, const cplx (*G)[SMALL_MAXN], int d, int ca, int cb, int cn, int n, int keep, int ncap,
                                    int lane) {
  const int lines = d * cn;  // independent lines of the neighbour, each of ncap entries along the bond
  for (int line = lane; line < lines; line += 64) {
    cplx old[SMALL_MAXN];
    long base, stride;
    if (LEFT) { base = (long)line * ca; stride = 1; }                                                // line = (s, z)
    else { const int t = line / cn, c = line - t * cn; base = (long)t * cb * cn + c; stride = cn; }  // line = (t, c)
#pragma unroll
    for (int j = 0; j < SMALL_MAXN; ++j) old[j] = (j < n) ? Nb[base + j * stride] : cplx{0.0, 0.0};
    for (int k = 0; k < ncap; ++k) {
      real ax = 0.0, ay = 0.0;
      if (k < keep) {
#pragma unroll
        for (int j = 0; j < SMALL_MAXN; ++j) {
          if (j < n) {
            const cplx g = G[k][j];
            ax = fma(g.x, old[j].x, fma(-g.y, old[j].y, ax));
            ay = fma(g.x, old[j].y, fma(g.y, old[j].x, ay));
          }
        }
      }
      Nb[base + k * stride] = cplx{ax, ay};
    }
  }
}

// LDS of one wavefront of the fused small-bond kernels
struct SmallLds {
  cplx* Y;     // [SMALL_MAXN][pitch] in dynamic LDS; pitch = 16, 32 or 64 rows (the smaller, the more wavefronts per CU)
  int pitch;
  cplx G[SMALL_MAXN][SMALL_MAXN];
  cplx diag[SMALL_MAXN];
  real norm[SMALL_MAXN];
  real beta[SMALL_MAXN];
  int perm[SMALL_MAXN];
  int keep;
};

template <bool LEFT>
__device__ inline void svd_shift_small_body(const SmallShiftDesc& p, int b, int lane, SmallLds& sm) {
  cplx* Y = sm.Y;
  const int pitch = sm.pitch;
  cplx (*G)[SMALL_MAXN] = sm.G;
  real* sNorm = sm.norm;
  int* sPerm = sm.perm;
  int& sKeep = sm.keep;
  const int d = p.d, ca = p.ca, cb = p.cb;
  cplx* __restrict__ A = p.site + (long)b * p.site_b0;
  int* chi = p.chi + (long)b * p.chi_stride;
  const int chiL = chi[0], chiR = chi[1];
  const int R = LEFT ? d * cb : d * ca;       // rows of X (padded extents: rows beyond the actual bond are zero)
  const int n = LEFT ? chiL : chiR;           // actual columns
  const int ncap = LEFT ? ca : cb;            // padded columns
  int nsv = LEFT ? min(chiL, d * chiR) : min(d * chiL, chiR);
  if (nsv > n) nsv = n;
  // element (row r, column j) of X inside the site tensor
  auto site_index = [&](int r, int j) -> long {
    if (LEFT) { const int t = r / cb, c = r - t * cb; return ((long)t * ca + j) * cb + c; }
    return (long)r * cb + j;
  };
  real fro = 0.0;
  for (int j = 0; j < n; ++j) {
    cplx v{0.0, 0.0};
    if (lane < R) {
      v = A[site_index(lane, j)];
      if (LEFT) v.y = -v.y;
    }
    if (lane < pitch) Y[j * pitch + lane] = v;
    fro = fma(v.x, v.x, fma(v.y, v.y, fro));
  }
  fro = wave_sum(fro);
  const real floor2 = TJM_NOISE_FLOOR2 * fro;
  __syncthreads();
  const bool converged = small_jacobi(Y, pitch, n, lane, floor2);
  if (!converged && lane == 0 && p.flags) atomicOr(p.flags + 1, 1);
  // norms, descending rank sort (ties by index), truncation
  for (int j = 0; j < n; ++j) {
    const cplx v = (lane < pitch) ? Y[j * pitch + lane] : cplx{0.0, 0.0};
    const real s2 = wave_sum(fma(v.x, v.x, v.y * v.y));
    if (lane == 0) sNorm[j] = s2;
  }
  __syncthreads();
  if (lane < n) {
    const real v = tjm_sort_key(sNorm[lane]);
    int rank = 0;
    for (int o = 0; o < n; ++o) {
      const real u = tjm_sort_key(sNorm[o]);
      rank += (u > v || (u == v && o < lane)) ? 1 : 0;
    }
    sPerm[rank] = lane;
  }
  __syncthreads();
  if (lane == 0) {
    int keep = 0;
    if (nsv > 0) {
      keep = nsv;
      real discard = 0.0;
      for (int idx = 0; idx < nsv; ++idx) {
        const real s = sqrt(sNorm[sPerm[nsv - 1 - idx]]);
        discard += s * s;
        if (discard >= p.threshold) {
          keep = nsv - idx;
          if (keep < p.min_keep) keep = p.min_keep;
          break;
        }
      }
#ifdef TJM_F32
      while (keep > 1 && keep > p.min_keep && sqrt(sNorm[sPerm[keep - 1]]) <= TJM_RANK_TOL * sqrt(sNorm[sPerm[0]])) --keep;  // see truncation_keep
#endif
      if (keep < p.min_keep) keep = p.min_keep;
      if (keep > nsv) keep = nsv;
    }
    sKeep = keep;
    chi[LEFT ? 0 : 1] = keep;
  }
  __syncthreads();
  const int keep = sKeep;
  // weighted factor from the untouched input:  right: G[k][j] = sum_r conj(U[r][k]) X0[r][j]   (= S V^H)
  //                                            left:  G[j][k] = sum_r conj(X0[r][j]) V[r][k]   (= U S), stored as G[k][j] too
  for (int e = lane; e < keep * n; e += 64) {
    const int k = e / n, j = e - k * n;
    const int col = sPerm[k];
    const real inv = 1.0 / sqrt(sNorm[col]);
    real ax = 0.0, ay = 0.0;
    for (int r = 0; r < R; ++r) {
      const cplx u = Y[col * pitch + r];
      cplx x = A[site_index(r, j)];
      if (LEFT) x.y = -x.y;
      if (LEFT) {  // conj(x) * u
        ax = fma(x.x, u.x, fma(x.y, u.y, ax));
        ay = fma(x.x, u.y, fma(-x.y, u.x, ay));
      } else {     // conj(u) * x
        ax = fma(u.x, x.x, fma(u.y, x.y, ax));
        ay = fma(u.x, x.y, fma(-u.y, x.x, ay));
      }
    }
    G[k][j] = cplx{ax * inv, ay * inv};
  }
  __syncthreads();
  // isometric factor into the site tensor (zero beyond keep, all padded columns written)
  if (lane < R) {
    for (int k = 0; k < ncap; ++k) {
      cplx v{0.0, 0.0};
      if (k < keep) {
        const int col = sPerm[k];
        const real inv = 1.0 / sqrt(sNorm[col]);
        v = Y[col * pitch + lane];
        v.x *= inv;
        v.y *= LEFT ? -inv : inv;
      }
      A[site_index(lane, k)] = v;
    }
  }
  small_absorb<LEFT>(p.nb + (long)b * p.nb_b0, G, d, ca, cb, p.cn, n, keep, ncap, lane);
}

extern __shared__ real small_dyn_lds[];

template <bool LEFT>
__global__ __launch_bounds__(64) void svd_shift_small_kernel(SmallShiftDesc p, int pitch) {
  __shared__ SmallLds sm;
  if (threadIdx.x == 0) { sm.Y = reinterpret_cast<cplx*>(small_dyn_lds); sm.pitch = pitch; }
  __syncthreads();
  int b = blockIdx.x;
  if (p.ids) b = p.ids[b];
  svd_shift_small_body<LEFT>(p, b, threadIdx.x, sm);
}

// ---- small bonds: the two-site split in one kernel ------------------------------------------------
// theta (d capL x d capR, at most 64 rows on the isometric side and MAXN actual columns on the other) -> left, right with the
// truncation rule of svd_finish_kernel.  distribution 0: X = theta, U = normalised rotated columns, right = U^H theta;
// distribution 1: X = theta^H, V likewise, left = theta V.  Columns are the ACTUAL ones (t, c < chi) of this trajectory.
// A kept singular value at the rounding floor (<= 1e-11 sigma_0: its column was never rotated) raises flags[2]; the caller then
// repeats the batch on the general path, which completes such columns to an orthonormal set.
template <int MAXN>
__global__ __launch_bounds__(64) void svd_split_small_kernel(SvdSplitDesc p, TruncSpec tr, int* flags, int pitch) {
  cplx* Y = reinterpret_cast<cplx*>(small_dyn_lds);  // [MAXN][pitch]
  __shared__ cplx G[MAXN][MAXN];
  __shared__ real sNorm[MAXN];
  __shared__ int sPerm[MAXN];
  __shared__ int sKeep;
  int b = blockIdx.x;
  if (p.ids) b = p.ids[b];
  const int lane = threadIdx.x;
  const int d = p.d, capL = p.capL, capR = p.capR, capM = p.capM;
  const bool d0 = p.distribution == 0;
  const cplx* __restrict__ T = p.theta + (long)b * p.theta_b0;
  const int chiL = p.chiL[(long)b * p.chi_stride], chiR = p.chiR[(long)b * p.chi_stride];
  const int R = d0 ? d * capL : d * capR;            // rows of X (padded extent)
  const int chiC = d0 ? chiR : chiL;                 // actual bond on the column side
  const int capC = d0 ? capR : capL;
  const int n = d * chiC;                            // actual columns
  const int nsv = min(d * chiL, d * chiR);
  auto padded_col = [&](int j) { const int t = j / chiC; return t * capC + (j - t * chiC); };
  auto theta_at = [&](int r, int j) -> cplx {        // X[r][j]
    if (d0) return T[(long)r * p.ld_theta + padded_col(j)];
    cplx v = T[(long)padded_col(j) * p.ld_theta + r];
    v.y = -v.y;
    return v;
  };
  real fro = 0.0;
  for (int j = 0; j < n; ++j) {
    const cplx v = (lane < R) ? theta_at(lane, j) : cplx{0.0, 0.0};
    if (lane < pitch) Y[j * pitch + lane] = v;
    fro = fma(v.x, v.x, fma(v.y, v.y, fro));
  }
  fro = wave_sum(fro);
  __syncthreads();
  const bool converged = small_jacobi(Y, pitch, n, lane, TJM_NOISE_FLOOR2 * fro);
  if (!converged && lane == 0) atomicOr(flags + 3, 1);
  for (int j = 0; j < n; ++j) {
    const cplx v = (lane < pitch) ? Y[j * pitch + lane] : cplx{0.0, 0.0};
    const real s2 = wave_sum(fma(v.x, v.x, v.y * v.y));
    if (lane == 0) sNorm[j] = s2;
  }
  __syncthreads();
  if (lane < n) {
    const real v = tjm_sort_key(sNorm[lane]);
    int rank = 0;
    for (int o = 0; o < n; ++o) {
      const real u = tjm_sort_key(sNorm[o]);
      rank += (u > v || (u == v && o < lane)) ? 1 : 0;
    }
    sPerm[rank] = lane;
  }
  __syncthreads();
  if (lane == 0) {
    const int keep = truncation_keep(tr, min(nsv, n), [&](int k) { return sqrt(sNorm[sPerm[k]]); });
    sKeep = keep;
    tr.chiOut[(long)b * tr.chi_stride] = keep;
    if (keep > 0 && sqrt(sNorm[sPerm[keep - 1]]) <= TJM_RANK_TOL * sqrt(sNorm[sPerm[0]])) atomicOr(flags + 2, 1);
  }
  if (tr.spectrum)
    for (int k = lane; k < tr.spec_ld; k += 64) tr.spectrum[(long)b * tr.spec_ld + k] = (k < n) ? sqrt(sNorm[sPerm[k]]) : 0.0;
  __syncthreads();
  const int keep = sKeep;
  // weighted factor G[k][j] = sum_r conj(Q[r][k]) X0[r][j]  (d0: S V^H)   or   sum_r conj(X0[r][j]) Q[r][k]  (d1: U S), Q = Y / sigma
  for (int e = lane; e < keep * n; e += 64) {
    const int k = e / n, j = e - k * n;
    const int col = sPerm[k];
    const real inv = 1.0 / sqrt(sNorm[col]);
    real ax = 0.0, ay = 0.0;
    for (int r = 0; r < R; ++r) {
      const cplx u = Y[col * pitch + r];
      const cplx x = theta_at(r, j);
      if (d0) {  // conj(u) * x
        ax = fma(u.x, x.x, fma(u.y, x.y, ax));
        ay = fma(u.x, x.y, fma(-u.y, x.x, ay));
      } else {   // conj(x) * u
        ax = fma(x.x, u.x, fma(x.y, u.y, ax));
        ay = fma(x.x, u.y, fma(-x.y, u.x, ay));
      }
    }
    G[k][j] = cplx{ax * inv, ay * inv};
  }
  __syncthreads();
  cplx* __restrict__ Lt = p.left + (long)b * p.left_b0;     // [d][capL][capM]
  cplx* __restrict__ Rt = p.right + (long)b * p.right_b0;   // [d][capM][capR]
  // isometric side: one row of X per lane
  if (lane < R) {
    for (int k = 0; k < capM; ++k) {
      cplx v{0.0, 0.0};
      if (k < keep) {
        const int col = sPerm[k];
        const real inv = 1.0 / sqrt(sNorm[col]);
        v = Y[col * pitch + lane];
        v.x *= inv;
        v.y *= d0 ? inv : -inv;
      }
      if (d0) Lt[(long)lane * capM + k] = v;                                        // left[(s,a)][k] = U
      else { const int t = lane / capR, c = lane - t * capR; Rt[((long)t * capM + k) * capR + c] = v; }  // right[t][k][c] = conj(V)
    }
  }
  // weighted side: every padded entry, zero outside the actual block
  const long total = (long)d * capC * capM;
  for (long e = lane; e < total; e += 64) {
    cplx v{0.0, 0.0};
    if (d0) {  // right[t][k][c]
      const int c = (int)(e % capR);
      const long q = e / capR;
      const int k = (int)(q % capM), t = (int)(q / capM);
      if (k < keep && c < chiR) v = G[k][t * chiR + c];
      Rt[e] = v;
    } else {   // left[s][a][k]
      const int k = (int)(e % capM);
      const long q = e / capM;
      const int a = (int)(q % capL), s_ = (int)(q / capL);
      if (k < keep && a < chiL) v = G[k][s_ * chiL + a];
      Lt[e] = v;
    }
  }
}

// ---- small bonds: Householder QR of one site in one kernel ---------------------------------------
// qr_site for d*cap <= 64 rows and cap <= 16 columns: rows in bond-major order (bond, p) so that the actual rows are the leading
// lanes and padded rows stay exactly zero; reflectors v_k = x - alpha e_k (alpha = -e^{i arg x_k} |x|, H_k Hermitian) kept in
// place below the diagonal; R into the bond matrix Cm (and, for the shifts, straight into the neighbour); Q = H_0 ... H_{k-1}
// applied to unit vectors.  Thin-QR bond rule k = min(rows, columns) of np.linalg.qr as in qr_bond_dims_kernel.
template <bool RIGHT>
__device__ inline void qr_site_small_body(const SmallQrDesc& p, int b, int lane, SmallLds& sm) {
  cplx* Z = sm.Y;
  const int pitch = sm.pitch;
  cplx (*G)[SMALL_MAXN] = sm.G;
  cplx* sDiag = sm.diag;
  real* sBeta = sm.beta;
  const int d = p.d, ca = p.ca, cb = p.cb;
  cplx* __restrict__ A = p.site + (long)b * p.site_b0;
  int* chi = p.chi + (long)b * p.chi_stride;
  const int chiL = chi[0], chiR = chi[1];
  const int R = RIGHT ? d * ca : d * cb;
  const int rows_act = RIGHT ? d * chiL : d * chiR;
  const int n = RIGHT ? chiR : chiL;
  const int ncap = RIGHT ? cb : ca;
  const int kn = min(rows_act, n);
  const int bond = lane / d, ph = lane - bond * d;
  auto site_index = [&](int col) -> long { return RIGHT ? ((long)ph * ca + bond) * cb + col : ((long)ph * ca + col) * cb + bond; };
  for (int j = 0; j < n; ++j)
    if (lane < pitch) Z[j * pitch + lane] = (lane < rows_act) ? A[site_index(j)] : cplx{0.0, 0.0};
  for (int e = lane; e < SMALL_MAXN * SMALL_MAXN; e += 64) G[e / SMALL_MAXN][e % SMALL_MAXN] = cplx{0.0, 0.0};
  __syncthreads();
  for (int k = 0; k < kn; ++k) {
    const bool in = lane >= k && lane < rows_act;
    cplx x = in ? Z[k * pitch + lane] : cplx{0.0, 0.0};
    const real nx2 = wave_sum(fma(x.x, x.x, x.y * x.y));
    if (nx2 == 0.0) {  // nothing below the diagonal and a zero pivot: H_k = 1
      if (lane == 0) { sDiag[k] = cplx{0.0, 0.0}; sBeta[k] = 0.0; }
      __syncthreads();
      continue;
    }
    const cplx xk = Z[k * pitch + k];
    const real nx = sqrt(nx2), ak = sqrt(fma(xk.x, xk.x, xk.y * xk.y));
    const real px = ak > 0.0 ? xk.x / ak : 1.0, py = ak > 0.0 ? xk.y / ak : 0.0;
    const cplx alpha{-px * nx, -py * nx};
    const real beta = 1.0 / (nx * (nx + ak));  // 2 / |v|^2
    if (lane == k) { x.x -= alpha.x; x.y -= alpha.y; }
    __syncthreads();  // every lane has read Z[k][k]
    if (in) Z[k * pitch + lane] = x;
    if (lane == 0) { sDiag[k] = alpha; sBeta[k] = beta; }
    for (int j = k + 1; j < n; ++j) {
      cplx y = in ? Z[j * pitch + lane] : cplx{0.0, 0.0};
      const real wr = beta * wave_sum(fma(x.x, y.x, x.y * y.y));   // beta * conj(v) . y
      const real wi = beta * wave_sum(fma(x.x, y.y, -x.y * y.x));
      if (in) {
        y.x -= wr * x.x - wi * x.y;
        y.y -= wr * x.y + wi * x.x;
        Z[j * pitch + lane] = y;
      }
    }
    __syncthreads();
  }
  // R (kn x n, upper trapezoidal): G[k][j], and the padded bond matrix
  for (int e = lane; e < kn * n; e += 64) {
    const int k = e / n, j = e - k * n;
    if (j >= k) G[k][j] = (j == k) ? sDiag[k] : Z[j * pitch + k];
  }
  __syncthreads();
  if (p.bond) {
    cplx* Cb = p.bond + (long)b * ncap * ncap;
    for (int e = lane; e < ncap * ncap; e += 64) {
      const int r = e / ncap, c = e - r * ncap;
      const int k = RIGHT ? r : c, j = RIGHT ? c : r;
      Cb[e] = (k < kn && j < n) ? G[k][j] : cplx{0.0, 0.0};
    }
  }
  if (lane == 0) {
    if (p.nloc) p.nloc[b] = RIGHT ? kn * chiR : chiL * kn;
    chi[RIGHT ? 1 : 0] = kn;
  }
  // Q columns: H_0 ... H_c e_c
  for (int c = 0; c < ncap; ++c) {
    cplx q{0.0, 0.0};
    if (c < kn) {
      if (lane == c) q.x = 1.0;
      for (int k = c; k >= 0; --k) {
        const real beta = sBeta[k];
        if (beta == 0.0) continue;
        const bool in = lane >= k && lane < rows_act;
        const cplx v = in ? Z[k * pitch + lane] : cplx{0.0, 0.0};
        const real wr = beta * wave_sum(fma(v.x, q.x, v.y * q.y));
        const real wi = beta * wave_sum(fma(v.x, q.y, -v.y * q.x));
        q.x -= wr * v.x - wi * v.y;
        q.y -= wr * v.y + wi * v.x;
      }
    }
    if (lane < R) A[site_index(c)] = q;
  }
  if (p.nb) small_absorb<!RIGHT>(p.nb + (long)b * p.nb_b0, G, d, ca, cb, p.cn, n, kn, ncap, lane);
}

template <bool RIGHT>
__global__ __launch_bounds__(64) void qr_site_small_kernel(SmallQrDesc p, int pitch) {
  __shared__ SmallLds sm;
  if (threadIdx.x == 0) { sm.Y = reinterpret_cast<cplx*>(small_dyn_lds); sm.pitch = pitch; }
  __syncthreads();
  int b = blockIdx.x;
  if (p.ids) b = p.ids[b];
  qr_site_small_body<RIGHT>(p, b, threadIdx.x, sm);
}

// ---- small bonds: a whole sweep of centre shifts in one kernel -------------------------------------
// The chain direction is sequential but every trajectory is independent: one wavefront walks its trajectory through the list of
// steps (local one-site factor of the dissipator, then an SVD or QR shift of the centre), so a dissipation sweep, the QR walk
// before a jump and the renormalising sweep after it are one launch each instead of one per site.
__global__ __launch_bounds__(64) void small_sweep_kernel(SmallSweepDesc p) {
  __shared__ SmallLds sm;
  if (threadIdx.x == 0) { sm.Y = reinterpret_cast<cplx*>(small_dyn_lds); sm.pitch = p.pitch; }
  __syncthreads();
  int b = blockIdx.x;
  if (p.ids) b = p.ids[b];
  const int lane = threadIdx.x;
  for (int t = 0; t < p.nsteps; ++t) {
    const SmallSweepStep st = p.steps[t];
    const SmallSiteRef site = p.sites[st.site];
    if (st.op != 0) {  // A[s'] = sum_s m[s'][s] A[s]  (d = 2), or a plain scalar
      cplx* A = site.A + (long)b * site.b0;
      const long plane = (long)site.ca * site.cb;
      for (long e = lane; e < plane; e += 64) {
        const cplx x0 = A[e], x1 = A[plane + e];
        cplx y0{0.0, 0.0}, y1{0.0, 0.0};
        if (st.op == 1) {
          cfma(y0, st.m[0], x0); cfma(y0, st.m[1], x1);
          cfma(y1, st.m[2], x0); cfma(y1, st.m[3], x1);
        } else {
          y0 = cplx{st.scal * x0.x, st.scal * x0.y};
          y1 = cplx{st.scal * x1.x, st.scal * x1.y};
        }
        A[e] = y0;
        A[plane + e] = y1;
      }
      __threadfence_block();
      __syncthreads();
    }
    if (st.kind == 0) continue;
    const bool towards_right = (st.kind == 1 || st.kind == 3);
    const SmallSiteRef nb = p.sites[towards_right ? st.site + 1 : st.site - 1];
    if (st.kind <= 2) {
      SmallShiftDesc q;
      q.site = site.A; q.site_b0 = site.b0; q.nb = nb.A; q.nb_b0 = nb.b0;
      q.d = p.d; q.ca = site.ca; q.cb = site.cb; q.cn = towards_right ? nb.cb : nb.ca;
      q.chi = p.chi + st.site; q.chi_stride = p.chi_stride; q.threshold = p.threshold; q.min_keep = p.min_keep;
      q.ids = nullptr; q.nb0 = 0; q.flags = p.flags;
      if (towards_right) svd_shift_small_body<false>(q, b, lane, sm);
      else svd_shift_small_body<true>(q, b, lane, sm);
    } else {
      SmallQrDesc q;
      q.site = site.A; q.site_b0 = site.b0; q.bond = nullptr; q.nb = nb.A; q.nb_b0 = nb.b0;
      q.d = p.d; q.ca = site.ca; q.cb = site.cb; q.cn = towards_right ? nb.cb : nb.ca;
      q.chi = p.chi + st.site; q.chi_stride = p.chi_stride; q.nloc = nullptr; q.ids = nullptr; q.nb0 = 0;
      if (towards_right) qr_site_small_body<true>(q, b, lane, sm);
      else qr_site_small_body<false>(q, b, lane, sm);
    }
    __threadfence_block();
    __syncthreads();
  }
}

// Column norms of the X part, descending rank sort, truncation (svd_utils.py:22-104).
__global__ __launch_bounds__(256) void svd_finish_kernel(TruncSpec d, SvdWorkspace w, int ncols_pad, int rx, int rtot, const int* ids) {
  __shared__ real sN[1024];
  __shared__ int sPerm[1024];
  int b = blockIdx.x;
  if (ids) b = ids[b];
  const cplx* Yb = w.Y + (long)b * w.y_b0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c = wave; c < ncols_pad; c += 4) {
    const cplx* col = Yb + (long)c * rtot;
    real acc = 0.0;
    for (int r = lane; r < rx; r += 64) {
      cplx v = col[r];
      acc = fma(v.x, v.x, acc);
      acc = fma(v.y, v.y, acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) sN[c] = acc;
  }
  __syncthreads();
  for (int c = tid; c < ncols_pad; c += 256) {
    const real v = tjm_sort_key(sN[c]);
    int rank = 0;
    for (int o = 0; o < ncols_pad; ++o) {
      const real u = tjm_sort_key(sN[o]);
      rank += (u > v || (u == v && o < c)) ? 1 : 0;
    }
    sPerm[rank] = c;
  }
  __syncthreads();
  int* perm = w.perm + (long)b * ncols_pad;
  real* norms = w.norms + (long)b * ncols_pad;
  for (int k = tid; k < ncols_pad; k += 256) {
    perm[k] = sPerm[k];
    norms[k] = sqrt(sN[sPerm[k]]);
  }
  __syncthreads();
  if (tid == 0) {
    const int m_act = d.mulA * d.chiA[(long)b * d.chi_stride];
    const int n_act = d.mulB * d.chiB[(long)b * d.chi_stride];
    int nsv = m_act < n_act ? m_act : n_act;
    if (nsv > ncols_pad) nsv = ncols_pad;
    if (d.overflow_each) {  // one word per trajectory instead of the sticky flag: the caller decides whose clip counts
      d.overflow_each[b] = 0;
      d.overflow = d.overflow_each + b;
    }
    const int keep = truncation_keep(d, nsv, [&](int k) { return sqrt(sN[sPerm[k]]); });
    d.chiOut[(long)b * d.chi_stride] = keep;
    // kept columns at the rounding-noise floor were never rotated: their normalised columns are not orthogonal to the rest,
    // so the caller must not take them as singular vectors (n_active[2] != 0 selects the re-orthonormalising path)
    if (keep > 0 && w.n_active != nullptr) {
      const real s0 = sqrt(sN[sPerm[0]]);
      if (sqrt(sN[sPerm[keep - 1]]) <= TJM_RANK_TOL * s0) atomicOr(w.n_active + 2, 1);
    }
  }
  if (d.spectrum) {
    for (int k = tid; k < d.spec_ld; k += 256) d.spectrum[(long)b * d.spec_ld + k] = (k < ncols_pad) ? sqrt(sN[sPerm[k]]) : 0.0;
  }
}

// out[b][k*o_k + r1*o_r1 + r0*o_r0] = scale_k * op(Y[perm[k]][row_off + r1*n_r0 + r0])  for k < keep, else 0
__global__ __launch_bounds__(256) void svd_extract_kernel(ExtractDesc x, SvdWorkspace w, int ncols_pad, int rtot, const int* chi_keep,
                                                         int chi_stride, const int* ids) {
  int b = blockIdx.y;
  if (ids) b = ids[b];
  const cplx* Yb = w.Y + (long)b * w.y_b0;
  const int* perm = w.perm + (long)b * ncols_pad;
  const real* sig = w.norms + (long)b * ncols_pad;
  const int keep = chi_keep[(long)b * chi_stride];
  cplx* out = x.out + (long)b * x.out_b0;
  const long nrows = (long)x.n_r1 * x.n_r0;
  const long total = nrows * x.n_k;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    // iterate with the row index fastest: contiguous reads of a Y column
    const int k = (int)(e / nrows);
    const long r = e % nrows;
    const long ro = x.row_map ? x.row_map[(long)b * x.row_map_ld + r] : r;
    const int r1 = (int)(ro / x.n_r0), r0 = (int)(ro % x.n_r0);
    cplx v{0.0, 0.0};
    if (k < keep) {
      v = Yb[(long)perm[k] * rtot + x.row_off + r];
      if (x.conj) v.y = -v.y;
      if (x.scale_mode == 1) { v.x *= sig[k]; v.y *= sig[k]; }
      else if (x.scale_mode == 2) { const real inv = (sig[k] > 0.0) ? 1.0 / sig[k] : 0.0; v.x *= inv; v.y *= inv; }
      else if (x.scale_mode == 3) { const real r = sqrt(sig[k]); v.x *= r; v.y *= r; }
      else if (x.scale_mode == 4) { const real inv = (sig[k] > 0.0) ? 1.0 / sqrt(sig[k]) : 0.0; v.x *= inv; v.y *= inv; }
      else if (x.scale_mode == 5) {  // normalised; an exactly zero column (structural padding) becomes the unit vector of its own index
        if (sig[k] > 0.0) { const real inv = 1.0 / sig[k]; v.x *= inv; v.y *= inv; }
        else v = cplx{(r == perm[k]) ? real(1.0) : real(0.0), 0.0};
      }
    }
    out[(long)k * x.o_k + (long)r1 * x.o_r1 + (long)r0 * x.o_r0] = v;
  }
}

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Optional live timing of the dominant kernel (jacobi_cross16x_kernel) with HIP events on the launch stream.  Several engines of one
// process run their factorisations on host threads of their own (bench.py --engines): the event pool and the pending samples belong
// to the calling thread, the totals are shared behind a mutex.
struct CrossProfile {
  int every = 0;  // 0 = off, otherwise every N-th launch is bracketed by events
  double total_ms = 0.0, total_bytes = 0.0;
  long samples = 0;
};
struct CrossSampler {
  long counter = 0;
  std::vector<hipEvent_t> pool;
  std::vector<std::pair<int, real>> pending;  // (event pair index, bytes)
  size_t used = 0;
};
CrossProfile g_prof;
thread_local CrossSampler t_prof;
// jacobi_solve: the convergence counts of sweep k are read behind sweep k + 1.  Events belong to a device: one pair per device the
// calling thread has used (a process may drive engines on several GPUs: Simulator(device="cuda:k"))
struct SweepEvents { hipEvent_t ev[16][2] = {}; };
thread_local SweepEvents t_sweep_events;
std::mutex g_prof_mutex;
// Work actually executed by the tiled Jacobi kernels since the last reset (read once per sweep with the convergence flag):
// rotation slots x rows (every pair of a visited tile costs its dot product and its - possibly identity - rotation) and
// applied rotations x rows; bench.py turns them into executed flops next to the nominal 88 n^3.
struct JacobiWork { double slot_rows = 0.0, rotation_rows = 0.0; long sweeps = 0, solves = 0; };
JacobiWork g_work;
const bool g_debug = getenv("TJM_DEBUG_SVD") != nullptr;

void prof_collect() {
  double ms_sum = 0.0, bytes_sum = 0.0;
  long n = 0;
  for (auto& p : t_prof.pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, t_prof.pool[2 * p.first], t_prof.pool[2 * p.first + 1]) == hipSuccess) {
      ms_sum += ms;
      bytes_sum += p.second;
      ++n;
    }
  }
  t_prof.pending.clear();
  t_prof.used = 0;
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  g_prof.total_ms += ms_sum;
  g_prof.total_bytes += bytes_sum;
  g_prof.samples += n;
}

}  // namespace

bool svd_shift_small_fits(int d, int ca, int cb, bool left) {
  static const bool off = getenv("TJM_NO_SMALL_SHIFT") != nullptr;
  if (off) return false;
  const int rows = left ? d * cb : d * ca, cols = left ? ca : cb;
  return rows <= 64 && cols <= SMALL_MAXN;
}

// rows of the LDS columns of the one-wavefront kernels: the smallest of 16, 32, 64 that holds `rows`
static int small_pitch(int rows) { return rows <= 16 ? 16 : (rows <= 32 ? 32 : 64); }

int small_sweep_pitch(int rows) { return small_pitch(rows); }

int launch_qr_site_small(const SmallQrDesc& p, bool right, hipStream_t s) {
  if (p.nb0 <= 0) return TJM_OK;
  const int pitch = small_pitch(right ? p.d * p.ca : p.d * p.cb);
  const size_t lds = (size_t)SMALL_MAXN * pitch * sizeof(cplx);
  if (right) hipLaunchKernelGGL(qr_site_small_kernel<true>, dim3(p.nb0), dim3(64), lds, s, p, pitch);
  else hipLaunchKernelGGL(qr_site_small_kernel<false>, dim3(p.nb0), dim3(64), lds, s, p, pitch);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

int launch_small_sweep(const SmallSweepDesc& p, hipStream_t s) {
  if (p.nb0 <= 0 || p.nsteps <= 0) return TJM_OK;
  hipLaunchKernelGGL(small_sweep_kernel, dim3(p.nb0), dim3(64), (size_t)SMALL_MAXN * p.pitch * sizeof(cplx), s, p);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

int launch_svd_shift_small(const SmallShiftDesc& p, bool left, hipStream_t s) {
  if (p.nb0 <= 0) return TJM_OK;
  const int pitch = small_pitch(left ? p.d * p.cb : p.d * p.ca);
  const size_t lds = (size_t)SMALL_MAXN * pitch * sizeof(cplx);
  if (left) hipLaunchKernelGGL(svd_shift_small_kernel<true>, dim3(p.nb0), dim3(64), lds, s, p, pitch);
  else hipLaunchKernelGGL(svd_shift_small_kernel<false>, dim3(p.nb0), dim3(64), lds, s, p, pitch);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}


void profile_enable(int every) {
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  g_prof.every = every;
  g_prof.total_ms = 0.0;
  g_prof.total_bytes = 0.0;
  g_prof.samples = 0;
}

void jacobi_work_get(double* out4, bool reset) {
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  out4[0] = g_work.slot_rows; out4[1] = g_work.rotation_rows; out4[2] = (double)g_work.sweeps; out4[3] = (double)g_work.solves;
  if (reset) g_work = JacobiWork();
}

void profile_get(double* total_ms, double* total_bytes, long* samples) {
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  *total_ms = g_prof.total_ms;
  *total_bytes = g_prof.total_bytes;
  *samples = g_prof.samples;
}

long svd_y_elems(int max_dim) {
  const int p32 = round_up(max_dim, 32);
  return (long)p32 * round_up(round_up(max_dim, 16) + p32, 64);
}

// One layout of the Jacobi workspace for every user (engine, stand-alone ABI calls): buffers behind base, returns bytes used.
size_t svd_carve(SvdWorkspace& w, char* base, int max_dim, int B) {
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += (bytes + 255) / 256 * 256; return p; };
  const int p = round_up(max_dim, 32);
  w.y_b0 = svd_y_elems(max_dim);
  w.Y = reinterpret_cast<cplx*>(take((size_t)B * w.y_b0 * sizeof(cplx)));
  w.norms = reinterpret_cast<real*>(take((size_t)B * p * sizeof(real)));
  w.perm = reinterpret_cast<int*>(take((size_t)B * p * sizeof(int)));
  w.fro2 = reinterpret_cast<real*>(take((size_t)B * sizeof(real)));
  w.rec = reinterpret_cast<real*>(take((size_t)B * (MAXBLK / 4) * REC_PER_VISIT * 4 * sizeof(real)));
  w.stamps = reinterpret_cast<int*>(take((size_t)B * STAMP_STRIDE * sizeof(int)));
  w.nrot = reinterpret_cast<int*>(take((size_t)B * sizeof(int)));
  w.done = reinterpret_cast<int*>(take((size_t)B * sizeof(int)));
  w.n_active = reinterpret_cast<int*>(take(256));
  return off;
}

size_t svd_workspace_bytes(int max_dim, int B) {
  SvdWorkspace w;
  return svd_carve(w, nullptr, max_dim, B) + 4096;
}

int jacobi_solve(const JacobiSource& src, const TruncSpec& tr, const SvdWorkspace& w, hipStream_t s, JacobiShape* shape_out,
                 int* sweeps_out, bool accumulate, const JacobiOpts* opts) {
  if (src.nb0 <= 0) return TJM_OK;
  const JacobiOpts defaults;
  const JacobiOpts& op = opts ? *opts : defaults;
  static const bool no16 = getenv("TJM_NO_TILE16") != nullptr;
  static const bool no_split = getenv("TJM_NO_SPLIT") != nullptr;
  // without the accumulated unitary the X rows are padded to whole 64-row groups (zero rows cost nothing in the dot products and
  // let the split X kernel serve every height up to 512)
  // X-only solves of the complex64 arithmetic keep up to 1024 rows of a column in the registers of one wavefront (16 row groups, in
  // pairs: whole groups of 128 rows above 512) - the splits of bonds up to 512 (round 6; TJM_NO_X1024: the 8-column kernels as before)
#ifdef TJM_F32
  static const bool no_x1024 = getenv("TJM_NO_X1024") != nullptr;
  const int x_rows_max = no_x1024 ? 512 : 1024;
#else
  const int x_rows_max = 512;
#endif
  int rx_top_ = accumulate ? round_up(src.rx, 16) : round_up(src.rx, 64);
  if (!accumulate && rx_top_ > 512 && rx_top_ <= x_rows_max) rx_top_ = round_up(src.rx, 128);
  const int rx_top = rx_top_;
  const int ncols32 = round_up(src.ncols, 32);
  // split X / W scheme: 16-column blocks; X rows in whole groups of 64 up to 512 (with the accumulated unitary: exactly 256 or 512
  // and W rows in groups of 64)
  const int wrows32 = accumulate ? ncols32 : 0;  // rows of the accumulated unitary stacked under X
  const bool split16 = !no16 && !no_split && ncols32 >= 32 && ((!accumulate && rx_top <= x_rows_max) || ((rx_top == 256 || rx_top == 512) && ncols32 % 64 == 0)) &&
                       (w.rec != nullptr || !accumulate) && src.nb0 <= 65535 && round_up(rx_top + wrows32, 64) <= 64 * MAXRK;
  // fused 16-column blocks for the smaller matrices (two stacked columns per wavefront fit registers and LDS up to 512 rows)
  const bool tile16 = split16 || (!no16 && ncols32 >= 32 && round_up(rx_top + wrows32, 64) <= 512);
  const int ncols_pad = tile16 ? ncols32 : round_up(src.ncols, 16);
  const int rtot = round_up(rx_top + (accumulate ? ncols_pad : 0), 64);
  if (rtot > 64 * MAXRK) return TJM_ERR_NOT_IMPLEMENTED;  // register-resident columns: rx + ncols <= 1024
  if ((long)ncols_pad * rtot > w.y_b0) return TJM_ERR_WORKSPACE;
  const bool big = rtot > 512;  // 16 row groups per column instead of 8
  static const bool no_lds = getenv("TJM_NO_LDS_JACOBI") != nullptr;
  if (!no_lds && ncols_pad <= 64 && rtot <= 128) {  // the whole problem fits one workgroup's LDS: all sweeps in one launch
    static std::atomic<bool> lds_attr{false};  // several engines of one process call this from their own host threads
    if (!lds_attr.load(std::memory_order_acquire)) {
      TJM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_lds_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
      TJM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_lds_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024));
      lds_attr.store(true, std::memory_order_release);
    }
    const long total = (long)ncols_pad * rtot;
    int gx = (int)((total + 1023) / 1024);
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(jacobi_load_kernel, dim3(gx, src.nb0), dim3(256), 0, s, src, w.Y, w.y_b0, ncols_pad, rx_top, rtot);
    JacobiArgs g;
    g.Y = w.Y; g.y_b0 = w.y_b0; g.rtot = rtot; g.rx = rx_top; g.nblk = ncols_pad / NB; g.tol2 = TJM_JACOBI_TOL2; g.fro2 = nullptr; g.nrot = nullptr;
    g.done = nullptr; g.ids = src.ids; g.round = 0; g.stamps = nullptr; g.clock = 0; g.mode = 0; g.rec = nullptr; g.work = nullptr; g.fold = 0;
    g.floor_scale = op.floor_scale;
    TJM_HIP_CHECK(hipMemsetAsync(w.n_active, 0, 3 * sizeof(int), s));
    const size_t lds_bytes = (size_t)ncols_pad * rtot * sizeof(cplx) + (size_t)ncols_pad * sizeof(real) + 32 * sizeof(int);
    if (rtot <= 64) hipLaunchKernelGGL(jacobi_lds_kernel<1>, dim3(src.nb0), dim3(1024), lds_bytes, s, g, ncols_pad, 40, w.n_active);
    else hipLaunchKernelGGL(jacobi_lds_kernel<2>, dim3(src.nb0), dim3(1024), lds_bytes, s, g, ncols_pad, 40, w.n_active);
    hipLaunchKernelGGL(svd_finish_kernel, dim3(src.nb0), dim3(256), 0, s, tr, w, ncols_pad, rx_top, rtot, src.ids);
    TJM_HIP_CHECK(hipGetLastError());
    TJM_HIP_CHECK(hipMemcpyAsync(w.h_pinned, w.n_active, sizeof(int), hipMemcpyDeviceToHost, s));
    TJM_HIP_CHECK(hipStreamSynchronize(s));
    if (sweeps_out) *sweeps_out = 0;
    if (shape_out) { shape_out->ncols_pad = ncols_pad; shape_out->rx_top = rx_top; shape_out->rtot = rtot; }
    return (*w.h_pinned == 0) ? TJM_OK : TJM_ERR_NUMERIC;
  }
  static std::atomic<bool> attr_set{false};  // several engines of one process call this from their own host threads
  if (!attr_set.load(std::memory_order_acquire)) {
    const int big_lds = 136 * 1024;
    TJM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_cross_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    TJM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_cross_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, big_lds));
    TJM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_diag_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    TJM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_diag_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, big_lds));
    TJM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_cross16_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#define TJM_X16_ATTR(K, BYTES)                                                                                                                         \
  TJM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_cross16x_kernel<K, false>), hipFuncAttributeMaxDynamicSharedMemorySize, BYTES)); \
  TJM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_cross16x_kernel<K, true>), hipFuncAttributeMaxDynamicSharedMemorySize, BYTES))
    TJM_X16_ATTR(4, 80 * 1024);
    TJM_X16_ATTR(5, big_lds);
    TJM_X16_ATTR(6, big_lds);
    TJM_X16_ATTR(7, big_lds);
    TJM_X16_ATTR(8, big_lds);
#ifdef TJM_F32
    TJM_X16_ATTR(10, big_lds);
    TJM_X16_ATTR(12, big_lds);
    TJM_X16_ATTR(14, big_lds);
    TJM_X16_ATTR(16, big_lds);
#endif
#undef TJM_X16_ATTR
    attr_set.store(true, std::memory_order_release);
  }
  const size_t lds = (size_t)NB * rtot * sizeof(cplx) + NB * sizeof(real) + 16 * sizeof(int);
  const int tb = (src.nb0 + 255) / 256;
  hipLaunchKernelGGL(svd_reset_kernel, dim3(tb), dim3(256), 0, s, w.nrot, w.done, src.nb0, src.ids, w.n_active);
  {
    const long total = (long)ncols_pad * rtot;
    int gx = (int)((total + 1023) / 1024);
    if (gx > 256) gx = 256;
    // op.preloaded: the caller has written X into Y already (column-major, pitch rtot; needs rx_top == rx and no padding columns)
    if (op.preloaded && (rx_top != src.rx || ncols_pad != src.ncols || rtot != rx_top || accumulate)) return TJM_ERR_ARG;
    if (!op.preloaded) hipLaunchKernelGGL(jacobi_load_kernel, dim3(gx, src.nb0), dim3(256), 0, s, src, w.Y, w.y_b0, ncols_pad, rx_top, rtot);
    hipLaunchKernelGGL(jacobi_fro_kernel, dim3(src.nb0), dim3(256), 0, s, w.Y, w.y_b0, ncols_pad, rx_top, rtot, w.fro2, src.ids);
    hipLaunchKernelGGL(jacobi_stamp_init_kernel, dim3(src.nb0), dim3(256), 0, s, w.Y, w.y_b0, ncols_pad / NB, rx_top, rtot, w.stamps, src.ids);
  }
  JacobiArgs g;
  g.Y = w.Y;
  g.y_b0 = w.y_b0;
  g.rtot = rtot;
  g.rx = rx_top;
  g.nblk = ncols_pad / NB;
  g.tol2 = op.tol2;  // relative off-diagonal tolerance, squared (default 1e-13 in fp64)
#ifdef TJM_F32
  {
    // fp32: the rounding noise of an inner product over r rows is ~ eps sqrt(r) |p| |q| / c0; at 1024 rows the fixed tolerance 2e-6 sits
    // 3.9 sigma above it (measured: ~50 of 5e5 pairs "rotate" in every late sweep of a 1024 x 512 centre shift, sweep after sweep),
    // so the tolerance follows the rows: tol >= TJM_JACOBI_TOL_ROWS x eps x sqrt(rows) (1.6: six sigma at 1024 rows = 3e-6, below the
    // fixed 2e-6 up to 450 rows).  Solves that stop by fraction (the mixed split's first phase) keep the fixed tolerance.
    static const double c_rows = getenv("TJM_JACOBI_TOL_ROWS") ? atof(getenv("TJM_JACOBI_TOL_ROWS")) : 1.6;
    if (op.stop_fraction <= 0.0 && c_rows > 0.0) {
      const double t = c_rows * 5.96e-8 * std::sqrt((double)rx_top);
      if ((real)(t * t) > g.tol2) g.tol2 = (real)(t * t);
    }
  }
#endif
  g.floor_scale = op.floor_scale;
  g.fro2 = w.fro2;
  g.nrot = w.nrot;
  g.done = w.done;
  g.ids = src.ids;
  g.round = 0;
  g.stamps = w.stamps;
  g.clock = 1;
  g.mode = 0;
  g.work = w.n_active + 16 + 4;  // (slot of the first sweep; zeroed by svd_reset_kernel above)
  { std::lock_guard<std::mutex> lock(g_prof_mutex); ++g_work.solves; }
  if (g.nblk > MAXBLK) return TJM_ERR_NOT_IMPLEMENTED;
  const int nrounds = tile16 ? (g.nblk / 2 - 1) : (g.nblk - 1);
  const int npairs = tile16 ? g.nblk / 4 : g.nblk / 2;
  const size_t lds16 = (size_t)2 * NB * rtot * sizeof(cplx) + 2 * NB * sizeof(real) + 16 * sizeof(int);
#ifdef TJM_F32
  const int rx_slot = (rx_top + 127) / 128 * 128;  // ColFrag keeps row groups in pairs
#else
  const int rx_slot = rx_top;
#endif
  const size_t lds16x = (size_t)2 * NB * rx_slot * sizeof(cplx) + 2 * NB * sizeof(real) + 16 * sizeof(int);
  // four columns of each block per wavefront (jacobi_cross16q_kernel): X-only solves of the complex64 arithmetic up to 512 rows (round 6:
  // every such solve, not only the complex64 phase of the mixed split - the rotation rule is the same, the order inside a visit differs;
  // TJM_QUAD_ONLY_MIXED: as before, only where the caller asks for it; 256 < rows <= 512: eight row groups, two wavefronts per SIMD)
  bool quad16 = false;
#ifdef TJM_F32
  {
    static const bool no_quad = getenv("TJM_NO_QUAD_TILE") != nullptr;
    static const bool only_mixed = getenv("TJM_QUAD_ONLY_MIXED") != nullptr;
    static const bool no_quad512 = getenv("TJM_NO_QUAD512") != nullptr;
    quad16 = !no_quad && (op.quad || !only_mixed) && split16 && !accumulate && rx_top <= (no_quad512 ? 256 : 512);
    static std::atomic<bool> q16_attr{false};
    if (quad16 && rx_top > 256 && !q16_attr.load(std::memory_order_acquire)) {
#define TJM_Q16_ATTR(K)                                                                                                                              \
  TJM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_cross16q_kernel<K, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024)); \
  TJM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_cross16q_kernel<K, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024))
      TJM_Q16_ATTR(5); TJM_Q16_ATTR(6); TJM_Q16_ATTR(7); TJM_Q16_ATTR(8);
#undef TJM_Q16_ATTR
      q16_attr.store(true, std::memory_order_release);
    }
  }
#endif
  const size_t lds16q = lds16x;
  // three rounds per load (jacobi_quad64_kernel): groups of 16 blocks of 16 columns = the points of AG(2,4), the groups against each
  // other two rounds per load; TJM_NO_QUAD64: one round per launch
  bool quad64 = false;
  [[maybe_unused]] const int ngroups = ncols_pad / 256;
#ifdef TJM_F32
  {
    static const bool no_quad64 = getenv("TJM_NO_QUAD64") != nullptr;
    static const bool no_quad64_groups = getenv("TJM_NO_QUAD64_GROUPS") != nullptr;
    quad64 = quad16 && !no_quad64 && ncols_pad % 256 == 0 && ngroups <= (no_quad64_groups ? 1 : 2) && (rx_top == 256 || rx_top == 512);  // (the kernel plays any number of groups; the X-only solves of the path have one or two, and only those are tested)
    static std::atomic<bool> q64_attr{false};
    if (quad64 && !q64_attr.load(std::memory_order_acquire)) {
      TJM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_quad64_kernel<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
      TJM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_quad64_kernel<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
      TJM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_quad64_kernel<8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024));
      TJM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_quad64_kernel<8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024));
      q64_attr.store(true, std::memory_order_release);
    }
  }
#endif
  [[maybe_unused]] const size_t lds64 = 2 * lds16x + 16 * sizeof(real) + 16 * sizeof(int);
  g.ngroups = ngroups;
  g.ground = 0;
  static const int q64_ablate = getenv("TJM_Q64_ABLATE") ? atoi(getenv("TJM_Q64_ABLATE")) : 0;
  g.ablate = q64_ablate;
  // launches of one sweep of the grouped schedule: 5 plane classes (three rounds each), then for every round of the circle method on
  // the groups 8 shifts (two rounds each)
  const int q64_geven = ngroups + (ngroups & 1);
  const int q64_launches = quad64 ? 5 + (ngroups > 1 ? 8 * (q64_geven - 1) : 0) : 0;
  g.rec = accumulate ? w.rec : nullptr;
  // in-block pairs folded into the tile visits (jacobi_cross16x_kernel): needs the 15 tournament rounds of a 16-column block inside
  // one sweep, and no rotation record (the W replay kernel knows the fixed column assignment only)
  static const bool no_fold = getenv("TJM_NO_FOLD") != nullptr;
  g.fold = (!no_fold && split16 && !accumulate && nrounds >= 15) ? 1 : 0;
  if (g_debug && src.ncols >= 128)
    fprintf(stderr, "[svd] ncols_pad %d rx_top %d rtot %d batch %d: %s%s\n", ncols_pad, rx_top, rtot, src.nb0,
            quad64 ? "jacobi_quad64_kernel" : quad16 ? "jacobi_cross16q_kernel" : split16 ? "jacobi_cross16x_kernel" : tile16 ? "jacobi_cross16_kernel" : "jacobi_cross_kernel",
            g.fold ? " (in-block pairs folded)" : "");
  const int max_sweeps = op.max_sweeps;
  // op.stop_fraction: a caller that refines the result anyway (the complex64 phase of the mixed-precision split) does not pay for the
  // sweeps that only confirm convergence: a trajectory is done after a sweep that rotated at most this fraction of its pairs
  const int stop_at = op.stop_fraction > 0.0 ? (int)(op.stop_fraction * 0.5 * (double)ncols_pad * (ncols_pad - 1)) : 0;
  const int nb = src.nb0;
  const int tbc = (nb + 255) / 256;
  int sweep_c = 0;
  int n_live = nb;
  bool conv_c = false;
  bool late = op.late_start;  // the previous sweep rotated less than 70 % of its pairs: the check-first variant of the tile kernel
  static const bool no_late = getenv("TJM_NO_LATE_SWEEPS") != nullptr;
  static const bool sync_each = getenv("TJM_SVD_SYNC_EACH") != nullptr;
  bool pipelined = !sync_each && g_prof.every == 0;  // (the sampler reads its events at the end of every sweep)
  hipEvent_t* t_sweep_ev = nullptr;
  if (pipelined) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { (void)hipGetLastError(); pipelined = false; }
    else {
      t_sweep_ev = t_sweep_events.ev[dev];
      if (!t_sweep_ev[0]) {
        if (hipEventCreate(&t_sweep_ev[0]) != hipSuccess || hipEventCreate(&t_sweep_ev[1]) != hipSuccess) {
          (void)hipGetLastError();
          t_sweep_ev[0] = t_sweep_ev[1] = nullptr;
          pipelined = false;
        }
      }
    }
  }
  int sweeps_done = 0;
  bool extra_queued = false;
  for (; sweep_c < max_sweeps && !conv_c; ++sweep_c) {
    int* const slot = w.n_active + 16 + 8 * (sweep_c & 1);  // this sweep's counters; the other slot is the next sweep's
    g.work = slot + 4;
    ++g.clock;
    g.mode = 0;
    if (!g.fold) {
      if (big) hipLaunchKernelGGL(jacobi_diag_kernel<16>, dim3(g.nblk, nb), dim3(256), lds, s, g);
      else hipLaunchKernelGGL(jacobi_diag_kernel<8>, dim3(g.nblk, nb), dim3(256), lds, s, g);
    }
    if (tile16 && !g.fold) {  // pairs between the two 8-column halves of every 16-column block
      ++g.clock;
      g.mode = 1;
      if (big) hipLaunchKernelGGL(jacobi_cross_kernel<16>, dim3(g.nblk / 2, nb), dim3(512), lds, s, g);
      else hipLaunchKernelGGL(jacobi_cross_kernel<8>, dim3(g.nblk / 2, nb), dim3(512), lds, s, g);
      g.mode = 0;
    }
    for (int r = 0; r < (quad64 ? q64_launches : nrounds); ++r) {
      g.round = r;
      if (quad64) {
        g.mode = r < 5 ? 0 : 1;
        if (r >= 5) { g.ground = (r - 5) >> 3; g.round = (r - 5) & 7; }
      }
      ++g.clock;
      const bool timed = g_prof.every > 0 && split16 && (t_prof.counter++ % g_prof.every == 0);  // only the dominant (split X) kernel is sampled
      int slot = -1;
      if (timed) {
        slot = (int)t_prof.used++;
        while (t_prof.pool.size() < 2 * t_prof.used) {
          hipEvent_t ev;
          TJM_HIP_CHECK(hipEventCreate(&ev));
          t_prof.pool.push_back(ev);
        }
        TJM_HIP_CHECK(hipEventRecord(t_prof.pool[2 * slot], s));
      }
      if (quad64) {  // parallel class r of the block planes: three rounds (clock, clock + 1, clock + 2) in one launch; group pairs: two
#ifdef TJM_F32
        const dim3 grid64(g.mode == 0 ? 4 * ngroups : 8 * (q64_geven / 2), nb);
        if (rx_top == 256) {
          if (late) hipLaunchKernelGGL((jacobi_quad64_kernel<4, true>), grid64, dim3(512), lds64, s, g);
          else hipLaunchKernelGGL((jacobi_quad64_kernel<4, false>), grid64, dim3(512), lds64, s, g);
        } else {
          if (late) hipLaunchKernelGGL((jacobi_quad64_kernel<8, true>), grid64, dim3(512), lds64, s, g);
          else hipLaunchKernelGGL((jacobi_quad64_kernel<8, false>), grid64, dim3(512), lds64, s, g);
        }
        const int nquads = g.mode == 0 ? 4 * ngroups : 8 * (ngroups / 2);
#else
        const int nquads = 0;
#endif
        g.clock += g.mode == 0 ? 2 : 1;
        if (timed) {
          TJM_HIP_CHECK(hipEventRecord(t_prof.pool[2 * slot + 1], s));
          t_prof.pending.emplace_back(slot, (real)nquads * n_live * 8.0 * NB * rx_top * sizeof(cplx) * 2.0);
        }
      } else if (quad16) {
        const dim3 gridq(npairs, nb), blockq(256);
#define TJM_Q16_LAUNCH(K)                                                                     \
  if (late) hipLaunchKernelGGL((jacobi_cross16q_kernel<K, true>), gridq, blockq, lds16q, s, g); \
  else hipLaunchKernelGGL((jacobi_cross16q_kernel<K, false>), gridq, blockq, lds16q, s, g)
        switch (rx_top / 64) {
          case 1: TJM_Q16_LAUNCH(1); break;
          case 2: TJM_Q16_LAUNCH(2); break;
          case 3: TJM_Q16_LAUNCH(3); break;
          case 4: TJM_Q16_LAUNCH(4); break;
#ifdef TJM_F32
          case 5: TJM_Q16_LAUNCH(5); break;
          case 6: TJM_Q16_LAUNCH(6); break;
          case 7: TJM_Q16_LAUNCH(7); break;
          default: TJM_Q16_LAUNCH(8); break;
#else
          default: break;
#endif
        }
#undef TJM_Q16_LAUNCH
        if (timed) {
          TJM_HIP_CHECK(hipEventRecord(t_prof.pool[2 * slot + 1], s));
          t_prof.pending.emplace_back(slot, (real)npairs * n_live * 4.0 * NB * rx_top * sizeof(cplx) * 2.0);
        }
      } else if (split16) {
        const dim3 gridx(npairs, nb), blockx(512);
#define TJM_X16_LAUNCH(K)                                                                     \
  if (late) hipLaunchKernelGGL((jacobi_cross16x_kernel<K, true>), gridx, blockx, lds16x, s, g); \
  else hipLaunchKernelGGL((jacobi_cross16x_kernel<K, false>), gridx, blockx, lds16x, s, g)
        switch (rx_top / 64) {  // row groups of 64 held in registers
          case 1: TJM_X16_LAUNCH(1); break;
          case 2: TJM_X16_LAUNCH(2); break;
          case 3: TJM_X16_LAUNCH(3); break;
          case 4: TJM_X16_LAUNCH(4); break;
          case 5: TJM_X16_LAUNCH(5); break;
          case 6: TJM_X16_LAUNCH(6); break;
          case 7: TJM_X16_LAUNCH(7); break;
#ifdef TJM_F32
          case 8: TJM_X16_LAUNCH(8); break;
          case 10: TJM_X16_LAUNCH(10); break;
          case 12: TJM_X16_LAUNCH(12); break;
          case 14: TJM_X16_LAUNCH(14); break;
          default: TJM_X16_LAUNCH(16); break;
#else
          default: TJM_X16_LAUNCH(8); break;
#endif
        }
#undef TJM_X16_LAUNCH
        if (timed) {  // the timed kernel is the X-rows kernel alone: its tile is the X part of the 32 columns, read + written once
          TJM_HIP_CHECK(hipEventRecord(t_prof.pool[2 * slot + 1], s));
          t_prof.pending.emplace_back(slot, (real)npairs * n_live * 4.0 * NB * rx_top * sizeof(cplx) * 2.0);
        }
        if (accumulate) hipLaunchKernelGGL(jacobi_cross16w_kernel, dim3(npairs, ncols_pad / 64, nb), dim3(64), 0, s, g, rx_top);
      } else if (tile16) hipLaunchKernelGGL(jacobi_cross16_kernel<8>, dim3(npairs, nb), dim3(512), lds16, s, g);
      else if (big) hipLaunchKernelGGL(jacobi_cross_kernel<16>, dim3(npairs, nb), dim3(512), lds, s, g);
      else hipLaunchKernelGGL(jacobi_cross_kernel<8>, dim3(npairs, nb), dim3(512), lds, s, g);
    }
    hipLaunchKernelGGL(svd_sweep_check_kernel, dim3(tbc), dim3(256), 0, s, w.nrot, w.done, slot, nb, g.ids, stop_at, w.n_active + 16 + 8 * ((sweep_c + 1) & 1));
    // The counts of the sweep travel to the host BEHIND the sweep (round 5): the host queues sweep k + 1 before it looks at sweep k, so
    // the device never idles for the round trip.  Nothing numerical depends on it - a trajectory's `done` flag lives on the device
    // (every kernel of a queued sweep returns at once for a finished trajectory), and the check-first variant the counts select
    // applies the same rotations as the unconditional one: the host only decides when to stop queuing.  Price: one masked sweep
    // (launches that find every trajectory done) at the end.  TJM_SVD_SYNC_EACH / the launch sampler: one round trip per sweep.
    int* hp = w.h_pinned + 16 + 8 * (sweep_c & 1);
    TJM_HIP_CHECK(hipMemcpyAsync(hp, slot, 5 * sizeof(int), hipMemcpyDeviceToHost, s));
    auto digest = [&](const int* h, int k) {  // the counts of sweep k have arrived
      conv_c = (h[0] == 0);
      {
        std::lock_guard<std::mutex> lock(g_prof_mutex);
        g_work.slot_rows += (double)h[4] * rx_top;
        g_work.rotation_rows += (double)h[1] * rx_top;
        if (h[4] > 0 || k == 0) ++g_work.sweeps;
      }
      if (g_debug && src.ncols >= 128) fprintf(stderr, "[svd] ncols %d rx %d sweep %d live %d rotations %d\n", src.ncols, src.rx, k, h[0], h[1]);
      n_live = h[0];
      // measured on the MI355X (headline step): fraction 0.25 -> 5.94, 0.7 -> 6.05, 1 (every sweep after the first) -> 5.81 trajectories/s
      static const double late_frac = getenv("TJM_LATE_FRACTION") ? atof(getenv("TJM_LATE_FRACTION")) : 0.7;
      late = !no_late && !accumulate &&
             (op.late_after_first || (double)h[1] < late_frac * 0.5 * (double)ncols_pad * (ncols_pad - 1) * std::max(n_live, 1));
    };
    if (!pipelined) {
      TJM_HIP_CHECK(hipStreamSynchronize(s));
      digest(hp, sweep_c);
      if (g_prof.every > 0) prof_collect();
      continue;
    }
    TJM_HIP_CHECK(hipEventRecord(t_sweep_ev[sweep_c & 1], s));
    if (sweep_c > 0) {
      TJM_HIP_CHECK(hipEventSynchronize(t_sweep_ev[(sweep_c - 1) & 1]));
      digest(w.h_pinned + 16 + 8 * ((sweep_c - 1) & 1), sweep_c - 1);
      if (conv_c) { sweeps_done = sweep_c; extra_queued = true; break; }  // sweep sweep_c is queued and masked: sweeps 0 ... sweep_c - 1 did the work
    }
    if (sweep_c + 1 == max_sweeps) {  // the last sweep the cap allows: wait for it
      TJM_HIP_CHECK(hipStreamSynchronize(s));
      digest(hp, sweep_c);
    }
  }
  if (extra_queued) sweep_c = sweeps_done;
  const int sweep = sweep_c;
  const bool converged = conv_c;
  if (sweeps_out) *sweeps_out = sweep;
  hipLaunchKernelGGL(svd_finish_kernel, dim3(src.nb0), dim3(256), 0, s, tr, w, ncols_pad, rx_top, rtot, src.ids);  // (n_active[2] was zeroed by svd_reset_kernel)
  TJM_HIP_CHECK(hipGetLastError());
  if (shape_out) {
    shape_out->ncols_pad = ncols_pad;
    shape_out->rx_top = rx_top;
    shape_out->rtot = rtot;
  }
  return (converged || op.allow_unconverged) ? TJM_OK : TJM_ERR_NUMERIC;
}

int svd_extract(const ExtractDesc& x, const SvdWorkspace& w, const JacobiShape& sh, const int* chi_keep, int chi_stride, int nb0,
                const int* ids, hipStream_t s) {
  const long total = (long)x.n_r1 * x.n_r0 * x.n_k;
  if (total <= 0 || nb0 <= 0) return TJM_OK;
  int gx = (int)((total + 1023) / 1024);
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(svd_extract_kernel, dim3(gx, nb0), dim3(256), 0, s, x, w, sh.ncols_pad, sh.rtot, chi_keep, chi_stride, ids);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

// Two-site split: theta (m x n, rows (s,a), cols (t,c)) -> left[d][capL][capM], right[d][capM][capR].
int svd_split(const SvdSplitDesc& d, const SvdWorkspace& w, hipStream_t s, int* sweeps_out) {
  if (d.nb0 <= 0) return TJM_OK;
  // distribution 2 = "sqrt" (decompositions.py:166-171: sqrt(S) into both factors, used by _sync_bond_dim of the dynamic sweep,
  // sweep_utils.py:146-160): the factorisation of distribution 0 with the scale of the singular values moved at extraction
  const bool sqrt_dist = d.distribution == 2;
  const bool orient0 = d.distribution != 1;
  {  // small bonds: everything in one kernel, unless a kept singular value sits at the rounding floor
    static const bool off = getenv("TJM_NO_SMALL_SHIFT") != nullptr;
    const int rows = orient0 ? d.m : d.n, cols = orient0 ? d.n : d.m;
    if (!off && !sqrt_dist && rows <= 64 && cols <= 16 && d.m == d.d * d.capL && d.n == d.d * d.capR && w.n_active != nullptr) {  // wider: the LDS-resident kernel
      TruncSpec tr;
      tr.trunc_mode = d.trunc_mode; tr.threshold = d.threshold; tr.max_bond = d.max_bond; tr.min_keep = d.min_keep;
      tr.cap = d.capM; tr.overflow = d.overflow;
      tr.chiA = d.chiL; tr.mulA = d.d; tr.chiB = d.chiR; tr.mulB = d.d; tr.chiOut = d.chiM; tr.chi_stride = d.chi_stride;
      tr.spectrum = d.spectrum; tr.spec_ld = d.spec_ld;
      TJM_HIP_CHECK(hipMemsetAsync(w.n_active + 2, 0, 2 * sizeof(int), s));
      const int pitch = small_pitch(rows);
      hipLaunchKernelGGL(svd_split_small_kernel<16>, dim3(d.nb0), dim3(64), (size_t)16 * pitch * sizeof(cplx), s, d, tr, w.n_active, pitch);
      TJM_HIP_CHECK(hipGetLastError());
      TJM_HIP_CHECK(hipMemcpyAsync(w.h_pinned, w.n_active + 2, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
      TJM_HIP_CHECK(hipStreamSynchronize(s));
      if (sweeps_out) *sweeps_out = 0;
      if (w.h_pinned[0] == 0 && w.h_pinned[1] == 0) return TJM_OK;
      // otherwise fall through: the general path handles rank-deficient kept sets (theta is untouched, the outputs are rewritten)
    }
  }
  JacobiSource src;
  src.src = d.theta; src.src_b0 = d.theta_b0; src.conj = orient0; src.tri = 0;
  if (orient0) {  // X = theta^H : rows = theta columns, columns = theta rows
    src.rx = d.n; src.ncols = d.m;
    src.r_n0 = d.n; src.s_r1 = 0; src.s_r0 = 1;
    src.c_n0 = d.m; src.s_c1 = 0; src.s_c0 = d.ld_theta;
  } else {                    // X = theta
    src.rx = d.m; src.ncols = d.n;
    src.r_n0 = d.m; src.s_r1 = 0; src.s_r0 = d.ld_theta;
    src.c_n0 = d.n; src.s_c1 = 0; src.s_c0 = 1;
  }
  src.nb0 = d.nb0; src.ids = d.ids;
  TruncSpec tr;
  tr.trunc_mode = d.trunc_mode; tr.threshold = d.threshold; tr.max_bond = d.max_bond; tr.min_keep = d.min_keep;
  tr.cap = d.capM; tr.overflow = d.overflow;
  tr.chiA = d.chiL; tr.mulA = d.d; tr.chiB = d.chiR; tr.mulB = d.d; tr.chiOut = d.chiM; tr.chi_stride = d.chi_stride;
  tr.spectrum = d.spectrum; tr.spec_ld = d.spec_ld;
  JacobiShape sh;
  int rc = jacobi_solve(src, tr, w, s, &sh, sweeps_out);
  if (rc != TJM_OK) return rc;
  // left[(s,a)][k]: rows (s,a) of the isometric factor U (dist 0: W part) or of U S (dist 1: X part)
  ExtractDesc xl;
  xl.out = d.left; xl.out_b0 = d.left_b0; xl.n_k = d.capM; xl.o_k = 1;
  xl.n_r1 = 1; xl.n_r0 = d.d * d.capL; xl.o_r1 = 0; xl.o_r0 = d.capM;
  xl.row_off = orient0 ? sh.rx_top : 0; xl.conj = 0; xl.scale_mode = sqrt_dist ? 3 : 0;
  if ((rc = svd_extract(xl, w, sh, d.chiM, d.chi_stride, d.nb0, d.ids, s)) != TJM_OK) return rc;
  // right[t][k][c] = conj of rows (t,c): S V^H (dist 0: X part) or V^H (dist 1: W part)
  ExtractDesc xr;
  xr.out = d.right; xr.out_b0 = d.right_b0; xr.n_k = d.capM; xr.o_k = d.capR;
  xr.n_r1 = d.d; xr.n_r0 = d.capR; xr.o_r1 = (long)d.capM * d.capR; xr.o_r0 = 1;
  xr.row_off = orient0 ? 0 : sh.rx_top; xr.conj = 1; xr.scale_mode = sqrt_dist ? 4 : 0;
  return svd_extract(xr, w, sh, d.chiM, d.chi_stride, d.nb0, d.ids, s);
}


// Two-site split with QR preconditioning: Z = theta (dist 0) or theta^H (dist 1) with its columns sorted by decreasing
// norm (the first step of a column-pivoted QR; it makes R more strongly graded and saves about a quarter of the sweeps)
// = Q R, Jacobi on R^H with accumulated W, isometric factor = Q W, weighted factor = rotated R^H.  Same outputs as svd_split.
// Doubly preconditioned variant for square theta (Drmac-Veselic with two QR factorisations): the OTHER orientation
// Z = theta^H (dist 0) / theta (dist 1), columns sorted, Z = Q R; R^H = Q1 R1; Jacobi on X = R1^H, X W = Y.  Then
//   Z = (Q Y) (Q1 W)^H :  isometric factor = Q1 W (rows in the sorted column order of Z), weighted factor = (Q Y).
// R1^H is closer to diagonal than R^H and the iteration needs about a fifth fewer sweeps again.
// Direct variant (default): the factorisation is taken in the orientation in which the ISOMETRIC factor is the left singular
// basis of the factored matrix, Z = theta (dist 0) / theta^H (dist 1), Z = Q R, R^H = Q1 R1, X = R1^H, Y = X W:
//   Z = (Q Ytilde) Sigma (Q1 W)^H ,  Ytilde = Y Sigma^-1 .
// One-sided Jacobi delivers the normalised columns of Y orthonormal to the convergence tolerance and with high relative
// accuracy, so Q Ytilde (reflectors times an orthonormal set) is the isometric factor without W, and the weighted factor
// is the exact projection of the input on it, isoᴴ theta resp. theta iso — one MFMA GEMM.  When a kept singular value sits
// at the noise floor (rank-deficient input with min_keep / threshold 0) its column was never rotated; the routine reports
// that through *needs_completion and the caller falls back to the re-orthonormalising variant below.
static int svd_split_qr2_direct(const SvdSplitDesc& d, const SvdWorkspace& w, const QrWorkspace& q, hipStream_t s, int* sweeps_out,
                                bool* needs_completion, bool complete_here) {
  const int N = d.m > d.n ? d.m : d.n;  // a rectangular theta (chain positions where the two outer bonds differ) is embedded in N x N
  const int cm = d.capM;
  const QrWorkspace q2 = q.second();
  int rc;
  if ((rc = qr_prepare(d.theta, d.theta_b0, d.m, d.n, d.distribution, d.d, q, d.nb0, d.ids, s, d.m == d.n ? 0 : N)) != TJM_OK) return rc;
  if ((rc = qr_factor(q, N, N, d.nb0, d.ids, s)) != TJM_OK) return rc;
  if ((rc = qr_adjoint_triangle(q, N, d.nb0, d.ids, s)) != TJM_OK) return rc;
  if ((rc = qr_factor(q2, N, N, d.nb0, d.ids, s)) != TJM_OK) return rc;
  JacobiSource src;  // X = R1^H
  src.src = q2.Z; src.src_b0 = q2.z_b0; src.rx = N; src.ncols = N; src.conj = 1; src.tri = 1;
  src.r_n0 = N; src.s_r1 = 0; src.s_r0 = N; src.c_n0 = N; src.s_c1 = 0; src.s_c0 = 1;
  src.nb0 = d.nb0; src.ids = d.ids;
  TruncSpec tr;
  tr.trunc_mode = d.trunc_mode; tr.threshold = d.threshold; tr.max_bond = d.max_bond; tr.min_keep = d.min_keep;
  tr.cap = d.capM; tr.overflow = d.overflow;
  tr.chiA = d.chiL; tr.mulA = d.d; tr.chiB = d.chiR; tr.mulB = d.d; tr.chiOut = d.chiM; tr.chi_stride = d.chi_stride;
  tr.spectrum = d.spectrum; tr.spec_ld = d.spec_ld;
  JacobiShape sh;
  if ((rc = jacobi_solve(src, tr, w, s, &sh, sweeps_out, false)) != TJM_OK) return rc;
  TJM_HIP_CHECK(hipMemcpyAsync(w.h_pinned, w.n_active + 2, sizeof(int), hipMemcpyDeviceToHost, s));
  TJM_HIP_CHECK(hipStreamSynchronize(s));
  *needs_completion = (*w.h_pinned != 0);
  if (*needs_completion && !complete_here) return TJM_OK;
  ExtractDesc xy;  // Ytilde: normalised kept columns of Y into Z (N x capM, column-major)
  xy.out = q.Z; xy.out_b0 = q.z_b0; xy.n_k = cm; xy.o_k = N; xy.n_r1 = 1; xy.n_r0 = N;
  xy.o_r1 = 0; xy.o_r0 = 1; xy.row_off = 0; xy.conj = 0; xy.scale_mode = 2;
  if (*needs_completion) {
    // A kept singular value at the rounding floor (rank-deficient theta with min_keep, threshold 0): its column of Ytilde is
    // noise.  The thin Householder Q factor of Ytilde keeps the orthonormal columns (up to a unit phase, which the projection
    // below absorbs) and completes the others to an orthonormal set; the second factorisation's reflectors are free by now.
    xy.out = q2.Z; xy.out_b0 = q2.z_b0;
    if ((rc = svd_extract(xy, w, sh, d.chiM, d.chi_stride, d.nb0, d.ids, s)) != TJM_OK) return rc;
    if ((rc = qr_factor(q2, N, cm, d.nb0, d.ids, s)) != TJM_OK) return rc;
    if ((rc = qr_identity(q.Z, q.z_b0, N, cm, d.nb0, s, d.ids, d.chiM, d.chi_stride)) != TJM_OK) return rc;
    if ((rc = qr_apply_q(q2, N, cm, q.Z, q.z_b0, cm, d.nb0, d.ids, s)) != TJM_OK) return rc;
    *needs_completion = false;
  } else if ((rc = svd_extract(xy, w, sh, d.chiM, d.chi_stride, d.nb0, d.ids, s)) != TJM_OK) return rc;
  if ((rc = qr_apply_q(q, N, N, q.Z, q.z_b0, cm, d.nb0, d.ids, s)) != TJM_OK) return rc;  // iso = Q Ytilde, rows bond-major
  ExtractDesc xi;
  GemmDesc g;
  memset(&g, 0, sizeof(g));
  g.nb0 = d.nb0; g.nb2 = 1; g.nks = d.d; g.ids = d.ids;
  if (d.distribution == 0) {
    // left[(s,a)][k] = iso[a*d+s][k] ; right[t][k][c] = sum_{(a,s)} conj(iso[a*d+s][k]) theta[(s,a)][(t,c)]
    xi.out = d.left; xi.out_b0 = d.left_b0; xi.n_k = cm; xi.o_k = 1; xi.n_r1 = d.capL; xi.n_r0 = d.d; xi.o_r1 = cm;
    xi.o_r0 = (long)d.capL * cm; xi.row_off = 0; xi.conj = 0; xi.scale_mode = 0;
    g.nb1 = d.d;
    g.A = q.Z; g.a_rs = N; g.a_ks = 1; g.a_cs = d.d; g.a_b0 = q.z_b0; g.conjA = 1; g.M = cm; g.K = d.capL;
    g.B = d.theta; g.b_ks = (long)d.capL * d.ld_theta; g.b_rs = d.ld_theta; g.b_cs = 1; g.b_b0 = d.theta_b0; g.b_b1 = d.capR; g.N = d.capR;
    g.C = d.right; g.c_rs = d.capR; g.c_b0 = d.right_b0; g.c_b1 = (long)cm * d.capR;
  } else {
    // right[t][k][c] = conj(iso[c*d+t][k]) ; left[(s,a)][k] = sum_{(t,c)} theta[(s,a)][(t,c)] iso[c*d+t][k]
    xi.out = d.right; xi.out_b0 = d.right_b0; xi.n_k = cm; xi.o_k = d.capR; xi.n_r1 = d.capR; xi.n_r0 = d.d;
    xi.o_r1 = 1; xi.o_r0 = (long)cm * d.capR; xi.row_off = 0; xi.conj = 1; xi.scale_mode = 0;
    g.nb1 = 1;
    g.A = d.theta; g.a_rs = d.ld_theta; g.a_ks = d.capR; g.a_cs = 1; g.a_b0 = d.theta_b0; g.M = d.m; g.K = d.capR;
    g.B = q.Z; g.b_ks = 1; g.b_rs = d.d; g.b_cs = N; g.b_b0 = q.z_b0; g.N = cm;
    g.C = d.left; g.c_rs = cm; g.c_b0 = d.left_b0;
  }
  if ((rc = launch_gemm(g, s)) != TJM_OK) return rc;  // reads theta and iso before the scatter below overwrites nothing they need
  return qr_scatter(q.Z, q.z_b0, N, xi, d.chiM, d.chi_stride, d.nb0, d.ids, s);
}

static int svd_split_qr2(const SvdSplitDesc& d, const SvdWorkspace& w, const QrWorkspace& q, hipStream_t s, int* sweeps_out) {
  const int N = d.m;
  static const bool reorth = getenv("TJM_REORTH_SPLIT") != nullptr;
  const bool only_direct = d.m != d.n || d.ids != nullptr;  // the variants below are written for square matrices of whole batches
  if ((!reorth && getenv("TJM_ACCUMULATE_W") == nullptr && d.capM <= N) || only_direct) {
    bool needs_completion = false;
    const int rc0 = svd_split_qr2_direct(d, w, q, s, sweeps_out, &needs_completion, only_direct);
    if (rc0 != TJM_OK || !needs_completion) return rc0;
  }
  const int fdist = 1 - d.distribution;
  const QrWorkspace q2 = q.second();
  int rc;
  if ((rc = qr_prepare(d.theta, d.theta_b0, d.m, d.n, fdist, d.d, q, d.nb0, d.ids, s)) != TJM_OK) return rc;
  if ((rc = qr_factor(q, N, N, d.nb0, d.ids, s)) != TJM_OK) return rc;
  if ((rc = qr_adjoint_triangle(q, N, d.nb0, d.ids, s)) != TJM_OK) return rc;
  if ((rc = qr_factor(q2, N, N, d.nb0, d.ids, s)) != TJM_OK) return rc;
  JacobiSource src;  // X = R1^H
  src.src = q2.Z; src.src_b0 = q2.z_b0; src.rx = N; src.ncols = N; src.conj = 1; src.tri = 1;
  src.r_n0 = N; src.s_r1 = 0; src.s_r0 = N; src.c_n0 = N; src.s_c1 = 0; src.s_c0 = 1;
  src.nb0 = d.nb0; src.ids = d.ids;
  TruncSpec tr;
  tr.trunc_mode = d.trunc_mode; tr.threshold = d.threshold; tr.max_bond = d.max_bond; tr.min_keep = d.min_keep;
  tr.cap = d.capM; tr.overflow = d.overflow;
  tr.chiA = d.chiL; tr.mulA = d.d; tr.chiB = d.chiR; tr.mulB = d.d; tr.chiOut = d.chiM; tr.chi_stride = d.chi_stride;
  tr.spectrum = d.spectrum; tr.spec_ld = d.spec_ld;
  JacobiShape sh;
  static const bool accumulate_w = getenv("TJM_ACCUMULATE_W") != nullptr;
  if (!accumulate_w && d.capM <= N && (long)N * d.capM + (long)d.capM * d.capM <= w.y_b0) {
    // ---- accumulation-free variant: rotate X only (half the Jacobi traffic, no replay kernel).  With Y = X W the left
    // singular vectors of Z are Uhat = Q Y Sigma^-1, and the isometric factor follows from the input itself,
    //   G = op(theta) Uhat Sigma^-1   (op = identity for dist 0, adjoint for dist 1),
    // whose columns are orthonormal up to eps * sigma_max / sigma_k.  A Householder QR G = Qu Ru restores an exactly
    // isometric factor Qu, and Ru (identity up to that error) goes into the weighted factor, so the product is unchanged:
    //   dist 0: theta ~ Qu (Ru Sigma Uhat^H),   dist 1: theta ~ (Uhat Sigma Ru^H) Qu^H.
    if ((rc = jacobi_solve(src, tr, w, s, &sh, sweeps_out, false)) != TJM_OK) return rc;
    const int cm = d.capM;
    ExtractDesc xy;  // Uhat-to-be: normalised kept columns of Y into Z (N x capM, column-major)
    xy.out = q.Z; xy.out_b0 = q.z_b0; xy.n_k = cm; xy.o_k = N; xy.n_r1 = 1; xy.n_r0 = N;
    xy.o_r1 = 0; xy.o_r0 = 1; xy.row_off = 0; xy.conj = 0; xy.scale_mode = 2;
    if ((rc = svd_extract(xy, w, sh, d.chiM, d.chi_stride, d.nb0, d.ids, s)) != TJM_OK) return rc;
    if ((rc = qr_apply_q(q, N, N, q.Z, q.z_b0, cm, d.nb0, d.ids, s)) != TJM_OK) return rc;  // Uhat, rows bond-major
    cplx* G = w.Y;                  // [N][capM] row-major, natural row order; Y itself is no longer needed
    cplx* Rs = w.Y + (long)N * cm;  // [capM][capM] row-major
    {
      GemmDesc g;
      memset(&g, 0, sizeof(g));
      g.nb0 = d.nb0; g.nb1 = 1; g.nb2 = 1;
      g.B = q.Z; g.C = G;
      g.N = cm; g.b_cs = N; g.c_rs = cm;
      g.b_b0 = q.z_b0; g.c_b0 = w.y_b0; g.a_b0 = d.theta_b0;
      g.A = d.theta;
      if (d.distribution == 0) {
        // G[(s,a)][k] = sum_{(c,t)} theta[(s,a)][(t,c)] Uhat[c*d+t][k]
        g.M = d.m; g.a_rs = d.ld_theta;
        g.nks = d.d; g.K = d.capR; g.a_ks = d.capR; g.a_cs = 1; g.b_ks = 1; g.b_rs = d.d;
      } else {
        // G[(t,c)][k] = sum_{(a,s)} conj(theta[(s,a)][(t,c)]) Uhat[a*d+s][k]
        g.M = d.n; g.a_rs = 1; g.conjA = 1;
        g.nks = d.d; g.K = d.capL; g.a_ks = (long)d.capL * d.ld_theta; g.a_cs = d.ld_theta; g.b_ks = 1; g.b_rs = d.d;
      }
      if ((rc = launch_gemm(g, s)) != TJM_OK) return rc;
    }
    if ((rc = qr_gather_scaled(G, w.y_b0, N, cm, d.d, w.norms, sh.ncols_pad, d.chiM, d.chi_stride, q2.Z, q2.z_b0, d.nb0, s)) != TJM_OK) return rc;
    if ((rc = qr_factor(q2, N, cm, d.nb0, d.ids, s)) != TJM_OK) return rc;
    if ((rc = qr_r_times_sigma(q2.Z, q2.z_b0, N, cm, w.norms, sh.ncols_pad, d.chiM, d.chi_stride, Rs, w.y_b0, d.nb0, s)) != TJM_OK) return rc;
    if ((rc = qr_identity(G, w.y_b0, N, cm, d.nb0, s)) != TJM_OK) return rc;                       // thin Qu = Q2 [I; 0]
    if ((rc = qr_apply_q(q2, N, cm, G, w.y_b0, cm, d.nb0, d.ids, s)) != TJM_OK) return rc;
    ExtractDesc xi;
    GemmDesc g;
    memset(&g, 0, sizeof(g));
    g.nks = 1; g.nb0 = d.nb0; g.nb1 = d.d; g.nb2 = 1; g.K = cm;
    if (d.distribution == 0) {
      // left[(s,a)][k] = Qu[a*d+s][k] ; right[t][k][c] = sum_j Rs[k][j] conj(Uhat[c*d+t][j])
      xi.out = d.left; xi.out_b0 = d.left_b0; xi.n_k = cm; xi.o_k = 1; xi.n_r1 = d.capL; xi.n_r0 = d.d; xi.o_r1 = cm;
      xi.o_r0 = (long)d.capL * cm; xi.row_off = 0; xi.conj = 0; xi.scale_mode = 0;
      g.A = Rs; g.a_rs = cm; g.a_cs = 1; g.a_b0 = w.y_b0; g.M = cm;
      g.B = q.Z; g.b_rs = N; g.b_cs = d.d; g.b_b0 = q.z_b0; g.b_b1 = 1; g.conjB = 1; g.N = d.capR;
      g.C = d.right; g.c_rs = d.capR; g.c_b0 = d.right_b0; g.c_b1 = (long)cm * d.capR;
    } else {
      // right[t][k][c] = conj(Qu[c*d+t][k]) ; left[s][a][k] = sum_j Uhat[a*d+s][j] conj(Rs[k][j])
      xi.out = d.right; xi.out_b0 = d.right_b0; xi.n_k = cm; xi.o_k = d.capR; xi.n_r1 = d.capR; xi.n_r0 = d.d;
      xi.o_r1 = 1; xi.o_r0 = (long)cm * d.capR; xi.row_off = 0; xi.conj = 1; xi.scale_mode = 0;
      g.A = q.Z; g.a_rs = d.d; g.a_cs = N; g.a_b0 = q.z_b0; g.a_b1 = 1; g.M = d.capL;
      g.B = Rs; g.b_rs = 1; g.b_cs = cm; g.b_b0 = w.y_b0; g.conjB = 1; g.N = cm;
      g.C = d.left; g.c_rs = cm; g.c_b0 = d.left_b0; g.c_b1 = (long)d.capL * cm;
    }
    if ((rc = qr_scatter(G, w.y_b0, N, xi, d.chiM, d.chi_stride, d.nb0, d.ids, s)) != TJM_OK) return rc;
    return launch_gemm(g, s);
  }
  if ((rc = jacobi_solve(src, tr, w, s, &sh, sweeps_out)) != TJM_OK) return rc;
  // isometric factor Q1 W
  TJM_HIP_CHECK(hipMemsetAsync(q2.Z, 0, (size_t)q2.z_b0 * sizeof(cplx) * (size_t)d.nb0, s));
  ExtractDesc xw;
  xw.out = q2.Z; xw.out_b0 = q2.z_b0; xw.n_k = d.capM; xw.o_k = N; xw.n_r1 = 1; xw.n_r0 = (N < sh.ncols_pad) ? N : sh.ncols_pad;
  xw.o_r1 = 0; xw.o_r0 = 1; xw.row_off = sh.rx_top; xw.conj = 0; xw.scale_mode = 0;
  if ((rc = svd_extract(xw, w, sh, d.chiM, d.chi_stride, d.nb0, d.ids, s)) != TJM_OK) return rc;
  if ((rc = qr_apply_q(q2, N, N, q2.Z, q2.z_b0, d.capM, d.nb0, d.ids, s)) != TJM_OK) return rc;
  // weighted factor Q Y (Y = rotated X rows)
  ExtractDesc xy;
  xy.out = q.Z; xy.out_b0 = q.z_b0; xy.n_k = d.capM; xy.o_k = N; xy.n_r1 = 1; xy.n_r0 = N;
  xy.o_r1 = 0; xy.o_r0 = 1; xy.row_off = 0; xy.conj = 0; xy.scale_mode = 0;
  if ((rc = svd_extract(xy, w, sh, d.chiM, d.chi_stride, d.nb0, d.ids, s)) != TJM_OK) return rc;
  if ((rc = qr_apply_q(q, N, N, q.Z, q.z_b0, d.capM, d.nb0, d.ids, s)) != TJM_OK) return rc;
  ExtractDesc xi, xx;  // xi: isometric (rows = sorted columns of Z, natural index through the permutation), xx: weighted (bond-major rows)
  if (d.distribution == 0) {
    // Z = theta^H: left[(s,a)][k] = (Q1 W)[j][k] with (s,a) = colperm[j] ; right[t][k][c] = conj((Q Y)[(c,t)][k])
    xi.out = d.left; xi.out_b0 = d.left_b0; xi.n_k = d.capM; xi.o_k = 1; xi.n_r1 = 1; xi.n_r0 = N; xi.o_r1 = 0; xi.o_r0 = d.capM;
    xi.row_off = 0; xi.conj = 0; xi.scale_mode = 0;
    xx.out = d.right; xx.out_b0 = d.right_b0; xx.n_k = d.capM; xx.o_k = d.capR; xx.n_r1 = d.capR; xx.n_r0 = d.d;
    xx.o_r1 = 1; xx.o_r0 = (long)d.capM * d.capR; xx.row_off = 0; xx.conj = 1; xx.scale_mode = 0;
  } else {
    // Z = theta: right[t][k][c] = conj((Q1 W)[j][k]) with (t,c) = colperm[j] ; left[(s,a)][k] = (Q Y)[(a,s)][k]
    xi.out = d.right; xi.out_b0 = d.right_b0; xi.n_k = d.capM; xi.o_k = d.capR; xi.n_r1 = d.d; xi.n_r0 = d.capR;
    xi.o_r1 = (long)d.capM * d.capR; xi.o_r0 = 1; xi.row_off = 0; xi.conj = 1; xi.scale_mode = 0;
    xx.out = d.left; xx.out_b0 = d.left_b0; xx.n_k = d.capM; xx.o_k = 1; xx.n_r1 = d.capL; xx.n_r0 = d.d; xx.o_r1 = d.capM;
    xx.o_r0 = (long)d.capL * d.capM; xx.row_off = 0; xx.conj = 0; xx.scale_mode = 0;
  }
  xi.row_map = q.colperm();
  xi.row_map_ld = q.w_ld;
  if ((rc = qr_scatter(q2.Z, q2.z_b0, N, xi, d.chiM, d.chi_stride, d.nb0, d.ids, s)) != TJM_OK) return rc;
  return qr_scatter(q.Z, q.z_b0, N, xx, d.chiM, d.chi_stride, d.nb0, d.ids, s);
}

#ifndef TJM_F32
// ---- mixed-precision two-site split (fp64 library only; tjm_mixed.h) -----------------------------------------------------------
// Z = theta (dist 0) / theta^H (dist 1) is the matrix whose LEFT singular basis is the isometric factor (the orientation of the
// direct variant above); N = its size, cm = the storage of the new bond.
//  1. complex64: all right singular vectors of Z, approximately - the left singular basis of Z^H by the doubly preconditioned
//     complex64 Jacobi (the pairwise, sequential part of the work at the packed-fp32 rate) - V0, orthonormal to ~2e-6.
//  2. fp64, matrix cores only.  A polar step makes the basis unitary, V = V0 (I + E)^(-1/2) ~ V0 (I - E/2 + 3 E^2/8), E = V0^H V0 - I,
//     certified per trajectory through ||E^2||_F.  Then X = Z V has columns orthogonal to ~1e-6, and the Gram matrix G = X^H X says
//     what is left to do: to first order the unitary that diagonalises G is I + C with C_ij = G_ij / (G_jj - G_ii) (anti-Hermitian),
//     so V <- V (I + C + C^2/2) squares the off-diagonal part (1e-6 -> 1e-10 -> rounding: quadratic convergence, two rounds, four
//     GEMMs each).  This is a whole Jacobi sweep done as matrix products - possible because every rotation angle is tiny; a pair
//     whose angle is not (near-degenerate singular values with a sizeable coupling) gets a clamped correction and, if the final
//     check still finds it, the trajectory goes to the fp64 Jacobi kernels for the rest (from X, which is nearly diagonal by then).
//     A last polar step removes what the truncated exponentials left of non-unitarity, X = Z V is formed once more from theta
//     itself and its Gram matrix is the final check: |G_ij| <= 1e-13 sigma_i sigma_j + 1e-14 ||theta|| max(sigma_i, sigma_j) for
//     every pair that matters - the second term because X is computed with ABSOLUTE error ~1e-15 ||theta||, so the direction of a
//     column of norm sigma carries 1e-15 / sigma of noise that no iteration on V can remove.
//  3. Pairs that do not matter.  The truncation keeps at most cm singular triplets.  Columns beyond cm + 32 (in the complex64
//     order) only have to stay orthogonal to the first cm + 32; among themselves they are never corrected (C_ij = 0): their squared
//     norms still add up to the discarded weight exactly, and a scaled Gershgorin bound on their Gram block, read from the same G,
//     certifies per trajectory that none of their singular values reaches the cm-th largest of the others.  This is where the
//     complex64 basis is worst (directions below 1e-6 of the largest singular value are noise in fp32), so skipping them also skips
//     the slowest part of the convergence.  Only for the discarded-weight rules and when no spectrum is asked for.
//  4. Isometric factor = the kept columns of X, normalised and then made exactly isometric by the same polar step (their
//     orthogonality is 1e-15 ||theta|| / sigma_k, the absolute error above; the span does not change), weighted factor = its
//     projection of theta (one GEMM), as in the direct variant.  The product is the projection of theta on the span of the kept
//     columns of theta V with V unitary to rounding: nothing of the complex64 phase survives except the choice of the starting basis.
// Whatever does not fit (polar certificate, a kept singular value at the noise floor) sends the whole batch to the all-fp64 path;
// single trajectories that fail the final check or the Gershgorin certificate are finished by the fp64 Jacobi kernels.
namespace {

struct float2_t { float x, y; };

// scale[b] = the power of two that brings ||theta_b||_F to about 2^24 ; fro2[b] = ||theta_b||_F^2.  The complex64 rotation set-up
// squares squared column norms (make_rotation): with ||X||_F ~ 1 a column at 1e-8 of the largest one underflows there and is never
// rotated; scaled like this, columns from 1e7 down to the fp32 rounding floor of the matrix (and its own smallest singular values)
// stay inside the range.  Singular vectors do not depend on the scale, and a power of two changes no mantissa.
__global__ __launch_bounds__(256) void c64_scale_kernel(const cplx* __restrict__ in, long in_b0, long n, real* __restrict__ scale, real* __restrict__ fro2) {
  __shared__ real sh[4];
  const cplx* ib = in + (long)blockIdx.x * in_b0;
  real acc = 0.0;
  for (long e = threadIdx.x; e < n; e += blockDim.x) {
    const cplx v = ib[e];
    acc = fma(v.x, v.x, fma(v.y, v.y, acc));
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const real f2 = sh[0] + sh[1] + sh[2] + sh[3];
    int e = 0;
    if (f2 > 0.0 && f2 == f2 && f2 < 1e300) frexp(f2, &e);  // f2 = m 2^e, 0.5 <= m < 1
    scale[blockIdx.x] = ldexp(1.0, 24 - e / 2);
    fro2[blockIdx.x] = f2;
  }
}

__global__ __launch_bounds__(256) void to_c64_kernel(const cplx* __restrict__ in, long in_b0, float2_t* __restrict__ out, long out_b0, long n,
                                                    const real* __restrict__ scale) {
  const cplx* ib = in + (long)blockIdx.y * in_b0;
  float2_t* ob = out + (long)blockIdx.y * out_b0;
  const real sc = scale[blockIdx.y];
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const cplx v = ib[e];
    ob[e] = float2_t{(float)(v.x * sc), (float)(v.y * sc)};
  }
}

// The complex64 basis (N x N column-major, rows bond-major: bond * d + phys, the row order of the preconditioner) into fp64 with its
// rows in the NATURAL order of theta's index (phys * cap + bond): X = Z V is then a plain K = N product with contiguous runs of V.
// (Everything done to the basis afterwards acts on its columns, so the row order is free.)
__global__ __launch_bounds__(256) void from_c64_kernel(const float2_t* __restrict__ in, long in_b0, cplx* __restrict__ out, long out_b0, int N, int d) {
  const float2_t* ib = in + (long)blockIdx.y * in_b0;
  cplx* ob = out + (long)blockIdx.y * out_b0;
  const long n = (long)N * N;
  const int cap = N / d;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e / N), q = (int)(e % N);  // destination row q = phys * cap + bond
    const int ph = q / cap, bond = q % cap;
    const float2_t v = ib[(long)k * N + (long)bond * d + ph];
    ob[e] = cplx{(real)v.x, (real)v.y};
  }
}

// E <- E - diag(1 for i < keep) (row-major n x n; keep == nullptr: the whole diagonal)
__global__ __launch_bounds__(256) void polar_residual_kernel(cplx* __restrict__ E, long e_b0, int n, const int* __restrict__ keep, int keep_stride,
                                                            const int* __restrict__ ids) {
  const int b = ids ? ids[blockIdx.y] : blockIdx.y;
  cplx* Eb = E + (long)b * e_b0;
  const int kp = keep ? keep[(long)b * keep_stride] : n;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < kp && i < n; i += gridDim.x * blockDim.x) Eb[(long)i * n + i].x -= 1.0;
}

// T = I - E / 2 + c2 E2 (c2 = 3/8: the series to second order; 0: first order) ; e2fro2[b] += ||E2||_F^2
// (E Hermitian: ||E||_2^2 = ||E^2||_2 <= ||E^2||_F, the certificate below)
__global__ __launch_bounds__(256) void polar_poly_kernel(const cplx* __restrict__ E, long e_b0, const cplx* __restrict__ E2, long e2_b0,
                                                        cplx* __restrict__ T, long t_b0, int N, real c2, real* __restrict__ e2fro2,
                                                        const int* __restrict__ ids) {
  __shared__ real sh[4];
  const int b = ids ? ids[blockIdx.y] : blockIdx.y;
  const cplx* Eb = E + (long)b * e_b0;
  const cplx* Fb = E2 + (long)b * e2_b0;
  cplx* Tb = T + (long)b * t_b0;
  real acc = 0.0;
  for (int i = blockIdx.x; i < N; i += gridDim.x)  // rows to the workgroups, a row's entries to the threads: no 64-bit index divisions
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
      const long e = (long)i * N + j;
      const cplx a = Eb[e], b = Fb[e];
      cplx v{fma(c2, b.x, -0.5 * a.x), fma(c2, b.y, -0.5 * a.y)};
      if (i == j) v.x += 1.0;
      Tb[e] = v;
      acc = fma(b.x, b.x, fma(b.y, b.y, acc));
    }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0 && e2fro2) atomicAdd(&e2fro2[b], sh[0] + sh[1] + sh[2] + sh[3]);
}

// need[b] = 1 and flag |= 1 when a trajectory's certificate fails: the polar series I - E/2 + 3 E^2/8 leaves
// (5/16) ||E||_2^3 <= (5/16) ||E^2||_F^(3/2)
__global__ void polar_check_kernel(const real* e2fro2, int nb0, real tol2, int* flag, int* need) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb0) return;
  const int bad = !(e2fro2[b] <= tol2) ? 1 : 0;
  need[b] = bad;
  if (bad) atomicOr(flag, 1);
}

__global__ void and_flags_kernel(int* a, const int* b, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = (a[i] != 0 && b[i] != 0) ? 1 : 0;
}

// dst[b] = src[b] for the listed trajectories (the ones that took the second polar step: the others keep the basis they had)
__global__ __launch_bounds__(256) void listed_copy_kernel(cplx* __restrict__ dst, const cplx* __restrict__ src, long b0, long n, const int* __restrict__ ids) {
  const int b = ids[blockIdx.y];
  const cplx* sb = src + (long)b * b0;
  cplx* db = dst + (long)b * b0;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) db[e] = sb[e];
}

// Does the pair (i, j) of the Gram matrix still need work?  d_i, d_j its squared column norms, f2 = |G_ij|^2, tr = ||theta||_F^2.
__device__ inline bool pair_open(real di, real dj, real f2, real tr) {
  const real floor2 = TJM_NOISE_FLOOR2 * tr;
  if (!(di > floor2 && dj > floor2)) return false;                       // a numerically null column: nothing to orthogonalise
  const real dmax = di > dj ? di : dj;
  return f2 > TJM_JACOBI_TOL2 * di * dj + real(1e-28) * tr * dmax;        // (1e-13 sigma_i sigma_j)^2 + (1e-14 ||theta|| sigma_max)^2
}

// Columns that are not corrected among themselves (point 3): position >= cm always (never kept), and either beyond skip_col or at the
// rounding floor of the complex64 arithmetic (sigma < 1e-5 ||theta||: their directions are noise in fp32 wherever they sit).
__device__ inline bool far_column(int j, real dj, int cm, int skip_col, real tr) { return j >= cm && (j >= skip_col || dj < real(1e-10) * tr); }

// First-order correction of the basis from the Gram matrix G = X^H X (row-major N x N): C_ij = G_ij / (G_jj - G_ii) for the open pairs
// that are not both "far" columns and whose angle |C_ij| is at most zmax; 0 elsewhere.  C is anti-Hermitian by construction.
//
// With E = V^H V - I given (the LAST round): the basis is made unitary in the same step.  To first order V (I - E/2) is unitary and
// turns G into G - (E D + D E) / 2 (D = diag G), so the correction is taken from G'_ij = G_ij - E_ij (d_i + d_j) / 2 with
// d'_i = d_i (1 - E_ii), and the kernel writes W = C - E/2 (diagonal included) instead of C; the caller applies I + W + W^2/2.  zmax is
// 1e-4 in that round: |C|^3 / 6 <= 2e-13 is what the truncated exponential then leaves of non-unitarity, with nothing behind it to
// repair more; a pair that still wants a larger angle after a round of quadratic convergence is not converging (it is left alone and
// shows up in the final check if it matters).
__global__ __launch_bounds__(256) void refine_corr_kernel(const cplx* __restrict__ G, long g_b0, int N, int skip_col, int cm, const real* __restrict__ fro2,
                                                         real zmax, cplx* __restrict__ Cm, long c_b0, const cplx* __restrict__ E, long e_b0,
                                                         float2_t* __restrict__ C32, long c32_b0) {
  const cplx* Gb = G + (long)blockIdx.y * g_b0;
  float2_t* Cs = C32 ? C32 + (long)blockIdx.y * c32_b0 : nullptr;  // a complex64 copy for the square (tjm32::mixed_square)
  const cplx* Eb = E ? E + (long)blockIdx.y * e_b0 : nullptr;
  cplx* Cb = Cm + (long)blockIdx.y * c_b0;
  const real tr = fro2[blockIdx.y];
  // rows dealt to the workgroups, a row's entries to the threads (no 64-bit index divisions; the diagonal - every entry needs d_i and
  // d_j - staged once per workgroup: the element-per-thread form of round 4 ran at 2.2 TB/s)
  __shared__ real sD[1024], sDe[1024];
  for (int k = threadIdx.x; k < N; k += blockDim.x) {
    sD[k] = Gb[(long)k * N + k].x;
    sDe[k] = Eb ? Eb[(long)k * N + k].x : real(0.0);
  }
  __syncthreads();
  for (int i = blockIdx.x; i < N; i += gridDim.x)
  for (int j = threadIdx.x; j < N; j += blockDim.x) {
    const long e = (long)i * N + j;
    cplx c{0.0, 0.0};
    if (i == j) {
      if (Eb) c = cplx{-0.5 * Eb[e].x, 0.0};
    } else {
      real di = sD[i], dj = sD[j];
      cplx f = Gb[e];
      if (Eb) {
        const cplx ee = Eb[e];
        const real h = 0.5 * (di + dj);
        f = cplx{fma(-h, ee.x, f.x), fma(-h, ee.y, f.y)};
        di *= 1.0 - sDe[i];
        dj *= 1.0 - sDe[j];
      }
      const real f2 = fma(f.x, f.x, f.y * f.y);
      if (!(far_column(i, di, cm, skip_col, tr) && far_column(j, dj, cm, skip_col, tr)) && pair_open(di, dj, f2, tr)) {
        const real gap = dj - di;
        const real ag = fabs(gap), af = sqrt(f2);
        // a (near-)degenerate pair whose angle is not small is left alone: one bounded step would spoil the quadratic convergence
        // of every pair that shares a column with it (cross terms zmax x |C|), and inside K or inside the complement the mixing is
        // harmless; if it straddles the truncation the final check sends the trajectory to the Jacobi kernels
        if (af <= zmax * ag) { const real inv = 1.0 / gap; c = cplx{f.x * inv, f.y * inv}; }
      }
      if (Eb) { const cplx ee = Eb[e]; c.x = fma(-0.5, ee.x, c.x); c.y = fma(-0.5, ee.y, c.y); }
    }
    Cb[e] = c;
    if (Cs) Cs[e] = float2_t{(float)c.x, (float)c.y};
  }
}

// T = I + C + C2 / 2   (exp(C) to second order: unitary up to |C|^3 / 6, which the last polar step removes)
__global__ __launch_bounds__(256) void refine_poly_kernel(const cplx* __restrict__ Cm, long c_b0, const cplx* __restrict__ C2, long c2_b0,
                                                         cplx* __restrict__ T, long t_b0, int N, const float2_t* __restrict__ C2s, long c2s_b0) {
  const cplx* Cb = Cm + (long)blockIdx.y * c_b0;
  const cplx* Db = C2 + (long)blockIdx.y * c2_b0;
  const float2_t* Ds = C2s ? C2s + (long)blockIdx.y * c2s_b0 : nullptr;  // the square from the complex64 GEMM instead
  cplx* Tb = T + (long)blockIdx.y * t_b0;
  for (int i = blockIdx.x; i < N; i += gridDim.x)
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
      const long e = (long)i * N + j;
      const cplx a = Cb[e];
      cplx b;
      if (Ds) { const float2_t f = Ds[e]; b = cplx{(real)f.x, (real)f.y}; }
      else b = Db[e];
      cplx v{fma(0.5, b.x, a.x), fma(0.5, b.y, a.y)};
      if (i == j) v.x += 1.0;
      Tb[e] = v;
    }
}

// Final check on G = X^H X, one workgroup per trajectory, after svd_finish_kernel has ranked the columns by norm and applied the
// truncation rule (perm, keep).  K = the kept columns.  What the result needs:
//   bit 0  every kept column is a singular direction with respect to everything that is not kept: the tilt |G_ij| / |d_i - d_j| of a
//          pair (i in K, j not in K) is at most 1e-13 (plus the rounding floor of X, 1e-14 ||theta|| sigma_i / d_i);
//   bit 1  nothing outside K has a singular value above the smallest kept one: with D = diag(sigma) the matrix D^-1 G_cc D of the
//          complement has the eigenvalues of G_cc and row sums d_i + sum_j |G_ij| sigma_j / sigma_i, so the largest of those bounds
//          its largest squared singular value (scaled Gershgorin) - it has to stay below the smallest kept squared norm;
//   bit 2  the kept columns are orthogonal to 1e-5 among themselves (their polar step takes it from there; a near-degenerate pair
//          that the correction left alone sits at the 2e-6 of the complex64 basis).
// Pairs inside K or inside the complement do not have to be diagonal: mixing them changes neither span.
// strict (a spectrum is asked for, or a truncation rule that reads single discarded values): every pair has to be diagonal to the
// Jacobi tolerance, because every column norm is then reported as a singular value.
__global__ __launch_bounds__(256) void refine_check_kernel(const cplx* __restrict__ G, long g_b0, int N, const real* __restrict__ fro2,
                                                          const int* __restrict__ perm_all, const int* __restrict__ keep_all, int keep_stride,
                                                          int strict, const int* __restrict__ skip, int* __restrict__ status, int* __restrict__ n_bad) {
  __shared__ real sd[1024];
  __shared__ unsigned char inK[1024];
  __shared__ int s_flags;
  __shared__ real s_bound[4];
  __shared__ real s_min;
  const int b = blockIdx.x, tid = threadIdx.x;
  const cplx* Gb = G + (long)b * g_b0;
  const int* perm = perm_all + (long)b * N;
  const int keep = keep_all[(long)b * keep_stride];
  const real tr = fro2[b];
  for (int i = tid; i < N; i += 256) { sd[i] = Gb[(long)i * N + i].x; inK[i] = 0; }
  if (tid == 0) s_flags = 0;
  __syncthreads();
  for (int k = tid; k < keep; k += 256) inK[perm[k]] = 1;
  if (tid == 0) s_min = keep > 0 ? sd[perm[keep - 1]] : real(0.0);
  __syncthreads();
  int flags = 0;
  const long total = (long)N * N;
  for (long e = tid; e < total; e += 256) {
    const int i = (int)(e / N), j = (int)(e % N);
    if (i >= j) continue;
    const cplx f = Gb[e];
    const real f2 = fma(f.x, f.x, f.y * f.y);
    const real di = sd[i], dj = sd[j];
    if (strict) {
      if (pair_open(di, dj, f2, tr)) flags |= 1;
    } else if (inK[i] != inK[j]) {
      const real gap = di - dj, dmax = di > dj ? di : dj;
      if (f2 > real(1e-26) * gap * gap + real(1e-28) * tr * dmax) flags |= 1;
    } else if (inK[i]) {
      if (f2 > real(1e-10) * di * dj) flags |= 4;
    }
  }
  real bound = 0.0;
  for (int i = tid; i < N; i += 256) {
    if (inK[i]) continue;
    const real di = sd[i];
    real row = di;
    if (di > 0.0) {
      const real isi = 1.0 / sqrt(di);
      for (int j = 0; j < N; ++j)
        if (j != i && !inK[j]) { const cplx f = Gb[(long)i * N + j]; row = fma(sqrt(fma(f.x, f.x, f.y * f.y) * sd[j]), isi, row); }
    }
    bound = row > bound ? row : bound;
  }
  for (int off = 32; off > 0; off >>= 1) { const real o = __shfl_down(bound, off); bound = o > bound ? o : bound; }
  if ((tid & 63) == 0) s_bound[tid >> 6] = bound;
  if (flags) atomicOr(&s_flags, flags);
  __syncthreads();
  if (tid == 0) {
    int st = s_flags;
    real bmax = s_bound[0];
    for (int k = 1; k < 4; ++k) bmax = s_bound[k] > bmax ? s_bound[k] : bmax;
    if (!strict && keep > 0 && keep < N && !(bmax <= s_min)) st |= 2;
    if (skip[b]) st = 0;  // a trajectory whose basis could not be made unitary: the fp64 path serves it at the end
    status[b] = st;
    if (st) atomicAdd(n_bad, 1);
  }
}

// out[b] = 1 when the trajectory has to be served by the all-fp64 path: its polar certificate failed twice (gave_up), or a KEPT
// singular value sits at the rounding floor - its column was never rotated / corrected, the completing variant of the fp64 path
// handles it (the per-trajectory form of the flag svd_finish_kernel raises for the batch)
// clipped[b] != 0: the truncation of trajectory b was clipped by the storage of the new bond - raised on the engine's sticky flag
// only for a trajectory this path serves (one that goes to the fp64 path is truncated again there, from singular values that can
// be trusted: the column norms of a trajectory whose basis failed the polar certificate are not)
__global__ void mixed_fallback_kernel(const real* __restrict__ norms, int ncols_pad, const int* __restrict__ keep_all, int keep_stride,
                                      const int* __restrict__ gave_up, int nb0, int* __restrict__ out, int* __restrict__ count,
                                      const int* __restrict__ clipped, int* __restrict__ overflow) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb0) return;
  const int keep = keep_all[(long)b * keep_stride];
  const real* nr = norms + (long)b * ncols_pad;
  const int bad = (gave_up[b] != 0 || (keep > 0 && nr[keep - 1] <= TJM_RANK_TOL * nr[0])) ? 1 : 0;
  out[b] = bad;
  if (bad) atomicAdd(count, 1);
  else if (overflow && clipped[b]) atomicOr(overflow, 1);
}

struct MixedStats { long solves = 0, c64_sweeps = 0, f64_sweeps = 0, fallbacks = 0, jacobi_trajectories = 0, second_polar = 0, gemms = 0, c64_gemms = 0; double gemm_flops = 0.0; };
MixedStats g_mixed;

}  // namespace

void mixed_stats_get(double* out10, bool reset) {
  double w4[4];
  tjm32::jacobi_work_get(w4, reset);  // executed work of the complex64 Jacobi kernels: rotation slots x rows, applied rotations x rows
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  out10[0] = (double)g_mixed.solves; out10[1] = (double)g_mixed.c64_sweeps; out10[2] = (double)g_mixed.f64_sweeps;
  out10[3] = (double)g_mixed.fallbacks; out10[4] = (double)g_mixed.jacobi_trajectories; out10[5] = (double)g_mixed.second_polar;
  out10[6] = w4[0]; out10[7] = w4[1];
  out10[8] = (double)g_mixed.gemms; out10[9] = g_mixed.gemm_flops;
  if (reset) g_mixed = MixedStats();
}

void mixed_qr_profile_enable(int every) { tjm32::qr_profile_enable(every); }
void mixed_qr_profile_get(double* out5) { tjm32::qr_profile_get(out5); }
void mixed_profile_enable(int every) { tjm32::profile_enable(every); }
void mixed_profile_get(double* total_ms, double* total_bytes, long* samples) { tjm32::profile_get(total_ms, total_bytes, samples); }

// largest square split the mixed path serves: 1024 rows since round 6 - the complex64 instance's tile kernel keeps 1024-row columns in
// registers now, so it preconditions the splits of bonds up to 512 as well (one stream, 16 matrices of 1024 x 1024: 105 against 254 ms
// per batched split on the all-fp64 path; the complex64 basis is orthonormal to ~N eps32 there, so every batch takes the listed second
// polar step).  TJM_MIXED_MAX_DIM=512: as until round 5.
static int mixed_max_dim() {
  static const int v = getenv("TJM_MIXED_MAX_DIM") ? atoi(getenv("TJM_MIXED_MAX_DIM")) : 1024;
  return v < 128 ? 128 : (v > 1024 ? 1024 : v);
}

size_t mixed_split_workspace_bytes(int max_dim, int B) {
  static const bool off = getenv("TJM_NO_MIXED_SPLIT") != nullptr;
  if (off || max_dim < 128 || max_dim > mixed_max_dim() || max_dim % 64 != 0) return 0;
  // the complex64 phase, two more fp64 matrices per trajectory next to the four of the (idle) fp64 preconditioner, a status word
  return tjm32::mixed_workspace_bytes(max_dim, B) + 2 * (((size_t)B * max_dim * max_dim * sizeof(cplx) + 255) / 256 * 256) +
         4 * (((size_t)B * sizeof(int) + 255) / 256 * 256) + 1024;
}

bool mixed_split_fits(const SvdSplitDesc& d, const QrWorkspace& q, const MixedWorkspace* mx) {
  static const bool off = getenv("TJM_NO_MIXED_SPLIT") != nullptr;
  if (off || mx == nullptr || mx->base == nullptr || q.Z2 == nullptr || d.ids != nullptr) return false;
  if (d.distribution != 0 && d.distribution != 1) return false;
  if (d.m != d.n || d.ld_theta != d.n || d.m < 128 || d.m % 64 != 0 || d.m > mx->max_dim || d.nb0 > mx->B) return false;
  return d.capM <= d.m && (long)d.m * d.m <= q.z_b0 && (long)d.m * d.m <= q.v_b0;
}

// returns TJM_OK with *done = true when the outputs are written, *done = false when the batch has to take the fp64 path
// launch_gemm with the nominal work (8 M N K nks real flops per matrix and inner batch) added to the counters of the mixed split
static int mixed_gemm(const GemmDesc& g, hipStream_t s) {
  {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    ++g_mixed.gemms;
    const double tiles = (g.M + 63) / 64;  // a Hermitian result computes the tiles on and above the diagonal only
    g_mixed.gemm_flops += (g.hermitian ? (tiles + 1.0) / (2.0 * tiles) : 1.0) * 8.0 * g.M * g.N * (double)g.K * g.nks * g.nb0 * g.nb1 * g.nb2;
  }
  return launch_gemm(g, s);
}

static int svd_split_mixed(const SvdSplitDesc& d, const SvdWorkspace& w, const QrWorkspace& q, const MixedWorkspace& mx, hipStream_t s,
                           int* sweeps_out, bool* done) {
  *done = false;
  const int N = d.m, cm = d.capM, nb = d.nb0;
  const long nn = (long)N * N;
  int rc;
  int gx = (int)((nn + 1023) / 1024);
  if (gx > 128) gx = 128;
  // ---- workspace: [complex64 phase | fp64 matrix | fp64 matrix | status | id list | polar flags]
  const size_t c64_bytes = tjm32::mixed_workspace_bytes(mx.max_dim, mx.B);
  const size_t mat_bytes = ((size_t)mx.B * mx.max_dim * mx.max_dim * sizeof(cplx) + 255) / 256 * 256;
  char* tail = static_cast<char*>(mx.base) + (c64_bytes + 255) / 256 * 256;
  const size_t int_bytes = ((size_t)mx.B * sizeof(int) + 255) / 256 * 256;
  if ((size_t)(tail - static_cast<char*>(mx.base)) + 2 * mat_bytes + 4 * int_bytes > mx.bytes) return TJM_ERR_WORKSPACE;
  const long x_b0 = (long)mx.max_dim * mx.max_dim;
  cplx* S2 = reinterpret_cast<cplx*>(tail);                 // squares (E^2, C^2)
  cplx* Iso = reinterpret_cast<cplx*>(tail + mat_bytes);    // the kept columns before their polar step
  int* status = reinterpret_cast<int*>(tail + 2 * mat_bytes);
  int* idlist = status + int_bytes / sizeof(int);
  int* pneed = idlist + int_bytes / sizeof(int);  // per trajectory: the polar certificate failed (first step: wants a second; after it: gives up)
  int* clipped = pneed + int_bytes / sizeof(int);  // per trajectory: the truncation rule wanted more than the storage of the new bond holds
  // fp64 matrices in the buffers of the (idle) fp64 preconditioner
  cplx* Va = q.Z;  cplx* Vb = q.Z2;  const long v_b0 = q.z_b0;  // the basis, ping-pong (N x N column-major)
  cplx* Gm = q.V;  cplx* Cm = q.V2;  const long g_b0 = q.v_b0;  // Gram matrix / series, correction (N x N row-major)
  real* scale = w.norms;            // [B] scratch until the finish kernel writes the norms
  real* e2fro2 = w.norms + nb;      // [B]
  real* fro2 = w.fro2;              // [B] ||theta||_F^2 (the fp64 Jacobi of a fallback recomputes it: same quantity)
  int* flag = w.n_active + 3;

  float2_t* th32 = static_cast<float2_t*>(mx.base);
  hipLaunchKernelGGL(c64_scale_kernel, dim3(nb), dim3(256), 0, s, d.theta, d.theta_b0, nn, scale, fro2);
  hipLaunchKernelGGL(to_c64_kernel, dim3(gx, nb), dim3(256), 0, s, d.theta, d.theta_b0, th32, x_b0, nn, scale);
  TJM_HIP_CHECK(hipGetLastError());
  tjm32::MixedBasisDesc mb;
  mb.N = N; mb.d = d.d; mb.dist = 1 - d.distribution; mb.nb0 = nb; mb.h_pinned = w.h_pinned;
  static const int c64_cap = getenv("TJM_MIXED_C64_SWEEPS") ? atoi(getenv("TJM_MIXED_C64_SWEEPS")) : 10;
  mb.max_sweeps = c64_cap;
  static const double c64_stop = getenv("TJM_MIXED_C64_STOP") ? atof(getenv("TJM_MIXED_C64_STOP")) : 0.1;
  mb.stop_fraction = c64_stop;
  const void* basis = nullptr;
  long basis_b0 = 0;
  int c64_sweeps = 0;
  if ((rc = tjm32::mixed_left_basis(mb, mx.base, c64_bytes, mx.max_dim, mx.B, s, &basis, &basis_b0, &c64_sweeps)) != TJM_OK) return rc;
  hipLaunchKernelGGL(from_c64_kernel, dim3(gx, nb), dim3(256), 0, s, static_cast<const float2_t*>(basis), basis_b0, Va, v_b0, N, d.d);

  const int* l_ids = nullptr;  // the batched products below run over all nb trajectories, or over a list of l_nb of them
  int l_nb = nb;
  auto square = [&](int n) { GemmDesc g; memset(&g, 0, sizeof(g)); g.nks = 1; g.nb0 = l_nb; g.ids = l_ids; g.nb1 = 1; g.nb2 = 1; g.M = n; g.N = n; g.K = n; return g; };
  // Gram matrix of a column-major N x N matrix: G[i][j] = sum_r conj(A[r + i N]) A[r + j N]
  auto gram = [&](const cplx* A, long a_b0, cplx* G, long gb0) {
    GemmDesc g = square(N);
    g.A = A; g.a_rs = N; g.a_cs = 1; g.a_b0 = a_b0; g.conjA = 1;
    g.B = A; g.b_rs = 1; g.b_cs = N; g.b_b0 = a_b0;
    g.C = G; g.c_rs = N; g.c_b0 = gb0;
    g.hermitian = 1;
    return mixed_gemm(g, s);
  };
  // row-major product of two row-major N x N matrices; herm: the product is Hermitian (the square of a Hermitian or of an
  // anti-Hermitian matrix), only the tiles on and above the diagonal are computed
  auto rowmul = [&](const cplx* A, long a_b0, const cplx* Bm, long b_b0, cplx* Cc, long c_b0, bool herm) {
    GemmDesc g = square(N);
    g.hermitian = herm ? 1 : 0;
    g.A = A; g.a_rs = N; g.a_cs = 1; g.a_b0 = a_b0;
    g.B = Bm; g.b_rs = N; g.b_cs = 1; g.b_b0 = b_b0;
    g.C = Cc; g.c_rs = N; g.c_b0 = c_b0;
    return mixed_gemm(g, s);
  };
  // column-major Vout = Vin T (T row-major), written as the row-major matrix C[j][r] = sum_i T[i][j] Vin[r + i N]
  auto apply = [&](const cplx* Vin, const cplx* T, long t_b0, cplx* Vout) {
    GemmDesc g = square(N);
    g.A = T; g.a_rs = 1; g.a_cs = N; g.a_b0 = t_b0;
    g.B = Vin; g.b_rs = N; g.b_cs = 1; g.b_b0 = v_b0;
    g.C = Vout; g.c_rs = N; g.c_b0 = v_b0;
    return mixed_gemm(g, s);
  };
  // polar step Vin -> Vout; second = true: series to second order, cert accumulates ||E^2||_F^2 per trajectory;
  // second = false: I - E/2 only (for ||E|| <~ 1e-6 the second-order term is below rounding)
  auto polar = [&](const cplx* Vin, cplx* Vout, bool second, real* cert) {
    int r;
    if ((r = gram(Vin, v_b0, Gm, g_b0)) != TJM_OK) return r;
    hipLaunchKernelGGL(polar_residual_kernel, dim3((N + 255) / 256, l_nb), dim3(256), 0, s, Gm, g_b0, N, (const int*)nullptr, 0, l_ids);
    if (second) {
      if ((r = rowmul(Gm, g_b0, Gm, g_b0, S2, x_b0, true)) != TJM_OK) return r;  // E is Hermitian
      hipLaunchKernelGGL(polar_poly_kernel, dim3(gx, l_nb), dim3(256), 0, s, Gm, g_b0, S2, x_b0, Cm, g_b0, N, real(0.375), cert, l_ids);
    } else {
      hipLaunchKernelGGL(polar_poly_kernel, dim3(gx, l_nb), dim3(256), 0, s, Gm, g_b0, Gm, g_b0, Cm, g_b0, N, real(0.0), (real*)nullptr, l_ids);
    }
    return apply(Vin, Cm, g_b0, Vout);
  };
  // X = Z V into the Jacobi workspace, column-major: Y[k N + r] = sum_q V[q + k N] Z[r][q], q in theta's own index order
  auto form_x = [&](const cplx* V) {
    GemmDesc g = square(N);
    g.A = V; g.a_rs = N; g.a_cs = 1; g.a_b0 = v_b0;
    g.B = d.theta; g.b_b0 = d.theta_b0;
    if (d.distribution == 0) {  // Z = theta: r = (s, a), q = (t, c)
      g.b_rs = 1; g.b_cs = d.ld_theta;
    } else {                    // Z = theta^H: r = (t, c), q = (s, a): conj(theta[q][r])
      g.b_rs = d.ld_theta; g.b_cs = 1; g.conjB = 1;
    }
    g.C = w.Y; g.c_rs = N; g.c_b0 = w.y_b0;
    return mixed_gemm(g, s);
  };
  if (nn > w.y_b0) return TJM_ERR_WORKSPACE;

  // ---- polar step of the complex64 basis, certified
  TJM_HIP_CHECK(hipMemsetAsync(e2fro2, 0, (size_t)nb * sizeof(real), s));
  TJM_HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(int), s));
  if ((rc = polar(Va, Vb, true, e2fro2)) != TJM_OK) return rc;
  // (5/16) ||E^2||_F^(3/2) <= 1e-13  <=>  ||E^2||_F^2 <= 2.2e-17 ; anything larger says the complex64 basis is not what it should be
  hipLaunchKernelGGL(polar_check_kernel, dim3((nb + 255) / 256), dim3(256), 0, s, e2fro2, nb, real(2.2e-17), flag, pneed);
  TJM_HIP_CHECK(hipMemcpyAsync(w.h_pinned + 6, flag, sizeof(int), hipMemcpyDeviceToHost, s));
  TJM_HIP_CHECK(hipStreamSynchronize(s));
  cplx* Vstart = Vb;
  cplx* Vspare = Va;
  bool took_second = false;
  if (w.h_pinned[6] != 0) {
    // the complex64 iteration stopped a little early for some trajectory (its basis is orthonormal to 1e-5 rather than 2e-6): the
    // polar iteration converges cubically, a second step of the same kind brings its residual to rounding - and is certified again.
    // Only the trajectories that asked for it take it (index list: a trajectory's result must not depend on who shares its batch,
    // and a third of the batches of the evolved state hold such a trajectory - usually one or two of 256); one that fails again is
    // left to the fp64 kernels at the end.
    std::vector<int> hn(nb), second_ids;
    TJM_HIP_CHECK(hipMemcpy(hn.data(), pneed, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost));
    for (int b = 0; b < nb; ++b) if (hn[b]) second_ids.push_back(b);
    TJM_HIP_CHECK(hipMemcpy(idlist, second_ids.data(), second_ids.size() * sizeof(int), hipMemcpyHostToDevice));
    TJM_HIP_CHECK(hipMemsetAsync(e2fro2, 0, (size_t)nb * sizeof(real), s));
    TJM_HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(int), s));
    l_ids = idlist;
    l_nb = (int)second_ids.size();
    rc = polar(Vb, Va, true, e2fro2);
    l_ids = nullptr;
    l_nb = nb;
    if (rc != TJM_OK) return rc;
    hipLaunchKernelGGL(listed_copy_kernel, dim3(gx, (unsigned)second_ids.size()), dim3(256), 0, s, Vb, Va, v_b0, nn, idlist);
    hipLaunchKernelGGL(polar_check_kernel, dim3((nb + 255) / 256), dim3(256), 0, s, e2fro2, nb, real(2.2e-17), flag, status);
    hipLaunchKernelGGL(and_flags_kernel, dim3((nb + 255) / 256), dim3(256), 0, s, pneed, status, nb);  // gives up: asked for the second step AND failed it
    took_second = true;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    ++g_mixed.second_polar;
  }
  if (g_debug) {
    std::vector<real> h(nb);
    TJM_HIP_CHECK(hipMemcpyAsync(h.data(), e2fro2, (size_t)nb * sizeof(real), hipMemcpyDeviceToHost, s));
    TJM_HIP_CHECK(hipStreamSynchronize(s));
    real worst = 0.0;
    for (int b = 0; b < nb; ++b) worst = (h[b] > worst || h[b] != h[b]) ? h[b] : worst;
    fprintf(stderr, "[svd-mixed] certificate ||E^2||_F: worst %.3e (limit 4.7e-9)\n", sqrt(worst));
  }
  // ---- quadratic refinement on the matrix cores
  // columns beyond skip_col are not corrected among themselves (point 3 above)
  const bool can_skip = (d.trunc_mode == 0 || d.trunc_mode == 3) && d.spectrum == nullptr && getenv("TJM_MIXED_NO_SKIP") == nullptr;
  const int skip_col = (can_skip && cm + 32 <= N - 32) ? cm + 32 : N;
  const int cm_far = can_skip ? cm : N;  // first position a never-corrected column may have (N: every pair is corrected)
  static const int n_ref = getenv("TJM_MIXED_REFINE") ? atoi(getenv("TJM_MIXED_REFINE")) : 2;
  cplx* Vcur = Vstart;
  cplx* Vnext = Vspare;
  // Two rounds (the default): the basis itself is not carried along.  X_1 = X_0 T_1 and X_2 = X_1 T_2 are products of what is there
  // already (one GEMM each instead of V T and theta (V T); the rounding of X T puts eps |T_jk| sigma_j into column k - 2e-12 of a
  // column six decades below the largest in the first round, 2e-14 in the last - where theta V puts eps sigma_max), and the
  // non-unitarity the truncated exponential T_1 leaves is E_1 = T_1^H T_1 - I (the basis behind X_0 is unitary to rounding after the
  // certified polar step), which the last round removes together with its correction (W = C - E_1 / 2) as before.
  // TJM_MIXED_UPDATE_V: the rounds on V (V <- V T, X = theta V afresh), two more GEMMs per split.
  static const bool update_v = getenv("TJM_MIXED_UPDATE_V") != nullptr;
  const bool on_x = !update_v && n_ref == 2 && 2 * nn <= w.y_b0;
  // The squares C^2 and W^2 of those rounds need three digits (C^2 / 2 stands next to I + C with |C| <= 0.01 and what it misses is
  // measured and removed by the next round - E_1 and the Gram matrix are taken from the actual T_1 and X_1; |W| <= 1e-4 leaves
  // 1e-7 |W^2| <= 3e-13 at the very worst): they come from the complex64 GEMM at twice the rate, fed by a complex64 copy that the
  // correction kernel writes along.  TJM_MIXED_FP64_SQUARES: in fp64.
  static const bool fp64_squares = getenv("TJM_MIXED_FP64_SQUARES") != nullptr;
  float2_t* sq_in = nullptr;
  const float2_t* sq_out = nullptr;
  long c32_b0 = 0;
  if (on_x && !fp64_squares) {
    void* in = nullptr;
    const void* out = nullptr;
    if ((rc = tjm32::mixed_square_buffers(mx.base, c64_bytes, mx.max_dim, mx.B, &in, &out, &c32_b0)) != TJM_OK) return rc;
    sq_in = static_cast<float2_t*>(in);
    sq_out = static_cast<const float2_t*>(out);
  }
  auto square_of_c = [&](bool herm) {  // S2 (fp64) or sq_out (complex64) = Cm x Cm
    if (sq_in == nullptr) return rowmul(Cm, g_b0, Cm, g_b0, S2, x_b0, herm);
    { std::lock_guard<std::mutex> lock(g_prof_mutex); ++g_mixed.c64_gemms; }
    return tjm32::mixed_square(mx.base, c64_bytes, mx.max_dim, mx.B, N, nb, herm ? 1 : 0, s);
  };
  if (on_x) {
    cplx* Xa = w.Y;
    cplx* Xb = w.Y + nn;  // second half of every trajectory's slab (laid out for rows + columns of the accumulating variant)
    auto times = [&](const cplx* Xin, const cplx* T, long t_b0, cplx* Xout) {  // column-major Xout = Xin T (T row-major)
      GemmDesc g = square(N);
      g.A = T; g.a_rs = 1; g.a_cs = N; g.a_b0 = t_b0;
      g.B = Xin; g.b_rs = N; g.b_cs = 1; g.b_b0 = w.y_b0;
      g.C = Xout; g.c_rs = N; g.c_b0 = w.y_b0;
      return mixed_gemm(g, s);
    };
    if ((rc = form_x(Vcur)) != TJM_OK) return rc;
    if ((rc = gram(Xa, w.y_b0, Gm, g_b0)) != TJM_OK) return rc;
    hipLaunchKernelGGL(refine_corr_kernel, dim3(gx, nb), dim3(256), 0, s, Gm, g_b0, N, skip_col, cm_far, fro2, real(0.01), Cm, g_b0, (const cplx*)nullptr, 0L, sq_in, c32_b0);
    if ((rc = square_of_c(true)) != TJM_OK) return rc;  // C is anti-Hermitian: C^2 is Hermitian
    hipLaunchKernelGGL(refine_poly_kernel, dim3(gx, nb), dim3(256), 0, s, Cm, g_b0, S2, x_b0, Gm, g_b0, N, sq_out, c32_b0);  // T_1
    if ((rc = times(Xa, Gm, g_b0, Xb)) != TJM_OK) return rc;
    {  // E_1 = T_1^H T_1 - I
      GemmDesc g = square(N);
      g.A = Gm; g.a_rs = 1; g.a_cs = N; g.a_b0 = g_b0; g.conjA = 1;
      g.B = Gm; g.b_rs = N; g.b_cs = 1; g.b_b0 = g_b0;
      g.C = Iso; g.c_rs = N; g.c_b0 = x_b0;
      g.hermitian = 1;
      if ((rc = mixed_gemm(g, s)) != TJM_OK) return rc;
      hipLaunchKernelGGL(polar_residual_kernel, dim3((N + 255) / 256, nb), dim3(256), 0, s, Iso, x_b0, N, (const int*)nullptr, 0, (const int*)nullptr);
    }
    if ((rc = gram(Xb, w.y_b0, Gm, g_b0)) != TJM_OK) return rc;
    hipLaunchKernelGGL(refine_corr_kernel, dim3(gx, nb), dim3(256), 0, s, Gm, g_b0, N, skip_col, cm_far, fro2, real(1e-4), Cm, g_b0, (const cplx*)Iso, x_b0, sq_in, c32_b0);
    if ((rc = square_of_c(false)) != TJM_OK) return rc;  // W = C - E/2 is neither
    hipLaunchKernelGGL(refine_poly_kernel, dim3(gx, nb), dim3(256), 0, s, Cm, g_b0, S2, x_b0, Gm, g_b0, N, sq_out, c32_b0);  // T_2
    if ((rc = times(Xb, Gm, g_b0, Xa)) != TJM_OK) return rc;
  }
  for (int it = 0; it < n_ref && !on_x; ++it) {
    if ((rc = form_x(Vcur)) != TJM_OK) return rc;
    if ((rc = gram(w.Y, w.y_b0, Gm, g_b0)) != TJM_OK) return rc;
    if (it + 1 < n_ref) {
      hipLaunchKernelGGL(refine_corr_kernel, dim3(gx, nb), dim3(256), 0, s, Gm, g_b0, N, skip_col, cm_far, fro2, real(0.01), Cm, g_b0, (const cplx*)nullptr, 0L, (float2_t*)nullptr, c32_b0);
      if ((rc = rowmul(Cm, g_b0, Cm, g_b0, S2, x_b0, true)) != TJM_OK) return rc;  // C is anti-Hermitian: C^2 is Hermitian
      hipLaunchKernelGGL(refine_poly_kernel, dim3(gx, nb), dim3(256), 0, s, Cm, g_b0, S2, x_b0, Gm, g_b0, N, (const float2_t*)nullptr, c32_b0);
      if ((rc = apply(Vcur, Gm, g_b0, Vnext)) != TJM_OK) return rc;
    } else {
      // last round: the truncated exponentials of the rounds before are unitary to |C|^3 / 6 <= 2e-7 (|C_ij| <= 0.01); E = V^H V - I
      // goes into the same factor as the last correction, W = C - E/2, applied as I + W + W^2/2
      if ((rc = gram(Vcur, v_b0, Iso, x_b0)) != TJM_OK) return rc;
      hipLaunchKernelGGL(polar_residual_kernel, dim3((N + 255) / 256, nb), dim3(256), 0, s, Iso, x_b0, N, (const int*)nullptr, 0, (const int*)nullptr);
      hipLaunchKernelGGL(refine_corr_kernel, dim3(gx, nb), dim3(256), 0, s, Gm, g_b0, N, skip_col, cm_far, fro2, real(1e-4), Cm, g_b0, (const cplx*)Iso, x_b0, (float2_t*)nullptr, c32_b0);
      if ((rc = rowmul(Cm, g_b0, Cm, g_b0, S2, x_b0, false)) != TJM_OK) return rc;  // W = C - E/2 is neither
      hipLaunchKernelGGL(refine_poly_kernel, dim3(gx, nb), dim3(256), 0, s, Cm, g_b0, S2, x_b0, Gm, g_b0, N, (const float2_t*)nullptr, c32_b0);
      if ((rc = apply(Vcur, Gm, g_b0, Vnext)) != TJM_OK) return rc;
    }
    std::swap(Vcur, Vnext);
  }
  if (!on_x && (rc = form_x(Vcur)) != TJM_OK) return rc;
  if ((rc = gram(w.Y, w.y_b0, Gm, g_b0)) != TJM_OK) return rc;
  TruncSpec tr;
  tr.trunc_mode = d.trunc_mode; tr.threshold = d.threshold; tr.max_bond = d.max_bond; tr.min_keep = d.min_keep;
  tr.cap = d.capM; tr.overflow = nullptr; tr.overflow_each = clipped;  // (the fp64 Jacobi of a failed check finishes its trajectories with the same spec)
  tr.chiA = d.chiL; tr.mulA = d.d; tr.chiB = d.chiR; tr.mulB = d.d; tr.chiOut = d.chiM; tr.chi_stride = d.chi_stride;
  tr.spectrum = d.spectrum; tr.spec_ld = d.spec_ld;
  JacobiShape sh;
  sh.ncols_pad = N; sh.rx_top = N; sh.rtot = N;
  // norms, order, truncation of every trajectory from the columns of X; then the check of what was kept
  TJM_HIP_CHECK(hipMemsetAsync(w.n_active + 2, 0, 4 * sizeof(int), s));  // [2]: finish kernel's rank flag, [5]: the check's count ([3] = the polar flag, read already; [4] unused)
  hipLaunchKernelGGL(svd_finish_kernel, dim3(nb), dim3(256), 0, s, tr, w, N, N, N, (const int*)nullptr);
  const int strict = can_skip ? 0 : 1;
  hipLaunchKernelGGL(refine_check_kernel, dim3(nb), dim3(256), 0, s, Gm, g_b0, N, fro2, w.perm, d.chiM, d.chi_stride, strict, pneed, status, w.n_active + 5);
  TJM_HIP_CHECK(hipGetLastError());
  TJM_HIP_CHECK(hipMemcpyAsync(w.h_pinned + 7, w.n_active + 5, sizeof(int), hipMemcpyDeviceToHost, s));
  TJM_HIP_CHECK(hipStreamSynchronize(s));
  const int n_bad = w.h_pinned[7];
  if (g_debug) fprintf(stderr, "[svd-mixed] N %d c64 sweeps %d second polar %d trajectories left to the fp64 Jacobi %d of %d\n", N, c64_sweeps, (int)took_second, n_bad, nb);
  int f64_sweeps = 0;
  if (n_bad > 0) {
    // the trajectories the check did not pass: fp64 Jacobi sweeps on their X (every pair; nearly diagonal already)
    std::vector<int> hs(nb), ids;
    TJM_HIP_CHECK(hipMemcpy(hs.data(), status, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost));
    for (int b = 0; b < nb; ++b) if (hs[b]) ids.push_back(b);
    if (g_debug) {
      int c1 = 0, c2 = 0, c4 = 0;
      for (int b = 0; b < nb; ++b) { c1 += hs[b] & 1; c2 += (hs[b] >> 1) & 1; c4 += (hs[b] >> 2) & 1; }
      fprintf(stderr, "[svd-mixed]   kept/not-kept tilt open in %d, Gershgorin certificate failed in %d, kept set not orthogonal in %d trajectories (skip_col %d)\n", c1, c2, c4, skip_col);
    }
    TJM_HIP_CHECK(hipMemcpyAsync(idlist, ids.data(), ids.size() * sizeof(int), hipMemcpyHostToDevice, s));
    TJM_HIP_CHECK(hipStreamSynchronize(s));  // ids is a local
    JacobiSource src;
    src.src = nullptr; src.src_b0 = 0; src.rx = N; src.ncols = N; src.conj = 0; src.tri = 0;
    src.r_n0 = N; src.s_r1 = 0; src.s_r0 = 0; src.c_n0 = N; src.s_c1 = 0; src.s_c0 = 0;
    src.nb0 = (int)ids.size(); src.ids = idlist;
    JacobiOpts op;
    op.preloaded = true;
    op.late_start = true;
    if ((rc = jacobi_solve(src, tr, w, s, &sh, &f64_sweeps, false, &op)) != TJM_OK) return rc;  // ends with the finish kernel of these
  }
  // which trajectories does the all-fp64 path have to serve (per trajectory: nobody else's result depends on it)
  TJM_HIP_CHECK(hipMemsetAsync(w.n_active + 5, 0, sizeof(int), s));
  hipLaunchKernelGGL(mixed_fallback_kernel, dim3((nb + 255) / 256), dim3(256), 0, s, w.norms, N, d.chiM, d.chi_stride, pneed, nb, status, w.n_active + 5,
                     clipped, d.overflow);
  TJM_HIP_CHECK(hipMemcpyAsync(w.h_pinned + 7, w.n_active + 5, sizeof(int), hipMemcpyDeviceToHost, s));
  TJM_HIP_CHECK(hipStreamSynchronize(s));
  const int n_fb = w.h_pinned[7];
  std::vector<int> fb_ids;
  if (n_fb > 0) {
    std::vector<int> hs(nb);
    TJM_HIP_CHECK(hipMemcpy(hs.data(), status, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost));
    for (int b = 0; b < nb; ++b) if (hs[b]) fb_ids.push_back(b);
  }
  {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    ++g_mixed.solves;
    g_mixed.c64_sweeps += c64_sweeps;
    g_mixed.f64_sweeps += f64_sweeps;
    g_mixed.jacobi_trajectories += n_bad;
    g_mixed.fallbacks += n_fb;
  }
  if (sweeps_out) *sweeps_out = f64_sweeps;
  // (n_fb == nb: nobody is left for this path.  The listed trajectories are still served by the SAME code as a partial list - the
  // plain fp64 split with an index list - so that a trajectory's last bits do not depend on who shares its batch.)
  if (n_fb < nb) {
  // ---- isometric factor: normalised kept columns, made exactly isometric by a polar step of their own; then the projection
  ExtractDesc xi;
  GemmDesc gg, gt, gp;  // Gram of the raw isometry, raw x T, projection
  memset(&gg, 0, sizeof(gg)); memset(&gt, 0, sizeof(gt)); memset(&gp, 0, sizeof(gp));
  gg.nb0 = gt.nb0 = gp.nb0 = nb; gg.nb1 = gg.nb2 = gt.nb2 = gp.nb2 = 1;
  gg.M = cm; gg.N = cm; gg.C = Gm; gg.c_rs = cm; gg.c_b0 = g_b0;
  if ((long)cm * cm > g_b0 || (long)N * cm > x_b0) return TJM_ERR_WORKSPACE;
  if (d.distribution == 0) {
    // raw[(s,a)][k] = Ytilde[(s,a)][k] (the layout of `left`) ; left = raw T ; right[t][k][c] = sum_(s,a) conj(left[(s,a)][k]) theta[(s,a)][(t,c)]
    xi.out = Iso; xi.out_b0 = x_b0; xi.n_k = cm; xi.o_k = 1; xi.n_r1 = 1; xi.n_r0 = N; xi.o_r1 = 0; xi.o_r0 = cm;
    xi.row_off = 0; xi.conj = 0; xi.scale_mode = 2;
    gg.nks = 1; gg.K = N;
    gg.A = Iso; gg.a_rs = 1; gg.a_cs = cm; gg.a_b0 = x_b0; gg.conjA = 1;
    gg.B = Iso; gg.b_rs = cm; gg.b_cs = 1; gg.b_b0 = x_b0;
    gt.nks = 1; gt.nb1 = 1; gt.M = N; gt.N = cm; gt.K = cm;
    gt.A = Iso; gt.a_rs = cm; gt.a_cs = 1; gt.a_b0 = x_b0;
    gt.B = Cm; gt.b_rs = cm; gt.b_cs = 1; gt.b_b0 = g_b0;
    gt.C = d.left; gt.c_rs = cm; gt.c_b0 = d.left_b0;
    gp.nks = 1; gp.nb1 = d.d;
    gp.A = d.left; gp.a_rs = 1; gp.a_cs = cm; gp.a_b0 = d.left_b0; gp.conjA = 1; gp.M = cm; gp.K = N;
    gp.B = d.theta; gp.b_rs = d.ld_theta; gp.b_cs = 1; gp.b_b0 = d.theta_b0; gp.b_b1 = d.capR; gp.N = d.capR;
    gp.C = d.right; gp.c_rs = d.capR; gp.c_b0 = d.right_b0; gp.c_b1 = (long)cm * d.capR;
  } else {
    // raw[t][k][c] = conj(Ytilde[(t,c)][k]) (the layout of `right`) ; right[t] = T^H raw[t] ; left[(s,a)][k] = sum_(t,c) theta[(s,a)][(t,c)] conj(right[t][k][c])
    const long tkc = (long)cm * d.capR;
    xi.out = Iso; xi.out_b0 = x_b0; xi.n_k = cm; xi.o_k = d.capR; xi.n_r1 = d.d; xi.n_r0 = d.capR;
    xi.o_r1 = tkc; xi.o_r0 = 1; xi.row_off = 0; xi.conj = 1; xi.scale_mode = 2;
    // Gram[i][j] = sum_(t,c) conj(iso[(t,c)][i]) iso[(t,c)][j] = sum_t sum_c raw[t][i][c] conj(raw[t][j][c])
    gg.nks = d.d; gg.K = d.capR;
    gg.A = Iso; gg.a_rs = d.capR; gg.a_cs = 1; gg.a_ks = tkc; gg.a_b0 = x_b0;
    gg.B = Iso; gg.b_rs = 1; gg.b_cs = d.capR; gg.b_ks = tkc; gg.b_b0 = x_b0; gg.conjB = 1;
    gt.nks = 1; gt.nb1 = d.d; gt.M = cm; gt.N = d.capR; gt.K = cm;
    gt.A = Cm; gt.a_rs = 1; gt.a_cs = cm; gt.a_b0 = g_b0; gt.conjA = 1;
    gt.B = Iso; gt.b_rs = d.capR; gt.b_cs = 1; gt.b_b0 = x_b0; gt.b_b1 = tkc;
    gt.C = d.right; gt.c_rs = d.capR; gt.c_b0 = d.right_b0; gt.c_b1 = tkc;
    gp.nks = d.d; gp.nb1 = 1;
    gp.A = d.theta; gp.a_rs = d.ld_theta; gp.a_cs = 1; gp.a_ks = d.capR; gp.a_b0 = d.theta_b0; gp.M = d.m; gp.K = d.capR;
    gp.B = d.right; gp.b_rs = 1; gp.b_cs = d.capR; gp.b_ks = tkc; gp.b_b0 = d.right_b0; gp.conjB = 1; gp.N = cm;
    gp.C = d.left; gp.c_rs = cm; gp.c_b0 = d.left_b0;
  }
  if ((rc = svd_extract(xi, w, sh, d.chiM, d.chi_stride, nb, nullptr, s)) != TJM_OK) return rc;
  gg.hermitian = 1;
  if ((rc = mixed_gemm(gg, s)) != TJM_OK) return rc;
  const int gxc = (int)std::min<long>(128, ((long)cm * cm + 1023) / 1024);
  hipLaunchKernelGGL(polar_residual_kernel, dim3((cm + 255) / 256, nb), dim3(256), 0, s, Gm, g_b0, cm, d.chiM, d.chi_stride, (const int*)nullptr);
  // (the kept columns are orthogonal to ~1e-15 ||theta|| / sigma_k, 2e-6 for a near-degenerate pair left alone: first order is exact to rounding)
  hipLaunchKernelGGL(polar_poly_kernel, dim3(gxc, nb), dim3(256), 0, s, Gm, g_b0, Gm, g_b0, Cm, g_b0, cm, real(0.0), (real*)nullptr, (const int*)nullptr);
  TJM_HIP_CHECK(hipGetLastError());
  if ((rc = mixed_gemm(gt, s)) != TJM_OK) return rc;
  if ((rc = mixed_gemm(gp, s)) != TJM_OK) return rc;
  }
  if (n_fb > 0) {
    // the plain fp64 split (Jacobi with the accumulated unitary: isometric whatever the rank) for the listed trajectories; it
    // rewrites their outputs and their bond entry
    TJM_HIP_CHECK(hipMemcpyAsync(idlist, fb_ids.data(), fb_ids.size() * sizeof(int), hipMemcpyHostToDevice, s));
    TJM_HIP_CHECK(hipStreamSynchronize(s));  // fb_ids is a local
    SvdSplitDesc d2 = d;
    d2.ids = idlist;
    d2.nb0 = n_fb;
    int sw2 = 0;
    // (above 512 rows the stacked-column split does not hold the matrix: the X-only direct variant serves index lists as well)
    if ((rc = (N > 512 ? svd_split_qr2(d2, w, q, s, &sw2) : svd_split(d2, w, s, &sw2))) != TJM_OK) return rc;
  }
  *done = true;
  return TJM_OK;
}
#else
void mixed_stats_get(double* out10, bool) { for (int i = 0; i < 10; ++i) out10[i] = 0.0; }
void mixed_qr_profile_enable(int) {}
void mixed_qr_profile_get(double* out5) { for (int i = 0; i < 5; ++i) out5[i] = 0.0; }
void mixed_profile_enable(int) {}
void mixed_profile_get(double* total_ms, double* total_bytes, long* samples) { *total_ms = 0.0; *total_bytes = 0.0; *samples = 0; }
size_t mixed_split_workspace_bytes(int, int) { return 0; }
#endif

int svd_split_qr(const SvdSplitDesc& d, const SvdWorkspace& w, const QrWorkspace& q, hipStream_t s, int* sweeps_out, const MixedWorkspace* mx) {
  if (d.nb0 <= 0) return TJM_OK;
  // beyond the stacked-column Jacobi (rows + columns <= 1024): X-only direct variant.  TJM_FORCE_LARGE_SPLIT sends every split of
  // at least 32 x 32 down that path (diagnostic: the large-bond code at sizes the rest of the suite covers)
  static const bool force_large = getenv("TJM_FORCE_LARGE_SPLIT") != nullptr;
  if (d.distribution == 2) return (d.m > 512 || d.n > 512) ? TJM_ERR_NOT_IMPLEMENTED : svd_split(d, w, s, sweeps_out);
  const bool large = (d.m > 512 || d.n > 512) || (force_large && q.Z2 != nullptr && d.m >= 32 && d.n >= 32);
  if (d.ids && !large) return svd_split(d, w, s, sweeps_out);  // index-list batches take the plain path
  if (d.ld_theta != d.n) return TJM_ERR_ARG;
  static const bool single_qr = getenv("TJM_SINGLE_QR") != nullptr;
  if (large) {
    if (q.Z2 == nullptr) return TJM_ERR_WORKSPACE;
#ifndef TJM_F32
    if (!force_large && !single_qr && mixed_split_fits(d, q, mx)) {  // square splits of 513 ... 1024 rows (TJM_MIXED_MAX_DIM=512: not served, no workspace)
      bool done = false;
      const int rcm = svd_split_mixed(d, w, q, *mx, s, sweeps_out, &done);
      if (rcm != TJM_OK || done) return rcm;
    }
#endif
    return svd_split_qr2(d, w, q, s, sweeps_out);
  }
#ifndef TJM_F32
  if (!single_qr && mixed_split_fits(d, q, mx)) {
    bool done = false;
    const int rcm = svd_split_mixed(d, w, q, *mx, s, sweeps_out, &done);
    if (rcm != TJM_OK || done) return rcm;
  }
#endif
  if (!single_qr && q.Z2 != nullptr && d.m == d.n && d.m >= 32) return svd_split_qr2(d, w, q, s, sweeps_out);
  const int zr = (d.distribution == 0) ? d.m : d.n;
  const int zc = (d.distribution == 0) ? d.n : d.m;
  const int kmax = zr < zc ? zr : zc;
  int rc;
  if ((rc = qr_prepare(d.theta, d.theta_b0, d.m, d.n, d.distribution, d.d, q, d.nb0, d.ids, s)) != TJM_OK) return rc;
  if ((rc = qr_factor(q, zr, zc, d.nb0, d.ids, s)) != TJM_OK) return rc;
  JacobiSource src;  // X = R^H : X[r][c] = conj(R[c][r]) , R[i][j] = Z[j * zr + i] for i <= j
  src.src = q.Z; src.src_b0 = q.z_b0; src.rx = zc; src.ncols = kmax; src.conj = 1; src.tri = 1;
  src.r_n0 = zc; src.s_r1 = 0; src.s_r0 = zr; src.c_n0 = kmax; src.s_c1 = 0; src.s_c0 = 1;
  src.nb0 = d.nb0; src.ids = d.ids;
  TruncSpec tr;
  tr.trunc_mode = d.trunc_mode; tr.threshold = d.threshold; tr.max_bond = d.max_bond; tr.min_keep = d.min_keep;
  tr.cap = d.capM; tr.overflow = d.overflow;
  tr.chiA = d.chiL; tr.mulA = d.d; tr.chiB = d.chiR; tr.mulB = d.d; tr.chiOut = d.chiM; tr.chi_stride = d.chi_stride;
  tr.spectrum = d.spectrum; tr.spec_ld = d.spec_ld;
  JacobiShape sh;
  if ((rc = jacobi_solve(src, tr, w, s, &sh, sweeps_out)) != TJM_OK) return rc;
  // isometric factor Q W: W (sorted, first capM columns) into Z, rows beyond the W block are zero
  TJM_HIP_CHECK(hipMemsetAsync(q.Z, 0, (size_t)q.z_b0 * sizeof(cplx) * (size_t)d.nb0, s));
  ExtractDesc xw;
  xw.out = q.Z; xw.out_b0 = q.z_b0; xw.n_k = d.capM; xw.o_k = zr; xw.n_r1 = 1; xw.n_r0 = (zr < sh.ncols_pad) ? zr : sh.ncols_pad;
  xw.o_r1 = 0; xw.o_r0 = 1; xw.row_off = sh.rx_top; xw.conj = 0; xw.scale_mode = 0;
  if ((rc = svd_extract(xw, w, sh, d.chiM, d.chi_stride, d.nb0, d.ids, s)) != TJM_OK) return rc;
  if ((rc = qr_apply_q(q, zr, zc, q.Z, q.z_b0, d.capM, d.nb0, d.ids, s)) != TJM_OK) return rc;
  ExtractDesc xi, xx;
  if (d.distribution == 0) {
    // left[(s,a)][k] = (Q W)[(s,a)][k] ; right[t][k][c] = conj(Xfinal[(t,c)][k])
    xi.out = d.left; xi.out_b0 = d.left_b0; xi.n_k = d.capM; xi.o_k = 1; xi.n_r1 = d.capL; xi.n_r0 = d.d; xi.o_r1 = d.capM;
    xi.o_r0 = (long)d.capL * d.capM;  // rows of Q W are bond-major (a, s)
    xi.row_off = 0; xi.conj = 0; xi.scale_mode = 0;
    xx.out = d.right; xx.out_b0 = d.right_b0; xx.n_k = d.capM; xx.o_k = d.capR; xx.n_r1 = d.d; xx.n_r0 = d.capR;
    xx.o_r1 = (long)d.capM * d.capR; xx.o_r0 = 1; xx.row_off = 0; xx.conj = 1; xx.scale_mode = 0;
  } else {
    // right[t][k][c] = conj((Q W)[(t,c)][k]) ; left[(s,a)][k] = Xfinal[(s,a)][k]
    xi.out = d.right; xi.out_b0 = d.right_b0; xi.n_k = d.capM; xi.o_k = d.capR; xi.n_r1 = d.capR; xi.n_r0 = d.d;
    xi.o_r1 = 1; xi.o_r0 = (long)d.capM * d.capR;  // rows of Q W are bond-major (c, t)
    xi.row_off = 0; xi.conj = 1; xi.scale_mode = 0;
    xx.out = d.left; xx.out_b0 = d.left_b0; xx.n_k = d.capM; xx.o_k = 1; xx.n_r1 = 1; xx.n_r0 = d.d * d.capL; xx.o_r1 = 0; xx.o_r0 = d.capM;
    xx.row_off = 0; xx.conj = 0; xx.scale_mode = 0;
  }
  if ((rc = qr_scatter(q.Z, q.z_b0, zr, xi, d.chiM, d.chi_stride, d.nb0, d.ids, s)) != TJM_OK) return rc;
  // rows of R^H follow the norm-sorted column order of Z: undo the permutation while scattering
  xx.row_map = q.colperm();
  xx.row_map_ld = q.w_ld;
  return svd_extract(xx, w, sh, d.chiM, d.chi_stride, d.nb0, d.ids, s);
}

}  // namespace tjm
