// HBM-bound helper kernels of the TJM sweep: MPO application between the two GEMMs of
// H_eff / environment updates, the fused Lanczos vector operations, the per-trajectory
// Krylov bookkeeping (exp(-i dt T_k) e_1, adaptive stop), the basis combine, and the small
// per-site kernels (norms, scalings, local operators, jump decisions).
//
// Reference: core/methods/matrix_exponential.py:33-173 (expm_krylov),
// core/methods/tdvp/primitives.py:77-226, core/methods/stochastic_process.py:190-292.
#include <atomic>

#include "tjm_kernels.h"

namespace tjm {

// ------------------------------------------------------------------------------------------
// MPO apply:  out[po][bo][a][B] = sum_{pi,bi} Wm[(po,bo),(pi,bi)] * in[pi][bi][a][B]
// (strides per index are free, B is contiguous on both sides).  One thread per (a, B).  The operator is staged in LDS when it fits
// (w_in_lds; above 64 KiB the launcher raises the kernel's limit first); one larger than the LDS of a CU - a pair of four-level
// sites with an MPO bond above 6, a qubit pair above 25 - is read through the caches instead (every lane reads the same element).
// ------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(256) void mpo_apply_kernel(MpoApplyDesc d, int w_in_lds) {
  extern __shared__ real smem[];
  int b0 = blockIdx.y;
  if (d.ids) b0 = d.ids[b0];
  if (d.active && d.active[b0] == 0) return;
  const int nin = P * d.din, nout = P * d.dout;
  const cplx* __restrict__ sW = d.Wm;
  if (w_in_lds) {
    cplx* stage = reinterpret_cast<cplx*>(smem);
    for (int i = threadIdx.x; i < nin * nout; i += blockDim.x) stage[i] = d.Wm[i];
    __syncthreads();
    sW = stage;
  }
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)d.na * d.nB) return;
  const int a = idx / d.nB, Bc = idx % d.nB;
  const cplx* __restrict__ in = d.in + (long)b0 * d.in_b0 + (long)a * d.in_sa + Bc;
  cplx* __restrict__ out = d.out + (long)b0 * d.out_b0 + (long)a * d.out_sa + Bc;
  const cplx* __restrict__ in_alt = d.in_alt ? d.in_alt + (long)b0 * d.in_alt_b0 + (long)a * d.in_alt_sa + Bc : nullptr;
  cplx* __restrict__ out_alt = d.out_alt ? d.out_alt + (long)b0 * d.out_alt_b0 + (long)a * d.out_alt_sa + Bc : nullptr;
  constexpr int MAXD = 6;  // MPO bond dimensions held in registers (Ising 3, Heisenberg 5, exponential-sum models K + 2)
  if (d.din <= MAXD) {
    // every input element is loaded once and kept in registers for all dout outputs
    cplx x[P][MAXD];
#pragma unroll
    for (int pi = 0; pi < P; ++pi)
#pragma unroll
      for (int bi = 0; bi < MAXD; ++bi)
        x[pi][bi] = (bi < d.din) ? ((bi == d.in_alt_ch) ? in_alt[(long)pi * d.in_alt_sp] : in[(long)pi * d.in_sp + (long)bi * d.in_sb]) : cplx{0.0, 0.0};
    for (int bo = 0; bo < d.dout; ++bo) {
      cplx acc[P];
#pragma unroll
      for (int po = 0; po < P; ++po) acc[po] = cplx{0.0, 0.0};
#pragma unroll
      for (int bi = 0; bi < MAXD; ++bi) {
        if (bi < d.din) {
#pragma unroll
          for (int po = 0; po < P; ++po)
#pragma unroll
            for (int pi = 0; pi < P; ++pi) cfma(acc[po], sW[(po * d.dout + bo) * nin + pi * d.din + bi], x[pi][bi]);
        }
      }
      if (bo == d.out_alt_ch) {
#pragma unroll
        for (int po = 0; po < P; ++po) out_alt[(long)po * d.out_alt_sp] = acc[po];
      } else {
#pragma unroll
        for (int po = 0; po < P; ++po) out[(long)po * d.out_sp + (long)bo * d.out_sb] = acc[po];
      }
    }
    return;
  }
  for (int bo = 0; bo < d.dout; ++bo) {
    cplx acc[P];
#pragma unroll
    for (int po = 0; po < P; ++po) acc[po] = cplx{0.0, 0.0};
    for (int bi = 0; bi < d.din; ++bi) {
      cplx x[P];
#pragma unroll
      for (int pi = 0; pi < P; ++pi) x[pi] = (bi == d.in_alt_ch) ? in_alt[(long)pi * d.in_alt_sp] : in[(long)pi * d.in_sp + (long)bi * d.in_sb];
#pragma unroll
      for (int po = 0; po < P; ++po)
#pragma unroll
        for (int pi = 0; pi < P; ++pi) cfma(acc[po], sW[(po * d.dout + bo) * nin + pi * d.din + bi], x[pi]);
    }
    if (bo == d.out_alt_ch) {
#pragma unroll
      for (int po = 0; po < P; ++po) out_alt[(long)po * d.out_alt_sp] = acc[po];
    } else {
#pragma unroll
      for (int po = 0; po < P; ++po) out[(long)po * d.out_sp + (long)bo * d.out_sb] = acc[po];
    }
  }
}

namespace {
constexpr size_t LDS_DEFAULT = 64 * 1024, LDS_CU = 160 * 1024;  // dynamic LDS a launch may ask for without / with the attribute

template <int P>
int launch_mpo_apply_p(const MpoApplyDesc& d, dim3 grid, hipStream_t stream) {
  size_t sh = sizeof(cplx) * (size_t)(P * d.din) * (P * d.dout);
  int w_in_lds = 1;
  if (sh > LDS_CU) {
    w_in_lds = 0;
    sh = 0;
  } else if (sh > LDS_DEFAULT) {
    static std::atomic<bool> raised{false};  // several engines of one process call this from their own host threads  // once per process (one process per GPU)
    if (!raised.load(std::memory_order_acquire)) {
      TJM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(mpo_apply_kernel<P>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_CU));
      raised.store(true, std::memory_order_release);
    }
  }
  hipLaunchKernelGGL(mpo_apply_kernel<P>, grid, dim3(256), sh, stream, d, w_in_lds);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}
}  // namespace

int launch_mpo_apply(const MpoApplyDesc& d, hipStream_t stream) {
  const long n = (long)d.na * d.nB;
  if (n <= 0 || d.nb0 <= 0) return TJM_OK;
  dim3 grid((unsigned)((n + 255) / 256), d.nb0);
  switch (d.P) {
    case 1: return launch_mpo_apply_p<1>(d, grid, stream);
    case 2: return launch_mpo_apply_p<2>(d, grid, stream);
    case 4: return launch_mpo_apply_p<4>(d, grid, stream);
    case 3: return launch_mpo_apply_p<3>(d, grid, stream);    // qutrit site
    case 9: return launch_mpo_apply_p<9>(d, grid, stream);    // qutrit pair
    case 16: return launch_mpo_apply_p<16>(d, grid, stream);  // pair of four-level sites
    default: return TJM_ERR_NOT_IMPLEMENTED;
  }
}

// ------------------------------------------------------------------------------------------
// block reduction helper (256 threads, result valid in every thread)
// ------------------------------------------------------------------------------------------
__device__ inline real block_sum(real v, real* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  real t = 0.0;
  const int nw = (blockDim.x + 63) >> 6;
  for (int w = 0; w < nw; ++w) t += sh[w];
  return t;
}

// part[b][blk] = sum over this block's chunk of |x|^2          (x = vector j of trajectory b)
__global__ __launch_bounds__(256) void normsq_partial_kernel(const cplx* __restrict__ x, long x_b0, int n, real* part,
                                                            int nblk, const int* ids, const int* active) {
  __shared__ real sh[4];
  int b = blockIdx.y;
  if (ids) b = ids[b];
  if (active && active[b] == 0) return;
  const cplx* xb = x + (long)b * x_b0;
  real acc = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nblk * blockDim.x) {
    cplx v = xb[i];
    acc = fma(v.x, v.x, acc);
    acc = fma(v.y, v.y, acc);
  }
  acc = block_sum(acc, sh);
  if (threadIdx.x == 0) part[(long)b * nblk + blockIdx.x] = acc;
}

// part[b][blk] = Re <v, w> over the chunk
__global__ __launch_bounds__(256) void dot_partial_kernel(const cplx* __restrict__ v, const cplx* __restrict__ w, long v_b0,
                                                         long w_b0, int n, real* part, int nblk, const int* ids,
                                                         const int* active) {
  __shared__ real sh[4];
  int b = blockIdx.y;
  if (ids) b = ids[b];
  if (active && active[b] == 0) return;
  const cplx* vb = v + (long)b * v_b0;
  const cplx* wb = w + (long)b * w_b0;
  real acc = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nblk * blockDim.x) {
    cplx a = vb[i], c = wb[i];
    acc = fma(a.x, c.x, acc);
    acc = fma(a.y, c.y, acc);
  }
  acc = block_sum(acc, sh);
  if (threadIdx.x == 0) part[(long)b * nblk + blockIdx.x] = acc;
}

// w -= alpha v_j + beta_{j-1} v_{j-1};  part2[b][blk] = sum |w|^2.  alpha = sum(part1[b][:]).
// The stored vectors are unnormalised (KrylovState::svec): on entry w = H V[j], v_j = s_j V[j], v_{j-1} = s_{j-1} V[j-1] and part1
// holds <V[j], H V[j]>, so alpha = s_j^2 sum(part1) and the new (unnormalised) vector is s_j (w - alpha V[j]) - beta_{j-1} s_{j-1} V[j-1].
__global__ __launch_bounds__(256) void lanczos_axpy_kernel(cplx* __restrict__ w, const cplx* __restrict__ vj,
                                                          const cplx* __restrict__ vjm1, long v_b0, int n,
                                                          const real* part1, real* part2, int nblk,
                                                          const real* beta, int beta_ld, int j, const int* ids,
                                                          const int* active, const real* __restrict__ svec, int nblk1) {
  __shared__ real sh[4];
  int b = blockIdx.y;
  if (ids) b = ids[b];
  if (active && active[b] == 0) return;
  real alpha = 0.0;
  for (int i = 0; i < nblk1; ++i) alpha += part1[(long)b * nblk1 + i];  // nblk1 partial sums: dot_partial_kernel's blocks or the tiles of the GEMM epilogue
  const real sj = svec[(long)b * beta_ld + j];
  alpha *= sj * sj;
  const real bprev = (j > 0) ? beta[(long)b * beta_ld + j - 1] * svec[(long)b * beta_ld + j - 1] : 0.0;
  cplx* wb = w + (long)b * v_b0;
  const cplx* vb = vj + (long)b * v_b0;
  const cplx* ub = vjm1 + (long)b * v_b0;
  real acc = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nblk * blockDim.x) {
    cplx x = wb[i], a = vb[i];
    x.x = sj * fma(-alpha, a.x, x.x);
    x.y = sj * fma(-alpha, a.y, x.y);
    if (j > 0) {
      cplx u = ub[i];
      x.x = fma(-bprev, u.x, x.x);
      x.y = fma(-bprev, u.y, x.y);
    }
    wb[i] = x;
    acc = fma(x.x, x.x, acc);
    acc = fma(x.y, x.y, acc);
  }
  acc = block_sum(acc, sh);
  if (threadIdx.x == 0) part2[(long)b * nblk + blockIdx.x] = acc;
}

// x *= scale[b]
__global__ __launch_bounds__(256) void scale_kernel(cplx* __restrict__ x, long x_b0, long n, const real* scale, const int* ids,
                                                   const int* active) {
  int b = blockIdx.y;
  if (ids) b = ids[b];
  if (active && active[b] == 0) return;
  const real s = scale[b];
  cplx* xb = x + (long)b * x_b0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    cplx v = xb[i];
    v.x *= s;
    v.y *= s;
    xb[i] = v;
  }
}

// ------------------------------------------------------------------------------------------
// phi = exp(-i dt T_k) e_1 for the real symmetric tridiagonal T_k (k <= 64), one wavefront.
// Shift by the Gershgorin centre, split the remaining phase radius rho into s = ceil(rho)
// sub-steps and apply a degree-22 Taylor polynomial per sub-step (|arg| <= 1): unitary to
// ~1e-16 per sub-step, no eigen-decomposition.  The reference diagonalises T_k with LAPACK
// (matrix_exponential.py:147-163); both evaluate the same analytic function.
// Lane i holds entry i.  Returns phi_i in (pr, pi).
// ------------------------------------------------------------------------------------------
__device__ inline void tridiag_expm_e1(const real* alpha, const real* beta, int k, real dt, int lane, real& pr,
                                       real& pi, int ov_idx = -1, real ov_val = 0.0) {
  // alpha[ov_idx] may have been written by this very wavefront an instant ago: take it from a register
  const real a = (lane < k) ? ((lane == ov_idx) ? ov_val : alpha[lane]) : 0.0;
  const real bu = (lane < k - 1) ? beta[lane] : 0.0;                 // couples lane <-> lane+1
  const real bl = (lane >= 1 && lane < k) ? beta[lane - 1] : 0.0;    // couples lane <-> lane-1
  real lo = (lane < k) ? a - fabs(bu) - fabs(bl) : 1e300;
  real hi = (lane < k) ? a + fabs(bu) + fabs(bl) : -1e300;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = fmin(lo, __shfl_xor(lo, o, 64));
    hi = fmax(hi, __shfl_xor(hi, o, 64));
  }
  const real mu = 0.5 * (lo + hi);
  const real rho = fabs(dt) * 0.5 * (hi - lo);
  // a non-finite tridiagonal matrix (non-finite input state or Hamiltonian) must not turn into 2^31 sub-steps: its result is
  // non-finite whatever the count, and the caller's measurement / jump-weight check reports it
  int nsub = (rho < real(1e6)) ? (int)ceil(rho) : 1;
  if (nsub < 1) nsub = 1;
  const real h = dt / nsub;
  const real as = a - mu;
  real xr = (lane == 0) ? 1.0 : 0.0, xi = 0.0;
  for (int s = 0; s < nsub; ++s) {
    real tr = xr, ti = xi, sr = xr, si = xi;
    for (int n = 1; n <= 22; ++n) {
      // t <- (-i h / n) * T t
      real ur = __shfl_up(tr, 1, 64), ui = __shfl_up(ti, 1, 64);
      real dr = __shfl_down(tr, 1, 64), di = __shfl_down(ti, 1, 64);
      if (lane == 0) { ur = 0.0; ui = 0.0; }
      real yr = as * tr + bl * ur + bu * dr;
      real yi = as * ti + bl * ui + bu * di;
      const real c = h / n;
      tr = c * yi;   // (-i)(yr + i yi) = yi - i yr
      ti = -c * yr;
      sr += tr;
      si += ti;
    }
    xr = sr;
    xi = si;
  }
  // global phase exp(-i dt mu)
  real sn, cs;
  tjm_sincos(-dt * mu, &sn, &cs);
  pr = xr * cs - xi * sn;
  pi = xr * sn + xi * cs;
  if (lane >= k) { pr = 0.0; pi = 0.0; }
}

__global__ __launch_bounds__(64) void tridiag_expm_test_kernel(const real* alpha, const real* beta, int k, real dt,
                                                              real* out) {
  real pr, pi;
  tridiag_expm_e1(alpha, beta, k, dt, threadIdx.x, pr, pi);
  if ((int)threadIdx.x < k) { out[2 * threadIdx.x] = pr; out[2 * threadIdx.x + 1] = pi; }
}

int launch_tridiag_expm_test(const real* alpha, const real* beta, int k, real dt, real* out, hipStream_t s) {
  if (k < 1 || k > 64) return TJM_ERR_ARG;
  hipLaunchKernelGGL(tridiag_expm_test_kernel, dim3(1), dim3(64), 0, s, alpha, beta, k, dt, out);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

// ------------------------------------------------------------------------------------------
// Small bonds: the whole expm_krylov of one block in one kernel.  One workgroup per trajectory; the block (<= 256 entries), both
// environment slices, the MPO matrix and the two intermediates of the three-stage contraction live in LDS; the recurrence, the
// breakdown / adaptive-stop tests of lanczos_finalize_kernel and the final combination follow in the same launch.  Replaces
// about ten launches and one host synchronisation per Lanczos iteration.
// ------------------------------------------------------------------------------------------
constexpr int KS_MAXN = 1024;          // P * ca * cb: capacity 16 for a two-site block
constexpr size_t KS_MAX_LDS = 150 * 1024;  // bytes of dynamic LDS (one workgroup per CU at the upper end)
constexpr int KS_THROUGHPUT_N = 256;   // up to this block size the kernel also wins with thousands of trajectories in flight
constexpr int KS_LATENCY_BATCH = 1024; // larger blocks: only while the batch leaves the GEMM path launch-bound

static size_t krylov_small_lds(int P, int ca, int cb, int Dl, int Dr) {
  const size_t n = (size_t)P * ca * cb + (size_t)P * ca * cb * (Dl + Dr) + (size_t)ca * Dl * ca + (size_t)cb * Dr * cb + (size_t)P * Dl * P * Dr;
  return n * sizeof(cplx);
}

__global__ __launch_bounds__(256) void krylov_site_small_kernel(SmallKrylovDesc p) {
  extern __shared__ real ks_smem[];
  __shared__ cplx sCoef[64];
  __shared__ real sAl[64], sBe[64], sh[4];
  __shared__ int sDone, sK;
  int b = blockIdx.x;
  if (p.ids) b = p.ids[b];
  const int tid = threadIdx.x;
  const int P = p.P, ca = p.ca, cb = p.cb, Dl = p.Dl, Dr = p.Dr, m = p.mmax;
  const int N = P * ca * cb, nT1 = P * ca * Dr * cb, nT2 = P * ca * Dl * cb;
  cplx* sX = reinterpret_cast<cplx*>(ks_smem);
  cplx* sT1 = sX + N;
  cplx* sY = sT1;  // the product vector overwrites the first intermediate, which is dead by then (nT1 >= N)
  cplx* sT2 = sT1 + nT1;
  cplx* sL = sT2 + nT2;
  cplx* sR = sL + ca * Dl * ca;
  cplx* sW = sR + cb * Dr * cb;
  cplx* __restrict__ Vb = p.V + (long)b * p.v_b0;
  for (int e = tid; e < ca * Dl * ca; e += 256) sL[e] = p.Lenv[(long)b * p.l_b0 + e];
  for (int e = tid; e < cb * Dr * cb; e += 256) sR[e] = p.Renv[(long)b * p.r_b0 + e];
  for (int e = tid; e < P * Dl * P * Dr; e += 256) sW[e] = p.Wm[e];
  real acc = 0.0;
  for (int e = tid; e < N; e += 256) {
    const cplx v = Vb[e];
    sX[e] = v;
    acc = fma(v.x, v.x, fma(v.y, v.y, acc));
  }
  const real nrm = sqrt(block_sum(acc, sh));
  cplx* __restrict__ ob = p.out + (long)b * p.out_b0;
  auto out_index = [&](int e) -> long {
    long i3 = e % p.n3, r = e / p.n3;
    long i2 = r % p.n2;
    r /= p.n2;
    long i1 = r % p.n1, i0 = r / p.n1;
    return i0 * p.o0 + i1 * p.o1 + i2 * p.o2 + i3;
  };
  if (nrm == 0.0) {  // the result is the zero vector
    for (int e = tid; e < N; e += 256) ob[out_index(e)] = cplx{0.0, 0.0};
    return;
  }
  const real inv0 = 1.0 / nrm;
  for (int e = tid; e < N; e += 256) {
    cplx v = sX[e];
    v.x *= inv0; v.y *= inv0;
    sX[e] = v;
    Vb[e] = v;
  }
  const real eps_cut = tjm_breakdown_cut(p.nloc[b]);
  const int na = p.chi_l ? min(p.chi_l[(long)b * p.chi_stride], ca) : ca;
  const int nb = p.chi_r ? min(p.chi_r[(long)b * p.chi_stride], cb) : cb;
  real bprev = 0.0;
  int kfinal = 0;
  __syncthreads();
  for (int j = 0; j < m; ++j) {
    // Every loop runs over the ACTUAL bonds (na, nb) of this trajectory: entries beyond them are exactly zero in x, and the
    // intermediates are only read at the positions that were written.
    // T1[(p,a),(r,B)] = sum_b x[(p,a),b] R[b,(r,B)]
    for (int e = tid; e < P * na * Dr * nb; e += 256) {
      const int Bc = e % nb, r = (e / nb) % Dr, a = (e / (nb * Dr)) % na, pp = e / (nb * Dr * na);
      const int pa = pp * ca + a;
      cplx t{0.0, 0.0};
      for (int q = 0; q < nb; ++q) cfma(t, sX[pa * cb + q], sR[(q * Dr + r) * cb + Bc]);
      sT1[(pa * Dr + r) * cb + Bc] = t;
    }
    __syncthreads();
    // T2[o][a][l][B] = sum_{p,r} W[(o,l),(p,r)] T1[p][a][r][B]
    for (int e = tid; e < P * na * Dl * nb; e += 256) {
      const int Bc = e % nb, l = (e / nb) % Dl, a = (e / (nb * Dl)) % na, o = e / (nb * Dl * na);
      cplx t{0.0, 0.0};
      for (int pp = 0; pp < P; ++pp)
        for (int r = 0; r < Dr; ++r) cfma(t, sW[(o * Dl + l) * (P * Dr) + pp * Dr + r], sT1[((pp * ca + a) * Dr + r) * cb + Bc]);
      sT2[((o * ca + a) * Dl + l) * cb + Bc] = t;
    }
    __syncthreads();
    // y[o][A][B] = sum_{(a,l)} L[(a,l)][A] T2[o][(a,l)][B] ;  alpha = Re <x, y>
    real dot = 0.0;
    for (int e = tid; e < N; e += 256) {
      const int Bc = e % cb, A = (e / cb) % ca, o = e / (cb * ca);
      cplx t{0.0, 0.0};
      if (A < na && Bc < nb)
        for (int a = 0; a < na; ++a)
          for (int l = 0; l < Dl; ++l) cfma(t, sL[(a * Dl + l) * ca + A], sT2[((o * ca + a) * Dl + l) * cb + Bc]);
      sY[e] = t;
      const cplx x = sX[e];
      dot = fma(x.x, t.x, fma(x.y, t.y, dot));
    }
    const real alpha = block_sum(dot, sh);
    // w = y - alpha x - beta_{j-1} x_{j-1}
    real s2 = 0.0;
    for (int e = tid; e < N; e += 256) {
      cplx w = sY[e];
      const cplx x = sX[e];
      w.x = fma(-alpha, x.x, w.x);
      w.y = fma(-alpha, x.y, w.y);
      if (j > 0) {
        const cplx u = Vb[(long)(j - 1) * p.v_ld + e];
        w.x = fma(-bprev, u.x, w.x);
        w.y = fma(-bprev, u.y, w.y);
      }
      sY[e] = w;
      s2 = fma(w.x, w.x, fma(w.y, w.y, s2));
    }
    const real bj = sqrt(block_sum(s2, sh));
    if (tid == 0) {
      sAl[j] = alpha;
      if (j < m - 1) sBe[j] = bj;
      sDone = 0;
    }
    __syncthreads();
    const int k = j + 1;
    if (tid < 64) {  // breakdown and adaptive stop (lanczos_finalize_kernel)
      real pr = 0.0, pi = 0.0;
      bool done = false;
      if (j < m - 1 && (bj < eps_cut * tjm_breakdown_scale(j == 0 ? alpha : sAl[0], j == 0 ? bj : sBe[0]) || !(bj > real(0.0)))) {  // beta = 0 (H v = 0: zero Hamiltonian, dissipation-only model) is a breakdown whatever the scale says: 1 / beta must never be formed
        tridiag_expm_e1(sAl, sBe, k, p.dt, tid, pr, pi);
        done = true;
      } else if (j >= 1 || j == m - 1) {
        tridiag_expm_e1(sAl, sBe, k, p.dt, tid, pr, pi);
        if (j == m - 1) done = true;
        else {
          const real lr = __shfl(pr, k - 1, 64), li = __shfl(pi, k - 1, 64);
          done = (bj * sqrt(lr * lr + li * li) < p.tol);
        }
      }
      if (done) {
        if (tid < k) sCoef[tid] = cplx{pr * nrm, pi * nrm};
        if (tid == 0) { sDone = 1; sK = k; }
      }
    }
    __syncthreads();
    if (sDone) { kfinal = sK; break; }
    const real invb = 1.0 / bj;
    for (int e = tid; e < N; e += 256) {
      cplx w = sY[e];
      w.x *= invb; w.y *= invb;
      sX[e] = w;
      Vb[(long)(j + 1) * p.v_ld + e] = w;
    }
    bprev = bj;
    __syncthreads();
  }
  if (tid == 0 && p.matvecs) atomicAdd(p.matvecs, (unsigned long long)kfinal);
  // out = sum_{j < kfinal} coef_j V_j
  for (int e = tid; e < N; e += 256) {
    cplx t{0.0, 0.0};
    for (int j = 0; j < kfinal; ++j) cfma(t, sCoef[j], Vb[(long)j * p.v_ld + e]);
    ob[out_index(e)] = t;
  }
}

bool krylov_small_fits(int P, int ca, int cb, int Dl, int Dr, int mmax, int nb0) {
  static const bool off = getenv("TJM_NO_SMALL_KRYLOV") != nullptr;
  if (off || mmax > 63) return false;
  const int N = P * ca * cb;
  if (N > KS_MAXN || krylov_small_lds(P, ca, cb, Dl, Dr) > KS_MAX_LDS) return false;
  return N <= KS_THROUGHPUT_N || nb0 <= KS_LATENCY_BATCH;
}

int launch_krylov_site_small(const SmallKrylovDesc& p, hipStream_t s) {
  if (p.nb0 <= 0) return TJM_OK;
  static std::atomic<bool> attr{false};  // several engines of one process call this from their own host threads
  if (!attr.load(std::memory_order_acquire)) {
    TJM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(krylov_site_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)KS_MAX_LDS));
    attr.store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL(krylov_site_small_kernel, dim3(p.nb0), dim3(256), krylov_small_lds(p.P, p.ca, p.cb, p.Dl, p.Dr), s, p);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

// Lanczos start: vnorm = sqrt(sum part), status, first scale
__global__ __launch_bounds__(64) void lanczos_init_kernel(KrylovState ks, const real* part, int nblk, int nb, const int* ids) {
  int b = blockIdx.x;
  if (ids) b = ids[b];
  if (threadIdx.x != 0) return;
  real s = 0.0;
  for (int i = 0; i < nblk; ++i) s += part[(long)b * nblk + i];
  const real nrm = sqrt(s);
  ks.vnorm[b] = nrm;
  if (nrm == 0.0) {
    ks.status[b] = 0;  // finished: result is the zero vector
    ks.kfinal[b] = 1;
    ks.coef[(long)b * ks.mmax] = cplx{0.0, 0.0};
    ks.scale[b] = 0.0;
    ks.svec[(long)b * ks.mmax] = 0.0;
  } else {
    ks.status[b] = 1;
    ks.kfinal[b] = 0;
    ks.scale[b] = 1.0 / nrm;
    ks.svec[(long)b * ks.mmax] = 1.0 / nrm;
    atomicAdd(ks.n_active, 1);
  }
}

// One wavefront per trajectory: store alpha_j / beta_j, breakdown and adaptive-stop tests
// (matrix_exponential.py:100-163), coefficient vector on exit.
__global__ __launch_bounds__(64) void lanczos_finalize_kernel(KrylovState ks, const real* part1, const real* part2, int nblk,
                                                             int j, real dt, real tol, const int* nloc, const int* ids, int nblk1) {
  int b = blockIdx.x;
  if (ids) b = ids[b];
  if (ks.status[b] == 0) return;
  const int lane = threadIdx.x;
  const int m = ks.mmax;
  real* al = ks.alpha + (long)b * m;
  real* be = ks.beta + (long)b * m;
  real a = 0.0, s2 = 0.0;
  for (int i = 0; i < nblk1; ++i) a += part1[(long)b * nblk1 + i];
  for (int i = 0; i < nblk; ++i) s2 += part2[(long)b * nblk + i];
  const real bj = sqrt(s2);
  {
    const real sj = ks.svec[(long)b * m + j];  // part1 holds <V[j], H V[j]> of the unnormalised vector
    a *= sj * sj;
  }
  if (lane == 0) {
    al[j] = a;
    if (j < m - 1) be[j] = bj;
  }
  const real eps_cut = tjm_breakdown_cut(nloc[b]);
  bool done = false;
  const int k = j + 1;
  real pr = 0.0, pi = 0.0;
  if (j < m - 1 && (bj < eps_cut * tjm_breakdown_scale(j == 0 ? a : al[0], j == 0 ? bj : be[0]) || !(bj > real(0.0)))) {  // beta = 0: see krylov_site_small_kernel
    tridiag_expm_e1(al, be, k, dt, lane, pr, pi, j, a);
    done = true;
  } else if (j >= 1 || j == m - 1) {
    tridiag_expm_e1(al, be, k, dt, lane, pr, pi, j, a);
    if (j == m - 1) {
      done = true;
    } else {
      const real lr = __shfl(pr, k - 1, 64), li = __shfl(pi, k - 1, 64);
      done = (bj * sqrt(lr * lr + li * li) < tol);
    }
  }
  if (done) {
    const real nrm = ks.vnorm[b];
    if (lane < k) ks.coef[(long)b * m + lane] = cplx{pr * nrm, pi * nrm};
    if (lane == 0) {
      ks.status[b] = 0;
      ks.kfinal[b] = k;
    }
  } else if (lane == 0) {
    ks.scale[b] = 1.0 / bj;
    ks.svec[(long)b * m + j + 1] = 1.0 / bj;
    atomicAdd(ks.n_active, 1);
  }
}

// out = sum_{j<kfinal} coef_j V_j, with a 4-level output index permutation.
__global__ __launch_bounds__(256) void krylov_combine_kernel(const cplx* __restrict__ V, long v_b0, long v_ld, KrylovState ks,
                                                            cplx* __restrict__ out, long out_b0, int n1, int n2, int n3, long o0,
                                                            long o1, long o2, long n, const int* ids) {
  __shared__ cplx sc[64];
  int b = blockIdx.y;
  if (ids) b = ids[b];
  const int k = ks.kfinal[b];
  if (threadIdx.x < k) {  // the stored vectors are unnormalised: their scales go into the coefficients
    cplx c = ks.coef[(long)b * ks.mmax + threadIdx.x];
    const real sv = ks.svec[(long)b * ks.mmax + threadIdx.x];
    sc[threadIdx.x] = cplx{c.x * sv, c.y * sv};
  }
  __syncthreads();
  const cplx* Vb = V + (long)b * v_b0;
  cplx* ob = out + (long)b * out_b0;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    cplx acc{0.0, 0.0};
    for (int j = 0; j < k; ++j) cfma(acc, sc[j], Vb[(long)j * v_ld + e]);
    long i3 = e % n3, r = e / n3;
    long i2 = r % n2;
    r /= n2;
    long i1 = r % n1, i0 = r / n1;
    ob[i0 * o0 + i1 * o1 + i2 * o2 + i3] = acc;
  }
}

// ------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------
static inline int nblk_for(long n) {
  long nb = (n + 1023) / 1024;
  if (nb < 1) nb = 1;
  if (nb > TJM_MAX_PART) nb = TJM_MAX_PART;
  return (int)nb;
}

int launch_normsq_partial(const cplx* x, long x_b0, int n, real* part, int nb0, const int* ids, const int* active,
                          hipStream_t s, int* nblk_out) {
  const int nblk = nblk_for(n);
  *nblk_out = nblk;
  hipLaunchKernelGGL(normsq_partial_kernel, dim3(nblk, nb0), dim3(256), 0, s, x, x_b0, n, part, nblk, ids, active);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

int launch_dot_partial(const cplx* v, const cplx* w, long v_b0, long w_b0, int n, real* part, int nb0, const int* ids,
                       const int* active, hipStream_t s, int* nblk_out) {
  const int nblk = nblk_for(n);
  *nblk_out = nblk;
  hipLaunchKernelGGL(dot_partial_kernel, dim3(nblk, nb0), dim3(256), 0, s, v, w, v_b0, w_b0, n, part, nblk, ids, active);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

int launch_lanczos_axpy(cplx* w, const cplx* vj, const cplx* vjm1, long v_b0, int n, const real* part1, real* part2,
                        int nblk, const real* beta, int beta_ld, int j, int nb0, const int* ids, const int* active,
                        hipStream_t s, const real* svec, int nblk1) {
  hipLaunchKernelGGL(lanczos_axpy_kernel, dim3(nblk, nb0), dim3(256), 0, s, w, vj, vjm1, v_b0, n, part1, part2, nblk, beta,
                     beta_ld, j, ids, active, svec, nblk1);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

// ------------------------------------------------------------------------------------------
// Identity channels of the environments.  An MPO written as a finite-state machine carries "nothing has happened yet" in one bond
// index and "everything is done" in another (the identity rows / columns of W, mpo.py:326-406): the slice of a left environment
// in the former and of a right environment in the latter is <A|A> of the sites on that side, i.e. the identity matrix whenever those
// sites are isometric - which the sweep guarantees by construction.  env_identity_check_kernel CERTIFIES it for the environment
// handed to a Krylov call (first and last channel; every trajectory of the call): flags[0] / flags[1] are raised when the
// first / last channel of some trajectory differs from the identity on its actual bond by more than tol.  A certified channel lets
// heff_apply skip its third of the two GEMMs (the product with the identity is a copy).
// env layout [c][D][c]: env[(a * D + w) * c + A].
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void env_identity_check_kernel(const cplx* __restrict__ env, long b0, int c, int D, const int* chi, int chi_stride,
                                                                real tol, int* flags, const int* ids) {
  int b = blockIdx.x;
  if (ids) b = ids[b];
  const int n = chi ? min(chi[(long)b * chi_stride], c) : c;
  const cplx* e = env + (long)b * b0;
  int bad0 = 0, bad1 = 0;
  for (long t = threadIdx.x; t < (long)n * n; t += blockDim.x) {
    const int a = (int)(t / n), A = (int)(t - (long)a * n);
    const real want = (a == A) ? real(1.0) : real(0.0);
    const cplx v0 = e[((long)a * D) * c + A], v1 = e[((long)a * D + (D - 1)) * c + A];
    bad0 |= !(fabs(v0.x - want) <= tol && fabs(v0.y) <= tol);   // written so that a NaN counts as a mismatch
    bad1 |= !(fabs(v1.x - want) <= tol && fabs(v1.y) <= tol);
  }
  if (bad0) atomicOr(flags, 1);
  if (bad1) atomicOr(flags + 1, 1);
}

int launch_env_identity_check(const cplx* env, long b0, int c, int D, const int* chi, int chi_stride, real tol, int* flags, int nb0, const int* ids,
                              hipStream_t s) {
  if (nb0 <= 0) return TJM_OK;
  hipLaunchKernelGGL(env_identity_check_kernel, dim3(nb0), dim3(256), 0, s, env, b0, c, D, chi, chi_stride, tol, flags, ids);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

int launch_scale(cplx* x, long x_b0, long n, const real* scale, int nb0, const int* ids, const int* active, hipStream_t s) {
  int gx = (int)((n + 1023) / 1024);
  if (gx < 1) gx = 1;
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(scale_kernel, dim3(gx, nb0), dim3(256), 0, s, x, x_b0, n, scale, ids, active);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

int launch_lanczos_init(const KrylovState& ks, const real* part, int nblk, int nb0, const int* ids, hipStream_t s) {
  hipLaunchKernelGGL(lanczos_init_kernel, dim3(nb0), dim3(64), 0, s, ks, part, nblk, nb0, ids);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

int launch_lanczos_finalize(const KrylovState& ks, const real* part1, const real* part2, int nblk, int j, real dt,
                            real tol, const int* nloc, int nb0, const int* ids, hipStream_t s, int nblk1) {
  hipLaunchKernelGGL(lanczos_finalize_kernel, dim3(nb0), dim3(64), 0, s, ks, part1, part2, nblk, j, dt, tol, nloc, ids, nblk1);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

int launch_krylov_combine(const cplx* V, long v_b0, long v_ld, const KrylovState& ks, cplx* out, long out_b0, int n0, int n1,
                          int n2, int n3, long o0, long o1, long o2, int nb0, const int* ids, hipStream_t s) {
  const long n = (long)n0 * n1 * n2 * n3;
  int gx = (int)((n + 1023) / 1024);
  if (gx < 1) gx = 1;
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(krylov_combine_kernel, dim3(gx, nb0), dim3(256), 0, s, V, v_b0, v_ld, ks, out, out_b0, n1, n2, n3, o0, o1,
                     o2, n, ids);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

// ------------------------------------------------------------------------------------------
// small per-site kernels
// ------------------------------------------------------------------------------------------
// out[b] = sum |x_b|^2   (one block per trajectory)
__global__ __launch_bounds__(256) void normsq_kernel(const cplx* __restrict__ x, long x_b0, long n, real* out, const int* ids) {
  __shared__ real sh[4];
  int b = blockIdx.x;
  if (ids) b = ids[b];
  const cplx* xb = x + (long)b * x_b0;
  real acc = 0.0;
  for (long i = threadIdx.x; i < n; i += blockDim.x) {
    cplx v = xb[i];
    acc = fma(v.x, v.x, acc);
    acc = fma(v.y, v.y, acc);
  }
  acc = block_sum(acc, sh);
  if (threadIdx.x == 0) out[b] = acc;
}

int launch_normsq(const cplx* x, long x_b0, long n, real* out, int nb0, const int* ids, hipStream_t s) {
  hipLaunchKernelGGL(normsq_kernel, dim3(nb0), dim3(256), 0, s, x, x_b0, n, out, ids);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

// x[p][r] <- sum_q O_b[p][q] x[q][r]   (physical-leg operator, per-trajectory operator index)
// ops: table of d x d matrices; op_index[b] < 0 => identity (skip)
__global__ __launch_bounds__(256) void apply_local_kernel(cplx* __restrict__ x, long x_b0, int d, long rest, const cplx* ops,
                                                         const int* op_index, const int* ids) {
  int b = blockIdx.y;
  if (ids) b = ids[b];
  const int oi = op_index ? op_index[b] : 0;
  if (oi < 0) return;
  const cplx* O = ops + (long)oi * d * d;
  cplx* xb = x + (long)b * x_b0;
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < rest; r += (long)gridDim.x * blockDim.x) {
    cplx v[4], y[4];
    for (int q = 0; q < d; ++q) v[q] = xb[(long)q * rest + r];
    for (int p = 0; p < d; ++p) {
      cplx acc{0.0, 0.0};
      for (int q = 0; q < d; ++q) cfma(acc, O[p * d + q], v[q]);
      y[p] = acc;
    }
    for (int p = 0; p < d; ++p) xb[(long)p * rest + r] = y[p];
  }
}

int launch_apply_local(cplx* x, long x_b0, int d, long rest, const cplx* ops, const int* op_index, int nb0, const int* ids,
                       hipStream_t s) {
  if (d > 4) return TJM_ERR_NOT_IMPLEMENTED;
  int gx = (int)((rest + 255) / 256);
  if (gx > 128) gx = 128;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(apply_local_kernel, dim3(gx, nb0), dim3(256), 0, s, x, x_b0, d, rest, ops, op_index, ids);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

// E[b][i][a][i] = 1 for i < n (boundary environment, primitives.py:161-168)
__global__ void identity_env_kernel(cplx* E, long e_b0, int n, int D, int nb0) {
  int b = blockIdx.x;
  for (int t = threadIdx.x; t < n * D * n; t += blockDim.x) {
    int i = t / (D * n), r = t % (D * n), k = r % n;
    E[(long)b * e_b0 + t] = cplx{(i == k) ? real(1) : real(0), 0.0};
  }
}

int launch_identity_env(cplx* E, long e_b0, int n, int D, int nb0, hipStream_t s) {
  hipLaunchKernelGGL(identity_env_kernel, dim3(nb0), dim3(64), 0, s, E, e_b0, n, D, nb0);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

// M[b][p][q] = <x_p | y_q> = sum_r conj(x[p][r]) y[q][r]   (d x d physical overlap matrix)
__global__ __launch_bounds__(256) void phys_overlap_kernel(const cplx* __restrict__ x, const cplx* __restrict__ y, long x_b0,
                                                          long y_b0, int d, long rest, cplx* M, const int* ids) {
  __shared__ real sh[4];
  int b = blockIdx.x;
  if (ids) b = ids[b];
  const cplx* xb = x + (long)b * x_b0;
  const cplx* yb = y + (long)b * y_b0;
  for (int p = 0; p < d; ++p)
    for (int q = 0; q < d; ++q) {
      real ar = 0.0, ai = 0.0;
      for (long r = threadIdx.x; r < rest; r += blockDim.x) {
        cplx a = xb[(long)p * rest + r], c = yb[(long)q * rest + r];
        ar += a.x * c.x + a.y * c.y;
        ai += a.x * c.y - a.y * c.x;
      }
      ar = block_sum(ar, sh);
      ai = block_sum(ai, sh);
      if (threadIdx.x == 0) M[((long)b * d + p) * d + q] = cplx{ar, ai};
    }
}

int launch_phys_overlap(const cplx* x, const cplx* y, long x_b0, long y_b0, int d, long rest, cplx* M, int nb0, const int* ids,
                        hipStream_t s) {
  hipLaunchKernelGGL(phys_overlap_kernel, dim3(nb0), dim3(256), 0, s, x, y, x_b0, y_b0, d, rest, M, ids);
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

}  // namespace tjm
