// Shared declarations for the MI355X (gfx950) Tensor-Jump-Method kernels.
// All device data is complex128 stored interleaved (re, im) - the NumPy / torch layout.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#include "../../include/tjm_hip.h"

namespace tjm {

struct cplx {
  double x, y;
};

__host__ __device__ inline cplx cmake(double a, double b) { return cplx{a, b}; }
__host__ __device__ inline cplx cadd(cplx a, cplx b) { return cplx{a.x + b.x, a.y + b.y}; }
__host__ __device__ inline cplx csub(cplx a, cplx b) { return cplx{a.x - b.x, a.y - b.y}; }
__host__ __device__ inline cplx cmul(cplx a, cplx b) { return cplx{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__host__ __device__ inline cplx cconj(cplx a) { return cplx{a.x, -a.y}; }
__host__ __device__ inline cplx cscale(cplx a, double s) { return cplx{a.x * s, a.y * s}; }
// acc += a*b
__host__ __device__ inline void cfma(cplx& acc, cplx a, cplx b) {
  acc.x = fma(a.x, b.x, acc.x);
  acc.x = fma(-a.y, b.y, acc.x);
  acc.y = fma(a.x, b.y, acc.y);
  acc.y = fma(a.y, b.x, acc.y);
}

typedef double d4 __attribute__((ext_vector_type(4)));

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
};

int launch_gemm(const GemmDesc& g, hipStream_t stream);

}  // namespace tjm
