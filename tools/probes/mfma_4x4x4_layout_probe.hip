// Operand / result lane map of v_mfma_f64_4x4x4_4b_f64 (__builtin_amdgcn_mfma_f64_4x4x4f64), found by experiment: wavefront (la, lb)
// sets A = 1 in lane la only and B = 1 in lane lb only; the lanes of D that come out non-zero tell which (block, row, k) lane la and
// which (block, k, column) lane lb hold.  Output: one line per (la, lb) with a hit: la lb -> lanes of D.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(double* out) {
  const int la = blockIdx.x / 64, lb = blockIdx.x % 64, lane = threadIdx.x;
  const double a = (lane == la) ? 1.0 : 0.0, b = (lane == lb) ? 1.0 : 0.0;
  const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  out[(long)blockIdx.x * 64 + lane] = d;
}

int main() {
  double* out;
  if (hipMalloc(&out, 4096 * 64 * sizeof(double)) != hipSuccess) return 1;
  hipLaunchKernelGGL(probe, dim3(4096), dim3(64), 0, 0, out);
  std::vector<double> h(4096 * 64);
  if (hipMemcpy(h.data(), out, h.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  for (int p = 0; p < 4096; ++p) {
    bool any = false;
    for (int l = 0; l < 64; ++l) any = any || h[(long)p * 64 + l] != 0.0;
    if (!any) continue;
    printf("%d %d ->", p / 64, p % 64);
    for (int l = 0; l < 64; ++l) if (h[(long)p * 64 + l] != 0.0) printf(" %d", l);
    printf("\n");
  }
  return 0;
}
