// Does v_mfma_f64_4x4x4_4b_f64 honour the A-broadcast controls (cbsz:2 abid:b) on gfx950?  If it does, four instructions with
// abid = 0..3 on the SAME operand registers give a full 16 x 16 x 4 product: A[row = lane % 16][k = lane / 16],
// B[k = lane / 16][col = lane % 16] - the operand layout of v_mfma_f64_16x16x4_f64 - and D_b[lane] = C[4 b + lane / 16][lane % 16].
// Also: the rate of such groups of four (12 groups = the GEMM tile's 48 accumulators) and the effect of neg:[..] modifiers.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_cbsz_probe.hip -o tools/probes/bin/mfma_cbsz_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

__global__ void layout(double* out, const double* a, const double* b) {
  const int lane = threadIdx.x;
  const double av = a[lane], bv = b[lane];
  out[lane]       = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, 0.0, 2, 0, 0);
  out[64 + lane]  = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, 0.0, 2, 1, 0);
  out[128 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, 0.0, 2, 2, 0);
  out[192 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, 0.0, 2, 3, 0);
  out[256 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, 0.0, 2, 1, 1);   // neg A
  out[320 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, 0.0, 1, 2, 0);   // cbsz 1: pairs of blocks
}

template <int NG, bool BCAST>
__global__ __launch_bounds__(256) void rate(double* out, long long* cyc, int iters, double seed) {
  double acc[NG][4];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[g][b] = 0.0;
  double av[4], bv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { av[i] = seed + i * 0.125 * (threadIdx.x & 3); bv[i] = 0.999 - i * 0.0625 * (threadIdx.x & 7); }
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const double x = av[g & 3], y = bv[(g >> 2) & 3];
      if (BCAST) {
        acc[g][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, acc[g][0], 2, 0, 0);
        acc[g][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, acc[g][1], 2, 1, 0);
        acc[g][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, acc[g][2], 2, 2, 0);
        acc[g][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, acc[g][3], 2, 3, 0);
      } else {
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[g][b] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, acc[g][b], 0, 0, 0);
      }
    }
  }
  const long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int g = 0; g < NG; ++g) s += acc[g][0] + acc[g][1] + acc[g][2] + acc[g][3];
  if (s == 12345.678) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NG, bool BCAST>
static void run_rate(const char* name, int blocks, int iters) {
  double* out; long long* cyc;
  hipMalloc(&out, 64); hipMalloc(&cyc, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((rate<NG, BCAST>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters / 10, 1.5);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((rate<NG, BCAST>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.5);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  long long h = 0; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double n_mfma = (double)iters * NG * 4;
  printf("{\"variant\": \"%s\", \"groups\": %d, \"blocks\": %d, \"ms\": %.3f, \"TFLOPs\": %.2f, \"cycles_per_mfma_per_wave\": %.1f}\n",
         name, NG, blocks, ms, n_mfma * 512.0 * blocks * 4.0 / 1e9 / ms, (double)h / n_mfma);
}

int main() {
  std::vector<double> a(64), b(64), o(384);
  srand(7);
  for (int l = 0; l < 64; ++l) { a[l] = rand() / (double)RAND_MAX - 0.5; b[l] = rand() / (double)RAND_MAX - 0.5; }
  double *da, *db, *dout;
  hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dout, 384 * 8);
  hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, dout, da, db);
  if (hipMemcpy(o.data(), dout, 384 * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  // hypothesis: A[r][k] = a[16 k + r], B[k][c] = b[16 k + c]; D_b[lane] = C[4 b + lane / 16][lane % 16]
  double worst = 0, worst_neg = 0;
  for (int bb = 0; bb < 4; ++bb)
    for (int l = 0; l < 64; ++l) {
      const int r = 4 * bb + l / 16, c = l % 16;
      double ref = 0;
      for (int k = 0; k < 4; ++k) ref += a[16 * k + r] * b[16 * k + c];
      worst = fmax(worst, fabs(ref - o[64 * bb + l]));
      if (bb == 1) worst_neg = fmax(worst_neg, fabs(-ref - o[256 + l]));
    }
  // cbsz 1 abid 2?  (abid is taken modulo the group: blocks {0,1} read block 0, {2,3} read block 2 - report which hypothesis fits)
  double w_pair_a = 0, w_pair_b = 0;
  for (int l = 0; l < 64; ++l) {
    const int blk = (l % 16) / 4, i = l / 16, c = l % 16;
    const int srcA = (blk & 2) | 0, srcB = (blk & 2) | 1;
    double ra = 0, rb = 0;
    for (int k = 0; k < 4; ++k) { ra += a[16 * k + 4 * srcA + i] * b[16 * k + c]; rb += a[16 * k + 4 * srcB + i] * b[16 * k + c]; }
    w_pair_a = fmax(w_pair_a, fabs(ra - o[320 + l])); w_pair_b = fmax(w_pair_b, fabs(rb - o[320 + l]));
  }
  printf("{\"cbsz2_abid_full_16x16_product_max_err\": %.3e, \"neg_a_max_err\": %.3e, \"cbsz1_abid2_reads_even_block_err\": %.3e, \"cbsz1_abid2_reads_odd_block_err\": %.3e}\n",
         worst, worst_neg, w_pair_a, w_pair_b);
  run_rate<12, true>("4x4x4 cbsz:2 abid:0..3, 12 groups (48 accumulators)", 256, 5000);
  run_rate<12, true>("4x4x4 cbsz:2 abid:0..3, 12 groups (48 accumulators)", 512, 5000);
  run_rate<12, false>("4x4x4 no broadcast, 12 groups", 256, 5000);
  run_rate<12, false>("4x4x4 no broadcast, 12 groups", 512, 5000);
  run_rate<4, true>("4x4x4 cbsz:2 abid:0..3, 4 groups", 512, 10000);
  return 0;
}
