// How many shader cycles does one v_mfma_f64_16x16x4_f64 cost in a bare loop (no memory traffic), and what clock does the chip hold
// meanwhile?  If a loop of independent MFMAs takes ~64 cycles per instruction and wavefront, the 48 TFLOP/s of profiles/r03/pipe_probe.json
// (61 % of the quoted 78.6) are the CLOCK under fp64 matrix load; if it takes ~105, they are issue.
// Build + run (GPU box): hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_cycles_probe.hip -o /tmp/mfma_cycles_probe && /tmp/mfma_cycles_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC, int VARIANT>
__global__ __launch_bounds__(256) void probe(double* out, long long* cyc, int iters, double seed) {
  d4 m[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) m[i] = d4{0, 0, 0, 0};
  const double x = seed * 1.0000001 + threadIdx.x * 1e-9, y = 0.999999;
  double av[4], bv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { av[i] = x + i * 0.125 * (threadIdx.x & 3); bv[i] = y - i * 0.0625 * (threadIdx.x & 7); }
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (VARIANT == 0) m[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, m[i], 0, 0, 0);
      else if (VARIANT == 1) {
        double r = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, m[i][0], 0, 0, 0);
        m[i][0] = r;
      } else if (VARIANT == 2) {  // a different A and B register for every instruction (as in a GEMM tile)
        double r = __builtin_amdgcn_mfma_f64_4x4x4f64(av[i & 3], bv[(i >> 2) & 3], m[i][0], 0, 0, 0);
        m[i][0] = r;
      } else {                    // ... and the B operand rotated inside its row of 16 lanes first (DPP row_ror:4 on both halves)
        const double yy = bv[i & 3];
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(yy), 0x124, 0xF, 0xF, true);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(yy), 0x124, 0xF, 0xF, true);
        bv[i & 3] = __hiloint2double(hi, lo);
        double r = __builtin_amdgcn_mfma_f64_4x4x4f64(av[(i >> 2) & 3], bv[i & 3], m[i][0], 0, 0, 0);
        m[i][0] = r;
      }
    }
  }
  const long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += m[i][0] + m[i][1] + m[i][2] + m[i][3];
  if (s == 12345.678) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC, int VARIANT>
static void run(const char* name, int blocks, int iters) {
  double* out; long long* cyc;
  hipMalloc(&out, 64); hipMalloc(&cyc, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<NACC, VARIANT>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters / 10, 1.5);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((probe<NACC, VARIANT>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.5);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  long long h = 0; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double n_mfma = (double)iters * NACC;               // per wavefront
  const double flops_per = VARIANT == 0 ? 2048.0 : 512.0;   // 16x16x4 x 2 ; 4 blocks of 4x4x4 x 2
  const double total = n_mfma * flops_per * blocks * 4.0;
  printf("{\"variant\": \"%s\", \"accumulators\": %d, \"blocks\": %d, \"waves_per_simd\": %.2f, \"ms\": %.3f, \"TFLOPs\": %.2f, \"cycles_per_mfma_per_wave\": %.1f, \"implied_clock_GHz\": %.3f}\n",
         name, NACC, blocks, blocks * 4.0 / 1024.0, ms, total / 1e9 / ms, (double)h / n_mfma, (double)h / (ms * 1e6));
  hipFree(out); hipFree(cyc);
}

int main() {
  run<8, 0>("v_mfma_f64_16x16x4_f64", 256, 40000);     // one wavefront per SIMD
  run<8, 0>("v_mfma_f64_16x16x4_f64", 512, 40000);     // two
  run<8, 0>("v_mfma_f64_16x16x4_f64", 1024, 20000);    // four
  run<8, 0>("v_mfma_f64_16x16x4_f64", 2048, 10000);    // eight
  run<1, 0>("v_mfma_f64_16x16x4_f64", 256, 100000);    // one dependent chain per wavefront
  run<2, 0>("v_mfma_f64_16x16x4_f64", 256, 100000);
  run<12, 0>("v_mfma_f64_16x16x4_f64", 512, 20000);    // the GEMM kernel's shape: 12 accumulators, two workgroups per CU
  run<8, 1>("v_mfma_f64_4x4x4_4b_f64", 256, 40000);
  run<8, 1>("v_mfma_f64_4x4x4_4b_f64", 1024, 20000);
  run<16, 2>("v_mfma_f64_4x4x4_4b_f64, 4 x 4 different operand registers", 256, 20000);
  run<16, 2>("v_mfma_f64_4x4x4_4b_f64, 4 x 4 different operand registers", 512, 20000);
  run<16, 3>("v_mfma_f64_4x4x4_4b_f64, B operand rotated by DPP before every instruction", 256, 20000);
  run<16, 3>("v_mfma_f64_4x4x4_4b_f64, B operand rotated by DPP before every instruction", 512, 20000);
  run<48, 2>("v_mfma_f64_4x4x4_4b_f64, 48 accumulators (the GEMM tile)", 512, 5000);
  return 0;
}
