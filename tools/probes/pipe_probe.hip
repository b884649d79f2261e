// Do fp64 MFMA and fp64 vector FMA share an issue port / datapath on gfx950?  Three kernels with NO memory traffic:
//   valu: a chain-free stream of v_fma_f64 (8 independent accumulators per lane)
//   mfma: a stream of v_mfma_f64_16x16x4_f64 (8 independent accumulators per wavefront)
//   both: the two interleaved in ONE wavefront's instruction stream
// and the first two launched on two streams at once (co-resident workgroups).  If the pipes were independent, "both" would take
// max(valu, mfma) and the concurrent launch would take max of the two; if they share the double-precision datapath, the sum.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/pipe_probe.hip -o tools/probes/libpipe_probe.so
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 1 valu, 2 mfma, 3 both
__global__ __launch_bounds__(256) void probe_kernel(double* out, int iters, double seed) {
  double a[8];
  d4 m[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; m[i] = d4{0, 0, 0, 0}; }
  const double x = seed * 1.0000001, y = 0.999999;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE & 2) m[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, m[i], 0, 0, 0);
      if (MODE & 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) a[(i + r) & 7] = __builtin_fma(a[(i + r) & 7], y, x);  // 16 vector FMAs per MFMA: 64 issue cycles each side
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + m[i][0] + m[i][1] + m[i][2] + m[i][3];
  if (s == 12345.678) out[0] = s;
}

static float run(int mode_a, int mode_b, int blocks, int iters) {
  double* out;
  hipMalloc(&out, 64);
  hipStream_t s1, s2;
  hipStreamCreate(&s1);
  hipStreamCreate(&s2);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto launch = [&](int mode, hipStream_t s) {
    if (mode == 1) hipLaunchKernelGGL(probe_kernel<1>, dim3(blocks), dim3(256), 0, s, out, iters, 1.5);
    if (mode == 2) hipLaunchKernelGGL(probe_kernel<2>, dim3(blocks), dim3(256), 0, s, out, iters, 1.5);
    if (mode == 3) hipLaunchKernelGGL(probe_kernel<3>, dim3(blocks), dim3(256), 0, s, out, iters, 1.5);
  };
  launch(mode_a, s1);  // warm-up
  if (mode_b) launch(mode_b, s2);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipStreamWaitEvent(s1, e0, 0);
  hipStreamWaitEvent(s2, e0, 0);
  launch(mode_a, s1);
  if (mode_b) launch(mode_b, s2);
  hipEvent_t d1, d2;
  hipEventCreate(&d1);
  hipEventCreate(&d2);
  hipEventRecord(d1, s1);
  hipEventRecord(d2, s2);
  hipStreamWaitEvent(0, d1, 0);
  hipStreamWaitEvent(0, d2, 0);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipFree(out);
  return ms;
}

extern "C" int pipe_probe(int blocks, int iters, float* ms5) {
  ms5[0] = run(1, 0, blocks, iters);  // vector FMAs alone
  ms5[1] = run(2, 0, blocks, iters);  // MFMAs alone
  ms5[2] = run(3, 0, blocks, iters);  // interleaved in one wavefront
  ms5[3] = run(1, 2, blocks, iters);  // two kernels on two streams
  ms5[4] = run(1, 1, blocks, iters);  // control: the vector kernel twice on two streams
  return 0;
}
