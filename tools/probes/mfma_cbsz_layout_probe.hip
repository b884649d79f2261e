// Lane map of v_mfma_f64_4x4x4_4b_f64 under the broadcast controls: for every (cbsz, abid) a one-hot A (lane la) and a one-hot B
// (lane lb); prints, per (cbsz, abid), which (A block, B block) pairs meet and in which D block the product lands.
// Lanes: 16 k + 4 blk + i (A) / + j (B); D: 16 i + 4 blk + j  (profiles/r05/mfma_4x4x4_layout.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CBSZ, int ABID>
__global__ void probe(double* out) {
  const int la = blockIdx.x / 64, lb = blockIdx.x % 64, lane = threadIdx.x;
  const double a = (lane == la) ? 1.0 : 0.0, b = (lane == lb) ? 1.0 : 0.0;
  out[(long)blockIdx.x * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0);
}

template <int CBSZ, int ABID>
static void run(double* out, std::vector<double>& h) {
  hipLaunchKernelGGL((probe<CBSZ, ABID>), dim3(4096), dim3(64), 0, 0, out);
  if (hipMemcpy(h.data(), out, h.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return; }
  // meet[blkA][blkB] = D block (or -1), checked for consistency over i, j, k
  int meet[4][4]; bool ok = true;
  for (int x = 0; x < 4; ++x) for (int y = 0; y < 4; ++y) meet[x][y] = -1;
  for (int p = 0; p < 4096; ++p) {
    const int la = p / 64, lb = p % 64;
    const int ka = la / 16, ba = (la % 16) / 4, i = la % 4, kb = lb / 16, bb = (lb % 16) / 4, j = lb % 4;
    for (int l = 0; l < 64; ++l) {
      if (h[(long)p * 64 + l] == 0.0) continue;
      const int di = l / 16, dblk = (l % 16) / 4, dj = l % 4;
      if (ka != kb || di != i || dj != j) ok = false;
      if (meet[ba][bb] != -1 && meet[ba][bb] != dblk && meet[ba][bb] < 100) meet[ba][bb] = 100;  // several D blocks
      else if (meet[ba][bb] == -1) meet[ba][bb] = dblk;
    }
  }
  printf("cbsz %d abid %d: consistent %d; (A block, B block) -> D block:", CBSZ, ABID, (int)ok);
  for (int x = 0; x < 4; ++x) for (int y = 0; y < 4; ++y) if (meet[x][y] != -1) printf(" (%d,%d)->%d", x, y, meet[x][y]);
  printf("\n");
}

int main() {
  double* out;
  if (hipMalloc(&out, 4096 * 64 * sizeof(double)) != hipSuccess) return 1;
  std::vector<double> h(4096 * 64);
  run<0, 0>(out, h); run<0, 1>(out, h);
  run<1, 0>(out, h); run<1, 1>(out, h); run<1, 2>(out, h); run<1, 3>(out, h);
  run<2, 0>(out, h); run<2, 1>(out, h); run<2, 2>(out, h); run<2, 3>(out, h);
  run<3, 0>(out, h); run<3, 1>(out, h);
  return 0;
}
