// Strided batched complex128 GEMM on the gfx950 fp64 matrix cores.
//
// One kernel serves every bond x bond x phys contraction of the TDVP sweep (merge_two_site,
// the three tensordots of project_site / project_bond, the environment updates and the
// centre-shift absorptions; reference: core/methods/tdvp/primitives.py:77-226,
// core/methods/decompositions.py:87-102).  Layout permutations of the reference's
// np.tensordot / transpose chains are absorbed into operand strides, so no tensor is ever
// transposed in memory.
//
// Tiling: 256 threads = 4 wavefronts (2x2), block tile 64x64x16, every wave owns a 32x32
// sub-tile = 2x2 MFMA tiles of v_mfma_f64_16x16x4_f64.  A complex product is THREE real
// MFMAs into three accumulators (P += ArBr, Q += AiBi, S += (Ar+Ai)(Br+Bi); Re = P - Q, Im = S - P - Q, the 3M scheme);
// conjugation of either operand is a sign applied when its tile is stored to LDS.  Operands are staged through LDS as separate re/im planes,
// k-major with a row pitch of 80 doubles so that the two k-groups of a ds_read_b64 half-wave
// land in disjoint banks.  Global loads for tile k+1 are issued before the MFMAs of tile k.  With 96 accumulator
// registers the kernel takes about 205 registers: two workgroups share a CU and one's barriers hide behind the other's MFMAs.
#include <array>
#include <mutex>
#include <vector>

#include "tjm_common.h"
#include "tjm_kernels.h"

namespace tjm {

namespace {

constexpr int BM = 64, BN = 64, BK = 16;
constexpr int PITCH = 80;  // doubles per k-row in LDS (64 + 16: second k-group -> banks 32..63)
constexpr int KP = 18;     // TR: a k-contiguous operand is kept m-major in LDS, 16 k values + 2 per row (64 x 18 <= 16 x 80)

// TR: k-contiguous operands (the Gram products X^H X of the mixed split: both operands) are stored as they arrive - 16 consecutive k
// of one row per 16 lanes go to 16 consecutive LDS words - instead of k-major with the XOR swizzle, whose stores still collide
// two-way (k and k + 8 share a bank pair); the MFMA operand reads walk the rows with pitch 18 (16 rows x 2 k-groups of a half-wave
// hit 32 distinct bank pairs).
template <bool A_MCONTIG, bool B_NCONTIG, bool TR>
__global__ __launch_bounds__(256, 2) void zgemm_kernel(GemmDesc g) {
  __shared__ real sAll[4 * BK * PITCH];  // one array: the epilogue of a Hermitian product reuses it as a transposition buffer
  real* const sAr = sAll;
  real* const sAi = sAll + BK * PITCH;
  real* const sBr = sAll + 2 * BK * PITCH;
  real* const sBi = sAll + 3 * BK * PITCH;

  int z = blockIdx.y * gridDim.z + blockIdx.z;  // batches beyond the 65535 of one grid dimension spill into y (launch_gemm)
  if (z >= g.nb0 * g.nb1 * g.nb2) return;
  const int b2 = z % g.nb2;
  z /= g.nb2;
  const int b1 = z % g.nb1;
  int b0 = z / g.nb1;
  if (g.ids) b0 = g.ids[b0];
  if (g.active && g.active[b0] == 0) return;

  const int tiles_n = (g.N + BN - 1) / BN;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  if (g.hermitian && tm > tn) return;  // written by the workgroup of the mirror tile
  const int m0 = tm * BM, n0 = tn * BN;

  const cplx* __restrict__ Ab = g.A + (long)b0 * g.a_b0 + (long)b1 * g.a_b1 + (long)b2 * g.a_b2;
  const cplx* __restrict__ Bb = g.B + (long)b0 * g.b_b0 + (long)b1 * g.b_b1 + (long)b2 * g.b_b2;
  cplx* __restrict__ Cb = g.C + (long)b0 * g.c_b0 + (long)b1 * g.c_b1 + (long)b2 * g.c_b2;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;

  // per-thread staging coordinates: 4 elements of A (64x16) and 4 of B (16x64)
  int am[4], ak[4], bk[4], bn[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    int idx = tid + e * 256;
    if (A_MCONTIG) { am[e] = idx & 63; ak[e] = idx >> 6; }
    else           { ak[e] = idx & 15; am[e] = idx >> 4; }
    if (B_NCONTIG) { bn[e] = idx & 63; bk[e] = idx >> 6; }
    else           { bk[e] = idx & 15; bn[e] = idx >> 4; }
  }

  // three accumulators per MFMA tile (the 3M scheme): P = Ar Br, Q = Ai' Bi', S = (Ar + Ai')(Br + Bi'), where the primes carry the
  // conjugation signs (applied once, when the tile is stored to LDS); Re = P - Q and Im = S - P - Q in the epilogue.  A quarter fewer
  // MFMAs than the four-product form (measured: +4.2 % on the whole headline step); the operand sums are VALU additions issued in
  // the shadow of the matrix pipe.  The rounding error of Im is bounded relative to |A| |B| instead of |Im| (normwise stable).
  real4 accP[2][2], accQ[2][2], accS[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      accP[i][j] = real4{0, 0, 0, 0};
      accQ[i][j] = real4{0, 0, 0, 0};
      accS[i][j] = real4{0, 0, 0, 0};
    }
  const real sgnA = g.conjA ? -1.0 : 1.0, sgnB = g.conjB ? -1.0 : 1.0;

  const int ktiles = (g.K + BK - 1) / BK;
  const int total = ktiles * g.nks;
  cplx ra[4], rb[4];

  auto load_tile = [&](int it) {
    const int ks = it / ktiles;
    const int k0 = (it - ks * ktiles) * BK;
    const cplx* Ap = Ab + (long)ks * g.a_ks;
    const cplx* Bp = Bb + (long)ks * g.b_ks;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int m = m0 + am[e], k = k0 + ak[e];
      ra[e] = (m < g.M && k < g.K) ? Ap[(long)m * g.a_rs + (long)k * g.a_cs] : cplx{0.0, 0.0};
      int kk = k0 + bk[e], n = n0 + bn[e];
      rb[e] = (kk < g.K && n < g.N) ? Bp[(long)kk * g.b_rs + (long)n * g.b_cs] : cplx{0.0, 0.0};
    }
  };
  // k-contiguous operands are fetched with 16 consecutive k per 16 lanes (256-byte runs); stored k-major that would put the
  // 16 lanes on two bank pairs (pitch 80: bank = 16 (k & 1) + column), an 8-way conflict.  XOR-ing the column with
  // 4 ((k >> 1) & 3) spreads them over 8 bank pairs and stays inside the 16-column group every MFMA operand read covers, so the
  // reads remain conflict-free.
  auto swz = [](int k) { return 4 * ((k >> 1) & 3); };
  auto store_tile = [&]() {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ia = A_MCONTIG ? ak[e] * PITCH + am[e] : TR ? am[e] * KP + ak[e] : ak[e] * PITCH + (am[e] ^ swz(ak[e]));
      const int ib = B_NCONTIG ? bk[e] * PITCH + bn[e] : TR ? bn[e] * KP + bk[e] : bk[e] * PITCH + (bn[e] ^ swz(bk[e]));
      sAr[ia] = ra[e].x;
      sAi[ia] = sgnA * ra[e].y;
      sBr[ib] = rb[e].x;
      sBi[ib] = sgnB * rb[e].y;
    }
  };

  load_tile(0);
  const int li = lane & 15, lk = lane >> 4;
  for (int it = 0; it < total; ++it) {
    __syncthreads();  // previous tile's reads are done
    store_tile();
    __syncthreads();
    if (it + 1 < total) load_tile(it + 1);  // in flight during the MFMAs below
#pragma unroll
    for (int s = 0; s < BK / 4; ++s) {
      const int krow = (4 * s + lk) * PITCH;
      const int la = A_MCONTIG ? li : (li ^ swz(4 * s + lk));
      const int lb = B_NCONTIG ? li : (li ^ swz(4 * s + lk));
      real ar[2], ai[2], br[2], bi[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ia = (!A_MCONTIG && TR) ? (wm + 16 * i + li) * KP + 4 * s + lk : krow + wm + 16 * i + la;
        const int ib = (!B_NCONTIG && TR) ? (wn + 16 * i + li) * KP + 4 * s + lk : krow + wn + 16 * i + lb;
        ar[i] = sAr[ia];
        ai[i] = sAi[ia];
        br[i] = sBr[ib];
        bi[i] = sBi[ib];
      }
      // twelve independent accumulators: none is touched again before eleven other MFMAs
      real as[2], bs[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) { as[i] = ar[i] + ai[i]; bs[i] = br[i] + bi[i]; }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) accP[i][j] = TJM_MFMA(ar[i], br[j], accP[i][j]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) accQ[i][j] = TJM_MFMA(ai[i], bi[j], accQ[i][j]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) accS[i][j] = TJM_MFMA(as[i], bs[j], accS[i][j]);
    }
  }

  // epilogue
  const cplx* __restrict__ Db = g.dot_part ? g.dot_with + (long)b0 * g.c_b0 + (long)b1 * g.c_b1 + (long)b2 * g.c_b2 : nullptr;
  real dot_acc = 0.0;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int m = m0 + wm + 16 * i + TJM_ACC_ROW(lane, r);
        int n = n0 + wn + 16 * j + li;
        if (m < g.M && n < g.N) {
          cplx v;
          v.x = accP[i][j][r] - accQ[i][j][r];
          v.y = accS[i][j][r] - accP[i][j][r] - accQ[i][j][r];
          if (g.accumulate != 0) {
            const cplx old = Cb[(long)m * g.c_rs + n];
            v.x = (g.accumulate > 0) ? old.x + v.x : old.x - v.x;
            v.y = (g.accumulate > 0) ? old.y + v.y : old.y - v.y;
          }
          Cb[(long)m * g.c_rs + n] = v;
          if (Db) {
            const cplx u = Db[(long)m * g.c_rs + n];
            dot_acc = fma(u.x, v.x, dot_acc);
            dot_acc = fma(u.y, v.y, dot_acc);
          }
          if (g.hermitian == 2 && tm != tn) Cb[(long)n * g.c_rs + m] = cplx{v.x, -v.y};  // (TJM_GEMM_DIRECT_MIRROR: the form of rounds 1 - 4)
        }
      }
  if (g.dot_part) {  // the workgroup's share of Re <dot_with, C>: lanes, then wavefronts, in a fixed order
    __syncthreads();  // (the operand tiles are done with)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dot_acc += __shfl_down(dot_acc, o, 64);
    if (lane == 0) sAll[wave] = dot_acc;
    __syncthreads();
    if (tid == 0)
      g.dot_part[(long)b0 * g.dot_ld + (long)(b1 * g.nb2 + b2) * gridDim.x + blockIdx.x] = (sAll[0] + sAll[1]) + (sAll[2] + sAll[3]);
  }
  // Hermitian product: the mirror tile (tn, tm) = this tile's conjugate transpose.  Written straight from the accumulators every lane
  // would store 16 bytes at a stride of a whole row (rounds 1 - 4: the Gram products of the mixed split ran 25 % below the other
  // products); each wavefront turns its 16 x 32 halves over in LDS instead and writes rows of 16 consecutive elements.
  if (g.hermitian == 1 && tm != tn) {
    constexpr int TP = 17;                       // pitch of the transposed 32 x 16 block (complex elements)
    cplx* sT = reinterpret_cast<cplx*>(sAll) + wave * (32 * TP);   // 4 x 544 complex <= 2560
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __syncthreads();  // the operand tiles (first half) / the reads of the half before are done with
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          sT[(16 * j + li) * TP + TJM_ACC_ROW(lane, r)] =
              cplx{accP[i][j][r] - accQ[i][j][r], -(accS[i][j][r] - accP[i][j][r] - accQ[i][j][r])};
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int e = lane + 64 * q, nn = e >> 4, mm = e & 15;
        const int n = n0 + wn + nn, m = m0 + wm + 16 * i + mm;
        if (m < g.M && n < g.N) Cb[(long)n * g.c_rs + m] = sT[nn * TP + mm];
      }
    }
  }
}

#ifndef TJM_F32
// ------------------------------------------------------------------------------------------------------------------------------
// The same product on v_mfma_f64_4x4x4_4b_f64 (four independent 4 x 4 x 4 blocks per instruction: 512 flops every 16 cycles), for
// shapes made of whole tiles (M, N multiples of 64, K of 8) and outputs of up to 48 tiles.  (In a bare register loop the instruction
// reaches 75 TFLOP/s where v_mfma_f64_16x16x4_f64 stops at 48 - tools/probes/mfma_cycles_probe.hip -; inside a GEMM both occupy the
// pipe for 16 cycles per 512 flops - SQ_VALU_MFMA_BUSY_CYCLES, profiles/r05/gemm_pmc_K512.txt - and alone on the device the two
// kernels tie on the headline's shapes.  What this kernel adds is how it is fed and how it shares the device: no register staging,
// no LDS bank conflicts, the next tile's operands in flight behind the current tile's last products, an XCD-aware tile order,
// 178 - 190 registers: +2 % end to end, and the direct H_eff form rests on its per-term operand table: +3 %.)  The instruction multiplies block blk of A (rows 4 blk .. 4 blk + 3 of a 16 x 4 operand tile, lane
// 16 k + 4 blk + i) with block blk of B (lane 16 k + 4 blk + j) into D (lane 16 i + 4 blk + j), so a 16 x 16 output tile takes four
// instructions on the same A tile with B in four lane ARRANGEMENTS (column block (blk + r) & 3 in the lanes of block blk, r = 0..3).
//   * Operand tiles live in LDS in the instruction's own lane order - one 16 x 4 tile = 64 lanes x 16 bytes (re, im) = 1 KiB - filled by
//     global_load_lds_dwordx4 (one instruction per tile; the per-lane SOURCE address carries the operand's strides, so every layout of
//     A and B is served by the same code and no staging registers are needed).  A fragment read is one ds_read_b128 at lane x 16; the
//     arrangements of B are reads of the same tile at (lane & 48) | ((lane + 4 r) & 15).
//   * A wavefront owns 64 rows x 16 columns: 4 A tiles x 4 arrangements x the three real products of the 3M scheme = 48 instructions
//     and 48 accumulator registers (doubles) per k-group, fed by 8 ds_read_b128.  The four wavefronts share the A tiles.
//   * Two k-tiles of 16 in LDS (2 x 32 KiB): the loads of tile t + 1 are in flight during the products of tile t, one barrier per tile.
//   * Conjugation: the sums of the third product carry the signs (one fma each), Q = Ai Bi is kept raw and signed in the epilogue.
// ------------------------------------------------------------------------------------------------------------------------------
typedef double d2v __attribute__((ext_vector_type(2)));
#define TJM_MFMA4(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0)

// Persistent: the grid is two workgroups per CU, workgroup w takes tiles w, w + grid, ...; the loop runs over (tile, k-tile) pairs so
// that the first operand tiles of the NEXT output tile are in flight during the last products of the current one and the stores of
// an output tile drain behind the next tile's products (at K = 128 the per-tile fixed cost was a fifth of the kernel).
// KG = k-groups of 4 per staged k-tile: 4 (k-tiles of 16, 64 KiB of LDS, two workgroups per CU) or 2 (k-tiles of 8, 32 KiB, THREE workgroups
// per CU inside 168 registers: a barrier every 96 products instead of 192, but three wavefronts per SIMD to fill each other's stalls -
// measured: no gain, it serves the shapes whose K is a multiple of 8 only).
template <int ABL, int KG>  // ABL: timing ablations (wrong results): 1 no staging in the loop, 2 no barrier, 4 fragments of k-group 0 for all, 8 no operand sums
__global__ __launch_bounds__(256, KG == 2 ? 3 : 2) void zgemm4_kernel(GemmDesc g, int tiles_m, int tiles_n, long total_tiles, int xcd_map, unsigned long long* work_counter) {
  // [buffer][A tiles (row group, k-group) 16 x 64 | B tiles (column group, k-group) 16 x 64], then four doubles of the dot-product
  // epilogue (ONE array: a second LDS object beside a global_load_lds target costs a full wait per read)
  constexpr int SB = 512 * KG, HB = 256 * KG, WB = 64 * KG, BK4 = 4 * KG;  // doubles-pairs per buffer / offset of the B tiles / per wavefront; k per tile
  __shared__ d2v sm[2 * SB + 2];
  double* const red = reinterpret_cast<double*>(sm + 2 * SB);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // in a scalar register: the LDS targets of the loads are scalar arithmetic
  const int li = lane & 15, lk = lane >> 4;
  // tiles per batch entry: a Hermitian product enumerates its upper triangle only (a workgroup that walked all T x T positions met the
  // same position in every matrix - 64 slots per XCD, 16 positions - and those on a mirror position idled through the whole launch)
  const int per = (g.hermitian && g.hermitian != 3) ? tiles_m * (tiles_m + 1) / 2 : tiles_m * tiles_n;
  const int ktiles = g.K / BK4;
  const int total = ktiles * g.nks;

  // tile bookkeeping (all wave-uniform): `cur` is the tile being multiplied, `nxt` the one whose operands are being staged
  struct Tile { long t, pos; int m0, n0, b0, b1, b2; bool mirror; };
  auto decode = [&](long t, Tile& T) -> bool {  // false: nothing to do for tile t
    T.t = t;
    int z = (int)(t / per);
    int tt = (int)(t - (long)z * per);
    int tm, tn;
    if (g.hermitian == 3) {  // (TJM_GEMM_HERM_ALL: every position enumerated, mirror positions skipped - the first build)
      tm = tt / tiles_n;
      tn = tt - tm * tiles_n;
      if (tm > tn) return false;
    } else if (g.hermitian) {  // only the tiles on and above the diagonal are enumerated (per = T (T + 1) / 2): row tm holds T - tm of them
      tm = 0;
      while (tt >= tiles_n - tm) { tt -= tiles_n - tm; ++tm; }
      tn = tm + tt;
    } else {
      tm = tt / tiles_n;
      tn = tt - tm * tiles_n;
    }
    T.b2 = z % g.nb2;
    z /= g.nb2;
    T.b1 = z % g.nb1;
    T.b0 = z / g.nb1;
    if (g.ids) T.b0 = g.ids[T.b0];
    T.m0 = tm * BM;
    T.n0 = tn * BN;
    T.mirror = g.hermitian && tm != tn;
    if (g.active && g.active[T.b0] == 0) return false;
    return true;
  };
  // Position p of this workgroup's list -> tile id.  XCD-aware: workgroups b and b + 8 share an XCD (and its 4 MiB L2), so batch
  // entry z goes to the workgroups with b % 8 == z % 8 and the 64 of them walk its tiles together - an entry's operands (1 - 2 MB)
  // are then re-read from ONE L2; dealt flat, the tiles of ten entries are in flight on every XCD at once.  Speed only.
  // (The unit dealt to an XCD is a TRAJECTORY with all its inner batches: they share the operand without inner batch strides - the
  // environment blocks of an H_eff apply are read by the products of all d^2 physical index pairs.)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const long per_traj = (long)per * g.nb1 * g.nb2;
  auto tile_at = [&](long pos) -> long {
    if (!xcd_map) return blockIdx.x + pos * (long)gridDim.x;
    const long u = slot + pos * (long)nslots;
    const long zl = u / per_traj;
    return (zl * 8 + xcd) * per_traj + (u - zl * per_traj);
  };
  auto next_valid = [&](long pos, Tile& T) -> bool {
    for (;; ++pos) {
      const long t = tile_at(pos);
      if (t >= total_tiles) break;
      if (decode(t, T)) { T.pos = pos; return true; }
    }
    T.t = total_tiles;
    return false;
  };
  Tile cur, nxt;
  if (!next_valid(0, cur)) return;

  // Staging.  This wavefront fills the A tiles of row group `wave` and the B tiles of column group `wave`: per k-tile four pieces
  // (k-groups) of each, one global_load_lds_dwordx4 per piece.  `ap` / `bp` point at piece 0 of the k-tile to be staged next; the
  // per-lane part of the address is fixed, everything that moves is scalar.
  const long a_lane = (long)(16 * wave + li) * g.a_rs + (long)lk * g.a_cs;
  const long b_lane = (long)lk * g.b_rs + (long)(16 * wave + li) * g.b_cs;
  const long a_kg = 4 * g.a_cs, b_kg = 4 * g.b_rs;          // k-group step (elements)
  const long a_kt = BK4 * g.a_cs, b_kt = BK4 * g.b_rs;        // k-tile step
  const long a_wrap = g.a_ks - (long)ktiles * a_kt + a_kt;  // ... from the last k-tile of one k-split term to the first of the next
  const long b_wrap = g.b_ks - (long)ktiles * b_kt + b_kt;
  const cplx* ap;
  const cplx* bp;
  int kt_st = 0;  // k-tile (inside its k-split term) that ap / bp point at
  const cplx* bp0 = nullptr;  // (b_perm) the tile's B pointer without the block offset
  int ks_st = 0, st_b1 = 0;   // (b_perm) k-split term and inner batch of the k-tile that ap / bp point at
  auto set_sources = [&](const Tile& T) {
    ap = g.A + ((long)T.b0 * g.a_b0 + (long)T.b1 * g.a_b1 + (long)T.b2 * g.a_b2 + (long)T.m0 * g.a_rs) + a_lane;
    if (g.b_perm) {
      bp0 = g.B + ((long)T.b0 * g.b_b0 + (long)T.n0 * g.b_cs) + b_lane;
      bp = bp0 + (long)g.b_perm[T.b1] * g.b_perm_stride;
      st_b1 = T.b1;
    } else {
      bp = g.B + ((long)T.b0 * g.b_b0 + (long)T.b1 * g.b_b1 + (long)T.b2 * g.b_b2 + (long)T.n0 * g.b_cs) + b_lane;
    }
    kt_st = 0;
    ks_st = 0;
  };
  auto stage_piece = [&](int kg, int buf) {
    d2v* dst = sm + buf * SB + wave * WB + kg * 64;
    TJM_GLDS16(ap + kg * a_kg, dst);
    TJM_GLDS16(bp + kg * b_kg, dst + HB);
  };
  auto advance = [&]() {
    if (++kt_st == ktiles) {
      kt_st = 0;
      ap += a_wrap;
      ++ks_st;
      if (g.b_perm) bp = bp0 + (long)g.b_perm[(ks_st < g.nks ? ks_st : 0) * g.nb1 + st_b1] * g.b_perm_stride;  // (behind the tile's last term nothing is staged from it)
      else bp += b_wrap;
    } else { ap += a_kt; bp += b_kt; }
  };
  double accP[4][4], accQ[4][4], accS[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) { accP[i][r] = 0.0; accQ[i][r] = 0.0; accS[i][r] = 0.0; }
  const double sgnA = g.conjA ? -1.0 : 1.0, sgnB = g.conjB ? -1.0 : 1.0;
  const double sq = sgnA * sgnB;
  const int blk = (lane >> 2) & 3, lj = lane & 3;
  int boff[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) boff[r] = HB + wave * WB + ((lane & 48) | ((lane + 4 * r) & 15));

  int tiles_done = 0;  // (measurement: executed output tiles of this workgroup, added to work_counter at the end)
  set_sources(cur);
#pragma unroll
  for (int kg = 0; kg < KG; ++kg) stage_piece(kg, 0);
  advance();
  __syncthreads();
  int it = 0, buf = 0;
  long dot_slot = -1;
  // (coef) the factor of the k-split term being multiplied: applied to the A fragments (the environment block), skipped when it is 1
  int kt_c = 0, ks_c = 0;
  cplx cf = g.coef ? g.coef[cur.b1] : cplx{1.0, 0.0};
  for (;;) {
    // what is staged during this k-tile's products: the tile's next k-tile, or the first one of the workgroup's next tile
    const bool last = it + 1 == total;
    bool pending = true, have_next = true;
    if (last) {
      have_next = next_valid(cur.pos + 1, nxt);
      pending = have_next;
      if (have_next) set_sources(nxt);
    } else if (ABL & 1) pending = false;
    const d2v* sb = sm + buf * SB;
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
      d2v a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = sb[(i * KG + ((ABL & 4) ? 0 : kg)) * 64 + lane];
#pragma unroll
      for (int r = 0; r < 4; ++r) b[r] = sb[boff[r] + ((ABL & 4) ? 0 : kg) * 64];
      if (pending) stage_piece(kg, buf ^ 1);  // two loads per k-group: spread over the k-tile, they never queue up in front of the products
      if (g.coef && !(cf.x == 1.0 && cf.y == 0.0)) {
        if (cf.y == 0.0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) { a[i].x *= cf.x; a[i].y *= cf.x; }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const double re = cf.x * a[i].x - cf.y * a[i].y;
            a[i].y = fma(cf.x, a[i].y, cf.y * a[i].x);
            a[i].x = re;
          }
        }
      }
      double as[4], bs[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) as[i] = (ABL & 8) ? a[i].x : fma(sgnA, a[i].y, a[i].x);
#pragma unroll
      for (int r = 0; r < 4; ++r) bs[r] = (ABL & 8) ? b[r].y : fma(sgnB, b[r].y, b[r].x);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) accP[i][r] = TJM_MFMA4(a[i].x, b[r].x, accP[i][r]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) accQ[i][r] = TJM_MFMA4(a[i].y, b[r].y, accQ[i][r]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) accS[i][r] = TJM_MFMA4(as[i], bs[r], accS[i][r]);
    }
    if (pending) advance();
    if (g.coef && !last && ++kt_c == ktiles) {  // the next k-tile belongs to the tile's next k-split term
      kt_c = 0;
      ++ks_c;
      cf = g.coef[ks_c * g.nb1 + cur.b1];
    }
    if (last) {
      // epilogue of tile `cur`: lane l of accumulator (i, r) is row 16 i + 4 blk + (l >> 4), column 4 ((blk + r) & 3) + (l & 3).
      // (Issuing the stores behind the barrier below - its vmcnt(0) also counts stores - was measured: no gain, 64 more registers.)
      cplx* __restrict__ Cb = g.C + (long)cur.b0 * g.c_b0 + (long)cur.b1 * g.c_b1 + (long)cur.b2 * g.c_b2;
      const cplx* __restrict__ Db = g.dot_part ? g.dot_with + (long)cur.b0 * g.c_b0 + (long)cur.b1 * g.c_b1 + (long)cur.b2 * g.c_b2 : nullptr;
      double dot_acc = 0.0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = cur.m0 + 16 * i + 4 * blk + lk;
          const int n = cur.n0 + 16 * wave + 4 * ((blk + r) & 3) + lj;
          cplx v;
          v.x = accP[i][r] - sq * accQ[i][r];
          v.y = accS[i][r] - accP[i][r] - sq * accQ[i][r];
          accP[i][r] = 0.0; accQ[i][r] = 0.0; accS[i][r] = 0.0;
          if (g.accumulate != 0) {
            const cplx old = Cb[(long)m * g.c_rs + n];
            v.x = (g.accumulate > 0) ? old.x + v.x : old.x - v.x;
            v.y = (g.accumulate > 0) ? old.y + v.y : old.y - v.y;
          }
          Cb[(long)m * g.c_rs + n] = v;
          if (Db) {
            const cplx u = Db[(long)m * g.c_rs + n];
            dot_acc = fma(u.x, v.x, dot_acc);
            dot_acc = fma(u.y, v.y, dot_acc);
          }
          if (cur.mirror) Cb[(long)n * g.c_rs + m] = cplx{v.x, -v.y};
        }
      if (g.dot_part) {  // the workgroup's share of Re <dot_with, C>: lanes by shuffles, wavefronts through LDS in a fixed order
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dot_acc += __shfl_down(dot_acc, o, 64);
        if (lane == 0) red[wave] = dot_acc;
        const int z = (int)(cur.t / per);
        dot_slot = (long)cur.b0 * g.dot_ld + (long)(z % (g.nb1 * g.nb2)) * per + (cur.t - (long)z * per);
      }
      ++tiles_done;
      if (!have_next) break;
      cur = nxt;
      it = 0;
      if (g.coef) { kt_c = 0; ks_c = 0; cf = g.coef[cur.b1]; }
    } else {
      ++it;
    }
    buf ^= 1;
    if (!(ABL & 2)) __syncthreads();  // the staged loads have landed (vmcnt) and every wavefront is done with the buffer just multiplied
    if (dot_slot >= 0) {  // (the next tile's partial sums are written at least one barrier later)
      if (tid == 0) g.dot_part[dot_slot] = (red[0] + red[1]) + (red[2] + red[3]);
      dot_slot = -1;
    }
  }
  if (dot_slot >= 0) {
    __syncthreads();
    if (tid == 0) g.dot_part[dot_slot] = (red[0] + red[1]) + (red[2] + red[3]);
  }
  if (work_counter && tid == 0) atomicAdd(work_counter, (unsigned long long)tiles_done * (unsigned long long)((long)g.K * g.nks));
}
#endif  // !TJM_F32

// Row order of a wavefront's 16 rows in heff_stage12_kernel: chosen per arithmetic type so that the four accumulator registers of a
// lane hold every physical index p of its bond value(s) (TJM_ACC_ROW: fp64 register v = row (l >> 4) + 4 v, fp32 = row 4 (l >> 4) + v).
#ifdef TJM_F32
template <int P> __device__ constexpr int s12_row_p(int rr) { return rr % P; }        // a-major rows: rr = a_loc * P + p
template <int P> __device__ constexpr int s12_row_a(int rr) { return rr / P; }
template <int P> __device__ constexpr int s12_reg(int pi, int ai) { return (P == 4) ? pi : (2 * ai + pi); }
template <int P> __device__ inline int s12_aloc(int lk, int ai) { return (P == 4) ? lk : (2 * lk + ai); }
#else
template <int P> __device__ constexpr int s12_row_p(int rr) { return rr / (16 / P); }  // p-major rows: rr = p * (16 / P) + a_loc
template <int P> __device__ constexpr int s12_row_a(int rr) { return rr % (16 / P); }
template <int P> __device__ constexpr int s12_reg(int pi, int ai) { return (P == 4) ? pi : (2 * pi + ai); }
template <int P> __device__ inline int s12_aloc(int lk, int ai) { return (P == 4) ? lk : (lk + 4 * ai); }
#endif

// ------------------------------------------------------------------------------------------------------------------------------
// H_eff apply, stages 1 + 2 fused (HeffStage12Desc, tjm_common.h).  Same LDS staging and MFMA schedule as zgemm_kernel, but the
// block tile is laid out so that every lane ends up with ALL inputs of the MPO stage for its points in its own accumulators:
//   rows  (64): wavefront w owns 16 rows = P physical indices x (16 / P) bond values, p-major -> accumulator register v of lane l is
//               row (l >> 4) + 4 v = (p = v, a = l >> 4) for P = 4, (p = v / 2, a = (l >> 4) + 4 (v & 1)) for P = 2;
//   cols  (64): the NCH = Dr - 1 non-identity channels x (64 / NCH) bond values, channel-major -> MFMA column tile j is channel
//               j / (4 / NCH), and lane l holds column l & 15 of it.
// Every wavefront multiplies its 16 rows into all four column tiles (16 MFMAs per k-step, as the 2 x 2 arrangement).  The epilogue
// is mpo_apply_kernel's arithmetic on registers: 144 complex multiply-adds per point at P = 4, D = 3, with W staged in LDS; the
// identity input channel comes straight from x, the identity output channel goes straight into y.  The 2 + 2 MB per trajectory
// that T1 cost on its way through HBM (and the launch of the MPO stage) are gone.
// ------------------------------------------------------------------------------------------------------------------------------
template <int P, int NCH>
__global__ __launch_bounds__(256, 2) void heff_stage12_kernel(HeffStage12Desc d, int tiles, int xcd_map) {
  __shared__ real sAr[BK * PITCH];
  __shared__ real sAi[BK * PITCH];
  __shared__ real sBr[BK * PITCH];
  __shared__ real sBi[BK * PITCH];
  __shared__ cplx sW[4 * 6 * 4 * 6];  // (P Dl) x (P Dr) <= 24 x 24
  __shared__ unsigned sMask[4 * 6];   // per row of W: bit k set when entry k is non-zero (MPOs of local Hamiltonians are mostly zeros)
  constexpr int AT = 64 / P;     // bond values of the row tile
  constexpr int APW = 16 / P;    // ... per wavefront
  constexpr int BT = 64 / NCH;   // bond values of the column tile
  constexpr int CPT = 4 / NCH;   // MFMA column tiles per channel
  // XCD-aware order (1-D grid): workgroups w and w + 8 share an XCD and its L2, so the tiles of one trajectory - which all read the
  // same x (1 MB) and R (0.8 MB) - go to workgroups of equal w % 8; dealt in launch order, tile t of EVERY trajectory lands on XCD t % 8.
  int b0 = blockIdx.y, tile = blockIdx.x;
  if (xcd_map) {
    const int q = blockIdx.x >> 3;
    tile = q % tiles;
    b0 = (q / tiles) * 8 + (blockIdx.x & 7);
    if (b0 >= d.nb0) return;
  }
  if (d.ids) b0 = d.ids[b0];
  if (d.active && d.active[b0] == 0) return;
  const int ca = d.ca, cb = d.cb, Dl = d.Dl, Dr = d.Dr, rch = d.rch;
  const int tiles_b = (cb + BT - 1) / BT;
  const int a0 = (tile / tiles_b) * AT, B0 = (tile % tiles_b) * BT;
  const cplx* __restrict__ xb = d.x + (long)b0 * d.x_b0;
  const cplx* __restrict__ Rb = d.R + (long)b0 * d.r_b0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int ch_shift = (rch == 0) ? 1 : 0;  // channel c of the tile is channel c + ch_shift of R (rch is the first or the last one)
  for (int t = tid; t < P * Dl * P * Dr; t += 256) sW[t] = d.Wm[t];

  // staging coordinates: A (x) is k-contiguous, B (R) is n-contiguous
  int am[4], ak[4], bk[4], bn[4];
  long arow[4], bcol[4];
  bool aok[4], bok[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = tid + e * 256;
    ak[e] = idx & 15; am[e] = idx >> 4;
    bn[e] = idx & 63; bk[e] = idx >> 6;
    const int w = am[e] >> 4, rr = am[e] & 15;
    const int p = s12_row_p<P>(rr), a = a0 + w * APW + s12_row_a<P>(rr);
    aok[e] = a < ca;
    arow[e] = ((long)p * ca + a) * cb;
    const int c = bn[e] / BT, Bc = B0 + bn[e] % BT;
    bok[e] = Bc < cb;
    bcol[e] = (long)(c + ch_shift) * cb + Bc;
  }
  real4 accRe[4], accIm[4], accS[4];  // the three-product scheme of zgemm_kernel: P, Q, S in the loop, folded into Re / Im before the epilogue
#pragma unroll
  for (int j = 0; j < 4; ++j) { accRe[j] = real4{0, 0, 0, 0}; accIm[j] = real4{0, 0, 0, 0}; accS[j] = real4{0, 0, 0, 0}; }
  const int ktiles = (cb + BK - 1) / BK;
  cplx ra[4], rb[4];
  auto load_tile = [&](int it) {
    const int k0 = it * BK;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k0 + ak[e];
      ra[e] = (aok[e] && k < cb) ? xb[arow[e] + k] : cplx{0.0, 0.0};
      const int kk = k0 + bk[e];
      rb[e] = (bok[e] && kk < cb) ? Rb[(long)kk * Dr * cb + bcol[e]] : cplx{0.0, 0.0};
    }
  };
  auto swz = [](int k) { return 4 * ((k >> 1) & 3); };
  auto store_tile = [&]() {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int caL = am[e] ^ swz(ak[e]);
      sAr[ak[e] * PITCH + caL] = ra[e].x;
      sAi[ak[e] * PITCH + caL] = ra[e].y;
      sBr[bk[e] * PITCH + bn[e]] = rb[e].x;
      sBi[bk[e] * PITCH + bn[e]] = rb[e].y;
    }
  };
  load_tile(0);
  for (int it = 0; it < ktiles; ++it) {
    __syncthreads();
    store_tile();
    __syncthreads();
    if (it + 1 < ktiles) load_tile(it + 1);
#pragma unroll
    for (int s = 0; s < BK / 4; ++s) {
      const int krow = (4 * s + lk) * PITCH;
      const int la = li ^ swz(4 * s + lk);
      const real ar = sAr[krow + 16 * wave + la], ai = sAi[krow + 16 * wave + la];
      real br[4], bi[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { br[j] = sBr[krow + 16 * j + li]; bi[j] = sBi[krow + 16 * j + li]; }
      const real as = ar + ai;
      real bs[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bs[j] = br[j] + bi[j];
#pragma unroll
      for (int j = 0; j < 4; ++j) accRe[j] = TJM_MFMA(ar, br[j], accRe[j]);
#pragma unroll
      for (int j = 0; j < 4; ++j) accIm[j] = TJM_MFMA(ai, bi[j], accIm[j]);
#pragma unroll
      for (int j = 0; j < 4; ++j) accS[j] = TJM_MFMA(as, bs[j], accS[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const real pp = accRe[j][v], qq = accIm[j][v];
      accRe[j][v] = pp - qq;
      accIm[j][v] = accS[j][v] - pp - qq;
    }
  // ---- epilogue: the MPO stage on this lane's points.  One row of W (P Dr entries, broadcast reads from LDS) per output (po, bo),
  // used for all points of the lane; the zero entries of the row (83 % of them for a nearest-neighbour Pauli Hamiltonian) are
  // skipped by scalar branches on the row's bit mask.
  if (tid < P * Dl) {
    unsigned m = 0;
    bool real_row = true;
    for (int k = 0; k < P * Dr; ++k) {
      const cplx wv = sW[tid * (P * Dr) + k];
      if (wv.x != 0.0 || wv.y != 0.0) m |= 1u << k;
      real_row = real_row && wv.y == 0.0;
    }
    sMask[tid] = m | (real_row ? 0x80000000u : 0u);  // bit 31: every entry of the row is real (Pauli-sum Hamiltonians without Y)
  }
  __syncthreads();
  cplx* __restrict__ T2b = d.T2 + (long)b0 * d.t_b0;
  cplx* __restrict__ yb = d.y + (long)b0 * d.y_b0;
  const int nin = P * Dr;
  constexpr int NA = 4 / P;          // bond values of this lane: one for P = 4, two for P = 2
  constexpr int NPT = NA * CPT;      // points of this lane
  cplx xin[NPT][P];
  bool ok[NPT];
  long base[NPT];                    // ((0 * ca + a) * cb + Bc): offset of the point in a [P][ca][cb] tensor
#pragma unroll
  for (int ai_ = 0; ai_ < NA; ++ai_)
#pragma unroll
    for (int jj = 0; jj < CPT; ++jj) {
      const int pt = ai_ * CPT + jj;
      const int a = a0 + wave * APW + s12_aloc<P>(lk, ai_), Bc = B0 + 16 * jj + li;
      ok[pt] = a < ca && Bc < cb;
      base[pt] = (long)a * cb + Bc;
#pragma unroll
      for (int pi = 0; pi < P; ++pi) xin[pt][pi] = ok[pt] ? xb[(long)pi * ca * cb + base[pt]] : cplx{0.0, 0.0};
    }
  for (int bo = 0; bo < Dl; ++bo) {
    if (d.skip_t2 && bo != d.lch) continue;
#pragma unroll
    for (int po = 0; po < P; ++po) {
      const cplx* wrow = sW + (po * Dl + bo) * nin;
      const unsigned mask = __builtin_amdgcn_readfirstlane(sMask[po * Dl + bo]);
      cplx out[NPT];
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) out[pt] = cplx{0.0, 0.0};
#pragma unroll
      for (int pi = 0; pi < P; ++pi) {
        if (mask & (1u << (pi * Dr + rch))) {
          const cplx wx = wrow[pi * Dr + rch];
          if (mask & 0x80000000u) {
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) { out[pt].x = fma(wx.x, xin[pt][pi].x, out[pt].x); out[pt].y = fma(wx.x, xin[pt][pi].y, out[pt].y); }
          } else {
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) cfma(out[pt], wx, xin[pt][pi]);
          }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          if (!(mask & (1u << (pi * Dr + c + ch_shift)))) continue;
          const cplx wc = wrow[pi * Dr + c + ch_shift];
#pragma unroll
          for (int ai_ = 0; ai_ < NA; ++ai_)
#pragma unroll
            for (int jj = 0; jj < CPT; ++jj) {
              const int v = s12_reg<P>(pi, ai_);
              cplx t1;
              t1.x = accRe[c * CPT + jj][v];
              t1.y = accIm[c * CPT + jj][v];
              if (mask & 0x80000000u) {
                out[ai_ * CPT + jj].x = fma(wc.x, t1.x, out[ai_ * CPT + jj].x);
                out[ai_ * CPT + jj].y = fma(wc.x, t1.y, out[ai_ * CPT + jj].y);
              } else cfma(out[ai_ * CPT + jj], wc, t1);
            }
        }
      }
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) {
        if (!ok[pt]) continue;
        if (bo == d.lch) yb[(long)po * ca * cb + base[pt]] = out[pt];
        else T2b[((long)po * ca * Dl + bo) * cb + (base[pt] / cb) * (long)Dl * cb + base[pt] % cb] = out[pt];
      }
    }
  }
}

// Small bonds (chi <= 8, or the narrow products of the centre shifts): a 64 x 64 block tile would be mostly padding and
// spend its time on barriers.  Here every wavefront owns one 16 x 16 output tile of one batch entry and feeds the MFMA
// straight from global memory - the operands of such a product are a few KiB and stay in L2 - with the next k-group's loads
// in flight during the four MFMAs of the current one.  No LDS, no barriers, four independent tiles per workgroup.
__global__ __launch_bounds__(256) void zgemm_small_kernel(GemmDesc g, int tiles_m, int tiles_n, long total_tiles) {
  const int lane = threadIdx.x & 63;
  const long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= total_tiles) return;
  const int per = tiles_m * tiles_n;
  int z = (int)(t / per);
  const int tt = (int)(t - (long)z * per);
  const int m0 = (tt / tiles_n) * 16, n0 = (tt % tiles_n) * 16;
  const int b2 = z % g.nb2;
  z /= g.nb2;
  const int b1 = z % g.nb1;
  int b0 = z / g.nb1;
  if (g.ids) b0 = g.ids[b0];
  if (g.active && g.active[b0] == 0) return;
  const cplx* __restrict__ Ab = g.A + (long)b0 * g.a_b0 + (long)b1 * g.a_b1 + (long)b2 * g.a_b2;
  const cplx* __restrict__ Bb = g.B + (long)b0 * g.b_b0 + (long)b1 * g.b_b1 + (long)b2 * g.b_b2;
  cplx* __restrict__ Cb = g.C + (long)b0 * g.c_b0 + (long)b1 * g.c_b1 + (long)b2 * g.c_b2;
  const int li = lane & 15, lk = lane >> 4;
  const int m = m0 + li, n = n0 + li;
  const bool mok = m < g.M, nok = n < g.N;
  const real sgnA = g.conjA ? -1.0 : 1.0, sgnB = g.conjB ? -1.0 : 1.0;
  const int ksteps = (g.K + 3) / 4;
  const int total = ksteps * g.nks;
  auto fetch = [&](int it, cplx& a, cplx& b) {
    const int ks = it / ksteps;
    const int k = (it - ks * ksteps) * 4 + lk;
    const bool kok = k < g.K;
    a = (mok && kok) ? Ab[(long)ks * g.a_ks + (long)m * g.a_rs + (long)k * g.a_cs] : cplx{0.0, 0.0};
    b = (nok && kok) ? Bb[(long)ks * g.b_ks + (long)k * g.b_rs + (long)n * g.b_cs] : cplx{0.0, 0.0};
  };
  real4 accRe = real4{0, 0, 0, 0}, accIm = real4{0, 0, 0, 0};
  cplx a, b, an, bn;
  fetch(0, a, b);
  for (int it = 0; it < total; ++it) {
    if (it + 1 < total) fetch(it + 1, an, bn);
    const real ai = sgnA * a.y, bi = sgnB * b.y;
    accRe = TJM_MFMA(a.x, b.x, accRe);
    accIm = TJM_MFMA(a.x, bi, accIm);
    accRe = TJM_MFMA(-ai, bi, accRe);
    accIm = TJM_MFMA(ai, b.x, accIm);
    a = an;
    b = bn;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int mm = m0 + TJM_ACC_ROW(lane, r);
    if (mm < g.M && nok) {
      cplx v;
      v.x = accRe[r];
      v.y = accIm[r];
      if (g.accumulate != 0) {
        const cplx old = Cb[(long)mm * g.c_rs + n];
        v.x = (g.accumulate > 0) ? old.x + v.x : old.x - v.x;
        v.y = (g.accumulate > 0) ? old.y + v.y : old.y - v.y;
      }
      Cb[(long)mm * g.c_rs + n] = v;
    }
  }
}

}  // namespace

// ---- measurement: launch sampler of zgemm4_kernel (tjm_profile_gemm of the C ABI).  Every N-th launch is bracketed by HIP events
// on its stream; the executed work is counted ON THE DEVICE (output tiles x K of every workgroup: masked trajectories and the
// mirror tiles of Hermitian products are not counted), the bytes are the operands and the result of the launch once each.
#ifndef TJM_F32
namespace {
struct GemmProfile {
  int every = 0;
  long counter = 0, launches = 0, samples = 0;
  double total_ms = 0.0, bytes = 0.0, bytes_all = 0.0;
  unsigned long long* dev_units = nullptr;  // tiles x K executed (device memory of the device that enabled the sampler)
  int dev = -1;
  std::vector<hipEvent_t> pool;
  std::vector<std::pair<int, double>> pending;  // (event pair, bytes)
  size_t used = 0;
};
GemmProfile g_gp;
std::mutex g_gp_mutex;
void gemm_harvest_locked(bool wait) {
  size_t keep = 0;
  for (size_t i = 0; i < g_gp.pending.size(); ++i) {
    const auto p = g_gp.pending[i];
    if (wait) (void)hipEventSynchronize(g_gp.pool[2 * p.first + 1]);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_gp.pool[2 * p.first], g_gp.pool[2 * p.first + 1]) == hipSuccess) {
      g_gp.total_ms += ms;
      g_gp.bytes += p.second;
      ++g_gp.samples;
    } else if (!wait) g_gp.pending[keep++] = p;
  }
  g_gp.pending.resize(wait ? 0 : keep);
  if (g_gp.pending.empty()) g_gp.used = 0;
}
}  // namespace
#endif

void gemm_profile_enable(int every) {
#ifndef TJM_F32
  std::lock_guard<std::mutex> lock(g_gp_mutex);
  if (every > 0 && g_gp.dev_units == nullptr) {
    if (hipGetDevice(&g_gp.dev) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&g_gp.dev_units), sizeof(unsigned long long)) != hipSuccess) {
      g_gp.dev_units = nullptr;
      return;
    }
  }
  if (every > 0) {
    (void)hipMemset(g_gp.dev_units, 0, sizeof(unsigned long long));
    g_gp.counter = g_gp.launches = g_gp.samples = 0;
    g_gp.total_ms = g_gp.bytes = g_gp.bytes_all = 0.0;
    g_gp.pending.clear();
    g_gp.used = 0;
  }
  g_gp.every = every > 0 ? every : 0;
#else
  (void)every;
#endif
}

// out6 = { summed duration of the sampled launches [ms], sampled launches, all launches, executed tiles x K (device counter, all launches),
//          nominal bytes of the sampled launches, nominal bytes of all launches }; call after the streams have been synchronised
void gemm_profile_get(double* out6) {
  for (int i = 0; i < 6; ++i) out6[i] = 0.0;
#ifndef TJM_F32
  std::lock_guard<std::mutex> lock(g_gp_mutex);
  gemm_harvest_locked(true);
  unsigned long long units = 0;
  if (g_gp.dev_units) (void)hipMemcpy(&units, g_gp.dev_units, sizeof(units), hipMemcpyDeviceToHost);
  out6[0] = g_gp.total_ms; out6[1] = (double)g_gp.samples; out6[2] = (double)g_gp.launches; out6[3] = (double)units;
  out6[4] = g_gp.bytes; out6[5] = g_gp.bytes_all;
#endif
}

#ifndef TJM_F32
namespace {
// TJM_GEMM_SHAPES=1 (diagnostic): the distinct product shapes of the process with their launch counts, printed at exit
struct ShapeLog {
  std::mutex m;
  std::vector<std::pair<std::array<long, 10>, long>> rows;
  ~ShapeLog() {
    for (auto& r : rows)
      fprintf(stderr, "[tjm_gemm] M %ld N %ld K %ld nks %ld batches %ld herm %ld acc %ld a_mcontig %ld b_ncontig %ld dot %ld : %ld launches\n", r.first[0], r.first[1],
              r.first[2], r.first[3], r.first[4], r.first[5], r.first[6], r.first[7], r.first[8], r.first[9], r.second);
  }
};
ShapeLog g_shapes;
}  // namespace
#endif

int launch_gemm(const GemmDesc& g_in, hipStream_t stream) {
  GemmDesc g = g_in;
#ifndef TJM_F32
  static const bool log_shapes = getenv("TJM_GEMM_SHAPES") != nullptr;
  if (log_shapes) {
    const std::array<long, 10> key = {g.M, g.N, g.K, g.nks, (long)g.nb0 * g.nb1 * g.nb2, g.hermitian, g.accumulate, (g.a_rs == 1 && g.a_cs != 1) ? 1 : 0, g.b_cs == 1 ? 1 : 0, g.dot_part ? 1 : 0};
    std::lock_guard<std::mutex> lock(g_shapes.m);
    bool found = false;
    for (auto& r : g_shapes.rows)
      if (r.first == key) { ++r.second; found = true; break; }
    if (!found) g_shapes.rows.emplace_back(key, 1L);
  }
#endif
  static const bool direct_mirror = getenv("TJM_GEMM_DIRECT_MIRROR") != nullptr;
  if (g.hermitian && direct_mirror) g.hermitian = 2;
  if (g.M <= 0 || g.N <= 0 || g.nb0 <= 0 || g.nb1 <= 0 || g.nb2 <= 0) return TJM_OK;
  if (g.K <= 0 || g.nks <= 0) return TJM_ERR_ARG;
  // a Hermitian product added to an old C: the mirror tiles of the two kernels disagree on what they would write and no
  // caller needs it; the dot-product epilogue's per-tile partial sums are double-buffered by the k loop, which needs two k-tiles per
  // output tile (K >= 32 in all, the engine asks from 64 on)
  if (g.hermitian && g.accumulate) return TJM_ERR_NOT_IMPLEMENTED;
  if (g.dot_part != nullptr && (long)g.K * g.nks < 32) return TJM_ERR_NOT_IMPLEMENTED;
  static const bool no_small = getenv("TJM_NO_SMALL_GEMM") != nullptr;
  if (!no_small && g.dot_part == nullptr && (g.M <= 32 || g.N <= 32) && (long)g.K * g.nks <= 512) {  // long sums keep the LDS-tiled kernel's four-wave k loop
    const int tiles_m = (g.M + 15) / 16, tiles_n = (g.N + 15) / 16;
    const long total_tiles = (long)tiles_m * tiles_n * g.nb0 * g.nb1 * g.nb2;
    hipLaunchKernelGGL(zgemm_small_kernel, dim3((unsigned)((total_tiles + 3) / 4)), dim3(256), 0, stream, g, tiles_m, tiles_n, total_tiles);
    TJM_HIP_CHECK(hipGetLastError());
    return TJM_OK;
  }
  // 16384 trajectories x d^2 inner batches already exceed the 65535 of grid z: at most 32768 in z, the rest in y
  const long batches = (long)g.nb0 * g.nb1 * g.nb2;
  const long gz = batches < 32768 ? batches : 32768, gy = (batches + gz - 1) / gz;
  if (gy > 65535) return TJM_ERR_ARG;
  dim3 grid(((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN), (unsigned)gy, (unsigned)gz);
  dim3 block(256);
#ifndef TJM_F32
  static const bool mfma4 = getenv("TJM_GEMM_16X16") == nullptr;  // A/B switch: the 16 x 16 x 4 kernel for every shape
  // zgemm4_kernel for whole-tile shapes of up to 48 tiles per batch entry.  (Larger outputs - the products of config 4: 256 x 1024,
  // 512 x 768, 1024 x 256 at K = 256 - run 3 - 5 % FASTER on the register-staged kernel, profiles/r05/gemm_cfg4_shapes.txt, and config 4
  // as a whole 5.0 against 4.8 trajectories/s; row-shaped instead of tile-shaped staging pieces did not change that, and cost the loop
  // its immediate LDS offsets.  TJM_GEMM_4X4_ALL lifts the limit.)
  static const bool all4 = getenv("TJM_GEMM_4X4_ALL") != nullptr;
  const bool use4 = mfma4 && g.M % BM == 0 && g.N % BN == 0 && g.K % 8 == 0 && (all4 || (g.M / BM) * (g.N / BN) <= 48);
  if ((g.b_perm || g.coef) && !use4) return TJM_ERR_NOT_IMPLEMENTED;  // (gemm4_serves)
  if (use4) {
    static int slots = 0, slots3 = 0;  // two resident workgroups per CU (64 KiB of LDS, <= 256 registers)
    if (slots == 0) {
      int dev = 0, cus = 0;
      TJM_HIP_CHECK(hipGetDevice(&dev));
      TJM_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
      slots = 2 * (cus > 0 ? cus : 256);
    }
    const int tiles_m = g.M / BM, tiles_n = g.N / BN;
    static const bool herm_all = getenv("TJM_GEMM_HERM_ALL") != nullptr;
    if (g.hermitian && herm_all) g.hermitian = 3;
    const long total_tiles = ((g.hermitian && g.hermitian != 3) ? (long)tiles_m * (tiles_m + 1) / 2 : (long)tiles_m * tiles_n) * batches;
    static const bool flat = getenv("TJM_GEMM_FLAT_TILES") != nullptr;
    const int xcd_map = (!flat && g.nb0 >= 16 && total_tiles >= slots && slots % 8 == 0) ? 1 : 0;
    static const int abl = getenv("TJM_GEMM_ABL") ? atoi(getenv("TJM_GEMM_ABL")) : 0;
    static const int kgenv = getenv("TJM_GEMM_KG") ? atoi(getenv("TJM_GEMM_KG")) : 4;
    const int kg = (kgenv == 2 || g.K % 16 != 0) ? 2 : 4;
    if (kg == 2 && slots3 == 0) slots3 = slots / 2 * 3;
    const int use_slots = kg == 2 ? slots3 : slots;
    const unsigned nwg2 = (unsigned)(total_tiles < use_slots ? total_tiles : use_slots);
    // the XCD-aware walk derives its slot count from the grid (gridDim.x >> 3): only for grids that are whole multiples of 8 (ADVICE r5:
    // 3 x CUs workgroups of the K % 16 == 8 instance on a part whose CU count is not a multiple of 8 walked some tiles twice)
    const int xm = (xcd_map && total_tiles >= use_slots && use_slots % 8 == 0) ? 1 : 0;
    // measurement (off unless tjm_profile_gemm was called): bracket every N-th launch, count the executed tiles on the device
    unsigned long long* wc = nullptr;
    int ev = -1;
    double nbytes = 0.0;
    std::unique_lock<std::mutex> plock(g_gp_mutex, std::defer_lock);
    if (g_gp.every > 0) {
      plock.lock();
      int dev = -1;
      if (g_gp.every > 0 && hipGetDevice(&dev) == hipSuccess && dev == g_gp.dev) {
        wc = g_gp.dev_units;
        nbytes = 16.0 * (double)batches * (((double)g.M * g.K + (double)g.K * g.N) * g.nks + (double)g.M * g.N * (g.accumulate != 0 ? 2.0 : 1.0));
        ++g_gp.launches;
        g_gp.bytes_all += nbytes;
        if (g_gp.counter++ % g_gp.every == 0) {
          gemm_harvest_locked(false);
          if (g_gp.pool.size() < 2 * (g_gp.used + 1)) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) { g_gp.pool.push_back(a); g_gp.pool.push_back(b); }
          }
          if (g_gp.pool.size() >= 2 * (g_gp.used + 1)) {
            ev = (int)g_gp.used++;
            (void)hipEventRecord(g_gp.pool[2 * ev], stream);
          }
        }
      }
    }
#define TJM_Z4(A) do { if (kg == 2) hipLaunchKernelGGL((zgemm4_kernel<A, 2>), dim3(nwg2), block, 0, stream, g, tiles_m, tiles_n, total_tiles, xm, wc); \
                       else hipLaunchKernelGGL((zgemm4_kernel<A, 4>), dim3(nwg2), block, 0, stream, g, tiles_m, tiles_n, total_tiles, xm, wc); } while (0)
    switch (abl) {
      case 1: TJM_Z4(1); break; case 2: TJM_Z4(2); break; case 15: TJM_Z4(15); break;
      default: TJM_Z4(0);
    }
    if (ev >= 0) {
      (void)hipEventRecord(g_gp.pool[2 * ev + 1], stream);
      g_gp.pending.emplace_back(ev, nbytes);
    }
    if (plock.owns_lock()) plock.unlock();
#undef TJM_Z4
    TJM_HIP_CHECK(hipGetLastError());
    return TJM_OK;
  }
#endif
  const bool am = (g.a_rs == 1 && g.a_cs != 1);
  const bool bn = (g.b_cs == 1);
  static const bool swizzled = getenv("TJM_GEMM_SWIZZLED_LDS") != nullptr;  // diagnostic: the k-major layout for every operand
  if (am && bn) hipLaunchKernelGGL((zgemm_kernel<true, true, false>), grid, block, 0, stream, g);
  else if (swizzled) {
    if (am && !bn) hipLaunchKernelGGL((zgemm_kernel<true, false, false>), grid, block, 0, stream, g);
    else if (!am && bn) hipLaunchKernelGGL((zgemm_kernel<false, true, false>), grid, block, 0, stream, g);
    else hipLaunchKernelGGL((zgemm_kernel<false, false, false>), grid, block, 0, stream, g);
  } else {
    if (am && !bn) hipLaunchKernelGGL((zgemm_kernel<true, false, true>), grid, block, 0, stream, g);
    else if (!am && bn) hipLaunchKernelGGL((zgemm_kernel<false, true, true>), grid, block, 0, stream, g);
    else hipLaunchKernelGGL((zgemm_kernel<false, false, true>), grid, block, 0, stream, g);
  }
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

bool heff_stage12_fits(int P, int ca, int cb, int Dl, int Dr, int rch) {
  static const bool off = getenv("TJM_NO_FUSED_MPO") != nullptr;
  if (off || rch < 0 || (rch != 0 && rch != Dr - 1)) return false;
  const int nch = Dr - 1;
  if (P != 2 && P != 4) return false;
  if (nch != 1 && nch != 2 && nch != 4) return false;
  if (Dl > 6 || Dr > 6 || ca < 16 || cb < 16) return false;
  return true;
}

// Does launch_gemm take this product to zgemm4_kernel (the only kernel that knows b_perm / coef)?
bool gemm4_serves(int M, int N, int K) {
#ifdef TJM_F32
  (void)M; (void)N; (void)K;
  return false;
#else
  static const bool mfma4 = getenv("TJM_GEMM_16X16") == nullptr;
  static const bool all4 = getenv("TJM_GEMM_4X4_ALL") != nullptr;
  return mfma4 && M % BM == 0 && N % BN == 0 && K % 8 == 0 && !(M <= 32 || N <= 32) && (all4 || (M / BM) * (N / BN) <= 48);
#endif
}

int launch_heff_stage12(const HeffStage12Desc& d, hipStream_t stream) {
  if (d.nb0 <= 0) return TJM_OK;
  const int nch = d.Dr - 1;
  const int AT = 64 / d.P, BT = 64 / nch;
  const int tiles = ((d.ca + AT - 1) / AT) * ((d.cb + BT - 1) / BT);
  static const bool flat = getenv("TJM_GEMM_FLAT_TILES") != nullptr;
  const int xcd_map = (!flat && d.nb0 >= 16 && (long)tiles * ((d.nb0 + 7) / 8 * 8) < (1L << 31)) ? 1 : 0;
  dim3 grid(tiles, d.nb0);
  if (xcd_map) grid = dim3((unsigned)(tiles * ((d.nb0 + 7) / 8 * 8)), 1);
#define TJM_S12(PP, NN) hipLaunchKernelGGL((heff_stage12_kernel<PP, NN>), grid, dim3(256), 0, stream, d, tiles, xcd_map)
  if (d.P == 4) {
    if (nch == 1) TJM_S12(4, 1); else if (nch == 2) TJM_S12(4, 2); else TJM_S12(4, 4);
  } else {
    if (nch == 1) TJM_S12(2, 1); else if (nch == 2) TJM_S12(2, 2); else TJM_S12(2, 4);
  }
#undef TJM_S12
  TJM_HIP_CHECK(hipGetLastError());
  return TJM_OK;
}

}  // namespace tjm
