// hipsim: a host-side interpreter of the HIP execution model for the kernels of yaqs_amd/csrc (TEST INFRASTRUCTURE ONLY).
//
// The container this repository is developed in has no GPU.  To exercise the device code before it reaches an MI355X, tests/hipsim
// compiles the UNCHANGED .hip sources of yaqs_amd/csrc with the host compiler against this header instead of <hip/hip_runtime.h>:
// every thread of a workgroup is a fibre with its own stack, __syncthreads / wavefront barriers / cross-lane operations switch
// between fibres, and the cross-lane and matrix instructions the kernels use (ds_bpermute shuffles, DPP row operations,
// v_readlane, v_permlane{16,32}_swap, v_mfma_f64_16x16x4_f64 with the gfx950 operand and result maps) are spelled out lane by lane.
// The result, tests/hipsim/_build/libtjm_sim.so, exports the same C ABI as libtjm_hip.so.  Only tests load it (tests/simengine.py);
// the package never does: yaqs_amd/_lib.py knows one library, the HIP one, and fails without it.  It says nothing about speed or
// about data races between workgroups; it checks indices, strides, control flow and arithmetic.
#pragma once
#define HIPSIM 1  // the kernel sources see the interpreter (tjm_common.h: TJM_GLDS16)
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <tuple>
#include <type_traits>
#include <utility>

// ---- language keywords -------------------------------------------------------------------------------------------------------
#define __host__
#define __device__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __align__(n) alignas(n)

struct dim3 {
  uint32_t x, y, z;
  constexpr dim3(uint32_t x_ = 1, uint32_t y_ = 1, uint32_t z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace hipsim {

struct Wave {
  unsigned gen = 0;
  int arrived = 0, live = 0;
  unsigned long long live_mask = 0;
  unsigned long long xbuf[2][64][2];
};
struct Fiber {
  dim3 tid;
  int lin = 0, lane = 0;
  Wave* wave = nullptr;
  void* sp = nullptr;
  int waiting = 0;  // 0 runnable, 1 workgroup barrier, 2 wavefront barrier, 3 store held back (see hipsim_rt.cpp)
  uintptr_t stack_lo = 0;
  unsigned wait_gen = 0;
  unsigned xop = 0;
  bool done = false;
};
struct Block {
  dim3 bid, bdim, gdim;
  unsigned gen = 0, store_gen = 0;
  int arrived = 0, live = 0;
  void* dyn = nullptr;
};

extern thread_local Fiber* cur;
extern thread_local Block* blk;

void sync_block();
void sync_wave();
// every live lane of the wavefront deposits (a, b); returns the 64 deposited pairs (zero for lanes that have left the kernel)
const unsigned long long (*exchange(unsigned long long a, unsigned long long b))[2];
void launch_impl(const char* name, dim3 grid, dim3 block, size_t shmem, void (*tramp)(void*), void* arg, const void* kernel);

inline void* dyn_shared() { return blk->dyn; }

template <class T>
inline unsigned long long bits(T v) {
  static_assert(sizeof(T) <= 8, "shuffle payload");
  unsigned long long u = 0;
  std::memcpy(&u, &v, sizeof(T));
  return u;
}
template <class T>
inline T unbits(unsigned long long u) {
  T v;
  std::memcpy(&v, &u, sizeof(T));
  return v;
}

template <class K, class T>
struct Thunk {
  K k;
  T args;
  static void run(void* p) {
    Thunk* t = static_cast<Thunk*>(p);
    std::apply(t->k, t->args);
  }
};

template <class... P, class... A>
inline void launch(const char* name, void (*k)(P...), dim3 grid, dim3 block, size_t shmem, A&&... a) {
  using Tup = std::tuple<std::decay_t<P>...>;
  Thunk<void (*)(P...), Tup> t{k, Tup(std::forward<A>(a)...)};
  launch_impl(name, grid, block, shmem, &Thunk<void (*)(P...), Tup>::run, &t, reinterpret_cast<const void*>(k));
}

}  // namespace hipsim

#define threadIdx (hipsim::cur->tid)
#define blockIdx (hipsim::blk->bid)
#define blockDim (hipsim::blk->bdim)
#define gridDim (hipsim::blk->gdim)
#define warpSize 64

// ---- barriers and fences -----------------------------------------------------------------------------------------------------
#define __syncthreads() hipsim::sync_block()
// A wavefront runs in lock step on the device, so code may pass data between its lanes through memory with nothing but a fence
// (for the compiler) in between: here a fence is a meeting point of the wavefront.
#define __threadfence_block() hipsim::sync_wave()
#define __threadfence() hipsim::sync_wave()
#define __builtin_amdgcn_wave_barrier() hipsim::sync_wave()
#define __builtin_amdgcn_fence(order, scope) hipsim::sync_wave()

// ---- cross-lane operations ---------------------------------------------------------------------------------------------------
template <class T>
inline T __shfl(T v, int src, int width = 64) {
  auto x = hipsim::exchange(hipsim::bits(v), 0);
  const int lane = hipsim::cur->lane, base = lane & ~(width - 1);
  return hipsim::unbits<T>(x[base + (src & (width - 1))][0]);
}
template <class T>
inline T __shfl_xor(T v, int mask, int width = 64) {
  auto x = hipsim::exchange(hipsim::bits(v), 0);
  const int lane = hipsim::cur->lane, base = lane & ~(width - 1);
  int src = lane ^ mask;
  if (src < base || src >= base + width) src = lane;
  return hipsim::unbits<T>(x[src][0]);
}
template <class T>
inline T __shfl_down(T v, unsigned delta, int width = 64) {
  auto x = hipsim::exchange(hipsim::bits(v), 0);
  const int lane = hipsim::cur->lane, base = lane & ~(width - 1);
  int src = lane + (int)delta;
  if (src >= base + width) src = lane;
  return hipsim::unbits<T>(x[src][0]);
}
template <class T>
inline T __shfl_up(T v, unsigned delta, int width = 64) {
  auto x = hipsim::exchange(hipsim::bits(v), 0);
  const int lane = hipsim::cur->lane, base = lane & ~(width - 1);
  int src = lane - (int)delta;
  if (src < base) src = lane;
  return hipsim::unbits<T>(x[src][0]);
}
inline int __any(int pred) {
  auto x = hipsim::exchange(pred ? 1ull : 0ull, 0);
  int r = 0;
  for (int l = 0; l < 64; ++l) r |= (int)x[l][0];
  return r;
}
inline unsigned long long __ballot(int pred) {
  auto x = hipsim::exchange(pred ? 1ull : 0ull, 0);
  unsigned long long r = 0;
  for (int l = 0; l < 64; ++l) r |= (x[l][0] & 1ull) << l;
  return r;
}
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int hipsim_readlane(int v, int lane) {
  auto x = hipsim::exchange((unsigned long long)(unsigned)v, 0);
  return (int)(unsigned)x[lane & 63][0];
}
#define __builtin_amdgcn_readlane(v, lane) hipsim_readlane(v, lane)
// v_readfirstlane_b32: the value of the first lane (all 64 lanes are active wherever the kernels use it)
#define __builtin_amdgcn_readfirstlane(v) hipsim_readlane((int)(v), 0)

// v_mov_b32 with a DPP control (row_mask = bank_mask = 0xF): quad_perm, row_shl / row_shr / row_ror, row_mirror, row_half_mirror.
// Lanes whose source falls outside the row read `old` (bound_ctrl: 0).
inline int hipsim_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  auto x = hipsim::exchange((unsigned long long)(unsigned)src, 0);
  const int lane = hipsim::cur->lane, row = lane & ~15, i = lane & 15;
  int s = -1;
  if (ctrl >= 0x00 && ctrl <= 0xFF) s = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
  else if (ctrl >= 0x101 && ctrl <= 0x10F) { const int j = i + (ctrl & 15); s = j < 16 ? row + j : -1; }         // row_shl
  else if (ctrl >= 0x111 && ctrl <= 0x11F) { const int j = i - (ctrl & 15); s = j >= 0 ? row + j : -1; }         // row_shr
  else if (ctrl >= 0x121 && ctrl <= 0x12F) s = row + ((i - (ctrl & 15) + 16) & 15);                              // row_ror
  else if (ctrl == 0x140) s = row + (15 - i);                                                                    // row_mirror
  else if (ctrl == 0x141) s = row + (i & 8) + (7 - (i & 7));                                                     // row_half_mirror
  else { fprintf(stderr, "[hipsim] DPP control 0x%x not modelled\n", ctrl); abort(); }
  if (row_mask != 0xF || bank_mask != 0xF) { fprintf(stderr, "[hipsim] DPP masks not modelled\n"); abort(); }
  if (s < 0) return bound_ctrl ? 0 : old;
  return (int)(unsigned)x[s][0];
}
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) hipsim_update_dpp(old, src, ctrl, rm, bm, bc)

struct hipsim_pair {
  unsigned v[2];
  unsigned operator[](int i) const { return v[i]; }
};
// v_permlane16_swap: the odd rows (16 lanes) of vdst trade places with the even rows of src; returns (vdst', src')
inline hipsim_pair hipsim_permlane16_swap(unsigned vdst, unsigned src, bool, bool) {
  auto x = hipsim::exchange(vdst, src);
  const int lane = hipsim::cur->lane;
  hipsim_pair r;
  if ((lane >> 4) & 1) { r.v[0] = (unsigned)x[lane - 16][1]; r.v[1] = src; }
  else { r.v[0] = vdst; r.v[1] = (unsigned)x[lane + 16][0]; }
  return r;
}
// v_permlane32_swap: the upper half of vdst trades places with the lower half of src
inline hipsim_pair hipsim_permlane32_swap(unsigned vdst, unsigned src, bool, bool) {
  auto x = hipsim::exchange(vdst, src);
  const int lane = hipsim::cur->lane;
  hipsim_pair r;
  if (lane >= 32) { r.v[0] = (unsigned)x[lane - 32][1]; r.v[1] = src; }
  else { r.v[0] = vdst; r.v[1] = (unsigned)x[lane + 32][0]; }
  return r;
}
#define __builtin_amdgcn_permlane16_swap(a, b, fi, bc) hipsim_permlane16_swap((unsigned)(a), (unsigned)(b), fi, bc)
#define __builtin_amdgcn_permlane32_swap(a, b, fi, bc) hipsim_permlane32_swap((unsigned)(a), (unsigned)(b), fi, bc)

// v_mfma_f64_16x16x4_f64: lane l gives A[l & 15][l >> 4] and B[l >> 4][l & 15]; result register v of lane l is D[(l >> 4) + 4 v][l & 15]
typedef double hipsim_d4 __attribute__((ext_vector_type(4)));
inline hipsim_d4 hipsim_mfma_f64_16x16x4(double a, double b, hipsim_d4 c, int, int, int) {
  auto x = hipsim::exchange(hipsim::bits(a), hipsim::bits(b));
  const int lane = hipsim::cur->lane, col = lane & 15, r0 = lane >> 4;
  hipsim_d4 d = c;
  for (int v = 0; v < 4; ++v) {
    const int row = r0 + 4 * v;
    double acc = d[v];
    for (int k = 0; k < 4; ++k) acc = std::fma(hipsim::unbits<double>(x[16 * k + row][0]), hipsim::unbits<double>(x[16 * k + col][1]), acc);
    d[v] = acc;
  }
  return d;
}
#define __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, x, y, z) hipsim_mfma_f64_16x16x4(a, b, c, x, y, z)

// v_mfma_f64_4x4x4_4b_f64: four independent 4 x 4 x 4 products per instruction.  Lane map found by experiment on the MI355X
// (tools/probes/mfma_4x4x4_layout_probe.hip, gpurun_out/r05/mfma_4x4x4_layout.txt): lane l gives A[blk][i][k] with k = l >> 4,
// blk = (l >> 2) & 3, i = l & 3, and B[blk][k][j] with the same split (j = l & 3); the result of lane l is D[blk][i][j] with
// i = l >> 4, blk = (l >> 2) & 3, j = l & 3.
inline double hipsim_mfma_f64_4x4x4(double a, double b, double c, int, int, int) {
  auto x = hipsim::exchange(hipsim::bits(a), hipsim::bits(b));
  const int lane = hipsim::cur->lane, i = lane >> 4, blk = (lane >> 2) & 3, j = lane & 3;
  double acc = c;
  for (int k = 0; k < 4; ++k) acc = std::fma(hipsim::unbits<double>(x[16 * k + 4 * blk + i][0]), hipsim::unbits<double>(x[16 * k + 4 * blk + j][1]), acc);
  return acc;
}
#define __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, x, y, z) hipsim_mfma_f64_4x4x4(a, b, c, x, y, z)

// global_load_lds_dwordx4: every lane's 16 bytes go from its own global address to LDS at (the wave-uniform base) + lane x 16.
// The interpreter runs one lane at a time, so the base each lane passes must be the same for the whole wavefront (as the instruction
// takes it from M0): lane 0's pointer is broadcast and compared.
inline void hipsim_glds16(const void* gptr, void* lbase) {
  auto x = hipsim::exchange((unsigned long long)(uintptr_t)lbase, 0);
  if (x[0][0] != x[hipsim::cur->lane][0]) { fprintf(stderr, "[hipsim] global_load_lds: LDS base differs between lanes\n"); abort(); }
  std::memcpy((char*)(uintptr_t)x[0][0] + 16 * hipsim::cur->lane, gptr, 16);
}

// v_mfma_f32_16x16x4_f32: same operand maps; result register v of lane l is D[4 (l >> 4) + v][l & 15] (the dtype-independent C/D map)
typedef float hipsim_f4 __attribute__((ext_vector_type(4)));
inline hipsim_f4 hipsim_mfma_f32_16x16x4(float a, float b, hipsim_f4 c, int, int, int) {
  auto x = hipsim::exchange(hipsim::bits(a), hipsim::bits(b));
  const int lane = hipsim::cur->lane, col = lane & 15, r0 = 4 * (lane >> 4);
  hipsim_f4 d = c;
  for (int v = 0; v < 4; ++v) {
    const int row = r0 + v;
    float acc = d[v];
    for (int k = 0; k < 4; ++k) acc = std::fmaf(hipsim::unbits<float>(x[16 * k + row][0]), hipsim::unbits<float>(x[16 * k + col][1]), acc);
    d[v] = acc;
  }
  return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) hipsim_mfma_f32_16x16x4(a, b, c, x, y, z)

// ---- scalar helpers ----------------------------------------------------------------------------------------------------------
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline long min(long a, long b) { return a < b ? a : b; }
inline long max(long a, long b) { return a > b ? a : b; }
inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
inline double min(double a, double b) { return std::fmin(a, b); }
inline double max(double a, double b) { return std::fmax(a, b); }
inline double rsqrt(double x) { return 1.0 / std::sqrt(x); }
// v_rsq_f64 / v_rcp_f64 are estimates good to 2^29 ulp (relative 2^-23): modelled at exactly that resolution (the low 29 bits of the
// mantissa cleared), so code that needs the full precision has to earn it with its own Newton steps here as on the device
inline double hipsim_estimate(double v) {
  if (!std::isfinite(v)) return v;
  unsigned long long u;
  std::memcpy(&u, &v, 8);
  u &= ~((1ull << 29) - 1ull);
  std::memcpy(&v, &u, 8);
  return v;
}
#define __builtin_amdgcn_rsq(x) hipsim_estimate(1.0 / std::sqrt((double)(x)))
#define __builtin_amdgcn_rcp(x) hipsim_estimate(1.0 / (double)(x))
#define __builtin_amdgcn_rsqf(x) (1.0f / std::sqrt((float)(x)))
#define __builtin_amdgcn_rcpf(x) (1.0f / (float)(x))
inline int __float_as_int(float v) { return hipsim::unbits<int>(hipsim::bits(v)); }
inline float __int_as_float(int v) { return hipsim::unbits<float>(hipsim::bits(v)); }
inline int __double2loint(double v) { return (int)(unsigned)(hipsim::bits(v) & 0xffffffffull); }
inline int __double2hiint(double v) { return (int)(unsigned)(hipsim::bits(v) >> 32); }
inline double __hiloint2double(int hi, int lo) {
  return hipsim::unbits<double>(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}

// ---- atomics (workgroups may run on several host threads) -------------------------------------------------------------------------
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline float atomicAdd(float* p, float v) {
  unsigned* q = reinterpret_cast<unsigned*>(p);
  unsigned o = __atomic_load_n(q, __ATOMIC_RELAXED);
  for (;;) {
    const unsigned n = hipsim::unbits<unsigned>(hipsim::bits(hipsim::unbits<float>(o) + v));
    if (__atomic_compare_exchange_n(q, &o, n, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return hipsim::unbits<float>(o);
  }
}
inline double atomicAdd(double* p, double v) {
  unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
  unsigned long long o = __atomic_load_n(q, __ATOMIC_RELAXED);
  for (;;) {
    const unsigned long long n = hipsim::bits(hipsim::unbits<double>(o) + v);
    if (__atomic_compare_exchange_n(q, &o, n, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return hipsim::unbits<double>(o);
  }
}
inline int atomicOr(int* p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
inline int atomicMax(int* p, int v) {
  int o = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (o < v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return o;
}

// ---- runtime API (every stream is synchronous; "device memory" is host memory) ------------------------------------------------------
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
constexpr hipError_t hipErrorInvalidValue = 1;
typedef struct ihipStream_t* hipStream_t;
typedef struct hipsim_event* hipEvent_t;
struct hipsim_event {
  std::chrono::steady_clock::time_point t;
};
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
constexpr unsigned hipHostMallocDefault = 0;
struct hipPointerAttribute_t {
  int type;
  int device;
  void* devicePointer;
  void* hostPointer;
};

inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipsim error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 4; return hipSuccess; }  // a small 'device': persistent grids stay small
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t = nullptr) {
  if (n) std::memmove(dst, src, n);
  return hipSuccess;
}
inline hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind k) { return hipMemcpyAsync(dst, src, n, k); }
inline hipError_t hipMemsetAsync(void* dst, int v, size_t n, hipStream_t = nullptr) {
  if (n) std::memset(dst, v, n);
  return hipSuccess;
}
inline hipError_t hipMemset(void* dst, int v, size_t n) { return hipMemsetAsync(dst, v, n); }
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : 2; }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind,
                                   hipStream_t = nullptr) {
  for (size_t r = 0; r < height; ++r) std::memmove(static_cast<char*>(dst) + r * dpitch, static_cast<const char*>(src) + r * spitch, width);
  return hipSuccess;
}
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) {
  *p = std::malloc(n ? n : 1);
  return *p ? hipSuccess : hipErrorInvalidValue;
}
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipsim_event; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
namespace hipsim {
void set_max_dynamic_lds(const void* kernel, int bytes);
}
template <class F>
inline hipError_t hipFuncSetAttribute(F f, hipFuncAttribute attr, int value) {
  if (attr == hipFuncAttributeMaxDynamicSharedMemorySize) hipsim::set_max_dynamic_lds(reinterpret_cast<const void*>(f), value);
  return hipSuccess;
}
inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) {
  a->type = 2;
  a->device = 0;
  a->devicePointer = const_cast<void*>(p);
  a->hostPointer = nullptr;
  return hipSuccess;
}

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  hipsim::launch(#kernel, kernel, dim3(grid), dim3(block), (size_t)(shmem), ##__VA_ARGS__)
