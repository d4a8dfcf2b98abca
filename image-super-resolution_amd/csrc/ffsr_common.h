// Shared helpers for the FreqFusionSR gfx950 kernels (device code is CDNA4-only: wave64, MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FFSR_OK 0
#define FFSR_EINVAL (-1)     // bad shape / stride / alignment argument
#define FFSR_ELAUNCH (-2)    // hipGetLastError() after launch was not hipSuccess

#define FFSR_CHECK(cond)            \
  do {                              \
    if (!(cond)) return FFSR_EINVAL; \
  } while (0)

// Launch bookkeeping.  Every kernel launch of the library goes through FFSR_LAUNCH: it first clears the thread's sticky
// "last error" (the host framework's own event / stream queries leave hipErrorNotReady there), launches, and records a
// failure of THIS launch in a thread-local flag.  ffsr_launch_status() at the end of an entry point reports (and resets)
// that flag, so a multi-launch entry point (FFT band split, selective scan, ...) fails if ANY of its launches failed, not
// only the last one.
inline int& ffsr_launch_failed() {
  static thread_local int failed = 0;
  return failed;
}
static inline int ffsr_launch_status() {
  int& f = ffsr_launch_failed();
  const int rc = f ? FFSR_ELAUNCH : FFSR_OK;
  f = 0;
  return rc;
}
#define FFSR_LAUNCH(...)                                         \
  do {                                                           \
    (void)hipGetLastError();                                     \
    hipLaunchKernelGGL(__VA_ARGS__);                             \
    if (hipGetLastError() != hipSuccess) ffsr_launch_failed() = 1; \
  } while (0)

// Kernels with more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize set once PER DEVICE (the code
// object is loaded per device).  `done_mask` = a static per kernel instantiation: bit d = already set on device d.
static inline int ffsr_allow_dynamic_lds(const void* const* fns, int n, int bytes, unsigned long long* done_mask) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return FFSR_ELAUNCH;
  const unsigned long long bit = 1ull << (dev & 63);
  if (__atomic_load_n(done_mask, __ATOMIC_ACQUIRE) & bit) return FFSR_OK;
  for (int i = 0; i < n; ++i)
    if (hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return FFSR_ELAUNCH;
  __atomic_fetch_or(done_mask, bit, __ATOMIC_RELEASE);
  return FFSR_OK;
}

// Products of the split-bf16 GEMM kernels: 3 = hi*hi + hi*lo + lo*hi (default, ~1e-5 relative per product), 1 = hi*hi only
// (plain bf16 operands, fp32 accumulate: FFSR_GEMM_MODE=bf16, ffsr_set_gemm_terms).  Process-wide mode switch, set before use.
extern int g_ffsr_gemm_terms;

// activation codes shared by the GEMM/conv epilogue, the depthwise conv and the elementwise kernels
enum FfsrAct : int {
  FFSR_ACT_NONE = 0,
  FFSR_ACT_GELU = 1,     // exact erf form (torch.nn.GELU default)
  FFSR_ACT_RELU = 2,
  FFSR_ACT_LRELU = 3,    // slope passed separately
  FFSR_ACT_SIGMOID = 4,
  FFSR_ACT_SILU = 5,
};

__device__ __forceinline__ float ffsr_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }

__device__ __forceinline__ float ffsr_act(float v, int act, float slope) {
  switch (act) {
    case FFSR_ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    case FFSR_ACT_RELU: return v > 0.f ? v : 0.f;
    case FFSR_ACT_LRELU: return v > 0.f ? v : v * slope;
    case FFSR_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    case FFSR_ACT_SILU: return v / (1.0f + expf(-v));
    default: return v;
  }
}

// Counter-based random bits for dropout masks: mask element `idx` of the draw identified by `seed` -> 32 uniform bits
// (splitmix64 finaliser of seed * golden + idx; stateless, so the backward pass regenerates the forward pass's mask from the
// same (seed, idx)).  keep = bits >= threshold, threshold = p * 2^32.
__device__ __forceinline__ unsigned ffsr_rng_u32(unsigned long long seed, unsigned long long idx) {
  unsigned long long z = seed * 0x9E3779B97F4A7C15ull + idx;
  z ^= z >> 30;
  z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27;
  z *= 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (unsigned)(z >> 32);
}
static inline unsigned ffsr_drop_threshold(float p) {
  const double t = (double)p * 4294967296.0;
  return t <= 0.0 ? 0u : (t >= 4294967295.0 ? 0xffffffffu : (unsigned)t);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

// ---- split-bf16 planes: x ~ hi + lo with hi = bf16(x) (round to nearest even), lo = bf16(x - hi).  Two values per call
// (packed pairs: low half = first value).
typedef __bf16 ffsr_bf16x2 __attribute__((ext_vector_type(2)));
typedef float ffsr_floatx2 __attribute__((ext_vector_type(2)));
typedef unsigned ffsr_uintx4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ffsr_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
  const ffsr_floatx2 x = {x0, x1};
  const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(x, ffsr_bf16x2));
  const ffsr_floatx2 r = {x0 - __builtin_bit_cast(float, h << 16), x1 - __builtin_bit_cast(float, h & 0xffff0000u)};
  hi = h;
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, ffsr_bf16x2));
}
// 8 consecutive channels of one row -> 16 bytes of the hi plane and 16 bytes of the lo plane
__device__ __forceinline__ void ffsr_store_planes8(unsigned short* hi, unsigned short* lo, const float* v) {
  unsigned h[4], l[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) ffsr_split2(v[2 * c], v[2 * c + 1], h[c], l[c]);
  *reinterpret_cast<ffsr_uintx4*>(hi) = ffsr_uintx4{h[0], h[1], h[2], h[3]};
  *reinterpret_cast<ffsr_uintx4*>(lo) = ffsr_uintx4{l[0], l[1], l[2], l[3]};
}
