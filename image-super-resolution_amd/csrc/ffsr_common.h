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

static inline int ffsr_launch_status() { return hipGetLastError() == hipSuccess ? FFSR_OK : FFSR_ELAUNCH; }

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
