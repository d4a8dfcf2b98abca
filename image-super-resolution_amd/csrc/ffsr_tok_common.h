// Shared pieces of the token-stationary kernels (ffsr_tok.hip).
#pragma once
#include "ffsr_common.h"
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct TokArgs {
  const float* x;            // [M, ldx] fp32 rows, K channels used
  const unsigned char* w1;   // fragment-major bf16 hi / lo of W1': [steps][G tiles][KS1][2][64 lanes][8]
  const float* b1;           // [steps * G * 16] hidden bias in packed tile order (zero padded)
  const unsigned char* w2;   // [steps][NT2][2][64][8], k slots in accumulator order, rows in the lane-column order
  const float* b2;           // [N] or null
  const float* cvec;         // [N] or null: column scale of (y + b2)
  const float* res;          // [M, ldr] or null: added after the column scale (scaled by rscale * rvec[n])
  const float* rvec;         // [N] or null
  const float* g2;           // [N] post-LayerNorm weight or null (no post-LN)
  const float* be2;          // [N] post-LayerNorm bias
  const float* res2;         // [M, ldr2] or null: added after the post-LN
  float* out;                // [M, ldo] or null
  unsigned short* o_hi;      // planes [M, ldp] or null
  unsigned short* o_lo;
  int ldx, ldr, ldr2, ldo, ldp;
  int M, K, N, steps;
  int pre_ln;
  int res_is_x;              // res == x (same rows, same stride), no column scale: the accumulators start from the input row
  float eps1, eps2, cscale, rscale;
  int act;                   // MODE 2: epilogue activation (FfsrAct)
  float slope;
  // optional TAIL of the chain (MODE 0 / 1): a third linear layer applied to the chain's output row while it is still in
  // registers -- out3 = act3(W3 y + b3) * cscale3 + res3 * rscale3 (DRCT's dense-block "adjust" 1x1 convolutions)
  const unsigned char* w3;   // fragment-major [tsteps][2][KS1][2][64][8] (pack_tok_gemm layout: rows in lane-column order) or null
  const float* b3;           // [tsteps * 32]
  const float* res3;         // [M, ldr3] or null
  float* out3;               // [M, ldo3]
  int ldr3, ldo3, N3, tsteps, act3;
  float slope3, cscale3, rscale3;
  // optional HEAD (template KS0 > 0): a linear layer K0 -> K in FRONT of the chain; x then holds the head's input rows
  // [M, ldx] with K0 channels and the chain's input row is produced in registers:
  //   x1 = LN0?(W0 x + b0) + hres + hres2 * hvec2[row / rows_per_batch]
  // (Swin: x1 = x + proj(attn); GRL: x1 = x + norm1(proj(attn)) + conv_branch * channel_attention).  In the chain, the
  // residual / second residual of x1 come from the registers (RX: out = x1 + mlp(..); otherwise post-LN + x1).
  // MODE 3 = the head alone with the chain's epilogue (bias-free: b0 rides in the head): out = (W0 x + b0) * cvec * cscale
  // + res * rvec * rscale [+ post-LN -> planes; out_pre_ln: the fp32 output is the value BEFORE the post-LN].
  const unsigned char* w0;   // fragment-major [K / 32 steps][2][KS0][2][64][8] (pack_tok_gemm layout)
  const float* b0;           // [hsteps * 32]
  const float* g0;           // [K] LayerNorm of the head's output row, or null
  const float* be0;
  const float* hres;         // [M, ldhr] or null
  const float* hres2;        // [M, ldhr2] or null
  const float* hvec2;        // [batches, K] or null (1)
  const float* hxs;          // [batches, K0] or null: per-image channel scale of the head's INPUT row (NAFNet's channel attention)
  int K0, ldhr, ldhr2, rows_per_batch, hsteps, out_pre_ln;
  float eps0;
  // MODE 3 prologue (MambaIR's out_norm + gate in front of out_proj, mambair_arch.py:381-385): the head's input row is
  //   a = LN(x[0] + x[2 xstride] + x[xstride] + x[3 xstride]; pg, pb, peps) * silu(z)      (xdirs == 4; z != null)
  const float* z;            // [M, ldz] or null
  const float* pg;           // [K0] LayerNorm weight or null
  const float* pb;
  long long xstride;         // elements between the partial inputs
  int xdirs, ldz;
  float peps;
};

template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N <= 63, "vmcnt");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// tools/tok_ablate.sh builds this file with -DFFSR_TOK_ABL=<bits> (timing-only diagnostics, results are wrong):
// 1 no activation math, 2 no fragment reads from LDS, 4 no barriers / waits, 8 no LDS-DMA fills, 16 no MFMAs
#ifndef FFSR_TOK_ABL
#define FFSR_TOK_ABL 0
#endif
__device__ __forceinline__ floatx4 mfma16(bf16x8 a, bf16x8 b, floatx4 c) {
#if FFSR_TOK_ABL & 16
  c[0] += __builtin_bit_cast(floatx4, a)[0] * __builtin_bit_cast(floatx4, b)[0];
  return c;
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}

// gelu(v) = max(v, 0) - g,  g = 0.5 |v| erfc(|v| / sqrt 2)   (identical for both signs of v, no select, no cancellation).
// erfc by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, the size of an fp32 rounding of erf itself): one rcp, one exp2, four
// FMAs -- the activation runs between the two GEMMs of a wave, where ocml's erff (~40 instructions with branches) would cost
// as much issue time as the MFMAs it sits between.  The 0.5 is folded into the polynomial's coefficients.
__device__ __forceinline__ float gelu_fast(float v) {
#if FFSR_TOK_ABL & 1
  return v;
#endif
  const float ax = fabsf(v) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float pl = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
  pl = fmaf(pl, t, 0.5f * 1.421413741f);
  pl = fmaf(pl, t, 0.5f * -0.284496736f);
  pl = fmaf(pl, t, 0.5f * 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(ax * ax * -1.44269504088896340736f);
  return fmaxf(v, 0.f) - fabsf(v) * (pl * t) * e;
}

// 4 floats -> 4 bf16 hi (2 registers) + 4 bf16 lo
__device__ __forceinline__ void split4(const floatx4 v, unsigned& h0, unsigned& h1, unsigned& l0, unsigned& l1) {
  ffsr_split2(v[0], v[1], h0, l0);
  ffsr_split2(v[2], v[3], h1, l1);
}

}  // namespace
