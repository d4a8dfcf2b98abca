// Token-stationary fused chains:  out = epilogue( W2 . act( W1 . prologue(x) + b1 ) + b2 )  per token, one kernel.
//
// Why: the K <= 320 token GEMMs of the Swin / GRL blocks and NAFNet's pointwise halves are latency- and byte-bound as
// separate launches (LayerNorm -> fc1 -> GELU -> fc2 -> +res = 3 launches, ~950 MB of HBM traffic per block at 340x510,
// of which 2 x 277 MB is the hidden layer's round trip).  Here a wave owns 16 tokens for the whole chain:
//   * its input row block lives in REGISTERS as the B operand of v_mfma_f32_16x16x32_bf16 (bf16 hi / lo planes, split
//     on the fly from the fp32 row; optional LayerNorm of the row first -- the affine part is folded into W1 / b1 at
//     pack time), lane = (token l & 15, k-quarter q = l >> 4);
//   * both GEMMs are computed TRANSPOSED, D = W_tile (A operand: 16 output features x 32 k) x X^T (B operand: 32 k x
//     16 tokens), so every result tile has its token on the LANE and 4 features in its 4 registers.  The activated
//     hidden tile is therefore, after bf16 conversion, directly the B operand of the second GEMM (its k order is a
//     fixed permutation that the packed W2 absorbs): the hidden layer never leaves the registers -- no LDS, no HBM;
//   * the rows of W2 are packed in the order that makes lane q's output columns {32 s + 8 q + 0..7} -- the columns of
//     the input row the same lane loaded.  When the residual IS the input row (x + mlp(norm(x))), the accumulators are
//     simply initialised with the fp32 row: the residual costs no second read;
//   * the weights stream through an LDS ring by LDS-DMA in FRAGMENT-MAJOR order (packed once at load time: each 1 KB
//     piece is exactly one wave's A fragment, lane-linear), so staging needs no swizzle and every ds_read_b128 is a
//     contiguous, conflict-free 1 KB read; all waves of the workgroup share the stream (one raw s_barrier per slot,
//     counted vmcnt keeps D-2 slots in flight);
//   * workgroups are PERSISTENT (one per CU) and walk a list of 16 x WAVES-token tiles; the ring of weight fills runs
//     on across tile boundaries and the rows of the NEXT tile are loaded into registers while the current tile computes.
//     (First version, one tile per workgroup: rocprof + ablation builds showed ~100 us of a 227 us launch was the
//     exposed load / store phase -- at one workgroup per CU nothing overlapped it.)
//   * products are 3-term split-bf16 (hi*hi + hi*lo + lo*hi, fp32 accumulate) like the other GEMM kernels;
//   * the epilogue (bias, column scale, residual, optional LayerNorm of the output row + second residual, fp32 and / or
//     bf16-plane stores) works per lane on 4 consecutive columns: 16-byte fp32 stores, 8-byte plane stores.
// HBM traffic of a Swin MLP block: read x once, write out once.
#include "ffsr_tok_common.h"

namespace {

// The wave's 16 rows, fp32, in operand order: lane (token l & 15, q = l >> 4) holds columns 32 s + 8 q + 4 h + 0..3.
// BF (branch-free): a column past K reads column 0 of the (always valid) row and is zeroed by a select -- a load under a bounds
// branch is compiled as an exec-masked block of its own and waited for one at a time.  Measured: the single-GEMM kernels and the
// plain chains gain 2-8 % (qkv 383 -> 376 us, NAFNet's gated half 403 -> 369 us); the head kernels, already at 256 registers, lose
// what the freer load scheduling costs them in spills (NAFNet head 543 -> 604 us, DRCT 559 -> 580 us) and keep the branches.
template <int KS1, bool BF = true>
__device__ __forceinline__ void tok_load_rows(const float* xr, int K, int q, floatx4 (&xv)[KS1][2]) {
#pragma unroll
  for (int s = 0; s < KS1; ++s)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = 32 * s + 8 * q + 4 * h;     // K % 4 == 0
      if (KS1 > 4 && s < KS1 - 1) {   // (compile-time; KS1 = ceil(K / 32) is the launchers' contract: only the last step can be partial)
        xv[s][h] = *reinterpret_cast<const floatx4*>(xr + k);
        continue;
      }
      if constexpr (BF) {
        const bool in = k < K;
        const floatx4 v = *reinterpret_cast<const floatx4*>(xr + (in ? k : 0));
        xv[s][h] = in ? v : floatx4{0.f, 0.f, 0.f, 0.f};
      } else {
        xv[s][h] = (k < K) ? *reinterpret_cast<const floatx4*>(xr + k) : floatx4{0.f, 0.f, 0.f, 0.f};
      }
    }
}

// rows -> (optionally normalised: two-pass LayerNorm statistics over the row, which is spread over the lanes l, l^16, l^32,
// l^48) bf16 hi / lo B operands of the first GEMM.  xv is left untouched (it may serve as the residual).
template <int KS1>
__device__ __forceinline__ void tok_prepare(int K, bool ln, float eps, int q, const floatx4 (&xv)[KS1][2], bf16x8 (&a_hi)[KS1],
                                            bf16x8 (&a_lo)[KS1]) {
  float mean = 0.f, rstd = 1.f;
  if (ln) {
    float s1 = 0.f;
#pragma unroll
    for (int s = 0; s < KS1; ++s)
#pragma unroll
      for (int h = 0; h < 2; ++h) s1 += (xv[s][h][0] + xv[s][h][1]) + (xv[s][h][2] + xv[s][h][3]);
    s1 += __shfl_xor(s1, 16, 64);
    s1 += __shfl_xor(s1, 32, 64);
    mean = s1 / (float)K;
    float s2 = 0.f;
#pragma unroll
    for (int s = 0; s < KS1; ++s)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bool in = 32 * s + 8 * q + 4 * h < K;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float d = in ? xv[s][h][c] - mean : 0.f;
          s2 = fmaf(d, d, s2);
        }
      }
    s2 += __shfl_xor(s2, 16, 64);
    s2 += __shfl_xor(s2, 32, 64);
    rstd = rsqrtf(s2 / (float)K + eps);
  }
#pragma unroll
  for (int s = 0; s < KS1; ++s) {
    floatx4 v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bool in = 32 * s + 8 * q + 4 * h < K;
#pragma unroll
      for (int c = 0; c < 4; ++c) v[h][c] = in ? (xv[s][h][c] - mean) * rstd : 0.f;
    }
    unsigned hh[4], ll[4];
    split4(v[0], hh[0], hh[1], ll[0], ll[1]);
    split4(v[1], hh[2], hh[3], ll[2], ll[3]);
    a_hi[s] = __builtin_bit_cast(bf16x8, uintx4{hh[0], hh[1], hh[2], hh[3]});
    a_lo[s] = __builtin_bit_cast(bf16x8, uintx4{ll[0], ll[1], ll[2], ll[3]});
  }
}

// Epilogue of a wave's 16 tokens: accumulator tile n of lane (token, q) holds columns 32 (n >> 1) + 8 q + 4 (n & 1) + 0..3
// (the packed W2 row order), i.e. the same column groups the lane loaded of its input row.
// XR2: the residual after the post-LN is the row xr the wave holds (a head's result), not a global read.
// ALLIN: N is a multiple of 32 (no partial tile at all; selected at run time for the narrow NAFNet shapes, where half the tiles are
// "the last pair").
template <int NT2, bool XR2, bool ALLIN = false>
__device__ __forceinline__ void tok_epilogue_body(const TokArgs& p, floatx4 (&acc)[NT2], const floatx4 (&xr)[NT2 / 2][2], long long tok,
                                                  bool tok_ok, size_t row, int q) {
  const int N = p.N;
  // Every vector / row read below is UNCONDITIONAL per lane (a column past N reads column 0 and is zeroed afterwards; N % 4 == 0) and
  // the loops sit INSIDE the wave-uniform pointer tests: a load under a lane-dependent bounds branch becomes an exec-masked block of
  // its own whose result is waited for before the next block starts -- NT2 dependent round trips per tile instead of one.
  // (the compiler barrier keeps these loads BELOW the GEMM loop: they depend on nothing the loop computes and would otherwise be
  //  hoisted in front of it, 50-150 registers live across the whole tile)
  if constexpr (NT2 > 8) asm volatile("" ::: "memory");      // (the narrow NAFNet shapes have the registers and lose 20 % behind it)
  int cl[NT2];
  bool in[NT2];
#pragma unroll
  for (int n = 0; n < NT2; ++n) {
    // NT2 = 2 ceil(N / 32) (the launchers' choice): only the last tile PAIR can reach past N -- for the others `in` is a compile-time
    // true and the column a constant offset from one per-lane base (no select, no address register of its own)
    const int col = 32 * (n >> 1) + 8 * q + 4 * (n & 1);
    in[n] = (ALLIN || n < NT2 - 2) ? true : col < N;
    cl[n] = in[n] ? col : 0;
  }
  if (p.b2) {
#pragma unroll
    for (int n = 0; n < NT2; ++n) acc[n] += *reinterpret_cast<const floatx4*>(p.b2 + cl[n]);
  }
  if (!p.res_is_x) {
    if (p.cvec) {
#pragma unroll
      for (int n = 0; n < NT2; ++n) acc[n] *= *reinterpret_cast<const floatx4*>(p.cvec + cl[n]) * p.cscale;
    } else {
#pragma unroll
      for (int n = 0; n < NT2; ++n) acc[n] *= p.cscale;
    }
    if (p.res) {
      const float* rr = p.res + row * p.ldr;
      if (p.rvec) {
#pragma unroll
        for (int n = 0; n < NT2; ++n)
          acc[n] += *reinterpret_cast<const floatx4*>(rr + cl[n]) * (*reinterpret_cast<const floatx4*>(p.rvec + cl[n]) * p.rscale);
      } else {
#pragma unroll
        for (int n = 0; n < NT2; ++n) acc[n] += *reinterpret_cast<const floatx4*>(rr + cl[n]) * p.rscale;
      }
    }
  }
#pragma unroll
  for (int n = 0; n < NT2; ++n) {
    if (!in[n]) acc[n] = floatx4{0.f, 0.f, 0.f, 0.f};
    if (p.out_pre_ln && tok_ok && in[n]) *reinterpret_cast<floatx4*>(p.out + (size_t)tok * p.ldo + cl[n]) = acc[n];
  }
  if (p.g2) {       // LayerNorm over the N output columns of the token (two-pass), affine, second residual
    float s1 = 0.f;
#pragma unroll
    for (int n = 0; n < NT2; ++n) s1 += (acc[n][0] + acc[n][1]) + (acc[n][2] + acc[n][3]);
    s1 += __shfl_xor(s1, 16, 64);
    s1 += __shfl_xor(s1, 32, 64);
    const float mean = s1 / (float)N;
    float s2 = 0.f;
#pragma unroll
    for (int n = 0; n < NT2; ++n) {
      const bool in = 32 * (n >> 1) + 8 * q + 4 * (n & 1) < N;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float d = in ? acc[n][c] - mean : 0.f;
        acc[n][c] = d;
        s2 = fmaf(d, d, s2);
      }
    }
    s2 += __shfl_xor(s2, 16, 64);
    s2 += __shfl_xor(s2, 32, 64);
    const float rstd = rsqrtf(s2 / (float)N + p.eps2);
#pragma unroll
    for (int n = 0; n < NT2; ++n) {
      const floatx4 v = acc[n] * rstd * *reinterpret_cast<const floatx4*>(p.g2 + cl[n]) + *reinterpret_cast<const floatx4*>(p.be2 + cl[n]);
      if (in[n]) acc[n] = v;        // (acc is 0 in the padding columns and stays 0)
    }
    if constexpr (XR2) {
#pragma unroll
      for (int n = 0; n < NT2; ++n)
        if (in[n]) acc[n] += xr[n >> 1][n & 1];
    } else if (p.res2) {
      const float* rr = p.res2 + row * p.ldr2;
#pragma unroll
      for (int n = 0; n < NT2; ++n) {
        const floatx4 r = *reinterpret_cast<const floatx4*>(rr + cl[n]);
        if (in[n]) acc[n] += r;
      }
    }
  }
  if (!tok_ok) return;
#pragma unroll
  for (int n = 0; n < NT2; ++n) {
    const int col = 32 * (n >> 1) + 8 * q + 4 * (n & 1);
    if (p.out && !p.out_pre_ln && in[n]) *reinterpret_cast<floatx4*>(p.out + (size_t)tok * p.ldo + col) = acc[n];
    if (p.o_hi && (ALLIN || n < NT2 - 2 || col < p.ldp)) {      // columns N .. ldp-1 of the planes are written as zeros (acc is zero there; ldp >= N)
      unsigned h0, h1, l0, l1;
      split4(acc[n], h0, h1, l0, l1);
      *reinterpret_cast<uintx2*>(p.o_hi + (size_t)tok * p.ldp + col) = uintx2{h0, h1};
      *reinterpret_cast<uintx2*>(p.o_lo + (size_t)tok * p.ldp + col) = uintx2{l0, l1};
    }
  }
}

// The narrow shapes (NT2 <= 8: NAFNet's c = 64 / 128) keep the per-tile guarded form: measured 374 us against 444-461 us for the
// batched form at 1408 x 2048 x 64 (a pure streaming pass whose two 16-column tile pairs are half "last pair").
template <int NT2, bool XR2>
__device__ __forceinline__ void tok_epilogue_guarded(const TokArgs& p, floatx4 (&acc)[NT2], const floatx4 (&xr)[NT2 / 2][2], long long tok,
                                             bool tok_ok, size_t row, int q) {
  const int N = p.N;
#pragma unroll
  for (int n = 0; n < NT2; ++n) {
    const int col = 32 * (n >> 1) + 8 * q + 4 * (n & 1);
    floatx4 v = acc[n];
    if (col < N) {            // N % 4 == 0: the four columns are valid together
      if (p.b2) v += *reinterpret_cast<const floatx4*>(p.b2 + col);
      if (!p.res_is_x) {
        floatx4 cs = {p.cscale, p.cscale, p.cscale, p.cscale};
        if (p.cvec) cs *= *reinterpret_cast<const floatx4*>(p.cvec + col);
        v *= cs;
        if (p.res) {
          floatx4 rs = {p.rscale, p.rscale, p.rscale, p.rscale};
          if (p.rvec) rs *= *reinterpret_cast<const floatx4*>(p.rvec + col);
          v += *reinterpret_cast<const floatx4*>(p.res + row * p.ldr + col) * rs;
        }
      }
    } else {
      v = floatx4{0.f, 0.f, 0.f, 0.f};
    }
    acc[n] = v;
    if (p.out_pre_ln && tok_ok && col < N) *reinterpret_cast<floatx4*>(p.out + (size_t)tok * p.ldo + col) = v;
  }
  if (p.g2) {       // LayerNorm over the N output columns of the token (two-pass), affine, second residual
    float s1 = 0.f;
#pragma unroll
    for (int n = 0; n < NT2; ++n) s1 += (acc[n][0] + acc[n][1]) + (acc[n][2] + acc[n][3]);
    s1 += __shfl_xor(s1, 16, 64);
    s1 += __shfl_xor(s1, 32, 64);
    const float mean = s1 / (float)N;
    float s2 = 0.f;
#pragma unroll
    for (int n = 0; n < NT2; ++n) {
      const bool in = 32 * (n >> 1) + 8 * q + 4 * (n & 1) < N;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float d = in ? acc[n][c] - mean : 0.f;
        acc[n][c] = d;
        s2 = fmaf(d, d, s2);
      }
    }
    s2 += __shfl_xor(s2, 16, 64);
    s2 += __shfl_xor(s2, 32, 64);
    const float rstd = rsqrtf(s2 / (float)N + p.eps2);
#pragma unroll
    for (int n = 0; n < NT2; ++n) {
      const int col = 32 * (n >> 1) + 8 * q + 4 * (n & 1);
      if (col < N) {
        floatx4 v = acc[n] * rstd * *reinterpret_cast<const floatx4*>(p.g2 + col) + *reinterpret_cast<const floatx4*>(p.be2 + col);
        if constexpr (XR2) v += xr[n >> 1][n & 1];
        else if (p.res2) v += *reinterpret_cast<const floatx4*>(p.res2 + row * p.ldr2 + col);
        acc[n] = v;
      }
    }
  }
  if (!tok_ok) return;
#pragma unroll
  for (int n = 0; n < NT2; ++n) {
    const int col = 32 * (n >> 1) + 8 * q + 4 * (n & 1);
    if (p.out && !p.out_pre_ln && col < N) *reinterpret_cast<floatx4*>(p.out + (size_t)tok * p.ldo + col) = acc[n];
    if (p.o_hi && col < p.ldp) {      // columns N .. ldp-1 of the planes are written as zeros (acc is zero there)
      unsigned h0, h1, l0, l1;
      split4(acc[n], h0, h1, l0, l1);
      *reinterpret_cast<uintx2*>(p.o_hi + (size_t)tok * p.ldp + col) = uintx2{h0, h1};
      *reinterpret_cast<uintx2*>(p.o_lo + (size_t)tok * p.ldp + col) = uintx2{l0, l1};
    }
  }
}

template <int NT2, bool XR2>
__device__ __forceinline__ void tok_epilogue(const TokArgs& p, floatx4 (&acc)[NT2], const floatx4 (&xr)[NT2 / 2][2], long long tok,
                                             bool tok_ok, size_t row, int q) {
  if constexpr (NT2 <= 8) tok_epilogue_guarded<NT2, XR2>(p, acc, xr, tok, tok_ok, row, q);
  else tok_epilogue_body<NT2, XR2, false>(p, acc, xr, tok, tok_ok, row, q);
}

// The head's result row (bias already added; tile (s, g) of lane (token, q) = columns 32 s + 8 q + 4 g + 0..3):
//   hv = LN0?(hv) + hres + hres2 * hvec2[batch];  columns >= K stay zero (they are the k padding of the chain's first GEMM)
// HPF: the hres rows were prefetched into hr together with the head's input rows (narrow shapes, where a second exposed load
// phase per tile costs more than the 2 KS1 x 4 registers).
template <int KS1, bool HPF>
__device__ __forceinline__ void tok_head_epilogue(const TokArgs& p, floatx4 (&hv)[KS1][2], const floatx4 (&hr)[KS1][2], size_t row, int q) {
  const int N = p.K;
  if (p.g0) {
    float s1 = 0.f;
#pragma unroll
    for (int s = 0; s < KS1; ++s)
#pragma unroll
      for (int h = 0; h < 2; ++h) s1 += (hv[s][h][0] + hv[s][h][1]) + (hv[s][h][2] + hv[s][h][3]);   // padding columns are exact zeros
    s1 += __shfl_xor(s1, 16, 64);
    s1 += __shfl_xor(s1, 32, 64);
    const float mean = s1 / (float)N;
    float s2 = 0.f;
#pragma unroll
    for (int s = 0; s < KS1; ++s)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bool in = 32 * s + 8 * q + 4 * h < N;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float d = in ? hv[s][h][c] - mean : 0.f;
          hv[s][h][c] = d;
          s2 = fmaf(d, d, s2);
        }
      }
    s2 += __shfl_xor(s2, 16, 64);
    s2 += __shfl_xor(s2, 32, 64);
    const float rstd = rsqrtf(s2 / (float)N + p.eps0);
#pragma unroll
    for (int s = 0; s < KS1; ++s)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int col = 32 * s + 8 * q + 4 * h;
        const bool in = s < KS1 - 1 ? true : col < N;
        const int c0 = in ? col : 0;
        const floatx4 v = hv[s][h] * rstd * *reinterpret_cast<const floatx4*>(p.g0 + c0) + *reinterpret_cast<const floatx4*>(p.be0 + c0);
        if (in) hv[s][h] = v;
      }
  }
  // (unconditional loads with clamped columns inside the wave-uniform pointer tests, see tok_epilogue)
  asm volatile("" ::: "memory");
  if constexpr (HPF) {
#pragma unroll
    for (int s = 0; s < KS1; ++s)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if (s < KS1 - 1 || 32 * s + 8 * q + 4 * h < N) hv[s][h] += hr[s][h];
  } else if (p.hres) {
    const float* rr = p.hres + row * p.ldhr;
#pragma unroll
    for (int s = 0; s < KS1; ++s)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int col = 32 * s + 8 * q + 4 * h;
        const bool in = s < KS1 - 1 ? true : col < N;       // (N > 32 (KS1 - 1): only the last step is partial)
        const floatx4 r = *reinterpret_cast<const floatx4*>(rr + (in ? col : 0));
        if (in) hv[s][h] += r;
      }
  }
  if (p.hres2) {
    const float* rr = p.hres2 + row * p.ldhr2;
    const float* v2 = p.hvec2 ? p.hvec2 + (row / (size_t)p.rows_per_batch) * N : nullptr;
#pragma unroll
    for (int s = 0; s < KS1; ++s)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int col = 32 * s + 8 * q + 4 * h;
        const bool in = s < KS1 - 1 ? true : col < N;
        const int c0 = in ? col : 0;
        floatx4 r = *reinterpret_cast<const floatx4*>(rr + c0);
        if (v2) r *= *reinterpret_cast<const floatx4*>(v2 + c0);
        if (in) hv[s][h] += r;
      }
  }
}

// MODE 3's input row: the sum of the xdirs partial rows (order of mambair_arch.py:381), LayerNorm with affine, * silu(z).
template <int KS0>
__device__ __forceinline__ void tok_gate_prologue(const TokArgs& p, const float* xr, size_t row, int q, floatx4 (&xv)[KS0][2]) {
  const int K = p.K0;
  // (compiler barriers between the partial rows: hoisting all 4 x 24 loads of a lane in front of the adds needs ~400 registers
  //  and spills under the 256 of two waves per SIMD; one row in flight next to the running sum fits)
  if (p.xdirs == 4) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      asm volatile("" ::: "memory");
      floatx4 t[KS0][2];
      tok_load_rows<KS0, false>(xr + (d == 0 ? 2 : d == 1 ? 1 : 3) * p.xstride, K, q, t);
#pragma unroll
      for (int s = 0; s < KS0; ++s) { xv[s][0] += t[s][0]; xv[s][1] += t[s][1]; }
    }
    asm volatile("" ::: "memory");
  }
  if (p.pg) {
    float s1 = 0.f;
#pragma unroll
    for (int s = 0; s < KS0; ++s)
#pragma unroll
      for (int h = 0; h < 2; ++h) s1 += (xv[s][h][0] + xv[s][h][1]) + (xv[s][h][2] + xv[s][h][3]);
    s1 += __shfl_xor(s1, 16, 64);
    s1 += __shfl_xor(s1, 32, 64);
    const float mean = s1 / (float)K;
    float s2 = 0.f;
#pragma unroll
    for (int s = 0; s < KS0; ++s)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bool in = 32 * s + 8 * q + 4 * h < K;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float d = in ? xv[s][h][c] - mean : 0.f;
          xv[s][h][c] = d;
          s2 = fmaf(d, d, s2);
        }
      }
    s2 += __shfl_xor(s2, 16, 64);
    s2 += __shfl_xor(s2, 32, 64);
    const float rstd = 1.0f / sqrtf(s2 / (float)K + p.peps);
#pragma unroll
    for (int s = 0; s < KS0; ++s)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int col = 32 * s + 8 * q + 4 * h;
        if (col < K)
          xv[s][h] = xv[s][h] * rstd * *reinterpret_cast<const floatx4*>(p.pg + col) + *reinterpret_cast<const floatx4*>(p.pb + col);
      }
  }
  if (p.z) {
    floatx4 t[KS0][2];
    tok_load_rows<KS0, false>(p.z + row * p.ldz, K, q, t);
#pragma unroll
    for (int s = 0; s < KS0; ++s)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int c = 0; c < 4; ++c) xv[s][h][c] *= t[s][h][c] * __builtin_amdgcn_rcpf(1.0f + __expf(-t[s][h][c]));   // (1 ulp rcp: no IEEE division sequence)
  }
}

// MODE 0: hidden = GELU(W1 x + b1)            (G = 2 tiles of 16 hidden features per 32-deep step of the second GEMM)
// MODE 1: hidden = (W1a x + b1a) * (W1b x + b1b)   (SimpleGate; G = 4: tiles 0,1 = first halves, 2,3 = second halves)
// MODE 2: no second GEMM: out = act(W1 pre(x) + b1) * cvec * cscale, 32 output features per step, stored per step (the rows of
//         W1 are packed in the lane-column order, so a lane stores 8 consecutive columns per step); NT2 is unused (2)
// MODE 3: the HEAD alone (below) followed by the chain's epilogue; NT2 = 2 KS1.
// RX: the residual is the input row itself (accumulators start from it; requires K == N, NT2 == 2 KS1)
// TERMS 1 (plain-bf16 mode): one MFMA per product on the hi halves of the fragments.
// KS0 > 0: HEAD -- a linear layer (KS0 k steps of 32 -> the chain's K columns) in front of the chain, see TokArgs; the loaded
// rows are the head's input, the chain runs on the head's result row, which never leaves the registers.
template <int KS1, int NT2, int WAVES, int D, int MODE, bool RX, bool PFETCH, int TERMS, int KS0>
__global__ __launch_bounds__(WAVES * 64) void tok_chain_kernel(TokArgs p) {
  constexpr bool HEAD = KS0 > 0;
  constexpr int KSL = HEAD ? KS0 : KS1;            // k steps of the rows loaded from HBM
  constexpr int G = MODE == 1 ? 4 : 2;
  constexpr int F0 = HEAD ? 2 * KS0 * 2 : 0;       // 1 KB pieces of a head fill (two 16-feature tiles x KS0 k steps x hi / lo)
  constexpr int F1 = MODE == 3 ? 0 : G * KS1 * 2;  // ... of a W1 fill
  constexpr int F2 = MODE >= 2 ? 0 : NT2 * 2;      // ... of a W2 fill
  constexpr int F12 = F1 > F2 ? F1 : F2;
  constexpr int FMAX = F0 > F12 ? F0 : F12;
  static_assert(MODE != 3 || (HEAD && NT2 == 2 * KS1), "MODE 3 is the head alone");
  constexpr int SLOT = FMAX * 1024;
  constexpr int PPW = (FMAX + WAVES - 1) / WAVES;  // LDS-DMA instructions per wave and fill (padded with dummy pieces)
  static_assert(!RX || NT2 == 2 * KS1, "residual-from-input needs the same column groups on both sides");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // D slots | WAVES dummy KB | b1 [steps * G * 16] floats
  unsigned char* const dummy = smem + D * SLOT;
  float* const b1s = reinterpret_cast<float*>(dummy + WAVES * 1024);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4;
  const int hsteps = HEAD ? KS1 : 0;                                    // head fills come first in a tile's fill sequence
  const int nfill = hsteps + (MODE == 2 ? p.steps : MODE == 3 ? 0 : 2 * p.steps) + p.tsteps;   // + the optional tail GEMM's
  const int ntile = (p.M + 16 * WAVES - 1) / (16 * WAVES);
  const int my_tiles = (ntile - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // tiles blockIdx.x, + gridDim.x, ...
  const int total_fills = my_tiles * nfill;

  // fill fg (counted over all tiles of this workgroup): f = fg % nfill; even f = W1 of step f/2, odd = W2 of step f/2 -> slot
  // fg % D.  Every wave issues PPW pieces per fill (pieces past the fill's size and fills past the end go to the wave's dummy
  // KB: the vmcnt bookkeeping stays uniform).
  auto issue_fill = [&](int fg) {
#if !(FFSR_TOK_ABL & 8)
    const bool live = fg < total_fills;
    const int ft = fg % nfill;
    const bool head = HEAD && ft < hsteps;
    const int f = ft - hsteps;                       // index within the chain's own fills
    const int nmain = nfill - hsteps - p.tsteps;
    const bool tail = !head && f >= nmain;           // (tail fills have the W1 shape of 2 tiles: F3 = 2 KS1 2 pieces)
    const bool second = MODE != 2 && !head && !tail && (f & 1);
    constexpr int F3 = 2 * KS1 * 2;
    const int npiece = head ? F0 : tail ? F3 : (second ? F2 : F1);
    const unsigned char* src0 = head ? p.w0 + (size_t)ft * (F0 * 1024)
                              : tail ? p.w3 + (size_t)(f - nmain) * (F3 * 1024)
                              : MODE == 2 ? p.w1 + (size_t)f * (F1 * 1024)
                              : (second ? p.w2 + (size_t)(f >> 1) * (F2 * 1024) : p.w1 + (size_t)(f >> 1) * (F1 * 1024));
    unsigned char* slot = smem + (fg % D) * SLOT;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int i = wave + WAVES * j;
      const bool ok = live && i < npiece;
      const unsigned char* src = (ok ? src0 + (size_t)i * 1024 : p.w1) + lane * 16;
      unsigned char* dst = ok ? slot + i * 1024 : dummy + wave * 1024;
      __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)dst, 16, 0, 0);
    }
#endif
  };

#pragma unroll
  for (int f = 0; f < D - 1; ++f) issue_fill(f);

  // hidden bias -> LDS (read per step with one ds_read_b128 per tile; a global load inside the loop would make the compiler
  // drain the LDS-DMA ring with vmcnt(0))
  for (int i = threadIdx.x; i < p.steps * G * 16; i += WAVES * 64) b1s[i] = p.b1[i];
  float* const b3s = b1s + p.steps * G * 16;
  for (int i = threadIdx.x; i < p.tsteps * 32; i += WAVES * 64) b3s[i] = p.b3[i];
  float* const b0s = b3s + p.tsteps * 32;
  for (int i = threadIdx.x; i < hsteps * 32; i += WAVES * 64) b0s[i] = p.b0[i];
  const int KL = HEAD ? p.K0 : p.K;                 // channels of the loaded rows

  auto tile_row = [&](int tile, long long& tok, bool& ok) -> size_t {
    tok = ((long long)tile * WAVES + wave) * 16 + (lane & 15);
    ok = tok < p.M;
    return (size_t)(ok ? tok : p.M - 1);
  };

  constexpr bool HPF = HEAD && PFETCH && MODE != 3 && KS1 <= 4;      // the head's residual rows travel with its input rows
  floatx4 xv[KSL][2];      // the rows of the tile about to be computed (fp32)
  floatx4 hr[KS1][2];      // HPF: the hres rows of that tile (zeros without an hres)
  {
    long long t_;
    bool o_;
    const size_t r_ = tile_row(blockIdx.x, t_, o_);
    tok_load_rows<KSL, !HEAD>(p.x + r_ * p.ldx, KL, q, xv);
    if constexpr (HPF) {
      if (p.hres) tok_load_rows<KS1, false>(p.hres + r_ * p.ldhr, p.K, q, hr);
      else {
#pragma unroll
        for (int s = 0; s < KS1; ++s) hr[s][0] = hr[s][1] = floatx4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the bias stores (before the first barrier)

  int fbase = 0;           // global fill index of this tile's fill 0
  for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x, fbase += nfill) {
    long long tok;
    bool tok_ok;
    const size_t row = tile_row(tile, tok, tok_ok);
    bf16x8 a_hi[KS1], a_lo[KS1];
    floatx4 hv[KS1][2];     // the chain's input row: the loaded row, or the head's result
    if constexpr (!HEAD) {
      tok_prepare<KS1>(p.K, p.pre_ln != 0, p.eps1, q, xv, a_hi, a_lo);
#pragma unroll
      for (int s = 0; s < KS1; ++s) {
        hv[s][0] = xv[s][0];
        hv[s][1] = xv[s][1];
      }
    }
    // PFETCH: the next tile's rows start their way from HBM now and are first touched after this tile's epilogue (48 .. 80
    // registers in flight; the wide shapes, which would spill, fetch them after the epilogue instead)
    auto fetch_next = [&]() {
      // (unconditional: past the last tile it re-reads a valid row nobody uses -- a conditional load would keep the OLD rows
      //  alive through the whole tile as the other arm of the merge, 48 .. 80 registers)
      const int nt = min(tile + (int)gridDim.x, ntile - 1);
      long long t_;
      bool o_;
      const size_t r_ = tile_row(nt, t_, o_);
      tok_load_rows<KSL, !HEAD>(p.x + r_ * p.ldx, KL, q, xv);
      if constexpr (HPF) {
        if (p.hres) tok_load_rows<KS1, false>(p.hres + r_ * p.ldhr, p.K, q, hr);
        else {
#pragma unroll
          for (int s = 0; s < KS1; ++s) hr[s][0] = hr[s][1] = floatx4{0.f, 0.f, 0.f, 0.f};
        }
      }
    };
    if constexpr (!HEAD && PFETCH) fetch_next();

    if constexpr (HEAD) {
      // ---- head GEMM: 32 output features per step (two tiles over all KS0 k steps), results in the chain's operand order
      floatx4 hrc[KS1][2];      // (this tile's residual rows: hr is about to be refilled for the next tile)
      if constexpr (HPF) {
#pragma unroll
        for (int s = 0; s < KS1; ++s) {
          hrc[s][0] = hr[s][0];
          hrc[s][1] = hr[s][1];
        }
      }
      {
        bf16x8 i_hi[KS0], i_lo[KS0];
        if constexpr (MODE == 3) {
          if (p.xdirs == 4 || p.pg || p.z) tok_gate_prologue<KS0>(p, p.x + row * p.ldx, row, q, xv);
        } else {
          if (p.hxs) {
            const float* v = p.hxs + (row / (size_t)p.rows_per_batch) * p.K0;
#pragma unroll
            for (int s = 0; s < KS0; ++s)
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                const int col = 32 * s + 8 * q + 4 * h;
                if (col < p.K0) xv[s][h] *= *reinterpret_cast<const floatx4*>(v + col);
              }
          }
        }
        tok_prepare<KS0>(p.K0, false, 0.f, q, xv, i_hi, i_lo);
        if constexpr (PFETCH) fetch_next();
#pragma unroll
        for (int hs = 0; hs < KS1; ++hs) {
#if !(FFSR_TOK_ABL & 4)
          wait_vm<(D - 2) * PPW>();
          __builtin_amdgcn_s_barrier();
#endif
          issue_fill(fbase + hs + D - 1);
          const unsigned char* S0 = smem + ((fbase + hs) % D) * SLOT + lane * 16;
          floatx4 hx[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
          for (int s = 0; s < KS0; ++s) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              const bf16x8 wh = *reinterpret_cast<const bf16x8*>(S0 + ((g * KS0 + s) * 2 + 0) * 1024);
              const bf16x8 wl = *reinterpret_cast<const bf16x8*>(S0 + ((g * KS0 + s) * 2 + 1) * 1024);
              if constexpr (TERMS == 3) {
                hx[g] = mfma16(wh, i_lo[s], hx[g]);
                hx[g] = mfma16(wl, i_hi[s], hx[g]);
              }
              hx[g] = mfma16(wh, i_hi[s], hx[g]);
            }
          }
          const float* bs = b0s + hs * 32 + 4 * q;
          hv[hs][0] = hx[0] + *reinterpret_cast<const floatx4*>(bs);          // (zero-padded rows / bias: columns >= K are 0)
          hv[hs][1] = hx[1] + *reinterpret_cast<const floatx4*>(bs + 16);
        }
      }
      if constexpr (MODE != 3) {
        tok_head_epilogue<KS1, HPF>(p, hv, hrc, row, q);
        tok_prepare<KS1>(p.K, p.pre_ln != 0, p.eps1, q, hv, a_hi, a_lo);
      }
    }
    if constexpr (MODE == 3) {
      floatx4 acc[NT2];
#pragma unroll
      for (int n = 0; n < NT2; ++n) acc[n] = hv[n >> 1][n & 1];
      tok_epilogue<NT2, false>(p, acc, hv, tok, tok_ok, row, q);
      if constexpr (!PFETCH) fetch_next();
      continue;
    }
    floatx4 acc[NT2];
#pragma unroll
    for (int n = 0; n < NT2; ++n) {
      if constexpr (RX) acc[n] = hv[n >> 1][n & 1];       // x + ...: the residual is the row we already hold
      else acc[n] = floatx4{0.f, 0.f, 0.f, 0.f};
    }
    const int fb = fbase + hsteps;      // this tile's first chain fill

    if constexpr (MODE == 2) {
      for (int st = 0; st < p.steps; ++st) {
#if !(FFSR_TOK_ABL & 4)
        wait_vm<(D - 2) * PPW>();      // (the stores of the last steps are younger than the fill waited for: conservative)
        __builtin_amdgcn_s_barrier();
#endif
        issue_fill(fb + st + D - 1);
        const unsigned char* S1 = smem + ((fb + st) % D) * SLOT + lane * 16;
        floatx4 hx[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < KS1; ++s) {
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            const bf16x8 wh = *reinterpret_cast<const bf16x8*>(S1 + ((g * KS1 + s) * 2 + 0) * 1024);
            const bf16x8 wl = *reinterpret_cast<const bf16x8*>(S1 + ((g * KS1 + s) * 2 + 1) * 1024);
            if constexpr (TERMS == 3) {
              hx[g] = mfma16(wh, a_lo[s], hx[g]);
              hx[g] = mfma16(wl, a_hi[s], hx[g]);
            }
            hx[g] = mfma16(wh, a_hi[s], hx[g]);
          }
        }
        // lane (token, q): tile g holds the output columns 32 st + 8 q + 4 g + 0..3
        const float* bs = b1s + st * 32 + 4 * q;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int col = 32 * st + 8 * q + 4 * g;
          floatx4 v = hx[g] + *reinterpret_cast<const floatx4*>(bs + 16 * g);
          if (p.act == FFSR_ACT_GELU) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = gelu_fast(v[c]);
          } else if (p.act != FFSR_ACT_NONE) {      // ReLU / LeakyReLU (slope 0 / given)
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = v[c] > 0.f ? v[c] : v[c] * p.slope;
          }
          if (col < p.N) {
            floatx4 cs = {p.cscale, p.cscale, p.cscale, p.cscale};
            if (p.cvec) cs *= *reinterpret_cast<const floatx4*>(p.cvec + col);
            v *= cs;
          } else {
            v = floatx4{0.f, 0.f, 0.f, 0.f};
          }
          if (tok_ok) {
            if (p.out && col < p.N) *reinterpret_cast<floatx4*>(p.out + (size_t)tok * p.ldo + col) = v;
            if (p.o_hi && col < p.ldp) {
              unsigned h0, h1, l0, l1;
              split4(v, h0, h1, l0, l1);
              *reinterpret_cast<uintx2*>(p.o_hi + (size_t)tok * p.ldp + col) = uintx2{h0, h1};
              *reinterpret_cast<uintx2*>(p.o_lo + (size_t)tok * p.ldp + col) = uintx2{l0, l1};
            }
          }
        }
      }
      if constexpr (!PFETCH) fetch_next();
      continue;
    }
    for (int st = 0; st < p.steps; ++st) {
      // ---- first GEMM of the step: G hidden tiles from slot (fbase + 2 st) % D
#if !(FFSR_TOK_ABL & 4)
      wait_vm<(D - 2) * PPW>();
      __builtin_amdgcn_s_barrier();
#endif
      issue_fill(fb + 2 * st + D - 1);
      const unsigned char* S1 = smem + ((fb + 2 * st) % D) * SLOT + lane * 16;
      floatx4 hx[G];
#pragma unroll
      for (int g = 0; g < G; ++g) hx[g] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS1; ++s) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
#if FFSR_TOK_ABL & 2
          const bf16x8 wh = a_hi[(s + 1) % KS1], wl = a_lo[(s + 1) % KS1];
          asm volatile("" ::"v"(S1));
#else
          const bf16x8 wh = *reinterpret_cast<const bf16x8*>(S1 + ((g * KS1 + s) * 2 + 0) * 1024);
          const bf16x8 wl = *reinterpret_cast<const bf16x8*>(S1 + ((g * KS1 + s) * 2 + 1) * 1024);
#endif
          if constexpr (TERMS == 3) {
            hx[g] = mfma16(wh, a_lo[s], hx[g]);
            hx[g] = mfma16(wl, a_hi[s], hx[g]);
          }
          hx[g] = mfma16(wh, a_hi[s], hx[g]);
        }
      }
      // ---- bias, activation, conversion: the two 16-feature tiles become the B operand of the second GEMM's k step
      const float* bs = b1s + st * (G * 16) + 4 * q;
      floatx4 ha, hb;
      if constexpr (MODE == 1) {
        ha = (hx[0] + *reinterpret_cast<const floatx4*>(bs)) * (hx[2] + *reinterpret_cast<const floatx4*>(bs + 32));
        hb = (hx[1] + *reinterpret_cast<const floatx4*>(bs + 16)) * (hx[3] + *reinterpret_cast<const floatx4*>(bs + 48));
      } else {
        ha = hx[0] + *reinterpret_cast<const floatx4*>(bs);
        hb = hx[1] + *reinterpret_cast<const floatx4*>(bs + 16);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          ha[c] = gelu_fast(ha[c]);
          hb[c] = gelu_fast(hb[c]);
        }
      }
      unsigned hh[4], ll[4];
      split4(ha, hh[0], hh[1], ll[0], ll[1]);
      split4(hb, hh[2], hh[3], ll[2], ll[3]);
      const bf16x8 h_hi = __builtin_bit_cast(bf16x8, uintx4{hh[0], hh[1], hh[2], hh[3]});
      const bf16x8 h_lo = __builtin_bit_cast(bf16x8, uintx4{ll[0], ll[1], ll[2], ll[3]});

      // ---- second GEMM: one 32-deep k step of every output tile from slot (fbase + 2 st + 1) % D
#if !(FFSR_TOK_ABL & 4)
      wait_vm<(D - 2) * PPW>();
      __builtin_amdgcn_s_barrier();
#endif
      issue_fill(fb + 2 * st + D);
      const unsigned char* S2 = smem + ((fb + 2 * st + 1) % D) * SLOT + lane * 16;
#pragma unroll
      for (int n = 0; n < NT2; ++n) {
#if FFSR_TOK_ABL & 2
        const bf16x8 wh = a_hi[n % KS1], wl = a_lo[n % KS1];
        asm volatile("" ::"v"(S2));
#else
        const bf16x8 wh = *reinterpret_cast<const bf16x8*>(S2 + (n * 2 + 0) * 1024);
        const bf16x8 wl = *reinterpret_cast<const bf16x8*>(S2 + (n * 2 + 1) * 1024);
#endif
        if constexpr (TERMS == 3) {
          acc[n] = mfma16(wh, h_lo, acc[n]);
          acc[n] = mfma16(wl, h_hi, acc[n]);
        }
        acc[n] = mfma16(wh, h_hi, acc[n]);
      }
    }
    if constexpr (NT2 == 2 * KS1) tok_epilogue<NT2, HEAD && !RX>(p, acc, hv, tok, tok_ok, row, q);
    else {
      floatx4 none[NT2 / 2][2];
      tok_epilogue<NT2, false>(p, acc, none, tok, tok_ok, row, q);
    }
    if constexpr (MODE != 2 && NT2 == 2 * KS1) {
      if (p.tsteps) {
        // ---- tail: the chain's output row (acc, already in operand order: tile pair (2 s, 2 s + 1) = k step s) feeds one more GEMM
        bf16x8 t_hi[KS1], t_lo[KS1];
#pragma unroll
        for (int s = 0; s < KS1; ++s) {
          unsigned hh[4], ll[4];
          split4(acc[2 * s], hh[0], hh[1], ll[0], ll[1]);
          split4(acc[2 * s + 1], hh[2], hh[3], ll[2], ll[3]);
          t_hi[s] = __builtin_bit_cast(bf16x8, uintx4{hh[0], hh[1], hh[2], hh[3]});
          t_lo[s] = __builtin_bit_cast(bf16x8, uintx4{ll[0], ll[1], ll[2], ll[3]});
        }
        const int fmain = fbase + nfill - p.tsteps;
        for (int ts = 0; ts < p.tsteps; ++ts) {
#if !(FFSR_TOK_ABL & 4)
          wait_vm<(D - 2) * PPW>();
          __builtin_amdgcn_s_barrier();
#endif
          issue_fill(fmain + ts + D - 1);
          const unsigned char* S3 = smem + ((fmain + ts) % D) * SLOT + lane * 16;
          floatx4 hx[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
          for (int s = 0; s < KS1; ++s) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              const bf16x8 wh = *reinterpret_cast<const bf16x8*>(S3 + ((g * KS1 + s) * 2 + 0) * 1024);
              const bf16x8 wl = *reinterpret_cast<const bf16x8*>(S3 + ((g * KS1 + s) * 2 + 1) * 1024);
              if constexpr (TERMS == 3) {
                hx[g] = mfma16(wh, t_lo[s], hx[g]);
                hx[g] = mfma16(wl, t_hi[s], hx[g]);
              }
              hx[g] = mfma16(wh, t_hi[s], hx[g]);
            }
          }
          const float* bs = b3s + ts * 32 + 4 * q;
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            const int col = 32 * ts + 8 * q + 4 * g;
            floatx4 v = hx[g] + *reinterpret_cast<const floatx4*>(bs + 16 * g);
            if (p.act3 != FFSR_ACT_NONE) {
#pragma unroll
              for (int c = 0; c < 4; ++c) v[c] = v[c] > 0.f ? v[c] : v[c] * p.slope3;
            }
            v *= p.cscale3;
            if (tok_ok && col < p.N3) {
              if (p.res3) v += *reinterpret_cast<const floatx4*>(p.res3 + row * p.ldr3 + col) * p.rscale3;
              *reinterpret_cast<floatx4*>(p.out3 + (size_t)tok * p.ldo3 + col) = v;
            }
          }
        }
      }
    }
    if constexpr (!PFETCH) fetch_next();
  }
  wait_vm<0>();     // the dummy pieces of the last fills
}

template <int KS1, int NT2, int WAVES, int D, int MODE, bool RX, bool PFETCH, int TERMS, int KS0>
int launch_tok4(const TokArgs& a, hipStream_t st) {
  constexpr int G = MODE == 1 ? 4 : 2;
  constexpr int F0 = 4 * KS0, F1 = MODE == 3 ? 0 : G * KS1 * 2, F2 = MODE >= 2 ? 0 : NT2 * 2;
  constexpr int F12 = F1 > F2 ? F1 : F2, FMAX = F0 > F12 ? F0 : F12;
  const int lds = D * FMAX * 1024 + WAVES * 1024 + a.steps * G * 16 * 4 + a.tsteps * 32 * 4 + (KS0 > 0 ? KS1 * 32 * 4 : 0);
  if (lds > 160 * 1024) return FFSR_EINVAL;
  static unsigned long long attr_set = 0;
  static int num_cu = 0;
  const void* fn = reinterpret_cast<const void*>(&tok_chain_kernel<KS1, NT2, WAVES, D, MODE, RX, PFETCH, TERMS, KS0>);
  if (ffsr_allow_dynamic_lds(&fn, 1, 160 * 1024, &attr_set) != FFSR_OK) return FFSR_ELAUNCH;
  if (!num_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return FFSR_ELAUNCH;
    num_cu = prop.multiProcessorCount;
  }
  const int per = WAVES * 16;
  const int ntile = (a.M + per - 1) / per;
  const int wg_per_cu = lds <= 80 * 1024 ? 2 : 1;     // small rings (NAFNet widths): two workgroups share a CU
  const int grid = ntile < num_cu * wg_per_cu ? ntile : num_cu * wg_per_cu;
  FFSR_LAUNCH((tok_chain_kernel<KS1, NT2, WAVES, D, MODE, RX, PFETCH, TERMS, KS0>), dim3(grid), dim3(WAVES * 64), lds, st, a);
  return ffsr_launch_status();
}

template <int KS1, int NT2, int WAVES, int D, int MODE, bool RX, bool PFETCH, int KS0>
int launch_tok(const TokArgs& a, hipStream_t st) {
  return g_ffsr_gemm_terms == 1 ? launch_tok4<KS1, NT2, WAVES, D, MODE, RX, PFETCH, 1, KS0>(a, st)
                                : launch_tok4<KS1, NT2, WAVES, D, MODE, RX, PFETCH, 3, KS0>(a, st);
}

template <int KS1, int NT2, int MODE, bool RX, int KS0 = 0>
int launch_tok_w(const TokArgs& a, int waves, hipStream_t st) {
  constexpr int G = MODE == 1 ? 4 : 2;
  constexpr int F0 = 4 * KS0, F1 = MODE == 3 ? 0 : G * KS1 * 2, F2 = MODE >= 2 ? 0 : NT2 * 2;
  constexpr int F12 = F1 > F2 ? F1 : F2, FMAX = F0 > F12 ? F0 : F12;
  // ring depth: 4 slots while they fit in ~128 KB; the single-GEMM modes (one fill per short step, stores in the loop) 5
  constexpr int D = (MODE == 2 || MODE == 3) ? (FMAX * 5 <= 140 ? 5 : (FMAX * 4 <= 148 ? 4 : 3)) : (FMAX * 4 <= 128 ? 4 : 3);
  // the next tile's rows are prefetched into registers where that fits the 256 registers of two waves per SIMD
  constexpr bool PF8 = KS0 > 0 ? (RX && KS0 <= 6 && KS1 <= 6) : (MODE == 2 || KS1 <= 8);
  // (measured and dropped: 4-wave workgroups with a 3-slot ring, so that TWO workgroups share a CU and drift apart -- K = 180 MLP
  //  254 -> 243 us, in_proj 234 -> 227 us, but the head kernels need > 256 registers per wave at 4 waves and fall to one workgroup)
  switch (waves) {
    case 4: return launch_tok<KS1, NT2, 4, D, MODE, RX, true, KS0>(a, st);
    case 8: return launch_tok<KS1, NT2, 8, D, MODE, RX, PF8, KS0>(a, st);
    default: return FFSR_EINVAL;
  }
}

}  // namespace

#ifndef FFSR_TOK_ONLY_PROJ   // (analysis builds of the projection kernel alone: hipcc -DFFSR_TOK_ONLY_PROJ -S)
namespace {
struct TokHead {
  const void* w0;
  const float* b0;
  const float* g0;
  const float* be0;
  float eps0;
  const float* hres;
  int ldhr;
  const float* hres2;
  int ldhr2;
  const float* hvec2;
  int rows_per_batch;
  const float* hxs;
};

int tok_chain_common(const float* x, int ldx, const void* w1, const float* b1, const void* w2, const float* b2,
                     const float* cvec, const float* res, int ldr, const float* rvec, const float* g2, const float* be2,
                     const float* res2, int ldr2, float* out, int ldo, void* out_hi, void* out_lo, int ldp, long long M, int K,
                     int N, int steps, int mode, int pre_ln, float eps1, float eps2, float cscale, float rscale, int waves,
                     const void* w3, const float* b3, const float* res3, int ldr3, float* out3, int ldo3, int N3, int act3,
                     float slope3, float cscale3, float rscale3, void* stream, const TokHead* hd = nullptr) {
  const bool tail = w3 != nullptr;
  FFSR_CHECK(x && w1 && b1 && w2 && (out || (out_hi && out_lo) || tail) && M > 0 && M < (1ll << 31));
  if (hd) {       // the loaded rows are the head's input (K0 = K channels); the chain's own input never exists in memory
    FFSR_CHECK(hd->w0 && hd->b0 && ((uintptr_t)hd->w0 & 15) == 0 && K == N && !res && !res2 && !cvec && !rvec);
    FFSR_CHECK(!hd->hxs || (hd->rows_per_batch > 0 && ((uintptr_t)hd->hxs & 15) == 0));
    FFSR_CHECK(!hd->g0 || (hd->be0 && ((uintptr_t)hd->g0 & 15) == 0 && ((uintptr_t)hd->be0 & 15) == 0));
    FFSR_CHECK(!hd->hres || (hd->ldhr >= K && (hd->ldhr & 3) == 0 && ((uintptr_t)hd->hres & 15) == 0));
    FFSR_CHECK(!hd->hres2 || (hd->ldhr2 >= K && (hd->ldhr2 & 3) == 0 && ((uintptr_t)hd->hres2 & 15) == 0));
    FFSR_CHECK(!hd->hvec2 || (hd->hres2 && hd->rows_per_batch > 0 && ((uintptr_t)hd->hvec2 & 15) == 0));
    FFSR_CHECK(cscale == 1.0f && rscale == 1.0f);
  }
  FFSR_CHECK(K > 0 && N > 0 && steps > 0 && (K & 3) == 0 && (N & 3) == 0 && ldx >= K && (ldx & 3) == 0);
  FFSR_CHECK(((uintptr_t)x & 15) == 0 && ((uintptr_t)w1 & 15) == 0 && ((uintptr_t)w2 & 15) == 0 && ((uintptr_t)b1 & 3) == 0);
  FFSR_CHECK(!out || (ldo >= N && (ldo & 3) == 0 && ((uintptr_t)out & 15) == 0));
  FFSR_CHECK(!res || (ldr >= N && (ldr & 3) == 0 && ((uintptr_t)res & 15) == 0));
  FFSR_CHECK(!res2 || (g2 && ldr2 >= N && (ldr2 & 3) == 0 && ((uintptr_t)res2 & 15) == 0));
  FFSR_CHECK(!g2 || be2);
  FFSR_CHECK((!b2 || ((uintptr_t)b2 & 15) == 0) && (!cvec || ((uintptr_t)cvec & 15) == 0) && (!rvec || ((uintptr_t)rvec & 15) == 0) &&
             (!g2 || (((uintptr_t)g2 & 15) == 0 && ((uintptr_t)be2 & 15) == 0)));
  FFSR_CHECK(!out_hi || (out_lo && (ldp & 31) == 0 && ldp >= N && ldp < N + 32 && ((uintptr_t)out_hi & 7) == 0 && ((uintptr_t)out_lo & 7) == 0));
  FFSR_CHECK(mode == 0 || mode == 1);
  FFSR_CHECK(!hd || !g2 || mode == 0);
  if (tail) {
    FFSR_CHECK(b3 && out3 && N3 > 0 && (N3 & 3) == 0 && ldo3 >= N3 && (ldo3 & 3) == 0 && ((uintptr_t)out3 & 15) == 0 && ((uintptr_t)w3 & 15) == 0);
    FFSR_CHECK(!res3 || (ldr3 >= N3 && (ldr3 & 3) == 0 && ((uintptr_t)res3 & 15) == 0));
    FFSR_CHECK(act3 == FFSR_ACT_NONE || act3 == FFSR_ACT_RELU || act3 == FFSR_ACT_LRELU);
    FFSR_CHECK((K + 31) / 32 == (N + 31) / 32);       // the tail contracts over the chain's N outputs with the K-step count of its input
  }
  TokArgs a = {};
  a.x = x; a.w1 = (const unsigned char*)w1; a.b1 = b1; a.w2 = (const unsigned char*)w2; a.b2 = b2; a.cvec = cvec; a.res = res;
  a.rvec = rvec; a.g2 = g2; a.be2 = be2; a.res2 = res2; a.out = out; a.o_hi = (unsigned short*)out_hi; a.o_lo = (unsigned short*)out_lo;
  a.ldx = ldx; a.ldr = ldr; a.ldr2 = ldr2; a.ldo = ldo; a.ldp = ldp; a.M = (int)M; a.K = K; a.N = N; a.steps = steps;
  a.pre_ln = pre_ln; a.eps1 = eps1; a.eps2 = eps2; a.cscale = cscale; a.rscale = rscale; a.act = 0; a.slope = 0.f;
  if (tail) {
    a.w3 = (const unsigned char*)w3; a.b3 = b3; a.res3 = res3; a.out3 = out3; a.ldr3 = ldr3; a.ldo3 = ldo3; a.N3 = N3;
    a.tsteps = (N3 + 31) / 32; a.act3 = act3; a.slope3 = act3 == FFSR_ACT_RELU ? 0.f : slope3; a.cscale3 = cscale3; a.rscale3 = rscale3;
  }
  const int ks1 = (K + 31) / 32, nt2 = (N + 31) / 32 * 2;      // an even number of 16-column tiles
  // x + f(x): the residual is the row the wave holds anyway (no second read), when nothing scales either side
  a.res_is_x = (res == x && ldr == ldx && K == N && !cvec && !rvec && cscale == 1.0f && rscale == 1.0f && nt2 == 2 * ks1) ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  if (waves == 0) waves = 8;
  if (hd) {
    a.w0 = (const unsigned char*)hd->w0; a.b0 = hd->b0; a.g0 = hd->g0; a.be0 = hd->be0; a.hres = hd->hres; a.hres2 = hd->hres2;
    a.hvec2 = hd->hvec2; a.hxs = hd->hxs; a.K0 = K; a.ldhr = hd->ldhr; a.ldhr2 = hd->ldhr2; a.rows_per_batch = hd->rows_per_batch;
    a.hsteps = ks1; a.eps0 = hd->eps0;
    // without a post-LN the chain is x1 + mlp(norm(x1)) (RX: accumulators start from x1); with one it is norm(mlp(x1)) + x1
    a.res_is_x = g2 ? 0 : 1;
    if (mode == 1) {      // NAFNet: conv3 (+ channel attention, beta) in front of the gated half, c = 64 / 128
      if (!a.res_is_x) return FFSR_EINVAL;
      if (ks1 == 2 && nt2 == 4) return launch_tok_w<2, 4, 1, true, 2>(a, waves, st);
      if (ks1 == 4 && nt2 == 8) return launch_tok_w<4, 8, 1, true, 4>(a, waves, st);
      return FFSR_EINVAL;
    }
    if (!a.res_is_x) return (ks1 == 6 && nt2 == 12) ? launch_tok_w<6, 12, 0, false, 6>(a, waves, st) : FFSR_EINVAL;   // GRL: C = 180
#define FFSR_TOK_HEAD_CASE(KS, NT) \
  if (ks1 == KS && nt2 == NT) return launch_tok_w<KS, NT, 0, true, KS>(a, waves, st)
    FFSR_TOK_HEAD_CASE(6, 12);
    FFSR_TOK_HEAD_CASE(7, 14);
    FFSR_TOK_HEAD_CASE(8, 16);
    FFSR_TOK_HEAD_CASE(9, 18);
    FFSR_TOK_HEAD_CASE(10, 20);
#undef FFSR_TOK_HEAD_CASE
    return FFSR_EINVAL;
  }
#define FFSR_TOK_CASE(KS, NT, MD)                                                       \
  if (ks1 == KS && nt2 == NT && mode == MD)                                             \
    return a.res_is_x ? launch_tok_w<KS, NT, MD, true>(a, waves, st) : launch_tok_w<KS, NT, MD, false>(a, waves, st)
  // Swin / GRL MLPs (K = N = 180 .. 308)
  FFSR_TOK_CASE(4, 8, 0);      // the fusion network's collaborative FFN (128 -> 256 -> 128 over 4 expert tokens per pixel)
  FFSR_TOK_CASE(6, 12, 0);
  FFSR_TOK_CASE(7, 14, 0);
  FFSR_TOK_CASE(8, 16, 0);
  FFSR_TOK_CASE(9, 18, 0);
  FFSR_TOK_CASE(10, 20, 0);
  // NAFNet pointwise halves (c = 64, 128)
  FFSR_TOK_CASE(2, 4, 1);
  FFSR_TOK_CASE(4, 8, 1);
#undef FFSR_TOK_CASE
  return FFSR_EINVAL;
}
}  // namespace

// See include/ffsr.h for the contract.
extern "C" int ffsr_tok_chain_f32(const float* x, int ldx, const void* w1, const float* b1, const void* w2, const float* b2,
                                  const float* cvec, const float* res, int ldr, const float* rvec, const float* g2,
                                  const float* be2, const float* res2, int ldr2, float* out, int ldo, void* out_hi,
                                  void* out_lo, int ldp, long long M, int K, int N, int steps, int mode, int pre_ln,
                                  float eps1, float eps2, float cscale, float rscale, int waves, void* stream) {
  return tok_chain_common(x, ldx, w1, b1, w2, b2, cvec, res, ldr, rvec, g2, be2, res2, ldr2, out, ldo, out_hi, out_lo, ldp, M, K, N,
                          steps, mode, pre_ln, eps1, eps2, cscale, rscale, waves, nullptr, nullptr, nullptr, 0, nullptr, 0, 0, 0, 0.f,
                          1.f, 1.f, stream);
}

extern "C" int ffsr_tok_chain_tail_f32(const float* x, int ldx, const void* w1, const float* b1, const void* w2, const float* b2,
                                       const float* res, int ldr, float* out, int ldo, long long M, int K, int N, int steps,
                                       int mode, int pre_ln, float eps1, const void* w3, const float* b3, const float* res3,
                                       int ldr3, float* out3, int ldo3, int N3, int act3, float slope3, float cscale3,
                                       float rscale3, int waves, void* stream) {
  FFSR_CHECK(w3);
  return tok_chain_common(x, ldx, w1, b1, w2, b2, nullptr, res, ldr, nullptr, nullptr, nullptr, nullptr, 0, out, ldo, nullptr, nullptr,
                          0, M, K, N, steps, mode, pre_ln, eps1, 1e-5f, 1.f, 1.f, waves, w3, b3, res3, ldr3, out3, ldo3, N3, act3,
                          slope3, cscale3, rscale3, stream);
}

// See include/ffsr.h for the contract.
extern "C" int ffsr_tok_gemm_f32(const float* x, int ldx, const void* w1, const float* b1, const float* cvec, float* out,
                                 int ldo, void* out_hi, void* out_lo, int ldp, long long M, int K, int N, int pre_ln,
                                 float eps1, int act, float slope, float cscale, int waves, void* stream) {
  FFSR_CHECK(x && w1 && b1 && (out || (out_hi && out_lo)) && M > 0 && M < (1ll << 31));
  FFSR_CHECK(K > 0 && N > 0 && (K & 3) == 0 && (N & 3) == 0 && ldx >= K && (ldx & 3) == 0);
  FFSR_CHECK(((uintptr_t)x & 15) == 0 && ((uintptr_t)w1 & 15) == 0 && ((uintptr_t)b1 & 3) == 0);
  FFSR_CHECK(!out || (ldo >= N && (ldo & 3) == 0 && ((uintptr_t)out & 15) == 0));
  FFSR_CHECK(!cvec || ((uintptr_t)cvec & 15) == 0);
  FFSR_CHECK(!out_hi || (out_lo && (ldp & 31) == 0 && ldp >= N && ldp < N + 32 && ((uintptr_t)out_hi & 7) == 0 && ((uintptr_t)out_lo & 7) == 0));
  FFSR_CHECK(act == FFSR_ACT_NONE || act == FFSR_ACT_GELU || act == FFSR_ACT_RELU || act == FFSR_ACT_LRELU);
  TokArgs a = {};
  a.x = x; a.w1 = (const unsigned char*)w1; a.b1 = b1; a.cvec = cvec; a.out = out;
  a.o_hi = (unsigned short*)out_hi; a.o_lo = (unsigned short*)out_lo;
  a.ldx = ldx; a.ldo = ldo; a.ldp = ldp; a.M = (int)M; a.K = K; a.N = N; a.steps = (N + 31) / 32;
  a.pre_ln = pre_ln; a.eps1 = eps1; a.cscale = cscale; a.rscale = 1.f; a.act = act;
  a.slope = act == FFSR_ACT_RELU ? 0.f : slope;
  const int ks1 = (K + 31) / 32;
  hipStream_t st = (hipStream_t)stream;
  if (waves == 0) waves = 8;
  switch (ks1) {
    case 2: return launch_tok_w<2, 2, 2, false>(a, waves, st);
    case 4: return launch_tok_w<4, 2, 2, false>(a, waves, st);
    case 6: return launch_tok_w<6, 2, 2, false>(a, waves, st);
    case 7: return launch_tok_w<7, 2, 2, false>(a, waves, st);
    case 8: return launch_tok_w<8, 2, 2, false>(a, waves, st);
    case 9: return launch_tok_w<9, 2, 2, false>(a, waves, st);
    case 10: return launch_tok_w<10, 2, 2, false>(a, waves, st);
    case 12: return launch_tok_w<12, 2, 2, false>(a, waves, st);      // MambaIR x_proj: d_inner 360 -> 4 x (dt_rank + 2 d_state)
    default: return FFSR_EINVAL;
  }
}

// See include/ffsr.h for the contract.
extern "C" int ffsr_tok_head_chain_f32(const float* a, int lda, const float* ascale, const void* w0, const float* b0, const float* g0,
                                       const float* be0, float eps0, const float* hres, int ldhr, const float* hres2, int ldhr2,
                                       const float* hvec2, int rows_per_batch, const void* w1, const float* b1, const void* w2,
                                       const float* b2, const float* g2, const float* be2, float eps2, float* out, int ldo,
                                       void* out_hi, void* out_lo, int ldp, long long M, int K, int steps, int mode, int pre_ln,
                                       float eps1,
                                       const void* w3, const float* b3, const float* res3, int ldr3, float* out3, int ldo3,
                                       int N3, int act3, float slope3, float cscale3, float rscale3, int waves, void* stream) {
  FFSR_CHECK(w0 && b0);
  const TokHead hd = {w0, b0, g0, be0, eps0, hres, ldhr, hres2, ldhr2, hvec2, rows_per_batch, ascale};
  return tok_chain_common(a, lda, w1, b1, w2, b2, nullptr, nullptr, 0, nullptr, g2, be2, nullptr, 0, out, ldo, out_hi, out_lo, ldp, M,
                          K, K, steps, mode, pre_ln, eps1, eps2, 1.f, 1.f, waves, w3, b3, res3, ldr3, out3, ldo3, N3, act3, slope3,
                          cscale3, rscale3, stream, &hd);
}

#endif  // FFSR_TOK_ONLY_PROJ
// See include/ffsr.h for the contract.
extern "C" int ffsr_tok_proj_f32(const float* x, long long xstride, int xdirs, int ldx, const float* z, int ldz, const float* pg,
                                 const float* pb, float peps, const void* w0, const float* b0, const float* cvec,
                                 const float* res, int ldr, const float* rvec, const float* g2, const float* be2, float eps2,
                                 float* out, int ldo, int out_pre_ln, void* out_hi, void* out_lo, int ldp, long long M, int K,
                                 int N, float cscale, float rscale, int waves, void* stream) {
  FFSR_CHECK(x && w0 && b0 && (out || (out_hi && out_lo)) && M > 0 && M < (1ll << 31) && (xdirs == 1 || xdirs == 4));
  FFSR_CHECK(K > 0 && N > 0 && (K & 3) == 0 && (N & 3) == 0 && ldx >= K && (ldx & 3) == 0 && (xstride & 3) == 0);
  FFSR_CHECK(((uintptr_t)x & 15) == 0 && ((uintptr_t)w0 & 15) == 0 && ((uintptr_t)b0 & 3) == 0);
  FFSR_CHECK(!z || (ldz >= K && (ldz & 3) == 0 && ((uintptr_t)z & 15) == 0));
  FFSR_CHECK(!pg || (pb && ((uintptr_t)pg & 15) == 0 && ((uintptr_t)pb & 15) == 0));
  FFSR_CHECK(!out || (ldo >= N && (ldo & 3) == 0 && ((uintptr_t)out & 15) == 0));
  FFSR_CHECK(!res || (ldr >= N && (ldr & 3) == 0 && ((uintptr_t)res & 15) == 0));
  FFSR_CHECK((!cvec || ((uintptr_t)cvec & 15) == 0) && (!rvec || ((uintptr_t)rvec & 15) == 0));
  FFSR_CHECK(!g2 || (be2 && ((uintptr_t)g2 & 15) == 0 && ((uintptr_t)be2 & 15) == 0));
  FFSR_CHECK(!out_pre_ln || (out && g2));
  FFSR_CHECK(!out_hi || (out_lo && (ldp & 31) == 0 && ldp >= N && ldp < N + 32 && ((uintptr_t)out_hi & 7) == 0 && ((uintptr_t)out_lo & 7) == 0));
  TokArgs a = {};
  a.x = x; a.xstride = xstride; a.xdirs = xdirs; a.ldx = ldx; a.z = z; a.ldz = ldz; a.pg = pg; a.pb = pb; a.peps = peps;
  a.w0 = (const unsigned char*)w0; a.b0 = b0; a.w1 = a.w0; a.cvec = cvec; a.res = res; a.ldr = ldr; a.rvec = rvec; a.g2 = g2; a.be2 = be2;
  a.eps2 = eps2; a.out = out; a.ldo = ldo; a.out_pre_ln = out_pre_ln; a.o_hi = (unsigned short*)out_hi; a.o_lo = (unsigned short*)out_lo;
  a.ldp = ldp; a.M = (int)M; a.K0 = K; a.K = N; a.N = N; a.cscale = cscale; a.rscale = rscale;
  const int ks0 = (K + 31) / 32, ks1 = (N + 31) / 32;
  a.hsteps = ks1;
  hipStream_t st = (hipStream_t)stream;
  if (waves == 0) waves = 8;
  if (ks0 == 12 && ks1 == 6) return launch_tok_w<6, 12, 3, false, 12>(a, waves, st);     // MambaIR out_proj: 360 -> 180
  return FFSR_EINVAL;
}
