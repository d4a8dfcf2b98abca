// Backward pass of the fusion network (the cached-feature training step, train.py:323-336 -> loss.backward()).
// Streaming / reduction kernels; the convolution weight gradient (MFMA) lives in ffsr_wgrad.hip.
//
// Conventions as in the forward library: channels-last fp32 maps [rows, C] with a row stride ld, caller-owned buffers,
// nothing allocated inside.  Every reduction is a deterministic two-stage sum (per-block partials in caller-owned
// scratch, then one finishing block); parameter gradients are ACCUMULATED into their destination (dst += value) so that
// modules applied several times per step (the shared LKA block: 9 bands / 4 experts) and gradient accumulation over
// micro-batches (train.py:332-335) need no extra pass.
#include "ffsr_common.h"

namespace {

constexpr int RB = 256;

inline int grid_for(long long n, int per = RB) { return (int)((n + per - 1) / per); }

// ---------------------------------------------------------------------------------------------- weight repacking
// src: nn.Conv2d layout [N, Cin, KH, KW].  Packed row r, column t * Cp + c (tap-major, Cp channels per tap):
//   transpose = 0 (forward operator):  r = n, value = src[n, c, t]                         rows = N,   channels = Cin
//   transpose = 1 (input gradient):    r = c, value = src[n, c, T - 1 - t] at channel n    rows = Cin, channels = N
// (the input gradient of a stride-1 "same" convolution is the convolution of dY with the spatially flipped, transposed
// kernel).  Writes the fp32 matrix [rows, T * Cp] (dst_f32, optional) and / or the zero-padded bf16 hi / lo planes
// [rows_pad, ldw] of the split-bf16 kernels (optional).
__global__ void pack_conv_kernel(const float* __restrict__ src, int N, int Cin, int T, int transpose, float* __restrict__ dst_f32,
                                 int Cp, unsigned short* __restrict__ hi, unsigned short* __restrict__ lo, int rows_pad, int ldw) {
  const int rows = transpose ? Cin : N, ch = transpose ? N : Cin;
  const int K = T * Cp;
  const int ncol = hi ? ldw : K, nrow = hi ? rows_pad : rows;
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)nrow * ncol) return;
  const int r = (int)(idx / ncol), k = (int)(idx % ncol);
  float v = 0.f;
  if (r < rows && k < K) {
    const int t = k / Cp, c = k % Cp;
    if (c < ch) v = transpose ? src[((size_t)c * Cin + r) * T + (T - 1 - t)] : src[((size_t)r * Cin + c) * T + t];
    if (dst_f32) dst_f32[(size_t)r * K + k] = v;
  }
  if (hi) {
    unsigned h, l;
    ffsr_split2(v, 0.f, h, l);
    hi[(size_t)r * ldw + k] = (unsigned short)(h & 0xffffu);
    lo[(size_t)r * ldw + k] = (unsigned short)(l & 0xffffu);
  }
}

// depthwise weights [C, 1, KH, KW] -> tap-major [T, C]; flip = 1: taps reversed (input-gradient operator)
__global__ void pack_dw_kernel(const float* __restrict__ src, int C, int T, int flip, float* __restrict__ dst) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= C * T) return;
  const int t = idx / C, c = idx % C;
  dst[idx] = src[(size_t)c * T + (flip ? T - 1 - t : t)];
}

// ---------------------------------------------------------------------------------------------- activations
// d act(v) / dv; ref = v (from_output 0) or act(v) (from_output 1: ReLU, LeakyReLU, sigmoid only)
__device__ __forceinline__ float act_grad(float r, int act, float slope, int from_output) {
  switch (act) {
    case FFSR_ACT_GELU: {   // 0.5 (1 + erf(v / sqrt 2)) + v exp(-v^2 / 2) / sqrt(2 pi)
      return 0.5f * (1.0f + erff(r * 0.70710678118654752440f)) + r * 0.39894228040143267794f * expf(-0.5f * r * r);
    }
    case FFSR_ACT_RELU: return r > 0.f ? 1.f : 0.f;
    case FFSR_ACT_LRELU: return r > 0.f ? 1.f : slope;
    case FFSR_ACT_SIGMOID: {
      const float s = from_output ? r : 1.0f / (1.0f + expf(-r));
      return s * (1.0f - s);
    }
    case FFSR_ACT_SILU: {
      const float s = 1.0f / (1.0f + expf(-r));
      return s * (1.0f + r * (1.0f - s));
    }
    case 6: return (r >= 0.f && r <= 1.f) ? 1.f : 0.f;   // clamp(v, 0, 1): torch passes the gradient on the closed interval
    case 7: return r >= slope ? 1.f : 0.f;               // clamp(v, min=slope)
    default: return 1.f;
  }
}
// dx = (accumulate ? dx : 0) + alpha * dy * act'(ref);  V = 4: C % 4 == 0 and 16-byte aligned rows
template <int V>
__global__ void act_bwd_kernel(const float* __restrict__ dy, int ldy, const float* __restrict__ ref, int ldr,
                               float* __restrict__ dx, int ldx, long long M, int C, int act, float slope, int from_output,
                               float alpha, int accumulate) {
  const int cv = C / V;
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * cv) return;
  const long long m = idx / cv;
  const int c = (int)(idx - m * cv) * V;
  if constexpr (V == 4) {
    const floatx4 g = *reinterpret_cast<const floatx4*>(dy + m * ldy + c);
    const floatx4 r = *reinterpret_cast<const floatx4*>(ref + m * ldr + c);
    floatx4 o = {0.f, 0.f, 0.f, 0.f};
    if (accumulate) o = *reinterpret_cast<const floatx4*>(dx + m * ldx + c);
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] += alpha * g[i] * act_grad(r[i], act, slope, from_output);
    *reinterpret_cast<floatx4*>(dx + m * ldx + c) = o;
  } else {
    const float g = alpha * dy[m * ldy + c] * act_grad(ref[m * ldr + c], act, slope, from_output);
    float* o = dx + m * ldx + c;
    *o = accumulate ? *o + g : g;
  }
}

// The same with the result written as bf16 hi / lo planes [M, ldp] (ldp == C, C % 32 == 0): dY of a wide layer goes straight to the
// planes GEMM (input gradient) and the planes weight-gradient kernel, no fp32 copy and no split pass.  8 channels per thread.
__global__ void act_bwd_planes_kernel(const float* __restrict__ dy, int ldy, const float* __restrict__ ref, int ldr,
                                      unsigned short* __restrict__ o_hi, unsigned short* __restrict__ o_lo, int ldp, long long M, int C,
                                      int act, float slope, int from_output, float alpha) {
  const int cv = C / 8;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * cv) return;
  const long long m = idx / cv;
  const int c = (int)(idx - m * cv) * 8;
  float o[8];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const floatx4 g = *reinterpret_cast<const floatx4*>(dy + m * ldy + c + 4 * h);
    const floatx4 r = *reinterpret_cast<const floatx4*>(ref + m * ldr + c + 4 * h);
#pragma unroll
    for (int i = 0; i < 4; ++i) o[4 * h + i] = alpha * g[i] * act_grad(r[i], act, slope, from_output);
  }
  ffsr_store_planes8(o_hi + m * ldp + c, o_lo + m * ldp + c, o);
}

// out = alpha * (sa ? sa[0] : 1) * a + beta * (sb ? sb[0] : 1) * b   (b optional; sa / sb = learnable device scalars)
template <int V>
__global__ void axpby_dev_kernel(const float* __restrict__ a, int lda, const float* __restrict__ sa, float alpha,
                                 const float* __restrict__ b, int ldb, const float* __restrict__ sb, float beta,
                                 float* __restrict__ out, int ldo, long long M, int C) {
  const int cv = C / V;
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * cv) return;
  const long long m = idx / cv;
  const int c = (int)(idx - m * cv) * V;
  const float fa = alpha * (sa ? sa[0] : 1.f), fb = beta * (sb ? sb[0] : 1.f);
  if constexpr (V == 4) {
    floatx4 y = *reinterpret_cast<const floatx4*>(a + m * lda + c) * fa;
    if (b) y += *reinterpret_cast<const floatx4*>(b + m * ldb + c) * fb;
    *reinterpret_cast<floatx4*>(out + m * ldo + c) = y;
  } else {
    float y = fa * a[m * lda + c];
    if (b) y += fb * b[m * ldb + c];
    out[m * ldo + c] = y;
  }
}

// ---------------------------------------------------------------------------------------------- reductions
// block-wide sum of a double (all threads get nothing back; thread 0 holds the result)
__device__ __forceinline__ double block_sum_d(double v, double* sh) {
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int o = RB / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  return sh[0];
}

// part[block] = sum over this block's share of a[m, c] * b[m, c]   (b NULL: sum of a)
__global__ __launch_bounds__(RB) void dot_partial_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b,
                                                         int ldb, long long M, int C, double* __restrict__ part) {
  __shared__ double sh[RB];
  const long long n = M * C;
  double s = 0.0;
  for (long long i = (long long)blockIdx.x * RB + threadIdx.x; i < n; i += (long long)gridDim.x * RB) {
    const long long m = i / C;
    const int c = (int)(i - m * C);
    const float x = a[m * lda + c];
    s += b ? (double)(x * b[m * ldb + c]) : (double)x;
  }
  const double t = block_sum_d(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}
// out[0] = (accumulate ? out[0] : 0) + scale * (gate ? gate[0] : 1) * sum part
__global__ __launch_bounds__(RB) void dot_finish_kernel(const double* __restrict__ part, int n, float* __restrict__ out,
                                                        float scale, int accumulate) {
  __shared__ double sh[RB];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += RB) s += part[i];
  const double t = block_sum_d(s, sh);
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + (float)(scale * t);
}

// column reductions: part[chunk, c] = sum over the rows of the chunk of a'[m, c] * (b ? b'[m, c] : 1) with
// b' = b - shift[c] * shift_scale (shift_mode >= 1) and a' = a - shift[c] * shift_scale (shift_mode == 2): CENTRED second
// moments (BatchNorm's variance and dgamma are sums around the batch mean; E[x^2] - mu^2 from fp32 sums cancels when
// |mean| >> std).  grid (ceil(C / 64), nchunk), block 256 = 64 channels x 4 row lanes
__global__ __launch_bounds__(RB) void coldot_partial_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b,
                                                            int ldb, long long M, int C, int nchunk, float* __restrict__ part,
                                                            const float* __restrict__ shift, float shift_scale, int shift_mode) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const long long per = (M + nchunk - 1) / nchunk;
  const long long r0 = blockIdx.y * per, r1 = (r0 + per < M) ? r0 + per : M;
  float s0 = 0.f, s1 = 0.f;
  if (c < C) {
    const float sb = shift_mode >= 1 ? shift[c] * shift_scale : 0.f;
    const float sa = shift_mode == 2 ? sb : 0.f;
    long long r = r0 + rl;
    for (; r + 4 < r1; r += 8) {
      s0 += (a[r * lda + c] - sa) * (b ? b[r * ldb + c] - sb : 1.f);
      s1 += (a[(r + 4) * lda + c] - sa) * (b ? b[(r + 4) * ldb + c] - sb : 1.f);
    }
    for (; r < r1; r += 4) s0 += (a[r * lda + c] - sa) * (b ? b[r * ldb + c] - sb : 1.f);
  }
  red[rl][threadIdx.x & 63] = s0 + s1;
  __syncthreads();
  if (rl == 0 && c < C)
    part[(size_t)blockIdx.y * C + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// out[c * ostride] = (accumulate ? out : 0) + scale * sum_k part[k * pstride + c]   (double accumulation)
// block 256 = 32 columns x 8 chunk lanes (four independent chains each: 32 loads of a column in flight), grid = ceil(C / 32)
__global__ __launch_bounds__(RB) void colsum_finish_kernel(const float* __restrict__ part, int nchunk, int pstride, int C,
                                                           float* __restrict__ out, int ostride, float scale, int accumulate) {
  __shared__ double red[8][32];
  const int l32 = threadIdx.x & 31, kl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + l32;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (c < C) {
    int k = kl;
    for (; k + 24 < nchunk; k += 32) {
      s0 += part[(size_t)k * pstride + c];
      s1 += part[(size_t)(k + 8) * pstride + c];
      s2 += part[(size_t)(k + 16) * pstride + c];
      s3 += part[(size_t)(k + 24) * pstride + c];
    }
    for (; k < nchunk; k += 8) s0 += part[(size_t)k * pstride + c];
  }
  red[kl][l32] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (kl == 0 && c < C) {
    double s = red[0][l32];
#pragma unroll
    for (int j = 1; j < 8; ++j) s += red[j][l32];
    float* o = out + (size_t)c * ostride;
    *o = (accumulate ? *o : 0.f) + (float)(scale * s);
  }
}

// out[m * ldo] = alpha * sum_c a[m, c] * b[m, c]     (gradient of a row-broadcast factor); LPR lanes per row so that a wave
// reads whole rows contiguously
template <int LPR>
__global__ __launch_bounds__(RB) void rowdot_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                                    float* __restrict__ out, int ldo, long long M, int C, float alpha, int accumulate) {
  const long long m = ((long long)blockIdx.x * RB + threadIdx.x) / LPR;
  const int l = threadIdx.x % LPR;
  float s = 0.f;
  if (m < M)
    for (int c = l; c < C; c += LPR) s = fmaf(a[m * lda + c], b[m * ldb + c], s);
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (m < M && l == 0) {
    float* o = out + m * ldo;
    *o = (accumulate ? *o : 0.f) + alpha * s;
  }
}

// ---------------------------------------------------------------------------------------------- BatchNorm2d (train mode)
// stats: sums [2, C] = (sum x, sum x^2) over M rows.  Writes mean / rstd [C] and the fused affine of the normalisation
// (scale = gamma rstd, shift = beta - mean scale) and applies nn.BatchNorm2d's running-statistics update
// (momentum 0.1, unbiased variance; large_kernel_attention.py:84,128,131 in model.train()).
__global__ void bn_stats_finish_kernel(const float* __restrict__ sums, long long M, int C, const float* __restrict__ gamma,
                                       const float* __restrict__ beta, float eps, float momentum, float* __restrict__ mean_rstd,
                                       float* __restrict__ scale_shift, float* __restrict__ run_mean, float* __restrict__ run_var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mu = (double)sums[c] / (double)M;
  // sums[C + c] = sum (x - m)^2 around m = fp32(sums[c] / M); the exact variance is that minus (mu - m)^2
  const double dm = mu - (double)(sums[c] * (1.f / (float)M));
  double var = (double)sums[C + c] / (double)M - dm * dm;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  mean_rstd[c] = (float)mu;
  mean_rstd[C + c] = rstd;
  const float sc = gamma[c] * rstd;
  scale_shift[c] = sc;
  scale_shift[C + c] = beta[c] - (float)mu * sc;
  if (run_mean) {
    const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mu;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unbiased;
  }
}
// sums [2, C] = (sum dy, sum dy * (x - mu)).  Coefficients of dx = A dy + Bx x + K (per channel) and the parameter gradients:
//   xhat = (x - mu) rstd,  dgamma = sum dy xhat = rstd S2,  dbeta = S1,
//   dx = gamma rstd (dy - S1 / M - xhat dgamma / M)
__global__ void bn_bwd_finish_kernel(const float* __restrict__ sums, long long M, int C, const float* __restrict__ gamma,
                                     const float* __restrict__ mean_rstd, float* __restrict__ coef, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double S1 = sums[c], S2 = sums[C + c], mu = mean_rstd[c], rstd = mean_rstd[C + c];
  const double dg = rstd * S2;
  const double g = (double)gamma[c] * rstd;
  const double k = rstd * dg / (double)M;          // dx = g dy - g k (x - mu) - g S1 / M
  coef[c] = (float)g;
  coef[C + c] = (float)(-g * k);
  coef[2 * C + c] = (float)(g * (k * mu - S1 / (double)M));
  dgamma[c] += (float)dg;
  dbeta[c] += (float)S1;
}
// out = A[c] p + Bx[c] q + K[c]
__global__ void affine2_kernel(const float* __restrict__ p, int ldp, const float* __restrict__ q, int ldq,
                               const float* __restrict__ coef, float* __restrict__ out, int ldo, long long M, int C) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * C) return;
  const long long m = idx / C;
  const int c = (int)(idx - m * C);
  out[m * ldo + c] = coef[c] * p[m * ldp + c] + coef[C + c] * q[m * ldq + c] + coef[2 * C + c];
}

// ---------------------------------------------------------------------------------------------- LayerNorm backward
// One wave per row (C <= 256: 4 channels per lane).  dx = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma;
// part[block, 0 / 1, c] = this block's sums of dy xhat / dy (finished by colsum_finish_kernel).
__global__ __launch_bounds__(RB) void layernorm_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                           const float* __restrict__ dy, int ldy, float* __restrict__ dx, int lddx,
                                                           float* __restrict__ part, long long M, int C, float eps,
                                                           long long rows_per_block) {
  __shared__ float acc[2][4][256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float dgam[4] = {0.f, 0.f, 0.f, 0.f}, dbet[4] = {0.f, 0.f, 0.f, 0.f};
  const long long r0 = blockIdx.x * rows_per_block, r1 = (r0 + rows_per_block < M) ? r0 + rows_per_block : M;
  float gm[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) gm[i] = (lane + 64 * i < C) ? gamma[lane + 64 * i] : 0.f;
  for (long long r = r0 + wv; r < r1; r += 4) {
    float xv[4], gv[4], s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      xv[i] = c < C ? x[r * ldx + c] : 0.f;
      gv[i] = c < C ? dy[r * ldy + c] : 0.f;
      s += xv[i];
    }
    const float mu = wave_sum(s) / (float)C;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float d = (lane + 64 * i < C) ? xv[i] - mu : 0.f;
      v += d * d;
    }
    const float rstd = rsqrtf(wave_sum(v) / (float)C + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xv[i] = (xv[i] - mu) * rstd;                 // xhat
      dgam[i] += gv[i] * xv[i];
      dbet[i] += gv[i];
      gv[i] *= gm[i];                              // g = dy gamma
      sg += gv[i];
      sgx += (lane + 64 * i < C) ? gv[i] * xv[i] : 0.f;
    }
    const float mg = wave_sum(sg) / (float)C, mgx = wave_sum(sgx) / (float)C;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      if (c < C) dx[r * lddx + c] = rstd * (gv[i] - mg - xv[i] * mgx);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    acc[0][wv][lane + 64 * i] = dgam[i];
    acc[1][wv][lane + 64 * i] = dbet[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += RB) {
    part[((size_t)blockIdx.x * 2 + 0) * C + c] = (acc[0][0][c] + acc[0][1][c]) + (acc[0][2][c] + acc[0][3][c]);
    part[((size_t)blockIdx.x * 2 + 1) * C + c] = (acc[1][0][c] + acc[1][1][c]) + (acc[1][2][c] + acc[1][3][c]);
  }
}

// ---------------------------------------------------------------------------------------------- depthwise conv weight grad
// part[chunk, t, c] = sum over the pixels of the chunk of dy[pix, c] * x[pix + offset(t), c]   (zero padding)
// grid (ceil(C / 64), nchunk), block 256 = 64 channels x 4 pixel lanes; KH x KW compile-time (5x5, 1x21, 21x1 of the LKA
// chain, large_kernel_attention.py:58-76); every tap has its own accumulator register.
template <int KH, int KW>
__global__ __launch_bounds__(RB) void dwconv_wgrad_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy, int ldy,
                                                          float* __restrict__ part, int B, int H, int W, int C, int ph, int pw,
                                                          int nchunk) {
  constexpr int T = KH * KW;
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int pl = threadIdx.x >> 6;
  const long long P = (long long)B * H * W, per = (P + nchunk - 1) / nchunk;
  const long long p0 = blockIdx.y * per, p1 = (p0 + per < P) ? p0 + per : P;
  float acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = 0.f;
  if (c < C && p0 + pl < p1) {
    long long p = p0 + pl;
    int xx = (int)(p % W), yy = (int)((p / W) % H);
    for (; p < p1; p += 4) {
      const float g = dy[p * ldy + c];
      const float* xc = x + p * ldx + c;
#pragma unroll
      for (int ky = 0; ky < KH; ++ky) {
        const int sy = yy + ky - ph;
        const bool yok = sy >= 0 && sy < H;
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
          const int sx = xx + kx - pw;
          const bool ok = yok && sx >= 0 && sx < W;
          const long long off = ((long long)(ky - ph) * W + (kx - pw)) * ldx;
          const float v = ok ? xc[off] : 0.f;
          acc[ky * KW + kx] = fmaf(g, v, acc[ky * KW + kx]);
        }
      }
      xx += 4;
      while (xx >= W) {
        xx -= W;
        if (++yy == H) yy = 0;
      }
    }
  }
#pragma unroll
  for (int t = 0; t < T; ++t) {
    red[pl][threadIdx.x & 63] = acc[t];
    __syncthreads();
    if (pl == 0 && c < C)
      part[((size_t)blockIdx.y * T + t) * C + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    __syncthreads();
  }
}
// dw[c, t] += sum_chunk part[chunk, t, c]      (dw in nn.Conv2d layout [C, 1, KH, KW]).  8 threads per output, four independent
// chains each, combined in a fixed order (a thread per output summed its ~1000 partials as one dependent chain: 60 us)
__global__ __launch_bounds__(256) void dwconv_wgrad_finish_kernel(const float* __restrict__ part, int nchunk, int T, int C,
                                                                  float* __restrict__ dw) {
  __shared__ double red[8][32];
  const int lane32 = threadIdx.x & 31, sg = threadIdx.x >> 5;
  const int idx = blockIdx.x * 32 + lane32;
  const bool live = idx < T * C;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (live) {
    const float* pp = part + idx;                      // part[(k * T + t) * C + c] = part[k * T * C + idx]
    const size_t st = (size_t)T * C;
    int k = sg;
    for (; k + 24 < nchunk; k += 32) {
      s0 += pp[(size_t)k * st];
      s1 += pp[(size_t)(k + 8) * st];
      s2 += pp[(size_t)(k + 16) * st];
      s3 += pp[(size_t)(k + 24) * st];
    }
    for (; k < nchunk; k += 8) s0 += pp[(size_t)k * st];
  }
  red[sg][lane32] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sg == 0 && live) {
    double a = red[0][lane32];
#pragma unroll
    for (int j = 1; j < 8; ++j) a += red[j][lane32];
    const int t = idx / C, c = idx % C;
    dw[(size_t)c * T + t] += (float)a;
  }
}

// ---------------------------------------------------------------------------------------------- resampler adjoints
__device__ __forceinline__ void bilin_coord(int o, float scale, int n, int& i0, int& i1, float& l) {
  float s = ((float)o + 0.5f) * scale - 0.5f;   // identical to the forward kernel (ffsr_pointwise.hip, ATen's area_pixel source index)
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > n - 1) i0 = n - 1;
  i1 = i0 + (i0 < n - 1 ? 1 : 0);
  l = s - (float)i0;
}
// weight with which output index o reads input index i (0 if it does not)
__device__ __forceinline__ float bilin_weight(int o, int i, float scale, int n) {
  int i0, i1;
  float l;
  bilin_coord(o, scale, n, i0, i1, l);
  return (i0 == i ? 1.f - l : 0.f) + (i1 == i ? l : 0.f);
}
// adjoint of F.interpolate(bilinear, align_corners=False): din[b, y, x, c] (+)= mul * sum_{oy, ox} wy wx dout[b, oy, ox, c]
// gather form (deterministic): the output rows / columns that can read input index i lie in [lo(i), hi(i)]
__global__ void bilinear_bwd_kernel(const float* __restrict__ dout, int ldo, float* __restrict__ din, int ldi, int B, int Hi, int Wi,
                                    int Ho, int Wo, int C, float sh, float sw, float mul, int accumulate) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)B * Hi * Wi * C) return;
  const int c = (int)(idx % C);
  long long pix = idx / C;
  const int x = (int)(pix % Wi);
  long long t = pix / Wi;
  const int y = (int)(t % Hi), b = (int)(t / Hi);
  // src(o) = (o + 0.5) s - 0.5 in (i - 1, i + 1)  <=>  o in ((i - 0.5) / s - 0.5, (i + 1.5) / s - 0.5); widened by one
  int oy0 = (int)floorf(((float)y - 0.5f) / sh - 0.5f) - 1, oy1 = (int)ceilf(((float)y + 1.5f) / sh - 0.5f) + 1;
  int ox0 = (int)floorf(((float)x - 0.5f) / sw - 0.5f) - 1, ox1 = (int)ceilf(((float)x + 1.5f) / sw - 0.5f) + 1;
  if (y == 0) oy0 = 0;                         // clamped sources (src < 0) all read index 0
  if (x == 0) ox0 = 0;
  if (y == Hi - 1) oy1 = Ho - 1;
  if (x == Wi - 1) ox1 = Wo - 1;
  oy0 = max(oy0, 0), ox0 = max(ox0, 0), oy1 = min(oy1, Ho - 1), ox1 = min(ox1, Wo - 1);
  float acc = 0.f;
  for (int oy = oy0; oy <= oy1; ++oy) {
    const float wy = bilin_weight(oy, y, sh, Hi);
    if (wy == 0.f) continue;
    float row = 0.f;
    for (int ox = ox0; ox <= ox1; ++ox) {
      const float wx = bilin_weight(ox, x, sw, Wi);
      if (wx != 0.f) row = fmaf(wx, dout[(((size_t)b * Ho + oy) * Wo + ox) * ldo + c], row);
    }
    acc = fmaf(wy, row, acc);
  }
  float* o = din + pix * ldi + c;
  *o = (accumulate ? *o : 0.f) + mul * acc;
}
// 4 channels per thread, 16-byte loads, the column weights of the window computed once per thread (the scalar kernel above
// re-derives them for every row of the window: 834 us for the 64x64 -> 256x256 x 128-channel adjoint, 3.7 ms per training step)
template <int MAXW>
__global__ void bilinear_bwd4_kernel(const float* __restrict__ dout, int ldo, float* __restrict__ din, int ldi, int B, int Hi, int Wi,
                                     int Ho, int Wo, int C, float sh, float sw, float mul, int accumulate) {
  const int c4n = C / 4;
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)B * Hi * Wi * c4n) return;
  const int c = (int)(idx % c4n) * 4;
  long long pix = idx / c4n;
  const int x = (int)(pix % Wi);
  long long t = pix / Wi;
  const int y = (int)(t % Hi), b = (int)(t / Hi);
  int oy0 = (int)floorf(((float)y - 0.5f) / sh - 0.5f) - 1, oy1 = (int)ceilf(((float)y + 1.5f) / sh - 0.5f) + 1;
  int ox0 = (int)floorf(((float)x - 0.5f) / sw - 0.5f) - 1, ox1 = (int)ceilf(((float)x + 1.5f) / sw - 0.5f) + 1;
  if (y == 0) oy0 = 0;
  if (x == 0) ox0 = 0;
  if (y == Hi - 1) oy1 = Ho - 1;
  if (x == Wi - 1) ox1 = Wo - 1;
  oy0 = max(oy0, 0), ox0 = max(ox0, 0), oy1 = min(oy1, Ho - 1), ox1 = min(ox1, Wo - 1);
  float wxs[MAXW];
#pragma unroll
  for (int j = 0; j < MAXW; ++j) wxs[j] = (ox0 + j <= ox1) ? bilin_weight(ox0 + j, x, sw, Wi) : 0.f;
  floatx4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int oy = oy0; oy <= oy1; ++oy) {
    const float wy = bilin_weight(oy, y, sh, Hi);
    if (wy == 0.f) continue;
    const float* rowp = dout + (((size_t)b * Ho + oy) * Wo + ox0) * ldo + c;
    floatx4 row = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < MAXW; ++j)
      if (ox0 + j <= ox1 && wxs[j] != 0.f) row += *reinterpret_cast<const floatx4*>(rowp + (size_t)j * ldo) * wxs[j];
    acc += row * wy;
  }
  float* o = din + pix * ldi + c;
  floatx4 r = acc * mul;
  if (accumulate) r += *reinterpret_cast<const floatx4*>(o);
  *reinterpret_cast<floatx4*>(o) = r;
}
// adjoint of F.avg_pool2d(x, 2, 2): din[b, y, x, c] = 0.25 dout[b, y / 2, x / 2, c] (rows / columns beyond 2 * (H / 2) get 0)
__global__ void avgpool2_bwd_kernel(const float* __restrict__ dout, int ldo, float* __restrict__ din, int ldi, int B, int H, int W,
                                    int C, int accumulate) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)B * H * W * C) return;
  const int c = (int)(idx % C);
  long long pix = idx / C;
  const int x = (int)(pix % W);
  long long t = pix / W;
  const int y = (int)(t % H), b = (int)(t / H);
  const int Ho = H / 2, Wo = W / 2;
  float g = 0.f;
  if (y / 2 < Ho && x / 2 < Wo) g = 0.25f * dout[(((size_t)b * Ho + y / 2) * Wo + x / 2) * ldo + c];
  float* o = din + pix * ldi + c;
  *o = (accumulate ? *o : 0.f) + g;
}

// ---------------------------------------------------------------------------------------------- per-pixel MHA backward
// qkv [S*T, 3E] (q | k | v, heads of 16), dout [S*T, E].  Kernel A: thread = (sequence, head, query t): recomputes the
// softmax row P[t, :], writes dq[t] and the rows P[t, :], dS[t, :] (dS = P (dP - sum P dP), dP = dO V^T) to scratch;
// kernel B: thread = (sequence, head, key j): dk[j] = scale sum_t dS[t, j] q[t],  dv[j] = sum_t P[t, j] dO[t].
// Attention dropout (drop_thr > 0): the forward pass used P' = P * m, m[t, j] = keep ? 1 / (1 - p) : 0 regenerated here from the
// same (seed, element index); then dP = (dO V^T) * m, dS = P (dP - sum P dP) and dv uses P'.
template <int T>
__global__ __launch_bounds__(RB, 2) void pixel_mha_bwd_a_kernel(   // (without the bound: 128 registers, 220 spilled at T = 9)
    const float* __restrict__ qkv, int ldq, const float* __restrict__ dout, int ldo,
                                       float* __restrict__ dqkv, int lddq, float* __restrict__ scratch, long long S, int E, int heads,
                                       unsigned drop_thr, float keep_scale, unsigned long long seed) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= S * heads * T) return;
  const int t = (int)(idx % T);
  long long r = idx / T;
  const int h = (int)(r % heads);
  const long long s = r / heads;
  const float scale = 0.25f;      // 1 / sqrt(16)
  const float* base = qkv + (s * T) * ldq + h * 16;
  float q[16], dO[16];
#pragma unroll
  for (int d = 0; d < 16; ++d) q[d] = base[(size_t)t * ldq + d], dO[d] = dout[(s * T + t) * ldo + h * 16 + d];
  float p[T], dp[T], mx = -3.0e38f;
#pragma unroll
  for (int j = 0; j < T; ++j) {
    const float* kj = base + (size_t)j * ldq + E;
    const float* vj = base + (size_t)j * ldq + 2 * E;
    float a = 0.f, g = 0.f;
#pragma unroll
    for (int d = 0; d < 16; ++d) a = fmaf(q[d] * scale, kj[d], a), g = fmaf(dO[d], vj[d], g);
    p[j] = a, dp[j] = g;
    mx = fmaxf(mx, a);
  }
  float den = 0.f;
#pragma unroll
  for (int j = 0; j < T; ++j) p[j] = expf(p[j] - mx), den += p[j];
  float mk[T];
#pragma unroll
  for (int j = 0; j < T; ++j) {
    mk[j] = 1.f;
    if (drop_thr) {
      const unsigned long long e = (((unsigned long long)s * heads + h) * T + t) * T + j;
      mk[j] = ffsr_rng_u32(seed, e) >= drop_thr ? keep_scale : 0.f;
    }
    dp[j] *= mk[j];
  }
  float dot = 0.f;
#pragma unroll
  for (int j = 0; j < T; ++j) p[j] /= den, dot = fmaf(p[j], dp[j], dot);
  float dq[16];
#pragma unroll
  for (int d = 0; d < 16; ++d) dq[d] = 0.f;
  float* sc = scratch + idx * (2 * T);
#pragma unroll
  for (int j = 0; j < T; ++j) {
    const float ds = p[j] * (dp[j] - dot);
    sc[j] = p[j] * mk[j], sc[T + j] = ds;
    const float* kj = base + (size_t)j * ldq + E;
#pragma unroll
    for (int d = 0; d < 16; ++d) dq[d] = fmaf(ds * scale, kj[d], dq[d]);
  }
  float* o = dqkv + (s * T + t) * lddq + h * 16;
#pragma unroll
  for (int d = 0; d < 16; ++d) o[d] = dq[d];
}
template <int T>
__global__ void pixel_mha_bwd_b_kernel(const float* __restrict__ qkv, int ldq, const float* __restrict__ dout, int ldo,
                                       float* __restrict__ dqkv, int lddq, const float* __restrict__ scratch, long long S, int E,
                                       int heads) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= S * heads * T) return;
  const int j = (int)(idx % T);
  long long r = idx / T;
  const int h = (int)(r % heads);
  const long long s = r / heads;
  const float scale = 0.25f;
  float dk[16], dv[16];
#pragma unroll
  for (int d = 0; d < 16; ++d) dk[d] = 0.f, dv[d] = 0.f;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const float* sc = scratch + ((s * heads + h) * T + t) * (2 * T);
    const float pj = sc[j], dsj = sc[T + j] * scale;
    const float* qt = qkv + (s * T + t) * ldq + h * 16;
    const float* dOt = dout + (s * T + t) * ldo + h * 16;
#pragma unroll
    for (int d = 0; d < 16; ++d) dk[d] = fmaf(dsj, qt[d], dk[d]), dv[d] = fmaf(pj, dOt[d], dv[d]);
  }
  float* o = dqkv + (s * T + j) * lddq + h * 16;
#pragma unroll
  for (int d = 0; d < 16; ++d) o[E + d] = dk[d], o[2 * E + d] = dv[d];
}

// ---------------------------------------------------------------------------------------------- fusion tails
// softmax over the C <= 8 channels of a row, and its backward dx = y (dy - sum y dy)
__global__ void softmax_c_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, long long M, int C) {
  const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  float v[8], mx = -3.0e38f, den = 0.f;
  for (int c = 0; c < C; ++c) v[c] = x[m * ldx + c], mx = fmaxf(mx, v[c]);
  for (int c = 0; c < C; ++c) v[c] = expf(v[c] - mx), den += v[c];
  for (int c = 0; c < C; ++c) y[m * ldy + c] = v[c] / den;
}
__global__ void softmax_c_bwd_kernel(const float* __restrict__ y, int ldy, const float* __restrict__ dy, int lddy,
                                     float* __restrict__ dx, int ldx, long long M, int C) {
  const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  float dot = 0.f;
  for (int c = 0; c < C; ++c) dot = fmaf(y[m * ldy + c], dy[m * lddy + c], dot);
  for (int c = 0; c < C; ++c) dx[m * ldx + c] = y[m * ldy + c] * (dy[m * lddy + c] - dot);
}
// expert-weighted sum (enhanced_fusion_v2.py:744-747, 761-768): x [M, 12] = 4 experts x 3 channels, g [M, 4]
//   out[c] = sum_e x[3 e + c] g[e] / (normalize ? sum_e g[e] + 1e-8 : 1)
__global__ void expert_sum_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ g, int ldg, float* __restrict__ out,
                                  int ldo, long long M, int normalize) {
  const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  float gv[4], gs = 0.f;
  for (int e = 0; e < 4; ++e) gv[e] = g[m * ldg + e], gs += gv[e];
  const float inv = normalize ? 1.f / (gs + 1e-8f) : 1.f;
  for (int c = 0; c < 3; ++c) {
    float s = 0.f;
    for (int e = 0; e < 4; ++e) s = fmaf(x[m * ldx + 3 * e + c], gv[e], s);
    out[m * ldo + c] = s * inv;
  }
}
// dx[3 e + c] (+)= dy[c] g[e] inv;   dg[e] = inv (sum_c dy[c] x[3 e + c] - (normalize ? sum_c dy[c] out[c] : 0))
__global__ void expert_sum_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ g, int ldg,
                                      const float* __restrict__ dy, int lddy, float* __restrict__ dx, int lddx,
                                      float* __restrict__ dg, int lddg, long long M, int normalize, int accumulate_dx) {
  const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  float gv[4], gs = 0.f, d[3];
  for (int e = 0; e < 4; ++e) gv[e] = g[m * ldg + e], gs += gv[e];
  const float inv = normalize ? 1.f / (gs + 1e-8f) : 1.f;
  for (int c = 0; c < 3; ++c) d[c] = dy[m * lddy + c];
  float dyo = 0.f;     // sum_c dy[c] out[c]
  float xe[4];
  for (int e = 0; e < 4; ++e) {
    xe[e] = 0.f;
    for (int c = 0; c < 3; ++c) {
      const float xv = x[m * ldx + 3 * e + c];
      xe[e] = fmaf(d[c], xv, xe[e]);
      float* o = dx + m * lddx + 3 * e + c;
      const float gx = d[c] * gv[e] * inv;
      *o = accumulate_dx ? *o + gx : gx;
    }
    dyo = fmaf(xe[e], gv[e], dyo);
  }
  dyo *= inv;
  for (int e = 0; e < 4; ++e) dg[m * lddg + e] = inv * (xe[e] - (normalize ? dyo : 0.f));
}
// DynamicExpertSelector tail (enhanced_fusion_v2.py:462-465): thr = 0.7 - 0.5 d, s_e = sigmoid(T (raw_e - thr)),
// den = clamp(sum_e s_e + 1e-8, min 0.3), gates_e = s_e / den.  Backward: given dgates -> draw, dd, part[block] = dT share.
__global__ __launch_bounds__(RB) void selector_gates_bwd_kernel(const float* __restrict__ raw, int ldr, const float* __restrict__ diff,
                                                                int ldd, const float* __restrict__ temperature,
                                                                const float* __restrict__ dgates, int ldg, float* __restrict__ draw,
                                                                int lddr, float* __restrict__ ddiff, int lddd,
                                                                double* __restrict__ part, long long M) {
  __shared__ double sh[RB];
  const float T = temperature[0];
  double dT = 0.0;
  for (long long m = (long long)blockIdx.x * RB + threadIdx.x; m < M; m += (long long)gridDim.x * RB) {
    const float thr = 0.7f - 0.5f * diff[m * ldd];
    float s[4], z[4], sum = 0.f;
    for (int e = 0; e < 4; ++e) {
      z[e] = raw[m * ldr + e] - thr;
      s[e] = 1.0f / (1.0f + expf(-T * z[e]));
      sum += s[e];
    }
    const float pre = sum + 1e-8f;
    const float den = fmaxf(pre, 0.3f);
    float dden = 0.f, ds[4];
    for (int e = 0; e < 4; ++e) {
      const float dgv = dgates[m * ldg + e];
      ds[e] = dgv / den;
      dden -= dgv * s[e] / (den * den);
    }
    if (pre >= 0.3f)                                   // clamp(min): gradient passes where the input is >= min
      for (int e = 0; e < 4; ++e) ds[e] += dden;
    float dthr = 0.f;
    for (int e = 0; e < 4; ++e) {
      const float du = ds[e] * s[e] * (1.f - s[e]);    // through the sigmoid, u = T z
      draw[m * lddr + e] = du * T;
      dthr -= du * T;
      dT += (double)(du * z[e]);
    }
    ddiff[m * lddd] = -0.5f * dthr;
  }
  const double t = block_sum_d(dT, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// ---------------------------------------------------------------------------------------------- FFT mask gradient
// lo = irfft2(X m), hi = irfft2(X (1 - m)) (multi_domain_frequency.py:376-383) are linear in the mask:
//   dL/dm[ky, k] = w_k sum_{b, c} Re(X conj(Ghat)),  Ghat = rfft2(G, ortho), G = dL/dlo - dL/dhi, w_k = 1 on the DC / Nyquist
// columns and 2 elsewhere (the c2r transform reads those columns once, the others stand for a conjugate pair).
// Xlo / Xhi: the masked spectra X m and X (1 - m) the forward pass left in its work buffer (X = Xlo + Xhi).
__global__ void fft_mask_grad_kernel(const float2* __restrict__ Xlo, const float2* __restrict__ Xhi, const float2* __restrict__ Gh,
                                     float* __restrict__ dmask, int BC, int H, int W, int Wf) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= H * Wf) return;
  const int k = idx % Wf;
  const float wk = (k == 0 || (!(W & 1) && k == Wf - 1)) ? 1.f : 2.f;
  float s = 0.f;
  for (int bc = 0; bc < BC; ++bc) {
    const size_t o = (size_t)bc * H * Wf + idx;
    const float2 a = Xlo[o], b = Xhi[o], g = Gh[o];
    s += (a.x + b.x) * g.x + (a.y + b.y) * g.y;
  }
  dmask[idx] = wk * s;
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int ffsr_zero_f32(float* p, long long n, void* stream) {
  FFSR_CHECK(p && n >= 0);
  if (n == 0) return FFSR_OK;
  return hipMemsetAsync(p, 0, (size_t)n * sizeof(float), ST) == hipSuccess ? FFSR_OK : FFSR_ELAUNCH;
}

extern "C" int ffsr_pack_conv_f32(const float* w, int N, int Cin, int KH, int KW, int transpose, float* dst_f32, int Cp,
                                  void* dst_hi, void* dst_lo, int rows_pad, int ldw, void* stream) {
  const int rows = transpose ? Cin : N, ch = transpose ? N : Cin, T = KH * KW;
  FFSR_CHECK(w && N > 0 && Cin > 0 && T > 0 && Cp >= ch && (dst_f32 || dst_hi) && (!dst_hi == !dst_lo));
  if (dst_hi) FFSR_CHECK(rows_pad >= rows && ldw >= T * Cp);
  const long long n = dst_hi ? (long long)rows_pad * ldw : (long long)rows * T * Cp;
  FFSR_LAUNCH(pack_conv_kernel, dim3(grid_for(n)), dim3(RB), 0, ST, w, N, Cin, T, transpose, dst_f32, Cp,
              (unsigned short*)dst_hi, (unsigned short*)dst_lo, rows_pad, ldw);
  return ffsr_launch_status();
}

extern "C" int ffsr_pack_dwconv_f32(const float* w, int C, int KH, int KW, int flip, float* dst, void* stream) {
  FFSR_CHECK(w && dst && C > 0 && KH * KW > 0);
  FFSR_LAUNCH(pack_dw_kernel, dim3(grid_for((long long)C * KH * KW)), dim3(RB), 0, ST, w, C, KH * KW, flip, dst);
  return ffsr_launch_status();
}

extern "C" int ffsr_act_bwd_f32(const float* dy, int ldy, const float* ref, int ldr, float* dx, int ldx, long long M, int C,
                                int act, float slope, int from_output, float alpha, int accumulate, void* stream) {
  FFSR_CHECK(dy && ref && dx && M > 0 && C > 0 && ldy >= C && ldr >= C && ldx >= C && act >= 0 && act <= 7);
  FFSR_CHECK(!from_output || act == FFSR_ACT_RELU || act == FFSR_ACT_LRELU || act == FFSR_ACT_SIGMOID || act == FFSR_ACT_NONE);
  auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  if (C % 4 == 0 && ldy % 4 == 0 && ldr % 4 == 0 && ldx % 4 == 0 && al16(dy) && al16(ref) && al16(dx))
    FFSR_LAUNCH(act_bwd_kernel<4>, dim3(grid_for(M * (C / 4))), dim3(RB), 0, ST, dy, ldy, ref, ldr, dx, ldx, M, C, act, slope,
                from_output, alpha, accumulate);
  else
    FFSR_LAUNCH(act_bwd_kernel<1>, dim3(grid_for(M * C)), dim3(RB), 0, ST, dy, ldy, ref, ldr, dx, ldx, M, C, act, slope,
                from_output, alpha, accumulate);
  return ffsr_launch_status();
}

extern "C" int ffsr_act_bwd_planes_f32(const float* dy, int ldy, const float* ref, int ldr, void* out_hi, void* out_lo, int ldp,
                                       long long M, int C, int act, float slope, int from_output, float alpha, void* stream) {
  FFSR_CHECK(dy && ref && out_hi && out_lo && M > 0 && C > 0 && (C & 31) == 0 && ldp == C && ldy >= C && ldr >= C && act >= 0 && act <= 7);
  FFSR_CHECK(!from_output || act == FFSR_ACT_RELU || act == FFSR_ACT_LRELU || act == FFSR_ACT_SIGMOID || act == FFSR_ACT_NONE);
  FFSR_CHECK(ldy % 4 == 0 && ldr % 4 == 0 && (((uintptr_t)dy | (uintptr_t)ref | (uintptr_t)out_hi | (uintptr_t)out_lo) & 15) == 0);
  FFSR_LAUNCH(act_bwd_planes_kernel, dim3(grid_for(M * (C / 8))), dim3(RB), 0, ST, dy, ldy, ref, ldr, (unsigned short*)out_hi,
              (unsigned short*)out_lo, ldp, M, C, act, slope, from_output, alpha);
  return ffsr_launch_status();
}

extern "C" int ffsr_axpby_dev_f32(const float* a, int lda, const float* sa, float alpha, const float* b, int ldb,
                                  const float* sb, float beta, float* out, int ldo, long long M, int C, void* stream) {
  FFSR_CHECK(a && out && M > 0 && C > 0 && lda >= C && ldo >= C && (!b || ldb >= C));
  auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  if (C % 4 == 0 && lda % 4 == 0 && ldo % 4 == 0 && (!b || ldb % 4 == 0) && al16(a) && al16(out) && (!b || al16(b)))
    FFSR_LAUNCH(axpby_dev_kernel<4>, dim3(grid_for(M * (C / 4))), dim3(RB), 0, ST, a, lda, sa, alpha, b, ldb, sb, beta, out, ldo, M, C);
  else
    FFSR_LAUNCH(axpby_dev_kernel<1>, dim3(grid_for(M * C)), dim3(RB), 0, ST, a, lda, sa, alpha, b, ldb, sb, beta, out, ldo, M, C);
  return ffsr_launch_status();
}

extern "C" int ffsr_dot_acc_f32(const float* a, int lda, const float* b, int ldb, long long M, int C, double* partial,
                                int n_partial, float* out, float scale, int accumulate, void* stream) {
  FFSR_CHECK(a && partial && out && M > 0 && C > 0 && lda >= C && (!b || ldb >= C) && n_partial >= 1);
  long long want = (M * C + RB * 8 - 1) / (RB * 8);
  const int nb = (int)(want < 1 ? 1 : (want < n_partial ? want : n_partial));
  FFSR_LAUNCH(dot_partial_kernel, dim3(nb), dim3(RB), 0, ST, a, lda, b, ldb, M, C, partial);
  FFSR_LAUNCH(dot_finish_kernel, dim3(1), dim3(RB), 0, ST, partial, nb, out, scale, accumulate);
  return ffsr_launch_status();
}

// out[c * ostride] (+)= scale * sum_m a[m, c] * (b ? b[m, c] : 1);  part: scratch of nchunk * C floats
extern "C" int ffsr_coldot_acc_f32(const float* a, int lda, const float* b, int ldb, long long M, int C, float* part,
                                   int nchunk, float* out, int ostride, float scale, int accumulate, void* stream) {
  FFSR_CHECK(a && part && out && M > 0 && C > 0 && lda >= C && (!b || ldb >= C) && nchunk >= 1 && nchunk <= 65535 && ostride >= 1);
  FFSR_LAUNCH(coldot_partial_kernel, dim3((C + 63) / 64, nchunk), dim3(RB), 0, ST, a, lda, b, ldb, M, C, nchunk, part,
              (const float*)nullptr, 0.f, 0);
  FFSR_LAUNCH(colsum_finish_kernel, dim3((C + 31) / 32), dim3(RB), 0, ST, part, nchunk, C, C, out, ostride, scale, accumulate);
  return ffsr_launch_status();
}

extern "C" int ffsr_rowdot_f32(const float* a, int lda, const float* b, int ldb, float* out, int ldo, long long M, int C,
                               float alpha, int accumulate, void* stream) {
  FFSR_CHECK(a && b && out && M > 0 && C > 0 && lda >= C && ldb >= C && ldo >= 1);
  if (C >= 48)
    FFSR_LAUNCH(rowdot_kernel<32>, dim3(grid_for(M * 32)), dim3(RB), 0, ST, a, lda, b, ldb, out, ldo, M, C, alpha, accumulate);
  else if (C >= 12)
    FFSR_LAUNCH(rowdot_kernel<8>, dim3(grid_for(M * 8)), dim3(RB), 0, ST, a, lda, b, ldb, out, ldo, M, C, alpha, accumulate);
  else
    FFSR_LAUNCH(rowdot_kernel<1>, dim3(grid_for(M)), dim3(RB), 0, ST, a, lda, b, ldb, out, ldo, M, C, alpha, accumulate);
  return ffsr_launch_status();
}

// nn.BatchNorm2d in train mode, statistics part.  stat [2, C] <- (mean, rstd); scale_shift [2, C] <- fused affine
// y = x * scale + shift (apply with ffsr_unary_f32); run_mean / run_var (may be NULL) updated in place (momentum 0.1).
// part: scratch 2 * nchunk * C floats; sums: scratch 2 * C floats.
extern "C" int ffsr_bn_train_stats_f32(const float* x, int ldx, long long M, int C, const float* gamma, const float* beta,
                                       float eps, float momentum, float* part, int nchunk, float* sums, float* stat,
                                       float* scale_shift, float* run_mean, float* run_var, void* stream) {
  FFSR_CHECK(x && gamma && beta && part && sums && stat && scale_shift && M > 0 && C > 0 && ldx >= C && nchunk >= 1 && nchunk <= 65535);
  FFSR_CHECK(!run_mean == !run_var);
  const dim3 g((C + 63) / 64, nchunk);
  FFSR_LAUNCH(coldot_partial_kernel, g, dim3(RB), 0, ST, x, ldx, (const float*)nullptr, 0, M, C, nchunk, part,
              (const float*)nullptr, 0.f, 0);
  FFSR_LAUNCH(colsum_finish_kernel, dim3((C + 31) / 32), dim3(RB), 0, ST, part, nchunk, C, C, sums, 1, 1.f, 0);
  // second pass around the batch mean: sums[C + c] = sum (x - mean)^2 (two-pass variance, as accurate as torch's Welford)
  FFSR_LAUNCH(coldot_partial_kernel, g, dim3(RB), 0, ST, x, ldx, x, ldx, M, C, nchunk, part, sums, 1.f / (float)M, 2);
  FFSR_LAUNCH(colsum_finish_kernel, dim3((C + 31) / 32), dim3(RB), 0, ST, part, nchunk, C, C, sums + C, 1, 1.f, 0);
  FFSR_LAUNCH(bn_stats_finish_kernel, dim3((C + 63) / 64), dim3(64), 0, ST, sums, M, C, gamma, beta, eps, momentum, stat,
              scale_shift, run_mean, run_var);
  return ffsr_launch_status();
}

// BatchNorm2d backward: dx = dL/dx (may alias dy), dgamma / dbeta accumulated.  stat = (mean, rstd) of the forward pass.
// part: 2 * nchunk * C floats; sums: 2 * C; coef: 3 * C floats of scratch.
extern "C" int ffsr_bn_train_bwd_f32(const float* x, int ldx, const float* dy, int ldy, float* dx, int lddx, long long M, int C,
                                     const float* gamma, const float* stat, float* part, int nchunk, float* sums, float* coef,
                                     float* dgamma, float* dbeta, void* stream) {
  FFSR_CHECK(x && dy && dx && gamma && stat && part && sums && coef && dgamma && dbeta && M > 0 && C > 0 && ldx >= C &&
             ldy >= C && lddx >= C && nchunk >= 1 && nchunk <= 65535);
  const dim3 g((C + 63) / 64, nchunk);
  FFSR_LAUNCH(coldot_partial_kernel, g, dim3(RB), 0, ST, dy, ldy, (const float*)nullptr, 0, M, C, nchunk, part,
              (const float*)nullptr, 0.f, 0);
  FFSR_LAUNCH(colsum_finish_kernel, dim3((C + 31) / 32), dim3(RB), 0, ST, part, nchunk, C, C, sums, 1, 1.f, 0);
  // sums[C + c] = sum dy (x - mean): dgamma's sum taken around the batch mean (no S2 - mu S1 cancellation)
  FFSR_LAUNCH(coldot_partial_kernel, g, dim3(RB), 0, ST, dy, ldy, x, ldx, M, C, nchunk, part, stat, 1.f, 1);
  FFSR_LAUNCH(colsum_finish_kernel, dim3((C + 31) / 32), dim3(RB), 0, ST, part, nchunk, C, C, sums + C, 1, 1.f, 0);
  FFSR_LAUNCH(bn_bwd_finish_kernel, dim3((C + 63) / 64), dim3(64), 0, ST, sums, M, C, gamma, stat, coef, dgamma, dbeta);
  FFSR_LAUNCH(affine2_kernel, dim3(grid_for(M * C)), dim3(RB), 0, ST, dy, ldy, x, ldx, coef, dx, lddx, M, C);
  return ffsr_launch_status();
}

// nn.LayerNorm backward over rows of C <= 256 channels.  dx (may not alias x), dgamma / dbeta accumulated.
// part: scratch of 2 * nblock * C floats.
extern "C" int ffsr_layernorm_bwd_f32(const float* x, int ldx, const float* gamma, float eps, const float* dy, int ldy, float* dx,
                                      int lddx, float* part, int nblock, float* dgamma, float* dbeta, long long M, int C,
                                      void* stream) {
  FFSR_CHECK(x && gamma && dy && dx && part && dgamma && dbeta && M > 0 && C > 0 && C <= 256 && ldx >= C && ldy >= C &&
             lddx >= C && nblock >= 1);
  long long rpb = (M + nblock - 1) / nblock;
  const int nb = (int)((M + rpb - 1) / rpb);
  FFSR_LAUNCH(layernorm_bwd_kernel, dim3(nb), dim3(RB), 0, ST, x, ldx, gamma, dy, ldy, dx, lddx, part, M, C, eps, rpb);
  FFSR_LAUNCH(colsum_finish_kernel, dim3((C + 31) / 32), dim3(RB), 0, ST, part, nb, 2 * C, C, dgamma, 1, 1.f, 1);
  FFSR_LAUNCH(colsum_finish_kernel, dim3((C + 31) / 32), dim3(RB), 0, ST, part + C, nb, 2 * C, C, dbeta, 1, 1.f, 1);
  return ffsr_launch_status();
}

// depthwise convolution (stride 1, zero padding): dw [C, 1, KH, KW] += sum_pix dy[pix, c] x[pix + tap, c]; KH * KW <= 25.
// part: scratch of nchunk * KH * KW * C floats.
extern "C" int ffsr_dwconv_wgrad_f32(const float* x, int ldx, const float* dy, int ldy, float* dw, float* part, int nchunk, int B,
                                     int H, int W, int C, int KH, int KW, int pad_h, int pad_w, void* stream) {
  FFSR_CHECK(x && dy && dw && part && B > 0 && H > 0 && W > 0 && C > 0 && ldx >= C && ldy >= C && nchunk >= 1 && nchunk <= 65535);
  const dim3 g((C + 63) / 64, nchunk);
  if (KH == 5 && KW == 5)
    FFSR_LAUNCH((dwconv_wgrad_kernel<5, 5>), g, dim3(RB), 0, ST, x, ldx, dy, ldy, part, B, H, W, C, pad_h, pad_w, nchunk);
  else if (KH == 1 && KW == 21)
    FFSR_LAUNCH((dwconv_wgrad_kernel<1, 21>), g, dim3(RB), 0, ST, x, ldx, dy, ldy, part, B, H, W, C, pad_h, pad_w, nchunk);
  else if (KH == 21 && KW == 1)
    FFSR_LAUNCH((dwconv_wgrad_kernel<21, 1>), g, dim3(RB), 0, ST, x, ldx, dy, ldy, part, B, H, W, C, pad_h, pad_w, nchunk);
  else if (KH == 3 && KW == 3)
    FFSR_LAUNCH((dwconv_wgrad_kernel<3, 3>), g, dim3(RB), 0, ST, x, ldx, dy, ldy, part, B, H, W, C, pad_h, pad_w, nchunk);
  else
    return FFSR_EINVAL;
  FFSR_LAUNCH(dwconv_wgrad_finish_kernel, dim3((unsigned)((KH * KW * C + 31) / 32)), dim3(256), 0, ST, part, nchunk, KH * KW, C, dw);
  return ffsr_launch_status();
}

extern "C" int ffsr_bilinear_bwd_f32(const float* dout, int ldo, float* din, int ldi, int B, int Hi, int Wi, int Ho, int Wo, int C,
                                     float mul, int accumulate, void* stream) {
  FFSR_CHECK(dout && din && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0 && ldo >= C && ldi >= C);
  const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
  // window width of an input column: 2 / sw + 3 output columns at most
  const bool vec = C % 4 == 0 && ldo % 4 == 0 && ldi % 4 == 0 && (((uintptr_t)dout | (uintptr_t)din) & 15) == 0 && 2.f / sw + 3.f <= 12.f;
  if (vec)
    FFSR_LAUNCH(bilinear_bwd4_kernel<12>, dim3(grid_for((long long)B * Hi * Wi * (C / 4))), dim3(RB), 0, ST, dout, ldo, din, ldi, B, Hi,
                Wi, Ho, Wo, C, sh, sw, mul, accumulate);
  else
    FFSR_LAUNCH(bilinear_bwd_kernel, dim3(grid_for((long long)B * Hi * Wi * C)), dim3(RB), 0, ST, dout, ldo, din, ldi, B, Hi, Wi,
                Ho, Wo, C, sh, sw, mul, accumulate);
  return ffsr_launch_status();
}

extern "C" int ffsr_avgpool2_bwd_f32(const float* dout, int ldo, float* din, int ldi, int B, int H, int W, int C, int accumulate,
                                     void* stream) {
  FFSR_CHECK(dout && din && B > 0 && H >= 2 && W >= 2 && C > 0 && ldo >= C && ldi >= C);
  FFSR_LAUNCH(avgpool2_bwd_kernel, dim3(grid_for((long long)B * H * W * C)), dim3(RB), 0, ST, dout, ldo, din, ldi, B, H, W, C,
              accumulate);
  return ffsr_launch_status();
}

// backward of ffsr_pixel_mha_f32 (eval-mode attention core, dropout 0): dqkv [S*T, 3E]; scratch: 2 * S * heads * T * T floats
extern "C" int ffsr_pixel_mha_bwd_f32(const float* qkv, int ldq, const float* dout, int ldo, float* dqkv, int lddq, float* scratch,
                                      float p_drop, long long seed,
                                      long long S, int T, int E, int heads, void* stream) {
  FFSR_CHECK(qkv && dout && dqkv && scratch && S > 0 && (T == 9 || T == 4) && heads * 16 == E && ldq >= 3 * E && ldo >= E &&
             lddq >= 3 * E);
  FFSR_CHECK(p_drop >= 0.f && p_drop < 1.f);
  const unsigned thr = ffsr_drop_threshold(p_drop);
  const float ks = 1.0f / (1.0f - p_drop);
  const int g = grid_for(S * heads * T);
  if (T == 9) {
    FFSR_LAUNCH(pixel_mha_bwd_a_kernel<9>, dim3(g), dim3(RB), 0, ST, qkv, ldq, dout, ldo, dqkv, lddq, scratch, S, E, heads, thr, ks, (unsigned long long)seed);
    FFSR_LAUNCH(pixel_mha_bwd_b_kernel<9>, dim3(g), dim3(RB), 0, ST, qkv, ldq, dout, ldo, dqkv, lddq, scratch, S, E, heads);
  } else {
    FFSR_LAUNCH(pixel_mha_bwd_a_kernel<4>, dim3(g), dim3(RB), 0, ST, qkv, ldq, dout, ldo, dqkv, lddq, scratch, S, E, heads, thr, ks, (unsigned long long)seed);
    FFSR_LAUNCH(pixel_mha_bwd_b_kernel<4>, dim3(g), dim3(RB), 0, ST, qkv, ldq, dout, ldo, dqkv, lddq, scratch, S, E, heads);
  }
  return ffsr_launch_status();
}

extern "C" int ffsr_softmax_c_f32(const float* x, int ldx, float* y, int ldy, long long M, int C, void* stream) {
  FFSR_CHECK(x && y && M > 0 && C >= 1 && C <= 8 && ldx >= C && ldy >= C);
  FFSR_LAUNCH(softmax_c_kernel, dim3(grid_for(M)), dim3(RB), 0, ST, x, ldx, y, ldy, M, C);
  return ffsr_launch_status();
}
extern "C" int ffsr_softmax_c_bwd_f32(const float* y, int ldy, const float* dy, int lddy, float* dx, int ldx, long long M, int C,
                                      void* stream) {
  FFSR_CHECK(y && dy && dx && M > 0 && C >= 1 && C <= 8 && ldy >= C && lddy >= C && ldx >= C);
  FFSR_LAUNCH(softmax_c_bwd_kernel, dim3(grid_for(M)), dim3(RB), 0, ST, y, ldy, dy, lddy, dx, ldx, M, C);
  return ffsr_launch_status();
}

extern "C" int ffsr_expert_sum_f32(const float* x, int ldx, const float* g, int ldg, float* out, int ldo, long long M,
                                   int normalize, void* stream) {
  FFSR_CHECK(x && g && out && M > 0 && ldx >= 12 && ldg >= 4 && ldo >= 3);
  FFSR_LAUNCH(expert_sum_kernel, dim3(grid_for(M)), dim3(RB), 0, ST, x, ldx, g, ldg, out, ldo, M, normalize);
  return ffsr_launch_status();
}
extern "C" int ffsr_expert_sum_bwd_f32(const float* x, int ldx, const float* g, int ldg, const float* dy, int lddy, float* dx,
                                       int lddx, float* dg, int lddg, long long M, int normalize, int accumulate_dx,
                                       void* stream) {
  FFSR_CHECK(x && g && dy && dx && dg && M > 0 && ldx >= 12 && ldg >= 4 && lddy >= 3 && lddx >= 12 && lddg >= 4);
  FFSR_LAUNCH(expert_sum_bwd_kernel, dim3(grid_for(M)), dim3(RB), 0, ST, x, ldx, g, ldg, dy, lddy, dx, lddx, dg, lddg, M,
              normalize, accumulate_dx);
  return ffsr_launch_status();
}

// backward of ffsr_selector_gates_f32: draw [M, 4], ddiff [M, 1], dtemperature[0] += ...; partial: n_partial doubles
extern "C" int ffsr_selector_gates_bwd_f32(const float* raw, int ldr, const float* diff, int ldd, const float* temperature,
                                           const float* dgates, int ldg, float* draw, int lddr, float* ddiff, int lddd,
                                           double* partial, int n_partial, float* dtemperature, long long M, void* stream) {
  FFSR_CHECK(raw && diff && temperature && dgates && draw && ddiff && partial && dtemperature && M > 0 && ldr >= 4 && ldd >= 1 &&
             ldg >= 4 && lddr >= 4 && lddd >= 1 && n_partial >= 1);
  long long want = (M + RB - 1) / RB;
  const int nb = (int)(want < n_partial ? want : n_partial);
  FFSR_LAUNCH(selector_gates_bwd_kernel, dim3(nb), dim3(RB), 0, ST, raw, ldr, diff, ldd, temperature, dgates, ldg, draw, lddr,
              ddiff, lddd, partial, M);
  FFSR_LAUNCH(dot_finish_kernel, dim3(1), dim3(RB), 0, ST, partial, nb, dtemperature, 1.f, 1);
  return ffsr_launch_status();
}

// dmask [H, W/2+1] = gradient of the loss w.r.t. the FFT band mask; xlo / xhi = the masked spectra of the forward pass
// (work + 2 n and work + 4 n floats of ffsr_fft_bands_f32's work buffer, n = B*3*H*(W/2+1)), ghat = ortho rfft2 of
// G = dL/dlo - dL/dhi (ffsr_rfft2_ortho_f32), all complex [B*3, H, W/2+1].
extern "C" int ffsr_fft_mask_grad_f32(const float* xlo, const float* xhi, const float* ghat, float* dmask, int B, int H, int W,
                                      void* stream) {
  FFSR_CHECK(xlo && xhi && ghat && dmask && B > 0 && H > 1 && W > 1);
  const int Wf = W / 2 + 1;
  FFSR_LAUNCH(fft_mask_grad_kernel, dim3(grid_for((long long)H * Wf)), dim3(RB), 0, ST, (const float2*)xlo, (const float2*)xhi,
              (const float2*)ghat, dmask, B * 3, H, W, Wf);
  return ffsr_launch_status();
}
