// HBM-bound row/pixel kernels on channels-last (NHWC) fp32 tensors, gfx950:
// LayerNorm over channels, elementwise combinators, column means, depthwise convolutions (incl. NAFNet's
// dw3x3 + SimpleGate + pooled partial sums), bilinear / bicubic resamplers, 2x2 average pooling.
// A tensor is a matrix [M pixels, C channels] with a row stride ld (floats, multiple of 4 so that every row is
// 16-byte aligned); lanes run along channels so every wave access is a contiguous row segment.
#include "ffsr_common.h"
#include <cstdlib>

namespace {

template <int V> struct Vec;
template <> struct Vec<4> { using T = floatx4; };
template <> struct Vec<1> { using T = float; };

template <int V> __device__ __forceinline__ typename Vec<V>::T ld(const float* p) {
  return *reinterpret_cast<const typename Vec<V>::T*>(p);
}
template <int V> __device__ __forceinline__ void st(float* p, typename Vec<V>::T v) {
  *reinterpret_cast<typename Vec<V>::T*>(p) = v;
}
__device__ __forceinline__ float elem(const float& v, int) { return v; }
__device__ __forceinline__ float elem(const floatx4& v, int i) { return v[i]; }
__device__ __forceinline__ void set(float& v, int, float x) { v = x; }
__device__ __forceinline__ void set(floatx4& v, int i, float x) { v[i] = x; }

inline int grid_for(long long n, int block = 256) { return (int)((n + block - 1) / block); }

// ------------------------------------------------------------------------------------------- LayerNorm
// one wave per row, up to 16 elements per lane (C <= 1024)
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ g,
                                                        const float* __restrict__ b, float eps, float* __restrict__ out,
                                                        int ldo, const float* __restrict__ r1, int ldr1,
                                                        const float* __restrict__ r2, int ldr2, int M, int C) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* xr = x + (size_t)row * ldx;
  float v[16];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    int c = lane + 64 * i;
    v[i] = c < C ? xr[c] : 0.f;
    s += v[i];
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    int c = lane + 64 * i;
    float d = c < C ? v[i] - mean : 0.f;
    q += d * d;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    int c = lane + 64 * i;
    if (c < C) {
      float y = (v[i] - mean) * rstd * g[c] + b[c];
      if (r1) y += r1[(size_t)row * ldr1 + c];
      if (r2) y += r2[(size_t)row * ldr2 + c];
      out[(size_t)row * ldo + c] = y;
    }
  }
}

// Vectorised form (C % 4 == 0, 16-byte aligned rows): HALF a wave per row, 8 channels per lane and pass (two 16-byte
// loads), NPASS passes cover C <= 256 * NPASS.  Outputs: fp32 `out` (optional) and / or the bf16 hi / lo planes of the
// result (optional; [M, ldp], ldp % 32 == 0, columns C..ldp-1 written as zeros) for ffsr_conv2d_planes.
template <int NPASS, int LPR = 32>   // LPR lanes per row (32, or 16 / 8 for C <= 128 / 64 so that narrow rows fill the wave)
__global__ __launch_bounds__(256) void layernorm_v8_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ g,
                                                           const float* __restrict__ b, float eps, float* __restrict__ out,
                                                           int ldo, unsigned short* __restrict__ ohi,
                                                           unsigned short* __restrict__ olo, int ldp,
                                                           const float* __restrict__ r1, int ldr1,
                                                           const float* __restrict__ r2, int ldr2,
                                                           const float* __restrict__ r2vec, int rpb, int M, int C) {
  static_assert(LPR == 32 || NPASS == 1, "narrow rows are single pass");
  const int row = blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
  const int l = threadIdx.x % LPR;
  if (row >= M) return;     // whole lane groups leave together: the shuffles below stay inside a group
  const float* xr = x + (size_t)row * ldx;
  floatx4 v[NPASS][2];
  float s = 0.f;
#pragma unroll
  for (int p = 0; p < NPASS; ++p)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int c = (p * LPR + l) * 8 + 4 * hh;
      v[p][hh] = c < C ? *reinterpret_cast<const floatx4*>(xr + c) : floatx4{0.f, 0.f, 0.f, 0.f};
      s += (v[p][hh][0] + v[p][hh][1]) + (v[p][hh][2] + v[p][hh][3]);
    }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int p = 0; p < NPASS; ++p)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int c = (p * LPR + l) * 8 + 4 * hh;
      if (c < C) {
        const floatx4 d = v[p][hh] - mean;
        q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
      }
    }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = 1.0f / sqrtf(q / (float)C + eps);
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
    const int c0 = (p * LPR + l) * 8;
    float y[8];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int c = c0 + 4 * hh;
      floatx4 t = {0.f, 0.f, 0.f, 0.f};
      if (c < C) {
        t = (v[p][hh] - mean) * rstd * *reinterpret_cast<const floatx4*>(g + c) + *reinterpret_cast<const floatx4*>(b + c);
        if (r1) t += *reinterpret_cast<const floatx4*>(r1 + (size_t)row * ldr1 + c);
        if (r2) {
          floatx4 t2 = *reinterpret_cast<const floatx4*>(r2 + (size_t)row * ldr2 + c);
          if (r2vec) t2 *= *reinterpret_cast<const floatx4*>(r2vec + (size_t)(row / rpb) * C + c);   // per-(batch, channel) scale
          t += t2;
        }
        if (out) *reinterpret_cast<floatx4*>(out + (size_t)row * ldo + c) = t;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) y[4 * hh + e] = t[e];
    }
    if (ohi && c0 < ldp) ffsr_store_planes8(ohi + (size_t)row * ldp + c0, olo + (size_t)row * ldp + c0, y);
  }
}

// ------------------------------------------------------------------------------------------- elementwise
// out = clamp( act(x * pre) * alpha * cscale[n] + beta + cbias[n] )
template <int V>
__global__ void unary_kernel(const float* __restrict__ x, int ldx, float* __restrict__ out, int ldo, long long M, int C,
                             int act, float slope, float pre, float alpha, float beta, const float* __restrict__ cscale, const float* __restrict__ cbias,
                             int do_clamp, float lo, float hi) {
  const int cv = C / V;
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * cv) return;
  long long m = idx / cv;
  int c = (int)(idx - m * cv) * V;
  auto v = ld<V>(x + m * ldx + c);
#pragma unroll
  for (int i = 0; i < V; ++i) {
    float y = ffsr_act(elem(v, i) * pre, act, slope) * alpha;
    if (cscale) y *= cscale[c + i];
    y += beta;
    if (cbias) y += cbias[c + i];
    if (do_clamp) y = fminf(fmaxf(y, lo), hi);
    set(v, i, y);
  }
  st<V>(out + m * ldo + c, v);
}

// out = alpha * a * avec[n] + beta * b * bvec[batch(m), n]      (b / avec / bvec optional)
template <int V>
__global__ void scale_add_kernel(const float* __restrict__ a, int lda, const float* __restrict__ avec,
                                 const float* __restrict__ b, int ldb, const float* __restrict__ bvec, int rows_per_batch,
                                 float* __restrict__ out, int ldo, long long M, int C, float alpha, float beta) {
  const int cv = C / V;
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * cv) return;
  long long m = idx / cv;
  int c = (int)(idx - m * cv) * V;
  auto va = ld<V>(a + m * lda + c);
  typename Vec<V>::T vb = va;
  if (b) vb = ld<V>(b + m * ldb + c);
  const float* bv = bvec ? bvec + (size_t)(m / rows_per_batch) * C : nullptr;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    float y = alpha * elem(va, i) * (avec ? avec[c + i] : 1.f);
    if (b) y += beta * elem(vb, i) * (bv ? bv[c + i] : 1.f);
    set(va, i, y);
  }
  st<V>(out + m * ldo + c, va);
}

// out = alpha * a * b' + gamma * c     b' = b[m, n] (bmode 0) or b[m] broadcast over channels (bmode 1); c optional
template <int V>
__global__ void mul_add_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb, int bmode,
                               const float* __restrict__ cc, int ldc, float* __restrict__ out, int ldo, long long M, int C,
                               float alpha, float gamma) {
  const int cv = C / V;
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * cv) return;
  long long m = idx / cv;
  int c = (int)(idx - m * cv) * V;
  auto va = ld<V>(a + m * lda + c);
  typename Vec<V>::T vb = va, vc = va;
  float bs = 0.f;
  if (bmode == 0) vb = ld<V>(b + m * ldb + c);
  else bs = b[m * ldb];
  if (cc) vc = ld<V>(cc + m * ldc + c);
#pragma unroll
  for (int i = 0; i < V; ++i) {
    float y = alpha * elem(va, i) * (bmode == 0 ? elem(vb, i) : bs);
    if (cc) y += gamma * elem(vc, i);
    set(va, i, y);
  }
  st<V>(out + m * ldo + c, va);
}

// ------------------------------------------------------------------------------------------- column sums
// part[b, chunk, c] = sum over the rows of the chunk;  grid = (ceil(C/64), nchunk, B), block 256 = 64 ch x 4 row lanes
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int ldx, float* __restrict__ part,
                                                             int R, int C, int nchunk) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int chunk = blockIdx.y, b = blockIdx.z;
  const int per = (R + nchunk - 1) / nchunk;
  const int r0 = chunk * per, r1 = min(R, r0 + per);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;   // four independent chains: the loads of one trip are in flight together
  if (c < C) {
    const float* xp = x + (size_t)b * R * ldx + c;
    int r = r0 + rl;
    for (; r + 12 < r1; r += 16) {
      s0 += xp[(size_t)r * ldx];
      s1 += xp[(size_t)(r + 4) * ldx];
      s2 += xp[(size_t)(r + 8) * ldx];
      s3 += xp[(size_t)(r + 12) * ldx];
    }
    for (; r < r1; r += 4) s0 += xp[(size_t)r * ldx];
  }
  red[rl][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rl == 0 && c < C)
    part[((size_t)b * nchunk + chunk) * C + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// out[b, c] = scale * sum_k part[b, k, c]; grid = (ceil(C/64), B), block 1024 = 64 channels x 16 chunk lanes
__global__ __launch_bounds__(1024) void colsum_finish_kernel(const float* __restrict__ part, float* __restrict__ out, int B, int C,
                                                             int nchunk, float scale) {
  __shared__ float red[16][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int kl = threadIdx.x >> 6, b = blockIdx.y;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < C) {
    const float* pp = part + (size_t)b * nchunk * C + c;
    int k = kl;
    for (; k + 48 < nchunk; k += 64) {
      s0 += pp[(size_t)k * C];
      s1 += pp[(size_t)(k + 16) * C];
      s2 += pp[(size_t)(k + 32) * C];
      s3 += pp[(size_t)(k + 48) * C];
    }
    for (; k < nchunk; k += 16) s0 += pp[(size_t)k * C];
  }
  red[kl][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (kl == 0 && c < C) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += red[i][threadIdx.x];
    out[(size_t)b * C + c] = s * scale;
  }
}

// RCAN channel attention tail in one workgroup per batch: finish the spatial mean from the partial column sums, then
//   att[c] = sigmoid(W2 relu(W1 mean + b1) + b2)          (mambair_arch.py:20-38, grl mixed_attn_block.py:942-961)
// w1 [sq, ldw1] (row j = squeeze unit j over the C channels), w2 [C, ldw2].  C <= 1024, sq <= 64.  The four launches this
// replaces (column-sum finish, two M = 1 GEMMs on a 64 x 64 tile: 12 + 27 + 13 us of pure latency) sat on the critical path
// of every GRL / MambaIR block.
__global__ __launch_bounds__(1024) void channel_attention_kernel(const float* __restrict__ part, int nchunk, float inv_rows,
                                                                 const float* __restrict__ w1, int ldw1, const float* __restrict__ b1,
                                                                 const float* __restrict__ w2, int ldw2, const float* __restrict__ b2,
                                                                 float* __restrict__ out, int C, int sq) {
  __shared__ float red[16][64];
  __shared__ float mean[1024];
  __shared__ float hid[64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int cl = tid & 63, kl = tid >> 6;                       // 64 channels x 16 chunk lanes
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + cl;
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
      const float* pp = part + (size_t)b * nchunk * C + c;
      int k = kl;
      for (; k + 16 < nchunk; k += 32) {
        s0 += pp[(size_t)k * C];
        s1 += pp[(size_t)(k + 16) * C];
      }
      for (; k < nchunk; k += 16) s0 += pp[(size_t)k * C];
    }
    red[kl][cl] = s0 + s1;
    __syncthreads();
    if (kl == 0 && c < C) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) s += red[i][cl];
      mean[c] = s * inv_rows;
    }
    __syncthreads();
  }
  // squeeze: one wave per hidden unit (16 waves)
  for (int j = kl; j < sq; j += 16) {
    float a = 0.f;
    for (int c = cl; c < C; c += 64) a = fmaf(w1[(size_t)j * ldw1 + c], mean[c], a);
    a = wave_sum(a);
    if (cl == 0) hid[j] = fmaxf(a + (b1 ? b1[j] : 0.f), 0.f);
  }
  __syncthreads();
  for (int c = tid; c < C; c += 1024) {
    float a = b2 ? b2[c] : 0.f;
    for (int j = 0; j < sq; ++j) a = fmaf(w2[(size_t)c * ldw2 + j], hid[j], a);
    out[(size_t)b * C + c] = 1.0f / (1.0f + expf(-a));
  }
}

// ------------------------------------------------------------------------------------------- depthwise conv
// weights tap-major [KH*KW, C]; stride 1; zero padding
template <int V>
__global__ void dwconv_kernel(const float* __restrict__ in, int ldi, const float* __restrict__ w, const float* __restrict__ bias,
                              float* __restrict__ out, int ldo, int B, int H, int W, int C, int KH, int KW, int ph, int pw,
                              int act) {
  const int cv = C / V;
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)B * H * W * cv;
  if (idx >= total) return;
  long long pix = idx / cv;
  int c = (int)(idx - pix * cv) * V;
  int x = (int)(pix % W);
  long long t = pix / W;
  int y = (int)(t % H);
  int b = (int)(t / H);
  typename Vec<V>::T acc;
#pragma unroll
  for (int i = 0; i < V; ++i) set(acc, i, bias ? bias[c + i] : 0.f);
  for (int ky = 0; ky < KH; ++ky) {
    int yy = y + ky - ph;
    if (yy < 0 || yy >= H) continue;
    for (int kx = 0; kx < KW; ++kx) {
      int xx = x + kx - pw;
      if (xx < 0 || xx >= W) continue;
      auto v = ld<V>(in + (((size_t)b * H + yy) * W + xx) * ldi + c);
      auto wv = ld<V>(w + (size_t)(ky * KW + kx) * C + c);
#pragma unroll
      for (int i = 0; i < V; ++i) set(acc, i, fmaf(elem(v, i), elem(wv, i), elem(acc, i)));
    }
  }
#pragma unroll
  for (int i = 0; i < V; ++i) set(acc, i, ffsr_act(elem(acc, i), act, 0.f));
  st<V>(out + pix * ldo + c, acc);
}

// Depthwise 3x3 (stride 1, zero padding 1), lane = channel, one wave = a run of consecutive pixels, 4 pixels per trip:
// the 3 x 6 input window of the trip is fetched with 18 (x2 for the gate) independent, branch-free loads (clamped
// coordinates, values zeroed by a 0/1 mask: a load under a bounds branch is waited for one at a time), 9 loads per
// output pixel instead of 18.
//   GATE (NAFNet, nafnet_arch.py:118-121): t = dw3x3(in [.., 2C]); out[.., c] = t[c] * t[c + C];
//                                          part[b, chunk, c] = sum of out over the chunk's pixels (SCA pooling)
//   else (MambaIR, mambair_arch.py:239-247,378): out = act(dw3x3(in [.., C]))
// grid = (ceil(C/64), nchunk, B); block 256 = 4 waves, each a contiguous quarter of the chunk's pixel range.
template <bool GATE>
__global__ __launch_bounds__(256) void dw3x3_run_kernel(const float* __restrict__ in, int ldi, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out, int ldo,
                                                        float* __restrict__ part, int H, int W, int C, int nchunk, int act) {
  __shared__ float red[4][64];
  constexpr int NH = GATE ? 2 : 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs; give every XCD a contiguous band of the image
  // (consecutive chunks = vertically adjacent pixels) so that the 3 input rows a chunk needs are shared in ONE L2.
  // (A float4-per-lane form of this kernel -- 16-byte loads, 64/G pixel runs per wave -- measured 20 % slower: ~200
  // VGPRs for the 3 x 4 x 2 vector window leave 2 waves per SIMD.)
  const int nxy = gridDim.x * gridDim.y, lin = blockIdx.x + gridDim.x * blockIdx.y;
  const int q8 = nxy >> 3, r8 = nxy & 7, xcd = lin & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin >> 3);
  const int c = (logical % gridDim.x) * 64 + lane;
  const bool live = c < C;
  const int cc = live ? c : C - 1;
  const int chunk = logical / gridDim.x, b = blockIdx.z;
  const int R = H * W;
  const int per = (R + nchunk - 1) / nchunk;
  const int q = (per + 3) / 4;
  const int r0 = chunk * per + wave * q, r1 = min(min(R, (chunk + 1) * per), r0 + q);
  const int CW = NH * C;
  float wt[NH][9], bs[NH];
#pragma unroll
  for (int hh = 0; hh < NH; ++hh) {
#pragma unroll
    for (int k = 0; k < 9; ++k) wt[hh][k] = w[k * CW + hh * C + cc];
    bs[hh] = bias ? bias[hh * C + cc] : 0.f;
  }
  const float* inb = in + (size_t)b * R * ldi + cc;
  float* outb = out + (size_t)b * R * ldo + c;
  float s = 0.f;
  for (int r = r0; r < r1; r += 4) {
    const int y = r / W, x = r - y * W;
    const int np = min(4, r1 - r);
    if (x + np <= W) {   // the trip stays inside one image row (wave-uniform)
      float v[NH][3][6];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int yy = y + ky - 1;
        const bool yok = yy >= 0 && yy < H;
        const float* rowp = inb + (size_t)min(max(yy, 0), H - 1) * W * ldi;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int xx = x + j - 1;
          const float m = (yok && xx >= 0 && xx < W) ? 1.f : 0.f;
          const float* p = rowp + (size_t)min(max(xx, 0), W - 1) * ldi;
#pragma unroll
          for (int hh = 0; hh < NH; ++hh) v[hh][ky][j] = p[hh * C] * m;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j < np) {
          float a[NH];
#pragma unroll
          for (int hh = 0; hh < NH; ++hh) {
            a[hh] = bs[hh];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
              for (int kx = 0; kx < 3; ++kx) a[hh] = fmaf(v[hh][ky][j + kx], wt[hh][ky * 3 + kx], a[hh]);
          }
          const float g = GATE ? a[0] * a[NH - 1] : ffsr_act(a[0], act, 0.f);
          if (live) outb[(size_t)(r + j) * ldo] = g;
          s += g;
        }
      }
    } else {             // the trip crosses a row end: pixel by pixel
      for (int j = 0; j < np; ++j) {
        const int rr = r + j, y2 = rr / W, x2 = rr - y2 * W;
        float a[NH];
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) a[hh] = bs[hh];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int yy = y2 + ky - 1;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int xx = x2 + kx - 1;
            const float m = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? 1.f : 0.f;
            const float* p = inb + ((size_t)min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1)) * ldi;
#pragma unroll
            for (int hh = 0; hh < NH; ++hh) a[hh] = fmaf(p[hh * C] * m, wt[hh][ky * 3 + kx], a[hh]);
          }
        }
        const float g = GATE ? a[0] * a[NH - 1] : ffsr_act(a[0], act, 0.f);
        if (live) outb[(size_t)rr * ldo] = g;
        s += g;
      }
    }
  }
  if (GATE) {
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && live)
      part[((size_t)b * nchunk + chunk) * C + c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
  }
}

// Same operator, register-window form: one wave = a strip of R = 4 image rows x XS columns of 64 channels.  A (R + 2) x 3 input
// window per lane slides along x: every step fetches ONE new column (R + 2 loads per input half) and emits R output pixels, i.e.
// 1.5 loads per output instead of the run kernel's 4.5 (its 3 x 6 window serves 4 pixels of one row) -- the run kernel is bound by
// the L2 -> L1 refill of its window overlap (1063 us for NAFNet's 1408 x 2048 x 128 level, 2.1 TB/s of algorithmic traffic).
// The columns of the next U = 4 (2 with the gate's two input halves) steps are in flight while the current U are computed.  The 4 waves of a workgroup take 4
// vertically adjacent strips (their halo rows hit in L1 / the XCD's L2).  Same grid contract as dw3x3_run_kernel.
template <bool GATE>
__global__ __launch_bounds__(256, GATE ? 3 : 2) void dw3x3_strip_kernel(const float* __restrict__ in, int ldi, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ out, int ldo,
                                                          float* __restrict__ part, int H, int W, int C, int nchunk, int act,
                                                          int XS) {
  __shared__ float red[4][64];
  constexpr int NH = GATE ? 2 : 1, R = 4, U = GATE ? 2 : 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nxy = gridDim.x * gridDim.y, lin = blockIdx.x + gridDim.x * blockIdx.y;
  const int q8 = nxy >> 3, r8 = nxy & 7, xcd = lin & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin >> 3);
  const int c = (logical % gridDim.x) * 64 + lane;
  const bool live = c < C;
  const int cc = live ? c : C - 1;
  const int chunk = logical / gridDim.x, b = blockIdx.z;
  const int CW = NH * C;
  float wt[NH][9], bs[NH];
#pragma unroll
  for (int hh = 0; hh < NH; ++hh) {
#pragma unroll
    for (int k = 0; k < 9; ++k) wt[hh][k] = w[k * CW + hh * C + cc];
    bs[hh] = bias ? bias[hh * C + cc] : 0.f;
  }
  // 32-bit element offsets from the (uniform) base pointers: scalar-base + vector-offset loads, half the address registers
  // (the launcher takes this kernel only while B * H * W * max(ldi, ldo) < 2^31)
  const unsigned inb = (unsigned)b * H * W * ldi + cc;
  const unsigned outb = (unsigned)b * H * W * ldo + c;
  const int nsy = (H + R - 1) / R, nsx = (W + XS - 1) / XS, nstrip = nsy * nsx;
  float s = 0.f;
  for (int sidx = chunk * 4 + wave; sidx < nstrip; sidx += nchunk * 4) {
    const int sx = sidx / nsy, sy = sidx - sx * nsy;          // vertically adjacent strips are consecutive
    const int y0 = sy * R, x0 = sx * XS, x1 = min(W, x0 + XS);
    unsigned rowp[R + 2];
    float rmask[R + 2];
#pragma unroll
    for (int r = 0; r < R + 2; ++r) {
      const int yy = y0 + r - 1;
      rmask[r] = (yy >= 0 && yy < H) ? 1.f : 0.f;
      rowp[r] = inb + (unsigned)min(max(yy, 0), H - 1) * W * ldi;
    }
    auto load_col = [&](int xx, float (&col)[NH][R + 2]) {      // branch-free: clamped address, value zeroed by the masks
      const float xm = (xx >= 0 && xx < W) ? 1.f : 0.f;
      const unsigned xo = (unsigned)min(max(xx, 0), W - 1) * ldi;
#pragma unroll
      for (int r = 0; r < R + 2; ++r)
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) col[hh][r] = in[rowp[r] + xo + hh * C] * (xm * rmask[r]);
    };
    float win[3][NH][R + 2];       // columns x - 1, x, x + 1
    load_col(x0 - 1, win[0]);
    load_col(x0, win[1]);
    auto consume = [&](const float (&buf)[U][NH][R + 2], int xb) {      // buf = columns xb + 1 .. xb + U: outputs at xb .. xb + U - 1
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int hh = 0; hh < NH; ++hh)
#pragma unroll
          for (int r = 0; r < R + 2; ++r) win[2][hh][r] = buf[u][hh][r];
        const int x = xb + u;
        if (x < x1) {
#pragma unroll
          for (int r = 0; r < R; ++r) {
            float a[NH];
#pragma unroll
            for (int hh = 0; hh < NH; ++hh) {
              a[hh] = bs[hh];
#pragma unroll
              for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) a[hh] = fmaf(win[kx][hh][r + ky], wt[hh][ky * 3 + kx], a[hh]);
            }
            const float g = GATE ? a[0] * a[NH - 1] : ffsr_act(a[0], act, 0.f);
            if (y0 + r < H) {
              if (live) out[outb + ((unsigned)(y0 + r) * W + x) * ldo] = g;
              s += g;
            }
          }
        }
#pragma unroll
        for (int hh = 0; hh < NH; ++hh)
#pragma unroll
          for (int r = 0; r < R + 2; ++r) {
            win[0][hh][r] = win[1][hh][r];
            win[1][hh][r] = win[2][hh][r];
          }
      }
    };
    // two column buffers in alternation: the loads of one are in flight while the other is consumed (past the strip's end the
    // clamped addresses re-read valid pixels whose values nobody uses)
    float bufA[U][NH][R + 2], bufB[U][NH][R + 2];
#pragma unroll
    for (int u = 0; u < U; ++u) load_col(x0 + 1 + u, bufA[u]);
    for (int xb = x0; xb < x1; xb += 2 * U) {
#pragma unroll
      for (int u = 0; u < U; ++u) load_col(xb + U + 1 + u, bufB[u]);
      consume(bufA, xb);
#pragma unroll
      for (int u = 0; u < U; ++u) load_col(xb + 2 * U + 1 + u, bufA[u]);
      consume(bufB, xb + U);
    }
  }
  if (GATE) {
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && live)
      part[((size_t)b * nchunk + chunk) * C + c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
  }
}

// ------------------------------------------------------------------------------------------- resamplers
// torch F.interpolate(mode='bilinear', align_corners=False, size=(Ho,Wo)): src = (dst+0.5)*in/out - 0.5 clamped at 0
__device__ __forceinline__ void bilin_coord(int d, float scale, int n, int& i0, int& i1, float& l) {
  float s = ((float)d + 0.5f) * scale - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > n - 1) i0 = n - 1;
  i1 = i0 + (i0 < n - 1 ? 1 : 0);
  l = s - (float)i0;
}
template <int V>
__global__ void bilinear_kernel(const float* __restrict__ in, int ldi, float* __restrict__ out, int ldo, int B, int Hi, int Wi,
                                int Ho, int Wo, int C, float sh, float sw, float mul, int accumulate) {
  const int cv = C / V;
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)B * Ho * Wo * cv;
  if (idx >= total) return;
  long long pix = idx / cv;
  int c = (int)(idx - pix * cv) * V;
  int x = (int)(pix % Wo);
  long long t = pix / Wo;
  int y = (int)(t % Ho);
  int b = (int)(t / Ho);
  int y0, y1, x0, x1;
  float ly, lx;
  bilin_coord(y, sh, Hi, y0, y1, ly);
  bilin_coord(x, sw, Wi, x0, x1, lx);
  const float* base = in + (size_t)b * Hi * Wi * ldi + c;
  auto v00 = ld<V>(base + ((size_t)y0 * Wi + x0) * ldi);
  auto v01 = ld<V>(base + ((size_t)y0 * Wi + x1) * ldi);
  auto v10 = ld<V>(base + ((size_t)y1 * Wi + x0) * ldi);
  auto v11 = ld<V>(base + ((size_t)y1 * Wi + x1) * ldi);
  float* op = out + pix * ldo + c;
  typename Vec<V>::T o;
  if (accumulate) o = ld<V>(op);
  const float hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    // same association as ATen's upsample_bilinear2d: hy*(hx*v00 + lx*v01) + ly*(hx*v10 + lx*v11)
    float r = hy * (hx * elem(v00, i) + lx * elem(v01, i)) + ly * (hx * elem(v10, i) + lx * elem(v11, i));
    r *= mul;
    set(o, i, accumulate ? elem(o, i) + r : r);
  }
  st<V>(op, o);
}

// torch bicubic (A = -0.75), align_corners=False, scale_factor=4: src = (dst+0.5)/4 - 0.5, taps clamped to the border
__device__ __forceinline__ void cubic_w(float t, float w[4]) {
  const float A = -0.75f;
  float x = t + 1.f;
  w[0] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
  x = t;
  w[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  x = 1.f - t;
  w[2] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  x = 2.f - t;
  w[3] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
}
__global__ void bicubic_up_kernel(const float* __restrict__ in, int ldi, float* __restrict__ out, int ldo, int B, int H, int W,
                                  int C, int scale) {
  const int Ho = H * scale, Wo = W * scale;
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)B * Ho * Wo * C;
  if (idx >= total) return;
  long long pix = idx / C;
  int c = (int)(idx - pix * C);
  int x = (int)(pix % Wo);
  long long t = pix / Wo;
  int y = (int)(t % Ho);
  int b = (int)(t / Ho);
  const float inv = 1.0f / (float)scale;
  float sy = ((float)y + 0.5f) * inv - 0.5f, sx = ((float)x + 0.5f) * inv - 0.5f;
  int iy = (int)floorf(sy), ix = (int)floorf(sx);
  float wy[4], wx[4];
  cubic_w(sy - (float)iy, wy);
  cubic_w(sx - (float)ix, wx);
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int yy = min(max(iy - 1 + j, 0), H - 1);
    float row = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int xx = min(max(ix - 1 + i, 0), W - 1);
      row += in[(((size_t)b * H + yy) * W + xx) * ldi + c] * wx[i];
    }
    acc += row * wy[j];
  }
  out[pix * ldo + c] = acc;
}

template <int V>
__global__ void avgpool2_kernel(const float* __restrict__ in, int ldi, float* __restrict__ out, int ldo, int B, int H, int W,
                                int C) {
  const int Ho = H / 2, Wo = W / 2, cv = C / V;
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)B * Ho * Wo * cv;
  if (idx >= total) return;
  long long pix = idx / cv;
  int c = (int)(idx - pix * cv) * V;
  int x = (int)(pix % Wo);
  long long t = pix / Wo;
  int y = (int)(t % Ho);
  int b = (int)(t / Ho);
  const float* p = in + (((size_t)b * H + 2 * y) * W + 2 * x) * ldi + c;
  auto a = ld<V>(p), b2 = ld<V>(p + ldi), c2 = ld<V>(p + (size_t)W * ldi), d = ld<V>(p + (size_t)W * ldi + ldi);
#pragma unroll
  for (int i = 0; i < V; ++i) set(a, i, (elem(a, i) + elem(b2, i) + elem(c2, i) + elem(d, i)) * 0.25f);
  st<V>(out + pix * ldo + c, a);
}

// ------------------------------------------------------------------------------------------- image I/O helpers
// uint8 HWC -> float/255 (io.py:100-104); channels beyond C in the output row are left untouched (zero padded buffer)
__global__ void u8_to_f32_kernel(const unsigned char* __restrict__ in, float* __restrict__ out, int ldo, long long M, int C) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * C) return;
  long long m = idx / C;
  int c = (int)(idx - m * C);
  out[m * ldo + c] = (float)in[idx] / 255.0f;
}
// clamp(0,1) * 255 -> round half to even -> uint8 (io.py:107-112; numpy round == rintf)
__global__ void f32_to_u8_kernel(const float* __restrict__ in, int ldi, unsigned char* __restrict__ out, long long M, int C) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * C) return;
  long long m = idx / C;
  int c = (int)(idx - m * C);
  float v = fminf(fmaxf(in[m * ldi + c], 0.f), 1.f) * 255.0f;
  out[idx] = (unsigned char)rintf(v);
}
// F.pad(x, (0, pw, 0, ph), mode='reflect') (io.py:71-78): right / bottom only
__global__ void pad_reflect_kernel(const float* __restrict__ in, int ldi, float* __restrict__ out, int ldo, int B, int H, int W,
                                   int Hp, int Wp, int C) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)B * Hp * Wp * C) return;
  int c = (int)(idx % C);
  long long pix = idx / C;
  int x = (int)(pix % Wp);
  long long t = pix / Wp;
  int y = (int)(t % Hp), b = (int)(t / Hp);
  int ys = y < H ? y : 2 * (H - 1) - y, xs = x < W ? x : 2 * (W - 1) - x;
  out[pix * ldo + c] = in[(((size_t)b * H + ys) * W + xs) * ldi + c];
}
// crop the top-left [Ho, Wo] window (io.py:81-83 and the feature crops :234,245,268)
__global__ void crop_kernel(const float* __restrict__ in, int ldi, float* __restrict__ out, int ldo, int B, int H, int W, int Ho,
                            int Wo, int C, int do_clamp) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)B * Ho * Wo * C) return;
  int c = (int)(idx % C);
  long long pix = idx / C;
  int x = (int)(pix % Wo);
  long long t = pix / Wo;
  int y = (int)(t % Ho), b = (int)(t / Ho);
  float v = in[(((size_t)b * H + y) * W + x) * ldi + c];
  if (do_clamp) v = fminf(fmaxf(v, 0.f), 1.f);
  out[pix * ldo + c] = v;
}

// out[b, i, j, c] (+)= scale * in[b, ay_i*i + ay_j*j + cy, ax_i*i + ax_j*j + cx, c]: flips / rot90 of the 8-fold
// geometric self-ensemble (scripts/extract_test_tta_cache.py:97-104, scripts/generate_fast_submission.py:55-61)
__global__ void dihedral_kernel(const float* __restrict__ in, int ldi, float* __restrict__ out, int ldo, int B, int Hi, int Wi,
                                int Ho, int Wo, int C, int ay_i, int ay_j, int cy, int ax_i, int ax_j, int cx, float scale,
                                int accumulate) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)B * Ho * Wo * C) return;
  int c = (int)(idx % C);
  long long pix = idx / C;
  int j = (int)(pix % Wo);
  long long t = pix / Wo;
  int i = (int)(t % Ho), b = (int)(t / Ho);
  int y = ay_i * i + ay_j * j + cy, x = ax_i * i + ax_j * j + cx;
  float v = in[(((size_t)b * Hi + y) * Wi + x) * ldi + c] * scale;
  float* o = out + pix * ldo + c;
  *o = accumulate ? *o + v : v;
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// FFSR_DW_STRIP=0 keeps the run kernel everywhere (A/B measurements, tools/dw_bench.py)
inline bool dw3x3_strip_enabled() {
  static const int on = [] {
    const char* e = getenv("FFSR_DW_STRIP");
    return (e && e[0] == '0') ? 0 : 1;
  }();
  return on != 0;
}

inline int dw3x3_strip_maxc() {
  static const int c = [] {
    const char* e = getenv("FFSR_DW_STRIP_MAXC");
    return e ? atoi(e) : 64;
  }();
  return c;
}

// strip width of dw3x3_strip_kernel: the launch should have >= 8192 waves (measured: 1408 x 2048 x 64 gated 1504 / 830 / 506 /
// 566 us with strips cut for 2048 / 4096 / 8192 / 16384 waves -- short strips keep more independent load streams in flight and
// balance better than long ones save on their 2-column lead-in)
inline int dw3x3_strip_width(int H, int W, int cgroups, int B) {
  const long long rows = (long long)((H + 3) / 4) * cgroups * B;
  static const int target = [] {
    const char* e = getenv("FFSR_DW_STRIP_WAVES");
    return e ? atoi(e) : 8192;
  }();
  int xs = 512;
  while (xs > 32 && rows * ((W + xs - 1) / xs) < target) xs >>= 1;
  return xs;
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int ffsr_layernorm_planes_f32(const float* x, int ldx, const float* gamma, const float* beta, float eps,
                                         float* out, int ldo, void* out_hi, void* out_lo, int ldp, const float* res1,
                                         int ldr1, const float* res2, int ldr2, const float* res2_vec, int rows_per_batch,
                                         int M, int C, void* stream) {
  const float* r2vec = res2_vec;
  const int rpb = rows_per_batch > 0 ? rows_per_batch : M;
  FFSR_CHECK(!r2vec || (res2 && al16(r2vec)));
  FFSR_CHECK(x && gamma && beta && (out || (out_hi && out_lo)) && M > 0 && C > 0 && C <= 1024 && ldx >= C);
  FFSR_CHECK(!out || ldo >= C);
  FFSR_CHECK(!out_hi || (out_lo && (ldp & 31) == 0 && ldp >= C && ldp < C + 32 && al16(out_hi) && al16(out_lo)));
  const bool v8 = (C % 4 == 0) && (ldx % 4 == 0) && al16(x) && al16(gamma) && al16(beta) && (!out || (ldo % 4 == 0 && al16(out))) &&
                  (!res1 || (ldr1 % 4 == 0 && al16(res1))) && (!res2 || (ldr2 % 4 == 0 && al16(res2)));
  if (!v8) {
    FFSR_CHECK(out && !out_hi && !r2vec);   // the scalar fallback writes fp32 only and has no scaled residual
    FFSR_LAUNCH(layernorm_kernel, dim3((M + 3) / 4), dim3(256), 0, ST, x, ldx, gamma, beta, eps, out, ldo, res1,
                       ldr1, res2, ldr2, M, C);
    return ffsr_launch_status();
  }
  unsigned short* oh = (unsigned short*)out_hi;
  unsigned short* ol = (unsigned short*)out_lo;
  const dim3 block(256);
  dim3 grid((M + 7) / 8);
  if (C <= 128) {   // narrow rows: 8 / 16 lanes per row
    const int lpr = C <= 64 ? 8 : 16;
    grid = dim3((M + 256 / lpr - 1) / (256 / lpr));
    if (lpr == 8)
      FFSR_LAUNCH((layernorm_v8_kernel<1, 8>), grid, block, 0, ST, x, ldx, gamma, beta, eps, out, ldo, oh, ol, ldp, res1,
                         ldr1, res2, ldr2, r2vec, rpb, M, C);
    else
      FFSR_LAUNCH((layernorm_v8_kernel<1, 16>), grid, block, 0, ST, x, ldx, gamma, beta, eps, out, ldo, oh, ol, ldp, res1,
                         ldr1, res2, ldr2, r2vec, rpb, M, C);
    return ffsr_launch_status();
  }
#define FFSR_LN(NP)                                                                                                      \
  FFSR_LAUNCH(layernorm_v8_kernel<NP>, grid, block, 0, ST, x, ldx, gamma, beta, eps, out, ldo, oh, ol, ldp, res1, \
                     ldr1, res2, ldr2, r2vec, rpb, M, C)
  if (C <= 256) FFSR_LN(1);
  else if (C <= 512) FFSR_LN(2);
  else if (C <= 768) FFSR_LN(3);
  else FFSR_LN(4);
#undef FFSR_LN
  return ffsr_launch_status();
}

extern "C" int ffsr_layernorm_f32(const float* x, int ldx, const float* gamma, const float* beta, float eps, float* out,
                                  int ldo, const float* res1, int ldr1, const float* res2, int ldr2, int M, int C,
                                  void* stream) {
  FFSR_CHECK(out);
  return ffsr_layernorm_planes_f32(x, ldx, gamma, beta, eps, out, ldo, nullptr, nullptr, 0, res1, ldr1, res2, ldr2, nullptr, 0,
                                   M, C, stream);
}

extern "C" int ffsr_unary_f32(const float* x, int ldx, float* out, int ldo, long long M, int C, int act, float slope,
                              float pre, float alpha, float beta, const float* cscale, const float* cbias, int do_clamp,
                              float lo, float hi, void* stream) {
  FFSR_CHECK(x && out && M > 0 && C > 0);
  bool v4 = (C % 4 == 0) && (ldx % 4 == 0) && (ldo % 4 == 0) && al16(x) && al16(out);
  if (v4)
    FFSR_LAUNCH(unary_kernel<4>, dim3(grid_for(M * (C / 4))), dim3(256), 0, ST, x, ldx, out, ldo, M, C, act, slope,
                       pre, alpha, beta, cscale, cbias, do_clamp, lo, hi);
  else
    FFSR_LAUNCH(unary_kernel<1>, dim3(grid_for(M * C)), dim3(256), 0, ST, x, ldx, out, ldo, M, C, act, slope, pre, alpha,
                       beta, cscale, cbias, do_clamp, lo, hi);
  return ffsr_launch_status();
}

extern "C" int ffsr_scale_add_f32(const float* a, int lda, const float* avec, const float* b, int ldb, const float* bvec,
                                  int rows_per_batch, float* out, int ldo, long long M, int C, float alpha, float beta,
                                  void* stream) {
  FFSR_CHECK(a && out && M > 0 && C > 0 && rows_per_batch > 0);
  bool v4 = (C % 4 == 0) && (lda % 4 == 0) && (ldo % 4 == 0) && (!b || ldb % 4 == 0) && al16(a) && al16(out) && (!b || al16(b));
  if (v4)
    FFSR_LAUNCH(scale_add_kernel<4>, dim3(grid_for(M * (C / 4))), dim3(256), 0, ST, a, lda, avec, b, ldb, bvec,
                       rows_per_batch, out, ldo, M, C, alpha, beta);
  else
    FFSR_LAUNCH(scale_add_kernel<1>, dim3(grid_for(M * C)), dim3(256), 0, ST, a, lda, avec, b, ldb, bvec,
                       rows_per_batch, out, ldo, M, C, alpha, beta);
  return ffsr_launch_status();
}

extern "C" int ffsr_mul_add_f32(const float* a, int lda, const float* b, int ldb, int bmode, const float* c, int ldc,
                                float* out, int ldo, long long M, int C, float alpha, float gamma, void* stream) {
  FFSR_CHECK(a && b && out && M > 0 && C > 0 && (bmode == 0 || bmode == 1));
  bool v4 = (C % 4 == 0) && (lda % 4 == 0) && (ldo % 4 == 0) && (bmode == 1 || ldb % 4 == 0) && (!c || ldc % 4 == 0) &&
            al16(a) && al16(out) && (bmode == 1 || al16(b)) && (!c || al16(c));
  if (v4)
    FFSR_LAUNCH(mul_add_kernel<4>, dim3(grid_for(M * (C / 4))), dim3(256), 0, ST, a, lda, b, ldb, bmode, c, ldc, out,
                       ldo, M, C, alpha, gamma);
  else
    FFSR_LAUNCH(mul_add_kernel<1>, dim3(grid_for(M * C)), dim3(256), 0, ST, a, lda, b, ldb, bmode, c, ldc, out, ldo,
                       M, C, alpha, gamma);
  return ffsr_launch_status();
}

extern "C" int ffsr_colmean_f32(const float* x, int ldx, float* out, float* part, int B, int R, int C, int nchunk,
                                void* stream) {
  FFSR_CHECK(x && out && part && B > 0 && R > 0 && C > 0 && nchunk > 0 && nchunk <= 65535 && B <= 65535);
  FFSR_LAUNCH(colsum_partial_kernel, dim3((C + 63) / 64, nchunk, B), dim3(256), 0, ST, x, ldx, part, R, C, nchunk);
  FFSR_LAUNCH(colsum_finish_kernel, dim3((C + 63) / 64, B), dim3(1024), 0, ST, part, out, B, C, nchunk, 1.0f / (float)R);
  return ffsr_launch_status();
}

extern "C" int ffsr_channel_attention_f32(const float* x, int ldx, float* part, int nchunk, const float* w1, int ldw1,
                                          const float* b1, const float* w2, int ldw2, const float* b2, float* out, int B, int R,
                                          int C, int sq, void* stream) {
  FFSR_CHECK(x && part && w1 && w2 && out && B > 0 && R > 0 && C > 0 && C <= 1024 && sq > 0 && sq <= 64 && ldx >= C &&
             ldw1 >= C && ldw2 >= sq && nchunk > 0 && nchunk <= 65535 && B <= 65535);
  FFSR_LAUNCH(colsum_partial_kernel, dim3((C + 63) / 64, nchunk, B), dim3(256), 0, ST, x, ldx, part, R, C, nchunk);
  FFSR_LAUNCH(channel_attention_kernel, dim3(B), dim3(1024), 0, ST, part, nchunk, 1.0f / (float)R, w1, ldw1, b1, w2, ldw2, b2,
              out, C, sq);
  return ffsr_launch_status();
}

extern "C" int ffsr_dwconv2d_f32(const float* in, int ldi, const float* w, const float* bias, float* out, int ldo, int B,
                                 int H, int W, int C, int KH, int KW, int pad_h, int pad_w, int act, void* stream) {
  FFSR_CHECK(in && w && out && B > 0 && H > 0 && W > 0 && C > 0 && KH > 0 && KW > 0);
  long long pix = (long long)B * H * W;
  if (KH == 3 && KW == 3 && pad_h == 1 && pad_w == 1 && pix >= 4096 && B <= 65535) {   // sliding-window kernel
    const long long per_img = (long long)H * W;
    const int nchunk = (int)(per_img / 256 < 1 ? 1 : (per_img / 256 > 8192 ? 8192 : per_img / 256));
    if (dw3x3_strip_enabled() && pix >= 65536 && W >= 64 && pix * (ldi > ldo ? ldi : ldo) < (1ll << 31)) {     // measured: 163 vs 227 us at 352 x 512 x 360
      const int xs = dw3x3_strip_width(H, W, (C + 63) / 64, B);
      const int ns = ((H + 3) / 4) * ((W + xs - 1) / xs);
      const int nck = (ns + 3) / 4 > 65535 ? 65535 : (ns + 3) / 4;
      FFSR_LAUNCH(dw3x3_strip_kernel<false>, dim3((C + 63) / 64, nck, B), dim3(256), 0, ST, in, ldi, w, bias, out, ldo,
                         nullptr, H, W, C, nck, act, xs);
      return ffsr_launch_status();
    }
    FFSR_LAUNCH(dw3x3_run_kernel<false>, dim3((C + 63) / 64, nchunk, B), dim3(256), 0, ST, in, ldi, w, bias, out, ldo,
                       nullptr, H, W, C, nchunk, act);
    return ffsr_launch_status();
  }
  bool v4 = (C % 4 == 0) && (ldi % 4 == 0) && (ldo % 4 == 0) && al16(in) && al16(out) && al16(w);
  if (v4)
    FFSR_LAUNCH(dwconv_kernel<4>, dim3(grid_for(pix * (C / 4))), dim3(256), 0, ST, in, ldi, w, bias, out, ldo, B, H, W,
                       C, KH, KW, pad_h, pad_w, act);
  else
    FFSR_LAUNCH(dwconv_kernel<1>, dim3(grid_for(pix * C)), dim3(256), 0, ST, in, ldi, w, bias, out, ldo, B, H, W, C,
                       KH, KW, pad_h, pad_w, act);
  return ffsr_launch_status();
}

extern "C" int ffsr_dw3x3_gate_pool_f32(const float* in, int ldi, const float* w, const float* bias, float* out, int ldo,
                                        float* pooled, float* part, int B, int H, int W, int C, int nchunk, void* stream) {
  FFSR_CHECK(in && w && bias && out && pooled && part && B > 0 && H > 0 && W > 0 && C > 0 && nchunk > 0 && nchunk <= 65535);
  // measured (tools/dw_bench.py, strips of >= 8192 waves): wins on the widest level (64 channels: 506 vs 1065 us at 1408 x 2048,
  // 257 vs 288 us at 1024 x 1024); from 128 channels on the run kernel's 5 waves per SIMD x 36 loads in flight are as fast or
  // faster (352 x 512 x 256: 200 vs 325 us)
  if (dw3x3_strip_enabled() && C <= dw3x3_strip_maxc() && (long long)H * W >= 65536 && W >= 64 && (long long)B * H * W * (ldi > ldo ? ldi : ldo) < (1ll << 31))     // (any nchunk: a workgroup walks the strips chunk * 4 + wave, + 4 nchunk, ...)
    FFSR_LAUNCH(dw3x3_strip_kernel<true>, dim3((C + 63) / 64, nchunk, B), dim3(256), 0, ST, in, ldi, w, bias, out, ldo, part,
                       H, W, C, nchunk, 0, dw3x3_strip_width(H, W, (C + 63) / 64, B));
  else
    FFSR_LAUNCH(dw3x3_run_kernel<true>, dim3((C + 63) / 64, nchunk, B), dim3(256), 0, ST, in, ldi, w, bias, out, ldo, part,
                       H, W, C, nchunk, 0);
  FFSR_LAUNCH(colsum_finish_kernel, dim3((C + 63) / 64, B), dim3(1024), 0, ST, part, pooled, B, C, nchunk,
                     1.0f / (float)(H * W));
  return ffsr_launch_status();
}

extern "C" int ffsr_bilinear_f32(const float* in, int ldi, float* out, int ldo, int B, int Hi, int Wi, int Ho, int Wo, int C,
                                 float mul, int accumulate, void* stream) {
  FFSR_CHECK(in && out && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0);
  const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
  long long pix = (long long)B * Ho * Wo;
  bool v4 = (C % 4 == 0) && (ldi % 4 == 0) && (ldo % 4 == 0) && al16(in) && al16(out);
  if (v4)
    FFSR_LAUNCH(bilinear_kernel<4>, dim3(grid_for(pix * (C / 4))), dim3(256), 0, ST, in, ldi, out, ldo, B, Hi, Wi, Ho,
                       Wo, C, sh, sw, mul, accumulate);
  else
    FFSR_LAUNCH(bilinear_kernel<1>, dim3(grid_for(pix * C)), dim3(256), 0, ST, in, ldi, out, ldo, B, Hi, Wi, Ho, Wo, C,
                       sh, sw, mul, accumulate);
  return ffsr_launch_status();
}

extern "C" int ffsr_bicubic_up_f32(const float* in, int ldi, float* out, int ldo, int B, int H, int W, int C, int scale,
                                   void* stream) {
  FFSR_CHECK(in && out && B > 0 && H > 0 && W > 0 && C > 0 && scale > 0);
  FFSR_LAUNCH(bicubic_up_kernel, dim3(grid_for((long long)B * H * W * scale * scale * C)), dim3(256), 0, ST, in, ldi,
                     out, ldo, B, H, W, C, scale);
  return ffsr_launch_status();
}

extern "C" int ffsr_avgpool2_f32(const float* in, int ldi, float* out, int ldo, int B, int H, int W, int C, void* stream) {
  FFSR_CHECK(in && out && B > 0 && H > 1 && W > 1 && C > 0);
  long long pix = (long long)B * (H / 2) * (W / 2);
  bool v4 = (C % 4 == 0) && (ldi % 4 == 0) && (ldo % 4 == 0) && al16(in) && al16(out);
  if (v4)
    FFSR_LAUNCH(avgpool2_kernel<4>, dim3(grid_for(pix * (C / 4))), dim3(256), 0, ST, in, ldi, out, ldo, B, H, W, C);
  else
    FFSR_LAUNCH(avgpool2_kernel<1>, dim3(grid_for(pix * C)), dim3(256), 0, ST, in, ldi, out, ldo, B, H, W, C);
  return ffsr_launch_status();
}

extern "C" int ffsr_u8_to_f32(const unsigned char* in, float* out, int ldo, long long M, int C, void* stream) {
  FFSR_CHECK(in && out && M > 0 && C > 0 && ldo >= C);
  FFSR_LAUNCH(u8_to_f32_kernel, dim3(grid_for(M * C)), dim3(256), 0, ST, in, out, ldo, M, C);
  return ffsr_launch_status();
}

extern "C" int ffsr_f32_to_u8(const float* in, int ldi, unsigned char* out, long long M, int C, void* stream) {
  FFSR_CHECK(in && out && M > 0 && C > 0 && ldi >= C);
  FFSR_LAUNCH(f32_to_u8_kernel, dim3(grid_for(M * C)), dim3(256), 0, ST, in, ldi, out, M, C);
  return ffsr_launch_status();
}

extern "C" int ffsr_pad_reflect_f32(const float* in, int ldi, float* out, int ldo, int B, int H, int W, int Hp, int Wp, int C,
                                    void* stream) {
  FFSR_CHECK(in && out && B > 0 && H > 1 && W > 1 && Hp >= H && Wp >= W && Hp - H < H && Wp - W < W && C > 0);
  FFSR_LAUNCH(pad_reflect_kernel, dim3(grid_for((long long)B * Hp * Wp * C)), dim3(256), 0, ST, in, ldi, out, ldo, B, H,
                     W, Hp, Wp, C);
  return ffsr_launch_status();
}

extern "C" int ffsr_crop_f32(const float* in, int ldi, float* out, int ldo, int B, int H, int W, int Ho, int Wo, int C,
                             int do_clamp, void* stream) {
  FFSR_CHECK(in && out && B > 0 && Ho > 0 && Wo > 0 && Ho <= H && Wo <= W && C > 0);
  FFSR_LAUNCH(crop_kernel, dim3(grid_for((long long)B * Ho * Wo * C)), dim3(256), 0, ST, in, ldi, out, ldo, B, H, W, Ho,
                     Wo, C, do_clamp);
  return ffsr_launch_status();
}

extern "C" int ffsr_dihedral_f32(const float* in, int ldi, float* out, int ldo, int B, int Hi, int Wi, int Ho, int Wo, int C,
                                 int ay_i, int ay_j, int cy, int ax_i, int ax_j, int cx, float scale, int accumulate,
                                 void* stream) {
  FFSR_CHECK(in && out && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0);
  // the four corners of the output must map inside the input
  const int is[2] = {0, Ho - 1}, js[2] = {0, Wo - 1};
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b) {
      int y = ay_i * is[a] + ay_j * js[b] + cy, x = ax_i * is[a] + ax_j * js[b] + cx;
      FFSR_CHECK(y >= 0 && y < Hi && x >= 0 && x < Wi);
    }
  FFSR_LAUNCH(dihedral_kernel, dim3(grid_for((long long)B * Ho * Wo * C)), dim3(256), 0, ST, in, ldi, out, ldo, B, Hi,
                     Wi, Ho, Wo, C, ay_i, ay_j, cy, ax_i, ax_j, cx, scale, accumulate);
  return ffsr_launch_status();
}
