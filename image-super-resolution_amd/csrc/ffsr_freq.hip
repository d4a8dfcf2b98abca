// Phase-2 frequency decomposition of the fusion net, gfx950.  All three transforms act on the
// 3-channel LR image (tiny, launch-bound), and write straight into the band tensor bands[pixel][9 bands][4]
// (channels padded 3 -> 4, pad = 0) that the cross-band attention consumes as token rows.
//   bands 0-2: 8x8 block DCT-II, zig-zag thirds              (multi_domain_frequency.py:146-196)
//   bands 3-6: one-level db4 DWT subbands LL, LH, HL, HH      (:251-299; bilinear upsample done by ffsr_bilinear_f32)
//   bands 7-8: rfft2 low / high split with a learned soft mask (:352-385), as separable dense DFTs
#include "ffsr_common.h"

namespace {

__device__ __forceinline__ int reflect_idx(int i, int n) {  // torch 'reflect' padding index
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

// ---------------------------------------------------------------------------------------------- DCT
// one wave per (8x8 block, channel): lane = (row, col) of the block; 4 waves per workgroup, each with its own LDS.
// D: [8][8] DCT basis, masks: [3][64], scale[3].
__global__ __launch_bounds__(256) void dct_bands_kernel(const float* __restrict__ img, int ldi, const float* __restrict__ D,
                                                        const float* __restrict__ masks, const float* __restrict__ scale,
                                                        float* __restrict__ bands, int ldb, int B, int H, int W, int nbh,
                                                        int nbw) {
  __shared__ float Ds[64], Ms[3][64];
  __shared__ float Xs[4][64], Ts[4][64], Ys[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (threadIdx.x < 64) {
    Ds[lane] = D[lane];
    Ms[0][lane] = masks[lane];
    Ms[1][lane] = masks[64 + lane];
    Ms[2][lane] = masks[128 + lane];
  }
  const int total = B * nbh * nbw * 3;
  int idx = blockIdx.x * 4 + wv;
  const bool live = idx < total;
  if (!live) idx = total - 1;
  const int c = idx % 3;
  int t = idx / 3;
  const int bx = t % nbw;
  t /= nbw;
  const int by = t % nbh, b = t / nbh;
  const int i = lane >> 3, j = lane & 7;
  {
    int y = reflect_idx(by * 8 + i, H), x = reflect_idx(bx * 8 + j, W);
    Xs[wv][lane] = img[(((size_t)b * H + y) * W + x) * ldi + c];
  }
  __syncthreads();
  float s = 0.f;  // T[i][k=j] = sum_m X[i][m] D[k][m]
#pragma unroll
  for (int m = 0; m < 8; ++m) s = fmaf(Xs[wv][i * 8 + m], Ds[j * 8 + m], s);
  Ts[wv][lane] = s;
  __syncthreads();
  s = 0.f;  // Y[k=i][l=j] = sum_m D[k][m] T[m][l]
#pragma unroll
  for (int m = 0; m < 8; ++m) s = fmaf(Ds[i * 8 + m], Ts[wv][m * 8 + j], s);
  Ys[wv][lane] = s;
  __syncthreads();
  for (int band = 0; band < 3; ++band) {
    s = 0.f;  // U[k=i][j] = sum_l (mask o Y)[k][l] D[l][j]
#pragma unroll
    for (int l = 0; l < 8; ++l) s = fmaf(Ys[wv][i * 8 + l] * Ms[band][i * 8 + l], Ds[l * 8 + j], s);
    __syncthreads();  // previous band's readers of Ts are done
    Ts[wv][lane] = s;
    __syncthreads();
    s = 0.f;  // S[i][j] = sum_k D[k][i] U[k][j]
#pragma unroll
    for (int k = 0; k < 8; ++k) s = fmaf(Ds[k * 8 + i], Ts[wv][k * 8 + j], s);
    const int y = by * 8 + i, x = bx * 8 + j;
    if (live && y < H && x < W) bands[(((size_t)b * H + y) * W + x) * ldb + band * 4 + c] = s * scale[band];
  }
}

// ---------------------------------------------------------------------------------------------- DWT (db4, 1 level)
// sub[b, i, j, s*4 + c], s = LL, LH, HL, HH ; Hd = (H+6)/2+1, Wd = (W+6)/2+1 ; filters lo/hi: 8 taps (correlation)
__global__ void dwt_kernel(const float* __restrict__ img, int ldi, const float* __restrict__ lo, const float* __restrict__ hi,
                           float* __restrict__ sub, int B, int H, int W, int Hd, int Wd) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * Hd * Wd * 3) return;
  const int c = idx % 3;
  int t = idx / 3;
  const int j = t % Wd;
  t /= Wd;
  const int i = t % Hd, b = t / Hd;
  float f_lo[8], f_hi[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    f_lo[k] = lo[k];
    f_hi[k] = hi[k];
  }
  float ll = 0.f, lh = 0.f, hl = 0.f, hh = 0.f;
  for (int ty = 0; ty < 8; ++ty) {
    const int y = reflect_idx(2 * i + ty - 7, H);
    float rlo = 0.f, rhi = 0.f;  // row-filtered values at (y, j)
#pragma unroll
    for (int tx = 0; tx < 8; ++tx) {
      const int x = reflect_idx(2 * j + tx - 7, W);
      const float v = img[(((size_t)b * H + y) * W + x) * ldi + c];
      rlo = fmaf(v, f_lo[tx], rlo);
      rhi = fmaf(v, f_hi[tx], rhi);
    }
    ll = fmaf(rlo, f_lo[ty], ll);
    lh = fmaf(rlo, f_hi[ty], lh);
    hl = fmaf(rhi, f_lo[ty], hl);
    hh = fmaf(rhi, f_hi[ty], hh);
  }
  float* o = sub + (((size_t)b * Hd + i) * Wd + j) * 16 + c;
  o[0] = ll;
  o[4] = lh;
  o[8] = hl;
  o[12] = hh;
}

// ---------------------------------------------------------------------------------------------- DFT split
// twiddle tables: tw[j] = (cos(2 pi j / n), sin(2 pi j / n)), computed on the host in double precision.
// K1: rows  Zr[b,c,y,k] = sum_x img[b,y,x,c] e^{-2 pi i k x / W}            k < Wf = W/2+1
__global__ void dft_rows_kernel(const float* __restrict__ img, int ldi, const float2* __restrict__ twW, float2* __restrict__ Zr,
                                int B, int H, int W, int Wf) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * 3 * H * Wf) return;
  const int k = idx % Wf;
  int t = idx / Wf;
  const int y = t % H;
  t /= H;
  const int c = t % 3, b = t / 3;
  const float* row = img + ((size_t)b * H + y) * W * ldi + c;
  float re = 0.f, im = 0.f;
  int j = 0;
  for (int x = 0; x < W; ++x) {
    const float v = row[(size_t)x * ldi];
    const float2 w = twW[j];
    re = fmaf(v, w.x, re);
    im = fmaf(-v, w.y, im);
    j += k;
    if (j >= W) j -= W;
  }
  Zr[idx] = make_float2(re, im);
}
// K2: columns + ortho scale + mask split.  Zlo/Zhi[b,c,ky,k]
__global__ void dft_cols_mask_kernel(const float2* __restrict__ Zr, const float2* __restrict__ twH,
                                     const float* __restrict__ mask, float2* __restrict__ Zlo, float2* __restrict__ Zhi, int BC,
                                     int H, int Wf, float norm) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= BC * H * Wf) return;
  const int k = idx % Wf;
  int t = idx / Wf;
  const int ky = t % H, bc = t / H;
  const float2* col = Zr + (size_t)bc * H * Wf + k;
  float re = 0.f, im = 0.f;
  int j = 0;
  for (int y = 0; y < H; ++y) {
    const float2 z = col[(size_t)y * Wf];
    const float2 w = twH[j];  // e^{-i th} = (cos, -sin)
    re += z.x * w.x + z.y * w.y;
    im += z.y * w.x - z.x * w.y;
    j += ky;
    if (j >= H) j -= H;
  }
  re *= norm;
  im *= norm;
  const float m = mask ? mask[ky * Wf + k] : 1.f;
  Zlo[idx] = make_float2(re * m, im * m);
  if (Zhi) Zhi[idx] = make_float2(re * (1.f - m), im * (1.f - m));
}
// K3: inverse columns  U[s,b,c,y,k] = sum_ky Z[s,b,c,ky,k] e^{+2 pi i ky y / H}
__global__ void idft_cols_kernel(const float2* __restrict__ Z, const float2* __restrict__ twH, float2* __restrict__ U, int SBC,
                                 int H, int Wf) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= SBC * H * Wf) return;
  const int k = idx % Wf;
  int t = idx / Wf;
  const int y = t % H, sbc = t / H;
  const float2* col = Z + (size_t)sbc * H * Wf + k;
  float re = 0.f, im = 0.f;
  int j = 0;
  for (int ky = 0; ky < H; ++ky) {
    const float2 z = col[(size_t)ky * Wf];
    const float2 w = twH[j];
    re += z.x * w.x - z.y * w.y;
    im += z.x * w.y + z.y * w.x;
    j += y;
    if (j >= H) j -= H;
  }
  U[idx] = make_float2(re, im);
}
// K4: inverse rows, complex -> real (imaginary part of the DC and Nyquist bins ignored, like pocketfft's c2r)
//     bands[pix][7 + s][c] = scale[s] * norm * sum_k w_k Re(U e^{+2 pi i k x / W})
__global__ void idft_rows_kernel(const float2* __restrict__ U, const float2* __restrict__ twW, const float* __restrict__ scale,
                                 float* __restrict__ bands, int ldb, int B, int H, int W, int Wf, float norm) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 2 * B * 3 * H * W) return;
  const int x = idx % W;
  int t = idx / W;
  const int y = t % H;
  t /= H;
  const int c = t % 3;
  t /= 3;
  const int b = t % B, s = t / B;
  const float2* row = U + ((((size_t)s * B + b) * 3 + c) * H + y) * Wf;
  float acc = row[0].x;
  int j = 0;
  const int kend = (W & 1) ? Wf : Wf - 1;  // bins with weight 2
  for (int k = 1; k < kend; ++k) {
    j += x;
    if (j >= W) j -= W;
    const float2 z = row[k];
    const float2 w = twW[j];
    acc += 2.f * (z.x * w.x - z.y * w.y);
  }
  if (!(W & 1)) acc += row[Wf - 1].x * ((x & 1) ? -1.f : 1.f);
  bands[(((size_t)b * H + y) * W + x) * ldb + (7 + s) * 4 + c] = acc * norm * scale[s];
}

inline int grid_for(long long n) { return (int)((n + 255) / 256); }

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int ffsr_dct_bands_f32(const float* img, int ldi, const float* D, const float* masks, const float* scale,
                                  float* bands, int ldb, int B, int H, int W, void* stream) {
  FFSR_CHECK(img && D && masks && scale && bands && B > 0 && H >= 8 && W >= 8 && ldb >= 36);
  const int nbh = (H + 7) / 8, nbw = (W + 7) / 8;
  FFSR_LAUNCH(dct_bands_kernel, dim3((B * nbh * nbw * 3 + 3) / 4), dim3(256), 0, ST, img, ldi, D, masks, scale, bands,
                     ldb, B, H, W, nbh, nbw);
  return ffsr_launch_status();
}

extern "C" int ffsr_dwt_db4_f32(const float* img, int ldi, const float* lo, const float* hi, float* sub, int B, int H, int W,
                                void* stream) {
  FFSR_CHECK(img && lo && hi && sub && B > 0 && H >= 8 && W >= 8);
  const int Hd = (H + 6) / 2 + 1, Wd = (W + 6) / 2 + 1;
  FFSR_LAUNCH(dwt_kernel, dim3(grid_for((long long)B * Hd * Wd * 3)), dim3(256), 0, ST, img, ldi, lo, hi, sub, B, H, W,
                     Hd, Wd);
  return ffsr_launch_status();
}

// work: float2 scratch of 5 * B*3*H*Wf elements
extern "C" int ffsr_fft_bands_f32(const float* img, int ldi, const float* twW, const float* twH, const float* mask,
                                  const float* scale, float* work, float* bands, int ldb, int B, int H, int W, void* stream) {
  FFSR_CHECK(img && twW && twH && mask && scale && work && bands && B > 0 && H > 1 && W > 1 && ldb >= 36);
  const int Wf = W / 2 + 1;
  const size_t n = (size_t)B * 3 * H * Wf;
  float2* Zr = reinterpret_cast<float2*>(work);
  float2* Zs = Zr + n;       // [2][B*3*H*Wf]  (lo, hi)
  float2* U = Zs + 2 * n;    // [2][B*3*H*Wf]
  const float norm = 1.0f / sqrtf((float)H * (float)W);
  FFSR_LAUNCH(dft_rows_kernel, dim3(grid_for(n)), dim3(256), 0, ST, img, ldi, (const float2*)twW, Zr, B, H, W, Wf);
  FFSR_LAUNCH(dft_cols_mask_kernel, dim3(grid_for(n)), dim3(256), 0, ST, Zr, (const float2*)twH, mask, Zs, Zs + n, B * 3,
                     H, Wf, norm);
  FFSR_LAUNCH(idft_cols_kernel, dim3(grid_for(2 * n)), dim3(256), 0, ST, Zs, (const float2*)twH, U, 2 * B * 3, H, Wf);
  FFSR_LAUNCH(idft_rows_kernel, dim3(grid_for((long long)2 * B * 3 * H * W)), dim3(256), 0, ST, U, (const float2*)twW,
                     scale, bands, ldb, B, H, W, Wf, norm);
  return ffsr_launch_status();
}

// torch.fft.rfft2(img, norm="ortho") of a 3-channel map: spec [B*3, H, W/2+1] complex (interleaved re, im).  Used by the
// backward pass of the FFT band split (the adjoint of irfft2 is a weighted rfft2: see ffsr_fft_mask_grad_f32).
// work: float2 scratch of B*3*H*(W/2+1) elements.
extern "C" int ffsr_rfft2_ortho_f32(const float* img, int ldi, const float* twW, const float* twH, float* work, float* spec,
                                    int B, int H, int W, void* stream) {
  FFSR_CHECK(img && twW && twH && work && spec && B > 0 && H > 1 && W > 1 && ldi >= 3);
  const int Wf = W / 2 + 1;
  const size_t n = (size_t)B * 3 * H * Wf;
  const float norm = 1.0f / sqrtf((float)H * (float)W);
  FFSR_LAUNCH(dft_rows_kernel, dim3(grid_for(n)), dim3(256), 0, ST, img, ldi, (const float2*)twW, (float2*)work, B, H, W, Wf);
  FFSR_LAUNCH(dft_cols_mask_kernel, dim3(grid_for(n)), dim3(256), 0, ST, (const float2*)work, (const float2*)twH,
              (const float*)nullptr, (float2*)spec, (float2*)nullptr, B * 3, H, Wf, norm);
  return ffsr_launch_status();
}
