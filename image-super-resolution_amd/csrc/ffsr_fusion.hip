// Fused HR-resolution elementwise stages of the 7-phase fusion, gfx950.  All are HBM-bound:
// one pass over the expert stack instead of the reference's dozens of full-resolution temporaries.
#include "ffsr_common.h"

namespace {

struct Bil {
  int y0, y1, x0, x1;
  float ly, lx;
};
__device__ __forceinline__ void coord(int d, float scale, int n, int& i0, int& i1, float& l) {
  float s = ((float)d + 0.5f) * scale - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > n - 1) i0 = n - 1;
  i1 = i0 + (i0 < n - 1 ? 1 : 0);
  l = s - (float)i0;
}
__device__ __forceinline__ Bil make_bil(int y, int x, int h, int w, float sh, float sw) {
  Bil b;
  coord(y, sh, h, b.y0, b.y1, b.ly);
  coord(x, sw, w, b.x0, b.x1, b.lx);
  return b;
}
__device__ __forceinline__ float sample(const float* base, int ld, int w, const Bil& b, int c) {
  const float v00 = base[((size_t)b.y0 * w + b.x0) * ld + c], v01 = base[((size_t)b.y0 * w + b.x1) * ld + c];
  const float v10 = base[((size_t)b.y1 * w + b.x0) * ld + c], v11 = base[((size_t)b.y1 * w + b.x1) * ld + c];
  return (1.f - b.ly) * ((1.f - b.lx) * v00 + b.lx * v01) + b.ly * ((1.f - b.lx) * v10 + b.lx * v11);
}
__device__ __forceinline__ float gelu(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// gates = sigmoid(T * (raw - (0.7 - 0.5 * diff))) / clamp(sum + 1e-8, min 0.3)     enhanced_fusion_v2.py:462-465
__global__ void selector_gates_kernel(const float* __restrict__ raw, int ldr, const float* __restrict__ diff, int ldd,
                                      const float* __restrict__ temp, float* __restrict__ gates, int ldg, long long M) {
  long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const float thr = 0.7f - 0.5f * diff[m * ldd];
  const float T = temp[0];
  float g[4], s = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    g[e] = 1.0f / (1.0f + expf(-(T * (raw[m * ldr + e] - thr))));
    s += g[e];
  }
  s = fmaxf(s + 1e-8f, 0.3f);
#pragma unroll
  for (int e = 0; e < 4; ++e) gates[m * ldg + e] = g[e] / s;
}

// phase 4 tail.  t_lr = conv1x1(128->32)(lka_out) at LR (hoisted before the bilinear: both are linear);
// mod = sigmoid(W2 gelu(bilinear(t_lr)) + b2); out = clamp(img * (1 + 0.2 (mod - 0.5)), 0, 1)
__global__ __launch_bounds__(256) void modulate_kernel(   // (without the bound the compiler caps at 128 registers: 95 spilled)
    const float* __restrict__ t_lr, int ldt, const float* __restrict__ w2,
                                const float* __restrict__ b2, const float* __restrict__ img, int ldi, float* __restrict__ out,
                                int ldo, int B, int h, int w, int Hh, int Wh, float sh, float sw) {
  __shared__ float W2[96], B2[3];
  if (threadIdx.x < 96) W2[threadIdx.x] = w2[threadIdx.x];
  if (threadIdx.x < 3) B2[threadIdx.x] = b2[threadIdx.x];
  __syncthreads();
  long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= (long long)B * Hh * Wh) return;
  const int x = (int)(pix % Wh);
  long long t = pix / Wh;
  const int y = (int)(t % Hh), b = (int)(t / Hh);
  const Bil bl = make_bil(y, x, h, w, sh, sw);
  const float* base = t_lr + (size_t)b * h * w * ldt;
  const float* p00 = base + ((size_t)bl.y0 * w + bl.x0) * ldt;
  const float* p01 = base + ((size_t)bl.y0 * w + bl.x1) * ldt;
  const float* p10 = base + ((size_t)bl.y1 * w + bl.x0) * ldt;
  const float* p11 = base + ((size_t)bl.y1 * w + bl.x1) * ldt;
  const float hy = 1.f - bl.ly, hx = 1.f - bl.lx;
  float m0 = B2[0], m1 = B2[1], m2 = B2[2];
#pragma unroll
  for (int c4 = 0; c4 < 32; c4 += 4) {
    const floatx4 a = *reinterpret_cast<const floatx4*>(p00 + c4), bb = *reinterpret_cast<const floatx4*>(p01 + c4);
    const floatx4 cc = *reinterpret_cast<const floatx4*>(p10 + c4), d = *reinterpret_cast<const floatx4*>(p11 + c4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float u = gelu(hy * (hx * a[i] + bl.lx * bb[i]) + bl.ly * (hx * cc[i] + bl.lx * d[i]));
      m0 = fmaf(W2[c4 + i], u, m0);
      m1 = fmaf(W2[32 + c4 + i], u, m1);
      m2 = fmaf(W2[64 + c4 + i], u, m2);
    }
  }
  const float mm[3] = {m0, m1, m2};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float mod = 1.0f / (1.0f + expf(-mm[c]));
    const float v = img[pix * ldi + c] * (1.0f + 0.2f * (mod - 0.5f));
    out[pix * ldo + c] = fminf(fmaxf(v, 0.f), 1.f);
  }
}

// phases 5b + 5c + 6 (enhanced_fusion_v2.py:735-774).  fw = [W1 16x3 | b1 16 | W2 4x16 | b2 4] of freq_weight_conv.
// Improvement flags (io.py:186-193): fw == NULL (multi_resolution_fusion off) -> `hier` IS the fused image (simple_fusion's
// output), no frequency-guided part; gates == NULL (dynamic_expert_selection off) -> phase 6 is skipped.
__global__ void route_kernel(const float* __restrict__ enh, int lde, const float* __restrict__ hier, int ldh,
                             const float* __restrict__ routing, int ldr, const float* __restrict__ fw,
                             const float* __restrict__ gates, int ldg, const float* __restrict__ diff, int ldd,
                             float* __restrict__ out, int ldo, int B, int h, int w, int Hh, int Wh, float sh, float sw) {
  __shared__ float F[132];
  if (fw && threadIdx.x < 132) F[threadIdx.x] = fw[threadIdx.x];
  __syncthreads();
  long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= (long long)B * Hh * Wh) return;
  const int x = (int)(pix % Wh);
  long long t = pix / Wh;
  const int y = (int)(t % Hh), b = (int)(t / Hh);
  const Bil bl = make_bil(y, x, h, w, sh, sw);
  const size_t lrb = (size_t)b * h * w;
  // frequency-guided expert weights
  float wt[4] = {0.f, 0.f, 0.f, 0.f};
  if (fw) {
    float r[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) r[c] = sample(routing + lrb * ldr, ldr, w, bl, c);
    float lg[4] = {F[128], F[129], F[130], F[131]};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float hdn = gelu(F[48 + j] + F[j * 3] * r[0] + F[j * 3 + 1] * r[1] + F[j * 3 + 2] * r[2]);
#pragma unroll
      for (int e = 0; e < 4; ++e) lg[e] = fmaf(F[64 + e * 16 + j], hdn, lg[e]);
    }
    const float mx = fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3]));
    float ws = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      wt[e] = expf(lg[e] - mx);
      ws += wt[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) wt[e] /= ws;
  }
  float g[4] = {0.f, 0.f, 0.f, 0.f}, gs = 1.f, bw = 0.f;
  if (gates) {
    gs = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      g[e] = sample(gates + lrb * ldg, ldg, w, bl, e);
      gs += g[e];
    }
    gs += 1e-8f;
    bw = 0.3f + 0.4f * sample(diff + lrb * ldd, ldd, w, bl, 0);
  }
  const float* ep = enh + pix * lde;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float fq = 0.f, dy = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v = ep[3 * e + c];
      fq += v * wt[e];
      dy += v * g[e];
    }
    const float fused = fw ? hier[pix * ldh + c] * 0.7f + fq * 0.3f : hier[pix * ldh + c];
    out[pix * ldo + c] = gates ? (1.f - bw) * fused + bw * (dy / gs) : fused;
  }
  out[pix * ldo + 3] = 0.f;
}

// enhanced = clamp(sr + gate * strength * edge, 0, 1); out = clamp(enhanced + rscale * bilinear(lr), 0, 1)
// edge == NULL (edge_enhancement off): enhanced = sr
__global__ void edge_final_kernel(const float* __restrict__ sr, int lds, const float* __restrict__ edge, int lde,
                                  const float* __restrict__ gate, int ldg, const float* __restrict__ strength,
                                  const float* __restrict__ lr, int ldl, const float* __restrict__ rscale, float* __restrict__ out,
                                  int ldo, int B, int h, int w, int Hh, int Wh, float sh, float sw) {
  long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= (long long)B * Hh * Wh) return;
  const int x = (int)(pix % Wh);
  long long t = pix / Wh;
  const int y = (int)(t % Hh), b = (int)(t / Hh);
  const Bil bl = make_bil(y, x, h, w, sh, sw);
  const float gs = edge ? gate[pix * ldg] * strength[0] : 0.f;
  const float rs = rscale[0];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = sr[pix * lds + c];
    if (edge) {
      v += gs * edge[pix * lde + c];
      v = fminf(fmaxf(v, 0.f), 1.f);
    }
    v += rs * sample(lr + (size_t)b * h * w * ldl, ldl, w, bl, c);
    out[pix * ldo + c] = fminf(fmaxf(v, 0.f), 1.f);
  }
}

// SSIM map of calculate_ssim_torch (src/utils/metrics.py:129-186): inputs are the Gaussian-filtered moments
// mu1, mu2, E[x1^2], E[x2^2], E[x1 x2] (one channel each); out = ((2 mu1 mu2 + C1)(2 s12 + C2)) / ((mu1^2 + mu2^2 + C1)(s1 + s2 + C2))
__global__ void ssim_map_kernel(const float* __restrict__ mu1, const float* __restrict__ mu2, const float* __restrict__ e11,
                                const float* __restrict__ e22, const float* __restrict__ e12, int ld, float* __restrict__ out,
                                int ldo, long long M) {
  long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  const float a = mu1[m * ld], b = mu2[m * ld];
  const float a2 = a * a, b2 = b * b, ab = a * b;
  const float s1 = e11[m * ld] - a2, s2 = e22[m * ld] - b2, s12 = e12[m * ld] - ab;
  out[m * ldo] = ((2.f * ab + C1) * (2.f * s12 + C2)) / ((a2 + b2 + C1) * (s1 + s2 + C2));
}

inline int grid_for(long long n) { return (int)((n + 255) / 256); }

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int ffsr_selector_gates_f32(const float* raw, int ldr, const float* diff, int ldd, const float* temperature,
                                       float* gates, int ldg, long long M, void* stream) {
  FFSR_CHECK(raw && diff && temperature && gates && M > 0);
  FFSR_LAUNCH(selector_gates_kernel, dim3(grid_for(M)), dim3(256), 0, ST, raw, ldr, diff, ldd, temperature, gates, ldg, M);
  return ffsr_launch_status();
}

extern "C" int ffsr_modulate_f32(const float* t_lr, int ldt, const float* w2, const float* b2, const float* img, int ldi,
                                 float* out, int ldo, int B, int h, int w, int Hh, int Wh, void* stream) {
  FFSR_CHECK(t_lr && w2 && b2 && img && out && B > 0 && h > 0 && w > 0 && Hh > 0 && Wh > 0 && (ldt % 4) == 0 && ldt >= 32);
  FFSR_CHECK(((uintptr_t)t_lr & 15) == 0);
  FFSR_LAUNCH(modulate_kernel, dim3(grid_for((long long)B * Hh * Wh)), dim3(256), 0, ST, t_lr, ldt, w2, b2, img, ldi, out,
                     ldo, B, h, w, Hh, Wh, (float)h / (float)Hh, (float)w / (float)Wh);
  return ffsr_launch_status();
}

extern "C" int ffsr_fusion_route_f32(const float* enh, int lde, const float* hier, int ldh, const float* routing, int ldr,
                                     const float* fw, const float* gates, int ldg, const float* diff, int ldd, float* out,
                                     int ldo, int B, int h, int w, int Hh, int Wh, void* stream) {
  FFSR_CHECK(enh && hier && (routing || !fw) && (!gates == !diff) && out && B > 0 && h > 0 && w > 0 && Hh > 0 && Wh > 0 && ldo >= 4);
  FFSR_LAUNCH(route_kernel, dim3(grid_for((long long)B * Hh * Wh)), dim3(256), 0, ST, enh, lde, hier, ldh, routing, ldr,
                     fw, gates, ldg, diff, ldd, out, ldo, B, h, w, Hh, Wh, (float)h / (float)Hh, (float)w / (float)Wh);
  return ffsr_launch_status();
}

extern "C" int ffsr_edge_final_f32(const float* sr, int lds, const float* edge, int lde, const float* gate, int ldg,
                                   const float* strength, const float* lr, int ldl, const float* rscale, float* out, int ldo,
                                   int B, int h, int w, int Hh, int Wh, void* stream) {
  FFSR_CHECK(sr && (!edge || (gate && strength)) && lr && rscale && out && B > 0 && h > 0 && w > 0 && Hh > 0 && Wh > 0);
  FFSR_LAUNCH(edge_final_kernel, dim3(grid_for((long long)B * Hh * Wh)), dim3(256), 0, ST, sr, lds, edge, lde, gate, ldg,
                     strength, lr, ldl, rscale, out, ldo, B, h, w, Hh, Wh, (float)h / (float)Hh, (float)w / (float)Wh);
  return ffsr_launch_status();
}

extern "C" int ffsr_ssim_map_f32(const float* mu1, const float* mu2, const float* e11, const float* e22, const float* e12,
                                 int ld, float* out, int ldo, long long M, void* stream) {
  FFSR_CHECK(mu1 && mu2 && e11 && e22 && e12 && out && M > 0 && ld > 0 && ldo > 0);
  FFSR_LAUNCH(ssim_map_kernel, dim3(grid_for(M)), dim3(256), 0, ST, mu1, mu2, e11, e22, e12, ld, out, ldo, M);
  return ffsr_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------
// Thin 3x3 convolution: N <= 4 output channels (the RGB / gate heads at HR resolution: conv_last of the expert tails,
// refine.10, to_rgb.2, edge fusion.2 / edge_gate.2 / attn.2, difficulty_net.4).  On the GEMM kernels such a layer pads N to a
// 32-column MFMA tile (and triples it for the split-bf16 products): 10 - 30x more matrix work than the layer has, at
// 16 TFLOP/s "useful".  Here it is what it is, a streaming reduction: LPP = Cin / 4 lanes share a pixel (4 channels per lane,
// 16-byte loads, a group's loads cover the pixel's channel row contiguously), a group walks a horizontal run of pixels (the
// groups of a block take vertically adjacent rows, so the 3x row overlap is served by the CU's own cache) with a
// sliding 3x3 register window (3 new loads per pixel), the weights of the lane's 4 channels live in registers, exact fp32 FMA,
// the LPP partial sums are combined by DPP adds and lanes 0 .. N-1 of the group store one output each.
//   out[pix, n] = res[pix, n] * rscale + act(sum_{tap, c} in[pix + tap, c] * w[n, tap, c] + bias[n]) * cscale
// Sum over an aligned group of LPP lanes (every lane gets it): DPP adds inside a 16-lane row (quad swaps, then the half-row and
// the row mirrored onto themselves), one LDS-crossbar shuffle for the 32-lane case.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int LPP>
__device__ __forceinline__ float group_sum(float v) {
  if (LPP >= 2) v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
  if (LPP >= 4) v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
  if (LPP >= 8) v = dpp_add<0x141>(v);   // row_half_mirror
  if (LPP >= 16) v = dpp_add<0x140>(v);  // row_mirror
  if (LPP >= 32) v += __shfl_xor(v, 16, 64);
  return v;
}

template <int N, int LPP>
__global__ __launch_bounds__(256) void conv3x3_thin_kernel(const float* __restrict__ in, int ldi, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ out, int ldo,
                                                           const float* __restrict__ res, int ldr, int B, int H, int W, int run,
                                                           int act, float slope, float cscale, float rscale) {
  constexpr int CIN = 4 * LPP, GPB = 256 / LPP;              // groups (pixel runs) per block
  const int g = threadIdx.x / LPP, l = threadIdx.x % LPP;
  // a block = GPB vertically adjacent rows x one run of columns: its groups read GPB + 2 input rows between them, at the same
  // time, on one CU (groups of other blocks -- other XCDs, other L2s -- share only the two halo rows)
  const int runs_per_row = (W + run - 1) / run, nyb = (H + GPB - 1) / GPB;
  const int xr = (int)(blockIdx.x % runs_per_row);
  const int yb = (int)((blockIdx.x / runs_per_row) % nyb), b = (int)(blockIdx.x / runs_per_row / nyb);
  const bool live = yb * GPB + g < H;                        // (idle groups still take part in the lane exchanges)
  const int y = min(yb * GPB + g, H - 1);
  const int x0 = xr * run, x1 = min(W, x0 + run);
  // rows outside the image: the row index is clamped (the loads stay unconditional) and the lane's weights of that kernel row
  // are zero; columns outside: the column is clamped and the loaded vector multiplied by 0.  All addresses are 32-bit byte
  // offsets from `in` (the entry point checks the map is below 4 GiB).
  floatx4 wt[N][9];
  unsigned roff[3];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int yy = y + ky - 1;
    const float rowm = (yy >= 0 && yy < H) ? 1.f : 0.f;
    roff[ky] = (unsigned)(((b * H + min(max(yy, 0), H - 1)) * W) * ldi + 4 * l) * 4u;
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
        wt[n][ky * 3 + kx] = *reinterpret_cast<const floatx4*>(w + ((size_t)n * 9 + ky * 3 + kx) * CIN + 4 * l) * rowm;
  }
  const char* inb = reinterpret_cast<const char*>(in);
  const unsigned ldb = (unsigned)ldi * 4u, cmax = (unsigned)(W - 1) * ldb;
  auto col = [&](int ky, unsigned coff, float m) -> floatx4 {
    return *reinterpret_cast<const floatx4*>(inb + (size_t)(roff[ky] + coff)) * m;
  };
  // ring of R columns x 3 rows, the loop unrolled R times so that every index is static (no register moves): at pixel x the
  // window is ring[j], ring[j+1], ring[j+2] (columns x-1, x, x+1), columns x+2 .. x+R-3 are in flight and column x+R-2 is
  // issued into the slot column x-2 just left.
  constexpr int R = N == 4 ? 6 : 8;
  floatx4 ring[R][3];
#pragma unroll
  for (int j = 0; j < R - 1; ++j) {
    const int c = x0 - 1 + j;
    const unsigned coff = (unsigned)min(max(c, 0), W - 1) * ldb;
    const float m = (c >= 0 && c < W) ? 1.f : 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) ring[j][ky] = col(ky, coff, m);
  }
  unsigned coff = (unsigned)min(x0 + R - 2, W - 1) * ldb;   // column x + R - 2 of the pixel x about to be computed
  size_t pix = ((size_t)b * H + y) * W + x0;
  const float bl = (bias && l < N) ? bias[l] : 0.f;
  for (int xb = x0; xb < x1; xb += R) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int x = xb + j;
      const float m = (x + R - 2 < W) ? 1.f : 0.f;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) ring[(j + R - 1) % R][ky] = col(ky, coff, m);
      coff = min(coff + ldb, cmax);
      float acc[N];
#pragma unroll
      for (int n = 0; n < N; ++n) {
        floatx4 a = ring[j % R][0] * wt[n][0];
#pragma unroll
        for (int t = 1; t < 9; ++t) a += ring[(j + t % 3) % R][t / 3] * wt[n][t];
        acc[n] = group_sum<LPP>((a[0] + a[1]) + (a[2] + a[3]));
      }
      if (live && l < N && x < x1) {
        float v = acc[0];
#pragma unroll
        for (int n = 1; n < N; ++n) v = (l == n) ? acc[n] : v;
        v = ffsr_act(v + bl, act, slope) * cscale;
        if (res) v += res[pix * ldr + l] * rscale;
        out[pix * ldo + l] = v;
      }
      ++pix;
    }
  }
}

template <int N>
static int launch_thin(const float* in, int ldi, const float* w, const float* bias, float* out, int ldo, const float* res, int ldr,
                       int B, int H, int W, int Cin, int act, float slope, float cscale, float rscale, hipStream_t st) {
  const int run = W >= 64 ? 32 : (W >= 16 ? 8 : W);
  const long long runs_per_row = (W + run - 1) / run;
#define FFSR_THIN(LPP)                                                                                                      \
  FFSR_LAUNCH((conv3x3_thin_kernel<N, LPP>), dim3((unsigned)(runs_per_row * ((H + 256 / LPP - 1) / (256 / LPP)) * B)), dim3(256), 0, st, in, ldi, w, \
              bias, out, ldo, res, ldr, B, H, W, run, act, slope, cscale, rscale)
  switch (Cin) {
    case 8: FFSR_THIN(2); break;
    case 16: FFSR_THIN(4); break;
    case 32: FFSR_THIN(8); break;
    case 64: FFSR_THIN(16); break;
    case 128: FFSR_THIN(32); break;
    default: return FFSR_EINVAL;
  }
#undef FFSR_THIN
  return ffsr_launch_status();
}

extern "C" int ffsr_conv3x3_thin_f32(const float* in, int ldi, const float* wgt, const float* bias, float* out, int ldo,
                                     const float* res, int ldr, int B, int H, int W, int Cin, int N, int act, float slope,
                                     float cscale, float rscale, void* stream) {
  FFSR_CHECK(in && wgt && out && B > 0 && H > 0 && W > 0 && N >= 1 && N <= 4 && ldo >= N && (!res || ldr >= N));
  FFSR_CHECK((Cin == 8 || Cin == 16 || Cin == 32 || Cin == 64 || Cin == 128) && ldi >= Cin && (ldi & 3) == 0 &&
             ((uintptr_t)in & 15) == 0 && ((uintptr_t)wgt & 15) == 0);
  FFSR_CHECK((long long)B * H * W * ldi * 4 < (1ll << 32) && N <= Cin / 4);   // 32-bit byte offsets; lanes 0 .. N-1 of a pixel's Cin/4 lanes store
  hipStream_t st = (hipStream_t)stream;
  switch (N) {
    case 1: return launch_thin<1>(in, ldi, wgt, bias, out, ldo, res, ldr, B, H, W, Cin, act, slope, cscale, rscale, st);
    case 2: return launch_thin<2>(in, ldi, wgt, bias, out, ldo, res, ldr, B, H, W, Cin, act, slope, cscale, rscale, st);
    case 3: return launch_thin<3>(in, ldi, wgt, bias, out, ldo, res, ldr, B, H, W, Cin, act, slope, cscale, rscale, st);
    default: return launch_thin<4>(in, ldi, wgt, bias, out, ldo, res, ldr, B, H, W, Cin, act, slope, cscale, rscale, st);
  }
}
