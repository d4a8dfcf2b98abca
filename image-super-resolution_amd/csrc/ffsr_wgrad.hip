// Weight gradient of a stride-1 convolution / linear layer on the f32 MFMA (gfx950), for the fusion network's training
// step (autograd of nn.Conv2d / nn.Linear under loss.backward(), train.py:336):
//
//   dW[n, c, ky, kx] += sum_pix dY[pix, n] * X[pix + (ky - ph, kx - pw), c]          (zero padding)
//
// The contraction runs over PIXELS, the slow dimension of both channels-last operands, so one MFMA step takes its two
// pixels as k and 32 consecutive channels as the row / column index: v_mfma_f32_32x32x2_f32 wants, per lane, one element
// A[m = lane & 31][k = lane >> 5] and B[k][n = lane & 31] -- i.e. 32 consecutive channels of pixel k.  Both fragments
// are therefore plain ds_read_b32 of consecutive floats of a staged [pixel][channel] tile: no transposes anywhere
// (a bf16 MFMA would need k = 8 consecutive pixels per lane, a transpose of both operands).  Products are exact fp32.
//
// grid = (row tiles x column tiles, taps, pixel splits); one workgroup = 4 waves arranged WN x WC x WK over a
// (32 FN WN) x (32 FC WC) tile of (n, c): waves along WK take alternate pixel pairs of a staged chunk and are summed
// through LDS at the end.  Chunks of 16 pixels are double buffered (global -> registers -> LDS, one barrier per chunk);
// per-pixel validity of the shifted tap (image borders, end of the split) is computed two chunks ahead by 16 threads.
// Every split writes its partial tile; a second kernel sums the splits in double precision and ACCUMULATES into dW in
// nn.Conv2d layout [N, Cin, KH, KW] (deterministic, no atomics).
#include "ffsr_common.h"

namespace {

constexpr int KC = 16;   // pixels per staged chunk

struct WgradArgs {
  const float* x;
  const float* dy;
  float* part;
  int ldx, ldy;
  int B, H, W, Cin, N, KH, KW, ph, pw;
  long long P, per_split;
  int n_tiles, c_tiles;
};

template <int WN, int WC, int FN, int FC>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs p) {
  constexpr int WK = 4 / (WN * WC);
  constexpr int TN = 32 * FN * WN, TC = 32 * FC * WC;
  constexpr int SA = TN + 32, SB = TC + 32;                 // row strides: the two lane halves land on different banks
  constexpr int STAGE = KC * (SA + SB);
  constexpr int RED = (WK > 1) ? WK * TN * TC : 0;
  constexpr int LDS_FLOATS = (2 * STAGE > RED) ? 2 * STAGE : RED;
  __shared__ float lds[LDS_FLOATS];
  __shared__ int meta[2][2][KC];                            // [slot][0: dY row valid, 1: X row valid][pixel of the chunk]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wk = wave % WK, wc = (wave / WK) % WC, wn = wave / (WK * WC);
  const int tile = blockIdx.x;
  const int n0 = (tile / p.c_tiles) * TN, c0 = (tile % p.c_tiles) * TC;
  const int tap = blockIdx.y;
  const int dyo = tap / p.KW - p.ph, dxo = tap % p.KW - p.pw;
  const long long shift = (long long)dyo * p.W + dxo;
  const long long p_begin = (long long)blockIdx.z * p.per_split;
  const long long p_end = (p_begin + p.per_split < p.P) ? p_begin + p.per_split : p.P;
  const int nchunks = (int)((p_end - p_begin + KC - 1) / KC);

  auto make_meta = [&](int chunk) {         // threads 0 .. KC-1
    const long long q = p_begin + (long long)chunk * KC + tid;
    int va = 0, vb = 0;
    if (q < p_end) {
      va = 1;
      const int xx = (int)(q % p.W);
      const int yy = (int)((q / p.W) % p.H);
      const int sy = yy + dyo, sx = xx + dxo;
      vb = (sy >= 0 && sy < p.H && sx >= 0 && sx < p.W) ? 1 : 0;
    }
    meta[chunk & 1][0][tid] = va;
    meta[chunk & 1][1][tid] = vb;
  };

  constexpr int LA = KC * TN / 256, LB = KC * TC / 256;     // elements per thread and chunk
  float ra[LA], rb[LB];
  auto fetch = [&](int chunk) {
    const long long q0 = p_begin + (long long)chunk * KC;
    const int* mv = &meta[chunk & 1][0][0];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int e = tid + i * 256;
      const int k = e / TN, col = e % TN;
      const int n = n0 + col;
      ra[i] = (mv[k] && n < p.N) ? p.dy[(q0 + k) * p.ldy + n] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      const int e = tid + i * 256;
      const int k = e / TC, col = e % TC;
      const int c = c0 + col;
      rb[i] = (mv[KC + k] && c < p.Cin) ? p.x[(q0 + k + shift) * p.ldx + c] : 0.f;
    }
  };
  auto stash = [&](int buf) {
    float* As = lds + buf * STAGE;
    float* Bs = As + KC * SA;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int e = tid + i * 256;
      As[(e / TN) * SA + e % TN] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      const int e = tid + i * 256;
      Bs[(e / TC) * SB + e % TC] = rb[i];
    }
  };

  floatx16 acc[FN][FC];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FC; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (nchunks > 0) {
    if (tid < KC) {
      make_meta(0);
      if (nchunks > 1) make_meta(1);
    }
    __syncthreads();
    fetch(0);
    stash(0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
      const int buf = ch & 1;
      if (ch + 1 < nchunks) fetch(ch + 1);                   // meta of chunk ch+1 was published one barrier ago
      const float* As = lds + buf * STAGE;
      const float* Bs = As + KC * SA;
#pragma unroll
      for (int kp = wk; kp < KC / 2; kp += WK) {
        float a[FN], b[FC];
#pragma unroll
        for (int i = 0; i < FN; ++i) a[i] = As[(2 * kp + h) * SA + (wn * FN + i) * 32 + r];
#pragma unroll
        for (int j = 0; j < FC; ++j) b[j] = Bs[(2 * kp + h) * SB + (wc * FC + j) * 32 + r];
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
          for (int j = 0; j < FC; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      if (ch + 1 < nchunks) stash(buf ^ 1);                  // the other buffer was last read before the previous barrier
      if (ch + 2 < nchunks && tid < KC) make_meta(ch + 2);   // slot of chunk ch: its readers (fetch(ch)) are done
      __syncthreads();
    }
  }

  // ---- sum the WK pixel-interleaved waves through LDS, then store the partial tile
  if constexpr (WK > 1) {
    float* red = lds;                                        // all stage reads finished at the last barrier
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int j = 0; j < FC; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (wn * FN + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          const int col = (wc * FC + j) * 32 + r;
          red[(wk * TN + row) * TC + col] = acc[i][j][e];
        }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FC; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = (wn * FN + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            const int col = (wc * FC + j) * 32 + r;
            float s = acc[i][j][e];
#pragma unroll
            for (int q = 1; q < WK; ++q) s += red[(q * TN + row) * TC + col];
            acc[i][j][e] = s;
          }
    }
  }
  if (wk == 0) {
    const int T = p.KH * p.KW;
    float* dst = p.part + ((size_t)blockIdx.z * T + tap) * p.N * p.Cin;
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int j = 0; j < FC; ++j) {
        const int c = c0 + (wc * FC + j) * 32 + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int n = n0 + (wn * FN + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (n < p.N && c < p.Cin) dst[(size_t)n * p.Cin + c] = acc[i][j][e];
        }
      }
  }
}

// dW[n, c, t] += sum_s part[s, t, n, c]
__global__ void wgrad_finish_kernel(const float* __restrict__ part, int S, int T, int N, int Cin, float* __restrict__ dw) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long per = (long long)N * Cin;
  if (idx >= per * T) return;
  const int t = (int)(idx / per);
  const long long nc = idx % per;
  double s = 0.0;
  for (int k = 0; k < S; ++k) s += part[((size_t)k * T + t) * per + nc];
  dw[nc * T + t] += (float)s;
}

template <int WN, int WC, int FN, int FC>
void launch(const WgradArgs& a, dim3 grid, hipStream_t st) {
  FFSR_LAUNCH((conv_wgrad_kernel<WN, WC, FN, FC>), grid, dim3(256), 0, st, a);
}

}  // namespace

// x [B,H,W,ldx] (Cin channels), dy [B,H,W,ldy] (N channels): dw [N, Cin, KH, KW] += the weight gradient of the stride-1
// convolution with zero padding (pad_h, pad_w) (a linear layer: B = H = 1, W = rows, KH = KW = 1).
// partial: caller-owned scratch of partial_floats floats (>= KH*KW*N*Cin; more lets the pixels be split over more
// workgroups).
extern "C" int ffsr_conv_wgrad_f32(const float* x, int ldx, const float* dy, int ldy, float* dw, float* partial,
                                   long long partial_floats, int B, int H, int W, int Cin, int N, int KH, int KW, int pad_h,
                                   int pad_w, void* stream) {
  FFSR_CHECK(x && dy && dw && partial && B > 0 && H > 0 && W > 0 && Cin > 0 && N > 0 && KH > 0 && KW > 0 && ldx >= Cin && ldy >= N);
  FFSR_CHECK(pad_h >= 0 && pad_w >= 0 && pad_h < KH && pad_w < KW);
  const int T = KH * KW;
  const long long per_tile = (long long)T * N * Cin;
  FFSR_CHECK(partial_floats >= per_tile);
  WgradArgs a;
  a.x = x, a.dy = dy, a.part = partial, a.ldx = ldx, a.ldy = ldy;
  a.B = B, a.H = H, a.W = W, a.Cin = Cin, a.N = N, a.KH = KH, a.KW = KW, a.ph = pad_h, a.pw = pad_w;
  a.P = (long long)B * H * W;
  const int tn = N <= 32 ? 32 : (N <= 64 ? 64 : 128), tc = Cin <= 32 ? 32 : (Cin <= 64 ? 64 : 128);
  a.n_tiles = (N + tn - 1) / tn, a.c_tiles = (Cin + tc - 1) / tc;
  const long long tiles = (long long)a.n_tiles * a.c_tiles * T;
  // splits: enough workgroups to fill the chip a few times, at least 256 pixels each, bounded by the scratch
  long long S = (2048 + tiles - 1) / tiles;
  const long long max_by_px = (a.P + 255) / 256;
  if (S > max_by_px) S = max_by_px;
  if (S > partial_floats / per_tile) S = partial_floats / per_tile;
  if (S > 65535) S = 65535;
  if (S < 1) S = 1;
  a.per_split = ((a.P + S - 1) / S + KC - 1) / KC * KC;
  S = (a.P + a.per_split - 1) / a.per_split;
  FFSR_CHECK(T <= 65535);
  const dim3 grid((unsigned)(a.n_tiles * a.c_tiles), (unsigned)T, (unsigned)S);
  hipStream_t st = (hipStream_t)stream;
  if (tn == 32 && tc == 32) launch<1, 1, 1, 1>(a, grid, st);
  else if (tn == 32 && tc == 64) launch<1, 2, 1, 1>(a, grid, st);
  else if (tn == 64 && tc == 32) launch<2, 1, 1, 1>(a, grid, st);
  else if (tn == 64 && tc == 64) launch<2, 2, 1, 1>(a, grid, st);
  else if (tn == 32 && tc == 128) launch<1, 4, 1, 1>(a, grid, st);
  else if (tn == 128 && tc == 32) launch<4, 1, 1, 1>(a, grid, st);
  else if (tn == 64 && tc == 128) launch<2, 2, 1, 2>(a, grid, st);
  else if (tn == 128 && tc == 64) launch<2, 2, 2, 1>(a, grid, st);
  else launch<2, 2, 2, 2>(a, grid, st);
  FFSR_LAUNCH(wgrad_finish_kernel, dim3((unsigned)((per_tile + 255) / 256)), dim3(256), 0, st, partial, (int)S, T, N, Cin, dw);
  return ffsr_launch_status();
}
