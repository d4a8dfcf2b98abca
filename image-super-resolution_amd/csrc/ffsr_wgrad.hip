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
// grid = row tiles x column tiles x taps x pixel splits (flattened, XCD-aware order); one workgroup = 4 waves arranged WN x WC x WK over a
// (32 FN WN) x (32 FC WC) tile of (n, c): waves along WK take alternate pixel pairs of a staged chunk and are summed
// through LDS at the end.  Chunks of 16 pixels are double buffered (global -> registers -> LDS, one barrier per chunk);
// per-pixel validity of the shifted tap (image borders, end of the split) is computed two chunks ahead by the first wave
// (one ballot -> a 16-bit mask per operand).
// Every split writes its partial tile; a second kernel sums the splits in double precision and ACCUMULATES into dW in
// nn.Conv2d layout [N, Cin, KH, KW] (deterministic, no atomics).
#include "ffsr_common.h"
#include <type_traits>

namespace {

constexpr int KC = 16;   // pixels per staged chunk

struct WgradArgs {
  const float* x;
  const unsigned short* x_hi;   // X as bf16 hi / lo planes [pixels][ldp] instead of x (conv_wgrad3_bf16x3_kernel<.., XPL = true> only)
  const unsigned short* x_lo;
  int ldp;
  const unsigned short* dy_hi;  // dY as planes [pixels][ldq] instead of dy (<.., YPL = true>)
  const unsigned short* dy_lo;
  int ldq;
  const float* dy;
  float* part;
  float* bias_part;     // [splits][N] column sums of dY (bias gradient) or null
  int ldx, ldy;
  int B, H, W, Cin, N, KH, KW, ph, pw;
  long long P, per_split;
  int n_tiles, c_tiles;
};

template <int WN, int WC, int FN, int FC, bool VEC>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs p) {
  constexpr int WK = 4 / (WN * WC);
  constexpr int TN = 32 * FN * WN, TC = 32 * FC * WC;
  constexpr int SA = TN + 32, SB = TC + 32;                 // row strides: the two lane halves land on different banks
  constexpr int STAGE = KC * (SA + SB);
  constexpr int RED = (WK > 1) ? WK * TN * TC : 0;
  constexpr int LDS_FLOATS = (2 * STAGE > RED) ? 2 * STAGE : RED;
  __shared__ float lds[LDS_FLOATS];
  __shared__ unsigned meta[2][2];                           // [slot][0: dY rows valid, 1: X rows valid]: bit k = pixel k of the chunk

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wk = wave % WK, wc = (wave / WK) % WC, wn = wave / (WK * WC);
  // XCD-aware order: workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2).  All (tile, tap) workgroups of
  // one pixel split read the same dY / X rows, so consecutive LOGICAL ids (tile fastest, then tap, then split) are mapped
  // to the same XCD: the 9 taps of a 3x3 layer then share one L2 instead of pulling the rows into eight.
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int q8 = nwg >> 3, rr8 = nwg & 7, xcd = orig & 7;
  const int logical = (xcd < rr8 ? xcd * (q8 + 1) : rr8 * (q8 + 1) + (xcd - rr8) * q8) + (orig >> 3);
  const int T_ = p.KH * p.KW, ntile = p.n_tiles * p.c_tiles;
  const int tile = logical % ntile, tap = (logical / ntile) % T_, split = logical / (ntile * T_);
  const int n0 = (tile / p.c_tiles) * TN, c0 = (tile % p.c_tiles) * TC;
  const int dyo = tap / p.KW - p.ph, dxo = tap % p.KW - p.pw;
  const long long shift = (long long)dyo * p.W + dxo;
  const long long p_begin = (long long)split * p.per_split;
  const long long p_end = (p_begin + p.per_split < p.P) ? p_begin + p.per_split : p.P;
  const int nchunks = (int)((p_end - p_begin + KC - 1) / KC);

  auto make_meta = [&](int chunk) {         // the first wave; lanes 0 .. KC-1 hold the pixels of the chunk (32-bit pixel indices)
    const int q = (int)p_begin + chunk * KC + lane;
    bool va = false, vb = false;
    if (lane < KC && q < (int)p_end) {
      va = true;
      const int row = q / p.W;
      const int xx = q - row * p.W, yy = row % p.H;
      const int sy = yy + dyo, sx = xx + dxo;
      vb = sy >= 0 && sy < p.H && sx >= 0 && sx < p.W;
    }
    const unsigned ma = (unsigned)__ballot(va), mb = (unsigned)__ballot(vb);
    if (lane == 0) {
      meta[chunk & 1][0] = ma;
      meta[chunk & 1][1] = mb;
    }
  };

  // VEC: N % 4 == 0, Cin % 4 == 0, 16-byte aligned rows -> 16-byte global loads and LDS stores (4x fewer loader instructions:
  // the loader's address arithmetic runs on the same SIMD issue port as the MFMAs)
  constexpr int V = VEC ? 4 : 1;
  constexpr int EA = KC * TN / V, EB = KC * TC / V;                      // vector slots of a staged tile
  constexpr int LA = (EA + 255) / 256, LB = (EB + 255) / 256;            // loads per thread and chunk
  typedef typename std::conditional<VEC, floatx4, float>::type vec_t;
  vec_t ra[LA], rb[LB];
  auto fetch = [&](int chunk) {
    const long long q0 = p_begin + (long long)chunk * KC;
    const unsigned ma = meta[chunk & 1][0], mb = meta[chunk & 1][1];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int e = tid + i * 256;
      const int k = e / (TN / V), col = (e % (TN / V)) * V;
      const int n = n0 + col;
      vec_t v = vec_t{};
      if (e < EA && ((ma >> k) & 1u) && n < p.N) v = *reinterpret_cast<const vec_t*>(p.dy + (q0 + k) * p.ldy + n);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      const int e = tid + i * 256;
      const int k = e / (TC / V), col = (e % (TC / V)) * V;
      const int c = c0 + col;
      vec_t v = vec_t{};
      if (e < EB && ((mb >> k) & 1u) && c < p.Cin) v = *reinterpret_cast<const vec_t*>(p.x + (q0 + k + shift) * p.ldx + c);
      rb[i] = v;
    }
  };
  // bias gradient = column sums of dY: the (tap 0, first column tile) workgroups add up the dY tiles they stage anyway
  // (a thread always loads the same V columns: 256 is a multiple of TN / V)
  const bool do_bias = p.bias_part != nullptr && tap == 0 && (tile % p.c_tiles) == 0;
  vec_t bsum = vec_t{};
  auto stash = [&](int buf) {
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < LA; ++i) bsum += ra[i];
    }
    float* As = lds + buf * STAGE;
    float* Bs = As + KC * SA;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int e = tid + i * 256;
      if (e < EA) *reinterpret_cast<vec_t*>(As + (e / (TN / V)) * SA + (e % (TN / V)) * V) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      const int e = tid + i * 256;
      if (e < EB) *reinterpret_cast<vec_t*>(Bs + (e / (TC / V)) * SB + (e % (TC / V)) * V) = rb[i];
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
    if (wave == 0) {
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
      // fragments are double buffered in registers: the LDS reads of pixel pair kp + 1 are issued before the MFMAs of pair
      // kp (reusing one register set makes every read wait for the last MFMA of the previous group to pick up its
      // operands, and every MFMA group wait a full LDS latency: measured 47 % MFMA-busy, 60 % of wave time in issue stalls)
      constexpr int NKP = (KC / 2 + WK - 1) / WK;
      float a[2][FN], b[2][FC];
      auto frag = [&](int set, int kp) {
#pragma unroll
        for (int i = 0; i < FN; ++i) a[set][i] = As[(2 * kp + h) * SA + (wn * FN + i) * 32 + r];
#pragma unroll
        for (int j = 0; j < FC; ++j) b[set][j] = Bs[(2 * kp + h) * SB + (wc * FC + j) * 32 + r];
      };
      frag(0, wk);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);     // (the first pair's reads: the groups below then hold the NEXT pair's)
#pragma unroll
      for (int q = 0; q < NKP; ++q) {
        if (q + 1 < NKP) frag((q + 1) & 1, wk + (q + 1) * WK);
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
          for (int j = 0; j < FC; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q & 1][i], b[q & 1][j], acc[i][j], 0, 0, 0);
        // scheduling shape of one step: the two (paired) LDS reads of the NEXT pixel pair, then this pair's MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, FN * FC, 0);
      }
      if (ch + 1 < nchunks) stash(buf ^ 1);                  // the other buffer was last read before the previous barrier
      if (ch + 2 < nchunks && wave == 0) make_meta(ch + 2);  // slot of chunk ch: its readers (fetch(ch)) are done
      __syncthreads();
    }
  }

  // ---- bias partial: the 256 / (TN / V) threads that share a column group are summed through LDS
  if (do_bias) {          // workgroup-uniform; all stage reads finished at the last barrier
    constexpr int CG = TN / V;
    vec_t* red2 = reinterpret_cast<vec_t*>(lds);
    red2[tid] = bsum;
    __syncthreads();
    if (tid < CG) {
      vec_t t = red2[tid];
#pragma unroll
      for (int j = 1; j < 256 / CG; ++j) t += red2[tid + j * CG];
      float* dst = p.bias_part + (size_t)split * p.N + n0 + tid * V;
      if constexpr (VEC) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n0 + tid * V + e < p.N) dst[e] = t[e];
      } else {
        if (n0 + tid < p.N) dst[0] = t;
      }
    }
    __syncthreads();
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
    float* dst = p.part + ((size_t)split * T + tap) * p.N * p.Cin;
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

// ---- thin 3-wide layers: the three horizontal taps of a kernel row in ONE workgroup -----------------------------------------
// With at most 32 x 32 (n, c) per wave the per-tap kernel above is not MFMA- but L2-bound: each of the 9 tap workgroups
// stages its own copy of the dY rows and of the (shifted) X rows.  Here a workgroup owns a kernel ROW ky: the X tile is staged
// with one pixel of halo on either side (KC + 2 rows) and serves kx = 0, 1, 2 through fragment reads at row offsets 0 / 1 / 2;
// dY is staged once for the three taps; every wave keeps three accumulators.  What a shifted read picks up across an image
// border (the neighbouring row's pixel) is replaced by zero at fragment time from a per-chunk, per-tap validity mask.
// Grid = tiles x KH x splits: a third of the workgroups, a third of the staged bytes, the same MFMA work.
// KC = pixels per staged chunk: 64 for the 32 x 32 tile (4 % faster than 16), 16 elsewhere (32 measured 15-20 % slower on the
// 32 x 128 / 128 x 32 / 64 x 64 tiles: fewer resident workgroups)
template <int WN, int WC, int KC>
__global__ __launch_bounds__(256) void conv_wgrad3_kernel(WgradArgs p) {
  typedef unsigned long long mask_t;
  constexpr int WK = 4 / (WN * WC);
  constexpr int TN = 32 * WN, TC = 32 * WC, KB = KC + 2;
  constexpr int SA = TN + 32, SB = TC + 32;
  constexpr int STAGE = KC * SA + KB * SB;
  constexpr int RED = (WK > 1) ? WK * TN * TC : 0;
  constexpr int LDS_FLOATS = (2 * STAGE > RED) ? 2 * STAGE : RED;
  __shared__ float lds[LDS_FLOATS];
  __shared__ mask_t meta[2][4];                           // [slot][0: dY rows valid, 1 + kx: X valid for tap kx]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wk = wave % WK, wc = (wave / WK) % WC, wn = wave / (WK * WC);
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int q8 = nwg >> 3, rr8 = nwg & 7, xcd = orig & 7;
  const int logical = (xcd < rr8 ? xcd * (q8 + 1) : rr8 * (q8 + 1) + (xcd - rr8) * q8) + (orig >> 3);
  const int ntile = p.n_tiles * p.c_tiles;
  const int tile = logical % ntile, ky = (logical / ntile) % p.KH, split = logical / (ntile * p.KH);
  const int n0 = (tile / p.c_tiles) * TN, c0 = (tile % p.c_tiles) * TC;
  const int dyo = ky - p.ph;
  const long long shift = (long long)dyo * p.W - 1;         // staged X row j <-> flat pixel q0 + shift + j  (pad_w = 1)
  const long long p_begin = (long long)split * p.per_split;
  const long long p_end = (p_begin + p.per_split < p.P) ? p_begin + p.per_split : p.P;
  const int nchunks = (int)((p_end - p_begin + KC - 1) / KC);

  auto make_meta = [&](int chunk) {         // the first wave; lanes 0 .. KC-1 = the pixels of the chunk
    const int q = (int)p_begin + chunk * KC + lane;
    bool va = false, v0 = false, v1 = false, v2 = false;
    if (lane < KC && q < (int)p_end) {
      va = true;
      const int row = q / p.W;
      const int xx = q - row * p.W, sy = row % p.H + dyo;
      const bool yok = sy >= 0 && sy < p.H;
      v0 = yok && xx >= 1;
      v1 = yok;
      v2 = yok && xx + 1 < p.W;
    }
    const mask_t ma = __ballot(va), m0 = __ballot(v0), m1 = __ballot(v1), m2 = __ballot(v2);
    if (lane == 0) {
      meta[chunk & 1][0] = ma;
      meta[chunk & 1][1] = m0;
      meta[chunk & 1][2] = m1;
      meta[chunk & 1][3] = m2;
    }
  };

  constexpr int EA = KC * TN, EB = KB * TC;
  constexpr int LA = (EA + 255) / 256, LB = (EB + 255) / 256;
  float ra[LA], rb[LB];
  mask_t mk_next[3] = {0, 0, 0};       // tap masks of the chunk being fetched (read before the barrier that lets the first
                                            // wave overwrite this meta slot two chunks later)
  const bool do_bias = p.bias_part != nullptr && ky == 0 && (tile % p.c_tiles) == 0;
  float bsum = 0.f;
  auto fetch = [&](int chunk) {
    const long long q0 = p_begin + (long long)chunk * KC;
    const mask_t ma = meta[chunk & 1][0];
    mk_next[0] = meta[chunk & 1][1], mk_next[1] = meta[chunk & 1][2], mk_next[2] = meta[chunk & 1][3];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int e = tid + i * 256;
      const int k = e / TN, n = n0 + e % TN;
      ra[i] = (e < EA && ((ma >> k) & 1) && n < p.N) ? p.dy[(q0 + k) * p.ldy + n] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      const int e = tid + i * 256;
      const int j = e / TC, c = c0 + e % TC;
      const long long g = q0 + shift + j;                    // any in-range pixel may be read: invalid taps are zeroed at fragment time
      rb[i] = (e < EB && g >= 0 && g < p.P && c < p.Cin) ? p.x[g * p.ldx + c] : 0.f;
    }
  };
  auto stash = [&](int buf) {
    float* As = lds + buf * STAGE;
    float* Bs = As + KC * SA;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int e = tid + i * 256;
      if (do_bias) bsum += ra[i];
      if (e < EA) As[(e / TN) * SA + e % TN] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      const int e = tid + i * 256;
      if (e < EB) Bs[(e / TC) * SB + e % TC] = rb[i];
    }
  };

  floatx16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  if (nchunks > 0) {
    if (wave == 0) {
      make_meta(0);
      if (nchunks > 1) make_meta(1);
    }
    __syncthreads();
    fetch(0);
    stash(0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
      const int buf = ch & 1;
      const mask_t m0 = mk_next[0], m1 = mk_next[1], m2 = mk_next[2];   // masks of chunk ch (loaded by its fetch)
      if (ch + 1 < nchunks) fetch(ch + 1);
      const float* As = lds + buf * STAGE;
      const float* Bs = As + KC * SA;
#pragma unroll
      for (int kp = wk; kp < KC / 2; kp += WK) {
        const int k = 2 * kp + h;
        const float a = As[k * SA + wn * 32 + r];
        const float* bp = Bs + k * SB + wc * 32 + r;
        const float b0 = ((m0 >> k) & 1) ? bp[0] : 0.f;
        const float b1 = ((m1 >> k) & 1) ? bp[SB] : 0.f;
        const float b2 = ((m2 >> k) & 1) ? bp[2 * SB] : 0.f;
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b2, acc[2], 0, 0, 0);
      }
      if (ch + 1 < nchunks) stash(buf ^ 1);
      if (ch + 2 < nchunks && wave == 0) make_meta(ch + 2);
      __syncthreads();
    }
  }

  if (do_bias) {          // a thread always loaded the same column tid % TN: sum the 256 / TN threads of a column
    lds[tid] = bsum;
    __syncthreads();
    if (tid < TN) {
      float t = lds[tid];
#pragma unroll
      for (int j = 1; j < 256 / TN; ++j) t += lds[tid + j * TN];
      if (n0 + tid < p.N) p.bias_part[(size_t)split * p.N + n0 + tid] = t;
    }
    __syncthreads();
  }
  const int T = p.KH * 3;
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    floatx16 v = acc[kx];
    if constexpr (WK > 1) {
      float* red = lds;
#pragma unroll
      for (int e = 0; e < 16; ++e) red[(wk * TN + wn * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * TC + wc * 32 + r] = v[e];
      __syncthreads();
      if (wk == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float s = v[e];
#pragma unroll
          for (int q = 1; q < WK; ++q) s += red[(q * TN + wn * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * TC + wc * 32 + r];
          v[e] = s;
        }
      }
      __syncthreads();
    }
    if (wk == 0) {
      float* dst = p.part + ((size_t)split * T + ky * 3 + kx) * p.N * p.Cin;
      const int c = c0 + wc * 32 + r;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = n0 + wn * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (n < p.N && c < p.Cin) dst[(size_t)n * p.Cin + c] = v[e];
      }
    }
  }
}

// ---- split-bf16 variant for the wide 3-wide layers (N, Cin multiples of 128: the refine stack's 128 -> 128 convs at HR) ----------
// The f32 MFMA above runs at 1/16 of the bf16 rate; here the products are the GEMM kernels' three split-bf16 terms
// (hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate): a third of the matrix-pipe time per pixel pair.  A bf16
// MFMA wants 8 consecutive k (= pixels) per lane, the slow dimension of both channels-last operands: the staged tiles stay
// [pixel][channel] images (bf16 hi and lo, 256-byte rows, 16-byte chunks XOR-swizzled by the row) and BOTH fragments come out of
// ds_read_b64_tr_b16, gfx950's transposing LDS read (per 16 lanes: 4 rows x 16 columns, delivered column-major) -- no transpose
// pass anywhere.  As in conv_wgrad3_kernel a workgroup owns a kernel row: the X strip (KC + 2 pixels) serves kx = 0, 1, 2
// through reads at row offsets 0 / 1 / 2, dY is staged once.  What a shifted tap would pick up across an image border is
// removed in the read itself: the lane that supplies that pixel's row address points at a row of zeros (the masks are
// wave-uniform bit sets, one bit test per read).  fp32 -> hi / lo happens once per element on the way into LDS.
// Measured on 32 x 256 x 256 pixels, 128 -> 128 (tools/wgrad_bench.py; f32 MFMA kernel: 5.7 ms): MFMAs + dY fragment reads only
// 1.06 ms (the matrix pipe's own time for 3 x 618 GFLOP), + X fragment reads 1.27 ms, the staging work alone (loads, split,
// LDS writes, no MFMA) 0.93 ms, everything 1.78 ms = 348 TFLOP/s: on a SIMD the vector work and the MFMAs add up rather than
// overlap, however finely they are interleaved (sched_group_barrier shapes, one or two waves per SIMD: +-3 %).  What helped:
// two waves per SIMD with the border masks moved into the read addresses (2.03 -> 1.92), global loads kept in flight a whole
// iteration ahead (-> 1.78).  Operands that arrive as bf16 planes (XPL / YPL below) remove the conversion work and change
// nothing (1.83 fp32 in, 1.93 planes in: two 256-byte rows per pixel instead of one 512-byte row); a variant of the planes
// kernel staged by LDS-DMA (swizzle applied on the global side, no vector work, no staging registers; results bit-identical)
// ran at 2.06 ms and was dropped.  One kernel row (1x3) takes exactly a third of the 3x3 time (tools/probe/wgrad_kh1.py): no
// loss to row sharing between workgroups either -- what is left is the workgroup's own pipeline (fragment-read latency at two
// waves per SIMD, the per-chunk barrier of 8 waves).
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
typedef short wg_short4 __attribute__((ext_vector_type(4)));
typedef short wg_short8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned wg_uint2 __attribute__((ext_vector_type(2)));
typedef unsigned wg_uint4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned wg_img_off(int row, int ch) {   // byte offset of 16-byte chunk ch (8 channels) of a row
  return 256u * (unsigned)row + 16u * (unsigned)(ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}
__device__ __forceinline__ wg_short4 wg_tr_read(const unsigned char* base, unsigned off) {
  typedef __attribute__((address_space(3))) wg_short4 lds_short4;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)(base + off));
}

// EDGE: N / Cin are not multiples of the tile (76 -> 64, 96 -> 32 at HR): channels past the edge are staged as zeros and the
// waves / fragments that hold nothing but padding skip their reads and MFMAs.
// XPL: X arrives as bf16 hi / lo planes (the layer's forward input as the planes GEMM consumed it): its rows go to LDS as they are.
// YPL: dY arrives as planes too (ffsr_act_bwd_planes_f32): nothing is converted in this kernel any more.
template <int KC, bool EDGE, bool XPL, bool YPL>
__global__ __launch_bounds__(512) void conv_wgrad3_bf16x3_kernel(WgradArgs p) {
  typedef unsigned long long mask_t;
  constexpr int TN = 128, TC = 128, KB = KC + 2, NT = 512;
  constexpr int A_PLANE = KC * 256, B_PLANE = (KB + 1) * 256, STAGE = 2 * A_PLANE + 2 * B_PLANE;   // X planes: + one row of zeros
  extern __shared__ __attribute__((aligned(16))) unsigned char wg_smem[];
  mask_t (*meta)[4] = reinterpret_cast<mask_t (*)[4]>(wg_smem + 2 * STAGE);   // [chunk & 3][0: dY rows valid, 1 + kx: tap kx valid]

  // 8 waves = 2 per SIMD: one wave's instruction stream (fragment reads, masks, the fp32 -> bf16 staging work: ~10 vector
  // instructions per MFMA) cannot keep the matrix pipe fed by itself -- with two, one wave's vector work issues while the
  // other's MFMA executes.  Waves as 2 (n) x 4 (c): 64 x 32 of the 128 x 128 tile each, three taps -> 6 accumulators.
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 2, wc = wave & 3;
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int q8 = nwg >> 3, rr8 = nwg & 7, xcd = orig & 7;
  const int logical = (xcd < rr8 ? xcd * (q8 + 1) : rr8 * (q8 + 1) + (xcd - rr8) * q8) + (orig >> 3);
  const int ntile = p.n_tiles * p.c_tiles;
  const int tile = logical % ntile, ky = (logical / ntile) % p.KH, split = logical / (ntile * p.KH);
  const int n0 = (tile / p.c_tiles) * TN, c0 = (tile % p.c_tiles) * TC;
  const int dyo = ky - p.ph;
  const long long shift = (long long)dyo * p.W - 1;         // staged X row j <-> flat pixel q0 + shift + j  (pad_w = 1)
  const long long p_begin = (long long)split * p.per_split;
  const long long p_end = (p_begin + p.per_split < p.P) ? p_begin + p.per_split : p.P;
  const int nchunks = (int)((p_end - p_begin + KC - 1) / KC);

  auto make_meta = [&](int chunk) {         // the first wave; lanes 0 .. KC-1 = the pixels of the chunk
    const int q = (int)p_begin + chunk * KC + lane;
    bool va = false, v0 = false, v1 = false, v2 = false;
    if (lane < KC && q < (int)p_end) {
      va = true;
      const int row = q / p.W;
      const int xx = q - row * p.W, sy = row % p.H + dyo;
      const bool yok = sy >= 0 && sy < p.H;
      v0 = yok && xx >= 1;
      v1 = yok;
      v2 = yok && xx + 1 < p.W;
    }
    const mask_t ma = __ballot(va), m0 = __ballot(v0), m1 = __ballot(v1), m2 = __ballot(v2);
    if (lane == 0) {
      meta[chunk & 3][0] = ma;
      meta[chunk & 3][1] = m0;
      meta[chunk & 3][2] = m1;
      meta[chunk & 3][3] = m2;
    }
  };

  // staging: a thread moves float4 units (4 channels of one pixel); its channel unit u is fixed, its rows are rowt + 16 i.
  // Loads are unconditional (a pixel outside the split / the tensor is read from a clamped address and dropped when it is
  // written to LDS): no branch per load.
  constexpr int RPT = NT / 32, LA = KC / RPT, LB = (KB + RPT - 1) / RPT, NP = LA + LB;
  const int u = tid & 31, rowt = tid >> 5;
  const bool n_ok = !EDGE || n0 + 4 * u < p.N, c_ok = !EDGE || c0 + 4 * u < p.Cin;
  const unsigned uoff_chunk = (unsigned)(u >> 1), uoff_half = 8u * (unsigned)(u & 1);
  // Two chunks are under way besides the one being multiplied: chunk c + 1 sits in registers (loaded during the previous
  // iteration) and is converted / written to the other LDS buffer piece by piece; as soon as a piece's registers are free the
  // same piece of chunk c + 2 is loaded into them -- global loads are in flight all the time, a full iteration ahead of their use.
  floatx4 ra[LA], rb[LB];
  mask_t mk1[3] = {0, 0, 0}, ma1 = 0;      // chunk c + 1 (in registers): tap masks for its MFMA steps, row mask for its LDS image
  mask_t mk2[3] = {0, 0, 0}, ma2 = 0;      // chunk c + 2 (being loaded)
  long long q01 = 0, q02 = 0;
  const bool do_bias = p.bias_part != nullptr && ky == 0 && (tile % p.c_tiles) == 0;
  floatx4 bsum = {0.f, 0.f, 0.f, 0.f};
  const floatx4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto uniform64 = [](mask_t v) -> mask_t {   // an LDS read lands in vector registers: keep the (wave-uniform) masks in scalar ones
    const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)v), hi32 = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((mask_t)hi32 << 32) | lo32;
  };
  auto begin2 = [&](int chunk) {            // chunk >= nchunks: past the end (valid addresses, nothing kept)
    const bool real = chunk < nchunks;
    const int cc = real ? chunk : nchunks - 1;
    ma2 = real ? uniform64(meta[cc & 3][0]) : 0ull, q02 = p_begin + (long long)cc * KC;
    mk2[0] = uniform64(meta[cc & 3][1]), mk2[1] = uniform64(meta[cc & 3][2]), mk2[2] = uniform64(meta[cc & 3][3]);
  };
  auto advance = [&]() { ma1 = ma2, q01 = q02, mk1[0] = mk2[0], mk1[1] = mk2[1], mk1[2] = mk2[2]; };
  auto fetch_piece = [&](int i) {                      // of chunk "2"; pieces 0 .. LA-1: dY rows, LA .. NP-1: X strip rows
    if (i < LA) {
      const long long g = q02 + rowt + RPT * i;
      if constexpr (YPL) {
        const size_t o = (size_t)(g < p.P ? g : p.P - 1) * p.ldq + (n_ok ? n0 + 4 * u : 0);
        const wg_uint2 h = *reinterpret_cast<const wg_uint2*>(p.dy_hi + o), l2 = *reinterpret_cast<const wg_uint2*>(p.dy_lo + o);
        ra[i] = __builtin_bit_cast(floatx4, wg_uint4{h[0], h[1], l2[0], l2[1]});
      } else {
        ra[i] = *reinterpret_cast<const floatx4*>(p.dy + (g < p.P ? g : p.P - 1) * p.ldy + (n_ok ? n0 + 4 * u : 0));
      }
    } else if (i < NP) {
      long long g = q02 + shift + rowt + RPT * (i - LA);      // any in-range pixel may be read: invalid taps read the row of zeros
      g = g < 0 ? 0 : (g < p.P ? g : p.P - 1);
      if constexpr (XPL) {
        const size_t o = (size_t)g * p.ldp + (c_ok ? c0 + 4 * u : 0);
        const wg_uint2 h = *reinterpret_cast<const wg_uint2*>(p.x_hi + o), l2 = *reinterpret_cast<const wg_uint2*>(p.x_lo + o);
        rb[i - LA] = __builtin_bit_cast(floatx4, wg_uint4{h[0], h[1], l2[0], l2[1]});
      } else {
        rb[i - LA] = *reinterpret_cast<const floatx4*>(p.x + g * p.ldx + (c_ok ? c0 + 4 * u : 0));
      }
    }
  };
  auto put = [&](unsigned char* hi_plane, unsigned char* lo_plane, int row, const floatx4& v) {
    unsigned h0, h1, l0, l1;
    ffsr_split2(v[0], v[1], h0, l0);
    ffsr_split2(v[2], v[3], h1, l1);
    const unsigned o = wg_img_off(row, (int)uoff_chunk) + uoff_half;
    *reinterpret_cast<wg_uint2*>(hi_plane + o) = wg_uint2{h0, h1};
    *reinterpret_cast<wg_uint2*>(lo_plane + o) = wg_uint2{l0, l1};
  };
  auto stash_piece = [&](int buf, int i) {             // of chunk "1": the LDS side of its fetch_piece(i)
    unsigned char* base = wg_smem + buf * STAGE;
    if (i < LA) {
      const floatx4 v = (((ma1 >> (rowt + RPT * i)) & 1) && n_ok) ? ra[i] : zero4;
      if constexpr (YPL) {
        const wg_uint4 w = __builtin_bit_cast(wg_uint4, v);      // (hi pair 0, hi pair 1, lo pair 0, lo pair 1)
        if (do_bias) {                                            // (block-uniform) the bias gradient wants the values back: hi + lo
          const floatx4 hv = {__builtin_bit_cast(float, w[0] << 16), __builtin_bit_cast(float, w[0] & 0xffff0000u),
                              __builtin_bit_cast(float, w[1] << 16), __builtin_bit_cast(float, w[1] & 0xffff0000u)};
          const floatx4 lv = {__builtin_bit_cast(float, w[2] << 16), __builtin_bit_cast(float, w[2] & 0xffff0000u),
                              __builtin_bit_cast(float, w[3] << 16), __builtin_bit_cast(float, w[3] & 0xffff0000u)};
          bsum += hv + lv;
        }
        const unsigned o = wg_img_off(rowt + RPT * i, (int)uoff_chunk) + uoff_half;
        *reinterpret_cast<wg_uint2*>(base + o) = wg_uint2{w[0], w[1]};
        *reinterpret_cast<wg_uint2*>(base + A_PLANE + o) = wg_uint2{w[2], w[3]};
      } else {
        if (do_bias) bsum += v;
        put(base, base + A_PLANE, rowt + RPT * i, v);
      }
    } else if (i < NP) {
      const int j = rowt + RPT * (i - LA);
      const long long g = q01 + shift + j;
      const floatx4 v = (g >= 0 && g < p.P && c_ok) ? rb[i - LA] : zero4;
      if constexpr (XPL) {
        if (j < KB) {
          const wg_uint4 w = __builtin_bit_cast(wg_uint4, v);
          const unsigned o = wg_img_off(j, (int)uoff_chunk) + uoff_half;
          *reinterpret_cast<wg_uint2*>(base + 2 * A_PLANE + o) = wg_uint2{w[0], w[1]};
          *reinterpret_cast<wg_uint2*>(base + 2 * A_PLANE + B_PLANE + o) = wg_uint2{w[2], w[3]};
        }
      } else {
        if (j < KB) put(base + 2 * A_PLANE, base + 2 * A_PLANE + B_PLANE, j, v);
      }
    }
  };

  floatx16 acc[3][2];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][i][e] = 0.f;

  // transposing-read geometry of this lane: group gq of 16 lanes takes columns 16 (gq & 1) .. + 15 of a 32-column fragment and
  // pixels 8 (gq >> 1) .. + 7 of the 16-pixel step; lane 4 q + pp of the group supplies row q, columns 4 pp .. 4 pp + 3.
  // The swizzle of a row depends on (row & 3) and ((row >> 2) & 3) only, both unchanged by + 16 s: every read of the chunk is one
  // of these per-lane byte offsets + 4096 s as an immediate.
  const int gq = (lane >> 4) & 3, li = lane & 15, rq = li >> 2, pp = li & 3;
  const int krow = 8 * (gq >> 1) + rq;
  const unsigned half8 = 8u * (unsigned)(pp & 1);
  const int chA = (wn * 64 + 16 * (gq & 1)) / 8 + (pp >> 1), chB = (wc * 32 + 16 * (gq & 1)) / 8 + (pp >> 1);
  unsigned offA[2][2], offB[3][2];
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
#pragma unroll
    for (int fn = 0; fn < 2; ++fn) offA[fn][rr] = wg_img_off(krow + 4 * rr, chA + 4 * fn) + half8;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) offB[kx][rr] = wg_img_off(krow + kx + 4 * rr, chB) + half8;
  }
  auto frag = [&](const unsigned char* plane, const unsigned (&off)[2]) -> wg_bf16x8 {
    const wg_short4 v0 = wg_tr_read(plane, off[0]);
    const wg_short4 v1 = wg_tr_read(plane, off[1]);
    const wg_short8 v = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    return __builtin_bit_cast(wg_bf16x8, v);
  };
  // A tap that must not see pixel k (image border, rows outside the image, end of the split) reads the X row it would pair with
  // pixel k from the plane's row of zeros instead: a transposing read takes one row address per lane, so the mask costs three
  // vector instructions per read pair (bit of the lane's row -> offset select) and no fragment copies.
  const unsigned zoff = 256u * KB + 8u * (unsigned)(lane & 31);
  unsigned dB[3][2];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) dB[kx][rr] = offB[kx][rr] - zoff;

  // which of this wave's two 32-row fragments hold real output channels, given that its 32 columns hold real input channels
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const bool act_c = !EDGE || c0 + (wave_u & 3) * 32 < p.Cin;
  const bool act_n[2] = {act_c && (!EDGE || n0 + (wave_u >> 2) * 64 < p.N), act_c && (!EDGE || n0 + (wave_u >> 2) * 64 + 32 < p.N)};

  if (nchunks > 0) {
    if (tid < 256)           // the rows of zeros: 2 buffers x 2 planes x 64 dwords
      *reinterpret_cast<unsigned*>(wg_smem + (tid >> 7) * STAGE + 2 * A_PLANE + ((tid >> 6) & 1) * B_PLANE + 256 * KB + 4 * (tid & 63)) = 0u;
    if (wave == 0) {
      make_meta(0);
      if (nchunks > 1) make_meta(1);
      if (nchunks > 2) make_meta(2);
    }
    __syncthreads();
    begin2(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) fetch_piece(i);
    advance();
    begin2(1);
#pragma unroll
    for (int i = 0; i < NP; ++i) {      // chunk 0 into LDS buffer 0, chunk 1 on its way into the registers
      stash_piece(0, i);
      fetch_piece(i);
    }
    mask_t mk0[3] = {mk1[0], mk1[1], mk1[2]};   // tap masks of the chunk about to be multiplied
    advance();
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
      const int buf = ch & 1;
      const mask_t mk[3] = {mk0[0], mk0[1], mk0[2]};   // tap masks of chunk ch (wave-uniform, in scalar registers)
      // Side work of the chunk, dealt over its (step, tap) slots: piece by piece, chunk ch + 1 goes from the registers to the other
      // LDS buffer (its readers finished before the last barrier) and chunk ch + 2 is loaded in its place.  Past the last chunk
      // the loads repeat the last rows and are dropped: no branch.
      begin2(ch + 2);
      constexpr int SLOTS = 3 * (KC / 16), PPER = (NP + SLOTS - 1) / SLOTS;
      const unsigned char* Ahi = wg_smem + buf * STAGE;
      const unsigned char* Alo = Ahi + A_PLANE;
      const unsigned char* Bhi = Ahi + 2 * A_PLANE;
      const unsigned char* Blo = Bhi + B_PLANE;
      // X fragments are read one (step, tap) slot ahead into the other of two register sets: the slot's MFMAs cover their LDS latency
      wg_bf16x8 bh[2], bl[2];
      auto read_b = [&](int set, int s, int kx) {
        unsigned ob[2];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {      // bit 16 s + 4 rr + krow of the tap's mask: the pixel of the row this lane addresses
          const unsigned m = (unsigned)(mk[kx] >> (16 * s + 4 * rr));
          ob[rr] = zoff + ((0u - ((m >> krow) & 1u)) & (dB[kx][rr] + 4096u * s));
        }
        bh[set] = frag(Bhi, ob), bl[set] = frag(Blo, ob);
      };
      if (act_n[0]) read_b(0, 0, 0);
      static_for<0, KC / 16>([&](auto s_tag) {
        constexpr int s = decltype(s_tag)::value;
        wg_bf16x8 ah[2], al[2];
#pragma unroll
        for (int fn = 0; fn < 2; ++fn)
          if (act_n[fn]) {
            ah[fn] = frag(Ahi + 4096 * s, offA[fn]);
            al[fn] = frag(Alo + 4096 * s, offA[fn]);
          }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int slot = 3 * s + kx;
          if (act_n[0]) {                    // (wave-uniform; always true without EDGE)
            if (slot + 1 < SLOTS) read_b((slot + 1) & 1, (slot + 1) / 3, (slot + 1) % 3);
#pragma unroll
            for (int fn = 0; fn < 2; ++fn)
              if (act_n[fn]) {
                acc[kx][fn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[fn], bh[slot & 1], acc[kx][fn], 0, 0, 0);
                acc[kx][fn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[fn], bl[slot & 1], acc[kx][fn], 0, 0, 0);
                acc[kx][fn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[fn], bh[slot & 1], acc[kx][fn], 0, 0, 0);
              }
          }
#pragma unroll
          for (int i = 0; i < PPER; ++i) {
            stash_piece(buf ^ 1, slot * PPER + i);
            fetch_piece(slot * PPER + i);
          }
        }
      });
      mk0[0] = mk1[0], mk0[1] = mk1[1], mk0[2] = mk1[2];
      advance();
      if (ch + 3 < nchunks && wave == 0) make_meta(ch + 3);   // slot of chunk ch - 1, last read when chunk ch - 1 was "2" (three iterations ago)
      __syncthreads();
    }
  }

  if (do_bias) {          // a thread always loaded the same 4 columns 4 u .. 4 u + 3: sum the 16 threads of a column unit
    float* red = reinterpret_cast<float*>(wg_smem);
    __syncthreads();
    *reinterpret_cast<floatx4*>(red + rowt * TN + 4 * u) = bsum;
    __syncthreads();
    if (tid < TN) {
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < RPT; ++j) t += red[j * TN + tid];
      if (!EDGE || n0 + tid < p.N) p.bias_part[(size_t)split * p.N + n0 + tid] = t;
    }
  }
  const int T = p.KH * 3, r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    float* dst = p.part + ((size_t)split * T + ky * 3 + kx) * p.N * p.Cin;
    const int c = c0 + wc * 32 + r;
#pragma unroll
    for (int fn = 0; fn < 2; ++fn)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = n0 + wn * 64 + fn * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (!EDGE || (n < p.N && c < p.Cin)) dst[(size_t)n * p.Cin + c] = acc[kx][fn][e];
      }
  }
}

// ---- thin 3x3 layers at HR resolution: N <= 4 outputs (refine.10 128 -> 3, the edge / gate heads) or Cin <= 4 inputs (refine.0
// 3 -> 128) ----------------------------------------------------------------------------------------------------------------------
// On the MFMA kernels the thin side pads to a 32-wide tile: 8 - 32x the layer's work (2.5 - 2.8 ms per layer at 32 x 256 x 256).
// Here the gradient is what it is, a streaming outer-product accumulation on the vector ALU, the mirror image of
// conv3x3_thin_kernel (ffsr_fusion.hip): LPP lanes share a pixel, a lane owns 4 channels of the WIDE side and keeps its slice
// of dW in registers (N x 9 or 9 x 4 float4 accumulators) while its group walks runs of pixels with a ring of X columns;
// the thin side's few values per pixel are broadcast loads.  Groups of a block take vertically adjacent rows; a block walks
// several (row block, run) tiles, sums its groups through LDS in a fixed order and writes ONE partial tile (the finishing
// kernel adds the blocks in double precision).  Exact fp32 FMA.
template <int LPP>
__device__ __forceinline__ void thin_tile(int tile, int runs_per_row, int nyb, int run, int H, int W, int g, int& b, int& y, bool& live,
                                          int& x0, int& x1) {
  constexpr int GPB = 256 / LPP;
  const int xr = tile % runs_per_row, yb = (tile / runs_per_row) % nyb;
  b = tile / (runs_per_row * nyb);
  live = yb * GPB + g < H;
  y = min(yb * GPB + g, H - 1);
  x0 = xr * run, x1 = min(W, x0 + run);
}

// N <= 4 outputs, Cin = 4 LPP: dw[n, c, tap] += sum_pix dy[pix, n] x[pix + tap, c]
template <int N, int LPP>
__global__ __launch_bounds__(256) void wgrad_thin_out_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy, int ldy,
                                                             float* __restrict__ part, float* __restrict__ bias_part, int B, int H, int W,
                                                             int run, int ntiles) {
  constexpr int CIN = 4 * LPP, GPB = 256 / LPP, R = N == 4 ? 6 : 8;
  __shared__ floatx4 red[GPB][9][LPP];
  const int g = threadIdx.x / LPP, l = threadIdx.x % LPP;
  const int runs_per_row = (W + run - 1) / run, nyb = (H + GPB - 1) / GPB;
  const char* xb_ = reinterpret_cast<const char*>(x);
  const unsigned ldb = (unsigned)ldx * 4u, cmax = (unsigned)(W - 1) * ldb;
  const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
  floatx4 acc[N][9];
  float bs[N];
#pragma unroll
  for (int n = 0; n < N; ++n) {
    bs[n] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[n][t] = zero;
  }
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int b, y, x0, x1;
    bool live;
    thin_tile<LPP>(tile, runs_per_row, nyb, run, H, W, g, b, y, live, x0, x1);
    unsigned roff[3];
    float rowm[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = y + ky - 1;
      rowm[ky] = (yy >= 0 && yy < H) ? 1.f : 0.f;          // zero padding: rows / columns outside the image contribute nothing
      roff[ky] = (unsigned)(((b * H + min(max(yy, 0), H - 1)) * W) * ldx + 4 * l) * 4u;
    }
    auto col = [&](int ky, unsigned coff, float m) -> floatx4 {
      return *reinterpret_cast<const floatx4*>(xb_ + (size_t)(roff[ky] + coff)) * (m * rowm[ky]);
    };
    floatx4 ring[R][3];       // columns x-1, x, x+1 = ring[j], [j+1], [j+2]; x+2 .. x+R-3 in flight; x+R-2 issued (conv3x3_thin_kernel)
#pragma unroll
    for (int j = 0; j < R - 1; ++j) {
      const int c = x0 - 1 + j;
      const unsigned coff = (unsigned)min(max(c, 0), W - 1) * ldb;
      const float m = (c >= 0 && c < W) ? 1.f : 0.f;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) ring[j][ky] = col(ky, coff, m);
    }
    unsigned coff = (unsigned)min(x0 + R - 2, W - 1) * ldb;
    const float* dyp = dy + (((size_t)b * H + y) * W) * ldy;
    const float lm = live ? 1.f : 0.f;
    auto dyload = [&](int xx) -> floatx4 { return *reinterpret_cast<const floatx4*>(dyp + (size_t)min(xx, W - 1) * ldy); };
    floatx4 dq[2] = {dyload(x0), dyload(x0 + 1)};          // dY two pixels ahead of its use
    for (int xb = x0; xb < x1; xb += R) {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const int xx = xb + j;
        const float m = (xx + R - 2 < W) ? 1.f : 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) ring[(j + R - 1) % R][ky] = col(ky, coff, m);
        coff = min(coff + ldb, cmax);
        const floatx4 d4 = dq[j & 1] * (xx < x1 ? lm : 0.f);
        dq[j & 1] = dyload(xx + 2);
#pragma unroll
        for (int n = 0; n < N; ++n) {
          bs[n] += d4[n];
#pragma unroll
          for (int t = 0; t < 9; ++t) acc[n][t] += ring[(j + t % 3) % R][t / 3] * d4[n];
        }
      }
    }
  }
  // groups -> one partial tile of the block: part[block][tap][n][c]
  float* dst = part + (size_t)blockIdx.x * 9 * N * CIN;
#pragma unroll
  for (int n = 0; n < N; ++n) {
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 9; ++t) red[g][t][l] = acc[n][t];
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * LPP; i += 256) {
      const int t = i / LPP, ll = i % LPP;
      floatx4 s = red[0][t][ll];
#pragma unroll
      for (int gg = 1; gg < GPB; ++gg) s += red[gg][t][ll];
      *reinterpret_cast<floatx4*>(dst + ((size_t)t * N + n) * CIN + 4 * ll) = s;
    }
  }
  if (bias_part) {          // every lane of a group holds the same sums: lane 0 of each group speaks
    __syncthreads();
    float* rb = reinterpret_cast<float*>(&red[0][0][0]);
    if (l == 0)
#pragma unroll
      for (int n = 0; n < N; ++n) rb[g * N + n] = bs[n];
    __syncthreads();
    if (threadIdx.x < N) {
      float s = 0.f;
      for (int gg = 0; gg < GPB; ++gg) s += rb[gg * N + threadIdx.x];
      bias_part[(size_t)blockIdx.x * N + threadIdx.x] = s;
    }
  }
}

// Cin <= 4 inputs (map stride 4), N = 4 LPP outputs: a lane owns 4 output channels, the 9 x 4 input values of a pixel are broadcast loads
template <int LPP>
__global__ __launch_bounds__(256) void wgrad_thin_in_kernel(const float* __restrict__ x, const float* __restrict__ dy, int ldy,
                                                            float* __restrict__ part, float* __restrict__ bias_part, int B, int H, int W,
                                                            int Cin, int run, int ntiles) {
  constexpr int NN = 4 * LPP, GPB = 256 / LPP, R = 6;
  __shared__ floatx4 red[GPB][4][LPP];
  const int g = threadIdx.x / LPP, l = threadIdx.x % LPP;
  const int runs_per_row = (W + run - 1) / run, nyb = (H + GPB - 1) / GPB;
  const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
  floatx4 acc[9][4], bs = zero;        // [tap][input channel] over the lane's 4 output channels
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = zero;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int b, y, x0, x1;
    bool live;
    thin_tile<LPP>(tile, runs_per_row, nyb, run, H, W, g, b, y, live, x0, x1);
    const float* rowp[3];
    float rowm[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = y + ky - 1;
      rowm[ky] = (yy >= 0 && yy < H) ? 1.f : 0.f;
      rowp[ky] = x + (((size_t)b * H + min(max(yy, 0), H - 1)) * W) * 4;
    }
    auto col = [&](int ky, int c) -> floatx4 {
      return *reinterpret_cast<const floatx4*>(rowp[ky] + (size_t)min(max(c, 0), W - 1) * 4) * ((c >= 0 && c < W) ? rowm[ky] : 0.f);
    };
    floatx4 ring[R][3];
#pragma unroll
    for (int j = 0; j < R - 1; ++j)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) ring[j][ky] = col(ky, x0 - 1 + j);
    const float* dyp = dy + (((size_t)b * H + y) * W) * ldy + 4 * l;
    const float lm = live ? 1.f : 0.f;
    auto dyload = [&](int xx) -> floatx4 { return *reinterpret_cast<const floatx4*>(dyp + (size_t)min(xx, W - 1) * ldy); };
    floatx4 dq[2] = {dyload(x0), dyload(x0 + 1)};          // dY two pixels ahead of its use
    for (int xb = x0; xb < x1; xb += R) {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const int xx = xb + j;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) ring[(j + R - 1) % R][ky] = col(ky, xx + R - 2);
        const floatx4 d4 = dq[j & 1] * (xx < x1 ? lm : 0.f);
        dq[j & 1] = dyload(xx + 2);
        bs += d4;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const floatx4 xv = ring[(j + t % 3) % R][t / 3];
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[t][c] += d4 * xv[c];
        }
      }
    }
  }
  // part[block][tap][n][c], c < Cin
  float* dst = part + (size_t)blockIdx.x * 9 * NN * Cin;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) red[g][c][l] = acc[t][c];
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * LPP; i += 256) {
      const int c = i / LPP, ll = i % LPP;
      floatx4 s = red[0][c][ll];
#pragma unroll
      for (int gg = 1; gg < GPB; ++gg) s += red[gg][c][ll];
      if (c < Cin)
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[((size_t)t * NN + 4 * ll + q) * Cin + c] = s[q];
    }
  }
  if (bias_part) {
    __syncthreads();
    red[g][0][l] = bs;
    __syncthreads();
    if (threadIdx.x < LPP) {
      floatx4 s = red[0][0][threadIdx.x];
#pragma unroll
      for (int gg = 1; gg < GPB; ++gg) s += red[gg][0][threadIdx.x];
      *reinterpret_cast<floatx4*>(bias_part + (size_t)blockIdx.x * NN + 4 * threadIdx.x) = s;
    }
  }
}

// dW[n, c, t] += sum_s part[s, t, n, c].  One output per 8 threads: each sums every 8th split (four independent double chains:
// the loads of one trip are in flight together), the 8 partial sums are added in a fixed order -- a thread per output walked
// its S partials serially (S / 4 dependent load latencies: 12 - 100 us per layer, 185 layers per training step).
__global__ __launch_bounds__(256) void wgrad_finish_kernel(const float* __restrict__ part, int S, int T, int N, int Cin,
                                                           float* __restrict__ dw) {
  __shared__ double red[8][32];
  const int lane32 = threadIdx.x & 31, sg = threadIdx.x >> 5;
  const long long idx = (long long)blockIdx.x * 32 + lane32;
  const long long per = (long long)N * Cin;
  const bool live = idx < per * T;
  const int t = live ? (int)(idx / per) : 0;
  const long long nc = live ? idx % per : 0;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (live) {
    const float* pp = part + (size_t)t * per + nc;
    const size_t st = (size_t)T * per;
    int k = sg;
    for (; k + 24 < S; k += 32) {
      s0 += pp[(size_t)k * st];
      s1 += pp[(size_t)(k + 8) * st];
      s2 += pp[(size_t)(k + 16) * st];
      s3 += pp[(size_t)(k + 24) * st];
    }
    for (; k < S; k += 8) s0 += pp[(size_t)k * st];
  }
  red[sg][lane32] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sg == 0 && live) {
    double a = red[0][lane32];
#pragma unroll
    for (int j = 1; j < 8; ++j) a += red[j][lane32];
    dw[nc * T + t] += (float)a;
  }
}

template <int WN, int WC, int FN, int FC>
void launch(const WgradArgs& a, dim3 grid, hipStream_t st) {
  const bool vec = (a.N % 4 == 0) && (a.Cin % 4 == 0) && (a.ldx % 4 == 0) && (a.ldy % 4 == 0) &&
                   ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.dy)) & 15) == 0;
  if (vec)
    FFSR_LAUNCH((conv_wgrad_kernel<WN, WC, FN, FC, true>), grid, dim3(256), 0, st, a);
  else
    FFSR_LAUNCH((conv_wgrad_kernel<WN, WC, FN, FC, false>), grid, dim3(256), 0, st, a);
}

}  // namespace

// x [B,H,W,ldx] (Cin channels), dy [B,H,W,ldy] (N channels): dw [N, Cin, KH, KW] += the weight gradient of the stride-1
// convolution with zero padding (pad_h, pad_w) (a linear layer: B = H = 1, W = rows, KH = KW = 1); dbias [N] += column sums of
// dy (the bias gradient; null = skip).
// partial: caller-owned scratch of partial_floats floats (>= KH*KW*N*Cin + N; more lets the pixels be split over more
// workgroups).
// thin 3x3 layers at HR resolution on the vector-ALU kernels; returns 1 if it took the layer, 0 if not, < 0 on error
static int wgrad_thin(const float* x, int ldx, const float* dy, int ldy, float* dw, float* dbias, float* partial, long long partial_floats,
                      int B, int H, int W, int Cin, int N, hipStream_t st) {
  const long long P = (long long)B * H * W;
  const bool aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(partial)) & 15) == 0 &&
                       (ldx % 4 == 0) && (ldy % 4 == 0);
  if (!aligned || P < 16384 || P * (long long)(ldx > ldy ? ldx : ldy) * 4 >= (1ll << 32)) return 0;
  const bool thin_out = N <= 4 && (Cin == 8 || Cin == 16 || Cin == 32 || Cin == 64 || Cin == 128) && N <= Cin / 4;
  const bool thin_in = Cin <= 4 && ldx == 4 && (N == 8 || N == 16 || N == 32 || N == 64 || N == 128);
  if (!thin_out && !thin_in) return 0;
  const int lpp = (thin_out ? Cin : N) / 4, gpb = 256 / lpp;
  const int run = W >= 64 ? 32 : (W >= 16 ? 8 : W);
  const long long ntiles = (long long)B * ((H + gpb - 1) / gpb) * ((W + run - 1) / run);
  const long long per_block = 9ll * N * Cin + (dbias ? N : 0);
  long long blocks = ntiles < 1024 ? ntiles : 1024;
  if (blocks > partial_floats / per_block) blocks = partial_floats / per_block;
  if (blocks < 1 || ntiles >= (1ll << 31)) return 0;
  float* bias_part = dbias ? partial + (size_t)blocks * 9 * N * Cin : nullptr;
  const dim3 grid((unsigned)blocks);
#define FFSR_THIN_OUT(NV, LPP) \
  FFSR_LAUNCH((wgrad_thin_out_kernel<NV, LPP>), grid, dim3(256), 0, st, x, ldx, dy, ldy, partial, bias_part, B, H, W, run, (int)ntiles)
#define FFSR_THIN_OUT_N(LPP)                                                                                     \
  switch (N) {                                                                                                   \
    case 1: FFSR_THIN_OUT(1, LPP); break;                                                                        \
    case 2: FFSR_THIN_OUT(2, LPP); break;                                                                        \
    case 3: FFSR_THIN_OUT(3, LPP); break;                                                                        \
    default: FFSR_THIN_OUT(4, LPP); break;                                                                       \
  }
#define FFSR_THIN_IN(LPP) \
  FFSR_LAUNCH((wgrad_thin_in_kernel<LPP>), grid, dim3(256), 0, st, x, dy, ldy, partial, bias_part, B, H, W, Cin, run, (int)ntiles)
  if (thin_out) {
    switch (lpp) {
      case 2: FFSR_THIN_OUT_N(2); break;
      case 4: FFSR_THIN_OUT_N(4); break;
      case 8: FFSR_THIN_OUT_N(8); break;
      case 16: FFSR_THIN_OUT_N(16); break;
      default: FFSR_THIN_OUT_N(32); break;
    }
  } else {
    switch (lpp) {
      case 2: FFSR_THIN_IN(2); break;
      case 4: FFSR_THIN_IN(4); break;
      case 8: FFSR_THIN_IN(8); break;
      case 16: FFSR_THIN_IN(16); break;
      default: FFSR_THIN_IN(32); break;
    }
  }
#undef FFSR_THIN_OUT
#undef FFSR_THIN_OUT_N
#undef FFSR_THIN_IN
  const long long nw = 9ll * N * Cin;
  FFSR_LAUNCH(wgrad_finish_kernel, dim3((unsigned)((nw + 31) / 32)), dim3(256), 0, st, partial, (int)blocks, 9, N, Cin, dw);
  if (dbias)
    FFSR_LAUNCH(wgrad_finish_kernel, dim3((unsigned)((N + 31) / 32)), dim3(256), 0, st, bias_part, (int)blocks, 1, N, 1, dbias);
  const int rc = ffsr_launch_status();
  return rc == FFSR_OK ? 1 : rc;
}

static int wgrad_impl(const float* x, int ldx, const float* dy, int ldy, float* dw, float* dbias, float* partial,
                      long long partial_floats, int B, int H, int W, int Cin, int N, int KH, int KW, int pad_h, int pad_w,
                      bool split_bf16, void* stream, const void* x_hi = nullptr, const void* x_lo = nullptr, int ldp = 0,
                      const void* dy_hi = nullptr, const void* dy_lo = nullptr, int ldq = 0) {
  const bool xpl = x_hi != nullptr, ypl = dy_hi != nullptr;
  alignas(16) static const float dummy_v[4] = {0.f, 0.f, 0.f, 0.f};
  const float* const dummy = dummy_v;
  if (ypl) {
    FFSR_CHECK(xpl && dy_lo && !dy && (ldq & 31) == 0 && ldq >= N &&
               ((reinterpret_cast<uintptr_t>(dy_hi) | reinterpret_cast<uintptr_t>(dy_lo)) & 15) == 0);
    dy = dummy, ldy = N;     // (placeholders for the shared checks: the planes kernel never reads dy)
  }
  if (xpl) {
    FFSR_CHECK(x_lo && !x && (ldp & 31) == 0 && ldp >= Cin && split_bf16 &&
               ((reinterpret_cast<uintptr_t>(x_hi) | reinterpret_cast<uintptr_t>(x_lo)) & 15) == 0);
    x = dummy, ldx = Cin;        // (placeholders for the shared checks below: the planes kernel never reads x)
  }
  FFSR_CHECK(x && dy && dw && partial && B > 0 && H > 0 && W > 0 && Cin > 0 && N > 0 && KH > 0 && KW > 0 && ldx >= Cin && ldy >= N);
  FFSR_CHECK(pad_h >= 0 && pad_w >= 0 && pad_h < KH && pad_w < KW && (long long)B * H * W < (1ll << 31) - 64);
  const int T = KH * KW;
  const long long per_tile = (long long)T * N * Cin + (dbias ? N : 0);     // floats of scratch per pixel split
  FFSR_CHECK(partial_floats >= per_tile);
  if (!xpl && KH == 3 && KW == 3 && pad_h == 1 && pad_w == 1 && (N <= 4 || Cin <= 4)) {
    const int took = wgrad_thin(x, ldx, dy, ldy, dw, dbias, partial, partial_floats, B, H, W, Cin, N, (hipStream_t)stream);
    if (took != 0) return took < 0 ? took : FFSR_OK;
  }
  WgradArgs a;
  a.x_hi = (const unsigned short*)x_hi, a.x_lo = (const unsigned short*)x_lo, a.ldp = ldp;
  a.dy_hi = (const unsigned short*)dy_hi, a.dy_lo = (const unsigned short*)dy_lo, a.ldq = ldq;
  a.x = x, a.dy = dy, a.part = partial, a.bias_part = nullptr, a.ldx = ldx, a.ldy = ldy;
  a.B = B, a.H = H, a.W = W, a.Cin = Cin, a.N = N, a.KH = KH, a.KW = KW, a.ph = pad_h, a.pw = pad_w;
  a.P = (long long)B * H * W;
  const int tn = N <= 32 ? 32 : (N <= 64 ? 64 : 128), tc = Cin <= 32 ? 32 : (Cin <= 64 ? 64 : 128);
  a.n_tiles = (N + tn - 1) / tn, a.c_tiles = (Cin + tc - 1) / tc;
  // (wide3, below, always works on 128 x 128 tiles)
  // thin 3-wide layers (one 32 x 32 MFMA tile per wave): the three horizontal taps share a workgroup (conv_wgrad3_kernel)
  const bool row3 = KW == 3 && pad_w == 1 && (tn / 32) * (tc / 32) <= 4 && W >= 2;
  // wide 3-wide layers in split-bf16 mode: the transposing-read bf16 kernel (float4 staging: 4-channel units, 16-byte rows)
  const bool wide_full = (N % 128 == 0) && (Cin % 128 == 0);
  const bool wide_edge = !wide_full && (N % 4 == 0) && (Cin % 4 == 0) && (N > 64 || Cin > 64) && N >= 32 && Cin >= 32 && (long long)B * H * W >= 65536;
  const bool wide3 = split_bf16 && KW == 3 && pad_w == 1 && W >= 2 && (wide_full || wide_edge) &&
                     (ldx % 4 == 0) && (ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0;
  if (xpl && !wide3) return FFSR_EINVAL;     // planes input: only the shapes the bf16 kernel takes
  const long long tiles = (long long)a.n_tiles * a.c_tiles * ((row3 || wide3) ? KH : T);
  // splits: enough workgroups to fill the chip a few times, at least 256 pixels each, bounded by the scratch.  The MFMA pipe
  // of a SIMD serves its resident waves one after the other, so the kernel takes as long as the CU that hosts the most
  // workgroups: the count is chosen so that tiles x taps x splits fills whole multiples of the 256 CUs (9 taps x 64 splits =
  // 576 workgroups = 2.25 per CU ran at 75 % of 9 x 85 = 765).
  long long S = (2048 + tiles - 1) / tiles;
  const long long max_by_px = (a.P + 255) / 256;
  if (S > max_by_px) S = max_by_px;
  if (S > partial_floats / per_tile) S = partial_floats / per_tile;
  if (S > 65535) S = 65535;
  if (S < 1) S = 1;
  {
    // the SMALLEST split count that gives about 6 workgroups per CU (3 were 25 % slower on the thin tiles) and balanced: every extra split is one
    // more partial tile for the finishing kernel to read
    long long lo = wide3 ? 512 / tiles : (1500 + tiles - 1) / tiles;   // (the wide bf16 kernel: one workgroup per CU at a time)
    if (lo > S) lo = S;
    if (lo < 1) lo = 1;
    long long best = S;
    double best_eff = 0.0;
    for (long long c = lo; c <= S; ++c) {
      const long long n = tiles * c;
      const double eff = (double)n / (double)(((n + 255) / 256) * 256);
      if (eff > best_eff + 1e-9) best_eff = eff, best = c;
      if (eff >= 0.97) break;
    }
    S = best;
  }
  a.per_split = ((a.P + S - 1) / S + 63) / 64 * 64;      // a multiple of every kernel's chunk length (16 / 32 / 64 pixels)
  S = (a.P + a.per_split - 1) / a.per_split;
  FFSR_CHECK(tiles * S < (1ll << 31));
  if (dbias) a.bias_part = partial + (size_t)S * T * N * Cin;
  const dim3 grid((unsigned)(tiles * S));
  hipStream_t st = (hipStream_t)stream;
  if (wide3) {
    constexpr int KCW = 64, LDS = 2 * (2 * KCW * 256 + 2 * (KCW + 3) * 256) + 128;   // two stages + the 4 x 4 chunk masks
    static unsigned long long attr_set = 0;
    {
      const void* fns[6] = {reinterpret_cast<const void*>(&conv_wgrad3_bf16x3_kernel<KCW, false, false, false>),
                            reinterpret_cast<const void*>(&conv_wgrad3_bf16x3_kernel<KCW, true, false, false>),
                            reinterpret_cast<const void*>(&conv_wgrad3_bf16x3_kernel<KCW, false, true, false>),
                            reinterpret_cast<const void*>(&conv_wgrad3_bf16x3_kernel<KCW, true, true, false>),
                            reinterpret_cast<const void*>(&conv_wgrad3_bf16x3_kernel<KCW, false, true, true>),
                            reinterpret_cast<const void*>(&conv_wgrad3_bf16x3_kernel<KCW, true, true, true>)};
      if (ffsr_allow_dynamic_lds(fns, 6, LDS, &attr_set) != FFSR_OK) return FFSR_ELAUNCH;
    }
#define FFSR_WG(E, X, Y) FFSR_LAUNCH((conv_wgrad3_bf16x3_kernel<KCW, E, X, Y>), grid, dim3(512), LDS, st, a)
    if (ypl) { if (wide_full) FFSR_WG(false, true, true); else FFSR_WG(true, true, true); }
    else if (xpl) { if (wide_full) FFSR_WG(false, true, false); else FFSR_WG(true, true, false); }
    else { if (wide_full) FFSR_WG(false, false, false); else FFSR_WG(true, false, false); }
#undef FFSR_WG
  } else if (row3) {
    if (tn == 32 && tc == 32) FFSR_LAUNCH((conv_wgrad3_kernel<1, 1, 64>), grid, dim3(256), 0, st, a);
    else if (tn == 32 && tc == 64) FFSR_LAUNCH((conv_wgrad3_kernel<1, 2, 16>), grid, dim3(256), 0, st, a);
    else if (tn == 64 && tc == 32) FFSR_LAUNCH((conv_wgrad3_kernel<2, 1, 16>), grid, dim3(256), 0, st, a);
    else if (tn == 64 && tc == 64) FFSR_LAUNCH((conv_wgrad3_kernel<2, 2, 16>), grid, dim3(256), 0, st, a);
    else if (tn == 32 && tc == 128) FFSR_LAUNCH((conv_wgrad3_kernel<1, 4, 16>), grid, dim3(256), 0, st, a);
    else FFSR_LAUNCH((conv_wgrad3_kernel<4, 1, 16>), grid, dim3(256), 0, st, a);
  } else if (tn == 32 && tc == 32) launch<1, 1, 1, 1>(a, grid, st);
  else if (tn == 32 && tc == 64) launch<1, 2, 1, 1>(a, grid, st);
  else if (tn == 64 && tc == 32) launch<2, 1, 1, 1>(a, grid, st);
  else if (tn == 64 && tc == 64) launch<2, 2, 1, 1>(a, grid, st);
  else if (tn == 32 && tc == 128) launch<1, 4, 1, 1>(a, grid, st);
  else if (tn == 128 && tc == 32) launch<4, 1, 1, 1>(a, grid, st);
  else if (tn == 64 && tc == 128) launch<2, 2, 1, 2>(a, grid, st);
  else if (tn == 128 && tc == 64) launch<2, 2, 2, 1>(a, grid, st);
  else launch<2, 2, 2, 2>(a, grid, st);
  const long long nw = (long long)T * N * Cin;
  FFSR_LAUNCH(wgrad_finish_kernel, dim3((unsigned)((nw + 31) / 32)), dim3(256), 0, st, partial, (int)S, T, N, Cin, dw);
  if (dbias)
    FFSR_LAUNCH(wgrad_finish_kernel, dim3((unsigned)((N + 31) / 32)), dim3(256), 0, st, a.bias_part, (int)S, 1, N, 1, dbias);
  return ffsr_launch_status();
}

extern "C" int ffsr_conv_wgrad_f32(const float* x, int ldx, const float* dy, int ldy, float* dw, float* dbias, float* partial,
                                   long long partial_floats, int B, int H, int W, int Cin, int N, int KH, int KW, int pad_h,
                                   int pad_w, void* stream) {
  return wgrad_impl(x, ldx, dy, ldy, dw, dbias, partial, partial_floats, B, H, W, Cin, N, KH, KW, pad_h, pad_w, false, stream);
}

// The same contract with split-bf16 products (hi*hi + hi*lo + lo*hi, fp32 accumulate: the arithmetic of ffsr_conv2d_bf16x3) where
// a kernel for the shape exists (3-wide layers with N and Cin multiples of 128); other shapes take the exact fp32 path.
extern "C" int ffsr_conv_wgrad_bf16x3(const float* x, int ldx, const float* dy, int ldy, float* dw, float* dbias, float* partial,
                                      long long partial_floats, int B, int H, int W, int Cin, int N, int KH, int KW, int pad_h,
                                      int pad_w, void* stream) {
  return wgrad_impl(x, ldx, dy, ldy, dw, dbias, partial, partial_floats, B, H, W, Cin, N, KH, KW, pad_h, pad_w, true, stream);
}

// ffsr_conv_wgrad_bf16x3 with X given as the bf16 hi / lo planes [B*H*W, ldp] the planes GEMM consumed in the forward pass (ldp % 32
// == 0, pad channels zero): the rows go to LDS as they are, no fp32 copy of the activation has to exist.  Only the shapes the
// bf16 kernel takes (3-wide, N and Cin multiples of 4, one of them > 64, both >= 32, >= 65536 pixels unless multiples of 128);
// FFSR_EINVAL otherwise.  dY: the fp32 map dy, or (dy NULL) the planes dy_hi / dy_lo [B*H*W, ldq] of ffsr_act_bwd_planes_f32.
extern "C" int ffsr_conv_wgrad_bf16x3_planes(const void* x_hi, const void* x_lo, int ldp, const float* dy, int ldy, const void* dy_hi,
                                             const void* dy_lo, int ldq, float* dw, float* dbias, float* partial,
                                             long long partial_floats, int B, int H, int W, int Cin, int N, int KH, int KW, int pad_h,
                                             int pad_w, void* stream) {
  return wgrad_impl(nullptr, 0, dy, ldy, dw, dbias, partial, partial_floats, B, H, W, Cin, N, KH, KW, pad_h, pad_w, true, stream,
                    x_hi, x_lo, ldp, dy_hi, dy_lo, ldq);
}
