// Implicit-GEMM convolution / token GEMM on the fp32-input MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
//   out[m, n] = epi( sum_k A[m, k] * Wt[n, k] )        m = output pixel (b, oy, ox), n = output channel
//   A[m, (ky, kx, ci)] = in[b, oy*stride - pad + ky, ox*stride - pad + kx, ci]   (NHWC, zero padded)
//
// This one kernel is K5/K6/K12/K15 of SURVEY.md section 2.3: every nn.Linear / 1x1 conv (KH=KW=1) and every
// dense kxk conv of the four experts and the fusion net.  Numerics: the f32 MFMA is an exact k-ordered
// fmaf chain, i.e. the same arithmetic class as the reference's fp32 CPU path.
//
// Tiling: 256 threads = 4 waves; block tile BM x BN, K step 32; each wave owns TM x TN tiles of 32x32.
// A and B tiles are staged global -> registers -> LDS (double buffered, one barrier per K step); fragments are
// read as ds_read_b128 with a 36-float row stride (conflict-free for the b128 lane groups).  Inside a group
// of 8 k's lane-half h holds k = 8j+4h..8j+4h+3, MFMA step s contracts k = 8j+s and 8j+4+s (both operands use
// the same permutation, so the sum is unchanged).  blockIdx -> tile mapping is XCD-aware: the tiles of one
// XCD are consecutive (same A rows, neighbouring N tiles) so A is fetched into one L2 only.
#include "ffsr_common.h"

namespace {

constexpr int BK = 32;
constexpr int LSTR = 36;  // LDS row stride in floats (36/4 = 9 odd -> b128 reads conflict-free)

struct ConvArgs {
  const float* in;
  const float* wgt;
  const float* bias;
  float* out;
  const float* res;
  const float* cvec;
  const float* rvec;
  const float* akscale;
  int B, H, W, Cin, ldi;
  int N, Ho, Wo, ldo, ldr, ldw;
  int KH, KW, stride, pad_h, pad_w;
  int act;
  float slope, cscale, rscale;
  int shuffle;
  int M, Ktot, akrows;  // akrows = rows (pixels) per akscale batch
};

template <int BM, int BN, int WAVES_M, int WAVES_N, bool HAS_AK, int STAGES>
__global__ __launch_bounds__(256) void conv_gemm_kernel(ConvArgs p) {
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int AR = BM / 32, BR = BN / 32;  // rows per thread in the loaders
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  __shared__ __attribute__((aligned(16))) float smem[STAGES][(BM + BN) * LSTR];

  // ---- XCD-aware tile id (bijective for any grid size)
  const int nwg = gridDim.x;
  const int orig = blockIdx.x;
  const int q = nwg >> 3, rr = nwg & 7, xcd = orig & 7;
  const int tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (orig >> 3);
  const int ntn = (p.N + BN - 1) / BN;
  const int m0 = (tile / ntn) * BM;
  const int n0 = (tile % ntn) * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int kofs = (tid & 7) * 4;
  const int rbase = tid >> 3;

  // ---- per-row state of the A loader
  int a_iy0[AR], a_ix0[AR], a_pix[AR];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    int m = m0 + rbase + 32 * i;
    if (m < p.M) {
      int b = m / HoWo, rem = m - b * HoWo;
      int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      a_iy0[i] = oy * p.stride - p.pad_h;
      a_ix0[i] = ox * p.stride - p.pad_w;
      a_pix[i] = b * p.H * p.W;
    } else {
      a_iy0[i] = -(1 << 28);  // always out of bounds
      a_ix0[i] = 0;
      a_pix[i] = 0;
    }
  }

  floatx4 a_reg[AR], b_reg[BR];
  float a_ok[AR], b_ok[BR];  // 1.0 / 0.0 masks (see load_tiles)
  const bool is1x1 = (p.KH * p.KW == 1);

  auto load_tiles = [&](int kt) {
    const int k = kt * BK + kofs;
    const bool kval = k < p.Ktot;
    int ky = 0, kx = 0, ci = k;
    if (!is1x1) {
      int tap = k / p.Cin;
      ci = k - tap * p.Cin;
      ky = tap / p.KW;
      kx = tap - ky * p.KW;
    }
    // NOTE: every load is issued unconditionally from a clamped (always valid) address and zeroed afterwards by a
    // MULTIPLY with a 0/1 mask: a load under `if (ok)` -- or a select the compiler can sink the load into -- makes
    // hipcc branch around it and wait vmcnt(0) per load, which serialises the whole prefetch
    // (cdna_hip_programming.md section 5, trap (c)).  Clamped reads hit offset 0 of a live tensor (finite data).
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int yy = a_iy0[i] + ky, xx = a_ix0[i] + kx;
      const bool ok = kval && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
      a_ok[i] = ok ? 1.f : 0.f;
      const size_t off = ok ? (size_t)(a_pix[i] + yy * p.W + xx) * p.ldi + ci : 0;
      a_reg[i] = *reinterpret_cast<const floatx4*>(p.in + off);
    }
    if (HAS_AK) {
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const int m = min(m0 + rbase + 32 * i, p.M - 1);
        const size_t off = (size_t)(m / p.akrows) * p.Ktot + (kval ? k : 0);
        a_reg[i] *= *reinterpret_cast<const floatx4*>(p.akscale + off);
      }
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      const int n = n0 + rbase + 32 * i;
      const bool ok = kval && n < p.N;
      b_ok[i] = ok ? 1.f : 0.f;
      const size_t off = ok ? (size_t)n * p.ldw + k : 0;
      b_reg[i] = *reinterpret_cast<const floatx4*>(p.wgt + off);
    }
  };
  auto store_tiles = [&](int buf) {
    float* As = smem[buf];
    float* Bs = As + BM * LSTR;
#pragma unroll
    for (int i = 0; i < AR; ++i) *reinterpret_cast<floatx4*>(As + (rbase + 32 * i) * LSTR + kofs) = a_reg[i] * a_ok[i];
#pragma unroll
    for (int i = 0; i < BR; ++i) *reinterpret_cast<floatx4*>(Bs + (rbase + 32 * i) * LSTR + kofs) = b_reg[i] * b_ok[i];
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int wrow = (wave / WAVES_N) * WM, wcol = (wave % WAVES_N) * WN;
  const int r = lane & 31, h = lane >> 5;
  const int nk = (p.Ktot + BK - 1) / BK;

  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tiles(kt + 1);
    const float* As = smem[STAGES == 2 ? (kt & 1) : 0];
    const float* Bs = As + BM * LSTR;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      floatx4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = *reinterpret_cast<const floatx4*>(As + (wrow + i * 32 + r) * LSTR + 8 * j + 4 * h);
#pragma unroll
      for (int i = 0; i < TN; ++i)
        bf[i] = *reinterpret_cast<const floatx4*>(Bs + (wcol + i * 32 + r) * LSTR + 8 * j + 4 * h);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int jj = 0; jj < TN; ++jj)
            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[jj][s], acc[i][jj], 0, 0, 0);
    }
    if (STAGES == 1) __syncthreads();  // all waves done reading the single buffer
    if (kt + 1 < nk) store_tiles(STAGES == 2 ? ((kt + 1) & 1) : 0);
    __syncthreads();
  }

  // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
  for (int jn = 0; jn < TN; ++jn) {
    const int n = n0 + wcol + jn * 32 + r;
    if (n >= p.N) continue;
    const float bia = p.bias ? p.bias[n] : 0.f;
    const float cs = (p.cvec ? p.cvec[n] : 1.f) * p.cscale;
    const float rs = (p.rvec ? p.rvec[n] : 1.f) * p.rscale;
#pragma unroll
    for (int im = 0; im < TM; ++im) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wrow + im * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (m >= p.M) continue;
        float v = ffsr_act(acc[im][jn][e] + bia, p.act, p.slope) * cs;
        size_t opix;
        int oc = n;
        if (p.shuffle) {
          int b = m / HoWo, rem = m - b * HoWo;
          int oy = rem / p.Wo, ox = rem - oy * p.Wo;
          oc = n >> 2;
          opix = ((size_t)b * 2 * p.Ho + 2 * oy + ((n >> 1) & 1)) * (2 * p.Wo) + 2 * ox + (n & 1);
        } else {
          opix = (size_t)m;
        }
        if (p.res) v += p.res[opix * p.ldr + oc] * rs;
        p.out[opix * p.ldo + oc] = v;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Split-bf16 ("bf16x3") variant: every fp32 operand is split on the fly into hi = bf16(x), lo = bf16(x - hi) and the
// product is evaluated as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  The dropped lo*lo
// term and the split residue bound the per-product relative error by ~3 * 2^-18 (1.1e-5) -- 3 MFMAs at 16x the f32
// MFMA rate.  Same tiling as the exact kernel; LDS holds bf16 hi/lo planes in unpadded 64-byte rows with the planes
// kernel's XOR chunk swizzle (round 1 used 80-byte rows: conflict-free for the b128 fragment reads but not for the staging
// stores -- 33 % of its LDS cycles were bank conflicts, profiles/r02_gemm_pmc.md).  Entry point: ffsr_conv2d_bf16x3 (the engine's default GEMM mode).
// ---------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef unsigned uintx2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split4(const floatx4 v, uintx2& hi, uintx2& lo) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const floatx2 x = {v[2 * i], v[2 * i + 1]};
    const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
    const floatx2 r = {x[0] - __builtin_bit_cast(float, h << 16), x[1] - __builtin_bit_cast(float, h & 0xffff0000u)};
    hi[i] = h;
    lo[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The kernel (the default data path).  Compared with the exact kernel's loader:
//  * weights arrive PRE-SPLIT (bf16 hi / lo planes [Npad, Kpad], zero padded to multiples of BN / 32 at pack time):
//    the B tile is 16-byte loads straight into LDS rows, no checks, no VALU;
//  * out-of-image taps / rows read a caller-provided zero page instead of being masked: no multiplies, no NaN hazard;
//  * A addressing is hoisted: per row one base offset + a bitmask of valid taps; per K step one table lookup
//    (tap -> pixel offset, in LDS) -- the 1x1 (token GEMM) case degenerates to base + k;
//  * BN is 64 or 128 (one A pass serves twice the columns).
// ---------------------------------------------------------------------------------------------------------------
struct ConvArgs3 {
  const float* in;
  const unsigned short* whi;
  const unsigned short* wlo;
  const float* zeros;  // >= 64 bytes of zeros
  const float* bias;
  float* out;
  const float* res;
  const float* cvec;
  const float* rvec;
  const float* akscale;
  int B, H, W, Cin, ldi;
  int N, Ho, Wo, ldo, ldr, ldw;  // ldw = padded K (elements) of the weight planes
  int KH, KW, stride, pad_h, pad_w;
  int act;
  float slope, cscale, rscale;
  int shuffle;
  int M, Ktot, akrows;
};

// TERMS 3: hi*hi + hi*lo + lo*hi; TERMS 1: plain bf16 operands (hi*hi only: no lo planes staged, one MFMA per product)
template <int BN, bool HAS_AK, int TERMS>
__global__ __launch_bounds__(256) void conv_gemm_bf16x3_v3_kernel(ConvArgs3 p) {
  // BN = 64 / 128: 2 x 2 waves, wave tile 64 x BN/2.  BN = 32 (thin convs, N <= 32: the 3- / 16- / 32-channel heads of the
  // fusion net at HR resolution): 4 x 1 waves, wave tile 32 x 32 -- half the MFMA work of padding N to 64.
  constexpr int BM = 128, WAVES_N = BN >= 64 ? 2 : 1, WAVES_M = 4 / WAVES_N;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int AR = BM / 32, BRW = (BN + 63) / 64;
  // 64-byte rows, no padding: the 16-byte chunk c of row r lives in slot c ^ ((r >> 2) & 3) (the planes kernel's scheme):
  // conflict-free for the 8- / 16-byte staging stores AND for the ds_read_b128 fragment reads
  constexpr int RS = 64;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (BM + BN) * RS];
  __shared__ int tapoff[32];
  unsigned char* const Ahi = smem;
  unsigned char* const Alo = Ahi + BM * RS;
  unsigned char* const Bhi = Alo + BM * RS;
  unsigned char* const Blo = Bhi + BN * RS;

  const int nwg = gridDim.x;
  const int orig = blockIdx.x;
  const int q = nwg >> 3, rr = nwg & 7, xcd = orig & 7;
  const int tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (orig >> 3);
  const int ntn = (p.N + BN - 1) / BN;
  const int m0 = (tile / ntn) * BM;
  const int n0 = (tile % ntn) * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int kofs = (tid & 7) * 4;
  const int rbase = tid >> 3;
  const int ntap = p.KH * p.KW;
  if (tid < ntap) tapoff[tid] = ((tid / p.KW) * p.W + (tid % p.KW)) * p.ldi;

  // ---- per-row base offset (element index of tap (0,0)) and bitmask of in-image taps
  long long a_base[AR];
  unsigned a_mask[AR];
  const int HoWo = p.Ho * p.Wo;
  const bool plain = ntap == 1 && p.stride == 1 && p.pad_h == 0 && p.pad_w == 0;   // token GEMM: row m of A is pixel m
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = m0 + rbase + 32 * i;
    a_base[i] = 0;
    a_mask[i] = 0;
    if (m < p.M) {
      if (plain) {
        a_base[i] = (long long)m * p.ldi;
        a_mask[i] = 1u;
      } else {
        const int b = m / HoWo, rem = m - b * HoWo;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        const int iy0 = oy * p.stride - p.pad_h, ix0 = ox * p.stride - p.pad_w;
        a_base[i] = ((long long)(b * p.H + iy0) * p.W + ix0) * p.ldi;
        unsigned mk = 0;
        for (int t = 0; t < ntap; ++t) {
          const int yy = iy0 + t / p.KW, xx = ix0 + t % p.KW;
          if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) mk |= 1u << t;
        }
        a_mask[i] = mk;
      }
    }
  }
  __syncthreads();  // tapoff visible

  // Tried and rejected on MI355X (tools/gemm_bench.py): a two-K-step-ahead register prefetch (+24 VGPRs -> 2 instead
  // of 4 waves per SIMD: 25-30 % slower -- occupancy hides the load latency better than a deeper per-wave prefetch);
  // BN = 128 (2 waves per SIMD: slower on every shape); an A-stationary "row strip" kernel reading the weight
  // fragments straight from L2 (fragment-shaped 32-byte loads saturate the texture-address path: 20-50 % slower).
  floatx4 a_reg[AR];
  floatx4 bh_reg[BRW], bl_reg[BRW];   // 8 bf16 each
  // per-(batch, k) A scaling (NAFNet's SCA): workgroup-uniform test whether the whole 128-row tile lies in one batch image
  const bool ak_uniform = HAS_AK && (m0 / p.akrows) == (min(m0 + BM, p.M) - 1) / p.akrows;
  const size_t ak_base = HAS_AK ? (size_t)(m0 / p.akrows) * p.Ktot : 0;
  const int brow = tid >> 2, bseg = (tid & 3) * 8;  // B loader: row, first k of its 8-element segment
  auto load_tiles = [&](int kt) {
    const int k = kt * BK + kofs;
    int tap = 0, ci = k;
    if (ntap > 1) {
      tap = k / p.Cin;
      ci = k - tap * p.Cin;
    }
    const bool kval = k < p.Ktot;
    const int toff = tapoff[kval ? tap : 0] + ci;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const bool ok = kval && ((a_mask[i] >> tap) & 1u);
      const float* src = ok ? p.in + (a_base[i] + toff) : p.zeros;
      a_reg[i] = *reinterpret_cast<const floatx4*>(src);
    }
    if (HAS_AK) {
      if (ak_uniform) {    // all rows of this tile in one batch image (always at batch 1): one scale vector per K step
        const floatx4 sc = *reinterpret_cast<const floatx4*>(p.akscale + ak_base + (kval ? k : 0));
#pragma unroll
        for (int i = 0; i < AR; ++i) a_reg[i] *= sc;
      } else {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
          const int m = min(m0 + rbase + 32 * i, p.M - 1);
          const size_t off = (size_t)(m / p.akrows) * p.Ktot + (kval ? k : 0);
          a_reg[i] *= *reinterpret_cast<const floatx4*>(p.akscale + off);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < BRW; ++j) {
      if (BN < 64 && brow >= BN) break;                                          // BN = 32: half the loader threads idle
      const size_t off = (size_t)(n0 + brow + 64 * j) * p.ldw + kt * BK + bseg;   // planes are padded: always valid
      bh_reg[j] = *reinterpret_cast<const floatx4*>(p.whi + off);
      if constexpr (TERMS == 3) bl_reg[j] = *reinterpret_cast<const floatx4*>(p.wlo + off);
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      uintx2 hi, lo;
      split4(a_reg[i], hi, lo);
      const int o = (rbase + 32 * i) * RS + ((((kofs >> 3) ^ ((rbase >> 2) & 3)) << 4) | ((kofs * 2) & 15));
      *reinterpret_cast<uintx2*>(Ahi + o) = hi;
      if constexpr (TERMS == 3) *reinterpret_cast<uintx2*>(Alo + o) = lo;
    }
#pragma unroll
    for (int j = 0; j < BRW; ++j) {
      if (BN < 64 && brow >= BN) break;
      const int o = (brow + 64 * j) * RS + (((tid & 3) ^ ((brow >> 2) & 3)) << 4);
      *reinterpret_cast<floatx4*>(Bhi + o) = bh_reg[j];
      if constexpr (TERMS == 3) *reinterpret_cast<floatx4*>(Blo + o) = bl_reg[j];
    }
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int wrow = (wave / WAVES_N) * WM, wcol = (wave % WAVES_N) * WN;
  const int r = lane & 31, h = lane >> 5;
  const int nk = (p.Ktot + BK - 1) / BK;

  load_tiles(0);
  store_tiles();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tiles(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ko = ((2 * ks + h) ^ ((r >> 2) & 3)) << 4;
      bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = *reinterpret_cast<const bf16x8*>(Ahi + (wrow + i * 32 + r) * RS + ko);
        if constexpr (TERMS == 3) al[i] = *reinterpret_cast<const bf16x8*>(Alo + (wrow + i * 32 + r) * RS + ko);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const bf16x8*>(Bhi + (wcol + j * 32 + r) * RS + ko);
        if constexpr (TERMS == 3) bl[j] = *reinterpret_cast<const bf16x8*>(Blo + (wcol + j * 32 + r) * RS + ko);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (TERMS == 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    if (kt + 1 < nk) store_tiles();
    __syncthreads();
  }

  // ---- epilogue.  Lean path (no PixelShuffle): one pointer per lane, compile-time row offsets, the activation switch
  // hoisted out of the element loop; rows are bounds-checked only in the last row tile.
  const bool full_rows = m0 + BM <= p.M;
  if constexpr (TN == 2) {
    if (p.shuffle == 3) {
      // SimpleGate fused into the store (nafnet_arch.py:21-24, x1 * x2 over the channel halves): the host packs the weight
      // rows so that a wave's two 32-column tiles hold x1[j..j+31] and x2[j..j+31]; the product is lane-local.
      // out / res / cvec / rvec have N / 2 columns.
      const int na = n0 + wcol + r, no = (n0 + wcol) / 2 + r;
      if (na >= p.N) return;   // (N % 64 == 0: whole pairs)
      const float ba = p.bias ? p.bias[na] : 0.f, bb = p.bias ? p.bias[na + 32] : 0.f;
      const float cs = (p.cvec ? p.cvec[no] : 1.f) * p.cscale;
      const float rs = (p.rvec ? p.rvec[no] : 1.f) * p.rscale;
      const bool gvec = ((p.ldo & 3) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0) &&
                        (!p.res || ((p.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(p.res) & 15) == 0)) &&
                        (!p.rvec || (reinterpret_cast<uintptr_t>(p.rvec) & 15) == 0);
#pragma unroll
      for (int im = 0; im < TM; ++im) {
        const int mrow = m0 + wrow + im * 32 + 4 * h;
        if (gvec) {      // same 16-byte store path as the plain epilogue below: transpose the 32 gated columns through LDS
          float* T = reinterpret_cast<float*>(smem) + wave * (32 * 36);
#pragma unroll
          for (int e = 0; e < 16; ++e)
            T[((e & 3) + 8 * (e >> 2) + 4 * h) * 36 + r] = (acc[im][0][e] + ba) * (acc[im][1][e] + bb) * cs;
          const int erow = lane >> 2, ecol = (lane & 3) * 8;
          const int nn = (n0 + wcol) / 2 + ecol;
          floatx4 rs0 = {p.rscale, p.rscale, p.rscale, p.rscale}, rs1 = rs0;
          if (p.res && p.rvec) {
            rs0 *= *reinterpret_cast<const floatx4*>(p.rvec + nn);
            rs1 *= *reinterpret_cast<const floatx4*>(p.rvec + nn + 4);
          }
#pragma unroll
          for (int pass = 0; pass < 2; ++pass) {
            const int row = pass * 16 + erow;
            const int m = m0 + wrow + im * 32 + row;
            if (m >= p.M) continue;
            const float* tp = T + row * 36 + ecol;
            floatx4 o0 = *reinterpret_cast<const floatx4*>(tp), o1 = *reinterpret_cast<const floatx4*>(tp + 4);
            if (p.res) {
              const float* rp2 = p.res + (size_t)m * p.ldr + nn;
              o0 += *reinterpret_cast<const floatx4*>(rp2) * rs0;
              o1 += *reinterpret_cast<const floatx4*>(rp2 + 4) * rs1;
            }
            float* op2 = p.out + (size_t)m * p.ldo + nn;
            *reinterpret_cast<floatx4*>(op2) = o0;
            *reinterpret_cast<floatx4*>(op2 + 4) = o1;
          }
          continue;
        }
        float* op = p.out + (size_t)mrow * p.ldo + no;
        const float* rp = p.res ? p.res + (size_t)mrow * p.ldr + no : nullptr;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int dr = (e & 3) + 8 * (e >> 2);
          if (full_rows || mrow + dr < p.M) {
            float o = (acc[im][0][e] + ba) * (acc[im][1][e] + bb) * cs;
            if (rp) o += rp[dr * p.ldr] * rs;
            op[dr * p.ldo] = o;
          }
        }
      }
      return;
    }
  }
  // wave-uniform: may whole 32-column tiles be written with 16-byte stores?
  const bool vec_ok = ((p.ldo & 3) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0) &&
                      (!p.res || ((p.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(p.res) & 15) == 0)) &&
                      (!p.rvec || (reinterpret_cast<uintptr_t>(p.rvec) & 15) == 0);
#pragma unroll
  for (int jn = 0; jn < TN; ++jn) {
    const int n = n0 + wcol + jn * 32 + r;
    if (n >= p.N) continue;
    const float bia = p.bias ? p.bias[n] : 0.f;
    const float cs = (p.cvec ? p.cvec[n] : 1.f) * p.cscale;
    const float rs = (p.rvec ? p.rvec[n] : 1.f) * p.rscale;
#pragma unroll
    for (int im = 0; im < TM; ++im) {
      floatx16 v = acc[im][jn];
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] += bia;
      switch (p.act) {   // wave-uniform
        case FFSR_ACT_NONE: break;
        case FFSR_ACT_GELU:
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752440f));
          break;
        default:
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] = ffsr_act(v[e], p.act, p.slope);
      }
      const int mrow = m0 + wrow + im * 32 + 4 * h;   // + (e&3) + 8*(e>>2)
      if (!p.shuffle && vec_ok && n0 + wcol + jn * 32 + 32 <= p.N) {
        // 16-byte stores: the 32x32 tile is transposed through this wave's scratch (the staging LDS is idle now) so that a
        // lane owns 8 consecutive columns of one row (the accumulator layout has one column per lane = 4-byte stores;
        // A/B in round 2: the 340x510 step 368.6 -> 362.7 ms)
        float* T = reinterpret_cast<float*>(smem) + wave * (32 * 36);
#pragma unroll
        for (int e = 0; e < 16; ++e) T[((e & 3) + 8 * (e >> 2) + 4 * h) * 36 + r] = v[e] * cs;
        const int erow = lane >> 2, ecol = (lane & 3) * 8;
        const int nn = n0 + wcol + jn * 32 + ecol;
        floatx4 rs0 = {p.rscale, p.rscale, p.rscale, p.rscale}, rs1 = rs0;
        if (p.res && p.rvec) {
          rs0 *= *reinterpret_cast<const floatx4*>(p.rvec + nn);
          rs1 *= *reinterpret_cast<const floatx4*>(p.rvec + nn + 4);
        }
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
          const int row = pass * 16 + erow;
          const int m = m0 + wrow + im * 32 + row;
          if (m >= p.M) continue;
          const float* tp = T + row * 36 + ecol;
          floatx4 o0 = *reinterpret_cast<const floatx4*>(tp), o1 = *reinterpret_cast<const floatx4*>(tp + 4);
          if (p.res) {
            const float* rp = p.res + (size_t)m * p.ldr + nn;
            o0 += *reinterpret_cast<const floatx4*>(rp) * rs0;
            o1 += *reinterpret_cast<const floatx4*>(rp + 4) * rs1;
          }
          float* op = p.out + (size_t)m * p.ldo + nn;
          *reinterpret_cast<floatx4*>(op) = o0;
          *reinterpret_cast<floatx4*>(op + 4) = o1;
        }
        continue;
      }
      if (!p.shuffle) {
        float* op = p.out + (size_t)mrow * p.ldo + n;
        const float* rp = p.res ? p.res + (size_t)mrow * p.ldr + n : nullptr;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int dr = (e & 3) + 8 * (e >> 2);
          if (full_rows || mrow + dr < p.M) {
            float o = v[e] * cs;
            if (rp) o += rp[dr * p.ldr] * rs;
            op[dr * p.ldo] = o;
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = mrow + (e & 3) + 8 * (e >> 2);
          if (m >= p.M) continue;
          const int b = m / HoWo, rem = m - b * HoWo;
          const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
          const int oc = n >> 2;
          const size_t opix = ((size_t)b * 2 * p.Ho + 2 * oy + ((n >> 1) & 1)) * (2 * p.Wo) + 2 * ox + (n & 1);
          float o = v[e] * cs;
          if (p.res) o += p.res[opix * p.ldr + oc] * rs;
          p.out[opix * p.ldo + oc] = o;
        }
      }
    }
  }
}

template <int BN>
int launch_v3(const ConvArgs3& a, hipStream_t st) {
  int tiles = ((a.M + 127) / 128) * ((a.N + BN - 1) / BN);
  if (g_ffsr_gemm_terms == 1) {
    if (a.akscale) FFSR_LAUNCH((conv_gemm_bf16x3_v3_kernel<BN, true, 1>), dim3(tiles), dim3(256), 0, st, a);
    else FFSR_LAUNCH((conv_gemm_bf16x3_v3_kernel<BN, false, 1>), dim3(tiles), dim3(256), 0, st, a);
  } else if (a.akscale)
    FFSR_LAUNCH((conv_gemm_bf16x3_v3_kernel<BN, true, 3>), dim3(tiles), dim3(256), 0, st, a);
  else
    FFSR_LAUNCH((conv_gemm_bf16x3_v3_kernel<BN, false, 3>), dim3(tiles), dim3(256), 0, st, a);
  return ffsr_launch_status();
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES>
int launch(const ConvArgs& a, hipStream_t st) {
  int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  if (a.akscale)
    FFSR_LAUNCH((conv_gemm_kernel<BM, BN, WAVES_M, WAVES_N, true, STAGES>), dim3(tiles), dim3(256), 0, st, a);
  else
    FFSR_LAUNCH((conv_gemm_kernel<BM, BN, WAVES_M, WAVES_N, false, STAGES>), dim3(tiles), dim3(256), 0, st, a);
  return ffsr_launch_status();
}

}  // namespace

// See include/ffsr.h for the contract.
extern "C" int ffsr_conv2d_f32(const float* in, const float* wgt, const float* bias, float* out, const float* res,
                               const float* cvec, const float* rvec, const float* akscale, int B, int H, int W,
                               int Cin, int ldi, int N, int ldo, int ldr, int KH, int KW, int stride, int pad_h,
                               int pad_w, int act, float slope, float cscale, float rscale, int shuffle,
                               int akrows, int tile_hint, void* stream) {
  FFSR_CHECK(in && wgt && out);
  FFSR_CHECK(B > 0 && H > 0 && W > 0 && N > 0 && KH > 0 && KW > 0 && stride > 0);
  FFSR_CHECK(Cin > 0 && (Cin & 3) == 0 && (ldi & 3) == 0 && ldi >= Cin);
  FFSR_CHECK(((uintptr_t)in & 15) == 0 && ((uintptr_t)wgt & 15) == 0);
  FFSR_CHECK(shuffle == 0 || (shuffle == 2 && (N & 3) == 0));
  FFSR_CHECK(!akscale || (KH * KW == 1 && akrows > 0 && ((uintptr_t)akscale & 15) == 0));
  ConvArgs a;
  a.in = in; a.wgt = wgt; a.bias = bias; a.out = out; a.res = res; a.cvec = cvec; a.rvec = rvec; a.akscale = akscale;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.ldi = ldi; a.N = N;
  a.Ho = (H + 2 * pad_h - KH) / stride + 1;
  a.Wo = (W + 2 * pad_w - KW) / stride + 1;
  FFSR_CHECK(a.Ho > 0 && a.Wo > 0);
  a.ldo = ldo; a.ldr = ldr; a.KH = KH; a.KW = KW; a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w;
  a.act = act; a.slope = slope; a.cscale = cscale; a.rscale = rscale; a.shuffle = shuffle;
  a.Ktot = KH * KW * Cin; a.ldw = a.Ktot; a.akrows = akrows > 0 ? akrows : 1;
  long long M = (long long)B * a.Ho * a.Wo;
  FFSR_CHECK(M < (1ll << 31) && (long long)B * H * W < (1ll << 31));
  a.M = (int)M;
  hipStream_t st = (hipStream_t)stream;
  // tile choice (measured with tools/gemm_bench.py on MI355X): the 128x64 tile with ONE LDS stage (27 KB -> 4
  // workgroups per CU) beats the double-buffered tiles on every shape of this path; 256x32 serves N <= 32 and
  // 64x64 the tiny [B, C] channel-attention GEMMs.  tile_hint overrides (1/2/3/4 double-buffered 128x128 / 128x64 /
  // 256x32 / 64x64; 11/12/13 their single-stage forms).
  int choice = tile_hint;
  if (choice == 0) {
    if (N <= 32) choice = 13;
    else choice = 12;
    if (a.M <= 64 * 24 && N > 32) choice = 4;
  }
  switch (choice) {
    case 1: return launch<128, 128, 2, 2, 2>(a, st);
    case 2: return launch<128, 64, 2, 2, 2>(a, st);
    case 3: return launch<256, 32, 4, 1, 2>(a, st);
    case 4: return launch<64, 64, 2, 2, 2>(a, st);
    case 11: return launch<128, 128, 2, 2, 1>(a, st);   // single LDS stage: more workgroups per CU
    case 12: return launch<128, 64, 2, 2, 1>(a, st);
    case 13: return launch<256, 32, 4, 1, 1>(a, st);
    default: return FFSR_EINVAL;
  }
}

// Split-bf16 convolution with pre-split weights: see include/ffsr.h.
int g_ffsr_gemm_terms = 3;

// See include/ffsr.h.
extern "C" int ffsr_set_gemm_terms(int terms) {
  FFSR_CHECK(terms == 1 || terms == 3);
  g_ffsr_gemm_terms = terms;
  return FFSR_OK;
}

extern "C" int ffsr_conv2d_bf16x3(const float* in, const void* wgt_hi, const void* wgt_lo, int ldw, int n_rows_padded,
                                  const float* zeros, const float* bias, float* out, const float* res, const float* cvec,
                                  const float* rvec, const float* akscale, int B, int H, int W, int Cin, int ldi, int N,
                                  int ldo, int ldr, int KH, int KW, int stride, int pad_h, int pad_w, int act, float slope,
                                  float cscale, float rscale, int shuffle, int akrows, int bn, void* stream) {
  FFSR_CHECK(in && wgt_hi && wgt_lo && zeros && out);
  FFSR_CHECK(B > 0 && H > 0 && W > 0 && N > 0 && KH > 0 && KW > 0 && KH * KW <= 32 && stride > 0);
  FFSR_CHECK(Cin > 0 && (Cin & 3) == 0 && (ldi & 3) == 0 && ldi >= Cin);
  FFSR_CHECK(((uintptr_t)in & 15) == 0 && ((uintptr_t)wgt_hi & 15) == 0 && ((uintptr_t)wgt_lo & 15) == 0 &&
             ((uintptr_t)zeros & 15) == 0);
  FFSR_CHECK(shuffle == 0 || (shuffle == 2 && (N & 3) == 0) || (shuffle == 3 && (N & 63) == 0 && bn == 128 && act == FFSR_ACT_NONE));
  FFSR_CHECK(!akscale || (KH * KW == 1 && akrows > 0 && ((uintptr_t)akscale & 15) == 0));
  FFSR_CHECK(bn == 32 || bn == 64 || bn == 128);
  const int Ktot = KH * KW * Cin;
  FFSR_CHECK((ldw & 31) == 0 && ldw >= Ktot && (n_rows_padded % 128) == 0 && n_rows_padded >= N);
  ConvArgs3 a;
  a.in = in; a.whi = (const unsigned short*)wgt_hi; a.wlo = (const unsigned short*)wgt_lo; a.zeros = zeros; a.bias = bias;
  a.out = out; a.res = res; a.cvec = cvec; a.rvec = rvec; a.akscale = akscale;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.ldi = ldi; a.N = N;
  a.Ho = (H + 2 * pad_h - KH) / stride + 1;
  a.Wo = (W + 2 * pad_w - KW) / stride + 1;
  FFSR_CHECK(a.Ho > 0 && a.Wo > 0);
  a.ldo = ldo; a.ldr = ldr; a.ldw = ldw; a.KH = KH; a.KW = KW; a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w;
  a.act = act; a.slope = slope; a.cscale = cscale; a.rscale = rscale; a.shuffle = shuffle;
  a.Ktot = Ktot; a.akrows = akrows > 0 ? akrows : 1;
  long long M = (long long)B * a.Ho * a.Wo;
  FFSR_CHECK(M < (1ll << 31) && (long long)B * H * W < (1ll << 31));
  a.M = (int)M;
  if (bn == 32) return launch_v3<32>(a, (hipStream_t)stream);
  return bn == 64 ? launch_v3<64>(a, (hipStream_t)stream) : launch_v3<128>(a, (hipStream_t)stream);
}
