// DIAGNOSTIC BUILD ONLY (tools/): the bf16x3 GEMM kernel with s_memtime stamps, never shipped or timed.
#include "../image-super-resolution_amd/csrc/ffsr_common.h"
namespace {
constexpr int BK = 32;
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
  unsigned long long* stamps;   // [grid][8]
};

template <int BN, bool HAS_AK>
__global__ __launch_bounds__(256) void conv_gemm_bf16x3_v3_kernel(ConvArgs3 p) {
  unsigned long long T[8];
  int ti = 0;
#define STAMP() do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); if (ti < 8) T[ti++] = t_; } while (0)
  STAMP();
  constexpr int BM = 128, WM = 64, WN = BN / 2;
  constexpr int TM = 2, TN = WN / 32;
  constexpr int AR = BM / 32, BRW = BN / 64;
  constexpr int RS = 80;
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
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const int m = min(m0 + rbase + 32 * i, p.M - 1);
        const size_t off = (size_t)(m / p.akrows) * p.Ktot + (kval ? k : 0);
        a_reg[i] *= *reinterpret_cast<const floatx4*>(p.akscale + off);
      }
    }
#pragma unroll
    for (int j = 0; j < BRW; ++j) {
      const size_t off = (size_t)(n0 + brow + 64 * j) * p.ldw + kt * BK + bseg;   // planes are padded: always valid
      bh_reg[j] = *reinterpret_cast<const floatx4*>(p.whi + off);
      bl_reg[j] = *reinterpret_cast<const floatx4*>(p.wlo + off);
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      uintx2 hi, lo;
      split4(a_reg[i], hi, lo);
      const int o = (rbase + 32 * i) * RS + kofs * 2;
      *reinterpret_cast<uintx2*>(Ahi + o) = hi;
      *reinterpret_cast<uintx2*>(Alo + o) = lo;
    }
#pragma unroll
    for (int j = 0; j < BRW; ++j) {
      const int o = (brow + 64 * j) * RS + bseg * 2;
      *reinterpret_cast<floatx4*>(Bhi + o) = bh_reg[j];
      *reinterpret_cast<floatx4*>(Blo + o) = bl_reg[j];
    }
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int wrow = (wave >> 1) * WM, wcol = (wave & 1) * WN;
  const int r = lane & 31, h = lane >> 5;
  const int nk = (p.Ktot + BK - 1) / BK;

  STAMP();   // 1: prologue address setup done
  load_tiles(0);
  store_tiles();
  __syncthreads();
  STAMP();   // 2: first tile landed in LDS
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tiles(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ko = 32 * ks + 16 * h;
      bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = *reinterpret_cast<const bf16x8*>(Ahi + (wrow + i * 32 + r) * RS + ko);
        al[i] = *reinterpret_cast<const bf16x8*>(Alo + (wrow + i * 32 + r) * RS + ko);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const bf16x8*>(Bhi + (wcol + j * 32 + r) * RS + ko);
        bl[j] = *reinterpret_cast<const bf16x8*>(Blo + (wcol + j * 32 + r) * RS + ko);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    if (kt == 0) STAMP();   // 3: first compute done
    if (kt + 1 < nk) store_tiles();
    __syncthreads();
    if (kt == 0) STAMP();   // 4: second tile stored (load latency exposed here)
  }
  STAMP();   // 5: main loop done

  // ---- epilogue.  Lean path (no PixelShuffle): one pointer per lane, compile-time row offsets, the activation switch
  // hoisted out of the element loop; rows are bounds-checked only in the last row tile.
  const bool full_rows = m0 + BM <= p.M;
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
  STAMP();   // 6: epilogue done
  if (threadIdx.x == 0) for (int i = 0; i < 8; ++i) p.stamps[(size_t)blockIdx.x * 8 + i] = i < ti ? T[i] : 0;
}

}
extern "C" int probe_gemm(const float* in, const void* whi, const void* wlo, int ldw, const float* zeros, float* out,
                          int M, int K, int N, unsigned long long* stamps, void* stream) {
  ConvArgs3 a;
  a.in = in; a.whi = (const unsigned short*)whi; a.wlo = (const unsigned short*)wlo; a.zeros = zeros; a.bias = nullptr;
  a.out = out; a.res = nullptr; a.cvec = nullptr; a.rvec = nullptr; a.akscale = nullptr;
  a.B = 1; a.H = 1; a.W = M; a.Cin = K; a.ldi = K; a.N = N; a.Ho = 1; a.Wo = M; a.ldo = N; a.ldr = 0; a.ldw = ldw;
  a.KH = 1; a.KW = 1; a.stride = 1; a.pad_h = 0; a.pad_w = 0; a.act = 0; a.slope = 0; a.cscale = 1; a.rscale = 1;
  a.shuffle = 0; a.M = M; a.Ktot = K; a.akrows = 1; a.stamps = stamps;
  int tiles = ((M + 127) / 128) * ((N + 63) / 64);
  hipLaunchKernelGGL((conv_gemm_bf16x3_v3_kernel<64, false>), dim3(tiles), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
