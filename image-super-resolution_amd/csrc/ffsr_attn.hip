// Window attention kernels (gfx950).
//
// (1) ffsr_window_attn_f32  -- DRCT's (shifted) 16x16 window MSA, fp32 MFMA 32x32x2, fused
//     QK^T*scale + relative-position bias (+ shift mask computed from region ids) -> softmax -> PV.
//     The cyclic roll, window partition / reverse of the reference are folded into the token addressing.
// (2) ffsr_grl_window_attn_f32 / ffsr_grl_stripe_attn_f32 -- GRL's 8x8 cosine window attention and the two-hop
//     anchored stripe attention.  64 tokens x head_dim 30: one wave per (window, head) on the VALU.
// (3) ffsr_pixel_mha_f32 -- the fusion net's per-pixel multi-head attention across 9 bands / 4 experts (K11).
#include "ffsr_common.h"

namespace {

// --------------------------------------------------------------------------------------------------------------
// (1) DRCT window attention.  grid = (heads, windows*B), 256 threads.
// qkv: [B*H*W, ldq] with q at col h*hd, k at C + h*hd, v at 2C + h*hd.  bias: the relative_position_bias_table
// [(2ws-1)^2, heads] itself (gathered on the fly from an LDS copy of the head's column).
// --------------------------------------------------------------------------------------------------------------
struct WinArgs {
  const float* qkv;
  const float* bias;
  float* out;
  int ldq, ldo, C, H, W, ws, shift, heads, hd, hdp, masked;
  float scale;
};

__device__ __forceinline__ int region_id(int p, int n, int ws, int shift) {
  // python slices [0,-ws), [-ws,-shift), [-shift, end) on an axis of length n
  return p < n - ws ? 0 : (p < n - shift ? 1 : 2);
}

// One workgroup (4 waves) per (window, head); each wave owns 64 queries.  Flash-style over 64-key tiles with the
// product computed TRANSPOSED: S^T = K Q^T puts the query on the MFMA lane and the keys in the 16 accumulator
// registers, so (a) the softmax row statistics are in-lane reductions plus one xor-32 exchange, and (b) the P^T tile
// is already laid out as the B operand of O^T += V^T P^T -- probabilities never touch LDS.  Q fragments live in
// registers; K / V tiles are staged through LDS.
template <int DT, int QT>  // DT = 32-wide head_dim tiles (hd <= 32*DT); QT = 32-query tiles per wave (WG = 128*QT queries)
__global__ __launch_bounds__(256, (QT == 1 ? (DT == 1 ? 4 : (DT == 2 ? 3 : 2)) : (DT <= 2 ? 2 : 1))) void window_attn_kernel(WinArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NGMAX = 4 * DT;       // 8-wide k groups
  constexpr int WS = 16, N = WS * WS;  // window edge / tokens per window (checked by the launcher)
  const int QS = p.hdp + 4;           // LDS row stride (floats); (hdp+4)/4 is odd -> b128 reads conflict-free
  float* KV = lds;                    // [128][QS]: rows 0-63 K tile, 64-127 V tile (also Q staging) + 64 floats slack
  int* tok_pix = reinterpret_cast<int*>(KV + 128 * QS + 64);  // [N]
  int* tok_reg = tok_pix + N;                                  // [N]
  float* rpb = reinterpret_cast<float*>(tok_reg + N);          // [(2ws-1)^2] relative position bias of this head

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int head = blockIdx.x;
  const int nwx = p.W / WS, nwy = p.H / WS;
  const int win = blockIdx.y % (nwx * nwy), b = blockIdx.y / (nwx * nwy);
  const int wy = win / nwx, wx = win % nwx;
  for (int t = tid; t < N; t += 256) {
    int py = t / WS, px = t % WS;
    int ys = wy * WS + py, xs = wx * WS + px;  // position in the rolled image
    int y = ys + p.shift, x = xs + p.shift;
    if (y >= p.H) y -= p.H;
    if (x >= p.W) x -= p.W;
    tok_pix[t] = (b * p.H + y) * p.W + x;
    tok_reg[t] = p.masked ? region_id(ys, p.H, WS, p.shift) * 3 + region_id(xs, p.W, WS, p.shift) : 0;
  }
  // only the windows of the last window row / column straddle regions of the shifted image
  const bool use_mask = p.masked && (wy == nwy - 1 || wx == nwx - 1);
  const int span = 2 * WS - 1;
  for (int t = tid; t < span * span; t += 256) rpb[t] = p.bias[(size_t)t * p.heads + head];
  const int hd = p.hd, hdp = p.hdp, ng = hdp >> 3;
  const int r32 = lane & 31, hh = lane >> 5;
  const float* qbase = p.qkv + head * hd;
  const int q0 = (QT == 1 ? (int)blockIdx.z * 128 : 0) + wave * 32 * QT;   // first query (window token) of this wave

  // ---- Q fragments -> registers (staged through LDS, 128 tokens per round)
  floatx4 qf[QT][NGMAX];
  // tile loader: rows wave, wave+4, ... of a 64-row tile; all global loads of a batch are issued before any LDS
  // store (unconditional, clamped column + 0/1 mask: a load under a branch would be waited for one by one)
  constexpr int DI = DT > 2 ? 2 : 1;   // 64-lane column passes per row (hd <= 64 * DI)
  constexpr int RB = DT == 1 ? 8 : 4;  // rows per batch per wave (register budget)
  float cmask[DI];
  int ccol[DI];
#pragma unroll
  for (int di = 0; di < DI; ++di) {
    const int d = lane + 64 * di;
    cmask[di] = d < hd ? 1.f : 0.f;
    ccol[di] = min(d, hd - 1);
  }
  auto load_rows = [&](int tok0, int col0, float scale, float* dst) {   // 64 tokens tok0.. -> dst[64][QS]
#pragma unroll 1
    for (int batch = 0; batch < 16 / RB; ++batch) {
      float reg[RB][DI];
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int rr = wave + 4 * (batch * RB + r);
        const float* src = qbase + (size_t)tok_pix[tok0 + rr] * p.ldq + col0;
#pragma unroll
        for (int di = 0; di < DI; ++di) reg[r][di] = src[ccol[di]];
      }
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int rr = wave + 4 * (batch * RB + r);
#pragma unroll
        for (int di = 0; di < DI; ++di)
          if (lane + 64 * di < hdp) dst[rr * QS + lane + 64 * di] = reg[r][di] * cmask[di] * scale;
      }
    }
  };
  // stage this workgroup's 128*QT queries through the K/V buffer, 128 at a time
  for (int round = 0; round < QT; ++round) {
    const int qb = (QT == 1 ? (int)blockIdx.z * 128 : round * 128);
    __syncthreads();
    load_rows(qb, 0, p.scale, KV);
    load_rows(qb + 64, 0, p.scale, KV + 64 * QS);
    __syncthreads();
    if (QT == 1) {
#pragma unroll
      for (int j = 0; j < NGMAX; ++j)
        if (j < ng) qf[0][j] = *reinterpret_cast<const floatx4*>(KV + (wave * 32 + r32) * QS + 8 * j + 4 * hh);
    } else if ((wave >> 1) == round) {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int j = 0; j < NGMAX; ++j)
          if (j < ng) qf[qt][j] = *reinterpret_cast<const floatx4*>(KV + ((wave & 1) * 64 + qt * 32 + r32) * QS + 8 * j + 4 * hh);
    }
  }
  int qreg[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) qreg[qt] = tok_reg[q0 + qt * 32 + r32];

  floatx16 o[DT][QT];
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int q = 0; q < QT; ++q)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[i][q][e] = 0.f;
  float mrow[QT], lrow[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    mrow[qt] = -3.0e38f;
    lrow[qt] = 0.f;
  }

#pragma unroll 1
  for (int kt = 0; kt < N / 64; ++kt) {
    __syncthreads();  // previous tile fully consumed
    load_rows(kt * 64, p.C, 1.0f, KV);                // K tile
    load_rows(kt * 64, 2 * p.C, 1.0f, KV + 64 * QS);  // V tile
    __syncthreads();
#pragma unroll 1
    for (int ks = 0; ks < 2; ++ks) {
      const float* Kt = KV + (ks * 32) * QS;
      const float* Vt = KV + (64 + ks * 32) * QS;
      const int key0 = kt * 64 + ks * 32 + 4 * hh;  // + (e&3) + 8*(e>>2)
      floatx16 s[QT];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
        for (int e = 0; e < 16; ++e) s[qt][e] = 0.f;
#pragma unroll
        for (int j = 0; j < NGMAX; ++j) {
          if (j < ng) {
            const floatx4 a = *reinterpret_cast<const floatx4*>(Kt + r32 * QS + 8 * j + 4 * hh);
#pragma unroll
            for (int t = 0; t < 4; ++t) s[qt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], qf[qt][j][t], s[qt], 0, 0, 0);
          }
        }
        // relative position bias: table[(qy - ky + ws-1) * (2ws-1) + (qx - kx + ws-1)]  (drct_arch.py:148-158,188-191)
        const int q = q0 + qt * 32 + r32;
        const int qidx = (q / WS + WS - 1) * span + (q % WS) + WS - 1;
        // (one base pointer per tile and compile-time offsets: all 16 table reads are issued as one batch)
        const float* rb = rpb + (qidx - (kt * 4 + ks * 2) * span - 4 * hh);
        float bv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) bv[e] = rb[-((e >> 3) * span + (e & 3) + 8 * ((e >> 2) & 1))];
        if (use_mask) {   // workgroup-uniform
#pragma unroll
          for (int e = 0; e < 16; ++e) bv[e] += tok_reg[key0 + (e & 3) + 8 * (e >> 2)] != qreg[qt] ? -100.0f : 0.f;
        }
        float mx = -3.0e38f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          s[qt][e] += bv[e];
          mx = fmaxf(mx, s[qt][e]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(mrow[qt], mx);
        const float corr = __expf(mrow[qt] - mnew);
        mrow[qt] = mnew;
        float ps = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          s[qt][e] = __expf(s[qt][e] - mnew);
          ps += s[qt][e];
        }
        lrow[qt] = lrow[qt] * corr + ps;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) o[i][qt][e] *= corr;
      }
#pragma unroll
      for (int i = 0; i < DT; ++i) {
        float vv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) vv[e] = Vt[((e & 3) + 8 * (e >> 2) + 4 * hh) * QS + i * 32 + r32];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) o[i][qt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[e], s[qt][e], o[i][qt], 0, 0, 0);
        }
      }
    }
  }
  // ---- normalise and store O^T: lane = query, registers = head-dim rows
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const float inv = 1.0f / (lrow[qt] + __shfl_xor(lrow[qt], 32, 64));
    float* orow = p.out + (size_t)tok_pix[q0 + qt * 32 + r32] * p.ldo + head * hd;
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int d = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        if (d < hd) orow[d] = o[i][qt][e] * inv;
      }
  }
}

// --------------------------------------------------------------------------------------------------------------
// (1b) The same attention with both contractions on the bf16 MFMA as 3-term split products (x = hi + lo in bf16,
// x*y ~ hi*hi + hi*lo + lo*hi, fp32 accumulate: ~1e-5 relative, the arithmetic of ffsr_conv2d_bf16x3) -- 16x the f32
// MFMA rate at 3 instructions per product.  Same decomposition as window_attn_kernel<DT, 1> (128 queries per
// workgroup, S^T = K Q^T so the query sits on the lane, flash-style over 64-key tiles); what changes is the operand
// staging: K (and Q) tiles live in LDS as bf16 hi / lo planes [key][d], V as TRANSPOSED planes [d][key], because the
// second product O^T += V^T P^T takes P^T straight from the S^T accumulator registers (registers 8s..8s+7 of a lane,
// converted pairwise, are the B fragment of k-step s) and that fixes the key order of the A fragment to
// key = 16s + 8(j>>2) + 4h + (j&3) for element j of lane half h: two 8-byte reads of a V^T row.
// --------------------------------------------------------------------------------------------------------------
typedef __bf16 abf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned auintx2 __attribute__((ext_vector_type(2)));
typedef unsigned auintx4 __attribute__((ext_vector_type(4)));

// NW = 4: 128 queries per workgroup, 2 workgroups per (window, head), each staging all K / V tiles;
// NW = 8: one 512-thread workgroup per (window, head): K / V are fetched and split once.
template <int KS, int NW>   // head_dim padded to 16 * KS; NW waves of 32 queries
__global__ __launch_bounds__(64 * NW, (NW == 8 ? 2 : (KS <= 2 ? 4 : (KS <= 4 ? 3 : 2)))) void window_attn_x3_kernel(WinArgs p) {
  constexpr int NT = 64 * NW;           // threads
  constexpr int RPW = 64 / NW;          // rows of a 64-token tile staged by one wave
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  constexpr int HDP = 16 * KS, DT = (KS + 1) / 2;
  constexpr int WS = 16, N = WS * WS;
  constexpr int RSK = HDP * 2 + 16;   // K / Q plane row stride in bytes: an odd number of 16-byte slots -> b128 reads conflict-free
  constexpr int RSV = 136;            // V^T plane row stride: 64 keys * 2 B + 8 -> b64 reads of 32 rows conflict-free
  constexpr int KPL = 64 * RSK, VPL = DT * 32 * RSV;
  unsigned char* Khi = ldsb;
  unsigned char* Klo = Khi + KPL;
  unsigned char* Vhi = Klo + KPL;
  unsigned char* Vlo = Vhi + VPL;
  int* tok_pix = reinterpret_cast<int*>(Vlo + VPL);
  unsigned char* tok_reg = reinterpret_cast<unsigned char*>(tok_pix + N);   // [N] region id of every window token (bytes)
  float* rpb = reinterpret_cast<float*>(tok_reg + N);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware work order (1-D grid): workgroups are dealt round-robin over the 8 XCDs, so the workgroups of one XCD
  // take a contiguous range of (window, query half, head) triples with the head fastest -- all heads of a window read
  // the same qkv rows (each head only hd of the 3C floats of a row), which then stay in ONE L2.
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  constexpr int NZ = NW == 4 ? 2 : 1;
  const int head = logical % p.heads;
  const int zq = (logical / p.heads) % NZ;            // which 128-query half (NW = 4)
  const int wlin = logical / (p.heads * NZ);          // window index over the batch
  const int nwx = p.W / WS, nwy = p.H / WS;
  const int win = wlin % (nwx * nwy), b = wlin / (nwx * nwy);
  const int wy = win / nwx, wx = win % nwx;
  for (int t = tid; t < N; t += NT) {
    int py = t / WS, px = t % WS;
    int ys = wy * WS + py, xs = wx * WS + px;  // position in the rolled image
    int y = ys + p.shift, x = xs + p.shift;
    if (y >= p.H) y -= p.H;
    if (x >= p.W) x -= p.W;
    tok_pix[t] = (b * p.H + y) * p.W + x;
    tok_reg[t] = (unsigned char)(p.masked ? region_id(ys, p.H, WS, p.shift) * 3 + region_id(xs, p.W, WS, p.shift) : 0);
  }
  const bool use_mask = p.masked && (wy == nwy - 1 || wx == nwx - 1);
  const int span = 2 * WS - 1;
  for (int t = tid; t < span * span; t += NT) rpb[t] = p.bias[(size_t)t * p.heads + head];
  const int hd = p.hd;
  const int r32 = lane & 31, hh = lane >> 5;
  const float* qbase = p.qkv + head * hd;
  const int q0 = (NW == 4 ? zq * 128 : 0) + wave * 32;   // first query (window token) of this wave

  // ---- row loader (K tile / Q round): 64 tokens -> bf16 hi / lo planes [64][RSK].  A row slice is only head_dim floats
  // (120 B at head_dim 30): one row per load instruction would use 15 of the 64 lanes, so the lanes are split into LR
  // rows x LP channel pairs (LP = HDP/2 rounded up to a power of two) and every instruction fetches LR rows.
  // All global loads of a batch are issued before the conversions (clamped column + 0/1 mask, no branches around loads).
  constexpr int LP = HDP <= 32 ? 16 : (HDP <= 64 ? 32 : 64), LR = 64 / LP;   // lanes per row, rows per instruction
  constexpr int KI = RPW / LR;                                               // row-load instructions per wave and tile
  const int pr = lane % LP, rl = lane / LP;
  const int d0 = 2 * pr;
  const float m0 = d0 < hd ? 1.f : 0.f, m1 = d0 + 1 < hd ? 1.f : 0.f;
  const int c0 = min(d0, hd - 1), c1 = min(d0 + 1, hd - 1);
  auto load_rows = [&](int tok0, int col0, float scale) {
    float reg[KI][2];
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      const int rr = RPW * wave + LR * i + rl;
      const float* src = qbase + (size_t)tok_pix[tok0 + rr] * p.ldq + col0;
      reg[i][0] = src[c0];
      reg[i][1] = src[c1];
    }
    if (d0 < HDP) {
#pragma unroll
      for (int i = 0; i < KI; ++i) {
        const int rr = RPW * wave + LR * i + rl;
        unsigned h, l;
        ffsr_split2(reg[i][0] * m0 * scale, reg[i][1] * m1 * scale, h, l);
        *reinterpret_cast<unsigned*>(Khi + rr * RSK + pr * 4) = h;
        *reinterpret_cast<unsigned*>(Klo + rr * RSK + pr * 4) = l;
      }
    }
  };
  // ---- K / V tile loader: 64 tokens -> K planes [64][RSK] (as load_rows) and TRANSPOSED V planes [d][64 keys]
  // (lanes = LVR key pairs x LV channels, this wave's RPW keys as packed pairs).  Every global load of the tile is issued
  // before the first conversion, so the workgroup pays ONE exposed memory latency per tile.  (Measured and rejected:
  // issuing the loads of tile t+1 before the work on tile t -- 48-64 more live registers cost a wave per SIMD: 0-12 %
  // slower.)
  constexpr int DI = HDP > 64 ? 2 : 1;
  constexpr int LV = HDP <= 32 ? 32 : 64, LVR = 64 / LV;      // channel lanes, key pairs per instruction
  constexpr int VI = RPW / (2 * LVR);                          // key-pair rounds per wave and tile
  const int vd = lane % LV, vg = lane / LV;
  auto load_kv = [&](int tok0) {
    float kreg[KI][2], vreg[DI][VI][2];
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      const float* src = qbase + (size_t)tok_pix[tok0 + RPW * wave + LR * i + rl] * p.ldq + p.C;
      kreg[i][0] = src[c0];
      kreg[i][1] = src[c1];
    }
#pragma unroll
    for (int di = 0; di < DI; ++di) {
      const int cc = min(vd + 64 * di, hd - 1);
#pragma unroll
      for (int i = 0; i < VI; ++i) {
        const int key = RPW * wave + 2 * (LVR * i + vg);
        vreg[di][i][0] = (qbase + (size_t)tok_pix[tok0 + key] * p.ldq + 2 * p.C)[cc];
        vreg[di][i][1] = (qbase + (size_t)tok_pix[tok0 + key + 1] * p.ldq + 2 * p.C)[cc];
      }
    }
    if (d0 < HDP) {
#pragma unroll
      for (int i = 0; i < KI; ++i) {
        const int rr = RPW * wave + LR * i + rl;
        unsigned h, l;
        ffsr_split2(kreg[i][0] * m0, kreg[i][1] * m1, h, l);
        *reinterpret_cast<unsigned*>(Khi + rr * RSK + pr * 4) = h;
        *reinterpret_cast<unsigned*>(Klo + rr * RSK + pr * 4) = l;
      }
    }
#pragma unroll
    for (int di = 0; di < DI; ++di) {
      const int d = vd + 64 * di;
      const float mk = d < hd ? 1.f : 0.f;
      if (d < DT * 32) {
#pragma unroll
        for (int i = 0; i < VI; ++i) {
          const int key = RPW * wave + 2 * (LVR * i + vg);
          unsigned h, l;
          ffsr_split2(vreg[di][i][0] * mk, vreg[di][i][1] * mk, h, l);
          *reinterpret_cast<unsigned*>(Vhi + d * RSV + key * 2) = h;
          *reinterpret_cast<unsigned*>(Vlo + d * RSV + key * 2) = l;
        }
      }
    }
  };

  // ---- Q fragments (B operand of S^T = K Q^T) -> registers, staged 64 queries at a time through the K planes
  abf16x8 qh[KS], ql[KS];
  for (int round = 0; round < NW / 2; ++round) {
    __syncthreads();
    load_rows((NW == 4 ? zq * 128 : 0) + round * 64, 0, p.scale);
    __syncthreads();
    if ((wave >> 1) == round) {
      const int row = (wave & 1) * 32 + r32;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        qh[ks] = *reinterpret_cast<const abf16x8*>(Khi + row * RSK + (16 * ks + 8 * hh) * 2);
        ql[ks] = *reinterpret_cast<const abf16x8*>(Klo + row * RSK + (16 * ks + 8 * hh) * 2);
      }
    }
  }
  const int qreg = tok_reg[q0 + r32];

  floatx16 o[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
  float mrow = -3.0e38f, lrow = 0.f;
  const int q = q0 + r32;
  const int qidx = (q / WS + WS - 1) * span + (q % WS) + WS - 1;

#pragma unroll 1
  for (int kt = 0; kt < N / 64; ++kt) {
    __syncthreads();  // previous tile fully consumed
    load_kv(kt * 64);
    __syncthreads();
#pragma unroll 1
    for (int sub = 0; sub < 2; ++sub) {
      // S^T[32 keys][32 queries]
      floatx16 s;
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = 0.f;
      const unsigned char* kr = Khi + (sub * 32 + r32) * RSK + 16 * hh;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const abf16x8 kh = *reinterpret_cast<const abf16x8*>(kr + 32 * ks);
        const abf16x8 kl = *reinterpret_cast<const abf16x8*>(kr + KPL + 32 * ks);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[ks], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[ks], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[ks], s, 0, 0, 0);
      }
      // relative position bias: table[(qy - ky + ws-1) * (2ws-1) + (qx - kx + ws-1)]  (drct_arch.py:148-158,188-191).
      // key = key0 + (e&3) + 8*(e>>2) with key0 = 64 kt + 32 sub + 4 hh, so ky = 4 kt + 2 sub + (e>>3) and
      // kx = 4 hh + (e&3) + 8*((e>>2)&1): one base pointer per sub-tile, compile-time offsets, all 16 reads in one batch
      const int key0 = kt * 64 + sub * 32 + 4 * hh;
      const float* rb = rpb + (qidx - (kt * 4 + sub * 2) * span - 4 * hh);
      float bv[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) bv[e] = rb[-((e >> 3) * span + (e & 3) + 8 * ((e >> 2) & 1))];
      if (use_mask) {   // workgroup-uniform; region ids of 4 consecutive keys in one 32-bit read
        unsigned rg[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) rg[g] = *reinterpret_cast<const unsigned*>(tok_reg + key0 + 8 * g);
#pragma unroll
        for (int e = 0; e < 16; ++e) bv[e] += (((rg[e >> 2] >> (8 * (e & 3))) & 0xffu) != (unsigned)qreg) ? -100.0f : 0.f;
      }
      float mx = -3.0e38f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        s[e] += bv[e];
        mx = fmaxf(mx, s[e]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mnew = fmaxf(mrow, mx);
      const float corr = __expf(mrow - mnew);
      mrow = mnew;
      float ps = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        s[e] = __expf(s[e] - mnew);
        ps += s[e];
      }
      lrow = lrow * corr + ps;
#pragma unroll
      for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[i][e] *= corr;
      // P^T fragments of the two 16-key steps, straight from the accumulator registers
      abf16x8 ph[2], pl[2];
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        unsigned h4[4], l4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) ffsr_split2(s[8 * st + 2 * c], s[8 * st + 2 * c + 1], h4[c], l4[c]);
        ph[st] = __builtin_bit_cast(abf16x8, auintx4{h4[0], h4[1], h4[2], h4[3]});
        pl[st] = __builtin_bit_cast(abf16x8, auintx4{l4[0], l4[1], l4[2], l4[3]});
      }
      // O^T[d][query] += V^T[d][key] P^T[key][query]; element j of lane half h <-> key 16 st + 8 (j >> 2) + 4 h + (j & 3)
#pragma unroll
      for (int i = 0; i < DT; ++i) {
        const unsigned char* vr = Vhi + (i * 32 + r32) * RSV + (sub * 32 + 4 * hh) * 2;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          const auintx2 a0 = *reinterpret_cast<const auintx2*>(vr + 32 * st), a1 = *reinterpret_cast<const auintx2*>(vr + 32 * st + 16);
          const auintx2 b0 = *reinterpret_cast<const auintx2*>(vr + VPL + 32 * st), b1 = *reinterpret_cast<const auintx2*>(vr + VPL + 32 * st + 16);
          const abf16x8 vh = __builtin_bit_cast(abf16x8, auintx4{a0[0], a0[1], a1[0], a1[1]});
          const abf16x8 vl = __builtin_bit_cast(abf16x8, auintx4{b0[0], b0[1], b1[0], b1[1]});
          o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph[st], o[i], 0, 0, 0);
          o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl[st], o[i], 0, 0, 0);
          o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph[st], o[i], 0, 0, 0);
        }
      }
    }
  }
  // ---- normalise and store O^T: lane = query, registers = head-dim rows
  const float inv = 1.0f / (lrow + __shfl_xor(lrow, 32, 64));
  float* orow = p.out + (size_t)tok_pix[q0 + r32] * p.ldo + head * hd;
  // registers 4g .. 4g+3 of a tile are 4 consecutive channels starting at an even d: 8-byte stores whenever this head's
  // slice starts on an even column (one 4-byte store per lane and channel is store-issue bound: hd 30: 271 -> 237 us)
  const bool pair_ok = (((head * hd) & 1) == 0) && ((p.ldo & 1) == 0) && ((reinterpret_cast<uintptr_t>(p.out) & 7) == 0);
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int e = 0; e < 16; e += 2) {
      const int d = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
      if (pair_ok && d + 1 < hd) {
        *reinterpret_cast<float2*>(orow + d) = float2{o[i][e] * inv, o[i][e + 1] * inv};
      } else {
        if (d < hd) orow[d] = o[i][e] * inv;
        if (d + 1 < hd) orow[d + 1] = o[i][e + 1] * inv;
      }
    }
}

// --------------------------------------------------------------------------------------------------------------
// (2) GRL attention: one wave per (window, head).  Tokens of an 8x8 window; lane = query token.
// --------------------------------------------------------------------------------------------------------------
// 1-D grid, workgroups dealt round-robin over the 8 XCDs: give every XCD a contiguous range of logical ids (bijective
// for any grid size).  All heads of a window read the same qkv rows (each only HD of their floats); with the head as the
// fastest logical index they run at the same time on CUs of ONE L2 instead of re-fetching the rows per XCD.
__device__ __forceinline__ int xcd_logical_id() {
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
}

__device__ __forceinline__ float inv_norm(const float* v, int n) {
  float s = 0.f;
  for (int i = 0; i < n; ++i) s = fmaf(v[i], v[i], s);
  return 1.0f / fmaxf(sqrtf(s), 1e-12f);  // F.normalize eps
}

// qkv: [P, ldq]; this branch's q/k/v start at column col0 with layout [3][heads][HD]; out: [P, ldo] at ocol0 + h*HD
// biasT: [heads][64 keys][64 queries]; logit: [heads] already exp(min(logit_scale, ln 100))
template <int HD>
__global__ __launch_bounds__(64) void grl_window_kernel(const float* __restrict__ qkv, int ldq, int col0,
                                                        const float* __restrict__ biasT, const float* __restrict__ logit,
                                                        float* __restrict__ out, int ldo, int ocol0, int H, int W, int heads,
                                                        int shift) {
  constexpr int WS = 8, N = 64;
  __shared__ __attribute__((aligned(16))) float Ks[N][HD + 2];
  __shared__ __attribute__((aligned(16))) float Vs[N][HD + 2];
  __shared__ int regs[N];
  const int lane = threadIdx.x;
  const int logical = xcd_logical_id();                 // XCD-aware order, head fastest: the heads of a window share one L2
  const int head = logical % heads;
  const int nwx = W / WS, nwy = H / WS;
  const int win = (logical / heads) % (nwx * nwy), b = (logical / heads) / (nwx * nwy);
  const int wy = win / nwx, wx = win % nwx;
  const int ys = wy * WS + lane / WS, xs = wx * WS + lane % WS;
  int y = ys + shift, x = xs + shift;
  if (y >= H) y -= H;
  if (x >= W) x -= W;
  const size_t pix = ((size_t)b * H + y) * W + x;
  const float* row = qkv + pix * ldq + col0 + head * HD;
  float q[HD], k[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) {
    q[d] = row[d];
    k[d] = row[heads * HD + d];
    Vs[lane][d] = row[2 * heads * HD + d];
  }
  const float qn = inv_norm(q, HD) * logit[head], kn = inv_norm(k, HD);
#pragma unroll
  for (int d = 0; d < HD; ++d) {
    q[d] *= qn;
    Ks[lane][d] = k[d] * kn;
  }
  const int myreg = shift ? region_id(ys, H, WS, shift) * 3 + region_id(xs, W, WS, shift) : 0;
  regs[lane] = myreg;
  __syncthreads();
  float s[N];
  float m = -3.0e38f;
  const float* bt = biasT + (size_t)head * N * N + lane;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    float a = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) a = fmaf(q[d], Ks[j][d], a);
    a += bt[j * N];
    if (shift && regs[j] != myreg) a += -100.0f;
    s[j] = a;
    m = fmaxf(m, a);
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    s[j] = expf(s[j] - m);
    sum += s[j];
  }
  const float inv = 1.0f / sum;
  float o[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) o[d] = 0.f;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const float pj = s[j] * inv;
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = fmaf(pj, Vs[j][d], o[d]);
  }
  float* orow = out + pix * ldo + ocol0 + head * HD;
#pragma unroll
  for (int d = 0; d < HD; ++d) orow[d] = o[d];
}

// (Measured and rejected: an LDS-free split-bf16 MFMA form of this kernel -- K / Q / V fetched from global memory straight
// into MFMA fragment layout, 48 MFMAs per window-head instead of ~8000 FMAs per lane -- runs in the same 170 us as the
// VALU kernel at 352x512: with 120-byte row slices per head the kernel is bound by its scattered global reads, not by
// arithmetic.  What helps is the XCD-aware, head-fastest order above (-6 %).  Also rejected: staging the q / k / v rows
// through LDS with row-coalesced float2 loads and stores instead of one row per lane -- 181 us against 157 us; 8-byte
// instead of 4-byte accesses to the lane's own rows -- no change.)

// anchors: [B, H/2, W/2, lda] with head h at column h*HD.  bias1T: [heads][64 keys][16 anchors] (anchor <- window),
// bias2T: [heads][16 anchors][64 queries] (window <- anchor).  No stripe shift in GRL-B (grl/__init__.py:139).
template <int HD>
__global__ __launch_bounds__(64) void grl_stripe_kernel(const float* __restrict__ qkv, int ldq, int col0,
                                                        const float* __restrict__ anchor, int lda,
                                                        const float* __restrict__ bias1T, const float* __restrict__ bias2T,
                                                        const float* __restrict__ logit1, const float* __restrict__ logit2,
                                                        float* __restrict__ out, int ldo, int ocol0, int H, int W, int heads) {
  constexpr int WS = 8, N = 64, AW = 4, NA = 16;
  __shared__ __attribute__((aligned(16))) float Ks[N][HD + 2];
  __shared__ __attribute__((aligned(16))) float Vs[N][HD + 2];
  __shared__ __attribute__((aligned(16))) float As[NA][HD + 2];   // normalised anchors
  __shared__ __attribute__((aligned(16))) float Gs[NA][HD + 2];   // anchors' gathered values
  __shared__ float P1[NA][N + 1];
  const int lane = threadIdx.x;
  const int logical = xcd_logical_id();                 // XCD-aware order, head fastest: the heads of a window share one L2
  const int head = logical % heads;
  const int nwx = W / WS, nwy = H / WS;
  const int win = (logical / heads) % (nwx * nwy), b = (logical / heads) / (nwx * nwy);
  const int wy = win / nwx, wx = win % nwx;
  const int y = wy * WS + lane / WS, x = wx * WS + lane % WS;
  const size_t pix = ((size_t)b * H + y) * W + x;
  const float* row = qkv + pix * ldq + col0 + head * HD;
  float q[HD], k[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) {
    q[d] = row[d];
    k[d] = row[heads * HD + d];
    Vs[lane][d] = row[2 * heads * HD + d];
  }
  const float kn = inv_norm(k, HD);
#pragma unroll
  for (int d = 0; d < HD; ++d) Ks[lane][d] = k[d] * kn;
  if (lane < NA) {
    const int ay = wy * AW + lane / AW, ax = wx * AW + lane % AW;
    const float* ar = anchor + (((size_t)b * (H / 2) + ay) * (W / 2) + ax) * lda + head * HD;
    float a[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) a[d] = ar[d];
    const float an = inv_norm(a, HD);
#pragma unroll
    for (int d = 0; d < HD; ++d) As[lane][d] = a[d] * an;
  }
  __syncthreads();
  // ---- hop 1: anchors attend to the stripe's tokens. lane = (anchor a, key quarter kq)
  {
    const int a = lane & 15, kq = lane >> 4;
    const float l1 = logit1[head];
    float s[16];
    float m = -3.0e38f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const int j = kq * 16 + jj;
      float acc = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) acc = fmaf(As[a][d], Ks[j][d], acc);
      acc = acc * l1 + bias1T[((size_t)head * N + j) * NA + a];
      s[jj] = acc;
      m = fmaxf(m, acc);
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      s[jj] = expf(s[jj] - m);
      sum += s[jj];
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) P1[a][kq * 16 + jj] = s[jj] * inv;
  }
  __syncthreads();
  // G[a][d] = sum_j P1[a][j] V[j][d]   (16*HD outputs over 64 lanes)
  for (int i = lane; i < NA * HD; i += 64) {
    const int a = i / HD, d = i - a * HD;
    float acc = 0.f;
    for (int j = 0; j < N; ++j) acc = fmaf(P1[a][j], Vs[j][d], acc);
    Gs[a][d] = acc;
  }
  __syncthreads();
  // ---- hop 2: tokens attend to the anchors. lane = query token
  const float qn = inv_norm(q, HD) * logit2[head];
  float s2[NA];
  float m2 = -3.0e38f;
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) acc = fmaf(q[d], As[a][d], acc);
    acc = acc * qn + bias2T[((size_t)head * NA + a) * N + lane];
    s2[a] = acc;
    m2 = fmaxf(m2, acc);
  }
  float sum2 = 0.f;
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    s2[a] = expf(s2[a] - m2);
    sum2 += s2[a];
  }
  const float inv2 = 1.0f / sum2;
  float* orow = out + pix * ldo + ocol0 + head * HD;
#pragma unroll
  for (int d = 0; d < HD; ++d) {
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < NA; ++a) acc = fmaf(s2[a], Gs[a][d], acc);
    orow[d] = acc * inv2;
  }
}

// --------------------------------------------------------------------------------------------------------------
// (3) per-pixel MHA: qkv [S*T, 3E] (q | k | v, heads of 16) -> out [S*T, E].  thread = (sequence, token, head)
// drop_thr > 0 (training): attention dropout -- probability (seq, head, t, j) is kept iff ffsr_rng_u32(seed, its index) >=
// drop_thr and scaled by keep_scale = 1 / (1 - p)  (nn.MultiheadAttention(dropout=0.1), large_kernel_attention.py:196,298)
// --------------------------------------------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(256) void pixel_mha_kernel(   // (without the bound: 128 registers, the T = 9 rows spill)
    const float* __restrict__ qkv, int ldq, float* __restrict__ out, int ldo, long long S,
                                 int E, int heads, unsigned drop_thr, float keep_scale, unsigned long long seed) {
  constexpr int HD = 16;
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= S * T * heads) return;
  const int h = (int)(idx % heads);
  const long long st = idx / heads;  // sequence*T + token
  const long long seq = st / T;
  const float* qrow = qkv + st * ldq + h * HD;
  float q[HD];
#pragma unroll
  for (int d = 0; d < HD; d += 4) {
    floatx4 v = *reinterpret_cast<const floatx4*>(qrow + d);
    q[d] = v[0] * 0.25f; q[d + 1] = v[1] * 0.25f; q[d + 2] = v[2] * 0.25f; q[d + 3] = v[3] * 0.25f;  // 1/sqrt(16)
  }
  float s[T];
  float m = -3.0e38f;
#pragma unroll
  for (int j = 0; j < T; ++j) {
    const float* krow = qkv + (seq * T + j) * ldq + E + h * HD;
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      floatx4 v = *reinterpret_cast<const floatx4*>(krow + d);
      acc = fmaf(q[d], v[0], acc); acc = fmaf(q[d + 1], v[1], acc);
      acc = fmaf(q[d + 2], v[2], acc); acc = fmaf(q[d + 3], v[3], acc);
    }
    s[j] = acc;
    m = fmaxf(m, acc);
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < T; ++j) {
    s[j] = expf(s[j] - m);
    sum += s[j];
  }
  const float inv = 1.0f / sum;
  float o[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) o[d] = 0.f;
#pragma unroll
  for (int j = 0; j < T; ++j) {
    const float* vrow = qkv + (seq * T + j) * ldq + 2 * E + h * HD;
    float pj = s[j] * inv;
    if (drop_thr) {
      const unsigned long long e = (((unsigned long long)seq * heads + h) * T + (unsigned long long)(st - seq * T)) * T + j;
      pj = ffsr_rng_u32(seed, e) >= drop_thr ? pj * keep_scale : 0.f;
    }
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      floatx4 v = *reinterpret_cast<const floatx4*>(vrow + d);
      o[d] = fmaf(pj, v[0], o[d]); o[d + 1] = fmaf(pj, v[1], o[d + 1]);
      o[d + 2] = fmaf(pj, v[2], o[d + 2]); o[d + 3] = fmaf(pj, v[3], o[d + 3]);
    }
  }
  float* orow = out + st * ldo + h * HD;
#pragma unroll
  for (int d = 0; d < HD; d += 4) {
    floatx4 v = {o[d], o[d + 1], o[d + 2], o[d + 3]};
    *reinterpret_cast<floatx4*>(orow + d) = v;
  }
}

inline bool grid_ok(long long y) { return y > 0 && y <= 65535; }

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int ffsr_window_attn_f32(const float* qkv, int ldq, const float* bias, float* out, int ldo, int B, int H, int W,
                                    int C, int heads, int ws, int shift, float scale, int variant, void* stream) {
  FFSR_CHECK(qkv && bias && out && B > 0 && heads > 0 && C % heads == 0);
  FFSR_CHECK(ws > 0 && H % ws == 0 && W % ws == 0 && (ws * ws) % 64 == 0 && ws * ws <= 256);
  FFSR_CHECK(shift >= 0 && shift < ws && ldq >= 3 * C && ldo >= C);
  const int hd = C / heads;
  FFSR_CHECK(hd <= 128);
  WinArgs a;
  a.qkv = qkv; a.bias = bias; a.out = out; a.ldq = ldq; a.ldo = ldo; a.C = C; a.H = H; a.W = W; a.ws = ws;
  a.shift = shift; a.heads = heads; a.hd = hd; a.hdp = (hd + 7) / 8 * 8; a.masked = shift > 0; a.scale = scale;
  const int N = ws * ws;
  FFSR_CHECK(ws == 16);
  FFSR_CHECK(grid_ok((H / ws) * (W / ws) * B));
  if (variant == 0 || variant == 4) {
    // split-bf16 MFMA kernel (default)
    const int KS = (hd + 15) / 16, DTx = (KS + 1) / 2;
    const size_t ldsx = (size_t)2 * 64 * (KS * 32 + 16) + (size_t)2 * DTx * 32 * 136 + N * 4 + N +
                        (size_t)(2 * ws - 1) * (2 * ws - 1) * 4;
    const bool wide = variant == 4;   // 0: two 256-thread workgroups per (window, head); 4: one 512-thread workgroup
    dim3 gridx((unsigned)(heads * (H / ws) * (W / ws) * B * (wide ? 1 : 2)));
#define LAUNCH_X3(K)                                                                                                  \
  {                                                                                                                   \
    if (ldsx > 64 * 1024) {                                                                                           \
      (void)hipFuncSetAttribute((const void*)window_attn_x3_kernel<K, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)ldsx);                                                                           \
      (void)hipFuncSetAttribute((const void*)window_attn_x3_kernel<K, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)ldsx);                                                                           \
    }                                                                                                                 \
    if (wide) FFSR_LAUNCH((window_attn_x3_kernel<K, 8>), gridx, dim3(512), ldsx, ST, a);                        \
    else FFSR_LAUNCH((window_attn_x3_kernel<K, 4>), gridx, dim3(256), ldsx, ST, a);                             \
  }
    switch (KS) {
      case 1: LAUNCH_X3(1); break;
      case 2: LAUNCH_X3(2); break;
      case 3: LAUNCH_X3(3); break;
      case 4: LAUNCH_X3(4); break;
      case 5: LAUNCH_X3(5); break;
      case 6: LAUNCH_X3(6); break;
      case 7: LAUNCH_X3(7); break;
      default: LAUNCH_X3(8); break;
    }
#undef LAUNCH_X3
    return ffsr_launch_status();
  }
  const size_t lds = (size_t)(128 * (a.hdp + 4) + 64) * 4 + 2 * N * 4 + (size_t)(2 * ws - 1) * (2 * ws - 1) * 4;
  FFSR_CHECK(lds <= 160 * 1024);
  const int DT = (hd + 31) / 32;
  // exact f32-MFMA kernel (variant 1 / 2 / 3).  QT = 1 (128 queries per workgroup, 2 workgroups per window-head) keeps
  // the register file small enough for 2+ waves per SIMD at the large head dims; QT = 2 loads K/V once per window-head.
  const int QT = variant == 2 ? 2 : 1;   // measured: QT = 1 is 20-35 % faster at every head dim
  dim3 grid(heads, (H / ws) * (W / ws) * B, QT == 1 ? 2 : 1);
#define LAUNCH_WIN(D, Q)                                                                                            \
  {                                                                                                                 \
    if (lds > 64 * 1024)                                                                                            \
      (void)hipFuncSetAttribute((const void*)window_attn_kernel<D, Q>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                                          \
    FFSR_LAUNCH((window_attn_kernel<D, Q>), grid, dim3(256), lds, ST, a);                                     \
  }
  switch (DT * 10 + QT) {
    case 11: LAUNCH_WIN(1, 1); break;
    case 12: LAUNCH_WIN(1, 2); break;
    case 21: LAUNCH_WIN(2, 1); break;
    case 22: LAUNCH_WIN(2, 2); break;
    case 31: LAUNCH_WIN(3, 1); break;
    case 32: LAUNCH_WIN(3, 2); break;
    case 41: LAUNCH_WIN(4, 1); break;
    default: LAUNCH_WIN(4, 2); break;
  }
#undef LAUNCH_WIN
  return ffsr_launch_status();
}

extern "C" int ffsr_grl_window_attn_f32(const float* qkv, int ldq, int col0, const float* biasT, const float* logit,
                                        float* out, int ldo, int ocol0, int B, int H, int W, int heads, int hd, int shift,
                                        void* stream) {
  FFSR_CHECK(qkv && biasT && logit && out && B > 0 && H % 8 == 0 && W % 8 == 0 && heads > 0 && shift >= 0 && shift < 8);
  dim3 grid((unsigned)((H / 8) * (W / 8) * B * heads));
  switch (hd) {
    case 30: FFSR_LAUNCH(grl_window_kernel<30>, grid, dim3(64), 0, ST, qkv, ldq, col0, biasT, logit, out, ldo, ocol0, H, W, heads, shift); break;
    case 10: FFSR_LAUNCH(grl_window_kernel<10>, grid, dim3(64), 0, ST, qkv, ldq, col0, biasT, logit, out, ldo, ocol0, H, W, heads, shift); break;
    default: return FFSR_EINVAL;
  }
  return ffsr_launch_status();
}

extern "C" int ffsr_grl_stripe_attn_f32(const float* qkv, int ldq, int col0, const float* anchor, int lda,
                                        const float* bias1T, const float* bias2T, const float* logit1, const float* logit2,
                                        float* out, int ldo, int ocol0, int B, int H, int W, int heads, int hd, void* stream) {
  FFSR_CHECK(qkv && anchor && bias1T && bias2T && logit1 && logit2 && out && B > 0 && H % 8 == 0 && W % 8 == 0 && heads > 0);
  dim3 grid((unsigned)((H / 8) * (W / 8) * B * heads));
  switch (hd) {
    case 30: FFSR_LAUNCH(grl_stripe_kernel<30>, grid, dim3(64), 0, ST, qkv, ldq, col0, anchor, lda, bias1T, bias2T, logit1, logit2, out, ldo, ocol0, H, W, heads); break;
    case 10: FFSR_LAUNCH(grl_stripe_kernel<10>, grid, dim3(64), 0, ST, qkv, ldq, col0, anchor, lda, bias1T, bias2T, logit1, logit2, out, ldo, ocol0, H, W, heads); break;
    default: return FFSR_EINVAL;
  }
  return ffsr_launch_status();
}

extern "C" int ffsr_pixel_mha_f32(const float* qkv, int ldq, float* out, int ldo, long long S, int T, int E, int heads,
                                  float p_drop, long long seed,
                                  void* stream) {
  FFSR_CHECK(qkv && out && S > 0 && heads > 0 && E == heads * 16 && (ldq % 4) == 0 && (ldo % 4) == 0);
  FFSR_CHECK(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0);
  FFSR_CHECK(p_drop >= 0.f && p_drop < 1.f);
  const unsigned thr = ffsr_drop_threshold(p_drop);
  const float ks = 1.0f / (1.0f - p_drop);
  long long n = S * T * heads;
  dim3 grid((unsigned)((n + 255) / 256));
  switch (T) {
    case 9: FFSR_LAUNCH(pixel_mha_kernel<9>, grid, dim3(256), 0, ST, qkv, ldq, out, ldo, S, E, heads, thr, ks, (unsigned long long)seed); break;
    case 4: FFSR_LAUNCH(pixel_mha_kernel<4>, grid, dim3(256), 0, ST, qkv, ldq, out, ldo, S, E, heads, thr, ks, (unsigned long long)seed); break;
    default: return FFSR_EINVAL;
  }
  return ffsr_launch_status();
}
