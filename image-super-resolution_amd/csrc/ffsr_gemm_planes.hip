// Implicit-GEMM convolution / token GEMM whose A operand arrives PRE-SPLIT: two bf16 planes (hi = bf16(x),
// lo = bf16(x - hi)) of shape [pixels, Cp] (Cp % 32 == 0, pad channels zero) instead of one fp32 matrix.  Same bytes in
// HBM as fp32, same arithmetic as ffsr_conv2d_bf16x3 (hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16, fp32
// accumulate) -- but the main loop has NO vector ALU work and NO LDS stores: both operands go global -> LDS by LDS-DMA
// (global_load_lds_dwordx4), S stages deep, counted s_waitcnt vmcnt + one raw s_barrier per 32-deep K step.
//
//   out[m, n] = res[m, n] * rvec[n] * rscale + act(sum_k A[m, k] * Wt[n, k] + bias[n]) * cvec[n] * cscale
//   A[m, (tap, c)] = in[pixel(m) + tap offset, c]   (zero page for out-of-image taps / rows >= M)
//
// Tile: BM x BN (BM = 128 / 256, BN = 64 / 128 / 192 / 256), BM/64 x 2 waves, wave tile 64 x BN/2 (2 x TN MFMA tiles of
// 32x32).  Measured and rejected (MI355X, cold caches, tools/planes_bench.py COLD=1): (a) delaying the second-resident
// workgroups by half a tile time (de-phasing the load and store phases of the two workgroups of a CU): no change;
// (b) a persistent role-split kernel -- one 512-thread workgroup per CU walking a tile list, 4 loader waves running
// the LDS-DMA stream 2-3 K steps ahead across tile boundaries, 4 consumer waves doing fragments + MFMA + epilogue:
// correct, 0-10 % SLOWER on every shape (K = 180 token GEMMs and 3x3 convs alike).  With cold caches the K <= 360 token
// GEMMs already run at 57-84 % of what the chip's plain streaming kernels reach on the same bytes (~4 TB/s): what is
// left for them is fewer bytes (fusion), not a different pipeline.  (c) One workgroup walking all column tiles of its
// 128-row panel (the A panel then stays in that CU's L1 / L2): 6-15 % slower -- each tile switch drains the epilogue's
// stores through the shared vmcnt, and the grid loses parallelism.  The bytes each CU pulls through the L2 -> LDS path per MFMA scale with 1/BM + 1/BN; on the
// long-K conv shapes this kernel moves ~15 B/clk/CU through that path at ~36-45 % MFMA utilisation.  That is NOT the
// hardware's fill limit: tools/probe/ldsdma_rate.hip (4 KiB per wave and round, wait + barrier per round) sustains 24 / 47 /
// 70 B/clk/CU from an L2-resident source with 4 / 8 / 16 waves per CU, i.e. the rate grows with the bytes in flight.  With
// two stages the loads of step t+1 have exactly one step of MFMA work (~0.5 us) to land, less than the loaded L2 latency:
// the step time is latency-, not throughput-bound.  Three 32-deep stages of the 64-column tile (72 KB, still 2 WGs/CU) are
// 13 % faster than two on the Cin 180 -> 45 / 60 convs; at BN >= 128 the third stage costs the second resident workgroup
// (measured: slower).  (d) Thinner stages instead -- 16-deep K steps (32-byte LDS rows), 3 / 4 / 5 of them in flight in the
// same 80 KB, correct, burst issue: 13-40 % SLOWER on every shape (token GEMMs, 3x3 convs, the HR refine conv) and slower the
// more stages: halving the MFMA burst between barriers costs more than the longer prefetch distance returns.  Fewer staged
// bytes per step do help (the tap-strip kernel below).
// LDS image of one stage: A_hi | A_lo | B_hi | B_lo, rows of 64 B (32 bf16), no padding: one LDS-DMA instruction
// writes 16 rows x 64 B lane-linear.  Bank conflicts are removed on the SOURCE side: the 16-byte chunk c of row r is
// stored in slot c ^ ((r >> 2) & 3) of its row (each lane fetches the chunk that belongs in its slot), and the
// fragment reads apply the same XOR -> every ds_read_b128 lane group covers 16 distinct 16-byte slots of the bank row.
// Epilogue: accumulator tiles are transposed through the (idle) staging LDS so that every lane owns 8 consecutive
// columns of one row: fp32 output as 16-byte stores and/or bf16 hi/lo planes as 16-byte stores (for the next GEMM).
#include "ffsr_common.h"
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// tools/planes_probe builds this file with -DFFSR_PLANES_PROBE: s_memtime stamps per workgroup (diagnostic only; the
// product library is built without it and carries no stamp code).
#ifdef FFSR_PLANES_PROBE
#define FFSR_STAMP(i)                                                                         \
  do {                                                                                        \
    unsigned long long t_;                                                                    \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                \
    if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 8 + (i)] = t_;            \
  } while (0)
#else
#define FFSR_STAMP(i)
#endif

struct PlaneArgs {
  const unsigned short* a_hi;   // [B*H*W][Cp]
  const unsigned short* a_lo;
  const unsigned short* w_hi;   // [Npad][taps*Cp]
  const unsigned short* w_lo;
  const unsigned short* zeros;  // >= 64 B of zeros
  const float* bias;
  float* out;                   // fp32 output [M][ldo] or null
  const float* res;
  const float* cvec;
  const float* rvec;
  unsigned short* o_hi;         // plane output [M][ldp] or null
  unsigned short* o_lo;
  int B, H, W, Cp;
  int N, Ho, Wo, ldo, ldr, ldp;
  int KH, KW, stride, pad_h, pad_w;
  int act;
  int pre_out;                  // 1: the fp32 output receives the PRE-activation value, the planes the activated one (training: z and GELU(z))
  float slope, cscale, rscale;
  int M, nk;                    // nk = taps * Cp / 32
#ifdef FFSR_PLANES_PROBE
  unsigned long long* stamps;   // [grid][8]
#endif
};

__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& lo) { ffsr_split2(x0, x1, hi, lo); }

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
constexpr int pieces_in_group(int g, int groups, int loads) {
  int n = 0;
  for (int idx = 0; idx < loads; ++idx) n += (idx * groups / loads == g);
  return n;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  else static_assert(N < 0, "add the vmcnt literal");
}

// ---- epilogue of one 128 x BN tile (shared by the kernels below).  C/D layout of the 32x32 MFMA: col = lane & 31,
// row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5).  Per 32x32 tile: bias / activation / column scale in the accumulator
// layout (column = lane), transpose through the 32 x 36-float scratch T of this wave, then lane = (row 16*pass + lane/4,
// columns 8*(lane&3) .. +7): fp32 output as 16-byte stores and / or bf16 hi / lo planes as 16-byte stores.
template <int TM, int TN, bool GELU>
__device__ __forceinline__ void planes_epilogue(const PlaneArgs& p, floatx16 (&acc)[TM][TN], float* T, int m0, int n0, int wrow,
                                                int wcol, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const int erow = lane >> 2, ecol = (lane & 3) * 8;
  const int ldp = p.ldp;
  // wave-uniform: may the 8-column groups use 16-byte accesses?
  const bool vec_ok = (!p.out || ((p.ldo & 3) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0)) &&
                      (!p.res || ((p.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(p.res) & 15) == 0)) &&
                      (!p.rvec || (reinterpret_cast<uintptr_t>(p.rvec) & 15) == 0);
#pragma unroll
  for (int jn = 0; jn < TN; ++jn) {
    const int ncol = n0 + wcol + jn * 32;          // first column of this tile
    if (ncol >= p.N && ncol >= ldp) continue;      // wave-uniform: nothing to write
    const int nc = min(ncol + r, p.N - 1);
    const float bia = p.bias ? p.bias[nc] : 0.f;
    const float cs_ = (p.cvec ? p.cvec[nc] : 1.f) * p.cscale;
    const int nn = ncol + ecol;                    // first of this lane's 8 output columns
    const bool fast = vec_ok && nn + 8 <= p.N;
    floatx4 rs0 = {p.rscale, p.rscale, p.rscale, p.rscale}, rs1 = rs0;
    if (p.res && p.rvec && fast) {
      rs0 *= *reinterpret_cast<const floatx4*>(p.rvec + nn);
      rs1 *= *reinterpret_cast<const floatx4*>(p.rvec + nn + 4);
    }
#pragma unroll
    for (int im = 0; im < TM; ++im) {
      floatx16 v = acc[im][jn];
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] += bia;
      if (!p.pre_out) {                            // (wave-uniform; pre_out: the activation is applied to the plane values below)
        if constexpr (GELU) {
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752440f));
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.slope;
        }
      }
      // The residual rows of this 32 x 32 tile are fetched HERE, before the transposition, unconditionally (row clamped to M - 1; an
      // edge lane reads the row's first 8 columns and ignores them): under the per-row `m < M` / `fast` tests below each load was an
      // exec-masked block of its own, waited for before the next one was issued.  tools/res_cost.py: + res cost 60-68 us per
      // launch where the read itself is 26 us.  (Fetching a whole column tile's rows at once, 32 more live registers, slowed the
      // kernel by 30 % WITHOUT a residual: keep it to the 16 of one tile.)
      const bool res_pre = p.res && vec_ok && p.N >= 8;          // wave-uniform
      floatx4 rv[2][2];
      if (res_pre) {
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
          const int m = min(m0 + wrow + im * 32 + pass * 16 + erow, p.M - 1);
          const float* rp = p.res + (size_t)m * p.ldr + (fast ? nn : 0);
          rv[pass][0] = *reinterpret_cast<const floatx4*>(rp);
          rv[pass][1] = *reinterpret_cast<const floatx4*>(rp + 4);
        }
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) T[((e & 3) + 8 * (e >> 2) + 4 * h) * 36 + r] = v[e] * cs_;
      // (the same wave wrote and reads T: LDS operations of one wave complete in order)
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int row = pass * 16 + erow;
        const int m = m0 + wrow + im * 32 + row;
        if (m >= p.M) continue;
        const float* tp = T + row * 36 + ecol;
        if (fast) {
          floatx4 o0 = *reinterpret_cast<const floatx4*>(tp), o1 = *reinterpret_cast<const floatx4*>(tp + 4);
          if (res_pre) {
            o0 += rv[pass][0] * rs0;
            o1 += rv[pass][1] * rs1;
          } else if (p.res) {
            const float* rp = p.res + (size_t)m * p.ldr + nn;
            o0 += *reinterpret_cast<const floatx4*>(rp) * rs0;
            o1 += *reinterpret_cast<const floatx4*>(rp + 4) * rs1;
          }
          if (p.out) {
            float* op = p.out + (size_t)m * p.ldo + nn;
            // (non-temporal stores here and on the planes below, round 2: the planes kernel's mean launch 225 -> 276 us,
            //  the whole 340x510 step 365 -> 372 ms -- the write-allocating L2 path is the faster one for these tiles)
            *reinterpret_cast<floatx4*>(op) = o0;
            *reinterpret_cast<floatx4*>(op + 4) = o1;
          }
          if (p.o_hi) {
            if (p.pre_out) {
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                if constexpr (GELU) {
                  o0[c] = 0.5f * o0[c] * (1.0f + erff(o0[c] * 0.70710678118654752440f));
                  o1[c] = 0.5f * o1[c] * (1.0f + erff(o1[c] * 0.70710678118654752440f));
                } else {
                  o0[c] = o0[c] > 0.f ? o0[c] : o0[c] * p.slope;
                  o1[c] = o1[c] > 0.f ? o1[c] : o1[c] * p.slope;
                }
              }
            }
            unsigned hh[4], ll[4];
            split2(o0[0], o0[1], hh[0], ll[0]);
            split2(o0[2], o0[3], hh[1], ll[1]);
            split2(o1[0], o1[1], hh[2], ll[2]);
            split2(o1[2], o1[3], hh[3], ll[3]);
            const uintx4 hi4 = {hh[0], hh[1], hh[2], hh[3]}, lo4 = {ll[0], ll[1], ll[2], ll[3]};
            *reinterpret_cast<uintx4*>(p.o_hi + (size_t)m * ldp + nn) = hi4;
            *reinterpret_cast<uintx4*>(p.o_lo + (size_t)m * ldp + nn) = lo4;
          }
        } else {
          // edge / unaligned columns: one pair at a time (rare: N % 8 != 0 tails, odd strides)
#pragma nounroll
          for (int c = 0; c < 8; c += 2) {
            float x[2];
#pragma unroll
            for (int d = 0; d < 2; ++d) {
              const int n = nn + c + d;
              float o = 0.f;
              if (n < p.N) {
                o = tp[c + d];
                if (p.res) o += p.res[(size_t)m * p.ldr + n] * (p.rvec ? p.rvec[n] : 1.f) * p.rscale;
                if (p.out) p.out[(size_t)m * p.ldo + n] = o;
              }
              if (p.pre_out) o = GELU ? 0.5f * o * (1.0f + erff(o * 0.70710678118654752440f)) : (o > 0.f ? o : o * p.slope);
              x[d] = o;
            }
            if (p.o_hi && nn + c < ldp) {   // ldp is even: pairs are whole; columns >= N are written as zeros
              unsigned hh, ll;
              split2(x[0], x[1], hh, ll);
              *reinterpret_cast<unsigned*>(p.o_hi + (size_t)m * ldp + nn + c) = hh;
              *reinterpret_cast<unsigned*>(p.o_lo + (size_t)m * ldp + nn + c) = ll;
            }
          }
        }
      }
    }
  }
}

// GELU: exact-erf GELU epilogue; otherwise the epilogue activation is v > 0 ? v : v * slope (none: slope 1, ReLU: 0).
// SCHED 1: the LDS-DMA pieces of the next K step are issued one at a time between the MFMA triples of the current step
// (a burst of 10 pieces right after the barrier costs 100-185 issue cycles each with the matrix pipe idle).
// TERMS 3: hi*hi + hi*lo + lo*hi.  TERMS 1 (plain-bf16 mode): only the hi planes are staged (half the LDS-DMA pieces, half the
// operand bytes) and each product is one MFMA.
template <int BM, int BN, int STAGES, bool GELU, int SCHED, int TERMS>
__global__ __launch_bounds__(BM * 2) void conv_gemm_planes_kernel(PlaneArgs p) {
  FFSR_STAMP(0);
  constexpr int NW = BM / 32;                                // waves: (BM / 64) x 2
  constexpr int TM = 2, TN = BN / 64;
  constexpr int PLANE_A = BM * 64, PLANE_B = BN * 64;        // bytes of one plane of one stage
  constexpr int STAGE = 2 * (PLANE_A + PLANE_B);
  constexpr int NB = (BN / 16 + NW - 1) / NW;                // B pieces (LDS-DMA instructions) per wave and plane
  constexpr bool B_EVEN = (BN / 16) % NW == 0;               // every wave owns NB pieces (else the last round is partial)
  constexpr int PLN = TERMS == 3 ? 2 : 1;                    // planes staged per operand
  constexpr int LOADS = PLN * (2 + NB);                      // LDS-DMA instructions per wave and K step (full waves)
  static_assert(STAGES == 2 || B_EVEN, "counted vmcnt needs the same number of loads in every wave");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // STAGES * STAGE bytes (>= 4 * 9216 for the epilogue)

  const int nwg = gridDim.x;
  const int orig = blockIdx.x;
  const int q8 = nwg >> 3, rr = nwg & 7, xcd = orig & 7;
  const int tile = (xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8) + (orig >> 3);
  const int ntn = (p.N + BN - 1) / BN;
  const int m0 = (tile / ntn) * BM;
  const int n0 = (tile % ntn) * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane >> 2;                                // row inside a 16-row LDS-DMA piece

  // ---- A loader state: this lane fetches rows 32*wave + lrow and + 16 (pieces 2*wave, 2*wave + 1) of both planes
  const int ntap = p.KH * p.KW;
  const int HoWo = p.Ho * p.Wo;
  unsigned a_off[2];     // byte offset of (pixel of tap (0,0), chunk slot of this lane) in a plane
  unsigned a_mask[2];    // bit t = tap t is inside the image
  const ptrdiff_t lo_delta = reinterpret_cast<const unsigned char*>(p.a_lo) - reinterpret_cast<const unsigned char*>(p.a_hi);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 32 * wave + 16 * i + lrow;
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
    const int m = m0 + row;
    a_off[i] = 0;
    a_mask[i] = 0;
    if (m < p.M) {
      const int b = m / HoWo, rem = m - b * HoWo;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      const int iy0 = oy * p.stride - p.pad_h, ix0 = ox * p.stride - p.pad_w;
      a_off[i] = (unsigned)(((b * p.H + iy0) * p.W + ix0) * p.Cp + chunk * 8) * 2u;   // may wrap for negative iy0/ix0: only used when the tap is valid
      unsigned mk = 0;
      for (int t = 0; t < ntap; ++t) {
        const int yy = iy0 + t / p.KW, xx = ix0 + t % p.KW;
        if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) mk |= 1u << t;
      }
      a_mask[i] = mk;
    }
  }
  // ---- B loader state: rows 16*(wave + NW*j) + lrow of the BN-row tile
  unsigned b_off[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int row = 16 * (wave + NW * j) + lrow;
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
    b_off[j] = (unsigned)((n0 + row) * (p.nk * 32) + chunk * 8) * 2u;
  }

  // issue position (advanced once per issued K step): tap, 32-channel chunk inside the tap, tap pixel offset
  int i_tap = 0, i_cc = 0, i_ky = 0, i_kx = 0;
  const int cpt = p.Cp >> 5;
  unsigned i_koff = 0;   // byte offset of K step in a weight row
  // one LDS-DMA piece of K step (i_tap, i_cc): idx 0..3 = A (row block idx>>1, plane idx&1), 4.. = B (piece (idx-4)>>1, plane)
  auto issue_piece = [&](int stage, int li, bool valid) {
    const int idx = TERMS == 3 ? li : 2 * li;      // hi-only staging: the even (hi-plane) pieces of the 3-term numbering
    unsigned char* base = smem + stage * STAGE;
    if (idx < 4) {
      const int i = idx >> 1;
      const unsigned toff = (unsigned)((i_ky * p.W + i_kx) * p.Cp + i_cc * 32) * 2u;
      const bool ok = valid && ((a_mask[i] >> i_tap) & 1u);
      const unsigned char* src = reinterpret_cast<const unsigned char*>(p.a_hi) + (size_t)(a_off[i] + toff);
      if (idx & 1) src += lo_delta;
      src = ok ? src : reinterpret_cast<const unsigned char*>(p.zeros);
      unsigned char* dst = base + (2 * wave + i) * 1024 + (idx & 1) * PLANE_A;
      __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)dst, 16, 0, 0);
    } else {
      const int j = (idx - 4) >> 1;
      if (!B_EVEN && j == NB - 1 && wave + NW * j >= BN / 16) return;   // wave-uniform
      const size_t o = (size_t)b_off[j] + i_koff;
      const unsigned char* src = reinterpret_cast<const unsigned char*>((idx & 1) ? p.w_lo : p.w_hi) + o;
      src = valid ? src : reinterpret_cast<const unsigned char*>(p.zeros);
      unsigned char* dst = base + 2 * PLANE_A + (wave + NW * j) * 1024 + (idx & 1) * PLANE_B;
      __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)dst, 16, 0, 0);
    }
  };
  auto advance = [&]() {
    i_koff += 64;
    if (++i_cc == cpt) {
      i_cc = 0;
      ++i_tap;
      if (++i_kx == p.KW) { i_kx = 0; ++i_ky; }
    }
  };
  auto issue = [&](int stage, bool valid) {
#pragma unroll
    for (int idx = 0; idx < LOADS; ++idx) issue_piece(stage, idx, valid);
    advance();
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int wrow = (wave >> 1) * 64, wcol = (wave & 1) * (BN / 2);
  const int r = lane & 31, h = lane >> 5;
  const int swz = (r >> 2) & 3;
  const int fo0 = r * 64 + (((0 + h) ^ swz) << 4);   // fragment byte offsets inside a 32-row block, K half 0 / 1
  const int fo1 = r * 64 + (((2 + h) ^ swz) << 4);
  const int nk = p.nk;

  FFSR_STAMP(1);   // address set-up done
  // ---- prologue: STAGES - 1 K steps in flight
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s) issue(s, s < nk);   // (steps past nk fetch the zero page: same vmcnt bookkeeping)

  int cs = 0;                 // stage being computed
  int is = STAGES - 1;        // stage being filled
  for (int kt = 0; kt < nk; ++kt) {
    // retire K step kt: STAGES - 2 younger steps stay in flight (steps past nk are zero-page fetches)
    wait_vmcnt<(STAGES - 2) * LOADS>();
    __builtin_amdgcn_s_barrier();   // every wave's pieces of step kt have landed; every wave is done reading stage `is`
    if (kt == 0) FFSR_STAMP(2);     // first tile landed
    if (kt == 1) FFSR_STAMP(3);     // first K step computed + second tile landed
    const bool more = kt + STAGES - 1 < nk;
    if (SCHED == 0) issue(is, more);

    const unsigned char* A = smem + cs * STAGE + wrow * 64;
    const unsigned char* Bt = smem + cs * STAGE + 2 * PLANE_A + wcol * 64;
    if (++cs == STAGES) cs = 0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int fo = ks ? fo1 : fo0;
      bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = *reinterpret_cast<const bf16x8*>(A + i * 2048 + fo);
        if constexpr (TERMS == 3) al[i] = *reinterpret_cast<const bf16x8*>(A + PLANE_A + i * 2048 + fo);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const bf16x8*>(Bt + j * 2048 + fo);
        if constexpr (TERMS == 3) bl[j] = *reinterpret_cast<const bf16x8*>(Bt + PLANE_B + j * 2048 + fo);
      }
      static_for<0, TM * TN>([&](auto tc) {
        constexpr int t = decltype(tc)::value, i = t / TN, j = t % TN;
        if constexpr (TERMS == 3) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        }
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        if constexpr (SCHED == 1) {
          constexpr int GROUPS = 2 * TM * TN;
          const int g = ks * TM * TN + t;                       // MFMA triple index inside the K step
          // spread the LOADS pieces over the GROUPS triples
#pragma unroll
          for (int idx = 0; idx < LOADS; ++idx)
            if (idx * GROUPS / LOADS == g) issue_piece(is, idx, more);
          __builtin_amdgcn_sched_group_barrier(0x8, TERMS, 0);  // the product's MFMAs, then its pieces
          constexpr int c0 = pieces_in_group(t, GROUPS, LOADS), c1 = pieces_in_group(TM * TN + t, GROUPS, LOADS);
          if (ks == 0) { if constexpr (c0 > 0) __builtin_amdgcn_sched_group_barrier(0x20, c0, 0); }
          else { if constexpr (c1 > 0) __builtin_amdgcn_sched_group_barrier(0x20, c1, 0); }
        }
      });
    }
    if (SCHED == 1) advance();
    if (++is == STAGES) is = 0;
  }
  FFSR_STAMP(4);                  // main loop done
  wait_vmcnt<0>();                // zero-page fetches of the last steps
  __builtin_amdgcn_s_barrier();   // all fragment reads done: the staging LDS becomes the transpose scratch

  // ---- epilogue (the staging LDS is idle now: it serves as the transpose scratch)
  planes_epilogue<TM, TN, GELU>(p, acc, reinterpret_cast<float*>(smem) + wave * (32 * 36), m0, n0, wrow, wcol, lane);
  FFSR_STAMP(5);                  // epilogue done (stores issued)
}

template <int BM, int BN, int STAGES, bool GELU, int SCHED, int TERMS>
int launch_planes4(const PlaneArgs& a, hipStream_t st) {
  constexpr int STAGE = 2 * (BM + BN) * 64;
  static_assert(STAGES * STAGE >= (BM / 32) * 32 * 36 * 4, "epilogue scratch");
  static_assert(STAGES * STAGE <= 160 * 1024, "LDS");
  const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  static unsigned long long attr_set = 0;
  const void* fn = reinterpret_cast<const void*>(&conv_gemm_planes_kernel<BM, BN, STAGES, GELU, SCHED, TERMS>);
  if (ffsr_allow_dynamic_lds(&fn, 1, STAGES * STAGE, &attr_set) != FFSR_OK) return FFSR_ELAUNCH;
  FFSR_LAUNCH((conv_gemm_planes_kernel<BM, BN, STAGES, GELU, SCHED, TERMS>), dim3(tiles), dim3(BM * 2), STAGES * STAGE, st, a);
  return ffsr_launch_status();
}

template <int BM, int BN, int STAGES, bool GELU, int SCHED>
int launch_planes3(const PlaneArgs& a, hipStream_t st) {
  return g_ffsr_gemm_terms == 1 ? launch_planes4<BM, BN, STAGES, GELU, SCHED, 1>(a, st) : launch_planes4<BM, BN, STAGES, GELU, SCHED, 3>(a, st);
}

#ifdef FFSR_PLANES_PROBE
unsigned long long* g_probe_stamps = nullptr;
#endif
int g_planes_sched = 1;   // 0 = burst issue (stages + 10, diagnostic), 1 = interleaved issue

template <int BM, int BN, int STAGES, bool GELU>
int launch_planes2(const PlaneArgs& a, hipStream_t st) {
  return g_planes_sched ? launch_planes3<BM, BN, STAGES, GELU, 1>(a, st) : launch_planes3<BM, BN, STAGES, GELU, 0>(a, st);
}

template <int BM, int BN, int STAGES>
int launch_planes(PlaneArgs& a, hipStream_t st) {
  if (a.act == FFSR_ACT_GELU) return launch_planes2<BM, BN, STAGES, true>(a, st);
  if (a.act == FFSR_ACT_NONE) a.slope = 1.f;
  else if (a.act == FFSR_ACT_RELU) a.slope = 0.f;
  else if (a.act != FFSR_ACT_LRELU) return FFSR_EINVAL;   // sigmoid / SiLU epilogues stay on ffsr_conv2d_bf16x3
  return launch_planes2<BM, BN, STAGES, false>(a, st);
}

// ---- 3x3 (stride 1, pad 1) convolutions: the three horizontal taps share their A rows.
// The tile kernel above fetches a fresh 128-row A tile per tap: the time to land a step's staged bytes (24..32 KB per K step
// against 0.5..1 us of MFMA at N = 45 / 60 / 128), not the matrix pipe, bounds the thin-N 3x3 convs.  Output rows are consecutive pixels, so the A rows of tap (ky, kx) are the rows of tap (ky, 0) moved
// down by kx: one 130-row strip (144 staged) per (ky, 32-channel chunk) serves kx = 0, 1, 2 through fragment reads at row
// offsets 0 / 1 / 2 -- 9 pieces per plane instead of 24.  What a shifted read picks up across an image border (the
// neighbouring pixel of the previous / next image row) is zeroed in the fragment registers from the per-row tap mask.
// TERMS 1 (plain-bf16 mode): one MFMA per product on the hi planes (both planes are still staged here).
template <int BN, bool GELU, int SCHED, int TERMS>
__global__ __launch_bounds__(256) void conv3_strip_planes_kernel(PlaneArgs p) {
  constexpr int BM = 128, TM = 2, TN = BN / 64, NW = 4;
  constexpr int SROWS = 144;                                  // 130 used, staged in 16-row pieces
  constexpr int APL = SROWS * 64, ASTRIP = 2 * APL;           // bytes: one plane, one strip (hi | lo)
  constexpr int PLANE_B = BN * 64, BST = 2 * PLANE_B;
  constexpr int NB = BN / 16 / NW;                            // B pieces per wave and plane
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 * ASTRIP | 2 * BST
  unsigned char* const Abase = smem;
  unsigned char* const Bbase = smem + 2 * ASTRIP;

  const int nwg = gridDim.x;
  const int orig = blockIdx.x;
  const int q8 = nwg >> 3, rr = nwg & 7, xcd = orig & 7;
  const int tile = (xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8) + (orig >> 3);
  const int ntn = (p.N + BN - 1) / BN;
  const int m0 = (tile / ntn) * BM;
  const int n0 = (tile % ntn) * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane >> 2;
  const int cpt = p.Cp >> 5;
  const long long npix = (long long)p.B * p.H * p.W;
  const ptrdiff_t lo_delta = reinterpret_cast<const unsigned char*>(p.a_lo) - reinterpret_cast<const unsigned char*>(p.a_hi);

  // strip piece pc (16 rows) of super-step (ky, cc) -> strip buffer `buf`: strip row s holds input pixel m0 - 1 + s + (ky - 1) W
  auto issue_a1 = [&](int buf, int pc, int ky, int cc, int plane) {
    const int srow = 16 * pc + lrow;
    const long long q = (long long)m0 - 1 + srow + (long long)(ky - 1) * p.W;
    const int chunk = (lane & 3) ^ ((srow >> 2) & 3);
    const bool ok = q >= 0 && q < npix;
    const unsigned char* src = reinterpret_cast<const unsigned char*>(p.a_hi) + ((size_t)(ok ? q : 0) * p.Cp + cc * 32 + chunk * 8) * 2;
    if (plane) src += lo_delta;
    unsigned char* dst = Abase + buf * ASTRIP + pc * 1024 + plane * APL;
    __builtin_amdgcn_global_load_lds(ok ? src : reinterpret_cast<const unsigned char*>(p.zeros), (lds_ptr_t)dst, 16, 0, 0);
  };
  auto issue_a = [&](int buf, int pc, int ky, int cc) {
    issue_a1(buf, pc, ky, cc, 0);
    issue_a1(buf, pc, ky, cc, 1);
  };
  unsigned b_off[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int row = 16 * (wave + NW * j) + lrow;
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
    b_off[j] = (unsigned)((n0 + row) * (p.nk * 32) + chunk * 8) * 2u;
  }
  auto issue_b1 = [&](int buf, unsigned koff, int j, int plane) {
    unsigned char* dst = Bbase + buf * BST + (wave + NW * j) * 1024 + plane * PLANE_B;
    const size_t o = (size_t)b_off[j] + koff;
    __builtin_amdgcn_global_load_lds(reinterpret_cast<const unsigned char*>(plane ? p.w_lo : p.w_hi) + o, (lds_ptr_t)dst, 16, 0, 0);
  };
  auto issue_b = [&](int buf, int ky, int kx, int cc) {
    const unsigned koff = (unsigned)(((ky * 3 + kx) * cpt + cc) * 64);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      issue_b1(buf, koff, j, 0);
      issue_b1(buf, koff, j, 1);
    }
  };

  const int wrow = (wave >> 1) * 64, wcol = (wave & 1) * (BN / 2);
  const int r = lane & 31, h = lane >> 5;
  // per fragment row: bit t of the mask = tap t reads inside the image
  unsigned fmask[TM];
  const int HW = p.H * p.W;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wrow + i * 32 + r;
    unsigned mk = 0;
    if (m < p.M) {
      const int rem = m % HW;
      const int oy = rem / p.W, ox = rem - oy * p.W;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = oy + t / 3 - 1, xx = ox + t % 3 - 1;
        if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) mk |= 1u << t;
      }
    }
    fmask[i] = mk;
  }
  const bool any_masked = __any((fmask[0] & fmask[1]) != 0x1ffu);   // wave-uniform: interior tiles skip the masking

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int swz = (r >> 2) & 3;
  const int bfo0 = r * 64 + (((0 + h) ^ swz) << 4), bfo1 = r * 64 + (((2 + h) ^ swz) << 4);

  // prologue: strip of super-step 0 (pieces dealt round-robin to the waves) and the weights of step 0
  for (int pc = wave; pc < SROWS / 16; pc += NW) issue_a(0, pc, 0, 0);
  issue_b(0, 0, 0, 0);

  const int nss = 3 * cpt;          // super-steps (ky, cc); 3 K steps (kx) each
  int ky = 0, cc = 0;
  for (int ss = 0; ss < nss; ++ss) {
    int nky = ky, ncc = cc + 1;
    if (ncc == cpt) { ncc = 0; ++nky; }
    const bool more_ss = ss + 1 < nss;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();   // this step's strip / weights have landed; the buffers refilled below are no longer read
      const int t = 3 * ss + kx;
      // next step's weights; the next super-step's strip is fetched in thirds (pieces 0-3, 4-7, 8) while this one is consumed
      const bool has_b = kx < 2 || more_ss;
      const int bky = kx < 2 ? ky : nky, bkx = kx < 2 ? kx + 1 : 0, bcc = kx < 2 ? cc : ncc;
      const unsigned bkoff = (unsigned)(((bky * 3 + bkx) * cpt + bcc) * 64);
      const int apc = 4 * kx + wave;
      const bool has_a = more_ss && apc < SROWS / 16;
      constexpr int LOADS = 2 * NB + 2;
      auto issue_piece = [&](int idx) {      // 0 .. 2 NB - 1: weight pieces; 2 NB, 2 NB + 1: the strip piece's planes
        if (idx < 2 * NB) {
          if (has_b) issue_b1((t + 1) & 1, bkoff, idx >> 1, idx & 1);
        } else if (has_a) {
          issue_a1((ss + 1) & 1, apc, nky, ncc, idx - 2 * NB);
        }
      };
      if constexpr (SCHED == 0) {
#pragma unroll
        for (int idx = 0; idx < LOADS; ++idx) issue_piece(idx);
      }
      const unsigned char* A = Abase + (ss & 1) * ASTRIP;
      const unsigned char* Bt = Bbase + (t & 1) * BST + wcol * 64;
      const int tap = ky * 3 + kx;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int s = wrow + i * 32 + r + kx;
          const int fo = s * 64 + (((2 * ks + h) ^ ((s >> 2) & 3)) << 4);
          ah[i] = *reinterpret_cast<const bf16x8*>(A + fo);
          al[i] = *reinterpret_cast<const bf16x8*>(A + APL + fo);
          if (any_masked && !((fmask[i] >> tap) & 1u)) {
            const uintx4 z = {0u, 0u, 0u, 0u};
            ah[i] = __builtin_bit_cast(bf16x8, z);
            al[i] = __builtin_bit_cast(bf16x8, z);
          }
        }
        const int bfo = ks ? bfo1 : bfo0;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          bh[j] = *reinterpret_cast<const bf16x8*>(Bt + j * 2048 + bfo);
          bl[j] = *reinterpret_cast<const bf16x8*>(Bt + PLANE_B + j * 2048 + bfo);
        }
        static_for<0, TM * TN>([&](auto tc) {
          constexpr int tt = decltype(tc)::value, i = tt / TN, j = tt % TN;
          if constexpr (TERMS == 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          if constexpr (SCHED == 1) {
            // one LDS-DMA piece after each MFMA triple instead of a burst behind the barrier
            constexpr int GROUPS = 2 * TM * TN;
            const int g = ks * TM * TN + tt;
#pragma unroll
            for (int idx = 0; idx < LOADS; ++idx)
              if (idx * GROUPS / LOADS == g) issue_piece(idx);
            __builtin_amdgcn_sched_group_barrier(0x8, TERMS, 0);
            constexpr int c0 = pieces_in_group(tt, GROUPS, LOADS), c1 = pieces_in_group(TM * TN + tt, GROUPS, LOADS);
            if (ks == 0) { if constexpr (c0 > 0) __builtin_amdgcn_sched_group_barrier(0x20, c0, 0); }
            else { if constexpr (c1 > 0) __builtin_amdgcn_sched_group_barrier(0x20, c1, 0); }
          }
        });
      }
    }
    ky = nky;
    cc = ncc;
  }
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  planes_epilogue<TM, TN, GELU>(p, acc, reinterpret_cast<float*>(smem) + wave * (32 * 36), m0, n0, wrow, wcol, lane);
}

template <int BN, bool GELU, int SCHED, int TERMS>
int launch_strip4(const PlaneArgs& a, hipStream_t st) {
  constexpr int LDS = 2 * (2 * 144 * 64) + 2 * (2 * BN * 64);
  static_assert(LDS >= 4 * 32 * 36 * 4, "epilogue scratch");
  const int tiles = ((a.M + 127) / 128) * ((a.N + BN - 1) / BN);
  static unsigned long long attr_set = 0;
  const void* fn = reinterpret_cast<const void*>(&conv3_strip_planes_kernel<BN, GELU, SCHED, TERMS>);
  if (ffsr_allow_dynamic_lds(&fn, 1, LDS, &attr_set) != FFSR_OK) return FFSR_ELAUNCH;
  FFSR_LAUNCH((conv3_strip_planes_kernel<BN, GELU, SCHED, TERMS>), dim3(tiles), dim3(256), LDS, st, a);
  return ffsr_launch_status();
}
template <int BN, bool GELU, int SCHED>
int launch_strip3(const PlaneArgs& a, hipStream_t st) {
  return g_ffsr_gemm_terms == 1 ? launch_strip4<BN, GELU, SCHED, 1>(a, st) : launch_strip4<BN, GELU, SCHED, 3>(a, st);
}
// Measured (tools/strip_bench.py): with the 64-column tile (6 MFMAs per K half) the burst of 4 pieces behind the barrier is
// faster than one piece per MFMA triple (N 45: 132 vs 147 us, N 60: 156 vs 161, 64 -> 64 at HR/2: 187 vs 201); with the
// 128-column tile the interleaved issue is marginally ahead (212 vs 216 us, 2274 vs 2284 us).  A third weight buffer
// (weights fetched two steps ahead behind a counted vmcnt, 60 KB = 2 WGs/CU instead of 3): bit-identical, no change
// (157 vs 156 us, 144.5 vs 142.8 us) -- not kept.
template <int BN, bool GELU>
int launch_strip2(const PlaneArgs& a, hipStream_t st) {
  return launch_strip3<BN, GELU, (BN >= 128 ? 1 : 0)>(a, st);
}

template <int BN>
int launch_strip(PlaneArgs& a, hipStream_t st) {
  if (a.act == FFSR_ACT_GELU) return launch_strip2<BN, true>(a, st);
  if (a.act == FFSR_ACT_NONE) a.slope = 1.f;
  else if (a.act == FFSR_ACT_RELU) a.slope = 0.f;
  else if (a.act != FFSR_ACT_LRELU) return FFSR_EINVAL;
  return launch_strip2<BN, false>(a, st);
}

// fp32 [M, C] (row stride ldx) -> bf16 hi / lo planes [M, ldp] (columns C..ldp-1 zero).  One thread = 8 columns.
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, int ldx, unsigned short* __restrict__ hi,
                                                           unsigned short* __restrict__ lo, int ldp, long long M, int C) {
  const int groups = ldp >> 3;
  const long long total = M * groups;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long m = i / groups;
    const int c0 = (int)(i - m * groups) * 8;
    const float* xp = x + m * ldx + c0;
    float v[8];
    if (c0 + 8 <= C && (ldx & 3) == 0) {
      const floatx4 a = *reinterpret_cast<const floatx4*>(xp), b = *reinterpret_cast<const floatx4*>(xp + 4);
#pragma unroll
      for (int c = 0; c < 4; ++c) { v[c] = a[c]; v[4 + c] = b[c]; }
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = c0 + c < C ? xp[c] : 0.f;
    }
    uintx4 h4, l4;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      unsigned hh, ll;
      split2(v[2 * c], v[2 * c + 1], hh, ll);
      h4[c] = hh;
      l4[c] = ll;
    }
    *reinterpret_cast<uintx4*>(hi + m * ldp + c0) = h4;
    *reinterpret_cast<uintx4*>(lo + m * ldp + c0) = l4;
  }
}

}  // namespace

// See include/ffsr.h for the contract.
extern "C" int ffsr_conv2d_planes(const void* a_hi, const void* a_lo, int Cp, const void* wgt_hi, const void* wgt_lo,
                                  int n_rows_padded, const void* zeros, const float* bias, float* out, const float* res,
                                  const float* cvec, const float* rvec, void* out_hi, void* out_lo, int ldp, int B, int H,
                                  int W, int N, int ldo, int ldr, int KH, int KW, int stride, int pad_h, int pad_w, int act,
                                  float slope, float cscale, float rscale, int bm, int bn, int stages, void* stream) {
  FFSR_CHECK(a_hi && a_lo && wgt_hi && wgt_lo && zeros && (out || (out_hi && out_lo)));
  FFSR_CHECK(B > 0 && H > 0 && W > 0 && N > 0 && KH > 0 && KW > 0 && KH * KW <= 32 && stride > 0);
  FFSR_CHECK(Cp > 0 && (Cp & 31) == 0);
  FFSR_CHECK(((uintptr_t)a_hi & 15) == 0 && ((uintptr_t)a_lo & 15) == 0 && ((uintptr_t)wgt_hi & 15) == 0 &&
             ((uintptr_t)wgt_lo & 15) == 0 && ((uintptr_t)zeros & 15) == 0);
  FFSR_CHECK(bn == 64 || bn == 128 || bn == 192 || bn == 256);
  FFSR_CHECK(bm == 0 || bm == 128 || bm == 256);
  FFSR_CHECK(n_rows_padded % bn == 0 && n_rows_padded >= N);
  FFSR_CHECK(!out || (ldo >= N));
  FFSR_CHECK(!res || ldr >= N);
  FFSR_CHECK(!out_hi || (out_lo && (ldp & 31) == 0 && ldp >= N && ldp < N + 32 && ((uintptr_t)out_hi & 15) == 0 &&
                         ((uintptr_t)out_lo & 15) == 0));
  PlaneArgs a;
  a.a_hi = (const unsigned short*)a_hi; a.a_lo = (const unsigned short*)a_lo;
  a.w_hi = (const unsigned short*)wgt_hi; a.w_lo = (const unsigned short*)wgt_lo; a.zeros = (const unsigned short*)zeros;
  a.bias = bias; a.out = out; a.res = res; a.cvec = cvec; a.rvec = rvec;
  a.o_hi = (unsigned short*)out_hi; a.o_lo = (unsigned short*)out_lo;
  a.B = B; a.H = H; a.W = W; a.Cp = Cp; a.N = N;
  a.Ho = (H + 2 * pad_h - KH) / stride + 1;
  a.Wo = (W + 2 * pad_w - KW) / stride + 1;
  FFSR_CHECK(a.Ho > 0 && a.Wo > 0);
  a.ldo = ldo; a.ldr = ldr; a.ldp = ldp; a.KH = KH; a.KW = KW; a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w;
  a.pre_out = (act & 0x100) ? 1 : 0;
  act &= 0xff;
  // (act | 0x100: fp32 output = pre-activation, planes = activated; no residual / scales in that mode)
  FFSR_CHECK(!a.pre_out || (out && out_hi && !res && !cvec && cscale == 1.0f));
  a.act = act; a.slope = slope; a.cscale = cscale; a.rscale = rscale;
  const long long M = (long long)B * a.Ho * a.Wo;
  FFSR_CHECK(M < (1ll << 31));
  FFSR_CHECK((long long)B * H * W * Cp * 2 < (1ll << 32));                       // 32-bit byte offsets into the A planes
  FFSR_CHECK((long long)n_rows_padded * KH * KW * Cp * 2 < (1ll << 32));         // ... and into the weight planes
  a.M = (int)M;
  a.nk = KH * KW * (Cp / 32);
#ifdef FFSR_PLANES_PROBE
  a.stamps = g_probe_stamps;
#endif
  hipStream_t st = (hipStream_t)stream;
  if (bm == 0) bm = 128;
  if (stages == 0) stages = 2;
  if (stages == 4) {   // tap-strip variant: 3x3, stride 1, pad 1 only
    FFSR_CHECK(KH == 3 && KW == 3 && stride == 1 && pad_h == 1 && pad_w == 1 && bm == 128 && (bn == 64 || bn == 128));
    FFSR_CHECK((long long)B * H * W * Cp * 2 < (1ll << 40));
    return bn == 64 ? launch_strip<64>(a, st) : launch_strip<128>(a, st);
  }
  if (stages >= 10) { g_planes_sched = 0; stages -= 10; } else g_planes_sched = 1;   // stages + 10: burst-issue variant (diagnostic)
  switch ((bm / 128) * 10000 + bn * 10 + stages) {
    case 10642: return launch_planes<128, 64, 2>(a, st);
    case 10643: return launch_planes<128, 64, 3>(a, st);
    case 11282: return launch_planes<128, 128, 2>(a, st);
    case 11283: return launch_planes<128, 128, 3>(a, st);
    case 11922: return launch_planes<128, 192, 2>(a, st);
    case 21282: return launch_planes<256, 128, 2>(a, st);
    case 21283: return launch_planes<256, 128, 3>(a, st);
    case 21922: return launch_planes<256, 192, 2>(a, st);
    case 22562: return launch_planes<256, 256, 2>(a, st);
    default: return FFSR_EINVAL;
  }
}

extern "C" int ffsr_split_planes(const float* x, int ldx, void* hi, void* lo, int ldp, long long M, int C, void* stream) {
  FFSR_CHECK(x && hi && lo && M > 0 && C > 0 && (ldp & 31) == 0 && ldp >= C && ldx >= C);
  FFSR_CHECK(((uintptr_t)hi & 15) == 0 && ((uintptr_t)lo & 15) == 0 && ((uintptr_t)x & 15) == 0);
  const long long total = M * (ldp >> 3);
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  FFSR_LAUNCH(split_planes_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, (unsigned short*)hi,
                     (unsigned short*)lo, ldp, M, C);
  return ffsr_launch_status();
}

#ifdef FFSR_PLANES_PROBE
extern "C" void ffsr_planes_probe_set(unsigned long long* stamps) { g_probe_stamps = stamps; }
#endif
