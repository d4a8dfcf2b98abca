// Mamba selective scan (S6) for the 4-direction SS2D of MambaIR, gfx950.
//
//   delta_t = softplus(dt_bias[k,d] + sum_r dtw[k,d,r] * dts_t[k,r])        (dt projection fused in)
//   h_t[n]  = exp(delta_t * A[k,d,n]) * h_{t-1}[n] + delta_t * B_t[k,n] * u_t[d]
//   y_t[d]  = sum_n C_t[k,n] * h_t[n] + D[k,d] * u_t[d]
//
// Layout (chosen for coalescing, not the reference's [B, K*D, L]): u is pixel-major [L, Dm] so a wave reads 64
// consecutive channels of one token; the per-token projections xdbl [L, 4*(R+2N)] are wave-uniform.  The four
// traversals (row-major, column-major and their reverses, mambair_arch.py:343-344) are index maps t -> pixel, so no
// gathered copy of u is ever materialised, and every direction writes its y back at the pixel it belongs to
// (the inverse scatter of mambair_arch.py:365-369); the 4 partial outputs are summed by the fused norm/gate kernel.
//
// Parallelisation: lane = channel d (16 states in registers); the sequence is cut into chunks:
//   pass A: per chunk, local end state with h_in = 0 and the chunk's total decay exp(A * sum delta)
//   pass B: serial carry over chunks (tiny)            pass C: per chunk, recurrence from the true h_in, emits y.
//
// What bounds it (rocprofv3 --pmc at L = 180224, Dm = 360, tools/scan_pmc.sh): the chunk kernels are vector-ALU bound.
// One step of one wave is ~104 VALU instructions (87 at 4 issue cycles + 17 v_exp/v_log at 8 = SQ_ACTIVE_INST_VALU of
// ~485 cycles) and, with 4 waves per SIMD, advances every ~410 cycles (SQ_WAVE_CYCLES); LDS (11 broadcast ds_read_b128 per
// step, no conflicts) and HBM are far from their limits.  Measured and rejected:
//   * state pairs on v_pk_mul_f32 / v_pk_fma_f32 (104 -> 75 instructions per step): identical time -- a packed f32
//     instruction occupies the ALU for two passes, so only the lane-operation count matters;
//   * capping the scan's occupancy (dynamic-LDS pad, 2 instead of 4 waves per SIMD) so that GEMM workgroups of the other
//     experts' streams could co-reside: MambaIR alone 136 -> 176 ms and the whole step slower by the same 40 ms.
// What is left is the lane-operation count itself: 16 states x (mul, exp2, mul, fma, fma) per (step, channel, direction),
// twice (pass A and pass C).
#include "ffsr_common.h"

namespace {

constexpr int NS = 16;  // d_state

struct ScanArgs {
  const float* u;     // [B, L, ldu]
  const float* xdbl;  // [B, L, ldx]
  const float* dtw;   // [4, Dm, R]
  const float* dtb;   // [4, Dm]
  const float* A;     // [4, Dm, NS]
  const float* Dv;    // [4, Dm]
  float* y;           // [4, B, L, ldy]
  float* hstate;      // [B, 4, nchunk, Dm, NS]  local end state (pass A) -> incoming state (pass B)
  float* decay;       // [B, 4, nchunk, Dm, NS]
  int B, H, W, L, Dm, ldu, ldx, ldy, R, chunk, nchunk;
  int kbase, nk;      // this launch covers the directions kbase .. kbase + nk - 1 (scratch is indexed by the local direction)
  int pairs;          // 1: y has TWO planes -- direction k writes plane k & 1, the directions 2 and 3 ADD to what 0 and 1 wrote
};

__device__ __forceinline__ int tok_pixel(int k, int t, int H, int W, int L) {
  if (k & 2) t = L - 1 - t;
  if (k & 1) {
    int x = t / H, yy = t - x * H;
    return yy * W + x;
  }
  return t;
}

// softplus with torch's threshold 20, on the hardware exp2/log2: log1p(e) = e(1 - e(1/2 - e/3)) for small e = exp(x)
// (relative error < 4e-7), log(1 + e) otherwise.  Branch-free selects.
__device__ __forceinline__ float softplus_f(float x) {
  const float e = __builtin_amdgcn_exp2f(fminf(x, 20.f) * 1.4426950408889634f);
  const float small = e * (1.f - e * (0.5f - e * 0.33333334f));
  const float big = __builtin_amdgcn_logf(1.f + e) * 0.6931471805599453f;
  const float sp = e < 0.01f ? small : big;
  return x > 20.f ? x : sp;
}

// Steps are processed in groups of G: the group's u values (registers) and projection rows (LDS, wave-uniform reads)
// are fetched one group ahead, so the recurrence never waits on HBM latency.
template <int R, bool EMIT>
__global__ __launch_bounds__(64, 4) void scan_chunk_kernel(ScanArgs p) {
  constexpr int G = 8, XW = R + 2 * NS, XN = (G * XW + 63) / 64;
  __shared__ float xs[2][G * XW];
  const int lane = threadIdx.x;
  const int d = blockIdx.x * 64 + lane;
  const int c = blockIdx.y;
  const int kl = blockIdx.z % p.nk, b = blockIdx.z / p.nk;
  const int k = p.kbase + kl;
  const bool live = d < p.Dm;
  const int dd = live ? d : p.Dm - 1;
  float w[R], a2[NS], h[NS];
#pragma unroll
  for (int r = 0; r < R; ++r) w[r] = p.dtw[((size_t)k * p.Dm + dd) * R + r];
#pragma unroll
  for (int n = 0; n < NS; ++n) a2[n] = p.A[((size_t)k * p.Dm + dd) * NS + n] * 1.4426950408889634f;  // exp(x) = 2^(x log2 e)
  const float bias = p.dtb[k * p.Dm + dd];
  const float dskip = p.Dv[k * p.Dm + dd];
  const size_t sidx = ((((size_t)b * p.nk + kl) * p.nchunk + c) * p.Dm + dd) * NS;
#pragma unroll
  for (int n = 0; n < NS; ++n) h[n] = EMIT ? p.hstate[sidx + n] : 0.f;
  float dsum = 0.f;
  const int t0 = c * p.chunk, t1 = min(p.L, t0 + p.chunk);
  const float* ub = p.u + (size_t)b * p.L * p.ldu;
  const float* xb = p.xdbl + (size_t)b * p.L * p.ldx + k * XW;
  float* yb = p.y + ((size_t)(p.pairs ? (k & 1) : k) * p.B + b) * p.L * p.ldy;
  const bool accum = EMIT && p.pairs && k >= 2;      // (wave-uniform)

  float u_cur[G], u_nxt[G], x_nxt[XN];
  float yo_cur[G], yo_nxt[G];                        // accum: what the first pair left at the pixels of the group
  int pix_cur[G], pix_nxt[G];
  auto fetch = [&](int tg) {  // global -> registers for the group starting at step tg
    const int mypix = tok_pixel(k, min(tg + (lane & (G - 1)), p.L - 1), p.H, p.W, p.L);
#pragma unroll
    for (int r = 0; r < G; ++r) {
      pix_nxt[r] = __builtin_amdgcn_readlane(mypix, r);      // wave-uniform: lives in a scalar register
      u_nxt[r] = ub[(size_t)pix_nxt[r] * p.ldu + dd];
      if (EMIT) yo_nxt[r] = accum ? yb[(size_t)pix_nxt[r] * p.ldy + dd] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < XN; ++i) {
      const int idx = lane + 64 * i;
      const int row = idx / XW, col = idx - row * XW;
      const int rp = __shfl(mypix, row & (G - 1), 64);
      x_nxt[i] = idx < G * XW ? xb[(size_t)rp * p.ldx + col] : 0.f;
    }
  };
  auto commit = [&](int buf) {  // registers -> LDS / current
#pragma unroll
    for (int i = 0; i < XN; ++i)
      if (lane + 64 * i < G * XW) xs[buf][lane + 64 * i] = x_nxt[i];
#pragma unroll
    for (int r = 0; r < G; ++r) {
      u_cur[r] = u_nxt[r];
      pix_cur[r] = pix_nxt[r];
      if (EMIT) yo_cur[r] = yo_nxt[r];
    }
  };
  fetch(t0);
  commit(0);
  __syncthreads();
  int buf = 0;
  for (int tg = t0; tg < t1; tg += G) {
    const bool more = tg + G < t1;
    if (more) fetch(tg + G);
    const float* xg = xs[buf];
#pragma unroll
    for (int r = 0; r < G; ++r) {
      if (tg + r < t1) {
        const float* xr = xg + r * XW;  // wave-uniform LDS reads (broadcast)
        const float uu = u_cur[r];
        float dt = bias;
#pragma unroll
        for (int q = 0; q < R; ++q) dt = fmaf(w[q], xr[q], dt);
        const float delta = softplus_f(dt);
        const float du = delta * uu;
        float yv = 0.f;
#pragma unroll
        for (int n = 0; n < NS; ++n) {
          const float dA = __builtin_amdgcn_exp2f(delta * a2[n]);
          h[n] = fmaf(dA, h[n], du * xr[R + n]);
          if (EMIT) yv = fmaf(xr[R + NS + n], h[n], yv);
        }
        if (EMIT) {
          if (live) yb[(size_t)pix_cur[r] * p.ldy + d] = fmaf(dskip, uu, yv) + yo_cur[r];
        } else {
          dsum += delta;
        }
      }
    }
    if (more) commit(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  if (!EMIT && live) {
#pragma unroll
    for (int n = 0; n < NS; ++n) {
      p.hstate[sidx + n] = h[n];
      p.decay[sidx + n] = __builtin_amdgcn_exp2f(a2[n] * dsum);
    }
  }
}

// hstate[c] <- state entering chunk c.  The carry over chunks is the affine recurrence h' = decay * h + local, one
// independent chain per (batch, direction, channel, state) = only B*4*Dm*16 chains: a thread per chain walks ~512
// chunks with dependent loads on a quarter of the chip.  Here a 1024-thread block covers 64 chains x SEG = 16 segments
// of the chunk range: every thread composes the affine map of its segment (h -> P h + Q), the 16 maps are combined
// through LDS, and the segment is walked a second time from its true incoming state.  Lanes run along the chains, so
// every access is a contiguous 256-byte row segment.
constexpr int CARRY_SEG = 16;
__global__ __launch_bounds__(64 * CARRY_SEG) void scan_carry_kernel(float* __restrict__ hstate, const float* __restrict__ decay,
                                                                    int nchunk, int per) {
  __shared__ float Ps[CARRY_SEG][64], Qs[CARRY_SEG][64];
  const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane, g = blockIdx.y;
  const bool live = i < per;
  const int ii = live ? i : per - 1;
  const int cps = (nchunk + CARRY_SEG - 1) / CARRY_SEG;
  const int c0 = seg * cps, c1 = min(nchunk, c0 + cps);
  const size_t base = (size_t)g * nchunk * per + ii;
  float P = 1.f, Q = 0.f;
  constexpr int U = 8;  // loads of U chunks are issued together (they do not depend on the recurrence)
  for (int cc = c0; cc < c1; cc += U) {
    float hl[U], dc[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const size_t o = base + (size_t)min(cc + j, nchunk - 1) * per;
      hl[j] = hstate[o];
      dc[j] = decay[o];
    }
#pragma unroll
    for (int j = 0; j < U; ++j)
      if (cc + j < c1) {
        P *= dc[j];
        Q = fmaf(dc[j], Q, hl[j]);
      }
  }
  Ps[seg][lane] = P;
  Qs[seg][lane] = Q;
  __syncthreads();
  float hin = 0.f;
  for (int sgm = 0; sgm < seg; ++sgm) hin = fmaf(Ps[sgm][lane], hin, Qs[sgm][lane]);
  for (int cc = c0; cc < c1; cc += U) {
    float hl[U], dc[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const size_t o = base + (size_t)min(cc + j, nchunk - 1) * per;
      hl[j] = hstate[o];
      dc[j] = decay[o];
    }
#pragma unroll
    for (int j = 0; j < U; ++j)
      if (cc + j < c1) {
        if (live) hstate[base + (size_t)(cc + j) * per] = hin;
        hin = fmaf(dc[j], hin, hl[j]);
      }
  }
}


// y = LayerNorm_Dm(y0 + y1 + y2 + y3) * silu(z)        (SS2D.forward, mambair_arch.py:380-384)
__global__ __launch_bounds__(256) void mamba_norm_gate_kernel(const float* __restrict__ y, size_t ystride, int ldy,
                                                              const float* __restrict__ z, int ldz,
                                                              const float* __restrict__ g, const float* __restrict__ be,
                                                              float eps, float* __restrict__ out, int ldo, int M, int C) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  float v[8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int c = lane + 64 * i;
    float t = 0.f;
    if (c < C) {
      const float* p0 = y + (size_t)row * ldy + c;
      t = ((p0[0] + p0[2 * ystride]) + p0[ystride]) + p0[3 * ystride];  // y1+y2+y3+y4 order of mambair_arch.py:381
    }
    v[i] = t;
    s += t;
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float dlt = (lane + 64 * i < C) ? v[i] - mean : 0.f;
    q += dlt * dlt;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int c = lane + 64 * i;
    if (c < C) {
      float zz = z[(size_t)row * ldz + c];
      float yn = (v[i] - mean) * rstd * g[c] + be[c];
      out[(size_t)row * ldo + c] = yn * (zz / (1.0f + expf(-zz)));
    }
  }
}

// Vectorised form (C % 4 == 0, 16-byte aligned rows): half a wave per row, 8 channels per lane and pass; fp32 output
// and / or the bf16 hi / lo planes of the result ([M, ldp], pad columns zero) for ffsr_conv2d_planes (out_proj).
template <int NPASS>
__global__ __launch_bounds__(256) void mamba_norm_gate_v8_kernel(const float* __restrict__ y, size_t ystride, int ldy,
                                                                 const float* __restrict__ z, int ldz,
                                                                 const float* __restrict__ g, const float* __restrict__ be,
                                                                 float eps, float* __restrict__ out, int ldo,
                                                                 unsigned short* __restrict__ ohi,
                                                                 unsigned short* __restrict__ olo, int ldp, int M, int C,
                                                                 int ndir) {
  const int row = blockIdx.x * 8 + (threadIdx.x >> 5);
  const int l = threadIdx.x & 31;
  if (row >= M) return;
  floatx4 v[NPASS][2];
  float s = 0.f;
#pragma unroll
  for (int p = 0; p < NPASS; ++p)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int c = (p * 32 + l) * 8 + 4 * hh;
      floatx4 t = {0.f, 0.f, 0.f, 0.f};
      if (c < C) {
        const float* p0 = y + (size_t)row * ldy + c;
        // y1+y2+y3+y4 in the order of mambair_arch.py:381
        if (ndir == 4)
          t = ((*reinterpret_cast<const floatx4*>(p0) + *reinterpret_cast<const floatx4*>(p0 + 2 * ystride)) +
               *reinterpret_cast<const floatx4*>(p0 + ystride)) + *reinterpret_cast<const floatx4*>(p0 + 3 * ystride);
        else        // the pair planes of ffsr_selective_scan4_pairs_f32: (y0 + y2) + (y1 + y3)
          t = *reinterpret_cast<const floatx4*>(p0) + *reinterpret_cast<const floatx4*>(p0 + ystride);
      }
      v[p][hh] = t;
      s += (t[0] + t[1]) + (t[2] + t[3]);
    }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int p = 0; p < NPASS; ++p)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int c = (p * 32 + l) * 8 + 4 * hh;
      if (c < C) {
        const floatx4 d = v[p][hh] - mean;
        q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
      }
    }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = 1.0f / sqrtf(q / (float)C + eps);
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
    const int c0 = (p * 32 + l) * 8;
    float r[8];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int c = c0 + 4 * hh;
      floatx4 t = {0.f, 0.f, 0.f, 0.f};
      if (c < C) {
        const floatx4 zz = *reinterpret_cast<const floatx4*>(z + (size_t)row * ldz + c);
        const floatx4 yn = (v[p][hh] - mean) * rstd * *reinterpret_cast<const floatx4*>(g + c) + *reinterpret_cast<const floatx4*>(be + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = yn[e] * (zz[e] / (1.0f + expf(-zz[e])));
        if (out) *reinterpret_cast<floatx4*>(out + (size_t)row * ldo + c) = t;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) r[4 * hh + e] = t[e];
    }
    if (ohi && c0 < ldp) ffsr_store_planes8(ohi + (size_t)row * ldp + c0, olo + (size_t)row * ldp + c0, r);
  }
}

template <int R>
int run_scan(const ScanArgs& a, hipStream_t st) {
  dim3 grid((a.Dm + 63) / 64, a.nchunk, a.nk * a.B);
  FFSR_LAUNCH((scan_chunk_kernel<R, false>), grid, dim3(64), 0, st, a);
  FFSR_LAUNCH(scan_carry_kernel, dim3((a.Dm * NS + 63) / 64, a.B * a.nk), dim3(64 * CARRY_SEG), 0, st, a.hstate, a.decay,
                     a.nchunk, a.Dm * NS);
  FFSR_LAUNCH((scan_chunk_kernel<R, true>), grid, dim3(64), 0, st, a);
  return ffsr_launch_status();
}

}  // namespace

namespace {
int scan_common(const float* u, int ldu, const float* xdbl, int ldx, const float* dtw, const float* dtb, const float* A,
                const float* Dv, float* y, int ldy, float* hstate, float* decay, int B, int H, int W, int Dm, int R, int d_state,
                int chunk, int pairs, void* stream) {
  FFSR_CHECK(u && xdbl && dtw && dtb && A && Dv && y && hstate && decay);
  FFSR_CHECK(B > 0 && H > 0 && W > 0 && Dm > 0 && d_state == NS && chunk > 0);
  FFSR_CHECK(ldu >= Dm && ldy >= Dm && ldx >= 4 * (R + 2 * NS));
  ScanArgs a;
  a.u = u; a.xdbl = xdbl; a.dtw = dtw; a.dtb = dtb; a.A = A; a.Dv = Dv; a.y = y; a.hstate = hstate; a.decay = decay;
  a.B = B; a.H = H; a.W = W; a.L = H * W; a.Dm = Dm; a.ldu = ldu; a.ldx = ldx; a.ldy = ldy; a.R = R; a.chunk = chunk;
  a.nchunk = (a.L + chunk - 1) / chunk;
  FFSR_CHECK(a.nchunk <= 65535 && 4 * B <= 65535);
  hipStream_t st = (hipStream_t)stream;
  a.pairs = pairs;
  for (int kb = 0; kb < 4; kb += pairs ? 2 : 4) {      // pairs: the directions {0, 1}, then {2, 3} on top of them (stream order)
    a.kbase = kb;
    a.nk = pairs ? 2 : 4;
    int rc;
    switch (R) {
      case 12: rc = run_scan<12>(a, st); break;
      case 3: rc = run_scan<3>(a, st); break;
      default: return FFSR_EINVAL;
    }
    if (rc != FFSR_OK) return rc;
  }
  return FFSR_OK;
}
}  // namespace

extern "C" int ffsr_selective_scan4_f32(const float* u, int ldu, const float* xdbl, int ldx, const float* dtw,
                                        const float* dtb, const float* A, const float* Dv, float* y, int ldy,
                                        float* hstate, float* decay, int B, int H, int W, int Dm, int R, int d_state,
                                        int chunk, void* stream) {
  return scan_common(u, ldu, xdbl, ldx, dtw, dtb, A, Dv, y, ldy, hstate, decay, B, H, W, Dm, R, d_state, chunk, 0, stream);
}

// See include/ffsr.h.
extern "C" int ffsr_selective_scan4_pairs_f32(const float* u, int ldu, const float* xdbl, int ldx, const float* dtw,
                                              const float* dtb, const float* A, const float* Dv, float* y, int ldy,
                                              float* hstate, float* decay, int B, int H, int W, int Dm, int R, int d_state,
                                              int chunk, void* stream) {
  return scan_common(u, ldu, xdbl, ldx, dtw, dtb, A, Dv, y, ldy, hstate, decay, B, H, W, Dm, R, d_state, chunk, 1, stream);
}

namespace {
int norm_gate_common(const float* y, long long ystride, int ndir, int ldy, const float* z, int ldz, const float* gamma,
                     const float* beta, float eps, float* out, int ldo, void* out_hi, void* out_lo, int ldp, int M, int C,
                     void* stream) {
  FFSR_CHECK(y && z && gamma && beta && (out || (out_hi && out_lo)) && M > 0 && C > 0 && C <= 512 && (ndir == 4 || ndir == 2));
  auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  FFSR_CHECK(!out_hi || (out_lo && (ldp & 31) == 0 && ldp >= C && ldp < C + 32 && al(out_hi) && al(out_lo)));
  const bool v8 = (C % 4 == 0) && (ldy % 4 == 0) && (ystride % 4 == 0) && (ldz % 4 == 0) && al(y) && al(z) && al(gamma) &&
                  al(beta) && (!out || (ldo % 4 == 0 && al(out)));
  hipStream_t st = (hipStream_t)stream;
  if (!v8) {
    FFSR_CHECK(out && !out_hi && ndir == 4);
    FFSR_LAUNCH(mamba_norm_gate_kernel, dim3((M + 3) / 4), dim3(256), 0, st, y, (size_t)ystride, ldy, z, ldz, gamma,
                       beta, eps, out, ldo, M, C);
    return ffsr_launch_status();
  }
  unsigned short* oh = (unsigned short*)out_hi;
  unsigned short* ol = (unsigned short*)out_lo;
  if (C <= 256)
    FFSR_LAUNCH(mamba_norm_gate_v8_kernel<1>, dim3((M + 7) / 8), dim3(256), 0, st, y, (size_t)ystride, ldy, z, ldz,
                       gamma, beta, eps, out, ldo, oh, ol, ldp, M, C, ndir);
  else
    FFSR_LAUNCH(mamba_norm_gate_v8_kernel<2>, dim3((M + 7) / 8), dim3(256), 0, st, y, (size_t)ystride, ldy, z, ldz,
                       gamma, beta, eps, out, ldo, oh, ol, ldp, M, C, ndir);
  return ffsr_launch_status();
}
}  // namespace

extern "C" int ffsr_mamba_norm_gate_planes_f32(const float* y, long long ystride, int ldy, const float* z, int ldz,
                                               const float* gamma, const float* beta, float eps, float* out, int ldo,
                                               void* out_hi, void* out_lo, int ldp, int M, int C, void* stream) {
  return norm_gate_common(y, ystride, 4, ldy, z, ldz, gamma, beta, eps, out, ldo, out_hi, out_lo, ldp, M, C, stream);
}

// See include/ffsr.h: the same over the TWO pair planes of ffsr_selective_scan4_pairs_f32.
extern "C" int ffsr_mamba_norm_gate_pairs_f32(const float* y, long long ystride, int ldy, const float* z, int ldz,
                                              const float* gamma, const float* beta, float eps, float* out, int ldo,
                                              void* out_hi, void* out_lo, int ldp, int M, int C, void* stream) {
  return norm_gate_common(y, ystride, 2, ldy, z, ldz, gamma, beta, eps, out, ldo, out_hi, out_lo, ldp, M, C, stream);
}

extern "C" int ffsr_mamba_norm_gate_f32(const float* y, long long ystride, int ldy, const float* z, int ldz,
                                        const float* gamma, const float* beta, float eps, float* out, int ldo, int M, int C,
                                        void* stream) {
  FFSR_CHECK(out);
  return ffsr_mamba_norm_gate_planes_f32(y, ystride, ldy, z, ldz, gamma, beta, eps, out, ldo, nullptr, nullptr, 0, M, C, stream);
}
