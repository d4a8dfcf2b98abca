// Optimiser side of the cached-feature training step (train.py:251-384 of the reference), gfx950.
//
//   sr = clamp(model(...), 0, 1);  loss = mean |sr - hr| / accumulation_steps      (train.py:326-333, perceptual_loss.py:68-100)
//   clip_grad_norm_(params, max_norm)                                               (train.py:347-352)
//   AdamW step, EMA update                                                          (train.py:354-359, checkpoint_manager.py:349-356)
//
// Everything is HBM-bound elementwise / reduction work over a flat fp32 buffer (the fusion net has ~1.4 M parameters; the
// loss runs over B x 3 x 256 x 256 pixels).  Reductions are two-stage and deterministic (fixed partial layout, no
// atomics); the clip coefficient is derived on the device from the reduced squared norm, so the step needs no host sync.
// The backward kernels of the fusion phases live in ffsr_backward.hip / ffsr_wgrad.hip; this file sees the flat gradient
// buffer they filled.
#include "ffsr_common.h"

namespace {

constexpr int RED_BLOCK = 256;

__device__ __forceinline__ float block_sum(float v, float* red) {   // red: >= 4 floats of LDS
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// part[block] = sum over the block's grid-stride elements of |clamp(sr) - hr|; grad = scale * sign(clamp(sr) - hr) where
// the clamp passes the gradient (0 <= sr <= 1, the closed interval torch.clamp differentiates through), else 0.
__global__ __launch_bounds__(RED_BLOCK) void l1_clamp_kernel(const float* __restrict__ sr, int ldsr, const float* __restrict__ hr,
                                                            int ldhr, float* __restrict__ grad, int ldg, float* __restrict__ part,
                                                            long long M, int C, float gscale) {
  __shared__ float red[4];
  const long long total = M * C;
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * RED_BLOCK + threadIdx.x; i < total; i += (long long)gridDim.x * RED_BLOCK) {
    const long long m = i / C;
    const int c = (int)(i - m * C);
    const float x = sr[m * ldsr + c];
    const float xc = fminf(fmaxf(x, 0.f), 1.f);
    const float d = xc - hr[m * ldhr + c];
    acc += fabsf(d);
    if (grad) {
      const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      grad[m * ldg + c] = (x >= 0.f && x <= 1.f) ? sg * gscale : 0.f;
    }
  }
  const float s = block_sum(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(RED_BLOCK) void sumsq_kernel(const float* __restrict__ x, long long n, float* __restrict__ part) {
  __shared__ float red[4];
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * RED_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * RED_BLOCK) acc = fmaf(x[i], x[i], acc);
  const float s = block_sum(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// out[0] = scale * sum(part[0..n))   (one block; fixed summation tree)
__global__ __launch_bounds__(RED_BLOCK) void finish_sum_kernel(const float* __restrict__ part, int n, float* __restrict__ out, float scale) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += RED_BLOCK) acc += part[i];
  const float s = block_sum(acc, red);
  if (threadIdx.x == 0) out[0] = s * scale;
}

// torch.optim.AdamW (decoupled weight decay, no amsgrad) on a flat buffer, preceded by clip_grad_norm_'s scaling and
// followed by EMAModel.update:
//   coef = min(1, max_norm / (sqrt(sumsq) + 1e-6))  (max_norm <= 0: no clipping);  g = grad * coef
//   p *= 1 - lr * wd;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps);   ema = d * ema + (1 - d) * p
__global__ __launch_bounds__(256) void adamw_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, float* __restrict__ ema, long long n,
                                                       const float* __restrict__ sumsq, float max_norm, float lr, float b1,
                                                       float b2, float eps, float wd, float bc1, float bc2_sqrt, float decay) {
  float coef = 1.f;
  if (sumsq && max_norm > 0.f) coef = fminf(1.f, max_norm / (sqrtf(sumsq[0]) + 1e-6f));
  const float step_size = lr / bc1;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float gi = g[i] * coef;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    pi -= step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
    if (ema) ema[i] = decay * ema[i] + (1.f - decay) * pi;
  }
}

int red_blocks(long long n) {
  long long b = (n + RED_BLOCK * 8 - 1) / (RED_BLOCK * 8);
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

}  // namespace

#define ST ((hipStream_t)stream)

// See include/ffsr.h for the contracts.
extern "C" int ffsr_l1_clamp_loss_f32(const float* sr, int ldsr, const float* hr, int ldhr, float* grad, int ldg, float* partial,
                                      int n_partial, float* loss, long long M, int C, float loss_scale, void* stream) {
  FFSR_CHECK(sr && hr && partial && loss && M > 0 && C > 0 && ldsr >= C && ldhr >= C && (!grad || ldg >= C));
  const int blocks = red_blocks(M * C);
  FFSR_CHECK(n_partial >= blocks);
  const float inv = loss_scale / (float)((double)M * C);
  FFSR_LAUNCH(l1_clamp_kernel, dim3(blocks), dim3(RED_BLOCK), 0, ST, sr, ldsr, hr, ldhr, grad, ldg, partial, M, C, inv);
  FFSR_LAUNCH(finish_sum_kernel, dim3(1), dim3(RED_BLOCK), 0, ST, partial, blocks, loss, inv);
  return ffsr_launch_status();
}

extern "C" int ffsr_sumsq_f32(const float* x, long long n, float* partial, int n_partial, float* out, void* stream) {
  FFSR_CHECK(x && partial && out && n > 0);
  const int blocks = red_blocks(n);
  FFSR_CHECK(n_partial >= blocks);
  FFSR_LAUNCH(sumsq_kernel, dim3(blocks), dim3(RED_BLOCK), 0, ST, x, n, partial);
  FFSR_LAUNCH(finish_sum_kernel, dim3(1), dim3(RED_BLOCK), 0, ST, partial, blocks, out, 1.0f);
  return ffsr_launch_status();
}

extern "C" int ffsr_adamw_ema_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema, long long n,
                                  const float* grad_sumsq, float max_norm, float lr, float beta1, float beta2, float eps,
                                  float weight_decay, int step, float ema_decay, void* stream) {
  FFSR_CHECK(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1);
  FFSR_CHECK(lr >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps > 0.f && weight_decay >= 0.f);
  FFSR_CHECK(!ema || (ema_decay >= 0.f && ema_decay <= 1.f));
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  long long b = (n + 256 * 4 - 1) / (256 * 4);
  const int blocks = (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
  FFSR_LAUNCH(adamw_ema_kernel, dim3(blocks), dim3(256), 0, ST, param, grad, exp_avg, exp_avg_sq, ema, n, grad_sumsq, max_norm,
              lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), ema_decay);
  return ffsr_launch_status();
}
