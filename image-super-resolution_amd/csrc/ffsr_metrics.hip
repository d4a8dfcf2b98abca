// uint8 PSNR / SSIM of the evaluation script (second half): utils/utils_image.py:148-189 (cal_psnr_ssim).
// Integer work end to end: the Y plane is OpenCV's 8-bit fixed-point RGB -> YCrCb luma, the squared-error sum and the
// 7x7 window moments are exact integers; only the SSIM quotient and the final means are double precision.
#include "ffsr_common.h"

namespace {

// cv2.cvtColor(img, cv2.COLOR_RGB2YCrCb)[:, :, 0] for uint8: Y = (R*4899 + G*9617 + B*1868 + 2^13) >> 14
// (OpenCV's fixed-point BT.601 luma, yuv_shift = 14: R2Y 0.299, G2Y 0.587, B2Y 0.114 scaled by 2^14).
__device__ __forceinline__ unsigned char luma_cv(unsigned r, unsigned g, unsigned b) {
  return (unsigned char)((r * 4899u + g * 9617u + b * 1868u + 8192u) >> 14);
}

// img [H, W, 3] uint8 -> planes [P, Hc, Wc] uint8 of the window [crop, H - crop) x [crop, W - crop): P = 1 (Y) or 3 (R, G, B)
__global__ void u8_planes_kernel(const unsigned char* __restrict__ img, int W, int crop, int y_channel,
                                 unsigned char* __restrict__ planes, int Hc, int Wc) {
  const long long n = (long long)Hc * Wc;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int y = (int)(i / Wc), x = (int)(i % Wc);
    const unsigned char* p = img + ((long long)(y + crop) * W + (x + crop)) * 3;
    if (y_channel) {
      planes[i] = luma_cv(p[0], p[1], p[2]);
    } else {
      planes[i] = p[0];
      planes[n + i] = p[1];
      planes[2 * n + i] = p[2];
    }
  }
}

// partial[2 * block + 0] = sum (a - b)^2 over this block's pixels (exact in double: < 2^53),
// partial[2 * block + 1] = sum of the SSIM map over this block's interior pixels (window fully inside the plane)
__global__ void psnr_ssim_u8_kernel(const unsigned char* __restrict__ pa, const unsigned char* __restrict__ pb, int P,
                                    int Hc, int Wc, double* __restrict__ partial) {
  const long long n = (long long)P * Hc * Wc;
  unsigned long long sq = 0;
  double ss = 0.0;
  const double C1 = (0.01 * 255.0) * (0.01 * 255.0), C2 = (0.03 * 255.0) * (0.03 * 255.0);
  const double cov_norm = 49.0 / 48.0;          // skimage: use_sample_covariance=True, NP = win_size^2
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int d = (int)pa[i] - (int)pb[i];
    sq += (unsigned long long)(d * d);
    const long long r = i % ((long long)Hc * Wc);
    const int y = (int)(r / Wc), x = (int)(r % Wc);
    if (y >= 3 && y < Hc - 3 && x >= 3 && x < Wc - 3) {
      unsigned sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
      for (int dy = -3; dy <= 3; ++dy) {
        const unsigned char* qa = pa + i + (long long)dy * Wc - 3;
        const unsigned char* qb = pb + i + (long long)dy * Wc - 3;
#pragma unroll
        for (int dx = 0; dx < 7; ++dx) {
          const unsigned a = qa[dx], b = qb[dx];
          sx += a, sy += b, sxx += a * a, syy += b * b, sxy += a * b;
        }
      }
      const double ux = sx / 49.0, uy = sy / 49.0;
      const double vx = cov_norm * (sxx / 49.0 - ux * ux), vy = cov_norm * (syy / 49.0 - uy * uy);
      const double vxy = cov_norm * (sxy / 49.0 - ux * uy);
      ss += ((2.0 * ux * uy + C1) * (2.0 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2));
    }
  }
  __shared__ double sh_s[256];
  __shared__ unsigned long long sh_q[256];
  sh_s[threadIdx.x] = ss, sh_q[threadIdx.x] = sq;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh_s[threadIdx.x] += sh_s[threadIdx.x + o], sh_q[threadIdx.x] += sh_q[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[2 * blockIdx.x] = (double)sh_q[0], partial[2 * blockIdx.x + 1] = sh_s[0];
}

__global__ void psnr_ssim_final_kernel(const double* __restrict__ partial, int nb, double n_px, double n_int,
                                       double* __restrict__ out) {
  __shared__ double sh[2][256];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) a += partial[2 * i], b += partial[2 * i + 1];
  sh[0][threadIdx.x] = a, sh[1][threadIdx.x] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[0][threadIdx.x] += sh[0][threadIdx.x + o], sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0][0] / n_px, out[1] = n_int > 0 ? sh[1][0] / n_int : 0.0;
}

}  // namespace

extern "C" int ffsr_u8_planes(const unsigned char* img, int H, int W, int crop, int y_channel, unsigned char* planes,
                              void* stream) {
  FFSR_CHECK(img && planes && H > 0 && W > 0 && crop >= 0 && H - 2 * crop > 0 && W - 2 * crop > 0);
  const int Hc = H - 2 * crop, Wc = W - 2 * crop;
  const long long n = (long long)Hc * Wc;
  const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  FFSR_LAUNCH(u8_planes_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, img, W, crop, y_channel, planes, Hc, Wc);
  return ffsr_launch_status();
}

extern "C" int ffsr_psnr_ssim_u8(const unsigned char* pa, const unsigned char* pb, int P, int Hc, int Wc, double* partial,
                                 int n_partial, double* out, void* stream) {
  FFSR_CHECK(pa && pb && partial && out && P >= 1 && Hc >= 7 && Wc >= 7 && n_partial >= 1);
  const long long n = (long long)P * Hc * Wc;
  long long want = (n + 255) / 256;
  const int grid = (int)(want < n_partial ? want : n_partial);
  FFSR_LAUNCH(psnr_ssim_u8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, pa, pb, P, Hc, Wc, partial);
  FFSR_LAUNCH(psnr_ssim_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, grid, (double)n,
              (double)P * (Hc - 6) * (Wc - 6), out);
  return ffsr_launch_status();
}
