/* CPU restatement of the selective-scan recurrence in plain C (TEST INFRASTRUCTURE: oracle / cpu_baseline only).
 *
 * PARITY UNPINNED -- same status and same definition as oracle/ffsr_oracle/scan.py: the reference calls the
 * un-vendored CUDA wheel mamba-ssm 2.3.0 (mambair_arch.py:11,356-362; scripts/kaggle_inference_fixed.py:19);
 * this follows the published recurrence
 *     delta = softplus(dt + bias) (threshold 20);  h = exp(delta*A) h + delta*B_t*u_t;  y = <C_t, h> + D u_t
 * with u, dt, y: [Bsz, Dm, L]; A: [Dm, N]; Bm, Cm: [Bsz, G, N, L]; D, bias: [Dm]; fp32, h_0 = 0.
 * Built by __graft_entry__.build() / oracle/ffsr_oracle/scan_c.py:  gcc -O2 -fopenmp -shared -fPIC.
 */
#include <math.h>
#include <stdlib.h>

int ffsr_oracle_selective_scan(const float* u, const float* dt, const float* A, const float* Bm, const float* Cm,
                               const float* D, const float* bias, float* y, int Bsz, int Dm, int L, int N, int G,
                               int softplus) {
  if (N > 64 || Dm % G != 0) return -1;
  const int rep = Dm / G;
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < Bsz; ++b) {
    for (int d = 0; d < Dm; ++d) {
      float h[64];
      for (int n = 0; n < N; ++n) h[n] = 0.f;
      const float* ud = u + ((size_t)b * Dm + d) * L;
      const float* dd = dt + ((size_t)b * Dm + d) * L;
      const float* Bg = Bm + ((size_t)b * G + d / rep) * N * L;
      const float* Cg = Cm + ((size_t)b * G + d / rep) * N * L;
      float* yd = y + ((size_t)b * Dm + d) * L;
      const float* a = A + (size_t)d * N;
      const float bi = bias ? bias[d] : 0.f, dsk = D ? D[d] : 0.f;
      for (int t = 0; t < L; ++t) {
        float x = dd[t] + bi;
        float delta = softplus ? (x > 20.f ? x : log1pf(expf(x))) : x;
        float du = delta * ud[t], acc = 0.f;
        for (int n = 0; n < N; ++n) {
          h[n] = expf(delta * a[n]) * h[n] + du * Bg[(size_t)n * L + t];
          acc += h[n] * Cg[(size_t)n * L + t];
        }
        yd[t] = acc + dsk * ud[t];
      }
    }
  }
  return 0;
}
