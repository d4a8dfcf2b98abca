/* CPU restatement of the selective-scan recurrence in plain C (TEST INFRASTRUCTURE: oracle / cpu_baseline only).
 *
 * PARITY UNPINNED -- same status and same definition as oracle/ffsr_oracle/scan.py: the reference calls the
 * un-vendored CUDA wheel mamba-ssm 2.3.0 (mambair_arch.py:11,356-362; scripts/kaggle_inference_fixed.py:19);
 * this follows the published recurrence
 *     delta = softplus(dt + bias) (threshold 20);  h = exp(delta*A) h + delta*B_t*u_t;  y = <C_t, h> + D u_t
 * with u, dt, y: [Bsz, Dm, L]; A: [Dm, N]; Bm, Cm: [Bsz, G, N, L]; D, bias: [Dm]; fp32, h_0 = 0.
 * Built by __graft_entry__.build() / oracle/ffsr_oracle/scan_c.py:  gcc -O3 -march=native -fopenmp -shared -fPIC ... -lmvec.
 */
#include <math.h>
#include <stdlib.h>

/* glibc ships vector variants of expf in libmvec (_ZGV?N?v_expf, < 4 ulp) but its headers only advertise them under
 * -ffast-math; this redeclaration advertises them to the `omp simd` loop below without relaxing any other arithmetic. */
#pragma omp declare simd notinbranch
extern float expf(float);

/* The 16 states of a step are independent: the inner loop is an `omp simd` loop, so gcc calls glibc's vector expf
 * (libmvec, < 4 ulp) instead of 16 scalar calls; B / C are transposed once per call to [L][N] so that the states of a
 * step are contiguous.  Same recurrence, same fp32 arithmetic per state; only the order of the 16-term sum <C_t, h>
 * differs (vector lanes).  ~6x faster than the scalar loop, which is what lets the oracle run a whole 340x510 image in
 * the GPU test suite. */
int ffsr_oracle_selective_scan(const float* u, const float* dt, const float* A, const float* Bm, const float* Cm,
                               const float* D, const float* bias, float* y, int Bsz, int Dm, int L, int N, int G,
                               int softplus) {
  if (N > 64 || Dm % G != 0) return -1;
  const int rep = Dm / G;
  float* Bt = (float*)malloc((size_t)Bsz * G * L * N * sizeof(float));
  float* Ct = (float*)malloc((size_t)Bsz * G * L * N * sizeof(float));
  if (!Bt || !Ct) {
    free(Bt), free(Ct);
    return -2;
  }
#pragma omp parallel for schedule(static)
  for (int bg = 0; bg < Bsz * G; ++bg)
    for (int n = 0; n < N; ++n)
      for (int t = 0; t < L; ++t) {
        Bt[((size_t)bg * L + t) * N + n] = Bm[((size_t)bg * N + n) * L + t];
        Ct[((size_t)bg * L + t) * N + n] = Cm[((size_t)bg * N + n) * L + t];
      }
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < Bsz; ++b) {
    for (int d = 0; d < Dm; ++d) {
      float h[64] __attribute__((aligned(64)));
      float a[64] __attribute__((aligned(64)));
      for (int n = 0; n < 64; ++n) h[n] = 0.f, a[n] = n < N ? A[(size_t)d * N + n] : 0.f;
      const float* ud = u + ((size_t)b * Dm + d) * L;
      const float* dd = dt + ((size_t)b * Dm + d) * L;
      const float* Bg = Bt + ((size_t)b * G + d / rep) * L * N;
      const float* Cg = Ct + ((size_t)b * G + d / rep) * L * N;
      float* yd = y + ((size_t)b * Dm + d) * L;
      const float bi = bias ? bias[d] : 0.f, dsk = D ? D[d] : 0.f;
      for (int t = 0; t < L; ++t) {
        float x = dd[t] + bi;
        float delta = softplus ? (x > 20.f ? x : log1pf(expf(x))) : x;
        float du = delta * ud[t], acc = 0.f;
        const float* Bs = Bg + (size_t)t * N;
        const float* Cs = Cg + (size_t)t * N;
#pragma omp simd reduction(+ : acc)
        for (int n = 0; n < N; ++n) {
          h[n] = expf(delta * a[n]) * h[n] + du * Bs[n];
          acc += h[n] * Cs[n];
        }
        yd[t] = acc + dsk * ud[t];
      }
    }
  }
  free(Bt), free(Ct);
  return 0;
}
