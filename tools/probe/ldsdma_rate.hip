// Micro-benchmark: sustained global -> LDS fill rate of LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction) per CU,
// as a function of the waves per workgroup and workgroups per CU, from an L2-resident source (what the planes GEMM's loader
// sees).  Also the same bytes through global_load_dwordx4 + ds_write_b128 for comparison.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/ldsdma_rate tools/probe/ldsdma_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

// each wave: `iters` rounds of `pieces` LDS-DMA instructions (16 B per lane) from its own 16 KiB window of src (L2 hits)
template <int PIECES, bool DMA>
__global__ void fill_kernel(const unsigned char* __restrict__ src, size_t window, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned char* dst = smem + wave * (PIECES * 1024);
  // every wave walks the same `window` bytes from its own starting offset: 16 KiB windows stay in the CU's vector L1,
  // 2 MiB is L2-resident but misses L1, 512 MiB streams from HBM
  const size_t base = ((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * 20480 % window;
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned char* s = src + (base + (size_t)it * (PIECES * 1024)) % window;
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      if (DMA) {
        __builtin_amdgcn_global_load_lds(s + p * 1024 + lane * 16, (lds_ptr_t)(dst + p * 1024), 16, 0, 0);
      } else {
        const uintx4 v = *reinterpret_cast<const uintx4*>(s + p * 1024 + lane * 16);
        *reinterpret_cast<uintx4*>(dst + p * 1024 + lane * 16) = v;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc += *reinterpret_cast<unsigned*>(dst + lane * 4);
    __syncthreads();
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int PIECES, bool DMA>
double run(const unsigned char* src, size_t window, unsigned* sink, int waves, int wg_per_cu, int iters) {
  const int lds = waves * PIECES * 1024;
  const int pad = 160 * 1024 / wg_per_cu;          // dynamic LDS so that exactly wg_per_cu workgroups fit a CU
  const int shmem = pad > lds ? (pad < 160 * 1024 ? pad - 1024 : pad) : lds;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&fill_kernel<PIECES, DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int grid = 256 * wg_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((fill_kernel<PIECES, DMA>), dim3(grid), dim3(waves * 64), shmem, 0, src, window, iters / 10, sink);
  hipEventRecord(e0);
  hipLaunchKernelGGL((fill_kernel<PIECES, DMA>), dim3(grid), dim3(waves * 64), shmem, 0, src, window, iters, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double total = (double)grid * waves * PIECES * 1024.0 * iters;
  return total / (ms * 1e-3) / 256.0;              // bytes per second per CU
}

int main() {
  const size_t bytes = (size_t)512 << 20;
  unsigned char* src; unsigned* sink;
  hipMalloc(&src, bytes + 65536); hipMalloc(&sink, 4);
  hipMemset(src, 1, bytes + 65536);
  for (size_t window : {(size_t)16 << 10, (size_t)2 << 20, bytes}) {
    printf("source window %zu KiB\nwaves/WG WG/CU   lds-dma GB/s per CU (B/clk at 2.0 GHz)   global_load + ds_write\n", window >> 10);
    for (int wg_per_cu : {1, 2, 4})
      for (int waves : {1, 2, 4, 8}) {
        if (waves * wg_per_cu > 16) continue;
        const double a = run<4, true>(src, window, sink, waves, wg_per_cu, 2000);
        const double b = run<4, false>(src, window, sink, waves, wg_per_cu, 2000);
        printf("%8d %5d  %10.1f  (%5.1f)   %10.1f  (%5.1f)\n", waves, wg_per_cu, a / 1e9, a / 2.0e9, b / 1e9, b / 2.0e9);
        fflush(stdout);
      }
  }
  return 0;
}
