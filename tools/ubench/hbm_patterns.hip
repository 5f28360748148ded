// Microbenchmark: HBM -> LDS (LDS-DMA) bandwidth of the whole chip for the two ways a pointwise conv can walk its
// activation rows (row = `rowb` bytes = Cin * 2, tile = 256 rows per workgroup, 256 threads):
//   mode 0 "kstep": K chunk by K chunk -- each step fetches 128 B of every row of the tile (conv_dma / conv_pp), ring of 3 steps
//   mode 1 "rows" : the tile's rows in address order, the same number of bytes in flight
// No compute, nothing written: only the access pattern and the in-flight depth differ.
//   hipcc --offload-arch=gfx950 -O3 -o hbm_patterns hbm_patterns.hip && ./hbm_patterns [rows_millions] [rowb]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ void dma16(const char* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int MODE>
__global__ __launch_bounds__(256) void k(const char* __restrict__ src, long ntiles, int rowb, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int chunks = rowb / 128;                    // K steps per tile
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const char* base = src + tile * 256 * (long)rowb;
    if (MODE == 0) {
      for (int c = 0; c < chunks; ++c) {
        char* slot = lds + (c % 3) * 32768;
        // 256 rows x 128 B = 32 KB: 32 wave-instructions of 8 rows x 128 B, 8 per wave
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r0 = (wave * 8 + j) * 8;
          dma16(base + (long)(r0 + (lane >> 3)) * rowb + c * 128 + (lane & 7) * 16, slot + (wave * 8 + j) * 1024);
        }
        if (c >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");     // two steps stay in flight
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      const long bytes = 256L * rowb;
      int n = 0;
      for (long off = 0; off < bytes; off += 4096) {                     // 4 waves x 1 KB, address order
        dma16(base + off + wave * 1024 + lane * 16, lds + ((off >> 12) % 24) * 4096 + wave * 1024);
        if (++n > 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  if (*(volatile int*)(lds + threadIdx.x * 4) == 0x12345678) sink[0] = 1.f;
}

int main(int argc, char** argv) {
  const long rows = (long)((argc > 1 ? atof(argv[1]) : 4.0) * 1e6) / 256 * 256;
  const int rowb = argc > 2 ? atoi(argv[2]) : 512;
  char* d; float* sink;
  hipMalloc(&d, rows * rowb); hipMalloc(&sink, 4);
  hipMemset(d, 1, rows * rowb);
  const long ntiles = rows / 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode)
    for (int persist = 0; persist < 2; ++persist) {
      const int grid = persist ? 256 * 1 : (int)ntiles;
      auto fn = mode == 0 ? k<0> : k<1>;
      hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
      float best = 1e9;
      for (int it = 0; it < 5; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(fn, dim3(grid), dim3(256), 98304, 0, d, ntiles, rowb, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      printf("rowb %4d  %-5s %-10s grid %7d : %7.3f ms  %7.1f GB/s\n", rowb, mode ? "rows" : "kstep", persist ? "persistent" : "per-tile", grid, best,
             rows * (double)rowb / best / 1e6);
    }
  return 0;
}
