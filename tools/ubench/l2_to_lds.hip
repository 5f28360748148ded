// Microbenchmark: how fast can one CU pull L2-resident data into LDS?
//   mode 0: global_load_lds_dwordx4 (LDS-DMA), 8 rows x 128 B per wave-instruction
//   mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128 (register staged)
//   mode 2: global_load_dwordx4 -> VGPR only (no LDS), the L2 -> CU path itself
//   mode 3: LDS-DMA in the conv kernels' shape: 8 rows x 128 B per instruction, rows `stride` bytes apart,
//           16-byte chunks XOR-swizzled with the row (the A-operand gather of conv_pp.h)
// Every workgroup (WAVES waves) re-reads its own `span` bytes (L2-resident after the first pass) `iters` times,
// `depth` wave-instructions in flight per wave.  Prints aggregate TB/s and B/clk/CU (at the measured wall time and
// an assumed 2.1 GHz).   Build: hipcc --offload-arch=gfx950 -O3 -o l2_to_lds l2_to_lds.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, long span, int iters, float* sink, int stride = 1024) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const char* base = src + (long)blockIdx.x * (MODE == 3 ? span / 128 * stride : span);
  const long per_pass = (long)nw * DEPTH * 1024;          // bytes per workgroup per inner step
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int it = 0; it < iters; ++it) {
    for (long off = 0; off + per_pass <= span; off += per_pass) {
      uint4 v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const char* p = base + off + ((long)(d * nw + wave)) * 1024 + lane * 16;
        if (MODE == 3) {
          const long row = (off >> 7) + (long)(d * nw + wave) * 8 + (lane >> 3);       // 128 useful bytes per row
          p = base + row * stride + (((lane & 7) ^ ((lane >> 3) & 7)) << 4);
        }
        if (MODE == 0 || MODE == 3) {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                           (__attribute__((address_space(3))) void*)(lds + (wave * DEPTH + d) * 1024), 16, 0, 0);
        } else {
          v[d] = *(const uint4*)p;
        }
      }
      if (MODE == 0 || MODE == 3) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else if (MODE == 1) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) *(uint4*)(lds + (wave * DEPTH + d) * 1024 + lane * 16) = v[d];
      } else {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) { acc.x ^= v[d].x; acc.y ^= v[d].y; acc.z ^= v[d].z; acc.w ^= v[d].w; }
      }
    }
  }
  if (MODE != 2) {
    __syncthreads();
    acc = *(uint4*)(lds + threadIdx.x * 16);
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1.f;
}

template <int MODE, int DEPTH>
static void run(const char* name, const char* d_src, long span, int iters, int blocks, int waves, float* sink, int stride = 1024) {
  const size_t smem = (size_t)waves * DEPTH * 1024;
  hipFuncSetAttribute((const void*)k<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(blocks), dim3(waves * 64), smem, 0, d_src, span, 2, sink, stride);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(blocks), dim3(waves * 64), smem, 0, d_src, span, iters, sink, stride);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const long per_pass = (long)waves * DEPTH * 1024;
  const double bytes = (double)blocks * iters * (span / per_pass) * per_pass;
  printf("%-34s depth %2d waves %2d blocks %4d: %8.3f ms  %7.2f TB/s  %6.1f B/clk/CU (256 CUs @2.1 GHz)\n", name, DEPTH, waves, blocks, ms,
         bytes / ms / 1e9, bytes / (ms * 1e-3) / 256.0 / 2.1e9);
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 256;
  const long span = (argc > 2 ? atol(argv[2]) : 64) * 1024L;      // KB per workgroup (L2 resident: 256 x 64 KB = 16 MB over 8 XCDs)
  const int iters = argc > 3 ? atoi(argv[3]) : 400;
  char* d_src; float* sink;
  hipMalloc(&d_src, (size_t)blocks * span * 8 + (1 << 20));
  hipMemset(d_src, 1, (size_t)blocks * span * 8 + (1 << 20));
  hipMalloc(&sink, 4);
  printf("blocks %d, %ld KB per workgroup, %d passes\n", blocks, span / 1024, iters);
  run<0, 2>("LDS-DMA (global_load_lds x4)", d_src, span, iters, blocks, 8, sink);
  run<0, 4>("LDS-DMA (global_load_lds x4)", d_src, span, iters, blocks, 8, sink);
  run<0, 8>("LDS-DMA (global_load_lds x4)", d_src, span, iters, blocks, 8, sink);
  run<1, 4>("global_load x4 -> ds_write_b128", d_src, span, iters, blocks, 8, sink);
  run<1, 8>("global_load x4 -> ds_write_b128", d_src, span, iters, blocks, 8, sink);
  run<2, 4>("global_load x4 -> VGPR", d_src, span, iters, blocks, 8, sink);
  run<2, 8>("global_load x4 -> VGPR", d_src, span, iters, blocks, 8, sink);
  // gather shape: the workgroup's rows are `stride` bytes apart, so it spans span/128*stride bytes of the buffer
  run<3, 4>("LDS-DMA gather, stride 128 B", d_src, span, iters, blocks, 8, sink, 128);
  run<3, 4>("LDS-DMA gather, stride 384 B", d_src, span, iters, blocks, 8, sink, 384);
  run<3, 4>("LDS-DMA gather, stride 960 B", d_src, span, iters, blocks, 8, sink, 960);
  run<3, 4>("LDS-DMA gather, stride 1024 B", d_src, span, iters, blocks, 8, sink, 1024);
  return 0;
}
