// ds_read_b64_tr_b16 fragment-read patterns of the streaming weight-gradient kernels (wgrad_rs.hip & co.): cycles per
// wave-instruction with 8 waves per CU (one 512-thread workgroup per CU, as those kernels run), for position-major tiles with
// 128-byte rows: lane l reads 8 bytes at row q*8 + h*4 + (p >> 2) (+ shift), columns col0 + (p & 3) * 4, p = l & 15, q = l >> 4.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_tr_patterns.hip -o tools/ubench/lds_tr_patterns && tools/ubench/lds_tr_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) short s16x4;

template <int PAT> __device__ __forceinline__ int swz(int r) {
  if (PAT == 0) return ((r >> 1) & 1) << 1;                                   // round 2-3: bit 1 only
  if (PAT == 1) return (((r >> 1) & 1) | (((r >> 3) & 1) << 1)) << 1;         // round 4: bits 1 and 3
  if (PAT == 2) return 0;                                                      // no swizzle
  return (r & 3) << 1;                                                         // bits 0-1
}

template <int PAT>
__global__ __launch_bounds__(512) void k(float* out, int iters, int shift) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 64 KB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 65536 / 4; i += 512) ((float*)smem)[i] = (float)i;
  __syncthreads();
  int addr[8];
  const int p = lane & 15, q = lane >> 4;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    if (PAT == 4) {          // linear: 8 contiguous bytes per lane
      addr[s] = (wave * 4096 + s * 512 + lane * 8) & 0xffff;
    } else {
      const int h = s & 1, tap = s >> 1;                       // 4 "taps" = shifts 0..3 of the row index
      const int pos = wave * 40 + shift + tap + q * 8 + h * 4 + (p >> 2);
      const int col = (wave & 3) * 16 + (p & 3) * 4;
      addr[s] = (pos * 128 + (((col >> 3) ^ swz<PAT>(pos)) * 16) + (col & 7) * 2) & 0xffff;
    }
  }
  s16x4 acc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    s16x4 v[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) v[s] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + addr[s]));
#pragma unroll
    for (int s = 0; s < 8; ++s) acc += v[s];
    asm volatile("" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x * 2] = (float)(t1 - t0) / (float)(iters * 8);
  if (acc[0] == 0x1234) out[1] = 1.f;
}

template <int PAT>
static void run(const char* name, int shift) {
  float* d;
  (void)hipMalloc(&d, 8192);
  (void)hipFuncSetAttribute((const void*)k<PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 90000);
  k<PAT><<<256, 512, 90000>>>(d, 2000, shift);      // 90 KB: one workgroup per CU
  k<PAT><<<256, 512, 90000>>>(d, 2000, shift);
  float h[2];
  (void)hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
  printf("%-36s shift %2d: %6.2f cycles per ds_read_b64_tr_b16 per wave, 8 waves / CU -> %5.2f LDS cycles per instruction if LDS-bound\n", name, shift, h[0], h[0] / 8);
  (void)hipFree(d);
}

int main() {
  for (int shift : {0, 1, 2, 5}) {
    run<0>("^ bit 1 of the row (rounds 2-3)", shift);
    run<1>("^ bits 1 and 3 of the row (round 4)", shift);
    run<2>("no swizzle", shift);
    run<3>("^ (row & 3)", shift);
    run<4>("linear 8 B per lane", shift);
  }
  return 0;
}
