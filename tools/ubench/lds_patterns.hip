// ds_read_b128 fragment-read patterns of the MFMA conv kernels: LDS cycles per wave-instruction for each swizzle.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_patterns.hip -o tools/ubench/lds_patterns && tools/ubench/lds_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int PAT>
__device__ __forceinline__ int frag_addr(int lane, int base_row, int s) {
  if (PAT == 0) {          // 16x16x32: 16 rows x 4 chunks, chunk ^ (row & 7)
    const int row = base_row + (s >> 1) * 16 + (lane & 15), c = (s & 1) * 4 + (lane >> 4);
    return row * 128 + ((c ^ (row & 7)) << 4);
  } else if (PAT == 1) {   // 32x32x16: 32 rows x 2 chunks, chunk ^ ((row >> 1) & 7)
    const int row = base_row + (lane & 31), c = (s & 3) * 2 + (lane >> 5);
    return row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
  } else if (PAT == 2) {   // 32 rows x 2 chunks, chunk ^ (row & 7)
    const int row = base_row + (lane & 31), c = (s & 3) * 2 + (lane >> 5);
    return row * 128 + ((c ^ (row & 7)) << 4);
  } else if (PAT == 3) {   // 32 rows x 2 chunks, no swizzle
    const int row = base_row + (lane & 31), c = (s & 3) * 2 + (lane >> 5);
    return row * 128 + (c << 4);
  } else if (PAT == 4) {   // 32 rows, chunk pairs assigned by lane parity of half: lanes 0-31 -> rows, chunk = 2s + ((lane>>5) ^ (row & 1))
    const int row = base_row + (lane & 31), c = (s & 3) * 2 + ((lane >> 5) ^ (row & 1));
    return row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
  } else {                 // linear 16 B per lane (conflict-free by construction)
    return (base_row * 128 + lane * 16 + s * 1024) & 0x7fff;
  }
}

template <int PAT>
__global__ __launch_bounds__(256) void k(float* out, int iters, int shift) {
  extern __shared__ char smem[];   // 32 KB per workgroup: 4 workgroups = 16 waves per CU
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 32768 / 4; i += 256) ((float*)smem)[i] = (float)i;
  __syncthreads();
  int addr[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) addr[s] = frag_addr<PAT>(lane, wave * 64 + shift + (PAT == 0 ? 0 : (s >> 2) * 32), s) & 0x7fff;
  u32x4 acc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    u32x4 v[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) v[s] = *(const u32x4*)(smem + addr[s]);
#pragma unroll
    for (int s = 0; s < 8; ++s) acc += v[s];
    asm volatile("" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x * 2] = (float)(t1 - t0) / (float)(iters * 8);
  if (acc[0] == 0x12345678u) out[1] = 1.f;
}

template <int PAT>
static void run(const char* name, int shift) {
  float* d;
  (void)hipMalloc(&d, 4096);
  (void)hipFuncSetAttribute((const void*)k<PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
  k<PAT><<<1024, 256, 32768>>>(d, 2000, shift);
  k<PAT><<<1024, 256, 32768>>>(d, 2000, shift);
  float h[2];
  (void)hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
  printf("%-44s shift %2d: %6.2f cycles per ds_read_b128 per wave (16 waves / CU: x16 = LDS cycles per CU and instruction if LDS-bound)\n", name, shift, h[0]);
  (void)hipFree(d);
}

int main() {
  for (int shift : {0, 1, 19}) {
    run<0>("16 rows x 4 chunks, ^ (row & 7)", shift);
    run<1>("32 rows x 2 chunks, ^ ((row >> 1) & 7)", shift);
    run<2>("32 rows x 2 chunks, ^ (row & 7)", shift);
    run<3>("32 rows x 2 chunks, no swizzle", shift);
    run<4>("32 rows, half ^ row parity, ^ ((row>>1)&7)", shift);
    run<5>("linear", shift);
  }
  return 0;
}
