#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* p) {
  unsigned x = threadIdx.x, y = 100 + threadIdx.x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
  p[threadIdx.x] = x; p[64 + threadIdx.x] = y;
}
int main() {
  unsigned* d; hipMalloc(&d, 512); k<<<1, 64>>>(d); unsigned h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; i += 8) printf("lane %2d: x=%3u y=%3u\n", i, h[i], h[64 + i]);
  return 0;
}
