// Does a kernel's SCALAR load (s_load through the scalar data cache) see what an EARLIER kernel of the same stream -- or of another
// stream ordered by an event -- stored with vector stores?  (round 5: audionet.conv7.bias / batchnorm7.weight gradients differed
// run to run by ~2e-4 in elements j = k mod 8 of 16 consecutive blocks; the finalize kernels read `out[j]` with s_load_dword.)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/kcache_stale.hip -o tools/ubench/kcache_stale && tools/ubench/kcache_stale
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// one block per element: uniform address -> the compiler emits s_load_dword for x[blockIdx.x]
__global__ void rmw_scalar(float* x, float add) {
  const float old = x[blockIdx.x];
  if (threadIdx.x == 0) x[blockIdx.x] = old + add;
}
__global__ void fill_vec(float* x, int n, float v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = v;
}

// a long-running kernel of ANOTHER stream on most CUs (the weight-gradient stream's persistent kernels beside the main stream)
__global__ void busy(const float* y, float* sink, long cycles) {
  const long t0 = clock64();
  float acc = 0.f;
  while (clock64() - t0 < cycles) acc += y[blockIdx.x & 63];      // (uniform address: scalar loads keep the scalar cache busy)
  if (acc == 123.456f) sink[0] = acc;
}

int main() {
  const int n = 4096;
  float *x, *h = (float*)malloc(n * 4);
  CK(hipMalloc(&x, n * 4));
  hipStream_t s1, s2;
  CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  long bad_same = 0, bad_cross = 0, bad_realloc = 0;
  float *yb, *sink;
  CK(hipMalloc(&yb, 4096)); CK(hipMalloc(&sink, 64)); CK(hipMemset(yb, 0, 4096));
  hipStream_t s3; CK(hipStreamCreate(&s3));
  const bool with_busy = getenv("BUSY") != nullptr;
  for (int it = 0; it < 2000; ++it) {
    if (with_busy && (it % 4) == 0) hipLaunchKernelGGL(busy, dim3(208), dim3(512), 65536, s3, yb, sink, 2000000L);   // ~1 ms, one workgroup per CU (64 KB of LDS)
    // same stream: fill(7) ; rmw(+1) -> 8 ; fill(0) ; rmw(+1) -> must be 1
    hipLaunchKernelGGL(fill_vec, dim3(n / 256), dim3(256), 0, s1, x, n, 7.f);
    hipLaunchKernelGGL(rmw_scalar, dim3(n), dim3(64), 0, s1, x, 1.f);
    hipLaunchKernelGGL(fill_vec, dim3(n / 256), dim3(256), 0, s1, x, n, 0.f);
    hipLaunchKernelGGL(rmw_scalar, dim3(n), dim3(64), 0, s1, x, 1.f);
    CK(hipMemcpyAsync(h, x, n * 4, hipMemcpyDeviceToHost, s1));
    CK(hipStreamSynchronize(s1));
    for (int i = 0; i < n; ++i) bad_same += h[i] != 1.f;
    // cross stream: fill on s2, event, rmw on s1
    hipLaunchKernelGGL(fill_vec, dim3(n / 256), dim3(256), 0, s2, x, n, 3.f);
    CK(hipEventRecord(ev, s2)); CK(hipStreamWaitEvent(s1, ev, 0));
    hipLaunchKernelGGL(rmw_scalar, dim3(n), dim3(64), 0, s1, x, 1.f);
    CK(hipMemcpyAsync(h, x, n * 4, hipMemcpyDeviceToHost, s1));
    CK(hipStreamSynchronize(s1));
    for (int i = 0; i < n; ++i) bad_cross += h[i] != 4.f;
    // the address changes owner through a memset (what torch.zeros / a DMA upload does)
    CK(hipMemsetAsync(x, 0, n * 4, s1));
    hipLaunchKernelGGL(rmw_scalar, dim3(n), dim3(64), 0, s1, x, 2.f);
    CK(hipMemcpyAsync(h, x, n * 4, hipMemcpyDeviceToHost, s1));
    CK(hipStreamSynchronize(s1));
    for (int i = 0; i < n; ++i) bad_realloc += h[i] != 2.f;
  }
  printf("%s", with_busy ? "(beside a persistent kernel of another stream on 208 CUs) " : "");
  printf("stale scalar reads over 2000 rounds x %d elements: same stream %ld, cross stream (event) %ld, after memset %ld\n", n, bad_same, bad_cross, bad_realloc);
  return 0;
}
