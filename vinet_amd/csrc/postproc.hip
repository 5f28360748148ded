// Saliency-map post-processing: what generate_result.py:95-104 `process()` and
// train.py:251-253 `validate()` do to a predicted map on the HOST with cv2 / torchvision, as two device kernels:
//
//   vinet_resize_blur    cv2.resize(smap, (oW, oH)) [INTER_LINEAR] -> cv2.GaussianBlur(., (11, 11), 0) [sigma 2.0,
//                        BORDER_REFLECT_101] (utils.py:61-64), optionally with the per-map min / max of the result;
//   vinet_normalize_u8   utils.py:66-78 img_save(normalize=True): make_grid's min-max normalisation, * 255 + 0.5, clamp,
//                        round half to even, uint8.
//
// The arithmetic follows oracle/postproc_cpu.py operation by operation (float32, no contraction: the library is built
// with -ffp-contract=off), so the uint8 maps are bit-identical to the restatement's.
//
// resize_blur is one launch: a 256-thread workgroup owns a 32 x 64 tile of the output.  It (1) evaluates the resized
// image on the tile plus a 5-pixel halo (42 x 74 samples, four L2 reads each; halo positions outside the image are
// reflected FIRST, so the tile never depends on a neighbour) into LDS, (2) runs the 11-tap row filter from LDS into a
// second LDS plane (42 x 64), (3) the symmetric column filter into registers, stores, and folds the tile's min / max
// into the map's with one wave reduction and one atomic pair per wave.  The source map is read once (+ halo), the
// output written once; a 224 x 384 -> 360 x 640 map is 60 workgroups, so batches of maps fill the chip.
#include "common.h"

namespace {

constexpr int PP_K = 11, PP_R = 5, PP_TH = 32, PP_TW = 64;
constexpr int PP_RH = PP_TH + 2 * PP_R, PP_RW = PP_TW + 2 * PP_R;        // 42 x 74
constexpr int PP_RLD = PP_RW + 1, PP_HLD = PP_TW + 1;

struct PPKernel { float k[PP_K]; };

// cv::getGaussianKernel(11, -1, CV_32F): t_i = (float)exp(-0.5 / sigma^2 * (i - 5)^2), k_i = (float)(t_i / sum t)
static PPKernel pp_gaussian() {
  PPKernel g;
  const double sigma = 0.3 * ((PP_K - 1) * 0.5 - 1) + 0.8;
  const double scale2x = -0.5 / (sigma * sigma);
  double sum = 0;
  for (int i = 0; i < PP_K; ++i) {
    const double x = i - (PP_K - 1) * 0.5;
    g.k[i] = (float)exp(scale2x * x * x);
    sum += g.k[i];
  }
  const double inv = 1.0 / sum;
  for (int i = 0; i < PP_K; ++i) g.k[i] = (float)(g.k[i] * inv);
  return g;
}

VN_DEV int pp_reflect101(int i, int n) {
  if (n == 1) return 0;
  while ((unsigned)i >= (unsigned)n) i = i < 0 ? -i : 2 * (n - 1) - i;
  return i;
}

// order-preserving map float -> uint32 (for atomicMin / atomicMax on floats of either sign) and back
VN_DEV uint32_t pp_key(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
VN_DEV float pp_unkey(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__global__ void pp_minmax_init_kernel(uint32_t* mm, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) { mm[2 * i] = 0xffffffffu; mm[2 * i + 1] = 0u; }
}

__global__ __launch_bounds__(256) void resize_blur_kernel(const float* __restrict__ src, int H, int W, float* __restrict__ dst,
                                                          int oH, int oW, double scale_y, double scale_x, PPKernel g,
                                                          uint32_t* __restrict__ minmax) {
  __shared__ float Rz[PP_RH * PP_RLD];        // resized samples, tile + halo
  __shared__ float Hz[PP_RH * PP_HLD];        // after the row filter
  const int tid = threadIdx.x;
  const int b = blockIdx.z, y0 = blockIdx.y * PP_TH, x0 = blockIdx.x * PP_TW;
  const float* s = src + (long)b * H * W;

  // (1) the resized image on the tile + halo
  for (int i = tid; i < PP_RH * PP_RW; i += 256) {
    const int ly = i / PP_RW, lx = i - ly * PP_RW;
    const int gy = pp_reflect101(y0 + ly - PP_R, oH), gx = pp_reflect101(x0 + lx - PP_R, oW);
    // resize.cpp: f = (float)((d + 0.5) * scale - 0.5); s = floor(f); f -= s
    float fx = (float)((gx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { sx = 0; fx = 0.f; }
    if (sx >= W - 1) { sx = W - 1; fx = 0.f; }
    const int sx1 = sx + 1 < W ? sx + 1 : W - 1;
    float fy = (float)((gy + 0.5) * scale_y - 0.5);
    int sy = (int)floorf(fy);
    fy -= (float)sy;
    const int sy0 = sy < 0 ? 0 : (sy > H - 1 ? H - 1 : sy), sy1 = sy + 1 < 0 ? 0 : (sy + 1 > H - 1 ? H - 1 : sy + 1);
    const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
    const float r0 = s[(long)sy0 * W + sx] * a0 + s[(long)sy0 * W + sx1] * a1;      // HResizeLinear
    const float r1 = s[(long)sy1 * W + sx] * a0 + s[(long)sy1 * W + sx1] * a1;
    Rz[ly * PP_RLD + lx] = r0 * b0 + r1 * b1;                                         // VResizeLinear
  }
  __syncthreads();
  // (2) row filter (RowFilter: s = k0*x0; s += kj*xj)
  for (int i = tid; i < PP_RH * PP_TW; i += 256) {
    const int ly = i / PP_TW, lx = i - ly * PP_TW;
    const float* r = Rz + ly * PP_RLD + lx;
    float acc = r[0] * g.k[0];
#pragma unroll
    for (int j = 1; j < PP_K; ++j) acc = acc + r[j] * g.k[j];
    Hz[ly * PP_HLD + lx] = acc;
  }
  __syncthreads();
  // (3) column filter (SymmColumnFilter: s = k5*x0 + sum kj*(x+j + x-j)), store, min / max
  float vmin = INFINITY, vmax = -INFINITY;
  for (int i = tid; i < PP_TH * PP_TW; i += 256) {
    const int ly = i / PP_TW, lx = i - ly * PP_TW;
    const float* c = Hz + (ly + PP_R) * PP_HLD + lx;
    float acc = c[0] * g.k[PP_R];
#pragma unroll
    for (int j = 1; j <= PP_R; ++j) acc = acc + (c[j * PP_HLD] + c[-j * PP_HLD]) * g.k[PP_R + j];
    const int gy = y0 + ly, gx = x0 + lx;
    if (gy < oH && gx < oW) {
      dst[((long)b * oH + gy) * oW + gx] = acc;
      vmin = fminf(vmin, acc); vmax = fmaxf(vmax, acc);
    }
  }
  if (minmax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      vmin = fminf(vmin, __shfl_xor(vmin, o));
      vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    }
    if ((tid & 63) == 0 && vmin <= vmax) {
      atomicMin(minmax + 2 * b, pp_key(vmin));
      atomicMax(minmax + 2 * b + 1, pp_key(vmax));
    }
  }
}

__global__ __launch_bounds__(256) void normalize_u8_kernel(const float* __restrict__ src, const uint32_t* __restrict__ minmax,
                                                           long n, uint8_t* __restrict__ dst) {
  const int b = blockIdx.y;
  const float mn = pp_unkey(minmax[2 * b]), mx = pp_unkey(minmax[2 * b + 1]);
  const float neg_mn = -mn;
  const float div = (float)((double)mx - (double)mn + 1e-5);       // make_grid: max - min + 1e-5 as a Python float
  const float* s = src + (long)b * n;
  uint8_t* d = dst + (long)b * n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float x = s[i];
    x = fminf(fmaxf(x, mn), mx);
    x = (x + neg_mn) / div;
    float y = x * 255.f + 0.5f;
    y = fminf(fmaxf(y, 0.f), 255.f);
    d[i] = (uint8_t)rintf(y);                                      // round half to even
  }
}

__global__ __launch_bounds__(256) void minmax_kernel(const float* __restrict__ src, long n, uint32_t* __restrict__ minmax) {
  const int b = blockIdx.y;
  const float* s = src + (long)b * n;
  float vmin = INFINITY, vmax = -INFINITY;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float x = s[i];
    vmin = fminf(vmin, x); vmax = fmaxf(vmax, x);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    vmin = fminf(vmin, __shfl_xor(vmin, o));
    vmax = fmaxf(vmax, __shfl_xor(vmax, o));
  }
  if ((threadIdx.x & 63) == 0 && vmin <= vmax) {
    atomicMin(minmax + 2 * b, pp_key(vmin));
    atomicMax(minmax + 2 * b + 1, pp_key(vmax));
  }
}

}  // namespace

extern "C" int vinet_resize_blur(const float* src, int32_t B, int32_t H, int32_t W, float* dst, int32_t oH, int32_t oW,
                                 uint32_t* minmax, void* stream) {
  VN_CHECK_ARG(src && dst && B > 0 && H > 0 && W > 0 && oH > 0 && oW > 0, "resize_blur: bad arguments");
  VN_CHECK_ARG(B <= 65535 && (oH + PP_TH - 1) / PP_TH <= 65535, "resize_blur: grid too large");
  static const PPKernel g = pp_gaussian();
  hipStream_t st = (hipStream_t)stream;
  if (minmax) hipLaunchKernelGGL(pp_minmax_init_kernel, dim3((B + 255) / 256), dim3(256), 0, st, minmax, B);
  // resize.cpp: inv_scale = dsize / ssize; scale = 1 / inv_scale
  const double scale_x = 1.0 / ((double)oW / (double)W), scale_y = 1.0 / ((double)oH / (double)H);
  hipLaunchKernelGGL(resize_blur_kernel, dim3((oW + PP_TW - 1) / PP_TW, (oH + PP_TH - 1) / PP_TH, B), dim3(256), 0, st, src, H, W,
                     dst, oH, oW, scale_y, scale_x, g, minmax);
  return vn_launch_status("resize_blur");
}

extern "C" int vinet_minmax(const float* src, int32_t B, int64_t n, uint32_t* minmax, void* stream) {
  VN_CHECK_ARG(src && minmax && B > 0 && B <= 65535 && n > 0, "minmax: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(pp_minmax_init_kernel, dim3((B + 255) / 256), dim3(256), 0, st, minmax, B);
  long blocks = (n + 256 * 8 - 1) / (256 * 8);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(minmax_kernel, dim3((unsigned)blocks, B), dim3(256), 0, st, src, (long)n, minmax);
  return vn_launch_status("minmax");
}

extern "C" int vinet_normalize_u8(const float* src, const uint32_t* minmax, int32_t B, int64_t n, uint8_t* dst, void* stream) {
  VN_CHECK_ARG(src && minmax && dst && B > 0 && B <= 65535 && n > 0, "normalize_u8: bad arguments");
  long blocks = (n + 256 * 8 - 1) / (256 * 8);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(normalize_u8_kernel, dim3((unsigned)blocks, B), dim3(256), 0, (hipStream_t)stream, src, minmax, (long)n, dst);
  return vn_launch_status("normalize_u8");
}
