// Input pipeline of the reference on device: decoded uint8 frames and ground-truth maps are
// uploaded as bytes, at their own resolution, and become the network's float32 inputs here.
//
//   vinet_frames_preprocess   dataloader.py:243-250 / generate_result.py:77-88  img_transform:
//                             transforms.Resize((oH, oW)) [= PIL Image.resize(BILINEAR) on RGB bytes] -> ToTensor (/255)
//                             -> Normalize(mean, std);  uint8 [N][H][W][3] -> float32 [N][3][oH][oW]
//   vinet_gt_preprocess       dataloader.py:283-296: uint8 'L' map -> float64 -> (train) cv2.resize(gt, (oW, oH)) -> / 255 when
//                             the map's maximum exceeds 1 -> float32 [N][oH][oW]
//
// PIL's 8-bit resampler is integer arithmetic (libImaging/Resample.c): per axis the triangle filter is sampled over
// [center - support, center + support), support = max(in / out, 1), normalised in double, converted to 22-bit fixed point; the
// horizontal pass runs first and its result is ROUNDED TO BYTES before the vertical pass.  The kernels keep exactly that:
// coefficients are computed on device in double (the same IEEE operations in the same order, no contraction), pass 1 writes
// the byte image [N][H][oW][3] to the caller's scratch, pass 2 reads it, clips, divides by 255, normalises and scatters to the
// three planes.  Results equal PIL's byte for byte (oracle/preproc_cpu.py is pinned against the real Image.resize).
#include "common.h"

namespace {

constexpr int RS_BITS = 32 - 8 - 2;      // Resample.c PRECISION_BITS

VN_DEV double rs_triangle(double x) {
  if (x < 0.0) x = -x;
  return x < 1.0 ? 1.0 - x : 0.0;
}

// Resample.c precompute_coeffs + normalize_coeffs_8bpc, one thread per output index
__global__ void resample_coeffs_kernel(int in_size, int out_size, double scale, int ksize, int* __restrict__ bounds,
                                       int* __restrict__ kk) {
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  if (xx >= out_size) return;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  const double ss = 1.0 / filterscale;
  const double center = (xx + 0.5) * scale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) ww += rs_triangle((x + xmin - center + 0.5) * ss);
  int* k = kk + (long)xx * ksize;
  for (int x = 0; x < ksize; ++x) {
    int v = 0;
    if (x < xmax) {
      double w = rs_triangle((x + xmin - center + 0.5) * ss);
      if (ww != 0.0) w /= ww;
      v = w < 0 ? (int)(-0.5 + w * (double)(1 << RS_BITS)) : (int)(0.5 + w * (double)(1 << RS_BITS));
    }
    k[x] = v;
  }
  bounds[2 * xx] = xmin;
  bounds[2 * xx + 1] = xmax;
}

VN_DEV int rs_clip8(int acc) {
  const int v = acc >> RS_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// pass 1: [N][H][W][3] -> [N][H][oW][3], one thread per output pixel (three channels)
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* __restrict__ src, int H, int W, int oW, int ksize,
                                                         const int* __restrict__ bounds, const int* __restrict__ kk,
                                                         uint8_t* __restrict__ tmp, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int xx = (int)(i % oW);
  const long row = i / oW;                               // n * H + y
  const int xmin = bounds[2 * xx], xmax = bounds[2 * xx + 1];
  const int* k = kk + (long)xx * ksize;
  const uint8_t* s = src + (row * W + xmin) * 3;
  int a0 = 1 << (RS_BITS - 1), a1 = a0, a2 = a0;
  for (int x = 0; x < xmax; ++x) {
    const int w = k[x];
    a0 += (int)s[3 * x] * w; a1 += (int)s[3 * x + 1] * w; a2 += (int)s[3 * x + 2] * w;
  }
  uint8_t* d = tmp + i * 3;
  d[0] = (uint8_t)rs_clip8(a0); d[1] = (uint8_t)rs_clip8(a1); d[2] = (uint8_t)rs_clip8(a2);
}

struct Norm3 { float mean[3], std[3]; };

// pass 2: [N][H][oW][3] bytes -> float32 planes [N][3][oH][oW]: vertical resample, / 255, (x - mean) / std
__global__ __launch_bounds__(256) void resample_v_norm_kernel(const uint8_t* __restrict__ tmp, int H, int oH, int oW, int ksize,
                                                              const int* __restrict__ bounds, const int* __restrict__ kk,
                                                              Norm3 nm, float* __restrict__ dst, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % oW);
  const long r = i / oW;
  const int yy = (int)(r % oH);
  const long n = r / oH;
  const int ymin = bounds[2 * yy], ymax = bounds[2 * yy + 1];
  const int* k = kk + (long)yy * ksize;
  const uint8_t* s = tmp + ((n * H + ymin) * (long)oW + x) * 3;
  int a0 = 1 << (RS_BITS - 1), a1 = a0, a2 = a0;
  for (int y = 0; y < ymax; ++y) {
    const int w = k[y];
    const uint8_t* p = s + (long)y * oW * 3;
    a0 += (int)p[0] * w; a1 += (int)p[1] * w; a2 += (int)p[2] * w;
  }
  const long plane = (long)oH * oW;
  float* d = dst + n * 3 * plane + (long)yy * oW + x;
  d[0] = ((float)rs_clip8(a0) / 255.f - nm.mean[0]) / nm.std[0];
  d[plane] = ((float)rs_clip8(a1) / 255.f - nm.mean[1]) / nm.std[1];
  d[2 * plane] = ((float)rs_clip8(a2) / 255.f - nm.mean[2]) / nm.std[2];
}

// ---- ground-truth maps ------------------------------------------------------------------------------------------------
// cv2.resize on CV_64F: HResizeLinear<double, double, float> / VResizeLinear: float32 coefficients, double arithmetic.
__global__ __launch_bounds__(256) void gt_resize_kernel(const uint8_t* __restrict__ src, int H, int W, int oH, int oW, double scale_y,
                                                        double scale_x, double* __restrict__ out, unsigned long long* __restrict__ maxkey) {
  const int b = blockIdx.y;
  const long n = (long)oH * oW;
  const uint8_t* s = src + (long)b * H * W;
  double vmax = 0.0;                                      // maps are non-negative
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int gy = (int)(i / oW), gx = (int)(i - (long)gy * oW);
    double v;
    if (oH == H && oW == W) {
      v = (double)s[i];
    } else {
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
      const double a0 = (double)(1.f - fx), a1 = (double)fx, b0 = (double)(1.f - fy), b1 = (double)fy;
      const double r0 = (double)s[(long)sy0 * W + sx] * a0 + (double)s[(long)sy0 * W + sx1] * a1;
      const double r1 = (double)s[(long)sy1 * W + sx] * a0 + (double)s[(long)sy1 * W + sx1] * a1;
      v = r0 * b0 + r1 * b1;
    }
    out[(long)b * n + i] = v;
    vmax = fmax(vmax, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) vmax = fmax(vmax, __shfl_xor(vmax, o));
  if ((threadIdx.x & 63) == 0) atomicMax(maxkey + b, (unsigned long long)__double_as_longlong(vmax));   // non-negative doubles order as integers
}

__global__ void gt_key_init_kernel(unsigned long long* k, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) k[i] = 0ull;
}

__global__ __launch_bounds__(256) void gt_scale_kernel(const double* __restrict__ in, const unsigned long long* __restrict__ maxkey, long n,
                                                       float* __restrict__ dst) {
  const int b = blockIdx.y;
  const bool div = __longlong_as_double((long long)maxkey[b]) > 1.0;          // `if np.max(gt) > 1.0: gt = gt / 255.0`
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const double v = in[(long)b * n + i];
    dst[(long)b * n + i] = (float)(div ? v / 255.0 : v);
  }
}

// dataloader.py:106-121: out[off + i] = float(hanning(M)[i]) * wav[start + i], zeros elsewhere.  numpy 1.18.5 (requirements.txt:91):
// hanning(M) = 0.5 - 0.5 * cos(2 pi n / (M - 1)), n = 0..M-1, in double; M == 1 -> [1.0]
__global__ __launch_bounds__(256) void audio_excerpt_kernel(const float* __restrict__ wav, long start, long M, long off, long win,
                                                            float* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= win) return;
  const long j = i - off;
  float v = 0.f;
  if (j >= 0 && j < M) {
    const double h = M == 1 ? 1.0 : 0.5 - 0.5 * cos(2.0 * 3.141592653589793 * (double)j / (double)(M - 1));
    v = (float)h * wav[start + j];
  }
  out[i] = v;
}

static int rs_ksize(int in_size, int out_size) {
  double fs = (double)in_size / (double)out_size;
  if (fs < 1.0) fs = 1.0;
  return (int)ceil(1.0 * fs) * 2 + 1;
}
static long rs_align(long v) { return (v + 255) / 256 * 256; }

struct PreWs { long bh, kh, bv, kv, tmp, total; };
static PreWs pre_ws(int N, int H, int W, int oH, int oW) {
  PreWs w;
  long o = 0;
  w.bh = o; o += rs_align((long)oW * 2 * 4);
  w.kh = o; o += rs_align((long)oW * rs_ksize(W, oW) * 4);
  w.bv = o; o += rs_align((long)oH * 2 * 4);
  w.kv = o; o += rs_align((long)oH * rs_ksize(H, oH) * 4);
  w.tmp = o; o += rs_align((long)N * H * oW * 3);
  w.total = o;
  return w;
}

}  // namespace

extern "C" int64_t vinet_frames_preprocess_ws_bytes(int32_t N, int32_t H, int32_t W, int32_t oH, int32_t oW) {
  if (N <= 0 || H <= 0 || W <= 0 || oH <= 0 || oW <= 0) return -1;
  return pre_ws(N, H, W, oH, oW).total;
}

extern "C" int vinet_frames_preprocess(const uint8_t* src, int32_t N, int32_t H, int32_t W, float* dst, int32_t oH, int32_t oW,
                                       const float* mean_std, void* ws, void* stream) {
  VN_CHECK_ARG(src && dst && mean_std && ws && N > 0 && H > 0 && W > 0 && oH > 0 && oW > 0, "frames_preprocess: bad arguments");
  VN_CHECK_ARG(((uintptr_t)ws % 16) == 0, "frames_preprocess: scratch must be 16-byte aligned");
  const PreWs w = pre_ws(N, H, W, oH, oW);
  char* base = (char*)ws;
  int *bh = (int*)(base + w.bh), *kh = (int*)(base + w.kh), *bv = (int*)(base + w.bv), *kv = (int*)(base + w.kv);
  uint8_t* tmp = (uint8_t*)(base + w.tmp);
  const int ksh = rs_ksize(W, oW), ksv = rs_ksize(H, oH);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(resample_coeffs_kernel, dim3((oW + 127) / 128), dim3(128), 0, st, W, oW, (double)W / (double)oW, ksh, bh, kh);
  hipLaunchKernelGGL(resample_coeffs_kernel, dim3((oH + 127) / 128), dim3(128), 0, st, H, oH, (double)H / (double)oH, ksv, bv, kv);
  const long t1 = (long)N * H * oW, t2 = (long)N * oH * oW;
  hipLaunchKernelGGL(resample_h_kernel, dim3((unsigned)((t1 + 255) / 256)), dim3(256), 0, st, src, H, W, oW, ksh, bh, kh, tmp, t1);
  Norm3 nm;
  for (int c = 0; c < 3; ++c) { nm.mean[c] = mean_std[c]; nm.std[c] = mean_std[3 + c]; }
  hipLaunchKernelGGL(resample_v_norm_kernel, dim3((unsigned)((t2 + 255) / 256)), dim3(256), 0, st, tmp, H, oH, oW, ksv, bv, kv, nm, dst, t2);
  return vn_launch_status("frames_preprocess");
}

extern "C" int64_t vinet_gt_preprocess_ws_bytes(int32_t N, int32_t oH, int32_t oW) {
  if (N <= 0 || oH <= 0 || oW <= 0) return -1;
  return rs_align((long)N * 8) + (long)N * oH * oW * 8;
}

extern "C" int vinet_gt_preprocess(const uint8_t* src, int32_t N, int32_t H, int32_t W, float* dst, int32_t oH, int32_t oW, void* ws,
                                   void* stream) {
  VN_CHECK_ARG(src && dst && ws && N > 0 && N <= 65535 && H > 0 && W > 0 && oH > 0 && oW > 0, "gt_preprocess: bad arguments");
  VN_CHECK_ARG(((uintptr_t)ws % 16) == 0, "gt_preprocess: scratch must be 16-byte aligned");
  unsigned long long* keys = (unsigned long long*)ws;
  double* buf = (double*)((char*)ws + rs_align((long)N * 8));
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gt_key_init_kernel, dim3((N + 255) / 256), dim3(256), 0, st, keys, N);
  const long n = (long)oH * oW;
  long blocks = (n + 256 * 4 - 1) / (256 * 4);
  if (blocks > 1024) blocks = 1024;
  const double scale_x = 1.0 / ((double)oW / (double)W), scale_y = 1.0 / ((double)oH / (double)H);
  hipLaunchKernelGGL(gt_resize_kernel, dim3((unsigned)blocks, N), dim3(256), 0, st, src, H, W, oH, oW, scale_y, scale_x, buf, keys);
  hipLaunchKernelGGL(gt_scale_kernel, dim3((unsigned)blocks, N), dim3(256), 0, st, buf, keys, n, dst);
  return vn_launch_status("gt_preprocess");
}

extern "C" int vinet_audio_excerpt(const float* wav, int64_t n_samples, int64_t start, int64_t end, float* out, int32_t win, void* stream) {
  VN_CHECK_ARG(wav && out && n_samples >= 0 && win > 0 && start >= 0, "audio_excerpt: bad arguments");
  // wav[:, start:end+1] (dataloader.py:104): Python slicing clamps to the waveform
  long hi = end + 1 < n_samples ? end + 1 : n_samples;
  long M = hi - start;
  if (M < 0) M = 0;
  VN_CHECK_ARG(M <= win, "audio_excerpt: the excerpt (%ld samples) is longer than the window (%d)", M, win);
  const long off = win / 2 - M / 2;
  hipLaunchKernelGGL(audio_excerpt_kernel, dim3((win + 255) / 256), dim3(256), 0, (hipStream_t)stream, wav, (long)start, M, off, (long)win, out);
  return vn_launch_status("audio_excerpt");
}
