// (1,2,2) trilinear 2x upsample (nn.Upsample, model.py:254) forward / backward.
#include "elementwise.h"

// ============================================================================
// Upsample (1,2,2) trilinear, align_corners=False  (separable .25/.75 stencil)
// ============================================================================
template <typename T>
__global__ VN_NO_PK_F32 void upsample2x_kernel(TView x, TView y, long total) {     // (VN_NO_PK_F32: common.h -- hipcc emitted the affected form here)
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint32_t vox_u = fdiv((uint32_t)i, y.dQ);
  const int q = (int)((uint32_t)i - vox_u * (uint32_t)(y.C / 4));
  const long vox = (long)vox_u;
  int b, t, ho, wo;
  decode_vox(y, vox, b, t, ho, wo);
  // src = max((o + .5)/2 - .5, 0); i0 = floor(src); l1 = src - i0; i1 = i0 + (i0 < n-1)
  const float sh = fmaxf((ho + 0.5f) * 0.5f - 0.5f, 0.f), sw = fmaxf((wo + 0.5f) * 0.5f - 0.5f, 0.f);
  const int h0 = (int)sh, w0 = (int)sw;
  const int h1 = h0 + (h0 < x.H - 1 ? 1 : 0), w1 = w0 + (w0 < x.W - 1 ? 1 : 0);
  const float lh1 = sh - h0, lh0 = 1.f - lh1, lw1 = sw - w0, lw0 = 1.f - lw1;
  const T* base = (const T*)x.p;
  const float4 v00 = ldq<T>(base + vox_off(x, b, t, h0, w0) + q * 4), v01 = ldq<T>(base + vox_off(x, b, t, h0, w1) + q * 4);
  const float4 v10 = ldq<T>(base + vox_off(x, b, t, h1, w0) + q * 4), v11 = ldq<T>(base + vox_off(x, b, t, h1, w1) + q * 4);
  float4 o;
  o.x = lh0 * (lw0 * v00.x + lw1 * v01.x) + lh1 * (lw0 * v10.x + lw1 * v11.x);
  o.y = lh0 * (lw0 * v00.y + lw1 * v01.y) + lh1 * (lw0 * v10.y + lw1 * v11.y);
  o.z = lh0 * (lw0 * v00.z + lw1 * v01.z) + lh1 * (lw0 * v10.z + lw1 * v11.z);
  o.w = lh0 * (lw0 * v00.w + lw1 * v01.w) + lh1 * (lw0 * v10.w + lw1 * v11.w);
  stq<T>((T*)y.p + vox_off(y, b, t, ho, wo) + q * 4, o);
}

// 8-channel, 2 x 2-output-block form: one lane owns input voxel (h, w) x 8 channels and writes outputs
// (2h..2h+1, 2w..2w+1) from the 3 x 3 clamped neighbourhood it loads once -- voxel decode and addressing are paid
// once per 64 bytes written (the one-output-quad kernel above spends more time on indices than on data: 2.2 TB/s).
// Same arithmetic per output as above: o = lh0*(lw0*v00 + lw1*v01) + lh1*(lw0*v10 + lw1*v11).
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_blk8_kernel(TView x, TView y, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int G = x.C >> 3;
  const long vox = i / G;
  const int g = (int)(i - vox * G);
  int b, t, h, w;
  decode_vox(x, vox, b, t, h, w);
  const int hm = h > 0 ? h - 1 : 0, hp = h < x.H - 1 ? h + 1 : h;
  const int wm = w > 0 ? w - 1 : 0, wp = w < x.W - 1 ? w + 1 : w;
  const int hs[3] = {hm, h, hp}, ws[3] = {wm, w, wp};
  float v[3][3][8];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int c = 0; c < 3; ++c) ld8<T>((const T*)x.p + vox_off(x, b, t, hs[a], ws[c]) + g * 8, v[a][c]);
  // output 2i   : rows (i-1, i), weights (.25, .75); at i == 0 the source clamps to row 0: weights (1, 0) on rows (0, min(1, n-1))
  // output 2i+1 : rows (i, min(i+1, n-1)), weights (.75, .25)
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      int r0, r1, c0, c1;
      float lh0, lh1, lw0, lw1;
      if (a == 0) { if (h > 0) { r0 = 0; r1 = 1; lh0 = 0.25f; lh1 = 0.75f; } else { r0 = 1; r1 = 2; lh0 = 1.f; lh1 = 0.f; } }
      else { r0 = 1; r1 = 2; lh0 = 0.75f; lh1 = 0.25f; }      // (at the last row both rows are row h: .75 a + .25 a, as above)
      if (c == 0) { if (w > 0) { c0 = 0; c1 = 1; lw0 = 0.25f; lw1 = 0.75f; } else { c0 = 1; c1 = 2; lw0 = 1.f; lw1 = 0.f; } }
      else { c0 = 1; c1 = 2; lw0 = 0.75f; lw1 = 0.25f; }
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        // (runtime-selected rows / columns of the register cache: resolved by selects, the indices are 0/1 or 1/2)
        const float v00 = r0 == 0 ? (c0 == 0 ? v[0][0][e] : v[0][1][e]) : (c0 == 0 ? v[1][0][e] : v[1][1][e]);
        const float v01 = r0 == 0 ? (c1 == 1 ? v[0][1][e] : v[0][2][e]) : (c1 == 1 ? v[1][1][e] : v[1][2][e]);
        const float v10 = r1 == 1 ? (c0 == 0 ? v[1][0][e] : v[1][1][e]) : (c0 == 0 ? v[2][0][e] : v[2][1][e]);
        const float v11 = r1 == 1 ? (c1 == 1 ? v[1][1][e] : v[1][2][e]) : (c1 == 1 ? v[2][1][e] : v[2][2][e]);
        o[e] = lh0 * (lw0 * v00 + lw1 * v01) + lh1 * (lw0 * v10 + lw1 * v11);
      }
      st8<T>((T*)y.p + vox_off(y, b, t, 2 * h + a, 2 * w + c) + g * 8, o);
    }
}

extern "C" int vinet_upsample2x(const VinetTensor* x, const VinetTensor* y, int32_t dtype, void* stream) {
  VN_CHECK_ARG(x && y && quad_ok(*x, esize(dtype)) && quad_ok(*y, esize(dtype)) && x->C == y->C && x->B == y->B &&
                   x->T == y->T && y->H == 2 * x->H && y->W == 2 * x->W, "upsample2x: bad views");
  if (g_vinet_opt_up_blk && oct_ok(*x) && oct_ok(*y)) {
    const long total8 = view_voxels(*x) * (x->C / 8);
    DISPATCH_T(dtype, T, hipLaunchKernelGGL(upsample2x_blk8_kernel<T>, dim3(ew_grid(total8)), dim3(256), 0, (hipStream_t)stream,
                                            make_view(*x), make_view(*y), total8);)
    return vn_launch_status("upsample2x(blk8)");
  }
  const long total = view_voxels(*y) * (y->C / 4);
  DISPATCH_T(dtype, T, hipLaunchKernelGGL(upsample2x_kernel<T>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                                          make_view(*x), make_view(*y), total);)
  return vn_launch_status("upsample2x");
}

// 1-D transpose stencil: input i receives from outputs 2i-1 (.25), 2i (.75 or 1 at i=0),
// 2i+1 (.75 or 1 at i=n-1), 2i+2 (.25)
VN_DEV void up_bwd_taps(int i, int n, int* o, float* wgt) {
  o[0] = 2 * i - 1; wgt[0] = i >= 1 ? 0.25f : 0.f;
  o[1] = 2 * i;     wgt[1] = i == 0 ? 1.f : 0.75f;
  o[2] = 2 * i + 1; wgt[2] = i == n - 1 ? 1.f : 0.75f;
  o[3] = 2 * i + 2; wgt[3] = i <= n - 2 ? 0.25f : 0.f;
}

// MASK: dx = [xf > 0] * (transposed stencil of dy) -- the backward of the ReLU in front of the upsample (xf = its output, the
// upsample's input) folded into the pass that writes the gradient (the decoder: conv -> ReLU -> upsample, model.py:256-258)
template <typename T, bool MASK>
__global__ void upsample2x_bwd_kernel(TView dy, TView dx, TView xf, int accumulate, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint32_t vox_u = fdiv((uint32_t)i, dx.dQ);
  const int q = (int)((uint32_t)i - vox_u * (uint32_t)(dx.C / 4));
  const long vox = (long)vox_u;
  int b, t, h, w;
  decode_vox(dx, vox, b, t, h, w);
  int oh[4], ow[4];
  float wh[4], ww[4];
  up_bwd_taps(h, dx.H, oh, wh);
  up_bwd_taps(w, dx.W, ow, ww);
  float4 g = make_float4(0, 0, 0, 0);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    if (wh[a] == 0.f) continue;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (ww[c] == 0.f) continue;
      const float k = wh[a] * ww[c];
      const float4 d = ldq<T>((const T*)dy.p + vox_off(dy, b, t, oh[a], ow[c]) + q * 4);
      g.x += k * d.x; g.y += k * d.y; g.z += k * d.z; g.w += k * d.w;
    }
  }
  T* dst = (T*)dx.p + vox_off(dx, b, t, h, w) + q * 4;
  if constexpr (MASK) {
    const float4 f = ldq<T>((const T*)xf.p + vox_off(xf, b, t, h, w) + q * 4);
    g.x = f.x > 0.f ? g.x : 0.f; g.y = f.y > 0.f ? g.y : 0.f; g.z = f.z > 0.f ? g.z : 0.f; g.w = f.w > 0.f ? g.w : 0.f;
  }
  if (accumulate) { const float4 o = ldq<T>(dst); g.x += o.x; g.y += o.y; g.z += o.z; g.w += o.w; }
  stq<T>(dst, g);
}

// 8-channel form of the backward gather (16-byte loads, index math per 16 instead of 8 bytes written)
template <typename T, bool MASK>
__global__ __launch_bounds__(256) void upsample2x_bwd8_kernel(TView dy, TView dx, TView xf, int accumulate, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int G = dx.C >> 3;
  const long vox = i / G;
  const int g = (int)(i - vox * G);
  int b, t, h, w;
  decode_vox(dx, vox, b, t, h, w);
  int oh[4], ow[4];
  float wh[4], ww[4];
  up_bwd_taps(h, dx.H, oh, wh);
  up_bwd_taps(w, dx.W, ow, ww);
  float gr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    if (wh[a] == 0.f) continue;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (ww[c] == 0.f) continue;
      const float k = wh[a] * ww[c];
      float d[8];
      ld8<T>((const T*)dy.p + vox_off(dy, b, t, oh[a], ow[c]) + g * 8, d);
#pragma unroll
      for (int e = 0; e < 8; ++e) gr[e] += k * d[e];
    }
  }
  T* dst = (T*)dx.p + vox_off(dx, b, t, h, w) + g * 8;
  if constexpr (MASK) {
    float f[8];
    ld8<T>((const T*)xf.p + vox_off(xf, b, t, h, w) + g * 8, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) gr[e] = f[e] > 0.f ? gr[e] : 0.f;
  }
  if (accumulate) {
    float o[8];
    ld8<T>(dst, o);
#pragma unroll
    for (int e = 0; e < 8; ++e) gr[e] += o[e];
  }
  st8<T>(dst, gr);
}

static int launch_upsample2x_bwd(const VinetTensor* dy, const VinetTensor* dx, const VinetTensor* xf, int32_t dtype, int32_t accumulate,
                                void* stream) {
  VN_CHECK_ARG(dy && dx && quad_ok(*dy, esize(dtype)) && quad_ok(*dx, esize(dtype)) && dx->C == dy->C && dx->B == dy->B &&
                   dx->T == dy->T && dy->H == 2 * dx->H && dy->W == 2 * dx->W, "upsample2x_bwd: bad views");
  VN_CHECK_ARG(!xf || (quad_ok(*xf, esize(dtype)) && same_dims(*xf, *dx)), "upsample2x_bwd: the forward tensor must have dx's extent");
  const long total = view_voxels(*dx) * (dx->C / 4);
  const TView fv = xf ? make_view(*xf) : make_view(*dx);
  if (g_vinet_opt_up_blk && oct_ok(*dx) && oct_ok(*dy) && (!xf || oct_ok(*xf))) {
    const long total8 = view_voxels(*dx) * (dx->C / 8);
    if (xf) { DISPATCH_T(dtype, T, hipLaunchKernelGGL((upsample2x_bwd8_kernel<T, true>), dim3(ew_grid(total8)), dim3(256), 0, (hipStream_t)stream,
                                                      make_view(*dy), make_view(*dx), fv, accumulate, total8);) }
    else { DISPATCH_T(dtype, T, hipLaunchKernelGGL((upsample2x_bwd8_kernel<T, false>), dim3(ew_grid(total8)), dim3(256), 0, (hipStream_t)stream,
                                                   make_view(*dy), make_view(*dx), fv, accumulate, total8);) }
    return vn_launch_status("upsample2x_bwd(8)");
  }
  if (xf) { DISPATCH_T(dtype, T, hipLaunchKernelGGL((upsample2x_bwd_kernel<T, true>), dim3(ew_grid(total)), dim3(256), 0,
                                                    (hipStream_t)stream, make_view(*dy), make_view(*dx), fv, accumulate, total);) }
  else { DISPATCH_T(dtype, T, hipLaunchKernelGGL((upsample2x_bwd_kernel<T, false>), dim3(ew_grid(total)), dim3(256), 0,
                                                 (hipStream_t)stream, make_view(*dy), make_view(*dx), fv, accumulate, total);) }
  return vn_launch_status("upsample2x_bwd");
}

extern "C" int vinet_upsample2x_bwd(const VinetTensor* dy, const VinetTensor* dx, int32_t dtype, int32_t accumulate,
                                    void* stream) {
  return launch_upsample2x_bwd(dy, dx, nullptr, dtype, accumulate, stream);
}

/* dx = [xf > 0] * upsample2x^T(dy): the backward of conv -> ReLU -> upsample (model.py:256-258) behind the conv in ONE pass; xf = the
 * ReLU's output (= the upsample's input), dx's extent.  Stores (the ReLU gates the whole gradient, so nothing may be there yet). */
extern "C" int vinet_upsample2x_bwd_relu(const VinetTensor* dy, const VinetTensor* dx, const VinetTensor* xf, int32_t dtype, void* stream) {
  VN_CHECK_ARG(xf != nullptr, "upsample2x_bwd_relu: forward tensor missing");
  return launch_upsample2x_bwd(dy, dx, xf, dtype, 0, stream);
}
