// HBM-bound kernels of the ViNet path: layout/dtype conversion, weight packing,
// BatchNorm (statistics, fold, backward), activation backward, MaxPool3d,
// 2x bilinear upsample.  All work on channels-last views and move 4 channels
// (8 bytes bf16 / 16 bytes fp32) per lane with channels fastest across lanes,
// so every wave touches whole contiguous rows.
#include "common.h"

extern int g_vinet_opt_pool_twalk;
extern int g_vinet_opt_pool_lds;
extern int g_vinet_opt_pool_pk;
extern int g_vinet_opt_up_blk;
extern int g_vinet_opt_reduce_il;
extern int g_vinet_opt_pool_blk;
extern int g_vinet_opt_pool_pk;

// ---- 4-channel ("quad") typed access -----------------------------------------
template <typename T> VN_DEV float4 ldq(const T* p);
template <> VN_DEV float4 ldq<float>(const float* p) { return *(const float4*)p; }
template <> VN_DEV float4 ldq<bf16_t>(const bf16_t* p) {
  const uint2 q = *(const uint2*)p;
  return make_float4(__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16),
                     __uint_as_float(q.y & 0xffff0000u));
}
template <typename T> VN_DEV void stq(T* p, float4 v);
template <> VN_DEV void stq<float>(float* p, float4 v) { *(float4*)p = v; }
template <> VN_DEV void stq<bf16_t>(bf16_t* p, float4 v) { *(uint2*)p = make_uint2(pack2bf(v.x, v.y), pack2bf(v.z, v.w)); }

// ---- 8-channel typed access (16 bytes bf16 / 32 bytes fp32): the wide form the streaming kernels use
//      whenever the channel count and alignment allow it --------------------------------------------
template <typename T> VN_DEV void ld8(const T* p, float* f);
template <> VN_DEV void ld8<float>(const float* p, float* f) {
  const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
template <> VN_DEV void ld8<bf16_t>(const bf16_t* p, float* f) { unpack16<bf16_t>(*(const uint4*)p, f); }
template <typename T> VN_DEV void st8(T* p, const float* f);
template <> VN_DEV void st8<float>(float* p, const float* f) {
  *(float4*)p = make_float4(f[0], f[1], f[2], f[3]);
  *(float4*)(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
}
template <> VN_DEV void st8<bf16_t>(bf16_t* p, const float* f) { *(uint4*)p = pack16<bf16_t>(f); }
static inline bool oct_ok(const VinetTensor& t) {
  return t.ptr && t.C > 0 && (t.C % 8) == 0 && (t.ld % 8) == 0 && t.ld >= t.C && (t.sB % 8) == 0 && (((uintptr_t)t.ptr) % 16) == 0;
}

VN_DEV float4 affine4(float4 v, const Affine& a, int c) {
  if (a.scale) {
    const float4 s = *(const float4*)(a.scale + c);
    const float4 h = *(const float4*)(a.shift + c);
    v.x = fmaf(v.x, s.x, h.x); v.y = fmaf(v.y, s.y, h.y); v.z = fmaf(v.z, s.z, h.z); v.w = fmaf(v.w, s.w, h.w);
  }
  if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
  return v;
}

VN_DEV void decode_vox(const TView& v, long vox, int& b, int& t, int& h, int& w) {
  decode_m((int)vox, v.dW, v.dH, v.dT, b, t, h, w);
}
// element offset of voxel `vox` (same iteration space as `v`): no decode for linear views
VN_DEV long vox_lin(const TView& v, long vox) {
  if (v.linear) return vox * (long)v.ld;
  int b, t, h, w;
  decode_m((int)vox, v.dW, v.dH, v.dT, b, t, h, w);
  return vox_off(v, b, t, h, w);
}
static inline long view_voxels(const VinetTensor& t) { return (long)t.B * t.T * t.H * t.W; }
static inline bool quad_ok(const VinetTensor& t, int esz) {
  return t.ptr && t.C > 0 && (t.C % 4) == 0 && (t.ld % 4) == 0 && t.ld >= t.C && (t.sB % 4) == 0 &&
         (((uintptr_t)t.ptr) % (4 * esz)) == 0;
}
static inline int esize(int dtype) { return dtype == VINET_F32 ? 4 : 2; }
static inline bool same_dims(const VinetTensor& a, const VinetTensor& b) {
  return a.B == b.B && a.T == b.T && a.H == b.H && a.W == b.W && a.C == b.C;
}
static inline int ew_grid(long n) {   // one thread per item; indices are 32-bit (fast division)
  if (n >= (1L << 31)) { vinet_set_error("elementwise launch too large (%ld items)", n); return 0; }
  long g = (n + 255) / 256; return (int)(g < 1 ? 1 : g);
}

#define DISPATCH_T(dt, T, ...)                         \
  if ((dt) == VINET_F32) { using T = float; __VA_ARGS__ } \
  else { using T = bf16_t; __VA_ARGS__ }

// ============================================================================
// weight packing
// ============================================================================
template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ w, int N, int Cin, int ntaps, int transpose, int stem,
                                    int rows, int Kp, int nslices, T* __restrict__ out) {
  const long total = (long)nslices * rows * Kp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % Kp);
    const int r = (int)((i / Kp) % rows);
    const int s = (int)(i / ((long)Kp * rows));
    float v = 0.f;
    if (stem) {  // out[kh][n][kw*4+c], w[n][c][kh*7+kw]
      const int kw = k >> 2, c = k & 3;
      if (kw < 7 && c < Cin) v = w[((long)r * Cin + c) * ntaps + s * 7 + kw];
    } else if (!transpose) {  // out[t][n][c]
      if (k < Cin) v = w[((long)r * Cin + k) * ntaps + s];
    } else {  // out[t][c][n]
      if (k < N) v = w[((long)k * Cin + r) * ntaps + s];
    }
    store1<T>(out + i, v);
  }
}

extern "C" int vinet_pack_weights(const float* w, int32_t N, int32_t Cin, int32_t ntaps, int32_t transpose,
                                  int32_t stem, int32_t dtype, void* out, void* stream) {
  VN_CHECK_ARG(w && out && N > 0 && Cin > 0 && ntaps > 0, "pack_weights: bad arguments");
  int rows, Kp, nslices;
  if (stem) {
    VN_CHECK_ARG(ntaps == 49 && Cin <= 4 && !transpose, "pack_weights stem: need 1x7x7, Cin<=4");
    rows = N; Kp = 32; nslices = 7;
  } else if (!transpose) { rows = N; Kp = (Cin + 31) / 32 * 32; nslices = ntaps; }
  else { rows = Cin; Kp = (N + 31) / 32 * 32; nslices = ntaps; }
  const long total = (long)nslices * rows * Kp;
  int grid = ew_grid(total); if (grid > 8192) grid = 8192;
  DISPATCH_T(dtype, T, hipLaunchKernelGGL(pack_weights_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, N,
                                          Cin, ntaps, transpose, stem, rows, Kp, nslices, (T*)out);)
  return vn_launch_status("pack_weights");
}

// Multi-tensor form: every weight of the model is re-packed after each optimizer step, 170 launches of a few
// microseconds each when done one by one.  `table` (device memory) holds 8 int64 per job:
//   { w pointer, out pointer, N, Cin, ntaps, transpose | stem << 1, first output index (prefix sum), ld | col << 32 }
// ld != 0 (transposed jobs only): rows of the destination are `ld` elements apart and this job owns columns
// [col, col + N) of them -- several convs that share an input, packed side by side along K for ONE dgrad.
// plus one trailing row whose prefix field is the total; one thread per output element, job by binary search.
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const long* __restrict__ table, int njobs, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int lo = 0, hi = njobs;               // last job with prefix <= i
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (table[mid * 8 + 6] <= i) lo = mid; else hi = mid;
    }
    const long* J = table + lo * 8;
    const float* w = (const float*)J[0];
    T* out = (T*)J[1];
    const int N = (int)J[2], Cin = (int)J[3], ntaps = (int)J[4];
    const int transpose = (int)(J[5] & 1), stem = (int)((J[5] >> 1) & 1);
    const long e = i - J[6];
    const int rows = (stem || !transpose) ? N : Cin;
    const int Kp = stem ? 32 : ((transpose ? N : Cin) + 31) / 32 * 32;
    const int k = (int)(e % Kp);
    const int r = (int)((e / Kp) % rows);
    const int sl = (int)(e / ((long)Kp * rows));
    float v = 0.f;
    if (stem) {
      const int kw = k >> 2, c = k & 3;
      if (kw < 7 && c < Cin) v = w[((long)r * Cin + c) * ntaps + sl * 7 + kw];
    } else if (!transpose) {
      if (k < Cin) v = w[((long)r * Cin + k) * ntaps + sl];
    } else {
      if (k < N) v = w[((long)k * Cin + r) * ntaps + sl];
      const long ld = J[7] & 0xffffffffl;
      if (ld) {          // side-by-side destination: only the job's own columns are written
        if (k < N) store1<T>(out + ((long)sl * rows + r) * ld + (J[7] >> 32) + k, v);
        continue;
      }
    }
    store1<T>(out + e, v);
  }
}

extern "C" int vinet_pack_weights_multi(const int64_t* table, int32_t njobs, int64_t total, int32_t dtype, void* stream) {
  VN_CHECK_ARG(table && njobs > 0 && total > 0, "pack_weights_multi: bad arguments");
  int grid = ew_grid(total); if (grid > 16384) grid = 16384;
  DISPATCH_T(dtype, T, hipLaunchKernelGGL(pack_weights_multi_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                          (const long*)table, njobs, (long)total);)
  return vn_launch_status("pack_weights_multi");
}

__global__ void unpack_wgrad_kernel(float* __restrict__ dw, int N, int Cin, int ntaps, int stem, int Kp,
                                    int flags, float* __restrict__ grad) {
  const int accumulate = flags & 1, clear = flags & 2;
  const long total = (long)N * Cin * ntaps;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i % ntaps);
    const int c = (int)((i / ntaps) % Cin);
    const int n = (int)(i / ((long)ntaps * Cin));
    long src;
    if (stem) { const int kh = t / 7, kw = t % 7; src = ((long)kh * N + n) * 32 + kw * 4 + c; }
    else src = ((long)t * N + n) * Kp + c;
    const float v = dw[src];
    grad[i] = accumulate ? grad[i] + v : v;
    if (clear) dw[src] = 0.f;      // every packed element is read by exactly one thread: hand the buffer back zeroed
  }
  if (clear) {
    // columns no torch element maps to (channel padding; split-K atomics may have touched them)
    const int nsl = stem ? 7 : ntaps, kp = stem ? 32 : Kp;
    const long ptotal = (long)nsl * N * kp;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < ptotal; i += (long)gridDim.x * blockDim.x) {
      const int col = (int)(i % kp);
      const bool valid = stem ? (col < 28 && (col & 3) < Cin) : (col < Cin);
      if (!valid) dw[i] = 0.f;
    }
  }
}

extern "C" int vinet_unpack_wgrad(float* dw, int32_t N, int32_t Cin, int32_t ntaps, int32_t stem,
                                  int32_t flags, float* grad, void* stream) {
  VN_CHECK_ARG(dw && grad && N > 0 && Cin > 0 && ntaps > 0, "unpack_wgrad: bad arguments");
  const int Kp = stem ? 32 : (Cin + 31) / 32 * 32;
  const long total = (long)N * Cin * ntaps;
  int grid = ew_grid(total); if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dw, N, Cin, ntaps, stem, Kp,
                     flags, grad);
  return vn_launch_status("unpack_wgrad");
}

// ============================================================================
// NCDHW <-> channels-last
// ============================================================================
template <typename T>
__global__ void import_ncdhw_kernel(const float* __restrict__ src, long sb, long sc, long st, long sh, long sw, int C,
                                    TView dst, long nvox) {
  const long vox = (long)blockIdx.x * blockDim.x + threadIdx.x;   // voxels fastest: coalesced planar reads
  if (vox >= nvox) return;
  const int q = blockIdx.y;
  int b, t, h, w;
  decode_vox(dst, vox, b, t, h, w);
  const float* s = src + b * sb + t * st + h * sh + w * sw;
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { const int c = q * 4 + e; v[e] = c < C ? s[c * sc] : 0.f; }
  stq<T>((T*)dst.p + vox_off(dst, b, t, h, w) + q * 4, make_float4(v[0], v[1], v[2], v[3]));
}

extern "C" int vinet_import_ncdhw(const float* src, int64_t sb, int64_t sc, int64_t st, int64_t sh, int64_t sw,
                                  int32_t C, const VinetTensor* dst, int32_t dst_dtype, void* stream) {
  VN_CHECK_ARG(src && dst && quad_ok(*dst, esize(dst_dtype)) && C > 0 && C <= dst->C, "import_ncdhw: bad arguments");
  const long nvox = view_voxels(*dst);
  DISPATCH_T(dst_dtype, T, hipLaunchKernelGGL(import_ncdhw_kernel<T>, dim3(ew_grid(nvox), dst->C / 4), dim3(256), 0,
                                              (hipStream_t)stream, src, sb, sc, st, sh, sw, C, make_view(*dst), nvox);)
  return vn_launch_status("import_ncdhw");
}

// import into a zero-padded buffer: dst voxel (h, w) <- src(h - pad_top, w - pad_left), zero outside
template <typename T>
__global__ void import_pad_kernel(const float* __restrict__ src, long sb, long sc, long st, long sh, long sw, int C,
                                  int Hs, int Ws, int pad_top, int pad_left, TView dst, long nvox) {
  const long vox = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (vox >= nvox) return;
  const int q = blockIdx.y;
  int b, t, h, w;
  decode_vox(dst, vox, b, t, h, w);
  const int hs = h - pad_top, ws = w - pad_left;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if ((unsigned)hs < (unsigned)Hs && (unsigned)ws < (unsigned)Ws) {
    const float* s = src + b * sb + t * st + hs * sh + ws * sw;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const int c = q * 4 + e; if (c < C) v[e] = s[c * sc]; }
  }
  stq<T>((T*)dst.p + vox_off(dst, b, t, h, w) + q * 4, make_float4(v[0], v[1], v[2], v[3]));
}

extern "C" int vinet_import_ncdhw_pad(const float* src, int64_t sb, int64_t sc, int64_t st, int64_t sh, int64_t sw,
                                      int32_t C, int32_t Hs, int32_t Ws, int32_t pad_top, int32_t pad_left,
                                      const VinetTensor* dst, int32_t dst_dtype, void* stream) {
  VN_CHECK_ARG(src && dst && quad_ok(*dst, esize(dst_dtype)) && C > 0 && C <= dst->C && Hs > 0 && Ws > 0 &&
                   pad_top >= 0 && pad_left >= 0 && pad_top + Hs <= dst->H && pad_left + Ws <= dst->W,
               "import_ncdhw_pad: bad arguments");
  const long nvox = view_voxels(*dst);
  DISPATCH_T(dst_dtype, T, hipLaunchKernelGGL(import_pad_kernel<T>, dim3(ew_grid(nvox), dst->C / 4), dim3(256), 0,
                                              (hipStream_t)stream, src, sb, sc, st, sh, sw, C, Hs, Ws, pad_top, pad_left,
                                              make_view(*dst), nvox);)
  return vn_launch_status("import_ncdhw_pad");
}

template <typename T>
__global__ void export_ncdhw_kernel(TView src, Affine pre, float* __restrict__ dst, long sb, long sc, long st, long sh,
                                    long sw, int accumulate, long nvox) {
  const long vox = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (vox >= nvox) return;
  const int q = blockIdx.y;
  int b, t, h, w;
  decode_vox(src, vox, b, t, h, w);
  float4 v = ldq<T>((const T*)src.p + vox_off(src, b, t, h, w) + q * 4);
  v = affine4(v, pre, q * 4);
  float* d = dst + b * sb + t * st + h * sh + w * sw + (long)(q * 4) * sc;
  const float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) d[e * sc] = accumulate ? d[e * sc] + o[e] : o[e];
}

extern "C" int vinet_export_ncdhw(const VinetTensor* src, int32_t src_dtype, VinetAffine pre, float* dst, int64_t sb,
                                  int64_t sc, int64_t st, int64_t sh, int64_t sw, int32_t accumulate, void* stream) {
  VN_CHECK_ARG(src && dst && quad_ok(*src, esize(src_dtype)), "export_ncdhw: bad arguments");
  const long nvox = view_voxels(*src);
  DISPATCH_T(src_dtype, T, hipLaunchKernelGGL(export_ncdhw_kernel<T>, dim3(ew_grid(nvox), src->C / 4), dim3(256), 0,
                                              (hipStream_t)stream, make_view(*src), make_affine(pre), dst, sb, sc, st,
                                              sh, sw, accumulate, nvox);)
  return vn_launch_status("export_ncdhw");
}

template <typename TI, typename TO>
__global__ void copy_affine_kernel(TView src, Affine pre, TView dst, int accumulate, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint32_t vox_u = fdiv((uint32_t)i, src.dQ);
  const int q = (int)((uint32_t)i - vox_u * (uint32_t)(src.C / 4));
  const long vox = (long)vox_u;
  float4 v = ldq<TI>((const TI*)src.p + vox_lin(src, vox) + q * 4);
  v = affine4(v, pre, q * 4);
  TO* d = (TO*)dst.p + vox_lin(dst, vox) + q * 4;
  if (accumulate) { const float4 o = ldq<TO>(d); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
  stq<TO>(d, v);
}

// 8-channel form (structure of bn_bwd_apply8_kernel): a lane keeps the scale / shift of its 8 channels in registers
// and streams voxels, 16-byte loads and stores, 4 voxels in flight -- the quad kernel above pays a voxel decode and
// two coefficient loads per 8 bytes.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void copy_affine8_kernel(TView src, Affine pre, TView dst, int accumulate, long nvox, long vb) {
  const int G = src.C / 8;
  const int Gb = G < 256 ? G : 256;
  const int R = 256 / Gb;
  const int r = threadIdx.x / Gb;
  const int g0 = threadIdx.x % Gb;
  if (r >= R) return;
  const long v0 = (long)blockIdx.x * vb;
  long v1 = v0 + vb; if (v1 > nvox) v1 = nvox;
  for (int g = g0; g < G; g += Gb) {
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = pre.scale ? pre.scale[g * 8 + e] : 1.f; sh[e] = pre.scale ? pre.shift[g * 8 + e] : 0.f; }
    constexpr int U = 4;
    for (long vq = v0 + r; vq < v1; vq += (long)R * U) {
      float xv[U][8], ov[U][8];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long v = vq + (long)u * R;
        ok[u] = v < v1;
        if (ok[u]) {
          ld8<TI>((const TI*)src.p + vox_lin(src, v) + g * 8, xv[u]);
          if (accumulate) ld8<TO>((const TO*)dst.p + vox_lin(dst, v) + g * 8, ov[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!ok[u]) continue;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float t = pre.scale ? fmaf(xv[u][e], sc[e], sh[e]) : xv[u][e];
          if (pre.relu) t = fmaxf(t, 0.f);
          o[e] = accumulate ? t + ov[u][e] : t;
        }
        st8<TO>((TO*)dst.p + vox_lin(dst, vq + (long)u * R) + g * 8, o);
      }
    }
  }
}

extern "C" int vinet_copy_affine(const VinetTensor* src, int32_t src_dtype, VinetAffine pre, const VinetTensor* dst,
                                 int32_t dst_dtype, int32_t accumulate, void* stream) {
  VN_CHECK_ARG(src && dst && quad_ok(*src, esize(src_dtype)) && quad_ok(*dst, esize(dst_dtype)) && same_dims(*src, *dst),
               "copy_affine: bad views");
  const long total = view_voxels(*src) * (src->C / 4);
  hipStream_t s = (hipStream_t)stream;
  const TView sv = make_view(*src), dv = make_view(*dst);
  const Affine a = make_affine(pre);
  if (src_dtype == VINET_BF16 && dst_dtype == VINET_BF16 && oct_ok(*src) && oct_ok(*dst) && view_voxels(*src) >= 65536) {
    const long nvox = view_voxels(*src);
    const int G = src->C / 8, R = 256 / (G < 256 ? G : 256);
    long vb = R * 16;
    while ((nvox + vb - 1) / vb > 16384) vb *= 2;
    hipLaunchKernelGGL((copy_affine8_kernel<bf16_t, bf16_t>), dim3((unsigned)((nvox + vb - 1) / vb)), dim3(256), 0, s, sv, a, dv, accumulate, nvox, vb);
    return vn_launch_status("copy_affine8");
  }
  const dim3 g(ew_grid(total)), blk(256);
  if (src_dtype == VINET_F32 && dst_dtype == VINET_F32) hipLaunchKernelGGL((copy_affine_kernel<float, float>), g, blk, 0, s, sv, a, dv, accumulate, total);
  else if (src_dtype == VINET_F32) hipLaunchKernelGGL((copy_affine_kernel<float, bf16_t>), g, blk, 0, s, sv, a, dv, accumulate, total);
  else if (dst_dtype == VINET_F32) hipLaunchKernelGGL((copy_affine_kernel<bf16_t, float>), g, blk, 0, s, sv, a, dv, accumulate, total);
  else hipLaunchKernelGGL((copy_affine_kernel<bf16_t, bf16_t>), g, blk, 0, s, sv, a, dv, accumulate, total);
  return vn_launch_status("copy_affine");
}

// ============================================================================
// BatchNorm
// ============================================================================
__global__ void bn_finalize_kernel(const float* __restrict__ partials, int rows, int C, int ld, double count,
                                   const float* gamma, const float* beta, float eps, float momentum,
                                   float* running_mean, float* running_var, float* mean_o, float* invstd_o,
                                   float* scale_o, float* shift_o) {
  const int c = blockIdx.x;
  double s = 0.0, q = 0.0;
  for (int r = threadIdx.x; r < rows; r += blockDim.x) {
    s += (double)partials[((long)r * 2 + 0) * ld + c];
    q += (double)partials[((long)r * 2 + 1) * ld + c];
  }
  __shared__ double red[2][4];
  s = wave_sum_d(s); q = wave_sum_d(q);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    q = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = (float)(g * invstd);
    if (mean_o) mean_o[c] = (float)mean;
    if (invstd_o) invstd_o[c] = (float)invstd;
    scale_o[c] = sc;
    shift_o[c] = b - (float)mean * sc;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
  }
}

extern "C" int vinet_bn_finalize(const float* partials, int32_t rows, int32_t C, int32_t ld, double count, const float* gamma,
                                 const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                 float* mean, float* invstd, float* scale, float* shift, void* stream) {
  VN_CHECK_ARG(partials && rows > 0 && C > 0 && (ld == 0 || ld >= C) && count > 0 && scale && shift, "bn_finalize: bad arguments");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, partials, rows, C, ld ? ld : C, count, gamma,
                     beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift);
  return vn_launch_status("bn_finalize");
}

// Pre-reduction of a tall partials table (the 64-channel stem layers produce 172 032 rows at 64 clips: the
// one-workgroup-per-channel finalize would walk them with 4-byte reads 2*C*4 bytes apart).  Block (chunk, 64-channel
// group): 64 channels x 4 row lanes, whole-row coalesced reads, double accumulation, out[chunk][2][C].
__global__ __launch_bounds__(256) void bn_partials_fold_kernel(const float* __restrict__ partials, int rows, int C, int per,
                                                               float* __restrict__ out) {
  const int c = blockIdx.y * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int r0 = blockIdx.x * per;
  int r1 = r0 + per; if (r1 > rows) r1 = rows;
  double s = 0.0, q = 0.0;
  if (c < C)
    for (int r = r0 + rl; r < r1; r += 4) {
      s += (double)partials[((long)r * 2 + 0) * C + c];
      q += (double)partials[((long)r * 2 + 1) * C + c];
    }
  __shared__ double red[2][4][64];
  red[0][rl][threadIdx.x & 63] = s; red[1][rl][threadIdx.x & 63] = q;
  __syncthreads();
  if (rl == 0 && c < C) {
    const int l = threadIdx.x;
    out[((long)blockIdx.x * 2 + 0) * C + c] = (float)(red[0][0][l] + red[0][1][l] + red[0][2][l] + red[0][3][l]);
    out[((long)blockIdx.x * 2 + 1) * C + c] = (float)(red[1][0][l] + red[1][1][l] + red[1][2][l] + red[1][3][l]);
  }
}

extern "C" int vinet_bn_partials_fold(const float* partials, int32_t rows, int32_t C, float* out, int32_t out_rows, void* stream) {
  VN_CHECK_ARG(partials && out && rows > 0 && C > 0 && out_rows > 0 && out_rows <= rows, "bn_partials_fold: bad arguments");
  const int per = (rows + out_rows - 1) / out_rows;
  VN_CHECK_ARG((long)(out_rows - 1) * per < rows, "bn_partials_fold: out_rows=%d leaves empty chunks for rows=%d", out_rows, rows);
  hipLaunchKernelGGL(bn_partials_fold_kernel, dim3(out_rows, (C + 63) / 64), dim3(256), 0, (hipStream_t)stream, partials, rows, C, per, out);
  return vn_launch_status("bn_partials_fold");
}

__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                               const float* conv_bias, float eps, int C, float* scale, float* shift, float* invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float is = 1.f / sqrtf(rv[c] + eps);
  const float sc = (gamma ? gamma[c] : 1.f) * is;
  scale[c] = sc;
  shift[c] = (beta ? beta[c] : 0.f) + ((conv_bias ? conv_bias[c] : 0.f) - rm[c]) * sc;
  if (invstd) invstd[c] = is;
}

extern "C" int vinet_bn_fold(const float* gamma, const float* beta, const float* running_mean,
                             const float* running_var, const float* conv_bias, float eps, int32_t C, float* scale,
                             float* shift, float* invstd, void* stream) {
  VN_CHECK_ARG(running_mean && running_var && scale && shift && C > 0, "bn_fold: bad arguments");
  hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                     running_mean, running_var, conv_bias, eps, C, scale, shift, invstd);
  return vn_launch_status("bn_fold");
}

// Per-channel reductions over voxels.  Thread (q, r): channel quad q, voxel lane
// r; a block covers `vb` consecutive voxels and writes one partial row.
// MODE 0: (sum x, sum x^2) of x;  MODE 1: (sum dz*mask, sum dz*mask*xhat).

// 8-channel form of channel_reduce_kernel: same partials contract ([rows][2][C], block b owns voxels
// [b*vb, (b+1)*vb)), twice the bytes per load instruction.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void channel_reduce8_kernel(TView x, TView dz, Affine fwd, const float* mean,
                                                              const float* invstd, long nvox, long vb,
                                                              float* __restrict__ partials) {
  const int G = x.C / 8;
  const int Gb = G < 256 ? G : 256;
  const int R = 256 / Gb;
  const int r = threadIdx.x / Gb;
  const int g0 = threadIdx.x % Gb;
  __shared__ float red[256 * 16];
  constexpr int U = 4;
  // vb == 0: interleaved rounds -- in round i block b reads voxels [(i*gridDim.x + b)*R*U, +R*U), so the whole grid
  // walks one contiguous window of the tensor instead of gridDim.x streams a fixed stride apart
  const long v0 = vb ? (long)blockIdx.x * vb : (long)blockIdx.x * R * U;
  long v1 = vb ? v0 + vb : nvox; if (v1 > nvox) v1 = nvox;
  const long vstep = vb ? (long)R * U : (long)gridDim.x * R * U;
  for (int g = g0; g < G; g += Gb) {
    float s[8], p[8], mu[8], is[8], sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; p[e] = 0.f; mu[e] = 0.f; is[e] = 1.f; sc[e] = 1.f; sh[e] = 0.f; }
    if (r < R) {
      if (MODE == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { mu[e] = mean[g * 8 + e]; is[e] = invstd[g * 8 + e]; }
        if (fwd.relu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { sc[e] = fwd.scale ? fwd.scale[g * 8 + e] : 1.f; sh[e] = fwd.shift ? fwd.shift[g * 8 + e] : 0.f; }
        }
      }
      for (long vq = v0 + r; vq < v1; vq += vstep) {
        float xv[U][8], gv[U][8];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const long v = vq + (long)u * R;
          ok[u] = v < v1;
          if (ok[u]) {
            ld8<T>((const T*)x.p + vox_lin(x, v) + g * 8, xv[u]);
            if (MODE == 1) ld8<T>((const T*)dz.p + vox_lin(dz, v) + g * 8, gv[u]);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (!ok[u]) continue;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            if (MODE == 0) {
              s[e] += xv[u][e]; p[e] += xv[u][e] * xv[u][e];
            } else {
              float gg = gv[u][e];
              if (fwd.relu && !(fmaf(xv[u][e], sc[e], sh[e]) > 0.f)) gg = 0.f;
              s[e] += gg;
              p[e] += gg * (xv[u][e] - mu[e]) * is[e];
            }
          }
        }
      }
    }
    __syncthreads();
    if (r < R) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { red[threadIdx.x * 16 + e] = s[e]; red[threadIdx.x * 16 + 8 + e] = p[e]; }
    }
    __syncthreads();
    if (r == 0) {
      for (int rr = 1; rr < R; ++rr)
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] += red[(rr * Gb + g0) * 16 + e]; p[e] += red[(rr * Gb + g0) * 16 + 8 + e]; }
      float* o = partials + (long)blockIdx.x * 2 * x.C + g * 8;
      *(float4*)o = make_float4(s[0], s[1], s[2], s[3]);
      *(float4*)(o + 4) = make_float4(s[4], s[5], s[6], s[7]);
      *(float4*)(o + x.C) = make_float4(p[0], p[1], p[2], p[3]);
      *(float4*)(o + x.C + 4) = make_float4(p[4], p[5], p[6], p[7]);
    }
  }
}
static inline int stats_rows_for(long nvox) {
  long rows = (nvox + 63) / 64;
  if (rows > 1024) rows = 1024;
  if (rows < 1) rows = 1;
  return (int)rows;
}
extern "C" int vinet_stats_rows(const VinetTensor* x) { return x ? stats_rows_for(view_voxels(*x)) : -1; }

template <typename T, int MODE>
__global__ __launch_bounds__(256) void channel_reduce_kernel(TView x, TView dz, Affine fwd, const float* mean,
                                                             const float* invstd, long nvox, long vb,
                                                             float* __restrict__ partials) {
  const int Q = x.C / 4;
  const int Qb = Q < 256 ? Q : 256;
  const int R = 256 / Qb;
  const int r = threadIdx.x / Qb;
  const int q0 = threadIdx.x % Qb;
  __shared__ float red[256 * 8];
  const long v0 = (long)blockIdx.x * vb;
  long v1 = v0 + vb; if (v1 > nvox) v1 = nvox;
  for (int q = q0; q < Q; q += Qb) {
    float s[4] = {0, 0, 0, 0}, p[4] = {0, 0, 0, 0};
    if (r < R) {
      float4 mu = make_float4(0, 0, 0, 0), is = make_float4(1, 1, 1, 1);
      if (MODE == 1) { mu = *(const float4*)(mean + q * 4); is = *(const float4*)(invstd + q * 4); }
      constexpr int U = 4;   // independent voxels per iteration: 2*U loads in flight per lane
      for (long vb = v0 + r; vb < v1; vb += (long)R * U) {
        float4 xv[U], gv[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const long v = vb + (long)u * R;
          ok[u] = v < v1;
          xv[u] = make_float4(0, 0, 0, 0); gv[u] = make_float4(0, 0, 0, 0);
          if (ok[u]) {
            xv[u] = ldq<T>((const T*)x.p + vox_lin(x, v) + q * 4);
            if (MODE == 1) gv[u] = ldq<T>((const T*)dz.p + vox_lin(dz, v) + q * 4);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (!ok[u]) continue;
          if (MODE == 0) {
            s[0] += xv[u].x; s[1] += xv[u].y; s[2] += xv[u].z; s[3] += xv[u].w;
            p[0] += xv[u].x * xv[u].x; p[1] += xv[u].y * xv[u].y; p[2] += xv[u].z * xv[u].z; p[3] += xv[u].w * xv[u].w;
          } else {
            float4 g = gv[u];
            if (fwd.relu) {
              Affine na = fwd; na.relu = 0;
              const float4 z = affine4(xv[u], na, q * 4);
              if (!(z.x > 0.f)) g.x = 0.f;
              if (!(z.y > 0.f)) g.y = 0.f;
              if (!(z.z > 0.f)) g.z = 0.f;
              if (!(z.w > 0.f)) g.w = 0.f;
            }
            s[0] += g.x; s[1] += g.y; s[2] += g.z; s[3] += g.w;
            p[0] += g.x * (xv[u].x - mu.x) * is.x; p[1] += g.y * (xv[u].y - mu.y) * is.y;
            p[2] += g.z * (xv[u].z - mu.z) * is.z; p[3] += g.w * (xv[u].w - mu.w) * is.w;
          }
        }
      }
    }
    __syncthreads();
    if (r < R) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { red[threadIdx.x * 8 + e] = s[e]; red[threadIdx.x * 8 + 4 + e] = p[e]; }
    }
    __syncthreads();
    if (r == 0) {
      for (int rr = 1; rr < R; ++rr)
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[e] += red[(rr * Qb + q0) * 8 + e]; p[e] += red[(rr * Qb + q0) * 8 + 4 + e]; }
      float* o = partials + (long)blockIdx.x * 2 * x.C + q * 4;
      *(float4*)o = make_float4(s[0], s[1], s[2], s[3]);
      *(float4*)(o + x.C) = make_float4(p[0], p[1], p[2], p[3]);
    }
  }
}

template <int MODE>
static int launch_channel_reduce(const VinetTensor* x, const VinetTensor* dz, int dtype, VinetAffine fwd,
                                 const float* mean, const float* invstd, float* partials, void* stream) {
  const long nvox = view_voxels(*x);
  const int rows = stats_rows_for(nvox);
  const long vb = (nvox + rows - 1) / rows;
  const TView xv = make_view(*x), dv = dz ? make_view(*dz) : xv;
  if (oct_ok(*x) && (!dz || oct_ok(*dz))) {
    DISPATCH_T(dtype, T, hipLaunchKernelGGL((channel_reduce8_kernel<T, MODE>), dim3(rows), dim3(256), 0,
                                            (hipStream_t)stream, xv, dv, make_affine(fwd), mean, invstd, nvox, g_vinet_opt_reduce_il ? 0 : vb, partials);)
    return vn_launch_status("channel_reduce8");
  }
  DISPATCH_T(dtype, T, hipLaunchKernelGGL((channel_reduce_kernel<T, MODE>), dim3(rows), dim3(256), 0,
                                          (hipStream_t)stream, xv, dv, make_affine(fwd), mean, invstd, nvox, vb, partials);)
  return vn_launch_status("channel_reduce");
}

extern "C" int vinet_channel_stats(const VinetTensor* x, int32_t dtype, float* partials, void* stream) {
  VN_CHECK_ARG(x && partials && quad_ok(*x, esize(dtype)), "channel_stats: bad arguments");
  VinetAffine none = {nullptr, nullptr, 0};
  return launch_channel_reduce<0>(x, nullptr, dtype, none, nullptr, nullptr, partials, stream);
}

extern "C" int vinet_bn_bwd_reduce(const VinetTensor* dz, const VinetTensor* x_raw, int32_t dtype, VinetAffine fwd,
                                   const float* mean, const float* invstd, float* partials, void* stream) {
  VN_CHECK_ARG(dz && x_raw && partials && mean && invstd && quad_ok(*dz, esize(dtype)) && quad_ok(*x_raw, esize(dtype)) &&
                   same_dims(*dz, *x_raw), "bn_bwd_reduce: bad arguments");
  return launch_channel_reduce<1>(x_raw, dz, dtype, fwd, mean, invstd, partials, stream);
}

// one workgroup per channel: rows are reduced in parallel (fp64), lane 0 finishes
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partials, int rows, int C, int ld,
                                                              double count, const float* scale, int train, float* dgamma,
                                                              float* dbeta, const float* invstd, float* c1, float* c2) {
  const int c = blockIdx.x;
  double s = 0.0, p = 0.0;
  for (int r = threadIdx.x; r < rows; r += blockDim.x) {
    s += (double)partials[((long)r * 2) * ld + c];
    p += (double)partials[((long)r * 2 + 1) * ld + c];
  }
  __shared__ double red[2][4];
  s = wave_sum_d(s); p = wave_sum_d(p);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = p; }
  __syncthreads();
  if (threadIdx.x == 0) {
    s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    p = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    if (dgamma) dgamma[c] += (float)p;
    if (dbeta) dbeta[c] += (float)s;
    if (c1) c1[c] = train ? (float)(s / count) : 0.f;
    if (c2) c2[c] = train ? (float)(p / count) : 0.f;
  }
}

extern "C" int vinet_bn_bwd_finalize(const float* partials, int32_t rows, int32_t C, int32_t ld, double count, const float* scale,
                                     int32_t train, float* dgamma_acc, float* dbeta_acc, const float* invstd, float* c1,
                                     float* c2, void* stream) {
  VN_CHECK_ARG(partials && rows > 0 && C > 0 && (ld == 0 || ld >= C) && count > 0, "bn_bwd_finalize: bad arguments");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, partials, rows, C, ld ? ld : C, count,
                     scale, train, dgamma_acc, dbeta_acc, invstd, c1, c2);
  return vn_launch_status("bn_bwd_finalize");
}

template <typename T>
__global__ void bn_bwd_apply_kernel(TView dz, TView x, Affine fwd, const float* mean, const float* invstd,
                                    const float* c1, const float* c2, TView dx, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint32_t vox_u = fdiv((uint32_t)i, x.dQ);
  const int q = (int)((uint32_t)i - vox_u * (uint32_t)(x.C / 4));
  const long vox = (long)vox_u;
  const float4 xv = ldq<T>((const T*)x.p + vox_lin(x, vox) + q * 4);
  float4 g = ldq<T>((const T*)dz.p + vox_lin(dz, vox) + q * 4);
  const float4 sc = *(const float4*)(fwd.scale + q * 4);
  if (fwd.relu) {
    const float4 sh = *(const float4*)(fwd.shift + q * 4);
    if (!(fmaf(xv.x, sc.x, sh.x) > 0.f)) g.x = 0.f;
    if (!(fmaf(xv.y, sc.y, sh.y) > 0.f)) g.y = 0.f;
    if (!(fmaf(xv.z, sc.z, sh.z) > 0.f)) g.z = 0.f;
    if (!(fmaf(xv.w, sc.w, sh.w) > 0.f)) g.w = 0.f;
  }
  const float4 mu = *(const float4*)(mean + q * 4), is = *(const float4*)(invstd + q * 4);
  const float4 a1 = *(const float4*)(c1 + q * 4), a2 = *(const float4*)(c2 + q * 4);
  float4 o;
  o.x = sc.x * (g.x - a1.x - (xv.x - mu.x) * is.x * a2.x);
  o.y = sc.y * (g.y - a1.y - (xv.y - mu.y) * is.y * a2.y);
  o.z = sc.z * (g.z - a1.z - (xv.z - mu.z) * is.z * a2.z);
  o.w = sc.w * (g.w - a1.w - (xv.w - mu.w) * is.w * a2.w);
  stq<T>((T*)dx.p + vox_lin(dx, vox) + q * 4, o);
}

// 8-channel, voxel-looping form: the six per-channel parameter vectors are folded into four
// coefficients held in registers (dx = A*g + B*x + D with the ReLU gate from sc*x + sh), so a
// lane issues two 16-byte loads and one 16-byte store per voxel and nothing else.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply8_kernel(TView dz, TView x, Affine fwd, const float* mean, const float* invstd,
                                                            const float* c1, const float* c2, TView dx, long nvox, long vb) {
  const int G = x.C / 8;
  const int Gb = G < 256 ? G : 256;
  const int R = 256 / Gb;
  const int r = threadIdx.x / Gb;
  const int g0 = threadIdx.x % Gb;
  if (r >= R) return;
  const long v0 = (long)blockIdx.x * vb;
  long v1 = v0 + vb; if (v1 > nvox) v1 = nvox;
  for (int g = g0; g < G; g += Gb) {
    float A[8], Bc[8], D[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = g * 8 + e;
      const float sc = fwd.scale[c], k = sc * invstd[c] * c2[c];
      A[e] = sc; Bc[e] = -k; D[e] = fmaf(k, mean[c], -sc * c1[c]);
      sh[e] = fwd.shift[c];
    }
    constexpr int U = 4;
    for (long vq = v0 + r; vq < v1; vq += (long)R * U) {
      float xv[U][8], gv[U][8];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long v = vq + (long)u * R;
        ok[u] = v < v1;
        if (ok[u]) {
          ld8<T>((const T*)x.p + vox_lin(x, v) + g * 8, xv[u]);
          ld8<T>((const T*)dz.p + vox_lin(dz, v) + g * 8, gv[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!ok[u]) continue;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float gg = gv[u][e];
          if (fwd.relu && !(fmaf(xv[u][e], A[e], sh[e]) > 0.f)) gg = 0.f;
          o[e] = fmaf(A[e], gg, fmaf(Bc[e], xv[u][e], D[e]));
        }
        st8<T>((T*)dx.p + vox_lin(dx, vq + (long)u * R) + g * 8, o);
      }
    }
  }
}

extern "C" int vinet_bn_bwd_apply(const VinetTensor* dz, const VinetTensor* x_raw, int32_t dtype, VinetAffine fwd,
                                  const float* mean, const float* invstd, const float* c1, const float* c2,
                                  const VinetTensor* dx, void* stream) {
  VN_CHECK_ARG(dz && x_raw && dx && fwd.scale && fwd.shift && mean && invstd && c1 && c2, "bn_bwd_apply: null argument");
  VN_CHECK_ARG(quad_ok(*dz, esize(dtype)) && quad_ok(*x_raw, esize(dtype)) && quad_ok(*dx, esize(dtype)) &&
                   same_dims(*dz, *x_raw) && same_dims(*dz, *dx), "bn_bwd_apply: bad views");
  if (oct_ok(*dz) && oct_ok(*x_raw) && oct_ok(*dx)) {
    const long nvox = view_voxels(*dz);
    const int G = dz->C / 8, R = 256 / (G < 256 ? G : 256);
    long vb = R * 16;                                  // 4 rounds of 4 voxels per lane
    while ((nvox + vb - 1) / vb > 16384) vb *= 2;
    DISPATCH_T(dtype, T, hipLaunchKernelGGL(bn_bwd_apply8_kernel<T>, dim3((unsigned)((nvox + vb - 1) / vb)), dim3(256), 0,
                                            (hipStream_t)stream, make_view(*dz), make_view(*x_raw), make_affine(fwd), mean,
                                            invstd, c1, c2, make_view(*dx), nvox, vb);)
    return vn_launch_status("bn_bwd_apply8");
  }
  const long total = view_voxels(*dz) * (dz->C / 4);
  DISPATCH_T(dtype, T, hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3(ew_grid(total)), dim3(256), 0,
                                          (hipStream_t)stream, make_view(*dz), make_view(*x_raw), make_affine(fwd), mean,
                                          invstd, c1, c2, make_view(*dx), total);)
  return vn_launch_status("bn_bwd_apply");
}

template <typename TG, typename TZ, typename TO>
__global__ void act_bwd_kernel(TView dz, TView z, int act, TView dy, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint32_t vox_u = fdiv((uint32_t)i, z.dQ);
  const int q = (int)((uint32_t)i - vox_u * (uint32_t)(z.C / 4));
  const long vox = (long)vox_u;
  const float4 zv = ldq<TZ>((const TZ*)z.p + vox_lin(z, vox) + q * 4);
  float4 g = ldq<TG>((const TG*)dz.p + vox_lin(dz, vox) + q * 4);
  if (act == VINET_ACT_RELU) {
    if (!(zv.x > 0.f)) g.x = 0.f;
    if (!(zv.y > 0.f)) g.y = 0.f;
    if (!(zv.z > 0.f)) g.z = 0.f;
    if (!(zv.w > 0.f)) g.w = 0.f;
  } else if (act == VINET_ACT_SIGMOID) {
    g.x *= zv.x * (1.f - zv.x); g.y *= zv.y * (1.f - zv.y); g.z *= zv.z * (1.f - zv.z); g.w *= zv.w * (1.f - zv.w);
  }
  stq<TO>((TO*)dy.p + vox_lin(dy, vox) + q * 4, g);
}

extern "C" int vinet_act_bwd(const VinetTensor* dz, int32_t dz_dtype, const VinetTensor* z, int32_t z_dtype,
                             int32_t act, const VinetTensor* dy, int32_t dy_dtype, void* stream) {
  VN_CHECK_ARG(dz && z && dy && quad_ok(*dz, esize(dz_dtype)) && quad_ok(*z, esize(z_dtype)) &&
                   quad_ok(*dy, esize(dy_dtype)) && same_dims(*dz, *z) && same_dims(*dz, *dy), "act_bwd: bad views");
  const long total = view_voxels(*z) * (z->C / 4);
  const dim3 g(ew_grid(total)), blk(256);
  hipStream_t s = (hipStream_t)stream;
  const TView a = make_view(*dz), b = make_view(*z), c = make_view(*dy);
#define ACT_CASE(D1, T1, D2, T2, D3, T3) \
  if (dz_dtype == D1 && z_dtype == D2 && dy_dtype == D3) { hipLaunchKernelGGL((act_bwd_kernel<T1, T2, T3>), g, blk, 0, s, a, b, act, c, total); return vn_launch_status("act_bwd"); }
  ACT_CASE(VINET_F32, float, VINET_F32, float, VINET_F32, float)
  ACT_CASE(VINET_BF16, bf16_t, VINET_BF16, bf16_t, VINET_BF16, bf16_t)
  ACT_CASE(VINET_F32, float, VINET_F32, float, VINET_BF16, bf16_t)
  ACT_CASE(VINET_F32, float, VINET_BF16, bf16_t, VINET_BF16, bf16_t)
  ACT_CASE(VINET_BF16, bf16_t, VINET_F32, float, VINET_BF16, bf16_t)
#undef ACT_CASE
  vinet_set_error("act_bwd: unsupported dtype combination %d/%d/%d", dz_dtype, z_dtype, dy_dtype);
  return -1;
}

__global__ __launch_bounds__(256) void channel_sum_finalize_kernel(const float* __restrict__ partials, int rows, int C,
                                                                   int Cout, float* out, int accumulate) {
  const int j = blockIdx.x;
  double s = 0.0;
  const int per = C / Cout;   // channels folded onto output j: j, j + Cout, ...
  for (int e = threadIdx.x; e < rows * per; e += blockDim.x) {
    const int r = e / per, c = j + (e % per) * Cout;
    s += (double)partials[((long)r * 2) * C + c];
  }
  __shared__ double red[4];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    s = red[0] + red[1] + red[2] + red[3];
    out[j] = accumulate ? out[j] + (float)s : (float)s;
  }
}

extern "C" int vinet_channel_sum(const VinetTensor* x, int32_t dtype, float* workspace, int32_t Cout, float* out,
                                 int32_t accumulate, void* stream) {
  VN_CHECK_ARG(x && workspace && out && Cout > 0 && quad_ok(*x, esize(dtype)) && x->C % Cout == 0, "channel_sum: bad arguments");
  VinetAffine none = {nullptr, nullptr, 0};
  int rc = launch_channel_reduce<0>(x, nullptr, dtype, none, nullptr, nullptr, workspace, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(channel_sum_finalize_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, workspace,
                     stats_rows_for(view_voxels(*x)), x->C, Cout, out, accumulate);
  return vn_launch_status("channel_sum");
}

// ============================================================================
// MaxPool3d
// ============================================================================
struct PoolP { int kT, kH, kW, sT, sH, sW, pT, pH, pW; FastDiv dsT, dsH, dsW; };
static inline PoolP make_poolp(const VinetPoolDesc* d) {
  PoolP p = {d->kT, d->kH, d->kW, d->sT, d->sH, d->sW, d->pT, d->pH, d->pW, make_fastdiv((uint32_t)d->sT), make_fastdiv((uint32_t)d->sH), make_fastdiv((uint32_t)d->sW)};
  return p;
}

// Pooling semantics: a pending affine (+ReLU) is applied AND rounded to the activation dtype before the comparison --
// what a bf16 pipeline that had stored the BN+ReLU output would pool over, and what every other consumer of a
// pending activation sees (the conv kernels round at fragment time).  It also makes the packed 16-bit kernel below
// agree with these fp32-compare kernels on every tie.
template <typename T> VN_DEV float pool_round(float v) { return v; }
template <> VN_DEV float pool_round<bf16_t>(float v) { return bf2f(f2bf(v)); }

template <typename T>
__global__ void maxpool_fwd_kernel(PoolP p, TView x, Affine pre, TView y, uint8_t* __restrict__ argmax, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint32_t vox_u = fdiv((uint32_t)i, y.dQ);
  const int q = (int)((uint32_t)i - vox_u * (uint32_t)(y.C / 4));
  const long vox = (long)vox_u;
  int b, to, ho, wo;
  decode_vox(y, vox, b, to, ho, wo);
  float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  int bi[4] = {0, 0, 0, 0};
  for (int kt = 0; kt < p.kT; ++kt) {
    const int t = to * p.sT - p.pT + kt;
    if ((unsigned)t >= (unsigned)x.T) continue;
    for (int kh = 0; kh < p.kH; ++kh) {
      const int h = ho * p.sH - p.pH + kh;
      if ((unsigned)h >= (unsigned)x.H) continue;
      for (int kw = 0; kw < p.kW; ++kw) {
        const int w = wo * p.sW - p.pW + kw;
        if ((unsigned)w >= (unsigned)x.W) continue;
        float4 v = ldq<T>((const T*)x.p + vox_off(x, b, t, h, w) + q * 4);
        v = affine4(v, pre, q * 4);
        if (pre.scale) { v.x = pool_round<T>(v.x); v.y = pool_round<T>(v.y); v.z = pool_round<T>(v.z); v.w = pool_round<T>(v.w); }
        const int tap = (kt * p.kH + kh) * p.kW + kw;
        const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (f[e] > best[e] || (f[e] != f[e] && best[e] == best[e])) { best[e] = f[e]; bi[e] = tap; }
      }
    }
  }
  stq<T>((T*)y.p + vox_off(y, b, to, ho, wo) + q * 4, make_float4(best[0], best[1], best[2], best[3]));
  if (argmax) *(uint32_t*)(argmax + vox * y.C + q * 4) = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
}

// kT == 3, sT == 1, pT == 1 (the Inception branch-3 pools, model_utils.py:178): one thread
// walks T for a fixed output (h,w), keeps the maxima of the last three (kH x kW) planes and
// so reads kH*kW instead of 3*kH*kW inputs per output.  Same first-max tie rule: planes in
// t order, (h,w) scan order inside a plane, strict comparisons.
template <typename T>
__global__ void maxpool_tslide_kernel(PoolP p, TView x, Affine pre, TView y, uint8_t* __restrict__ argmax, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint32_t col = fdiv((uint32_t)i, y.dQ);
  const int q = (int)((uint32_t)i - col * (uint32_t)(y.C / 4));
  const uint32_t r1 = fdiv(col, y.dW);
  const int wo = (int)(col - r1 * (uint32_t)y.W);
  const uint32_t r2 = fdiv(r1, y.dH);
  const int ho = (int)(r1 - r2 * (uint32_t)y.H);
  const int b = (int)r2;
  float pm[3][4];
  int pa[3][4];
  const int khw = p.kH * p.kW;
  // plane tp feeds outputs tp-1, tp, tp+1; output t is complete once plane t+1 is in
  for (int tp = 0; tp <= x.T; ++tp) {
    const int slot = tp % 3;
    if (tp < x.T) {
      float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      int bi[4] = {0, 0, 0, 0};
      for (int kh = 0; kh < p.kH; ++kh) {
        const int h = ho * p.sH - p.pH + kh;
        if ((unsigned)h >= (unsigned)x.H) continue;
        for (int kw = 0; kw < p.kW; ++kw) {
          const int w = wo * p.sW - p.pW + kw;
          if ((unsigned)w >= (unsigned)x.W) continue;
          float4 v = ldq<T>((const T*)x.p + vox_off(x, b, tp, h, w) + q * 4);
          v = affine4(v, pre, q * 4);
          if (pre.scale) { v.x = pool_round<T>(v.x); v.y = pool_round<T>(v.y); v.z = pool_round<T>(v.z); v.w = pool_round<T>(v.w); }
        if (pre.scale) { v.x = pool_round<T>(v.x); v.y = pool_round<T>(v.y); v.z = pool_round<T>(v.z); v.w = pool_round<T>(v.w); }
          const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (f[e] > best[e] || (f[e] != f[e] && best[e] == best[e])) { best[e] = f[e]; bi[e] = kh * p.kW + kw; }
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { pm[slot][e] = best[e]; pa[slot][e] = bi[e]; }
    }
    const int to = tp - 1;
    if (to < 0) continue;
    float o[4];
    int oi[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = -INFINITY; oi[e] = 0; }
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      const int t = to - 1 + kt;
      if (t < 0 || t >= x.T) continue;
      const int sl = t % 3;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (pm[sl][e] > o[e] || (pm[sl][e] != pm[sl][e] && o[e] == o[e])) { o[e] = pm[sl][e]; oi[e] = kt * khw + pa[sl][e]; }
    }
    stq<T>((T*)y.p + vox_off(y, b, to, ho, wo) + q * 4, make_float4(o[0], o[1], o[2], o[3]));
    if (argmax) {
      const long ovox = (((long)b * y.T + to) * y.H + ho) * y.W + wo;
      *(uint32_t*)(argmax + ovox * y.C + q * 4) = (uint32_t)oi[0] | ((uint32_t)oi[1] << 8) | ((uint32_t)oi[2] << 16) | ((uint32_t)oi[3] << 24);
    }
  }
}

// 8 channels per lane forms of the two forward kernels above (same scan order and tie rule)
VN_DEV void affine8(float* v, const Affine& a, int c) {
  if (a.scale) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], a.scale[c + e], a.shift[c + e]);
  }
  if (a.relu) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd8_kernel(PoolP p, TView x, Affine pre, TView y, uint8_t* __restrict__ argmax, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int G = y.C >> 3;
  const uint32_t vox_u = (uint32_t)(i / G);
  const int g = (int)(i - (long)vox_u * G);
  int b, to, ho, wo;
  decode_vox(y, (long)vox_u, b, to, ho, wo);
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sc[e] = pre.scale ? pre.scale[g * 8 + e] : 1.f; sh[e] = pre.scale ? pre.shift[g * 8 + e] : 0.f; }
  float best[8];
  unsigned long long bi = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) best[e] = -INFINITY;
  for (int kt = 0; kt < p.kT; ++kt) {
    const int t = to * p.sT - p.pT + kt;
    if ((unsigned)t >= (unsigned)x.T) continue;
    for (int kh = 0; kh < p.kH; ++kh) {
      const int h = ho * p.sH - p.pH + kh;
      if ((unsigned)h >= (unsigned)x.H) continue;
      for (int kw = 0; kw < p.kW; ++kw) {
        const int w = wo * p.sW - p.pW + kw;
        if ((unsigned)w >= (unsigned)x.W) continue;
        float f[8];
        ld8<T>((const T*)x.p + vox_off(x, b, t, h, w) + g * 8, f);
        const unsigned long long tap = (unsigned long long)((kt * p.kH + kh) * p.kW + kw);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float v = fmaf(f[e], sc[e], sh[e]);
          if (pre.relu) v = fmaxf(v, 0.f);
          if (pre.scale) v = pool_round<T>(v);
          if (v > best[e] || (v != v && best[e] == best[e])) { best[e] = v; bi = (bi & ~(0xffull << (8 * e))) | (tap << (8 * e)); }
        }
      }
    }
  }
  st8<T>((T*)y.p + vox_off(y, b, to, ho, wo) + g * 8, best);
  if (argmax) *(unsigned long long*)(argmax + (long)vox_u * y.C + g * 8) = bi;
}

// 8-channel T-walking forward for kT == 3, sT == 1, pT == 1 (any in-plane window): one lane owns an output
// column (b, ho, wo, 8 channels), computes each input plane's (kH x kW) window maximum ONCE and keeps the
// last three in named registers (explicit rotation: a runtime-indexed ring would live in scratch).  Tie rule
// as everywhere: planes in t order, (h, w) scan order inside a plane, strict comparisons.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_tslide8_kernel(PoolP p, TView x, Affine pre, TView y, uint8_t* __restrict__ argmax, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int G = y.C >> 3;
  const uint32_t col = (uint32_t)(i / G);
  const int g = (int)(i - (long)col * G);
  const uint32_t r1 = fdiv(col, y.dW);
  const int wo = (int)(col - r1 * (uint32_t)y.W);
  const uint32_t r2 = fdiv(r1, y.dH);
  const int ho = (int)(r1 - r2 * (uint32_t)y.H);
  const int b = (int)r2;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sc[e] = pre.scale ? pre.scale[g * 8 + e] : 1.f; sh[e] = pre.scale ? pre.shift[g * 8 + e] : 0.f; }
  const int khw = p.kH * p.kW;
  float m_a[8], m_b[8], m_c[8];                 // plane maxima of planes tp-2, tp-1, tp
  unsigned long long i_a = 0, i_b = 0, i_c = 0;   // ... and their in-plane argmax codes (8 x 8 bit)
#pragma unroll
  for (int e = 0; e < 8; ++e) { m_a[e] = -INFINITY; m_b[e] = -INFINITY; m_c[e] = -INFINITY; }
  for (int tp = 0; tp <= x.T; ++tp) {
    // rotate: (a, b, c) <- (b, c, new plane tp)
#pragma unroll
    for (int e = 0; e < 8; ++e) { m_a[e] = m_b[e]; m_b[e] = m_c[e]; m_c[e] = -INFINITY; }
    i_a = i_b; i_b = i_c; i_c = 0;
    if (tp < x.T) {
      for (int kh = 0; kh < p.kH; ++kh) {
        const int h = ho * p.sH - p.pH + kh;
        if ((unsigned)h >= (unsigned)x.H) continue;
        for (int kw = 0; kw < p.kW; ++kw) {
          const int w = wo * p.sW - p.pW + kw;
          if ((unsigned)w >= (unsigned)x.W) continue;
          float f[8];
          ld8<T>((const T*)x.p + vox_off(x, b, tp, h, w) + g * 8, f);
          const unsigned long long code = (unsigned long long)(kh * p.kW + kw);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float v = fmaf(f[e], sc[e], sh[e]);
            if (pre.relu) v = fmaxf(v, 0.f);
            if (pre.scale) v = pool_round<T>(v);
            if (v > m_c[e] || (v != v && m_c[e] == m_c[e])) { m_c[e] = v; i_c = (i_c & ~(0xffull << (8 * e))) | (code << (8 * e)); }
          }
        }
      }
    }
    const int to = tp - 1;       // complete once plane tp = to + 1 is in: window planes (a, b, c) = (to-1, to, to+1)
    if (to < 0) continue;
    float o[8];
    unsigned long long oi = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float best = -INFINITY;
      unsigned long long bi = 0;
      // kt = 0: plane to-1 (exists iff to >= 1; otherwise m_a is -inf and never wins against a real plane)
      if (to >= 1 && (m_a[e] > best || (m_a[e] != m_a[e] && best == best))) { best = m_a[e]; bi = (i_a >> (8 * e)) & 0xffull; }
      if (m_b[e] > best || (m_b[e] != m_b[e] && best == best)) { best = m_b[e]; bi = (unsigned long long)khw + ((i_b >> (8 * e)) & 0xffull); }
      if (to + 1 < x.T && (m_c[e] > best || (m_c[e] != m_c[e] && best == best))) { best = m_c[e]; bi = 2ull * khw + ((i_c >> (8 * e)) & 0xffull); }
      o[e] = best;
      oi |= bi << (8 * e);
    }
    st8<T>((T*)y.p + vox_off(y, b, to, ho, wo) + g * 8, o);
    if (argmax) {
      const long ovox = (((long)b * y.T + to) * y.H + ho) * y.W + wo;
      *(unsigned long long*)(argmax + ovox * y.C + g * 8) = oi;
    }
  }
}

// 3x3x3 / s1 / p1 forward through an LDS halo tile: a 512-thread workgroup owns 8 x 8 outputs x 64 channels of
// one clip and walks T.  Per input plane the 10 x 10 halo (affine + ReLU applied once per element, fp32) is staged
// in LDS (double buffered: one barrier per plane), every lane takes its 3 x 3 window maximum from LDS, and the
// last three plane maxima live in named registers as in maxpool_tslide8_kernel.  One global read per input
// element (plus halo) instead of nine cached ones: the T-walking kernel is bound by the CU's load path (0.8 TB/s).
// Out-of-range halo positions hold -inf: strict comparisons never select them, so the tie rule is unchanged.
template <typename T>
__global__ __launch_bounds__(512) void maxpool_k3s1_lds_kernel(TView x, Affine pre, TView y, uint8_t* __restrict__ argmax,
                                                               int tilesH, int tilesW) {
  __shared__ __attribute__((aligned(16))) float P[2][100][64];
  const int tid = threadIdx.x;
  const int oct = tid & 7, pos = tid >> 3;          // 8 channel octets x 64 positions
  const int ph = pos >> 3, pw = pos & 7;
  int bid = blockIdx.x;
  const int ncg = (x.C + 63) >> 6;                  // channel groups of 64 (the last may be partial)
  const int cg = bid % ncg; bid /= ncg;
  const int tw = bid % tilesW; bid /= tilesW;
  const int th = bid % tilesH; bid /= tilesH;
  const int b = bid;
  const int h0 = th * 8, w0 = tw * 8, c0 = cg * 64 + oct * 8;
  const int ho = h0 + ph, wo = w0 + pw;
  const bool out_ok = ho < y.H && wo < y.W && c0 < x.C;
  // scale / shift of the tile's 64 channels in LDS (16 more live registers per lane would cost a workgroup of
  // occupancy, which this latency-bound walk cannot afford)
  __shared__ __attribute__((aligned(16))) float S[2][64];
  if (tid < 64) {
    const bool ok = pre.scale != nullptr && cg * 64 + tid < x.C;
    S[0][tid] = ok ? pre.scale[cg * 64 + tid] : 1.f;
    S[1][tid] = ok ? pre.shift[cg * 64 + tid] : 0.f;
  }
  __syncthreads();
  float m_a[8], m_b[8], m_c[8];
  unsigned long long i_a = 0, i_b = 0, i_c = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) { m_a[e] = -INFINITY; m_b[e] = -INFINITY; m_c[e] = -INFINITY; }
  const int T_ = x.T;
  for (int tp = 0; tp <= T_; ++tp) {
    float (*buf)[64] = P[tp & 1];
    if (tp < T_) {
      // stage the halo of plane tp: 100 positions x 8 octets = 800 items over 512 threads
      for (int it = tid; it < 800; it += 512) {
        const int o8 = oct, hp = it >> 3;           // (it & 7) == oct
        const int hh = h0 - 1 + hp / 10, ww = w0 - 1 + hp % 10;
        float v[8];
        if (cg * 64 + o8 * 8 >= x.C) continue;        // partial last channel group
        if ((unsigned)hh < (unsigned)x.H && (unsigned)ww < (unsigned)x.W) {
          ld8<T>((const T*)x.p + vox_off(x, b, tp, hh, ww) + cg * 64 + o8 * 8, v);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            v[e] = fmaf(v[e], S[0][oct * 8 + e], S[1][oct * 8 + e]);
            if (pre.relu) v[e] = fmaxf(v[e], 0.f);
            if (pre.scale) v[e] = pool_round<T>(v[e]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = -INFINITY;
        }
        *(float4*)&buf[hp][o8 * 8] = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)&buf[hp][o8 * 8 + 4] = make_float4(v[4], v[5], v[6], v[7]);
      }
    }
    __syncthreads();     // plane tp staged; every lane finished reading the buffer staged two planes ago
#pragma unroll
    for (int e = 0; e < 8; ++e) { m_a[e] = m_b[e]; m_b[e] = m_c[e]; m_c[e] = -INFINITY; }
    i_a = i_b; i_b = i_c; i_c = 0;
    if (tp < T_) {
      // window maximum of the plane, 4 channels at a time: the max by four 3-input maxima, then the FIRST tap
      // that equals it (scan from the last tap down, so the smallest index is the one left standing) -- 20 VALU
      // per element instead of ~45 for compare-and-track.  v_max3 drops NaNs where aten propagates them: a NaN
      // anywhere in the window (sum test) takes the compare-and-track path.
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        float f[9][4];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const float4 q = *(const float4*)&buf[(ph + kh) * 10 + pw + kw][oct * 8 + half * 4];
            f[kh * 3 + kw][0] = q.x; f[kh * 3 + kw][1] = q.y; f[kh * 3 + kw][2] = q.z; f[kh * 3 + kw][3] = q.w;
          }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int e = half * 4 + c;
          const float sum = ((f[0][c] + f[1][c]) + (f[2][c] + f[3][c])) + ((f[4][c] + f[5][c]) + (f[6][c] + f[7][c])) + f[8][c];
          float m;
          unsigned idx;
          if (sum == sum) {
            m = fmaxf(fmaxf(fmaxf(f[0][c], f[1][c]), f[2][c]), fmaxf(fmaxf(fmaxf(f[3][c], f[4][c]), f[5][c]), fmaxf(fmaxf(f[6][c], f[7][c]), f[8][c])));
            idx = 8;
#pragma unroll
            for (int k = 7; k >= 0; --k) idx = (f[k][c] == m) ? (unsigned)k : idx;
          } else {
            m = -INFINITY; idx = 0;
#pragma unroll
            for (int k = 0; k < 9; ++k)
              if (f[k][c] > m || (f[k][c] != f[k][c] && m == m)) { m = f[k][c]; idx = (unsigned)k; }
          }
          m_c[e] = m;
          i_c |= (unsigned long long)idx << (8 * e);
        }
      }
    }
    const int to = tp - 1;
    if (to < 0 || !out_ok) continue;
    float o[8];
    unsigned long long oi = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float best = -INFINITY;
      unsigned long long bi = 0;
      if (to >= 1 && (m_a[e] > best || (m_a[e] != m_a[e] && best == best))) { best = m_a[e]; bi = (i_a >> (8 * e)) & 0xffull; }
      if (m_b[e] > best || (m_b[e] != m_b[e] && best == best)) { best = m_b[e]; bi = 9ull + ((i_b >> (8 * e)) & 0xffull); }
      if (to + 1 < T_ && (m_c[e] > best || (m_c[e] != m_c[e] && best == best))) { best = m_c[e]; bi = 18ull + ((i_c >> (8 * e)) & 0xffull); }
      o[e] = best;
      oi |= bi << (8 * e);
    }
    st8<T>((T*)y.p + vox_off(y, b, to, ho, wo) + c0, o);
    if (argmax) {
      const long ovox = (((long)b * y.T + to) * y.H + ho) * y.W + wo;
      *(unsigned long long*)(argmax + ovox * y.C + c0) = oi;
    }
  }
}


// order-preserving 16-bit codes of two packed bf16 (see maxpool_k3s1_pk_kernel) and their inverse
VN_DEV uint32_t pool_code2(uint32_t u) {
  const uint32_t m = ((u >> 15) & 0x00010001u) * 0x7fffu;
  return u ^ (m | 0x80008000u);
}
VN_DEV uint32_t pool_decode2(uint32_t k) {
  const uint32_t m = (((k >> 15) & 0x00010001u) ^ 0x00010001u) * 0x7fffu;
  return k ^ (m | 0x80008000u);
}

// Generic window (any k / stride / padding, up to 255 taps), bf16, on packed keys: one lane = one output voxel x 8
// channels; every in-range tap is loaded (16 B), transformed, coded and folded with key = code << 16 | (ntaps-1-tap).
// 7 VALU per element and tap instead of ~12 for affine + round + compare-and-track in fp32.
__global__ __launch_bounds__(256) void maxpool_fwd8_pk_kernel(PoolP p, TView x, Affine pre, TView y, uint8_t* __restrict__ argmax, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int G = y.C >> 3;
  const uint32_t vox_u = (uint32_t)(i / G);
  const int g = (int)(i - (long)vox_u * G);
  int b, to, ho, wo;
  decode_vox(y, (long)vox_u, b, to, ho, wo);
  f32x2_v sc2[4], sh2[4];
  const bool aff = pre.scale != nullptr;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    sc2[e] = aff ? (f32x2_v){pre.scale[g * 8 + 2 * e], pre.scale[g * 8 + 2 * e + 1]} : (f32x2_v){1.f, 1.f};
    sh2[e] = aff ? (f32x2_v){pre.shift[g * 8 + 2 * e], pre.shift[g * 8 + 2 * e + 1]} : (f32x2_v){0.f, 0.f};
  }
  uint32_t best[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) best[e] = 0;
  const uint32_t last = (uint32_t)(p.kT * p.kH * p.kW - 1);
  for (int kt = 0; kt < p.kT; ++kt) {
    const int t = to * p.sT - p.pT + kt;
    if ((unsigned)t >= (unsigned)x.T) continue;
    for (int kh = 0; kh < p.kH; ++kh) {
      const int h = ho * p.sH - p.pH + kh;
      if ((unsigned)h >= (unsigned)x.H) continue;
      for (int kw = 0; kw < p.kW; ++kw) {
        const int w = wo * p.sW - p.pW + kw;
        if ((unsigned)w >= (unsigned)x.W) continue;
        const uint4 q = *(const uint4*)((const bf16_t*)x.p + vox_off(x, b, t, h, w) + g * 8);
        uint32_t w4[4] = {q.x, q.y, q.z, q.w};
        const uint32_t ck = last - (uint32_t)((kt * p.kH + kh) * p.kW + kw);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (aff) {
            if (pre.relu) w4[e] = pre_relu_pair(w4[e], sc2[e], sh2[e]);
            else w4[e] = pack2bf(fmaf(__uint_as_float(w4[e] << 16), sc2[e].x, sh2[e].x), fmaf(__uint_as_float(w4[e] & 0xffff0000u), sc2[e].y, sh2[e].y));
          } else if (pre.relu) {
            asm("v_pk_max_i16 %0, %1, 0" : "=v"(w4[e]) : "v"(w4[e]));
          }
          const uint32_t c = pool_code2(w4[e]);
          const uint32_t klo = (c << 16) | ck, khi = (c & 0xffff0000u) | ck;
          best[2 * e] = best[2 * e] > klo ? best[2 * e] : klo;
          best[2 * e + 1] = best[2 * e + 1] > khi ? best[2 * e + 1] : khi;
        }
      }
    }
  }
  uint32_t ov[4];
  unsigned long long bi = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const uint32_t k = best[e];
    if (e & 1) ov[e >> 1] |= k & 0xffff0000u; else ov[e >> 1] = k >> 16;
    bi |= (unsigned long long)(last - (k & 0xffu)) << (8 * e);
  }
  uint4 o;
  o.x = pool_decode2(ov[0]); o.y = pool_decode2(ov[1]); o.z = pool_decode2(ov[2]); o.w = pool_decode2(ov[3]);
  *(uint4*)((bf16_t*)y.p + vox_off(y, b, to, ho, wo) + g * 8) = o;
  if (argmax) *(unsigned long long*)(argmax + (long)vox_u * y.C + g * 8) = bi;
}

// bf16 form of the kernel above on PACKED 32-bit keys.  The halo holds order-preserving 16-bit codes of the
// (affine + ReLU'd, bf16-rounded) activations: code = bits ^ 0x8000 for non-negative values, ~bits for negative
// ones, so unsigned integer order = numeric order and 0 is below everything (out-of-range taps).  A window
// candidate becomes key = code << 16 | (8 - tap): v_max3_u32 over the nine keys yields the maximum AND, in its low
// bits, the first tap that attains it (ties: the larger low field = the smaller tap) -- 1 op to build a key, half an
// op to fold it, instead of ~20 for max-then-find-first in fp32.  The three plane keys are folded the same way with
// +18 / +9 / +0 (earlier plane wins ties), so tap = 26 - (key & 0xff).  Half the LDS bytes per plane, 9 instead of 18
// ds_read_b128 per lane and plane.  (A sign-bit NaN would order lowest; aten's NaN-wins only holds for positive NaNs.)
__global__ __launch_bounds__(512) void maxpool_k3s1_pk_kernel(TView x, Affine pre, TView y, uint8_t* __restrict__ argmax,
                                                              int tilesH, int tilesW) {
  __shared__ __attribute__((aligned(16))) uint16_t P[2][100][64];
  __shared__ __attribute__((aligned(16))) float S[2][64];
  const int tid = threadIdx.x;
  const int oct = tid & 7, pos = tid >> 3;
  const int ph = pos >> 3, pw = pos & 7;
  int bid = blockIdx.x;
  const int ncg = (x.C + 63) >> 6;
  const int cg = bid % ncg; bid /= ncg;
  const int tw = bid % tilesW; bid /= tilesW;
  const int th = bid % tilesH; bid /= tilesH;
  const int b = bid;
  const int h0 = th * 8, w0 = tw * 8, c0 = cg * 64 + oct * 8;
  const int ho = h0 + ph, wo = w0 + pw;
  const bool out_ok = ho < y.H && wo < y.W && c0 < x.C;
  const bool aff = pre.scale != nullptr;
  if (tid < 64) {
    const bool ok = aff && cg * 64 + tid < x.C;
    S[0][tid] = ok ? pre.scale[cg * 64 + tid] : 1.f;
    S[1][tid] = ok ? pre.shift[cg * 64 + tid] : 0.f;
  }
  __syncthreads();
  uint32_t m_a[8], m_b[8], m_c[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { m_a[e] = 0; m_b[e] = 0; m_c[e] = 0; }
  const int T_ = x.T;
  for (int tp = 0; tp <= T_; ++tp) {
    uint16_t (*buf)[64] = P[tp & 1];
    if (tp < T_) {
      for (int it = tid; it < 800; it += 512) {
        const int hp = it >> 3;                      // (it & 7) == oct
        const int hh = h0 - 1 + hp / 10, ww = w0 - 1 + hp % 10;
        if (cg * 64 + oct * 8 >= x.C) continue;
        uint4 q = make_uint4(0, 0, 0, 0);
        if ((unsigned)hh < (unsigned)x.H && (unsigned)ww < (unsigned)x.W) {
          q = *(const uint4*)((const bf16_t*)x.p + vox_off(x, b, tp, hh, ww) + cg * 64 + oct * 8);
          if (aff) {
            const float2* sp = (const float2*)&S[0][oct * 8];
            const float2* hp2 = (const float2*)&S[1][oct * 8];
            uint32_t* w4 = (uint32_t*)&q;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float2 s2 = sp[e], h2 = hp2[e];
              if (pre.relu) w4[e] = pre_relu_pair(w4[e], (f32x2_v){s2.x, s2.y}, (f32x2_v){h2.x, h2.y});
              else w4[e] = pack2bf(fmaf(__uint_as_float(w4[e] << 16), s2.x, h2.x), fmaf(__uint_as_float(w4[e] & 0xffff0000u), s2.y, h2.y));
            }
          } else if (pre.relu) {
            uint32_t* w4 = (uint32_t*)&q;
#pragma unroll
            for (int e = 0; e < 4; ++e) asm("v_pk_max_i16 %0, %1, 0" : "=v"(w4[e]) : "v"(w4[e]));
          }
          q.x = pool_code2(q.x); q.y = pool_code2(q.y); q.z = pool_code2(q.z); q.w = pool_code2(q.w);
        }
        *(uint4*)&buf[hp][oct * 8] = q;
      }
    }
    __syncthreads();     // plane tp staged; every lane finished reading the buffer staged two planes ago
#pragma unroll
    for (int e = 0; e < 8; ++e) { m_a[e] = m_b[e]; m_b[e] = m_c[e]; m_c[e] = 0; }
    if (tp < T_) {
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const uint4 q = *(const uint4*)&buf[(ph + kh) * 10 + pw + kw][oct * 8];
          const uint32_t ck = 8u - (uint32_t)(kh * 3 + kw);
          const uint32_t w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t klo = (w4[e] << 16) | ck, khi = (w4[e] & 0xffff0000u) | ck;
            m_c[2 * e] = m_c[2 * e] > klo ? m_c[2 * e] : klo;
            m_c[2 * e + 1] = m_c[2 * e + 1] > khi ? m_c[2 * e + 1] : khi;
          }
        }
    }
    const int to = tp - 1;
    if (to < 0 || !out_ok) continue;
    uint32_t ov[4];
    unsigned long long oi = 0;
    const bool va = to >= 1, vc = to + 1 < T_;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint32_t ka = va ? m_a[e] + 18u : 0u, kb = m_b[e] + 9u, kc = vc ? m_c[e] : 0u;
      uint32_t k = ka > kb ? ka : kb;
      k = k > kc ? k : kc;
      if (e & 1) ov[e >> 1] |= k & 0xffff0000u; else ov[e >> 1] = k >> 16;
      oi |= (unsigned long long)(26u - (k & 0xffu)) << (8 * e);
    }
    uint4 o;
    o.x = pool_decode2(ov[0]); o.y = pool_decode2(ov[1]); o.z = pool_decode2(ov[2]); o.w = pool_decode2(ov[3]);
    *(uint4*)((bf16_t*)y.p + vox_off(y, b, to, ho, wo) + c0) = o;
    if (argmax) {
      const long ovox = (((long)b * y.T + to) * y.H + ho) * y.W + wo;
      *(unsigned long long*)(argmax + ovox * y.C + c0) = oi;
    }
  }
}

extern "C" int vinet_maxpool3d(const VinetPoolDesc* d, const VinetTensor* x, VinetAffine pre, const VinetTensor* y,
                               uint8_t* argmax, void* stream) {
  VN_CHECK_ARG(d && x && y && quad_ok(*x, esize(d->dtype)) && quad_ok(*y, esize(d->dtype)) && x->C == y->C && x->B == y->B,
               "maxpool3d: bad views");
  VN_CHECK_ARG(d->kT * d->kH * d->kW <= 255 && d->kT > 0 && d->kH > 0 && d->kW > 0, "maxpool3d: window too large");
  const PoolP p = make_poolp(d);
  const bool k3s1 = d->kT == 3 && d->kH == 3 && d->kW == 3 && d->sT == 1 && d->sH == 1 && d->sW == 1 && d->pT == 1 && d->pH == 1 &&
                    d->pW == 1 && y->T == x->T && y->H == x->H && y->W == x->W;
  if (k3s1 && g_vinet_opt_pool_lds && x->T >= 2 && oct_ok(*x) && oct_ok(*y) && (!argmax || ((uintptr_t)argmax % 8) == 0) &&
      (g_vinet_opt_pool_lds >= 2 || (long)y->B * y->H * y->W * (y->C / 8) >= 65536)) {
    const int tilesH = (y->H + 7) / 8, tilesW = (y->W + 7) / 8;
    const long blocks = (long)y->B * tilesH * tilesW * ((y->C + 63) / 64);
    const bool pre_ok = !pre.scale || pre.shift;
    if (d->dtype == VINET_BF16 && g_vinet_opt_pool_pk && pre_ok && x->ld % 8 == 0 && ((uintptr_t)x->ptr % 16) == 0 && ((uintptr_t)y->ptr % 16) == 0 &&
        x->sB % 8 == 0 && y->sB % 8 == 0) {
      hipLaunchKernelGGL(maxpool_k3s1_pk_kernel, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, make_view(*x),
                         make_affine(pre), make_view(*y), argmax, tilesH, tilesW);
      return vn_launch_status("maxpool3d(k3s1 packed keys)");
    }
    DISPATCH_T(d->dtype, T, hipLaunchKernelGGL(maxpool_k3s1_lds_kernel<T>, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream,
                                               make_view(*x), make_affine(pre), make_view(*y), argmax, tilesH, tilesW);)
    return vn_launch_status("maxpool3d(k3s1 lds)");
  }
  if (d->kT == 3 && d->sT == 1 && d->pT == 1 && y->T == x->T && x->T >= 2 && oct_ok(*x) && oct_ok(*y) &&
      (!argmax || ((uintptr_t)argmax % 8) == 0) &&
      (g_vinet_opt_pool_twalk >= 2 || (long)y->B * y->H * y->W * (y->C / 8) >= 65536)) {
    // (fewer columns than that cannot fill the chip while each lane walks T serially: batch-1 inference
    //  takes the one-thread-per-output kernel below)
    const long cols8 = (long)y->B * y->H * y->W * (y->C / 8);
    DISPATCH_T(d->dtype, T, hipLaunchKernelGGL(maxpool_tslide8_kernel<T>, dim3(ew_grid(cols8)), dim3(256), 0,
                                               (hipStream_t)stream, p, make_view(*x), make_affine(pre), make_view(*y), argmax, cols8);)
    return vn_launch_status("maxpool3d(tslide8)");
  }
  if (d->kT == 3 && d->sT == 1 && d->pT == 1 && y->T == x->T && x->T >= 2 && !(oct_ok(*x) && oct_ok(*y))) {
    const long cols = (long)y->B * y->H * y->W * (y->C / 4);
    DISPATCH_T(d->dtype, T, hipLaunchKernelGGL(maxpool_tslide_kernel<T>, dim3(ew_grid(cols)), dim3(256), 0,
                                               (hipStream_t)stream, p, make_view(*x), make_affine(pre), make_view(*y), argmax, cols);)
    return vn_launch_status("maxpool3d(tslide)");
  }
  if (oct_ok(*x) && oct_ok(*y) && (!argmax || ((uintptr_t)argmax % 8) == 0)) {
    const long total8 = view_voxels(*y) * (y->C / 8);
    if (d->dtype == VINET_BF16 && g_vinet_opt_pool_pk && (!pre.scale || pre.shift) && ((uintptr_t)x->ptr % 16) == 0 && ((uintptr_t)y->ptr % 16) == 0 &&
        x->sB % 8 == 0 && y->sB % 8 == 0) {
      hipLaunchKernelGGL(maxpool_fwd8_pk_kernel, dim3(ew_grid(total8)), dim3(256), 0, (hipStream_t)stream, p, make_view(*x),
                         make_affine(pre), make_view(*y), argmax, total8);
      return vn_launch_status("maxpool3d(8, packed keys)");
    }
    DISPATCH_T(d->dtype, T, hipLaunchKernelGGL(maxpool_fwd8_kernel<T>, dim3(ew_grid(total8)), dim3(256), 0,
                                               (hipStream_t)stream, p, make_view(*x), make_affine(pre), make_view(*y), argmax, total8);)
    return vn_launch_status("maxpool3d(8)");
  }
  const long total = view_voxels(*y) * (y->C / 4);
  DISPATCH_T(d->dtype, T, hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3(ew_grid(total)), dim3(256), 0,
                                             (hipStream_t)stream, p, make_view(*x), make_affine(pre), make_view(*y), argmax, total);)
  return vn_launch_status("maxpool3d");
}

// backward as a gather over the (at most ceil(k/s)^3) windows covering each input voxel
template <typename T>
__global__ void maxpool_bwd_kernel(PoolP p, TView dy, const uint8_t* __restrict__ argmax, TView dx, int accumulate,
                                   long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint32_t vox_u = fdiv((uint32_t)i, dx.dQ);
  const int q = (int)((uint32_t)i - vox_u * (uint32_t)(dx.C / 4));
  const long vox = (long)vox_u;
  int b, t, h, w;
  decode_vox(dx, vox, b, t, h, w);
  float g[4] = {0, 0, 0, 0};
  // windows: o*s - pad <= pos <= o*s - pad + k - 1
  // (all numerators are >= 0 after the max: fast unsigned division)
  const int to1 = min((int)fdiv((uint32_t)(t + p.pT), p.dsT), dy.T - 1), ho1 = min((int)fdiv((uint32_t)(h + p.pH), p.dsH), dy.H - 1),
            wo1 = min((int)fdiv((uint32_t)(w + p.pW), p.dsW), dy.W - 1);
  const int to0 = (int)fdiv((uint32_t)max(0, t + p.pT - p.kT + p.sT), p.dsT), ho0 = (int)fdiv((uint32_t)max(0, h + p.pH - p.kH + p.sH), p.dsH),
            wo0 = (int)fdiv((uint32_t)max(0, w + p.pW - p.kW + p.sW), p.dsW);
  for (int to = to0; to <= to1; ++to) {
    const int kt = t + p.pT - to * p.sT;
    if (kt < 0 || kt >= p.kT) continue;
    for (int ho = ho0; ho <= ho1; ++ho) {
      const int kh = h + p.pH - ho * p.sH;
      if (kh < 0 || kh >= p.kH) continue;
      for (int wo = wo0; wo <= wo1; ++wo) {
        const int kw = w + p.pW - wo * p.sW;
        if (kw < 0 || kw >= p.kW) continue;
        const uint32_t tap = (uint32_t)((kt * p.kH + kh) * p.kW + kw);
        const long ovox = (((long)b * dy.T + to) * dy.H + ho) * dy.W + wo;
        const uint32_t am = *(const uint32_t*)(argmax + ovox * dy.C + q * 4);
        // a window routes its gradient to exactly one of its taps: most candidates do not match,
        // and then dy is not read at all (zero-byte test on am ^ tap-in-every-byte)
        const uint32_t xr = am ^ (tap * 0x01010101u);
        if (!((xr - 0x01010101u) & ~xr & 0x80808080u)) continue;
        const float4 d = ldq<T>((const T*)dy.p + vox_off(dy, b, to, ho, wo) + q * 4);
        if ((xr & 0xffu) == 0) g[0] += d.x;
        if ((xr & 0xff00u) == 0) g[1] += d.y;
        if ((xr & 0xff0000u) == 0) g[2] += d.z;
        if ((xr & 0xff000000u) == 0) g[3] += d.w;
      }
    }
  }
  T* dst = (T*)dx.p + vox_off(dx, b, t, h, w) + q * 4;
  if (accumulate) { const float4 o = ldq<T>(dst); g[0] += o.x; g[1] += o.y; g[2] += o.z; g[3] += o.w; }
  stq<T>(dst, make_float4(g[0], g[1], g[2], g[3]));
}

// 3x3x3 / stride 1 / pad 1 (the Inception branch-3 pools, model_utils.py:178): every input voxel
// is covered by up to 27 windows.  One thread owns 8 channels of one voxel, issues all 27
// argmax loads (8 codes = 8 bytes each) before looking at any of them, and reads dy only for
// the windows that route a gradient here.  The generic kernel walks the same 27 windows as a
// dependent load -> compare -> branch chain and is latency bound (0.5 TB/s).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_k3s1_kernel(TView dy, const uint8_t* __restrict__ argmax, TView dx,
                                                               int accumulate, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int G = dx.C >> 3;
  const uint32_t vox_u = (uint32_t)(i / G);          // G is small and launch-invariant; one division per thread
  const int g = (int)(i - (long)vox_u * G);
  int b, t, h, w;
  decode_vox(dx, (long)vox_u, b, t, h, w);
  const int T_ = dx.T, H = dx.H, W = dx.W;
  const long am_c = (long)vox_u * dx.C + g * 8;
  const long dy_c = vox_off(dy, b, t, h, w) + g * 8;
  unsigned long long am[27];
#pragma unroll
  for (int kt = 0; kt < 3; ++kt)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        // window (t+1-kt, h+1-kh, w+1-kw) holds this voxel as its tap (kt,kh,kw)
        const int dt = 1 - kt, dh = 1 - kh, dw = 1 - kw;
        const bool ok = (unsigned)(t + dt) < (unsigned)T_ && (unsigned)(h + dh) < (unsigned)H && (unsigned)(w + dw) < (unsigned)W;
        const long d = ((long)(dt * H + dh) * W + dw) * (long)dx.C;
        am[(kt * 3 + kh) * 3 + kw] = ok ? *(const unsigned long long*)(argmax + am_c + d) : ~0ull;
      }
  float gr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int kt = 0; kt < 3; ++kt)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int tap = (kt * 3 + kh) * 3 + kw;
        const unsigned long long xr = am[tap] ^ (0x0101010101010101ull * (unsigned long long)tap);
        if (!((xr - 0x0101010101010101ull) & ~xr & 0x8080808080808080ull)) continue;   // no zero byte: no match
        const int dt = 1 - kt, dh = 1 - kh, dw = 1 - kw;
        const T* src = (const T*)dy.p + dy_c + ((long)(dt * H + dh) * W + dw) * (long)dy.ld;
        const float4 d0 = ldq<T>(src), d1 = ldq<T>(src + 4);
        const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (((xr >> (8 * e)) & 0xffull) == 0) gr[e] += dv[e];
      }
  T* dst = (T*)dx.p + vox_off(dx, b, t, h, w) + g * 8;
  if (accumulate) {
    const float4 o0 = ldq<T>(dst), o1 = ldq<T>(dst + 4);
    gr[0] += o0.x; gr[1] += o0.y; gr[2] += o0.z; gr[3] += o0.w; gr[4] += o1.x; gr[5] += o1.y; gr[6] += o1.z; gr[7] += o1.w;
  }
  stq<T>(dst, make_float4(gr[0], gr[1], gr[2], gr[3]));
  stq<T>(dst + 4, make_float4(gr[4], gr[5], gr[6], gr[7]));
}

// generic backward, 8 channels per lane (same gather as maxpool_bwd_kernel)
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd8_kernel(PoolP p, TView dy, const uint8_t* __restrict__ argmax, TView dx, int accumulate,
                                                           long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int G = dx.C >> 3;
  const uint32_t vox_u = (uint32_t)(i / G);
  const int g = (int)(i - (long)vox_u * G);
  int b, t, h, w;
  decode_vox(dx, (long)vox_u, b, t, h, w);
  float gr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int to1 = min((int)fdiv((uint32_t)(t + p.pT), p.dsT), dy.T - 1), ho1 = min((int)fdiv((uint32_t)(h + p.pH), p.dsH), dy.H - 1),
            wo1 = min((int)fdiv((uint32_t)(w + p.pW), p.dsW), dy.W - 1);
  const int to0 = (int)fdiv((uint32_t)max(0, t + p.pT - p.kT + p.sT), p.dsT), ho0 = (int)fdiv((uint32_t)max(0, h + p.pH - p.kH + p.sH), p.dsH),
            wo0 = (int)fdiv((uint32_t)max(0, w + p.pW - p.kW + p.sW), p.dsW);
  for (int to = to0; to <= to1; ++to) {
    const int kt = t + p.pT - to * p.sT;
    if (kt < 0 || kt >= p.kT) continue;
    for (int ho = ho0; ho <= ho1; ++ho) {
      const int kh = h + p.pH - ho * p.sH;
      if (kh < 0 || kh >= p.kH) continue;
      for (int wo = wo0; wo <= wo1; ++wo) {
        const int kw = w + p.pW - wo * p.sW;
        if (kw < 0 || kw >= p.kW) continue;
        const unsigned long long tap = (unsigned long long)((kt * p.kH + kh) * p.kW + kw);
        const long ovox = (((long)b * dy.T + to) * dy.H + ho) * dy.W + wo;
        const unsigned long long am = *(const unsigned long long*)(argmax + ovox * dy.C + g * 8);
        const unsigned long long xr = am ^ (tap * 0x0101010101010101ull);
        if (!((xr - 0x0101010101010101ull) & ~xr & 0x8080808080808080ull)) continue;
        float dv[8];
        ld8<T>((const T*)dy.p + vox_off(dy, b, to, ho, wo) + g * 8, dv);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (((xr >> (8 * e)) & 0xffull) == 0) gr[e] += dv[e];
      }
    }
  }
  T* dst = (T*)dx.p + vox_off(dx, b, t, h, w) + g * 8;
  if (accumulate) {
    float o[8];
    ld8<T>(dst, o);
#pragma unroll
    for (int e = 0; e < 8; ++e) gr[e] += o[e];
  }
  st8<T>(dst, gr);
}

// 1x3x3 / s(1,2,2) / p(0,1,1) backward (the two big spatial pools, model.py:696,700): one lane owns the 2 x 2 input
// block {2ho, 2ho+1} x {2wo, 2wo+1} x 8 channels.  Only the four windows (ho..ho+1, wo..wo+1) reach it -- (ho,wo) all
// four inputs, (ho,wo+1) and (ho+1,wo) two each, (ho+1,wo+1) one -- so 4 argmax words and at most 4 dy rows serve 4
// outputs, and the voxel decode and window arithmetic are paid once per 64 bytes written instead of once per 16:
// the generic gather is bound by exactly that integer work (2.1 TB/s of tensors on the 112 x 192 pool).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_k133s2_kernel(TView dy, const uint8_t* __restrict__ argmax, TView dx, int accumulate,
                                                                 int HB, int WB, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int G = dx.C >> 3;
  long r = i / G;
  const int g = (int)(i - r * G);
  const int wb = (int)(r % WB); r /= WB;
  const int hb = (int)(r % HB); r /= HB;
  const int t = (int)(r % dx.T);
  const int b = (int)(r / dx.T);
  const int h0 = 2 * hb, w0 = 2 * wb;
  // windows q = dh*2 + dw at (hb + dh, wb + dw)
  unsigned long long am[4];
  bool wok[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int ho = hb + (q >> 1), wo = wb + (q & 1);
    wok[q] = ho < dy.H && wo < dy.W;
    const long ovox = (((long)b * dy.T + t) * dy.H + (wok[q] ? ho : 0)) * dy.W + (wok[q] ? wo : 0);
    am[q] = wok[q] ? *(const unsigned long long*)(argmax + ovox * dy.C + g * 8) : ~0ull;
  }
  float dv[4][8];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    // taps of window q that land in the block: rows kh in {1,2} (dh = 0) or {0} (dh = 1); same for columns
    bool any = false;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const bool in = ((q >> 1) ? kh == 0 : kh >= 1) && ((q & 1) ? kw == 0 : kw >= 1);
        if (!in) continue;
        const unsigned long long x = am[q] ^ ((unsigned long long)(kh * 3 + kw) * 0x0101010101010101ull);
        any |= ((x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull) != 0;
      }
    if (any) ld8<T>((const T*)dy.p + vox_off(dy, b, t, hb + (q >> 1), wb + (q & 1)) + g * 8, dv[q]);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) dv[q][e] = 0.f;
    }
  }
#pragma unroll
  for (int ih = 0; ih < 2; ++ih)
#pragma unroll
    for (int iw = 0; iw < 2; ++iw) {
      const int h = h0 + ih, w = w0 + iw;
      if (h >= dx.H || w >= dx.W) continue;
      float gr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int dh = q >> 1, dw = q & 1;
        // input (h, w) as tap (kh, kw) of window (hb + dh, wb + dw): kh = h + 1 - 2*(hb + dh) = ih + 1 - 2*dh
        const int kh = ih + 1 - 2 * dh, kw = iw + 1 - 2 * dw;
        if (kh < 0 || kw < 0) continue;                       // (compile-time: the window does not reach this input)
        const unsigned long long x = am[q] ^ ((unsigned long long)(kh * 3 + kw) * 0x0101010101010101ull);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (((x >> (8 * e)) & 0xffull) == 0) gr[e] += dv[q][e];
      }
      T* dst = (T*)dx.p + vox_off(dx, b, t, h, w) + g * 8;
      if (accumulate) {
        float o[8];
        ld8<T>(dst, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) gr[e] += o[e];
      }
      st8<T>(dst, gr);
    }
}

// 3x3x3 / s1 / p1 backward, T-walking form: one lane owns an input column (b, h, w, 8 channels) and walks the
// output planes; the argmax word of each of the 9 in-plane neighbour windows is read ONCE per plane and tested
// against the three temporal taps it could route to, accumulating into three named accumulators (inputs
// to-1, to, to+1).  9 argmax reads per voxel instead of 27.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_k3s1_twalk_kernel(TView dy, const uint8_t* __restrict__ argmax, TView dx,
                                                                     int accumulate, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int G = dx.C >> 3;
  const uint32_t col = (uint32_t)(i / G);
  const int g = (int)(i - (long)col * G);
  const uint32_t r1 = fdiv(col, dx.dW);
  const int w = (int)(col - r1 * (uint32_t)dx.W);
  const uint32_t r2 = fdiv(r1, dx.dH);
  const int h = (int)(r1 - r2 * (uint32_t)dx.H);
  const int b = (int)r2;
  const int T_ = dx.T, H = dx.H, W = dx.W;
  float g_m[8], g_0[8], g_p[8];      // gradients of inputs to-1, to, to+1 while output plane `to` is processed
#pragma unroll
  for (int e = 0; e < 8; ++e) { g_m[e] = 0.f; g_0[e] = 0.f; g_p[e] = 0.f; }
  for (int to = 0; to <= T_; ++to) {
    if (to < T_) {
      unsigned long long am[9];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int ho = h + 1 - kh, wo = w + 1 - kw;     // the window that holds (h, w) as its in-plane tap (kh, kw)
          const bool ok = (unsigned)ho < (unsigned)H && (unsigned)wo < (unsigned)W;
          const long ovox = (((long)b * T_ + to) * H + (ok ? ho : 0)) * W + (ok ? wo : 0);
          am[kh * 3 + kw] = ok ? *(const unsigned long long*)(argmax + ovox * dx.C + g * 8) : ~0ull;
        }
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const unsigned long long s = (unsigned long long)(kh * 3 + kw) * 0x0101010101010101ull;
          // codes kt*9 + s for kt = 0 (input to-1), 1 (input to), 2 (input to+1)
          const unsigned long long x0 = am[kh * 3 + kw] ^ s, x1 = am[kh * 3 + kw] ^ (s + 9ull * 0x0101010101010101ull),
                                   x2 = am[kh * 3 + kw] ^ (s + 18ull * 0x0101010101010101ull);
          const unsigned long long z0 = (x0 - 0x0101010101010101ull) & ~x0, z1 = (x1 - 0x0101010101010101ull) & ~x1,
                                   z2 = (x2 - 0x0101010101010101ull) & ~x2;
          if (!((z0 | z1 | z2) & 0x8080808080808080ull)) continue;
          float dv[8];
          ld8<T>((const T*)dy.p + vox_off(dy, b, to, h + 1 - kh, w + 1 - kw) + g * 8, dv);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            if (((x0 >> (8 * e)) & 0xffull) == 0) g_m[e] += dv[e];
            if (((x1 >> (8 * e)) & 0xffull) == 0) g_0[e] += dv[e];
            if (((x2 >> (8 * e)) & 0xffull) == 0) g_p[e] += dv[e];
          }
        }
    }
    const int t = to - 1;          // input plane to-1 has now seen all of its windows (planes to-2, to-1, to)
    if (t >= 0) {
      T* dst = (T*)dx.p + vox_off(dx, b, t, h, w) + g * 8;
      if (accumulate) {
        float o[8];
        ld8<T>(dst, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) g_m[e] += o[e];
      }
      st8<T>(dst, g_m);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { g_m[e] = g_0[e]; g_0[e] = g_p[e]; g_p[e] = 0.f; }
  }
}

// Same walk for bf16 with every load of a plane issued up front: the 9 argmax words AND the 9 gradient vectors of the
// in-plane neighbour windows are fetched unconditionally (they are L1 / L2 hits for 8 of 9 lanes), 18 independent loads
// per lane and plane, so a lane pays one memory latency per plane instead of two dependent ones
// (the conditional gradient loads of the form above left the kernel latency-bound at 1.4 TB/s).  Routing is branch-free:
// byte code - (kh*3+kw) is 0 / 9 / 18 for the temporal taps to-1 / to / to+1.
__global__ __launch_bounds__(256) void maxpool_bwd_k3s1_tw2_kernel(TView dy, const uint8_t* __restrict__ argmax, TView dx,
                                                                   int accumulate, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int G = dx.C >> 3;
  const uint32_t col = (uint32_t)(i / G);
  const int g = (int)(i - (long)col * G);
  const uint32_t r1 = fdiv(col, dx.dW);
  const int w = (int)(col - r1 * (uint32_t)dx.W);
  const uint32_t r2 = fdiv(r1, dx.dH);
  const int h = (int)(r1 - r2 * (uint32_t)dx.H);
  const int b = (int)r2;
  const int T_ = dx.T, H = dx.H, W = dx.W;
  // lane-relative addresses of the 9 windows (the lane's own voxel where the window does not exist: any valid address)
  const uint8_t* amp = argmax + ((((long)b * T_) * H + h) * W + w) * (long)dx.C + g * 8;
  const unsigned short* dyp = (const unsigned short*)dy.p + vox_off(dy, b, 0, h, w) + g * 8;
  const long am_plane = (long)H * W * dx.C, dy_plane = (long)H * W * dy.ld;
  uint32_t okmask = 0;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int ho = h + 1 - kh, wo = w + 1 - kw;
      okmask |= (((unsigned)ho < (unsigned)H && (unsigned)wo < (unsigned)W) ? 1u : 0u) << (kh * 3 + kw);
    }
  float g_m[8], g_0[8], g_p[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { g_m[e] = 0.f; g_0[e] = 0.f; g_p[e] = 0.f; }
  for (int to = 0; to <= T_; ++to) {
    if (to < T_) {
      unsigned long long am[9];
      uint4 dv[9];
#pragma unroll
      for (int s = 0; s < 9; ++s) {
        const int dvox = (1 - s / 3) * W + (1 - s % 3);
        const bool ok = (okmask >> s) & 1u;
        am[s] = *(const unsigned long long*)(amp + (ok ? dvox * dx.C : 0));
        dv[s] = *(const uint4*)(dyp + (ok ? dvox * dy.ld : 0));
      }
#pragma unroll
      for (int s = 0; s < 9; ++s) {
        const unsigned long long a = ((okmask >> s) & 1u) ? am[s] : ~0ull;
        const uint32_t alo = (uint32_t)a, ahi = (uint32_t)(a >> 32);
        const uint32_t q[4] = {dv[s].x, dv[s].y, dv[s].z, dv[s].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const uint32_t c = (((e < 4) ? alo : ahi) >> (8 * (e & 3))) & 0xffu;
          const int dlt = (int)c - s;
          const float v = (e & 1) ? __uint_as_float(q[e >> 1] & 0xffff0000u) : __uint_as_float(q[e >> 1] << 16);
          g_m[e] += (dlt == 0) ? v : 0.f;
          g_0[e] += (dlt == 9) ? v : 0.f;
          g_p[e] += (dlt == 18) ? v : 0.f;
        }
      }
      amp += am_plane;
      dyp += dy_plane;
    }
    const int t = to - 1;
    if (t >= 0) {
      unsigned short* dst = (unsigned short*)dx.p + vox_off(dx, b, t, h, w) + g * 8;
      if (accumulate) {
        float o[8];
        ld8<unsigned short>(dst, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) g_m[e] += o[e];
      }
      st8<unsigned short>(dst, g_m);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { g_m[e] = g_0[e]; g_0[e] = g_p[e]; g_p[e] = 0.f; }
  }
}

extern "C" int vinet_maxpool3d_bwd(const VinetPoolDesc* d, const VinetTensor* dy, const uint8_t* argmax,
                                   const VinetTensor* dx, int32_t accumulate, void* stream) {
  VN_CHECK_ARG(d && dy && dx && argmax && quad_ok(*dy, esize(d->dtype)) && quad_ok(*dx, esize(d->dtype)) &&
                   dx->C == dy->C && dx->B == dy->B, "maxpool3d_bwd: bad views");
  const PoolP p = make_poolp(d);
  const bool k3s1 = d->kT == 3 && d->kH == 3 && d->kW == 3 && d->sT == 1 && d->sH == 1 && d->sW == 1 && d->pT == 1 && d->pH == 1 &&
                    d->pW == 1 && dy->T == dx->T && dy->H == dx->H && dy->W == dx->W;
  if (k3s1 && dx->C % 8 == 0 && dx->ld % 8 == 0 && dy->ld % 8 == 0 && dx->sB % 8 == 0 && dy->sB % 8 == 0 &&
      ((uintptr_t)dx->ptr % 16) == 0 && ((uintptr_t)dy->ptr % 16) == 0 && ((uintptr_t)argmax % 8) == 0) {
    const long cols8 = (long)dx->B * dx->H * dx->W * (dx->C / 8);
    if (g_vinet_opt_pool_twalk != 3 && d->dtype == VINET_BF16 && (g_vinet_opt_pool_twalk >= 2 || (g_vinet_opt_pool_twalk && cols8 >= 65536)) &&
        (long)dx->H * dx->W * dx->C < (1l << 30) && (long)dx->H * dx->W * dy->ld < (1l << 30)) {   // 3: the conditional-load form (A/B)
      hipLaunchKernelGGL(maxpool_bwd_k3s1_tw2_kernel, dim3(ew_grid(cols8)), dim3(256), 0, (hipStream_t)stream, make_view(*dy), argmax,
                         make_view(*dx), accumulate, cols8);
      return vn_launch_status("maxpool3d_bwd(k3s1 tw2)");
    }
    if (g_vinet_opt_pool_twalk >= 2 || (g_vinet_opt_pool_twalk && cols8 >= 65536)) {   // 2: force (tests)   // enough columns to fill the chip
      DISPATCH_T(d->dtype, T, hipLaunchKernelGGL(maxpool_bwd_k3s1_twalk_kernel<T>, dim3(ew_grid(cols8)), dim3(256), 0,
                                                 (hipStream_t)stream, make_view(*dy), argmax, make_view(*dx), accumulate, cols8);)
      return vn_launch_status("maxpool3d_bwd(k3s1 twalk)");
    }
    const long total8 = view_voxels(*dx) * (dx->C / 8);
    DISPATCH_T(d->dtype, T, hipLaunchKernelGGL(maxpool_bwd_k3s1_kernel<T>, dim3(ew_grid(total8)), dim3(256), 0, (hipStream_t)stream,
                                               make_view(*dy), argmax, make_view(*dx), accumulate, total8);)
    return vn_launch_status("maxpool3d_bwd(k3s1)");
  }
  if (oct_ok(*dx) && oct_ok(*dy) && ((uintptr_t)argmax % 8) == 0) {
    const long total8 = view_voxels(*dx) * (dx->C / 8);
    if (g_vinet_opt_pool_blk && d->kT == 1 && d->sT == 1 && d->pT == 0 && d->kH == 3 && d->kW == 3 && d->sH == 2 && d->sW == 2 && d->pH == 1 &&
        d->pW == 1 && dy->T == dx->T && dy->H == (dx->H - 1) / 2 + 1 && dy->W == (dx->W - 1) / 2 + 1) {
      const int HB = (dx->H + 1) / 2, WB = (dx->W + 1) / 2;
      const long nthr = (long)dx->B * dx->T * HB * WB * (dx->C / 8);
      DISPATCH_T(d->dtype, T, hipLaunchKernelGGL(maxpool_bwd_k133s2_kernel<T>, dim3(ew_grid(nthr)), dim3(256), 0, (hipStream_t)stream,
                                                 make_view(*dy), argmax, make_view(*dx), accumulate, HB, WB, nthr);)
      return vn_launch_status("maxpool3d_bwd(k133s2)");
    }
    DISPATCH_T(d->dtype, T, hipLaunchKernelGGL(maxpool_bwd8_kernel<T>, dim3(ew_grid(total8)), dim3(256), 0, (hipStream_t)stream,
                                               p, make_view(*dy), argmax, make_view(*dx), accumulate, total8);)
    return vn_launch_status("maxpool3d_bwd8");
  }
  const long total = view_voxels(*dx) * (dx->C / 4);
  DISPATCH_T(d->dtype, T, hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3(ew_grid(total)), dim3(256), 0,
                                             (hipStream_t)stream, p, make_view(*dy), argmax, make_view(*dx), accumulate, total);)
  return vn_launch_status("maxpool3d_bwd");
}

// ============================================================================
// Upsample (1,2,2) trilinear, align_corners=False  (separable .25/.75 stencil)
// ============================================================================
template <typename T>
__global__ void upsample2x_kernel(TView x, TView y, long total) {
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

// ---- "skinny" weight gradient: a pointwise conv with at most 8 output channels (ViNet's 32 -> 1 head, model.py:279: the
// channel-padded dy has 8 columns, 7 of them exactly zero) over tens of millions of voxels is a per-channel reduction, not a
// GEMM: dw[n][c] = sum_v dy[v][n] * x[v][c].  The 64 x 64 MFMA tile spent 0.9 ms (0.8 TF/s) on it; here a lane owns 8 input
// channels of a strided share of the voxels with an 8 x 8 block of fp32 accumulators, lanes of equal channel group meet by
// wave shuffles, waves in LDS, and a workgroup adds its 8 x Cin block to dw with one atomic per element.
__global__ __launch_bounds__(256) void wgrad_skinny_kernel(TView x, TView dy, long nvox, int Kp, float* __restrict__ dw) {
  __shared__ float red[4][8][64];                  // [wave][group][n * 8 + e]
  const int G = x.C >> 3;                          // 1, 2, 4 or 8 groups of 8 input channels
  const int tid = threadIdx.x, g = tid % G, r = tid / G, R = 256 / G;
  float acc[8][8];
#pragma unroll
  for (int n = 0; n < 8; ++n)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[n][e] = 0.f;
  for (long v = (long)blockIdx.x * R + r; v < nvox; v += (long)gridDim.x * R) {
    float xv[8], gv[8];
    ld8<bf16_t>((const bf16_t*)x.p + vox_lin(x, v) + g * 8, xv);
    ld8<bf16_t>((const bf16_t*)dy.p + vox_lin(dy, v), gv);
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[n][e] = fmaf(gv[n], xv[e], acc[n][e]);
  }
  // lanes g, g + G, g + 2G, ... of a wave hold the same channel group
#pragma unroll
  for (int n = 0; n < 8; ++n)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = acc[n][e];
      for (int o = 32; o >= G; o >>= 1) a += __shfl_xor(a, o);
      acc[n][e] = a;
    }
  const int lane = tid & 63, wave = tid >> 6;
  if (lane < G) {
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[wave][lane][n * 8 + e] = acc[n][e];
  }
  __syncthreads();
  for (int i = tid; i < G * 64; i += 256) {
    const int gg = i >> 6, ne = i & 63, n = ne >> 3, e = ne & 7;
    const float a = red[0][gg][ne] + red[1][gg][ne] + red[2][gg][ne] + red[3][gg][ne];
    atomicAdd(dw + (long)n * Kp + gg * 8 + e, a);
  }
}

int g_vinet_opt_wgrad_skinny = 1;   // 0 = off, 2 = every eligible shape (tests)

bool vinet_wgrad_use_skinny(const VinetWgradDesc* d) {
  if (!g_vinet_opt_wgrad_skinny || d->dtype != VINET_BF16 || d->mode != VINET_CONV_GENERIC || d->ntaps != 1 || d->pre.scale || d->pre.relu ||
      d->bnb_z)
    return false;
  const int Cin = d->x.C;
  const bool shape = d->dy.C == 8 && (Cin == 8 || Cin == 16 || Cin == 32 || Cin == 64) && d->Kp >= Cin && d->sT == 1 && d->sH == 1 && d->sW == 1 &&
                     d->x.T == d->dy.T && d->x.H == d->dy.H && d->x.W == d->dy.W && oct_ok(d->x) && oct_ok(d->dy);
  if (!shape) return false;
  return g_vinet_opt_wgrad_skinny >= 2 || (long)d->dy.B * d->dy.T * d->dy.H * d->dy.W >= (1L << 20);
}

int vinet_launch_wgrad_skinny(const VinetWgradDesc* d, hipStream_t s) {
  // taps: a pointwise conv has one tap, (0, 0, 0, slice 0) -- nothing to read from the device-side table
  const long nvox = view_voxels(d->dy);
  const int R = 256 / (d->x.C / 8);
  long blocks = (nvox + R - 1) / R;
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(wgrad_skinny_kernel, dim3((unsigned)blocks), dim3(256), 0, s, make_view(d->x), make_view(d->dy), nvox, d->Kp, d->dw);
  return vn_launch_status("wgrad_skinny");
}

// ---- 1-D unfold (im2col along T) of a single-channel signal: SoundNet's first conv (model.py:751: Conv2d(1, 16, (64, 1),
// stride 2, padding 32)) has one input channel and 64 taps -- as a (k,1,1) conv its K axis would be 64 taps x 32 padded
// channels with one real column in 32.  Unfolded, y[b, m, c] = x[b, s*m - p + c, channel 0] (zero outside), it is a pointwise
// conv with 64 input channels: 0.58 GB written once per step instead of a 32x padded K loop in forward and weight gradient.
template <typename T>
__global__ __launch_bounds__(256) void unfold1d_kernel(TView x, TView y, int stride, int pad, long total8) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total8) return;
  const int G = y.C >> 3;
  const long row = i / G;                          // b * To + m
  const int g = (int)(i - row * G);
  const long b = row / y.T;
  const int m = (int)(row - b * y.T);
  const T* src = (const T*)x.p + b * x.sB;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const long pos = (long)m * stride - pad + g * 8 + e;
    v[e] = (pos >= 0 && pos < x.T) ? load1<T>(src + pos * x.ld) : 0.f;
  }
  st8<T>((T*)y.p + b * y.sB + (long)m * y.ld + g * 8, v);
}

extern "C" int vinet_unfold1d(const VinetTensor* x, const VinetTensor* y, int32_t dtype, int32_t stride, int32_t pad, void* stream) {
  VN_CHECK_ARG(x && y && (dtype == VINET_F32 || dtype == VINET_BF16) && x->ptr && y->ptr && x->B == y->B && x->H == 1 && x->W == 1 &&
                   y->H == 1 && y->W == 1 && x->C >= 1 && y->C % 8 == 0 && y->ld % 8 == 0 && y->sB % 8 == 0 &&
                   ((uintptr_t)y->ptr % 16) == 0 && stride >= 1 && pad >= 0 && y->T == (x->T + 2 * pad - y->C) / stride + 1,
               "unfold1d: bad views");
  const long total8 = (long)y->B * y->T * (y->C / 8);
  DISPATCH_T(dtype, T, hipLaunchKernelGGL(unfold1d_kernel<T>, dim3(ew_grid(total8)), dim3(256), 0, (hipStream_t)stream, make_view(*x),
                                          make_view(*y), stride, pad, total8);)
  return vn_launch_status("unfold1d");
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

template <typename T>
__global__ void upsample2x_bwd_kernel(TView dy, TView dx, int accumulate, long total) {
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
  if (accumulate) { const float4 o = ldq<T>(dst); g.x += o.x; g.y += o.y; g.z += o.z; g.w += o.w; }
  stq<T>(dst, g);
}

// 8-channel form of the backward gather (16-byte loads, index math per 16 instead of 8 bytes written)
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_bwd8_kernel(TView dy, TView dx, int accumulate, long total) {
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
  if (accumulate) {
    float o[8];
    ld8<T>(dst, o);
#pragma unroll
    for (int e = 0; e < 8; ++e) gr[e] += o[e];
  }
  st8<T>(dst, gr);
}

extern "C" int vinet_upsample2x_bwd(const VinetTensor* dy, const VinetTensor* dx, int32_t dtype, int32_t accumulate,
                                    void* stream) {
  VN_CHECK_ARG(dy && dx && quad_ok(*dy, esize(dtype)) && quad_ok(*dx, esize(dtype)) && dx->C == dy->C && dx->B == dy->B &&
                   dx->T == dy->T && dy->H == 2 * dx->H && dy->W == 2 * dx->W, "upsample2x_bwd: bad views");
  const long total = view_voxels(*dx) * (dx->C / 4);
  if (g_vinet_opt_up_blk && oct_ok(*dx) && oct_ok(*dy)) {
    const long total8 = view_voxels(*dx) * (dx->C / 8);
    DISPATCH_T(dtype, T, hipLaunchKernelGGL(upsample2x_bwd8_kernel<T>, dim3(ew_grid(total8)), dim3(256), 0, (hipStream_t)stream,
                                            make_view(*dy), make_view(*dx), accumulate, total8);)
    return vn_launch_status("upsample2x_bwd(8)");
  }
  DISPATCH_T(dtype, T, hipLaunchKernelGGL(upsample2x_bwd_kernel<T>, dim3(ew_grid(total)), dim3(256), 0,
                                          (hipStream_t)stream, make_view(*dy), make_view(*dx), accumulate, total);)
  return vn_launch_status("upsample2x_bwd");
}

__global__ void fill_f32_kernel(float* p, long n, float v) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
extern "C" int vinet_fill_f32(float* p, int64_t n, float value, void* stream) {
  VN_CHECK_ARG(p && n >= 0, "fill_f32: bad arguments");
  if (n == 0) return 0;
  int grid = ew_grid(n); if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(fill_f32_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, (long)n, value);
  return vn_launch_status("fill_f32");
}
