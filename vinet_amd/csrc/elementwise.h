// Shared device / host helpers of the HBM-bound kernel families (layout.hip, bn.hip, pool.hip, resample.hip): typed 4- and
// 8-channel access, pending-affine application, voxel decoding, launch sizing.  All of them work on channels-last views
// and move 4 or 8 channels (8 / 16 bytes bf16) per lane with channels fastest across lanes, so every wave touches whole
// contiguous rows.
#pragma once
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
// non-temporal forms (common.h: ld16_nt / st16_nt) for passes over tensors that have left every cache before they are read again
template <typename T> VN_DEV void ld8_nt(const T* p, float* f);
template <> VN_DEV void ld8_nt<float>(const float* p, float* f) {
  const uint4 a = ld16_nt(p), b = ld16_nt(p + 4);
  f[0] = __uint_as_float(a.x); f[1] = __uint_as_float(a.y); f[2] = __uint_as_float(a.z); f[3] = __uint_as_float(a.w);
  f[4] = __uint_as_float(b.x); f[5] = __uint_as_float(b.y); f[6] = __uint_as_float(b.z); f[7] = __uint_as_float(b.w);
}
template <> VN_DEV void ld8_nt<bf16_t>(const bf16_t* p, float* f) { unpack16<bf16_t>(ld16_nt(p), f); }
template <typename T> VN_DEV void st8_nt(T* p, const float* f);
template <> VN_DEV void st8_nt<float>(float* p, const float* f) {
  st16_nt(p, make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])));
  st16_nt(p + 4, make_uint4(__float_as_uint(f[4]), __float_as_uint(f[5]), __float_as_uint(f[6]), __float_as_uint(f[7])));
}
template <> VN_DEV void st8_nt<bf16_t>(bf16_t* p, const float* f) { st16_nt(p, pack16<bf16_t>(f)); }
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

