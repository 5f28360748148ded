// Shared device/host helpers for libvinet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "vinet_hip.h"

typedef uint16_t bf16_t;  // storage type; arithmetic is always fp32
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_v;
typedef __attribute__((ext_vector_type(4))) float f32x4_v;
typedef __attribute__((ext_vector_type(4))) short s16x4_v;

#define VN_DEV __device__ __forceinline__

// MI355X erratum found in round 5 ("the run-to-run mismatch"; tools/reduce_race_repro.py --pkvariants is the minimal
// reproduction): a packed fp32 VALU instruction (v_pk_add_f32 / v_pk_mul_f32 / ...) whose LOW result half takes the HIGH half of
// its second source (`op_sel:[x,1]`) reads that half as ZERO in lanes 48..63 now and then while a wave of ANOTHER kernel issues
// MFMAs on the same SIMD (another DISPATCH: MFMA waves of the same launch never trigger it).  hipcc emits the form wherever it
// allocated a register pair in swapped order.  Kernels in which it did
// are compiled without packed fp32 instructions (this attribute), and tests/test_isa_audit.py fails the CPU suite if the form
// appears anywhere in the built library.
#define VN_NO_PK_F32 __attribute__((target("no-packed-fp32-ops")))

VN_DEV float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// fp32 -> bf16, round-to-nearest-even, NaN stays NaN (same as aten): ONE instruction on gfx950 (v_cvt_pk_bf16_f32; the
// integer formulation -- NaN test, rounding add, shift -- is 6-8 VALU per element, and the pools, the upsample, the conv
// epilogues and the stem's streaming kernels are VALU-issue bound where they write bf16).  -DVINET_SOFT_BF16: the integer form.
#if defined(VINET_SOFT_BF16) || !defined(__HIP_DEVICE_COMPILE__)
VN_DEV bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
VN_DEV uint32_t pack2bf(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }
#else
VN_DEV uint32_t pack2bf(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
VN_DEV bf16_t f2bf(float f) { return (bf16_t)pack2bf(f, f); }
#endif

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
  static constexpr int EG = 4;  // elements per 16-byte group
  static constexpr int DT = VINET_F32;
};
template <> struct ElemTraits<bf16_t> {
  static constexpr int EG = 8;
  static constexpr int DT = VINET_BF16;
};

// 16 bytes of T <-> fp32 lanes
template <typename T> VN_DEV void unpack16(const uint4& u, float* f);
template <> VN_DEV void unpack16<float>(const uint4& u, float* f) {
  f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y); f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
}
template <> VN_DEV void unpack16<bf16_t>(const uint4& u, float* f) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
  f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
  f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}
template <typename T> VN_DEV uint4 pack16(const float* f);
template <> VN_DEV uint4 pack16<float>(const float* f) {
  return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}
template <> VN_DEV uint4 pack16<bf16_t>(const float* f) {
  return make_uint4(pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7]));
}

template <typename T> VN_DEV float load1(const T* p);
template <> VN_DEV float load1<float>(const float* p) { return *p; }
template <> VN_DEV float load1<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> VN_DEV void store1(T* p, float v);
template <> VN_DEV void store1<float>(float* p, float v) { *p = v; }
template <> VN_DEV void store1<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

// Division by a launch-invariant 31-bit integer as multiply-high + shift (exact for
// m < 2^31).  Runtime `/` and `%` cost ~40 instructions each on the GPU; the voxel decode
// m -> (b,t,h,w) needs three of them per row.
struct FastDiv { uint32_t magic, shift, d; };
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f; f.d = d;
  if (d <= 1) { f.magic = 0; f.shift = 0; return f; }
  uint32_t l = 0; while ((1u << l) < d) ++l;               // ceil(log2 d)
  f.magic = (uint32_t)((((unsigned long long)1 << (31 + l)) / d) + 1);
  f.shift = l - 1;
  return f;
}
VN_DEV uint32_t fdiv(uint32_t m, const FastDiv& f) { return f.d <= 1 ? m : (__umulhi(m, f.magic) >> f.shift); }
// m -> (b, to, ho, wo) for an iteration space [B][To][Ho][Wo]
VN_DEV void decode_m(int m, const FastDiv& dW, const FastDiv& dH, const FastDiv& dT, int& b, int& to, int& ho, int& wo) {
  const uint32_t t1 = fdiv((uint32_t)m, dW);
  wo = m - (int)t1 * (int)dW.d;
  const uint32_t t2 = fdiv(t1, dH);
  ho = (int)t1 - (int)t2 * (int)dH.d;
  const uint32_t bb = fdiv(t2, dT);
  to = (int)t2 - (int)bb * (int)dT.d;
  b = (int)bb;
}

// Device-side tensor view (mirror of VinetTensor with typed helpers).
struct TView {
  char* p;
  int B, T, H, W, C, ld;
  long sB;
  int linear;            // voxel v lives at element offset v*ld (full-extent view, possibly a channel slice)
  FastDiv dW, dH, dT;    // voxel index -> (b,t,h,w) without runtime division
  FastDiv dQ;            // flat (voxel, 4-channel group) index -> voxel
};
static inline TView make_view(const VinetTensor& t) {
  TView v;
  v.p = (char*)t.ptr; v.B = t.B; v.T = t.T; v.H = t.H; v.W = t.W; v.C = t.C; v.ld = t.ld; v.sB = t.sB;
  v.linear = (t.sB == (int64_t)t.T * t.H * t.W * t.ld) ? 1 : 0;
  v.dW = make_fastdiv((uint32_t)t.W); v.dH = make_fastdiv((uint32_t)t.H); v.dT = make_fastdiv((uint32_t)t.T);
  v.dQ = make_fastdiv((uint32_t)(t.C / 4 > 0 ? t.C / 4 : 1));
  return v;
}
// element offset of voxel (b,t,h,w), channel 0
VN_DEV long vox_off(const TView& v, int b, int t, int h, int w) {
  return (long)b * v.sB + ((long)(t * v.H + h) * v.W + w) * (long)v.ld;
}

struct Affine {
  const float* scale;
  const float* shift;
  int relu;
};
static inline Affine make_affine(const VinetAffine& a) {
  Affine r; r.scale = a.scale; r.shift = a.shift; r.relu = a.relu; return r;
}

// XCD-aware bijective block remap: consecutive logical ids land on the same
// XCD (blockIdx b runs on XCD b % 8), so neighbouring tiles share an L2.
VN_DEV int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, loc = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// Wave-uniform read of a small read-only table through the scalar cache (s_load):
// a plain global load of a uniform address compiles to a VECTOR load followed by
// s_waitcnt vmcnt(0), which drains every LDS-DMA / prefetch in flight.
VN_DEV int4 load_tap(const int4* taps, int i) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(VINET_TAP_VECTOR_LOAD)
  typedef const __attribute__((address_space(4))) int* const_int_ptr;
  const_int_ptr p = (const_int_ptr)(taps + i);
  return make_int4(p[0], p[1], p[2], p[3]);
#else
  return taps[i];
#endif
}

// LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane, lane-linear on the LDS side) issued from INLINE ASM, for kernels that
// read the staged tiles with TRANSPOSING LDS reads (ds_read_b64_tr_b16, the weight gradients).  Issued through the builtin,
// hipcc knows the instruction as a store to LDS that may alias any LDS load it cannot disambiguate -- and it cannot for the
// transpose-read builtin -- so it puts `s_waitcnt vmcnt(0)` in front of every group of reads: the counted vmcnt(n) pipelines
// of conv_wgrad_pp / conv_wgrad_dma / conv_wgrad_tf drained completely in every phase (round 6, found in the disassembly;
// the conv kernels' plain ds_read_b128 do not trigger it).  From asm the compiler sees neither the LDS store nor M0; the
// waits are the kernels' own counted ones, as designed.  `lds_dst` must be wave-uniform (an SGPR).  M0 is written behind the
// compiler's back (clobber declared; clang warns that M0 is reserved, silenced here): a kernel that uses this helper issues ALL
// its LDS-DMAs through it and uses nothing else that lives in M0 (no builtin LDS-DMA, no s_movrel, no GWS / sendmsg) -- true of
// conv_wgrad_pp, conv_wgrad_dma and the PRE instantiations of conv_dma / conv_dma3 / conv_ht, which select it per template.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
VN_DEV void lds_dma16_asm(const char* src, char* lds_dst) {
  const unsigned d = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_dst;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(d) : "memory", "m0");
}
#pragma clang diagnostic pop

// 16-byte non-temporal load / store (the compiler merges the four dwords into one dwordx4 with the nt bit): for streaming passes over
// GB-sized tensors that nothing re-reads before they have left every cache
VN_DEV uint4 ld16_nt(const void* p) {
  const uint32_t* q = (const uint32_t*)p;
  return make_uint4(__builtin_nontemporal_load(q), __builtin_nontemporal_load(q + 1), __builtin_nontemporal_load(q + 2), __builtin_nontemporal_load(q + 3));
}
VN_DEV void st16_nt(void* p, uint4 v) {
  uint32_t* q = (uint32_t*)p;
  __builtin_nontemporal_store(v.x, q); __builtin_nontemporal_store(v.y, q + 1); __builtin_nontemporal_store(v.z, q + 2); __builtin_nontemporal_store(v.w, q + 3);
}

// MFMA with the accumulator PINNED in the AGPR file.  With the builtin, hipcc keeps the
// accumulators of an address-heavy pipelined loop in VGPRs and spills/reloads all of them
// through AGPRs every iteration (2 x 96 v_accvgpr moves per 24 MFMAs measured); an asm
// operand with the "a" constraint cannot leave the accumulator file.  Accumulate chains need
// no wait states; callers must call mfma_drain() before anything else reads the result.
VN_DEV void mfma_bf16_acc(f32x4_v& acc, const bf16x8_v& a, const bf16x8_v& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// the same product TRANSPOSED (operands swapped: rows of the result tile = rows of b): a lane then holds four consecutive
// columns of ONE row of a x b^T -- four consecutive channels of a voxel in the conv kernels (conv_igemm.h: conv_epilogue)
VN_DEV void mfma_bf16_acc_t(f32x4_v& acc, const bf16x8_v& a, const bf16x8_v& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(b), "v"(a));
}
VN_DEV uint32_t cvt_pk_bf16_f32(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
// wait states between the last MFMA (8 passes) and a non-MFMA reader of its result
VN_DEV void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); }
// wait states between a VALU write of an MFMA A/B operand and the MFMA that reads it
VN_DEV void valu_to_mfma_pad() { asm volatile("s_nop 1"); }

// relu(x*s + h) on a packed bf16 pair in 5 VALU instructions (unpack 2, v_pk_fma_f32, v_cvt_pk_bf16_f32,
// v_pk_max_i16 against 0) instead of 7 (two fma, two max): as signed 16-bit integers every negative bf16
// (including -0 and a sign-bit NaN) is < 0, every non-negative one is >= 0 and ordered like its value.
// Out-of-range activations of the PRE kernels are fetched from a page of 0xFFFF (NEGATIVE quiet NaN): the NaN
// survives the fma and the conversion with its sign and the integer max turns it into +0 -- the same
// "padding needs no mask" property the float max(NaN, 0) = 0 gave.  (-DVINET_PRE_FLOAT_MAX: the 7-instruction form.)
typedef __attribute__((ext_vector_type(2))) float f32x2_v;
VN_DEV uint32_t pre_relu_pair(uint32_t u, f32x2_v s, f32x2_v h) {
#ifdef VINET_PRE_FLOAT_MAX
  const float lo = fmaxf(fmaf(__uint_as_float(u << 16), s.x, h.x), 0.f);
  const float hi = fmaxf(fmaf(__uint_as_float(u & 0xffff0000u), s.y, h.y), 0.f);
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
#else
  f32x2_v x, y;
  x.x = __uint_as_float(u << 16);
  x.y = __uint_as_float(u & 0xffff0000u);
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(y) : "v"(x), "v"(s), "v"(h));
  uint32_t p, r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p) : "v"(y.x), "v"(y.y));
  asm("v_pk_max_i16 %0, %1, 0" : "=v"(r) : "v"(p));
  return r;
#endif
}

VN_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
VN_DEV double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- host side -----------------------------------------------------------
void vinet_set_error(const char* fmt, ...);
#define VN_CHECK_ARG(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      vinet_set_error(__VA_ARGS__);        \
      return -1;                           \
    }                                      \
  } while (0)

static inline int vn_launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    vinet_set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
// CUs a persistent weight-gradient launch may occupy (VinetWgradDesc::max_cus; 0 = all 256)
static inline int vn_wgrad_cus(const VinetWgradDesc* d) {
  return d->max_cus <= 0 ? 256 : (d->max_cus < 8 ? 8 : (d->max_cus > 256 ? 256 : d->max_cus));
}
// fp32 tensors in memory: the exact fp32-MFMA path and its split-bf16 arithmetic form (VINET_F32S)
static inline bool vn_f32_storage(int dt) { return dt == VINET_F32 || dt == VINET_F32S; }
static inline int vn_div_up(long a, long b) { return (int)((a + b - 1) / b); }
// `overlap`: a read-only conv input may be an OVERLAPPED view (ld < C): consecutive W positions share
// channels.  The folded RGB stem uses it (position = 2 pixels, "channels" = 8 pixels x 4).
static inline bool vn_tensor_ok(const VinetTensor& t, int eg, bool overlap = false) {
  return t.ptr && t.B > 0 && t.T > 0 && t.H > 0 && t.W > 0 && t.C > 0 && (overlap || t.ld >= t.C) && t.ld > 0 && (t.ld % eg) == 0 &&
         (t.C % eg) == 0 && (((uintptr_t)t.ptr) & 15) == 0 && (t.sB % eg) == 0;
}
