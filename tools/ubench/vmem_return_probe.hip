// Probe for the round-5 defect ("the run-to-run mismatch"): does a VALU instruction issued right behind
// `s_waitcnt vmcnt(0)` always see ALL of a global_load_dwordx4's return data?
//
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -shared -fPIC -o tools/ubench/libvmem_return_probe.so tools/ubench/vmem_return_probe.hip
//   python tools/reduce_race_repro.py --probe ...        (drives it beside the library's weight-gradient kernels)
//
// One workgroup of 256 threads per launch (the shape of the failing `channel_reduce8_kernel` launch).  Each lane prefills four
// fixed registers with a sentinel, loads 16 bytes into them, waits vmcnt(0), copies the four registers at once ("early"), idles ~150
// cycles, copies them again ("late").  Early != expected while late == expected means the wait returned before the data had landed.
#include <hip/hip_runtime.h>
#include <stdint.h>

struct ProbeOut {
  unsigned launches, early_bad, late_bad, pad;
  unsigned sample[12];      // first bad: lane, dword, early, late, expected, launch id
};

__device__ __forceinline__ uint32_t pat(uint32_t i, uint32_t d, uint32_t salt) { return (i * 2654435761u) ^ (d * 0x9e3779b9u) ^ salt ^ 0x5bd1e995u; }

__global__ __launch_bounds__(256) void fill_kernel(uint32_t* buf, uint32_t salt) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
  for (uint32_t d = 0; d < 4; ++d) buf[i * 4 + d] = pat(i, d, salt);
}

__global__ __launch_bounds__(256) void probe_kernel(const uint32_t* buf, uint32_t salt, uint32_t id, ProbeOut* out) {
  const uint32_t i = threadIdx.x;
  const uint32_t* addr = buf + i * 4;
  uint32_t e0, e1, e2, e3, l0, l1, l2, l3;
  const uint32_t sent = 0xdeadbeefu;
  asm volatile(
      "v_mov_b32 v20, %[sent]\n\tv_mov_b32 v21, %[sent]\n\tv_mov_b32 v22, %[sent]\n\tv_mov_b32 v23, %[sent]\n\t"
      "s_nop 4\n\t"
      "global_load_dwordx4 v[20:23], %[addr], off\n\t"
      "s_waitcnt vmcnt(0)\n\t"
      "v_mov_b32 %[e0], v20\n\tv_mov_b32 %[e1], v21\n\tv_mov_b32 %[e2], v22\n\tv_mov_b32 %[e3], v23\n\t"
      "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
      "v_mov_b32 %[l0], v20\n\tv_mov_b32 %[l1], v21\n\tv_mov_b32 %[l2], v22\n\tv_mov_b32 %[l3], v23\n\t"
      : [e0] "=&v"(e0), [e1] "=&v"(e1), [e2] "=&v"(e2), [e3] "=&v"(e3), [l0] "=&v"(l0), [l1] "=&v"(l1), [l2] "=&v"(l2), [l3] "=&v"(l3)
      : [addr] "v"(addr), [sent] "v"(sent)
      : "v20", "v21", "v22", "v23", "memory");
  const uint32_t ev[4] = {e0, e1, e2, e3}, lv[4] = {l0, l1, l2, l3};
  if (i == 0) atomicAdd(&out->launches, 1u);
#pragma unroll
  for (uint32_t d = 0; d < 4; ++d) {
    const uint32_t x = pat(i, d, salt);
    if (ev[d] != x) {
      if (atomicAdd(&out->early_bad, 1u) == 0) {
        out->sample[0] = i; out->sample[1] = d; out->sample[2] = ev[d]; out->sample[3] = lv[d]; out->sample[4] = x; out->sample[5] = id;
      }
    }
    if (lv[d] != x) atomicAdd(&out->late_bad, 1u);
  }
}

extern "C" int probe_fill(void* buf, uint32_t salt, void* stream) {
  hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (uint32_t*)buf, salt);
  return (int)hipGetLastError();
}
extern "C" int probe_launch(const void* buf, uint32_t salt, uint32_t id, void* out, void* stream) {
  hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)buf, salt, id, (ProbeOut*)out);
  return (int)hipGetLastError();
}

// ---- synthetic co-runners for tools/reduce_race_repro.py: what property of the weight-gradient kernels triggers the defect? ----------
// mode 0: fp32 global atomics (global_atomic_add_f32, no return), mode 1: plain stores to the same addresses, mode 2: returning atomics
__global__ __launch_bounds__(256) void corun_kernel(float* buf, unsigned n, int iters, int mode) {
  const unsigned t = blockIdx.x * 256 + threadIdx.x;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    const unsigned i = (t * 17u + (unsigned)it * 4099u) % n;
    if (mode == 0) __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)(buf + i), 1.0f);
    else if (mode == 1) buf[i] = (float)it;
    else acc += atomicAdd(buf + i, 1.0f);
  }
  if (acc == -1.f) buf[0] = acc;
}
// mode 3: LDS transpose reads (ds_read_b64_tr_b16, new on gfx950: every weight-gradient kernel of the library uses them, no forward
// conv does), mode 4: plain ds_read_b64 of the same addresses
typedef short s16x4_v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void corun_lds_kernel(float* buf, int iters, int mode) {
  __shared__ __attribute__((aligned(16))) short tile[16 * 1024];
  for (int i = threadIdx.x; i < 16 * 1024; i += 256) tile[i] = (short)(i * 7);
  __syncthreads();
  int acc = 0;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int it = 0; it < iters; ++it) {
    const short* src = tile + ((w * 4096 + ((lane + it) & 63) * 32 + ((it >> 6) & 3) * 8) & (16 * 1024 - 4));
    s16x4_v v;
    if (mode == 3) v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_v*)src);
    else v = *(const s16x4_v*)src;
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345678) buf[0] = (float)acc;
}
extern "C" int corun_lds_launch(void* buf, int iters, int mode, int blocks, void* stream) {
  hipLaunchKernelGGL(corun_lds_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)buf, iters, mode);
  return (int)hipGetLastError();
}
// mode 5: MFMAs on registers (no memory traffic), few registers: the victim can share a SIMD with these waves
typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void corun_mfma_kernel(float* buf, int iters) {
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x + i); b[i] = (short)(0x3f00 + i); }
  f32x4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
  }
  if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.f) buf[0] = c0[0];
}
extern "C" int corun_mfma_launch(void* buf, int iters, int blocks, void* stream) {
  hipLaunchKernelGGL(corun_mfma_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)buf, iters);
  return (int)hipGetLastError();
}

// ---- a VALU-only victim: packed fp32 adds / multiplies on registers, checked against the scalar result ----------------------------
// (the failing reduction kernel accumulates with v_pk_add_f32 / v_pk_mul_f32)
struct PkOut { unsigned launches, bad_add, bad_mul, pad; unsigned sample[8]; };
__global__ __launch_bounds__(256) void pk_probe_kernel(int iters, unsigned id, PkOut* out) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const unsigned t = threadIdx.x;
  unsigned bad_a = 0, bad_m = 0, first = 0xffffffffu;
  for (int it = 0; it < iters; ++it) {
    f32x2 a = {1.0f + (float)(t & 31) * 0.03125f + (float)it, 2.0f + (float)(t >> 3)};
    f32x2 b = {0.5f + (float)(it & 7), 3.0f - (float)(t & 7) * 0.125f};
    f32x2 s_, m_;
    asm volatile("v_pk_add_f32 %0, %2, %3\n\tv_pk_mul_f32 %1, %2, %3" : "=&v"(s_), "=&v"(m_) : "v"(a), "v"(b));
    const float s0 = a.x + b.x, s1 = a.y + b.y, m0 = a.x * b.x, m1 = a.y * b.y;
    if (s_.x != s0 || s_.y != s1) { ++bad_a; if (first == 0xffffffffu) first = it; }
    if (m_.x != m0 || m_.y != m1) { ++bad_m; if (first == 0xffffffffu) first = it; }
  }
  if (t == 0) atomicAdd(&out->launches, 1u);
  if (bad_a) atomicAdd(&out->bad_add, bad_a);
  if (bad_m) atomicAdd(&out->bad_mul, bad_m);
  if ((bad_a | bad_m) && atomicAdd(&out->pad, 1u) == 0) { out->sample[0] = t; out->sample[1] = first; out->sample[2] = id; }
}
// ---- instruction-sequence probes (mode): which piece of the failing kernel's code loses data beside MFMA waves? -------------------
//  1: v_pk_add_f32 with op_sel:[0,1] op_sel_hi:[1,0] (the halves of the second operand swapped: how the kernel accumulates the
//     voxels it loaded under a condition) on registers
//  2: the conditional block: s_and_saveexec_b64 / s_cbranch_execz / global_load_dwordx4 / s_waitcnt vmcnt(0) / 8 unpack VALUs /
//     s_or_b64 exec -- unpacked values against a second, unhurried load of the same 16 bytes
//  3: mode 2 followed by the swapped packed adds into accumulators (the whole pattern)
struct SeqOut { unsigned launches, bad, pad0, pad1; unsigned sample[12]; };
template <int MODE>
__global__ __launch_bounds__(256) void seq_probe_kernel(const unsigned* buf, int reps, unsigned id, SeqOut* out) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const unsigned t = threadIdx.x;
  unsigned bad = 0, first_rep = 0, first_slot = 0, got = 0, want = 0;
  for (int rep = 0; rep < reps; ++rep) {
    if (MODE == 1) {
      f32x2 acc = {1.0f + (float)rep, 2.0f + (float)(t & 15)};
      f32x2 b = {0.25f * (float)(t & 7), 8.0f + (float)(t >> 4)};
      f32x2 r_;
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(r_) : "v"(acc), "v"(b));
      const float w0 = acc.x + b.y, w1 = acc.y + b.x;
      if (r_.x != w0) { if (!bad) { first_rep = rep; first_slot = 0; got = __float_as_uint(r_.x); want = __float_as_uint(w0); } ++bad; }
      if (r_.y != w1) { if (!bad) { first_rep = rep; first_slot = 1; got = __float_as_uint(r_.y); want = __float_as_uint(w1); } ++bad; }
    } else {
      const unsigned* addr = buf + ((t + (unsigned)rep * 256u) & 16383u) * 4;
      unsigned e[8];
      const int lim = 1 << 30;
      asm volatile(
          "v_mov_b32 v20, 0\n\tv_mov_b32 v21, 0\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0\n\t"
          "v_mov_b32 %[e0], 0\n\tv_mov_b32 %[e1], 0\n\tv_mov_b32 %[e2], 0\n\tv_mov_b32 %[e3], 0\n\t"
          "v_mov_b32 %[e4], 0\n\tv_mov_b32 %[e5], 0\n\tv_mov_b32 %[e6], 0\n\tv_mov_b32 %[e7], 0\n\t"
          "v_cmp_gt_i32_e64 s[20:21], %[lim], %[idx]\n\t"
          "s_and_saveexec_b64 s[22:23], s[20:21]\n\t"
          "s_cbranch_execz 1f\n\t"
          "global_load_dwordx4 v[20:23], %[addr], off\n\t"
          "s_waitcnt vmcnt(0)\n\t"
          "v_lshlrev_b32 %[e0], 16, v20\n\tv_and_b32 %[e1], 0xffff0000, v20\n\t"
          "v_lshlrev_b32 %[e2], 16, v21\n\tv_and_b32 %[e3], 0xffff0000, v21\n\t"
          "v_lshlrev_b32 %[e4], 16, v22\n\tv_and_b32 %[e5], 0xffff0000, v22\n\t"
          "v_lshlrev_b32 %[e6], 16, v23\n\tv_and_b32 %[e7], 0xffff0000, v23\n"
          "1:\n\t"
          "s_or_b64 exec, exec, s[22:23]\n\t"
          : [e0] "=&v"(e[0]), [e1] "=&v"(e[1]), [e2] "=&v"(e[2]), [e3] "=&v"(e[3]), [e4] "=&v"(e[4]), [e5] "=&v"(e[5]), [e6] "=&v"(e[6]), [e7] "=&v"(e[7])
          : [addr] "v"(addr), [lim] "s"(lim), [idx] "v"(t)
          : "v20", "v21", "v22", "v23", "s20", "s21", "s22", "s23", "memory");
      if (MODE == 3) {
        // s[k] += e[k] the way the kernel does it: packed adds with swapped halves into a zero accumulator, then compare
        f32x2 a0 = {0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
        f32x2 p0 = {__uint_as_float(e[1]), __uint_as_float(e[0])}, p1 = {__uint_as_float(e[3]), __uint_as_float(e[2])};
        f32x2 p2 = {__uint_as_float(e[5]), __uint_as_float(e[4])}, p3 = {__uint_as_float(e[7]), __uint_as_float(e[6])};
        asm volatile("v_pk_add_f32 %0, %0, %4 op_sel:[0,1] op_sel_hi:[1,0]\n\tv_pk_add_f32 %1, %1, %5 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                     "v_pk_add_f32 %2, %2, %6 op_sel:[0,1] op_sel_hi:[1,0]\n\tv_pk_add_f32 %3, %3, %7 op_sel:[0,1] op_sel_hi:[1,0]"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3));
        e[0] = __float_as_uint(a0.x); e[1] = __float_as_uint(a0.y); e[2] = __float_as_uint(a1.x); e[3] = __float_as_uint(a1.y);
        e[4] = __float_as_uint(a2.x); e[5] = __float_as_uint(a2.y); e[6] = __float_as_uint(a3.x); e[7] = __float_as_uint(a3.y);
      }
      const volatile unsigned* va = addr;
      const unsigned w[4] = {va[0], va[1], va[2], va[3]};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        unsigned x = (k & 1) ? (w[k >> 1] & 0xffff0000u) : (w[k >> 1] << 16);
        if (MODE == 3) x = __float_as_uint(0.f + __uint_as_float(x));
        if (e[k] != x) { if (!bad) { first_rep = rep; first_slot = k; got = e[k]; want = x; } ++bad; }
      }
    }
  }
  if (t == 0) atomicAdd(&out->launches, 1u);
  if (bad) {
    if (atomicAdd(&out->bad, bad) == 0) {
      out->sample[0] = t; out->sample[1] = first_rep; out->sample[2] = first_slot; out->sample[3] = got; out->sample[4] = want; out->sample[5] = id;
    }
  }
}
// ---- which operand selections of the packed fp32 VALU instructions are affected?  r = a (op) b with the given op_sel / op_sel_hi;
//      expected value computed with scalar instructions from the same registers ------------------------------------------------------
#define PK_VARIANT(NAME, INSTR, MODS, LO, HI)                                                                                  \
  __global__ __launch_bounds__(256) void NAME(int reps, unsigned id, SeqOut* out) {                                            \
    typedef float f32x2 __attribute__((ext_vector_type(2)));                                                                   \
    const unsigned t = threadIdx.x;                                                                                            \
    unsigned bad = 0, fr = 0, fs = 0, got = 0, want = 0;                                                                       \
    for (int rep = 0; rep < reps; ++rep) {                                                                                     \
      f32x2 a = {1.0f + (float)rep, 2.0f + (float)(t & 15)};                                                                   \
      f32x2 b = {0.25f * (float)(1 + (t & 7)), 8.0f + (float)(t >> 4)};                                                        \
      f32x2 r_;                                                                                                                \
      asm volatile(INSTR " %0, %1, %2 " MODS : "=&v"(r_) : "v"(a), "v"(b));                                                     \
      const float w0 = (LO), w1 = (HI);                                                                                        \
      if (r_.x != w0) { if (!bad) { fr = rep; fs = 0; got = __float_as_uint(r_.x); want = __float_as_uint(w0); } ++bad; }       \
      if (r_.y != w1) { if (!bad) { fr = rep; fs = 1; got = __float_as_uint(r_.y); want = __float_as_uint(w1); } ++bad; }       \
    }                                                                                                                          \
    if (t == 0) atomicAdd(&out->launches, 1u);                                                                                 \
    if (bad && atomicAdd(&out->bad, bad) == 0) {                                                                               \
      out->sample[0] = t; out->sample[1] = fr; out->sample[2] = fs; out->sample[3] = got; out->sample[4] = want; out->sample[5] = id; \
    }                                                                                                                          \
  }
PK_VARIANT(pkv_add_plain, "v_pk_add_f32", "", a.x + b.x, a.y + b.y)
PK_VARIANT(pkv_add_swap1, "v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0]", a.x + b.y, a.y + b.x)
PK_VARIANT(pkv_add_bcast_lo1, "v_pk_add_f32", "op_sel_hi:[1,0]", a.x + b.x, a.y + b.x)
PK_VARIANT(pkv_add_bcast_hi1, "v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,1]", a.x + b.y, a.y + b.y)
PK_VARIANT(pkv_add_swap0, "v_pk_add_f32", "op_sel:[1,0] op_sel_hi:[0,1]", a.y + b.x, a.x + b.y)
PK_VARIANT(pkv_mul_swap1, "v_pk_mul_f32", "op_sel:[0,1] op_sel_hi:[1,0]", a.x * b.y, a.y * b.x)
PK_VARIANT(pkv_mul_bcast_lo0, "v_pk_mul_f32", "op_sel_hi:[0,1]", a.x * b.x, a.x * b.y)
PK_VARIANT(pkv_mul_hi0, "v_pk_mul_f32", "op_sel:[1,0]", a.y * b.x, a.y * b.y)
// the same within ONE kernel: a 512-thread workgroup whose waves 0..3 run the swapped packed add and whose waves 4..7 (one per SIMD,
// beside them) issue MFMAs -- does the co-runner have to be another kernel / queue, or just another wave of the SIMD?
__global__ __launch_bounds__(512) void pk_self_kernel(int reps, unsigned id, SeqOut* out) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const unsigned t = threadIdx.x;
  if (t >= 256) {
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + (t & 63) + i); b[i] = (short)(0x3f00 + i); }
    f32x4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int it = 0; it < reps * 2; ++it) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
    }
    if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.f) out->pad1 = 1;
    return;
  }
  unsigned bad = 0;
  for (int rep = 0; rep < reps; ++rep) {
    f32x2 a = {1.0f + (float)rep, 2.0f + (float)(t & 15)};
    f32x2 b = {0.25f * (float)(1 + (t & 7)), 8.0f + (float)(t >> 4)};
    f32x2 r_;
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(r_) : "v"(a), "v"(b));
    if (r_.x != a.x + b.y || r_.y != a.y + b.x) ++bad;
  }
  if (t == 0) atomicAdd(&out->launches, 1u);
  if (bad) atomicAdd(&out->bad, bad);
}
// ... and within one DISPATCH: odd workgroups issue MFMAs, even ones the swapped packed add (256 threads each, 1024 workgroups: they
// share CUs and SIMDs, but belong to the same kernel launch on the same queue)
// role: -1 = odd workgroups multiply, even ones add (one dispatch); 0 = every workgroup adds; 1 = every workgroup multiplies (two
// dispatches of the SAME kernel -- same code, same register allocation -- on two queues)
__global__ __launch_bounds__(256) void pk_grid_kernel(int reps, SeqOut* out, int role) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const unsigned t = threadIdx.x;
  if (role < 0 ? (blockIdx.x & 1) : role) {
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + (t & 63) + i); b[i] = (short)(0x3f00 + i); }
    f32x4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int it = 0; it < reps * 2; ++it) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
    }
    if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.f) out->pad1 = 1;
    return;
  }
  unsigned bad = 0;
  for (int rep = 0; rep < reps; ++rep) {
    f32x2 a = {1.0f + (float)rep, 2.0f + (float)(t & 15)};
    f32x2 b = {0.25f * (float)(1 + (t & 7)), 8.0f + (float)(t >> 4)};
    f32x2 r_;
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(r_) : "v"(a), "v"(b));
    if (r_.x != a.x + b.y || r_.y != a.y + b.x) ++bad;
  }
  if (t == 0) atomicAdd(&out->launches, 1u);
  if (bad) atomicAdd(&out->bad, bad);
}
extern "C" int pk_grid_launch2(int reps, int blocks, void* out, void* stream, int role) {
  hipLaunchKernelGGL(pk_grid_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reps, (SeqOut*)out, role);
  return (int)hipGetLastError();
}
extern "C" int pk_grid_launch(int reps, int blocks, void* out, void* stream) {
  hipLaunchKernelGGL(pk_grid_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reps, (SeqOut*)out, -1);
  return (int)hipGetLastError();
}
extern "C" int pk_self_launch(int reps, unsigned id, void* out, void* stream) {
  hipLaunchKernelGGL(pk_self_kernel, dim3(1), dim3(512), 0, (hipStream_t)stream, reps, id, (SeqOut*)out);
  return (int)hipGetLastError();
}
extern "C" int pk_variant_launch(int which, int reps, unsigned id, void* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  void (*k[8])(int, unsigned, SeqOut*) = {pkv_add_plain, pkv_add_swap1, pkv_add_bcast_lo1, pkv_add_bcast_hi1, pkv_add_swap0, pkv_mul_swap1,
                                          pkv_mul_bcast_lo0, pkv_mul_hi0};
  if (which < 0 || which > 7) return -1;
  hipLaunchKernelGGL(k[which], dim3(1), dim3(256), 0, s, reps, id, (SeqOut*)out);
  return (int)hipGetLastError();
}
extern "C" int seq_probe_launch(int mode, const void* buf, int reps, unsigned id, void* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (mode == 1) hipLaunchKernelGGL(seq_probe_kernel<1>, dim3(1), dim3(256), 0, s, (const unsigned*)buf, reps, id, (SeqOut*)out);
  else if (mode == 2) hipLaunchKernelGGL(seq_probe_kernel<2>, dim3(1), dim3(256), 0, s, (const unsigned*)buf, reps, id, (SeqOut*)out);
  else hipLaunchKernelGGL(seq_probe_kernel<3>, dim3(1), dim3(256), 0, s, (const unsigned*)buf, reps, id, (SeqOut*)out);
  return (int)hipGetLastError();
}
extern "C" int pk_probe_launch(int iters, unsigned id, void* out, void* stream) {
  hipLaunchKernelGGL(pk_probe_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, iters, id, (PkOut*)out);
  return (int)hipGetLastError();
}
// ---- the failing reduction, restated stand-alone (bn.hip: channel_reduce8_kernel<bf16, 0> on [nvox][C] bf16, one workgroup) with
//      switches to find the instruction pattern that fails beside MFMA waves of another kernel ------------------------------------------
//  V = 0: as in the library (loads under `if (ok[u])`, all four voxels unpacked to fp32 before use)
//  V = 1: unconditional clamped loads (the variant that never failed in the step)
//  V = 2: V = 0 without the sum of squares (no v_pk_mul_f32)
//  V = 3: V = 0 with 32-bit index arithmetic (no 64-bit VALU: v_cmp_*_i64, v_mad_u64_u32, v_lshl_add_u64)
template <int V>
__global__ __launch_bounds__(256) void reduce_victim_kernel(const unsigned short* x, int C, long nvox, float* partials) {
  const int G = C / 8, Gb = G < 256 ? G : 256, R = 256 / Gb, r = threadIdx.x / Gb, g0 = threadIdx.x % Gb;
  __shared__ float red[256 * 16];
  constexpr int U = 4;
  for (int g = g0; g < G; g += Gb) {
    float s[8], p[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; p[e] = 0.f; }
    if (r < R) {
      if (V == 3) {
        const int nv = (int)nvox;
        for (int vq = r; vq < nv; vq += R * U) {
          float xv[U][8];
          bool ok[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int v = vq + u * R;
            ok[u] = v < nv;
            if (ok[u]) {
              const uint4 q = *(const uint4*)(x + (unsigned)(v * C + g * 8));
              const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
              for (int e = 0; e < 8; ++e) xv[u][e] = __uint_as_float((e & 1) ? (w[e >> 1] & 0xffff0000u) : (w[e >> 1] << 16));
            }
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[e] += xv[u][e]; p[e] += xv[u][e] * xv[u][e]; }
          }
        }
      } else {
        for (long vq = r; vq < nvox; vq += (long)R * U) {
          float xv[U][8];
          bool ok[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const long v = vq + (long)u * R;
            ok[u] = v < nvox;
            if (V == 1 || ok[u]) {
              const long vc = ok[u] ? v : vq;
              const uint4 q = *(const uint4*)(x + vc * C + g * 8);
              const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
              for (int e = 0; e < 8; ++e) xv[u][e] = __uint_as_float((e & 1) ? (w[e >> 1] & 0xffff0000u) : (w[e >> 1] << 16));
            }
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[e] += xv[u][e]; if (V != 2) p[e] += xv[u][e] * xv[u][e]; }
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
      float* o = partials + g * 8;
      *(float4*)o = make_float4(s[0], s[1], s[2], s[3]);
      *(float4*)(o + 4) = make_float4(s[4], s[5], s[6], s[7]);
      *(float4*)(o + C) = make_float4(p[0], p[1], p[2], p[3]);
      *(float4*)(o + C + 4) = make_float4(p[4], p[5], p[6], p[7]);
    }
  }
}
extern "C" int reduce_victim_launch(int variant, const void* x, int C, long nvox, void* partials, void* stream) {
  const unsigned short* xp = (const unsigned short*)x;
  float* pp = (float*)partials;
  hipStream_t s = (hipStream_t)stream;
  if (variant == 0) hipLaunchKernelGGL(reduce_victim_kernel<0>, dim3(1), dim3(256), 0, s, xp, C, nvox, pp);
  else if (variant == 1) hipLaunchKernelGGL(reduce_victim_kernel<1>, dim3(1), dim3(256), 0, s, xp, C, nvox, pp);
  else if (variant == 2) hipLaunchKernelGGL(reduce_victim_kernel<2>, dim3(1), dim3(256), 0, s, xp, C, nvox, pp);
  else hipLaunchKernelGGL(reduce_victim_kernel<3>, dim3(1), dim3(256), 0, s, xp, C, nvox, pp);
  return (int)hipGetLastError();
}
extern "C" int corun_launch(void* buf, unsigned n, int iters, int mode, int blocks, void* stream) {
  hipLaunchKernelGGL(corun_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)buf, n, iters, mode);
  return (int)hipGetLastError();
}
