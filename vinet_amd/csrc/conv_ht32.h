// conv_ht.h's halo-tile convolution on v_mfma_f32_32x32x16_bf16 (round 3).
//
// Why: v_mfma_f32_16x16x32_bf16 issues at ~20 cycles per SIMD when two waves share the matrix pipe (MI355X_MICROARCH.md:
// "~5 cyc" per CU against "~8 cyc" for 32x32x16 with twice the work), i.e. it tops out at 80 % of the bf16 peak, and under
// MFMA load the chip clocks at ~1.7 GHz: s_memtime stamps of conv_ht_kernel<6,16,3> on the 480 -> 192 5x3x3 decoder conv
// show 2030 cycles per K step and workgroup for 2 x 48 x 20 = 1920 cycles of matrix-pipe time -- the K loop IS pipe-bound,
// on the slower opcode.  The 32x32x16 form does the same K step in 2 x 24 x 32 = 1536.
//
// Same structure as conv_ht_kernel (spatial halo image staged once per (temporal tap, 64-channel chunk), weight tiles
// through a ring, one raw barrier per K step, zero-page padding, exact vmcnt).  What changes:
//   * wave tile = 64 positions x BN columns as 2 x (BN / 32) accumulator tiles of 32 x 32 (16 registers each, pinned in
//     AGPRs); WEIGHTS are the A operand, so a lane holds D[n = 8 (r / 4) + 4 (lane / 32) + r % 4][m = lane % 32]: four
//     groups of four consecutive channels of one position.
//   * a fragment is 32 rows x 16 k: lane -> row lane % 32, 16-byte chunk 2 s + lane / 32 of the 128-byte row (s = 0..3).
//     ds_read_b128 serves lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32) per LDS cycle: 16 rows of equal parity
//     mix, so the chunk swizzle is XOR ((row >> 1) & 7) -- conflict-free for 32 consecutive rows at ANY alignment (the nine
//     spatial taps shift the activation fragments by +-1 and +-(TW + 2) positions).  Applied on the DMA source address
//     (a piece of 8 rows starts at a multiple of 8: (row >> 1) & 7 = (piece & 1) << 2 | (row & 7) >> 1) and on the read.
//   * epilogue: conv_epilogue32 below -- no LDS transposition: v_cvt_pk_bf16_f32 pairs, one v_permlane32_swap per dword
//     between the two position tiles, 16-byte stores of 8 consecutive channels; BN partial sums by two DPP row rotations
//     and a 32-way LDS table.
#pragma once
#include "conv_ht.h"

typedef __attribute__((ext_vector_type(16))) float f32x16_v;

VN_DEV void mfma32_bf16_acc(f32x16_v& acc, const bf16x8_v& a, const bf16x8_v& b) {
#ifndef VINET_HT32_ACC_VGPR
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
#else
  // (tuning build -DVINET_HT32_ACC_VGPR: accumulators as architectural VGPR tuples -- the register file is unified, so the
  // allocator then has all 256 registers of the wave instead of 128 + 128; measured: hipcc fills them and spills 430 bytes)
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#endif
}
// wait states between the last 32x32x16 MFMA (16 passes) and a non-MFMA reader of its result
VN_DEV void mfma32_drain() { asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory"); }

// upper 32 lanes of x <-> lower 32 lanes of y (operands come out of inline-asm VALU code: 2 wait states by hand)
VN_DEV void permlane32_swap(uint32_t& x, uint32_t& y) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
}
VN_DEV void permlane32_swap_f(float& x, float& y) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
}

template <int NT32>
constexpr int conv_epi32_bytes() { return 32 * NT32 * 32 * 2 * 4; }     // 32 partial rows x BN columns x (sum, sum^2)

// Epilogue of a 4-wave workgroup whose waves are stacked along M, each holding acc[2][NT32] tiles of 32 x 32 (see above).
// Rows: wave-local 16-row groups gi = 2 I + (lane & 31) / 16 of position tile I, mapped to voxels by EpiRows exactly as in
// conv_epilogue (conv_igemm.h).  Same semantics: per-channel affine, BN partial sums per statistics row `tile_m`, ReLU /
// sigmoid, accumulate, any placement, bf16 or fp32 output.
template <int NT32>
VN_DEV void conv_epilogue32(const ConvArgs& a, f32x16_v (&acc)[2][NT32], char* smem, int tile_m, int tile_n, const EpiRows er) {
  constexpr int BN = NT32 * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pcol = lane & 31, h = lane >> 5, p16 = pcol & 15, half = pcol >> 4;
  float* red = (float*)smem;                       // [32 partial rows][BN][2]
  const int n_wave = tile_n * BN;
  const bool do_stats = a.stats != nullptr;
  const bool relu = a.act == VINET_ACT_RELU, sigm = a.act == VINET_ACT_SIGMOID;
  const float relu_floor = relu ? 0.f : -INFINITY;
  auto group_m0 = [&](int i) { return er.m0 + (i / er.ipr) * er.rstride + (i % er.ipr) * 16; };
  auto group_rows = [&](int i) { return ((i / er.ipr) < er.nrows && (i % er.ipr) < er.ncols) ? 16 : 0; };
  auto voxel_off = [&](int m) {
    if (a.y_linear) return (long)m * a.ldy;
    int b, to, ho, wo;
    decode_m(m, a.dW, a.dH, a.dT, b, to, ho, wo);
    return (long)b * a.sBy + ((long)((to * a.omT + a.ooT) * a.yH + (ho * a.omH + a.ooH)) * a.yW + (wo * a.omW + a.ooW)) * (long)a.ldy;
  };
  const bool has_aff = a.out_scale != nullptr || a.out_shift != nullptr;
  // is my position of tile I inside the iteration space (statistics masks); my store target after the swap is tile h
  const bool rok0 = p16 < group_rows(half), rok1 = p16 < group_rows(2 + half);
  const bool vok = h ? rok1 : rok0;
  const long voff = vok ? voxel_off(group_m0(2 * h + half) + p16) : 0;
  const bool fast8 = a.vec_ok && !a.out_f32 && !sigm && (a.N & 7) == 0 && (a.ldy & 7) == 0 && (a.sBy & 7) == 0 && (((uintptr_t)a.y) & 15) == 0;

#pragma unroll
  for (int J = 0; J < NT32; ++J) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n4 = n_wave + J * 32 + 8 * g + 4 * h;        // my four channels before the swap
      const int n8 = n_wave + J * 32 + 8 * g;                // the eight channels I store after it
      float sc[4], sh[4];
      bool cok[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        cok[r] = n4 + r < a.Nw;
        sc[r] = (a.out_scale && cok[r]) ? a.out_scale[n4 + r] : 1.f;
        sh[r] = (a.out_shift && cok[r]) ? a.out_shift[n4 + r] : 0.f;
      }
      uint4 old = make_uint4(0, 0, 0, 0);
      if (fast8 && a.accumulate && vok && n8 < a.N) old = *(const uint4*)((const bf16_t*)a.y + voff + n8);
      float v0[4], v1[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {      // (plain reads: an "a"-constrained asm on one element of a 16-register tuple makes hipcc copy the whole tile out and back)
        v0[r] = acc[0][J][4 * g + r];
        v1[r] = acc[1][J][4 * g + r];
      }
      if (has_aff) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { v0[r] = fmaf(v0[r], sc[r], sh[r]); v1[r] = fmaf(v1[r], sc[r], sh[r]); }
      }
      if (do_stats) {
        float ss[4], qq[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float t0 = (rok0 && cok[r]) ? v0[r] : 0.f, t1 = (rok1 && cok[r]) ? v1[r] : 0.f;
          ss[r] = t0 + t1;
          qq[r] = fmaf(t1, t1, t0 * t0);
        }
        // two DPP rotations leave lanes p16 = 0..3 of every 16-lane row with four distinct partial sums: 32 partial rows
        // per workgroup in the LDS table, added up by the last phase below
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ss[r] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ss[r]), 0x128, 0xf, 0xf, false));
          ss[r] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ss[r]), 0x124, 0xf, 0xf, false));
          qq[r] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, qq[r]), 0x128, 0xf, 0xf, false));
          qq[r] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, qq[r]), 0x124, 0xf, 0xf, false));
        }
        if (p16 < 4) {
          const int part = (wave * 2 + half) * 4 + p16;
          float* dst = red + ((long)part * BN + (J * 32 + 8 * g + 4 * h)) * 2;
          *(float4*)dst = make_float4(ss[0], qq[0], ss[1], qq[1]);
          *(float4*)(dst + 4) = make_float4(ss[2], qq[2], ss[3], qq[3]);
        }
      }
      if (fast8) {
        if (!a.accumulate) {
          uint32_t x0 = cvt_pk_bf16_f32(v0[0], v0[1]), x1 = cvt_pk_bf16_f32(v0[2], v0[3]);
          uint32_t y0 = cvt_pk_bf16_f32(v1[0], v1[1]), y1 = cvt_pk_bf16_f32(v1[2], v1[3]);
          if (relu) { x0 = pk_relu_bf16(x0); x1 = pk_relu_bf16(x1); y0 = pk_relu_bf16(y0); y1 = pk_relu_bf16(y1); }
          permlane32_swap(x0, y0);
          permlane32_swap(x1, y1);
          if (vok && n8 < a.N) *(uint4*)((bf16_t*)a.y + voff + n8) = make_uint4(x0, x1, y0, y1);
        } else {
          // y += result: ONE rounding, of old + new in fp32
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v0[r] = fmaxf(v0[r], relu_floor); v1[r] = fmaxf(v1[r], relu_floor);
            permlane32_swap_f(v0[r], v1[r]);
          }
          const uint32_t ow[4] = {old.x, old.y, old.z, old.w};
          uint32_t o[4];
          o[0] = cvt_pk_bf16_f32(v0[0] + __uint_as_float(ow[0] << 16), v0[1] + __uint_as_float(ow[0] & 0xffff0000u));
          o[1] = cvt_pk_bf16_f32(v0[2] + __uint_as_float(ow[1] << 16), v0[3] + __uint_as_float(ow[1] & 0xffff0000u));
          o[2] = cvt_pk_bf16_f32(v1[0] + __uint_as_float(ow[2] << 16), v1[1] + __uint_as_float(ow[2] & 0xffff0000u));
          o[3] = cvt_pk_bf16_f32(v1[2] + __uint_as_float(ow[3] << 16), v1[3] + __uint_as_float(ow[3] & 0xffff0000u));
          if (vok && n8 < a.N) *(uint4*)((bf16_t*)a.y + voff + n8) = make_uint4(o[0], o[1], o[2], o[3]);
        }
      } else {
        // general form (fp32 output, sigmoid, channel counts that are no multiple of 8): element by element from my own
        // registers -- no swap: I store my four channels of BOTH position tiles
#pragma unroll
        for (int I = 0; I < 2; ++I) {
          const int gi = 2 * I + half;
          if (!(p16 < group_rows(gi))) continue;
          const long off = voxel_off(group_m0(gi) + p16) + n4;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (n4 + r >= a.N) continue;
            float o = fmaxf(I ? v1[r] : v0[r], relu_floor);
            if (sigm) o = 1.f / (1.f + __expf(-o));
            if (a.out_f32) {
              float* dst = (float*)a.y + off + r;
              *dst = a.accumulate ? *dst + o : o;
            } else {
              bf16_t* dst = (bf16_t*)a.y + off + r;
              *dst = f2bf(a.accumulate ? bf2f(*dst) + o : o);
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // one channel group at a time: hoisted accumulator reads of later groups spill
    }
  }
  if (do_stats) {
    __syncthreads();    // the 32 partial rows of every column are in the table
    if (tid < BN) {
      const int n = n_wave + tid;
      if (n < a.N) {
        float ss = 0.f, qq = 0.f;
#pragma unroll 8
        for (int part = 0; part < 32; ++part) {
          const float2 v = *(const float2*)(red + ((long)part * BN + tid) * 2);
          ss += v.x; qq += v.y;
        }
        a.stats[((long)tile_m * 2 + 0) * a.N + n] = ss;
        a.stats[((long)tile_m * 2 + 1) * a.N + n] = qq;
      }
    }
  }
}

template <int NT32, int TW, int BSLOTS, bool TM = false, bool PRE = false>
struct ConvHt32Cfg {
  static constexpr int THREADS = 256, BM = 256, TR = TM ? 4 : BM / TW, HW = TM ? 64 : TW + 2, HR = TR + 2;
  static constexpr int NPOS = HR * HW;                  // halo positions
  static constexpr int HPIECES = (NPOS + 7) / 8;        // DMA pieces of 8 positions x 128 B
  static constexpr int HL = (HPIECES + 3) / 4;          // halo DMAs per wave
  static constexpr int HALO_BYTES = HL * 4 * 1024;
  static constexpr int BN = NT32 * 32;
  static constexpr int BL = NT32;                       // weight DMAs per wave and K step (BN / 8 pieces over 4 waves)
  static constexpr int BSLOT_BYTES = BN * 128;
  static constexpr int KLOOP_BYTES = HALO_BYTES + BSLOTS * BSLOT_BYTES;   // PRE: scale[Kp], shift[Kp] (fp32) behind it
  static constexpr int EPI_BYTES = conv_epi32_bytes<NT32>();
  static int smem_bytes(int Kp) {
    const int k = KLOOP_BYTES + (PRE ? 2 * Kp * 4 : 0);
    return k > EPI_BYTES ? k : EPI_BYTES;
  }
  static_assert(TW == 32 || TW == 16, "shapes");
  static_assert(BSLOTS >= 2 && BL * (BSLOTS - 2) <= 63, "vmcnt immediate range");
};

template <int NT32, int TW, int BSLOTS, bool TM, bool PRE>
__global__ __launch_bounds__(256, 2) void conv_ht32_kernel(const ConvArgs a) {
  using Cfg = ConvHt32Cfg<NT32, TW, BSLOTS, TM, PRE>;
  constexpr int HW = Cfg::HW, TR = Cfg::TR, HL = Cfg::HL, BL = Cfg::BL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const halo = smem;
  char* const bring = smem + Cfg::HALO_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* zero = (const char*)g_vinet_zero_page;

#ifdef VINET_CONV_TIMING
  const unsigned long long tm0 = __builtin_amdgcn_s_memtime();
  unsigned long long tm_halo = 0, tm_h0 = 0;
#endif
#ifdef VINET_CONV_TIMING2
  unsigned long long tq_a = 0, tq_w = 0, tq_i = 0, tq_b = 0;
#endif
  // ---- workgroup -> (column tile, spatial tile, frame): as conv_ht_kernel -----------------------------------------------
  const uint32_t wg = (uint32_t)xcd_remap(blockIdx.x, gridDim.x);
  const uint32_t sp = fdiv(wg, a.ht_dN);                      // spatial tile index = statistics row
  const int tile_n = (int)(wg - sp * (uint32_t)a.tilesN);
  const uint32_t q1 = fdiv(sp, a.ht_dW);
  const int tw_i = (int)(sp - q1 * (uint32_t)a.ht_tilesW);
  const uint32_t frame = fdiv(q1, a.ht_dH);
  const int th_i = (int)(q1 - frame * (uint32_t)a.ht_tilesH);
  const uint32_t bb = fdiv(frame, a.ht_dTo);
  const int to = TM ? th_i * 4 : (int)(frame - bb * (uint32_t)a.To), b = (int)bb;
  const int h0 = th_i * TR, w0 = tw_i * TW;
  const int p0 = tw_i * 64, HWtot = a.Hi * a.Wi;     // (temporal mode: first position of the tile, positions per frame)

  // ---- this lane's DMA role: row (lane >> 3) of an 8-row piece, LDS slot (lane & 7), source chunk slot ^ swizzle(row).
  //      Every piece this wave stages is piece number 4 j + wave: its parity is the wave's -------------------------------------
  const int prow = lane >> 3;
  const int src_chunk = (lane & 7) ^ (((wave & 1) << 2) | (prow >> 1));
  int hal_off[HL];            // element offset of this lane's halo position inside a frame (+ its chunk); 0 when out of range
  unsigned hal_ok = 0;
#pragma unroll
  for (int j = 0; j < HL; ++j) {
    const int p = (j * 4 + wave) * 8 + prow;
    const int hr = p / HW, hc = p - hr * HW;
    if constexpr (TM) {     // halo row = frame t0 - 1 + hr (added at issue time: uniform per piece), column = position p0 + hc
      const bool ok = (p < Cfg::NPOS) & (p0 + hc < HWtot);
      hal_off[j] = ok ? (p0 + hc) * a.ldx + src_chunk * 8 : 0;
      hal_ok |= (unsigned)ok << j;
    } else {
      const int h = h0 - 1 + hr, w = w0 - 1 + hc;
      const bool ok = (p < Cfg::NPOS) & ((unsigned)h < (unsigned)a.Hi) & ((unsigned)w < (unsigned)a.Wi);
      hal_off[j] = ok ? (h * a.Wi + w) * a.ldx + src_chunk * 8 : 0;
      hal_ok |= (unsigned)ok << j;
    }
  }
  const char* const xb = a.x + (long)b * a.sBx * 2;
  const long frame_elems = (long)a.Hi * a.Wi * a.ldx;
  long b_off[BL];
  unsigned b_ok[BL];
#pragma unroll
  for (int j = 0; j < BL; ++j) {
    const int n = (j * 4 + wave) * 8 + prow;
    const int nn = tile_n * Cfg::BN + n;
    b_ok[j] = (unsigned)(nn < a.Nw);
    b_off[j] = ((long)(b_ok[j] ? nn : 0) * a.Kp + src_chunk * 8) * 2;
  }
  const long slice_bytes = (long)a.Nw * a.Kp * 2;

  auto dma = [&](const char* src, char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  unsigned hal_live = 0;      // which of this lane's halo elements are real activations in the image staged last (PRE)
  auto issue_halo = [&](int dt, int c0) {
    const unsigned cok = (unsigned)(c0 + src_chunk * 8 < a.Cin);
    hal_live = 0;
#pragma unroll
    for (int j = 0; j < HL; ++j) {
      const int t = TM ? to - 1 + ((j * 4 + wave) * 8) / HW : to * a.sT + dt;
      const unsigned ok = cok & (unsigned)((unsigned)t < (unsigned)a.Ti) & ((hal_ok >> j) & 1u);
      const char* base = xb + ((long)t * frame_elems + c0) * 2;
      const char* src = zero + (((base + (long)hal_off[j] * 2) - zero) & -(long)ok);
      dma(src, halo + (j * 4 + wave) * 1024);
      hal_live |= ok << j;
    }
  };
  float* const aff = (float*)(smem + Cfg::KLOOP_BYTES);
  auto xform_halo = [&](int c0) {
    const float* sp_ = aff + c0 + src_chunk * 8;
    const float4 s0 = *(const float4*)sp_, s1 = *(const float4*)(sp_ + 4);
    const float4 h0_ = *(const float4*)(sp_ + a.Kp), h1_ = *(const float4*)(sp_ + a.Kp + 4);
    const f32x2_v sc2[4] = {{s0.x, s0.y}, {s0.z, s0.w}, {s1.x, s1.y}, {s1.z, s1.w}};
    const f32x2_v sh2[4] = {{h0_.x, h0_.y}, {h0_.z, h0_.w}, {h1_.x, h1_.y}, {h1_.z, h1_.w}};
#pragma unroll
    for (int j = 0; j < HL; ++j) {
      uint4* q = (uint4*)(halo + (j * 4 + wave) * 1024 + lane * 16);
      const uint4 v = *q;
      const uint32_t m = (hal_live >> j) & 1u ? 0xffffffffu : 0u;
      *q = make_uint4(pre_relu_pair(v.x, sc2[0], sh2[0]) & m, pre_relu_pair(v.y, sc2[1], sh2[1]) & m,
                      pre_relu_pair(v.z, sc2[2], sh2[2]) & m, pre_relu_pair(v.w, sc2[3], sh2[3]) & m);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the raw s_barrier that follows does not wait for LDS stores)
  };
  if constexpr (PRE) {
    for (int c = tid; c < a.Kp; c += 256) {
      const bool in = c < a.Cin;
      aff[c] = in ? a.in_scale[c] : 0.f;
      aff[a.Kp + c] = in ? a.in_shift[c] : 0.f;
    }
    __syncthreads();   // (plain loads above are complete before any DMA is counted)
  }
  auto issue_b1 = [&](int j, int slot, bool live, int slice, int c0) {      // weight DMA j (of BL) of (slice, c0) into ring slot `slot`
    char* dst = bring + slot * Cfg::BSLOT_BYTES + wave * 1024;
    const long delta = (long)slice * slice_bytes + (long)c0 * 2;
    const unsigned ok = (unsigned)live & (unsigned)(c0 + src_chunk * 8 < a.Kp) & b_ok[j];
    const char* src = zero + (((a.w + b_off[j] + delta) - zero) & -(long)ok);
    dma(src, dst + j * 4096);
  };
  auto issue_b = [&](int slot, bool live, int slice, int c0) {
#pragma unroll
    for (int j = 0; j < BL; ++j) issue_b1(j, slot, live, slice, c0);
  };

  // ---- fragments ---------------------------------------------------------------------------------------------------------
  f32x16_v acc[2][NT32];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT32; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int pcol = lane & 31, kh = lane >> 5;
  // halo position of this lane's row of activation fragment I for the centre tap
  int pl[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if constexpr (TM) {
      pl[i] = (wave + 1) * HW + i * 32 + pcol;                // wave = output frame of the tile, fragment i = 32 positions
    } else if constexpr (TW == 32) {
      pl[i] = (2 * wave + i + 1) * HW + 1 + pcol;             // fragment i = image row 2 wave + i of the tile
    } else {
      pl[i] = (4 * wave + 2 * i + (pcol >> 4) + 1) * HW + 1 + (pcol & 15);   // two image rows of 16 positions
    }
  }
  constexpr int WROWS = TM ? 1 : TR / 4;
  const int hw0 = h0 + wave * WROWS;
  EpiRows er;
  if constexpr (TM) {
    er.m0 = (int)(((uint32_t)b * (uint32_t)a.To + (uint32_t)(to + wave)) * (uint32_t)HWtot + (uint32_t)p0);
    er.ipr = 4;
    er.rstride = 0;
    er.nrows = to + wave < a.To ? 1 : 0;
    er.ncols = (HWtot - p0) / 16 < 4 ? (HWtot - p0) / 16 : 4;
  } else {
    er.m0 = (int)((frame * (uint32_t)a.Ho + (uint32_t)hw0) * (uint32_t)a.Wo + (uint32_t)w0);
    er.ipr = TW / 16;
    er.rstride = a.Wo;
    er.nrows = a.Ho - hw0 < 0 ? 0 : (a.Ho - hw0 > WROWS ? WROWS : a.Ho - hw0);
    er.ncols = TW / 16;
  }
  // weight fragment rows: n = 32 J + pcol, (n >> 1) & 7 = (pcol >> 1) & 7; chunk 2 s + kh
  int wfo[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) wfo[s] = pcol * 128 + ((((2 * s + kh) ^ ((pcol >> 1) & 7))) << 4);

  // One K step (64 channels of one tap) = two halves of two 16-wide MFMA K slices each.  The fragment reads of a half are
  // issued BEFORE the MFMAs of the previous half, so a wave's own LDS latency hides behind its own matrix work: with the
  // reads of a whole step up front (conv_ht.h) both resident workgroups of a CU sat in their read phases together and the
  // matrix pipe idled a third of the time (s_memtime: 2350 cycles per K step for 2 x 790 of MFMA).
  struct Frags { bf16x8_v x[2][2], w[2][NT32]; };
  auto read_half = [&](Frags& f, int slot, int tapoff, int half) {
    const char* Bs = bring + slot * Cfg::BSLOT_BYTES;
    int pp[2], ps[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { pp[i] = pl[i] + tapoff; ps[i] = (pp[i] >> 1) & 7; pp[i] <<= 7; }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int s = half * 2 + s2;
#pragma unroll
      for (int i = 0; i < 2; ++i) f.x[s2][i] = *(const bf16x8_v*)(halo + pp[i] + (((2 * s + kh) ^ ps[i]) << 4));
#pragma unroll
      for (int j = 0; j < NT32; ++j) f.w[s2][j] = *(const bf16x8_v*)(Bs + j * 4096 + wfo[s]);
    }
  };
  auto mma_half = [&](const Frags& f) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT32; ++j) mfma32_bf16_acc(acc[i][j], f.w[s2][j], f.x[s2][i]);
  };
  // ---- pipeline: as conv_ht_kernel ---------------------------------------------------------------------------------------
  constexpr int GT = 9;
#ifdef VINET_CONV_TIMING
  const unsigned long long tm1 = __builtin_amdgcn_s_memtime();
#endif
  int t0 = 0;
  while (t0 < a.ntaps) {
    unsigned long long tq0 = 0, tq1 = 0;
    uint32_t tq2 = 0;
    const int dt = load_tap(a.taps, t0).x;
    int nt = 0;
#pragma unroll
    for (int j = 0; j < GT; ++j) {
      const bool in = (t0 + j < a.ntaps) && (nt == j);
      const int4 tp = load_tap(a.taps, t0 + j < a.ntaps ? t0 + j : t0);
      const bool same = in && (TM || tp.x == dt);
      const int off = TM ? tp.x * HW : tp.y * HW + tp.z;
      const unsigned long long e = (unsigned long long)(((tp.w & 0xff) << 8) | ((off + 128) & 0xff));
      if (j < 4) tq0 |= e << (j * 16);
      else if (j < 8) tq1 |= e << ((j - 4) * 16);
      else tq2 = (uint32_t)e;
      nt += same ? 1 : 0;
    }
    auto tap_word = [&](int j) {
      const unsigned long long q = j < 4 ? tq0 : tq1;
      const uint32_t w16 = j < 8 ? (uint32_t)(q >> ((j & 3) * 16)) : tq2;
      return w16 & 0xffffu;
    };
    for (int c0 = 0; c0 < a.Kp; c0 += 64) {
#ifdef VINET_CONV_TIMING
      tm_h0 = __builtin_amdgcn_s_memtime();
#endif
      __builtin_amdgcn_s_barrier();                // everyone has finished reading the old halo image and ring
      asm volatile("" ::: "memory");
      issue_halo(dt, c0);
#pragma unroll
      for (int j = 0; j < BSLOTS - 1; ++j) issue_b(j, j < nt, (int)(tap_word(j) >> 8), c0);
      int slot = 0, fill = BSLOTS - 1;
      // first step of the block: its data (and the halo image) must have landed everywhere before anyone reads
      wait_vmcnt<BL*(BSLOTS - 2)>();
      if constexpr (PRE) xform_halo(c0);
#ifdef VINET_CONV_TIMING
      tm_halo += __builtin_amdgcn_s_memtime() - tm_h0;
#endif
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue_b(fill, BSLOTS - 1 < nt, (int)(tap_word(BSLOTS - 1 < GT ? BSLOTS - 1 : 0) >> 8), c0);
      fill = fill + 1 == BSLOTS ? 0 : fill + 1;
      Frags fa, fb;
      read_half(fa, slot, (int)(tap_word(0) & 0xffu) - 128, 0);
      for (int j = 0; j < nt; ++j) {
#ifdef VINET_CONV_TIMING2
        const unsigned long long q0 = __builtin_amdgcn_s_memtime();
#endif
        read_half(fb, slot, (int)(tap_word(j) & 0xffu) - 128, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma_half(fa);
        __builtin_amdgcn_sched_barrier(0);
#ifdef VINET_CONV_TIMING2
        const unsigned long long q1 = __builtin_amdgcn_s_memtime();
        tq_a += q1 - q0;
        unsigned long long q2 = q1, q3 = q1;
#endif
        if (j + 1 < nt) {
          // my reads of this step's slot are back (they were issued a dozen MFMAs ago: free) -- nobody is left reading it
          // when the DMAs issued behind the barrier refill the slot read one step earlier
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          wait_vmcnt<BL*(BSLOTS - 2)>();             // my weight DMAs of step j + 1 have landed
          __builtin_amdgcn_s_barrier();              // everyone's have
          asm volatile("" ::: "memory");
#ifdef VINET_CONV_TIMING2
          q2 = __builtin_amdgcn_s_memtime();
#endif
          const int jn = j + BSLOTS;
          slot = slot + 1 == BSLOTS ? 0 : slot + 1;
          issue_b(fill, jn < nt, (int)(tap_word(jn < GT ? jn : 0) >> 8), c0);
#ifdef VINET_CONV_TIMING2
          q3 = __builtin_amdgcn_s_memtime();
#endif
          fill = fill + 1 == BSLOTS ? 0 : fill + 1;
          read_half(fa, slot, (int)(tap_word(j + 1) & 0xffu) - 128, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        mma_half(fb);
        __builtin_amdgcn_sched_barrier(0);
#ifdef VINET_CONV_TIMING2
        tq_w += q2 - q1; tq_i += q3 - q2; tq_b += __builtin_amdgcn_s_memtime() - q3;
#endif
      }
      asm volatile("" ::: "memory");
    }
    t0 += nt;
  }
  wait_vmcnt<0>();
  mfma32_drain();
  __syncthreads();
#ifdef VINET_CONV_TIMING
  const unsigned long long tm2 = __builtin_amdgcn_s_memtime();
  const float* dbg_ptr = a.out_shift;
  ConvArgs a2 = a;
  a2.out_shift = nullptr; a2.out_scale = nullptr;
  conv_epilogue32<NT32>(a2, acc, smem, (int)sp, tile_n, er);
  if (tid == 0 && dbg_ptr) {   // tuning build only: out_shift doubles as a [grid][4] float dump
    const unsigned long long tm3 = __builtin_amdgcn_s_memtime();
    float* dbg = (float*)dbg_ptr + (long)blockIdx.x * 4;
#ifdef VINET_CONV_TIMING2
    dbg[0] = (float)tq_a; dbg[1] = (float)tq_w; dbg[2] = (float)tq_i; dbg[3] = (float)tq_b;      // readB+mmaA / wait+barrier / DMA issue / readA+mmaB
#else
    dbg[0] = (float)(tm1 - tm0); dbg[1] = (float)(tm2 - tm1); dbg[2] = (float)(tm3 - tm2); dbg[3] = (float)tm_halo;
#endif
  }
#else
  conv_epilogue32<NT32>(a, acc, smem, (int)sp, tile_n, er);
#endif
}

template <int NT32, int TW, int BSLOTS, bool TM = false, bool PRE = false>
static int launch_conv_ht32_cfg(const ConvArgs& a, hipStream_t s) {
  using Cfg = ConvHt32Cfg<NT32, TW, BSLOTS, TM, PRE>;
  auto kern = conv_ht32_kernel<NT32, TW, BSLOTS, TM, PRE>;
  static bool attr_done[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_done[dev & 63]) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::smem_bytes(1024));
    if (e != hipSuccess) { vinet_set_error("hipFuncSetAttribute(conv_ht32): %s", hipGetErrorString(e)); return (int)e; }
    attr_done[dev & 63] = true;
  }
  const long Bn = a.M / ((long)a.To * a.Ho * a.Wo);
  const long grid = (long)a.tilesN * a.ht_tilesW * a.ht_tilesH * (TM ? 1 : a.To) * Bn;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), Cfg::smem_bytes(a.Kp), s, a);
  return vn_launch_status("conv_ht32");
}
