// Pointwise (1x1x1, unit stride) convolution as a wave-independent streaming GEMM (bf16, gfx950): the entry convs of the
// Inception blocks (model_utils.py:176-187, one joint launch per block), the BasicConv3d 1x1x1 layers (model_utils.py:126)
// and every data gradient of those.
//
// Why a fifth conv generation: these layers are memory-bound GEMMs with a tiny K loop (Cin = 64...288: 2...9 steps of 32) at
// millions of rows.  conv_dma.h spends 2300...2800 cycles per K step of 32 (barrier + DMA issue + wait) and 12...17 k cycles
// in the epilogue of a 256 x 96 tile whose K loop is 8 steps: 1.8...2.2 TB/s of tensor traffic at 250...350 TF/s, under half of
// either roofline (profiles/r3_pw_ab.txt).  There is nothing to share between the rows of such a GEMM except the weights:
//
//   * the WEIGHT tile (BN = 32 / 64 / 96 output channels x all of K) is staged in LDS once per workgroup and stays there; the
//     workgroup walks a block of rows (4 waves x up to 16 tiles of 64).  Row stride K * 2 + 16 bytes: a ds_read_b128 fragment read (16 rows x 16 B per
//     lane group) touches 16 different 16-byte bank groups for every K that is a multiple of 32.
//   * the ACTIVATIONS never touch LDS: with the operands swapped (weights as the MFMA A operand, conv_igemm.h) a lane's B
//     fragment is 8 consecutive channels of ONE voxel, i.e. one 16-byte global load, and each fragment feeds NT MFMAs from
//     registers.  A wave owns 64 rows x BN columns and runs its own software pipeline: a ring of four K steps in registers,
//     loads three steps (1.2...1.7 k MFMA cycles) ahead of their use, ACROSS row tiles -- the first steps of the next tile are in
//     flight under the epilogue of this one.  Waves never synchronise after the weight tile is staged, so one wave's epilogue
//     (stores, BN partial sums) overlaps the other seven waves' loads and MFMAs on the CU.
//   * the ring loads are inline-asm global_load_dwordx4 with hand-counted s_waitcnt vmcnt: the compiler's own counter model
//     joins the (conditional) epilogue path with the plain path conservatively and drains the ring once per tile.  The counting
//     does not depend on how many memory instructions the epilogue issues: before an epilogue everything but the newest ring
//     step is waited for (vmcnt(4)), which confirms the next two steps; every other step waits for vmcnt(12), i.e. for
//     everything older than the three newest ring steps -- both hold whatever else sits in the queue.
//   * epilogue per wave: permlane swap as in conv_epilogue, then a wave-private LDS image and whole-row stores (below);
//     BN partial sums are reduced over the wave's rows with DPP adds and accumulated in a wave-private LDS row across all of the
//     wave's tiles, so a launch writes ONE statistics row per workgroup (vinet_conv3d_stats_rows).
#pragma once
#include "conv_dma.h"

template <int NT>
struct ConvPwCfg {
  static constexpr int BN = NT * 16, MT = 4;
  static int row_stride(int Kp) { return Kp * 2 + 16; }
  // [weight tile][out scale, out shift: BN each][PRE: in scale, in shift: Kp each][statistics: 4 waves x BN x (sum, sum^2)]
  // [output image: 4 waves x 32 rows x BN bf16]
  static int smem_bytes(int Kp, bool pre) { return BN * row_stride(Kp) + 2 * BN * 4 + (pre ? 2 * Kp * 4 : 0) + 4 * BN * 2 * 4 + 4 * 32 * BN * 2; }
};

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_v;

VN_DEV void pw_load16(u32x4_v& dst, const char* src) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src) : "memory");
}
// everything older than the N newest vector-memory instructions of this wave has completed.  NO register operands: a "+v" tie
// makes the compiler copy the (still in flight) ring registers into the asm's operand registers in front of the wait.  The
// consumers cannot move above the wait: it sits in a basic block of its own (conditional) or is followed by a scheduling barrier.
template <int N>
VN_DEV void pw_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

template <int NT, bool PRE>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(2, NT >= 4 ? 2 : 3))) void conv_pw_kernel(const ConvArgs a) {
  using Cfg = ConvPwCfg<NT>;
  constexpr int MT = 4, BN = Cfg::BN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, q = lane >> 4;
  const int RS = a.Kp * 2 + 16;
  char* const Ws = smem;
  float* const osc = (float*)(smem + BN * RS);
  float* const aff = osc + 2 * BN;
  float* const red = aff + (PRE ? 2 * a.Kp : 0);
  char* const stg = (char*)(red + 4 * BN * 2) + wave * (32 * BN * 2);     // this wave's 32-row output image
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = wg % a.tilesN, gm = wg / a.tilesN;
  const int n_wave = tile_n * BN;

  // ---- stage the weight tile and the per-channel constants ---------------------------------------------------------------
  {
    const int kq = a.Kp >> 3;                     // 16-byte pieces per weight row
    for (int e = tid; e < BN * kq; e += 256) {
      const int n = e / kq, c = e - n * kq;
      const int nn = n_wave + n;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (nn < a.Nw) v = *(const uint4*)(a.w + ((long)nn * a.Kp + c * 8) * 2);
      *(uint4*)(Ws + n * RS + c * 16) = v;
    }
    for (int n = tid; n < BN; n += 256) {
      const int nn = n_wave + n;
      const bool ok = nn < a.Nw;
      osc[n] = (a.out_scale && ok) ? a.out_scale[nn] : 1.f;
      osc[BN + n] = (a.out_shift && ok) ? a.out_shift[nn] : 0.f;
    }
    if constexpr (PRE) {
      for (int c = tid; c < a.Kp; c += 256) {
        const bool in = c < a.Cin;
        aff[c] = in ? a.in_scale[c] : 0.f;
        aff[a.Kp + c] = in ? a.in_shift[c] : 0.f;
      }
    }
    for (int e = tid; e < 4 * BN * 2; e += 256) red[e] = 0.f;
  }
  __syncthreads();      // (the compiler drains its own loads here: the ring starts with an empty queue)

  // a wave owns a.chunks_per_split CONSECUTIVE 64-row tiles: enough of them to amortise the weight staging, few enough that the
  // grid is many workgroups per CU -- the launch shares the chip with the persistent weight-gradient workgroups of the second
  // stream, and a grid of exactly "two workgroups per CU" ran in two rounds there (profiles/r3_pw_ab.txt)
  const int tpw = a.chunks_per_split;
  const int wt0 = (gm * 4 + wave) * tpw;
  const int nwt_all = (a.M + 63) >> 6;            // 64-row wave tiles of the problem
  const int nwt = wt0 + tpw < nwt_all ? wt0 + tpw : nwt_all;      // (this wave's end)
  constexpr int wstride = 1;
  const int KS = a.Kp >> 5;
  const bool x_lin = a.sBx == (long)a.Ti * a.Hi * a.Wi * a.ldx;
  const char* const page = (PRE ? (const char*)g_vinet_nan_page : (const char*)g_vinet_zero_page) + q * 16;

  // ---- issue side: K step i_s of wave tile i_wt -----------------------------------------------------------------------------
  int i_wt = wt0, i_s = 0;
  const char* rowp[MT];                           // my voxel of row group i, channel group q of K step 0
  auto set_rows = [&](int wt) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      int m = wt * 64 + i * 16 + p;
      m = m < a.M ? m : a.M - 1;                  // rows past the end re-read the last row (never stored, never counted)
      long off;
      if (x_lin) {
        off = (long)m * a.ldx;
      } else {
        int b, to, ho, wo;
        decode_m(m, a.dW, a.dH, a.dT, b, to, ho, wo);
        off = (long)b * a.sBx + ((long)(to * a.Hi + ho) * a.Wi + wo) * (long)a.ldx;
      }
      rowp[i] = a.x + (off + q * 8) * 2;
    }
  };
  auto issue = [&](u32x4_v (&dst)[MT]) {
    // (past the last tile the wave keeps loading its last rows: every step issues exactly MT loads, the counts stay exact)
    const bool kok = i_s * 32 + q * 8 < a.Cin;
#pragma unroll
    for (int i = 0; i < MT; ++i) pw_load16(dst[i], kok ? rowp[i] + i_s * 64 : page);
    if (++i_s == KS) {
      i_s = 0;
      i_wt += wstride;
      if (i_wt < nwt) set_rows(i_wt);
    }
  };

  // ---- compute side ---------------------------------------------------------------------------------------------------------------
  f32x4_v acc[MT][NT];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4_v){0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();
  int c_wt = wt0, c_s = 0;
  const bool do_stats = a.stats != nullptr;
  const bool relu = a.act == VINET_ACT_RELU;
  const bool has_aff = a.out_scale != nullptr || a.out_shift != nullptr;
  float* const myred = red + wave * BN * 2;

  auto voxel_off = [&](int m) {
    if (a.y_linear) return (long)m * a.ldy;
    int b, to, ho, wo;
    decode_m(m, a.dW, a.dH, a.dT, b, to, ho, wo);
    return (long)b * a.sBy + ((long)(to * a.yH + ho) * a.yW + wo) * (long)a.ldy;
  };

  auto epilogue = [&](int wt) {
    const int m0 = wt * 64;
    const bool all_in = n_wave + BN <= a.Nw && m0 + 64 <= a.M;
    bool rok[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) rok[i] = m0 + i * 16 + p < a.M;
    {
      // ---- plain store (vinet_conv3d sends accumulating launches to conv_dma.h): BN partial sums first (a second read of the accumulators is free), then the tile goes through a
      // wave-private LDS image, 32 rows at a time, and leaves as WHOLE rows: BN / 8 consecutive lanes write the BN * 2 contiguous
      // bytes of a voxel.  (The 16 bytes a lane holds after the permlane swap sit next to only one other lane's: stored directly
      // they reach L2 as 32-byte fragments, four instructions per 128-byte line, and the kernel ran at HALF the speed of its own
      // loads + MFMAs -- profiles/r3_pw_ab.txt, "nostore".)
      if (do_stats) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float4 sc4 = *(const float4*)&osc[j * 16 + q * 4], sh4 = *(const float4*)&osc[BN + j * 16 + q * 4];
          const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
          float ss[4] = {0.f, 0.f, 0.f, 0.f}, qq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float v = acc[i][j][r];
              if (has_aff) v = fmaf(v, sc[r], sh[r]);
              if (!all_in) v = (rok[i] && n_wave + j * 16 + q * 4 + r < a.Nw) ? v : 0.f;
              ss[r] += v; qq[r] = fmaf(v, v, qq[r]);
            }
#pragma unroll
          for (int r = 0; r < 4; ++r) { ss[r] = row16_sum(ss[r]); qq[r] = row16_sum(qq[r]); }
          if (p == 0) {
            float* dst = myred + (j * 16 + q * 4) * 2;
            const float4 r0 = *(const float4*)dst, r1 = *(const float4*)(dst + 4);
            *(float4*)dst = make_float4(r0.x + ss[0], r0.y + qq[0], r0.z + ss[1], r0.w + qq[1]);
            *(float4*)(dst + 4) = make_float4(r1.x + ss[2], r1.y + qq[2], r1.z + ss[3], r1.w + qq[3]);
          }
        }
      }
      constexpr int PPR = BN / 8;                           // 16-byte pieces per row of the image
      // piece index XOR swz(row): 16 rows x one piece (the writes) and PPR consecutive pieces of a row (the reads) are both
      // conflict-free with unpadded rows of 64 / 128 / 192 bytes
      auto swz = [](int row) { return NT == 4 ? ((row >> 1) & 7) : ((row >> 2) & 3); };
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int wrow = (q & 1) * 16 + p;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          float v0[4], v1[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) { v0[r] = acc[2 * k][j][r]; v1[r] = acc[2 * k + 1][j][r]; }
          if (has_aff) {
            const float4 sc4 = *(const float4*)&osc[j * 16 + q * 4], sh4 = *(const float4*)&osc[BN + j * 16 + q * 4];
            const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) { v0[r] = fmaf(v0[r], sc[r], sh[r]); v1[r] = fmaf(v1[r], sc[r], sh[r]); }
          }
          uint32_t x0 = cvt_pk_bf16_f32(v0[0], v0[1]), x1 = cvt_pk_bf16_f32(v0[2], v0[3]);
          uint32_t y0 = cvt_pk_bf16_f32(v1[0], v1[1]), y1 = cvt_pk_bf16_f32(v1[2], v1[3]);
          if (relu) { x0 = pk_relu_bf16(x0); x1 = pk_relu_bf16(x1); y0 = pk_relu_bf16(y0); y1 = pk_relu_bf16(y1); }
          permlane16_swap(x0, y0);
          permlane16_swap(x1, y1);
          *(uint4*)(stg + wrow * (BN * 2) + (((2 * j + (q >> 1)) ^ swz(wrow)) * 16)) = make_uint4(x0, x1, y0, y1);
        }
        wave_lds_fence();
#pragma unroll
        for (int it = 0; it < (32 * PPR) / 64; ++it) {
          const int e = lane + 64 * it;
          const int row = e / PPR, piece = e - row * PPR;
          const int m = m0 + 32 * k + row, n = n_wave + piece * 8;
          const uint4 v = *(const uint4*)(stg + row * (BN * 2) + ((piece ^ swz(row)) * 16));
#ifndef PW_NO_STORE      // (tuning build -DPW_NO_STORE: the kernel without its stores -- results are garbage, the time is the point)
          if (m < a.M && n < a.N) *(uint4*)((bf16_t*)a.y + voxel_off(m) + n) = v;
#else
          if (v.x == 0x12345678u && n < a.N) *(uint4*)((bf16_t*)a.y + voxel_off(m < a.M ? m : 0) + n) = v;
#endif
        }
        wave_lds_fence();      // every lane has read the image before the next half overwrites it
      }
    }
  };

  // K step c_s of tile c_wt from ring step `src`; returns true when that finished a tile
  auto compute = [&](u32x4_v (&src)[MT]) {
    bf16x8_v af[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) af[i] = __builtin_bit_cast(bf16x8_v, src[i]);
    if constexpr (PRE) {
      const float* sp = aff + c_s * 32 + q * 8;
      const float4 s0 = *(const float4*)sp, s1 = *(const float4*)(sp + 4);
      const float4 h0 = *(const float4*)(sp + a.Kp), h1 = *(const float4*)(sp + a.Kp + 4);
      const f32x2_v sc2[4] = {{s0.x, s0.y}, {s0.z, s0.w}, {s1.x, s1.y}, {s1.z, s1.w}};
      const f32x2_v sh2[4] = {{h0.x, h0.y}, {h0.z, h0.w}, {h1.x, h1.y}, {h1.z, h1.w}};
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        u32x4_v u = src[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = pre_relu_pair(u[e], sc2[e], sh2[e]);
        af[i] = __builtin_bit_cast(bf16x8_v, u);
      }
    }
    const char* wp = Ws + p * RS + c_s * 64 + q * 16;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const bf16x8_v b = *(const bf16x8_v*)(wp + j * 16 * RS);
#pragma unroll
      for (int i = 0; i < MT; ++i)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, af[i], acc[i][j], 0, 0, 0);   // (weights as A: transposed tile)
    }
    return ++c_s == KS;
  };

  u32x4_v ring[4][MT];
  if (c_wt >= nwt) goto done;        // (more waves than tiles: nothing to do but the final hand-off)
  set_rows(i_wt);
  issue(ring[0]);
  issue(ring[1]);
  issue(ring[2]);
  {
    int confirmed = 0;                // ring steps ahead of the current one whose loads are known to have landed
#define PW_STEP(NEXT, CUR)                                                                              \
    issue(ring[NEXT]);                                                                                  \
    if (confirmed > 0) --confirmed; else pw_wait<12>();                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    if (compute(ring[CUR])) {                                                                           \
      pw_wait<4>();                /* everything but the newest ring step: the next two steps are confirmed */ \
      confirmed = 2;                                                                                    \
      epilogue(c_wt);                                                                                   \
      zero_acc();                                                                                       \
      c_s = 0;                                                                                          \
      c_wt += wstride;                                                                                  \
      if (c_wt >= nwt) break;                                                                           \
    }
    for (;;) {
      PW_STEP(3, 0)
      PW_STEP(0, 1)
      PW_STEP(1, 2)
      PW_STEP(2, 3)
    }
#undef PW_STEP
  }
done:
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the ring's tail loads (nobody reads them)
  if (do_stats) {
    __syncthreads();
    for (int n = tid; n < BN; n += 256) {
      const int nn = n_wave + n;
      if (nn < a.N) {
        float ss = 0.f, qq = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) { ss += red[(w2 * BN + n) * 2]; qq += red[(w2 * BN + n) * 2 + 1]; }
        a.stats[((long)gm * 2 + 0) * a.N + nn] = ss;
        a.stats[((long)gm * 2 + 1) * a.N + nn] = qq;
      }
    }
  }
}

template <int NT, bool PRE>
static int launch_conv_pw_cfg(const ConvArgs& a, hipStream_t s) {
  using Cfg = ConvPwCfg<NT>;
  auto kern = conv_pw_kernel<NT, PRE>;
  static bool attr_done[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_done[dev & 63]) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (e != hipSuccess) { vinet_set_error("hipFuncSetAttribute(conv_pw): %s", hipGetErrorString(e)); return (int)e; }
    attr_done[dev & 63] = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.tilesM * a.tilesN), dim3(256), Cfg::smem_bytes(a.Kp, PRE), s, a);
  return vn_launch_status("conv_pw");
}
