// LDS-DMA pipelined implicit-GEMM convolution for the split-bf16 form (VINET_F32S): fp32 activations in memory, weights
// packed as hi / lo bf16 planes (layout.hip: pack_store<VINET_F32S>), three bf16 MFMAs per product, fp32 accumulate.
//
// The register-staged form of this arithmetic (conv_igemm.h, SPLIT) runs at the speed of the round-1 kernels: its loader
// gathers, bounds-checks and splits every element in VALU code (50-120 TF/s of fp32-equivalent work).  Here the operands go
// L2 / HBM -> LDS by `global_load_lds_dwordx4` exactly as in conv_dma.h and the SPLIT happens at fragment time, once per
// 8-element fragment and K step, amortised over the NT column tiles the fragment multiplies (6 VALU per pair of elements):
//
//   * tile 128 x (16 NT) output voxels x channels, K step = 32 channels, 4 waves stacked along M (wave = 32 rows x all columns);
//   * A stage: 128 rows x 128 B (32 fp32), written in 8-row x 128-byte pieces (whole cache lines of a voxel row), 16-byte chunk
//     index XOR (row & 7) on the DMA source and on the reads; lane group q reads pieces q and q + 4 of its row (channels
//     4q .. 4q+3 and 16+4q .. 16+4q+3 of the chunk): adjacent lane groups read adjacent pieces, the conflict-free pattern of
//     conv_pp.h -- and the K order the weight pack is permuted to;
//   * B stage: 16 NT rows x 128 B = [32 bf16 hi | 32 bf16 lo] of one output channel, the same pieces and swizzle: lane group
//     q reads piece q (hi) and piece 4 + q (lo);
//   * out-of-range taps / rows / channels read a zero page (a page of fp32 NaNs under a pending BatchNorm + ReLU, applied in
//     fp32 registers before the split: max(NaN, 0) = 0); every wave issues exactly LPS DMAs per stage, counted vmcnt, one raw
//     s_barrier per K step, STAGES - 1 steps of loads in flight (conv_dma.h);
//   * epilogue: the shared conv_epilogue (fp32 outputs, statistics, activation, accumulate, any placement).
#pragma once
#include "conv_dma.h"

__device__ __attribute__((aligned(128))) uint4 g_vinet_zero_page3[8];
__device__ __attribute__((aligned(128))) uint4 g_vinet_nan_page3[8] = {
    {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu},
    {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu},
    {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu},
    {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}};

template <int NT, int STAGES, bool PRE>
struct ConvDma3Cfg {
  static constexpr int MT = 2, BM = 128, BN = 16 * NT;
  static constexpr int BROWS = (BN + 31) / 32 * 32;         // B rows staged (whole rounds of 4 waves x 8 rows)
  static constexpr int A_LOADS = BM / 32;                    // 8-row pieces per wave
  static constexpr int B_LOADS = BROWS / 32;
  static constexpr int LPS = A_LOADS + B_LOADS;
  static constexpr int STAGE_BYTES = (BM + BROWS) * 128;
  static constexpr int KLOOP_BYTES = STAGES * STAGE_BYTES;
  static constexpr int EPI_BYTES = conv_epi_bytes<MT, NT, 4, 1>();
  static int smem_bytes(int Kp) {
    const int k = KLOOP_BYTES + (PRE ? 2 * Kp * 4 : 0);
    return k > EPI_BYTES ? k : EPI_BYTES;
  }
};

template <int NT, int STAGES, bool PRE>
__global__ __launch_bounds__(256, 2) void conv_dma3_kernel(const ConvArgs a) {
  using Cfg = ConvDma3Cfg<NT, STAGES, PRE>;
  constexpr int MT = Cfg::MT, BM = Cfg::BM, BN = Cfg::BN, A_LOADS = Cfg::A_LOADS, B_LOADS = Cfg::B_LOADS;
  static_assert(STAGES >= 2 && Cfg::LPS * (STAGES - 2) <= 63, "vmcnt immediate range");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = wg % a.tilesN, tile_m = wg / a.tilesN;
  const char* zero = (const char*)g_vinet_zero_page3;
  const char* apad = PRE ? (const char*)g_vinet_nan_page3 : zero;
  float* aff = (float*)(smem + STAGES * Cfg::STAGE_BYTES);         // PRE: scale[0..Kp), shift[0..Kp)
  if constexpr (PRE) {
    for (int c = tid; c < a.Kp; c += 256) {
      const bool in = c < a.Cin;
      aff[c] = in ? a.in_scale[c] : 0.f;
      aff[a.Kp + c] = in ? a.in_shift[c] : 0.f;
    }
    __syncthreads();
  }
  // this lane's DMA role: row (lane >> 3) of an 8-row piece, LDS slot (lane & 7), source piece = slot ^ row
  const int prow = lane >> 3;
  const int src_piece = (lane & 7) ^ prow;
  const char* a_ptr[A_LOADS];
  int a_t[A_LOADS], a_h[A_LOADS], a_w[A_LOADS];
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) {
    const int row = (i * 4 + wave) * 8 + prow;
    const int m = tile_m * BM + row;
    if (m < a.M) {
      int b, to, ho, wo;
      decode_m(m, a.dW, a.dH, a.dT, b, to, ho, wo);
      a_t[i] = to * a.sT; a_h[i] = ho * a.sH; a_w[i] = wo * a.sW;
      const long off = (long)b * a.sBx + ((long)(a_t[i] * a.Hi + a_h[i]) * a.Wi + a_w[i]) * (long)a.ldx + src_piece * 4;
      a_ptr[i] = a.x + off * 4;
    } else {
      a_ptr[i] = apad; a_t[i] = -(1 << 28); a_h[i] = 0; a_w[i] = 0;
    }
  }
  const char* b_ptr[B_LOADS];
  unsigned b_ok[B_LOADS];
#pragma unroll
  for (int j = 0; j < B_LOADS; ++j) {
    const int n = (j * 4 + wave) * 8 + prow;
    const int nn = tile_n * BN + n;
    b_ok[j] = (unsigned)(n < BN) & (unsigned)(nn < a.Nw);
    b_ptr[j] = a.w + (long)(b_ok[j] ? nn : 0) * (long)a.Kp * 4 + src_piece * 16;      // row = Kp / 32 chunks of 128 B
  }
  const int cpt = a.Kp / 32;
  const int nchunks = a.ntaps * cpt;
  const long slice_bytes = (long)a.Nw * a.Kp * 4;

  int isu = 0, isu_tap = 0, isu_c = 0;
  // (PRE: the affine tables are plain LDS loads, in front of which hipcc drains vmcnt(0) when the DMA is a builtin: common.h)
  auto dma = [&](const char* src, char* dst) {
    if constexpr (PRE) lds_dma16_asm(src, dst);
    else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  auto issue = [&](int slot) {
    char* stage = smem + slot * Cfg::STAGE_BYTES;
    if (isu < nchunks) {
      const int4 tp = load_tap(a.taps, isu_tap);
      const long tap_delta = (((long)(tp.x * a.Hi + tp.y) * a.Wi + tp.z) * (long)a.ldx + isu_c) * 4;
      const unsigned cin_ok = (unsigned)(isu_c + src_piece * 4 < a.Cin);
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i) {
        const int ti = a_t[i] + tp.x, hi = a_h[i] + tp.y, wi = a_w[i] + tp.z;
        const unsigned ok = cin_ok & (unsigned)((unsigned)ti < (unsigned)a.Ti) & (unsigned)((unsigned)hi < (unsigned)a.Hi) &
                            (unsigned)((unsigned)wi < (unsigned)a.Wi);
        const char* src = apad + (((a_ptr[i] + tap_delta) - apad) & -(long)ok);
        dma(src, stage + (i * 4 + wave) * 1024);
      }
      const long wdelta = (long)tp.w * slice_bytes + (long)isu_c * 4;
#pragma unroll
      for (int j = 0; j < B_LOADS; ++j) {
        const char* src = zero + (((b_ptr[j] + wdelta) - zero) & -(long)b_ok[j]);
        dma(src, stage + BM * 128 + (j * 4 + wave) * 1024);
      }
      isu_c += 32;
      if (isu_c >= a.Kp) { isu_c = 0; ++isu_tap; }
    } else {
#pragma unroll
      for (int i = 0; i < A_LOADS + B_LOADS; ++i) dma(zero, stage + (i * 4 + wave) * 1024);     // keep the DMA count per stage exact
    }
    ++isu;
  };

  f32x4_v acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4_v){0.f, 0.f, 0.f, 0.f};

  // fragment reads: row (lane & 15) of a 16-row group, pieces q and q + 4, swizzled with row & 7 (= lane & 7)
  const int q = lane >> 4;
  const int fo0 = (lane & 15) * 128 + ((q ^ (lane & 7)) << 4), fo1 = (lane & 15) * 128 + (((q + 4) ^ (lane & 7)) << 4);
  int cmp_c = 0;
  auto compute = [&](int slot) {
    const char* As = smem + slot * Cfg::STAGE_BYTES + (wave * MT * 16) * 128;
    const char* Bs = smem + slot * Cfg::STAGE_BYTES + BM * 128;
    uint4 ar[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i) { ar[i][0] = *(const uint4*)(As + i * 2048 + fo0); ar[i][1] = *(const uint4*)(As + i * 2048 + fo1); }
    bf16x8_v bh[NT], bl[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) { bh[j] = *(const bf16x8_v*)(Bs + j * 2048 + fo0); bl[j] = *(const bf16x8_v*)(Bs + j * 2048 + fo1); }
    float4 s0, s1, h0, h1;
    if constexpr (PRE) {      // this lane's channels: cmp_c + 4q .. +3 and cmp_c + 16 + 4q .. +3
      const float* sp = aff + cmp_c + 4 * q;
      s0 = *(const float4*)sp; s1 = *(const float4*)(sp + 16);
      h0 = *(const float4*)(sp + a.Kp); h1 = *(const float4*)(sp + a.Kp + 16);
      cmp_c += 32;
      if (cmp_c >= a.Kp) cmp_c = 0;
    }
    bf16x8_v ah[MT], al[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      float f[8] = {__uint_as_float(ar[i][0].x), __uint_as_float(ar[i][0].y), __uint_as_float(ar[i][0].z), __uint_as_float(ar[i][0].w),
                    __uint_as_float(ar[i][1].x), __uint_as_float(ar[i][1].y), __uint_as_float(ar[i][1].z), __uint_as_float(ar[i][1].w)};
      if constexpr (PRE) {
        const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = fmaxf(fmaf(f[e], sc[e], sh[e]), 0.f);       // (padding: NaN page -> max(NaN, 0) = 0)
      }
      union { bf16x8_v v; uint32_t u[4]; } H, Lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) split_pair(f[2 * e], f[2 * e + 1], H.u[e], Lo.u[e]);
      ah[i] = H.v; al[i] = Lo.v;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {     // small terms first; weights as the A operand (transposed tile: conv_epilogue)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
      }
  };

#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s) issue(s);
  int slot = 0, fill = STAGES - 1;
  for (int it = 0; it < nchunks; ++it) {
    wait_vmcnt<Cfg::LPS*(STAGES - 2)>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(fill);
    compute(slot);
    asm volatile("" ::: "memory");
    slot = slot + 1 == STAGES ? 0 : slot + 1;
    fill = fill + 1 == STAGES ? 0 : fill + 1;
  }
  wait_vmcnt<0>();
  __syncthreads();
  conv_epilogue<MT, NT, 4, 1>(a, acc, smem, tile_m, tile_n);
}

template <int NT, bool PRE>
static int launch_conv_dma3_cfg(const ConvArgs& a, hipStream_t s) {
  using Cfg = ConvDma3Cfg<NT, 3, PRE>;
  auto kern = conv_dma3_kernel<NT, 3, PRE>;
  static bool attr_done[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_done[dev & 63]) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::smem_bytes(1024));
    if (e != hipSuccess) { vinet_set_error("hipFuncSetAttribute(conv_dma3): %s", hipGetErrorString(e)); return (int)e; }
    attr_done[dev & 63] = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.tilesM * a.tilesN), dim3(256), Cfg::smem_bytes(a.Kp), s, a);
  return vn_launch_status("conv_dma3");
}
