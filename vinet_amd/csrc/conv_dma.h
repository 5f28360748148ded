// LDS-DMA pipelined implicit-GEMM convolution (bf16, gfx950).
//
// Same GEMM view, tiling, tap tables, weight packs and epilogue as
// conv_igemm_kernel, but for inputs that need NO transform on load (every dgrad,
// the whole decoder, all of inference).  Then operands can go HBM/L2 -> LDS
// directly with `global_load_lds_dwordx4` (no VGPR round trip), which makes a
// deep software pipeline cheap:
//
//   * STAGES ring slots of [BM + BNP rows][32 k] bf16, rows UNPADDED (64 B): the
//     DMA writes wave-uniform-base + lane*16, i.e. 16 rows x 4 chunks per
//     wave-instruction.  Bank conflicts of the ds_read_b128 fragment reads are
//     removed by an XOR swizzle of the 16-byte chunk index with (row>>2)&3,
//     applied on the SOURCE address of the DMA and again on the read.
//   * out-of-range taps / rows / channels read a 64-byte zero page instead of
//     being skipped, so every wave issues exactly LPS DMAs per stage and the
//     counted `s_waitcnt vmcnt(LPS*(STAGES-2))` is exact.
//   * one raw s_barrier per K step: [wait my DMAs of stage i] [barrier]
//     [issue stage i+STAGES-1 into the slot read last step] [ds_read + MFMA].
//     STAGES-1 K-steps of loads stay in flight across barriers.
#pragma once
#include "conv_igemm.h"

__device__ __attribute__((aligned(64))) uint4 g_vinet_zero_page[4];
// PRE variant: activations that are out of range must be zero AFTER relu(scale*x+shift).
// They are fetched from a page of 0xFFFF (NEGATIVE bf16 quiet NaN): the NaN keeps its sign through the fma and
// the bf16 conversion, and pre_relu_pair's integer max (common.h) maps every sign-bit pattern to +0, so padding
// needs no per-element mask.
__device__ __attribute__((aligned(64))) uint4 g_vinet_nan_page[4] = {
    {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu},
    {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}};

VN_DEV uint32_t cvt_pk_bf16(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

template <int MT, int NT, int WARPS_M, int WARPS_N, int STAGES, bool PRE = false>
struct ConvDmaCfg {
  static constexpr int BM = 16 * MT * WARPS_M;
  static constexpr int BN = 16 * NT * WARPS_N;
  static constexpr int BNP = (BN + 63) / 64 * 64;   // B rows in LDS (whole wave-instructions)
  static constexpr int A_LOADS = BM / 64;
  static constexpr int B_LOADS = BNP / 64;
  static constexpr int LPS = A_LOADS + B_LOADS;      // DMAs per wave per stage
  static constexpr int STAGE_BYTES = (BM + BNP) * 64;
  static constexpr int WNC = NT * 16, EROW = WNC + 4;
  static constexpr int KLOOP_BYTES = STAGES * STAGE_BYTES;   // PRE adds scale[Kp], shift[Kp] behind the ring
  static constexpr int EPI_BYTES = conv_epi_bytes<MT, NT, WARPS_M, WARPS_N>();
  static int smem_bytes(int Kp) {
    const int k = KLOOP_BYTES + (PRE ? 2 * Kp * 4 : 0);
    return k > EPI_BYTES ? k : EPI_BYTES;
  }
};

template <int N> VN_DEV void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// __launch_bounds__(256, 2): two waves per SIMD (two workgroups per CU).  With one wave per
// SIMD the wave's DMA issue (~6 x 120 cycles), address VALU and MFMAs serialise (1900 cycles
// per K step for 384 cycles of MFMA, s_memtime) and the epilogue's write burst overlaps with
// nothing; a second resident workgroup fills both gaps.
template <int MT, int NT, int WARPS_M, int WARPS_N, int STAGES, bool PRE, bool BNB = false>
__global__ __launch_bounds__(256, 2) void conv_dma_kernel(const ConvArgs a) {
  using Cfg = ConvDmaCfg<MT, NT, WARPS_M, WARPS_N, STAGES, PRE>;
  constexpr int BM = Cfg::BM, BN = Cfg::BN, A_LOADS = Cfg::A_LOADS, B_LOADS = Cfg::B_LOADS;
  static_assert(WARPS_M * WARPS_N == 4, "4 waves");
  static_assert(STAGES >= 2 && Cfg::LPS * (STAGES - 2) <= 63, "vmcnt immediate range");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WARPS_N, wn = wave % WARPS_N;
#ifdef VINET_CONV_TIMING
  const unsigned long long tm0 = __builtin_amdgcn_s_memtime();
#endif
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = wg % a.tilesN, tile_m = conv_tile_perm(a, wg / a.tilesN);
  const char* zero = (const char*)g_vinet_zero_page;
  const char* apad = PRE ? (const char*)g_vinet_nan_page : zero;   // what out-of-range ACTIVATIONS read
  float* aff = (float*)(smem + STAGES * Cfg::STAGE_BYTES);         // PRE: scale[0..Kp), shift[0..Kp)
  if constexpr (PRE) {
    for (int c = tid; c < a.Kp; c += 256) {
      const bool in = c < a.Cin;
      aff[c] = in ? a.in_scale[c] : 0.f;
      aff[a.Kp + c] = in ? a.in_shift[c] : 0.f;
    }
    __syncthreads();   // plain loads above are complete before any DMA is counted
  }

  // this lane's DMA role: row (lane>>2) of a 16-row group, LDS slot (lane&3);
  // it fetches source chunk slot ^ swizzle(row)
  const int lrow = lane >> 2;
  const int src_chunk = (lane & 3) ^ ((lrow >> 2) & 3);

  // ---- per-lane A rows (fixed across the K loop) -----------------------------
  const char* a_ptr[A_LOADS];
  int a_t[A_LOADS], a_h[A_LOADS], a_w[A_LOADS];
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) {
    const int row = i * 64 + wave * 16 + lrow;
    const int m = tile_m * BM + row;
    if (m < a.M) {
      int b, to, ho, wo;
      decode_m(m, a.dW, a.dH, a.dT, b, to, ho, wo);
      a_t[i] = to * a.sT; a_h[i] = ho * a.sH; a_w[i] = wo * a.sW;
      const long off = (long)b * a.sBx + ((long)(a_t[i] * a.Hi + a_h[i]) * a.Wi + a_w[i]) * (long)a.ldx + src_chunk * 8;
      a_ptr[i] = a.x + off * 2;
    } else {
      a_ptr[i] = apad; a_t[i] = -(1 << 28); a_h[i] = 0; a_w[i] = 0;
    }
  }
  // ---- per-lane B rows ----------------------------------------------------------
  const char* b_ptr[B_LOADS];
  unsigned b_ok[B_LOADS];
#pragma unroll
  for (int j = 0; j < B_LOADS; ++j) {
    const int n = j * 64 + wave * 16 + lrow;
    const int nn = tile_n * BN + n;
    b_ok[j] = (unsigned)(n < BN) & (unsigned)(nn < a.Nw);
    b_ptr[j] = a.w + ((long)(b_ok[j] ? nn : 0) * (long)a.Kp + src_chunk * 8) * 2;
  }

  const int cpt = a.Kp / 32;
  int nchunks = a.ntaps * cpt;
  const long slice_bytes = (long)a.Nw * a.Kp * 2;

  // issue state: K chunk `isu` = (tap isu_tap, channel offset isu_c)
  int isu = 0, isu_tap = 0, isu_c = 0;
  if (a.splits > 1) {          // this workgroup's share of the K loop
    isu = blockIdx.y * a.chunks_per_split;
    const int cend = isu + a.chunks_per_split;
    if (cend < nchunks) nchunks = cend;
    isu_tap = isu / cpt;
    isu_c = (isu - isu_tap * cpt) * 32;
  }
  const int chunk0 = isu;
  auto issue = [&](int slot) {
    char* stage = smem + slot * Cfg::STAGE_BYTES;
    if (isu < nchunks) {
      const int4 tp = load_tap(a.taps, isu_tap);
      const long tap_delta = (((long)(tp.x * a.Hi + tp.y) * a.Wi + tp.z) * (long)a.ldx + isu_c) * 2;
      // branch-free source selection: every DMA must be ONE unconditional instruction per
      // wave (a select lowered to divergent branches would issue it twice and break the
      // vmcnt accounting)
      const unsigned cin_ok = (unsigned)(isu_c + src_chunk * 8 < a.Cin);
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i) {
        const int ti = a_t[i] + tp.x, hi = a_h[i] + tp.y, wi = a_w[i] + tp.z;
        const unsigned ok = cin_ok & (unsigned)((unsigned)ti < (unsigned)a.Ti) & (unsigned)((unsigned)hi < (unsigned)a.Hi) &
                            (unsigned)((unsigned)wi < (unsigned)a.Wi);
        const char* src = apad + (((a_ptr[i] + tap_delta) - apad) & -(long)ok);
        if constexpr (PRE) lds_dma16_asm(src, (char*)(stage + (i * 64 + wave * 16) * 64));
        else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(stage + (i * 64 + wave * 16) * 64), 16, 0, 0);
      }
      const long wdelta = (long)tp.w * slice_bytes + (long)isu_c * 2;
#pragma unroll
      for (int j = 0; j < B_LOADS; ++j) {
        const char* src = zero + (((b_ptr[j] + wdelta) - zero) & -(long)b_ok[j]);
        if constexpr (PRE) lds_dma16_asm(src, (char*)(stage + (BM + j * 64 + wave * 16) * 64));
        else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(stage + (BM + j * 64 + wave * 16) * 64), 16, 0, 0);
      }
      isu_c += 32;
      if (isu_c >= a.Kp) { isu_c = 0; ++isu_tap; }
    } else {
      // past the end: keep the DMA count per stage exact
#pragma unroll
      for (int i = 0; i < A_LOADS + B_LOADS; ++i)
        if constexpr (PRE) lds_dma16_asm(zero, (char*)(stage + (i * 64 + wave * 16) * 64));
        else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)zero,
                                         (__attribute__((address_space(3))) void*)(stage + (i * 64 + wave * 16) * 64), 16, 0, 0);
    }
    ++isu;
  };

  f32x4_v acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4_v){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets: row (lane&15) of a 16-row group, k chunk (lane>>4), swizzled
  const int frag_off = (lane & 15) * 64 + (((lane >> 4) ^ ((lane >> 2) & 3)) * 16);

  int cmp_c = isu_c;   // channel offset of the chunk being computed (PRE)
  auto compute = [&](int slot) {
    const char* As = smem + slot * Cfg::STAGE_BYTES + (wm * MT * 16) * 64 + frag_off;
    const char* Bs = smem + slot * Cfg::STAGE_BYTES + (BM + wn * NT * 16) * 64 + frag_off;
    bf16x8_v af[MT], bfr[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) af[i] = *(const bf16x8_v*)(As + i * 16 * 64);
    if constexpr (PRE) {
      // this lane's 8 k-elements are channels cmp_c + (lane>>4)*8 .. +7 in every fragment
      const float* sp = aff + cmp_c + (lane >> 4) * 8;
      const float4 s0 = *(const float4*)sp, s1 = *(const float4*)(sp + 4);
      const float4 h0 = *(const float4*)(sp + a.Kp), h1 = *(const float4*)(sp + a.Kp + 4);
      const f32x2_v sc2[4] = {{s0.x, s0.y}, {s0.z, s0.w}, {s1.x, s1.y}, {s1.z, s1.w}};
      const f32x2_v sh2[4] = {{h0.x, h0.y}, {h0.z, h0.w}, {h1.x, h1.y}, {h1.z, h1.w}};
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        union { bf16x8_v v; uint32_t u[4]; } q;
        q.v = af[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) q.u[e] = pre_relu_pair(q.u[e], sc2[e], sh2[e]);
        af[i] = q.v;
      }
      cmp_c += 32;
      if (cmp_c >= a.Kp) cmp_c = 0;
      valu_to_mfma_pad();
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) bfr[j] = *(const bf16x8_v*)(Bs + j * 16 * 64);
#ifndef VINET_PP_NO_MMA   // (tuning build: K loop without MFMAs)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        mfma_bf16_acc_t(acc[i][j], af[i], bfr[j]);
#endif
  };

  // ---- pipeline ------------------------------------------------------------------
#ifdef VINET_CONV_TIMING
  const unsigned long long tm1 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s) issue(s);
  int slot = 0, fill = STAGES - 1;
  for (int it = chunk0; it < nchunks; ++it) {
    wait_vmcnt<Cfg::LPS*(STAGES - 2)>();      // my DMAs of stage `it` have landed
    __builtin_amdgcn_s_barrier();              // everyone's have; everyone finished reading slot `fill`
    asm volatile("" ::: "memory");
    issue(fill);
    compute(slot);
    asm volatile("" ::: "memory");
    slot = slot + 1 == STAGES ? 0 : slot + 1;
    fill = fill + 1 == STAGES ? 0 : fill + 1;
  }
  wait_vmcnt<0>();                             // drain the tail DMAs before LDS is reused
  mfma_drain();                                // accumulators are about to be read by VALU code
  __syncthreads();
#ifdef VINET_CONV_TIMING
  const unsigned long long tm2 = __builtin_amdgcn_s_memtime();
#endif
  if (a.splits > 1) {          // raw partial sums -> this split's slab of the workspace [splits][M][N]
    const int m_wave = tile_m * BM + wm * MT * 16, n_wave = tile_n * BN + wn * NT * 16;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
      {
        // (transposed tiles, conv_epilogue: a lane holds 4 consecutive columns of one row -> one 16-byte store where it can)
        const int m = m_wave + i * 16 + (lane & 15), n = n_wave + j * 16 + (lane >> 4) * 4;
        float* dst = a.ws + ((long)blockIdx.y * a.M + m) * a.N + n;
        if (m < a.M && n + 3 < a.N && (a.N & 3) == 0) {
          *(float4*)dst = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        } else if (m < a.M) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < a.N) dst[r] = acc[i][j][r];
        }
      }
    return;
  }
  conv_epilogue<MT, NT, WARPS_M, WARPS_N, BNB>(a, acc, smem, tile_m, tile_n);
#ifdef VINET_CONV_TIMING
  if (tid == 0 && a.out_shift) {   // tuning build only: out_shift doubles as a [grid][4] float dump
    const unsigned long long tm3 = __builtin_amdgcn_s_memtime();
    float* dbg = (float*)a.out_shift + (long)blockIdx.x * 4;
    dbg[0] = (float)(tm1 - tm0); dbg[1] = (float)(tm2 - tm1); dbg[2] = (float)(tm3 - tm2); dbg[3] = (float)nchunks;
  }
#endif
}

template <int MT, int NT, int WM, int WN, int STAGES, bool PRE, bool BNB = false>
static int launch_conv_dma_cfg(const ConvArgs& a, hipStream_t s) {
  using Cfg = ConvDmaCfg<MT, NT, WM, WN, STAGES, PRE>;
  auto kern = conv_dma_kernel<MT, NT, WM, WN, STAGES, PRE, BNB>;
  static bool attr_done[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_done[dev & 63]) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::smem_bytes(1024));
    if (e != hipSuccess) { vinet_set_error("hipFuncSetAttribute(conv_dma): %s", hipGetErrorString(e)); return (int)e; }
    attr_done[dev & 63] = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.tilesM * a.tilesN, a.splits > 1 ? a.splits : 1), dim3(256), Cfg::smem_bytes(a.Kp), s, a);
  return vn_launch_status("conv_dma");
}
