// 256x256x64 "ping-pong" implicit-GEMM convolution (bf16, gfx950): the kernel for the
// layers that carry the FLOPs (every dgrad, the decoder, inference, and forward convs whose
// input carries no pending affine).
//
// Why a second generation next to conv_dma.h (measured there: ~500 TF/s in isolation):
//   * conv_dma moves 16 rows x 64 B per LDS-DMA instruction (K step 32).  Half cache lines
//     double the work of the CU's texture-address path for the same bytes; here a K step is
//     64 channels, so every DMA instruction moves 8 rows x 128 B = whole lines.
//   * a 256x128 tile with 4 waves loads 24 KB per 2.1 MFLOP.  Here 8 waves share a 256x256
//     tile: 64 KB per 8.4 MFLOP, 1.5x the flops per byte staged.
//   * one workgroup per CU, its 8 waves in two groups of 4 (upper / lower half of the rows)
//     that run ONE BARRIER APART: while one group issues its 16 MFMAs of a phase the other
//     does its ds_reads, address arithmetic and DMA issue for the next phase, and vice versa.
//
// K loop: an iteration = two K tiles (LDS buffers 0 and 1) = 8 phases.  A K tile is staged
// as four half-tiles of 128 LDS rows x 128 B = 16 KB:
//     A_h : rows {wr*128 + h*64 + r}  (the h-th 64-row block of both wave rows)
//     B_j : cols {wc*64 + j*32 + c}   (the j-th 32-col block of all four wave columns)
// and consumed as four 64x32 C quadrants per wave: (A0,B0) (A0,B1) (A1,B1) (A1,B0), so a
// phase reads 12, 4, 8, 0 fragments.  Every phase issues exactly one half-tile (2 DMAs per
// lane), five phases before its first read and two or more after the last read of the data
// it overwrites:
//     phase   1      2      3      4      5      6      7      8
//     reads   A0 B0  B1     A1     -      A0 B0  B1     A1     -      (buffer 0 | buffer 1)
//     issue   B1(o)  A1(o)  A0(e') B0(e') B1(e') A1(e') A0(o') B0(o')
//   (o = odd tile of this iteration, e'/o' = tiles of the next iteration)
// followed by `s_waitcnt vmcnt(8)`: at most four half-tiles stay in flight, so the one issued
// four phases ago has landed before this phase's barrier and is read next phase.
//
// LDS rows are 128 B = 8 chunks of 16 B; chunk c of row r is stored at slot c ^ (r & 7)
// (applied to the SOURCE address of the DMA, whose LDS side is lane-linear), which makes
// every ds_read_b128 fragment read conflict-free.
//
// Out-of-range taps / rows / channels read a 64-byte zero page, decided by a per-row 64-bit
// tap validity mask computed once in the prologue (ntaps <= 64).
#pragma once
#include "conv_igemm.h"

extern __device__ uint4 g_vinet_zero_page[4];   // defined once per translation unit that includes conv_dma.h

// Two shapes: 256x256 (waves 2 x 4, wave tile 128x64) and 256x192 (waves 4 x 2, wave tile 64x96;
// ViNet is full of 192-channel layers).  The B half-tiles of the 192 shape hold 96 rows: their
// last four DMA pieces are dummies (zero page) so every lane still issues two DMAs per phase.
template <int WM, int WN, int BN_>
struct ConvPPCfg {
  static constexpr int BM = 256, BN = BN_, BK = 64, THREADS = 512;
  static constexpr int MT = BM / (16 * WM), NT = BN / (16 * WN);   // 16x16 tiles per wave
  static constexpr int QM = MT / 2, QN = NT / 2;                   // ... per C quadrant
  static_assert(WM * WN == 8 && MT % 2 == 0 && NT % 2 == 0, "8 waves, even quadrants");
  static constexpr int HALF_BYTES = 128 * 128;          // 16 KB
  static constexpr int BUF_BYTES = 4 * HALF_BYTES;      // A0 A1 B0 B1
  static constexpr int TAP_OFF = 2 * BUF_BYTES;         // 128 KB of staging, then the tap table
  static constexpr int MAX_TAPS = 64;
  static constexpr int SMEM = TAP_OFF + MAX_TAPS * 16;
  static constexpr int A_OFF = 0, B_OFF = 2 * HALF_BYTES;
};

VN_DEV void pp_barrier() { asm volatile("s_barrier" ::: "memory"); }
template <int N> VN_DEV void pp_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
VN_DEV void pp_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// K-tile cursor: tile index -> (tap, channel offset); advanced two tiles at a time
struct PPCursor {
  int kt, tap, c0;
};

template <int WM, int WN, int BN_>
__global__ __launch_bounds__(512, 2) void conv_pp_kernel(const ConvArgs a) {
  using Cfg = ConvPPCfg<WM, WN, BN_>;
  constexpr int MT = Cfg::MT, NT = Cfg::NT, QM = Cfg::QM, QN = Cfg::QN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;
  const int grp = wave >> 2;                       // 0: upper rows, 1: lower rows (runs one barrier behind)
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = wg % a.tilesN, tile_m = conv_tile_perm(a, wg / a.tilesN);
  const char* zero = (const char*)g_vinet_zero_page;

  const int cpt = (a.Kp + 63) >> 6;            // K tiles per tap
  const int nkt = a.ntaps * cpt;
  const long slice_bytes = (long)a.Nw * a.Kp * 2;

  // ---- tap table -> LDS: {activation byte delta, weight slice byte delta} ----------------
  {
    long* ti = (long*)(smem + Cfg::TAP_OFF);
    for (int t = tid; t < a.ntaps; t += Cfg::THREADS) {
      const int4 tp = a.taps[t];
      ti[2 * t] = ((long)(tp.x * a.Hi + tp.y) * a.Wi + tp.z) * (long)a.ldx * 2;
      ti[2 * t + 1] = (long)tp.w * slice_bytes;
    }
  }

  // ---- this lane's DMA role ---------------------------------------------------------------
  // a DMA piece = 8 LDS rows x 128 B; lane -> row (lane>>3), slot (lane&7); source chunk = slot ^ row
  const int prow = lane >> 3;
  const int src_chunk = (lane & 7) ^ prow;
  // A rows of this lane: piece q of half h = LDS row lr = q*64 + wave*8 + prow of that half
  //   -> wave row lr / (QM*16), row in quadrant lr % (QM*16) -> tile row wrow*MT*16 + h*QM*16 + r
  const char* a_ptr[2][2];
  unsigned long long a_mask[2][2];
  {
    int t0[2][2], h0[2][2], w0[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int lr = q * 64 + wave * 8 + prow;
        const int m = tile_m * Cfg::BM + (lr / (QM * 16)) * (MT * 16) + h * (QM * 16) + lr % (QM * 16);
        a_mask[q][h] = 0;
        if (m < a.M) {
          int b, to, ho, wo;
          decode_m(m, a.dW, a.dH, a.dT, b, to, ho, wo);
          t0[q][h] = to * a.sT; h0[q][h] = ho * a.sH; w0[q][h] = wo * a.sW;
          a_ptr[q][h] = a.x + ((long)b * a.sBx + ((long)(t0[q][h] * a.Hi + h0[q][h]) * a.Wi + w0[q][h]) * (long)a.ldx + src_chunk * 8) * 2;
        } else {
          t0[q][h] = -(1 << 28); h0[q][h] = 0; w0[q][h] = 0;   // never in range
          a_ptr[q][h] = zero;
        }
      }
    for (int t = 0; t < a.ntaps; ++t) {
      const int4 tp = load_tap(a.taps, t);
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bool ok = ((unsigned)(t0[q][h] + tp.x) < (unsigned)a.Ti) & ((unsigned)(h0[q][h] + tp.y) < (unsigned)a.Hi) &
                          ((unsigned)(w0[q][h] + tp.z) < (unsigned)a.Wi);
          a_mask[q][h] |= (unsigned long long)ok << t;
        }
    }
  }
  const int a_clim = a.Cin - src_chunk * 8;      // chunk valid for channel offset c0 iff c0 < a_clim
  // B rows: LDS row lr of half j -> wave col lr / (QN*16) -> tile col wcol*NT*16 + j*QN*16 + lr % (QN*16);
  // rows past BN/2 (192-wide shape) are dummies
  const char* b_ptr[2][2];
  unsigned b_ok[2][2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int lr = q * 64 + wave * 8 + prow;
      const int nn = tile_n * Cfg::BN + (lr / (QN * 16)) * (NT * 16) + j * (QN * 16) + lr % (QN * 16);
      b_ok[q][j] = (unsigned)(lr < Cfg::BN / 2) & (unsigned)(nn < a.Nw);
      b_ptr[q][j] = a.w + ((long)(b_ok[q][j] ? nn : 0) * (long)a.Kp + src_chunk * 8) * 2;
    }
  const int b_clim = a.Kp - src_chunk * 8;

  __syncthreads();   // tap table visible; no DMA issued yet

  // ---- DMA issue --------------------------------------------------------------------------
  const long* tapinfo = (const long*)(smem + Cfg::TAP_OFF);
  auto dma = [&](const char* src, char* dst) {
#ifdef VINET_PP_NO_DMA   // tuning build: K loop without staging traffic
    return;
#endif
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  // half-tile `h` of A for the K tile under cursor `c`, into buffer `buf`
  auto issue_a = [&](int buf, int h, const PPCursor& c) {
    char* dst = smem + buf * Cfg::BUF_BYTES + Cfg::A_OFF + h * Cfg::HALF_BYTES + wave * 1024;
    const bool live = c.kt < nkt;
    const int tap = live ? c.tap : 0;
    const long delta = tapinfo[2 * tap] + (long)c.c0 * 2;
    const unsigned cok = (unsigned)live & (unsigned)(c.c0 < a_clim);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const unsigned ok = cok & (unsigned)((a_mask[q][h] >> tap) & 1ull);
      const char* src = zero + (((a_ptr[q][h] + delta) - zero) & -(long)ok);
      dma(src, dst + q * 8192);
    }
  };
  auto issue_b = [&](int buf, int j, const PPCursor& c) {
    char* dst = smem + buf * Cfg::BUF_BYTES + Cfg::B_OFF + j * Cfg::HALF_BYTES + wave * 1024;
    const bool live = c.kt < nkt;
    const int tap = live ? c.tap : 0;
    const long delta = tapinfo[2 * tap + 1] + (long)c.c0 * 2;
    const unsigned cok = (unsigned)live & (unsigned)(c.c0 < b_clim);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const unsigned ok = cok & b_ok[q][j];
      const char* src = zero + (((b_ptr[q][j] + delta) - zero) & -(long)ok);
      dma(src, dst + q * 8192);
    }
  };
  // K tiles run CHANNEL-CHUNK major, taps inner: consecutive tiles re-read the same input lines
  // shifted by one tap (L1/L2 hits), instead of coming back to them a whole tap (Kp/64 tiles x
  // every workgroup on the XCD) later, by which time the 4 MB L2 has lost them
  auto advance2 = [&](PPCursor& c) {
    c.kt += 2;
    c.tap += 2;
    if (c.tap >= a.ntaps) { c.tap -= a.ntaps; c.c0 += 64; }
    if (c.tap >= a.ntaps) { c.tap -= a.ntaps; c.c0 += 64; }
  };

  // ---- fragments --------------------------------------------------------------------------
  f32x4_v acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4_v){0.f, 0.f, 0.f, 0.f};
  bf16x8_v af[QM][2], b0[QN][2], b1[QN][2];

  // lane -> row (lane&15) of a 16-row group, k chunk s*4 + (lane>>4), swizzled by row&7
  const int frow = (lane & 15) * 128;
  const int foff0 = frow + ((((lane >> 4)) ^ (lane & 7)) << 4);
  const int foff1 = frow + (((4 + (lane >> 4)) ^ (lane & 7)) << 4);
  const int a_frag = Cfg::A_OFF + wr * (QM * 2048);   // + h * HALF_BYTES + i * 2048
  const int b_frag = Cfg::B_OFF + wc * (QN * 2048);   // + j * HALF_BYTES + jj * 2048

  auto read_a = [&](int buf, int h) {
    const char* base = smem + buf * Cfg::BUF_BYTES + a_frag + h * Cfg::HALF_BYTES;
#pragma unroll
    for (int i = 0; i < QM; ++i) {
      af[i][0] = *(const bf16x8_v*)(base + i * 2048 + foff0);
      af[i][1] = *(const bf16x8_v*)(base + i * 2048 + foff1);
    }
  };
  auto read_b = [&](int buf, int j, bf16x8_v (&bf)[QN][2]) {
    const char* base = smem + buf * Cfg::BUF_BYTES + b_frag + j * Cfg::HALF_BYTES;
#pragma unroll
    for (int jj = 0; jj < QN; ++jj) {
      bf[jj][0] = *(const bf16x8_v*)(base + jj * 2048 + foff0);
      bf[jj][1] = *(const bf16x8_v*)(base + jj * 2048 + foff1);
    }
  };
  auto mma = [&](int h, int j, const bf16x8_v (&bf)[QN][2]) {
#ifdef VINET_PP_NO_MMA   // tuning build: K loop without MFMAs
    return;
#endif
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < QM; ++i)
#pragma unroll
        for (int jj = 0; jj < QN; ++jj) mfma_bf16_acc_t(acc[h * QM + i][j * QN + jj], af[i][s], bf[jj][s]);
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- prologue: tiles 0 (all four halves) and 1 (A0, B0) ---------------------------------
  PPCursor ce{0, 0, 0}, co{1, 1, 0};
  if (co.tap >= a.ntaps) { co.tap = 0; co.c0 = 64; }
  issue_a(0, 0, ce); issue_b(0, 0, ce); issue_b(0, 1, ce); issue_a(0, 1, ce);
  advance2(ce);
  issue_a(1, 0, co); issue_b(1, 0, co);
  pp_wait_vm<8>();           // A0, B0 of tile 0 have landed (mine)
  pp_barrier();              // ... and everyone's
  if (grp == 1) pp_barrier(); // lower-row group runs one barrier behind

  // one phase: [reads] [issue] [vmcnt] | barrier | lgkmcnt(0) MFMA | barrier
#ifdef VINET_CONV_TIMING
  unsigned long long tacc[4] = {0, 0, 0, 0}, tq0, tq1;
#define PP_T0 tq0 = __builtin_amdgcn_s_memtime();
#define PP_T(k) { tq1 = __builtin_amdgcn_s_memtime(); tacc[k] += tq1 - tq0; tq0 = tq1; }
#else
#define PP_T0
#define PP_T(k)
#endif
#define PP_PHASE(READS, ISSUE, MMA)                 \
  {                                                  \
    PP_T0                                            \
    READS;                                           \
    ISSUE;                                           \
    pp_wait_vm<8>();                                 \
    PP_T(0)                                          \
    __builtin_amdgcn_sched_barrier(0);               \
    pp_barrier();                                    \
    PP_T(1)                                          \
    pp_wait_lgkm0();                                 \
    __builtin_amdgcn_sched_barrier(0);               \
    MMA;                                             \
    __builtin_amdgcn_sched_barrier(0);               \
    PP_T(2)                                          \
    pp_barrier();                                    \
    PP_T(3)                                          \
  }

  const int niter = (nkt + 1) >> 1;
  for (int it = 0; it < niter; ++it) {
    PP_PHASE((read_a(0, 0), read_b(0, 0, b0)), issue_b(1, 1, co), mma(0, 0, b0));
    PP_PHASE(read_b(0, 1, b1), issue_a(1, 1, co), mma(0, 1, b1));
    advance2(co);
    PP_PHASE(read_a(0, 1), issue_a(0, 0, ce), mma(1, 1, b1));
    PP_PHASE((void)0, issue_b(0, 0, ce), mma(1, 0, b0));
    // (a missing last odd tile was staged from the zero page: its MFMAs add zeros)
    PP_PHASE((read_a(1, 0), read_b(1, 0, b0)), issue_b(0, 1, ce), mma(0, 0, b0));
    PP_PHASE(read_b(1, 1, b1), issue_a(0, 1, ce), mma(0, 1, b1));
    advance2(ce);
    PP_PHASE(read_a(1, 1), issue_a(1, 0, co), mma(1, 1, b1));
    PP_PHASE((void)0, issue_b(1, 0, co), mma(1, 0, b0));
  }
#undef PP_PHASE
#ifdef VINET_CONV_TIMING
  if ((tid & 255) == 0 && a.out_shift) {   // tuning build: out_shift doubles as a [grid][2 groups][4] dump
    float* dbg = (float*)a.out_shift + ((long)blockIdx.x * 2 + grp) * 4;
    for (int k = 0; k < 4; ++k) dbg[k] = (float)tacc[k] / (float)(niter * 8);
  }
#endif
  if (grp == 0) pp_barrier();   // re-align the two groups
  pp_wait_vm<0>();             // tail DMAs (zero page) before LDS is reused
  mfma_drain();
  __syncthreads();
  conv_epilogue<MT, NT, WM, WN>(a, acc, smem, tile_m, tile_n);
}

template <int WM, int WN, int BN_>
static int launch_conv_pp_cfg(const ConvArgs& a, hipStream_t s) {
  using Cfg = ConvPPCfg<WM, WN, BN_>;
  auto kern = conv_pp_kernel<WM, WN, BN_>;
  static bool attr_done[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_done[dev & 63]) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM);
    if (e != hipSuccess) { vinet_set_error("hipFuncSetAttribute(conv_pp): %s", hipGetErrorString(e)); return (int)e; }
    attr_done[dev & 63] = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.tilesM * a.tilesN), dim3(Cfg::THREADS), Cfg::SMEM, s, a);
  return vn_launch_status("conv_pp");
}
