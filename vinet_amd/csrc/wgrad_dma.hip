// Weight gradient, LDS-DMA multi-tap kernel (bf16, gfx950).
//
//   dw[slice_g][n][c] += sum_m dy[m][n] * pre(x[m (+) tap_g])[c]      for the TG taps of a group
//
// One workgroup (4 waves, 2x2) owns a TN x TC tile of (n, c), a group of up to
// TG taps and a split-K range of voxel chunks (32 voxels each).  Per chunk the
// dY tile is fetched ONCE and reused for every tap of the group (the 64x64
// single-tap kernel re-streamed dY and X once per tap: 7x for the stem's
// 7x1x1 conv); all TG x (TN x TC) accumulators live in registers.
//
//   * operands go HBM/L2 -> LDS with global_load_lds_dwordx4 into a STAGES-slot
//     ring ([32 voxel rows][TN or TC channels], rows unpadded); the 16-byte
//     chunk index is XOR-swizzled with the row so the K-major fragment reads
//     (ds_read_b64_tr_b16, hardware transpose) do not serialise on banks;
//   * out-of-range sources read a zero page (dY, plain X) or a NaN page (X
//     with a pending BN+ReLU: pre_relu_pair maps the sign-bit NaN to 0), so every wave issues
//     exactly LPS DMAs per stage and the counted vmcnt wait is exact;
//   * the pending affine is applied at fragment time: a B fragment holds 8
//     voxels of ONE channel per lane, so scale/shift are two scalars per lane;
//   * voxel index -> (b,t,h,w) uses multiply-high fast division.
#include <type_traits>

#include "common.h"

extern int g_vinet_opt_wgrad_tg;
extern int g_vinet_opt_tperm;

// (device globals are per translation unit without -fgpu-rdc: own copies of the pad pages)
__device__ __attribute__((aligned(64))) uint4 g_wg_zero_page[4];
__device__ __attribute__((aligned(64))) uint4 g_wg_nan_page[4] = {
    {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu},
    {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}};

struct WgradDmaArgs {
  const char* x;
  const char* dy;
  float* dw;
  const int4* taps;
  const float* in_scale;
  const float* in_shift;
  int Ti, Hi, Wi, Cin, ldx;
  long sBx;
  int To, Ho, Wo, N, ldy;
  long sBy;
  int sT, sH, sW;
  int ntaps, Kp, M;
  int tilesN, tilesC, splitK, chunks_per_split, nchunks;
  FastDiv dW, dH, dT;
  int perm_P, perm_T;        // voxel-chunk order with t fastest (see ConvArgs::perm_P); 0 = identity
  FastDiv dPT, dPermT;
};

// logical chunk index -> chunk position in memory order: (b, spatial chunk c, t) with t fastest, so the
// temporal taps of consecutive K steps re-read the same (h,w) rows of neighbouring frames from L1/L2
VN_DEV int wg_chunk_perm(const WgradDmaArgs& a, int i) {
  if (a.perm_P == 0) return i;
  const uint32_t b = fdiv((uint32_t)i, a.dPT);
  const uint32_t rem = (uint32_t)i - b * (uint32_t)(a.perm_P * a.perm_T);
  const uint32_t c = fdiv(rem, a.dPermT);
  const uint32_t t = rem - c * (uint32_t)a.perm_T;
  return (int)((b * (uint32_t)a.perm_T + t) * (uint32_t)a.perm_P + c);
}

template <int N_> VN_DEV void wg_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }
VN_DEV uint32_t wg_cvt_pk_bf16(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

template <int TN, int TC, int TG, int STAGES, bool PRE>
struct WgCfg {
  static constexpr int KV = 32;
  static constexpr int D_BYTES = KV * TN * 2, X_BYTES = KV * TC * 2;
  static constexpr int STAGE_BYTES = D_BYTES + TG * X_BYTES + (TC == 32 ? 1024 : 0);   // (+ dummy piece)
  static constexpr int D_IPW = (TN / 64);          // DMA instructions per wave for the dY tile (KV*TN*2/1024/4)
  static constexpr int X_IPW = (TC / 64);
  // TC == 32 (the folded stem, Cin = 32): an X tile is 32 rows x 64 B = two wave-instructions; the TG*2 pieces of
  // a stage are dealt round-robin to the four waves (pieces past the end are zero-page dummies into a scratch
  // piece behind the stage) so that every wave still issues the same number of DMAs
  static constexpr int X32_PPW = (TG * 2 + 3) / 4;
  static constexpr int LPS = D_IPW + (TC == 32 ? X32_PPW : TG * X_IPW);
  static constexpr int SMEM = STAGES * STAGE_BYTES;
  static constexpr int MT = TN / 32, NT = TC / 32;  // 16x16 fragments per wave (2x2 waves)
};

// tile row r (0..31) of a [32][TW channels] bf16 tile, 16-byte chunk ch -> byte offset with swizzle
template <int TW> VN_DEV int wg_swz(int r) {
  if (TW == 32) return ((r >> 3) & 1) << 1;   // 64-byte rows: rows r and r+8 share banks, move one to the other half
  return TW == 64 ? ((((r >> 1) & 1) | (((r >> 3) & 1) << 1)) << 1) : ((r & 3) << 1);   // (64: bits 1 and 3, see wgrad_rs.hip)
}

template <int TN, int TC, int TG, int STAGES, bool PRE>
__global__ __launch_bounds__(256) void conv_wgrad_dma_kernel(const WgradDmaArgs a) {
  using Cfg = WgCfg<TN, TC, TG, STAGES, PRE>;
  constexpr int MT = Cfg::MT, NT = Cfg::NT, KV = 32;
  static_assert(Cfg::LPS * (STAGES - 2) <= 63 && STAGES >= 2, "vmcnt range");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tile_c = blockIdx.x % a.tilesC, tile_n = blockIdx.x / a.tilesC;
  const int tap0 = blockIdx.y * TG;
  const int chunk0 = blockIdx.z * a.chunks_per_split;
  int chunk1 = chunk0 + a.chunks_per_split;
  if (chunk1 > a.nchunks) chunk1 = a.nchunks;
  const int nloc = chunk1 - chunk0;
  const int n0 = tile_n * TN, c0 = tile_c * TC;
  const char* zero = (const char*)g_wg_zero_page;
  const char* xpad = PRE ? (const char*)g_wg_nan_page : zero;

  // taps of this group (scalar registers); invalid ones are flagged
  int4 tp[TG];
  bool tap_ok[TG];
#pragma unroll
  for (int g = 0; g < TG; ++g) {
    tap_ok[g] = tap0 + g < a.ntaps;
    tp[g] = load_tap(a.taps, tap_ok[g] ? tap0 + g : tap0);
  }

  // DMA roles.  dY tile: rows of TN*2 bytes; X tiles: rows of TC*2 bytes.
  constexpr int D_CPR = TN * 2 / 16, D_RPI = 64 / D_CPR;   // chunks per row, rows per wave-instruction
  constexpr int X_CPR = TC * 2 / 16, X_RPI = 64 / X_CPR;
  const int d_lr = lane / D_CPR, d_ch = lane % D_CPR;
  const int x_lr = lane / X_CPR, x_ch = lane % X_CPR;

  int isu = 0;   // local chunk counter being issued
  auto issue = [&](int slot) {
    char* stage = smem + slot * Cfg::STAGE_BYTES;
    const bool live = isu < nloc;
    const int mbase = wg_chunk_perm(a, live ? chunk0 + isu : 0) * KV;
    // ---- dY tile ----
#pragma unroll
    for (int j = 0; j < Cfg::D_IPW; ++j) {
      const int q = wave + 4 * j;                 // wave-instruction index within the tile
      const int r = q * D_RPI + d_lr;             // tile row = voxel within the chunk
      const int m = mbase + r;
      const uint32_t t1 = fdiv((uint32_t)m, a.dW);
      const int wo = m - (int)t1 * a.Wo;
      const uint32_t t2 = fdiv(t1, a.dH);
      const int ho = (int)t1 - (int)t2 * a.Ho;
      const uint32_t b = fdiv(t2, a.dT);
      const int to = (int)t2 - (int)b * a.To;
      const int sch = d_ch ^ wg_swz<TN>(r);
      const int n = n0 + sch * 8;
      const unsigned ok = (unsigned)live & (unsigned)(m < a.M) & (unsigned)(n < a.N);
      const char* p = a.dy + ((long)b * a.sBy + ((long)(to * a.Ho + ho) * a.Wo + wo) * (long)a.ldy + n) * 2;
      const char* src = zero + ((p - zero) & -(long)ok);
      lds_dma16_asm(src, (stage + q * 1024));      // (asm: common.h -- no compiler-made vmcnt(0) drains in front of the transpose reads)
    }
    // ---- X tiles, one per tap of the group ----
    if constexpr (TC == 32) {
      // piece pc = wave + 4j -> half q = pc & 1 = wave & 1 for every piece of this wave: one voxel decode per chunk
      const int q = wave & 1;
      const int r = q * X_RPI + x_lr;
      const int m = mbase + r;
      const uint32_t t1 = fdiv((uint32_t)m, a.dW);
      const int wo = m - (int)t1 * a.Wo;
      const uint32_t t2 = fdiv(t1, a.dH);
      const int ho = (int)t1 - (int)t2 * a.Ho;
      const uint32_t b = fdiv(t2, a.dT);
      const int to = (int)t2 - (int)b * a.To;
      const int sch = x_ch ^ wg_swz<TC>(r);
      const int c = c0 + sch * 8;
      const unsigned rowok = (unsigned)live & (unsigned)(m < a.M) & (unsigned)(c < a.Cin);
      const long base = (long)b * a.sBx + c;
      const int t0 = to * a.sT, h0 = ho * a.sH, w0 = wo * a.sW;
#pragma unroll
      for (int j = 0; j < Cfg::X32_PPW; ++j) {
        const int g = (wave >> 1) + 2 * j;           // tap of piece wave + 4j
        const bool pok = g < TG;
        int4 tg = tp[0];
        bool tok = tap_ok[0];
#pragma unroll
        for (int k = 1; k < TG; ++k)
          if (g == k) { tg = tp[k]; tok = tap_ok[k]; }
        const int ti = t0 + tg.x, hi = h0 + tg.y, wi = w0 + tg.z;
        const unsigned ok = rowok & (unsigned)pok & (unsigned)tok & (unsigned)((unsigned)ti < (unsigned)a.Ti) &
                            (unsigned)((unsigned)hi < (unsigned)a.Hi) & (unsigned)((unsigned)wi < (unsigned)a.Wi);
        const char* p = a.x + (base + ((long)(ti * a.Hi + hi) * a.Wi + wi) * (long)a.ldx) * 2;
        const char* src = xpad + ((p - xpad) & -(long)ok);
        char* dst = pok ? stage + Cfg::D_BYTES + g * Cfg::X_BYTES + q * 1024 : stage + Cfg::D_BYTES + TG * Cfg::X_BYTES;
        lds_dma16_asm(src, dst);      // (asm: common.h -- no compiler-made vmcnt(0) drains in front of the transpose reads)
      }
    }
#pragma unroll
    for (int j = 0; j < Cfg::X_IPW; ++j) {
      const int q = wave + 4 * j;
      const int r = q * X_RPI + x_lr;
      const int m = mbase + r;
      const uint32_t t1 = fdiv((uint32_t)m, a.dW);
      const int wo = m - (int)t1 * a.Wo;
      const uint32_t t2 = fdiv(t1, a.dH);
      const int ho = (int)t1 - (int)t2 * a.Ho;
      const uint32_t b = fdiv(t2, a.dT);
      const int to = (int)t2 - (int)b * a.To;
      const int sch = x_ch ^ wg_swz<TC>(r);
      const int c = c0 + sch * 8;
      const unsigned rowok = (unsigned)live & (unsigned)(m < a.M) & (unsigned)(c < a.Cin);
      const long base = (long)b * a.sBx + c;
#pragma unroll
      for (int g = 0; g < TG; ++g) {
        const int ti = to * a.sT + tp[g].x, hi = ho * a.sH + tp[g].y, wi = wo * a.sW + tp[g].z;
        const unsigned ok = rowok & (unsigned)tap_ok[g] & (unsigned)((unsigned)ti < (unsigned)a.Ti) &
                            (unsigned)((unsigned)hi < (unsigned)a.Hi) & (unsigned)((unsigned)wi < (unsigned)a.Wi);
        const char* p = a.x + (base + ((long)(ti * a.Hi + hi) * a.Wi + wi) * (long)a.ldx) * 2;
        const char* src = xpad + ((p - xpad) & -(long)ok);
        lds_dma16_asm(src, (stage + Cfg::D_BYTES + g * Cfg::X_BYTES + q * 1024));      // (asm: common.h -- no compiler-made vmcnt(0) drains in front of the transpose reads)
      }
    }
    ++isu;
  };

  // per-lane scale/shift of the channels its B fragments hold
  float sc[NT], sh[NT];
  if constexpr (PRE) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int c = c0 + wn * (TC / 2) + j * 16 + (lane & 15);
      sc[j] = c < a.Cin ? a.in_scale[c] : 0.f;
      sh[j] = c < a.Cin ? a.in_shift[c] : 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // plain loads done before DMAs are counted
  }

  f32x4_v acc[TG][MT][NT];
#pragma unroll
  for (int g = 0; g < TG; ++g)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[g][i][j] = (f32x4_v){0.f, 0.f, 0.f, 0.f};

  // K-major fragment (8 voxels x 1 channel per lane) of a [32][TW] tile by two transpose reads
  auto frag = [&](const char* tile, int col0, auto twc) -> bf16x8_v {
    constexpr int TW = decltype(twc)::value;
    union { bf16x8_v v; s16x4_v h[2]; } u;
    const int p = lane & 15;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int krow = (lane >> 4) * 8 + h * 4 + (p >> 2);
      const int col = col0 + (p & 3) * 4;                       // element column
      const int ch = (col >> 3) ^ wg_swz<TW>(krow);             // swizzled 16-byte chunk
      const char* src = tile + krow * (TW * 2) + ch * 16 + (col & 7) * 2;
      u.h[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_v*)src);
    }
    return u.v;
  };

  auto compute = [&](int slot) {
    const char* stage = smem + slot * Cfg::STAGE_BYTES;
    bf16x8_v af[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) af[i] = frag(stage, wm * (TN / 2) + i * 16, std::integral_constant<int, TN>());
#pragma unroll
    for (int g = 0; g < TG; ++g) {
      const char* xt = stage + Cfg::D_BYTES + g * Cfg::X_BYTES;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        bf16x8_v bf = frag(xt, wn * (TC / 2) + j * 16, std::integral_constant<int, TC>());
        if constexpr (PRE) {
          union { bf16x8_v v; uint32_t w[4]; } q;
          q.v = bf;
#pragma unroll
          for (int e = 0; e < 4; ++e) q.w[e] = pre_relu_pair(q.w[e], (f32x2_v){sc[j], sc[j]}, (f32x2_v){sh[j], sh[j]});
          bf = q.v;
          valu_to_mfma_pad();
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
          mfma_bf16_acc(acc[g][i][j], af[i], bf);
      }
    }
  };

  // ---- pipeline (same scheme as conv_dma_kernel) ----------------------------------
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s) issue(s);
  int slot = 0, fill = STAGES - 1;
  for (int it = 0; it < nloc; ++it) {
    wg_wait_vmcnt<Cfg::LPS*(STAGES - 2)>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(fill);
    compute(slot);
    asm volatile("" ::: "memory");
    slot = slot + 1 == STAGES ? 0 : slot + 1;
    fill = fill + 1 == STAGES ? 0 : fill + 1;
  }
  wg_wait_vmcnt<0>();
  mfma_drain();

#pragma unroll
  for (int g = 0; g < TG; ++g) {
    if (!tap_ok[g]) continue;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + wm * (TN / 2) + i * 16 + (lane >> 4) * 4 + r;
          const int c = c0 + wn * (TC / 2) + j * 16 + (lane & 15);
          if (n < a.N && c < a.Kp) {
            float* dst = a.dw + ((long)tp[g].w * a.N + n) * (long)a.Kp + c;
            atomicAdd(dst, acc[g][i][j][r]);      // (always +=: dw is zero on entry, or holds the other terms of a split-bf16 sum / an earlier use of a shared weight)
          }
        }
  }
}

template <int TN, int TC, int TG, int STAGES, bool PRE>
static int launch_wg(WgradDmaArgs& a, hipStream_t s) {
  using Cfg = WgCfg<TN, TC, TG, STAGES, PRE>;
  auto kern = conv_wgrad_dma_kernel<TN, TC, TG, STAGES, PRE>;
  static bool attr_done[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_done[dev & 63]) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM);
    if (e != hipSuccess) { vinet_set_error("hipFuncSetAttribute(wgrad_dma): %s", hipGetErrorString(e)); return (int)e; }
    attr_done[dev & 63] = true;
  }
  a.tilesN = vn_div_up(a.N, TN);
  a.tilesC = vn_div_up(a.Cin, TC);
  const int groups = vn_div_up(a.ntaps, TG);
  a.nchunks = vn_div_up(a.M, 32);
  const long base_blocks = (long)a.tilesN * a.tilesC * groups;
  long sk = (1024 + base_blocks - 1) / base_blocks;        // ~4 workgroups per CU
  // every split adds TG*TN*TC fp32 atomics: keep at least 32 chunks (1024 voxels) of work behind them
  if (sk > a.nchunks / 32) sk = a.nchunks / 32;
  if (sk < 1) sk = 1;
  if (sk > 2048) sk = 2048;
  a.chunks_per_split = vn_div_up(a.nchunks, sk);
  a.splitK = vn_div_up(a.nchunks, a.chunks_per_split);
  hipLaunchKernelGGL(kern, dim3(a.tilesN * a.tilesC, groups, a.splitK), dim3(256), Cfg::SMEM, s, a);
  return vn_launch_status("conv_wgrad_dma");
}

// pick (tile, taps per group) from the conv geometry
extern int g_vinet_opt_wgrad_tg;
// Taps per group, from in-process A/B on ViNet layer shapes (tools/conv_ab.py --wgrad):
// occupancy beats dY reuse -- 3 taps per group (3 workgroups/CU) is as fast or faster than 9
// (1 workgroup/CU, LDS-limited) except on the very large-M 3x3 layers, and small-M layers
// want single-tap groups (more workgroups to fill 256 CUs).  64x64 tiles throughout.
static const char* wg_pick(int N, int Cin, int ntaps, long M, int* tn, int* tg) {
  *tn = 64;
  if (g_vinet_opt_wgrad_tg > 0 && ntaps % g_vinet_opt_wgrad_tg == 0) { *tg = g_vinet_opt_wgrad_tg; return nullptr; }
  if (ntaps == 1 || M < 16384) *tg = 1;
  else if (ntaps == 7) *tg = 7;
  else if (ntaps % 9 == 0 && M >= 500000) *tg = 9;
  else if (ntaps % 3 == 0) *tg = 3;
  else if (ntaps % 2 == 0) *tg = 2;
  else *tg = 1;
  return nullptr;
}

int vinet_wgrad_dma_name(const VinetWgradDesc* d, char* buf, int n) {
  int tn, tg;
  wg_pick(d->dy.C, d->x.C, d->ntaps, (long)d->dy.B * d->dy.T * d->dy.H * d->dy.W, &tn, &tg);
  const int tc = (d->x.C <= 32 && tn == 64 && tg == 7 && !d->pre.scale) ? 32 : tn;
  snprintf(buf, n, "conv_wgrad_dma_kernel<%d,%d,%d,%s>", tn, tc, tg, d->pre.scale ? "pre" : "plain");
  return 0;
}

int vinet_launch_wgrad_dma(const VinetWgradDesc* d, hipStream_t s) {
  WgradDmaArgs a;
  a.x = (const char*)d->x.ptr; a.dy = (const char*)d->dy.ptr; a.dw = d->dw; a.taps = (const int4*)d->taps;
  a.in_scale = d->pre.scale; a.in_shift = d->pre.shift;
  a.Ti = d->x.T; a.Hi = d->x.H; a.Wi = d->x.W; a.Cin = d->x.C; a.ldx = d->x.ld; a.sBx = d->x.sB;
  a.To = d->dy.T; a.Ho = d->dy.H; a.Wo = d->dy.W; a.N = d->dy.C; a.ldy = d->dy.ld; a.sBy = d->dy.sB;
  a.sT = d->sT; a.sH = d->sH; a.sW = d->sW;
  a.ntaps = d->ntaps; a.Kp = d->Kp;
  a.M = (int)((long)d->dy.B * d->dy.T * d->dy.H * d->dy.W);
  a.dW = make_fastdiv(a.Wo); a.dH = make_fastdiv(a.Ho); a.dT = make_fastdiv(a.To);
  a.perm_P = a.perm_T = 0;
  a.dPT = a.dPermT = make_fastdiv(1);
  if (g_vinet_opt_tperm && a.To > 1 && ((long)a.Ho * a.Wo) % 32 == 0) {
    a.perm_P = (int)(((long)a.Ho * a.Wo) / 32);
    a.perm_T = a.To;
    a.dPT = make_fastdiv((uint32_t)(a.perm_P * a.perm_T));
    a.dPermT = make_fastdiv((uint32_t)a.perm_T);
  }
  int tn, tg;
  wg_pick(a.N, a.Cin, a.ntaps, a.M, &tn, &tg);
  const bool pre = d->pre.scale != nullptr;
  if (a.Cin <= 32 && tn == 64 && tg == 7 && !pre) return launch_wg<64, 32, 7, 2, false>(a, s);   // the folded stem (4 stages / 2 workgroups per CU measured 1.7x slower:
                                                                                                   // the kernel is bound by its address VALU, more resident waves win)
#define WG(TN_, TG_, ST_) \
  if (tn == TN_ && tg == TG_) return pre ? launch_wg<TN_, TN_, TG_, ST_, true>(a, s) : launch_wg<TN_, TN_, TG_, ST_, false>(a, s);
  WG(128, 1, 3) WG(64, 1, 3) WG(64, 2, 3) WG(64, 3, 3) WG(64, 7, 2) WG(64, 9, 2)
#undef WG
  vinet_set_error("wgrad dma: no kernel for tile %d taps/group %d", tn, tg);
  return -1;
}
