// Weight gradient, 256 x 256 x 64 "ping-pong" kernel (bf16, gfx950).
//
//   dw[slice(tap)][n][c] += sum_m dy[m][n] * pre(x[m (+) tap])[c]
//
// as a GEMM whose reduction runs over the output voxels m:  rows = 256 output channels n,
// columns = four 64-channel SEGMENTS (tap, c0) of the (tap, input channel) space, K tile = 64
// voxels.  Same machinery as conv_pp.h -- 8 waves in two groups one barrier apart, two LDS
// buffers of four 16 KB half-tiles, one half-tile (2 LDS-DMAs per lane, whole 128-byte lines)
// issued per phase, `s_waitcnt vmcnt(8)` -- but both operands are stored voxel-major
// ([64 voxels][128 channels] per half-tile), so the K-major MFMA fragments come from
// ds_read_b64_tr_b16 transpose reads.  The 64x64 kernel of wgrad_dma.hip stages 20 B per kFLOP
// and is bound by the CU's load path (250-390 TF/s); this one stages 7.8 B per kFLOP.
//
//   half-tile A_h : n' = wr*64 + r      <-> n = n0 + wr*128 + h*64 + r           (dy)
//   half-tile B_j : c' = g*64 + cc      <-> segment 4*tile + 2*g + j, channel c0 + cc
//                   wave column wc = 2*g + p owns cc in [p*32, p*32+32) of both its segments
//   LDS row = 256 B = 16 chunks; chunk k of voxel row r is stored at k ^ swz(r),
//   swz(r) = ((r & 3) | ((r >> 3) & 1) << 2) << 1: the 8 rows a 32-lane transpose read touches
//   land in 8 different 32-byte windows (conflict-free).
//
// A pending BN+ReLU on x is applied at fragment time (8 voxels of ONE channel per lane: two
// scalars), with the NaN-page padding trick of conv_dma.h.  Split-K over voxel ranges with fp32
// atomics, as in wgrad_dma.hip.
#include "common.h"

__device__ __attribute__((aligned(64))) uint4 g_wpp_zero_page[4];
__device__ __attribute__((aligned(64))) uint4 g_wpp_nan_page[4] = {
    {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu},
    {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}};

struct WgradPPArgs {
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
  int tilesN, tilesS, nseg, cpt;     // column tiles of 4 segments; segments = ntaps * cpt, cpt = ceil(Cin/64)
  int nkt, kt_per_split, splitK;
  FastDiv dW, dH, dT, dCpt;
  int perm_P, perm_T;        // K-tile order with t fastest (see ConvArgs::perm_P); 0 = identity
  FastDiv dPT, dPermT;
#ifdef VINET_CONV_TIMING
  float* dbg;   // tuning build: [grid][2 groups][4] mean cycles per phase part
#endif
};

struct WppCfg {
  static constexpr int THREADS = 512;
  static constexpr int HALF_BYTES = 64 * 256;        // 16 KB
  static constexpr int BUF_BYTES = 4 * HALF_BYTES;   // A0 A1 B0 B1
  static constexpr int TAB_OFF = 2 * BUF_BYTES;      // PRE: scale[4 segments][64], shift[4][64] (fp32)
  static constexpr int SMEM = 2 * BUF_BYTES + 2 * 256 * 4;
  static constexpr int A_OFF = 0, B_OFF = 2 * HALF_BYTES;
};

VN_DEV void wpp_barrier() { asm volatile("s_barrier" ::: "memory"); }
template <int N> VN_DEV void wpp_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
VN_DEV void wpp_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
VN_DEV uint32_t wpp_cvt_pk_bf16(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// the two voxel rows (q = 0, 1) of this lane in K tile `kt`: byte offset of x at tap (0,0,0) and validity bits
// (bit q*2+j: x row q is in range for the tap of half j; bit 4+q: the row exists).  dy is addressed as
// m * ldy (linear views only), so it needs no per-row state.
struct WppCursor {
  int kt, mb;                // logical K tile, first voxel of the tile in memory order
  long xo[2];
  unsigned bits;
};

// QM = 16-row fragments per wave and row quadrant: 4 -> 256 output channels per tile, 3 -> 192 (ViNet's
// most common width).  The LDS image keeps 64 columns per wave row either way; with QM = 3 the last 16
// are never fetched (their DMA lanes read the zero page) nor read.
template <bool PRE, int QM>
__global__ __launch_bounds__(512, 2) void conv_wgrad_pp_kernel(const WgradPPArgs a) {
  using Cfg = WppCfg;
  constexpr int TN = 64 * QM;      // rows (output channels) per tile
  constexpr int WR = 32 * QM;      // ... per wave row
  constexpr int HR = 16 * QM;      // ... per wave row and half
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3, grp = wave >> 2;
  const int g_w = wc >> 1, p_w = wc & 1;
  // Workgroups are dealt to the 8 XCDs round-robin in dispatch order: remap so that the column / row tiles of ONE
  // voxel range (which stage the same dy rows and overlapping x rows) are neighbours on one XCD and share its L2
  const int wg_lin = xcd_remap((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
  const int bx = wg_lin % (int)gridDim.x, by = wg_lin / (int)gridDim.x;
  const int tile_s = bx % a.tilesS, tile_n = bx / a.tilesS;
  const int kt0 = by * a.kt_per_split;
  int kt1 = kt0 + a.kt_per_split;
  if (kt1 > a.nkt) kt1 = a.nkt;
  const int n0 = tile_n * TN, seg0 = tile_s * 4;
  const char* zero = (const char*)g_wpp_zero_page;
  const char* xpad = PRE ? (const char*)g_wpp_nan_page : zero;

  // ---- this lane's DMA role: piece = 4 voxel rows x 256 B; lane -> row (lane>>4), slot (lane&15) ----
  const int prow = wave * 4 + (lane >> 4);                       // voxel row within a 32-row block (q adds 32)
  const int swz_d = (((prow & 3) | (((prow >> 3) & 1) << 2)) << 1);
  const int sch = (lane & 15) ^ swz_d;                           // source chunk of this lane's LDS slot
  // A (dy): chunk -> wave row sch>>3, 8 channels (sch&7)*8 of its 64
  int a_noff[2];
  unsigned a_nok[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int n = n0 + (sch >> 3) * WR + h * HR + (sch & 7) * 8;
    a_nok[h] = (unsigned)(n < a.N) & (unsigned)((sch & 7) * 8 < HR);
    a_noff[h] = n * 2;
  }
  const long ldy2 = (long)a.ldy * 2;
  // B (x): chunk -> segment pair sch>>3, 8 channels (sch&7)*8 of the segment; half j picks the segment
  int b_off[2];
  int b_dt[2], b_dh[2], b_dw[2];
  unsigned b_ok[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int seg = seg0 + (sch >> 3) * 2 + j;
    const bool sok = seg < a.nseg;
    const int tap = sok ? (int)fdiv((uint32_t)seg, a.dCpt) : 0;
    const int c = ((sok ? seg : 0) - tap * a.cpt) * 64 + (sch & 7) * 8;
    const int4 tp = a.taps[tap];
    b_dt[j] = tp.x; b_dh[j] = tp.y; b_dw[j] = tp.z;
    b_ok[j] = (unsigned)sok & (unsigned)(c < a.Cin);
    b_off[j] = (int)((((long)(tp.x * a.Hi + tp.y) * a.Wi + tp.z) * (long)a.ldx + c) * 2);
  }

  // ---- per-lane constants of the B fragments: output segment / channel, pending affine --------------
  // scale / shift of the tile's 4 x 64 channels live in LDS (a per-lane copy costs 8 VGPRs this kernel
  // does not have); lane reads entry (g_w*2 + j)*64 + p_w*32 + jj*16 + (lane & 15)
  float* tab = (float*)(smem + Cfg::TAB_OFF);
  const float* tab_lane = tab + g_w * 128 + p_w * 32 + (lane & 15);
  if constexpr (PRE) {
    if (tid < 256) {
      const int seg = seg0 + (tid >> 6);
      const int tap = seg < a.nseg ? (int)fdiv((uint32_t)seg, a.dCpt) : 0;
      const int c = (seg - tap * a.cpt) * 64 + (tid & 63);
      const bool ok = seg < a.nseg && c < a.Cin;
      tab[tid] = ok ? a.in_scale[c] : 0.f;
      tab[256 + tid] = ok ? a.in_shift[c] : 0.f;
    }
    __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // plain loads are done before any DMA is counted

  auto set_cursor = [&](WppCursor& c, int kt) {
    c.kt = kt;
    {
      int pk = kt < kt1 ? kt : 0;
      if (a.perm_P != 0) {
        const uint32_t b = fdiv((uint32_t)pk, a.dPT);
        const uint32_t rem = (uint32_t)pk - b * (uint32_t)(a.perm_P * a.perm_T);
        const uint32_t cc = fdiv(rem, a.dPermT);
        pk = (int)((b * (uint32_t)a.perm_T + (rem - cc * (uint32_t)a.perm_T)) * (uint32_t)a.perm_P + cc);
      }
      c.mb = pk * 64;
    }
    c.bits = 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int m = c.mb + q * 32 + prow;
      const bool live = (kt < kt1) & (m < a.M);
      int b, to, ho, wo;
      decode_m(live ? m : 0, a.dW, a.dH, a.dT, b, to, ho, wo);
      const int ti = to * a.sT, hi = ho * a.sH, wi = wo * a.sW;
      c.xo[q] = ((long)b * a.sBx + ((long)(ti * a.Hi + hi) * a.Wi + wi) * (long)a.ldx) * 2;
      c.bits |= (unsigned)live << (4 + q);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned ok = (unsigned)live & b_ok[j] & (unsigned)((unsigned)(ti + b_dt[j]) < (unsigned)a.Ti) &
                            (unsigned)((unsigned)(hi + b_dh[j]) < (unsigned)a.Hi) & (unsigned)((unsigned)(wi + b_dw[j]) < (unsigned)a.Wi);
        c.bits |= ok << (q * 2 + j);
      }
    }
  };
  auto dma = [&](const char* src, char* dst) { lds_dma16_asm(src, dst); };      // (asm: see common.h -- no compiler-made vmcnt(0) drains)
  auto issue_a = [&](int buf, int h, const WppCursor& c) {
    char* dst = smem + buf * Cfg::BUF_BYTES + Cfg::A_OFF + h * Cfg::HALF_BYTES + wave * 1024;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const unsigned ok = ((c.bits >> (4 + q)) & 1u) & a_nok[h];
      const char* p = a.dy + (long)(c.mb + q * 32 + prow) * ldy2 + a_noff[h];
      dma(zero + ((p - zero) & -(long)ok), dst + q * 8192);
    }
  };
  auto issue_b = [&](int buf, int j, const WppCursor& c) {
    char* dst = smem + buf * Cfg::BUF_BYTES + Cfg::B_OFF + j * Cfg::HALF_BYTES + wave * 1024;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const unsigned ok = (c.bits >> (q * 2 + j)) & 1u;
      const char* p = a.x + c.xo[q] + b_off[j];
      dma(xpad + ((p - xpad) & -(long)ok), dst + q * 8192);
    }
  };

  // ---- fragments (transpose reads) ------------------------------------------------------------------
  f32x4_v acc[2 * QM][4];
#pragma unroll
  for (int i = 0; i < 2 * QM; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_v){0.f, 0.f, 0.f, 0.f};
  bf16x8_v af[QM][2], b0[2][2], b1[2][2];

  // lane -> voxel row kb*8 + hh*4 + pr (kb = lane>>4, pr = (lane&15)>>2), 4 channels (lane&3)*4 of a 16-wide block
  const int pl = lane & 15, kb = lane >> 4, pr = pl >> 2;
  const int swz_f = ((pr | ((kb & 1) << 2)) << 1);
  const int f_row = (kb * 8 + pr) * 256 + ((pl & 3) >> 1) * 16 + (pl & 1) * 8;   // + hh*1024 + s*8192
  int fa[QM], fb[2];
#pragma unroll
  for (int i = 0; i < QM; ++i) fa[i] = f_row + (((wr * 8 + i * 2) ^ swz_f) << 4);
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) fb[jj] = f_row + (((g_w * 8 + p_w * 4 + jj * 2) ^ swz_f) << 4);

  auto tr_frag = [&](const char* tile, int off, int s) -> bf16x8_v {
    union { bf16x8_v v; s16x4_v h[2]; } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_v*)(tile + off + s * 8192));
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_v*)(tile + off + s * 8192 + 1024));
    return u.v;
  };
  auto read_a = [&](int buf, int h) {
    const char* tile = smem + buf * Cfg::BUF_BYTES + Cfg::A_OFF + h * Cfg::HALF_BYTES;
#pragma unroll
    for (int i = 0; i < QM; ++i) {
      af[i][0] = tr_frag(tile, fa[i], 0);
      af[i][1] = tr_frag(tile, fa[i], 1);
    }
  };
  auto read_b = [&](int buf, int j, bf16x8_v (&bf)[2][2]) {
    const char* tile = smem + buf * Cfg::BUF_BYTES + Cfg::B_OFF + j * Cfg::HALF_BYTES;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      bf[jj][0] = tr_frag(tile, fb[jj], 0);
      bf[jj][1] = tr_frag(tile, fb[jj], 1);
    }
  };
  // pending BN+ReLU of x on the fragments of half j (after the reads have landed)
  auto pre_b = [&](int j, bf16x8_v (&bf)[2][2]) {
    if constexpr (PRE) {
      wpp_wait_lgkm0();
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const float scl = tab_lane[j * 64 + jj * 16], sft = tab_lane[256 + j * 64 + jj * 16];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          union { bf16x8_v v; uint32_t w[4]; } q;
          q.v = bf[jj][s];
#pragma unroll
          for (int e = 0; e < 4; ++e) q.w[e] = pre_relu_pair(q.w[e], (f32x2_v){scl, scl}, (f32x2_v){sft, sft});
          bf[jj][s] = q.v;
        }
      }
      valu_to_mfma_pad();
    }
  };
  auto mma = [&](int h, int j, const bf16x8_v (&bf)[2][2]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < QM; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) mfma_bf16_acc(acc[h * QM + i][j * 2 + jj], af[i][s], bf[jj][s]);
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- prologue: tiles kt0 (all four halves) and kt0+1 (A0, B0) -----------------------------------
  WppCursor ce, co;
  set_cursor(ce, kt0);
  set_cursor(co, kt0 + 1);
  issue_a(0, 0, ce); issue_b(0, 0, ce); issue_b(0, 1, ce); issue_a(0, 1, ce);
  set_cursor(ce, kt0 + 2);
  issue_a(1, 0, co); issue_b(1, 0, co);
  wpp_wait_vm<8>();
  wpp_barrier();
  if (grp == 1) wpp_barrier();

#ifdef VINET_CONV_TIMING
  unsigned long long tacc[4] = {0, 0, 0, 0}, tq0, tq1;
#define WPP_T0 tq0 = __builtin_amdgcn_s_memtime();
#define WPP_T(k) { tq1 = __builtin_amdgcn_s_memtime(); tacc[k] += tq1 - tq0; tq0 = tq1; }
#else
#define WPP_T0
#define WPP_T(k)
#endif
  // PREP (pending affine on fresh B fragments) runs in the memory half of the phase, behind the DMA
  // issue, so that the MFMA half stays pure MFMA
#define WPP_PHASE(READS, ISSUE, PREP, MMA)          \
  {                                                  \
    WPP_T0                                           \
    READS;                                           \
    ISSUE;                                           \
    PREP;                                            \
    wpp_wait_vm<8>();                                \
    WPP_T(0)                                         \
    __builtin_amdgcn_sched_barrier(0);               \
    wpp_barrier();                                   \
    WPP_T(1)                                         \
    wpp_wait_lgkm0();                                \
    __builtin_amdgcn_sched_barrier(0);               \
    MMA;                                             \
    __builtin_amdgcn_sched_barrier(0);               \
    WPP_T(2)                                         \
    wpp_barrier();                                   \
    WPP_T(3)                                         \
  }

  const int niter = (kt1 - kt0 + 1) >> 1;
  for (int it = 0; it < niter; ++it) {
    WPP_PHASE((read_a(0, 0), read_b(0, 0, b0)), issue_b(1, 1, co), pre_b(0, b0), mma(0, 0, b0));
    WPP_PHASE(read_b(0, 1, b1), issue_a(1, 1, co), pre_b(1, b1), mma(0, 1, b1));
    set_cursor(co, co.kt + 2);
    WPP_PHASE(read_a(0, 1), issue_a(0, 0, ce), (void)0, mma(1, 1, b1));
    WPP_PHASE((void)0, issue_b(0, 0, ce), (void)0, mma(1, 0, b0));
    WPP_PHASE((read_a(1, 0), read_b(1, 0, b0)), issue_b(0, 1, ce), pre_b(0, b0), mma(0, 0, b0));
    WPP_PHASE(read_b(1, 1, b1), issue_a(0, 1, ce), pre_b(1, b1), mma(0, 1, b1));
    set_cursor(ce, ce.kt + 2);
    WPP_PHASE(read_a(1, 1), issue_a(1, 0, co), (void)0, mma(1, 1, b1));
    WPP_PHASE((void)0, issue_b(1, 0, co), (void)0, mma(1, 0, b0));
  }
#undef WPP_PHASE
#ifdef VINET_CONV_TIMING
  if ((tid & 255) == 0 && a.dbg) {
    float* dbg = a.dbg + ((long)(by * gridDim.x + bx) * 2 + grp) * 4;
    for (int k = 0; k < 4; ++k) dbg[k] = (float)tacc[k] / (float)(niter * 8);
  }
#endif
  if (grp == 0) wpp_barrier();
  wpp_wait_vm<0>();
  mfma_drain();

  // ---- epilogue: acc[h*4+i][j*2+jj] -> dw[slice][n][c] ------------------------------------------------
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int seg = seg0 + g_w * 2 + j;
    if (seg >= a.nseg) continue;
    const int tap = (int)fdiv((uint32_t)seg, a.dCpt);
    const int cbase = (seg - tap * a.cpt) * 64 + p_w * 32;
    const int4 tp = load_tap(a.taps, tap);
    float* slice = a.dw + (long)tp.w * a.N * (long)a.Kp;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int c = cbase + jj * 16 + (lane & 15);
      if (c >= a.Kp) continue;
#pragma unroll
      for (int hi = 0; hi < 2 * QM; ++hi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + wr * WR + hi * 16 + (lane >> 4) * 4 + r;
          if (n < a.N) {
            float* dst = slice + (long)n * a.Kp + c;
#ifdef VINET_WPP_NO_ATOMIC   // tuning build: what do the split-K atomics cost?
            *dst = acc[hi][j * 2 + jj][r];
#else
            atomicAdd(dst, acc[hi][j * 2 + jj][r]);      // (always +=: see wgrad_dma.hip)
#endif
          }
        }
    }
  }
}

template <bool PRE, int QM>
static int launch_wpp(const WgradPPArgs& a, hipStream_t s) {
  auto kern = conv_wgrad_pp_kernel<PRE, QM>;
  static bool attr_done[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_done[dev & 63]) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, WppCfg::SMEM);
    if (e != hipSuccess) { vinet_set_error("hipFuncSetAttribute(wgrad_pp): %s", hipGetErrorString(e)); return (int)e; }
    attr_done[dev & 63] = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.tilesN * a.tilesS, a.splitK), dim3(WppCfg::THREADS), WppCfg::SMEM, s, a);
  return vn_launch_status("conv_wgrad_pp");
}

extern int g_vinet_opt_wgrad_pp;
extern int g_vinet_opt_tperm;
int g_vinet_opt_wgrad_pp_cap = 1;   // the 256 x 256 ping-pong weight gradient honours VinetWgradDesc::max_cus (one round of workgroups under a cap)
#ifdef VINET_CONV_TIMING
static float* g_wpp_dbg = nullptr;
extern "C" void vinet_debug_wgrad_ptr(float* p) { g_wpp_dbg = p; }
#endif

// row tile: 192 or 256 output channels, whichever pads N less
static int wpp_tn(int N) {
  const int p256 = (N + 255) / 256 * 256, p192 = (N + 191) / 192 * 192;
  return p192 < p256 ? 192 : 256;
}

int vinet_wgrad_pp_rows(int N) {
  return g_vinet_opt_wgrad_pp == 3 ? 256 : (g_vinet_opt_wgrad_pp == 4 ? 192 : wpp_tn(N));
}

// The ping-pong kernel wants mostly full tiles (rows: output channels; columns: 4 segments of 64
// input channels), a long voxel range per workgroup and enough workgroups for 256 CUs.
// Measured (tools/conv_ab.py --wgrad): it wins once >= 60% of the tile is real work.
bool vinet_wgrad_use_pp(const VinetWgradDesc* d) {
  if (!g_vinet_opt_wgrad_pp) return false;
  const bool dy_linear = d->dy.sB == (int64_t)d->dy.T * d->dy.H * d->dy.W * d->dy.ld;
  if (g_vinet_opt_wgrad_pp >= 2) return dy_linear;   // tuning: force
  const int N = d->dy.C, Cin = d->x.C;
  const long M = (long)d->dy.B * d->dy.T * d->dy.H * d->dy.W;
  const int nseg = d->ntaps * ((Cin + 63) / 64);
  const int tn = wpp_tn(N);
  const int npad = (N + tn - 1) / tn * tn, spad = (nseg + 3) / 4 * 4;
  const long max_blocks = (long)(npad / tn) * (spad / 4) * (M / 64 / 32);
  // (the fragment-time affine costs the kernel ~25%.  Pointwise layers: the alternative is the 64 x 64 DMA kernel at 180...270
  //  TF/s, and the ping-pong kernel wins from 65 % tile use and 190 workgroups on -- 528 -> 448 at 14 x 24: 1.10 -> 0.56 ms,
  //  832 -> 624 at 7 x 12: 0.25 -> 0.15, 832 -> 448: 0.18 -> 0.12; tools/conv_ab.py --wgrad --only pw, 192 clips)
  const bool pw = d->ntaps == 1;
  const double need = pw ? 0.6 : (d->pre.scale ? 0.7 : 0.6);
  return dy_linear && M >= 32768 && max_blocks >= (pw ? 190 : 256) && (double)N * nseg >= need * (double)npad * spad;
}

int vinet_launch_wgrad_pp(const VinetWgradDesc* d, hipStream_t s) {
  WgradPPArgs a;
  a.x = (const char*)d->x.ptr; a.dy = (const char*)d->dy.ptr; a.dw = d->dw; a.taps = (const int4*)d->taps;
  a.in_scale = d->pre.scale; a.in_shift = d->pre.shift;
  a.Ti = d->x.T; a.Hi = d->x.H; a.Wi = d->x.W; a.Cin = d->x.C; a.ldx = d->x.ld; a.sBx = d->x.sB;
  a.To = d->dy.T; a.Ho = d->dy.H; a.Wo = d->dy.W; a.N = d->dy.C; a.ldy = d->dy.ld; a.sBy = d->dy.sB;
  a.sT = d->sT; a.sH = d->sH; a.sW = d->sW;
  a.ntaps = d->ntaps; a.Kp = d->Kp;
  a.M = (int)((long)d->dy.B * d->dy.T * d->dy.H * d->dy.W);
  a.dW = make_fastdiv(a.Wo); a.dH = make_fastdiv(a.Ho); a.dT = make_fastdiv(a.To);
#ifdef VINET_CONV_TIMING
  a.dbg = g_wpp_dbg;
#endif
  a.perm_P = a.perm_T = 0;
  a.dPT = a.dPermT = make_fastdiv(1);
  if (g_vinet_opt_tperm && a.To > 1 && ((long)a.Ho * a.Wo) % 64 == 0) {
    a.perm_P = (int)(((long)a.Ho * a.Wo) / 64);
    a.perm_T = a.To;
    a.dPT = make_fastdiv((uint32_t)(a.perm_P * a.perm_T));
    a.dPermT = make_fastdiv((uint32_t)a.perm_T);
  }
  a.cpt = (a.Cin + 63) / 64;
  a.dCpt = make_fastdiv(a.cpt);
  a.nseg = a.ntaps * a.cpt;
  const int tn = vinet_wgrad_pp_rows(a.N);
  a.tilesN = vn_div_up(a.N, tn);
  a.tilesS = vn_div_up(a.nseg, 4);
  a.nkt = vn_div_up(a.M, 64);
  // split-K: one workgroup per CU, so the grid should be a whole number of 256-workgroup rounds.
  // Pick the split count that wastes the least of its last round (>= 32 K tiles per split; among
  // near-equal fills the fewest splits: every split adds a 256x256 fp32 atomics pass per tile).
  // Under a CU cap (VinetWgradDesc::max_cus < 256: the caller's other stream wants the rest of the chip) the launch must not hold
  // more workgroups than the cap at ANY time: a grid of several rounds refills every CU the moment a workgroup retires, and a
  // workgroup lives for a millisecond -- a 7-microsecond BatchNorm finalize launch of the main stream took 1.39 ms beside the
  // 832 -> 480 decoder weight gradient (176 tiles x 7 splits = 4.8 rounds of 256; profiles/r4_experiments.txt).  So: one round.
  const long base_blocks = (long)a.tilesN * a.tilesS;
  const int cus = g_vinet_opt_wgrad_pp_cap ? vn_wgrad_cus(d) : 256;
  long sk = 1;
  {
    long max_sk = a.nkt / 32;
    if (max_sk < 1) max_sk = 1;
    if (max_sk > 1024) max_sk = 1024;
    if (cus < 256 && base_blocks <= cus && max_sk > cus / base_blocks) max_sk = cus / base_blocks;
    double best = -1.0;
    for (long k = 1; k <= max_sk; ++k) {
      const long blocks = base_blocks * k;
      const long rounds = (blocks + cus - 1) / cus;
      const double fill = (double)blocks / (double)(rounds * cus);
      if (fill > best + 0.04) { best = fill; sk = k; }
    }
  }
  a.kt_per_split = vn_div_up(a.nkt, sk);
  if (a.kt_per_split & 1) ++a.kt_per_split;                  // whole iterations (two K tiles)
  a.splitK = vn_div_up(a.nkt, a.kt_per_split);
  if (tn == 192) return d->pre.scale ? launch_wpp<true, 3>(a, s) : launch_wpp<false, 3>(a, s);
  return d->pre.scale ? launch_wpp<true, 4>(a, s) : launch_wpp<false, 4>(a, s);
}
