// Weight gradient of kT x 3 x 3 convs with stride (kT,1,1), padding (0,1,1) (input / output channels in chunks of 64;
// image widths 32, 64, 96 one row per step, 48 and 24 two / four rows per step) -- written for the last big
// decoder layer, 192 -> 64, 5x3x3 / (5,1,1) at 20 x 56 x 96 (model.py:273; 3.0 TFLOP per step at 128 clips) -- as a
// row-streaming kernel:
//
//   dw[kt*9 + kh*3 + kw][n][c] += sum_{b,to,h,w} dy[b,to,h,w,n] * x[b, to*kT + kt, h+kh-1, w+kw-1, c]
//
// conv_wgrad_dma_kernel<64,64,9> stages nine shifted copies of every x row (17 B per kFLOP: 347 TF/s, bound by the
// CU's load path).  Temporal taps do not overlap (stride = kT), spatial ones do, so a workgroup fixes (kt, a 64-channel
// chunk of x) and walks the image rows of its (clip, output frame) items with the THREE live x rows in an LDS ring:
// one new x row and one dy row per step, 3.5 B per kFLOP, and the nine taps are nine fragment ADDRESSES into the
// ring (row kh, position w + kw; the ring rows carry a zero position on either side).
//
//   * 512 threads = 8 waves; wave w owns input channels [16(w&3), +16) of the chunk x all 9 taps x output channels
//     [32(w>>2), +32): 18 accumulator tiles (72 AGPRs; the whole 9 x 64 x 64 block is 144 registers per lane at 256
//     threads, more than the accumulator file holds beside two waves per SIMD); per step 9 x 2 x (W/32) MFMAs from
//     9 x (W/32) x-fragments and 2 x (W/32) dy-fragments (ds_read_b64_tr_b16, both operands position-major, 16-byte
//     chunk XOR as wgrad_dma.hip);
//   * loads of the next row are issued before the MFMAs of the current one (named registers: see wgrad_ts.hip);
//   * the grid is (kT x Cin/64) groups x workers; a worker keeps its accumulators over all its items and flushes
//     once with fp32 atomics (dw is zero on entry).
#include "common.h"

// (16 zero bytes every lane of a DMA can point at: rows past the image)
__device__ __attribute__((aligned(64))) uint4 g_vinet_zero_page_rs[4];

struct WgradRsArgs {
  const char* x;
  const char* dy;
  float* dw;
  long sBx, sBy;
  int Ti, To, H, W, ldx, ldy;
  int kT, cchunks, nchunks, Kp, N, Cin;
  int items, workers;          // items = B * To, workers per group
  FastDiv dTo;
  float* dbg;
};

// 128-byte rows: a 32-lane transpose read touches rows {a .. a+3, a+8 .. a+11}; rows of equal parity share a 128-byte half of the
// 256-byte bank row, so bits 1 and 3 of the row pick one of its four 32-byte windows (bit 1 alone left rows r and r + 8 on
// the same banks: a 2-way conflict on every ds_read_b64_tr_b16)
#ifdef VINET_CONV_TIMING
// tuning build: per-workgroup cycle stamps of conv_wgrad_rsm_kernel (tools/wrs_phases.py): [grid][8] floats
// {total, prologues, MFMA phases, barrier 1, write phases, barrier 2, steps, items}
static float* g_wrs_dbg = nullptr;
extern "C" void vinet_debug_wrs_ptr(float* p) { g_wrs_dbg = p; }
#endif
VN_DEV int wrs_swz(int r) { return (((r >> 1) & 1) | (((r >> 3) & 1) << 1)) << 1; }

template <int KS>              // W / 32: K steps per image row (1, 2 or 3)
__global__ __launch_bounds__(512, 1) void conv_wgrad_rs_kernel(const WgradRsArgs a) {
  constexpr int W = KS * 32, RP = W + 2;            // positions per ring row (zero pads at 0 and W + 1)
  constexpr int XROW = RP * 128, DROW = W * 128;
  constexpr int NPC = W * 8;                        // 16-byte pieces of one row (x or dy)
  constexpr int PPT = (NPC + 511) / 512;            // ... per thread
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;                                // 3 x rows
  char* dyb = smem + 3 * XROW;                      // 2 dy rows
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int groups = a.kT * a.cchunks * a.nchunks;
  // XCD-aware: workgroup b runs on XCD b % 8 with its own L2; the groups of ONE worker walk the same (clip, frame) items at the same
  // time -- the n-chunk groups of a (tap, channel chunk) read the same x rows, the (tap, channel chunk) groups of an n chunk the
  // same dy tiles -- so consecutive LOGICAL ids (groups of a worker, n fastest) are put on one XCD (round 6: the r5 counters showed
  // 3.7x the algorithmic bytes past the L2 for the W = 48 sites, each group's copy fetched through a different XCD's L2)
  const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int grp = lid % groups, worker = lid / groups;
  const int n0 = (grp % a.nchunks) * 64;                              // 64 output channels of dy ...
  const int gc = grp / a.nchunks;
  const int kt = gc / a.cchunks, c0 = (gc - kt * a.cchunks) * 64;     // ... one temporal tap, 64 input channels of x

  const int ct = wave & 3, nh = wave >> 2;
  // piece roles: piece q = tid + 512*j -> position q >> 3 (0..W-1), chunk q & 7
  int x_off[PPT], d_off[PPT], g_pos[PPT];
  bool p_ok[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int q = tid + 512 * j;
    p_ok[j] = q < NPC;
    const int pos = p_ok[j] ? q >> 3 : 0, ch = q & 7;
    g_pos[j] = pos;
    x_off[j] = (pos + 1) * 128 + ((ch ^ wrs_swz(pos + 1)) * 16);
    d_off[j] = pos * 128 + ((ch ^ wrs_swz(pos)) * 16);
  }
  const int l_chunk = tid & 7;
  // partial last chunks (Cin, N multiples of 8 only): this lane's 8 channels may lie past the end -> zeros, from a clamped address
  const bool cx_ok = c0 + l_chunk * 8 < a.Cin, dn_ok = n0 + l_chunk * 8 < a.N;
  const int cx_off = cx_ok ? c0 + l_chunk * 8 : 0, dn_off = dn_ok ? n0 + l_chunk * 8 : 0;
  // zero the pad positions of the three ring rows once (never written again)
  if (tid < 3 * 2 * 8) {     // (48 threads)
    const int row = tid / 16, side = (tid >> 3) & 1, ch = tid & 7;
    *(uint4*)(ring + row * XROW + (side ? (W + 1) * 128 : 0) + ch * 16) = make_uint4(0, 0, 0, 0);
  }

  f32x4_v acc[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[t][i] = (f32x4_v){0.f, 0.f, 0.f, 0.f};

  // K-major fragment: 8 positions x 1 channel per lane; `shift` moves the positions along the ring row (tap kw)
  auto frag = [&](const char* row, int ks, int shift, int col0) -> bf16x8_v {
    union { bf16x8_v v; s16x4_v h[2]; } u;
    const int p = lane & 15;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int pos = ks * 32 + (lane >> 4) * 8 + h * 4 + (p >> 2) + shift;
      const int col = col0 + (p & 3) * 4;
      const int ch = (col >> 3) ^ wrs_swz(pos);
      u.h[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_v*)(row + pos * 128 + ch * 16 + (col & 7) * 2));
    }
    return u.v;
  };

  const uint4 z4 = make_uint4(0, 0, 0, 0);
  // (a `cond ? vec : zero` on a 16-byte value is lowered to a two-entry scratch array indexed by the condition, with
  //  a wait on the prefetch load in front of the store: mask the words instead)
  auto keep = [](uint4 v, bool on) -> uint4 {
    const uint32_t m = on ? 0xffffffffu : 0u;
    return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m);
  };
  for (int item = worker; item < a.items; item += a.workers) {
    const int b = (int)fdiv((uint32_t)item, a.dTo);
    const int to = item - b * a.To;
    const int t = to * a.kT + kt;
    const char* xb = a.x + ((long)b * a.sBx + (long)t * a.H * a.W * a.ldx + cx_off) * 2;     // + (h*W + pos) * ldx * 2
    const char* db = a.dy + ((long)b * a.sBy + (long)to * a.H * a.W * a.ldy + dn_off) * 2;
    const long x_rowb = (long)a.W * a.ldx * 2, d_rowb = (long)a.W * a.ldy * 2;

    // ---- prologue: x rows -1 (zeros) and 0 into slots 2 and 0, dy row 0 ---------------------------------------
#pragma unroll
    for (int j = 0; j < PPT; ++j)
      if (p_ok[j]) {
        *(uint4*)(ring + 2 * XROW + x_off[j]) = z4;
        *(uint4*)(ring + 0 * XROW + x_off[j]) = keep(*(const uint4*)(xb + (long)g_pos[j] * a.ldx * 2), cx_ok);
        *(uint4*)(dyb + d_off[j]) = keep(*(const uint4*)(db + (long)g_pos[j] * a.ldy * 2), dn_ok);
      }
    // x row 1 (the "next row" of a virtual step -1); unconditional loads from a clamped row, zero by select
    {
      const char* x1 = xb + (1 < a.H ? x_rowb : 0);
      const uint4 r0 = *(const uint4*)(x1 + (long)g_pos[0] * a.ldx * 2);
      const uint4 r1 = *(const uint4*)(x1 + (long)g_pos[PPT > 1 ? 1 : 0] * a.ldx * 2);
      if (p_ok[0]) *(uint4*)(ring + 1 * XROW + x_off[0]) = keep(r0, 1 < a.H && cx_ok);
      if (PPT > 1 && p_ok[PPT > 1 ? 1 : 0]) *(uint4*)(ring + 1 * XROW + x_off[PPT > 1 ? 1 : 0]) = keep(r1, 1 < a.H && cx_ok);
      if (PPT > 2) {
        const uint4 r2 = *(const uint4*)(x1 + (long)g_pos[PPT > 2 ? 2 : 0] * a.ldx * 2);
        if (p_ok[PPT > 2 ? 2 : 0]) *(uint4*)(ring + 1 * XROW + x_off[PPT > 2 ? 2 : 0]) = keep(r2, 1 < a.H && cx_ok);
      }
    }
    __syncthreads();

    // ring slot of image row r: (r + 3) % 3; step h reads rows h-1, h, h+1 and dy row h (buffer h & 1)
    for (int h = 0; h < a.H; ++h) {
      // ---- loads for step h+1: x row h+2, dy row h+1 -------------------------------------------------------------
      const bool more = h + 1 < a.H;
      const bool xin = more && h + 2 < a.H && cx_ok;
      const char* xs = xb + (xin ? h + 2 : 0) * x_rowb;
      const char* ds = db + (more ? h + 1 : 0) * d_rowb;
      uint4 nx0 = z4, nx1 = z4, nx2 = z4, nd0 = z4, nd1 = z4, nd2 = z4;
      nx0 = *(const uint4*)(xs + (long)g_pos[0] * a.ldx * 2);
      nd0 = *(const uint4*)(ds + (long)g_pos[0] * a.ldy * 2);
      if (PPT > 1) { nx1 = *(const uint4*)(xs + (long)g_pos[PPT > 1 ? 1 : 0] * a.ldx * 2); nd1 = *(const uint4*)(ds + (long)g_pos[PPT > 1 ? 1 : 0] * a.ldy * 2); }
      if (PPT > 2) { nx2 = *(const uint4*)(xs + (long)g_pos[PPT > 2 ? 2 : 0] * a.ldx * 2); nd2 = *(const uint4*)(ds + (long)g_pos[PPT > 2 ? 2 : 0] * a.ldy * 2); }

      // ---- MFMAs --------------------------------------------------------------------------------------------------
      const char* dt = dyb + (h & 1) * DROW;
      const int s_m = (h + 2) % 3, s_0 = h % 3, s_p = (h + 1) % 3;        // slots of rows h-1, h, h+1
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8_v af[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = frag(dt, ks, 0, (nh * 2 + i) * 16);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const char* row = ring + (kh == 0 ? s_m : kh == 1 ? s_0 : s_p) * XROW;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const bf16x8_v bf = frag(row, ks, kw, ct * 16);
#pragma unroll
            for (int i = 0; i < 2; ++i) mfma_bf16_acc(acc[kh * 3 + kw][i], af[i], bf);
          }
        }
      }
      __syncthreads();          // row h-1's slot and the other dy buffer are free
      if (more) {
        char* xslot = ring + ((h + 2) % 3) * XROW;          // row h+2 replaces row h-1
        char* dn = dyb + ((h + 1) & 1) * DROW;
        if (p_ok[0]) { *(uint4*)(xslot + x_off[0]) = keep(nx0, xin); *(uint4*)(dn + d_off[0]) = keep(nd0, dn_ok); }
        if (PPT > 1 && p_ok[PPT > 1 ? 1 : 0]) { *(uint4*)(xslot + x_off[PPT > 1 ? 1 : 0]) = keep(nx1, xin); *(uint4*)(dn + d_off[PPT > 1 ? 1 : 0]) = keep(nd1, dn_ok); }
        if (PPT > 2 && p_ok[PPT > 2 ? 2 : 0]) { *(uint4*)(xslot + x_off[PPT > 2 ? 2 : 0]) = keep(nx2, xin); *(uint4*)(dn + d_off[PPT > 2 ? 2 : 0]) = keep(nd2, dn_ok); }
      }
      __syncthreads();
    }
  }
  mfma_drain();
  // dw[kt*9 + tap][n][c0 + c]: n = (nh*2 + i)*16 + (lane>>4)*4 + r, c = ct*16 + (lane & 15)
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = (nh * 2 + i) * 16 + (lane >> 4) * 4 + r, c = c0 + ct * 16 + (lane & 15);
        if (n0 + n < a.N && c < a.Cin) atomicAdd(a.dw + ((long)(kt * 9 + t) * a.N + n0 + n) * (long)a.Kp + c, acc[t][i][r]);
      }
}

// Several image rows per step for narrow images (W = 48, 24): P = 32*KS positions = R rows of W, so the MFMA K steps
// stay full.  The ring holds the R + 2 live x rows; a fragment position p maps to (row p / W, column p % W), i.e. the
// nine taps are still nine addresses.  Rows past the end of the image are zero in the ring and in the dy tile.
template <int KS, int WW>
__global__ __launch_bounds__(512, 1) void conv_wgrad_rsm_kernel(const WgradRsArgs a) {
  constexpr int P = KS * 32, R = P / WW, RING = R + 2;
  static_assert(R * WW == P && R >= 2, "whole rows per step");
  constexpr int XROW = (WW + 2) * 128, DROW = P * 128;
  constexpr int NPC = P * 8, PPT = (NPC + 511) / 512;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;                                // RING x rows
  char* dyb = smem + RING * XROW;                   // 2 dy tiles
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int groups = a.kT * a.cchunks * a.nchunks;
  // XCD-aware: workgroup b runs on XCD b % 8 with its own L2; the groups of ONE worker walk the same (clip, frame) items at the same
  // time -- the n-chunk groups of a (tap, channel chunk) read the same x rows, the (tap, channel chunk) groups of an n chunk the
  // same dy tiles -- so consecutive LOGICAL ids (groups of a worker, n fastest) are put on one XCD (round 6: the r5 counters showed
  // 3.7x the algorithmic bytes past the L2 for the W = 48 sites, each group's copy fetched through a different XCD's L2)
  const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int grp = lid % groups, worker = lid / groups;
  const int n0 = (grp % a.nchunks) * 64;
  const int gc = grp / a.nchunks;
  const int kt = gc / a.cchunks, c0 = (gc - kt * a.cchunks) * 64;
  const int ct = wave & 3, nh = wave >> 2;
  const int l_chunk = tid & 7;
  const int H = a.H;
  const bool cx_ok = c0 + l_chunk * 8 < a.Cin, dn_ok = n0 + l_chunk * 8 < a.N;
  const int cx_off = cx_ok ? c0 + l_chunk * 8 : 0, dn_off = dn_ok ? n0 + l_chunk * 8 : 0;

  // piece roles: piece q = tid + 512*j -> tile position q >> 3 = (row pr, column pw), chunk q & 7
  int x_in[PPT], d_off[PPT], g_pos[PPT], p_r[PPT];
  bool p_ok[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int q = tid + 512 * j;
    p_ok[j] = q < NPC;
    const int pos = p_ok[j] ? q >> 3 : 0, ch = q & 7;
    const int pr = pos / WW, pw = pos - pr * WW;
    g_pos[j] = pos; p_r[j] = pr;
    x_in[j] = (pw + 1) * 128 + ((ch ^ wrs_swz(pw + 1)) * 16);
    d_off[j] = pos * 128 + ((ch ^ wrs_swz(pos)) * 16);
  }
  for (int i = tid; i < RING * 2 * 8; i += 512) {            // zero pad positions of every ring row, once
    const int row = i / 16, side = (i >> 3) & 1, ch = i & 7;
    *(uint4*)(ring + row * XROW + (side ? (WW + 1) * 128 : 0) + ch * 16) = make_uint4(0, 0, 0, 0);
  }
  // fragment positions of this lane: tile position -> (row, column)
  int f_r[KS][2], f_w[KS][2];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int pos = ks * 32 + (lane >> 4) * 8 + h * 4 + ((lane & 15) >> 2);
      f_r[ks][h] = pos / WW; f_w[ks][h] = pos - (pos / WW) * WW;
    }

  f32x4_v acc[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[t][i] = (f32x4_v){0.f, 0.f, 0.f, 0.f};

  auto frag_dy = [&](const char* tile, int ks, int col0) -> bf16x8_v {
    union { bf16x8_v v; s16x4_v h[2]; } u;
    const int p = lane & 15;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int pos = ks * 32 + (lane >> 4) * 8 + h * 4 + (p >> 2);
      const int col = col0 + (p & 3) * 4;
      const int ch = (col >> 3) ^ wrs_swz(pos);
      u.h[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_v*)(tile + pos * 128 + ch * 16 + (col & 7) * 2));
    }
    return u.v;
  };
  auto keep = [](uint4 v, bool on) -> uint4 {
    const uint32_t m = on ? 0xffffffffu : 0u;
    return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m);
  };
#ifdef VINET_CONV_TIMING
  unsigned long long tm_pro = 0, tm_mma = 0, tm_b1 = 0, tm_wr = 0, tm_b2 = 0, tm_steps = 0, tm_items = 0;
  const unsigned long long tm_start = __builtin_amdgcn_s_memtime();
#define WRS_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define WRS_T(var)
#endif
#ifdef VINET_WRS_PRIO
  // the second-dispatched half of the workgroup loses every arbitration (age) to the first: static priority for it
  if (wave >= 4) __builtin_amdgcn_s_setprio(VINET_WRS_PRIO);
#endif
  const int nsteps = (H + R - 1) / R;

  for (int item = worker; item < a.items; item += a.workers) {
    const int b = (int)fdiv((uint32_t)item, a.dTo);
    const int to = item - b * a.To;
    const int t = to * a.kT + kt;
    WRS_T(t_item0);
    const char* xb = a.x + ((long)b * a.sBx + (long)t * H * WW * a.ldx + cx_off) * 2;      // + (y*W + w) * ldx * 2
    const char* db = a.dy + ((long)b * a.sBy + (long)to * H * WW * a.ldy + dn_off) * 2;

    // ---- prologue: image rows -1 (zero) and 0..R into slots 0..R+1 (slot of row y = (y + 1) % RING), dy rows 0..R-1
    for (int q = tid; q < (R + 1) * WW * 8; q += 512) {
      const int pos = q >> 3, ch = q & 7;                     // (q & 7 == l_chunk)
      const int y = pos / WW, w = pos - y * WW;
      uint4 v = *(const uint4*)(xb + (long)(y < H ? pos : 0) * a.ldx * 2);
      v = keep(v, y < H && cx_ok);
      *(uint4*)(ring + (y + 1) * XROW + (w + 1) * 128 + ((ch ^ wrs_swz(w + 1)) * 16)) = v;
    }
    for (int q = tid; q < WW * 8; q += 512)
      *(uint4*)(ring + ((q >> 3) + 1) * 128 + (((q & 7) ^ wrs_swz((q >> 3) + 1)) * 16)) = make_uint4(0, 0, 0, 0);     // row -1
#pragma unroll
    for (int j = 0; j < PPT; ++j)
      if (p_ok[j]) {
        const bool in = p_r[j] < H && dn_ok;
        const uint4 v = *(const uint4*)(db + (long)(in ? g_pos[j] : 0) * a.ldy * 2);
        *(uint4*)(dyb + d_off[j]) = keep(v, in);
      }
    __syncthreads();

#ifdef VINET_CONV_TIMING
    tm_pro += __builtin_amdgcn_s_memtime() - t_item0; ++tm_items;
#endif
    int s_base = 0;                                          // slot of image row h0 - 1
    for (int st = 0; st < nsteps; ++st) {
      WRS_T(t_s0);
      const int h0 = st * R;
      // ---- loads for the next step: x rows h0+R+1 .. h0+2R, dy rows h0+R .. h0+2R-1 ---------------------------------
      const bool more = st + 1 < nsteps;
      const bool xi0 = more && cx_ok && h0 + R + 1 + p_r[0] < H, xi1 = more && cx_ok && h0 + R + 1 + p_r[PPT > 1 ? 1 : 0] < H;
      const bool di0 = more && dn_ok && h0 + R + p_r[0] < H, di1 = more && dn_ok && h0 + R + p_r[PPT > 1 ? 1 : 0] < H;
      const uint4 nx0 = *(const uint4*)(xb + (long)(xi0 ? (h0 + R + 1) * WW + g_pos[0] : 0) * a.ldx * 2);
      const uint4 nx1 = *(const uint4*)(xb + (long)(xi1 ? (h0 + R + 1) * WW + g_pos[PPT > 1 ? 1 : 0] : 0) * a.ldx * 2);
      const uint4 nd0 = *(const uint4*)(db + (long)(di0 ? (h0 + R) * WW + g_pos[0] : 0) * a.ldy * 2);
      const uint4 nd1 = *(const uint4*)(db + (long)(di1 ? (h0 + R) * WW + g_pos[PPT > 1 ? 1 : 0] : 0) * a.ldy * 2);

      // ---- MFMAs ------------------------------------------------------------------------------------------------------
      const char* dt = dyb + (st & 1) * DROW;
#ifdef VINET_WRS_AHEAD     // experiment: reads of a K step before its MFMAs, reads of the next step before those (two register sets)
      bf16x8_v fa[2][2], fb[2][9];
      auto read_step = [&](int buf, int ks) {
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[buf][i] = frag_dy(dt, ks, (nh * 2 + i) * 16);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            union { bf16x8_v v; s16x4_v h[2]; } u;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              int slot = s_base + f_r[ks][h] + kh;
              slot -= slot >= RING ? RING : 0;
              const int xp = f_w[ks][h] + kw;
              const int col = ct * 16 + (lane & 3) * 4;
              const int ch = (col >> 3) ^ wrs_swz(xp);
              u.h[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                  (__attribute__((address_space(3))) s16x4_v*)(ring + slot * XROW + xp * 128 + ch * 16 + (col & 7) * 2));
            }
            fb[buf][kh * 3 + kw] = u.v;
          }
      };
      read_step(0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) {
          read_step((ks + 1) & 1, ks + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int i = 0; i < 2; ++i) mfma_bf16_acc(acc[t][i], fa[ks & 1][i], fb[ks & 1][t]);
        __builtin_amdgcn_sched_barrier(0);
      }
#else
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8_v af[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = frag_dy(dt, ks, (nh * 2 + i) * 16);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            union { bf16x8_v v; s16x4_v h[2]; } u;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              int slot = s_base + f_r[ks][h] + kh;           // image row h0 + f_r + kh - 1
              slot -= slot >= RING ? RING : 0;
              const int xp = f_w[ks][h] + kw;                // ring position of image column f_w + kw - 1
              const int col = ct * 16 + (lane & 3) * 4;
              const int ch = (col >> 3) ^ wrs_swz(xp);
#ifdef VINET_WRS_NO_LDS      // ablation (tools/wrs_phases.py): MFMAs on whatever the registers hold, no fragment reads
              (void)slot; (void)ch;
              asm volatile("" : "=v"(u.h[h]));
#else
              u.h[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                  (__attribute__((address_space(3))) s16x4_v*)(ring + slot * XROW + xp * 128 + ch * 16 + (col & 7) * 2));
#endif
            }
#ifdef VINET_WRS_NO_MMA      // ablation: fragment reads only
            asm volatile("" :: "v"(u.v), "v"(af[0]), "v"(af[1]));
#else
#pragma unroll
            for (int i = 0; i < 2; ++i) mfma_bf16_acc(acc[kh * 3 + kw][i], af[i], u.v);
#endif
          }
      }
#endif
      WRS_T(t_s1);
      __syncthreads();
      WRS_T(t_s2);
      if (more) {
        char* dn = dyb + ((st + 1) & 1) * DROW;
        if (p_ok[0]) {
          int sl = s_base + p_r[0]; sl -= sl >= RING ? RING : 0;       // new row h0+R+1+pr replaces row h0-1+pr
          *(uint4*)(ring + sl * XROW + x_in[0]) = keep(nx0, xi0);
          *(uint4*)(dn + d_off[0]) = keep(nd0, di0);
        }
        if (PPT > 1 && p_ok[PPT > 1 ? 1 : 0]) {
          int sl = s_base + p_r[PPT > 1 ? 1 : 0]; sl -= sl >= RING ? RING : 0;
          *(uint4*)(ring + sl * XROW + x_in[PPT > 1 ? 1 : 0]) = keep(nx1, xi1);
          *(uint4*)(dn + d_off[PPT > 1 ? 1 : 0]) = keep(nd1, di1);
        }
      }
      s_base += R; s_base -= s_base >= RING ? RING : 0;
      WRS_T(t_s3);
      __syncthreads();
#ifdef VINET_CONV_TIMING
      { const unsigned long long t_s4 = __builtin_amdgcn_s_memtime();
        tm_mma += t_s1 - t_s0; tm_b1 += t_s2 - t_s1; tm_wr += t_s3 - t_s2; tm_b2 += t_s4 - t_s3; ++tm_steps; }
#endif
    }
  }
#ifdef VINET_CONV_TIMING
  if (a.dbg && (tid & 63) == 0) {     // one row per wave
    float* o = a.dbg + ((long)blockIdx.x * 8 + wave) * 8;
    o[0] = (float)(__builtin_amdgcn_s_memtime() - tm_start); o[1] = (float)tm_pro; o[2] = (float)tm_mma; o[3] = (float)tm_b1;
    o[4] = (float)tm_wr; o[5] = (float)tm_b2; o[6] = (float)tm_steps; o[7] = (float)tm_items;
  }
#endif
  mfma_drain();
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = (nh * 2 + i) * 16 + (lane >> 4) * 4 + r, c = c0 + ct * 16 + (lane & 15);
        if (n0 + n < a.N && c < a.Cin) atomicAdd(a.dw + ((long)(kt * 9 + t) * a.N + n0 + n) * (long)a.Kp + c, acc[t][i][r]);
      }
}


// ---- round 4: the same row-streaming weight gradient with FOUR waves per workgroup -------------------------------------------
// tools/wrs_phases.py (s_memtime stamps, ablation builds) on the 8-wave kernels above: a step's 54 MFMAs per wave (864 cycles of
// matrix pipe, 1728 per SIMD with its two waves) take 3600 cycles; without the MFMAs 2400, without the fragment reads 1900 --
// the two waves of a SIMD are bound by INSTRUCTION ISSUE: 66 transpose reads + ~190 address / wait instructions beside 54
// MFMAs, 5.4 per MFMA where ~2 fit into an MFMA's shadow (reads issued ahead of their MFMAs: older waves 2140 cycles, younger
// 3190; static priority only swaps the two).  Here a wave owns 16 input channels x ALL 64 output channels x 9 taps (36
// accumulator tiles, 144 registers): an x fragment feeds four MFMAs instead of two (0.72 transpose reads per MFMA instead of
// 1.22), fragment addresses are two precomputed tables (row-in-ring offset, position offset) and one add per read, and a
// 256-thread workgroup leaves room for a second one on the CU (two independent (tap, channel chunk) groups, or a main-stream
// kernel beside the weight-gradient stream).  P = 32 KS positions per step = R image rows of WW (R = 1: W = 32, 64, 96;
// R = 2, 4: W = 48, 24).
// Round 6: the step's staging is LDS-DMA and the ring is twice as deep.  tools/isa_audit.py on the round-4 form of this kernel: 344
// non-MFMA instructions beside 108 MFMAs per step (3.19 per MFMA) -- 82 v_add (one per transpose read: table + runtime ring slot) and
// ~85 for the register staging of the next rows (loads, masks, multiplies, ds_write) -- on ONE wave per SIMD (310 registers), where
// nothing hides issue time.  Now:
//   * the next step's x rows and dy tile go straight to LDS (global_load_lds, 16 bytes per lane, lane-linear on the LDS side, the
//     chunk swizzle applied on the SOURCE address as in conv_ht.h); the address is a scalar base (item, image row) + a per-lane
//     32-bit offset fixed at kernel start; rows past the image read the zero page.  Lanes of a partial channel chunk read chunk 0
//     instead of zeros: what they feed are columns c >= Cin / rows n >= N of the product, which are never written.
//   * RING = 2R + 2 rows: the R incoming rows have slots of their own, so a step is {issue DMA, multiply, wait, ONE barrier}
//     instead of {load, multiply, barrier, write, barrier}.
//   * the ring walk is unrolled over its period (U = RING / gcd(R, RING) steps), so every ring slot is a compile-time constant
//     and goes into the transpose read's immediate offset: the address register of an x fragment is a table entry, no add.  A K
//     step whose 32 positions straddle two image rows (W = 48: the middle one) keeps a per-lane row bit in its table entries;
//     only where its second row wraps around the ring (one tap row in U * 3) an add per read is left.
constexpr int wrs_gcd(int a, int b) { return b == 0 ? a : wrs_gcd(b, a % b); }
template <int V> struct WrsIC { static constexpr int value = V; };

template <int KS, int WW>
__global__ __launch_bounds__(256, 1) void conv_wgrad_rs4_kernel(const WgradRsArgs a) {
  constexpr int P = KS * 32, R = P / WW, RING = 2 * R + 2, U = RING / wrs_gcd(R, RING);
  static_assert(R * WW == P && R >= 1 && WW % 8 == 0, "whole rows per step, 8-position DMA pieces inside one row");
  constexpr int XROW = (WW + 2) * 128, DROW = P * 128;
  constexpr int PPT = KS;                           // 1 KB DMA pieces (8 positions x 128 B) per wave and step: P / 8 / 4
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;                                // RING x rows
  char* dyb = smem + RING * XROW;                   // 2 dy tiles
  const int tid = threadIdx.x, lane = tid & 63, ct = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int groups = a.kT * a.cchunks * a.nchunks;
  // (XCD-aware: see conv_wgrad_rs_kernel)
  const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int grp = lid % groups, worker = lid / groups;
  const int n0 = (grp % a.nchunks) * 64;
  const int gc = grp / a.nchunks;
  const int kt = gc / a.cchunks, c0 = (gc - kt * a.cchunks) * 64;
  const int l_chunk = tid & 7;
  const int H = a.H;
  const char* zero = (const char*)g_vinet_zero_page_rs;

  // ---- DMA roles: piece j of wave ct = tile positions j * 32 + ct * 8 .. + 7 (one image row: WW % 8 == 0); lane = (position
  // lane >> 3, LDS slot lane & 7), source chunk = slot ^ swizzle(position)
  int xo[PPT], yo[PPT];              // per-lane byte offsets against the scalar row bases
  int pr_[PPT], pw0_[PPT];           // (wave-uniform) image row inside the step and first column of the piece
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int pos0 = j * 32 + ct * 8, pos = pos0 + (lane >> 3);
    const int pr = pos0 / WW, pw = pos - pr * WW;
    pr_[j] = pr; pw0_[j] = pos0 - pr * WW;
    const int xc = l_chunk ^ wrs_swz(pw + 1), dc = l_chunk ^ wrs_swz(pos);
    xo[j] = (pw * a.ldx + c0 + (c0 + xc * 8 < a.Cin ? xc * 8 : 0)) * 2;
    yo[j] = (pos * a.ldy + n0 + (n0 + dc * 8 < a.N ? dc * 8 : 0)) * 2;
  }
  auto dma = [&](const char* base, int off, char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (unsigned)off),
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  for (int i = tid; i < RING * 2 * 8; i += 256) {            // zero pad positions of every ring row, once
    const int row = i / 16, side = (i >> 3) & 1, ch = i & 7;
    *(uint4*)(ring + row * XROW + (side ? (WW + 1) * 128 : 0) + ch * 16) = make_uint4(0, 0, 0, 0);
  }
  // ---- fragment tables of this lane.  K-major fragments: 8 positions x 1 channel per lane, two transpose reads (h) of 4 positions.
  //   x : ring + [slot of (the K step's first row + kh)] * XROW  (compile time)  + xs[ks][h][kw]
  //       xs = (position f_w + kw of the ring row, swizzled chunk) + frl * XROW, frl = 1 for lanes whose positions lie in the K step's
  //       SECOND image row (a straddling K step); where that second row wraps around the ring: - xwrap[ks][h] (= frl * RING * XROW)
  //   dy: tile + da[ks][h] + (((2 i + db) ^ ds[ks][h]) << 4)   (column tile i of the 64 output channels)
  const int p = lane & 15, q = lane >> 4;
  int xs[KS][2][3], xwrap[KS][2], da[KS][2], ds[KS][2];
  const int db = (p & 3) >> 1;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int pos = ks * 32 + q * 8 + h * 4 + (p >> 2);
      const int fr = pos / WW, fw = pos - fr * WW;
      const int frl = fr - (ks * 32) / WW;
      const int col = ct * 16 + (p & 3) * 4;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) xs[ks][h][kw] = frl * XROW + (fw + kw) * 128 + (((col >> 3) ^ wrs_swz(fw + kw)) * 16) + (col & 7) * 2;
      xwrap[ks][h] = frl * RING * XROW;
      da[ks][h] = pos * 128 + ((p & 3) & 1) * 8;
      ds[ks][h] = wrs_swz(pos);
    }

  f32x4_v acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = (f32x4_v){0.f, 0.f, 0.f, 0.f};

  auto keep = [](uint4 v, bool on) -> uint4 {
    const uint32_t m = on ? 0xffffffffu : 0u;
    return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m);
  };
  // Transposing LDS reads as INLINE ASM, their waits counted by hand.  Through the builtin, hipcc knows they are LDS loads and --
  // because an LDS-DMA may alias any LDS load it cannot disambiguate -- puts `s_waitcnt vmcnt(0)` in front of every group of
  // them: each batch would drain the DMAs of the next step's rows that were issued a moment ago (found in the disassembly of
  // this kernel's first DMA form, and in conv_wgrad_pp / conv_wgrad_dma / conv_wgrad_tf, whose counted vmcnt(8) pipelines it defeats).
  // LDS answers in order, so `lgkmcnt(n)` in front of a batch's MFMAs with n = the reads issued for the NEXT batch is exact.
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  auto tr_read = [](unsigned addr, auto offc) -> s16x4_v {
    s16x4_v r;
#ifdef VINET_WRS_NO_LDS      // ablation (tools/wrs_phases.py): MFMAs on whatever the registers hold
    asm volatile("" : "=v"(r) : "v"(addr));
#else
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(decltype(offc)::value));
#endif
    return r;
  };
#ifdef VINET_CONV_TIMING
  unsigned long long tm_pro = 0, tm_mma = 0, tm_b1 = 0, tm_wr = 0, tm_b2 = 0, tm_steps = 0, tm_items = 0;
  const unsigned long long tm_start = __builtin_amdgcn_s_memtime();
#endif
  const int nsteps = (H + R - 1) / R;
  const bool cx_ok = c0 + l_chunk * 8 < a.Cin, dn_ok = n0 + l_chunk * 8 < a.N;
  const int cx_off = cx_ok ? c0 + l_chunk * 8 : 0, dn_off = dn_ok ? n0 + l_chunk * 8 : 0;

  for (int item = worker; item < a.items; item += a.workers) {
    const int b = (int)fdiv((uint32_t)item, a.dTo);
    const int to = item - b * a.To;
    const int t = to * a.kT + kt;
    WRS_T(t_item0);
    const char* xi = a.x + ((long)b * a.sBx + (long)t * H * WW * a.ldx) * 2;           // + ((y*W + w) * ldx + channel) * 2
    const char* di = a.dy + ((long)b * a.sBy + (long)to * H * WW * a.ldy) * 2;
    const char* xb = xi + cx_off * 2;
    const char* db_ = di + dn_off * 2;
    const long xrowb = (long)WW * a.ldx * 2, drowb = (long)WW * a.ldy * 2;
    const char* xnext = xi + (long)(R + 1) * xrowb;        // image row h0 + R + 1 of the step being multiplied (advanced by R rows per step)
    const char* dnext = di + (long)R * drowb;              // dy row h0 + R
    long prx[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) prx[j] = (long)pr_[j] * xrowb;

    // ---- prologue (plain loads): image rows -1 (zero) and 0..R into slots 0..R+1 (slot of row y = (y + 1) % RING), dy rows 0..R-1
    for (int e = tid; e < (R + 1) * WW * 8; e += 256) {
      const int pos = e >> 3, ch = e & 7;                     // (e & 7 == l_chunk)
      const int y = pos / WW, w = pos - y * WW;
      uint4 v = *(const uint4*)(xb + (long)(y < H ? pos : 0) * a.ldx * 2);
      v = keep(v, y < H && cx_ok);
      *(uint4*)(ring + (y + 1) * XROW + (w + 1) * 128 + ((ch ^ wrs_swz(w + 1)) * 16)) = v;
    }
    for (int e = tid; e < WW * 8; e += 256)
      *(uint4*)(ring + ((e >> 3) + 1) * 128 + (((e & 7) ^ wrs_swz((e >> 3) + 1)) * 16)) = make_uint4(0, 0, 0, 0);     // row -1
    for (int e = tid; e < P * 8; e += 256) {
      const int pos = e >> 3, ch = e & 7;
      const bool in = pos / WW < H && dn_ok;
      const uint4 v = *(const uint4*)(db_ + (long)(in ? pos : 0) * a.ldy * 2);
      *(uint4*)(dyb + pos * 128 + ((ch ^ wrs_swz(pos)) * 16)) = keep(v, in);
    }
    __syncthreads();

#ifdef VINET_CONV_TIMING
    tm_pro += __builtin_amdgcn_s_memtime() - t_item0; ++tm_items;
#endif
    // one step; uc = its phase in the ring walk (compile time): image row h0 - 1 sits in slot (uc * R) % RING
    auto step = [&](auto uc, int st) {
      constexpr int SB = (decltype(uc)::value * R) % RING;
      WRS_T(t_s0);
      const int h0 = st * R;
      // ---- DMA for the next step: x rows h0+R+1 .. h0+2R into slots (SB + R + 2 + pr) % RING, dy rows h0+R .. h0+2R-1 into the other
      // tile.  Issued in PPT parts BEHIND the MFMAs of the first batches (one wave per SIMD: what is issued while the matrix pipe
      // works is free, what is issued in front of it is not).  Branch-free: rows past the image select the zero page by masks.
      auto issue_dma = [&](int j) {
        char* dn = dyb + ((st + 1) & 1) * DROW;
        const bool xin = h0 + R + 1 + pr_[j] < H, din = h0 + R + pr_[j] < H;      // (wave-uniform)
        int sl = SB + R + 2 + pr_[j]; sl -= sl >= RING ? RING : 0;
        const char* xsrc = xnext + prx[j];
        xsrc = xin ? xsrc : zero;                                                   // (one s_cselect_b64 each; the offsets by mask)
        const char* dsrc = din ? dnext : zero;
        dma(xsrc, xo[j] & -(int)xin, ring + sl * XROW + (pw0_[j] + 1) * 128);
        dma(dsrc, yo[j] & -(int)din, dn + (j * 32 + ct * 8) * 128);
      };
      // ---- MFMAs: per K step the four dy fragments, then per kernel row kh three x fragments and their twelve MFMAs; the
      // reads of a batch are issued before the MFMAs of the batch in front of it ------------------------------------------
      const unsigned dt = lds0 + RING * XROW + (st & 1) * DROW;
      union Frag { bf16x8_v v; s16x4_v h[2]; };
      Frag fa[4], fb[2][3];
      auto read_dy = [&](int ks) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int h = 0; h < 2; ++h) fa[i].h[h] = tr_read(dt + da[ks][h] + (((2 * i + db) ^ ds[ks][h]) << 4), WrsIC<0>{});
      };
      auto read_x = [&](int buf, auto ksc, auto khc) {
        constexpr int ks = decltype(ksc)::value, kh = decltype(khc)::value;
        constexpr int f0 = (ks * 32) / WW;                             // first image row of the K step (inside the step)
        constexpr bool straddle = (ks * 32 + 31) / WW != f0;
        constexpr int slotA = (SB + f0 + kh) % RING;
        constexpr bool wraps = straddle && slotA + 1 == RING;          // the second row of a straddling K step sits in slot 0
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            fb[buf][kw].h[h] = tr_read(lds0 + (wraps ? xs[ks][h][kw] - xwrap[ks][h] : xs[ks][h][kw]), WrsIC<slotA * XROW>{});
      };
      read_dy(0);
      read_x(0, WrsIC<0>{}, WrsIC<0>{});
      __builtin_amdgcn_sched_barrier(0);
      auto batch = [&](auto bc) {                            // batch = (ks, kh)
        constexpr int b3 = decltype(bc)::value, ks = b3 / 3, kh = b3 % 3;
        if constexpr (b3 + 1 < KS * 3) {
          read_x((b3 + 1) & 1, WrsIC<(b3 + 1) / 3>{}, WrsIC<(b3 + 1) % 3>{});
          asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");      // everything older than the six reads just issued has answered
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
#ifdef VINET_WRS_NO_MMA      // ablation: fragment reads only
            asm volatile("" :: "v"(fa[i].v), "v"(fb[b3 & 1][kw].v));
#else
            mfma_bf16_acc(acc[kh * 3 + kw][i], fa[i].v, fb[b3 & 1][kw].v);
#endif
          }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (b3 < PPT) {                             // (behind this batch's twelve MFMAs)
          issue_dma(b3);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (kh == 2 && ks + 1 < KS) {              // (behind the K step's last MFMAs: they have read fa long before LDS answers)
          read_dy(ks + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      batch(WrsIC<0>{}); batch(WrsIC<1>{}); batch(WrsIC<2>{});
      if constexpr (KS > 1) { batch(WrsIC<3>{}); batch(WrsIC<4>{}); batch(WrsIC<5>{}); }
      if constexpr (KS > 2) { batch(WrsIC<6>{}); batch(WrsIC<7>{}); batch(WrsIC<8>{}); }
      WRS_T(t_s1);
      xnext += (long)R * xrowb; dnext += (long)R * drowb;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // my pieces of the next step's rows have landed ...
      __syncthreads();                                       // ... everyone's have, and everyone has finished reading this step's
      WRS_T(t_s2);
#ifdef VINET_CONV_TIMING
      tm_mma += t_s1 - t_s0; tm_b1 += t_s2 - t_s1; ++tm_steps;
#endif
    };
    for (int st = 0; st < nsteps; st += U) {
      step(WrsIC<0>{}, st);
      if constexpr (U > 1) { if (st + 1 < nsteps) step(WrsIC<1 % U>{}, st + 1); }
      if constexpr (U > 2) { if (st + 2 < nsteps) step(WrsIC<2 % U>{}, st + 2); }
      if constexpr (U > 3) { if (st + 3 < nsteps) step(WrsIC<3 % U>{}, st + 3); }
      if constexpr (U > 4) { if (st + 4 < nsteps) step(WrsIC<4 % U>{}, st + 4); }
      static_assert(U <= 5, "ring period");
    }
  }
#ifdef VINET_CONV_TIMING
  if (a.dbg && (tid & 63) == 0) {     // one row per wave
    float* o = a.dbg + ((long)blockIdx.x * 4 + ct) * 8;
    o[0] = (float)(__builtin_amdgcn_s_memtime() - tm_start); o[1] = (float)tm_pro; o[2] = (float)tm_mma; o[3] = (float)tm_b1;
    o[4] = (float)tm_wr; o[5] = (float)tm_b2; o[6] = (float)tm_steps; o[7] = (float)tm_items;
  }
#endif
  mfma_drain();
  // dw[kt*9 + tap][n0 + n][c0 + c]: n = i*16 + (lane>>4)*4 + r, c = ct*16 + (lane & 15)
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = i * 16 + (lane >> 4) * 4 + r, c = c0 + ct * 16 + (lane & 15);
        if (n0 + n < a.N && c < a.Cin) atomicAdd(a.dw + ((long)(kt * 9 + t) * a.N + n0 + n) * (long)a.Kp + c, acc[t][i][r]);
      }
}

int g_vinet_opt_wgrad_rs4 = 1;  // the four-wave form for W = 24, 48, 32, 64, 96 (0 = the eight-wave kernels above)
int g_vinet_opt_wgrad_rs = 1;   // 0 = off, 2 = force on every eligible shape (tests)

// VinetWgradDesc::tline == 4: the caller promises taps (kt, kh-1, kw-1, slice (kt*3 + kh)*3 + kw), kt < ntaps / 9
bool vinet_wgrad_use_rs(const VinetWgradDesc* d) {
  if (!g_vinet_opt_wgrad_rs || d->tline != 4 || d->dtype != VINET_BF16 || d->mode != VINET_CONV_GENERIC) return false;
  if (d->pre.scale || d->pre.relu || d->bnb_z) return false;
  const int kT = d->ntaps / 9;
  const bool shape = d->ntaps % 9 == 0 && kT >= 1 && d->sT == kT && d->sH == 1 && d->sW == 1 && d->dy.C % 8 == 0 && d->x.C % 8 == 0 && d->Kp >= d->x.C &&
                     kT * ((d->x.C + 63) / 64) * ((d->dy.C + 63) / 64) <= 256 &&
                     d->x.T == kT * d->dy.T && d->x.H == d->dy.H && d->x.W == d->dy.W && ((d->dy.W % 32 == 0 && d->dy.W <= 192) || d->dy.W == 48 || d->dy.W == 24) && d->dy.H >= 2 &&
                     d->x.ld % 8 == 0 && d->dy.ld % 8 == 0 && d->x.sB % 8 == 0 && d->dy.sB % 8 == 0 && ((uintptr_t)d->x.ptr % 16) == 0 &&
                     ((uintptr_t)d->dy.ptr % 16) == 0;
  if (!shape) return false;
  if (g_vinet_opt_wgrad_rs >= 2) return true;
  return (long)d->dy.B * d->dy.T >= 64 && d->dy.H >= 8;
}

int vinet_launch_wgrad_rs(const VinetWgradDesc* d, hipStream_t s) {
  WgradRsArgs a;
  a.x = (const char*)d->x.ptr; a.dy = (const char*)d->dy.ptr; a.dw = d->dw;
  a.sBx = d->x.sB; a.sBy = d->dy.sB;
  a.Ti = d->x.T; a.To = d->dy.T; a.H = d->dy.H; a.W = d->dy.W; a.ldx = d->x.ld; a.ldy = d->dy.ld;
  a.kT = d->ntaps / 9; a.cchunks = (d->x.C + 63) / 64; a.nchunks = (d->dy.C + 63) / 64; a.Kp = d->Kp; a.N = d->dy.C; a.Cin = d->x.C;
  a.items = d->dy.B * a.To;
  a.dTo = make_fastdiv((uint32_t)a.To);
  a.dbg = nullptr;
#ifdef VINET_CONV_TIMING
  a.dbg = g_wrs_dbg;
#endif
  const int groups = a.kT * a.cchunks * a.nchunks;
  int workers = vn_wgrad_cus(d) / groups;       // one 512-thread workgroup per CU, never a second round; the caller's cap (VinetWgradDesc::max_cus) leaves CUs to its other stream
  if (workers < 1) workers = 1;
  if (workers > a.items) workers = a.items;
  a.workers = workers;
  if (g_vinet_opt_wgrad_rs4 && (a.W == 24 || a.W == 48 || a.W == 32 || a.W == 64 || a.W == 96)) {
    const int ks4 = (a.W == 24 || a.W == 48) ? 3 : a.W / 32;
    const int rows = ks4 * 32 / a.W;
    const int smem4 = (2 * rows + 2) * (a.W + 2) * 128 + 2 * ks4 * 32 * 128;      // RING = 2R + 2 ring rows + two dy tiles
    int w4 = 2 * vn_wgrad_cus(d) / groups;       // two 256-thread workgroups per CU
    if (w4 < 1) w4 = 1;
    if (w4 > a.items) w4 = a.items;
    a.workers = w4;
    auto launch4 = [&](auto kern) -> int {
      static bool attr_done[64] = {false};
      int dev = 0;
      (void)hipGetDevice(&dev);
      if (!attr_done[dev & 63]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) { vinet_set_error("hipFuncSetAttribute(wgrad_rs4): %s", hipGetErrorString(e)); return (int)e; }
        attr_done[dev & 63] = true;
      }
      hipLaunchKernelGGL(kern, dim3(groups * w4), dim3(256), smem4, s, a);
      return vn_launch_status("conv_wgrad_rs4");
    };
    if (a.W == 48) return launch4(conv_wgrad_rs4_kernel<3, 48>);
    if (a.W == 24) return launch4(conv_wgrad_rs4_kernel<3, 24>);
    if (a.W == 32) return launch4(conv_wgrad_rs4_kernel<1, 32>);
    if (a.W == 64) return launch4(conv_wgrad_rs4_kernel<2, 64>);
    return launch4(conv_wgrad_rs4_kernel<3, 96>);
  }
  const bool multi = a.W == 48 || a.W == 24;
  const int ks = multi ? 3 : a.W / 32;
  const int ringrows = multi ? 96 / a.W + 2 : 3;
  const int smem = ringrows * (a.W + 2) * 128 + 2 * ks * 32 * 128;
  auto launch = [&](auto kern) -> int {
    if (smem > 64 * 1024) {                     // rows of 128 ... 192 positions: 82 ... 124 KB of ring + dy buffers
      static bool attr_done[64] = {false};      // one per kernel instantiation (generic lambda)
      int dev = 0;
      (void)hipGetDevice(&dev);
      if (!attr_done[dev & 63]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) { vinet_set_error("hipFuncSetAttribute(wgrad_rs): %s", hipGetErrorString(e)); return (int)e; }
        attr_done[dev & 63] = true;
      }
    }
    hipLaunchKernelGGL(kern, dim3(groups * workers), dim3(512), smem, s, a);
    return vn_launch_status("conv_wgrad_rs");
  };
  if (a.W == 48) return launch(conv_wgrad_rsm_kernel<3, 48>);
  if (a.W == 24) return launch(conv_wgrad_rsm_kernel<3, 24>);
  if (ks == 1) return launch(conv_wgrad_rs_kernel<1>);
  if (ks == 2) return launch(conv_wgrad_rs_kernel<2>);
  if (ks == 3) return launch(conv_wgrad_rs_kernel<3>);
  if (ks == 4) return launch(conv_wgrad_rs_kernel<4>);
  if (ks == 5) return launch(conv_wgrad_rs_kernel<5>);
  return launch(conv_wgrad_rs_kernel<6>);
}
