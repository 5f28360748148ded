// (3,1,1) / stride-1 temporal conv with Cin = 64, 128 or 192 input channels as a FRAME-STREAMING kernel: the conv_t of the
// SepConv3d blocks (model_utils.py:148) at base1 (192 -> 192 at 16 x 56 x 96, the largest conv site of the encoder) and in the
// Inception branches, forward and data gradient (a (3,1,1) correlation over dy with the transposed pack).
//
//   y[b, t, p, n] = act(scale[n] * sum_{i < 3} sum_c w[slice_i][n][c] * pre(x[b, t + dt_i, p])[c] + shift[n]),  dt_i in {-1, 0, 1}
//
// The halo-tile kernel's temporal mode (conv_ht.h, TM) stages 6 input frames of 64 positions for 4 output frames, ONE 64-channel
// chunk at a time: three image re-stagings (barrier, 48 KB of DMA, full drain) per tile for nine K steps -- every structure that
// re-stages per chunk or per tap lands at 600...680 TF/s on K = 3 x Cin (profiles/r4_experiments.txt).  Here, as in conv_ts.hip:
//
//   * a workgroup owns 64 positions of one clip x 64 output channels and WALKS the frames; its weights (3 taps x Cin x 64) live in
//     REGISTERS for the whole walk (wave (wm, wn) = positions [32 wm, +32) x channels [32 wn, +32): 3 x Cin/32 x 2 A-fragments, read
//     from the accumulator file), so a step stages nothing but the one new input frame;
//   * input frames go global -> LDS by `global_load_lds_dwordx4` (no registers in flight) into a ring of 6 frames, THREE frames ahead
//     of the one being multiplied; frame = Cin/64 planes of [64 positions][128 B], 16-byte chunk XOR (position & 7) applied on the
//     source address and on the fragment reads (conflict-free ds_read_b128, as conv_ht.h); frames outside the clip are not loaded,
//     their taps are skipped (wave-uniform);
//   * PRE: a pending BatchNorm + ReLU of the input is applied once per loaded element, in LDS, by the wave that loaded it;
//   * per step and wave: 3 x Cin/32 x 2 fragment reads, 3 x Cin/32 x 4 MFMAs, Cin/32 DMA issues, two workgroup barriers; counted
//     `s_waitcnt vmcnt` (loads AND the step's output stores count: one in-order counter on gfx950);
//   * epilogue as conv_ts.hip (operands swapped: 4 consecutive channels per lane, one 8-byte LDS write per tile, 16-byte
//     coalesced stores), BN partial sums in registers over the whole walk: one `stats` row per (clip, position tile).
#include "conv_dma.h"

struct ConvTfArgs {
  const char* x;
  char* y;
  const char* w;
  const int4* taps;
  const float* in_scale;
  const float* in_shift;
  const float* out_scale;
  const float* out_shift;
  float* stats;
  int T, HW, ldx, ldy, Cin, Kp, N, Nw;
  long sBx, sBy;
  int act;
  int items, patches, ntiles;
  FastDiv dPN, dNt;         // items -> (b, patch, ntile): divide by patches * ntiles, by ntiles
};

VN_DEV void mfma_tf(f32x4_v& acc, const bf16x8_v& w, const bf16x8_v& a) {       // weights (accumulator file) as the A operand
  asm volatile("s_nop 3\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(w), "v"(a));
}
VN_DEV void mfma_tf_v(f32x4_v& acc, const bf16x8_v& w, const bf16x8_v& a) {     // ... weights in architectural registers
  asm volatile("s_nop 3\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
}
VN_DEV float tf_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
  return v;
}

// MT = 16-position row tiles per wave: the workgroup owns P = 32 MT positions.  MT = 2 (64 positions): ring of 6 frames, three in
// flight, 158 KB of LDS at Cin = 192 -- ONE workgroup per CU, one wave per SIMD, and everything a wave does (DMA issue, fragment
// reads, MFMAs, epilogue, barriers) serialises: 4800 cycles per step for 1152 cycles of matrix pipe, slower than the halo tiles.
// MT = 1 (32 positions): ring of 5 frames, two in flight, 68 KB -- TWO workgroups per CU, the second wave of a SIMD fills the gaps.
template <int KS, int MT>
struct ConvTfCfg {
  static constexpr int P = 32 * MT;
  static constexpr int RS = MT == 2 ? 6 : 5, D = MT == 2 ? 3 : 2;   // ring slots, frames in flight ahead of the step
  static constexpr int PLANE = P * 128;                 // [P positions][128 B]
  static constexpr int FRAME = KS * PLANE;
  static constexpr int LPF = KS * MT;                   // DMAs per wave and frame (KS P/8 pieces of 8 positions over 4 waves)
  static constexpr int NST = MT;                        // 16-byte output stores per thread and step
  static constexpr int STAGE = P * 128;
  static constexpr int SMEM = RS * FRAME + STAGE + 1024 + 2 * 64 * KS * 4 + 1024;      // + red + PRE constants + a dead KB
  static constexpr int WPC = SMEM <= 80 * 1024 ? 2 : 1; // workgroups per CU
  // vm operations a wave issues per step: NST output stores, then LPF loads (one in-order counter for both on gfx950).  At the top
  // of step t the loads of frame t + 1 must have landed; what may still be in flight is everything issued AFTER them:
  //   the rest of the prologue (frames t + 2 .. D) while t + 1 <= D, and the steps since (all of them / the last D - 1)
  static constexpr int vmwait(int t) {
    return (t + 1 <= D ? (D - (t + 1)) * LPF + t * (NST + LPF) : (D - 1) * (NST + LPF));
  }
  static_assert((D - 1) * (NST + LPF) <= 63 && D <= 3, "vmcnt immediate range; steps 0, 1 and >= 2 are told apart below");
};

// EPI: bit 0 = BN partial sums, bit 1 = output affine; the two-workgroups-per-CU form (128 + 128 registers per wave) is instantiated
// per combination -- 16 registers of sums and 16 of constants it does not need are what make Cin = 192 fit or spill
template <int KS, int MT, bool PRE, int EPI>
__global__ __launch_bounds__(256, (ConvTfCfg<KS, MT>::WPC)) void conv_tf_kernel(const ConvTfArgs a) {
  constexpr bool STATS = (EPI & 1) != 0, AFF = (EPI & 2) != 0;
  using Cfg = ConvTfCfg<KS, MT>;
  constexpr int RS = Cfg::RS, D = Cfg::D, FRAME = Cfg::FRAME, LPF = Cfg::LPF, P = Cfg::P, PLANE = Cfg::PLANE, NST = Cfg::NST;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ring = smem;
  char* const stage = smem + RS * FRAME;
  float* const red = (float*)(stage + Cfg::STAGE);
  float* const aff = red + 256;                         // PRE: scale[Cin], shift[Cin]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* const dead = (char*)(aff + 2 * 64 * KS);        // target of the count-keeping DMAs past the end of a clip (never read)
  const int wm = wave >> 1, wn = wave & 1;
  const int ep = lane & 15, eq = lane >> 4;
  const int l_chunk = tid & 7, l_row = tid >> 3;        // store role: 16-byte chunk of a row, rows l_row + 32 j
  int l_off[NST];
#pragma unroll
  for (int j = 0; j < NST; ++j) {
    const int r = l_row + 32 * j;
    l_off[j] = r * 128 + ((l_chunk ^ (r & 7)) * 16);
  }
  // DMA role of this lane inside an 8-position piece: row (lane >> 3), LDS slot (lane & 7) <- source chunk slot ^ row
  const int prow = lane >> 3;
  const int src_chunk = (lane & 7) ^ prow;
  if constexpr (PRE) {
    for (int c = tid; c < 64 * KS; c += 256) { aff[c] = a.in_scale[c]; aff[64 * KS + c] = a.in_shift[c]; }
  }
  // taps: frame offset and weight slice of each of the three
  int dtp[3], slc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int4 tp = load_tap(a.taps, i);
    dtp[i] = tp.x; slc[i] = tp.w;
  }
  const float relu_floor = a.act == VINET_ACT_RELU ? 0.f : -INFINITY;
  const bool sigm = a.act == VINET_ACT_SIGMOID;
  // wave (wm, wn): positions [16 MT wm, + 16 MT) x channels [32 wn, + 32)
  int st_off[MT][2];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int row = wm * 16 * MT + mt * 16 + ep, col = wn * 32 + nt * 16 + eq * 4;
      st_off[mt][nt] = row * 128 + (((col >> 3) ^ (row & 7)) * 16) + (col & 7) * 2;
    }
  // fragment read offsets inside a plane: position row p, 16-byte chunk (half * 4 + q) ^ (p & 7)
  int fr_off[MT][2];                                    // [mt][half]
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int p = wm * 16 * MT + mt * 16 + ep;
      fr_off[mt][h] = p * 128 + (((h * 4 + eq) ^ (p & 7)) << 4);
    }
  auto dma = [&](const char* src, char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };

  int cur_tile = -1;
  bf16x8_v wr[3][2 * KS][2];                            // [tap][k step][n tile]
  float osc[2][4], osh[2][4];
  // consecutive items = the n tiles of one (clip, patch): neighbours on one XCD share its L2 for x
  for (int item = xcd_remap(blockIdx.x, gridDim.x); item < a.items; item += gridDim.x) {
    const uint32_t bp = fdiv((uint32_t)item, a.dNt);
    const int ntile = item - (int)bp * a.ntiles;
    const int b = (int)fdiv((uint32_t)item, a.dPN);
    const int patch = (int)bp - b * a.patches;
    const int pos0 = patch * P, n0 = ntile * 64;
    if (ntile != cur_tile) {      // (persistent grid, n tile fastest: a workgroup keeps its tile when the grid is a multiple of ntiles)
      cur_tile = ntile;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int ks = 0; ks < 2 * KS; ++ks)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const int n = n0 + wn * 32 + nt * 16 + (lane & 15), c = ks * 32 + eq * 8;
            wr[i][ks][nt] = n < a.Nw ? *(const bf16x8_v*)(a.w + (((long)slc[i] * a.Nw + n) * a.Kp + c) * 2) : (bf16x8_v){0, 0, 0, 0, 0, 0, 0, 0};
          }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + wn * 32 + nt * 16 + eq * 4 + r;
          osc[nt][r] = (AFF && a.out_scale && n < a.Nw) ? a.out_scale[n] : 1.f;
          osh[nt][r] = (AFF && a.out_shift && n < a.Nw) ? a.out_shift[n] : 0.f;
        }
    }
    const char* xb = a.x + ((long)b * a.sBx + (long)(pos0 + prow) * a.ldx + src_chunk * 8) * 2;
    const long x_plane = (long)a.HW * a.ldx * 2, x_r8 = 8L * a.ldx * 2;
    char* yb = a.y + ((long)b * a.sBy + (long)(pos0 + l_row) * a.ldy + n0 + l_chunk * 8) * 2;
    const long y_plane = (long)a.HW * a.ldy * 2, y_r32 = 32L * a.ldy * 2;

    // frame f -> ring slot f % RS: this wave's pieces (plane s, position block pb) = piece index j * 4 + wave of 8 KS
    auto issue_frame = [&](int f) {
      char* dst0 = ring + (f % RS) * FRAME;
      const char* src0 = xb + (long)f * x_plane;
#pragma unroll
      for (int j = 0; j < LPF; ++j) {
        const int piece = j * 4 + wave, s = piece / (P / 8), pb = piece % (P / 8);
        dma(src0 + pb * x_r8 + s * 128, dst0 + s * PLANE + pb * 1024);
      }
    };
    auto issue_nothing = [&]() {      // past the end of the clip: the same number of DMAs into a dead kilobyte, so that the counted waits stay exact
#pragma unroll
      for (int j = 0; j < LPF; ++j) dma(xb, dead);
    };
    // PRE: relu(scale * x + shift) on this wave's pieces of frame f, in place
    auto xform_frame = [&](int f) {
      char* dst0 = ring + (f % RS) * FRAME;
#pragma unroll
      for (int j = 0; j < LPF; ++j) {
        const int piece = j * 4 + wave, s = piece / (P / 8), pb = piece % (P / 8);
        const float* sp = aff + s * 64 + src_chunk * 8;
        const float4 s0 = *(const float4*)sp, s1 = *(const float4*)(sp + 4);
        const float4 h0 = *(const float4*)(sp + 64 * KS), h1 = *(const float4*)(sp + 64 * KS + 4);
        uint4* q = (uint4*)(dst0 + s * PLANE + pb * 1024 + lane * 16);
        const uint4 v = *q;
        *q = make_uint4(pre_relu_pair(v.x, (f32x2_v){s0.x, s0.y}, (f32x2_v){h0.x, h0.y}), pre_relu_pair(v.y, (f32x2_v){s0.z, s0.w}, (f32x2_v){h0.z, h0.w}),
                        pre_relu_pair(v.z, (f32x2_v){s1.x, s1.y}, (f32x2_v){h1.x, h1.y}), pre_relu_pair(v.w, (f32x2_v){s1.z, s1.w}, (f32x2_v){h1.z, h1.w}));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    __syncthreads();                                   // the previous item's ring and stage are free (and the PRE table is written)
    // ---- prologue: frames 0 .. D in flight; frame 0 (and 1) must have landed before step 0 ----------------------------------
#pragma unroll
    for (int f = 0; f <= D; ++f) {
      if (f < a.T) issue_frame(f); else issue_nothing();
    }
    if constexpr (PRE) {
      wait_vmcnt<D * LPF>();                           // frame 0
      xform_frame(0);
    }
    float ssum[2][4], ssq[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { ssum[nt][r] = 0.f; ssq[nt][r] = 0.f; }

    for (int t = 0; t < a.T; ++t) {
      // frame t + 1 (the newest this step reads) has landed: everything older than the D - 1 youngest steps' operations
      if (t == 0) wait_vmcnt<Cfg::vmwait(0)>();
      else if (t == 1) wait_vmcnt<Cfg::vmwait(1)>();
      else wait_vmcnt<Cfg::vmwait(2)>();
      if constexpr (PRE) {
        if (t + 1 < a.T) xform_frame(t + 1);
      }
      __builtin_amdgcn_s_barrier();                    // everyone's pieces of frame t + 1; the stage tile of step t - 1 has been read
      asm volatile("" ::: "memory");

      f32x4_v acc[MT][2];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = (f32x4_v){0.f, 0.f, 0.f, 0.f};
      // fragments of a whole tap are read before its MFMAs, and the NEXT tap's reads are issued before the current tap's MFMAs:
      // with one wave per SIMD nothing else hides the LDS latency (read, wait, 4 MFMAs, read, wait ... ran 5300 cycles per step)
      constexpr int NBUF = Cfg::WPC == 1 ? 2 : 1;      // (two waves per SIMD hide the reads behind each other's MFMAs: one buffer, 24 registers fewer)
      bf16x8_v af[NBUF][2 * KS][MT];
      auto read_tap = [&](int i, int bufi) {
        const int f = t + dtp[i];
        if (f < 0 || f >= a.T) return;                 // (wave-uniform: the frame lies outside the clip)
        const char* fr = ring + (f % RS) * FRAME;
#pragma unroll
        for (int ks = 0; ks < 2 * KS; ++ks)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) af[bufi][ks][mt] = *(const bf16x8_v*)(fr + (ks >> 1) * PLANE + fr_off[mt][ks & 1]);
      };
      if constexpr (NBUF == 2) read_tap(0, 0);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        if constexpr (NBUF == 2) { if (i + 1 < 3) read_tap(i + 1, (i + 1) & 1); }
        else read_tap(i, 0);
        __builtin_amdgcn_sched_barrier(0);
        const int f = t + dtp[i];
        if (f < 0 || f >= a.T) continue;
#pragma unroll
        for (int ks = 0; ks < 2 * KS; ++ks)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
              // (two workgroups per CU = 128 + 128 registers per wave: the 36 weight fragments of Cin = 192 do not fit the
              //  accumulator half -- the third tap's live in the architectural half)
              if (KS == 3 && NBUF == 1 && i == 2) mfma_tf_v(acc[mt][nt], wr[i][ks][nt], af[0][ks][mt]);
              else mfma_tf(acc[mt][nt], wr[i][ks][nt], af[NBUF == 2 ? (i & 1) : 0][ks][mt]);
            }
      }
      mfma_drain();
      // ---- epilogue: lane holds channels 4q .. 4q+3 of position p of each 16 x 16 tile ------------------------------------
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = AFF ? fmaf(acc[mt][nt][r], osc[nt][r], osh[nt][r]) : (float)acc[mt][nt][r];
            if constexpr (STATS) { ssum[nt][r] += v; ssq[nt][r] = fmaf(v, v, ssq[nt][r]); }
            o[r] = fmaxf(v, relu_floor);
            if (sigm) o[r] = 1.f / (1.f + __expf(-o[r]));
          }
          *(uint2*)(stage + st_off[mt][nt]) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                    // the output tile is complete; every wave has finished reading frame t - 1
      asm volatile("" ::: "memory");
      {
        char* yf = yb + (long)t * y_plane;
        uint4 o[NST];
#pragma unroll
        for (int j = 0; j < NST; ++j) o[j] = *(const uint4*)(stage + l_off[j]);
#pragma unroll
        for (int j = 0; j < NST; ++j) *(uint4*)(yf + j * y_r32) = o[j];
      }
      if (t + D + 1 < a.T) issue_frame(t + D + 1); else issue_nothing();
    }
    wait_vmcnt<0>();
    if (STATS && a.stats) {
      __syncthreads();
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ss = tf_row16_sum(ssum[nt][r]), qq = tf_row16_sum(ssq[nt][r]);
          if (ep == 0) {
            const int col = wn * 32 + nt * 16 + eq * 4 + r;
            red[(wm * 64 + col) * 2 + 0] = ss;
            red[(wm * 64 + col) * 2 + 1] = qq;
          }
        }
      __syncthreads();
      if (tid < 64) {
        const long row = (long)b * a.patches + patch;
        a.stats[(row * 2 + 0) * a.N + n0 + tid] = red[tid * 2] + red[(64 + tid) * 2];
        a.stats[(row * 2 + 1) * a.N + n0 + tid] = red[tid * 2 + 1] + red[(64 + tid) * 2 + 1];
      }
    }
  }
}

int g_vinet_opt_conv_tf = 1;   // 0 = off, 1 = heuristic, 2 = force on every eligible shape (tests)

// VinetConvDesc::tline == 1 with three taps, padding 1, unit stride: taps (dt, 0, 0), dt in {-1, 0, 1}
bool vinet_conv_use_tf(const VinetConvDesc* d) {
  if (!g_vinet_opt_conv_tf || d->tline != 1 || d->ntaps != 3 || d->tpad != 1 || d->dtype != VINET_BF16 || d->out_dtype != VINET_BF16 ||
      d->mode != VINET_CONV_GENERIC) return false;
  if (d->pre.scale && !(d->pre.relu && d->pre.shift)) return false;
  if (d->pre.relu && !d->pre.scale) return false;
  if (d->accumulate || (d->act != VINET_ACT_NONE && d->act != VINET_ACT_RELU)) return false;
  const long HW = (long)d->oH * d->oW;
  const int N = d->y.C, Nw = d->n_valid > 0 ? d->n_valid : N;
  const bool shape = (d->x.C == 64 || d->x.C == 128 || d->x.C == 192) && d->Kp == d->x.C && N % 64 == 0 && Nw == N && d->sT == 1 && d->sH == 1 &&
                     d->sW == 1 && d->omT == 1 && d->omH == 1 && d->omW == 1 && d->ooT == 0 && d->ooH == 0 && d->ooW == 0 && d->oT == d->x.T &&
                     d->y.T == d->oT && d->x.H == d->oH && d->x.W == d->oW && d->y.H == d->oH && d->y.W == d->oW && HW % 64 == 0 &&
                     d->x.ld % 8 == 0 && d->y.ld % 8 == 0 && d->x.sB % 8 == 0 && d->y.sB % 8 == 0 && ((uintptr_t)d->x.ptr % 16) == 0 &&
                     ((uintptr_t)d->y.ptr % 16) == 0 && ((uintptr_t)d->w % 16) == 0;
  if (!shape) return false;
  if (g_vinet_opt_conv_tf >= 2) return true;
  return (long)d->x.B * (HW / 64) * ((N + 63) / 64) >= 1024 && d->oT >= 4;      // a persistent grid's worth of items, a walk worth its prologue
}

template <int KS, int MT, bool PRE, int EPI>
static int launch_conv_tf_k(const ConvTfArgs& a, hipStream_t s) {
  using Cfg = ConvTfCfg<KS, MT>;
  auto k = conv_tf_kernel<KS, MT, PRE, EPI>;
  static bool attr_done[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_done[dev & 63]) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM);
    if (e != hipSuccess) { vinet_set_error("hipFuncSetAttribute(conv_tf): %s", hipGetErrorString(e)); return (int)e; }
    attr_done[dev & 63] = true;
  }
  // persistent grid: as many workgroups as the chip holds, a multiple of the n tiles so that a workgroup keeps its weights
  int grid = 256 * Cfg::WPC;
  grid -= grid % a.ntiles;
  if (grid > a.items) grid = a.items;
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), Cfg::SMEM, s, a);
  return vn_launch_status("conv_tf");
}
template <int KS, int MT>
static int launch_conv_tf(const ConvTfArgs& a, bool pre, hipStream_t s) {
  const int epi = (a.stats ? 1 : 0) | ((a.out_scale || a.out_shift) ? 2 : 0);
  if constexpr (MT == 2) {      // (one workgroup per CU, 512 registers per wave: one generic epilogue)
    return pre ? launch_conv_tf_k<KS, 2, true, 3>(a, s) : launch_conv_tf_k<KS, 2, false, 3>(a, s);
  } else {
    switch (epi | (pre ? 4 : 0)) {
      case 0: return launch_conv_tf_k<KS, 1, false, 0>(a, s);
      case 1: return launch_conv_tf_k<KS, 1, false, 1>(a, s);
      case 2: return launch_conv_tf_k<KS, 1, false, 2>(a, s);
      case 3: return launch_conv_tf_k<KS, 1, false, 3>(a, s);
      case 4: return launch_conv_tf_k<KS, 1, true, 0>(a, s);
      case 5: return launch_conv_tf_k<KS, 1, true, 1>(a, s);
      case 6: return launch_conv_tf_k<KS, 1, true, 2>(a, s);
      default: return launch_conv_tf_k<KS, 1, true, 3>(a, s);
    }
  }
}

int g_vinet_opt_conv_tf_p = 32;   // positions per workgroup: 32 (two workgroups per CU) or 64

int vinet_conv_tf_positions() { return g_vinet_opt_conv_tf_p == 64 ? 64 : 32; }     // per workgroup = per statistics row

int vinet_launch_conv_tf(const VinetConvDesc* d, hipStream_t s) {
  ConvTfArgs a;
  a.x = (const char*)d->x.ptr; a.y = (char*)d->y.ptr; a.w = (const char*)d->w; a.taps = (const int4*)d->taps;
  a.in_scale = d->pre.scale; a.in_shift = d->pre.shift;
  a.out_scale = d->out_scale; a.out_shift = d->out_shift; a.stats = d->stats;
  a.T = d->x.T; a.HW = d->oH * d->oW; a.ldx = d->x.ld; a.ldy = d->y.ld; a.Cin = d->x.C; a.Kp = d->Kp; a.N = d->y.C;
  a.Nw = d->n_valid > 0 ? d->n_valid : d->y.C;
  a.sBx = d->x.sB; a.sBy = d->y.sB; a.act = d->act;
  const int P = g_vinet_opt_conv_tf_p == 64 ? 64 : 32;
  a.patches = a.HW / P;
  a.ntiles = (a.N + 63) / 64;
  a.items = d->x.B * a.patches * a.ntiles;
  a.dNt = make_fastdiv((uint32_t)a.ntiles);
  a.dPN = make_fastdiv((uint32_t)(a.patches * a.ntiles));
  const bool pre = d->pre.scale != nullptr;
  if (P == 64) {
    if (a.Cin == 64) return launch_conv_tf<1, 2>(a, pre, s);
    if (a.Cin == 128) return launch_conv_tf<2, 2>(a, pre, s);
    return launch_conv_tf<3, 2>(a, pre, s);
  }
  if (a.Cin == 64) return launch_conv_tf<1, 1>(a, pre, s);
  if (a.Cin == 128) return launch_conv_tf<2, 1>(a, pre, s);
  return launch_conv_tf<3, 1>(a, pre, s);
}
