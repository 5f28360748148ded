// Temporal conv (k x 1 x 1, stride (s,1,1)) with 64 input and 64 output channels as a frame-streaming kernel:
// forward of the stem's partner, 64 -> 64 7x1x1 / 2 at 32 x 112 x 192 (model_utils.py:149), and each stride
// phase of its data gradient (a stride-1 temporal correlation over dy with 3 or 4 taps).
//
//   y[b, to*om + oo, p, n] = act(scale[n] * sum_i sum_c w[slice_i][n][c] * pre(x[b, to*s + off_i, p])[c] + shift[n])
//
// conv_dma_kernel re-stages the input frame of every tap: k/s = 3.5 reads of x (9 GB of L2-miss traffic per launch
// for 2.6 GB of tensors at 128 clips, 1.9 TB/s).  Here -- as in wgrad_ts.hip -- a workgroup owns 64 (h,w) positions
// of one clip and walks the output frames with the k live input frames in an LDS ring: each input element is
// fetched once (global -> registers -> pending BN+ReLU once -> LDS), each output element written once.
//
//   * 256 threads = 2 x 2 waves: wave (wm, wn) owns positions [32wm, 32wm+32) x output channels [32wn, 32wn+32);
//     its slice of ALL k taps' weights lives in registers (k x 2 x 2 B-fragments = 112 VGPRs at k = 7), so the K
//     loop reads only activations from LDS (ds_read_b128, rows of 128 B, chunk ^ (row & 7) swizzle);
//   * epilogue per output frame: scale / shift / activation, BN partial sums (one row of `stats` per 64 positions:
//     vinet_conv3d_tile_m reports 64), bf16 tile through LDS, 16-byte coalesced stores (optional read-modify-write);
//   * the tap table is read on the device; the host only needs the caller's promise (VinetConvDesc::tline) that the
//     taps are temporal and their offsets form the contiguous range [-tpad, -tpad + ntaps - 1].
#include "common.h"

struct ConvTsArgs {
  const char* x;
  char* y;
  const char* w;
  const int4* taps;
  const float* in_scale;
  const float* in_shift;
  const float* out_scale;
  const float* out_shift;
  float* stats;
  int Ti, To, HW, ldx, ldy;
  long sBx, sBy;
  int k, s, pad, omT, ooT, act, accumulate;
  int items, patches;
  FastDiv dPatches;
  // frame segments (conv_ts_kernel, small batches): an item is output frames [seg * seg_frames, ...) of a
  // 64-position patch, so that a batch-1 clip (336 patches) fills the chip; 1 = all frames
  int segs, seg_frames;
  FastDiv dSegs;
};

// MFMAs of this kernel: B (the weights) is read from the accumulator file -- the 112 weight registers do not fit in
// the 128 architectural VGPRs the compiler budgets beside everything else, and gfx90a+ MFMAs read A / B from
// either file.  Wherever the register allocator keeps a value, it may copy it (v_accvgpr_write / _read) right in
// front of the inline-asm MFMA, a VALU-write -> MFMA-read hazard it cannot see through the asm: every MFMA
// carries its own wait states.  (4 cycles x 56 MFMAs per output frame, against ~3000 cycles of HBM time.)
// Operands SWAPPED (round 4): the weights are the A operand, so the accumulators arrive transposed -- lane (p = lane & 15,
// q = lane >> 4) holds 4 consecutive CHANNELS 4q .. 4q+3 of position p per 16 x 16 tile: the epilogue packs them with two
// conversions and ONE 8-byte LDS write per tile instead of four 2-byte ones with their address arithmetic (the kernel was
// VALU-issue bound on that: ~12 instructions per output element, ~5 now), and the BN partial sums stay in registers over all
// output frames of the item (one `stats` row per item).
VN_DEV void mfma_bf16_acc_bacc(f32x4_v& acc, const bf16x8_v& a, const bf16x8_v& b) {
  asm volatile("s_nop 3\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(b), "v"(a));
}
// first MFMA of an output frame: C = 0 as an inline constant (early clobber: the result must not share registers
// with the weights)
VN_DEV void mfma_bf16_first_bacc(f32x4_v& acc, const bf16x8_v& a, const bf16x8_v& b) {
  asm volatile("s_nop 3\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&a"(acc) : "a"(b), "v"(a));
}
VN_DEV float ts_row16_sum(float v) {      // sum over the 16 lanes of a row (DPP row_ror 8, 4, 2, 1): every lane gets the total
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
  return v;
}

// AFF: an output affine (folded eval-mode BatchNorm, bias) is applied; the training form (raw output + statistics) does without
// its 16 constant registers
template <bool PRE, bool AFF>
__global__ __launch_bounds__(256, 2) void conv_ts_kernel(const ConvTsArgs a) {
  constexpr int KMAX = 7, TILE = 64 * 64 * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;                                   // KMAX frames [64 positions][64 channels]
  char* stage = smem + KMAX * TILE;                    // output tile [64 positions][64 channels] bf16
  float* red = (float*)(smem + (KMAX + 1) * TILE);     // [2 position halves][64 channels][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int k = a.k, s = a.s;

  // load / store role: 16-byte piece (row = position, chunk = 8 channels), rows l_row and l_row + 32
  const int l_chunk = tid & 7, l_row = tid >> 3;
  int l_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = l_row + 32 * j;
    l_off[j] = r * 128 + ((l_chunk ^ (r & 7)) * 16);
  }
  f32x2_v sc2[4], sh2[4];
  if constexpr (PRE) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sc2[e] = (f32x2_v){a.in_scale[l_chunk * 8 + 2 * e], a.in_scale[l_chunk * 8 + 2 * e + 1]};
      sh2[e] = (f32x2_v){a.in_shift[l_chunk * 8 + 2 * e], a.in_shift[l_chunk * 8 + 2 * e + 1]};
    }
  }
  auto xform = [&](uint4 v) -> uint4 {
    if constexpr (PRE) {
      v.x = pre_relu_pair(v.x, sc2[0], sh2[0]); v.y = pre_relu_pair(v.y, sc2[1], sh2[1]);
      v.z = pre_relu_pair(v.z, sc2[2], sh2[2]); v.w = pre_relu_pair(v.w, sc2[3], sh2[3]);
    }
    return v;
  };

  // ---- taps (scalar) and this wave's weights (registers) ------------------------------------------------
  int rel[KMAX];                                       // tap i reads frame to*s - pad + rel[i]
  bf16x8_v wr[KMAX][2][2];                             // [tap][n tile][k step]: B fragment = 8 channels of one n
#pragma unroll
  for (int i = 0; i < KMAX; ++i) {
    rel[i] = 0;
    if (i < k) {
      const int4 tp = load_tap(a.taps, i);
      rel[i] = tp.x + a.pad;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int n = wn * 32 + nt * 16 + (lane & 15), c = ks * 32 + (lane >> 4) * 8;
          wr[i][nt][ks] = *(const bf16x8_v*)(a.w + (((long)tp.w * 64 + n) * 64 + c) * 2);
        }
    } else {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wr[i][nt][ks] = (bf16x8_v){0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  // epilogue role: position p of a 16-row tile, channel quad q of a 16-column tile; constants of my 2 x 4 output channels
  const int ep = lane & 15, eq = lane >> 4;
  float osc[2][4], osh[2][4];
  int st_off[2][2];                                    // byte offset of my 8 bytes (4 channels) of tile (mt, nt) in the output stage
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = wn * 32 + nt * 16 + eq * 4 + r;
      osc[nt][r] = (AFF && a.out_scale) ? a.out_scale[n] : 1.f;
      osh[nt][r] = (AFF && a.out_shift) ? a.out_shift[n] : 0.f;
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = wm * 32 + mt * 16 + ep, col = wn * 32 + nt * 16 + eq * 4;
      st_off[mt][nt] = row * 128 + (((col >> 3) ^ (row & 7)) * 16) + (col & 7) * 2;
    }
  }
  const float relu_floor = a.act == VINET_ACT_RELU ? 0.f : -INFINITY;
  const bool sigm = a.act == VINET_ACT_SIGMOID;

  for (int item = blockIdx.x; item < a.items; item += gridDim.x) {
    const int pitem = a.segs > 1 ? (int)fdiv((uint32_t)item, a.dSegs) : item;      // (patch item, frame segment)
    const int t0 = (item - pitem * a.segs) * a.seg_frames;
    const int t1 = t0 + a.seg_frames < a.To ? t0 + a.seg_frames : a.To;
    const int b = (int)fdiv((uint32_t)pitem, a.dPatches);
    const int patch = pitem - b * a.patches;
    const int pos0 = patch * 64;
    const char* xb = a.x + ((long)b * a.sBx + (long)(pos0 + l_row) * a.ldx + l_chunk * 8) * 2;
    char* yb = a.y + ((long)b * a.sBy + (long)(pos0 + l_row) * a.ldy + l_chunk * 8) * 2;
    const long x_plane = (long)a.HW * a.ldx * 2, y_plane = (long)a.HW * a.ldy * 2;
    const long x_r32 = 32L * a.ldx * 2, y_r32 = 32L * a.ldy * 2;

    // ---- prologue: the k frames of the first output frame --------------------------------------------------
    for (int g = 0; g < k; ++g) {
      const int p = t0 * s + g - a.pad;
      uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0;
      if ((unsigned)p < (unsigned)a.Ti) {
        v0 = xform(*(const uint4*)(xb + p * x_plane));
        v1 = xform(*(const uint4*)(xb + p * x_plane + x_r32));
      }
      char* slot = ring + ((p + 2 * KMAX) % k) * TILE;
      *(uint4*)(slot + l_off[0]) = v0;
      *(uint4*)(slot + l_off[1]) = v1;
    }
    __syncthreads();

    // BN partial sums of my 2 x 4 channels over every output frame of the item (one `stats` row per ITEM)
    float ssum[2][4], ssq[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { ssum[nt][r] = 0.f; ssq[nt][r] = 0.f; }

    for (int to = t0; to < t1; ++to) {
      // ---- loads of the next step's s new frames (named scalars, unconditional: see wgrad_ts.hip) ----------
      const bool more = to + 1 < t1;
      const int pnew = (to + 1) * s - a.pad + k - s;
      const bool in0 = more && (unsigned)pnew < (unsigned)a.Ti;
      const bool in1 = more && s == 2 && (unsigned)(pnew + 1) < (unsigned)a.Ti;
      const char* xs0 = xb + (in0 ? pnew : 0) * x_plane;
      const char* xs1 = xb + (in1 ? pnew + 1 : 0) * x_plane;
      const uint4 nx00 = *(const uint4*)xs0, nx01 = *(const uint4*)(xs0 + x_r32);
      const uint4 nx10 = *(const uint4*)xs1, nx11 = *(const uint4*)(xs1 + x_r32);
      char* yf = yb + (long)(to * a.omT + a.ooT) * y_plane;
      uint4 old0 = make_uint4(0, 0, 0, 0), old1 = old0;
      if (a.accumulate) { old0 = *(const uint4*)yf; old1 = *(const uint4*)(yf + y_r32); }

      // ---- MFMAs ---------------------------------------------------------------------------------------------
      f32x4_v acc[2][2];
      const int s0 = (to * s - a.pad + 2 * KMAX) % k;
#pragma unroll
      for (int i = 0; i < KMAX; ++i) {
        if (i < k) {
          int si = s0 + rel[i];
          si -= si >= k ? k : 0;
          const char* fr = ring + si * TILE;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            bf16x8_v af[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
              const int row = wm * 32 + mt * 16 + (lane & 15), ch = ks * 4 + (lane >> 4);
              af[mt] = *(const bf16x8_v*)(fr + row * 128 + ((ch ^ (row & 7)) * 16));
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
              for (int nt = 0; nt < 2; ++nt) {
                if (i == 0 && ks == 0) mfma_bf16_first_bacc(acc[mt][nt], af[mt], wr[i][nt][ks]);
                else mfma_bf16_acc_bacc(acc[mt][nt], af[mt], wr[i][nt][ks]);
              }
          }
        }
      }
      mfma_drain();
      // ---- epilogue: lane holds channels 4q .. 4q+3 of position p of each 16 x 16 tile ------------------------------
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = AFF ? fmaf(acc[mt][nt][r], osc[nt][r], osh[nt][r]) : (float)acc[mt][nt][r];
            ssum[nt][r] += v; ssq[nt][r] = fmaf(v, v, ssq[nt][r]);
            o[r] = fmaxf(v, relu_floor);
            if (sigm) o[r] = 1.f / (1.f + __expf(-o[r]));
          }
          *(uint2*)(stage + st_off[mt][nt]) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
        }
      __syncthreads();          // ring frames of this step are free, the output tile is complete
      {
        uint4 o0 = *(const uint4*)(stage + l_off[0]), o1 = *(const uint4*)(stage + l_off[1]);
        if (a.accumulate) {
          auto add2 = [](uint32_t p, uint32_t q) -> uint32_t {
            return pack2bf(__uint_as_float(p << 16) + __uint_as_float(q << 16),
                           __uint_as_float(p & 0xffff0000u) + __uint_as_float(q & 0xffff0000u));
          };
          o0 = make_uint4(add2(o0.x, old0.x), add2(o0.y, old0.y), add2(o0.z, old0.z), add2(o0.w, old0.w));
          o1 = make_uint4(add2(o1.x, old1.x), add2(o1.y, old1.y), add2(o1.z, old1.z), add2(o1.w, old1.w));
        }
        *(uint4*)yf = o0;
        *(uint4*)(yf + y_r32) = o1;
      }
      if (more) {
        const uint4 z = make_uint4(0, 0, 0, 0);
        char* slot0 = ring + ((pnew + 2 * KMAX) % k) * TILE;
        *(uint4*)(slot0 + l_off[0]) = in0 ? xform(nx00) : z;
        *(uint4*)(slot0 + l_off[1]) = in0 ? xform(nx01) : z;
        if (s == 2) {
          char* slot1 = ring + ((pnew + 1 + 2 * KMAX) % k) * TILE;
          *(uint4*)(slot1 + l_off[0]) = in1 ? xform(nx10) : z;
          *(uint4*)(slot1 + l_off[1]) = in1 ? xform(nx11) : z;
        }
      }
      __syncthreads();
    }
    if (a.stats) {     // once per item: 16 positions of a row by DPP, the two position halves (wm) in LDS
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ss = ts_row16_sum(ssum[nt][r]), qq = ts_row16_sum(ssq[nt][r]);
          if (ep == 0) {
            const int col = wn * 32 + nt * 16 + eq * 4 + r;
            red[(wm * 64 + col) * 2 + 0] = ss;
            red[(wm * 64 + col) * 2 + 1] = qq;
          }
        }
      __syncthreads();
      if (tid < 64) {
        a.stats[((long)item * 2 + 0) * 64 + tid] = red[tid * 2] + red[(64 + tid) * 2];
        a.stats[((long)item * 2 + 1) * 64 + tid] = red[tid * 2 + 1] + red[(64 + tid) * 2 + 1];
      }
      __syncthreads();
    }
  }
}


// ---- the same frame-streaming kernel in the split-bf16 form (VINET_F32S: fp32 x and y, hi / lo weight planes) --------------------
// The split happens ONCE per loaded element, on the way into the LDS ring (where the pending BatchNorm + ReLU is applied anyway):
// a frame is kept as a hi and a lo bf16 plane, [32 positions][64 channels] each, with the channels of every 32-wide chunk in the K
// order of vinet_pack_weights(VINET_F32S) (lane group q: elements 4q .. 4q+3 and 16+4q .. 16+4q+3 -- the 4 consecutive channels a
// loader thread holds land as 8 contiguous bytes), so a fragment is one ds_read_b128 per plane and the MFMAs read hi / lo operands
// directly: weights_lo x hi + weights_hi x lo + weights_hi x hi.  32 positions per workgroup (ring 7 x 8 KB + an fp32 stage tile of
// 8 KB: two workgroups per CU); the four waves split the 64 output channels (16 each, all 32 positions), so a wave's weights are
// 7 x 2 hi + 7 x 2 lo fragments = 112 registers, as in the bf16 kernel.  conv_dma3 staged every tap's frame again (10.7 ms forward,
// 5.9 + 4.8 ms for the two stride phases of the data gradient at 64 clips, 1.0-1.6 TB/s).
VN_DEV void ts3_split(float x0, float x1, uint32_t& hi, uint32_t& lo) {
  hi = pack2bf(x0, x1);
  lo = pack2bf(x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u));
}
// (the compiler's own MFMA: it pads the hazards of three dependent products on one accumulator itself -- an inline-asm form with
//  fixed wait states returned stale accumulator elements for the second row tile)
VN_DEV void mfma_ts3(f32x4_v& acc, const bf16x8_v& w, const bf16x8_v& a) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, a, acc, 0, 0, 0);
}

template <bool PRE>
__global__ __launch_bounds__(256, 2) void conv_ts3_kernel(const ConvTsArgs a) {
  constexpr int KMAX = 7, P = 32, PLANE = P * 128, FRAME = 2 * PLANE, STAGE = P * 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;                                   // KMAX frames: [hi plane | lo plane], [32 positions][128 B]
  char* stage = smem + KMAX * FRAME;                   // output tile [32 positions][64 channels] fp32, 16-byte chunk ^ (row & 15)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = a.k, s = a.s;
  const int ep = lane & 15, eq = lane >> 4;
  // load / store role: 16 bytes = 4 fp32 channels 4 c4 .. 4 c4 + 3 of row l_row (+ 16)
  const int c4 = tid & 15, l_row = tid >> 4;
  // where my 4 channels go inside a plane row: 16-byte column (chunk * 4 + group), 8-byte half, XOR (row & 7) on the column
  const int w_col16 = (c4 >> 3) * 4 + (c4 & 3), w_half = (c4 >> 2) & 1;
  int w_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = l_row + 16 * j;
    w_off[j] = r * 128 + ((w_col16 ^ (r & 7)) * 16) + w_half * 8;
  }
  float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psh = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (PRE) { psc = *(const float4*)(a.in_scale + c4 * 4); psh = *(const float4*)(a.in_shift + c4 * 4); }
  // fp32 -> (pending BatchNorm + ReLU) -> hi / lo halves of my 4 channels
  auto put = [&](char* slot, int j, uint4 v, bool live) {
    float f[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    if constexpr (PRE) {
      f[0] = fmaxf(fmaf(f[0], psc.x, psh.x), 0.f); f[1] = fmaxf(fmaf(f[1], psc.y, psh.y), 0.f);
      f[2] = fmaxf(fmaf(f[2], psc.z, psh.z), 0.f); f[3] = fmaxf(fmaf(f[3], psc.w, psh.w), 0.f);
    }
    uint32_t h0, l0, h1, l1;
    ts3_split(f[0], f[1], h0, l0);
    ts3_split(f[2], f[3], h1, l1);
    *(uint2*)(slot + w_off[j]) = live ? make_uint2(h0, h1) : make_uint2(0, 0);
    *(uint2*)(slot + PLANE + w_off[j]) = live ? make_uint2(l0, l1) : make_uint2(0, 0);
  };

  // ---- taps (scalar) and this wave's weights: channels 16 wave .. + 15, hi and lo planes of every tap ------------------
  int rel[KMAX];
  bf16x8_v wh[KMAX][2], wl[KMAX][2];                   // [tap][k step]
#pragma unroll
  for (int i = 0; i < KMAX; ++i) {
    rel[i] = 0;
    if (i < k) {
      const int4 tp = load_tap(a.taps, i);
      rel[i] = tp.x + a.pad;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int n = wave * 16 + (lane & 15);
        const char* wr = a.w + ((((long)tp.w * 64 + n) * 2 + ks) * 64 + eq * 8) * 2;     // row = 2 chunks of [32 hi | 32 lo]
        wh[i][ks] = *(const bf16x8_v*)wr;
        wl[i][ks] = *(const bf16x8_v*)(wr + 64);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) { wh[i][ks] = (bf16x8_v){0, 0, 0, 0, 0, 0, 0, 0}; wl[i][ks] = wh[i][ks]; }
    }
  }
  float osc[4], osh[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = wave * 16 + eq * 4 + r;
    osc[r] = a.out_scale ? a.out_scale[n] : 1.f;
    osh[r] = a.out_shift ? a.out_shift[n] : 0.f;
  }
  const bool aff_out = a.out_scale != nullptr || a.out_shift != nullptr;
  int st_off[2], fr_off[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int row = mt * 16 + ep;
    st_off[mt] = row * 256 + (((wave * 4 + eq) ^ (row & 15)) * 16);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) fr_off[mt][ks] = row * 128 + (((ks * 4 + eq) ^ (row & 7)) * 16);
  }
  const float relu_floor = a.act == VINET_ACT_RELU ? 0.f : -INFINITY;
  const bool sigm = a.act == VINET_ACT_SIGMOID;

  for (int item = blockIdx.x; item < a.items; item += gridDim.x) {
    const int b = (int)fdiv((uint32_t)item, a.dPatches);
    const int patch = item - b * a.patches;
    const int pos0 = patch * P;
    const char* xb = a.x + ((long)b * a.sBx + (long)(pos0 + l_row) * a.ldx + c4 * 4) * 4;
    char* yb = a.y + ((long)b * a.sBy + (long)(pos0 + l_row) * a.ldy + c4 * 4) * 4;
    const long x_plane = (long)a.HW * a.ldx * 4, y_plane = (long)a.HW * a.ldy * 4;
    const long x_r16 = 16L * a.ldx * 4, y_r16 = 16L * a.ldy * 4;

    for (int g = 0; g < k; ++g) {
      const int p = g - a.pad;
      const bool live = (unsigned)p < (unsigned)a.Ti;
      const uint4 v0 = *(const uint4*)(xb + (live ? p : 0) * x_plane), v1 = *(const uint4*)(xb + (live ? p : 0) * x_plane + x_r16);
      char* slot = ring + ((p + 2 * KMAX) % k) * FRAME;
      put(slot, 0, v0, live);
      put(slot, 1, v1, live);
    }
    __syncthreads();

    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
    for (int to = 0; to < a.To; ++to) {
      const bool more = to + 1 < a.To;
      const int pnew = (to + 1) * s - a.pad + k - s;
      const bool in0 = more && (unsigned)pnew < (unsigned)a.Ti;
      const bool in1 = more && s == 2 && (unsigned)(pnew + 1) < (unsigned)a.Ti;
      const char* xs0 = xb + (in0 ? pnew : 0) * x_plane;
      const char* xs1 = xb + (in1 ? pnew + 1 : 0) * x_plane;
      const uint4 nx00 = *(const uint4*)xs0, nx01 = *(const uint4*)(xs0 + x_r16);
      const uint4 nx10 = *(const uint4*)xs1, nx11 = *(const uint4*)(xs1 + x_r16);
      char* yf = yb + (long)(to * a.omT + a.ooT) * y_plane;
      float4 old0 = make_float4(0.f, 0.f, 0.f, 0.f), old1 = old0;
      if (a.accumulate) { old0 = *(const float4*)yf; old1 = *(const float4*)(yf + y_r16); }

      f32x4_v acc[2];
      acc[0] = (f32x4_v){0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0];
      const int s0 = (to * s - a.pad + 2 * KMAX) % k;
#pragma unroll
      for (int i = 0; i < KMAX; ++i) {
        if (i < k) {
          int si = s0 + rel[i];
          si -= si >= k ? k : 0;
          const char* fr = ring + si * FRAME;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            bf16x8_v ah[2], al[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
              ah[mt] = *(const bf16x8_v*)(fr + fr_off[mt][ks]);
              al[mt] = *(const bf16x8_v*)(fr + PLANE + fr_off[mt][ks]);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {      // small terms first
              mfma_ts3(acc[mt], wl[i][ks], ah[mt]);
              mfma_ts3(acc[mt], wh[i][ks], al[mt]);
              mfma_ts3(acc[mt], wh[i][ks], ah[mt]);
            }
          }
        }
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = aff_out ? fmaf(acc[mt][r], osc[r], osh[r]) : (float)acc[mt][r];
          ssum[r] += v; ssq[r] = fmaf(v, v, ssq[r]);
          o[r] = fmaxf(v, relu_floor);
          if (sigm) o[r] = 1.f / (1.f + __expf(-o[r]));
        }
        *(float4*)(stage + st_off[mt]) = make_float4(o[0], o[1], o[2], o[3]);
      }
      __syncthreads();          // ring frames of this step are free, the output tile is complete
      {
        float4 o0 = *(const float4*)(stage + l_row * 256 + ((c4 ^ (l_row & 15)) * 16));
        float4 o1 = *(const float4*)(stage + (l_row + 16) * 256 + ((c4 ^ ((l_row + 16) & 15)) * 16));
        if (a.accumulate) {
          o0.x += old0.x; o0.y += old0.y; o0.z += old0.z; o0.w += old0.w;
          o1.x += old1.x; o1.y += old1.y; o1.z += old1.z; o1.w += old1.w;
        }
        *(float4*)yf = o0;
        *(float4*)(yf + y_r16) = o1;
      }
      if (more) {
        char* slot0 = ring + ((pnew + 2 * KMAX) % k) * FRAME;
        put(slot0, 0, nx00, in0);
        put(slot0, 1, nx01, in0);
        if (s == 2) {
          char* slot1 = ring + ((pnew + 1 + 2 * KMAX) % k) * FRAME;
          put(slot1, 0, nx10, in1);
          put(slot1, 1, nx11, in1);
        }
      }
      __syncthreads();
    }
    if (a.stats) {     // a wave holds all 32 positions of its 16 channels: 16 lanes of a row by DPP, no LDS
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ss = ts_row16_sum(ssum[r]), qq = ts_row16_sum(ssq[r]);
        if (ep == 0) {
          const int n = wave * 16 + eq * 4 + r;
          a.stats[((long)item * 2 + 0) * 64 + n] = ss;
          a.stats[((long)item * 2 + 1) * 64 + n] = qq;
        }
      }
    }
  }
}

int g_vinet_opt_conv_ts = 1;   // 0 = off, 2 = force on every eligible shape (tests)

int g_vinet_opt_conv_ts_segs = 1;   // frame segments for launches without statistics (0 = whole patches only)
// frame segments per patch: enough items for one round of 512 workgroups, at least 4 output frames each (a segment re-reads
// k - s input frames of its predecessor); 1 in the split form.  (Not a function of d->stats: the engine asks
// vinet_conv3d_stats_rows before it has a statistics buffer to point at.)
int vinet_conv_ts_segments(const VinetConvDesc* d) {
  if (!g_vinet_opt_conv_ts_segs || d->dtype == VINET_F32S) return 1;
  const long patches = (long)d->x.B * (((long)d->oH * d->oW) / 64);
  long segs = (512 + patches - 1) / patches;
  if (segs > d->oT / 4) segs = d->oT / 4;
  if (segs < 1) segs = 1;
  const long seg_frames = (d->oT + segs - 1) / segs;
  return (int)((d->oT + seg_frames - 1) / seg_frames);  // (the count the launch really uses: equal segments, the last one ragged)
}

bool vinet_conv_use_ts(const VinetConvDesc* d) {
  const bool split = d->dtype == VINET_F32S && d->out_dtype == VINET_F32;       // conv_ts3_kernel: fp32 tensors, hi / lo weight planes
  if (!g_vinet_opt_conv_ts || d->tline != 1 || !((d->dtype == VINET_BF16 && d->out_dtype == VINET_BF16) || split) || d->mode != VINET_CONV_GENERIC) return false;
  if (d->pre.scale && !(d->pre.relu && d->pre.shift)) return false;
  if (d->pre.relu && !d->pre.scale) return false;
  const long HW = (long)d->oH * d->oW;
  const int al = split ? 4 : 8;
  const bool shape = d->x.C == 64 && d->y.C == 64 && (d->n_valid == 0 || d->n_valid == 64) && d->Kp == 64 && d->ntaps >= 2 && d->ntaps <= 7 &&
                     (d->sT == 1 || d->sT == 2) && d->ntaps >= d->sT && d->sH == 1 && d->sW == 1 && d->omH == 1 && d->omW == 1 && d->ooH == 0 &&
                     d->ooW == 0 && d->x.H == d->oH && d->x.W == d->oW && d->y.H == d->oH && d->y.W == d->oW && HW % 64 == 0 &&
                     d->tpad >= 0 && d->tpad < d->ntaps && d->x.ld % al == 0 && d->y.ld % al == 0 && d->x.sB % al == 0 && d->y.sB % al == 0 &&
                     ((uintptr_t)d->x.ptr % 16) == 0 && ((uintptr_t)d->y.ptr % 16) == 0;
  if (!shape) return false;
  if (g_vinet_opt_conv_ts >= 2) return true;
  // the frames of a patch split into segments, so small batches fill the chip: batch 1 = 336 patches x 2 segments of 8 output
  // frames (conv_dma on this layer: 66 us per clip whatever the batch)
  if (!split && g_vinet_opt_conv_ts_segs && d->oT >= 4) return (long)d->x.B * (HW / 64) * vinet_conv_ts_segments(d) >= 384;
  return (long)d->x.B * (HW / 64) >= 2048 && d->oT >= 4;
}
// positions per workgroup = per statistics row
int vinet_conv_ts_positions(const VinetConvDesc* d) { return d->dtype == VINET_F32S ? 32 : 64; }

int vinet_launch_conv_ts(const VinetConvDesc* d, hipStream_t s) {
  ConvTsArgs a;
  a.x = (const char*)d->x.ptr; a.y = (char*)d->y.ptr; a.w = (const char*)d->w; a.taps = (const int4*)d->taps;
  a.in_scale = d->pre.scale; a.in_shift = d->pre.shift;
  a.out_scale = d->out_scale; a.out_shift = d->out_shift; a.stats = d->stats;
  a.Ti = d->x.T; a.To = d->oT; a.HW = d->oH * d->oW; a.ldx = d->x.ld; a.ldy = d->y.ld; a.sBx = d->x.sB; a.sBy = d->y.sB;
  a.k = d->ntaps; a.s = d->sT; a.pad = d->tpad; a.omT = d->omT; a.ooT = d->ooT; a.act = d->act; a.accumulate = d->accumulate;
  VN_CHECK_ARG((d->oT - 1) * d->omT + d->ooT < d->y.T && d->ooT >= 0 && d->omT > 0, "conv_ts: output placement outside y");
  if (d->dtype == VINET_F32S) {
    a.patches = a.HW / 32;
    a.segs = 1; a.seg_frames = a.To; a.dSegs = make_fastdiv(1u);
    a.items = d->x.B * a.patches;
    a.dPatches = make_fastdiv((uint32_t)a.patches);
    const int smem3 = 7 * 2 * 32 * 128 + 32 * 256;
    static bool attr3_done[64] = {false};
    int dev3 = 0;
    (void)hipGetDevice(&dev3);
    if (!attr3_done[dev3 & 63]) {
      hipError_t e = hipFuncSetAttribute((const void*)conv_ts3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem3);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_ts3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem3);
      if (e != hipSuccess) { vinet_set_error("hipFuncSetAttribute(conv_ts3): %s", hipGetErrorString(e)); return (int)e; }
      attr3_done[dev3 & 63] = true;
    }
    int grid3 = 512;
    if (grid3 > a.items) grid3 = a.items;
    if (d->pre.scale) hipLaunchKernelGGL(conv_ts3_kernel<true>, dim3(grid3), dim3(256), smem3, s, a);
    else hipLaunchKernelGGL(conv_ts3_kernel<false>, dim3(grid3), dim3(256), smem3, s, a);
    return vn_launch_status("conv_ts3");
  }
  a.patches = a.HW / 64;
  a.segs = vinet_conv_ts_segments(d);
  a.seg_frames = (a.To + a.segs - 1) / a.segs;
  a.segs = (a.To + a.seg_frames - 1) / a.seg_frames;
  a.dSegs = make_fastdiv((uint32_t)a.segs);
  a.items = d->x.B * a.patches * a.segs;
  a.dPatches = make_fastdiv((uint32_t)a.patches);
  const int smem = 8 * 64 * 64 * 2 + 2 * 64 * 2 * 4;
  void (*const kern[2][2])(const ConvTsArgs) = {{conv_ts_kernel<false, false>, conv_ts_kernel<false, true>},
                                                {conv_ts_kernel<true, false>, conv_ts_kernel<true, true>}};     // [PRE][AFF]
  static bool attr_done[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_done[dev & 63]) {
    for (int i = 0; i < 4; ++i) {
      hipError_t e = hipFuncSetAttribute((const void*)kern[i >> 1][i & 1], hipFuncAttributeMaxDynamicSharedMemorySize, smem);
      if (e != hipSuccess) { vinet_set_error("hipFuncSetAttribute(conv_ts): %s", hipGetErrorString(e)); return (int)e; }
    }
    attr_done[dev & 63] = true;
  }
  int grid = 512;
  if (grid > a.items) grid = a.items;
  hipLaunchKernelGGL(kern[d->pre.scale ? 1 : 0][(d->out_scale || d->out_shift) ? 1 : 0], dim3(grid), dim3(256), smem, s, a);
  return vn_launch_status("conv_ts");
}

// ---------------------------------------------------------------------------------------------------------------
// Data gradient of a STRIDED temporal conv in one launch (VinetConvDesc::tline == 3):
//
//   dx[b, ti, p, c] (+)= sum over kt with (ti + pad - kt) % s == 0 of  sum_n dy[b, (ti + pad - kt) / s, p, n] * wt[kt][c][n]
//
// The per-phase form (one conv_ts launch per stride phase, engine._phase_taps_1d) reads dy once per phase.  Here the
// workgroup walks ALL input frames ti; the dy frames live in the LDS ring (a new one every s steps), every tap's
// weights stay in registers, and the taps of the step's phase are picked by a uniform predicate: dy is read once.
struct ConvTsdArgs {
  const char* x;        // dy [B][To][HW][64]
  char* y;              // dx [B][Ti][HW][64]
  const char* w;        // transposed pack [k][64 c][64 n]
  int To, Ti, HW, ldx, ldy;
  long sBx, sBy;
  int k, s, pad, accumulate;
  int items, patches;
  FastDiv dPatches;
  // BNB (VinetConvDesc::bnb_*): dx is the gradient behind a BatchNorm + ReLU whose raw input is z (same dims as dx): the
  // partial sums of vinet_bn_bwd_reduce(dx, z) leave with it, one row per item
  const char* z;
  long sBz;
  int ldz, z_relu;
  const float* z_scale;
  const float* z_shift;
  const float* z_mean;
  const float* z_invstd;
  float* partials;
};

// BNB: the BatchNorm-backward reduce pass over (dx, z) folded into the epilogue.  With the operands swapped a lane holds 4
// consecutive channels of a position: one 8-byte load of z per tile (prefetched with the frame's dy), gate + two fused
// multiply-adds per element into sums that stay in registers over ALL frames of the item, one cross-lane reduction per item.
// The stem's first BatchNorm (64 channels x 32 x 112 x 192 per clip: the largest BN'd tensor of the net) otherwise costs a
// separate pass over dx and z at the very end of the backward pass, where nothing is left to overlap it with.
template <bool BNB>
__global__ __launch_bounds__(256, 2) void conv_tsd_kernel(const ConvTsdArgs a) {
  constexpr int KMAX = 7, TILE = 64 * 64 * 2, NSLOT = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;                                   // NSLOT dy frames, slot = frame & 7
  char* stage = smem + NSLOT * TILE;
  float* red = (float*)(smem + (NSLOT + 1) * TILE);    // BNB: [2 position halves][64 channels][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int k = a.k, s = a.s;
  const int l_chunk = tid & 7, l_row = tid >> 3;
  int l_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = l_row + 32 * j;
    l_off[j] = r * 128 + ((l_chunk ^ (r & 7)) * 16);
  }
  bf16x8_v wr[KMAX][2][2];
#pragma unroll
  for (int i = 0; i < KMAX; ++i)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int c = wn * 32 + nt * 16 + (lane & 15), n = ks * 32 + (lane >> 4) * 8;
        if (i < k) wr[i][nt][ks] = *(const bf16x8_v*)(a.w + (((long)i * 64 + c) * 64 + n) * 2);
        else wr[i][nt][ks] = (bf16x8_v){0, 0, 0, 0, 0, 0, 0, 0};
      }
  const bf16x8_v zero_a = (bf16x8_v){0, 0, 0, 0, 0, 0, 0, 0};
  int st_off[2][2];                                    // byte offset of my 8 bytes (4 channels) of tile (mt, nt) in the output stage
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int row = wm * 32 + mt * 16 + (lane & 15), col = wn * 32 + nt * 16 + (lane >> 4) * 4;
      st_off[mt][nt] = row * 128 + (((col >> 3) ^ (row & 7)) * 16) + (col & 7) * 2;
    }
  // BNB: gate constants (relu(scale * z + shift) > 0) of the 64 channels in LDS -- read back a float4 at a time in the epilogue:
  // 16 registers fewer over the K loop --, element offsets of my 8 bytes of z per tile
  float* zconst = red + 256;                           // [64 scale][64 shift]
  int z_off[2][2];
  if constexpr (BNB) {
    if (tid < 64) {
      zconst[tid] = a.z_scale ? a.z_scale[tid] : 1.f;
      zconst[64 + tid] = a.z_shift ? a.z_shift[tid] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        z_off[mt][nt] = (wm * 32 + mt * 16 + (lane & 15)) * a.ldz + wn * 32 + nt * 16 + (lane >> 4) * 4;
  }

  for (int item = blockIdx.x; item < a.items; item += gridDim.x) {
    const int b = (int)fdiv((uint32_t)item, a.dPatches);
    const int pos0 = (item - b * a.patches) * 64;
    const char* xb = a.x + ((long)b * a.sBx + (long)(pos0 + l_row) * a.ldx + l_chunk * 8) * 2;
    char* yb = a.y + ((long)b * a.sBy + (long)(pos0 + l_row) * a.ldy + l_chunk * 8) * 2;
    const long x_plane = (long)a.HW * a.ldx * 2, y_plane = (long)a.HW * a.ldy * 2;
    const long x_r32 = 32L * a.ldx * 2, y_r32 = 32L * a.ldy * 2;
    const char* zb = nullptr;
    long z_plane = 0;
    float s1[2][4], s2[2][4];                            // sum g, sum g * z of my channels over the item (g gated by the ReLU)
    if constexpr (BNB) {
      zb = a.z + ((long)b * a.sBz + (long)pos0 * a.ldz) * 2;
      z_plane = (long)a.HW * a.ldz * 2;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[nt][r] = 0.f; s2[nt][r] = 0.f; }
    }

    // ---- prologue: dy frames 0 .. pad / s (everything step 0 can read) -------------------------------------
    const int q_first = a.pad / s;
    for (int f = 0; f <= q_first && f < a.To; ++f) {
      char* slot = ring + (f & (NSLOT - 1)) * TILE;
      *(uint4*)(slot + l_off[0]) = *(const uint4*)(xb + f * x_plane);
      *(uint4*)(slot + l_off[1]) = *(const uint4*)(xb + f * x_plane + x_r32);
    }
    __syncthreads();

    for (int ti = 0; ti < a.Ti; ++ti) {
      const int tp = ti + a.pad;
      const int q = tp / s, r = tp - q * s;
      // next step needs frame q + 1 iff (tp + 1) % s == 0
      const int qn = (tp + 1) / s;
      const bool newf = ti + 1 < a.Ti && qn != q && qn < a.To;
      const char* xs = xb + (newf ? qn : 0) * x_plane;
      const uint4 nx0 = *(const uint4*)xs, nx1 = *(const uint4*)(xs + x_r32);
      char* yf = yb + (long)ti * y_plane;
      uint4 old0 = make_uint4(0, 0, 0, 0), old1 = old0;
      if (a.accumulate) { old0 = *(const uint4*)yf; old1 = *(const uint4*)(yf + y_r32); }
      uint2 zq[2][2];
      if constexpr (BNB) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) zq[mt][nt] = *(const uint2*)(zb + (long)ti * z_plane + (long)z_off[mt][nt] * 2);
      }

      f32x4_v acc[2][2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) mfma_bf16_first_bacc(acc[mt][nt], zero_a, wr[0][nt][0]);     // = 0
#pragma unroll
      for (int i = 0; i < KMAX; ++i) {
        const int d = i - r;                                   // tap i belongs to this step's phase iff d >= 0, d % s == 0
        if (i < k && d >= 0 && d % s == 0) {
          const int f = q - d / s;                             // dy frame it reads
          if (f >= 0 && f < a.To) {
            const char* fr = ring + (f & (NSLOT - 1)) * TILE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              bf16x8_v af[2];
#pragma unroll
              for (int mt = 0; mt < 2; ++mt) {
                const int row = wm * 32 + mt * 16 + (lane & 15), ch = ks * 4 + (lane >> 4);
                af[mt] = *(const bf16x8_v*)(fr + row * 128 + ((ch ^ (row & 7)) * 16));
              }
#pragma unroll
              for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) mfma_bf16_acc_bacc(acc[mt][nt], af[mt], wr[i][nt][ks]);
            }
          }
        }
      }
      mfma_drain();
      // lane (p = lane & 15, q = lane >> 4) holds channels 4q .. 4q+3 of position p of each tile (operands swapped): 8 bytes per tile
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const uint2 g2 = make_uint2(pack2bf(acc[mt][nt][0], acc[mt][nt][1]), pack2bf(acc[mt][nt][2], acc[mt][nt][3]));
          *(uint2*)(stage + st_off[mt][nt]) = g2;
          if constexpr (BNB) {      // on the ROUNDED gradient: what the apply pass (and a separate reduce pass) reads back
            const uint32_t gw[2] = {g2.x, g2.y}, zw[2] = {zq[mt][nt].x, zq[mt][nt].y};
            const int c4 = wn * 32 + nt * 16 + (lane >> 4) * 4;
            const float4 sc4 = *(const float4*)(zconst + c4), sh4 = *(const float4*)(zconst + 64 + c4);
            const float zsc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, zsh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float xv = __uint_as_float((r & 1) ? (zw[r >> 1] & 0xffff0000u) : (zw[r >> 1] << 16));
              float gg = __uint_as_float((r & 1) ? (gw[r >> 1] & 0xffff0000u) : (gw[r >> 1] << 16));
              if (a.z_relu && !(fmaf(xv, zsc[r], zsh[r]) > 0.f)) gg = 0.f;
              s1[nt][r] += gg;
              s2[nt][r] = fmaf(gg, xv, s2[nt][r]);
            }
          }
        }
      __syncthreads();
      {
        uint4 o0 = *(const uint4*)(stage + l_off[0]), o1 = *(const uint4*)(stage + l_off[1]);
        if (a.accumulate) {
          auto add2 = [](uint32_t p, uint32_t q2) -> uint32_t {
            return pack2bf(__uint_as_float(p << 16) + __uint_as_float(q2 << 16),
                           __uint_as_float(p & 0xffff0000u) + __uint_as_float(q2 & 0xffff0000u));
          };
          o0 = make_uint4(add2(o0.x, old0.x), add2(o0.y, old0.y), add2(o0.z, old0.z), add2(o0.w, old0.w));
          o1 = make_uint4(add2(o1.x, old1.x), add2(o1.y, old1.y), add2(o1.z, old1.z), add2(o1.w, old1.w));
        }
        *(uint4*)yf = o0;
        *(uint4*)(yf + y_r32) = o1;
      }
      if (newf) {
        char* slot = ring + (qn & (NSLOT - 1)) * TILE;
        *(uint4*)(slot + l_off[0]) = nx0;
        *(uint4*)(slot + l_off[1]) = nx1;
      }
      __syncthreads();
    }
    if constexpr (BNB) {     // once per item: 16 positions of a row by DPP, the two position halves (wm) in LDS
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ss = ts_row16_sum(s1[nt][r]), qq = ts_row16_sum(s2[nt][r]);
          if ((lane & 15) == 0) {
            const int col = wn * 32 + nt * 16 + (lane >> 4) * 4 + r;
            red[(wm * 64 + col) * 2 + 0] = ss;
            red[(wm * 64 + col) * 2 + 1] = qq;
          }
        }
      __syncthreads();
      if (tid < 64) {      // sum g, and sum g * (z - mean) * invstd = (sum g z - mean sum g) * invstd
        const float ss = red[tid * 2] + red[(64 + tid) * 2], qq = red[tid * 2 + 1] + red[(64 + tid) * 2 + 1];
        a.partials[((long)item * 2 + 0) * 64 + tid] = ss;
        a.partials[((long)item * 2 + 1) * 64 + tid] = (qq - a.z_mean[tid] * ss) * a.z_invstd[tid];
      }
      __syncthreads();
    }
  }
}

// tline == 3: x = dy, y = dx, ntaps = k, sT = s of the FORWARD conv, tpad = its temporal padding, w = transposed pack
bool vinet_conv_use_tsd(const VinetConvDesc* d) {
  if (!g_vinet_opt_conv_ts || d->tline != 3 || d->dtype != VINET_BF16 || d->out_dtype != VINET_BF16 || d->mode != VINET_CONV_GENERIC) return false;
  if (d->pre.scale || d->pre.relu || d->stats || d->out_scale || d->out_shift || d->act != VINET_ACT_NONE) return false;
  const long HW = (long)d->oH * d->oW;
  const int k = d->ntaps, s = d->sT, p = d->tpad;
  const bool shape = d->x.C == 64 && d->y.C == 64 && (d->n_valid == 0 || d->n_valid == 64) && d->Kp == 64 && k >= 2 && k <= 7 && s >= 2 && s <= 4 &&
                     k >= s && (k + s - 1) / s <= 7 && p >= 0 && p < k && d->x.H == d->oH && d->x.W == d->oW && d->y.H == d->oH && d->y.W == d->oW &&
                     HW % 64 == 0 && d->oT == d->y.T && d->x.T == (d->y.T + 2 * p - k) / s + 1 && d->omT == 1 && d->omH == 1 && d->omW == 1 &&
                     d->ooT == 0 && d->ooH == 0 && d->ooW == 0 && d->x.ld % 8 == 0 && d->y.ld % 8 == 0 && d->x.sB % 8 == 0 && d->y.sB % 8 == 0 &&
                     ((uintptr_t)d->x.ptr % 16) == 0 && ((uintptr_t)d->y.ptr % 16) == 0;
  if (!shape) return false;
  if (g_vinet_opt_conv_ts >= 2) return true;
  return (long)d->x.B * (HW / 64) >= 2048 && d->oT >= 4;
}

// rows of BatchNorm-backward partial sums a tline == 3 launch with bnb_* set writes (one per item), 0 = cannot
int vinet_conv_tsd_bnb_rows(const VinetConvDesc* d) {
  if (!d || !vinet_conv_use_tsd(d) || d->accumulate) return 0;
  if (!d->bnb_z || !d->bnb_mean || !d->bnb_invstd || d->bnb_ld % 4 != 0 || d->bnb_sB % 4 != 0 || ((uintptr_t)d->bnb_z % 8) != 0) return 0;
  if (d->bnb_fwd.relu && !(d->bnb_fwd.scale && d->bnb_fwd.shift)) return 0;
  return (int)((long)d->x.B * (((long)d->oH * d->oW) / 64));
}

int vinet_launch_conv_tsd(const VinetConvDesc* d, hipStream_t s) {
  ConvTsdArgs a;
  a.x = (const char*)d->x.ptr; a.y = (char*)d->y.ptr; a.w = (const char*)d->w;
  a.To = d->x.T; a.Ti = d->y.T; a.HW = d->oH * d->oW; a.ldx = d->x.ld; a.ldy = d->y.ld; a.sBx = d->x.sB; a.sBy = d->y.sB;
  a.k = d->ntaps; a.s = d->sT; a.pad = d->tpad; a.accumulate = d->accumulate;
  a.patches = a.HW / 64;
  a.items = d->x.B * a.patches;
  a.dPatches = make_fastdiv((uint32_t)a.patches);
  const int smem = 9 * 64 * 64 * 2 + 1024 + 512;
  static bool attr_done[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_done[dev & 63]) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_tsd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_tsd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) { vinet_set_error("hipFuncSetAttribute(conv_tsd): %s", hipGetErrorString(e)); return (int)e; }
    attr_done[dev & 63] = true;
  }
  int grid = 512;
  if (grid > a.items) grid = a.items;
  a.z = nullptr; a.partials = nullptr;
  if (d->bnb_partials) {
    VN_CHECK_ARG(vinet_conv_tsd_bnb_rows(d) > 0, "conv tsd: the BatchNorm-backward statistics (bnb_*) are not available for this problem; ask vinet_conv3d_bn_bwd_stats_rows first");
    a.z = (const char*)d->bnb_z; a.sBz = d->bnb_sB; a.ldz = d->bnb_ld; a.z_relu = d->bnb_fwd.relu;
    a.z_scale = d->bnb_fwd.scale; a.z_shift = d->bnb_fwd.shift; a.z_mean = d->bnb_mean; a.z_invstd = d->bnb_invstd;
    a.partials = d->bnb_partials;
    hipLaunchKernelGGL(conv_tsd_kernel<true>, dim3(grid), dim3(256), smem, s, a);
    return vn_launch_status("conv_tsd<bnb>");
  }
  hipLaunchKernelGGL(conv_tsd_kernel<false>, dim3(grid), dim3(256), smem, s, a);
  return vn_launch_status("conv_tsd");
}
