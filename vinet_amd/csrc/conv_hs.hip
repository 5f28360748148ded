// Forward of the RGB stem (3 -> 64, 1x7x7, stride (1,2,2), model_utils.py:144) over the folded view of the zero-padded
// 4-channel clip (see wgrad_hs.hip for the view): a persistent workgroup owns a 64-wide column strip of one frame
// and walks the output rows with the 7 live input rows (1072 B each) in an LDS ring -- 2 new rows per step, every
// input byte fetched once, and the overlapped view is just overlapping ds_read_b128 addresses (position v starts
// 16 bytes after position v-1).  conv_dma_kernel staged seven 64-byte-row tiles per 256 positions (1.7 TB/s).
//
//   y[b,t,ho,wo,n] = act(scale[n] * sum_kh sum_j w[kh][n][j] * x[b,t,2ho+kh, 16 B * wo + j] + shift[n]),  j = kw*4 + c
//
//   * 2 x 2 waves, wave (wm, wn) = positions [32wm, +32) x channels [32wn, +32); its slice of all 7 rows' weights in
//     registers (14 B-fragments, read from the accumulator file as in conv_ts.hip);
//   * per output row: scale / shift / activation, BN partial sums (one `stats` row per 64 positions), bf16 tile
//     through LDS, 16-byte coalesced stores;
//   * the padded image makes every access in range: no bounds checks.
#include "common.h"

struct ConvHsArgs {
  const char* x;
  char* y;
  const char* w;
  const float* out_scale;
  const float* out_shift;
  float* stats;
  long sBx, sBy;
  int T, Hp, Wv, ldx;
  int oH, oW, ldy, act;
  int items, strips;
  FastDiv dStrips, dT;
  // row segments (conv_hs_kernel, small batches): an item is rows [seg * seg_rows, ...) of a strip, so that
  // a batch-1 clip (96 strips) still fills the chip; 1 = whole strips
  int segs, seg_rows;
  FastDiv dSegs;
};

// MFMAs (see conv_ts.hip): B (the weights) is read from the accumulator file -- the 112 weight registers do not fit in
// the 128 architectural VGPRs the compiler budgets beside everything else, and gfx90a+ MFMAs read A / B from
// either file.  Wherever the register allocator keeps a value, it may copy it (v_accvgpr_write / _read) right in
// front of the inline-asm MFMA, a VALU-write -> MFMA-read hazard it cannot see through the asm: every MFMA
// carries its own wait states.  (4 cycles x 56 MFMAs per output frame, against ~3000 cycles of HBM time.)
// Operands SWAPPED (round 4): the weights are the A operand, so the accumulators arrive transposed -- lane (p = lane & 15,
// q = lane >> 4) holds 4 consecutive CHANNELS 4q .. 4q+3 of position p per 16 x 16 tile (conv_igemm.h: conv_epilogue).
VN_DEV void mfma_hs_acc(f32x4_v& acc, const bf16x8_v& a, const bf16x8_v& b) {
  asm volatile("s_nop 3\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(b), "v"(a));
}
// first MFMA of an output frame: C = 0 as an inline constant (early clobber: the result must not share registers
// with the weights)
VN_DEV void mfma_hs_first(f32x4_v& acc, const bf16x8_v& a, const bf16x8_v& b) {
  asm volatile("s_nop 3\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&a"(acc) : "a"(b), "v"(a));
}
VN_DEV float hs_row16_sum(float v) {      // sum over the 16 lanes of a row (DPP row_ror 8, 4, 2, 1): every lane gets the total
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
  return v;
}

__global__ __launch_bounds__(256, 2) void conv_hs_kernel(const ConvHsArgs a) {
  constexpr int ROW = 1088, NP = 67, TILE = 64 * 64 * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* stage = smem;                                  // output tile [64 positions][64 channels] bf16
  float* red = (float*)(smem + TILE);                  // [2 position halves][64 channels][2]
  char* ring = smem + TILE + 1024;                     // 7 input rows
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l_chunk = tid & 7, l_row = tid >> 3;
  int l_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = l_row + 32 * j;
    l_off[j] = r * 128 + ((l_chunk ^ (r & 7)) * 16);
  }
  const bool xl = tid < 2 * NP;
  const int x_r = tid >= NP ? 1 : 0, x_p = tid - x_r * NP;

  bf16x8_v wr[7][2];
#pragma unroll
  for (int g = 0; g < 7; ++g)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int n = wn * 32 + nt * 16 + (lane & 15);
      wr[g][nt] = *(const bf16x8_v*)(a.w + (((long)g * 64 + n) * 32 + (lane >> 4) * 8) * 2);
    }
  // epilogue role: position p of a 16-row tile, channel quad q of a 16-column tile
  const int ep = lane & 15, eq = lane >> 4;
  float osc[2][4], osh[2][4];
  int st_off[2][2];                                    // byte offset of my 8 bytes (4 channels) of tile (mt, nt) in the output stage
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = wn * 32 + nt * 16 + eq * 4 + r;
      osc[nt][r] = a.out_scale ? a.out_scale[n] : 1.f;
      osh[nt][r] = a.out_shift ? a.out_shift[n] : 0.f;
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
    const int sitem = a.segs > 1 ? (int)fdiv((uint32_t)item, a.dSegs) : item;      // (strip item, row segment)
    const int h0 = (item - sitem * a.segs) * a.seg_rows;
    const int h1 = h0 + a.seg_rows < a.oH ? h0 + a.seg_rows : a.oH;
    const int bt = (int)fdiv((uint32_t)sitem, a.dStrips);
    const int strip = sitem - bt * a.strips;
    const int wo0 = strip * 64;
    const int b = (int)fdiv((uint32_t)bt, a.dT);
    const int t = bt - b * a.T;
    const char* xrow0 = a.x + ((long)b * a.sBx + ((long)t * a.Hp * a.Wv + wo0) * (long)a.ldx) * 2;
    const long x_rowb = (long)a.Wv * a.ldx * 2;
    char* yb = a.y + ((long)b * a.sBy + ((long)t * a.oH * a.oW + wo0 + l_row) * (long)a.ldy + l_chunk * 8) * 2;
    const long y_rowb = (long)a.oW * a.ldy * 2, y_r32 = 32L * a.ldy * 2;

    for (int q = tid; q < 7 * NP; q += 256) {      // input rows 2 h0 .. 2 h0 + 6 into their ring slots (row % 7)
      const int h = q / NP, pc = q - h * NP;
      *(uint4*)(ring + ((2 * h0 + h) % 7) * ROW + pc * 16) = *(const uint4*)(xrow0 + (2 * h0 + h) * x_rowb + pc * 16);
    }
    __syncthreads();

    // BN partial sums of my 2 x 4 channels over every row of the strip (one `stats` row per ITEM: vinet_conv3d_stats_rows)
    float ssum[2][4], ssq[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { ssum[nt][r] = 0.f; ssq[nt][r] = 0.f; }

    for (int ho = h0; ho < h1; ++ho) {
      const bool more = ho + 1 < h1;
      const int hn = more ? 2 * ho + 7 + x_r : 0;
      const uint4 nx = *(const uint4*)(xrow0 + hn * x_rowb + (xl ? x_p : 0) * 16);

      f32x4_v acc[2][2];
      const int s0 = (2 * ho) % 7;
#pragma unroll
      for (int g = 0; g < 7; ++g) {
        const int si = s0 + g - (s0 + g >= 7 ? 7 : 0);
        const char* row = ring + si * ROW;
        bf16x8_v af[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int v = wm * 32 + mt * 16 + (lane & 15);
          af[mt] = *(const bf16x8_v*)(row + v * 16 + (lane >> 4) * 16);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            if (g == 0) mfma_hs_first(acc[mt][nt], af[mt], wr[g][nt]);
            else mfma_hs_acc(acc[mt][nt], af[mt], wr[g][nt]);
          }
      }
      mfma_drain();
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = fmaf(acc[mt][nt][r], osc[nt][r], osh[nt][r]);
            ssum[nt][r] += v; ssq[nt][r] = fmaf(v, v, ssq[nt][r]);
            o[r] = fmaxf(v, relu_floor);
            if (sigm) o[r] = 1.f / (1.f + __expf(-o[r]));
          }
          *(uint2*)(stage + st_off[mt][nt]) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
        }
      __syncthreads();
      {
        char* yf = yb + ho * y_rowb;
        *(uint4*)yf = *(const uint4*)(stage + l_off[0]);
        *(uint4*)(yf + y_r32) = *(const uint4*)(stage + l_off[1]);
      }
      if (more && xl) *(uint4*)(ring + ((2 * ho + 7 + x_r) % 7) * ROW + x_p * 16) = nx;
      __syncthreads();
    }
    if (a.stats) {     // once per item: 16 positions of a row by DPP, the two position halves (wm) in LDS
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ss = hs_row16_sum(ssum[nt][r]), qq = hs_row16_sum(ssq[nt][r]);
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


// ---- the same strip kernel in the split-bf16 form (VINET_F32S): fp32 folded clip, fp32 output, hi / lo weight planes ------------
// A position of the overlapped view is 8 fp32 pixels-x-channels = 32 bytes (position v starts 32 bytes after position v - 1); the K
// order inside a 32-wide chunk is the one vinet_pack_weights(VINET_F32S) gives the weight planes (conv_dma3.h): lane group q holds
// elements 4q .. 4q+3 and 16+4q .. 16+4q+3, i.e. the two 16-byte pieces at 32 v + 16 q and 32 v + 64 + 16 q of the row; the 8
// values are split into hi / lo in registers and every product takes three MFMAs (lo.hi + hi.lo + hi.hi, weights as the A operand).
// conv_dma3 staged the seven rows of every 128 output positions again (12.2 ms at 64 clips, 1.0 TB/s); here every input byte is
// fetched once.
// (the compiler's own MFMA: three dependent products per accumulator -- it pads their hazards itself; conv_ts3_kernel's inline-asm
//  form with fixed wait states returned stale accumulator elements)
VN_DEV void mfma_hs3(f32x4_v& acc, const bf16x8_v& w, const bf16x8_v& a) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, a, acc, 0, 0, 0);
}
VN_DEV void hs3_split(float x0, float x1, uint32_t& hi, uint32_t& lo) {
  hi = pack2bf(x0, x1);
  lo = pack2bf(x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u));
}

__global__ __launch_bounds__(256, 2) void conv_hs3_kernel(const ConvHsArgs a) {
  constexpr int ROW = 2176, NP = 134, TILE = 64 * 64 * 4;      // 67 positions x 32 B per input row; fp32 output tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* stage = smem;                                  // output tile [64 positions][64 channels] fp32, 16-byte chunk ^ (row & 15)
  float* red = (float*)(smem + TILE);                  // [2 position halves][64 channels][2]
  char* ring = smem + TILE + 1024;                     // 7 input rows
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l_chunk = tid & 15, l_row = tid >> 4;      // store role: 16-byte chunk of a row, rows l_row + 16 j
  const int ep = lane & 15, eq = lane >> 4;

  bf16x8_v wh[7][2], wl[7][2];
#pragma unroll
  for (int g = 0; g < 7; ++g)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int n = wn * 32 + nt * 16 + (lane & 15);
      const char* wr = a.w + (((long)g * 64 + n) * 64 + eq * 8) * 2;     // row = [32 hi | 32 lo] bf16
      wh[g][nt] = *(const bf16x8_v*)wr;
      wl[g][nt] = *(const bf16x8_v*)(wr + 64);
    }
  float osc[2][4], osh[2][4];
  int st_off[2][2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = wn * 32 + nt * 16 + eq * 4 + r;
      osc[nt][r] = a.out_scale ? a.out_scale[n] : 1.f;
      osh[nt][r] = a.out_shift ? a.out_shift[n] : 0.f;
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = wm * 32 + mt * 16 + ep, chunk = wn * 8 + nt * 4 + eq;
      st_off[mt][nt] = row * 256 + ((chunk ^ (row & 15)) * 16);
    }
  }
  const float relu_floor = a.act == VINET_ACT_RELU ? 0.f : -INFINITY;
  const bool sigm = a.act == VINET_ACT_SIGMOID;
  // input role: piece tid of the step's 2 x 134 pieces, and piece 256 + tid for the first 12 threads
  const int p0r = tid >= NP ? 1 : 0, p0c = tid - p0r * NP;
  const bool two = tid < 2 * NP - 256;
  const int p1c = 256 + tid - NP;                      // (row 1)

  for (int item = blockIdx.x; item < a.items; item += gridDim.x) {
    const int bt = (int)fdiv((uint32_t)item, a.dStrips);
    const int strip = item - bt * a.strips;
    const int wo0 = strip * 64;
    const int b = (int)fdiv((uint32_t)bt, a.dT);
    const int t = bt - b * a.T;
    const char* xrow0 = a.x + ((long)b * a.sBx + ((long)t * a.Hp * a.Wv + wo0) * (long)a.ldx) * 4;
    const long x_rowb = (long)a.Wv * a.ldx * 4;
    char* yb = a.y + ((long)b * a.sBy + ((long)t * a.oH * a.oW + wo0 + l_row) * (long)a.ldy) * 4 + l_chunk * 16;
    const long y_rowb = (long)a.oW * a.ldy * 4, y_r16 = 16L * a.ldy * 4;

    for (int q = tid; q < 7 * NP; q += 256) {
      const int h = q / NP, pc = q - h * NP;
      *(uint4*)(ring + h * ROW + pc * 16) = *(const uint4*)(xrow0 + h * x_rowb + pc * 16);
    }
    __syncthreads();

    float ssum[2][4], ssq[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { ssum[nt][r] = 0.f; ssq[nt][r] = 0.f; }

    for (int ho = 0; ho < a.oH; ++ho) {
      const bool more = ho + 1 < a.oH;
      const int hn = more ? 2 * ho + 7 : 0;
      const uint4 nx0 = *(const uint4*)(xrow0 + (long)(hn + (more ? p0r : 0)) * x_rowb + p0c * 16);
      const uint4 nx1 = *(const uint4*)(xrow0 + (long)(hn + (more ? 1 : 0)) * x_rowb + (two ? p1c : 0) * 16);

      f32x4_v acc[2][2];
      const int s0 = (2 * ho) % 7;
#pragma unroll
      for (int g = 0; g < 7; ++g) {
        const int si = s0 + g - (s0 + g >= 7 ? 7 : 0);
        const char* row = ring + si * ROW;
        bf16x8_v ah[2], al[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int v = wm * 32 + mt * 16 + (lane & 15);
          const uint4 f0 = *(const uint4*)(row + v * 32 + eq * 16), f1 = *(const uint4*)(row + v * 32 + 64 + eq * 16);
          union { bf16x8_v v8; uint32_t u[4]; } H, Lo;
          hs3_split(__uint_as_float(f0.x), __uint_as_float(f0.y), H.u[0], Lo.u[0]);
          hs3_split(__uint_as_float(f0.z), __uint_as_float(f0.w), H.u[1], Lo.u[1]);
          hs3_split(__uint_as_float(f1.x), __uint_as_float(f1.y), H.u[2], Lo.u[2]);
          hs3_split(__uint_as_float(f1.z), __uint_as_float(f1.w), H.u[3], Lo.u[3]);
          ah[mt] = H.v8; al[mt] = Lo.v8;
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {      // small terms first
            if (g == 0) acc[mt][nt] = (f32x4_v){0.f, 0.f, 0.f, 0.f};
            mfma_hs3(acc[mt][nt], wl[g][nt], ah[mt]);
            mfma_hs3(acc[mt][nt], wh[g][nt], al[mt]);
            mfma_hs3(acc[mt][nt], wh[g][nt], ah[mt]);
          }
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = fmaf(acc[mt][nt][r], osc[nt][r], osh[nt][r]);
            ssum[nt][r] += v; ssq[nt][r] = fmaf(v, v, ssq[nt][r]);
            o[r] = fmaxf(v, relu_floor);
            if (sigm) o[r] = 1.f / (1.f + __expf(-o[r]));
          }
          *(float4*)(stage + st_off[mt][nt]) = make_float4(o[0], o[1], o[2], o[3]);
        }
      __syncthreads();
      {
        char* yf = yb + ho * y_rowb;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = l_row + 16 * j;
          *(uint4*)(yf + j * y_r16) = *(const uint4*)(stage + row * 256 + ((l_chunk ^ (row & 15)) * 16));
        }
      }
      if (more) {
        *(uint4*)(ring + ((2 * ho + 7 + p0r) % 7) * ROW + p0c * 16) = nx0;
        if (two) *(uint4*)(ring + ((2 * ho + 8) % 7) * ROW + p1c * 16) = nx1;
      }
      __syncthreads();
    }
    if (a.stats) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ss = hs_row16_sum(ssum[nt][r]), qq = hs_row16_sum(ssq[nt][r]);
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

int g_vinet_opt_conv_hs = 1;   // 0 = off, 2 = force on every eligible shape (tests)
int g_vinet_opt_conv_hs_segs = 1;   // row segments for launches without statistics (0 = whole strips only)

// row segments per strip: enough items for one round of 768 workgroups, at least 7 output rows each (a segment re-reads 5 input
// rows of its upper neighbour); 1 in the split form.  (Not a function of d->stats: the engine asks vinet_conv3d_stats_rows
// before it has a statistics buffer to point at.)
int vinet_conv_hs_segments(const VinetConvDesc* d) {
  if (!g_vinet_opt_conv_hs_segs || d->dtype == VINET_F32S) return 1;
  const long strips = (long)d->x.B * d->oT * (d->oW / 64);
  long segs = (768 + strips - 1) / strips;
  if (segs > d->oH / 7) segs = d->oH / 7;
  if (segs < 1) segs = 1;
  const long seg_rows = (d->oH + segs - 1) / segs;
  return (int)((d->oH + seg_rows - 1) / seg_rows);      // (the count the launch really uses: equal segments, the last one ragged)
}

// VinetConvDesc::tline == 2: the caller promises taps (0, kh, 0, slice kh), kh = 0..6 (the folded stem)
bool vinet_conv_use_hs(const VinetConvDesc* d) {
  const bool split = d->dtype == VINET_F32S && d->out_dtype == VINET_F32;       // conv_hs3_kernel: fp32 folded clip, fp32 output
  if (!g_vinet_opt_conv_hs || d->tline != 2 || !((d->dtype == VINET_BF16 && d->out_dtype == VINET_BF16) || split) || d->mode != VINET_CONV_GENERIC) return false;
  if (split && (d->y.ld % 4 != 0 || d->y.sB % 4 != 0 || d->x.sB % 4 != 0)) return false;
  if (d->pre.scale || d->pre.relu || d->accumulate) return false;
  const bool shape = d->x.C == 32 && d->x.ld == 8 && d->Kp == 32 && d->ntaps == 7 && d->sT == 1 && d->sH == 2 && d->sW == 1 &&
                     d->y.C == 64 && (d->n_valid == 0 || d->n_valid == 64) && d->oW % 64 == 0 && d->oT == d->x.T && d->x.W >= d->oW + 3 &&
                     d->x.H >= 2 * d->oH + 5 && d->omT == 1 && d->omH == 1 && d->omW == 1 && d->ooT == 0 && d->ooH == 0 && d->ooW == 0 &&
                     d->y.T == d->oT && d->y.H == d->oH && d->y.W == d->oW && (split || (d->y.ld % 8 == 0 && d->y.sB % 8 == 0 && d->x.sB % 8 == 0)) &&
                     ((uintptr_t)d->x.ptr % 16) == 0 && ((uintptr_t)d->y.ptr % 16) == 0;
  if (!shape) return false;
  if (g_vinet_opt_conv_hs >= 2) return true;
  // a strip splits into row segments, so small batches fill the chip too: batch 1 = 96 strips x 8 segments of 14 rows (conv_dma's
  // stem form: 66 us per clip whatever the batch)
  if (!split && g_vinet_opt_conv_hs_segs && d->oH >= 8) return (long)d->x.B * d->oT * (d->oW / 64) * vinet_conv_hs_segments(d) >= 384;
  return (long)d->x.B * d->oT * (d->oW / 64) >= 512 && d->oH >= 8;      // (from 6 clips of 32 x 224 x 384 on: 8 clips 385 -> 402 clips/s with both strip kernels, profiles/r4_experiments.txt)
}

int vinet_launch_conv_hs(const VinetConvDesc* d, hipStream_t s) {
  ConvHsArgs a;
  a.x = (const char*)d->x.ptr; a.y = (char*)d->y.ptr; a.w = (const char*)d->w;
  a.out_scale = d->out_scale; a.out_shift = d->out_shift; a.stats = d->stats;
  a.sBx = d->x.sB; a.sBy = d->y.sB;
  a.T = d->x.T; a.Hp = d->x.H; a.Wv = d->x.W; a.ldx = d->x.ld;
  a.oH = d->oH; a.oW = d->oW; a.ldy = d->y.ld; a.act = d->act;
  a.strips = a.oW / 64;
  a.segs = vinet_conv_hs_segments(d);
  a.seg_rows = (a.oH + a.segs - 1) / a.segs;
  a.segs = (a.oH + a.seg_rows - 1) / a.seg_rows;
  a.dSegs = make_fastdiv((uint32_t)a.segs);
  a.items = d->x.B * a.T * a.strips * a.segs;
  a.dStrips = make_fastdiv((uint32_t)a.strips);
  a.dT = make_fastdiv((uint32_t)a.T);
  if (d->dtype == VINET_F32S) {
    const int smem3 = 64 * 64 * 4 + 1024 + 7 * 2176;
    int grid3 = 512;    // 2 workgroups per CU (28 weight fragments in the accumulator file)
    if (grid3 > a.items) grid3 = a.items;
    hipLaunchKernelGGL(conv_hs3_kernel, dim3(grid3), dim3(256), smem3, s, a);
    return vn_launch_status("conv_hs3");
  }
  const int smem = 64 * 64 * 2 + 1024 + 7 * 1088;
  int grid = 768;     // 3 workgroups per CU (152 registers, 17 KB of LDS)
  if (grid > a.items) grid = a.items;
  hipLaunchKernelGGL(conv_hs_kernel, dim3(grid), dim3(256), smem, s, a);
  return vn_launch_status("conv_hs");
}
