// Halo-tile implicit-GEMM convolution for kernels with 3x3 SPATIAL taps (bf16, gfx950): the 1x3x3 convs of the
// SepConv3d blocks (model_utils.py:144), the kT x 3 x 3 / stride-kT decoder convs (model.py:256-276) and every
// data gradient of those (one launch per temporal stride phase, each a 1x3x3 conv over dy).
//
// Why a third conv generation next to conv_dma.h / conv_pp.h: those stage an im2col A tile per K step, i.e. the
// activation patch is fetched from L2 once PER TAP (9x for a 3x3 kernel) -- 11.7 (conv_dma) / 7.8 (conv_pp) bytes
// staged per kFLOP, and the CU's L2 -> LDS path, not MFMA or HBM, bounds them.  Here a
// workgroup owns a SPATIAL tile of one output frame (TR rows x TW columns = 256 positions) and stages the input
// patch with its one-pixel halo ((TR+2) x (TW+2) positions x 64 channels = 43.5 KB) ONCE per (temporal tap,
// channel chunk); the nine spatial taps are nine fragment ADDRESSES into that halo image, not nine copies.  Per
// K step (tap, 64 channels) only the weight tile (BN x 128 B, L2 resident, contiguous) is staged: 4.8 KB of halo
// + 8...16 KB of weights per 2.1...4.2 MFLOP = 4...6 B/kFLOP.
//
//   * 256 threads = 4 waves stacked along M (wave = 64 positions x all BN columns), two workgroups per CU.
//   * LDS: [halo image: 352 positions x 128 B][B ring: BSLOTS x BN rows x 128 B].  Both are written by
//     `global_load_lds_dwordx4` in 8-row x 128-B pieces (whole cache lines), lane-linear on the LDS side, with the
//     16-byte chunk index XOR (row & 7) applied on the SOURCE address and again on the ds_read_b128: a fragment
//     read touches 16 consecutive rows (positions p0..p0+15, any alignment), whose (p & 7) take all 8 values twice
//     -> conflict-free (MI355X_MICROARCH.md, LDS table: ds_read_b128 lane groups).
//   * out-of-image halo positions, frames outside the clip and channels past Cin read a zero page (branch-free
//     source selection: every wave issues exactly HL / BL DMAs, so the counted `s_waitcnt vmcnt` is exact).
//   * K order: temporal tap (group) outer, channel chunk middle, the group's spatial taps inner.  One raw
//     s_barrier per K step; the weight tiles of the next BSLOTS-1 steps are in flight across it.  A new (group,
//     chunk) re-stages the halo image: one extra barrier + a full drain, hidden by the CU's second workgroup.
//   * epilogue: the shared conv_epilogue (BN partial sums, activation, accumulate, arbitrary placement) with the
//     row -> voxel map of the spatial tile.
#pragma once
#include "conv_dma.h"

// TM (temporal mode): the same machinery turned by 90 degrees for the (3,1,1) / stride-1 temporal convs of the SepConv3d
// blocks (model_utils.py:148) and their data gradients: a workgroup owns 64 positions x 4 consecutive output frames (one
// frame per wave), the "halo image" is the 6 input frames t0-1 .. t0+4 of those 64 positions (no spatial halo), the three
// taps are three ROW offsets into it: every activation byte is staged once per 4 output frames instead of three times.
// PRE: a pending BatchNorm + ReLU of the input (the conv_t of a SepConv3d reads conv_s's raw output) is applied ONCE per
// staged element, in LDS, by the wave that staged it -- conv_dma's PRE form pays it at every fragment read, i.e. per tap.
// SPLIT (VINET_F32S, conv_dma3.h's arithmetic on this kernel's tiles): fp32 activations, so a 128-byte halo row holds 32 channels
// and a K step is (tap, 32 channels); the weight rows are [32 bf16 hi | 32 bf16 lo] of vinet_pack_weights(VINET_F32S) -- the same
// 128 bytes, pieces and swizzle as the bf16 form.  A fragment read fetches pieces q and q + 4 of a row in both forms: 2 x 8 bf16
// there, 8 fp32 here (split into hi / lo in registers, once per fragment and K step: three MFMAs per product).
template <int NT, int TW, int BSLOTS, bool TM = false, bool PRE = false, bool SPLIT = false>
struct ConvHtCfg {
  static constexpr int KC = SPLIT ? 32 : 64;            // channels per K step
  static constexpr int ES = SPLIT ? 4 : 2;              // bytes per activation element
  static constexpr int PE = 16 / ES;                    // elements per 16-byte piece
  static constexpr int THREADS = 256, BM = 256, TR = TM ? 4 : BM / TW, HW = TM ? 64 : TW + 2, HR = TR + 2;
  static constexpr int NPOS = HR * HW;                  // halo positions
  static constexpr int HPIECES = (NPOS + 7) / 8;        // DMA pieces of 8 positions x 128 B
  static constexpr int HL = (HPIECES + 3) / 4;          // halo DMAs per wave
  static constexpr int HALO_BYTES = HL * 4 * 1024;
  static constexpr int BN = NT * 16;
  static constexpr int BL = NT / 2;                     // weight DMAs per wave and K step (BN / 8 pieces over 4 waves)
  static constexpr int BSLOT_BYTES = BN * 128;
  static constexpr int KLOOP_BYTES = HALO_BYTES + BSLOTS * BSLOT_BYTES;   // PRE: scale[Kp], shift[Kp] (fp32) behind it
  static constexpr int EROW = BN + 4;
  static constexpr int EPI_BYTES = conv_epi_bytes<4, NT, 4, 1>();
  static int smem_bytes(int Kp) {
    const int k = KLOOP_BYTES + (PRE ? 2 * Kp * 4 : 0);
    return k > EPI_BYTES ? k : EPI_BYTES;
  }
  static_assert(NT % 2 == 0 && (TW == 32 || TW == 16), "shapes");
  static_assert(BSLOTS >= 2 && BL * (BSLOTS - 2) <= 63, "vmcnt immediate range");
};

template <int NT, int TW, int BSLOTS, bool TM, bool PRE, bool SPLIT, bool BNB = false>
__global__ __launch_bounds__(256, 2) void conv_ht_kernel(const ConvArgs a) {
  using Cfg = ConvHtCfg<NT, TW, BSLOTS, TM, PRE, SPLIT>;
  constexpr int HW = Cfg::HW, TR = Cfg::TR, HL = Cfg::HL, BL = Cfg::BL, MT = 4, KC = Cfg::KC, ES = Cfg::ES, PE = Cfg::PE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const halo = smem;
  char* const bring = smem + Cfg::HALO_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* zero = (const char*)g_vinet_zero_page;

#ifdef VINET_CONV_TIMING
  const unsigned long long tm0 = __builtin_amdgcn_s_memtime();
  unsigned long long tm_halo = 0, tm_h0 = 0;
#endif
  // ---- workgroup -> (column tile, spatial tile, frame) ---------------------------------------------------------------
  const uint32_t wg = (uint32_t)xcd_remap(blockIdx.x, gridDim.x);
  const uint32_t sp = fdiv(wg, a.ht_dN);                      // spatial tile index = statistics row
  const int tile_n = (int)(wg - sp * (uint32_t)a.tilesN);
  const uint32_t q1 = fdiv(sp, a.ht_dW);
  const int tw_i = (int)(sp - q1 * (uint32_t)a.ht_tilesW);
  const uint32_t frame = fdiv(q1, a.ht_dH);
  const int th_i = (int)(q1 - frame * (uint32_t)a.ht_tilesH);
  // spatial: frame = b * To + to, tile (th_i, tw_i) of that frame.  temporal: frame = b, th_i = block of 4 output frames,
  // tw_i = block of 64 positions (ht_dTo divides by 1)
  const uint32_t bb = fdiv(frame, a.ht_dTo);
  const int to = TM ? th_i * 4 : (int)(frame - bb * (uint32_t)a.To), b = (int)bb;
  const int h0 = th_i * TR, w0 = tw_i * TW;
  const int p0 = tw_i * 64, HWtot = a.Hi * a.Wi;     // (temporal mode: first position of the tile, positions per frame)

  // ---- this lane's DMA role: row (lane >> 3) of an 8-row piece, LDS slot (lane & 7), source chunk slot ^ key(position) ----
  // Swizzle key of a halo position.  LINA (every form but the spatial PRE one): the position's COLUMN inside its halo row, & 7 --
  // a fragment read touches 16 consecutive columns of ONE halo row, whose keys take all 8 values twice (conflict-free exactly as
  // with p & 7), and the key no longer depends on the halo ROW: the four A fragments of a lane (rows r, r + 1, ... of the tile, or
  // columns c, c + 16) then share one swizzle term per tap and differ by COMPILE-TIME byte offsets, which go into the ds_read's
  // immediate -- 4 vector instructions of address arithmetic per K step instead of 26 (round 6; tools/isa_audit.py).  Temporal
  // mode: HW = 64, so column & 7 == p & 7 and nothing moves.  Spatial PRE keeps p & 7 (its in-LDS affine reads ONE scale / shift
  // group per lane, which needs one source chunk per lane).
  constexpr bool LINA = TM || !PRE;
  const int prow = lane >> 3;
  const int src_chunk = (lane & 7) ^ prow;                 // weight ring (rows n & 7 == prow) and the p & 7 halo forms
  int hal_off[HL];            // element offset of this lane's halo position inside a frame (+ its chunk); 0 when out of range
  unsigned hal_ok = 0;
  unsigned long long hal_scq = 0;                          // (LINA, spatial) this lane's source chunk of piece j, 3 bits each
#pragma unroll
  for (int j = 0; j < HL; ++j) {
    const int p = (j * 4 + wave) * 8 + prow;
    const int hr = p / HW, hc = p - hr * HW;
    const int sc = (LINA && !TM) ? ((lane & 7) ^ (hc & 7)) : src_chunk;
    hal_scq |= (unsigned long long)sc << (3 * j);
    if constexpr (TM) {     // halo row = frame t0 - 1 + hr (added at issue time: uniform per piece), column = position p0 + hc
      const bool ok = (p < Cfg::NPOS) & (p0 + hc < HWtot);
      hal_off[j] = ok ? (p0 + hc) * a.ldx + sc * PE : 0;
      hal_ok |= (unsigned)ok << j;
    } else {
      const int h = h0 - 1 + hr, w = w0 - 1 + hc;
      const bool ok = (p < Cfg::NPOS) & ((unsigned)h < (unsigned)a.Hi) & ((unsigned)w < (unsigned)a.Wi);
      hal_off[j] = ok ? (h * a.Wi + w) * a.ldx + sc * PE : 0;
      hal_ok |= (unsigned)ok << j;
    }
  }
  const char* const xb = a.x + (long)b * a.sBx * ES;
  const long frame_elems = (long)a.Hi * a.Wi * a.ldx;
  // weight piece j of this lane: row n of the column tile, chunk src_chunk -- as a POINTER for (slice 0, channel chunk 0); a K step
  // adds one wave-uniform 64-bit offset (v_lshl_add_u64 with a scalar pair: one instruction per DMA, where the masked zero-page
  // selection took eight).  No zero page on this side any more: rows past Nw read the LAST real row (their columns are never
  // stored: conv_epilogue masks n < N), steps past the group's last tap read a real tap into a slot nobody consumes, and the K
  // overhang of a last half chunk (Kp % 64 == 32, bf16 form) is redirected 64 bytes back inside the same row -- finite weights
  // against activations that ARE zero there (the halo side keeps its zero page for channels >= Cin).
  const char* wp[BL];
#pragma unroll
  for (int j = 0; j < BL; ++j) {
    const int n = (j * 4 + wave) * 8 + prow;
    const int nn = tile_n * Cfg::BN + n;
    wp[j] = a.w + ((long)(nn < a.Nw ? nn : a.Nw - 1) * a.Kp * (SPLIT ? 2 : 1) + src_chunk * 8) * 2;      // (split: a row is Kp hi + Kp lo)
  }
  const long slice_bytes = (long)a.Nw * a.Kp * (SPLIT ? 4 : 2);

  // (PRE: xform_halo's plain LDS loads / stores would make hipcc drain vmcnt(0) -- the weight tiles in flight too -- when the DMA
  //  is a builtin: common.h, lds_dma16_asm)
  auto dma = [&](const char* src, char* dst) {
    if constexpr (PRE) lds_dma16_asm(src, dst);
    else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                          (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  // halo image of (temporal offset dt, channel chunk c0)
  // which of this lane's halo elements are real activations in the image staged last (PRE: the others must stay zero)
  unsigned hal_live = 0;
  auto issue_halo = [&](int dt, int c0) {
    // channels past Cin read the zero page: only a LAST chunk can hold any (lim < 8 pieces of PE channels are real)
    const int lim = (a.Cin - c0 + PE - 1) / PE;
    hal_live = 0;
#pragma unroll
    for (int j = 0; j < HL; ++j) {
      // spatial: one frame (temporal offset dt of the tap group); temporal: piece (j, wave) belongs to frame to - 1 + its halo row
      const int t = TM ? to - 1 + ((j * 4 + wave) * 8) / HW : to * a.sT + dt;
      const int sc = (LINA && !TM) ? (int)((hal_scq >> (3 * j)) & 7u) : src_chunk;
      const unsigned ok = (unsigned)(sc < lim) & (unsigned)((unsigned)t < (unsigned)a.Ti) & ((hal_ok >> j) & 1u);
      const char* base = xb + ((long)t * frame_elems + c0) * ES;
      const char* src = zero + (((base + (long)hal_off[j] * ES) - zero) & -(long)ok);
      dma(src, halo + (j * 4 + wave) * 1024);
      hal_live |= ok << j;
    }
  };
  // PRE: relu(scale * x + shift) on this wave's own pieces of the image, in place (after its DMAs have landed, before the
  // barrier that publishes the image); padding stays zero
  float* const aff = (float*)(smem + Cfg::KLOOP_BYTES);
  auto xform_halo = [&](int c0) {
    if constexpr (SPLIT) {     // this lane's piece = 4 fp32 channels c0 + 4 src_chunk .. + 3
      const float* sp_ = aff + c0 + src_chunk * 4;
      const float4 sc = *(const float4*)sp_, sh = *(const float4*)(sp_ + a.Kp);
#pragma unroll
      for (int j = 0; j < HL; ++j) {
        float4* q = (float4*)(halo + (j * 4 + wave) * 1024 + lane * 16);
        const float4 v = *q;
        const bool live = (hal_live >> j) & 1u;
        *q = live ? make_float4(fmaxf(fmaf(v.x, sc.x, sh.x), 0.f), fmaxf(fmaf(v.y, sc.y, sh.y), 0.f), fmaxf(fmaf(v.z, sc.z, sh.z), 0.f),
                                fmaxf(fmaf(v.w, sc.w, sh.w), 0.f))
                  : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
    const float* sp_ = aff + c0 + src_chunk * 8;
    const float4 s0 = *(const float4*)sp_, s1 = *(const float4*)(sp_ + 4);
    const float4 h0_ = *(const float4*)(sp_ + a.Kp), h1_ = *(const float4*)(sp_ + a.Kp + 4);
    const f32x2_v sc2[4] = {{s0.x, s0.y}, {s0.z, s0.w}, {s1.x, s1.y}, {s1.z, s1.w}};
    const f32x2_v sh2[4] = {{h0_.x, h0_.y}, {h0_.z, h0_.w}, {h1_.x, h1_.y}, {h1_.z, h1_.w}};
#pragma unroll
    for (int j = 0; j < HL; ++j) {
      uint4* q = (uint4*)(halo + (j * 4 + wave) * 1024 + lane * 16);
      const uint4 v = *q;
      const uint32_t m = (hal_live >> j) & 1u ? 0xffffffffu : 0u;
      *q = make_uint4(pre_relu_pair(v.x, sc2[0], sh2[0]) & m, pre_relu_pair(v.y, sc2[1], sh2[1]) & m,
                      pre_relu_pair(v.z, sc2[2], sh2[2]) & m, pre_relu_pair(v.w, sc2[3], sh2[3]) & m);
    }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the raw s_barrier that follows does not wait for LDS stores)
  };
  if constexpr (PRE) {
    for (int c = tid; c < a.Kp; c += 256) {
      const bool in = c < a.Cin;
      aff[c] = in ? a.in_scale[c] : 0.f;
      aff[a.Kp + c] = in ? a.in_shift[c] : 0.f;
    }
    __syncthreads();   // (plain loads above are complete before any DMA is counted)
  }
  // weight tile at wave-uniform byte offset `delta` (slice * slice_bytes + chunk) into ring slot `slot`
  auto issue_b = [&](int slot, long delta) {
    char* dst = bring + slot * Cfg::BSLOT_BYTES + wave * 1024;
#pragma unroll
    for (int j = 0; j < BL; ++j) dma(wp[j] + delta, dst + j * 4096);
  };

  // ---- fragments ---------------------------------------------------------------------------------------------------------
  f32x4_v acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4_v){0.f, 0.f, 0.f, 0.f};
  // halo position of this lane's row of A fragment i for the centre tap; rows of a fragment are 16 consecutive columns
  int pl[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    if constexpr (TM) {
      pl[i] = (wave + 1) * HW + i * 16 + (lane & 15);         // wave = output frame of the tile, fragment i = 16 positions
    } else {
      const int r = TW == 32 ? 2 * wave + (i >> 1) : 4 * wave + i;
      const int c = TW == 32 ? (i & 1) * 16 : 0;
      pl[i] = (r + 1) * HW + (c + 1) + (lane & 15);
    }
  }
  // row -> voxel map of this wave for the epilogue: TR / 4 image rows of the tile, TW / 16 row groups each
  // (temporal: one "row" = this wave's output frame, four groups of 16 positions, as many of them as the frame still has)
  constexpr int WROWS = TM ? 1 : TR / 4;
  const int hw0 = h0 + wave * WROWS;
  EpiRows er;
  if constexpr (TM) {
    er.m0 = (int)(((uint32_t)b * (uint32_t)a.To + (uint32_t)(to + wave)) * (uint32_t)HWtot + (uint32_t)p0);
    er.ipr = 4;
    er.rstride = 0;
    er.nrows = to + wave < a.To ? 1 : 0;
    er.ncols = (HWtot - p0) / 16 < 4 ? (HWtot - p0) / 16 : 4;
  } else {
    er.m0 = (int)((frame * (uint32_t)a.Ho + (uint32_t)hw0) * (uint32_t)a.Wo + (uint32_t)w0);
    er.ipr = TW / 16;
    er.rstride = a.Wo;
    er.nrows = a.Ho - hw0 < 0 ? 0 : (a.Ho - hw0 > WROWS ? WROWS : a.Ho - hw0);
    er.ncols = TW / 16;
  }
  const int kq = lane >> 4;                                           // this lane's 16-byte k group inside a 32-wide K half
  const int bfo0 = (lane & 15) * 128 + (((kq) ^ (lane & 7)) << 4);    // weight fragment rows: n & 7 == lane & 7
  const int bfo1 = (lane & 15) * 128 + (((4 + kq) ^ (lane & 7)) << 4);
  // LINA: byte address of fragment 0's row for a tap = a128 + (tap's halo offset + 128) * 128 [scalar] + swizzle term, where the
  // term is ((column + dx) & 7) ^ kq, in units of 16 bytes: computed on (column + 1) << 4 with the tap's (dx + 1) << 4 added
  // [scalar], masked to bits 4..6 and XORed with kq << 4 -- one add, one bitop3, one add3; the second K half is the same address
  // ^ 64 (chunk 4 + kq).  Temporal mode: the key is the lane's own (p & 7), the same for every tap.  Fragment i adds a constant.
  const int a128 = (pl[0] - 128) << 7;
  const int cl16 = TM ? ((pl[0] & 7) << 4) : ((lane & 15) << 4);    // spatial: column = 1 + (lane & 15) + dx = (lane & 15) + (dx + 1)
  const int kq16 = kq << 4;
  auto frag_off = [](int i) constexpr {        // byte offset of A fragment i against fragment 0
    return TM ? i * 16 * 128 : (TW == 32 ? ((i >> 1) * HW + (i & 1) * 16) * 128 : i * HW * 128);
  };

  // one K step: the tap word's halo offset (biased by +128, times 128 bytes) and dx term ((dx + 1) << 4; 0 in temporal mode)
  auto compute = [&](int slot, int off128, int dx16) {
    const char* Bs = bring + slot * Cfg::BSLOT_BYTES;
    int ab0 = 0, ab1 = 0;
    if constexpr (LINA) {
      const int t16 = TM ? (cl16 ^ kq16) : ((((cl16 + dx16) & 0x70)) ^ kq16);
      ab0 = a128 + off128 + t16;
      ab1 = ab0 ^ 64;
    }
    const int tapoff = (off128 >> 7) - 128;      // (the p & 7 form only)
    auto a_addr = [&](int i, int kk) -> const char* {
      if constexpr (LINA) return halo + (kk ? ab1 : ab0) + frag_off(i);
      else {
        const int p = pl[i] + tapoff;
        return halo + (p << 7) + ((((kk << 2) + kq) ^ (p & 7)) << 4);
      }
    };
    if constexpr (SPLIT) {
      // 8 fp32 of this lane's K group (pieces q and q + 4 of the row: the K order the hi / lo weight planes are packed in)
      uint4 ar[MT][2];
      bf16x8_v bh[NT], bl[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        ar[i][0] = *(const uint4*)a_addr(i, 0);
        ar[i][1] = *(const uint4*)a_addr(i, 1);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        bh[j] = *(const bf16x8_v*)(Bs + j * 2048 + bfo0);
        bl[j] = *(const bf16x8_v*)(Bs + j * 2048 + bfo1);
      }
      __builtin_amdgcn_sched_barrier(0);
      bf16x8_v ah[MT], al[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        union { bf16x8_v v; uint32_t u[4]; } H, Lo;
        split_pair(__uint_as_float(ar[i][0].x), __uint_as_float(ar[i][0].y), H.u[0], Lo.u[0]);
        split_pair(__uint_as_float(ar[i][0].z), __uint_as_float(ar[i][0].w), H.u[1], Lo.u[1]);
        split_pair(__uint_as_float(ar[i][1].x), __uint_as_float(ar[i][1].y), H.u[2], Lo.u[2]);
        split_pair(__uint_as_float(ar[i][1].z), __uint_as_float(ar[i][1].w), H.u[3], Lo.u[3]);
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
    } else {
    // all fragment reads of the K step are issued before its first MFMA (the second half's LDS latency hides behind
    // the first half's MFMAs; hipcc otherwise reads, waits, multiplies, reads, waits, multiplies)
    bf16x8_v af[2][MT], bfr[2][NT];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int i = 0; i < MT; ++i) af[kk][i] = *(const bf16x8_v*)a_addr(i, kk);
#pragma unroll
      for (int j = 0; j < NT; ++j) bfr[kk][j] = *(const bf16x8_v*)(Bs + j * 2048 + (kk ? bfo1 : bfo0));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) mfma_bf16_acc_t(acc[i][j], af[kk][i], bfr[kk][j]);
    }
  };

  // ---- pipeline ----------------------------------------------------------------------------------------------------------
  // for each group of taps with equal temporal offset (<= 9: the 3x3 footprint), for each 64-channel chunk: stage the
  // halo image and run the group's taps, fully unrolled -- the tap rows live in scalar registers (loaded once per group),
  // ring slots and vmcnt counts are compile-time.  The weight tiles of the next BSLOTS-1 taps are in flight across
  // each step's barrier; the last steps of a block issue zero-page DMAs so that the counts stay exact.
  constexpr int GT = 9;
#ifdef VINET_CONV_TIMING
  const unsigned long long tm1 = __builtin_amdgcn_s_memtime();
#endif
  int t0 = 0;
  while (t0 < a.ntaps) {
    // tap words of the group, 16 bits each, word j at bit 16 j of a 144-bit scalar queue (q0, q1, q2):
    //   [7:0] halo offset + 128   [9:8] dx + 1 (0 in temporal mode)   [15:10] weight slice (< 64: vinet_conv_use_ht)
    // Every step takes word 0 (its own tap) and the slice of word BSLOTS - 1 (the weight tile it sends for), then the queue shifts
    // down by one word -- zeros come in at the top: slice 0, a real tap, for the steps that prefetch past the group's end.
    unsigned long long tq0 = 0, tq1 = 0;
    uint32_t tq2 = 0;
    const int dt = load_tap(a.taps, t0).x;
    int nt = 0;
#pragma unroll
    for (int j = 0; j < GT; ++j) {
      const bool in = (t0 + j < a.ntaps) && (nt == j);
      const int4 tp = load_tap(a.taps, t0 + j < a.ntaps ? t0 + j : t0);
      const bool same = in && (TM || tp.x == dt);           // temporal mode: every tap is (dt, 0, 0): one group
      const int off = TM ? tp.x * HW : tp.y * HW + tp.z;     // halo offset of the tap: rows are frames there
      const unsigned long long e = (unsigned long long)(((tp.w & 0x3f) << 10) | ((TM ? 0 : (tp.z + 1) & 3) << 8) | ((off + 128) & 0xff));
      if (j < 4) tq0 |= e << (j * 16);
      else if (j < 8) tq1 |= e << ((j - 4) * 16);
      else tq2 = (uint32_t)e;
      nt += same ? 1 : 0;
    }
    static_assert(BSLOTS - 1 <= 3, "the prefetched word must sit in the queue's first 64 bits");
    for (int c0 = 0; c0 < a.Kp; c0 += KC) {
#ifdef VINET_CONV_TIMING
      tm_h0 = __builtin_amdgcn_s_memtime();
#endif
      // last half chunk of the bf16 form (Kp % 64 == 32): this lane's chunks 4..7 lie past the row: 64 bytes back (see wp)
      const long kadj = (!SPLIT && c0 + KC > a.Kp && src_chunk >= 4) ? -64 : 0;
#pragma unroll
      for (int j = 0; j < BL; ++j) wp[j] += kadj;
      const long cbytes = (long)c0 * (SPLIT ? 4 : 2);
      unsigned long long q0 = tq0, q1 = tq1;
      uint32_t q2 = tq2;
      __builtin_amdgcn_s_barrier();                // everyone has finished reading the old halo image and ring
      asm volatile("" ::: "memory");
      issue_halo(dt, c0);
#pragma unroll
      for (int j = 0; j < BSLOTS - 1; ++j) issue_b(j, (long)((uint32_t)(q0 >> (16 * j + 10)) & 0x3fu) * slice_bytes + cbytes);
      int slot = 0, fill = BSLOTS - 1;
      for (int j = 0; j < nt; ++j) {
        wait_vmcnt<BL*(BSLOTS - 2)>();             // halo image (first step) and my weight DMAs of this step have landed
        if constexpr (PRE) {
          if (j == 0) xform_halo(c0);
        }
#ifdef VINET_CONV_TIMING
        if (j == 0) tm_halo += __builtin_amdgcn_s_memtime() - tm_h0;
#endif
        __builtin_amdgcn_s_barrier();              // everyone's have; everyone finished reading slot `fill`
        asm volatile("" ::: "memory");
        const uint32_t w0 = (uint32_t)q0;
        issue_b(fill, (long)((uint32_t)(q0 >> (16 * (BSLOTS - 1) + 10)) & 0x3fu) * slice_bytes + cbytes);
        compute(slot, (int)((w0 & 0xffu) << 7), (int)((w0 >> 4) & 0x30u));
        asm volatile("" ::: "memory");
        q0 = (q0 >> 16) | (q1 << 48);
        q1 = (q1 >> 16) | ((unsigned long long)q2 << 48);
        q2 = 0;
        slot = slot + 1 == BSLOTS ? 0 : slot + 1;
        fill = fill + 1 == BSLOTS ? 0 : fill + 1;
      }
#pragma unroll
      for (int j = 0; j < BL; ++j) wp[j] -= kadj;
    }
    t0 += nt;
  }
  wait_vmcnt<0>();
  mfma_drain();
  __syncthreads();
#ifdef VINET_CONV_TIMING
  const unsigned long long tm2 = __builtin_amdgcn_s_memtime();
  const float* dbg_ptr = a.out_shift;
  ConvArgs a2 = a;
  a2.out_shift = nullptr; a2.out_scale = nullptr;
  conv_epilogue<MT, NT, 4, 1, BNB>(a2, acc, smem, (int)sp, tile_n, er);
  if (tid == 0 && dbg_ptr) {   // tuning build only: out_shift doubles as a [grid][4] float dump
    const unsigned long long tm3 = __builtin_amdgcn_s_memtime();
    float* dbg = (float*)dbg_ptr + (long)blockIdx.x * 4;
    dbg[0] = (float)(tm1 - tm0); dbg[1] = (float)(tm2 - tm1); dbg[2] = (float)(tm3 - tm2); dbg[3] = (float)tm_halo;
  }
#else
  conv_epilogue<MT, NT, 4, 1, BNB>(a, acc, smem, (int)sp, tile_n, er);
#endif
}

template <int NT, int TW, int BSLOTS, bool TM = false, bool PRE = false, bool SPLIT = false, bool BNB = false>
static int launch_conv_ht_cfg(const ConvArgs& a, hipStream_t s) {
  using Cfg = ConvHtCfg<NT, TW, BSLOTS, TM, PRE, SPLIT>;
  auto kern = conv_ht_kernel<NT, TW, BSLOTS, TM, PRE, SPLIT, BNB>;
  static bool attr_done[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_done[dev & 63]) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::smem_bytes(1024));
    if (e != hipSuccess) { vinet_set_error("hipFuncSetAttribute(conv_ht): %s", hipGetErrorString(e)); return (int)e; }
    attr_done[dev & 63] = true;
  }
  const long Bn = a.M / ((long)a.To * a.Ho * a.Wo);
  const long grid = (long)a.tilesN * a.ht_tilesW * a.ht_tilesH * (TM ? 1 : a.To) * Bn;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), Cfg::smem_bytes(a.Kp), s, a);
  return vn_launch_status("conv_ht");
}
