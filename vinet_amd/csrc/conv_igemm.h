// Implicit-GEMM 3-D convolution on MFMA for gfx950 (channels-last activations).
//
//   M = B*oT*oH*oW output voxels, N = output channels, K = ntaps * Kp
//   workgroup = 256 threads = 4 waves, tile BM x BN, K step 32
//   A (activations): gathered per tap with bounds checks, optional consumer-side
//      BN+ReLU applied in registers, staged global -> VGPR -> LDS (double buffer)
//   B (weights): packed [slice][N][Kp], staged the same way
//   bf16: v_mfma_f32_16x16x32_bf16, fragments by ds_read_b128 from a padded
//      (80-byte row) LDS image;  fp32: v_mfma_f32_16x16x4_f32 (exact fp32).
//   epilogue: per-channel affine, BN partial statistics (sum, sum^2) per M tile,
//      ReLU / sigmoid, LDS-transposed so each lane stores 4 consecutive
//      channels, optional accumulate, arbitrary output placement.
//
// The same kernel serves forward convs, every dgrad (the caller passes the
// transposed weight pack and per-phase tap tables) and SoundNet's 1-D convs.
#pragma once
#include "common.h"

struct ConvArgs {
  const char* x;
  char* y;
  const char* w;
  const int4* taps;
  const float* in_scale;
  const float* in_shift;
  const float* out_scale;
  const float* out_shift;
  float* stats;
  int Ti, Hi, Wi, Cin, ldx;
  long sBx;
  int To, Ho, Wo;
  int sT, sH, sW;
  int yT, yH, yW, N, ldy;
  long sBy;
  int omT, omH, omW, ooT, ooH, ooW;
  int ntaps, Kp, M, tilesM, tilesN, Nw;
  int in_relu, act, accumulate, out_f32, vec_ok;
  int epi_rows;          // bf16 fast path of conv_epilogue: whole-row stores through a wave-private LDS image (0 = 16-byte stores per lane)
  int y_linear;          // output offset of voxel m is simply m*ldy (full-extent, possibly channel-sliced view)
  FastDiv dW, dH, dT;    // fast division by Wo, Ho, To
  // M-tile order (BM = 256 kernels, Ho*Wo % 256 == 0): logical tile i -> (b, spatial chunk c, t) with t
  // FASTEST, i.e. memory tile (b*To + t)*P + c.  Consecutive workgroups then read the same (h,w) window of
  // neighbouring frames: the temporal taps of a k x 1 x 1 conv hit in L2 instead of re-fetching every
  // input plane once per tap (measured on the 7x1x1 stem conv: 10.3 GB of L2-miss traffic for 4.2 GB of
  // tensors).  perm_P == 0: identity.
  int perm_P, perm_T;
  FastDiv dPT, dPermT;
  // split-K (conv_dma_kernel, no statistics): blockIdx.y owns K chunks [y*per, (y+1)*per) and stores its raw
  // accumulators to slab y of the fp32 workspace; conv_splitk_finish_kernel adds the slabs in a fixed order
  // (run-to-run deterministic) and applies the epilogue
  int splits, chunks_per_split;
  float* ws;
  // spatial halo tiles (conv_ht.h): tiles per image in H and W, workgroup index -> (column tile, spatial tile) decode
  int ht_tilesH, ht_tilesW;
  FastDiv ht_dN, ht_dW, ht_dH, ht_dTo;
  // BatchNorm-backward partial sums out of a data gradient's epilogue (VinetConvDesc::bnb_*, round 5): y is the gradient g behind
  // relu(bnb_scale * z + bnb_shift) (z = bnb_z: same extent as y, own row / clip strides); the launch also writes the partial
  // sums of vinet_bn_bwd_reduce(g, z) -- (sum g * gate, sum g * gate * (z - mean) * invstd) per channel -- one row per
  // statistics row of the kernel ([rows][2][N], like `stats`), formed on the ROUNDED values the launch stores (after the
  // accumulation when accumulate is set: the launch must then be the LAST writer of y).
  const char* bnb_z;
  int bnb_ldz, bnb_z_linear, bnb_relu;
  long bnb_sBz;
  const float *bnb_scale, *bnb_shift, *bnb_mean, *bnb_invstd;
  float* bnb_partials;
};

// logical M-tile index -> tile position in memory order
VN_DEV int conv_tile_perm(const ConvArgs& a, int i) {
  if (a.perm_P == 0) return i;
  const uint32_t b = fdiv((uint32_t)i, a.dPT);
  const uint32_t rem = (uint32_t)i - b * (uint32_t)(a.perm_P * a.perm_T);
  const uint32_t c = fdiv(rem, a.dPermT);
  const uint32_t t = rem - c * (uint32_t)a.perm_T;
  return (int)((b * (uint32_t)a.perm_T + t) * (uint32_t)a.perm_P + c);
}

// LDS the epilogue needs behind a K loop (every kernel sizes its dynamic LDS with max(K loop, this))
// per wave: the general path's fp32 row group, or the bf16 fast path's output image (MT x 16 rows x WNC bf16, conv_epilogue)
// with its row-offset table
template <int MT, int NT>
constexpr int conv_epi_wave_bytes() {
  constexpr int WNC = NT * 16;
  // (tiles of more than four row groups per wave -- conv_pp.h -- keep the 16-byte stores: their image would not fit)
#ifdef VINET_EXPERIMENTS
  constexpr int general = 16 * (WNC + 4) * 4, image = MT <= 4 ? MT * 16 * WNC * 2 + MT * 16 * 8 : 0;
#else
  // (the row image belongs to the `epi_rows` option, a side-build experiment: the shipped kernels must not pay its LDS --
  //  with it the 256 x 128 conv_dma tile needed 84 KB and ran ONE workgroup per CU)
  constexpr int general = 16 * (WNC + 4) * 4, image = 0;
#endif
  return general > image ? general : image;
}
// partial rows per wave row in the workgroup's statistics table: 4 for the forward statistics (lanes p = 0..3 of a row after two
// DPP rotations), 8 for the BatchNorm-backward sums, which are formed behind the permlane swap, where the two halves q & 1 of a
// wave hold the same channels of different row groups
constexpr int CONV_EPI_RED_ROWS = 8;
template <int MT, int NT, int WARPS_M, int WARPS_N>
constexpr int conv_epi_bytes() {
  constexpr int W = WARPS_M * WARPS_N, WNC = NT * 16, BN = WNC * WARPS_N;
  // per-wave staging + statistics table: up to CONV_EPI_RED_ROWS partial rows per wave row
  return W * conv_epi_wave_bytes<MT, NT>() + CONV_EPI_RED_ROWS * WARPS_M * BN * 2 * 4;
}

// SPLIT (T = float only; VINET_F32S): fp32 tensors in memory, bf16 matrix arithmetic on a two-term split of every operand --
// x = hi + lo with hi = bf16(x) (round to nearest even) and lo = bf16(x - hi), 16 significant bits -- and THREE MFMAs per product
// (hi*hi + hi*lo + lo*hi, fp32 accumulate; the lo*lo term is below 2^-16 of the product): 3/16 of the fp32-MFMA cost for an error of
// ~2^-17 per operand instead of bf16's 2^-9.  The split happens once per loaded element, on the way into LDS, which then holds a hi
// and a lo bf16 image of each tile (64-byte rows, 16-byte chunk index XOR (row >> 2) & 3: conflict-free ds_read_b128 fragments).
template <typename T, int MT, int NT, int WARPS_M, int WARPS_N, bool SPLIT = false>
struct ConvCfg {
  static_assert(!SPLIT || sizeof(T) == 4, "the split form reads fp32 tensors");
  static constexpr int BM = 16 * MT * WARPS_M;
  static constexpr int BN = 16 * NT * WARPS_N;
  static constexpr int BK = 32;
  static constexpr int EG = ElemTraits<T>::EG;
  static constexpr int G = BK / EG;                     // 16-byte groups per K row
  static constexpr int RS = SPLIT ? 64 : BK * (int)sizeof(T) + 16;   // LDS row stride (bytes): padded, or swizzled bf16 rows
  static constexpr int A_BYTES = (SPLIT ? 2 : 1) * BM * RS;
  static constexpr int B_BYTES = (SPLIT ? 2 : 1) * BN * RS;
  static constexpr int WNC = NT * 16;                   // columns per wave
  static constexpr int EROW = WNC + 4;                  // epilogue LDS row stride (floats)
  static constexpr int KLOOP_BYTES = 2 * (A_BYTES + B_BYTES);
  static constexpr int EPI_BYTES = conv_epi_bytes<MT, NT, WARPS_M, WARPS_N>();
  static constexpr int SMEM = KLOOP_BYTES > EPI_BYTES ? KLOOP_BYTES : EPI_BYTES;
};

// ---- shared epilogue: per-channel affine, BN partial statistics, activation,
//      LDS-transposed 4-channel vector stores, accumulate, arbitrary placement.
//      Must be entered after a workgroup barrier (it reuses the K-loop LDS).
//      The staging area is PRIVATE to each wave, so the write->read hand-off needs only
//      wave-level ordering.  (A __syncthreads() here would also wait for the wave's own
//      global stores -- vmcnt counts stores on gfx950 -- and cost a full write round trip
//      per row group: 3.9k cycles each, measured with s_memtime.) ------------------------
VN_DEV void wave_lds_fence() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// Row -> voxel map of a wave's 16-row groups.  Row-tiled kernels: group i starts at voxel m0 + 16 i (ipr == 0).  Spatial
// halo tiles (conv_ht.h): a wave owns `ipr` groups per image row, group i starts at m0 + (i / ipr) * rstride + (i % ipr) * 16
// and is inside the iteration space iff i / ipr < nrows and i % ipr < ncols.
struct EpiRows {
  int m0, ipr, rstride, nrows, ncols;
};

// ---- epilogue, second generation (round 3) ---------------------------------------------------------------------------
// The K loops issue their MFMAs with the operands SWAPPED (weights as the A operand), so the accumulators arrive
// transposed:   acc[i][j][r] = D[ m = 16 i + (lane & 15) ][ n = 16 j + 4 (lane >> 4) + r ]
// i.e. a lane holds 4 CONSECUTIVE CHANNELS of one voxel per 16x16 tile instead of 4 voxels of one channel.  What that buys
// (the epilogue was VALU-issue bound: 20k cycles per 256 x 96 tile, as long as the nine K steps of a 64-channel 3x3 conv):
//   * no LDS transposition at all on the bf16 path: two v_cvt_pk_bf16_f32 pack the lane's 4 channels of a tile, one
//     v_permlane16_swap_b32 per dword between the tiles of TWO row groups hands every lane 8 consecutive channels of one
//     voxel (rows 0 / 2 of the wave keep group 2k, rows 1 / 3 take group 2k + 1), one 16-byte store per lane and tile pair --
//     instead of 8 two-byte LDS writes with their address arithmetic, two wave-level fences, a ds_read_b128 and the store;
//   * ReLU is one v_pk_max_i16 per packed pair (as signed 16-bit integers every negative bf16 is < 0);
//   * per-channel constants are 4 registers per column tile (float4 loads) instead of one per tile and lane;
//   * BN partial sums: a lane owns its 4 channels' (sum, sum^2) over the row groups, then four DPP row_ror adds reduce the 16
//     lanes of a row (one voxel each) -- the old layout needed a cross-row shuffle pair per column tile.
// The accumulate form (data gradients joining an existing gradient) swaps the fp32 values instead (one rounding of
// old + new, as before); the general path (fp32 output, sigmoid head, channel counts that are no multiple of 8) still goes
// through a wave-private LDS tile, written 4 channels (one ds_write_b128) at a time.
// odd rows of x <-> even rows of y (rows = 16 lanes).  The operands come straight out of inline-asm VALU instructions, which
// the compiler's hazard recogniser cannot see: gfx950 wants 2 wait states between a VALU write and a permlane swap reading it.
VN_DEV void permlane16_swap(uint32_t& x, uint32_t& y) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
}
VN_DEV void permlane16_swap_f(float& x, float& y) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
}
// sum over the 16 lanes of a row (every lane of the row receives the total)
VN_DEV float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));   // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));   // row_ror:1
  return v;
}
// the same after two rotations only: lane p holds the sum over lanes {p, p + 4, p + 8, p + 12} of its row
VN_DEV float row16_sum4(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
  return v;
}
VN_DEV uint32_t pk_relu_bf16(uint32_t u) {
  uint32_t r;
  asm("v_pk_max_i16 %0, %1, 0" : "=v"(r) : "v"(u));
  return r;
}

// `rows` (optional): see EpiRows; default = rows tile_m*BM + ... as usual.  `tile_m` stays the statistics row of the workgroup.
// BNB: the instantiation that can also form the BatchNorm-backward partial sums (ConvArgs::bnb_*).  A template parameter, not a
// run-time branch: the sums need ~50 more live registers in the store loop, and as a run-time path they raised the register
// allocation of EVERY kernel that shares this epilogue (conv_dma<4,4,2,2>: 64 -> 127 VGPRs, 4 -> 2 waves per SIMD).  The BNB
// kernels are instantiated in their own translation unit (conv_bnb.hip) for plain-input conv_dma / conv_ht shapes only.
template <int MT, int NT, int WARPS_M, int WARPS_N, bool BNB = false>
VN_DEV void conv_epilogue(const ConvArgs& a, f32x4_v (&acc)[MT][NT], char* smem, int tile_m, int tile_n, const EpiRows rows = EpiRows{0, -1, 0, 0, 0}) {
  constexpr int BM = 16 * MT * WARPS_M, BN = 16 * NT * WARPS_N;
  static_assert(MT % 2 == 0, "row groups are handled in pairs");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WARPS_N, wn = wave % WARPS_N;
  constexpr int WNC = NT * 16, EROW = WNC + 4;
  const int p = lane & 15, q = lane >> 4;              // my voxel inside a row group, my channel quad inside a column tile
  float* Ew = (float*)(smem + wave * conv_epi_wave_bytes<MT, NT>());
  float* red = (float*)(smem + conv_epi_bytes<MT, NT, WARPS_M, WARPS_N>() - CONV_EPI_RED_ROWS * WARPS_M * BN * 2 * 4);
  const int m_wave = tile_m * BM + wm * MT * 16;
  const int n_wave = tile_n * BN + wn * WNC;
  const bool do_stats = a.stats != nullptr;
  // (tiles of more than four row groups per wave -- conv_pp.h -- do not carry the BatchNorm-backward sums: register budget)
  const bool do_bnb = BNB && MT <= 4 && a.bnb_partials != nullptr;
  const bool relu = a.act == VINET_ACT_RELU;
  const float relu_floor = relu ? 0.f : -INFINITY;   // branch-free ReLU: max(v, floor)
  const EpiRows er = rows.ipr >= 0 ? rows : EpiRows{m_wave, 0, 0, 0, 0};      // (ipr == -1: the default row-tiled map)
  // first voxel of row group i, and how many of its 16 rows lie inside the iteration space
  auto group_m0 = [&](int i) { return er.ipr ? er.m0 + (i / er.ipr) * er.rstride + (i % er.ipr) * 16 : er.m0 + i * 16; };
  auto group_rows = [&](int i) { return er.ipr ? (((i / er.ipr) < er.nrows && (i % er.ipr) < er.ncols) ? 16 : 0) : a.M - (er.m0 + i * 16); };
  // element offset of voxel m (channel 0) in y: any placement (a stride phase of a data gradient scatters its rows)
  auto voxel_off = [&](int m) {
    if (a.y_linear) return (long)m * a.ldy;
    int b, to, ho, wo;
    decode_m(m, a.dW, a.dH, a.dT, b, to, ho, wo);
    return (long)b * a.sBy + ((long)((to * a.omT + a.ooT) * a.yH + (ho * a.omH + a.ooH)) * a.yW + (wo * a.omW + a.ooW)) * (long)a.ldy;
  };
  const bool sigm = a.act == VINET_ACT_SIGMOID;
  const bool has_aff = a.out_scale != nullptr || a.out_shift != nullptr;
  // per-channel constants of my quad in column tile j
  auto quad_consts = [&](int j, float (&sc)[4], float (&sh)[4], bool (&cok)[4]) {
    const int n0 = n_wave + j * 16 + q * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      cok[r] = n0 + r < a.Nw;
      sc[r] = (a.out_scale && cok[r]) ? a.out_scale[n0 + r] : 1.f;
      sh[r] = (a.out_shift && cok[r]) ? a.out_shift[n0 + r] : 0.f;
    }
  };
  // my row's (voxel's) share of the statistics of column tile j -> the workgroup's reduction table
  // (two DPP rotations leave lanes p = 0..3 of a row with four distinct partial sums: 4 partial rows per wave in the table,
  //  added up by the last phase -- half the DPP adds of a full row reduction)
  auto put_stats = [&](int j, float (&ss)[4], float (&qq)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { ss[r] = row16_sum4(ss[r]); qq[r] = row16_sum4(qq[r]); }
    if (p < 4) {
      const int col = wn * WNC + j * 16 + q * 4;
      float* dst = red + ((long)(wm * 4 + p) * BN + col) * 2;
      *(float4*)dst = make_float4(ss[0], qq[0], ss[1], qq[1]);
      *(float4*)(dst + 4) = make_float4(ss[2], qq[2], ss[3], qq[3]);
    }
  };
  bool rok[MT];                                        // is my voxel of row group i inside the iteration space
#pragma unroll
  for (int i = 0; i < MT; ++i) rok[i] = p < group_rows(i);

  // every row group of this wave and every column of its tile inside the iteration space (wave-uniform)?
  bool all_in = n_wave + WNC <= a.Nw;
#pragma unroll
  for (int i = 0; i < MT; ++i) all_in = all_in && group_rows(i) >= 16;
  const bool fast8 = a.vec_ok && !a.out_f32 && !sigm && (a.N & 7) == 0 && (a.ldy & 7) == 0 && (a.sBy & 7) == 0 && (((uintptr_t)a.y) & 15) == 0;
  if (fast8) {
    // after the swap my 16 bytes are channels n8 .. n8 + 7 of voxel p in row group 2k + (q & 1)
    constexpr int NP = MT / 2;
    // Plain stores (no accumulate) leave as WHOLE ROWS: the 16-byte pieces go into a wave-private LDS image (MT x 16 rows x WNC
    // bf16, piece index XOR img_swz(row): 16 rows x one piece and the pieces of a row are both conflict-free), then WNC / 8
    // consecutive lanes write the WNC * 2 contiguous bytes of a voxel.  Stored directly, a lane's 16 bytes sit next to only one
    // other lane's: 32-byte fragments, four instructions per 128-byte line -- the pointwise kernel ran at half the speed of its
    // own loads + MFMAs that way (conv_pw.h, profiles/r3_pw_ab.txt).  Row offsets: one voxel decode per lane, through LDS.
#ifdef VINET_EXPERIMENTS
    const bool staged = MT <= 4 && a.epi_rows && !a.accumulate;
#else
    constexpr bool staged = false;      // (option epi_rows: measured neutral to slower, profiles/r3_epi_rows_ab.txt -- side builds only)
#endif
    constexpr int PPR = WNC / 8, IMG_G = (PPR % 16 == 0) ? 16 : (PPR % 8 == 0) ? 8 : (PPR % 4 == 0) ? 4 : 2, IMG_P = 16 / IMG_G;
    auto img_swz = [](int row) { return (row / IMG_P) & (IMG_G - 1); };
    char* const img = (char*)Ew;
    long* const rowoff = (long*)(img + MT * 16 * WNC * 2);
    long voff[NP], zoff[NP];
    bool vok[NP];
    // element offset of voxel m in z (bnb: y is placed densely over its whole extent, so voxel m of the iteration space is
    // voxel m of z's extent as well)
    auto z_off = [&](int m) {
      if (a.bnb_z_linear) return (long)m * a.bnb_ldz;
      int b, to, ho, wo;
      decode_m(m, a.dW, a.dH, a.dT, b, to, ho, wo);
      return (long)b * a.bnb_sBz + ((long)(to * a.yH + ho) * a.yW + wo) * (long)a.bnb_ldz;
    };
    if (staged) {
#pragma unroll
      for (int r0 = 0; r0 < MT * 16; r0 += 64) {
        const int ig = (r0 >> 4) + q;           // row r0 + lane: group (r0 + lane) >> 4, voxel p
        if (ig < MT) rowoff[r0 + lane] = p < group_rows(ig) ? voxel_off(group_m0(ig) + p) : -1;
      }
#pragma unroll
      for (int k = 0; k < NP; ++k) { vok[k] = false; voff[k] = 0; }
    } else {
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const int ig = 2 * k + (q & 1);
        vok[k] = p < group_rows(ig);
        voff[k] = vok[k] ? voxel_off(group_m0(ig) + p) : 0;
        if constexpr (BNB) zoff[k] = (do_bnb && vok[k]) ? z_off(group_m0(ig) + p) : 0;
      }
    }
    uint4 zl[NP];
    if constexpr (BNB) if (do_bnb) {
      // the wave's per-channel constants -> its private LDS area (unused on this path): [scale | shift | mean | invstd][WNC]
      for (int c = lane; c < WNC; c += 64) {
        const int n = n_wave + c;
        const bool ok = n < a.N;
        Ew[c] = (ok && a.bnb_scale) ? a.bnb_scale[n] : 1.f;
        Ew[WNC + c] = (ok && a.bnb_shift) ? a.bnb_shift[n] : 0.f;
        Ew[2 * WNC + c] = ok ? a.bnb_mean[n] : 0.f;
        Ew[3 * WNC + c] = ok ? a.bnb_invstd[n] : 0.f;
      }
      const int n80 = n_wave + (q >> 1) * 8;
#pragma unroll
      for (int k = 0; k < NP; ++k)
        zl[k] = (vok[k] && n80 < a.N) ? *(const uint4*)((const bf16_t*)a.bnb_z + zoff[k] + n80) : make_uint4(0, 0, 0, 0);
      wave_lds_fence();
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float sc[4], sh[4], ss[4] = {0.f, 0.f, 0.f, 0.f}, qq[4] = {0.f, 0.f, 0.f, 0.f};
      bool cok[4];
      quad_consts(j, sc, sh, cok);
      const int n8 = n_wave + j * 16 + (q >> 1) * 8;
      const bool nok8 = n8 < a.N;
      uint4 old[NP];
      if (a.accumulate) {
#pragma unroll
        for (int k = 0; k < NP; ++k)
          old[k] = (vok[k] && nok8) ? *(const uint4*)((const bf16_t*)a.y + voff[k] + n8) : make_uint4(0, 0, 0, 0);
      }
      // BatchNorm-backward sums of my 8 channels n8 .. n8 + 7 over my voxels (one per row-group pair).  z of column tile j + 1
      // is requested before tile j is processed (the loads of a tile would otherwise be consumed one memory latency after their
      // issue, NT times per wave tile); the per-channel constants come from the wave's LDS table.
      uint4 zn[NP];
      f32x2_v bsc[4], bsh[4], bmu[4], bs[4], bp[4];
      if constexpr (BNB) if (do_bnb) {
        if (j + 1 < NT) {
          const int n8n = n8 + 16;
#pragma unroll
          for (int k = 0; k < NP; ++k)
            zn[k] = (vok[k] && n8n < a.N) ? *(const uint4*)((const bf16_t*)a.bnb_z + zoff[k] + n8n) : make_uint4(0, 0, 0, 0);
        }
        const int c8 = j * 16 + (q >> 1) * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float4 t0 = *(const float4*)&Ew[c8 + 4 * h], t1 = *(const float4*)&Ew[WNC + c8 + 4 * h], t2 = *(const float4*)&Ew[2 * WNC + c8 + 4 * h];
          bsc[2 * h] = (f32x2_v){t0.x, t0.y}; bsc[2 * h + 1] = (f32x2_v){t0.z, t0.w};
          bsh[2 * h] = (f32x2_v){t1.x, t1.y}; bsh[2 * h + 1] = (f32x2_v){t1.z, t1.w};
          bmu[2 * h] = (f32x2_v){t2.x, t2.y}; bmu[2 * h + 1] = (f32x2_v){t2.z, t2.w};
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) { bs[h] = (f32x2_v){0.f, 0.f}; bp[h] = (f32x2_v){0.f, 0.f}; }
      }
      // g = the 8 bf16 this lane stores for pair k (rounded, accumulated): gate with the forward ReLU, add to the sums.
      // Lanes outside the iteration space contribute nothing.
      auto bnb_acc = [&](int k, uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3) {
        if (!(vok[k] && nok8)) return;
        const uint32_t ow[4] = {o0, o1, o2, o3}, zw[4] = {zl[k].x, zl[k].y, zl[k].z, zl[k].w};
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          f32x2_v g = {__uint_as_float(ow[h] << 16), __uint_as_float(ow[h] & 0xffff0000u)};
          const f32x2_v z = {__uint_as_float(zw[h] << 16), __uint_as_float(zw[h] & 0xffff0000u)};
          if (a.bnb_relu) {
            const f32x2_v t = {fmaf(z.x, bsc[h].x, bsh[h].x), fmaf(z.y, bsc[h].y, bsh[h].y)};
            g.x = t.x > 0.f ? g.x : 0.f;
            g.y = t.y > 0.f ? g.y : 0.f;
          }
          const f32x2_v dz_ = z - bmu[h];
          bs[h] += g;
          bp[h].x = fmaf(g.x, dz_.x, bp[h].x);
          bp[h].y = fmaf(g.y, dz_.y, bp[h].y);
        }
      };
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        float v0[4], v1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {      // explicit reads: the accumulator stays in the AGPR file until this very use
          asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v0[r]) : "a"(acc[2 * k][j][r]));
          asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v1[r]) : "a"(acc[2 * k + 1][j][r]));
        }
        if (has_aff) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { v0[r] = fmaf(v0[r], sc[r], sh[r]); v1[r] = fmaf(v1[r], sc[r], sh[r]); }
        }
        if (do_stats) {
          if (all_in) {      // (wave-uniform: no masks)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              ss[r] += v0[r]; qq[r] = fmaf(v0[r], v0[r], qq[r]);
              ss[r] += v1[r]; qq[r] = fmaf(v1[r], v1[r], qq[r]);
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float t0 = (rok[2 * k] && cok[r]) ? v0[r] : 0.f, t1 = (rok[2 * k + 1] && cok[r]) ? v1[r] : 0.f;
              ss[r] += t0; qq[r] = fmaf(t0, t0, qq[r]);
              ss[r] += t1; qq[r] = fmaf(t1, t1, qq[r]);
            }
          }
        }
        if (!a.accumulate) {
          uint32_t x0 = cvt_pk_bf16_f32(v0[0], v0[1]), x1 = cvt_pk_bf16_f32(v0[2], v0[3]);
          uint32_t y0 = cvt_pk_bf16_f32(v1[0], v1[1]), y1 = cvt_pk_bf16_f32(v1[2], v1[3]);
          if (relu) { x0 = pk_relu_bf16(x0); x1 = pk_relu_bf16(x1); y0 = pk_relu_bf16(y0); y1 = pk_relu_bf16(y1); }
          permlane16_swap(x0, y0);
          permlane16_swap(x1, y1);
          if (staged) {
            const int row = (2 * k + (q & 1)) * 16 + p;
            *(uint4*)(img + row * (WNC * 2) + (((2 * j + (q >> 1)) ^ img_swz(row)) * 16)) = make_uint4(x0, x1, y0, y1);
          } else if (vok[k] && nok8) {
            *(uint4*)((bf16_t*)a.y + voff[k] + n8) = make_uint4(x0, x1, y0, y1);
          }
          if constexpr (BNB) if (do_bnb) bnb_acc(k, x0, x1, y0, y1);
        } else {
          // y += result: ONE rounding, of old + new in fp32
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v0[r] = fmaxf(v0[r], relu_floor); v1[r] = fmaxf(v1[r], relu_floor);
            permlane16_swap_f(v0[r], v1[r]);
          }
          const uint32_t ow[4] = {old[k].x, old[k].y, old[k].z, old[k].w};
          uint32_t o[4];
          o[0] = cvt_pk_bf16_f32(v0[0] + __uint_as_float(ow[0] << 16), v0[1] + __uint_as_float(ow[0] & 0xffff0000u));
          o[1] = cvt_pk_bf16_f32(v0[2] + __uint_as_float(ow[1] << 16), v0[3] + __uint_as_float(ow[1] & 0xffff0000u));
          o[2] = cvt_pk_bf16_f32(v1[0] + __uint_as_float(ow[2] << 16), v1[1] + __uint_as_float(ow[2] & 0xffff0000u));
          o[3] = cvt_pk_bf16_f32(v1[2] + __uint_as_float(ow[3] << 16), v1[3] + __uint_as_float(ow[3] & 0xffff0000u));
          if (vok[k] && nok8) *(uint4*)((bf16_t*)a.y + voff[k] + n8) = make_uint4(o[0], o[1], o[2], o[3]);
          if constexpr (BNB) if (do_bnb) bnb_acc(k, o[0], o[1], o[2], o[3]);
        }
      }
      if (do_stats) put_stats(j, ss, qq);
      if constexpr (BNB) if (do_bnb) {
        // the 16 lanes of a row hold 16 voxels of the same 8 channels: two DPP rotations leave lanes p = 0..3 with four partial
        // sums; rows q and q ^ 1 hold the same channels of the other row group of each pair -> 8 partial rows per wave row
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          bs[h].x = row16_sum4(bs[h].x); bs[h].y = row16_sum4(bs[h].y);
          bp[h].x = row16_sum4(bp[h].x); bp[h].y = row16_sum4(bp[h].y);
        }
        if (p < 4 && nok8) {
          const int c8 = j * 16 + (q >> 1) * 8;
          float* dst = red + ((long)(wm * 8 + (q & 1) * 4 + p) * BN + wn * WNC + c8) * 2;
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            const float i0 = Ew[3 * WNC + c8 + 2 * h], i1 = Ew[3 * WNC + c8 + 2 * h + 1];
            *(float4*)(dst + 4 * h) = make_float4(bs[h].x, bp[h].x * i0, bs[h].y, bp[h].y * i1);
          }
        }
        if (j + 1 < NT) {
#pragma unroll
          for (int k = 0; k < NP; ++k) zl[k] = zn[k];
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // one column tile at a time: hoisted accumulator reads of later tiles spill
    }
    if (staged) {
      wave_lds_fence();
#pragma unroll
      for (int it = 0; it < (MT * 16 * PPR) / 64; ++it) {
        const int e = lane + 64 * it;
        const int row = e / PPR, piece = e - row * PPR;
        const long off = rowoff[row];
        const int n = n_wave + piece * 8;
        const uint4 v = *(const uint4*)(img + row * (WNC * 2) + ((piece ^ img_swz(row)) * 16));
        if (off >= 0 && n < a.N) *(uint4*)((bf16_t*)a.y + off + n) = v;
      }
    }
  } else {
    // ---- general path: fp32 output, sigmoid, odd channel counts; one row group at a time through a wave-private LDS tile
    // (statistics first, one column tile at a time: 8 running sums live instead of 8 per column tile beside the staging loop)
    if (do_stats) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        float sc[4], sh[4], ss[4] = {0.f, 0.f, 0.f, 0.f}, qq[4] = {0.f, 0.f, 0.f, 0.f};
        bool cok[4];
        quad_consts(j, sc, sh, cok);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float av;
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(av) : "a"(acc[i][j][r]));
            const float v = fmaf(av, sc[r], sh[r]);
            const float vs = (cok[r] && rok[i]) ? v : 0.f;
            ss[r] += vs; qq[r] += vs * vs;
          }
        put_stats(j, ss, qq);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const bool fast = a.vec_ok && !a.out_f32 && !a.accumulate && a.y_linear;
    constexpr int VPR = WNC / 4;             // 4-channel vectors per tile row
    constexpr int ITERS = (16 * VPR) / 64;   // store instructions per lane per row group
    static_assert((16 * VPR) % 64 == 0, "row group must divide over the wave");
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        float sc[4], sh[4], o[4];
        bool cok[4];
        quad_consts(j, sc, sh, cok);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float av;
          asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(av) : "a"(acc[i][j][r]));
          const float v = fmaf(av, sc[r], sh[r]);
          o[r] = fmaxf(v, relu_floor);
          if (sigm) o[r] = 1.f / (1.f + __expf(-o[r]));
        }
        *(float4*)&Ew[p * EROW + j * 16 + q * 4] = make_float4(o[0], o[1], o[2], o[3]);
      }
      wave_lds_fence();
      if (fast) {
#pragma unroll
        for (int k = 0; k < ITERS; ++k) {
          const int e = lane + 64 * k;
          const int rr = e / VPR, cc = (e % VPR) * 4;
          const int m = group_m0(i) + rr;
          const bool mok = rr < group_rows(i);
          const int n = n_wave + cc;
          const float4 v = *(const float4*)&Ew[rr * EROW + cc];
          if (mok && n < a.N)
            *(uint2*)((bf16_t*)a.y + (long)m * a.ldy + n) = make_uint2(pack2bf(v.x, v.y), pack2bf(v.z, v.w));
        }
      } else {
        for (int e = lane; e < 16 * VPR; e += 64) {
          const int rr = e / VPR, cc = (e % VPR) * 4;
          const int m = group_m0(i) + rr;
          const bool mok = rr < group_rows(i);
          const int n = n_wave + cc;
          if (mok && n < a.N) {
            const float4 v = *(const float4*)&Ew[rr * EROW + cc];
            const long off = voxel_off(m) + n;
            float o[4] = {v.x, v.y, v.z, v.w};
            if (a.vec_ok) {
              if (a.out_f32) {
                float* dst = (float*)a.y + off;
                if (a.accumulate) { const float4 q4 = *(const float4*)dst; o[0] += q4.x; o[1] += q4.y; o[2] += q4.z; o[3] += q4.w; }
                *(float4*)dst = make_float4(o[0], o[1], o[2], o[3]);
              } else {
                bf16_t* dst = (bf16_t*)a.y + off;
                if (a.accumulate) {
                  const uint2 q2 = *(const uint2*)dst;
                  o[0] += __uint_as_float(q2.x << 16); o[1] += __uint_as_float(q2.x & 0xffff0000u);
                  o[2] += __uint_as_float(q2.y << 16); o[3] += __uint_as_float(q2.y & 0xffff0000u);
                }
                *(uint2*)dst = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
              }
            } else {
#pragma unroll
              for (int e2 = 0; e2 < 4; ++e2) {
                if (n + e2 < a.N) {
                  if (a.out_f32) {
                    float* dst = (float*)a.y + off + e2;
                    *dst = a.accumulate ? *dst + o[e2] : o[e2];
                  } else {
                    bf16_t* dst = (bf16_t*)a.y + off + e2;
                    *dst = f2bf(a.accumulate ? bf2f(*dst) + o[e2] : o[e2]);
                  }
                }
              }
            }
          }
        }
      }
      wave_lds_fence();   // all lanes have read this row group before the next one overwrites it
    }
  }
  if (do_stats) {
    __syncthreads();    // cross-wave hand-off of the per-wave column sums
    if (tid < BN) {
      const int n = tile_n * BN + tid;
      if (n < a.N) {
        float ss = 0.f, qq = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < 4 * WARPS_M; ++w2) { ss += red[(w2 * BN + tid) * 2]; qq += red[(w2 * BN + tid) * 2 + 1]; }
        a.stats[((long)tile_m * 2 + 0) * a.N + n] = ss;
        a.stats[((long)tile_m * 2 + 1) * a.N + n] = qq;
      }
    }
  } else if (BNB && do_bnb) {
    __syncthreads();
    if (tid < BN) {
      const int n = tile_n * BN + tid;
      if (n < a.N) {
        float ss = 0.f, qq = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < 8 * WARPS_M; ++w2) { ss += red[(w2 * BN + tid) * 2]; qq += red[(w2 * BN + tid) * 2 + 1]; }
        a.bnb_partials[((long)tile_m * 2 + 0) * a.N + n] = ss;
        a.bnb_partials[((long)tile_m * 2 + 1) * a.N + n] = qq;
      }
    }
  }
}

// hi / lo bf16 pairs of two fp32 values: hi = RNE(x) packed, lo = RNE(x - hi) packed (6 VALU per pair)
VN_DEV void split_pair(float x0, float x1, uint32_t& hi, uint32_t& lo) {
  hi = cvt_pk_bf16_f32(x0, x1);
  lo = cvt_pk_bf16_f32(x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u));
}

template <typename T, int MT, int NT, int WARPS_M, int WARPS_N, int MODE, bool SPLIT = false>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a) {
  using Cfg = ConvCfg<T, MT, NT, WARPS_M, WARPS_N, SPLIT>;
  constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK, EG = Cfg::EG, G = Cfg::G, RS = Cfg::RS;
  constexpr int A_LOADS = (BM * G) / 256;
  constexpr int B_ITEMS = BN * G;
  constexpr int B_LOADS = (B_ITEMS + 255) / 256;
  static_assert((BM * G) % 256 == 0, "A tile must divide evenly over 256 threads");
  static_assert(WARPS_M * WARPS_N == 4, "4 waves");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WARPS_N, wn = wave % WARPS_N;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = wg % a.tilesN, tile_m = wg / a.tilesN;
  const int g = tid % G;

  // ---- per-thread A rows (fixed across the K loop) -------------------------
  long a_base[A_LOADS];
  int a_t[A_LOADS], a_h[A_LOADS], a_w[A_LOADS];
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) {
    const int row = (i * 256 + tid) / G;
    const int m = tile_m * BM + row;
    if (m < a.M) {
      int b, to, ho, wo;
      decode_m(m, a.dW, a.dH, a.dT, b, to, ho, wo);
      a_base[i] = (long)b * a.sBx;
      a_t[i] = to * a.sT; a_h[i] = ho * a.sH; a_w[i] = wo * a.sW;
    } else {
      a_base[i] = 0; a_t[i] = -(1 << 28); a_h[i] = 0; a_w[i] = 0;
    }
  }

  const int cpt = a.Kp / BK;          // K chunks per tap
  const int nchunks = a.ntaps * cpt;
  const bool has_pre = a.in_scale != nullptr;

  uint4 ra[A_LOADS], rb[B_LOADS];

  auto load_tiles = [&](int it) {
    const int tap = it / cpt;
    const int c0 = (it - tap * cpt) * BK;
    const int4 tp = load_tap(a.taps, tap);
    if constexpr (MODE == VINET_CONV_GENERIC) {
      const int c = c0 + g * EG;
      const bool cin_ok = c < a.Cin;
      float sc[EG], sh[EG];
      if (has_pre && cin_ok) {
#pragma unroll
        for (int e = 0; e < EG; ++e) { sc[e] = a.in_scale[c + e]; sh[e] = a.in_shift[c + e]; }
      }
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i) {
        const int ti = a_t[i] + tp.x, hi = a_h[i] + tp.y, wi = a_w[i] + tp.z;
        const bool ok = cin_ok && (unsigned)ti < (unsigned)a.Ti && (unsigned)hi < (unsigned)a.Hi &&
                        (unsigned)wi < (unsigned)a.Wi;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (ok) {
          const long off = a_base[i] + ((long)(ti * a.Hi + hi) * a.Wi + wi) * (long)a.ldx + c;
          v = *(const uint4*)(a.x + off * (long)sizeof(T));
          if (has_pre) {
            float f[EG];
            unpack16<T>(v, f);
#pragma unroll
            for (int e = 0; e < EG; ++e) {
              f[e] = fmaf(f[e], sc[e], sh[e]);
              if (a.in_relu) f[e] = fmaxf(f[e], 0.f);
            }
            v = pack16<T>(f);
          }
        }
        ra[i] = v;
      }
    } else {
      // stem: K chunk = 8 consecutive W positions x 4 channels; x has C == 4
      constexpr int PP = EG / 4;  // 4-channel pixels per 16-byte group
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i) {
        const int ti = a_t[i] + tp.x, hi = a_h[i] + tp.y;
        const bool row_ok = (unsigned)ti < (unsigned)a.Ti && (unsigned)hi < (unsigned)a.Hi;
        const long rowoff = a_base[i] + ((long)(ti * a.Hi + hi) * a.Wi) * (long)a.ldx;
        uint32_t words[4] = {0, 0, 0, 0};
#pragma unroll
        for (int p = 0; p < PP; ++p) {
          const int wi = a_w[i] + tp.z + g * PP + p;
          if (row_ok && (unsigned)wi < (unsigned)a.Wi) {
            const char* src = a.x + (rowoff + (long)wi * a.ldx) * (long)sizeof(T);
            if constexpr (sizeof(T) == 2) {
              const uint2 q = *(const uint2*)src;
              words[2 * p] = q.x; words[2 * p + 1] = q.y;
            } else {
              const uint4 q = *(const uint4*)src;
              words[0] = q.x; words[1] = q.y; words[2] = q.z; words[3] = q.w;
            }
          }
        }
        ra[i] = make_uint4(words[0], words[1], words[2], words[3]);
      }
    }
#pragma unroll
    for (int j = 0; j < B_LOADS; ++j) {
      const int idx = j * 256 + tid;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (idx < B_ITEMS) {
        const int n = idx / G, gg = idx % G;
        const int nn = tile_n * BN + n;
        if (nn < a.Nw) {
          const long off = ((long)tp.w * a.Nw + nn) * (long)a.Kp + c0 + gg * EG;
          v = *(const uint4*)(a.w + off * (long)sizeof(T));
        }
      }
      rb[j] = v;
    }
  };

  auto store_tiles = [&](int buf) {
    char* As = smem + buf * Cfg::A_BYTES;
    char* Bs = smem + 2 * Cfg::A_BYTES + buf * Cfg::B_BYTES;
    if constexpr (SPLIT) {
      // activations: four fp32 of K group gg (channels 4gg .. 4gg+3 of the chunk) -> 8 bytes of the hi image and 8 of the lo
      // image, at K positions 8 (gg & 3) + 4 (gg >> 2) + {0..3}: the channel permutation of the split weight packs
      // (layout.hip: pack_store<VINET_F32S>), so that a lane group's 16 bytes hold the same eight channels on both sides
      auto put = [&](char* img, int rows, int row, int gg, const uint4& v) {
        uint32_t h0, l0, h1, l1;
        split_pair(__uint_as_float(v.x), __uint_as_float(v.y), h0, l0);
        split_pair(__uint_as_float(v.z), __uint_as_float(v.w), h1, l1);
        char* dst = img + row * 64 + ((((gg & 3) ^ ((row >> 2) & 3))) << 4) + (gg >> 2) * 8;
        *(uint2*)dst = make_uint2(h0, h1);
        *(uint2*)(dst + rows * 64) = make_uint2(l0, l1);
      };
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i) put(As, BM, (i * 256 + tid) / G, g, ra[i]);
      // weights arrive split and permuted: 16-byte piece gg of a row's 128-byte chunk is K positions 8 (gg & 3) .. +7 of the
      // hi (gg < 4) or lo plane
#pragma unroll
      for (int j = 0; j < B_LOADS; ++j) {
        const int idx = j * 256 + tid;
        if (idx < B_ITEMS) {
          const int row = idx / G, gg = idx % G;
          *(uint4*)(Bs + (gg >> 2) * BN * 64 + row * 64 + ((((gg & 3) ^ ((row >> 2) & 3))) << 4)) = rb[j];
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i) {
        const int row = (i * 256 + tid) / G;
        *(uint4*)(As + row * RS + g * 16) = ra[i];
      }
#pragma unroll
      for (int j = 0; j < B_LOADS; ++j) {
        const int idx = j * 256 + tid;
        if (idx < B_ITEMS) *(uint4*)(Bs + (idx / G) * RS + (idx % G) * 16) = rb[j];
      }
    }
  };

  f32x4_v acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4_v){0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) {
    const char* As = smem + buf * Cfg::A_BYTES + (wm * MT * 16 + (lane & 15)) * RS;
    const char* Bs = smem + 2 * Cfg::A_BYTES + buf * Cfg::B_BYTES + (wn * NT * 16 + (lane & 15)) * RS;
    if constexpr (SPLIT) {
      const int fo = ((lane >> 4) ^ ((lane >> 2) & 3)) << 4;     // my 8 k of the step, swizzled like the stores
      bf16x8_v ah[MT], al[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        ah[i] = *(const bf16x8_v*)(As + i * 16 * 64 + fo);
        al[i] = *(const bf16x8_v*)(As + BM * 64 + i * 16 * 64 + fo);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const bf16x8_v bh = *(const bf16x8_v*)(Bs + j * 16 * 64 + fo);
        const bf16x8_v bl = *(const bf16x8_v*)(Bs + BN * 64 + j * 16 * 64 + fo);
#pragma unroll
        for (int i = 0; i < MT; ++i) {     // small terms first (weights as A: transposed tile, see conv_epilogue)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah[i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al[i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah[i], acc[i][j], 0, 0, 0);
        }
      }
    } else if constexpr (sizeof(T) == 2) {
      bf16x8_v af[MT], bfr[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = *(const bf16x8_v*)(As + i * 16 * RS + (lane >> 4) * 16);
#pragma unroll
      for (int j = 0; j < NT; ++j) bfr[j] = *(const bf16x8_v*)(Bs + j * 16 * RS + (lane >> 4) * 16);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // (weights as A: transposed tile, see conv_epilogue)
    } else {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        float af[MT], bfr[NT];
        const int kb = (kk * 4 + (lane >> 4)) * 4;
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = *(const float*)(As + i * 16 * RS + kb);
#pragma unroll
        for (int j = 0; j < NT; ++j) bfr[j] = *(const float*)(Bs + j * 16 * RS + kb);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[j], af[i], acc[i][j], 0, 0, 0);
      }
    }
  };

  // ---- main loop: one barrier per K chunk, loads of chunk i+1 in flight
  //      while the MFMAs of chunk i run -------------------------------------
  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  for (int it = 0; it < nchunks; ++it) {
    const int buf = it & 1;
    if (it + 1 < nchunks) load_tiles(it + 1);
    compute(buf);
    if (it + 1 < nchunks) store_tiles(buf ^ 1);
    __syncthreads();
  }

  conv_epilogue<MT, NT, WARPS_M, WARPS_N>(a, acc, smem, tile_m, tile_n);
}

// ---- host-side dispatch helpers ---------------------------------------------
struct ConvTile { int MT, NT, WM, WN; int BM() const { return 16 * MT * WM; } int BN() const { return 16 * NT * WN; } };

// Tile choice is a pure function of (dtype, mode, M, N) so callers can size the
// statistics workspace (vinet_conv3d_tile_m).
ConvTile vinet_pick_conv_tile(int dtype, int mode, long M, int N, long kchunks, bool may_split = false);
int vinet_launch_conv_bf16(const ConvTile& t, int mode, const ConvArgs& a, hipStream_t s);
int vinet_launch_conv_f32(const ConvTile& t, int mode, const ConvArgs& a, hipStream_t s, bool split = false);
int vinet_launch_conv_dma_bf16(const ConvTile& t, const ConvArgs& a, hipStream_t s);
int vinet_launch_conv_pp_bf16(int bn, const ConvArgs& a, hipStream_t s);
int vinet_launch_conv_ht_bf16(int nt, int tw, int tm, int pre, const ConvArgs& a, hipStream_t s);
int vinet_launch_conv_pw_bf16(int nt, const ConvArgs& a, hipStream_t s);

template <typename T, int MT, int NT, int WM, int WN, int MODE, bool SPLIT = false>
static int launch_conv_cfg(const ConvArgs& a, hipStream_t s) {
  using Cfg = ConvCfg<T, MT, NT, WM, WN, SPLIT>;
  auto kern = conv_igemm_kernel<T, MT, NT, WM, WN, MODE, SPLIT>;
  static bool attr_done[64] = {false};  // per device; benign race (same value written)
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_done[dev & 63]) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM);
    if (e != hipSuccess) { vinet_set_error("hipFuncSetAttribute(conv): %s", hipGetErrorString(e)); return (int)e; }
    attr_done[dev & 63] = true;
  }
  const int grid = a.tilesM * a.tilesN;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), Cfg::SMEM, s, a);
  return vn_launch_status("conv_igemm");
}
