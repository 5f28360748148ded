// Host side of the convolution entry points + library-wide error state.
#include <stdarg.h>
#include <stdlib.h>

#include "conv_igemm.h"

static thread_local char g_err[512] = "";
extern int g_vinet_opt_tperm;
extern int g_vinet_opt_epi_rows;
extern int g_vinet_opt_n64_tile;
extern int g_vinet_opt_sk_tile;
extern int g_vinet_opt_n192_tile;
extern int g_vinet_opt_n64_kmax;
extern int g_vinet_opt_n128_kmax;
extern int g_vinet_opt_n128_tile;

void vinet_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vinet_last_error(void) { return g_err; }
extern "C" int vinet_abi_version(void) { return VINET_ABI_VERSION; }

extern int g_vinet_opt_splitk;
// may this launch split its K loop?  (no statistics, no accumulation, and scratch lent -- or, for the size query that
// precedes the lending, assumed lent)
static bool vinet_conv_may_split(const VinetConvDesc* d, bool query) {
  return g_vinet_opt_splitk && d->dtype == VINET_BF16 && !d->stats && !d->accumulate && !d->bnb_partials && (query || d->splitk_ws != nullptr);
}

// Largest BN whose padded width is within 25% of the best achievable padding;
// then shrink BM while the grid would leave most of the 256 CUs idle.
ConvTile vinet_pick_conv_tile(int dtype, int mode, long M, int N, long kchunks, bool may_split) {
  if (mode == VINET_CONV_STEM) return dtype == VINET_BF16 ? ConvTile{4, 4, 4, 1} : ConvTile{2, 4, 4, 1};
  static const int nts[6] = {8, 6, 4, 3, 2, 1};
  int best_pad = 1 << 30;
  for (int i = 0; i < 6; ++i) {
    const int bn = nts[i] * 16;
    const int pad = ((N + bn - 1) / bn) * bn;
    if (pad < best_pad) best_pad = pad;
  }
  int nt = 1;
  for (int i = 0; i < 6; ++i) {
    const int bn = nts[i] * 16;
    const int pad = ((N + bn - 1) / bn) * bn;
    if (pad * 4 <= best_pad * 5) { nt = nts[i]; break; }
  }
  if (vn_f32_storage(dtype)) return ConvTile{2, nt, 4, 1};  // BM = 128
  // long K loops on grids far below the chip (batch-1 decoder convs) whose caller lends split-K scratch: split-K supplies the
  // workgroups, so the tile can be the large one (fewer LDS bytes per MFMA, and 128 x 192 covers N = 192 without padding it
  // to 256): batch-1 replay 644 -> 668 fps, batch 2 959 -> 980
  if (may_split && g_vinet_opt_sk_tile && kchunks >= 96 && M >= 1024 && ((M + 127) / 128) * ((N + 127) / 128) < 256) {
    if ((g_vinet_opt_sk_tile & 1) && N % 192 == 0) return ConvTile{4, 6, 2, 2};
    if ((g_vinet_opt_sk_tile & 2) && N >= 128) return ConvTile{4, 4, 2, 2};
  }
  if (may_split && (g_vinet_opt_sk_tile & 12) && kchunks >= 96 && M >= 4096 && N > 32 && N <= 64 && (M + 63) / 64 < 512)
    return (g_vinet_opt_sk_tile & 8) ? ConvTile{4, 4, 4, 1} : ConvTile{4, 2, 2, 2};      // 64-wide: 256 x 64 / 128 x 64 instead of 64 x 64
  ConvTile t{4, nt, 4, 1};                               // BM = 256
  if (g_vinet_opt_n64_tile && nt == 4) return g_vinet_opt_n64_tile == 1 ? ConvTile{4, 2, 2, 2} : ConvTile{2, 2, 2, 2};   // tuning
  // 64-wide outputs with a short K loop (the stem, its 7x1x1 partner and their dgrads: 7-16 K steps at
  // 10-20 M voxels) are prologue / epilogue bound: 128-row tiles put 4-6 workgroups on a CU instead of 2-3
  // (+5...17 % measured, tools/conv_ab.py; still +7 % at 54 K steps -- the data gradient of 64 -> 192 1x3x3 -- and
  // neutral at 270)
  if (nt == 4 && kchunks > 0 && kchunks <= g_vinet_opt_n64_kmax && M >= (1L << 20)) return ConvTile{4, 2, 2, 2};
  // 192-wide tile (128 x 192, waves 2 x 2): one column tile covers N = 192, so x is staged once per row tile instead of
  // twice and the B rows need no padding to whole wave-instructions (96 -> 128): +7...10 % on the 192-channel layers
  // at 56 x 96 and 28 x 48 (tools/conv_ab.py, 64 clips)
  if (nt == 6 && g_vinet_opt_n192_tile && N % 192 == 0 && (g_vinet_opt_n192_tile >= 2 || ((M + 127) / 128) * (N / 192) >= 512))
    return ConvTile{4, 6, 2, 2};
  // the 96-wide shape has no small-M variants: a grid that covers under a quarter of the CUs (batch-1 decoder
  // convs) moves to the 128-wide family, which does
  if (nt == 6 && ((M + 255) / 256) * ((N + 95) / 96) < 64) nt = 8;
  if (g_vinet_opt_n128_tile && nt == 8) return g_vinet_opt_n128_tile == 1 ? ConvTile{4, 4, 2, 2} : ConvTile{2, 4, 2, 2};   // tuning
  // 128-wide outputs with a short K loop (the pointwise convs and their data gradients: 6-20 K steps) are prologue /
  // epilogue bound like the 64-wide ones: 128 x 128 tiles put twice the workgroups on a CU (+13 % at 8-9 K steps, +5...10 % at
  // 16-17, but -14 % at 54: tools/conv_ab.py)
  if (nt == 8 && kchunks > 0 && kchunks <= g_vinet_opt_n128_kmax && ((M + 127) / 128) * ((N + 127) / 128) >= 1024) return ConvTile{4, 4, 2, 2};
  if (nt == 8 || nt == 4) {
    const int bn = nt * 16;
    const long tilesN = (N + bn - 1) / bn;
    const long blocks256 = ((M + 255) / 256) * tilesN;
    if (blocks256 < 512) {
      const long blocks128 = ((M + 127) / 128) * tilesN;
      if (blocks128 >= 512 || bn == 64) t = (bn == 128) ? ConvTile{4, 4, 2, 2} : ConvTile{4, 2, 2, 2};  // BM 128
      if (blocks128 < 512) t = (bn == 128) ? ConvTile{2, 4, 2, 2} : ConvTile{2, 2, 2, 2};               // BM 64
    }
  }
  return t;
}

static int fill_args(const VinetConvDesc* d, ConvArgs& a, ConvTile& t) {
  VN_CHECK_ARG(d != nullptr, "conv: null descriptor");
  VN_CHECK_ARG(d->dtype == VINET_F32 || d->dtype == VINET_BF16 || d->dtype == VINET_F32S, "conv: bad dtype %d", d->dtype);
  VN_CHECK_ARG(d->out_dtype == VINET_F32 || d->out_dtype == VINET_BF16, "conv: bad out_dtype %d", d->out_dtype);
  VN_CHECK_ARG(d->mode == VINET_CONV_GENERIC || d->mode == VINET_CONV_STEM, "conv: bad mode %d", d->mode);
  const int eg = vn_f32_storage(d->dtype) ? 4 : 8;
  VN_CHECK_ARG(vn_tensor_ok(d->x, d->mode == VINET_CONV_STEM ? 4 : eg, true),
               "conv: bad x view (C=%d ld=%d must be multiples of %d, 16-byte aligned)", d->x.C, d->x.ld, eg);
  VN_CHECK_ARG(d->y.ptr && d->y.C > 0 && d->y.ld >= d->y.C, "conv: bad y view");
  VN_CHECK_ARG(d->x.B == d->y.B, "conv: batch mismatch %d vs %d", d->x.B, d->y.B);
  VN_CHECK_ARG(d->ntaps > 0 && d->taps && d->w, "conv: taps/weights missing");
  VN_CHECK_ARG(d->Kp > 0 && d->Kp % 32 == 0, "conv: Kp=%d must be a multiple of 32", d->Kp);
  if (d->mode == VINET_CONV_STEM) VN_CHECK_ARG(d->x.C == 4 && d->Kp == 32, "conv stem: x.C must be 4 and Kp 32");
  else VN_CHECK_ARG(d->Kp >= d->x.C, "conv: Kp=%d < Cin=%d", d->Kp, d->x.C);
  VN_CHECK_ARG(d->oT > 0 && d->oH > 0 && d->oW > 0, "conv: empty iteration space");
  VN_CHECK_ARG(d->sT > 0 && d->sH > 0 && d->sW > 0 && d->omT > 0 && d->omH > 0 && d->omW > 0, "conv: bad strides");
  VN_CHECK_ARG((d->oT - 1) * d->omT + d->ooT < d->y.T && (d->oH - 1) * d->omH + d->ooH < d->y.H &&
                   (d->oW - 1) * d->omW + d->ooW < d->y.W && d->ooT >= 0 && d->ooH >= 0 && d->ooW >= 0,
               "conv: output placement outside y");
  const long M = (long)d->x.B * d->oT * d->oH * d->oW;
  VN_CHECK_ARG(M < (1L << 31), "conv: M too large");

  a.x = (const char*)d->x.ptr; a.y = (char*)d->y.ptr; a.w = (const char*)d->w; a.taps = (const int4*)d->taps;
  a.in_scale = d->pre.scale; a.in_shift = d->pre.shift; a.in_relu = d->pre.relu;
  a.out_scale = d->out_scale; a.out_shift = d->out_shift; a.stats = d->stats;
  a.Ti = d->x.T; a.Hi = d->x.H; a.Wi = d->x.W; a.Cin = d->x.C; a.ldx = d->x.ld; a.sBx = d->x.sB;
  a.To = d->oT; a.Ho = d->oH; a.Wo = d->oW;
  a.sT = d->sT; a.sH = d->sH; a.sW = d->sW;
  a.yT = d->y.T; a.yH = d->y.H; a.yW = d->y.W; a.N = d->y.C; a.ldy = d->y.ld; a.sBy = d->y.sB;
  a.omT = d->omT; a.omH = d->omH; a.omW = d->omW; a.ooT = d->ooT; a.ooH = d->ooH; a.ooW = d->ooW;
  a.ntaps = d->ntaps; a.Kp = d->Kp; a.M = (int)M;
  a.Nw = d->n_valid > 0 ? d->n_valid : d->y.C;
  VN_CHECK_ARG(a.Nw <= d->y.C, "conv: n_valid > y.C");
  a.dW = make_fastdiv(d->oW); a.dH = make_fastdiv(d->oH); a.dT = make_fastdiv(d->oT);
  a.y_linear = (d->omT == 1 && d->omH == 1 && d->omW == 1 && d->ooT == 0 && d->ooH == 0 && d->ooW == 0 &&
                d->y.T == d->oT && d->y.H == d->oH && d->y.W == d->oW &&
                d->y.sB == (int64_t)d->y.T * d->y.H * d->y.W * d->y.ld) ? 1 : 0;
  a.act = d->act; a.accumulate = d->accumulate; a.out_f32 = d->out_dtype == VINET_F32;
  a.epi_rows = g_vinet_opt_epi_rows;
  const int oeb = a.out_f32 ? 4 : 2;
  a.vec_ok = (a.N % 4 == 0) && (a.ldy % 4 == 0) && (a.sBy % 4 == 0) && ((((uintptr_t)d->y.ptr) % (4 * oeb)) == 0);
  t = vinet_pick_conv_tile(d->dtype, d->mode, M, a.N, (long)d->ntaps * (d->Kp / 32), vinet_conv_may_split(d, false));
  a.perm_P = a.perm_T = 0;
  a.dPT = a.dPermT = make_fastdiv(1);
  if (g_vinet_opt_tperm && d->dtype == VINET_BF16 && d->mode == VINET_CONV_GENERIC && t.BM() == 256 && d->oT > 1 &&
      ((long)d->oH * d->oW) % 256 == 0) {
    a.perm_P = (int)(((long)d->oH * d->oW) / 256);
    a.perm_T = d->oT;
    a.dPT = make_fastdiv((uint32_t)(a.perm_P * a.perm_T));
    a.dPermT = make_fastdiv((uint32_t)a.perm_T);
  }
  a.tilesM = vn_div_up(M, t.BM());
  a.tilesN = vn_div_up(a.N, t.BN());
  a.splits = 1; a.chunks_per_split = 0; a.ws = nullptr;
  a.ht_tilesH = a.ht_tilesW = 0;
  a.ht_dN = a.ht_dW = a.ht_dH = a.ht_dTo = make_fastdiv(1);
  a.bnb_z = nullptr; a.bnb_partials = nullptr; a.bnb_scale = a.bnb_shift = a.bnb_mean = a.bnb_invstd = nullptr;
  a.bnb_ldz = 0; a.bnb_sBz = 0; a.bnb_z_linear = 0; a.bnb_relu = 0;
  return 0;
}

static bool use_pp(const VinetConvDesc* d);
static bool use_ht(const VinetConvDesc* d);
int vinet_launch_conv_dma3(int nt, const ConvArgs& a, hipStream_t s);
int g_vinet_opt_dma3 = 1;        // LDS-DMA kernel for the split-bf16 form (0 = the register-staged kernel everywhere)
// conv_dma3.h: fp32 tensors + split-bf16 arithmetic; a pending affine only as BatchNorm + ReLU (NaN-page padding)
static bool use_dma3(const VinetConvDesc* d) {
  const bool pre_ok = !d->pre.scale || (d->pre.relu && d->pre.shift && d->Kp <= 1024);
  return g_vinet_opt_dma3 && d->dtype == VINET_F32S && d->mode == VINET_CONV_GENERIC && pre_ok && !(d->pre.relu && !d->pre.scale);
}
static int dma3_nt(const VinetConvDesc* d) { return d->y.C <= 32 ? 2 : 4; }
static bool use_pw(const VinetConvDesc* d);
struct HtShape { int nt, tw, tm, pre; };
struct PwShape { int nt, tilesN, gm, tpw; };
static PwShape pw_shape(const VinetConvDesc* d);
static HtShape ht_shape(const VinetConvDesc* d);
extern int g_vinet_opt_ht, g_vinet_opt_ht3, g_vinet_opt_ht_minhw, g_vinet_opt_ht_t, g_vinet_opt_ht_pre, g_vinet_opt_ht_t_minhw;
bool vinet_conv_use_ts(const VinetConvDesc* d);
int vinet_conv_ts_positions(const VinetConvDesc* d);
int vinet_conv_ts_segments(const VinetConvDesc* d);
int vinet_conv_hs_segments(const VinetConvDesc* d);
bool vinet_conv_use_hs(const VinetConvDesc* d);
bool vinet_conv_use_tsd(const VinetConvDesc* d);
int vinet_launch_conv_tsd(const VinetConvDesc* d, hipStream_t s);
int vinet_launch_conv_hs(const VinetConvDesc* d, hipStream_t s);
extern int g_vinet_opt_conv_hs;
extern int g_vinet_opt_conv_hs_segs;
int vinet_launch_conv_ts(const VinetConvDesc* d, hipStream_t s);
extern int g_vinet_opt_conv_ts;
extern int g_vinet_opt_conv_ts_segs;
extern int g_vinet_opt_wgrad_hs;
extern int g_vinet_opt_wgrad_rs;
extern int g_vinet_opt_wgrad_rs4;
extern int g_vinet_opt_wgrad_tf;
extern int g_vinet_opt_wgrad_skinny;
extern int g_vinet_opt_bn_lean;
extern int g_vinet_opt_bnb_epi;
extern int g_vinet_opt_reduce_small;
extern int g_vinet_opt_bn_rows;
extern int g_vinet_opt_pack_tiled;
extern int g_vinet_opt_wgrad_pp_cap;
extern int g_vinet_opt_wgrad_ts_cap;

extern "C" int vinet_conv3d_tile_m(const VinetConvDesc* d) {
  if (!d) return -1;
  if (vinet_conv_use_ts(d) || vinet_conv_use_hs(d)) return 64;
  if (use_ht(d) || use_pp(d)) return 256;
  const long M = (long)d->x.B * d->oT * d->oH * d->oW;
  return vinet_pick_conv_tile(d->dtype, d->mode, M, d->y.C, (long)d->ntaps * (d->Kp / 32), vinet_conv_may_split(d, false)).BM();
}

/* rows of the [rows][2][N] statistics table this problem's launch fills (one per M tile; the halo-tile kernel's tiles
 * are spatial, so partial tiles at the image border make it more than ceil(M / tile_m)) */
extern "C" int vinet_conv3d_stats_rows(const VinetConvDesc* d) {
  if (!d) return -1;
  const long M = (long)d->x.B * d->oT * d->oH * d->oW;
  // the strip / frame-streaming kernels of the stem keep their partial sums in registers over a whole item (a 64-wide strip of
  // one frame, 64 positions of one clip) and write ONE row per item
  if (vinet_conv_use_hs(d)) return (int)((long)d->x.B * d->oT * (d->oW / 64) * vinet_conv_hs_segments(d));
  if (vinet_conv_use_ts(d)) return (int)((long)d->x.B * (((long)d->oH * d->oW) / vinet_conv_ts_positions(d)) * vinet_conv_ts_segments(d));
  if (!vinet_conv_use_ts(d) && !vinet_conv_use_hs(d) && use_pw(d)) return pw_shape(d).gm;   // one row per workgroup (4 waves x up to 16 tiles of 64 rows)
  if (!vinet_conv_use_ts(d) && !vinet_conv_use_hs(d) && use_ht(d)) {
    const HtShape h = ht_shape(d);
    if (h.tm) return (int)((long)d->x.B * vn_div_up(d->oT, 4) * vn_div_up((long)d->oH * d->oW, 64));
    return (int)((long)d->x.B * d->oT * vn_div_up(d->oH, 256 / h.tw) * vn_div_up(d->oW, h.tw));
  }
  const int bm = vinet_conv3d_tile_m(d);
  return (int)((M + bm - 1) / bm);
}

int g_vinet_opt_dma = 1;
int g_vinet_opt_epi_rows = 0;   // conv epilogue (bf16 fast path): 1 = whole-row stores through a wave-private LDS image.  Measured (tools/conv_ab.py --opt epi_rows=0,1, profiles/r3_epi_rows_ab.txt): neutral on conv_dma, 1...7 % slower on the halo-tile kernels, whole step 306.5 -> 307.4 ms: off.  (The pointwise kernel, conv_pw.h, always stores whole rows: there it is worth 2x.)
int g_vinet_opt_pp_pw_kt = 8;    // ping-pong kernel on pointwise layers from this many K tiles of 64 (16 = as for every other layer; 5 wins alone from Cin = 304 on, but costs the training step 0.3 % beside the weight-gradient stream: 8 = Cin >= 480)
int g_vinet_opt_pw_maxtn = 4;   // pointwise kernel: at most this many column tiles (each re-reads x)
int g_vinet_opt_pw = 1;        // pointwise streaming kernel (conv_pw.h) for 1x1x1 convs and their data gradients (2 = also on small grids: tests)
extern int g_vinet_opt_splitk;
int g_vinet_opt_n64_tile = 0;   // tuning: 64-wide layers on 128x64 (1) or 64x64 (2) tiles instead of 256x64
int g_vinet_opt_pool_blk = 1;   // 1x3x3/s(1,2,2) max-pool backward per 2x2 input block
int g_vinet_opt_up_blk = 1;     // 8-channel upsample kernels (forward per 2x2 output block)
int g_vinet_opt_n128_tile = 0;   // tuning: 128-wide layers on 128x128 (1) or 64x128 (2) tiles instead of 256x128
int g_vinet_opt_n128_kmax = 64;  // 128-wide outputs: 128-row tiles up to this many K steps of 32 (0 = never).  Whole step (alternating runs, end of round 3): 0: 304.3 ms, 20: 301.1...301.5, 40: 299.1...300.1, 64: 299.2...299.8, 100: 299.1
int g_vinet_opt_n64_kmax = 64;   // 64-wide outputs: 128-row tiles up to this many K steps of 32
int g_vinet_opt_n192_tile = 1;   // 128 x 192 tiles (waves 2 x 2) for N % 192 == 0 instead of 256 x 96 (0 = off, 2 = also on small grids: tests)
int g_vinet_opt_reduce_il = 1;  // channel reductions: blocks interleave rounds over one window (0 = one contiguous range per block)
int g_vinet_opt_pool_pk = 1;    // bf16: packed 32-bit-key form of the LDS halo-tile pool (0 = the fp32-compare kernel)
int g_vinet_opt_pool_lds = 1;   // LDS halo-tile 3x3x3/s1 max-pool forward (C % 64 == 0)
int g_vinet_opt_pool_twalk = 1; // T-walking 3x3x3/s1 max-pool backward (2 = force on small grids, 3 = conditional-load form, 4 = bf16 without the EXEC-mask routing)
int g_vinet_opt_tperm = 0;      // t-fastest M-tile order (L2 reuse across temporal taps): measured neutral on the whole step, off
int g_vinet_opt_wgrad_tr = 1;
int g_vinet_opt_wgrad_dma = 1;
int g_vinet_opt_pp = 1;         // 256x256x64 ping-pong kernel for large plain convs
int g_vinet_opt_wgrad_pp = 1;   // 256x256x64 ping-pong wgrad for large layers (2 = force)
int g_vinet_opt_wgrad_ts = 1;   // frame-streaming wgrad for temporal 64 -> 64 convs (2 = force on any eligible shape)
int g_vinet_opt_wgrad_tg = 0;   // tuning: force taps per group in the DMA wgrad (0 = heuristic)

extern "C" int vinet_set_option(const char* name, int32_t value) {
  if (name && !strcmp(name, "dma")) { g_vinet_opt_dma = value; return 0; }
  if (name && !strcmp(name, "dma3")) { g_vinet_opt_dma3 = value; return 0; }
  if (name && !strcmp(name, "bnb_epi")) { g_vinet_opt_bnb_epi = value; return 0; }
  if (name && !strcmp(name, "reduce_small")) { g_vinet_opt_reduce_small = value; return 0; }
  if (name && !strcmp(name, "pw")) { g_vinet_opt_pw = value; return 0; }
#ifndef VINET_EXPERIMENTS
  // measured-slower variants live in side builds only (python -c "from vinet_amd import build; build.build_variant('exp', ['-DVINET_EXPERIMENTS'])")
  if (name && value && (!strcmp(name, "epi_rows") || (!strcmp(name, "bn_lean") && value == 2))) {
    vinet_set_error("set_option: %s=%d needs a -DVINET_EXPERIMENTS build of the library", name, value);
    return -2;
  }
#endif
  if (name && !strcmp(name, "epi_rows")) { g_vinet_opt_epi_rows = value; return 0; }
  if (name && !strcmp(name, "pw_maxtn")) { g_vinet_opt_pw_maxtn = value; return 0; }
  if (name && !strcmp(name, "pp_pw_kt")) { g_vinet_opt_pp_pw_kt = value; return 0; }
  if (name && !strcmp(name, "pool_blk")) { g_vinet_opt_pool_blk = value; return 0; }
  if (name && !strcmp(name, "up_blk")) { g_vinet_opt_up_blk = value; return 0; }
  if (name && !strcmp(name, "n128_tile")) { g_vinet_opt_n128_tile = value; return 0; }
  if (name && !strcmp(name, "n128_kmax")) { g_vinet_opt_n128_kmax = value; return 0; }
  if (name && !strcmp(name, "n64_kmax")) { g_vinet_opt_n64_kmax = value; return 0; }
  if (name && !strcmp(name, "n192_tile")) { g_vinet_opt_n192_tile = value; return 0; }
  if (name && !strcmp(name, "reduce_il")) { g_vinet_opt_reduce_il = value; return 0; }
  if (name && !strcmp(name, "pool_pk")) { g_vinet_opt_pool_pk = value; return 0; }
  if (name && !strcmp(name, "pool_lds")) { g_vinet_opt_pool_lds = value; return 0; }
  if (name && !strcmp(name, "pool_twalk")) { g_vinet_opt_pool_twalk = value; return 0; }
  if (name && !strcmp(name, "n64_tile")) { g_vinet_opt_n64_tile = value; return 0; }
  if (name && !strcmp(name, "tperm")) { g_vinet_opt_tperm = value; return 0; }
  if (name && !strcmp(name, "pp")) { g_vinet_opt_pp = value; return 0; }
  if (name && !strcmp(name, "ht")) { g_vinet_opt_ht = value; return 0; }
  if (name && !strcmp(name, "ht3")) { g_vinet_opt_ht3 = value; return 0; }
  if (name && !strcmp(name, "bn_lean")) { g_vinet_opt_bn_lean = value; return 0; }
  if (name && !strcmp(name, "wgrad_ts_cap")) { g_vinet_opt_wgrad_ts_cap = value; return 0; }
  if (name && !strcmp(name, "wgrad_pp_cap")) { g_vinet_opt_wgrad_pp_cap = value; return 0; }
  if (name && !strcmp(name, "pack_tiled")) { g_vinet_opt_pack_tiled = value; return 0; }
  if (name && !strcmp(name, "bn_rows")) { g_vinet_opt_bn_rows = value < 1 ? 1 : value; return 0; }
  if (name && !strcmp(name, "ht_minhw")) { g_vinet_opt_ht_minhw = value; return 0; }
  if (name && !strcmp(name, "ht_t")) { g_vinet_opt_ht_t = value; return 0; }
  if (name && !strcmp(name, "ht_pre")) { g_vinet_opt_ht_pre = value; return 0; }
  if (name && !strcmp(name, "ht_t_minhw")) { g_vinet_opt_ht_t_minhw = value; return 0; }
  if (name && !strcmp(name, "splitk")) { g_vinet_opt_splitk = value; return 0; }
  if (name && !strcmp(name, "sk_tile")) { g_vinet_opt_sk_tile = value; return 0; }
  if (name && !strcmp(name, "conv_hs_segs")) { g_vinet_opt_conv_hs_segs = value; return 0; }
  if (name && !strcmp(name, "conv_ts_segs")) { g_vinet_opt_conv_ts_segs = value; return 0; }
  if (name && !strcmp(name, "wgrad_pp")) { g_vinet_opt_wgrad_pp = value; return 0; }
  if (name && !strcmp(name, "wgrad_tr")) { g_vinet_opt_wgrad_tr = value; return 0; }
  if (name && !strcmp(name, "wgrad_dma")) { g_vinet_opt_wgrad_dma = value; return 0; }
  if (name && !strcmp(name, "conv_hs")) { g_vinet_opt_conv_hs = value; return 0; }
  if (name && !strcmp(name, "conv_ts")) { g_vinet_opt_conv_ts = value; return 0; }
  if (name && !strcmp(name, "wgrad_rs")) { g_vinet_opt_wgrad_rs = value; return 0; }
  if (name && !strcmp(name, "wgrad_rs4")) { g_vinet_opt_wgrad_rs4 = value; return 0; }
  if (name && !strcmp(name, "wgrad_skinny")) { g_vinet_opt_wgrad_skinny = value; return 0; }
  if (name && !strcmp(name, "wgrad_tf")) { g_vinet_opt_wgrad_tf = value; return 0; }
  if (name && !strcmp(name, "wgrad_hs")) { g_vinet_opt_wgrad_hs = value; return 0; }
  if (name && !strcmp(name, "wgrad_ts")) { g_vinet_opt_wgrad_ts = value; return 0; }
  if (name && !strcmp(name, "wgrad_tg")) { g_vinet_opt_wgrad_tg = value; return 0; }
  vinet_set_error("set_option: unknown option %s", name ? name : "(null)");
  return -1;
}

static bool use_dma(const VinetConvDesc* d) {
  // a pending affine is supported when it comes with ReLU (BN+ReLU, the only kind the nets
  // produce): the NaN-page padding trick needs the max(.,0); Kp <= 1024 for the LDS table
  const bool pre_ok = !d->pre.scale || (d->pre.relu && d->pre.shift && d->Kp <= 1024);
  return g_vinet_opt_dma && d->dtype == VINET_BF16 && d->mode == VINET_CONV_GENERIC && pre_ok &&
         !(d->pre.relu && !d->pre.scale);
}

// conv_pw.h: the caller promises (tline == 6) a single tap (0, 0, 0, slice 0); unit strides, dense placement, bf16 in and out,
// channel counts in whole 16-byte groups, a weight tile (32 / 64 / 96 columns x Kp) that leaves room for two workgroups per CU
static PwShape pw_shape(const VinetConvDesc* d) {
  PwShape h;
  const int N = d->y.C;
  const long lds_max = 80 * 1024;      // two workgroups per CU
  auto fits = [&](int nt) {            // ConvPwCfg<nt>::smem_bytes
    return (long)nt * 16 * (d->Kp * 2 + 16) + 2L * nt * 16 * 4 + (d->pre.scale ? 2L * d->Kp * 4 : 0) + 4L * nt * 16 * 8 + 4L * 32 * nt * 16 * 2 <= lds_max;
  };
  int best = 0, bestpad = 1 << 30;
  const int nts[3] = {6, 4, 2};
  for (int i = 0; i < 3; ++i) {
    if (!fits(nts[i])) continue;
    const int bn = nts[i] * 16, pad = (N + bn - 1) / bn * bn;
    if (pad < bestpad) { bestpad = pad; best = nts[i]; }
  }
  h.nt = best;
  h.tilesN = best ? (N + best * 16 - 1) / (best * 16) : 0;
  const long M = (long)d->x.B * d->oT * d->oH * d->oW;
  const long nwt = (M + 63) / 64;
  // tiles per wave: at least ~8 workgroups per CU-slot in the grid (the dispatcher balances them over whatever the second
  // stream leaves free), at most 16 (the weight tile is staged once per 4 x 16 tiles = 4096 rows: under 5 % of the row traffic)
  long tpw = best ? nwt / (4L * 4096 / h.tilesN) : 1;
  tpw = tpw < 1 ? 1 : (tpw > 16 ? 16 : tpw);
  h.tpw = (int)tpw;
  const long gm = (nwt + 4 * tpw - 1) / (4 * tpw);
  h.gm = (int)(gm < 1 ? 1 : gm);
  return h;
}
static bool use_pw(const VinetConvDesc* d) {
  if (!g_vinet_opt_pw || !use_dma(d) || d->tline != 6 || d->ntaps != 1) return false;
  if (d->sT != 1 || d->sH != 1 || d->sW != 1 || d->omT != 1 || d->omH != 1 || d->omW != 1 || d->ooT || d->ooH || d->ooW) return false;
  if (d->oT != d->x.T || d->oH != d->x.H || d->oW != d->x.W || d->y.T != d->oT || d->y.H != d->oH || d->y.W != d->oW) return false;
  if (d->out_dtype != VINET_BF16 || (d->act != VINET_ACT_NONE && d->act != VINET_ACT_RELU) || d->accumulate) return false;
  if (d->y.C % 8 || d->y.ld % 8 || d->y.sB % 8 || ((uintptr_t)d->y.ptr) % 16 || d->x.sB % 8) return false;
  const PwShape h = pw_shape(d);
  if (!h.nt) return false;
  if (g_vinet_opt_pw >= 2) return true;
  // a grid that cannot fill the chip stays with conv_dma (which splits its K loop over workgroups there).  Every column tile
  // re-reads x (from L2 at best): measured at 192 clips (tools/conv_ab.py --pw, profiles/r3_pw_ab.txt) the kernel wins with up
  // to four column tiles at Cin <= 288 (256 -> 288: 1.58 -> 1.48 ms plain, 1.86 -> 1.51 with a pending affine; 192 -> 176:
  // 0.90 -> 0.82; 64 -> 64: 0.87 -> 0.79 = the HBM roofline) and with any number of tiles at Cin <= 64 (64 -> 512: 0.22 -> 0.16),
  // and loses where a 512-channel input leaves room for a 32- / 64-column weight tile only (512 -> 256: 0.24 -> 0.53)
  const long M = (long)d->x.B * d->oT * d->oH * d->oW;
  return M >= 64L * 2048 && d->Kp <= 320 && (h.tilesN <= g_vinet_opt_pw_maxtn || d->Kp <= 64);
}

// tile width of the ping-pong kernel: 256 or 192, whichever pads N less (ties: 256)
static int pp_bn(int N) {
  if (g_vinet_opt_pp == 3) return 256;   // tuning: force shapes
  if (g_vinet_opt_pp == 4) return 192;
  // measured (tools/conv_ab.py): the 192 shape does 12 MFMAs per phase against the same staging
  // work, so it only wins when it saves at least ~15% of the padded columns
  const int p256 = (N + 255) / 256 * 256, p192 = (N + 191) / 192 * 192;
  return p192 * 20 <= p256 * 17 ? 192 : 256;
}

// conv_pp.h: plain bf16 inputs, enough K tiles to amortise the 6-half-tile prologue, enough
// output channels to use a 256-wide tile, enough tiles to occupy the chip
static bool use_ht(const VinetConvDesc* d);
static bool use_pp(const VinetConvDesc* d) {
  if (use_ht(d)) return false;
  if (!g_vinet_opt_pp || !use_dma(d) || d->pre.scale || d->ntaps > 64) return false;
  const long M = (long)d->x.B * d->oT * d->oH * d->oW;
  const int N = d->y.C;
  const long nkt = (long)d->ntaps * ((d->Kp + 63) / 64);
  const int bn = pp_bn(N);
  const long tiles = ((M + 255) / 256) * ((N + bn - 1) / bn);
  if (g_vinet_opt_pp >= 2) return true;   // tuning: force
  // pointwise layers the streaming kernel does not take (Cin = 304...832 at 14 x 24 / 7 x 12): the alternative is conv_dma at
  // 2.6 k cycles per K step of 32, and the ping-pong kernel wins from 5 K tiles on -- 480 -> 304: 0.42 -> 0.33 ms, 512 -> 296:
  // 0.38 -> 0.32, 832 -> 624: 0.155 -> 0.106, 304 -> 480: 0.41 -> 0.37 (tools/conv_ab.py, 192 clips)
  return N >= 160 && nkt >= (d->ntaps == 1 ? g_vinet_opt_pp_pw_kt : 16) && tiles >= 128;
}

// conv_ht.h: the caller promises (tline == 5) that every tap is (dt, dh, dw, slice) with |dh|, |dw| <= 1 and that taps
// of equal dt are contiguous in the table; plain bf16 input, unit spatial stride, output extent = input extent, W a
// multiple of 16.  Shape: tile width 32 when W allows it, else 16; column tile = the narrowest of 64 / 96 / 128 that
// pads N least (192 = 2 x 96).
int g_vinet_opt_ht = 1;         // 0 = off, 1 = heuristic, 2 = every eligible conv (tests)
int g_vinet_opt_ht_minhw = 28 * 48;
int g_vinet_opt_ht_t = 1;        // temporal mode of the halo-tile kernel for (3,1,1) / stride-1 convs: 604 -> 647 TF/s plain, 482 -> 514 with a
                                 // pending affine (reuse is only 2x and the image is re-staged every three K steps); whole step neutral (+0...0.6 %)
int g_vinet_opt_ht_pre = 0;      // spatial mode on inputs with a pending BatchNorm + ReLU (1 = on).  The kernel itself wins (570 -> 790 TF/s
                                 // against conv_dma's per-fragment form) but the engine then skips the materialisation pass, and the row-streaming
                                 // weight gradients of those layers want plain inputs: whole step 613 -> 599 clips/s.  Off.
int g_vinet_opt_ht_t_minhw = 14 * 24;
int g_vinet_opt_ht3 = 1;         // halo-tile kernels for the split-bf16 form (VINET_F32S; 0 = conv_dma3 everywhere)
int vinet_launch_conv_ht_f32s(int nt, int tw, int tm, int pre, const ConvArgs& a, hipStream_t s);
// temporal mode: tline == 1 with three taps, padding 1, unit stride = taps (dt, 0, 0), dt in {-1, 0, 1}
static bool ht_temporal(const VinetConvDesc* d) {
  return d->tline == 1 && d->ntaps == 3 && d->tpad == 1 && d->sT == 1;
}
static HtShape ht_shape(const VinetConvDesc* d) {
  HtShape h;
  h.tm = ht_temporal(d) ? 1 : 0;
  h.pre = d->pre.scale ? 1 : 0;
  h.tw = (d->oW % 32 == 0) ? 32 : 16;
  const int N = d->n_valid > 0 ? d->n_valid : d->y.C;
  // 96-, 64- or 32-wide column tiles, whichever pads N least (ties: the widest); a 128-wide tile spills (12 B / lane);
  // the temporal and PRE forms exist for 96 and 64; the split-bf16 form (three MFMAs per product) for 64 and 32
  int best = 4, bestpad = 1 << 30;
  const int nts[3] = {6, 4, 2};
  const bool split = d->dtype == VINET_F32S;
  for (int i = split ? 1 : 0; i < ((h.tm || h.pre) && !split ? 2 : 3); ++i) {
    const int bn = nts[i] * 16, pad = (N + bn - 1) / bn * bn;
    if (pad < bestpad) { bestpad = pad; best = nts[i]; }
  }
  // split form: the 32-wide tile halves the work per staged halo image; it is worth it only where the 64-wide one would pad N by more
  // than 15 % (N = 32).  (N = 480, the data gradient of the 480 -> 192 decoder conv: 512 columns of 64-wide tiles run at 290 TF/s,
  // 480 columns of 32-wide ones at 190.)
  if (split && best == 2 && (N + 63) / 64 * 64 * 100 <= (N + 31) / 32 * 32 * 115) best = 4;
  h.nt = best;
  return h;
}
static bool use_ht(const VinetConvDesc* d) {
  const bool split = d->dtype == VINET_F32S;
  if (!g_vinet_opt_ht || !(split ? g_vinet_opt_ht3 && use_dma3(d) : use_dma(d)) || (d->pre.relu && !d->pre.scale)) return false;
  const bool tm = ht_temporal(d);
  if (!tm && d->tline != 5) return false;
  if (d->pre.scale && !(d->pre.relu && d->pre.shift && d->Kp <= 1024)) return false;
  if (d->sH != 1 || d->sW != 1 || d->oH != d->x.H || d->oW != d->x.W || d->ntaps > 64 || d->ntaps < 2) return false;
  if (tm ? ((d->oH * d->oW) % 16 != 0 || d->oT != d->x.T) : (d->oW % 16 != 0)) return false;
  if (g_vinet_opt_ht >= 2) return true;
  if (tm && !(g_vinet_opt_ht_t)) return false;
  if (!tm && d->pre.scale && !g_vinet_opt_ht_pre && !split) return false;      // (split form: no materialisation pass exists to lose; its weight gradients split x with the affine applied)
  // measured (tools/conv_ab.py --ht, 64 clips): wins wherever the 64-channel K chunks are (nearly) full -- 504 -> 995 TF/s on
  // the 192 -> 64 5x3x3 decoder conv, 528 -> 840 on the data gradient of 64 -> 192, 894 -> 1026 on 480 -> 192 (conv_pp before),
  // 331 -> 510 on 64 -> 32 -- and loses where a chunk is half padding (Cin = 32: 410 -> 310; Cin = 96: 608 -> 535)
  const int N = d->y.C;
  const int k64 = split ? d->Kp : (d->Kp + 63) / 64 * 64;     // (split form: K steps of 32 channels, never padded)
  // a grid that cannot fill the chip (batch-1 inference: 84 tiles for the 192 -> 64 decoder conv) stays with conv_dma, which
  // splits its K loop over workgroups there (561 fps at batch 1 with graph replay; 515 with the halo tiles)
  const HtShape h = ht_shape(d);
  const long tiles = (tm ? (long)d->x.B * vn_div_up(d->oT, 4) * vn_div_up((long)d->oH * d->oW, 64)
                         : (long)d->x.B * d->oT * vn_div_up(d->oH, 256 / h.tw) * vn_div_up(d->oW, h.tw)) * vn_div_up(N, h.nt * 16);
  return tiles >= 384 && (long)d->oH * d->oW >= (tm ? g_vinet_opt_ht_t_minhw : g_vinet_opt_ht_minhw) && N >= 32 && k64 * 20 <= d->Kp * 23;
}

// ---- split-K for grids that cannot fill the chip (batch-1 inference) ---------------------------
int g_vinet_opt_splitk = 1;     // 0 = off; n >= 2 = tuning: minimum K chunks (of 32) per split
int g_vinet_opt_sk_tile = 7;    // bit 0: 128 x 192 tiles, bit 1: 128 x 128 tiles for long-K small-grid convs (vinet_pick_conv_tile)
struct SplitK { int splits, per; long bytes; };
static SplitK splitk_plan(const VinetConvDesc* d, bool query) {
  SplitK p{1, 0, 0};
  if (!g_vinet_opt_splitk || !use_dma(d) || use_pp(d) || use_ht(d) || use_pw(d) || vinet_conv_use_ts(d) || vinet_conv_use_hs(d) || d->stats || d->accumulate || d->bnb_partials) return p;
  const long M = (long)d->x.B * d->oT * d->oH * d->oW;
  const int nchunks = d->ntaps * (d->Kp / 32);
  const ConvTile t = vinet_pick_conv_tile(d->dtype, d->mode, M, d->y.C, nchunks, vinet_conv_may_split(d, query));
  const long tiles = vn_div_up(M, t.BM()) * vn_div_up(d->y.C, t.BN());
  // batch-1 graph replay with the Inception branches side by side (three convs share the chip): 6: 663 fps, 8: 679, 12: 694,
  // 16: 702, 20: 688 (batch 4: 1265 / 1302 / 1313 / 1333 / 1348)
  const int min_per = g_vinet_opt_splitk >= 2 ? g_vinet_opt_splitk : 16;
  // workgroup slots of the chip for this tile shape: 3 stages of (BM + BN) rows x 64 B in 160 KB of LDS, 4 at most
  const long smem = 3L * (t.BM() + t.BN()) * 64 + (d->pre.scale ? 2L * d->Kp * 4 : 0);
  long per_cu = (160 * 1024) / smem;
  if (per_cu > 4) per_cu = 4;
  const long slots = 256 * per_cu;
  if (tiles * 2 > slots) return p;
  int s = (int)(slots / tiles);
  if (s > nchunks / min_per) s = nchunks / min_per;
  if (s > 16) s = 16;
  if (s < 2) return p;
  p.per = (nchunks + s - 1) / s;
  p.splits = (nchunks + p.per - 1) / p.per;
  p.bytes = (long)p.splits * M * d->y.C * 4;
  return p;
}

extern "C" int64_t vinet_conv3d_splitk_bytes(const VinetConvDesc* d) {
  if (!d) return 0;
  return splitk_plan(d, true).bytes;
}

// y = act(scale * sum_s ws[s][m][n] + shift) with the placement of the conv epilogue
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(const ConvArgs a) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)a.M * a.N) return;
  const int m = (int)(e / a.N), n = (int)(e - (long)m * a.N);
  float v = 0.f;
  for (int s = 0; s < a.splits; ++s) v += a.ws[((long)s * a.M + m) * a.N + n];
  const bool nok = n < a.Nw;
  v = fmaf(v, (a.out_scale && nok) ? a.out_scale[n] : 1.f, (a.out_shift && nok) ? a.out_shift[n] : 0.f);
  if (a.act == VINET_ACT_RELU) v = fmaxf(v, 0.f);
  else if (a.act == VINET_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
  long off;
  if (a.y_linear) {
    off = (long)m * a.ldy + n;
  } else {
    int b, to, ho, wo;
    decode_m(m, a.dW, a.dH, a.dT, b, to, ho, wo);
    off = (long)b * a.sBy + ((long)((to * a.omT + a.ooT) * a.yH + (ho * a.omH + a.ooH)) * a.yW + (wo * a.omW + a.ooW)) * (long)a.ldy + n;
  }
  if (a.out_f32) ((float*)a.y)[off] = v;
  else ((bf16_t*)a.y)[off] = f2bf(v);
}

// the same, four consecutive columns per thread (N % 4 == 0, 16-byte aligned rows of y): one 16-byte load per slab and one
// 8- / 16-byte store instead of four of each; the sums are formed in the same order (same bits)
__global__ __launch_bounds__(256) void conv_splitk_finish4_kernel(const ConvArgs a) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  const int n4 = a.N >> 2;
  if (e >= (long)a.M * n4) return;
  const int m = (int)(e / n4), n = (int)(e - (long)m * n4) * 4;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < a.splits; ++s) {
    const float4 q = *(const float4*)(a.ws + ((long)s * a.M + m) * a.N + n);
    v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const bool nok = n + r < a.Nw;
    v[r] = fmaf(v[r], (a.out_scale && nok) ? a.out_scale[n + r] : 1.f, (a.out_shift && nok) ? a.out_shift[n + r] : 0.f);
    if (a.act == VINET_ACT_RELU) v[r] = fmaxf(v[r], 0.f);
    else if (a.act == VINET_ACT_SIGMOID) v[r] = 1.f / (1.f + __expf(-v[r]));
  }
  long off;
  if (a.y_linear) {
    off = (long)m * a.ldy + n;
  } else {
    int b, to, ho, wo;
    decode_m(m, a.dW, a.dH, a.dT, b, to, ho, wo);
    off = (long)b * a.sBy + ((long)((to * a.omT + a.ooT) * a.yH + (ho * a.omH + a.ooH)) * a.yW + (wo * a.omW + a.ooW)) * (long)a.ldy + n;
  }
  if (a.out_f32) {
    *(float4*)((float*)a.y + off) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    union { bf16_t h[4]; uint2 u; } o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o.h[r] = f2bf(v[r]);
    *(uint2*)((bf16_t*)a.y + off) = o.u;
  }
}

extern "C" int vinet_conv3d_kernel_name(const VinetConvDesc* d, char* buf, int32_t n) {
  if (!d || !buf || n <= 0) return -1;
  const long M = (long)d->x.B * d->oT * d->oH * d->oW;
  const ConvTile t = vinet_pick_conv_tile(d->dtype, d->mode, M, d->y.C, (long)d->ntaps * (d->Kp / 32), vinet_conv_may_split(d, false));
  if (d->tline == 3) snprintf(buf, n, vinet_conv_use_tsd(d) ? "conv_tsd_kernel" : "(unsupported)");
  else if (vinet_conv_use_hs(d)) snprintf(buf, n, d->dtype == VINET_F32S ? "conv_hs3_kernel" : "conv_hs_kernel");
  else if (vinet_conv_use_ts(d)) snprintf(buf, n, d->dtype == VINET_F32S ? "conv_ts3_kernel<%s>" : "conv_ts_kernel<%s>", d->pre.scale ? "pre" : "plain");
  else if (use_pw(d)) snprintf(buf, n, "conv_pw_kernel<%d,%s>", pw_shape(d).nt * 16, d->pre.scale ? "pre" : "plain");
  else if (use_ht(d)) {
    const HtShape h = ht_shape(d);
    const char* fam = d->dtype == VINET_F32S ? "conv_ht3_kernel" : "conv_ht_kernel";     // (ht3: the split-bf16 instantiations)
    if (h.tm) snprintf(buf, n, "%s<%d,t,%s>", fam, h.nt * 16, h.pre ? "pre" : "plain");
    else snprintf(buf, n, h.pre ? "%s<%d,%d,pre>" : "%s<%d,%d>", fam, h.nt * 16, h.tw);
  }
  else if (use_pp(d)) snprintf(buf, n, "conv_pp_kernel<%d>", pp_bn(d->y.C));
  else if (use_dma(d)) snprintf(buf, n, "conv_dma_kernel<%d,%d,%d,%d,3,%s>", t.MT, t.NT, t.WM, t.WN, d->pre.scale ? "pre" : "plain");
  else if (use_dma3(d)) snprintf(buf, n, "conv_dma3_kernel<%d,3,%s>", dma3_nt(d) * 16, d->pre.scale ? "pre" : "plain");
  else snprintf(buf, n, "conv_igemm_kernel<%s,%d,%d,%d,%d,%d>", d->dtype == VINET_BF16 ? "bf16" : (d->dtype == VINET_F32S ? "float/split" : "float"), t.MT, t.NT, t.WM, t.WN, d->mode);
  return 0;
}

extern "C" int vinet_conv3d_applies_pre_once(const VinetConvDesc* d) {
  return d && d->pre.scale && !vinet_conv_use_ts(d) && !vinet_conv_use_hs(d) && use_ht(d) ? 1 : 0;
}

int vinet_conv_tsd_bnb_rows(const VinetConvDesc* d);
int g_vinet_opt_bnb_epi = 1;     // BatchNorm-backward partial sums out of the shared conv epilogue (0 = only the fused temporal data gradient)
// The shared epilogue (conv_igemm.h: conv_epilogue, bf16 fast path) forms the sums for any bf16 data gradient that covers y densely
// (one launch = the whole extent: unit output strides, no offsets) with whole 8-channel groups, in the kernels whose waves hold
// at most four row groups (conv_dma, conv_ht, the register-staged kernel); not the ping-pong kernel (register budget), the
// pointwise streaming kernel and the stem's streaming kernels (own epilogues), split-K launches or fp32 tensors.
int vinet_launch_conv_dma_bnb(const ConvTile& t, const ConvArgs& a, hipStream_t s);
int vinet_launch_conv_ht_bnb(int nt, int tw, int tm, const ConvArgs& a, hipStream_t s);
static bool bnb_epi_ok(const VinetConvDesc* d) {
  if (!g_vinet_opt_bnb_epi || !d->bnb_z || !d->bnb_mean || !d->bnb_invstd) return false;
  if (d->bnb_fwd.relu && !(d->bnb_fwd.scale && d->bnb_fwd.shift)) return false;
  if (d->dtype != VINET_BF16 || d->out_dtype != VINET_BF16 || d->mode != VINET_CONV_GENERIC) return false;
  if (d->stats || d->act != VINET_ACT_NONE || d->out_scale || d->out_shift || (d->n_valid > 0 && d->n_valid != d->y.C)) return false;
  if (d->omT != 1 || d->omH != 1 || d->omW != 1 || d->ooT || d->ooH || d->ooW || d->y.T != d->oT || d->y.H != d->oH || d->y.W != d->oW) return false;
  if (d->y.C % 8 || d->y.ld % 8 || d->y.sB % 8 || ((uintptr_t)d->y.ptr) % 16) return false;
  if (d->bnb_ld % 8 || d->bnb_sB % 8 || ((uintptr_t)d->bnb_z) % 16 || d->bnb_ld < d->y.C) return false;
  if (d->pre.scale || d->pre.relu || !use_dma(d)) return false;       // (data gradients read plain tensors)
  if (vinet_conv_use_hs(d) || vinet_conv_use_ts(d) || use_pw(d) || (!use_ht(d) && use_pp(d))) return false;
  return true;
}
extern "C" int vinet_conv3d_bn_bwd_stats_rows(const VinetConvDesc* d) {
  if (!d) return 0;
  if (d->tline == 3) return vinet_conv_tsd_bnb_rows(d);      // the fused temporal data gradient of the stem
  return bnb_epi_ok(d) ? vinet_conv3d_stats_rows(d) : 0;
}

extern "C" int vinet_conv3d_fuses_dgrad_phases(const VinetConvDesc* d) {
  return d && d->tline == 3 && vinet_conv_use_tsd(d) ? 1 : 0;
}

extern "C" int vinet_conv3d(const VinetConvDesc* d, void* stream) {
  if (d && d->tline == 3) {     // whole data gradient of a strided temporal conv: no tap table, x = dy, y = dx
    VN_CHECK_ARG(vinet_conv_use_tsd(d), "conv: tline == 3 (fused stride phases) is not available for this problem; ask vinet_conv3d_fuses_dgrad_phases first");
    return vinet_launch_conv_tsd(d, (hipStream_t)stream);
  }
  ConvArgs a;
  ConvTile t;
  int rc = fill_args(d, a, t);
  if (rc) return rc;
  if (d->bnb_partials) {
    VN_CHECK_ARG(bnb_epi_ok(d), "conv: the BatchNorm-backward statistics (bnb_*) are not available for this problem; ask vinet_conv3d_bn_bwd_stats_rows first");
    a.bnb_z = (const char*)d->bnb_z; a.bnb_ldz = d->bnb_ld; a.bnb_sBz = d->bnb_sB;
    a.bnb_z_linear = (d->bnb_sB == (int64_t)d->y.T * d->y.H * d->y.W * d->bnb_ld) ? 1 : 0;
    a.bnb_relu = d->bnb_fwd.relu; a.bnb_scale = d->bnb_fwd.scale; a.bnb_shift = d->bnb_fwd.shift;
    a.bnb_mean = d->bnb_mean; a.bnb_invstd = d->bnb_invstd; a.bnb_partials = d->bnb_partials;
  }
  if (vinet_conv_use_hs(d)) return vinet_launch_conv_hs(d, (hipStream_t)stream);
  if (vinet_conv_use_ts(d)) return vinet_launch_conv_ts(d, (hipStream_t)stream);
  if (use_pw(d)) {
    const PwShape h = pw_shape(d);
    a.tilesN = h.tilesN;
    a.tilesM = h.gm;
    a.chunks_per_split = h.tpw;
    return vinet_launch_conv_pw_bf16(h.nt, a, (hipStream_t)stream);
  }
  if (use_ht(d)) {
    const HtShape h = ht_shape(d);
    a.tilesN = vn_div_up(a.N, h.nt * 16);
    if (h.tm) {      // tiles: 64 positions x 4 output frames
      a.ht_tilesH = vn_div_up(d->oT, 4);
      a.ht_tilesW = vn_div_up((long)d->oH * d->oW, 64);
      a.tilesM = (int)((long)d->x.B * a.ht_tilesH * a.ht_tilesW);
      a.ht_dTo = make_fastdiv(1);
    } else {
      a.ht_tilesH = vn_div_up(d->oH, 256 / h.tw);
      a.ht_tilesW = vn_div_up(d->oW, h.tw);
      a.tilesM = (int)((long)d->x.B * d->oT * a.ht_tilesH * a.ht_tilesW);
      a.ht_dTo = make_fastdiv((uint32_t)d->oT);
    }
    a.ht_dN = make_fastdiv((uint32_t)a.tilesN); a.ht_dW = make_fastdiv((uint32_t)a.ht_tilesW);
    a.ht_dH = make_fastdiv((uint32_t)a.ht_tilesH);
    if (a.bnb_partials) return vinet_launch_conv_ht_bnb(h.nt, h.tw, h.tm, a, (hipStream_t)stream);
    return d->dtype == VINET_F32S ? vinet_launch_conv_ht_f32s(h.nt, h.tw, h.tm, h.pre, a, (hipStream_t)stream)
                                  : vinet_launch_conv_ht_bf16(h.nt, h.tw, h.tm, h.pre, a, (hipStream_t)stream);
  }
  if (use_pp(d)) {
    const int bn = pp_bn(a.N);
    a.tilesM = vn_div_up(a.M, 256);
    a.tilesN = vn_div_up(a.N, bn);
    return vinet_launch_conv_pp_bf16(bn, a, (hipStream_t)stream);
  }
  if (use_dma(d)) {
    const SplitK sp = splitk_plan(d, false);
    if (sp.splits > 1 && d->splitk_ws && d->splitk_ws_bytes >= sp.bytes) {
      a.splits = sp.splits; a.chunks_per_split = sp.per; a.ws = d->splitk_ws;
      rc = vinet_launch_conv_dma_bf16(t, a, (hipStream_t)stream);
      if (rc) return rc;
      const long n = (long)a.M * a.N;
      const int esz = a.out_f32 ? 4 : 2;
      const bool vec4 = (a.N & 3) == 0 && (a.ldy & 3) == 0 && ((uintptr_t)a.y % 16) == 0 && (((long)a.sBy * esz) % 16) == 0;
      if (vec4) hipLaunchKernelGGL(conv_splitk_finish4_kernel, dim3((unsigned)vn_div_up(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, a);
      else hipLaunchKernelGGL(conv_splitk_finish_kernel, dim3((unsigned)vn_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, a);
      return vn_launch_status("conv_splitk_finish");
    }
    if (a.bnb_partials) return vinet_launch_conv_dma_bnb(t, a, (hipStream_t)stream);
    return vinet_launch_conv_dma_bf16(t, a, (hipStream_t)stream);
  }
  if (use_dma3(d)) {
    const int nt = dma3_nt(d);
    a.tilesM = vn_div_up(a.M, 128);
    a.tilesN = vn_div_up(a.N, nt * 16);
    return vinet_launch_conv_dma3(nt, a, (hipStream_t)stream);
  }
  if (d->dtype == VINET_BF16) return vinet_launch_conv_bf16(t, d->mode, a, (hipStream_t)stream);
  return vinet_launch_conv_f32(t, d->mode, a, (hipStream_t)stream, d->dtype == VINET_F32S);
}
