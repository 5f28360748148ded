// fp32-tensor instantiations of the implicit-GEMM convolution: exact fp32 MFMA (parity path, VINET_F32) and the split-bf16
// form (VINET_F32S: three bf16 MFMAs per product on hi / lo halves of every operand, conv_igemm.h).
#include "conv_dma3.h"
#include "conv_ht.h"

#define CASE(MT_, NT_, WM_, WN_)                                                            \
  if (t.MT == MT_ && t.NT == NT_ && t.WM == WM_ && t.WN == WN_)                             \
    return split ? launch_conv_cfg<float, MT_, NT_, WM_, WN_, VINET_CONV_GENERIC, true>(a, s) \
                 : launch_conv_cfg<float, MT_, NT_, WM_, WN_, VINET_CONV_GENERIC>(a, s);

int vinet_launch_conv_f32(const ConvTile& t, int mode, const ConvArgs& a, hipStream_t s, bool split) {
  if (mode == VINET_CONV_STEM)
    return split ? launch_conv_cfg<float, 2, 4, 4, 1, VINET_CONV_STEM, true>(a, s) : launch_conv_cfg<float, 2, 4, 4, 1, VINET_CONV_STEM>(a, s);
  CASE(2, 8, 4, 1) CASE(2, 6, 4, 1) CASE(2, 4, 4, 1) CASE(2, 3, 4, 1) CASE(2, 2, 4, 1) CASE(2, 1, 4, 1)
  CASE(2, 2, 2, 2)
  vinet_set_error("conv f32: no kernel for tile MT=%d NT=%d WM=%d WN=%d", t.MT, t.NT, t.WM, t.WN);
  return -1;
}

// LDS-DMA form of the split-bf16 arithmetic (conv_dma3.h): 128 x 64 tiles (128 x 32 for narrow outputs)
int vinet_launch_conv_dma3(int nt, const ConvArgs& a, hipStream_t s) {
  if (nt == 2) return a.in_scale ? launch_conv_dma3_cfg<2, true>(a, s) : launch_conv_dma3_cfg<2, false>(a, s);
  return a.in_scale ? launch_conv_dma3_cfg<4, true>(a, s) : launch_conv_dma3_cfg<4, false>(a, s);
}

// halo-tile kernel in the split-bf16 form (conv_ht.h, SPLIT): 256 positions x 64 / 32 columns, K step = (tap, 32 channels)
int vinet_launch_conv_ht_f32s(int nt, int tw, int tm, int pre, const ConvArgs& a, hipStream_t s) {
  if (tm) {
    if (nt == 4) return pre ? launch_conv_ht_cfg<4, 32, 3, true, true, true>(a, s) : launch_conv_ht_cfg<4, 32, 3, true, false, true>(a, s);
    if (nt == 2) return pre ? launch_conv_ht_cfg<2, 32, 3, true, true, true>(a, s) : launch_conv_ht_cfg<2, 32, 3, true, false, true>(a, s);
  } else if (pre) {
    if (tw == 32 && nt == 4) return launch_conv_ht_cfg<4, 32, 3, false, true, true>(a, s);
    if (tw == 32 && nt == 2) return launch_conv_ht_cfg<2, 32, 3, false, true, true>(a, s);
    if (tw == 16 && nt == 4) return launch_conv_ht_cfg<4, 16, 3, false, true, true>(a, s);
    if (tw == 16 && nt == 2) return launch_conv_ht_cfg<2, 16, 3, false, true, true>(a, s);
  } else if (tw == 32) {
    if (nt == 2) return launch_conv_ht_cfg<2, 32, 3, false, false, true>(a, s);
    if (nt == 4) return launch_conv_ht_cfg<4, 32, 3, false, false, true>(a, s);
  } else if (tw == 16) {
    if (nt == 2) return launch_conv_ht_cfg<2, 16, 3, false, false, true>(a, s);
    if (nt == 4) return launch_conv_ht_cfg<4, 16, 3, false, false, true>(a, s);
  }
  vinet_set_error("conv ht f32s: no kernel for nt=%d tw=%d tm=%d pre=%d", nt, tw, tm, pre);
  return -1;
}
