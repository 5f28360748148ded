// Data-gradient kernels that ALSO write the BatchNorm-backward partial sums of the layer in front of them
// (VinetConvDesc::bnb_*; conv_igemm.h: conv_epilogue<..., BNB = true>): the plain-input shapes of conv_dma and conv_ht, in their
// own translation unit so that the forward / plain data-gradient instantiations keep their register allocation
// (train.py:193 -> model_utils.py:132,145,149: every BasicConv3d / SepConv3d BatchNorm runs in training mode, and its backward
// needs (sum g, sum g * xhat) over the whole tensor before the gradient can pass through it).
#include "conv_dma.h"
#include "conv_ht.h"

#define DMA_BNB_CASE(MT_, NT_, WM_, WN_)                                \
  if (t.MT == MT_ && t.NT == NT_ && t.WM == WM_ && t.WN == WN_)         \
    return launch_conv_dma_cfg<MT_, NT_, WM_, WN_, 3, false, true>(a, s);

int vinet_launch_conv_dma_bnb(const ConvTile& t, const ConvArgs& a, hipStream_t s) {
  DMA_BNB_CASE(4, 8, 4, 1) DMA_BNB_CASE(4, 6, 4, 1) DMA_BNB_CASE(4, 4, 4, 1) DMA_BNB_CASE(4, 3, 4, 1) DMA_BNB_CASE(4, 2, 4, 1) DMA_BNB_CASE(4, 1, 4, 1)
  DMA_BNB_CASE(4, 4, 2, 2) DMA_BNB_CASE(4, 2, 2, 2) DMA_BNB_CASE(2, 4, 2, 2) DMA_BNB_CASE(2, 2, 2, 2) DMA_BNB_CASE(4, 6, 2, 2)
  vinet_set_error("conv dma bf16 (bnb): no kernel for tile MT=%d NT=%d WM=%d WN=%d", t.MT, t.NT, t.WM, t.WN);
  return -1;
}

int vinet_launch_conv_ht_bnb(int nt, int tw, int tm, const ConvArgs& a, hipStream_t s) {
  if (tm) {
    if (nt == 4) return launch_conv_ht_cfg<4, 32, 3, true, false, false, true>(a, s);
    if (nt == 6) return launch_conv_ht_cfg<6, 32, 2, true, false, false, true>(a, s);
  } else if (tw == 32) {
    if (nt == 2) return launch_conv_ht_cfg<2, 32, 3, false, false, false, true>(a, s);
    if (nt == 4) return launch_conv_ht_cfg<4, 32, 3, false, false, false, true>(a, s);
    if (nt == 6) return launch_conv_ht_cfg<6, 32, 3, false, false, false, true>(a, s);
  } else if (tw == 16) {
    if (nt == 2) return launch_conv_ht_cfg<2, 16, 3, false, false, false, true>(a, s);
    if (nt == 4) return launch_conv_ht_cfg<4, 16, 3, false, false, false, true>(a, s);
    if (nt == 6) return launch_conv_ht_cfg<6, 16, 3, false, false, false, true>(a, s);
  }
  vinet_set_error("conv ht bf16 (bnb): no kernel for nt=%d tw=%d tm=%d", nt, tw, tm);
  return -1;
}
