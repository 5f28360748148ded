// bf16 instantiations of the implicit-GEMM convolution.
#include "conv_dma.h"
#include "conv_pp.h"
#include "conv_ht.h"
#include "conv_pw.h"

#define CASE(MT_, NT_, WM_, WN_)                                                            \
  if (t.MT == MT_ && t.NT == NT_ && t.WM == WM_ && t.WN == WN_)                             \
    return launch_conv_cfg<bf16_t, MT_, NT_, WM_, WN_, VINET_CONV_GENERIC>(a, s);

int vinet_launch_conv_bf16(const ConvTile& t, int mode, const ConvArgs& a, hipStream_t s) {
  if (mode == VINET_CONV_STEM) return launch_conv_cfg<bf16_t, 4, 4, 4, 1, VINET_CONV_STEM>(a, s);
  CASE(4, 8, 4, 1) CASE(4, 6, 4, 1) CASE(4, 4, 4, 1) CASE(4, 3, 4, 1) CASE(4, 2, 4, 1) CASE(4, 1, 4, 1)
  CASE(4, 4, 2, 2) CASE(4, 2, 2, 2) CASE(2, 4, 2, 2) CASE(2, 2, 2, 2) CASE(4, 6, 2, 2)
  vinet_set_error("conv bf16: no kernel for tile MT=%d NT=%d WM=%d WN=%d", t.MT, t.NT, t.WM, t.WN);
  return -1;
}

#define DMA_CASE(MT_, NT_, WM_, WN_)                                                   \
  if (t.MT == MT_ && t.NT == NT_ && t.WM == WM_ && t.WN == WN_)                        \
    return a.in_scale ? launch_conv_dma_cfg<MT_, NT_, WM_, WN_, 3, true>(a, s)          \
                      : launch_conv_dma_cfg<MT_, NT_, WM_, WN_, 3, false>(a, s);

// LDS-DMA pipelined kernel (conv_dma.h); a pending affine+ReLU is applied at fragment-read time
int vinet_launch_conv_dma_bf16(const ConvTile& t, const ConvArgs& a, hipStream_t s) {
  DMA_CASE(4, 8, 4, 1) DMA_CASE(4, 6, 4, 1) DMA_CASE(4, 4, 4, 1) DMA_CASE(4, 3, 4, 1) DMA_CASE(4, 2, 4, 1) DMA_CASE(4, 1, 4, 1)
  DMA_CASE(4, 4, 2, 2) DMA_CASE(4, 2, 2, 2) DMA_CASE(2, 4, 2, 2) DMA_CASE(2, 2, 2, 2) DMA_CASE(4, 6, 2, 2)
  vinet_set_error("conv dma bf16: no kernel for tile MT=%d NT=%d WM=%d WN=%d", t.MT, t.NT, t.WM, t.WN);
  return -1;
}

// 256x256x64 ping-pong kernel (conv_pp.h): plain inputs only
int vinet_launch_conv_pp_bf16(int bn, const ConvArgs& a, hipStream_t s) {
  return bn == 192 ? launch_conv_pp_cfg<4, 2, 192>(a, s) : launch_conv_pp_cfg<2, 4, 256>(a, s);
}

// halo-tile kernel (conv_ht.h): nt = 16-column tiles per workgroup, tw = tile width (spatial mode), tm = temporal mode
// ((3,1,1) taps), pre = pending BatchNorm + ReLU applied once per staged element
int vinet_launch_conv_ht_bf16(int nt, int tw, int tm, int pre, const ConvArgs& a, hipStream_t s) {
  if (tm) {
    if (nt == 4) return pre ? launch_conv_ht_cfg<4, 32, 3, true, true>(a, s) : launch_conv_ht_cfg<4, 32, 3, true, false>(a, s);
    if (nt == 6) return pre ? launch_conv_ht_cfg<6, 32, 2, true, true>(a, s) : launch_conv_ht_cfg<6, 32, 2, true, false>(a, s);
  } else if (pre) {
    if (tw == 32 && nt == 4) return launch_conv_ht_cfg<4, 32, 3, false, true>(a, s);
    if (tw == 32 && nt == 6) return launch_conv_ht_cfg<6, 32, 2, false, true>(a, s);
    if (tw == 16 && nt == 4) return launch_conv_ht_cfg<4, 16, 3, false, true>(a, s);
    if (tw == 16 && nt == 6) return launch_conv_ht_cfg<6, 16, 2, false, true>(a, s);
  } else if (tw == 32) {
    if (nt == 2) return launch_conv_ht_cfg<2, 32, 3>(a, s);
    if (nt == 4) return launch_conv_ht_cfg<4, 32, 3>(a, s);
    if (nt == 6) return launch_conv_ht_cfg<6, 32, 3>(a, s);
  } else if (tw == 16) {
    if (nt == 2) return launch_conv_ht_cfg<2, 16, 3>(a, s);
    if (nt == 4) return launch_conv_ht_cfg<4, 16, 3>(a, s);
    if (nt == 6) return launch_conv_ht_cfg<6, 16, 3>(a, s);
  }
  vinet_set_error("conv ht bf16: no kernel for nt=%d tw=%d tm=%d pre=%d", nt, tw, tm, pre);
  return -1;
}

// pointwise streaming kernel (conv_pw.h): nt = 16-column tiles of the resident weight tile
int vinet_launch_conv_pw_bf16(int nt, const ConvArgs& a, hipStream_t s) {
  const bool pre = a.in_scale != nullptr;
  if (nt == 2) return pre ? launch_conv_pw_cfg<2, true>(a, s) : launch_conv_pw_cfg<2, false>(a, s);
  if (nt == 4) return pre ? launch_conv_pw_cfg<4, true>(a, s) : launch_conv_pw_cfg<4, false>(a, s);
  if (nt == 6) return pre ? launch_conv_pw_cfg<6, true>(a, s) : launch_conv_pw_cfg<6, false>(a, s);
  vinet_set_error("conv pw bf16: no kernel for nt=%d", nt);
  return -1;
}
