/*
 * vinet_hip.h -- C ABI of libvinet_hip.so: the MI355X (gfx950) kernels behind
 * the ViNet / AViNet saliency hot path.
 *
 * The reference (samyak0210/ViNet) has no FFI seam of its own: its hot path is
 * `nn.Module`s calling aten.  The seam this library fills is therefore "the
 * aten ops that model_utils.py / model.py / loss.py execute", one entry point
 * per kernel family x direction.  Each declaration cites the reference call
 * site(s) (file:line under /root/reference) whose device work it replaces.
 *
 * Conventions
 *  - plain C, POD structs, no torch types.  All pointers are DEVICE pointers
 *    unless named `host_*`.
 *  - the caller owns every buffer (inputs, outputs, workspaces).  The library
 *    allocates nothing, creates no streams and never synchronises: work is
 *    enqueued on the `hipStream_t` passed in (as `void*`), on the device that
 *    is current for the calling thread.
 *  - re-entrant; safe to call from several host threads (one per device /
 *    autograd worker threads).
 *  - return value: 0 = ok; negative = invalid argument (see
 *    vinet_last_error()); positive = hipError_t from the launch.
 *  - activations are CHANNELS-LAST 5-D views `[B][T][H][W][C]` (the torch
 *    `channels_last_3d` memory format of a logical NCDHW tensor), described by
 *    VinetTensor: W-stride `ld` elements (>= C, so a view may be a channel
 *    slice of a wider concat buffer), H-stride `W*ld`, T-stride `H*W*ld`,
 *    batch stride `sB` (so a view may also be a T slice of a longer buffer).
 *  - dtype: VINET_F32 (parity path) or VINET_BF16 (throughput path); all
 *    accumulation, BN statistics and loss arithmetic are fp32/fp64.  Conv and
 *    weight-gradient descriptors (and vinet_pack_weights) also take VINET_F32S:
 *    fp32 tensors, bf16 matrix arithmetic on hi / lo halves of both operands.
 */
#ifndef VINET_HIP_H
#define VINET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VINET_ABI_VERSION 13   /* 13 (round 6): tline 5 / 1 also promise weight slices < 64; entry points unchanged */

enum { VINET_F32 = 0, VINET_BF16 = 1,
       /* conv / weight-gradient descriptors only: fp32 tensors (as VINET_F32), bf16 matrix arithmetic on a two-term split of both
        * operands (x = hi + lo, 16 significant bits; hi*hi + hi*lo + lo*hi, fp32 accumulate) -- 3/16 of the fp32-MFMA cost, error
        * ~2^-17 per operand.  The fast configuration that still meets north_star's 1e-3 / exact-argmax contract. */
       VINET_F32S = 2 };
enum { VINET_ACT_NONE = 0, VINET_ACT_RELU = 1, VINET_ACT_SIGMOID = 2 };
enum { VINET_CONV_GENERIC = 0, VINET_CONV_STEM = 1 };

typedef struct VinetTensor {
  void* ptr;      /* element [0,0,0,0,0] of the view */
  int32_t B, T, H, W, C;
  int32_t ld;     /* elements between consecutive W positions */
  int64_t sB;     /* elements between consecutive batch items */
} VinetTensor;

/* A per-channel affine (+ReLU) applied to a tensor WHILE IT IS LOADED: the
 * consumer-side form of BatchNorm3d(+ReLU) in training mode, where the
 * producer conv stores its raw output and the statistics only exist after the
 * whole tensor has been written (model_utils.py:132-133,145-146,149-150).
 * scale == NULL means identity. */
typedef struct VinetAffine {
  const float* scale;
  const float* shift;
  int32_t relu;
} VinetAffine;

/* ------------------------------------------------------------------------
 * Convolution as implicit GEMM on MFMA.
 * Replaces nn.Conv3d forward and its two backward halves wherever the
 * reference calls them: model_utils.py:131,144,148 (BasicConv3d / SepConv3d),
 * model.py:256,261,266,271,275,280,282 (decoder), and, with T as the 1-D axis,
 * nn.Conv2d (k,1) of SoundNet model.py:750-791.
 *
 *   y[b, (to,ho,wo) -> storage pos, n] (+)= act( out_scale[n] * sum_{tap,c}
 *        pre(x[b, to*sT+dt_tap, ho*sH+dh_tap, wo*sW+dw_tap, c]) * w[slice_tap][n][c] + out_shift[n] )
 *
 *  - `taps`: ntaps x int32[4] = (dt, dh, dw, weight slice); padding is folded
 *    into the offsets (dt = kt - padT ...), out-of-range reads are zero.  A
 *    transposed convolution (dgrad of a strided conv) is expressed by the
 *    caller as one launch per stride phase with that phase's tap subset.
 *  - `w`: packed [nslices][N][Kp], Kp = Cin rounded up to 32, same dtype as x
 *    (see vinet_pack_weights).
 *  - output storage position = (to*omT+ooT, ho*omH+ooH, wo*omW+ooW) inside
 *    `y`, so producers write straight into concat buffers / phase-strided
 *    gradients (torch.cat at model_utils.py:187, model.py:290,296,302 is never
 *    materialised as a copy).
 *  - `stats` (optional): per-M-tile partial sums of the pre-activation values,
 *    layout [tilesM][2][N] fp32 (sum, sum of squares), tilesM =
 *    vinet_conv3d_stats_rows(desc); reduced by vinet_bn_finalize.
 *  - mode VINET_CONV_STEM: x has C == 4 (3 + zero pad), taps index kernel
 *    rows only and each K chunk of 32 is 8 consecutive W positions x 4
 *    channels (the 1x7x7 stride-2 stem, model.py:693 / model_utils.py:144).
 * ---------------------------------------------------------------------- */
typedef struct VinetConvDesc {
  int32_t dtype;        /* of x and w */
  int32_t out_dtype;    /* of y */
  int32_t mode;         /* VINET_CONV_GENERIC / VINET_CONV_STEM */
  VinetTensor x;        /* C = Cin (multiple of 16 bytes worth of elements) */
  VinetTensor y;        /* output storage, C = N */
  int32_t oT, oH, oW;   /* iteration space; M = B*oT*oH*oW */
  int32_t sT, sH, sW;
  int32_t omT, omH, omW, ooT, ooH, ooW;
  int32_t ntaps;
  const int32_t* taps;
  const void* w;
  int32_t Kp;
  VinetAffine pre;      /* applied to x on load */
  const float* out_scale;
  const float* out_shift;
  int32_t act;
  int32_t accumulate;   /* y += ... */
  float* stats;
  int32_t n_valid;      /* 0 = y.C; else only channels < n_valid have weights /
                           affine (the rest of a channel-padded y gets act(0)) */
  float* splitk_ws;     /* optional fp32 scratch for split-K (low-M, long-K convs: batch-1 inference): at least
                           vinet_conv3d_splitk_bytes(desc) bytes, contents irrelevant on entry and undefined
                           on return; NULL (or too small) = never split */
  int64_t splitk_ws_bytes;
  int32_t tline;        /* 1 = the caller promises a purely temporal kernel: every tap is (dt, 0, 0, slice) and the
                           dt form the contiguous range [-tpad, -tpad + ntaps - 1] in any order (the tap table is
                           device memory, the library cannot look); lets 64 -> 64 channel layers take the
                           frame-streaming kernel (conv_ts.hip).
                           2 = the taps are (0, kh, 0, slice kh), kh = 0..6: the folded RGB stem (row-streaming strip
                           kernel, conv_hs.hip).
                           3 = not a tap promise but a different problem: the WHOLE data gradient of a strided
                           temporal conv in one launch (instead of one launch per stride phase): x = dy, y = dx,
                           w = the transposed pack, ntaps / sT / tpad = kernel length, stride and padding of the
                           FORWARD conv, oT = y.T; `taps` is ignored.  Only where
                           vinet_conv3d_fuses_dgrad_phases(desc) returns 1.
                           5 = 3 x 3 spatial footprint: every tap is (dt, dh, dw, slice) with |dh| <= 1 and |dw| <= 1, and
                           taps of equal dt are contiguous in the table (ConvPlan.fwd_taps order and every stride phase of
                           its data gradient): lets plain-input layers take the halo-tile kernel (conv_ht.h), which
                           stages the activation patch once per (dt, 64 channels) instead of once per tap.  The weight
                           slice of a tap must be < 64 (six bits of the kernel's packed tap word; the net's largest
                           kernel, 5 x 3 x 3, has 45).  tline 1 (temporal line) carries the same promise for conv_ht's
                           temporal mode.
                           6 = pointwise: ntaps == 1 and the tap is (0, 0, 0, slice 0) (a 1x1x1 / stride-1 conv or its data
                           gradient): lets large problems take the wave-streaming kernel (conv_pw.h), which keeps the
                           weight tile in LDS and feeds the activations to the matrix cores straight from registers.
                           0 = no promise. */
  int32_t tpad;
  /* Optional, data gradients only (ABI 11): y is the gradient g behind a BatchNorm (+ ReLU) whose RAW input is bnb_z -- same
   * extent as y, rows of bnb_ld elements, clips bnb_sB elements apart, dtype of y; bnb_fwd = the forward affine of that
   * BatchNorm (scale, shift, relu: the ReLU gate is scale * z + shift > 0), bnb_mean / bnb_invstd its batch statistics.
   * With bnb_partials != NULL the launch ALSO writes the partial sums of vinet_bn_bwd_reduce(g, z) -- [rows][2][C] fp32,
   * rows = vinet_conv3d_bn_bwd_stats_rows(desc) -- so the caller skips that pass (train.py:193 -> model_utils.py:145: the
   * stem's first BatchNorm, whose gradient comes out of the fused temporal data gradient, tline == 3; ABI 12: every BatchNorm
   * whose output gradient is last written by a data gradient of the shared conv epilogue -- model_utils.py:132,145,149).  The sums
   * are formed on the values the launch STORES (rounded to y's dtype; with accumulate != 0 on old + new, so the caller must make
   * this launch the last writer of y).  Only where the rows query returns > 0.  tline == 3: accumulate must be 0. */
  const void* bnb_z;
  int32_t bnb_ld;
  int64_t bnb_sB;
  VinetAffine bnb_fwd;
  const float* bnb_mean;
  const float* bnb_invstd;
  float* bnb_partials;
} VinetConvDesc;

int vinet_conv3d(const VinetConvDesc* desc, void* stream);
/* Bytes of fp32 scratch with which vinet_conv3d would split this problem's K loop over several workgroups
 * (each split stores its partial sums to its own slab; a finishing pass adds the slabs in a fixed order, so the
 * result is run-to-run deterministic, and applies the epilogue); 0 when it would not split (enough tiles,
 * short K, statistics or accumulate requested). */
int64_t vinet_conv3d_splitk_bytes(const VinetConvDesc* desc);
/* 1 if vinet_conv3d accepts this tline == 3 descriptor (fused stride phases of a temporal data gradient). */
int vinet_conv3d_fuses_dgrad_phases(const VinetConvDesc* desc);
/* Rows of BatchNorm-backward partial sums ([rows][2][C] fp32) that vinet_conv3d writes for this descriptor when bnb_partials is
 * set (bnb_z, bnb_ld, bnb_sB, bnb_fwd, bnb_mean, bnb_invstd filled in; bnb_partials itself is not looked at); 0 = the kernel
 * this problem takes cannot fold the reduce pass in: leave bnb_partials NULL and call vinet_bn_bwd_reduce. */
int vinet_conv3d_bn_bwd_stats_rows(const VinetConvDesc* desc);
/* 1 if the kernel vinet_conv3d picks for this problem applies the pending affine `pre` ONCE per staged activation (the
 * halo-tile kernel transforms its LDS image in place) rather than at every fragment read: callers that would otherwise
 * materialise relu(bn(x)) with vinet_copy_affine first can skip that pass. */
int vinet_conv3d_applies_pre_once(const VinetConvDesc* desc);
/* BM of the tile configuration vinet_conv3d will pick for this problem. */
int vinet_conv3d_tile_m(const VinetConvDesc* desc);
/* Rows of the `stats` table ([rows][2][N]) the launch fills: ceil(M / tile_m) for the row-tiled kernels, the number
 * of spatial tiles for the halo-tile kernel (partial tiles at the image border).  Size `stats` with this. */
int vinet_conv3d_stats_rows(const VinetConvDesc* desc);
/* Name of the kernel instantiation vinet_conv3d will launch for this problem
 * (profilers report kernels by that name). */
int vinet_conv3d_kernel_name(const VinetConvDesc* desc, char* buf, int32_t n);

/* Weight gradient: dw[slice_tap][n][c] (+)= sum_m dy[m][n] * pre(x[m shifted by tap])[c]
 * (the wgrad half of convolution_backward, train.py:216).  `dw` is fp32
 * [nslices][N][Kp] and MUST be zero on entry (split-K partial products are
 * accumulated with fp32 atomics); vinet_unpack_wgrad converts it to the
 * torch [N][Cin][kT][kH][kW] layout. */
typedef struct VinetWgradDesc {
  int32_t dtype;
  int32_t mode;
  VinetTensor x;
  VinetTensor dy;       /* [B][oT][oH][oW][N] */
  int32_t sT, sH, sW;
  int32_t ntaps;
  const int32_t* taps;
  float* dw;
  int32_t Kp;
  VinetAffine pre;
  int32_t tline;        /* 1 = the caller promises a purely temporal kernel: tap kt is (kt - tpad, 0, 0, slice kt),
                           kt = 0..ntaps-1 (the tap table lives in device memory, the library cannot look);
                           lets 64 -> 64 channel layers take the frame-streaming kernel.
                           2 = the taps are (0, kh, 0, slice kh), kh = 0..6: the folded RGB stem (row-streaming strip
                           kernel).
                           4 = the taps are (kt, kh-1, kw-1, slice (kt*3 + kh)*3 + kw), kt < ntaps / 9: a kT x 3 x 3
                           kernel with stride (kT,1,1), padding (0,1,1) (the decoder convs; row-streaming kernel for
                           64 output channels).  0 = no promise. */
  int32_t tpad;
  /* Optional fused BatchNorm(+ReLU) backward.  With bnb_z != NULL, `dy` is the gradient w.r.t. the OUTPUT of the
   * BatchNorm that follows this conv, bnb_z the raw conv output (same B/T/H/W/C as dy), and the kernel forms
   *   dz = scale * (dy * mask - c1 - (z - mean) * invstd * c2)        (vinet_bn_bwd_apply's arithmetic, rounded to
   * the activation dtype) on the fly instead of reading a stored dz: for a conv whose input needs no gradient (the
   * RGB stem) dz has no other consumer, so the apply pass (2 reads + 1 write of the layer's largest tensor) and
   * the wgrad's own read of dz become 2 reads.  Only kernels that say so support it: ask
   * vinet_conv3d_wgrad_fuses_bn_bwd(desc) first; vinet_conv3d_wgrad rejects the descriptor otherwise. */
  const void* bnb_z;
  int32_t bnb_ld;
  int64_t bnb_sB;
  VinetAffine bnb_fwd;           /* forward scale / shift / relu of that BatchNorm */
  const float* bnb_mean;
  const float* bnb_invstd;
  const float* bnb_c1;
  const float* bnb_c2;
  /* Compute units the PERSISTENT weight-gradient kernels (one 512-thread workgroup per CU: the row- and frame-streaming
   * forms) may occupy for this launch: a caller that runs its weight gradients on a second stream beside the data-gradient
   * chain leaves the rest of the chip to that chain.  0 = the whole chip (256).  Per launch and per descriptor, so that
   * concurrent callers (one host thread per device, autograd worker threads) never share mutable state. */
  int32_t max_cus;
} VinetWgradDesc;

int vinet_conv3d_wgrad(const VinetWgradDesc* desc, void* stream);
/* 1 if vinet_conv3d_wgrad would apply desc->bnb_* itself for this problem (fill the bnb_* fields before asking). */
int vinet_conv3d_wgrad_fuses_bn_bwd(const VinetWgradDesc* desc);
int vinet_conv3d_wgrad_kernel_name(const VinetWgradDesc* desc, char* buf, int32_t n);

/* fp32 torch-layout master weights [N][Cin][ntaps] -> packed compute weights.
 *  transpose == 0: out[t][n][c]       (Kp = pad32(Cin))   forward / wgrad layout
 *  transpose == 1: out[t][c][n]       (Kp = pad32(N))     dgrad layout
 *  stem      == 1: out[kh][n][kw*4+c] (Kp = 32; ntaps = 7*7, Cin = 3)
 * dtype VINET_F32S (the packs VINET_F32S descriptors read): the same logical arrays, stored as hi / lo bf16 planes -- every
 * 32-wide K chunk of a row becomes 128 bytes [32 bf16 hi | 32 bf16 lo], hi = bf16(w), lo = bf16(w - hi), K positions permuted so
 * that positions 8q .. 8q+7 hold channels {4q .. 4q+3, 16+4q .. 16+4q+3} (the order in which the kernels' lane groups read fp32
 * activations in 16-byte pieces).  Same byte count as the fp32 pack; callers size the buffer as for VINET_F32.
 * Replaces the implicit weight reads of every nn.Conv3d above. */
int vinet_pack_weights(const float* w, int32_t N, int32_t Cin, int32_t ntaps, int32_t transpose, int32_t stem,
                       int32_t dtype, void* out, void* stream);

/* Multi-tensor vinet_pack_weights: one launch for every weight of the model (they all go stale together
 * after the optimizer step; replaces ~170 launches of model_utils.py conv weights per training step).
 * `table` is DEVICE memory: (njobs + 1) rows of 8 int64
 *   { w (const float*), out (packed, dtype), N, Cin, ntaps, transpose | stem << 1, first output index, ld | col << 32 }
 * the last row carrying only the total in its prefix field; layouts exactly as vinet_pack_weights.  A transposed
 * job with ld != 0 writes its N columns at [col, col + N) of rows `ld` elements apart (padding columns are the
 * caller's to zero): the 1x1x1 convs of an Inception block that share an input are packed side by side along K
 * so that their data gradients are ONE conv (model_utils.py:176-187). */
int vinet_pack_weights_multi(const int64_t* table, int32_t njobs, int64_t total, int32_t dtype, void* stream);
/* packed fp32 dw -> torch layout; grad (+)= dw.  flags: bit 0 = accumulate into grad (else store), bit 1 = hand
 * `dw` back ZEROED (rows [0,N) of every slice, padding columns included), so a caller-owned persistent workspace
 * meets vinet_conv3d_wgrad's "zero on entry" contract at the next step without a fill launch. */
int vinet_unpack_wgrad(float* dw, int32_t N, int32_t Cin, int32_t ntaps, int32_t stem, int32_t flags,
                       float* grad, void* stream);
/* Multi-tensor vinet_unpack_wgrad: the weight gradients of a whole backward pass (train.py:216; 84 Conv3d weights of
 * ViNet-32, model.py / model_utils.py) handed to their `.grad` tensors in one launch.  `table` is DEVICE memory:
 * (njobs + 1) rows of 8 int64 { dw (packed fp32), grad (torch layout fp32), N, Cin, ntaps, stem, first PACKED index, 0 },
 * the last row carrying only the total number of packed elements in its prefix field (a job has nsl * N * Kp of them:
 * nsl = 7 / Kp = 32 for the stem, ntaps / rup(Cin, 32) otherwise).  `flags` as vinet_unpack_wgrad, for every job. */
int vinet_unpack_wgrad_multi(const int64_t* table, int32_t njobs, int64_t total, int32_t flags, void* stream);

/* ------------------------------------------------------------------------
 * Layout / dtype conversion at the module boundary.
 * NCDHW-logical fp32 tensor with arbitrary element strides (train.py:205
 * hands over a permuted view) -> channels-last, channel-padded `dst`
 * (dst.C >= C, pad channels zeroed); and back.
 * ---------------------------------------------------------------------- */
int vinet_import_ncdhw(const float* src, int64_t sb, int64_t sc, int64_t st, int64_t sh, int64_t sw, int32_t C,
                       const VinetTensor* dst, int32_t dst_dtype, void* stream);
/* Same, into a zero-padded buffer: dst voxel (h, w) <- src(h - pad_top, w - pad_left), zeros
 * elsewhere (src is [.., Hs, Ws]).  Used for the RGB stem: with 3 rows / 3 pixels of padding the
 * 1x7x7 stride-2 conv (model_utils.py:144 via model.py:693) reads, for every output and kernel
 * row, 8 pixels x 4 channels = 64 contiguous, 16-byte aligned, always in-bounds bytes, i.e. it is
 * a generic 7-tap conv over the overlapped view [B][T][Hs+6][(Ws+8)/2][C=32], ld = 8. */
int vinet_import_ncdhw_pad(const float* src, int64_t sb, int64_t sc, int64_t st, int64_t sh, int64_t sw, int32_t C,
                           int32_t Hs, int32_t Ws, int32_t pad_top, int32_t pad_left, const VinetTensor* dst,
                           int32_t dst_dtype, void* stream);
/* dst[b,c,t,h,w] (fp32, strides given) = pre(src); accumulate adds. */
int vinet_export_ncdhw(const VinetTensor* src, int32_t src_dtype, VinetAffine pre, float* dst, int64_t sb, int64_t sc,
                       int64_t st, int64_t sh, int64_t sw, int32_t accumulate, void* stream);
/* channels-last -> channels-last copy with optional affine+relu (materialises
 * a pending BN, writes a skip connection into a T-concat buffer, converts
 * dtype).  dst (+)= pre(src). */
int vinet_copy_affine(const VinetTensor* src, int32_t src_dtype, VinetAffine pre, const VinetTensor* dst,
                      int32_t dst_dtype, int32_t accumulate, void* stream);

/* ------------------------------------------------------------------------
 * BatchNorm3d / BatchNorm2d (model_utils.py:132,145,149; model.py:752-786).
 * ---------------------------------------------------------------------- */
/* Training: reduce conv-epilogue partials [rows][2][C] -> batch mean / biased
 * var; write mean, invstd, and the consumer-side affine scale = gamma*invstd,
 * shift = beta - mean*scale; update running stats with `momentum` (unbiased
 * variance), as aten native_batch_norm does. */
/* `ld` (0 = C): row stride of the partials when the C channels are a slice of a wider epilogue's columns. */
int vinet_bn_finalize(const float* partials, int32_t rows, int32_t C, int32_t ld, double count, const float* gamma,
                      const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                      float* mean, float* invstd, float* scale, float* shift, void* stream);
/* Optional pre-reduction of a tall partials table [rows][2][C] into out[out_rows][2][C] (chunks of
 * ceil(rows / out_rows) consecutive rows; (out_rows - 1) * chunk < rows required), coalesced: the finalize kernels
 * walk the rows with one workgroup per channel, which is slow for the 10^5 rows of the 64-channel stem layers. */
int vinet_bn_partials_fold(const float* partials, int32_t rows, int32_t C, float* out, int32_t out_rows, void* stream);
/* Eval: scale = gamma / sqrt(running_var + eps), shift = beta + (conv_bias - running_mean)*scale. */
int vinet_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                  const float* conv_bias /* optional */, float eps, int32_t C, float* scale, float* shift,
                  float* invstd /* optional */, void* stream);
/* Generic per-channel statistics of a tensor (used when the producer is not a
 * conv epilogue): partials [rows][2][C]; returns rows via vinet_stats_rows. */
int vinet_channel_stats(const VinetTensor* x, int32_t dtype, float* partials, void* stream);
int vinet_stats_rows(const VinetTensor* x);
/* Backward of y = relu?(scale*x_raw + shift) followed by BN-train statistics:
 *  pass 1: partials[rows][2][C] of (sum dz*mask, sum dz*mask*xhat)
 *  pass 2: dx_raw = scale*(dz*mask - c1 - xhat*c2)   (c1 = c2 = 0 in eval mode)
 * native_batch_norm_backward + threshold_backward (train.py:216). */
int vinet_bn_bwd_reduce(const VinetTensor* dz, const VinetTensor* x_raw, int32_t dtype, VinetAffine fwd,
                        const float* mean, const float* invstd, float* partials, void* stream);
int vinet_bn_bwd_finalize(const float* partials, int32_t rows, int32_t C, int32_t ld, double count, const float* scale,
                          int32_t train, float* dgamma_acc, float* dbeta_acc, const float* invstd, float* c1, float* c2,
                          void* stream);
int vinet_bn_bwd_apply(const VinetTensor* dz, const VinetTensor* x_raw, int32_t dtype, VinetAffine fwd,
                       const float* mean, const float* invstd, const float* c1, const float* c2,
                       const VinetTensor* dx, void* stream);
/* vinet_bn_bwd_apply on fp32 tensors + vinet_split_bf16 of its result in the same pass: dx as usual and its hi / lo bf16 planes
 * (hi = bf16(v), lo = bf16(v - hi)) -- the operand planes of the VINET_F32S weight gradient (three bf16 launches).  C % 8 == 0. */
int vinet_bn_bwd_apply_split(const VinetTensor* dz, const VinetTensor* x_raw, VinetAffine fwd, const float* mean,
                             const float* invstd, const float* c1, const float* c2, const VinetTensor* dx,
                             const VinetTensor* hi, const VinetTensor* lo, void* stream);
/* dy = dz * act'(z) for z = relu(.) or sigmoid(.) outputs (threshold_backward /
 * sigmoid_backward of the decoder, model.py:257-283).  z may be fp32. */
int vinet_act_bwd(const VinetTensor* dz, int32_t dz_dtype, const VinetTensor* z, int32_t z_dtype, int32_t act,
                  const VinetTensor* dy, int32_t dy_dtype, void* stream);
/* per-channel sum over all voxels (conv bias gradients):
 * out[j] (+)= sum over voxels and over channels c with c % Cout == j.
 * workspace: fp32 [vinet_stats_rows(x)][2][x.C]. */
int vinet_channel_sum(const VinetTensor* x, int32_t dtype, float* workspace, int32_t Cout, float* out,
                      int32_t accumulate, void* stream);

/* ------------------------------------------------------------------------
 * MaxPool3d (model.py:696,700,705,713,714, model_utils.py:178, model.py:229;
 * MaxPool2d (k,1) of SoundNet model.py:753,759,774 with T as the pooled axis).
 * -inf padding, first maximum in (t,h,w) scan order wins ties.  `argmax`
 * (optional, uint8 [B][oT][oH][oW][C] dense) holds the window-relative tap
 * index for the backward gather.
 * ---------------------------------------------------------------------- */
typedef struct VinetPoolDesc {
  int32_t dtype;
  int32_t kT, kH, kW, sT, sH, sW, pT, pH, pW;
} VinetPoolDesc;
int vinet_maxpool3d(const VinetPoolDesc* d, const VinetTensor* x, VinetAffine pre, const VinetTensor* y,
                    uint8_t* argmax, void* stream);
int vinet_maxpool3d_bwd(const VinetPoolDesc* d, const VinetTensor* dy, const uint8_t* argmax, const VinetTensor* dx,
                        int32_t accumulate, void* stream);

/* nn.Upsample(scale_factor=(1,2,2), mode='trilinear', align_corners=False)
 * (model.py:254,258,263,268,273,278) and its backward. */
int vinet_upsample2x(const VinetTensor* x, const VinetTensor* y, int32_t dtype, void* stream);
int vinet_upsample2x_bwd(const VinetTensor* dy, const VinetTensor* dx, int32_t dtype, int32_t accumulate, void* stream);
/* The same with the backward of the ReLU in FRONT of the upsample folded in (the decoder's conv -> ReLU -> upsample,
 * model.py:256-258): dx = [xf > 0] * upsample2x^T(dy), xf = the ReLU's output (the upsample's input, dx's extent).  Stores. */
int vinet_upsample2x_bwd_relu(const VinetTensor* dy, const VinetTensor* dx, const VinetTensor* xf, int32_t dtype, void* stream);

/* SoundNet's first conv (model.py:751: Conv2d(1, 16, (64,1), stride (2,1), padding (32,0))) as a pointwise conv over the
 * unfolded waveform: y[b, m, 0, 0, c] = x[b, stride*m - pad + c, 0, 0, channel 0], zero outside; x = [B][L][1][1][C>=1],
 * y = [B][(L + 2 pad - k)/stride + 1][1][1][k], k = y.C (a multiple of 8).  The conv's weight [N][1][k][1] is, flat, the
 * pointwise weight [N][k]. */
int vinet_unfold1d(const VinetTensor* x, const VinetTensor* y, int32_t dtype, int32_t stride, int32_t pad, void* stream);

/* ------------------------------------------------------------------------
 * Losses (loss.py:13-99): per-sample reductions in fp64, maps fp32 (or fp64
 * ground truth).  `which`: 0 kldiv, 1 cc, 2 similarity, 3 nss (loss.py:101-120, the
 * validation metric; forward only, vinet_loss_bwd rejects it).
 *  fwd: per_sample[b] and the batch mean in *loss (fp32, device);
 *       `saved` (fp64 [B][8]) keeps the per-sample reductions for backward.
 *  bwd: ds[b,i] (+)= gscale * dloss/ds.
 * ---------------------------------------------------------------------- */
int vinet_loss_fwd(int32_t which, const float* s, const void* gt, int32_t gt_is_f64, int32_t B, int32_t n,
                   double* saved, float* loss, void* stream);
int vinet_loss_bwd(int32_t which, const float* s, const void* gt, int32_t gt_is_f64, int32_t B, int32_t n,
                   const double* saved, const float* gscale, float coeff, int32_t accumulate, float* ds, void* stream);

/* torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8) over one flat fp32 buffer
 * (train.py:188,217). bias corrections are passed by the host. */
int vinet_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                    float eps, float bias_c1, float bias_c2, float grad_scale, void* stream);

/* nn.Bilinear(42, 3, 336) fusion (model.py:230,236):
 *  out[b, o, c] = sum_ij x1[b, i, c] * w[o][i][j] * x2[b, j, c] + bias[o]
 * with x1 = [B][I][C], x2 = [B][J][C], out = [B][O][C] channels-last. */
int vinet_bilinear_fwd(const void* x1, const void* x2, int32_t dtype, const float* w, const float* bias, int32_t B,
                       int32_t C, int32_t I, int32_t J, int32_t O, void* out, void* stream);
int vinet_bilinear_bwd(const void* x1, const void* x2, const void* dout, int32_t dtype, const float* w, int32_t B,
                       int32_t C, int32_t I, int32_t J, int32_t O, void* dx1, void* dx2, float* dw, float* dbias,
                       void* stream);

/* ------------------------------------------------------------------------
 * Saliency-map post-processing: the host-side cv2 / torchvision steps of
 * generate_result.py:95-104 process() and train.py:251-253 validate(), on device.  Maps are dense float32
 * [B][H][W]; arithmetic follows opencv-python 3.4.3 / torchvision 0.5.0 (requirements.txt:96,178) as restated in
 * oracle/postproc_cpu.py, float32 without contraction.
 *
 * vinet_resize_blur: dst[b] = cv2.GaussianBlur(cv2.resize(src[b], (oW, oH)), (11, 11), 0)   (INTER_LINEAR; sigma 2.0,
 *   BORDER_REFLECT_101; utils.py:61-64 blur()).  `minmax` (optional, [B][2] words) receives the per-map minimum and
 *   maximum of dst as order-preserving keys -- opaque, for vinet_normalize_u8 only.
 * vinet_minmax: the same keys for maps that did not come out of vinet_resize_blur.
 * vinet_normalize_u8: utils.py:66-78 img_save(tensor, normalize=True) for each map on its own:
 *   x = (clamp(x, min, max) - min) / (max - min + 1e-5);  u8 = round_half_even(clamp(x * 255 + 0.5, 0, 255)).
 * ---------------------------------------------------------------------- */
int vinet_resize_blur(const float* src, int32_t B, int32_t H, int32_t W, float* dst, int32_t oH, int32_t oW,
                      uint32_t* minmax, void* stream);
int vinet_minmax(const float* src, int32_t B, int64_t n, uint32_t* minmax, void* stream);
int vinet_normalize_u8(const float* src, const uint32_t* minmax, int32_t B, int64_t n, uint8_t* dst, void* stream);

/* ------------------------------------------------------------------------
 * Input pipeline: decoded frames / ground-truth maps arrive as BYTES at their own
 * resolution and become the network's float32 inputs on device.  Scratch is the caller's (`*_ws_bytes`, 16-byte aligned).
 *
 * vinet_frames_preprocess: dataloader.py:243-250 / generate_result.py:77-88 img_transform --
 *   transforms.Resize((oH, oW)) [PIL Image.resize(BILINEAR): 22-bit fixed-point triangle filter, horizontal pass rounded
 *   to bytes, then vertical pass; Pillow pinned by requirements.txt:106] -> ToTensor (/ 255) -> Normalize.
 *   src uint8 [N][H][W][3] (RGB interleaved, = np.asarray(img.convert('RGB'))), dst float32 [N][3][oH][oW];
 *   mean_std: six HOST floats (mean r,g,b, std r,g,b), read during the call.
 * vinet_gt_preprocess: dataloader.py:283-296 -- uint8 'L' maps [N][H][W] -> float64 -> cv2.resize to (oW, oH) when the
 *   sizes differ (train mode; INTER_LINEAR, float32 weights, double arithmetic) -> / 255 when the map's maximum
 *   exceeds 1 -> float32 [N][oH][oW].
 * ---------------------------------------------------------------------- */
int64_t vinet_frames_preprocess_ws_bytes(int32_t N, int32_t H, int32_t W, int32_t oH, int32_t oW);
int vinet_frames_preprocess(const uint8_t* src, int32_t N, int32_t H, int32_t W, float* dst, int32_t oH, int32_t oW,
                            const float* mean_std, void* ws, void* stream);
/* vinet_audio_excerpt: dataloader.py:89-122 get_audio_feature -- out (win = 70560 floats) = zeros, with
 *   float(np.hanning(M)) * wav[start : end + 1] centred at win / 2 (offset win/2 - M/2; M = the clamped slice length;
 *   hanning in double with numpy 1.18.5's formula, requirements.txt:91).  wav is the device-resident waveform of the video
 *   (already scaled by 2^-23, dataloader.py:63). */
int vinet_audio_excerpt(const float* wav, int64_t n_samples, int64_t start, int64_t end, float* out, int32_t win, void* stream);
int64_t vinet_gt_preprocess_ws_bytes(int32_t N, int32_t oH, int32_t oW);
int vinet_gt_preprocess(const uint8_t* src, int32_t N, int32_t H, int32_t W, float* dst, int32_t oH, int32_t oW, void* ws,
                        void* stream);

/* misc */
/* Tuning / A-B switches (process-wide): "dma" (1 = use the LDS-DMA conv kernel where
 * legal, default 1), "wgrad_tr" (1 = hardware transpose reads in the register-staged wgrad, default 1),
 * "wgrad_dma" (1 = LDS-DMA multi-tap wgrad kernel where legal, default 1),
 * "pp" (256x256x64 / 256x192x64 ping-pong conv kernel: 0 off, 1 heuristic (default), 2 force, 3 / 4 force the
 * 256- / 192-wide shape), "wgrad_pp" (ping-pong wgrad kernel, same values; 3 / 4 force the 256- / 192-row tile),
 * "wgrad_tg" (taps per group of the 64x64 wgrad kernel, 0 = heuristic), "pool_twalk" (T-walking 3x3x3/s1 pool
 * kernels: 0 off, 1 large tensors (default), 2 always), "tperm" (t-fastest tile order, default 0),
 * "n64_tile" (64-wide convs: 0 heuristic, 1 force 128-row, 2 force 64-row tiles),
 * "conv_ts" / "conv_hs" / "wgrad_ts" / "wgrad_hs" / "wgrad_rs" / "wgrad_tf" (streaming kernels: 0 off, 1 heuristic
 * (default), 2 every eligible shape), "reduce_il" (channel reductions walk one window, default 1).
 * None of them changes results beyond floating-point accumulation order. */
int vinet_set_option(const char* name, int32_t value);
/* fp32 view (with its pending affine applied) -> hi = bf16(v) and lo = bf16(v - hi) planes of the same dims: the operands of
 * the bf16 kernels when they serve the VINET_F32S arithmetic as three accumulating launches (hi*hi + lo*hi + hi*lo; the weight
 * gradients: every bf16 weight-gradient kernel ADDS into `dw`).  No reference counterpart: a numerics tool of this path. */
int vinet_split_bf16(const VinetTensor* src, VinetAffine pre, const VinetTensor* hi, const VinetTensor* lo, void* stream);
int vinet_fill_f32(float* p, int64_t n, float value, void* stream);
/* Test / tuning aid, no reference counterpart: one wave that idles for ~`cycles` shader clocks on `stream` (delays whatever is
 * enqueued behind it; touches no memory).  Used to perturb the two-stream backward schedule in eager mode. */
int vinet_debug_spin(int64_t cycles, void* stream);
int vinet_abi_version(void);
const char* vinet_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* VINET_HIP_H */
