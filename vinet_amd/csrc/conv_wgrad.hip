// Weight-gradient convolution for gfx950: for every tap,
//   dw[slice][n][c] += sum_m dy[m][n] * pre(x[m shifted by tap])[c]
// i.e. a GEMM whose reduction axis is the voxel index m.  In channels-last
// storage both operands are K-major *rows* ([voxel][channel]); MFMA wants 8
// consecutive k per lane, so the [32 voxel][64 channel] LDS tiles are read with
// the gfx950 hardware transpose read (ds_read_b64_tr_b16) for bf16 and with
// plain ds_read_b32 for fp32 (mfma_f32_16x16x4f32 takes one float per lane).
//
// grid = (tilesN * tilesC, ntaps, splitK); workgroup = 4 waves (2x2), each wave
// a 32x32 block of the 64(n) x 64(c) tile; split-K partials are combined with
// fp32 atomics into the caller-zeroed dw buffer.
#include "common.h"

extern int g_vinet_opt_wgrad_tr;
extern int g_vinet_opt_wgrad_dma;
extern int g_vinet_opt_wgrad_tg;
int vinet_launch_wgrad_dma(const VinetWgradDesc* d, hipStream_t s);
int vinet_wgrad_dma_name(const VinetWgradDesc* d, char* buf, int n);
bool vinet_wgrad_use_pp(const VinetWgradDesc* d);
bool vinet_wgrad_use_ts(const VinetWgradDesc* d);
bool vinet_wgrad_use_hs(const VinetWgradDesc* d);
bool vinet_wgrad_use_rs(const VinetWgradDesc* d);
int vinet_launch_wgrad_rs(const VinetWgradDesc* d, hipStream_t s);
bool vinet_wgrad_use_skinny(const VinetWgradDesc* d);
int vinet_launch_wgrad_skinny(const VinetWgradDesc* d, hipStream_t s);
bool vinet_wgrad_use_tf(const VinetWgradDesc* d);
int vinet_launch_wgrad_tf(const VinetWgradDesc* d, hipStream_t s);
int vinet_launch_wgrad_hs(const VinetWgradDesc* d, hipStream_t s);
int vinet_launch_wgrad_ts(const VinetWgradDesc* d, hipStream_t s);
int vinet_wgrad_pp_rows(int N);
int vinet_launch_wgrad_pp(const VinetWgradDesc* d, hipStream_t s);

struct WgradArgs {
  const char* x;
  const char* dy;
  float* dw;
  const int4* taps;
  const float* in_scale;
  const float* in_shift;
  int in_relu;
  int Ti, Hi, Wi, Cin, ldx;
  long sBx;
  int To, Ho, Wo, N, ldy;
  long sBy;
  int sT, sH, sW;
  int ntaps, Kp, M;
  int tilesN, tilesC, splitK, chunks_per_split, nchunks;
  int use_tr;
  FastDiv dW, dH, dT;
};

// hi / lo bf16 pairs of two fp32 values (conv_igemm.h: the split-bf16 form, VINET_F32S)
VN_DEV void wg_split_pair(float x0, float x1, uint32_t& hi, uint32_t& lo) {
  hi = cvt_pk_bf16_f32(x0, x1);
  lo = cvt_pk_bf16_f32(x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u));
}

// SPLIT (T = float, VINET_F32S): fp32 tensors, each operand split into hi + lo bf16 on the way into LDS (a hi and a lo image per
// tile), three bf16 MFMAs per product -- the weight-gradient counterpart of conv_igemm_kernel's split form.
template <typename T, int MODE, bool SPLIT = false>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs a) {
  constexpr int TN = 64, TC = 64, KV = 32;
  constexpr int EG = ElemTraits<T>::EG;
  constexpr int GW = TN / EG;                       // 16-byte groups per tile row
  constexpr int RSW = SPLIT ? TN * 2 + 16 : TN * (int)sizeof(T) + 16;      // LDS row stride (bytes)
  constexpr int LOADS = KV * GW / 256;               // per operand per thread
  constexpr int IMG_BYTES = KV * RSW;                // one [32 voxel][64 channel] image
  constexpr int TILE_BYTES = (SPLIT ? 2 : 1) * IMG_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // D[2], X[2]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile_c = blockIdx.x % a.tilesC, tile_n = blockIdx.x / a.tilesC;
  const int4 tp = load_tap(a.taps, blockIdx.y);
  const int chunk0 = blockIdx.z * a.chunks_per_split;
  int chunk1 = chunk0 + a.chunks_per_split;
  if (chunk1 > a.nchunks) chunk1 = a.nchunks;
  const int n0 = tile_n * TN, c0 = tile_c * TC;
  const bool has_pre = a.in_scale != nullptr;

  uint4 rd[LOADS], rx[LOADS];

  auto load_tiles = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      const int idx = i * 256 + tid;
      const int row = idx / GW, gg = idx % GW;
      const int m = chunk * KV + row;
      uint4 vd = make_uint4(0, 0, 0, 0), vx = make_uint4(0, 0, 0, 0);
      if (m < a.M) {
        int b, to, ho, wo;
        decode_m(m, a.dW, a.dH, a.dT, b, to, ho, wo);
        const int n = n0 + gg * EG;
        if (n < a.N) {
          const long off = (long)b * a.sBy + ((long)(to * a.Ho + ho) * a.Wo + wo) * (long)a.ldy + n;
          vd = *(const uint4*)(a.dy + off * (long)sizeof(T));
        }
        const int ti = to * a.sT + tp.x, hi = ho * a.sH + tp.y;
        if constexpr (MODE == VINET_CONV_GENERIC) {
          const int wi = wo * a.sW + tp.z;
          const int c = c0 + gg * EG;
          if (c < a.Cin && (unsigned)ti < (unsigned)a.Ti && (unsigned)hi < (unsigned)a.Hi &&
              (unsigned)wi < (unsigned)a.Wi) {
            const long off = (long)b * a.sBx + ((long)(ti * a.Hi + hi) * a.Wi + wi) * (long)a.ldx + c;
            vx = *(const uint4*)(a.x + off * (long)sizeof(T));
            if (has_pre) {
              float f[EG];
              unpack16<T>(vx, f);
#pragma unroll
              for (int e = 0; e < EG; ++e) {
                f[e] = fmaf(f[e], a.in_scale[c + e], a.in_shift[c + e]);
                if (a.in_relu) f[e] = fmaxf(f[e], 0.f);
              }
              vx = pack16<T>(f);
            }
          }
        } else {
          constexpr int PP = EG / 4;
          if (gg * EG < 32 && (unsigned)ti < (unsigned)a.Ti && (unsigned)hi < (unsigned)a.Hi) {
            const long rowoff = (long)b * a.sBx + ((long)(ti * a.Hi + hi) * a.Wi) * (long)a.ldx;
            uint32_t words[4] = {0, 0, 0, 0};
#pragma unroll
            for (int p = 0; p < PP; ++p) {
              const int wi = wo * a.sW + tp.z + gg * PP + p;
              if ((unsigned)wi < (unsigned)a.Wi) {
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
            vx = make_uint4(words[0], words[1], words[2], words[3]);
          }
        }
      }
      rd[i] = vd; rx[i] = vx;
    }
  };
  auto store_tiles = [&](int buf) {
    char* Ds = smem + buf * TILE_BYTES;
    char* Xs = smem + (2 + buf) * TILE_BYTES;
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      const int idx = i * 256 + tid;
      const int row = idx / GW, gg = idx % GW;
      if constexpr (SPLIT) {
        auto put = [&](char* img, const uint4& v) {
          uint32_t h0, l0, h1, l1;
          wg_split_pair(__uint_as_float(v.x), __uint_as_float(v.y), h0, l0);
          wg_split_pair(__uint_as_float(v.z), __uint_as_float(v.w), h1, l1);
          *(uint2*)(img + row * RSW + gg * 8) = make_uint2(h0, h1);
          *(uint2*)(img + IMG_BYTES + row * RSW + gg * 8) = make_uint2(l0, l1);
        };
        put(Ds, rd[i]);
        put(Xs, rx[i]);
      } else {
        *(uint4*)(Ds + row * RSW + gg * 16) = rd[i];
        *(uint4*)(Xs + row * RSW + gg * 16) = rx[i];
      }
    }
  };

  f32x4_v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_v){0.f, 0.f, 0.f, 0.f};

  // K-major fragment of a [32 voxel][64 ch] LDS tile: 8 consecutive voxels
  // (k = (lane>>4)*8 ..+7) for channel col0 + (lane&15).
  auto frag_bf16 = [&](const char* tile, int col0) -> bf16x8_v {
    union { bf16x8_v v; s16x4_v h[2]; uint16_t s[8]; } u;
    if (a.use_tr) {
      const int p = lane & 15;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int krow = (lane >> 4) * 8 + h * 4 + (p >> 2);
        const char* src = tile + krow * RSW + (col0 + (p & 3) * 4) * 2;
        u.h[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_v*)src);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        u.s[e] = *(const uint16_t*)(tile + ((lane >> 4) * 8 + e) * RSW + (col0 + (lane & 15)) * 2);
    }
    return u.v;
  };

  auto compute = [&](int buf) {
    const char* Ds = smem + buf * TILE_BYTES;
    const char* Xs = smem + (2 + buf) * TILE_BYTES;
    if constexpr (SPLIT) {
      bf16x8_v ah[2], al[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) { ah[i] = frag_bf16(Ds, wm * 32 + i * 16); al[i] = frag_bf16(Ds + IMG_BYTES, wm * 32 + i * 16); }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bf16x8_v bh = frag_bf16(Xs, wn * 32 + j * 16), bl = frag_bf16(Xs + IMG_BYTES, wn * 32 + j * 16);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh, acc[i][j], 0, 0, 0);
        }
      }
    } else if constexpr (sizeof(T) == 2) {
      bf16x8_v af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = frag_bf16(Ds, wm * 32 + i * 16);
#pragma unroll
      for (int j = 0; j < 2; ++j) bfr[j] = frag_bf16(Xs, wn * 32 + j * 16);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const int krow = kk * 4 + (lane >> 4);
        float af[2], bfr[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *(const float*)(Ds + krow * RSW + (wm * 32 + i * 16 + (lane & 15)) * 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) bfr[j] = *(const float*)(Xs + krow * RSW + (wn * 32 + j * 16 + (lane & 15)) * 4);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
    }
  };

  if (chunk0 < chunk1) {
    load_tiles(chunk0);
    store_tiles(0);
    __syncthreads();
    for (int ch = chunk0; ch < chunk1; ++ch) {
      const int buf = (ch - chunk0) & 1;
      if (ch + 1 < chunk1) load_tiles(ch + 1);
      compute(buf);
      if (ch + 1 < chunk1) store_tiles(buf ^ 1);
      __syncthreads();
    }
  }

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r;
        const int c = c0 + wn * 32 + j * 16 + (lane & 15);
        if (n < a.N && c < a.Kp) {
          float* dst = a.dw + ((long)tp.w * a.N + n) * (long)a.Kp + c;
          atomicAdd(dst, acc[i][j][r]);      // (always +=: dw is zero on entry, or holds the other terms of a split-bf16 sum / an earlier use of a shared weight)
        }
      }
}

// LDS transpose-read self test: fills a [16][16] bf16 tile with value k*16+i and
// returns what each lane receives, so the host can verify the fragment mapping
// assumed by frag_bf16 on real hardware.
__global__ void tr16_selftest_kernel(uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t tile[32 * 16];
  const int lane = threadIdx.x;
  for (int e = lane; e < 32 * 16; e += 64) tile[e] = (uint16_t)e;  // value = k*16 + i
  __syncthreads();
  const int p = lane & 15;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int krow = (lane >> 4) * 8 + h * 4 + (p >> 2);
    const uint16_t* src = &tile[krow * 16 + (p & 3) * 4];
    s16x4_v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_v*)src);
#pragma unroll
    for (int e = 0; e < 4; ++e) out[lane * 8 + h * 4 + e] = (uint16_t)v[e];
  }
}

extern "C" int vinet_selftest_tr16(uint16_t* out_dev, void* stream) {
  hipLaunchKernelGGL(tr16_selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out_dev);
  return vn_launch_status("tr16_selftest");
}

static bool wgrad_use_dma(const VinetWgradDesc* d) {
  const bool pre_ok = !d->pre.scale || (d->pre.relu && d->pre.shift);
  return g_vinet_opt_wgrad_dma && d->dtype == VINET_BF16 && d->mode == VINET_CONV_GENERIC && pre_ok &&
         !(d->pre.relu && !d->pre.scale);
}

extern "C" int vinet_conv3d_wgrad_kernel_name(const VinetWgradDesc* d, char* buf, int32_t n) {
  if (!d || !buf || n <= 0) return -1;
  if (vinet_wgrad_use_skinny(d)) { snprintf(buf, n, "wgrad_skinny_kernel"); return 0; }
  if (vinet_wgrad_use_rs(d)) { snprintf(buf, n, "conv_wgrad_rs_kernel<W%d>", d->dy.W); return 0; }
  if (vinet_wgrad_use_hs(d)) { snprintf(buf, n, d->bnb_z ? "conv_wgrad_hs_kernel<bn_bwd>" : "conv_wgrad_hs_kernel"); return 0; }
  if (vinet_wgrad_use_ts(d)) { snprintf(buf, n, "conv_wgrad_ts_kernel<%s>", d->pre.scale ? "pre" : "plain"); return 0; }
  if (vinet_wgrad_use_tf(d)) { snprintf(buf, n, "conv_wgrad_tf_kernel<%s>", d->pre.scale ? "pre" : "plain"); return 0; }
  if (wgrad_use_dma(d) && vinet_wgrad_use_pp(d)) { snprintf(buf, n, "conv_wgrad_pp_kernel<%s,%d>", d->pre.scale ? "pre" : "plain", vinet_wgrad_pp_rows(d->dy.C)); return 0; }
  if (wgrad_use_dma(d)) return vinet_wgrad_dma_name(d, buf, n);
  snprintf(buf, n, "conv_wgrad_kernel<%s,%d>", d->dtype == VINET_BF16 ? "bf16" : (d->dtype == VINET_F32S ? "float/split" : "float"), d->mode);
  return 0;
}

extern "C" int vinet_conv3d_wgrad_fuses_bn_bwd(const VinetWgradDesc* d) {
  return d && d->bnb_z && vinet_wgrad_use_hs(d) ? 1 : 0;
}

extern "C" int vinet_conv3d_wgrad(const VinetWgradDesc* d, void* stream) {
  VN_CHECK_ARG(d != nullptr, "wgrad: null descriptor");
  VN_CHECK_ARG(!d->bnb_z || vinet_conv3d_wgrad_fuses_bn_bwd(d), "wgrad: fused BN backward requested for a problem whose kernel cannot apply it");
  VN_CHECK_ARG(d->dtype == VINET_F32 || d->dtype == VINET_BF16 || d->dtype == VINET_F32S, "wgrad: bad dtype %d", d->dtype);
  const int eg = vn_f32_storage(d->dtype) ? 4 : 8;
  VN_CHECK_ARG(vn_tensor_ok(d->x, d->mode == VINET_CONV_STEM ? 4 : eg, true), "wgrad: bad x view");
  VN_CHECK_ARG(vn_tensor_ok(d->dy, eg), "wgrad: bad dy view");
  VN_CHECK_ARG(d->x.B == d->dy.B, "wgrad: batch mismatch");
  VN_CHECK_ARG(d->ntaps > 0 && d->taps && d->dw, "wgrad: taps/dw missing");
  VN_CHECK_ARG(d->Kp > 0 && d->Kp % 32 == 0, "wgrad: Kp must be a multiple of 32");
  if (d->mode == VINET_CONV_STEM) VN_CHECK_ARG(d->x.C == 4 && d->Kp == 32, "wgrad stem: x.C must be 4, Kp 32");
  else VN_CHECK_ARG(d->Kp >= d->x.C, "wgrad: Kp < Cin");

  if (vinet_wgrad_use_skinny(d)) return vinet_launch_wgrad_skinny(d, (hipStream_t)stream);
  if (vinet_wgrad_use_rs(d)) return vinet_launch_wgrad_rs(d, (hipStream_t)stream);
  if (vinet_wgrad_use_hs(d)) return vinet_launch_wgrad_hs(d, (hipStream_t)stream);
  if (vinet_wgrad_use_ts(d)) return vinet_launch_wgrad_ts(d, (hipStream_t)stream);
  if (vinet_wgrad_use_tf(d)) return vinet_launch_wgrad_tf(d, (hipStream_t)stream);
  if (wgrad_use_dma(d) && vinet_wgrad_use_pp(d)) return vinet_launch_wgrad_pp(d, (hipStream_t)stream);
  if (wgrad_use_dma(d)) return vinet_launch_wgrad_dma(d, (hipStream_t)stream);
  WgradArgs a;
  a.x = (const char*)d->x.ptr; a.dy = (const char*)d->dy.ptr; a.dw = d->dw; a.taps = (const int4*)d->taps;
  a.in_scale = d->pre.scale; a.in_shift = d->pre.shift; a.in_relu = d->pre.relu;
  a.Ti = d->x.T; a.Hi = d->x.H; a.Wi = d->x.W; a.Cin = d->x.C; a.ldx = d->x.ld; a.sBx = d->x.sB;
  a.To = d->dy.T; a.Ho = d->dy.H; a.Wo = d->dy.W; a.N = d->dy.C; a.ldy = d->dy.ld; a.sBy = d->dy.sB;
  a.sT = d->sT; a.sH = d->sH; a.sW = d->sW;
  a.ntaps = d->ntaps; a.Kp = d->Kp;
  const long M = (long)d->dy.B * d->dy.T * d->dy.H * d->dy.W;
  VN_CHECK_ARG(M < (1L << 31), "wgrad: M too large");
  a.M = (int)M;
  a.tilesN = vn_div_up(a.N, 64);
  a.tilesC = vn_div_up(d->mode == VINET_CONV_STEM ? 32 : a.Cin, 64);
  a.nchunks = vn_div_up(M, 32);
  const long base_blocks = (long)a.tilesN * a.tilesC * a.ntaps;
  long sk = (2048 + base_blocks - 1) / base_blocks;
  if (sk > a.nchunks / 4) sk = a.nchunks / 4;
  if (sk < 1) sk = 1;
  if (sk > 4096) sk = 4096;
  a.chunks_per_split = vn_div_up(a.nchunks, sk);
  a.splitK = vn_div_up(a.nchunks, a.chunks_per_split);
  a.use_tr = g_vinet_opt_wgrad_tr;
  a.dW = make_fastdiv(a.Wo); a.dH = make_fastdiv(a.Ho); a.dT = make_fastdiv(a.To);

  dim3 grid(a.tilesN * a.tilesC, a.ntaps, a.splitK);
  hipStream_t s = (hipStream_t)stream;
  if (d->dtype == VINET_BF16) {
    if (d->mode == VINET_CONV_STEM) hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, VINET_CONV_STEM>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, VINET_CONV_GENERIC>), grid, dim3(256), 0, s, a);
  } else if (d->dtype == VINET_F32S) {
    if (d->mode == VINET_CONV_STEM) hipLaunchKernelGGL((conv_wgrad_kernel<float, VINET_CONV_STEM, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_wgrad_kernel<float, VINET_CONV_GENERIC, true>), grid, dim3(256), 0, s, a);
  } else {
    if (d->mode == VINET_CONV_STEM) hipLaunchKernelGGL((conv_wgrad_kernel<float, VINET_CONV_STEM>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_wgrad_kernel<float, VINET_CONV_GENERIC>), grid, dim3(256), 0, s, a);
  }
  return vn_launch_status("conv_wgrad");
}
