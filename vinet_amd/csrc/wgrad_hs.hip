// Weight gradient of the RGB stem (3 -> 64, 1x7x7, stride (1,2,2), model_utils.py:144 via SepConv3d(3,64,7,2,3)) over
// the folded view of the zero-padded 4-channel clip (vinet_import_ncdhw_pad): one view position = 2 pixels, its 32
// "channels" = 8 pixels x 4, taps = the 7 kernel rows, stride (1,2,1):
//
//   dw[kh][n][kw*4+c] += sum_{b,t,ho,wo} dz[b,t,ho,wo,n] * x[b, t, 2ho+kh, pixel 2wo+kw, c]
//
// conv_wgrad_dma_kernel<64,32,7> stages seven 32-voxel x tiles of 64-byte rows per step although neighbouring
// positions share 6 of their 8 pixels and neighbouring output rows 5 of their 7 input rows: 14 KB of x staging for
// 4 KB of dz (1.4 TB/s, bound by the DMA address arithmetic).  Here a persistent workgroup owns a 64-wide column strip
// of one frame and walks the output rows with the 7 live input rows (134 pixels = 1072 B each) in an LDS ring: per
// step 2 new rows and one dz tile (8 KB) are fetched, every byte once, and the overlapped view turns into
// overlapping fragment ADDRESSES (position v starts 16 bytes after position v-1) instead of re-staged bytes.
//
//   * 256 threads = 4 waves; wave w owns output channels [16w, 16w+16) x all 7 x 32 columns: 14 accumulator tiles;
//   * both operands are position-major: K-major fragments by ds_read_b64_tr_b16 (dz tile with the 16-byte chunk XOR
//     of wgrad_dma.hip, x rows unswizzled);
//   * loads of step ho+1 are issued before the MFMAs of step ho (named registers, unconditional: see wgrad_ts.hip);
//   * one fp32-atomic flush per workgroup (dw is zero on entry).
// The padded image makes every access in range: no bounds checks at all.
#include "common.h"

struct WgradHsArgs {
  const char* x;
  const char* dy;
  float* dw;
  long sBx, sBy;
  int T, Hp, Wv, ldx;          // x view: [B][T][Hp][Wv][32], ld = 8
  int oH, oW, ldy;
  int items, strips;           // items = B * T * strips, strips = oW / 64
  FastDiv dStrips, dT;
  // fused BatchNorm(+ReLU) backward on the dz operand (VinetWgradDesc::bnb_*): dy is the gradient behind the BN
  const char* z;
  long sBz;
  int ldz, relu;
  const float* f_scale; const float* f_shift; const float* mean; const float* invstd; const float* c1; const float* c2;
};

// 128-byte rows: a 32-lane transpose read touches rows {a .. a+3, a+8 .. a+11}; rows of equal parity share a 128-byte half of the
// 256-byte bank row, so bits 1 and 3 of the row pick one of its four 32-byte windows (bit 1 alone left rows r and r + 8 on
// the same banks: a 2-way conflict on every ds_read_b64_tr_b16)
VN_DEV int whs_swz(int r) { return (((r >> 1) & 1) | (((r >> 3) & 1) << 1)) << 1; }

template <bool BNB>
__global__ __launch_bounds__(256, 2) void conv_wgrad_hs_kernel(const WgradHsArgs a) {
  constexpr int ROW = 1088, NP = 67;                 // ring slot: 134 pixels x 8 B = 67 pieces of 16 B (+ pad)
  constexpr int DZT = 64 * 64 * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* dzb = smem;                                  // 2 dz tiles [64 positions][64 channels]
  char* ring = smem + 2 * DZT;                       // 7 rows
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int l_chunk = tid & 7, l_row = tid >> 3;     // dz pieces: rows l_row, l_row + 32
  int l_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = l_row + 32 * j;
    l_off[j] = r * 128 + ((l_chunk ^ whs_swz(r)) * 16);
  }
  // fused BN backward: dz = A*(g*mask) + Bc*z + D with the gate from A*z + sh (bn.hip::bn_bwd_apply8_kernel);
  // a thread always handles the same 8 channels
  float cA[8], cB[8], cD[8], cS[8];
  if constexpr (BNB) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = l_chunk * 8 + e;
      const float sc = a.f_scale[c], k = sc * a.invstd[c] * a.c2[c];
      cA[e] = sc; cB[e] = -k; cD[e] = fmaf(k, a.mean[c], -sc * a.c1[c]);
      cS[e] = a.f_shift ? a.f_shift[c] : 0.f;
    }
  }
  auto bnb = [&](uint4 g, uint4 z) -> uint4 {
    if constexpr (!BNB) return g;
    const uint32_t gu[4] = {g.x, g.y, g.z, g.w}, zu[4] = {z.x, z.y, z.z, z.w};
    uint32_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float r[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const float gv = __uint_as_float(hh ? (gu[e] & 0xffff0000u) : (gu[e] << 16));
        const float zv = __uint_as_float(hh ? (zu[e] & 0xffff0000u) : (zu[e] << 16));
        const int c = 2 * e + hh;
        float gg = gv;
        if (a.relu && !(fmaf(zv, cA[c], cS[c]) > 0.f)) gg = 0.f;
        r[hh] = fmaf(cA[c], gg, fmaf(cB[c], zv, cD[c]));
      }
      o[e] = pack2bf(r[0], r[1]);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
  };
  // x pieces of a step: 2 rows x 67 pieces = 134 over threads 0..133
  const bool xl = tid < 2 * NP;
  const int x_r = tid >= NP ? 1 : 0, x_p = tid - x_r * NP;

  f32x4_v acc[7][2];
#pragma unroll
  for (int g = 0; g < 7; ++g)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[g][j] = (f32x4_v){0.f, 0.f, 0.f, 0.f};

  auto frag_dz = [&](const char* tile, int ks, int col0) -> bf16x8_v {
    union { bf16x8_v v; s16x4_v h[2]; } u;
    const int p = lane & 15;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int krow = ks * 32 + (lane >> 4) * 8 + h * 4 + (p >> 2);
      const int col = col0 + (p & 3) * 4;
      const int ch = (col >> 3) ^ whs_swz(krow);
      u.h[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_v*)(tile + krow * 128 + ch * 16 + (col & 7) * 2));
    }
    return u.v;
  };
  // position v of the strip starts at byte 16*v of a ring row; columns = 8 pixels x 4 channels
  auto frag_x = [&](const char* row, int ks, int col0) -> bf16x8_v {
    union { bf16x8_v v; s16x4_v h[2]; } u;
    const int p = lane & 15;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int krow = ks * 32 + (lane >> 4) * 8 + h * 4 + (p >> 2);
      const int col = col0 + (p & 3) * 4;
      u.h[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_v*)(row + krow * 16 + col * 2));
    }
    return u.v;
  };

  for (int item = blockIdx.x; item < a.items; item += gridDim.x) {
    const int bt = (int)fdiv((uint32_t)item, a.dStrips);
    const int wo0 = (item - bt * a.strips) * 64;
    const int b = (int)fdiv((uint32_t)bt, a.dT);
    const int t = bt - b * a.T;
    const char* xrow0 = a.x + ((long)b * a.sBx + ((long)t * a.Hp * a.Wv + wo0) * (long)a.ldx) * 2;   // row h adds h * x_rowb
    const long x_rowb = (long)a.Wv * a.ldx * 2;
    const char* db = a.dy + ((long)b * a.sBy + ((long)t * a.oH * a.oW + wo0 + l_row) * (long)a.ldy + l_chunk * 8) * 2;
    const long d_rowb = (long)a.oW * a.ldy * 2, d_r32 = 32L * a.ldy * 2;
    const char* zb = BNB ? a.z + ((long)b * a.sBz + ((long)t * a.oH * a.oW + wo0 + l_row) * (long)a.ldz + l_chunk * 8) * 2 : db;
    const long z_rowb = (long)a.oW * a.ldz * 2, z_r32 = 32L * a.ldz * 2;

    // ---- prologue: input rows 0..6, dz row 0 --------------------------------------------------------------------
    for (int q = tid; q < 7 * NP; q += 256) {
      const int h = q / NP, pc = q - h * NP;
      *(uint4*)(ring + h * ROW + pc * 16) = *(const uint4*)(xrow0 + h * x_rowb + pc * 16);
    }
    *(uint4*)(dzb + l_off[0]) = bnb(*(const uint4*)db, *(const uint4*)zb);
    *(uint4*)(dzb + l_off[1]) = bnb(*(const uint4*)(db + d_r32), *(const uint4*)(zb + z_r32));
    __syncthreads();

    for (int ho = 0; ho < a.oH; ++ho) {
      const bool more = ho + 1 < a.oH;
      const int hn = more ? 2 * ho + 7 + x_r : 0;                       // (clamped: always a readable row)
      const uint4 nx = *(const uint4*)(xrow0 + hn * x_rowb + (xl ? x_p : 0) * 16);
      const char* ds = db + (more ? ho + 1 : ho) * d_rowb;
      const uint4 nd0 = *(const uint4*)ds, nd1 = *(const uint4*)(ds + d_r32);
      const char* zs = zb + (more ? ho + 1 : ho) * z_rowb;
      uint4 nz0 = nd0, nz1 = nd1;
      if constexpr (BNB) { nz0 = *(const uint4*)zs; nz1 = *(const uint4*)(zs + z_r32); }

      const char* dt = dzb + (ho & 1) * DZT;
      const int s0 = (2 * ho) % 7;                                      // ring slot of input row 2ho
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8_v af = frag_dz(dt, ks, wave * 16);
#pragma unroll
        for (int g = 0; g < 7; ++g) {
          const int si = s0 + g - (s0 + g >= 7 ? 7 : 0);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const bf16x8_v bf = frag_x(ring + si * ROW, ks, j * 16);
            mfma_bf16_acc(acc[g][j], af, bf);
          }
        }
      }
      __syncthreads();          // rows 2ho, 2ho+1 are free
      if (more) {
        if (xl) *(uint4*)(ring + ((2 * ho + 7 + x_r) % 7) * ROW + x_p * 16) = nx;
        char* dn = dzb + ((ho + 1) & 1) * DZT;
        *(uint4*)(dn + l_off[0]) = bnb(nd0, nz0);
        *(uint4*)(dn + l_off[1]) = bnb(nd1, nz1);
      }
      __syncthreads();
    }
  }
  mfma_drain();
  // dw[kh][n][col]: n = wave*16 + (lane>>4)*4 + r, col = j*16 + (lane & 15)
#pragma unroll
  for (int g = 0; g < 7; ++g)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = wave * 16 + (lane >> 4) * 4 + r, col = j * 16 + (lane & 15);
        atomicAdd(a.dw + ((long)g * 64 + n) * 32 + col, acc[g][j][r]);
      }
}

int g_vinet_opt_wgrad_hs = 1;   // 0 = off, 2 = force on every eligible shape (tests)

// VinetWgradDesc::tline == 2: the caller promises taps (0, kh, 0, slice kh), kh = 0..6 (the folded stem)
bool vinet_wgrad_use_hs(const VinetWgradDesc* d) {
  if (!g_vinet_opt_wgrad_hs || d->tline != 2 || d->dtype != VINET_BF16 || d->mode != VINET_CONV_GENERIC) return false;
  if (d->pre.scale || d->pre.relu) return false;
  if (d->bnb_z && !(d->bnb_fwd.scale && d->bnb_mean && d->bnb_invstd && d->bnb_c1 && d->bnb_c2 && d->bnb_ld % 8 == 0 &&
                    d->bnb_sB % 8 == 0 && ((uintptr_t)d->bnb_z % 16) == 0)) return false;
  const bool shape = d->x.C == 32 && d->x.ld == 8 && d->Kp == 32 && d->ntaps == 7 && d->sT == 1 && d->sH == 2 && d->sW == 1 &&
                     d->dy.C == 64 && d->dy.W % 64 == 0 && d->x.T == d->dy.T && d->x.W >= d->dy.W + 3 && d->x.H >= 2 * d->dy.H + 5 &&
                     d->dy.ld % 8 == 0 && d->dy.sB % 8 == 0 && d->x.sB % 8 == 0 && ((uintptr_t)d->x.ptr % 16) == 0 &&
                     ((uintptr_t)d->dy.ptr % 16) == 0;
  if (!shape) return false;
  if (g_vinet_opt_wgrad_hs >= 2) return true;
  return (long)d->dy.B * d->dy.T * (d->dy.W / 64) >= 512 && d->dy.H >= 8;
}

int vinet_launch_wgrad_hs(const VinetWgradDesc* d, hipStream_t s) {
  WgradHsArgs a;
  a.x = (const char*)d->x.ptr; a.dy = (const char*)d->dy.ptr; a.dw = d->dw;
  a.sBx = d->x.sB; a.sBy = d->dy.sB;
  a.T = d->x.T; a.Hp = d->x.H; a.Wv = d->x.W; a.ldx = d->x.ld;
  a.oH = d->dy.H; a.oW = d->dy.W; a.ldy = d->dy.ld;
  a.strips = a.oW / 64;
  a.items = d->dy.B * a.T * a.strips;
  a.dStrips = make_fastdiv((uint32_t)a.strips);
  a.dT = make_fastdiv((uint32_t)a.T);
  const int smem = 2 * 64 * 64 * 2 + 7 * 1088;
  int grid = 768;     // 3 workgroups per CU (130 registers, 24 KB of LDS)
  if (grid > a.items) grid = a.items;
  a.z = (const char*)d->bnb_z; a.sBz = d->bnb_sB; a.ldz = d->bnb_ld; a.relu = d->bnb_fwd.relu;
  a.f_scale = d->bnb_fwd.scale; a.f_shift = d->bnb_fwd.shift; a.mean = d->bnb_mean; a.invstd = d->bnb_invstd;
  a.c1 = d->bnb_c1; a.c2 = d->bnb_c2;
  if (d->bnb_z) hipLaunchKernelGGL(conv_wgrad_hs_kernel<true>, dim3(grid), dim3(256), smem, s, a);
  else hipLaunchKernelGGL(conv_wgrad_hs_kernel<false>, dim3(grid), dim3(256), smem, s, a);
  return vn_launch_status("conv_wgrad_hs");
}
