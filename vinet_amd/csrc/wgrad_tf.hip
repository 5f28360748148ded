// Weight gradient of 3x1x1 / stride 1 / pad 1 TEMPORAL convs with many channels (the conv_t halves of the separable
// convs: 192 -> 192 at 16 x 56 x 96 and 16 x 28 x 48, the Inception conv_t layers; model_utils.py:149):
//
//   dw[kt][n][c] += sum_{b,t,p} dy[b,t,p,n] * pre(x[b, t + kt - 1, p])[c]
//
// conv_wgrad_pp_kernel<pre,192> stages every x tile once per tap and applies the pending BN+ReLU at fragment time on
// every read (360 TF/s, MFMA pipe 19 % busy).  Here -- wgrad_ts.hip / wgrad_rs.hip again -- a 512-thread workgroup fixes
// 64 input channels and up to 192 output channels and walks the frames of its (clip, 96-position strip) items with
// the three live x tiles in an LDS ring: one new x tile (affine + ReLU applied ONCE, on the way in) and one dy tile
// per step, the three taps are three ring slots.
//
//   * wave w: input channels [16(w&3), +16) x 3 taps x output channels [NTW*16*(w>>2), +NTW*16): 3*NTW accumulator
//     tiles (NTW = 6: 192 output channels per workgroup, 72 AGPRs; NTW = 4: 128; NTW = 2: 64);
//   * both operands position-major -> ds_read_b64_tr_b16; per step 9 x-fragments and 3*NTW dy-fragments feed
//     9*NTW MFMAs per wave;
//   * loads of frame t+1 are issued before the MFMAs of frame t (named registers, masked selects: wgrad_rs.hip);
//   * grid = (channel chunks x output chunks) x workers, never more than one workgroup per CU; one atomic flush.
#include "common.h"

struct WgradTfArgs {
  const char* x;
  const char* dy;
  float* dw;
  const float* in_scale;
  const float* in_shift;
  long sBx, sBy;
  int T, HW, ldx, ldy, Cin, N, Kp;
  int cchunks, nchunks, strips, items, workers;    // items = B * strips, strips = HW / 96
  FastDiv dStrips;
};

// 128-byte rows: a 32-lane transpose read touches rows {a .. a+3, a+8 .. a+11}; rows of equal parity share a 128-byte half of the
// 256-byte bank row, so bits 1 and 3 of the row pick one of its four 32-byte windows (bit 1 alone left rows r and r + 8 on
// the same banks: a 2-way conflict on every ds_read_b64_tr_b16)
VN_DEV int wtf_swz(int r) { return (((r >> 1) & 1) | (((r >> 3) & 1) << 1)) << 1; }

template <bool PRE, int NTW>
__global__ __launch_bounds__(512, 1) void conv_wgrad_tf_kernel(const WgradTfArgs a) {
  constexpr int P = 96, KS = 3;
  constexpr int NT = NTW * 32;                       // output channels per workgroup (two wave rows of NTW*16)
  constexpr int XT = P * 128, DT = P * NT * 2;       // bytes of an x tile / a dy tile
  constexpr int DCH = NT / 8;                        // 16-byte chunks per dy row
  constexpr int XPT = 2, DPT = (P * DCH + 511) / 512;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;                                 // 3 x tiles [96 positions][64 channels]
  char* dyb = smem + 3 * XT;                         // 2 dy tiles [96 positions][NT channels]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave & 3, nh = wave >> 2;
  const int groups = a.cchunks * a.nchunks;
  // the channel chunks of one worker walk the same items at the same time: keep them on one XCD so that dy is read
  // from HBM once and from that XCD's L2 by the others
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int grp = lin % groups, worker = lin / groups;
  const int c0 = (grp % a.cchunks) * 64, n0 = (grp / a.cchunks) * NT;

  // x pieces: q = tid + 512*j -> position q >> 3, chunk q & 7 (= tid & 7)
  const int l_chunk = tid & 7;
  const bool cx_ok = c0 + l_chunk * 8 < a.Cin;
  const int cx_off = cx_ok ? c0 + l_chunk * 8 : 0;
  int x_pos[XPT], x_off[XPT];
  bool x_ok[XPT];
#pragma unroll
  for (int j = 0; j < XPT; ++j) {
    const int q = tid + 512 * j;
    x_ok[j] = q < P * 8;
    const int pos = x_ok[j] ? q >> 3 : 0;
    x_pos[j] = pos;
    x_off[j] = pos * 128 + ((l_chunk ^ wtf_swz(pos)) * 16);
  }
  // dy pieces: q -> position q / DCH, chunk q % DCH
  int d_pos[DPT], d_off[DPT], d_ch[DPT];
  bool d_ok[DPT];
#pragma unroll
  for (int j = 0; j < DPT; ++j) {
    const int q = tid + 512 * j;
    const bool in = q < P * DCH;
    const int pos = in ? q / DCH : 0, ch = in ? q - pos * DCH : 0;
    d_ok[j] = in && n0 + ch * 8 < a.N;
    d_pos[j] = pos; d_ch[j] = d_ok[j] ? n0 + ch * 8 : 0;
    d_off[j] = in ? pos * (NT * 2) + ((ch ^ wtf_swz(pos)) * 16) : -1;
  }
  f32x2_v sc2[4], sh2[4];
  if constexpr (PRE) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = cx_off + 2 * e;
      sc2[e] = (f32x2_v){a.in_scale[c], a.in_scale[c + 1]};
      sh2[e] = (f32x2_v){a.in_shift[c], a.in_shift[c + 1]};
    }
  }
  auto xform = [&](uint4 v, bool on) -> uint4 {
    if constexpr (PRE) {
      v.x = pre_relu_pair(v.x, sc2[0], sh2[0]); v.y = pre_relu_pair(v.y, sc2[1], sh2[1]);
      v.z = pre_relu_pair(v.z, sc2[2], sh2[2]); v.w = pre_relu_pair(v.w, sc2[3], sh2[3]);
    }
    const uint32_t m = on ? 0xffffffffu : 0u;
    return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m);
  };
  auto keep = [](uint4 v, bool on) -> uint4 {
    const uint32_t m = on ? 0xffffffffu : 0u;
    return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m);
  };

  f32x4_v acc[3][NTW];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int i = 0; i < NTW; ++i) acc[g][i] = (f32x4_v){0.f, 0.f, 0.f, 0.f};

  // K-major fragment (8 positions x 1 channel per lane) of a position-major tile with `rowb` bytes per position
  auto frag = [&](const char* tile, int rowb, int ks, int col0) -> bf16x8_v {
    union { bf16x8_v v; s16x4_v h[2]; } u;
    const int p = lane & 15;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int pos = ks * 32 + (lane >> 4) * 8 + h * 4 + (p >> 2);
      const int col = col0 + (p & 3) * 4;
      const int ch = (col >> 3) ^ wtf_swz(pos);
      u.h[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_v*)(tile + pos * rowb + ch * 16 + (col & 7) * 2));
    }
    return u.v;
  };
  const uint4 z4 = make_uint4(0, 0, 0, 0);

  for (int item = worker; item < a.items; item += a.workers) {
    const int b = (int)fdiv((uint32_t)item, a.dStrips);
    const int pos0 = (item - b * a.strips) * P;
    const char* xb = a.x + ((long)b * a.sBx + (long)pos0 * a.ldx + cx_off) * 2;           // + (t*HW + pos) * ldx * 2
    const char* db = a.dy + ((long)b * a.sBy + (long)pos0 * a.ldy) * 2;                   // + (t*HW + pos) * ldy * 2 + channel
    const long x_plane = (long)a.HW * a.ldx * 2, d_plane = (long)a.HW * a.ldy * 2;

    // ---- prologue: x frames -1 (zeros), 0, 1 into slots 2, 0, 1; dy frame 0 ---------------------------------------------
#pragma unroll
    for (int j = 0; j < XPT; ++j)
      if (x_ok[j]) {
        const uint4 f0 = *(const uint4*)(xb + (long)x_pos[j] * a.ldx * 2);
        const uint4 f1 = *(const uint4*)(xb + (1 < a.T ? x_plane : 0) + (long)x_pos[j] * a.ldx * 2);
        *(uint4*)(ring + 2 * XT + x_off[j]) = z4;
        *(uint4*)(ring + 0 * XT + x_off[j]) = xform(f0, cx_ok);
        *(uint4*)(ring + 1 * XT + x_off[j]) = xform(f1, cx_ok && 1 < a.T);
      }
#pragma unroll
    for (int j = 0; j < DPT; ++j)
      if (d_off[j] >= 0) {
        const uint4 v = *(const uint4*)(db + ((long)d_pos[j] * a.ldy + d_ch[j]) * 2);
        *(uint4*)(dyb + d_off[j]) = keep(v, d_ok[j]);
      }
    __syncthreads();

    for (int t = 0; t < a.T; ++t) {
      // ---- loads for step t+1: x frame t+2, dy frame t+1 ------------------------------------------------------------------
      const bool more = t + 1 < a.T;
      const bool xin = more && t + 2 < a.T && cx_ok;
      const char* xs = xb + (xin ? t + 2 : 0) * x_plane;
      const char* ds = db + (more ? t + 1 : 0) * d_plane;
      const uint4 nx0 = *(const uint4*)(xs + (long)x_pos[0] * a.ldx * 2);
      const uint4 nx1 = *(const uint4*)(xs + (long)x_pos[1] * a.ldx * 2);
      uint4 nd[DPT];
#pragma unroll
      for (int j = 0; j < DPT; ++j) nd[j] = *(const uint4*)(ds + ((long)d_pos[j] * a.ldy + d_ch[j]) * 2);

      // ---- MFMAs: tap g reads x frame t + g - 1 = ring slot (t + g + 2) % 3 ---------------------------------------------
      const char* dt = dyb + (t & 1) * DT;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8_v af[NTW];
#pragma unroll
        for (int i = 0; i < NTW; ++i) af[i] = frag(dt, NT * 2, ks, (nh * NTW + i) * 16);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          const bf16x8_v bf = frag(ring + ((t + g + 2) % 3) * XT, 128, ks, ct * 16);
#pragma unroll
          for (int i = 0; i < NTW; ++i) mfma_bf16_acc(acc[g][i], af[i], bf);
        }
      }
      __syncthreads();          // frame t-1's slot and the other dy tile are free
      if (more) {
        char* xslot = ring + ((t + 2) % 3) * XT;          // frame t+2 replaces frame t-1
        char* dn = dyb + ((t + 1) & 1) * DT;
        if (x_ok[0]) *(uint4*)(xslot + x_off[0]) = xform(nx0, xin);
        if (x_ok[1]) *(uint4*)(xslot + x_off[1]) = xform(nx1, xin);
#pragma unroll
        for (int j = 0; j < DPT; ++j)
          if (d_off[j] >= 0) *(uint4*)(dn + d_off[j]) = keep(nd[j], d_ok[j]);
      }
      __syncthreads();
    }
  }
  mfma_drain();
  // dw[kt][n][c]: n = n0 + (nh*NTW + i)*16 + (lane>>4)*4 + r, c = c0 + ct*16 + (lane & 15)
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + (nh * NTW + i) * 16 + (lane >> 4) * 4 + r, c = c0 + ct * 16 + (lane & 15);
        if (n < a.N && c < a.Cin) atomicAdd(a.dw + ((long)g * a.N + n) * (long)a.Kp + c, acc[g][i][r]);
      }
}

int g_vinet_opt_wgrad_tf = 1;   // 0 = off, 2 = force on every eligible shape (tests)

static int wtf_ntw(int N) { return N > 128 ? 6 : N > 64 ? 4 : 2; }

// VinetWgradDesc::tline == 1 (temporal line) with 3 taps, stride 1, pad 1
bool vinet_wgrad_use_tf(const VinetWgradDesc* d) {
  if (!g_vinet_opt_wgrad_tf || d->tline != 1 || d->dtype != VINET_BF16 || d->mode != VINET_CONV_GENERIC || d->bnb_z) return false;
  if (d->pre.scale && !(d->pre.relu && d->pre.shift)) return false;
  if (d->pre.relu && !d->pre.scale) return false;
  const long HW = (long)d->dy.H * d->dy.W;
  const int ntw = wtf_ntw(d->dy.C);
  const int groups = ((d->x.C + 63) / 64) * ((d->dy.C + ntw * 32 - 1) / (ntw * 32));
  const bool shape = d->ntaps == 3 && d->tpad == 1 && d->sT == 1 && d->sH == 1 && d->sW == 1 && d->x.T == d->dy.T && d->x.H == d->dy.H &&
                     d->x.W == d->dy.W && HW % 96 == 0 && d->x.C % 8 == 0 && d->dy.C % 8 == 0 && d->Kp >= d->x.C && groups <= 256 &&
                     d->x.ld % 8 == 0 && d->dy.ld % 8 == 0 && d->x.sB % 8 == 0 && d->dy.sB % 8 == 0 && ((uintptr_t)d->x.ptr % 16) == 0 &&
                     ((uintptr_t)d->dy.ptr % 16) == 0;
  if (!shape) return false;
  if (g_vinet_opt_wgrad_tf >= 2) return true;
  return (long)d->dy.B * (HW / 96) >= 512 && d->dy.T >= 4 && d->dy.C >= 96 && d->x.C >= 64;
}

int vinet_launch_wgrad_tf(const VinetWgradDesc* d, hipStream_t s) {
  WgradTfArgs a;
  a.x = (const char*)d->x.ptr; a.dy = (const char*)d->dy.ptr; a.dw = d->dw;
  a.in_scale = d->pre.scale; a.in_shift = d->pre.shift;
  a.sBx = d->x.sB; a.sBy = d->dy.sB;
  a.T = d->dy.T; a.HW = d->dy.H * d->dy.W; a.ldx = d->x.ld; a.ldy = d->dy.ld; a.Cin = d->x.C; a.N = d->dy.C; a.Kp = d->Kp;
  const int ntw = wtf_ntw(a.N), nt = ntw * 32;
  a.cchunks = (a.Cin + 63) / 64; a.nchunks = (a.N + nt - 1) / nt;
  a.strips = a.HW / 96;
  a.items = d->dy.B * a.strips;
  a.dStrips = make_fastdiv((uint32_t)a.strips);
  const int groups = a.cchunks * a.nchunks;
  int workers = vn_wgrad_cus(d) / groups;       // one 512-thread workgroup per CU, never a second round; the caller's cap (VinetWgradDesc::max_cus) leaves CUs to its other stream
  if (workers < 1) workers = 1;
  if (workers > a.items) workers = a.items;
  a.workers = workers;
  const int smem = 3 * 96 * 128 + 2 * 96 * nt * 2;
  const bool pre = d->pre.scale != nullptr;
  auto launch = [&](auto kern) -> int {
    static bool attr_done[64] = {false};      // one per kernel instantiation (generic lambda)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_done[dev & 63]) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
      if (e != hipSuccess) { vinet_set_error("hipFuncSetAttribute(wgrad_tf): %s", hipGetErrorString(e)); return (int)e; }
      attr_done[dev & 63] = true;
    }
    hipLaunchKernelGGL(kern, dim3(groups * workers), dim3(512), smem, s, a);
    return vn_launch_status("conv_wgrad_tf");
  };
  if (ntw == 6) return pre ? launch(conv_wgrad_tf_kernel<true, 6>) : launch(conv_wgrad_tf_kernel<false, 6>);
  if (ntw == 4) return pre ? launch(conv_wgrad_tf_kernel<true, 4>) : launch(conv_wgrad_tf_kernel<false, 4>);
  return pre ? launch(conv_wgrad_tf_kernel<true, 2>) : launch(conv_wgrad_tf_kernel<false, 2>);
}
