// Weight gradient of a purely TEMPORAL conv (k x 1 x 1, stride (s,1,1)) with 64 input and 64 output channels:
// the partner of the stem, 64 -> 64 7x1x1 / 2 at 32 x 112 x 192 (model_utils.py:149, S3D base1.0.conv_t).
//
//   dw[kt][n][c] += sum_{b,to,h,w} dy[b,to,h,w,n] * pre(x[b, to*s + kt - pad, h, w])[c]
//
// conv_wgrad_dma_kernel<64,64,7> stages one dY tile and SEVEN x tiles per 32 voxels (the seven taps read seven
// different frames): 3.5x the x tensor goes through the CU's load path (43 GB of L2-miss traffic per launch at
// 128 clips for 17 GB of tensors).  Here a workgroup owns 64 (h,w) positions of one clip and walks to = 0..To-1
// with the k live input frames of those positions in an LDS ring: every step fetches only the s NEW frames
// (and one dY tile) -- each input element is staged exactly once.
//
//   * 256 threads = 4 waves; wave w owns input channels [16w, 16w+16) of all k taps and all 64 output channels:
//     k x 4 accumulator tiles (16x16 fp32) = 112 AGPRs at k = 7;
//   * frames go global -> registers -> LDS (not LDS-DMA): the pending BN+ReLU is applied ONCE per element on the
//     way (a thread always handles the same 8 channels: scale/shift live in registers) and frames outside
//     [0, Ti) are written as zeros; loads for step to+1 are issued before the MFMAs of step to;
//   * LDS tiles are [64 positions][64 channels] bf16 with the 16-byte chunk XOR of wgrad_dma.hip; both operands
//     are position-major, K-major fragments come from ds_read_b64_tr_b16;
//   * persistent grid: each workgroup loops over (clip, patch) items and flushes its accumulators once, with
//     fp32 atomics (dw is zero on entry, as for the split-K kernels).
#include "common.h"

struct WgradTsArgs {
  const char* x;
  const char* dy;
  float* dw;
  const float* in_scale;
  const float* in_shift;
  int Ti, To, HW, ldx, ldy;
  long sBx, sBy;
  int k, s, pad, Kp;
  int items, patches;     // items = B * patches, patches = HW / 64
  FastDiv dPatches;
};

// 128-byte rows: a 32-lane transpose read touches rows {a .. a+3, a+8 .. a+11}; rows of equal parity share a 128-byte half of the
// 256-byte bank row, so bits 1 and 3 of the row pick one of its four 32-byte windows (bit 1 alone left rows r and r + 8 on
// the same banks: a 2-way conflict on every ds_read_b64_tr_b16)
VN_DEV int wts_swz(int r) { return (((r >> 1) & 1) | (((r >> 3) & 1) << 1)) << 1; }   // = wg_swz<64> of wgrad_dma.hip

template <bool PRE>
__global__ __launch_bounds__(256, 2) void conv_wgrad_ts_kernel(const WgradTsArgs a) {
  constexpr int KMAX = 7, TILE = 64 * 64 * 2;       // 8 KB
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;                                // KMAX frames
  char* dyb = smem + KMAX * TILE;                   // 2 dY tiles
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = a.k, s = a.s;

  // load role: 16-byte piece (row = position, chunk = 8 channels); two pieces per tile and thread
  const int l_chunk = tid & 7, l_row = tid >> 3;    // rows l_row and l_row + 32
  int l_off[2];                                     // LDS byte offsets of the two pieces
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = l_row + 32 * j;
    l_off[j] = r * 128 + ((l_chunk ^ wts_swz(r)) * 16);
  }
  f32x2_v sc2[4], sh2[4];
  if constexpr (PRE) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sc2[e] = (f32x2_v){a.in_scale[l_chunk * 8 + 2 * e], a.in_scale[l_chunk * 8 + 2 * e + 1]};
      sh2[e] = (f32x2_v){a.in_shift[l_chunk * 8 + 2 * e], a.in_shift[l_chunk * 8 + 2 * e + 1]};
    }
  }
  auto xform = [&](uint4 v) -> uint4 {
    if constexpr (PRE) {
      v.x = pre_relu_pair(v.x, sc2[0], sh2[0]); v.y = pre_relu_pair(v.y, sc2[1], sh2[1]);
      v.z = pre_relu_pair(v.z, sc2[2], sh2[2]); v.w = pre_relu_pair(v.w, sc2[3], sh2[3]);
    }
    return v;
  };

  f32x4_v acc[KMAX][4];
#pragma unroll
  for (int g = 0; g < KMAX; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[g][i] = (f32x4_v){0.f, 0.f, 0.f, 0.f};

  // K-major fragment (8 positions x 1 channel per lane) of rows [32*ks, 32*ks+32) of a tile
  auto frag = [&](const char* tile, int ks, int col0) -> bf16x8_v {
    union { bf16x8_v v; s16x4_v h[2]; } u;
    const int p = lane & 15;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int krow = ks * 32 + (lane >> 4) * 8 + h * 4 + (p >> 2);
      const int col = col0 + (p & 3) * 4;
      const int ch = (col >> 3) ^ wts_swz(krow);
      const char* src = tile + krow * 128 + ch * 16 + (col & 7) * 2;
      u.h[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_v*)src);
    }
    return u.v;
  };

  for (int item = blockIdx.x; item < a.items; item += gridDim.x) {
    const int b = (int)fdiv((uint32_t)item, a.dPatches);
    const int pos0 = (item - b * a.patches) * 64;
    // element offsets of this thread's two pieces inside a frame / dY plane
    const char* xb = a.x + ((long)b * a.sBx + (long)(pos0 + l_row) * a.ldx + l_chunk * 8) * 2;
    const char* db = a.dy + ((long)b * a.sBy + (long)(pos0 + l_row) * a.ldy + l_chunk * 8) * 2;
    const long x_plane = (long)a.HW * a.ldx * 2, d_plane = (long)a.HW * a.ldy * 2;
    const long x_r32 = 32L * a.ldx * 2, d_r32 = 32L * a.ldy * 2;

    // ---- prologue: the k frames of step 0 and dY[0] --------------------------------------------
    for (int g = 0; g < k; ++g) {
      const int p = g - a.pad;
      uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0;
      if ((unsigned)p < (unsigned)a.Ti) {
        v0 = xform(*(const uint4*)(xb + p * x_plane));
        v1 = xform(*(const uint4*)(xb + p * x_plane + x_r32));
      }
      char* slot = ring + ((p + 2 * KMAX) % k) * TILE;
      *(uint4*)(slot + l_off[0]) = v0;
      *(uint4*)(slot + l_off[1]) = v1;
    }
    *(uint4*)(dyb + l_off[0]) = *(const uint4*)db;
    *(uint4*)(dyb + l_off[1]) = *(const uint4*)(db + d_r32);
    __syncthreads();

    for (int to = 0; to < a.To; ++to) {
      // ---- issue the loads of step to+1: s new frames, one dY tile -----------------------------
      const bool more = to + 1 < a.To;
      // (named scalars and unconditional loads: register arrays that are conditionally initialised end up in scratch,
      //  with an s_waitcnt right behind the load -- no prefetch left)
      const int pnew = (to + 1) * s - a.pad + k - s;          // first new frame
      const bool in0 = more && (unsigned)pnew < (unsigned)a.Ti;
      const bool in1 = more && s == 2 && (unsigned)(pnew + 1) < (unsigned)a.Ti;
      const char* xs0 = xb + (in0 ? pnew : 0) * x_plane;
      const char* xs1 = xb + (in1 ? pnew + 1 : 0) * x_plane;
      const char* ds = db + (more ? to + 1 : to) * d_plane;
      const uint4 nx00 = *(const uint4*)xs0, nx01 = *(const uint4*)(xs0 + x_r32);
      const uint4 nx10 = *(const uint4*)xs1, nx11 = *(const uint4*)(xs1 + x_r32);
      const uint4 nd0 = *(const uint4*)ds, nd1 = *(const uint4*)(ds + d_r32);
      // ---- MFMAs of step to ----------------------------------------------------------------------
      const char* dt = dyb + (to & 1) * TILE;
      const int s0 = (to * s - a.pad + 2 * KMAX) % k;       // ring slot of tap 0; tap g sits g slots further (mod k)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8_v af[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = frag(dt, ks, i * 16);
#pragma unroll
        for (int g = 0; g < KMAX; ++g) {
          if (g < k) {
            const int si = s0 + g - (s0 + g >= k ? k : 0);
            const bf16x8_v bf = frag(ring + si * TILE, ks, wave * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) mfma_bf16_acc(acc[g][i], af[i], bf);
          }
        }
      }
      __syncthreads();          // every wave is done with the frames about to be replaced
      if (more) {
        const uint4 z = make_uint4(0, 0, 0, 0);
        char* slot0 = ring + ((pnew + 2 * KMAX) % k) * TILE;
        *(uint4*)(slot0 + l_off[0]) = in0 ? xform(nx00) : z;
        *(uint4*)(slot0 + l_off[1]) = in0 ? xform(nx01) : z;
        if (s == 2) {
          char* slot1 = ring + ((pnew + 1 + 2 * KMAX) % k) * TILE;
          *(uint4*)(slot1 + l_off[0]) = in1 ? xform(nx10) : z;
          *(uint4*)(slot1 + l_off[1]) = in1 ? xform(nx11) : z;
        }
        char* dn = dyb + ((to + 1) & 1) * TILE;
        *(uint4*)(dn + l_off[0]) = nd0;
        *(uint4*)(dn + l_off[1]) = nd1;
      }
      __syncthreads();
    }
  }
  mfma_drain();
  // dw[kt][n][c]: n = i*16 + (lane>>4)*4 + r, c = wave*16 + (lane & 15)
#pragma unroll
  for (int g = 0; g < KMAX; ++g) {
    if (g >= k) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = i * 16 + (lane >> 4) * 4 + r, c = wave * 16 + (lane & 15);
        atomicAdd(a.dw + ((long)g * 64 + n) * (long)a.Kp + c, acc[g][i][r]);
      }
  }
}

extern int g_vinet_opt_wgrad_ts;

// the caller promises temporal taps (VinetWgradDesc::tline); everything else is checked here
bool vinet_wgrad_use_ts(const VinetWgradDesc* d) {
  if (!g_vinet_opt_wgrad_ts || !d->tline || d->dtype != VINET_BF16 || d->mode != VINET_CONV_GENERIC) return false;
  if (d->pre.scale && !(d->pre.relu && d->pre.shift)) return false;
  if (d->pre.relu && !d->pre.scale) return false;
  const long HW = (long)d->dy.H * d->dy.W;
  const bool shape = d->x.C == 64 && d->dy.C == 64 && d->Kp == 64 && d->ntaps >= 2 && d->ntaps <= 7 && (d->sT == 1 || d->sT == 2) &&
                     d->sH == 1 && d->sW == 1 && d->x.H == d->dy.H && d->x.W == d->dy.W && HW % 64 == 0 && d->tpad >= 0 &&
                     d->tpad < d->ntaps && d->ntaps >= d->sT;
  if (!shape) return false;
  // every output frame must only read frames the ring holds: (To-1)*s - pad + k - 1 may exceed Ti-1 (zeros), fine
  if (g_vinet_opt_wgrad_ts >= 2) return true;     // tuning / tests: force
  return (long)d->dy.B * (HW / 64) >= 2048 && d->dy.T >= 4;      // enough items for a persistent grid, a walk worth its prologue
}

int g_vinet_opt_wgrad_ts_cap = 0;   // 1 = the persistent grid honours VinetWgradDesc::max_cus (measured: 677.5 -> 675.6 clips/s -- this launch is on the tail of the step, where the weight-gradient stream is the longer one)
int vinet_launch_wgrad_ts(const VinetWgradDesc* d, hipStream_t s) {
  WgradTsArgs a;
  a.x = (const char*)d->x.ptr; a.dy = (const char*)d->dy.ptr; a.dw = d->dw;
  a.in_scale = d->pre.scale; a.in_shift = d->pre.shift;
  a.Ti = d->x.T; a.To = d->dy.T; a.HW = d->dy.H * d->dy.W; a.ldx = d->x.ld; a.ldy = d->dy.ld;
  a.sBx = d->x.sB; a.sBy = d->dy.sB;
  a.k = d->ntaps; a.s = d->sT; a.pad = d->tpad; a.Kp = d->Kp;
  a.patches = a.HW / 64;
  a.items = d->dy.B * a.patches;
  a.dPatches = make_fastdiv((uint32_t)a.patches);
  const int smem = 9 * 64 * 64 * 2;
  auto kp = conv_wgrad_ts_kernel<true>;
  auto kn = conv_wgrad_ts_kernel<false>;
  static bool attr_done[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_done[dev & 63]) {
    hipError_t e = hipFuncSetAttribute((const void*)kp, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)kn, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) { vinet_set_error("hipFuncSetAttribute(wgrad_ts): %s", hipGetErrorString(e)); return (int)e; }
    attr_done[dev & 63] = true;
  }
  int grid = g_vinet_opt_wgrad_ts_cap ? 2 * vn_wgrad_cus(d) : 512;      // two workgroups per CU, on no more CUs than the caller's cap
  if (grid > a.items) grid = a.items;
  if (d->pre.scale) hipLaunchKernelGGL(kp, dim3(grid), dim3(256), smem, s, a);
  else hipLaunchKernelGGL(kn, dim3(grid), dim3(256), smem, s, a);
  return vn_launch_status("conv_wgrad_ts");
}
