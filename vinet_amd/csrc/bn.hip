// BatchNorm3d / 2d kernels (train and eval, model_utils.py:132,145,149): per-channel statistics and their fold into a
// pending scale / shift, backward reduce + finalize + apply, activation backward, per-channel sums (conv bias gradients).
#include "elementwise.h"

// ============================================================================
// BatchNorm
// ============================================================================
__global__ void bn_finalize_kernel(const float* __restrict__ partials, int rows, int C, int ld, double count,
                                   const float* gamma, const float* beta, float eps, float momentum,
                                   float* running_mean, float* running_var, float* mean_o, float* invstd_o,
                                   float* scale_o, float* shift_o) {
  const int c = blockIdx.x;
  double s = 0.0, q = 0.0;
  for (int r = threadIdx.x; r < rows; r += blockDim.x) {
    s += (double)partials[((long)r * 2 + 0) * ld + c];
    q += (double)partials[((long)r * 2 + 1) * ld + c];
  }
  __shared__ double red[2][4];
  s = wave_sum_d(s); q = wave_sum_d(q);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    q = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = (float)(g * invstd);
    if (mean_o) mean_o[c] = (float)mean;
    if (invstd_o) invstd_o[c] = (float)invstd;
    scale_o[c] = sc;
    shift_o[c] = b - (float)mean * sc;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
  }
}

extern "C" int vinet_bn_finalize(const float* partials, int32_t rows, int32_t C, int32_t ld, double count, const float* gamma,
                                 const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                 float* mean, float* invstd, float* scale, float* shift, void* stream) {
  VN_CHECK_ARG(partials && rows > 0 && C > 0 && (ld == 0 || ld >= C) && count > 0 && scale && shift, "bn_finalize: bad arguments");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, partials, rows, C, ld ? ld : C, count, gamma,
                     beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift);
  return vn_launch_status("bn_finalize");
}

// Pre-reduction of a tall partials table (the 64-channel stem layers produce 172 032 rows at 64 clips: the
// one-workgroup-per-channel finalize would walk them with 4-byte reads 2*C*4 bytes apart).  Block (chunk, 64-channel
// group): 64 channels x 4 row lanes, whole-row coalesced reads, double accumulation, out[chunk][2][C].
__global__ __launch_bounds__(256) void bn_partials_fold_kernel(const float* __restrict__ partials, int rows, int C, int per,
                                                               float* __restrict__ out) {
  const int c = blockIdx.y * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int r0 = blockIdx.x * per;
  int r1 = r0 + per; if (r1 > rows) r1 = rows;
  double s = 0.0, q = 0.0;
  if (c < C) {
    // eight rows per trip, all sixteen loads issued before the first add: the one-row loop waited out a memory latency per
    // row (37 us per launch for a 10 MB table; the adds keep their order, so the sums are bit-identical)
    int r = r0 + rl;
    for (; r + 28 < r1; r += 32) {
      float a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a[u] = partials[((long)(r + 4 * u) * 2 + 0) * C + c];
        b[u] = partials[((long)(r + 4 * u) * 2 + 1) * C + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s += (double)a[u]; q += (double)b[u]; }
    }
    for (; r < r1; r += 4) {
      s += (double)partials[((long)r * 2 + 0) * C + c];
      q += (double)partials[((long)r * 2 + 1) * C + c];
    }
  }
  __shared__ double red[2][4][64];
  red[0][rl][threadIdx.x & 63] = s; red[1][rl][threadIdx.x & 63] = q;
  __syncthreads();
  if (rl == 0 && c < C) {
    const int l = threadIdx.x;
    out[((long)blockIdx.x * 2 + 0) * C + c] = (float)(red[0][0][l] + red[0][1][l] + red[0][2][l] + red[0][3][l]);
    out[((long)blockIdx.x * 2 + 1) * C + c] = (float)(red[1][0][l] + red[1][1][l] + red[1][2][l] + red[1][3][l]);
  }
}

extern "C" int vinet_bn_partials_fold(const float* partials, int32_t rows, int32_t C, float* out, int32_t out_rows, void* stream) {
  VN_CHECK_ARG(partials && out && rows > 0 && C > 0 && out_rows > 0 && out_rows <= rows, "bn_partials_fold: bad arguments");
  const int per = (rows + out_rows - 1) / out_rows;
  VN_CHECK_ARG((long)(out_rows - 1) * per < rows, "bn_partials_fold: out_rows=%d leaves empty chunks for rows=%d", out_rows, rows);
  hipLaunchKernelGGL(bn_partials_fold_kernel, dim3(out_rows, (C + 63) / 64), dim3(256), 0, (hipStream_t)stream, partials, rows, C, per, out);
  return vn_launch_status("bn_partials_fold");
}

__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                               const float* conv_bias, float eps, int C, float* scale, float* shift, float* invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float is = 1.f / sqrtf(rv[c] + eps);
  const float sc = (gamma ? gamma[c] : 1.f) * is;
  scale[c] = sc;
  shift[c] = (beta ? beta[c] : 0.f) + ((conv_bias ? conv_bias[c] : 0.f) - rm[c]) * sc;
  if (invstd) invstd[c] = is;
}

extern "C" int vinet_bn_fold(const float* gamma, const float* beta, const float* running_mean,
                             const float* running_var, const float* conv_bias, float eps, int32_t C, float* scale,
                             float* shift, float* invstd, void* stream) {
  VN_CHECK_ARG(running_mean && running_var && scale && shift && C > 0, "bn_fold: bad arguments");
  hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                     running_mean, running_var, conv_bias, eps, C, scale, shift, invstd);
  return vn_launch_status("bn_fold");
}

// Per-channel reductions over voxels.  Thread (q, r): channel quad q, voxel lane
// r; a block covers `vb` consecutive voxels and writes one partial row.
// MODE 0: (sum x, sum x^2) of x;  MODE 1: (sum dz*mask, sum dz*mask*xhat).

// 8-channel form of channel_reduce_kernel: same partials contract ([rows][2][C], block b owns voxels
// [b*vb, (b+1)*vb)), twice the bytes per load instruction.
template <typename T, int MODE>
__global__ __launch_bounds__(256) VN_NO_PK_F32 void channel_reduce8_kernel(TView x, TView dz, Affine fwd, const float* mean,
                                                              const float* invstd, long nvox, long vb,
                                                              float* __restrict__ partials) {
  const int G = x.C / 8;
  const int Gb = G < 256 ? G : 256;
  const int R = 256 / Gb;
  const int r = threadIdx.x / Gb;
  const int g0 = threadIdx.x % Gb;
  __shared__ float red[256 * 16];
  constexpr int U = 4;
  // vb == 0: interleaved rounds -- in round i block b reads voxels [(i*gridDim.x + b)*R*U, +R*U), so the whole grid
  // walks one contiguous window of the tensor instead of gridDim.x streams a fixed stride apart
  const long v0 = vb ? (long)blockIdx.x * vb : (long)blockIdx.x * R * U;
  long v1 = vb ? v0 + vb : nvox; if (v1 > nvox) v1 = nvox;
  const long vstep = vb ? (long)R * U : (long)gridDim.x * R * U;
  for (int g = g0; g < G; g += Gb) {
    float s[8], p[8], mu[8], is[8], sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; p[e] = 0.f; mu[e] = 0.f; is[e] = 1.f; sc[e] = 1.f; sh[e] = 0.f; }
    if (r < R) {
      if (MODE == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { mu[e] = mean[g * 8 + e]; is[e] = invstd[g * 8 + e]; }
        if (fwd.relu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { sc[e] = fwd.scale ? fwd.scale[g * 8 + e] : 1.f; sh[e] = fwd.shift ? fwd.shift[g * 8 + e] : 0.f; }
        }
      }
      for (long vq = v0 + r; vq < v1; vq += vstep) {
        float xv[U][8], gv[U][8];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const long v = vq + (long)u * R;
          ok[u] = v < v1;
          if (ok[u]) {
            if (MODE == 1) {      // (the backward reduce pass: both tensors are GBs and come back a pass later -- non-temporal)
              ld8_nt<T>((const T*)x.p + vox_lin(x, v) + g * 8, xv[u]);
              ld8_nt<T>((const T*)dz.p + vox_lin(dz, v) + g * 8, gv[u]);
            } else {
              ld8<T>((const T*)x.p + vox_lin(x, v) + g * 8, xv[u]);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (!ok[u]) continue;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            if (MODE == 0) {
              s[e] += xv[u][e]; p[e] += xv[u][e] * xv[u][e];
            } else {
              float gg = gv[u][e];
              if (fwd.relu && !(fmaf(xv[u][e], sc[e], sh[e]) > 0.f)) gg = 0.f;
              s[e] += gg;
              p[e] += gg * (xv[u][e] - mu[e]) * is[e];
            }
          }
        }
      }
    }
    __syncthreads();
    if (r < R) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { red[threadIdx.x * 16 + e] = s[e]; red[threadIdx.x * 16 + 8 + e] = p[e]; }
    }
    __syncthreads();
    if (r == 0) {
      for (int rr = 1; rr < R; ++rr)
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] += red[(rr * Gb + g0) * 16 + e]; p[e] += red[(rr * Gb + g0) * 16 + 8 + e]; }
      float* o = partials + (long)blockIdx.x * 2 * x.C + g * 8;
      *(float4*)o = make_float4(s[0], s[1], s[2], s[3]);
      *(float4*)(o + 4) = make_float4(s[4], s[5], s[6], s[7]);
      *(float4*)(o + x.C) = make_float4(p[0], p[1], p[2], p[3]);
      *(float4*)(o + x.C + 4) = make_float4(p[4], p[5], p[6], p[7]);
    }
  }
}
// Register-lean bf16 form of channel_reduce8_kernel<bf16_t, 1> (round 3).  The generic form unpacks its 4 x 2 loads of
// 16 bytes into 64 floats before it uses them: 170 VGPRs, two waves per SIMD -- and ONE beside a resident weight-gradient
// workgroup (the persistent wgrad kernels of the second stream hold 2 x 168...254 of a SIMD's 512 registers), which is why
// BatchNorm backward ran at 2.7 TB/s inside the step against 5.0 alone.  Here the loads stay packed (4 registers each) until
// the voxel is consumed; same arithmetic, same order of operations per lane, same partials contract.
template <int U, int WPE>
__global__ __launch_bounds__(256, WPE) void bn_bwd_reduce8_bf16_kernel(TView x, TView dz, Affine fwd, const float* mean,
                                                                  const float* invstd, long nvox, long vb,
                                                                  float* __restrict__ partials) {
  const int G = x.C / 8;
  const int Gb = G < 256 ? G : 256;
  const int R = 256 / Gb;
  const int r = threadIdx.x / Gb;
  const int g0 = threadIdx.x % Gb;
  __shared__ float red[256 * 16];
  const long v0 = vb ? (long)blockIdx.x * vb : (long)blockIdx.x * R * U;
  long v1 = vb ? v0 + vb : nvox; if (v1 > nvox) v1 = nvox;
  const long vstep = vb ? (long)R * U : (long)gridDim.x * R * U;
  const bool relu = fwd.relu != 0;
  for (int g = g0; g < G; g += Gb) {
    float s[8], p[8], mu[8], is[8], sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; p[e] = 0.f; mu[e] = 0.f; is[e] = 1.f; sc[e] = 1.f; sh[e] = 0.f; }
    if (r < R) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { mu[e] = mean[g * 8 + e]; is[e] = invstd[g * 8 + e]; }
      if (relu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = fwd.scale ? fwd.scale[g * 8 + e] : 1.f; sh[e] = fwd.shift ? fwd.shift[g * 8 + e] : 0.f; }
      }
      for (long vq = v0 + r; vq < v1; vq += vstep) {
        uint4 xr[U], gr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const long v = vq + (long)u * R;
          const long vc = v < v1 ? v : vq;                 // clamped: the load is unconditional, the use is not
          // (GB-sized tensors: by the time the apply pass re-reads them they have left every cache anyway -- non-temporal: 15.6 -> 14.5 ms
          //  for the step's 58 launches)
          xr[u] = ld16_nt((const bf16_t*)x.p + vox_lin(x, vc) + g * 8);
          gr[u] = ld16_nt((const bf16_t*)dz.p + vox_lin(dz, vc) + g * 8);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (vq + (long)u * R >= v1) continue;
          const uint32_t xw[4] = {xr[u].x, xr[u].y, xr[u].z, xr[u].w}, gw[4] = {gr[u].x, gr[u].y, gr[u].z, gr[u].w};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float xv = __uint_as_float((e & 1) ? (xw[e >> 1] & 0xffff0000u) : (xw[e >> 1] << 16));
            float gg = __uint_as_float((e & 1) ? (gw[e >> 1] & 0xffff0000u) : (gw[e >> 1] << 16));
            if (relu && !(fmaf(xv, sc[e], sh[e]) > 0.f)) gg = 0.f;
            s[e] += gg;
            p[e] += gg * (xv - mu[e]) * is[e];
          }
        }
      }
    }
    __syncthreads();
    if (r < R) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { red[threadIdx.x * 16 + e] = s[e]; red[threadIdx.x * 16 + 8 + e] = p[e]; }
    }
    __syncthreads();
    if (r == 0) {
      for (int rr = 1; rr < R; ++rr)
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] += red[(rr * Gb + g0) * 16 + e]; p[e] += red[(rr * Gb + g0) * 16 + 8 + e]; }
      float* o = partials + (long)blockIdx.x * 2 * x.C + g * 8;
      *(float4*)o = make_float4(s[0], s[1], s[2], s[3]);
      *(float4*)(o + 4) = make_float4(s[4], s[5], s[6], s[7]);
      *(float4*)(o + x.C) = make_float4(p[0], p[1], p[2], p[3]);
      *(float4*)(o + x.C + 4) = make_float4(p[4], p[5], p[6], p[7]);
    }
  }
}
int g_vinet_opt_bn_lean = 1;    // register-lean bf16 BatchNorm-backward kernels (0 = the generic 8-channel forms)

int g_vinet_opt_bn_rows = 1024;  // cap on the workgroups (= partial rows) of a channel reduction
static inline int stats_rows_for(long nvox) {
  long rows = (nvox + 63) / 64;
  if (rows > g_vinet_opt_bn_rows) rows = g_vinet_opt_bn_rows;
  if (rows < 1) rows = 1;
  return (int)rows;
}
extern "C" int vinet_stats_rows(const VinetTensor* x) { return x ? stats_rows_for(view_voxels(*x)) : -1; }

template <typename T, int MODE>
__global__ __launch_bounds__(256) void channel_reduce_kernel(TView x, TView dz, Affine fwd, const float* mean,
                                                             const float* invstd, long nvox, long vb,
                                                             float* __restrict__ partials) {
  const int Q = x.C / 4;
  const int Qb = Q < 256 ? Q : 256;
  const int R = 256 / Qb;
  const int r = threadIdx.x / Qb;
  const int q0 = threadIdx.x % Qb;
  __shared__ float red[256 * 8];
  const long v0 = (long)blockIdx.x * vb;
  long v1 = v0 + vb; if (v1 > nvox) v1 = nvox;
  for (int q = q0; q < Q; q += Qb) {
    float s[4] = {0, 0, 0, 0}, p[4] = {0, 0, 0, 0};
    if (r < R) {
      float4 mu = make_float4(0, 0, 0, 0), is = make_float4(1, 1, 1, 1);
      if (MODE == 1) { mu = *(const float4*)(mean + q * 4); is = *(const float4*)(invstd + q * 4); }
      constexpr int U = 4;   // independent voxels per iteration: 2*U loads in flight per lane
      for (long vb = v0 + r; vb < v1; vb += (long)R * U) {
        float4 xv[U], gv[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const long v = vb + (long)u * R;
          ok[u] = v < v1;
          xv[u] = make_float4(0, 0, 0, 0); gv[u] = make_float4(0, 0, 0, 0);
          if (ok[u]) {
            xv[u] = ldq<T>((const T*)x.p + vox_lin(x, v) + q * 4);
            if (MODE == 1) gv[u] = ldq<T>((const T*)dz.p + vox_lin(dz, v) + q * 4);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (!ok[u]) continue;
          if (MODE == 0) {
            s[0] += xv[u].x; s[1] += xv[u].y; s[2] += xv[u].z; s[3] += xv[u].w;
            p[0] += xv[u].x * xv[u].x; p[1] += xv[u].y * xv[u].y; p[2] += xv[u].z * xv[u].z; p[3] += xv[u].w * xv[u].w;
          } else {
            float4 g = gv[u];
            if (fwd.relu) {
              Affine na = fwd; na.relu = 0;
              const float4 z = affine4(xv[u], na, q * 4);
              if (!(z.x > 0.f)) g.x = 0.f;
              if (!(z.y > 0.f)) g.y = 0.f;
              if (!(z.z > 0.f)) g.z = 0.f;
              if (!(z.w > 0.f)) g.w = 0.f;
            }
            s[0] += g.x; s[1] += g.y; s[2] += g.z; s[3] += g.w;
            p[0] += g.x * (xv[u].x - mu.x) * is.x; p[1] += g.y * (xv[u].y - mu.y) * is.y;
            p[2] += g.z * (xv[u].z - mu.z) * is.z; p[3] += g.w * (xv[u].w - mu.w) * is.w;
          }
        }
      }
    }
    __syncthreads();
    if (r < R) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { red[threadIdx.x * 8 + e] = s[e]; red[threadIdx.x * 8 + 4 + e] = p[e]; }
    }
    __syncthreads();
    if (r == 0) {
      for (int rr = 1; rr < R; ++rr)
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[e] += red[(rr * Qb + q0) * 8 + e]; p[e] += red[(rr * Qb + q0) * 8 + 4 + e]; }
      float* o = partials + (long)blockIdx.x * 2 * x.C + q * 4;
      *(float4*)o = make_float4(s[0], s[1], s[2], s[3]);
      *(float4*)(o + x.C) = make_float4(p[0], p[1], p[2], p[3]);
    }
  }
}

// Tiny tensors (at most 64 voxels: ONE partial row -- the last SoundNet layers at small batches, 6 x 1024 at two clips): a thread
// per channel walks the voxels in order.  No LDS, no cross-thread step, one exit.  (Round 5: this kernel was the first mitigation of
// the run-to-run mismatch of SoundNet's last-layer gradients, before its cause was found -- the packed-fp32 erratum described in
// common.h, which hit channel_reduce8_kernel<bf16, 0> and nothing else in the library.  It stays: a single 64-thread wave per 64
// channels is also the cheaper launch for such a tensor.)
template <typename T, int MODE>
__global__ __launch_bounds__(64) void channel_reduce_small_kernel(TView x, TView dz, Affine fwd, const float* mean, const float* invstd,
                                                                  int nvox, float* __restrict__ partials) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= x.C) return;
  float s = 0.f, p = 0.f;
  const float mu = MODE == 1 ? mean[c] : 0.f, is = MODE == 1 ? invstd[c] : 1.f;
  const float sc = (MODE == 1 && fwd.relu && fwd.scale) ? fwd.scale[c] : 1.f, sh = (MODE == 1 && fwd.relu && fwd.shift) ? fwd.shift[c] : 0.f;
  for (int v = 0; v < nvox; ++v) {
    const float xv = load1<T>((const T*)x.p + vox_lin(x, v) + c);
    if (MODE == 0) {
      s += xv; p += xv * xv;
    } else {
      float gg = load1<T>((const T*)dz.p + vox_lin(dz, v) + c);
      if (fwd.relu && !(fmaf(xv, sc, sh) > 0.f)) gg = 0.f;
      s += gg;
      p += gg * (xv - mu) * is;
    }
  }
  partials[c] = s;
  partials[x.C + c] = p;
}

int g_vinet_opt_reduce_small = 1;   // tensors of <= 64 voxels take channel_reduce_small_kernel
template <int MODE>
static int launch_channel_reduce(const VinetTensor* x, const VinetTensor* dz, int dtype, VinetAffine fwd,
                                 const float* mean, const float* invstd, float* partials, void* stream) {
  const long nvox = view_voxels(*x);
  const int rows = stats_rows_for(nvox);
  if (g_vinet_opt_reduce_small && nvox <= 64 && rows == 1) {
    const TView xs = make_view(*x), ds = dz ? make_view(*dz) : xs;
    DISPATCH_T(dtype, T, hipLaunchKernelGGL((channel_reduce_small_kernel<T, MODE>), dim3((x->C + 63) / 64), dim3(64), 0,
                                            (hipStream_t)stream, xs, ds, make_affine(fwd), mean, invstd, (int)nvox, partials);)
    return vn_launch_status("channel_reduce_small");
  }
  const long vb = (nvox + rows - 1) / rows;
  const TView xv = make_view(*x), dv = dz ? make_view(*dz) : xv;
  if (MODE == 1 && g_vinet_opt_bn_lean && dtype == VINET_BF16 && dz && oct_ok(*x) && oct_ok(*dz)) {
#ifdef VINET_EXPERIMENTS
    if (g_vinet_opt_bn_lean == 2)
      hipLaunchKernelGGL((bn_bwd_reduce8_bf16_kernel<2, 6>), dim3(rows), dim3(256), 0, (hipStream_t)stream, xv, dv, make_affine(fwd), mean,
                         invstd, nvox, g_vinet_opt_reduce_il ? 0 : vb, partials);
    else
#endif
      hipLaunchKernelGGL((bn_bwd_reduce8_bf16_kernel<4, 4>), dim3(rows), dim3(256), 0, (hipStream_t)stream, xv, dv, make_affine(fwd), mean,
                         invstd, nvox, g_vinet_opt_reduce_il ? 0 : vb, partials);
    return vn_launch_status("bn_bwd_reduce8_bf16");
  }
  if (oct_ok(*x) && (!dz || oct_ok(*dz))) {
    DISPATCH_T(dtype, T, hipLaunchKernelGGL((channel_reduce8_kernel<T, MODE>), dim3(rows), dim3(256), 0,
                                            (hipStream_t)stream, xv, dv, make_affine(fwd), mean, invstd, nvox, g_vinet_opt_reduce_il ? 0 : vb, partials);)
    return vn_launch_status("channel_reduce8");
  }
  DISPATCH_T(dtype, T, hipLaunchKernelGGL((channel_reduce_kernel<T, MODE>), dim3(rows), dim3(256), 0,
                                          (hipStream_t)stream, xv, dv, make_affine(fwd), mean, invstd, nvox, vb, partials);)
  return vn_launch_status("channel_reduce");
}

extern "C" int vinet_channel_stats(const VinetTensor* x, int32_t dtype, float* partials, void* stream) {
  VN_CHECK_ARG(x && partials && quad_ok(*x, esize(dtype)), "channel_stats: bad arguments");
  VinetAffine none = {nullptr, nullptr, 0};
  return launch_channel_reduce<0>(x, nullptr, dtype, none, nullptr, nullptr, partials, stream);
}

extern "C" int vinet_bn_bwd_reduce(const VinetTensor* dz, const VinetTensor* x_raw, int32_t dtype, VinetAffine fwd,
                                   const float* mean, const float* invstd, float* partials, void* stream) {
  VN_CHECK_ARG(dz && x_raw && partials && mean && invstd && quad_ok(*dz, esize(dtype)) && quad_ok(*x_raw, esize(dtype)) &&
                   same_dims(*dz, *x_raw), "bn_bwd_reduce: bad arguments");
  return launch_channel_reduce<1>(x_raw, dz, dtype, fwd, mean, invstd, partials, stream);
}

// one workgroup per channel: rows are reduced in parallel (fp64), lane 0 finishes
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partials, int rows, int C, int ld,
                                                              double count, const float* scale, int train, float* dgamma,
                                                              float* dbeta, const float* invstd, float* c1, float* c2) {
  const int c = blockIdx.x;
  double s = 0.0, p = 0.0;
  for (int r = threadIdx.x; r < rows; r += blockDim.x) {
    s += (double)partials[((long)r * 2) * ld + c];
    p += (double)partials[((long)r * 2 + 1) * ld + c];
  }
  __shared__ double red[2][4];
  s = wave_sum_d(s); p = wave_sum_d(p);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = p; }
  __syncthreads();
  if (threadIdx.x == 0) {
    s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    p = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    // (atomics, not `x[c] += v`: the read of a plain read-modify-write of this uniform address is a SCALAR load, i.e. it goes
    //  through the scalar data cache, while the accumulators are the optimizer's flat gradient buffer, which other kernels
    //  rewrite between steps.  Introduced in round 5 while hunting the run-to-run gradient mismatch; that turned out to be the
    //  packed-fp32 erratum of common.h, not this -- the atomics stay as the form that does not depend on cache behaviour.)
    if (dgamma) atomicAdd(dgamma + c, (float)p);
    if (dbeta) atomicAdd(dbeta + c, (float)s);
    if (c1) c1[c] = train ? (float)(s / count) : 0.f;
    if (c2) c2[c] = train ? (float)(p / count) : 0.f;
  }
}

extern "C" int vinet_bn_bwd_finalize(const float* partials, int32_t rows, int32_t C, int32_t ld, double count, const float* scale,
                                     int32_t train, float* dgamma_acc, float* dbeta_acc, const float* invstd, float* c1,
                                     float* c2, void* stream) {
  VN_CHECK_ARG(partials && rows > 0 && C > 0 && (ld == 0 || ld >= C) && count > 0, "bn_bwd_finalize: bad arguments");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, partials, rows, C, ld ? ld : C, count,
                     scale, train, dgamma_acc, dbeta_acc, invstd, c1, c2);
  return vn_launch_status("bn_bwd_finalize");
}

template <typename T>
__global__ void bn_bwd_apply_kernel(TView dz, TView x, Affine fwd, const float* mean, const float* invstd,
                                    const float* c1, const float* c2, TView dx, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint32_t vox_u = fdiv((uint32_t)i, x.dQ);
  const int q = (int)((uint32_t)i - vox_u * (uint32_t)(x.C / 4));
  const long vox = (long)vox_u;
  const float4 xv = ldq<T>((const T*)x.p + vox_lin(x, vox) + q * 4);
  float4 g = ldq<T>((const T*)dz.p + vox_lin(dz, vox) + q * 4);
  const float4 sc = *(const float4*)(fwd.scale + q * 4);
  if (fwd.relu) {
    const float4 sh = *(const float4*)(fwd.shift + q * 4);
    if (!(fmaf(xv.x, sc.x, sh.x) > 0.f)) g.x = 0.f;
    if (!(fmaf(xv.y, sc.y, sh.y) > 0.f)) g.y = 0.f;
    if (!(fmaf(xv.z, sc.z, sh.z) > 0.f)) g.z = 0.f;
    if (!(fmaf(xv.w, sc.w, sh.w) > 0.f)) g.w = 0.f;
  }
  const float4 mu = *(const float4*)(mean + q * 4), is = *(const float4*)(invstd + q * 4);
  const float4 a1 = *(const float4*)(c1 + q * 4), a2 = *(const float4*)(c2 + q * 4);
  float4 o;
  o.x = sc.x * (g.x - a1.x - (xv.x - mu.x) * is.x * a2.x);
  o.y = sc.y * (g.y - a1.y - (xv.y - mu.y) * is.y * a2.y);
  o.z = sc.z * (g.z - a1.z - (xv.z - mu.z) * is.z * a2.z);
  o.w = sc.w * (g.w - a1.w - (xv.w - mu.w) * is.w * a2.w);
  stq<T>((T*)dx.p + vox_lin(dx, vox) + q * 4, o);
}

// 8-channel, voxel-looping form: the six per-channel parameter vectors are folded into four
// coefficients held in registers (dx = A*g + B*x + D with the ReLU gate from sc*x + sh), so a
// lane issues two 16-byte loads and one 16-byte store per voxel and nothing else.
// SPLIT (T = float, the VINET_F32S training form): the result also leaves as hi / lo bf16 planes (hi = bf16(v), lo = bf16(v - hi):
// vinet_split_bf16's arithmetic) -- the operand planes of the three bf16 weight-gradient launches, without a second pass over dx
template <typename T, bool SPLIT>
__global__ __launch_bounds__(256) void bn_bwd_apply8_kernel(TView dz, TView x, Affine fwd, const float* mean, const float* invstd,
                                                            const float* c1, const float* c2, TView dx, TView hi, TView lo, long nvox, long vb) {
  const int G = x.C / 8;
  const int Gb = G < 256 ? G : 256;
  const int R = 256 / Gb;
  const int r = threadIdx.x / Gb;
  const int g0 = threadIdx.x % Gb;
  if (r >= R) return;
  const long v0 = (long)blockIdx.x * vb;
  long v1 = v0 + vb; if (v1 > nvox) v1 = nvox;
  for (int g = g0; g < G; g += Gb) {
    float A[8], Bc[8], D[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = g * 8 + e;
      const float sc = fwd.scale[c], k = sc * invstd[c] * c2[c];
      A[e] = sc; Bc[e] = -k; D[e] = fmaf(k, mean[c], -sc * c1[c]);
      sh[e] = fwd.shift[c];
    }
    constexpr int U = 4;
    for (long vq = v0 + r; vq < v1; vq += (long)R * U) {
      float xv[U][8], gv[U][8];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long v = vq + (long)u * R;
        ok[u] = v < v1;
        if (ok[u]) {
          ld8_nt<T>((const T*)x.p + vox_lin(x, v) + g * 8, xv[u]);      // (z is dead after this pass; dx and its planes are read a pass later)
          ld8<T>((const T*)dz.p + vox_lin(dz, v) + g * 8, gv[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!ok[u]) continue;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float gg = gv[u][e];
          if (fwd.relu && !(fmaf(xv[u][e], A[e], sh[e]) > 0.f)) gg = 0.f;
          o[e] = fmaf(A[e], gg, fmaf(Bc[e], xv[u][e], D[e]));
        }
        st8_nt<T>((T*)dx.p + vox_lin(dx, vq + (long)u * R) + g * 8, o);
        if constexpr (SPLIT) {
          uint32_t h[4], l[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            h[e] = pack2bf(o[2 * e], o[2 * e + 1]);
            l[e] = pack2bf(o[2 * e] - __uint_as_float(h[e] << 16), o[2 * e + 1] - __uint_as_float(h[e] & 0xffff0000u));
          }
          st16_nt((bf16_t*)hi.p + vox_lin(hi, vq + (long)u * R) + g * 8, make_uint4(h[0], h[1], h[2], h[3]));
          st16_nt((bf16_t*)lo.p + vox_lin(lo, vq + (long)u * R) + g * 8, make_uint4(l[0], l[1], l[2], l[3]));
        }
      }
    }
  }
}

// register-lean bf16 form of bn_bwd_apply8_kernel (162 VGPRs there): packed loads, one voxel unpacked at a time
template <int U, int WPE>
__global__ __launch_bounds__(256, WPE) void bn_bwd_apply8_bf16_kernel(TView dz, TView x, Affine fwd, const float* mean, const float* invstd,
                                                                 const float* c1, const float* c2, TView dx, long nvox, long vb) {
  const int G = x.C / 8;
  const int Gb = G < 256 ? G : 256;
  const int R = 256 / Gb;
  const int r = threadIdx.x / Gb;
  const int g0 = threadIdx.x % Gb;
  if (r >= R) return;
  const long v0 = (long)blockIdx.x * vb;
  long v1 = v0 + vb; if (v1 > nvox) v1 = nvox;
  const bool relu = fwd.relu != 0;
  for (int g = g0; g < G; g += Gb) {
    float A[8], Bc[8], D[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = g * 8 + e;
      const float sc = fwd.scale[c], k = sc * invstd[c] * c2[c];
      A[e] = sc; Bc[e] = -k; D[e] = fmaf(k, mean[c], -sc * c1[c]);
      sh[e] = fwd.shift[c];
    }
    for (long vq = v0 + r; vq < v1; vq += (long)R * U) {
      uint4 xr[U], gr[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long v = vq + (long)u * R;
        const long vc = v < v1 ? v : vq;
        // (z is dead after this pass and dx is next read a pass later: non-temporal both -- 26.1 -> 25.1 ms for the step's 58 launches)
        xr[u] = ld16_nt((const bf16_t*)x.p + vox_lin(x, vc) + g * 8);
        gr[u] = *(const uint4*)((const bf16_t*)dz.p + vox_lin(dz, vc) + g * 8);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (vq + (long)u * R >= v1) continue;
        const uint32_t xw[4] = {xr[u].x, xr[u].y, xr[u].z, xr[u].w}, gw[4] = {gr[u].x, gr[u].y, gr[u].z, gr[u].w};
        uint32_t ow[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          float o2[2];
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2) {
            const int e = 2 * h + k2;
            const float xv = __uint_as_float(k2 ? (xw[h] & 0xffff0000u) : (xw[h] << 16));
            float gg = __uint_as_float(k2 ? (gw[h] & 0xffff0000u) : (gw[h] << 16));
            if (relu && !(fmaf(xv, A[e], sh[e]) > 0.f)) gg = 0.f;
            o2[k2] = fmaf(A[e], gg, fmaf(Bc[e], xv, D[e]));
          }
          ow[h] = pack2bf(o2[0], o2[1]);
        }
        st16_nt((bf16_t*)dx.p + vox_lin(dx, vq + (long)u * R) + g * 8, make_uint4(ow[0], ow[1], ow[2], ow[3]));
      }
    }
  }
}

extern "C" int vinet_bn_bwd_apply(const VinetTensor* dz, const VinetTensor* x_raw, int32_t dtype, VinetAffine fwd,
                                  const float* mean, const float* invstd, const float* c1, const float* c2,
                                  const VinetTensor* dx, void* stream) {
  VN_CHECK_ARG(dz && x_raw && dx && fwd.scale && fwd.shift && mean && invstd && c1 && c2, "bn_bwd_apply: null argument");
  VN_CHECK_ARG(quad_ok(*dz, esize(dtype)) && quad_ok(*x_raw, esize(dtype)) && quad_ok(*dx, esize(dtype)) &&
                   same_dims(*dz, *x_raw) && same_dims(*dz, *dx), "bn_bwd_apply: bad views");
  if (oct_ok(*dz) && oct_ok(*x_raw) && oct_ok(*dx)) {
    const long nvox = view_voxels(*dz);
    const int G = dz->C / 8, R = 256 / (G < 256 ? G : 256);
    long vb = R * 16;                                  // 4 rounds of 4 voxels per lane
    while ((nvox + vb - 1) / vb > 16384) vb *= 2;
    if (g_vinet_opt_bn_lean && dtype == VINET_BF16) {
#ifdef VINET_EXPERIMENTS
      if (g_vinet_opt_bn_lean == 2)
        hipLaunchKernelGGL((bn_bwd_apply8_bf16_kernel<2, 6>), dim3((unsigned)((nvox + vb - 1) / vb)), dim3(256), 0, (hipStream_t)stream,
                           make_view(*dz), make_view(*x_raw), make_affine(fwd), mean, invstd, c1, c2, make_view(*dx), nvox, vb);
      else
#endif
        hipLaunchKernelGGL((bn_bwd_apply8_bf16_kernel<4, 4>), dim3((unsigned)((nvox + vb - 1) / vb)), dim3(256), 0, (hipStream_t)stream,
                           make_view(*dz), make_view(*x_raw), make_affine(fwd), mean, invstd, c1, c2, make_view(*dx), nvox, vb);
      return vn_launch_status("bn_bwd_apply8_bf16");
    }
    DISPATCH_T(dtype, T, hipLaunchKernelGGL((bn_bwd_apply8_kernel<T, false>), dim3((unsigned)((nvox + vb - 1) / vb)), dim3(256), 0,
                                            (hipStream_t)stream, make_view(*dz), make_view(*x_raw), make_affine(fwd), mean,
                                            invstd, c1, c2, make_view(*dx), make_view(*dx), make_view(*dx), nvox, vb);)
    return vn_launch_status("bn_bwd_apply8");
  }
  const long total = view_voxels(*dz) * (dz->C / 4);
  DISPATCH_T(dtype, T, hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3(ew_grid(total)), dim3(256), 0,
                                          (hipStream_t)stream, make_view(*dz), make_view(*x_raw), make_affine(fwd), mean,
                                          invstd, c1, c2, make_view(*dx), total);)
  return vn_launch_status("bn_bwd_apply");
}

/* fp32 tensors (the VINET_F32S training form): vinet_bn_bwd_apply + vinet_split_bf16 of its result in one pass -- dx as usual, and
 * its hi / lo bf16 planes (views of dx's extent) for the split-bf16 weight gradient.  Channel counts in whole groups of 8. */
extern "C" int vinet_bn_bwd_apply_split(const VinetTensor* dz, const VinetTensor* x_raw, VinetAffine fwd, const float* mean,
                                        const float* invstd, const float* c1, const float* c2, const VinetTensor* dx,
                                        const VinetTensor* hi, const VinetTensor* lo, void* stream) {
  VN_CHECK_ARG(dz && x_raw && dx && hi && lo && fwd.scale && fwd.shift && mean && invstd && c1 && c2, "bn_bwd_apply_split: null argument");
  VN_CHECK_ARG(oct_ok(*dz) && oct_ok(*x_raw) && oct_ok(*dx) && oct_ok(*hi) && oct_ok(*lo) && same_dims(*dz, *x_raw) && same_dims(*dz, *dx) &&
                   same_dims(*dz, *hi) && same_dims(*dz, *lo) && quad_ok(*dz, 4) && quad_ok(*x_raw, 4) && quad_ok(*dx, 4) && quad_ok(*hi, 2) &&
                   quad_ok(*lo, 2), "bn_bwd_apply_split: bad views (fp32 tensors, bf16 planes, C a multiple of 8)");
  const long nvox = view_voxels(*dz);
  const int G = dz->C / 8, R = 256 / (G < 256 ? G : 256);
  long vb = R * 16;
  while ((nvox + vb - 1) / vb > 16384) vb *= 2;
  hipLaunchKernelGGL((bn_bwd_apply8_kernel<float, true>), dim3((unsigned)((nvox + vb - 1) / vb)), dim3(256), 0, (hipStream_t)stream,
                     make_view(*dz), make_view(*x_raw), make_affine(fwd), mean, invstd, c1, c2, make_view(*dx), make_view(*hi), make_view(*lo), nvox, vb);
  return vn_launch_status("bn_bwd_apply8<split>");
}

template <typename TG, typename TZ, typename TO>
__global__ void act_bwd_kernel(TView dz, TView z, int act, TView dy, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint32_t vox_u = fdiv((uint32_t)i, z.dQ);
  const int q = (int)((uint32_t)i - vox_u * (uint32_t)(z.C / 4));
  const long vox = (long)vox_u;
  const float4 zv = ldq<TZ>((const TZ*)z.p + vox_lin(z, vox) + q * 4);
  float4 g = ldq<TG>((const TG*)dz.p + vox_lin(dz, vox) + q * 4);
  if (act == VINET_ACT_RELU) {
    if (!(zv.x > 0.f)) g.x = 0.f;
    if (!(zv.y > 0.f)) g.y = 0.f;
    if (!(zv.z > 0.f)) g.z = 0.f;
    if (!(zv.w > 0.f)) g.w = 0.f;
  } else if (act == VINET_ACT_SIGMOID) {
    g.x *= zv.x * (1.f - zv.x); g.y *= zv.y * (1.f - zv.y); g.z *= zv.z * (1.f - zv.z); g.w *= zv.w * (1.f - zv.w);
  }
  stq<TO>((TO*)dy.p + vox_lin(dy, vox) + q * 4, g);
}

extern "C" int vinet_act_bwd(const VinetTensor* dz, int32_t dz_dtype, const VinetTensor* z, int32_t z_dtype,
                             int32_t act, const VinetTensor* dy, int32_t dy_dtype, void* stream) {
  VN_CHECK_ARG(dz && z && dy && quad_ok(*dz, esize(dz_dtype)) && quad_ok(*z, esize(z_dtype)) &&
                   quad_ok(*dy, esize(dy_dtype)) && same_dims(*dz, *z) && same_dims(*dz, *dy), "act_bwd: bad views");
  const long total = view_voxels(*z) * (z->C / 4);
  const dim3 g(ew_grid(total)), blk(256);
  hipStream_t s = (hipStream_t)stream;
  const TView a = make_view(*dz), b = make_view(*z), c = make_view(*dy);
#define ACT_CASE(D1, T1, D2, T2, D3, T3) \
  if (dz_dtype == D1 && z_dtype == D2 && dy_dtype == D3) { hipLaunchKernelGGL((act_bwd_kernel<T1, T2, T3>), g, blk, 0, s, a, b, act, c, total); return vn_launch_status("act_bwd"); }
  ACT_CASE(VINET_F32, float, VINET_F32, float, VINET_F32, float)
  ACT_CASE(VINET_BF16, bf16_t, VINET_BF16, bf16_t, VINET_BF16, bf16_t)
  ACT_CASE(VINET_F32, float, VINET_F32, float, VINET_BF16, bf16_t)
  ACT_CASE(VINET_F32, float, VINET_BF16, bf16_t, VINET_BF16, bf16_t)
  ACT_CASE(VINET_BF16, bf16_t, VINET_F32, float, VINET_BF16, bf16_t)
#undef ACT_CASE
  vinet_set_error("act_bwd: unsupported dtype combination %d/%d/%d", dz_dtype, z_dtype, dy_dtype);
  return -1;
}

__global__ __launch_bounds__(256) void channel_sum_finalize_kernel(const float* __restrict__ partials, int rows, int C,
                                                                   int Cout, float* out, int accumulate) {
  const int j = blockIdx.x;
  double s = 0.0;
  const int per = C / Cout;   // channels folded onto output j: j, j + Cout, ...
  for (int e = threadIdx.x; e < rows * per; e += blockDim.x) {
    const int r = e / per, c = j + (e % per) * Cout;
    s += (double)partials[((long)r * 2) * C + c];
  }
  __shared__ double red[4];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    s = red[0] + red[1] + red[2] + red[3];
    if (accumulate) atomicAdd(out + j, (float)s);      // (not a scalar-cache read-modify-write: see bn_bwd_finalize_kernel)
    else out[j] = (float)s;
  }
}

extern "C" int vinet_channel_sum(const VinetTensor* x, int32_t dtype, float* workspace, int32_t Cout, float* out,
                                 int32_t accumulate, void* stream) {
  VN_CHECK_ARG(x && workspace && out && Cout > 0 && quad_ok(*x, esize(dtype)) && x->C % Cout == 0, "channel_sum: bad arguments");
  VinetAffine none = {nullptr, nullptr, 0};
  int rc = launch_channel_reduce<0>(x, nullptr, dtype, none, nullptr, nullptr, workspace, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(channel_sum_finalize_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, workspace,
                     stats_rows_for(view_voxels(*x)), x->C, Cout, out, accumulate);
  return vn_launch_status("channel_sum");
}
